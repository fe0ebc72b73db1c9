"""TEST INFRASTRUCTURE — slow, independent restatement of the rules csrc/iba_mads.hpp claims to follow, for diffing iterate
sequences (the reference's set-up of NOMAD 4 is iba_global.cpp:551-602: 7 variables, bounds, OBJ + 3 PB
constraints from BALoss::eval_x :377-396, OrthoMADS 2N, INITIAL_POLL_SIZE, MIN_MESH_SIZE, MAX_BB_EVAL, VNS).

NOMAD itself is absent from this image and from /root/reference (third party, version unpinned: README.md:53), so this is NOT
pinned to NOMAD: parity unpinned beyond the published rules. What it pins is that the C++ driver implements the rules it states:

  * mesh / frame sizes  delta_l = frame0 * 4^-l, Delta_l = frame0 * 2^-l  (Audet & Dennis 2006), frame0 = min(init_frame,
    0.1 * box width); stop when every delta < min_mesh;
  * poll = OrthoMADS 2N (Abramson, Audet, Dennis & Le Digabel 2009): Halton point t (bases 2, 3, 5, 7, 11, 13, 17; t starts at
    7 + 101 |seed| and advances by one per direction set) -> unit vector q -> Householder H = I - 2 q q^T; columns scaled to
    infinity norm = frame size, rounded to the mesh, +- each; `bases_per_poll` such sets per poll centre;
  * progressive barrier (Audet & Dennis 2009): h = sum max(c_j, 0)^2, feasible incumbent (h = 0, lowest f) and infeasible
    incumbent (h <= h_max, non-dominated), polls around BOTH; dominating success -> coarser mesh, improving -> same mesh and
    h_max tightened to the largest h seen below the infeasible incumbent's, failure -> finer mesh and h_max = h of the
    infeasible incumbent;
  * one speculative point along the last dominating step; a cache keyed on x rounded to min_mesh / 16; the evaluation budget
    cuts the last batch short;
  * variable-neighbourhood restarts (Audet, Bechard & Le Digabel 2008): shake the incumbent by k half-frames along the next
    Halton direction, new descent with a fresh barrier, keep it only if it beats the incumbent by more than 1e-7 relative.

Everything is plain Python floats (IEEE double, the same operations in the same order as the rules state them), so for a
deterministic black box the sequence of evaluated points is bit-identical to the C++ driver's.
Only tests/ may import this module.
"""
import math

N = 7
PRIMES = (2, 3, 5, 7, 11, 13, 17)
INF = float("inf")


def constraint_violation(c):
    h = 0.0
    for cj in c:
        if not (cj <= 0):
            if cj != cj:
                return INF
            h += cj * cj
    return h


def halton(t, p):
    f, r = 1.0, 0.0
    while t > 0:
        f /= p
        r += f * (t % p)
        t //= p
    return r


def _round_half_away(v):
    return int(math.floor(abs(v) + 0.5)) * (1 if v >= 0 else -1)


def _nearbyint(v):
    return float(round(v))   # Python rounds half to even, as the default IEEE mode does


class Point:
    __slots__ = ("x", "f", "c", "h")

    def __init__(self, x):
        self.x = list(x)
        self.f, self.c, self.h = INF, [0.0, 0.0, 0.0], INF

    def copy(self):
        p = Point(self.x)
        p.f, p.c, p.h = self.f, list(self.c), self.h
        return p


def default_options(x0):
    lb = (-0.1, -0.1, -0.1, -0.3, -0.3, -0.3, -1.0)   # iba_calib_global.yml:39-40
    return dict(max_bb_eval=5000, lb=[x0[i] + lb[i] for i in range(N)], ub=[x0[i] - lb[i] for i in range(N)], init_frame=[0.5] * N, min_mesh=1e-6, seed=0,
                bases_per_poll=2, max_batch=64, speculative=True, frame_box_fraction=0.1, vns_max_idle=6, vns_max_k=6, vns_amplitude=0.5, vns_mesh_index=0)


def minimize(x0, opt, eval_batch):
    """eval_batch(list of 7-lists) -> list of (f, (c1, c2, c3)). Returns (result dict, trace): trace = every point handed to the
    black box, in order, as (x tuple, f)."""
    o = dict(default_options(x0))
    o.update(opt)
    lbv, ubv = o["lb"], o["ub"]
    cache = {}
    trace = []
    stats = dict(evaluations=0, iterations=0, batches=0, cache_hits=0, restarts=0, mesh_index=0)
    state = dict(t=7 + abs(int(o["seed"])) * 101)
    frame0 = []
    for i in range(N):
        box = o["frame_box_fraction"] * (ubv[i] - lbv[i])
        frame0.append(min(o["init_frame"][i], box) if box > 0 else o["init_frame"][i])

    def key_of(x):
        return tuple(_round_half_away(x[i] / (o["min_mesh"] * 0.0625)) for i in range(N))

    def clamp(x):
        for i in range(N):
            x[i] = min(max(x[i], lbv[i]), ubv[i])

    def run(trial):
        todo, keys = [], []
        for i, p in enumerate(trial):
            p.f, p.h = INF, INF
            k = key_of(p.x)
            if k in cache:
                q = cache[k]
                p.x, p.f, p.c, p.h = list(q.x), q.f, list(q.c), q.h
                stats["cache_hits"] += 1
                continue
            if k not in keys:
                todo.append(i)
                keys.append(k)
        budget = o["max_bb_eval"] - stats["evaluations"]
        if len(todo) > budget:
            todo = todo[:max(budget, 0)]
        for s in range(0, len(todo), o["max_batch"]):
            chunk = todo[s:s + o["max_batch"]]
            out = eval_batch([list(trial[i].x) for i in chunk])
            stats["batches"] += 1
            for i, (f, c) in zip(chunk, out):
                p = trial[i]
                p.f, p.c = f, list(c)
                p.h = constraint_violation(p.c) if p.f == p.f else INF
                cache[key_of(p.x)] = p.copy()
                stats["evaluations"] += 1
                trace.append((tuple(p.x), p.f))

    def halton_dir():
        q = [2.0 * halton(state["t"], PRIMES[i]) - 1.0 for i in range(N)]
        nq = 0.0
        for v in q:
            nq += v * v
        state["t"] += 1
        nq = math.sqrt(nq)
        if not (nq > 1e-12):
            q = [1.0] + [0.0] * (N - 1)
            nq = 1.0
        return [v / nq for v in q]

    def descent(start, l0):
        inc = dict(F=None, I=None)      # feasible / infeasible incumbent
        bar = dict(hmax=INF)
        seen = []

        def absorb(pts):
            success = 0
            for p in pts:
                if not (p.h < INF) or p.f != p.f:
                    continue
                seen.append(p.copy())
                if p.h == 0.0:
                    if inc["F"] is None or p.f < inc["F"].f:
                        inc["F"] = p.copy()
                        success = 2
                elif p.h <= bar["hmax"]:
                    xi = inc["I"]
                    if xi is None:
                        inc["I"] = p.copy()
                        success = max(success, 1)
                    elif (p.h < xi.h and p.f <= xi.f) or (p.h <= xi.h and p.f < xi.f):
                        inc["I"] = p.copy()
                        success = 2
                    elif p.h < xi.h:
                        success = max(success, 1)
            return success

        c0 = Point(start)
        clamp(c0.x)
        first = [c0.copy()]
        run(first)
        absorb(first)
        l = l0
        last_dir, have_dir = [0.0] * N, False
        while True:
            frame = [frame0[i] * math.ldexp(1.0, -l) for i in range(N)]
            mesh = [frame0[i] * math.ldexp(1.0, -2 * l) for i in range(N)]
            fine = all(m < o["min_mesh"] for m in mesh)
            stats["mesh_index"] = l
            if fine or stats["evaluations"] >= o["max_bb_eval"]:
                break
            stats["iterations"] += 1
            trial = []

            def add_poll(ctr):
                for _ in range(o["bases_per_poll"]):
                    q = halton_dir()
                    for j in range(N):
                        col = [(1.0 if i == j else 0.0) - 2.0 * q[i] * q[j] for i in range(N)]
                        cmax = 0.0
                        for v in col:
                            cmax = max(cmax, abs(v))
                        for sgn in (1.0, -1.0):
                            p = Point(ctr.x)
                            for i in range(N):
                                step = sgn * frame[i] * col[i] / cmax
                                p.x[i] = ctr.x[i] + mesh[i] * _nearbyint(step / mesh[i])
                            clamp(p.x)
                            trial.append(p)

            if inc["F"] is not None:
                add_poll(inc["F"])
            if inc["I"] is not None:
                add_poll(inc["I"])
            if inc["F"] is None and inc["I"] is None:
                add_poll(c0)
            if o["speculative"] and have_dir:
                ctr = inc["F"] if inc["F"] is not None else inc["I"]
                p = Point(ctr.x)
                for i in range(N):
                    p.x[i] = ctr.x[i] + mesh[i] * _nearbyint(last_dir[i] / mesh[i])
                clamp(p.x)
                trial.append(p)
            oldF, oldI = inc["F"], inc["I"]
            run(trial)
            success = absorb(trial)
            have_dir = False
            if success == 2:
                via_f = inc["F"] is not None and (oldF is None or inc["F"].f < oldF.f)
                now = inc["F"] if via_f else inc["I"]
                was = (oldF if oldF is not None else now) if via_f else (oldI if oldI is not None else now)
                nd = 0.0
                for i in range(N):
                    last_dir[i] = now.x[i] - was.x[i]
                    nd += last_dir[i] * last_dir[i]
                have_dir = nd > 0
                l = max(l - 1, 0)
            elif success == 1:
                xi = inc["I"]
                hm = 0.0
                for p in seen:
                    if p.h < xi.h and p.h > hm:
                        hm = p.h
                bar["hmax"] = hm if hm > 0 else xi.h
                if xi is not None and xi.h > bar["hmax"]:
                    pick = None
                    for p in seen:
                        if p.h > 0 and p.h <= bar["hmax"] and (pick is None or p.f < pick.f):
                            pick = p
                    inc["I"] = pick.copy() if pick is not None else None
            else:
                if inc["I"] is not None:
                    bar["hmax"] = inc["I"].h
                l += 1
        if inc["F"] is not None:
            return inc["F"], True
        if inc["I"] is not None:
            return inc["I"], False
        best = c0.copy()
        best.f, best.h = INF, INF
        for p in seen:
            if p.h < best.h:
                best = p
        return best, False

    def better(a, fa, b, fb):
        if fa != fb:
            return fa
        if fa:
            return a.f < b.f - 1e-7 * abs(b.f) - 1e-300
        return a.h < b.h * (1.0 - 1e-7) or (a.h == b.h and a.f < b.f - 1e-7 * abs(b.f))

    best, best_feas = descent(list(x0), 0)
    k, idle = 1, 0
    while stats["evaluations"] < o["max_bb_eval"] and idle < o["vns_max_idle"]:
        u = halton_dir()
        umax = 0.0
        for v in u:
            umax = max(umax, abs(v))
        start = [best.x[i] + float(k) * o["vns_amplitude"] * frame0[i] * u[i] / umax for i in range(N)]
        loc, loc_feas = descent(start, o["vns_mesh_index"])
        stats["restarts"] += 1
        if better(loc, loc_feas, best, best_feas):
            best, best_feas, k, idle = loc, loc_feas, 1, 0
        else:
            k, idle = min(k + 1, o["vns_max_k"]), idle + 1
    res = dict(stats)
    res.update(x=list(best.x), f=best.f, c=list(best.c), feasible=1 if best_feas else 0, stop_reason=2 if stats["evaluations"] >= o["max_bb_eval"] else 1)
    return res, trace


# the analytic black boxes of iba_mads_selftest (csrc/iba_capi.hip), restated
_A = (0.3, -0.2, 0.1, 0.25, -0.15, 0.05, 9.5)


def selftest_box(problem):
    def ev(X):
        out = []
        for x in X:
            f, c = 0.0, [-1.0, -1.0, -1.0]
            if problem == 0:
                for i in range(N):
                    f += (1.0 + i) * (x[i] - _A[i]) * (x[i] - _A[i])
            elif problem == 1:
                f = (x[0] - 1.0) * (x[0] - 1.0)
                for i in range(1, N):
                    f += (x[i] - _A[i]) * (x[i] - _A[i])
                c[0] = x[0] - 0.5
            elif problem == 3:
                two_pi = 6.283185307179586
                for i in range(N):
                    d = x[i] - _A[i] - 0.0123 * (i + 1)
                    f += 2.0 * d * d + 0.3 * (1.0 - math.cos(two_pi * d / 0.08))
            else:
                m, s1 = 0.0, 0.0
                for i in range(N):
                    d = abs(x[i] - _A[i])
                    m = max(m, d)
                    s1 += d
                f = m + 0.1 * s1
                c[0] = 0.2 - x[1]
                c[1] = x[3] + x[4] - 0.05
            out.append((f, tuple(c)))
        return out
    return ev
