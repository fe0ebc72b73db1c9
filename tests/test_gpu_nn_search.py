"""GPU parity of the kd search itself (iba_debug_nn = the search kernel's own per-lane search, lane_nn_visit, run to its end) against brute force
in float64 with the same expression order and the lowest-index tie rule. The search filters leaves in float32 and
confirms in double, so the queries are chosen to sit where a float cannot tell two points apart: exact midpoints of
point pairs (exact ties), midpoints moved by 1e-6 ... 1e-15 of the pair distance (near ties on both sides of what
float32 resolves), scan points themselves (distance 0), far-away queries, and random ones. Bar: index and squared
distance bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _brute(pts64, q):
    best_i = np.zeros(len(q), np.int64)
    best_d = np.zeros(len(q))
    for s in range(0, len(q), 256):
        qq = q[s:s + 256]
        dx = qq[:, None, 0] - pts64[None, :, 0]
        dy = qq[:, None, 1] - pts64[None, :, 1]
        dz = qq[:, None, 2] - pts64[None, :, 2]
        d2 = (dx * dx + dy * dy) + dz * dz                      # the reference's (and the kernel's) association order
        i = np.argmin(d2, axis=1)                               # first minimum = lowest index on exact ties
        best_i[s:s + 256] = i
        best_d[s:s + 256] = d2[np.arange(len(qq)), i]
    return best_i, best_d


def _queries(pts64, rng):
    n = len(pts64)
    a = rng.integers(0, n, 1500)
    nb = np.array([np.argsort(((pts64 - pts64[i]) ** 2).sum(1))[1] for i in a[:300]])     # true nearest neighbours of 300 points
    b = np.concatenate([nb, rng.integers(0, n, 1200)])
    mid = 0.5 * (pts64[a] + pts64[b])
    sep = pts64[b] - pts64[a]
    qs = [mid]                                                                            # exact ties between a and b (when nothing is closer)
    for eps in (1e-6, 1e-8, 1e-10, 1e-12, 1e-15):
        qs.append(mid + eps * sep)
        qs.append(mid - eps * sep)
    qs.append(pts64[rng.integers(0, n, 500)])                                             # distance exactly 0
    qs.append(pts64[rng.integers(0, n, 500)] + rng.normal(0, 1e-7, (500, 3)))             # inside the float rounding of the query
    lo, hi = pts64.min(0), pts64.max(0)
    qs.append(rng.uniform(lo, hi, (1500, 3)))
    qs.append(rng.uniform(lo - 30, hi + 30, (500, 3)))                                    # far outside the scan's box
    return np.concatenate(qs)


@pytest.mark.parametrize("mode", [1, 2, 3, 4])   # the query alone on either path of the kernel, and with a partner 1e-7 beside it
def test_kd_search_is_exact_where_float_is_blind(pkg, synth, abi, mode):
    prob, meta = synth.make_scene(n_frames=2, pts_per_frame=6000, n_keypoints=300, seed=41, new_mappoints=50, scan_kp=100)
    a = {k: v.copy() for k, v in prob.arrays.items()}
    pts = a["pts_xyz"].reshape(-1, 3)
    pts[6000 + 100:6000 + 400] = pts[6000 + 1000:6000 + 1300]          # frame 1 also carries 300 exact duplicates
    prob2 = abi.Problem(**a)
    h = pkg.IbaHandle(prob2, abi.reference_yaml_params())
    rng = np.random.default_rng(7 + mode)
    for f in (0, 1):
        pts64 = prob2.frame_points(f).astype(np.float64)
        q = _queries(pts64, rng)
        gi, gd = h.debug_nn(f, q, mode)
        bi, bd = _brute(pts64, q)
        assert np.array_equal(gd, bd), (f, np.flatnonzero(gd != bd)[:5])
        assert np.array_equal(gi.astype(np.int64), bi), (f, np.flatnonzero(gi != bi)[:5])
    h.close()


def test_left_over_entries_in_rounds_of_leaves_equal_leaf_by_leaf(pkg, synth, abi, ob, monkeypatch):
    """Dense scans (120 k points: 59-point leaves) leave entries to the tree search that the anchored lists cannot certify; round 5
    searches them in ROUNDS of leaves (walk to the next leaves first, scan them together, confirm once: wave_nn_round) instead of leaf by
    leaf (IBA_NN_ROUNDS=0, rounds 3-4). The visited set may only grow, the nearest neighbour may not change: the partial blocks of 24
    candidates — a tight poll, a wide one, and both kinds of evaluation — bit for bit, the neighbours against the oracle's counters."""
    import torch
    prob, meta = synth.make_scene(n_frames=3, pts_per_frame=120000, seed=11)
    p = abi.reference_yaml_params()
    rng = np.random.default_rng(11)
    xs = np.vstack([synth.perturb(meta["x_gt"], rng, n=16), synth.perturb(meta["x_gt"], rng, rot=3e-3, trans=0.03, scale_rel=4e-3, n=8)])
    blocks = {}
    for rounds in ("1", "0"):
        monkeypatch.setenv("IBA_NN_ROUNDS", rounds)
        h = pkg.IbaHandle(prob, p)
        out = []
        with torch.cuda.stream(torch.cuda.Stream()):
            st = torch.cuda.current_stream().cuda_stream
            for fn in (h.eval_full_partial, h.eval_cost_partial):
                d = torch.zeros(len(xs) * 64, dtype=torch.float64, device="cuda")
                fn(xs, d.data_ptr(), st)
                torch.cuda.synchronize()
                out.append(d.cpu().numpy().copy())
        left = h.nn_left_to_tree
        assert left > 0, "the scene must leave entries to the tree search"
        if rounds == "1":
            cost = h.eval_cost(xs[:3])
            for g, r in zip(cost, ob.Oracle(prob).eval_cost(p, xs[:3], nthreads=8)):
                assert (g.cnt_3d_3d, g.valid_cnt_3d_3d, g.n_corr) == (r.cnt_3d_3d, r.valid_cnt_3d_3d, r.n_corr) and abs(g.f2 - r.f2) <= 1e-10 * r.f2
        h.close()
        blocks[rounds] = out
    for a, b in zip(blocks["1"], blocks["0"]):
        assert np.array_equal(a, b, equal_nan=True), np.argwhere(a != b)[:5]
