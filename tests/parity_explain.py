"""Parity of the normal equations as a CHECKED invariant. The advertised gate is "every entry of H and b within
1e-10 of itself". Over thousands of random scenes some candidates miss it, for two reasons that round 3 only asserted in prose:
  (a) CONDITIONING OF A BLOCK. An IBA_PlaneFactor is two nested quotients (IBACalib2.hpp:163-183): Z0 = (n_c . p_c) / den with
      den = Cxz n_cx + Cyz n_cy + n_cz, then u = fx P1x / P1z + cx. When the viewing ray lies almost in the plane, den is a
      difference of nearly equal terms and the block amplifies the last bits of everything it is given. What round 4 measured
      (tools/device_vs_simulation.py, tools/three_ways_probe.py):
        * the device kernel and a CPU evaluation of the SAME formulas in the SAME order (oracle_block_three_ways) agree BIT FOR BIT
          on a block when their inputs are equal — the kernel computes what it says;
        * the inputs are not always equal: the plane normal n0 comes from the closed-form 3x3 eigen-solver (acos / cos), which the
          device evaluates with its own libm — the two normals differ by up to 1e-12 — and on some candidates the derivatives of
          R, t (Jet<6> on the device's host side, Dual<7> in the oracle) differ in the last bit;
        * the forward error of the analytic chain rule against long double is that of the oracle's duals (ratio 0.997 .. 1.19).
      So a deviating block is first re-evaluated by the oracle WITH THE DEVICE'S NORMAL (iba_debug_plane ->
      oracle_block_rows_with_normal), and what is left is held against a yardstick E that is MEASURED per block, the largest of
        - the forward error of the oracle's own double evaluation against long double (oracle_block_forward_error),
        - the block's sensitivity to one unit in the last place of x (oracle_block_sensitivity, long double),
        - its sensitivity to one unit in the last place of every derived input R, t, dR, dt, n0, signs at random, in the kernel's
          operation order (oracle_block_input_sensitivity):
      two correct double evaluations of the block cannot be expected to agree better than a small multiple of E.
  (b) CONDITIONING OF A SUM. An entry of H or b is a sum over 10^3..10^5 blocks; off-diagonal entries cancel. Two summation orders of
      the same terms differ by about eps sqrt(n) of the sum of the ABSOLUTE values of the terms, which for a cancelling entry is far
      more than 1e-10 of the entry.
explain() turns both into tests. Whenever a candidate misses the plain gate:
  (i)   every residual block's rows (r, J) must agree within ROW_FLOOR of the block's scale; a block beyond that must, once the
        oracle evaluates it with the device's plane normal (which may differ from the oracle's by at most NORMAL_TOL, or by what the
        measured accuracy of the point's covariance does to the normal), agree within ROW_FLOOR — or, a plane factor, EQUAL BIT FOR BIT
        the CPU evaluation of the kernel's formulas from the device's own inputs (iba_debug_cand, iba_debug_plane ->
        oracle_block_kernel_order) — or lie within C_FWD * E, E the block's measured yardstick;
  (ii)  with the deviating blocks removed from BOTH sides, every entry of the rebuilt H and b must be within 1e-10 of itself OR
        within SUM_TOL of the sum of the absolute values of its terms plus what row deviations at the ROW_FLOOR level — which (i)
        allows every block — propagate to, to first order (a block with |J| = 1e6 whose rows agree to 1e-12 of that scale may still
        move a small entry of its rows, and with it an entry of H, by more than 1e-10 of the entry: 4 of 6000 scenes);
  (iii) the candidate's own miss must be accounted for by (i) and (ii): either some block deviates, or the missed entries pass the
        sum-conditioning gate.
Anything else — a well-conditioned block that deviates, a deviation beyond the bound, an entry off by more than both gates — raises.
Used by tests/test_gpu_conditioning.py and tools/soak_parity.py (test infrastructure: imports the oracle)."""
import importlib
import os

import numpy as np

pkg_mod = importlib.import_module("spatial-temporal-lidar-camera-calibration_amd")

EPS = 2.220446049250313e-16
GATE = 1e-10          # per entry, relative to the entry itself
ENTRY_FLOOR = 1e-6    # entries below this fraction of the largest are compared against the largest instead
ROW_FLOOR = 1e-12     # rows of a well-conditioned block agree to this (of the block's scale)
C_FWD = 4.0           # rows of a deviating block, inputs equalised: within C_FWD * E, E = the block's measured yardstick (forward error, sensitivities)
NORMAL_TOL = 1e-11    # device vs oracle plane normal (unit vectors, sign-aligned), absolute: two libm evaluations of the closed-form eigenvector ...
C_NORMAL = 16.0       # ... or within this multiple of what the covariance's own accuracy (double vs long double, one-pass raw moments) does to the normal at that point (oracle_plane_normal_sensitivity)
SUM_TOL = 1e-12       # an entry of H / b: within this of the sum of the absolute values of its terms (two summation orders)


def entry_deviation(a, b):
    """worst |a - b| relative to |b| over the entries of b above ENTRY_FLOOR of its largest; the others relative to the largest"""
    a, b = np.asarray(a, float).ravel(), np.asarray(b, float).ravel()
    big = np.max(np.abs(b)) if b.size else 0.0
    if big == 0.0:
        return float(np.max(np.abs(a))) if a.size else 0.0
    den = np.where(np.abs(b) > ENTRY_FLOOR * big, np.abs(b), big)
    return float(np.max(np.abs(a - b) / den))


def normal_from_rows(r, J, bid, kind, params, skip=(), absolute=False):
    """H = sum_blocks w J^T J, b = sum_blocks w J^T r with the Huber IRLS weight of each block (w = 1 if |r| <= a else a / |r|;
    a = robust_kernel_delta for IBA_PlaneFactor blocks, robust_kernel_3ddelta for the 3d-3d ones: iba_local.cpp:263, 291).
    absolute: the sums of the absolute values of the terms instead (the scale two summation orders may differ by)."""
    H, b = np.zeros((7, 7)), np.zeros(7)
    if len(r) == 0:
        return H, b
    starts = np.concatenate([[0], np.where(np.diff(bid) != 0)[0] + 1, [len(bid)]])
    skip = set(int(s) for s in skip)
    for i in range(len(starts) - 1):
        lo, hi = starts[i], starts[i + 1]
        if int(bid[lo]) in skip:
            continue
        rr, JJ = r[lo:hi], J[lo:hi]
        nrm = np.sqrt(np.sum(rr * rr))
        a = params.robust_kernel_delta if kind[lo] == 0 else params.robust_kernel_3ddelta
        w = 1.0 if nrm <= a else a / nrm
        if absolute == "row_floor":
            # what row deviations of ROW_FLOOR of THIS block's scale — the level every block's rows are held to — can do to an entry, to
            # first order: d(J_i J_j) <= tol (|J_i| + |J_j|), d(J_i r) <= tol (|J_i| + |r|), tol = ROW_FLOOR * scale
            tol = ROW_FLOOR * max(float(np.max(np.abs(JJ))), float(np.max(np.abs(rr))), 1.0)
            cs = np.sum(np.abs(JJ), axis=0)
            H += w * tol * (cs[:, None] + cs[None, :])
            b += w * tol * (cs + float(np.sum(np.abs(rr))))
        elif absolute:
            H += w * (np.abs(JJ).T @ np.abs(JJ))
            b += w * (np.abs(JJ).T @ np.abs(rr))
        else:
            H += w * (JJ.T @ JJ)
            b += w * (JJ.T @ rr)
    return H, b


def entries_ok(a, ref, abs_sum, row_floor=None):
    """every entry within GATE of itself (ENTRY_FLOOR rule) or within SUM_TOL of the sum of the absolute values of its terms (+ what row
    deviations at the ROW_FLOOR level, which every block is allowed, propagate to: row_floor, from normal_from_rows(absolute="row_floor"))"""
    a, ref, abs_sum = (np.asarray(v, float).ravel() for v in (a, ref, abs_sum))
    big = np.max(np.abs(ref)) if ref.size else 0.0
    den = np.where(np.abs(ref) > ENTRY_FLOOR * big, np.abs(ref), big)
    d = np.abs(a - ref)
    allow = SUM_TOL * abs_sum + (np.asarray(row_floor, float).ravel() if row_floor is not None else 0.0)
    ok = (d <= GATE * den) | (d <= allow)
    return bool(np.all(ok)), float(np.max(np.where(ok, 0.0, d / np.maximum(den, 1e-300)))) if ref.size else 0.0


def explain(h, o, p, x, nthreads=1):
    """-> dict(status="clean" | "explained", flagged=<blocks>, min_cond=.., worst_entry=.., cancelling_entries=<bool>); raises
    AssertionError when a deviation is NOT explained. h: device handle, o: oracle, p: parameters, x: one candidate."""
    g = h.eval_normal(x)[0]
    r = o.eval_normal(p, x, nthreads=nthreads)[0]
    assert g.counts() == r.counts(), (g.counts(), r.counts())
    worst = max(entry_deviation(g.H_np(), r.H_np()), entry_deviation(g.b_np(), r.b_np()))
    if worst <= GATE:
        return {"status": "clean", "flagged": 0, "worst_entry": worst}
    h.build_problem(x)
    o.build_problem(p, x)
    rg, Jg, bg, kg = h.eval_residuals(x)
    ro, Jo, bo, ko, _ = o.eval_residuals(x)
    assert np.array_equal(kg, ko) and np.array_equal(bg, bo), "the two sides hold different residual blocks"
    cond = o.block_conditioning(x)
    fwd = np.maximum(np.maximum(o.block_forward_error(x), o.block_sensitivity(x)), o.block_input_sensitivity(x))   # the yardstick E of a block: see (a)
    starts = np.concatenate([[0], np.where(np.diff(bo) != 0)[0] + 1, [len(bo)]])
    assert len(cond) == len(starts) - 1 == len(fwd)
    flagged, min_cond, worst_ratio, worst_normal, n_by_normal, n_bitwise, worst_ratio_bitwise, cand58 = [], 1.0, 0.0, 0.0, 0, 0, 0.0, None
    pts_of = {}
    for i in range(len(starts) - 1):
        lo, hi = starts[i], starts[i + 1]
        scale = max(float(np.max(np.abs(Jo[lo:hi]))), float(np.max(np.abs(ro[lo:hi]))), 1.0)
        dev = max(float(np.max(np.abs(Jg[lo:hi] - Jo[lo:hi]))), float(np.max(np.abs(rg[lo:hi] - ro[lo:hi])))) / scale
        if dev <= ROW_FLOOR:
            continue
        flagged.append(int(bo[lo]))
        min_cond = min(min_cond, float(cond[i]))
        # a deviating block: first give both sides the same INPUT — the device's own plane normal substituted into the oracle's block
        dev2, e2, dn = dev, float(fwd[i]), 0.0
        got = o.block_normal(int(bo[lo]))
        if got is not None:
            n_o, q, frame = got
            if frame not in pts_of:
                pts_of[frame] = h.problem.frame_points(frame).astype(np.float64)
            hit = np.where(np.all(pts_of[frame] == q[None, :], axis=1))[0]
            assert len(hit) >= 1, "block %d: its scan point is not in frame %d" % (i, frame)
            n_d = h.debug_plane(frame, int(hit[0]), which=1)[0]
            n_raw = n_d.copy()
            if np.dot(n_d, n_o) < 0:
                n_d = -n_d                                            # the sign of an eigenvector is arbitrary
            dn = float(np.max(np.abs(n_d - n_o)))
            worst_normal = max(worst_normal, dn)
            if dn > NORMAL_TOL:   # beyond the usual agreement: the eigen-solver itself must be that sensitive at this point, measured
                sn = o.plane_normal_sensitivity(frame, int(hit[0]), p.neigh_radius, p.neigh_max_pts)
                assert dn <= C_NORMAL * sn, "block %d: the device's plane normal differs from the oracle's by %.2e; the covariance's own accuracy moves it by %.2e only" % (i, dn, sn)
            r2, J2, e2 = o.block_rows_with_normal(int(bo[lo]), x, n_d, hi - lo)
            if ko[lo] == 1 and np.max(np.abs(r2 + rg[lo:hi])) < np.max(np.abs(r2 - rg[lo:hi])):
                r2, J2 = -r2, -J2                                     # a flipped normal flips the 1-d point-to-plane residual: harmless for H, b
            scale2 = max(float(np.max(np.abs(J2))), float(np.max(np.abs(r2))), 1.0)
            dev2 = max(float(np.max(np.abs(Jg[lo:hi] - J2))), float(np.max(np.abs(rg[lo:hi] - r2)))) / scale2
        if dev2 <= ROW_FLOOR:
            n_by_normal += 1                                          # the normal's last bits were all of it
            continue
        # the strongest statement first: a plane-factor block evaluated on the CPU in the kernel's operation order from the DEVICE's own
        # inputs (Sim3Exp and its derivatives as the library's host side computes them, the device's plane normal) must equal the device's
        # rows BIT FOR BIT — then everything that separates device and oracle on this block is the last bits of those inputs
        if ko[lo] == 0 and got is not None:
            if cand58 is None:
                cand58 = pkg_mod.debug_cand(x)
            sim = o.block_kernel_order(int(bo[lo]), cand58, n_raw, hi - lo)
            if sim is not None and np.array_equal(sim[:, 0], rg[lo:hi]) and np.array_equal(sim[:, 1:], Jg[lo:hi]):
                n_bitwise += 1
                worst_ratio_bitwise = max(worst_ratio_bitwise, dev2 / max(e2, 1e-300))
                continue
        bound2 = C_FWD * e2
        if dev2 > bound2 and os.environ.get("IBA_EXPLAIN_DEBUG"):
            tw = o.block_three_ways(int(bo[lo]), x, hi - lo)
            if tw is not None:
                dvc = np.concatenate([rg[lo:hi, None], Jg[lo:hi]], 1)
                print("block %d debug: |device - long double| %.3e, |oracle - long double| %.3e, |device-order simulation - long double| %.3e, |device - its simulation| %.3e (of scale %.3e)"
                      % (i, np.max(np.abs(dvc - tw[1])), np.max(np.abs(tw[0] - tw[1])), np.max(np.abs(tw[2] - tw[1])), np.max(np.abs(dvc - tw[2])), scale), flush=True)
        assert dev2 <= bound2, ("block %d (kind %d) deviates by %.2e of its scale; with the device's own plane normal (|dn| = %.1e) in the oracle's block still %.2e, "
                                "beyond %.0f x the block's measured yardstick %.2e (conditioning %.2e): not explained" % (i, ko[lo], dev, dn, dev2, C_FWD, e2, cond[i]))
        worst_ratio = max(worst_ratio, dev2 / max(e2, 1e-300))
    # with the deviating blocks removed: 1e-10 per entry, or the accuracy of a sum of that many cancelling terms
    Hg, bgv = normal_from_rows(rg, Jg, bg, kg, p, skip=flagged)
    Ho, bov = normal_from_rows(ro, Jo, bo, ko, p, skip=flagged)
    Ha, ba = normal_from_rows(ro, Jo, bo, ko, p, skip=flagged, absolute=True)
    Hf, bf = normal_from_rows(ro, Jo, bo, ko, p, skip=flagged, absolute="row_floor")
    okH, wH = entries_ok(Hg, Ho, Ha, Hf)
    okb, wb = entries_ok(bgv, bov, ba, bf)
    assert okH and okb, "with the %d ill-conditioned block(s) removed an entry is still off by %.2e of itself and by more than %.0e of its terms' absolute sum" % (len(flagged), max(wH, wb), SUM_TOL)
    cancelling = False
    if not flagged:   # the miss must then be the summation order on cancelling entries — of the DEVICE's own sums against the oracle's
        Hfa, bfa = normal_from_rows(ro, Jo, bo, ko, p, absolute=True)
        Hff, bff = normal_from_rows(ro, Jo, bo, ko, p, absolute="row_floor")
        ok1, w1 = entries_ok(g.H_np(), r.H_np(), Hfa, Hff)
        ok2, w2 = entries_ok(g.b_np(), r.b_np(), bfa, bff)
        assert ok1 and ok2, "H / b miss the gate by %.2e although no residual block deviates and the entries do not cancel: a summation defect" % max(w1, w2)
        cancelling = True
    # (the rows are the same numbers the device summed: its own H rebuilt from its rows)
    Hfull, _ = normal_from_rows(rg, Jg, bg, kg, p)
    assert np.max(np.abs(Hfull - g.H_np())) <= 1e-9 * np.max(np.abs(g.H_np())), "iba_eval_residuals and iba_eval_normal disagree"
    return {"status": "explained", "flagged": len(flagged), "min_cond": min_cond, "worst_entry": worst, "cancelling_entries": cancelling, "worst_dev_over_forward_error": worst_ratio, "worst_normal_difference": worst_normal, "settled_by_the_normal": n_by_normal, "bit_identical_to_the_cpu_evaluation_of_the_kernel": n_bitwise, "their_worst_deviation_over_yardstick": worst_ratio_bitwise}
