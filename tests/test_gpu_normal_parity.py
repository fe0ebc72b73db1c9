"""GPU parity of the Jacobian path (BuildProblem + residual blocks + Huber IRLS normal equations,
iba_local.cpp:145-323, IBACalib2.hpp:152-184, 570-584, 611-625) through the C-ABI against the oracle,
whose Jacobians come from forward-mode duals exactly like the reference's Ceres/g2o autodiff.

Bars: factor/residual counts bit-exact; H, b, cost, chi2 within 1e-9 relative to the largest entry
(analytic chain rule vs dual numbers + summation order)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cmp_normal(g, o, rel=1e-9):
    assert g.counts() == o.counts(), (g.counts(), o.counts())
    Hg, Ho = g.H_np(), o.H_np()
    assert np.allclose(Hg, Hg.T)
    assert np.max(np.abs(Hg - Ho)) <= rel * np.max(np.abs(Ho)), np.max(np.abs(Hg - Ho)) / np.max(np.abs(Ho))
    # per-entry relative check on the diagonal (entries span 1e5..1e10)
    assert np.allclose(np.diag(Hg), np.diag(Ho), rtol=1e-8, atol=0)
    assert np.max(np.abs(g.b_np() - o.b_np())) <= rel * max(np.max(np.abs(o.b_np())), 1e-30) * 10
    assert abs(g.cost - o.cost) <= rel * abs(o.cost) + 1e-300
    assert abs(g.chi2 - o.chi2) <= rel * abs(o.chi2) + 1e-300


def test_eval_normal_matches_oracle(pkg, synth, abi, ob, scene_small):
    prob, meta = scene_small
    p = abi.reference_yaml_params()
    h = pkg.IbaHandle(prob, p)
    o = ob.Oracle(prob)
    rng = np.random.default_rng(11)
    xs = np.vstack([meta["x_gt"][None], synth.perturb(meta["x_gt"], rng, n=3), synth.perturb(meta["x_gt"], rng, rot=0.01, trans=0.05, scale_rel=0.02, n=2)])
    g = h.eval_normal(xs)
    oo = o.eval_normal(p, xs)
    assert g[0].n_factor_3d2d > 100 and g[0].n_factor_p2pl + g[0].n_factor_p2pt > 100
    for a, b in zip(g, oo):
        _cmp_normal(a, b)
    h.close()


def test_frozen_association(pkg, synth, abi, ob, scene_small):
    """iba_build_problem at x0, then residual blocks evaluated at other x (what Ceres does inside Solve)."""
    prob, meta = scene_small
    p = abi.reference_yaml_params()
    h = pkg.IbaHandle(prob, p)
    o = ob.Oracle(prob)
    rng = np.random.default_rng(12)
    x0 = synth.perturb(meta["x_gt"], rng, n=1)[0]
    h.build_problem(x0)
    o.build_problem(p, x0)
    xs = synth.perturb(x0, rng, rot=2e-3, trans=2e-2, scale_rel=5e-3, n=5)
    for a, b in zip(h.eval_factors(xs), o.eval_factors(p, xs)):
        _cmp_normal(a, b)
    h.close()


def test_local_params_differ_from_cost_params(pkg, synth, abi, ob, scene_small):
    """neigh_radius/neigh_max_pts different from norm_radius/norm_max_pts -> second plane cache."""
    prob, meta = scene_small
    p = abi.reference_yaml_params()
    p.neigh_radius = 0.45
    p.neigh_max_pts = 20
    p.local_norm_reg_threshold = 0.03
    p.robust_kernel_3ddelta = 0.05
    p.robust_kernel_delta = 1.0
    h = pkg.IbaHandle(prob, p)
    o = ob.Oracle(prob)
    _cmp_normal(h.eval_normal(meta["x_gt"])[0], o.eval_normal(p, meta["x_gt"])[0])
    h.close()


def test_eval_full_equals_separate_calls(pkg, synth, abi, scene_small):
    """iba_eval_full (one fused pass) must reproduce iba_eval_cost + iba_eval_normal: counters bit for bit;
    sums to the last ulps (records and residual blocks are summed in a different, but fixed, grouping)."""
    prob, meta = scene_small
    h = pkg.IbaHandle(prob, abi.reference_yaml_params())
    rng = np.random.default_rng(21)
    xs = np.vstack([synth.perturb(meta["x_gt"], rng, n=3), synth.perturb(meta["x_gt"], rng, rot=0.03, trans=0.1, scale_rel=0.05, n=2)])
    cf, nf = h.eval_full(xs)
    cs, ns = h.eval_cost(xs), h.eval_normal(xs)
    for a, b in zip(cf, cs):
        da, db = a.as_dict(), b.as_dict()
        for k in da:
            if k in ("f1", "f2", "C"):
                assert (np.isnan(da[k]) and np.isnan(db[k])) or abs(da[k] - db[k]) <= 1e-14 * abs(db[k]), (k, da[k], db[k])
            else:
                assert da[k] == db[k], k
    for a, b in zip(nf, ns):
        # same residual blocks, summed in a different (fixed) order: the fused mode's work list also holds cost-only keypoints
        assert a.counts() == b.counts()
        assert np.max(np.abs(a.H_np() - b.H_np())) <= 1e-12 * np.max(np.abs(b.H_np())) and np.max(np.abs(a.b_np() - b.b_np())) <= 1e-11 * np.max(np.abs(b.b_np()))
        assert abs(a.cost - b.cost) <= 1e-13 * abs(b.cost)
    # and it is run-to-run reproducible bit for bit
    cf2, nf2 = h.eval_full(xs)
    assert all(np.array_equal(a.H_np(), b.H_np()) and a.cost == b.cost for a, b in zip(nf, nf2)) and all(a.f1 == b.f1 and a.f2 == b.f2 for a, b in zip(cf, cf2))
    h.close()


def test_per_residual_rows_match_oracle(pkg, synth, abi, ob, scene_small):
    """iba_eval_residuals: every residual and Jacobian row of the frozen problem (what the Ceres/g2o adaptors
    consume), in BuildProblem order, against the oracle's dual-number rows."""
    prob, meta = scene_small
    p = abi.reference_yaml_params()
    h = pkg.IbaHandle(prob, p)
    o = ob.Oracle(prob)
    rng = np.random.default_rng(31)
    x0 = synth.perturb(meta["x_gt"], rng, n=1)[0]
    h.build_problem(x0)
    o.build_problem(p, x0)
    x = synth.perturb(x0, rng, rot=1e-3, trans=1e-2, scale_rel=2e-3, n=1)[0]
    rg, Jg, bg, kg = h.eval_residuals(x)
    ro, Jo, bo, ko, _ = o.eval_residuals(x)
    assert len(rg) == len(ro) > 1000 and np.array_equal(kg, ko) and np.array_equal(bg, bo)
    assert np.allclose(rg, ro, rtol=1e-9, atol=1e-9)
    assert np.allclose(Jg, Jo, rtol=1e-8, atol=1e-8 * np.abs(Jo).max())
    h.close()


def test_plane_blocks_as_the_g2o_edge_sees_them(pkg, synth, abi, ob, scene_small):
    """IBAPlaneEdge (IBACalib.hpp:103-140), the per-keypoint g2o twin of IBA_PlaneFactor: a 20-dimensional error — the block's 2 NConv
    residuals, zero-padded — and its 20 x 7 Jacobian. The rows iba_eval_residuals hands out for a plane-factor block, padded the same
    way, against the oracle's edge (oracle_eval_plane_edge20): what a g2o user of the per-edge form would bind to."""
    prob, meta = scene_small
    p = abi.reference_yaml_params()
    h = pkg.IbaHandle(prob, p)
    o = ob.Oracle(prob)
    rng = np.random.default_rng(57)
    x0 = synth.perturb(meta["x_gt"], rng, n=1)[0]
    h.build_problem(x0)
    o.build_problem(p, x0)
    x = synth.perturb(x0, rng, rot=1e-3, trans=1e-2, scale_rel=2e-3, n=1)[0]
    rg, Jg, bg, kg = h.eval_residuals(x)
    starts = np.concatenate([[0], np.where(np.diff(bg) != 0)[0] + 1, [len(bg)]])
    cond = o.block_conditioning(x)
    seen = 0
    for i in range(len(starts) - 1):
        lo, hi = starts[i], starts[i + 1]
        if kg[lo] != 0 or cond[i] < 1e-2:      # (ill-conditioned blocks have their own test: tests/test_gpu_conditioning.py)
            continue
        edge = o.plane_edge20(int(bg[lo]), x)
        assert edge is not None and hi - lo <= 20 and (hi - lo) % 2 == 0
        e20, J20 = np.zeros(20), np.zeros((20, 7))
        e20[: hi - lo] = rg[lo:hi]; J20[: hi - lo] = Jg[lo:hi]
        scale = max(np.abs(edge[1]).max(), np.abs(edge[0]).max(), 1.0)
        assert np.max(np.abs(e20 - edge[0])) <= 1e-10 * scale and np.max(np.abs(J20 - edge[1])) <= 1e-10 * scale
        assert not edge[0][hi - lo:].any() and not edge[1][hi - lo:].any()
        seen += 1
    assert seen > 200
    h.close()


def test_mfma_factor_kernel_matches_oracle(pkg, synth, abi, ob, scene_small, monkeypatch):
    """IBA_FACTOR_MFMA=1: the normal equations accumulated on v_mfma_f64_16x16x4_f64 (rank-1 rows through LDS) instead of the
    VALU: same bars as the default kernel, and equal to it to summation order."""
    prob, meta = scene_small
    p = abi.reference_yaml_params()
    xs = np.vstack([meta["x_gt"][None], synth.perturb(meta["x_gt"], np.random.default_rng(11), n=3)])
    hv = pkg.IbaHandle(prob, p)
    monkeypatch.setenv("IBA_FACTOR_MFMA", "1")
    hm = pkg.IbaHandle(prob, p)
    monkeypatch.delenv("IBA_FACTOR_MFMA")
    o = ob.Oracle(prob)
    gm, gv, r = hm.eval_normal(xs), hv.eval_normal(xs), o.eval_normal(p, xs)
    for a, v, b in zip(gm, gv, r):
        assert a.counts() == b.counts() == v.counts()
        scale = np.abs(b.H_np()).max()
        assert np.max(np.abs(a.H_np() - b.H_np())) <= 1e-9 * scale and np.max(np.abs(a.b_np() - b.b_np())) <= 1e-9 * np.abs(b.b_np()).max()
        assert np.max(np.abs(a.H_np() - v.H_np())) <= 1e-12 * scale
        assert abs(a.cost - b.cost) <= 1e-10 * abs(b.cost) and abs(a.chi2 - b.chi2) <= 1e-10 * abs(b.chi2)
    hm.build_problem(xs[1])
    o.build_problem(p, xs[1])
    a, b = hm.eval_factors(xs[2])[0], o.eval_factors(p, xs[2])[0]
    assert a.counts() == b.counts() and np.max(np.abs(a.H_np() - b.H_np())) <= 1e-9 * np.abs(b.H_np()).max()
    hm.close()
    hv.close()


def test_whitened_block_reproduces_the_frozen_problem(pkg, synth, abi, ob, scene_small):
    """iba_eval_whitened (the Ceres / g2o adaptors' only call): J^T J, J^T r and |r|^2 / 2 against the ORACLE's H, b and cost
    of the same frozen problem, at the build point and away from it."""
    import ctypes as C
    prob, meta = scene_small
    p = abi.reference_yaml_params()
    h = pkg.IbaHandle(prob, p)
    o = ob.Oracle(prob)
    xs = synth.perturb(meta["x_gt"], np.random.default_rng(17), n=3)
    h.build_problem(xs[0])
    o.build_problem(p, xs[0])
    for x in xs:
        x = np.ascontiguousarray(x)
        r = np.zeros(8)
        J = np.zeros((8, 7))
        st = h.lib.iba_eval_whitened(h.h, x.ctypes.data_as(C.c_void_p), r.ctypes.data_as(C.c_void_p), J.ctypes.data_as(C.c_void_p))
        assert st == 0
        ref = o.eval_factors(p, x)[0]
        H, b = ref.H_np(), ref.b_np()
        assert np.max(np.abs(J.T @ J - H)) <= 1e-9 * np.abs(H).max()
        assert np.max(np.abs(J.T @ r - b)) <= 1e-9 * np.abs(b).max()
        assert abs(0.5 * (r @ r) - ref.cost) <= 1e-9 * ref.cost
    h.close()
