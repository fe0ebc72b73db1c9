"""The oracle's restated evaluation path: committed golden outputs (regression pin), independent
brute-force cross-checks of the association, Jacobians against central differences, and properties."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden", "path_small_scene.npz")


@pytest.fixture(scope="module")
def gold(abi):
    z = np.load(GOLD)
    prob = abi.Problem(**{k[6:]: z[k] for k in z.files if k.startswith("scene_")})
    return z, prob


def test_golden_cost_and_normal(ob, abi, gold):
    z, prob = gold
    p = abi.reference_yaml_params()
    o = ob.Oracle(prob)
    cost = o.eval_cost(p, z["xs"])
    ci = np.array([[c.valid_cnt_3d_2d, c.cnt_3d_2d, c.cnt_3d_3d, c.valid_cnt_3d_3d, c.valid_pl_3d_3d, c.valid_pt_3d_3d, c.frames_used, c.n_corr] for c in cost])
    assert np.array_equal(ci, z["cost_i"])
    cf = np.array([[c.f1, c.f2, c.C] for c in cost])
    assert np.allclose(cf, z["cost_f"], rtol=1e-12, atol=0, equal_nan=True)
    nrm = o.eval_normal(p, z["xs"])
    ni = np.array([[n.n_factor_3d2d, n.n_factor_p2pl, n.n_factor_p2pt, n.n_residuals, n.frames_used, n.n_corr] for n in nrm])
    assert np.array_equal(ni, z["normal_i"])
    for n, H, b, s in zip(nrm, z["normal_H"], z["normal_b"], z["normal_s"]):
        assert np.allclose(n.H_np(), H, rtol=1e-10, atol=1e-10 * max(1.0, np.abs(H).max()))
        assert np.allclose(n.b_np(), b, rtol=1e-9, atol=1e-9 * max(1.0, np.abs(b).max()))
        assert np.allclose([n.cost, n.chi2], s, rtol=1e-12)
    kp, pt = o.correspondences(p, z["xs"][0], 2)
    assert np.array_equal(kp, z["corr_f2_kp"]) and np.array_equal(pt, z["corr_f2_pt"])
    # last candidate is far off: every frame skipped -> DBL_MAX sentinels, NaN C (iba_global.cpp:330-338)
    assert cost[-1].frames_used == 0 and cost[-1].f1 == np.finfo(np.float64).max and cost[-1].f2 == np.finfo(np.float64).max and np.isnan(cost[-1].C)


def test_correspondences_vs_numpy_brute_force(ob, abi, synth, gold):
    """FindProjectCorrespondences (iba_global.cpp:55-96) restated independently in numpy."""
    z, prob = gold
    p = abi.reference_yaml_params()
    o = ob.Oracle(prob)
    x = z["xs"][1]
    R, t, _ = ob.sim3exp(x)
    fx, fy, cx, cy, W, H = prob.arrays["intrinsics"][:6]
    for f in range(prob.n_frames):
        P = prob.frame_points(f).astype(np.float64)
        pc = P @ R.T + t
        zc = pc[:, 2]
        with np.errstate(all="ignore"):
            u = (fx * pc[:, 0] + cx * zc) / zc
            v = (fx * pc[:, 1] + cy * zc) / zc
        ok = (zc > 0) & (u >= 0) & (u < W) & (v >= 0) & (v < H)
        ids = np.flatnonzero(ok)
        kps = prob.frame_keypoints(f).astype(np.float64)
        d2 = (kps[:, None, 0] - u[ids][None, :]) ** 2 + (kps[:, None, 1] - v[ids][None, :]) ** 2
        j = np.argmin(d2, 1)
        best = d2[np.arange(len(kps)), j]
        sel = best <= p.max_pixel_dist ** 2
        kp, pt = o.correspondences(p, x, f)
        assert np.array_equal(kp, np.flatnonzero(sel).astype(np.uint32))
        assert np.array_equal(pt, ids[j[sel]].astype(np.uint32))


def test_residual_jacobians_vs_central_differences(ob, abi, gold):
    """The dual-number Jacobians of IBA_PlaneFactor / Point2Plane / Point2Point (what Ceres' autodiff yields)."""
    z, prob = gold
    p = abi.reference_yaml_params()
    o = ob.Oracle(prob)
    x = z["xs"][1]
    assert o.build_problem(p, x) > 50
    r, J, bid, kind, fk = o.eval_residuals(x)
    assert set(np.unique(kind)) <= {0, 1, 2} and len(r) == len(J)
    Jn = np.zeros_like(J)
    for k in range(7):
        h = 1e-6 * max(1.0, abs(x[k]))
        xp, xm = x.copy(), x.copy()
        xp[k] += h
        xm[k] -= h
        Jn[:, k] = (o.eval_residuals(xp)[0] - o.eval_residuals(xm)[0]) / (2 * h)
    assert np.allclose(J, Jn, rtol=2e-5, atol=2e-5 * np.abs(J).max())
    # normal equations are consistent with the per-residual output: H = sum w J^T J, b = sum w J^T r
    no = o.eval_factors(p, x)[0]
    H = np.zeros((7, 7))
    b = np.zeros(7)
    cost = 0.0
    for blk in np.unique(bid):
        m = bid == blk
        s = float(r[m] @ r[m])
        a = p.robust_kernel_delta if kind[m][0] == 0 else p.robust_kernel_3ddelta
        rho0, w = ob.huber(a, s)
        H += w * J[m].T @ J[m]
        b += w * J[m].T @ r[m]
        cost += 0.5 * rho0
    assert np.allclose(no.H_np(), H, rtol=1e-10, atol=1e-10 * np.abs(H).max()) and np.allclose(no.b_np(), b, rtol=1e-9, atol=1e-9 * np.abs(b).max())
    assert np.isclose(no.cost, cost, rtol=1e-12)


def test_cost_variants_and_bbo(ob, abi, gold):
    z, prob = gold
    o = ob.Oracle(prob)
    x = z["xs"][0]
    p = abi.reference_yaml_params()
    base = o.eval_cost(p, x)[0]
    p0 = abi.reference_yaml_params()
    p0.err_weight[1] = 0.0      # 3d-3d disabled: counters bumped once per processed frame (iba_global.cpp:214-220)
    c = o.eval_cost(p0, x)[0]
    assert c.cnt_3d_3d == c.frames_used == c.valid_cnt_3d_3d and c.f2 == 0.0 and c.f1 == base.f1
    pn = abi.reference_yaml_params()
    pn.use_plane = 0
    c = o.eval_cost(pn, x)[0]
    assert c.valid_pl_3d_3d == 0 and c.valid_pt_3d_3d == c.valid_cnt_3d_3d and c.f2 >= base.f2
    bb = o.eval_bbo(p, x, 0.094, 0.95)[0]
    assert np.isclose(bb.f, base.f1 + base.f2) and np.isclose(bb.c1, base.C - 0.094) and np.isclose(bb.c2, -base.C - 0.094)
    assert np.isclose(bb.c3, 0.95 - base.valid_cnt_3d_2d / (base.cnt_3d_2d + 1))
    # OpenMP over frames (iba_func.cpp:203) gives the same counters and the same sums up to rounding
    c8 = o.eval_cost(p, x, nthreads=4)[0]
    assert (c8.cnt_3d_2d, c8.cnt_3d_3d, c8.n_corr) == (base.cnt_3d_2d, base.cnt_3d_3d, base.n_corr) and np.isclose(c8.f1, base.f1, rtol=1e-12)


def test_frame_ranges_sum_to_whole(ob, abi, gold):
    z, prob = gold
    p = abi.reference_yaml_params()
    o = ob.Oracle(prob)
    x = z["xs"][0]
    whole = o.eval_cost_raw(p, x, 0, prob.n_frames)
    parts = o.eval_cost_raw(p, x, 0, 2) + o.eval_cost_raw(p, x, 2, prob.n_frames)
    assert np.allclose(whole, parts, rtol=1e-13)
    assert whole[3] == whole[10] - 1    # HE term exists for every processed frame but the last (iba_global.cpp:264)


def test_plane_edge_twin_and_block_conditioning(synth, abi, ob):
    """IBAPlaneEdge (the g2o twin of IBA_PlaneFactor, IBACalib.hpp:103-140) = the block's rows zero-padded to 20; the conditioning
    measure of a block is 1 for 3d-3d blocks, in (0, 1] for plane factors, and goes to 0 when the viewing ray is rotated into the plane."""
    prob, meta = synth.make_scene(n_frames=3, pts_per_frame=3000, seed=3)
    p = abi.reference_yaml_params()
    o = ob.Oracle(prob)
    x = meta["x_gt"]
    o.build_problem(p, x)
    r, J, bid, kind, _ = o.eval_residuals(x)
    cond = o.block_conditioning(x)
    starts = np.concatenate([[0], np.where(np.diff(bid) != 0)[0] + 1])
    assert len(cond) == len(starts)
    seen = 0
    for i, lo in enumerate(starts):
        rows = np.where(bid == bid[lo])[0]
        if kind[lo] == 0 and len(rows) <= 20:
            e, Je = o.plane_edge20(int(bid[lo]), x)
            assert np.array_equal(e[:len(rows)], r[rows]) and np.all(e[len(rows):] == 0) and np.array_equal(Je[:len(rows)], J[rows]) and np.all(Je[len(rows):] == 0)
            assert 0 < cond[i] <= 1
            seen += 1
        elif kind[lo] != 0:
            assert cond[i] == 1.0 and o.plane_edge20(int(bid[lo]), x) is None
    assert seen > 20


def test_test_edge_against_torch_autograd_and_the_cost_loop(ob, abi, synth, gold):
    """IBATestEdge (IBACalib.hpp:14-71, functor :40-58) as factor kind 3 (iba_params.factor_3d2d_kind = 1). The reference declares the
    edge and constructs it nowhere, so it is pinned three ways: (1) its rows and Jacobians against an independent torch.float64
    autograd evaluation of p1 = R_i (R_cl p0 + t_cl) + s t_i through scipy-checked Sim3Exp formulas; (2) central differences;
    (3) its EDGE SET against BAError's own 3d-2d loop (iba_global.cpp:291-328): exactly cnt_3d_2d of them lie inside the image, and
    the mean of |r| over those below corr_3d_2d_threshold is the cost tuple's f1."""
    torch = pytest.importorskip("torch")
    z, prob = gold
    p = abi.reference_yaml_params()
    p.factor_3d2d_kind = 1
    o = ob.Oracle(prob)
    x = z["xs"][1]
    n_blocks = o.build_problem(p, x)
    r, J, bid, kind, fk = o.eval_residuals(x)
    assert set(np.unique(kind)) == {1, 2, 3} and n_blocks == len(np.unique(bid))
    m3 = kind == 3
    assert m3.sum() % 2 == 0 and np.all(np.bincount(bid[m3])[np.unique(bid[m3])] == 2)    # 2-row blocks, one per (correspondence, covisible keyframe)
    # (3) the edge set is the 3d-2d loop's: every in-image edge is one of cnt_3d_2d, the gated mean of the distances is f1
    c = o.eval_cost(p, x[None])[0]
    a = prob.arrays
    W, H = a["intrinsics"][4], a["intrinsics"][5]
    ru, rv = r[m3][0::2], r[m3][1::2]
    # observation = residual + matched keypoint; the matched keypoint of an edge is not returned: recover it from the problem arrays
    kp_off, co, mo = a["kp_offset"].astype(np.int64), a["covis_offset"].astype(np.int64), a["match_offset"].astype(np.int64)
    uv1 = []
    edge_fk = fk[m3][0::2]
    seen = {}
    for f, k in edge_fk:
        i = seen.get((f, k), 0)
        seen[(f, k)] = i + 1
        hits = []
        for gs in range(co[f], co[f + 1]):
            mm = np.flatnonzero(a["match_kp_ref"][mo[gs]:mo[gs + 1]] == k)
            if len(mm):
                kc = a["match_kp_covis"][mo[gs] + mm[0]]
                hits.append(a["kp_uv"].reshape(-1, 2)[kp_off[a["covis_frame"][gs]] + kc])
        uv1.append(hits[i])
    uv1 = np.array(uv1, np.float64)
    ou, ov = ru + uv1[:, 0], rv + uv1[:, 1]
    inside = (ou >= 0) & (ou < W) & (ov >= 0) & (ov < H)
    dist = np.sqrt(ru * ru + rv * rv)
    assert inside.sum() == c.cnt_3d_2d and (inside & (dist < p.corr_3d_2d_threshold)).sum() == c.valid_cnt_3d_2d
    assert np.isclose(dist[inside & (dist < p.corr_3d_2d_threshold)].mean(), c.f1, rtol=1e-12)
    # with err_weight[1] = 0 no 3d-3d block is built: the point-to-pixel-only problem (BASELINE configs[0])
    p0 = abi.reference_yaml_params()
    p0.factor_3d2d_kind = 1
    p0.err_weight[1] = 0.0
    o.build_problem(p0, x)
    r0, J0, _, kind0, _ = o.eval_residuals(x)
    assert set(np.unique(kind0)) == {3} and np.array_equal(r0, r[m3]) and np.array_equal(J0, J[m3])
    no = o.eval_factors(p0, x)[0]
    assert no.n_factor_3d2d == m3.sum() // 2 and no.n_factor_p2pl == 0 and no.n_factor_p2pt == 0 and no.n_residuals == m3.sum()
    # (1) torch autograd on 40 edges
    def sim3(xt):
        w, u, s = xt[:3], xt[3:6], xt[6]
        th = torch.sqrt((w * w).sum())
        Om = torch.zeros(3, 3, dtype=torch.float64)
        Om[0, 1], Om[0, 2], Om[1, 0], Om[1, 2], Om[2, 0], Om[2, 1] = -w[2], w[1], w[2], -w[0], -w[1], w[0]
        I = torch.eye(3, dtype=torch.float64)
        R = I + torch.sin(th) / th * Om + (1 - torch.cos(th)) / th ** 2 * Om @ Om
        V = I + (1 - torch.cos(th)) / th ** 2 * Om + (th - torch.sin(th)) / th ** 3 * Om @ Om
        return R, V @ u, s
    fx, fy, cx, cy = (float(v) for v in a["intrinsics"][:4])
    rows = np.flatnonzero(m3)[0::2][:40]
    o.build_problem(p, x)
    pt_off = a["pt_offset"].astype(np.int64)
    for e, row in enumerate(rows):
        f, k = fk[row]
        kp_, pt_ = o.correspondences(p, x, int(f))
        p0v = torch.tensor(prob.frame_points(int(f))[pt_[list(kp_).index(k)]].astype(np.float64))
        # which covisible keyframe: the i-th matching slot of this keypoint, in slot order
        i = int((edge_fk[:list(np.flatnonzero(m3)[0::2]).index(row)] == [f, k]).all(1).sum())
        slots = [gs for gs in range(co[f], co[f + 1]) if k in a["match_kp_ref"][mo[gs]:mo[gs + 1]]]
        gs = slots[i]
        rel = torch.tensor(a["covis_relpose"].reshape(-1, 12)[gs].astype(np.float64)).reshape(3, 4)
        xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
        def fun(xt):
            R, t, s = sim3(xt)
            p1 = rel[:, :3] @ (R @ p0v + t) + s * rel[:, 3]
            return torch.stack([fx * p1[0] / p1[2] + cx, fy * p1[1] / p1[2] + cy]) - torch.tensor(uv1[e])
        val = fun(xt).detach().numpy()
        Jt = torch.autograd.functional.jacobian(fun, xt).numpy()
        assert np.allclose(val, r[row:row + 2], rtol=1e-11, atol=1e-9) and np.allclose(Jt, J[row:row + 2], rtol=1e-9, atol=1e-9 * np.abs(Jt).max())
    # (2) central differences over every row
    Jn = np.zeros_like(J)
    for kk in range(7):
        hh = 1e-6 * max(1.0, abs(x[kk]))
        xp, xm = x.copy(), x.copy()
        xp[kk] += hh
        xm[kk] -= hh
        Jn[:, kk] = (o.eval_residuals(xp)[0] - o.eval_residuals(xm)[0]) / (2 * hh)
    assert np.allclose(J[m3], Jn[m3], rtol=2e-5, atol=2e-5 * np.abs(J[m3]).max())
