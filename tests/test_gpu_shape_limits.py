"""Shapes the reference allows and round 1 refused: more than 10 covisible keyframes per frame
(GetCovisiblesByWeightSafe is unbounded, iba_global.cpp:259), norm_max_pts / neigh_max_pts above 32, more than 64 candidates per call."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
INT = ("valid_cnt_3d_2d", "cnt_3d_2d", "cnt_3d_3d", "valid_cnt_3d_3d", "valid_pl_3d_3d", "valid_pt_3d_3d", "frames_used", "n_corr")


def _check(h, o, p, xs):
    cost, nrm = h.eval_full(xs)
    for a, b in zip(cost, o.eval_cost(p, xs, nthreads=8)):
        for k in INT:
            assert getattr(a, k) == getattr(b, k), k
        assert abs(a.f1 - b.f1) <= 1e-10 * abs(b.f1) and abs(a.f2 - b.f2) <= 1e-10 * abs(b.f2)
    for a, b in zip(nrm, o.eval_normal(p, xs, nthreads=8)):
        assert a.counts() == b.counts()
        # (ill-conditioned plane blocks bound the agreement of the sums: tests/test_gpu_golden_and_shapes.py::test_c2_ill_conditioned_block)
        assert np.max(np.abs(a.H_np() - b.H_np())) <= 1e-8 * np.abs(b.H_np()).max() and np.max(np.abs(a.b_np() - b.b_np())) <= 1e-8 * np.abs(b.b_np()).max()
    return cost, nrm


def test_fourteen_covisible_keyframes(pkg, synth, abi, ob):
    prob, meta = synth.make_scene(n_frames=18, pts_per_frame=3000, n_keypoints=800, seed=41, n_covis=14)
    assert int(np.diff(prob.arrays["covis_offset"].astype(np.int64)).max()) == 14
    p = abi.reference_yaml_params()
    h, o = pkg.IbaHandle(prob, p), ob.Oracle(prob)
    xs = np.vstack([meta["x_gt"][None], synth.perturb(meta["x_gt"], np.random.default_rng(41), n=3)])
    cost, nrm = _check(h, o, p, xs)
    assert cost[0].cnt_3d_2d > 3 * cost[0].n_corr // 2     # many covisible reprojections per correspondence
    h.build_problem(xs[1])
    o.build_problem(p, xs[1])
    rg, Jg, _, kg = h.eval_residuals(xs[2])
    ro, Jo, _, ko, _ = o.eval_residuals(xs[2])
    assert np.array_equal(kg, ko) and np.allclose(rg, ro, rtol=1e-9, atol=1e-9) and np.allclose(Jg, Jo, rtol=1e-8, atol=1e-8 * np.abs(Jo).max())
    h.close()
    hr = pkg.IbaHandle(prob, abi.reference_yaml_params(plane_cache=0))     # the per-evaluation refit path takes the same slot packing
    cr, nr = hr.eval_full(xs)
    for a, b in zip(cost, cr):   # bit for bit: same kernels, same summation order, the planes fitted by the same code
        assert all(x == y or (x != x and y != y) for x, y in zip(a.as_dict().values(), b.as_dict().values()))
    for a, b in zip(nrm, nr):
        assert a.counts() == b.counts() and np.array_equal(a.H_np(), b.H_np()) and a.cost == b.cost
    hr.close()


def test_neighbour_lists_longer_than_32(pkg, synth, abi, ob):
    prob, meta = synth.make_scene(n_frames=4, pts_per_frame=9000, n_keypoints=900, seed=42)
    p = abi.reference_yaml_params()
    p.norm_max_pts, p.neigh_max_pts, p.norm_radius, p.neigh_radius = 50, 48, 0.9, 0.8
    h, o = pkg.IbaHandle(prob, p), ob.Oracle(prob)
    xs = np.vstack([meta["x_gt"][None], synth.perturb(meta["x_gt"], np.random.default_rng(42), n=2)])
    _check(h, o, p, xs)
    h.set_params(abi.reference_yaml_params(plane_cache=0))    # refit mode with the long lists
    pr = abi.reference_yaml_params(plane_cache=0)
    pr.norm_max_pts, pr.neigh_max_pts, pr.norm_radius, pr.neigh_radius = 50, 48, 0.9, 0.8
    h.set_params(pr)
    _check(h, o, pr, xs[:2])
    h.close()


def test_more_than_sixty_four_candidates_per_call(pkg, synth, abi, scene_small):
    prob, meta = scene_small
    p = abi.reference_yaml_params()
    h = pkg.IbaHandle(prob, p)
    xs = synth.perturb(meta["x_gt"], np.random.default_rng(43), n=150)
    cost, nrm = h.eval_full(xs)
    assert len(cost) == 150
    for lo, hi in ((0, 64), (64, 128), (128, 150)):
        c2, n2 = h.eval_full(xs[lo:hi])
        for a, b in zip(cost[lo:hi], c2):
            assert a.as_dict() == b.as_dict()
        for a, b in zip(nrm[lo:hi], n2):
            assert np.array_equal(a.H_np(), b.H_np()) and a.counts() == b.counts()
    bb = h.eval_bbo(xs, 0.094, 0.95)
    assert len(bb) == 150 and bb[100].f == cost[100].f1 * p.err_weight[0] + cost[100].f2 * p.err_weight[1]
    c3 = h.eval_cost(xs)   # the cost-only chain lists other keypoints: same terms, another summation order
    for a, b in zip(c3, cost):
        assert all(getattr(a, k) == getattr(b, k) for k in INT) and abs(a.f1 - b.f1) <= 1e-13 * b.f1 and abs(a.f2 - b.f2) <= 1e-13 * b.f2
    # the same batch with every plane fitted inside its evaluation (three chunks through the fit kernels): bit for bit
    h.set_params(abi.reference_yaml_params(plane_cache=0))
    cr, nr = h.eval_full(xs)
    for a, b in zip(cost, cr):
        assert all(x == y or (x != x and y != y) for x, y in zip(a.as_dict().values(), b.as_dict().values()))
    for a, b in zip(nrm, nr):
        assert np.array_equal(a.H_np(), b.H_np()) and np.array_equal(a.b_np(), b.b_np()) and a.counts() == b.counts()
    h.close()


def test_twenty_eight_covisible_keyframes(pkg, synth, abi, ob):
    """GetCovisiblesByWeightSafe (iba_global.cpp:259) is unbounded; the flag word holds one match bit per covisible keyframe:
    30 of them since round 3 (22 before); a second word carries the slots 30..61 since round 4 (test_forty_covisible_keyframes)."""
    prob, meta = synth.make_scene(n_frames=32, pts_per_frame=2500, n_keypoints=700, seed=43, n_covis=28, new_mappoints=120, scan_kp=160)
    assert int(np.diff(prob.arrays["covis_offset"].astype(np.int64)).max()) == 28
    p = abi.reference_yaml_params()
    h, o = pkg.IbaHandle(prob, p), ob.Oracle(prob)
    xs = np.vstack([meta["x_gt"][None], synth.perturb(meta["x_gt"], np.random.default_rng(43), n=4)])
    cost, nrm = _check(h, o, p, xs)
    assert cost[0].cnt_3d_2d > 3 * cost[0].n_corr
    h.close()


def test_forty_covisible_keyframes(pkg, synth, abi, ob):
    """Beyond the 30 match bits of the flag word (r04): the covisible slots 30..61 of a keypoint sit in a second word that only frames
    with that many covisible keyframes read. 40 of them, against the oracle — cost path (3d-2d terms of every slot), normal equations
    (IBA_PlaneFactor blocks of up to 80 rows), residual rows, the refit mode; 63 are refused."""
    prob, meta = synth.make_scene(n_frames=44, pts_per_frame=2500, n_keypoints=700, seed=45, n_covis=40, new_mappoints=120, scan_kp=160)
    assert int(np.diff(prob.arrays["covis_offset"].astype(np.int64)).max()) == 40
    p = abi.reference_yaml_params()
    h, o = pkg.IbaHandle(prob, p), ob.Oracle(prob)
    xs = np.vstack([meta["x_gt"][None], synth.perturb(meta["x_gt"], np.random.default_rng(45), n=4)])
    cost, nrm = _check(h, o, p, xs)
    assert cost[0].cnt_3d_2d > 4 * cost[0].n_corr
    h.build_problem(xs[1])
    o.build_problem(p, xs[1])
    rg, Jg, bg, kg = h.eval_residuals(xs[2])
    ro, Jo, bo, ko, _ = o.eval_residuals(xs[2])
    assert np.array_equal(kg, ko) and np.array_equal(bg, bo) and np.allclose(rg, ro, rtol=1e-9, atol=1e-9) and np.allclose(Jg, Jo, rtol=1e-8, atol=1e-8 * np.abs(Jo).max())
    co, mo = prob.arrays["covis_offset"].astype(np.int64), prob.arrays["match_offset"].astype(np.int64)
    beyond = sum(int(mo[gs + 1] - mo[gs]) for f in range(44) for gs in range(co[f] + 30, co[f + 1]))
    assert beyond > 1000 and np.bincount(bo[ko == 0]).max() > 40   # matches in the slots the second word carries; blocks of more than 20 covisible matches
    h.close()
    hr = pkg.IbaHandle(prob, abi.reference_yaml_params(plane_cache=0))
    cr, nr = hr.eval_full(xs[:2])
    for a, b in zip(cost, cr):
        assert all(x == y or (x != x and y != y) for x, y in zip(a.as_dict().values(), b.as_dict().values()))
    hr.close()
    prob63, _ = synth.make_scene(n_frames=66, pts_per_frame=600, n_keypoints=200, seed=46, n_covis=63, new_mappoints=40, scan_kp=60)
    with pytest.raises(pkg.IbaError):
        pkg.IbaHandle(prob63, p)


@pytest.mark.parametrize("n_covis", [43, 62])
def test_more_than_forty_two_covisible_keyframes(pkg, synth, abi, ob, n_covis):
    """the association kernels staged the relative poses with ONE store per thread — 512 of the up to 744
    doubles — so the slots 42.. of a frame with 43..62 covisible keyframes were read from uninitialised LDS by the 3d-2d phase.
    43 (the first slot beyond one store per thread) and 62 (the limit), on BOTH association kernels (shared pair search: a tight
    batch; per-candidate: a batch as wide as the search box), against the oracle."""
    prob, meta = synth.make_scene(n_frames=n_covis + 4, pts_per_frame=2000, n_keypoints=600, seed=47, n_covis=n_covis, new_mappoints=100, scan_kp=140)
    assert int(np.diff(prob.arrays["covis_offset"].astype(np.int64)).max()) == n_covis
    p = abi.reference_yaml_params()
    h, o = pkg.IbaHandle(prob, p), ob.Oracle(prob)
    xs = np.vstack([meta["x_gt"][None], synth.perturb(meta["x_gt"], np.random.default_rng(47), n=3)])
    cost, _ = _check(h, o, p, xs)
    assert h.last_path == 1
    co, mo = prob.arrays["covis_offset"].astype(np.int64), prob.arrays["match_offset"].astype(np.int64)
    beyond = sum(int(mo[gs + 1] - mo[gs]) for f in range(prob.n_frames) for gs in range(co[f] + 42, co[f + 1]))
    assert beyond > 50 and cost[0].cnt_3d_2d > 4 * cost[0].n_corr   # matches in the slots beyond the 42nd
    xw = meta["x_gt"][None, :] + np.random.default_rng(48).uniform(-1, 1, (7, 7)) * np.array([0.03, 0.03, 0.03, 0.1, 0.1, 0.1, 0.2])   # seven candidates far from each other: more than four groups
    _check(h, o, p, xw)
    assert h.last_path == 0
    cc = h.eval_cost(xs)   # the cost-only chain
    for a, b in zip(cc, cost):
        assert all(getattr(a, k) == getattr(b, k) for k in INT) and abs(a.f1 - b.f1) <= 1e-13 * b.f1
    h.close()


def test_max_pixel_dist_changes_on_a_live_handle(pkg, synth, abi, ob):
    """max_pixel_dist used to be baked into the handle (the reject bitmap of the association is the keypoints dilated by it);
    iba_set_params now rebuilds the bitmap. Both association kernels, against the oracle, for a tighter and a wider gate."""
    prob, meta = synth.make_scene(n_frames=6, pts_per_frame=5000, n_keypoints=1200, seed=44)
    p = abi.reference_yaml_params()
    h, o = pkg.IbaHandle(prob, p), ob.Oracle(prob)
    xs = np.vstack([meta["x_gt"][None], synth.perturb(meta["x_gt"], np.random.default_rng(44), n=5)])
    base, _ = _check(h, o, p, xs)
    seen = {base[0].n_corr}
    for mpd in (0.7, 3.5, 1.5):
        q = abi.reference_yaml_params()
        q.max_pixel_dist = mpd
        h.set_params(q)
        for sel in (xs, xs[:2]):     # six candidates share their pair search, two search per candidate
            cost, _ = _check(h, o, q, sel)
        seen.add(cost[0].n_corr)
        for f in (0, 5):
            gk, gp = h.correspondences(xs[1], f)
            ok, op = o.correspondences(q, xs[1], f)
            assert np.array_equal(gk, ok) and np.array_equal(gp, op)
    assert len(seen) == 3          # the gate did change the correspondence sets
    assert cost[0].n_corr == base[0].n_corr
    h.close()
