"""GPU parity that does not lean on a live oracle, and the BASELINE.json configurations at their own shapes.

* the committed golden fixture tests/golden/path_small_scene.npz (scene + the oracle's outputs, written by
  tests/golden/make_golden.py) is replayed through the C-ABI: counters and the correspondence set bit-exact, f1 / f2 1e-10,
  every entry of H, b above 1e-6 of the largest within 1e-10 RELATIVE TO ITSELF, cost and chi2 1e-10;
* the same per-entry bar against the live oracle on the 12-keyframe scene and on one candidate of the C2 scene
  (BASELINE.md parity gate);
* C3 (600 keyframes, 6 M points) against the oracle on one candidate; C4's one-rank share (250 keyframes x 10 k points of
  the 2000-keyframe / 8-GPU configuration); C5 (300 keyframes x 10 k points, scale free): iba_calibrate_lm against the same
  LM driven by the CPU oracle, final SE(3) within the north_star tolerance;
* the ill-conditioned residual block (viewing ray almost in its plane: residual 2e4 px, Jacobian entries 3e9) that limits
  per-entry agreement of the SUMS to ~1e-7 on some candidates: its rows agree with the oracle to 1e-9 of the row's scale, and
  every entry of H, b agrees to 1e-9 of the largest;
* a batch whose interleaved work list has stretches of 64 entries none of which wants a search (regression: the search
  kernel's waves must keep claiming)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import lm_ref  # noqa: E402
import parity_gate  # noqa: E402

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "path_small_scene.npz")
INT = ("valid_cnt_3d_2d", "cnt_3d_2d", "cnt_3d_3d", "valid_cnt_3d_3d", "valid_pl_3d_3d", "valid_pt_3d_3d", "frames_used", "n_corr")
NCPU = os.cpu_count() or 8


def per_entry(g, o, rel=1e-10, floor=1e-6):
    """every entry of o above floor * max|o| is matched to rel of ITSELF; the rest to rel * floor * max|o|"""
    g, o = np.asarray(g, np.float64), np.asarray(o, np.float64)
    big = np.max(np.abs(o))
    if big == 0:
        assert not np.any(g)
        return
    m = np.abs(o) > floor * big
    assert np.all(np.abs(g - o)[m] <= rel * np.abs(o)[m]), float(np.max(np.abs(g - o)[m] / np.abs(o)[m]))
    assert np.all(np.abs(g - o)[~m] <= rel * floor * big * 10)


def cmp_cost(g, o, rel=1e-10):
    for k in INT:
        assert getattr(g, k) == getattr(o, k), (k, getattr(g, k), getattr(o, k))
    for k in ("f1", "f2"):
        assert abs(getattr(g, k) - getattr(o, k)) <= rel * max(abs(getattr(o, k)), 1e-300), k
    assert (np.isnan(g.C) and np.isnan(o.C)) or abs(g.C - o.C) <= 1e-10 * abs(o.C) + 1e-15   # device acos / tan vs glibc: measured 5e-12 over 1500 random scenes (tools/soak_parity.py, r03)


def cmp_normal(g, o, t=None):
    """t: the long-double evaluation (Oracle.eval_normal_truth). With it the device is held to the truth (tests/parity_gate.py: within 1e-10 of the
    exact value, or no further from it than 1.5 x the double oracle's own error); without it, to 1e-10 of the double oracle's entries."""
    assert g.counts() == o.counts(), (g.counts(), o.counts())
    if t is not None:
        parity_gate.normal_vs_truth(g, o, t)
        per_entry(g.H_np(), o.H_np(), rel=2e-9); per_entry(g.b_np(), o.b_np(), rel=2e-9)   # (and the two double evaluations stay near each other)
        return
    per_entry(g.H_np(), o.H_np())
    per_entry(g.b_np(), o.b_np())
    assert abs(g.cost - o.cost) <= 1e-10 * abs(o.cost) and abs(g.chi2 - o.chi2) <= 1e-10 * abs(o.chi2)


def test_golden_fixture_replayed_on_the_gpu(pkg, abi):
    z = np.load(GOLD)
    prob = abi.Problem(**{k[len("scene_"):]: z[k] for k in z.files if k.startswith("scene_")})
    p = abi.reference_yaml_params()
    h = pkg.IbaHandle(prob, p)
    xs = z["xs"]
    cost, nrm = h.eval_full(xs)
    for b in range(len(xs)):
        assert [getattr(cost[b], k) for k in INT] == list(z["cost_i"][b])
        for j, k in enumerate(("f1", "f2")):
            assert abs(getattr(cost[b], k) - z["cost_f"][b, j]) <= 1e-10 * abs(z["cost_f"][b, j])
        assert (np.isnan(cost[b].C) and np.isnan(z["cost_f"][b, 2])) or abs(cost[b].C - z["cost_f"][b, 2]) <= 1e-10 * abs(z["cost_f"][b, 2]) + 1e-15
        c = nrm[b].counts()
        assert [c[k] for k in ("n_factor_3d2d", "n_factor_p2pl", "n_factor_p2pt", "n_residuals", "frames_used", "n_corr")] == list(z["normal_i"][b])
        per_entry(nrm[b].H_np(), z["normal_H"][b])
        per_entry(nrm[b].b_np(), z["normal_b"][b])
        assert abs(nrm[b].cost - z["normal_s"][b, 0]) <= 1e-10 * z["normal_s"][b, 0] and abs(nrm[b].chi2 - z["normal_s"][b, 1]) <= 1e-10 * z["normal_s"][b, 1]
    kp, pt = h.correspondences(xs[0], 2)
    assert np.array_equal(kp, z["corr_f2_kp"]) and np.array_equal(pt, z["corr_f2_pt"])
    # the separate entry points see the same numbers
    for a, b in zip(h.eval_cost(xs), cost):   # the cost-only chain lists other keypoints: same terms, another summation order
        da, db = a.as_dict(), b.as_dict()
        assert all(da[k] == db[k] or (da[k] != da[k] and db[k] != db[k]) or (k in ("f1", "f2") and abs(da[k] - db[k]) <= 1e-13 * abs(db[k])) for k in da), (da, db)
    h.close()


def test_per_entry_gate_small_scene(pkg, synth, abi, ob, scene_small):
    prob, meta = scene_small
    p = abi.reference_yaml_params()
    h = pkg.IbaHandle(prob, p)
    o = ob.Oracle(prob)
    rng = np.random.default_rng(21)
    xs = np.vstack([meta["x_gt"][None], synth.perturb(meta["x_gt"], rng, n=4), synth.perturb(meta["x_gt"], rng, rot=0.01, trans=0.05, scale_rel=0.02, n=2)])
    cost, nrm = h.eval_full(xs)
    for a, b in zip(cost, o.eval_cost(p, xs, nthreads=min(NCPU, 12))):
        cmp_cost(a, b)
    for a, b in zip(nrm, o.eval_normal(p, xs, nthreads=min(NCPU, 12))):
        cmp_normal(a, b)
    h.build_problem(xs[1])
    o.build_problem(p, xs[1])
    for a, b in zip(h.eval_factors(xs[:4]), o.eval_factors(p, xs[:4])):   # frozen association, other candidates
        cmp_normal(a, b)
    h.close()


@pytest.fixture(scope="module")
def c2(synth):
    return synth.make_scene(n_frames=200, pts_per_frame=10000, seed=0)


def test_c2_per_entry_gate(pkg, synth, abi, ob, c2):
    prob, meta = c2
    p = abi.reference_yaml_params()
    h = pkg.IbaHandle(prob, p)
    o = ob.Oracle(prob)
    for seed in (31, 33):
        x = synth.perturb(meta["x_gt"], np.random.default_rng(seed), n=1)
        cost, nrm = h.eval_full(x)
        cmp_cost(cost[0], o.eval_cost(p, x, nthreads=min(NCPU, 64))[0])
        cmp_normal(nrm[0], o.eval_normal(p, x, nthreads=min(NCPU, 64))[0], o.eval_normal_truth(p, x)[0])
    h.close()


def test_c2_ill_conditioned_block(pkg, synth, abi, ob, c2):
    """Candidate 32 of the C2 scene re-associates a plane factor whose viewing ray is almost parallel to its plane
    (Z0 = num / den, den a difference of nearly equal terms): |r| = 2e4 px, |J| = 3e9. The analytic chain rule and the oracle's
    duals are two double evaluations of a quotient with a condition number of ~1e6: they agree to 3e-10 of the row's scale, and
    this one block moves entries of b by 8e-8 of themselves (measured, tools/worst_block.py; the same with and without FMA).
    Bars here: every row to 1e-9 of its scale, every entry of H to 1e-9 and of b to 1e-8 of the largest entry (the block's
    J^T r term is 1e4 x 3e9 against |b| = 2e9), counters exact."""
    prob, meta = c2
    p = abi.reference_yaml_params()
    h = pkg.IbaHandle(prob, p)
    o = ob.Oracle(prob)
    x = synth.perturb(meta["x_gt"], np.random.default_rng(32), n=1)[0]
    g, r = h.eval_normal(x)[0], o.eval_normal(p, x, nthreads=min(NCPU, 64))[0]
    assert g.counts() == r.counts()
    assert np.max(np.abs(g.H_np() - r.H_np())) <= 1e-9 * np.max(np.abs(r.H_np())) and np.max(np.abs(g.b_np() - r.b_np())) <= 1e-8 * np.max(np.abs(r.b_np()))
    h.build_problem(x)
    o.build_problem(p, x)
    rg, Jg, _, kg = h.eval_residuals(x)
    ro, Jo, _, ko, _ = o.eval_residuals(x)
    assert np.array_equal(kg, ko) and np.abs(ro).max() > 1e4 and np.abs(Jo).max() > 1e9   # the block is there
    scale = np.maximum(np.maximum(np.abs(Jo).max(axis=1), np.abs(ro)), 1.0)
    assert np.all(np.abs(Jg - Jo).max(axis=1) <= 1e-9 * scale) and np.all(np.abs(rg - ro) <= 1e-9 * scale)
    h.close()


def test_c3_six_million_points_vs_oracle(pkg, synth, abi, ob, c2):
    prob, meta = synth.tile_scene(*c2, 3)
    assert prob.n_frames == 600 and prob.n_points == 6_000_000
    p = abi.reference_yaml_params()
    h = pkg.IbaHandle(prob, p)
    x = synth.perturb(meta["x_gt"], np.random.default_rng(33), n=1)
    cost, nrm = h.eval_full(x)
    o = ob.Oracle(prob)
    cmp_cost(cost[0], o.eval_cost(p, x, nthreads=min(NCPU, 64))[0])
    cmp_normal(nrm[0], o.eval_normal(p, x, nthreads=min(NCPU, 64))[0], o.eval_normal_truth(p, x)[0])
    h.close()


def test_c4_one_rank_share(pkg, synth, abi, ob):
    """configs[3]: 20 M points x 2000 keyframes on 8 GPUs = 250 keyframes x 10 k points per rank. One rank's share at its own
    shape, as a frame range of a larger descriptor (the covisible / hand-eye neighbours across the range boundary are
    resolved at creation), against the oracle restricted to the same frames."""
    base, meta = synth.make_scene(n_frames=125, pts_per_frame=10000, seed=12)
    prob, meta = synth.tile_scene(base, meta, 3)           # 375 keyframes; rank 0 of the 8-GPU partition owns 250 of 2000
    p = abi.reference_yaml_params()
    f0, f1 = 60, 310                                       # a range that starts and ends inside copies of the trajectory
    h = pkg.IbaHandle(prob, p, frame_begin=f0, frame_end=f1)
    assert h.n_points == 2_500_000
    xs = synth.perturb(meta["x_gt"], np.random.default_rng(33), n=2)
    h.eval_full(xs)
    part = h.debug_last_partials(len(xs))
    o = ob.Oracle(prob)
    for b in range(len(xs)):
        raw = o.eval_cost_raw(p, xs[b], f0, f1)           # the oracle's unnormalised sums over the same frame range
        stride = pkg.partial_stride()
        q = part[b]
        # slots: SUM_3D2D, SUM_3D3D, HE_SUM, HE_CNT, CNT_3D2D, VALID_3D2D, CNT_3D3D, VALID_3D3D, VALID_PL, VALID_PT, FRAMES, NCORR
        assert list(q[3:12].astype(np.int64)) == list(np.asarray(raw[3:12]).astype(np.int64)), (q[:12], raw)
        assert abs(q[0] - raw[0]) <= 1e-10 * abs(raw[0]) and abs(q[1] - raw[1]) <= 1e-10 * abs(raw[1]) and abs(q[2] - raw[2]) <= 1e-9 * abs(raw[2])
    h.close()


def test_c5_scale_free_lm_at_its_own_shape(pkg, synth, abi, ob):
    """configs[4]: IBACalib2 scale-aware variant (extrinsic + monocular scale), 300 keyframes x 10 k points: the device LM
    caller against the same LM on the CPU oracle from the same start; final SE(3) and scale within the north_star tolerance."""
    prob, meta = synth.make_scene(n_frames=300, pts_per_frame=10000, seed=13)
    p = abi.reference_yaml_params()
    h = pkg.IbaHandle(prob, p)
    o = ob.Oracle(prob)
    x0 = synth.perturb(meta["x_gt"], np.random.default_rng(34), rot=1e-3, trans=0.01, scale_rel=3e-3, n=1)[0]

    def ev(x):
        n = o.eval_factors(p, x)[0]
        return n.H_np(), n.b_np(), n.cost

    xc, sc = lm_ref.calibrate_lm(x0, lambda x: o.build_problem(p, x), ev, max_outer=4)
    xg, rg = h.calibrate_lm(x0, max_outer_iterations=4)
    rot, trans, scl = lm_ref.se3_error(xg, xc, synth.sim3_exp)
    assert rot < 1e-4 and trans < 1e-3 and scl < 1e-4, (rot, trans, scl)
    assert rg.outer_iterations == sc["outer"] and rg.evaluations == sc["evals"]
    assert abs(rg.final_cost - sc["final_cost"]) <= 1e-8 * sc["final_cost"]
    assert abs(xg[6] - xc[6]) <= 1e-6 * abs(xc[6])
    h.close()


def test_sparse_work_lists_keep_the_search_waves_claiming(pkg, synth, abi, ob):
    """err_weight[1] = 0 leaves only the association-path searches, for keypoints that own a MapPoint AND a covisible match AND
    a valid local neighbourhood: the interleaved work list of 8 candidates then has runs of 64 entries without a search."""
    prob, meta = synth.make_scene(n_frames=3, pts_per_frame=6000, n_keypoints=600, seed=5008)
    p = abi.reference_yaml_params()
    p.err_weight[1] = 0.0
    h = pkg.IbaHandle(prob, p)
    o = ob.Oracle(prob)
    xs = synth.perturb(meta["x_gt"], np.random.default_rng(5008), rot=1e-4, trans=5e-4, scale_rel=2e-4, n=8)
    cost, nrm = h.eval_full(xs)
    for a, b in zip(cost, o.eval_cost(p, xs)):
        cmp_cost(a, b)
    for a, b in zip(nrm, o.eval_normal(p, xs)):
        assert a.counts() == b.counts()
        per_entry(a.H_np(), b.H_np(), rel=1e-9)
    h.close()
