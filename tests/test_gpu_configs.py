"""The configurations BASELINE.json names, at their own shapes (synthetic scenes of SURVEY.md 8(d)):
  C1  50 keyframes x 4 k points (200 k), point-to-pixel only            -> GPU vs oracle, exact counters
  C2  200 keyframes x 10 k points (2 M), combined cost + normal equations -> GPU vs oracle on one candidate, and the
      size-independent property the multi-GPU path rests on: partial blocks of disjoint frame ranges add up to the whole
  C3  3 concatenated trajectories (600 keyframes, 6 M points) in 4 frame shards -> same additivity, all-reduce emulated
      by a host sum (the collective itself is covered by tests/test_distributed_gloo.py)
  C5  scale-free Jacobian path: normal equations vs oracle on the big scene (the LM end-to-end check is
      tests/test_gpu_calibrate.py)."""
import numpy as np
import pytest

import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import parity_gate  # noqa: E402

pytestmark = pytest.mark.gpu


def _partials(pkg, h, xs, kind):
    """This handle's partial block for the candidates xs (what a rank contributes to the all-reduce)."""
    assert kind == "full"
    h.eval_full(xs)
    return h.debug_last_partials(len(xs))


def _cmp_cost(g, o, rel=1e-10):
    for k in ("valid_cnt_3d_2d", "cnt_3d_2d", "cnt_3d_3d", "valid_cnt_3d_3d", "valid_pl_3d_3d", "valid_pt_3d_3d", "frames_used", "n_corr"):
        assert getattr(g, k) == getattr(o, k), (k, getattr(g, k), getattr(o, k))
    for k in ("f1", "f2"):
        assert abs(getattr(g, k) - getattr(o, k)) <= rel * max(abs(getattr(o, k)), 1e-300), k
    assert abs(g.C - o.C) <= 1e-10 * abs(o.C) + 1e-15


def test_c1_point_to_pixel_only(pkg, synth, abi, ob):
    prob, meta = synth.make_scene(n_frames=50, pts_per_frame=4000, seed=2)
    p = abi.reference_yaml_params()
    p.err_weight[1] = 0.0
    h = pkg.IbaHandle(prob, p)
    xs = np.vstack([meta["x_gt"][None], synth.perturb(meta["x_gt"], np.random.default_rng(2), n=3)])
    g = h.eval_cost(xs)
    o = ob.Oracle(prob).eval_cost(p, xs)
    for a, b in zip(g, o):
        _cmp_cost(a, b)
        assert a.cnt_3d_3d == a.frames_used and a.valid_cnt_3d_3d == a.frames_used   # iba_global.cpp:214-220
    h.close()


@pytest.fixture(scope="module")
def c2(synth):
    return synth.make_scene(n_frames=200, pts_per_frame=10000, seed=0)


def test_c2_combined_cost_and_normal_vs_oracle(pkg, synth, abi, ob, c2):
    prob, meta = c2
    p = abi.reference_yaml_params()
    h = pkg.IbaHandle(prob, p)
    x = synth.perturb(meta["x_gt"], np.random.default_rng(3), n=1)
    cost, nrm = h.eval_full(x)
    orc = ob.Oracle(prob)
    _cmp_cost(cost[0], orc.eval_cost(p, x)[0])
    on = orc.eval_normal(p, x)[0]
    # H, b, cost, chi^2 against the long-double evaluation (tests/parity_gate.py): within 1e-10 of the exact value per entry, or no further from it than
    # 1.5 x the double oracle's own error (round 6; rounds 1-5 held C2's H, b to 1e-9 of the double oracle)
    parity_gate.normal_vs_truth(nrm[0], on, orc.eval_normal_truth(p, x)[0])
    h.close()


def test_c2_frame_shards_add_up(pkg, synth, abi, c2):
    prob, meta = c2
    p = abi.reference_yaml_params()
    xs = synth.perturb(meta["x_gt"], np.random.default_rng(4), n=5)
    whole = pkg.IbaHandle(prob, p)
    full = _partials(pkg, whole, xs, "full")
    whole.close()
    acc = np.zeros_like(full)
    for r in range(8):   # the 8-GPU partition of one node
        a, b = pkg.shard_frames(prob.n_frames, 8, r, np.diff(prob.arrays["pt_offset"].astype(np.int64)))
        hs = pkg.IbaHandle(prob, p, frame_begin=a, frame_end=b)
        acc += _partials(pkg, hs, xs, "full")
        hs.close()
    cw, cs = pkg.finalize_cost(p, full), pkg.finalize_cost(p, acc)
    nw, ns = pkg.finalize_normal(p, full), pkg.finalize_normal(p, acc)
    for a, b in zip(cw, cs):
        _cmp_cost(b, a, rel=1e-12)
    for a, b in zip(nw, ns):
        assert a.counts() == b.counts()
        assert np.allclose(a.H_np(), b.H_np(), rtol=1e-12, atol=1e-12 * np.abs(a.H_np()).max())
        assert np.allclose(a.b_np(), b.b_np(), rtol=1e-12, atol=1e-12 * np.abs(a.b_np()).max())


def test_c3_three_trajectories_four_shards(pkg, synth, abi, c2):
    prob, meta = synth.tile_scene(*c2, 3)
    assert prob.n_frames == 600 and prob.n_points == 6_000_000
    p = abi.reference_yaml_params()
    xs = synth.perturb(meta["x_gt"], np.random.default_rng(6), n=2)
    acc = None
    for r in range(4):
        a, b = pkg.shard_frames(prob.n_frames, 4, r)
        hs = pkg.IbaHandle(prob, p, frame_begin=a, frame_end=b)
        part = _partials(pkg, hs, xs, "full")
        acc = part if acc is None else acc + part
        hs.close()
    c3 = pkg.finalize_cost(p, acc)
    # three copies of the same trajectory: counters are 3x those of one copy, means are unchanged (a checksum of checksums)
    one = pkg.IbaHandle(c2[0], p)
    c1 = one.eval_cost(xs)
    one.close()
    for a, b in zip(c3, c1):
        assert a.n_corr == 3 * b.n_corr and a.cnt_3d_3d == 3 * b.cnt_3d_3d and a.cnt_3d_2d == 3 * b.cnt_3d_2d and a.frames_used == 3 * b.frames_used
        assert abs(a.f1 - b.f1) <= 1e-12 * b.f1 and abs(a.f2 - b.f2) <= 1e-12 * b.f2


def test_c4_full_size_eight_shards(pkg, synth, abi, c2):
    """C4 whole: 2000 keyframes x 10 k points (20 M points) on ONE device, then as the 8 frame ranges of an 8-GPU node: the
    partial blocks of the ranges add up to the unsharded block — counters bit for bit, sums to 1e-13 — and, the scene being ten
    copies of C2's trajectory, every counter is ten times C2's (a checksum of checksums). Four candidates: the batch shares its
    pair search, the neighbour lists are anchored."""
    prob, meta = synth.tile_scene(*c2, 10)
    assert prob.n_frames == 2000 and prob.n_points == 20_000_000
    p = abi.reference_yaml_params()
    xs = synth.perturb(meta["x_gt"], np.random.default_rng(12), n=4)
    whole = pkg.IbaHandle(prob, p)
    full = _partials(pkg, whole, xs, "full")
    assert whole.last_path == 1
    whole.close()
    acc = np.zeros_like(full)
    for r in range(8):
        a, b = pkg.shard_frames(prob.n_frames, 8, r, np.diff(prob.arrays["pt_offset"].astype(np.int64)))
        hs = pkg.IbaHandle(prob, p, frame_begin=a, frame_end=b)
        acc += _partials(pkg, hs, xs, "full")
        hs.close()
    cw, cs = pkg.finalize_cost(p, full), pkg.finalize_cost(p, acc)
    nw, ns = pkg.finalize_normal(p, full), pkg.finalize_normal(p, acc)
    for a, b in zip(cw, cs):
        _cmp_cost(b, a, rel=1e-13)
    for a, b in zip(nw, ns):
        assert a.counts() == b.counts()
        assert np.allclose(a.H_np(), b.H_np(), rtol=1e-13, atol=1e-13 * np.abs(a.H_np()).max())
        assert np.allclose(a.b_np(), b.b_np(), rtol=1e-13, atol=1e-13 * np.abs(a.b_np()).max())
        assert abs(a.cost - b.cost) <= 1e-13 * abs(a.cost)
    one = pkg.IbaHandle(c2[0], p)
    c1 = one.eval_cost(xs)
    one.close()
    for a, b in zip(cw, c1):
        assert a.n_corr == 10 * b.n_corr and a.cnt_3d_3d == 10 * b.cnt_3d_3d and a.cnt_3d_2d == 10 * b.cnt_3d_2d and a.frames_used == 10 * b.frames_used
        assert abs(a.f1 - b.f1) <= 1e-12 * b.f1 and abs(a.f2 - b.f2) <= 1e-12 * b.f2


def test_kitti_sized_scans(pkg, synth, abi, ob):
    """Raw KITTI scans are ~120 k points (~60 k with PointCloudOnlyPositiveX): deeper kd-trees (D = 12), candidate queues and
    work lists several times longer than at the bench shape. Cost tuple and normal equations vs the oracle."""
    prob, meta = synth.make_scene(n_frames=3, pts_per_frame=60000, seed=8)   # depth-capped tree: 30 points per leaf
    p = abi.reference_yaml_params()
    h = pkg.IbaHandle(prob, p)
    xs = np.vstack([meta["x_gt"][None], synth.perturb(meta["x_gt"], np.random.default_rng(8), n=4)])   # five candidates: the batch shares its pair search
    cost, nrm = h.eval_full(xs)
    assert h.last_path == 1
    orc = ob.Oracle(prob)
    oc, on = orc.eval_cost(p, xs), orc.eval_normal(p, xs)
    for a, b in zip(cost, oc):
        _cmp_cost(a, b)
    for a, b in zip(nrm, on):
        assert a.counts() == b.counts()
        assert np.allclose(a.H_np(), b.H_np(), rtol=1e-9, atol=1e-9 * np.abs(b.H_np()).max())
    for f in range(3):
        gk, gp = h.correspondences(xs[1], f)
        ok, op = orc.correspondences(p, xs[1], f)
        assert np.array_equal(gk, ok) and np.array_equal(gp, op)
    h.close()


def test_raw_kitti_sweep_size(pkg, synth, abi, ob):
    """120 k points per scan: the kd-tree hits its depth cap (59 points per leaf), the candidate queue and the culling list
    are at their largest."""
    prob, meta = synth.make_scene(n_frames=2, pts_per_frame=120000, seed=9)
    p = abi.reference_yaml_params()
    h = pkg.IbaHandle(prob, p)
    xs = np.vstack([meta["x_gt"][None], synth.perturb(meta["x_gt"], np.random.default_rng(9), n=1)])
    cost, nrm = h.eval_full(xs)
    orc = ob.Oracle(prob)
    for a, b in zip(cost, orc.eval_cost(p, xs)):
        _cmp_cost(a, b)
    for a, b in zip(nrm, orc.eval_normal(p, xs)):
        assert a.counts() == b.counts()
        assert np.allclose(a.H_np(), b.H_np(), rtol=1e-9, atol=1e-9 * np.abs(b.H_np()).max())
    h.set_params(abi.reference_yaml_params(plane_cache=0))     # refit path on the same scans
    c0 = h.eval_cost(xs[:1])[0]
    assert c0.n_corr == cost[0].n_corr and c0.valid_pl_3d_3d == cost[0].valid_pl_3d_3d and abs(c0.f2 - cost[0].f2) <= 1e-12 * cost[0].f2
    h.close()
