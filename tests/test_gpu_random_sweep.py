"""GPU parity over a sweep of seeded random scenes and candidates (cost tuple, counters, normal equations): the fixed
scenes of the other test files pin known shapes, this one varies scan size (so the kd depth and leaf size), keypoint
count, perturbation size (tiny: every query ends in its first leaf; large: long far-side tails) and plane settings."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CASES = [
    # n_frames, pts, kp, seed, rot, trans, scale_rel
    (3, 700, 400, 101, 1e-4, 1e-3, 1e-4),      # shallow tree (D small), near the planted extrinsic
    (3, 2500, 900, 102, 2e-3, 2e-2, 5e-2),     # matches thin out, MapPoints off the surfaces by 5 % of their depth
    (2, 9000, 2000, 103, 2e-2, 5e-2, 3e-2),    # bench-like scan size
    (2, 30000, 1500, 104, 1e-2, 5e-2, 2e-2),   # deeper tree, leaves near the cap
    (4, 1500, 300, 105, 5e-2, 1.5e-1, 1e-1),   # the survey's largest perturbation: hardly any correspondence, frames skipped
]


def _cmp_cost(g, o):
    for k in ("valid_cnt_3d_2d", "cnt_3d_2d", "cnt_3d_3d", "valid_cnt_3d_3d", "valid_pl_3d_3d", "valid_pt_3d_3d", "frames_used", "n_corr"):
        assert getattr(g, k) == getattr(o, k), (k, getattr(g, k), getattr(o, k))
    for k in ("f1", "f2"):
        a, b = getattr(g, k), getattr(o, k)
        assert (np.isnan(a) and np.isnan(b)) or a == b or abs(a - b) <= 1e-10 * abs(b), (k, a, b)
    assert (np.isnan(g.C) and np.isnan(o.C)) or abs(g.C - o.C) <= 1e-10 * abs(o.C) + 1e-15


@pytest.mark.parametrize("case", CASES, ids=[f"F{c[0]}_P{c[1]}_K{c[2]}" for c in CASES])
def test_random_scene_sweep(pkg, synth, abi, ob, case):
    nf, pts, kp, seed, rot, trans, srel = case
    prob, meta = synth.make_scene(n_frames=nf, pts_per_frame=pts, n_keypoints=kp, seed=seed)
    p = abi.reference_yaml_params()
    h = pkg.IbaHandle(prob, p)
    o = ob.Oracle(prob)
    rng = np.random.default_rng(seed)
    xs = synth.perturb(meta["x_gt"], rng, rot=rot, trans=trans, scale_rel=srel, n=6)
    cf, nf_ = h.eval_full(xs)
    oc = o.eval_cost(p, xs)
    on = o.eval_normal(p, xs)
    for b in range(len(xs)):
        _cmp_cost(cf[b], oc[b])
        assert nf_[b].counts() == on[b].counts()
        Ho = on[b].H_np()
        if np.max(np.abs(Ho)) > 0:
            assert np.max(np.abs(nf_[b].H_np() - Ho)) <= 1e-9 * np.max(np.abs(Ho))
            assert np.max(np.abs(nf_[b].b_np() - on[b].b_np())) <= 1e-8 * max(np.max(np.abs(on[b].b_np())), 1e-30)
    # the sweep must not be vacuous: correspondences, 3d-3d terms and factors exist (everywhere near the planted extrinsic,
    # somewhere under the large perturbations)
    busy = [c.n_corr > 50 and c.cnt_3d_3d > 10 and n.n_factor_3d2d > 10 for c, n in zip(oc, on)]
    assert all(busy) if seed == 101 else (any(busy) or seed == 105), [(c.n_corr, c.cnt_3d_3d) for c in oc]
    # cost-only and association-only kernels are separate instantiations of the same phases
    cc = h.eval_cost(xs)
    for b in range(len(xs)):
        _cmp_cost(cc[b], oc[b])
    h.close()
