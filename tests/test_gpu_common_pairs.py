"""The 2d-3d association of a batch of nearby candidates shares ONE pair search per keyframe (iba_pairs_kernel +
iba_assoc2_kernel); a lone candidate or a wide batch searches per candidate (iba_assoc_kernel). Both must give the SAME BITS —
a candidate's result may not depend on what else is in the batch — and both must equal the oracle."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _handle(pkg, prob, p, mode, **more):
    """mode 0: every candidate searches for itself — 2d-3d pairs AND the 3-D nearest neighbours (no anchored lists); through the
    options struct of iba_create_ex (round 3 steered this with environment variables)"""
    return pkg.IbaHandle(prob, p, options=dict(common_pairs=mode, anchored_lists=0 if mode == 0 else 1, **more))


def _same_bits(pkg, h0, h1, xs, want_path):
    c0, n0 = h0.eval_full(xs)
    p0 = h0.debug_last_partials(min(len(xs), 64))
    assert h0.last_path == 0
    c1, n1 = h1.eval_full(xs)
    p1 = h1.debug_last_partials(min(len(xs), 64))
    assert h1.last_path in (want_path if isinstance(want_path, tuple) else (want_path,)), h1.last_path
    assert np.array_equal(p0, p1), np.argwhere(p0 != p1)[:5]
    for a, b in zip(c0, c1):
        assert a.as_dict() == b.as_dict()
    for a, b in zip(n0, n1):
        assert np.array_equal(a.H_np(), b.H_np()) and np.array_equal(a.b_np(), b.b_np()) and a.cost == b.cost and a.counts() == b.counts()
    a0, a1 = h0.eval_cost(xs), h1.eval_cost(xs)
    for a, b in zip(a0, a1):
        assert a.as_dict() == b.as_dict()


def test_shared_search_gives_the_same_bits(pkg, synth, abi, scene_small):
    prob, meta = scene_small
    p = abi.reference_yaml_params()
    h0, h1, h2 = _handle(pkg, prob, p, 0), _handle(pkg, prob, p, 1), _handle(pkg, prob, p, 2)
    rng = np.random.default_rng(11)
    tight = synth.perturb(meta["x_gt"], rng, n=64)
    _same_bits(pkg, h0, h1, tight, 1)
    _same_bits(pkg, h0, h1, tight[:4], 1)
    _same_bits(pkg, h0, h1, tight[:3], 1)
    _same_bits(pkg, h0, h1, tight[:1], 1)                       # a single candidate too (zero spread: the tightest lists)
    # ten times the spread: still shared by default? the nominal spread decides; forced sharing (mode 2) must agree anyway
    wide10 = synth.perturb(meta["x_gt"], rng, rot=5e-3, trans=5e-2, scale_rel=1e-2, n=16)
    _same_bits(pkg, h0, h2, wide10, 1)
    # as wide as the reference's whole search box (+-0.1 rad, +-0.3 m): windows of tens of pixels, hard points, list overflows
    box = meta["x_gt"][None, :] + rng.uniform(-1, 1, (12, 7)) * np.array([0.1, 0.1, 0.1, 0.3, 0.3, 0.3, 1.0])
    _same_bits(pkg, h0, h2, box, 1)
    _same_bits(pkg, h0, h1, box, (0, 2))                        # the default falls back for such a batch (or finds tight groups in it)
    # the frozen problem and the callers are untouched by the mode
    for h in (h0, h1):
        h.build_problem(tight[1])
    for a, b in zip(h0.eval_factors(tight[:5]), h1.eval_factors(tight[:5])):
        assert np.array_equal(a.H_np(), b.H_np()) and a.cost == b.cost
    # plane_cache = 0 runs on the same association
    q = abi.reference_yaml_params(plane_cache=0)
    h0.set_params(q)
    h1.set_params(q)
    _same_bits(pkg, h0, h1, tight[:8], 1)
    for h in (h0, h1, h2):
        h.close()


def test_shared_search_against_the_oracle(pkg, synth, abi, ob):
    """not only equal to the other kernel: equal to the CPU restatement (correspondence-derived counters exact, sums 1e-10)"""
    prob, meta = synth.make_scene(n_frames=8, pts_per_frame=6000, n_keypoints=1500, seed=21)
    p = abi.reference_yaml_params()
    h = _handle(pkg, prob, p, 2)
    o = ob.Oracle(prob)
    rng = np.random.default_rng(5)
    for xs in (synth.perturb(meta["x_gt"], rng, n=6), synth.perturb(meta["x_gt"], rng, rot=8e-3, trans=6e-2, scale_rel=2e-2, n=6)):
        g = h.eval_cost(xs)
        assert h.last_path == 1
        r = o.eval_cost(p, xs)
        for a, b in zip(g, r):
            assert (a.n_corr, a.cnt_3d_2d, a.valid_cnt_3d_2d, a.cnt_3d_3d, a.valid_cnt_3d_3d, a.frames_used) == (b.n_corr, b.cnt_3d_2d, b.valid_cnt_3d_2d, b.cnt_3d_3d, b.valid_cnt_3d_3d, b.frames_used)
            assert abs(a.f1 - b.f1) <= 1e-10 * abs(b.f1) and abs(a.f2 - b.f2) <= 1e-10 * abs(b.f2)
        gn, rn = h.eval_normal(xs), o.eval_normal(p, xs)
        for a, b in zip(gn, rn):
            assert a.counts() == b.counts()
            assert np.max(np.abs(a.H_np() - b.H_np())) <= 1e-9 * np.max(np.abs(b.H_np()))
    h.close()


def test_points_at_the_camera_plane_and_tiny_lists(pkg, synth, abi, ob):
    """Scan points whose depth cannot be bounded away from zero over the batch go through the hard list (exact per candidate);
    with the lists squeezed to a few entries every frame overflows and every candidate rescans. Both equal the per-candidate path."""
    prob, meta = synth.make_scene(n_frames=5, pts_per_frame=3000, n_keypoints=800, seed=31, new_mappoints=100, scan_kp=150)
    a = {k: v.copy() for k, v in prob.arrays.items()}
    pts = a["pts_xyz"].reshape(-1, 3)
    # 300 points per frame moved onto the camera plane, in front of the lens: LiDAR x ~ 0 is camera z ~ 0 for the KITTI-like extrinsic
    R, t, _ = synth.sim3_exp(meta["x_gt"])
    rng = np.random.default_rng(3)
    for f in range(5):
        qc = np.stack([rng.uniform(-0.02, 0.02, 300), rng.uniform(-0.01, 0.01, 300), rng.uniform(-0.02, 0.03, 300)], 1)   # camera frame, metres
        pts[3000 * f:3000 * f + 300] = ((qc - t) @ R).astype(np.float32)
    prob2 = abi.Problem(**a)
    p = abi.reference_yaml_params()
    xs = synth.perturb(meta["x_gt"], np.random.default_rng(9), n=8)
    h0, h2 = _handle(pkg, prob2, p, 0), _handle(pkg, prob2, p, 2)
    _same_bits(pkg, h0, h2, xs, 1)
    o = ob.Oracle(prob2)
    for a_, b_ in zip(h2.eval_cost(xs), o.eval_cost(p, xs)):
        assert (a_.n_corr, a_.cnt_3d_2d, a_.cnt_3d_3d) == (b_.n_corr, b_.cnt_3d_2d, b_.cnt_3d_3d)
    h2.close()
    h3 = _handle(pkg, prob2, p, 2, pair_list_capacity=64)
    _same_bits(pkg, h0, h3, xs, 1)
    h3.close()
    h0.close()


def test_anchored_neighbour_lists_follow_the_candidates(pkg, synth, abi, scene_small):
    """The 1-NN search is memoised around an anchor extrinsic (iba_anchor_kernel): lanes certify their pick from the anchor's
    neighbour lists or search the tree. Near the anchor, far from it (certificates fail), after the anchor has moved, with one
    candidate or many, after a parameter change: always the bits of the handle that searches the tree for every lane."""
    prob, meta = scene_small
    p = abi.reference_yaml_params()
    h0, h1 = _handle(pkg, prob, p, 0), _handle(pkg, prob, p, 1)
    rng = np.random.default_rng(3)
    x1 = meta["x_gt"]
    x2 = x1 + np.array([0.004, -0.003, 0.002, 0.05, -0.04, 0.03, 0.3])   # ~30 cm of query motion: outside every list's reach

    def same(xs):
        c0, n0 = h0.eval_full(xs)
        c1, n1 = h1.eval_full(xs)
        for a, b in zip(c0, c1):
            assert a.as_dict() == b.as_dict()
        for a, b in zip(n0, n1):
            assert np.array_equal(a.H_np(), b.H_np()) and np.array_equal(a.b_np(), b.b_np()) and a.cost == b.cost and a.counts() == b.counts()

    assert h1.anchor_builds == 0
    same(x1[None, :])                                            # one candidate (BALoss::eval_x): the anchor is built here
    assert h1.anchor_builds == 1 and h0.anchor_builds == 0
    same(synth.perturb(x1, rng, n=9))                            # near the anchor: certified picks
    same(synth.perturb(x1, rng, rot=3e-3, trans=3e-2, scale_rel=6e-3, n=9))   # a spread of ~20 cm: some certify, some search
    assert h1.anchor_builds <= 2                                 # (its central candidate may sit outside the first anchor's reach: the free second set is built at once then)
    for _ in range(5):
        same(synth.perturb(x2, rng, n=5))                        # far away: no certificate of the first anchor holds; a free set, or the older one after four calls, follows
    assert 2 <= h1.anchor_builds <= 3
    tight = dict(rot=1e-4, trans=1e-3, scale_rel=1e-4)           # polls well inside an anchor's reach (6 cm of query motion)
    for _ in range(8):                                           # the two centres in turn: the sets settle on them
        same(synth.perturb(x2, rng, n=5, **tight))
        same(synth.perturb(x1, rng, n=5, **tight))
    nb = h1.anchor_builds
    assert nb <= 5
    for _ in range(4):
        same(synth.perturb(x2, rng, n=5, **tight))               # an optimiser's two incumbents: each keeps its lists
        same(synth.perturb(x1, rng, n=5, **tight))
    assert h1.anchor_builds == nb
    same(np.concatenate([synth.perturb(x1, rng, n=4, **tight), synth.perturb(x2, rng, n=4, **tight)]))   # both in one batch: every candidate reads the set nearest to it
    assert h1.anchor_builds == nb
    # both sets taken: a third centre replaces the least recently used one — a set is replaced at most every fourth call since ITS build
    x3 = x1 - np.array([0.004, -0.003, 0.002, 0.05, -0.04, 0.03, 0.3])
    x4 = x1 + np.array([-0.004, 0.003, 0.002, -0.05, 0.04, 0.03, -0.2])
    x5 = x1 + np.array([0.003, 0.004, -0.002, 0.04, 0.05, -0.03, 0.2])
    same(synth.perturb(x3, rng, n=5))
    assert h1.anchor_builds == nb + 1
    same(synth.perturb(x4, rng, n=5))
    assert h1.anchor_builds == nb + 2
    same(synth.perturb(x5, rng, n=5))                            # both sets are younger than four calls: no rebuild, every lane searches the tree
    assert h1.anchor_builds == nb + 2
    for _ in range(4):
        same(synth.perturb(x2, rng, n=5, **tight))
    assert h1.anchor_builds == nb + 3                            # the anchors have followed the candidates
    same(synth.perturb(x2, rng, n=40, **tight))
    assert h1.anchor_builds == nb + 3
    q = abi.reference_yaml_params()
    q.norm_reg_threshold *= 0.5                                  # the lists carry the planes' verdicts: a parameter change rebuilds them
    q.local_norm_reg_threshold *= 2.0
    h0.set_params(q)
    h1.set_params(q)
    same(synth.perturb(x2, rng, n=6))
    assert h1.anchor_builds == nb + 4
    h0.close()
    h1.close()


def test_dense_scans_leftover_entries_and_long_pair_lists(pkg, synth, abi):
    """120 k points per scan (the reference's scan size): the 8-neighbour lists are complete only out to a few centimetres, so a few
    per cent of the entries fail their certificate and are searched in the tree — by whole waves when a block has only a handful
    (wave_nn_visit) — and a keyframe's pair list outgrows the association kernel's register window (noted pairs). Same bits as the
    per-candidate search without lists, for a batch and for a single candidate."""
    prob, meta = synth.make_scene(n_frames=3, pts_per_frame=120000, seed=5)
    p = abi.reference_yaml_params()
    h0, h1 = _handle(pkg, prob, p, 0), _handle(pkg, prob, p, 1)
    xs = synth.perturb(meta["x_gt"], np.random.default_rng(2), n=24)
    h1.eval_full(xs[:1])                     # the anchor is built around the first call's candidate
    _same_bits(pkg, h0, h1, xs, 1)
    left = h1.nn_left_to_tree
    assert left > 0, "no entry was left to the tree search: the scene no longer exercises that path"
    _same_bits(pkg, h0, h1, xs[:8], 1)
    _same_bits(pkg, h0, h1, xs[5:6], 1)
    h0.close(); h1.close()


def test_pair_lists_are_reused_while_the_batches_stay_inside_their_bound(pkg, synth, abi, scene_small):
    """The pair lists of a call are built for an inflated bound around its reference candidate and serve the following calls whose
    batches stay inside it (an optimiser's late polls, a line search): fewer pair searches than calls, and not one bit of any result
    differs from a handle that searches on every call, nor from the per-candidate path."""
    prob, meta = scene_small
    p = abi.reference_yaml_params()
    h_every = _handle(pkg, prob, p, 1, pair_memo=0)
    h_memo, h_single = _handle(pkg, prob, p, 1), _handle(pkg, prob, p, 0)
    rng = np.random.default_rng(17)
    centre = meta["x_gt"].copy()
    calls = 0
    for step in range(14):
        n = int(rng.choice([1, 3, 8, 14]))
        scale = 1.0 if step < 3 else 0.35   # a few wide polls, then the mesh shrinks
        xs = synth.perturb(centre, rng, rot=3e-4 * scale, trans=3e-3 * scale, scale_rel=1e-3 * scale, n=n)
        centre = xs[0] if step % 4 == 3 else centre   # the incumbent moves now and then
        for fn in ("eval_cost", "eval_full"):
            a, b, c = getattr(h_memo, fn)(xs), getattr(h_every, fn)(xs), getattr(h_single, fn)(xs)
            calls += 1
            assert h_memo.last_path == 1 and h_every.last_path == 1 and h_single.last_path == 0
            ca, cb, cc = (a, b, c) if fn == "eval_cost" else (a[0], b[0], c[0])
            for u, v, w in zip(ca, cb, cc):
                assert u.as_dict() == v.as_dict() == w.as_dict(), (step, fn)
            if fn == "eval_full":
                for u, v, w in zip(a[1], b[1], c[1]):
                    assert np.array_equal(u.H_np(), v.H_np()) and np.array_equal(u.H_np(), w.H_np()) and u.counts() == v.counts() == w.counts()
    assert h_every.pairs_builds == calls
    assert h_memo.pairs_builds < calls // 2, (h_memo.pairs_builds, calls)
    # new parameters: the lists are cut again
    p2 = abi.reference_yaml_params(); p2.max_pixel_dist = 2.5
    before = h_memo.pairs_builds
    h_memo.set_params(p2); h_single.set_params(p2)
    xs = synth.perturb(centre, rng, rot=1e-4, trans=1e-3, scale_rel=1e-4, n=5)
    for u, w in zip(h_memo.eval_cost(xs), h_single.eval_cost(xs)):
        assert u.as_dict() == w.as_dict()
    assert h_memo.pairs_builds == before + 1
    for hh in (h_memo, h_every, h_single):
        hh.close()


def test_a_batch_of_several_tight_groups_shares_one_search_per_group(pkg, synth, abi, ob, scene_small):
    """An optimiser's batch is usually two polls — around its feasible and its infeasible incumbent (iba_mads.hpp) — each tight, far
    from each other: wide as a whole, so round 3 sent it to the per-candidate kernels. The planner clusters such a batch into at
    most four groups with one pair list each (path 2). Same bits as the handle where every candidate searches for itself, whatever
    the grouping, the order of the candidates, the number of groups; lists are reused per group across calls; equal to the oracle."""
    prob, meta = scene_small
    p = abi.reference_yaml_params()
    h0, h1 = _handle(pkg, prob, p, 0), _handle(pkg, prob, p, 1)
    rng = np.random.default_rng(23)
    c1 = meta["x_gt"]
    c2 = c1 + np.array([0.02, -0.015, 0.01, 0.1, -0.08, 0.06, 0.2])      # tens of pixels away
    c3 = c1 + np.array([-0.03, 0.01, 0.02, -0.15, 0.05, -0.1, -0.3])
    c4 = c1 + np.array([0.01, 0.03, -0.02, 0.05, 0.12, 0.1, 0.5])
    two = np.vstack([synth.perturb(c1, rng, n=28), synth.perturb(c2, rng, n=29)])
    _same_bits(pkg, h0, h1, two, 2)
    _same_bits(pkg, h0, h1, two[rng.permutation(len(two))], 2)           # interleaved: the map candidate -> list slot is per candidate
    three = np.vstack([synth.perturb(c1, rng, n=20), synth.perturb(c2, rng, n=1), synth.perturb(c3, rng, n=30)])   # a singleton group
    _same_bits(pkg, h0, h1, three, 2)
    four = np.vstack([synth.perturb(c, rng, rot=8e-4, trans=8e-3, scale_rel=1e-2, n=16) for c in (c1, c2, c3, c4)])  # four looser groups (7-10 px each), 64 candidates
    _same_bits(pkg, h0, h1, four, 2)
    five = np.vstack([four[:50], synth.perturb(c1 + 0.05, rng, n=5)])    # a fifth centre: more groups than slots -> every candidate for itself
    _same_bits(pkg, h0, h1, five, (0, 2))
    # against the oracle
    o = ob.Oracle(prob)
    g, r = h1.eval_cost(two[20:36]), o.eval_cost(p, two[20:36])
    assert h1.last_path == 2
    for a, b in zip(g, r):
        assert (a.n_corr, a.cnt_3d_2d, a.valid_cnt_3d_2d, a.cnt_3d_3d, a.valid_cnt_3d_3d, a.frames_used) == (b.n_corr, b.cnt_3d_2d, b.valid_cnt_3d_2d, b.cnt_3d_3d, b.valid_cnt_3d_3d, b.frames_used)
        assert abs(a.f1 - b.f1) <= 1e-10 * abs(b.f1) and abs(a.f2 - b.f2) <= 1e-10 * abs(b.f2)
    # a drifting two-centre sequence (the late polls of a mesh search): both centres keep their lists across calls
    before, calls = h1.pairs_builds, 0
    a1, a2 = c1.copy(), c2.copy()
    for step in range(10):
        xs = np.vstack([synth.perturb(a1, rng, rot=1e-4, trans=1e-3, scale_rel=3e-4, n=14), synth.perturb(a2, rng, rot=1e-4, trans=1e-3, scale_rel=3e-4, n=15)])
        if step % 3 == 2:
            a1 = xs[0]
        u, w = h1.eval_cost(xs), h0.eval_cost(xs)
        assert h1.last_path == 2
        calls += 1
        for x_, y_ in zip(u, w):
            assert x_.as_dict() == y_.as_dict(), step
    assert h1.pairs_builds - before < calls, (h1.pairs_builds - before, calls)   # (two searches per call without reuse)
    # one group of the two alone, then the other: single-group calls find the slots the clustered calls left
    n0 = h1.pairs_builds
    for x_, y_ in zip(h1.eval_cost(synth.perturb(a2, rng, rot=5e-5, trans=5e-4, scale_rel=1e-4, n=6)), h0.eval_cost(synth.perturb(a2, np.random.default_rng(1), rot=5e-5, trans=5e-4, scale_rel=1e-4, n=6))):
        pass
    xs = synth.perturb(a2, rng, rot=5e-5, trans=5e-4, scale_rel=1e-4, n=6)
    for x_, y_ in zip(h1.eval_cost(xs), h0.eval_cost(xs)):
        assert x_.as_dict() == y_.as_dict()
    assert h1.last_path == 1
    # plane_cache = 0 runs on the same association
    q = abi.reference_yaml_params(plane_cache=0)
    h0.set_params(q); h1.set_params(q)
    _same_bits(pkg, h0, h1, two[18:40], 2)
    h0.close(); h1.close()
