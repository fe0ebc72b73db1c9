"""iba_factor2_kernel (round 6, opt-in: IBA_FACTOR_V2=1): the normal equations summed by waves that each take an equal share of a candidate's whole
work list (csrc/iba_factor2_kernel.hpp) instead of one wave per (keyframe, candidate). The cut of a list into ranges may not change what is
summed: every way of cutting (1 range, 3, 7, one per keyframe, the default rule) must give the sums of the one-wave-per-keyframe kernel
of rounds 2-5 (the default) up to summation order, the counts exactly, and the oracle's to the usual bars. Shapes that stress the walk:
keyframes with a handful of entries (a round of 64 entries spans more keyframes than the LDS ring holds), empty keyframes, more covisible
keyframes than one flag word (MANY), the frozen problem's shared list, refitted planes, IBATestEdge."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _handle(pkg, prob, p, monkeypatch, v1=False, waves=0, **kw):
    """v1: the default kernel (one wave per (keyframe, candidate)); otherwise iba_factor2_kernel (IBA_FACTOR_V2=1, a debug override: read only with
    IBA_DEBUG_ENV=1, which tests/conftest.py sets), `waves` ranges per candidate (0: its own rule)"""
    if not v1:
        monkeypatch.setenv("IBA_FACTOR_V2", "1")
    if waves:
        monkeypatch.setenv("IBA_FACTOR_WAVES_PER_CAND", str(waves))
    try:
        h = pkg.IbaHandle(prob, p, **kw)
        assert (h.debug_factor_ranges(4) == 0) == bool(v1)
        return h
    finally:
        monkeypatch.delenv("IBA_FACTOR_V2", raising=False)
        monkeypatch.delenv("IBA_FACTOR_WAVES_PER_CAND", raising=False)


def _close(a, b, tol=1e-12):
    """counts equal, H / b / cost / chi2 equal to summation order (tol of the largest entry; entries are sums of thousands of terms)"""
    assert a.counts() == b.counts(), (a.counts(), b.counts())
    sH, sb = np.abs(b.H_np()).max(), np.abs(b.b_np()).max()
    assert np.max(np.abs(a.H_np() - b.H_np())) <= tol * sH, np.max(np.abs(a.H_np() - b.H_np())) / sH
    assert np.max(np.abs(a.b_np() - b.b_np())) <= tol * sb
    assert abs(a.cost - b.cost) <= tol * abs(b.cost) and abs(a.chi2 - b.chi2) <= tol * abs(b.chi2)


def test_every_cut_of_the_lists_gives_the_same_sums(pkg, synth, abi, ob, scene_small, monkeypatch):
    prob, meta = scene_small
    p = abi.reference_yaml_params()
    rng = np.random.default_rng(5)
    xs = np.vstack([meta["x_gt"][None], synth.perturb(meta["x_gt"], rng, n=5)])
    h1 = _handle(pkg, prob, p, monkeypatch, v1=True)
    ref = h1.eval_normal(xs)
    assert sum(r.n_factor_3d2d + r.n_factor_p2pl + r.n_factor_p2pt for r in ref) > 3000
    o = ob.Oracle(prob)
    orc = o.eval_normal(p, xs)
    for waves in (0, 1, 3, 7, 12):   # 0: the default rule (one range per keyframe at this size); 1: ONE wave walks all 12 keyframes of a candidate
        h = _handle(pkg, prob, p, monkeypatch, waves=waves)
        got = h.eval_normal(xs)
        for a, b, c in zip(got, ref, orc):
            _close(a, b)
            assert a.counts() == c.counts()
            assert np.max(np.abs(a.H_np() - c.H_np())) <= 1e-9 * np.abs(c.H_np()).max()
        # the cost tuple's chain with the normal equations behind it, and single candidates
        cf, nf = h.eval_full(xs[:3])
        for a, b in zip(nf, ref[:3]):
            _close(a, b)
        _close(h.eval_normal(xs[4])[0], ref[4])
        # the frozen problem: every candidate walks the SAME list (row 0)
        h.build_problem(xs[1]); h1.build_problem(xs[1])
        for a, b in zip(h.eval_factors(xs[:4]), h1.eval_factors(xs[:4])):
            _close(a, b)
        h.close()
    h1.close()


def test_keyframes_with_a_handful_of_entries_and_empty_ones(pkg, synth, abi, ob, monkeypatch):
    """40 keyframes of 60 keypoints: a keyframe's list holds a few entries (or none: fewer than 30 correspondences skip the keyframe), so a round of
    64 entries spans more keyframes than the ring of 8 holds, queued blocks wait across many keyframes, and slots are overwritten under them"""
    prob, meta = synth.make_scene(n_frames=40, pts_per_frame=3000, n_keypoints=200, seed=9)
    p = abi.reference_yaml_params()
    p.num_min_corr = 4   # (BuildProblem's own threshold, iba_local.cpp:203: sparse keyframes stay in with a few blocks each)
    rng = np.random.default_rng(2)
    xs = synth.perturb(meta["x_gt"], rng, n=4)
    h1 = _handle(pkg, prob, p, monkeypatch, v1=True)
    ref = h1.eval_normal(xs)
    o = ob.Oracle(prob)
    orc = o.eval_normal(p, xs)
    total = sum(r.n_factor_3d2d + r.n_factor_p2pl + r.n_factor_p2pt for r in ref)
    assert total > 100 and min(r.frames_used for r in ref) >= 20   # a few blocks per keyframe, most keyframes in
    for waves in (1, 2, 5, 0):
        h = _handle(pkg, prob, p, monkeypatch, waves=waves)
        for a, b, c in zip(h.eval_normal(xs), ref, orc):
            _close(a, b)
            assert a.counts() == c.counts()
        h.close()
    h1.close()


def test_many_covisible_keyframes_refit_and_test_edge(pkg, synth, abi, ob, monkeypatch):
    # more covisible keyframes than the flag word holds (MANY: two match words, a ring of fewer, larger records)
    prob, meta = synth.make_scene(n_frames=40, pts_per_frame=3000, n_keypoints=500, n_covis=34, seed=4)
    p = abi.reference_yaml_params()
    p.num_min_corr = 4   # (~8 plane factors per keyframe, every keyframe in)
    xs = synth.perturb(meta["x_gt"], np.random.default_rng(8), n=3)
    h1 = _handle(pkg, prob, p, monkeypatch, v1=True)
    ref = h1.eval_normal(xs)
    assert min(r.frames_used for r in ref) == 40
    for waves in (1, 4, 0):
        h = _handle(pkg, prob, p, monkeypatch, waves=waves)
        for a, b in zip(h.eval_normal(xs), ref):
            _close(a, b)
        h.close()
    h1.close()
    # planes refitted per evaluation (every candidate reads its own plane records) and the IBATestEdge blocks
    prob, meta = synth.make_scene(n_frames=10, pts_per_frame=4000, seed=6)
    xs = synth.perturb(meta["x_gt"], np.random.default_rng(3), n=4)
    o = ob.Oracle(prob)
    for kw in (dict(plane_cache=0), dict(factor_3d2d_kind=1), dict(factor_3d2d_kind=1, w1=0.0)):
        q = abi.reference_yaml_params(plane_cache=kw.get("plane_cache", 1))
        q.factor_3d2d_kind = kw.get("factor_3d2d_kind", 0)
        q.err_weight[1] = kw.get("w1", 1.0)
        h1 = _handle(pkg, prob, q, monkeypatch, v1=True)
        ref = h1.eval_normal(xs)
        orc = o.eval_normal(q, xs)
        for waves in (1, 3, 0):
            h = _handle(pkg, prob, q, monkeypatch, waves=waves)
            for a, b, c in zip(h.eval_normal(xs), ref, orc):
                _close(a, b)
                assert a.counts() == c.counts() and np.max(np.abs(a.H_np() - c.H_np())) <= 1e-9 * np.abs(c.H_np()).max()
            h.close()
        h1.close()


def test_a_batch_that_fills_the_machine(pkg, synth, abi, ob, monkeypatch):
    """64 candidates x 24 keyframes: the default rule gives every candidate 24 ranges... and a batch of 300 gives each 6, across keyframe boundaries;
    both against the one-wave-per-keyframe kernel, the first three candidates against the oracle"""
    prob, meta = synth.make_scene(n_frames=24, pts_per_frame=5000, seed=12)
    p = abi.reference_yaml_params()
    xs = synth.perturb(meta["x_gt"], np.random.default_rng(1), n=300)
    h, h1 = _handle(pkg, prob, p, monkeypatch), _handle(pkg, prob, p, monkeypatch, v1=True)
    o = ob.Oracle(prob)
    for n in (64, 300):
        got, ref = h.eval_normal(xs[:n]), h1.eval_normal(xs[:n])
        for a, b in zip(got, ref):
            _close(a, b)
    for a, c in zip(got[:3], o.eval_normal(p, xs[:3], nthreads=4)):
        assert a.counts() == c.counts() and np.max(np.abs(a.H_np() - c.H_np())) <= 1e-9 * np.abs(c.H_np()).max()
    h.close(); h1.close()


def test_environment_overrides_need_the_debug_switch(pkg, synth, abi, scene_small, monkeypatch):
    """IBA_FACTOR_V2=1 (like every environment override of the library) is read only together with IBA_DEBUG_ENV=1"""
    prob, meta = scene_small
    p = abi.reference_yaml_params()
    monkeypatch.setenv("IBA_FACTOR_V2", "1")
    monkeypatch.setenv("IBA_DEBUG_ENV", "0")
    h = pkg.IbaHandle(prob, p)
    assert h.debug_factor_ranges(64) == 0
    h.close()
    monkeypatch.setenv("IBA_DEBUG_ENV", "1")
    h = pkg.IbaHandle(prob, p)
    assert h.debug_factor_ranges(64) > 0
    h.close()
