"""The launch shapes bench.py times, against the oracle, inside the driver-run suite.

bench.py's headline evaluates 64 tight candidates on 200 keyframes x 10 k points x 2000 keypoints: 12 800 (candidate, keyframe) blocks, so the
shared-pair association runs as iba_assoc2_kernel<4, false, 256> (blocks of 256 threads from 1024 blocks up: csrc/iba_capi.hip, assoc2_threads),
the search kernel with an odd number of slices per keyframe, the factor kernel on its plain (keyframe, candidate) grid. Until round 5 only a
builder-run soak held exactly that launch against the oracle. Here:
  * bench.py's own scene (seed 0) and its first batch (xs_all[0]: rng(0), 64 candidates), one fused evaluation: every counter of every candidate
    exact, f1 / f2 / C 1e-10, the normal equations of every candidate within 2e-9 of the double oracle per entry and, for the first candidates, held to
    the long-double truth (tests/parity_gate.py); the launch is asserted to have used 256-thread association blocks;
  * 43 and 62 covisible keyframes per keyframe with the association FORCED to 256-thread blocks (IBA_ASSOC2_THREADS=256): the staging loop of the
    relative poses runs more than once per thread there (MANY);
  * three keyframes of 120 k points with 24 candidates: the dense-scan path of the pair search (blocks that test their boxes first) under 256-thread blocks."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import parity_gate  # noqa: E402

pytestmark = pytest.mark.gpu
INT = ("valid_cnt_3d_2d", "cnt_3d_2d", "cnt_3d_3d", "valid_cnt_3d_3d", "valid_pl_3d_3d", "valid_pt_3d_3d", "frames_used", "n_corr")
NCPU = os.cpu_count() or 8


def _entries(g, o, rel):
    g, o = np.asarray(g, float).ravel(), np.asarray(o, float).ravel()
    big = np.max(np.abs(o))
    m = np.abs(o) > 1e-6 * big
    return float(np.max(np.abs(g - o)[m] / np.abs(o)[m])) <= rel and float(np.max(np.abs(g - o)[~m], initial=0.0)) <= rel * 1e-5 * big


def _cost_equal(a, b):
    for k in INT:
        assert getattr(a, k) == getattr(b, k), (k, getattr(a, k), getattr(b, k))
    assert abs(a.f1 - b.f1) <= 1e-10 * abs(b.f1) and abs(a.f2 - b.f2) <= 1e-10 * abs(b.f2)
    assert (np.isnan(a.C) and np.isnan(b.C)) or abs(a.C - b.C) <= 1e-10 * abs(b.C) + 1e-15


def test_the_headline_launch_against_the_oracle(pkg, synth, abi, ob):
    prob, meta = synth.make_scene(n_frames=200, pts_per_frame=10000, n_keypoints=2000, seed=0)   # bench.py:149
    p = abi.reference_yaml_params()
    xs = synth.perturb(meta["x_gt"], np.random.default_rng(0), n=64)                              # bench.py:171, xs_all[0]
    h = pkg.IbaHandle(prob, p)
    cost, nrm = h.eval_full(xs)
    assert h.last_path == 1 and h.last_assoc2_threads == 256, (h.last_path, h.last_assoc2_threads)
    assert h.debug_factor_ranges(64) == 0   # (the default factor kernel: one wave per (keyframe, candidate))
    o = ob.Oracle(prob)
    oc = o.eval_cost(p, xs, nthreads=min(NCPU, 64))
    on = o.eval_normal(p, xs, nthreads=min(NCPU, 64))
    worst, explained = 0.0, []
    for b in range(64):
        _cost_equal(cost[b], oc[b])
        assert nrm[b].counts() == on[b].counts(), b
        # device vs the double oracle: 2e-9 per entry (what a few ill-conditioned plane factors leave of two double evaluations: tests/parity_gate.py);
        # a candidate beyond that goes through the checked explanation (tests/parity_explain.py)
        near = _entries(nrm[b].H_np(), on[b].H_np(), 2e-9) and _entries(nrm[b].b_np(), on[b].b_np(), 2e-9) and abs(nrm[b].cost - on[b].cost) <= 1e-9 * on[b].cost
        if not near:   # (a block with |r| = 2e4 px moves the cost itself by 2e-9 of the total: candidate 32's kind)
            import parity_explain
            res = parity_explain.explain(h, o, p, xs[b], nthreads=min(NCPU, 64))
            assert res["status"] == "explained" and res["flagged"] > 0, (b, res)
            explained.append(b)
        worst = max(worst, parity_gate.worst_rel(nrm[b].H_np(), on[b].H_np())[0])
    # ... and against the long-double evaluation: a few candidates, and every one that needed the explanation. A candidate that misses this gate too
    # (the device computes with ITS plane normals, which differ from the oracle's in the last bits, and an ill-conditioned block amplifies that)
    # must be one the block-by-block explanation accounts for
    for b in sorted(set([0, 1, 2, 63] + explained)):
        try:
            parity_gate.normal_vs_truth(nrm[b], on[b], o.eval_normal_truth(p, xs[b])[0])
        except AssertionError:
            if b not in explained:
                import parity_explain
                res = parity_explain.explain(h, o, p, xs[b], nthreads=min(NCPU, 64))
                assert res["status"] == "explained" and res["flagged"] > 0, (b, res)
    # the cost-only chain at the same launch shape: the same cost tuple
    for a, b in zip(h.eval_cost(xs), cost):
        for k in INT:
            assert getattr(a, k) == getattr(b, k)
        assert a.C == b.C and abs(a.f1 - b.f1) <= 1e-13 * b.f1
    h.close()


@pytest.mark.parametrize("n_covis", [43, 62])
def test_many_covisible_keyframes_on_256_thread_association_blocks(pkg, synth, abi, ob, n_covis, monkeypatch):
    prob, meta = synth.make_scene(n_frames=n_covis + 4, pts_per_frame=2000, n_keypoints=600, seed=47, n_covis=n_covis, new_mappoints=100, scan_kp=140)
    p = abi.reference_yaml_params()
    monkeypatch.setenv("IBA_ASSOC2_THREADS", "256")
    h = pkg.IbaHandle(prob, p)
    monkeypatch.delenv("IBA_ASSOC2_THREADS")
    o = ob.Oracle(prob)
    xs = np.vstack([meta["x_gt"][None], synth.perturb(meta["x_gt"], np.random.default_rng(47), n=5)])
    cost, nrm = h.eval_full(xs)
    assert h.last_path == 1 and h.last_assoc2_threads == 256
    for a, b in zip(cost, o.eval_cost(p, xs, nthreads=8)):
        _cost_equal(a, b)
    assert cost[0].cnt_3d_2d > 4 * cost[0].n_corr   # matches in the slots beyond the 42nd
    for a, b in zip(nrm, o.eval_normal(p, xs, nthreads=8)):
        assert a.counts() == b.counts() and np.max(np.abs(a.H_np() - b.H_np())) <= 1e-8 * np.abs(b.H_np()).max()
    # the same candidates on the default handle (512-thread blocks at this size): the same bits
    h2 = pkg.IbaHandle(prob, p)
    c2, n2 = h2.eval_full(xs)
    assert h2.last_assoc2_threads == 512
    for a, b in zip(cost, c2):
        assert a.as_dict() == b.as_dict() or all(x == y or (x != x and y != y) for x, y in zip(a.as_dict().values(), b.as_dict().values()))
    for a, b in zip(nrm, n2):
        assert np.array_equal(a.H_np(), b.H_np()) and np.array_equal(a.b_np(), b.b_np())
    h.close(); h2.close()


def test_dense_scans_on_256_thread_association_blocks(pkg, synth, abi, ob, monkeypatch):
    prob, meta = synth.make_scene(n_frames=3, pts_per_frame=120000, n_keypoints=2000, seed=5)
    p = abi.reference_yaml_params()
    monkeypatch.setenv("IBA_ASSOC2_THREADS", "256")
    h = pkg.IbaHandle(prob, p)
    monkeypatch.delenv("IBA_ASSOC2_THREADS")
    o = ob.Oracle(prob)
    xs = synth.perturb(meta["x_gt"], np.random.default_rng(6), n=24)
    cost, nrm = h.eval_full(xs)
    assert h.last_path == 1 and h.last_assoc2_threads == 256
    for a, b in zip(cost, o.eval_cost(p, xs, nthreads=min(NCPU, 64))):
        _cost_equal(a, b)
    for a, b in zip(nrm, o.eval_normal(p, xs, nthreads=min(NCPU, 64))):
        assert a.counts() == b.counts() and np.max(np.abs(a.H_np() - b.H_np())) <= 1e-8 * np.abs(b.H_np()).max()
    h.close()
