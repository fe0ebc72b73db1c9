"""Shapes that push the frame kernel off its common paths: more MapPoint keypoints than threads (several passes of the first
kd round, parked queries beyond the LDS capacity), 3000 keypoints per frame, per-evaluation plane refits on long work
lists, and a candidate queue too small for the survivors of the pre-cull (inline exact path + rescan)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _check(pkg, synth, ob, prob, meta, params):
    h = pkg.IbaHandle(prob, params)
    o = ob.Oracle(prob)
    xs = np.vstack([meta["x_gt"][None], synth.perturb(meta["x_gt"], np.random.default_rng(1), n=2),
                    synth.perturb(meta["x_gt"], np.random.default_rng(2), rot=0.02, trans=0.1, scale_rel=0.03, n=1)])
    gc, gn = h.eval_full(xs)
    oc, on = o.eval_cost(params, xs), o.eval_normal(params, xs)
    for a, b in zip(gc, oc):
        for k in ("valid_cnt_3d_2d", "cnt_3d_2d", "cnt_3d_3d", "valid_cnt_3d_3d", "valid_pl_3d_3d", "valid_pt_3d_3d", "frames_used", "n_corr"):
            assert getattr(a, k) == getattr(b, k), k
        assert abs(a.f1 - b.f1) <= 1e-10 * abs(b.f1) and abs(a.f2 - b.f2) <= 1e-10 * abs(b.f2)
    for a, b in zip(gn, on):
        assert a.counts() == b.counts()
        assert np.allclose(a.H_np(), b.H_np(), rtol=1e-9, atol=1e-9 * np.abs(b.H_np()).max())
    n = gc[0].n_corr / prob.n_frames
    h.close()
    return n


def test_more_mappoint_keypoints_than_threads(pkg, synth, abi, ob):
    prob, meta = synth.make_scene(n_frames=4, pts_per_frame=20000, n_keypoints=2000, seed=3, new_mappoints=900, scan_kp=300)
    assert _check(pkg, synth, ob, prob, meta, abi.reference_yaml_params()) > 512


def test_three_thousand_keypoints(pkg, synth, abi, ob):
    prob, meta = synth.make_scene(n_frames=4, pts_per_frame=8000, n_keypoints=3000, seed=4, new_mappoints=1200, scan_kp=600)
    _check(pkg, synth, ob, prob, meta, abi.reference_yaml_params())


def test_plane_refit_on_long_lists(pkg, synth, abi, ob):
    prob, meta = synth.make_scene(n_frames=3, pts_per_frame=12000, n_keypoints=2000, seed=5, new_mappoints=700, scan_kp=300)
    _check(pkg, synth, ob, prob, meta, abi.reference_yaml_params(plane_cache=0))


def test_candidate_queue_overflow(pkg, synth, abi, ob):
    os.environ["IBA_CAND_BYTES"] = "2048"   # diagnostic knob read at handle creation: queue of 512 entries
    try:
        prob, meta = synth.make_scene(n_frames=4, pts_per_frame=20000, n_keypoints=2000, seed=6)
        _check(pkg, synth, ob, prob, meta, abi.reference_yaml_params())
    finally:
        del os.environ["IBA_CAND_BYTES"]


def test_pair_list_overflow(pkg, synth, abi, ob):
    os.environ["IBA_PAIR_BYTES"] = "1024"   # diagnostic knob: room for 64 (point, keypoint) pairs -> inline exact tests + rescan for the ties
    try:
        prob, meta = synth.make_scene(n_frames=3, pts_per_frame=9000, n_keypoints=2000, seed=8)
        _check(pkg, synth, ob, prob, meta, abi.reference_yaml_params())
    finally:
        del os.environ["IBA_PAIR_BYTES"]
