"""ORB-only extrinsic BA on the device (csrc/iba_ba.hip) against the CPU oracle: one linearisation (H, b, chi2, per-edge
chi2), the whole optimise/classify schedule, and recovery of a planted extrinsic."""
import importlib
import os
import sys

import numpy as np
import pytest
from scipy.spatial.transform import Rotation

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
ba = importlib.import_module("spatial-temporal-lidar-camera-calibration_amd.ba")
from oracle import ba as oba  # noqa: E402
import ba_scene  # noqa: E402

pytestmark = pytest.mark.gpu


def test_linearisation_matches_oracle():
    prob, x_gt = ba_scene.make(n_frames=25, pts_per_frame=200, seed=5, ba=ba)
    h = ba.BaHandle(prob)
    rng = np.random.default_rng(2)
    N = len(prob.edge_frame)
    for trial in range(4):
        x = x_gt + np.concatenate([rng.normal(0, 0.01, 3), rng.normal(0, 0.05, 3), [rng.normal(0, 0.3)]])
        active = None if trial == 0 else (rng.random(N) < 0.8).astype(np.uint8)
        robust = trial != 3
        H, b, chi, c2 = h.eval(x, active, robust)
        Ho, bo, chio, c2o = oba.evaluate(prob, x, active, robust)
        assert np.allclose(c2, c2o, rtol=1e-11, atol=1e-12)
        assert abs(chi - chio) <= 1e-11 * chio
        assert np.allclose(H, Ho, rtol=1e-10, atol=1e-10 * np.abs(Ho).max()) and np.allclose(b, bo, rtol=1e-10, atol=1e-10 * np.abs(bo).max())
    H2, b2, chi2_, _ = h.eval(x, active, robust)
    assert np.array_equal(H, H2) and np.array_equal(b, b2) and chi == chi2_      # fixed-order sums: bitwise reproducible
    h.close()


def test_schedule_matches_oracle_and_recovers_planted():
    prob, x_gt = ba_scene.make(n_frames=40, pts_per_frame=300, seed=6, ba=ba)
    rng = np.random.default_rng(3)
    x0 = x_gt + np.concatenate([rng.normal(0, 0.01, 3), rng.normal(0, 0.03, 3), [0.3]])
    h = ba.BaHandle(prob)
    x, r = h.optimize(x0)
    xo, n_in_o, log = oba.optimize(prob, x0)
    assert r.n_inliers == n_in_o and [r.n_bad[i] for i in range(4)] == [l[1] for l in log]
    assert np.allclose(x, xo, rtol=0, atol=1e-8)
    assert np.allclose([r.chi2[i] for i in range(4)], [l[0] for l in log], rtol=1e-9)
    dR = Rotation.from_rotvec(x[:3]).as_matrix() @ Rotation.from_rotvec(x_gt[:3]).as_matrix().T
    assert np.linalg.norm(Rotation.from_matrix(dR).as_rotvec()) < 1e-3
    assert np.linalg.norm(x[3:6] - x_gt[3:6]) < 0.03 and abs(x[6] - x_gt[6]) < 0.03
    assert r.n_edges == len(prob.edge_frame) and 0.9 * r.n_edges < r.n_inliers < r.n_edges
    h.close()


def test_edges_from_dataset_directory(tmp_path):
    """Dataset directory -> iba_dataset_load_ba -> device linearisation and schedule = CPU oracle on the same edge list."""
    synth = importlib.import_module("spatial-temporal-lidar-camera-calibration_amd.synth")
    from oracle import formats as ofmt
    prob, meta = synth.make_scene(n_frames=8, pts_per_frame=1500, n_keypoints=400, seed=13, new_mappoints=80, scan_kp=100)
    paths = ofmt.write_dataset(str(tmp_path), prob, meta)
    edges = ba.load_ba_dataset(**paths)
    assert len(edges.edge_frame) == sum(len(m) for m in meta["mp2kp"])
    R, t, s = synth.sim3_exp(meta["x_gt"])
    x0 = np.concatenate([Rotation.from_matrix(R).as_rotvec(), t, [s]])      # CalibVertex: rotation vector, translation, scale
    h = ba.BaHandle(edges)
    H, b, chi, c2 = h.eval(x0)
    Ho, bo, chio, c2o = oba.evaluate(edges, x0)
    assert np.allclose(c2, c2o, rtol=1e-11, atol=1e-12) and abs(chi - chio) <= 1e-11 * chio
    assert np.allclose(H, Ho, rtol=1e-10, atol=1e-10 * np.abs(Ho).max()) and np.allclose(b, bo, rtol=1e-10, atol=1e-10 * np.abs(bo).max())
    x, r = h.optimize(x0)
    xo, n_in_o, log = oba.optimize(edges, x0)
    assert r.n_inliers == n_in_o and np.allclose(x, xo, rtol=0, atol=1e-7)
    h.close()
