"""Pins the oracle's kd-tree/kNN restatement (oracle/oracle_kdtree.hpp) to the reference's vendored
nanoflann v1.5.0: (1) committed golden vectors generated from the reference's own headers
(tests/golden/make_golden.py), (2) live comparison against oracle/_ref when that build is present."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden", "knn_nanoflann_v150.npz")
CASES = ["2d_leaf10", "3d_leaf30", "3d_leaf30_small", "2d_leaf10_dups"]


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("k", [1, 30])
def test_oracle_knn_matches_nanoflann_golden(ob, name, k):
    z = np.load(GOLD)
    dim, leaf = (int(v) for v in z[f"{name}_meta"])
    idx, d2, cnt = ob.knn("oracle", dim, z[f"{name}_pts"], leaf, z[f"{name}_q"], k)
    # bit-exact: indices (including tie order: first visited wins, nanoflann.hpp:213), squared distances, counts
    assert np.array_equal(cnt, z[f"{name}_k{k}_cnt"])
    assert np.array_equal(idx, z[f"{name}_k{k}_idx"])
    assert np.array_equal(d2, z[f"{name}_k{k}_d2"])


def test_oracle_knn_matches_live_reference_build(ob):
    if ob.ref_lib() is None:
        pytest.skip("oracle/_ref not built (no /root/reference on this box)")
    rng = np.random.default_rng(1)
    for dim, leaf in ((2, 10), (3, 30), (3, 1)):
        for n in (1, 7, 31, 2500):
            pts = rng.normal(size=(n, dim)).astype(np.float32).astype(np.float64)
            if n > 100:
                pts[50:60] = pts[40:50]
            q = np.vstack([rng.normal(size=(64, dim)), pts[: min(n, 32)]])
            for k in (1, 30):
                a = ob.knn("oracle", dim, pts, leaf, q, k)
                b = ob.knn("ref", dim, pts, leaf, q, k)
                assert all(np.array_equal(x, y) for x, y in zip(a, b)), (dim, leaf, n, k)


def test_knn_is_exact_vs_brute_force(ob):
    rng = np.random.default_rng(2)
    pts = (rng.normal(size=(3000, 3)) * [20, 8, 1.5]).astype(np.float32).astype(np.float64)
    q = rng.normal(size=(50, 3)) * [20, 8, 1.5]
    idx, d2, cnt = ob.knn("oracle", 3, pts, 30, q, 30)
    for i in range(len(q)):
        d = ((q[i] - pts) ** 2).sum(1)
        o = np.argsort(d, kind="stable")[:30]
        assert np.array_equal(np.sort(idx[i]), np.sort(o))
        assert np.allclose(d2[i], d[o], rtol=1e-14)


def test_knn_empty_and_tiny(ob):
    idx, d2, cnt = ob.knn("oracle", 3, np.zeros((0, 3)), 30, np.zeros((2, 3)), 5)
    assert np.all(cnt == 0)
    idx, d2, cnt = ob.knn("oracle", 2, np.array([[1.0, 2.0], [3.0, 4.0]]), 10, np.array([[0.0, 0.0]]), 5)
    assert cnt[0] == 2 and list(idx[0][:2]) == [0, 1] and d2[0][0] == 5.0


def test_oracle_knn_on_the_kitti_sized_fixture(ob, synth):
    """tests/golden/knn_nanoflann_v150_big.npz (round 5): nanoflann's kNN(30) / 1-NN INDICES on a 40 000-point scan that is regenerated from its
    seed (hash-checked). The restated tree must return the same indices; the GPU tier runs the device searches against the same file."""
    import hashlib
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "knn_nanoflann_v150_big.npz"))
    nf, ppf, nk, seed, nm, sk = (int(v) for v in z["scene"])
    prob, _ = synth.make_scene(n_frames=nf, pts_per_frame=ppf, n_keypoints=nk, seed=seed, new_mappoints=nm, scan_kp=sk)
    pts = prob.frame_points(0).astype(np.float32)
    assert np.array_equal(np.frombuffer(hashlib.sha256(pts.tobytes()).digest(), np.uint8), z["pts_sha256"])
    p64 = pts.astype(np.float64)
    idx, d2, cnt = ob.knn("oracle", 3, p64, 30, p64[z["self_sel"].astype(np.int64)], 30)
    assert np.all(cnt == 30) and np.array_equal(idx, z["self_k30_idx"].astype(np.uint32))
    i1, d1, _ = ob.knn("oracle", 3, p64, 30, z["q"], 1)
    assert np.array_equal(i1[:, 0], z["q_k1_idx"].astype(np.uint32))
