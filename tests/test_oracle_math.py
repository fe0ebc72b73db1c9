"""Known-answer tests of the oracle's closed-form math against independent implementations
(scipy / numpy), as SURVEY.md §8(c) prescribes in the absence of reference tests."""
import numpy as np
import pytest
from scipy.linalg import expm, logm
from scipy.spatial.transform import Rotation


def _hat(w):
    return np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0.0]])


def _se3_exp_ref(x):
    A = np.zeros((4, 4))
    A[:3, :3] = _hat(x[:3])
    A[:3, 3] = x[3:6]
    T = expm(A)
    return T[:3, :3], T[:3, 3]


@pytest.mark.parametrize("scale", [1e-6, 5e-5, 0.3, 1.7, 3.0])
def test_sim3exp_vs_matrix_exponential(ob, scale):
    rng = np.random.default_rng(int(scale * 1e6) % 97)
    for _ in range(10):
        w = rng.normal(size=3)
        w *= scale / np.linalg.norm(w)
        x = np.concatenate([w, rng.normal(size=3), [rng.uniform(0.5, 12)]])
        R, t, s = ob.sim3exp(x)
        Rr, tr = _se3_exp_ref(x)
        assert np.allclose(R, Rr, atol=1e-9) and np.allclose(t, tr, atol=1e-9) and s == x[6]   # s enters raw (g2o_tools.h:138)
        assert np.allclose(R, Rotation.from_rotvec(w).as_matrix(), atol=1e-9)
        R2, t2 = ob.se3exp(x[:6])
        assert np.array_equal(R, R2) and np.array_equal(t, t2)
        Ri, ti = ob.se3exp(-x[:6])      # exp(-xi) = exp(xi)^-1 (IBACalib2.hpp:573-577)
        assert np.allclose(Ri, R.T, atol=1e-12) and np.allclose(ti, -R.T @ t, atol=1e-12)


def test_se3log_vs_logm_and_roundtrip(ob):
    rng = np.random.default_rng(4)
    for _ in range(20):
        w = rng.normal(size=3)
        w *= rng.uniform(0.001, 3.0) / np.linalg.norm(w)   # |omega| < pi: principal branch
        x = np.concatenate([w, rng.normal(size=3)])
        R, t = _se3_exp_ref(x)
        l = ob.se3log(R, t)
        assert np.allclose(l, x, atol=1e-9)
        T = np.eye(4)
        T[:3, :3], T[:3, 3] = R, t
        L = np.real(logm(T))
        assert np.allclose(l[:3], [L[2, 1], L[0, 2], L[1, 0]], atol=1e-8) and np.allclose(l[3:], L[:3, 3], atol=1e-8)
    # small-angle branch |d| > 0.99999 and a slightly non-orthonormal (float32-valued) rotation
    x = np.array([1e-4, -2e-4, 1.5e-4, 0.3, -0.1, 0.2])
    R, t = _se3_exp_ref(x)
    assert np.allclose(ob.se3log(R, t), x, atol=1e-9)
    R32 = R.astype(np.float32).astype(np.float64)
    assert np.allclose(ob.se3log(R32, t), x, atol=1e-6)


def test_covariance_is_one_pass_raw_moment_form(ob):
    rng = np.random.default_rng(5)
    pts = rng.normal(size=(200, 3)) * [0.3, 0.2, 0.01] + [20.0, -5.0, -1.7]
    idx = rng.choice(200, 25, replace=False).astype(np.uint32)
    c = ob.covariance(pts, idx)
    assert np.allclose(c, np.cov(pts[idx].T, bias=True), rtol=1e-8, atol=1e-12) and np.array_equal(c, c.T)
    assert np.array_equal(ob.covariance(pts, np.zeros(0, np.uint32)), np.eye(3))   # pointcloud.h:128-130


def test_fast_eigen_smallest_eigenvector(ob):
    rng = np.random.default_rng(6)
    for _ in range(200):
        A = rng.normal(size=(3, 3))
        S = A @ np.diag(rng.uniform(1e-4, 1.0, 3) ** 2) @ A.T
        S = 0.5 * (S + S.T)
        v, ev = ob.fast_eigen(S)
        w, V = np.linalg.eigh(S)
        assert abs(abs(v @ V[:, 0]) - 1) < 1e-6, (v, V[:, 0])
        assert np.allclose(np.sort(ev) * S.max(), w, rtol=1e-6, atol=1e-9 * abs(w).max())   # evals of the SCALED matrix (:386)
    # planar cloud: exact zero eigenvalue
    P = rng.normal(size=(30, 3)) * [1, 1, 0]
    v, _ = ob.fast_eigen(np.cov(P.T, bias=True))
    assert abs(abs(v[2]) - 1) < 1e-9
    # diagonal fallbacks (:453-461) and the all-zero matrix (:387-389)
    assert list(ob.fast_eigen(np.diag([0.5, 2.0, 3.0]))[0]) == [1, 0, 0]
    assert list(ob.fast_eigen(np.diag([2.0, 0.5, 3.0]))[0]) == [0, 1, 0]
    assert list(ob.fast_eigen(np.diag([2.0, 3.0, 0.5]))[0]) == [0, 0, 1]
    assert list(ob.fast_eigen(np.zeros((3, 3)))[0]) == [0, 0, 0]


def test_sim3exp_jets_vs_central_differences(ob):
    rng = np.random.default_rng(7)
    for _ in range(5):
        x = np.concatenate([rng.normal(size=3) * 0.7, rng.normal(size=3), [10.0]])
        R, t, dR, dt = ob.sim3exp_jet(x)
        for k in range(6):
            h = 1e-6
            xp, xm = x.copy(), x.copy()
            xp[k] += h
            xm[k] -= h
            Rp, tp, _ = ob.sim3exp(xp)
            Rm, tm, _ = ob.sim3exp(xm)
            assert np.allclose(dR[:, k], ((Rp - Rm) / (2 * h)).reshape(9), atol=1e-7)
            assert np.allclose(dt[:, k], (tp - tm) / (2 * h), atol=1e-7)
        assert np.all(dR[:, 6] == 0) and np.all(dt[:, 6] == 0)


def test_huber_matches_formula(ob):
    for a in (1.0, 2.98):
        for s in (0.0, 0.5, a * a, a * a + 1e-9, 50.0):
            r0, r1 = ob.huber(a, s)
            if s <= a * a:
                assert r0 == s and r1 == 1.0
            else:
                assert np.isclose(r0, 2 * a * np.sqrt(s) - a * a) and np.isclose(r1, a / np.sqrt(s))
