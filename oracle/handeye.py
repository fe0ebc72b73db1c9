"""TEST INFRASTRUCTURE (oracle): hand-eye initialiser of the reference restated with numpy / scipy, independently of
csrc/iba_handeye.cpp (numpy's LAPACK SVD and solve instead of the hand-written Jacobi / elimination, scipy's
least_squares with a Huber loss instead of the hand-written LM). Only tests/ may import this.

Restates HECalib (HECalib.h:12-57), pose2Motion (kitti_tools.h:160-165), the EdgeHE residual (NLHECalib.hpp:27-48) and the
cost HECalibRobustKernelg2o minimises (:121-163) and the line-process annealing of HECalibLineProcessg2o (:189-277). Eigen and g2o are absent from the image: PARITY WITH THEM IS UNPINNED;
what is pinned is agreement of two independent restatements and recovery of planted extrinsics."""
import numpy as np
from scipy.optimize import least_squares
from scipy.spatial.transform import Rotation


def pose_to_motion(poses):
    poses = np.asarray(poses, np.float64)
    T = np.tile(np.eye(4), (len(poses), 1, 1))
    T[:, :3, :4] = poses[:, :3, :4]
    return np.array([(T[i + 1] @ np.linalg.inv(T[i]))[:3, :4] for i in range(len(T) - 1)])


def rotvec(R):
    return Rotation.from_matrix(R).as_rotvec()


def handeye(Ta, Tb):
    Ta, Tb = np.asarray(Ta, np.float64), np.asarray(Tb, np.float64)
    alpha = np.array([rotvec(T[:3, :3]) for T in Ta])
    beta = np.array([rotvec(T[:3, :3]) for T in Tb])
    H = (beta - beta.mean(0)).T @ (alpha - alpha.mean(0))          # sum (beta - mean)(alpha - mean)^T
    U, _, Vt = np.linalg.svd(H)
    R = Vt.T @ U.T
    if np.linalg.det(R) < 0:
        Vt[2] *= -1
        R = Vt.T @ U.T
    A = np.concatenate([np.concatenate([T[:3, :3] - np.eye(3), T[:3, 3:4]], 1) for T in Ta], 0)
    b = np.concatenate([R @ T[:3, 3] for T in Tb])
    x = np.linalg.solve(A.T @ A, A.T @ b)
    return np.concatenate([R, x[:3, None]], 1), float(x[3])


def handeye_degenerate(Ta, Tb, dg_threshold=0.01):
    """DGHECalib (HECalib.h:66-120): the rotation as handeye's (:73-107), NO translation (:109), scale = sum |ta| |tb| / sum |ta|^2 over the pairs whose
    camera rotation angle is below dg_threshold (:82, :112-119). -> (rigid 3x4, scale, number of such pairs)"""
    Ta, Tb = np.asarray(Ta, np.float64), np.asarray(Tb, np.float64)
    alpha = np.array([rotvec(T[:3, :3]) for T in Ta])
    beta = np.array([rotvec(T[:3, :3]) for T in Tb])
    H = (beta - beta.mean(0)).T @ (alpha - alpha.mean(0))
    U, _, Vt = np.linalg.svd(H)
    R = Vt.T @ U.T
    if np.linalg.det(R) < 0:
        Vt[2] *= -1
        R = Vt.T @ U.T
    dg = np.linalg.norm(alpha, axis=1) < dg_threshold
    ta, tb = np.linalg.norm(Ta[dg, :3, 3], axis=1), np.linalg.norm(Tb[dg, :3, 3], axis=1)
    with np.errstate(invalid="ignore", divide="ignore"):
        scale = float(np.float64(np.sum(ta * tb)) / np.float64(np.sum(ta * ta)))
    return np.concatenate([R, np.zeros((3, 1))], 1), scale, int(dg.sum())


def sim3_exp(x):
    w = np.asarray(x[:3], np.float64)
    th = np.linalg.norm(w)
    Om = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-4:
        a, b, c = 1.0, 0.5, 1.0 / 6.0
    else:
        a, b, c = np.sin(th) / th, (1 - np.cos(th)) / th ** 2, (th - np.sin(th)) / th ** 3
    R = np.eye(3) + a * Om + b * Om @ Om
    V = np.eye(3) + b * Om + c * Om @ Om
    return R, V @ np.asarray(x[3:6], np.float64), float(x[6])


def edge_residuals(x, Ta, Tb):
    """EdgeHE::computeError for every pair (weight 1): R beta - alpha + (Ra - I) t + s ta - R tb."""
    R, t, s = sim3_exp(x)
    out = []
    for A, B in zip(Ta, Tb):
        out.append(R @ rotvec(B[:3, :3]) - rotvec(A[:3, :3]) + (A[:3, :3] - np.eye(3)) @ t + s * A[:3, 3] - R @ B[:3, 3])
    return np.array(out)


def robust_cost(x, Ta, Tb, delta, regulation, ratio):
    e2 = (edge_residuals(x, Ta, Tb) ** 2).sum(1)
    rho = np.where(e2 <= delta * delta, e2, 2 * delta * np.sqrt(e2) - delta * delta)
    reg = len(Ta) * ratio * float(np.sum(np.asarray(x[3:6]) ** 2)) if regulation else 0.0
    return float(rho.sum() + reg)


def handeye_robust_minimum(Ta, Tb, rigid0, scale0, delta=0.1, regulation=True, ratio=0.005):
    """Minimiser of the same robust cost with an exact Jacobian (finite differences inside scipy): where the reference's
    optimiser should end up if its approximate Jacobian still leads downhill."""
    x0 = np.concatenate([rotvec(np.asarray(rigid0)[:3, :3]), np.asarray(rigid0)[:3, 3], [scale0]])
    Ta, Tb = np.asarray(Ta, np.float64), np.asarray(Tb, np.float64)

    def fun(x):
        e = edge_residuals(x, Ta, Tb)
        n = np.sqrt((e ** 2).sum(1))
        w = np.where(n <= delta, 1.0, np.sqrt(np.maximum(2 * delta * n - delta * delta, 0)) / np.maximum(n, 1e-300))   # sqrt(rho)/|e|
        r = (e * w[:, None]).reshape(-1)
        if regulation:
            r = np.concatenate([r, np.sqrt(len(Ta) * ratio) * x[3:6]])
        return r

    sol = least_squares(fun, x0, method="lm", xtol=1e-14, ftol=1e-14, gtol=1e-14)
    return sol.x, float(np.sum(sol.fun ** 2))


def handeye_lineprocess(Ta, Tb, rigid0, scale0, mu0=64.0, divid_factor=1.4, min_mu=1e-1, ex_max_iter=20, regulation=True, ratio=0.005):
    """HECalibLineProcessg2o's outer loop (NLHECalib.hpp:228-248) with every inner solve run to convergence by scipy:
    weights w^2 = (mu / (mu + chi2))^2 with chi2 under the previous weights, regulariser information sum(w^2) * ratio."""
    x = np.concatenate([rotvec(np.asarray(rigid0)[:3, :3]), np.asarray(rigid0)[:3, 3], [scale0]])
    Ta, Tb = np.asarray(Ta, np.float64), np.asarray(Tb, np.float64)
    n = len(Ta)
    info = np.ones(n)

    def solve(x0, info, reg_info):
        def fun(xx):
            r = (edge_residuals(xx, Ta, Tb) * np.sqrt(info)[:, None]).reshape(-1)
            if regulation:
                r = np.concatenate([r, np.sqrt(reg_info) * xx[3:6]])
            return r
        return least_squares(fun, x0, method="lm", xtol=1e-14, ftol=1e-14, gtol=1e-14).x

    x = solve(x, info, n * ratio)
    mu = mu0
    for _ in range(ex_max_iter):
        e2 = info * (edge_residuals(x, Ta, Tb) ** 2).sum(1)
        w = mu / (mu + e2)
        info = w * w
        x = solve(x, info, info.sum() * ratio)
        mu /= divid_factor
        if mu < min_mu:
            break
    return x, info
