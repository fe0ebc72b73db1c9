"""Synthetic ORB-only extrinsic BA problem with a planted (T_cl, scale): MapPoints of a monocular map (camera-0 frame,
unit = metres / scale), LiDAR odometry poses, keypoint observations with pixel noise and a fraction of gross outliers.
Edge constants are built the way OptimizeExtrinsicGlobal does (Optimizer.cc:1620-1676)."""
import numpy as np
from scipy.spatial.transform import Rotation


def make(n_frames=20, pts_per_frame=150, seed=0, pix_noise=0.5, outlier_frac=0.05, ba=None):
    rng = np.random.default_rng(seed)
    Rcl = Rotation.from_rotvec(np.array([1.2, -1.2, 1.2]) + rng.normal(0, 0.02, 3)).as_matrix()   # lidar x-forward -> camera z-forward
    tcl = np.array([0.05, -0.08, -0.27])
    s = 9.5
    Tcl = np.eye(4)
    Tcl[:3, :3], Tcl[:3, 3] = Rcl, tcl
    fx, fy, cx, cy = 718.856, 718.856, 607.1928, 185.2157
    Twl = [np.eye(4)]
    for i in range(n_frames - 1):
        step = np.eye(4)
        step[:3, :3] = Rotation.from_rotvec(rng.normal(0, 0.03, 3)).as_matrix()
        step[:3, 3] = np.array([1.0, 0, 0]) + rng.normal(0, 0.05, 3)
        Twl.append(Twl[-1] @ step)
    Twl = np.array(Twl)
    frame_T6, frame_intr, ef, eX, eo, ei, es = [], [], [], [], [], [], []
    sig2 = [1.44 ** -k for k in range(8)]
    for f in range(n_frames):
        Tlw = np.linalg.inv(Twl[f])
        # e->Tlw_quat: the code stores rotation / translation of `Twl` under the name Tlw (Optimizer.cc:1627-1632); a caller
        # that wants the edge to mean "world-lidar -> lidar_f" passes vTwl = inverse poses. The planted scene does that.
        frame_T6.append(np.concatenate([Rotation.from_matrix(Tlw[:3, :3]).as_rotvec(), Tlw[:3, 3]]))
        frame_intr.append([fx, fy, cx, cy])
        # points in front of camera f, expressed in lidar_0 = world-lidar coordinates
        pc = np.stack([rng.uniform(-8, 8, pts_per_frame), rng.uniform(-2, 2, pts_per_frame), rng.uniform(4, 40, pts_per_frame)], 1)
        pl_f = (pc - tcl) @ Rcl                      # camera_f -> lidar_f
        pw = pl_f @ Twl[f, :3, :3].T + Twl[f, :3, 3]  # -> lidar world
        Xc0_metric = pw @ Rcl.T + tcl               # -> camera-0 frame (metric)
        Xw = (Xc0_metric / s).astype(np.float32).astype(np.float64)   # the ORB map is scale-free, CV_32F
        u = fx * pc[:, 0] / pc[:, 2] + cx + rng.normal(0, pix_noise, pts_per_frame)
        v = fy * pc[:, 1] / pc[:, 2] + cy + rng.normal(0, pix_noise, pts_per_frame)
        out = rng.random(pts_per_frame) < outlier_frac
        u[out] += rng.normal(0, 40, out.sum())
        v[out] += rng.normal(0, 40, out.sum())
        octave = rng.integers(0, 8, pts_per_frame)
        for j in range(pts_per_frame):
            ef.append(f); eX.append(Xw[j]); eo.append([np.float32(u[j]), np.float32(v[j])]); ei.append(np.float32(sig2[octave[j]])); es.append(j)
    x_gt = np.concatenate([Rotation.from_matrix(Rcl).as_rotvec(), tcl, [s]])
    prob = ba.BaProblem(np.array(frame_T6), np.array(frame_intr), np.array(ef), np.array(eX), np.array(eo, np.float64), np.array(ei, np.float64), np.array(es))
    return prob, x_gt
