"""Seeded synthetic "street canyon" datasets with a planted LiDAR->camera extrinsic.

The reference cannot be run offline (needs KITTI + ORB-SLAM2 + F-LOAM outputs), so bench.py and the
parity tests use scenes that reproduce the *shape* of its inputs: per-keyframe float32 scans,
~2000 float32 ORB keypoints per keyframe with KITTI-00 intrinsics
(config/orb_ori/KITTI00-02.yaml:8-11), scale-free monocular MapPoints and float32 keyframe poses,
covisibility lists with keypoint<->keypoint matches, LiDAR odometry poses.

Conventions: LiDAR frame x forward / y left / z up; camera frame x right / y down / z forward;
x = [omega, upsilon, s] with (R, t) = Sim3Exp(x) mapping LiDAR -> camera and s the metres-per-ORB-unit
scale (g2o_tools.h:105-140; iba_global.cpp:189-192, 208, 232).
"""
import numpy as np

from .abi import Problem

KITTI00 = dict(fx=718.856, fy=718.856, cx=607.1928, cy=185.2157, W=1241.0, H=376.0)


def _rodrigues(w):
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0.0]])
    if th < 1e-12:
        return np.eye(3) + K
    return np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th**2 * (K @ K)


def _V(w):
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0.0]])
    if th < 1e-8:
        return np.eye(3) + 0.5 * K + K @ K / 6
    return np.eye(3) + (1 - np.cos(th)) / th**2 * K + (th - np.sin(th)) / th**3 * (K @ K)


def _rotvec(R):
    c = np.clip((np.trace(R) - 1) / 2, -1, 1)
    th = np.arccos(c)
    v = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    if th < 1e-10:
        return 0.5 * v
    return th / (2 * np.sin(th)) * v


def sim3_log(R, t, s):
    w = _rotvec(R)
    return np.concatenate([w, np.linalg.solve(_V(w), t), [s]])


def sim3_exp(x):
    return _rodrigues(x[:3]), _V(x[:3]) @ x[3:6], x[6]


def _T(R, t):
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = t
    return T


def gt_extrinsic():
    axes = np.array([[0.0, -1, 0], [0, 0, -1], [1, 0, 0]])
    R = _rodrigues(np.array([0.010, -0.020, 0.015])) @ axes
    t = np.array([0.02, -0.08, -0.27])
    return R, t


def _raycast(o, D, planes_z, walls_y, boxes, max_range):
    """o (3,), D (n,3) unit dirs in world; returns hit distance (inf if none)."""
    n = D.shape[0]
    tbest = np.full(n, np.inf)
    with np.errstate(divide="ignore", invalid="ignore"):
        t = (planes_z - o[2]) / D[:, 2]
        t[~(t > 0.5)] = np.inf
        tbest = np.minimum(tbest, t)
        for wy in walls_y:
            t = (wy - o[1]) / D[:, 1]
            t[~(t > 0.5)] = np.inf
            tbest = np.minimum(tbest, t)
        for b in boxes:  # slabs; b = (lo3, hi3)
            lo, hi = b
            t1 = (lo[None, :] - o[None, :]) / D
            t2 = (hi[None, :] - o[None, :]) / D
            tn = np.nanmax(np.minimum(t1, t2), axis=1)
            tf = np.nanmin(np.maximum(t1, t2), axis=1)
            hit = (tn <= tf) & (tn > 0.5)
            t = np.where(hit, tn, np.inf)
            tbest = np.minimum(tbest, t)
    tbest[tbest > max_range] = np.inf
    return tbest


def make_scene(n_frames=50, pts_per_frame=4000, n_keypoints=2000, seed=0, n_covis=3, s_star=10.0, kp_noise=0.5,
               new_mappoints=250, scan_kp=300, range_noise=0.02, max_range=80.0, intr=KITTI00):
    """Returns (Problem, meta). meta: x_gt (7,), Twl (F,4,4), s_star."""
    rng = np.random.default_rng(seed)
    F, P, K = int(n_frames), int(pts_per_frame), int(n_keypoints)
    fx, fy, cx, cy, W, H = (intr[k] for k in ("fx", "fy", "cx", "cy", "W", "H"))
    Rcl, tcl = gt_extrinsic()
    Tcl = _T(Rcl, tcl)
    Tlc = np.linalg.inv(Tcl)
    x_gt = sim3_log(Rcl, tcl, s_star)

    # LiDAR trajectory (lidar f -> lidar-world); first pose is the identity (iba_global.cpp:479-484)
    yaw = np.concatenate([[0.0], rng.uniform(-0.05, 0.05, F - 1)])
    lat = np.concatenate([[0.0], np.cumsum(rng.normal(0, 0.02, F - 1))])
    Twl = np.zeros((F, 4, 4))
    for f in range(F):
        c, s = np.cos(yaw[f]), np.sin(yaw[f])
        Twl[f] = _T(np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]]), np.array([1.0 * f, lat[f], 0.0]))

    # world: ground, two walls, boxes on both sides of the route
    ground_z, walls_y = -1.73, (-8.0, 8.0)
    nb = max(4, int((F + 90) / 6))
    bx = np.arange(nb) * 6.0 + rng.uniform(-1.5, 1.5, nb) + 4.0
    by = rng.uniform(3.0, 7.0, nb) * rng.choice([-1.0, 1.0], nb)
    bh = rng.uniform(0.6, 2.5, nb)
    bs = rng.uniform(0.4, 1.5, (nb, 2))
    boxes_all = [(np.array([bx[i] - bs[i, 0], by[i] - bs[i, 1], ground_z]), np.array([bx[i] + bs[i, 0], by[i] + bs[i, 1], ground_z + bh[i]])) for i in range(nb)]

    # HDL-64-like pattern restricted to the forward sector
    n_az = int(np.ceil(P * 1.7 / 64))
    elev = np.deg2rad(np.linspace(-24.8, 2.0, 64))
    pts_all = np.zeros((F, P, 3), dtype=np.float32)
    for f in range(F):
        az = np.deg2rad(np.linspace(-80, 80, n_az) + rng.uniform(-0.05, 0.05))
        E, A = np.meshgrid(elev, az, indexing="ij")
        dl = np.stack([np.cos(E) * np.cos(A), np.cos(E) * np.sin(A), np.sin(E)], -1).reshape(-1, 3)
        o = Twl[f, :3, 3]
        Dw = dl @ Twl[f, :3, :3].T
        near = [b for b in boxes_all if abs(0.5 * (b[0][0] + b[1][0]) - o[0]) < max_range + 3]
        t = _raycast(o, Dw, ground_z, walls_y, near, max_range)
        ok = np.flatnonzero(np.isfinite(t))
        if ok.size < P:
            raise RuntimeError(f"frame {f}: only {ok.size} valid returns for {P} requested")
        sel = np.sort(rng.choice(ok, P, replace=False))
        r = t[sel] + rng.normal(0, range_noise, P)
        pts_all[f] = (dl[sel] * r[:, None]).astype(np.float32)

    def project(pc):
        z = pc[:, 2]
        with np.errstate(divide="ignore", invalid="ignore"):
            u = fx * pc[:, 0] / z + cx
            v = fy * pc[:, 1] / z + cy
        ok = (z > 0.3) & (u >= 2) & (u < W - 2) & (v >= 2) & (v < H - 2)
        return u, v, ok

    # MapPoints: created from scan points of frame f, observed by frames f..f+n_covis
    obs = [dict() for _ in range(F)]  # frame -> {mp_id: (u, v)}
    mp_world = []                      # lidar-world metric coordinates
    scan_only = []                     # per frame: (u, v) keypoints that are scan projections w/o MapPoint
    for f in range(F):
        pl = pts_all[f].astype(np.float64)
        pc = pl @ Rcl.T + tcl
        u, v, ok = project(pc)
        cand = np.flatnonzero(ok)
        n_new = min(new_mappoints, cand.size // 2)
        pick = rng.choice(cand, n_new + min(scan_kp, cand.size - n_new), replace=False)
        new_ids = pick[:n_new]
        so = pick[n_new:]
        scan_only.append(np.stack([u[so], v[so]], 1) + rng.normal(0, kp_noise, (so.size, 2)))
        Pw = pl[new_ids] @ Twl[f, :3, :3].T + Twl[f, :3, 3]
        base = len(mp_world)
        mp_world.extend(Pw)
        for g in range(f, min(F, f + n_covis + 1)):
            Tlw = np.linalg.inv(Twl[g])
            pcg = (Pw @ Tlw[:3, :3].T + Tlw[:3, 3]) @ Rcl.T + tcl
            ug, vg, okg = project(pcg)
            nz = rng.normal(0, kp_noise, (n_new, 2))
            ug, vg = ug + nz[:, 0], vg + nz[:, 1]
            og = obs[g]
            for j in np.flatnonzero(okg):
                og[base + int(j)] = (ug[j], vg[j])
    mp_world = np.array(mp_world).reshape(-1, 3)

    # assemble keypoints per frame
    kp_uv, kp_has, kp_mpw, kp_off = [], [], [], [0]
    mp2kp = [dict() for _ in range(F)]
    mp_cam0 = mp_world @ Rcl.T + tcl if len(mp_world) else np.zeros((0, 3))  # ORB world = camera 0
    for f in range(F):
        ids = list(obs[f].keys())
        a = np.array([obs[f][m] for m in ids]).reshape(-1, 2)
        b = scan_only[f]
        nA, nB = len(ids), len(b)
        if nA + nB > K:
            nB = max(0, K - nA)
            b = b[:nB]
            if nA > K:
                ids, a, nA = ids[:K], a[:K], K
        nC = K - nA - nB
        c = np.stack([rng.uniform(0, W - 1, nC), rng.uniform(0, H - 1, nC)], 1)
        uv = np.concatenate([a, b, c], 0)
        uv[:, 0] = np.clip(uv[:, 0], 0, W - 1.001)
        uv[:, 1] = np.clip(uv[:, 1], 0, H - 1.001)
        perm = rng.permutation(K)
        inv = np.empty(K, dtype=np.int64)
        inv[perm] = np.arange(K)
        uvp = uv[perm]
        has = np.zeros(K, np.uint8)
        mpw = np.zeros((K, 3), np.float32)
        for j, m in enumerate(ids):
            kpid = int(inv[j])
            mp2kp[f][m] = kpid
            has[kpid] = 1
            mpw[kpid] = (mp_cam0[m] / s_star).astype(np.float32)
        kp_uv.append(uvp.astype(np.float32))
        kp_has.append(has)
        kp_mpw.append(mpw)
        kp_off.append(kp_off[-1] + K)

    # keyframe poses in the ORB world (scale-free), float32 like cv::Mat CV_32F
    Tcw32 = np.zeros((F, 4, 4), np.float32)
    Twc32 = np.zeros((F, 4, 4), np.float32)
    for f in range(F):
        Twc = Tcl @ Twl[f] @ Tlc
        Twc[:3, 3] /= s_star
        Tcw32[f] = np.linalg.inv(Twc).astype(np.float32)
        Twc32[f] = np.linalg.inv(Tcw32[f].astype(np.float64)).astype(np.float32)  # GetPoseInverse(): float Twc

    covis_off, covis_frame, covis_rel, match_off, m_ref, m_cov = [0], [], [], [0], [], []
    for f in range(F):
        nxt = [g for g in range(f + 1, min(F, f + n_covis + 1))]
        prv = [g for g in range(f - 1, -1, -1)][: max(0, n_covis - len(nxt))]
        for g in nxt + prv:
            covis_frame.append(g)
            covis_rel.append((Tcw32[g] @ Twc32[f])[:3, :].reshape(12))  # float32 product (iba_global.cpp:280)
            shared = [m for m in mp2kp[f] if m in mp2kp[g]]
            for m in shared:
                m_ref.append(mp2kp[f][m])
                m_cov.append(mp2kp[g][m])
            match_off.append(match_off[-1] + len(shared))
        covis_off.append(len(covis_frame))

    Tc_next = np.zeros((F, 12), np.float32)
    Tl_next = np.zeros((F, 12), np.float64)
    for f in range(F):
        if f < F - 1:
            Tc_next[f] = (Tcw32[f + 1] @ Twc32[f])[:3, :].reshape(12)
            Tl_next[f] = (np.linalg.inv(Twl[f + 1]) @ Twl[f])[:3, :].reshape(12)
        else:
            Tc_next[f] = np.eye(4, dtype=np.float32)[:3, :].reshape(12)
            Tl_next[f] = np.eye(4)[:3, :].reshape(12)

    prob = Problem(
        pt_offset=np.arange(F + 1, dtype=np.uint64) * P,
        pts_xyz=pts_all.reshape(-1),
        intrinsics=np.tile(np.array([fx, fy, cx, cy, W, H]), F),
        kp_offset=np.array(kp_off, np.uint64),
        kp_uv=np.concatenate(kp_uv).reshape(-1),
        kp_has_mappoint=np.concatenate(kp_has),
        kp_mappoint_w=np.concatenate(kp_mpw).reshape(-1),
        Tcw=Tcw32[:, :3, :].reshape(-1),
        covis_offset=np.array(covis_off, np.uint64),
        covis_frame=np.array(covis_frame, np.int32),
        covis_relpose=np.array(covis_rel, np.float32).reshape(-1),
        match_offset=np.array(match_off, np.uint64),
        match_kp_ref=np.array(m_ref, np.int32),
        match_kp_covis=np.array(m_cov, np.int32),
        Tc_next=Tc_next.reshape(-1),
        Tl_next=Tl_next.reshape(-1),
    )
    # identities behind the flat arrays (what the on-disk KeyFrame / Map files of the reference carry)
    mp_orb_f32 = (mp_cam0 / s_star).astype(np.float32) if len(mp_cam0) else np.zeros((0, 3), np.float32)
    meta = dict(x_gt=x_gt, Twl=Twl, s_star=s_star, seed=seed, mp2kp=mp2kp, mp_orb_f32=mp_orb_f32)
    return prob, meta


def perturb(x, rng, rot=5e-4, trans=5e-3, scale_rel=1e-3, n=1):
    """n seeded perturbations of x (n,7): small enough that the planted matches survive."""
    x = np.asarray(x, dtype=np.float64)
    d = np.concatenate([rng.normal(0, rot, (n, 3)), rng.normal(0, trans, (n, 3)), rng.normal(0, scale_rel, (n, 1)) * x[6]], 1)
    return x[None, :] + d


def tile_scene(prob, meta, times):
    """Concatenate `times` independent copies of a scene along the frame axis (used for the large
    configs: generation cost stays that of one base scene). Covisibility and HE pairs stay inside
    each copy except that the last frame of a copy pairs with identity transforms."""
    a = prob.arrays
    F = prob.n_frames
    out = {}
    N, K, S, M = int(a["pt_offset"][-1]), int(a["kp_offset"][-1]), int(a["covis_offset"][-1]), int(a["match_offset"][-1])

    def cat_off(name, tot):
        parts = [a[name][:-1] + np.uint64(i * tot) for i in range(times)]
        return np.concatenate(parts + [np.array([times * tot], np.uint64)])

    out["pt_offset"] = cat_off("pt_offset", N)
    out["kp_offset"] = cat_off("kp_offset", K)
    out["covis_offset"] = cat_off("covis_offset", S)
    out["match_offset"] = cat_off("match_offset", M)
    for name in ("pts_xyz", "intrinsics", "kp_uv", "kp_has_mappoint", "kp_mappoint_w", "Tcw", "covis_relpose", "match_kp_ref", "match_kp_covis", "Tc_next", "Tl_next"):
        out[name] = np.tile(a[name], times)
    out["covis_frame"] = np.concatenate([a["covis_frame"] + np.int32(i * F) for i in range(times)])
    return Problem(**out), dict(meta)
