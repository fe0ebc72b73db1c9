"""ctypes mirror of include/iba_mi355x.h (the C-ABI drop-in boundary) + the flat problem container.

Plumbing only: numpy arrays in, POD structs across the boundary. No compute happens here.
"""
import ctypes as C

import numpy as np

IBA_MAX_BATCH = 64
IBA_MAX_CHAIN = 512


class IbaProblemDesc(C.Structure):
    _fields_ = [
        ("n_frames", C.c_int32),
        ("pt_offset", C.c_void_p),
        ("pts_xyz", C.c_void_p),
        ("intrinsics", C.c_void_p),
        ("kp_offset", C.c_void_p),
        ("kp_uv", C.c_void_p),
        ("kp_has_mappoint", C.c_void_p),
        ("kp_mappoint_w", C.c_void_p),
        ("Tcw", C.c_void_p),
        ("covis_offset", C.c_void_p),
        ("covis_frame", C.c_void_p),
        ("covis_relpose", C.c_void_p),
        ("match_offset", C.c_void_p),
        ("match_kp_ref", C.c_void_p),
        ("match_kp_covis", C.c_void_p),
        ("Tc_next", C.c_void_p),
        ("Tl_next", C.c_void_p),
    ]


class IbaParams(C.Structure):
    _fields_ = [
        ("max_pixel_dist", C.c_double),
        ("num_min_corr_cost", C.c_int32),
        ("corr_3d_2d_threshold", C.c_double),
        ("corr_3d_3d_threshold", C.c_double),
        ("norm_max_pts", C.c_int32),
        ("norm_min_pts", C.c_int32),
        ("norm_radius", C.c_double),
        ("norm_reg_threshold", C.c_double),
        ("min_diff_dist", C.c_double),
        ("err_weight", C.c_double * 2),
        ("use_plane", C.c_int32),
        ("num_min_corr", C.c_int32),
        ("max_3d_dist", C.c_double),
        ("neigh_radius", C.c_double),
        ("neigh_max_pts", C.c_int32),
        ("neigh_min_pts", C.c_int32),
        ("local_min_diff_dist", C.c_double),
        ("local_norm_reg_threshold", C.c_double),
        ("robust_kernel_delta", C.c_double),
        ("robust_kernel_3ddelta", C.c_double),
        ("plane_cache", C.c_int32),
        ("factor_3d2d_kind", C.c_int32),
    ]


class IbaCreateOptions(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("common_pairs", C.c_int32), ("common_max_px", C.c_double), ("max_pair_groups", C.c_int32), ("pair_memo", C.c_int32),
                ("pair_memo_max_batch", C.c_int32), ("pair_inflation", C.c_double), ("anchored_lists", C.c_int32), ("anchor_reach", C.c_double), ("side_stream", C.c_int32),
                ("spin_wait", C.c_int32), ("factor_mfma", C.c_int32), ("pair_list_capacity", C.c_int32), ("max_chain_batch", C.c_int32), ("chain_fold", C.c_int32)]


class IbaCostOut(C.Structure):
    _fields_ = [
        ("f1", C.c_double),
        ("f2", C.c_double),
        ("C", C.c_double),
        ("valid_cnt_3d_2d", C.c_int32),
        ("cnt_3d_2d", C.c_int32),
        ("cnt_3d_3d", C.c_int32),
        ("valid_cnt_3d_3d", C.c_int32),
        ("valid_pl_3d_3d", C.c_int32),
        ("valid_pt_3d_3d", C.c_int32),
        ("frames_used", C.c_int32),
        ("n_corr", C.c_int32),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class IbaNormalOut(C.Structure):
    _fields_ = [
        ("H", C.c_double * 49),
        ("b", C.c_double * 7),
        ("cost", C.c_double),
        ("chi2", C.c_double),
        ("n_factor_3d2d", C.c_int32),
        ("n_factor_p2pl", C.c_int32),
        ("n_factor_p2pt", C.c_int32),
        ("n_residuals", C.c_int32),
        ("frames_used", C.c_int32),
        ("n_corr", C.c_int32),
    ]

    def H_np(self):
        return np.array(self.H[:], dtype=np.float64).reshape(7, 7)

    def b_np(self):
        return np.array(self.b[:], dtype=np.float64)

    def counts(self):
        return {k: getattr(self, k) for k in ("n_factor_3d2d", "n_factor_p2pl", "n_factor_p2pt", "n_residuals", "frames_used", "n_corr")}


class IbaLmOptions(C.Structure):
    _fields_ = [("max_outer_iterations", C.c_int32), ("max_inner_iterations", C.c_int32), ("min_diff", C.c_double),
                ("function_tolerance", C.c_double), ("gradient_tolerance", C.c_double), ("parameter_tolerance", C.c_double),
                ("initial_trust_region_radius", C.c_double)]


class IbaLmResult(C.Structure):
    _fields_ = [("x", C.c_double * 7), ("outer_iterations", C.c_int32), ("inner_iterations", C.c_int32), ("evaluations", C.c_int32),
                ("converged", C.c_int32), ("initial_cost", C.c_double), ("final_cost", C.c_double)]


class IbaMadsOptions(C.Structure):
    _fields_ = [("max_bb_eval", C.c_int32), ("lb", C.c_double * 7), ("ub", C.c_double * 7), ("init_frame", C.c_double * 7), ("min_mesh", C.c_double),
                ("he_threshold", C.c_double), ("valid_rate", C.c_double), ("seed", C.c_int32), ("bases_per_poll", C.c_int32), ("speculative", C.c_int32), ("vns_max_idle", C.c_int32)]


class IbaMadsResult(C.Structure):
    _fields_ = [("x", C.c_double * 7), ("f", C.c_double), ("c1", C.c_double), ("c2", C.c_double), ("c3", C.c_double), ("feasible", C.c_int32),
                ("evaluations", C.c_int32), ("iterations", C.c_int32), ("batches", C.c_int32), ("cache_hits", C.c_int32), ("restarts", C.c_int32), ("stop_reason", C.c_int32)]


class IbaBbo(C.Structure):
    _fields_ = [("f", C.c_double), ("c1", C.c_double), ("c2", C.c_double), ("c3", C.c_double)]


def reference_yaml_params(plane_cache=1):
    """config/calib/00/iba_calib_global.yml:21-35 on top of the struct defaults
    (IBAGlobalParams iba_global.cpp:26-52, IBALocalParams IBACalib2.hpp:108-137)."""
    p = IbaParams()
    p.max_pixel_dist = 1.5
    p.num_min_corr_cost = 30
    p.corr_3d_2d_threshold = 40.0
    p.corr_3d_3d_threshold = 10.0
    p.norm_max_pts = 30
    p.norm_min_pts = 5
    p.norm_radius = 0.6
    p.norm_reg_threshold = 0.02
    p.min_diff_dist = 0.2
    p.err_weight[0] = 1.0
    p.err_weight[1] = 1.0
    p.use_plane = 1
    p.num_min_corr = 30
    p.max_3d_dist = 1.0
    p.neigh_radius = 0.6
    p.neigh_max_pts = 30
    p.neigh_min_pts = 5
    p.local_min_diff_dist = 0.2
    p.local_norm_reg_threshold = 0.02
    p.robust_kernel_delta = 2.98
    p.robust_kernel_3ddelta = 1.0
    p.plane_cache = plane_cache
    return p


def copy_params(p):
    q = IbaParams()
    C.memmove(C.byref(q), C.byref(p), C.sizeof(IbaParams))
    return q


_FIELDS = {
    "pt_offset": np.uint64,
    "pts_xyz": np.float32,
    "intrinsics": np.float64,
    "kp_offset": np.uint64,
    "kp_uv": np.float32,
    "kp_has_mappoint": np.uint8,
    "kp_mappoint_w": np.float32,
    "Tcw": np.float32,
    "covis_offset": np.uint64,
    "covis_frame": np.int32,
    "covis_relpose": np.float32,
    "match_offset": np.uint64,
    "match_kp_ref": np.int32,
    "match_kp_covis": np.int32,
    "Tc_next": np.float32,
    "Tl_next": np.float64,
}


class Problem:
    """Flat arrays of iba_problem_desc (see the header for the reference object each one replaces)."""

    def __init__(self, **arrays):
        self.arrays = {}
        for name, dt in _FIELDS.items():
            a = np.ascontiguousarray(arrays[name], dtype=dt)
            self.arrays[name] = a
        self.n_frames = int(len(self.arrays["pt_offset"]) - 1)
        self._check()

    def _check(self):
        a = self.arrays
        F = self.n_frames
        assert len(a["kp_offset"]) == F + 1 and len(a["covis_offset"]) == F + 1
        N = int(a["pt_offset"][-1])
        K = int(a["kp_offset"][-1])
        S = int(a["covis_offset"][-1])
        assert a["pts_xyz"].size == 3 * N and a["kp_uv"].size == 2 * K
        assert a["kp_has_mappoint"].size == K and a["kp_mappoint_w"].size == 3 * K
        assert a["intrinsics"].size == 6 * F and a["Tcw"].size == 12 * F
        assert a["covis_frame"].size == S and a["covis_relpose"].size == 12 * S and len(a["match_offset"]) == S + 1
        M = int(a["match_offset"][-1])
        assert a["match_kp_ref"].size == M and a["match_kp_covis"].size == M
        assert a["Tc_next"].size == 12 * F and a["Tl_next"].size == 12 * F

    @property
    def n_points(self):
        return int(self.arrays["pt_offset"][-1])

    @property
    def n_keypoints(self):
        return int(self.arrays["kp_offset"][-1])

    def desc(self):
        d = IbaProblemDesc()
        d.n_frames = self.n_frames
        for name in _FIELDS:
            setattr(d, name, self.arrays[name].ctypes.data)
        return d

    def frame_points(self, f):
        o = self.arrays["pt_offset"]
        return self.arrays["pts_xyz"].reshape(-1, 3)[int(o[f]):int(o[f + 1])]

    def frame_keypoints(self, f):
        o = self.arrays["kp_offset"]
        return self.arrays["kp_uv"].reshape(-1, 2)[int(o[f]):int(o[f + 1])]

    def save(self, path):
        np.savez_compressed(path, **self.arrays)

    @staticmethod
    def load(path):
        z = np.load(path)
        return Problem(**{k: z[k] for k in _FIELDS})

    @staticmethod
    def from_scans(scans, intr=(718.856, 718.856, 607.1928, 185.2157, 1241.0, 376.0)):
        """A problem that holds scans only (no keypoint, no covisibility): what iba_geo_correspondences and the kd-search probes need.
        scans: list of [P_f, 3] float32 arrays."""
        F = len(scans)
        pt_off = np.concatenate([[0], np.cumsum([len(s) for s in scans])]).astype(np.uint64)
        eye34 = np.tile(np.eye(3, 4, dtype=np.float32).ravel(), F)
        return Problem(pt_offset=pt_off, pts_xyz=np.concatenate([np.asarray(s, np.float32).reshape(-1, 3) for s in scans]).ravel() if F else np.zeros(0, np.float32),
                       intrinsics=np.tile(np.asarray(intr, np.float64), F), kp_offset=np.zeros(F + 1, np.uint64), kp_uv=np.zeros(0, np.float32),
                       kp_has_mappoint=np.zeros(0, np.uint8), kp_mappoint_w=np.zeros(0, np.float32), Tcw=eye34, covis_offset=np.zeros(F + 1, np.uint64),
                       covis_frame=np.zeros(0, np.int32), covis_relpose=np.zeros(0, np.float32), match_offset=np.zeros(1, np.uint64),
                       match_kp_ref=np.zeros(0, np.int32), match_kp_covis=np.zeros(0, np.int32), Tc_next=eye34.copy(), Tl_next=np.tile(np.eye(3, 4).ravel(), F))
