"""Host-side mirror of the reference's loaders for the IBA path: thin ctypes bindings of the iba_dataset_* / iba_read_* /
iba_write_* entry points (include/iba_mi355x.h, csrc/iba_io.cpp). Plumbing only: files in, numpy arrays out.

Reference counterparts: readPointCloud (io_tools.h:142-196), ReadPoseList / readSim3 / writeSim3 (kitti_tools.h:66-158),
System::RestoreSystemFromFile (System.cc:612-694) and the set-up part of main() (iba_global.cpp:398-505,
iba_local.cpp:325-406)."""
import ctypes as C

import numpy as np

from . import IbaError, load_library
from .abi import _FIELDS, IbaProblemDesc, Problem


class IbaDatasetPaths(C.Structure):
    _fields_ = [("frame_id_file", C.c_char_p), ("lidar_pose_file", C.c_char_p), ("pointcloud_dir", C.c_char_p), ("keyframe_dir", C.c_char_p),
                ("map_file", C.c_char_p), ("pointcloud_skip", C.c_int32), ("only_positive_x", C.c_int32), ("num_best_covis", C.c_int32),
                ("min_covis_weight", C.c_int32)]


def _lib():
    L = load_library()
    if not getattr(L, "_io_ready", False):
        L.iba_io_last_error.restype = C.c_char_p
        L.iba_dataset_load.argtypes = [C.POINTER(IbaDatasetPaths), C.POINTER(C.c_void_p)]
        L.iba_dataset_desc.restype = C.POINTER(IbaProblemDesc)
        L.iba_dataset_desc.argtypes = [C.c_void_p]
        L.iba_dataset_free.argtypes = [C.c_void_p]
        L.iba_dataset_frame_ids.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.iba_read_kitti_bin.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_int64)]
        L.iba_read_pose_list.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_double)), C.POINTER(C.c_int64)]
        L.iba_read_sim3.argtypes = [C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.iba_write_sim3.argtypes = [C.c_char_p, C.POINTER(C.c_double), C.c_double]
        L.iba_io_free.argtypes = [C.c_void_p]
        L.iba_sim3_to_x.argtypes = [C.POINTER(C.c_double), C.c_double, C.POINTER(C.c_double)]
        L.iba_x_to_sim3.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.iba_pose_to_motion.argtypes = [C.POINTER(C.c_double), C.c_int64, C.POINTER(C.c_double)]
        L.iba_handeye.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int64, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.iba_handeye_robust.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int64, C.POINTER(C.c_double), C.c_double, C.c_double, C.c_int32,
                                         C.c_double, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.iba_handeye_lineprocess.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int64, C.POINTER(C.c_double), C.c_double, C.c_int32,
                                              C.c_double, C.c_double, C.c_double, C.c_int32, C.c_int32, C.c_double, C.POINTER(C.c_double),
                                              C.POINTER(C.c_double)]
        L._io_ready = True
    return L


def _chk(L, st):
    if st != 0:
        raise IbaError(st, (L.iba_io_last_error() or b"").decode())


def read_kitti_bin(path, skip=1, only_positive_x=False):
    L = _lib()
    p = C.POINTER(C.c_float)()
    n = C.c_int64(0)
    _chk(L, L.iba_read_kitti_bin(str(path).encode(), skip, 1 if only_positive_x else 0, C.byref(p), C.byref(n)))
    out = np.ctypeslib.as_array(p, shape=(max(n.value, 1) * 3,))[: 3 * n.value].copy().reshape(-1, 3)
    L.iba_io_free(p)
    return out


def read_pose_list(path):
    L = _lib()
    p = C.POINTER(C.c_double)()
    n = C.c_int64(0)
    _chk(L, L.iba_read_pose_list(str(path).encode(), C.byref(p), C.byref(n)))
    out = np.ctypeslib.as_array(p, shape=(max(n.value, 1) * 12,))[: 12 * n.value].copy().reshape(-1, 3, 4)
    L.iba_io_free(p)
    return out


def read_sim3(path):
    L = _lib()
    r = (C.c_double * 12)()
    s = C.c_double(0)
    _chk(L, L.iba_read_sim3(str(path).encode(), r, C.byref(s)))
    return np.array(r[:]).reshape(3, 4), s.value


def write_sim3(path, rigid, scale):
    L = _lib()
    r = (C.c_double * 12)(*np.asarray(rigid, np.float64)[:3, :4].reshape(-1))
    _chk(L, L.iba_write_sim3(str(path).encode(), r, float(scale)))


def sim3_to_x(rigid, scale):
    L = _lib()
    r = (C.c_double * 12)(*np.asarray(rigid, np.float64)[:3, :4].reshape(-1))
    x = (C.c_double * 7)()
    _chk(L, L.iba_sim3_to_x(r, float(scale), x))
    return np.array(x[:])


def x_to_sim3(x):
    L = _lib()
    xv = (C.c_double * 7)(*np.asarray(x, np.float64).reshape(7))
    r = (C.c_double * 12)()
    s = C.c_double(0)
    _chk(L, L.iba_x_to_sim3(xv, r, C.byref(s)))
    return np.array(r[:]).reshape(3, 4), s.value


_COUNTS = {  # field -> (count as a function of F, N, K, S, M)
    "pt_offset": lambda F, N, K, S, M: F + 1, "pts_xyz": lambda F, N, K, S, M: 3 * N, "intrinsics": lambda F, N, K, S, M: 6 * F,
    "kp_offset": lambda F, N, K, S, M: F + 1, "kp_uv": lambda F, N, K, S, M: 2 * K, "kp_has_mappoint": lambda F, N, K, S, M: K,
    "kp_mappoint_w": lambda F, N, K, S, M: 3 * K, "Tcw": lambda F, N, K, S, M: 12 * F, "covis_offset": lambda F, N, K, S, M: F + 1,
    "covis_frame": lambda F, N, K, S, M: S, "covis_relpose": lambda F, N, K, S, M: 12 * S, "match_offset": lambda F, N, K, S, M: S + 1,
    "match_kp_ref": lambda F, N, K, S, M: M, "match_kp_covis": lambda F, N, K, S, M: M, "Tc_next": lambda F, N, K, S, M: 12 * F,
    "Tl_next": lambda F, N, K, S, M: 12 * F,
}


def _view(ptr, dtype, count):
    if count == 0:
        return np.zeros(0, dtype)
    buf = (C.c_char * (count * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=count).copy()


def load_dataset(frame_id_file, lidar_pose_file, pointcloud_dir, keyframe_dir, map_file, pointcloud_skip=1, only_positive_x=False,
                 num_best_covis=3, min_covis_weight=100):
    """Packs a dataset directory of the reference pipeline into a Problem (arrays copied out of the C++ object).
    Returns (problem, mn_id, mn_frame_id)."""
    L = _lib()
    paths = IbaDatasetPaths(str(frame_id_file).encode(), str(lidar_pose_file).encode(), str(pointcloud_dir).encode(), str(keyframe_dir).encode(),
                            str(map_file).encode(), int(pointcloud_skip), 1 if only_positive_x else 0, int(num_best_covis), int(min_covis_weight))
    h = C.c_void_p()
    _chk(L, L.iba_dataset_load(C.byref(paths), C.byref(h)))
    try:
        d = L.iba_dataset_desc(h).contents
        F = d.n_frames
        off = lambda name: _view(getattr(d, name), np.uint64, F + 1)
        N, K, S = int(off("pt_offset")[-1]), int(off("kp_offset")[-1]), int(off("covis_offset")[-1])
        M = int(_view(d.match_offset, np.uint64, S + 1)[-1])
        arrays = {name: _view(getattr(d, name), dt, _COUNTS[name](F, N, K, S, M)) for name, dt in _FIELDS.items()}
        ids = np.zeros((F, 2), np.int32)
        for f in range(F):
            a, b = C.c_int32(0), C.c_int32(0)
            _chk(L, L.iba_dataset_frame_ids(h, f, C.byref(a), C.byref(b)))
            ids[f] = (a.value, b.value)
    finally:
        L.iba_dataset_free(h)
    return Problem(**arrays), ids[:, 0].copy(), ids[:, 1].copy()


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def pose_to_motion(poses):
    """pose2Motion (kitti_tools.h:160-165): T(i+1) * T(i)^-1 for poses given as (n, 3|4, 4)."""
    L = _lib()
    P = np.ascontiguousarray(np.asarray(poses, np.float64)[:, :3, :4]).reshape(-1, 12)
    M = np.zeros((len(P) - 1, 12))
    st = L.iba_pose_to_motion(_dp(P), len(P), _dp(M))
    if st != 0:
        raise IbaError(st, "iba_pose_to_motion")
    return M.reshape(-1, 3, 4)


def handeye(Ta, Tb):
    """HECalib (HECalib.h:12-57): returns (rigid 3x4 = T_AB, scale)."""
    L = _lib()
    A = np.ascontiguousarray(np.asarray(Ta, np.float64)[:, :3, :4]).reshape(-1, 12)
    B = np.ascontiguousarray(np.asarray(Tb, np.float64)[:, :3, :4]).reshape(-1, 12)
    r = np.zeros(12)
    s = C.c_double(0)
    st = L.iba_handeye(_dp(A), _dp(B), len(A), _dp(r), C.byref(s))
    if st != 0:
        raise IbaError(st, "iba_handeye")
    return r.reshape(3, 4), s.value


def handeye_degenerate(Ta, Tb, dg_threshold=0.01):
    """DGHECalib (HECalib.h:66-120): returns (rigid 3x4 with zero translation, scale, number of degenerate pairs)."""
    L = _lib()
    A = np.ascontiguousarray(np.asarray(Ta, np.float64)[:, :3, :4]).reshape(-1, 12)
    B = np.ascontiguousarray(np.asarray(Tb, np.float64)[:, :3, :4]).reshape(-1, 12)
    r = np.zeros(12)
    s = C.c_double(0)
    nd = C.c_int64(0)
    L.iba_handeye_degenerate.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int64, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    st = L.iba_handeye_degenerate(_dp(A), _dp(B), len(A), float(dg_threshold), _dp(r), C.byref(s), C.byref(nd))
    if st != 0:
        raise IbaError(st, "iba_handeye_degenerate")
    return r.reshape(3, 4), s.value, nd.value


def handeye_robust(Ta, Tb, rigid0, scale0, robust_kernel_size=0.1, regulation=True, regulation_ratio=0.005, iterations=10):
    """HECalibRobustKernelg2o (NLHECalib.hpp:121-163) on the device-independent host path."""
    L = _lib()
    A = np.ascontiguousarray(np.asarray(Ta, np.float64)[:, :3, :4]).reshape(-1, 12)
    B = np.ascontiguousarray(np.asarray(Tb, np.float64)[:, :3, :4]).reshape(-1, 12)
    r0 = np.ascontiguousarray(np.asarray(rigid0, np.float64)[:3, :4]).reshape(12)
    r = np.zeros(12)
    s = C.c_double(0)
    st = L.iba_handeye_robust(_dp(A), _dp(B), len(A), _dp(r0), float(scale0), float(robust_kernel_size), 1 if regulation else 0, float(regulation_ratio),
                              int(iterations), _dp(r), C.byref(s))
    if st != 0:
        raise IbaError(st, "iba_handeye_robust")
    return r.reshape(3, 4), s.value


def handeye_lineprocess(Ta, Tb, rigid0, scale0, inner_iterations=10, mu0=64.0, divid_factor=1.4, min_mu=1e-1, ex_max_iter=20,
                        regulation=True, regulation_ratio=0.005):
    """HECalibLineProcessg2o (NLHECalib.hpp:189-277); inner_iterations=10 is what the reference runs whatever its
    in_max_iter argument says."""
    L = _lib()
    A = np.ascontiguousarray(np.asarray(Ta, np.float64)[:, :3, :4]).reshape(-1, 12)
    B = np.ascontiguousarray(np.asarray(Tb, np.float64)[:, :3, :4]).reshape(-1, 12)
    r0 = np.ascontiguousarray(np.asarray(rigid0, np.float64)[:3, :4]).reshape(12)
    r = np.zeros(12)
    s = C.c_double(0)
    st = L.iba_handeye_lineprocess(_dp(A), _dp(B), len(A), _dp(r0), float(scale0), int(inner_iterations), float(mu0), float(divid_factor),
                                   float(min_mu), int(ex_max_iter), 1 if regulation else 0, float(regulation_ratio), _dp(r), C.byref(s))
    if st != 0:
        raise IbaError(st, "iba_handeye_lineprocess")
    return r.reshape(3, 4), s.value


def read_cv_yaml_numbers(path, key):
    """The numbers of one top-level entry of a cv::FileStorage YAML file through the reader iba_dataset_load uses."""
    L = _lib()
    L.iba_read_cv_yaml_numbers.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_double), C.c_int32, C.POINTER(C.c_int32)]
    n = C.c_int32(0)
    _chk(L, L.iba_read_cv_yaml_numbers(str(path).encode(), key.encode(), None, 0, C.byref(n)))
    out = np.zeros(max(n.value, 1))
    _chk(L, L.iba_read_cv_yaml_numbers(str(path).encode(), key.encode(), _dp(out), n.value, C.byref(n)))
    return out[: n.value]


class RunConfig:
    """The reference's run configuration file (config/calib/NN/iba_calib_global.yml and its iba_func / iba_local siblings) as the
    C-ABI's structs: iba_run_config_* (csrc/iba_config.cpp; what main() reads with yaml-cpp, iba_global.cpp:412-471)."""

    def __init__(self, path):
        from .abi import IbaMadsOptions, IbaParams
        self._P, self._M = IbaParams, IbaMadsOptions
        self.L = L = load_library()
        L.iba_run_config_last_error.restype = C.c_char_p
        L.iba_run_config_load.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        L.iba_run_config_free.argtypes = [C.c_void_p]
        L.iba_run_config_get.restype = C.c_char_p
        L.iba_run_config_get.argtypes = [C.c_void_p, C.c_char_p]
        L.iba_run_config_path.restype = C.c_char_p
        L.iba_run_config_path.argtypes = [C.c_void_p, C.c_char_p]
        L.iba_run_config_params.argtypes = [C.c_void_p, C.c_int32, C.POINTER(IbaParams)]
        L.iba_run_config_paths.argtypes = [C.c_void_p, C.c_int32, C.POINTER(IbaDatasetPaths)]
        L.iba_run_config_mads.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(IbaMadsOptions)]
        self.h = C.c_void_p()
        self._chk(L.iba_run_config_load(str(path).encode(), C.byref(self.h)))

    def _chk(self, st):
        if st != 0:
            raise IbaError(st, (self.L.iba_run_config_last_error() or b"").decode())

    def close(self):
        if self.h:
            self.L.iba_run_config_free(self.h)
            self.h = C.c_void_p()

    def get(self, dotted_key):
        v = self.L.iba_run_config_get(self.h, dotted_key.encode())
        return None if v is None else v.decode()

    def path(self, io_key):
        v = self.L.iba_run_config_path(self.h, io_key.encode())
        return None if v is None else v.decode()

    def params(self, local_stage=False):
        p = self._P()
        self._chk(self.L.iba_run_config_params(self.h, 1 if local_stage else 0, C.byref(p)))
        return p

    def paths(self, local_stage=False):
        d = IbaDatasetPaths()
        self._chk(self.L.iba_run_config_paths(self.h, 1 if local_stage else 0, C.byref(d)))
        return {k: (getattr(d, k).decode() if isinstance(getattr(d, k), bytes) else getattr(d, k)) for k, _ in IbaDatasetPaths._fields_}

    def mads(self, x0):
        x0 = np.ascontiguousarray(x0, np.float64)
        o = self._M()
        self._chk(self.L.iba_run_config_mads(self.h, _dp(x0), C.byref(o)))
        return o
