"""ctypes mirror of the ORB-only extrinsic bundle adjustment entry points (include/iba_mi355x.h, csrc/iba_ba.hip):
calibEdge / OptimizeExtrinsicGlobal|Local of the reference (Optimizer.cc:65-205, 1399-1744). Plumbing only."""
import ctypes as C

import numpy as np

from . import IbaError, load_library


class IbaBaDesc(C.Structure):
    _fields_ = [("n_edges", C.c_int64), ("n_frames", C.c_int32), ("frame_Tlw6", C.c_void_p), ("frame_intr", C.c_void_p), ("edge_frame", C.c_void_p),
                ("edge_Xw", C.c_void_p), ("edge_obs", C.c_void_p), ("edge_info", C.c_void_p), ("edge_slot", C.c_void_p)]


class IbaBaResult(C.Structure):
    _fields_ = [("x", C.c_double * 7), ("n_inliers", C.c_int32), ("n_edges", C.c_int32), ("lm_iterations", C.c_int32), ("evaluations", C.c_int32),
                ("chi2", C.c_double * 4), ("n_bad", C.c_int32 * 4)]


class BaProblem:
    """Flat edge list. frame_Tlw6 (F,6), frame_intr (F,4), edge_frame (N,), edge_Xw (N,3), edge_obs (N,2), edge_info (N,), edge_slot (N,)."""

    def __init__(self, frame_Tlw6, frame_intr, edge_frame, edge_Xw, edge_obs, edge_info, edge_slot=None):
        self.frame_Tlw6 = np.ascontiguousarray(frame_Tlw6, np.float64).reshape(-1, 6)
        self.frame_intr = np.ascontiguousarray(frame_intr, np.float64).reshape(-1, 4)
        self.edge_frame = np.ascontiguousarray(edge_frame, np.int32)
        N = len(self.edge_frame)
        self.edge_Xw = np.ascontiguousarray(edge_Xw, np.float64).reshape(N, 3)
        self.edge_obs = np.ascontiguousarray(edge_obs, np.float64).reshape(N, 2)
        self.edge_info = np.ascontiguousarray(edge_info, np.float64).reshape(N)
        self.edge_slot = np.ascontiguousarray(np.arange(N) if edge_slot is None else edge_slot, np.int32)

    def desc(self):
        d = IbaBaDesc()
        d.n_edges, d.n_frames = len(self.edge_frame), len(self.frame_Tlw6)
        for k in ("frame_Tlw6", "frame_intr", "edge_frame", "edge_Xw", "edge_obs", "edge_info", "edge_slot"):
            setattr(d, k, getattr(self, k).ctypes.data)
        return d


class BaHandle:
    def __init__(self, problem, device=0):
        self.lib = load_library()
        self.lib.iba_ba_last_error.restype = C.c_char_p
        self.lib.iba_ba_last_error.argtypes = [C.c_void_p]
        self.lib.iba_ba_destroy.argtypes = [C.c_void_p]
        self.problem = problem
        self.h = C.c_void_p()
        d = problem.desc()
        st = self.lib.iba_ba_create(C.byref(d), C.c_int(device), C.byref(self.h))
        if st != 0:
            raise IbaError(st, self.lib.iba_ba_last_error(None).decode())

    def _chk(self, st):
        if st != 0:
            raise IbaError(st, self.lib.iba_ba_last_error(self.h).decode())

    def close(self):
        if self.h:
            self.lib.iba_ba_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def eval(self, x, active=None, robust=True, want_chi2=True):
        x = np.ascontiguousarray(x, np.float64)
        N = len(self.problem.edge_frame)
        H, b, chi = np.zeros(49), np.zeros(7), C.c_double(0)
        chi2 = np.zeros(max(N, 1)) if want_chi2 else None
        act = None if active is None else np.ascontiguousarray(active, np.uint8)
        self._chk(self.lib.iba_ba_eval(self.h, x.ctypes.data_as(C.c_void_p), None if act is None else act.ctypes.data_as(C.c_void_p), C.c_int32(1 if robust else 0),
                                       H.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), C.byref(chi), None if chi2 is None else chi2.ctypes.data_as(C.c_void_p)))
        return H.reshape(7, 7), b, chi.value, (None if chi2 is None else chi2[:N])

    def optimize(self, x0):
        x0 = np.ascontiguousarray(x0, np.float64)
        r = IbaBaResult()
        self._chk(self.lib.iba_ba_optimize(self.h, x0.ctypes.data_as(C.c_void_p), C.byref(r)))
        return np.array(r.x[:]), r


def load_ba_dataset(frame_id_file, lidar_pose_file, pointcloud_dir, keyframe_dir, map_file, global_variant=True, **_):
    """Edge list of OptimizeExtrinsicGlobal (or ...Local) from a dataset directory of the reference pipeline (iba_dataset_load_ba)."""
    from .formats import IbaDatasetPaths, _lib
    L = _lib()
    L.iba_dataset_load_ba.argtypes = [C.POINTER(IbaDatasetPaths), C.c_int32, C.POINTER(C.c_void_p)]
    L.iba_ba_dataset_desc.restype = C.POINTER(IbaBaDesc)
    L.iba_ba_dataset_desc.argtypes = [C.c_void_p]
    L.iba_ba_dataset_free.argtypes = [C.c_void_p]
    paths = IbaDatasetPaths(str(frame_id_file).encode(), str(lidar_pose_file).encode(), str(pointcloud_dir).encode(), str(keyframe_dir).encode(),
                            str(map_file).encode(), 1, 0, 3, 100)
    h = C.c_void_p()
    st = L.iba_dataset_load_ba(C.byref(paths), 1 if global_variant else 0, C.byref(h))
    if st != 0:
        raise IbaError(st, (L.iba_io_last_error() or b"").decode())
    try:
        d = L.iba_ba_dataset_desc(h).contents
        N, F = int(d.n_edges), int(d.n_frames)

        def view(ptr, dt, n):
            if n == 0:
                return np.zeros(0, dt)
            return np.frombuffer((C.c_char * (n * np.dtype(dt).itemsize)).from_address(ptr), dtype=dt, count=n).copy()

        prob = BaProblem(view(d.frame_Tlw6, np.float64, 6 * F), view(d.frame_intr, np.float64, 4 * F), view(d.edge_frame, np.int32, N),
                         view(d.edge_Xw, np.float64, 3 * N), view(d.edge_obs, np.float64, 2 * N), view(d.edge_info, np.float64, N), view(d.edge_slot, np.int32, N))
    finally:
        L.iba_ba_dataset_free(h)
    return prob
