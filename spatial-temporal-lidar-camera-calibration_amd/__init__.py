"""MI355X-native IBA cross-modality evaluation path — Python plumbing around the C-ABI library.

The product is csrc/ (hand-written HIP for gfx950 behind include/iba_mi355x.h). This module only
builds/loads libiba_mi355x.so and wraps the entry points with numpy arrays for tests and bench.py.
It never computes anything itself and never touches oracle/: if the HIP library (or a GPU) is
missing, calls fail loudly.

The directory name carries a hyphen; import it with
    importlib.import_module("spatial-temporal-lidar-camera-calibration_amd")
"""
import ctypes as C
import os
import subprocess

import numpy as np

from .abi import (IBA_MAX_BATCH, IbaCreateOptions, IbaLmOptions, IbaLmResult, IbaMadsOptions, IbaMadsResult, IbaBbo, IbaCostOut, IbaNormalOut, IbaParams, IbaProblemDesc, Problem, copy_params,
                  reference_yaml_params)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("IBA_LIB", os.path.join(_HERE, "libiba_mi355x.so"))  # IBA_LIB: diagnostic builds only
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "iba_mi355x.h")
# every header of the boundary: the product surface and the diagnostics (tests check that the library exports all they declare)
HEADER_PATHS = [HEADER_PATH, os.path.join(os.path.dirname(_HERE), "include", "iba_mi355x_debug.h")]

STATUS = {0: "IBA_OK", 1: "IBA_ERR_INVALID_ARG", 2: "IBA_ERR_NO_DEVICE", 3: "IBA_ERR_HIP", 4: "IBA_ERR_UNSUPPORTED", 5: "IBA_ERR_STATE", 6: "IBA_ERR_IO"}


class IbaError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"{STATUS.get(status, status)}: {msg}")
        self.status = status


def build_extension(force=False):
    """hipcc --offload-arch=gfx950 (cross-compiles without a GPU). Returns the .so path."""
    src_dir = os.path.join(_HERE, "csrc")
    # make tracks the dependencies itself (every .hip / .cpp / .hpp of csrc and the public header)
    if force:
        subprocess.check_call(["make", "-C", src_dir, "-s", "-B"])
    else:
        subprocess.check_call(["make", "-C", src_dir, "-s"])
    return LIB_PATH


_lib = None
ABI_VERSION = 2   # IBA_ABI_VERSION of include/iba_mi355x.h these ctypes structs mirror


def load_library():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise IbaError(2, f"{LIB_PATH} is not built (run __graft_entry__.build()); there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        L.iba_abi_version.restype = C.c_int32
        if L.iba_abi_version() != ABI_VERSION:   # iba_params has no struct_size: a stale library would read past (or short of) the struct
            raise IbaError(1, f"{LIB_PATH} speaks ABI {L.iba_abi_version()}, these bindings ABI {ABI_VERSION}: rebuild the library")
        L.iba_last_error.restype = C.c_char_p
        L.iba_last_error.argtypes = [C.c_void_p]
        L.iba_create.argtypes = [C.POINTER(IbaProblemDesc), C.POINTER(IbaParams), C.c_int, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
        L.iba_create_ex.argtypes = [C.POINTER(IbaProblemDesc), C.POINTER(IbaParams), C.c_int, C.c_int32, C.c_int32, C.POINTER(IbaCreateOptions), C.POINTER(C.c_void_p)]
        L.iba_destroy.argtypes = [C.c_void_p]
        L.iba_num_points.restype = C.c_int64
        L.iba_num_points.argtypes = [C.c_void_p]
        L.iba_num_keypoints.restype = C.c_int64
        L.iba_num_keypoints.argtypes = [C.c_void_p]
        L.iba_eval_cost_partial.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
        L.iba_eval_normal_partial.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
        L.iba_eval_full_partial.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
        L.iba_eval_factors_partial.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
        L.iba_comm_allreduce.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def default_params():
    p = IbaParams()
    load_library().iba_default_params(C.byref(p))
    return p


def create_options(**fields):
    """iba_default_create_options with fields overridden (engine options: none changes a result bit)"""
    o = IbaCreateOptions()
    load_library().iba_default_create_options(C.byref(o))
    for k, v in fields.items():
        if k not in dict(IbaCreateOptions._fields_):
            raise AttributeError(k)
        setattr(o, k, v)
    return o


def partial_stride():
    return int(load_library().iba_partial_stride())


def finalize_cost(params, partials):
    partials = np.ascontiguousarray(partials, np.float64).reshape(-1, partial_stride())
    B = len(partials)
    out = (IbaCostOut * B)()
    st = load_library().iba_finalize_cost(C.byref(params), _p(partials), C.c_int32(B), out)
    if st != 0:
        raise IbaError(st, "iba_finalize_cost")
    return list(out)


def finalize_normal(params, partials):
    partials = np.ascontiguousarray(partials, np.float64).reshape(-1, partial_stride())
    B = len(partials)
    out = (IbaNormalOut * B)()
    st = load_library().iba_finalize_normal(C.byref(params), _p(partials), C.c_int32(B), out)
    if st != 0:
        raise IbaError(st, "iba_finalize_normal")
    return list(out)


class IbaHandle:
    """iba_handle wrapper. One evaluation at a time per handle (as BALoss::eval_x)."""

    def __init__(self, problem, params=None, device=0, frame_begin=0, frame_end=None, options=None):
        """options: an IbaCreateOptions (create_options(...)) or a dict of its fields; None = the defaults (iba_create)"""
        self.lib = load_library()
        self.problem = problem
        self.params = copy_params(params) if params is not None else default_params()
        self._desc = problem.desc()
        self.h = C.c_void_p(None)
        fe = problem.n_frames if frame_end is None else frame_end
        self.frame_begin, self.frame_end = frame_begin, fe
        if options is None:
            st = self.lib.iba_create(C.byref(self._desc), C.byref(self.params), C.c_int(device), C.c_int32(frame_begin), C.c_int32(fe), C.byref(self.h))
        else:
            self.options = create_options(**options) if isinstance(options, dict) else options
            st = self.lib.iba_create_ex(C.byref(self._desc), C.byref(self.params), C.c_int(device), C.c_int32(frame_begin), C.c_int32(fe), C.byref(self.options), C.byref(self.h))
        if st != 0:
            raise IbaError(st, self.lib.iba_last_error(None).decode())

    def _chk(self, st):
        if st != 0:
            raise IbaError(st, self.lib.iba_last_error(self.h).decode())

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            self.lib.iba_destroy(self.h)
            self.h = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_params(self, params):
        self.params = copy_params(params)
        self._chk(self.lib.iba_set_params(self.h, C.byref(self.params)))

    @staticmethod
    def _x(x):
        x = np.ascontiguousarray(np.atleast_2d(x), np.float64)
        assert x.shape[1] == 7
        return x

    def eval_cost(self, x):
        x = self._x(x)
        B = len(x)
        out = (IbaCostOut * B)()
        self._chk(self.lib.iba_eval_cost(self.h, _p(x), C.c_int32(B), out))
        return list(out)

    def eval_bbo(self, x, he_threshold, valid_rate):
        x = self._x(x)
        B = len(x)
        out = (IbaBbo * B)()
        self._chk(self.lib.iba_eval_bbo(self.h, _p(x), C.c_int32(B), C.c_double(he_threshold), C.c_double(valid_rate), out))
        return list(out)

    def eval_normal(self, x):
        x = self._x(x)
        B = len(x)
        out = (IbaNormalOut * B)()
        self._chk(self.lib.iba_eval_normal(self.h, _p(x), C.c_int32(B), out))
        return list(out)

    def eval_full(self, x):
        x = self._x(x)
        B = len(x)
        cost = (IbaCostOut * B)()
        nrm = (IbaNormalOut * B)()
        self._chk(self.lib.iba_eval_full(self.h, _p(x), C.c_int32(B), cost, nrm))
        return list(cost), list(nrm)

    def calibrate_lm(self, x0, **opts):
        """iba_local's outer loop (re-associate, LM on the frozen problem, until allClose) on the device path."""
        o = IbaLmOptions()
        self.lib.iba_default_lm_options(C.byref(o))
        for k, v in opts.items():
            setattr(o, k, v)
        r = IbaLmResult()
        x0 = np.ascontiguousarray(x0, np.float64)
        self._chk(self.lib.iba_calibrate_lm(self.h, _p(x0), C.byref(o), C.byref(r)))
        return np.array(r.x[:]), r

    def calibrate_mads(self, x0, trace=False, record=False, **opts):
        """Global stage: batch-aware MADS on BALoss::eval_x's objective + 3 progressive-barrier constraints
        (the caller the reference gets from NOMAD, iba_global.cpp:551-602). opts override iba_default_mads_options;
        lb/ub given as 7-vectors are ABSOLUTE bounds. trace=True also returns the evaluated points (rows of x[7], f);
        record=True returns them together with the size of every black-box call (the batches iba_eval_bbo was given)."""
        x0 = np.ascontiguousarray(x0, np.float64)
        o = mads_options(x0, **opts)
        r = IbaMadsResult()
        if not trace and not record:
            self._chk(self.lib.iba_calibrate_mads(self.h, _p(x0), C.byref(o), C.byref(r)))
            return np.array(r.x[:]), r
        tr = np.zeros((int(o.max_bb_eval) + IBA_MAX_BATCH, 8))   # (the last batch may overshoot the budget by less than one batch)
        n = C.c_int32(0)
        if not record:
            self._chk(self.lib.iba_calibrate_mads_trace(self.h, _p(x0), C.byref(o), C.byref(r), _p(tr), C.c_int32(len(tr)), C.byref(n)))
            return np.array(r.x[:]), r, tr[: min(n.value, len(tr))]
        bs = np.zeros(len(tr), np.int32)
        nb = C.c_int32(0)
        self._chk(self.lib.iba_calibrate_mads_record(self.h, _p(x0), C.byref(o), C.byref(r), _p(tr), C.c_int32(len(tr)), C.byref(n), _p(bs), C.c_int32(len(bs)), C.byref(nb)))
        return np.array(r.x[:]), r, tr[: min(n.value, len(tr))], bs[: min(nb.value, len(bs))].copy()

    def build_problem(self, x):
        x = np.ascontiguousarray(x, np.float64)
        self._chk(self.lib.iba_build_problem(self.h, _p(x)))

    def eval_factors(self, x):
        x = self._x(x)
        B = len(x)
        out = (IbaNormalOut * B)()
        self._chk(self.lib.iba_eval_factors(self.h, _p(x), C.c_int32(B), out))
        return list(out)

    def eval_residuals(self, x):
        x = np.ascontiguousarray(x, np.float64)
        n = C.c_int64(0)
        self._chk(self.lib.iba_eval_residuals(self.h, _p(x), None, None, None, None, C.byref(n)))
        m = n.value
        r = np.zeros(m)
        J = np.zeros((m, 7))
        bid = np.zeros(m, np.int32)
        kind = np.zeros(m, np.int32)
        if m:
            self._chk(self.lib.iba_eval_residuals(self.h, _p(x), _p(r), _p(J), _p(bid), _p(kind), C.byref(n)))
        return r, J, bid, kind

    def correspondences(self, x, frame):
        o = self.problem.arrays["kp_offset"]
        K = int(o[frame + 1] - o[frame])
        kp = np.zeros(max(K, 1), np.uint32)
        pt = np.zeros(max(K, 1), np.uint32)
        n = C.c_int32(0)
        x = np.ascontiguousarray(x, np.float64)
        self._chk(self.lib.iba_get_correspondences(self.h, _p(x), C.c_int32(frame), _p(kp), _p(pt), C.c_int32(K), C.byref(n)))
        return kp[: n.value].copy(), pt[: n.value].copy()

    # --- multi-GPU building blocks: partial sums into caller-owned device memory ---
    def debug_nn(self, frame, queries, mode=1):
        """Exact 1-NN of LiDAR-frame query points in the scan of a (local) frame through the search kernel's own kd search
        (mode 1: as the association path's query, 2: as the cost path's, 3 / 4: both paths together, the query first / second):
        (original point indices, exact squared distances)."""
        q = np.ascontiguousarray(queries, np.float64).reshape(-1, 3)
        idx = np.zeros(len(q), np.uint32)
        d2 = np.zeros(len(q), np.float64)
        self._chk(self.lib.iba_debug_nn(self.h, C.c_int32(frame), _p(q), C.c_int32(len(q)), C.c_int32(mode),
                                        idx.ctypes.data_as(C.POINTER(C.c_uint32)), _p(d2)))
        return idx, d2

    def call_latency(self, xs, kind="cost", iters=200):
        """(median_ms, min_ms) of a blocking entry point called `iters` times back to back from C (iba_debug_call_latency): kind = "cost"
        (iba_eval_cost), "full" (iba_eval_full) or "factors" (iba_eval_factors)"""
        xs = np.ascontiguousarray(xs, np.float64).reshape(-1, 7)
        med, mn = C.c_double(0), C.c_double(0)
        self.lib.iba_debug_call_latency.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        self._chk(self.lib.iba_debug_call_latency(self.h, _p(xs), C.c_int32(len(xs)), C.c_int32({"cost": 0, "full": 1, "factors": 2}[kind]), C.c_int32(iters), C.byref(med), C.byref(mn)))
        return med.value, mn.value

    def debug_knn(self, frame, points, k=30, r2=np.inf):
        """The sorted neighbour lists the plane fits build (iba_plane_kernel's list builder) around scan points given by ORIGINAL
        index: (idx [n, k], d2 [n, k], cnt [n])."""
        pts = np.ascontiguousarray(points, np.uint32).reshape(-1)
        n = len(pts)
        idx = np.zeros((n, k), np.uint32)
        d2 = np.zeros((n, k), np.float64)
        cnt = np.zeros(n, np.int32)
        self.lib.iba_debug_knn.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
        self._chk(self.lib.iba_debug_knn(self.h, C.c_int32(frame), _p(pts), C.c_int32(n), C.c_int32(k), C.c_double(r2), _p(idx), _p(d2), _p(cnt)))
        return idx, d2, cnt

    def debug_plane(self, frame, point, which=1):
        """(normal[3], reg_sum, far_d2, k) of the memoised local plane at ORIGINAL scan point index `point` of `frame` (which: 0 cost path, 1 local)"""
        out = np.zeros(5)
        k = C.c_int32(0)
        self._chk(self.lib.iba_debug_plane(self.h, C.c_int32(frame), C.c_uint32(int(point)), C.c_int32(which), _p(out), C.byref(k)))
        return out[:3].copy(), float(out[3]), float(out[4]), k.value

    def eval_cost_partial(self, x, d_partials_ptr, stream_ptr=None):
        x = self._x(x)
        self._chk(self.lib.iba_eval_cost_partial(self.h, _p(x), C.c_int32(len(x)), C.c_void_p(d_partials_ptr), C.c_void_p(stream_ptr)))

    def eval_normal_partial(self, x, d_partials_ptr, stream_ptr=None):
        x = self._x(x)
        self._chk(self.lib.iba_eval_normal_partial(self.h, _p(x), C.c_int32(len(x)), C.c_void_p(d_partials_ptr), C.c_void_p(stream_ptr)))

    @property
    def last_assoc2_threads(self):
        self.lib.iba_debug_last_assoc2_threads.argtypes = [C.c_void_p]
        self.lib.iba_debug_last_assoc2_threads.restype = C.c_int32
        return int(self.lib.iba_debug_last_assoc2_threads(self.h))

    def geo_correspondences(self, frame, src_xyz, max_distance=0.05):
        """GeoCalib.h:18-33 computeCorrespondence with the scan of local keyframe `frame` as the target cloud -> (source indices, target indices)"""
        src = np.ascontiguousarray(np.asarray(src_xyz, np.float64).reshape(-1, 3))
        n = len(src)
        o_s = np.zeros(max(n, 1), np.uint32); o_t = np.zeros(max(n, 1), np.uint32); cnt = C.c_int32(0)
        self.lib.iba_geo_correspondences.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_double, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]
        self._chk(self.lib.iba_geo_correspondences(self.h, C.c_int32(frame), _p(src), C.c_int32(n), C.c_double(max_distance), _p(o_s), _p(o_t), C.byref(cnt)))
        return o_s[: cnt.value].copy(), o_t[: cnt.value].copy()

    def debug_factor_ranges(self, B):
        """ranges per candidate iba_factor2_kernel would cut a batch of B into; 0 = the default factor kernel runs"""
        self.lib.iba_debug_factor_ranges.argtypes = [C.c_void_p, C.c_int32]
        self.lib.iba_debug_factor_ranges.restype = C.c_int32
        return int(self.lib.iba_debug_factor_ranges(self.h, C.c_int32(B)))

    @property
    def last_nn_threads(self):
        """threads per block of the last search launch (64: one-wave blocks; 256)"""
        self.lib.iba_debug_last_nn_threads.argtypes = [C.c_void_p]
        self.lib.iba_debug_last_nn_threads.restype = C.c_int32
        return int(self.lib.iba_debug_last_nn_threads(self.h))

    @property
    def last_nn_list(self):
        """> 0: the last search launch was iba_nn_list_kernel (opt-in, IBA_NN_LIST=1) with that many workers per (XCD, group of candidates)"""
        self.lib.iba_debug_last_nn_list.argtypes = [C.c_void_p]
        self.lib.iba_debug_last_nn_list.restype = C.c_int32
        return int(self.lib.iba_debug_last_nn_list(self.h))

    def debug_last_partials(self, B):
        out = np.zeros((B, partial_stride()))
        self._chk(self.lib.iba_debug_last_partials(self.h, _p(out), C.c_int32(B)))
        return out

    def eval_full_partial(self, x, d_partials_ptr, stream_ptr=None):
        x = self._x(x)
        self._chk(self.lib.iba_eval_full_partial(self.h, _p(x), C.c_int32(len(x)), C.c_void_p(d_partials_ptr), C.c_void_p(stream_ptr)))

    @property
    def last_path(self):
        """1: the last evaluation shared the 2d-3d pair search over the batch, 0: every candidate searched for itself"""
        self.lib.iba_debug_last_path.argtypes = [C.c_void_p]
        return int(self.lib.iba_debug_last_path(self.h))

    @property
    def pairs_builds(self):
        self.lib.iba_debug_pairs_builds.argtypes = [C.c_void_p]
        return int(self.lib.iba_debug_pairs_builds(self.h))

    def rescans(self, reset=True):
        """association blocks since the last reset that took the rescan-every-point fallback (speed only)"""
        self.lib.iba_debug_rescans.restype = C.c_int64
        self.lib.iba_debug_rescans.argtypes = [C.c_void_p, C.c_int32]
        return int(self.lib.iba_debug_rescans(self.h, 1 if reset else 0))

    def counters(self, reset=True):
        """(association blocks that rescanned every point, assoc2 blocks whose winner note list overflowed, 0, 0) since the last reset"""
        out = (C.c_uint32 * 4)()
        self.lib.iba_debug_counters.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_int32]
        self._chk(self.lib.iba_debug_counters(self.h, out, 1 if reset else 0))
        return tuple(int(v) for v in out)

    @property
    def pair_lists(self):
        """(overflowed pair lists, pair lists read, longest list) of the last call"""
        out = (C.c_int32 * 3)()
        self.lib.iba_debug_pair_lists.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
        self._chk(self.lib.iba_debug_pair_lists(self.h, out))
        return int(out[0]), int(out[1]), int(out[2])

    @property
    def mean_pairs(self):
        self.lib.iba_debug_mean_pairs.restype = C.c_double
        self.lib.iba_debug_mean_pairs.argtypes = [C.c_void_p]
        return float(self.lib.iba_debug_mean_pairs(self.h))

    @property
    def nn_left_to_tree(self):
        self.lib.iba_debug_nn_left_to_tree.restype = C.c_double
        self.lib.iba_debug_nn_left_to_tree.argtypes = [C.c_void_p]
        return float(self.lib.iba_debug_nn_left_to_tree(self.h))

    @property
    def anchor_builds(self):
        self.lib.iba_debug_anchor_builds.argtypes = [C.c_void_p]
        return int(self.lib.iba_debug_anchor_builds(self.h))

    def set_timing(self, on=True):
        self._chk(self.lib.iba_set_timing(self.h, C.c_int32(1 if on else 0)))

    def last_kernel_ms(self):
        a = C.c_float(0)
        b = C.c_float(0)
        self._chk(self.lib.iba_last_kernel_ms(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    @property
    def n_points(self):
        return int(self.lib.iba_num_points(self.h))

    @property
    def n_keypoints(self):
        return int(self.lib.iba_num_keypoints(self.h))


class IbaGroup:
    """iba_group wrapper: one process, several GPUs of a node, frames sharded over them, one RCCL all-reduce per evaluation."""

    def __init__(self, problem, params=None, devices=(0,), host_reduce=False):
        """host_reduce: sum the partial blocks on the host in rank order (IBA_GROUP_REDUCE_HOST; no RCCL, a device may repeat)."""
        self.lib = load_library()
        self.problem = problem
        self.params = copy_params(params) if params is not None else default_params()
        self._desc = problem.desc()
        self.g = C.c_void_p(None)
        dev = (C.c_int32 * len(devices))(*devices)
        self.lib.iba_group_last_error.restype = C.c_char_p
        self.lib.iba_group_last_error.argtypes = [C.c_void_p]
        self.lib.iba_group_destroy.argtypes = [C.c_void_p]
        self.lib.iba_group_last_issue_us.restype = C.c_double
        self.lib.iba_group_last_issue_us.argtypes = [C.c_void_p]
        self.lib.iba_group_comm_ranks.argtypes = [C.c_void_p]
        st = self.lib.iba_group_create_ex(C.byref(self._desc), C.byref(self.params), dev, C.c_int32(len(devices)), C.c_int32(1 if host_reduce else 0), C.byref(self.g))
        if st != 0:
            raise IbaError(st, self.lib.iba_group_last_error(None).decode())

    @property
    def comm_ranks(self):
        """ncclCommCount of the group's communicator (0: host reduction)."""
        return int(self.lib.iba_group_comm_ranks(self.g))

    @property
    def last_issue_us(self):
        return float(self.lib.iba_group_last_issue_us(self.g))

    @property
    def last_enqueue_us(self):
        """host time until the last device's launch chain and collective of the last call were enqueued"""
        self.lib.iba_group_last_enqueue_us.restype = C.c_double
        self.lib.iba_group_last_enqueue_us.argtypes = [C.c_void_p]
        return float(self.lib.iba_group_last_enqueue_us(self.g))

    def set_params(self, params):
        self.params = copy_params(params)
        self._chk(self.lib.iba_group_set_params(self.g, C.byref(self.params)))

    def _chk(self, st):
        if st != 0:
            raise IbaError(st, self.lib.iba_group_last_error(self.g).decode())

    def close(self):
        if getattr(self, "g", None) and self.g.value:
            self.lib.iba_group_destroy(self.g)
            self.g = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def frame_range(self, rank):
        a, b = C.c_int32(0), C.c_int32(0)
        self._chk(self.lib.iba_group_frame_range(self.g, C.c_int32(rank), C.byref(a), C.byref(b)))
        return a.value, b.value

    def eval_cost(self, x):
        x = IbaHandle._x(x)
        out = (IbaCostOut * len(x))()
        self._chk(self.lib.iba_group_eval_cost(self.g, _p(x), C.c_int32(len(x)), out))
        return list(out)

    def eval_full(self, x):
        x = IbaHandle._x(x)
        cost, nrm = (IbaCostOut * len(x))(), (IbaNormalOut * len(x))()
        self._chk(self.lib.iba_group_eval_full(self.g, _p(x), C.c_int32(len(x)), cost, nrm))
        return list(cost), list(nrm)

    def eval_normal(self, x):
        x = IbaHandle._x(x)
        nrm = (IbaNormalOut * len(x))()
        self._chk(self.lib.iba_group_eval_normal(self.g, _p(x), C.c_int32(len(x)), nrm))
        return list(nrm)

    def build_problem(self, x):
        x = np.ascontiguousarray(x, np.float64)
        self._chk(self.lib.iba_group_build_problem(self.g, _p(x)))

    def eval_factors(self, x):
        x = IbaHandle._x(x)
        nrm = (IbaNormalOut * len(x))()
        self._chk(self.lib.iba_group_eval_factors(self.g, _p(x), C.c_int32(len(x)), nrm))
        return list(nrm)

    def calibrate_lm(self, x0, **opts):
        o = IbaLmOptions()
        self.lib.iba_default_lm_options(C.byref(o))
        for k, v in opts.items():
            setattr(o, k, v)
        r = IbaLmResult()
        x0 = np.ascontiguousarray(x0, np.float64)
        self._chk(self.lib.iba_group_calibrate_lm(self.g, _p(x0), C.byref(o), C.byref(r)))
        return np.array(r.x[:]), r

    def calibrate_mads(self, x0, **opts):
        x0 = np.ascontiguousarray(x0, np.float64)
        o = mads_options(x0, **opts)
        r = IbaMadsResult()
        self._chk(self.lib.iba_group_calibrate_mads(self.g, _p(x0), C.byref(o), C.byref(r)))
        return np.array(r.x[:]), r


def rccl_info():
    """(text, runtime version code, header version code) of the librccl this process runs (loaded lazily by the library)."""
    L = load_library()
    buf = C.create_string_buffer(1024)
    rv, hv = C.c_int32(0), C.c_int32(0)
    st = L.iba_rccl_info(buf, C.c_int32(1024), C.byref(rv), C.byref(hv))
    if st != 0:
        raise IbaError(st, buf.value.decode())
    return buf.value.decode(), rv.value, hv.value


def mads_options(x0, **opts):
    L = load_library()
    x0 = np.ascontiguousarray(x0, np.float64)
    o = IbaMadsOptions()
    L.iba_default_mads_options(_p(x0), C.byref(o))
    for k, v in opts.items():
        if k in ("lb", "ub", "init_frame"):
            for i in range(7):
                getattr(o, k)[i] = float(v[i])
        else:
            setattr(o, k, v)
    return o


def debug_cand(x):
    """(host only) R[9], t[3], dR[3][9], dt[6][3], s of a candidate exactly as the factor kernel reads them: 58 doubles"""
    L = load_library()
    x = np.ascontiguousarray(x, np.float64)
    out = np.zeros(58)
    L.iba_debug_cand.argtypes = [C.c_void_p, C.c_void_p]
    st = L.iba_debug_cand(x.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    if st != 0:
        raise IbaError(st, "iba_debug_cand")
    return out


def debug_div2_selftest(num0, num1, den, device=0):
    """(q0, q1, ref0, ref1, n_fast): num0 / den and num1 / den as the kernels' shared-reciprocal division computes them, as the compiler's
    f64 division does, and how many triples took the shared-reciprocal path (include/iba_mi355x_debug.h)."""
    L = load_library()
    a, b, d = (np.ascontiguousarray(v, np.float64).reshape(-1) for v in (num0, num1, den))
    assert len(a) == len(b) == len(d)
    out = [np.zeros(len(a)) for _ in range(4)]
    nf = C.c_int64(0)
    L.iba_debug_div2_selftest.argtypes = [C.c_int32] + [C.c_void_p] * 3 + [C.c_int64] + [C.c_void_p] * 4 + [C.POINTER(C.c_int64)]
    st = L.iba_debug_div2_selftest(C.c_int32(device), _p(a), _p(b), _p(d), C.c_int64(len(a)), *[_p(o) for o in out], C.byref(nf))
    if st != 0:
        raise IbaError(st, "iba_debug_div2_selftest")
    return out[0], out[1], out[2], out[3], nf.value


def mads_selftest(problem, x0, trace=False, **opts):
    """The MADS driver on a built-in analytic black box (host only, no GPU). trace=True also returns the evaluated points."""
    L = load_library()
    x0 = np.ascontiguousarray(x0, np.float64)
    o = mads_options(x0, **opts)
    r = IbaMadsResult()
    tr = np.zeros((int(o.max_bb_eval) if trace else 1, 8))
    n = C.c_int32(0)
    st = L.iba_mads_selftest_trace(C.c_int32(problem), _p(x0), C.byref(o), C.byref(r), _p(tr) if trace else None, C.c_int32(len(tr) if trace else 0), C.byref(n))
    if st != 0:
        raise IbaError(st, "iba_mads_selftest")
    return (np.array(r.x[:]), r, tr[: n.value]) if trace else (np.array(r.x[:]), r)


def shard_frames(n_frames, world_size, rank, weights=None):
    """Contiguous frame range of `rank`, balanced by `weights` (points per frame; uniform if None)."""
    w = np.ones(n_frames) if weights is None else np.asarray(weights, np.float64)
    c = np.concatenate([[0.0], np.cumsum(w)])
    tot = c[-1]
    cuts = [int(np.searchsorted(c, tot * r / world_size, side="left")) for r in range(world_size + 1)]
    cuts[0], cuts[-1] = 0, n_frames
    for i in range(1, world_size + 1):
        cuts[i] = max(cuts[i], cuts[i - 1])
    return cuts[rank], cuts[rank + 1]
