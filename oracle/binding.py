"""TEST INFRASTRUCTURE ONLY — ctypes binding of oracle/liboracle.so (the CPU restatement of the
reference path) and oracle/_ref/libref_nanoflann.so (the reference's own nanoflann).

May be imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only.
"""
import ctypes as C
import importlib
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_abi = importlib.import_module("spatial-temporal-lidar-camera-calibration_amd.abi")
IbaCostOut, IbaNormalOut, IbaBbo, IbaParams = _abi.IbaCostOut, _abi.IbaNormalOut, _abi.IbaBbo, _abi.IbaParams


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = [os.path.join(_HERE, f) for f in ("iba_oracle.cpp", "oracle_math.hpp", "oracle_kdtree.hpp")]
    stale = (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s", "all"])
    return so


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.oracle_create.restype = C.c_void_p
        _lib.oracle_create.argtypes = [C.c_void_p, C.c_int, C.c_int]
        _lib.oracle_destroy.argtypes = [C.c_void_p]
    return _lib


def ref_lib():
    """The reference's nanoflann (None when oracle/_ref was never built: no /root/reference)."""
    global _ref
    if _ref is None:
        p = os.path.join(_HERE, "_ref", "libref_nanoflann.so")
        if not os.path.exists(p):
            return None
        _ref = C.CDLL(p)
    return _ref


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def geo_correspondences(which, src, tgt, max_distance):
    """GeoCalib.h:18-33 computeCorrespondence. which: 'oracle' (this directory's restatement) | 'ref' (the reference's own nanoflann, oracle/_ref).
    src [n, 3], tgt [m, 3] doubles -> (source indices, target indices) of the kept pairs."""
    L = lib() if which == "oracle" else ref_lib()
    fn = L.oracle_geo_correspondences if which == "oracle" else L.ref_geo_correspondences
    src = np.ascontiguousarray(src, np.float64).reshape(-1, 3); tgt = np.ascontiguousarray(tgt, np.float64).reshape(-1, 3)
    o_s = np.zeros(max(len(src), 1), np.uint32); o_t = np.zeros(max(len(src), 1), np.uint32); n = C.c_int64(0)
    rc = fn(_p(src), C.c_uint64(len(src)), _p(tgt), C.c_uint64(len(tgt)), C.c_double(max_distance), _p(o_s), _p(o_t), C.byref(n))
    assert rc == 0
    return o_s[: n.value].copy(), o_t[: n.value].copy()


def knn(which, dim, pts, leaf, queries, k):
    """which: 'oracle' | 'ref'. Returns (idx[nq,k] u32, d2[nq,k] f64, cnt[nq] i32)."""
    L = lib() if which == "oracle" else ref_lib()
    fn = L.oracle_knn if which == "oracle" else L.ref_nanoflann_knn
    pts = np.ascontiguousarray(pts, np.float64)
    queries = np.ascontiguousarray(queries, np.float64)
    nq = len(queries)
    idx = np.zeros((nq, k), np.uint32)
    d2 = np.zeros((nq, k), np.float64)
    cnt = np.zeros(nq, np.int32)
    rc = fn(C.c_int(dim), _p(pts), C.c_uint64(len(pts)), C.c_int(leaf), _p(queries), C.c_uint64(nq), C.c_int(k), _p(idx), _p(d2), _p(cnt))
    assert rc == 0
    return idx, d2, cnt


class Oracle:
    def __init__(self, problem, leaf2d=10, leaf3d=30):
        self.problem = problem
        self._desc = problem.desc()
        self.h = lib().oracle_create(C.byref(self._desc), leaf2d, leaf3d)

    def close(self):
        if self.h:
            lib().oracle_destroy(C.c_void_p(self.h))
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def eval_cost(self, params, x, nthreads=1):
        x = np.ascontiguousarray(np.atleast_2d(x), np.float64)
        B = len(x)
        out = (IbaCostOut * B)()
        lib().oracle_eval_cost(C.c_void_p(self.h), C.byref(params), _p(x), C.c_int(B), out, C.c_int(nthreads))
        return list(out)

    def eval_cost_raw(self, params, x, f_begin, f_end):
        """un-normalised sums over frames [f_begin, f_end) in the order of the product's partial block"""
        x = np.ascontiguousarray(x, np.float64)
        raw = np.zeros(12)
        lib().oracle_eval_cost_raw(C.c_void_p(self.h), C.byref(params), _p(x), C.c_int(f_begin), C.c_int(f_end), _p(raw))
        return raw

    def eval_bbo(self, params, x, he_threshold, valid_rate, nthreads=1):
        x = np.ascontiguousarray(np.atleast_2d(x), np.float64)
        B = len(x)
        out = (IbaBbo * B)()
        lib().oracle_eval_bbo(C.c_void_p(self.h), C.byref(params), _p(x), C.c_int(B), C.c_double(he_threshold), C.c_double(valid_rate), out, C.c_int(nthreads))
        return list(out)

    def correspondences(self, params, x, frame):
        K = int(self.problem.arrays["kp_offset"][frame + 1] - self.problem.arrays["kp_offset"][frame])
        kp = np.zeros(K, np.uint32)
        pt = np.zeros(K, np.uint32)
        n = C.c_int(0)
        x = np.ascontiguousarray(x, np.float64)
        lib().oracle_get_correspondences(C.c_void_p(self.h), C.byref(params), _p(x), C.c_int(frame), _p(kp), _p(pt), C.c_int(K), C.byref(n))
        return kp[: n.value].copy(), pt[: n.value].copy()

    def build_problem(self, params, x):
        x = np.ascontiguousarray(x, np.float64)
        lib().oracle_build_problem(C.c_void_p(self.h), C.byref(params), _p(x))
        return lib().oracle_num_factors(C.c_void_p(self.h))

    def eval_factors(self, params, x):
        x = np.ascontiguousarray(np.atleast_2d(x), np.float64)
        B = len(x)
        out = (IbaNormalOut * B)()
        lib().oracle_eval_factors(C.c_void_p(self.h), C.byref(params), _p(x), C.c_int(B), out)
        return list(out)

    def set_frame_range(self, f_begin=0, f_end=-1):
        """test aid: BuildProblem / eval_normal over the frames [f_begin, f_end) only (one rank's share)"""
        lib().oracle_set_frame_range(C.c_void_p(self.h), C.c_int(f_begin), C.c_int(f_end))

    @staticmethod
    def set_exact_sums(on):
        """test aid: add the per-block contributions of the normal equations in long double (see iba_oracle.cpp)"""
        lib().oracle_set_exact_sums(C.c_int(int(on)))   # True / 1: long-double SUMS of the double rows; 2: rows and sums in long double ("the truth")

    def eval_normal_truth(self, params, x):
        """the normal equations with rows, weights and sums in long double (x87 80-bit), rounded to double at the end: what the reference's formulas are
        worth at x. Single-threaded (one candidate: ~1 s per 50 k blocks)."""
        Oracle.set_exact_sums(2)
        try:
            return self.eval_normal(params, x, nthreads=1)
        finally:
            Oracle.set_exact_sums(0)

    def eval_normal(self, params, x, nthreads=1):
        x = np.ascontiguousarray(np.atleast_2d(x), np.float64)
        B = len(x)
        out = (IbaNormalOut * B)()
        lib().oracle_eval_normal(C.c_void_p(self.h), C.byref(params), _p(x), C.c_int(B), out, C.c_int(nthreads))
        return list(out)

    def eval_residuals(self, x):
        x = np.ascontiguousarray(x, np.float64)
        n = C.c_int64(0)
        lib().oracle_eval_residuals(C.c_void_p(self.h), _p(x), None, None, None, None, None, C.byref(n))
        m = n.value
        r = np.zeros(m)
        J = np.zeros((m, 7))
        bid = np.zeros(m, np.int32)
        kind = np.zeros(m, np.int32)
        fk = np.zeros((m, 2), np.int32)
        if m:
            lib().oracle_eval_residuals(C.c_void_p(self.h), _p(x), _p(r), _p(J), _p(bid), _p(kind), _p(fk), C.byref(n))
        return r, J, bid, kind, fk

    def block_conditioning(self, x):
        """per residual block of the frozen problem, at x: how far the block's quotients are from cancelling (1 = well-conditioned,
        -> 0 = a plane factor whose viewing ray lies in its plane / whose reprojection depth vanishes)"""
        x = np.ascontiguousarray(x, np.float64)
        n = C.c_int64(0)
        lib().oracle_block_conditioning(C.c_void_p(self.h), _p(x), None, C.byref(n))
        c = np.ones(max(n.value, 1))
        if n.value:
            lib().oracle_block_conditioning(C.c_void_p(self.h), _p(x), _p(c), C.byref(n))
        return c[: n.value]

    def block_forward_error(self, x):
        """per residual block of the frozen problem, at x: the forward error of the oracle's own double evaluation, measured against
        the same formulas in long double, relative to the block's scale (two correct double evaluations agree no better than this)"""
        x = np.ascontiguousarray(x, np.float64)
        n = C.c_int64(0)
        lib().oracle_block_forward_error(C.c_void_p(self.h), _p(x), None, C.byref(n))
        e = np.zeros(max(n.value, 1))
        if n.value:
            lib().oracle_block_forward_error(C.c_void_p(self.h), _p(x), _p(e), C.byref(n))
        return e[: n.value]

    def block_sensitivity(self, x):
        """per residual block of the frozen problem, at x: how far its long-double rows move (relative to the block's scale) when x moves
        by one unit in the last place — what two correct double evaluations that derive R, t, dR, dt from x on their own may differ by"""
        x = np.ascontiguousarray(x, np.float64)
        n = C.c_int64(0)
        lib().oracle_block_sensitivity(C.c_void_p(self.h), _p(x), None, C.byref(n))
        e = np.zeros(max(n.value, 1))
        if n.value:
            lib().oracle_block_sensitivity(C.c_void_p(self.h), _p(x), _p(e), C.byref(n))
        return e[: n.value]

    def block_input_sensitivity(self, x):
        """per residual block (plane factors; 0 for the others): the largest relative change of its rows, in the device kernel's operation
        order, when every derived input (R, t, dR, dt, n0) moves by one unit in the last place (8 random sign draws)"""
        x = np.ascontiguousarray(x, np.float64)
        n = C.c_int64(0)
        lib().oracle_block_input_sensitivity(C.c_void_p(self.h), _p(x), None, C.byref(n))
        e = np.zeros(max(n.value, 1))
        if n.value:
            lib().oracle_block_input_sensitivity(C.c_void_p(self.h), _p(x), _p(e), C.byref(n))
        return e[: n.value]

    def block_kernel_order(self, block, cand58, normal, rows):
        """rows x 8 (r, J) of a plane-factor block in the device kernel's operation order from the DEVICE's inputs (iba_debug_cand, its plane normal); None for other kinds"""
        c = np.ascontiguousarray(cand58, np.float64); n = np.ascontiguousarray(normal, np.float64)
        out = np.zeros((rows, 8))
        if lib().oracle_block_kernel_order(C.c_void_p(self.h), C.c_int64(block), _p(c), _p(n), _p(out)) != 0:
            return None
        return out

    def block_three_ways(self, block, x, rows, variant=0):
        """(Dual<7> double, long double rounded, the device kernel's operation order in double) rows x 8 (r, J) of a plane-factor block; None for other kinds"""
        x = np.ascontiguousarray(x, np.float64)
        o0, o1, o2 = np.zeros((rows, 8)), np.zeros((rows, 8)), np.zeros((rows, 8))
        if lib().oracle_block_three_ways(C.c_void_p(self.h), C.c_int64(block), _p(x), C.c_int(variant), _p(o0), _p(o1), _p(o2)) != 0:
            return None
        return o0, o1, o2

    def block_normal(self, block):
        """(plane normal, the scan point it was fitted at, frame) of a residual block of the frozen problem; None for a point-to-point block"""
        n, q, f = np.zeros(3), np.zeros(3), C.c_int32(0)
        if lib().oracle_block_normal(C.c_void_p(self.h), C.c_int64(block), _p(n), _p(q), C.byref(f)) != 0:
            return None
        return n, q, f.value

    def block_rows_with_normal(self, block, x, normal, rows):
        """(r, J, forward error) of one block of the frozen problem at x with `normal` in place of its own plane normal"""
        x = np.ascontiguousarray(x, np.float64)
        n = np.ascontiguousarray(normal, np.float64)
        r, J, e = np.zeros(rows), np.zeros((rows, 7)), C.c_double(0)
        st = lib().oracle_block_rows_with_normal(C.c_void_p(self.h), C.c_int64(block), _p(x), _p(n), _p(r), _p(J), C.byref(e))
        assert st == 0
        return r, J, e.value

    def plane_edge20(self, factor_index, x):
        """IBAPlaneEdge (the g2o twin of IBA_PlaneFactor, IBACalib.hpp:103-140): the block's residuals zero-padded to 20, J 20 x 7"""
        x = np.ascontiguousarray(x, np.float64)
        e = np.zeros(20)
        J = np.zeros((20, 7))
        st = lib().oracle_eval_plane_edge20(C.c_void_p(self.h), C.c_int64(factor_index), _p(x), _p(e), _p(J))
        return (e, J) if st == 0 else None

    def plane_at(self, frame, pt_idx, radius, max_pts):
        k = C.c_int32(0)
        far = C.c_double(0)
        nrm = np.zeros(3)
        reg = C.c_double(0)
        lib().oracle_plane_at(C.c_void_p(self.h), C.c_int(frame), C.c_uint32(pt_idx), C.c_double(radius), C.c_int(max_pts), C.byref(k), C.byref(far), _p(nrm), C.byref(reg))
        return k.value, far.value, nrm, reg.value


    def plane_normal_sensitivity(self, frame, pt_idx, radius, max_pts):
        """how far the plane normal at a scan point moves when its covariance entries move by one ulp (8 random sign draws)"""
        sn = C.c_double(0)
        lib().oracle_plane_normal_sensitivity(C.c_void_p(self.h), C.c_int(frame), C.c_uint32(pt_idx), C.c_double(radius), C.c_int(max_pts), C.byref(sn))
        return sn.value


# ---- unit-level helpers for the known-answer tests ----
def sim3exp(x):
    x = np.ascontiguousarray(x, np.float64)
    R = np.zeros(9)
    t = np.zeros(3)
    s = C.c_double(0)
    lib().oracle_sim3exp(_p(x), _p(R), _p(t), C.byref(s))
    return R.reshape(3, 3), t, s.value


def sim3exp_jet(x):
    x = np.ascontiguousarray(x, np.float64)
    R = np.zeros(9)
    t = np.zeros(3)
    dR = np.zeros((9, 7))
    dt = np.zeros((3, 7))
    lib().oracle_sim3exp_jet(_p(x), _p(R), _p(t), _p(dR), _p(dt))
    return R.reshape(3, 3), t, dR, dt


def se3exp(x):
    x = np.ascontiguousarray(x, np.float64)
    R = np.zeros(9)
    t = np.zeros(3)
    lib().oracle_se3exp(_p(x), _p(R), _p(t))
    return R.reshape(3, 3), t


def se3log(R, t):
    R = np.ascontiguousarray(R, np.float64).reshape(9)
    t = np.ascontiguousarray(t, np.float64)
    out = np.zeros(6)
    lib().oracle_se3log(_p(R), _p(t), _p(out))
    return out


def covariance(pts, idx):
    pts = np.ascontiguousarray(pts, np.float64)
    idx = np.ascontiguousarray(idx, np.uint32)
    cov = np.zeros(9)
    lib().oracle_covariance(_p(pts), _p(idx), C.c_uint64(len(idx)), _p(cov))
    return cov.reshape(3, 3)


def fast_eigen(cov):
    cov = np.ascontiguousarray(cov, np.float64).reshape(9)
    v = np.zeros(3)
    ev = np.zeros(3)
    lib().oracle_fast_eigen(_p(cov), _p(v), _p(ev))
    return v, ev


def huber(a, s):
    r0 = C.c_double(0)
    r1 = C.c_double(0)
    lib().oracle_huber(C.c_double(a), C.c_double(s), C.byref(r0), C.byref(r1))
    return r0.value, r1.value


def max_threads():
    """host cores usable by this process (OMP_NUM_THREADS may be pinned to 1 by the launcher)"""
    import os
    try:
        return len(os.sched_getaffinity(0))
    except Exception:
        return os.cpu_count() or 1
