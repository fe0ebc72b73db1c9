"""TEST INFRASTRUCTURE (oracle): ORB-only extrinsic BA of the reference on the CPU. The edge (calibEdge, Optimizer.cc:65-205)
comes from oracle/ba_oracle.cpp (forward-mode duals); the optimiser below restates g2o's Levenberg-Marquardt and the
reference's four optimise/classify rounds (Optimizer.cc:1511-1556) in plain numpy, independently of csrc/iba_ba.hip.
g2o is absent: PARITY WITH IT IS UNPINNED. Only tests/ may import this."""
import ctypes as C

import numpy as np

from . import binding

DELTA = float(np.sqrt(5.991))


class _Desc(C.Structure):
    _fields_ = [("n_edges", C.c_int64), ("n_frames", C.c_int32), ("frame_Tlw6", C.c_void_p), ("frame_intr", C.c_void_p), ("edge_frame", C.c_void_p),
                ("edge_Xw", C.c_void_p), ("edge_obs", C.c_void_p), ("edge_info", C.c_void_p), ("edge_slot", C.c_void_p)]


def _desc(p):
    d = _Desc()
    d.n_edges, d.n_frames = len(p.edge_frame), len(p.frame_Tlw6)
    for k in ("frame_Tlw6", "frame_intr", "edge_frame", "edge_Xw", "edge_obs", "edge_info", "edge_slot"):
        setattr(d, k, getattr(p, k).ctypes.data)
    return d


def edge(x, Xw, Tlw6, intr, obs):
    L = binding.lib()
    e, J = np.zeros(2), np.zeros(14)
    a = [np.ascontiguousarray(v, np.float64) for v in (x, Xw, Tlw6, intr, obs)]
    L.oracle_ba_edge(*[v.ctypes.data_as(C.c_void_p) for v in a], e.ctypes.data_as(C.c_void_p), J.ctypes.data_as(C.c_void_p))
    return e, J.reshape(2, 7)


def evaluate(p, x, active=None, robust=True):
    L = binding.lib()
    N = len(p.edge_frame)
    d = _desc(p)
    H, b, chi, chi2 = np.zeros(49), np.zeros(7), C.c_double(0), np.zeros(max(N, 1))
    x = np.ascontiguousarray(x, np.float64)
    act = None if active is None else np.ascontiguousarray(active, np.uint8)
    L.oracle_ba_eval(C.byref(d), x.ctypes.data_as(C.c_void_p), None if act is None else act.ctypes.data_as(C.c_void_p), C.c_int(1 if robust else 0), C.c_double(DELTA),
                     H.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), C.byref(chi), chi2.ctypes.data_as(C.c_void_p))
    return H.reshape(7, 7), b, chi.value, chi2[:N]


def optimize(p, x0, evaluate_fn=None):
    """The reference's schedule; evaluate_fn(x, active, robust) -> (H, b, chi, chi2_edges) (default: the CPU oracle)."""
    ev = evaluate_fn or (lambda x, a, r: evaluate(p, x, a, r))
    N = len(p.edge_frame)
    x0 = np.asarray(x0, np.float64)
    active = np.ones(N, np.uint8)
    stored = np.zeros(N)
    flags = np.zeros(int(p.edge_slot.max()) + 1 if N else 0, np.uint8)
    robust, nbad, x = True, 0, x0.copy()
    log = []
    for rnd in range(4):
        x = x0.copy()                                  # v->setEstimate(p_tcl)
        lam, ni = 0.0, 2.0
        for it in range(10):                           # optimizer.optimize(10)
            H, b, cur, _ = ev(x, active, robust)
            if it == 0:
                lam, ni = 1e-5 * np.abs(np.diag(H)).max(), 2.0
            rho, q, finite = 0.0, 0, True
            while True:
                try:
                    dx = np.linalg.solve(H + lam * np.eye(7), b)
                    np.linalg.cholesky(H + lam * np.eye(7))
                    tmp = ev(x + dx, active, robust)[2]
                    scale = float(dx @ (lam * dx + b)) + 1e-3
                except np.linalg.LinAlgError:
                    tmp, scale = np.finfo(np.float64).max, 1e-3
                rho = (cur - tmp) / scale
                if rho > 0 and np.isfinite(tmp):
                    lam *= max(1.0 / 3.0, min(1.0 - (2 * rho - 1) ** 3, 2.0 / 3.0))
                    ni, cur, x = 2.0, tmp, x + dx
                else:
                    lam *= ni
                    ni *= 2
                    if not np.isfinite(lam):
                        finite = False
                        break
                q += 1
                if not (rho < 0 and q < 10):
                    break
            if q == 10 or rho == 0 or not finite:
                break
        _, _, chi, now = ev(x, active, robust)
        nbad = 0
        for i in range(N):                             # sequential on purpose: the flags alias across keyframes
            idx = p.edge_slot[i]
            if active[i] or flags[idx]:
                stored[i] = now[i]
            if np.float32(stored[i]) > np.float32(5.991):
                flags[idx], active[i] = 1, 0
                nbad += 1
            else:
                flags[idx], active[i] = 0, 1
        log.append((chi, nbad))
        if rnd == 2:
            robust = False
        if N < 10:
            break
    return x, N - nbad, log
