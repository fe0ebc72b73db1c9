"""GPU box. "factors + sums" time (library event timing) of the chain cost + normal equations and of the frozen-problem chain, by scene shape
and batch size: iba_factor2_kernel (default) (IBA_FACTOR_V2=1) vs the one-wave-per-keyframe kernel (default). usage: python tools/factor_shapes.py"""
import importlib, sys, os
import ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("spatial-temporal-lidar-camera-calibration_amd")
synth = importlib.import_module("spatial-temporal-lidar-camera-calibration_amd.synth")
abi = importlib.import_module("spatial-temporal-lidar-camera-calibration_amd.abi")
L = pkg.load_library()
L.iba_last_phase_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]
def rest(h):
    a, n, r = C.c_float(0), C.c_float(0), C.c_float(0)
    L.iba_last_phase_ms(h.h, C.byref(a), C.byref(n), C.byref(r))
    return r.value
def med(h, fn, xs):
    ts = []
    for _ in range(7):
        fn(xs); ts.append(rest(h))
    return float(np.median(ts[2:]))
for F, P in ((200, 10000), (25, 10000), (40, 120000)):
    prob, meta = synth.make_scene(n_frames=F, pts_per_frame=P, seed=0)
    hs = {}
    os.environ["IBA_DEBUG_ENV"] = "1"
    for name, v2 in (("factor2", "1"), ("v1", "0")):
        os.environ["IBA_FACTOR_V2"] = v2
        hs[name] = pkg.IbaHandle(prob, abi.reference_yaml_params())
        hs[name].set_timing(True)
    os.environ.pop("IBA_FACTOR_V2")
    for B in [int(a) for a in sys.argv[1:]] or (1, 4, 14, 64, 256, 512):
        xs = synth.perturb(meta["x_gt"], np.random.default_rng(0), n=B)
        line = "F %3d x %6d pts, B %3d:" % (F, P, B)
        for name, h in hs.items():
            line += "  %s full %.4f ms" % (name, med(h, h.eval_full, xs))
        for name, h in hs.items():
            h.build_problem(xs[0])
            line += "  %s frozen %.4f ms" % (name, med(h, h.eval_factors, xs))
        print(line, flush=True)
    for h in hs.values():
        h.close()
