"""B = 64 only: per-kernel time of the two-kernel chain for the library selected by IBA_LIB."""
import importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
pkg = importlib.import_module("spatial-temporal-lidar-camera-calibration_amd")
synth = importlib.import_module("spatial-temporal-lidar-camera-calibration_amd.synth")
abi = importlib.import_module("spatial-temporal-lidar-camera-calibration_amd.abi")
prob, meta = synth.make_scene(n_frames=200, pts_per_frame=10000, seed=0)
h = pkg.IbaHandle(prob, abi.reference_yaml_params())
h.set_timing(True)
L = pkg.load_library()
L.iba_last_phase_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]
def phases():
    a, n, r = C.c_float(0), C.c_float(0), C.c_float(0)
    L.iba_last_phase_ms(h.h, C.byref(a), C.byref(n), C.byref(r))
    return a.value, n.value, r.value
tag = "%s CG=%s NS=%s" % (os.environ.get("IBA_LIB", "default").split("/")[-1], os.environ.get("IBA_NN_CG", "-"), os.environ.get("IBA_NN_NS", "-"))
xs = synth.perturb(meta["x_gt"], np.random.default_rng(0), n=64)
for mode, fn in (("full", h.eval_full), ("cost", h.eval_cost)):
    ts = []
    for _ in range(6):
        fn(xs)
        ts.append(phases())
    t = np.median(np.array(ts[1:]), axis=0)
    print("%s %s: assoc %.3f ms  nn %.3f ms  rest %.3f ms  sum %.3f ms" % (tag, mode, t[0], t[1], t[2], t.sum()), flush=True)
