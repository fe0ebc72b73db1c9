"""kernel times (HIP events of the library) around 64 candidates per call, per call and per candidate: is a batch size a bad one for the block -> XCD mapping?
(r05: 64 candidates 0.170 ms of association, 65 candidates 0.166 ms; an odd number of blocks per keyframe in iba_assoc2_kernel took 64 to 0.167 — inside the
noise between boxes, not kept. The search kernel has such a pathology for real: a power of two of blocks per keyframe, see nn_ns in csrc/iba_capi.hip.)"""
import importlib, os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
PKG = "spatial-temporal-lidar-camera-calibration_amd"
pkg = importlib.import_module(PKG); synth = importlib.import_module(PKG + ".synth"); abi = importlib.import_module(PKG + ".abi")
prob, meta = synth.make_scene(n_frames=200, pts_per_frame=10000, seed=0)
h = pkg.IbaHandle(prob, abi.reference_yaml_params())
h.set_timing(True)
L = pkg.load_library()
L.iba_last_phase_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]
def phases():
    a, n, r = C.c_float(0), C.c_float(0), C.c_float(0)
    L.iba_last_phase_ms(h.h, C.byref(a), C.byref(n), C.byref(r))
    return a.value, n.value, r.value
for B in (56, 60, 63, 64, 65, 72, 80):
    xs = synth.perturb(meta["x_gt"], np.random.default_rng(0), n=B)
    ts = []
    for _ in range(7):
        h.eval_full(xs); ts.append(phases())
    t = np.median(np.array(ts[2:]), axis=0)
    print("B=%3d  assoc %.4f  nn %.4f  rest %.4f   per candidate: assoc %.3f us  nn %.3f us  rest %.3f us" % (B, t[0], t[1], t[2], 1e3 * t[0] / B, 1e3 * t[1] / B, 1e3 * t[2] / B), flush=True)
