"""Record the black-box calls of the global stage on the bench scene (200 keyframes x 10 k points, the start of bench.py's
global_then_local): every evaluated x in order and the size of every iba_eval_bbo batch -> gpurun_out/mads_trace.npz.
tools/mads_trace_stats.py then histograms the batches by spread / number of tight clusters (no GPU needed)."""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401  (one HIP runtime per process: torch's first)

PKG = "spatial-temporal-lidar-camera-calibration_amd"
pkg = importlib.import_module(PKG)
synth = importlib.import_module(PKG + ".synth")
abi = importlib.import_module(PKG + ".abi")

frames = int(os.environ.get("FRAMES", "200"))
prob, meta = synth.make_scene(n_frames=frames, pts_per_frame=10000, n_keypoints=2000, seed=0)
params = abi.reference_yaml_params()
h = pkg.IbaHandle(prob, params, device=0)
xg0 = meta["x_gt"] + np.array([0.009, -0.006, 0.005, 0.06, -0.04, 0.05, 0.4])
h.calibrate_mads(xg0, max_bb_eval=2000)   # warm
t0 = time.perf_counter()
xg, mr, tr, bs = h.calibrate_mads(xg0, record=True, max_bb_eval=100000)
dt = time.perf_counter() - t0
print("mads: %d evaluations in %d batches, %.3f s, feasible %d, f %.6f" % (mr.evaluations, mr.batches, dt, mr.feasible, mr.f))
assert bs.sum() == len(tr), (bs.sum(), len(tr))
# replay with the path and the kernel phases recorded (a fresh handle: every cross-call mechanism starts cold, as in the run itself)
import ctypes as C
h.close()
h = pkg.IbaHandle(prob, params, device=0)
h.set_timing(True)
L = pkg.load_library()
L.iba_last_phase_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]
paths, times, phases, anchors = [], [], [], []
o = pkg.mads_options(xg0)
at = 0
for b in bs:
    X = tr[at:at + b, :7]
    a0 = h.anchor_builds
    t1 = time.perf_counter()
    h.eval_bbo(X, o.he_threshold, o.valid_rate)
    times.append(time.perf_counter() - t1)
    pa, pn, pr = C.c_float(0), C.c_float(0), C.c_float(0)
    L.iba_last_phase_ms(h.h, C.byref(pa), C.byref(pn), C.byref(pr))
    phases.append((pa.value, pn.value, pr.value))
    paths.append(h.last_path)
    anchors.append(h.anchor_builds - a0)
    at += b
paths, times, phases, anchors = np.array(paths), np.array(times), np.array(phases), np.array(anchors)
print("replay: %.3f s (with phase events), %d anchor builds" % (times.sum(), anchors.sum()))
for pth in (0, 1, 2):
    m = paths == pth
    if m.any():
        print("  path %d: %4d batches, mean B %.1f, %.3f s, wall/batch %.3f ms, assoc %.3f ms, nn %.3f ms, rest %.3f ms" % (pth, m.sum(), bs[m].mean(), times[m].sum(), 1e3 * times[m].mean(), phases[m, 0].mean(), phases[m, 1].mean(), phases[m, 2].mean()))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "mads_trace.npz"), x=tr, batch_sizes=bs, path=paths, wall=times, phases=phases, anchors=anchors, x0=xg0, x_gt=meta["x_gt"], fx=718.856)
