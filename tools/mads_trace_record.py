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
# replay with the path recorded
paths, times = [], []
o = pkg.mads_options(xg0)
at = 0
for b in bs:
    X = tr[at:at + b, :7]
    t1 = time.perf_counter()
    h.eval_bbo(X, o.he_threshold, o.valid_rate)
    times.append(time.perf_counter() - t1)
    paths.append(h.last_path)
    at += b
paths, times = np.array(paths), np.array(times)
print("replay: %.3f s, %d of %d batches shared their pair search; time share of the others %.2f" % (times.sum(), paths.sum(), len(paths), times[paths == 0].sum() / times.sum()))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "mads_trace.npz"), x=tr, batch_sizes=bs, path=paths, wall=times, x0=xg0, x_gt=meta["x_gt"], fx=718.856)
