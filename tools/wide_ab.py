"""a batch as wide as the reference's search box (every candidate searches for itself: iba_assoc_kernel) — wall per 64 candidates, median of 30; IBA_LIB selects the build"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
PKG = "spatial-temporal-lidar-camera-calibration_amd"
pkg = importlib.import_module(PKG); synth = importlib.import_module(PKG + ".synth"); abi = importlib.import_module(PKG + ".abi")
prob, meta = synth.make_scene(n_frames=200, pts_per_frame=10000, seed=0)
h = pkg.IbaHandle(prob, abi.reference_yaml_params())
xw = meta["x_gt"][None, :] + np.random.default_rng(7).uniform(-1, 1, (64, 7)) * np.array([0.1, 0.1, 0.1, 0.3, 0.3, 0.3, 1.0])
for _ in range(3): h.eval_full(xw)
ts = []
for _ in range(30):
    t0 = time.perf_counter(); h.eval_full(xw); ts.append(time.perf_counter() - t0)
print("%s wide batch: %.3f ms per 64 (%.0f evals/s), path %d" % (os.environ.get("IBA_LIB", "default").split("/")[-1], np.median(ts) * 1e3, 64 / np.median(ts), h.last_path), flush=True)
