"""wall time per call at small batch sizes (cost + normal equations, cost alone, frozen factors), median of 200 calls"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (one HIP runtime per process: torch's first)
PKG = "spatial-temporal-lidar-camera-calibration_amd"
pkg = importlib.import_module(PKG); synth = importlib.import_module(PKG + ".synth"); abi = importlib.import_module(PKG + ".abi")
prob, meta = synth.make_scene(n_frames=200, pts_per_frame=10000, seed=0)
h = pkg.IbaHandle(prob, abi.reference_yaml_params())
rng = np.random.default_rng(0)
h.build_problem(meta["x_gt"])
for B in (1, 8, 14, 64):
    xs = synth.perturb(meta["x_gt"], rng, n=B)
    for name, fn in (("full", h.eval_full), ("cost", h.eval_cost), ("factors", h.eval_factors)):
        for _ in range(5):
            fn(xs)
        ts = []
        for _ in range(200):
            t0 = time.perf_counter(); fn(xs); ts.append(time.perf_counter() - t0)
        med_c, min_c = h.call_latency(xs, name, 200)   # the same calls in a C loop (iba_debug_call_latency): no Python in the clock
        print("B=%2d %-8s median %.3f ms  (%.0f evals/s)   from C: median %.3f ms, min %.3f ms (%.0f evals/s)" % (B, name, np.median(ts) * 1e3, B / np.median(ts), med_c, min_c, B / (med_c * 1e-3)), flush=True)
