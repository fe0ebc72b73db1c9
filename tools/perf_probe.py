"""Quick GPU perf probe of the cost path at a given shape (not the bench contract)."""
import importlib, sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("spatial-temporal-lidar-camera-calibration_amd")
synth = importlib.import_module("spatial-temporal-lidar-camera-calibration_amd.synth")
abi = importlib.import_module("spatial-temporal-lidar-camera-calibration_amd.abi")
F = int(sys.argv[1]) if len(sys.argv) > 1 else 200
P = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
t = time.time(); prob, meta = synth.make_scene(n_frames=F, pts_per_frame=P, seed=0); print("gen %.1fs" % (time.time() - t), flush=True)
t = time.time(); h = pkg.IbaHandle(prob, abi.reference_yaml_params()); print("create %.2fs" % (time.time() - t), flush=True)
h.set_timing(True)
rng = np.random.default_rng(0)
for B in (1, 2, 4, 8, 16, 32, 64):
    xs = synth.perturb(meta["x_gt"], rng, n=B)
    out = h.eval_cost(xs)
    reps = 10
    t = time.time()
    for _ in range(reps):
        out = h.eval_cost(xs)
    dt = (time.time() - t) / reps
    fk, tot = h.last_kernel_ms()
    print(f"B={B:3d} wall {dt*1e3:8.3f} ms  frame_kernel {fk:8.3f} ms total_dev {tot:8.3f} ms  -> {B/dt:9.0f} evals/s  (kernel-only {B/(fk*1e-3):9.0f}/s)  n_corr={out[0].n_corr} cnt3d3d={out[0].cnt_3d_3d}", flush=True)
