import importlib, sys, os, time
import numpy as np
sys.path.insert(0, os.getcwd())
pkg = importlib.import_module("spatial-temporal-lidar-camera-calibration_amd")
synth = importlib.import_module("spatial-temporal-lidar-camera-calibration_amd.synth")
abi = importlib.import_module("spatial-temporal-lidar-camera-calibration_amd.abi")
prob, meta = synth.make_scene(n_frames=30, pts_per_frame=6000, n_keypoints=1000, seed=4)
x_gt = meta["x_gt"]
rng = np.random.default_rng(1)
x0 = x_gt + np.concatenate([rng.normal(0, 0.01, 3), rng.normal(0, 0.05, 3), [0.4]])
h = pkg.IbaHandle(prob, abi.reference_yaml_params())
lb = x0 + np.array([-0.1,-0.1,-0.1,-0.3,-0.3,-0.3,-1.0]); ub = x0 + np.array([0.1,0.1,0.1,0.3,0.3,0.3,1.0])
frame0 = 0.1 * (ub - lb)
t = time.time()
best, r = h.calibrate_mads(x0, max_bb_eval=3000, lb=lb, ub=ub)
bf = r.f; total = r.evaluations
print("first descent f=%.4f evals=%d x-gt=%s" % (bf, total, np.round(best - x_gt, 4)))
rs = np.random.default_rng(7)
k = 1
for it in range(40):
    u = rs.uniform(-1, 1, 7); u /= np.abs(u).max()
    start = np.clip(best + k * 0.5 * frame0 * u, lb, ub)
    x, r = h.calibrate_mads(start, max_bb_eval=1500, lb=lb, ub=ub, restarts=0)
    total += r.evaluations
    if r.feasible and r.f < bf:
        best, bf, k = x, r.f, 1
        print("it %d k accepted f=%.4f evals=%d x-gt=%s" % (it, bf, total, np.round(best - x_gt, 4)))
    else:
        k = min(k + 1, 6)
print("final f=%.4f total evals=%d time %.2fs x-gt=%s" % (bf, total, time.time() - t, np.round(best - x_gt, 4)))
