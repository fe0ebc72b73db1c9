import importlib, os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
pkg = importlib.import_module("spatial-temporal-lidar-camera-calibration_amd"); synth = importlib.import_module("spatial-temporal-lidar-camera-calibration_amd.synth"); abi = importlib.import_module("spatial-temporal-lidar-camera-calibration_amd.abi")
prob, meta = synth.make_scene(n_frames=200, pts_per_frame=10000, seed=0)
h = pkg.IbaHandle(prob, abi.reference_yaml_params())
xs = synth.perturb(meta["x_gt"], np.random.default_rng(0), n=64)
for _ in range(3): h.eval_full(xs)
print("left to tree (entries):", h.nn_left_to_tree)
