import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
PKG = "spatial-temporal-lidar-camera-calibration_amd"
pkg = importlib.import_module(PKG); synth = importlib.import_module(PKG + ".synth"); abi = importlib.import_module(PKG + ".abi")
prob, meta = synth.make_scene(n_frames=200, pts_per_frame=10000, seed=0)
h = pkg.IbaHandle(prob, abi.reference_yaml_params())
rng = np.random.default_rng(0)
for B in (24, 32, 38, 48):
    for rot in (1e-4, 5e-4):
        xs = synth.perturb(meta["x_gt"], rng, rot=rot, trans=10 * rot, scale_rel=2 * rot, n=B)
        for _ in range(3): h.eval_cost(xs)
        ts = []
        for _ in range(40):
            t0 = time.perf_counter(); h.eval_cost(xs); ts.append(time.perf_counter() - t0)
        print("B=%d rot %.0e: %.3f ms (pairs %.0f)" % (B, rot, np.median(ts) * 1e3, h.mean_pairs), flush=True)
