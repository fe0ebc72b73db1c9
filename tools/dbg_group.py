"""debug: group create / eval / destroy in a loop, with faulthandler (which Python line aborts)"""
import faulthandler, importlib, os, sys
faulthandler.enable(all_threads=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
PKG = "spatial-temporal-lidar-camera-calibration_amd"
pkg = importlib.import_module(PKG); synth = importlib.import_module(PKG + ".synth"); abi = importlib.import_module(PKG + ".abi")
mode = sys.argv[1] if len(sys.argv) > 1 else "rccl"
if mode == "torch":
    import torch
    torch.cuda.set_device(0)
    mode = "rccl"
prob, meta = synth.make_scene(n_frames=12, pts_per_frame=4000, seed=1)
p = abi.reference_yaml_params()
xs = synth.perturb(meta["x_gt"], np.random.default_rng(0), n=8)
print(pkg.rccl_info(), flush=True)
for it in range(3):
    g = pkg.IbaGroup(prob, p, devices=(0,) if mode == "rccl" else (0, 0), host_reduce=(mode != "rccl"))
    print("created", it, flush=True)
    c, n = g.eval_full(xs)
    print("evaluated", it, c[0].f1, flush=True)
    g.close()
    print("closed", it, flush=True)
print("done", flush=True)
