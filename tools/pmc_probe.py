"""Small fixed workload for rocprofv3 --pmc runs: 3 cost evaluations of 64 candidates at the C2 shape."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("spatial-temporal-lidar-camera-calibration_amd")
synth = importlib.import_module("spatial-temporal-lidar-camera-calibration_amd.synth")
abi = importlib.import_module("spatial-temporal-lidar-camera-calibration_amd.abi")
prob, meta = synth.make_scene(n_frames=int(os.environ.get("PMC_FRAMES", "200")), pts_per_frame=int(os.environ.get("PMC_PTS", "10000")), seed=0)   # (PMC_FRAMES / PMC_PTS: another shape, tools/pmc_icache.sh)
h = pkg.IbaHandle(prob, abi.reference_yaml_params())
xs = synth.perturb(meta["x_gt"], np.random.default_rng(0), n=64)
mode = sys.argv[1] if len(sys.argv) > 1 else "cost"
for _ in range(int(os.environ.get("PMC_PROBE_ITERS", "3"))):
    (h.eval_cost if mode == "cost" else h.eval_full)(xs)
