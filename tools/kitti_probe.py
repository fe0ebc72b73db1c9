"""the reference's scan size (120 k points per scan) on 40 keyframes, 64 candidates: evaluations for rocprofv3 / timing
usage: python tools/kitti_probe.py [iters]"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
PKG = "spatial-temporal-lidar-camera-calibration_amd"
pkg = importlib.import_module(PKG); synth = importlib.import_module(PKG + ".synth"); abi = importlib.import_module(PKG + ".abi")
kf = int(os.environ.get("KITTI_KF", "40"))
prob, meta = synth.make_scene(n_frames=kf, pts_per_frame=int(os.environ.get("KITTI_PTS", "120000")), seed=0)
tile = int(os.environ.get("KITTI_TILE", "1"))   # KITTI_TILE=5: the whole 200 keyframes on the device (the 40 ray-cast ones five times)
if tile > 1:
    prob = synth.tile_scene(prob, meta, tile)[0]
    kf *= tile
h = pkg.IbaHandle(prob, abi.reference_yaml_params())
xs = synth.perturb(meta["x_gt"], np.random.default_rng(0), n=int(os.environ.get("KITTI_B", "64")))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for _ in range(3): h.eval_full(xs)
ts = []
for _ in range(n):
    t0 = time.perf_counter(); c, _n = h.eval_full(xs); ts.append(time.perf_counter() - t0)
print("kitti shape %d KF: %.3f ms per call, n_corr %.0f, cnt_3d_3d %.0f, anchor builds %d" % (kf, np.median(ts) * 1e3, np.mean([a.n_corr for a in c]), np.mean([a.cnt_3d_3d for a in c]), h.anchor_builds), flush=True)
print("pairs per keyframe: %.0f" % h.mean_pairs, flush=True)
print("entries left to the tree search: %.0f of ~%.0f wanted (all candidates)" % (h.nn_left_to_tree, sum(a.cnt_3d_3d for a in c)), flush=True)
