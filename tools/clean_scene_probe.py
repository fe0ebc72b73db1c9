"""final SE(3) on a noise-free scene: where does the iba_local LM end relative to the planted extrinsic, and why"""
import importlib, os, sys
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import lm_ref
pkg = importlib.import_module("spatial-temporal-lidar-camera-calibration_amd")
synth = importlib.import_module("spatial-temporal-lidar-camera-calibration_amd.synth")
abi = importlib.import_module("spatial-temporal-lidar-camera-calibration_amd.abi")
p = abi.reference_yaml_params()
for kw in (dict(kp_noise=0.0, range_noise=0.0), dict(kp_noise=0.0, range_noise=0.02), dict(kp_noise=0.5, range_noise=0.0)):
    prob, meta = synth.make_scene(n_frames=60, pts_per_frame=10000, n_keypoints=2000, seed=0, **kw)
    h = pkg.IbaHandle(prob, p)
    xg = meta["x_gt"]
    h.build_problem(xg)
    n = h.eval_factors(xg[None])[0]
    print(kw, "at x_gt: cost", n.cost, "chi2", n.chi2, "factors", n.n_factor_3d2d, n.n_factor_p2pl, n.n_factor_p2pt, "|b|", np.linalg.norm(n.b_np()))
    x0 = synth.perturb(xg, np.random.default_rng(5), rot=1e-3, trans=0.01, scale_rel=3e-3, n=1)[0]
    for opts in (dict(max_outer_iterations=10), dict(max_outer_iterations=30, min_diff=1e-9, function_tolerance=1e-14, parameter_tolerance=1e-12)):
        xf, r = h.calibrate_lm(x0, **opts)
        e = lm_ref.se3_error(xf, xg, synth.sim3_exp)
        print("   ", opts, "->", "outer", r.outer_iterations, "evals", r.evaluations, "cost %.6g -> %.6g" % (r.initial_cost, r.final_cost), "err rad/m", e, "scale", xf[6] - xg[6])
    h.close()
