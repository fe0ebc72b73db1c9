import importlib, os, sys
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import lm_ref
pkg = importlib.import_module("spatial-temporal-lidar-camera-calibration_amd")
synth = importlib.import_module("spatial-temporal-lidar-camera-calibration_amd.synth")
abi = importlib.import_module("spatial-temporal-lidar-camera-calibration_amd.abi")
from oracle import binding as ob
ob.lib()
seed = int(sys.argv[1])
rng = np.random.default_rng(seed)
prob, meta = synth.make_scene(n_frames=int(rng.integers(4, 13)), pts_per_frame=int(rng.choice([2000, 4000, 8000])), n_keypoints=int(rng.choice([800, 2000])), seed=seed)
p = abi.reference_yaml_params()
h = pkg.IbaHandle(prob, p); o = ob.Oracle(prob)
amp = float(rng.choice([3e-4, 1e-3, 2e-3]))
x0 = synth.perturb(meta["x_gt"], rng, rot=amp, trans=10 * amp, scale_rel=3 * amp, n=1)[0]
trace = {"cpu": [], "gpu": []}
def ev_cpu(x):
    r = o.eval_factors(p, x)[0]; trace["cpu"].append((r.cost, tuple(x))); return r.H_np(), r.b_np(), r.cost
def ev_gpu(x):
    r = h.eval_factors(np.array([x]))[0]; trace["gpu"].append((r.cost, tuple(x))); return r.H_np(), r.b_np(), r.cost
xc, sc = lm_ref.calibrate_lm(x0, lambda x: o.build_problem(p, x), ev_cpu, max_outer=6)
xg2, sg2 = lm_ref.calibrate_lm(x0, lambda x: h.build_problem(x), ev_gpu, max_outer=6)
xg, rg = h.calibrate_lm(x0, max_outer_iterations=6)
print("cpu-LM(oracle)   cost", sc["final_cost"], "evals", sc["evals"])
print("numpy-LM(gpu ev) cost", sg2["final_cost"], "evals", sg2["evals"])
print("C++ LM (device)  cost", rg.final_cost, "evals", rg.evaluations)
n = min(len(trace["cpu"]), len(trace["gpu"]))
for i in range(n):
    a, b = trace["cpu"][i], trace["gpu"][i]
    dx = np.max(np.abs(np.array(a[1]) - np.array(b[1])))
    flag = "" if (abs(a[0] - b[0]) <= 1e-9 * abs(a[0]) and dx < 1e-12) else "   <--"
    print(i, f"{a[0]:.10g} {b[0]:.10g} dx={dx:.2e}{flag}")
    if flag and i > 0 and "<--" in flag:
        # dump the evaluation at the same x through both
        x = np.array(a[1])
        ro = o.eval_factors(p, x)[0]; rgp = h.eval_factors(np.array([x]))[0]
        print("  same x: oracle cost", ro.cost, "gpu cost", rgp.cost, "counts", ro.counts(), rgp.counts())
        print("  H dev", np.max(np.abs(ro.H_np() - rgp.H_np())) / np.max(np.abs(ro.H_np())), "b dev", np.max(np.abs(ro.b_np() - rgp.b_np())) / np.max(np.abs(ro.b_np())))
        break
print("se3 err numpyLM(gpu) vs cpu", lm_ref.se3_error(xg2, xc, synth.sim3_exp))
print("se3 err C++LM vs cpu", lm_ref.se3_error(xg, xc, synth.sim3_exp))
