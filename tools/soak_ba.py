"""Randomised parity soak of the ORB-only extrinsic BA (GPU box): device linearisation and the whole optimise / classify
schedule vs the CPU oracle over seeded random edge sets and starts. usage: python tools/soak_ba.py <n_scenes>"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
ba = importlib.import_module("spatial-temporal-lidar-camera-calibration_amd.ba")
from oracle import ba as oba
import ba_scene
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
bad = 0; t0 = time.time(); worst_x = 0.0
for sc in range(n):
    seed = 8000 + sc
    rng = np.random.default_rng(seed)
    prob, x_gt = ba_scene.make(n_frames=int(rng.integers(3, 40)), pts_per_frame=int(rng.choice([20, 100, 300])), seed=seed, ba=ba)
    h = ba.BaHandle(prob)
    N = len(prob.edge_frame)
    msgs = []
    for trial in range(3):
        x = x_gt + np.concatenate([rng.normal(0, 0.01, 3), rng.normal(0, 0.05, 3), [rng.normal(0, 0.3)]])
        active = None if trial == 0 else (rng.random(N) < 0.8).astype(np.uint8)
        robust = trial != 2
        H, b, chi, c2 = h.eval(x, active, robust)
        Ho, bo, chio, c2o = oba.evaluate(prob, x, active, robust)
        if not (np.allclose(c2, c2o, rtol=1e-10, atol=1e-12) and abs(chi - chio) <= 1e-10 * chio and
                np.allclose(H, Ho, rtol=1e-9, atol=1e-9 * np.abs(Ho).max()) and np.allclose(b, bo, rtol=1e-9, atol=1e-9 * np.abs(bo).max())):
            msgs.append(("lin", trial))
    x0 = x_gt + np.concatenate([rng.normal(0, 0.01, 3), rng.normal(0, 0.03, 3), [rng.normal(0, 0.2)]])
    x, r = h.optimize(x0)
    xo, n_in_o, log = oba.optimize(prob, x0)
    dx = float(np.max(np.abs(x - xo)))
    if r.n_inliers != n_in_o or [r.n_bad[i] for i in range(4)] != [l[1] for l in log]: msgs.append(("schedule counts", r.n_inliers, n_in_o))
    elif dx > 1e-6: msgs.append(("x", dx))
    else: worst_x = max(worst_x, dx)
    h.close(); bad += bool(msgs)
    print("BAD" if msgs else "ok ", seed, f"edges={N} inliers={r.n_inliers} dx={dx:.1e}", msgs[:3], flush=True)
print(f"{n - bad}/{n} BA problems in parity (linearisation 1e-9, inlier schedule exact, x within 1e-6; worst {worst_x:.1e}), {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
