"""GPU box. VERDICT r5 #8, measured before built: how much of the optimiser's WIDE batches (the ones that take the per-candidate association because they do not
cluster into <= 4 tight groups) is a tight subset plus a few far candidates? Records the black-box calls of the bench's MADS run, and for every batch the host
planner (iba_debug_plan_groups: the planner of the library itself) rejects, finds the largest prefix — candidates ordered by distance from the batch's median —
that the planner accepts. A mixed dispatch (tight part on the shared pair search, the rest per candidate) can only win on those prefixes.
usage: python tools/mixed_dispatch_probe.py"""
import ctypes as C, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("spatial-temporal-lidar-camera-calibration_amd"); synth = importlib.import_module("spatial-temporal-lidar-camera-calibration_amd.synth"); abi = importlib.import_module("spatial-temporal-lidar-camera-calibration_amd.abi")
prob, meta = synth.make_scene(n_frames=200, pts_per_frame=10000, seed=0)
h = pkg.IbaHandle(prob, abi.reference_yaml_params())
xg0 = meta["x_gt"] + np.array([0.009, -0.006, 0.005, 0.06, -0.04, 0.05, 0.4])
xg, mr, mtr, mbs = h.calibrate_mads(xg0, record=True, max_bb_eval=100000)
L = h.lib
L.iba_debug_plan_groups.argtypes = [C.c_void_p, C.c_int32, C.c_double, C.c_double, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
max_fx = float(np.max(prob.arrays["fx"])) if "fx" in prob.arrays else 718.856


def groups(x):
    x = np.ascontiguousarray(x, np.float64)
    go = np.zeros(len(x), np.int32); gp = np.zeros(8); ng = C.c_int32(0)
    st = L.iba_debug_plan_groups(x.ctypes.data, len(x), max_fx, 20.0, 4, go.ctypes.data, gp.ctypes.data, C.byref(ng))
    return ng.value if st == 0 else 0


at = 0
wide_b = wide_e = covered_e = 0
hist = {}
for nb in mbs:
    nb = int(nb)
    x = mtr[at:at + nb, :7]; at += nb
    if nb < 5 or groups(x) > 0:
        continue
    wide_b += 1; wide_e += nb
    med = np.median(x, axis=0)
    d = np.max(np.abs(x - med) * np.array([12, 12, 12, 1, 1, 1, 0.1]), axis=1)
    order = np.argsort(d)
    lo, hi = 0, nb   # largest prefix the planner accepts (>= 4 candidates to be worth a pair search)
    while lo < hi:
        mid = (lo + hi + 1) // 2
        if mid >= 4 and groups(x[order[:mid]]) > 0: lo = mid
        else: hi = mid - 1
    covered_e += lo
    k = int(10 * lo / nb)
    hist[k] = hist.get(k, 0) + 1
print("MADS run: %d evaluations in %d batches; wide batches (>= 5 candidates, no plan): %d with %d evaluations" % (at, len(mbs), wide_b, wide_e))
print("largest plannable prefix: %d of those evaluations (%.1f %%)" % (covered_e, 100.0 * covered_e / max(wide_e, 1)))
print("wide batches by covered tenth:", {("%d0-%d9 %%" % (k, k)): v for k, v in sorted(hist.items())})
