"""Phase breakdown of the fused frame kernel (diagnostic `make stamps` build, IBA_LIB=...stamps.so)."""
import importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("spatial-temporal-lidar-camera-calibration_amd")
synth = importlib.import_module("spatial-temporal-lidar-camera-calibration_amd.synth")
abi = importlib.import_module("spatial-temporal-lidar-camera-calibration_amd.abi")
F = int(sys.argv[1]) if len(sys.argv) > 1 else 200
P = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
prob, meta = synth.make_scene(n_frames=F, pts_per_frame=P, seed=0)
h = pkg.IbaHandle(prob, abi.reference_yaml_params())
h.set_timing(True)
rng = np.random.default_rng(0)
xs = synth.perturb(meta["x_gt"], rng, n=B)
for _ in range(3):
    h.eval_full(xs)
fk, tot = h.last_kernel_ms()
print(f"B={B} frame kernel {fk:.3f} ms, total device {tot:.3f} ms")
pp = h.debug_last_partials(B)
st = pp[:, 56:64].mean(0) / F
names = ["p0 init", "p1 project(1a+1b)", "p2 ties", "p3 count", "f4 list+plane+3d2d", "f5 NN", "f6 refit", "f7 finalize"]
print("mean cycles per block:", ", ".join(f"{n} {v:.0f}" for n, v in zip(names, st[:7])), "| total", st[:7].sum(), "| 1a only", st[7])

import ctypes
if not os.environ.get("IBA_FINE"): sys.exit(0)
L = pkg.load_library()
z = (ctypes.c_ulonglong * 64)()
L.iba_debug_counters(z, 1)
h.eval_full(xs)
L.iba_debug_counters(z, 1)
nb = max(1, z[24])
print("blocks", nb)
for r in range(8):
    if z[24 + r]:
        print(f"  round {r:2d}: blocks reaching {z[24+r]/nb:6.3f}  items/block(reaching) {z[8+r]/z[24+r]:8.2f}  cycles/block(reaching) {z[16+r]/z[24+r]:8.0f}  cycles/block(all) {z[16+r]/nb:8.0f}")
for nm, o in (("fresh", 48), ("resume", 56)):
    calls = max(1, z[o + 6]); lv = max(1, z[o + 5])
    print(f"thread0 {nm}: calls {z[o+6]} leaves {z[o+5]} | per call: prologue {z[o]/calls:.0f} total {z[o+7]/calls:.0f} | per leaf: enter+descent {z[o+1]/lv:.0f} leaf {z[o+2]/lv:.0f} reduce {z[o+3]/lv:.0f} select {z[o+4]/lv:.0f}")

for nm, o in (("fresh", 32), ("resume", 36)):
    n = max(1, z[o + 3])
    print(f"thread0 round body {nm}: passes {z[o+3]} | queries {z[o]/n:.0f}  step+park {z[o+1]/n:.0f}  wait-at-barrier {z[o+2]/n:.0f}")

if os.environ.get("IBA_HIST"):
    tot = max(1, sum(z[40 + i] for i in range(8)))
    print("visits per query [1,2,3,4,5-8,9-16,17-32,33+]:", [round(z[40 + i] / tot, 4) for i in range(8)], "mean", z[6] / tot, "queries/block", tot / nb)
    tb = max(1, sum(z[56 + i] for i in range(8)))
    print("max visits per block [1,2,3,4,5-8,9-16,17-32,33+]:", [round(z[56 + i] / tb, 4) for i in range(8)], "mean", z[7] / tb)
