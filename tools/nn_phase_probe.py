"""GPU box, diag build (make -C csrc diag; IBA_LIB=<package dir>/libiba_diag.so). Cycles per phase of iba_nn_kernel at the bench shape: thread 0 of every block
(iba_debug_phase_cycles). usage: python tools/nn_phase_probe.py [frames] [pts] [B]"""
import importlib, os, sys
import ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("spatial-temporal-lidar-camera-calibration_amd"); synth = importlib.import_module("spatial-temporal-lidar-camera-calibration_amd.synth"); abi = importlib.import_module("spatial-temporal-lidar-camera-calibration_amd.abi")
F = int(sys.argv[1]) if len(sys.argv) > 1 else 200
P = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
prob, meta = synth.make_scene(n_frames=F, pts_per_frame=P, seed=0)
h = pkg.IbaHandle(prob, abi.reference_yaml_params())
xs = synth.perturb(meta["x_gt"], np.random.default_rng(0), n=B)
L = pkg.load_library()
L.iba_debug_phase_cycles12.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int32]
out = (C.c_uint64 * 12)()
for _ in range(3): h.eval_full(xs)
L.iba_debug_phase_cycles12(h.h, out, 1)
h.eval_full(xs)
L.iba_debug_phase_cycles12(h.h, out, 1)
v = [int(x) for x in out]
if os.environ.get("IBA_NN_LIST", "1") != "0" and B >= 6:
    names = ["start-up + between items", "item scalars, entries two items ahead", "picks (with their waits for the row)", "issuing the next rows", "item barrier", "sums + record", "tree searches"]
else:
  names = ["start-up (list lengths, candidate constants, first barrier)", "entries + MapPoints of a step", "list rows of a step", "picks", "wait for the block's other waves", "left-over searches", "sums + records"]
tot = sum(v[:7]) + sum(v[8:12])
print("F %d x %d pts, B %d: %d blocks reached the end; cycles of thread 0 per block, mean %.0f" % (F, P, B, v[7], tot / max(v[7], 1)))
for n, c in zip(names, v[:7]):
    print("   %-62s %8.0f  (%.1f %%)" % (n, c / max(v[7], 1), 100.0 * c / max(tot, 1)))
if sum(v[8:12]):
    for n, c in zip(["pick: queries", "pick: certificate (first use of the row)", "pick: nearest listed point", "pick: results"], v[8:12]):
        print("   %-62s %8.0f  (%.1f %%)" % (n, c / max(v[7], 1), 100.0 * c / max(tot, 1)))
