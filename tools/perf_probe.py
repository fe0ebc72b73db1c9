"""Quick GPU perf probe of the cost path at a given shape (not the bench contract)."""
import importlib, sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("spatial-temporal-lidar-camera-calibration_amd")
synth = importlib.import_module("spatial-temporal-lidar-camera-calibration_amd.synth")
abi = importlib.import_module("spatial-temporal-lidar-camera-calibration_amd.abi")
F = int(sys.argv[1]) if len(sys.argv) > 1 else 200
P = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
t = time.time(); prob, meta = synth.make_scene(n_frames=F, pts_per_frame=P, seed=0); print("gen %.1fs" % (time.time() - t), flush=True)
t = time.time(); h = pkg.IbaHandle(prob, abi.reference_yaml_params()); print("create %.2fs" % (time.time() - t), flush=True)
h.set_timing(True)
rng = np.random.default_rng(0)
for B in (1, 2, 4, 8, 16, 32, 64):
    xs = synth.perturb(meta["x_gt"], rng, n=B)
    out = h.eval_cost(xs)
    reps = 10
    t = time.time()
    for _ in range(reps):
        out = h.eval_cost(xs)
    dt = (time.time() - t) / reps
    fk, tot = h.last_kernel_ms()
    print(f"B={B:3d} wall {dt*1e3:8.3f} ms  frame_kernel {fk:8.3f} ms total_dev {tot:8.3f} ms  -> {B/dt:9.0f} evals/s  (kernel-only {B/(fk*1e-3):9.0f}/s)  n_corr={out[0].n_corr} cnt3d3d={out[0].cnt_3d_3d}", flush=True)
if os.environ.get("IBA_LIB", "").endswith("stamps.so"):
    for B in (1, 16):
        xs = synth.perturb(meta["x_gt"], rng, n=B)
        h.eval_cost(xs)
        pp = h.debug_last_partials(B)
        st = pp[:, 56:63].mean(0) / F
        names = ["p0 init", "p1 project", "p2 ties", "p3 count", "p4a 3d2d", "p4b 3d3d", "p4c HE+reduce"]
        print(f"B={B} mean cycles per block:", ", ".join(f"{n} {v:.0f}" for n, v in zip(names, st)), " total", st.sum(), " [1a only:", int(pp[:, 63].mean() / F), "]")
    import ctypes
    L = pkg.load_library()
    z = (ctypes.c_ulonglong * 64)()
    L.iba_debug_counters(z, 1)
    xs = synth.perturb(meta["x_gt"], rng, n=16)
    h.eval_cost(xs)
    L.iba_debug_counters(z, 1)
    nb = max(1, z[2])
    print("per-wave search:", [int(z[8+w]/nb) for w in range(16)])
    print("per-wave pre   :", [int(z[24+w]/nb) for w in range(16)])
    print("per-wave post  :", [int(z[40+w]/nb) for w in range(16)])
    print("leaf-visit histogram [1,2,3,4,5-8,9-16,17-32,33+]:", [int(z[48+i]) for i in range(8)], "mean", z[56]/max(1,sum(z[48+i] for i in range(8))), "max", z[57])
    print(f"SEG wave0 per search call: descent {z[4]/max(1,z[7]):.0f}  leaf+reduce {z[5]/max(1,z[7]):.0f}  ascent-scan {z[6]/max(1,z[7]):.0f}  calls {z[7]}")
    print(f"wave0 cycles per block in nn_search: descent {z[4]/nb:.0f} leaf {z[5]/nb:.0f} reduce {z[6]/nb:.0f} ascent {z[7]/nb:.0f}")
    print(f"OLD per query: desc {z[4]/max(1,z[3]):.1f} asc {z[5]/max(1,z[3]):.1f} leaves {z[0]/max(1,z[3]):.2f}; per wave-call: max leaves {z[6]/(nb*16):.1f} max desc {z[7]/(nb*16):.1f}")
    print(f"per block: leaf visits {z[0]/nb:.0f}, queue candidates {z[1]/nb:.0f}, 3d-3d queries {z[3]/nb:.0f} -> leaf visits per query {z[0]/max(1,z[3]):.1f}")
