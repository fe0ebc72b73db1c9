"""bench.py — IBA residual+Jacobian evaluations/sec on the BASELINE.json workload.

One "step" = one batch of B candidate extrinsics x in R^7, each evaluated against every keyframe the
job holds, producing BOTH the BAError tuple (iba_global.cpp:169-344) AND the Gauss-Newton normal
equations of the iba_local problem re-associated at x (iba_local.cpp:145-323 + IBACalib2.hpp factors).
value = evaluations/s = N_ranks-wide: all ranks evaluate the same B candidates on their own frames
(frames shard across GPUs, weak scaling: 200 keyframes x 10k points per GPU), one sum all-reduce of
the partial blocks per call.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = "spatial-temporal-lidar-camera-calibration_amd"

FRAMES_PER_GPU = 200
PTS_PER_FRAME = 10000
KEYPOINTS = 2000
BATCH = 64
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def algorithmic_bytes(n_points, n_frames, n_keypoints, n_corr, n_corr3, n_covis):
    """minimum traffic of ONE evaluation of the reference's formulation with float32
    points, each datum touched once."""
    return (12.0 * n_points + 8.0 * n_keypoints + 8.0 * n_corr + n_corr3 * (12.0 * 31 + 24) + n_corr * n_covis * 12.0
            + n_frames * (256.0 + 96.0 * n_covis) + 640.0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--frames", type=int, default=FRAMES_PER_GPU)
    ap.add_argument("--pts", type=int, default=PTS_PER_FRAME)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    pkg = importlib.import_module(PKG)
    synth = importlib.import_module(PKG + ".synth")
    abi = importlib.import_module(PKG + ".abi")

    # ---- workload: configs[1] shape per GPU (200 KF x 10k pts = 2M points), tiled once per rank ----
    base, meta = synth.make_scene(n_frames=args.frames, pts_per_frame=args.pts, n_keypoints=KEYPOINTS, seed=0)
    prob = base if world == 1 else synth.tile_scene(base, meta, world)[0]
    params = abi.reference_yaml_params()
    f0, f1 = rank * args.frames, (rank + 1) * args.frames
    h = pkg.IbaHandle(prob, params, device=local_rank, frame_begin=f0, frame_end=f1)
    h.set_timing(True)
    stride = pkg.partial_stride()
    B = args.batch
    rng = np.random.default_rng(0)
    xs_all = [synth.perturb(meta["x_gt"], rng, n=B) for _ in range(4)]   # x0 +- seeded perturbations
    d_cost = torch.zeros(B * stride, dtype=torch.float64, device=dev)
    d_norm = torch.zeros(B * stride, dtype=torch.float64, device=dev)

    def step(i):
        xs = xs_all[i % len(xs_all)]
        st = torch.cuda.current_stream().cuda_stream
        h.eval_full_partial(xs, d_cost.data_ptr(), st)   # cost tuple + normal equations from one pass over the scans
        if world > 1:   # frames shard across ranks: ONE sum all-reduce of the partial blocks (RCCL over xGMI)
            dist.all_reduce(d_cost)
        pc = d_cost.cpu().numpy()
        return pkg.finalize_cost(params, pc), pkg.finalize_normal(params, pc)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        out = step(i)
    sync()
    t0 = time.perf_counter()
    kms = []
    for i in range(args.steps):
        out = step(i)
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- dominant kernel (fused frame kernel), timed with HIP events on its launch stream ----
    xs = xs_all[0]
    kms = []
    for _ in range(5):
        h.eval_full_partial(xs, d_cost.data_ptr(), torch.cuda.current_stream().cuda_stream)
        kms.append(h.last_kernel_ms()[0])
    torch.cuda.synchronize()
    frame_ms = float(np.median(kms))
    cost0 = pkg.finalize_cost(params, d_cost.cpu().numpy())
    if world > 1:
        pass  # d_cost here holds this rank's partial only; counts below are per-rank (what one launch processes)
    n_slots = len(prob.arrays["covis_frame"]) / prob.n_frames
    per_eval = np.mean([algorithmic_bytes(h.n_points, args.frames, h.n_keypoints, c.n_corr, c.cnt_3d_3d, n_slots) for c in cost0])
    achieved = B * per_eval / (frame_ms * 1e-3) / 1e9

    evals = B * args.steps
    value = evals / dt
    res = {
        "metric": "IBA residual+Jacobian evals/sec",
        "value": value,
        "unit": "evals/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": "configs[1]-shaped synthetic street scene per GPU: %d keyframes x %d pts (%.1fM pts), %d keypoints/KF, IBACalib cost (3d-2d + 3d-3d plane + hand-eye) + iba_local normal equations per candidate"
                        % (args.frames, args.pts, args.frames * args.pts / 1e6, KEYPOINTS),
            "frames_per_gpu": args.frames, "points_per_frame": args.pts, "keypoints_per_frame": KEYPOINTS,
            "candidates_per_step": B, "total_frames": prob.n_frames, "total_points": prob.n_points,
            "plane_cache": int(params.plane_cache),
            "mean_n_corr": float(np.mean([c.n_corr for c in out[0]])), "mean_cnt_3d_3d": float(np.mean([c.cnt_3d_3d for c in out[0]])),
            "mean_factors": float(np.mean([n.n_factor_3d2d + n.n_factor_p2pl + n.n_factor_p2pt for n in out[1]])),
            "parallelism": "frames sharded over %d GPU(s), 1 all-reduce of %d doubles per call" % (world, B * stride),
        },
        "roofline": {
            "bound": "hbm", "kernel": "iba_frame_kernel<MODE_BOTH>", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": None,
            "algorithmic_bytes_per_eval": per_eval, "evals_per_launch": B, "launch_ms": frame_ms,
        },
    }
    tfile = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if os.path.exists(tfile):
        try:
            res["roofline"]["traffic"] = json.load(open(tfile)).get("hbm_bytes_per_launch")
        except Exception:
            pass

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # ---- CPU baseline: the oracle (a port of the reference algorithm) on this box's host cores ----
        from oracle import binding as ob
        o = ob.Oracle(base)
        nthreads = min(ob.max_threads(), args.frames)
        nc1 = 2
        t0 = time.perf_counter()
        o.eval_cost(params, xs[:nc1], nthreads=1)
        o.eval_normal(params, xs[:nc1], nthreads=1)
        t1 = (time.perf_counter() - t0) / nc1
        nco = 8 if nthreads >= 8 else 2
        t0 = time.perf_counter()
        o.eval_cost(params, xs[:nco], nthreads=nthreads)
        o.eval_normal(params, xs[:nco], nthreads=nthreads)
        to = (time.perf_counter() - t0) / nco
        res["cpu_baseline"] = {
            "value": 1.0 / to, "unit": "evals/s", "cores": nthreads, "kind": "port",
            "sample": "%d of the %d candidates of one step (cost tuple + normal equations each), OpenMP over keyframes as iba_func.cpp:203 / iba_local.cpp:162" % (nco, B),
            "single_thread_evals_per_s": 1.0 / t1,
            "single_thread_note": "%d candidates, 1 thread = what the reference's NOMAD loop runs (iba_global.cpp:385)" % nc1,
        }
    if rank == 0:
        print(json.dumps(res))
    h.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
