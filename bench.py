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
    ap.add_argument("--plane-cache", type=int, default=1, help="1 = memoise the x-independent local-plane fits (default), 0 = refit inside every evaluation")
    ap.add_argument("--no-extras", action="store_true", help="skip the plane-refit throughput and the LM calibration (profiling runs)")
    args = ap.parse_args()

    t_start = time.perf_counter()

    def stage(msg):
        if int(os.environ.get("RANK", "0")) == 0:
            print("[bench %7.1fs] %s" % (time.perf_counter() - t_start, msg), file=sys.stderr, flush=True)

    import torch
    import torch.distributed as dist
    stage("torch imported")

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or bool(os.environ.get("IBA_FORCE_DIST"))   # IBA_FORCE_DIST: exercise the RCCL path with one rank
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    stage("process group ready" if use_dist else "single process")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    pkg = importlib.import_module(PKG)
    synth = importlib.import_module(PKG + ".synth")
    abi = importlib.import_module(PKG + ".abi")

    # ---- workload: configs[1] shape per GPU (200 KF x 10k pts = 2M points), tiled once per rank ----
    base, meta = synth.make_scene(n_frames=args.frames, pts_per_frame=args.pts, n_keypoints=KEYPOINTS, seed=0)
    prob = base if world == 1 else synth.tile_scene(base, meta, world)[0]
    params = abi.reference_yaml_params(plane_cache=args.plane_cache)
    f0, f1 = rank * args.frames, (rank + 1) * args.frames
    h = pkg.IbaHandle(prob, params, device=local_rank, frame_begin=f0, frame_end=f1)
    stage("scene generated, handle created (static indices + plane memo)")
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
        if use_dist:   # frames shard across ranks: ONE sum all-reduce of the partial blocks (RCCL over xGMI)
            dist.all_reduce(d_cost)
        pc = d_cost.cpu().numpy()
        return pkg.finalize_cost(params, pc), pkg.finalize_normal(params, pc)

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        out = step(i)
    sync()
    stage("warmup done")
    t0 = time.perf_counter()
    kms = []
    for i in range(args.steps):
        out = step(i)
    sync()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    stage("timed steps done")
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
    # Unit of work = one candidate evaluated against ONE GPU's share: 200 keyframes / 2 M points (configs[1]). Weak scaling:
    # at N GPUs every candidate is evaluated against N x that many keyframes (configs[2], [3] grow the map with the GPU
    # count), i.e. N units per candidate; `value` is the aggregate over all ranks, the rate of the N-times larger sharded
    # problem itself is reported as config.sharded_problem_evals_per_s.
    value = world * evals / dt
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
            "unit_definition": "1 eval = one candidate x against %d keyframes / %.1fM points (+ normal equations); at N GPUs a candidate covers N x %d keyframes = N units"
                               % (args.frames, args.frames * args.pts / 1e6, args.frames),
            "sharded_problem_evals_per_s": evals / dt,
        },
        "roofline": {
            "bound": "hbm", "kernel": "iba_frame_kernel<MODE_BOTH>", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": None,
            "algorithmic_bytes_per_eval": per_eval, "evals_per_launch": B, "launch_ms": frame_ms,
        },
    }
    if not args.no_extras:
        # (1) same workload with the local planes refitted inside every evaluation, as the reference does
        other = abi.reference_yaml_params(plane_cache=1 - args.plane_cache)
        h.set_params(other)
        for i in range(2):
            h.eval_full_partial(xs_all[i], d_cost.data_ptr(), torch.cuda.current_stream().cuda_stream)
        sync()
        t0 = time.perf_counter()
        nrep = 5
        for i in range(nrep):
            h.eval_full_partial(xs_all[i % len(xs_all)], d_cost.data_ptr(), torch.cuda.current_stream().cuda_stream)
            if use_dist:
                dist.all_reduce(d_cost)
            d_cost.cpu()
        sync()
        t_other = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([t_other], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            t_other = float(t.item())
        res["value_plane_refit" if args.plane_cache else "value_plane_cache"] = world * B * nrep / t_other
        h.set_params(params)
        # (2) final SE(3): iba_local's outer loop + LM on the device path from a perturbed start (N=1 only)
        if world == 1:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import lm_ref
            x0 = synth.perturb(meta["x_gt"], np.random.default_rng(5), rot=1e-3, trans=0.01, scale_rel=3e-3, n=1)[0]
            t0 = time.perf_counter()
            xf, lr = h.calibrate_lm(x0, max_outer_iterations=10)
            e0 = lm_ref.se3_error(x0, meta["x_gt"], synth.sim3_exp)
            e1 = lm_ref.se3_error(xf, meta["x_gt"], synth.sim3_exp)
            res["final_se3"] = {"start_err_rad_m": [e0[0], e0[1]], "final_err_rad_m_vs_planted": [e1[0], e1[1]], "outer_iterations": lr.outer_iterations,
                                "evaluations": lr.evaluations, "seconds": time.perf_counter() - t0, "initial_cost": lr.initial_cost, "final_cost": lr.final_cost,
                                "note": "parity of the final SE(3) with the CPU path (1e-4 rad / 1e-3 m) is asserted in tests/test_gpu_calibrate.py"}
            # (3) time-to-calibration of the two-stage pipeline (README steps 3 + 4): batch-aware MADS on the cost path from a
            # start as far off as a hand-eye initialiser may be inside the reference's search box, then the LM polish
            xg0 = meta["x_gt"] + np.array([0.009, -0.006, 0.005, 0.06, -0.04, 0.05, 0.4])
            t0 = time.perf_counter()
            xg, mr = h.calibrate_mads(xg0, max_bb_eval=100000)
            t_mads = time.perf_counter() - t0
            xl, lr2 = h.calibrate_lm(xg, max_outer_iterations=10)
            t_all = time.perf_counter() - t0
            eg0 = lm_ref.se3_error(xg0, meta["x_gt"], synth.sim3_exp)
            eg1 = lm_ref.se3_error(xg, meta["x_gt"], synth.sim3_exp)
            eg2 = lm_ref.se3_error(xl, meta["x_gt"], synth.sim3_exp)
            res["global_then_local"] = {"start_err_rad_m": [eg0[0], eg0[1]], "after_mads_err_rad_m": [eg1[0], eg1[1]], "after_lm_err_rad_m": [eg2[0], eg2[1]],
                                        "scale_start_mads_lm_planted": [float(xg0[6]), float(xg[6]), float(xl[6]), float(meta["x_gt"][6])],
                                        "mads_evaluations": mr.evaluations, "mads_batches": mr.batches, "mads_restarts": mr.restarts, "mads_feasible": mr.feasible,
                                        "mads_f": mr.f, "mads_seconds": t_mads, "total_seconds": t_all,
                                        "note": "reference budget: 5000 NOMAD evaluations on one CPU thread (iba_calib_global.yml:42) at ~2 evals/s"}
            # (4) ORB-only extrinsic BA on a planted edge list of the C2 scale: 200 keyframes x 600 observations
            import ba_scene
            ba = importlib.import_module(PKG + ".ba")
            bprob, bx_gt = ba_scene.make(n_frames=200, pts_per_frame=600, seed=1, ba=ba)
            bh = ba.BaHandle(bprob)
            bx0 = bx_gt + np.array([0.008, -0.006, 0.005, 0.03, -0.02, 0.02, 0.3])
            bh.eval(bx0, want_chi2=False)
            t0 = time.perf_counter()
            for _ in range(20):
                bh.eval(bx0, want_chi2=False)
            t_lin = (time.perf_counter() - t0) / 20
            t0 = time.perf_counter()
            bx, br = bh.optimize(bx0)
            t_opt = time.perf_counter() - t0
            be = lm_ref.se3_error_rt(bx, bx_gt) if hasattr(lm_ref, "se3_error_rt") else None
            res["orb_only_ba"] = {"edges": int(br.n_edges), "inliers": int(br.n_inliers), "linearisation_ms": t_lin * 1e3,
                                  "edges_per_s": br.n_edges / t_lin, "optimize_seconds": t_opt, "device_evaluations": int(br.evaluations),
                                  "err_vs_planted": {"rot_rad": float(np.linalg.norm(bx[:3] - bx_gt[:3])), "trans_m": float(np.linalg.norm(bx[3:6] - bx_gt[3:6])), "scale": float(abs(bx[6] - bx_gt[6]))}}
            bh.close()
    tfile = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if os.path.exists(tfile):
        try:
            res["roofline"]["traffic"] = json.load(open(tfile)).get("hbm_bytes_per_launch")
        except Exception:
            pass

    stage("extras done")
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # ---- CPU baseline: the oracle (a port of the reference algorithm) on this box's host cores ----
        from oracle import binding as ob
        o = ob.Oracle(base)
        ncpu = ob.max_threads()
        scan = {}
        for nt in sorted(set([1, 8, 32, min(ncpu, args.frames)])):
            if nt > ncpu:
                continue
            nc = 2
            t0 = time.perf_counter()
            o.eval_cost(params, xs[:nc], nthreads=nt)
            o.eval_normal(params, xs[:nc], nthreads=nt)
            scan[nt] = nc / (time.perf_counter() - t0)
        best = max(scan, key=scan.get)
        res["cpu_baseline"] = {
            "value": scan[best], "unit": "evals/s", "cores": best, "kind": "port",
            "sample": "2 of the %d candidates of one step (cost tuple + normal equations each) per thread count; OpenMP over keyframes with the reference's "
                      "critical sections (iba_func.cpp:203, iba_global.cpp:239,318; iba_local.cpp:162); best thread count reported" % B,
            "evals_per_s_by_threads": {str(k): v for k, v in scan.items()},
            "single_thread_note": "1 thread = what the reference's NOMAD loop runs (iba_global.cpp:385)",
            "host_cores": ncpu,
        }
    if rank == 0:
        print(json.dumps(res))
    h.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
