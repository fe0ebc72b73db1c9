"""Histogram of a recorded MADS run's black-box batches (gpurun_out/mads_trace.npz from tools/mads_trace_record.py) by the
nominal projection spread the library's planner computes (plan_pairs: fx (12 rho_row + tau_max) 1.8 / 10) — for the WHOLE batch
and for the groups a greedy clustering finds — no GPU needed."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
synth = importlib.import_module("spatial-temporal-lidar-camera-calibration_amd.synth")
z = np.load(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "mads_trace.npz"))
X, bs, path, wall, fx = z["x"][:, :7], z["batch_sizes"], z["path"], z["wall"], float(z["fx"])


def rt(x):
    R, t, s = synth.sim3_exp(x)
    return np.asarray(R), np.asarray(t)


def spread_px(Rs, ts, ref):
    R0, t0 = Rs[ref], ts[ref]
    A = np.einsum("bij,kj->bik", Rs, R0)            # R_b R_0^T
    a = ts - np.einsum("bij,j->bi", A, t0)
    rho = np.abs(A - np.eye(3)).max(0)
    tau = np.abs(a).max(0)
    return fx * (rho.sum(1).max() * 12.0 + tau.max()) * 1.8 / 10.0


def pair_px(Ri, ti, Rj, tj):
    A = Ri @ Rj.T
    a = ti - A @ tj
    return fx * (np.abs(A - np.eye(3)).sum(1).max() * 12.0 + np.abs(a).max()) * 1.8 / 10.0


def ref_of(Rs, ts):
    m = np.concatenate([Rs.reshape(len(Rs), 9), ts], 1)
    d = np.abs(m - m.mean(0)) * np.array([12.0] * 9 + [1.0] * 3)
    return int(np.argmin(d.max(1)))


def cluster(Rs, ts, max_px, max_groups):
    """greedy: seeds = farthest-point; every candidate joins the nearest seed; a group is fine when its spread around its own reference <= max_px"""
    n = len(Rs)
    seeds = [ref_of(Rs, ts)]
    while True:
        d = np.array([[pair_px(Rs[b], ts[b], Rs[s], ts[s]) for s in seeds] for b in range(n)])
        lab = d.argmin(1)
        groups = [np.where(lab == g)[0] for g in range(len(seeds))]
        sp = [spread_px(Rs[g], ts[g], ref_of(Rs[g], ts[g])) if len(g) else 0.0 for g in groups]
        if max(sp) <= max_px or len(seeds) >= max_groups:
            return groups, sp
        worst = int(np.argmax(sp))
        far = groups[worst][int(np.argmax(d[groups[worst], worst]))]
        seeds.append(int(far))


at = 0
rows = []
for i, b in enumerate(bs):
    xb = X[at:at + b]; at += b
    RT = [rt(x) for x in xb]
    Rs = np.array([r for r, _ in RT]); ts = np.array([t for _, t in RT])
    whole = spread_px(Rs, ts, ref_of(Rs, ts))
    groups, sp = cluster(Rs, ts, 20.0, 4)
    rows.append((b, whole, len(groups), max(sp), min(len(g) for g in groups), path[i], wall[i]))
rows = np.array(rows)
print("batches %d, evaluations %d, wall %.3f s" % (len(rows), bs.sum(), wall.sum()))
edges = [0, 2, 5, 10, 20, 40, 80, 160, 1e9]
print("whole-batch nominal spread (px): count, time share")
for lo, hi in zip(edges[:-1], edges[1:]):
    m = (rows[:, 1] >= lo) & (rows[:, 1] < hi)
    print("  [%5g, %5g): %5d batches, %.3f of the time, mean B %.1f" % (lo, hi, m.sum(), wall[m].sum() / wall.sum(), rows[m, 0].mean() if m.any() else 0))
wide = rows[:, 1] > 20.0
print("wide batches (> 20 px): %d; of them clustered into <= 4 groups all <= 20 px: %d" % (wide.sum(), (wide & (rows[:, 3] <= 20.0)).sum()))
for ng in (1, 2, 3, 4):
    m = wide & (rows[:, 2] == ng)
    print("  %d groups: %d batches (ok %d), time share %.3f, smallest group mean %.1f" % (ng, m.sum(), (m & (rows[:, 3] <= 20)).sum(), wall[m].sum() / wall.sum(), rows[m, 4].mean() if m.any() else 0))
m = wide & (rows[:, 3] > 20.0)
print("still wide after clustering: %d batches, time share %.3f; their max group spread: median %.1f px" % (m.sum(), wall[m].sum() / wall.sum(), np.median(rows[m, 3]) if m.any() else 0))
np.save(os.path.join(ROOT, "gpurun_out", "mads_trace_rows.npy"), rows)
