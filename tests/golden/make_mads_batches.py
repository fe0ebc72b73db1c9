"""Fixture for the CPU test of the pair-search planner: a sample of the black-box batches of a recorded MADS run
(gpurun_out/mads_trace.npz, written on the GPU box by tools/mads_trace_record.py: bench scene, 200 keyframes) with, per batch,
the number of groups and the group of every candidate as an INDEPENDENT numpy restatement of the planner's rule finds them
(nominal spread = fx (12 rho_row + tau_max) 1.8 / 10; farthest-point seeds from the candidate nearest the batch mean; nearest seed
wins; accept when every group is within 20 px; give up beyond 4 groups) -> tests/golden/mads_batches_sample.npz."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
synth = importlib.import_module("spatial-temporal-lidar-camera-calibration_amd.synth")
z = np.load(os.path.join(ROOT, "gpurun_out", "mads_trace.npz"))
X, bs, fx = z["x"][:, :7], z["batch_sizes"], float(z["fx"])
MAX_PX, MAX_GROUPS = 20.0, 4


def rt(x):
    R, t, _ = synth.sim3_exp(x)
    return np.asarray(R), np.asarray(t)


def px_of(Rs, ts, R0, t0):
    A = np.einsum("bij,kj->bik", Rs, R0)
    a = ts - np.einsum("bij,j->bi", A, t0)
    rho = np.abs(A - np.eye(3)).max(0) * (1 + 1e-9) + 1e-15
    tau = np.abs(a).max(0) * (1 + 1e-9) + 1e-15
    return fx * (rho.sum(1).max() * 12.0 + tau.max()) * 1.8 / 10.0


def ref_of(Rs, ts):
    m = np.concatenate([Rs.reshape(len(Rs), 9), ts], 1)
    d = np.abs(m - m.mean(0)) * np.array([12.0] * 9 + [1.0] * 3)
    return int(np.argmin(d.max(1)))


def plan(xb):
    RT = [rt(x) for x in xb]
    Rs = np.array([r for r, _ in RT]); ts = np.array([t for _, t in RT])
    n = len(xb)
    r0 = ref_of(Rs, ts)
    if px_of(Rs, ts, Rs[r0], ts[r0]) <= MAX_PX:
        return 1, np.zeros(n, np.int32)
    seeds = [r0]
    dist = [np.array([px_of(Rs[b:b + 1], ts[b:b + 1], Rs[r0], ts[r0]) for b in range(n)])]
    while True:
        d = np.min(np.array(dist), 0)
        far = int(np.argmax(d))
        if len(seeds) >= MAX_GROUPS:
            return 0, np.zeros(n, np.int32)
        seeds.append(far)
        dist.append(np.array([px_of(Rs[b:b + 1], ts[b:b + 1], Rs[far], ts[far]) for b in range(n)]))
        lab = np.argmin(np.array(dist), 0)
        ok = True
        for g in range(len(seeds)):
            idx = np.where(lab == g)[0]
            if len(idx) == 0:
                ok = False; break
            r = idx[ref_of(Rs[idx], ts[idx])]
            if px_of(Rs[idx], ts[idx], Rs[r], ts[r]) > MAX_PX:
                ok = False; break
        if ok:
            return len(seeds), lab.astype(np.int32)


starts = np.concatenate([[0], np.cumsum(bs)])
pick = sorted(set(np.linspace(0, len(bs) - 1, 90).astype(int)))
xs, sizes, ngroups, labels = [], [], [], []
for i in pick:
    xb = X[starts[i]:starts[i + 1]]
    ng, lab = plan(xb)
    xs.append(xb); sizes.append(len(xb)); ngroups.append(ng); labels.append(lab)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "mads_batches_sample.npz"), x=np.vstack(xs), batch_sizes=np.array(sizes, np.int32), n_groups=np.array(ngroups, np.int32),
                    group_of=np.concatenate(labels), fx=fx, max_px=MAX_PX, max_groups=MAX_GROUPS, batch_index=np.array(pick, np.int32))
print("batches", len(pick), "groups histogram", np.bincount(ngroups))
