"""iba_group: the frames of a problem sharded over the GPUs of a node inside one process, one RCCL all-reduce of the partial
blocks per evaluation (csrc/iba_group.hip). The GPU box of the test tier has ONE device: the group then has one member and the
all-reduce is RCCL's one-rank collective, in place on the device block — everything but the number of ranks is the code path
of an 8-GPU run. Results must equal the single-device entry points bit for bit (a sum over one rank is the identity)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_group_of_one_equals_the_single_device_path(pkg, synth, abi, scene_small):
    prob, meta = scene_small
    p = abi.reference_yaml_params()
    xs = np.vstack([meta["x_gt"][None], synth.perturb(meta["x_gt"], np.random.default_rng(3), n=7)])
    h = pkg.IbaHandle(prob, p)
    g = pkg.IbaGroup(prob, p, devices=(0,))
    assert g.frame_range(0) == (0, prob.n_frames)
    c1, n1 = h.eval_full(xs)
    c2, n2 = g.eval_full(xs)
    for a, b in zip(c1, c2):
        assert a.as_dict() == b.as_dict()
    for a, b in zip(n1, n2):
        assert a.counts() == b.counts() and np.array_equal(a.H_np(), b.H_np()) and np.array_equal(a.b_np(), b.b_np()) and a.cost == b.cost and a.chi2 == b.chi2
    for a, b in zip(h.eval_cost(xs), g.eval_cost(xs)):
        assert a.as_dict() == b.as_dict()
    for a, b in zip(h.eval_normal(xs[:3]), g.eval_normal(xs[:3])):
        assert np.array_equal(a.H_np(), b.H_np())
    # frozen problem + the callers
    h.build_problem(xs[1])
    g.build_problem(xs[1])
    for a, b in zip(h.eval_factors(xs[:4]), g.eval_factors(xs[:4])):
        assert a.counts() == b.counts() and np.array_equal(a.H_np(), b.H_np()) and np.array_equal(a.b_np(), b.b_np())
    x0 = synth.perturb(meta["x_gt"], np.random.default_rng(5), rot=1e-3, trans=0.01, scale_rel=3e-3, n=1)[0]
    xa, ra = h.calibrate_lm(x0, max_outer_iterations=4)
    xb, rb = g.calibrate_lm(x0, max_outer_iterations=4)
    assert np.array_equal(xa, xb) and ra.evaluations == rb.evaluations and ra.final_cost == rb.final_cost
    xa, ra = h.calibrate_mads(xs[2], max_bb_eval=600)
    xb, rb = g.calibrate_mads(xs[2], max_bb_eval=600)
    assert np.array_equal(xa, xb) and ra.evaluations == rb.evaluations and ra.f == rb.f
    g.close()
    h.close()


def test_comm_allreduce_entry_point_with_torch_rccl(pkg, synth, abi, scene_small):
    """One process per GPU with the caller's own communicator: the partial block of iba_eval_full_partial summed by
    torch.distributed's RCCL all-reduce (world of one on this box) and finalised on the host = iba_eval_full."""
    import os
    import torch
    import torch.distributed as dist
    prob, meta = scene_small
    p = abi.reference_yaml_params()
    xs = synth.perturb(meta["x_gt"], np.random.default_rng(4), n=5)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29600 + os.getpid() % 300))
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        h = pkg.IbaHandle(prob, p)
        d = torch.zeros(len(xs) * pkg.partial_stride(), dtype=torch.float64, device="cuda:0")
        h.eval_full_partial(xs, d.data_ptr(), torch.cuda.current_stream().cuda_stream)
        dist.all_reduce(d)
        part = d.cpu().numpy()
        cost, nrm = pkg.finalize_cost(p, part), pkg.finalize_normal(p, part)
        c1, n1 = h.eval_full(xs)
        for a, b in zip(c1, cost):
            assert a.as_dict() == b.as_dict()
        for a, b in zip(n1, nrm):
            assert np.array_equal(a.H_np(), b.H_np()) and a.counts() == b.counts()
        h.close()
    finally:
        dist.destroy_process_group()
