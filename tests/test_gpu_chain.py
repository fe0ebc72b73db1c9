"""The launch chain of round 5: up to 512 candidates per chain (iba_create_options.max_chain_batch), the candidate block carried to the
device by spare blocks of the chain's first kernel and the hand-eye terms evaluated beside the searches (chain_fold), against the
chain of rounds 1-4 (64 candidates, a staging launch at the head: max_chain_batch = 64, chain_fold = 0). A candidate's result may not
depend on the chain it rode in: the 64-double partial blocks are compared BIT FOR BIT — cost tuple, normal equations, every counter."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _blocks(h, xs, kind):
    import torch
    stride = 64
    d = torch.zeros(len(xs) * stride, dtype=torch.float64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    {"full": h.eval_full_partial, "cost": h.eval_cost_partial, "normal": h.eval_normal_partial}[kind](xs, d.data_ptr(), st)
    torch.cuda.synchronize()
    return d.cpu().numpy().reshape(len(xs), stride).copy()


def test_one_chain_of_many_equals_chains_of_64(pkg, synth, abi, ob, scene_small):
    import torch
    prob, meta = scene_small
    p = abi.reference_yaml_params()
    rng = np.random.default_rng(71)
    xs = np.vstack([synth.perturb(meta["x_gt"], rng, n=300), synth.perturb(meta["x_gt"], rng, rot=2e-3, trans=0.02, scale_rel=4e-3, n=37)])   # 337: one chain of 337
    with torch.cuda.stream(torch.cuda.Stream()):
        big = pkg.IbaHandle(prob, p)                                                             # defaults: chains of up to 512, folded head
        old = pkg.IbaHandle(prob, p, options={"max_chain_batch": 64, "chain_fold": 0})           # the chain of rounds 1-4
        mid = pkg.IbaHandle(prob, p, options={"max_chain_batch": 100, "chain_fold": 1})          # four chains, the last one short
        for kind in ("full", "cost", "normal"):
            a, b, c = _blocks(big, xs, kind), _blocks(old, xs, kind), _blocks(mid, xs, kind)
            assert np.array_equal(a, b, equal_nan=True), (kind, np.argwhere(a != b)[:5])
            assert np.array_equal(a, c, equal_nan=True), kind
        # ... and the blocking entry points, against the oracle on a few of them
        cost, nrm = big.eval_full(xs)
        o = ob.Oracle(prob)
        pick = [0, 63, 64, 299, 300, 336]
        for i, r in zip(pick, o.eval_cost(p, xs[pick], nthreads=8)):
            g = cost[i]
            assert (g.cnt_3d_2d, g.valid_cnt_3d_2d, g.cnt_3d_3d, g.valid_cnt_3d_3d, g.n_corr, g.frames_used) == (r.cnt_3d_2d, r.valid_cnt_3d_2d, r.cnt_3d_3d, r.valid_cnt_3d_3d, r.n_corr, r.frames_used)
            assert abs(g.f1 - r.f1) <= 1e-10 * r.f1 and abs(g.f2 - r.f2) <= 1e-10 * r.f2 and abs(g.C - r.C) <= 1e-10 * abs(r.C)
        for i, r in zip(pick, o.eval_normal(p, xs[pick], nthreads=8)):
            assert nrm[i].counts() == r.counts() and np.max(np.abs(nrm[i].H_np() - r.H_np())) <= 1e-8 * np.abs(r.H_np()).max()
        # the mixed batch above is too wide to share one pair search (the per-candidate kernel carried the head); a tight one shares it
        a, b = _blocks(big, xs[:300], "full"), _blocks(old, xs[:300], "full")
        assert big.last_path == 1 and np.array_equal(a, b, equal_nan=True)
        # the frozen problem's residual blocks at 337 other x in one chain (tools/sequence_fuzz.py caught the frozen counts reaching only the
        # first 64 candidates of a longer chain)
        for hh in (big, old):
            hh.build_problem(xs[3])
        fa, fb = big.eval_factors(xs), old.eval_factors(xs)
        assert all(a.counts() == b.counts() and np.array_equal(a.H_np(), b.H_np()) and np.array_equal(a.b_np(), b.b_np()) and a.cost == b.cost for a, b in zip(fa, fb))
        assert fa[336].frames_used == fa[0].frames_used > 0 and fa[336].n_corr == fa[0].n_corr > 0
        big.close(); old.close(); mid.close()
    with pytest.raises(pkg.IbaError):
        pkg.IbaHandle(prob, p, options={"max_chain_batch": 513})


def test_small_batches_and_the_hand_eye_terms_without_a_search(pkg, synth, abi, ob, scene_small):
    """The hand-eye terms ride in front of the search kernel's grid; a cost evaluation WITHOUT a search kernel (err_weight[1] = 0: no
    3d-3d term — BASELINE configs[0]) evaluates them inside the summing kernel instead. Both against the oracle, for 1, 2, 5 and 17
    candidates (17: the derivative half of the candidates is copied by a launch of its own; up to 16 it rides with the head)."""
    prob, meta = scene_small
    rng = np.random.default_rng(72)
    o = ob.Oracle(prob)
    for w1 in (1.0, 0.0):
        p = abi.reference_yaml_params()
        p.err_weight[1] = w1
        h = pkg.IbaHandle(prob, p)
        for n in (1, 2, 5, 17):
            xs = synth.perturb(meta["x_gt"], rng, n=n)
            cost, nrm = h.eval_full(xs)
            c2 = h.eval_cost(xs)
            for g, g2, r in zip(cost, c2, o.eval_cost(p, xs, nthreads=8)):
                assert (g.cnt_3d_2d, g.valid_cnt_3d_2d, g.cnt_3d_3d, g.n_corr, g.frames_used) == (r.cnt_3d_2d, r.valid_cnt_3d_2d, r.cnt_3d_3d, r.n_corr, r.frames_used)
                assert abs(g.C - r.C) <= 1e-10 * abs(r.C) and g.C == g2.C and abs(g.f1 - r.f1) <= 1e-10 * r.f1
            for g, r in zip(nrm, o.eval_normal(p, xs, nthreads=8)):
                assert g.counts() == r.counts() and np.max(np.abs(g.H_np() - r.H_np())) <= 1e-8 * np.abs(r.H_np()).max()
        h.close()


def test_group_of_shards_takes_long_chains(pkg, synth, abi, scene_small):
    """iba_group_* with batches beyond 64: one chain per device (the work buffers of every shard grow on its own thread before the
    chain is issued), host-reduced shards on this box; against the single handle, counters exact, sums to 1e-12."""
    prob, meta = scene_small
    p = abi.reference_yaml_params()
    xs = synth.perturb(meta["x_gt"], np.random.default_rng(73), n=200)
    h = pkg.IbaHandle(prob, p)
    c0, n0 = h.eval_full(xs)
    h.close()
    g = pkg.IbaGroup(prob, p, devices=(0, 0), host_reduce=True)
    c1, n1 = g.eval_full(xs)
    for a, b in zip(c0, c1):
        assert (a.cnt_3d_2d, a.valid_cnt_3d_2d, a.cnt_3d_3d, a.valid_cnt_3d_3d, a.n_corr, a.frames_used) == (b.cnt_3d_2d, b.valid_cnt_3d_2d, b.cnt_3d_3d, b.valid_cnt_3d_3d, b.n_corr, b.frames_used)
        assert abs(a.f1 - b.f1) <= 1e-12 * a.f1 and abs(a.f2 - b.f2) <= 1e-12 * a.f2 and abs(a.C - b.C) <= 1e-12 * abs(a.C)
    for a, b in zip(n0, n1):
        assert a.counts() == b.counts() and np.max(np.abs(a.H_np() - b.H_np())) <= 1e-12 * np.abs(a.H_np()).max()
    cc = g.eval_cost(xs)   # the cost-only chain of the group
    assert len(cc) == 200 and all(a.cnt_3d_2d == b.cnt_3d_2d and abs(a.f1 - b.f1) <= 1e-12 * a.f1 for a, b in zip(c0, cc))
    g.close()
