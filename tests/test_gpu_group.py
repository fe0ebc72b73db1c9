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


def test_partial_block_summed_by_torch_rccl(pkg, synth, abi, scene_small):
    """One process per GPU, torchrun style (what bench.py does under the driver): the partial block of iba_eval_full_partial
    summed by torch.distributed's RCCL all-reduce (world of one on this box) and finalised on the host = iba_eval_full."""
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
        # an explicit stream: torch's default stream has the handle 0, which the C-ABI reads as "the handle's own stream".
        # (kept alive until the handle is closed: the handle's ring events have been recorded on it)
        ts = torch.cuda.Stream()
        with torch.cuda.stream(ts):
            d = torch.zeros(len(xs) * pkg.partial_stride(), dtype=torch.float64, device="cuda:0")
            st = torch.cuda.current_stream().cuda_stream
            assert st != 0
            h.eval_full_partial(xs, d.data_ptr(), st)
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


def test_comm_allreduce_entry_point(pkg, synth, abi, scene_small):
    """iba_comm_allreduce ITSELF — the C entry INTEGRATION.md gives C++ / MPI callers — on a one-rank ncclComm_t made by
    iba_comm_init_all: the block of iba_eval_full_partial, summed in place on the device, equals iba_eval_full."""
    import ctypes as C
    import torch
    prob, meta = scene_small
    p = abi.reference_yaml_params()
    xs = synth.perturb(meta["x_gt"], np.random.default_rng(4), n=5)
    L = pkg.load_library()
    L.iba_comm_init_all.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_int32), C.c_int32]
    L.iba_comm_count.argtypes = [C.c_void_p]
    L.iba_comm_destroy.argtypes = [C.c_void_p]
    comm = (C.c_void_p * 1)()
    dev = (C.c_int32 * 1)(0)
    assert L.iba_comm_init_all(comm, dev, 1) == 0
    assert L.iba_comm_count(comm[0]) == 1
    h = pkg.IbaHandle(prob, p)
    stride = pkg.partial_stride()
    ts = torch.cuda.Stream()   # an explicit stream (torch's default stream is the handle 0 = "the handle's own stream" to the C-ABI)
    with torch.cuda.stream(ts):
        d = torch.full((len(xs) * stride,), float("nan"), dtype=torch.float64, device="cuda:0")
        st = ts.cuda_stream
        assert st != 0
        h.eval_full_partial(xs, d.data_ptr(), st)
        assert L.iba_comm_allreduce(comm[0], C.c_void_p(d.data_ptr()), C.c_int32(len(xs)), C.c_void_p(st)) == 0
        ts.synchronize()
        part = d.cpu().numpy()
    cost, nrm = pkg.finalize_cost(p, part), pkg.finalize_normal(p, part)
    c1, n1 = h.eval_full(xs)
    for a, b in zip(c1, cost):
        assert a.as_dict() == b.as_dict()
    for a, b in zip(n1, nrm):
        assert np.array_equal(a.H_np(), b.H_np()) and np.array_equal(a.b_np(), b.b_np()) and a.counts() == b.counts()
    # bad arguments are refused, not dereferenced
    assert L.iba_comm_allreduce(None, C.c_void_p(d.data_ptr()), C.c_int32(1), C.c_void_p(st)) == 1
    assert L.iba_comm_allreduce(comm[0], None, C.c_int32(1), C.c_void_p(st)) == 1
    assert L.iba_comm_destroy(comm[0]) == 0
    h.close()


def test_which_librccl_runs(pkg):
    """The library loads librccl lazily; inside a torch process that is torch's own copy. Its version is compared with the
    headers the library was compiled against: the entry points used (ncclCommInitAll, ncclAllReduce, ncclCommCount,
    ncclCommDestroy) have kept their signatures through NCCL 2.x, so the major version must agree; a minor difference is
    reported, not hidden."""
    import torch  # noqa: F401  (maps torch/lib/librccl.so first, as in bench.py)
    text, rv, hv = pkg.rccl_info()
    print("librccl:", text)
    assert "path=" in text and rv > 20000 and hv > 20000
    assert rv // 10000 == hv // 10000, text
    g_ranks = None
    # the group's communicator agrees with RCCL's own count
    synth = __import__("importlib").import_module(pkg.__name__ + ".synth")
    prob, meta = synth.make_scene(n_frames=3, pts_per_frame=2000, n_keypoints=500, seed=2, new_mappoints=80, scan_kp=120)
    g = pkg.IbaGroup(prob, None, devices=(0,))
    g_ranks = g.comm_ranks
    g.close()
    assert g_ranks == 1


def _same_up_to_sum_order(c1, n1, c2, n2):
    for a, b in zip(c1, c2):
        da, db = a.as_dict(), b.as_dict()
        for k in da:
            if isinstance(da[k], int):
                assert da[k] == db[k], k
            else:
                assert (np.isnan(da[k]) and np.isnan(db[k])) or abs(da[k] - db[k]) <= 1e-12 * abs(db[k]), (k, da[k], db[k])
    for a, b in zip(n1, n2):
        assert a.counts() == b.counts()
        sc = np.max(np.abs(b.H_np()))
        assert np.max(np.abs(a.H_np() - b.H_np())) <= 1e-12 * sc and np.max(np.abs(a.b_np() - b.b_np())) <= 1e-12 * np.max(np.abs(b.b_np()))
        assert abs(a.cost - b.cost) <= 1e-12 * abs(b.cost)


def test_group_of_two_shards_on_this_box(pkg, synth, abi, scene_small):
    """The n > 1 logic on a one-GPU box: two frame shards, two handles, two issuing threads on device 0, the blocks summed on the
    host in rank order (IBA_GROUP_REDUCE_HOST) — everything of a two-GPU run but the collective. Counters equal the
    unsharded handle's exactly, sums to summation order; three shards likewise; the LM and MADS callers end where the
    single handle ends."""
    prob, meta = scene_small
    p = abi.reference_yaml_params()
    xs = np.vstack([meta["x_gt"][None], synth.perturb(meta["x_gt"], np.random.default_rng(3), n=70)])   # 71 candidates: two chunks
    h = pkg.IbaHandle(prob, p)
    c1, n1 = h.eval_full(xs)
    for devs in ((0, 0), (0, 0, 0)):
        g = pkg.IbaGroup(prob, p, devices=devs, host_reduce=True)
        assert g.comm_ranks == 0
        r = [g.frame_range(i) for i in range(len(devs))]
        assert r[0][0] == 0 and r[-1][1] == prob.n_frames and all(r[i][1] == r[i + 1][0] for i in range(len(devs) - 1)) and all(b > a for a, b in r)
        c2, n2 = g.eval_full(xs)
        _same_up_to_sum_order(c2, n2, c1, n1)
        cc = g.eval_cost(xs[:9])
        _same_up_to_sum_order(cc, [], c1[:9], [])
        # two calls give the same bits (rank-order host sum, fixed-order device sums)
        c3, n3 = g.eval_full(xs)
        for a, b in zip(n2, n3):
            assert np.array_equal(a.H_np(), b.H_np()) and a.cost == b.cost
        h.build_problem(xs[1])
        g.build_problem(xs[1])
        _same_up_to_sum_order([], g.eval_factors(xs[:4]), [], h.eval_factors(xs[:4]))
        if len(devs) == 2:
            x0 = synth.perturb(meta["x_gt"], np.random.default_rng(5), rot=1e-3, trans=0.01, scale_rel=3e-3, n=1)[0]
            xa, ra = h.calibrate_lm(x0, max_outer_iterations=4)
            xb, rb = g.calibrate_lm(x0, max_outer_iterations=4)
            assert np.max(np.abs(xa - xb)) <= 1e-9 and ra.outer_iterations == rb.outer_iterations
            assert g.last_issue_us > 0
        g.close()
    h.close()


def test_group_of_two_devices(pkg, synth, abi, scene_small):
    """Two real devices and the RCCL all-reduce between them (skipped on a one-GPU box; the driver's multi-GPU tier runs it)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    prob, meta = scene_small
    p = abi.reference_yaml_params()
    xs = np.vstack([meta["x_gt"][None], synth.perturb(meta["x_gt"], np.random.default_rng(3), n=20)])
    h = pkg.IbaHandle(prob, p)
    c1, n1 = h.eval_full(xs)
    cur = torch.cuda.current_device()
    g = pkg.IbaGroup(prob, p, devices=(0, 1))
    assert g.comm_ranks == 2
    c2, n2 = g.eval_full(xs)
    assert torch.cuda.current_device() == cur     # the caller's device is left alone
    _same_up_to_sum_order(c2, n2, c1, n1)
    h.build_problem(xs[1])
    g.build_problem(xs[1])
    _same_up_to_sum_order([], g.eval_factors(xs[:4]), [], h.eval_factors(xs[:4]))
    x0 = synth.perturb(meta["x_gt"], np.random.default_rng(5), rot=1e-3, trans=0.01, scale_rel=3e-3, n=1)[0]
    xa, ra = h.calibrate_lm(x0, max_outer_iterations=4)
    xb, rb = g.calibrate_lm(x0, max_outer_iterations=4)
    assert np.max(np.abs(xa - xb)) <= 1e-9
    g.close()
    h.close()


def _strip_keypoints(abi, prob, frames):
    """the problem with every keypoint of `frames` (and every covisibility slot that touches them) removed"""
    a = {k: v.copy() for k, v in prob.arrays.items()}
    F = prob.n_frames
    ko, co, mo = a["kp_offset"].astype(np.int64), a["covis_offset"].astype(np.int64), a["match_offset"].astype(np.int64)
    keepk = np.ones(int(ko[-1]), bool)
    for f in frames:
        keepk[ko[f]:ko[f + 1]] = False
    new_ko = np.concatenate([[0], np.cumsum([keepk[ko[f]:ko[f + 1]].sum() for f in range(F)])])
    keeps = np.ones(int(co[-1]), bool)
    for f in range(F):
        for gs in range(co[f], co[f + 1]):
            if f in frames or int(a["covis_frame"][gs]) in frames:
                keeps[gs] = False
    new_co = np.concatenate([[0], np.cumsum([keeps[co[f]:co[f + 1]].sum() for f in range(F)])])
    keepm = np.zeros(int(mo[-1]), bool)
    cnt = []
    for gs in range(int(co[-1])):
        if keeps[gs]:
            keepm[mo[gs]:mo[gs + 1]] = True
            cnt.append(mo[gs + 1] - mo[gs])
    a["kp_offset"] = new_ko.astype(np.uint64)
    a["kp_uv"] = a["kp_uv"].reshape(-1, 2)[keepk].reshape(-1)
    a["kp_has_mappoint"] = a["kp_has_mappoint"][keepk]
    a["kp_mappoint_w"] = a["kp_mappoint_w"].reshape(-1, 3)[keepk].reshape(-1)
    a["covis_offset"] = new_co.astype(np.uint64)
    a["covis_frame"] = a["covis_frame"][keeps]
    a["covis_relpose"] = a["covis_relpose"].reshape(-1, 12)[keeps].reshape(-1)
    a["match_offset"] = np.concatenate([[0], np.cumsum(cnt)]).astype(np.uint64)
    a["match_kp_ref"] = a["match_kp_ref"][keepm]
    a["match_kp_covis"] = a["match_kp_covis"][keepm]
    return abi.Problem(**a)


def test_shard_without_any_keypoint(pkg, synth, abi, ob):
    """A shard whose frames hold no keypoint at all contributes ZERO records (round 2 read uninitialised factor records
    there). Frames 3..5 of six lose their keypoints: the handle over them alone returns zeros, and the two-shard sum equals
    the oracle on the whole problem."""
    prob0, meta = synth.make_scene(n_frames=6, pts_per_frame=2500, n_keypoints=800, seed=13, new_mappoints=100, scan_kp=150)
    prob = _strip_keypoints(abi, prob0, (3, 4, 5))
    p = abi.reference_yaml_params()
    p.num_min_corr = 10
    p.num_min_corr_cost = 10
    xs = synth.perturb(meta["x_gt"], np.random.default_rng(2), n=3)
    o = ob.Oracle(prob)
    hz = pkg.IbaHandle(prob, p, frame_begin=3, frame_end=6)
    for rep in range(2):
        cz, nz = hz.eval_full(xs)
        part = hz.debug_last_partials(len(xs))
        assert np.all(part[:, 12:53] == 0.0), "factor slots of a keypoint-less shard must be exact zeros"
        for n in nz:
            assert all(v == 0 for v in n.counts().values()) and np.all(n.H_np() == 0) and n.cost == 0
    hz.build_problem(xs[0])
    for n in hz.eval_factors(xs):
        assert all(v == 0 for v in n.counts().values()) and np.all(n.H_np() == 0) and n.cost == 0
    hz.close()
    g = pkg.IbaGroup(prob, p, devices=(0, 0), host_reduce=True)
    assert g.frame_range(1) == (3, 6)
    c2, n2 = g.eval_full(xs)
    co, no = o.eval_cost(p, xs), o.eval_normal(p, xs)
    for a, b in zip(c2, co):
        assert (a.n_corr, a.cnt_3d_2d, a.valid_cnt_3d_2d, a.cnt_3d_3d, a.valid_cnt_3d_3d, a.frames_used) == (b.n_corr, b.cnt_3d_2d, b.valid_cnt_3d_2d, b.cnt_3d_3d, b.valid_cnt_3d_3d, b.frames_used)
        assert abs(a.f1 - b.f1) <= 1e-10 * abs(b.f1) and abs(a.f2 - b.f2) <= 1e-10 * abs(b.f2)
    for a, b in zip(n2, no):
        assert a.counts() == b.counts()
        assert np.max(np.abs(a.H_np() - b.H_np())) <= 1e-9 * np.max(np.abs(b.H_np()))
    g.close()


def test_back_to_back_partial_calls_on_a_caller_stream(pkg, synth, abi, scene_small):
    """Several iba_eval_full_partial / iba_eval_cost_partial calls enqueued on the caller's stream without a host wait between
    them (what a pipelined caller does): the staging launch of call n + 1 runs on the handle's side stream and must stay behind
    everything call n reads (candidate ring slot, hand-eye terms); every block equals the synchronous result bit for bit."""
    import torch
    prob, meta = scene_small
    p = abi.reference_yaml_params()
    rng = np.random.default_rng(11)
    sets = [synth.perturb(meta["x_gt"], rng, n=n) for n in (7, 1, 16, 3, 16, 7, 2, 9)]
    h = pkg.IbaHandle(prob, p)
    ts = torch.cuda.Stream()
    stride = pkg.partial_stride()
    with torch.cuda.stream(ts):
        st = torch.cuda.current_stream().cuda_stream
        outs = [torch.zeros(len(x) * stride, dtype=torch.float64, device="cuda:0") for x in sets]
        for rep in range(3):
            for i, (x, d) in enumerate(zip(sets, outs)):
                if (i + rep) % 3 == 2:
                    h.eval_cost_partial(x, d.data_ptr(), st)
                else:
                    h.eval_full_partial(x, d.data_ptr(), st)
        ts.synchronize()
        parts = [d.cpu().numpy() for d in outs]
    for i, (x, part) in enumerate(zip(sets, parts)):
        cost = pkg.finalize_cost(p, part)
        c1, n1 = h.eval_full(x)
        for a, b in zip(c1, cost):
            assert a.as_dict() == b.as_dict(), i
        if (i + 2) % 3 != 2:   # the last repetition of this set was a full evaluation: the normal equations are in the block too
            for a, b in zip(n1, pkg.finalize_normal(p, part)):
                assert np.array_equal(a.H_np(), b.H_np()) and a.counts() == b.counts(), i
    h.close()


def _group_with_env(pkg, prob, p, env, **kw):
    import os
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return pkg.IbaGroup(prob, p, **kw)
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


def test_a_failing_device_is_an_error_not_a_hang(pkg, synth, abi, scene_small):
    """Round 3: a worker whose launch chain failed returned before its all-reduce while its peers sat in theirs forever
    (ADVICE r3). Now every worker meets its peers at a barrier between launch chain and collective: a failure
    before it (injected with IBA_DEBUG_FAIL_RANK on one of two shards) makes the call return an error within a second, nobody
    enters the collective, and the NEXT call on the same group works and gives the single-device bits. A failure at the
    collective itself (IBA_DEBUG_FAIL_PHASE=2: the rank never enqueues its all-reduce; one-rank RCCL group on this box) aborts
    the communicator: the call returns an error, the group is broken (every later call fails at once), destroy does not hang."""
    import time
    prob, meta = scene_small
    p = abi.reference_yaml_params()
    xs = synth.perturb(meta["x_gt"], np.random.default_rng(8), n=6)
    h = pkg.IbaHandle(prob, p)
    want = h.eval_cost(xs)
    # (1) two host-reduced shards on this device, shard 1 fails before the collective phase
    g = _group_with_env(pkg, prob, p, {"IBA_DEBUG_FAIL_RANK": "1", "IBA_DEBUG_FAIL_PHASE": "1"}, devices=(0, 0), host_reduce=True)
    t0 = time.perf_counter()
    with pytest.raises(pkg.IbaError) as ei:
        g.eval_cost(xs)
    assert time.perf_counter() - t0 < 1.0
    assert "injected failure" in str(ei.value)
    got = g.eval_cost(xs)                                              # the injection was one-shot: the group is intact
    for a, b in zip(want, got):
        assert (a.n_corr, a.cnt_3d_2d, a.valid_cnt_3d_2d, a.cnt_3d_3d, a.valid_cnt_3d_3d) == (b.n_corr, b.cnt_3d_2d, b.valid_cnt_3d_2d, b.cnt_3d_3d, b.valid_cnt_3d_3d)
        assert abs(a.f1 - b.f1) <= 1e-12 * abs(a.f1)
    g.close()
    # (2) the same with RCCL (one rank): fails before the collective, the next call works
    g = _group_with_env(pkg, prob, p, {"IBA_DEBUG_FAIL_RANK": "0", "IBA_DEBUG_FAIL_PHASE": "1"}, devices=(0,))
    t0 = time.perf_counter()
    with pytest.raises(pkg.IbaError):
        g.eval_full(xs)
    assert time.perf_counter() - t0 < 1.0
    for a, b in zip(want, g.eval_cost(xs)):
        assert a.as_dict() == b.as_dict()
    g.close()
    # (3) the collective itself fails on a rank: communicator aborted, group broken, no hang anywhere
    g = _group_with_env(pkg, prob, p, {"IBA_DEBUG_FAIL_RANK": "0", "IBA_DEBUG_FAIL_PHASE": "2"}, devices=(0,))
    t0 = time.perf_counter()
    with pytest.raises(pkg.IbaError) as ei:
        g.eval_cost(xs)
    assert time.perf_counter() - t0 < 1.0 and "ncclAllReduce" in str(ei.value)
    with pytest.raises(pkg.IbaError) as ei:
        g.eval_cost(xs)
    assert "broken" in str(ei.value)
    t0 = time.perf_counter()
    g.close()
    assert time.perf_counter() - t0 < 5.0
    for a, b in zip(want, h.eval_cost(xs)):                           # the device and the single handle are unharmed
        assert a.as_dict() == b.as_dict()
    h.close()
