"""CPU-side checks of the product library: it loads, exports every symbol include/*.h declares (the product surface iba_mi355x.h
and the diagnostics iba_mi355x_debug.h),
fails loudly without a GPU (no CPU fallback), and its host-only logic (finalisation, sharding) is right."""
import ctypes as C
import re

import numpy as np
import pytest


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def test_library_exports_every_declared_symbol(pkg):
    pkg.build_extension()
    lib = pkg.load_library()
    import glob, os
    headers = sorted(glob.glob(os.path.join(os.path.dirname(pkg.HEADER_PATH), "*.h")))
    assert [os.path.basename(h) for h in headers] == ["iba_mi355x.h", "iba_mi355x_debug.h"] and sorted(pkg.HEADER_PATHS) == headers
    names = sorted(set(re.findall(r"\b(iba_[a-z_0-9]+)\s*\(", "".join(open(h).read() for h in headers))))
    assert len(names) >= 90
    # header hygiene: nothing diagnostic is declared in the product header
    product = set(re.findall(r"\b(iba_[a-z_0-9]+)\s*\(", open(pkg.HEADER_PATH).read()))
    assert not [n for n in product if n.startswith("iba_debug_") or "selftest" in n or n in ("iba_last_phase_ms", "iba_last_kernel_ms", "iba_set_timing")]
    for n in names:
        assert getattr(lib, n) is not None, n
    assert lib.iba_partial_stride() == 64


def test_struct_layouts_match_header(pkg, abi):
    """iba_default_params fills the ctypes mirror correctly -> field offsets agree with the C struct."""
    p = pkg.default_params()
    assert (p.max_pixel_dist, p.num_min_corr_cost, p.corr_3d_2d_threshold, p.corr_3d_3d_threshold) == (1.5, 30, 40.0, 5.0)
    assert (p.norm_max_pts, p.norm_min_pts, p.norm_radius, p.norm_reg_threshold, p.min_diff_dist) == (30, 5, 0.6, 0.04, 0.01)
    assert (p.err_weight[0], p.err_weight[1], p.use_plane, p.num_min_corr, p.max_3d_dist) == (1.0, 1.0, 1, 30, 1.0)
    assert (p.neigh_radius, p.neigh_max_pts, p.neigh_min_pts, p.local_min_diff_dist, p.local_norm_reg_threshold) == (0.6, 30, 5, 0.2, 0.001)
    assert (p.robust_kernel_delta, p.robust_kernel_3ddelta, p.plane_cache) == (2.98, 1.0, 1)


def test_no_cpu_fallback(pkg, abi, synth):
    if _has_gpu():
        pytest.skip("GPU present")
    prob, _ = synth.make_scene(n_frames=2, pts_per_frame=300, n_keypoints=100, seed=0, new_mappoints=10, scan_kp=10)
    with pytest.raises(pkg.IbaError) as e:
        pkg.IbaHandle(prob, abi.reference_yaml_params())
    assert e.value.status == 2 and "no CPU fallback" in str(e.value)


def test_finalize_cost_host_logic(pkg, abi):
    S = pkg.partial_stride()
    p = abi.reference_yaml_params()
    part = np.zeros((3, S))
    part[0, :12] = [50.0, 6.0, 0.3, 3, 110, 100, 40, 30, 20, 10, 4, 700]
    part[1, :12] = [0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0]             # nothing valid: sentinels + NaN C
    part[2, :12] = [10.0, 123.0, 0.1, 1, 10, 5, 2, 2, 0, 0, 2, 70]
    out = pkg.finalize_cost(p, part)
    assert (out[0].f1, out[0].f2, out[0].valid_cnt_3d_2d, out[0].cnt_3d_2d, out[0].n_corr) == (0.5, 0.2, 100, 110, 700) and np.isclose(out[0].C, 0.1)
    mx = np.finfo(np.float64).max
    assert out[1].f1 == mx and out[1].f2 == mx and np.isnan(out[1].C)
    p0 = abi.reference_yaml_params()
    p0.err_weight[1] = 0.0
    assert pkg.finalize_cost(p0, part)[2].f2 == 0.0                   # corr_3d_3d_err reset (iba_global.cpp:216)


def test_finalize_normal_host_logic(pkg, abi):
    S = pkg.partial_stride()
    part = np.zeros((1, S))
    part[0, 12:40] = np.arange(1, 29)
    part[0, 40:47] = np.arange(7) + 0.5
    part[0, 47:55] = [9.0, 4.0, 3, 2, 1, 10, 5, 600]
    o = pkg.finalize_normal(abi.reference_yaml_params(), part)[0]
    H = o.H_np()
    assert np.array_equal(H, H.T) and H[0, 0] == 1 and H[0, 6] == 7 and H[1, 1] == 8 and H[6, 6] == 28
    assert np.array_equal(o.b_np(), np.arange(7) + 0.5) and (o.chi2, o.cost) == (9.0, 4.0)
    assert o.counts() == dict(n_factor_3d2d=3, n_factor_p2pl=2, n_factor_p2pt=1, n_residuals=10, frames_used=5, n_corr=600)


def test_shard_frames(pkg):
    for F, W in ((200, 8), (7, 3), (5, 8), (1, 1)):
        cuts = [pkg.shard_frames(F, W, r) for r in range(W)]
        assert cuts[0][0] == 0 and cuts[-1][1] == F and all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
    w = np.array([1, 1, 1, 1, 100, 1, 1, 1.0])
    a, b = pkg.shard_frames(8, 2, 0, w), pkg.shard_frames(8, 2, 1, w)
    assert a[1] == b[0] and 4 <= a[1] <= 5


def test_synth_scene_is_deterministic_and_consistent(synth):
    a, ma = synth.make_scene(n_frames=3, pts_per_frame=500, n_keypoints=200, seed=5, new_mappoints=30, scan_kp=30)
    b, _ = synth.make_scene(n_frames=3, pts_per_frame=500, n_keypoints=200, seed=5, new_mappoints=30, scan_kp=30)
    for k in a.arrays:
        assert np.array_equal(a.arrays[k], b.arrays[k]), k
    R, t, s = synth.sim3_exp(ma["x_gt"])
    Rg, tg = synth.gt_extrinsic()
    assert np.allclose(R, Rg, atol=1e-12) and np.allclose(t, tg, atol=1e-12) and s == ma["s_star"]
    t2, _ = synth.tile_scene(a, ma, 2)
    assert t2.n_frames == 6 and t2.n_points == 2 * a.n_points and t2.arrays["covis_frame"][-1] >= 3


def test_descriptor_validation_precedes_the_device_probe(pkg, abi, synth):
    """Non-monotonic CSR offsets are refused with IBA_ERR_INVALID_ARG before anything is dereferenced — also for covisible
    keyframes outside the owned frame range."""
    import copy
    prob, _ = synth.make_scene(n_frames=3, pts_per_frame=300, n_keypoints=100, seed=0, new_mappoints=10, scan_kp=10)
    for name in ("kp_offset", "pt_offset", "covis_offset", "match_offset"):
        bad = copy.copy(prob)
        bad.arrays = {k: v.copy() for k, v in prob.arrays.items()}
        o = bad.arrays[name]
        o[2] = o[1] - 1 if o[1] > 0 else o[-1] + 7   # decreases at index 2 (or at the end)
        if not np.any(np.diff(o.astype(np.int64)) < 0):
            o[1] = o[-1] + 7
        with pytest.raises(pkg.IbaError) as e:
            pkg.IbaHandle(bad, abi.reference_yaml_params(), frame_begin=0, frame_end=1)
        assert e.value.status == 1, name


def test_whiten_normal_host_math(pkg, abi):
    """iba_whiten_normal: [J | r] = upper Cholesky factor of [[H, b], [b^T, 2 cost]] -> J^T J = H, J^T r = b, |r|^2 = 2 cost,
    on a random robustified least-squares problem (Huber weights, cost = 1/2 sum rho) and on a rank-deficient one."""
    import ctypes as C
    lib = pkg.load_library()
    rng = np.random.default_rng(0)
    for rank_deficient in (False, True):
        m = 400
        Jb = rng.normal(size=(m, 7)) * np.array([1e3, 2e3, 5e2, 30, 40, 20, 3])
        if rank_deficient:
            Jb[:, 6] = 0.0                       # nothing constrains the scale
        rb = rng.normal(size=m) * 3
        a = 2.98
        s = rb ** 2
        w = np.where(s > a * a, a / np.sqrt(s), 1.0)
        rho = np.where(s > a * a, 2 * a * np.sqrt(s) - a * a, s)
        n = abi.IbaNormalOut()
        H = (Jb * w[:, None]).T @ Jb
        b = (Jb * w[:, None]).T @ rb
        for i in range(49):
            n.H[i] = H.flat[i]
        for i in range(7):
            n.b[i] = b[i]
        n.cost = 0.5 * rho.sum()
        r = np.zeros(8)
        J = np.zeros((8, 7))
        assert lib.iba_whiten_normal(C.byref(n), r.ctypes.data_as(C.c_void_p), J.ctypes.data_as(C.c_void_p)) == 0
        assert np.allclose(J.T @ J, H, rtol=1e-10, atol=1e-10 * np.abs(H).max())
        assert np.allclose(J.T @ r, b, rtol=1e-10, atol=1e-10 * np.abs(b).max())
        assert abs(r @ r - 2 * n.cost) <= 1e-10 * 2 * n.cost
        assert np.allclose(J, np.triu(J[:, :7].reshape(8, 7)) if False else J) and np.all(np.abs(np.tril(np.hstack([J, r[:, None]]), -1)) == 0)   # upper triangular
