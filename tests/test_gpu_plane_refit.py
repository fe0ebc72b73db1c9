"""plane_cache = 0: every local plane (kNN(30) + covariance + closed-form eigen, iba_global.cpp:125-147,
pointcloud.h:733-760 / 699-717) is refitted inside each evaluation, exactly the work the reference does.
Every term must equal the memoised mode's (all counters exact, every distance the same: the two modes differ only in the
order in which they sum, so the means agree to a few ulp), and match the oracle to the usual bars."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_refit_equals_memoised_and_oracle(pkg, synth, abi, ob, scene_small):
    prob, meta = scene_small
    pc = abi.reference_yaml_params(plane_cache=1)
    pr = abi.reference_yaml_params(plane_cache=0)
    hc = pkg.IbaHandle(prob, pc)
    hr = pkg.IbaHandle(prob, pr)
    o = ob.Oracle(prob)
    rng = np.random.default_rng(41)
    xs = np.vstack([meta["x_gt"][None], synth.perturb(meta["x_gt"], rng, n=3)])
    cc, nc = hc.eval_full(xs)
    cr, nr = hr.eval_full(xs)
    for a, b in zip(cc, cr):
        da, db = a.as_dict(), b.as_dict()
        for key in da:
            if key in ("f1", "f2", "C"):
                assert abs(da[key] - db[key]) <= 1e-13 * abs(db[key]), key   # same terms, different summation order
            else:
                assert da[key] == db[key], key
    for a, b in zip(nc, nr):
        assert a.counts() == b.counts() and np.array_equal(a.H_np(), b.H_np()) and np.array_equal(a.b_np(), b.b_np()) and a.cost == b.cost
    oc = o.eval_cost(pr, xs)
    for a, b in zip(cr, oc):
        assert (a.cnt_3d_3d, a.valid_pl_3d_3d, a.valid_pt_3d_3d, a.n_corr) == (b.cnt_3d_3d, b.valid_pl_3d_3d, b.valid_pt_3d_3d, b.n_corr)
        assert abs(a.f2 - b.f2) <= 1e-10 * abs(b.f2)
    # separate entry points and the frozen problem in refit mode
    assert hr.eval_cost(xs)[1].as_dict()["cnt_3d_3d"] == cr[1].cnt_3d_3d
    n1 = hr.eval_normal(xs)[2]
    assert n1.counts() == nr[2].counts() and np.max(np.abs(n1.H_np() - nr[2].H_np())) <= 1e-12 * np.max(np.abs(n1.H_np()))
    hr.build_problem(xs[1])
    o.build_problem(pr, xs[1])
    g, r = hr.eval_factors(xs[2])[0], o.eval_factors(pr, xs[2])[0]
    assert g.counts() == r.counts() and np.max(np.abs(g.H_np() - r.H_np())) <= 1e-9 * np.max(np.abs(r.H_np()))
    rg, Jg, _, kg = hr.eval_residuals(xs[2])
    ro, Jo, _, ko, _ = o.eval_residuals(xs[2])
    assert np.array_equal(kg, ko) and np.allclose(rg, ro, rtol=1e-9, atol=1e-9)
    hc.close()
    hr.close()


def test_refit_with_distinct_local_params(pkg, synth, abi, ob, scene_small):
    prob, meta = scene_small
    p = abi.reference_yaml_params(plane_cache=0)
    p.neigh_radius = 0.45
    p.neigh_max_pts = 20
    h = pkg.IbaHandle(prob, p)
    o = ob.Oracle(prob)
    c, n = h.eval_full(meta["x_gt"])
    oc, on = o.eval_cost(p, meta["x_gt"])[0], o.eval_normal(p, meta["x_gt"])[0]
    assert (c[0].cnt_3d_3d, c[0].valid_pl_3d_3d) == (oc.cnt_3d_3d, oc.valid_pl_3d_3d) and n[0].counts() == on.counts()
    assert np.max(np.abs(n[0].H_np() - on.H_np())) <= 1e-9 * np.max(np.abs(on.H_np()))
    h.close()
