"""Search logic of the global-stage caller (csrc/iba_mads.hpp) on analytic black boxes, no GPU: convergence to known
optima, progressive barrier from an infeasible start, bounds, budget, determinism, batching."""
import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("spatial-temporal-lidar-camera-calibration_amd")

A = np.array([0.3, -0.2, 0.1, 0.25, -0.15, 0.05, 9.5])
X0 = np.array([0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 10.0])
BOX = dict(lb=X0 - 1.0, ub=X0 + 1.0, vns_max_idle=0)   # plain descent; the restarts have their own test


def test_smooth_bowl_converges_to_the_minimiser():
    x, r = pkg.mads_selftest(0, X0, **BOX)
    assert r.feasible == 1 and r.stop_reason == 1           # ran down to the minimum mesh size
    # at the minimum mesh size 1e-6 = 0.5 * 4^-l the poll radius is 0.5 * 2^-l ~ 5e-4: that is the resolution of the answer
    assert np.allclose(x, A, atol=1e-3) and r.f < 1e-5
    assert r.evaluations <= 5000 and r.batches <= r.iterations + 1   # one black-box call per iteration (+ x0)


def test_active_constraint_and_infeasible_start():
    x0 = X0.copy()
    x0[0] = 0.9                                              # violates c = x0 - 0.5 <= 0
    x, r = pkg.mads_selftest(1, x0, **BOX)
    assert r.feasible == 1 and r.c1 <= 0
    assert abs(x[0] - 0.5) < 1e-3 and np.allclose(x[1:], A[1:], atol=5e-3)   # active constraint: progress only along the feasible side
    assert abs(r.f - 0.25) < 1e-3


def test_nonsmooth_two_constraints():
    x, r = pkg.mads_selftest(2, X0, **BOX)
    assert r.feasible == 1 and r.c1 <= 0 and r.c2 <= 0
    # optimum: x1 pushed to 0.2 (c1 active); x3 + x4 <= 0.05 active
    assert x[1] >= 0.2 - 1e-12 and x[3] + x[4] <= 0.05 + 1e-12
    best = 0.4 + 0.1 * 0.45   # max |x - a| = 0.4 on x1, others free except the shared 0.05 budget on x3 + x4
    assert r.f <= best + 0.05


def test_bounds_budget_and_determinism():
    lb, ub = X0 - 1.0, X0 + 1.0
    ub[0] = 0.1                                              # minimiser a0 = 0.3 is outside: solution on the bound
    x, r = pkg.mads_selftest(0, X0, lb=lb, ub=ub, vns_max_idle=0)
    assert abs(x[0] - 0.1) < 1e-9 and np.all(x >= lb - 1e-15) and np.all(x <= ub + 1e-15)
    x2, r2 = pkg.mads_selftest(0, X0, lb=lb, ub=ub, vns_max_idle=0)
    assert np.array_equal(x, x2) and r.evaluations == r2.evaluations
    x3, r3 = pkg.mads_selftest(0, X0, lb=lb, ub=ub, seed=3, vns_max_idle=0)
    assert r3.evaluations != r.evaluations or not np.array_equal(x, x3)   # another Halton stream
    x4, r4 = pkg.mads_selftest(0, X0, max_bb_eval=200, **BOX)
    assert r4.evaluations <= 200 and r4.stop_reason == 2
    x5, r5 = pkg.mads_selftest(0, X0, bases_per_poll=1, speculative=0, **BOX)   # plain OrthoMADS 2N
    assert np.allclose(x5, A, atol=2e-3)
    with pytest.raises(pkg.IbaError):
        pkg.mads_selftest(0, X0, lb=ub, ub=lb)


def test_variable_neighbourhood_restarts_leave_local_basins():
    lb, ub = X0 - 0.5, X0 + 0.5
    f_local = []
    for idle in (0, 8):
        x, r = pkg.mads_selftest(3, X0, lb=lb, ub=ub, vns_max_idle=idle, max_bb_eval=200000)
        f_local.append(r.f)
        if idle:
            assert r.restarts >= 1 and r.stop_reason == 1
    assert f_local[1] < 0.5 * f_local[0]          # the restarts found a much deeper basin than the single descent
    x, r = pkg.mads_selftest(0, X0, lb=lb, ub=ub, vns_max_idle=3, max_bb_eval=200000)
    assert r.restarts >= 3 and r.stop_reason == 1 and np.allclose(x, A, atol=1e-3)   # the restarts end by themselves


# ---- the driver against its independent restatement (oracle/mads.py): identical evaluation sequences ----
def _diff_sequences(problem, x0, **opts):
    from oracle import mads as om
    x, r, tr = pkg.mads_selftest(problem, x0, trace=True, **opts)
    o = {k: (list(v) if k in ("lb", "ub", "init_frame") else v) for k, v in opts.items()}
    if "speculative" in o:
        o["speculative"] = bool(o["speculative"])
    ro, tro = om.minimize(list(x0), o, om.selftest_box(problem))
    assert len(tr) == r.evaluations == ro["evaluations"] == len(tro), (len(tr), r.evaluations, ro["evaluations"])
    ox = np.array([t[0] for t in tro])
    of = np.array([t[1] for t in tro])
    assert np.array_equal(tr[:, :7], ox), "first differing evaluation: %d" % int(np.argmax(np.any(tr[:, :7] != ox, axis=1)))
    assert np.array_equal(tr[:, 7], of)
    assert np.array_equal(x, np.array(ro["x"])) and r.f == ro["f"] and r.feasible == ro["feasible"]
    assert (r.iterations, r.batches, r.cache_hits, r.restarts, r.stop_reason) == (ro["iterations"], ro["batches"], ro["cache_hits"], ro["restarts"], ro["stop_reason"])
    return r


def test_iterates_equal_the_restatement_on_the_four_boxes():
    """Every point handed to the black box, in order, bit for bit: mesh / frame update, OrthoMADS directions (Halton stream),
    progressive barrier incl. the infeasible incumbent and the h_max update, speculative point, cache, budget cut."""
    x_inf = X0.copy()
    x_inf[0] = 0.9
    r = _diff_sequences(0, X0, **BOX)
    assert r.evaluations > 300
    r = _diff_sequences(1, x_inf, **BOX)                      # infeasible start: both incumbents live
    assert r.feasible == 1
    _diff_sequences(2, X0, **BOX)                             # two constraints active at the optimum
    _diff_sequences(0, X0, max_bb_eval=200, **BOX)            # the budget cuts a batch short
    _diff_sequences(0, X0, bases_per_poll=1, speculative=0, seed=3, **BOX)   # plain OrthoMADS 2N, another Halton stream
    lb, ub = X0 - 1.0, X0 + 1.0
    ub[0] = 0.1
    _diff_sequences(0, X0, lb=lb, ub=ub, vns_max_idle=0)      # the minimiser is outside the box


def test_restarts_equal_the_restatement():
    """Variable-neighbourhood restarts: shake amplitude k, idle counter, the 'better' rule — on the box with many local
    basins, where restarts do change the answer."""
    lb, ub = X0 - 0.5, X0 + 0.5
    r = _diff_sequences(3, X0, lb=lb, ub=ub, vns_max_idle=3, max_bb_eval=60000)
    assert r.restarts >= 3
    _diff_sequences(1, X0, lb=lb, ub=ub, vns_max_idle=2, max_bb_eval=60000)
