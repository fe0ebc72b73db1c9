"""Final calibrated SE(3): iba_calibrate_lm on the device path vs the same LM (tests/lm_ref.py) driven by the CPU
oracle, on the same synthetic problem. north_star tolerance: 1e-4 rad / 1e-3 m."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import lm_ref  # noqa: E402

pytestmark = pytest.mark.gpu


def test_final_se3_matches_cpu_path(pkg, synth, abi, ob, scene_small):
    prob, meta = scene_small
    p = abi.reference_yaml_params()
    h = pkg.IbaHandle(prob, p)
    o = ob.Oracle(prob)
    x0 = synth.perturb(meta["x_gt"], np.random.default_rng(5), rot=1e-3, trans=0.01, scale_rel=3e-3, n=1)[0]

    def ev(x):
        n = o.eval_factors(p, x)[0]
        return n.H_np(), n.b_np(), n.cost

    xc, sc = lm_ref.calibrate_lm(x0, lambda x: o.build_problem(p, x), ev, max_outer=6)
    xg, rg = h.calibrate_lm(x0, max_outer_iterations=6)
    rot, trans, scl = lm_ref.se3_error(xg, xc, synth.sim3_exp)
    assert rot < 1e-4 and trans < 1e-3 and scl < 1e-4, (rot, trans, scl)
    assert rg.outer_iterations == sc["outer"] and abs(rg.final_cost - sc["final_cost"]) <= 1e-6 * sc["final_cost"]
    assert rg.final_cost < 0.6 * rg.initial_cost          # the solve does reduce the robustified cost
    h.close()
