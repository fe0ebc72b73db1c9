"""The gate of the normal equations (H, b, cost, chi^2) since round 6: held to the TRUTH, not to the CPU oracle's own rounding.

Until round 5 the device's sums were compared with the oracle's double-precision duals, entry by entry, at 1e-10 of the entry. That gate measures
how alike two double evaluations round — and it parked a faster kernel: the Jacobian chain written with fused multiply-adds agrees with the
(unfused) double oracle to ~2e-10 instead of 2e-13, while being CLOSER to the exact value. Measured at the C2 shape (tools/entry_truth.py, six
candidates): the double oracle is 1.3e-10 .. 3.1e-9 per entry from a long-double evaluation of the same formulas (rows, Huber weights and sums in
x87 80-bit: Oracle.eval_normal_truth) — a few ill-conditioned plane factors amplify the last bits of any double evaluation —, the device
5e-11 .. 2.9e-9, nearer the truth than the double oracle on every one of them.

The bar therefore: with t the long-double value, o the double oracle and g the device,
  * worst relative error of g over the entries above 1e-6 of the largest  <=  max(1e-10, SLACK x the same figure of o): the device is within 1e-10
    of the exact value wherever the reference's own double arithmetic is, and never further from it than the reference's arithmetic by more than SLACK;
  * the small entries against the largest; cost and chi^2 likewise; every counter exactly.
Test infrastructure (imports nothing from the product)."""
import numpy as np

REL = 1e-10      # per entry, relative to the entry itself
FLOOR = 1e-6     # entries below this fraction of the largest are compared against the largest instead
SLACK = 1.5      # device error <= SLACK x the double oracle's own error against the long-double value


def worst_rel(a, t, floor=FLOOR):
    """worst |a - t| / |t| over the entries of t above floor * max|t|, and worst |a - t| / max|t| over the others"""
    a, t = np.asarray(a, np.float64).ravel(), np.asarray(t, np.float64).ravel()
    big = np.max(np.abs(t)) if t.size else 0.0
    if big == 0.0:
        return (float(np.max(np.abs(a))) if a.size else 0.0), 0.0
    m = np.abs(t) > floor * big
    w_big = float(np.max(np.abs(a - t)[m] / np.abs(t)[m])) if m.any() else 0.0
    w_small = float(np.max(np.abs(a - t)[~m]) / big) if (~m).any() else 0.0
    return w_big, w_small


def entries_vs_truth(g, o, t, what=""):
    """device g, double oracle o, long-double truth t (arrays of one shape)"""
    gb, gs = worst_rel(g, t)
    ob_, os_ = worst_rel(o, t)
    assert gb <= max(REL, SLACK * ob_), "%s: device %.2e from the long-double value, the double oracle %.2e" % (what, gb, ob_)
    assert gs <= max(REL * FLOOR * 10, SLACK * os_), "%s (small entries): device %.2e, the double oracle %.2e of the largest entry" % (what, gs, os_)
    return gb, ob_


def normal_vs_truth(g, o, t):
    """iba_normal_out-like objects (counts(), H_np(), b_np(), cost, chi2): device, double oracle, long-double oracle"""
    assert g.counts() == o.counts() == t.counts(), (g.counts(), o.counts())
    rH = entries_vs_truth(g.H_np(), o.H_np(), t.H_np(), "H")
    rb = entries_vs_truth(g.b_np(), o.b_np(), t.b_np(), "b")
    for k in ("cost", "chi2"):
        gv, ov, tv = getattr(g, k), getattr(o, k), getattr(t, k)
        assert abs(gv - tv) <= max(REL * abs(tv), SLACK * abs(ov - tv)), (k, gv, ov, tv)
    return {"H": rH, "b": rb}
