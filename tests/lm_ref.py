"""TEST INFRASTRUCTURE: the same outer-loop + Ceres-style LM as csrc/iba_lm.hpp, in numpy, driven by any
evaluator (the oracle in the tests). Used to check the final calibrated SE(3) of the device path."""
import numpy as np


def calibrate_lm(x0, build, evalf, max_outer=30, max_inner=30, min_diff=1e-6, ftol=1e-6, gtol=1e-10, ptol=1e-8, radius0=1e4):
    x = np.array(x0, dtype=np.float64)
    last = x.copy()
    stats = dict(outer=0, inner=0, evals=0, converged=False)
    for outer in range(max_outer):
        build(x)
        H, g, cost = evalf(x)
        stats["evals"] += 1
        if outer == 0:
            stats["initial_cost"] = cost
        radius, dec = radius0, 2.0
        scale = 1.0 / (1.0 + np.sqrt(np.maximum(np.diag(H), 0.0)))
        for it in range(max_inner):
            stats["inner"] += 1
            if np.max(np.abs(g)) <= gtol:
                break
            Hs = scale[:, None] * H * scale[None, :]
            gs = scale * g
            A = Hs + np.diag(np.clip(np.diag(Hs), 1e-6, 1e32) / radius)
            try:
                L = np.linalg.cholesky(A)
                ds = -np.linalg.solve(L.T, np.linalg.solve(L, gs))
                model = -ds @ (gs + 0.5 * Hs @ ds)
                ok = model > 0
            except np.linalg.LinAlgError:
                ok = False
            if not ok:
                radius = max(1e-32, radius / dec)
                dec *= 2
                if radius <= 1e-32:
                    break
                continue
            d = scale * ds
            xn = x + d
            Hn, gn, cn = evalf(xn)   # Ceres order: candidate first, tolerance tests before acceptance (x is kept on a stop)
            stats["evals"] += 1
            if np.linalg.norm(d) <= ptol * (np.linalg.norm(x) + ptol):
                break
            if abs(cost - cn) <= ftol * cost:
                break
            rho = (cost - cn) / model
            if rho > 1e-3:
                x, H, g, cost = xn, Hn, gn, cn
                t = 2.0 * rho - 1.0
                radius = min(1e16, radius / max(1.0 / 3.0, 1.0 - t ** 3))
                dec = 2.0
            else:
                radius = max(1e-32, radius / dec)
                dec *= 2
                if radius <= 1e-32:
                    break
        stats["final_cost"] = cost
        stats["outer"] = outer + 1
        if np.all(np.abs(last - x) <= min_diff):
            stats["converged"] = True
            break
        last = x.copy()
    return x, stats


def se3_error(xa, xb, sim3_exp):
    """(rotation angle [rad], translation distance [m], relative scale difference) between two x."""
    Ra, ta, sa = sim3_exp(xa)
    Rb, tb, sb = sim3_exp(xb)
    c = np.clip((np.trace(Ra.T @ Rb) - 1) / 2, -1, 1)
    return float(np.arccos(c)), float(np.linalg.norm(ta - tb)), float(abs(sa - sb) / abs(sb))
