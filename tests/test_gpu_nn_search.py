"""GPU parity of the kd search itself (iba_debug_nn = the search kernel's own per-lane search, lane_nn_visit, run to its end) against brute force
in float64 with the same expression order and the lowest-index tie rule. The search filters leaves in float32 and
confirms in double, so the queries are chosen to sit where a float cannot tell two points apart: exact midpoints of
point pairs (exact ties), midpoints moved by 1e-6 ... 1e-15 of the pair distance (near ties on both sides of what
float32 resolves), scan points themselves (distance 0), far-away queries, and random ones. Bar: index and squared
distance bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _brute(pts64, q):
    best_i = np.zeros(len(q), np.int64)
    best_d = np.zeros(len(q))
    for s in range(0, len(q), 256):
        qq = q[s:s + 256]
        dx = qq[:, None, 0] - pts64[None, :, 0]
        dy = qq[:, None, 1] - pts64[None, :, 1]
        dz = qq[:, None, 2] - pts64[None, :, 2]
        d2 = (dx * dx + dy * dy) + dz * dz                      # the reference's (and the kernel's) association order
        i = np.argmin(d2, axis=1)                               # first minimum = lowest index on exact ties
        best_i[s:s + 256] = i
        best_d[s:s + 256] = d2[np.arange(len(qq)), i]
    return best_i, best_d


def _queries(pts64, rng):
    n = len(pts64)
    a = rng.integers(0, n, 1500)
    nb = np.array([np.argsort(((pts64 - pts64[i]) ** 2).sum(1))[1] for i in a[:300]])     # true nearest neighbours of 300 points
    b = np.concatenate([nb, rng.integers(0, n, 1200)])
    mid = 0.5 * (pts64[a] + pts64[b])
    sep = pts64[b] - pts64[a]
    qs = [mid]                                                                            # exact ties between a and b (when nothing is closer)
    for eps in (1e-6, 1e-8, 1e-10, 1e-12, 1e-15):
        qs.append(mid + eps * sep)
        qs.append(mid - eps * sep)
    qs.append(pts64[rng.integers(0, n, 500)])                                             # distance exactly 0
    qs.append(pts64[rng.integers(0, n, 500)] + rng.normal(0, 1e-7, (500, 3)))             # inside the float rounding of the query
    lo, hi = pts64.min(0), pts64.max(0)
    qs.append(rng.uniform(lo, hi, (1500, 3)))
    qs.append(rng.uniform(lo - 30, hi + 30, (500, 3)))                                    # far outside the scan's box
    return np.concatenate(qs)


@pytest.mark.parametrize("mode", [1, 2, 3, 4])   # the query alone on either path of the kernel, and with a partner 1e-7 beside it
def test_kd_search_is_exact_where_float_is_blind(pkg, synth, abi, mode):
    prob, meta = synth.make_scene(n_frames=2, pts_per_frame=6000, n_keypoints=300, seed=41, new_mappoints=50, scan_kp=100)
    a = {k: v.copy() for k, v in prob.arrays.items()}
    pts = a["pts_xyz"].reshape(-1, 3)
    pts[6000 + 100:6000 + 400] = pts[6000 + 1000:6000 + 1300]          # frame 1 also carries 300 exact duplicates
    prob2 = abi.Problem(**a)
    h = pkg.IbaHandle(prob2, abi.reference_yaml_params())
    rng = np.random.default_rng(7 + mode)
    for f in (0, 1):
        pts64 = prob2.frame_points(f).astype(np.float64)
        q = _queries(pts64, rng)
        gi, gd = h.debug_nn(f, q, mode)
        bi, bd = _brute(pts64, q)
        assert np.array_equal(gd, bd), (f, np.flatnonzero(gd != bd)[:5])
        assert np.array_equal(gi.astype(np.int64), bi), (f, np.flatnonzero(gi != bi)[:5])
    h.close()
