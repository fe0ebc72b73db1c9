"""The projections divide twice by one depth (iba_global.cpp:70-75: u = (K p).x / z, v = (K p).y / z; :308-313 the same for the covisible
reprojection). The kernels compute the refined reciprocal of the compiler's own f64 division sequence ONCE and finish the two quotients from
it (csrc/iba_kernels.hpp, div2) wherever no intermediate can leave the normal range, and take the compiler's division everywhere else. The
quotients must be the IEEE quotients: bit for bit against the device's plain division AND against numpy's on this host."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, np.float64).view(np.uint64)


def _check(pkg, n0, n1, d, min_fast_share):
    q0, q1, r0, r1, nf = pkg.debug_div2_selftest(n0, n1, d)
    with np.errstate(all="ignore"):
        c0, c1 = n0 / d, n1 / d
    for q, r, c in ((q0, r0, c0), (q1, r1, c1)):
        nan = np.isnan(c)
        assert np.array_equal(np.isnan(q), nan) and np.array_equal(np.isnan(r), nan)
        bad = np.flatnonzero((_bits(q) != _bits(r)) & ~nan)
        assert bad.size == 0, ("shared reciprocal vs plain division", bad[:5], n0[bad[:5]], n1[bad[:5]], d[bad[:5]])
        bad = np.flatnonzero((_bits(q) != _bits(c)) & ~nan)
        assert bad.size == 0, ("device vs host quotient", bad[:5])
    assert nf >= min_fast_share * len(d), (nf, len(d))
    return nf


def test_projection_sized_operands(pkg):
    """what a projection divides: numerators fx x + cx z of a few hundred pixel-metres, depths of 0.1 .. 100 m — all on the shared-reciprocal path"""
    rng = np.random.default_rng(5)
    n = 4_000_000
    z = np.exp(rng.uniform(np.log(0.05), np.log(200.0), n))
    fx, cx, cy = 718.856, 607.1928, 185.2157
    x, y = rng.normal(0, 20, n), rng.normal(0, 5, n)
    nf = _check(pkg, fx * x + cx * z, fx * y + cy * z, z, 0.999)
    assert nf == n
    # float32-born operands (scan points times a rotation: short significands, many exact quotients and ties to even)
    z = rng.uniform(0.1, 80, n).astype(np.float32).astype(np.float64)
    a = (rng.integers(-4000, 4000, n) * 0.25) * z
    b = rng.integers(-2**20, 2**20, n).astype(np.float64)
    _check(pkg, a, b, z, 0.99)


def test_random_bit_patterns_and_the_edges(pkg):
    """every exponent: random bit patterns (denormals, infinities, NaNs among them), operands around the two thresholds of the fast path
    (|den| = 2^-100, 2^100; |quotient| = 2^-700), zeros of both signs, quotients that overflow or underflow"""
    rng = np.random.default_rng(6)
    n = 2_000_000
    bits = rng.integers(0, 2**64, size=(3, n), dtype=np.uint64)
    n0, n1, d = (bits[i].view(np.float64) for i in range(3))
    _check(pkg, n0, n1, d, 0.0)
    # a moderate denominator with numerators of every exponent: the afterwards test (|q| >= 2^-700) decides
    d = np.exp2(rng.uniform(-110, 110, n)) * rng.choice([-1.0, 1.0], n)
    e = rng.uniform(-1074, 1023, n)
    n0 = np.ldexp(rng.uniform(1, 2, n), e.astype(np.int64)) * rng.choice([-1.0, 1.0], n)
    n1 = np.ldexp(rng.uniform(1, 2, n), rng.integers(-900, -600, n)) * d      # quotients around 2^-700
    nf = _check(pkg, n0, n1, d, 0.05)
    assert nf < n
    edge = np.array([0.0, -0.0, 5e-324, -5e-324, 2.2250738585072014e-308, 1.7976931348623157e308, np.inf, -np.inf, np.nan, 1.0, -1.0, 3.0, 1 / 3,
                     2.0**-100, np.nextafter(2.0**-100, 0), 2.0**100, np.nextafter(2.0**100, 0), 2.0**-700, np.nextafter(2.0**-700, 0), 2.0**-801, 2.0**-1022, 2.0**1023])
    A, B, D = np.meshgrid(edge, edge, edge, indexing="ij")
    _check(pkg, A.ravel(), B.ravel(), D.ravel(), 0.0)
