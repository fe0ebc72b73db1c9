"""iba_nn_list_kernel (round 6, opt-in: IBA_NN_LIST=1): the search kernel's pass over the anchored neighbour lists as a persistent grid
(csrc/iba_nn_list_kernel.hpp). Same searches, same result slots, same fixed-order sums: every number of an evaluation must equal the default
kernel's BIT FOR BIT — cost tuples, counts, normal equations — on groups of 8 and of 4 candidates, with entries left to the tree search,
for the cost alone (cost-path queries only), the normal equations alone (association-path queries only) and both."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _handle(pkg, prob, p, monkeypatch, on):
    monkeypatch.setenv("IBA_NN_LIST", "1" if on else "0")   # (a debug override: read only with IBA_DEBUG_ENV=1, which tests/conftest.py sets)
    try:
        return pkg.IbaHandle(prob, p)
    finally:
        monkeypatch.delenv("IBA_NN_LIST", raising=False)


def _tup(c):
    """a cost tuple as bytes-comparable values (IbaCostOut)"""
    return tuple(getattr(c, f) for f, _ in c._fields_)


def _same(a, b):
    assert a.counts() == b.counts(), (a.counts(), b.counts())
    assert np.array_equal(a.H_np(), b.H_np()) and np.array_equal(a.b_np(), b.b_np())
    assert a.cost == b.cost and a.chi2 == b.chi2


@pytest.mark.parametrize("n_cand", [64, 13, 6])
def test_same_bits_as_the_default_search_kernel(pkg, synth, abi, monkeypatch, n_cand):
    prob, meta = synth.make_scene(n_frames=24, pts_per_frame=5000, seed=12)
    p = abi.reference_yaml_params()
    xs = synth.perturb(meta["x_gt"], np.random.default_rng(1), n=n_cand)
    h1, h0 = _handle(pkg, prob, p, monkeypatch, True), _handle(pkg, prob, p, monkeypatch, False)
    c1, n1 = h1.eval_full(xs)
    assert h1.last_nn_list > 0
    c0, n0 = h0.eval_full(xs)
    assert h0.last_nn_list == 0
    for a, b in zip(n1, n0):
        _same(a, b)
    for a, b in zip(c1, c0):
        assert _tup(a) == _tup(b)
    # the cost alone (cost-path queries only) and the normal equations alone
    for a, b in zip(h1.eval_cost(xs), h0.eval_cost(xs)):
        assert _tup(a) == _tup(b)
    assert h1.last_nn_list > 0
    for a, b in zip(h1.eval_normal(xs), h0.eval_normal(xs)):
        _same(a, b)
    # far candidates: entries the lists cannot settle go to the tree search between two runs of the walk
    far = synth.perturb(meta["x_gt"], np.random.default_rng(3), rot=4e-3, trans=4e-2, n=n_cand)
    c1, n1 = h1.eval_full(far); c0, n0 = h0.eval_full(far)
    for a, b in zip(n1, n0):
        _same(a, b)
    for a, b in zip(c1, c0):
        assert _tup(a) == _tup(b)
    h1.close(); h0.close()


def test_a_sparse_scene_and_the_frozen_problem(pkg, synth, abi, monkeypatch):
    prob, meta = synth.make_scene(n_frames=40, pts_per_frame=3000, n_keypoints=200, seed=9)
    p = abi.reference_yaml_params()
    p.num_min_corr = 4
    xs = synth.perturb(meta["x_gt"], np.random.default_rng(2), n=16)
    h1, h0 = _handle(pkg, prob, p, monkeypatch, True), _handle(pkg, prob, p, monkeypatch, False)
    for a, b in zip(h1.eval_normal(xs), h0.eval_normal(xs)):
        _same(a, b)
    assert h1.last_nn_list > 0
    h1.build_problem(xs[1]); h0.build_problem(xs[1])
    for a, b in zip(h1.eval_factors(xs[:8]), h0.eval_factors(xs[:8])):
        _same(a, b)
    h1.close(); h0.close()


def test_the_block_shape_of_the_search_kernel_does_not_change_a_bit(pkg, synth, abi, monkeypatch):
    """iba_nn_kernel runs one-wave blocks of two candidates (small kd trees, 12 candidates or more per launch) or four-wave blocks of up to eight: the same searches, the
    same result slots, and sums whose order depends on the list positions alone — every number must be identical, here at 64, 13 (one-wave) and 6 (four-wave either way)"""
    prob, meta = synth.make_scene(n_frames=24, pts_per_frame=5000, seed=12)
    p = abi.reference_yaml_params()
    xs = synth.perturb(meta["x_gt"], np.random.default_rng(1), n=64)
    monkeypatch.setenv("IBA_NN_SMALL", "0")
    try:
        h4 = pkg.IbaHandle(prob, p)
    finally:
        monkeypatch.delenv("IBA_NN_SMALL", raising=False)
    h1 = pkg.IbaHandle(prob, p)
    for n in (64, 13, 6):
        c1, n1 = h1.eval_full(xs[:n]); c4, n4 = h4.eval_full(xs[:n])
        assert h1.last_nn_threads == (64 if n >= 12 else 256) and h4.last_nn_threads == 256
        for a, b in zip(n1, n4):
            _same(a, b)
        for a, b in zip(c1, c4):
            assert _tup(a) == _tup(b)
        for a, b in zip(h1.eval_cost(xs[:n]), h4.eval_cost(xs[:n])):
            assert _tup(a) == _tup(b)
    far = synth.perturb(meta["x_gt"], np.random.default_rng(3), rot=4e-3, trans=4e-2, n=32)   # entries left to the tree search
    for a, b in zip(h1.eval_normal(far), h4.eval_normal(far)):
        _same(a, b)
    h1.close(); h4.close()
