"""The multiply-high division used on the device since round 3 (GemmParams::nbn_magic, conv_stack's qdiv, im2col2d's mdiv):
q = (n * (floor(2^32 / d) + 1)) >> 32.  The kernels rely on it being EXACT for every n with n * d < 2^32 (the launchers check
that range and otherwise take the dividing path); this checks the arithmetic itself, exhaustively over the ranges the model's
shapes produce and at the edge of the stated condition."""
import numpy as np


def magic(d):
    return (1 << 32) // d + 1


def test_multiply_high_equals_floor_division_inside_the_stated_range():
    n = np.arange(0, 1 << 17, dtype=np.uint64)
    for d in list(range(2, 700)) + [768, 1024, 2048, 4096, 4801, 16064, 32768]:
        m = np.uint64(magic(d))
        ok = n * np.uint64(d) < np.uint64(1 << 32)
        q = (n * m) >> np.uint64(32)
        assert np.array_equal(q[ok], (n // np.uint64(d))[ok]), d


def test_the_largest_dividends_the_condition_admits():
    rng = np.random.default_rng(0)
    for d in (2, 3, 5, 7, 9, 24, 33, 72, 257, 576, 1000, 4097, 65535):
        m = np.uint64(magic(d))
        top = ((1 << 32) - 1) // d                      # the largest n with n * d < 2^32
        n = np.concatenate([np.arange(max(0, top - 5000), top + 1, dtype=np.uint64),
                            rng.integers(0, top + 1, 20000).astype(np.uint64)])
        assert np.array_equal((n * m) >> np.uint64(32), n // np.uint64(d)), d


def test_the_condition_is_not_vacuous():
    # beyond n * d < 2^32 the identity does fail somewhere (so the launchers' range checks matter)
    d = 7
    m = np.uint64(magic(d))
    n = np.arange((1 << 32) - 200000, (1 << 32) - 1, dtype=np.uint64)
    assert not np.array_equal((n * m) >> np.uint64(32), n // np.uint64(d))
