"""GPU test of read_log (mchap_amd/csrc/read_log.hpp), the logarithm every likelihood kernel takes of its per-read terms
(assemble/likelihood.py:54-59: log of the mean haplotype probability of a read).  The kernels' parity tolerance on
log-likelihoods (1e-10 relative) only bounds it indirectly; this pins it directly: within one unit in the last place of
numpy's float64 log (and below one ulp of the long-double value) over 10^6 arguments, including subnormals, the values a
likelihood actually sees (products of a few probabilities), exact powers of two, 0, negatives and NaN."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _read_log(x):
    from mchap_amd import _lib

    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty_like(x)
    _lib.check(_lib.lib().mchap_read_log_batch(_lib.ptr(x), C.c_int64(x.size), _lib.ptr(out)))
    return out


def _ulps(a, b):
    """distance in representable doubles between a and b (same sign, finite)"""
    ia, ib = a.view(np.int64), b.view(np.int64)
    return np.abs(ia - ib)


def test_read_log_within_one_ulp_of_float64_log():
    rng = np.random.default_rng(7)
    parts = [
        rng.random(450_000),                                   # (0, 1): single probabilities
        np.exp(rng.uniform(-700.0, 700.0, 200_000)),           # the whole exponent range
        rng.random(150_000) * rng.random(150_000) * rng.random(150_000) / 4.0 + rng.random(150_000) / 4.0,  # means of products
        1.0 + rng.uniform(-1e-3, 1e-3, 100_000),               # around 1: cancellation in f = m - 1
        np.ldexp(rng.random(100_000), rng.integers(-1074, -1021, 100_000)),  # subnormals and the smallest normals
        np.ldexp(1.0, np.arange(-1074, 1024)).astype(np.float64),            # exact powers of two
        np.sqrt(0.5) * (1.0 + rng.uniform(-1e-12, 1e-12, 50_000)),           # the reduction's branch point
        np.array([np.nextafter(1.0, 0.0), 1.0, np.nextafter(1.0, 2.0), np.finfo(np.float64).tiny, np.finfo(np.float64).max, 5e-324]),
    ]
    x = np.concatenate(parts)
    x = x[x > 0]
    assert x.size >= 1_000_000
    got = _read_log(x)
    ref = np.log(x)
    assert np.isfinite(got).all()
    both_zero = (got == 0) & (ref == 0)
    d = _ulps(got, ref)
    d[both_zero] = 0
    assert d.max() <= 1, "max distance %d ulp at x = %r" % (d.max(), x[int(np.argmax(d))])
    # ... and below one ulp of the true value (long double on the host: 64-bit significand)
    sub = rng.choice(x.size, 200_000, replace=False)
    true = np.log(x[sub].astype(np.longdouble))
    err = np.abs(got[sub].astype(np.longdouble) - true) / np.spacing(np.abs(ref[sub])).astype(np.longdouble)
    err = err[np.abs(ref[sub]) > 0]
    assert float(err.max()) < 1.0, float(err.max())
    assert (d == 0).mean() > 0.9  # (it is the correctly rounded value for most arguments)


def test_read_log_special_values():
    got = _read_log(np.array([0.0, -0.0, -1.0, -np.inf, np.nan, -5e-324, 1.0]))
    assert got[0] == -np.inf and got[1] == -np.inf      # log(0) = -inf: a read no haplotype explains
    assert np.isnan(got[2:6]).all()
    assert got[6] == 0.0


def test_wave_sum_tree():
    """wave_sum (csrc/denovo_kernel.hpp): the sum over the 64 lanes of a wavefront is the XOR butterfly's tree -- offsets 32, 16,
    8, 4, 2, 1 -- whatever instructions carry it (round 4: DPP row rotations for the last four steps).  Emulated here in float64
    with numpy, operand for operand; the device value must have the same bits, in every lane."""
    from mchap_amd import _lib

    rng = np.random.default_rng(11)
    n = 4096
    x = np.concatenate([
        rng.standard_normal((n // 4, 64)) * np.exp(rng.uniform(-30, 30, (n // 4, 64))),   # wildly different magnitudes: every rounding shows
        -rng.random((n // 4, 64)) * 700.0,                                               # what a likelihood adds: c * log(p) <= 0
        np.where(rng.random((n // 4, 64)) < 0.3, 0.0, -rng.random((n // 4, 64))),        # padding lanes: +0.0 terms
        rng.integers(-3, 4, (n // 4, 64)).astype(np.float64) * 2.0 ** rng.integers(-40, 40, (n // 4, 64)),
    ])
    out = np.empty(n, dtype=np.float64)
    _lib.check(_lib.lib().mchap_wave_sum_batch(_lib.ptr(np.ascontiguousarray(x)), C.c_int64(n), _lib.ptr(out)))
    v = x.copy()
    lanes = np.arange(64)
    for o in (32, 16, 8, 4, 2, 1):
        v = v + v[:, lanes ^ o]
    assert not np.isnan(out).any()  # (NaN: the lanes of a wavefront disagreed)
    assert np.array_equal(out.view(np.uint64), v[:, 0].view(np.uint64))


@pytest.mark.parametrize("group", [2, 3, 4])
def test_read_log_product_is_the_sum_of_the_logarithms(group):
    """Round 5: where the read weights are 0 / 1 a lane takes ONE logarithm of the product of its (up to four) reads' terms --
    mantissas multiplied, exponents summed, so nothing underflows (read_log.hpp read_log_product).  Against the sum of the
    logarithms in long double: within two units in the last place of the result, over the range a likelihood sees -- products
    of a few to a hundred probabilities, i.e. terms down to 1e-300 whose product no double holds --, with padding reads (factor
    exactly 1) anywhere in the group, and 0 / NaN as the sum would give them."""
    from mchap_amd import _lib

    rng = np.random.default_rng(11 + group)
    n = 300_000
    x = np.concatenate([
        rng.random((n // 3, group)),                                        # single probabilities
        np.exp(rng.uniform(-690.0, 0.0, (n // 3, group))),                  # down to 1e-300: the plain product would underflow
        np.prod(rng.random((n // 3, group, 6)), axis=2) / 4.0 + 1e-9,       # means of products
    ])
    pad = rng.random(x.shape) < 0.15
    x[pad] = 1.0
    out = np.empty(len(x))
    _lib.check(_lib.lib().mchap_read_log_product_batch(_lib.ptr(np.ascontiguousarray(x)), C.c_int64(len(x)), group, _lib.ptr(out)))
    true = np.log(x.astype(np.longdouble)).sum(axis=1)
    ulp = np.spacing(np.abs(true.astype(np.float64))).astype(np.longdouble)
    # two units in the last place of the result, plus the roundings of the (up to three) multiplications as an ABSOLUTE error:
    # terms within 1e-14 of 1 lose relative accuracy in their product's logarithm, which is of no consequence in a sum of
    # per-read terms of magnitude 0.1 .. 700
    tol = 2.0 * ulp + np.longdouble(group - 1) * np.longdouble(2.0) ** -53
    err = np.abs(out.astype(np.longdouble) - true)
    assert bool((err <= tol).all()), float((err / tol).max())
    # a group of padding reads around one real read is that read's logarithm, bit for bit
    one = rng.random((1000, group))
    keep = rng.integers(0, group, 1000)
    mask = np.arange(group)[None, :] != keep[:, None]
    one[mask] = 1.0
    o1 = np.empty(1000)
    _lib.check(_lib.lib().mchap_read_log_product_batch(_lib.ptr(np.ascontiguousarray(one)), C.c_int64(1000), group, _lib.ptr(o1)))
    assert np.array_equal(o1, _read_log(one[np.arange(1000), keep]))
    # special values
    sp = np.ones((4, group))
    sp[0, 0] = 0.0
    sp[1, group - 1] = np.nan
    sp[2, :] = 0.5
    o2 = np.empty(4)
    _lib.check(_lib.lib().mchap_read_log_product_batch(_lib.ptr(sp), C.c_int64(4), group, _lib.ptr(o2)))
    assert o2[0] == -np.inf and np.isnan(o2[1]) and o2[3] == 0.0
    assert abs(o2[2] - group * np.log(0.5)) < 1e-15
