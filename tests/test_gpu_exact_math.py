"""The kernels replace `1.0f / a` and `sqrtf(x)` by short sequences (v_rcp_f32 / v_sqrt_f32 + exact fma residuals) that must
return the very bits of the IEEE-754 operations the reference's CPU code performs (IntersectTriangle's `f = 1.0 / a`,
pathtracer.cpp:384; glm::normalize's inversesqrt; the samplers' sqrt, :606-611, :734-739).  The proof is the enumeration of
all 2^32 inputs in tools/microbench/exact_math.hip (profiles/r02/exact_math.json); this test holds the helpers AS COMPILED
INTO libptk.so (ptk_probe_math) against the host's correctly rounded float32 division and square root on a few million
inputs: random bit patterns, the edges of each helper's domain and the inputs the enumeration singled out."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from pbrpathtracer_amd import ptk
    c = ptk.Context(0)
    yield c
    c.close()


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _same(got, want):
    """bit-identical, NaNs matching NaNs"""
    nan = np.isnan(want)
    return np.array_equal(np.isnan(got), nan) and np.array_equal(_bits(got)[~nan], _bits(want)[~nan])


def _random_floats(n, seed, lo_exp, hi_exp, signed=True):
    """uniform over bit patterns with biased exponent in [lo_exp, hi_exp]"""
    rng = np.random.default_rng(seed)
    e = rng.integers(lo_exp, hi_exp + 1, n, dtype=np.uint32)
    m = rng.integers(0, 1 << 23, n, dtype=np.uint32)
    s = rng.integers(0, 2, n, dtype=np.uint32) if signed else np.zeros(n, np.uint32)
    return ((s << 31) | (e << 23) | m).view(np.float32)


def test_short_reciprocal_is_the_ieee_quotient(ctx):
    x = np.concatenate([_random_floats(4_000_000, 1, 1, 252),
                        np.array([2.0 ** -126, -(2.0 ** -126), 2.0 ** 126, -(2.0 ** 126), 1.0, -1.0, 3.0, 1e-30, 1e30], np.float32),
                        (np.arange(1 << 16, dtype=np.uint32) + np.uint32(0x3F7F8000)).view(np.float32)])      # around 1.0
    with np.errstate(all="ignore"):
        want = (np.float32(1.0) / x).astype(np.float32)
    assert _same(ctx.probe_math(0, x), want)
    # ... and with the special cases the normalisations can meet: zeros, infinities, NaN
    y = np.concatenate([x[:500_000], np.array([0.0, -0.0, np.inf, -np.inf, np.nan], np.float32)])
    with np.errstate(all="ignore"):
        want = (np.float32(1.0) / y).astype(np.float32)
    assert _same(ctx.probe_math(1, y), want)


def test_short_square_root_is_the_ieee_root(ctx):
    edge = np.array([0x00000000, 0x00000001, 0x007fffff, 0x00800000, 0x0b6e9372, 0x0b6e9373, 0x0b7fffff, 0x0b800000, 0x0b800001,
                     0x0c7fffff, 0x0c800000, 0x3f800000, 0x3f7fffff, 0x3f800001, 0x7f7fffff, 0x7f800000, 0x7fc00000, 0x80000000,
                     0xbf800000], np.uint32).view(np.float32)
    x = np.concatenate([_random_floats(4_000_000, 2, 0, 254, signed=False),            # denormals and tiny values included
                        _random_floats(200_000, 3, 0, 30, signed=False),               # the range the guard sends to sqrtf
                        np.float32(1.0) - np.arange(1 << 16, dtype=np.float32) * np.float32(2.0 ** -24),      # 1 - w * w shapes
                        np.arange(1 << 16, dtype=np.float32) * np.float32(2.0 ** -24),                         # unit-interval draws
                        edge])
    with np.errstate(all="ignore"):
        want = np.sqrt(x).astype(np.float32)
    assert _same(ctx.probe_math(2, x), want)


def test_normalisation_factor_is_one_over_the_rounded_root(ctx):
    """glm::normalize = v * inversesqrt(dot(v, v)) with inversesqrt(x) = 1 / sqrt(x): two roundings, in that order"""
    x = np.concatenate([_random_floats(2_000_000, 4, 27, 247, signed=False), np.array([0.0, np.inf, 1.0, 4.0, 2.0], np.float32)])
    with np.errstate(all="ignore"):
        want = (np.float32(1.0) / np.sqrt(x).astype(np.float32)).astype(np.float32)
    assert _same(ctx.probe_math(3, x), want)
