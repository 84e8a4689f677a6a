"""include/kabc_math.h on the device vs the host build of the same header (the oracle's
probe), bit for bit, function by function: the arithmetic contract that makes the
kernels reproducible on the CPU.  ~10^6 points per function plus the edge patterns."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
N = 1 << 20


def _same(orc, name, x):
    from kissabc_jl_amd import _lib
    got = _lib.math_probe(name, x)
    want = orc.math_vec(name, x)
    eq = (got == want) | (np.isnan(got) & np.isnan(want))
    assert eq.all(), (name, np.asarray(x).ravel()[np.argmin(eq.reshape(len(eq), -1).all(axis=1))])


def _pos(rng, lo, hi):
    return np.exp2(rng.uniform(lo, hi, N))


def test_log_exp_log1p_lgamma(orc):
    rng = np.random.default_rng(1)
    edge = np.array([0.0, -0.0, 1.0, np.inf, -1.0, np.nan, 5e-324, 2.2250738585072014e-308,
                     1.7976931348623157e308])
    _same(orc, "log", np.concatenate([_pos(rng, -1070, 1023), rng.uniform(0.5, 2.0, N), edge]))
    _same(orc, "log_pn", np.concatenate([_pos(rng, -1020, 1023), rng.uniform(0.0, 1.0, N) + 2.0 ** -53]))
    _same(orc, "exp", np.concatenate([rng.uniform(-750, 710, N), rng.normal(0, 1, N), edge, -edge]))
    xb = np.concatenate([rng.uniform(-700, 700, N), rng.normal(0, 1, N), [0.0, -0.0, 700.0, -700.0]])
    _same(orc, "exp_bounded", xb)
    assert np.array_equal(orc.math_vec("exp_bounded", xb), orc.math_vec("exp", xb))
    _same(orc, "log1p", np.concatenate([rng.uniform(-1, 1, N), _pos(rng, -60, 60), -_pos(rng, -60, 0), edge]))
    _same(orc, "lgamma", np.concatenate([_pos(rng, -20, 20), rng.uniform(0, 40, N), edge]))


def test_sqrt_rint_sincos(orc):
    rng = np.random.default_rng(2)
    edge = np.array([0.0, -0.0, np.inf, -1.0, np.nan, 5e-324, 2.2250738585072014e-308])
    _same(orc, "sqrt", np.concatenate([_pos(rng, -1070, 1023), edge]))
    # sqrt_pn: positive normal arguments well inside the exponent range, incl. the
    # Box-Muller range of -2 log u and exact squares (round-to-nearest ties cannot occur,
    # but results landing exactly on a double can)
    sq = np.floor(rng.uniform(1, 2 ** 26, N)) ** 2
    x = np.concatenate([_pos(rng, -700, 700), rng.uniform(2e-16, 80, N), sq,
                        np.nextafter(sq, np.inf), np.nextafter(sq, 0)])
    _same(orc, "sqrt_pn", x)
    from kissabc_jl_amd import _lib
    assert np.array_equal(_lib.math_probe("sqrt_pn", x), np.sqrt(x))   # correctly rounded
    _same(orc, "rint", np.concatenate([rng.uniform(-1e6, 1e6, N), np.arange(-50, 50) + 0.5, edge]))
    _same(orc, "sincos2pi", np.concatenate([rng.uniform(0, 1, N), np.arange(0, 9) / 8.0]))


def test_variates_from_bits(orc):
    rng = np.random.default_rng(3)
    bits = rng.integers(0, 2 ** 64, N, dtype=np.uint64)
    bits[:4] = [0, 2 ** 64 - 1, 2 ** 12 - 1, 2 ** 63]
    _same(orc, "u01", bits.view(np.float64))
    from kissabc_jl_amd import _lib
    u = _lib.math_probe("u01", bits.view(np.float64))
    k = (bits >> np.uint64(12)).astype(np.float64)      # exact: k < 2^52
    assert np.array_equal(u, (k + 0.5) * 2.0 ** -52) and u.min() > 0 and u.max() < 1
    _same(orc, "normal_pair", rng.integers(0, 2 ** 64, 2 * N, dtype=np.uint64).view(np.float64))
    n = rng.integers(1, 2 ** 32, N, dtype=np.uint64)
    n[:3] = [1, 2 ** 32 - 1, 2 ** 31]
    pairs = np.empty(2 * N)
    pairs[0::2] = bits.view(np.float64)
    pairs[1::2] = n.astype(np.float64)
    _same(orc, "index32", pairs)
    idx = _lib.math_probe("index32", pairs)
    want = np.array([(int(b) * int(m)) >> 64 for b, m in zip(bits[:20000], n[:20000])], dtype=np.float64)
    assert np.array_equal(idx[:20000], want)
