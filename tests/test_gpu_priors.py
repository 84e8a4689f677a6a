"""-m gpu: the device prior kernels (kabc_factored_*) against scipy golden
vectors, the reference's exact Factored tests, and the oracle (bit-exact)."""
import numpy as np
import pytest

from helpers import load_prior_golden, make_dist

pytestmark = pytest.mark.gpu


def test_reference_factored_testset_on_device(k, gpu_ctx):
    # test/runtests.jl:8-22
    d = k.Factored(k.Uniform(0, 1), k.Uniform(100, 101))
    s = d.rand(100, seed=3)
    assert np.all((s[:, 0] >= 0) & (s[:, 0] <= 1) & (s[:, 1] >= 100) & (s[:, 1] <= 101))
    assert d.pdf((0.0, 0.0)) == 0.0
    assert d.pdf((0.5, 100.5)) == 1.0
    assert d.logpdf((0.5, 100.5)) == 0.0
    assert d.logpdf((0.0, 0.0)) == -np.inf
    assert len(d) == 2
    m = k.Factored(k.Uniform(0.00, 1.0), k.DiscreteUniform(1, 2))
    sp = m.rand(seed=9)
    assert 0 <= sp[0] <= 1 and sp[1] in (1.0, 2.0)
    assert m.pdf(sp) == 0.5
    assert m.logpdf(sp) == pytest.approx(np.log(0.5), rel=1e-15)


def test_reference_push_testset_on_device(k, gpu_ctx):
    # test/runtests.jl:24-31
    assert list(k.Factored(k.Normal(), k.DiscreteUniform()).push_p([2, 1.0])) == [2.0, 1.0]
    assert list(k.Factored(k.DiscreteUniform(0, 9), k.DiscreteUniform(0, 9)).push_p(
        [2.5, 3.5])) == [2.0, 4.0]


@pytest.mark.parametrize("case", load_prior_golden(), ids=lambda c: f"{c['kind']}{c['params']}")
def test_device_logpdf_golden_and_oracle(k, orc, gpu_ctx, case):
    d = k.Factored(make_dist(k, case["kind"], case["params"]))
    x = case["x"].reshape(-1, 1)
    got = d.logpdf(x)
    ref = case["logpdf"]
    fin = np.isfinite(ref)
    assert np.array_equal(got[~fin], ref[~fin])
    assert np.allclose(got[fin], ref[fin], rtol=1e-12, atol=1e-12)   # vs scipy
    assert np.array_equal(got, orc.factored_logpdf(d, x))              # vs oracle: bit-exact


def test_device_rand_bit_exact_vs_oracle(k, orc, gpu_ctx):
    from kissabc_jl_amd import _cdefs as cd
    d = k.Factored(k.Uniform(1, 3), k.TruncatedNormal(0, 0.1, 0, 100), k.Beta(15, 2),
                   k.NegativeBinomial(900 / 195, (900 / 195) / (30 + 900 / 195)),
                   k.DiscreteUniform(1, 10), k.Normal(1, 0.5), k.Gamma(0.4, 3.0),
                   k.Exponential(2.0), k.LogNormal(0.3, 0.6), k.Beta(0.5, 0.7))
    got = d.rand(5000, seed=42)
    ref = orc.push_p(d, orc.factored_rand(d, 5000, seed=42, domain=cd.DOM_AIS_INIT))
    assert np.array_equal(got, ref)
