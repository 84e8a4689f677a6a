"""CPU: run-time compiled prior families (kabc_compile_prior_plugin; the reference's Factored
takes any UnivariateDistribution: src/priors.jl:11, logpdf :31-33, rand :43, push_p
src/types.jl:30-32).  The oracle runs the SAME snippet (gcc); its values are pinned by scipy
golden vectors (tests/golden/user_priors_logpdf.json + generator) and by the draws' distribution.
The hipRTC compilation itself is exercised here too: it needs no GPU."""
import numpy as np
import pytest
from scipy import stats

from helpers import load_prior_golden, make_dist


@pytest.mark.parametrize("case", load_prior_golden("user_priors_logpdf.json"),
                         ids=lambda c: f"{c['kind']}{c['params']}")
def test_user_family_logpdf_matches_scipy_golden(orc, k, case):
    d = make_dist(k, case["kind"], case["params"])
    assert d.kind >= 100
    got = orc.factored_logpdf(d, case["x"].reshape(-1, 1))
    ref = case["logpdf"]
    fin = np.isfinite(ref)
    assert np.array_equal(got[~fin], ref[~fin])
    assert np.allclose(got[fin], ref[fin], rtol=2e-12, atol=2e-12)


def test_user_family_samplers(orc, k):
    n = 40000
    x = orc.factored_rand(k.Laplace(0.5, 2.0), n, seed=3)[:, 0]
    assert stats.kstest(x, stats.laplace(0.5, 2.0).cdf).pvalue > 1e-3
    g = stats.gamma(2.0, scale=1.5)
    x = orc.factored_rand(k.Truncated(k.Gamma(2.0, 1.5), 0.5, 6.0), n, seed=4)[:, 0]
    assert x.min() >= 0.5 and x.max() <= 6.0
    cdf = lambda v: (g.cdf(v) - g.cdf(0.5)) / (g.cdf(6.0) - g.cdf(0.5))   # noqa: E731
    assert stats.kstest(x, cdf).pvalue > 1e-3
    for lam in (3.0, 40.0):
        x = orc.factored_rand(k.Poisson(lam), n, seed=5)[:, 0]
        assert np.array_equal(x, np.rint(x)) and x.min() >= 0
        assert abs(x.mean() - lam) < 5 * np.sqrt(lam / n) and abs(x.var() - lam) < 0.08 * lam


def test_user_family_mixes_with_builtin_components(orc, k):
    """Factored(Poisson, Normal, Laplace): push_p rounds the discrete family, logpdf is the
    left-to-right sum (src/priors.jl:30-36)."""
    d = k.Factored(k.Poisson(3.0), k.Normal(1.0, 0.5), k.Laplace(0.0, 1.0))
    x = np.array([[2.4, 1.2, -0.3], [3.5, 0.0, 2.0], [-0.6, 1.0, 0.0]])
    xp = orc.push_p(d, x)
    assert np.array_equal(xp[:, 0], [2.0, 4.0, -1.0]) and np.array_equal(xp[:, 1:], x[:, 1:])
    ref = (stats.poisson(3.0).logpmf(xp[:, 0]) + stats.norm(1.0, 0.5).logpdf(x[:, 1]) +
           stats.laplace(0.0, 1.0).logpdf(x[:, 2]))
    got = orc.factored_logpdf(d, xp)
    assert got[2] == -np.inf and np.allclose(got[:2], ref[:2], rtol=1e-12)


def test_snippet_errors_come_back_with_the_compilers_message(k):
    bad = "KABC_HD double kabc_user_prior_logpdf(double x, const double* p, const double* tab) { return nope; }"
    with pytest.raises(k.KabcError, match="nope"):
        k.UserPrior(bad)
    # the same family registered twice is one kind
    assert k.Poisson(2.0).kind == k.Poisson(7.0).kind
    with pytest.raises(ValueError, match="four parameters"):
        k.UserPrior(k.Poisson.SOURCE, params=(1, 2, 3, 4, 5))


def test_specialised_model_compiles_without_a_gpu(k, tmp_path, monkeypatch):
    """kabc_compile_model: the translation unit of ONE model (families and parameters as
    constants) goes through hipRTC and lands in the on-disk cache; a second request is a cache hit."""
    import glob
    import time
    monkeypatch.setenv("KABC_RTC_CACHE_DIR", str(tmp_path))
    prior = k.Factored(k.NegativeBinomial(4.6, 0.13), k.Beta(15, 2), k.Poisson(3.0))
    m = k.ApproxKernelizedPosterior(prior, k.costs.GaussDist([40.0, 0.8, 3.0]), 3.0)
    t0 = time.perf_counter()
    h = k.compile_model(m, families=1)
    t_first = time.perf_counter() - t0
    assert h > 0 and len(glob.glob(str(tmp_path / "kabc_*.co"))) == 1
    assert t_first < 20.0
    # pure boxes are left to the prebuilt kernels
    box = k.ApproxKernelizedPosterior(k.Factored(k.Uniform(0, 1), k.DiscreteUniform(1, 4)),
                                      k.costs.GaussDist([0.5, 2.0]), 1.0)
    assert k.compile_model(box) == 0


def test_truncated_gamma_sampler_small_alpha_and_narrow_windows(orc, k):
    """round-4 advisor finding: with alpha < 1 the Marsaglia-Tsang boost U^(1/alpha) must be drawn
    afresh for every proposal (one U reused over the retries weights it by 1 / P(accept | U):
    mean 0.951 against 0.872 for Truncated(Gamma(0.7, 2), 0, 3)); and a window of little mass must
    not put a point mass on its boundary: narrow ones take the uniform envelope, the rest is refused."""
    n = 200000
    for a, th, lo, hi, env in ((0.7, 2.0, 0.0, 3.0, "parent"), (0.4, 1.0, 0.05, 2.5, "parent"),
                               (2.0, 1.5, 7.0, 7.5, "uniform"), (0.6, 1.0, 2.0, 2.2, "uniform"),
                               (3.0, 1.0, 0.1, 0.6, "uniform")):
        d = k.Truncated(k.Gamma(a, th), lo, hi)
        assert d.envelope == env
        x = orc.factored_rand(d, n, seed=11)[:, 0]
        g = stats.gamma(a, scale=th)
        cdf = lambda v: (g.cdf(v) - g.cdf(lo)) / (g.cdf(hi) - g.cdf(lo))   # noqa: E731
        assert x.min() >= lo and x.max() <= hi
        assert stats.kstest(x, cdf).pvalue > 1e-3, (a, th, lo, hi)
        exact = g.expect(lambda v: v, lb=lo, ub=hi, conditional=True)
        assert abs(x.mean() - exact) < 5 * x.std() / np.sqrt(n), (a, th, lo, hi)
        assert np.mean(x == lo) + np.mean(x == hi) == 0.0      # no point mass on the boundary
    with pytest.raises(ValueError, match="neither rejection envelope"):
        k.Truncated(k.Gamma(2.0, 1.0), 8.0, 40.0)              # mass 3e-3, wide: inversion territory
