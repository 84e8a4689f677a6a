"""CPU: the Factored prior surface of the oracle against (a) every exact-value
test the reference holds (test/runtests.jl:8-31) and (b) scipy golden vectors."""
import numpy as np
import pytest
from scipy import stats

from helpers import load_prior_golden, make_dist


def test_reference_factored_testset(orc, k):
    # test/runtests.jl:8-22
    d = k.Factored(k.Uniform(0, 1), k.Uniform(100, 101))
    draws = orc.push_p(d, orc.factored_rand(d, 200, seed=3))
    assert np.all((draws[:, 0] >= 0) & (draws[:, 0] <= 1))
    assert np.all((draws[:, 1] >= 100) & (draws[:, 1] <= 101))
    assert orc.factored_pdf(d, (0.0, 0.0))[0] == 0.0
    assert orc.factored_pdf(d, (0.5, 100.5))[0] == 1.0
    assert orc.factored_logpdf(d, (0.5, 100.5))[0] == 0.0
    assert orc.factored_logpdf(d, (0.0, 0.0))[0] == -np.inf
    assert len(d) == 2
    m = k.Factored(k.Uniform(0.00, 1.0), k.DiscreteUniform(1, 2))
    s = orc.push_p(m, orc.factored_rand(m, 200, seed=4))
    assert np.all((s[:, 0] >= 0) & (s[:, 0] <= 1)) and set(np.unique(s[:, 1])) <= {1.0, 2.0}
    assert orc.factored_pdf(m, s[0])[0] == 0.5
    assert orc.factored_logpdf(m, s[0])[0] == pytest.approx(np.log(0.5), rel=1e-15)


def test_reference_push_testset(orc, k):
    # test/runtests.jl:24-31
    assert orc.push_p(k.Normal(), [1])[0, 0] == 1.0
    assert orc.push_p(k.DiscreteUniform(), [1.0])[0, 0] == 1
    assert list(orc.push_p(k.Factored(k.Normal(), k.DiscreteUniform()), [2, 1.0])[0]) == [2.0, 1]
    assert list(orc.push_p(k.Factored(k.Normal(), k.Normal()), [2, 1])[0]) == [2.0, 1.0]
    # push_p(Product([Normal(), Normal()]), [2, 1]) /′ [2.0, 1.0]  (vector-valued walker)
    assert list(orc.push_p(k.Product([k.Normal(), k.Normal()]), [2, 1])[0]) == [2.0, 1.0]
    # round(Int, x) is ties-to-even
    assert list(orc.push_p(k.Factored(k.DiscreteUniform(0, 9), k.DiscreteUniform(0, 9)),
                           [2.5, 3.5])[0]) == [2.0, 4.0]


@pytest.mark.parametrize("case", load_prior_golden(), ids=lambda c: f"{c['kind']}{c['params']}")
def test_logpdf_matches_scipy_golden(orc, k, case):
    d = make_dist(k, case["kind"], case["params"])
    got = orc.factored_logpdf(d, case["x"].reshape(-1, 1))
    ref = case["logpdf"]
    fin = np.isfinite(ref)
    assert np.array_equal(got[~fin], ref[~fin])
    # tolerance: 1e-12 absolute-or-relative (kabc_lgamma is ~4e-15 relative)
    assert np.allclose(got[fin], ref[fin], rtol=1e-12, atol=1e-12)


SAMPLERS = [
    ("Uniform", (1, 3), stats.uniform(1, 2)),
    ("Normal", (1, 0.5), stats.norm(1, 0.5)),
    ("TruncNormal", (0, 0.1, 0, 100), stats.truncnorm(0, 1000, loc=0, scale=0.1)),
    ("Beta", (15, 2), stats.beta(15, 2)),
    ("Beta", (0.5, 0.7), stats.beta(0.5, 0.7)),
    ("Exponential", (2.5,), stats.expon(scale=2.5)),
    ("Gamma", (0.4, 3.0), stats.gamma(0.4, scale=3.0)),
    ("Gamma", (7.5, 0.5), stats.gamma(7.5, scale=0.5)),
    ("LogNormal", (0.3, 0.6), stats.lognorm(0.6, scale=np.exp(0.3))),
]


@pytest.mark.parametrize("kind,params,ref", SAMPLERS, ids=lambda v: str(v)[:24])
def test_continuous_samplers_ks(orc, k, kind, params, ref):
    x = orc.factored_rand(make_dist(k, kind, params), 40000, seed=99)[:, 0]
    assert stats.kstest(x, ref.cdf).pvalue > 1e-3


@pytest.mark.parametrize("kind,params,ref", [
    ("DiscreteUniform", (1, 10), stats.randint(1, 11)),
    ("NegativeBinomial", (900 / 195, (900 / 195) / (30 + 900 / 195)), None),
    ("NegativeBinomial", (3.0, 0.6), None),
], ids=["du", "nb_socks", "nb_small"])
def test_discrete_samplers_chi2(orc, k, kind, params, ref):
    ref = ref or stats.nbinom(*params)
    n = 60000
    x = orc.factored_rand(make_dist(k, kind, params), n, seed=5)[:, 0]
    assert np.array_equal(x, np.rint(x)) and x.min() >= 0
    hi = int(ref.ppf(0.999))
    lo = int(ref.ppf(0.0005))
    edges = np.arange(lo, hi + 2)
    obs = np.histogram(x, bins=edges - 0.5)[0].astype(float)
    exp = n * ref.pmf(edges[:-1])
    keep = exp > 20
    chi2 = ((obs[keep] - exp[keep]) ** 2 / exp[keep]).sum()
    assert stats.chi2(keep.sum() - 1).sf(chi2) > 1e-4
    assert abs(x.mean() - ref.mean()) < 5 * ref.std() / np.sqrt(n)


def test_vector_valued_priors_are_products_of_their_components(orc, k):
    """Product([...]) and the diagonal MvNormal (test/runtests.jl:30,186): logpdf = sum of the
    components' log-densities (scipy), rand / push_p per component."""
    from scipy import stats
    mv = k.MvNormal(4, 1.0)                                  # MultivariateNormal(4, 1.0)
    assert len(mv) == 4 and mv.vector_valued and isinstance(mv, k.Factored)
    x = np.random.default_rng(0).normal(size=(50, 4)) * 2
    ref = stats.multivariate_normal(np.zeros(4), np.eye(4)).logpdf(x)
    assert np.allclose(orc.factored_logpdf(mv, x), ref, rtol=1e-13, atol=1e-13)
    d = k.MvNormal([1.0, -2.0, 0.5], [0.5, 2.0, 1.5])
    ref = stats.multivariate_normal([1.0, -2.0, 0.5], np.diag(np.array([0.5, 2.0, 1.5]) ** 2)).logpdf(x[:, :3])
    assert np.allclose(orc.factored_logpdf(d, x[:, :3]), ref, rtol=1e-13, atol=1e-13)
    pr = k.Product([k.Uniform(0, 2), k.DiscreteUniform(1, 4), k.Normal(0, 1)])
    xs = orc.push_p(pr, orc.factored_rand(pr, 100, seed=2))
    assert np.all((xs[:, 0] >= 0) & (xs[:, 0] <= 2)) and np.array_equal(xs[:, 1], np.rint(xs[:, 1]))
    assert k.MultivariateNormal is k.MvNormal


def _mv_cases():
    rng = np.random.default_rng(11)
    out = []
    for D in (1, 2, 3, 8, 16):
        A = rng.normal(size=(D, D))
        out.append((rng.normal(size=D) * 2, A @ A.T + 0.3 * np.eye(D)))
    out.append((np.array([1.0, -1.0]), np.array([[1.0, 0.999], [0.999, 1.0]])))   # nearly singular
    return out


@pytest.mark.parametrize("mu,cov", _mv_cases(), ids=lambda v: f"D{np.asarray(v).shape[0]}")
def test_full_covariance_mvnormal_against_scipy(orc, k, mu, cov):
    """MvNormal(μ, Σ) with a covariance matrix (the reference takes any Distribution as a prior:
    src/types.jl:30,34-35,52): log-density against scipy.stats.multivariate_normal, draws by their
    first two moments, push_p = identity, pdf = exp(logpdf)."""
    from scipy import stats
    d = k.MvNormal(mu, cov)
    D = len(mu)
    assert len(d) == D and d.vector_valued
    x = np.random.default_rng(1).normal(size=(300, D)) * 2 + mu
    ref = stats.multivariate_normal(mu, cov).logpdf(x).reshape(-1)
    got = orc.factored_logpdf(d, x)
    assert np.allclose(got, ref, rtol=1e-11, atol=1e-11)
    assert np.allclose(orc.factored_pdf(d, x), np.exp(ref), rtol=1e-10, atol=1e-300)
    assert np.array_equal(orc.push_p(d, x), x)
    n = 200000
    s = orc.factored_rand(d, n, seed=5)
    se = np.sqrt(np.diagonal(cov) / n)
    assert np.all(np.abs(s.mean(0) - mu) < 5 * se)
    C = np.cov(s.T).reshape(D, D)
    assert np.all(np.abs(C - cov) < 6 * np.sqrt((cov ** 2 + np.outer(np.diagonal(cov), np.diagonal(cov))) / n))


def test_mvnormal_argument_checks(k):
    assert "full covariance" not in repr(k.MvNormal([0.0, 1.0], [[4.0, 0.0], [0.0, 9.0]]))   # diagonal Σ: Normals
    assert [c.sigma for c in k.MvNormal([0.0, 1.0], [[4.0, 0.0], [0.0, 9.0]]).p] == [2.0, 3.0]
    with pytest.raises(k.KabcError, match="positive definite"):
        k.MvNormal(np.zeros(2), np.array([[1.0, 2.0], [2.0, 1.0]]))
    with pytest.raises(k.KabcError, match="symmetric"):
        k.MvNormal(np.zeros(2), np.array([[1.0, 0.5], [0.4, 1.0]]))
    with pytest.raises(ValueError, match="length"):
        k.MvNormal(np.zeros(17), np.eye(17) + 0.1)
