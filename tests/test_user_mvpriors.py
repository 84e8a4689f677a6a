"""Joint (multivariate) user priors -- kabc_compile_mvprior_plugin, include/kabc.h: the reference hands ANY
Distribution to rand / logpdf as the prior (src/types.jl:30,34-35,52; src/smc.jl:92-93), a multivariate one
as well; here a joint density that is not a product of univariate ones is a C snippet with one logpdf and
one rand of the whole vector.  Shipped as snippets: Dirichlet, a Gaussian AR(1) process.

CPU: the oracle's evaluation of the snippets against scipy golden vectors (tests/golden/mvpriors_logpdf.json),
sampler moments.  GPU (-m gpu): the device evaluates the same bits; AIS and smc with such priors are
bit-exact against the oracle at 3 and 20 parameters (the run-time-dimension kernels beyond 16)."""
import json
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mvpriors_logpdf.json")


def _make(k, case):
    p = case["params"]
    if case["family"] == "Dirichlet":
        return k.Dirichlet(p["alpha"])
    return k.Ar1Normal(p["D"], p["mu"], p["sigma"], p["rho"])


def test_oracle_matches_scipy_golden(k, orc):
    cases = json.load(open(GOLDEN))["cases"]
    assert len(cases) == 6
    for c in cases:
        pri = _make(k, c)
        got = orc.factored_logpdf(pri, np.array(c["x"]))
        assert np.allclose(got, c["logpdf"], rtol=1e-12, atol=1e-10), (c["family"], c["params"])
    # outside the support of a Dirichlet: -Inf (off the simplex, a negative coordinate)
    d = k.Dirichlet([2.0, 3.0, 4.0])
    assert orc.factored_logpdf(d, [[0.2, 0.3, 0.6], [-0.1, 0.5, 0.6], [0.2, 0.3, 0.5]]).tolist()[:2] == [-np.inf, -np.inf]
    assert np.isfinite(orc.factored_logpdf(d, [[0.2, 0.3, 0.5]])[0])


def test_oracle_sampler_moments(k, orc):
    a = np.array([2.0, 3.0, 4.0, 1.5])
    x = orc.factored_rand(k.Dirichlet(a), 20000, seed=5)
    assert np.allclose(x.sum(1), 1.0, atol=1e-14) and (x > 0).all()
    assert np.allclose(x.mean(0), a / a.sum(), atol=0.01)
    v = a * (a.sum() - a) / (a.sum() ** 2 * (a.sum() + 1))
    assert np.allclose(x.var(0), v, rtol=0.06)
    ar = k.Ar1Normal(6, mu=1.0, sigma=0.5, rho=0.7)
    y = orc.factored_rand(ar, 40000, seed=6)
    assert np.allclose(y.mean(0), 1.0, atol=0.02)
    assert np.allclose(y.var(0), 0.25 / (1 - 0.49), rtol=0.05)
    assert np.allclose(np.corrcoef(y[:, 2], y[:, 3])[0, 1], 0.7, atol=0.02)
    assert np.allclose(np.corrcoef(y[:, 0], y[:, 2])[0, 1], 0.49, atol=0.02)


def test_joint_prior_does_not_mix(k):
    d = k.Dirichlet([1.0, 2.0, 3.0])
    mixed = k.Factored(d.p[0], d.p[1], k.Normal(0, 1))
    with pytest.raises((k.KabcError, ValueError)):
        mixed.logpdf([[0.2, 0.3, 0.5]])   # (resolved by the library: all D components or none)
    with pytest.raises(ValueError):
        k.UserMvPrior(k.Dirichlet.SOURCE, [(1.0, 2.0, 3.0, 4.0)])       # at most three parameters per component


@pytest.mark.gpu
def test_device_evaluates_the_same_bits(k, orc, gpu_ctx):
    for c in json.load(open(GOLDEN))["cases"]:
        pri = _make(k, c)
        x = np.array(c["x"])
        assert np.array_equal(pri.logpdf(x), orc.factored_logpdf(pri, x)), (c["family"], c["params"])
        assert np.array_equal(pri.rand(64, seed=3), orc.push_p(pri, orc.factored_rand(pri, 64, seed=3)))


@pytest.mark.gpu
@pytest.mark.parametrize("D", [3, 20])
@pytest.mark.parametrize("N", [64, 2000])     # the one-workgroup driver / a launch per half-generation
def test_ais_with_a_joint_prior_bit_exact(k, orc, gpu_ctx, D, N):
    """a latent AR(1) series under a Gaussian-distance cost: every move keeps the support"""
    pri = k.Ar1Normal(D, mu=0.2, sigma=0.8, rho=0.6)
    model = k.ApproxKernelizedPosterior(pri, k.costs.GaussDist(np.linspace(-0.5, 0.5, D)), 0.7)
    ens = k.AisEnsemble(model, N, seed=13).init()
    o = orc.OracleAIS(model, N, seed=13).init()
    for a, b in zip(ens.state()[:3], o.state()[:3]):
        assert np.array_equal(a, b)                                   # step(init): rand + logpdf of the joint prior
    assert np.array_equal(ens.advance(5, 3, collect=True), o.generations_sync(5, 3))
    assert ens.stats() == o.stats()


@pytest.mark.gpu
def test_ais_on_the_simplex_bit_exact(k, orc, gpu_ctx):
    """a Dirichlet prior: stretch and walk proposals are affine combinations of simplex points (they stay on
    sum(x) = 1 up to rounding), DE proposals leave it and are rejected by the prior -- as in the reference"""
    pri = k.Dirichlet([2.0, 3.0, 4.0])
    model = k.ApproxKernelizedPosterior(pri, k.costs.GaussDist([0.2, 0.3, 0.5]), 0.2)
    ens = k.AisEnsemble(model, 40, seed=4).init()
    o = orc.OracleAIS(model, 40, seed=4).init()
    got = ens.advance(30, 2, collect=True)
    assert np.array_equal(got, o.generations_sync(30, 2))
    st = ens.stats()
    assert st == o.stats() and 0 < st["accepted"] < st["proposals"]


@pytest.mark.gpu
@pytest.mark.parametrize("D,n", [(3, 100), (3, 3000), (20, 2500)])   # one workgroup / loop kernel / run-time dimension
def test_smc_with_a_joint_prior_bit_exact(k, orc, gpu_ctx, D, n):
    pri = k.Ar1Normal(D, mu=0.0, sigma=1.0, rho=0.5)
    cost = k.costs.GaussDist(np.full(D, 0.3))
    kw = dict(nparticles=n, alpha=0.9, epstol=0.4 * np.sqrt(D), seed=2)
    got = k.smc(pri, cost, return_array=True, **kw)
    ref = orc.smc(pri, cost, **kw)
    assert got.info["log"] == ref["log"] and got.eps == ref["eps"]
    assert np.array_equal(got.info["theta_all"], ref["theta_all"]) and np.array_equal(got.C, ref["C"])
