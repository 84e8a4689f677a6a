"""-m gpu: run-time compiled prior families (kabc_compile_prior_plugin) and kernels specialised
for one model (kabc_compile_model) through the C ABI, against the oracle running the SAME
snippets.  Bar: BIT-EXACT.  The reference's Factored takes any UnivariateDistribution
(src/priors.jl:11, :31-33, :43; src/types.jl:30-32)."""
import numpy as np
import pytest

from helpers import load_prior_golden, make_dist

pytestmark = pytest.mark.gpu

NOISY = """
KABC_HD double kabc_user_cost(const double* x, int D, const double* params, const double* data,
                              int64_t ndata, kabc_cost_rng_t* rng) {
    double z0, z1, s = 0.0;
    kabc_cost_rng_normal2(rng, &z0, &z1);
    for (int k = 0; k < D; ++k) { const double d = x[k] - params[k]; s += d * d; }
    return kabc_sqrt(s) + 0.01 * kabc_fabs(z0);
}
"""


def _user_prior(k):
    return k.Factored(k.Poisson(3.0), k.Laplace(0.5, 2.0), k.Truncated(k.Gamma(2.0, 1.5), 0.5, 6.0),
                      k.Normal(1.0, 0.5))


@pytest.mark.parametrize("case", load_prior_golden("user_priors_logpdf.json"),
                         ids=lambda c: f"{c['kind']}{c['params']}")
def test_device_user_family_logpdf_golden_and_oracle(k, orc, gpu_ctx, case):
    d = k.Factored(make_dist(k, case["kind"], case["params"]))
    x = case["x"].reshape(-1, 1)
    got = d.logpdf(x)
    ref = case["logpdf"]
    fin = np.isfinite(ref)
    assert np.array_equal(got[~fin], ref[~fin])
    assert np.allclose(got[fin], ref[fin], rtol=2e-12, atol=2e-12)   # vs scipy
    assert np.array_equal(got, orc.factored_logpdf(d, x))              # vs oracle: bit-exact


def test_device_user_family_rand_and_push_p_bit_exact(k, orc, gpu_ctx):
    from kissabc_jl_amd import _cdefs as cd
    d = k.Factored(k.Poisson(3.0), k.Poisson(40.0), k.Laplace(0.5, 2.0),
                   k.Truncated(k.Gamma(2.0, 1.5), 0.5, 6.0), k.Truncated(k.Gamma(0.7, 2.0), 0.0, 3.0),
                   k.Beta(15, 2), k.Uniform(1, 3))
    got = d.rand(4000, seed=42)
    ref = orc.push_p(d, orc.factored_rand(d, 4000, seed=42, domain=cd.DOM_AIS_INIT))
    assert np.array_equal(got, ref)
    x = np.random.default_rng(0).normal(size=(100, len(d))) * 3
    assert np.array_equal(d.push_p(x), orc.push_p(d, x))
    assert np.array_equal(d.logpdf(d.push_p(x)), orc.factored_logpdf(d, orc.push_p(d, x)))


def _ais_check(k, orc, model, N, nt=5, gens=3, seed=11):
    ens = k.AisEnsemble(model, N, seed=seed).init()
    o = orc.OracleAIS(model, N, seed=seed).init()
    x0, lp0, ll0, _ = ens.state()
    xo, lpo, llo, _ = o.state()
    assert np.array_equal(x0, xo) and np.array_equal(lp0, lpo) and np.array_equal(ll0, llo)
    ens.set_debug(nt)
    got = ens.advance(1, nt, collect=True)
    dbg = ens.get_debug(nt)
    ref, tr = o.generations_sync(1, nt, trace=True)
    assert np.array_equal(got, ref)
    for col in (0, 1, 5):
        assert np.array_equal(dbg[:, :, col], tr[0, :, :, col])
    ens.set_debug(0)
    assert np.array_equal(ens.advance(gens, nt, collect=True), o.generations_sync(gens, nt))
    xs, lps, lls, t = ens.state()
    xo, lpo, llo, to = o.state()
    assert t == to
    assert np.array_equal(xs, xo) and np.array_equal(lps, lpo) and np.array_equal(lls, llo)
    assert ens.stats() == o.stats()
    return ens


@pytest.mark.parametrize("posterior", ["kernelized", "threshold"])
def test_ais_with_user_families_bit_exact(k, orc, gpu_ctx, posterior):
    prior = _user_prior(k)
    cost = k.costs.GaussDist([3.0, 0.0, 2.0, 1.0])
    model = (k.ApproxKernelizedPosterior(prior, cost, 1.5) if posterior == "kernelized"
             else k.ApproxPosterior(prior, cost, 4.0))
    _ais_check(k, orc, model, 333)


def test_ais_user_families_with_user_cost_bit_exact(k, orc, gpu_ctx):
    """both halves of the model compiled at run time: a user cost AND user prior families"""
    prior = _user_prior(k)
    cost = k.costs.UserCost(NOISY, dims=[4], params=[3.0, 0.0, 2.0, 1.0], name="noisy_dist")
    orc.register_user_cost(cost)
    _ais_check(k, orc, k.ApproxKernelizedPosterior(prior, cost, 1.5), 200)


@pytest.mark.parametrize("path", ["loop", "kernels"])
def test_smc_with_user_families_bit_exact(k, orc, gpu_ctx, monkeypatch, path):
    monkeypatch.setenv("KABC_SMC_LOOP", "1" if path == "loop" else "0")
    prior = _user_prior(k)
    cost = k.costs.GaussDist([3.0, 0.0, 2.0, 1.0])
    kw = dict(nparticles=1500, alpha=0.9, epstol=0.3, mcmc_retrys=1)
    got = k.smc(prior, cost, seed=5, return_array=True, **kw)
    ref = orc.smc(prior, cost, seed=5, **kw)
    assert got.info["iterations"] == ref["iterations"] > 3 and got.info["log"] == ref["log"]
    assert got.eps == ref["eps"] and np.array_equal(got.info["theta_all"], ref["theta_all"])
    assert np.array_equal(got.C, ref["C"]) and np.array_equal(got.info["alive"], ref["alive"])
    assert got.info["cost_evals"] == ref["cost_evals"]


def test_abcde_and_pfilter_with_user_families_bit_exact(k, orc, gpu_ctx):
    prior = _user_prior(k)
    cost = k.costs.GaussDist([3.0, 0.0, 2.0, 1.0])
    got = k.ABCDE(prior, cost, 0.5, seed=9, return_array=True, nparticles=300, generations=25)
    ref = orc.abcde(prior, cost, 0.5, seed=9, nparticles=300, generations=25)
    assert np.array_equal(got.P, ref["P"]) and np.array_equal(got.C, ref["C"])
    assert got.info["nsims"] == ref["nsims"]
    got = k.pfilter(prior, cost, 300, seed=4, return_array=True, max_iters=12)
    ref = orc.pfilter(prior, cost, 300, seed=4, max_iters=12)
    assert np.array_equal(got.P, ref["P"]) and np.array_equal(got.C, ref["C"])
    assert got.info["nreps"] == ref["nreps"]


def test_unregistered_kind_and_long_priors_fail_loudly(k, gpu_ctx):
    from kissabc_jl_amd import distributions as ds

    class Bogus(ds.UnivariateDistribution):
        kind = 250

        def params(self):
            return (1.0,)

    with pytest.raises(k.KabcError, match="invalid prior|not a registered"):
        k.AisEnsemble(k.ApproxKernelizedPosterior(k.Factored(Bogus(), k.Normal(0, 1)),
                                                  k.costs.GaussDist([0.0, 0.0]), 1.0), 64)
    # (17 components with a user family: the run-time-dimension kernels, see below; the device
    # path's own bound is KABC_MAX_DIM_DYN = 256 for every prior)
    with pytest.raises(ValueError, match="256"):
        k.Factored(*([k.Poisson(3.0)] + [k.Normal(0, 1)] * 256))


# ---- kernels specialised for one model (kabc_compile_model) --------------------------------
def _spec_models(k):
    rng = np.random.default_rng(7)
    socks = k.Factored(k.NegativeBinomial(900 / 195, (900 / 195) / (30 + 900 / 195)), k.Beta(15, 2))
    four = k.Factored(k.Gamma(2.5, 0.7), k.LogNormal(0.3, 0.6), k.Beta(2.0, 3.0), k.Exponential(2.0))
    H16 = k.Factored(k.Normal(0, 5), k.Uniform(0, 5), *[k.Normal(0, 1)] * 14)
    N2 = k.Factored(k.Normal(0, 5), k.Normal(0, 5))
    mixed = k.Factored(k.NegativeBinomial(3.0, 0.4), k.NegativeBinomial(7.5, 0.2), k.NegativeBinomial(2.0, 0.5),
                       k.DiscreteUniform(1, 10), k.TruncatedNormal(0, 1, -1, 2), k.Poisson(3.0))
    return {
        "socks": (k.ApproxKernelizedPosterior(socks, k.costs.GaussDist([40.0, 0.8]), 3.0), 400, 9),
        "four_family_d4": (k.ApproxKernelizedPosterior(four, k.costs.NormShell(2.0), 0.5), 333, 5),
        "c4_prior_d16": (k.ApproxKernelizedPosterior(H16, k.costs.HierGaussSim(rng.normal(size=14)), 0.3), 640, 5),
        "normal_d8": (k.ApproxKernelizedPosterior(k.Factored(*[k.Normal(0.5 * j, 1.0 + j) for j in range(8)]),
                                                  k.costs.GaussDist(np.arange(8.0) / 4), 0.5), 1024, 5),
        "threshold_mixed_user": (k.ApproxPosterior(mixed, k.costs.NormShell(8.0), 6.0), 300, 9),
    }


@pytest.mark.parametrize("name", ["socks", "four_family_d4", "c4_prior_d16", "normal_d8",
                                  "threshold_mixed_user"])
def test_specialised_ais_kernels_bit_exact(k, orc, gpu_ctx, monkeypatch, name):
    """the same model on the prebuilt kernels, on its specialised kernels and on the oracle"""
    model, N, nt = _spec_models(k)[name]
    has_user = any(c.kind >= 100 for c in model.prior.p)
    monkeypatch.setenv("KABC_SPECIALIZE", "0")   # (the default would specialise on its own)
    base = None if has_user else k.AisEnsemble(model, N, seed=3).init().advance(3, nt, collect=True)
    monkeypatch.delenv("KABC_SPECIALIZE")
    h = k.compile_model(model, families=1)
    assert h > 0
    try:
        ens = _ais_check(k, orc, model, N, nt=nt, gens=2, seed=3)
        ens.close()
        if base is not None:
            e2 = k.AisEnsemble(model, N, seed=3).init()
            assert e2.spec_state() == ("active", 0)
            assert np.array_equal(e2.advance(3, nt, collect=True), base)
    finally:
        k._lib.check(k._lib.load().kabc_model_release(h))


@pytest.mark.parametrize("path", ["loop", "kernels"])
def test_specialised_smc_kernels_bit_exact(k, orc, gpu_ctx, monkeypatch, path):
    monkeypatch.setenv("KABC_SMC_LOOP", "1" if path == "loop" else "0")
    rng = np.random.default_rng(1)
    zstar = rng.normal(size=14)
    ybar = 1.0 + 0.5 * zstar + rng.normal(size=14) / np.sqrt(8)
    H16 = k.Factored(k.Normal(0, 5), k.Uniform(0, 5), *[k.Normal(0, 1)] * 14)
    socks = k.Factored(k.NegativeBinomial(900 / 195, (900 / 195) / (30 + 900 / 195)), k.Beta(15, 2))
    for prior, cost, kw in [(H16, k.costs.HierGaussSim(ybar), dict(nparticles=4096, alpha=0.95, epstol=0.05)),
                            (socks, k.costs.GaussDist([40.0, 0.8]), dict(nparticles=1000, alpha=0.9, epstol=0.5))]:
        monkeypatch.setenv("KABC_SPECIALIZE", "0")
        base = k.smc(prior, cost, seed=5, return_array=True, **kw)
        monkeypatch.delenv("KABC_SPECIALIZE")
        h = k.compile_model(prior, cost, families=2)
        assert h > 0
        try:
            got = k.smc(prior, cost, seed=5, return_array=True, **kw)
        finally:
            k._lib.check(k._lib.load().kabc_model_release(h))
        ref = orc.smc(prior, cost, seed=5, **kw)
        for r in (base, got):
            assert r.info["iterations"] == ref["iterations"] and r.info["log"] == ref["log"]
            assert r.eps == ref["eps"] and np.array_equal(r.info["theta_all"], ref["theta_all"])
            assert np.array_equal(r.C, ref["C"])


def test_small_all_normal_priors_stay_on_the_prebuilt_class(k, gpu_ctx):
    """plain Normals up to seven parameters: the prebuilt NORMAL class is the faster kernel (C2)"""
    N2 = k.Factored(k.Normal(0, 5), k.Normal(0, 5))
    model = k.ApproxKernelizedPosterior(N2, k.costs.GaussDist([1.0, -0.5]), 0.1)
    assert k.compile_model(model, families=1) == 0
    assert k.AisEnsemble(model, 256, seed=1).init().spec_state() == ("none", -1)


def test_specialize_env_and_release(k, orc, gpu_ctx, monkeypatch):
    """KABC_SPECIALIZE=1: every entry point specialises at first sight of a model; =0 never."""
    four = k.Factored(k.Gamma(2.5, 0.7), k.LogNormal(0.3, 0.6), k.Beta(2.0, 3.0), k.Exponential(2.0))
    model = k.ApproxKernelizedPosterior(four, k.costs.NormShell(2.5), 0.5)
    ref = orc.OracleAIS(model, 200, seed=8).init().generations_sync(3, 4)
    for v in ("1", "0"):
        monkeypatch.setenv("KABC_SPECIALIZE", v)
        assert np.array_equal(k.AisEnsemble(model, 200, seed=8).init().advance(3, 4, collect=True), ref)


# ---- beyond 16 parameters: the run-time-dimension kernels compiled with the snippets --------
def _big_prior(k, D):
    comps = [k.Poisson(3.0), k.Normal(0, 2), k.Laplace(0.0, 1.5), k.Uniform(-3, 3), k.Truncated(k.Gamma(2.0, 1.5), 0.5, 6.0)]
    return k.Factored(*[comps[j % len(comps)] for j in range(D)])


@pytest.mark.parametrize("D", [17, 40])
def test_user_families_beyond_16_parameters_bit_exact(k, orc, gpu_ctx, D):
    """The reference's Factored has no bound on its length and takes any UnivariateDistribution
    (src/priors.jl:11): user families with length(prior) > KABC_MAX_DIM run on the
    run-time-dimension kernels, compiled by hipRTC with the families' snippets -- AIS, smc, ABCDE
    and pfilter, bit-exact against the oracle running the same text."""
    prior = _big_prior(k, D)
    target = np.where(np.arange(D) % 5 == 0, 3.0, np.where(np.arange(D) % 5 == 4, 2.0, 0.3))
    cost = k.costs.GaussDist(target)
    model = k.ApproxKernelizedPosterior(prior, cost, 2.0)
    N, nt, seed = 3 * D + 50, 3, 6
    ens = k.AisEnsemble(model, N, seed=seed).init()
    o = orc.OracleAIS(model, N, seed=seed).init()
    x0, lp0, ll0, _ = ens.state()
    xo, lpo, llo, _ = o.state()
    assert np.array_equal(x0, xo) and np.array_equal(lp0, lpo) and np.array_equal(ll0, llo)
    assert np.array_equal(ens.advance(3, nt, collect=True), o.generations_sync(3, nt))
    assert ens.stats() == o.stats()
    kw = dict(nparticles=40 * D, alpha=0.9, epstol=6.0 if D == 17 else 12.0)
    g = k.smc(prior, cost, seed=2, return_array=True, **kw)
    r = orc.smc(prior, cost, seed=2, **kw)
    assert g.info["iterations"] == r["iterations"] > 2 and g.eps == r["eps"]
    assert np.array_equal(g.info["theta_all"], r["theta_all"]) and np.array_equal(g.C, r["C"])
    ga = k.ABCDE(prior, cost, 4.0, seed=9, return_array=True, nparticles=300, generations=15)
    ra = orc.abcde(prior, cost, 4.0, seed=9, nparticles=300, generations=15)
    assert np.array_equal(ga.P, ra["P"]) and np.array_equal(ga.C, ra["C"])
    gp = k.pfilter(prior, cost, 300, seed=4, return_array=True, max_iters=6)
    rp = orc.pfilter(prior, cost, 300, seed=4, max_iters=6)
    assert np.array_equal(gp.P, rp["P"]) and np.array_equal(gp.C, rp["C"])


def test_hiprtc_user_cost_beyond_16_parameters_bit_exact(k, orc, gpu_ctx):
    """a user cost compiled in process (no hipcc, no plugin .so) with length(prior) = 17 and 40:
    AIS, smc, ABCDE and pfilter on the run-time-dimension kernels of its hipRTC unit"""
    src = """
KABC_HD double kabc_user_cost(const double* x, int D, const double* params, const double* data,
                              int64_t ndata, kabc_cost_rng_t* rng) {
    double z0, z1, s = 0.0;
    kabc_cost_rng_normal2(rng, &z0, &z1);
    for (int i = 0; i < D; ++i) s += (x[i] - params[0]) * (x[i] - params[0]);
    return kabc_sqrt(s) + 0.01 * kabc_fabs(z0);
}
"""
    cost = k.costs.UserCost(src, dims=[17, 40], params=[0.25], name="noisy_norm_big")
    orc.register_user_cost(cost)
    for D in (17, 40):
        prior = k.Factored(*[k.Normal(0, 1.5), k.Uniform(-2, 2), k.Beta(2, 3)] * (D // 3) + [k.Normal(0, 1)] * (D % 3))
        model = k.ApproxKernelizedPosterior(prior, cost, 1.0)
        N, seed = 3 * D + 30, 3
        ens = k.AisEnsemble(model, N, seed=seed).init()
        assert np.array_equal(ens.advance(2, 4, collect=True),
                              orc.OracleAIS(model, N, seed=seed).init().generations_sync(2, 4))
        kw = dict(nparticles=50 * D, alpha=0.9, epstol=4.0 if D == 17 else 7.0)
        g = k.smc(prior, cost, seed=2, return_array=True, **kw)
        r = orc.smc(prior, cost, seed=2, **kw)
        assert g.info["iterations"] == r["iterations"] and g.eps == r["eps"]
        assert np.array_equal(g.info["theta_all"], r["theta_all"])
        ga = k.ABCDE(prior, cost, 3.0, seed=9, return_array=True, nparticles=200, generations=10)
        ra = orc.abcde(prior, cost, 3.0, seed=9, nparticles=200, generations=10)
        assert np.array_equal(ga.P, ra["P"]) and np.array_equal(ga.C, ra["C"])
        gp = k.pfilter(prior, cost, 200, seed=4, return_array=True, max_iters=5)
        rp = orc.pfilter(prior, cost, 200, seed=4, max_iters=5)
        assert np.array_equal(gp.P, rp["P"]) and np.array_equal(gp.C, rp["C"])
