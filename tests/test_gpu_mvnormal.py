"""-m gpu: a full-covariance MvNormal(μ, Σ) prior (include/kabc_mvnormal.h; the reference takes
any Distribution as a prior: src/types.jl:30, :34-35, :52; src/smc.jl:92) on every device path
against the oracle, bit for bit: the prior kernels, AIS (all three posterior kinds, batched
chains, emulated ranks), smc on both drivers, ABCDE, pfilter."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _mv(k, D, seed=0, spread=1.0):
    rng = np.random.default_rng(100 + seed + D)
    A = rng.normal(size=(D, D)) * spread
    return k.MvNormal(rng.normal(size=D), A @ A.T + 0.4 * np.eye(D))


@pytest.mark.parametrize("D", [1, 2, 5, 16])
def test_prior_kernels_bit_exact(k, orc, gpu_ctx, D):
    from kissabc_jl_amd import _cdefs as cd
    d = _mv(k, D)
    x = np.random.default_rng(2).normal(size=(2000, D)) * 3
    assert np.array_equal(d.logpdf(x), orc.factored_logpdf(d, x))
    assert np.array_equal(d.push_p(x), x)
    got = d.rand(3000, seed=42)
    assert np.array_equal(got, orc.factored_rand(d, 3000, seed=42, domain=cd.DOM_AIS_INIT))


@pytest.mark.parametrize("D,kind", [(2, "kernelized"), (5, "threshold"), (8, "kernelized"), (16, "kernelized"),
                                    (3, "common")])
def test_ais_bit_exact(k, orc, gpu_ctx, D, kind):
    prior = _mv(k, D)
    cost = k.costs.GaussDist(np.linspace(-1, 1, D))
    model = {"kernelized": lambda: k.ApproxKernelizedPosterior(prior, cost, 1.5),
             "threshold": lambda: k.ApproxPosterior(prior, cost, 6.0),
             "common": lambda: k.CommonLogDensity(D, prior, k.costs.NormShell(1.0))}[kind]()
    N, nt = 777, 6
    ens = k.AisEnsemble(model, N, seed=4).init()
    o = orc.OracleAIS(model, N, seed=4).init()
    assert all(np.array_equal(a, b) for a, b in zip(ens.state()[:3], o.state()[:3]))
    assert np.array_equal(ens.advance(3, nt, collect=True), o.generations_sync(3, nt))
    assert all(np.array_equal(a, b) for a, b in zip(ens.state()[:3], o.state()[:3]))
    assert ens.stats() == o.stats()


def test_ais_chains_and_emulated_ranks(k, orc, gpu_ctx, monkeypatch):
    prior = _mv(k, 4)
    model = k.ApproxKernelizedPosterior(prior, k.costs.Rosenbrock(), 2.0)
    seeds = [5, 6]
    ens = k.AisEnsemble(model, 120, seeds=seeds).init()
    got = ens.advance(2, 5, collect=True)
    for c, sd in enumerate(seeds):
        assert np.array_equal(got[:, c], orc.OracleAIS(model, 120, seed=sd).init().generations_sync(2, 5))
    monkeypatch.setenv("KABC_EXCHANGE_CHUNKS", "2")
    grp = k.EnsembleGroup(model, 900, seed=17, devices=[0] * 3, backend="p2p").init()
    o = orc.OracleAIS(model, 900, seed=17).init()
    grp.advance(2, 4)
    o.generations_sync(2, 4, collect=False)
    for r in range(3):
        assert np.array_equal(grp.ensemble(r), o.state()[0])
    assert grp.stats() == o.stats()
    grp.close()


@pytest.mark.parametrize("path", ["loop", "kernels"])
@pytest.mark.parametrize("D", [2, 6])
def test_smc_bit_exact(k, orc, gpu_ctx, monkeypatch, path, D):
    monkeypatch.setenv("KABC_SMC_LOOP", "1" if path == "loop" else "0")
    prior = _mv(k, D, spread=2.0)
    cost = k.costs.GaussDist(np.linspace(0.5, -0.5, D))
    kw = dict(nparticles=1500, alpha=0.9, epstol=0.2 if D == 2 else 0.8, mcmc_retrys=1, seed=3)
    got = k.smc(prior, cost, return_array=True, **kw)
    ref = orc.smc(prior, cost, **kw)
    assert got.info["log"] == ref["log"] and got.eps == ref["eps"]
    assert np.array_equal(got.info["theta_all"], ref["theta_all"])
    assert np.array_equal(got.info["alive"], ref["alive"]) and np.array_equal(got.C, ref["C"])


def test_abcde_and_pfilter_bit_exact(k, orc, gpu_ctx):
    prior = _mv(k, 3, spread=2.0)
    cost = k.costs.GaussDist([0.3, -0.2, 0.1])
    kw = dict(nparticles=600, generations=25, seed=3)
    r, ro = k.ABCDE(prior, cost, 0.05, return_array=True, **kw), orc.abcde(prior, cost, 0.05, **kw)
    assert np.array_equal(r.P, ro["P"]) and np.array_equal(r.C, ro["C"])
    kw = dict(q=0.7, eff_tol=0.1, epstol=0.3, max_iters=30, seed=3)
    r, ro = k.pfilter(prior, cost, 800, return_array=True, **kw), orc.pfilter(prior, cost, 800, **kw)
    assert np.array_equal(r.P, ro["P"]) and np.array_equal(r.C, ro["C"])


def test_components_do_not_mix_and_handles_are_checked(k, gpu_ctx):
    """C ABI: an MvNormal prior is all D components, one handle, p[1] = index"""
    from kissabc_jl_amd import _cdefs as cd, _lib
    lib = _lib.load()
    d = _mv(k, 3)
    arr = d.to_c()
    out = np.empty(1)
    x = np.zeros((1, 3))

    def call(a, D=3):
        return lib.kabc_factored_logpdf(gpu_ctx.handle, a, D, 1, x.ctypes.data_as(cd.c_double_p),
                                        out.ctypes.data_as(cd.c_double_p))
    assert call(arr) == 0
    mixed = d.to_c()
    mixed[1] = k.Normal(0, 1).to_c()
    assert call(mixed) != 0 and b"does not mix" in lib.kabc_last_error()
    bad = d.to_c()
    bad[2].p[1] = 1.0
    assert call(bad) != 0 and b"must carry" in lib.kabc_last_error()
    stale = d.to_c()
    for c in stale:
        c.p[0] = 1e6
    assert call(stale) != 0 and b"not a registered" in lib.kabc_last_error()
    assert call(arr, D=2) != 0


@pytest.mark.parametrize("case", range(16))
def test_random_cases_bit_exact(k, orc, gpu_ctx, case):
    rng = np.random.default_rng(40000 + case)
    D = int(rng.integers(1, 17))
    A = rng.normal(size=(D, D)) * rng.uniform(0.3, 2.0)
    prior = k.MvNormal(rng.normal(size=D), A @ A.T + rng.uniform(0.05, 1.0) * np.eye(D))
    cost = [k.costs.GaussDist(rng.uniform(-1, 1, D)), k.costs.NormShell(float(rng.uniform(0.5, 2)))][case % 2]
    model = k.ApproxKernelizedPosterior(prior, cost, float(rng.uniform(0.5, 3))) if case % 3 else \
        k.ApproxPosterior(prior, cost, float(rng.uniform(3, 12)))
    N, nt, gens, seed = int(rng.integers(D + 5, 2000)), int(rng.integers(1, 7)), int(rng.integers(1, 4)), int(rng.integers(0, 2 ** 31))
    ens = k.AisEnsemble(model, N, seed=seed).init()
    o = orc.OracleAIS(model, N, seed=seed).init()
    assert np.array_equal(ens.advance(gens, nt, collect=True), o.generations_sync(gens, nt))
    assert all(np.array_equal(a, b) for a, b in zip(ens.state()[:3], o.state()[:3]))
    assert ens.stats() == o.stats()


# (the hipcc-built plugin form passes too -- KABC_TEST_HIPCC_PLUGIN=1 adds it: ~90 s of compilation)
@pytest.mark.parametrize("form", ["hiprtc"] + (["hipcc"] if __import__("os").environ.get("KABC_TEST_HIPCC_PLUGIN") else []))
def test_user_cost_with_an_mvnormal_prior(k, orc, gpu_ctx, monkeypatch, form):
    """a user DeviceCost (run-time compiled kernels: hipRTC, or the hipcc-built plugin) under an
    MvNormal prior: the plugin's kernels are instantiated from the same headers"""
    monkeypatch.setenv("KABC_USER_PLUGIN", form)
    src = """
KABC_HD double kabc_user_cost(const double* x, int D, const double* params,
                              const double* data, int64_t ndata, kabc_cost_rng_t* rng) {
    double s = 0.0;
    for (int k = 0; k < D; ++k) s += (x[k] - params[k]) * (x[k] - params[k]);
    return kabc_sqrt(s) + 0.25 * kabc_fabs(x[0] * x[1]);
}
"""
    user = k.costs.UserCost(src, dims=[3], params=[0.3, -0.2, 0.1], name=f"mvn_user_{form}", posteriors=["kernelized"])
    orc.register_user_cost(user)
    prior = _mv(k, 3)
    model = k.ApproxKernelizedPosterior(prior, user, 1.0)
    ens = k.AisEnsemble(model, 400, seed=6).init()
    o = orc.OracleAIS(model, 400, seed=6).init()
    assert np.array_equal(ens.advance(3, 5, collect=True), o.generations_sync(3, 5))
    kw = dict(nparticles=700, alpha=0.9, epstol=0.5, seed=2)
    r, ro = k.smc(prior, user, return_array=True, **kw), orc.smc(prior, user, **kw)
    assert np.array_equal(r.info["theta_all"], ro["theta_all"]) and r.eps == ro["eps"]
