"""length(prior) > KABC_MAX_DIM = 16: the run-time-dimension AIS kernels
(csrc/ais_dyn_kernels.hpp).  The reference has no bound on the number of parameters
(src/priors.jl:10-13); beyond 16 the device keeps the walker rows in memory instead of
registers.  Same draws, same operation order: bit-identical to the oracle, like the fast
path -- state, per-transition records (move, accept, partners), trace and counters."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _models(k):
    rng = np.random.default_rng(4)
    return {
        "rosen_d17_box": (k.ApproxKernelizedPosterior(k.Factored(*[k.Uniform(-5, 5)] * 17),
                                                      k.costs.Rosenbrock(), 2.0), 300),
        "gauss_d40_normal": (k.ApproxKernelizedPosterior(k.MvNormal(40, 3.0),
                                                         k.costs.GaussDist(rng.normal(size=40)), 0.5), 257),
        "shell_d24_threshold": (k.ApproxPosterior(k.Factored(*[k.Normal(0, 1)] * 24),
                                                  k.costs.NormShell(4.0), 0.5), 128),
        "hier_d34_mixed": (k.ApproxKernelizedPosterior(
            k.Factored(k.Normal(0, 5), k.Uniform(0, 5), *[k.Normal(0, 1)] * 30, k.Gamma(2.0, 1.0),
                       k.DiscreteUniform(-3, 3)),
            k.costs.HierGaussSim(rng.normal(size=32)), 1.0), 100),
        "gauss_d128": (k.ApproxPosterior(k.Product([k.Uniform(-2, 2)] * 128),
                                         k.costs.GaussDist(np.zeros(128)), 12.0), 1000),
    }


@pytest.mark.parametrize("name", ["rosen_d17_box", "gauss_d40_normal", "shell_d24_threshold",
                                  "hier_d34_mixed", "gauss_d128"])
def test_dynamic_dimension_bit_exact(k, orc, gpu_ctx, name):
    model, N = _models(k)[name]
    nt, gens, seed = 6, 3, 21
    e = k.AisEnsemble(model, N, seed=seed).init()
    o = orc.OracleAIS(model, N, seed=seed).init()
    for got, ref in zip(e.state()[:3], o.state()[:3]):
        assert np.array_equal(got, ref)                      # step(init)
    e.set_debug(nt)
    tr = e.advance(1, nt, collect=True)
    rec = e.get_debug(nt)
    tro, reco = o.generations_sync(1, nt, trace=True)
    assert np.array_equal(tr, tro)
    n0 = (N + 1) // 2
    ro = reco[0].copy()
    ro[:n0, :, 2:5] = np.where(ro[:n0, :, 2:5] >= 0, ro[:n0, :, 2:5] - n0, -1)   # partner ids -> rows
    assert np.array_equal(rec, ro)
    e.set_debug(0)
    tr = e.advance(gens, nt, collect=True)
    assert np.array_equal(tr, o.generations_sync(gens, nt))
    for got, ref in zip(e.state(), o.state()):
        assert np.array_equal(got, ref)
    assert e.stats() == o.stats()
    e.close()


def test_dynamic_dimension_limits_and_sharding(k, orc, gpu_ctx):
    with pytest.raises(ValueError):
        k.Factored(*[k.Normal(0, 1)] * 257)
    big = k.Factored(*[k.Normal(0, 1)] * 20)
    r = k.pfilter(big, k.costs.GaussDist(np.zeros(20)), 500, max_iters=3, seed=2, return_array=True)
    assert r.P.shape[1] == 20                # pfilter / ABCDE go beyond 16 too (tests/test_pfilter.py)
    # walker-sharded (3 emulated ranks, P2P exchange) at D = 20
    model = k.ApproxKernelizedPosterior(big, k.costs.GaussDist(np.ones(20)), 0.5)
    grp = k.EnsembleGroup(model, 301, seed=9, devices=[0, 0, 0], backend="p2p").init()
    grp.advance(3, 4)
    o = orc.OracleAIS(model, 301, seed=9).init()
    o.generations_sync(3, 4, collect=False)
    x, lp, ll = grp.state()
    xo, lpo, llo, _ = o.state()
    assert np.array_equal(x, xo) and np.array_equal(lp, lpo) and np.array_equal(ll, llo)
    grp.close()


def test_many_walkers_of_many_parameters(k, orc, gpu_ctx, monkeypatch):
    """the team kernels' rows live in dynamic LDS: a launch with enough walkers for teams of 4 lanes and 200
    parameters per walker would ask for 143 KB per wavefront -- the host widens the team until the rows fit
    (csrc/ais_dyn_kernels.hpp ais_dyn_team, csrc/smc_dyn_kernels.hpp smc_dyn_team)"""
    monkeypatch.delenv("KABC_DYN_TEAM", raising=False)
    monkeypatch.delenv("KABC_SMC_DYN_TEAM", raising=False)
    D, N = 200, 36000
    pri = k.Factored(*[k.Normal(0.1, 1.5)] * D)
    cost = k.costs.GaussDist(np.linspace(-0.5, 0.5, D))
    model = k.ApproxKernelizedPosterior(pri, cost, 3.0)
    e = k.AisEnsemble(model, N, seed=2).init()
    o = orc.OracleAIS(model, N, seed=2).init()
    e.advance(1, 2)
    o.generations_sync(1, 2, collect=False)
    x, lp, ll = e.state()[:3]
    xo, lpo, llo = o.state()[:3]
    assert np.array_equal(x, xo) and np.array_equal(lp, lpo) and np.array_equal(ll, llo)
    assert e.stats() == o.stats()
    e.close()
    kw = dict(nparticles=33000, alpha=0.5, epstol=13.0, seed=3)
    got = k.smc(pri, cost, return_array=True, **kw)
    ref = orc.smc(pri, cost, **kw)
    assert got.info["log"] == ref["log"] and len(ref["log"]) >= 2
    assert np.array_equal(got.info["theta_all"], ref["theta_all"]) and np.array_equal(got.C, ref["C"])


@pytest.mark.parametrize("form", ["hiprtc", "hipcc"])
def test_dynamic_dimension_user_cost_and_utilities(k, orc, gpu_ctx, monkeypatch, form):
    """A run-time compiled user cost at D = 20 -- compiled in process (hipRTC: the run-time-dimension
    kernels of its unit) or as a plugin .so built by hipcc (which carries its own instantiation of
    them) -- and the Factored utilities (logpdf / push_p / rand on the device) beyond 16 components."""
    monkeypatch.setenv("KABC_USER_PLUGIN", form)
    src = """
KABC_HD double kabc_user_cost(const double* x, int D, const double* params,
                              const double* data, int64_t ndata, kabc_cost_rng_t* rng) {
    double z0, z1, s = 0.0;
    kabc_cost_rng_normal2(rng, &z0, &z1);
    for (int k = 0; k < D; ++k) s += (x[k] - params[0]) * (x[k] - params[0]);
    return kabc_sqrt(s) + 0.01 * kabc_fabs(z0);
}"""
    cost = k.costs.UserCost(src, dims=[20], params=[0.25], name="dyn_user_" + form, posteriors=["kernelized"])
    orc.register_user_cost(cost)
    pri = k.Factored(*[k.Normal(0, 1)] * 19, k.DiscreteUniform(-2, 2))
    model = k.ApproxKernelizedPosterior(pri, cost, 0.7)
    e = k.AisEnsemble(model, 200, seed=5).init()
    tr = e.advance(3, 5, collect=True)
    o = orc.OracleAIS(model, 200, seed=5).init()
    assert np.array_equal(tr, o.generations_sync(3, 5))
    assert e.stats() == o.stats()
    e.close()
    x = np.random.default_rng(1).normal(size=(64, 20))
    assert np.array_equal(pri.logpdf(x), orc.factored_logpdf(pri, x))
    assert np.array_equal(pri.push_p(x), orc.push_p(pri, x))
    assert np.array_equal(pri.rand(32, seed=3), orc.push_p(pri, orc.factored_rand(pri, 32, seed=3)))


@pytest.mark.parametrize("team", [None, "0", "4", "8", "16"])
@pytest.mark.parametrize("name", ["gauss_d20", "hier_d40_mixed", "shell_d17_retrys"])
def test_smc_dynamic_dimension_bit_exact(k, orc, gpu_ctx, name, team, monkeypatch):
    """smc() beyond 16 parameters: run-time-dimension init / propose+accept kernels on the
    kernel-per-phase path (csrc/smc_dyn_kernels.hpp), bit-exact vs the oracle -- with the team of lanes
    per particle the host picks (64 at these sizes), with teams of 4 / 8 / 16, and thread per particle."""
    if team is None:
        monkeypatch.delenv("KABC_SMC_DYN_TEAM", raising=False)
    else:
        monkeypatch.setenv("KABC_SMC_DYN_TEAM", team)
    rng = np.random.default_rng(8)
    cases = {
        "gauss_d20": (k.Factored(*[k.Normal(0, 2)] * 20), k.costs.GaussDist(rng.normal(size=20)),
                      dict(nparticles=3000, alpha=0.9, epstol=3.0)),
        "hier_d40_mixed": (k.Factored(k.Normal(0, 5), k.Uniform(0, 5), *[k.Normal(0, 1)] * 36,
                                      k.Gamma(2.0, 1.0), k.DiscreteUniform(-3, 3)),
                           k.costs.HierGaussSim(rng.normal(size=38)), dict(nparticles=2000, epstol=1.2)),
        "shell_d17_retrys": (k.MvNormal(17, 1.0), k.costs.NormShell(2.0),
                             dict(nparticles=1500, alpha=0.8, epstol=0.3, mcmc_retrys=3, mcmc_tol=0.2)),
    }
    pri, cost, kw = cases[name]
    got = k.smc(pri, cost, seed=4, return_array=True, **kw)
    ref = orc.smc(pri, cost, seed=4, **kw)
    assert got.info["log"] == ref["log"] and got.eps == ref["eps"]
    assert np.array_equal(got.info["theta_all"], ref["theta_all"])
    assert np.array_equal(got.info["alive"], ref["alive"]) and np.array_equal(got.C, ref["C"])
    assert got.info["cost_evals"] == ref["cost_evals"] and got.info["proposals"] == ref["proposals"]
