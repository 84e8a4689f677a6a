"""-m gpu: the reference's own statistical testsets (test/runtests.jl), run through the
product API on the device.  `x ≈ c` is MonteCarloMeasurements' comparison,
|mean - c| < 2 std (Particles.isapprox) -- the reference's criterion, nothing weaker.  (The
same testsets on the reference's SERIAL schedule: tests/test_oracle_reference_testsets_serial.py.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_normal_to_dirac(k, gpu_ctx):
    # test/runtests.jl:77-86
    abc = k.ApproxKernelizedPosterior(k.Normal(1, 0.2), k.costs.DiracSq(1.5), 0.001)
    res = k.sample(abc, k.AIS(12), 500, discard_initial=1000, seed=1)
    sim = k.Particles(np.asarray(res) ** 2 + 1)
    assert sim.isapprox(1.5)
    P = k.smc(k.Normal(1, 0.2), k.costs.DiracSq(1.5), epstol=0.1, seed=1).P
    assert P.isapprox(0.707)


def test_normal_plus_discrete_uniform(k, gpu_ctx):
    # test/runtests.jl:106-114
    pri = k.Factored(k.Normal(1, 0.5), k.DiscreteUniform(1, 10))
    model = k.ApproxPosterior(pri, k.costs.NoisyQuadDU(5.5), 0.01)
    res = k.sample(model, k.AIS(100), 1000, discard_initial=10000, seed=1, return_array=True)
    assert np.array_equal(res[:, 1], np.rint(res[:, 1]))
    sim = k.Particles((res[:, 0] ** 2 + res[:, 1]) * res[:, 0])
    assert sim.isapprox(5.5)
    assert k.smc(pri, k.costs.NoisyQuadDU(5.5), seed=1).P[1].isapprox(5)


def test_drifted_wiener(k, gpu_ctx):
    # test/runtests.jl:116-131
    rng = np.random.default_rng(1)
    t = np.arange(31.0)
    tdata = np.sqrt(0.25 * t * t + 4.0 * t) * (0.95 + 0.1 * rng.random())
    prior = k.Factored(k.Uniform(0, 1), k.Uniform(0, 4))
    cost = k.costs.WienerRms(tdata)
    sim = k.sample(k.ApproxPosterior(prior, cost, 0.1), k.AIS(50), 100, discard_initial=50000, seed=1)
    assert sim[0].isapprox(0.5) and sim[1].isapprox(2.0)
    P = k.smc(prior, cost, min_r_ess=0.55, seed=1).P
    assert P[0].isapprox(0.5) and P[1].isapprox(2.0)


def test_mixture_model_deciles(k, gpu_ctx):
    # test/runtests.jl:133-175
    st_n = np.array([0.0, 0.04680825481526908, 0.1057221226763449, 0.2682111969397526,
                     0.8309228020477986])

    def st(r):
        q = np.quantile(np.asarray(r), np.arange(0.1, 0.95, 0.1))
        return ((q - q[::-1]) / 2)[4:]

    prior, cost = k.Uniform(-10, 10), k.costs.Mixture(0.0)
    kw = dict(ntransitions=100, discard_initial=5000, seed=1)
    res = k.sample(k.ApproxPosterior(prior, cost, 0.01), k.AIS(50), 2000, **kw)
    resk = k.sample(k.ApproxKernelizedPosterior(prior, cost, 0.01 / np.sqrt(2)), k.AIS(50), 2000, **kw)
    ressmc = k.smc(prior, cost, nparticles=2000, alpha=0.9, epstol=0.01, mcmc_retrys=500,
                   mcmc_tol=0.9, seed=1).P
    for r in (res, resk, ressmc):
        assert np.mean(np.abs(st(r) - st_n)) < 0.1


def test_issue_10(k, gpu_ctx):
    # test/runtests.jl:177-182
    plan = k.ApproxPosterior(k.Normal(0, 1), k.costs.AbsDiff(1.5), 0.01)
    res = k.sample(plan, k.AIS(20), 100, discard_initial=2000, seed=1)
    assert res.isapprox(1.5)


def test_four_dim_shell_with_chains(k, gpu_ctx):
    # test/runtests.jl:184-198: vector-valued walkers from MultivariateNormal(4, 1.0)
    plan = k.ApproxPosterior(k.MultivariateNormal(4, 1.0), k.costs.NormShell(1.5), 0.01)
    res = k.sample(plan, k.AIS(20), k.MCMCThreads(), 100, 4, discard_initial=10000, ntransitions=40,
                   seed=1, return_array=True)
    assert res.shape == (400, 4)
    assert np.mean(np.abs(np.sqrt((res ** 2).sum(1)) - 1.5)) < 0.01


def test_smc_testset(k, gpu_ctx):
    # test/runtests.jl:240-254
    pp = k.Factored(k.Normal(0, 5), k.Normal(0, 5))
    R = k.smc(pp, k.costs.NoisyBanana(0.0), alpha=0.9, nparticles=500, epstol=0.01, parallel=True,
              seed=1).P
    assert R[0].isapprox(1) and R[1].isapprox(1)
    R = k.smc(pp, k.costs.NoisyBanana(0.5), alpha=0.9, nparticles=1000, epstol=0.01, parallel=True,
              seed=1).P
    assert R[0].isapprox(1) and R[1].isapprox(1)
