"""-m gpu: edge cases of the path -- smallest legal ensembles, a 2^20-walker ensemble,
maximum dimension, argument errors."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N", [6, 7, 8, 65, 129])
def test_smallest_and_ragged_ensembles(k, orc, gpu_ctx, N):
    """N = D+5 is the reference's minimum (src/KissABC.jl:43); with D = 1 the halves hold 3
    walkers, exactly what the walk move's three distinct partners need."""
    model = k.ApproxKernelizedPosterior(k.Normal(0, 1), k.costs.AbsDiff(0.3), 0.5)
    got = k.AisEnsemble(model, N, seed=N).init().advance(6, 7, collect=True)
    ref = orc.OracleAIS(model, N, seed=N).init().generations_sync(6, 7)
    assert np.array_equal(got, ref)


def test_million_walkers_bit_exact(k, orc, gpu_ctx):
    N2 = k.Factored(k.Normal(0, 5), k.Normal(0, 5))
    model = k.ApproxKernelizedPosterior(N2, k.costs.GaussDist([1.0, -0.5]), 0.1)
    N = 1 << 20
    ens = k.AisEnsemble(model, N, seed=5).init()
    ens.advance(1, 2)
    o = orc.OracleAIS(model, N, seed=5).init()
    o.generations_sync(1, 2, collect=False)
    x, lp, ll, t = ens.state()
    xo, lpo, llo, to = o.state()
    assert t == to == 2
    assert np.array_equal(x, xo) and np.array_equal(lp, lpo) and np.array_equal(ll, llo)
    assert ens.stats() == o.stats()


def test_max_dimension_and_beyond(k, orc, gpu_ctx):
    D = k.KABC_MAX_DIM
    model = k.ApproxKernelizedPosterior(k.Factored(*[k.Normal(0, 1)] * D),
                                        k.costs.GaussDist(np.linspace(-1, 1, D)), 0.5)
    got = k.AisEnsemble(model, D + 5, seed=2).init().advance(3, 4, collect=True)
    assert np.array_equal(got, orc.OracleAIS(model, D + 5, seed=2).init().generations_sync(3, 4))
    # beyond KABC_MAX_DIM the run-time-dimension kernels take over (tests/test_gpu_dyn_dim.py) ...
    m17 = k.ApproxKernelizedPosterior(k.Factored(*[k.Normal(0, 1)] * (D + 1)),
                                      k.costs.GaussDist(np.linspace(-1, 1, D + 1)), 0.5)
    got = k.AisEnsemble(m17, D + 6, seed=2).init().advance(3, 4, collect=True)
    assert np.array_equal(got, orc.OracleAIS(m17, D + 6, seed=2).init().generations_sync(3, 4))
    # ... up to KABC_MAX_DIM_DYN
    from kissabc_jl_amd import _cdefs
    with pytest.raises(ValueError):
        k.Factored(*[k.Normal(0, 1)] * (_cdefs.KABC_MAX_DIM_DYN + 1))


def test_smc_minimal_particles_and_all_alive_ties(k, orc, gpu_ctx):
    pri = k.Factored(k.Normal(0, 5), k.Normal(0, 5))
    # nparticles = ceil(3D / min(alpha, min_r_ess)) is the reference's minimum (src/smc.jl:113-118)
    for kw in (dict(nparticles=7), dict(nparticles=10, alpha=0.8), dict(nparticles=24, alpha=0.5)):
        got = k.smc(pri, k.costs.GaussDist([1.0, -0.5]), seed=3, return_array=True, **kw)
        ref = orc.smc(pri, k.costs.GaussDist([1.0, -0.5]), seed=3, **kw)
        assert np.array_equal(got.info["theta_all"], ref["theta_all"]) and got.eps == ref["eps"]
        assert got.info["log"] == ref["log"]
    # heavily tied costs (integers): the select's "all keys of the range equal" path
    du = k.Factored(k.Normal(1, 0.5), k.DiscreteUniform(1, 10))
    tie = k.costs.UserCost('''
KABC_HD double kabc_user_cost(const double* x, int D, const double* params,
                              const double* data, int64_t ndata, kabc_cost_rng_t* rng) {
    return kabc_fabs(x[1] - 5.0) + kabc_floor(kabc_fabs(x[0]));
}''', dims=[2], name="ties")
    orc.register_user_cost(tie)
    got = k.smc(du, tie, nparticles=5000, alpha=0.9, epstol=0.5, seed=4, return_array=True)
    ref = orc.smc(du, tie, nparticles=5000, alpha=0.9, epstol=0.5, seed=4)
    assert got.info["log"] == ref["log"] and got.eps == ref["eps"]
    assert np.array_equal(got.info["theta_all"], ref["theta_all"])


def test_argument_errors(k, gpu_ctx):
    model = k.ApproxKernelizedPosterior(k.Normal(0, 1), k.costs.AbsDiff(0.3), 0.5)
    ens = k.AisEnsemble(model, 16, seed=1)
    with pytest.raises(k.KabcError):      # advance before init
        ens.advance(1, 1)
    ens.init()
    with pytest.raises(k.KabcError):      # ntransitions >= 1
        ens.advance(1, 0)
    with pytest.raises(k.KabcError):      # cost / dimension mismatch
        k.AisEnsemble(k.ApproxKernelizedPosterior(k.Normal(0, 1), k.costs.Rosenbrock(), 1.0), 16)
    with pytest.raises(k.KabcError):      # invalid prior parameters
        k.AisEnsemble(k.ApproxKernelizedPosterior(k.Normal(0, -1), k.costs.AbsDiff(0.3), 0.5), 16)
    # a half-ensemble must stay below 4 GiB (32-bit row offsets, include/kabc.h): refused before
    # anything is allocated
    big = k.ApproxKernelizedPosterior(k.Factored(*[k.Uniform(-5, 5)] * 8), k.costs.Rosenbrock(), 1.0)
    with pytest.raises(k.KabcError, match="4 GiB"):
        k.AisEnsemble(big, 2 ** 28, seed=1)
