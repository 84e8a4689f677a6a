"""Every AIS testset of the reference (test/runtests.jl:33-238) on the oracle's SERIAL
schedule -- the faithful restatement of src/KissABC.jl:66-80 (walker i gets `ntransitions`
consecutive transition!() calls against the live ensemble, is emitted, the cursor moves on)
-- judged by the reference's own criterion and nothing weaker:

    x ≈ c   <=>   |mean(x) - c| / std(x) < 2        (MonteCarloMeasurements, Particles vs Real)

`sample(model, AIS(N), Ns; discard_initial = d, ntransitions = nt)` is one init step,
d discarded step() calls and Ns kept ones.  Seeds are fixed, as the reference fixes
Random.seed!(1) (test/runtests.jl:6)."""
import numpy as np
import pytest


def approx(samples, c, lim=2.0):
    samples = np.asarray(samples, dtype=float)
    return abs(samples.mean() - c) / samples.std(ddof=1) < lim


def sample_serial(orc, model, N, Ns, seed, ntransitions=1, discard_initial=0):
    o = orc.OracleAIS(model, N, seed=seed).init()
    if discard_initial:
        o.steps_serial(discard_initial, ntransitions, collect=False)
    return o.steps_serial(Ns, ntransitions)


def test_socks_of_karl_broman(orc, k):
    # test/runtests.jl:33-60
    from kissabc_jl_amd.costs import DeviceCost
    from test_socks import SOCKS_SRC, _prior
    c = DeviceCost(100 + 50, params=[0.0, 11.0], name="socks_serial")
    c.source = SOCKS_SRC
    orc.register_user_cost(c)
    res = sample_serial(orc, k.ApproxPosterior(_prior(k), c, 0.1), 500, 5000, seed=1, ntransitions=100)
    assert np.array_equal(res[:, 0], np.rint(res[:, 0]))      # push_p on emission
    assert approx(res[:, 0], 46.2) and approx(res[:, 1], 0.866)


def test_normal_to_dirac(orc, k):
    # test/runtests.jl:77-86
    abc = k.ApproxKernelizedPosterior(k.Normal(1, 0.2), k.costs.DiracSq(1.5), 0.001)
    res = sample_serial(orc, abc, 12, 500, seed=1, discard_initial=1000)[:, 0]
    assert approx(res * res + 1, 1.5)


def test_normal_to_dirac_chains(orc, k):
    # test/runtests.jl:88-104: 50 chains x 100 samples, discard_initial = 50 * 12, chainsstack
    abc = k.ApproxKernelizedPosterior(k.Normal(1, 0.2), k.costs.DiracSq(1.5), 0.001)
    res = np.concatenate([sample_serial(orc, abc, 12, 100, seed=100 + c, discard_initial=600)[:, 0]
                          for c in range(50)])
    assert res.shape == (5000,) and approx(res * res + 1, 1.5)


def test_normal_plus_discrete_uniform(orc, k):
    # test/runtests.jl:106-113 (sim is re-drawn on the result, as `sim(Tuple(res))` does)
    pri = k.Factored(k.Normal(1, 0.5), k.DiscreteUniform(1, 10))
    res = sample_serial(orc, k.ApproxPosterior(pri, k.costs.NoisyQuadDU(5.5), 0.01), 100, 1000, seed=1,
                        discard_initial=10000)
    assert np.array_equal(res[:, 1], np.rint(res[:, 1]))
    z = np.random.default_rng(1).normal(size=res.shape[0])
    assert approx((res[:, 0] ** 2 + res[:, 1]) * (res[:, 0] + 0.01 * z), 5.5)


def test_drifted_wiener(orc, k):
    # test/runtests.jl:116-130
    t = np.arange(31.0)
    tdata = np.sqrt(0.25 * t * t + 4.0 * t) * (0.95 + 0.1 * np.random.default_rng(1).random())
    prior = k.Factored(k.Uniform(0, 1), k.Uniform(0, 4))
    res = sample_serial(orc, k.ApproxPosterior(prior, k.costs.WienerRms(tdata), 0.1), 50, 100, seed=1,
                        discard_initial=50000)
    assert approx(res[:, 0], 0.5) and approx(res[:, 1], 2.0)


def test_mixture_model_deciles(orc, k):
    # test/runtests.jl:133-170: |decile half-spreads - st_n| < 0.1 on average, both posteriors
    st_n = np.array([0.0, 0.04680825481526908, 0.1057221226763449, 0.2682111969397526,
                     0.8309228020477986])

    def st(r):
        q = np.quantile(r, np.arange(0.1, 0.95, 0.1))
        return ((q - q[::-1]) / 2)[4:]

    prior, cost = k.Uniform(-10, 10), k.costs.Mixture(0.0)
    for model in (k.ApproxPosterior(prior, cost, 0.01),
                  k.ApproxKernelizedPosterior(prior, cost, 0.01 / np.sqrt(2))):
        r = sample_serial(orc, model, 50, 2000, seed=1, ntransitions=100, discard_initial=5000)[:, 0]
        assert np.mean(np.abs(st(r) - st_n)) < 0.1


def test_issue_10(orc, k):
    # test/runtests.jl:177-181
    res = sample_serial(orc, k.ApproxPosterior(k.Normal(0, 1), k.costs.AbsDiff(1.5), 0.01), 20, 100,
                        seed=1, discard_initial=2000)[:, 0]
    assert approx(res, 1.5)


def test_four_dim_shell_chains(orc, k):
    # test/runtests.jl:184-198: mean(plan.cost(res)) < 0.01 (MultivariateNormal(4, 1.0) = N(0,1)^4)
    plan = k.ApproxPosterior(k.Factored(*[k.Normal(0, 1)] * 4), k.costs.NormShell(1.5), 0.01)
    res = np.concatenate([sample_serial(orc, plan, 20, 100, seed=7 + c, ntransitions=40,
                                        discard_initial=10000) for c in range(4)])
    assert np.mean(np.abs(np.sqrt((res ** 2).sum(1)) - 1.5)) < 0.01


BANANA = """
KABC_HD double kabc_user_cost(const double* x, int D, const double* params,
                              const double* data, int64_t ndata, kabc_cost_rng_t* rng) {
    const double a = x[0] - x[1] * x[1], b = x[1] - 1.0;
    return -100.0 * a * a - b * b;
}
"""
DISC = """
KABC_HD double kabc_user_cost(const double* x, int D, const double* params,
                              const double* data, int64_t ndata, kabc_cost_rng_t* rng) {
    return (x[0] * x[0] + x[1] * x[1] <= 1.0) ? 0.0 : params[0];
}
"""


def _user(orc, src, ident, params=()):
    from kissabc_jl_amd.costs import DeviceCost
    c = DeviceCost(100 + ident, params=list(params), name=f"user{ident}")
    c.source = src
    orc.register_user_cost(c)
    return c


def test_common_log_density_banana(orc, k):
    # test/runtests.jl:200-218: quantile(lπ(res), 0.97) > -0.69
    lpi = _user(orc, BANANA, 51)
    D = k.CommonLogDensity(2, k.Factored(k.Normal(0, 1), k.Normal(0, 1)), lpi)
    assert len(D) == 2
    res = sample_serial(orc, D, 50, 1000, seed=1, ntransitions=100, discard_initial=2000)
    a, b = res[:, 0] - res[:, 1] ** 2, res[:, 1] - 1.0
    assert np.quantile(-100 * a * a - b * b, 0.97) > -0.69


def test_infinite_costs(orc, k):
    # test/runtests.jl:221-238: all samples inside the unit half-disc; an all -Inf density errors
    init = k.Factored(k.Uniform(-1, 1), k.Uniform(0, 1))
    D = k.CommonLogDensity(2, init, _user(orc, DISC, 52, [-np.inf]))
    res = sample_serial(orc, D, 50, 1000, seed=1, ntransitions=100, discard_initial=5000)
    assert np.all((res ** 2).sum(1) <= 1.0)
    D2 = k.CommonLogDensity(2, init, _user(orc, "\n".join(DISC.splitlines()).replace(
        "(x[0] * x[0] + x[1] * x[1] <= 1.0) ? 0.0 : params[0]", "params[0]"), 53, [-np.inf]))
    with pytest.raises(orc.OracleError, match="Prior leads to ∞ costs too often"):
        sample_serial(orc, D2, 50, 10, seed=1)
