"""'Tiny Data, Approximate Bayesian Computation and the Socks of Karl Broman'
(test/runtests.jl:33-75): NegativeBinomial x Beta prior, discrete push_p, hard-threshold
posterior, a stochastic integer simulator -- here as a run-time compiled user cost.
Known answers of the reference: n_socks ≈ 46.2, prop_pairs ≈ 0.866."""
import numpy as np
import pytest

# The reference builds the sock drawer as an array and takes the first 11 of a random
# permutation.  Drawing 11 socks one by one without replacement, tracking only how many
# pairs are intact / half-picked and how many singletons remain, has exactly the same
# distribution and needs no array.
SOCKS_SRC = """
KABC_HD double kabc_user_cost(const double* x, int D, const double* params,
                              const double* data, int64_t ndata, kabc_cost_rng_t* rng) {
    const double n_socks = x[0], prop_pairs = x[1];
    const double n_picked = 11.0;
    const double n_pairs = kabc_rint(prop_pairs * kabc_floor(n_socks / 2.0));
    const double n_odd = n_socks - 2.0 * n_pairs;
    double full = n_pairs, half = 0.0, odd = n_odd;   /* intact pairs, half-picked pairs, singletons */
    double pairs_found = 0.0, drawn = 0.0;
    const double todo = (n_socks < n_picked) ? n_socks : n_picked;
    double u0 = 0.0, u1 = 0.0;
    for (int i = 0; i < 11; ++i) {
        if ((double)i >= todo) break;
        if ((i & 1) == 0) kabc_cost_rng_uniform2(rng, &u0, &u1);
        const double u = (i & 1) ? u1 : u0;
        const double total = 2.0 * full + half + odd;
        const double r = u * total;
        if (r < 2.0 * full) { full -= 1.0; half += 1.0; }
        else if (r < 2.0 * full + half) { half -= 1.0; pairs_found += 1.0; }
        else { odd -= 1.0; }
        drawn += 1.0;
    }
    const double lu = drawn - pairs_found;            /* length(unique(picked_socks)) */
    const double sample_pairs = todo - lu, sample_odds = lu - sample_pairs;
    return kabc_fabs(sample_pairs - params[0]) + kabc_fabs(sample_odds - params[1]);
}
"""


def _prior(k):
    prior_mu, prior_sd = 30.0, 15.0
    size = -prior_mu ** 2 / (prior_mu - prior_sd ** 2)
    return k.Factored(k.NegativeBinomial(size, size / (prior_mu + size)), k.Beta(15, 2))


def test_socks_oracle_smc(k, orc):
    from kissabc_jl_amd.costs import DeviceCost
    c = DeviceCost(100 + 61, params=[0.0, 11.0], name="socks")
    c.source = SOCKS_SRC
    orc.register_user_cost(c)
    r = orc.smc(_prior(k), c, nparticles=5000, alpha=0.99, r_epstol=0.0, epstol=0.01, seed=1)
    P = r["P"]
    assert np.array_equal(P[:, 0], np.rint(P[:, 0]))
    assert abs(P[:, 0].mean() - 46.2) < 2 * P[:, 0].std() and abs(P[:, 0].mean() - 46.2) < 4.0
    assert abs(P[:, 1].mean() - 0.866) < 2 * P[:, 1].std() and abs(P[:, 1].mean() - 0.866) < 0.03


@pytest.mark.gpu
def test_socks_on_device(k, orc, gpu_ctx):
    socks = k.costs.UserCost(SOCKS_SRC, dims=[2], params=[0.0, 11.0], name="socks")
    pri = _prior(k)
    model = k.ApproxPosterior(pri, socks, 0.1)
    res = k.sample(model, k.AIS(500), 5000, ntransitions=100, seed=1)
    assert res[0].isapprox(46.2) and res[1].isapprox(0.866)      # test/runtests.jl:59-60
    assert abs(res[0].mean() - 46.2) < 4.0 and abs(res[1].mean() - 0.866) < 0.03
    P = k.smc(pri, socks, nparticles=5000, alpha=0.99, r_epstol=0.0, epstol=0.01, seed=1).P
    assert P[0].isapprox(46.2) and P[1].isapprox(0.866)          # test/runtests.jl:73-74
    # and bit-exact against the oracle running the same snippet through gcc
    orc.register_user_cost(socks)
    got = k.AisEnsemble(model, 500, seed=3).init().advance(2, 10, collect=True)
    assert np.array_equal(got, orc.OracleAIS(model, 500, seed=3).init().generations_sync(2, 10))
