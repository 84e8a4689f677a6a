"""ABCDE (src/smc.jl:347-430): exported by the reference but undocumented and untested
there, so parity is oracle-vs-device (bit-exact) plus sanity on known posteriors."""
import numpy as np
import pytest


def _cases(k):
    N2 = k.Factored(k.Normal(0, 5), k.Normal(0, 5))
    return {
        "gauss": (N2, k.costs.GaussDist([1.0, -0.5]), 0.05, dict(nparticles=300, generations=100)),
        "gauss_alpha_early": (N2, k.costs.GaussDist([1.0, -0.5]), 0.3,
                              dict(nparticles=200, generations=60, alpha=0.3, earlystop=True)),
        "banana_noisy": (N2, k.costs.NoisyBanana(0.0), 0.05,
                         dict(nparticles=500, generations=50, proposal_width=0.8)),
        "dirac_d1": (k.Normal(1, 0.2), k.costs.DiracSq(1.5), 0.01, dict(nparticles=64)),
        "discrete": (k.Factored(k.Normal(1, 0.5), k.DiscreteUniform(1, 10)), k.costs.NoisyQuadDU(5.5),
                     0.05, dict(nparticles=128, generations=30)),
    }


def test_abcde_oracle_converges(orc, k):
    pri, cost, eps, kw = _cases(k)["gauss"]
    r = orc.abcde(pri, cost, eps, seed=1, **kw)
    assert r["generations_run"] == 100 and r["reached_eps"]
    assert np.all(r["C"] <= eps)
    assert np.all(np.abs(r["P"].mean(0) - [1.0, -0.5]) < 0.02)
    with pytest.raises(orc.OracleError) as e:
        orc.abcde(pri, cost, eps, alpha=1.0)
    assert str(e.value) == "α must be in 0 <= α < 1."
    r2 = orc.abcde(pri, cost, 0.3, seed=1, nparticles=200, generations=60, alpha=0.3, earlystop=True)
    assert r2["reached_eps"] and r2["generations_run"] < 60


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["gauss", "gauss_alpha_early", "banana_noisy", "dirac_d1", "discrete"])
def test_abcde_bit_exact(k, orc, gpu_ctx, name):
    pri, cost, eps, kw = _cases(k)[name]
    got = k.ABCDE(pri, cost, eps, seed=9, return_array=True, **kw)
    ref = orc.abcde(pri, cost, eps, seed=9, **kw)
    assert np.array_equal(got.P, ref["P"])
    assert np.array_equal(got.C, ref["C"])
    assert got.reached_ϵ == ref["reached_eps"]
    assert got.info["generations_run"] == ref["generations_run"]
    assert got.info["nsims"] == ref["nsims"]


@pytest.mark.gpu
@pytest.mark.parametrize("structure", ["blocks", "wavelet"])
@pytest.mark.parametrize("name", ["gauss_5000", "ties_4096", "early_9000", "power2_8192", "ragged_16500"])
def test_abcde_rank_structure_bit_exact(k, orc, gpu_ctx, monkeypatch, name, structure):
    """Larger ensembles: the donor draw s = rand((1:N)[Δs .<= Δs[i]]) (src/smc.jl:392)
    goes through a per-generation structure instead of two O(N) scans per particle: up to 32 768
    particles the costs sorted in blocks of 256 consecutive particles and one wavefront per draw
    (two launches per generation; the default there), beyond -- and under KABC_ABCDE_RANK=wavelet
    -- the costs sorted globally (own LSD radix sort) and a wavelet matrix over the particle
    order.  The oracle keeps the scans.  Heavy ties (integer-valued costs) make the <= prefix
    sets differ from the strict ranks; 16 500 particles: 65 blocks, two per lane, a ragged last one."""
    monkeypatch.setenv("KABC_ABCDE_RANK", structure)
    N2 = k.Factored(k.Normal(0, 5), k.Normal(0, 5))
    du = k.Factored(k.DiscreteUniform(-30, 30), k.DiscreteUniform(-30, 30))
    cases = {
        "gauss_5000": (N2, k.costs.GaussDist([1.0, -0.5]), 0.05, dict(nparticles=5000, generations=12)),
        "ties_4096": (du, k.costs.GaussDist([3.0, -2.0]), 0.5, dict(nparticles=4096, generations=10)),
        "early_9000": (N2, k.costs.GaussDist([1.0, -0.5]), 2.0,
                       dict(nparticles=9000, generations=10, alpha=0.3, earlystop=True)),
        "power2_8192": (N2, k.costs.NoisyBanana(0.0), 0.5, dict(nparticles=8192, generations=6)),
        "ragged_16500": (du, k.costs.GaussDist([3.0, -2.0]), 0.5, dict(nparticles=16500, generations=5, alpha=0.2)),
    }
    pri, cost, eps, kw = cases[name]
    got = k.ABCDE(pri, cost, eps, seed=9, return_array=True, **kw)
    ref = orc.abcde(pri, cost, eps, seed=9, **kw)
    assert np.array_equal(got.P, ref["P"]) and np.array_equal(got.C, ref["C"])
    assert got.info["generations_run"] == ref["generations_run"] and got.info["nsims"] == ref["nsims"]


@pytest.mark.gpu
@pytest.mark.parametrize("D", [17, 40])
def test_abcde_beyond_16_parameters_bit_exact(k, orc, gpu_ctx, D):
    """length(prior) > 16 (the reference has no bound, src/priors.jl:10-13): the run-time-dimension
    instantiation of the same kernels (rows in per-thread arrays), bit-exact; mixed families."""
    comps = [k.Normal(0, 2), k.Uniform(-3, 3), k.Gamma(2.5, 0.7), k.DiscreteUniform(-4, 4)]
    pri = k.Factored(*[comps[j % 4] for j in range(D)])
    cost = k.costs.GaussDist(np.linspace(-0.5, 1.5, D))
    kw = dict(nparticles=400, generations=25, proposal_width=0.9)
    got = k.ABCDE(pri, cost, 3.0, seed=5, return_array=True, **kw)
    ref = orc.abcde(pri, cost, 3.0, seed=5, **kw)
    assert got.P.shape == (400, D)
    assert np.array_equal(got.P, ref["P"]) and np.array_equal(got.C, ref["C"])
    assert got.info["generations_run"] == ref["generations_run"] and got.info["nsims"] == ref["nsims"]


@pytest.mark.gpu
@pytest.mark.parametrize("N,kw", [(2000, dict(generations=30)),
                                  (4095, dict(generations=8, alpha=0.3, earlystop=True)),
                                  (1536, dict(generations=10)),
                                  (257, dict(generations=40, proposal_width=0.7))])
def test_abcde_medium_donor_draws_bit_exact(k, orc, gpu_ctx, monkeypatch, N, kw):
    """256 <= N < 4096: the donor draw rand((1:N)[Δs .<= Δs[i]]) by teams of sixteen lanes
    (abcde_donor_kernel, the default below 1536 particles; ragged last ranges at 4095 and 257, ties from a
    discrete prior), through block-sorted costs (the default from 1536 on; 257 particles: two blocks, one of
    them a single particle), by the generation kernel's own scans (KABC_ABCDE_DONOR=0) and on whatever the
    library picks itself: the oracle's run, every time."""
    pri = k.Factored(k.DiscreteUniform(-10, 10), k.Normal(0, 3))
    cost = k.costs.GaussDist([3.0, -2.0])
    ref = orc.abcde(pri, cost, 0.5, nparticles=N, seed=11, **kw)
    for env in ({}, {"KABC_ABCDE_BLOCKS_FROM": "1000000000"}, {"KABC_ABCDE_BLOCKS_FROM": "256"},
                {"KABC_ABCDE_BLOCKS_FROM": "1000000000", "KABC_ABCDE_DONOR": "0"}):
        for name in ("KABC_ABCDE_BLOCKS_FROM", "KABC_ABCDE_DONOR"):
            monkeypatch.delenv(name, raising=False)
        for name, v in env.items():
            monkeypatch.setenv(name, v)
        r = k.ABCDE(pri, cost, 0.5, nparticles=N, seed=11, return_array=True, **kw)
        assert np.array_equal(r.P, ref["P"]) and np.array_equal(r.C, ref["C"]), env
        assert r.info["generations_run"] == ref["generations_run"] and r.info["nsims"] == ref["nsims"], env
