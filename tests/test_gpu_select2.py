"""-m gpu: the ε-selection of smc / pfilter as barrier-free kernels over all CUs
(csrc/smc_select2_kernels.hpp; the default from 2^17 particles) against the oracle's restatement of
src/smc.jl:134-153 / :298-301.  Bar: BIT-EXACT ε, ESS, alive mask, resample index (through θ
equality) per iteration.  KABC_SMC_SELECT2_FROM forces the kernels at sizes the oracle finishes in
seconds; KABC_SMC_SELECT2_BLOCKS varies the grid (one workgroup, ragged slices, more workgroups
than tiles)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _check(got, ref):
    assert got.info["iterations"] == ref["iterations"]
    assert got.info["log"] == ref["log"]
    assert got.eps == ref["eps"]
    assert np.array_equal(got.info["alive"], ref["alive"])
    assert np.array_equal(got.info["theta_all"], ref["theta_all"])
    assert np.array_equal(got.C, ref["C"])
    assert got.info["cost_evals"] == ref["cost_evals"] and got.info["proposals"] == ref["proposals"]


def _cases(k):
    N2 = k.Factored(k.Normal(0, 5), k.Normal(0, 5))
    rng = np.random.default_rng(1)
    ybar = 1.0 + 0.5 * rng.normal(size=14) + rng.normal(size=14) / np.sqrt(8)
    H16 = k.Factored(k.Normal(0, 5), k.Uniform(0, 5), *[k.Normal(0, 1)] * 14)
    return {
        "banana_inf": (N2, k.costs.NoisyBanana(0.5), dict(nparticles=5000, alpha=0.9, epstol=0.01)),
        "gauss_minress": (N2, k.costs.GaussDist([1.0, -0.5]), dict(nparticles=9000, min_r_ess=0.55, epstol=0.02)),
        "mixture_retrys": (k.Uniform(-10, 10), k.costs.Mixture(0.0),
                           dict(nparticles=4100, alpha=0.9, epstol=0.01, mcmc_retrys=500, mcmc_tol=0.9)),
        "hier16": (H16, k.costs.HierGaussSim(ybar), dict(nparticles=12345, alpha=0.95, epstol=0.08)),
        # heavy ties: a discrete cost (every alive key equal at the end: state 2, rank j above the bin)
        "du_ties": (k.Factored(k.Normal(1, 0.5), k.DiscreteUniform(1, 10)), k.costs.NoisyQuadDU(5.5),
                    dict(nparticles=3000)),
        "small_alpha": (N2, k.costs.GaussDist([0.3, 0.2]), dict(nparticles=8000, alpha=0.3, epstol=0.05)),
    }


@pytest.mark.parametrize("name", ["banana_inf", "gauss_minress", "mixture_retrys", "hier16", "du_ties", "small_alpha"])
@pytest.mark.parametrize("blocks", [None, "1", "3"])
def test_smc_on_barrier_free_select_bit_exact(k, orc, gpu_ctx, monkeypatch, name, blocks):
    monkeypatch.setenv("KABC_SMC_LOOP", "0")
    monkeypatch.setenv("KABC_SMC_SELECT2_FROM", "2048")
    if blocks:
        monkeypatch.setenv("KABC_SMC_SELECT2_BLOCKS", blocks)
    prior, cost, kw = _cases(k)[name]
    ref = orc.smc(prior, cost, seed=5, **kw)
    c0 = k.smc(prior, cost, seed=5, return_array=True, **kw)
    _check(c0, ref)
    assert ref["iterations"] > 2


def test_select2_is_what_runs_and_matches_the_one_kernel_select(k, orc, gpu_ctx, monkeypatch):
    """150 000 particles (147 tiles, ragged last slice; the default path at this size): the
    barrier-free kernels, the one-kernel select (KABC_SMC_SELECT2_FROM=0) and the oracle agree
    bit for bit; the give-up path (a candidate bin beyond the list) repeats the run on the
    one-kernel select and returns the same arrays."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from smc_c4_probe import c4_problem
    prior, cost = c4_problem()
    kw = dict(nparticles=150000, alpha=0.95, epstol=0.3, seed=3)
    r2 = k.smc(prior, cost, return_array=True, **kw)
    monkeypatch.setenv("KABC_SMC_SELECT2_FROM", "0")
    r1 = k.smc(prior, cost, return_array=True, **kw)
    monkeypatch.delenv("KABC_SMC_SELECT2_FROM")
    monkeypatch.setenv("KABC_SMC_SELECT2_GIVE_UP", "1")
    r3 = k.smc(prior, cost, return_array=True, **kw)
    ro = orc.smc(prior, cost, **kw)
    assert ro["iterations"] > 15
    for rr in (r2, r1, r3):
        _check(rr, ro)


def test_pfilter_on_barrier_free_select_bit_exact(k, orc, gpu_ctx, monkeypatch):
    monkeypatch.setenv("KABC_SMC_SELECT2_FROM", "2048")
    n2 = k.Factored(k.Normal(0, 5), k.Normal(0, 5))
    gd = k.costs.GaussDist([1.0, -0.5])
    for N, blocks in ((5000, None), (4097, "2")):
        if blocks:
            monkeypatch.setenv("KABC_SMC_SELECT2_BLOCKS", blocks)
        for scheme in ("0", "1"):
            monkeypatch.setenv("KABC_PF_PASSES", scheme)
            got = k.pfilter(n2, gd, N, seed=4, return_array=True, q=0.7, eff_tol=0.1, epstol=0.02)
            ref = orc.pfilter(n2, gd, N, seed=4, q=0.7, eff_tol=0.1, epstol=0.02)
            assert np.array_equal(got.P, ref["P"]) and np.array_equal(got.C, ref["C"])
            assert got.info["iterations"] == ref["iterations"] and got.info["eps"] == ref["eps"]


def test_smc_runtime_dimension_on_barrier_free_select(k, orc, gpu_ctx, monkeypatch):
    """length(prior) = 17 (run-time-dimension kernels): their propose / accept kernel takes over
    writing the all-alive mask after a resample too"""
    monkeypatch.setenv("KABC_SMC_SELECT2_FROM", "2048")
    prior = k.Factored(*[k.Normal(0, 2)] * 17)
    cost = k.costs.GaussDist(np.linspace(-1, 1, 17))
    kw = dict(nparticles=3000, alpha=0.9, epstol=2.0)
    _check(k.smc(prior, cost, seed=2, return_array=True, **kw), orc.smc(prior, cost, seed=2, **kw))


def test_nan_and_empty_costs_are_reported(k, gpu_ctx, monkeypatch):
    monkeypatch.setenv("KABC_SMC_LOOP", "0")
    monkeypatch.setenv("KABC_SMC_SELECT2_FROM", "2048")
    cost = k.costs.UserCost("KABC_HD double kabc_user_cost(const double* x, int D, const double* params, "
                            "const double* data, int64_t ndata, kabc_cost_rng_t* rng) { return x[0] > 0 ? x[0] : KABC_NAN; }",
                            dims=[1], name="nan_half")
    with pytest.raises(k.KabcError, match="NaN"):
        k.smc(k.Normal(0, 1), cost, nparticles=5000, seed=1)
