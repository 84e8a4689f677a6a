"""-m gpu: kabc_smc_run (HIP) against the oracle's restatement of smc()
(src/smc.jl:92-206).  Bar: BIT-EXACT final positions, costs, alive mask, ε and
per-iteration (ε, ESS, accepted, resampled, flag) -- which pins the device
quantile (radix select), the alive mask and the cyclic resample index
repeat(idxalive, ceil(N/m))[1:N] index-for-index."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cases(k):
    N2 = k.Factored(k.Normal(0, 5), k.Normal(0, 5))
    rng = np.random.default_rng(1)
    zstar = rng.normal(size=14)
    ybar = 1.0 + 0.5 * zstar + rng.normal(size=14) / np.sqrt(8)
    H16 = k.Factored(k.Normal(0, 5), k.Uniform(0, 5), *[k.Normal(0, 1)] * 14)
    return {
        "banana": (N2, k.costs.NoisyBanana(0.0), dict(nparticles=500, alpha=0.9, epstol=0.01)),
        "banana_inf": (N2, k.costs.NoisyBanana(0.5), dict(nparticles=1000, alpha=0.9, epstol=0.01)),
        "dirac": (k.Normal(1, 0.2), k.costs.DiracSq(1.5), dict(epstol=0.1)),
        "defaults_du": (k.Factored(k.Normal(1, 0.5), k.DiscreteUniform(1, 10)),
                        k.costs.NoisyQuadDU(5.5), dict()),
        "mixture_retrys": (k.Uniform(-10, 10), k.costs.Mixture(0.0),
                           dict(nparticles=2000, alpha=0.9, epstol=0.01, mcmc_retrys=500,
                                mcmc_tol=0.9)),
        "C4_hier16_small": (H16, k.costs.HierGaussSim(ybar),
                            dict(nparticles=4096, alpha=0.95, epstol=0.05)),
        "gauss_d2_minress": (N2, k.costs.GaussDist([1.0, -0.5]),
                             dict(nparticles=3000, min_r_ess=0.55, epstol=0.02)),
    }


@pytest.mark.parametrize("name", ["banana", "banana_inf", "dirac", "defaults_du",
                                  "mixture_retrys", "C4_hier16_small", "gauss_d2_minress"])
def test_smc_bit_exact(k, orc, gpu_ctx, name):
    prior, cost, kw = _cases(k)[name]
    got = k.smc(prior, cost, seed=5, return_array=True, **kw)
    ref = orc.smc(prior, cost, seed=5, **kw)
    assert got.info["iterations"] == ref["iterations"]
    assert got.info["log"] == ref["log"]
    assert got.eps == ref["eps"]
    assert np.array_equal(got.info["alive"], ref["alive"])
    assert np.array_equal(got.info["theta_all"], ref["theta_all"])
    assert np.array_equal(got.C, ref["C"])
    assert np.array_equal(got.P, ref["P"])
    assert got.info["cost_evals"] == ref["cost_evals"]
    assert got.info["proposals"] == ref["proposals"]


def test_smc_argument_errors_match_reference(k, gpu_ctx):
    pri = k.Factored(k.Normal(0, 5), k.Normal(0, 5))
    c = k.costs.NoisyBanana()
    for kw, msg in [(dict(min_r_ess=0.0), "min_r_ess must be > 0."),
                    (dict(mcmc_retrys=-1), "mcmc_retrys must be >= 0."),
                    (dict(alpha=0.0, min_r_ess=0.5, r_epstol=0.1), "alpha must be > 0."),
                    (dict(r_epstol=-1.0), "r_epstol must be >= 0"),
                    (dict(mcmc_tol=-0.1), "mcmc_tol must be >= 0"),
                    (dict(max_stretch=1.0), "max_stretch must be > 1"),
                    (dict(nparticles=6), "nparticles must be >= 7.")]:
        with pytest.raises(k.KabcError) as e:
            k.smc(pri, c, **kw)
        assert str(e.value) == msg


@pytest.mark.parametrize("blocks", [None, "1", "32"])
def test_c4_full_size_bit_exact(k, orc, gpu_ctx, monkeypatch, blocks):
    """BASELINE.json configs[3] at full size (32 768 particles, D = 16, hierarchical
    Gaussian simulator, ~190 ε-iterations): θ of every particle, ε and the iteration
    log equal the oracle's bit for bit -- with the select kernel on its default 16
    workgroups, on one, and on 32."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from smc_c4_probe import c4_problem
    if blocks is None:
        monkeypatch.delenv("KABC_SMC_SELECT_BLOCKS", raising=False)
    else:
        monkeypatch.setenv("KABC_SMC_SELECT_BLOCKS", blocks)
    prior, cost = c4_problem()
    kw = dict(nparticles=32768, alpha=0.95, epstol=0.05, seed=1)
    r = k.smc(prior, cost, return_array=True, **kw)
    ro = _oracle_c4(orc, prior, cost, kw)
    assert r.info["iterations"] == ro["iterations"] > 100
    assert r.eps == ro["eps"] and np.array_equal(r.info["theta_all"], ro["theta_all"])
    assert [it["eps"] for it in r.info["log"]] == [it["eps"] for it in ro["log"]]
    assert [it["ess"] for it in r.info["log"]] == [it["ess"] for it in ro["log"]]


_C4_ORACLE = {}


def _oracle_c4(orc, prior, cost, kw):
    if "r" not in _C4_ORACLE:      # 1.7 s of CPU, once for the three cases
        _C4_ORACLE["r"] = orc.smc(prior, cost, **kw)
    return _C4_ORACLE["r"]
