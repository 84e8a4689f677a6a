"""NegativeBinomial components on the AIS path with launches long enough (ntransitions >= 8) for
the kernel's per-launch lgamma(k + r) tables (ais_kernels.hpp `snb`, kabc_device.hpp the family's
case) and the lgamma(x + 1) lookup (include/kabc_lgamma1_table.h): the oracle computes both
lgammas directly, the trajectories must stay bit-identical -- one, two and three NegativeBinomial
components (the third has no table slot), counts below and beyond the 256 tabulated entries,
both posterior kinds, and a short launch (no tables) beside the long ones."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _priors(k):
    nb_small = k.NegativeBinomial(900 / 195, (900 / 195) / (30 + 900 / 195))   # the socks prior's
    nb_mid = k.NegativeBinomial(7.5, 0.2)
    nb_big = k.NegativeBinomial(60.0, 0.12)     # mean 440: most counts beyond the table
    return {
        "socks": k.Factored(nb_small, k.Beta(15, 2)),
        "two_nb": k.Factored(nb_mid, k.Normal(0, 2), nb_small),
        "three_nb": k.Factored(nb_mid, nb_small, k.Gamma(2.0, 1.5), k.NegativeBinomial(3.0, 0.4)),
        "beyond_table": k.Factored(nb_big, k.Uniform(-1, 1)),
        "straddle": k.Factored(k.NegativeBinomial(40.0, 0.15), k.NegativeBinomial(2.0, 0.5)),
    }


@pytest.mark.parametrize("name", ["socks", "two_nb", "three_nb", "beyond_table", "straddle"])
@pytest.mark.parametrize("nt", [3, 8, 25])
@pytest.mark.parametrize("kernelized", [True, False])
def test_negbin_lgamma_tables_bit_exact(k, orc, gpu_ctx, name, nt, kernelized):
    prior = _priors(k)[name]
    D = len(prior)
    cost = k.costs.GaussDist(np.resize(np.array([30.0, 0.5, 4.0, 1.0]), D) if name != "beyond_table"
                             else np.array([440.0, 0.0]))
    model = (k.ApproxKernelizedPosterior(prior, cost, 40.0) if kernelized
             else k.ApproxPosterior(prior, cost, 400.0))
    N, gens, seed = 700, 3, 4242
    ens = k.AisEnsemble(model, N, seed=seed).init()
    o = orc.OracleAIS(model, N, seed=seed).init()
    assert all(np.array_equal(a, b) for a, b in zip(ens.state()[:3], o.state()[:3]))
    got = ens.advance(gens, nt, collect=True)
    ref = o.generations_sync(gens, nt)
    assert np.array_equal(got, ref)
    assert all(np.array_equal(a, b) for a, b in zip(ens.state()[:3], o.state()[:3]))
    assert ens.stats() == o.stats()
    st = ens.stats()
    assert st["accepted"] > 0   # the chains move: the log-densities compared are not all -Inf
    if name == "beyond_table":
        assert np.max(got[..., 0]) >= 256   # counts past the tabulated range were visited


def test_negbin_tables_in_a_hiprtc_compiled_kernel(k, orc, gpu_ctx):
    """The same tables inside a half-generation kernel compiled at run time for a user cost
    (kabc_compile_cost_plugin): the snippet is the built-in Rosenbrock formula, so the trajectory
    must equal the built-in's and the oracle's."""
    from tests.test_user_cost import ROSEN_SRC
    user = k.costs.UserCost(ROSEN_SRC + "// negbin tables\n", dims=[2], posteriors=["kernelized"])
    prior = _priors(k)["socks"]
    N, nt, gens = 900, 12, 2
    got = k.AisEnsemble(k.ApproxKernelizedPosterior(prior, user, 50.0), N, seed=9).init().advance(
        gens, nt, collect=True)
    ref = k.AisEnsemble(k.ApproxKernelizedPosterior(prior, k.costs.Rosenbrock(), 50.0), N,
                        seed=9).init().advance(gens, nt, collect=True)
    assert np.array_equal(got, ref)
    o = orc.OracleAIS(k.ApproxKernelizedPosterior(prior, k.costs.Rosenbrock(), 50.0), N, seed=9).init()
    assert np.array_equal(got, o.generations_sync(gens, nt))
