"""-m gpu: the HIP AIS kernels against the CPU oracle's sync schedule, through the
C ABI.  Bar: BIT-EXACT positions, log-densities, accept masks and partner indices."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _models(k):
    U8 = k.Factored(*[k.Uniform(-5, 5)] * 8)
    N2 = k.Factored(k.Normal(0, 5), k.Normal(0, 5))
    rng = np.random.default_rng(7)
    H16 = k.Factored(k.Normal(0, 5), k.Uniform(0, 5), *[k.Normal(0, 1)] * 14)
    socks = k.Factored(k.NegativeBinomial(900 / 195, (900 / 195) / (30 + 900 / 195)), k.Beta(15, 2))
    mixed = k.Factored(k.Gamma(2.5, 0.7), k.LogNormal(0.3, 0.6), k.Exponential(2.0),
                       k.DiscreteUniform(1, 10), k.TruncatedNormal(0, 1, -1, 2))
    return {
        # every prior class x posterior kind x deterministic / stochastic cost, D = 1 .. 16
        "d1_dirac_kernelized": (k.ApproxKernelizedPosterior(k.Normal(1, 0.2), k.costs.DiracSq(1.5),
                                                            0.001), 12),
        "d1_mixture_threshold": (k.ApproxPosterior(k.Uniform(-10, 10), k.costs.Mixture(0.0), 0.5), 50),
        "d2_readme_sim": (k.ApproxKernelizedPosterior(
            k.Factored(k.Uniform(1, 3), k.Truncated(k.Normal(0, 0.1), 0, 100)),
            k.costs.NormalMeanStdSim(200, 2.0, 0.04), 0.005), 10),
        "d2_discrete_threshold": (k.ApproxPosterior(
            k.Factored(k.Normal(1, 0.5), k.DiscreteUniform(1, 10)), k.costs.NoisyQuadDU(5.5), 0.5), 100),
        "d2_general_negbin_beta": (k.ApproxKernelizedPosterior(socks, k.costs.GaussDist([40.0, 0.8]),
                                                               3.0), 200),
        "d5_general_mixed": (k.ApproxKernelizedPosterior(mixed, k.costs.NormShell(3.0), 0.5), 333),
        "d16_hier_sim": (k.ApproxKernelizedPosterior(H16, k.costs.HierGaussSim(rng.normal(size=14)),
                                                     0.3), 640),
        "d16_box": (k.ApproxKernelizedPosterior(k.Factored(*[k.Uniform(-3, 3)] * 16),
                                                k.costs.Rosenbrock(), 2.0), 130),
        "d2_wiener": (k.ApproxPosterior(k.Factored(k.Uniform(0, 1), k.Uniform(0, 4)),
                                        k.costs.WienerRms(np.sqrt(0.25 * np.arange(31.0) ** 2
                                                                  + 4.0 * np.arange(31.0))), 0.5), 50),
        "d2_banana_inf": (k.ApproxKernelizedPosterior(N2, k.costs.NoisyBanana(0.5), 5.0), 64),
        "C3_rosenbrock_d8": (k.ApproxKernelizedPosterior(U8, k.costs.Rosenbrock(), 1.0), 2048),
        "C2_gauss_d2": (k.ApproxKernelizedPosterior(N2, k.costs.GaussDist([1.0, -0.5]), 0.1), 4096),
        "threshold_d2": (k.ApproxPosterior(N2, k.costs.GaussDist([1.0, -0.5]), 0.5), 1000),
        "odd_N_d3": (k.ApproxKernelizedPosterior(
            k.Factored(k.Normal(0, 1), k.Uniform(-2, 2), k.Normal(1, 2)),
            k.costs.NormShell(1.5), 0.2), 1001),
    }


NAMES = ["C3_rosenbrock_d8", "C2_gauss_d2", "threshold_d2", "odd_N_d3", "d1_dirac_kernelized",
         "d1_mixture_threshold", "d2_readme_sim", "d2_discrete_threshold", "d2_general_negbin_beta",
         "d5_general_mixed", "d16_hier_sim", "d16_box", "d2_wiener", "d2_banana_inf"]


# the cases that fit the one-workgroup driver of small ensembles (csrc/ais_small_kernel.hpp: N <= 512,
# <= 256 from nine parameters on) run on BOTH drivers
SMALL = ["d1_dirac_kernelized", "d1_mixture_threshold", "d2_readme_sim", "d2_discrete_threshold", "d2_general_negbin_beta",
         "d5_general_mixed", "d16_box", "d2_wiener", "d2_banana_inf"]
CASES = [(n, "halves") for n in NAMES] + [(n, "small") for n in SMALL]


@pytest.mark.parametrize("name,driver", CASES)
def test_ais_generation_bit_exact(k, orc, gpu_ctx, monkeypatch, name, driver):
    model, N = _models(k)[name]
    nt, gens, seed = (1 if name == "d1_dirac_kernelized" else 5), 4, 11
    monkeypatch.setenv("KABC_AIS_SMALL", "1" if driver == "small" else "0")
    ens = k.AisEnsemble(model, N, seed=seed).init()
    assert ens.driver == driver
    o = orc.OracleAIS(model, N, seed=seed).init()
    # init parity (step(init), src/KissABC.jl:35-64)
    x0, lp0, ll0, _ = ens.state()
    xo, lpo, llo, _ = o.state()
    assert np.array_equal(x0, xo) and np.array_equal(lp0, lpo) and np.array_equal(ll0, llo)
    ens.set_debug(nt)
    got = ens.advance(1, nt, collect=True)
    dbg = ens.get_debug(nt)
    ref, tr = o.generations_sync(1, nt, trace=True)
    assert np.array_equal(got, ref)
    # index-exact: move ids, accept flags, cost-evaluated flags
    assert np.array_equal(dbg[:, :, 0], tr[0, :, :, 0])
    assert np.array_equal(dbg[:, :, 1], tr[0, :, :, 1])
    assert np.array_equal(dbg[:, :, 5], tr[0, :, :, 5])
    # partner ids: device reports rows inside the complementary half
    N0 = (N + 1) // 2
    base = np.where(np.arange(N) < N0, N0, 0)[:, None]
    for col in (2, 3, 4):
        d = dbg[:, :, col].astype(np.int64)
        r = tr[0, :, :, col].astype(np.int64)
        assert np.array_equal(np.where(d >= 0, d + base, -1), r)
    ens.set_debug(0)
    got = ens.advance(gens, nt, collect=True)
    ref = o.generations_sync(gens, nt)
    assert np.array_equal(got, ref)
    xs, lps, lls, t = ens.state()
    xo, lpo, llo, to = o.state()
    assert t == to == nt * (gens + 1)
    assert np.array_equal(xs, xo) and np.array_equal(lps, lpo) and np.array_equal(lls, llo)
    assert ens.stats() == o.stats()


@pytest.mark.parametrize("N,n_draws,nt,kib", [(10, 1000, 7, 0), (600, 201, 7, 8), (130, 37, 5, 1)])
def test_prepared_cost_prepass_bit_exact(k, orc, gpu_ctx, monkeypatch, N, n_draws, nt, kib):
    """README.md:31-57's simulator: the parameter-independent sums of every (walker, sub-step) of
    a launch come from the grid-wide pre-pass, one wavefront per cost evaluation, 64 lanes
    sharing its draws in the contract's summation order (include/kabc_costs.h) -- the oracle runs
    the same slices one after the other.  README size (AIS(10), 1000 draws), an odd number of
    draws, fewer pairs than lanes; `kib` bounds the pre-pass buffer so that a launch is cut into
    blocks of sub-steps (debug records and trace still line up).  (The launch-per-half-generation driver;
    the one-workgroup driver with this cost: tests/test_gpu_ais_small.py.)"""
    monkeypatch.setenv("KABC_AIS_SMALL", "0")
    if kib:
        monkeypatch.setenv("KABC_AUX_KIB", str(kib))
    prior = k.Factored(k.Uniform(1, 3), k.Truncated(k.Normal(0, 0.1), 0, 100))
    model = k.ApproxKernelizedPosterior(prior, k.costs.NormalMeanStdSim(n_draws, 2.0, 0.04), 0.005)
    ens = k.AisEnsemble(model, N, seed=3).init()
    o = orc.OracleAIS(model, N, seed=3).init()
    assert np.array_equal(ens.state()[2], o.state()[2])       # init evaluates the cost in one thread
    ens.set_debug(nt)
    got = ens.advance(1, nt, collect=True)
    dbg = ens.get_debug(nt)
    ref, tr = o.generations_sync(1, nt, trace=True)
    assert np.array_equal(got, ref)
    assert np.array_equal(dbg[:, :, 1], tr[0, :, :, 1]) and np.array_equal(dbg[:, :, 5], tr[0, :, :, 5])
    ens.set_debug(0)
    assert np.array_equal(ens.advance(3, nt, collect=True), o.generations_sync(3, nt))
    xs, lps, lls, _ = ens.state()
    xo, lpo, llo, _ = o.state()
    assert np.array_equal(xs, xo) and np.array_equal(lps, lpo) and np.array_equal(lls, llo)
    assert ens.stats() == o.stats()


def test_prepared_cost_prepass_batched_chains(k, orc, gpu_ctx, monkeypatch):
    """chains as a grid dimension: the pre-pass keys every chain's draws by its own seed"""
    monkeypatch.setenv("KABC_AIS_SMALL", "0")
    prior = k.Factored(k.Uniform(1, 3), k.Truncated(k.Normal(0, 0.1), 0, 100))
    model = k.ApproxKernelizedPosterior(prior, k.costs.NormalMeanStdSim(100, 2.0, 0.04), 0.005)
    seeds = [5, 6, 7]
    ens = k.AisEnsemble(model, 12, seeds=seeds).init()
    got = ens.advance(3, 4, collect=True)                    # [gen][chain][N][D]
    for c, sd in enumerate(seeds):
        ref = orc.OracleAIS(model, 12, seed=sd).init().generations_sync(3, 4)
        assert np.array_equal(got[:, c], ref), f"chain {c}"
