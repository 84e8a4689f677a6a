"""-m gpu: the one-workgroup AIS driver of small ensembles (csrc/ais_small_kernel.hpp: every
generation of a kabc_ais_advance call in ONE launch, both halves in LDS, producer waves running
ahead through a ring of LDS slots) against the CPU oracle's sync schedule and against the
launch-per-half-generation driver.  Bar: BIT-EXACT trace rows, state, counters, debug records.

The shapes are the reference's own: AIS(10) .. AIS(500), ntransitions = 1, long burn-ins
(src/KissABC.jl:66-80, test/runtests.jl:82-131,177-198)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _n2(k):
    return k.ApproxKernelizedPosterior(k.Factored(k.Normal(0, 5), k.Normal(0, 5)),
                                       k.costs.GaussDist([1.0, -0.5]), 0.1)


def _u8(k):
    return k.ApproxKernelizedPosterior(k.Factored(*[k.Uniform(-5, 5)] * 8), k.costs.Rosenbrock(), 1.0)


def _h16(k):
    rng = np.random.default_rng(7)
    H16 = k.Factored(k.Normal(0, 5), k.Uniform(0, 5), *[k.Normal(0, 1)] * 14)
    return k.ApproxKernelizedPosterior(H16, k.costs.HierGaussSim(rng.normal(size=14)), 0.3)


def _mixed5(k):
    mixed = k.Factored(k.Gamma(2.5, 0.7), k.LogNormal(0.3, 0.6), k.Exponential(2.0),
                       k.DiscreteUniform(1, 10), k.TruncatedNormal(0, 1, -1, 2))
    return k.ApproxKernelizedPosterior(mixed, k.costs.NormShell(3.0), 0.5)


def _check_against_oracle(k, orc, model, N, nt, gens, seed, chunks=(None,)):
    ens = k.AisEnsemble(model, N, seed=seed).init()
    assert ens.driver == "small"
    o = orc.OracleAIS(model, N, seed=seed).init()
    for g in (chunks if chunks[0] is not None else (gens,)):
        got = ens.advance(g, nt, collect=True)
        ref = o.generations_sync(g, nt)
        assert np.array_equal(got, ref)
    xs, lps, lls, t = ens.state()
    xo, lpo, llo, to = o.state()
    assert t == to
    assert np.array_equal(xs, xo) and np.array_equal(lps, lpo) and np.array_equal(lls, llo)
    assert ens.stats() == o.stats()
    return ens, o


# one consumer (N <= 128), two consumers with one / two batches each, ragged last batches, a
# second consumer that owns a batch of half 0 only (N = 129, 130), the maximum (512)
@pytest.mark.parametrize("N", [7, 10, 12, 50, 100, 127, 128, 129, 130, 200, 256, 257, 385, 511, 512])
@pytest.mark.parametrize("nt", [1, 4])
def test_small_driver_sizes(k, orc, gpu_ctx, N, nt):
    _check_against_oracle(k, orc, _n2(k), N, nt, 9, seed=N + nt)


@pytest.mark.parametrize("make,N", [(_u8, 13), (_u8, 64), (_u8, 300), (_u8, 512), (_h16, 21), (_h16, 100),
                                    (_h16, 256), (_mixed5, 10), (_mixed5, 333)])
@pytest.mark.parametrize("nt", [1, 7])
def test_small_driver_models(k, orc, gpu_ctx, make, N, nt):
    """BOX at D = 8 (the headline model), a stochastic cost whose leading normals the producers expand
    (late slot release, D = 16: the smaller ring and the 256-walker limit), the GENERAL class."""
    _check_against_oracle(k, orc, make(k), N, nt, 6, seed=3)


def test_small_driver_limits(k, gpu_ctx, monkeypatch):
    assert k.AisEnsemble(_u8(k), 512).driver == "small"
    assert k.AisEnsemble(_u8(k), 513).driver == "halves"
    assert k.AisEnsemble(_h16(k), 256).driver == "small"
    assert k.AisEnsemble(_h16(k), 257).driver == "halves"
    # a cost with a grid-wide pre-pass (README.md:31-57's simulator): one pre-pass launch per half for all of a
    # call's sub-steps, then the one workgroup
    readme = k.ApproxKernelizedPosterior(k.Factored(k.Uniform(1, 3), k.Truncated(k.Normal(0, 0.1), 0, 100)),
                                         k.costs.NormalMeanStdSim(100, 2.0, 0.04), 0.005)
    assert k.AisEnsemble(readme, 10).driver == "small"
    monkeypatch.setenv("KABC_AIS_SMALL", "0")
    assert k.AisEnsemble(_u8(k), 100).driver == "halves"


@pytest.mark.parametrize("N,nt,n_draws,kib", [(10, 100, 1000, 0), (10, 7, 999, 0), (12, 5, 37, 0), (130, 3, 64, 0),
                                              (10, 20, 200, 4)])
def test_prepared_cost_on_the_small_driver(k, orc, gpu_ctx, monkeypatch, N, nt, n_draws, kib):
    """README.md:31-57's simulator on the one-workgroup driver: the words of every (walker, sub-step) of a call
    from one pre-pass launch per half (csrc/ais_aux_kernels.hpp), copied into the ring by the producers;
    `kib` bounds the pre-pass buffer so that the call is cut into blocks of generations.  README size, an odd
    number of draws, fewer pairs than lanes, more than one batch per half; chains as a grid dimension."""
    monkeypatch.delenv("KABC_AIS_SMALL", raising=False)
    if kib:
        monkeypatch.setenv("KABC_AUX_KIB", str(kib))
    prior = k.Factored(k.Uniform(1, 3), k.Truncated(k.Normal(0, 0.1), 0, 100))
    model = k.ApproxKernelizedPosterior(prior, k.costs.NormalMeanStdSim(n_draws, 2.0, 0.04), 0.005)
    ens = k.AisEnsemble(model, N, seed=3).init()
    assert ens.driver == "small"
    o = orc.OracleAIS(model, N, seed=3).init()
    gens = 6 if kib else 3
    assert np.array_equal(ens.advance(gens, nt, collect=True), o.generations_sync(gens, nt))
    assert np.array_equal(ens.advance(2, nt, collect=True), o.generations_sync(2, nt))      # resumed: t goes on
    xs, lps, lls, _ = ens.state()
    xo, lpo, llo, _ = o.state()
    assert np.array_equal(xs, xo) and np.array_equal(lps, lpo) and np.array_equal(lls, llo)
    assert ens.stats() == o.stats()
    if N == 12:
        seeds = [5, 6, 7]
        b = k.AisEnsemble(model, 12, seeds=seeds).init()
        assert b.driver == "small"
        got = b.advance(3, 4, collect=True)                    # [gen][chain][N][D]
        for c, sd in enumerate(seeds):
            assert np.array_equal(got[:, c], orc.OracleAIS(model, 12, seed=sd).init().generations_sync(3, 4)), c


def test_reference_shaped_calls(k, orc, gpu_ctx):
    """discard_initial then keep, as sample() drives it (ceil(discard / N) generations without a trace,
    ceil(Ns / N) with one): AIS(50), 2000 steps of burn-in, 500 samples, ntransitions = 1"""
    model, N = _n2(k), 50
    _check_against_oracle(k, orc, model, N, 1, None, seed=1, chunks=(40, 10))
    out = k.sample(model, k.AIS(N), 500, discard_initial=2000, seed=1, return_array=True)
    o = orc.OracleAIS(model, N, seed=1).init()
    o.generations_sync(40, 1, collect=False)
    assert np.array_equal(out, o.generations_sync(10, 1).reshape(-1, 2)[:500])


def test_both_drivers_same_bits_long_run(k, orc, gpu_ctx, monkeypatch):
    """1000 generations in one launch against 2000 launches, with the debug records of the last one"""
    model, N, nt = _u8(k), 100, 1
    a = k.AisEnsemble(model, N, seed=9).init()
    monkeypatch.setenv("KABC_AIS_SMALL", "0")
    b = k.AisEnsemble(model, N, seed=9).init()
    assert (a.driver, b.driver) == ("small", "halves")
    a.set_debug(nt)
    b.set_debug(nt)
    ta = a.advance(1000, nt, collect=True)
    tb = b.advance(1000, nt, collect=True)
    assert np.array_equal(ta, tb)
    assert np.array_equal(a.get_debug(nt), b.get_debug(nt))
    for u, v in zip(a.state(), b.state()):
        assert np.array_equal(u, v)
    assert a.stats() == b.stats()


def test_debug_records_index_exact(k, orc, gpu_ctx):
    model, N, nt = _mixed5(k), 77, 5
    ens = k.AisEnsemble(model, N, seed=4).init()
    o = orc.OracleAIS(model, N, seed=4).init()
    ens.set_debug(nt)
    got = ens.advance(1, nt, collect=True)
    dbg = ens.get_debug(nt)
    ref, tr = o.generations_sync(1, nt, trace=True)
    assert np.array_equal(got, ref)
    for col in (0, 1, 5):
        assert np.array_equal(dbg[:, :, col], tr[0, :, :, col])
    N0 = (N + 1) // 2
    base = np.where(np.arange(N) < N0, N0, 0)[:, None]
    for col in (2, 3, 4):
        d = dbg[:, :, col].astype(np.int64)
        assert np.array_equal(np.where(d >= 0, d + base, -1), tr[0, :, :, col].astype(np.int64))


def test_resume_from_state(k, orc, gpu_ctx):
    """AISState round trip (src/KissABC.jl:25-33): get_state -> a new handle -> set_state"""
    model, N = _n2(k), 60
    a = k.AisEnsemble(model, N, seed=5).init()
    a.advance(7, 2)
    x, lp, ll, t = a.state()
    b = k.AisEnsemble(model, N, seed=5)
    b.set_state(x, lp, ll, t)
    assert np.array_equal(a.advance(5, 2, collect=True), b.advance(5, 2, collect=True))
    o = orc.OracleAIS(model, N, seed=5).init()
    o.generations_sync(7, 2, collect=False)
    o.generations_sync(5, 2, collect=False)
    assert np.array_equal(b.state()[0], o.state()[0])


def test_invalid_starting_sample(k, gpu_ctx):
    """accept(): `old log-density is invalid` -> error("starting sample invalid.")  src/types.jl:70"""
    model, N = _n2(k), 20
    ens = k.AisEnsemble(model, N, seed=1).init()
    x, lp, ll, t = ens.state()
    ll[13] = -np.inf
    ens.set_state(x, lp, ll, t)
    with pytest.raises(k.KabcError, match="starting sample invalid"):
        ens.advance(3, 1)


def test_batched_chains(k, orc, gpu_ctx):
    """MCMCThreads (src/KissABC.jl:96-104,108): chain = workgroup of the one launch"""
    model, N = _u8(k), 20
    seeds = [11, 12, 13, 14, 15]
    ens = k.AisEnsemble(model, N, seeds=seeds).init()
    assert ens.driver == "small"
    ens.advance(3, 1)
    got = ens.advance(6, 2, collect=True)                     # [gen][chain][N][D]
    for c, sd in enumerate(seeds):
        o = orc.OracleAIS(model, N, seed=sd).init()
        o.generations_sync(3, 1, collect=False)
        assert np.array_equal(got[:, c], o.generations_sync(6, 2)), f"chain {c}"
    out = k.sample(model, k.AIS(13), k.MCMCThreads(), 30, 4, seed=2, discard_initial=26, return_array=True)
    for c, sd in enumerate(k.api.chain_seeds(2, 4)):
        o = orc.OracleAIS(model, 13, seed=sd).init()
        o.generations_sync(2, 1, collect=False)
        assert np.array_equal(out[c * 30:(c + 1) * 30], o.generations_sync(3, 1).reshape(-1, 8)[:30])


def test_trace_in_several_blocks(k, orc, gpu_ctx, monkeypatch):
    """a call whose trace exceeds the device buffer runs as several launches (here: 1 MiB blocks)"""
    monkeypatch.setenv("KABC_TRACE_CHUNK_MIB", "1")
    model, N = _n2(k), 100
    ens = k.AisEnsemble(model, N, seed=6).init()
    got = ens.advance(1500, 1, collect=True)                  # 1600 B per generation: 3 blocks
    ref = orc.OracleAIS(model, N, seed=6).init().generations_sync(1500, 1)
    assert np.array_equal(got, ref)


def test_user_cost_and_threshold_posterior(k, orc, gpu_ctx):
    """a run-time compiled cost (hipRTC unit: its own small kernel) under ApproxPosterior"""
    cost = k.costs.UserCost('''
KABC_HD double kabc_user_cost(const double* x, int D, const double* params,
                              const double* data, int64_t ndata, kabc_cost_rng_t* rng) {
    double z0, z1;
    kabc_cost_rng_normal2(rng, &z0, &z1);
    return kabc_fabs(x[0] * x[1] - params[0] + 0.05 * z0) + 0.01 * kabc_fabs(z1);
}''', dims=[2], params=[1.5], name="small_prod")
    orc.register_user_cost(cost)
    model = k.ApproxPosterior(k.Factored(k.Uniform(0, 3), k.Normal(1, 1)), cost, 0.4)
    ens, _ = _check_against_oracle(k, orc, model, 40, 3, 8, seed=8)
    assert ens.driver == "small"
