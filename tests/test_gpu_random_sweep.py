"""Seeded random sweep of the AIS and SMC device paths against the oracle: random
dimension, Factored prior (all nine families), DeviceCost, posterior kind, ensemble
size, ntransitions and seed -- every trajectory must be bit-identical.  Widens the
hand-picked cases of test_gpu_ais_parity.py / test_gpu_smc_parity.py."""
import numpy as np
import pytest

import os

pytestmark = pytest.mark.gpu
# KABC_SWEEP_SCALE=k multiplies the number of random configurations (soak runs)
_SCALE = max(1, int(os.environ.get("KABC_SWEEP_SCALE", "1")))


def _random_prior(k, rng, D):
    comps = []
    for _ in range(D):
        kind = rng.integers(0, 9)
        if kind == 0:
            lo = rng.uniform(-6, 0)
            comps.append(k.Uniform(lo, lo + rng.uniform(1, 8)))
        elif kind == 1:
            comps.append(k.Normal(rng.uniform(-2, 2), rng.uniform(0.3, 4)))
        elif kind == 2:
            comps.append(k.TruncatedNormal(rng.uniform(-1, 1), rng.uniform(0.5, 2), -2.0, 3.0))
        elif kind == 3:
            comps.append(k.Beta(rng.uniform(0.8, 5), rng.uniform(0.8, 5)))
        elif kind == 4:
            a = int(rng.integers(-3, 3))
            comps.append(k.DiscreteUniform(a, a + int(rng.integers(2, 12))))
        elif kind == 5:
            comps.append(k.NegativeBinomial(rng.uniform(1.5, 8), rng.uniform(0.2, 0.8)))
        elif kind == 6:
            comps.append(k.Exponential(rng.uniform(0.5, 3)))
        elif kind == 7:
            comps.append(k.Gamma(rng.uniform(1.2, 4), rng.uniform(0.4, 2)))
        else:
            comps.append(k.LogNormal(rng.uniform(-0.5, 0.5), rng.uniform(0.2, 0.8)))
    return k.Factored(*comps)


def _random_cost(k, rng, D):
    choices = ["gauss", "shell"] + (["rosen"] if D >= 2 else []) + (["hier"] if D >= 3 else [])
    c = rng.choice(choices)
    if c == "gauss":
        return k.costs.GaussDist(rng.uniform(-1, 2, D))
    if c == "shell":
        return k.costs.NormShell(rng.uniform(0.5, 3))
    if c == "rosen":
        return k.costs.Rosenbrock()
    return k.costs.HierGaussSim(rng.normal(size=D - 2))


@pytest.mark.parametrize("case", range(60 * _SCALE))
def test_ais_random_case_bit_exact(k, orc, gpu_ctx, case):
    rng = np.random.default_rng(1000 + case)
    D = int(rng.integers(1, 17))
    prior = _random_prior(k, rng, D)
    cost = _random_cost(k, rng, D)
    if rng.random() < 0.6:
        model = k.ApproxKernelizedPosterior(prior, cost, float(rng.uniform(0.5, 5)))
    else:
        model = k.ApproxPosterior(prior, cost, float(rng.uniform(2, 20)))
    N = int(rng.integers(D + 5, 2500))
    nt, gens, seed = int(rng.integers(1, 8)), int(rng.integers(1, 4)), int(rng.integers(0, 2 ** 31))
    ens = k.AisEnsemble(model, N, seed=seed).init()
    o = orc.OracleAIS(model, N, seed=seed).init()
    assert all(np.array_equal(a, b) for a, b in zip(ens.state()[:3], o.state()[:3]))
    got = ens.advance(gens, nt, collect=True)
    ref = o.generations_sync(gens, nt)
    assert np.array_equal(got, ref)
    assert all(np.array_equal(a, b) for a, b in zip(ens.state()[:3], o.state()[:3]))
    assert ens.stats() == o.stats()


@pytest.mark.parametrize("case", range(32 * _SCALE))
def test_smc_random_case_bit_exact(k, orc, gpu_ctx, monkeypatch, case):
    rng = np.random.default_rng(5000 + case)
    D = int(rng.integers(1, 9))
    prior = _random_prior(k, rng, D)
    cost = _random_cost(k, rng, D)
    kw = dict(nparticles=int(rng.integers(max(50, 4 * D), 6000)), alpha=float(rng.uniform(0.6, 0.95)),
              epstol=float(rng.uniform(0.05, 0.5)), mcmc_retrys=int(rng.integers(0, 4)),
              seed=int(rng.integers(0, 2 ** 31)))
    blocks = [None, "1", "3", "5"][case % 4]
    if blocks is None:
        monkeypatch.delenv("KABC_SMC_SELECT_BLOCKS", raising=False)
    else:
        monkeypatch.setenv("KABC_SMC_SELECT_BLOCKS", blocks)
    r = k.smc(prior, cost, return_array=True, **kw)
    ro = orc.smc(prior, cost, **kw)
    assert r.info["iterations"] == ro["iterations"] and r.eps == ro["eps"]
    assert np.array_equal(r.info["theta_all"], ro["theta_all"])
    assert np.array_equal(r.info["alive"], ro["alive"])
    assert [(it["eps"], it["ess"], it["accepted"]) for it in r.info["log"]] == \
        [(it["eps"], it["ess"], it["accepted"]) for it in ro["log"]]


@pytest.mark.parametrize("case", range(8 * _SCALE))
def test_pfilter_random_case_bit_exact(k, orc, gpu_ctx, monkeypatch, case):
    rng = np.random.default_rng(9000 + case)
    D = int(rng.integers(1, 5))
    prior = _random_prior(k, rng, D)
    cost = _random_cost(k, rng, D)
    N = int(rng.integers(200, 5000))
    kw = dict(q=float(rng.uniform(0.5, 0.8)), eff_tol=0.1, epstol=float(rng.uniform(0.1, 0.5)),
              max_iters=40, seed=int(rng.integers(0, 2 ** 31)))
    if case % 2:
        monkeypatch.setenv("KABC_SMC_SELECT_BLOCKS", "2")
    else:
        monkeypatch.delenv("KABC_SMC_SELECT_BLOCKS", raising=False)
    r = k.pfilter(prior, cost, N, return_array=True, **kw)
    ro = orc.pfilter(prior, cost, N, **kw)
    assert np.array_equal(r.P, ro["P"]) and np.array_equal(r.C, ro["C"])


def _pooled_prior(k, rng, D, pool):
    """Components drawn from ONE prior class of the half-generation kernel (csrc/ais_kernels.hpp
    "Prior classes"): "box" Uniform / DiscreteUniform, "normal" plain Normal, "gaussbox" those plus
    truncated Normal -- the classes the unrestricted draw above almost never produces at D > 3."""
    comps = []
    for _ in range(D):
        kinds = {"box": (0, 4), "normal": (1,), "gaussbox": (0, 1, 2, 4)}[pool]
        kind = kinds[int(rng.integers(0, len(kinds)))]
        if kind == 0:
            lo = rng.uniform(-6, 0)
            comps.append(k.Uniform(lo, lo + rng.uniform(1, 8)))
        elif kind == 1:
            comps.append(k.Normal(rng.uniform(-2, 2), rng.uniform(0.3, 4)))
        elif kind == 2:
            comps.append(k.TruncatedNormal(rng.uniform(-1, 1), rng.uniform(0.5, 2), -2.0, 3.0))
        else:
            a = int(rng.integers(-3, 3))
            comps.append(k.DiscreteUniform(a, a + int(rng.integers(2, 12))))
    return k.Factored(*comps)


@pytest.mark.parametrize("case", range(36 * _SCALE))
def test_ais_random_prior_class_bit_exact(k, orc, gpu_ctx, case):
    rng = np.random.default_rng(20000 + case)
    pool = ["box", "normal", "gaussbox"][case % 3]
    D = int(rng.integers(1, 17))
    prior = _pooled_prior(k, rng, D, pool)
    if D == 2 and rng.random() < 0.5:  # a prepared cost: the pre-pass in front of the launch
        cost = k.costs.NormalMeanStdSim(int(rng.integers(1, 300)), 2.0, 0.05)
    else:
        cost = _random_cost(k, rng, D)
    kind = int(rng.integers(0, 3))
    if kind == 0:
        model = k.ApproxKernelizedPosterior(prior, cost, float(rng.uniform(0.5, 5)))
    elif kind == 1:
        model = k.ApproxPosterior(prior, cost, float(rng.uniform(2, 20)))
    else:   # classical MCMC: the cost's value IS the log-density, the prior only starts the walkers
        model = k.CommonLogDensity(D, prior, cost)
    N = int(rng.integers(D + 5, 2500))
    nt, gens, seed = int(rng.integers(1, 8)), int(rng.integers(1, 4)), int(rng.integers(0, 2 ** 31))
    ens = k.AisEnsemble(model, N, seed=seed).init()
    o = orc.OracleAIS(model, N, seed=seed).init()
    assert all(np.array_equal(a, b) for a, b in zip(ens.state()[:3], o.state()[:3]))
    got = ens.advance(gens, nt, collect=True)
    ref = o.generations_sync(gens, nt)
    assert np.array_equal(got, ref)
    assert all(np.array_equal(a, b) for a, b in zip(ens.state()[:3], o.state()[:3]))
    assert ens.stats() == o.stats()


@pytest.mark.parametrize("case", range(24 * _SCALE))
def test_sharded_random_case_bit_exact(k, orc, gpu_ctx, monkeypatch, case):
    """The walker-sharded path on emulated ranks (P2P backend, every rank on device 0) with
    random world size, ensemble size (down to shards that own nothing), exchange chunks, prior
    class, cost and ntransitions: every rank's copy of the ensemble equals the oracle's."""
    rng = np.random.default_rng(30000 + case)
    world = int(rng.integers(2, 9))
    K = [1, 2, 3, 4, 7, 16][int(rng.integers(0, 6))]
    monkeypatch.setenv("KABC_EXCHANGE_CHUNKS", str(K))
    D = int(rng.integers(1, 13))
    pool = ["any", "box", "normal", "gaussbox"][case % 4]
    prior = _random_prior(k, rng, D) if pool == "any" else _pooled_prior(k, rng, D, pool)
    cost = _random_cost(k, rng, D)
    if rng.random() < 0.5:
        model = k.ApproxKernelizedPosterior(prior, cost, float(rng.uniform(0.5, 5)))
    else:
        model = k.ApproxPosterior(prior, cost, float(rng.uniform(2, 20)))
    N = int(rng.integers(D + 5, 60)) if case % 3 == 0 else int(rng.integers(60, 5000))
    nt, gens, seed = int(rng.integers(1, 7)), int(rng.integers(1, 4)), int(rng.integers(0, 2 ** 31))
    grp = k.EnsembleGroup(model, N, seed=seed, devices=[0] * world, backend="p2p").init()
    o = orc.OracleAIS(model, N, seed=seed).init()
    assert np.array_equal(grp.ensemble(world - 1), o.state()[0])
    for _ in range(2):
        grp.advance(gens, nt)
        o.generations_sync(gens, nt, collect=False)
        xo, lpo, llo, _ = o.state()
        for r in range(world):
            assert np.array_equal(grp.ensemble(r), xo), f"rank {r}"
    x, lp, ll = grp.state()
    assert np.array_equal(x, xo) and np.array_equal(lp, lpo) and np.array_equal(ll, llo)
    assert grp.stats() == o.stats()
    grp.close()
