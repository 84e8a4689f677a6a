"""-m gpu: BASELINE.json configurations through the product API (sample / AIS),
full sizes, plus the torch-plumbed sharded driver on one GPU."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_c3_full_size_bit_exact_and_invariants(k, orc, gpu_ctx):
    """configs[2]: AIS 65536 walkers x 8-param Rosenbrock-like cost, ntransitions 16."""
    U8 = k.Factored(*[k.Uniform(-5, 5)] * 8)
    model = k.ApproxKernelizedPosterior(U8, k.costs.Rosenbrock(), 1.0)
    N, nt = 65536, 16
    ens = k.AisEnsemble(model, N, seed=1).init()
    got = ens.advance(2, nt, collect=True)
    ref = orc.OracleAIS(model, N, seed=1).init().generations_sync(2, nt)
    assert np.array_equal(got, ref)
    st = ens.stats()
    assert st["proposals"] == 2 * N * nt
    assert st["cost_evals"] <= st["proposals"] and st["accepted"] <= st["cost_evals"]
    # size-independent properties after many more generations
    ens.advance(30, nt)
    x, lp, ll, t = ens.state()
    assert t == 32 * nt
    assert np.all(np.abs(x) <= 5) and np.all(np.isfinite(lp + ll))
    assert np.all(lp == lp[0])                     # box prior: constant log-density inside
    c = np.sqrt((100 * (x[:, 1:] - x[:, :-1] ** 2) ** 2 + (1 - x[:, :-1]) ** 2).sum(1))
    assert np.allclose(ll, -0.5 * c * c, rtol=1e-13)   # stored loglik is consistent with x


def test_c2_posterior_means_within_1e3(k, gpu_ctx):
    """configs[1]: AIS 4096 walkers, 2-param Gaussian cost.  Analytic posterior
    (SURVEY §8c): mean = c*2500/2501, sd = sqrt(1/(1/25+100)); tolerance 1e-3."""
    prior = k.Factored(k.Normal(0, 5), k.Normal(0, 5))
    model = k.ApproxKernelizedPosterior(prior, k.costs.GaussDist([1.0, -0.5]), 0.1)
    res = k.sample(model, k.AIS(4096), 4096 * 300, ntransitions=4, discard_initial=4096 * 100,
                   seed=1, return_array=True)
    mean_ref = np.array([1.0, -0.5]) * 2500 / 2501
    assert res.shape == (4096 * 300, 2)
    assert np.all(np.abs(res.mean(0) - mean_ref) < 1e-3)
    assert np.all(np.abs(res.std(0) - (1 / (1 / 25 + 100)) ** 0.5) < 1e-3)


def test_readme_example_c1_on_device(k, gpu_ctx):
    """configs[0] (README.md:31-66): AIS(10), ntransitions=100, 1000 samples."""
    rng = np.random.default_rng(0)
    tdata = rng.normal(2.0, 0.04, 1000)
    prior = k.Factored(k.Uniform(1, 3), k.Truncated(k.Normal(0, 0.1), 0, 100))
    cost = k.costs.NormalMeanStdSim(1000, tdata.mean(), tdata.std(ddof=1))
    plan = k.ApproxKernelizedPosterior(prior, cost, 0.005)
    res = k.sample(plan, k.AIS(10), 1000, ntransitions=100, discard_initial=300, seed=1)
    assert len(res) == 2 and len(res[0]) == 1000
    assert abs(res[0].mean() - tdata.mean()) < 0.01
    assert abs(res[1].mean() - tdata.std(ddof=1)) < 0.005


def test_errors_match_reference(k, gpu_ctx):
    model = k.ApproxKernelizedPosterior(k.Factored(k.Normal(0, 5), k.Normal(0, 5)),
                                        k.costs.GaussDist([0, 0]), 0.1)
    with pytest.raises(k.KabcError) as e:
        k.sample(model, k.AIS(6), 10)
    assert str(e.value) == ("nparticles = 6 is insufficient, set number of particles in AIS(⋅) "
                            "atleast to 7")
    never = k.ApproxKernelizedPosterior(k.Factored(k.Uniform(0, 1), k.Uniform(0, 1)),
                                        k.costs.NoisyBanana(1.0), 0.1)
    with pytest.raises(k.KabcError) as e:
        k.sample(never, k.AIS(50), 10, retry_sampling=10)
    assert str(e.value) == ("Prior leads to ∞ costs too often, tune the prior or increase "
                            "`retry_sampling`.")
    ens = k.AisEnsemble(model, 16, seed=1).init()
    x, lp, ll, _ = ens.state()
    lp[3] = -np.inf
    ens.set_state(x, lp, ll, 0)
    with pytest.raises(k.KabcError) as e:
        ens.advance(1, 1)
    assert str(e.value) == "starting sample invalid."


def test_mcmcthreads_chains_are_independent(k, gpu_ctx):
    # test/runtests.jl:88-104
    abc = k.ApproxKernelizedPosterior(k.Normal(1, 0.2), k.costs.DiracSq(1.5), 0.001)
    res = k.sample(abc, k.AIS(12), k.MCMCThreads(), 100, 50, discard_initial=50 * 12, seed=3)
    mu = np.asarray(res)
    assert mu.shape == (5000,)
    sim = mu * mu + 1
    assert abs(sim.mean() - 1.5) < 2 * max(sim.std(), 1e-3)
    assert len(np.unique(mu[:100])) > 1 and not np.array_equal(mu[:100], mu[100:200])


def test_sharded_driver_world1_on_gpu(k, orc, gpu_ctx):
    """The torch-plumbed path bench.py uses: torch-owned half buffers + current stream."""
    import torch
    from kissabc_jl_amd.sharded import ShardedAIS
    U8 = k.Factored(*[k.Uniform(-5, 5)] * 8)
    model = k.ApproxKernelizedPosterior(U8, k.costs.Rosenbrock(), 1.0)
    sh = ShardedAIS(model, 4096, seed=5, device=torch.device("cuda", 0)).init()
    sh.advance(3, 7)
    torch.cuda.synchronize()
    pos = sh.positions().cpu().numpy()
    o = orc.OracleAIS(model, 4096, seed=5).init()
    o.generations_sync(3, 7, collect=False)
    assert np.array_equal(pos, o.state()[0])
    assert sh.global_stats() == o.stats()


def test_two_ranks_emulated_on_one_gpu(k, orc, gpu_ctx):
    """The sharded kernel path for rank > 0 (row offsets, global walker ids, partner
    draws over the GLOBAL complementary half): two world-2 handles share the same
    global half buffers on one GPU, so no collective is needed; the result must
    equal the single-process oracle bit for bit."""
    import torch
    from kissabc_jl_amd.sharded import HipEngine
    U8 = k.Factored(*[k.Uniform(-5, 5)] * 8)
    model = k.ApproxKernelizedPosterior(U8, k.costs.Rosenbrock(), 1.0)
    N, nt, gens, seed = 2048, 5, 3, 17
    dev = torch.device("cuda", 0)
    e0 = HipEngine(model, N, seed, 0, 2, dev)
    e1 = HipEngine(model, N, seed, 1, 2, dev, half_buffers=e0.half)
    for e in (e0, e1):
        e.init(100)
        e.synchronize()
    for _ in range(gens):
        for half in (0, 1):
            for e in (e0, e1):
                e.half_generation(half, nt)
                e.synchronize()
        for e in (e0, e1):
            e.end_generation(nt)
    pos = torch.cat(e0.half, 0).cpu().numpy()
    o = orc.OracleAIS(model, N, seed=seed).init()
    o.generations_sync(gens, nt, collect=False)
    xo, lpo, llo, _ = o.state()
    assert np.array_equal(pos, xo)
    # log-densities live with the owner: rank r owns rows [r*512, (r+1)*512) of each half
    x0, lp0, ll0, _ = e0.ens.state()
    x1, lp1, ll1, _ = e1.ens.state()
    q = N // 4
    lp = np.concatenate([lp0[:q], lp1[:q], lp0[q:], lp1[q:]])
    ll = np.concatenate([ll0[:q], ll1[:q], ll0[q:], ll1[q:]])
    assert np.array_equal(lp, lpo) and np.array_equal(ll, llo)
    s0, s1 = e0.stats(), e1.stats()
    assert {kk: s0[kk] + s1[kk] for kk in s0} == o.stats()
