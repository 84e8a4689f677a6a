"""Independent chains as a grid dimension (kabc_ais_create_batch): what
sample(model, AIS(N), MCMCThreads(), Ns, Nc) of src/KissABC.jl:96-104,108 maps to.
Every chain of a batch handle must be bit-identical to a single-chain handle with its
seed, and the batch must be much faster than running the chains one after the other
(the reference's own usage is tiny ensembles: AIS(10)...AIS(500), test/runtests.jl)."""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _models(k):
    return {
        "dirac_ais12": (k.ApproxKernelizedPosterior(k.Normal(1, 0.2), k.costs.DiracSq(1.5), 0.001), 12),
        "gauss_d2_odd": (k.ApproxKernelizedPosterior(k.Factored(k.Normal(0, 5), k.Normal(0, 5)),
                                                     k.costs.GaussDist([1.0, -0.5]), 0.1), 333),
        "rosen_d8": (k.ApproxPosterior(k.Factored(*[k.Uniform(-5, 5)] * 8), k.costs.Rosenbrock(), 30.0),
                     4096),
        "du_general": (k.ApproxPosterior(k.Factored(k.Normal(1, 0.5), k.DiscreteUniform(1, 10)),
                                         k.costs.NoisyQuadDU(5.5), 0.5), 100),
    }


@pytest.mark.parametrize("name", ["dirac_ais12", "gauss_d2_odd", "rosen_d8", "du_general"])
def test_batch_equals_single_chains(k, gpu_ctx, name):
    model, N = _models(k)[name]
    seeds = [3, 2 ** 40 + 17, 5, 99, 12345678901234, 7, 1]
    nt, gens = 5, 4
    b = k.AisEnsemble(model, N, seeds=seeds).init()
    b.advance(2, nt)
    trb = b.advance(gens, nt, collect=True)                  # [gens][chains][N][D]
    xb, lpb, llb, tb = b.state()
    eb = b.ensemble()
    tot = {"proposals": 0, "cost_evals": 0, "accepted": 0}
    for c, seed in enumerate(seeds):
        e = k.AisEnsemble(model, N, seed=seed).init()
        e.advance(2, nt)
        tr = e.advance(gens, nt, collect=True)
        x, lp, ll, t = e.state()
        assert np.array_equal(trb[:, c], tr), (name, c)
        assert np.array_equal(xb[c], x) and np.array_equal(lpb[c], lp) and np.array_equal(llb[c], ll)
        assert np.array_equal(eb[c], e.ensemble()) and tb == t
        for kk, v in e.stats().items():
            tot[kk] += v
        e.close()
    assert b.stats() == tot
    # state round trip through the strided copies
    b.set_state(xb, lpb, llb, tb)
    x2, lp2, ll2, _ = b.state()
    assert np.array_equal(x2, xb) and np.array_equal(lp2, lpb) and np.array_equal(ll2, llb)
    b.close()


def test_mcmcthreads_sample_is_the_chains_stacked(k, gpu_ctx):
    # test/runtests.jl:88-104: 50 chains x 100 samples x AIS(12) -> sim(res) ≈ 1.5
    from kissabc_jl_amd.api import chain_seeds
    model = k.ApproxKernelizedPosterior(k.Normal(1, 0.2), k.costs.DiracSq(1.5), 0.001)
    kw = dict(ntransitions=10, discard_initial=240)
    got = k.sample(model, k.AIS(12), k.MCMCThreads(), 100, 50, seed=4, return_array=True, **kw)
    assert got.shape == (5000, 1)
    for c, s in list(enumerate(chain_seeds(4, 50)))[::7]:
        one = k.sample(model, k.AIS(12), 100, seed=s, return_array=True, **kw)
        assert np.array_equal(got[c * 100:(c + 1) * 100], one)
    sim = got[:, 0] ** 2 + 1.0
    assert abs(sim.mean() - 1.5) < 2.0 * sim.std()      # MonteCarloMeasurements `≈`


def test_batch_is_faster_than_chain_after_chain(k, gpu_ctx):
    from kissabc_jl_amd.api import chain_seeds
    model = k.ApproxKernelizedPosterior(k.Normal(1, 0.2), k.costs.DiracSq(1.5), 0.001)
    kw = dict(ntransitions=100, discard_initial=1200)
    k.sample(model, k.AIS(12), k.MCMCThreads(), 120, 4, seed=1, **kw)      # warm-up
    t0 = time.perf_counter()
    k.sample(model, k.AIS(12), k.MCMCThreads(), 120, 50, seed=1, return_array=True, **kw)
    t_batch = time.perf_counter() - t0
    t0 = time.perf_counter()
    for s in chain_seeds(1, 50):
        k.sample(model, k.AIS(12), 120, seed=s, return_array=True, **kw)
    t_seq = time.perf_counter() - t0
    print(f"50 x AIS(12): batch {t_batch * 1e3:.1f} ms, chain after chain {t_seq * 1e3:.1f} ms")
    assert t_seq > 10.0 * t_batch


def test_batch_argument_checks(k, gpu_ctx):
    model, N = _models(k)["gauss_d2_odd"]
    with pytest.raises(k.KabcError):
        k.AisEnsemble(model, N, seeds=[])
    b = k.AisEnsemble(model, N, seeds=[1, 2])
    with pytest.raises(k.KabcError, match="single-chain"):
        b.set_debug(3)
    b.close()
