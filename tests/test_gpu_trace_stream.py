"""kabc_ais_advance(out_samples): the double-buffered trace stream (device chunks
drained on a copy stream while the next chunk computes) must deliver, generation by
generation, exactly the states a generation-at-a-time run passes through --
pinned and pageable destinations, one chunk and many."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _model(k, D):
    return k.ApproxKernelizedPosterior(k.Factored(*[k.Uniform(-5, 5)] * D),
                                       k.costs.Rosenbrock(), 1.0)


@pytest.mark.parametrize("N,D,gens", [(65536, 8, 24),   # 96 MiB trace: 4 MiB chunks, 24 of them
                                      (1000, 3, 7)])    # one small chunk, ragged batches
def test_trace_matches_stepwise_states(N, D, gens, monkeypatch):
    import kissabc_jl_amd as k
    from kissabc_jl_amd.api import AisEnsemble
    model = _model(k, D)
    step = AisEnsemble(model, N, seed=5)
    step.init(100)
    want = np.empty((gens, N, D))
    for g in range(gens):
        step.advance(1, 3)
        want[g] = step.state()[0]
    step.close()
    for pinned in ("1", "0"):
        monkeypatch.setenv("KABC_PINNED_TRACE", pinned)
        ens = AisEnsemble(model, N, seed=5)
        ens.init(100)
        got = ens.advance(gens, 3, collect=True)
        assert np.array_equal(got, want), f"pinned={pinned}"
        # a second call reuses the device chunks and continues the chain
        more = ens.advance(2, 3, collect=True)
        assert np.array_equal(more[-1], ens.state()[0])
        ens.close()


def test_sample_uses_stream_and_discards(monkeypatch):
    import kissabc_jl_amd as k
    model = _model(k, 2)
    a = k.sample(model, k.AIS(512), 512 * 5, ntransitions=4, discard_initial=512 * 3, seed=9,
                 return_array=True)
    monkeypatch.setenv("KABC_PINNED_TRACE", "0")
    b = k.sample(model, k.AIS(512), 512 * 5, ntransitions=4, discard_initial=512 * 3, seed=9,
                 return_array=True)
    assert a.shape == (512 * 5, 2) and np.array_equal(a, b)


def test_page_locked_blocks_go_back_to_a_free_list(gpu_ctx, monkeypatch):
    """Result arrays of 1 MiB and more sit in page-locked memory (kabc_host_alloc); when the last
    view of one dies its block serves the next array of the same size instead of being pinned
    again (pinning 268 MB costs more than ten copies into it), up to KABC_PINNED_CACHE_MB."""
    import gc

    from kissabc_jl_amd import _lib
    a = _lib.result_empty((70000, 2))            # 1.07 MiB: page-locked
    small = _lib.result_empty((1000, 2))         # ordinary memory
    addr = a.ctypes.data
    a[:] = 3.0
    del a
    gc.collect()
    b = _lib.result_empty((70000, 2))
    assert b.ctypes.data == addr                 # the same block, not a new pin
    c = _lib.result_empty((70000, 2))
    assert c.ctypes.data != addr                 # the free list is empty now: a second block
    assert small.ctypes.data not in (addr, c.ctypes.data)
    del b, c
    gc.collect()
    monkeypatch.setenv("KABC_PINNED_CACHE_MB", "0")   # nothing is kept beyond the cap
    d = _lib.result_empty((70001, 2))
    daddr = d.ctypes.data
    del d
    gc.collect()
    with _lib._pinned_lock:
        assert daddr not in _lib._pinned_free.get(70001 * 2 * 8, [])


def test_large_smc_results_do_not_depend_on_where_they_land(gpu_ctx, monkeypatch):
    """smc beyond 1 MiB of particles: page-locked result arrays (default) and ordinary ones hold
    the same run; two runs in a row reuse the arrays' blocks and the context's device buffers."""
    import kissabc_jl_amd as k
    pri = k.Factored(k.Normal(0, 5), k.Normal(0, 5))
    cost = k.costs.GaussDist([1.0, -0.5])
    kw = dict(nparticles=70000, alpha=0.9, epstol=0.5, seed=4)
    r1 = k.smc(pri, cost, return_array=True, **kw)
    t1, c1, e1 = r1.P.copy(), r1.C.copy(), r1.eps
    del r1
    r2 = k.smc(pri, cost, return_array=True, **kw)
    monkeypatch.setenv("KABC_PINNED_TRACE", "0")
    r3 = k.smc(pri, cost, return_array=True, **kw)
    for r in (r2, r3):
        assert r.eps == e1 and np.array_equal(r.P, t1) and np.array_equal(r.C, c1)
    assert set(r2.info["host_ms"]) == {"kabc_smc_run", "python_after"}
