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
