"""pfilter (src/smc.jl:275-340): exported by the reference but undocumented and untested
there, so parity is oracle-vs-device (bit-exact) plus sanity on a known posterior."""
import numpy as np
import pytest


def _cases(k):
    N2 = k.Factored(k.Normal(0, 5), k.Normal(0, 5))
    return {
        "gauss": (N2, k.costs.GaussDist([1.0, -0.5]), 400, dict(epstol=0.05)),
        "gauss_q9": (N2, k.costs.GaussDist([1.0, -0.5]), 300, dict(q=0.9, epstol=0.1)),
        "banana_noisy": (N2, k.costs.NoisyBanana(0.0), 500, dict(proposal_width=0.5, max_iters=25)),
        "tiny_N_is_raised": (N2, k.costs.GaussDist([1.0, -0.5]), 5, dict(max_iters=10)),
        "discrete": (k.Factored(k.Normal(1, 0.5), k.DiscreteUniform(1, 10)),
                     k.costs.NoisyQuadDU(5.5), 256, dict(max_iters=15)),
    }


def test_pfilter_oracle(orc, k):
    pri, cost, N, kw = _cases(k)["gauss"]
    r = orc.pfilter(pri, cost, N, seed=1, **kw)
    assert r["eps"] < 0.05 and r["iterations"] > 5
    assert np.all(np.abs(r["P"].mean(0) - [1.0, -0.5]) < 0.02)
    assert r["C"].max() <= r["eps"]            # every particle above ϵ was replaced (:320-322)
    # N*q <= 4*length(prior) => N = ceil((4D+1)/q)   (src/smc.jl:276-279)
    r = orc.pfilter(pri, cost, 5, seed=1, max_iters=3)
    assert r["P"].shape[0] == int(np.ceil(9 / 0.7))


@pytest.mark.gpu
@pytest.mark.parametrize("passes", ["0", "1"], ids=["loop-in-kernel", "launch-per-attempt"])
@pytest.mark.parametrize("name", ["gauss", "gauss_q9", "banana_noisy", "tiny_N_is_raised", "discrete"])
def test_pfilter_bit_exact(k, orc, gpu_ctx, name, passes, monkeypatch):
    """(every bad particle's rejection loop inside one launch -- the default -- or one launch per
    attempt, KABC_PF_PASSES=1: the same draws, the same particles)"""
    monkeypatch.setenv("KABC_PF_PASSES", passes)
    pri, cost, N, kw = _cases(k)[name]
    got = k.pfilter(pri, cost, N, seed=4, return_array=True, **kw)
    ref = orc.pfilter(pri, cost, N, seed=4, **kw)
    assert got.P.shape == ref["P"].shape
    assert np.array_equal(got.P, ref["P"])
    assert np.array_equal(got.C, ref["C"])
    assert got.info["eps"] == ref["eps"] and got.info["iterations"] == ref["iterations"]
    assert got.info["nreps"] == ref["nreps"] and got.info["eff"] == ref["eff"]


@pytest.mark.gpu
@pytest.mark.parametrize("D", [17, 33])
def test_pfilter_beyond_16_parameters_bit_exact(k, orc, gpu_ctx, D):
    """length(prior) > 16: the run-time-dimension instantiation of the attempt / init kernels
    (and the N*q <= 4 length(prior) enlargement rule at these sizes), bit-exact."""
    comps = [k.Normal(0, 2), k.Uniform(-3, 3), k.LogNormal(0.1, 0.4), k.DiscreteUniform(-4, 4)]
    pri = k.Factored(*[comps[j % 4] for j in range(D)])
    cost = k.costs.NormShell(2.0 * np.sqrt(D))
    kw = dict(max_iters=6, proposal_width=0.6, eff_tol=0.0)
    got = k.pfilter(pri, cost, 200, seed=6, return_array=True, **kw)
    ref = orc.pfilter(pri, cost, 200, seed=6, **kw)
    assert got.P.shape == ref["P"].shape and got.P.shape[1] == D
    assert np.array_equal(got.P, ref["P"]) and np.array_equal(got.C, ref["C"])
    assert got.info["eps"] == ref["eps"] and got.info["iterations"] == ref["iterations"]
    assert got.info["nreps"] == ref["nreps"]


@pytest.mark.gpu
@pytest.mark.parametrize("scheme", ["one-workgroup", "launch-per-phase"])
@pytest.mark.parametrize("case", ["defaults_100", "q9_255", "discrete_256", "raised_13", "d17_100", "nothing_bad"])
def test_pfilter_small_ensembles_bit_exact(k, orc, gpu_ctx, monkeypatch, capfd, case, scheme):
    """N <= 256 (the reference's default is 100): the whole loop in ONE launch of one workgroup
    (pf_small_kernel: ε by rank counting, idxok from ballots, the rejection loops with the attempt numbering
    of the other scheme, the stop tests by every thread) against the launches per phase (KABC_PF_SMALL=0)
    and the oracle: same particles, costs, ε, eff, iteration and proposal counts; verbose runs print the
    same lines."""
    monkeypatch.setenv("KABC_PF_SMALL", "1" if scheme == "one-workgroup" else "0")
    N2 = k.Factored(k.Normal(0, 5), k.Normal(0, 5))
    comps = [k.Normal(0, 2), k.Uniform(-3, 3), k.LogNormal(0.1, 0.4), k.DiscreteUniform(-4, 4)]
    cases = {
        "defaults_100": (N2, k.costs.GaussDist([1.0, -0.5]), 100, dict(epstol=0.05)),
        "q9_255": (N2, k.costs.NoisyBanana(0.0), 255, dict(q=0.9, max_iters=20, proposal_width=0.5)),
        "discrete_256": (k.Factored(k.Normal(1, 0.5), k.DiscreteUniform(1, 10)), k.costs.NoisyQuadDU(5.5), 256,
                         dict(max_iters=15)),
        "raised_13": (N2, k.costs.GaussDist([1.0, -0.5]), 5, dict(max_iters=10, verbose=True)),
        "d17_100": (k.Factored(*[comps[j % 4] for j in range(17)]), k.costs.NormShell(2.0 * np.sqrt(17)), 100,
                    dict(max_iters=6, proposal_width=0.6, eff_tol=0.0)),
        # every cost equal: nothing is above ϵ, eff = 0/0 = NaN ends the loop after one iteration (:327-333)
        "nothing_bad": (k.Factored(k.DiscreteUniform(3, 3), k.DiscreteUniform(4, 4)), k.costs.GaussDist([3.0, 4.0]),
                        50, dict()),
    }
    pri, cost, N, kw = cases[case]
    got = k.pfilter(pri, cost, N, seed=4, return_array=True, **kw)
    err = capfd.readouterr().err
    ref = orc.pfilter(pri, cost, N, seed=4, **{a: b for a, b in kw.items() if a != "verbose"})
    assert got.P.shape == ref["P"].shape
    assert np.array_equal(got.P, ref["P"]) and np.array_equal(got.C, ref["C"])
    assert got.info["eps"] == ref["eps"] and got.info["iterations"] == ref["iterations"]
    assert got.info["nreps"] == ref["nreps"]
    assert got.info["eff"] == ref["eff"] or (np.isnan(got.info["eff"]) and np.isnan(ref["eff"]))
    if kw.get("verbose"):
        lines = [ln for ln in err.splitlines() if ln.startswith("(iters, ")]
        assert len(lines) == ref["iterations"] and lines[-1].startswith(f"(iters, ϵ, eff) = ({ref['iterations']}, ")
