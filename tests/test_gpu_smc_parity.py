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
        # README.md:31-49,80-84: `smc(prior, cost)` with its defaults on the 1000-draw simulator
        # (a prepared cost: the pre-pass of every pass, one wavefront per evaluation), and a
        # larger ensemble with retry passes and an odd number of draws
        "readme_defaults": (k.Factored(k.Uniform(1, 3), k.Truncated(k.Normal(0, 0.1), 0, 100)),
                            k.costs.NormalMeanStdSim(1000, 2.0012, 0.0401), dict()),
        "readme_sim_retrys": (k.Factored(k.Uniform(1, 3), k.Truncated(k.Normal(0, 0.1), 0, 100)),
                              k.costs.NormalMeanStdSim(301, 2.0012, 0.0401),
                              dict(nparticles=1500, mcmc_retrys=2, epstol=0.02)),
    }


@pytest.fixture(params=["loop", "kernels", "kernels-select", "default"])
def smc_path(request, monkeypatch):
    """The device drivers of the ε-loop: the persistent cooperative kernel
    (csrc/smc_loop_kernel.hpp, the default from 257 to 65 536 particles), the
    kernel-per-phase path (KABC_SMC_LOOP=0; larger ensembles) -- "kernels": with mcmc_retrys = 0 its
    selection is the speculative one-exchange course (csrc/smc_dsel_kernels.hpp dsel2_*, the select
    kernel for the first two iterations and after a stall; KABC_SMC_SPEC_SELECT=1, the default from
    2^20 particles on), "kernels-select" (KABC_SMC_SPEC_SELECT=0): the select kernel throughout -- and, with nothing selected ("default"), whatever kabc_smc_run picks itself
    -- the one-workgroup kernel (csrc/smc_small_kernel.hpp) up to 256 particles."""
    monkeypatch.delenv("KABC_SMC_SPEC_SELECT", raising=False)
    if request.param == "default":
        monkeypatch.delenv("KABC_SMC_LOOP", raising=False)
    else:
        monkeypatch.setenv("KABC_SMC_LOOP", "1" if request.param == "loop" else "0")
        if request.param != "loop":   # (on its own the course is the default from 2^20 particles on)
            monkeypatch.setenv("KABC_SMC_SPEC_SELECT", "0" if request.param == "kernels-select" else "1")
    return request.param


@pytest.mark.parametrize("name", ["banana", "banana_inf", "dirac", "defaults_du",
                                  "mixture_retrys", "C4_hier16_small", "gauss_d2_minress",
                                  "readme_defaults", "readme_sim_retrys"])
def test_smc_bit_exact(k, orc, gpu_ctx, name, smc_path):
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


@pytest.mark.parametrize("path,blocks", [("loop", None), ("kernels", None), ("kernels", "1"),
                                         ("kernels", "32"), ("kernels-cooperative", "32"), ("spec", None)])
def test_c4_full_size_bit_exact(k, orc, gpu_ctx, monkeypatch, path, blocks):
    """BASELINE.json configs[3] at full size (32 768 particles, D = 16, hierarchical
    Gaussian simulator, ~190 ε-iterations): θ of every particle, ε and the iteration
    log equal the oracle's bit for bit -- on the persistent loop kernel, and on the
    kernel-per-phase path with the select kernel on its default 16 workgroups, on one,
    on 32, and on 32 launched cooperatively."""
    monkeypatch.setenv("KABC_SMC_LOOP", "1" if path == "loop" else "0")
    # ("spec": the speculative one-exchange selection, the default of the kernel-per-phase path; the
    # "kernels" variants are about the select kernel's grid)
    monkeypatch.setenv("KABC_SMC_SPEC_SELECT", "1" if path == "spec" else "0")
    # (the select kernel's grid: an ordinary launch that fits the device, or a cooperative one)
    monkeypatch.setenv("KABC_SMC_COOPERATIVE", "1" if path.endswith("cooperative") else "0")
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
    assert np.array_equal(r.info["alive"], ro["alive"]) and np.array_equal(r.C, ro["C"])
    assert r.info["cost_evals"] == ro["cost_evals"] and r.info["proposals"] == ro["proposals"]


def test_select_on_128_workgroups_bit_exact(k, orc, gpu_ctx, monkeypatch):
    """The kernel-per-phase path beyond the loop kernel's range, its select kernel on 128
    workgroups (what 2^21 particles and more get by default; 140 000 particles are 137 tiles, a
    ragged last slice): everything equals the oracle's bit for bit, and the default choice
    (32 workgroups at this size) gives the same arrays; so does the speculative one-exchange
    selection (the default of this path), whose statistics say how the run was driven."""
    monkeypatch.setenv("KABC_SMC_LOOP", "0")
    monkeypatch.setenv("KABC_SMC_SPEC_SELECT", "0")
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from smc_c4_probe import c4_problem
    prior, cost = c4_problem()
    kw = dict(nparticles=140000, alpha=0.95, epstol=0.25, seed=3)
    monkeypatch.setenv("KABC_SMC_SELECT_BLOCKS", "128")
    r = k.smc(prior, cost, return_array=True, **kw)
    monkeypatch.delenv("KABC_SMC_SELECT_BLOCKS", raising=False)
    r32 = k.smc(prior, cost, return_array=True, **kw)
    monkeypatch.setenv("KABC_SMC_SPEC_SELECT", "1")
    rs = k.smc(prior, cost, return_array=True, **kw)
    ro = orc.smc(prior, cost, **kw)
    assert r.info["iterations"] == ro["iterations"] > 20
    d = rs.info["dist"]
    assert d["batched"] and d["collectives"] == 0 and d["host_looks"] < ro["iterations"] // 2
    assert d["one_exchange_selections"] + d["phase_by_phase_selections"] == ro["iterations"]
    assert d["one_exchange_selections"] >= 0.8 * ro["iterations"], d
    assert not r32.info["dist"]["batched"]
    for rr in (r, r32, rs):
        assert rr.eps == ro["eps"] and np.array_equal(rr.info["theta_all"], ro["theta_all"])
        assert [it["eps"] for it in rr.info["log"]] == [it["eps"] for it in ro["log"]]
        assert [it["ess"] for it in rr.info["log"]] == [it["ess"] for it in ro["log"]]
        assert np.array_equal(rr.info["alive"], ro["alive"]) and np.array_equal(rr.C, ro["C"])


def test_one_exchange_selection_is_the_default_from_2_to_the_20(k, gpu_ctx, monkeypatch):
    """kabc_smc_run beyond 2^20 particles: batches of iterations with the speculative one-exchange selection
    (csrc/smc_dsel_kernels.hpp dsel2_*) -- the same arrays as the select kernel in every iteration, which the
    tests above hold against the oracle at sizes the oracle finishes in seconds."""
    monkeypatch.delenv("KABC_SMC_LOOP", raising=False)
    N2 = k.Factored(k.Normal(0, 5), k.Normal(0, 5))
    cost = k.costs.GaussDist([1.0, -0.5])
    kw = dict(nparticles=(1 << 20) + 777, alpha=0.9, epstol=0.3, seed=6, return_array=True)
    monkeypatch.delenv("KABC_SMC_SPEC_SELECT", raising=False)
    a = k.smc(N2, cost, **kw)
    monkeypatch.setenv("KABC_SMC_SPEC_SELECT", "0")
    b = k.smc(N2, cost, **kw)
    assert a.info["dist"]["batched"] and not b.info["dist"]["batched"]
    assert a.info["dist"]["one_exchange_selections"] >= a.info["iterations"] - 6 > 5
    assert a.eps == b.eps and a.info["log"] == b.info["log"]
    assert np.array_equal(a.info["theta_all"], b.info["theta_all"]) and np.array_equal(a.C, b.C)
    assert np.array_equal(a.info["alive"], b.info["alive"])


def test_loop_kernel_many_sizes(k, orc, gpu_ctx, monkeypatch):
    """The persistent kernel's selection machinery away from the happy path: particle
    counts that are not multiples of 256 or 64, one workgroup, heavy ties (discrete
    costs), α small (large resample ratios), no resampling at all."""
    monkeypatch.setenv("KABC_SMC_LOOP", "1")
    N2 = k.Factored(k.Normal(0, 5), k.Normal(0, 5))
    du = k.Factored(k.DiscreteUniform(-20, 20), k.DiscreteUniform(-20, 20))
    cases = [
        (N2, k.costs.GaussDist([1.0, -0.5]), dict(nparticles=37, alpha=0.9, epstol=0.05)),
        (N2, k.costs.GaussDist([1.0, -0.5]), dict(nparticles=257, alpha=0.5, min_r_ess=0.2, epstol=0.05)),
        (N2, k.costs.GaussDist([1.0, -0.5]), dict(nparticles=5000, alpha=0.3, min_r_ess=0.05, epstol=0.01)),
        (N2, k.costs.GaussDist([1.0, -0.5]), dict(nparticles=65536, alpha=0.95, epstol=0.2)),
        (du, k.costs.GaussDist([3.0, -2.0]), dict(nparticles=9001, alpha=0.8, epstol=0.5)),
        (du, k.costs.GaussDist([3.0, -2.0]), dict(nparticles=1000, alpha=0.99, min_r_ess=0.01, epstol=0.5)),
        (N2, k.costs.NoisyBanana(0.5), dict(nparticles=777, alpha=0.9, epstol=0.01, mcmc_retrys=3,
                                            mcmc_tol=0.3)),
    ]
    for prior, cost, kw in cases:
        for seed in (1, 2):
            got = k.smc(prior, cost, seed=seed, return_array=True, **kw)
            ref = orc.smc(prior, cost, seed=seed, **kw)
            assert got.info["log"] == ref["log"], (kw, seed)
            assert got.eps == ref["eps"] and np.array_equal(got.info["alive"], ref["alive"])
            assert np.array_equal(got.info["theta_all"], ref["theta_all"])
            assert np.array_equal(got.C, ref["C"])
            assert got.info["cost_evals"] == ref["cost_evals"]


def test_small_ensemble_kernel_many_cases(k, orc, gpu_ctx, monkeypatch):
    """The one-workgroup driver (csrc/smc_small_kernel.hpp, N <= 256) away from the happy path:
    the smallest legal ensembles, exactly 256 particles, heavy ties (integer-valued costs), α small
    (large resample ratios), no resampling at all, retry passes, a prepared cost through the ring of
    pre-pass words (README.md:31-49) with more iterations than one ring holds, ∞ costs, and a
    user cost compiled at run time.  KABC_SMC_SMALL=0 gives the same bits on the other drivers."""
    monkeypatch.delenv("KABC_SMC_LOOP", raising=False)
    N2 = k.Factored(k.Normal(0, 5), k.Normal(0, 5))
    du = k.Factored(k.DiscreteUniform(-20, 20), k.DiscreteUniform(-20, 20))
    rd = k.Factored(k.Uniform(1, 3), k.Truncated(k.Normal(0, 0.1), 0, 100))
    H6 = k.Factored(k.Normal(0, 5), k.Uniform(0, 5), *[k.Normal(0, 1)] * 4)
    noisy = k.costs.UserCost("""
KABC_HD double kabc_user_cost(const double* x, int D, const double* params, const double* data,
                              int64_t ndata, kabc_cost_rng_t* rng) {
    double z0, z1;
    kabc_cost_rng_normal2(rng, &z0, &z1);
    return kabc_fabs(x[0] - params[0]) + kabc_fabs(x[1] - params[1]) + 0.01 * kabc_fabs(z0);
}
""", dims=[2], params=[1.0, -0.5], name="l1_noisy")
    orc.register_user_cost(noisy)
    cases = [
        (N2, k.costs.GaussDist([1.0, -0.5]), dict(nparticles=7, alpha=0.95, epstol=0.5)),
        (N2, k.costs.GaussDist([1.0, -0.5]), dict(nparticles=37, alpha=0.9, epstol=0.05)),
        (N2, k.costs.GaussDist([1.0, -0.5]), dict(nparticles=256, alpha=0.5, min_r_ess=0.2, epstol=0.05)),
        (N2, k.costs.GaussDist([1.0, -0.5]), dict(nparticles=255, alpha=0.3, min_r_ess=0.05, epstol=0.02)),
        (du, k.costs.GaussDist([3.0, -2.0]), dict(nparticles=200, alpha=0.8, epstol=0.5)),
        (du, k.costs.GaussDist([3.0, -2.0]), dict(nparticles=129, alpha=0.99, min_r_ess=0.05, epstol=0.5)),
        (N2, k.costs.NoisyBanana(0.5), dict(nparticles=222, alpha=0.9, epstol=0.01, mcmc_retrys=3, mcmc_tol=0.3)),
        (k.Uniform(-10, 10), k.costs.Mixture(0.0), dict(nparticles=100, alpha=0.9, epstol=0.01, mcmc_retrys=50,
                                                       mcmc_tol=0.9)),
        (rd, k.costs.NormalMeanStdSim(1000, 2.0012, 0.0401), dict()),
        (rd, k.costs.NormalMeanStdSim(77, 2.0012, 0.0401), dict(nparticles=250, alpha=0.9)),
        (H6, k.costs.HierGaussSim(np.array([0.9, 1.3, 0.2, 1.1])), dict(nparticles=240, epstol=0.2)),
        (N2, noisy, dict(nparticles=150, epstol=0.05)),
    ]
    for prior, cost, kw in cases:
        for seed in (1, 2):
            ref = orc.smc(prior, cost, seed=seed, **kw)
            for small in ("1", "0"):
                monkeypatch.setenv("KABC_SMC_SMALL", small)
                got = k.smc(prior, cost, seed=seed, return_array=True, **kw)
                assert got.info["log"] == ref["log"], (kw, seed, small)
                assert got.eps == ref["eps"] and np.array_equal(got.info["alive"], ref["alive"])
                assert np.array_equal(got.info["theta_all"], ref["theta_all"])
                assert np.array_equal(got.C, ref["C"]) and np.array_equal(got.P, ref["P"])
                assert got.info["cost_evals"] == ref["cost_evals"] and got.info["proposals"] == ref["proposals"]
    assert ref["iterations"] > 3


_C4_ORACLE = {}


def _oracle_c4(orc, prior, cost, kw):
    if "r" not in _C4_ORACLE:      # 1.7 s of CPU, once for the three cases
        _C4_ORACLE["r"] = orc.smc(prior, cost, **kw)
    return _C4_ORACLE["r"]


def test_kernel_path_beyond_the_loop_kernel(k, orc, gpu_ctx, monkeypatch):
    """More than 65 536 particles: the alive mask no longer fits the loop kernel's LDS, the
    kernel-per-phase path (cooperative 32-workgroup select) takes over on its own."""
    monkeypatch.delenv("KABC_SMC_LOOP", raising=False)
    pri = k.Factored(k.Normal(0, 5), k.Normal(0, 5))
    kw = dict(nparticles=100001, alpha=0.9, epstol=0.3, seed=6)
    got = k.smc(pri, k.costs.GaussDist([1.0, -0.5]), return_array=True, **kw)
    ref = orc.smc(pri, k.costs.GaussDist([1.0, -0.5]), **kw)
    assert got.info["log"] == ref["log"] and got.eps == ref["eps"]
    assert np.array_equal(got.info["theta_all"], ref["theta_all"])
    assert np.array_equal(got.info["alive"], ref["alive"])


def test_loop_kernel_giving_up_repeats_the_run_on_the_kernel_path(k, orc, gpu_ctx, monkeypatch):
    """The persistent loop kernel has an internal capacity (candidates of one histogram bin).
    When it is exceeded the run is not an error of the user's problem: kabc_smc_run repeats it on
    the kernel-per-phase path inside the same call (the draws are counter-based: same run)."""
    prior = k.Factored(k.Normal(0, 2), k.Normal(0, 2))
    cost = k.costs.GaussDist([0.5, -0.25])
    kw = dict(nparticles=3000, epstol=0.05, seed=8)
    ref = orc.smc(prior, cost, **kw)
    monkeypatch.setenv("KABC_SMC_LOOP_GIVE_UP", "1")
    r = k.smc(prior, cost, return_array=True, **kw)
    assert r.eps == ref["eps"] and np.array_equal(r.info["theta_all"], ref["theta_all"])
    assert r.info["iterations"] == ref["iterations"]


@pytest.mark.parametrize("path", ["loop", "kernels"])
@pytest.mark.parametrize("D", [9, 12, 16])
def test_general_priors_beyond_eight_parameters(k, orc, gpu_ctx, monkeypatch, path, D):
    """GENERAL prior class with more than 8 components: the per-component log-density is an
    out-of-line function there (csrc/kabc_device.hpp kGeneralInlineD) -- the only call inside the
    propose/accept pass of the smc kernels.  Both drivers, all nine families in the prior."""
    monkeypatch.setenv("KABC_SMC_LOOP", "1" if path == "loop" else "0")
    fam = [k.Gamma(2.5, 0.7), k.LogNormal(0.3, 0.6), k.Beta(2, 3), k.Normal(0, 1), k.Uniform(-2, 3),
           k.Exponential(1.5), k.TruncatedNormal(0, 1, -1, 2), k.DiscreteUniform(0, 6),
           k.NegativeBinomial(4.0, 0.4)]
    prior = k.Factored(*[fam[i % 9] for i in range(D)])
    cost = k.costs.GaussDist(np.linspace(0.2, 1.5, D))
    kw = dict(nparticles=2000, alpha=0.9, epstol=2.0, mcmc_retrys=1, seed=11)
    got = k.smc(prior, cost, return_array=True, **kw)
    ref = orc.smc(prior, cost, **kw)
    assert got.info["log"] == ref["log"] and got.eps == ref["eps"]
    assert np.array_equal(got.info["theta_all"], ref["theta_all"])
    assert np.array_equal(got.info["alive"], ref["alive"]) and np.array_equal(got.C, ref["C"])


def test_select_launch_that_times_out_is_repeated_cooperatively(k, orc, gpu_ctx, monkeypatch):
    """The select kernel's device-wide barriers need its workgroups co-resident.  It is launched as an
    ordinary grid a quarter of the device's capacity at most; if that grid does not become resident
    within 0.2 s (several large runs, another tenant holding the CUs) the kernel gives up and the SAME
    run -- every draw is counter-based -- is repeated with cooperative launches, whose co-residency
    the runtime asserts.  Forced here two ways: the test hook, and a barrier time-out of 10 ns that
    a real 32-workgroup grid cannot meet.  (KABC_SMC_SPEC_SELECT=0: the select kernel in every iteration.)"""
    monkeypatch.setenv("KABC_SMC_LOOP", "0")
    monkeypatch.setenv("KABC_SMC_SPEC_SELECT", "0")
    N2 = k.Factored(k.Normal(0, 5), k.Normal(0, 5))
    cost = k.costs.GaussDist([1.0, -0.5])
    kw = dict(nparticles=140000, alpha=0.9, epstol=0.2, seed=3)
    ref = k.smc(N2, cost, return_array=True, **kw)
    ro = orc.smc(N2, cost, **kw)
    assert ref.eps == ro["eps"] and np.array_equal(ref.info["theta_all"], ro["theta_all"])
    for env in ({"KABC_SMC_SELECT_TIME_OUT": "1"}, {"KABC_SMC_BARRIER_TIMEOUT_MS": "0.0001"}):
        for kk, v in env.items():
            monkeypatch.setenv(kk, v)
        r = k.smc(N2, cost, return_array=True, **kw)
        assert r.eps == ref.eps and r.info["iterations"] == ref.info["iterations"]
        assert np.array_equal(r.info["theta_all"], ref.info["theta_all"]) and np.array_equal(r.C, ref.C)
        gp = k.pfilter(N2, cost, 140000, seed=4, return_array=True, max_iters=4)
        for kk in env:
            monkeypatch.delenv(kk)
        gp0 = k.pfilter(N2, cost, 140000, seed=4, return_array=True, max_iters=4)
        assert np.array_equal(gp.P, gp0.P) and np.array_equal(gp.C, gp0.C)


NAN_SRC = """
KABC_HD double kabc_user_cost(const double* x, int D, const double* params,
                              const double* data, int64_t ndata, kabc_cost_rng_t* rng) {
    const double r = kabc_sqrt(x[0] * x[0] + x[1] * x[1]);
    /* 0/0 once a particle comes within params[0] of the origin; params[0] < 0: in the initial draw already */
    return (r < params[0] || params[0] < 0.0) ? (r - r) / (r - r) : r;
}
"""


@pytest.mark.parametrize("where", ["init", "later"])
@pytest.mark.parametrize("driver", ["small", "loop", "kernels", "cost_loop", "particles"])
def test_nan_cost_is_the_references_quantile_error(k, orc, gpu_ctx, monkeypatch, driver, where):
    """`quantile` of costs with a NaN among the alive ones throws (Statistics: "quantiles are undefined in
    presence of NaNs"; src/smc.jl:134): the same message from every driver of the ε-loop and from both
    sharding modes, in the initial population or iterations later -- as the oracle's restatement"""
    cost = k.costs.UserCost(NAN_SRC, dims=[2], params=[-1.0 if where == "init" else 0.4], name="nan_" + where)
    orc.register_user_cost(cost)
    prior = k.Factored(k.Normal(0, 3), k.Normal(0, 3))
    N = 200 if driver == "small" else 3000
    kw = dict(nparticles=N, alpha=0.9, epstol=0.01, seed=3)
    with pytest.raises(Exception) as eo:
        orc.smc(prior, cost, **kw)
    assert "quantiles are undefined in presence of NaNs" in str(eo.value)
    if driver in ("loop", "kernels"):
        monkeypatch.setenv("KABC_SMC_LOOP", "1" if driver == "loop" else "0")
    if driver in ("cost_loop", "particles"):
        import threading
        comms = k.comm.init_all([0, 0, 0], "p2p")
        errs = []

        def run(c):
            try:
                k.smc(prior, cost, comm=c, shard=driver, return_array=True, **kw)
                errs.append("no error")
            except k.KabcError as e:
                errs.append(str(e))

        th = [threading.Thread(target=run, args=(c,)) for c in comms]
        for t in th:
            t.start()
        for t in th:
            t.join(timeout=300)
        for c in comms:
            c.close()
        assert errs == ["quantiles are undefined in presence of NaNs"] * 3
    else:
        with pytest.raises(k.KabcError) as e:
            k.smc(prior, cost, return_array=True, **kw)
        assert str(e.value) == "quantiles are undefined in presence of NaNs"
