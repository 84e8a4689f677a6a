"""User DeviceCosts compiled at run time (kabc_register_cost_plugin): the
device-path counterpart of "cost is an arbitrary closure" (src/types.jl:42,55)."""
import os

import numpy as np
import pytest

ROSEN_SRC = """
KABC_HD double kabc_user_cost(const double* x, int D, const double* params,
                              const double* data, int64_t ndata, kabc_cost_rng_t* rng) {
    double s = 0.0;
    for (int k = 0; k + 1 < D; ++k) {
        double a = x[k + 1] - x[k] * x[k];
        double b = 1.0 - x[k];
        s += 100.0 * a * a + b * b;
    }
    return kabc_sqrt(s);
}
"""

# a stochastic simulator nobody built in: AR(1)-ish summary with observation noise
SIM_SRC = """
KABC_HD double kabc_user_cost(const double* x, int D, const double* params,
                              const double* data, int64_t ndata, kabc_cost_rng_t* rng) {
    double y = params[0], acc = 0.0;
    for (int64_t t = 0; t < ndata; t += 2) {
        double z0, z1;
        kabc_cost_rng_normal2(rng, &z0, &z1);
        y = x[0] * y + x[1] * z0;
        acc += kabc_fabs(y - data[t]);
        if (t + 1 < ndata) {
            y = x[0] * y + x[1] * z1;
            acc += kabc_fabs(y - data[t + 1]);
        }
    }
    return acc / (double)ndata + kabc_fabs(x[2]) * 0.01;
}
"""


def test_plugin_source_is_generated(k):
    from kissabc_jl_amd import costs
    txt = costs._plugin_source(ROSEN_SRC, [4, 8])
    assert "#define KABC_USER_DIM_OK(D) ((D) == 4 || (D) == 8)" in txt
    assert txt.strip().endswith('#include "user_plugin.inc"')
    assert os.path.exists(os.path.join(costs._CSRC, "user_plugin.inc"))


def test_oracle_runs_a_user_cost(k, orc):
    """CPU leg: the snippet compiled with gcc gives exactly the built-in's numbers."""
    from kissabc_jl_amd.costs import DeviceCost
    c = DeviceCost(100 + 63, name="user")      # id only used by the oracle registry here
    c.source = ROSEN_SRC
    orc.register_user_cost(c)
    U = k.Factored(*[k.Uniform(-5, 5)] * 8)
    a = orc.OracleAIS(k.ApproxKernelizedPosterior(U, c, 1.0), 256, seed=3).init()
    b = orc.OracleAIS(k.ApproxKernelizedPosterior(U, k.costs.Rosenbrock(), 1.0), 256, seed=3).init()
    assert np.array_equal(a.generations_sync(3, 4), b.generations_sync(3, 4))


@pytest.mark.gpu
def test_user_cost_equals_builtin_and_oracle(k, orc, gpu_ctx):
    """Same formula as the built-in Rosenbrock => bit-identical trajectories, AIS and SMC."""
    user = k.costs.UserCost(ROSEN_SRC, dims=[2, 8], posteriors=["kernelized"])
    assert user.id >= 100
    U = k.Factored(*[k.Uniform(-5, 5)] * 8)
    N, nt, gens = 2048, 6, 3
    got = k.AisEnsemble(k.ApproxKernelizedPosterior(U, user, 1.0), N, seed=4).init().advance(
        gens, nt, collect=True)
    ref = k.AisEnsemble(k.ApproxKernelizedPosterior(U, k.costs.Rosenbrock(), 1.0), N,
                        seed=4).init().advance(gens, nt, collect=True)
    assert np.array_equal(got, ref)
    orc.register_user_cost(user)
    assert np.array_equal(
        got, orc.OracleAIS(k.ApproxKernelizedPosterior(U, user, 1.0), N, seed=4).init()
        .generations_sync(gens, nt))
    N2 = k.Factored(k.Normal(0, 5), k.Normal(0, 5))
    a = k.smc(N2, user, nparticles=1000, alpha=0.9, epstol=0.05, seed=2, return_array=True)
    b = k.smc(N2, k.costs.Rosenbrock(), nparticles=1000, alpha=0.9, epstol=0.05, seed=2,
              return_array=True)
    assert np.array_equal(a.P, b.P) and a.eps == b.eps
    c = k.ABCDE(N2, user, 0.05, nparticles=200, generations=30, seed=3, return_array=True)
    d = k.ABCDE(N2, k.costs.Rosenbrock(), 0.05, nparticles=200, generations=30, seed=3,
                return_array=True)
    assert np.array_equal(c.P, d.P) and np.array_equal(c.C, d.C)
    with pytest.raises(k.KabcError):          # D = 5 was not built
        k.AisEnsemble(k.ApproxKernelizedPosterior(k.Factored(*[k.Uniform(-5, 5)] * 5), user, 1.0),
                      64)


@pytest.mark.gpu
def test_user_stochastic_simulator_bit_exact_vs_oracle(k, orc, gpu_ctx):
    rng = np.random.default_rng(5)
    data = np.cumsum(rng.normal(size=24)) * 0.3
    user = k.costs.UserCost(SIM_SRC, dims=[3], params=[0.5], data=data, name="ar1",
                            posteriors=["kernelized"])
    orc.register_user_cost(user)
    prior = k.Factored(k.Uniform(-1, 1), k.Uniform(0, 2), k.Normal(0, 1))
    model = k.ApproxKernelizedPosterior(prior, user, 0.2)
    got = k.AisEnsemble(model, 500, seed=8).init().advance(4, 5, collect=True)
    assert np.array_equal(got, orc.OracleAIS(model, 500, seed=8).init().generations_sync(4, 5))
    r = k.smc(prior, user, nparticles=800, alpha=0.9, epstol=0.3, seed=8, return_array=True)
    o = orc.smc(prior, user, nparticles=800, alpha=0.9, epstol=0.3, seed=8)
    assert np.array_equal(r.info["theta_all"], o["theta_all"]) and r.eps == o["eps"]


# ---- CommonLogDensity (src/types.jl:105-128) with user log-densities ---------------
BANANA_LPI = """
KABC_HD double kabc_user_cost(const double* x, int D, const double* params,
                              const double* data, int64_t ndata, kabc_cost_rng_t* rng) {
    double a = x[0] - x[1] * x[1], b = x[1] - 1.0;      /* test/runtests.jl:204 */
    return -100.0 * a * a - b * b;
}
"""
DISC_LPI = """
KABC_HD double kabc_user_cost(const double* x, int D, const double* params,
                              const double* data, int64_t ndata, kabc_cost_rng_t* rng) {
    double s = x[0] * x[0] + x[1] * x[1];                /* test/runtests.jl:225 */
    return (s <= params[0]) ? 0.0 : -KABC_INF;
}
"""


def test_common_log_density_on_oracle(k, orc):
    from kissabc_jl_amd.costs import DeviceCost
    lpi = DeviceCost(100 + 62, name="banana")
    lpi.source = BANANA_LPI
    orc.register_user_cost(lpi)
    D = k.CommonLogDensity(2, k.Factored(k.Normal(0, 1), k.Normal(0, 1)), lpi)
    assert len(D) == 2                                    # test/runtests.jl:206
    o = orc.OracleAIS(D, 50, seed=1).init()
    x, lp, ll, _ = o.state()
    assert np.all(lp == 0.0)
    a = x[:, 0] - x[:, 1] ** 2
    assert np.allclose(ll, -100 * a * a - (x[:, 1] - 1) ** 2, rtol=1e-14)
    o.generations_sync(40, 100, collect=False)
    s = o.generations_sync(20, 100).reshape(-1, 2)
    lp_s = -100 * (s[:, 0] - s[:, 1] ** 2) ** 2 - (s[:, 1] - 1) ** 2
    assert np.quantile(lp_s, 0.97) > -0.69                # test/runtests.jl:217
    # documented output of the docstring example (src/KissABC.jl:151): 1.43 ± 1.4, 0.99 ± 0.67
    assert abs(s[:, 0].mean() - 1.43) < 0.35 and abs(s[:, 1].mean() - 0.99) < 0.2


@pytest.mark.gpu
def test_common_log_density_gpu_parity_and_reference_tests(k, orc, gpu_ctx):
    banana = k.costs.UserCost(BANANA_LPI, dims=[2], name="banana", posteriors=["common"])
    orc.register_user_cost(banana)
    D = k.CommonLogDensity(2, k.Factored(k.Normal(0, 1), k.Normal(0, 1)), banana)
    got = k.AisEnsemble(D, 50, seed=1).init().advance(5, 20, collect=True)
    assert np.array_equal(got, orc.OracleAIS(D, 50, seed=1).init().generations_sync(5, 20))
    res = k.sample(D, k.AIS(50), 1000, ntransitions=100, discard_initial=2000, seed=1,
                   return_array=True)
    lp = -100 * (res[:, 0] - res[:, 1] ** 2) ** 2 - (res[:, 1] - 1) ** 2
    assert np.quantile(lp, 0.97) > -0.69                  # test/runtests.jl:217
    # "Handling of ∞ costs", test/runtests.jl:221-238
    disc = k.costs.UserCost(DISC_LPI, dims=[2], params=[1.0], name="disc", posteriors=["common"])
    init = k.Factored(k.Uniform(-1, 1), k.Uniform(0, 1))
    res = k.sample(k.CommonLogDensity(2, init, disc), k.AIS(50), 1000, ntransitions=100,
                   discard_initial=5000, seed=2, return_array=True)
    assert np.all((res ** 2).sum(1) <= 1.0)
    never = k.costs.UserCost(DISC_LPI, dims=[2], params=[-1.0], name="never", posteriors=["common"])   # lπ = -Inf
    with pytest.raises(k.KabcError) as e:
        k.sample(k.CommonLogDensity(2, init, never), k.AIS(50), 10)
    assert "Prior leads to ∞ costs too often" in str(e.value)


# ---- prepared user cost (include/kabc_costs.h "prepared costs") -----------------------
PREP_SRC = r'''
#define KABC_USER_AUX_WORDS 2
KABC_HD void kabc_user_cost_prepare(const double* params, const double* data, int64_t ndata,
                                    kabc_cost_rng_t* rng, double* aux) {
    int n = (int)params[0];
    double sz = 0.0, szz = 0.0;
    for (int j = 0; j < n; j += 2) {
        double z0, z1;
        kabc_cost_rng_normal2(rng, &z0, &z1);
        sz += z0;  szz += z0 * z0;
        if (j + 1 < n) { sz += z1;  szz += z1 * z1; }
    }
    aux[0] = sz;  aux[1] = szz;
}
KABC_HD double kabc_user_cost(const double* x, int D, const double* params, const double* data,
                              int64_t ndata, kabc_cost_rng_t* rng) {
    double aux[2];
    if (rng->aux) { aux[0] = rng->aux[0];  aux[1] = rng->aux[rng->aux_stride]; }
    else kabc_user_cost_prepare(params, data, ndata, rng, aux);
    double dn = (double)(int)params[0];
    double mz = aux[0] / dn;
    double vz = (aux[1] - dn * mz * mz) / (dn - 1.0);
    if (vz < 0.0) vz = 0.0;
    double a = x[0] + x[1] * mz - params[1];
    double b = 50.0 * (kabc_fabs(x[1]) * kabc_sqrt(vz) - params[2]);
    return kabc_sqrt(a * a + b * b);
}
'''


@pytest.mark.gpu
def test_prepared_user_cost_equals_builtin_and_oracle(k, orc, gpu_ctx):
    """A user simulator with a prepare step (run by the AIS producer waves) gives the bits of
    the oracle, which evaluates the same snippet without any preparation.  (The built-in
    normal_meanstd_sim sums its draws in the 64-slice order of include/kabc_costs.h -- a
    different rounding of the same quantity: the two agree to ~1e-12, not to the bit.)"""
    prior = k.Factored(k.Uniform(1, 3), k.Truncated(k.Normal(0, 0.1), 0, 100))
    user = k.costs.UserCost(PREP_SRC, dims=[2], params=[200, 2.0, 0.04], name="prep",
                            posteriors=["kernelized"])
    orc.register_user_cost(user)
    N, gens, nt = 300, 3, 4
    mu = k.ApproxKernelizedPosterior(prior, user, 0.005)
    mb = k.ApproxKernelizedPosterior(prior, k.costs.NormalMeanStdSim(200, 2.0, 0.04), 0.005)
    got = k.AisEnsemble(mu, N, seed=2).init().advance(gens, nt, collect=True)
    ref = k.AisEnsemble(mb, N, seed=2).init().advance(gens, nt, collect=True)
    assert got.shape == ref.shape and np.isfinite(ref).all()
    assert np.array_equal(got, orc.OracleAIS(mu, N, seed=2).init().generations_sync(gens, nt))
    assert np.array_equal(ref, orc.OracleAIS(mb, N, seed=2).init().generations_sync(gens, nt))


# ---- kabc_compile_cost_plugin: the snippet compiled in process by hipRTC ---------------------
def test_hiprtc_checks_the_snippet_at_registration(k):
    """No GPU needed: hipRTC is a compiler.  A broken snippet comes back at once with the
    compiler's message; a good one gets an id and nothing else is compiled yet."""
    import time
    bad = ROSEN_SRC.replace("kabc_sqrt(s)", "kabc_sqrt(s) + undeclared_thing")
    with pytest.raises(k.KabcError, match="undeclared_thing"):
        k.costs.UserCost(bad, dims=[2], posteriors=["kernelized"])
    t0 = time.perf_counter()
    c = k.costs.UserCost(ROSEN_SRC + "// registration test\n", dims=[2, 8], posteriors=["kernelized"])
    assert c.id >= 100 and time.perf_counter() - t0 < 5.0
    import ctypes as C
    from kissabc_jl_amd import _lib
    out = C.c_int32()
    # (longer vectors run on the run-time-dimension kernels of the same unit, up to 256)
    _lib.check(_lib.load().kabc_compile_cost_plugin(ROSEN_SRC.encode(), (C.c_int32 * 1)(40), 1, 0, C.byref(out)))
    with pytest.raises(k.KabcError, match="1..256"):
        _lib.check(_lib.load().kabc_compile_cost_plugin(ROSEN_SRC.encode(), (C.c_int32 * 1)(257), 1, 0,
                                                        C.byref(out)))


def test_hiprtc_compiles_every_kernel_family(k):
    """Each family of a user cost compiles under hipRTC from the library's own headers (on a box
    without a GPU the code object is produced and only its load fails; with one it loads)."""
    import ctypes as C
    from kissabc_jl_amd import _lib
    lib = _lib.load()
    c = k.costs.UserCost(SIM_SRC + "// family test\n", dims=[3], params=[0.5], data=[0.0] * 8, name="ar1")
    for family, variant in [(0, 1), (1, 0), (2, 1), (3, 1), (4, 0), (5, 0), (6, 0), (7, 0)]:
        st = lib.kabc_plugin_precompile(c.id, family, 3, variant)
        msg = lib.kabc_last_error().decode()
        assert st == 0 or "compiled (" in msg, (family, variant, msg[:2000])
    assert lib.kabc_plugin_precompile(c.id, 0, 5, 1) != 0      # D = 5 was not listed


@pytest.mark.gpu
def test_hiprtc_and_hipcc_plugins_agree(k, orc, gpu_ctx, monkeypatch):
    """The two forms of a user cost -- hipRTC in process (default) and the plugin .so built by
    hipcc -- run the same kernels: identical trajectories (AIS and smc), and a first AIS use of
    the hipRTC form within a few seconds."""
    import time
    U = k.Factored(*[k.Uniform(-5, 5)] * 8)
    src = ROSEN_SRC + "// rtc-vs-hipcc\n"
    rtc = k.costs.UserCost(src, dims=[8], posteriors=["kernelized"])
    t0 = time.perf_counter()
    a = k.AisEnsemble(k.ApproxKernelizedPosterior(U, rtc, 1.0), 700, seed=9).init().advance(3, 5, collect=True)
    first_use = time.perf_counter() - t0
    monkeypatch.setenv("KABC_USER_PLUGIN", "hipcc")
    so = k.costs.UserCost(src, dims=[8], posteriors=["kernelized"])
    assert so.id != rtc.id
    b = k.AisEnsemble(k.ApproxKernelizedPosterior(U, so, 1.0), 700, seed=9).init().advance(3, 5, collect=True)
    assert np.array_equal(a, b)
    assert np.array_equal(a, k.AisEnsemble(k.ApproxKernelizedPosterior(U, k.costs.Rosenbrock(), 1.0), 700,
                                           seed=9).init().advance(3, 5, collect=True))
    ra = k.smc(U, rtc, nparticles=3000, epstol=2.0, seed=4, return_array=True)
    rb = k.smc(U, so, nparticles=3000, epstol=2.0, seed=4, return_array=True)
    assert ra.eps == rb.eps and np.array_equal(ra.info["theta_all"], rb.info["theta_all"])
    assert first_use < 10.0, first_use                          # two kernels through hipRTC + the run
    print(f"first AIS use of a hipRTC cost: {first_use:.2f} s")


# a log-density that brings its own sample_init (src/types.jl:105-113: `rng -> sample`): a ring,
# initial walkers drawn ON the ring (something no product of univariate priors can express)
RING_SRC = """
#define KABC_USER_SAMPLE_INIT 1
KABC_HD void kabc_user_sample_init(double* x, int D, const double* params,
                                   const double* data, int64_t ndata, kabc_cost_rng_t* rng) {
    double u0, u1, z0, z1;
    kabc_cost_rng_uniform2(rng, &u0, &u1);
    kabc_cost_rng_normal2(rng, &z0, &z1);
    double s, c;
    kabc_sincos2pi(u0, &s, &c);
    const double r = params[0] + 0.05 * z0;
    x[0] = r * c;
    x[1] = r * s;
}
KABC_HD double kabc_user_cost(const double* x, int D, const double* params,
                              const double* data, int64_t ndata, kabc_cost_rng_t* rng) {
    const double r = kabc_sqrt(x[0] * x[0] + x[1] * x[1]);
    const double d = (r - params[0]) / params[1];
    return -0.5 * d * d;      /* lπ: a ring of radius params[0], width params[1] */
}
"""


def test_user_sample_init_on_oracle(k, orc):
    ring = k.costs.UserCost(RING_SRC, dims=[2], params=[2.0, 0.1], name="ring", posteriors=["common"])
    orc.register_user_cost(ring)
    model = k.CommonLogDensity(2, k.InitFromSnippet(2), ring)
    o = orc.OracleAIS(model, 64, seed=3).init()
    x0 = o.state()[0]
    r0 = np.hypot(x0[:, 0], x0[:, 1])
    assert np.all(np.abs(r0 - 2.0) < 0.4) and r0.std() > 0.01       # drawn on the ring by the snippet
    x = o.steps_serial(64 * 50, 5)
    assert abs(np.hypot(x[:, 0], x[:, 1]).mean() - 2.0) < 0.05
    # a snippet without the hook cannot serve such a model
    plain = k.costs.UserCost(BANANA_LPI + "// no init\\n", dims=[2], name="plain", posteriors=["common"])
    orc.register_user_cost(plain)
    with pytest.raises(orc.OracleError, match="kabc_user_sample_init"):
        orc.OracleAIS(k.CommonLogDensity(2, k.InitFromSnippet(2), plain), 64, seed=3).init()


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["hiprtc", "hipcc"])
def test_user_sample_init_gpu_bit_exact(k, orc, gpu_ctx, monkeypatch, form):
    """CommonLogDensity with an arbitrary sample_init: the snippet draws the initial walkers on
    the device (ais_init_kernel calls kabc_user_sample_init), bit-identical to the oracle calling
    the same function through gcc -- both forms of a user cost."""
    if form == "hipcc":
        monkeypatch.setenv("KABC_USER_PLUGIN", "hipcc")
    ring = k.costs.UserCost(RING_SRC + f"// {form}\\n", dims=[2], params=[2.0, 0.1], name="ring",
                            posteriors=["common"])
    orc.register_user_cost(ring)
    model = k.CommonLogDensity(2, k.InitFromSnippet(2), ring)
    ens = k.AisEnsemble(model, 500, seed=8).init()
    o = orc.OracleAIS(model, 500, seed=8).init()
    for a, b in zip(ens.state()[:3], o.state()[:3]):
        assert np.array_equal(a, b)
    assert np.array_equal(ens.advance(3, 6, collect=True), o.generations_sync(3, 6))
    res = k.sample(model, k.AIS(200), 4000, ntransitions=20, discard_initial=2000, seed=1, return_array=True)
    assert abs(np.hypot(res[:, 0], res[:, 1]).mean() - 2.0) < 0.03
    # the kind is refused where nothing can draw it
    plain = k.costs.UserCost(BANANA_LPI + f"// no init {form}\\n", dims=[2], name="plain", posteriors=["common"])
    with pytest.raises(k.KabcError, match="KABC_USER_SAMPLE_INIT"):
        k.AisEnsemble(k.CommonLogDensity(2, k.InitFromSnippet(2), plain), 64, seed=1)
    with pytest.raises(k.KabcError, match="KABC_USER_SAMPLE_INIT"):
        k.AisEnsemble(k.ApproxKernelizedPosterior(k.InitFromSnippet(2), k.costs.Rosenbrock(), 1.0), 64, seed=1)
