"""CPU: the oracle's AIS restatement (serial = the reference's schedule,
sync = the GPU schedule) against closed forms, the reference's error behaviour
and the reference's statistical known answers (test/runtests.jl, README)."""
import numpy as np
import pytest


def _c2(k):
    prior = k.Factored(k.Normal(0, 5), k.Normal(0, 5))
    return k.ApproxKernelizedPosterior(prior, k.costs.GaussDist([1.0, -0.5]), 0.1)


def test_nparticles_check_message(orc, k):
    # src/KissABC.jl:43-48
    with pytest.raises(orc.OracleError) as e:
        orc.OracleAIS(_c2(k), 6)
    assert str(e.value) == ("nparticles = 6 is insufficient, set number of particles in AIS(⋅) "
                            "atleast to 7")
    orc.OracleAIS(_c2(k), 7).init()


def test_retry_exhaustion_error(orc, k):
    # test/runtests.jl:226,237 + src/KissABC.jl:58-59: a density that is never valid
    m = k.ApproxKernelizedPosterior(k.Factored(k.Uniform(0, 1), k.Uniform(0, 1)),
                                    k.costs.NoisyBanana(1.0), 0.1)   # cost = +Inf always
    with pytest.raises(orc.OracleError) as e:
        orc.OracleAIS(m, 50, seed=3).init(retry_sampling=10)
    assert str(e.value) == ("Prior leads to ∞ costs too often, tune the prior or increase "
                            "`retry_sampling`.")
    # half of the draws invalid: init succeeds and every walker is valid
    m2 = k.ApproxKernelizedPosterior(k.Factored(k.Normal(0, 5), k.Normal(0, 5)),
                                     k.costs.NoisyBanana(0.5), 10.0)
    x, lp, ll, _ = orc.OracleAIS(m2, 200, seed=3).init().state()
    assert np.all(np.isfinite(lp + ll))


def test_init_state_is_prior_draws_and_loglike(orc, k):
    m = _c2(k)
    o = orc.OracleAIS(m, 64, seed=9).init()
    x, lp, ll, t = o.state()
    assert t == 0
    assert np.array_equal(x, orc.factored_rand(m.prior, 64, seed=9))
    assert np.array_equal(lp, orc.factored_logpdf(m.prior, x))
    c = np.sqrt(((x - np.array([1.0, -0.5])) ** 2).sum(1))
    assert np.allclose(ll, -0.5 * (c / 0.1) ** 2, rtol=1e-14)


@pytest.mark.parametrize("schedule", ["serial", "sync"])
def test_c2_posterior_matches_analytic_gaussian(orc, k, schedule):
    """Normal prior x Gaussian kernel => exactly Gaussian posterior.
    mean = c * 2500/2501, sd = sqrt(1/(1/25 + 100)); tolerance 1e-3 on the mean
    (BASELINE.json: 'posterior means within 1e-3')."""
    N, nt = 1024, 4
    o = orc.OracleAIS(_c2(k), N, seed=1).init()
    if schedule == "serial":
        o.steps_serial(N * 60, nt, collect=False)
        s = o.steps_serial(N * 400, nt)
    else:
        o.generations_sync(60, nt, collect=False)
        s = o.generations_sync(400, nt).reshape(-1, 2)
    mean_ref = np.array([1.0, -0.5]) * 2500 / 2501
    sd_ref = (1 / (1 / 25 + 100)) ** 0.5
    assert np.all(np.abs(s.mean(0) - mean_ref) < 1e-3)
    assert np.all(np.abs(s.std(0) - sd_ref) < 1.5e-3)


def test_serial_and_sync_schedules_agree_statistically(orc, k):
    U = k.Factored(*[k.Uniform(-5, 5)] * 4)
    m = k.ApproxKernelizedPosterior(U, k.costs.Rosenbrock(), 1.0)
    N, nt = 512, 8
    a = orc.OracleAIS(m, N, seed=2).init()
    b = orc.OracleAIS(m, N, seed=2).init()
    a.steps_serial(N * 100, nt, collect=False)
    b.generations_sync(100, nt, collect=False)
    sa = a.steps_serial(N * 300, nt)
    sb = b.generations_sync(300, nt).reshape(-1, 4)
    se = np.sqrt(sa.var(0) / 2000 + sb.var(0) / 2000)   # ~ensemble-level ESS
    assert np.all(np.abs(sa.mean(0) - sb.mean(0)) < 5 * se + 0.02)
    assert np.all(np.abs(sa.std(0) / sb.std(0) - 1) < 0.1)


def test_move_mixture_and_partner_uniformity(orc, k):
    """propose(): moves (1,1,1,1,2,2,3) => 4/7, 2/7, 1/7 (src/transition.jl:61-65);
    partners uniform over the complementary half, distinct."""
    N, nt = 256, 40
    o = orc.OracleAIS(_c2(k), N, seed=4).init()
    _, tr = o.generations_sync(1, nt, trace=True)
    mv = tr[0, :, :, 0].ravel()
    n = mv.size
    for m, p in ((1, 4 / 7), (2, 2 / 7), (3, 1 / 7)):
        assert abs((mv == m).mean() - p) < 4 * np.sqrt(p * (1 - p) / n)
    a, b, c = (tr[0, :, :, j] for j in (2, 3, 4))
    half0 = np.arange(N)[:, None] < N // 2
    assert np.all(np.where(half0, a >= N // 2, a < N // 2))           # complementary half
    de = tr[0, :, :, 0] >= 2
    assert np.all(a[de] != b[de])
    wk = tr[0, :, :, 0] == 3
    assert np.all((c[wk] != a[wk]) & (c[wk] != b[wk]))
    assert np.all(b[~de] == -1) and np.all(c[~wk] == -1)
    hist = np.bincount(a[~half0[:, 0]].ravel(), minlength=N)[: N // 2]
    exp = hist.sum() / (N // 2)
    assert ((hist - exp) ** 2 / exp).sum() < (N // 2) + 6 * np.sqrt(N)   # chi2 ~ dof


def test_reference_known_answers_small(orc, k):
    """Statistical pins of the reference's own tests, on the oracle (sync schedule):
    test/runtests.jl:77-86 (sim(res) ≈ 1.5), :177-182 (res ≈ 1.5),
    :106-113 (sim ≈ 5.5)."""
    m = k.ApproxKernelizedPosterior(k.Normal(1, 0.2), k.costs.DiracSq(1.5), 0.001)
    o = orc.OracleAIS(m, 12, seed=1).init()
    o.generations_sync(90, 1, collect=False)
    mu = o.generations_sync(42, 1).ravel()
    sim = mu * mu + 1
    assert abs(sim.mean() - 1.5) < 2 * max(sim.std(), 1e-3)
    m = k.ApproxPosterior(k.Normal(0, 1), k.costs.AbsDiff(1.5), 0.01)
    o = orc.OracleAIS(m, 20, seed=1).init()
    o.generations_sync(100, 1, collect=False)
    r = o.generations_sync(5, 1).ravel()
    assert abs(r.mean() - 1.5) < 2 * max(r.std(), 0.01)
    m = k.ApproxPosterior(k.Factored(k.Normal(1, 0.5), k.DiscreteUniform(1, 10)),
                          k.costs.NoisyQuadDU(5.5), 0.01)
    o = orc.OracleAIS(m, 100, seed=1).init()
    o.generations_sync(100, 1, collect=False)
    r = o.generations_sync(10, 1).reshape(-1, 2)
    assert np.array_equal(r[:, 1], np.rint(r[:, 1]))     # push_p on emission
    sim = (r[:, 0] ** 2 + r[:, 1]) * r[:, 0]
    assert abs(sim.mean() - 5.5) < 2 * max(sim.std(), 0.02)


def test_readme_example_c1(orc, k):
    """README.md:31-66: Normal(mu, sigma) inference; documented posterior
    2.0 ± 0.018, 0.0395 ± 0.00093 (tdata = 1000 draws of N(2, 0.04))."""
    rng = np.random.default_rng(0)
    tdata = rng.normal(2.0, 0.04, 1000)
    prior = k.Factored(k.Uniform(1, 3), k.Truncated(k.Normal(0, 0.1), 0, 100))
    cost = k.costs.NormalMeanStdSim(1000, tdata.mean(), tdata.std(ddof=1))
    m = k.ApproxKernelizedPosterior(prior, cost, 0.005)
    o = orc.OracleAIS(m, 10, seed=1).init()
    o.generations_sync(30, 100, collect=False)
    s = o.generations_sync(30, 100).reshape(-1, 2)
    assert abs(s[:, 0].mean() - tdata.mean()) < 0.01
    assert abs(s[:, 1].mean() - tdata.std(ddof=1)) < 0.005
    assert 0.0005 < s[:, 0].std() < 0.05


def test_set_state_invalid_start_is_an_error(orc, k):
    # src/types.jl:70 "starting sample invalid."
    o = orc.OracleAIS(_c2(k), 16, seed=1).init()
    x, lp, ll, _ = o.state()
    lp[3] = -np.inf
    o.set_state(x, lp, ll, 0)
    with pytest.raises(orc.OracleError) as e:
        o.generations_sync(1, 1)
    assert str(e.value) == "starting sample invalid."


