"""The eleven built-in DeviceCost formulas (include/kabc_costs.h) against independent numpy
restatements of the reference workloads they stand for.  The header is compiled into BOTH
the HIP kernels and the oracle, so device-vs-oracle bit parity cannot see a wrong formula;
these tests can: deterministic costs exactly (same summation order), stochastic simulators
by the moments / support their definition implies (20 000 draws, tolerances of 4-5
standard errors)."""
import numpy as np
import pytest


def _draws(orc, cost, x, n=20000, seed=11):
    return np.array([orc.cost_eval(cost, x, seed=seed, walker=w, t=3) for w in range(n)])


def test_gauss_dist(orc, k):          # ||x - c||_2
    rng = np.random.default_rng(0)
    for D in (1, 2, 5, 16):
        c = rng.normal(size=D)
        for _ in range(20):
            x = rng.normal(size=D) * 3
            s = 0.0
            for kk in range(D):
                s += (x[kk] - c[kk]) ** 2
            assert orc.cost_eval(k.costs.GaussDist(c), x) == np.sqrt(s)


def test_rosenbrock(orc, k):          # sqrt(sum 100 (x[k+1]-x[k]^2)^2 + (1-x[k])^2)
    rng = np.random.default_rng(1)
    for D in (2, 3, 8, 16):
        for _ in range(20):
            x = rng.uniform(-5, 5, size=D)
            s = 0.0
            for kk in range(D - 1):
                a, b = x[kk + 1] - x[kk] * x[kk], 1.0 - x[kk]
                s += 100.0 * a * a + b * b
            assert orc.cost_eval(k.costs.Rosenbrock(), x) == np.sqrt(s)
    assert orc.cost_eval(k.costs.Rosenbrock(), np.ones(8)) == 0.0


def test_scalar_deterministic_costs(orc, k):
    rng = np.random.default_rng(2)
    for _ in range(50):
        x = rng.normal() * 2
        assert orc.cost_eval(k.costs.DiracSq(1.5), [x]) == abs(x * x + 1.0 - 1.5)   # runtests.jl:79-80
        assert orc.cost_eval(k.costs.AbsDiff(1.5), [x]) == abs(x - 1.5)             # runtests.jl:178
    for D in (1, 4, 9):
        x = rng.normal(size=D)
        s = 0.0
        for v in x:
            s += v * v
        assert orc.cost_eval(k.costs.NormShell(1.5), x) == abs(np.sqrt(s) - 1.5)    # runtests.jl:186


def test_noisy_quad_du(orc, k):       # runtests.jl:108-109: |(n^2+du)(n + 0.01 randn) - target|
    n_, du, target = 1.3, 4.0, 5.5
    v = _draws(orc, k.costs.NoisyQuadDU(target), [n_, du])
    base = (n_ * n_ + du) * n_ - target          # 1.897: far from 0, the abs is inactive
    sd = 0.01 * (n_ * n_ + du)
    assert abs(v.mean() - base) < 5 * sd / np.sqrt(v.size)
    assert abs(v.std() / sd - 1) < 0.03
    assert abs(((v - base) / sd > 1.0).mean() - 0.158655) < 0.012    # Gaussian tail


def test_mixture(orc, k):             # runtests.jl:145-146: |mu + rand((0.1 randn, randn)) - target|
    v = _draws(orc, k.costs.Mixture(0.0), [10.0]) - 10.0     # e = 0.1 z or z, each w.p. 1/2
    assert abs(v.mean()) < 5 * np.sqrt(0.505 / v.size)
    assert abs(v.var() / 0.505 - 1) < 0.06
    # P(|e| < 0.3) = (P(|z| < 3) + P(|z| < 0.3)) / 2
    assert abs((np.abs(v) < 0.3).mean() - (0.9973002 + 0.2358228) / 2) < 0.015
    w = _draws(orc, k.costs.Mixture(2.0), [2.0], n=4000)      # abs active: |e| >= 0
    assert w.min() >= 0.0


def test_noisy_banana(orc, k):        # runtests.jl:242,248
    v = _draws(orc, k.costs.NoisyBanana(0.0), [1.0, 1.0])     # 50 (0.01 z0)^2 + (0.01 z1)^2
    assert np.all(np.isfinite(v)) and v.min() >= 0
    assert abs(v.mean() / (50e-4 + 1e-4) - 1) < 0.05
    x = np.array([0.3, -0.7])
    u = _draws(orc, k.costs.NoisyBanana(0.0), x)
    a0, b0 = x[0] - x[1] ** 2, x[1] - 1.0
    mean = 50 * (a0 * a0 + 1e-4) + (b0 * b0 + 1e-4)
    assert abs(u.mean() / mean - 1) < 0.01
    h = _draws(orc, k.costs.NoisyBanana(0.5), x)              # cost returning Inf half the time
    assert abs(np.isinf(h).mean() - 0.5) < 0.02
    assert abs(h[np.isfinite(h)].mean() / mean - 1) < 0.02


def test_wiener_rms(orc, k):          # runtests.jl:116-126
    t = np.arange(31.0)
    mu, sig = 0.5, 2.0
    curve = np.sqrt(mu * mu * t * t + sig * sig * t)
    zero = _draws(orc, k.costs.WienerRms(np.zeros(31)), [mu, sig], n=8000)   # = jit * mean(curve)
    r = zero / curve.mean()
    assert r.min() >= 0.95 and r.max() <= 1.05
    assert abs(r.mean() - 1.0) < 5 * (0.1 / np.sqrt(12)) / np.sqrt(r.size)
    assert abs(r.std() / (0.1 / np.sqrt(12)) - 1) < 0.05
    # against the data itself the cost is mean |curve (jit - 1)| = mean(curve) |jit - 1|
    exact = _draws(orc, k.costs.WienerRms(curve), [mu, sig], n=8000)
    assert abs(exact.mean() / (curve.mean() * 0.025) - 1) < 0.05 and exact.max() <= 0.05 * curve.mean() + 1e-12


def test_hier_gauss_sim(orc, k):      # ybar_g = m + s z_g + randn/sqrt(8); RMS(ybar - obs)
    rng = np.random.default_rng(3)
    for D in (3, 8, 16):
        G = D - 2
        z = rng.normal(size=G)
        m, s = 0.7, 1.9
        x = np.concatenate([[m, s], z])
        obs = m + s * z                                          # the noiseless simulation
        v = _draws(orc, k.costs.HierGaussSim(obs), x, n=12000)   # sqrt(mean(noise^2) / 8)
        q = v * v * 8 * G                                        # ~ chi^2_G
        assert abs(q.mean() / G - 1) < 5 * np.sqrt(2.0 / G / q.size) + 0.01
        assert abs(q.var() / (2 * G) - 1) < 0.12
        shifted = _draws(orc, k.costs.HierGaussSim(obs + 3.0), x, n=4000)
        assert abs((shifted ** 2).mean() - (9.0 + 1.0 / 8)) < 5 * np.sqrt(4.6 / G / shifted.size) + 0.01


def test_normal_meanstd_sim(orc, k):  # README.md:43-49: hypot(mean(x) - mean(tdata), 50 (std(x) - std(tdata)))
    n, mu, sig = 1000, 2.0, 0.04
    v = _draws(orc, k.costs.NormalMeanStdSim(n, mu, sig), [mu, sig], n=4000)
    want = sig * sig / n + 2500 * sig * sig / (2 * (n - 1))      # E[a^2] + E[b^2]
    assert abs((v * v).mean() / want - 1) < 0.08
    off = _draws(orc, k.costs.NormalMeanStdSim(n, mu + 0.1, sig), [mu, sig], n=2000)
    assert abs(off.mean() - 0.1) < 0.01                           # dominated by the mean offset
    wide = _draws(orc, k.costs.NormalMeanStdSim(n, mu, sig), [mu, 2 * sig], n=2000)
    assert abs(wide.mean() - 50 * sig) < 0.1                      # dominated by 50 (2 sig - sig)
