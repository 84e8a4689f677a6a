"""C4 (BASELINE.json configs[3]): smc() adaptive-eps on the 16-param hierarchical
Gaussian simulator, 32768 particles.  Prints wall time, per-pass kernel time and
the roofline figure of the propose+accept kernel (B = 32 D + 33 bytes per particle
update).  With --oracle also times the CPU oracle on the same problem."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kissabc_jl_amd as k  # noqa: E402


def c4_problem():
    rng = np.random.default_rng(1)
    zstar = rng.normal(size=14)
    ybar = 1.0 + 0.5 * zstar + rng.normal(size=14) / np.sqrt(8)
    prior = k.Factored(k.Normal(0, 5), k.Uniform(0, 5), *[k.Normal(0, 1)] * 14)
    return prior, k.costs.HierGaussSim(ybar)


if __name__ == "__main__":
    prior, cost = c4_problem()
    N = 32768
    kw = dict(nparticles=N, alpha=0.95, epstol=0.05, seed=1)
    if "--spec" in sys.argv:   # kernels specialised for this model (kabc_compile_model)
        k.compile_model(prior, cost, families=2)
    k.smc(prior, cost, **kw)  # warm-up
    k.smc(prior, cost, **kw)
    t0 = time.perf_counter()
    r = k.smc(prior, cost, return_array=True, **kw)
    wall = time.perf_counter() - t0
    D = 16
    B = 32 * D + 33
    out = {
        "config": "C4 smc N=32768 D=16 hier_gauss_sim alpha=0.95 epstol=0.05",
        "iterations": r.info["iterations"], "eps": r.eps, "n_alive": r.info["n_alive"],
        "wall_s": wall, "proposals": r.info["proposals"], "cost_evals": r.info["cost_evals"],
        "particle_updates_per_s_wall": r.info["proposals"] / wall,
        "mcmc_kernel_avg_ms": r.info["kernel_ms_mcmc"], "mcmc_launches": r.info["mcmc_launches"],
        "mcmc_kernel_GBps_algorithmic": N * B / (r.info["kernel_ms_mcmc"] * 1e-3) / 1e9,
        "posterior_mean_m_s": [float(r.P[:, 0].mean()), float(r.P[:, 1].mean())],
    }
    if "--oracle" in sys.argv:
        from oracle import oracle as orc
        t0 = time.perf_counter()
        ro = orc.smc(prior, cost, **kw)
        out["oracle_wall_s"] = time.perf_counter() - t0
        out["oracle_bit_exact"] = bool(np.array_equal(ro["theta_all"], r.info["theta_all"])
                                       and ro["eps"] == r.eps)
    print(json.dumps(out))
