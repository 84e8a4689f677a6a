"""smc beyond 16 parameters (csrc/smc_dyn_kernels.hpp): the propose / accept kernel's average duration and the wall
per iteration, D = 40 (and 17, 128) x 16 384 particles, a Gaussian-distance cost."""
import json
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import kissabc_jl_amd as k  # noqa: E402
for D, N in [(17, 16384), (40, 16384), (40, 131072), (128, 16384)] if len(sys.argv) < 2 else [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]:
    for name, comp in [("normal", lambda j: k.Normal(0.5, 1.0)),
                       ("mixed", lambda j: [k.Normal(0.5, 1), k.Gamma(2.5, 0.3), k.LogNormal(-0.7, 0.4), k.Beta(2, 3)][j % 4])]:
        prior = k.Factored(*[comp(j) for j in range(D)])
        cost = k.costs.GaussDist(np.linspace(0.2, 0.8, D))
        kw = dict(nparticles=N, alpha=0.9, epstol=0.55 * D ** 0.5, seed=3, return_array=True)
        k.smc(prior, cost, **kw)
        t0 = time.perf_counter()
        r = k.smc(prior, cost, **kw)
        wall = time.perf_counter() - t0
        it = r.info["iterations"]
        B = 32 * D + 33     # bytes per particle update
        ms = r.info["kernel_ms_mcmc"]
        print(json.dumps({"D": D, "N": N, "prior": name, "iterations": it, "us_per_iteration": round(wall * 1e6 / it, 1),
                          "mcmc_kernel_ms": round(ms, 4), "pass_GBps": round(N * B / ms / 1e6, 1),
                          "team": os.environ.get("KABC_SMC_DYN_TEAM", "")}), flush=True)
