"""smc on the socks prior of test/runtests.jl:46-72 (NegativeBinomial x Beta; the device cost is a
stand-in: gauss_dist to a point) at the reference's 5000 particles and at larger counts: wall per
eps-iteration, i.e. what the two lgammas of a NegativeBinomial log-density could matter."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kissabc_jl_amd as k  # noqa: E402

mu, sd = 30.0, 15.0
size = -mu * mu / (mu - sd * sd)
for fam in ("negbin", "poisson_like_uniform"):
    first = k.NegativeBinomial(size, size / (mu + size)) if fam == "negbin" else k.DiscreteUniform(0, 120)
    prior = k.Factored(first, k.Beta(15, 2))
    cost = k.costs.GaussDist([46.0, 0.866])
    for N in [int(a) for a in sys.argv[1:]] or [5000, 65536, 524288]:
        kw = dict(nparticles=N, alpha=0.99, epstol=0.01, r_epstol=0.0, seed=1)
        k.smc(prior, cost, **kw)
        ws = []
        for _ in range(3):
            t0 = time.perf_counter()
            r = k.smc(prior, cost, return_array=True, **kw)
            ws.append(time.perf_counter() - t0)
        w = sorted(ws)[1]
        print(json.dumps({"first_component": fam, "N": N, "iterations": r.info["iterations"], "wall_ms": w * 1e3,
                          "us_per_iteration": w * 1e6 / r.info["iterations"],
                          "mcmc_kernel_avg_us": r.info["kernel_ms_mcmc"] * 1e3}), flush=True)
