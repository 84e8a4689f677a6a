"""smc() on C4's model (16-param hierarchical Gaussian simulator) against the particle count:
the loop kernel (N <= 65 536) and the kernel-per-phase path beyond it.  Wall per eps-iteration and
particle updates per second."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kissabc_jl_amd as k  # noqa: E402
from tools.smc_c4_probe import c4_problem  # noqa: E402

prior, cost = c4_problem()
for N in [int(a) for a in sys.argv[1:]] or [32768, 65536, 131072, 524288, 2097152]:
    kw = dict(nparticles=N, alpha=0.95, epstol=0.05, seed=1)
    k.smc(prior, cost, **kw)
    walls = []
    for _ in range(3):
        t0 = time.perf_counter()
        r = k.smc(prior, cost, return_array=True, **kw)
        walls.append(time.perf_counter() - t0)
    wall = sorted(walls)[1]
    it = r.info["iterations"]
    print(json.dumps({"N": N, "loop_env": os.environ.get("KABC_SMC_LOOP"), "iterations": it, "wall_ms": wall * 1e3,
                      "us_per_iteration": wall * 1e6 / it, "updates_per_s": r.info["proposals"] / wall,
                      "GBps_algorithmic": r.info["proposals"] * 545 / wall / 1e9,
                      "mcmc_kernel_avg_ms": r.info["kernel_ms_mcmc"]}), flush=True)
