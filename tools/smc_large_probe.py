"""Where the wall time of a LARGE smc run goes (C4's model, N particles, kernel-per-phase path):
the library call (kernels + result copy) against the Python wrapper around it, and the raw
device-to-host rate into the same kind of buffer.  One JSON line per N."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kissabc_jl_amd as k  # noqa: E402
from tools.smc_c4_probe import c4_problem  # noqa: E402

prior, cost = c4_problem()
for N in [int(a) for a in sys.argv[1:]] or [524288, 2097152]:
    kw = dict(nparticles=N, alpha=0.95, epstol=0.05, seed=1)
    k.smc(prior, cost, **kw)
    ws = []
    for _ in range(3):
        t0 = time.perf_counter()
        r = k.smc(prior, cost, return_array=True, **kw)
        ws.append((time.perf_counter() - t0, r.info["host_ms"]))
    wall, host = sorted(ws, key=lambda a: a[0])[1]
    it = r.info["iterations"]
    print(json.dumps({"N": N, "iterations": it, "wall_ms": wall * 1e3, "host_ms": host,
                      "mcmc_kernel_ms_total": r.info["kernel_ms_mcmc"] * r.info["mcmc_launches"],
                      "updates_per_s": r.info["proposals"] / wall,
                      "contract_roofline_frac_by_wall": r.info["proposals"] * 545 / wall / 8e12}), flush=True)
