"""C4 smc wall time, steady state (median of 9 runs after 6 warm-up calls), and bit-exactness of
theta / eps against the oracle (once).  One JSON line."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kissabc_jl_amd as k  # noqa: E402
import bench  # noqa: E402

prior, cost, kw = bench.c4_problem(k)
for _ in range(6):
    r = k.smc(prior, cost, return_array=True, **kw)
ws = []
for _ in range(9):
    t0 = time.perf_counter()
    r = k.smc(prior, cost, return_array=True, **kw)
    ws.append(time.perf_counter() - t0)
out = {"wall_ms_median": sorted(ws)[4] * 1e3, "wall_ms_min": min(ws) * 1e3, "iterations": r.info["iterations"],
       "us_per_iteration": sorted(ws)[4] * 1e6 / r.info["iterations"]}
if "--oracle" in sys.argv:
    from oracle import oracle as orc
    ro = orc.smc(prior, cost, **kw)
    out["bit_exact"] = bool(np.array_equal(ro["theta_all"], r.info["theta_all"]) and ro["eps"] == r.eps)
print(json.dumps(out))
