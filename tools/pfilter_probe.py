"""pfilter wall time per iteration (Normal^2 + gauss_dist) against the particle count, with every bad
particle's rejection loop inside one launch (default) and with one launch per attempt (KABC_PF_PASSES=1)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kissabc_jl_amd as k  # noqa: E402

N2 = k.Factored(k.Normal(0, 5), k.Normal(0, 5))
cost = k.costs.GaussDist([1.0, -0.5])
kw = dict(q=0.7, eff_tol=0.1, epstol=0.02, seed=3)
for N in [int(a) for a in sys.argv[1:]] or [50, 1000, 16384, 262144]:
    row = {"N": N}
    for mode in ("0", "1"):
        os.environ["KABC_PF_PASSES"] = mode
        k.pfilter(N2, cost, N, return_array=True, **kw)
        ws = []
        for _ in range(5):
            t0 = time.perf_counter()
            r = k.pfilter(N2, cost, N, return_array=True, **kw)
            ws.append(time.perf_counter() - t0)
        row["loop_in_kernel" if mode == "0" else "launch_per_attempt"] = {
            "wall_ms": round(sorted(ws)[2] * 1e3, 3), "iterations": int(r.info["iterations"]),
            "us_per_iteration": round(sorted(ws)[2] * 1e6 / r.info["iterations"], 1)}
    print(json.dumps(row), flush=True)
