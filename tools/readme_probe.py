"""README.md:31-57 through sample(): AIS(10), 1000 samples, ntransitions = 100 (BASELINE.json
configs[0]); five calls, median wall time.  Run under rocprofv3 --kernel-trace --stats for the
split between the prepared-cost pre-pass and the half-generation kernel."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kissabc_jl_amd as k  # noqa: E402
import bench  # noqa: E402

m = bench.readme_problem(k)
k.sample(m, k.AIS(10), 1000, ntransitions=100, seed=1, return_array=True)
ws = []
for _ in range(5):
    t0 = time.perf_counter()
    r = k.sample(m, k.AIS(10), 1000, ntransitions=100, seed=1, return_array=True)
    ws.append(time.perf_counter() - t0)
print(json.dumps({"wall_ms_median": sorted(ws)[2] * 1e3, "posterior_mean": r.mean(0).tolist()}))
