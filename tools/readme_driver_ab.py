"""README.md:31-57 (BASELINE.json configs[0]: AIS(10), 1000 samples, ntransitions = 100, a 1000-draw simulator per
cost) through sample(), on the one-workgroup driver (default) and on a launch per half-generation
(KABC_AIS_SMALL=0); one chain and 50 chains at once; wall ms, median of 5 / 3."""
import json
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kissabc_jl_amd as k  # noqa: E402
import bench  # noqa: E402
rm = bench.readme_problem(k)
out = {}
for name, env in (("one_workgroup", "1"), ("launch_per_half_generation", "0"), ("one_workgroup_again", "1")):
    os.environ["KABC_AIS_SMALL"] = env
    k.sample(rm, k.AIS(10), 1000, ntransitions=100, seed=1, return_array=True)
    walls = []
    for _ in range(5):
        t0 = time.perf_counter()
        res = k.sample(rm, k.AIS(10), 1000, ntransitions=100, seed=1, return_array=True)
        walls.append(time.perf_counter() - t0)
    k.sample(rm, k.AIS(10), k.MCMCThreads(), 1000, 50, ntransitions=100, seed=1, return_array=True)
    w50 = []
    for _ in range(3):
        t0 = time.perf_counter()
        r50 = k.sample(rm, k.AIS(10), k.MCMCThreads(), 1000, 50, ntransitions=100, seed=1, return_array=True)
        w50.append(time.perf_counter() - t0)
    out[name] = {"wall_ms": round(sorted(walls)[2] * 1e3, 2), "chains_50_wall_ms": round(sorted(w50)[1] * 1e3, 2),
                 "posterior_mean": [round(v, 5) for v in res.mean(0).tolist()], "checksum": float(res.sum()),
                 "checksum_50": float(r50.sum())}
print(json.dumps(out))
