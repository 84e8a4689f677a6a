"""ABCDE and pfilter wall time on the device vs the CPU oracle (bit-exact check), for the
sizes the reference's defaults suggest and a large one.  Usage: python tools/abcde_probe.py"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kissabc_jl_amd as k  # noqa: E402
from oracle import oracle as orc  # noqa: E402

N2 = k.Factored(k.Normal(0, 5), k.Normal(0, 5))
cost = k.costs.GaussDist([1.0, -0.5])
out = {}
for N in (1000, 16384, 65536):
    kw = dict(nparticles=N, generations=50 if N <= 16384 else 10, seed=3)
    k.ABCDE(N2, cost, 0.01, return_array=True, **kw)
    t0 = time.perf_counter()
    r = k.ABCDE(N2, cost, 0.01, return_array=True, **kw)
    t1 = time.perf_counter()
    ro = orc.abcde(N2, cost, 0.01, **kw) if N <= 16384 else None
    t2 = time.perf_counter()
    # (the oracle's donor draw is O(N^2) per generation: not run at 65 536 -- null, not false)
    out[f"abcde_N{N}"] = {"device_s": t1 - t0, "oracle_s": (t2 - t1) if ro is not None else None,
                          "generations": r.info["generations_run"],
                          "bit_exact": bool(np.array_equal(r.P, ro["P"])) if ro is not None else None}
for N in (1000, 16384):
    kw = dict(q=0.7, eff_tol=0.1, epstol=0.02, seed=3)
    k.pfilter(N2, cost, N, return_array=True, **kw)
    t0 = time.perf_counter()
    r = k.pfilter(N2, cost, N, return_array=True, **kw)
    t1 = time.perf_counter()
    ro = orc.pfilter(N2, cost, N, **kw)
    t2 = time.perf_counter()
    out[f"pfilter_N{N}"] = {"device_s": t1 - t0, "oracle_s": t2 - t1, "info": {a: (float(b) if isinstance(b, (int, float)) else str(b)) for a, b in r.info.items()},
                            "bit_exact": bool(np.array_equal(r.P, ro["P"]))}
print(json.dumps(out))
