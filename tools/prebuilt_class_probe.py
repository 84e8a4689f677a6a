"""What the GENERAL prior class costs a prior of the NORMAL / SIMPLE class on the prebuilt half-generation kernel
(KABC_SPECIALIZE=0: no model-specific kernel) -- the question behind shrinking the prebuilt matrix to the BOX and
GENERAL classes.  us per half-generation launch at ntransitions = 100 (hipEvents, 8 launches per pair)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["KABC_SPECIALIZE"] = "0"
import kissabc_jl_amd as k  # noqa: E402
import bench  # noqa: E402

hier = k.Factored(k.Normal(0, 5), k.Uniform(0, 5), *[k.Normal(0, 1)] * 14)
ybar = np.random.default_rng(1).normal(size=14)
cases = [
    ("C2 normal d2 (NORMAL)", bench.c2_problem(k), 4096),
    ("normal8 gauss_dist (NORMAL)", k.ApproxKernelizedPosterior(k.Factored(*[k.Normal(0, 5)] * 8), k.costs.GaussDist(np.zeros(8)), 1.0), 65536),
    ("uniform+normal d4 (SIMPLE)", k.ApproxKernelizedPosterior(k.Factored(k.Uniform(-3, 3), k.Normal(0, 1), k.Uniform(0, 2), k.Normal(1, 2)),
                                                              k.costs.NormShell(2.0), 0.5), 65536),
    ("c4 prior + hier sim d16 (SIMPLE)", k.ApproxKernelizedPosterior(hier, k.costs.HierGaussSim(ybar), 0.3), 32768),
]
for name, model, N in cases:
    row = {"model": name, "N": N}
    for cls in ("own", "general"):
        if cls == "general":
            os.environ["KABC_PREBUILT_CLASS"] = "g"
        else:
            os.environ.pop("KABC_PREBUILT_CLASS", None)
        e = k.AisEnsemble(model, N, seed=1).init()
        t0 = time.perf_counter()
        e.advance(1, 100)
        k.default_context().synchronize()
        first = time.perf_counter() - t0
        e.advance(3, 100)
        e.set_timing(64, stride=8)
        e.advance(16, 100)
        ms, n = e.kernel_ms()
        row[cls] = {"us_per_launch": round(ms * 1e3, 2), "first_call_ms": round(first * 1e3, 2)}
        e.close()
    row["general_over_own"] = round(row["general"]["us_per_launch"] / row["own"]["us_per_launch"], 3)
    print(json.dumps(row), flush=True)
