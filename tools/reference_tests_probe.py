"""Wall time of sample() calls shaped like the reference's own tests (test/runtests.jl:82-290: small
ensembles, the default ntransitions = 1, long burn-ins), per call and per half-generation launch."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kissabc_jl_amd as k  # noqa: E402

N2 = k.Factored(k.Normal(0, 5), k.Normal(0, 5))
model = k.ApproxKernelizedPosterior(N2, k.costs.GaussDist([1.0, -0.5]), 0.1)
for walkers, nsamples, discard, nt in [(12, 500, 1000, 1), (100, 1000, 10000, 1), (50, 100, 50000, 1), (20, 100, 2000, 40),
                                        (10, 10000, 0, 50)]:
    kw = dict(ntransitions=nt, discard_initial=discard, seed=1, return_array=True)
    k.sample(model, k.AIS(walkers), nsamples, **kw)
    ws = []
    for _ in range(5):
        t0 = time.perf_counter()
        k.sample(model, k.AIS(walkers), nsamples, **kw)
        ws.append(time.perf_counter() - t0)
    w = sorted(ws)[2]
    gens = -(-(nsamples + discard) // walkers)
    print(json.dumps({"AIS": walkers, "samples": nsamples, "discard_initial": discard, "ntransitions": nt,
                      "wall_ms": round(w * 1e3, 3), "generations": gens,
                      "us_per_half_generation_launch": round(w * 1e6 / (2 * gens), 2)}), flush=True)
