import json, os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import kissabc_jl_amd as k
tdata = np.random.default_rng(0).normal(2.0, 0.04, 1000)
prior = k.Factored(k.Uniform(1, 3), k.Truncated(k.Normal(0, 0.1), 0, 100))
cost = k.costs.NormalMeanStdSim(1000, tdata.mean(), tdata.std(ddof=1))
N2 = k.Factored(k.Normal(0, 5), k.Normal(0, 5))
out = {}
for name, pr, co, kw in [("readme_smc", prior, cost, dict(nparticles=100, seed=1)),
                         ("gauss_d2_n100", N2, k.costs.GaussDist([1.0, -0.5]), dict(nparticles=100, seed=1, epstol=0.01)),
                         ("gauss_d2_n256", N2, k.costs.GaussDist([1.0, -0.5]), dict(nparticles=256, seed=1, epstol=0.01))]:
    for small in ("1", "0"):
        os.environ["KABC_SMC_SMALL"] = small
        for _ in range(3): k.smc(pr, co, return_array=True, **kw)
        ws = []
        for _ in range(7):
            t0 = time.perf_counter(); r = k.smc(pr, co, return_array=True, **kw); ws.append(time.perf_counter() - t0)
        out[f"{name}_small{small}"] = {"ms": round(sorted(ws)[3] * 1e3, 3), "iterations": r.info["iterations"]}
print(json.dumps(out))
