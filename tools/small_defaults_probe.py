import sys, time, os
sys.path.insert(0, "/root/repo")
import numpy as np
import kissabc_jl_amd as k
N2 = k.Factored(k.Normal(0, 5), k.Normal(0, 5))
cost = k.costs.GaussDist([1.0, -0.5])
def t(f, n=20):
    f(); ws=[]
    for _ in range(n):
        t0=time.perf_counter(); r=f(); ws.append(time.perf_counter()-t0)
    return sorted(ws)[n//2]*1e3, r
ms, r = t(lambda: k.ABCDE(N2, cost, 0.01, nparticles=50, generations=20, seed=3, return_array=True))
print("ABCDE defaults (50 x 20 generations): %.3f ms" % ms, r.info.get("generations_run"))
ms, r = t(lambda: k.ABCDE(N2, cost, 0.01, nparticles=50, generations=500, seed=3, return_array=True))
print("ABCDE 50 x 500 generations: %.3f ms" % ms, r.info.get("generations_run"))
ms, r = t(lambda: k.pfilter(N2, cost, 100, seed=3, return_array=True))
print("pfilter 100: %.3f ms" % ms, {a: r.info[a] for a in ("iterations",) if a in r.info})
ms, r = t(lambda: k.smc(N2, cost, seed=3, return_array=True))
print("smc defaults (100): %.3f ms" % ms, r.info["iterations"])
