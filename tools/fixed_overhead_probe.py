import sys, time
sys.path.insert(0, "/root/repo")
import kissabc_jl_amd as k
from tools.smc_c4_probe import c4_problem
prior, cost = c4_problem()
N2 = k.Factored(k.Normal(0, 5), k.Normal(0, 5)); g = k.costs.GaussDist([1.0, -0.5])
def t(f, n=30):
    f(); ws=[]
    for _ in range(n):
        t0=time.perf_counter(); r=f(); ws.append(time.perf_counter()-t0)
    return sorted(ws)[n//2]*1e3, r
for N in (100, 200, 1000, 32768):
    ms, r = t(lambda: k.smc(prior, cost, nparticles=N, epstol=1e9, seed=1, return_array=True))
    print("smc C4-model N=%d, one iteration: %.3f ms (iterations %d) host_ms %s" % (N, ms, r.info["iterations"], {a: round(b,3) for a,b in r.info["host_ms"].items()}))
ms, r = t(lambda: k.smc(N2, g, nparticles=100, epstol=1e9, seed=1, return_array=True))
print("smc gauss N=100, one iteration: %.3f ms" % ms, r.info["iterations"])
ms, r = t(lambda: k.pfilter(N2, g, 100, epstol=1e9, seed=1, return_array=True))
print("pfilter gauss N=100, one iteration: %.3f ms" % ms, r.info["iterations"])
ms, r = t(lambda: k.ABCDE(N2, g, 1e9, nparticles=50, generations=1, seed=1, return_array=True))
print("ABCDE 50 x 1 generation: %.3f ms" % ms)
