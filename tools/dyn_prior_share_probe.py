import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import kissabc_jl_amd as k
D = 40
def run(prior, name):
    cost = k.costs.GaussDist(np.linspace(0.2, 0.8, D))
    model = k.ApproxKernelizedPosterior(prior, cost, 2.0)
    ens = k.AisEnsemble(model, 16384, seed=1).init()
    ens.advance(2, 20)
    ens.set_timing(64, stride=1)
    ens.advance(5, 20)
    print(name, "kernel ms per half-generation launch (nt=20):", ens.kernel_ms())
box = k.Factored(*[k.Uniform(0, 1)] * D)
gen = k.Factored(*[[k.Normal(0.5, 1), k.Gamma(2.5, 0.3), k.LogNormal(-0.7, 0.4), k.Beta(2, 3)][j % 4] for j in range(D)])
run(box, "box D=40")
run(gen, "general D=40")
p = k.Factored(*[k.Normal(0.5, 1)] * D)
run(p, "normal D=40")
