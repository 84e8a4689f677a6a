"""AIS throughput beyond the bench configuration: evals/s and half-generation kernel
time for several (N, D, prior class, cost) combinations at ntransitions = 100 and 16.
Usage: python tools/config_sweep.py"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kissabc_jl_amd as k  # noqa: E402


def configs():
    rng = np.random.default_rng(1)
    U = lambda D: k.Factored(*[k.Uniform(-5, 5)] * D)          # noqa: E731
    Nn = lambda D: k.Factored(*[k.Normal(0, 5)] * D)           # noqa: E731
    H16 = k.Factored(k.Normal(0, 5), k.Uniform(0, 5), *[k.Normal(0, 1)] * 14)
    G4 = k.Factored(k.Gamma(2.5, 0.7), k.LogNormal(0.3, 0.6), k.Beta(2, 3), k.Normal(0, 1))
    RD = k.Factored(k.Uniform(1, 3), k.Truncated(k.Normal(0, 0.1), 0, 100))
    return [
        ("README example N=65536 D=2, normal_meanstd_sim n=1000 (500 Box-Muller pairs per cost)", k.ApproxKernelizedPosterior(RD, k.costs.NormalMeanStdSim(1000, 2.0, 0.04), 0.005), 65536),
        ("C2 N=4096 D=2 Normal priors, gauss_dist", k.ApproxKernelizedPosterior(Nn(2), k.costs.GaussDist([1.0, -0.5]), 0.1), 4096),
        ("C3 N=65536 D=8 box, rosenbrock (bench)", k.ApproxKernelizedPosterior(U(8), k.costs.Rosenbrock(), 1.0), 65536),
        ("N=65536 D=8 box, rosenbrock, ApproxPosterior", k.ApproxPosterior(U(8), k.costs.Rosenbrock(), 30.0), 65536),
        ("N=1048576 D=8 box, rosenbrock", k.ApproxKernelizedPosterior(U(8), k.costs.Rosenbrock(), 1.0), 1 << 20),
        ("N=65536 D=4 box, rosenbrock", k.ApproxKernelizedPosterior(U(4), k.costs.Rosenbrock(), 1.0), 65536),
        ("N=65536 D=16 box, rosenbrock", k.ApproxKernelizedPosterior(U(16), k.costs.Rosenbrock(), 2.0), 65536),
        ("N=65536 D=8 Normal priors (SIMPLE), gauss_dist", k.ApproxKernelizedPosterior(Nn(8), k.costs.GaussDist(np.zeros(8)), 1.0), 65536),
        ("N=65536 D=4 Gamma/LogNormal/Beta/Normal (GENERAL), norm_shell", k.ApproxKernelizedPosterior(G4, k.costs.NormShell(2.0), 0.5), 65536),
        ("N=32768 D=16 hier priors, hier_gauss_sim (stochastic cost)", k.ApproxKernelizedPosterior(H16, k.costs.HierGaussSim(rng.normal(size=14)), 0.3), 32768),
    ]


out = []
for name, model, N in configs():
    row = {"config": name}
    for nt in (100, 16):
        ens = k.AisEnsemble(model, N, seed=1).init()
        gens = max(2, min(60, int((4e8 if 'README' not in name else 2e7) / (N * nt))))
        ens.advance(3, nt)
        ens.set_timing(2 * gens, stride=1 if gens < 8 else 8)
        ens.advance(gens, nt)
        kms, nl = ens.kernel_ms()
        st = ens.stats()
        row[f"nt{nt}"] = {"kernel_us": round(kms * 1e3, 2), "G_evals_per_s_kernel": round(N / 2 * nt / (kms * 1e-3) / 1e9, 2),
                          "accept_rate": round(st["accepted"] / max(1, st["proposals"]), 3)}
        ens.close()
    out.append(row)
    print(json.dumps(row), flush=True)
