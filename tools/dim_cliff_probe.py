"""The step from the specialised kernels (D <= 16) to the run-time-dimension team kernels: AIS, 65 536 walkers,
20 transitions per launch, a box prior and a Gaussian-distance cost; kernel ms per half-generation launch,
evaluations per second and the algorithmic rate (24 D + 32 bytes per update)."""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import kissabc_jl_amd as k  # noqa: E402
N, NT = 65536, 20
for D in [int(a) for a in sys.argv[1:]] or [8, 12, 16, 17, 20, 24, 32, 40, 64]:
    prior = k.Factored(*[k.Uniform(0, 1)] * D)
    cost = k.costs.GaussDist(np.linspace(0.2, 0.8, D))
    model = k.ApproxKernelizedPosterior(prior, cost, 2.0)
    ens = k.AisEnsemble(model, N, seed=1).init()
    ens.advance(2, NT)
    ens.set_timing(64, stride=1)
    ens.advance(5, NT)
    ms = ens.kernel_ms()[0]
    evals = (N // 2) * NT / (ms * 1e-3)
    B = 24 * D + 32
    print(json.dumps({"D": D, "kernel_ms": round(ms, 4), "evals_per_s": round(evals / 1e9, 3), "GBps": round(evals * B / 1e9, 1),
                      "frac": round(evals * B / 8e12, 4)}), flush=True)
    ens.close()
