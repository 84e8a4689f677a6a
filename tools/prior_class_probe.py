"""Half-generation kernel time (ntransitions = 100) of the AIS kernel for the prior classes beside
BOX: NORMAL (Normal^8), SIMPLE (README prior; C4's prior + simulator), GENERAL (socks prior; the
four-family D = 4 prior).  One JSON line; used with KABC_LIB for A/B runs."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kissabc_jl_amd as k  # noqa: E402

socks = k.Factored(k.NegativeBinomial(900 / 195, (900 / 195) / (30 + 900 / 195)), k.Beta(15, 2))
G4 = k.Factored(k.Gamma(2.5, 0.7), k.LogNormal(0.3, 0.6), k.Beta(2, 3), k.Normal(0, 1))
H16 = k.Factored(k.Normal(0, 5), k.Uniform(0, 5), *[k.Normal(0, 1)] * 14)
RD = k.Factored(k.Uniform(1, 3), k.Truncated(k.Normal(0, 0.1), 0, 100))
cases = [
    ("normal8", k.ApproxKernelizedPosterior(k.Factored(*[k.Normal(0, 5)] * 8), k.costs.GaussDist(np.zeros(8)), 1.0), 65536),
    ("readme_prior_gauss", k.ApproxKernelizedPosterior(RD, k.costs.GaussDist([2.0, 0.04]), 0.05), 65536),
    ("c2", k.ApproxKernelizedPosterior(k.Factored(k.Normal(0, 5), k.Normal(0, 5)), k.costs.GaussDist([1.0, -0.5]), 0.1), 4096),
    ("socks", k.ApproxKernelizedPosterior(socks, k.costs.GaussDist([40.0, 0.8]), 3.0), 65536),
    ("general4", k.ApproxKernelizedPosterior(G4, k.costs.NormShell(2.0), 0.5), 65536),
    ("hier16_sim", k.ApproxKernelizedPosterior(H16, k.costs.HierGaussSim(np.random.default_rng(1).normal(size=14)), 0.3), 32768),
]
out = {}
for name, model, N in cases:
    ens = k.AisEnsemble(model, N, seed=1).init()
    ens.advance(3, 100)
    ens.set_timing(64, stride=8)
    ens.advance(32, 100)
    kms, _ = ens.kernel_ms()
    out[name] = round(kms * 1e3, 1)
    ens.close()
print(json.dumps(out))
