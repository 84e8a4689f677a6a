"""Half-generation kernel time (ntransitions = 100 unless --nt) of the prebuilt AIS kernel
(KABC_SPECIALIZE=0) against the model's own kernel (KABC_SPECIALIZE=1: compiled at create), alternating,
for the prior classes the reference's tests use; "own_kernel": false = the model stays on the prebuilt
kernels by design (pure boxes, plain Normals up to seven parameters).  One JSON line: {case: {"base_us", "spec_us", "frac_base", "frac_spec", "bit_exact"}}."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kissabc_jl_amd as k  # noqa: E402

nt = int(sys.argv[sys.argv.index("--nt") + 1]) if "--nt" in sys.argv else 100
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
only = [a for a in sys.argv[1:] if not a.startswith("--") and not a.isdigit()]


def run(model, N, spec):
    os.environ["KABC_SPECIALIZE"] = "1" if spec else "0"   # the model's own kernel, compiled at create / prebuilt
    ens = k.AisEnsemble(model, N, seed=1).init()
    ens.advance(3, nt)
    ens.set_timing(64, stride=8)
    ens.advance(32, nt)
    kms, _ = ens.kernel_ms()
    x = ens.state()[0]
    state = ens.spec_state()[0]
    ens.close()
    return kms * 1e3, x, state


out = {}
for name, model, N in cases:
    if only and name not in only:
        continue
    D = len(model.prior)
    B = 8 * (3 * D + 4)
    b1, xb, _ = run(model, N, False)
    s1, xs, st = run(model, N, True)
    b2, _, _ = run(model, N, False)
    s2, _, _ = run(model, N, True)
    b, s = min(b1, b2), min(s1, s2)
    fr = lambda us: (N // 2) * nt * B / (us * 1e-6) / 8e12   # noqa: E731
    out[name] = {"base_us": round(b, 1), "frac_base": round(fr(b), 3), "own_kernel": st == "active",
                 "spec_us": round(s, 1) if st == "active" else None,
                 "frac_spec": round(fr(s), 3) if st == "active" else None,
                 "bit_exact": bool(np.array_equal(xb, xs))}
print(json.dumps(out))
