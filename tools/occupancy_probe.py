"""Half-generation kernel time of the C3 model against the number of workgroups: how a launch
behaves below / at / above one residency wave (512 workgroups of 256 threads on 256 CUs at
amdgpu_waves_per_eu(2,2)).  Decides whether row-chunked launches can pipeline an exchange.
Usage: python tools/occupancy_probe.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kissabc_jl_amd as k  # noqa: E402

prior = k.Factored(*[k.Uniform(-5, 5)] * 8)
model = k.ApproxKernelizedPosterior(prior, k.costs.Rosenbrock(), 1.0)
for N in (8192, 16384, 32768, 49152, 65536, 98304, 131072, 262144):
    row = {"N": N, "workgroups": N // 2 // 64}
    for nt in (100, 16, 1):
        ens = k.AisEnsemble(model, N, seed=1).init()
        gens = max(8, min(400, int(2e8 / (N * nt))))
        ens.advance(3, nt)
        ens.set_timing(2 * gens, stride=8)
        ens.advance(gens, nt)
        kms, nl = ens.kernel_ms()
        row[f"nt{nt}_us"] = round(kms * 1e3, 2)
        ens.close()
    print(json.dumps(row), flush=True)
