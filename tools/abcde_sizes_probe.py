"""ABCDE wall time per run (Normal^2 + gauss_dist, 50 generations; 10 from 32 768 particles on) against the
particle count: the default donor-draw path (generation kernel's own scans below 256 particles, teams of sixteen
lanes -- abcde_donor_kernel -- below 4096, the rank structure beyond) and KABC_ABCDE_DONOR=0 (own scans below 4096)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kissabc_jl_amd as k  # noqa: E402

N2 = k.Factored(k.Normal(0, 5), k.Normal(0, 5))
cost = k.costs.GaussDist([1.0, -0.5])
for N in [int(a) for a in sys.argv[1:]] or [50, 256, 1000, 2000, 4000, 8192, 16384, 32768]:
    gens = 50 if N < 32768 else 10
    row = {"N": N, "generations": gens}
    for name, env in (("default", {"KABC_ABCDE_DONOR": "1"}), ("own_scans", {"KABC_ABCDE_DONOR": "0"}),
                      ("wavelet", {"KABC_ABCDE_RANK": "wavelet"}), ("blocks", {"KABC_ABCDE_BLOCKS_FROM": "256"})):
        if (name == "own_scans" and N >= 4096) or (name == "wavelet" and (N < 4096 or N > 131072)) or \
                (name == "blocks" and (N < 256 or N >= 4096)):
            continue
        os.environ.pop("KABC_ABCDE_RANK", None)
        os.environ.pop("KABC_ABCDE_BLOCKS_FROM", None)
        os.environ.update(env)
        kw = dict(nparticles=N, generations=gens, seed=3)
        k.ABCDE(N2, cost, 0.01, return_array=True, **kw)
        ws = []
        for _ in range(5):
            t0 = time.perf_counter()
            k.ABCDE(N2, cost, 0.01, return_array=True, **kw)
            ws.append(time.perf_counter() - t0)
        row[name + "_ms"] = round(sorted(ws)[2] * 1e3, 3)
    print(json.dumps(row), flush=True)
