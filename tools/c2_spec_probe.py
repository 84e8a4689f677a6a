"""C2 (4096 walkers, Normal(0,5)^2, gauss_dist) and the same model at 65 536 walkers: the prebuilt
NORMAL-class kernel (max-ILP scheduling, csrc/Makefile AIS_SCHED) against the model's own kernel
compiled with and without that scheduling strategy (KABC_SPEC_SCHED).  One process per variant:
a unit is compiled once per process."""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

if len(sys.argv) > 1:
    sys.path.insert(0, ROOT)
    import bench
    import kissabc_jl_amd as k
    out = {}
    for N in (4096, 65536):
        m = bench.c2_problem(k)
        e = k.AisEnsemble(m, N, seed=1).init()
        e.advance(5, 100)
        ts = []
        for _ in range(5):
            e.set_timing(200, stride=8)
            e.advance(100, 100)
            ts.append(e.kernel_ms()[0] * 1e3)
        out[str(N)] = {"kernel_us": sorted(ts)[2], "state": e.spec_state()[0]}
    print(json.dumps(out))
    sys.exit(0)

for name, env in (("prebuilt (KABC_SPECIALIZE=0)", {"KABC_SPECIALIZE": "0"}),
                  ("own kernel, max-ilp", {"KABC_SPECIALIZE": "1"}),
                  ("own kernel, default scheduling", {"KABC_SPECIALIZE": "1", "KABC_SPEC_SCHED": "0"})):
    d = tempfile.mkdtemp(prefix="kabc_c2_")
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], capture_output=True, text=True,
                       env=dict(os.environ, KABC_RTC_CACHE_DIR=d, **env))
    print(name, r.stdout.strip() or r.stderr[-400:], flush=True)
