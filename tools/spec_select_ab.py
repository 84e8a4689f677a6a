"""A/B of the kernel-per-phase path's selection on one box: the select kernel in every iteration
(KABC_SMC_SPEC_SELECT=0) against the speculative one-exchange course (default), C4's model, wall per iteration."""
import json
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kissabc_jl_amd as k  # noqa: E402
from tools.smc_c4_probe import c4_problem  # noqa: E402
prior, cost = c4_problem()
os.environ["KABC_SMC_LOOP"] = "0"
for N in [int(a) for a in sys.argv[1:]] or [32768, 131072, 524288, 2097152]:
    kw = dict(nparticles=N, alpha=0.95, epstol=0.05, seed=1, return_array=True)
    row = {"N": N}
    for spec in ("0", "1", "g1", "0", "1", "g1"):
        os.environ["KABC_SMC_SPEC_SELECT"] = "0" if spec == "0" else "1"
        os.environ.pop("KABC_DSEL2_DECIDE_G", None)
        if spec == "g1":
            os.environ["KABC_DSEL2_DECIDE_G"] = "1"
        k.smc(prior, cost, **kw)
        walls = []
        for _ in range(3):
            t0 = time.perf_counter()
            r = k.smc(prior, cost, **kw)
            walls.append(time.perf_counter() - t0)
        row.setdefault({"0": "select_kernel", "1": "one_exchange", "g1": "one_exchange_decide_on_1_workgroup"}[spec], []).append(
            round(sorted(walls)[1] * 1e6 / r.info["iterations"], 1))
        row["iterations"] = r.info["iterations"]
        if spec == "1":
            row["dist"] = r.info["dist"]
    print(json.dumps(row), flush=True)
