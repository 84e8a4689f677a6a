"""kabc_smc_run_dist_mode on a one-process-per-GPU RCCL communicator of world size 1 (all this pool offers):
wall per eps-iteration of C4's model for kabc_smc_run, the cost-loop mode and the particle-sharded mode.
At world 1 a mode's cost is its structure (launches, collectives, looks at the control block) -- the floor a
multi-GPU run starts from, not a scaling figure.  KABC_SMC_DIST_LOOKS=1: the round-5 course (a look per pass and
per selection phase, four to five small all-gathers per iteration)."""
import json
import os
import sys
import time

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29541")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
os.environ.setdefault("LOCAL_RANK", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kissabc_jl_amd as k  # noqa: E402
from tools.smc_c4_probe import c4_problem  # noqa: E402

prior, cost = c4_problem()
comm = k.Comm.from_env()
for N in [int(a) for a in sys.argv[1:]] or [131072, 2097152]:
    kw = dict(nparticles=N, alpha=0.95, epstol=0.05, seed=1, return_array=True)
    row = {"N": N}
    for mode in (None, "cost_loop", "particles"):
        extra = {} if mode is None else dict(comm=comm, shard=mode)
        k.smc(prior, cost, **kw, **extra)
        walls = []
        for _ in range(3):
            t0 = time.perf_counter()
            r = k.smc(prior, cost, **kw, **extra)
            walls.append(time.perf_counter() - t0)
        row[mode or "kabc_smc_run"] = {"us_per_iteration": round(sorted(walls)[1] * 1e6 / r.info["iterations"], 1),
                                       "iterations": r.info["iterations"], "eps": r.eps}
        if mode is not None:
            d = r.info["dist"]
            row[mode].update(collectives_per_iteration=d["collectives_per_iteration"], host_looks=d["host_looks"],
                             one_exchange_selections=d["one_exchange_selections"],
                             phase_by_phase_selections=d["phase_by_phase_selections"], batched=d["batched"])
    print(json.dumps(row), flush=True)
comm.close()
