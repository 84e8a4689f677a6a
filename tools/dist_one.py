"""one sharded smc run of C4's model on a communicator of world 1: python tools/dist_one.py N mode [reps]
(RCCL by default; DIST_ONE_P2P=1: the P2P backend -- rocprofv3 crashes at exit with RCCL loaded on this image)"""
import os
import sys
import time
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29543")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
os.environ.setdefault("LOCAL_RANK", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kissabc_jl_amd as k  # noqa: E402
from tools.smc_c4_probe import c4_problem  # noqa: E402
prior, cost = c4_problem()
N, mode = int(sys.argv[1]), sys.argv[2]
comm = k.comm.init_all([0], "p2p")[0] if os.environ.get("DIST_ONE_P2P") else k.Comm.from_env()
kw = dict(nparticles=N, alpha=0.95, epstol=0.05, seed=1, return_array=True)
extra = {} if mode == "plain" else dict(comm=comm, shard=mode)
for _ in range(int(sys.argv[3]) if len(sys.argv) > 3 else 2):
    t0 = time.perf_counter()
    r = k.smc(prior, cost, **kw, **extra)
    print(mode, N, round((time.perf_counter() - t0) * 1e6 / r.info["iterations"], 1), "us per iteration", r.info.get("dist"), flush=True)
comm.close()
