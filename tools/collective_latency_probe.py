"""What ONE small collective costs through the library at world size 1 (RCCL loaded, one GPU): the floor
any distributed epsilon-selection of smc would pay several times per iteration (all-reduce of the
histograms, candidate gather, index gather), against the redundant selection every rank of
kabc_smc_run_dist runs today on data it holds anyway (31 us at 131 072 particles, 83 us at 2 M:
profiles/r04_smc_large.txt).  Host-synchronous all-reduce of 8 bytes (kabc_comm_allreduce_sum_u64:
H2D, ncclAllReduce, D2H, stream sync) and the device-side time of the 2 MiB all-gather of an AIS
half-generation (kabc_ais_exchange_us)."""
import json
import os
import sys
import time

os.environ["KABC_FORCE_COLLECTIVE"] = "1"
os.environ.setdefault("MASTER_PORT", "29533")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import kissabc_jl_amd as k  # noqa: E402
from kissabc_jl_amd.comm import Comm  # noqa: E402

comm = Comm.from_env(device=0)
for _ in range(50):
    comm.barrier()
t0 = time.perf_counter()
n = 2000
for _ in range(n):
    comm.barrier()
host_us = (time.perf_counter() - t0) / n * 1e6
ens = k.AisEnsemble(bench.build_model(k), 65536, seed=1, ctx=comm.ctx, comm=comm).init()
ens.advance(5, 16)
ens.set_timing(64, stride=1)
ens.advance(32, 16)
x = ens.exchange_us()
print(json.dumps({"world": 1, "host_synchronous_allreduce_8B_us": host_us,
                  "allgather_2MiB_device_us": x["exchange_us_per_half"],
                  "kernels_us_per_half_nt16": x["compute_us_per_half"], "chunks": x["chunks"]}))
