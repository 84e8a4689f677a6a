"""Two PROCESSES, one GPU: each rank owns its own HIP context, stream and sharded AIS
handle (kabc_ais_create_sharded, rank-offset walker ids and rows) on cuda:0; the
exchange goes over gloo through host memory because RCCL refuses two ranks on one
device.  Everything else is the production path of sharded.py.  The result must equal
the single-process trajectory bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, N, nt, gens, seed, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import kissabc_jl_amd as k
    from kissabc_jl_amd.sharded import ShardedAIS
    model = k.ApproxKernelizedPosterior(k.Factored(*[k.Uniform(-5, 5)] * 8), k.costs.Rosenbrock(), 1.0)
    sh = ShardedAIS(model, N, seed=seed, device=torch.device("cuda", 0)).init()
    sh.advance(gens, nt)
    pos = sh.positions().cpu().numpy()
    st = sh.global_stats()
    if rank == 0:
        np.savez(out_path, pos=pos, proposals=st["proposals"], accepted=st["accepted"])
    dist.barrier()
    dist.destroy_process_group()


def test_two_processes_share_one_gpu(tmp_path, k, orc, gpu_ctx):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    N, nt, gens, seed = 4096, 5, 4, 17
    out = str(tmp_path / "two.npz")
    mp.start_processes(_worker, args=(2, port, N, nt, gens, seed, out), nprocs=2, join=True,
                       start_method="spawn")
    got = np.load(out)
    model = k.ApproxKernelizedPosterior(k.Factored(*[k.Uniform(-5, 5)] * 8), k.costs.Rosenbrock(), 1.0)
    o = orc.OracleAIS(model, N, seed=seed).init()
    o.generations_sync(gens, nt, collect=False)
    assert np.array_equal(got["pos"], o.state()[0])
    assert int(got["proposals"]) == N * nt * gens and int(got["accepted"]) == o.stats()["accepted"]
