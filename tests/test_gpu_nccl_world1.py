"""The torch.distributed variant of the walker-sharded path (sharded.ShardedAIS on
caller-lent torch buffers) on ONE GPU: with KABC_FORCE_COLLECTIVE=1 it issues its
all-gathers / barrier / all-reduce through a real "nccl" process group of world size 1.
(The library-owned exchange -- kabc_comm_*, what bench.py uses -- is in test_gpu_comm.py.)  Multi-GPU boxes are not available to the test
suite; this pins what can be pinned here -- process-group set-up on the explicit
stream, the in-place all_gather_into_tensor on the lent half buffers, ordering
against the kernels -- and checks the trajectory is still the single-process one."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys, json
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, {root!r})
import kissabc_jl_amd as k
from kissabc_jl_amd.sharded import ShardedAIS
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev, rank=0, world_size=1)
prior = k.Factored(*[k.Uniform(-5, 5)] * 4)
model = k.ApproxKernelizedPosterior(prior, k.costs.Rosenbrock(), 1.0)
sh = ShardedAIS(model, 1024, seed=11, device=dev).init()
sh.advance(5, 7)
x = sh.positions().cpu().numpy()
st = sh.global_stats()
dist.barrier()
dist.destroy_process_group()
np.save({out!r}, x)
import ctypes
ctypes.CDLL(None).fflush(None)   # RCCL's start-up banner sits in the C stdout buffer
print(json.dumps(st), flush=True)
"""


def _run_child(tmp_path, force):
    out = str(tmp_path / f"x_{force}.npy")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533",
               KABC_FORCE_COLLECTIVE="1" if force else "0")
    r = subprocess.run([sys.executable, "-c", CHILD.format(root=ROOT, out=out)], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    return np.load(out), json.loads(r.stdout.strip().splitlines()[-1])


def test_forced_collective_matches_plain(tmp_path):
    x1, s1 = _run_child(tmp_path, True)
    x0, s0 = _run_child(tmp_path, False)
    assert np.array_equal(x1, x0)
    assert s1 == s0 and s1["proposals"] == 1024 * 5 * 7
