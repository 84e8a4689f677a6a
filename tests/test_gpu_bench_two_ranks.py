"""bench.py's N > 1 branch end to end on a 1-GPU box: launched exactly as the driver
does (python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2), with
both ranks on device 0 and the exchange over gloo (KABC_BENCH_BACKEND / KABC_BENCH_DEVICE;
RCCL refuses two ranks on one device).  Checks the single JSON line of rank 0."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_one_gpu():
    env = dict(os.environ, KABC_BENCH_BACKEND="gloo", KABC_BENCH_DEVICE="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29571", os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "4", "--warmup", "1"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["walkers_total"] == 131072 and d["scaling"] == "weak"
    assert d["value"] > 0 and d["roofline"]["frac"] > 0 and "cpu_baseline" not in d
    assert d["config"]["ntransitions"] == 100 and "also_at_ntransitions_16" in d
    assert r.stdout.strip().splitlines()[-1] == lines[0]     # the JSON is the last line
