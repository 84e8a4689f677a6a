"""-m gpu: determinism when several processes share the GPU (tools/contention_stress.py).  Two
processes repeat C4 `smc` on the persistent loop kernel for a few seconds beside one that keeps the
memory system busy with AIS generations; every result must equal the process's first.  A short
run only catches a gross regression of the loop kernel's device-wide barrier -- the race this
round's XCD-aware barrier had needed ~10^4 contended calls to show (profiles/r03_contention.txt) --
but it keeps the tool, and the multi-process set-up it needs, exercised."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_smc_loop_kernel_is_deterministic_under_gpu_sharing(gpu_ctx):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "contention_stress.py"), "smc_loop", "2", "ais", "6"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "DIFFERING" not in r.stdout, r.stdout + r.stderr
    assert r.stdout.count("all identical") == 3
