"""`python bench.py --gpus N` without a launcher starts its N ranks itself (bench.py spawn_ranks):
the spawn / relay / failure logic with stub ranks (KABC_BENCH_STUB_RANK: no GPU, no library)."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(n, stub, extra=()):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["KABC_BENCH_STUB_RANK"] = stub
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "3",
                        "--warmup", "1", *extra], env=env, capture_output=True, text=True, timeout=120)
    return p, time.time() - t0


def test_spawned_ranks_get_launcher_variables_and_rank0_line_is_last():
    p, _ = _run(4, "ok")
    assert p.returncode == 0, p.stderr
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert lines[0] == "stub: a line that is not the result"
    d = json.loads(lines[-1])
    assert d["n_gpus"] == 4 and d["steps"] == 3 and d["warmup"] == 1
    assert d["ranks_env"] == ["0", "0", "4", "127.0.0.1"]


def test_a_failing_rank_fails_the_launch_and_ends_the_others():
    p, el = _run(3, "fail2")
    assert p.returncode == 7
    assert "rank 2 exited with code 7" in p.stderr
    assert not any(ln.lstrip().startswith("{") for ln in p.stdout.splitlines())   # no result line


def test_a_hanging_peer_is_ended_when_another_rank_fails():
    # ranks 0 and 2 hang (ranks waiting in a collective for a peer that died), rank 1 fails
    p, el = _run(3, "hang0,fail1,hang2")
    assert p.returncode == 7 and el < 60
    assert not any(ln.lstrip().startswith("{") for ln in p.stdout.splitlines())


def test_external_launcher_is_respected():
    """under torchrun-style variables bench.py does not spawn (it IS a rank)"""
    env = dict(os.environ, WORLD_SIZE="2", RANK="1", LOCAL_RANK="1", KABC_BENCH_STUB_RANK="ok")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env,
                       capture_output=True, text=True, timeout=60)
    assert p.returncode == 0 and p.stdout.strip() == ""   # rank 1 prints nothing
