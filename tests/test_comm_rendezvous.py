"""The file rendezvous that ships rank 0's RCCL unique id to the other ranks of a node
(kissabc_jl_amd.comm.exchange_unique_id): pure host logic, no GPU, no RCCL.

The file is named after the launcher's rendezvous variables; the record carries (pid, start time)
of the rank 0 that wrote it and is taken only while that process is alive.  So: a rank that starts
long after rank 0 published still finds the id, the ranks may each sit under their own wrapper
process, and a leftover of a killed job is never taken for the id of the next one."""
import os
import subprocess
import sys
import threading
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture
def comm(k, monkeypatch):
    from kissabc_jl_amd import comm as c
    from kissabc_jl_amd import _cdefs as cd
    monkeypatch.setattr(c, "unique_id", lambda: os.urandom(cd.KABC_COMM_ID_BYTES))
    return c


def test_ranks_receive_rank0_id(comm, tmp_path):
    got = {}

    def run(rank):
        got[rank] = comm.exchange_unique_id(rank, 4, key="t1", directory=str(tmp_path), timeout=20)

    th = [threading.Thread(target=run, args=(r,)) for r in (3, 2, 1)]
    for t in th:
        t.start()
    time.sleep(0.2)          # the readers are already polling when rank 0 writes
    run(0)
    for t in th:
        t.join()
    assert len(got[0]) == 128 and all(got[r] == got[0] for r in (1, 2, 3))


def test_late_rank_still_finds_the_id(comm, tmp_path, monkeypatch):
    """A rank that imports the package long after rank 0 published (staggered launch, slow
    import on a loaded node): the record's age does not matter."""
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("MASTER_PORT", "29517")
    uid = comm.exchange_unique_id(0, 2, directory=str(tmp_path))
    (path,) = list(tmp_path.iterdir())
    old = time.time() - 3600
    os.utime(path, (old, old))
    assert comm.exchange_unique_id(1, 2, directory=str(tmp_path), timeout=5) == uid


def test_leftover_of_an_earlier_launch_is_never_read(comm, tmp_path, monkeypatch):
    """Same rendezvous variables, an earlier launch whose rank 0 was killed (its atexit hook never
    ran): the record names a process that is gone -- or whose pid was recycled, with another start
    time -- and is ignored however fresh its mtime is; the timeout says so."""
    import struct
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("MASTER_PORT", "29518")
    stale = comm.exchange_unique_id(0, 2, directory=str(tmp_path))      # "earlier job", just now ...
    (path,) = list(tmp_path.iterdir())
    rec = path.read_bytes()
    pid, start = struct.unpack("<qq", rec[-16:])
    assert pid == os.getpid() and start > 0
    path.write_bytes(rec[:-16] + struct.pack("<qq", pid, start + 12345))   # ... by a process that is gone
    with pytest.raises(TimeoutError, match="leftover of an earlier launch"):
        comm.exchange_unique_id(1, 2, directory=str(tmp_path), timeout=0.5)
    path.write_bytes(rec[:-16] + struct.pack("<qq", 2 ** 22 + 7, start))   # no such pid
    with pytest.raises(TimeoutError, match="KABC_RDZV_KEY"):
        comm.exchange_unique_id(1, 2, directory=str(tmp_path), timeout=0.3)
    fresh = comm.exchange_unique_id(0, 2, directory=str(tmp_path))
    assert fresh != stale
    assert comm.exchange_unique_id(1, 2, directory=str(tmp_path), timeout=5) == fresh
    # ranks in separate pid namespaces cannot see rank 0's pid: the test can be switched off
    path.write_bytes(rec[:-16] + struct.pack("<qq", 2 ** 22 + 7, start))
    monkeypatch.setenv("KABC_RDZV_NO_LIVENESS", "1")
    assert comm.exchange_unique_id(1, 2, directory=str(tmp_path), timeout=5) == stale


def test_truncated_or_foreign_record_is_not_an_id(comm, tmp_path):
    path = tmp_path / f"kabc_uid_{os.getuid()}_t2.bin"
    path.write_bytes(b"\x01" * (8 + 128 + 16))          # right size, no magic: not ours
    with pytest.raises(TimeoutError):
        comm.exchange_unique_id(1, 2, key="t2", directory=str(tmp_path), timeout=0.3)
    fresh = comm.exchange_unique_id(0, 2, key="t2", directory=str(tmp_path))   # rank 0 replaces it
    assert comm.exchange_unique_id(1, 2, key="t2", directory=str(tmp_path), timeout=5) == fresh


def test_no_launcher_variables_no_guess(comm, tmp_path, monkeypatch):
    for v in ("MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT",
              "KABC_RDZV_KEY"):
        monkeypatch.delenv(v, raising=False)
    with pytest.raises(RuntimeError, match="KABC_RDZV_KEY"):
        comm.exchange_unique_id(1, 2, directory=str(tmp_path), timeout=0.2)
    monkeypatch.setenv("KABC_RDZV_KEY", "slurm job 77/step 0")
    uid = comm.exchange_unique_id(0, 2, directory=str(tmp_path))
    assert comm.exchange_unique_id(1, 2, directory=str(tmp_path), timeout=5) == uid


def test_world_one_needs_no_file(comm, tmp_path):
    assert len(comm.exchange_unique_id(0, 1, key="t3", directory=str(tmp_path))) == 128
    assert not list(tmp_path.iterdir())


_CHILD = r"""
import os, sys, time
sys.path.insert(0, {root!r})
os.environ["KABC_NO_TORCH_PRELOAD"] = "1"
rank, delay = int(sys.argv[1]), float(sys.argv[2])
time.sleep(delay)
from kissabc_jl_amd import comm
uid = comm.exchange_unique_id(rank, 2, directory={d!r}, timeout=30,
                              make_id=lambda: os.urandom(128))
print("UID", uid.hex(), flush=True)
if rank == 0:
    time.sleep(float(sys.argv[3]))      # stay alive: the record disappears when rank 0 exits
"""


@pytest.mark.parametrize("wrapped", [False, True], ids=["direct-children", "per-rank-wrapper-shells"])
def test_two_processes_of_one_launch(k, tmp_path, wrapped):
    """Two real processes derive the same default key from MASTER_*; rank 1 starts 6 s after rank 0
    has published (an mtime rule rejected exactly this case and then timed out).  `wrapped`: each
    rank runs under its OWN non-exec wrapper shell (torchrun --no-python wrapper.sh, mpirun
    bash -c, per-rank ssh) -- the ranks then have different parent processes, which a key built
    from the parent's pid cannot survive."""
    script = tmp_path / "child.py"
    script.write_text(_CHILD.format(root=ROOT, d=str(tmp_path)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29519" if not wrapped else "29520")
    env.pop("KABC_RDZV_KEY", None)

    def cmd(*a):
        base = [sys.executable, str(script), *a]
        return ["bash", "-c", " ".join(base) + "; exit $?"] if wrapped else base
    p0 = subprocess.Popen(cmd("0", "0", "9"), env=env, stdout=subprocess.PIPE, text=True)
    p1 = subprocess.Popen(cmd("1", "6", "0"), env=env, stdout=subprocess.PIPE, text=True)
    o1, _ = p1.communicate(timeout=120)
    o0, _ = p0.communicate(timeout=120)
    assert p0.returncode == 0 and p1.returncode == 0
    u0 = [ln for ln in o0.splitlines() if ln.startswith("UID")][0]
    u1 = [ln for ln in o1.splitlines() if ln.startswith("UID")][0]
    assert u0 == u1 and len(u0.split()[1]) == 256
    assert not [p for p in tmp_path.iterdir() if p.name.startswith("kabc_uid_")]   # cleaned up
