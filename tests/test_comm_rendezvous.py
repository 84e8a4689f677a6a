"""The file rendezvous that ships rank 0's RCCL unique id to the other ranks of a node
(kissabc_jl_amd.comm.exchange_unique_id): pure host logic, no GPU, no RCCL."""
import os
import threading
import time

import pytest


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


def test_stale_id_of_an_earlier_job_is_ignored(comm, tmp_path):
    path = tmp_path / f"kabc_uid_{os.getuid()}_t2.bin"
    path.write_bytes(b"\x01" * 128)
    old = time.time() - 3600
    os.utime(path, (old, old))
    with pytest.raises(TimeoutError):
        comm.exchange_unique_id(1, 2, key="t2", directory=str(tmp_path), timeout=0.5)
    fresh = comm.exchange_unique_id(0, 2, key="t2", directory=str(tmp_path))
    assert comm.exchange_unique_id(1, 2, key="t2", directory=str(tmp_path), timeout=5) == fresh


def test_world_one_needs_no_file(comm, tmp_path):
    assert len(comm.exchange_unique_id(0, 1, key="t3", directory=str(tmp_path))) == 128
    assert not list(tmp_path.iterdir())
