"""Communicators of the walker-sharded path: ctypes mirror of the `kabc_comm_*` /
`kabc_ais_*_multi` entry points of include/kabc.h.  The collective (RCCL all-gather,
or the P2P pull kernel) is issued INSIDE the library; nothing here needs torch.

The reference has no counterpart (its MCMCThreads / MCMCDistributed run independent
chains, src/KissABC.jl:9,108-109,175).

    one process per GPU (torchrun / mpirun / Distributed.jl style launch):
        comm = Comm.from_env()                   # RANK / WORLD_SIZE / LOCAL_RANK
        ens = AisEnsemble(model, N, seed=1, comm=comm).init()
        ens.advance(gens, ntransitions)          # kernels + ncclAllGather per half

    one process, several GPUs:
        grp = EnsembleGroup(model, N, seed=1, devices=[0, 1, 2, 3])   # RCCL, or backend="p2p"
        grp.init(); grp.advance(gens, ntransitions)
"""
import ctypes as C
import os
import time

import numpy as np

from . import _cdefs as cd
from . import _lib
from .api import AisEnsemble


def unique_id():
    buf = (C.c_uint8 * cd.KABC_COMM_ID_BYTES)()
    _lib.check(_lib.load().kabc_comm_unique_id(buf))
    return bytes(buf)


_UID_MAGIC = b"KABCUID2"


def _proc_start(pid):
    """kernel start time (clock ticks since boot) of a live process, or None"""
    try:
        with open(f"/proc/{pid}/stat", "rb") as f:
            st = f.read()
        return int(st[st.rindex(b")") + 2:].split()[19])   # field 22: starttime
    except (OSError, ValueError, IndexError):
        return None


def _publisher_alive(pid, start):
    """Is the rank 0 that wrote a record still running?  A record outlives its launch only when
    rank 0 was killed (its atexit hook never ran); (pid, start time) of a dead process never
    matches a live one, a recycled pid has another start time.  KABC_RDZV_NO_LIVENESS=1 skips the
    test (ranks in separate pid namespaces that share the rendezvous directory)."""
    if os.environ.get("KABC_RDZV_NO_LIVENESS") == "1" or not os.path.isdir("/proc/self"):
        return True
    return _proc_start(pid) == start


def rendezvous_key(world):
    """Default name of the rendezvous file: KABC_RDZV_KEY (any string the host guarantees unique
    per launch, e.g. a scheduler job id) or the launcher's rendezvous variables -- the same for
    every rank of a launch WHATEVER process started it (torchrun's agent, a per-rank wrapper
    shell, mpirun, ssh).  Two concurrent launches on one node cannot share MASTER_PORT; a leftover
    of an EARLIER launch with the same variables is told apart by the record itself (the
    publishing rank 0 must be alive: exchange_unique_id).  Without any of the variables there is
    nothing that tells two concurrent jobs of one user apart, and the id is refused rather than
    guessed."""
    clean = lambda v: "".join(ch if ch.isalnum() else "-" for ch in str(v))   # noqa: E731
    explicit = os.environ.get("KABC_RDZV_KEY")
    if explicit:
        return f"{clean(explicit)}_w{world}"
    env = [os.environ.get(v, "") for v in ("MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID",
                                           "TORCHELASTIC_RESTART_COUNT")]
    if not (env[1] or env[2]):
        raise RuntimeError(
            "cannot name the unique-id rendezvous: neither MASTER_PORT nor TORCHELASTIC_RUN_ID is "
            "set.  Launch the ranks with torchrun / `python -m torch.distributed.run`, or set "
            "KABC_RDZV_KEY to a string unique to this launch (and identical on all its ranks), or "
            "pass the id yourself: Comm.init_rank(uid, rank, world)")
    return f"{clean('_'.join(env))}_w{world}"


def exchange_unique_id(rank, world, key=None, directory=None, timeout=300.0, make_id=None):
    """Ship rank 0's RCCL unique id to the other ranks of ONE node through a file
    (the id is 128 opaque bytes; any channel the host owns would do -- a Julia host
    would use Distributed or MPI.bcast).  The file is named after `key` (default:
    rendezvous_key); the record holds the id and (pid, start time) of the rank 0 that wrote it.
    A reader takes a record only while that process is alive, so neither the record's age nor
    the shape of the launcher matters: a rank may start minutes after rank 0 published, each
    rank may sit under its own wrapper process, and the leftover of a killed job is never taken
    for the id of the next one.  Rank 0 replaces whatever is at the path atomically and removes
    the file when it exits."""
    import struct
    make_id = make_id or unique_id
    if world == 1:
        return make_id()
    directory = directory or os.environ.get("KABC_RDZV_DIR") or "/tmp"
    if key is None:
        key = rendezvous_key(world)
    path = os.path.join(directory, f"kabc_uid_{os.getuid()}_{key}.bin")
    if rank == 0:
        uid = make_id()
        tmp = f"{path}.{os.getpid()}.tmp"
        with open(tmp, "wb") as f:
            f.write(_UID_MAGIC + uid + struct.pack("<qq", os.getpid(), _proc_start(os.getpid()) or 0))
        os.replace(tmp, path)   # atomic: readers see the whole record or the previous file
        import atexit

        def _cleanup(p=path):    # nothing of this launch outlives it
            try:
                os.remove(p)
            except OSError:
                pass
        atexit.register(_cleanup)
        return uid
    t0 = time.time()
    seen_stale = False
    nrec = len(_UID_MAGIC) + cd.KABC_COMM_ID_BYTES + 16
    while True:
        try:
            with open(path, "rb") as f:
                rec = f.read()
            if len(rec) == nrec and rec.startswith(_UID_MAGIC):
                pid, start = struct.unpack("<qq", rec[-16:])
                if _publisher_alive(pid, start):
                    return rec[len(_UID_MAGIC):-16]
                seen_stale = True
        except OSError:
            pass
        if time.time() - t0 > timeout:
            raise TimeoutError(
                f"rank {rank}: no unique id at {path} after {timeout:.0f}s"
                + (" (a record is there, but the rank 0 that wrote it is gone: the leftover of an earlier launch)"
                   if seen_stale else "")
                + ".  Every rank of a launch must derive the same path: identical MASTER_ADDR / MASTER_PORT / "
                  "TORCHELASTIC_RUN_ID, or KABC_RDZV_KEY=<string unique to this launch>, and a shared "
                  "KABC_RDZV_DIR (default /tmp); ranks in separate pid namespaces: KABC_RDZV_NO_LIVENESS=1")
        time.sleep(0.02)


class Comm:
    """kabc_comm_t"""

    def __init__(self, handle, ctx, owner=True):
        self._h = C.c_void_p(handle)
        self.ctx = ctx
        self._owner = owner
        lib = _lib.load()
        self.rank = lib.kabc_comm_rank(self._h)
        self.world = lib.kabc_comm_world(self._h)

    @property
    def handle(self):
        return self._h

    @classmethod
    def init_rank(cls, uid, rank, world, device=0, ctx=None):
        ctx = ctx or _lib.Context(device)
        h = C.c_void_p()
        buf = (C.c_uint8 * cd.KABC_COMM_ID_BYTES).from_buffer_copy(uid)
        _lib.check(_lib.load().kabc_comm_init_rank(ctx.handle, buf, int(rank), int(world),
                                                   C.byref(h)))
        return cls(h.value, ctx)

    @classmethod
    def from_env(cls, device=None):
        """One process per GPU under torchrun-style variables; the unique id travels
        through exchange_unique_id."""
        rank = int(os.environ.get("RANK", "0"))
        world = int(os.environ.get("WORLD_SIZE", "1"))
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0"))
        return cls.init_rank(exchange_unique_id(rank, world), rank, world, device)

    def allreduce_sum(self, values):
        a = (C.c_uint64 * len(values))(*[int(v) for v in values])
        _lib.check(_lib.load().kabc_comm_allreduce_sum_u64(self._h, a, len(values)))
        return list(a)

    def allreduce_max(self, values):
        a = (C.c_double * len(values))(*[float(v) for v in values])
        _lib.check(_lib.load().kabc_comm_allreduce_max_f64(self._h, a, len(values)))
        return list(a)

    def barrier(self):
        _lib.check(_lib.load().kabc_comm_barrier(self._h))

    def close(self):
        if self._h and self._owner:
            _lib.load().kabc_comm_destroy(self._h)
        self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


BACKENDS = {"rccl": cd.COMM_RCCL, "p2p": cd.COMM_P2P}


def init_all(devices, backend="rccl"):
    """kabc_comm_init_all: one context + one communicator per entry of `devices`."""
    n = len(devices)
    devs = (C.c_int32 * n)(*[int(d) for d in devices])
    ctxs = (C.c_void_p * n)()
    comms = (C.c_void_p * n)()
    _lib.check(_lib.load().kabc_comm_init_all(n, devs, BACKENDS[backend], ctxs, comms))
    return [Comm(comms[i], _lib.Context(devices[i], _borrowed=ctxs[i])) for i in range(n)]


class EnsembleGroup:
    """The n shards of one AIS ensemble driven from ONE process
    (kabc_ais_init_multi / kabc_ais_advance_multi)."""

    def __init__(self, model, nparticles, seed=0, devices=(0,), backend="rccl"):
        self.comms = init_all(list(devices), backend)
        self.shards = [AisEnsemble(model, nparticles, seed=seed, comm=c) for c in self.comms]
        self.N, self.D = int(nparticles), len(model)
        self._hs = (C.c_void_p * len(self.shards))(*[s._h.value for s in self.shards])

    def init(self, retry_sampling=100):
        _lib.check(_lib.load().kabc_ais_init_multi(self._hs, len(self.shards), int(retry_sampling)))
        return self

    def advance(self, ngenerations, ntransitions=1):
        st = cd.Stats()
        _lib.check(_lib.load().kabc_ais_advance_multi(self._hs, len(self.shards),
                                                      int(ngenerations), int(ntransitions),
                                                      C.byref(st)))
        self.last_stats = {"proposals": st.proposals, "cost_evals": st.cost_evals,
                           "accepted": st.accepted}
        return self

    def ensemble(self, rank=0):
        return self.shards[rank].ensemble()

    def state(self):
        """(x, logprior, loglik) of all walkers in walker-id order, assembled from the
        owners (log-densities live with the owner only)."""
        n0 = (self.N + 1) // 2
        x = np.empty((self.N, self.D))
        lp = np.empty(self.N)
        ll = np.empty(self.N)
        lib = _lib.load()
        for s in self.shards:
            xs, lps, lls, _ = s.state()
            src = 0
            for half, base in ((0, 0), (1, n0)):
                for first, count in s.segments(half):
                    dst = base + first
                    x[dst:dst + count] = xs[src:src + count]
                    lp[dst:dst + count] = lps[src:src + count]
                    ll[dst:dst + count] = lls[src:src + count]
                    src += count
        return x, lp, ll

    def stats(self):
        tot = {"proposals": 0, "cost_evals": 0, "accepted": 0}
        for s in self.shards:
            for kk, v in s.stats().items():
                tot[kk] += v
        return tot

    def synchronize(self):
        for c in self.comms:
            c.ctx.synchronize()

    def close(self):
        for s in self.shards:
            s.close()
        for c in self.comms:
            c.close()
        self.shards, self.comms = [], []
