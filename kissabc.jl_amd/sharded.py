"""Multi-GPU AIS: one process per GPU, walkers sharded by row range, ONE
all-gather (RCCL over xGMI via torch.distributed, backend "nccl") per
half-generation to rebuild the complementary ensemble.

Why this is the only collective: a walker of the active half reads partners from
the *frozen* complementary half only (include/kabc.h "Schedule"), so ranks are
independent inside a half-generation; afterwards every rank needs the other
ranks' freshly updated rows of that half before the next half-generation reads
them as partners.  Log-densities never travel.  Draws are keyed by global walker
id, so the trajectory is identical for every world size (tests/test_sharded_gloo.py).

The reference has no multi-device path (its MCMCThreads/MCMCDistributed run
independent chains, src/KissABC.jl:108-109); this sharding is new.

torch is plumbing here: device buffers for the two global halves (so that the
collective and the kernels see the same memory), the current HIP stream, and
torch.distributed.
"""
import contextlib
import os

import torch
import torch.distributed as dist

from . import _lib
from .api import AisEnsemble


class HipEngine:
    """Per-rank compute: the gfx950 kernels updating this rank's rows in place
    inside torch-owned global half buffers.

    Stream discipline: the engine owns ONE explicit (non-default) torch stream; the
    kernels are enqueued on it through the C ABI and the collectives are issued
    while it is torch's current stream, so c10d orders RCCL against the kernels.
    (torch's default stream has handle 0, which the C ABI reads as "create a
    private stream" -- never pass it.)"""

    def __init__(self, model, n_total, seed, rank, world, device, half_buffers=None):
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        D = len(model)
        n0, n1 = (n_total + 1) // 2, n_total // 2
        self.stream = torch.cuda.Stream(self.device)
        with torch.cuda.stream(self.stream):
            self.half = half_buffers or [
                torch.zeros((n0, D), dtype=torch.float64, device=self.device),
                torch.zeros((n1, D), dtype=torch.float64, device=self.device)]
        self.ctx = _lib.Context(self.device.index or 0, self.stream.cuda_stream)
        self.ens = AisEnsemble(model, n_total, seed=seed, ctx=self.ctx,
                               sharded=(rank, world, self.half[0].data_ptr(),
                                        self.half[1].data_ptr()))
        self.rows = [(rank * (n0 // world), (rank + 1) * (n0 // world)),
                     (rank * (n1 // world), (rank + 1) * (n1 // world))]

    def init(self, retry_sampling):
        self.ens.init(retry_sampling)

    def half_generation(self, half, ntransitions):
        self.ens.half_generation(half, ntransitions)

    def end_generation(self, ntransitions):
        self.ens.end_generation(ntransitions)

    def stats(self):
        return self.ens.stats()

    def synchronize(self):
        self.stream.synchronize()

    # RCCL's in-place all-gather: the send buffer is this rank's slice of the receive
    # buffer (offset rank * count), the layout the half buffers already have -- no
    # staging copy.  (torch's own FSDP issues all_gather_into_tensor the same way.)
    inplace_gather = True


class ShardedAIS:
    """AIS(N) over `world` ranks.  `engine` is the per-rank compute object
    (HipEngine in production; the tests inject a CPU stand-in to exercise the
    exchange logic under gloo)."""

    def __init__(self, model, n_total, seed=0, device=None, group=None, engine=None):
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.group = group
        if n_total % (2 * self.world):
            raise ValueError("nparticles must be divisible by 2*world_size")
        self.N, self.D = int(n_total), len(model)
        self.engine = engine or HipEngine(model, n_total, seed, self.rank, self.world, device)

    def _stream_ctx(self):
        st = getattr(self.engine, "stream", None)
        return torch.cuda.stream(st) if st is not None else contextlib.nullcontext()

    def _gather(self, half):
        # KABC_FORCE_COLLECTIVE=1 issues the collective at world size 1 too, so that the
        # RCCL path can be exercised on a single-GPU box (tests/test_gpu_nccl_world1.py)
        if self.world == 1 and not (dist.is_initialized() and
                                    os.environ.get("KABC_FORCE_COLLECTIVE") == "1"):
            return
        with self._stream_ctx():
            buf = self.engine.half[half]
            lo, hi = self.engine.rows[half]
            own = buf[lo:hi]
            if buf.is_cuda and dist.get_backend(self.group) == "gloo":
                # test configuration only (several ranks sharing one GPU, where RCCL refuses
                # duplicate devices): exchange through host memory
                out = torch.empty(buf.shape, dtype=buf.dtype)
                dist.all_gather_into_tensor(out, own.cpu(), group=self.group)
                buf.copy_(out)
                return
            if not getattr(self.engine, "inplace_gather", False):
                own = own.clone()
            dist.all_gather_into_tensor(buf, own, group=self.group)

    def init(self, retry_sampling=100):
        self.engine.init(retry_sampling)
        self._gather(0)
        self._gather(1)
        return self

    def generation(self, ntransitions=1):
        for half in (0, 1):
            self.engine.half_generation(half, ntransitions)
            self._gather(half)
        self.engine.end_generation(ntransitions)

    def advance(self, ngenerations, ntransitions=1):
        for _ in range(int(ngenerations)):
            self.generation(ntransitions)

    def positions(self):
        """[N][D] positions in walker-id order (identical on every rank)."""
        with self._stream_ctx():
            out = torch.cat([self.engine.half[0], self.engine.half[1]], dim=0)
        self.synchronize()
        return out

    def synchronize(self):
        sync = getattr(self.engine, "synchronize", None)
        if sync:
            sync()

    def global_stats(self):
        st = self.engine.stats()
        with self._stream_ctx():
            dev = self.engine.half[0].device
            if dev.type == "cuda" and dist.is_initialized() and dist.get_backend(self.group) == "gloo":
                dev = torch.device("cpu")
            t = torch.tensor([st["proposals"], st["cost_evals"], st["accepted"]],
                             dtype=torch.int64, device=dev)
            if self.world > 1:
                dist.all_reduce(t, group=self.group)
        return dict(zip(("proposals", "cost_evals", "accepted"), (int(v) for v in t.tolist())))
