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


# ---- smc with sharded particles: the ε-selection's exchange logic -----------------------------
def _keys_of(x):
    """order-preserving map double -> u64 (csrc/smc_kernels.hpp key_of)"""
    import numpy as np
    u = np.ascontiguousarray(x, dtype=np.float64).view(np.uint64)
    neg = (u >> np.uint64(63)).astype(bool)
    return np.where(neg, ~u, u | np.uint64(1 << 63))


def _val_of(k):
    import numpy as np
    k = np.uint64(k)
    u = (k & np.uint64((1 << 63) - 1)) if (k >> np.uint64(63)) else ~k
    return float(np.array([u], dtype=np.uint64).view(np.float64)[0])


def sharded_select(X_own, alive_own, lo, N, alpha, min_r_ess, all_gather, cand_cap=4096):
    """Host mirror of the phases of csrc/smc_dsel_kernels.hpp (kabc_smc_run_dist_mode,
    KABC_SMC_DIST_PARTICLES) for ONE selection: this rank owns the particles [lo, lo + len(X_own)).
    `all_gather(array) -> [array of rank 0, array of rank 1, ...]` is the only communication
    (tests/test_sharded_gloo.py: torch.distributed over gloo).  Returns what every rank ends up with:
    (eps, flag, ESS, resample, alive_own_new, idxalive of the whole ensemble or None without a resample).
    src/smc.jl:134-147."""
    import numpy as np
    X_own = np.asarray(X_own, dtype=np.float64)
    alive_own = np.asarray(alive_own, dtype=bool)
    keys = _keys_of(X_own[alive_own])
    # begin: n, key range (the device takes them from the producers' gathered partials)
    st = all_gather(np.array([keys.size, int(keys.min()) if keys.size else 2**64 - 1,
                              int(keys.max()) if keys.size else 0], dtype=np.uint64))
    n = int(sum(int(s[0]) for s in st))
    if n == 0:
        raise ValueError("collection must be non-empty")
    klo = min(int(s[1]) for s in st)
    khi = max(int(s[2]) for s in st)
    mn = _val_of(klo)
    aleph = n * alpha + (1.0 - alpha)
    j = min(max(int(aleph), 1), n - 1) if n > 1 else 1
    gq = min(max(aleph - j, 0.0), 1.0)
    kt, nrange = j - 1, n
    state = 2 if klo == khi else (1 if n <= cand_cap else 0)
    while state == 0:     # hist + narrow: 1024 bins over the occupied key range
        span = khi - klo
        shift = max(span.bit_length() - 10, 0)
        sel = keys[(keys >= np.uint64(klo)) & (keys <= np.uint64(khi))]
        mine = np.bincount(((sel - np.uint64(klo)) >> np.uint64(shift)).astype(np.int64), minlength=1024)
        c = sum(np.asarray(h, dtype=np.int64) for h in all_gather(mine.astype(np.uint32)))
        before = np.concatenate([[0], np.cumsum(c)[:-1]])
        b = int(np.flatnonzero((c > 0) & (kt >= before) & (kt < before + c))[0])
        nlo = klo + (b << shift)
        nhi = min(nlo + (1 << shift) - 1, khi)
        klo, khi, kt, nrange = nlo, nhi, kt - int(before[b]), int(c[b])
        state = 2 if shift == 0 else (1 if nrange <= cand_cap else 0)
    need_above = False
    if state == 1:        # collect + rank
        sel = keys[(keys >= np.uint64(klo)) & (keys <= np.uint64(khi))]
        cand = np.sort(np.concatenate([np.asarray(a, dtype=np.uint64) for a in all_gather(sel)]))
        assert cand.size == nrange
        ka = int(cand[kt])
        if kt + 1 < cand.size:
            kb = int(cand[kt + 1])
        else:
            need_above = True
    else:                 # every key of the range equals klo
        ka = klo
        if kt + 1 < nrange:
            kb = klo
        else:
            need_above = True
    if need_above and n > 1:   # the smallest alive key above the range
        above = keys[keys > np.uint64(khi)]
        kb = min(int(a[0]) for a in all_gather(np.array([int(above.min()) if above.size else 2**64 - 1],
                                                        dtype=np.uint64)))
    a = _val_of(ka)
    b = a if n == 1 else _val_of(kb)
    eps = a + gq * (b - a) if (np.isfinite(a) and np.isfinite(b)) else (1.0 - gq) * a + gq * b
    flag = 0 if eps > mn else 1
    new_alive = (X_own <= eps) if flag else (X_own < eps)
    counts = [int(c[0]) for c in all_gather(np.array([int(new_alive.sum())], dtype=np.int64))]
    ESS = sum(counts)
    resample = alpha * ESS <= N * min_r_ess
    idx = None
    if resample:
        if ESS == 0:
            raise ValueError("collection must be non-empty")
        idx = np.concatenate([np.asarray(s, dtype=np.int64)
                              for s in all_gather(lo + np.flatnonzero(new_alive))])
        new_alive = np.ones_like(new_alive)
    return eps, flag, ESS, bool(resample), new_alive, idx


def sharded_select_one_exchange(X_own, alive_own, lo, N, alpha, min_r_ess, all_gather, window, X_all,
                                slot_cap=None, cand_cap=4096):
    """Host mirror of the ONE-exchange course of csrc/smc_dsel_kernels.hpp (dsel2_*): the rank ships,
    unasked, its alive keys inside `window` = (lo value, hi value) -- on the device predicted from the
    last two values of eps -- with their 1024-bin histogram, the count of its alive keys below the
    window, its smallest alive key above, and count / key range of its alive costs: ONE all_gather,
    after which every rank decides for itself.
    `X_all`: the costs of the whole ensemble (every rank holds them: the pass's all-gather delivered
    them), used for a resample's index only.  Returns sharded_select's tuple, or None where the device
    stalls (the target rank outside the window, a slot or the bin too full, eps == 0): the selection is
    then repeated phase by phase (sharded_select)."""
    import numpy as np
    X_own = np.asarray(X_own, dtype=np.float64)
    alive_own = np.asarray(alive_own, dtype=bool)
    keys = _keys_of(X_own[alive_own])
    wlo = int(_keys_of(np.array([window[0]]))[0]) if np.isfinite(window[0]) else 0
    whi = int(_keys_of(np.array([window[1]]))[0]) if np.isfinite(window[1]) else 2**64 - 1
    if wlo > whi:
        return None
    shift = max((whi - wlo).bit_length() - 10, 0)
    inside = keys[(keys >= np.uint64(wlo)) & (keys <= np.uint64(whi))]
    above = keys[keys > np.uint64(whi)]
    hist = np.bincount(((inside - np.uint64(wlo)) >> np.uint64(shift)).astype(np.int64), minlength=1024)
    H = 6
    head = np.array([int((keys < np.uint64(wlo)).sum()), inside.size,
                     int(above.min()) if above.size else 2**64 - 1,
                     keys.size, int(keys.min()) if keys.size else 2**64 - 1,
                     int(keys.max()) if keys.size else 0], dtype=np.uint64)
    parts = all_gather(np.concatenate([head, hist.astype(np.uint64), inside]))      # THE exchange
    n = sum(int(p[3]) for p in parts)
    if n == 0:
        raise ValueError("collection must be non-empty")
    kmin = min(int(p[4]) for p in parts)
    mn = _val_of(kmin)
    aleph = n * alpha + (1.0 - alpha)
    j = min(max(int(aleph), 1), n - 1) if n > 1 else 1
    gq = min(max(aleph - j, 0.0), 1.0)
    below = sum(int(p[0]) for p in parts)
    if slot_cap is not None and any(int(p[1]) > slot_cap for p in parts):
        return None
    cand_all = np.concatenate([p[H + 1024:H + 1024 + int(p[1])] for p in parts])
    kg_above = min(int(p[2]) for p in parts)
    kt = j - 1 - below
    if kt < 0 or kt >= cand_all.size:
        return None
    c = sum(np.asarray(p[H:H + 1024], dtype=np.int64) for p in parts)
    before = np.concatenate([[0], np.cumsum(c)[:-1]])
    b = int(np.flatnonzero((c > 0) & (kt >= before) & (kt < before + c))[0])
    if c[b] > cand_cap:
        return None
    blo = wlo + (b << shift)
    bhi = min(blo + (1 << shift) - 1, whi)
    cand = np.sort(cand_all[(cand_all >= np.uint64(blo)) & (cand_all <= np.uint64(bhi))])
    kt2 = kt - int(before[b])
    ka = int(cand[kt2])
    if kt2 + 1 < cand.size:
        kb = int(cand[kt2 + 1])
    else:   # the smallest alive key above the bin: among the window's keys, else above the window
        up = cand_all[cand_all > np.uint64(bhi)]
        kb = min(int(up.min()) if up.size else 2**64 - 1, kg_above)
    a = _val_of(ka)
    bb = a if n == 1 else _val_of(kb)
    eps = a + gq * (bb - a) if (np.isfinite(a) and np.isfinite(bb)) else (1.0 - gq) * a + gq * bb
    if eps == 0.0:
        return None      # (+0 and -0: two keys, one value)
    flag = 0 if eps > mn else 1
    vals = np.array([_val_of(int(kk)) for kk in cand])
    ESS = below + int(before[b]) + int(((vals <= eps) if flag else (vals < eps)).sum())
    resample = alpha * ESS <= N * min_r_ess
    new_alive = (X_own <= eps) if flag else (X_own < eps)
    idx = None
    if resample:
        if ESS == 0:
            raise ValueError("collection must be non-empty")
        X_all = np.asarray(X_all, dtype=np.float64)
        idx = np.flatnonzero((X_all <= eps) if flag else (X_all < eps))   # no exchange
        new_alive = np.ones_like(new_alive)
    return eps, flag, ESS, bool(resample), new_alive, idx
