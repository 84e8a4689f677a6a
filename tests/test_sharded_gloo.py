"""CPU, world_size = 2 and 4 over gloo: the multi-GPU driver's exchange logic
(kissabc_jl_amd.sharded.ShardedAIS: row ownership, one all-gather per
half-generation, counter stride) with the CPU oracle standing in for the
per-rank HIP engine.  The sharded trajectory must equal the single-process
trajectory BIT FOR BIT -- draws are keyed by global walker id."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleEngine:
    """Test stand-in for sharded.HipEngine: same interface, CPU oracle compute."""
    inplace_gather = False

    def __init__(self, model, n_total, seed, rank, world):
        from oracle import oracle as orc
        self.o = orc.OracleAIS(model, n_total, seed=seed)
        self.N, self.D = n_total, len(model)
        n0, n1 = (n_total + 1) // 2, n_total // 2
        self.n0 = n0
        self.half = [torch.zeros((n0, self.D), dtype=torch.float64),
                     torch.zeros((n1, self.D), dtype=torch.float64)]
        self.rows = [(rank * (n0 // world), (rank + 1) * (n0 // world)),
                     (rank * (n1 // world), (rank + 1) * (n1 // world))]

    def _pull(self):     # oracle state -> the torch half buffers (own rows only matter)
        x, self.lp, self.ll, self.t = self.o.state()
        for h, off in ((0, 0), (1, self.n0)):
            lo, hi = self.rows[h]
            self.half[h][lo:hi] = torch.from_numpy(x[off + lo:off + hi])

    def _push(self):     # gathered torch halves -> oracle positions
        x = torch.cat(self.half, 0).numpy()
        _, lp, ll, t = self.o.state()
        self.o.set_state(x, lp, ll, t)

    def init(self, retry_sampling):
        self.o.init(retry_sampling)   # every rank draws all walkers identically
        self._pull()

    def half_generation(self, half, nt):
        self._push()
        lo, hi = self.rows[half]
        self.o.half_generation(half, nt, lo, hi)
        self._pull()

    def end_generation(self, nt):
        self.o.end_generation(nt)

    def stats(self):
        return self.o.stats()


def _model(k):
    U = k.Factored(*[k.Uniform(-5, 5)] * 8)
    return k.ApproxKernelizedPosterior(U, k.costs.Rosenbrock(), 1.0)


def _worker(rank, world, port, N, nt, gens, seed, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import kissabc_jl_amd as k
    from kissabc_jl_amd.sharded import ShardedAIS
    model = _model(k)
    eng = OracleEngine(model, N, seed, rank, world)
    sh = ShardedAIS(model, N, seed=seed, engine=eng).init()
    sh.advance(gens, nt)
    pos = sh.positions().numpy()
    # log-densities live on the owner only: gather them for the comparison
    lp = torch.zeros(N, dtype=torch.float64)
    n0 = (N + 1) // 2
    for h, off in ((0, 0), (1, n0)):
        lo, hi = eng.rows[h]
        lp[off + lo:off + hi] = torch.from_numpy(eng.lp[off + lo:off + hi])
    dist.all_reduce(lp)
    st = sh.global_stats()
    if rank == 0:
        np.savez(out_path, pos=pos, lp=lp.numpy(), proposals=st["proposals"],
                 accepted=st["accepted"])
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_equals_single_process(tmp_path, orc, k, world):
    N, nt, gens, seed = 512, 3, 4, 21
    out = str(tmp_path / "sharded.npz")
    mp.spawn(_worker, args=(world, _free_port(), N, nt, gens, seed, out), nprocs=world, join=True)
    got = np.load(out)
    o = orc.OracleAIS(_model(k), N, seed=seed).init()
    o.generations_sync(gens, nt, collect=False)
    x, lp, ll, t = o.state()
    assert np.array_equal(got["pos"], x)
    assert np.array_equal(got["lp"], lp)
    st = o.stats()
    # every rank runs init for all walkers in the stand-in, but only its rows afterwards
    assert int(got["proposals"]) == st["proposals"] == N * nt * gens
    assert int(got["accepted"]) == st["accepted"]


def test_sharded_requires_divisible_ensemble(k):
    from kissabc_jl_amd.sharded import ShardedAIS

    class E:
        half = rows = None
    with pytest.raises(ValueError):
        # single process => world 1 => needs N % 2 == 0
        ShardedAIS(_model(k), 513, engine=E())


# ---- smc with sharded particles: the selection's exchange logic (sharded.sharded_select) ----------
def _select_direct(X, alive, alpha, min_r_ess):
    """src/smc.jl:134-147 on the whole ensemble (the formulas of csrc/smc_kernels.hpp smc_select_kernel)"""
    xs = np.sort(X[alive])
    n = xs.size
    aleph = n * alpha + (1.0 - alpha)
    j = min(max(int(aleph), 1), n - 1) if n > 1 else 1
    g = min(max(aleph - j, 0.0), 1.0)
    a, b = xs[j - 1], (xs[j] if n > 1 else xs[j - 1])
    eps = a + g * (b - a) if (np.isfinite(a) and np.isfinite(b)) else (1.0 - g) * a + g * b
    flag = 0 if eps > xs[0] else 1
    new = (X <= eps) if flag else (X < eps)
    ess = int(new.sum())
    res = alpha * ess <= X.size * min_r_ess
    return eps, flag, ess, bool(res), (np.ones_like(new) if res else new), (np.flatnonzero(new) if res else None)


def _select_cases():
    rng = np.random.default_rng(12)
    N = 20000
    out = {}
    x = rng.normal(size=N) ** 2
    out["smooth"] = (x, np.ones(N, bool), 0.95, 0.2)
    out["resample"] = (x, rng.random(N) < 0.5, 0.6, 0.9)
    t = np.round(rng.normal(size=N) * 3)          # heavy ties: whole ranges collapse to one key
    out["ties"] = (t, rng.random(N) < 0.8, 0.9, 0.2)
    out["all_equal"] = (np.full(N, 2.5), np.ones(N, bool), 0.5, 0.2)
    c = x.copy()
    c[rng.random(N) < 0.3] = np.inf               # KernelizedPosterior-style infinite costs
    out["with_inf"] = (c, np.ones(N, bool), 0.95, 0.2)
    s = np.concatenate([rng.normal(size=N - 3), [1e300, -1e300, 0.0]])   # 64-bit key range in use
    out["wide"] = (s, np.ones(N, bool), 0.999, 0.2)
    out["small"] = (rng.normal(size=300), rng.random(300) < 0.7, 0.9, 0.5)
    return out


def _select_worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from kissabc_jl_amd.sharded import sharded_select

    def all_gather(a):      # ragged: sizes first, then padded payloads (as the device pads its segments)
        a = np.ascontiguousarray(a)
        sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(sizes, torch.tensor([a.size]))
        m = max(int(s) for s in sizes)
        buf = np.zeros(max(m, 1), dtype=a.dtype)
        buf[:a.size] = a
        parts = [torch.zeros(max(m, 1) * a.dtype.itemsize, dtype=torch.uint8) for _ in range(world)]
        dist.all_gather(parts, torch.from_numpy(buf.view(np.uint8).copy()))
        return [p.numpy().view(a.dtype)[:int(s)] for p, s in zip(parts, sizes)]

    res = {}
    for name, (X, alive, alpha, mre) in _select_cases().items():
        N = X.size
        blocks = (N + 63) // 64          # ownership by blocks of 64 particles, as kabc_smc_run_dist
        per = (blocks + world - 1) // world
        lo, hi = min(rank * per * 64, N), min((rank + 1) * per * 64, N)
        eps, flag, ess, resample, new_own, idx = sharded_select(X[lo:hi], alive[lo:hi], lo, N, alpha, mre,
                                                                all_gather)
        full = np.concatenate(all_gather(new_own.astype(np.uint8))).astype(bool)
        res[name] = (eps, flag, ess, resample, full, idx)
    # the ONE-exchange course (dsel2_*): a window around the answer decides alone; a window off it stalls
    from kissabc_jl_amd.sharded import sharded_select_one_exchange
    for name, (X, alive, alpha, mre) in _select_cases().items():
        N = X.size
        blocks = (N + 63) // 64
        per = (blocks + world - 1) // world
        lo, hi = min(rank * per * 64, N), min((rank + 1) * per * 64, N)
        eps = res[name][0]
        # the masks of this case's selection are the previous iteration's: dead particles cost more than
        # every alive one in a run (they did not pass the last eps) -- the course relies on it
        Xr = X.copy()
        if np.isfinite(X[alive]).any():
            Xr[~alive] = np.nanmax(X[alive][np.isfinite(X[alive])]) + 1.0
        else:
            Xr[~alive] = np.inf
        spread = max(abs(eps) * 0.05, 0.05)
        for tag, window in (("hit", (eps - spread, eps + spread)), ("miss", (eps + spread, eps + 2 * spread)),
                            ("all", (-np.inf, np.inf))):
            r1 = sharded_select_one_exchange(Xr[lo:hi], alive[lo:hi], lo, N, alpha, mre, all_gather, window, Xr)
            if r1 is None:
                res[f"{name}@{tag}"] = (np.nan, -1, -1, False, np.zeros(1, bool), None)
            else:
                full = np.concatenate(all_gather(r1[4].astype(np.uint8))).astype(bool)
                res[f"{name}@{tag}"] = (r1[0], r1[1], r1[2], r1[3], full, r1[5])
    if rank == world - 1:   # (any rank: they all hold the same)
        np.savez(out_path, **{f"{n}_{i}": (np.array(-1) if v is None else np.asarray(v))
                              for n, r in res.items() for i, v in enumerate(r)})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_selection_equals_the_selection_on_the_whole_ensemble(tmp_path, world):
    out = str(tmp_path / "select.npz")
    mp.spawn(_select_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    got = np.load(out)
    for name, (X, alive, alpha, mre) in _select_cases().items():
        eps, flag, ess, resample, new, idx = _select_direct(X, alive, alpha, mre)
        assert float(got[f"{name}_0"]) == eps, name
        assert int(got[f"{name}_1"]) == flag and int(got[f"{name}_2"]) == ess, name
        assert bool(got[f"{name}_3"]) == resample, name
        assert np.array_equal(got[f"{name}_4"], new), name
        if resample:
            assert np.array_equal(got[f"{name}_5"], idx), name
        # one exchange: decided alone whenever it decides (a window around eps; the whole key range while
        # the target's bin fits the LDS list), stalled (NaN) with a window off the answer or eps == 0
        Xr = X.copy()
        Xr[~alive] = (np.nanmax(X[alive][np.isfinite(X[alive])]) + 1.0) if np.isfinite(X[alive]).any() else np.inf
        want = _select_direct(Xr, alive, alpha, mre)
        decided = 0
        for tag in ("hit", "miss", "all"):
            e1 = float(got[f"{name}@{tag}_0"])
            if np.isnan(e1):
                assert tag != "hit" or want[0] == 0.0 or not np.isfinite(want[0]) or name in ("ties", "all_equal"), (name, tag)
                continue
            decided += 1
            assert tag != "miss", name
            assert e1 == want[0] and int(got[f"{name}@{tag}_1"]) == want[1], (name, tag)
            assert int(got[f"{name}@{tag}_2"]) == want[2] and bool(got[f"{name}@{tag}_3"]) == want[3], (name, tag)
            assert np.array_equal(got[f"{name}@{tag}_4"], want[4]), (name, tag)
            if want[3]:
                assert np.array_equal(got[f"{name}@{tag}_5"], want[5]), (name, tag)
        assert decided >= 1 or name in ("ties", "all_equal") or want[0] == 0.0 or not np.isfinite(want[0]), name
