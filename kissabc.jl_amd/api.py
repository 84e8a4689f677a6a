"""Host-side mirror of the reference's user API for the walker-update path:

    ApproxKernelizedPosterior(prior, cost, scale)      src/types.jl:40-49
    ApproxPosterior(prior, cost, maxcost)              src/types.jl:76-82
    AIS(nparticles)                                    src/KissABC.jl:21-23
    sample(model, AIS(N), Ns; ntransitions, discard_initial, retry_sampling)
    sample(model, AIS(N), MCMCThreads(), Ns, Nc; ...)  src/KissABC.jl:106-173
    smc(prior, cost; kwargs...)                        src/smc.jl:92-206

Everything numerical happens in libkabc_hip.so (gfx950 kernels) behind the C
ABI of include/kabc.h.  This file only marshals arguments and shapes results
(bundle_samples / chainsstack, src/KissABC.jl:82-104).
"""
import collections
import concurrent.futures
import ctypes as C
import math
import time

import numpy as np

from . import _cdefs as cd
from . import _lib
from .costs import DeviceCost
from .distributions import Factored, UnivariateDistribution, as_factored


class Particles:
    """Minimal stand-in for MonteCarloMeasurements.Particles (one parameter's
    samples): the container `bundle_samples` builds (src/KissABC.jl:91)."""

    def __init__(self, particles):
        self.particles = np.asarray(particles)

    def __len__(self):
        return self.particles.shape[0]

    def mean(self):
        return float(np.mean(self.particles))

    def std(self):
        return float(np.std(self.particles, ddof=1))

    def isapprox(self, c, nsigma=2.0):
        """MonteCarloMeasurements `p ≈ c`: |mean − c| < nsigma · std."""
        return abs(self.mean() - c) < nsigma * self.std()

    def __array__(self, dtype=None, copy=None):
        return self.particles if dtype is None else self.particles.astype(dtype)

    def __repr__(self):
        return f"{self.mean():.4g} ± {self.std():.2g}"


class _ApproxModel:
    posterior = 0

    def __init__(self, prior, cost, eps):
        self.prior = as_factored(prior)
        self.scalar = isinstance(prior, UnivariateDistribution)
        if not isinstance(cost, DeviceCost):
            raise TypeError(
                "on the MI355X path `cost` must be a DeviceCost (kissabc_jl_amd.costs.*): a host "
                "closure cannot be called from a gfx950 kernel")
        self.cost = cost
        self.eps = float(eps)

    def __len__(self):  # length(density) = length(prior), src/types.jl:37
        return len(self.prior)

    def to_c(self):
        m = cd.Model()
        self._prior_c = self.prior.to_c()
        m.prior = C.cast(self._prior_c, C.POINTER(cd.Prior))
        m.D = len(self.prior)
        m.posterior = self.posterior
        m.eps = self.eps
        m.cost = self.cost.to_c()
        return m


class ApproxKernelizedPosterior(_ApproxModel):
    """Gaussian-kernel ABC density; `scale` = target_average_cost (src/types.jl:130-139)."""
    posterior = cd.POSTERIOR_KERNELIZED

    @property
    def scale(self):
        return self.eps


class ApproxPosterior(_ApproxModel):
    """Hard-threshold ABC density; `maxcost` (src/types.jl:140-149)."""
    posterior = cd.POSTERIOR_THRESHOLD

    @property
    def maxcost(self):
        return self.eps


class CommonLogDensity(_ApproxModel):
    """CommonLogDensity(nparameters, sample_init, lπ) -- src/types.jl:105-128, 151-161:
    classical MCMC on a log-density.  On the device path `lπ` is a DeviceCost that
    RETURNS THE LOG-DENSITY (built-in or costs.UserCost) and `sample_init` is a
    Factored / univariate distribution the initial walkers are drawn from -- or
    `InitFromSnippet(nparameters)`: the reference takes an arbitrary `rng -> sample` closure,
    here the log-density's C snippet may bring its own (`#define KABC_USER_SAMPLE_INIT 1` +
    `kabc_user_sample_init(x, D, params, data, ndata, rng)`, include/kabc_costs.h)."""
    posterior = cd.POSTERIOR_COMMON

    def __init__(self, nparameters, sample_init, lpi):
        super().__init__(sample_init, lpi, 1.0)
        if len(self.prior) != int(nparameters):
            raise ValueError("nparameters must equal the length of sample_init")

    @property
    def lπ(self):
        return self.cost


def compile_model(model_or_prior, cost=None, families=0):
    """Compile kernels specialised for ONE model (kabc_compile_model, include/kabc.h): the prior
    tuple's families and parameters become compile-time constants of a run-time compiled
    translation unit.  compile_model(model) for an ApproxKernelizedPosterior / ApproxPosterior,
    compile_model(prior, cost) for smc / ABCDE / pfilter; `families`: bit mask of
    cd.FAMILY_AIS / _SMC / _ABCDE / _PFILTER (0 = AIS + smc).  Afterwards sample / smc / ... on
    the same prior components and cost use the specialised kernels; results keep their bits.
    Returns the registration's handle (0: the model is left to the prebuilt kernels)."""
    if cost is None:
        m = model_or_prior.to_c()
        keep = model_or_prior
    else:
        keep = _ApproxModel(model_or_prior, cost, 1.0)
        m = keep.to_c()
    h = C.c_int32()
    _lib.check(_lib.load().kabc_compile_model(C.byref(m), int(families), C.byref(h)))
    del keep
    return int(h.value)


def set_specialize(mode):
    """kabc_set_specialize: "env" (KABC_SPECIALIZE decides), "off" (never specialise, never start the
    compiler worker process), "blocking" (compile at first sight), "background" (the worker)."""
    m = {"env": -1, "off": 0, "blocking": 1, "background": 2}[mode]
    _lib.check(_lib.load().kabc_set_specialize(m))


class AIS:
    """AIS(nparticles) -- src/KissABC.jl:21-23"""

    def __init__(self, nparticles):
        self.nparticles = int(nparticles)


class MCMCThreads:
    """Tag for independent chains (AbstractMCMC.MCMCThreads, re-exported at
    src/KissABC.jl:9,175).  On this path chains are independent ensembles with
    distinct seeds advanced TOGETHER: chain is a grid dimension of every launch."""


class AisEnsemble:
    """kabc_ais_t: the device-resident AISState (src/KissABC.jl:25-33)."""

    def __init__(self, model, nparticles, seed=0, ctx=None, sharded=None, comm=None, seeds=None):
        """`seeds` (a sequence) makes a BATCH handle: len(seeds) independent ensembles of
        `nparticles` walkers, chain = a grid dimension of every launch (kabc_ais_create_batch);
        state / trace arrays gain a leading chain axis."""
        self.model = model
        self.comm = comm
        self.nchains = 1 if seeds is None else len(seeds)
        self.batched = seeds is not None
        self.ctx = comm.ctx if comm is not None else (ctx or _lib.default_context())
        self.N = int(nparticles)
        self.D = len(model)
        self._cmodel = model.to_c()
        self._h = C.c_void_p()
        lib = _lib.load()
        if seeds is not None:
            arr = (C.c_uint64 * len(seeds))(*[int(v) & (2 ** 64 - 1) for v in seeds])
            _lib.check(lib.kabc_ais_create_batch(self.ctx.handle, C.byref(self._cmodel), self.N,
                                                 len(seeds), arr, C.byref(self._h)))
        elif comm is not None:
            # walker-sharded over the communicator's ranks; the library owns the exchange
            _lib.check(lib.kabc_ais_create_dist(comm.handle, C.byref(self._cmodel), self.N,
                                                int(seed), C.byref(self._h)))
        elif sharded is None:
            _lib.check(lib.kabc_ais_create(self.ctx.handle, C.byref(self._cmodel), self.N,
                                           int(seed), C.byref(self._h)))
        else:
            rank, world, p0, p1 = sharded
            _lib.check(lib.kabc_ais_create_sharded(self.ctx.handle, C.byref(self._cmodel), self.N,
                                                   rank, world, int(seed), C.c_void_p(p0),
                                                   C.c_void_p(p1), C.byref(self._h)))
        self.owned = (lib.kabc_ais_owned(self._h, 0), lib.kabc_ais_owned(self._h, 1))

    # step(rng, model, spl; retry_sampling) -- src/KissABC.jl:35-64
    def init(self, retry_sampling=100):
        _lib.check(_lib.load().kabc_ais_init(self._h, int(retry_sampling)))
        return self

    # step(rng, model, spl, state; ntransitions) x N x ngenerations -- src/KissABC.jl:66-80
    def advance(self, ngenerations, ntransitions=1, collect=False, out=None):
        """`collect=True` returns the sample trace [generation][walker][D]
        ([generation][chain][walker][D] for a batch handle); `out` may supply its buffer
        (C-contiguous float64, e.g. from _lib.pinned_empty)."""
        lib = _lib.load()
        ptr = None
        if collect or out is not None:
            shape = ((int(ngenerations), self.nchains, self.N, self.D) if self.batched
                     else (int(ngenerations), self.N, self.D))
            if out is None:
                out = _lib.pinned_empty(shape)
            if out.shape != shape or out.dtype != np.float64 or not out.flags.c_contiguous:
                raise ValueError(f"out must be a C-contiguous float64 array of shape {shape}")
            ptr = out.ctypes.data_as(cd.c_double_p)
        st = cd.Stats()
        _lib.check(lib.kabc_ais_advance(self._h, int(ngenerations), int(ntransitions), ptr,
                                        C.byref(st)))
        self.last_stats = {"proposals": st.proposals, "cost_evals": st.cost_evals,
                           "accepted": st.accepted}
        return out

    def half_generation(self, half, ntransitions, trace_ptr=None):
        _lib.check(_lib.load().kabc_ais_half_generation(
            self._h, int(half), int(ntransitions), C.c_void_p(trace_ptr) if trace_ptr else None))

    def end_generation(self, ntransitions):
        _lib.check(_lib.load().kabc_ais_end_generation(self._h, int(ntransitions)))

    def state(self):
        n = self.owned[0] + self.owned[1]
        lead = (self.nchains,) if self.batched else ()
        x = np.empty(lead + (n, self.D))
        lp = np.empty(lead + (n,))
        ll = np.empty(lead + (n,))
        t = C.c_uint64()
        _lib.check(_lib.load().kabc_ais_get_state(
            self._h, x.ctypes.data_as(cd.c_double_p), lp.ctypes.data_as(cd.c_double_p),
            ll.ctypes.data_as(cd.c_double_p), C.byref(t)))
        return x, lp, ll, t.value

    def set_state(self, x, lp, ll, t=0):
        x = np.ascontiguousarray(x, dtype=np.float64)
        lp = np.ascontiguousarray(lp, dtype=np.float64)
        ll = np.ascontiguousarray(ll, dtype=np.float64)
        _lib.check(_lib.load().kabc_ais_set_state(
            self._h, x.ctypes.data_as(cd.c_double_p), lp.ctypes.data_as(cd.c_double_p),
            ll.ctypes.data_as(cd.c_double_p), int(t)))

    def segments(self, half):
        """[(first global row of the half, rows), ...]: the owned row ranges of `half` in the
        order state() / get_debug() lay the owned rows out (kabc_ais_owned_segments; one range
        unless the handle is sharded with more than one exchange chunk)."""
        cap = cd.KABC_MAX_EXCHANGE_CHUNKS
        first, count = (C.c_int64 * cap)(), (C.c_int64 * cap)()
        n = _lib.load().kabc_ais_owned_segments(self._h, int(half), first, count, cap)
        return [(first[i], count[i]) for i in range(n)]

    def stats(self):
        st = cd.Stats()
        _lib.check(_lib.load().kabc_ais_get_stats(self._h, C.byref(st)))
        return {"proposals": st.proposals, "cost_evals": st.cost_evals, "accepted": st.accepted}

    SPEC_STATES = ("none", "pending", "active", "failed")   # KABC_SPEC_* of include/kabc.h

    def spec_state(self):
        """(state, launches_before_switch): which kernels the half-generation launches run on --
        the prebuilt ones ("none" / "pending" / "failed") or the model's own ("active", the default
        non-blocking specialisation of include/kabc.h); launches that ran before the switch, -1."""
        st, n = C.c_int32(0), C.c_int64(-1)
        _lib.check(_lib.load().kabc_ais_spec_state(self._h, C.byref(st), C.byref(n)))
        return self.SPEC_STATES[st.value], n.value

    @property
    def driver(self):
        """"small": kabc_ais_advance runs every generation of a call in one launch of one workgroup
        per chain (csrc/ais_small_kernel.hpp); "halves": one launch per half-generation."""
        return "small" if _lib.load().kabc_ais_driver(self._h) else "halves"

    def ensemble(self):
        """[N][D] unrounded positions of ALL walkers in walker-id order (for a sharded
        handle: this rank's copy after the last all-gather)."""
        x = np.empty(((self.nchains,) if self.batched else ()) + (self.N, self.D))
        _lib.check(_lib.load().kabc_ais_get_ensemble(self._h, x.ctypes.data_as(cd.c_double_p)))
        return x

    def set_timing(self, max_launches, stride=1):
        _lib.check(_lib.load().kabc_ais_set_timing(self._h, int(max_launches)))
        _lib.check(_lib.load().kabc_ais_set_timing_stride(self._h, int(stride)))

    def kernel_ms(self):
        n = C.c_int64()
        ms = _lib.load().kabc_ais_kernel_ms(self._h, C.byref(n))
        return ms, n.value

    def exchange_us(self):
        """kabc_ais_exchange_us: (compute, exchange, exposed) microseconds per half-generation and
        the number of exchange chunks, over the half-generations timed since set_timing (sharded
        handles; zeros otherwise)"""
        out = (C.c_double * 4)()
        _lib.check(_lib.load().kabc_ais_exchange_us(self._h, out))
        return {"compute_us_per_half": out[0], "exchange_us_per_half": out[1],
                "exposed_us_per_half": out[2], "chunks": int(out[3])}

    def set_debug(self, ntransitions):
        _lib.check(_lib.load().kabc_ais_set_debug(self._h, int(ntransitions)))

    def get_debug(self, ntransitions):
        n = self.owned[0] + self.owned[1]
        out = np.empty((n, int(ntransitions), 6), dtype=np.int32)
        _lib.check(_lib.load().kabc_ais_get_debug(
            self._h, out.ctypes.data_as(C.POINTER(C.c_int32)), out.size))
        return out

    def close(self):
        if self._h:
            _lib.load().kabc_ais_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def chain_seeds(seed, nchains):
    """The per-chain seeds sample(..., MCMCThreads(), Ns, Nc) derives from `seed`
    (AbstractMCMC seeds each chain from the parent rng; here: a golden-ratio stride)."""
    return [(int(seed) + 0x9E3779B97F4A7C15 * (c + 1)) % (1 << 63) for c in range(int(nchains))]


def _bundle(samples, scalar):
    """bundle_samples (src/KissABC.jl:82-94): [Ns][D] -> one Particles per parameter."""
    P = [Particles(samples[:, k]) for k in range(samples.shape[1])]
    return P[0] if (len(P) == 1 or scalar) else P


def sample(model, spl, *args, ntransitions=1, discard_initial=0, retry_sampling=100, seed=0,
           progress=False, ctx=None, return_array=False, **kwargs):
    """sample(model, AIS(N), Ns; ...) and sample(model, AIS(N), MCMCThreads(), Ns, Nc; ...).

    The device advances a whole generation (every walker `ntransitions` times) per
    launch pair and a generation yields the N samples the reference's N
    consecutive step() calls would emit (src/KissABC.jl:66-80), so
    ceil(discard_initial/N) generations are discarded and ceil(Ns/N) are kept.
    """
    if not isinstance(spl, AIS):
        raise TypeError("sampler must be AIS(nparticles)")
    if args and isinstance(args[0], MCMCThreads):
        # chains are a grid dimension of ONE device handle: every launch advances all Nc
        # ensembles (kabc_ais_create_batch); chain c is bit-identical to a single-chain run
        # with its seed
        _, Ns, Nc = args
        Ns, Nc, N, D = int(Ns), int(Nc), spl.nparticles, len(model)
        seeds = chain_seeds(seed, Nc)
        ens = AisEnsemble(model, N, ctx=ctx, seeds=seeds)
        gk = max(1, -(-Ns // N))
        big = gk * Nc * N * D * 8 > (1 << 20)   # (a small trace is not worth a helper thread: see below)
        pool = concurrent.futures.ThreadPoolExecutor(1) if big else None
        buf = pool.submit(_lib.pinned_empty, (gk, Nc, N, D)) if big else None
        try:
            ens.init(retry_sampling)
            gd = -(-int(discard_initial) // N)
            if gd:
                ens.advance(gd, ntransitions)
            tr = ens.advance(gk, ntransitions,
                             out=buf.result() if big else _lib.pinned_empty((gk, Nc, N, D)))   # [gk][Nc][N][D]
            chains = np.ascontiguousarray(tr.transpose(1, 0, 2, 3)).reshape(Nc, gk * N, D)[:, :Ns]
        finally:
            if pool is not None:
                pool.shutdown(wait=True)
            ens.close()
        stacked = chains.reshape(Nc * Ns, D)  # chainsstack, src/KissABC.jl:96-104
        return stacked if return_array else _bundle(stacked, model.scalar)
    (Ns,) = args
    Ns = int(Ns)
    N = spl.nparticles
    ens = AisEnsemble(model, N, seed=seed, ctx=ctx)
    gk = max(1, -(-Ns // N))
    shape = (gk, N, len(model))
    # the page-locked trace buffer is allocated on a helper thread (pinning costs
    # ~40 us per MiB) while init and the discarded generations run on the device; a small
    # trace (the reference's own test shapes) is not worth a thread: 0.1-0.2 ms of a 0.5 ms call
    big = gk * N * len(model) * 8 > (1 << 20)
    pool = concurrent.futures.ThreadPoolExecutor(1) if big else None
    buf = pool.submit(_lib.pinned_empty, shape) if big else None
    try:
        ens.init(retry_sampling)
        gd = -(-int(discard_initial) // N)
        if gd:
            ens.advance(gd, ntransitions)
        out = ens.advance(gk, ntransitions, out=buf.result() if big else _lib.pinned_empty(shape))
        out = out.reshape(gk * N, len(model))[:Ns]
    finally:
        if pool is not None:
            pool.shutdown(wait=True)
        ens.close()
    return out if return_array else _bundle(out, model.scalar)


class SmcResult(collections.namedtuple("SmcResult", ["P", "C", "eps", "info"])):
    """(P, C, ϵ) of src/smc.jl:205 (+ info); `.ϵ`/`.ε` alias `.eps`."""
    __slots__ = ()

    def __getattr__(self, name):
        if name in ("\u03b5", "\u03f5"):
            return self.eps
        raise AttributeError(name)


def smc(prior, cost, *, nparticles=100, alpha=0.95, mcmc_retrys=0, mcmc_tol=0.015, epstol=0.0,
        r_epstol=None, min_r_ess=None, max_stretch=2.0, verbose=False, parallel=False, seed=0,
        ctx=None, return_array=False, comm=None, shard=None):
    """smc(prior, cost; ...) -- src/smc.jl:92-206, same keywords and defaults.
    `parallel` is accepted and ignored (every particle is a GPU lane).
    `comm` (a comm.Comm): the cost loop is sharded over the communicator's ranks
    (kabc_smc_run_dist -- the reference's `parallel = true` leg across GPUs, for expensive
    simulators); collective, every rank gets the same result, equal to the single-GPU one.
    `shard` = "particles": the ranks own their particles and the epsilon-selection is sharded too
    (kabc_smc_run_dist_mode, KABC_SMC_DIST_PARTICLES); "cost_loop": the pass only; None:
    kabc_smc_run_dist's default (KABC_SMC_DIST).
    Returns (P, C, ϵ) as the reference does (+ an `info` dict)."""
    fac = as_factored(prior)
    scalar = isinstance(prior, UnivariateDistribution)
    if not isinstance(cost, DeviceCost):
        raise TypeError("`cost` must be a DeviceCost on the MI355X path")
    lib = _lib.load()
    ctx = ctx or _lib.default_context()
    o = cd.SmcOpts()
    lib.kabc_smc_default_opts(C.byref(o))
    o.nparticles = int(nparticles)
    o.alpha = float(alpha)
    o.mcmc_retrys = int(mcmc_retrys)
    o.verbose = int(bool(verbose))
    o.mcmc_tol = float(mcmc_tol)
    o.epstol = float(epstol)
    o.r_epstol = math.nan if r_epstol is None else float(r_epstol)
    o.min_r_ess = math.nan if min_r_ess is None else float(min_r_ess)
    o.max_stretch = float(max_stretch)
    o.seed = int(seed)
    N, D = int(nparticles), len(fac)
    n_alloc = max(N, 1)
    t_host0 = time.perf_counter()
    theta = _lib.result_empty((n_alloc, D))
    Cst = _lib.result_empty(n_alloc)
    alive = np.zeros(n_alloc, dtype=np.uint8)
    log = (cd.SmcIter * 4096)()
    r = cd.SmcResult()
    r.theta = theta.ctypes.data_as(cd.c_double_p)
    r.cost = Cst.ctypes.data_as(cd.c_double_p)
    r.alive = alive.ctypes.data_as(C.POINTER(C.c_uint8))
    r.iter_log = log
    r.iter_log_cap = 4096
    cc = cost.to_c()
    if shard not in (None, "cost_loop", "particles"):
        raise ValueError('shard is None, "cost_loop" or "particles"')
    if comm is not None and shard is not None:
        _lib.check(lib.kabc_smc_run_dist_mode(comm.handle, fac.to_c(), D, C.byref(cc), C.byref(o),
                                              1 if shard == "particles" else 0, C.byref(r)))
    elif comm is not None:
        _lib.check(lib.kabc_smc_run_dist(comm.handle, fac.to_c(), D, C.byref(cc), C.byref(o), C.byref(r)))
    else:
        _lib.check(lib.kabc_smc_run(ctx.handle, fac.to_c(), D, C.byref(cc), C.byref(o), C.byref(r)))
    t_host1 = time.perf_counter()
    mask = alive.view(np.bool_)          # (the library writes 0 / 1)
    # every particle alive (the usual end of a run): the result IS the array, not a gathered copy
    kept = theta if (r.n_alive == n_alloc and N > 0) else theta[mask]
    nit = min(r.iterations, 4096)
    info = {
        "iterations": r.iterations, "n_alive": r.n_alive, "cost_evals": r.cost_evals,
        "proposals": r.proposals, "alive": mask, "theta_all": theta,
        "kernel_ms_mcmc": r.kernel_ms_mcmc, "mcmc_launches": r.mcmc_launches,
        "log": [dict(eps=log[i].eps, ess=log[i].ess, accepted=log[i].accepted,
                     resampled=log[i].resampled, flag=log[i].flag, passes=log[i].mcmc_passes)
                for i in range(nit)],
    }
    if True:   # how the run was driven (kabc_smc_dist_stats)
        ds = (C.c_int64 * 8)()
        lib.kabc_smc_dist_stats(ds)
        info["dist"] = {"iterations": ds[0], "collectives": ds[1], "host_looks": ds[2],
                        "one_exchange_selections": ds[3], "phase_by_phase_selections": ds[4],
                        "passes": ds[5], "batched": bool(ds[6]), "collectives_per_usual_iteration": ds[7],
                        "collectives_per_iteration": round(ds[1] / max(ds[0], 1), 3)}
    P = kept if return_array else _bundle(kept, scalar)
    # where the wall time of this call went: kabc_smc_run (with its result copy) / this wrapper
    info["host_ms"] = {"kabc_smc_run": (t_host1 - t_host0) * 1e3,
                       "python_after": (time.perf_counter() - t_host1) * 1e3}
    return SmcResult(P, Cst, r.eps, info)


class AbcdeResult(collections.namedtuple("AbcdeResult", ["P", "C", "reached_eps", "info"])):
    """(P, C, reached_ϵ) of src/smc.jl:428 (+ info); `.reached_ϵ` aliases `.reached_eps`."""
    __slots__ = ()

    def __getattr__(self, name):
        if name in ("reached_\u03b5", "reached_\u03f5"):
            return self.reached_eps
        raise AttributeError(name)


def ABCDE(prior, cost, eps_target, *, nparticles=50, generations=20, α=0.0, alpha=None,
          parallel=False, earlystop=False, verbose=False, proposal_width=1.0, seed=0, ctx=None,
          return_array=False):
    """ABCDE(prior, cost, ϵ_target; ...) -- src/smc.jl:347-430, same keywords
    (`alpha` is an ASCII alias of `α`; `parallel` is accepted and ignored).
    Returns (P, C, reached_ϵ) as the reference does (+ info)."""
    fac = as_factored(prior)
    scalar = isinstance(prior, UnivariateDistribution)
    if not isinstance(cost, DeviceCost):
        raise TypeError("`cost` must be a DeviceCost on the MI355X path")
    lib = _lib.load()
    ctx = ctx or _lib.default_context()
    o = cd.AbcdeOpts()
    lib.kabc_abcde_default_opts(C.byref(o))
    o.nparticles, o.generations, o.eps_target = int(nparticles), int(generations), float(eps_target)
    o.alpha = float(α if alpha is None else alpha)
    o.proposal_width, o.earlystop, o.verbose, o.seed = (float(proposal_width), int(bool(earlystop)),
                                                        int(bool(verbose)), int(seed))
    N, D = max(int(nparticles), 1), len(fac)
    theta = np.empty((N, D))
    Cst = np.empty(N)
    r = cd.AbcdeResult()
    r.theta = theta.ctypes.data_as(cd.c_double_p)
    r.cost = Cst.ctypes.data_as(cd.c_double_p)
    cc = cost.to_c()
    _lib.check(lib.kabc_abcde_run(ctx.handle, fac.to_c(), D, C.byref(cc), C.byref(o), C.byref(r)))
    info = {"generations_run": r.generations_run, "nsims": r.nsims}
    return AbcdeResult(theta if return_array else _bundle(theta, scalar), Cst,
                       bool(r.reached_eps), info)


PfilterResult = collections.namedtuple("PfilterResult", ["P", "C", "info"])


def pfilter(prior, cost, N, *, q=0.7, eff_tol=0.1, epstol=-math.inf, max_iters=math.inf,
            proposal_width=0.75, verbose=False, parallel=False, seed=0, ctx=None,
            return_array=False):
    """pfilter(prior, cost, N; ...) -- src/smc.jl:275-340, same keywords
    (`parallel` accepted and ignored).  Returns (P, C) as the reference does (+ info)."""
    fac = as_factored(prior)
    scalar = isinstance(prior, UnivariateDistribution)
    if not isinstance(cost, DeviceCost):
        raise TypeError("`cost` must be a DeviceCost on the MI355X path")
    lib = _lib.load()
    ctx = ctx or _lib.default_context()
    o = cd.PfilterOpts()
    lib.kabc_pfilter_default_opts(C.byref(o))
    o.nparticles, o.q, o.eff_tol, o.epstol = int(N), float(q), float(eff_tol), float(epstol)
    o.proposal_width, o.verbose, o.seed = float(proposal_width), int(bool(verbose)), int(seed)
    o.max_iters = -1 if math.isinf(max_iters) else int(math.floor(max_iters))
    D = len(fac)
    n_eff = lib.kabc_pfilter_nparticles(int(N), float(q), D)
    theta = np.empty((n_eff, D))
    Cst = np.empty(n_eff)
    r = cd.PfilterResult()
    r.theta = theta.ctypes.data_as(cd.c_double_p)
    r.cost = Cst.ctypes.data_as(cd.c_double_p)
    cc = cost.to_c()
    _lib.check(lib.kabc_pfilter_run(ctx.handle, fac.to_c(), D, C.byref(cc), C.byref(o), C.byref(r)))
    info = {"eps": r.eps, "eff": r.eff, "iterations": r.iterations, "nreps": r.nreps,
            "cost_evals": r.cost_evals, "nparticles": n_eff}
    return PfilterResult(theta if return_array else _bundle(theta, scalar),
                         Cst if return_array else Particles(Cst), info)
