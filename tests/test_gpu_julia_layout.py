"""-m gpu: the array-layout assumptions of julia/KissABCHip.jl, replayed from ctypes with
COLUMN-MAJOR (Fortran-order) numpy arrays standing in for Julia's, against the oracle.

Julia's `Matrix{Float64}(undef, D, N)` handed to a C function that writes `[N][D]` row-major
holds walker n in column n; the shim relies on that in step(init) (kabc_ais_get_ensemble),
generation! (kabc_ais_advance's out_samples), sample(..., MCMCThreads(), ...) -- an
`Array{Float64,4}(undef, D, N, Nc, gk)` for `[gen][chain][N][D]`, then
`reshape(view(tr, :, :, c, :), D, N * gk)` -- and smc / ABCDE / pfilter (theta D x N).  The call
sequences below are the shim's, argument for argument (tests/test_julia_shim_static.py checks the
signatures themselves)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _fptr(a):
    assert a.flags["F_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _model(k):
    prior = k.Factored(k.Normal(1, 0.5), k.DiscreteUniform(1, 10), k.Uniform(-2, 2))
    return k.ApproxKernelizedPosterior(prior, k.costs.GaussDist([1.0, 4.0, 0.5]), 0.8)


def test_step_init_and_generation_cache_layout(k, orc, gpu_ctx):
    from kissabc_jl_amd import _cdefs as cd
    lib, model, N, D, nt, seed = k._lib.load(), _model(k), 37, 3, 4, 99
    cm = model.to_c()
    h = C.c_void_p()
    k._lib.check(lib.kabc_ais_create(gpu_ctx.handle, C.byref(cm), N, seed, C.byref(h)))
    try:
        k._lib.check(lib.kabc_ais_init(h, 100))
        x = np.empty((D, N), order="F")                       # Matrix{Float64}(undef, D, N)
        k._lib.check(lib.kabc_ais_get_ensemble(h, _fptr(x)))
        o = orc.OracleAIS(model, N, seed=seed).init()
        xo = o.state()[0]                                     # [N][D], unrounded
        assert np.array_equal(x[:, N - 1], xo[N - 1])         # wrap(model, view(x, :, N)): the LAST walker
        assert np.array_equal(x.T, xo)
        cache = np.empty((D, N), order="F")                   # st.cache
        k._lib.check(lib.kabc_ais_advance(h, 1, nt, _fptr(cache), None))
        ref = o.generations_sync(1, nt)[0]                    # [N][D] push_p'ed samples of the generation
        for i in range(N):                                    # view(st.cache, :, st.i)
            assert np.array_equal(cache[:, i], ref[i])
        assert set(np.unique(cache[1])) <= set(range(1, 11))  # the discrete coordinate arrives rounded
    finally:
        lib.kabc_ais_destroy(h)
    assert cd.KABC_VERSION == lib.kabc_version()


def test_mcmcthreads_batch_reshape(k, orc, gpu_ctx):
    lib, model, N, D, nt = k._lib.load(), _model(k), 12, 3, 3
    Nc, Ns = 4, 30
    seeds = np.array([11, 12, 13, 14], dtype=np.uint64)
    cm = model.to_c()
    h = C.c_void_p()
    k._lib.check(lib.kabc_ais_create_batch(gpu_ctx.handle, C.byref(cm), N, Nc,
                                           seeds.ctypes.data_as(C.POINTER(C.c_uint64)), C.byref(h)))
    try:
        k._lib.check(lib.kabc_ais_init(h, 100))
        gd, gk = -(-5 // N), max(1, -(-Ns // N))              # cld(discard_initial = 5, N), cld(Ns, N)
        k._lib.check(lib.kabc_ais_advance(h, gd, nt, None, None))
        tr = np.empty((D, N, Nc, gk), order="F")              # Array{Float64,4}(undef, D, N, Nc, gk)
        k._lib.check(lib.kabc_ais_advance(h, gk, nt, _fptr(tr), None))
        for c in range(Nc):
            cols = np.reshape(tr[:, :, c, :], (D, N * gk), order="F")   # reshape(permutedims(view(...)), D, N * gk)
            o = orc.OracleAIS(model, N, seed=int(seeds[c])).init()
            o.generations_sync(gd, nt, collect=False)
            ref = o.generations_sync(gk, nt).reshape(gk * N, D)
            for j in range(Ns):                               # samples = [wrap(model, view(cols, :, j)) for j in 1:Ns]
                assert np.array_equal(cols[:, j], ref[j]), (c, j)
    finally:
        lib.kabc_ais_destroy(h)


def test_smc_theta_matrix_layout(k, orc, gpu_ctx):
    from kissabc_jl_amd import _cdefs as cd
    lib = k._lib.load()
    prior = k.Factored(k.Normal(1, 0.5), k.DiscreteUniform(1, 10))
    cost = k.costs.NoisyQuadDU(5.5)
    N, D, seed = 300, 2, 7
    theta = np.empty((D, N), order="F")                       # Matrix{Float64}(undef, D, max(N, 1))
    Cv = np.empty(N)
    alive = np.zeros(N, dtype=np.uint8)
    o = cd.SmcOpts()
    lib.kabc_smc_default_opts(C.byref(o))
    # KabcSmcOpts(N, alpha, mcmc_retrys, verbose, mcmc_tol, epstol, r_epstol, min_r_ess, max_stretch, seed, 0)
    o.nparticles, o.alpha, o.mcmc_retrys, o.verbose, o.mcmc_tol, o.epstol = N, 0.95, 0, 0, 0.015, 0.0
    o.r_epstol, o.min_r_ess, o.max_stretch, o.seed, o.max_iterations = (1 - 0.95) ** 1.5 / 50, 0.95 ** 2, 2.0, seed, 0
    r = cd.SmcResult()
    r.theta, r.cost = _fptr(theta), Cv.ctypes.data_as(C.POINTER(C.c_double))
    r.alive = alive.ctypes.data_as(C.POINTER(C.c_uint8))
    pri, cc = prior.to_c(), cost.to_c()
    k._lib.check(lib.kabc_smc_run(gpu_ctx.handle, pri, D, C.byref(cc), C.byref(o), C.byref(r)))
    ref = orc.smc(prior, cost, nparticles=N, seed=seed)
    keep = np.flatnonzero(alive)                              # findall(!=(0x00), alive)
    for kk in range(D):                                       # P = [Particles(theta[k, keep]) for k in 1:D]
        assert np.array_equal(theta[kk, keep], ref["P"][:, kk])
    assert np.array_equal(Cv, ref["C"]) and r.eps == ref["eps"]
