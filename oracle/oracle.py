"""ctypes binding of the CPU ORACLE (oracle/_build/libkabc_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product (kissabc.jl_amd/) never imports it.
It reuses the product's *struct layouts* (kissabc_jl_amd._cdefs mirrors
include/kabc.h) and model descriptors, nothing else.
"""
import ctypes as C
import math
import os
import subprocess

import numpy as np

import kissabc_jl_amd  # noqa: F401  (registers the package)
from kissabc_jl_amd import _cdefs as cd
from kissabc_jl_amd.distributions import as_factored

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_build", "libkabc_oracle.so")
_lib = None


class TraceRec(C.Structure):
    _fields_ = [("move", C.c_int32), ("accepted", C.c_int32), ("a", C.c_int32),
                ("b", C.c_int32), ("c", C.c_int32), ("cost_evaluated", C.c_int32)]


class OracleError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(message)
        self.status = status


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        VP, dp = C.c_void_p, cd.c_double_p
        sig = {
            "orc_last_error": (C.c_char_p, []),
            "orc_philox4x32_10": (None, [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                         C.POINTER(C.c_uint32)]),
            "orc_philox4x32_r": (None, [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_int32,
                                        C.POINTER(C.c_uint32)]),
            "orc_philox_rounds": (C.c_int32, []),
            "orc_math_vec": (None, [C.c_int32, C.c_int64, dp, dp]),
            "orc_normal_pairs": (None, [C.c_int64, C.POINTER(C.c_uint64), dp]),
            "orc_div_rc_vec": (None, [C.c_int64, dp, dp, dp]),
            "orc_factored_logpdf": (C.c_int32, [C.POINTER(cd.Prior), C.c_int32, C.c_int64, dp, dp]),
            "orc_factored_pdf": (C.c_int32, [C.POINTER(cd.Prior), C.c_int32, C.c_int64, dp, dp]),
            "orc_push_p": (C.c_int32, [C.POINTER(cd.Prior), C.c_int32, C.c_int64, dp, dp]),
            "orc_factored_rand": (C.c_int32, [C.POINTER(cd.Prior), C.c_int32, C.c_uint64,
                                              C.c_uint32, C.c_int64, C.c_int64, C.c_uint64, dp]),
            "orc_cost_eval": (C.c_double, [C.POINTER(cd.Cost), C.c_int32, dp, C.c_uint64,
                                           C.c_uint32, C.c_uint64, C.c_uint32]),
            "orc_cdf_g_inv": (C.c_double, [C.c_double, C.c_double]),
            "orc_register_user_cost": (C.c_int32, [C.c_int32, C.c_void_p]),
            "orc_register_user_init": (C.c_int32, [C.c_int32, C.c_void_p]),
            "orc_register_user_prior": (C.c_int32, [C.c_int32, C.c_void_p, C.c_void_p, C.c_int32]),
            "orc_register_user_mvprior": (C.c_int32, [C.c_int32, C.c_void_p, C.c_void_p]),
            "orc_mvnormal_register": (C.c_int32, [dp, dp, C.c_int32, C.POINTER(C.c_int32)]),
            "orc_mvnormal_block": (C.c_uint64, [C.c_int32]),
            "orc_ais_create": (C.c_int32, [C.POINTER(cd.Model), C.c_int64, C.c_uint64,
                                           C.POINTER(VP)]),
            "orc_ais_init": (C.c_int32, [VP, C.c_int32]),
            "orc_ais_steps_serial": (C.c_int32, [VP, C.c_int64, C.c_int32, dp]),
            "orc_ais_generations_sync": (C.c_int32, [VP, C.c_int64, C.c_int32, dp,
                                                     C.POINTER(TraceRec)]),
            "orc_ais_half_generation": (C.c_int32, [VP, C.c_int32, C.c_int32, C.c_int64,
                                                    C.c_int64]),
            "orc_ais_end_generation": (C.c_int32, [VP, C.c_int32]),
            "orc_ais_get_state": (C.c_int32, [VP, dp, dp, dp, C.POINTER(C.c_uint64)]),
            "orc_ais_set_state": (C.c_int32, [VP, dp, dp, dp, C.c_uint64]),
            "orc_ais_get_stats": (C.c_int32, [VP, C.POINTER(cd.Stats)]),
            "orc_ais_destroy": (None, [VP]),
            "orc_smc_run": (C.c_int32, [C.POINTER(cd.Prior), C.c_int32, C.POINTER(cd.Cost),
                                        C.POINTER(cd.SmcOpts), C.POINTER(cd.SmcResult)]),
            "orc_quantile": (C.c_int32, [dp, C.c_int64, C.c_double, dp]),
            "orc_pfilter_nparticles": (C.c_int64, [C.c_int64, C.c_double, C.c_int32]),
            "orc_pfilter_run": (C.c_int32, [C.POINTER(cd.Prior), C.c_int32, C.POINTER(cd.Cost),
                                            C.POINTER(cd.PfilterOpts), C.POINTER(cd.PfilterResult)]),
            "orc_abcde_run": (C.c_int32, [C.POINTER(cd.Prior), C.c_int32, C.POINTER(cd.Cost),
                                          C.POINTER(cd.AbcdeOpts), C.POINTER(cd.AbcdeResult)]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def _check(st):
    if st != 0:
        raise OracleError(st, load().orc_last_error().decode("utf-8", "replace"))


def _dp(a):
    return a.ctypes.data_as(cd.c_double_p)


_user_prior_done = set()


def register_user_prior(dist):
    """Compile the C snippet of a kissabc_jl_amd.distributions.UserPrior with gcc and register it
    with the oracle under the same kind, so that the oracle evaluates the same family."""
    import hashlib
    if dist.kind in _user_prior_done:
        return
    text = ('#include "kabc_sampling_base.h"\n' + dist.source +
            "\ndouble orc_user_prior_logpdf_entry(double x, const double* p, const double* tab) {\n"
            "    return kabc_user_prior_logpdf(x, p, tab);\n}\n"
            "double orc_user_prior_rand_entry(const double* p, const kabc_slotwin_t* w) {\n"
            "    return kabc_user_prior_rand(p, w);\n}\n")
    tag = hashlib.sha1(text.encode()).hexdigest()[:16]
    bdir = os.path.join(_HERE, "_build")
    os.makedirs(bdir, exist_ok=True)
    so = os.path.join(bdir, f"libuserprior_{tag}.so")
    if not os.path.exists(so):
        src = os.path.join(bdir, f"userprior_{tag}.c")
        with open(src, "w") as f:
            f.write(text)
        subprocess.check_call(["gcc", "-O2", "-std=gnu11", "-fPIC", "-ffp-contract=off", "-mfma",
                               "-I", os.path.join(os.path.dirname(_HERE), "include"), "-shared",
                               "-o", so, src, "-lm"])
    lib = _user_libs.get(so) or C.CDLL(so)
    _user_libs[so] = lib
    _check(load().orc_register_user_prior(dist.kind, C.cast(lib.orc_user_prior_logpdf_entry, C.c_void_p),
                                          C.cast(lib.orc_user_prior_rand_entry, C.c_void_p),
                                          int(dist.discrete)))
    _user_prior_done.add(dist.kind)


def register_user_mvprior(fac):
    """the snippet of a kissabc_jl_amd.distributions.UserMvPrior (a JOINT prior), compiled with gcc and
    registered with the oracle under the same kind"""
    import hashlib
    if fac.kind in _user_prior_done:
        return
    text = ('#include "kabc_sampling_base.h"\n' + fac.source +
            "\ndouble orc_user_mvprior_logpdf_entry(const double* x, int D, const double* p, int st, const double* tab) {\n"
            "    return kabc_user_mvprior_logpdf(x, D, p, st, tab);\n}\n"
            "void orc_user_mvprior_rand_entry(double* out, int D, const double* p, int st, const kabc_slotwin_t* w) {\n"
            "    kabc_user_mvprior_rand(out, D, p, st, w);\n}\n")
    tag = hashlib.sha1(text.encode()).hexdigest()[:16]
    bdir = os.path.join(_HERE, "_build")
    os.makedirs(bdir, exist_ok=True)
    so = os.path.join(bdir, f"libusermvprior_{tag}.so")
    if not os.path.exists(so):
        src = os.path.join(bdir, f"usermvprior_{tag}.c")
        with open(src, "w") as f:
            f.write(text)
        subprocess.check_call(["gcc", "-O2", "-std=gnu11", "-fPIC", "-ffp-contract=off", "-mfma",
                               "-I", os.path.join(os.path.dirname(_HERE), "include"), "-shared",
                               "-o", so, src, "-lm"])
    lib = _user_libs.get(so) or C.CDLL(so)
    _user_libs[so] = lib
    _check(load().orc_register_user_mvprior(fac.kind, C.cast(lib.orc_user_mvprior_logpdf_entry, C.c_void_p),
                                            C.cast(lib.orc_user_mvprior_rand_entry, C.c_void_p)))
    _user_prior_done.add(fac.kind)


def _prior_c(fac):
    """kabc_prior_t[D] for the ORACLE: a full-covariance MvNormal is registered with the oracle's
    own registry and handed over resolved (p[1] = k, p[2] = the block's address as a double's
    bits, p[3] = D; include/kabc_mvnormal.h) -- the library resolves its own handles itself."""
    if hasattr(fac, "source") and getattr(fac, "kind", 0) >= cd.PRIOR_USER:   # a joint user prior
        register_user_mvprior(fac)
    for c in fac.p:   # user families: the oracle gets the same snippet, compiled for the host
        if hasattr(c, "source") and c.kind >= cd.PRIOR_USER:
            register_user_prior(c)
    if getattr(fac, "cov", None) is None:
        return fac.to_c()
    L = load()
    h = getattr(fac, "_oracle_handle", None)
    if h is None:
        hh = C.c_int32()
        _check(L.orc_mvnormal_register(_dp(fac.mu), _dp(fac.cov), len(fac), C.byref(hh)))
        h = fac._oracle_handle = int(hh.value)
    blk = np.array([L.orc_mvnormal_block(h)], dtype=np.uint64).view(np.float64)[0]
    arr = (cd.Prior * len(fac))()
    for k in range(len(fac)):
        arr[k] = cd.Prior(cd.PRIOR_MVNORMAL, 0, (C.c_double * 4)(float(h), float(k), blk, float(len(fac))))
    return arr


def _model_c(model):
    m = model.to_c()
    _prior_c(model.prior)   # (registers user families with the oracle)
    if getattr(model.prior, "cov", None) is not None:
        model._prior_c_oracle = _prior_c(model.prior)
        m.prior = C.cast(model._prior_c_oracle, C.POINTER(cd.Prior))
    return m


MATH_FN = {"log": 0, "exp": 1, "log1p": 2, "lgamma": 3, "sincos2pi": 4, "sqrt": 5, "rint": 6,
           "log_pn": 7, "sqrt_pn": 8, "u01": 9, "normal_pair": 10, "index32": 11,
           "exp_bounded": 12}
MATH_IN_W = {"normal_pair": 2, "index32": 2}
MATH_OUT_W = {"sincos2pi": 2, "normal_pair": 2}


def math_vec(name, x):
    """x: float64 array (for u01 / normal_pair / index32 the 64-bit words are passed as
    the bit patterns of doubles: use .view(np.float64) on a uint64 array)."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    n = x.size // MATH_IN_W.get(name, 1)
    ow = MATH_OUT_W.get(name, 1)
    out = np.empty(n * ow)
    load().orc_math_vec(MATH_FN[name], n, _dp(x), _dp(out))
    return out.reshape(-1, 2) if ow == 2 else out


def div_rc(x, c):
    x = np.ascontiguousarray(x, dtype=np.float64)
    c = np.ascontiguousarray(np.broadcast_to(c, x.shape), dtype=np.float64)
    out = np.empty_like(x)
    load().orc_div_rc_vec(x.size, _dp(x), _dp(c), _dp(out))
    return out


def philox(ctr, key, rounds=10):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    load().orc_philox4x32_r(c, k, rounds, o)
    return list(o)


def philox_rounds():
    return load().orc_philox_rounds()


def normal_pairs(r):
    r = np.ascontiguousarray(r, dtype=np.uint64)
    out = np.empty(r.size)
    load().orc_normal_pairs(r.size // 2, r.ctypes.data_as(C.POINTER(C.c_uint64)), _dp(out))
    return out.reshape(-1, 2)


def _rows(prior, x):
    fac = as_factored(prior)
    a = np.ascontiguousarray(np.asarray(x, dtype=np.float64)).reshape(-1, len(fac))
    return fac, a


def factored_logpdf(prior, x):
    fac, a = _rows(prior, x)
    out = np.empty(a.shape[0])
    _check(load().orc_factored_logpdf(_prior_c(fac), len(fac), a.shape[0], _dp(a), _dp(out)))
    return out


def factored_pdf(prior, x):
    fac, a = _rows(prior, x)
    out = np.empty(a.shape[0])
    _check(load().orc_factored_pdf(_prior_c(fac), len(fac), a.shape[0], _dp(a), _dp(out)))
    return out


def push_p(prior, x):
    fac, a = _rows(prior, x)
    out = np.empty_like(a)
    _check(load().orc_push_p(_prior_c(fac), len(fac), a.shape[0], _dp(a), _dp(out)))
    return out


def factored_rand(prior, n, seed=0, domain=cd.DOM_AIS_INIT, first_walker=0, attempt=0):
    fac = as_factored(prior)
    out = np.empty((n, len(fac)))
    _check(load().orc_factored_rand(_prior_c(fac), len(fac), seed, domain, first_walker, n, attempt,
                                    _dp(out)))
    return out


def cost_eval(cost, x, seed=0, walker=0, t=0, domain=cd.DOM_AIS_COST):
    x = np.ascontiguousarray(x, dtype=np.float64)
    cc = cost.to_c()
    return load().orc_cost_eval(C.byref(cc), x.size, _dp(x), seed, walker, t, domain)


_user_libs = {}


def register_user_cost(cost):
    """Compile the C snippet of a kissabc_jl_amd.costs.UserCost with gcc and register
    it under the same id, so that the oracle evaluates the same user function."""
    import hashlib
    text = ('#include "kabc_philox.h"\n' + cost.source +
            "\ndouble orc_user_cost_entry(const double* x, int D, const double* params, "
            "const double* data, int64_t ndata, kabc_cost_rng_t* rng) {\n"
            "    return kabc_user_cost(x, D, params, data, ndata, rng);\n}\n"
            "#ifdef KABC_USER_SAMPLE_INIT\n"
            "void orc_user_init_entry(double* x, int D, const double* params, const double* data, "
            "int64_t ndata, kabc_cost_rng_t* rng) {\n"
            "    kabc_user_sample_init(x, D, params, data, ndata, rng);\n}\n#endif\n")
    tag = hashlib.sha1(text.encode()).hexdigest()[:16]
    bdir = os.path.join(_HERE, "_build")
    os.makedirs(bdir, exist_ok=True)
    so = os.path.join(bdir, f"libuser_{tag}.so")
    if not os.path.exists(so):
        src = os.path.join(bdir, f"user_{tag}.c")
        with open(src, "w") as f:
            f.write(text)
        subprocess.check_call(["gcc", "-O2", "-std=gnu11", "-fPIC", "-ffp-contract=off", "-mfma",
                               "-I", os.path.join(os.path.dirname(_HERE), "include"), "-shared",
                               "-o", so, src, "-lm"])
    lib = _user_libs.get(so) or C.CDLL(so)
    _user_libs[so] = lib
    fn = C.cast(lib.orc_user_cost_entry, C.c_void_p)
    _check(load().orc_register_user_cost(cost.id, fn))
    if hasattr(lib, "orc_user_init_entry"):
        _check(load().orc_register_user_init(cost.id, C.cast(lib.orc_user_init_entry, C.c_void_p)))


def cdf_g_inv(u, a):
    return load().orc_cdf_g_inv(u, a)


def quantile(v, p):
    v = np.ascontiguousarray(v, dtype=np.float64)
    out = C.c_double()
    _check(load().orc_quantile(_dp(v), v.size, p, C.byref(out)))
    return out.value


class OracleAIS:
    """CPU restatement of AIS(N) on a model (serial and sync schedules)."""

    def __init__(self, model, nparticles, seed=0):
        self.model, self.N, self.D = model, int(nparticles), len(model)
        self._cm = _model_c(model)
        self._h = C.c_void_p()
        _check(load().orc_ais_create(C.byref(self._cm), self.N, seed, C.byref(self._h)))

    def init(self, retry_sampling=100):
        _check(load().orc_ais_init(self._h, retry_sampling))
        return self

    def steps_serial(self, nsteps, ntransitions=1, collect=True):
        out = np.empty((nsteps, self.D)) if collect else None
        _check(load().orc_ais_steps_serial(self._h, nsteps, ntransitions,
                                           _dp(out) if collect else None))
        return out

    def generations_sync(self, ngen, ntransitions=1, collect=True, trace=False):
        out = np.empty((ngen, self.N, self.D)) if collect else None
        tr = (TraceRec * (ngen * self.N * ntransitions))() if trace else None
        _check(load().orc_ais_generations_sync(self._h, ngen, ntransitions,
                                               _dp(out) if collect else None, tr))
        if trace:
            t = np.frombuffer(tr, dtype=np.int32).reshape(ngen, self.N, ntransitions, 6).copy()
            return out, t
        return out

    def half_generation(self, half, ntransitions, row_begin, row_end):
        _check(load().orc_ais_half_generation(self._h, half, ntransitions, row_begin, row_end))

    def end_generation(self, ntransitions):
        _check(load().orc_ais_end_generation(self._h, ntransitions))

    def state(self):
        x = np.empty((self.N, self.D))
        lp = np.empty(self.N)
        ll = np.empty(self.N)
        t = C.c_uint64()
        _check(load().orc_ais_get_state(self._h, _dp(x), _dp(lp), _dp(ll), C.byref(t)))
        return x, lp, ll, t.value

    def set_state(self, x, lp, ll, t=0):
        x = np.ascontiguousarray(x, dtype=np.float64)
        lp = np.ascontiguousarray(lp, dtype=np.float64)
        ll = np.ascontiguousarray(ll, dtype=np.float64)
        _check(load().orc_ais_set_state(self._h, _dp(x), _dp(lp), _dp(ll), t))

    def stats(self):
        st = cd.Stats()
        _check(load().orc_ais_get_stats(self._h, C.byref(st)))
        return {"proposals": st.proposals, "cost_evals": st.cost_evals, "accepted": st.accepted}

    def close(self):
        if self._h:
            load().orc_ais_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def smc(prior, cost, *, nparticles=100, alpha=0.95, mcmc_retrys=0, mcmc_tol=0.015, epstol=0.0,
        r_epstol=None, min_r_ess=None, max_stretch=2.0, verbose=False, seed=0):
    """CPU restatement of smc() (src/smc.jl:92-206); returns dict."""
    fac = as_factored(prior)
    o = cd.SmcOpts()
    o.nparticles, o.alpha, o.mcmc_retrys, o.verbose = int(nparticles), alpha, mcmc_retrys, int(verbose)
    o.mcmc_tol, o.epstol, o.max_stretch, o.seed = mcmc_tol, epstol, max_stretch, seed
    o.r_epstol = math.nan if r_epstol is None else r_epstol
    o.min_r_ess = math.nan if min_r_ess is None else min_r_ess
    o.max_iterations = 0
    N, D = int(nparticles), len(fac)
    theta = np.empty((max(N, 1), D))
    Cst = np.empty(max(N, 1))
    alive = np.zeros(max(N, 1), dtype=np.uint8)
    log = (cd.SmcIter * 4096)()
    r = cd.SmcResult()
    r.theta, r.cost = _dp(theta), _dp(Cst)
    r.alive = alive.ctypes.data_as(C.POINTER(C.c_uint8))
    r.iter_log, r.iter_log_cap = log, 4096
    cc = cost.to_c()
    _check(load().orc_smc_run(_prior_c(fac), D, C.byref(cc), C.byref(o), C.byref(r)))
    nit = min(r.iterations, 4096)
    return {
        "theta_all": theta, "C": Cst, "alive": alive.astype(bool), "eps": r.eps,
        "P": theta[alive.astype(bool)], "iterations": r.iterations, "n_alive": r.n_alive,
        "cost_evals": r.cost_evals, "proposals": r.proposals,
        "log": [dict(eps=log[i].eps, ess=log[i].ess, accepted=log[i].accepted,
                     resampled=log[i].resampled, flag=log[i].flag, passes=log[i].mcmc_passes)
                for i in range(nit)],
    }


def abcde(prior, cost, eps_target, *, nparticles=50, generations=20, alpha=0.0, earlystop=False,
          proposal_width=1.0, seed=0):
    """CPU restatement of ABCDE (src/smc.jl:347-430); returns dict."""
    fac = as_factored(prior)
    o = cd.AbcdeOpts()
    o.nparticles, o.generations, o.eps_target, o.alpha = int(nparticles), int(generations), eps_target, alpha
    o.proposal_width, o.earlystop, o.verbose, o.seed = proposal_width, int(earlystop), 0, seed
    N, D = max(int(nparticles), 1), len(fac)
    theta = np.empty((N, D))
    Cst = np.empty(N)
    r = cd.AbcdeResult()
    r.theta, r.cost = _dp(theta), _dp(Cst)
    cc = cost.to_c()
    _check(load().orc_abcde_run(_prior_c(fac), D, C.byref(cc), C.byref(o), C.byref(r)))
    return {"P": theta, "C": Cst, "reached_eps": bool(r.reached_eps),
            "generations_run": r.generations_run, "nsims": r.nsims}


def pfilter(prior, cost, N, *, q=0.7, eff_tol=0.1, epstol=-math.inf, max_iters=math.inf,
            proposal_width=0.75, seed=0):
    """CPU restatement of pfilter (src/smc.jl:275-340); returns dict."""
    fac = as_factored(prior)
    o = cd.PfilterOpts()
    o.nparticles, o.q, o.eff_tol, o.epstol = int(N), q, eff_tol, epstol
    o.proposal_width, o.verbose, o.seed = proposal_width, 0, seed
    o.max_iters = -1 if math.isinf(max_iters) else int(math.floor(max_iters))
    D = len(fac)
    n_eff = load().orc_pfilter_nparticles(int(N), q, D)
    theta = np.empty((n_eff, D))
    Cst = np.empty(n_eff)
    r = cd.PfilterResult()
    r.theta, r.cost = _dp(theta), _dp(Cst)
    cc = cost.to_c()
    _check(load().orc_pfilter_run(_prior_c(fac), D, C.byref(cc), C.byref(o), C.byref(r)))
    return {"P": theta, "C": Cst, "eps": r.eps, "eff": r.eff, "iterations": r.iterations,
            "nreps": r.nreps, "cost_evals": r.cost_evals}
