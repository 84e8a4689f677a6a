/*
 * kabc.h -- C ABI of the MI355X (gfx950) walker-update path of KissABC.
 *
 * The reference (KissABC.jl v3.0.1) has NO FFI: its "operator API" is the Julia
 * method contract of AbstractDensity (src/types.jl:3-8) driven by
 * AbstractMCMC.step (src/KissABC.jl:35-80) and by smc() (src/smc.jl:92-206).
 * Each entry point below cites the reference interface it replaces; the Julia
 * `ccall` stub a maintainer would add is shown in INTEGRATION.md and shipped as
 * kissabc.jl_amd/julia/KissABCHip.jl.
 *
 * Conventions: extern "C", plain pointers and sizes, no torch/HIP types in the
 * signatures (a HIP stream crosses as void*).  Every function returns a
 * kabc_status_t; kabc_last_error() returns a thread-local message that carries
 * the reference's own error text where the reference raises one.  Host buffers
 * are caller-owned; device memory is library-owned behind opaque handles unless
 * the caller lends device buffers explicitly (sharded mode).  A handle is
 * single-threaded; distinct handles may be used from distinct threads (each
 * owns a HIP stream) -- the MCMCThreads analogue (src/KissABC.jl:108).
 */
#ifndef KABC_H
#define KABC_H

#ifndef __HIPCC_RTC__
#include <stddef.h>
#include <stdint.h>
#else /* hipRTC has no system headers */
#include "kabc_rtc_types.h"
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define KABC_VERSION 321 /* 0.3.3: SmcDynArgs (shared with hipcc-built cost plugins) carries the particle range of a pass: sharded smc beyond KABC_MAX_DIM; 0.3.2: kabc_smc_dist_stats, kabc_ais_driver, kabc_set_specialize, kabc_rtc_cache_dir, kabc_compile_mvprior_plugin; 0.3.1: kernel argument structs shared with hipcc-built cost plugins changed (PfArgs, AbcdeArgs, PfCtrl, kabc_cost_rng_t); kabc_register_cost_plugin refuses a plugin built against another value */
#define KABC_MAX_DIM 16  /* length(prior) up to which the register-resident kernels are instantiated */
/* AIS, smc, ABCDE and pfilter accept length(prior) up to KABC_MAX_DIM_DYN: beyond KABC_MAX_DIM
 * run-time-dimension kernels keep the walker / particle rows in memory (several times slower per
 * evaluation, same results).  Run-time compiled user costs and user prior families follow; a model's own
 * specialised kernels stop at KABC_MAX_DIM.  The reference has no bound (src/priors.jl:10-13). */
#define KABC_MAX_DIM_DYN 256
/* AIS ensemble size: nparticles < 2^31 and (nparticles / 2) * length(prior) * 8 bytes < 4 GiB
 * (partner rows are addressed by a 32-bit byte offset into their half: 134 M walkers at
 * D = 8); beyond that kabc_ais_create* return KABC_ERR_UNSUPPORTED. */

typedef enum kabc_status {
    KABC_OK = 0,
    KABC_ERR_INVALID_ARG = 1,     /* reference: error(...) on argument checks            */
    KABC_ERR_RETRY_EXHAUSTED = 2, /* src/KissABC.jl:58-59                                */
    KABC_ERR_INVALID_STATE = 3,   /* src/types.jl:70 "starting sample invalid."         */
    KABC_ERR_DEVICE = 4,          /* HIP runtime error / no gfx950 device / no kernels   */
    KABC_ERR_UNSUPPORTED = 5,     /* model outside the DeviceCost / prior surface        */
    KABC_ERR_NAN_COST = 6         /* Statistics.quantile: "undefined in presence of NaNs" */
} kabc_status_t;

/* ---- Factored prior surface (src/priors.jl:10-49) ------------------------ */
typedef enum kabc_prior_kind {
    KABC_PRIOR_UNIFORM = 1,          /* Uniform(a,b)            p = (a, b)            */
    KABC_PRIOR_NORMAL = 2,           /* Normal(mu,sigma)        p = (mu, sigma)       */
    KABC_PRIOR_TRUNCNORMAL = 3,      /* Truncated(Normal(mu,sigma), lo, hi) p = (mu, sigma, lo, hi) */
    KABC_PRIOR_BETA = 4,             /* Beta(alpha,beta)        p = (alpha, beta)     */
    KABC_PRIOR_DISCRETE_UNIFORM = 5, /* DiscreteUniform(a,b)    p = (a, b)  [discrete] */
    KABC_PRIOR_NEGBINOMIAL = 6,      /* NegativeBinomial(r,p)   p = (r, p)  [discrete] */
    KABC_PRIOR_EXPONENTIAL = 7,      /* Exponential(theta)      p = (theta)           */
    KABC_PRIOR_GAMMA = 8,            /* Gamma(alpha, theta)     p = (alpha, theta)    */
    KABC_PRIOR_LOGNORMAL = 9,        /* LogNormal(mu, sigma)    p = (mu, sigma)       */
    /* CommonLogDensity(nparameters, sample_init, lπ) with an ARBITRARY sample_init
     * (src/types.jl:105-113: `rng -> sample`): the initial walkers are drawn by the cost
     * plugin's own kabc_user_sample_init (include/kabc_costs.h).  Every component of the
     * "prior" carries this kind; it has no density (KABC_POSTERIOR_COMMON never asks for one). */
    KABC_PRIOR_USER_INIT = 10,
    /* Component k of a full-covariance MvNormal(mu, Sigma) prior (the reference takes any
     * `Distribution`: src/types.jl:30, :34-35, :52; src/smc.jl:92): p = (handle, k) with the
     * handle of kabc_mvnormal_register below.  All D components of the prior carry this kind,
     * the same handle and k = their index; D <= KABC_MAX_DIM.  include/kabc_mvnormal.h. */
    KABC_PRIOR_MVNORMAL = 11,
    KABC_PRIOR__COUNT = 12,
    /* kinds >= KABC_PRIOR_USER: families compiled at run time from a C snippet
     * (kabc_compile_prior_plugin below) -- the reference's Factored takes ANY
     * UnivariateDistribution (src/priors.jl:11). */
    KABC_PRIOR_USER = 100
} kabc_prior_kind_t;

/* one univariate component of Factored(...) */
typedef struct kabc_prior {
    int32_t kind; /* kabc_prior_kind_t */
    int32_t reserved;
    double p[4];
} kabc_prior_t;

/* DeviceCost: replaces the `cost` closure (src/types.jl:42,55; src/smc.jl:94).
 * ids and formulas: include/kabc_costs.h */
typedef struct kabc_cost {
    int32_t id;
    int32_t nparams;
    const double* params; /* host pointer, copied at create */
    int64_t ndata;
    const double* data; /* host pointer, copied at create */
} kabc_cost_t;

typedef enum kabc_posterior_kind {
    KABC_POSTERIOR_KERNELIZED = 1, /* ApproxKernelizedPosterior, src/types.jl:40-75; eps = scale   */
    KABC_POSTERIOR_THRESHOLD = 2,  /* ApproxPosterior,           src/types.jl:76-104; eps = maxcost */
    /* CommonLogDensity(nparameters, sample_init, lπ), src/types.jl:105-128: plain MCMC on a
     * log-density.  `cost` IS lπ (returns the log-density), `prior` describes sample_init
     * (used only by step(init)); no push_p, no prior term, eps unused. */
    KABC_POSTERIOR_COMMON = 3
} kabc_posterior_kind_t;

/* ApproxKernelizedPosterior(prior, cost, scale) / ApproxPosterior(prior, cost, maxcost) */
typedef struct kabc_model {
    const kabc_prior_t* prior; /* D components */
    int32_t D;                 /* length(prior), 1..KABC_MAX_DIM (AIS: ..KABC_MAX_DIM_DYN) */
    int32_t posterior;         /* kabc_posterior_kind_t */
    double eps;                /* scale (kernelized) or maxcost (threshold) */
    kabc_cost_t cost;
} kabc_model_t;

/* counters the metric is computed from */
typedef struct kabc_stats {
    uint64_t proposals;  /* transition! calls                                   */
    uint64_t cost_evals; /* cost closure calls (skipped when logprior = -Inf)   */
    uint64_t accepted;   /* accepted transitions                                */
} kabc_stats_t;

typedef struct kabc_ctx kabc_ctx_t;
typedef struct kabc_ais kabc_ais_t;

int32_t kabc_version(void);
/* sizeof() of the i-th struct of this header as the library was compiled, in declaration order:
 * 0 kabc_prior_t, 1 kabc_cost_t, 2 kabc_model_t, 3 kabc_stats_t, 4 kabc_smc_opts_t,
 * 5 kabc_smc_iter_t, 6 kabc_smc_result_t, 7 kabc_abcde_opts_t, 8 kabc_abcde_result_t,
 * 9 kabc_pfilter_opts_t, 10 kabc_pfilter_result_t; -1 beyond.  A binding that mirrors the
 * structs by hand (ctypes, Julia `struct`) checks itself against this at load time. */
int32_t kabc_abi_sizeof(int32_t which);
/* offsetof() of the field-th member (declaration order, from 0) of the which-th struct (the
 * numbering of kabc_abi_sizeof) as the library was compiled; -1 beyond.  Together with
 * kabc_abi_sizeof this pins a hand-written mirror field by field (tests/test_julia_shim_static.py
 * checks julia/KissABCHip.jl and kissabc.jl_amd/_cdefs.py against it). */
int32_t kabc_abi_offsetof(int32_t which, int32_t field);
const char* kabc_last_error(void);
/* number of visible gfx950 devices (0 when none; never an error) */
int32_t kabc_device_count(void);

/* device context: selects the GPU, owns one HIP stream.  stream = NULL creates
 * a private stream; otherwise the caller's hipStream_t is used (e.g. torch's
 * current stream so that RCCL collectives issued by the host order correctly). */
kabc_status_t kabc_ctx_create(int32_t device_id, void* stream, kabc_ctx_t** out);
kabc_status_t kabc_ctx_destroy(kabc_ctx_t* ctx);
kabc_status_t kabc_ctx_synchronize(kabc_ctx_t* ctx);

/* Page-locked host memory for the buffers the caller hands to kabc_ais_advance
 * (out_samples): the sample trace is then DMA'd over PCIe while the next
 * generations compute.  Any other host pointer is accepted too (the runtime stages
 * pageable copies, several times slower).  The reference returns its samples in
 * GC-managed Julia arrays (src/KissABC.jl:82-94); this pair is what the host shim
 * allocates them with. */
kabc_status_t kabc_host_alloc(size_t bytes, void** out);
kabc_status_t kabc_host_free(void* p);

/* ---- Factored utilities (device kernels; host in, host out) ----------------
 * logpdf(d::Factored, x) for n rows x[n][D]      -- src/priors.jl:30-36
 * push_p(d::Factored, x)                          -- src/types.jl:29-32
 * rand(rng, d::Factored) for walkers first..first+n of stream (seed, domain, attempt)
 *                                                 -- src/priors.jl:42-43 */
/* MvNormal(mu, Sigma): mu[D], Sigma[D*D] row-major, symmetric positive definite, 1 <= D <=
 * KABC_MAX_DIM.  The library keeps the Cholesky factor, its inverse and the constants
 * (process lifetime); *handle goes into kabc_prior_t.p[0] of D components of kind
 * KABC_PRIOR_MVNORMAL (p[1] = component index).  A diagonal Sigma needs none of this: it is
 * Factored(Normal(mu_k, sqrt(Sigma_kk))...). */
kabc_status_t kabc_mvnormal_register(const double* mu, const double* cov, int32_t D, int32_t* handle);

kabc_status_t kabc_factored_logpdf(kabc_ctx_t* ctx, const kabc_prior_t* prior, int32_t D,
                                   int64_t n, const double* x, double* out);
kabc_status_t kabc_factored_push_p(kabc_ctx_t* ctx, const kabc_prior_t* prior, int32_t D,
                                   int64_t n, const double* x, double* out);
kabc_status_t kabc_factored_rand(kabc_ctx_t* ctx, const kabc_prior_t* prior, int32_t D,
                                 uint64_t seed, uint32_t domain, int64_t first_walker, int64_t n,
                                 uint64_t attempt, double* out);

/* ---- arithmetic-contract probe (verification only) ---------------------------
 * Evaluates one function of include/kabc_math.h on the device for n host inputs, so
 * that tests can compare the gfx950 code with the host build of the same header bit
 * for bit.  fn: 0 log, 1 exp, 2 log1p, 3 lgamma, 4 sincos2pi (out[2n]), 5 sqrt,
 * 6 rint, 7 log_pn, 8 sqrt_pn, 9 u01 (x = 64 random bits), 10 normal_pair
 * (x = pairs of 64-bit words, out[2n]), 11 index32 (x = pairs (bits, n), out as double). */
kabc_status_t kabc_math_probe(kabc_ctx_t* ctx, int32_t fn, int64_t n, const double* x,
                              double* out);

/* ---- user DeviceCost plugins ------------------------------------------------
 * Replaces "cost is an arbitrary closure" (src/types.jl:42,55; src/smc.jl:94) for
 * costs that can be written as a C function (signature: include/kabc_costs.h,
 * KABC_COST_USER).  `path` is a shared library built from the user's snippet +
 * kissabc.jl_amd/csrc/user_plugin.inc with hipcc --offload-arch=gfx950; on success
 * *out_cost_id (>= 100) is the id to put into kabc_cost_t.id. */
kabc_status_t kabc_register_cost_plugin(const char* path, int32_t* out_cost_id);
/* The same, without hipcc and without a file: `src` (the snippet -- KABC_HD double
 * kabc_user_cost(...), optionally KABC_USER_AUX_WORDS + kabc_user_cost_prepare) is compiled IN
 * PROCESS by hipRTC for gfx950.  dims[ndims]: the values of length(prior) the cost accepts
 * (1..KABC_MAX_DIM); posterior_mask: bit (kind - 1) per kabc_posterior_kind_t the cost will be
 * used with, 0 = all.  The snippet itself is compiled at once (its errors come back here, with
 * the compiler's message in kabc_last_error()); the kernels are compiled at first use, one
 * family and dimension at a time (about 1-3 s each: kabc_ais_create / kabc_smc_run / ... of
 * the first model that uses the cost).  SURVEY 8f-1; src/types.jl:42,55. */
kabc_status_t kabc_compile_cost_plugin(const char* src, const int32_t* dims, int32_t ndims,
                                       int32_t posterior_mask, int32_t* out_cost_id);
/* Compile (and load on the current device) one kernel family of a user cost ahead of its first
 * use.  family: 0 AIS half-generation (variant = prior class + 4 * (posterior kind - 1); prior
 * class 0 box, 1 constant/Gaussian-in-a-box, 2 general), 1 AIS init, 2 smc propose+accept
 * (variant = 1 for priors without Beta / Gamma / LogNormal / NegativeBinomial components, else
 * 0), 3 smc init, 4 smc persistent loop, 5 / 6 ABCDE init / generation, 7 pfilter attempt,
 * 10 the one-workgroup smc driver, 13 the one-workgroup AIS driver of small ensembles (variant as
 * family 0; prior classes 0 and 2). */
kabc_status_t kabc_plugin_precompile(int32_t cost_id, int32_t family, int32_t D, int32_t variant);

/* ---- user prior families ------------------------------------------------------
 * The reference's Factored is a tuple of ANY UnivariateDistribution: `logpdf`, `rand` and the
 * continuous / discrete split of push_p dispatch on each component's type (src/priors.jl:11,
 * :31-33, :43; src/types.jl:30-32).  The families of kabc_prior_kind_t are built in; any other
 * one is a C snippet defining
 *   KABC_HD double kabc_user_prior_logpdf(double x, const double* p, const double* tab);
 *   KABC_HD double kabc_user_prior_rand(const double* p, const kabc_slotwin_t* w);
 * x: the coordinate after push_p (rounded when the family is discrete); p: the component's
 * kabc_prior_t.p[4]; -Inf outside the support; tab: the table kabc_log_t / kabc_log1p_t /
 * kabc_lgamma_t of kabc_math.h take (the calling kernel's LDS copy).  rand draws from the
 * component's window of the counter stream: kabc_slot(w, j), j < KABC_SLOTS_PER_DIM, and the
 * helpers of kabc_sampling_base.h (kabc_sample_gamma1, kabc_sample_poisson).  Constants that are
 * expensive to derive (a truncation's log-mass, a normaliser's lgamma) belong into the snippet
 * text as literals: the host computes them once, as Distributions.jl does at construction.
 * `discrete` != 0: push_p rounds the coordinate (round(Int, .), src/types.jl:32).
 * The snippet alone is compiled at once by hipRTC (errors come back here with the compiler's
 * message).  *out_kind (>= KABC_PRIOR_USER) goes into kabc_prior_t.kind.  A prior with such a
 * component has no prebuilt kernels: every entry point that receives one compiles the kernel
 * family it needs for that (prior kinds, cost) pair at first use (2-20 s by prior class, kept in an on-disk
 * cache of code objects: KABC_RTC_CACHE_DIR, default next to the library) -- the way a user
 * cost does.  length(prior) <= KABC_MAX_DIM for such priors. */
kabc_status_t kabc_compile_prior_plugin(const char* src, int32_t discrete, int32_t* out_kind);

/* ---- joint (multivariate) user priors ---------------------------------------------
 * The reference hands ANY Distribution to rand / logpdf as the prior -- a multivariate one as well
 * (src/types.jl:30,34-35,52: unconditional_sample = rand(rng, prior), loglike calls logpdf(prior, x);
 * src/smc.jl:92-93).  A joint density that is not a product of univariate ones (a Dirichlet, an AR(1)
 * process, a copula) is a C snippet defining
 *   KABC_HD double kabc_user_mvprior_logpdf(const double* x, int D, const double* p, int pstride,
 *                                           const double* tab);
 *   KABC_HD void   kabc_user_mvprior_rand(double* out, int D, const double* p, int pstride,
 *                                         const kabc_slotwin_t* w);
 * x: the whole parameter vector (a joint prior is continuous: push_p is the identity); component k's
 * parameters are p[k * pstride + 0..2] = kabc_prior_t.p[0..2] of component k (p[3] belongs to the
 * library); -Inf outside the support; tab as for the univariate families.  rand fills out[0..D) from
 * the walker's window w: kabc_slot(w, j), j < D * KABC_SLOTS_PER_DIM, and the helpers of
 * kabc_sampling_base.h.  *out_kind (>= KABC_PRIOR_USER) goes into the kind of ALL D components of the
 * prior (it does not mix with other components).  Everything else is as for the univariate user
 * families: compiled at once by hipRTC, no prebuilt kernels, the kernel families of a (prior, cost)
 * pair compiled at first use and cached; length(prior) up to KABC_MAX_DIM_DYN. */
kabc_status_t kabc_compile_mvprior_plugin(const char* src, int32_t* out_kind);

/* ---- kernels specialised for ONE model ------------------------------------------
 * Compiles the kernel families of `families` (bit 0 AIS, 1 smc, 2 ABCDE, 3 pfilter, 4 AIS of small
 * ensembles; 0 = AIS + smc) for exactly this prior tuple and cost: every component's family and parameters are
 * compile-time constants of the generated translation unit -- no family dispatch, no
 * per-component parameter records in LDS, the normalisers folded.  Results are bit-identical to
 * the prebuilt kernels' (same formulas, same operation order).  Afterwards kabc_ais_create* /
 * kabc_smc_run / kabc_abcde_run / kabc_pfilter_run use the specialised kernels whenever they
 * receive the same prior components (bit for bit), D and cost id; model->posterior and ->eps do
 * not take part (the posterior kind is a template parameter of the AIS kernel, chosen at
 * kabc_ais_create).  Priors made of Uniform / DiscreteUniform components only are left to the
 * prebuilt kernels (their box test is already parameter-free); so are priors of plain Normals
 * up to seven parameters (the prebuilt class is the faster kernel there), full-covariance
 * MvNormal priors and length(prior) > KABC_MAX_DIM.  KABC_SPECIALIZE=1 in the environment makes
 * every entry point specialise on its own at first sight of a model.  Without hipRTC (or with
 * KABC_SPECIALIZE=0) the prebuilt kernels remain the path.  *out_handle (optional) identifies
 * the registration for kabc_model_release. */
#define KABC_FAMILY_AIS 1
#define KABC_FAMILY_SMC 2
#define KABC_FAMILY_ABCDE 4
#define KABC_FAMILY_PFILTER 8
#define KABC_FAMILY_AIS_SMALL 16 /* the AIS kernel of small ensembles (kabc_ais_driver) */
kabc_status_t kabc_compile_model(const kabc_model_t* model, int32_t families, int32_t* out_handle);
kabc_status_t kabc_model_release(int32_t handle);
/* THE DEFAULT (KABC_SPECIALIZE unset): kabc_ais_create* / kabc_smc_run / kabc_abcde_run /
 * kabc_pfilter_run specialise an eligible model ON THEIR OWN and never wait for the compiler --
 * the reference gets the same from Julia's per-type compilation of logpdf(::Factored)
 * (src/priors.jl:11,30-36).  The call starts on the prebuilt kernels; the model's unit is
 * compiled by a detached worker process (<library directory>/kabc_rtc_worker, KABC_RTC_WORKER)
 * into the on-disk cache of code objects (KABC_RTC_CACHE_DIR); an AIS handle looks for it at its
 * launch boundaries and switches kernels when it is there, the run-to-completion entry points
 * take it from their next call on.  The switch cannot be seen in the results (same bits).  A
 * unit found in the cache is loaded at once (~1 ms).  Without hipRTC, the worker or a writable
 * cache directory, and after a failed compilation, the prebuilt kernels stay, silently.
 * KABC_SPECIALIZE=1: compile at first sight, blocking; =0: never specialise. */
/* The same ahead of the first use, still without waiting: hands the units of `families` (as in
 * kabc_compile_model) to the worker unless they are in the cache already.  A no-op for models
 * that are not eligible. */
kabc_status_t kabc_prefetch_model(const kabc_model_t* model, int32_t families);
/* The same switch for a host application that embeds the library and must not have it fork
 * compiler processes (or must not depend on its environment): process-wide, takes effect for the
 * models seen from now on.  _ENV: KABC_SPECIALIZE decides (the default); _OFF: never specialise,
 * never start the worker (registered kabc_compile_model units are ignored too); _BLOCKING: compile
 * at first sight in the calling thread; _BACKGROUND: the worker process, whatever the environment
 * says.  The code-object cache the worker fills lives in a directory only this user can write to
 * (owned by the effective user, no group / world write permission, not a symbolic link:
 * KABC_RTC_CACHE_DIR, else <library directory>/rtc_cache, $XDG_CACHE_HOME/kabc_rtc_cache or
 * ~/.cache/kabc_rtc_cache, $TMPDIR/kabc_rtc_cache_<uid>); a directory that fails the test is not used,
 * and a cache file is loaded only when the digest of its unit and the checksum of its code match. */
#define KABC_SPECIALIZE_ENV (-1)
#define KABC_SPECIALIZE_OFF 0
#define KABC_SPECIALIZE_BLOCKING 1
#define KABC_SPECIALIZE_BACKGROUND 2
kabc_status_t kabc_set_specialize(int32_t mode);
/* the code-object cache directory in use ("" and 0: none -- disabled, or no candidate passed the
 * trust test): copies at most cap - 1 characters + NUL into out, returns the full length */
int32_t kabc_rtc_cache_dir(char* out, int32_t cap);
#define KABC_SPEC_NONE 0    /* prebuilt kernels: model not eligible, specialisation off or unavailable */
#define KABC_SPEC_PENDING 1 /* prebuilt (or generic) kernels while the worker compiles              */
#define KABC_SPEC_ACTIVE 2  /* the model's own kernels                                              */
#define KABC_SPEC_FAILED 3  /* the compilation failed: prebuilt (or generic) kernels for good       */
/* process-wide counters of the above: out[0] compilations handed to the worker, out[1] code
 * objects it delivered that were loaded, out[2] failures, out[3] units found in the cache at
 * first sight */
kabc_status_t kabc_spec_counters(uint64_t out[4]);
/* entry point of the worker process (csrc/rtc_worker.c); not for callers */
int32_t kabc_rtc_worker_main(const char* jobfile);

/* ---- AIS: sample(model, AIS(N), Ns; ntransitions, discard_initial, retry_sampling)
 *
 * Ensemble layout.  Walker ids g = 0..N-1.  Half 0 = ids [0, N0), half 1 = ids
 * [N0, N), N0 = ceil(N/2).  Each half is a row-major [rows][D] f64 array; the
 * log-density pair (logprior, loglikelihood|cost) of src/types.jl:57,90 is two
 * f64 arrays per half.
 *
 * Schedule.  The reference sweeps serially: step() gives walker i `ntransitions`
 * consecutive transition!() calls against the other, momentarily frozen walkers
 * (src/KissABC.jl:74-79).  Here one GENERATION does the same for every walker:
 * half 0 then half 1, each walker of the active half receiving `ntransitions`
 * consecutive transitions with partners drawn from the frozen complementary
 * half.  One generation therefore yields N samples = N reference step() calls.
 */

/* AIS(nparticles) bound to a model.  Replaces the sampler tag + model pair of
 * src/KissABC.jl:21-23,35-48.  Fails with the reference's message when
 * nparticles < length(model)+5 (src/KissABC.jl:43-48). */
kabc_status_t kabc_ais_create(kabc_ctx_t* ctx, const kabc_model_t* model, int64_t nparticles,
                              uint64_t seed, kabc_ais_t** out);

/* sample(model, AIS(N), MCMCThreads(), Ns, Nc) -- src/KissABC.jl:96-104,108: `nchains`
 * INDEPENDENT ensembles of nparticles walkers each in one handle, chain = a grid dimension of
 * every launch (50 x AIS(12) is 50 workgroups of one launch instead of 50 runs one after the
 * other).  Chain c uses seeds[c] and is bit-identical to a kabc_ais_create handle with that
 * seed.  Host layouts gain a leading chain axis: kabc_ais_advance's out_samples is
 * [ngenerations][nchains][N][D]; get/set_state and get_ensemble use [nchains][N][...]. */
kabc_status_t kabc_ais_create_batch(kabc_ctx_t* ctx, const kabc_model_t* model, int64_t nparticles,
                                    int32_t nchains, const uint64_t* seeds, kabc_ais_t** out);

/* Sharded variant: this process owns rows [rank*rows_h/world, (rank+1)*rows_h/world)
 * of each half of an ensemble of n_total walkers (n_total divisible by 2*world).
 * dev_half0 / dev_half1 are caller-provided DEVICE buffers of n_total/2 * D
 * doubles each (e.g. torch tensors) holding the GLOBAL halves: kernels update
 * the owned rows in place and read partners from any row; the host all-gathers
 * the owned segment after each kabc_ais_half_generation (one RCCL all-gather
 * per half).  Draws are keyed by global walker id, so results are identical for
 * every world size. */
kabc_status_t kabc_ais_create_sharded(kabc_ctx_t* ctx, const kabc_model_t* model,
                                      int64_t n_total, int32_t rank, int32_t world,
                                      uint64_t seed, void* dev_half0, void* dev_half1,
                                      kabc_ais_t** out);

/* step(rng, model, spl; retry_sampling) -- src/KissABC.jl:35-64: draw the owned
 * walkers from the prior, evaluate loglike, re-draw invalid ones; fails with
 * "Prior leads to ∞ costs too often, tune the prior or increase `retry_sampling`."
 * when more than retry_sampling * nparticles re-draws are needed. */
kabc_status_t kabc_ais_init(kabc_ais_t* h, int32_t retry_sampling);

/* Asynchronous: enqueue `ntransitions` transition!() calls (src/transition.jl:67-82)
 * for every owned walker of `half` on the context stream.  trace_row, if not
 * NULL, is a DEVICE pointer receiving push_p(x) of the owned rows of that half
 * ([rows_owned][D]) after the last transition (the sample step() returns,
 * src/KissABC.jl:78).  Advances nothing else; pair it with kabc_ais_end_generation. */
kabc_status_t kabc_ais_half_generation(kabc_ais_t* h, int32_t half, int32_t ntransitions,
                                       void* dev_trace_rows);
/* advance the transition counter by ntransitions after both halves ran */
kabc_status_t kabc_ais_end_generation(kabc_ais_t* h, int32_t ntransitions);

/* step(rng, model, spl, state; ntransitions) x N x ngenerations -- src/KissABC.jl:66-80.
 * Runs ngenerations generations (single-process handles only).  out_samples, if
 * not NULL, is a HOST buffer [ngenerations][N][D] receiving push_p(walker) in
 * walker-id order after each generation: exactly the samples the reference's
 * step() emits over N*ngenerations calls.  stats (optional) accumulates.
 * The trace is streamed out in chunks while later generations compute; hand over a
 * kabc_host_alloc buffer for the full PCIe rate (any host pointer works). */
kabc_status_t kabc_ais_advance(kabc_ais_t* h, int64_t ngenerations, int32_t ntransitions,
                               double* out_samples, kabc_stats_t* stats);

/* AISState (src/KissABC.jl:25-33) <-> host.  x: [N][D] unrounded positions in
 * walker-id order (owned rows only when sharded: [n_owned][D], half 0 rows
 * first); logprior, loglik: [N]; t = transitions done per walker. */
kabc_status_t kabc_ais_get_state(kabc_ais_t* h, double* x, double* logprior, double* loglik,
                                 uint64_t* t);
kabc_status_t kabc_ais_set_state(kabc_ais_t* h, const double* x, const double* logprior,
                                 const double* loglik, uint64_t t);
/* cumulative counters since create (device-side counters, read synchronously) */
kabc_status_t kabc_ais_get_stats(kabc_ais_t* h, kabc_stats_t* stats);
/* which kernels the handle's half-generation launches run on: *state = KABC_SPEC_*;
 * *launches_before_switch = launches that ran on the prebuilt kernels before the model's own
 * took over (0: specialised from the first launch; -1: not switched) */
kabc_status_t kabc_ais_spec_state(kabc_ais_t* h, int32_t* state, int64_t* launches_before_switch);
/* Which driver kabc_ais_advance runs this handle on: 1 = the one-workgroup kernel of small
 * ensembles (every generation of a call in ONE launch of one workgroup per chain, both halves in
 * LDS: csrc/ais_small_kernel.hpp -- nparticles <= 512, <= 256 from nine parameters on; the shape
 * of every sample() call in the reference's tests and examples, src/KissABC.jl:66-80,
 * test/runtests.jl:82-131), 0 = one launch per half-generation.  Same bits either way.
 * KABC_AIS_SMALL=0 in the environment of kabc_ais_create* keeps every handle on 0. */
int32_t kabc_ais_driver(const kabc_ais_t* h);
/* number of walkers this handle owns, and per half */
int64_t kabc_ais_owned(const kabc_ais_t* h, int32_t half);
/* Per-launch timing: bracket each of the next `max_launches` half-generation
 * kernels with a hipEvent pair on the context stream (0 disables).
 * kabc_ais_kernel_ms returns their average device time in ms and rewinds. */
kabc_status_t kabc_ais_set_timing(kabc_ais_t* h, int32_t max_launches);
/* one event pair brackets `stride` consecutive launches and the elapsed time is divided
 * by `stride` (a pair per launch adds ~3 us of marker overhead to each figure); default 1 */
kabc_status_t kabc_ais_set_timing_stride(kabc_ais_t* h, int32_t stride);
double kabc_ais_kernel_ms(kabc_ais_t* h, int64_t* nlaunches);
/* Exchange diagnostics of a sharded handle (kabc_ais_create_dist, one process per GPU) over the
 * half-generations kabc_ais_advance has run since kabc_ais_set_timing (at most 128 of them),
 * averages in MICROSECONDS per half-generation: out[0] compute -- the half's kernels on the context
 * stream; out[1] exchange -- the all-gather(s) of the half on the stream they run on (summed over
 * the exchange chunks); out[2] exposed -- how long the context stream then waits until the
 * gathered half is available to it (equal to the exchange when there is one chunk, what the
 * pipeline could not hide otherwise); out[3] the number of exchange chunks K.  Rewinds.  This is
 * what tells a slow collective from a slow kernel in a multi-GPU run (bench.py prints it). */
kabc_status_t kabc_ais_exchange_us(kabc_ais_t* h, double out[4]);
/* Test hook: record (move, accepted, a, b, c, cost_evaluated) as 6 int32 per
 * walker and sub-step of the NEXT generation ([N_owned][ntransitions][6], walker-id
 * order; partner ids are row indices inside the complementary half).  0 disables. */
kabc_status_t kabc_ais_set_debug(kabc_ais_t* h, int32_t ntransitions);
kabc_status_t kabc_ais_get_debug(kabc_ais_t* h, int32_t* out, int64_t n_int32);
kabc_status_t kabc_ais_destroy(kabc_ais_t* h);

/* ---- multi-GPU: walker-sharded AIS with ONE all-gather per half-generation ----------
 *
 * The reference has no device-to-device path: its only parallel legs are independent
 * chains (MCMCThreads / MCMCDistributed, src/KissABC.jl:9,108-109,175) and the threaded
 * cost loop of smc (src/smc.jl:120-123,168).  What follows is therefore new API, shaped
 * after the north star: the host (Julia `ccall`, C, Python ctypes) owns process
 * placement; the LIBRARY owns the collective -- RCCL (ncclAllGather over xGMI) loaded
 * from librccl.so at first use, issued on the context stream right behind the kernels.
 *
 * Walkers shard by row range: rank r owns rows [r*per_h, min((r+1)*per_h, rows_h)) of
 * each half, per_h = ceil(rows_h / world).  A half-generation needs no communication
 * (partners come from the frozen complementary half, which every rank holds in full);
 * afterwards the freshly updated rows are all-gathered in place so that the next
 * half-generation can draw partners from them.  Draws are keyed by GLOBAL walker id:
 * the trajectory is bit-identical for every world size.  Log-densities never travel.
 *
 * Pipelined exchange.  With K > 1 EXCHANGE CHUNKS the ownership is block-cyclic: chunk k of
 * a half is the row range [k*world*c, (k+1)*world*c), c = ceil(rows_h / (K*world)), and rank
 * r owns its r-th segment of c rows -- so the all-gather of chunk k is in place and
 * contiguous and runs on a second stream while the kernels of chunk k+1 compute; only the
 * last chunk's gather is exposed.  K is chosen at kabc_ais_create_dist: the environment
 * variable KABC_EXCHANGE_CHUNKS (1..KABC_MAX_EXCHANGE_CHUNKS), else one chunk per full
 * residency wave of the kernel (K = 1 up to 65 536 walkers per rank and half: a smaller
 * launch takes as long as a full one, so finer chunks would serialise the compute they are
 * meant to hide).  Results do not depend on K.  kabc_ais_owned_segments
 * reports the owned row ranges.
 *
 * Two ways to form the communicator:
 *   one process per GPU   kabc_comm_unique_id on rank 0 -> the host ships the 128 bytes
 *                         to the other ranks (Julia: Distributed/MPI.bcast, a file, a
 *                         socket) -> kabc_comm_init_rank everywhere (ncclCommInitRank)
 *   one process, n GPUs   kabc_comm_init_all: n contexts + n communicators at once
 *                         (ncclCommInitAll), driven with the *_multi entry points
 *                         (ncclGroupStart/End around the n all-gathers).
 * KABC_COMM_P2P (kabc_comm_init_all only) replaces RCCL by a pull kernel: every GPU
 * reads the peers' fresh rows straight over its xGMI links (peer-mapped pointers, all
 * links busy at once, one launch per half-generation).  It accepts repeated device ids,
 * which is how the test-suite runs 8 ranks on a 1-GPU box. */
typedef struct kabc_comm kabc_comm_t;
#define KABC_COMM_ID_BYTES 128
#define KABC_COMM_MAX_WORLD 16
#define KABC_MAX_EXCHANGE_CHUNKS 16
typedef enum kabc_comm_backend {
    KABC_COMM_RCCL = 1,
    KABC_COMM_P2P = 2
} kabc_comm_backend_t;

kabc_status_t kabc_comm_unique_id(uint8_t id[KABC_COMM_ID_BYTES]);
/* collective over all ranks; ctx selects the GPU and the stream the collectives run on */
kabc_status_t kabc_comm_init_rank(kabc_ctx_t* ctx, const uint8_t id[KABC_COMM_ID_BYTES],
                                  int32_t rank, int32_t world, kabc_comm_t** out);
/* single process: ctxs[ndev] and comms[ndev] receive one context (private stream) and
 * one communicator per entry of dev_ids */
kabc_status_t kabc_comm_init_all(int32_t ndev, const int32_t* dev_ids, int32_t backend,
                                 kabc_ctx_t** ctxs, kabc_comm_t** comms);
int32_t kabc_comm_rank(const kabc_comm_t* c);
int32_t kabc_comm_world(const kabc_comm_t* c);
kabc_ctx_t* kabc_comm_ctx(const kabc_comm_t* c);
/* host-value reductions over the ranks (blocking; RCCL communicators of the
 * one-process-per-GPU kind): used for counters, wall-clock maxima and as a barrier */
kabc_status_t kabc_comm_allreduce_sum_u64(kabc_comm_t* c, uint64_t* inout, int32_t n);
kabc_status_t kabc_comm_allreduce_max_f64(kabc_comm_t* c, double* inout, int32_t n);
kabc_status_t kabc_comm_barrier(kabc_comm_t* c);
/* destroys the communicator (and the context when kabc_comm_init_all created it) */
kabc_status_t kabc_comm_destroy(kabc_comm_t* c);

/* AIS(nparticles) sharded over the communicator's ranks; the library owns the (padded)
 * global half buffers.  Any nparticles >= length(model)+5 is accepted (shards may be
 * uneven or empty).  kabc_ais_init then also gathers both halves, and kabc_ais_advance
 * issues the all-gather after every half-generation (out_samples must be NULL: the
 * trace of a sharded ensemble is read with kabc_ais_get_ensemble). */
kabc_status_t kabc_ais_create_dist(kabc_comm_t* comm, const kabc_model_t* model,
                                   int64_t nparticles, uint64_t seed, kabc_ais_t** out);
/* the whole ensemble as this rank sees it after the last all-gather: x[N][D], walker-id
 * order, push_p NOT applied (identical on every rank) */
kabc_status_t kabc_ais_get_ensemble(kabc_ais_t* h, double* x);
/* The row ranges of `half` this handle owns, in the order kabc_ais_get_state / set_state /
 * get_debug lay the owned rows out: first[i] = global row inside the half, count[i] rows
 * (may be 0).  Returns the number of segments (= exchange chunks K; 1 for unsharded
 * handles), at most `cap` of them are written; -1 on a bad argument. */
int32_t kabc_ais_owned_segments(const kabc_ais_t* h, int32_t half, int64_t* first, int64_t* count,
                                int32_t cap);
/* single-process drivers for the n handles created on the n communicators of one
 * kabc_comm_init_all call (hs[i] on comms[i], every rank present exactly once) */
kabc_status_t kabc_ais_init_multi(kabc_ais_t** hs, int32_t n, int32_t retry_sampling);
kabc_status_t kabc_ais_advance_multi(kabc_ais_t** hs, int32_t n, int64_t ngenerations,
                                     int32_t ntransitions, kabc_stats_t* stats);

/* ---- smc(prior, cost; kwargs...) -- src/smc.jl:92-206 -------------------- */
typedef struct kabc_smc_opts {
    int64_t nparticles;  /* 100   */
    double alpha;        /* 0.95  */
    int32_t mcmc_retrys; /* 0     */
    int32_t verbose;     /* false */
    double mcmc_tol;     /* 0.015 */
    double epstol;       /* 0.0   */
    double r_epstol;     /* (1-alpha)^1.5/50 ; pass NaN for this default */
    double min_r_ess;    /* alpha^2          ; pass NaN for this default */
    double max_stretch;  /* 2.0   */
    uint64_t seed;
    int64_t max_iterations; /* safety bound on the outer loop; 0 = 100000 */
} kabc_smc_opts_t;

typedef struct kabc_smc_iter {
    double eps;         /* ϵ of this iteration (src/smc.jl:134)              */
    int64_t ess;        /* sum(alive) before resampling (src/smc.jl:142)     */
    int64_t accepted;   /* accepted[] at the end of the MCMC step            */
    int32_t resampled;  /* 1 if step 2 ran (src/smc.jl:145)                  */
    int32_t flag;       /* `flag` of src/smc.jl:135-141                      */
    int32_t mcmc_passes; /* r at loop exit                                   */
    int32_t reserved;
} kabc_smc_iter_t;

typedef struct kabc_smc_result {
    double* theta;  /* host [N][D], push_p'ed positions of ALL particles        */
    double* cost;   /* host [N]      = field C of the reference's return value  */
    uint8_t* alive; /* host [N]      ; P = theta[alive]                         */
    double eps;     /* field ϵ                                                  */
    int64_t iterations;
    int64_t n_alive;
    uint64_t cost_evals;
    uint64_t proposals;
    kabc_smc_iter_t* iter_log; /* optional host [iter_log_cap] */
    int64_t iter_log_cap;
    double kernel_ms_mcmc; /* avg device ms of the propose+accept kernel */
    int64_t mcmc_launches;
} kabc_smc_result_t;

void kabc_smc_default_opts(kabc_smc_opts_t* o);
kabc_status_t kabc_smc_run(kabc_ctx_t* ctx, const kabc_prior_t* prior, int32_t D,
                           const kabc_cost_t* cost, const kabc_smc_opts_t* opts,
                           kabc_smc_result_t* result);

/* smc with its COST LOOP sharded over the communicator's ranks -- the reference's own parallel
 * leg (`parallel = true`: Threads.@threads over the cost evaluations, src/smc.jl:120-123,168).
 * Every rank holds the whole ensemble and runs the epsilon-selection redundantly (identical
 * inputs, identical results); the propose / prior-MH / cost / accept pass is split by blocks
 * of 64 particles and ends with one grouped in-place all-gather of the rows it produced.
 * Worth it for an EXPENSIVE simulator only (from ~10 us per evaluation: a C4-sized pass
 * gathers 4.6 MB); a cheap cost is faster on one GPU (kabc_smc_run's persistent loop kernel).
 * Collective: every rank calls it with the same arguments and receives the same result; a rank
 * that fails before a pass's all-gather leaves the others waiting in it (RCCL) -- treat any
 * non-OK status as fatal for the whole job.
 * Draws are keyed by particle: the result equals kabc_smc_run's bit for bit.
 * Communicators: kabc_comm_init_rank (one process per GPU, RCCL); the P2P communicators of
 * kabc_comm_init_all when each rank is driven by its own host thread (how the test-suite
 * runs several ranks on one GPU).  length(prior) <= KABC_MAX_DIM_DYN: beyond KABC_MAX_DIM every rank draws and
 * costs the whole initial ensemble itself (counter-based draws: the same on every rank), the passes are shared out. */
kabc_status_t kabc_smc_run_dist(kabc_comm_t* comm, const kabc_prior_t* prior, int32_t D,
                                const kabc_cost_t* cost, const kabc_smc_opts_t* opts,
                                kabc_smc_result_t* result);

/* The same with the choice of what the ranks share out:
 *   KABC_SMC_DIST_COST_LOOP  (kabc_smc_run_dist's default) the propose / accept pass only; every rank
 *                            repeats the epsilon-selection (src/smc.jl:134-153) on the gathered costs.
 *   KABC_SMC_DIST_PARTICLES  SURVEY §8e "SMC": rank r OWNS the particles of its blocks -- their alive
 *                            mask lives on that rank only and the selection runs over the rank's own
 *                            costs.  ONE all-gather per selection: every rank ships, unasked, its alive
 *                            keys of a window predicted from the last two values of epsilon (a few
 *                            percent of its particles), their 1024-bin histogram, the count of its keys
 *                            below the window and its smallest key above; when the target rank falls
 *                            inside the window -- the usual course -- every rank derives epsilon, the
 *                            ESS and the resample decision from that, and a resample's index
 *                            (idx = repeat(idxalive, ...), :146-147) from the gathered costs it holds
 *                            anyway.  Otherwise (the first two selections, a decrement far off the
 *                            last, epsilon == 0) the selection is repeated phase by phase: all-gathered
 *                            histograms, candidate keys, counts, and on a resample the ranks' compacted
 *                            indices.  Partners are read from the gathered ensemble, as in the other
 *                            mode.  For large ensembles: the redundant selection is the serial fraction
 *                            of the other mode.
 * With mcmc_retrys = 0 (the reference's default) both modes enqueue batches of up to eight iterations
 * -- kernels and collectives: the pass's grouped all-gather, and the selection's one in the second mode
 * -- between two looks at the control block; with retry passes allowed the host looks after every pass.
 * KABC_SMC_DIST_LOOKS=1 forces that course.  kabc_smc_dist_stats tells what the last run did.
 * Both modes return kabc_smc_run's result bit for bit, on every rank.  kabc_smc_run_dist reads
 * KABC_SMC_DIST=particles|cost_loop (every rank must see the same value). */
#define KABC_SMC_DIST_COST_LOOP 0
#define KABC_SMC_DIST_PARTICLES 1
kabc_status_t kabc_smc_run_dist_mode(kabc_comm_t* comm, const kabc_prior_t* prior, int32_t D,
                                     const kabc_cost_t* cost, const kabc_smc_opts_t* opts, int32_t mode,
                                     kabc_smc_result_t* result);

/* How the calling thread's last kabc_smc_run / kabc_smc_run_dist[_mode] was driven: out[0] epsilon-
 * iterations, [1] collectives issued (grouped all-gathers; those of batches enqueued past the end of the
 * loop or behind a stalled selection included), [2] host looks at the control block (the batched course
 * and the sharded courses count them), [3] selections decided by the one exchange, [4] selections made
 * phase by phase / by the select kernel inside the batched course (the first two, and every stalled one),
 * [5] propose / accept passes, [6] 1 when batches of iterations with the one-exchange selection were
 * enqueued between looks -- sharded runs with mcmc_retrys = 0, and single-GPU runs of 2^20 particles and
 * more (the select kernel is faster below: KABC_SMC_SPEC_SELECT=1 / 0 forces / forbids the course on a
 * single GPU), [7] the collectives of ONE iteration in the batched course's usual case (2 with sharded
 * particles: the selection's payload and the pass's grouped all-gather; 1 with a sharded cost loop; 0 on
 * a single GPU), -1 when the run was not batched. */
void kabc_smc_dist_stats(int64_t out[8]);

/* ---- ABCDE(prior, cost, ϵ_target; kwargs...) -- src/smc.jl:347-430 -----------
 * ABC differential evolution (exported, undocumented and untested in the reference:
 * parity is oracle-vs-device only).  Generation-synchronous and double-buffered in
 * the reference already (nθs/nΔs/nlogπ, :374-376,413-422), so it maps 1:1. */
typedef struct kabc_abcde_opts {
    int64_t nparticles;     /* 50   */
    int64_t generations;    /* 20   */
    double eps_target;      /* ϵ_target (positional in the reference) */
    double alpha;           /* α = 0, must satisfy 0 <= α < 1 (:348) */
    double proposal_width;  /* 1.0  */
    int32_t earlystop;      /* false */
    int32_t verbose;
    uint64_t seed;
} kabc_abcde_opts_t;

typedef struct kabc_abcde_result {
    double* theta;            /* host [N][D], push_p'ed (:425)         */
    double* cost;             /* host [N] = Δs (field C)               */
    int32_t reached_eps;      /* maximum(Δs) <= ϵ_target (:422)        */
    int32_t reserved;
    int64_t generations_run;  /* iters                                  */
    uint64_t nsims;           /* sum(nsims) (:407)                      */
} kabc_abcde_result_t;

void kabc_abcde_default_opts(kabc_abcde_opts_t* o);
kabc_status_t kabc_abcde_run(kabc_ctx_t* ctx, const kabc_prior_t* prior, int32_t D,
                             const kabc_cost_t* cost, const kabc_abcde_opts_t* opts,
                             kabc_abcde_result_t* result);

/* ---- pfilter(prior, cost, N; kwargs...) -- src/smc.jl:275-340 -----------------
 * Rejection-refresh particle filter (exported, undocumented and untested in the
 * reference: parity is oracle-vs-device only).  Every iteration the particles above
 * the q-quantile of the costs are re-proposed from three distinct survivors until
 * they pass the prior-MH test and land below ϵ.  Up to 256 particles with a built-in cost the whole loop
 * is ONE launch of one workgroup (the reference's default N is 100); beyond, a selection launch + one launch
 * in which every particle runs its rejection loop to the end, per iteration.  Same result either way. */
typedef struct kabc_pfilter_opts {
    int64_t nparticles;     /* N (positional in the reference); raised to ceil((4D+1)/q) if N*q <= 4D */
    double q;               /* 0.7   */
    double eff_tol;         /* 0.1   */
    double epstol;          /* -Inf  */
    double proposal_width;  /* 0.75  */
    int64_t max_iters;      /* Inf -> pass -1 (any negative); 0 stops after the first iteration */
    int32_t verbose;
    int32_t reserved;
    uint64_t seed;
} kabc_pfilter_opts_t;

typedef struct kabc_pfilter_result {
    double* theta;      /* host [N_eff][D], push_p'ed (:334); N_eff = kabc_pfilter_nparticles() */
    double* cost;       /* host [N_eff] (field C)                                               */
    double eps;         /* ϵ of the last iteration                                              */
    double eff;         /* eff of the last iteration (:327)                                     */
    int64_t iterations;
    uint64_t nreps;     /* total proposals                                                      */
    uint64_t cost_evals;
} kabc_pfilter_result_t;

void kabc_pfilter_default_opts(kabc_pfilter_opts_t* o);
/* the particle count the reference actually uses (src/smc.jl:276-279) */
int64_t kabc_pfilter_nparticles(int64_t N, double q, int32_t D);
kabc_status_t kabc_pfilter_run(kabc_ctx_t* ctx, const kabc_prior_t* prior, int32_t D,
                               const kabc_cost_t* cost, const kabc_pfilter_opts_t* opts,
                               kabc_pfilter_result_t* result);

#ifdef __cplusplus
}
#endif
#endif /* KABC_H */
