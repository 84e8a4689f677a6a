/*
 * kabc_oracle.h -- CPU ORACLE for the KissABC walker-update path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under kissabc.jl_amd/ (the product) may
 * include, link, import or execute anything under oracle/.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and only as
 * the checker / reported baseline.
 *
 * What it is: a plain-C, serial restatement of the reference's algorithm for the
 * path (KissABC.jl v3.0.1: src/transition.jl, src/types.jl, src/priors.jl,
 * src/KissABC.jl:35-80, src/smc.jl:92-206); every function cites the reference
 * lines it follows.
 *
 * PARITY STATUS (see DESIGN.md "Oracle"): the reference is Julia and Julia is
 * not available in the build image, and its RNG streams / Distributions.jl /
 * Statistics.quantile are un-vendored dependencies.  The oracle is therefore
 * pinned by (1) every exact-value test the reference holds for the path
 * (test/runtests.jl:8-31 Factored pdf/logpdf/push_p), (2) scipy.stats golden
 * vectors for each prior family (tests/golden/), (3) the Random123 KAT vectors
 * for Philox, (4) glibc/mpmath for the math contract, (5) closed forms from the
 * source (cdf_g_inv end points, error messages, resample index pattern) and
 * (6) the reference's statistical known answers (test/runtests.jl, README).
 * Trajectory-level parity with Julia's own RNG streams is "parity unpinned".
 *
 * Two AIS schedules:
 *   serial  : the reference's sweep, one walker per step(), partners from all
 *             j != i at their current positions (src/KissABC.jl:66-80).
 *   sync    : the generation-synchronous red/black schedule of the GPU path
 *             (include/kabc.h "Schedule"); bit-exact target of the HIP kernels.
 */
#ifndef KABC_ORACLE_H
#define KABC_ORACLE_H

#include "kabc.h"

typedef struct orc_ais orc_ais_t;

/* per-transition record for index-exact parity checks */
typedef struct orc_trace_rec {
    int32_t move;     /* 1 stretch, 2 de, 3 walk (src/transition.jl:62) */
    int32_t accepted; /* 0/1 */
    int32_t a, b, c;  /* partner walker ids (-1 if unused) */
    int32_t cost_evaluated;
} orc_trace_rec_t;

const char* orc_last_error(void);

/* independent Philox4x32-10 (not the one in include/kabc_philox.h) */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
void orc_philox4x32_r(const uint32_t ctr[4], const uint32_t key[2], int32_t rounds, uint32_t out[4]);
int32_t orc_philox_rounds(void);

/* math-contract probes (vectorised wrappers over include/kabc_math.h) */
void orc_math_vec(int32_t fn, int64_t n, const double* x, double* out);
void orc_normal_pairs(int64_t n, const uint64_t* r, double* out);
void orc_div_rc_vec(int64_t n, const double* x, const double* c, double* out);

/* Factored surface: src/priors.jl:18-49, src/types.jl:27-32 */
int32_t orc_factored_logpdf(const kabc_prior_t* prior, int32_t D, int64_t n, const double* x,
                            double* out);
int32_t orc_factored_pdf(const kabc_prior_t* prior, int32_t D, int64_t n, const double* x,
                         double* out);
int32_t orc_push_p(const kabc_prior_t* prior, int32_t D, int64_t n, const double* x, double* out);
int32_t orc_factored_rand(const kabc_prior_t* prior, int32_t D, uint64_t seed, uint32_t domain,
                          int64_t first_walker, int64_t n, uint64_t attempt, double* out);
double orc_cost_eval(const kabc_cost_t* cost, int32_t D, const double* x, uint64_t seed,
                     uint32_t walker, uint64_t t, uint32_t domain);
double orc_cdf_g_inv(double u, double a);
/* register the gcc-compiled twin of a user DeviceCost plugin under its id */
int32_t orc_register_user_cost(int32_t id, void* fn);
/* a user prior family (kind >= KABC_PRIOR_USER): logpdf(x, p, tab), rand(p, window) of its snippet */
int32_t orc_register_user_prior(int32_t kind, void* logpdf, void* rnd, int32_t discrete);
int32_t orc_register_user_mvprior(int32_t kind, void* logpdf, void* rnd); /* a JOINT prior: kabc_compile_mvprior_plugin */
int32_t orc_register_user_init(int32_t id, void* fn);
/* MvNormal(mu, Sigma) priors (include/kabc_mvnormal.h): the oracle's registry; the components it is
 * handed are resolved by the caller: p[1] = k, p[2] = bits of orc_mvnormal_block(handle), p[3] = D */
int32_t orc_mvnormal_register(const double* mu, const double* cov, int32_t D, int32_t* handle);
uint64_t orc_mvnormal_block(int32_t handle);

/* AIS */
int32_t orc_ais_create(const kabc_model_t* model, int64_t nparticles, uint64_t seed,
                       orc_ais_t** out);
int32_t orc_ais_init(orc_ais_t* h, int32_t retry_sampling);
/* nsteps reference step() calls (serial schedule); out: [nsteps][D] or NULL */
int32_t orc_ais_steps_serial(orc_ais_t* h, int64_t nsteps, int32_t ntransitions, double* out);
/* ngenerations of the sync schedule; out: [ngen][N][D] or NULL;
 * trace: [ngen][N][ntransitions] records or NULL */
int32_t orc_ais_generations_sync(orc_ais_t* h, int64_t ngenerations, int32_t ntransitions,
                                 double* out, orc_trace_rec_t* trace);
/* one half of one generation restricted to rows [row_begin,row_end) of that half
 * (the unit a rank executes in the sharded driver); does not advance t */
int32_t orc_ais_half_generation(orc_ais_t* h, int32_t half, int32_t ntransitions,
                                int64_t row_begin, int64_t row_end);
int32_t orc_ais_end_generation(orc_ais_t* h, int32_t ntransitions);
int32_t orc_ais_get_state(orc_ais_t* h, double* x, double* logprior, double* loglik, uint64_t* t);
int32_t orc_ais_set_state(orc_ais_t* h, const double* x, const double* logprior,
                          const double* loglik, uint64_t t);
int32_t orc_ais_get_stats(orc_ais_t* h, kabc_stats_t* st);
void orc_ais_destroy(orc_ais_t* h);

/* smc(prior, cost; ...) -- src/smc.jl:92-206 */
int32_t orc_smc_run(const kabc_prior_t* prior, int32_t D, const kabc_cost_t* cost,
                    const kabc_smc_opts_t* opts, kabc_smc_result_t* result);
/* ABCDE(prior, cost, ϵ_target; ...) -- src/smc.jl:347-430 */
int32_t orc_abcde_run(const kabc_prior_t* prior, int32_t D, const kabc_cost_t* cost,
                      const kabc_abcde_opts_t* opts, kabc_abcde_result_t* result);
/* pfilter(prior, cost, N; ...) -- src/smc.jl:275-340 */
int64_t orc_pfilter_nparticles(int64_t N, double q, int32_t D);
int32_t orc_pfilter_run(const kabc_prior_t* prior, int32_t D, const kabc_cost_t* cost,
                        const kabc_pfilter_opts_t* opts, kabc_pfilter_result_t* result);
/* Statistics.quantile(v, p) (type 7), restated; v is not modified */
int32_t orc_quantile(const double* v, int64_t n, double p, double* out);

#endif
