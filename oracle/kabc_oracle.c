/*
 * kabc_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, see kabc_oracle.h).
 * Plain C99, serial.  Reference citations are relative to KissABC.jl v3.0.1.
 */
#include "kabc_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "kabc_costs.h"
#include "kabc_math.h"
#include "kabc_philox.h"
#include "kabc_sampling.h"

/* the oracle restates the reference for any length(prior) the product accepts
 * (KABC_MAX_DIM_DYN, include/kabc.h) */
#define ORC_MAX_DIM KABC_MAX_DIM_DYN

static __thread char g_err[512];
const char* orc_last_error(void) { return g_err; }
static int32_t fail(int32_t code, const char* msg) {
    snprintf(g_err, sizeof g_err, "%s", msg);
    return code;
}

/* ------------------------------------------------------------------------- */
/* independent Philox4x32-10 restatement (Salmon et al. SC'11, Fig. 2)        */
/* ------------------------------------------------------------------------- */
void orc_philox4x32_r(const uint32_t ctr[4], const uint32_t key[2], int32_t rounds, uint32_t out[4]) {
    uint32_t x0 = ctr[0], x1 = ctr[1], x2 = ctr[2], x3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int round = 0; round < rounds; ++round) {
        if (round > 0) {
            k0 += 0x9E3779B9u;
            k1 += 0xBB67AE85u;
        }
        uint64_t prod0 = (uint64_t)0xD2511F53u * x0;
        uint64_t prod1 = (uint64_t)0xCD9E8D57u * x2;
        uint32_t y0 = (uint32_t)(prod1 >> 32) ^ x1 ^ k0;
        uint32_t y1 = (uint32_t)(prod1 & 0xffffffffu);
        uint32_t y2 = (uint32_t)(prod0 >> 32) ^ x3 ^ k1;
        uint32_t y3 = (uint32_t)(prod0 & 0xffffffffu);
        x0 = y0;
        x1 = y1;
        x2 = y2;
        x3 = y3;
    }
    out[0] = x0;
    out[1] = x1;
    out[2] = x2;
    out[3] = x3;
}

void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    orc_philox4x32_r(ctr, key, 10, out);
}
/* the round count the stream contract was compiled with (include/kabc_philox.h) */
int32_t orc_philox_rounds(void) { return KABC_PHILOX_ROUNDS; }

typedef struct blk {
    uint64_t lo, hi;
    uint32_t w[4];
} blk_t;

/* stream block (seed, walker, t, slot, domain): counter layout of kabc_philox.h */
static blk_t stream(uint64_t seed, uint32_t walker, uint64_t t, uint32_t slot, uint32_t domain) {
    uint32_t ctr[4] = {walker, (uint32_t)t, slot, domain | ((uint32_t)(t >> 32) << 8)};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    blk_t b;
    orc_philox4x32_r(ctr, key, KABC_PHILOX_ROUNDS, b.w);
    b.lo = ((uint64_t)b.w[1] << 32) | b.w[0];
    b.hi = ((uint64_t)b.w[3] << 32) | b.w[2];
    return b;
}

/* ------------------------------------------------------------------------- */
/* math probes                                                                */
/* ------------------------------------------------------------------------- */
void orc_math_vec(int32_t fn, int64_t n, const double* x, double* out) {
    for (int64_t i = 0; i < n; ++i) {
        switch (fn) {
            case 0: out[i] = kabc_log(x[i]); break;
            case 1: out[i] = kabc_exp(x[i]); break;
            case 2: out[i] = kabc_log1p(x[i]); break;
            case 3: out[i] = kabc_lgamma(x[i]); break;
            case 4: {
                double s, c;
                kabc_sincos2pi(x[i], &s, &c);
                out[2 * i] = s;
                out[2 * i + 1] = c;
            } break;
            case 5: out[i] = kabc_sqrt(x[i]); break;
            case 6: out[i] = kabc_rint(x[i]); break;
            case 7: out[i] = kabc_log_pn(x[i]); break;
            case 8: out[i] = kabc_sqrt_pn(x[i]); break;
            case 9: out[i] = kabc_u01(kabc_bits(x[i])); break;
            case 10:
                kabc_normal_pair(kabc_bits(x[2 * i]), kabc_bits(x[2 * i + 1]), &out[2 * i],
                                 &out[2 * i + 1]);
                break;
            case 11:
                out[i] = (double)kabc_index32(kabc_bits(x[2 * i]), (uint32_t)x[2 * i + 1]);
                break;
            case 12: out[i] = kabc_exp_bounded(x[i]); break;
            default: out[i] = KABC_NAN;
        }
    }
}
void orc_div_rc_vec(int64_t n, const double* x, const double* c, double* out) {
    for (int64_t i = 0; i < n; ++i) out[i] = kabc_div_rc(x[i], c[i], 1.0 / c[i]);
}
void orc_normal_pairs(int64_t n, const uint64_t* r, double* out) {
    for (int64_t i = 0; i < n; ++i)
        kabc_normal_pair(r[2 * i], r[2 * i + 1], &out[2 * i], &out[2 * i + 1]);
}

/* ------------------------------------------------------------------------- */
/* Factored prior: logpdf / pdf / push_p  (src/priors.jl, src/types.jl:27-32) */
/* ------------------------------------------------------------------------- */
typedef struct prep {
    int32_t kind;
    int32_t discrete;
    double p[4];
    double c0, c1;
    double rb; /* RN(1 / p[1]) (or 1/p[0] for Exponential) for kabc_div_rc */
} prep_t;

static double std_normal_cdf(double z) { return 0.5 * erfc(-z * M_SQRT1_2); }

/* user prior families (kinds >= KABC_PRIOR_USER; the reference's Factored takes any
 * UnivariateDistribution, src/priors.jl:11): the same C snippet the device path compiles with
 * hipRTC (kabc_compile_prior_plugin), compiled with gcc by oracle.py and registered here under
 * the same kind */
typedef double (*orc_user_prior_logpdf_fn)(double, const double*, const double*);
typedef double (*orc_user_prior_rand_fn)(const double*, const kabc_slotwin_t*);
#define ORC_MAX_USER_PRIORS 256
static struct {
    orc_user_prior_logpdf_fn logpdf;
    orc_user_prior_rand_fn rand;
    int discrete;
} g_user_prior[ORC_MAX_USER_PRIORS];
int32_t orc_register_user_prior(int32_t kind, void* logpdf, void* rnd, int32_t discrete) {
    if (kind < KABC_PRIOR_USER || kind >= KABC_PRIOR_USER + ORC_MAX_USER_PRIORS || !logpdf || !rnd)
        return fail(KABC_ERR_INVALID_ARG, "bad user prior kind");
    g_user_prior[kind - KABC_PRIOR_USER].logpdf = (orc_user_prior_logpdf_fn)logpdf;
    g_user_prior[kind - KABC_PRIOR_USER].rand = (orc_user_prior_rand_fn)rnd;
    g_user_prior[kind - KABC_PRIOR_USER].discrete = discrete != 0;
    return KABC_OK;
}
/* joint user priors (kabc_compile_mvprior_plugin, include/kabc.h): one density / one draw of the whole
 * vector; all D components of the prior carry the kind (any Distribution as the prior: src/types.jl:30,
 * 34-35,52; src/smc.jl:92-93) */
typedef double (*orc_user_mvprior_logpdf_fn)(const double*, int, const double*, int, const double*);
typedef void (*orc_user_mvprior_rand_fn)(double*, int, const double*, int, const kabc_slotwin_t*);
static struct {
    orc_user_mvprior_logpdf_fn logpdf;
    orc_user_mvprior_rand_fn rand;
} g_user_mvprior[ORC_MAX_USER_PRIORS];
static double orc_joint_component_nan(double x, const double* p, const double* tab) {
    (void)x;
    (void)p;
    (void)tab;
    return KABC_NAN;  /* (the device's per-component value for such a kind: replaced by the joint density) */
}
static double orc_joint_component_rand_nan(const double* p, const kabc_slotwin_t* w) {
    (void)p;
    (void)w;
    return KABC_NAN;
}
int32_t orc_register_user_mvprior(int32_t kind, void* logpdf, void* rnd) {
    if (kind < KABC_PRIOR_USER || kind >= KABC_PRIOR_USER + ORC_MAX_USER_PRIORS || !logpdf || !rnd)
        return fail(KABC_ERR_INVALID_ARG, "orc_register_user_mvprior: bad kind / NULL function");
    g_user_mvprior[kind - KABC_PRIOR_USER].logpdf = (orc_user_mvprior_logpdf_fn)logpdf;
    g_user_mvprior[kind - KABC_PRIOR_USER].rand = (orc_user_mvprior_rand_fn)rnd;
    /* the univariate table knows the kind too (continuous; its component value is never used) */
    g_user_prior[kind - KABC_PRIOR_USER].logpdf = orc_joint_component_nan;
    g_user_prior[kind - KABC_PRIOR_USER].rand = orc_joint_component_rand_nan;
    g_user_prior[kind - KABC_PRIOR_USER].discrete = 0;
    return KABC_OK;
}
static int user_prior_joint(int kind) {
    return kind >= KABC_PRIOR_USER && kind < KABC_PRIOR_USER + ORC_MAX_USER_PRIORS &&
           g_user_mvprior[kind - KABC_PRIOR_USER].logpdf != 0;
}

static int user_prior_known(int kind) {
    return kind >= KABC_PRIOR_USER && kind < KABC_PRIOR_USER + ORC_MAX_USER_PRIORS &&
           g_user_prior[kind - KABC_PRIOR_USER].logpdf != 0;
}

/* derived constants of one component; host libm is used for the one-off
 * normalisers (lgamma, erfc), the math contract for everything else */
static int prepare_prior(const kabc_prior_t* pr, prep_t* q) {
    q->kind = pr->kind;
    q->discrete = kabc_prior_is_discrete(pr->kind);
    memcpy(q->p, pr->p, sizeof q->p);
    q->c0 = q->c1 = 0.0;
    const double a = pr->p[0], b = pr->p[1];
    q->rb = 1.0 / ((pr->kind == KABC_PRIOR_EXPONENTIAL) ? a : b);
    switch (pr->kind) {
        case KABC_PRIOR_USER_INIT: /* no density: CommonLogDensity's own sample_init (src/types.jl:112) */
            q->rb = 0.0;
            return 1;
        case KABC_PRIOR_MVNORMAL: /* resolved by oracle.py: p[1] = k, p[2] = the block, p[3] = D */
            q->rb = 0.0;
            return kabc_bits(pr->p[2]) != 0 && pr->p[3] >= 1.0;
        case KABC_PRIOR_UNIFORM:
            if (!(b > a)) return 0;
            q->c0 = -kabc_log(b - a);
            return 1;
        case KABC_PRIOR_NORMAL:
        case KABC_PRIOR_LOGNORMAL:
            if (!(b > 0)) return 0;
            q->c0 = kabc_log(b);
            return 1;
        case KABC_PRIOR_TRUNCNORMAL: {
            if (!(b > 0) || !(pr->p[3] > pr->p[2])) return 0;
            q->c0 = kabc_log(b);
            double zl = (pr->p[2] - a) / b, zh = (pr->p[3] - a) / b;
            double tp = (zl > 0) ? std_normal_cdf(-zl) - std_normal_cdf(-zh)
                                 : std_normal_cdf(zh) - std_normal_cdf(zl);
            q->c1 = log(tp);
            return 1;
        }
        case KABC_PRIOR_BETA:
            if (!(a > 0) || !(b > 0)) return 0;
            q->c0 = lgamma(a) + lgamma(b) - lgamma(a + b);
            return 1;
        case KABC_PRIOR_DISCRETE_UNIFORM:
            if (!(b >= a) || a != kabc_rint(a) || b != kabc_rint(b)) return 0;
            q->c0 = -kabc_log(b - a + 1.0);
            return 1;
        case KABC_PRIOR_NEGBINOMIAL:
            if (!(a > 0) || !(b > 0) || !(b <= 1)) return 0;
            q->c0 = a * kabc_log(b) - lgamma(a);
            q->c1 = kabc_log1p(-b);
            return 1;
        case KABC_PRIOR_EXPONENTIAL:
            if (!(a > 0)) return 0;
            q->c0 = kabc_log(a);
            return 1;
        case KABC_PRIOR_GAMMA:
            if (!(a > 0) || !(b > 0)) return 0;
            q->c0 = lgamma(a) + a * kabc_log(b);
            return 1;
        default:
            if (!user_prior_known(pr->kind)) return 0;
            q->discrete = g_user_prior[pr->kind - KABC_PRIOR_USER].discrete;
            q->rb = 0.0;
            return 1;
    }
}

/* Distributions.logpdf(p_k, x) for the supported families (textbook densities;
 * pinned by scipy.stats golden vectors, tests/golden/priors_logpdf.json) */
static double comp_logpdf(const prep_t* q, double x) {
    const double a = q->p[0], b = q->p[1];
    switch (q->kind) {
        case KABC_PRIOR_UNIFORM: return (x >= a && x <= b) ? q->c0 : -KABC_INF;
        case KABC_PRIOR_NORMAL: {
            double z = kabc_div_rc(x - a, b, q->rb);
            return -(z * z + KABC_LOG_2PI) / 2.0 - q->c0;
        }
        case KABC_PRIOR_TRUNCNORMAL: {
            if (!(x >= q->p[2] && x <= q->p[3])) return -KABC_INF;
            double z = kabc_div_rc(x - a, b, q->rb);
            return -(z * z + KABC_LOG_2PI) / 2.0 - q->c0 - q->c1;
        }
        case KABC_PRIOR_BETA: {
            if (!(x >= 0.0 && x <= 1.0)) return -KABC_INF;
            double t1 = (a == 1.0) ? 0.0 : (a - 1.0) * kabc_log(x);
            double t2 = (b == 1.0) ? 0.0 : (b - 1.0) * kabc_log1p(-x);
            return t1 + t2 - q->c0;
        }
        case KABC_PRIOR_DISCRETE_UNIFORM:
            return (x >= a && x <= b && x == kabc_rint(x)) ? q->c0 : -KABC_INF;
        case KABC_PRIOR_NEGBINOMIAL: {
            if (!(x >= 0.0) || x != kabc_rint(x)) return -KABC_INF;
            return q->c0 + x * q->c1 + kabc_lgamma(x + a) - kabc_lgamma(x + 1.0);
        }
        case KABC_PRIOR_EXPONENTIAL: return (x >= 0.0) ? -q->c0 - kabc_div_rc(x, a, q->rb) : -KABC_INF;
        case KABC_PRIOR_GAMMA: {
            if (!(x >= 0.0)) return -KABC_INF;
            double t1 = (a == 1.0) ? 0.0 : (a - 1.0) * kabc_log(x);
            return t1 - kabc_div_rc(x, b, q->rb) - q->c0;
        }
        case KABC_PRIOR_LOGNORMAL: {
            if (!(x > 0.0)) return -KABC_INF;
            double lx = kabc_log(x);
            double z = kabc_div_rc(lx - a, b, q->rb);
            return -(z * z + KABC_LOG_2PI) / 2.0 - q->c0 - lx;
        }
        default:
            if (user_prior_known(q->kind)) return g_user_prior[q->kind - KABC_PRIOR_USER].logpdf(x, q->p, kabc_log_tab);
            return KABC_NAN;
    }
}

/* push_p(density, p): continuous -> float(p), discrete -> round(Int, p)
 * (src/types.jl:29-32) */
static void push_p(const prep_t* q, int D, const double* x, double* out) {
    for (int k = 0; k < D; ++k) out[k] = q[k].discrete ? kabc_rint(x[k]) : x[k];
}

/* logpdf(d::Factored, x): s = logpdf(p[1],x[1]); for i=2:N s += ... (src/priors.jl:30-36) */
/* component k; an MvNormal component (include/kabc_mvnormal.h: Distributions.jl's
 * logpdf(MvNormal, x) split into D Normal-shaped terms) reads coordinates 0..k */
static double comp_logpdf_k(const prep_t* q, int D, int k, const double* x) {
    if (q[k].kind == KABC_PRIOR_MVNORMAL)
        return kabc_mvn_logpdf_comp(kabc_mvn_ptr_from_double(q[k].p[2]), D, k, x);
    return comp_logpdf(&q[k], x[k]);
}
static double factored_logpdf(const prep_t* q, int D, const double* x) {
    if (user_prior_joint(q[0].kind))  /* logpdf(prior, x) of a joint prior: one function of the vector */
        return g_user_mvprior[q[0].kind - KABC_PRIOR_USER].logpdf(x, D, q[0].p, (int)(sizeof(prep_t) / sizeof(double)),
                                                                  kabc_log_tab);
    double s = comp_logpdf_k(q, D, 0, x);
    for (int k = 1; k < D; ++k) s += comp_logpdf_k(q, D, k, x);
    return s;
}

/* MvNormal(mu, Sigma) priors: the oracle's own registry of prepared blocks (oracle.py resolves
 * the components it hands over: p[2] = orc_mvnormal_block(handle) as a double's bits, p[3] = D) */
static double* g_mvn_blk[256];
static int g_mvn_dim[256];
static int g_mvn_n;
int32_t orc_mvnormal_register(const double* mu, const double* cov, int32_t D, int32_t* handle) {
    if (!mu || !cov || !handle || D < 1 || D > KABC_MAX_DIM) return fail(KABC_ERR_INVALID_ARG, "bad MvNormal");
    if (g_mvn_n >= 256) return fail(KABC_ERR_UNSUPPORTED, "too many MvNormal priors");
    double* blk = (double*)malloc(sizeof(double) * (size_t)kabc_mvn_block_words(D));
    const int rc = kabc_mvn_prepare(D, mu, cov, blk);
    if (rc) {
        free(blk);
        return fail(KABC_ERR_INVALID_ARG, rc == 1 ? "Sigma is not symmetric" : "Sigma is not positive definite");
    }
    g_mvn_blk[g_mvn_n] = blk;
    g_mvn_dim[g_mvn_n] = D;
    *handle = ++g_mvn_n;
    return KABC_OK;
}
uint64_t orc_mvnormal_block(int32_t handle) {
    return (handle >= 1 && handle <= g_mvn_n) ? (uint64_t)(uintptr_t)g_mvn_blk[handle - 1] : 0;
}

static int prep_all(const kabc_prior_t* prior, int32_t D, prep_t* q) {
    if (D < 1 || D > ORC_MAX_DIM) return 0;
    for (int k = 0; k < D; ++k)
        if (!prepare_prior(&prior[k], &q[k])) return 0;
    return 1;
}

int32_t orc_factored_logpdf(const kabc_prior_t* prior, int32_t D, int64_t n, const double* x,
                            double* out) {
    prep_t q[ORC_MAX_DIM];
    if (!prep_all(prior, D, q)) return fail(KABC_ERR_INVALID_ARG, "invalid prior");
    for (int64_t i = 0; i < n; ++i) out[i] = factored_logpdf(q, D, x + i * D);
    return KABC_OK;
}

/* pdf(d::Factored, x): product of component pdfs (src/priors.jl:18-24) */
int32_t orc_factored_pdf(const kabc_prior_t* prior, int32_t D, int64_t n, const double* x,
                         double* out) {
    prep_t q[ORC_MAX_DIM];
    if (!prep_all(prior, D, q)) return fail(KABC_ERR_INVALID_ARG, "invalid prior");
    for (int64_t i = 0; i < n; ++i) {
        double s = exp(comp_logpdf_k(q, D, 0, x + i * D));
        for (int k = 1; k < D; ++k) s *= exp(comp_logpdf_k(q, D, k, x + i * D));
        out[i] = s;
    }
    return KABC_OK;
}

int32_t orc_push_p(const kabc_prior_t* prior, int32_t D, int64_t n, const double* x, double* out) {
    prep_t q[ORC_MAX_DIM];
    if (!prep_all(prior, D, q)) return fail(KABC_ERR_INVALID_ARG, "invalid prior");
    for (int64_t i = 0; i < n; ++i) push_p(q, D, x + i * D, out + i * D);
    return KABC_OK;
}

/* rand(rng, Factored) = ntuple(i -> rand(rng, p[i])) then op(float, .)
 * (src/priors.jl:42-43, src/KissABC.jl:50) */
static void factored_rand(const kabc_prior_t* prior, int D, uint64_t seed, uint32_t walker,
                          uint64_t attempt, uint32_t domain, double* out) {
    if (user_prior_joint(prior[0].kind)) {  /* rand(rng, prior) of a joint prior: the whole vector at once */
        kabc_slotwin_t w = {seed, attempt, walker, domain, 0u};
        g_user_mvprior[prior[0].kind - KABC_PRIOR_USER].rand(out, D, prior[0].p,
                                                               (int)(sizeof(kabc_prior_t) / sizeof(double)), &w);
        return;
    }
    for (int k = 0; k < D; ++k) {
        kabc_slotwin_t w = {seed, attempt, walker, domain, (uint32_t)k * KABC_SLOTS_PER_DIM};
        out[k] = user_prior_known(prior[k].kind) ? g_user_prior[prior[k].kind - KABC_PRIOR_USER].rand(prior[k].p, &w)
                                                 : kabc_sample_prior(&prior[k], &w);
    }
}

int32_t orc_factored_rand(const kabc_prior_t* prior, int32_t D, uint64_t seed, uint32_t domain,
                          int64_t first_walker, int64_t n, uint64_t attempt, double* out) {
    for (int64_t i = 0; i < n; ++i)
        factored_rand(prior, D, seed, (uint32_t)(first_walker + i), attempt, domain, out + i * D);
    return KABC_OK;
}

/* user DeviceCosts (ids >= KABC_COST_USER): the same C snippet the device plugin is
 * built from, compiled with gcc by oracle.py and registered here */
typedef double (*orc_user_cost_fn)(const double*, int, const double*, const double*, int64_t,
                                   kabc_cost_rng_t*);
static orc_user_cost_fn g_user_cost[64];
/* the snippet's own sample_init (KABC_USER_SAMPLE_INIT, include/kabc_costs.h): CommonLogDensity's
 * `sample_init(rng)` of src/types.jl:105-113 */
typedef void (*orc_user_init_fn)(double*, int, const double*, const double*, int64_t, kabc_cost_rng_t*);
static orc_user_init_fn g_user_init[64];
int32_t orc_register_user_init(int32_t id, void* fn) {
    if (id < KABC_COST_USER || id >= KABC_COST_USER + 64) return fail(KABC_ERR_INVALID_ARG, "bad user cost id");
    g_user_init[id - KABC_COST_USER] = (orc_user_init_fn)fn;
    return KABC_OK;
}
int32_t orc_register_user_cost(int32_t id, void* fn) {
    if (id < KABC_COST_USER || id >= KABC_COST_USER + 64) return fail(KABC_ERR_INVALID_ARG, "bad user cost id");
    g_user_cost[id - KABC_COST_USER] = (orc_user_cost_fn)fn;
    return KABC_OK;
}
static int cost_dim_ok_any(int id, int D) {
    if (id >= KABC_COST_USER) return id < KABC_COST_USER + 64 && g_user_cost[id - KABC_COST_USER] != 0;
    return kabc_cost_dim_ok(id, D);
}

double orc_cost_eval(const kabc_cost_t* cost, int32_t D, const double* x, uint64_t seed,
                     uint32_t walker, uint64_t t, uint32_t domain) {
    kabc_cost_rng_t rng = {seed, t, walker, domain, 0, 0, NULL, NULL};
    if (cost->id >= KABC_COST_USER)
        return g_user_cost[cost->id - KABC_COST_USER](x, D, cost->params, cost->data, cost->ndata,
                                                      &rng);
    return kabc_cost_eval(cost->id, x, D, cost->params, cost->data, cost->ndata, &rng);
}

/* cdf_g_inv(u, a) = (u*(sqrt(a) - sqrt(1/a)) + sqrt(1/a))^2  (src/transition.jl:46) */
double orc_cdf_g_inv(double u, double a) {
    double t = u * (kabc_sqrt(a) - kabc_sqrt(1.0 / a)) + kabc_sqrt(1.0 / a);
    return t * t;
}

/* ------------------------------------------------------------------------- */
/* AIS                                                                        */
/* ------------------------------------------------------------------------- */
struct orc_ais {
    int32_t D, posterior;
    double eps;
    kabc_prior_t prior[ORC_MAX_DIM];
    prep_t q[ORC_MAX_DIM];
    kabc_cost_t cost;
    double* cost_params;
    double* cost_data;
    int64_t N, N0; /* N0 = rows of half 0 */
    uint64_t seed;
    double* x;     /* [N][D] walker-id order (AISState.sample, src/KissABC.jl:27) */
    double* lp;    /* logprior */
    double* ll;    /* loglikelihood (kernelized) or cost (threshold), src/types.jl:57,90 */
    uint64_t* tc;  /* per-walker transition counter (serial schedule) */
    uint64_t t;    /* generation-synchronous counter */
    int64_t cursor; /* AISState.i (0-based), src/KissABC.jl:31 */
    kabc_stats_t st;
    int initialised;
};

typedef struct ld {
    double lp, ll;
} ld_t;

/* loglike(density, sample): src/types.jl:51-58 (kernelized), :84-91 (threshold) */
static ld_t loglike(orc_ais_t* h, const double* xp, uint32_t walker, uint64_t t, uint32_t dom,
                    int* cost_evaluated) {
    ld_t r;
    if (h->posterior == KABC_POSTERIOR_COMMON) { /* loglike = lπ(sample.x), src/types.jl:117-119 */
        r.lp = 0.0;
        r.ll = orc_cost_eval(&h->cost, h->D, xp, h->seed, walker, t, dom);
        *cost_evaluated = 1;
        return r;
    }
    r.lp = factored_logpdf(h->q, h->D, xp);
    *cost_evaluated = 0;
    if (h->posterior == KABC_POSTERIOR_KERNELIZED) {
        r.ll = r.lp;
        if (kabc_isfinite(r.lp)) {
            double c = orc_cost_eval(&h->cost, h->D, xp, h->seed, walker, t, dom);
            double q = kabc_div_rc(c, h->eps, 1.0 / h->eps);
            r.ll = -0.5 * (q * q); /* -0.5 * abs2(cost/scale) */
            *cost_evaluated = 1;
        }
    } else {
        r.ll = -r.lp;
        if (kabc_isfinite(r.lp)) {
            r.ll = orc_cost_eval(&h->cost, h->D, xp, h->seed, walker, t, dom);
            *cost_evaluated = 1;
        }
    }
    return r;
}

/* push_p(density, p): ABC posteriors project through the prior (src/types.jl:28),
 * a plain AbstractDensity (CommonLogDensity) is the identity (:27) */
static void model_push_p(const orc_ais_t* h, const double* x, double* out) {
    if (h->posterior == KABC_POSTERIOR_COMMON) memcpy(out, x, sizeof(double) * h->D);
    else push_p(h->q, h->D, x, out);
}

/* is_valid_logdensity: src/types.jl:60 (isfinite(sum(ld))), :93-94 */
static int is_valid(const orc_ais_t* h, ld_t v) {
    if (h->posterior != KABC_POSTERIOR_THRESHOLD) return kabc_isfinite(v.lp + v.ll); /* src/types.jl:60, :121 */
    return kabc_isfinite(v.ll) && kabc_isfinite(v.lp);
}

int32_t orc_ais_create(const kabc_model_t* m, int64_t N, uint64_t seed, orc_ais_t** out) {
    if (!m || !out || m->D < 1 || m->D > ORC_MAX_DIM)
        return fail(KABC_ERR_INVALID_ARG, "invalid model");
    if (N < m->D + 5) { /* src/KissABC.jl:43-48 */
        snprintf(g_err, sizeof g_err,
                 "nparticles = %lld is insufficient, set number of particles in AIS(⋅) atleast to %d",
                 (long long)N, m->D + 5);
        return KABC_ERR_INVALID_ARG;
    }
    if (!cost_dim_ok_any(m->cost.id, m->D))
        return fail(KABC_ERR_UNSUPPORTED, "cost id / dimension not supported");
    orc_ais_t* h = (orc_ais_t*)calloc(1, sizeof *h);
    h->D = m->D;
    h->posterior = m->posterior;
    h->eps = m->eps;
    memcpy(h->prior, m->prior, sizeof(kabc_prior_t) * m->D);
    if (!prep_all(h->prior, h->D, h->q)) {
        free(h);
        return fail(KABC_ERR_INVALID_ARG, "invalid prior parameters");
    }
    h->cost = m->cost;
    h->cost_params = (double*)malloc(sizeof(double) * (m->cost.nparams + 1));
    h->cost_data = (double*)malloc(sizeof(double) * (m->cost.ndata + 1));
    if (m->cost.nparams) memcpy(h->cost_params, m->cost.params, sizeof(double) * m->cost.nparams);
    if (m->cost.ndata) memcpy(h->cost_data, m->cost.data, sizeof(double) * m->cost.ndata);
    h->cost.params = h->cost_params;
    h->cost.data = h->cost_data;
    h->N = N;
    h->N0 = (N + 1) / 2;
    h->seed = seed;
    h->x = (double*)calloc(N * h->D, sizeof(double));
    h->lp = (double*)calloc(N, sizeof(double));
    h->ll = (double*)calloc(N, sizeof(double));
    h->tc = (uint64_t*)calloc(N, sizeof(uint64_t));
    *out = h;
    return KABC_OK;
}

void orc_ais_destroy(orc_ais_t* h) {
    if (!h) return;
    free(h->x);
    free(h->lp);
    free(h->ll);
    free(h->tc);
    free(h->cost_params);
    free(h->cost_data);
    free(h);
}

/* step(rng, model, spl::AIS; retry_sampling) -- src/KissABC.jl:35-64 */
/* unconditional_sample(rng, density): Particle(rand(rng, prior)) (src/types.jl:34), or the
 * CommonLogDensity's own sample_init(rng) (src/types.jl:112-113) when the snippet supplies it */
static void model_sample(orc_ais_t* h, uint32_t walker, uint64_t attempt, double* out) {
    if (h->prior[0].kind == KABC_PRIOR_USER_INIT) {
        kabc_cost_rng_t rng = {h->seed, attempt, walker, KABC_DOM_AIS_INIT, 0u};
        g_user_init[h->cost.id - KABC_COST_USER](out, h->D, h->cost.params, h->cost.data, h->cost.ndata, &rng);
        return;
    }
    factored_rand(h->prior, h->D, h->seed, walker, attempt, KABC_DOM_AIS_INIT, out);
}

int32_t orc_ais_init(orc_ais_t* h, int32_t retry_sampling) {
    const int D = h->D;
    if (h->prior[0].kind == KABC_PRIOR_USER_INIT &&
        (h->posterior != KABC_POSTERIOR_COMMON || h->cost.id < KABC_COST_USER ||
         !g_user_init[h->cost.id - KABC_COST_USER]))
        return fail(KABC_ERR_INVALID_ARG, "KABC_PRIOR_USER_INIT needs a CommonLogDensity whose snippet defines kabc_user_sample_init");
    int64_t retrys = (int64_t)retry_sampling * h->N; /* :52 */
    double xp[ORC_MAX_DIM];
    for (int64_t i = 0; i < h->N; ++i) { /* :50-51, attempt 0 */
        uint64_t attempt = 0;
        int ev;
        model_sample(h, (uint32_t)i, attempt, h->x + i * D);
        model_push_p(h, h->x + i * D, xp);
        ld_t v = loglike(h, xp, (uint32_t)i, attempt, KABC_DOM_AIS_INIT_COST, &ev);
        while (!is_valid(h, v)) { /* :54-60 */
            ++attempt;
            model_sample(h, (uint32_t)i, attempt, h->x + i * D);
            model_push_p(h, h->x + i * D, xp);
            v = loglike(h, xp, (uint32_t)i, attempt, KABC_DOM_AIS_INIT_COST, &ev);
            retrys -= 1;
            if (retrys < 0)
                return fail(KABC_ERR_RETRY_EXHAUSTED,
                            "Prior leads to ∞ costs too often, tune the prior or increase "
                            "`retry_sampling`.");
        }
        h->lp[i] = v.lp;
        h->ll[i] = v.ll;
    }
    memset(h->tc, 0, sizeof(uint64_t) * h->N);
    h->t = 0;
    h->cursor = 0;
    h->initialised = 1;
    return KABC_OK;
}

typedef struct partner_set {
    int64_t base; /* first eligible walker id */
    int64_t n;    /* number of eligible ids, contiguous from base ... */
    int64_t skip; /* ... except `skip` (the walker itself; -1 if not inside) */
} partner_set_t;

/* uniform draw from the eligible ids excluding up to three already chosen ones.
 * The reference rejects (`while a == i; a = rand(rng, eachindex(particles))`,
 * src/transition.jl:5-10,26-34,53-55); the distribution is uniform over the
 * remaining ids, drawn here without rejection from one 64-bit word. */
static int64_t draw_partner(uint64_t r, const partner_set_t* ps, const int64_t* excl, int nexcl) {
    int64_t ex[4];
    int ne = 0;
    if (ps->skip >= 0) ex[ne++] = ps->skip;
    for (int j = 0; j < nexcl; ++j) ex[ne++] = excl[j];
    /* sort ascending */
    for (int a = 1; a < ne; ++a)
        for (int b = a; b > 0 && ex[b - 1] > ex[b]; --b) {
            int64_t tmp = ex[b];
            ex[b] = ex[b - 1];
            ex[b - 1] = tmp;
        }
    int64_t v = ps->base + (int64_t)kabc_index(r, (uint64_t)(ps->n - ne));
    for (int j = 0; j < ne; ++j)
        if (v >= ex[j]) ++v;
    return v;
}

/* transition!(density, particles, logdensity, i, rng) -- src/transition.jl:67-82,
 * with propose (:61-65) and the three moves (:2-59) inlined.  `ps` is the set
 * partners are drawn from: all j != i (serial) or the complementary half (sync). */
static int transition(orc_ais_t* h, int64_t i, uint64_t t, const partner_set_t* ps,
                      orc_trace_rec_t* rec) {
    const int D = h->D;
    const uint32_t w = (uint32_t)i;
    const double* xi = h->x + i * D;
    double y[ORC_MAX_DIM], yp[ORC_MAX_DIM];
    double corr;

    blk_t B0 = stream(h->seed, w, t, 0, KABC_DOM_AIS_MOVE);
    blk_t B1 = stream(h->seed, w, t, 1, KABC_DOM_AIS_MOVE);
    /* p = rand(rng, (1,1,1,1,2,2,3))  (:62) */
    uint32_t m7 = (uint32_t)(((uint64_t)B0.w[2] * 7u) >> 32);
    int move = (m7 < 4) ? 1 : (m7 < 6) ? 2 : 3;
    int64_t a = draw_partner(B0.lo, ps, NULL, 0);
    int64_t b = -1, c = -1;
    const double* xa = h->x + a * D;

    if (move == 1) {
        /* stretch_propose :51-59, Z = sample_g(rng, 3.0) = cdf_g_inv(rand(rng), 3.0) */
        double u = kabc_u01(B1.hi);
        double Z = orc_cdf_g_inv(u, 3.0);
        for (int k = 0; k < D; ++k) {
            double W = (xi[k] - xa[k]) * Z; /* op(*, op(-, p[i], p[a]), Z) */
            y[k] = xa[k] + W;               /* op(+, p[a], W)              */
        }
        corr = (double)(D - 1) * kabc_log_pn(Z);
    } else if (move == 2) {
        /* de_propose :2-22 */
        blk_t B2 = stream(h->seed, w, t, 2, KABC_DOM_AIS_MOVE);
        b = draw_partner(B2.lo, ps, &a, 1);
        const double* xb = h->x + b * D;
        double z[ORC_MAX_DIM + 2];
        for (int j = 0; j < (D + 2) / 2; ++j) {
            blk_t Bn = stream(h->seed, w, t, 3 + j, KABC_DOM_AIS_MOVE);
            kabc_normal_pair(Bn.lo, Bn.hi, &z[2 * j], &z[2 * j + 1]);
        }
        double gamma = 2.38 / kabc_sqrt((double)(2 * D)) * kabc_exp_bounded(z[0] * 0.1);
        for (int k = 0; k < D; ++k) {
            double Wk = (xa[k] - xb[k]) * gamma;
            double s = kabc_fabs(xa[k] - xb[k]) + kabc_fabs(xi[k] - xb[k]) +
                       kabc_fabs(xa[k] - xi[k]);
            double Tk = kabc_div_rc(gamma * s, 300.0, 1.0 / 300.0) * z[1 + k];
            y[k] = xi[k] + Wk + Tk; /* op(+, p[i], W, T) = foldl */
        }
        corr = 0.0;
    } else {
        /* ais_walk_propose :24-43 */
        blk_t B2 = stream(h->seed, w, t, 2, KABC_DOM_AIS_MOVE);
        b = draw_partner(B2.lo, ps, &a, 1);
        int64_t ab[2] = {a, b};
        c = draw_partner(B2.hi, ps, ab, 2);
        const double* xb = h->x + b * D;
        const double* xc = h->x + c * D;
        double z[4];
        for (int j = 0; j < 2; ++j) {
            blk_t Bn = stream(h->seed, w, t, 3 + j, KABC_DOM_AIS_MOVE);
            kabc_normal_pair(Bn.lo, Bn.hi, &z[2 * j], &z[2 * j + 1]);
        }
        for (int k = 0; k < D; ++k) {
            double Xs = kabc_div_rc(xa[k] + (xb[k] + xc[k]), 3.0, 1.0 / 3.0);
            double Wk = z[0] * (xa[k] - Xs) + z[1] * (xb[k] - Xs) + z[2] * (xc[k] - Xs);
            y[k] = xi[k] + Wk;
        }
        corr = 0.0;
    }

    /* ld = loglike(density, push_p(density, p))  (:75) */
    int ev;
    model_push_p(h, y, yp);
    ld_t nw = loglike(h, yp, w, t, KABC_DOM_AIS_COST, &ev);
    ld_t old = {h->lp[i], h->ll[i]};
    h->st.proposals += 1;
    h->st.cost_evals += (uint64_t)ev;

    /* accept(...) src/types.jl:62-75 / :96-104 */
    int acc = 0;
    if (!kabc_isfinite(corr)) return -1; /* "ld_correction is invalid" */
    if (!is_valid(h, old)) return -2;    /* "starting sample invalid." */
    if (is_valid(h, nw)) {
        double e = -kabc_log_pn(kabc_u01(B1.lo)); /* randexp(rng) */
        if (h->posterior == KABC_POSTERIOR_KERNELIZED) {
            double lW = corr + (nw.lp + nw.ll) - (old.lp + old.ll);
            acc = (-e <= lW);
        } else if (h->posterior == KABC_POSTERIOR_COMMON) {
            double lW = corr + nw.ll - old.ll; /* src/types.jl:127 */
            acc = (-e <= lW);
        } else {
            double lW = corr + nw.lp - old.lp;
            double mx = (h->eps > old.ll) ? h->eps : old.ll; /* max(maxcost, old.cost) */
            double lW2 = mx - nw.ll;
            acc = (-e <= lW) && (lW2 >= 0.0);
        }
    }
    if (acc) {
        memcpy(h->x + i * D, y, sizeof(double) * D);
        h->lp[i] = nw.lp;
        h->ll[i] = nw.ll;
        h->st.accepted += 1;
    }
    if (rec) {
        rec->move = move;
        rec->accepted = acc;
        rec->a = (int32_t)a;
        rec->b = (int32_t)b;
        rec->c = (int32_t)c;
        rec->cost_evaluated = ev;
    }
    return acc;
}

static int32_t transition_error(int rc) {
    if (rc == -1) return fail(KABC_ERR_INVALID_STATE, "ld_correction is invalid");
    return fail(KABC_ERR_INVALID_STATE, "starting sample invalid.");
}

/* step(rng, model, spl, state; ntransitions) repeated nsteps times -- src/KissABC.jl:66-80 */
int32_t orc_ais_steps_serial(orc_ais_t* h, int64_t nsteps, int32_t ntransitions, double* out) {
    if (!h->initialised) return fail(KABC_ERR_INVALID_STATE, "orc_ais_init not called");
    const int D = h->D;
    for (int64_t s = 0; s < nsteps; ++s) {
        int64_t i = h->cursor;
        partner_set_t ps = {0, h->N, i};
        for (int r = 0; r < ntransitions; ++r) {
            int rc = transition(h, i, h->tc[i]++, &ps, NULL);
            if (rc < 0) return transition_error(rc);
        }
        if (out) model_push_p(h, h->x + i * D, out + s * D); /* :78 */
        h->cursor = (i + 1) % h->N;                           /* :79 */
    }
    return KABC_OK;
}

int32_t orc_ais_half_generation(orc_ais_t* h, int32_t half, int32_t ntransitions,
                                int64_t row_begin, int64_t row_end) {
    if (!h->initialised) return fail(KABC_ERR_INVALID_STATE, "orc_ais_init not called");
    const int64_t first = half ? h->N0 : 0;
    const int64_t rows = half ? h->N - h->N0 : h->N0;
    if (row_begin < 0 || row_end > rows) return fail(KABC_ERR_INVALID_ARG, "row range");
    partner_set_t ps;
    ps.base = half ? 0 : h->N0;
    ps.n = half ? h->N0 : h->N - h->N0;
    ps.skip = -1;
    for (int64_t r = row_begin; r < row_end; ++r) {
        int64_t i = first + r;
        for (int s = 0; s < ntransitions; ++s) {
            int rc = transition(h, i, h->t + (uint64_t)s, &ps, NULL);
            if (rc < 0) return transition_error(rc);
        }
    }
    return KABC_OK;
}

int32_t orc_ais_end_generation(orc_ais_t* h, int32_t ntransitions) {
    h->t += (uint64_t)ntransitions;
    for (int64_t i = 0; i < h->N; ++i) h->tc[i] = h->t;
    return KABC_OK;
}

int32_t orc_ais_generations_sync(orc_ais_t* h, int64_t ngen, int32_t ntransitions, double* out,
                                 orc_trace_rec_t* trace) {
    if (!h->initialised) return fail(KABC_ERR_INVALID_STATE, "orc_ais_init not called");
    const int D = h->D;
    for (int64_t g = 0; g < ngen; ++g) {
        for (int half = 0; half < 2; ++half) {
            const int64_t first = half ? h->N0 : 0;
            const int64_t rows = half ? h->N - h->N0 : h->N0;
            partner_set_t ps;
            ps.base = half ? 0 : h->N0;
            ps.n = half ? h->N0 : h->N - h->N0;
            ps.skip = -1;
            for (int64_t r = 0; r < rows; ++r) {
                int64_t i = first + r;
                for (int s = 0; s < ntransitions; ++s) {
                    orc_trace_rec_t* rec =
                        trace ? &trace[(g * h->N + i) * ntransitions + s] : NULL;
                    int rc = transition(h, i, h->t + (uint64_t)s, &ps, rec);
                    if (rc < 0) return transition_error(rc);
                }
            }
        }
        orc_ais_end_generation(h, ntransitions);
        if (out)
            for (int64_t i = 0; i < h->N; ++i)
                model_push_p(h, h->x + i * D, out + (g * h->N + i) * D);
    }
    return KABC_OK;
}

int32_t orc_ais_get_state(orc_ais_t* h, double* x, double* lp, double* ll, uint64_t* t) {
    if (x) memcpy(x, h->x, sizeof(double) * h->N * h->D);
    if (lp) memcpy(lp, h->lp, sizeof(double) * h->N);
    if (ll) memcpy(ll, h->ll, sizeof(double) * h->N);
    if (t) *t = h->t;
    return KABC_OK;
}

int32_t orc_ais_set_state(orc_ais_t* h, const double* x, const double* lp, const double* ll,
                          uint64_t t) {
    memcpy(h->x, x, sizeof(double) * h->N * h->D);
    memcpy(h->lp, lp, sizeof(double) * h->N);
    memcpy(h->ll, ll, sizeof(double) * h->N);
    h->t = t;
    for (int64_t i = 0; i < h->N; ++i) h->tc[i] = t;
    h->initialised = 1;
    return KABC_OK;
}

int32_t orc_ais_get_stats(orc_ais_t* h, kabc_stats_t* st) {
    *st = h->st;
    return KABC_OK;
}

/* ------------------------------------------------------------------------- */
/* smc -- src/smc.jl:92-206                                                   */
/* ------------------------------------------------------------------------- */
static int cmp_double(const void* a, const void* b) {
    double x = *(const double*)a, y = *(const double*)b;
    return (x > y) - (x < y);
}

/* Statistics.quantile(v, p), default alpha = beta = 1 ("type 7").  Statistics is a
 * Julia stdlib not present in the reference tree; restated from its documented
 * definition: h = (n-1)p + 1, linear interpolation between the two bracketing
 * order statistics, with the non-finite branch (1-γ)a + γb. */
static int quantile_sorted(const double* v, int64_t n, double p, double* out) {
    if (n < 1) return 0;
    double aleph = (double)n * p + (1.0 - p);
    int64_t j = (int64_t)aleph; /* trunc */
    if (j < 1) j = 1;
    if (j > n - 1) j = n - 1;
    if (n == 1) j = 1;
    double g = aleph - (double)j;
    if (g < 0.0) g = 0.0;
    if (g > 1.0) g = 1.0;
    double a = v[j - 1];
    double b = (n == 1) ? v[0] : v[j];
    if (kabc_isfinite(a) && kabc_isfinite(b))
        *out = a + g * (b - a);
    else
        *out = (1.0 - g) * a + g * b;
    return 1;
}

int32_t orc_quantile(const double* v, int64_t n, double p, double* out) {
    if (n < 1) return fail(KABC_ERR_INVALID_ARG, "collection must be non-empty");
    double* s = (double*)malloc(sizeof(double) * n);
    for (int64_t i = 0; i < n; ++i) {
        if (kabc_isnan(v[i])) {
            free(s);
            return fail(KABC_ERR_NAN_COST, "quantiles are undefined in presence of NaNs");
        }
        s[i] = v[i];
    }
    qsort(s, n, sizeof(double), cmp_double);
    quantile_sorted(s, n, p, out);
    free(s);
    return KABC_OK;
}

int32_t orc_smc_run(const kabc_prior_t* prior, int32_t D, const kabc_cost_t* cost,
                    const kabc_smc_opts_t* o, kabc_smc_result_t* res) {
    const int64_t N = o->nparticles;
    const double alpha = o->alpha;
    const double r_epstol = kabc_isnan(o->r_epstol) ? pow(1.0 - alpha, 1.5) / 50.0 : o->r_epstol;
    const double min_r_ess = kabc_isnan(o->min_r_ess) ? alpha * alpha : o->min_r_ess;
    /* :107-118 */
    if (!(min_r_ess > 0)) return fail(KABC_ERR_INVALID_ARG, "min_r_ess must be > 0.");
    if (!(o->mcmc_retrys >= 0)) return fail(KABC_ERR_INVALID_ARG, "mcmc_retrys must be >= 0.");
    if (!(alpha > 0)) return fail(KABC_ERR_INVALID_ARG, "alpha must be > 0.");
    if (!(r_epstol >= 0)) return fail(KABC_ERR_INVALID_ARG, "r_epstol must be >= 0");
    if (!(o->mcmc_tol >= 0)) return fail(KABC_ERR_INVALID_ARG, "mcmc_tol must be >= 0");
    if (!(o->max_stretch > 1)) return fail(KABC_ERR_INVALID_ARG, "max_stretch must be > 1");
    prep_t q[ORC_MAX_DIM];
    if (!prep_all(prior, D, q)) return fail(KABC_ERR_INVALID_ARG, "invalid prior");
    if (!cost_dim_ok_any(cost->id, D))
        return fail(KABC_ERR_UNSUPPORTED, "cost id / dimension not supported");
    {
        double mn = alpha < min_r_ess ? alpha : min_r_ess;
        int64_t min_n = (int64_t)ceil(3.0 * D / mn);
        if (N < min_n) {
            snprintf(g_err, sizeof g_err, "nparticles must be >= %lld.", (long long)min_n);
            return KABC_ERR_INVALID_ARG;
        }
    }
    const uint64_t seed = o->seed;
    double* th = (double*)malloc(sizeof(double) * N * D);
    double* th2 = (double*)malloc(sizeof(double) * N * D);
    double* Xs = (double*)malloc(sizeof(double) * N);
    double* X2 = (double*)malloc(sizeof(double) * N);
    double* lpi = (double*)malloc(sizeof(double) * N);
    double* lp2 = (double*)malloc(sizeof(double) * N);
    double* tmp = (double*)malloc(sizeof(double) * N);
    uint8_t* alive = (uint8_t*)malloc(N);
    int64_t* idx = (int64_t*)malloc(sizeof(int64_t) * N);
    double* prop = (double*)malloc(sizeof(double) * N * D); /* new_p θp */
    double* lprob = (double*)malloc(sizeof(double) * N);
    double xp[ORC_MAX_DIM];
    uint64_t cost_evals = 0, proposals = 0;
    int32_t rc = KABC_OK;

    /* :119-125 */
    for (int64_t i = 0; i < N; ++i) {
        factored_rand(prior, D, seed, (uint32_t)i, 0, KABC_DOM_SMC_INIT, th + i * D);
        push_p(q, D, th + i * D, xp);
        Xs[i] = orc_cost_eval(cost, D, xp, seed, (uint32_t)i, 0, KABC_DOM_SMC_INIT_COST);
        ++cost_evals;
        lpi[i] = factored_logpdf(q, D, xp);
        alive[i] = 1;
    }
    double eps = KABC_INF;
    int64_t iteration = 0;
    uint64_t pass = 0;
    const int64_t max_it = o->max_iterations > 0 ? o->max_iterations : 100000;
    const double sqrtD = kabc_sqrt((double)D);
    while (1) {
        ++iteration;
        double epsv = eps;
        /* Step 1 :134-143 */
        int64_t na = 0;
        double mn = KABC_INF;
        for (int64_t i = 0; i < N; ++i)
            if (alive[i]) {
                if (kabc_isnan(Xs[i])) {
                    rc = fail(KABC_ERR_NAN_COST, "quantiles are undefined in presence of NaNs");
                    goto done;
                }
                tmp[na++] = Xs[i];
                if (Xs[i] < mn) mn = Xs[i];
            }
        if (na == 0) {
            rc = fail(KABC_ERR_INVALID_STATE, "collection must be non-empty");
            goto done;
        }
        qsort(tmp, na, sizeof(double), cmp_double);
        quantile_sorted(tmp, na, alpha, &eps);
        int flag = 0;
        if (eps > mn) {
            for (int64_t i = 0; i < N; ++i) alive[i] = Xs[i] < eps;
        } else {
            for (int64_t i = 0; i < N; ++i) alive[i] = Xs[i] <= eps;
            flag = 1;
        }
        int64_t ESS = 0;
        for (int64_t i = 0; i < N; ++i) ESS += alive[i];
        int64_t ess_logged = ESS;
        if (o->verbose)
            fprintf(stderr, "(iteration, ϵ, ESS) = (%lld, %.17g, %lld)\n", (long long)iteration,
                    eps, (long long)ESS);
        /* Step 2 :145-153 */
        int resampled = 0;
        if (alpha * (double)ESS <= (double)N * min_r_ess) {
            if (ESS == 0) {
                rc = fail(KABC_ERR_INVALID_STATE, "no alive particle to resample from");
                goto done;
            }
            int64_t m = 0;
            for (int64_t i = 0; i < N; ++i)
                if (alive[i]) idx[m++] = i; /* idxalive = (1:N)[alive] */
            /* idx = repeat(idxalive, ceil(N/m))[1:N] */
            for (int64_t j = 0; j < N; ++j) {
                int64_t src = idx[j % m];
                memcpy(th2 + j * D, th + src * D, sizeof(double) * D);
                X2[j] = Xs[src];
                lp2[j] = lpi[src];
            }
            double* sw;
            sw = th; th = th2; th2 = sw;
            sw = Xs; Xs = X2; X2 = sw;
            sw = lpi; lpi = lp2; lp2 = sw;
            ESS = N;
            memset(alive, 1, N);
            resampled = 1;
        }
        /* Step 3 :156-193 */
        int64_t accepted = 0;
        int passes = 0;
        for (int r = 1; r <= 1 + o->mcmc_retrys; ++r) {
            ++pass;
            ++passes;
            /* new_p = map(1:N) ... :160-167 -- all proposals from the frozen ensemble */
            for (int64_t i = 0; i < N; ++i) {
                if (!alive[i]) continue;
                blk_t B0 = stream(seed, (uint32_t)i, pass, 0, KABC_DOM_SMC_MOVE);
                blk_t B1 = stream(seed, (uint32_t)i, pass, 1, KABC_DOM_SMC_MOVE);
                blk_t B2 = stream(seed, (uint32_t)i, pass, 2, KABC_DOM_SMC_MOVE);
                partner_set_t ps = {0, N, i};
                int64_t a = draw_partner(B0.lo, &ps, NULL, 0);
                int64_t b = draw_partner(B0.hi, &ps, &a, 1);
                double z0, z1;
                kabc_normal_pair(B1.lo, B1.hi, &z0, &z1);
                double s = o->max_stretch * z0 / sqrtD;
                for (int k = 0; k < D; ++k) {
                    double W = (th[b * D + k] - th[a * D + k]) * s;
                    prop[i * D + k] = th[i * D + k] + W;
                }
                lprob[i] = kabc_log(kabc_u01(B2.lo));
                ++proposals;
            }
            /* accept loop :168-191 */
            for (int64_t i = 0; i < N; ++i) {
                if (!alive[i]) continue;
                push_p(q, D, prop + i * D, xp);
                double lpp = factored_logpdf(q, D, xp);
                if (lpp < 0 && !kabc_isfinite(lpp)) continue; /* :173 */
                double lM = lpp - lpi[i] + 0.0;
                if (!(lM < 0.0)) lM = (lM != lM) ? lM : 0.0; /* min(lM, 0.0), NaN propagates */
                if (lprob[i] < lM) {
                    double Xp = orc_cost_eval(cost, D, xp, seed, (uint32_t)i, pass,
                                              KABC_DOM_SMC_COST);
                    ++cost_evals;
                    if (flag) {
                        if (Xp > eps) continue;
                    } else {
                        if (Xp >= eps) continue;
                    }
                    memcpy(th + i * D, prop + i * D, sizeof(double) * D);
                    Xs[i] = Xp;
                    lpi[i] = lpp;
                    ++accepted;
                }
            }
            if ((double)accepted >= o->mcmc_tol * (double)N) break; /* :192 */
        }
        if (res->iter_log && iteration <= res->iter_log_cap) {
            kabc_smc_iter_t* L = &res->iter_log[iteration - 1];
            L->eps = eps;
            L->ess = ess_logged;
            L->accepted = accepted;
            L->resampled = resampled;
            L->flag = flag;
            L->mcmc_passes = passes;
            L->reserved = 0;
        }
        /* :194-198 */
        if (2.0 * fabs(epsv - eps) < r_epstol * (fabs(epsv) + fabs(eps)) || eps <= o->epstol ||
            (double)accepted < o->mcmc_tol * (double)N)
            break;
        if (iteration >= max_it) break;
    }
    /* :200-205 */
    res->n_alive = 0;
    for (int64_t i = 0; i < N; ++i) {
        if (res->theta) push_p(q, D, th + i * D, res->theta + i * D);
        if (res->cost) res->cost[i] = Xs[i];
        if (res->alive) res->alive[i] = alive[i];
        res->n_alive += alive[i];
    }
    res->eps = eps;
    res->iterations = iteration;
    res->cost_evals = cost_evals;
    res->proposals = proposals;
done:
    free(th); free(th2); free(Xs); free(X2); free(lpi); free(lp2); free(tmp);
    free(alive); free(idx); free(prop); free(lprob);
    return rc;
}

/* ------------------------------------------------------------------------- */
/* ABCDE -- src/smc.jl:347-430 (exported, undocumented, untested upstream)    */
/* ------------------------------------------------------------------------- */
int32_t orc_abcde_run(const kabc_prior_t* prior, int32_t D, const kabc_cost_t* cost,
                      const kabc_abcde_opts_t* o, kabc_abcde_result_t* res) {
    if (!(o->alpha >= 0 && o->alpha < 1)) return fail(KABC_ERR_INVALID_ARG, "α must be in 0 <= α < 1.");
    const int64_t N = o->nparticles;
    if (N < 3) return fail(KABC_ERR_INVALID_ARG, "nparticles must be >= 3 (and < 2^31)");
    prep_t q[ORC_MAX_DIM];
    if (!prep_all(prior, D, q)) return fail(KABC_ERR_INVALID_ARG, "invalid prior");
    if (!cost_dim_ok_any(cost->id, D)) return fail(KABC_ERR_UNSUPPORTED, "cost id / dimension not supported");
    const uint64_t seed = o->seed;
    double* th = (double*)malloc(sizeof(double) * N * D);
    double* nth = (double*)malloc(sizeof(double) * N * D);
    double* dl = (double*)malloc(sizeof(double) * N);
    double* ndl = (double*)malloc(sizeof(double) * N);
    double* lp = (double*)malloc(sizeof(double) * N);
    double* nlp = (double*)malloc(sizeof(double) * N);
    double xp[ORC_MAX_DIM], tp[ORC_MAX_DIM];
    uint64_t nsims = 0;
    int32_t rc = KABC_OK;
    /* :349-366 */
    for (int64_t i = 0; i < N; ++i) {
        for (unsigned attempt = 0;; ++attempt) {
            factored_rand(prior, D, seed, (uint32_t)i, attempt, KABC_DOM_ABCDE_INIT, th + i * D);
            push_p(q, D, th + i * D, xp);
            lp[i] = factored_logpdf(q, D, xp);
            int eval = attempt > 0 || kabc_isfinite(lp[i]);
            dl[i] = eval ? orc_cost_eval(cost, D, th + i * D, seed, (uint32_t)i, attempt,
                                         KABC_DOM_ABCDE_INIT_COST)
                         : KABC_NAN;
            if (kabc_isfinite(dl[i]) && kabc_isfinite(lp[i])) break;
            if (attempt >= 100000u) {
                rc = fail(KABC_ERR_RETRY_EXHAUSTED, "ABCDE: the prior never produced a finite (cost, logpdf) pair for some particle");
                goto done;
            }
        }
    }
    const double gamma = o->proposal_width * 2.38 / sqrt((double)(2 * D)); /* :370 */
    int64_t iters = 0;
    while (iters < o->generations) { /* :372 */
        double el = KABC_INF, eh = -KABC_INF;
        for (int64_t i = 0; i < N; ++i) {
            if (dl[i] < el) el = dl[i];
            if (dl[i] > eh) eh = dl[i];
        }
        iters += 1;                                     /* :373: the reference counts the generation first ... */
        if (o->earlystop && eh <= o->eps_target) break; /* :379-381 ... and then leaves (iters includes it) */
        memcpy(nth, th, sizeof(double) * N * D);
        memcpy(ndl, dl, sizeof(double) * N);
        memcpy(nlp, lp, sizeof(double) * N);
        double pop = el + o->alpha * (eh - el);
        double eps_pop = o->eps_target > pop ? o->eps_target : pop;
        for (int64_t i = 0; i < N; ++i) {
            if (o->earlystop && dl[i] <= o->eps_target) continue;
            blk_t B0 = stream(seed, (uint32_t)i, (uint64_t)iters, 0, KABC_DOM_ABCDE_MOVE);
            blk_t B1 = stream(seed, (uint32_t)i, (uint64_t)iters, 1, KABC_DOM_ABCDE_MOVE);
            int64_t s = i;
            double eps = dl[i] <= o->eps_target ? o->eps_target : eps_pop;
            if (dl[i] > eps) { /* s = rand(trng, (1:N)[Δs .<= Δs[i]]) */
                int64_t c = 0;
                for (int64_t j = 0; j < N; ++j) c += dl[j] <= dl[i];
                int64_t m = (int64_t)kabc_index(B0.lo, (uint64_t)c);
                for (int64_t j = 0; j < N; ++j)
                    if (dl[j] <= dl[i]) {
                        if (m == 0) { s = j; break; }
                        --m;
                    }
            }
            partner_set_t ps = {0, N, s};
            int64_t a = draw_partner(B0.hi, &ps, NULL, 0);
            int64_t b = draw_partner(B1.lo, &ps, &a, 1);
            for (int k = 0; k < D; ++k)
                tp[k] = th[s * D + k] + (th[a * D + k] - th[b * D + k]) * gamma;
            push_p(q, D, tp, xp);
            double lpp = factored_logpdf(q, D, xp);
            double wp = lpp - lp[i];
            double mn = wp;
            if (!(wp < 0.0)) mn = (wp != wp) ? wp : 0.0;
            double lu = kabc_log_pn(kabc_u01(B1.hi));
            if (lu > mn) continue; /* :405 */
            nsims += 1;
            double dp = orc_cost_eval(cost, D, tp, seed, (uint32_t)i, (uint64_t)iters, KABC_DOM_ABCDE_COST);
            double thr = eps > dl[i] ? eps : dl[i];
            if (dp <= thr) {
                ndl[i] = dp;
                nlp[i] = lpp;
                memcpy(nth + i * D, tp, sizeof(double) * D);
            }
        }
        double* sw;
        sw = th; th = nth; nth = sw;
        sw = dl; dl = ndl; ndl = sw;
        sw = lp; lp = nlp; nlp = sw;
    }
    {
        double mx = -KABC_INF;
        for (int64_t i = 0; i < N; ++i) {
            if (res->theta) push_p(q, D, th + i * D, res->theta + i * D);
            if (res->cost) res->cost[i] = dl[i];
            if (dl[i] > mx) mx = dl[i];
        }
        res->reached_eps = mx <= o->eps_target;
        res->reserved = 0;
        res->generations_run = iters;
        res->nsims = nsims;
    }
done:
    free(th); free(nth); free(dl); free(ndl); free(lp); free(nlp);
    return rc;
}

/* ------------------------------------------------------------------------- */
/* pfilter -- src/smc.jl:275-340 (exported, undocumented, untested upstream)  */
/* ------------------------------------------------------------------------- */
int64_t orc_pfilter_nparticles(int64_t N, double q, int32_t D) {
    int64_t lowN = 4 * (int64_t)D; /* :276-279 */
    if ((double)N * q <= (double)lowN) N = (int64_t)ceil((double)(lowN + 1) / q);
    return N;
}

int32_t orc_pfilter_run(const kabc_prior_t* prior, int32_t D, const kabc_cost_t* cost,
                        const kabc_pfilter_opts_t* o, kabc_pfilter_result_t* res) {
    prep_t q[ORC_MAX_DIM];
    if (!prep_all(prior, D, q)) return fail(KABC_ERR_INVALID_ARG, "invalid prior");
    if (!cost_dim_ok_any(cost->id, D)) return fail(KABC_ERR_UNSUPPORTED, "cost id / dimension not supported");
    if (!(o->q > 0 && o->q <= 1) || o->nparticles < 1)
        return fail(KABC_ERR_INVALID_ARG, "pfilter needs 0 < q <= 1 and N >= 1");
    const int64_t N = orc_pfilter_nparticles(o->nparticles, o->q, D);
    const uint64_t seed = o->seed;
    double* th = (double*)malloc(sizeof(double) * N * D);
    double* Cc = (double*)malloc(sizeof(double) * N);
    double* lp = (double*)malloc(sizeof(double) * N);
    double* tmp = (double*)malloc(sizeof(double) * N);
    int64_t* idxok = (int64_t*)malloc(sizeof(int64_t) * N);
    uint8_t* bad = (uint8_t*)malloc(N);
    double xp[ORC_MAX_DIM], p[ORC_MAX_DIM];
    uint64_t total_reps = 0, cost_evals = 0;
    int32_t rc = KABC_OK;
    for (int64_t i = 0; i < N; ++i) { /* :280-294 */
        for (unsigned attempt = 0;; ++attempt) {
            factored_rand(prior, D, seed, (uint32_t)i, attempt, KABC_DOM_PF_INIT, th + i * D);
            push_p(q, D, th + i * D, xp);
            lp[i] = factored_logpdf(q, D, xp);
            int eval = attempt > 0 || kabc_isfinite(lp[i]);
            Cc[i] = eval ? orc_cost_eval(cost, D, th + i * D, seed, (uint32_t)i, attempt, KABC_DOM_PF_INIT_COST)
                         : KABC_NAN;
            if (kabc_isfinite(Cc[i]) && kabc_isfinite(lp[i])) break;
            if (attempt >= 100000u) {
                rc = fail(KABC_ERR_RETRY_EXHAUSTED, "pfilter: the prior never produced a finite (cost, logpdf) pair for some particle");
                goto done;
            }
        }
    }
    int64_t iters = 0;
    double eps = 0.0, eff = 0.0;
    while (1) {
        iters += 1;
        memcpy(tmp, Cc, sizeof(double) * N);
        qsort(tmp, N, sizeof(double), cmp_double);
        quantile_sorted(tmp, N, o->q, &eps); /* ϵ = quantile(C, q) */
        int64_t nok = 0, nbad = 0;
        for (int64_t i = 0; i < N; ++i) {
            bad[i] = Cc[i] > eps;
            if (bad[i]) ++nbad; else idxok[nok++] = i;
        }
        uint64_t nreps = 0;
        for (int64_t i = 0; i < N; ++i) {
            if (!bad[i]) continue;
            for (uint32_t attempt = 0;; ++attempt) { /* @label resample */
                uint64_t t = ((uint64_t)iters << 24) | attempt;
                blk_t B0 = stream(seed, (uint32_t)i, t, 0, KABC_DOM_PF_MOVE);
                blk_t B1 = stream(seed, (uint32_t)i, t, 1, KABC_DOM_PF_MOVE);
                blk_t B2 = stream(seed, (uint32_t)i, t, 2, KABC_DOM_PF_MOVE);
                partner_set_t ps = {0, nok, -1};
                int64_t pb = draw_partner(B0.lo, &ps, NULL, 0);
                int64_t pc = draw_partner(B0.hi, &ps, &pb, 1);
                int64_t bc[2] = {pb, pc};
                int64_t pd = draw_partner(B1.lo, &ps, bc, 2);
                int64_t b = idxok[pb], c = idxok[pc], d = idxok[pd];
                double z0, z1;
                kabc_normal_pair(B2.lo, B2.hi, &z0, &z1);
                double sc = z0 * o->proposal_width;
                for (int k = 0; k < D; ++k) p[k] = th[b * D + k] + (th[d * D + k] - th[c * D + k]) * sc;
                nreps += 1;
                push_p(q, D, p, xp);
                double ll = factored_logpdf(q, D, xp);
                double wp = ll - lp[i];
                double mn = wp;
                if (!(wp < 0.0)) mn = (wp != wp) ? wp : 0.0;
                double lu = kabc_log_pn(kabc_u01(B1.hi));
                if (lu > mn) continue; /* @goto resample */
                double Cp = orc_cost_eval(cost, D, p, seed, (uint32_t)i, t, KABC_DOM_PF_COST);
                cost_evals += 1;
                if (Cp > eps) continue;
                Cc[i] = Cp;
                memcpy(th + i * D, p, sizeof(double) * D);
                lp[i] = ll;
                break;
            }
        }
        total_reps += nreps;
        eff = (double)nbad / (double)nreps;
        if (eff < o->eff_tol) break;
        if (eps < o->epstol) break;
        if (o->max_iters >= 0 && iters > o->max_iters) break;  /* :332; < 0 = Inf */
        if (!(nreps > 0)) break;
    }
    for (int64_t i = 0; i < N; ++i) {
        if (res->theta) push_p(q, D, th + i * D, res->theta + i * D);
        if (res->cost) res->cost[i] = Cc[i];
    }
    res->eps = eps;
    res->eff = eff;
    res->iterations = iters;
    res->nreps = total_reps;
    res->cost_evals = cost_evals;
done:
    free(th); free(Cc); free(lp); free(tmp); free(idxok); free(bad);
    return rc;
}
