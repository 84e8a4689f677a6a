// kabc_device.hpp -- device-side building blocks shared by the gfx950 kernels.
//
// Data layout in HBM (DESIGN.md "Layout"):
//   half h of the ensemble : row-major [rows_h][D] f64  (one walker = one
//     D*8-byte row; D = 8 -> one 64-byte line, fetched by a single lane with
//     dwordx4 loads both for its own row and for a randomly drawn partner)
//   logprior / loglik|cost : [rows_owned] f64 each, per half
// Random draws are counter-based (include/kabc_philox.h): no RNG state in memory.
#pragma once

#ifndef __HIPCC_RTC__ /* hipRTC: built-in runtime declarations, no system headers */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <utility>
#endif

#include "kabc.h"
#include "kabc_costs.h"
#include "kabc_math.h"
#include "kabc_philox.h"
#include "kabc_sampling.h"

namespace kabc {

constexpr int kWave = 64;

// one prepared component of Factored(...): parameters + derived normalisers
struct PriorDev {
    int32_t kind;
    int32_t discrete;
    double p[4];
    double c0, c1;
    double rb;  // RN(1/p[1]) (1/p[0] for Exponential), for kabc_div_rc
};

struct PriorSet {
    PriorDev c[KABC_MAX_DIM];
};

// Distributions.logpdf(p_k, x) on the device.  Same formulas, same operation
// order as the host-side definition pinned by tests/golden/priors_logpdf.json.
// Two code shapes with identical results:
//   comp_logpdf_simple : the families whose logpdf needs no transcendental per
//       walker (Uniform, Normal, truncated Normal, DiscreteUniform, Exponential);
//       inlined into the hot loop.
//   comp_logpdf_general: every family; out of line for D > 8 so that D copies of
//       log/log1p/lgamma do not bloat the kernel past the instruction cache.
__device__ __forceinline__ double comp_logpdf_simple(int kind, const PriorDev& q, double x) {
    const double a = q.p[0], b = q.p[1], rb = q.rb;
    switch (kind) {
        case KABC_PRIOR_UNIFORM: return (x >= a && x <= b) ? q.c0 : -KABC_INF;
        case KABC_PRIOR_NORMAL: {
            const double z = kabc_div_rc(x - a, b, rb);
            return -(z * z + KABC_LOG_2PI) / 2.0 - q.c0;
        }
        case KABC_PRIOR_TRUNCNORMAL: {
            if (!(x >= q.p[2] && x <= q.p[3])) return -KABC_INF;
            const double z = kabc_div_rc(x - a, b, rb);
            return -(z * z + KABC_LOG_2PI) / 2.0 - q.c0 - q.c1;
        }
        case KABC_PRIOR_DISCRETE_UNIFORM:
            return (x >= a && x <= b && x == kabc_rint(x)) ? q.c0 : -KABC_INF;
        case KABC_PRIOR_EXPONENTIAL: return (x >= 0.0) ? -q.c0 - kabc_div_rc(x, a, rb) : -KABC_INF;
        default: return KABC_NAN;
    }
}

// no active lane of the wavefront satisfies c (one v_cmp / s_cmp pair: wave-uniform result)
__device__ __forceinline__ bool wave_none(bool c) { return __builtin_amdgcn_ballot_w64(c) == 0ull; }

// entries of a per-component lgamma(k + r) table (NegativeBinomial, see the family's case below)
constexpr int kNbEntries = 256;
// blocks of a kernel's NegativeBinomial tables: [0, kNbLg1Block) lgamma(k + r) of the first
// components, block kNbLg1Block the copy of kabc_lgamma1_tab (lgamma(k + 1))
constexpr int kNbLg1Block = 2;
// tab: kabc_log_tab or the kernel's LDS copy of it (same values; the global table is a dependent
// L2 round trip per log on the consumer's chain)
__device__ __forceinline__ double comp_logpdf_general_body(int kind, double a, double b, double p2,
                                                           double p3, double c0, double c1, double rb,
                                                           double x, const double* tab = kabc_log_tab,
                                                           const double* nbtab = nullptr) {
    switch (kind) {
        case KABC_PRIOR_UNIFORM: return (x >= a && x <= b) ? c0 : -KABC_INF;
        case KABC_PRIOR_NORMAL: {
            const double z = kabc_div_rc(x - a, b, rb);
            return -(z * z + KABC_LOG_2PI) / 2.0 - c0;
        }
        case KABC_PRIOR_TRUNCNORMAL: {
            if (!(x >= p2 && x <= p3)) return -KABC_INF;
            const double z = kabc_div_rc(x - a, b, rb);
            return -(z * z + KABC_LOG_2PI) / 2.0 - c0 - c1;
        }
        case KABC_PRIOR_BETA: {
            // Regular lanes: 2^-7 <= x < 1 -- x is a positive normal number and log1p(-x) sits on
            // its table branch.  When no lane in the support is irregular (a wave-uniform test:
            // one ballot), both logs run WITHOUT their special-case selects (zero / subnormal /
            // negative / inf for log; the small-argument polynomial, |x| >= 1 and x <= -1 for
            // log1p): kabc__log_core / kabc__log1p_core are what kabc_log_t / kabc_log1p_t
            // evaluate on such arguments, so the bits are the same, for ~35 instructions less per
            // transition on the consumer wave.  Lanes outside the support compute garbage that the
            // select below discards (the table index is masked, nothing can trap).
            const bool in = (x >= 0.0 && x <= 1.0);
            double lx, l1;
            if (wave_none(in && !(x >= 0x1p-7 && x < 1.0))) {
                lx = kabc__log_core(kabc_bits(x), 0, tab);
                const double mx = -x, u = 1.0 + mx;
                l1 = kabc__log1p_core(kabc_bits(u), mx - (u - 1.0), tab);
            } else {
                lx = kabc_log_t(x, tab);
                l1 = kabc_log1p_t(-x, tab);
            }
            const double t1 = (a == 1.0) ? 0.0 : (a - 1.0) * lx;
            const double t2 = (b == 1.0) ? 0.0 : (b - 1.0) * l1;
            return in ? t1 + t2 - c0 : -KABC_INF;
        }
        case KABC_PRIOR_DISCRETE_UNIFORM:
            return (x >= a && x <= b && x == kabc_rint(x)) ? c0 : -KABC_INF;
        case KABC_PRIOR_NEGBINOMIAL: {
            if (!(x >= 0.0) || x != kabc_rint(x)) return -KABC_INF;
            // lgamma(x + 1) of the integer x: a lookup of kabc_lgamma's own values below 256
            // (include/kabc_math.h kabc_lgamma1p_int_t), ~150 instructions less per transition.
            // lgamma(x + r): the AIS kernel tabulates it per NegativeBinomial component for the
            // launch (kNbEntries values of kabc_lgamma_t itself in LDS, ais_kernels.hpp); p2 is
            // then the component's table slot (a field the family does not use), else < 0 / no table
            // With tables (nbtab != NULL) lgamma(x + 1) comes from their last block too: the
            // kernel's LDS copy of kabc_lgamma1_tab -- the table in global memory is a dependent
            // L2 round trip per transition on the consumer's chain, and its wait also drains
            // the partner-row prefetch of the next sub-step.
            const bool in = (x >= 0.0) && (x < (double)kNbEntries);
            const int ix = in ? (int)x : 0;
            double lga, lg1;
            if (nbtab != nullptr) {  // (wave-uniform)
                static_assert(kNbEntries == KABC_LGAMMA1_N, "one index serves both tables");
                lg1 = nbtab[kNbLg1Block * kNbEntries + ix];
                if (p2 >= 0.0) {     // (wave-uniform)
                    lga = nbtab[(int)p2 * kNbEntries + ix];
                    if (x >= (double)kNbEntries) {
                        lga = kabc_lgamma_t(x + a, tab);
                        lg1 = kabc_lgamma_t(x + 1.0, tab);
                    }
                } else {
                    lga = kabc_lgamma_t(x + a, tab);
                    if (x >= (double)kNbEntries) lg1 = kabc_lgamma_t(x + 1.0, tab);
                }
            } else {
                lga = kabc_lgamma_t(x + a, tab);
                lg1 = kabc_lgamma1p_int_t(x, tab);
            }
            return c0 + x * c1 + lga - lg1;
        }
        case KABC_PRIOR_EXPONENTIAL: return (x >= 0.0) ? -c0 - kabc_div_rc(x, a, rb) : -KABC_INF;
        case KABC_PRIOR_GAMMA: {
            // (regular lanes: x positive, normal and finite -- the log core without its
            // special-case selects when no lane in the support is anything else; see Beta)
            const bool in = (x >= 0.0);
            const double lx = wave_none(in && !(x >= 0x1p-1022 && x < KABC_INF)) ? kabc__log_core(kabc_bits(x), 0, tab)
                                                                                 : kabc_log_t(x, tab);
            const double t1 = (a == 1.0) ? 0.0 : (a - 1.0) * lx;
            return in ? t1 - kabc_div_rc(x, b, rb) - c0 : -KABC_INF;
        }
        case KABC_PRIOR_LOGNORMAL: {
            const bool in = (x > 0.0);
            const double lx = wave_none(in && !(x >= 0x1p-1022 && x < KABC_INF)) ? kabc__log_core(kabc_bits(x), 0, tab)
                                                                                 : kabc_log_t(x, tab);
            const double z = kabc_div_rc(lx - a, b, rb);
            return in ? -(z * z + KABC_LOG_2PI) / 2.0 - c0 - lx : -KABC_INF;
        }
        default: {
            // a user family (kabc_compile_prior_plugin): the snippet's kabc_user_prior_logpdf,
            // compiled into this translation unit in front of the kernels (capi_plugin.hip)
#ifdef KABC_USER_PRIOR_LOGPDF
            if (kind >= KABC_PRIOR_USER) {
                const double pp[4] = {a, b, p2, p3};
                return KABC_USER_PRIOR_LOGPDF(kind, x, pp, tab);
            }
#endif
            return KABC_NAN;
        }
    }
}

static __device__ __noinline__ double comp_logpdf_general(int kind, double a, double b, double p2,
                                                          double p3, double c0, double c1, double rb,
                                                          double x, const double* tab,
                                                          const double* nbtab = nullptr) {
    return comp_logpdf_general_body(kind, a, b, p2, p3, c0, c1, rb, x, tab, nbtab);
}

// Inlined up to kGeneralInlineD components: a call costs the caller its live registers
// (spilled around it) -- 19 % of a launch at D = 4; beyond that D copies of the switch
// would bloat the kernel (and the build) for little.
constexpr int kGeneralInlineD = 8;
template <int D = KABC_MAX_DIM>
__device__ __forceinline__ double comp_logpdf(int kind, const PriorDev& q, double x,
                                              const double* tab = kabc_log_tab,
                                              const double* nbtab = nullptr) {
    if constexpr (D <= kGeneralInlineD)
        return comp_logpdf_general_body(kind, q.p[0], q.p[1], q.p[2], q.p[3], q.c0, q.c1, q.rb, x, tab, nbtab);
    else
        return comp_logpdf_general(kind, q.p[0], q.p[1], q.p[2], q.p[3], q.c0, q.c1, q.rb, x, tab, nbtab);
}

// the prepared block of an MvNormal prior (kabc_mvnormal.h) as a wave-uniform pointer
__device__ __forceinline__ const double* mvn_block(const PriorDev& q) {
    const uint64_t b = kabc_bits(q.p[2]);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(b >> 32));
    return (const double*)(uintptr_t)(((uint64_t)hi << 32) | lo);
}

// The whole MvNormal log-density.  (Inline: as an out-of-line function it was the first call in
// the smc kernels' propose/accept pass, and smc_loop_kernel -- 256 VGPRs + AGPR spill space -- then
// produced different trajectories for OTHER priors; `tests/test_gpu_random_sweep.py`.)
template <int D>
struct MvnRow {
    double v[D];
};
template <int D>
__device__ __forceinline__ double mvn_logpdf_total(const double* blk, MvnRow<D> x) {
    double sm = 0.0;
#pragma unroll
    for (int k = 0; k < D; ++k) {
        const double l = kabc_mvn_logpdf_comp(blk, D, k, x.v);
        sm = (k == 0) ? l : sm + l;
    }
    return sm;
}

// A JOINT user prior (kabc_compile_mvprior_plugin, include/kabc.h: any multivariate Distribution as the
// prior -- src/types.jl:30,34-35,52, src/smc.jl:92-93): every component carries its kind, and the
// log-density of the push_p'ed VECTOR replaces the sum over components (NaN for such a kind).  P: the
// prepared components (component k's parameters at P[k].p[0..2]); a unit without joint families compiles
// this to `return s`.
__device__ __forceinline__ double joint_logpdf_or(double s, int kind0, const double* xp, int D,
                                                  const PriorDev* P, const double* tab) {
#ifdef KABC_USER_MVPRIOR_LOGPDF
    if (kind0 >= KABC_PRIOR_USER && KABC_USER_PRIOR_IS_JOINT(kind0))
        return KABC_USER_MVPRIOR_LOGPDF(kind0, xp, D, &P[0].p[0], (int)(sizeof(PriorDev) / sizeof(double)), tab);
#endif
    (void)kind0;
    (void)xp;
    (void)D;
    (void)P;
    (void)tab;
    return s;
}

// host-side classification used to pick the kernel variant
inline bool prior_is_simple(int kind) {
    return kind == KABC_PRIOR_UNIFORM || kind == KABC_PRIOR_NORMAL ||
           kind == KABC_PRIOR_TRUNCNORMAL || kind == KABC_PRIOR_DISCRETE_UNIFORM ||
           kind == KABC_PRIOR_EXPONENTIAL;
}

#ifdef KABC_MODEL_SPEC
// A translation unit generated for ONE model (kabc_compile_model, capi_plugin.hip) defines
// namespace kabc_mspec { D, KIND[D], DISC[D], P[D][4], C0[D], C1[D], RB[D] } as constexpr data:
// push_p + logpdf(d::Factored, x) with every component's family and parameters known to the
// compiler -- the family switch folds away, the parameters are literals instead of LDS records,
// `(alpha == 1) ? 0 : ...` and the like are decided at compile time.  Same formulas, same
// operation order (comp_logpdf_general_body itself, inlined D times): same bits.
constexpr int model_nb_slot(int k) {  // the NegativeBinomial components' lgamma(j + r) tables
    int slot = 0;
    for (int j = 0; j < k; ++j) slot += kabc_mspec::KIND[j] == KABC_PRIOR_NEGBINOMIAL ? 1 : 0;
    return slot;
}
template <int D, int NBTABS, int K>
__device__ __forceinline__ double model_comp(const double* x, double* xp, const double* tab,
                                             const double* nbtab) {
    constexpr int kind = kabc_mspec::KIND[K];
    const double v = kabc_mspec::DISC[K] ? kabc_rint(x[K]) : x[K];
    xp[K] = v;
    // p[2] of a NegativeBinomial component is its table slot when the kernel keeps tables
    constexpr double p2 = (kind == KABC_PRIOR_NEGBINOMIAL)
                              ? (model_nb_slot(K) < NBTABS ? (double)model_nb_slot(K) : -1.0)
                              : kabc_mspec::P[K][2];
    return comp_logpdf_general_body(kind, kabc_mspec::P[K][0], kabc_mspec::P[K][1], p2,
                                    kabc_mspec::P[K][3], kabc_mspec::C0[K], kabc_mspec::C1[K],
                                    kabc_mspec::RB[K], v, tab, nbtab);
}
// (left to right: ((l0 + l1) + l2) + ..., as src/priors.jl:30-36 sums; a recursive template --
// hipRTC has no <utility>)
template <int D, int NBTABS, int K>
struct ModelSum {
    static __device__ __forceinline__ double run(const double* x, double* xp, const double* tab,
                                                 const double* nbtab) {
        const double s = ModelSum<D, NBTABS, K - 1>::run(x, xp, tab, nbtab);
        return s + model_comp<D, NBTABS, K>(x, xp, tab, nbtab);
    }
};
template <int D, int NBTABS>
struct ModelSum<D, NBTABS, 0> {
    static __device__ __forceinline__ double run(const double* x, double* xp, const double* tab,
                                                 const double* nbtab) {
        return model_comp<D, NBTABS, 0>(x, xp, tab, nbtab);
    }
};
template <int D, int NBTABS = 0>
__device__ __forceinline__ double model_logpdf_push(const double* x, double* xp, const double* tab,
                                                    const double* nbtab) {
    static_assert(D == kabc_mspec::D, "this translation unit was generated for another length(prior)");
    return ModelSum<D, NBTABS, D - 1>::run(x, xp, tab, nbtab);
}
#endif

// push_p (src/types.jl:27-32) followed by logpdf(d::Factored, x) = left-to-right
// sum over components (src/priors.jl:30-36).  xp receives push_p(x).
// P: the prepared components (in LDS on the hot path).
// FENCED: a scheduling fence per component (the AIS consumer needs it to stay inside its
// register budget); without it the compiler hoists the components' parameter reads and their
// latencies overlap (smc_loop_kernel: 2.3 -> ... us for 16 components read from LDS)
template <int D, bool SIMPLE = false, bool FENCED = true>
__device__ __forceinline__ double factored_logpdf_push(const PriorDev* __restrict__ P,
                                                       const double* x, double* xp,
                                                       const double* tab = kabc_log_tab,
                                                       const double* nbtab = nullptr) {
#ifdef KABC_MODEL_SPEC
    if constexpr (D == kabc_mspec::D) {
        (void)P;
        return model_logpdf_push<D>(x, xp, tab, nullptr);
    }
#endif
    double s = 0.0;
    [[maybe_unused]] int kind0 = 0;
#pragma unroll
    for (int k = 0; k < D; ++k) {
        if constexpr (FENCED) __builtin_amdgcn_sched_barrier(0);
        const PriorDev& q = P[k];
        // the prior is the same for every lane: kind / discrete as SCALARS make the family
        // dispatch a uniform branch instead of an exec-masked walk through every case
        const int kind = __builtin_amdgcn_readfirstlane(q.kind);
        if (k == 0) kind0 = kind;
        const bool disc = __builtin_amdgcn_readfirstlane(q.discrete) != 0;
        const double v = disc ? kabc_rint(x[k]) : x[k];
        xp[k] = v;
        const double l = SIMPLE ? comp_logpdf_simple(kind, q, v) : comp_logpdf<D>(kind, q, v, tab, nbtab);
        s = (k == 0) ? l : s + l;
    }
#ifndef KABC_NO_MVN_DEVICE  // (A/B builds: the case compiled out)
    if constexpr (!SIMPLE) {
        // A full-covariance MvNormal (kabc_mvnormal.h) is all D components or none.  Its
        // components went through the loop above like any other (continuous: xp = x; the family
        // switch returned NaN for each), so that the two paths meet on ONE value, the sum -- a
        // branch in front of the loop made them meet on all of xp as well, which cost every other
        // GENERAL prior 3-5 % of an AIS launch in register copies.  The block pointer is
        // wave-uniform.
        if (kind0 == KABC_PRIOR_MVNORMAL) {  // (component 0's family: the scalar the loop read)
            MvnRow<D> r;
#pragma unroll
            for (int j = 0; j < D; ++j) r.v[j] = xp[j];
            s = mvn_logpdf_total<D>(mvn_block(P[0]), r);
        }
    }
#endif
    if constexpr (!SIMPLE) s = joint_logpdf_or(s, kind0, xp, D, P, tab);
    return s;
}
template <int D, bool SIMPLE = false>
__device__ __forceinline__ double factored_logpdf_push(const PriorSet& P, const double* x,
                                                       double* xp) {
    return factored_logpdf_push<D, SIMPLE>(P.c, x, xp);
}

// Constant- or Gaussian-in-a-box components (the AIS kernel's SIMPLE / NORMAL prior classes, the smc
// loop kernel's SIMPLE class): one 64-byte LDS record per component,
//   Gaussian-in-a-box : { mu, sigma, RN(1/sigma), c0 = log sigma, c1 (0 unless truncated), lo, hi, - }
//   constant-in-a-box : { lo, hi, -, c0, -, -, -, - }
// (lo / hi of an untruncated Normal are -Inf / +Inf).  Bit g of gmask: component g is Gaussian;
// dmask: push_p rounds it.
struct GaussBoxPrior {
    const double (*rec)[8];  // LDS, [D][8]
    uint32_t gmask, dmask;
};
__device__ __forceinline__ bool gaussbox_is_gauss(int kind) {
    return kind == KABC_PRIOR_NORMAL || kind == KABC_PRIOR_TRUNCNORMAL;
}
// the record of one prepared component (thread k of a workgroup fills record k)
__device__ __forceinline__ void gaussbox_stage(double* r, const PriorDev& q) {
    const bool tr = q.kind == KABC_PRIOR_TRUNCNORMAL;
    r[0] = q.p[0];
    r[1] = q.p[1];
    r[2] = q.rb;
    r[3] = q.c0;
    r[4] = tr ? q.c1 : 0.0;
    r[5] = tr ? q.p[2] : -KABC_INF;
    r[6] = tr ? q.p[3] : KABC_INF;
    r[7] = gaussbox_is_gauss(q.kind) ? 1.0 : 0.0;
}
// host-side: can every component of the prior be a record of this kind?
inline bool prior_is_gaussbox(int kind) {
    return kind == KABC_PRIOR_UNIFORM || kind == KABC_PRIOR_DISCRETE_UNIFORM ||
           kind == KABC_PRIOR_NORMAL || kind == KABC_PRIOR_TRUNCNORMAL;
}

// push_p (src/types.jl:27-32) + logpdf(d::Factored, x) (src/priors.jl:30-36) for these classes.
// Returns the left-to-right sum of the components' in-support values and, in `in`, whether
// every component is inside its support; the caller takes lp = in ? sum : -Inf, which is what
// the reference's sum is when a term is -Inf (no term can be +Inf).  Same formulas and operation
// order as comp_logpdf_simple -- -(z^2 + log 2pi)/2 - log sigma [- c1], z = (x - mu)/sigma --
// with "- c1" also applied to an untruncated Normal, where c1 = 0 and x - 0 = x exactly.
// What changed against the per-component family switch (round 2): a Normal component executed
// ~30 instructions, four dependent LDS reads among them, each behind its own wait -- here the
// parameters of four components arrive in one batch and a component is 9 VALU instructions.
template <int D, bool ALLNORMAL, bool HASDISC>
__device__ __forceinline__ double gaussbox_logpdf_push(const GaussBoxPrior& G, const double* x,
                                                       double* xp, bool& in_out) {
    constexpr int GRP = D <= 8 ? 4 : 2;  // (D = 16 with four records in flight spilled 150 registers)
    bool in = true;
    double s = 0.0;
#pragma unroll
    for (int k0 = 0; k0 < D; k0 += GRP) {
        double q[GRP][7];
#pragma unroll
        for (int j = 0; j < GRP; ++j) {
            if (k0 + j < D) {
#pragma unroll
                for (int w = 0; w < (ALLNORMAL ? 4 : 7); ++w) q[j][w] = G.rec[k0 + j][w];
            }
        }
#pragma unroll
        for (int j = 0; j < GRP; ++j) {
            const int k = k0 + j;
            if (k < D) {
                double v = x[k];
                if constexpr (HASDISC) v = ((G.dmask >> k) & 1u) ? kabc_rint(v) : v;
                xp[k] = v;
                double l;
                if (ALLNORMAL || ((G.gmask >> k) & 1u)) {
                    if constexpr (!ALLNORMAL) in = in && (v >= q[j][5]) && (v <= q[j][6]);
                    const double z = kabc_div_rc(v - q[j][0], q[j][1], q[j][2]);
                    l = -(z * z + KABC_LOG_2PI) / 2.0 - q[j][3];
                    if constexpr (!ALLNORMAL) l = l - q[j][4];
                } else {
                    in = in && (v >= q[j][0]) && (v <= q[j][1]);
                    l = q[j][3];
                }
                s = (k == 0) ? l : s + l;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    in_out = in;
    return s;
}

// words of parameter-independent "prepared" state of a cost (include/kabc_costs.h)
#ifndef KABC_USER_AUX_WORDS
#define KABC_USER_AUX_WORDS_OR_0 0
#else
#define KABC_USER_AUX_WORDS_OR_0 KABC_USER_AUX_WORDS
#endif
constexpr int cost_aux_c(int cost) {
    return cost == KABC_COST_NORMAL_MEANSTD_SIM ? 2 : cost >= KABC_COST_USER ? KABC_USER_AUX_WORDS_OR_0 : 0;
}
static_assert(KABC_USER_AUX_WORDS_OR_0 <= KABC_COST_MAX_AUX, "KABC_USER_AUX_WORDS too large");

// leading blocks of a cost's stream that are consumed as normal pairs (kabc_cost_rng_normal2)
// whatever the parameters: these can be expanded ahead of the cost evaluation
// (KABC_RNG_PREFETCH, include/kabc_philox.h)
constexpr int cost_pre_blocks(int cost, int D) {
    return cost == KABC_COST_HIER_GAUSS_SIM ? (D - 2 + 1) / 2
         : (cost == KABC_COST_NOISY_QUAD_DU || cost == KABC_COST_MIXTURE ||
            cost == KABC_COST_NOISY_BANANA) ? 1 : 0;
}

// Leading cost parameters a kernel may keep in REGISTERS for a whole launch (they are the same for
// every lane and every transition).  Read through the pointer they are a global load per
// transition on the consumer's dependent chain (the NORMAL-class kernel fetched gauss_dist's D
// centres with four `global_load_dwordx4` in every sub-step).
constexpr int cost_reg_params(int cost, int D) {
    return cost == KABC_COST_GAUSS_DIST ? (D <= 8 ? D : 0)
         : cost == KABC_COST_NORMAL_MEANSTD_SIM ? 3
         : (cost == KABC_COST_DIRAC_SQ || cost == KABC_COST_ABS_DIFF || cost == KABC_COST_NORM_SHELL ||
            cost == KABC_COST_NOISY_QUAD_DU || cost == KABC_COST_MIXTURE || cost == KABC_COST_NOISY_BANANA) ? 1
         : 0;
}

// compile-time cost dispatch on the DeviceCost id (formulas: include/kabc_costs.h)
template <int COST, int D>
__device__ __forceinline__ double eval_cost(const double* xp, const double* __restrict__ params,
                                            const double* __restrict__ data, int64_t ndata,
                                            kabc_cost_rng_t* rng) {
    if constexpr (COST == KABC_COST_GAUSS_DIST) return kabc_cost_gauss_dist(xp, D, params);
    else if constexpr (COST == KABC_COST_ROSENBROCK) return kabc_cost_rosenbrock(xp, D);
    else if constexpr (COST == KABC_COST_HIER_GAUSS_SIM) return kabc_cost_hier_gauss_sim(xp, D, data, rng);
    else if constexpr (COST == KABC_COST_NORMAL_MEANSTD_SIM) return kabc_cost_normal_meanstd_sim(xp, params, rng);
    else if constexpr (COST == KABC_COST_DIRAC_SQ) return kabc_cost_dirac_sq(xp, params);
    else if constexpr (COST == KABC_COST_ABS_DIFF) return kabc_cost_abs_diff(xp, params);
    else if constexpr (COST == KABC_COST_NORM_SHELL) return kabc_cost_norm_shell(xp, D, params);
    else if constexpr (COST == KABC_COST_NOISY_QUAD_DU) return kabc_cost_noisy_quad_du(xp, params, rng);
    else if constexpr (COST == KABC_COST_MIXTURE) return kabc_cost_mixture(xp, params, rng);
    else if constexpr (COST == KABC_COST_NOISY_BANANA) return kabc_cost_noisy_banana(xp, params, rng);
    else if constexpr (COST == KABC_COST_WIENER_RMS) return kabc_cost_wiener_rms(xp, data, ndata, rng);
#ifdef KABC_USER_COST_DEFINED
    else if constexpr (COST == KABC_COST_USER) return kabc_user_cost(xp, D, params, data, ndata, rng);
#endif
    else return KABC_NAN;
}

// row load/store: 16-byte vector accesses when the row is a multiple of 16 bytes
template <int D>
__device__ __forceinline__ void load_row(const double* __restrict__ p, double* out) {
    if constexpr (D % 2 == 0) {
        const double2* q = reinterpret_cast<const double2*>(p);
#pragma unroll
        for (int k = 0; k < D / 2; ++k) {
            const double2 v = q[k];
            out[2 * k] = v.x;
            out[2 * k + 1] = v.y;
        }
    } else {
#pragma unroll
        for (int k = 0; k < D; ++k) out[k] = p[k];
    }
}
template <int D>
__device__ __forceinline__ void store_row(double* __restrict__ p, const double* v) {
    if constexpr (D % 2 == 0) {
        double2* q = reinterpret_cast<double2*>(p);
#pragma unroll
        for (int k = 0; k < D / 2; ++k) q[k] = make_double2(v[2 * k], v[2 * k + 1]);
    } else {
#pragma unroll
        for (int k = 0; k < D; ++k) p[k] = v[k];
    }
}

// sum a per-lane counter over the wave (DPP/permute shuffles) -- result in lane 0
__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v) {
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
    return v;
}

// inclusive prefix sum over the 64 lanes with DPP (row_shr 1/2/4/8, row_bcast 15/31): six
// VALU instructions instead of six LDS-crossbar shuffles (each ~100 cycles of latency)
__device__ __forceinline__ unsigned wave_scan_incl(unsigned v) {
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);  // row_shr:1
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);  // row_shr:2
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);  // row_shr:4
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);  // row_shr:8
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast:15
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);  // row_bcast:31
    return v;
}

// sum of a 32-bit per-lane counter over the wavefront, as a wave-uniform value (the inclusive
// scan's last lane): ~8 VALU instead of 12 dependent LDS-crossbar shuffles (~1 us at the very
// end of every launch of the AIS kernel)
__device__ __forceinline__ unsigned wave_total_u32(unsigned v) {
    return (unsigned)__builtin_amdgcn_readlane((int)wave_scan_incl(v), kWave - 1);
}

// compile-time mirror of kabc_cost_dim_ok (include/kabc_costs.h)
constexpr bool cost_dim_ok_c(int id, int D) {
    switch (id) {
        case KABC_COST_GAUSS_DIST: return D >= 1;
        case KABC_COST_ROSENBROCK: return D >= 2;
        case KABC_COST_HIER_GAUSS_SIM: return D >= 3;
        case KABC_COST_NORMAL_MEANSTD_SIM: return D == 2;
        case KABC_COST_DIRAC_SQ: return D == 1;
        case KABC_COST_ABS_DIFF: return D == 1;
        case KABC_COST_NORM_SHELL: return D >= 1;
        case KABC_COST_NOISY_QUAD_DU: return D == 2;
        case KABC_COST_MIXTURE: return D == 1;
        case KABC_COST_NOISY_BANANA: return D == 2;
        case KABC_COST_WIENER_RMS: return D == 2;
#ifdef KABC_USER_DIM_OK
        case KABC_COST_USER: return KABC_USER_DIM_OK(D);
#endif
        default: return false;
    }
}

// device-side counters shared by a handle
struct DevCounters {
    unsigned long long proposals;
    unsigned long long cost_evals;
    unsigned long long accepted;
    unsigned long long retries;   // init: total re-draws (src/KissABC.jl:57)
    int32_t error;                // 0 ok, 1 correction invalid, 2 starting sample invalid
    int32_t init_failed;          // retry budget exhausted
};

}  // namespace kabc
