// ais_kernels.hpp -- the AIS walker-update kernels (gfx950), templated on
// (D, DeviceCost id).  Instantiated per cost id in ais_inst_*.hip.
//
// Replaces, for a whole half-ensemble at once:
//   transition!            src/transition.jl:67-82
//   propose + three moves  src/transition.jl:2-65
//   push_p                 src/types.jl:27-32
//   loglike / accept       src/types.jl:51-75 (kernelized), :84-104 (threshold)
//   Factored logpdf        src/priors.jl:30-36
//   step(init)             src/KissABC.jl:35-64
#pragma once

#include "kabc_device.hpp"
#ifndef __HIPCC_RTC__
#include "launcher.hpp"
#endif

namespace kabc {

struct AisArgs {
    double* x_act;          // active half, GLOBAL rows [rows_act_total][D]
    const double* x_comp;   // complementary half, GLOBAL rows [n_comp][D] (frozen)
    double* lp;             // [rows_owned] logprior of the owned rows
    double* ll;             // [rows_owned] loglikelihood (kernelized) | cost (threshold)
    double* trace;          // optional [rows_owned][D]: push_p(x) after the last transition
    int32_t* dbg;           // optional [rows_owned][nt][6] per-transition records
    DevCounters* counters;
    unsigned long long* slots;  // [kCounterSlots][8] per-workgroup counter lines (no contention)
    const double* cost_params;
    const double* cost_data;
    int64_t cost_ndata;
    int64_t row_first;      // first owned row of the active half
    int64_t rows_owned;
    int64_t n_comp;         // rows of the complementary half
    uint64_t seed;
    uint64_t t0;            // transition counter of the first sub-step
    uint32_t id_base;       // global walker id of row 0 of the active half
    int32_t nt;             // ntransitions
    int32_t posterior;      // kabc_posterior_kind_t
    double eps;             // scale | maxcost
    double reps;            // RN(1/eps) for kabc_div_rc
    double box_lp;          // prior class BOX: the in-support log-density (ordered sum of c0)
    int32_t ablate;         // PROBES=1 builds only (KABC_ABLATE): 1 no consumer, 2 no producers, 64 HW_ID, 128 barrier time
    const PriorDev* prior;  // [D] prepared components, device memory (scalar-loaded)
    // batched independent chains (MCMCThreads, src/KissABC.jl:96-104,108): blockIdx.y = chain.
    // Chain c works on x_act + c * stride_act, x_comp + c * stride_comp, lp/ll + c * stride_own,
    // trace + c * stride_trace with the seed seeds[c]; walker ids are per chain, so every chain
    // is bit-identical to a single-chain run with its seed.  seeds == NULL: one chain.
    const uint64_t* seeds;
    int64_t stride_act, stride_comp, stride_own, stride_trace;
    // prepared cost words of every (sub-step, owned row) of this launch, computed by the grid-wide
    // pre-pass (ais_aux_kernels.hpp): [chain][nt][W][rows_owned]; NULL = the producers compute them
    const double* aux;
    int64_t stride_aux;
    // debug records of a launch that covers sub-steps [dbg_s0, dbg_s0 + nt) of dbg_nt
    int32_t dbg_nt, dbg_s0;
};

// leading normal-pair blocks of the cost's stream that the AIS producers expand (0: none).  Bounded
// by LDS: two buffers of kChunk sub-steps x 2 NPRE x 64 doubles beside the records.
constexpr int ais_pre_blocks(int cost, int D) {
    const int n = cost >= KABC_COST_USER ? 0 : cost_pre_blocks(cost, D);
    return n <= 8 ? n : 0;
}

constexpr int kLateFrom = 10;  // D above this: the consumer keeps ONE set of partner-row registers
constexpr int kCounterSlots = 1024;  // one 64-byte line per workgroup (mod 1024)

struct InitArgs {
    double* x_act;
    double* lp;
    double* ll;
    DevCounters* counters;
    const double* cost_params;
    const double* cost_data;
    int64_t cost_ndata;
    int64_t row_first;
    int64_t rows_owned;
    uint64_t seed;
    uint32_t id_base;
    int32_t posterior;
    int32_t cost_id;
    double eps;
    unsigned long long retry_budget;  // retry_sampling * nparticles (src/KissABC.jl:52)
    PriorSet prior;
    kabc_prior_t raw[KABC_MAX_DIM];
    // batched chains (blockIdx.y = chain), see AisArgs; the retry budget is per chain
    const uint64_t* seeds;
    unsigned long long* chain_retries;  // [nchains]
    int64_t stride_act, stride_own;
};

constexpr int kInitBlock = 64;

// ---- geometry of the half-generation kernel --------------------------------
// One workgroup = one BATCH of 64 walkers = 4 wavefronts with fixed roles:
//   wave 0      : CONSUMER.  Lane l owns walker l of the batch and runs the
//                 state-dependent chain proposal -> push_p -> prior logpdf ->
//                 cost -> accept for `ntransitions` consecutive sub-steps with
//                 x, logprior, loglik in registers.
//   waves 1..3  : PRODUCERS.  Everything a transition draws is a pure function
//                 of (seed, walker, t, slot) (include/kabc_philox.h), i.e. it does
//                 not depend on the walker's state.  Producer wave q prepares
//                 sub-step q of the NEXT chunk of kChunk sub-steps: move id,
//                 partner rows, log(u_accept), Z and (D-1)log Z, gamma, the
//                 normal variates -- as SoA records in LDS.
// The three moves need very different amounts of randomness (stretch 2 Philox
// blocks, DE 3 + (D+2)/2 Box-Muller blocks, walk 3 + 2).  A lane-per-walker
// kernel serialises all three under divergence in every wavefront; here the
// producers turn the variable part into dense work lists (wave ballot + mbcnt
// compaction in LDS) so that every Philox/Box-Muller instruction runs with
// (nearly) all 64 lanes doing useful work.  Records are double-buffered: the
// consumer reads chunk c while the producers fill chunk c+1; one workgroup
// barrier per chunk.
constexpr int kAisBlock = 256;
constexpr int kBatch = 64;   // walkers per workgroup
constexpr int kChunk = 3;    // sub-steps per record buffer = number of producer waves

template <int D>
struct RecGeom {
    static constexpr int NB = (D + 2) / 2;                 // normal blocks of a DE move
    // doubles per record: D + 1 used; one more so that the second normal of a DE lane's last
    // block (unused when D is even) has a slot and the Box-Muller phase stores both unconditionally
    static constexpr int NZ = (D + 2 > 4) ? (D + 2) : 4;
};

// SoA record buffer of one chunk, lane-contiguous (conflict-free ds_read_b64)
template <int Q>
struct SubStepIx {  // a sub-step's index within its chunk, as a type
    static constexpr int value = Q;
};

// (CH sub-steps per buffer: kChunk for the half-generation kernel, 1 for a ring slot of the
// one-workgroup kernel of ais_small_kernel.hpp)
template <int D, int CH>
struct RecBuf {
    uint32_t mva[CH][kBatch];   // (move << 30) | partner row a
    uint32_t bb[CH][kBatch];    // partner row b (DE, walk)
    uint32_t cc[CH][kBatch];    // partner row c (walk)
    double logu[CH][kBatch];    // log(u) = -randexp(rng)          (src/types.jl:74)
    // stretch: zs[0] = Z, zs[1] = (D-1) log Z     (src/transition.jl:56-58)
    // de     : zs[0] = gamma, zs[1..D] = randn per coordinate (:3, :13)
    // walk   : zs[0..2] = the three randn           (:38-40)
    double zs[CH][RecGeom<D>::NZ][kBatch];
};
template <int D>
using ChunkRec = RecBuf<D, kChunk>;

template <int POSTERIOR_RUNTIME = 0>
__device__ __forceinline__ bool ld_valid(int posterior, double lp, double ll) {
    // is_valid_logdensity: src/types.jl:60 and :93-94
    // (CommonLogDensity, :121: isfinite(ld), held as lp = 0, ll = lπ)
    return posterior != KABC_POSTERIOR_THRESHOLD ? kabc_isfinite(lp + ll)
                                                 : (kabc_isfinite(ll) && kabc_isfinite(lp));
}

// Prior classes (chosen on the host, identical results):
//   BOX     every component is Uniform / DiscreteUniform: logpdf is the constant
//           c0_1 + ... + c0_D (summed left to right on the host, as
//           src/priors.jl:30-36 would) inside the box and -Inf outside
//   SIMPLE  every component is constant-in-a-box (Uniform, DiscreteUniform) or Gaussian-in-a-box
//           (Normal, truncated Normal): one two-way wave-uniform branch per component, the
//           parameters of four components per batch of LDS reads (gaussbox_logpdf_push)
//   NORMAL  every component is a plain Normal: SIMPLE without the branch, the support tests and
//           the push_p rounding (test/runtests.jl:241)
//   GENERAL everything (Exponential, Beta, NegativeBinomial, Gamma, LogNormal ...)
enum { kPriorBox = 0, kPriorSimple = 1, kPriorGeneral = 2, kPriorNormal = 3 };
constexpr int kPriorClasses = 4;
// GENERAL class: NegativeBinomial components whose lgamma(k + r) is tabulated per launch in LDS
constexpr int kNbTabsGeneral = kNbLg1Block;

struct BoxPrior {
    const double* lo;   // LDS, [D]
    const double* hi;   // LDS, [D]
    uint32_t dmask;     // bit k: component k is discrete (push_p rounds)
    double lp;          // in-support log-density
};

// y inside the box [lo, hi]^D as ONE v_cmpx chain (D = 2..8, the sizes whose bounds live in
// registers): every comparison narrows EXEC itself, so the 2D s_and_b64 that combine ordinary
// v_cmp results disappear -- SALU instructions cost a wave an issue slot each, and these sat on
// a dependent VALU -> SALU -> VALU chain (C3: 155.4 -> 149.6 us per launch at ntransitions =
// 100).  Lanes still enabled at the end are inside the box; EXEC is restored before the
// statement ends.  NaN compares false, as `y >= lo && y <= hi` does.  The surviving EXEC is handed
// back as the boolean itself (inverse ballot): a flag register set under the narrowed EXEC and
// compared afterwards was three VALU instructions more.
#define KABC_BOXP(k) \
    "v_cmpx_ge_f64_e32 vcc, %[y" #k "], %[l" #k "]\n\tv_cmpx_le_f64_e32 vcc, %[y" #k "], %[h" #k "]\n\t"
#define KABC_BOXO(k) [y##k] "v"(y[k]), [l##k] "v"(lo[k]), [h##k] "v"(hi[k])
#if __has_builtin(__builtin_amdgcn_inverse_ballot_w64)
#define KABC_BOX_ASM(BODY, ...)                                                                   \
    unsigned long long saved, inside;                                                             \
    asm volatile("s_mov_b64 %[sv], exec\n\t" BODY "s_mov_b64 %[in], exec\n\ts_mov_b64 exec, %[sv]" \
                 : [in] "=&s"(inside), [sv] "=&s"(saved)                                          \
                 : __VA_ARGS__                                                                    \
                 : "vcc");                                                                        \
    return __builtin_amdgcn_inverse_ballot_w64(inside);
#else  // (a hipRTC older than the one hipcc ships with -- the copy bundled with PyTorch, say)
#define KABC_BOX_ASM(BODY, ...)                                                                   \
    unsigned flag = 0;                                                                            \
    unsigned long long saved;                                                                     \
    asm volatile("s_mov_b64 %[sv], exec\n\t" BODY "v_mov_b32 %[fl], 1\n\ts_mov_b64 exec, %[sv]"  \
                 : [fl] "+v"(flag), [sv] "=&s"(saved)                                             \
                 : __VA_ARGS__                                                                    \
                 : "vcc");                                                                        \
    return flag != 0u;
#endif
template <int D>
__device__ __forceinline__ bool box_contains(const double* y, const double* lo, const double* hi);
template <>
__device__ __forceinline__ bool box_contains<2>(const double* y, const double* lo, const double* hi) {
    KABC_BOX_ASM(KABC_BOXP(0) KABC_BOXP(1), KABC_BOXO(0), KABC_BOXO(1))
}
template <>
__device__ __forceinline__ bool box_contains<3>(const double* y, const double* lo, const double* hi) {
    KABC_BOX_ASM(KABC_BOXP(0) KABC_BOXP(1) KABC_BOXP(2), KABC_BOXO(0), KABC_BOXO(1), KABC_BOXO(2))
}
template <>
__device__ __forceinline__ bool box_contains<4>(const double* y, const double* lo, const double* hi) {
    KABC_BOX_ASM(KABC_BOXP(0) KABC_BOXP(1) KABC_BOXP(2) KABC_BOXP(3), KABC_BOXO(0), KABC_BOXO(1), KABC_BOXO(2),
                 KABC_BOXO(3))
}
template <>
__device__ __forceinline__ bool box_contains<5>(const double* y, const double* lo, const double* hi) {
    KABC_BOX_ASM(KABC_BOXP(0) KABC_BOXP(1) KABC_BOXP(2) KABC_BOXP(3) KABC_BOXP(4), KABC_BOXO(0), KABC_BOXO(1),
                 KABC_BOXO(2), KABC_BOXO(3), KABC_BOXO(4))
}
template <>
__device__ __forceinline__ bool box_contains<6>(const double* y, const double* lo, const double* hi) {
    KABC_BOX_ASM(KABC_BOXP(0) KABC_BOXP(1) KABC_BOXP(2) KABC_BOXP(3) KABC_BOXP(4) KABC_BOXP(5), KABC_BOXO(0),
                 KABC_BOXO(1), KABC_BOXO(2), KABC_BOXO(3), KABC_BOXO(4), KABC_BOXO(5))
}
template <>
__device__ __forceinline__ bool box_contains<7>(const double* y, const double* lo, const double* hi) {
    KABC_BOX_ASM(KABC_BOXP(0) KABC_BOXP(1) KABC_BOXP(2) KABC_BOXP(3) KABC_BOXP(4) KABC_BOXP(5) KABC_BOXP(6),
                 KABC_BOXO(0), KABC_BOXO(1), KABC_BOXO(2), KABC_BOXO(3), KABC_BOXO(4), KABC_BOXO(5), KABC_BOXO(6))
}
template <>
__device__ __forceinline__ bool box_contains<8>(const double* y, const double* lo, const double* hi) {
    KABC_BOX_ASM(KABC_BOXP(0) KABC_BOXP(1) KABC_BOXP(2) KABC_BOXP(3) KABC_BOXP(4) KABC_BOXP(5) KABC_BOXP(6)
                     KABC_BOXP(7),
                 KABC_BOXO(0), KABC_BOXO(1), KABC_BOXO(2), KABC_BOXO(3), KABC_BOXO(4), KABC_BOXO(5), KABC_BOXO(6),
                 KABC_BOXO(7))
}
#undef KABC_BOX_ASM
#undef KABC_BOXO
#undef KABC_BOXP

// loglike(density, push_p(density, y)) -- src/types.jl:51-58, :84-91
template <int D, int COST, int PC>
__device__ __forceinline__ void loglike(const PriorDev* __restrict__ P, const BoxPrior& B,
                                        const GaussBoxPrior& GB,
                                        int posterior, double eps, double reps, const double* y,
                                        const double* cost_params, const double* cost_data,
                                        int64_t ndata, kabc_cost_rng_t* rng, double& lp,
                                        double& ll, bool& ev, const double* logtab = kabc_log_tab,
                                        const double* nbtab = nullptr) {
    double yp[D];
    // A cheap deterministic cost is evaluated for every lane and selected afterwards: the
    // divergent region around it (exec save / branch / restore) costs the consumer wave more
    // issue slots than the arithmetic it would skip, and in practice some lane always needs it.
    constexpr bool kCheap = COST == KABC_COST_GAUSS_DIST || COST == KABC_COST_ROSENBROCK ||
                            COST == KABC_COST_DIRAC_SQ || COST == KABC_COST_ABS_DIFF ||
                            COST == KABC_COST_NORM_SHELL;
    if (posterior == KABC_POSTERIOR_COMMON) {
        // loglike(density::CommonLogDensity, sample) = lπ(sample.x)  src/types.jl:117-119;
        // push_p is the identity for a plain AbstractDensity (:27)
        lp = 0.0;
        ev = true;
        ll = eval_cost<COST, D>(y, cost_params, cost_data, ndata, rng);
        return;
    }
    if constexpr (PC == kPriorBox) {
        double lo[D], hi[D];
#pragma unroll
        for (int k = 0; k < D; ++k) {
            lo[k] = B.lo[k];
            hi[k] = B.hi[k];
        }
        bool in = true;
        if (__builtin_amdgcn_readfirstlane((int)B.dmask) == 0) {  // wave-uniform (a scalar branch): all components continuous, push_p = identity
            if constexpr (D >= 2 && D <= 8) in = box_contains<D>(y, lo, hi);
            else {
#pragma unroll
                for (int k = 0; k < D; ++k) in = in && (y[k] >= lo[k]) && (y[k] <= hi[k]);
            }
            if constexpr (kCheap) {
                // finish here, on y itself: merging this path with the rounding one through yp
                // costs D register-pair copies per transition
                // (outside the box the reference's values are lp = -Inf, ll = -/+Inf: an invalid
                // log-density that accept() rejects.  Here the caller gates on `ev` instead and
                // lp / ll are left as computed -- no selects on the consumer's issue stream)
                lp = B.lp;
                ev = in;
                const double c = eval_cost<COST, D>(y, cost_params, cost_data, ndata, rng);
                if (posterior == KABC_POSTERIOR_KERNELIZED) {
                    const double q = kabc_div_rc(c, eps, reps);
                    ll = -0.5 * (q * q);
                } else {
                    ll = c;
                }
                return;
            }
#pragma unroll
            for (int k = 0; k < D; ++k) yp[k] = y[k];
        } else {
#pragma unroll
            for (int k = 0; k < D; ++k) {
                const double v = ((B.dmask >> k) & 1u) ? kabc_rint(y[k]) : y[k];
                yp[k] = v;
                in = in && (v >= lo[k]) && (v <= hi[k]);
            }
            if constexpr (kCheap) {
                // finished here too: the two paths then meet on (lp, ll, ev) only, not on yp
                lp = B.lp;
                ev = in;
                const double c = eval_cost<COST, D>(yp, cost_params, cost_data, ndata, rng);
                if (posterior == KABC_POSTERIOR_KERNELIZED) {
                    const double q = kabc_div_rc(c, eps, reps);
                    ll = -0.5 * (q * q);
                } else {
                    ll = c;
                }
                return;
            }
        }
        lp = in ? B.lp : -KABC_INF;
    } else if constexpr (PC == kPriorSimple || PC == kPriorNormal) {
        bool in;
        double sum;
        if constexpr (PC == kPriorNormal) sum = gaussbox_logpdf_push<D, true, false>(GB, y, yp, in);
        else if (__builtin_amdgcn_readfirstlane((int)GB.dmask) == 0) sum = gaussbox_logpdf_push<D, false, false>(GB, y, yp, in);  // wave-uniform
        else sum = gaussbox_logpdf_push<D, false, true>(GB, y, yp, in);
        lp = in ? sum : -KABC_INF;
    } else {
#ifdef KABC_MODEL_SPEC
        // a translation unit generated for one model (kabc_compile_model): families and
        // parameters are compile-time constants (kabc_device.hpp model_logpdf_push)
        if constexpr (D == kabc_mspec::D) lp = model_logpdf_push<D, kNbTabsGeneral>(y, yp, logtab, nbtab);
        else
#endif
        lp = factored_logpdf_push<D, false>(P, y, yp, logtab, nbtab);
    }
    ev = kabc_isfinite(lp);
    if (posterior == KABC_POSTERIOR_KERNELIZED) {
        if constexpr (kCheap) {
            const double c = eval_cost<COST, D>(yp, cost_params, cost_data, ndata, rng);
            const double q = kabc_div_rc(c, eps, reps);
            ll = ev ? -0.5 * (q * q) : lp;
        } else {
            ll = lp;
            if (ev) {
                const double c = eval_cost<COST, D>(yp, cost_params, cost_data, ndata, rng);
                const double q = kabc_div_rc(c, eps, reps);
                ll = -0.5 * (q * q);
            }
        }
    } else {
        if constexpr (kCheap) {
            const double c = eval_cost<COST, D>(yp, cost_params, cost_data, ndata, rng);
            ll = ev ? c : -lp;
        } else {
            ll = -lp;
            if (ev) ll = eval_cost<COST, D>(yp, cost_params, cost_data, ndata, rng);
        }
    }
}

__device__ __forceinline__ void wave_lds_fence() {
    // LDS traffic of one wavefront completes in order; this only stops the
    // compiler from moving LDS accesses across the point.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// PRODUCER: fill sub-step `si` (transition counter t) of a record buffer for the
// 64 walkers of the batch.  Runs in ONE wavefront; lists are wave-private.
// Timing probes (KABC_ABLATE, tools/ablate_probe.py / barrier_probe.py / placement_probe.py)
// exist only in the PROBES=1 build of the library (libkabc_hip_probes.so): the run-time
// tests on A.ablate cost 3.5 % of a launch when compiled in.
#ifdef KABC_PROBES
#define KABL A.ablate
#else
#define KABL 0
#endif
struct NoMid {
    __device__ __forceinline__ void operator()() const {}
};
// `mid()` runs as soon as the partner rows of the sub-step (mva, bb, cc) are in the record: the
// prologue passes a workgroup barrier there, behind which the consumer requests its first rows.
// WALK = false (ais_small_kernel.hpp: the producers run ahead of the half-steps, so the partner rows a
// walk displacement is made of are not final yet): the three normals stay in zs[0..2], listB
// [nDE, nDE + nWK) names the walk lanes and counts = {nDE, nWK}; the consumer finishes the displacement.
// Args: anything with n_comp, x_comp (WALK only) and ablate.
template <int D, class Mid = NoMid, bool WALK = true, class Rec = ChunkRec<D>, class Args = AisArgs>
__device__ __forceinline__ void produce_substep(const Args& A, const uint64_t seed,
                                                Rec& R, int si,
                                                uint64_t t, uint32_t w_base, int n_active,
                                                uint8_t* listB, int lane,
                                                const double* logtab,
                                                const kabc_u128_t* pre01 = nullptr, Mid mid = Mid(),
                                                int32_t* counts = nullptr) {
    constexpr int NB = RecGeom<D>::NB;
    const uint64_t nc = (uint64_t)A.n_comp;
    const bool active = lane < n_active;
    int move = 0;
    uint32_t a = 0;
    // -- phase A: one (walker, t) per lane: move id, partner a, log u, stretch factor
    // Straight-line for all 64 lanes: the lanes past a ragged batch's end draw too (their move
    // is forced to 0, so they join no list and nobody reads their record), and the stretch
    // factor is formed and stored whatever the move -- a DE / walk lane's zs[0..1] are
    // overwritten by its normals in the Box-Muller phase below.
    {
        const uint32_t w = w_base + (uint32_t)lane;
        // (the prologue hands over the two blocks it expanded while the table was in flight)
        const kabc_u128_t B0 = pre01 ? pre01[0] : kabc_stream_block(seed, w, t, 0u, KABC_DOM_AIS_MOVE);
        const kabc_u128_t B1 = pre01 ? pre01[1] : kabc_stream_block(seed, w, t, 1u, KABC_DOM_AIS_MOVE);
        const uint32_t m7 = (uint32_t)(((uint64_t)B0.w[2] * 7u) >> 32);  // rand((1,1,1,1,2,2,3))
        move = !active ? 0 : (m7 < 4u) ? 1 : (m7 < 6u) ? 2 : 3;
        a = active ? kabc_index32(kabc_lo64(B0), (uint32_t)nc) : 0u;
        R.logu[si][lane] = kabc_log_pn_tab(kabc_u01(kabc_lo64(B1)), logtab);
        // Z = cdf_g_inv(rand(rng), 3.0); correction (D-1) log Z
        const double sq3 = kabc_sqrt(3.0), isq3 = kabc_sqrt(1.0 / 3.0);
        const double u = kabc_u01(kabc_hi64(B1));
        const double tz = u * (sq3 - isq3) + isq3;
        const double Z = tz * tz;
        R.zs[si][0][lane] = Z;
        R.zs[si][1][lane] = (double)(D - 1) * kabc_log_pn_tab(Z, logtab);
        // partner rows b (DE, walk) and c (walk): drawn for every lane, kept by the 3/7 that use
        // them -- 27 of 64 lanes is one pass either way, and the list indirection goes
        const kabc_u128_t B2 = kabc_stream_block(seed, w, t, 2u, KABC_DOM_AIS_MOVE);
        // (32-bit: row numbers fit 30 bits, mva packs them so -- the same values as the
        // contract's 64-bit formulation at a third of the instructions)
        uint32_t b = kabc_index32(kabc_lo64(B2), (uint32_t)nc - 1u);
        b += (b >= a) ? 1u : 0u;
        const uint32_t lo = a < b ? a : b, hi = a < b ? b : a;
        uint32_t c = kabc_index32(kabc_hi64(B2), (uint32_t)nc - 2u);
        c += (c >= lo) ? 1u : 0u;
        c += (c >= hi) ? 1u : 0u;
        R.bb[si][lane] = move >= 2 ? b : a;  // (a: valid row for the consumer's unconditional prefetch)
        R.cc[si][lane] = move >= 2 ? c : a;
    }
    R.mva[si][lane] = ((uint32_t)move << 30) | a;
    mid();
    // -- compaction of the move-dependent extra work (wave ballot + mbcnt)
    const unsigned long long mDE = __ballot(move == 2), mWK = __ballot(move == 3);
    const unsigned long long mB = mDE | mWK;
    const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    // one lane list: DE lanes first, then walk lanes.  Work item e of the Box-Muller
    // phase is decoded from it: e < nDE*NB -> (DE lane e / NB, block e % NB), else
    // (walk lane (e - nDE*NB) / 2, block (e - nDE*NB) % 2).
    const int nDE = __popcll(mDE), nB = __popcll(mB);
    const int nN = NB * nDE + 2 * (nB - nDE);
    if (move == 2) listB[__popcll(mDE & below)] = (uint8_t)lane;
    if (move == 3) listB[nDE + __popcll(mWK & below)] = (uint8_t)lane;
    wave_lds_fence();
    // -- walk work, dense: the state-independent displacement W of a walk lane is D independent
    // coordinates, so it is spread over the wave as (walk lane, coordinate) items -- 64 / D walk
    // lanes per pass -- instead of D coordinates in each of the ~9 walk lanes of 64 (one lane in
    // seven busy).  The three partner values of the first two passes (all there is in 99 % of
    // the sub-steps at D = 8) are fetched now and fly under the Box-Muller phase.
    constexpr int kWPP = kWave / D;  // walk lanes per pass
    const int nWK = nB - nDE;
    const int wq = lane / D, wk = lane - wq * D;
    auto walk_fetch = [&](int pass, int& l, double& va, double& vb, double& vc) {
        const int q = pass * kWPP + wq;
        l = -1;
        if (wq < kWPP && q < nWK) {
            l = listB[nDE + q];
            // (32-bit byte offsets from the scalar base, like the consumer's row loads)
            const char* cb = reinterpret_cast<const char*>(A.x_comp);
            const uint32_t ko = (uint32_t)wk * 8u;
            va = *reinterpret_cast<const double*>(cb + (size_t)((R.mva[si][l] & 0x3fffffffu) * (uint32_t)(D * 8) + ko));
            vb = *reinterpret_cast<const double*>(cb + (size_t)(R.bb[si][l] * (uint32_t)(D * 8) + ko));
            vc = *reinterpret_cast<const double*>(cb + (size_t)(R.cc[si][l] * (uint32_t)(D * 8) + ko));
        }
    };
    int wl0 = -1, wl1 = -1;
    double wa0, wb0, wc0, wa1, wb1, wc1;  // (defined exactly where wl0 / wl1 >= 0)
    if constexpr (WALK) {
        walk_fetch(0, wl0, wa0, wb0, wc0);
        walk_fetch(1, wl1, wa1, wb1, wc1);
    } else {
        if (counts && lane == 0) {
            counts[0] = nDE;
            counts[1] = nWK;
        }
    }
    // -- phase N: Box-Muller blocks, dense
#pragma unroll 1
    for (int e = lane; e < ((KABL & 8) ? 0 : nN); e += kWave) {
        // (unsigned arithmetic, both decodings evaluated and selected: no masked regions)
        const unsigned ue = (unsigned)e, ude = (unsigned)(nDE * NB);
        const bool is_de = ue < ude;
        const unsigned qd = ue / (unsigned)NB, e2 = ue - ude;
        const int q = (int)(is_de ? qd : (unsigned)nDE + (e2 >> 1));
        const int j = (int)(is_de ? ue - qd * (unsigned)NB : (e2 & 1u));
        const int l = listB[q];
        const kabc_u128_t Bn = kabc_stream_block(seed, w_base + (uint32_t)l, t, 3u + (uint32_t)j,
                                                 KABC_DOM_AIS_MOVE);
        double z0, z1;
        kabc_normal_pair_tab(kabc_lo64(Bn), kabc_hi64(Bn), &z0, &z1, logtab);
        // both variates are stored: the one past the last used index (DE with even D: D + 1;
        // walk: 3) lands in a slot nobody reads, or that the walk displacement overwrites
        R.zs[si][2 * j][l] = z0;
        R.zs[si][2 * j + 1][l] = z1;
    }
    wave_lds_fence();
    // -- phase C: gamma = 2.38/sqrt(2D) * exp(0.1 randn)   (src/transition.jl:3)
    if (KABL & 16) return;
    if (move == 2) {
        const double z0 = R.zs[si][0][lane];
        // |z0| <= sqrt(-2 log 2^-53) = 8.6 (kabc_u01 is never 0): the exponent is within +-0.86
        R.zs[si][0][lane] = 2.38 / kabc_sqrt((double)(2 * D)) * kabc_exp_bounded(z0 * 0.1);
    }
    // ais_walk_propose (src/transition.jl:24-43): Xs = (a + (b + c)) / 3,
    // W = z1 (a - Xs) + z2 (b - Xs) + z3 (c - Xs); the consumer adds x_i.  W_k overwrites the
    // walk lane's normals (zs[0..2]): every item of a pass reads them before any item writes --
    // the items of a pass are the lanes of ONE wavefront executing the same instructions, each
    // store depends on its lane's loads, and the LDS executes a wavefront's instructions in order.
    auto walk_finish = [&](int l, double va, double vb, double vc) {
        if (l >= 0) {
            const double z0 = R.zs[si][0][l], z1 = R.zs[si][1][l], z2 = R.zs[si][2][l];
            const double Xs = kabc_div_rc(va + (vb + vc), 3.0, 1.0 / 3.0);
            R.zs[si][wk][l] = z0 * (va - Xs) + z1 * (vb - Xs) + z2 * (vc - Xs);
        }
    };
    if constexpr (WALK) {
        if (nWK > 0) walk_finish(wl0, wa0, wb0, wc0);
        if (nWK > kWPP) walk_finish(wl1, wa1, wb1, wc1);
#pragma unroll 1
        for (int pass = 2; pass * kWPP < nWK; ++pass) {
            int l;
            double va, vb, vc;
            walk_fetch(pass, l, va, vb, vc);
            walk_finish(l, va, vb, vc);
        }
    }
}

// The leading normal pairs of a stochastic cost's stream (cost_pre_blocks), expanded by a producer
// wave for the 64 walkers of the batch: block j of walker `lane` -> pre[2 j][lane], pre[2 j + 1][lane]
// (what kabc_cost_rng_normal2 would compute in place: include/kabc_philox.h)
template <int NPRE>
__device__ __forceinline__ void produce_cost_normals(uint64_t seed, uint64_t t, uint32_t w_base, int lane,
                                                     double (*pre)[kBatch], const double* logtab) {
#pragma unroll 1
    for (int j = 0; j < NPRE; ++j) {
        const kabc_u128_t b = kabc_stream_block(seed, w_base + (uint32_t)lane, t, (uint32_t)j, KABC_DOM_AIS_COST);
        double z0, z1;
        kabc_normal_pair_tab(kabc_lo64(b), kabc_hi64(b), &z0, &z1, logtab);
        pre[2 * j][lane] = z0;
        pre[2 * j + 1][lane] = z1;
    }
}

// producer side of a prepared cost: lane = walker of the batch, the same sequential
// arithmetic kabc_cost_eval would do in place (include/kabc_costs.h)
template <int COST, int W>
__device__ __forceinline__ void prepare_cost_aux(const AisArgs& A, uint64_t t, uint32_t w_base,
                                                 int lane, double (*aux)[kBatch],
                                                 const double* logtab, int64_t r0, int n_active) {
    if (A.aux) {  // wave-uniform: the pre-pass has them, lane = walker copies W words
        if (lane < n_active) {
            const int64_t s = (int64_t)(t - A.t0);
#pragma unroll
            for (int j = 0; j < W; ++j) aux[j][lane] = A.aux[(s * W + j) * A.rows_owned + r0 + lane];
        }
        return;
    }
    kabc_cost_rng_t rng = {A.seed, t, w_base + (uint32_t)lane, KABC_DOM_AIS_COST, 0u, 0u, nullptr,
                           logtab};
    double a[W];
    kabc_cost_prepare(COST, A.cost_params, A.cost_data, A.cost_ndata, &rng, a);
#pragma unroll
    for (int j = 0; j < W; ++j) aux[j][lane] = a[j];
}

// PK = posterior kind (KABC_POSTERIOR_*) as a compile-time constant: with the kind read
// from the arguments the consumer carried three run-time branches per sub-step in
// loglike/accept and the SGPRs to feed them -- 8 % of the launch.
template <int D, int COST, int PC, int PK>
__global__ void __launch_bounds__(kAisBlock) __attribute__((amdgpu_waves_per_eu(2, 2)))
ais_half_kernel(const AisArgs A0) {
    AisArgs A = A0;
    if (A0.seeds) {  // wave-uniform: one chain of a batch per blockIdx.y
        const int64_t c = (int64_t)blockIdx.y;
        A.seed = A0.seeds[c];
        A.x_act += c * A0.stride_act;
        A.x_comp += c * A0.stride_comp;
        A.lp += c * A0.stride_own;
        A.ll += c * A0.stride_own;
        if (A0.trace) A.trace += c * A0.stride_trace;
        if (A0.aux) A.aux += c * A0.stride_aux;
    }
    __shared__ ChunkRec<D> rec[2];
    __shared__ uint8_t listB[kChunk][kBatch];

    // prepared prior components: read by the consumer with wave-uniform LDS
    // addresses (broadcast).  By-value kernel arguments made hipcc pin ~250 SGPRs
    // and spill them to VGPR lanes (385 v_readlane per transition).
    __shared__ PriorDev sprior[D];
    __shared__ double sbox_lo[D], sbox_hi[D];
    // SIMPLE / NORMAL classes: the components' records (GaussBoxPrior)
    constexpr bool kGaussBox = PC == kPriorSimple || PC == kPriorNormal;
    __shared__ __attribute__((aligned(16))) double sgb[kGaussBox ? D : 1][8];
    // the producers' copy of the log table (include/kabc_math.h): per-lane lookups
    __shared__ __attribute__((aligned(16))) double slogtab[KABC_MATH_TAB_WORDS];
    // GENERAL class: lgamma(k + r), k < kNbEntries, of the first kNbTabs NegativeBinomial
    // components -- kabc_lgamma_t's own values, computed once per launch by the whole workgroup
    // instead of once per transition by the consumer (kabc_device.hpp, the family's case)
    constexpr int kNbTabs = (PC == kPriorGeneral) ? kNbTabsGeneral : 0;
    __shared__ double snb[kNbTabs > 0 ? (kNbTabs + 1) * kNbEntries : 1];  // (+ the lgamma(k + 1) block)
    // prepared costs (include/kabc_costs.h): the parameter-independent part of the cost
    // of every sub-step, computed by the producers; word j of lane l at [buf][si][j][l]
    constexpr int kAuxW = cost_aux_c(COST);
    __shared__ double saux[2][kChunk][kAuxW > 0 ? kAuxW : 1][kBatch];
    // stochastic costs whose stream STARTS with normal pairs whatever the parameters
    // (cost_pre_blocks: the hierarchical simulator's D - 2 group noises, ...): the producers
    // expand those blocks for the sub-steps ahead -- all 64 lanes busy, three waves sharing the
    // work -- and the consumer, the serial part of the chain, reads the variates from LDS
    // instead of running Philox + Box-Muller itself (C4's prior + simulator under AIS: the
    // consumer did 7 blocks per transition, more than all the rest of its sub-step).  The draws
    // are counter-based, so expanding them early (or for a proposal the prior then rejects)
    // changes nothing.
    constexpr int kPre = ais_pre_blocks(COST, D);
    __shared__ double spre[2][kChunk][kPre > 0 ? 2 * kPre : 1][kBatch];

    // (wave index as a scalar: role branches and record addresses are wave-uniform)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & (kWave - 1);
    const int64_t r0 = (int64_t)blockIdx.x * kBatch;
    const int64_t rem = A.rows_owned - r0;
    const int n_active = rem >= kBatch ? kBatch : (int)rem;
    const uint32_t w_base = A.id_base + (uint32_t)(A.row_first + r0);
    const int nchunks = (A.nt + kChunk - 1) / kChunk;
    // KABC_ABLATE=128 (with debug records on): cycles each wave spends at the barriers
    const bool tprobe = (KABL & 128) && A.dbg;
    unsigned long long t_begin = 0, t_bar = 0;
    if (tprobe) t_begin = __builtin_amdgcn_s_memtime();
#define KABC_TIMED_BARRIER()                                              \
    do {                                                                  \
        if (tprobe) {                                                     \
            const unsigned long long tb_ = __builtin_amdgcn_s_memtime(); \
            __syncthreads();                                              \
            t_bar += __builtin_amdgcn_s_memtime() - tb_;                  \
        } else {                                                          \
            __syncthreads();                                              \
        }                                                                 \
    } while (0)

    // a row of the complementary half by a 32-bit byte offset from the (scalar) base: one shift
    // per row instead of a 64-bit shift and add (the host guarantees n_comp * D * 8 < 2^32)
    auto comp_row = [&](uint32_t idx) -> const double* {
        return reinterpret_cast<const double*>(reinterpret_cast<const char*>(A.x_comp) +
                                               (size_t)(idx * (uint32_t)(D * sizeof(double))));
    };
    // consumer state (wave 0)
    const bool active = (wave == 0) && (lane < n_active);
    const int64_t r = r0 + lane;
    const int64_t row = A.row_first + r;
    double x[D];
    double lp = 0.0, ll = 0.0;
    unsigned int n_eval = 0, n_acc = 0;
    int err = 0;
    if (active) {
        load_row<D>(A.x_act + row * D, x);
        lp = A.lp[r];
        ll = A.ll[r];
        if (!ld_valid(PK, lp, ll)) err = 2;  // accept(): "old log-density is invalid"
    }

    // the table's loads are issued here and land in LDS right before the barrier below
    static_assert(KABC_MATH_TAB_WORDS == 2 * kAisBlock, "two table words per thread");
    const double tab0 = kabc_log_tab[threadIdx.x], tab1 = kabc_log_tab[threadIdx.x + kAisBlock];
    if (threadIdx.x < D * (int)(sizeof(PriorDev) / 8))
        reinterpret_cast<double*>(sprior)[threadIdx.x] =
            reinterpret_cast<const double*>(A.prior)[threadIdx.x];
    if (threadIdx.x < D) {
        sbox_lo[threadIdx.x] = A.prior[threadIdx.x].p[0];
        sbox_hi[threadIdx.x] = A.prior[threadIdx.x].p[1];
    }
    uint32_t dmask = 0, gmask = 0;
    for (int k = 0; k < D; ++k) {
        dmask |= (A.prior[k].discrete ? 1u : 0u) << k;
        gmask |= (gaussbox_is_gauss(A.prior[k].kind) ? 1u : 0u) << k;
    }
    if constexpr (kGaussBox) {
        if (threadIdx.x < D) gaussbox_stage(sgb[threadIdx.x], A.prior[threadIdx.x]);
    }
    const GaussBoxPrior gbox = {sgb, gmask, dmask};
    // BOX class, D <= 8: the bounds live in registers for the whole launch (32 VGPRs at
    // D = 8) instead of being re-read from LDS in every sub-step (-1.5 %); larger D has
    // no registers to spare
    constexpr bool kBoxRegs = (PC == kPriorBox) && D <= 8;
    double blo[kBoxRegs ? D : 1], bhi[kBoxRegs ? D : 1];
    const BoxPrior box = {kBoxRegs ? blo : sbox_lo, kBoxRegs ? bhi : sbox_hi, dmask, A.box_lp};

    // while the table's loads are in flight the producers expand the two Philox blocks
    // every lane of their first sub-step needs (they depend on nothing else)
    kabc_u128_t pro01[2] = {};
    if (wave > 0 && wave - 1 < A.nt && lane < n_active) {
        const uint32_t w = w_base + (uint32_t)lane;
        const uint64_t t = A.t0 + (uint64_t)(wave - 1);
        pro01[0] = kabc_stream_block(A.seed, w, t, 0u, KABC_DOM_AIS_MOVE);
        pro01[1] = kabc_stream_block(A.seed, w, t, 1u, KABC_DOM_AIS_MOVE);
    }
    // the log table is staged by all four waves and read by the producers right away
    slogtab[threadIdx.x] = tab0;
    slogtab[threadIdx.x + kAisBlock] = tab1;
    // NegativeBinomial components (GENERAL class, launches long enough to pay for it): slot j of
    // snb = lgamma(k + r_j); which components have a slot is recomputed below from the same scan
#ifdef KABC_MODEL_SPEC
    // (a unit generated for one model addresses the table slots as compile-time constants
    // (kabc_device.hpp model_comp): its tables are always there)
    [[maybe_unused]] const bool nb_on = kNbTabs > 0;
#else
    [[maybe_unused]] const bool nb_on = kNbTabs > 0 && A.nt >= 8;
#endif
    if constexpr (kNbTabs > 0) {
        static_assert(kNbEntries == kAisBlock, "one table entry per thread");
        int slot = 0;
        bool has_nb = false;
        for (int k = 0; k < D; ++k) {  // (wave-uniform)
            if (A.prior[k].kind == KABC_PRIOR_NEGBINOMIAL) {
                has_nb = true;
                if (nb_on && slot < kNbTabs) {
                    snb[slot * kNbEntries + threadIdx.x] =
                        kabc_lgamma_t((double)threadIdx.x + A.prior[k].p[0], kabc_log_tab);
                    ++slot;
                }
            }
        }
        // the lgamma(k + 1) block: always there for a prior with such a component (one load per thread)
        if (has_nb) snb[kNbLg1Block * kNbEntries + threadIdx.x] = kabc_lgamma1_tab[threadIdx.x];
    }
    KABC_TIMED_BARRIER();
    if constexpr (kNbTabs > 0) {
        // p[2] of the staged NegativeBinomial components (a field the family does not use):
        // the table slot, or -1.  After the barrier: the staging copy above is complete; before
        // the next one: the consumer has not read the components yet.
        if (threadIdx.x < D && sprior[threadIdx.x].kind == KABC_PRIOR_NEGBINOMIAL) {
            int slot = 0;
            for (int k = 0; k < (int)threadIdx.x; ++k)
                slot += (sprior[k].kind == KABC_PRIOR_NEGBINOMIAL) ? 1 : 0;
            sprior[threadIdx.x].p[2] = (nb_on && slot < kNbTabs) ? (double)slot : -1.0;
        }
    }
    if constexpr (kBoxRegs) {  // per-lane copies from LDS (vector registers: scalar ones ran
                               // out and spilled when these were loaded as uniform values)
#pragma unroll
        for (int k = 0; k < D; ++k) {
            blo[k] = sbox_lo[k];
            bhi[k] = sbox_hi[k];
        }
    }

    // The stream seed as a VECTOR-register value for the producers: the ten Philox round keys
    // derived from it are loop-invariant either way, but as scalars they took twenty of the scalar
    // registers this kernel is short of (reloaded from spill lanes in every sub-step).
    uint64_t seed_v = A.seed;
    asm volatile("" : "+v"(seed_v));
    // The consumer's partner rows: (r0a, r0b) serve a chunk's first sub-step, (r1a, r1b) are the
    // second register set of the sub-step pipeline (see the consumer).  The rows of
    // chunk 0's first sub-step are in flight already when the chunk loop starts -- requested in the prologue as soon as the
    // producers have drawn the partners (a third of the way through their fill: the request's L2
    // round trip hides under the rest of it; 8.03 -> 7.79 us per launch at ntransitions = 1, nothing
    // at 16 and 100: profiles/r05_ab_prologue_prefetch.txt).  The same across every chunk boundary
    // (the previous chunk's last sub-step requesting the next chunk's first rows once wave 1 has
    // drawn them) was measured too and COSTS 1.6 % at 100: in steady state the SIMD's other waves
    // fill that round trip, and the flag test + register-set copy are consumer instructions.
    // The rows come from the frozen half: WHEN they are read cannot matter.
    constexpr bool kLate = D > kLateFrom;
    double r0a[D], r0b[D], r1a[kLate ? 1 : D], r1b[kLate ? 1 : D];
    // (wave-uniform and, outside the probes build, a compile-time constant)
    const bool pro_rows = !(KABL & 5);
    // prologue: producers fill chunk 0
    if (wave > 0) {
        const int si = wave - 1;
        if (si < A.nt && !(KABL & 4)) {
            produce_substep<D>(A, seed_v, rec[0], si, A.t0 + (uint64_t)si, w_base, n_active, listB[si],
                               lane, slogtab, pro01, [] { __syncthreads(); });
            if constexpr (kAuxW > 0)
                prepare_cost_aux<COST, kAuxW>(A, A.t0 + (uint64_t)si, w_base, lane, saux[0][si],
                                              slogtab, r0, n_active);
            if constexpr (kPre > 0)
                produce_cost_normals<kPre>(seed_v, A.t0 + (uint64_t)si, w_base, lane, spre[0][si], slogtab);
        } else {
            __syncthreads();
        }
    } else {
        __syncthreads();  // (the partner rows of every sub-step of chunk 0 are in the record)
#ifndef KABC_NO_XCHUNK_PREFETCH  // (A/B builds: tools/ab_ais_variant.sh)
        if (active && pro_rows) {
            load_row<D>(comp_row(rec[0].mva[0][lane] & 0x3fffffffu), r0a);
            load_row<D>(comp_row(rec[0].bb[0][lane]), r0b);
        }
#endif
    }
    KABC_TIMED_BARRIER();

    // The two roles run SEPARATE chunk loops (same trip count, one barrier per trip): values a
    // role keeps across its loop -- Philox keys and polynomial constants for the producers, the
    // walker's state and the box for the consumer -- then never occupy the other role's
    // registers (one shared loop made both sets live through both bodies: scalar-register
    // spills and re-materialised constants in every sub-step).
    if (wave > 0) {
#pragma unroll 1
        for (int c = 0; c < nchunks; ++c) {
            const int s0 = c * kChunk;
            // PRODUCER: sub-step (s0 + kChunk + wave - 1) of the next chunk
            const int si = wave - 1;
            const int s = s0 + kChunk + si;
            if (s < A.nt && !(KABL & 2)) {
                produce_substep<D>(A, seed_v, rec[(c + 1) & 1], si, A.t0 + (uint64_t)s, w_base, n_active,
                                   listB[si], lane, slogtab);
                if constexpr (kAuxW > 0)
                    prepare_cost_aux<COST, kAuxW>(A, A.t0 + (uint64_t)s, w_base, lane,
                                                  saux[(c + 1) & 1][si], slogtab, r0, n_active);
                if constexpr (kPre > 0)
                    produce_cost_normals<kPre>(seed_v, A.t0 + (uint64_t)s, w_base, lane,
                                               spre[(c + 1) & 1][si], slogtab);
            }
            KABC_TIMED_BARRIER();
        }
    } else {
#ifndef KABC_NO_REG_PARAMS
        // the cost's leading parameters in registers for the whole launch (cost_reg_params)
        constexpr int kRP = cost_reg_params(COST, D);
        double cpar[kRP > 0 ? kRP : 1];
#pragma unroll
        for (int k = 0; k < kRP; ++k) {
            cpar[k] = A.cost_params[k];
            asm volatile("" : "+v"(cpar[k]));  // (a value, not a re-loadable address)
        }
        const double* const cparams = kRP > 0 ? cpar : A.cost_params;
#else
        const double* const cparams = A.cost_params;
#endif
        // wave-uniform conditions of the sub-step as SCALARS (evaluated on a lane value inside the
        // `active` region they become lane masks: a v_cndmask + v_cmp + s_andn2 each, per sub-step)
        const bool dbg_on = __builtin_amdgcn_readfirstlane(A.dbg != nullptr ? 1 : 0) != 0;
#pragma unroll 1
        for (int c = 0; c < nchunks; ++c) {
            const int s0 = c * kChunk;
            if (active && !(KABL & 1)) {
                // CONSUMER
                const ChunkRec<D>& R = rec[(KABL & 2) ? 0 : (c & 1)];
                const int ns = (A.nt - s0 < kChunk) ? (A.nt - s0) : kChunk;
                // partner rows: (pa, pb) serve this sub-step, (na, nb) are the next one's, in
                // flight while this one computes.  Both rows are fetched for every lane
                // whatever its move (bb defaults to a): no divergence around the loads.
                // The sub-step body is instantiated twice with the two register sets
                // swapped, so no row is ever copied.
#ifndef KABC_NO_XCHUNK_PREFETCH
                if (c > 0 || !pro_rows)  // (scalar; chunk 0's were requested in the prologue)
#endif
                {
                    load_row<D>(comp_row(R.mva[0][lane] & 0x3fffffffu), r0a);
                    load_row<D>(comp_row(R.bb[0][lane]), r0b);
                }
                // `si` is a compile-time constant (SubStepIx<q>): every LDS address of the
                // sub-step is then (scalar buffer base + lane) + an immediate offset, not a
                // handful of VALU shifts and adds on the consumer's issue stream per sub-step.
                auto substep = [&](auto SI, const double (&pa)[D], const double (&pb)[D],
                                   double (&na)[D], double (&nb)[D]) __attribute__((always_inline)) {
                    constexpr int si = decltype(SI)::value;
                    const uint64_t t = A.t0 + (uint64_t)(s0 + si);
                    // (1) every LDS word of this sub-step in one batch, plus the partner
                    //     ids of the next one
                    constexpr int sq = si + 1 < kChunk ? si + 1 : si;
                    const uint32_t mva = R.mva[si][lane];
                    uint32_t mvan = R.mva[sq][lane], bn = R.bb[sq][lane];
                    if (si + 1 >= ns) {  // (scalar) the chunk's last sub-step: no next one in it
                        mvan = mva;
                        bn = R.bb[si][lane];
                    }
                    const double logu = R.logu[si][lane];
                    double zs[D + 1];
#pragma unroll
                    for (int j = 0; j < D + 1; ++j) zs[j] = R.zs[si][j][lane];
                    // (2) prefetch.  D <= kLateFrom: into the second register set, right away.
                    //     Larger D (two more rows would not fit 256 VGPRs: D = 16 spilled 480 B
                    //     and ran 4.3x slower than D = 8): into the SAME registers, as soon as
                    //     the proposal below has consumed the current rows.
                    if constexpr (!kLate) {
                        load_row<D>(comp_row(mvan & 0x3fffffffu), na);
                        load_row<D>(comp_row(bn), nb);
                    }
                    // scheduling fences: without them hipcc interleaves the phases of a
                    // sub-step for ILP and needs > 300 VGPRs (spills, 1 wave per SIMD)
                    __builtin_amdgcn_sched_barrier(0);
                    const uint32_t move = mva >> 30;
                    double y[D];
                    double corr = 0.0;
                    // ais_walk_propose  src/transition.jl:24-43 -- W does not depend on x_i: the
                    // producer has already formed it.  Evaluated for every lane (8 adds; a lane
                    // with another move overwrites y below): two independent masked regions
                    // instead of a three-way nest, i.e. fewer EXEC manipulations on the wave's
                    // single issue stream.
#pragma unroll
                    for (int k = 0; k < D; ++k) y[k] = x[k] + zs[k];
                    // x - a serves the stretch move as it is and the DE move as |a - x| (a
                    // negation is exact): formed once, for every lane, outside both regions
                    double xa[D];
#pragma unroll
                    for (int k = 0; k < D; ++k) xa[k] = x[k] - pa[k];
                    if (move == 1u) {
                        // stretch_propose  src/transition.jl:51-59
                        const double Z = zs[0];
                        corr = zs[1];
#pragma unroll
                        for (int k = 0; k < D; ++k) {
                            const double W = xa[k] * Z;
                            y[k] = pa[k] + W;
                        }
                    }
                    if (move == 2u) {
                        // de_propose  src/transition.jl:2-22
                        const double gamma = zs[0];
#pragma unroll
                        for (int k = 0; k < D; ++k) {
                            const double Wk = (pa[k] - pb[k]) * gamma;
                            const double sk = kabc_fabs(pa[k] - pb[k]) + kabc_fabs(x[k] - pb[k]) +
                                              kabc_fabs(xa[k]);
                            const double Tk = kabc_div_rc(gamma * sk, 300.0, 1.0 / 300.0) * zs[1 + k];
                            y[k] = x[k] + Wk + Tk;
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (kLate) {  // pa/pb are dead now: na/nb alias them
                        load_row<D>(comp_row(mvan & 0x3fffffffu), na);
                        load_row<D>(comp_row(bn), nb);
                    }
                    // ld = loglike(density, push_p(density, p))   src/transition.jl:75
                    kabc_cost_rng_t rng = {A.seed, t, w_base + (uint32_t)lane, KABC_DOM_AIS_COST, 0u,
                                           0u, nullptr, slogtab};
                    if constexpr (kAuxW > 0) {
                        rng.aux = &saux[(KABL & 2) ? 0 : (c & 1)][si][0][lane];
                        rng.aux_stride = kBatch;
                    }
                    if constexpr (kPre > 0) {
                        rng.pre = &spre[(KABL & 2) ? 0 : (c & 1)][si][0][lane];
                        rng.pre_n = (uint32_t)kPre;
                        rng.pre_stride = (uint32_t)kBatch;
                    }
                    double nlp, nll;
                    bool ev;
                    loglike<D, COST, PC>(sprior, box, gbox, PK, A.eps, A.reps, y, cparams,
                                         A.cost_data, A.cost_ndata, &rng, nlp, nll, ev, slogtab,
                                         kNbTabs > 0 ? snb : nullptr);
                    __builtin_amdgcn_sched_barrier(0);
                    n_eval += ev ? 1u : 0u;
                    // accept(...)  src/types.jl:62-75, :96-104
                    // (the old state's validity is checked once, before the first sub-step: it
                    // can only change through an accept, which requires a valid new state)
                    // straight-line: every comparison is evaluated, the flags are combined (a NaN /
                    // Inf log-density makes `valid` false whatever the comparisons say)
                    // `isfinite(ld_correction) || error(...)` (src/types.jl:69) cannot fire here:
                    // the correction is 0 (DE, walk) or (D-1) log Z with Z in [1/3, 3] (stretch), so
                    // the test is not on the consumer's issue stream (the oracle keeps it).
                    // (`ev` false = the proposal has no prior support: its log-density is
                    // invalid whatever loglike left in nlp / nll)
                    const bool valid = ev && ld_valid(PK, nlp, nll);
                    const double e = -logu;  // randexp(rng)
                    bool acc;
                    if (PK == KABC_POSTERIOR_KERNELIZED) {
                        const double lW = corr + (nlp + nll) - (lp + ll);
                        acc = valid && (-e <= lW);
                    } else if (PK == KABC_POSTERIOR_COMMON) {
                        const double lW = corr + nll - ll;  // src/types.jl:127
                        acc = valid && (-e <= lW);
                    } else {
                        const double lW = corr + nlp - lp;
                        const double mx = (A.eps > ll) ? A.eps : ll;
                        const double lW2 = mx - nll;
                        acc = valid && (-e <= lW) && (lW2 >= 0.0);
                    }
                    if (acc) {
#pragma unroll
                        for (int k = 0; k < D; ++k) x[k] = y[k];
                        lp = nlp;
                        ll = nll;
                        n_acc += 1u;
                    }
                    // (no code: pins the state to ONE set of registers here.  Without it hipcc
                    // keeps a second copy of x -- the value "after the loop" -- and every accept
                    // updates both: 8 v_mov_b64 per sub-step)
#pragma unroll
                    for (int k = 0; k < D; ++k) asm volatile("" : "+v"(x[k]));
                    if (dbg_on) {  // (a scalar: one s_cbranch, no lane mask on the consumer's stream)
                        int32_t* d = A.dbg + (r * A.dbg_nt + (A.dbg_s0 + s0 + si)) * 6;
                        d[0] = (int32_t)move;
                        d[1] = acc ? 1 : 0;
                        d[2] = (int32_t)(mva & 0x3fffffffu);
                        d[3] = move >= 2u ? (int32_t)R.bb[si][lane] : -1;
                        d[4] = move == 3u ? (int32_t)R.cc[si][lane] : -1;
                        d[5] = ev ? 1 : 0;
                    }
                };
                // an error (src/types.jl:69-70) is sticky and reported after the launch; the
                // remaining sub-steps still run (their result is discarded by the host), which
                // keeps the loop bounds wave-uniform
                static_assert(kChunk == 3, "the consumer spells out kChunk sub-steps");
                if constexpr (kLate) {
                    substep(SubStepIx<0>{}, r0a, r0b, r0a, r0b);
                    if (ns > 1) substep(SubStepIx<1>{}, r0a, r0b, r0a, r0b);
                    if (ns > 2) substep(SubStepIx<2>{}, r0a, r0b, r0a, r0b);
                } else {
                    substep(SubStepIx<0>{}, r0a, r0b, r1a, r1b);
                    if (ns > 1) substep(SubStepIx<1>{}, r1a, r1b, r0a, r0b);
                    if (ns > 2) substep(SubStepIx<2>{}, r0a, r0b, r1a, r1b);
                }
            }
            KABC_TIMED_BARRIER();
        }
    }

    if (tprobe && lane == 0) {
        const unsigned long long t_end = __builtin_amdgcn_s_memtime();
        int32_t* d = A.dbg + (r0 * A.nt) * 6 + 8 + wave * 2;
        d[0] = (int32_t)(t_end - t_begin);
        d[1] = (int32_t)t_bar;
    }
    if ((KABL & 64) && A.dbg && lane == 0) {
        // placement probe (KABC_ABLATE=64 with debug records on): HW_ID of each wave
        uint32_t hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        A.dbg[(r0 * A.nt) * 6 + wave] = (int32_t)hw;
    }
    if (wave == 0) {
        if (active) {
            store_row<D>(A.x_act + row * D, x);
            A.lp[r] = lp;
            A.ll[r] = ll;
            if (A.trace) {
                double xp[D];
#pragma unroll
                for (int k = 0; k < D; ++k)
                    xp[k] = (sprior[k].discrete && PK != KABC_POSTERIOR_COMMON)
                                ? kabc_rint(x[k]) : x[k];
                store_row<D>(A.trace + r * D, xp);
            }
        }
        // one atomic per batch and counter
        const unsigned long long se = wave_total_u32(n_eval);   // n <= ntransitions per lane
        const unsigned long long sa = wave_total_u32(n_acc);
        if (lane == 0) {
            unsigned long long* sl =
                A.slots + (size_t)((blockIdx.x + blockIdx.y * gridDim.x) & (kCounterSlots - 1)) * 8;
            atomicAdd(&sl[0], (unsigned long long)n_active * (unsigned long long)A.nt);
            atomicAdd(&sl[1], se);
            atomicAdd(&sl[2], sa);
        }
        if (err) atomicMax(&A.counters->error, err);
    }
}

// step(rng, model, spl::AIS; retry_sampling): one thread per owned walker; the
// retry budget is global (src/KissABC.jl:52-60), kept in a device counter.
template <int D>
__global__ void __launch_bounds__(kInitBlock) ais_init_kernel(const InitArgs A) {
    const int64_t r = (int64_t)blockIdx.x * kInitBlock + threadIdx.x;
    if (r >= A.rows_owned) return;
    const int64_t row = A.row_first + r;
    const uint32_t w = A.id_base + (uint32_t)row;
    const int64_t chain = (int64_t)blockIdx.y;
    const uint64_t seed = A.seeds ? A.seeds[chain] : A.seed;
    unsigned long long* retries = A.seeds ? A.chain_retries + chain : &A.counters->retries;
    double* x_act = A.x_act + (A.seeds ? chain * A.stride_act : 0);
    double* lp_out = A.lp + (A.seeds ? chain * A.stride_own : 0);
    double* ll_out = A.ll + (A.seeds ? chain * A.stride_own : 0);
    double x[D], xp[D];
    double lp = 0.0, ll = 0.0;
    uint64_t attempt = 0;
    while (true) {
#ifdef KABC_USER_SAMPLE_INIT
        // CommonLogDensity with the user's own sample_init (src/types.jl:105-113, :50 of
        // src/KissABC.jl: unconditional_sample): drawn by the plugin's function
        if (A.raw[0].kind == KABC_PRIOR_USER_INIT) {
            kabc_cost_rng_t irng = {seed, attempt, w, KABC_DOM_AIS_INIT, 0u};
            kabc_user_sample_init(x, D, A.cost_params, A.cost_data, A.cost_ndata, &irng);
        } else
#endif
        {
            for (int k = 0; k < D; ++k) {
                kabc_slotwin_t win = {seed, attempt, w, KABC_DOM_AIS_INIT,
                                      (uint32_t)k * KABC_SLOTS_PER_DIM};
                x[k] = kabc_sample_prior(&A.raw[k], &win);
            }
        }
        lp = factored_logpdf_push<D>(A.prior, x, xp);
        kabc_cost_rng_t rng = {seed, attempt, w, KABC_DOM_AIS_INIT_COST, 0u};
        if (A.posterior == KABC_POSTERIOR_COMMON) {
            lp = 0.0;
            ll = kabc_cost_eval(A.cost_id, x, D, A.cost_params, A.cost_data, A.cost_ndata, &rng);
        } else if (A.posterior == KABC_POSTERIOR_KERNELIZED) {
            ll = lp;
            if (kabc_isfinite(lp)) {
                const double c = kabc_cost_eval(A.cost_id, xp, D, A.cost_params, A.cost_data,
                                                A.cost_ndata, &rng);
                const double q = kabc_div_rc(c, A.eps, 1.0 / A.eps);
                ll = -0.5 * (q * q);
            }
        } else {
            ll = -lp;
            if (kabc_isfinite(lp))
                ll = kabc_cost_eval(A.cost_id, xp, D, A.cost_params, A.cost_data, A.cost_ndata,
                                    &rng);
        }
        if (ld_valid(A.posterior, lp, ll)) break;
        const unsigned long long used = atomicAdd(retries, 1ull) + 1ull;
        if (used > A.retry_budget) {
            A.counters->init_failed = 1;
            break;
        }
        ++attempt;
    }
    store_row<D>(x_act + row * D, x);
    lp_out[r] = lp;
    ll_out[r] = ll;
}

#ifndef __HIPCC_RTC__  // host side
// launchers (defined by the instantiation units)
// nchains = gridDim.y
using AisLaunchFn = void (*)(const AisArgs&, hipStream_t, unsigned nchains);
using AisLaunch = Launcher<AisArgs, unsigned>;       // host function or run-time compiled kernel
using AisInitLaunch = Launcher<InitArgs, unsigned>;
inline dim3 ais_half_geom(const AisArgs& a, unsigned nchains) {
    return dim3((unsigned)((a.rows_owned + kBatch - 1) / kBatch), nchains);
}
inline dim3 ais_init_geom(const InitArgs& a, unsigned nchains) {
    return dim3((unsigned)((a.rows_owned + kInitBlock - 1) / kInitBlock), nchains);
}
// pcx = prior class + kPriorClasses * (posterior kind - 1)
AisLaunch find_ais_kernel(int cost_id, int D, int pcx);
constexpr int kAisVariants = 3 * kPriorClasses;
struct ModelUnit;
bool launch_ais_init(int D, const InitArgs& a, hipStream_t s, unsigned nchains, ModelUnit* unit);

#endif

}  // namespace kabc
