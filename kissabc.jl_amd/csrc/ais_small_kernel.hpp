// ais_small_kernel.hpp -- sample(model, AIS(N), ...) for SMALL ensembles: ONE workgroup per chain,
// every generation of a kabc_ais_advance call inside ONE launch.
//
// Why: every testset and example of the reference runs AIS(10) ... AIS(500) with the default
// ntransitions = 1 and burn-ins of 10^3 .. 5 10^4 steps (src/KissABC.jl:71, test/runtests.jl:82-131,
// 177-198, examples/example_n1.jl:40).  On the half-generation kernel (ais_kernels.hpp) such a call
// is a chain of dependent launches of 5-9 us each for 1-2 us of work: the kernel boundary is what
// the device spends its time on.  With the whole ensemble in one workgroup the dependences of the
// schedule (half 0, then half 1, partners from the frozen complementary half) are
// ordered by LDS words, not by launches:
//   * both halves, their log-density pairs and the contract's tables live in LDS for the whole
//     launch; partner rows are LDS reads;
//   * CONSUMER waves (one per 64 walkers of a half, at most kAisSmallMaxConsumers; a consumer takes
//     the batches b = c, c + nC, ... of whichever half is active) run the state-dependent chain
//     proposal -> push_p -> prior -> cost -> accept of ais_half_kernel's consumer, sub-step after
//     sub-step, half after half, generation after generation;
//   * the remaining waves are PRODUCERS: what a transition draws is a pure function of
//     (seed, walker, t, slot), so they run AHEAD of the consumers through the (generation, half,
//     batch, sub-step) sequence and hand the records over through a ring of LDS slots -- a FULL
//     word per slot (producer -> consumer) and a DONE counter per consumer (consumer -> producers),
//     both plain LDS words polled with s_sleep; no workgroup barrier inside the loop;
//   * the one part of a record that depends on the ensemble -- the walk move's displacement
//     (src/transition.jl:24-43), which the half-generation kernel's producers form from the frozen
//     half -- is finished by the consumer itself when it takes the slot (the same dense
//     (walk lane, coordinate) items, the same expression);
//   * with two consumers a half-step ends in an LDS arrival counter between them.
// Same draws, same schedule, same operation order as ais_half_kernel: bit-identical to it and to
// the oracle's sync schedule (tests/test_gpu_ais_small.py), including the debug records, the trace
// rows push_p(x_i) of every generation (src/KissABC.jl:78), the counters and the "starting sample
// invalid." error (src/types.jl:70).
#pragma once

#include "ais_kernels.hpp"

namespace kabc {

constexpr int kAisSmallBlock = 512;
constexpr int kAisSmallWaves = kAisSmallBlock / kWave;
constexpr int kAisSmallMaxConsumers = 2;
// rows of a half the kernel holds (N <= 2 * rows): both halves + the ring must fit 160 KB of LDS
constexpr int ais_small_rmax(int D) { return D <= 8 ? 256 : 128; }
constexpr int kAisSmallLdsBudget = 160 * 1024;

struct AisSmallArgs {
    double* x[2];            // halves, GLOBAL [chain][rows[h]][D]
    double* lp[2];           // [chain][rows[h]]
    double* ll[2];
    double* trace;           // [generation - trace_from][chain][N][D]: push_p(x) after the generation, or NULL
    int32_t* dbg;            // optional [N][nt][6] per-transition records (of the LAST generation run)
    DevCounters* counters;
    unsigned long long* slots;  // [kCounterSlots][8]
    const double* cost_params;
    const double* cost_data;
    int64_t cost_ndata;
    int32_t rows[2];
    uint32_t id_base[2];     // global walker id of row 0 of each half
    uint64_t seed;
    uint64_t t0;             // transition counter of generation 0's first sub-step
    int32_t nt;              // ntransitions
    int32_t ngen;            // generations in this launch
    int32_t trace_from;      // first generation whose samples are written to `trace`
    int32_t nchains;
    double eps, reps, box_lp;
    const PriorDev* prior;   // [D]
    const uint64_t* seeds;   // [nchains] (batch handles), else NULL
    // a prepared cost with a grid-wide pre-pass (ais_aux_kernels.hpp): its words for EVERY sub-step of the
    // launch, per half [chain][ngen * nt][W][rows[h]]; NULL = the producers compute them
    const double* aux[2];
    int64_t stride_aux[2];   // doubles per chain
};

// one ring slot = the record of ONE (batch, sub-step) unit
template <int D, int NPRE2, int NAUX>
struct AisSmallSlot {
    RecBuf<D, 1> rec;
    double pre[NPRE2 > 0 ? NPRE2 : 1][kBatch];  // leading normal pairs of the cost's stream (ais_pre_blocks)
    double aux[NAUX > 0 ? NAUX : 1][kBatch];    // prepared-cost words (cost_aux_c)
    uint8_t listB[kBatch];                       // DE lanes, then walk lanes (produce_substep)
    int32_t counts[2];                           // nDE, nWK
    int32_t pad_[2];
};

template <int D, int COST, int PC>
struct AisSmallGeom {
    static constexpr int RMAX = ais_small_rmax(D);
    static constexpr int NPRE2 = 2 * ais_pre_blocks(COST, D);
    static constexpr int NAUX = cost_aux_c(COST);
    using Slot = AisSmallSlot<D, NPRE2, NAUX>;
    static constexpr int kNbTabs = (PC == kPriorGeneral) ? kNbTabsGeneral : 0;
    static constexpr int fixed_bytes =
        2 * RMAX * D * 8 + 4 * RMAX * 8 + KABC_MATH_TAB_WORDS * 8 + (kNbTabs > 0 ? (kNbTabs + 1) * kNbEntries * 8 : 8) +
        D * (int)sizeof(PriorDev) + 2 * D * 8 + D * 64 + 1024;
    static constexpr int fit = (kAisSmallLdsBudget - fixed_bytes) / (int)sizeof(Slot);
    // slots of the ring (even: split between two consumers)
    static constexpr int T = (fit > 12 ? 12 : fit) & ~1;
    static_assert(T >= 4, "the ring of the one-workgroup AIS kernel does not fit LDS");
};

struct AisSmallProdCtx {  // (what produce_substep reads of its arguments)
    int64_t n_comp;
    const double* x_comp;
    int32_t ablate;
};

// LDS hand-over words: the data they guard is LDS too, so the fences name that address space only
// (a fence over all address spaces would also wait for the trace rows' global stores)
__device__ __forceinline__ uint32_t lds_word_peek(const uint32_t* w) {
    return __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_word_publish(uint32_t* w, uint32_t v, int lane) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    if (lane == 0) __hip_atomic_store(w, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local"); }

template <int D, int COST, int PC, int PK>
__global__ void __launch_bounds__(kAisSmallBlock) ais_small_kernel(const AisSmallArgs A) {
    using G = AisSmallGeom<D, COST, PC>;
    using Slot = typename G::Slot;
    constexpr int RMAX = G::RMAX;
    constexpr int T = G::T;
    constexpr int kPre = ais_pre_blocks(COST, D);
    constexpr int kAuxW = cost_aux_c(COST);

    __shared__ __attribute__((aligned(16))) double sx[2][RMAX * D];
    __shared__ double slp[2][RMAX], sll[2][RMAX];
    __shared__ __attribute__((aligned(16))) Slot ring[T];
    __shared__ uint32_t s_full[T];                      // unit number + 1 of the record a slot holds
    __shared__ uint32_t s_done[kAisSmallMaxConsumers];  // units consumer c has taken
    __shared__ uint32_t s_arrive;                       // consumers' arrivals at half-step ends
    __shared__ PriorDev sprior[D];
    __shared__ double sbox_lo[D], sbox_hi[D];
    constexpr bool kGaussBox = PC == kPriorSimple || PC == kPriorNormal;
    __shared__ __attribute__((aligned(16))) double sgb[kGaussBox ? D : 1][8];
    __shared__ __attribute__((aligned(16))) double slogtab[KABC_MATH_TAB_WORDS];
    constexpr int kNbTabs = G::kNbTabs;
    __shared__ double snb[kNbTabs > 0 ? (kNbTabs + 1) * kNbEntries : 1];

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & (kWave - 1);
    const int64_t chain = (int64_t)blockIdx.x;
    const uint64_t seed = A.seeds ? A.seeds[chain] : A.seed;
    const int rows0 = A.rows[0], rows1 = A.rows[1];
    const int N = rows0 + rows1;
    double* const gx[2] = {A.x[0] + chain * rows0 * D, A.x[1] + chain * rows1 * D};
    double* const glp[2] = {A.lp[0] + chain * rows0, A.lp[1] + chain * rows1};
    double* const gll[2] = {A.ll[0] + chain * rows0, A.ll[1] + chain * rows1};

    // ---- stage: tables, prior, the ensemble, the hand-over words
    for (int i = tid; i < KABC_MATH_TAB_WORDS; i += kAisSmallBlock) slogtab[i] = kabc_log_tab[i];
    if (tid < D * (int)(sizeof(PriorDev) / 8))
        reinterpret_cast<double*>(sprior)[tid] = reinterpret_cast<const double*>(A.prior)[tid];
    if (tid < D) {
        sbox_lo[tid] = A.prior[tid].p[0];
        sbox_hi[tid] = A.prior[tid].p[1];
    }
    uint32_t dmask = 0, gmask = 0;
    for (int k = 0; k < D; ++k) {
        dmask |= (A.prior[k].discrete ? 1u : 0u) << k;
        gmask |= (gaussbox_is_gauss(A.prior[k].kind) ? 1u : 0u) << k;
    }
    if constexpr (kGaussBox) {
        if (tid < D) gaussbox_stage(sgb[tid], A.prior[tid]);
    }
    const GaussBoxPrior gbox = {sgb, gmask, dmask};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int rh = A.rows[h];
        for (int i = tid; i < rh * D; i += kAisSmallBlock) sx[h][i] = gx[h][i];
        for (int i = tid; i < rh; i += kAisSmallBlock) {
            slp[h][i] = glp[h][i];
            sll[h][i] = gll[h][i];
        }
    }
    if (tid < T) s_full[tid] = 0u;
    if (tid < kAisSmallMaxConsumers) s_done[tid] = 0u;
    if (tid == 0) s_arrive = 0u;
    if constexpr (kNbTabs > 0) {
        int slot = 0;
        bool has_nb = false;
        for (int k = 0; k < D; ++k) {  // (wave-uniform)
            if (A.prior[k].kind == KABC_PRIOR_NEGBINOMIAL) {
                has_nb = true;
                if (slot < kNbTabs) {
                    for (int i = tid; i < kNbEntries; i += kAisSmallBlock)
                        snb[slot * kNbEntries + i] = kabc_lgamma_t((double)i + A.prior[k].p[0], kabc_log_tab);
                    ++slot;
                }
            }
        }
        if (has_nb)
            for (int i = tid; i < kNbEntries; i += kAisSmallBlock) snb[kNbLg1Block * kNbEntries + i] = kabc_lgamma1_tab[i];
    }
    __syncthreads();
    if constexpr (kNbTabs > 0) {
        // p[2] of the staged NegativeBinomial components: the table slot, or -1 (ais_half_kernel)
        if (tid < D && sprior[tid].kind == KABC_PRIOR_NEGBINOMIAL) {
            int slot = 0;
            for (int k = 0; k < tid; ++k) slot += (sprior[k].kind == KABC_PRIOR_NEGBINOMIAL) ? 1 : 0;
            sprior[tid].p[2] = (slot < kNbTabs) ? (double)slot : -1.0;
        }
        __syncthreads();
    }

    // ---- roles
    const int nb0 = (rows0 + kBatch - 1) / kBatch, nb1 = (rows1 + kBatch - 1) / kBatch;
    const int nC = nb0 < kAisSmallMaxConsumers ? nb0 : kAisSmallMaxConsumers;  // (rows0 >= rows1, rows0 >= 1)
    const int S = T / nC;  // slots per consumer: consumer c's ring is ring[c * S .. c * S + S)
    const int nt = A.nt;

    if (wave >= nC) {
        // ================= PRODUCER q of consumer c
        const int p = wave - nC, P = kAisSmallWaves - nC;
        const int c = p % nC, q = p / nC;
        const int Pc = (P - c + nC - 1) / nC;
        const int Pe = Pc < S ? Pc : S;
        if (q < Pe) {
            const uint32_t nbc0 = nb0 > c ? (uint32_t)((nb0 - c + nC - 1) / nC) : 0u;
            const uint32_t nbc1 = nb1 > c ? (uint32_t)((nb1 - c + nC - 1) / nC) : 0u;
            const uint32_t per_gen = (nbc0 + nbc1) * (uint32_t)nt;
            const uint32_t total = per_gen * (uint32_t)A.ngen;  // (the host keeps it below 2^31)
            uint64_t seed_v = seed;
            asm volatile("" : "+v"(seed_v));
#pragma unroll 1
            for (uint32_t u = (uint32_t)q; u < total; u += (uint32_t)Pe) {
                const uint32_t slot = u % (uint32_t)S;
                if (u >= (uint32_t)S) {  // the slot's previous record (unit u - S) must have been taken
                    const uint32_t need = u - (uint32_t)S + 1u;
                    while (lds_word_peek(&s_done[c]) < need) __builtin_amdgcn_s_sleep(1);
                    lds_acquire();
                }
                const uint32_t g = u / per_gen;
                uint32_t r = u - g * per_gen;
                int h = 0;
                if (r >= nbc0 * (uint32_t)nt) {
                    h = 1;
                    r -= nbc0 * (uint32_t)nt;
                }
                const uint32_t j = r / (uint32_t)nt, s = r - j * (uint32_t)nt;
                const int b = c + (int)j * nC;
                const int rows_h = h ? rows1 : rows0;
                const int rem = rows_h - b * kBatch;
                const int n_active = rem >= kBatch ? kBatch : rem;
                const uint32_t w_base = A.id_base[h] + (uint32_t)(b * kBatch);
                const uint64_t t = A.t0 + (uint64_t)g * (uint64_t)nt + (uint64_t)s;
                Slot& SL = ring[c * S + (int)slot];
                const AisSmallProdCtx ctx = {(int64_t)(h ? rows0 : rows1), nullptr, 0};
                produce_substep<D, NoMid, false, RecBuf<D, 1>, AisSmallProdCtx>(
                    ctx, seed_v, SL.rec, 0, t, w_base, n_active, SL.listB, lane, slogtab, nullptr, NoMid(), SL.counts);
                if constexpr (kAuxW > 0) {
                    if (A.aux[h]) {  // (wave-uniform) the pre-pass has them: lane = walker copies its words
                        if (lane < n_active) {
                            const double* ax = A.aux[h] + chain * A.stride_aux[h];
                            const int64_t sa = (int64_t)g * (int64_t)nt + (int64_t)s;
#pragma unroll
                            for (int jw = 0; jw < kAuxW; ++jw)
                                SL.aux[jw][lane] = ax[(sa * kAuxW + jw) * (int64_t)rows_h + (int64_t)b * kBatch + lane];
                        }
                    } else {
                        // (a prepared cost without a grid-wide pre-pass: the same sequential arithmetic
                        // kabc_cost_eval would do in place -- include/kabc_costs.h)
                        kabc_cost_rng_t rng = {seed, t, w_base + (uint32_t)lane, KABC_DOM_AIS_COST, 0u, 0u, nullptr, slogtab};
                        double a[kAuxW];
                        kabc_cost_prepare(COST, A.cost_params, A.cost_data, A.cost_ndata, &rng, a);
#pragma unroll
                        for (int jw = 0; jw < kAuxW; ++jw) SL.aux[jw][lane] = a[jw];
                    }
                }
                if constexpr (kPre > 0) produce_cost_normals<kPre>(seed_v, t, w_base, lane, SL.pre, slogtab);
                lds_word_publish(&s_full[c * S + (int)slot], u + 1u, lane);
            }
        }
    } else {
        // ================= CONSUMER c
        const int c = wave;
        constexpr bool kBoxRegs = (PC == kPriorBox) && D <= 8;
        double blo[kBoxRegs ? D : 1], bhi[kBoxRegs ? D : 1];
        if constexpr (kBoxRegs) {
#pragma unroll
            for (int k = 0; k < D; ++k) {
                blo[k] = sbox_lo[k];
                bhi[k] = sbox_hi[k];
            }
        }
        const BoxPrior box = {kBoxRegs ? blo : sbox_lo, kBoxRegs ? bhi : sbox_hi, dmask, A.box_lp};
        constexpr int kRP = cost_reg_params(COST, D);
        double cpar[kRP > 0 ? kRP : 1];
#pragma unroll
        for (int k = 0; k < kRP; ++k) {
            cpar[k] = A.cost_params[k];
            asm volatile("" : "+v"(cpar[k]));
        }
        const double* const cparams = kRP > 0 ? cpar : A.cost_params;
        const bool dbg_on = __builtin_amdgcn_readfirstlane(A.dbg != nullptr ? 1 : 0) != 0;
        const bool trace_on = __builtin_amdgcn_readfirstlane(A.trace != nullptr ? 1 : 0) != 0;
        // a stochastic cost reads its slot's variates while it is evaluated: the slot goes back late
        constexpr bool kLateRelease = kPre > 0 || kAuxW > 0;

        // the NEXT unit's record (and, within a batch, its partner rows), requested one sub-step ahead: the hand-over
        // words, the record and the rows are three dependent LDS round trips -- at ntransitions >= 2 on a few walkers
        // they were a third of a sub-step
        // (up to eight parameters: beyond, the second register set does not fit beside the rows)
        constexpr bool kPrefetch = D <= 8;
        constexpr bool kPrefetchRows = kPrefetch;
        uint32_t n_mva = 0, n_bb = 0, n_cc = 0;
        double n_logu = 0.0;
        constexpr int kZ = D + 1 > 3 ? D + 1 : 3;  // DE: gamma + D normals; walk: three normals (D = 1: more than D + 1)
        double n_zs[kPrefetch ? kZ : 1];
        double n_pa[kPrefetchRows ? D : 1], n_pb[kPrefetchRows ? D : 1], n_pc[kPrefetchRows ? D : 1];
        bool have_nxt = false, have_rows = false;
        unsigned int n_eval = 0, n_acc = 0;
        int err = 0;
        uint32_t u = 0;  // this consumer's unit counter
        int slot = 0;
        uint32_t halfsteps = 0;
#pragma unroll 1
        for (int g = 0; g < A.ngen; ++g) {
#pragma unroll 1
            for (int h = 0; h < 2; ++h) {
                const int rows_h = h ? rows1 : rows0;
                const int nb = h ? nb1 : nb0;
                const double* const xc = sx[1 - h];  // the frozen complementary half
#pragma unroll 1
                for (int b = c; b < nb; b += nC) {
                    const int rem = rows_h - b * kBatch;
                    const int n_active = rem >= kBatch ? kBatch : rem;
                    const bool active = lane < n_active;
                    const int r = b * kBatch + lane;
                    const uint32_t w_base = A.id_base[h] + (uint32_t)(b * kBatch);
                    double x[D];
                    double lp = 0.0, ll = 0.0;
                    if (active) {
                        load_row<D>(&sx[h][r * D], x);
                        lp = slp[h][r];
                        ll = sll[h][r];
                        if (g == 0 && !ld_valid(PK, lp, ll)) err = 2;  // accept(): "old log-density is invalid"
                    }
#pragma unroll 1
                    for (int s = 0; s < nt; ++s) {
                        Slot& SL = ring[c * S + slot];
                        const uint64_t t = A.t0 + (uint64_t)g * (uint64_t)nt + (uint64_t)s;
                        // -- the unit's record: the registers requested during the previous sub-step, else
                        //    wait for the slot's FULL word and read it now
                        uint32_t mva, bbv, ccv;
                        double logu;
                        double zs[kZ];
                        if (kPrefetch && have_nxt) {
                            mva = n_mva;
                            bbv = n_bb;
                            ccv = n_cc;
                            logu = n_logu;
                            if constexpr (kPrefetch) {
#pragma unroll
                                for (int j = 0; j < kZ; ++j) zs[j] = n_zs[j];
                            }
                        } else {
                            while (lds_word_peek(&s_full[c * S + slot]) != u + 1u) __builtin_amdgcn_s_sleep(1);
                            lds_acquire();
                            mva = SL.rec.mva[0][lane];
                            bbv = SL.rec.bb[0][lane];
                            ccv = SL.rec.cc[0][lane];
                            logu = SL.rec.logu[0][lane];
#pragma unroll
                            for (int j = 0; j < kZ; ++j) zs[j] = SL.rec.zs[0][j][lane];
                        }
                        // -- the partner rows (a; b of DE / walk; c of walk -- b and c default to a: every lane
                        //    reads three valid rows, no divergence around the loads), requested during the previous
                        //    sub-step when that one worked on the same batch (the complementary half is frozen)
                        double pa[D], pb[D], pc[D];
                        if (kPrefetchRows && have_rows) {
                            if constexpr (kPrefetchRows) {
#pragma unroll
                                for (int k = 0; k < D; ++k) {
                                    pa[k] = n_pa[k];
                                    pb[k] = n_pb[k];
                                    pc[k] = n_pc[k];
                                }
                            }
                        } else {
                            load_row<D>(&xc[(mva & 0x3fffffffu) * D], pa);
                            load_row<D>(&xc[bbv * D], pb);
                            if constexpr (kPrefetch) load_row<D>(&xc[ccv * D], pc);  // (beyond: read by the walk lanes, below)
                        }
                        // is the NEXT unit's record there already?  (The producers run ahead: it usually is.)
                        const int nslot = slot + 1 == S ? 0 : slot + 1;
                        uint32_t nfull = 0u;
                        if constexpr (kPrefetch) nfull = lds_word_peek(&s_full[c * S + nslot]);
                        if constexpr (!kLateRelease) {
                            lds_word_publish(&s_done[c], u + 1u, lane);  // (this slot's words are in registers)
                        }
                        have_rows = false;
                        if constexpr (kPrefetch) {
                            have_nxt = __builtin_amdgcn_readfirstlane((int)nfull) == (int)(u + 2u);
                            if (have_nxt) {  // its words fly while this sub-step computes
                                lds_acquire();
                                const Slot& SN = ring[c * S + nslot];
                                n_mva = SN.rec.mva[0][lane];
                                n_bb = SN.rec.bb[0][lane];
                                n_cc = SN.rec.cc[0][lane];
                                n_logu = SN.rec.logu[0][lane];
#pragma unroll
                                for (int j = 0; j < kZ; ++j) n_zs[j] = SN.rec.zs[0][j][lane];
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        const uint32_t move = mva >> 30;
                        double y[D];
                        double corr = 0.0;
                        double xa[D];
                        bool acc = false, ev = false;
                        if (active) {
#pragma unroll
                        for (int k = 0; k < D; ++k) xa[k] = x[k] - pa[k];
                        if (move == 3u) {
                            if constexpr (!kPrefetch) load_row<D>(&xc[ccv * D], pc);
                            // ais_walk_propose  src/transition.jl:24-43: Xs = (a + (b + c)) / 3,
                            // W = z1 (a - Xs) + z2 (b - Xs) + z3 (c - Xs); the displacement depends on the ensemble,
                            // so it is formed here, not by the producers that run ahead of the half-steps
#pragma unroll
                            for (int k = 0; k < D; ++k) {
                                const double Xs = kabc_div_rc(pa[k] + (pb[k] + pc[k]), 3.0, 1.0 / 3.0);
                                const double W = zs[0] * (pa[k] - Xs) + zs[1] * (pb[k] - Xs) + zs[2] * (pc[k] - Xs);
                                y[k] = x[k] + W;
                            }
                        }
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
                                const double sk = kabc_fabs(pa[k] - pb[k]) + kabc_fabs(x[k] - pb[k]) + kabc_fabs(xa[k]);
                                const double Tk = kabc_div_rc(gamma * sk, 300.0, 1.0 / 300.0) * zs[1 + k];
                                y[k] = x[k] + Wk + Tk;
                            }
                        }
                        }  // (active: the proposal)
                        __builtin_amdgcn_sched_barrier(0);
                        if constexpr (kPrefetchRows) {
                            // the next sub-step of THIS batch: its partner rows come from the same frozen half
                            if (have_nxt && s + 1 < nt) {
                                load_row<D>(&xc[(n_mva & 0x3fffffffu) * D], n_pa);
                                load_row<D>(&xc[n_bb * D], n_pb);
                                load_row<D>(&xc[n_cc * D], n_pc);
                                have_rows = true;
                            }
                        }
                        if (active) {
                        // ld = loglike(density, push_p(density, p))   src/transition.jl:75
                        kabc_cost_rng_t rng = {seed, t, w_base + (uint32_t)lane, KABC_DOM_AIS_COST, 0u, 0u, nullptr, slogtab};
                        if constexpr (kAuxW > 0) {
                            rng.aux = &SL.aux[0][lane];
                            rng.aux_stride = kBatch;
                        }
                        if constexpr (kPre > 0) {
                            rng.pre = &SL.pre[0][lane];
                            rng.pre_n = (uint32_t)kPre;
                            rng.pre_stride = (uint32_t)kBatch;
                        }
                        double nlp, nll;
                        loglike<D, COST, PC>(sprior, box, gbox, PK, A.eps, A.reps, y, cparams, A.cost_data, A.cost_ndata,
                                             &rng, nlp, nll, ev, slogtab, kNbTabs > 0 ? snb : nullptr);
                        __builtin_amdgcn_sched_barrier(0);
                        n_eval += ev ? 1u : 0u;
                        // accept(...)  src/types.jl:62-75, :96-104 (as ais_half_kernel's consumer)
                        const bool valid = ev && ld_valid(PK, nlp, nll);
                        const double e = -logu;  // randexp(rng)
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
#pragma unroll
                        for (int k = 0; k < D; ++k) asm volatile("" : "+v"(x[k]));
                        if (dbg_on) {
                            const int64_t rg = (int64_t)(h ? rows0 : 0) + r;
                            int32_t* d = A.dbg + (rg * nt + s) * 6;
                            d[0] = (int32_t)move;
                            d[1] = acc ? 1 : 0;
                            d[2] = (int32_t)(mva & 0x3fffffffu);
                            d[3] = move >= 2u ? (int32_t)bbv : -1;
                            d[4] = move == 3u ? (int32_t)ccv : -1;
                            d[5] = ev ? 1 : 0;
                        }
                        }  // (active)
                        if constexpr (kLateRelease) {
                            lds_word_publish(&s_done[c], u + 1u, lane);  // (the cost has read the slot's variates)
                        }
                        ++u;
                        slot = slot + 1 == S ? 0 : slot + 1;
                    }
                    if (active) {
                        store_row<D>(&sx[h][r * D], x);
                        slp[h][r] = lp;
                        sll[h][r] = ll;
                        // the sample step() returns: push_p(x_i) after its last transition (src/KissABC.jl:78)
                        if (trace_on && g >= A.trace_from) {
                            double xp[D];
#pragma unroll
                            for (int k = 0; k < D; ++k)
                                xp[k] = (sprior[k].discrete && PK != KABC_POSTERIOR_COMMON) ? kabc_rint(x[k]) : x[k];
                            double* tr = A.trace + (((int64_t)(g - A.trace_from) * A.nchains + chain) * N +
                                                    (int64_t)(h ? rows0 : 0) + r) * D;
                            store_row<D>(tr, xp);
                        }
                    }
                }
                // the half-step ends: every consumer's rows of half h are in LDS before anybody draws
                // partners from them, and nobody still reads the other half's rows that come next
                ++halfsteps;
                if (nC > 1) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
                    if (lane == 0) __hip_atomic_fetch_add(&s_arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    const uint32_t want = halfsteps * (uint32_t)nC;
                    while (lds_word_peek(&s_arrive) < want) __builtin_amdgcn_s_sleep(0);
                    lds_acquire();
                } else {
                    wave_lds_fence();
                }
            }
        }
        // counters: one atomic per consumer and counter
        const unsigned long long se = wave_total_u32(n_eval);
        const unsigned long long sa = wave_total_u32(n_acc);
        if (lane == 0) {
            unsigned long long* sl = A.slots + (size_t)((unsigned)blockIdx.x & (kCounterSlots - 1)) * 8;
            if (c == 0) atomicAdd(&sl[0], (unsigned long long)N * (unsigned long long)nt * (unsigned long long)A.ngen);
            atomicAdd(&sl[1], se);
            atomicAdd(&sl[2], sa);
        }
        if (err) atomicMax(&A.counters->error, err);
    }
    __syncthreads();
    // ---- the state goes back where the other driver keeps it
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int rh = A.rows[h];
        for (int i = tid; i < rh * D; i += kAisSmallBlock) gx[h][i] = sx[h][i];
        for (int i = tid; i < rh; i += kAisSmallBlock) {
            glp[h][i] = slp[h][i];
            gll[h][i] = sll[h][i];
        }
    }
}

#ifndef __HIPCC_RTC__  // host side
using AisSmallLaunchFn = void (*)(const AisSmallArgs&, hipStream_t);
using AisSmallLaunch = Launcher<AisSmallArgs>;
inline dim3 ais_small_geom(const AisSmallArgs& a) { return dim3((unsigned)a.nchains); }
template <int D, int COST, int PC, int PK>
static void launch_ais_small(const AisSmallArgs& a, hipStream_t s) {
    if (a.nchains < 1) return;
    hipLaunchKernelGGL((ais_small_kernel<D, COST, PC, PK>), dim3((unsigned)a.nchains), dim3(kAisSmallBlock), 0, s, a);
}
// pcx = prior class + kPriorClasses * (posterior kind - 1); prebuilt for the classes BOX, NORMAL, GENERAL
AisSmallLaunch find_ais_small_kernel(int cost_id, int D, int pcx);
#endif

}  // namespace kabc
