// smc_loop_kernel.hpp -- the whole ε-loop of smc(prior, cost; ...) (src/smc.jl:131-199) as ONE
// persistent, cooperatively launched gfx950 kernel.
//
// The per-iteration work is small (C4: 32 768 particles, ~18 MB of algorithmic traffic) and the
// loop is long (190 iterations): three launches + their gaps per iteration cost more than the
// work.  Here the grid stays resident: one thread per particle, G = ceil(N/256) workgroups, a
// device-wide barrier where the reference's data dependences force one --
//
//   MCMC pass (src/smc.jl:160-191)  ->  B1  ->  ε-selection (:134-143)  ->  B2  ->  next pass
//
// i.e. TWO barriers per iteration in the common case:
//
//  * at the end of a pass every thread still holds its particle's cost in a register.  It adds
//    the cost's order-preserving key to a global 1024-bin histogram of a PREDICTED window below
//    the current ε (the next ε = quantile(Xs[alive], α) has so far always landed a few
//    ε-decrements below the current one) and its workgroup publishes a 64-byte record (alive
//    count, NaNs, key range, accepted / evaluated / proposed).  [B1]
//  * every workgroup reads the G records and the histogram and finds -- redundantly, so that
//    all of them take the same branches -- the bin b* holding the target rank.  Particles whose
//    key falls into b* append (key, index) to a global candidate list; every wavefront
//    publishes the ballot of "cost below b*" and "cost inside b*".  [B2]
//  * every workgroup ranks the (few dozen) candidates in LDS, which yields the two bracketing
//    order statistics and hence ε exactly (Statistics.quantile, type 7), and rebuilds the
//    complete new alive mask in LDS: ballot bits + the candidates below ε.  A popcount prefix
//    over the mask words gives ESS and turns the reference's resample index
//    repeat(idxalive, ceil(N/ESS))[1:N] (:146-147) into a pure LDS lookup
//    (binary search over word prefixes + select-in-word): no compaction pass, no index array.
//  * misses of the prediction, overfull bins and the first iterations (no prediction yet) take
//    extra histogram rounds over the exact key range, one barrier each.
//
// Retry passes (mcmc_retrys > 0) cost one barrier each.  The stop tests (:194-198), the
// iteration log and the buffer flip are evaluated by every workgroup from the same data.
//
// Barrier: sense-reversing atomic counter (tools/gridsync_probe.hip: 2.4 us at 32 workgroups,
// half the cost of cooperative groups' grid.sync()).  The kernel is launched with
// hipLaunchCooperativeKernel, so all G workgroups are co-resident by construction; the spin is
// bounded all the same (5 s of s_memrealtime) and ends in an error, never in a hung GPU.
#pragma once

#include "smc_kernels.hpp"

namespace kabc {

#ifndef KABC_LOOP_BLOCK
#define KABC_LOOP_BLOCK 256
#endif
// The phase stamps (KABC_SMC_STAMPS: s_memrealtime sums per phase, tools/smc_wall_probe.py) exist in the
// PROBES build of the library only (make PROBES=1 -> libkabc_hip_probes.so): their 24 accumulators are
// loop-carried 64-bit values -- compiled in, they took 78 of the kernel's 440 registers (round 6).
#ifdef KABC_PROBES
#define KABC_STAMPS_ON 1
#else
#define KABC_STAMPS_ON 0
#endif
constexpr int kLoopBlock = KABC_LOOP_BLOCK;
constexpr int kLoopWaves = kLoopBlock / kWave;
constexpr int kLoopMaxG = 256;              // N <= 65 536 (the mask lives in LDS)
constexpr int kLoopMaxWords = kLoopMaxG * kLoopWaves;
constexpr int kLoopBins = 1024;
constexpr int kLoopCand = 512;              // bin population that is ranked directly
constexpr int kLoopCandCap = 1024;          // capacity of the list (alive + dead particles of the bin)
constexpr int kLoopSlots = 8;               // ring of histogram / candidate slots, one per barrier


struct SmcLoopScratch {  // zeroed by the host before the launch
    unsigned long long bar_count, pad0[15];
    unsigned int bar_gen, pad1[31];
    unsigned int abort_flag, pad2[31];
    unsigned int hist[kLoopSlots][kLoopBins + 32];  // [kLoopBins] = alive keys below the window,
                                                    // [kLoopBins + 1] = alive keys above it
    unsigned int ncand[kLoopSlots][32];
    unsigned long long cand_key[kLoopSlots][kLoopCandCap];
    unsigned int cand_idx[kLoopSlots][kLoopCandCap];  // particle index | alive-before << 31
    unsigned long long mask_lt[2][kLoopMaxWords];     // per wavefront: cost below the bin
    unsigned long long mask_in[2][kLoopMaxWords];     // per wavefront: cost inside the bin
    // one record per workgroup and barrier: [0] alive | NaNs << 21 | accepted << 42 (of the pass
    // that just ended), [1] evaluated | proposed << 21, [2] min alive key, [3] ~max alive key
    unsigned long long part[2][kLoopMaxG][4];
    unsigned long long succ[2][kLoopMaxG];  // gather round: min alive key above the bin
    // XCD-aware barrier (below): one 128-byte line per word
    struct Line {
        unsigned long long v, pad[15];
    };
    Line xmembers[16];  // workgroups per XCD (counted once, in the prologue)
    Line nxcd;          // XCDs that hold at least one workgroup
    Line xcount[16];    // arrivals per XCD
    Line gcount;        // XCDs that have arrived
    Line gen2;          // generation of the XCD-aware barrier
    Line ticket[16];    // first workgroup of an XCD to leave a barrier
    Line xflag[16];     // the XCD's L2 has been invalidated for generation v
};

struct SmcLoopArgs {
    double* theta[2];
    double* X[2];
    double* lpi[2];
    uint8_t* alive;           // out: final alive mask
    SmcCtrl* ctrl;            // out: final loop state (same record the multi-kernel path leaves)
    SmcLoopScratch* scratch;
    kabc_smc_iter_t* log;
    int64_t log_cap;
    const double* cost_params;
    const double* cost_data;
    int64_t cost_ndata;
    int64_t N;
    uint64_t seed;
    double max_stretch, alpha, min_r_ess;
    SmcLoopParams loop;
    int32_t retry_n;          // 1 + mcmc_retrys
    unsigned long long* stamps;  // diagnostic (KABC_SMC_STAMPS): s_memrealtime sums per phase [16]
    const PriorDev* prior;    // [D] prepared components, device memory
};

// keys of this kernel: -0.0 is folded onto +0.0 so that key order and `<` on the values agree
__device__ __forceinline__ uint64_t loop_key(double x) { return key_of(x + 0.0); }

__device__ __forceinline__ uint64_t wave_min_u64(uint64_t v) {
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) {
        const uint64_t o = __shfl_xor(v, off, kWave);
        v = o < v ? o : v;
    }
    return v;
}
// minimum of a 64-bit per-lane value over the wavefront with DPP (row_shr 1/2/4/8, row_bcast
// 15/31; lanes without a source keep their own value), returned wave-uniform: ~40 VALU
// instead of 12 dependent LDS-crossbar shuffles
__device__ __forceinline__ uint64_t wave_min_u64_dpp(uint64_t v) {
#define KABC_MIN_STEP(ctrl, rmask)                                                                  \
    {                                                                                               \
        const unsigned lo_ = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)v, (int)(unsigned)v, ctrl, rmask, 0xf, false);                 \
        const unsigned hi_ = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)(v >> 32), (int)(unsigned)(v >> 32), ctrl, rmask, 0xf, false); \
        const uint64_t o_ = ((uint64_t)hi_ << 32) | lo_;                                            \
        v = o_ < v ? o_ : v;                                                                        \
    }
    KABC_MIN_STEP(0x111, 0xf)
    KABC_MIN_STEP(0x112, 0xf)
    KABC_MIN_STEP(0x114, 0xf)
    KABC_MIN_STEP(0x118, 0xf)
    KABC_MIN_STEP(0x142, 0xa)
    KABC_MIN_STEP(0x143, 0xc)
#undef KABC_MIN_STEP
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, kWave - 1);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), kWave - 1);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long wave_sum_all(unsigned long long v) {
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
    return v;
}

// ---- device-wide barriers ---------------------------------------------------------------
// PLAIN barrier number `nb` (0-based) of the G resident workgroups: a monotonically increasing
// arrival counter (no reset) and a generation word on its own cache line; every workgroup gives
// an agent-scope release before it arrives and an agent-scope acquire after it leaves.  On
// gfx950 those two fences are most of the barrier (tools/halfgen_floor_probe.hip: 5.3 us at 128
// workgroups, 2.4 us without them): a release writes the XCD's whole L2 back, an acquire
// invalidates it, and every one of the 16 workgroups of an XCD does both.  Used ONCE, in the
// prologue (it publishes the XCD populations the barrier below needs).
// Every wavefront's global stores must have been acknowledged by the L2 BEFORE its workgroup is
// counted as arrived: the release (buffer_wbl2) is issued by ONE wavefront -- of this workgroup or,
// with the XCD-aware barrier, of another one -- and only writes back what is in the L2 by then.
// __syncthreads() does not do that: outside threadgroup-split mode its workgroup-scope release
// waits for LDS / scalar traffic only (`s_waitcnt lgkmcnt(0); s_barrier` in the ISA).
__device__ __forceinline__ void loop_sync_stores_done() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}
__device__ __forceinline__ bool loop_plain_barrier(SmcLoopScratch* g, unsigned G, unsigned nb, int* s_ok) {
    loop_sync_stores_done();
    if (threadIdx.x == 0) {
        int ok = 1;
        __threadfence();
        if (atomicAdd(&g->bar_count, 1ull) + 1ull == (unsigned long long)(nb + 1u) * G) {
            __threadfence();
            atomicExch(&g->bar_gen, nb + 1u);
        }
        volatile unsigned* gen = &g->bar_gen;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        unsigned spins = 0;
        while (*gen < nb + 1u) {
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 1023u) == 0u && __builtin_amdgcn_s_memrealtime() - t0 > 500000000ull) {
                atomicExch(&g->abort_flag, 1u);
                ok = 0;
                break;
            }
        }
        __threadfence();
        *s_ok = ok;
    }
    __syncthreads();
    return *s_ok != 0;
}

// XCD-AWARE barrier (round 3; tools/xcd_barrier_probe.hip: rows written, barrier, random rows of
// other workgroups read and checked -- 9.4 -> 4.1 us per iteration at 128 workgroups, 33 -> 6.4 us
// at 512, no mismatch).  The cross-XCD coherence is paid once per XCD instead of once per
// workgroup:
//   arrive : a workgroup waits for its own stores (they are then in its XCD's L2, which all its
//            XCD's workgroups share) and counts itself on the XCD's counter; the LAST workgroup
//            of an XCD writes that L2 back (buffer_wbl2 sc1: every workgroup's dirty lines) and
//            counts the XCD on the global counter; the last XCD publishes the generation.
//   wait   : every workgroup polls the generation and then invalidates its CU's L1 and the XCD's
//            L2 (buffer_inv sc1), as before.
// The XCD of a workgroup is HW_REG_XCC_ID; the populations are counted in the kernel's prologue
// (nothing is assumed about the dispatcher's placement).  Arrive and wait stay separate calls:
// work placed between them runs in the shadow of the barrier.  wait() returns false on time-out
// / abort (uniform over the workgroup); the spins are bounded (5 s of s_memrealtime).
struct LoopXcd {
    unsigned xcd, members, nxcd;
};
__device__ __forceinline__ unsigned loop_xcc_id() {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 0xfu;
}
__device__ __forceinline__ unsigned long long loop_ld(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// thread 0; false on time-out or abort
__device__ __forceinline__ bool loop_spin(SmcLoopScratch* g, const unsigned long long* p,
                                          unsigned long long want) {
    if (loop_ld(p) >= want) return true;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    volatile unsigned* ab = &g->abort_flag;
    unsigned spins = 0;
    while (loop_ld(p) < want) {
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 1023u) == 0u) {
            if (*ab || __builtin_amdgcn_s_memrealtime() - t0 > 500000000ull) {  // 5 s @ 100 MHz
                atomicExch(&g->abort_flag, 1u);
                return false;
            }
        }
    }
    return true;
}
// (the part thread 0 runs once every wavefront's global writes have completed: after a
// __syncthreads; its own stores since then -- the workgroup's record -- are waited for here)
__device__ __forceinline__ void loop_barrier_arrive_t0(SmcLoopScratch* g, const LoopXcd& X, unsigned nb) {
#ifdef KABC_XBAR_PLAIN  // the round-2 barrier, for A/B runs: every workgroup releases at agent scope
    __threadfence();
    if (atomicAdd(&g->gcount.v, 1ull) + 1ull == (unsigned long long)(nb + 1u) * gridDim.x) {
        __threadfence();
        __hip_atomic_store(&g->gen2.v, (unsigned long long)(nb + 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return;
#endif
    // this thread's own stores (its workgroup's record, written after the workgroup-wide wait
    // above) must be in the L2 before the arrival is counted: a workgroup-scope fence does not wait
    // for them, and the atomic below is relaxed
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#ifdef KABC_XBAR_EARLYWB
    // start writing back what is dirty NOW (not waited for): the last arrival's write-back then
    // finds little left, as with the plain barrier, where early arrivals flush while they wait
    asm volatile("buffer_wbl2 sc1" ::: "memory");
#endif
    if (atomicAdd(&g->xcount[X.xcd].v, 1ull) + 1ull == (unsigned long long)(nb + 1u) * X.members) {
        asm volatile("buffer_wbl2 sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
        if (atomicAdd(&g->gcount.v, 1ull) + 1ull == (unsigned long long)(nb + 1u) * X.nxcd)
            __hip_atomic_store(&g->gen2.v, (unsigned long long)(nb + 1u), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
    }
}
__device__ __forceinline__ void loop_barrier_arrive(SmcLoopScratch* g, const LoopXcd& X, unsigned nb) {
    loop_sync_stores_done();
    if (threadIdx.x == 0) loop_barrier_arrive_t0(g, X, nb);
}
__device__ __forceinline__ bool loop_barrier_wait(SmcLoopScratch* g, const LoopXcd& X, unsigned nb, int* s_ok) {
    if (threadIdx.x == 0) {
        bool ok = loop_spin(g, &g->gen2.v, nb + 1u);
        // agent-scope acquire: this CU's vector L1 and the XCD's L2.  (There is no L1-only
        // invalidate on gfx950 -- buffer_inv sc0 is workgroup scope, a no-op outside
        // threadgroup-split mode -- so the acquire side stays per workgroup; the saving of this
        // barrier is the release side.)
        // (the invalidate completes asynchronously: wait for it HERE, before this wave passes the
        // workgroup barrier that lets the other waves load -- not by the accident of the load below)
        asm volatile("buffer_inv sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
        volatile unsigned* ab = &g->abort_flag;
        if (*ab) ok = false;
        *s_ok = ok ? 1 : 0;
    }
    __syncthreads();
    return *s_ok != 0;
}
__device__ __forceinline__ bool loop_barrier(SmcLoopScratch* g, const LoopXcd& X, unsigned nb, int* s_ok) {
    loop_barrier_arrive(g, X, nb);
    return loop_barrier_wait(g, X, nb, s_ok);
}

// m-th (0-based) set bit of the LDS mask through the exclusive popcount prefix
__device__ __forceinline__ int loop_select(const unsigned long long* words, const unsigned* excl,
                                           int nwords, unsigned m) {
    int lo = 0, hi = nwords - 1;  // last word with excl[w] <= m
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (excl[mid] <= m) lo = mid;
        else hi = mid - 1;
    }
    unsigned long long w = words[lo];
    unsigned r = m - excl[lo];
    int pos = 0;
#pragma unroll
    for (int width = 32; width >= 1; width >>= 1) {
        const unsigned c = (unsigned)__popcll(w & ((1ull << width) - 1ull));
        if (r >= c) {
            r -= c;
            w >>= width;
            pos += width;
        }
    }
    return lo * 64 + pos;
}

// fold of G per-workgroup records (4 words each: two packed sums, two minima) through LDS:
// independent, pipelined ds_reads instead of dependent cross-lane shuffle chains
struct LoopFold {
    unsigned long long a, b, kmin, kmaxn;
};
__device__ __forceinline__ LoopFold loop_fold(unsigned long long (*s_f)[4], unsigned long long (*s_f2)[4],
                                              unsigned G, int tid, unsigned long long r0,
                                              unsigned long long r1, unsigned long long r2,
                                              unsigned long long r3) {
    if (tid < (int)G) {
        s_f[tid][0] = r0;
        s_f[tid][1] = r1;
        s_f[tid][2] = r2;
        s_f[tid][3] = r3;
    }
    __syncthreads();
    if (tid < 16) {
        unsigned long long a = 0, b = 0, mn = ~0ull, mx = ~0ull;
        for (unsigned w = (unsigned)tid; w < G; w += 16u) {
            a += s_f[w][0];
            b += s_f[w][1];
            mn = s_f[w][2] < mn ? s_f[w][2] : mn;
            mx = s_f[w][3] < mx ? s_f[w][3] : mx;
        }
        s_f2[tid][0] = a;
        s_f2[tid][1] = b;
        s_f2[tid][2] = mn;
        s_f2[tid][3] = mx;
    }
    __syncthreads();
    LoopFold F = {0, 0, ~0ull, ~0ull};
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        F.a += s_f2[w][0];
        F.b += s_f2[w][1];
        F.kmin = s_f2[w][2] < F.kmin ? s_f2[w][2] : F.kmin;
        F.kmaxn = s_f2[w][3] < F.kmaxn ? s_f2[w][3] : F.kmaxn;
    }
    return F;
}

// x mod d for x < 2^24-ish and a wave-uniform d, with rcp = 1.0f / d: one multiply, a fix-up
// of +-1 (a 32-bit `%` is ~40 instructions)
__device__ __forceinline__ unsigned loop_mod(unsigned x, unsigned d, float rcp) {
    unsigned qd = (unsigned)((float)x * rcp);
    int r = (int)(x - qd * d);
    if (r < 0) r += (int)d;
    if (r >= (int)d) r -= (int)d;
    if (r >= (int)d) r -= (int)d;
    return (unsigned)r;
}

// three lookups "m-th alive particle" in lock step.  The alive particles are spread evenly
// (the mask is ~95 % ones everywhere), so the word holding the m-th one is guessed by
// interpolation, m * nwords / ESS, and corrected by a short walk over the prefix array: 1-3
// LDS reads instead of the 9 dependent ones of a binary search.  excl[nwords] = ESS.
__device__ __forceinline__ void loop_select3(const unsigned long long* words, const unsigned* excl,
                                             int nwords, float words_per_alive,
                                             const unsigned (&m)[3], int (&out)[3]) {
    int w[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        w[j] = (int)((float)m[j] * words_per_alive);
        w[j] = w[j] > nwords - 1 ? nwords - 1 : w[j];
    }
    bool moved = true;
    while (moved) {
        unsigned e0[3], e1[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            e0[j] = excl[w[j]];
            e1[j] = excl[w[j] + 1];
        }
        moved = false;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            if (e0[j] > m[j]) {
                --w[j];
                moved = true;
            } else if (e1[j] <= m[j]) {
                ++w[j];
                moved = true;
            }
        }
    }
    unsigned long long ww[3];
    unsigned r[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        ww[j] = words[w[j]];
        r[j] = m[j] - excl[w[j]];
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        int pos = 0;
#pragma unroll
        for (int width = 32; width >= 1; width >>= 1) {
            const unsigned c = (unsigned)__popcll(ww[j] & ((1ull << width) - 1ull));
            if (r[j] >= c) {
                r[j] -= c;
                ww[j] >>= width;
                pos += width;
            }
        }
        out[j] = w[j] * 64 + pos;
    }
}

template <int D, int COST, bool SIMPLE>
__global__ void __launch_bounds__(kLoopBlock) smc_loop_kernel(const SmcLoopArgs A) {
    __shared__ unsigned long long s_words[kLoopMaxWords];  // alive mask, one word per wavefront
    __shared__ unsigned int s_excl[kLoopMaxWords + 1];
    __shared__ unsigned long long s_ckey[kLoopCandCap];
    __shared__ unsigned int s_cidx[kLoopCandCap];
    __shared__ unsigned long long s_f[kLoopMaxG][4];
    __shared__ unsigned long long s_f2[16][4];
    __shared__ unsigned long long s_acc[4];  // publish accumulators: two packed sums, two minima
    // prepared prior components, read with wave-uniform LDS addresses (by-value kernel
    // arguments pin ~250 SGPRs and spill them to VGPR lanes, as in ais_half_kernel)
    __shared__ PriorDev s_prior[D];
    // the log / sin-cos table (include/kabc_math.h): every table-driven function of the pass (the
    // draws' Box-Muller pairs, log(rand), a GENERAL prior's logs, the cost's own draws) looks it up
    // here -- in global memory each look-up is a dependent L2 round trip
    __shared__ __attribute__((aligned(16))) double s_logtab[KABC_MATH_TAB_WORDS];
    __shared__ unsigned int s_scan[kLoopWaves];
    __shared__ unsigned long long s_ka, s_kb;
    __shared__ long long s_bin[4];  // b*, count, before, found
    __shared__ int s_ok;

    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
    const unsigned G = gridDim.x, bid = blockIdx.x;
    SmcLoopScratch* __restrict__ g = A.scratch;
    const int64_t N = A.N;
    const int64_t i = (int64_t)bid * kLoopBlock + tid;
    const bool in = i < N;
    const int nwords = (int)G * kLoopWaves;

    // (the stream seed as a vector-register value: the Philox round keys derived from it then do
    // not occupy twenty scalar registers; csrc/ais_kernels.hpp does the same)
    uint64_t seed_v = A.seed;
    asm volatile("" : "+v"(seed_v));
    unsigned long long t_prev = (KABC_STAMPS_ON && A.stamps && bid == 0) ? __builtin_amdgcn_s_memrealtime() : 0ull;
    // phase times are accumulated in registers and written once at the end (a global
    // read-modify-write per stamp cost more than the phases it measured)
    unsigned long long st_acc[24] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define KABC_LSTAMP(slot)                                                      \
    if (KABC_STAMPS_ON && A.stamps && bid == 0) {                                                \
        const unsigned long long t_now = __builtin_amdgcn_s_memrealtime();     \
        st_acc[slot] += t_now - t_prev;                                        \
        t_prev = t_now;                                                        \
    }
    // workgroup G-1 re-zeroes the histogram / candidate slot of barrier q-2: every workgroup
    // finished reading it before it arrived at barrier q-1
#define KABC_LOOP_RECYCLE()                                                                   \
    if (bid == G - 1u && q >= 2u) {                                                           \
        const unsigned z_ = (q - 2u) & (kLoopSlots - 1);                                      \
        for (int b_ = tid; b_ < kLoopBins + 32; b_ += kLoopBlock) g->hist[z_][b_] = 0u;       \
        if (tid == 0) g->ncand[z_][0] = 0u;                                                   \
    }

    if (tid == 0) {
        s_acc[0] = s_acc[1] = 0ull;
        s_acc[2] = s_acc[3] = ~0ull;
    }
    if (tid < D * (int)(sizeof(PriorDev) / 8))
        reinterpret_cast<double*>(s_prior)[tid] = reinterpret_cast<const double*>(A.prior)[tid];
    for (int j = tid; j < KABC_MATH_TAB_WORDS; j += kLoopBlock) s_logtab[j] = kabc_log_tab[j];
    __syncthreads();
    // the XCD populations of the XCD-aware barrier: counted here, published by one plain barrier
    LoopXcd X;
    X.xcd = loop_xcc_id();
    if (tid == 0 && atomicAdd(&g->xmembers[X.xcd].v, 1ull) == 0ull) atomicAdd(&g->nxcd.v, 1ull);
    const bool prologue_ok = loop_plain_barrier(g, G, 0u, &s_ok);
    X.members = (unsigned)loop_ld(&g->xmembers[X.xcd].v);
    X.nxcd = (unsigned)loop_ld(&g->nxcd.v);
    // loop state, identical in every workgroup
    unsigned q = 0;  // barriers passed
    int cur = 0;
    double eps = KABC_INF, eps_prev = KABC_INF, min_alive = 0.0;
    int flag = 0, resampled = 0, remap_pass = 0;
    long long iteration = 0, ess = 0, n_alive_now = N;
    unsigned long long pass = 0, acc_iter = 0, tot_evals = (unsigned long long)N, tot_props = 0;
    int passes_iter = 0;
    int error = prologue_ok ? 0 : 3;
    // own particle
    double Xi = in ? A.X[0][i] : 0.0;
    bool alive_i = in;
    unsigned n_acc = 0, n_eval = 0, n_prop = 0;  // of the pass that just ended
    // draws of the next pass, prepared in the shadow of B1
    constexpr int kPre = cost_pre_blocks(COST, D);
    unsigned nx_a = 0, nx_b = 0;
    double nx_s = 0.0, nx_lprob = 0.0;
    double nx_pre[kPre > 0 ? 2 * kPre : 1];
    constexpr int kPreA = (kPre + 1) / 2;   // blocks expanded in the shadow of B1; the rest in B2's
    uint64_t nps = 0;
    // the second half of the cost's normal pairs (shadow of B2, or -- on a retry pass, which has
    // no B2 -- right before the pass)
    auto expand_rest = [&]() __attribute__((always_inline)) {
        if constexpr (kPre > kPreA) {
            if (in) {
#pragma unroll
                for (int j = kPreA; j < kPre; ++j) {
                    const kabc_u128_t Bn =
                        kabc_stream_block(seed_v, (uint32_t)i, nps, (uint32_t)j, KABC_DOM_SMC_COST);
                    kabc_normal_pair_tab(kabc_lo64(Bn), kabc_hi64(Bn), &nx_pre[2 * j], &nx_pre[2 * j + 1],
                                         s_logtab);
                }
            }
        }
    };

    while (error == 0) {
        // ================= publish: record of this workgroup (+ histogram of the predicted window)
        const uint64_t key = loop_key(Xi);
        bool pred = false;
        uint64_t wlo = 0, whi = 0;
        int wshift = 0;
        if (iteration >= 2 && kabc_isfinite(eps) && kabc_isfinite(eps_prev)) {
            // window [key(ε) - 8 (key(ε_prev) - key(ε)), key(ε)]: alive costs never exceed ε
            whi = loop_key(eps);
            const uint64_t kp = loop_key(eps_prev);
            uint64_t d = kp > whi ? kp - whi : 1ull;
            d = d > (1ull << 56) ? (1ull << 56) : d;
            uint64_t span = d * 8ull;
            span = span < 4096ull ? 4096ull : span;
            wlo = whi > span ? whi - span : 0ull;
            const int bits = 64 - __clzll((long long)(whi - wlo));
            wshift = bits > 10 ? bits - 10 : 0;
            pred = true;
        }
        {
            const unsigned slot = q & (kLoopSlots - 1);
            bool below = false, above = false;
            if (pred && alive_i) {
                if (key < wlo) below = true;
                else if (key > whi) above = true;
                else atomicAdd(&g->hist[slot][(unsigned)((key - wlo) >> wshift)], 1u);
            }
            // counts are <= 65 536 in total: three 21-bit fields per word
            const unsigned long long c_al = (unsigned long long)__popcll(__ballot(alive_i));
            const unsigned long long c_nan = (unsigned long long)__popcll(__ballot(alive_i && Xi != Xi));
            const unsigned long long c_acc = (unsigned long long)__popcll(__ballot(n_acc != 0u));
            const unsigned long long c_ev = (unsigned long long)__popcll(__ballot(n_eval != 0u));
            const unsigned long long c_pr = (unsigned long long)__popcll(__ballot(n_prop != 0u));
            const unsigned long long c_bl = (unsigned long long)__popcll(__ballot(below));
            const unsigned long long c_ab = (unsigned long long)__popcll(__ballot(above));
            // (a 64-lane ds_min_u64 on one address serialises: 2.5 -> 7 us per iteration; the wave
            // reduction stays in shuffles, only the four wave results meet in LDS)
            const uint64_t kmn = wave_min_u64_dpp(alive_i ? key : ~0ull);
            const uint64_t kmxn = wave_min_u64_dpp(alive_i ? ~key : ~0ull);
            if (lane == 0) {  // four lanes per accumulator: no contention to speak of
                atomicAdd(&s_acc[0], c_al | (c_nan << 21) | (c_acc << 42));
                atomicAdd(&s_acc[1], c_ev | (c_pr << 21) | (c_bl << 42) | (c_ab << 53));
                atomicMin(&s_acc[2], kmn);
                atomicMin(&s_acc[3], kmxn);
            }
        }
        KABC_LSTAMP(21)
        loop_sync_stores_done();  // s_acc complete; every wavefront's global writes have reached the L2
        if (tid == 0) {
            const unsigned slot = q & (kLoopSlots - 1);
            const unsigned long long a = s_acc[0], b = s_acc[1];
            const unsigned bl = (unsigned)((b >> 42) & 0x7ffull), ab = (unsigned)(b >> 53);
            unsigned long long* P = g->part[q & 1][bid];
            P[0] = a;
            P[1] = b & ((1ull << 42) - 1ull);
            P[2] = s_acc[2];
            P[3] = s_acc[3];
            if (bl) atomicAdd(&g->hist[slot][kLoopBins], bl);
            if (ab) atomicAdd(&g->hist[slot][kLoopBins + 1], ab);
            s_acc[0] = s_acc[1] = 0ull;
            s_acc[2] = s_acc[3] = ~0ull;
            loop_barrier_arrive_t0(g, X, q);
        }
        KABC_LSTAMP(0)
        // ---- everything the NEXT pass draws, in the shadow of B1.  The streams are counter-based
        // (seed, particle, pass, slot), so nothing here depends on the ensemble: partner indices
        // (:163-164), the stretch normal, log(rand), and the leading normal pairs of the cost's own
        // stream.  If the loop ends instead, the work is discarded.  (Round 2 ran the first part of
        // this BEFORE arriving: with a release fence per workgroup an arrival waited ~3 us for the
        // write-back anyway.  With the XCD-aware release an arrival is 1 us, and 3 us of arithmetic
        // in front of it would delay the XCD's last arrival -- the one that starts the write-back.)
        if (in) {
            nps = pass + (iteration > 0 ? 2u : 1u);
            const uint32_t w = (uint32_t)i;
            const kabc_u128_t B0 = kabc_stream_block(seed_v, w, nps, 0u, KABC_DOM_SMC_MOVE);
            const kabc_u128_t B1 = kabc_stream_block(seed_v, w, nps, 1u, KABC_DOM_SMC_MOVE);
            const kabc_u128_t B2 = kabc_stream_block(seed_v, w, nps, 2u, KABC_DOM_SMC_MOVE);
            int64_t a = (int64_t)kabc_index32(kabc_lo64(B0), (uint32_t)N - 1u);
            a += (a >= i);
            const int64_t lo = a < i ? a : i, hi = a < i ? i : a;
            int64_t b = (int64_t)kabc_index32(kabc_hi64(B0), (uint32_t)N - 2u);
            b += (b >= lo);
            b += (b >= hi);
            nx_a = (unsigned)a;
            nx_b = (unsigned)b;
            double z0, z1;
            kabc_normal_pair_tab(kabc_lo64(B1), kabc_hi64(B1), &z0, &z1, s_logtab);
            nx_s = A.max_stretch * z0 / kabc_sqrt((double)D);
            nx_lprob = kabc_log_pn_tab(kabc_u01(kabc_lo64(B2)), s_logtab);  // u01 is a positive normal
            // the cost's normal pairs: the first half here, the rest in the shadow of B2
            if constexpr (kPre > 0) {
#pragma unroll
                for (int j = 0; j < kPreA; ++j) {
                    const kabc_u128_t Bn = kabc_stream_block(seed_v, w, nps, (uint32_t)j, KABC_DOM_SMC_COST);
                    kabc_normal_pair_tab(kabc_lo64(Bn), kabc_hi64(Bn), &nx_pre[2 * j], &nx_pre[2 * j + 1],
                                         s_logtab);
                }
            }
        }
        KABC_LSTAMP(22)
        if (!loop_barrier_wait(g, X, q, &s_ok)) {
            error = 3;
            break;
        }
        KABC_LSTAMP(1)
        // ================= B1 passed.  Everything the selection may need from global memory is
        // requested at once (a dependent round trip through the fabric costs 1-2 us): the
        // records, this thread's four bins of the window histogram and its two side counters.
        const unsigned slot1 = q & (kLoopSlots - 1);  // the slot the window histogram went to
        unsigned long long r0 = 0, r1 = 0, r2 = ~0ull, r3 = ~0ull;
        if (tid < (int)G) {
            const unsigned long long* P = g->part[q & 1][tid];
            r0 = P[0];
            r1 = P[1];
            r2 = P[2];
            r3 = P[3];
        }
        uint4 c4v = make_uint4(0u, 0u, 0u, 0u);
        unsigned w_below = 0, w_above = 0;
        if (pred) {
            if (tid < kLoopBins / 4) c4v = *reinterpret_cast<const uint4*>(&g->hist[slot1][tid * 4]);
            w_below = g->hist[slot1][kLoopBins];
            w_above = g->hist[slot1][kLoopBins + 1];
        }
        const LoopFold F = loop_fold(s_f, s_f2, G, tid, r0, r1, r2, r3);
        const long long n = (long long)(F.a & 0x1fffffull), nn = (long long)((F.a >> 21) & 0x1fffffull);
        const uint64_t kmin = F.kmin, kmaxn = F.kmaxn;
        acc_iter += (F.a >> 42);
        tot_evals += (F.b & 0x1fffffull);
        tot_props += ((F.b >> 21) & 0x1fffffull);
        ++q;
        KABC_LOOP_RECYCLE()
        n_acc = n_eval = n_prop = 0;

        if (iteration > 0) {
            // a pass of iteration `iteration` has ended: flip, retry or close the iteration
            pass += 1;
            passes_iter += 1;
            cur ^= 1;
            const bool enough = (double)acc_iter >= A.loop.mcmc_tol * (double)N;  // :192
            if (passes_iter < A.retry_n && !enough) {
                remap_pass = 0;  // later passes of an iteration read particle j from row j
                expand_rest();
                goto mcmc_pass;
            }
            if (bid == 0 && tid == 0 && A.log && iteration <= A.log_cap) {
                kabc_smc_iter_t L;
                L.eps = eps;
                L.ess = ess;
                L.accepted = (int64_t)acc_iter;
                L.resampled = resampled;
                L.flag = flag;
                L.mcmc_passes = passes_iter;
                L.reserved = 0;
                A.log[iteration - 1] = L;
            }
            const double acc = (double)acc_iter;  // :194-198
            if (2.0 * kabc_fabs(eps_prev - eps) < A.loop.r_epstol * (kabc_fabs(eps_prev) + kabc_fabs(eps)) ||
                eps <= A.loop.epstol || acc < A.loop.mcmc_tol * (double)N ||
                iteration >= A.loop.max_iterations)
                break;
        }
        if (n == 0 || nn > 0) {
            error = (nn > 0) ? 1 : 2;
            break;
        }
        KABC_LSTAMP(2)
        // ================= ε-selection of the next iteration (:134-143)
        {
            iteration += 1;
            acc_iter = 0;
            passes_iter = 0;
            const uint64_t kmax = ~kmaxn;
            const double mn = val_of(kmin);
            const double aleph = (double)n * A.alpha + (1.0 - A.alpha);
            long long j = (long long)aleph;
            if (j < 1) j = 1;
            if (j > n - 1) j = n - 1;
            if (n == 1) j = 1;
            double gq = aleph - (double)j;
            gq = gq < 0.0 ? 0.0 : (gq > 1.0 ? 1.0 : gq);
            long long kt = j - 1;             // rank (0-based) of order statistic a inside the range
            const bool need_b = (n > 1);      // order statistic b = rank kt + 1
            uint64_t rlo = kmin, rhi = kmax;  // key range known to hold rank kt
            int shift = 0;
            bool have_hist = false;
            if (pred && w_above == 0u && kt >= (long long)w_below) {
                have_hist = true;
                rlo = wlo;
                rhi = whi;
                shift = wshift;
                kt -= (long long)w_below;
            }
            int state = 0;  // 1 gather candidates, 2 every key of the range is the same
            long long cb = n;
            for (int round = 0; round < 12 && state == 0; ++round) {
                if (rlo == rhi) {  // (only reachable with kmin == kmax: all n keys equal)
                    state = 2;
                    break;
                }
                if (!have_hist) {
                    const int bits = 64 - __clzll((long long)(rhi - rlo));
                    shift = bits > 10 ? bits - 10 : 0;
                    const unsigned hslot = q & (kLoopSlots - 1);
                    if (alive_i && key >= rlo && key <= rhi)
                        atomicAdd(&g->hist[hslot][(unsigned)((key - rlo) >> shift)], 1u);
                    if (!loop_barrier(g, X, q, &s_ok)) {
                        error = 3;
                        break;
                    }
                    c4v = (tid < kLoopBins / 4) ? *reinterpret_cast<const uint4*>(&g->hist[hslot][tid * 4])
                                                : make_uint4(0u, 0u, 0u, 0u);
                    ++q;
                    KABC_LOOP_RECYCLE()
                }
                // the bin holding rank kt: scan of 1024 bins, 4 per thread
                const unsigned c4[4] = {c4v.x, c4v.y, c4v.z, c4v.w};
                const unsigned loc = c4[0] + c4[1] + c4[2] + c4[3];
                const unsigned incl = wave_scan_incl(loc);
                __syncthreads();
                if (lane == kWave - 1) s_scan[wave] = incl;
                if (tid == 0) s_bin[3] = 0;
                __syncthreads();
                unsigned woff = 0;
                for (int w = 0; w < wave; ++w) woff += s_scan[w];
                long long before = (long long)woff + incl - loc;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (c4[u] > 0u && kt >= before && kt < before + (long long)c4[u]) {  // one thread
                        s_bin[0] = tid * 4 + u;
                        s_bin[1] = c4[u];
                        s_bin[2] = before;
                        s_bin[3] = 1;
                    }
                    before += c4[u];
                }
                __syncthreads();
                if (!s_bin[3]) {  // cannot happen (kt < population of the range)
                    error = 2;
                    break;
                }
                const uint64_t nlo = rlo + ((uint64_t)s_bin[0] << shift);
                uint64_t nhi = nlo + ((1ull << shift) - 1ull);
                if (nhi > rhi || nhi < nlo) nhi = rhi;
                rlo = nlo;
                rhi = nhi;
                cb = s_bin[1];
                kt -= s_bin[2];
                have_hist = false;
                if (shift == 0) state = 2;
                else if (cb <= kLoopCand) state = 1;
            }
            if (error) break;
            if (state == 0) {
                error = 2;
                break;
            }
            KABC_LSTAMP(3)
            // ---- gather round: candidates of the bin, successor above it, the two ballots
            const unsigned gslot = q & (kLoopSlots - 1);
            const bool inbin = in && key >= rlo && key <= rhi;
            if (state == 1 && inbin) {
                const unsigned pos = atomicAdd(&g->ncand[gslot][0], 1u);
                if (pos < (unsigned)kLoopCandCap) {
                    g->cand_key[gslot][pos] = key;
                    g->cand_idx[gslot][pos] = (unsigned)i | (alive_i ? 0x80000000u : 0u);
                }
            }
            {
                const unsigned long long m_lt = __ballot(in && key < rlo);
                const unsigned long long m_in = __ballot(inbin);
                const uint64_t sc = wave_min_u64_dpp((alive_i && key > rhi) ? key : ~0ull);
                if (lane == 0) {
                    atomicMin(&s_acc[2], sc);
                    g->mask_lt[q & 1][bid * kLoopWaves + wave] = m_lt;
                    g->mask_in[q & 1][bid * kLoopWaves + wave] = m_in;
                }
                __syncthreads();
                if (tid == 0) {
                    g->succ[q & 1][bid] = s_acc[2];
                    s_acc[2] = ~0ull;
                }
            }
            KABC_LSTAMP(4)
            loop_barrier_arrive(g, X, q);
            expand_rest();
            if (!loop_barrier_wait(g, X, q, &s_ok)) {
                error = 3;
                break;
            }
            KABC_LSTAMP(5)
            // ================= B2 passed: one batch of loads again -- successors, mask words,
            // the candidate count and (speculatively) one candidate per thread
            const int per = (nwords + kLoopBlock - 1) / kLoopBlock;  // mask words per thread, <= 4
            const int w0 = tid * per;
            unsigned long long mw[4] = {0, 0, 0, 0}, mi[4] = {0, 0, 0, 0};
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (u < per && w0 + u < nwords) {
                    mw[u] = g->mask_lt[q & 1][w0 + u];
                    if (state == 2) mi[u] = g->mask_in[q & 1][w0 + u];
                }
            const uint64_t my_succ = (tid < (int)G) ? g->succ[q & 1][tid] : ~0ull;
            unsigned nc = 0;
            uint64_t ck = ~0ull;   // candidate `lane` of the list, in every wavefront
            unsigned ci = 0;
            if (state == 1) {
                nc = (unsigned)__builtin_amdgcn_readfirstlane((int)g->ncand[gslot][0]);  // a scalar
                ck = g->cand_key[gslot][lane];
                ci = g->cand_idx[gslot][lane];
            }
            // the first 64 candidates also go to LDS (dead / absent ones as the largest key, so
            // that they never count): all-pairs ranking and the mask patches then run as
            // unrolled loops of independent, pipelined broadcast reads
            if (tid < kWave) {
                const bool alive_c = (unsigned)tid < nc && (ci >> 31);
                s_ckey[tid] = alive_c ? ck : ~0ull;
                s_cidx[tid] = ((unsigned)tid < nc) ? (ci & 0x7fffffffu) : 0xffffffffu;
            }
            const LoopFold F2 = loop_fold(s_f, s_f2, G, tid, 0, 0, my_succ, ~0ull);
            const uint64_t succ = F2.kmin;
            KABC_LSTAMP(12)
            uint64_t ka = ~0ull, kb = ~0ull;
            const bool in_regs = (state == 1) && nc <= (unsigned)kWave;
            if (state == 1 && nc > (unsigned)kLoopCandCap) {
                error = 4;  // more (alive + dead) particles in one bin than the list holds
                break;
            }
            if (in_regs) {
                // the usual case (a few dozen candidates): every wavefront ranks the list itself,
                // lane = candidate, against the LDS copy -- no workgroup barrier
                const bool mine_ok = (unsigned)lane < nc && (ci >> 31);
                int rank = 0;
                const unsigned nc8 = (nc + 7u) & ~7u;
                for (unsigned o0 = 0; o0 < nc8; o0 += 8) {
                    uint64_t k2[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) k2[u] = s_ckey[o0 + u];
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        rank += (k2[u] < ck || (k2[u] == ck && o0 + u < (unsigned)lane && k2[u] != ~0ull)) ? 1 : 0;
                }
                const unsigned long long ma = __ballot(mine_ok && rank == (int)kt);
                const unsigned long long mb = __ballot(mine_ok && rank == (int)kt + 1);
                if (ma) {
                    const int l = __ffsll((long long)ma) - 1;
                    ka = ((uint64_t)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(ck >> 32), l) << 32) |
                         (unsigned)__builtin_amdgcn_readlane((int)(unsigned)ck, l);
                }
                if (mb) {
                    const int l = __ffsll((long long)mb) - 1;
                    kb = ((uint64_t)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(ck >> 32), l) << 32) |
                         (unsigned)__builtin_amdgcn_readlane((int)(unsigned)ck, l);
                }
            } else {
                if (state == 1) {
                    for (unsigned c = tid; c < nc; c += kLoopBlock) {
                        s_ckey[c] = g->cand_key[gslot][c];
                        s_cidx[c] = g->cand_idx[gslot][c];
                    }
                }
                if (tid == 0) {
                    s_ka = ~0ull;
                    s_kb = ~0ull;
                }
                __syncthreads();
                if (state == 1) {
                    // ranks among the candidates that were alive: key order, ties by list position
                    for (unsigned c = tid; c < nc; c += kLoopBlock) {
                        if (!(s_cidx[c] >> 31)) continue;
                        const uint64_t mine = s_ckey[c];
                        long long rank = 0;
                        for (unsigned o = 0; o < nc; ++o) {
                            const uint64_t k2 = s_ckey[o];
                            rank += ((s_cidx[o] >> 31) && (k2 < mine || (k2 == mine && o < c))) ? 1 : 0;
                        }
                        if (rank == kt) s_ka = mine;
                        if (rank == kt + 1) s_kb = mine;
                    }
                } else if (tid == 0) {  // state 2: every alive key of the range equals rlo
                    s_ka = rlo;
                    s_kb = (kt + 1 < cb) ? rlo : ~0ull;
                }
                __syncthreads();
                ka = s_ka;
                kb = s_kb;
            }
            KABC_LSTAMP(13)
            if (kb == ~0ull) kb = succ;
            {
                const double a = val_of(ka);
                const double b = need_b ? val_of(kb) : a;
                double e;
                if (kabc_isfinite(a) && kabc_isfinite(b)) e = a + gq * (b - a);
                else e = (1.0 - gq) * a + gq * b;
                eps_prev = eps;  // ϵv = ϵ
                eps = e;
                min_alive = mn;
                flag = (e > mn) ? 0 : 1;  // :135-141
            }
            // alive = Xs .< ϵ (or .<=): below the bin -> set; inside the bin -> by comparison.
            // Each thread owns `per` consecutive mask words (from the ballots); the candidates'
            // bits are added through LDS below; then it counts.
            {
                if (state == 2) {
                    const double x = val_of(rlo);
                    if (flag ? (x <= eps) : (x < eps)) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) mw[u] |= mi[u];
                    }
                }
                // The words go to LDS; the candidates that join the alive set OR their bit in (one
                // LDS atomic per candidate, issued by the lanes that hold them); every thread reads
                // its words back.  (Each thread scanning the whole candidate list for bits that
                // fall into its own words cost ~14 instructions per candidate and thread: 1.3 us
                // per iteration at C4's 24 candidates.)
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (u < per && w0 + u < nwords) s_words[w0 + u] = mw[u];
                __syncthreads();
                if (state != 2) {
                    if (in_regs) {
                        const double xc = val_of(ck);
                        if (wave == 0 && (unsigned)lane < nc && (flag ? (xc <= eps) : (xc < eps))) {
                            const unsigned idx = ci & 0x7fffffffu;
                            atomicOr(&s_words[idx >> 6], 1ull << (idx & 63u));
                        }
                    } else {
                        for (unsigned c = tid; c < nc; c += kLoopBlock) {
                            const double x = val_of(s_ckey[c]);
                            if (flag ? (x <= eps) : (x < eps)) {
                                const unsigned idx = s_cidx[c] & 0x7fffffffu;
                                atomicOr(&s_words[idx >> 6], 1ull << (idx & 63u));
                            }
                        }
                    }
                }
                __syncthreads();
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (u < per && w0 + u < nwords) mw[u] = s_words[w0 + u];
                unsigned loc = 0;
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (u < per && w0 + u < nwords) loc += (unsigned)__popcll(mw[u]);
                const unsigned incl = wave_scan_incl(loc);
                if (lane == kWave - 1) s_scan[wave] = incl;
                __syncthreads();
                unsigned woff = 0, total = 0;
                for (int w = 0; w < kLoopWaves; ++w) {
                    if (w < wave) woff += s_scan[w];
                    total += s_scan[w];
                }
                unsigned run = woff + incl - loc;
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (u < per && w0 + u < nwords) {
                        s_excl[w0 + u] = run;
                        run += (unsigned)__popcll(mw[u]);
                    }
                ess = (long long)total;
                if (tid == 0) s_excl[nwords] = total;
                __syncthreads();
            }
            KABC_LSTAMP(14)
            ++q;
            KABC_LOOP_RECYCLE()
            // Step 2 decision: α*ESS <= nparticles*min_r_ess  (:145)
            resampled = (A.alpha * (double)ess <= (double)N * A.min_r_ess) ? 1 : 0;
            if (resampled && ess == 0) {
                error = 2;
                break;
            }
            alive_i = in && (resampled ? true : ((s_words[i >> 6] >> (i & 63)) & 1ull) != 0ull);
            n_alive_now = resampled ? N : ess;
            remap_pass = resampled;
            KABC_LSTAMP(6)
            if (KABC_STAMPS_ON && A.stamps && bid == 0) {
                st_acc[8] += 1;
                st_acc[9] += nc;
                st_acc[10] += pred ? 1 : 0;
                st_acc[11] += (unsigned)(q);
            }
        }

    mcmc_pass:
        // ================= MCMC pass (:160-191): thread = particle, proposals from the frozen
        // buffer `cur`, results into `1 - cur`
        {
            const double* __restrict__ theta_src = A.theta[cur];
            const double* __restrict__ X_src = A.X[cur];
            const double* __restrict__ lpi_src = A.lpi[cur];
            const uint64_t ps = pass + 1u;
            if (in) {
                const bool remap = remap_pass != 0;
                const unsigned uess = (unsigned)ess;
                const uint32_t w = (uint32_t)i;
                const int64_t a = alive_i ? (int64_t)nx_a : i, b = alive_i ? (int64_t)nx_b : i;
                // idx = repeat(idxalive, ceil(N/m))[1:N]  (:146-147) through the LDS mask, the
                // three lookups in lock step; then all three rows are requested at once
                int64_t si = i, sa = a, sb = b;
                if (remap) {
                    const float rcp_ess = 1.0f / (float)uess;
                    const unsigned m3[3] = {loop_mod((unsigned)i, uess, rcp_ess),
                                            loop_mod((unsigned)a, uess, rcp_ess),
                                            loop_mod((unsigned)b, uess, rcp_ess)};
                    int o3[3];
                    loop_select3(s_words, s_excl, nwords, (float)nwords * rcp_ess, m3, o3);
                    si = o3[0];
                    sa = o3[1];
                    sb = o3[2];
                }
                KABC_LSTAMP(16)
                double th[D], ta[D], tb[D];
                load_row<D>(theta_src + si * D, th);
                double Xn = X_src[si];
                double lpi = lpi_src[si];
                load_row<D>(theta_src + sa * D, ta);
                load_row<D>(theta_src + sb * D, tb);
                if (alive_i) {
                    const double s = nx_s, lprob = nx_lprob;
                    kabc_cost_rng_t rng = {seed_v, ps, w, KABC_DOM_SMC_COST, 0u, 0u, nullptr, s_logtab};
                    if constexpr (kPre > 0) {
                        rng.pre = nx_pre;
                        rng.pre_n = (uint32_t)kPre;
                        rng.pre_stride = 1u;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    KABC_LSTAMP(17)
                    double prop[D], xp[D];
#pragma unroll
                    for (int k = 0; k < D; ++k) {
                        const double W = (tb[k] - ta[k]) * s;
                        prop[k] = th[k] + W;
                    }
                    n_prop = 1;
                    if (KABC_STAMPS_ON && A.stamps) {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        KABC_LSTAMP(18)
                    }
                    const double lpp = factored_logpdf_push<D, SIMPLE, false>(s_prior, prop, xp, s_logtab);
                    if (KABC_STAMPS_ON && A.stamps) {
                        asm volatile("" :: "v"(lpp));
                        KABC_LSTAMP(19)
                    }
                    if (!(lpp < 0.0 && !kabc_isfinite(lpp))) {  // :173
                        double lM = lpp - lpi + 0.0;
                        if (!(lM < 0.0)) lM = (lM != lM) ? lM : 0.0;
                        if (lprob < lM) {
                            const double Xp =
                                eval_cost<COST, D>(xp, A.cost_params, A.cost_data, A.cost_ndata, &rng);
                            n_eval = 1;
                            const bool reject = flag ? (Xp > eps) : (Xp >= eps);
                            if (!reject) {
#pragma unroll
                                for (int k = 0; k < D; ++k) th[k] = prop[k];
                                Xn = Xp;
                                lpi = lpp;
                                n_acc = 1;
                            }
                        }
                    }
                }
                if (KABC_STAMPS_ON && A.stamps) {
                    asm volatile("" :: "v"(Xn));
                    KABC_LSTAMP(20)
                }
                store_row<D>(A.theta[1 - cur] + i * D, th);
                A.X[1 - cur][i] = Xn;
                A.lpi[1 - cur][i] = lpi;
                Xi = Xn;
            }
        }
        KABC_LSTAMP(7)
    }

    if (KABC_STAMPS_ON && A.stamps && bid == 0 && tid == 0)
        for (int j = 0; j < 24; ++j) A.stamps[j] = st_acc[j];
    // ================= epilogue: the final alive mask and the control record
    if (in) A.alive[i] = alive_i ? 1 : 0;
    if (bid == 0 && tid == 0) {
        SmcCtrl c = {};
        c.eps = eps;
        c.eps_prev = eps_prev;
        c.min_alive = min_alive;
        c.ess = ess;
        c.n_alive = n_alive_now;
        c.iteration = iteration;
        c.flag = flag;
        c.resampled = resampled;
        c.error = error;
        c.done = 1;
        c.cur = cur;
        c.passes = passes_iter;
        c.pass = pass;
        c.accepted = acc_iter;
        c.cost_evals = tot_evals;
        c.proposals = tot_props;
        *A.ctrl = c;
    }
}

#ifndef __HIPCC_RTC__  // host side
// the persistent ε-loop kernel: cooperative launch (all G workgroups co-resident or an error)
template <int D, int COST, bool SIMPLE>
inline hipError_t launch_smc_loop(const SmcLoopArgs& a, unsigned G, hipStream_t s) {
    static const int max_blocks = [] {
        int dev = 0, per_cu = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess) return 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(
                &per_cu, (const void*)smc_loop_kernel<D, COST, SIMPLE>, kLoopBlock, 0) != hipSuccess)
            return 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
            return 0;
        return per_cu * cus;
    }();
    if ((int)G > max_blocks || G > (unsigned)kLoopMaxG) return hipErrorCooperativeLaunchTooLarge;
    SmcLoopArgs args = a;
    void* p[] = {&args};
    return hipLaunchCooperativeKernel((const void*)smc_loop_kernel<D, COST, SIMPLE>, dim3(G),
                                      dim3(kLoopBlock), p, 0, s);
}

using SmcLoopLaunchFn = hipError_t (*)(const SmcLoopArgs&, unsigned, hipStream_t);
// host launch function or a run-time compiled kernel (hipRTC plugin)
struct SmcLoopLaunch {
    SmcLoopLaunchFn fn = nullptr;
    void* mod = nullptr;
    SmcLoopLaunch() = default;
    SmcLoopLaunch(std::nullptr_t) {}
    SmcLoopLaunch(SmcLoopLaunchFn f) : fn(f) {}
    explicit SmcLoopLaunch(void* m) : mod(m) {}
    explicit operator bool() const { return fn != nullptr || mod != nullptr; }
    hipError_t operator()(const SmcLoopArgs& a, unsigned G, hipStream_t s) const {
        if (fn) return fn(a, G, s);
        if (G > (unsigned)kLoopMaxG) return hipErrorCooperativeLaunchTooLarge;
        return rtc_launch_cooperative(mod, G, (unsigned)kLoopBlock, &a, s);
    }
};
SmcLoopLaunch find_smc_loop_kernel(int cost_id, int D, bool simple_prior, ModelUnit* unit = nullptr);

#endif

}  // namespace kabc
