// smc_dsel_kernels.hpp -- the ε-selection of smc() (src/smc.jl:131-153) with the PARTICLES sharded
// over the ranks of a communicator (SURVEY §8e "SMC"; kabc_smc_run_dist with KABC_SMC_DIST_PARTICLES).
//
// Rank r owns the particles [p_lo, p_hi) (whole workgroups of 64 of the propose/accept kernel).  It
// holds the alive mask of its own range only and makes every pass of the selection over its own
// costs.  Two courses:
//   * the ONE-exchange course (dsel2_*, second half of this file; round 6): the usual one -- a predicted
//     key window shipped unasked, one all-gather, no host look; also taken, without any exchange, by the
//     sharded cost loop and by single-GPU runs of 2^20 particles and more;
//   * phase by phase (dsel_*, below; round 5): what a selection falls back to when the window misses,
//     and the course of runs with retry passes.  What the ranks have to agree on travels in four small
//     all-gathers (every rank then folds the world's contributions itself, in rank order: all ranks
//     compute the same words, nothing is broadcast):
//
//   begin    n, NaNs, key range of the alive costs: from the producers' per-workgroup partials, which
//            the pass's own all-gather already delivers (no exchange)
//   hist     1024-bin histogram of (key - klo) >> shift over the rank's alive costs   [all-gather 4 KB]
//   narrow   sum of the world's histograms, bin of the target rank -> next range; repeated until at
//            most 4096 keys are left (one round in practice)
//   collect  the rank's keys of that range                                  [all-gather <= 32 KB]
//   rank     the world's candidates narrowed in LDS, the last <= 64 ranked inside one wavefront ->
//            the two bracketing order statistics, ε, the `<` / `<=` flag  (the smallest key above the
//            range, when it is not among the candidates: one more pass + all-gather of one word)
//   count    new alive particles per workgroup slice and per rank              [all-gather 1 word]
//   compact  ESS, the resample decision, the rank's alive mask, its compacted alive indices in
//            ascending order                              [all-gather of the index, on a resample only]
//   finish   idx = repeat(idxalive, ...) source list assembled from the ranks' segments; control block
//
// The arithmetic is smc_select_kernel's (smc_kernels.hpp), phase by phase: same keys, same narrowing,
// same ε -- the result equals kabc_smc_run's bit for bit.  Single-workgroup kernels decide, grids
// without any device-wide barrier make the passes; the host drives the phases and looks at the
// DselState between them (the sharded path is host-synchronous per pass already).
#pragma once

#include "smc_kernels.hpp"

namespace kabc {

constexpr int kDselCandStride = kSelCand + 8;  // per rank: [0] count, [8 ...] keys
constexpr int kDselMaxGrid = 128;              // workgroups of a pass over one rank's particles
constexpr int kDsel2MaxGrid = 512;             // ... of the one-exchange course's passes (spec, apply, index)
constexpr int kDselSpecHead = 8;               // header words of a rank's slot of the one-exchange payload
constexpr int kDselSpecKeys = kDselSpecHead + kSelBins / 2;  // word offset of the keys
constexpr int kDselStage = 4096;               // window keys a workgroup stages in LDS
constexpr double kDselWinLo = 1.4, kDselWinHi = 0.65;  // the window [eps - 1.4 d, eps - 0.65 d], d = the last decrement

struct DselState {  // device; identical on every rank after every deciding kernel
    unsigned long long klo, khi, keya, keyb;
    long long kt, nrange, n;
    double gq, mn, eps;
    long long ESS;
    int32_t state;  // 0 narrowing, 1 collect and rank, 2 every key of the range equal, 3 keys known
    int32_t listed, needmin, need_scan, flag, resample, error, rounds;
    uint32_t ncand_all, pad;
    // the one-exchange course (dsel2_* below)
    unsigned long long wlo, whi;  // the predicted key window of this selection
    int32_t spec;                 // 1: a window was predicted, 2: every alive key is a candidate
    int32_t stalled;              // 1..: the course could not decide (reason); the host repeats the selection phase by phase
    long long stall_iteration;
};

struct DselArgs {
    const double* Xbuf[2];
    uint8_t* alive;  // this rank reads and writes [p_lo, p_hi)
    int32_t* cidx;   // [N] compacted alive indices of the whole ensemble (finish)
    SmcCtrl* ctrl;
    DselState* st;
    const unsigned long long* part;  // gathered per-workgroup cost statistics, [npart][4]
    int64_t npart;
    int64_t N, p_lo, p_hi;
    double alpha, min_r_ess;
    int32_t rank, world;
    unsigned int* hist;        // [world][kSelBins]
    unsigned long long* cand;  // [world][kDselCandStride]
    unsigned long long* misc;  // [world][8]: [0] smallest key above the range, [1] new alive count
    int32_t* seg;              // [world][seg_len]: compacted indices of each rank's range
    int64_t seg_len;
    unsigned int* sub_cnt;     // [kDsel2MaxGrid] new alive count per workgroup slice
    // the one-exchange course: [world][spec_stride] words, per rank kDselSpecHead header words
    // ([0] alive keys below the window, [1] keys inside, [2] smallest alive key above, [3] alive keys,
    // [4] NaNs among them, [5] smallest alive key, [6] ~largest), 1024 bin counts of the keys inside
    // (two per word), the keys inside
    unsigned long long* spec;
    int64_t spec_cap, spec_stride;
    unsigned long long* bin;   // [8 + kSelCand]: cursor, smallest key above the bin, ticket; the bin's keys
};

struct Dsel2End {  // the end of the previous iteration's pass, folded into dsel2_begin_kernel
    unsigned long long* slots;
    kabc_smc_iter_t* log;
    int64_t log_cap;
    SmcLoopParams P;
    double mcmc_tol;
    int32_t nregions, do_pass_end;
    int32_t end_only, pad;  // the pass end alone (before the host looks at the control block)
};

#ifdef KABC_SMC_SINGLE_UNIT
// the contiguous slice of workgroup `bid` of `G` inside [p_lo, p_hi), whole tiles of 1024
__device__ __forceinline__ void dsel_slice(const DselArgs& A, unsigned bid, unsigned G, int64_t* i_lo,
                                           int64_t* i_hi, int64_t* tile_lo, int64_t* tile_hi) {
    const int64_t len = A.p_hi - A.p_lo;
    const int64_t ntile = (len + kSelBlock - 1) / kSelBlock;
    const int64_t tpb = (ntile + G - 1) / G;
    int64_t t0 = (int64_t)bid * tpb, t1 = t0 + tpb;
    t0 = t0 < ntile ? t0 : ntile;
    t1 = t1 < ntile ? t1 : ntile;
    int64_t lo = A.p_lo + t0 * kSelBlock, hi = A.p_lo + t1 * kSelBlock;
    *i_lo = lo < A.p_hi ? lo : A.p_hi;
    *i_hi = hi < A.p_hi ? hi : A.p_hi;
    *tile_lo = t0;
    *tile_hi = t1;
}

__device__ __forceinline__ void dsel_fail(const DselArgs& A, int err) {
    A.st->error = err;
    A.ctrl->error = err;
    A.ctrl->done = 1;
}

// ε and the comparison flag from the two order statistics (src/smc.jl:134-141), thread 0
__device__ __forceinline__ void dsel_set_eps(const DselArgs& A, DselState& S) {
    const double a = val_of(S.keya);
    const double b = (S.n == 1) ? a : val_of(S.keyb);
    double eps;
    if (kabc_isfinite(a) && kabc_isfinite(b)) eps = a + S.gq * (b - a);
    else eps = (1.0 - S.gq) * a + S.gq * b;
    S.eps = eps;
    S.flag = (eps > S.mn) ? 0 : 1;
    S.state = 3;
}

// begin: one workgroup.  Folds the partials, validates, sets up the narrowing; clears this rank's
// contribution slots.
__global__ void __launch_bounds__(kSelBlock) dsel_begin_kernel(const DselArgs A) {
    __shared__ unsigned long long s_red[kSelBlock / kWave][4];
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wid = tid >> 6;
    if (A.ctrl->done) return;
    long long cnt = 0, nanc = 0;
    uint64_t kmin = ~0ull, kmaxn = ~0ull;
    for (int64_t b = tid; b < A.npart; b += kSelBlock) {
        const unsigned long long* p = A.part + (size_t)b * 4;
        cnt += (long long)p[0];
        nanc += (long long)p[1];
        kmin = p[2] < kmin ? p[2] : kmin;
        kmaxn = p[3] < kmaxn ? p[3] : kmaxn;
    }
    {
        const unsigned long long wc = wave_sum((unsigned long long)cnt);
        const unsigned long long wn = wave_sum((unsigned long long)nanc);
        for (int off = kWave / 2; off > 0; off >>= 1) {
            const uint64_t a = __shfl_down(kmin, off, kWave), b = __shfl_down(kmaxn, off, kWave);
            kmin = a < kmin ? a : kmin;
            kmaxn = b < kmaxn ? b : kmaxn;
        }
        if (lane == 0) {
            s_red[wid][0] = wc;
            s_red[wid][1] = wn;
            s_red[wid][2] = kmin;
            s_red[wid][3] = kmaxn;
        }
        __syncthreads();
    }
    for (int b = tid; b < kSelBins; b += kSelBlock) A.hist[(size_t)A.rank * kSelBins + b] = 0u;
    if (tid == 0) {
        long long n = 0, nn = 0;
        kmin = kmaxn = ~0ull;
        for (int w = 0; w < kSelBlock / kWave; ++w) {
            n += (long long)s_red[w][0];
            nn += (long long)s_red[w][1];
            kmin = s_red[w][2] < kmin ? s_red[w][2] : kmin;
            kmaxn = s_red[w][3] < kmaxn ? s_red[w][3] : kmaxn;
        }
        A.cand[(size_t)A.rank * kDselCandStride] = 0ull;
        A.misc[(size_t)A.rank * 8 + 0] = ~0ull;
        A.misc[(size_t)A.rank * 8 + 1] = 0ull;
        DselState S = {};
        S.n = n;
        if (n == 0 || nn > 0) {
            *A.st = S;
            dsel_fail(A, (nn > 0) ? 1 : 2);
            return;
        }
        const uint64_t kmax = ~kmaxn;
        S.mn = val_of(kmin);  // minimum(Xs[alive])
        // ranks of the two bracketing order statistics (Statistics.quantile, type 7)
        const double aleph = (double)n * A.alpha + (1.0 - A.alpha);
        long long j = (long long)aleph;
        if (j < 1) j = 1;
        if (j > n - 1) j = n - 1;
        if (n == 1) j = 1;
        double gq = aleph - (double)j;
        S.gq = gq < 0.0 ? 0.0 : (gq > 1.0 ? 1.0 : gq);
        S.klo = kmin;
        S.khi = kmax;
        S.kt = j - 1;
        S.nrange = n;
        S.state = (kmin == kmax) ? 2 : (n <= kSelCand ? 1 : 0);
        *A.st = S;
    }
}

// hist: this rank's alive keys of [klo, khi] into its slot of the gathered histogram
__global__ void __launch_bounds__(kSelBlock) dsel_hist_kernel(const DselArgs A) {
    __shared__ unsigned int hist[kSelBins];
    const int tid = threadIdx.x;
    if (A.ctrl->done || A.st->state != 0) return;
    const double* __restrict__ X = A.Xbuf[A.ctrl->cur];
    int64_t i_lo, i_hi, t0, t1;
    dsel_slice(A, blockIdx.x, gridDim.x, &i_lo, &i_hi, &t0, &t1);
    const uint64_t klo = A.st->klo, khi = A.st->khi;
    const uint64_t span = khi - klo;  // > 0
    const int bits = 64 - __clzll((long long)span);
    const int shift = bits > 10 ? bits - 10 : 0;
    for (int b = tid; b < kSelBins; b += kSelBlock) hist[b] = 0;
    __syncthreads();
    for_each_alive(A.alive, X, i_lo, i_hi, tid, [&](int64_t, double x) {
        const uint64_t k = key_of(x);
        if (k >= klo && k <= khi) atomicAdd(&hist[(unsigned)((k - klo) >> shift)], 1u);
    });
    __syncthreads();
    const unsigned c = hist[tid];
    if (c) atomicAdd(&A.hist[(size_t)A.rank * kSelBins + tid], c);
}

// narrow: one workgroup, after the all-gather of the histograms
__global__ void __launch_bounds__(kSelBlock) dsel_narrow_kernel(const DselArgs A) {
    __shared__ unsigned int s_wcnt[kSelBlock / kWave];
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wid = tid >> 6;
    if (A.ctrl->done || A.st->state != 0) return;
    const uint64_t klo = A.st->klo, khi = A.st->khi;
    const long long kt = A.st->kt;
    const int rounds = A.st->rounds;
    const uint64_t span = khi - klo;
    const int bits = 64 - __clzll((long long)span);
    const int shift = bits > 10 ? bits - 10 : 0;
    unsigned c = 0;
    for (int r = 0; r < A.world; ++r) c += A.hist[(size_t)r * kSelBins + tid];
    A.hist[(size_t)A.rank * kSelBins + tid] = 0u;  // this rank's slot, for the next round
    unsigned incl = c;
    for (int off = 1; off < kWave; off <<= 1) {
        const unsigned o = __shfl_up(incl, off, kWave);
        if (lane >= off) incl += o;
    }
    if (lane == kWave - 1) s_wcnt[wid] = incl;
    __syncthreads();
    unsigned woff = 0;
    for (int w = 0; w < wid; ++w) woff += s_wcnt[w];
    const long long before = (long long)woff + incl - c;
    __syncthreads();  // (every thread has read the state before one of them rewrites it)
    if (c > 0 && kt >= before && kt < before + (long long)c) {  // exactly one thread
        const uint64_t nlo = klo + ((uint64_t)tid << shift);
        uint64_t nhi = nlo + ((1ull << shift) - 1ull);
        if (nhi > khi || nhi < nlo) nhi = khi;
        A.st->klo = nlo;
        A.st->khi = nhi;
        A.st->kt = kt - before;
        A.st->nrange = c;
        A.st->state = (shift == 0) ? 2 : (c <= (unsigned)kSelCand ? 1 : 0);
        A.st->rounds = rounds + 1;
    }
}

// collect: this rank's alive keys of the range into its slot of the gathered candidate list
__global__ void __launch_bounds__(kSelBlock) dsel_collect_kernel(const DselArgs A) {
    __shared__ uint64_t cand[kSelCand];
    __shared__ unsigned int s_n, s_base;
    const int tid = threadIdx.x;
    if (A.ctrl->done || A.st->state != 1) return;
    const double* __restrict__ X = A.Xbuf[A.ctrl->cur];
    int64_t i_lo, i_hi, t0, t1;
    dsel_slice(A, blockIdx.x, gridDim.x, &i_lo, &i_hi, &t0, &t1);
    const uint64_t klo = A.st->klo, khi = A.st->khi;
    if (tid == 0) s_n = 0;
    __syncthreads();
    for_each_alive(A.alive, X, i_lo, i_hi, tid, [&](int64_t, double x) {
        const uint64_t k = key_of(x);
        if (k >= klo && k <= khi) cand[atomicAdd(&s_n, 1u)] = k;  // <= nrange <= kSelCand in all
    });
    __syncthreads();
    const unsigned mine = s_n;
    unsigned long long* slot = A.cand + (size_t)A.rank * kDselCandStride;
    if (tid == 0) s_base = mine ? (unsigned)atomicAdd(&slot[0], (unsigned long long)mine) : 0u;
    __syncthreads();
    for (unsigned q = tid; q < mine; q += kSelBlock) slot[8 + s_base + q] = cand[q];
}

// rank: one workgroup, after the all-gather of the candidates (or straight after the narrowing when
// every key of the range is equal): the two order statistics; ε unless a scan for the smallest key
// above the range is needed first
__global__ void __launch_bounds__(kSelBlock) dsel_rank_kernel(const DselArgs A) {
    __shared__ unsigned int hist[kSelBins];
    __shared__ uint64_t cand[kSelCand];
    __shared__ uint64_t sh_u[kSelBlock / kWave];
    __shared__ unsigned int s_wcnt[kSelBlock / kWave];
    __shared__ uint64_t s_klo, s_khi, s_keya, s_keyb;
    __shared__ long long s_kt, s_nrange;
    __shared__ unsigned int s_ncand;
    __shared__ int s_state, s_needmin;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wid = tid >> 6;
    if (A.ctrl->done) return;
    const int state0 = A.st->state;
    if (state0 != 1 && state0 != 2) return;
    const long long n = A.st->n;
    if (tid == 0) {
        s_klo = A.st->klo;
        s_khi = A.st->khi;
        s_kt = A.st->kt;
        s_nrange = A.st->nrange;
        s_state = state0;
        s_keya = s_keyb = 0;
        s_needmin = 0;
    }
    unsigned nc = 0;
    int listed = 0;
    if (state0 == 1) {
        // the world's candidates, rank after rank, into LDS (their order does not matter)
        unsigned off = 0;
        for (int r = 0; r < A.world; ++r) {
            const unsigned long long* slot = A.cand + (size_t)r * kDselCandStride;
            const unsigned m = (unsigned)slot[0];
            for (unsigned q = tid; q < m && off + q < (unsigned)kSelCand; q += kSelBlock) cand[off + q] = slot[8 + q];
            off += m;
        }
        nc = off < (unsigned)kSelCand ? off : (unsigned)kSelCand;  // == nrange
        listed = 1;
        __syncthreads();
        if (tid == 0) {
            A.cand[(size_t)A.rank * kDselCandStride] = 0ull;  // this rank's slot, for the next iteration
            s_state = (s_nrange <= kWave) ? 3 : 0;
        }
        __syncthreads();
        for (int round = 0; round < 12 && s_state == 0; ++round) {
            const uint64_t klo = s_klo, khi = s_khi;
            const uint64_t span = khi - klo;
            if (span == 0) {  // all remaining keys equal
                if (tid == 0) s_state = 2;
                __syncthreads();
                break;
            }
            const int bits = 64 - __clzll((long long)span);
            const int shift = bits > 10 ? bits - 10 : 0;
            for (int b = tid; b < kSelBins; b += kSelBlock) hist[b] = 0;
            __syncthreads();
            for (unsigned i = tid; i < nc; i += kSelBlock) {
                const uint64_t k = cand[i];
                if (k >= klo && k <= khi) atomicAdd(&hist[(unsigned)((k - klo) >> shift)], 1u);
            }
            __syncthreads();
            const unsigned c = hist[tid];
            unsigned incl = c;
            for (int off2 = 1; off2 < kWave; off2 <<= 1) {
                const unsigned o = __shfl_up(incl, off2, kWave);
                if (lane >= off2) incl += o;
            }
            if (lane == kWave - 1) s_wcnt[wid] = incl;
            __syncthreads();
            unsigned woff = 0;
            for (int w = 0; w < wid; ++w) woff += s_wcnt[w];
            const long long before = (long long)woff + incl - c, kt = s_kt;
            __syncthreads();
            if (c > 0 && kt >= before && kt < before + (long long)c) {
                const uint64_t nlo = klo + ((uint64_t)tid << shift);
                uint64_t nhi = nlo + ((1ull << shift) - 1ull);
                if (nhi > khi || nhi < nlo) nhi = khi;
                s_klo = nlo;
                s_khi = nhi;
                s_kt = kt - before;
                s_nrange = c;
                s_state = (shift == 0) ? 2 : (c <= (unsigned)kWave ? 3 : 0);
            }
            __syncthreads();
        }
        if (s_state == 3) {
            // <= 64 keys left in [klo, khi]: one per lane of wave 0, each lane counts how many precede it
            if (tid == 0) s_ncand = 0;
            __syncthreads();
            const uint64_t klo = s_klo, khi = s_khi;
            for (unsigned i = tid; i < nc; i += kSelBlock) {
                const uint64_t k = cand[i];
                if (k >= klo && k <= khi) {
                    const unsigned pos = atomicAdd(&s_ncand, 1u);
                    reinterpret_cast<uint64_t*>(hist)[pos] = k;  // hist is free now: 64 x u64
                }
            }
            __syncthreads();
            if (wid == 0) {
                const unsigned m = s_ncand;  // == s_nrange
                const uint64_t mine = (lane < (int)m) ? reinterpret_cast<uint64_t*>(hist)[lane] : ~0ull;
                unsigned rank = 0;
                for (unsigned q = 0; q < m; ++q) {
                    const uint64_t other = __shfl(mine, (int)q, kWave);
                    rank += (other < mine || (other == mine && q < (unsigned)lane)) ? 1u : 0u;
                }
                const long long kt = s_kt;
                if (lane < (int)m && rank == (unsigned)kt) s_keya = mine;
                if (lane < (int)m && rank == (unsigned)kt + 1u) s_keyb = mine;
                if (lane == 0) s_needmin = (kt + 1 < (long long)m) ? 0 : 1;
            }
            __syncthreads();
            if (tid == 0 && s_needmin) s_keyb = ~0ull;
        }
    }
    __syncthreads();
    if (s_state == 2 && tid == 0) {  // every key of the range equals klo
        s_keya = s_klo;
        s_needmin = (s_kt + 1 < s_nrange) ? 0 : 1;
        s_keyb = s_needmin ? ~0ull : s_klo;
    }
    __syncthreads();
    int need_scan = 0;
    if (s_needmin && n > 1) {
        // rank j is the smallest alive key above the final range: nearly always among the candidates
        const uint64_t khi = s_khi;
        uint64_t kgt = ~0ull;
        if (listed) {
            for (unsigned q = tid; q < nc; q += kSelBlock) {
                const uint64_t k = cand[q];
                if (k > khi) kgt = k < kgt ? k : kgt;
            }
            kgt = block_min_u64(kgt, sh_u);
        }
        if (kgt == ~0ull) need_scan = 1;  // (uniform: the same value in every thread)
        else if (tid == 0) s_keyb = kgt;
    }
    __syncthreads();
    if (tid == 0) {
        DselState S = *A.st;
        S.klo = s_klo;
        S.khi = s_khi;
        S.kt = s_kt;
        S.nrange = s_nrange;
        S.keya = s_keya;
        S.keyb = s_keyb;
        S.needmin = s_needmin;
        S.listed = listed;
        S.ncand_all = nc;
        S.need_scan = need_scan;
        if (!need_scan) dsel_set_eps(A, S);
        else S.state = 4;  // waiting for the scan
        *A.st = S;
    }
}

// scan for the smallest alive key above the range (only when the range ends the candidates' bin)
__global__ void __launch_bounds__(kSelBlock) dsel_above_kernel(const DselArgs A) {
    __shared__ uint64_t sh_u[kSelBlock / kWave];
    const int tid = threadIdx.x;
    if (A.ctrl->done || A.st->state != 4) return;
    const double* __restrict__ X = A.Xbuf[A.ctrl->cur];
    int64_t i_lo, i_hi, t0, t1;
    dsel_slice(A, blockIdx.x, gridDim.x, &i_lo, &i_hi, &t0, &t1);
    const uint64_t khi = A.st->khi;
    uint64_t kgt = ~0ull;
    for_each_alive(A.alive, X, i_lo, i_hi, tid, [&](int64_t, double x) {
        const uint64_t k = key_of(x);
        if (k > khi) kgt = k < kgt ? k : kgt;
    });
    kgt = block_min_u64(kgt, sh_u);
    if (tid == 0 && kgt != ~0ull) atomicMin(&A.misc[(size_t)A.rank * 8 + 0], (unsigned long long)kgt);
}
__global__ void dsel_above_fold_kernel(const DselArgs A) {
    if (A.ctrl->done || A.st->state != 4) return;
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        uint64_t kgt = ~0ull;
        for (int r = 0; r < A.world; ++r) {
            const uint64_t v = A.misc[(size_t)r * 8 + 0];
            kgt = v < kgt ? v : kgt;
        }
        A.misc[(size_t)A.rank * 8 + 0] = ~0ull;
        DselState S = *A.st;
        S.keyb = kgt;
        dsel_set_eps(A, S);
        *A.st = S;
    }
}

// count: new alive particles of every workgroup slice of this rank
__global__ void __launch_bounds__(kSelBlock) dsel_count_kernel(const DselArgs A) {
    __shared__ long long sh_ll[kSelBlock / kWave];
    const int tid = threadIdx.x;
    if (A.ctrl->done || A.st->state != 3) return;
    const double* __restrict__ X = A.Xbuf[A.ctrl->cur];
    int64_t i_lo, i_hi, t0, t1;
    dsel_slice(A, blockIdx.x, gridDim.x, &i_lo, &i_hi, &t0, &t1);
    const double eps = A.st->eps;
    const int flag = A.st->flag;
    long long mycnt = 0;
    for (int64_t i = i_lo + tid; i < i_hi; i += kSelBlock) {
        const double x = X[i];
        mycnt += (flag ? (x <= eps) : (x < eps)) ? 1 : 0;
    }
    mycnt = block_sum_ll(mycnt, sh_ll);
    if (tid == 0) {
        A.sub_cnt[blockIdx.x] = (unsigned)mycnt;
        if (mycnt) atomicAdd(&A.misc[(size_t)A.rank * 8 + 1], (unsigned long long)mycnt);
    }
}

// compact: after the all-gather of the ranks' counts.  ESS, the resample decision (src/smc.jl:145), this
// rank's alive mask, its compacted indices (ascending) into its segment.
__global__ void __launch_bounds__(kSelBlock) dsel_compact_kernel(const DselArgs A) {
    __shared__ unsigned int s_cnt4[4 * (kSelBlock / kWave)];
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wid = tid >> 6;
    if (A.ctrl->done || A.st->state != 3) return;
    const double* __restrict__ X = A.Xbuf[A.ctrl->cur];
    int64_t i_lo, i_hi, tile_lo, tile_hi;
    dsel_slice(A, blockIdx.x, gridDim.x, &i_lo, &i_hi, &tile_lo, &tile_hi);
    const double eps = A.st->eps;
    const int flag = A.st->flag;
    long long ESS = 0;
    for (int r = 0; r < A.world; ++r) ESS += (long long)A.misc[(size_t)r * 8 + 1];
    long long base = 0;  // inside this rank's segment
    for (unsigned b = 0; b < blockIdx.x; ++b) base += (long long)A.sub_cnt[b];
    const int resample = (A.alpha * (double)ESS <= (double)A.N * A.min_r_ess) ? 1 : 0;
    const bool fail = resample && ESS == 0;
    int32_t* __restrict__ seg = A.seg + (size_t)A.rank * A.seg_len;
    for (int64_t tile0 = tile_lo; tile0 < tile_hi; tile0 += 4) {
        bool al[4];
        unsigned long long bm[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t i = A.p_lo + (tile0 + u) * kSelBlock + tid;
            double x = 0.0;
            const bool in = (tile0 + u < tile_hi) && i < A.p_hi;
            if (in) x = X[i];
            al[u] = in && (flag ? (x <= eps) : (x < eps));
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            bm[u] = __ballot(al[u]);
            if (lane == 0) s_cnt4[u * (kSelBlock / kWave) + wid] = (unsigned)__popcll(bm[u]);
        }
        __syncthreads();
        const unsigned c = s_cnt4[lane];
        unsigned incl = c;
        for (int off = 1; off < kWave; off <<= 1) {
            const unsigned o = __shfl_up(incl, off, kWave);
            if (lane >= off) incl += o;
        }
        const unsigned excl = incl - c;
        const unsigned tot = (unsigned)__shfl((int)incl, kWave - 1, kWave);
        const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned woff = (unsigned)__shfl((int)excl, u * (kSelBlock / kWave) + wid, kWave);
            if (al[u])
                seg[base + woff + __popcll(bm[u] & below)] =
                    (int32_t)(A.p_lo + (tile0 + u) * kSelBlock + tid);
        }
        base += tot;
        __syncthreads();
    }
    if (resample && !fail) {
        for (int64_t i = i_lo + tid; i < i_hi; i += kSelBlock) A.alive[i] = 1;
    } else if (!fail) {
        for (int64_t i = i_lo + tid; i < i_hi; i += kSelBlock) {
            const double x = X[i];
            A.alive[i] = (flag ? (x <= eps) : (x < eps)) ? 1 : 0;
        }
    }
    if (blockIdx.x == 0 && tid == 0) {
        A.st->ESS = ESS;
        A.st->resample = resample;
        if (fail) dsel_fail(A, 2);
    }
}

// finish: on a resample, after the all-gather of the segments: the ensemble's compacted index; always:
// the iteration's control block (what smc_select_kernel's workgroup 0 leaves)
__global__ void __launch_bounds__(256) dsel_finish_kernel(const DselArgs A) {
    if (A.ctrl->done || A.st->state != 3) return;
    if (A.st->resample) {
        long long base = 0;
        for (int r = 0; r < A.world; ++r) {
            const long long m = (long long)A.misc[(size_t)r * 8 + 1];
            const int32_t* __restrict__ seg = A.seg + (size_t)r * A.seg_len;
            for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < m;
                 q += (long long)gridDim.x * blockDim.x)
                A.cidx[base + q] = seg[q];
            base += m;
        }
    }
}
__global__ void dsel_publish_kernel(const DselArgs A) {
    if (A.ctrl->done || A.st->state != 3) return;
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const DselState S = *A.st;
        A.misc[(size_t)A.rank * 8 + 1] = 0ull;
        A.ctrl->iteration += 1;
        A.ctrl->eps_prev = A.ctrl->eps;  // ϵv = ϵ
        A.ctrl->eps = S.eps;
        A.ctrl->min_alive = S.mn;
        A.ctrl->ess = S.ESS;
        A.ctrl->n_alive = S.resample ? A.N : S.ESS;
        A.ctrl->flag = S.flag;
        A.ctrl->resampled = S.resample;
        A.ctrl->accepted = 0;
        A.ctrl->passes = 0;
        A.ctrl->pass_open = 1;
        A.ctrl->use_ridx = 1;
    }
}

// ---- the ONE-exchange course -------------------------------------------------------------------
// The phases above need three dependent exchanges (histogram -> bin -> candidates -> counts) because
// the bin of the target rank is only known after the first.  But eps moves by about the same amount
// from one iteration to the next: the keys of a PREDICTED window [eps - 1.4 d, eps - 0.65 d]
// (d = the last decrement) -- a few percent of the alive particles -- are shipped unasked, with the
// count of the alive keys below the window, a 1024-bin histogram of the keys inside and the smallest
// alive key above: ONE all-gather, after which every rank holds everything the selection needs when
// the target rank falls inside the window (the new alive count included: keys below + keys of the
// window below eps).  A resample's compacted index is computed by every rank itself from the costs
// the pass's all-gather has delivered anyway (alive = X < eps for EVERY particle: a dead particle's
// cost is >= the eps that killed it, src/smc.jl:142).  When the prediction fails -- no history in
// the first two iterations, a decrement far off the last, a window too full, eps == 0 -- the
// deciding kernel raises `stalled`, every kernel enqueued behind it is a no-op, and the host, which
// looks once per batch of iterations, repeats that selection with the phases above.
//
//   begin    [pass end of the previous iteration +] the window
//   spec     grid over the rank's alive costs -> its slot of the payload: count, NaNs, key range,
//            keys below / inside / smallest above the window                         [all-gather]
//   decide   grid: target rank, window histogram -> bin, the bin's keys -> the last workgroup ranks
//            them in LDS -> eps, ESS, resample; the iteration's control block
//   apply    grid: the rank's alive mask; on a resample the per-slice counts over ALL particles
//   index    grid, on a resample: idx source list (ascending alive indices of the ensemble)

// slice `bid` of `G` of [lo, hi), whole tiles of 1024 counted from lo
__device__ __forceinline__ void dsel2_slice(int64_t lo, int64_t hi, unsigned bid, unsigned G, int64_t* i_lo,
                                            int64_t* i_hi, int64_t* tile_lo, int64_t* tile_hi) {
    const int64_t len = hi - lo;
    const int64_t ntile = (len + kSelBlock - 1) / kSelBlock;
    const int64_t tpb = (ntile + G - 1) / G;
    int64_t t0 = (int64_t)bid * tpb, t1 = t0 + tpb;
    t0 = t0 < ntile ? t0 : ntile;
    t1 = t1 < ntile ? t1 : ntile;
    const int64_t a = lo + t0 * kSelBlock, b = lo + t1 * kSelBlock;
    *i_lo = a < hi ? a : hi;
    *i_hi = b < hi ? b : hi;
    *tile_lo = t0;
    *tile_hi = t1;
}

// What every kernel of the course tests before it acts, read with independent loads up front: behind a
// kernel boundary each of them is a trip to memory, and `a || b || c` would make them one after the other.
struct Dsel2Head {
    int32_t done, cur, stalled, state, flag, resample;
    unsigned long long wlo, whi;
    double eps;
};
__device__ __forceinline__ Dsel2Head dsel2_head(const DselArgs& A) {
    Dsel2Head h;
    h.done = A.ctrl->done;
    h.cur = A.ctrl->cur;
    h.stalled = A.st->stalled;
    h.state = A.st->state;
    h.flag = A.st->flag;
    h.resample = A.st->resample;
    h.wlo = A.st->wlo;
    h.whi = A.st->whi;
    h.eps = A.st->eps;
    return h;
}

__device__ __forceinline__ int dsel2_shift(uint64_t span) {
    const int bits = span ? 64 - __clzll((long long)span) : 0;
    return bits > 10 ? bits - 10 : 0;
}

__global__ void __launch_bounds__(kSmcSlots) dsel2_begin_kernel(const DselArgs A, const Dsel2End E) {
    __shared__ unsigned long long sh[3][kSmcSlots / kWave];
    __shared__ int s_go;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wid = tid >> 6;
    const int done0 = A.ctrl->done, stalled0 = A.st->stalled, open0 = A.ctrl->pass_open;
    if (done0 | stalled0) return;
    if (E.do_pass_end) {  // smc_pass_end_kernel's work (smc_kernels.hpp), end of the iteration included
        const bool open = open0 != 0;
        if (open) {
            unsigned long long v[3];
            for (int j = 0; j < 3; ++j) {
                unsigned long long t = 0;
                for (int r = 0; r < E.nregions; ++r) {
                    unsigned long long* q = E.slots + ((size_t)r * kSmcSlots + tid) * 8 + j;
                    t += *q;
                    *q = 0;
                }
                v[j] = wave_sum(t);
            }
            if (lane == 0)
                for (int j = 0; j < 3; ++j) sh[j][wid] = v[j];
        }
        __syncthreads();
        if (tid == 0) {
            if (open) {
                unsigned long long t[3] = {0, 0, 0};
                for (int w = 0; w < kSmcSlots / kWave; ++w)
                    for (int j = 0; j < 3; ++j) t[j] += sh[j][w];
                A.ctrl->accepted += t[0];
                A.ctrl->cost_evals += t[1];
                A.ctrl->proposals += t[2];
                A.ctrl->pass += 1;
                A.ctrl->passes += 1;
                A.ctrl->cur ^= 1;
                A.ctrl->use_ridx = 0;
                if ((double)A.ctrl->accepted >= E.mcmc_tol * (double)A.N) A.ctrl->pass_open = 0;
            }
            smc_iter_end(A.ctrl, E.log, E.log_cap, A.N, E.P);
            s_go = A.ctrl->done ? 0 : 1;
        }
        __syncthreads();
        if (!s_go || E.end_only) return;
    }
    // this rank's slot of the payload, cleared for the spec kernel's atomics
    unsigned long long* slot = A.spec + (size_t)A.rank * A.spec_stride;
    for (int w = tid; w < kSelBins / 2; w += kSmcSlots) slot[kDselSpecHead + w] = 0ull;
    if (tid == 0) {
        slot[0] = 0ull;   // alive keys below the window
        slot[1] = 0ull;   // keys inside
        slot[2] = ~0ull;  // smallest alive key above
        slot[3] = 0ull;   // alive
        slot[4] = 0ull;   // NaNs among them
        slot[5] = ~0ull;  // smallest alive key
        slot[6] = ~0ull;  // ~(largest alive key)
        DselState S = {};
        S.state = 6;  // waiting for the payload
        // the window: every key while the whole ensemble fits a workgroup's stage; else around eps - d
        const double e1 = A.ctrl->eps, e0 = A.ctrl->eps_prev, d = e0 - e1;
        if (A.N <= A.spec_cap && A.N <= (int64_t)kDselStage) {
            S.spec = 2;
            S.wlo = 0ull;
            S.whi = ~0ull;
        } else if (kabc_isfinite(e1) && kabc_isfinite(d) && d > 0.0) {
            const uint64_t wlo = key_of(e1 - kDselWinLo * d), whi = key_of(e1 - kDselWinHi * d);
            if (wlo <= whi) {
                S.spec = 1;
                S.wlo = wlo;
                S.whi = whi;
            }
        }
        if (!S.spec) {
            S.stalled = 1;
            S.stall_iteration = A.ctrl->iteration;
        }
        *A.st = S;
    }
}

// spec: this rank's alive costs -> its slot of the payload (the statistics smc_block_stats' partials hold,
// taken here from the costs themselves: a workgroup folding N / 64 partials is slower than the grid reading X)
__global__ void __launch_bounds__(kSelBlock) dsel2_spec_kernel(const DselArgs A) {
    __shared__ unsigned int hist[kSelBins];
    __shared__ uint64_t stage[kDselStage];
    __shared__ unsigned long long s_red[kSelBlock / kWave][6];
    __shared__ unsigned int s_n, s_base;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wid = tid >> 6;
    const Dsel2Head H = dsel2_head(A);
    if (H.done | H.stalled | (H.state != 6)) return;
    const double* __restrict__ X = A.Xbuf[H.cur];
    int64_t i_lo, i_hi, t0, t1;
    dsel2_slice(A.p_lo, A.p_hi, blockIdx.x, gridDim.x, &i_lo, &i_hi, &t0, &t1);
    const uint64_t wlo = H.wlo, whi = H.whi;
    const int shift = dsel2_shift(whi - wlo);
    hist[tid] = 0;
    if (tid == 0) s_n = 0;
    __syncthreads();
    unsigned long long below = 0, cnt = 0, nanc = 0;
    uint64_t kgt = ~0ull, kmin = ~0ull, kmaxn = ~0ull;
    for_each_alive(A.alive, X, i_lo, i_hi, tid, [&](int64_t, double x) {
        const uint64_t k = key_of(x);
        ++cnt;
        if (x != x) ++nanc;
        kmin = k < kmin ? k : kmin;
        kmaxn = ~k < kmaxn ? ~k : kmaxn;
        if (k < wlo) {
            ++below;
        } else if (k <= whi) {
            atomicAdd(&hist[(unsigned)((k - wlo) >> shift)], 1u);
            const unsigned q = atomicAdd(&s_n, 1u);
            if (q < (unsigned)kDselStage) stage[q] = k;
        } else {
            kgt = k < kgt ? k : kgt;
        }
    });
    {
        below = wave_sum(below);
        cnt = wave_sum(cnt);
        nanc = wave_sum(nanc);
        for (int off = kWave / 2; off > 0; off >>= 1) {
            const uint64_t a = __shfl_down(kgt, off, kWave), b = __shfl_down(kmin, off, kWave),
                           c = __shfl_down(kmaxn, off, kWave);
            kgt = a < kgt ? a : kgt;
            kmin = b < kmin ? b : kmin;
            kmaxn = c < kmaxn ? c : kmaxn;
        }
        if (lane == 0) {
            s_red[wid][0] = below;
            s_red[wid][1] = cnt;
            s_red[wid][2] = nanc;
            s_red[wid][3] = kgt;
            s_red[wid][4] = kmin;
            s_red[wid][5] = kmaxn;
        }
        __syncthreads();
    }
    unsigned long long* slot = A.spec + (size_t)A.rank * A.spec_stride;
    const unsigned mine = s_n;
    if (tid == 0) {
        below = cnt = nanc = 0;
        kgt = kmin = kmaxn = ~0ull;
        for (int w = 0; w < kSelBlock / kWave; ++w) {
            below += s_red[w][0];
            cnt += s_red[w][1];
            nanc += s_red[w][2];
            kgt = s_red[w][3] < kgt ? s_red[w][3] : kgt;
            kmin = s_red[w][4] < kmin ? s_red[w][4] : kmin;
            kmaxn = s_red[w][5] < kmaxn ? s_red[w][5] : kmaxn;
        }
        // (a workgroup whose stage overflowed claims more than a slot holds: the deciding kernel sees it)
        const unsigned long long claim = mine <= (unsigned)kDselStage ? mine : (unsigned long long)A.spec_cap + 1ull;
        unsigned long long b0 = 0;
        if (claim) b0 = atomicAdd(&slot[1], claim);  // (the one atomic whose answer is waited for: first)
        if (below) atomicAdd(&slot[0], below);
        if (kgt != ~0ull) atomicMin(&slot[2], (unsigned long long)kgt);
        if (cnt) {
            atomicAdd(&slot[3], cnt);
            atomicMin(&slot[5], (unsigned long long)kmin);
            atomicMin(&slot[6], (unsigned long long)kmaxn);
        }
        if (nanc) atomicAdd(&slot[4], nanc);
        s_base = (!claim || b0 + claim <= (unsigned long long)A.spec_cap) ? (unsigned)b0 : 0xffffffffu;
    }
    __syncthreads();
    if (mine == 0) return;
    {  // two bin counts per word (little end first): the payload is an array of 64-bit words
        unsigned int* h32 = reinterpret_cast<unsigned int*>(slot + kDselSpecHead);
        const unsigned c = hist[tid];
        if (c) atomicAdd(&h32[tid], c);
    }
    const unsigned base = s_base;
    if (base == 0xffffffffu) return;
    for (unsigned q = tid; q < mine; q += kSelBlock) slot[kDselSpecKeys + base + q] = stage[q];
}

// decide: after the all-gather of the payload.  Every workgroup folds the headers and the window's
// histogram (a few KB: the same words in all of them), scans its share of the window's keys for those of
// the target rank's bin -> A.bin; the LAST one to finish (a ticket: no workgroup waits for another)
// ranks them in LDS: eps, ESS, the resample decision, the iteration's control block.
__global__ void __launch_bounds__(kSelBlock) dsel2_decide_kernel(const DselArgs A) {
    __shared__ unsigned int hist[kSelBins];
    __shared__ uint64_t cand[kSelCand];
    __shared__ uint64_t sh_u[kSelBlock / kWave];
    __shared__ long long sh_ll[kSelBlock / kWave];
    __shared__ unsigned int s_wcnt[kSelBlock / kWave];
    __shared__ uint64_t s_klo, s_khi, s_keya, s_keyb, s_kmin;
    __shared__ long long s_kt, s_nrange, s_before, s_nall;
    __shared__ double s_gq;
    __shared__ unsigned int s_ncand;
    __shared__ int s_state, s_needmin, s_stall, s_last;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wid = tid >> 6;
    const Dsel2Head H = dsel2_head(A);
    const long long iteration0 = A.ctrl->iteration;  // (what the last workgroup's thread 0 rewrites at the end)
    const double eps0 = A.ctrl->eps;
    // the window's histogram, rank by rank (requested before anything is waited for)
    unsigned c = 0;
    for (int r = 0; r < A.world; ++r)
        c += reinterpret_cast<const unsigned int*>(A.spec + (size_t)r * A.spec_stride + kDselSpecHead)[tid];
    if (H.done | H.stalled | (H.state != 6)) return;
    const uint64_t wlo = H.wlo, whi = H.whi;
    const int shift0 = dsel2_shift(whi - wlo);
    const bool lead = blockIdx.x == 0 && tid == 0;
    // headers: one rank per lane of wave 0 (world <= 64)
    if (wid == 0) {
        unsigned long long below = 0, nc = 0, kg = ~0ull, na = 0, nn = 0, kmin = ~0ull, kmaxn = ~0ull;
        int over = 0;
        if (lane < A.world) {
            const unsigned long long* slot = A.spec + (size_t)lane * A.spec_stride;
            below = slot[0];
            nc = slot[1];
            kg = slot[2];
            na = slot[3];
            nn = slot[4];
            kmin = slot[5];
            kmaxn = slot[6];
            over = nc > (unsigned long long)A.spec_cap ? 1 : 0;
        }
        below = wave_sum(below);
        const unsigned long long ncs = wave_sum(over ? 0ull : nc);
        na = wave_sum(na);
        nn = wave_sum(nn);
        over = __any(over) ? 1 : 0;
        for (int off = kWave / 2; off > 0; off >>= 1) {
            const unsigned long long o = __shfl_down(kg, off, kWave), p = __shfl_down(kmin, off, kWave),
                                     q = __shfl_down(kmaxn, off, kWave);
            kg = o < kg ? o : kg;
            kmin = p < kmin ? p : kmin;
            kmaxn = q < kmaxn ? q : kmaxn;
        }
        if (lane == 0) {
            const long long n = (long long)na;
            int stall = 0;
            long long kt = 0;
            double gq = 0.0;
            if (n == 0 || nn > 0) {
                stall = -((nn > 0) ? 1 : 2);  // an error of the run, not of the course
            } else {
                // ranks of the two bracketing order statistics (Statistics.quantile, type 7)
                const double aleph = (double)n * A.alpha + (1.0 - A.alpha);
                long long j = (long long)aleph;
                if (j < 1) j = 1;
                if (j > n - 1) j = n - 1;
                if (n == 1) j = 1;
                gq = aleph - (double)j;
                gq = gq < 0.0 ? 0.0 : (gq > 1.0 ? 1.0 : gq);
                kt = j - 1 - (long long)below;
                if (over) stall = 2;                                          // a slot too full
                else if (kt < 0 || kt >= (long long)ncs) stall = 3;            // the target rank outside the window
                else if (n > 1 && kt + 1 >= (long long)ncs && kg == ~0ull) stall = 4;  // (cannot happen: rank j exists)
            }
            s_stall = stall;
            s_state = -1;
            s_keya = 0;
            s_needmin = 0;
            s_kt = kt;
            s_gq = gq;
            s_nall = n;
            s_kmin = kmin;
            s_before = (long long)below;
            s_nrange = (long long)ncs;
            s_keyb = kg;  // smallest alive key above the window
        }
    }
    __syncthreads();
    if (s_stall) {
        if (lead) {
            if (s_stall < 0) {
                A.st->n = s_nall;
                dsel_fail(A, -s_stall);
            } else {
                A.st->stalled = s_stall;
                A.st->stall_iteration = iteration0;
            }
        }
        return;
    }
    const uint64_t kg_above = s_keyb;
    const long long n = s_nall;
    // the window's histogram -> the bin of the target rank
    unsigned incl = c;
    for (int off = 1; off < kWave; off <<= 1) {
        const unsigned o = __shfl_up(incl, off, kWave);
        if (lane >= off) incl += o;
    }
    if (lane == kWave - 1) s_wcnt[wid] = incl;
    __syncthreads();
    {
        unsigned woff = 0;
        for (int w = 0; w < wid; ++w) woff += s_wcnt[w];
        const long long before = (long long)woff + incl - c, kt = s_kt;
        __syncthreads();
        if (c > 0 && kt >= before && kt < before + (long long)c) {  // exactly one thread
            const uint64_t nlo = wlo + ((uint64_t)tid << shift0);
            uint64_t nhi = nlo + ((1ull << shift0) - 1ull);
            if (nhi > whi || nhi < nlo) nhi = whi;
            s_klo = nlo;
            s_khi = nhi;
            s_kt = kt - before;
            s_before += before;
            s_nrange = c;
            s_state = (shift0 == 0) ? 2 : (c <= (unsigned)kWave ? 3 : 0);
            s_stall = c > (unsigned)kSelCand ? 5 : 0;  // the bin too full for the LDS list
            s_keyb = 0;
        }
    }
    __syncthreads();
    if (s_stall || s_state < 0) {
        if (lead) {
            A.st->stalled = s_stall ? s_stall : 6;
            A.st->stall_iteration = iteration0;
        }
        return;
    }
    // this workgroup's share of the window's keys: the bin's into A.bin, the smallest above the bin
    unsigned long long* bin = A.bin;  // [0] cursor, [1] smallest key above the bin, [2] ticket, [8 ...] keys
    {
        uint64_t kgt = ~0ull;
        const uint64_t blo = s_klo, bhi = s_khi;
        const unsigned G = gridDim.x;
        for (int r = 0; r < A.world; ++r) {
            const unsigned long long* slot = A.spec + (size_t)r * A.spec_stride;
            const unsigned m = (unsigned)slot[1];
            const unsigned long long* keys = slot + kDselSpecKeys;
            for (unsigned q0 = blockIdx.x * kSelBlock + tid; q0 < m; q0 += 4u * G * kSelBlock) {
                uint64_t kk[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const unsigned q = q0 + (unsigned)u * G * kSelBlock;
                    kk[u] = q < m ? keys[q] : 0ull;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const unsigned q = q0 + (unsigned)u * G * kSelBlock;
                    if (q >= m) continue;
                    const uint64_t k = kk[u];
                    if (k > bhi) {
                        kgt = k < kgt ? k : kgt;
                    } else if (k >= blo) {
                        const unsigned long long pos = atomicAdd(&bin[0], 1ull);  // <= nrange <= kSelCand
                        if (pos < (unsigned long long)kSelCand)  // (device-coherent store: no L2 write-back needed)
                            __hip_atomic_store(&bin[8 + pos], k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
        }
        // every writer waits for its stores (they are performed at the device's coherence point), the
        // workgroup meets, then the ticket is drawn: the bin's keys are final for whoever draws the last one
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        kgt = block_min_u64(kgt, sh_u);
        if (tid == 0) {
            if (kgt != ~0ull) atomicMin(&bin[1], (unsigned long long)kgt);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            s_last = (atomicAdd(&bin[2], 1ull) == (unsigned long long)(G - 1u)) ? 1 : 0;
        }
        __syncthreads();
        if (!s_last) return;
    }
    // the last workgroup: the bin's keys into LDS, the slate clean for the next selection
    const unsigned long long nc64 = __hip_atomic_load(&bin[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned nc = nc64 < (unsigned long long)kSelCand ? (unsigned)nc64 : (unsigned)kSelCand;  // == s_nrange
    uint64_t kgt = __hip_atomic_load(&bin[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    kgt = kg_above < kgt ? kg_above : kgt;
    for (unsigned q = tid; q < nc; q += kSelBlock)
        cand[q] = __hip_atomic_load(&bin[8 + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (tid == 0) {
        bin[0] = 0ull;
        bin[1] = ~0ull;
        bin[2] = 0ull;
    }
    // rank inside LDS: dsel_rank_kernel's narrowing, then one key per lane
    for (int round = 0; round < 12 && s_state == 0; ++round) {
        const uint64_t klo = s_klo, khi = s_khi;
        const uint64_t span = khi - klo;
        if (span == 0) {
            if (tid == 0) s_state = 2;
            __syncthreads();
            break;
        }
        const int shift = dsel2_shift(span);
        hist[tid] = 0;
        __syncthreads();
        for (unsigned i = tid; i < nc; i += kSelBlock) {
            const uint64_t k = cand[i];
            if (k >= klo && k <= khi) atomicAdd(&hist[(unsigned)((k - klo) >> shift)], 1u);
        }
        __syncthreads();
        const unsigned cc = hist[tid];
        unsigned inc2 = cc;
        for (int off2 = 1; off2 < kWave; off2 <<= 1) {
            const unsigned o = __shfl_up(inc2, off2, kWave);
            if (lane >= off2) inc2 += o;
        }
        if (lane == kWave - 1) s_wcnt[wid] = inc2;
        __syncthreads();
        unsigned woff = 0;
        for (int w = 0; w < wid; ++w) woff += s_wcnt[w];
        const long long before = (long long)woff + inc2 - cc, kt = s_kt;
        __syncthreads();
        if (cc > 0 && kt >= before && kt < before + (long long)cc) {
            const uint64_t nlo = klo + ((uint64_t)tid << shift);
            uint64_t nhi = nlo + ((1ull << shift) - 1ull);
            if (nhi > khi || nhi < nlo) nhi = khi;
            s_klo = nlo;
            s_khi = nhi;
            s_kt = kt - before;
            s_nrange = cc;
            s_state = (shift == 0) ? 2 : (cc <= (unsigned)kWave ? 3 : 0);
        }
        __syncthreads();
    }
    if (s_state == 3) {
        if (tid == 0) s_ncand = 0;
        __syncthreads();
        const uint64_t klo = s_klo, khi = s_khi;
        for (unsigned i = tid; i < nc; i += kSelBlock) {
            const uint64_t k = cand[i];
            if (k >= klo && k <= khi) {
                const unsigned pos = atomicAdd(&s_ncand, 1u);
                reinterpret_cast<uint64_t*>(hist)[pos] = k;  // hist is free now: 64 x u64
            }
        }
        __syncthreads();
        if (wid == 0) {
            const unsigned m = s_ncand;
            const uint64_t mine = (lane < (int)m) ? reinterpret_cast<uint64_t*>(hist)[lane] : ~0ull;
            unsigned rank = 0;
            for (unsigned q = 0; q < m; ++q) {
                const uint64_t other = __shfl(mine, (int)q, kWave);
                rank += (other < mine || (other == mine && q < (unsigned)lane)) ? 1u : 0u;
            }
            const long long kt = s_kt;
            if (lane < (int)m && rank == (unsigned)kt) s_keya = mine;
            if (lane < (int)m && rank == (unsigned)kt + 1u) s_keyb = mine;
            if (lane == 0) s_needmin = (kt + 1 < (long long)m) ? 0 : 1;
        }
        __syncthreads();
    } else if (s_state == 2) {
        if (tid == 0) {  // every key of the range equals klo
            s_keya = s_klo;
            s_needmin = (s_kt + 1 < s_nrange) ? 0 : 1;
            s_keyb = s_klo;
        }
        __syncthreads();
    } else {
        if (tid == 0) {
            A.st->stalled = 6;  // (the LDS narrowing did not end: cannot happen)
            A.st->stall_iteration = iteration0;
        }
        return;
    }
    if (s_needmin && n > 1) {
        // rank j is the smallest alive key above the final range: among the bin's keys, else among the
        // window's keys above the bin / the ranks' smallest keys above the window
        const uint64_t khi = s_khi;
        uint64_t k2 = kgt;
        for (unsigned q = tid; q < nc; q += kSelBlock) {
            const uint64_t k = cand[q];
            if (k > khi) k2 = k < k2 ? k : k2;
        }
        k2 = block_min_u64(k2, sh_u);
        if (tid == 0) s_keyb = k2;
        __syncthreads();
    }
    // eps (src/smc.jl:134-141), then the new alive count: keys below the bin + the bin's keys below eps
    const double gq = s_gq;
    const double a = val_of(s_keya);
    const double b = (n == 1) ? a : val_of(s_keyb);
    double eps;
    if (kabc_isfinite(a) && kabc_isfinite(b)) eps = a + gq * (b - a);
    else eps = (1.0 - gq) * a + gq * b;
    const double mn = val_of(s_kmin);  // minimum(Xs[alive])
    const int flag = (eps > mn) ? 0 : 1;
    long long inbin = 0;
    for (unsigned q = tid; q < nc; q += kSelBlock) {
        const double x = val_of(cand[q]);
        inbin += (flag ? (x <= eps) : (x < eps)) ? 1 : 0;
    }
    inbin = block_sum_ll(inbin, sh_ll);
    if (tid == 0) {
        if (eps == 0.0 || (n > 1 && s_needmin && s_keyb == ~0ull)) {  // (+-0 are two keys and one value: phase by phase)
            A.st->stalled = 7;
            A.st->stall_iteration = iteration0;
            return;
        }
        const long long ESS = s_before + inbin;
        const int resample = (A.alpha * (double)ESS <= (double)A.N * A.min_r_ess) ? 1 : 0;
        DselState S = {};
        S.wlo = wlo;
        S.whi = whi;
        S.spec = 1;
        S.n = n;
        S.gq = gq;
        S.mn = mn;
        S.klo = s_klo;
        S.khi = s_khi;
        S.kt = s_kt;
        S.nrange = s_nrange;
        S.keya = s_keya;
        S.keyb = s_keyb;
        S.needmin = s_needmin;
        S.listed = 1;
        S.ncand_all = nc;
        S.eps = eps;
        S.flag = flag;
        S.ESS = ESS;
        S.resample = resample;
        S.state = 3;
        *A.st = S;
        if (resample && ESS == 0) {
            dsel_fail(A, 2);
            return;
        }
        A.ctrl->iteration = iteration0 + 1;
        A.ctrl->eps_prev = eps0;
        A.ctrl->eps = eps;
        A.ctrl->min_alive = mn;
        A.ctrl->ess = ESS;
        A.ctrl->n_alive = resample ? A.N : ESS;
        A.ctrl->flag = flag;
        A.ctrl->resampled = resample;
        A.ctrl->accepted = 0;
        A.ctrl->passes = 0;
        A.ctrl->pass_open = 1;
        A.ctrl->use_ridx = 1;
    }
}

// apply: the rank's alive mask; on a resample the alive count of every slice of the WHOLE ensemble
__global__ void __launch_bounds__(kSelBlock) dsel2_apply_kernel(const DselArgs A) {
    __shared__ long long sh_ll[kSelBlock / kWave];
    const int tid = threadIdx.x;
    const Dsel2Head H = dsel2_head(A);
    if (H.done | H.stalled | (H.state != 3)) return;
    const double* __restrict__ X = A.Xbuf[H.cur];
    const double eps = H.eps;
    const int flag = H.flag;
    int64_t i_lo, i_hi, t0, t1;
    if (!H.resample) {
        dsel2_slice(A.p_lo, A.p_hi, blockIdx.x, gridDim.x, &i_lo, &i_hi, &t0, &t1);
        for (int64_t i = i_lo + tid; i < i_hi; i += kSelBlock) {
            const double x = X[i];
            A.alive[i] = (flag ? (x <= eps) : (x < eps)) ? 1 : 0;
        }
        return;
    }
    dsel2_slice(0, A.N, blockIdx.x, gridDim.x, &i_lo, &i_hi, &t0, &t1);
    long long mycnt = 0;
    for (int64_t i = i_lo + tid; i < i_hi; i += kSelBlock) {
        const double x = X[i];
        mycnt += (flag ? (x <= eps) : (x < eps)) ? 1 : 0;
        A.alive[i] = 1;  // alive = trues(nparticles) (:147); every rank writes the whole mask
    }
    mycnt = block_sum_ll(mycnt, sh_ll);
    if (tid == 0) A.sub_cnt[blockIdx.x] = (unsigned)mycnt;
}

// index: on a resample, the ascending alive indices of the whole ensemble (what dsel_compact_kernel +
// the all-gather of the segments + dsel_finish_kernel leave), by every rank for itself
__global__ void __launch_bounds__(kSelBlock) dsel2_index_kernel(const DselArgs A) {
    __shared__ unsigned int s_cnt4[4 * (kSelBlock / kWave)];
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wid = tid >> 6;
    __shared__ long long s_base0;
    const Dsel2Head H = dsel2_head(A);
    // the slices before this one (<= kDsel2MaxGrid = 512 counts: eight per lane of wave 0)
    unsigned long long mine = 0;
    if (wid == 0) {
#pragma unroll
        for (int q = 0; q < kDsel2MaxGrid / kWave; ++q)
            if ((unsigned)lane + (unsigned)q * kWave < blockIdx.x) mine += A.sub_cnt[lane + q * kWave];
    }
    if (H.done | H.stalled | (H.state != 3) | !H.resample) return;
    const double* __restrict__ X = A.Xbuf[H.cur];
    int64_t i_lo, i_hi, tile_lo, tile_hi;
    dsel2_slice(0, A.N, blockIdx.x, gridDim.x, &i_lo, &i_hi, &tile_lo, &tile_hi);
    const double eps = H.eps;
    const int flag = H.flag;
    if (wid == 0) {
        mine = wave_sum(mine);
        if (lane == 0) s_base0 = (long long)mine;
    }
    __syncthreads();
    long long base = s_base0;
    for (int64_t tile0 = tile_lo; tile0 < tile_hi; tile0 += 4) {
        bool al[4];
        unsigned long long bm[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t i = (tile0 + u) * kSelBlock + tid;
            double x = 0.0;
            const bool in = (tile0 + u < tile_hi) && i < A.N;
            if (in) x = X[i];
            al[u] = in && (flag ? (x <= eps) : (x < eps));
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            bm[u] = __ballot(al[u]);
            if (lane == 0) s_cnt4[u * (kSelBlock / kWave) + wid] = (unsigned)__popcll(bm[u]);
        }
        __syncthreads();
        const unsigned c = s_cnt4[lane];
        unsigned incl = c;
        for (int off = 1; off < kWave; off <<= 1) {
            const unsigned o = __shfl_up(incl, off, kWave);
            if (lane >= off) incl += o;
        }
        const unsigned excl = incl - c;
        const unsigned tot = (unsigned)__shfl((int)incl, kWave - 1, kWave);
        const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned woff = (unsigned)__shfl((int)excl, u * (kSelBlock / kWave) + wid, kWave);
            if (al[u]) A.cidx[base + woff + __popcll(bm[u] & below)] = (int32_t)((tile0 + u) * kSelBlock + tid);
        }
        base += tot;
        __syncthreads();
    }
}
#endif  // KABC_SMC_SINGLE_UNIT

}  // namespace kabc
