// smc_dsel_kernels.hpp -- the ε-selection of smc() (src/smc.jl:131-153) with the PARTICLES sharded
// over the ranks of a communicator (SURVEY §8e "SMC"; kabc_smc_run_dist with KABC_SMC_DIST_PARTICLES).
//
// Rank r owns the particles [p_lo, p_hi) (whole workgroups of 64 of the propose/accept kernel).  It
// holds the alive mask of its own range only and makes every pass of the selection over its own
// costs; what the ranks have to agree on travels in four small all-gathers (every rank then folds the
// world's contributions itself, in rank order: all ranks compute the same words, nothing is broadcast):
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

struct DselState {  // device; identical on every rank after every deciding kernel
    unsigned long long klo, khi, keya, keyb;
    long long kt, nrange, n;
    double gq, mn, eps;
    long long ESS;
    int32_t state;  // 0 narrowing, 1 collect and rank, 2 every key of the range equal, 3 keys known
    int32_t listed, needmin, need_scan, flag, resample, error, rounds;
    uint32_t ncand_all, pad;
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
    unsigned int* sub_cnt;     // [kDselMaxGrid] new alive count per workgroup slice
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
#endif  // KABC_SMC_SINGLE_UNIT

}  // namespace kabc
