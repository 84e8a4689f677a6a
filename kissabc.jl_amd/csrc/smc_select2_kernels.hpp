// smc_select2_kernels.hpp -- the ε-selection of smc / pfilter for LARGE ensembles as ordinary
// kernels over all CUs, one per dependent phase, with NO device-wide barrier inside a kernel:
//
//   src/smc.jl:134-153 (ε = quantile(Xs[alive], α), alive mask, ESS, resample decision, idxalive)
//   src/smc.jl:298-301 (pfilter: ϵ = quantile(C, q), ok = !(C > ϵ), idxok)          [mode 1]
//
// smc_select_kernel (smc_kernels.hpp) does all of it in ONE launch of at most 128 workgroups that
// meet at five hand-rolled device-wide barriers: at 2 M particles its passes stream at the rate of
// the few CUs it occupies and a barrier costs 5 us (83 us per call), and the barriers need every
// workgroup co-resident -- an assumption the launch cannot assert.  A kernel boundary costs 3.6 us
// on this chip (profiles/r03_halfgen_floor.json) and asks nothing of the scheduler, so from 2^17
// particles on the phases are kernels:
//
//   sel2_partials  (G)   only when the producer of X left no per-workgroup statistics (pfilter)
//   sel2_stats     (1)   n, NaN check, key range of the alive costs, the two target ranks, the
//                        histogram window; zeroes the scratch of this call
//   sel2_hist      (G)   4096-bin histogram of the alive keys over the window (LDS, folded with
//                        one global atomic per occupied bin)
//   sel2_collect   (G)   every workgroup finds the bin of the target rank (redundantly), its
//                        particles append that bin's keys to the candidate list; the smallest key
//                        above the bin is kept beside (rank j + 1 when the bin ends at rank j)
//   sel2_finish    (G)   every workgroup ranks the candidates in LDS: ε exactly (type-7
//                        interpolation, Inf-safe, as the one-kernel select); mask + count of its
//                        slice; the slices' offsets by a decoupled look-back (a workgroup waits only
//                        for LOWER block ids, which were dispatched before it and wait for nobody
//                        above them); ordered compaction of the alive indices; the LAST workgroup
//                        to finish adds up ESS, decides the resample and publishes the control block.
//
// Results are the one-kernel select's bit for bit (same keys, same ranks, same interpolation).  One
// difference in the hand-over: after a resample every particle is alive, which the one-kernel
// select writes into `alive` itself; here the mask cannot be overwritten before ESS is known, so
// the propose / accept kernel that follows (smc_mcmc_kernel, smc_dyn_kernel) treats every particle
// as alive when ctrl->resampled is set and writes the ones back.
//
// A candidate bin with more keys than kSel2Cand (heavy ties around ε over a wide range) sets
// ctrl->error = 5: the host repeats the run with the one-kernel select (every draw is
// counter-based: the repetition is the same run).
#pragma once

#include "smc_kernels.hpp"

namespace kabc {

constexpr int kSel2Bins = 4096;
constexpr int kSel2Cand = 8192;    // candidate keys ranked in LDS (64 KB)
constexpr int kSel2MaxG = 1024;    // workgroups of the G-wide kernels
constexpr int kSel2Block = 1024;

struct SmcSel2Scratch {
    // -- written by sel2_stats
    long long n;                 // alive particles
    unsigned long long kmin, kmax;
    unsigned long long klo, khi; // histogram window
    long long kt;                // 0-based target rank inside the window
    double gq, mn;               // interpolation weight, minimum(Xs[alive])
    int32_t shift;
    int32_t state;               // 0 histogram + candidates, 2 every alive key equals klo, -1 nothing to do (error / done)
    // -- written by sel2_collect (block 0) / all blocks
    unsigned long long bin_lo, bin_hi;
    long long bin_kt, bin_n;
    unsigned long long kgt;      // smallest alive key above bin_hi (atomicMin), ~0 if none
    unsigned int ncand;
    unsigned int done_count;     // sel2_finish: workgroups that have finished
    unsigned int pad[5];
    unsigned int hist[kSel2Bins];
    unsigned int slice_cnt[kSel2MaxG];   // sel2_finish: count | 0x80000000 once published
    unsigned long long wg_part[kSel2MaxG][4];  // sel2_partials
    unsigned long long cand[kSel2Cand];
};

struct SmcSel2Args {
    SmcSelectArgs s;      // the arguments of the one-kernel select (same meaning)
    SmcSel2Scratch* g;
};

#ifdef KABC_SMC_SINGLE_UNIT

__device__ __forceinline__ void sel2_slice(int64_t N, unsigned G, unsigned bid, int64_t& i_lo, int64_t& i_hi,
                                           int64_t& tile_lo, int64_t& tile_hi) {
    const int64_t ntile = (N + kSel2Block - 1) / kSel2Block;
    const int64_t tpb = (ntile + G - 1) / G;
    tile_lo = (int64_t)bid * tpb;
    tile_hi = tile_lo + tpb < ntile ? tile_lo + tpb : ntile;
    if (tile_lo > ntile) tile_lo = ntile;
    i_lo = tile_lo * kSel2Block;
    i_hi = tile_hi * kSel2Block;
    i_lo = i_lo < N ? i_lo : N;
    i_hi = i_hi < N ? i_hi : N;
}

// (count, NaNs, min key, ~max key) of the alive costs of each workgroup's slice
__global__ void __launch_bounds__(kSel2Block) sel2_partials_kernel(const SmcSel2Args A) {
    __shared__ unsigned long long s_red[kSel2Block / kWave][4];
    if (A.s.ctrl->done) return;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wid = tid >> 6;
    const double* __restrict__ X = A.s.Xbuf[A.s.ctrl->cur];
    int64_t i_lo, i_hi, t0, t1;
    sel2_slice(A.s.N, gridDim.x, blockIdx.x, i_lo, i_hi, t0, t1);
    long long cnt = 0, nanc = 0;
    uint64_t kmin = ~0ull, kmaxn = ~0ull;
    for_each_alive(A.s.alive, X, i_lo, i_hi, tid, [&](int64_t, double x) {
        ++cnt;
        if (x != x) ++nanc;
        const uint64_t k = key_of(x);
        kmin = k < kmin ? k : kmin;
        kmaxn = ~k < kmaxn ? ~k : kmaxn;
    });
    const unsigned long long wc = wave_sum((unsigned long long)cnt), wn = wave_sum((unsigned long long)nanc);
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
    if (tid == 0) {
        unsigned long long n = 0, nn = 0, lo = ~0ull, hin = ~0ull;
        for (int w = 0; w < kSel2Block / kWave; ++w) {
            n += s_red[w][0];
            nn += s_red[w][1];
            lo = s_red[w][2] < lo ? s_red[w][2] : lo;
            hin = s_red[w][3] < hin ? s_red[w][3] : hin;
        }
        unsigned long long* p = A.g->wg_part[blockIdx.x];
        p[0] = n;
        p[1] = nn;
        p[2] = lo;
        p[3] = hin;
    }
}

// one workgroup: the statistics of the alive costs, the target ranks, the histogram window
__global__ void __launch_bounds__(kSel2Block) sel2_stats_kernel(const SmcSel2Args A, unsigned G) {
    __shared__ unsigned long long s_red[kSel2Block / kWave][4];
    SmcSel2Scratch* __restrict__ g = A.g;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wid = tid >> 6;
    // the scratch of this call
    for (int b = tid; b < kSel2Bins; b += kSel2Block) g->hist[b] = 0;
    for (int b = tid; b < kSel2MaxG; b += kSel2Block) g->slice_cnt[b] = 0;
    if (tid == 0) {
        g->ncand = 0;
        g->done_count = 0;
        g->kgt = ~0ull;
        g->state = -1;
    }
    if (A.s.ctrl->done) return;
    const unsigned long long* part = A.s.part ? A.s.part : &g->wg_part[0][0];
    const int64_t npart = A.s.part ? A.s.npart : (int64_t)G;
    long long cnt = 0, nanc = 0;
    uint64_t kmin = ~0ull, kmaxn = ~0ull;
    for (int64_t b = tid; b < npart; b += kSel2Block) {
        const unsigned long long* p = part + (size_t)b * 4;
        cnt += (long long)p[0];
        nanc += (long long)p[1];
        kmin = p[2] < kmin ? p[2] : kmin;
        kmaxn = p[3] < kmaxn ? p[3] : kmaxn;
    }
    const unsigned long long wc = wave_sum((unsigned long long)cnt), wn = wave_sum((unsigned long long)nanc);
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
    if (tid != 0) return;
    long long n = 0, nn = 0;
    kmin = kmaxn = ~0ull;
    for (int w = 0; w < kSel2Block / kWave; ++w) {
        n += (long long)s_red[w][0];
        nn += (long long)s_red[w][1];
        kmin = s_red[w][2] < kmin ? s_red[w][2] : kmin;
        kmaxn = s_red[w][3] < kmaxn ? s_red[w][3] : kmaxn;
    }
    const uint64_t kmax = ~kmaxn;
    if (n == 0 || nn > 0) {
        A.s.ctrl->error = (nn > 0) ? 1 : 2;
        A.s.ctrl->done = 1;
        return;
    }
    // ranks of the two bracketing order statistics (Statistics.quantile, type 7): as smc_select_kernel (b)
    const double aleph = (double)n * A.s.alpha + (1.0 - A.s.alpha);
    long long j = (long long)aleph;
    if (j < 1) j = 1;
    if (j > n - 1) j = n - 1;
    if (n == 1) j = 1;
    double gq = aleph - (double)j;
    gq = gq < 0.0 ? 0.0 : (gq > 1.0 ? 1.0 : gq);
    const uint64_t span = kmax - kmin;
    const int bits = span ? 64 - __clzll((long long)span) : 0;
    g->n = n;
    g->kmin = kmin;
    g->kmax = kmax;
    g->klo = kmin;
    g->khi = kmax;
    g->kt = j - 1;
    g->gq = gq;
    g->mn = val_of(kmin);
    g->shift = bits > 12 ? bits - 12 : 0;
    g->state = (kmin == kmax) ? 2 : 0;
}

__global__ void __launch_bounds__(kSel2Block) sel2_hist_kernel(const SmcSel2Args A) {
    __shared__ unsigned int hist[kSel2Bins];
    const SmcSel2Scratch* __restrict__ g = A.g;
    if (g->state != 0) return;  // uniform
    const int tid = threadIdx.x;
    const double* __restrict__ X = A.s.Xbuf[A.s.ctrl->cur];
    int64_t i_lo, i_hi, t0, t1;
    sel2_slice(A.s.N, gridDim.x, blockIdx.x, i_lo, i_hi, t0, t1);
    const uint64_t klo = g->klo, khi = g->khi;
    const int shift = g->shift;
    for (int b = tid; b < kSel2Bins; b += kSel2Block) hist[b] = 0;
    __syncthreads();
    for_each_alive(A.s.alive, X, i_lo, i_hi, tid, [&](int64_t, double x) {
        const uint64_t k = key_of(x);
        if (k >= klo && k <= khi) atomicAdd(&hist[(unsigned)((k - klo) >> shift)], 1u);
    });
    __syncthreads();
    for (int b = tid; b < kSel2Bins; b += kSel2Block) {
        const unsigned c = hist[b];
        if (c) atomicAdd(&A.g->hist[b], c);
    }
}

__global__ void __launch_bounds__(kSel2Block) sel2_collect_kernel(const SmcSel2Args A) {
    __shared__ unsigned int s_wsum[kSel2Block / kWave];
    __shared__ uint64_t s_lo, s_hi, sh_u[kSel2Block / kWave];
    __shared__ long long s_kt, s_n;
    __shared__ unsigned int s_cnt, s_base;
    __shared__ uint64_t s_keys[kSel2Block * 4];  // this workgroup's keys of the bin (one tile round)
    SmcSel2Scratch* __restrict__ g = A.g;
    if (g->state != 0) return;  // uniform
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wid = tid >> 6;
    const uint64_t klo = g->klo, khi = g->khi;
    const int shift = g->shift;
    const long long kt = g->kt;
    // the bin holding rank kt: every thread owns four consecutive bins
    constexpr int kPer = kSel2Bins / kSel2Block;
    unsigned c[kPer], tot = 0;
#pragma unroll
    for (int q = 0; q < kPer; ++q) {
        c[q] = g->hist[tid * kPer + q];
        tot += c[q];
    }
    unsigned incl = tot;
    for (int off = 1; off < kWave; off <<= 1) {
        const unsigned o = __shfl_up(incl, off, kWave);
        if (lane >= off) incl += o;
    }
    if (lane == kWave - 1) s_wsum[wid] = incl;
    __syncthreads();
    unsigned woff = 0;
    for (int w = 0; w < wid; ++w) woff += s_wsum[w];
    long long before = (long long)woff + incl - tot;
#pragma unroll
    for (int q = 0; q < kPer; ++q) {
        if (c[q] > 0 && kt >= before && kt < before + (long long)c[q]) {  // exactly one (thread, q)
            const uint64_t nlo = klo + ((uint64_t)(tid * kPer + q) << shift);
            uint64_t nhi = nlo + ((1ull << shift) - 1ull);
            if (nhi > khi || nhi < nlo) nhi = khi;
            s_lo = nlo;
            s_hi = nhi;
            s_kt = kt - before;
            s_n = c[q];
        }
        before += c[q];
    }
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    const uint64_t blo = s_lo, bhi = s_hi;
    if (s_n > (long long)kSel2Cand) {  // uniform over the grid
        if (blockIdx.x == 0 && tid == 0) {
            A.s.ctrl->error = 5;
            A.s.ctrl->done = 1;
        }
        return;
    }
    if (blockIdx.x == 0 && tid == 0) {
        g->bin_lo = blo;
        g->bin_hi = bhi;
        g->bin_kt = s_kt;
        g->bin_n = s_n;
    }
    const double* __restrict__ X = A.s.Xbuf[A.s.ctrl->cur];
    int64_t i_lo, i_hi, t0, t1;
    sel2_slice(A.s.N, gridDim.x, blockIdx.x, i_lo, i_hi, t0, t1);
    uint64_t kgt = ~0ull;
    // the slice in rounds of 4 tiles: the bin's keys of a round are gathered in LDS and appended to
    // the global list with one atomic per workgroup and round
    for (int64_t r_lo = i_lo; r_lo < i_hi; r_lo += 4 * kSel2Block) {
        const int64_t r_hi = r_lo + 4 * kSel2Block < i_hi ? r_lo + 4 * kSel2Block : i_hi;
        for_each_alive(A.s.alive, X, r_lo, r_hi, tid, [&](int64_t, double x) {
            const uint64_t k = key_of(x);
            if (k >= blo && k <= bhi) s_keys[atomicAdd(&s_cnt, 1u)] = k;
            else if (k > bhi) kgt = k < kgt ? k : kgt;
        });
        __syncthreads();
        const unsigned mine = s_cnt;
        if (tid == 0) s_base = mine ? atomicAdd(&g->ncand, mine) : 0u;
        __syncthreads();
        for (unsigned q = tid; q < mine; q += kSel2Block)
            if (s_base + q < (unsigned)kSel2Cand) g->cand[s_base + q] = s_keys[q];
        __syncthreads();
        if (tid == 0) s_cnt = 0;
        __syncthreads();
    }
    kgt = block_min_u64(kgt, sh_u);
    if (tid == 0 && kgt != ~0ull) atomicMin(&g->kgt, kgt);
}

// keys of rank kt and kt + 1 (0-based) among cand[0..nc) in LDS, by narrowing on 1024-bin
// histograms down to <= 64 keys ranked inside one wavefront (the list phase of smc_select_kernel);
// *needmin: rank kt + 1 lies beyond the list.  Called by all kSel2Block threads.
__device__ __forceinline__ void sel2_rank_two(const uint64_t* cand, unsigned nc, long long kt0, uint64_t lo0,
                                              uint64_t hi0, unsigned* hist /*[1024] u32 = [512] u64*/,
                                              unsigned* s_wcnt, uint64_t* s_out /*[6]*/) {
    // s_out: 0 klo, 1 khi, 2 kt, 3 state (0 narrowing, 2 all equal, 3 <= 64 left), 4 keya, 5 keyb / ~0 (needmin)
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wid = tid >> 6;
    if (tid == 0) {
        s_out[0] = lo0;
        s_out[1] = hi0;
        s_out[2] = (uint64_t)kt0;
        s_out[3] = (nc <= (unsigned)kWave) ? 3u : 0u;
        s_out[4] = s_out[5] = ~0ull;
    }
    __syncthreads();
    long long nrange = nc;
    for (int round = 0; round < 12 && s_out[3] == 0; ++round) {
        const uint64_t klo = s_out[0], khi = s_out[1];
        const uint64_t span = khi - klo;
        if (span == 0) {
            __syncthreads();
            if (tid == 0) s_out[3] = 2;
            __syncthreads();
            break;
        }
        const int bits = 64 - __clzll((long long)span);
        const int shift = bits > 10 ? bits - 10 : 0;
        for (int b = tid; b < 1024; b += kSel2Block) hist[b] = 0;
        __syncthreads();
        for (unsigned i = tid; i < nc; i += kSel2Block) {
            const uint64_t k = cand[i];
            if (k >= klo && k <= khi) atomicAdd(&hist[(unsigned)((k - klo) >> shift)], 1u);
        }
        __syncthreads();
        const unsigned c = hist[tid];
        unsigned incl = c;
        for (int off = 1; off < kWave; off <<= 1) {
            const unsigned o = __shfl_up(incl, off, kWave);
            if (lane >= off) incl += o;
        }
        if (lane == kWave - 1) s_wcnt[wid] = incl;
        __syncthreads();
        unsigned woff = 0;
        for (int w = 0; w < wid; ++w) woff += s_wcnt[w];
        const long long before = (long long)woff + incl - c, kt = (long long)s_out[2];
        __syncthreads();
        if (c > 0 && kt >= before && kt < before + (long long)c) {
            const uint64_t nlo = klo + ((uint64_t)tid << shift);
            uint64_t nhi = nlo + ((1ull << shift) - 1ull);
            if (nhi > khi || nhi < nlo) nhi = khi;
            s_out[0] = nlo;
            s_out[1] = nhi;
            s_out[2] = (uint64_t)(kt - before);
            s_out[3] = (shift == 0) ? 2u : (c <= (unsigned)kWave ? 3u : 0u);
            s_wcnt[kSel2Block / kWave] = c;  // (the range's population, for the all-equal case)
        }
        __syncthreads();
        nrange = -1;
    }
    __syncthreads();
    const uint64_t klo = s_out[0], khi = s_out[1];
    const long long kt = (long long)s_out[2];
    if (s_out[3] == 3) {
        unsigned* s_n = &s_wcnt[kSel2Block / kWave + 1];
        if (tid == 0) *s_n = 0;
        __syncthreads();
        uint64_t* few = reinterpret_cast<uint64_t*>(hist);
        for (unsigned i = tid; i < nc; i += kSel2Block) {
            const uint64_t k = cand[i];
            if (k >= klo && k <= khi) few[atomicAdd(s_n, 1u)] = k;
        }
        __syncthreads();
        if (wid == 0) {
            const unsigned m = *s_n;
            const uint64_t mine = (lane < (int)m) ? few[lane] : ~0ull;
            unsigned rank = 0;
            for (unsigned q = 0; q < m; ++q) {
                const uint64_t other = __shfl(mine, (int)q, kWave);
                rank += (other < mine || (other == mine && q < (unsigned)lane)) ? 1u : 0u;
            }
            if (lane < (int)m && rank == (unsigned)kt) s_out[4] = mine;
            if (lane < (int)m && rank == (unsigned)kt + 1u) s_out[5] = mine;
        }
        __syncthreads();
    } else {  // state 2: every key of [klo, khi] equals klo
        if (tid == 0) {
            const long long pop = (nrange >= 0) ? nrange : (long long)s_wcnt[kSel2Block / kWave];
            s_out[4] = klo;
            s_out[5] = (kt + 1 < pop) ? klo : ~0ull;
        }
        __syncthreads();
    }
    // rank kt + 1 beyond the final range but inside the list: the smallest listed key above khi
    if (s_out[5] == ~0ull) {
        uint64_t kgt = ~0ull;
        for (unsigned i = tid; i < nc; i += kSel2Block) {
            const uint64_t k = cand[i];
            if (k > khi) kgt = k < kgt ? k : kgt;
        }
        kgt = block_min_u64(kgt, reinterpret_cast<uint64_t*>(hist) + 64);
        if (tid == 0) s_out[5] = kgt;
        __syncthreads();
    }
}

__global__ void __launch_bounds__(kSel2Block) sel2_finish_kernel(const SmcSel2Args A) {
    __shared__ uint64_t cand[kSel2Cand];
    __shared__ unsigned int hist[1024];
    __shared__ unsigned int s_wcnt[kSel2Block / kWave + 2];
    __shared__ uint64_t s_out[6];
    __shared__ long long sh_ll[kSel2Block / kWave];
    __shared__ unsigned int s_cnt4[4 * (kSel2Block / kWave)];
    __shared__ double s_eps;
    __shared__ int s_flag, s_last;
    SmcSel2Scratch* __restrict__ g = A.g;
    const int state = g->state;
    if (state < 0) return;                         // uniform: error / done (sel2_stats)
    if (A.s.ctrl->error == 5) return;              // uniform: candidate list overflow (sel2_collect)
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wid = tid >> 6;
    const unsigned G = gridDim.x, bid = blockIdx.x;
    const int64_t N = A.s.N;
    const long long n = g->n;
    const double* __restrict__ X = A.s.Xbuf[A.s.ctrl->cur];
    // -- ε: the keys of ranks j - 1 and j
    uint64_t keya, keyb;
    if (state == 2) {
        keya = keyb = g->klo;  // every alive key is the same (n == 1: keyb is not used)
    } else {
        const unsigned nc = g->ncand;  // == bin_n
        for (unsigned q = tid; q < nc; q += kSel2Block) cand[q] = g->cand[q];
        __syncthreads();
        sel2_rank_two(cand, nc, g->bin_kt, g->bin_lo, g->bin_hi, hist, s_wcnt, s_out);
        keya = s_out[4];
        keyb = s_out[5];
        if (keyb == ~0ull) keyb = g->kgt;  // rank j is the smallest alive key above the bin
    }
    if (tid == 0) {
        const double a = val_of(keya);
        const double b = (n == 1) ? a : val_of(keyb);
        const double gq = g->gq;
        double eps;
        if (kabc_isfinite(a) && kabc_isfinite(b)) eps = a + gq * (b - a);
        else eps = (1.0 - gq) * a + gq * b;
        s_eps = eps;
        s_flag = (A.s.mode == 1) ? 1 : ((eps > g->mn) ? 0 : 1);  // src/smc.jl:135-141
    }
    __syncthreads();
    const double eps = s_eps;
    const int flag = s_flag;
    // -- mask + count of this workgroup's slice, published for the workgroups above
    int64_t i_lo, i_hi, tile_lo, tile_hi;
    sel2_slice(N, G, bid, i_lo, i_hi, tile_lo, tile_hi);
    long long mycnt = 0;
    for (int64_t i = i_lo + tid; i < i_hi; i += kSel2Block) {
        const double x = X[i];
        mycnt += (flag ? (x <= eps) : (x < eps)) ? 1 : 0;
    }
    mycnt = block_sum_ll(mycnt, sh_ll);
    if (tid == 0) {
        __threadfence();
        atomicExch(&g->slice_cnt[bid], (unsigned)mycnt | 0x80000000u);
    }
    // -- offset of the slice: the counts of the lower block ids (they were dispatched first and
    //    publish before they wait: no cycle)
    long long below = 0;
    for (unsigned b = tid; b < bid; b += kSel2Block) {
        volatile unsigned* p = &g->slice_cnt[b];
        unsigned v;
        while (!((v = *p) & 0x80000000u)) __builtin_amdgcn_s_sleep(1);
        below += (long long)(v & 0x7fffffffu);
    }
    long long base = block_sum_ll(below, sh_ll);
    // -- ordered compaction of the alive indices + the mask (tiles of 1024, four per round: as
    //    smc_select_kernel (d))
    for (int64_t tile0 = tile_lo; tile0 < tile_hi; tile0 += 4) {
        bool al[4];
        unsigned long long bm[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t i = (tile0 + u) * kSel2Block + tid;
            double x = 0.0;
            const bool in = (tile0 + u < tile_hi) && i < N;
            if (in) x = X[i];
            al[u] = in && (flag ? (x <= eps) : (x < eps));
            if (in) A.s.alive_out[i] = al[u] ? 1 : 0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            bm[u] = __ballot(al[u]);
            if (lane == 0) s_cnt4[u * (kSel2Block / kWave) + wid] = (unsigned)__popcll(bm[u]);
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
        const unsigned long long lower = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned woff = (unsigned)__shfl((int)excl, u * (kSel2Block / kWave) + wid, kWave);
            if (al[u])
                A.s.cidx[base + woff + __popcll(bm[u] & lower)] = (int32_t)((tile0 + u) * kSel2Block + tid);
        }
        base += tot;
        __syncthreads();
    }
    // -- the last workgroup to finish publishes the iteration's control block
    __syncthreads();
    if (tid == 0) {
        __threadfence();
        s_last = (atomicAdd(&g->done_count, 1u) == G - 1u) ? 1 : 0;
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    long long ess = 0;
    for (unsigned b = tid; b < G; b += kSel2Block)
        ess += (long long)(*(volatile unsigned*)&g->slice_cnt[b] & 0x7fffffffu);
    const long long ESS = block_sum_ll(ess, sh_ll);
    if (tid != 0) return;
    // Step 2 decision: α*ESS <= nparticles*min_r_ess  (src/smc.jl:145); after a resample every
    // particle is alive -- the propose / accept kernel writes that into `alive` (see the head)
    const int resample = (A.s.mode == 0 && A.s.alpha * (double)ESS <= (double)N * A.s.min_r_ess) ? 1 : 0;
    SmcCtrl* ctrl = A.s.ctrl;
    if (resample && ESS == 0) {
        ctrl->error = 2;
        ctrl->done = 1;
        return;
    }
    ctrl->iteration += 1;
    ctrl->eps_prev = ctrl->eps;  // ϵv = ϵ
    ctrl->eps = eps;
    ctrl->min_alive = g->mn;
    ctrl->ess = ESS;
    ctrl->n_alive = resample ? N : ESS;
    ctrl->flag = flag;
    ctrl->resampled = resample;
    ctrl->accepted = 0;
    ctrl->passes = 0;
    ctrl->pass_open = 1;
    ctrl->use_ridx = 1;
}

#endif  // KABC_SMC_SINGLE_UNIT

}  // namespace kabc
