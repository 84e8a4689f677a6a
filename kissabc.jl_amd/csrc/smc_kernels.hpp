// smc_kernels.hpp -- gfx950 kernels for smc(prior, cost; ...) (src/smc.jl:92-206).
//
//   smc_init_kernel    : :119-125  prior draws, cost, logprior for every particle
//   smc_select_kernel  : :134-153  ε = quantile(Xs[alive], α) by radix select on the
//                        order-preserving u64 image of the costs (two adjacent order
//                        statistics, type-7 interpolation), alive mask, ESS, resample
//                        decision and the cyclic-replication index
//                        repeat(idxalive, ceil(N/m))[1:N]  (index-exact)
//   smc_mcmc_kernel    : :160-191  proposal from the frozen ensemble + prior-MH +
//                        cost threshold; the resample gather is fused in through the
//                        index array, the ensemble is double-buffered
//   smc_finalize_kernel: :200      push_p of the final positions
#pragma once

#include "kabc_device.hpp"
#ifndef __HIPCC_RTC__
#include "launcher.hpp"
#endif

namespace kabc {

// Device-resident loop state of smc().  The ε-iteration is controlled ON THE DEVICE
// (stop tests, retry break, buffer flip) so that the host can enqueue several
// iterations without reading anything back; kernels of an iteration that is
// already over are no-ops.
struct SmcCtrl {
    double eps;                     // ϵ      (src/smc.jl:134)
    double eps_prev;                // ϵv     (:133)
    double min_alive;               // minimum(Xs[alive]) of the last select
    long long ess;                  // sum(alive) before resampling
    long long n_alive;              // sum(alive) after step 2
    long long iteration;
    int32_t flag;
    int32_t resampled;
    int32_t error;                  // 1 NaN cost among alive, 2 no alive particle
    int32_t done;                   // outer loop has terminated (:194-198) or failed
    int32_t cur;                    // buffer set holding the current ensemble
    int32_t use_ridx;               // next MCMC pass gathers through ridx (first of an iteration)
    int32_t pass_open;              // retry passes of this iteration may still run (:192)
    int32_t passes;                 // passes executed in this iteration
    unsigned long long pass;        // global pass counter = transition counter of the streams
    unsigned long long accepted;    // accumulated in this iteration
    unsigned long long cost_evals;  // cumulative
    unsigned long long proposals;   // cumulative
};

struct SmcLoopParams {
    double mcmc_tol, epstol, r_epstol;
    long long max_iterations;
};

struct SmcInitArgs {
    double* theta;
    double* X;
    double* lpi;
    uint8_t* alive;
    SmcCtrl* ctrl;
    const double* cost_params;
    const double* cost_data;
    int64_t cost_ndata;
    int64_t N;
    uint64_t seed;
    int32_t cost_id;
    PriorSet prior;
    kabc_prior_t raw[KABC_MAX_DIM];
    unsigned long long* part;  // [workgroups][4] cost statistics of the block, see smc_block_stats
    // cost loop sharded over ranks (kabc_smc_run_dist): this launch covers the workgroups
    // [wg0, wg0 + nwg) of the ensemble (64 particles each) when `sharded` is set, else all
    int64_t wg0, nwg;
    int32_t sharded;
};

constexpr int kSelBins = 1024;      // histogram bins per narrowing round
constexpr int kSelCand = 4096;      // candidate keys narrowed in LDS
constexpr int kSelRounds = 7;       // 7 x 10 bits > 64: global narrowing rounds at most
constexpr int kSelMaxBlocks = 128;  // workgroups of the select kernel (capi_smc.hip select_blocks)

// global scratch of the multi-workgroup select kernel: zeroed once by the host, left
// zeroed by every call (barrier words excepted: the generation only ever counts up)
struct SmcSelScratch {
    unsigned int bar_count, pad0[31];
    unsigned int bar_gen, pad1[31];
    unsigned int bar_abort, pad3[31];
    unsigned int ncand, pad2[31];
    unsigned int hist[kSelRounds][kSelBins];
    unsigned long long part_stats[kSelMaxBlocks][4];
    unsigned long long part_kgt[kSelMaxBlocks];
    unsigned int slice_cnt[kSelMaxBlocks];
    unsigned long long cand[kSelCand];
};

struct SmcSelectArgs {
    const double* Xbuf[2];
    uint8_t* alive;
    int32_t* ridx;   // unused (the resample index is cidx[j mod ESS], see smc_mcmc_kernel)
    int32_t* cidx;   // scratch: compacted alive indices
    SmcCtrl* ctrl;
    int64_t N;
    double alpha;
    double min_r_ess;
    unsigned long long* stamps;  // diagnostic (KABC_SMC_STAMPS): per-phase s_memtime sums [8]
    // mode 1 = pfilter (src/smc.jl:298-301): ϵ = quantile(C, q) over ALL particles (the
    // `alive` input is all ones), ok mask = !(C > ϵ) written to alive_out, idxok to
    // cidx, count to ctrl->ess; no resampling, no ridx.  mode 0: alive_out == alive.
    int32_t mode;
    uint8_t* alive_out;
    // per-workgroup (count, NaNs, min key, ~max key) of the alive costs, written by the
    // kernel that produced X (smc_init_kernel / smc_mcmc_kernel); NULL = scan X here
    const unsigned long long* part;
    int64_t npart;
    SmcSelScratch* scratch;  // used when the kernel runs with more than one workgroup
    // how long a workgroup waits at a device-wide barrier before the kernel gives up (ticks of
    // s_memrealtime, 100 MHz): 0.2 s for the ordinary launch (capi_smc.hip launch_select: the run is
    // then repeated with a cooperative launch, whose co-residency the runtime asserts), 5 s for that one
    unsigned long long barrier_timeout;
};

struct SmcMcmcArgs {
    double* theta[2];     // double-buffered ensemble; ctrl->cur selects the source
    double* X[2];
    double* lpi[2];
    const uint8_t* alive;
    const int32_t* cidx;  // compacted alive indices of the last select: after a resample the
                          // first pass of the iteration reads particle j from row cidx[j mod ESS]
    SmcCtrl* ctrl;
    unsigned long long* slots;  // [kSmcSlots][8]: accepted, cost_evals, proposals
    const double* cost_params;
    const double* cost_data;
    int64_t cost_ndata;
    int64_t N;
    uint64_t seed;
    double max_stretch;
    PriorSet prior;
    unsigned long long* part;  // [workgroups][4], see smc_block_stats
    int64_t wg0, nwg;          // sharded cost loop: the workgroups of this launch ...
    int32_t sharded;           // ... when set, else all
    // prepared cost words of this pass for every particle, [W][N] (ais_aux_kernels.hpp), or NULL;
    // aux_ring > 1: [aux_ring][W][N], pass t in slot t mod aux_ring (AuxArgs::ring)
    const double* aux;
    int32_t aux_ring;
};

struct SmcFinalArgs {
    const double* theta[2];
    const double* X[2];
    const SmcCtrl* ctrl;
    double* out;
    double* Xout;
    int64_t N;
    int32_t D;
    PriorSet prior;
    const PriorDev* dprior;  // [D] on the device when D > KABC_MAX_DIM (else NULL: `prior`)
};

constexpr int kSmcBlock = 64;
constexpr int kSmcSlots = 256;
constexpr int kSelBlock = 1024;

// order-preserving map double -> u64 (total order with -0 < +0, NaNs at the ends)
__device__ __forceinline__ uint64_t key_of(double x) {
    const uint64_t u = kabc_bits(x);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ULL);
}
__device__ __forceinline__ double val_of(uint64_t k) {
    const uint64_t u = (k >> 63) ? (k & 0x7fffffffffffffffULL) : ~k;
    return kabc_from_bits(u);
}

// The kernels that produce X leave what the next ε-selection needs to know about it --
// count, NaN count and key range of the ALIVE costs of their 64 particles -- so that the
// single-workgroup select kernel reads N/64 partials instead of scanning X once more
// (that scan was 12 of its 54 us at C4).  Called by all 64 threads of the workgroup.
__device__ __forceinline__ void smc_block_stats(unsigned long long* part, bool alive, double x,
                                                int64_t wg = -1) {
    if (!part) return;
    if (wg < 0) wg = (int64_t)blockIdx.x;
    const uint64_t k = key_of(x);
    unsigned long long cnt = wave_sum(alive ? 1ull : 0ull);
    unsigned long long nan = wave_sum((alive && x != x) ? 1ull : 0ull);
    uint64_t kmin = alive ? k : ~0ull, kmaxn = alive ? ~k : ~0ull;
    for (int off = kWave / 2; off > 0; off >>= 1) {
        const uint64_t a = __shfl_down(kmin, off, kWave), b = __shfl_down(kmaxn, off, kWave);
        kmin = a < kmin ? a : kmin;
        kmaxn = b < kmaxn ? b : kmaxn;
    }
    if ((threadIdx.x & (kWave - 1)) == 0) {
        unsigned long long* p = part + (size_t)wg * 4;
        p[0] = cnt;
        p[1] = nan;
        p[2] = kmin;
        p[3] = kmaxn;
    }
}

template <int D>
__global__ void __launch_bounds__(kSmcBlock) smc_init_kernel(const SmcInitArgs A) {
    const int64_t wg = A.wg0 + (int64_t)blockIdx.x;
    const int64_t i = wg * kSmcBlock + threadIdx.x;
    double c = 0.0;
    if (i < A.N) {
        double x[D], xp[D];
        for (int k = 0; k < D; ++k) {
            kabc_slotwin_t win = {A.seed, 0ull, (uint32_t)i, KABC_DOM_SMC_INIT,
                                  (uint32_t)k * KABC_SLOTS_PER_DIM};
            x[k] = kabc_sample_prior(&A.raw[k], &win);
        }
        const double lp = factored_logpdf_push<D>(A.prior, x, xp);
        kabc_cost_rng_t rng = {A.seed, 0ull, (uint32_t)i, KABC_DOM_SMC_INIT_COST, 0u};
        c = kabc_cost_eval(A.cost_id, xp, D, A.cost_params, A.cost_data, A.cost_ndata, &rng);
        store_row<D>(A.theta + i * D, x);
        A.X[i] = c;
        A.lpi[i] = lp;
        A.alive[i] = 1;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {  // (every rank of a sharded run keeps its own ctrl)
        SmcCtrl cc = {};
        cc.eps = KABC_INF;       // ϵ = Inf  (src/smc.jl:127)
        cc.eps_prev = KABC_INF;
        cc.cost_evals = (unsigned long long)A.N;
        *A.ctrl = cc;
    }
    smc_block_stats(A.part, i < A.N, c, wg);
}

#ifdef KABC_SMC_SINGLE_UNIT  // non-template kernels: defined once, in capi_smc.hip
// block-wide helpers for the single-workgroup select kernel ------------------
__device__ __forceinline__ long long block_sum_ll(long long v, long long* sh) {
    v = (long long)wave_sum((unsigned long long)v);
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) sh[wid] = v;
    __syncthreads();
    long long t = 0;
    for (int w = 0; w < kSelBlock / kWave; ++w) t += sh[w];
    return t;
}
__device__ __forceinline__ uint64_t block_min_u64(uint64_t v, uint64_t* sh) {
    for (int off = kWave / 2; off > 0; off >>= 1) {
        const uint64_t o = __shfl_down(v, off, kWave);
        v = o < v ? o : v;
    }
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) sh[wid] = v;
    __syncthreads();
    uint64_t t = ~0ull;
    for (int w = 0; w < kSelBlock / kWave; ++w) t = sh[w] < t ? sh[w] : t;
    return t;
}

// visit (i, key) of every alive particle of [i_lo, i_hi); four independent (alive, X)
// load pairs are issued per thread and iteration so that memory latency overlaps
template <class F>
__device__ __forceinline__ void for_each_alive(const uint8_t* __restrict__ alive,
                                               const double* __restrict__ X, int64_t i_lo,
                                               int64_t i_hi, int tid, F&& f) {
    constexpr int U = 4;
    for (int64_t i0 = i_lo + tid; i0 < i_hi; i0 += (int64_t)U * kSelBlock) {
        uint8_t al[U];
        double xv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = i0 + (int64_t)u * kSelBlock;
            al[u] = (i < i_hi) ? alive[i] : (uint8_t)0;
            xv[u] = (i < i_hi) ? X[i] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (al[u]) f(i0 + (int64_t)u * kSelBlock, xv[u]);
    }
}

// Device-wide barrier of the select kernel's G workgroups (cooperative launch: all of
// them are resident).  Sense-reversing: one atomic per workgroup, thread 0 spins on the
// generation word; 2.4 us at 32 x 1024 threads, about a kernel boundary (cooperative
// groups' grid.sync() is 5.1 us; tools/gridsync_probe.hip).  G == 1: __syncthreads.
// Residency: launch_select (capi_smc.hip) launches a grid that fits the device (G clamped to a
// quarter of occupancy x CUs) on an in-order stream, or cooperatively (KABC_SMC_COOPERATIVE=1, and
// whenever an ordinary launch has timed out).  The spin is bounded either way (A.barrier_timeout):
// on time-out the abort word is set and every workgroup leaves the kernel; after an ordinary launch
// kabc_smc_run / kabc_pfilter_run repeat the run with cooperative launches -- the same run, every
// draw is counter-based -- and only a cooperative launch that times out is KABC_ERR_DEVICE.
__device__ __forceinline__ bool sel_grid_barrier(SmcSelScratch* g, unsigned G, unsigned long long timeout) {
    __shared__ int s_ok;
    // every wavefront's own stores are in the L2 before thread 0 releases for the workgroup
    // (__syncthreads() waits for LDS / scalar traffic only; smc_loop_kernel.hpp loop_sync_stores_done)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (G > 1u) {
        if (threadIdx.x == 0) {
            int ok = 1;
            volatile unsigned* gen = &g->bar_gen;
            volatile unsigned* ab = &g->bar_abort;
            const unsigned my = *gen;
            __threadfence();
            if (atomicAdd(&g->bar_count, 1u) == G - 1u) {
                atomicExch(&g->bar_count, 0u);
                __threadfence();
                atomicAdd(&g->bar_gen, 1u);
            } else {
                const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                unsigned spins = 0;
                while (*gen == my) {
                    __builtin_amdgcn_s_sleep(1);
                    if ((++spins & 1023u) == 0u &&
                        (*ab || __builtin_amdgcn_s_memrealtime() - t0 > timeout)) {
                        atomicExch(&g->bar_abort, 1u);
                        ok = 0;
                        break;
                    }
                }
            }
            if (*ab) ok = 0;
            __threadfence();
            s_ok = ok;
        }
        __syncthreads();
        return s_ok != 0;
    }
    return true;
}
// every use: a timed-out barrier ends the kernel (uniformly: all workgroups see the abort word)
#define KABC_SEL_BARRIER(g, G)                                   \
    if (!sel_grid_barrier(g, G, A.barrier_timeout)) {            \
        if (blockIdx.x == 0 && threadIdx.x == 0) {               \
            A.ctrl->error = 3;                                   \
            A.ctrl->done = 1;                                    \
        }                                                        \
        return;                                                  \
    }

// ε-selection, alive mask, ESS, resample decision and index -- G workgroups of 1024
// (G = 1 for small N).  Every pass over the particles is split by contiguous slices;
// everything that follows a pass (bin search, candidate ranking, ε) is computed
// redundantly and identically by every workgroup from global scratch, so nothing has
// to be broadcast and all workgroups take the same branches (and barriers).
//
// quantile: the two bracketing order statistics of the alive costs are found by
// NARROWING on the order-preserving u64 keys: a 1024-bin histogram of
// (key - klo) >> shift over the current key range [klo, khi] (bins are spread over
// the range actually occupied, so atomics do not pile onto one bin the way a fixed
// leading-byte radix pass does), then the bin holding the target rank becomes the new
// range, until at most 4096 candidates remain; those are narrowed further in LDS and
// the last <= 64 ranked inside one wavefront.  Typically: 1 histogram pass + 1 collect
// pass + the compaction passes, 4 device-wide barriers.
__global__ void __launch_bounds__(kSelBlock) smc_select_kernel(const SmcSelectArgs A) {
    __shared__ unsigned int hist[kSelBins];
    __shared__ uint64_t cand[kSelCand];
    __shared__ long long sh_ll[kSelBlock / kWave];
    __shared__ uint64_t sh_u[kSelBlock / kWave];
    __shared__ unsigned int s_wcnt[kSelBlock / kWave];
    __shared__ uint64_t s_klo, s_khi, s_keya, s_keyb;
    __shared__ long long s_kt, s_nrange;
    __shared__ unsigned int s_ncand, s_ncand_all, s_base;
    __shared__ int s_listed;  // cand[0..s_ncand_all) holds every alive key of one first-round bin
    __shared__ int s_state;   // 0 narrowing, 1 collect+sort, 2 a known (all keys of range equal)
    __shared__ double s_eps;
    __shared__ int s_flag, s_needmin;

    const int tid = threadIdx.x, lane = tid & (kWave - 1), wid = tid >> 6;
    const unsigned G = gridDim.x, bid = blockIdx.x;
    SmcSelScratch* __restrict__ g = A.scratch;
    const int64_t N = A.N;
    if (A.ctrl->done) return;  // uniform: the loop ended in an earlier iteration
    unsigned long long t_prev = (A.stamps && bid == 0) ? __builtin_amdgcn_s_memtime() : 0ull;
#define KABC_STAMP(slot)                                                   \
    if (A.stamps && bid == 0 && tid == 0) {                                \
        const unsigned long long t_now = __builtin_amdgcn_s_memtime();     \
        A.stamps[slot] += t_now - t_prev;                                  \
        t_prev = t_now;                                                    \
    }
    const double* __restrict__ X = A.Xbuf[A.ctrl->cur];
    // this workgroup's contiguous slice (whole tiles of 1024)
    const int64_t ntile = (N + kSelBlock - 1) / kSelBlock;
    const int64_t tpb = (ntile + G - 1) / G;
    int64_t i_lo = (int64_t)bid * tpb * kSelBlock, i_hi = i_lo + tpb * kSelBlock;
    i_lo = i_lo < N ? i_lo : N;
    i_hi = i_hi < N ? i_hi : N;

    // (a) n = count(alive), NaN check, key range of the alive costs: from the producers'
    //     per-workgroup partials when there are any, else by scanning X
    long long cnt = 0, nanc = 0;
    uint64_t kmin = ~0ull, kmaxn = ~0ull;  // kmaxn = ~max
    if (A.part) {
        for (int64_t b = tid; b < A.npart; b += kSelBlock) {
            const unsigned long long* p = A.part + (size_t)b * 4;
            cnt += (long long)p[0];
            nanc += (long long)p[1];
            kmin = p[2] < kmin ? p[2] : kmin;
            kmaxn = p[3] < kmaxn ? p[3] : kmaxn;
        }
    } else {
        for_each_alive(A.alive, X, i_lo, i_hi, tid, [&](int64_t, double x) {
            ++cnt;
            if (x != x) ++nanc;
            const uint64_t k = key_of(x);
            kmin = k < kmin ? k : kmin;
            kmaxn = ~k < kmaxn ? ~k : kmaxn;
        });
    }
    // one combined workgroup reduction (four separate ones cost eight barriers)
    __shared__ unsigned long long s_red[kSelBlock / kWave][4];
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
    long long n = 0, nn = 0;
    kmin = kmaxn = ~0ull;
    for (int w = 0; w < kSelBlock / kWave; ++w) {
        n += (long long)s_red[w][0];
        nn += (long long)s_red[w][1];
        kmin = s_red[w][2] < kmin ? s_red[w][2] : kmin;
        kmaxn = s_red[w][3] < kmaxn ? s_red[w][3] : kmaxn;
    }
    if (!A.part && G > 1u) {
        if (tid == 0) {
            unsigned long long* p = g->part_stats[bid];
            p[0] = (unsigned long long)n;
            p[1] = (unsigned long long)nn;
            p[2] = kmin;
            p[3] = kmaxn;
        }
        KABC_SEL_BARRIER(g, G)
        n = nn = 0;
        kmin = kmaxn = ~0ull;
        for (unsigned b = 0; b < G; ++b) {  // G <= 128: every thread reads them all
            const unsigned long long* p = g->part_stats[b];
            n += (long long)p[0];
            nn += (long long)p[1];
            kmin = p[2] < kmin ? p[2] : kmin;
            kmaxn = p[3] < kmaxn ? p[3] : kmaxn;
        }
    }
    const uint64_t kmax = ~kmaxn;
    if (n == 0 || nn > 0) {
        if (bid == 0 && tid == 0) {
            A.ctrl->error = (nn > 0) ? 1 : 2;
            A.ctrl->done = 1;
        }
        return;
    }
    const double mn = val_of(kmin);  // minimum(Xs[alive])
    KABC_STAMP(0)

    // (b) ranks of the two bracketing order statistics (Statistics.quantile, type 7)
    const double aleph = (double)n * A.alpha + (1.0 - A.alpha);
    long long j = (long long)aleph;
    if (j < 1) j = 1;
    if (j > n - 1) j = n - 1;
    if (n == 1) j = 1;
    double gq = aleph - (double)j;
    gq = gq < 0.0 ? 0.0 : (gq > 1.0 ? 1.0 : gq);

    // (c) narrowing for the key of rank j-1 (0-based) inside [klo, khi]
    if (tid == 0) {
        s_klo = kmin;
        s_khi = kmax;
        s_kt = j - 1;
        s_nrange = n;
        s_state = (kmin == kmax) ? 2 : (n <= kSelCand ? 1 : 0);
        s_listed = 0;
        s_ncand_all = 0;
    }
    __syncthreads();
    int rounds_used = 0;
    for (int round = 0; round < kSelRounds && s_state == 0; ++round) {
        const uint64_t klo = s_klo, khi = s_khi;
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
        unsigned c = hist[tid];
        if (G > 1u) {  // fold the workgroups' histograms in global scratch (pre-zeroed)
            if (c) atomicAdd(&g->hist[round][tid], c);
            KABC_SEL_BARRIER(g, G)
            c = g->hist[round][tid];
        }
        rounds_used = round + 1;
        // parallel search of the bin holding rank kt: inclusive scan of 1024 bins
        unsigned incl = c;
        for (int off = 1; off < kWave; off <<= 1) {
            const unsigned o = __shfl_up(incl, off, kWave);
            if (lane >= off) incl += o;
        }
        if (lane == kWave - 1) s_wcnt[wid] = incl;
        __syncthreads();
        unsigned woff = 0;
        for (int w = 0; w < wid; ++w) woff += s_wcnt[w];
        const long long before = (long long)woff + incl - c, kt = s_kt;
        __syncthreads();
        if (c > 0 && kt >= before && kt < before + (long long)c) {  // exactly one thread
            const uint64_t nlo = klo + ((uint64_t)tid << shift);
            uint64_t nhi = nlo + ((1ull << shift) - 1ull);
            if (nhi > khi || nhi < nlo) nhi = khi;
            s_klo = nlo;
            s_khi = nhi;
            s_kt = kt - before;
            s_nrange = c;
            s_state = (shift == 0) ? 2 : (c <= (unsigned)kSelCand ? 1 : 0);
        }
        __syncthreads();
    }
    KABC_STAMP(1)
    if (s_state == 0) {  // cannot happen: 7 rounds x 10 bits > 64 bits
        if (bid == 0 && tid == 0) {
            A.ctrl->error = 2;
            A.ctrl->done = 1;
        }
        return;
    }
    if (s_state == 1) {
        // collect the <= 4096 keys of the range (every workgroup ends up with all of them
        // in LDS), keep narrowing ON THE LIST (4 keys per thread per round) until <= 64
        // keys remain, then rank those inside one wavefront.
        if (tid == 0) s_ncand = 0;
        __syncthreads();
        {
            const uint64_t klo = s_klo, khi = s_khi;
            for_each_alive(A.alive, X, i_lo, i_hi, tid, [&](int64_t, double x) {
                const uint64_t k = key_of(x);
                if (k >= klo && k <= khi) cand[atomicAdd(&s_ncand, 1u)] = k;
            });
        }
        __syncthreads();
        if (G > 1u) {
            const unsigned mine = s_ncand;
            if (tid == 0) s_base = mine ? atomicAdd(&g->ncand, mine) : 0u;
            __syncthreads();
            for (unsigned q = tid; q < mine; q += kSelBlock) g->cand[s_base + q] = cand[q];
            KABC_SEL_BARRIER(g, G)
            const unsigned all = g->ncand;  // == s_nrange <= kSelCand
            for (unsigned q = tid; q < all; q += kSelBlock) cand[q] = g->cand[q];
            if (tid == 0) s_ncand = all;
            __syncthreads();
        }
        const unsigned nc = s_ncand;
        if (tid == 0) {
            s_state = (s_nrange <= kWave) ? 3 : 0;
            s_listed = 1;
            s_ncand_all = nc;
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
            for (int off = 1; off < kWave; off <<= 1) {
                const unsigned o = __shfl_up(incl, off, kWave);
                if (lane >= off) incl += o;
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
            // <= 64 keys left in [klo, khi]: wave 0 gathers them (one per lane) and each
            // lane counts how many precede it -> ranks without sorting
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
    KABC_STAMP(2)
    if (s_state == 2 && tid == 0) {  // every key of the range equals klo
        s_keya = s_klo;
        s_needmin = (s_kt + 1 < s_nrange) ? 0 : 1;
        s_keyb = s_needmin ? ~0ull : s_klo;
    }
    __syncthreads();
    if (s_needmin && n > 1) {
        // rank j is the smallest alive key above the final range.  It is nearly always
        // among the candidates already in LDS (they cover the whole bin the first round
        // selected); only when the final range ends that bin is X scanned again.
        const uint64_t khi = s_khi;
        uint64_t kgt = ~0ull;
        if (s_listed) {
            for (unsigned q = tid; q < s_ncand_all; q += kSelBlock) {
                const uint64_t k = cand[q];
                if (k > khi) kgt = k < kgt ? k : kgt;
            }
            kgt = block_min_u64(kgt, sh_u);
        }
        if (kgt == ~0ull) {  // uniform over the grid: kgt is the same in every workgroup
            for_each_alive(A.alive, X, i_lo, i_hi, tid, [&](int64_t, double x) {
                const uint64_t k = key_of(x);
                if (k > khi) kgt = k < kgt ? k : kgt;
            });
            kgt = block_min_u64(kgt, sh_u);
            if (G > 1u) {
                if (tid == 0) g->part_kgt[bid] = kgt;
                KABC_SEL_BARRIER(g, G)
                kgt = ~0ull;
                for (unsigned b = 0; b < G; ++b) kgt = g->part_kgt[b] < kgt ? g->part_kgt[b] : kgt;
            }
        }
        if (tid == 0) s_keyb = kgt;
    }
    __syncthreads();
    if (tid == 0) {
        const double a = val_of(s_keya);
        const double b = (n == 1) ? a : val_of(s_keyb);
        double eps;
        if (kabc_isfinite(a) && kabc_isfinite(b)) eps = a + gq * (b - a);
        else eps = (1.0 - gq) * a + gq * b;
        s_eps = eps;
        s_flag = (A.mode == 1) ? 1 : ((eps > mn) ? 0 : 1);  // src/smc.jl:135-141
    }
    __syncthreads();
    const double eps = s_eps;
    const int flag = s_flag;
    KABC_STAMP(3)

    // (d) new alive mask over ALL particles, ESS, compaction of the alive indices in
    //     ascending order.  First the count of this workgroup's slice (its offset is the
    //     sum of the slices before it), then coalesced tiles of 1024 with ballot + mbcnt,
    //     four tiles per round: the 4 x 16 per-wave counts form one 64-entry vector
    //     that every wave scans with shuffles (tile-major = ascending index).
    long long mycnt = 0;
    for (int64_t i = i_lo + tid; i < i_hi; i += kSelBlock) {
        const double x = X[i];
        mycnt += (flag ? (x <= eps) : (x < eps)) ? 1 : 0;
    }
    mycnt = block_sum_ll(mycnt, sh_ll);
    long long base = 0, ESS = mycnt;
    if (G > 1u) {
        if (tid == 0) g->slice_cnt[bid] = (unsigned)mycnt;
        KABC_SEL_BARRIER(g, G)
        ESS = 0;
        for (unsigned b = 0; b < G; ++b) {
            const long long cb = (long long)g->slice_cnt[b];
            if (b < bid) base += cb;
            ESS += cb;
        }
    }
    __shared__ unsigned int s_cnt4[4 * (kSelBlock / kWave)];
    const int64_t tile_lo = (int64_t)bid * tpb, tile_hi = (tile_lo + tpb < ntile) ? tile_lo + tpb : ntile;
    for (int64_t tile0 = tile_lo; tile0 < tile_hi; tile0 += 4) {
        bool al[4];
        unsigned long long bm[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t i = (tile0 + u) * kSelBlock + tid;
            double x = 0.0;
            const bool in = (tile0 + u < tile_hi) && i < N;
            if (in) x = X[i];
            al[u] = in && (flag ? (x <= eps) : (x < eps));
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            bm[u] = __ballot(al[u]);
            if (lane == 0) s_cnt4[u * (kSelBlock / kWave) + wid] = (unsigned)__popcll(bm[u]);
        }
        __syncthreads();
        const unsigned c = s_cnt4[lane];  // entry `lane` of the 64-vector
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
                A.cidx[base + woff + __popcll(bm[u] & below)] =
                    (int32_t)((tile0 + u) * kSelBlock + tid);
        }
        base += tot;
        __syncthreads();
    }
    KABC_STAMP(4)
    // Step 2 decision: α*ESS <= nparticles*min_r_ess  (src/smc.jl:145)
    const int resample =
        (A.mode == 0 && A.alpha * (double)ESS <= (double)N * A.min_r_ess) ? 1 : 0;
    const bool fail = resample && ESS == 0;
    if (resample && !fail) {
        // idx = repeat(idxalive, ceil(N/m))[1:N]  (src/smc.jl:146-147) is not materialised:
        // the next MCMC pass reads particle j from row cidx[j mod ESS] itself (the compacted
        // index is complete at the kernel boundary), so no barrier and no pass is needed
        // here.  Nothing reads `alive` after the slice-count barrier above.
        for (int64_t jdx = i_lo + tid; jdx < i_hi; jdx += kSelBlock) A.alive[jdx] = 1;
    } else if (!fail) {
        for (int64_t i = i_lo + tid; i < i_hi; i += kSelBlock) {
            const double x = X[i];
            A.alive_out[i] = (flag ? (x <= eps) : (x < eps)) ? 1 : 0;
        }
    }
    KABC_STAMP(5)
    if (bid != 0) return;
    // workgroup 0 leaves the scratch zeroed for the next call (every workgroup has passed
    // a barrier since its last read of it) and publishes the iteration's control block
    if (G > 1u) {
        for (int r = 0; r < rounds_used; ++r) g->hist[r][tid] = 0;
        if (tid == 0) g->ncand = 0;
    }
    if (tid == 0) {
        if (fail) {
            A.ctrl->error = 2;
            A.ctrl->done = 1;
            return;
        }
        if (A.stamps) A.stamps[7] += 1;
        A.ctrl->iteration += 1;
        A.ctrl->eps_prev = A.ctrl->eps;  // ϵv = ϵ
        A.ctrl->eps = eps;
        A.ctrl->min_alive = mn;
        A.ctrl->ess = ESS;
        A.ctrl->n_alive = resample ? N : ESS;
        A.ctrl->flag = flag;
        A.ctrl->resampled = resample;
        A.ctrl->accepted = 0;
        A.ctrl->passes = 0;
        A.ctrl->pass_open = 1;
        A.ctrl->use_ridx = 1;
    }
}
#endif  // KABC_SMC_SINGLE_UNIT

// ---- rows gathered by lane groups (round 6; A/B: KABC_SMC_COOP_GATHER) -----------------------------------
// A lane that fetches "its" random row with D/2 16-byte loads makes every load instruction of its wavefront
// touch 64 different cache lines for 16 bytes each.  Here the D/2 lanes of a group fetch ONE row with one
// instruction (a whole 8D-byte row per line request: the guide's gather recipe, MI355X_MICROARCH.md
// "Indexed rows"), 64 / (D/2) rows per instruction, and the rows reach their owners through a wave-private
// LDS tile (row stride 8D + 16 bytes).  idx: this lane's row; out: this lane's row.  All 64 lanes take part.
template <int D>
struct CoopGather {
    static constexpr int LPR = D / 2;           // lanes per row (16 bytes each)
    static constexpr int RPI = kWave / LPR;     // rows per load instruction
    static constexpr int NI = kWave / RPI;      // load instructions per 64 rows (= LPR)
    static constexpr int STRIDE = D + 2;        // doubles per tile row
    static constexpr bool ok = (D % 2 == 0) && D >= 4 && (kWave % LPR == 0);
    double2 v[NI];                              // chunk (lane % LPR) of row (j * RPI + lane / LPR)
    __device__ __forceinline__ void issue(const double* __restrict__ src, int64_t idx, int lane) {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int owner = j * RPI + lane / LPR;
            const unsigned lo = (unsigned)__shfl((int)(unsigned)idx, owner, kWave);
            const unsigned hi = (unsigned)__shfl((int)(unsigned)((uint64_t)idx >> 32), owner, kWave);
            const int64_t r = (int64_t)(((uint64_t)hi << 32) | lo);
            v[j] = *reinterpret_cast<const double2*>(src + r * D + 2 * (lane % LPR));
        }
    }
    // through the wave's tile; the tile may be reused as soon as this returns (one wavefront: LDS in order)
    __device__ __forceinline__ void deliver(double* tile, int lane, double* out) {
#pragma unroll
        for (int j = 0; j < NI; ++j)
            *reinterpret_cast<double2*>(tile + (j * RPI + lane / LPR) * STRIDE + 2 * (lane % LPR)) = v[j];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int k = 0; k < D / 2; ++k) {
            const double2 w = *reinterpret_cast<const double2*>(tile + lane * STRIDE + 2 * k);
            out[2 * k] = w.x;
            out[2 * k + 1] = w.y;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
};

template <int D, int COST, bool SIMPLE>
__global__ void __launch_bounds__(kSmcBlock) smc_mcmc_kernel(const SmcMcmcArgs A) {
    const int64_t wg = A.wg0 + (int64_t)blockIdx.x;
    const int64_t i = wg * kSmcBlock + threadIdx.x;
    unsigned long long n_eval = 0, n_acc = 0, n_prop = 0;
    if (A.ctrl->done || !A.ctrl->pass_open) return;  // uniform no-op
    // the log / sin-cos table of the arithmetic contract in LDS: a particle's pass takes one log, one
    // Box-Muller pair and, for a simulator cost, several more -- each a dependent gather from the
    // table; with two waves per SIMD (the rows of a 16-parameter particle fill the registers) the
    // L2 round trip of the table in global memory is not hidden.  Same values.
    __shared__ __attribute__((aligned(16))) double s_logtab[KABC_MATH_TAB_WORDS];
    static_assert(KABC_MATH_TAB_WORDS % kSmcBlock == 0, "the workgroup stages the table in equal parts");
#pragma unroll
    for (int j = 0; j < KABC_MATH_TAB_WORDS / kSmcBlock; ++j)
        s_logtab[threadIdx.x + j * kSmcBlock] = kabc_log_tab[threadIdx.x + j * kSmcBlock];
    __syncthreads();
    const int cur = A.ctrl->cur;
    const bool gather = A.ctrl->use_ridx != 0;
    const uint64_t pass = A.ctrl->pass + 1u;
    const double* __restrict__ theta_src = A.theta[cur];
    const double* __restrict__ X_src = A.X[cur];
    const double* __restrict__ lpi_src = A.lpi[cur];
    double Xfin = 0.0;
    bool alive_i = false;
#ifdef KABC_SMC_COOP_GATHER
    constexpr bool kCoop = CoopGather<D>::ok;
#else
    constexpr bool kCoop = false;
#endif
    // (the cooperative gather: every lane of a wavefront fetches rows for its neighbours, so a wavefront with
    // any particle in range runs the whole body; lanes out of range work on row 0 and store nothing)
    __shared__ __attribute__((aligned(16))) double s_tile[kCoop ? kSmcBlock / kWave : 1][kCoop ? kWave * (D + 2) : 1];
    const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x >> 6;
    const bool in = i < A.N;
    if (kCoop ? (i - lane < A.N) : in) {
        // idx = repeat(idxalive, ceil(N/m))[1:N]  (src/smc.jl:146-147), evaluated on the fly
        const bool remap = gather && A.ctrl->resampled != 0;
        const unsigned ess = (unsigned)A.ctrl->ess;
        const int64_t ii = in ? i : 0;
        const int64_t si = remap ? (int64_t)A.cidx[(unsigned)ii % ess] : ii;
        double th[D];
        CoopGather<kCoop ? D : 4> gth, gta, gtb;
        if constexpr (kCoop) gth.issue(theta_src, si, lane);
        else load_row<D>(theta_src + si * D, th);
        double Xi = X_src[si];
        double lpi = lpi_src[si];
        alive_i = in && A.alive[ii] != 0;
        if (kCoop || alive_i) {
            const uint64_t N = (uint64_t)A.N;
            const uint32_t w = (uint32_t)ii;
            const kabc_u128_t B0 = kabc_stream_block(A.seed, w, pass, 0u, KABC_DOM_SMC_MOVE);
            const kabc_u128_t B1 = kabc_stream_block(A.seed, w, pass, 1u, KABC_DOM_SMC_MOVE);
            const kabc_u128_t B2 = kabc_stream_block(A.seed, w, pass, 2u, KABC_DOM_SMC_MOVE);
            // while a==i ... ; while b==i || b==a ...  (src/smc.jl:163-164)
            int64_t a = (int64_t)kabc_index32(kabc_lo64(B0), (uint32_t)N - 1u);
            a += (a >= ii);
            const int64_t lo = a < ii ? a : ii, hi = a < ii ? ii : a;
            int64_t b = (int64_t)kabc_index32(kabc_hi64(B0), (uint32_t)N - 2u);
            b += (b >= lo);
            b += (b >= hi);
            double z0, z1;
            kabc_normal_pair_tab(kabc_lo64(B1), kabc_hi64(B1), &z0, &z1, s_logtab);
            const double s = A.max_stretch * z0 / kabc_sqrt((double)D);
            const int64_t sa = remap ? (int64_t)A.cidx[(unsigned)a % ess] : a;
            const int64_t sb = remap ? (int64_t)A.cidx[(unsigned)b % ess] : b;
            double ta[D], tb[D], prop[D], xp[D];
            if constexpr (kCoop) {
                gta.issue(theta_src, sa, lane);
                gtb.issue(theta_src, sb, lane);
                gth.deliver(s_tile[wv], lane, th);
                gta.deliver(s_tile[wv], lane, ta);
                gtb.deliver(s_tile[wv], lane, tb);
            } else {
                load_row<D>(theta_src + sa * D, ta);
                load_row<D>(theta_src + sb * D, tb);
            }
          if (alive_i) {
#pragma unroll
            for (int k = 0; k < D; ++k) {
                const double W = (tb[k] - ta[k]) * s;
                prop[k] = th[k] + W;
            }
            const double lprob = kabc_log_t(kabc_u01(kabc_lo64(B2)), s_logtab);
            n_prop = 1;
            const double lpp = factored_logpdf_push<D, SIMPLE>(A.prior.c, prop, xp, s_logtab);
            if (!(lpp < 0.0 && !kabc_isfinite(lpp))) {  // :173
                double lM = lpp - lpi + 0.0;
                if (!(lM < 0.0)) lM = (lM != lM) ? lM : 0.0;
                if (lprob < lM) {
                    kabc_cost_rng_t rng = {A.seed, pass, w, KABC_DOM_SMC_COST, 0u, 0u, nullptr, s_logtab};
                    if (A.aux) {
                        const int64_t sl = A.aux_ring > 1 ? (int64_t)(pass % (uint64_t)A.aux_ring) : 0;
                        rng.aux = A.aux + sl * (int64_t)kabc_cost_aux_words(COST) * A.N + i;
                        rng.aux_stride = (uint32_t)A.N;
                    }
                    const double Xp =
                        eval_cost<COST, D>(xp, A.cost_params, A.cost_data, A.cost_ndata, &rng);
                    n_eval = 1;
                    const double eps = A.ctrl->eps;
                    const bool reject = A.ctrl->flag ? (Xp > eps) : (Xp >= eps);
                    if (!reject) {
#pragma unroll
                        for (int k = 0; k < D; ++k) th[k] = prop[k];
                        Xi = Xp;
                        lpi = lpp;
                        n_acc = 1;
                    }
                }
            }
          }  // (alive_i)
        }
        if (in) {
            store_row<D>(A.theta[1 - cur] + i * D, th);
            A.X[1 - cur][i] = Xi;
            A.lpi[1 - cur][i] = lpi;
            Xfin = Xi;
        }
    }
    smc_block_stats(A.part, alive_i, Xfin, wg);
    // one counter line per workgroup (mod kSmcSlots): same-line atomics from 512
    // workgroups cost ~20 us per launch
    const unsigned long long se = wave_sum(n_eval), sa = wave_sum(n_acc), sp = wave_sum(n_prop);
    if ((threadIdx.x & (kWave - 1)) == 0) {
        unsigned long long* sl = A.slots + (size_t)(blockIdx.x & (kSmcSlots - 1)) * 8;
        if (sa) atomicAdd(&sl[0], sa);
        if (se) atomicAdd(&sl[1], se);
        if (sp) atomicAdd(&sl[2], sp);
    }
}

#ifdef KABC_SMC_SINGLE_UNIT
__global__ void __launch_bounds__(256) smc_finalize_kernel(const SmcFinalArgs A) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= A.N) return;
    const int cur = A.ctrl->cur;
    for (int k = 0; k < A.D; ++k) {
        const double v = A.theta[cur][i * A.D + k];
        const bool disc = A.dprior ? (A.dprior[k].discrete != 0) : (A.prior.c[k].discrete != 0);
        A.out[i * A.D + k] = disc ? kabc_rint(v) : v;
    }
    A.Xout[i] = A.X[cur][i];
}

// end of an ε-iteration: log it and apply the stop tests of src/smc.jl:194-198 (one thread)
__device__ __forceinline__ void smc_iter_end(SmcCtrl* ctrl, kabc_smc_iter_t* log, int64_t log_cap,
                                             int64_t N, const SmcLoopParams& P) {
    ctrl->pass_open = 0;
    const long long it = ctrl->iteration;
    const double eps = ctrl->eps, epsv = ctrl->eps_prev;
    if (log && it <= log_cap) {
        kabc_smc_iter_t L;
        L.eps = eps;
        L.ess = ctrl->ess;
        L.accepted = (int64_t)ctrl->accepted;
        L.resampled = ctrl->resampled;
        L.flag = ctrl->flag;
        L.mcmc_passes = ctrl->passes;
        L.reserved = 0;
        log[it - 1] = L;
    }
    const double acc = (double)ctrl->accepted;
    if (2.0 * kabc_fabs(epsv - eps) < P.r_epstol * (kabc_fabs(epsv) + kabc_fabs(eps)) ||
        eps <= P.epstol || acc < P.mcmc_tol * (double)N || it >= P.max_iterations)
        ctrl->done = 1;
}

// after every MCMC pass: fold the per-workgroup counter lines, flip the buffers,
// apply `accepted[] >= mcmc_tol * nparticles && break` (src/smc.jl:192); after the last
// pass of an iteration (end_iter) also the iteration's end, in the same launch
__global__ void __launch_bounds__(kSmcSlots) smc_pass_end_kernel(SmcCtrl* ctrl,
                                                                 unsigned long long* slots,
                                                                 int64_t N, double mcmc_tol,
                                                                 int end_iter,
                                                                 kabc_smc_iter_t* log,
                                                                 int64_t log_cap,
                                                                 SmcLoopParams P, int nregions = 1) {
    __shared__ unsigned long long sh[3][kSmcSlots / kWave];
    if (ctrl->done) return;
    const int tid = threadIdx.x;
    if (ctrl->pass_open) {  // uniform: thread 0 changes it only after the barrier below
        // (nregions > 1: a sharded cost loop -- one block of counter lines per rank, gathered)
        unsigned long long v[3];
        for (int j = 0; j < 3; ++j) {
            unsigned long long t = 0;
            for (int r = 0; r < nregions; ++r) {
                unsigned long long* q = slots + ((size_t)r * kSmcSlots + tid) * 8 + j;
                t += *q;
                *q = 0;
            }
            v[j] = wave_sum(t);
        }
        if ((tid & (kWave - 1)) == 0)
            for (int j = 0; j < 3; ++j) sh[j][tid >> 6] = v[j];
        __syncthreads();
        if (tid == 0) {
            unsigned long long t[3] = {0, 0, 0};
            for (int w = 0; w < kSmcSlots / kWave; ++w)
                for (int j = 0; j < 3; ++j) t[j] += sh[j][w];
            ctrl->accepted += t[0];
            ctrl->cost_evals += t[1];
            ctrl->proposals += t[2];
            ctrl->pass += 1;
            ctrl->passes += 1;
            ctrl->cur ^= 1;
            ctrl->use_ridx = 0;
            if ((double)ctrl->accepted >= mcmc_tol * (double)N) ctrl->pass_open = 0;
        }
    }
    if (end_iter && tid == 0) smc_iter_end(ctrl, log, log_cap, N, P);
}

// the iteration's end on its own (when the host stopped enqueueing retry passes early)
__global__ void smc_iter_end_kernel(SmcCtrl* ctrl, kabc_smc_iter_t* log, int64_t log_cap,
                                    int64_t N, SmcLoopParams P) {
    if (ctrl->done) return;
    smc_iter_end(ctrl, log, log_cap, N, P);
}

#endif  // KABC_SMC_SINGLE_UNIT

#ifndef __HIPCC_RTC__  // host side
using SmcLaunchFn = void (*)(const SmcMcmcArgs&, hipStream_t);
using SmcLaunch = Launcher<SmcMcmcArgs>;       // host function or run-time compiled kernel
using SmcInitLaunch = Launcher<SmcInitArgs>;
// launch grid: the workgroups [wg0, wg0 + nwg) of a sharded run, else all ceil(N / 64)
template <class Args>
inline unsigned smc_grid(const Args& a) {
    return (unsigned)(a.sharded ? a.nwg : (a.N + kSmcBlock - 1) / kSmcBlock);
}
inline dim3 smc_mcmc_geom(const SmcMcmcArgs& a) { return dim3(smc_grid(a)); }
inline dim3 smc_init_geom(const SmcInitArgs& a) { return dim3(smc_grid(a)); }
struct ModelUnit;
SmcLaunch find_smc_kernel(int cost_id, int D, bool simple_prior, ModelUnit* unit = nullptr);
#endif

}  // namespace kabc
