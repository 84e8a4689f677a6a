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

namespace kabc {

struct SmcCtrl {
    double eps;
    double min_alive;               // minimum(Xs[alive]) of the last select
    long long ess;                  // sum(alive) before resampling
    long long n_alive;              // sum(alive) after step 2
    int32_t flag;
    int32_t resampled;
    int32_t error;                  // 1 NaN cost among alive, 2 no alive particle
    int32_t pad;
    unsigned long long accepted;    // reset by select, accumulated by the MCMC passes
    unsigned long long cost_evals;  // cumulative
    unsigned long long proposals;   // cumulative
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
};

struct SmcSelectArgs {
    const double* X;
    uint8_t* alive;
    int32_t* ridx;   // out: source row of particle i for the next MCMC pass
    int32_t* cidx;   // scratch: compacted alive indices
    SmcCtrl* ctrl;
    int64_t N;
    double alpha;
    double min_r_ess;
};

struct SmcMcmcArgs {
    const double* theta_src;
    const double* X_src;
    const double* lpi_src;
    double* theta_dst;
    double* X_dst;
    double* lpi_dst;
    const uint8_t* alive;
    const int32_t* ridx;  // NULL = identity
    SmcCtrl* ctrl;
    const double* cost_params;
    const double* cost_data;
    int64_t cost_ndata;
    int64_t N;
    uint64_t seed;
    uint64_t pass;
    double max_stretch;
    PriorSet prior;
};

struct SmcFinalArgs {
    const double* theta;
    double* out;
    int64_t N;
    int32_t D;
    PriorSet prior;
};

constexpr int kSmcBlock = 64;
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

template <int D>
__global__ void __launch_bounds__(kSmcBlock) smc_init_kernel(const SmcInitArgs A) {
    const int64_t i = (int64_t)blockIdx.x * kSmcBlock + threadIdx.x;
    if (i >= A.N) return;
    double x[D], xp[D];
    for (int k = 0; k < D; ++k) {
        kabc_slotwin_t win = {A.seed, 0ull, (uint32_t)i, KABC_DOM_SMC_INIT,
                              (uint32_t)k * KABC_SLOTS_PER_DIM};
        x[k] = kabc_sample_prior(&A.raw[k], &win);
    }
    const double lp = factored_logpdf_push<D>(A.prior, x, xp);
    kabc_cost_rng_t rng = {A.seed, 0ull, (uint32_t)i, KABC_DOM_SMC_INIT_COST, 0u};
    const double c =
        kabc_cost_eval(A.cost_id, xp, D, A.cost_params, A.cost_data, A.cost_ndata, &rng);
    store_row<D>(A.theta + i * D, x);
    A.X[i] = c;
    A.lpi[i] = lp;
    A.alive[i] = 1;
    if (i == 0) {
        A.ctrl->cost_evals = (unsigned long long)A.N;
        A.ctrl->proposals = 0;
        A.ctrl->accepted = 0;
        A.ctrl->error = 0;
    }
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

__global__ void __launch_bounds__(kSelBlock) smc_select_kernel(const SmcSelectArgs A) {
    __shared__ unsigned int hist[256];
    __shared__ long long sh_ll[kSelBlock / kWave];
    __shared__ uint64_t sh_u[kSelBlock / kWave];
    __shared__ uint64_t s_prefix;
    __shared__ long long s_k;
    __shared__ double s_eps;
    __shared__ int s_flag, s_resample;
    __shared__ long long s_scan[kSelBlock / kWave];

    const int tid = threadIdx.x;
    const int64_t N = A.N;

    // (a) n = count(alive), mn = minimum(Xs[alive]), NaN check
    long long cnt = 0, nanc = 0;
    uint64_t kmin = ~0ull;
    for (int64_t i = tid; i < N; i += kSelBlock) {
        if (A.alive[i]) {
            const double x = A.X[i];
            ++cnt;
            if (x != x) ++nanc;
            const uint64_t k = key_of(x);
            kmin = k < kmin ? k : kmin;
        }
    }
    const long long n = block_sum_ll(cnt, sh_ll);
    const long long nn = block_sum_ll(nanc, sh_ll);
    kmin = block_min_u64(kmin, sh_u);
    if (n == 0 || nn > 0) {
        if (tid == 0) A.ctrl->error = (nn > 0) ? 1 : 2;
        return;
    }
    const double mn = val_of(kmin);

    // (b) ranks of the two bracketing order statistics (Statistics.quantile, type 7)
    const double aleph = (double)n * A.alpha + (1.0 - A.alpha);
    long long j = (long long)aleph;
    if (j < 1) j = 1;
    if (j > n - 1) j = n - 1;
    if (n == 1) j = 1;
    double g = aleph - (double)j;
    g = g < 0.0 ? 0.0 : (g > 1.0 ? 1.0 : g);

    // radix select of rank j-1 (0-based) over the alive keys, 8 bits per pass
    if (tid == 0) {
        s_prefix = 0;
        s_k = j - 1;
    }
    uint64_t mask = 0;
    for (int shift = 56; shift >= 0; shift -= 8) {
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        const uint64_t prefix = s_prefix;
        for (int64_t i = tid; i < N; i += kSelBlock) {
            if (A.alive[i]) {
                const uint64_t k = key_of(A.X[i]);
                if ((k & mask) == prefix) atomicAdd(&hist[(unsigned)((k >> shift) & 0xff)], 1u);
            }
        }
        __syncthreads();
        if (tid == 0) {
            long long k = s_k;
            int bin = 0;
            for (; bin < 256; ++bin) {
                const long long c = hist[bin];
                if (k < c) break;
                k -= c;
            }
            s_k = k;
            s_prefix = prefix | ((uint64_t)bin << shift);
        }
        mask |= (0xffull << shift);
        __syncthreads();
    }
    const uint64_t key_a = s_prefix;
    // count(keys <= key_a) and min(keys > key_a)
    long long cle = 0;
    uint64_t kgt = ~0ull;
    for (int64_t i = tid; i < N; i += kSelBlock) {
        if (A.alive[i]) {
            const uint64_t k = key_of(A.X[i]);
            if (k <= key_a) ++cle;
            else kgt = k < kgt ? k : kgt;
        }
    }
    const long long n_le = block_sum_ll(cle, sh_ll);
    kgt = block_min_u64(kgt, sh_u);
    if (tid == 0) {
        const double a = val_of(key_a);
        const double b = (n == 1 || n_le >= j + 1) ? a : val_of(kgt);
        double eps;
        if (kabc_isfinite(a) && kabc_isfinite(b)) eps = a + g * (b - a);
        else eps = (1.0 - g) * a + g * b;
        s_eps = eps;
        s_flag = (eps > mn) ? 0 : 1;  // src/smc.jl:135-141
    }
    __syncthreads();
    const double eps = s_eps;
    const int flag = s_flag;

    // (d) new alive mask over ALL particles, ESS, compaction offsets.
    // contiguous chunk per thread so that the compacted order is ascending in i.
    const int64_t chunk = (N + kSelBlock - 1) / kSelBlock;
    const int64_t i0 = (int64_t)tid * chunk;
    const int64_t i1 = (i0 + chunk < N) ? i0 + chunk : N;
    long long mine = 0;
    for (int64_t i = i0; i < i1; ++i) {
        const double x = A.X[i];
        const bool al = flag ? (x <= eps) : (x < eps);
        mine += al ? 1 : 0;
    }
    // exclusive scan of `mine` over the block
    long long incl = mine;
    for (int off = 1; off < kWave; off <<= 1) {
        const long long o = __shfl_up(incl, off, kWave);
        if ((tid & 63) >= off) incl += o;
    }
    if ((tid & 63) == 63) s_scan[tid >> 6] = incl;
    __syncthreads();
    long long wave_off = 0, total = 0;
    for (int w = 0; w < kSelBlock / kWave; ++w) {
        if (w < (tid >> 6)) wave_off += s_scan[w];
        total += s_scan[w];
    }
    const long long excl = wave_off + incl - mine;
    const long long ESS = total;
    if (tid == 0) {
        // Step 2 decision: α*ESS <= nparticles*min_r_ess  (src/smc.jl:145)
        s_resample = (A.alpha * (double)ESS <= (double)N * A.min_r_ess) ? 1 : 0;
    }
    __syncthreads();
    const int resample = s_resample;
    if (resample && ESS == 0) {
        if (tid == 0) A.ctrl->error = 2;
        return;
    }
    if (resample) {
        long long o = excl;
        for (int64_t i = i0; i < i1; ++i) {
            const double x = A.X[i];
            const bool al = flag ? (x <= eps) : (x < eps);
            if (al) A.cidx[o++] = (int32_t)i;
        }
        __threadfence_block();
        __syncthreads();
        // idx = repeat(idxalive, ceil(N/m))[1:N]  (src/smc.jl:146-147)
        for (int64_t jdx = tid; jdx < N; jdx += kSelBlock) {
            A.ridx[jdx] = A.cidx[jdx % ESS];
            A.alive[jdx] = 1;
        }
    } else {
        for (int64_t i = i0; i < i1; ++i) {
            const double x = A.X[i];
            const bool al = flag ? (x <= eps) : (x < eps);
            A.alive[i] = al ? 1 : 0;
            A.ridx[i] = (int32_t)i;
        }
    }
    if (tid == 0) {
        A.ctrl->eps = eps;
        A.ctrl->min_alive = mn;
        A.ctrl->ess = ESS;
        A.ctrl->n_alive = resample ? N : ESS;
        A.ctrl->flag = flag;
        A.ctrl->resampled = resample;
        A.ctrl->accepted = 0;
    }
}

#endif  // KABC_SMC_SINGLE_UNIT

template <int D, int COST>
__global__ void __launch_bounds__(kSmcBlock) smc_mcmc_kernel(const SmcMcmcArgs A) {
    const int64_t i = (int64_t)blockIdx.x * kSmcBlock + threadIdx.x;
    unsigned long long n_eval = 0, n_acc = 0, n_prop = 0;
    if (i < A.N) {
        const int64_t si = A.ridx ? A.ridx[i] : i;
        double th[D];
        load_row<D>(A.theta_src + si * D, th);
        double Xi = A.X_src[si];
        double lpi = A.lpi_src[si];
        if (A.alive[i]) {
            const uint64_t N = (uint64_t)A.N;
            const uint32_t w = (uint32_t)i;
            const kabc_u128_t B0 = kabc_stream_block(A.seed, w, A.pass, 0u, KABC_DOM_SMC_MOVE);
            const kabc_u128_t B1 = kabc_stream_block(A.seed, w, A.pass, 1u, KABC_DOM_SMC_MOVE);
            const kabc_u128_t B2 = kabc_stream_block(A.seed, w, A.pass, 2u, KABC_DOM_SMC_MOVE);
            // while a==i ... ; while b==i || b==a ...  (src/smc.jl:163-164)
            int64_t a = (int64_t)kabc_index(kabc_lo64(B0), N - 1u);
            a += (a >= i);
            const int64_t lo = a < i ? a : i, hi = a < i ? i : a;
            int64_t b = (int64_t)kabc_index(kabc_hi64(B0), N - 2u);
            b += (b >= lo);
            b += (b >= hi);
            double z0, z1;
            kabc_normal_pair(kabc_lo64(B1), kabc_hi64(B1), &z0, &z1);
            const double s = A.max_stretch * z0 / kabc_sqrt((double)D);
            const int64_t sa = A.ridx ? A.ridx[a] : a;
            const int64_t sb = A.ridx ? A.ridx[b] : b;
            double ta[D], tb[D], prop[D], xp[D];
            load_row<D>(A.theta_src + sa * D, ta);
            load_row<D>(A.theta_src + sb * D, tb);
#pragma unroll
            for (int k = 0; k < D; ++k) {
                const double W = (tb[k] - ta[k]) * s;
                prop[k] = th[k] + W;
            }
            const double lprob = kabc_log(kabc_u01(kabc_lo64(B2)));
            n_prop = 1;
            const double lpp = factored_logpdf_push<D>(A.prior, prop, xp);
            if (!(lpp < 0.0 && !kabc_isfinite(lpp))) {  // :173
                double lM = lpp - lpi + 0.0;
                if (!(lM < 0.0)) lM = (lM != lM) ? lM : 0.0;
                if (lprob < lM) {
                    kabc_cost_rng_t rng = {A.seed, A.pass, w, KABC_DOM_SMC_COST, 0u};
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
        }
        store_row<D>(A.theta_dst + i * D, th);
        A.X_dst[i] = Xi;
        A.lpi_dst[i] = lpi;
    }
    const unsigned long long se = wave_sum(n_eval), sa = wave_sum(n_acc), sp = wave_sum(n_prop);
    if ((threadIdx.x & (kWave - 1)) == 0) {
        if (sa) atomicAdd(&A.ctrl->accepted, sa);
        if (se) atomicAdd(&A.ctrl->cost_evals, se);
        if (sp) atomicAdd(&A.ctrl->proposals, sp);
    }
}

#ifdef KABC_SMC_SINGLE_UNIT
__global__ void __launch_bounds__(256) smc_finalize_kernel(const SmcFinalArgs A) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= A.N) return;
    for (int k = 0; k < A.D; ++k) {
        const double v = A.theta[i * A.D + k];
        A.out[i * A.D + k] = A.prior.c[k].discrete ? kabc_rint(v) : v;
    }
}

#endif  // KABC_SMC_SINGLE_UNIT

using SmcLaunchFn = void (*)(const SmcMcmcArgs&, hipStream_t);
SmcLaunchFn find_smc_kernel(int cost_id, int D);

}  // namespace kabc
