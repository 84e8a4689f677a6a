// smc_small_kernel.hpp -- the ε-loop of smc(prior, cost; ...) (src/smc.jl:131-199) for SMALL
// ensembles -- nparticles <= 256, which includes the reference's default nparticles = 100
// (src/smc.jl:96) and README.md:80-84 -- in ONE workgroup, one thread per particle.
//
// Why a third driver: at this size the kernel-per-phase path is nothing but launches (select +
// propose/accept + pass-end per iteration: ~30 us of which the select kernel alone, built to rank
// 2^15..2^21 keys with 1024 threads, takes 16.5) and the persistent loop kernel pays its device-wide
// barrier protocol for a grid of one.  With every particle in one workgroup the reference's data
// dependences are workgroup barriers:
//   * the ensemble (theta, X, logprior) lives in LDS for the whole launch; partner rows of the
//     proposal theta_i + (theta_b - theta_a) s are LDS reads, the frozen-ensemble rule of
//     src/smc.jl:160-167 is one __syncthreads between "everybody has read" and "I write my row";
//   * ε = quantile(Xs[alive], α) (type 7): every thread counts the alive keys below its own
//     (N broadcast LDS reads) -- its rank -- and the two threads holding the bracketing order
//     statistics publish them; alive mask, ESS and the compacted index idxalive come from wave
//     ballots; the cyclic resample repeat(idxalive, ceil(N/ESS))[1:N] (:146-147) is a gather
//     through that index between two barriers;
//   * retry passes, the stop tests (:192-198) and the iteration log are evaluated by every
//     thread from the same LDS words.
// One launch runs up to `max_passes` passes (a prepared cost's ring of pre-pass words covers that
// many: ais_aux_kernels.hpp) or until the loop ends, and leaves the state where the other drivers
// keep it (theta / X / lpi of buffer set ctrl->cur, alive, SmcCtrl), so the finalize kernel and the
// host code are shared.  Same draws (counter streams keyed by particle and pass), same operation
// order: bit-identical to the other two drivers and to the oracle.
#pragma once

#include "smc_kernels.hpp"

namespace kabc {

constexpr int kSmallBlock = 256;
constexpr int kSmallWaves = kSmallBlock / kWave;

struct SmcSmallArgs {
    double* theta[2];
    double* X[2];
    double* lpi[2];
    uint8_t* alive;
    SmcCtrl* ctrl;
    kabc_smc_iter_t* log;
    int64_t log_cap;
    const double* cost_params;
    const double* cost_data;
    int64_t cost_ndata;
    int64_t N;
    uint64_t seed;
    double max_stretch, alpha, min_r_ess;
    SmcLoopParams loop;
    int32_t retry_n;     // 1 + mcmc_retrys
    int32_t max_passes;  // passes this launch may run (> 0 only with retry_n == 1), else until the loop ends
    const double* aux;   // prepared cost words [aux_ring][W][N] (pass t in slot t mod aux_ring), or NULL
    int32_t aux_ring;
    const PriorDev* prior;  // [D] prepared components, device memory
};

// sums of up to three 0/1 flags over the workgroup (every thread gets them), packed 10 bits each: wave
// ballots, one LDS word per wave, ONE workgroup barrier.  The caller alternates between two lines
// (`s_cnt`), so no trailing barrier is needed: a line is rewritten only after another barrier.
__device__ __forceinline__ unsigned small_count3(bool f0, bool f1, bool f2, unsigned* s_cnt, int wid, int lane) {
    const unsigned c = (unsigned)__popcll(__ballot(f0)) | ((unsigned)__popcll(__ballot(f1)) << 10) |
                       ((unsigned)__popcll(__ballot(f2)) << 20);
    if (lane == 0) s_cnt[wid] = c;
    __syncthreads();
    unsigned t = 0;
#pragma unroll
    for (int w = 0; w < kSmallWaves; ++w) t += s_cnt[w];
    return t;
}

template <int D, int COST, bool SIMPLE>
__global__ void __launch_bounds__(kSmallBlock) smc_small_kernel(const SmcSmallArgs A) {
    __shared__ __attribute__((aligned(16))) double s_th[kSmallBlock][D];
    __shared__ double s_X[kSmallBlock], s_lpi[kSmallBlock];
    __shared__ unsigned long long s_key[kSmallBlock];
    __shared__ int s_cidx[kSmallBlock];
    __shared__ unsigned s_cnt[kSmallWaves], s_cntA[kSmallWaves], s_cntB[kSmallWaves];
    __shared__ double s_ab[3];  // the two bracketing order statistics, the minimum
    __shared__ PriorDev s_prior[D];
    __shared__ __attribute__((aligned(16))) double s_logtab[KABC_MATH_TAB_WORDS];

    const int tid = threadIdx.x, lane = tid & (kWave - 1), wid = tid >> 6;
    const int N = (int)A.N;
    const bool in = tid < N;
    SmcCtrl c = *A.ctrl;  // (every thread: uniform values)
    if (c.done) return;
    const int cur = c.cur;
    // ---- stage: tables, prior, the ensemble
    static_assert(KABC_MATH_TAB_WORDS == 2 * kSmallBlock, "two table words per thread");
    s_logtab[tid] = kabc_log_tab[tid];
    s_logtab[tid + kSmallBlock] = kabc_log_tab[tid + kSmallBlock];
    if (tid < D * (int)(sizeof(PriorDev) / 8))
        reinterpret_cast<double*>(s_prior)[tid] = reinterpret_cast<const double*>(A.prior)[tid];
    bool alive_i = false;
    if (in) {
        double row[D];
        load_row<D>(A.theta[cur] + (size_t)tid * D, row);
#pragma unroll
        for (int k = 0; k < D; ++k) s_th[tid][k] = row[k];
        s_X[tid] = A.X[cur][tid];
        s_lpi[tid] = A.lpi[cur][tid];
        alive_i = A.alive[tid] != 0;
    }
    __syncthreads();

    const int R = A.retry_n;
    int passes_left = (A.max_passes > 0 && R == 1) ? A.max_passes : 0x7fffffff;
    const double sqrtD = kabc_sqrt((double)D);
    const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));

    while (passes_left >= R) {
        // ================= Step 1 (:134-143): ε = quantile(Xs[alive], α), alive mask, ESS
        const double Xi = in ? s_X[tid] : 0.0;
        const unsigned nc = small_count3(alive_i, alive_i && Xi != Xi, false, s_cntA, wid, lane);
        const unsigned n = nc & 1023u, nn = (nc >> 10) & 1023u;
        if (n == 0u || nn > 0u) {
            c.error = nn > 0u ? 1 : 2;
            c.done = 1;
            break;
        }
        // keys: dead particles sort behind every alive one (no alive key is all ones: NaNs ended
        // the run above); -0.0 folds onto +0.0 so that key order and `<` on the values agree
        const unsigned long long ki = alive_i ? key_of(Xi + 0.0) : ~0ull;
        s_key[tid] = ki;
        __syncthreads();
        unsigned rank = 0;  // alive keys before mine in (key, index) order
#pragma unroll 4
        for (int j = 0; j < N; ++j) {
            const unsigned long long kj = s_key[j];  // (broadcast read)
            rank += (kj < ki || (kj == ki && j < tid)) ? 1u : 0u;
        }
        // ranks of the two bracketing order statistics (Statistics.quantile, type 7)
        const double aleph = (double)n * A.alpha + (1.0 - A.alpha);
        long long jq = (long long)aleph;
        if (jq < 1) jq = 1;
        if (jq > (long long)n - 1) jq = (long long)n - 1;
        if (n == 1u) jq = 1;
        double gq = aleph - (double)jq;
        gq = gq < 0.0 ? 0.0 : (gq > 1.0 ? 1.0 : gq);
        if (alive_i) {
            if ((long long)rank == jq - 1) s_ab[0] = Xi;
            if ((long long)rank == (n == 1u ? 0 : jq)) s_ab[1] = Xi;
            if (rank == 0u) s_ab[2] = Xi;  // minimum(Xs[alive])
        }
        __syncthreads();
        const double qa = s_ab[0], qb = s_ab[1], mn = s_ab[2];
        double eps;
        if (kabc_isfinite(qa) && kabc_isfinite(qb)) eps = qa + gq * (qb - qa);
        else eps = (1.0 - gq) * qa + gq * qb;
        const int flag = (eps > mn) ? 0 : 1;  // :136-141
        alive_i = in && (flag ? (Xi <= eps) : (Xi < eps));  // over ALL particles (:137,:139)
        // ESS and the compacted index idxalive = (1:N)[alive]
        const unsigned long long bm = __ballot(alive_i);
        if (lane == 0) s_cnt[wid] = (unsigned)__popcll(bm);
        __syncthreads();
        unsigned woff = 0, ESS = 0;
#pragma unroll
        for (int w = 0; w < kSmallWaves; ++w) {
            woff += (w < wid) ? s_cnt[w] : 0u;
            ESS += s_cnt[w];
        }
        if (alive_i) s_cidx[woff + (unsigned)__popcll(bm & below)] = tid;
        __syncthreads();
        // ================= Step 2 (:145-153): cyclic resample
        const int resampled = (A.alpha * (double)ESS <= (double)N * A.min_r_ess) ? 1 : 0;
        if (resampled && ESS == 0u) {
            c.error = 2;
            c.done = 1;
            break;
        }
        if (resampled) {
            double row[D], xs = 0.0, ls = 0.0;
            if (in) {
                const int src = s_cidx[(unsigned)tid % ESS];  // repeat(idxalive, ceil(N/m))[1:N]
#pragma unroll
                for (int k = 0; k < D; ++k) row[k] = s_th[src][k];
                xs = s_X[src];
                ls = s_lpi[src];
            }
            __syncthreads();
            if (in) {
#pragma unroll
                for (int k = 0; k < D; ++k) s_th[tid][k] = row[k];
                s_X[tid] = xs;
                s_lpi[tid] = ls;
            }
            alive_i = in;
            __syncthreads();
        }
        c.iteration += 1;
        c.eps_prev = c.eps;  // ϵv = ϵ
        c.eps = eps;
        c.min_alive = mn;
        c.ess = (long long)ESS;
        c.n_alive = resampled ? (long long)N : (long long)ESS;
        c.flag = flag;
        c.resampled = resampled;
        c.accepted = 0;
        c.passes = 0;
        // ================= Step 3 (:156-193): propose from the frozen ensemble, then accept
        for (int r = 0; r < R; ++r) {
            const uint64_t pass = c.pass + 1u;
            bool acc = false, evald = false;
            double nth[D], nX = 0.0, nlp = 0.0;
            if (alive_i) {
                const uint32_t w = (uint32_t)tid;
                const int64_t i = tid;
                const kabc_u128_t B0 = kabc_stream_block(A.seed, w, pass, 0u, KABC_DOM_SMC_MOVE);
                const kabc_u128_t B1 = kabc_stream_block(A.seed, w, pass, 1u, KABC_DOM_SMC_MOVE);
                const kabc_u128_t B2 = kabc_stream_block(A.seed, w, pass, 2u, KABC_DOM_SMC_MOVE);
                // while a==i ... ; while b==i || b==a ...  (src/smc.jl:163-164)
                int64_t a = (int64_t)kabc_index32(kabc_lo64(B0), (uint32_t)N - 1u);
                a += (a >= i);
                const int64_t lo = a < i ? a : i, hi = a < i ? i : a;
                int64_t b = (int64_t)kabc_index32(kabc_hi64(B0), (uint32_t)N - 2u);
                b += (b >= lo);
                b += (b >= hi);
                double z0, z1;
                kabc_normal_pair_tab(kabc_lo64(B1), kabc_hi64(B1), &z0, &z1, s_logtab);
                const double s = A.max_stretch * z0 / sqrtD;
                double prop[D], xp[D];
#pragma unroll
                for (int k = 0; k < D; ++k) {
                    const double W = (s_th[b][k] - s_th[a][k]) * s;
                    prop[k] = s_th[tid][k] + W;
                }
                const double lprob = kabc_log_t(kabc_u01(kabc_lo64(B2)), s_logtab);
                const double lpp = factored_logpdf_push<D, SIMPLE, false>(s_prior, prop, xp, s_logtab);
                if (!(lpp < 0.0 && !kabc_isfinite(lpp))) {  // :173
                    double lM = lpp - s_lpi[tid] + 0.0;
                    if (!(lM < 0.0)) lM = (lM != lM) ? lM : 0.0;
                    if (lprob < lM) {
                        kabc_cost_rng_t rng = {A.seed, pass, w, KABC_DOM_SMC_COST, 0u, 0u, nullptr, s_logtab};
                        if (A.aux) {
                            const int64_t sl = A.aux_ring > 1 ? (int64_t)(pass % (uint64_t)A.aux_ring) : 0;
                            rng.aux = A.aux + sl * (int64_t)kabc_cost_aux_words(COST) * A.N + i;
                            rng.aux_stride = (uint32_t)A.N;
                        }
                        const double Xp = eval_cost<COST, D>(xp, A.cost_params, A.cost_data, A.cost_ndata, &rng);
                        evald = true;
                        const bool reject = flag ? (Xp > eps) : (Xp >= eps);
                        if (!reject) {
#pragma unroll
                            for (int k = 0; k < D; ++k) nth[k] = prop[k];
                            nX = Xp;
                            nlp = lpp;
                            acc = true;
                        }
                    }
                }
            }
            // (the barrier inside: every read of the frozen rows is done)
            const unsigned pc = small_count3(alive_i, acc, evald, s_cntB, wid, lane);
            const unsigned n_prop = pc & 1023u, n_acc = (pc >> 10) & 1023u, n_eval = pc >> 20;
            if (acc) {
#pragma unroll
                for (int k = 0; k < D; ++k) s_th[tid][k] = nth[k];
                s_X[tid] = nX;
                s_lpi[tid] = nlp;
            }
            if (R > 1) __syncthreads();  // (a retry pass reads the rows just written; uniform)
            c.accepted += n_acc;
            c.cost_evals += n_eval;
            c.proposals += n_prop;
            c.pass += 1;
            c.passes += 1;
            --passes_left;
            if ((double)c.accepted >= A.loop.mcmc_tol * (double)N) break;  // :192
        }
        // ================= end of the iteration: log, stop tests (:194-198)
        if (tid == 0 && A.log && c.iteration <= A.log_cap) {
            kabc_smc_iter_t L;
            L.eps = c.eps;
            L.ess = c.ess;
            L.accepted = (int64_t)c.accepted;
            L.resampled = c.resampled;
            L.flag = c.flag;
            L.mcmc_passes = c.passes;
            L.reserved = 0;
            A.log[c.iteration - 1] = L;
        }
        const double acc_it = (double)c.accepted;
        if (2.0 * kabc_fabs(c.eps_prev - c.eps) < A.loop.r_epstol * (kabc_fabs(c.eps_prev) + kabc_fabs(c.eps)) ||
            c.eps <= A.loop.epstol || acc_it < A.loop.mcmc_tol * (double)N || c.iteration >= A.loop.max_iterations) {
            c.done = 1;
            break;
        }
    }
    // ---- leave the state where the other drivers keep it
    __syncthreads();
    if (in) {
        double row[D];
#pragma unroll
        for (int k = 0; k < D; ++k) row[k] = s_th[tid][k];
        store_row<D>(A.theta[cur] + (size_t)tid * D, row);
        A.X[cur][tid] = s_X[tid];
        A.lpi[cur][tid] = s_lpi[tid];
        A.alive[tid] = alive_i ? 1 : 0;
    }
    if (tid == 0) {
        c.pass_open = 0;
        c.use_ridx = 0;
        *A.ctrl = c;
    }
}

#ifndef __HIPCC_RTC__  // host side
using SmcSmallLaunchFn = void (*)(const SmcSmallArgs&, hipStream_t);
using SmcSmallLaunch = Launcher<SmcSmallArgs>;
inline dim3 smc_small_geom(const SmcSmallArgs&) { return dim3(1); }
template <int D, int COST, bool SIMPLE>
inline void launch_smc_small(const SmcSmallArgs& a, hipStream_t s) {
    hipLaunchKernelGGL((smc_small_kernel<D, COST, SIMPLE>), dim3(1), dim3(kSmallBlock), 0, s, a);
}
SmcSmallLaunch find_smc_small_kernel(int cost_id, int D, bool simple_prior, ModelUnit* unit = nullptr);
#endif

}  // namespace kabc
