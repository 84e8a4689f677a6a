// smc_dyn_kernels.hpp -- smc() for length(prior) > KABC_MAX_DIM: src/smc.jl:119-125 (init) and
// :160-191 (propose + accept) with the dimension as a RUN-TIME value, rows in memory.  Same
// draws and operation order as smc_init_kernel / smc_mcmc_kernel (bit-identical to the oracle);
// it plugs into the kernel-per-phase path (select / pass_end / finalize do not depend on D).
#pragma once

#include "smc_kernels.hpp"

namespace kabc {

struct SmcDynArgs {
    double* theta[2];
    double* X[2];
    double* lpi[2];
    uint8_t* alive;
    const int32_t* cidx;
    SmcCtrl* ctrl;
    unsigned long long* slots;
    const double* cost_params;
    const double* cost_data;
    int64_t cost_ndata;
    int64_t N;
    uint64_t seed;
    double max_stretch;
    int32_t D, cost_id;
    const PriorDev* prior;      // [D] prepared components (device)
    const kabc_prior_t* raw;    // [D] raw components (device; init)
    double* scratch;            // [N][2][D]: proposal, push_p(proposal)
    unsigned long long* part;
};

template <int COST>
__device__ __forceinline__ double smc_dyn_cost(int cost_id, const double* x, int D, const double* params,
                                               const double* data, int64_t ndata, kabc_cost_rng_t* rng) {
#ifdef KABC_USER_COST_DEFINED
    if constexpr (COST == KABC_COST_USER) return kabc_user_cost(x, D, params, data, ndata, rng);
#endif
    return kabc_cost_eval(cost_id, x, D, params, data, ndata, rng);
}

__device__ __forceinline__ double smc_dyn_logpdf_push(const SmcDynArgs& A, const double* x, double* xp) {
    double s = 0.0;
    for (int k = 0; k < A.D; ++k) {
        const PriorDev q = A.prior[k];
        const double v = q.discrete ? kabc_rint(x[k]) : x[k];
        xp[k] = v;
        const double l = comp_logpdf_general_body(q.kind, q.p[0], q.p[1], q.p[2], q.p[3], q.c0, q.c1, q.rb, v);
        s = (k == 0) ? l : s + l;
    }
    return joint_logpdf_or(s, A.prior[0].kind, xp, A.D, A.prior, kabc_log_tab);
}

template <int COST>
__global__ void __launch_bounds__(kSmcBlock) smc_dyn_init_kernel(const SmcDynArgs A) {
    const int64_t i = (int64_t)blockIdx.x * kSmcBlock + threadIdx.x;
    double c = 0.0;
    if (i < A.N) {
        const int D = A.D;
        double* x = A.theta[0] + i * D;
        double* xp = A.scratch + (i * 2) * D;
        for (int k = 0; k < D; ++k) {
            kabc_slotwin_t win = {A.seed, 0ull, (uint32_t)i, KABC_DOM_SMC_INIT, (uint32_t)k * KABC_SLOTS_PER_DIM};
            x[k] = kabc_sample_prior(&A.raw[k], &win);  // (a pointer INTO the array: joint priors, kabc_sampling.h)
        }
        const double lp = smc_dyn_logpdf_push(A, x, xp);
        kabc_cost_rng_t rng = {A.seed, 0ull, (uint32_t)i, KABC_DOM_SMC_INIT_COST, 0u};
        c = smc_dyn_cost<COST>(A.cost_id, xp, D, A.cost_params, A.cost_data, A.cost_ndata, &rng);
        A.X[0][i] = c;
        A.lpi[0][i] = lp;
        A.alive[i] = 1;
        if (i == 0) {
            SmcCtrl cc = {};
            cc.eps = KABC_INF;  // ϵ = Inf  (src/smc.jl:127)
            cc.eps_prev = KABC_INF;
            cc.cost_evals = (unsigned long long)A.N;
            *A.ctrl = cc;
        }
    }
    smc_block_stats(A.part, i < A.N, c);
}

template <int COST>
__global__ void __launch_bounds__(kSmcBlock) smc_dyn_mcmc_kernel(const SmcDynArgs A) {
    const int64_t i = (int64_t)blockIdx.x * kSmcBlock + threadIdx.x;
    unsigned long long n_eval = 0, n_acc = 0, n_prop = 0;
    if (A.ctrl->done || !A.ctrl->pass_open) return;  // uniform no-op
    const int cur = A.ctrl->cur;
    const bool gather = A.ctrl->use_ridx != 0;
    const uint64_t pass = A.ctrl->pass + 1u;
    const int D = A.D;
    const double* __restrict__ theta_src = A.theta[cur];
    double Xfin = 0.0;
    bool alive_i = false;
    if (i < A.N) {
        // idx = repeat(idxalive, ceil(N/m))[1:N]  (src/smc.jl:146-147), evaluated on the fly
        const bool remap = gather && A.ctrl->resampled != 0;
        const unsigned ess = (unsigned)A.ctrl->ess;
        const int64_t si = remap ? (int64_t)A.cidx[(unsigned)i % ess] : i;
        const double* th = theta_src + si * D;
        double Xi = A.X[cur][si];
        double lpi = A.lpi[cur][si];
        double* dst = A.theta[1 - cur] + i * D;
        bool accepted = false;
        alive_i = A.alive[i] != 0;
        if (alive_i) {
            const uint64_t N = (uint64_t)A.N;
            const uint32_t w = (uint32_t)i;
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
            kabc_normal_pair(kabc_lo64(B1), kabc_hi64(B1), &z0, &z1);
            const double s = A.max_stretch * z0 / kabc_sqrt((double)D);
            const double* ta = theta_src + (remap ? (int64_t)A.cidx[(unsigned)a % ess] : a) * D;
            const double* tb = theta_src + (remap ? (int64_t)A.cidx[(unsigned)b % ess] : b) * D;
            double* prop = A.scratch + (i * 2) * D;
            double* xp = prop + D;
            for (int k = 0; k < D; ++k) {
                const double W = (tb[k] - ta[k]) * s;
                prop[k] = th[k] + W;
            }
            const double lprob = kabc_log(kabc_u01(kabc_lo64(B2)));
            n_prop = 1;
            const double lpp = smc_dyn_logpdf_push(A, prop, xp);
            if (!(lpp < 0.0 && !kabc_isfinite(lpp))) {  // :173
                double lM = lpp - lpi + 0.0;
                if (!(lM < 0.0)) lM = (lM != lM) ? lM : 0.0;
                if (lprob < lM) {
                    kabc_cost_rng_t rng = {A.seed, pass, w, KABC_DOM_SMC_COST, 0u};
                    const double Xp =
                        smc_dyn_cost<COST>(A.cost_id, xp, D, A.cost_params, A.cost_data, A.cost_ndata, &rng);
                    n_eval = 1;
                    const double eps = A.ctrl->eps;
                    const bool reject = A.ctrl->flag ? (Xp > eps) : (Xp >= eps);
                    if (!reject) {
                        for (int k = 0; k < D; ++k) dst[k] = prop[k];
                        Xi = Xp;
                        lpi = lpp;
                        n_acc = 1;
                        accepted = true;
                    }
                }
            }
        }
        if (!accepted)
            for (int k = 0; k < D; ++k) dst[k] = th[k];
        A.X[1 - cur][i] = Xi;
        A.lpi[1 - cur][i] = lpi;
        Xfin = Xi;
    }
    smc_block_stats(A.part, alive_i, Xfin);
    const unsigned long long se = wave_sum(n_eval), sa = wave_sum(n_acc), sp = wave_sum(n_prop);
    if ((threadIdx.x & (kWave - 1)) == 0) {
        unsigned long long* sl = A.slots + (size_t)(blockIdx.x & (kSmcSlots - 1)) * 8;
        if (sa) atomicAdd(&sl[0], sa);
        if (se) atomicAdd(&sl[1], se);
        if (sp) atomicAdd(&sl[2], sp);
    }
}

#ifndef __HIPCC_RTC__  // host side
using SmcDynLaunchFn = void (*)(const SmcDynArgs&, hipStream_t, int init);

template <int COST>
inline void launch_smc_dyn(const SmcDynArgs& a, hipStream_t s, int init) {
    const unsigned grid = (unsigned)((a.N + kSmcBlock - 1) / kSmcBlock);
    if (init) hipLaunchKernelGGL((smc_dyn_init_kernel<COST>), dim3(grid), dim3(kSmcBlock), 0, s, a);
    else hipLaunchKernelGGL((smc_dyn_mcmc_kernel<COST>), dim3(grid), dim3(kSmcBlock), 0, s, a);
}

// (as AisDynLaunch, ais_dyn_kernels.hpp)
struct SmcDynLaunch {
    SmcDynLaunchFn fn = nullptr;
    void* mod_mcmc = nullptr;
    void* mod_init = nullptr;
    SmcDynLaunch() = default;
    SmcDynLaunch(SmcDynLaunchFn f) : fn(f) {}
    SmcDynLaunch(void* mcmc, void* init) : mod_mcmc(mcmc), mod_init(init) {}
    explicit operator bool() const { return fn != nullptr || (mod_mcmc != nullptr && mod_init != nullptr); }
    void operator()(const SmcDynArgs& a, hipStream_t s, int init) const {
        if (fn) {
            fn(a, s, init);
            return;
        }
        const unsigned grid = (unsigned)((a.N + kSmcBlock - 1) / kSmcBlock);
        if (grid == 0) return;
        (void)rtc_launch(init ? mod_init : mod_mcmc, dim3(grid), dim3(kSmcBlock), &a, s);
    }
};
#endif

}  // namespace kabc
