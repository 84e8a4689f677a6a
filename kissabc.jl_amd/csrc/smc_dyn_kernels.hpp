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
    int64_t p0, p1;             // the particles of a pass: [p0, p1), p0 a multiple of 64 (a sharded run: this rank's)
};

template <int COST>
__device__ __forceinline__ double smc_dyn_cost(int cost_id, const double* x, int D, const double* params,
                                               const double* data, int64_t ndata, kabc_cost_rng_t* rng) {
#ifdef KABC_USER_COST_DEFINED
    if constexpr (COST == KABC_COST_USER) return kabc_user_cost(x, D, params, data, ndata, rng);
#endif
    // the built-in costs that take any number of parameters, dispatched at compile time: a kernel that carries
    // every built-in cost allocates the registers of the hungriest one (292 against ~150: one wavefront per
    // SIMD instead of three)
    if constexpr (COST == KABC_COST_GAUSS_DIST) return kabc_cost_gauss_dist(x, D, params);
    else if constexpr (COST == KABC_COST_ROSENBROCK) return kabc_cost_rosenbrock(x, D);
    else if constexpr (COST == KABC_COST_HIER_GAUSS_SIM) return kabc_cost_hier_gauss_sim(x, D, data, rng);
    else if constexpr (COST == KABC_COST_NORM_SHELL) return kabc_cost_norm_shell(x, D, params);
    else return kabc_cost_eval(cost_id, x, D, params, data, ndata, rng);
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
    const int64_t i = A.p0 + (int64_t)blockIdx.x * kSmcBlock + threadIdx.x;
    unsigned long long n_eval = 0, n_acc = 0, n_prop = 0;
    if (A.ctrl->done || !A.ctrl->pass_open) return;  // uniform no-op
    const int cur = A.ctrl->cur;
    const bool gather = A.ctrl->use_ridx != 0;
    const uint64_t pass = A.ctrl->pass + 1u;
    const int D = A.D;
    const double* __restrict__ theta_src = A.theta[cur];
    double Xfin = 0.0;
    bool alive_i = false;
    if (i < A.p1) {
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
    smc_block_stats(A.part, alive_i, Xfin, A.p0 / kSmcBlock + (int64_t)blockIdx.x);
    const unsigned long long se = wave_sum(n_eval), sa = wave_sum(n_acc), sp = wave_sum(n_prop);
    if ((threadIdx.x & (kWave - 1)) == 0) {
        unsigned long long* sl = A.slots + (size_t)(blockIdx.x & (kSmcSlots - 1)) * 8;
        if (sa) atomicAdd(&sl[0], sa);
        if (se) atomicAdd(&sl[1], se);
        if (sp) atomicAdd(&sl[2], sp);
    }
}


// ---- the propose / accept pass with a TEAM of T lanes per particle (round 6; as ais_dyn_half_kernel) ----
// Thread-per-particle, a lane walks its particle's D coordinates three times (proposal, push_p + log-density,
// copy) through strided global rows: 16 384 particles x 40 parameters = 256 wavefronts on 1024 SIMDs, 49 us
// per pass = 0.05 of the HBM rate (tools/smc_dyn_probe.py).  Here a particle belongs to T = 8 / 16 / 64 lanes
// of one wavefront: coordinates k = lane, lane + T, ... -- the source row and the two partner rows are read
// coalesced, the proposal, push_p and the component's log-density are per-coordinate work -- and what the
// contract fixes as sequential stays sequential, on the team's lane 0: the left-to-right sum of the
// components' log-densities (src/priors.jl:30-36) over the values the team left in LDS, the prior's
// Metropolis test, the cost (one function of the whole vector) and accept (src/smc.jl:172-186).  The
// proposal's push_p image and the log-densities live in LDS (2 rows of D per particle, dynamic), the
// prepared prior too.  The per-64-particle statistics the selection reads (smc_block_stats) come from a
// second, tiny kernel: a workgroup of this one is a wavefront of 64 / T particles.
__device__ __forceinline__ void smc_dyn_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__host__ __device__ inline int smc_dyn_row(int D) { return (D + 1) & ~1; }

template <int COST, int T>
__global__ void __launch_bounds__(kWave) smc_dyn_team_kernel(const SmcDynArgs A) {
    static_assert(T == 4 || T == 8 || T == 16 || T == 64, "lanes per particle");
    constexpr int kP = kWave / T;  // particles per wavefront = per workgroup
    extern __shared__ __attribute__((aligned(16))) double smc_dyn_lds[];
    const int done0 = A.ctrl->done, open0 = A.ctrl->pass_open;
    const int cur = A.ctrl->cur;
    const bool gather = A.ctrl->use_ridx != 0;
    const int resampled = A.ctrl->resampled;
    const unsigned ess = (unsigned)A.ctrl->ess;
    const uint64_t pass = A.ctrl->pass + 1u;
    const double eps = A.ctrl->eps;
    const int flag = A.ctrl->flag;
    if (done0 || !open0) return;  // uniform no-op
    const int lane = threadIdx.x, team = lane / T, tl = lane - team * T;
    const int D = A.D, Dp = smc_dyn_row(D);
    PriorDev* const sp = reinterpret_cast<PriorDev*>(smc_dyn_lds);
    double* const rows0 = smc_dyn_lds + (size_t)D * (sizeof(PriorDev) / sizeof(double));
    double* const xp = rows0 + (size_t)team * 2 * Dp;  // push_p(proposal)
    double* const lk = xp + Dp;                         // logpdf(p_k, xp_k)
    {
        static_assert(sizeof(PriorDev) % sizeof(double) == 0, "components are staged as doubles");
        const int nw = D * (int)(sizeof(PriorDev) / sizeof(double));
        for (int i = lane; i < nw; i += kWave) smc_dyn_lds[i] = reinterpret_cast<const double*>(A.prior)[i];
    }
    const int64_t i = A.p0 + (int64_t)blockIdx.x * kP + team;
    const bool lead = tl == 0;
    unsigned long long n_eval = 0, n_acc = 0, n_prop = 0;
    const double* __restrict__ theta_src = A.theta[cur];
    if (i < A.p1) {  // (team-uniform)
        const bool remap = gather && resampled != 0;
        const int64_t si = remap ? (int64_t)A.cidx[(unsigned)i % ess] : i;
        const double* th = theta_src + si * D;
        double* dst = A.theta[1 - cur] + i * D;
        double Xi = 0.0, lpi = 0.0;
        if (lead) {
            Xi = A.X[cur][si];
            lpi = A.lpi[cur][si];
        }
        const bool alive_i = A.alive[i] != 0;
        int acc_i = 0;
        const double* ta = th;
        const double* tb = th;
        double s = 0.0;
        if (alive_i) {
            const uint64_t N = (uint64_t)A.N;
            const uint32_t w = (uint32_t)i;
            // blocks 0, 1, 2 of the particle's stream: lane j < 3 of the team expands block j
            kabc_u128_t B0, B1, B2;
            {
                const kabc_u128_t Bm = kabc_stream_block(A.seed, w, pass, tl < 3 ? (uint32_t)tl : 0u, KABC_DOM_SMC_MOVE);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    B0.w[q] = (uint32_t)__shfl((int)Bm.w[q], team * T, kWave);
                    B1.w[q] = (uint32_t)__shfl((int)Bm.w[q], team * T + 1, kWave);
                    B2.w[q] = (uint32_t)__shfl((int)Bm.w[q], team * T + 2, kWave);
                }
            }
            // while a==i ... ; while b==i || b==a ...  (src/smc.jl:163-164)
            int64_t a = (int64_t)kabc_index32(kabc_lo64(B0), (uint32_t)N - 1u);
            a += (a >= i);
            const int64_t lo = a < i ? a : i, hi = a < i ? i : a;
            int64_t b = (int64_t)kabc_index32(kabc_hi64(B0), (uint32_t)N - 2u);
            b += (b >= lo);
            b += (b >= hi);
            ta = theta_src + (remap ? (int64_t)A.cidx[(unsigned)a % ess] : a) * D;
            tb = theta_src + (remap ? (int64_t)A.cidx[(unsigned)b % ess] : b) * D;
            double z0, z1;
            kabc_normal_pair(kabc_lo64(B1), kabc_hi64(B1), &z0, &z1);
            s = A.max_stretch * z0 / kabc_sqrt((double)D);
            // proposal, push_p, the components' log-densities: a coordinate per lane
            for (int k = tl; k < D; k += T) {
                const double W = (tb[k] - ta[k]) * s;
                const double pk = th[k] + W;
                const PriorDev q = sp[k];
                const double v = q.discrete ? kabc_rint(pk) : pk;
                xp[k] = v;
                lk[k] = comp_logpdf_general_body(q.kind, q.p[0], q.p[1], q.p[2], q.p[3], q.c0, q.c1, q.rb, v);
            }
            smc_dyn_lds_fence();
            if (lead) {
                const double lprob = kabc_log(kabc_u01(kabc_lo64(B2)));
                n_prop = 1;
                double sm = lk[0];  // left to right, as logpdf(d::Factored, x) sums
                for (int k = 1; k < D; ++k) sm = sm + lk[k];
                const double lpp = joint_logpdf_or(sm, sp[0].kind, xp, D, sp, kabc_log_tab);
                if (!(lpp < 0.0 && !kabc_isfinite(lpp))) {  // :173
                    double lM = lpp - lpi + 0.0;
                    if (!(lM < 0.0)) lM = (lM != lM) ? lM : 0.0;
                    if (lprob < lM) {
                        kabc_cost_rng_t rng = {A.seed, pass, w, KABC_DOM_SMC_COST, 0u};
                        const double Xp =
                            smc_dyn_cost<COST>(A.cost_id, xp, D, A.cost_params, A.cost_data, A.cost_ndata, &rng);
                        n_eval = 1;
                        const bool reject = flag ? (Xp > eps) : (Xp >= eps);
                        if (!reject) {
                            Xi = Xp;
                            lpi = lpp;
                            n_acc = 1;
                            acc_i = 1;
                        }
                    }
                }
            }
            acc_i = __shfl(acc_i, team * T, kWave);
        }
        if (acc_i) {  // (the proposal again, the same expression: a row of LDS per particle less)
            for (int k = tl; k < D; k += T) {
                const double W = (tb[k] - ta[k]) * s;
                dst[k] = th[k] + W;
            }
        } else {
            for (int k = tl; k < D; k += T) dst[k] = th[k];
        }
        if (lead) {
            A.X[1 - cur][i] = Xi;
            A.lpi[1 - cur][i] = lpi;
        }
    }
    const unsigned long long se = wave_sum(n_eval), sa = wave_sum(n_acc), sp2 = wave_sum(n_prop);
    if (lane == 0) {
        unsigned long long* sl = A.slots + (size_t)(blockIdx.x & (kSmcSlots - 1)) * 8;
        if (sa) atomicAdd(&sl[0], sa);
        if (se) atomicAdd(&sl[1], se);
        if (sp2) atomicAdd(&sl[2], sp2);
    }
}

// count / NaNs / key range of the alive costs of every 64 particles of the pass just made (what the
// thread-per-particle kernels leave with smc_block_stats at their end)
template <int COST>  // (a template only so that every unit that instantiates the pass carries its own copy)
__global__ void __launch_bounds__(kSmcBlock) smc_dyn_part_kernel(const SmcDynArgs A) {
    const int done0 = A.ctrl->done, open0 = A.ctrl->pass_open, cur = A.ctrl->cur;
    if (done0 || !open0) return;
    const int64_t i = A.p0 + (int64_t)blockIdx.x * kSmcBlock + threadIdx.x;
    const bool in = i < A.p1;
    const bool alive_i = in && A.alive[i] != 0;
    const double x = in ? A.X[1 - cur][i] : 0.0;
    smc_block_stats(A.part, alive_i, x, A.p0 / kSmcBlock + (int64_t)blockIdx.x);
}

#ifndef __HIPCC_RTC__  // host side
using SmcDynLaunchFn = void (*)(const SmcDynArgs&, hipStream_t, int init);

inline size_t smc_dyn_lds_bytes(int D, int T) {
    return (size_t)D * sizeof(PriorDev) + (size_t)(kWave / T) * 2 * (size_t)smc_dyn_row(D) * sizeof(double);
}
// lanes per particle: the narrowest team that leaves a wavefront for every SIMD and whose rows fit 64 KB of
// LDS; 0: the thread-per-particle kernel (KABC_SMC_DYN_TEAM=0 / 8 / 16 / 64: A/B runs)
inline int smc_dyn_team(int64_t N, int D) {
    // (measured, us per pass, tools/smc_dyn_probe.py, T = 4 / 8 / 16: 16 384 x 40: 25.5 / 22.4 / 29.8; 131 072 x 40:
    // 89.7 / 105.8 / 163.8; 16 384 x 128: 49.0 / 62.0 / 56.6 -- thread per particle: 48.9 / 442 / 138)
    int T = (N >= 32768 || D >= 96) ? 4 : 8;
    auto waves_ok = [&](int t) {  // a CU's 160 KB hold four wavefronts' rows (measured: 16 384 x 128: 48 us against 58 with eight)
        int want = 4;
        if (const char* e = std::getenv("KABC_DYN_LDS_WAVES")) {
            const int v = std::atoi(e);
            if (v >= 1 && v <= 16) want = v;
        }
        return ((size_t)160 << 10) / smc_dyn_lds_bytes(D, t) >= (size_t)want;
    };
    while (T < kWave && (N * T / kWave < 1024 || smc_dyn_lds_bytes(D, T) > ((size_t)60 << 10) || !waves_ok(T)))
        T = T == 4 ? 8 : T == 8 ? 16 : 64;
    if (const char* e = std::getenv("KABC_SMC_DYN_TEAM")) {
        const int v = e[0] ? std::atoi(e) : -1;
        if (v == 0 || ((v == 4 || v == 8 || v == 16 || v == 64) && smc_dyn_lds_bytes(D, v) <= ((size_t)60 << 10))) T = v;
    }
    if (T && smc_dyn_lds_bytes(D, T) > ((size_t)60 << 10)) T = 0;
    return T;
}

template <int COST, int T>
inline void launch_smc_dyn_team(const SmcDynArgs& a, hipStream_t s) {
    const int64_t n = a.p1 - a.p0;
    if (n <= 0) return;
    hipLaunchKernelGGL((smc_dyn_team_kernel<COST, T>), dim3((unsigned)((n + kWave / T - 1) / (kWave / T))), dim3(kWave),
                       smc_dyn_lds_bytes(a.D, T), s, a);
    hipLaunchKernelGGL((smc_dyn_part_kernel<COST>), dim3((unsigned)((n + kSmcBlock - 1) / kSmcBlock)), dim3(kSmcBlock), 0, s, a);
}

template <int COST>
inline void launch_smc_dyn(const SmcDynArgs& a, hipStream_t s, int init) {
    const unsigned grid = (unsigned)((a.N + kSmcBlock - 1) / kSmcBlock);
    if (init) {
        hipLaunchKernelGGL((smc_dyn_init_kernel<COST>), dim3(grid), dim3(kSmcBlock), 0, s, a);
        return;
    }
    const int64_t n = a.p1 - a.p0;
    if (n <= 0) return;
    switch (smc_dyn_team(n, a.D)) {
        case 4: launch_smc_dyn_team<COST, 4>(a, s); break;
        case 8: launch_smc_dyn_team<COST, 8>(a, s); break;
        case 16: launch_smc_dyn_team<COST, 16>(a, s); break;
        case 64: launch_smc_dyn_team<COST, 64>(a, s); break;
        default:
            hipLaunchKernelGGL((smc_dyn_mcmc_kernel<COST>), dim3((unsigned)((n + kSmcBlock - 1) / kSmcBlock)), dim3(kSmcBlock), 0, s, a);
    }
}

// a host launch function (built-in costs, plugin .so built by hipcc) or the kernels of a run-time compiled
// unit: plugin_registry.hpp kPfSmcDyn, variants 0 the thread-per-particle pass, 1 init, 2 / 3 / 4 the pass with
// teams of 8 / 16 / 64 lanes (a team of 4 becomes 8 there: a unit's compilation time), 5 the statistics
// kernel behind a team pass
struct SmcDynLaunch {
    SmcDynLaunchFn fn = nullptr;
    void* mod_mcmc = nullptr;
    void* mod_init = nullptr;
    void* mod_team[3] = {nullptr, nullptr, nullptr};
    void* mod_part = nullptr;
    SmcDynLaunch() = default;
    SmcDynLaunch(SmcDynLaunchFn f) : fn(f) {}
    SmcDynLaunch(void* mcmc, void* init, void* t8 = nullptr, void* t16 = nullptr, void* t64 = nullptr, void* part = nullptr)
        : mod_mcmc(mcmc), mod_init(init), mod_part(part) {
        mod_team[0] = t8;
        mod_team[1] = t16;
        mod_team[2] = t64;
    }
    explicit operator bool() const { return fn != nullptr || (mod_mcmc != nullptr && mod_init != nullptr); }
    void operator()(const SmcDynArgs& a, hipStream_t s, int init) const {
        if (fn) {
            fn(a, s, init);
            return;
        }
        const int64_t n = init ? a.N : a.p1 - a.p0;  // (init: every particle; a pass: its range)
        const unsigned grid = (unsigned)((n + kSmcBlock - 1) / kSmcBlock);
        if (n <= 0) return;
        int T = init ? 0 : smc_dyn_team(n, a.D);
        if (T == 4) T = smc_dyn_lds_bytes(a.D, 8) <= ((size_t)60 << 10) ? 8 : 0;
        void* team = T == 8 ? mod_team[0] : T == 16 ? mod_team[1] : T == 64 ? mod_team[2] : nullptr;
        if (team && mod_part) {
            (void)rtc_launch_lds(team, dim3((unsigned)((n + kWave / T - 1) / (kWave / T))), dim3(kWave), &a, s,
                                 (unsigned)smc_dyn_lds_bytes(a.D, T));
            (void)rtc_launch(mod_part, dim3(grid), dim3(kSmcBlock), &a, s);
            return;
        }
        (void)rtc_launch(init ? mod_init : mod_mcmc, dim3(grid), dim3(kSmcBlock), &a, s);
    }
};
#endif

}  // namespace kabc
