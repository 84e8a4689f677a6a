// ais_dyn_kernels.hpp -- AIS for length(prior) > KABC_MAX_DIM: the same transition!()
// (src/transition.jl:2-82, src/types.jl:27-128) with the dimension as a RUN-TIME value.
//
// The reference puts no bound on length(prior) (src/priors.jl:10-13).  The fast kernels
// (ais_kernels.hpp) keep a walker in registers and are instantiated for D = 1..KABC_MAX_DIM;
// beyond that this fallback keeps everything in memory: the walker's row in the ensemble, the
// proposal and its push_p image in two scratch rows per walker, the prepared prior in a device
// array.  Thread = walker, 64 per workgroup, one coordinate loop per move.  Same draws (the
// counter-based streams of include/kabc_philox.h), same operation order: bit-identical to the
// oracle like the fast path (tests/test_gpu_dyn_dim.py).  Built-in DeviceCosts are
// dispatched at run time (kabc_cost_eval); COST = KABC_COST_USER instantiates it for a plugin.
#pragma once

#include "ais_kernels.hpp"

namespace kabc {

struct AisDynArgs {
    double* x_act;          // active half, GLOBAL rows [rows_act_total][D]
    const double* x_comp;   // complementary half [n_comp][D] (frozen)
    double* lp;
    double* ll;
    double* trace;          // optional [rows_owned][D]
    int32_t* dbg;           // optional [rows_owned][nt][6]
    double* scratch;        // [rows_owned][2][D]: proposal y, push_p(y)
    DevCounters* counters;
    unsigned long long* slots;
    const double* cost_params;
    const double* cost_data;
    int64_t cost_ndata;
    int64_t row_first, rows_owned, n_comp;
    uint64_t seed, t0;
    uint32_t id_base;
    int32_t nt, posterior, cost_id, D;
    double eps, reps;
    const PriorDev* prior;        // [D] prepared components (device)
    const kabc_prior_t* raw;      // [D] raw components (device; init only)
    unsigned long long retry_budget;
};

// j-th N(0,1) of the move stream of (w, t): blocks 3, 4, ... hold the pairs
struct DynNormals {
    uint64_t seed, t;
    uint32_t w;
    int cur;
    double z0, z1;
    __device__ __forceinline__ double get(int j) {
        const int blk = j >> 1;
        if (blk != cur) {
            const kabc_u128_t B = kabc_stream_block(seed, w, t, 3u + (uint32_t)blk, KABC_DOM_AIS_MOVE);
            kabc_normal_pair(kabc_lo64(B), kabc_hi64(B), &z0, &z1);
            cur = blk;
        }
        return (j & 1) ? z1 : z0;
    }
};

template <int COST>
__device__ __forceinline__ double dyn_cost(int cost_id, const double* x, int D, const double* params,
                                           const double* data, int64_t ndata, kabc_cost_rng_t* rng) {
#ifdef KABC_USER_COST_DEFINED
    if constexpr (COST == KABC_COST_USER) return kabc_user_cost(x, D, params, data, ndata, rng);
#endif
    return kabc_cost_eval(cost_id, x, D, params, data, ndata, rng);
}

// loglike(density, push_p(density, y)) with y, xp in memory
template <int COST>
__device__ __forceinline__ void dyn_loglike(const AisDynArgs& A, const double* y, double* xp,
                                            kabc_cost_rng_t* rng, double& lp, double& ll, bool& ev) {
    const int D = A.D;
    if (A.posterior == KABC_POSTERIOR_COMMON) {
        lp = 0.0;
        ev = true;
        ll = dyn_cost<COST>(A.cost_id, y, D, A.cost_params, A.cost_data, A.cost_ndata, rng);
        return;
    }
    double s = 0.0;
    for (int k = 0; k < D; ++k) {
        const PriorDev q = A.prior[k];
        const double v = q.discrete ? kabc_rint(y[k]) : y[k];
        xp[k] = v;
        const double l = comp_logpdf_general_body(q.kind, q.p[0], q.p[1], q.p[2], q.p[3], q.c0, q.c1, q.rb, v);
        s = (k == 0) ? l : s + l;
    }
    lp = s;
    ev = kabc_isfinite(lp);
    if (A.posterior == KABC_POSTERIOR_KERNELIZED) {
        ll = lp;
        if (ev) {
            const double c = dyn_cost<COST>(A.cost_id, xp, D, A.cost_params, A.cost_data, A.cost_ndata, rng);
            const double q = kabc_div_rc(c, A.eps, A.reps);
            ll = -0.5 * (q * q);
        }
    } else {
        ll = -lp;
        if (ev) ll = dyn_cost<COST>(A.cost_id, xp, D, A.cost_params, A.cost_data, A.cost_ndata, rng);
    }
}

template <int COST>
__global__ void __launch_bounds__(kWave) ais_dyn_half_kernel(const AisDynArgs A) {
    const int64_t r = (int64_t)blockIdx.x * kWave + threadIdx.x;
    const bool active = r < A.rows_owned;
    const int D = A.D;
    unsigned n_eval = 0, n_acc = 0;
    int err = 0;
    if (active) {
        const int64_t row = A.row_first + r;
        const uint32_t w = A.id_base + (uint32_t)row;
        double* x = A.x_act + row * D;
        double* y = A.scratch + (r * 2) * D;
        double* xp = y + D;
        double lp = A.lp[r], ll = A.ll[r];
        const uint32_t nc = (uint32_t)A.n_comp;
        if (!ld_valid(A.posterior, lp, ll)) err = 2;
        for (int s = 0; s < A.nt; ++s) {
            const uint64_t t = A.t0 + (uint64_t)s;
            const kabc_u128_t B0 = kabc_stream_block(A.seed, w, t, 0u, KABC_DOM_AIS_MOVE);
            const kabc_u128_t B1 = kabc_stream_block(A.seed, w, t, 1u, KABC_DOM_AIS_MOVE);
            const uint32_t m7 = (uint32_t)(((uint64_t)B0.w[2] * 7u) >> 32);  // rand((1,1,1,1,2,2,3))
            const int move = (m7 < 4u) ? 1 : (m7 < 6u) ? 2 : 3;
            const int64_t a = (int64_t)kabc_index32(kabc_lo64(B0), nc);
            int64_t b = -1, c = -1;
            const double* xa = A.x_comp + a * D;
            double corr = 0.0;
            if (move == 1) {  // stretch_propose  src/transition.jl:51-59
                const double sq3 = kabc_sqrt(3.0), isq3 = kabc_sqrt(1.0 / 3.0);
                const double u = kabc_u01(kabc_hi64(B1));
                const double tz = u * (sq3 - isq3) + isq3;
                const double Z = tz * tz;
                for (int k = 0; k < D; ++k) {
                    const double W = (x[k] - xa[k]) * Z;
                    y[k] = xa[k] + W;
                }
                corr = (double)(D - 1) * kabc_log_pn(Z);
            } else {
                const kabc_u128_t B2 = kabc_stream_block(A.seed, w, t, 2u, KABC_DOM_AIS_MOVE);
                b = (int64_t)kabc_index32(kabc_lo64(B2), nc - 1u);
                b += (b >= a);
                const double* xb = A.x_comp + b * D;
                DynNormals zn = {A.seed, t, w, -1, 0.0, 0.0};
                if (move == 2) {  // de_propose  src/transition.jl:2-22
                    const double gamma = 2.38 / kabc_sqrt((double)(2 * D)) * kabc_exp_bounded(zn.get(0) * 0.1);
                    for (int k = 0; k < D; ++k) {
                        const double Wk = (xa[k] - xb[k]) * gamma;
                        const double sk = kabc_fabs(xa[k] - xb[k]) + kabc_fabs(x[k] - xb[k]) +
                                          kabc_fabs(xa[k] - x[k]);
                        const double Tk = kabc_div_rc(gamma * sk, 300.0, 1.0 / 300.0) * zn.get(1 + k);
                        y[k] = x[k] + Wk + Tk;
                    }
                } else {  // ais_walk_propose  src/transition.jl:24-43
                    const int64_t lo = a < b ? a : b, hi = a < b ? b : a;
                    c = (int64_t)kabc_index32(kabc_hi64(B2), nc - 2u);
                    c += (c >= lo);
                    c += (c >= hi);
                    const double* xc = A.x_comp + c * D;
                    const double z0 = zn.get(0), z1 = zn.get(1), z2 = zn.get(2);
                    for (int k = 0; k < D; ++k) {
                        const double Xs = kabc_div_rc(xa[k] + (xb[k] + xc[k]), 3.0, 1.0 / 3.0);
                        const double Wk = z0 * (xa[k] - Xs) + z1 * (xb[k] - Xs) + z2 * (xc[k] - Xs);
                        y[k] = x[k] + Wk;
                    }
                }
            }
            // ld = loglike(density, push_p(density, p))   src/transition.jl:75
            kabc_cost_rng_t rng = {A.seed, t, w, KABC_DOM_AIS_COST, 0u};
            double nlp, nll;
            bool ev;
            dyn_loglike<COST>(A, y, xp, &rng, nlp, nll, ev);
            n_eval += ev ? 1u : 0u;
            // accept(...)  src/types.jl:62-75, :96-104, :123-128
            bool acc = false;
            if (!kabc_isfinite(corr)) err = err ? err : 1;
            else if (ld_valid(A.posterior, nlp, nll)) {
                const double e = -kabc_log_pn(kabc_u01(kabc_lo64(B1)));  // randexp(rng)
                if (A.posterior == KABC_POSTERIOR_KERNELIZED) {
                    const double lW = corr + (nlp + nll) - (lp + ll);
                    acc = (-e <= lW);
                } else if (A.posterior == KABC_POSTERIOR_COMMON) {
                    const double lW = corr + nll - ll;
                    acc = (-e <= lW);
                } else {
                    const double lW = corr + nlp - lp;
                    const double mx = (A.eps > ll) ? A.eps : ll;
                    const double lW2 = mx - nll;
                    acc = (-e <= lW) && (lW2 >= 0.0);
                }
            }
            if (acc) {
                for (int k = 0; k < D; ++k) x[k] = y[k];
                lp = nlp;
                ll = nll;
                n_acc += 1u;
            }
            if (A.dbg) {
                int32_t* d = A.dbg + (r * A.nt + s) * 6;
                d[0] = move;
                d[1] = acc ? 1 : 0;
                d[2] = (int32_t)a;
                d[3] = (int32_t)b;
                d[4] = (int32_t)c;
                d[5] = ev ? 1 : 0;
            }
        }
        A.lp[r] = lp;
        A.ll[r] = ll;
        if (A.trace)
            for (int k = 0; k < D; ++k)
                A.trace[r * D + k] = (A.prior[k].discrete && A.posterior != KABC_POSTERIOR_COMMON)
                                         ? kabc_rint(x[k]) : x[k];
    }
    const unsigned long long se = wave_sum(n_eval), sa = wave_sum(n_acc);
    const unsigned long long na = wave_sum(active ? 1ull : 0ull);
    if (threadIdx.x == 0) {
        unsigned long long* sl = A.slots + (size_t)(blockIdx.x & (kCounterSlots - 1)) * 8;
        atomicAdd(&sl[0], na * (unsigned long long)A.nt);
        atomicAdd(&sl[1], se);
        atomicAdd(&sl[2], sa);
    }
    if (err) atomicMax(&A.counters->error, err);
}

// step(init) -- src/KissABC.jl:35-64 -- with the dimension at run time
template <int COST>
__global__ void __launch_bounds__(kWave) ais_dyn_init_kernel(const AisDynArgs A) {
    const int64_t r = (int64_t)blockIdx.x * kWave + threadIdx.x;
    if (r >= A.rows_owned) return;
    const int D = A.D;
    const int64_t row = A.row_first + r;
    const uint32_t w = A.id_base + (uint32_t)row;
    double* x = A.x_act + row * D;
    double* xp = A.scratch + (r * 2) * D;
    double lp = 0.0, ll = 0.0;
    uint64_t attempt = 0;
    while (true) {
        for (int k = 0; k < D; ++k) {
            kabc_slotwin_t win = {A.seed, attempt, w, KABC_DOM_AIS_INIT, (uint32_t)k * KABC_SLOTS_PER_DIM};
            const kabc_prior_t pr = A.raw[k];
            x[k] = kabc_sample_prior(&pr, &win);
        }
        kabc_cost_rng_t rng = {A.seed, attempt, w, KABC_DOM_AIS_INIT_COST, 0u};
        bool ev;
        dyn_loglike<COST>(A, x, xp, &rng, lp, ll, ev);
        if (ld_valid(A.posterior, lp, ll)) break;
        const unsigned long long used = atomicAdd(&A.counters->retries, 1ull) + 1ull;
        if (used > A.retry_budget) {
            A.counters->init_failed = 1;
            break;
        }
        ++attempt;
    }
    A.lp[r] = lp;
    A.ll[r] = ll;
}

#ifndef __HIPCC_RTC__  // host side
using AisDynLaunchFn = void (*)(const AisDynArgs&, hipStream_t, int init);

template <int COST>
inline void launch_ais_dyn(const AisDynArgs& a, hipStream_t s, int init) {
    const unsigned grid = (unsigned)((a.rows_owned + kWave - 1) / kWave);
    if (grid == 0) return;
    if (init) hipLaunchKernelGGL((ais_dyn_init_kernel<COST>), dim3(grid), dim3(kWave), 0, s, a);
    else hipLaunchKernelGGL((ais_dyn_half_kernel<COST>), dim3(grid), dim3(kWave), 0, s, a);
}

// a host launch function (built-in costs, plugin .so built by hipcc) or the pair of kernels of a
// run-time compiled unit (a hipRTC user cost, user prior families): plugin_registry.hpp kPfAisDyn
struct AisDynLaunch {
    AisDynLaunchFn fn = nullptr;
    void* mod_half = nullptr;
    void* mod_init = nullptr;
    AisDynLaunch() = default;
    AisDynLaunch(AisDynLaunchFn f) : fn(f) {}
    AisDynLaunch(void* half, void* init) : mod_half(half), mod_init(init) {}
    explicit operator bool() const { return fn != nullptr || (mod_half != nullptr && mod_init != nullptr); }
    void operator()(const AisDynArgs& a, hipStream_t s, int init) const {
        if (fn) {
            fn(a, s, init);
            return;
        }
        const unsigned grid = (unsigned)((a.rows_owned + kWave - 1) / kWave);
        if (grid == 0) return;
        (void)rtc_launch(init ? mod_init : mod_half, dim3(grid), dim3(kWave), &a, s);
    }
};
#endif

}  // namespace kabc
