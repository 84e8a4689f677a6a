// ais_kernels.hpp -- the AIS walker-update kernels (gfx950), templated on
// (D, DeviceCost id).  Instantiated per cost id in ais_inst_*.hip.
//
// Replaces, for a whole half-ensemble at once:
//   transition!            src/transition.jl:67-82
//   propose + three moves  src/transition.jl:2-65
//   push_p                 src/types.jl:109-114
//   loglike / accept       src/types.jl:133-157 (kernelized), :166-186 (threshold)
//   Factored logpdf        src/priors.jl:275-281
//   step(init)             src/KissABC.jl:35-64
#pragma once

#include "kabc_device.hpp"

namespace kabc {

struct AisArgs {
    double* x_act;          // active half, GLOBAL rows [rows_act_total][D]
    const double* x_comp;   // complementary half, GLOBAL rows [n_comp][D] (frozen)
    double* lp;             // [rows_owned] logprior of the owned rows
    double* ll;             // [rows_owned] loglikelihood (kernelized) | cost (threshold)
    double* trace;          // optional [rows_owned][D]: push_p(x) after the last transition
    int32_t* dbg;           // optional [rows_owned][nt][6] per-transition records
    DevCounters* counters;
    const double* cost_params;
    const double* cost_data;
    int64_t cost_ndata;
    int64_t row_first;      // first owned row of the active half
    int64_t rows_owned;
    int64_t n_comp;         // rows of the complementary half
    uint64_t seed;
    uint64_t t0;            // transition counter of the first sub-step
    uint32_t id_base;       // global walker id of row 0 of the active half
    int32_t nt;             // ntransitions
    int32_t posterior;      // kabc_posterior_kind_t
    double eps;             // scale | maxcost
    PriorSet prior;
};

struct InitArgs {
    double* x_act;
    double* lp;
    double* ll;
    DevCounters* counters;
    const double* cost_params;
    const double* cost_data;
    int64_t cost_ndata;
    int64_t row_first;
    int64_t rows_owned;
    uint64_t seed;
    uint32_t id_base;
    int32_t posterior;
    int32_t cost_id;
    double eps;
    unsigned long long retry_budget;  // retry_sampling * nparticles (src/KissABC.jl:52)
    PriorSet prior;
    kabc_prior_t raw[KABC_MAX_DIM];
};

constexpr int kAisBlock = 64;  // one wavefront per workgroup: 512 WGs at N/2 = 32768

template <int POSTERIOR_RUNTIME = 0>
__device__ __forceinline__ bool ld_valid(int posterior, double lp, double ll) {
    // is_valid_logdensity: src/types.jl:142 and :175-176
    return posterior == KABC_POSTERIOR_KERNELIZED ? kabc_isfinite(lp + ll)
                                                  : (kabc_isfinite(ll) && kabc_isfinite(lp));
}

// loglike(density, push_p(density, y)) -- src/types.jl:133-140, :166-173
template <int D, int COST>
__device__ __forceinline__ void loglike(const PriorSet& P, int posterior, double eps,
                                        const double* y, const double* cost_params,
                                        const double* cost_data, int64_t ndata,
                                        kabc_cost_rng_t* rng, double& lp, double& ll, bool& ev) {
    double yp[D];
    lp = factored_logpdf_push<D>(P, y, yp);
    ev = kabc_isfinite(lp);
    if (posterior == KABC_POSTERIOR_KERNELIZED) {
        ll = lp;
        if (ev) {
            const double c = eval_cost<COST, D>(yp, cost_params, cost_data, ndata, rng);
            const double q = c / eps;
            ll = -0.5 * (q * q);
        }
    } else {
        ll = -lp;
        if (ev) ll = eval_cost<COST, D>(yp, cost_params, cost_data, ndata, rng);
    }
}

template <int D, int COST>
__global__ void __launch_bounds__(kAisBlock) ais_half_kernel(const AisArgs A) {
    const int64_t r = (int64_t)blockIdx.x * kAisBlock + threadIdx.x;
    unsigned long long n_eval = 0, n_acc = 0;
    int err = 0;
    const bool active = r < A.rows_owned;
    if (active) {
        const int64_t row = A.row_first + r;
        const uint32_t w = A.id_base + (uint32_t)row;
        double x[D];
        load_row<D>(A.x_act + row * D, x);
        double lp = A.lp[r], ll = A.ll[r];
        const uint64_t nc = (uint64_t)A.n_comp;
        const double sq3 = kabc_sqrt(3.0), isq3 = kabc_sqrt(1.0 / 3.0);

        for (int s = 0; s < A.nt; ++s) {
            const uint64_t t = A.t0 + (uint64_t)s;
            const kabc_u128_t B0 = kabc_stream_block(A.seed, w, t, 0u, KABC_DOM_AIS_MOVE);
            const kabc_u128_t B1 = kabc_stream_block(A.seed, w, t, 1u, KABC_DOM_AIS_MOVE);
            // p = rand(rng, (1,1,1,1,2,2,3))
            const uint32_t m7 = (uint32_t)(((uint64_t)B0.w[2] * 7u) >> 32);
            const int move = (m7 < 4u) ? 1 : (m7 < 6u) ? 2 : 3;
            const int64_t a = (int64_t)kabc_index(kabc_lo64(B0), nc);
            int64_t b = -1, c = -1;
            double xa[D];
            load_row<D>(A.x_comp + a * D, xa);
            double y[D];
            double corr = 0.0;
            if (move == 1) {
                // stretch_propose, Z = cdf_g_inv(rand(rng), 3.0)
                const double u = kabc_u01(kabc_hi64(B1));
                const double tz = u * (sq3 - isq3) + isq3;
                const double Z = tz * tz;
#pragma unroll
                for (int k = 0; k < D; ++k) {
                    const double W = (x[k] - xa[k]) * Z;
                    y[k] = xa[k] + W;
                }
                corr = (double)(D - 1) * kabc_log(Z);
            } else {
                const kabc_u128_t B2 = kabc_stream_block(A.seed, w, t, 2u, KABC_DOM_AIS_MOVE);
                b = (int64_t)kabc_index(kabc_lo64(B2), nc - 1u);
                b += (b >= a);
                double xb[D];
                load_row<D>(A.x_comp + b * D, xb);
                if (move == 2) {
                    // de_propose
                    double z[D + 2];
#pragma unroll
                    for (int j = 0; j < (D + 2) / 2; ++j) {
                        const kabc_u128_t Bn =
                            kabc_stream_block(A.seed, w, t, 3u + (uint32_t)j, KABC_DOM_AIS_MOVE);
                        kabc_normal_pair(kabc_lo64(Bn), kabc_hi64(Bn), &z[2 * j], &z[2 * j + 1]);
                    }
                    const double gamma =
                        2.38 / kabc_sqrt((double)(2 * D)) * kabc_exp(z[0] * 0.1);
#pragma unroll
                    for (int k = 0; k < D; ++k) {
                        const double Wk = (xa[k] - xb[k]) * gamma;
                        const double sk = kabc_fabs(xa[k] - xb[k]) + kabc_fabs(x[k] - xb[k]) +
                                          kabc_fabs(xa[k] - x[k]);
                        const double Tk = gamma * sk / 300.0 * z[1 + k];
                        y[k] = x[k] + Wk + Tk;
                    }
                } else {
                    // ais_walk_propose
                    const int64_t lo = a < b ? a : b, hi = a < b ? b : a;
                    c = (int64_t)kabc_index(kabc_hi64(B2), nc - 2u);
                    c += (c >= lo);
                    c += (c >= hi);
                    double xc[D];
                    load_row<D>(A.x_comp + c * D, xc);
                    double z[4];
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const kabc_u128_t Bn =
                            kabc_stream_block(A.seed, w, t, 3u + (uint32_t)j, KABC_DOM_AIS_MOVE);
                        kabc_normal_pair(kabc_lo64(Bn), kabc_hi64(Bn), &z[2 * j], &z[2 * j + 1]);
                    }
#pragma unroll
                    for (int k = 0; k < D; ++k) {
                        const double Xs = (xa[k] + (xb[k] + xc[k])) / 3.0;
                        const double Wk =
                            z[0] * (xa[k] - Xs) + z[1] * (xb[k] - Xs) + z[2] * (xc[k] - Xs);
                        y[k] = x[k] + Wk;
                    }
                }
            }
            // ld = loglike(density, push_p(density, p))
            kabc_cost_rng_t rng = {A.seed, t, w, KABC_DOM_AIS_COST, 0u};
            double nlp, nll;
            bool ev;
            loglike<D, COST>(A.prior, A.posterior, A.eps, y, A.cost_params, A.cost_data,
                             A.cost_ndata, &rng, nlp, nll, ev);
            n_eval += ev ? 1u : 0u;
            // accept(...)
            bool acc = false;
            if (!kabc_isfinite(corr)) err = 1;
            else if (!ld_valid(A.posterior, lp, ll)) err = 2;
            else if (ld_valid(A.posterior, nlp, nll)) {
                const double e = -kabc_log(kabc_u01(kabc_lo64(B1)));  // randexp(rng)
                if (A.posterior == KABC_POSTERIOR_KERNELIZED) {
                    const double lW = corr + (nlp + nll) - (lp + ll);
                    acc = (-e <= lW);
                } else {
                    const double lW = corr + nlp - lp;
                    const double mx = (A.eps > ll) ? A.eps : ll;
                    const double lW2 = mx - nll;
                    acc = (-e <= lW) && (lW2 >= 0.0);
                }
            }
            if (acc) {
#pragma unroll
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
            if (err) break;
        }
        store_row<D>(A.x_act + row * D, x);
        A.lp[r] = lp;
        A.ll[r] = ll;
        if (A.trace) {
            double xp[D];
#pragma unroll
            for (int k = 0; k < D; ++k) xp[k] = A.prior.c[k].discrete ? kabc_rint(x[k]) : x[k];
            store_row<D>(A.trace + r * D, xp);
        }
    }
    // one atomic per wave and counter
    const unsigned long long se = wave_sum(n_eval);
    const unsigned long long sa = wave_sum(n_acc);
    const unsigned long long sp = wave_sum(active ? (unsigned long long)A.nt : 0ull);
    if ((threadIdx.x & (kWave - 1)) == 0) {
        atomicAdd(&A.counters->proposals, sp);
        atomicAdd(&A.counters->cost_evals, se);
        atomicAdd(&A.counters->accepted, sa);
    }
    if (err) atomicMax(&A.counters->error, err);
}

// step(rng, model, spl::AIS; retry_sampling): one thread per owned walker; the
// retry budget is global (src/KissABC.jl:52-60), kept in a device counter.
template <int D>
__global__ void __launch_bounds__(kAisBlock) ais_init_kernel(const InitArgs A) {
    const int64_t r = (int64_t)blockIdx.x * kAisBlock + threadIdx.x;
    if (r >= A.rows_owned) return;
    const int64_t row = A.row_first + r;
    const uint32_t w = A.id_base + (uint32_t)row;
    double x[D], xp[D];
    double lp = 0.0, ll = 0.0;
    uint64_t attempt = 0;
    while (true) {
        for (int k = 0; k < D; ++k) {
            kabc_slotwin_t win = {A.seed, attempt, w, KABC_DOM_AIS_INIT,
                                  (uint32_t)k * KABC_SLOTS_PER_DIM};
            x[k] = kabc_sample_prior(&A.raw[k], &win);
        }
        lp = factored_logpdf_push<D>(A.prior, x, xp);
        kabc_cost_rng_t rng = {A.seed, attempt, w, KABC_DOM_AIS_INIT_COST, 0u};
        if (A.posterior == KABC_POSTERIOR_KERNELIZED) {
            ll = lp;
            if (kabc_isfinite(lp)) {
                const double c = kabc_cost_eval(A.cost_id, xp, D, A.cost_params, A.cost_data,
                                                A.cost_ndata, &rng);
                const double q = c / A.eps;
                ll = -0.5 * (q * q);
            }
        } else {
            ll = -lp;
            if (kabc_isfinite(lp))
                ll = kabc_cost_eval(A.cost_id, xp, D, A.cost_params, A.cost_data, A.cost_ndata,
                                    &rng);
        }
        if (ld_valid(A.posterior, lp, ll)) break;
        const unsigned long long used = atomicAdd(&A.counters->retries, 1ull) + 1ull;
        if (used > A.retry_budget) {
            A.counters->init_failed = 1;
            break;
        }
        ++attempt;
    }
    store_row<D>(A.x_act + row * D, x);
    A.lp[r] = lp;
    A.ll[r] = ll;
}

// launchers (defined by the instantiation units)
using AisLaunchFn = void (*)(const AisArgs&, hipStream_t);
AisLaunchFn find_ais_kernel(int cost_id, int D);
void launch_ais_init(int D, const InitArgs& a, hipStream_t s);

}  // namespace kabc
