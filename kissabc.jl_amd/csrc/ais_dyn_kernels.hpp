// ais_dyn_kernels.hpp -- AIS for length(prior) > KABC_MAX_DIM: the same transition!()
// (src/transition.jl:2-82, src/types.jl:27-128) with the dimension as a RUN-TIME value.
//
// The reference puts no bound on length(prior) (src/priors.jl:10-13).  The fast kernels
// (ais_kernels.hpp) keep a walker in registers and are instantiated for D = 1..KABC_MAX_DIM;
// beyond that the walker's row, the proposal and its push_p image live in LDS rows sized at launch,
// the prepared prior in a device array, and a walker belongs to a TEAM of 16 lanes (see the half-
// generation kernel below).  Same draws (the
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

template <int COST>
__device__ __forceinline__ double dyn_cost(int cost_id, const double* x, int D, const double* params,
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

// loglike(density, push_p(density, y)) with y, xp in memory, by ONE thread (step(init))
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
    lp = joint_logpdf_or(s, A.prior[0].kind, xp, D, A.prior, kabc_log_tab);
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

// ---- the half-generation kernel: a TEAM of T lanes per walker ------------------------------------
// Round 6.  Thread-per-walker, every lane walked all three moves' coordinate loops and a DE lane its
// D/2 Box-Muller pairs one after the other (D = 40, 16 384 walkers: 128 wavefronts on 1024 SIMDs,
// 0.92 ms per launch of 20 transitions).  Here a walker belongs to T = 4 .. 64 lanes of one
// wavefront (the host picks T so that the launch has a wavefront for every SIMD: ais_dyn_team):
//   * coordinates k = lane, lane + T, ... of the team: partner rows are read coalesced -- up to
//     kDynInFlight coordinates per lane and row requested at once, BEFORE the move's normals are
//     generated, so the rows' L2 round trip hides under the Philox / Box-Muller arithmetic -- and the
//     proposal, push_p and the component's log-density are per-coordinate work;
//   * the normal pairs of the wavefront's DE / walk moves are ONE list dealt out over all 64 lanes (a
//     pair per lane and round) and handed over through the walkers' LDS rows;
//   * the prepared prior is staged in LDS once per launch (a component is 72 bytes: read from global
//     memory per coordinate and sub-step it was a third of the launch, for a box prior);
//   * what the contract fixes as SEQUENTIAL stays sequential, on the team's lane 0: the left-to-right
//     sum of the components' log-densities (src/priors.jl:30-36) over the values the team left in
//     LDS, the cost -- one function of the whole vector (src/types.jl:55; a user snippet) -- and accept.
// The walker's row x, the proposal y, push_p(y), the components' log-densities and the normals live
// in LDS (5 rows of D + 2 doubles per walker, dynamic: ais_dyn_lds_bytes); a team is part of one
// wavefront, whose LDS traffic completes in order: a hand-over is a compiler fence, not a barrier.
// Same draws, same expressions, same order of every sum: bit-identical to the oracle as before.
constexpr int kDynInFlight = 4;  // coordinates per lane and partner row requested before the first use
__host__ __device__ inline int ais_dyn_row(int D) { return (D + 3) & ~1; }  // >= D + 2 (normals 0 .. D + 1), even
template <int COST, int T>
__global__ void __launch_bounds__(kWave) ais_dyn_half_kernel(const AisDynArgs A) {
    static_assert(T == 4 || T == 8 || T == 16 || T == 32 || T == 64, "lanes per walker");
    constexpr int kWalkers = kWave / T;  // per wavefront = per workgroup
    extern __shared__ __attribute__((aligned(16))) double dyn_lds[];
    const int lane = threadIdx.x, team = lane / T, tl = lane - team * T;
    const int D = A.D, Dp = ais_dyn_row(D);
    PriorDev* const sp = reinterpret_cast<PriorDev*>(dyn_lds);  // [D] prepared components
    double* const rows0 = dyn_lds + (size_t)D * (sizeof(PriorDev) / sizeof(double));
    double* const xs = rows0 + (size_t)team * 5 * Dp;  // the walker's row
    double* const y = xs + Dp;                          // proposal
    double* const xp = y + Dp;                          // push_p(y)
    double* const lk = xp + Dp;                         // logpdf(p_k, xp_k)
    double* const zn = lk + Dp;                         // N(0,1) variates of the move: zn[j], j = 0 .. D
    const int64_t r = (int64_t)blockIdx.x * kWalkers + team;
    const bool active = r < A.rows_owned;  // (team-uniform)
    const bool lead = tl == 0;
    unsigned n_eval = 0, n_acc = 0;
    int err = 0;
    {
        static_assert(sizeof(PriorDev) % sizeof(double) == 0, "components are staged as doubles");
        const int nw = D * (int)(sizeof(PriorDev) / sizeof(double));
        for (int i = lane; i < nw; i += kWave) dyn_lds[i] = reinterpret_cast<const double*>(A.prior)[i];
    }
    if (active) {
        const int64_t row = A.row_first + r;
        const uint32_t w = A.id_base + (uint32_t)row;
        double* const xg = A.x_act + row * D;
        for (int k = tl; k < D; k += T) xs[k] = xg[k];
        double lp = A.lp[r], ll = A.ll[r];  // (every lane holds them; the team's lane 0 decides)
        const uint32_t nc = (uint32_t)A.n_comp;
        if (lead && !ld_valid(A.posterior, lp, ll)) err = 2;
        wave_lds_fence();
        constexpr int KB = kDynInFlight;
        const int nchunk = (D + KB * T - 1) / (KB * T);
        for (int s = 0; s < A.nt; ++s) {
            const uint64_t t = A.t0 + (uint64_t)s;
            // -- the move, its partners, the accept variate: blocks 0, 1, 2 of the stream, ONE Philox
            //    evaluation per wavefront (lane j < 3 of a team expands block j) handed round the team
            kabc_u128_t B0, B1, B2;
            {
                const kabc_u128_t Bm = kabc_stream_block(A.seed, w, t, tl < 3 ? (uint32_t)tl : 0u, KABC_DOM_AIS_MOVE);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    B0.w[i] = (uint32_t)__shfl((int)Bm.w[i], team * T, kWave);
                    B1.w[i] = (uint32_t)__shfl((int)Bm.w[i], team * T + 1, kWave);
                    B2.w[i] = (uint32_t)__shfl((int)Bm.w[i], team * T + 2, kWave);
                }
            }
            const uint32_t m7 = (uint32_t)(((uint64_t)B0.w[2] * 7u) >> 32);  // rand((1,1,1,1,2,2,3))
            const int move = (m7 < 4u) ? 1 : (m7 < 6u) ? 2 : 3;
            const int64_t a = (int64_t)kabc_index32(kabc_lo64(B0), nc);
            int64_t b = -1, c = -1;
            const double* xa = A.x_comp + a * D;
            const double* xb = xa;
            const double* xc = xa;
            if (move >= 2) {
                b = (int64_t)kabc_index32(kabc_lo64(B2), nc - 1u);
                b += (b >= a);
                xb = A.x_comp + b * D;
                if (move == 3) {
                    const int64_t lo = a < b ? a : b, hi = a < b ? b : a;
                    c = (int64_t)kabc_index32(kabc_hi64(B2), nc - 2u);
                    c += (c >= lo);
                    c += (c >= hi);
                    xc = A.x_comp + c * D;
                }
            }
            // -- the partner rows' first chunk is requested now (they come from the frozen half: WHEN they
            //    are read cannot matter), the move's normals are generated while it is in flight
            double va[KB], vb[KB], vc[KB];
            auto fetch = [&](int ch) {
#pragma unroll
                for (int j = 0; j < KB; ++j) {
                    const int kk = (ch * KB + j) * T + tl;
                    const bool in = kk < D;
                    va[j] = in ? xa[kk] : 0.0;
                    vb[j] = (in && move >= 2) ? xb[kk] : 0.0;
                    vc[j] = (in && move == 3) ? xc[kk] : 0.0;
                }
            };
            fetch(0);
            double corr = 0.0, f0 = 0.0, f1 = 0.0, f2 = 0.0;  // the move's scalars: Z | gamma | z0, z1, z2
            // -- the normal pairs of the wavefront's DE / walk moves (pair m of a walker = block 3 + m of its
            //    stream; DE: gamma's and one per coordinate, D + 1 values; walk: three), dealt out over ALL 64
            //    lanes: with a team of 4 and 17 parameters a DE walker's nine pairs were three rounds of
            //    Philox + Box-Muller on its own four lanes while the stretch walkers' lanes idled -- and every
            //    wavefront holds all three moves, so every wavefront paid them.  Now the wavefront's pairs
            //    (about 0.29 (D + 2) / 2 + 0.29 per walker) are one list, a pair per lane and round.
            {
                const unsigned long long de_mask = __ballot(lead && move == 2), wk_mask = __ballot(lead && move == 3);
                const int np_de = (D + 2) / 2;
                int pre[kWalkers + 1];
                pre[0] = 0;
#pragma unroll
                for (int q = 0; q < kWalkers; ++q)
                    pre[q + 1] = pre[q] + (((de_mask >> (q * T)) & 1ull) ? np_de : ((wk_mask >> (q * T)) & 1ull) ? 2 : 0);
                const int total = pre[kWalkers];
                // (the lanes here: the wavefront's active teams = its first lanes)
                const int nlanes = (int)__popcll(__ballot(true));
                for (int item = lane; item < total; item += nlanes) {
                    int tt = 0;
#pragma unroll
                    for (int q = 1; q < kWalkers; ++q) tt += (item >= pre[q]) ? 1 : 0;
                    int base = 0;
#pragma unroll
                    for (int q = 1; q < kWalkers; ++q) base = (q == tt) ? pre[q] : base;
                    const int m = item - base;
                    const uint32_t wt = A.id_base + (uint32_t)(A.row_first + (int64_t)blockIdx.x * kWalkers + tt);
                    double* const znt = rows0 + (size_t)tt * 5 * Dp + 4 * (size_t)Dp;
                    const kabc_u128_t Bn = kabc_stream_block(A.seed, wt, t, 3u + (uint32_t)m, KABC_DOM_AIS_MOVE);
                    double z0, z1;
                    kabc_normal_pair(kabc_lo64(Bn), kabc_hi64(Bn), &z0, &z1);
                    znt[2 * m] = z0;
                    znt[2 * m + 1] = z1;
                }
                wave_lds_fence();
            }
            if (move == 1) {  // stretch_propose  src/transition.jl:51-59
                const double sq3 = kabc_sqrt(3.0), isq3 = kabc_sqrt(1.0 / 3.0);
                const double u = kabc_u01(kabc_hi64(B1));
                const double tz = u * (sq3 - isq3) + isq3;
                f0 = tz * tz;
                corr = (double)(D - 1) * kabc_log_pn(f0);
            } else if (move == 2) {  // de_propose  src/transition.jl:2-22
                f0 = 2.38 / kabc_sqrt((double)(2 * D)) * kabc_exp_bounded(zn[0] * 0.1);
            } else {                 // ais_walk_propose  src/transition.jl:24-43
                f0 = zn[0];
                f1 = zn[1];
                f2 = zn[2];
            }
            for (int ch = 0; ch < nchunk; ++ch) {
                if (ch > 0) fetch(ch);
#pragma unroll
                for (int j = 0; j < KB; ++j) {
                    const int kk = (ch * KB + j) * T + tl;
                    if (kk < D) {
                        const double xk = xs[kk];
                        double yk;
                        if (move == 1) {
                            const double W = (xk - va[j]) * f0;
                            yk = va[j] + W;
                        } else if (move == 2) {
                            const double Wk = (va[j] - vb[j]) * f0;
                            const double sk = kabc_fabs(va[j] - vb[j]) + kabc_fabs(xk - vb[j]) + kabc_fabs(va[j] - xk);
                            const double Tk = kabc_div_rc(f0 * sk, 300.0, 1.0 / 300.0) * zn[1 + kk];
                            yk = xk + Wk + Tk;
                        } else {
                            const double Xs = kabc_div_rc(va[j] + (vb[j] + vc[j]), 3.0, 1.0 / 3.0);
                            const double Wk = f0 * (va[j] - Xs) + f1 * (vb[j] - Xs) + f2 * (vc[j] - Xs);
                            yk = xk + Wk;
                        }
                        y[kk] = yk;
                    }
                }
            }
            // -- push_p and the components' log-densities, a coordinate per lane
            if (A.posterior != KABC_POSTERIOR_COMMON) {
                for (int k = tl; k < D; k += T) {
                    const PriorDev q = sp[k];
                    const double v = q.discrete ? kabc_rint(y[k]) : y[k];
                    xp[k] = v;
                    lk[k] = comp_logpdf_general_body(q.kind, q.p[0], q.p[1], q.p[2], q.p[3], q.c0, q.c1, q.rb, v);
                }
            }
            wave_lds_fence();
            // -- ld = loglike(density, push_p(density, p)) and accept(...), the team's lane 0
            //    (src/transition.jl:75-80; src/types.jl:51-75, :84-104, :117-128)
            int acc_i = 0;
            if (lead) {
                kabc_cost_rng_t rng = {A.seed, t, w, KABC_DOM_AIS_COST, 0u};
                double nlp, nll;
                bool ev;
                if (A.posterior == KABC_POSTERIOR_COMMON) {
                    nlp = 0.0;
                    ev = true;
                    nll = dyn_cost<COST>(A.cost_id, y, D, A.cost_params, A.cost_data, A.cost_ndata, &rng);
                } else {
                    double sm = lk[0];  // left to right, as logpdf(d::Factored, x) sums (src/priors.jl:30-36)
                    for (int k = 1; k < D; ++k) sm = sm + lk[k];
                    nlp = joint_logpdf_or(sm, sp[0].kind, xp, D, sp, kabc_log_tab);
                    ev = kabc_isfinite(nlp);
                    if (A.posterior == KABC_POSTERIOR_KERNELIZED) {
                        nll = nlp;
                        if (ev) {
                            const double cst = dyn_cost<COST>(A.cost_id, xp, D, A.cost_params, A.cost_data, A.cost_ndata, &rng);
                            const double q = kabc_div_rc(cst, A.eps, A.reps);
                            nll = -0.5 * (q * q);
                        }
                    } else {
                        nll = -nlp;
                        if (ev) nll = dyn_cost<COST>(A.cost_id, xp, D, A.cost_params, A.cost_data, A.cost_ndata, &rng);
                    }
                }
                n_eval += ev ? 1u : 0u;
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
                    lp = nlp;
                    ll = nll;
                    n_acc += 1u;
                }
                acc_i = acc ? 1 : 0;
                if (A.dbg) {
                    int32_t* d = A.dbg + (r * A.nt + s) * 6;
                    d[0] = move;
                    d[1] = acc_i;
                    d[2] = (int32_t)a;
                    d[3] = (int32_t)b;
                    d[4] = (int32_t)c;
                    d[5] = ev ? 1 : 0;
                }
            }
            // the verdict goes to the team; accepted: x_i <- y  (src/transition.jl:77-78)
            acc_i = __shfl(acc_i, team * T, kWave);
            if (acc_i)
                for (int k = tl; k < D; k += T) xs[k] = y[k];
            wave_lds_fence();
        }
        for (int k = tl; k < D; k += T) xg[k] = xs[k];
        if (lead) {
            A.lp[r] = lp;
            A.ll[r] = ll;
        }
        if (A.trace)
            for (int k = tl; k < D; k += T)
                A.trace[r * D + k] = (sp[k].discrete && A.posterior != KABC_POSTERIOR_COMMON) ? kabc_rint(xs[k]) : xs[k];
    }
    const unsigned long long se = wave_sum(n_eval), sa = wave_sum(n_acc);
    const unsigned long long na = wave_sum((active && lead) ? 1ull : 0ull);
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
            x[k] = kabc_sample_prior(&A.raw[k], &win);  // (a pointer INTO the array: joint priors, kabc_sampling.h)
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

inline size_t ais_dyn_lds_bytes(int D, int T) {
    return (size_t)D * sizeof(PriorDev) + (size_t)(kWave / T) * 5 * (size_t)ais_dyn_row(D) * sizeof(double);
}
// wavefronts per CU the rows in LDS must leave room for (KABC_DYN_LDS_WAVES: A/B runs)
inline int dyn_lds_waves() {
    if (const char* e = std::getenv("KABC_DYN_LDS_WAVES")) {
        const int v = std::atoi(e);
        if (v >= 1 && v <= 16) return v;
    }
    return 8;
}
// lanes per walker (KABC_DYN_TEAM: A/B runs)
inline int ais_dyn_team(int64_t rows, int D) {
    // Measured (D = 40, 8192 rows per launch, 20 transitions, us per launch): T = 4: 211, 8: 143, 16: 208,
    // 32: 309, 64: 469 -- a wavefront's per-sub-step work is mostly the same whatever T (its Philox blocks, the
    // Box-Muller passes, the sequential sum / cost / accept on the lead lanes), so narrow teams -- more
    // walkers per wavefront -- win until the launch has fewer wavefronts than SIMDs (1024) to run on.
    // ... and until a wavefront's rows fit 60 KB of LDS (5 rows of D per walker: 16 walkers of 200 parameters
    // do not) and a CU's 160 KB hold eight wavefronts' rows (two per SIMD; the kernel's registers allow three):
    // D = 40, 32 768 rows: T = 4 (30 KB, five wavefronts per CU) 0.44 ms, T = 8 (16 KB) 0.34.
    int T = 4;
    while (T < kWave && (rows * T / kWave < 1024 || ais_dyn_lds_bytes(D, T) > ((size_t)60 << 10) ||
                         ((size_t)160 << 10) / ais_dyn_lds_bytes(D, T) < (size_t)dyn_lds_waves()))
        T *= 2;
    if (const char* e = std::getenv("KABC_DYN_TEAM")) {
        const int v = std::atoi(e);
        if ((v == 4 || v == 8 || v == 16 || v == 32 || v == 64) && ais_dyn_lds_bytes(D, v) <= ((size_t)60 << 10)) T = v;
    }
    return T;
}


template <int COST, int T>
inline void launch_ais_dyn_half(const AisDynArgs& a, hipStream_t s) {
    hipLaunchKernelGGL((ais_dyn_half_kernel<COST, T>), dim3((unsigned)((a.rows_owned + kWave / T - 1) / (kWave / T))),
                       dim3(kWave), ais_dyn_lds_bytes(a.D, T), s, a);
}
template <int COST>
inline void launch_ais_dyn(const AisDynArgs& a, hipStream_t s, int init) {
    if (a.rows_owned <= 0) return;
    if (init) {
        hipLaunchKernelGGL((ais_dyn_init_kernel<COST>), dim3((unsigned)((a.rows_owned + kWave - 1) / kWave)), dim3(kWave), 0, s, a);
        return;
    }
    // (a team of T lanes per walker, the walkers' rows and the prior in dynamic LDS)
    switch (ais_dyn_team(a.rows_owned, a.D)) {
        case 4: launch_ais_dyn_half<COST, 4>(a, s); break;
        case 8: launch_ais_dyn_half<COST, 8>(a, s); break;
        case 16: launch_ais_dyn_half<COST, 16>(a, s); break;
        case 32: launch_ais_dyn_half<COST, 32>(a, s); break;
        default: launch_ais_dyn_half<COST, 64>(a, s); break;
    }
}

// a host launch function (built-in costs, plugin .so built by hipcc) or the kernels of a run-time
// compiled unit (a hipRTC user cost, user prior families): plugin_registry.hpp kPfAisDyn, variants
// 0 / 2 / 3 = the half-generation kernel with teams of 16 / 8 / 64 lanes (three of the five sizes: a
// unit's compilation time), 1 = the init kernel
struct AisDynLaunch {
    AisDynLaunchFn fn = nullptr;
    void* mod_half[3] = {nullptr, nullptr, nullptr};
    void* mod_init = nullptr;
    AisDynLaunch() = default;
    AisDynLaunch(AisDynLaunchFn f) : fn(f) {}
    AisDynLaunch(void* h16, void* h8, void* h64, void* init) : mod_init(init) {
        mod_half[0] = h16;
        mod_half[1] = h8;
        mod_half[2] = h64;
    }
    explicit operator bool() const {
        return fn != nullptr || (mod_half[0] != nullptr && mod_half[1] != nullptr && mod_half[2] != nullptr && mod_init != nullptr);
    }
    void operator()(const AisDynArgs& a, hipStream_t s, int init) const {
        if (fn) {
            fn(a, s, init);
            return;
        }
        if (a.rows_owned <= 0) return;
        if (init) {
            (void)rtc_launch(mod_init, dim3((unsigned)((a.rows_owned + kWave - 1) / kWave)), dim3(kWave), &a, s);
            return;
        }
        int T = ais_dyn_team(a.rows_owned, a.D);
        T = T <= 8 ? 8 : T == 16 ? 16 : 64;
        (void)rtc_launch_lds(mod_half[T == 16 ? 0 : T == 8 ? 1 : 2], dim3((unsigned)((a.rows_owned + kWave / T - 1) / (kWave / T))),
                             dim3(kWave), &a, s, (unsigned)ais_dyn_lds_bytes(a.D, T));
    }
};
#endif

}  // namespace kabc
