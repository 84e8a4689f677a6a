// abcde_kernels.hpp -- gfx950 kernels for ABCDE (src/smc.jl:347-430).
// Not a hot path of the round: one thread per particle, D as the only template
// parameter (the DeviceCost is dispatched at run time by kabc_cost_eval, which a
// user plugin extends with its own branch).
#pragma once

#include "kabc_device.hpp"
#ifndef __HIPCC_RTC__
#include "launcher.hpp"
#endif

namespace kabc {

struct AbcdeCtrl {
    double eps_l, eps_h, eps_pop;
    int32_t cur;         // buffer set holding θs, Δs, logπ
    int32_t done;        // earlystop break (:379-381)
    int32_t error;       // 1: initial sampling never produced a finite (Δ, logπ)
    int32_t pad;
    long long iters;
    unsigned long long nsims;
};

struct AbcdeArgs {
    double* theta[2];
    double* delta[2];
    double* lpi[2];
    AbcdeCtrl* ctrl;
    const double* cost_params;
    const double* cost_data;
    int64_t cost_ndata;
    int64_t N;
    uint64_t seed;
    int32_t cost_id;
    int32_t earlystop;
    uint32_t dom_init, dom_init_cost;  // stream domains of the initial draws (ABCDE / pfilter)
    double eps_target;
    double alpha;
    double gamma;  // proposal_width * 2.38 / sqrt(2 * length(prior))  (:370)
    PriorSet prior;
    kabc_prior_t raw[KABC_MAX_DIM];
    // Rank structure of the current generation (capi_abcde.hip, large N): the costs in ascending
    // order and a wavelet matrix over the particle indices in that order.  It answers the
    // reference's donor draw  s = rand((1:N)[Δs .<= Δs[i]])  (:392) -- "the m-th smallest index
    // among the particles whose cost does not exceed mine" -- in O(log N) instead of two O(N)
    // scans per particle.  NULL = scan (small N).
    const double* sorted_delta;          // [N]
    const unsigned long long* wm_bits;   // [levels][wm_words]: bit p of level b = bit b of the
                                         // p-th index of the sequence entering that level
    const unsigned* wm_cnt;              // [levels][wm_words]: ones before each word
    const unsigned* wm_nz;               // [levels]: zeros of the level
    int32_t wm_levels;
    int64_t wm_words;
    // donor index of every particle, drawn by abcde_donor_kernel (256 <= N < 4096: sixteen lanes
    // share one particle's scans); NULL = the generation kernel scans itself
    const int32_t* donor;
    // 1: abcde_extrema_kernel first flips the buffer set (θs = nθs of the generation before,
    // :413-415) -- one launch per generation less than a flip kernel of its own
    int32_t flip_first;
    // length(prior) > KABC_MAX_DIM (kernels instantiated with D = 0): the dimension and the
    // prior as device arrays; rows and proposals then live in per-thread arrays of
    // KABC_MAX_DIM_DYN doubles (scratch memory) -- a fallback, several times slower per particle
    int32_t D_rt;
    const PriorDev* dprior;      // [D_rt] prepared components
    const kabc_prior_t* draw;    // [D_rt] raw components (prior sampling)
};

// ---- helpers of the kernels that exist both with a compile-time D (1..KABC_MAX_DIM) and,
// instantiated with D = 0, with a run-time one (ABCDE, pfilter) -------------------------------
template <int D>
struct DimOf {
    static constexpr int cap = D ? D : KABC_MAX_DIM_DYN;
    __device__ __forceinline__ static int get(int d_rt) { return D ? D : d_rt; }
};
template <int D>
__device__ __forceinline__ void load_row_n(const double* __restrict__ p, double* out, int n) {
    if constexpr (D != 0) load_row<D>(p, out);
    else
        for (int k = 0; k < n; ++k) out[k] = p[k];
}
template <int D>
__device__ __forceinline__ void store_row_n(double* __restrict__ p, const double* v, int n) {
    if constexpr (D != 0) store_row<D>(p, v);
    else
        for (int k = 0; k < n; ++k) p[k] = v[k];
}
// push_p + logpdf(d::Factored, x): compile-time D from the by-value PriorSet, run-time D from
// the device array (same formulas: comp_logpdf_general_body)
template <int D>
__device__ __forceinline__ double logpdf_push_n(const PriorSet& P, const PriorDev* dP, int n,
                                                const double* x, double* xp) {
    if constexpr (D != 0) return factored_logpdf_push<D>(P, x, xp);
    else {
        double s = 0.0;
        for (int k = 0; k < n; ++k) {
            const PriorDev q = dP[k];
            const double v = q.discrete ? kabc_rint(x[k]) : x[k];
            xp[k] = v;
            const double l = comp_logpdf_general_body(q.kind, q.p[0], q.p[1], q.p[2], q.p[3], q.c0, q.c1, q.rb, v);
            s = (k == 0) ? l : s + l;
        }
        return joint_logpdf_or(s, dP[0].kind, xp, n, dP, kabc_log_tab);
    }
}

// k-th smallest (0-based) of the first c indices of the cost-sorted order
__device__ __forceinline__ unsigned wm_quantile(const AbcdeArgs& A, unsigned c, unsigned k) {
    unsigned l = 0, r = c, val = 0;
    for (int b = A.wm_levels - 1; b >= 0; --b) {
        const unsigned long long* bits = A.wm_bits + (size_t)b * A.wm_words;
        const unsigned* cnt = A.wm_cnt + (size_t)b * A.wm_words;
        const unsigned ol = cnt[l >> 6] + (unsigned)__popcll(bits[l >> 6] & ((1ull << (l & 63u)) - 1ull));
        const unsigned orr = cnt[r >> 6] + (unsigned)__popcll(bits[r >> 6] & ((1ull << (r & 63u)) - 1ull));
        const unsigned zl = l - ol, zr = r - orr, zeros = zr - zl;
        if (k < zeros) {
            l = zl;
            r = zr;
        } else {
            k -= zeros;
            const unsigned nz = A.wm_nz[b];
            l = nz + ol;
            r = nz + orr;
            val |= 1u << b;
        }
    }
    return val;
}

constexpr int kAbcdeBlock = 64;
constexpr int kAbcdeScanMax = 4096;   // particles whose costs the generation kernel stages in LDS (32 KB)
constexpr unsigned kAbcdeMaxInitTries = 100000u;

// θs, logπ, Δs with the re-draw loop of :351-366
template <int DT>
__global__ void __launch_bounds__(kAbcdeBlock) abcde_init_kernel(const AbcdeArgs A) {
    const int64_t i = (int64_t)blockIdx.x * kAbcdeBlock + threadIdx.x;
    if (i >= A.N) return;
    constexpr int CAP = DimOf<DT>::cap;
    const int D = DimOf<DT>::get(A.D_rt);
    double x[CAP], xp[CAP];
    double lp = 0.0, dl = 0.0;
    for (unsigned attempt = 0;; ++attempt) {
        for (int k = 0; k < D; ++k) {
            kabc_slotwin_t win = {A.seed, (uint64_t)attempt, (uint32_t)i, A.dom_init,
                                  (uint32_t)k * KABC_SLOTS_PER_DIM};
            // (a pointer INTO the components' array: a joint prior's sampler reaches component 0 from component k)
            const kabc_prior_t* pr = DT ? &A.raw[k] : &A.draw[k];
            x[k] = kabc_sample_prior(pr, &win);
        }
        lp = logpdf_push_n<DT>(A.prior, A.dprior, D, x, xp);
        kabc_cost_rng_t rng = {A.seed, (uint64_t)attempt, (uint32_t)i, A.dom_init_cost, 0u};
        // first pass: the cost is only evaluated when logπ is finite (:357-359);
        // in the re-draw loop it always is (:364).  cost(θ.x): NOT push_p'ed.
        const bool eval = (attempt > 0) || kabc_isfinite(lp);
        dl = eval ? kabc_cost_eval(A.cost_id, x, D, A.cost_params, A.cost_data, A.cost_ndata, &rng)
                  : KABC_NAN;
        if (kabc_isfinite(dl) && kabc_isfinite(lp)) break;
        if (attempt >= kAbcdeMaxInitTries) {
            A.ctrl->error = 1;
            break;
        }
    }
    store_row_n<DT>(A.theta[0] + i * D, x, D);
    A.delta[0][i] = dl;
    A.lpi[0][i] = lp;
}

#ifdef KABC_ABCDE_SINGLE_UNIT  // non-template kernels: defined once, in capi_abcde.hip
// ϵ_l, ϵ_h = extrema(Δs); earlystop break; ϵ_pop (:377-382).  One workgroup.
__global__ void __launch_bounds__(1024) abcde_extrema_kernel(const AbcdeArgs A) {
    __shared__ double smin[16], smax[16];
    if (A.ctrl->done) return;
    const int cur = A.ctrl->cur ^ (A.flip_first ? 1 : 0);  // (every thread: nobody writes ctrl before the end)
    const double* dl = A.delta[cur];
    double mn = KABC_INF, mx = -KABC_INF;
    for (int64_t i = threadIdx.x; i < A.N; i += 1024) {
        const double v = dl[i];
        mn = v < mn ? v : mn;
        mx = v > mx ? v : mx;
    }
    for (int off = kWave / 2; off > 0; off >>= 1) {
        const double a = __shfl_down(mn, off, kWave), b = __shfl_down(mx, off, kWave);
        mn = a < mn ? a : mn;
        mx = b > mx ? b : mx;
    }
    if ((threadIdx.x & 63) == 0) {
        smin[threadIdx.x >> 6] = mn;
        smax[threadIdx.x >> 6] = mx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w) {
            mn = smin[w] < mn ? smin[w] : mn;
            mx = smax[w] > mx ? smax[w] : mx;
        }
        A.ctrl->cur = cur;
        A.ctrl->eps_l = mn;
        A.ctrl->eps_h = mx;
        A.ctrl->iters += 1;  // iters += 1 (:373): counted before the earlystop break (:379-381), as the reference does
        if (A.earlystop && mx <= A.eps_target) {
            A.ctrl->done = 1;
        } else {
            const double pop = mn + A.alpha * (mx - mn);
            A.ctrl->eps_pop = (A.eps_target > pop) ? A.eps_target : pop;  // max(ϵ_target, ...)
        }
    }
}

// s = rand(trng, (1:N)[Δs .<= Δs[i]])  (:392) for every particle that needs one, by TEAMS of
// sixteen lanes: each lane counts the hits of its sixteenth of the costs (LDS, eight per step), a
// team-wide prefix locates the lane holding the m-th hit, that lane finds it.  The generation
// kernel's own scans are one lane per particle: 2 N dependent steps (N = 2000: 90 us per
// generation on 32 wavefronts; with teams ~5 us).  Same draw, same index.
constexpr int kDonorTeam = 16, kDonorBlock = 256;
__global__ void __launch_bounds__(kDonorBlock) abcde_donor_kernel(const AbcdeArgs A, int32_t* donor) {
    __shared__ double s_dl[kAbcdeScanMax];
    if (A.ctrl->done) return;
    const int64_t N = A.N;
    const uint64_t g = (uint64_t)A.ctrl->iters;
    const double* __restrict__ DL = A.delta[A.ctrl->cur];
    for (int64_t j = threadIdx.x; j < N; j += kDonorBlock) s_dl[j] = DL[j];
    __syncthreads();
    const int t = threadIdx.x % kDonorTeam;
    const int64_t i = (int64_t)blockIdx.x * (kDonorBlock / kDonorTeam) + threadIdx.x / kDonorTeam;
    bool need = false;  // (the same in the sixteen lanes of a team)
    double di = 0.0;
    if (i < N) {
        di = s_dl[i];
        const bool skip = A.earlystop && di <= A.eps_target;                        // :384-386
        const double eps = (di <= A.eps_target) ? A.eps_target : A.ctrl->eps_pop;   // :390
        need = !skip && di > eps;
    }
    const int64_t per = (N + kDonorTeam - 1) / kDonorTeam;
    const int64_t lo = (int64_t)t * per < N ? (int64_t)t * per : N;
    const int64_t hi = lo + per < N ? lo + per : N;
    constexpr int U = 8;
    int c = 0;
    if (need) {
        int64_t j = lo;
        for (; j + U <= hi; j += U) {
            double v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = s_dl[j + u];
#pragma unroll
            for (int u = 0; u < U; ++u) c += (v[u] <= di) ? 1 : 0;
        }
        for (; j < hi; ++j) c += (s_dl[j] <= di) ? 1 : 0;
    }
    int incl = c;  // inclusive prefix over the team
#pragma unroll
    for (int off = 1; off < kDonorTeam; off <<= 1) {
        const int o = __shfl_up(incl, off, kDonorTeam);
        if (t >= off) incl += o;
    }
    const int total = __shfl(incl, kDonorTeam - 1, kDonorTeam);
    int found = -1;
    if (need) {  // total >= 1: the particle's own cost counts
        const kabc_u128_t B0 = kabc_stream_block(A.seed, (uint32_t)i, g, 0u, KABC_DOM_ABCDE_MOVE);
        int64_t m = (int64_t)kabc_index(kabc_lo64(B0), (uint64_t)total);
        const int excl = incl - c;
        if (m >= excl && m < incl) {  // this lane's range holds the m-th hit
            m -= excl;
            for (int64_t j = lo; j < hi && found < 0; ++j) {
                if (s_dl[j] <= di) {
                    if (m == 0) found = (int)j;
                    --m;
                }
            }
        }
    }
#pragma unroll
    for (int off = kDonorTeam / 2; off > 0; off >>= 1) {
        const int o = __shfl_xor(found, off, kDonorTeam);
        found = o > found ? o : found;
    }
    if (t == 0 && i < N) donor[i] = need ? found : (int32_t)i;
}

#endif  // KABC_ABCDE_SINGLE_UNIT

// one generation (:383-412); reads buffer cur, writes buffer 1-cur
template <int DT>
__global__ void __launch_bounds__(kAbcdeBlock) abcde_gen_kernel(const AbcdeArgs A) {
    const int64_t i = (int64_t)blockIdx.x * kAbcdeBlock + threadIdx.x;
    constexpr int CAP = DimOf<DT>::cap;
    const int D = DimOf<DT>::get(A.D_rt);
    if (A.ctrl->done) return;
    const int cur = A.ctrl->cur;
    const uint64_t g = (uint64_t)A.ctrl->iters;
    const double* __restrict__ TH = A.theta[cur];
    const double* __restrict__ DL = A.delta[cur];
    const double* __restrict__ LP = A.lpi[cur];
    unsigned long long sims = 0;
    // scan path (no rank structure: N < 4096): the generation's costs in LDS -- every particle
    // walks all of them twice for its donor draw; from global memory those were two dependent
    // chains of N loads per thread (1000 particles: 167 us per generation, 16 us from LDS)
    __shared__ double s_dl[kAbcdeScanMax];
    const bool lds_scan = !A.sorted_delta && !A.donor && A.N <= (int64_t)kAbcdeScanMax;  // uniform
    if (lds_scan) {
        for (int64_t j = threadIdx.x; j < A.N; j += kAbcdeBlock) s_dl[j] = DL[j];
        __syncthreads();
    }
    if (i < A.N) {
        const int64_t N = A.N;
        double th[CAP];
        load_row_n<DT>(TH + i * D, th, D);
        double di = DL[i], li = LP[i];
        const bool skip = A.earlystop && di <= A.eps_target;  // :384-386
        if (!skip) {
            const uint32_t w = (uint32_t)i;
            const kabc_u128_t B0 = kabc_stream_block(A.seed, w, g, 0u, KABC_DOM_ABCDE_MOVE);
            const kabc_u128_t B1 = kabc_stream_block(A.seed, w, g, 1u, KABC_DOM_ABCDE_MOVE);
            int64_t s = i;
            const double eps = (di <= A.eps_target) ? A.eps_target : A.ctrl->eps_pop;  // :390
            if (di > eps && A.sorted_delta) {
                // s = rand(trng, (1:N)[Δs .<= Δs[i]])  (:392) through the rank structure:
                // c = #{Δ <= Δ_i} by bisection of the sorted costs, then the m-th smallest
                // index among the first c entries of the cost-sorted order
                int64_t lo = 0, hi = N;  // first position whose cost exceeds di
                while (lo < hi) {
                    const int64_t mid = (lo + hi) >> 1;
                    if (A.sorted_delta[mid] <= di) lo = mid + 1;
                    else hi = mid;
                }
                const int64_t m = (int64_t)kabc_index(kabc_lo64(B0), (uint64_t)lo);
                s = (int64_t)wm_quantile(A, (unsigned)lo, (unsigned)m);
            } else if (di > eps && A.donor) {
                s = (int64_t)A.donor[i];  // (abcde_donor_kernel: the same draw, scanned by a team)
            } else if (di > eps) {
                // the same by two scans (small N): the m-th index, in ascending order, whose
                // cost does not exceed ours
                // (two inlined instances, so that the LDS one reads with ds_read, not flat loads)
                auto donor = [&](const double* __restrict__ dl) __attribute__((always_inline)) {
                    // eight costs per step, loaded together: a step then costs one memory latency,
                    // not eight (the element-by-element loops were chains of dependent loads)
                    constexpr int U = 8;
                    const int64_t NU = N - N % U;
                    int c = 0;
                    for (int64_t j = 0; j < NU; j += U) {
                        double v[U];
#pragma unroll
                        for (int u = 0; u < U; ++u) v[u] = dl[j + u];
#pragma unroll
                        for (int u = 0; u < U; ++u) c += (v[u] <= di) ? 1 : 0;
                    }
                    for (int64_t j = NU; j < N; ++j) c += (dl[j] <= di) ? 1 : 0;
                    int64_t m = (int64_t)kabc_index(kabc_lo64(B0), (uint64_t)c);
                    bool found = false;
                    for (int64_t j = 0; j < NU && !found; j += U) {
                        double v[U];
#pragma unroll
                        for (int u = 0; u < U; ++u) v[u] = dl[j + u];
                        int cnt = 0;
#pragma unroll
                        for (int u = 0; u < U; ++u) cnt += (v[u] <= di) ? 1 : 0;
                        if (m < cnt) {  // the m-th hit is in this group
#pragma unroll
                            for (int u = 0; u < U; ++u) {
                                if (!found && v[u] <= di) {
                                    if (m == 0) {
                                        s = j + u;
                                        found = true;
                                    }
                                    --m;
                                }
                            }
                        } else {
                            m -= cnt;
                        }
                    }
                    for (int64_t j = NU; j < N && !found; ++j) {
                        if (dl[j] <= di) {
                            if (m == 0) {
                                s = j;
                                found = true;
                            }
                            --m;
                        }
                    }
                };
                if (lds_scan) donor(s_dl);
                else donor(DL);
            }
            // while a == s ... ; while b == a || b == s ...  (:394-401)
            int64_t a = (int64_t)kabc_index(kabc_hi64(B0), (uint64_t)(N - 1));
            a += (a >= s);
            const int64_t lo = a < s ? a : s, hi = a < s ? s : a;
            int64_t b = (int64_t)kabc_index(kabc_lo64(B1), (uint64_t)(N - 2));
            b += (b >= lo);
            b += (b >= hi);
            double ts[CAP], ta[CAP], tb[CAP], tp[CAP], xp[CAP];
            load_row_n<DT>(TH + s * D, ts, D);
            load_row_n<DT>(TH + a * D, ta, D);
            load_row_n<DT>(TH + b * D, tb, D);
#pragma unroll
            for (int k = 0; k < D; ++k) tp[k] = ts[k] + (ta[k] - tb[k]) * A.gamma;  // :402
            const double lpp = logpdf_push_n<DT>(A.prior, A.dprior, D, tp, xp);
            const double wp = lpp - li;
            double mn = wp;
            if (!(wp < 0.0)) mn = (wp != wp) ? wp : 0.0;  // min(0, w_prior), NaN propagates
            const double lu = kabc_log_pn(kabc_u01(kabc_hi64(B1)));
            if (!(lu > mn)) {  // log(rand) > min(0,w_prior) && continue  (:405)
                sims = 1;
                kabc_cost_rng_t rng = {A.seed, g, w, KABC_DOM_ABCDE_COST, 0u};
                const double dp = kabc_cost_eval(A.cost_id, tp, D, A.cost_params, A.cost_data,
                                                 A.cost_ndata, &rng);  // cost(θp.x), :408
                const double thr = (eps > di) ? eps : di;  // max(ϵ, Δs[i])
                if (dp <= thr) {
                    di = dp;
                    li = lpp;
#pragma unroll
                    for (int k = 0; k < D; ++k) th[k] = tp[k];
                }
            }
        }
        store_row_n<DT>(A.theta[1 - cur] + i * D, th, D);
        A.delta[1 - cur][i] = di;
        A.lpi[1 - cur][i] = li;
    }
    const unsigned long long ssum = wave_sum(sims);
    if ((threadIdx.x & 63) == 0 && ssum) atomicAdd(&A.ctrl->nsims, ssum);
}

#ifdef KABC_ABCDE_SINGLE_UNIT
// θs = nθs ... (:413-415)
__global__ void abcde_flip_kernel(AbcdeCtrl* ctrl) {
    if (!ctrl->done) ctrl->cur ^= 1;
}

struct AbcdeFinalArgs {
    const double* theta[2];
    const double* delta[2];
    const AbcdeCtrl* ctrl;
    double* out;
    double* dout;
    int64_t N;
    int32_t D;
    PriorSet prior;
    const PriorDev* dprior;  // [D] on the device when D > KABC_MAX_DIM (else NULL: `prior`)
};
__global__ void __launch_bounds__(256) abcde_final_kernel(const AbcdeFinalArgs A) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= A.N) return;
    const int cur = A.ctrl->cur;
    for (int k = 0; k < A.D; ++k) {
        const double v = A.theta[cur][i * A.D + k];
        const bool disc = A.dprior ? (A.dprior[k].discrete != 0) : (A.prior.c[k].discrete != 0);
        A.out[i * A.D + k] = disc ? kabc_rint(v) : v;
    }
    A.dout[i] = A.delta[cur][i];
}

#endif  // KABC_ABCDE_SINGLE_UNIT

#ifndef __HIPCC_RTC__  // host side
using AbcdeLaunchFn = void (*)(const AbcdeArgs&, hipStream_t);
using AbcdeLaunch = Launcher<AbcdeArgs>;  // host function or run-time compiled kernel
inline dim3 abcde_geom(const AbcdeArgs& a) { return dim3((unsigned)((a.N + kAbcdeBlock - 1) / kAbcdeBlock)); }
#endif

}  // namespace kabc
