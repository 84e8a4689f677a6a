// pfilter_kernels.hpp -- gfx950 kernels for pfilter (src/smc.jl:275-340).
// ϵ, the survivor mask and idxok come from smc_select_kernel (mode 1); the kernel
// here is one ATTEMPT of the rejection loop (:306-325) for every particle that is
// still "bad": the host repeats passes until none is left.  Survivors are read-only
// during an iteration and bad particles write only themselves, so passes are
// race-free exactly as the reference's threaded loop is.
#pragma once

#include "kabc_device.hpp"
#include "abcde_kernels.hpp"
#include "smc_kernels.hpp"

namespace kabc {

struct PfCtrl {
    unsigned long long nreps;       // Σ localreps of this iteration (:324)
    unsigned long long remaining;   // bad particles not yet replaced
    unsigned long long cost_evals;  // cumulative
    unsigned long long total_reps;  // cumulative
    // the loop's own state, kept on the device so that several iterations are enqueued per host
    // round trip (pf_iter_end_kernel): iterations finished, ϵ and eff of the last one, the stop
    // decision of :328-333, an error (the select kernel's code, or 9 = a particle left unreplaced)
    long long iters;
    double eps, eff;
    int32_t done, error;
};

struct PfArgs {
    double* theta;
    double* C;
    double* lpi;
    uint8_t* pending;        // 1 = bad and not yet replaced in this iteration
    const int32_t* idxok;    // survivors, ascending (smc_select_kernel's cidx)
    const SmcCtrl* sel;      // eps, ess = number of survivors
    PfCtrl* ctrl;
    const double* cost_params;
    const double* cost_data;
    int64_t cost_ndata;
    int64_t N;
    uint64_t seed;
    uint64_t iteration;
    uint32_t attempt;
    // 1: every pending particle repeats its attempts inside ONE launch until it is replaced
    // (attempt = 0, 1, 2, ... -- the numbering the pass-per-attempt scheme gives it, so the same
    // draws); 0: this launch is attempt `attempt` for all of them (KABC_PF_PASSES=1)
    int32_t loop_attempts;
    int32_t cost_id;
    double proposal_width;
    PriorSet prior;
    int32_t D_rt;               // length(prior) > KABC_MAX_DIM: kernels instantiated with D = 0
    const PriorDev* dprior;     // [D_rt] prepared components on the device
};

constexpr int kPfBlock = 64;

#ifdef KABC_SMC_SINGLE_UNIT  // non-template kernel: defined once, in capi_smc.hip
// pending[i] = !ok[i]; remaining = N - n_ok   (idxbad, :301)
__global__ void __launch_bounds__(256) pf_mark_kernel(uint8_t* pending, const uint8_t* ok,
                                                      PfCtrl* ctrl, const SmcCtrl* sel, int64_t N) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (ctrl->done) return;  // uniform: the loop ended in an earlier iteration of this batch
    if (i < N) pending[i] = ok[i] ? 0 : 1;
    if (i == 0) {
        ctrl->nreps = 0;
        ctrl->remaining = (unsigned long long)(N - sel->ess);
    }
}

// end of an iteration (:326-333): eff, the stop tests -- on the device, so that the host looks
// once per batch of iterations instead of once per iteration
__global__ void pf_iter_end_kernel(PfCtrl* ctrl, SmcCtrl* sel, int64_t N, double eff_tol, double epstol,
                                   int64_t max_iters) {
    if (ctrl->done) return;
    if (sel->error) {
        ctrl->error = sel->error;
        ctrl->done = 1;
        sel->done = 1;
        return;
    }
    if (ctrl->remaining != 0ull) {  // (2^24 proposals did not replace some particle)
        ctrl->error = 9;
        ctrl->done = 1;
        sel->done = 1;
        return;
    }
    const long long iters = ctrl->iters + 1;
    const double eps = sel->eps;
    const double nbad = (double)(N - sel->ess);
    const double eff = nbad / (double)ctrl->nreps;  // :327 (0/0 = NaN when nothing was bad, as in Julia)
    ctrl->iters = iters;
    ctrl->eps = eps;
    ctrl->eff = eff;
    if (eff < eff_tol || eps < epstol || (max_iters >= 0 && iters > max_iters) || !(ctrl->nreps > 0ull)) {
        ctrl->done = 1;
        sel->done = 1;  // (the select kernels still enqueued become no-ops)
    }
}

#endif  // KABC_SMC_SINGLE_UNIT

// the rejection loop of ONE bad particle i (:306-325), attempts [a0, a_end): its proposals are built
// from survivors only (idxok[0 .. nok)), which nobody writes during the iteration, so the loop needs
// nothing from the other bad particles.  Attempt numbering = the draws' counter, whatever kernel runs it.
template <int DT>
__device__ __forceinline__ void pf_reject_loop(const PfArgs& A, int64_t i, uint64_t nok, double eps,
                                               const int32_t* idxok, uint32_t a0, uint32_t a_end,
                                               unsigned long long& reps, unsigned long long& evals,
                                               unsigned long long& done, double* c_new = nullptr) {
    constexpr int CAP = DimOf<DT>::cap;
    const int D = DimOf<DT>::get(A.D_rt);
    const uint32_t w = (uint32_t)i;
    const double lpi_i = A.lpi[i];
    for (uint32_t attempt = a0; attempt < a_end && !done; ++attempt) {
        const uint64_t t = (A.iteration << 24) | (uint64_t)attempt;
        const kabc_u128_t B0 = kabc_stream_block(A.seed, w, t, 0u, KABC_DOM_PF_MOVE);
        const kabc_u128_t B1 = kabc_stream_block(A.seed, w, t, 1u, KABC_DOM_PF_MOVE);
        const kabc_u128_t B2 = kabc_stream_block(A.seed, w, t, 2u, KABC_DOM_PF_MOVE);
        // b=c=d=rand(idxok); while c==b ...; while d==b || d==c ...  (:309-311)
        const int64_t pb = (int64_t)kabc_index(kabc_lo64(B0), nok);
        int64_t pc = (int64_t)kabc_index(kabc_hi64(B0), nok - 1u);
        pc += (pc >= pb);
        const int64_t lo = pb < pc ? pb : pc, hi = pb < pc ? pc : pb;
        int64_t pd = (int64_t)kabc_index(kabc_lo64(B1), nok - 2u);
        pd += (pd >= lo);
        pd += (pd >= hi);
        const int64_t b = idxok[pb], c = idxok[pc], d = idxok[pd];
        double z0, z1;
        kabc_normal_pair(kabc_lo64(B2), kabc_hi64(B2), &z0, &z1);
        const double sc = z0 * A.proposal_width;  // randn(trng)*proposal_width
        double tb[CAP], tc[CAP], td[CAP], p[CAP], xp[CAP];
        load_row_n<DT>(A.theta + b * D, tb, D);
        load_row_n<DT>(A.theta + c * D, tc, D);
        load_row_n<DT>(A.theta + d * D, td, D);
#pragma unroll
        for (int k = 0; k < D; ++k) p[k] = tb[k] + (td[k] - tc[k]) * sc;  // :312
        reps += 1;
        const double ll = logpdf_push_n<DT>(A.prior, A.dprior, D, p, xp);
        const double wp = ll - lpi_i;
        double mn = wp;
        if (!(wp < 0.0)) mn = (wp != wp) ? wp : 0.0;  // min(0.0, ll - logπ[i])
        const double lu = kabc_log_pn(kabc_u01(kabc_hi64(B1)));
        if (!(lu > mn)) {  // :316-318
            kabc_cost_rng_t rng = {A.seed, t, w, KABC_DOM_PF_COST, 0u};
            const double Cp = kabc_cost_eval(A.cost_id, p, D, A.cost_params, A.cost_data,
                                             A.cost_ndata, &rng);  // cost(p.x): NOT push_p'ed
            evals += 1;
            if (!(Cp > eps)) {  // :320-322
                store_row_n<DT>(A.theta + i * D, p, D);
                A.C[i] = Cp;
                A.lpi[i] = ll;
                if (A.pending) A.pending[i] = 0;
                if (c_new) *c_new = Cp;
                done = 1;
            }
        }
    }
}

template <int DT>
__global__ void __launch_bounds__(kPfBlock) pf_attempt_kernel(const PfArgs A) {
    const int64_t i = (int64_t)blockIdx.x * kPfBlock + threadIdx.x;
    unsigned long long reps = 0, evals = 0, done = 0;
    if (A.ctrl->done) return;  // uniform
    if (i < A.N && A.pending[i]) {
        const uint32_t a_end = A.loop_attempts ? (1u << 24) : A.attempt + 1u;
        pf_reject_loop<DT>(A, i, (uint64_t)A.sel->ess, A.sel->eps, A.idxok, A.attempt, a_end, reps, evals, done);
    }
    const unsigned long long sr = wave_sum(reps), se = wave_sum(evals), sd = wave_sum(done);
    if ((threadIdx.x & 63) == 0 && sr) {
        atomicAdd(&A.ctrl->nreps, sr);
        atomicAdd(&A.ctrl->total_reps, sr);
        if (se) atomicAdd(&A.ctrl->cost_evals, se);
        if (sd) atomicAdd(&A.ctrl->remaining, 0ull - sd);
    }
}

// ---- pfilter with at most 256 particles (the reference's default is N = 100) in ONE workgroup, one
// thread per particle, every iteration of the loop (:296-334) inside one launch: at this size the
// four launches of an iteration (select + mark + attempts + end) are 35 us of which the select kernel,
// built to rank 2^15..2^21 keys with 1024 threads, is the largest part.  Here ϵ = quantile(C, q)
// (type 7) comes from rank counting -- every thread counts the keys before its own (N broadcast LDS
// reads) and the two threads holding the bracketing order statistics publish them --, the survivor
// list idxok from wave ballots, the rejection loops are pf_reject_loop as in the other scheme (same
// attempt numbering, same draws), the stop tests are evaluated by every thread from the same words.
// The particles stay in global memory (rows of up to 256 parameters); workgroup barriers order the
// iterations.  Bit-identical to the launch-per-phase scheme and to the oracle.
constexpr int kPfSmallBlock = 256;
struct PfSmallArgs {
    PfArgs pf;  // (pending, idxok, sel unused)
    double q, eff_tol, epstol;
    int64_t max_iters;
    int32_t iters_this_launch;  // > 0: return after that many iterations (verbose runs: one)
};

template <int DT>
__global__ void __launch_bounds__(kPfSmallBlock) pf_small_kernel(const PfSmallArgs S) {
    __shared__ unsigned long long s_key[kPfSmallBlock];
    __shared__ int32_t s_idx[kPfSmallBlock];
    __shared__ unsigned s_cnt[kPfSmallBlock / kWave];
    __shared__ unsigned long long s_red[3][kPfSmallBlock / kWave];
    __shared__ double s_ab[2];
    PfArgs A = S.pf;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wid = tid >> 6;
    const int N = (int)A.N;
    const bool in = tid < N;
    PfCtrl c = *A.ctrl;  // (every thread: uniform values)
    if (c.done) return;
    double Ci = in ? A.C[tid] : 0.0;
    const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (int launched = 0; S.iters_this_launch <= 0 || launched < S.iters_this_launch; ++launched) {
        // ---- ϵ = quantile(C, q) over all particles (:298)
        const unsigned long long nanb = __ballot(in && Ci != Ci);
        if (lane == 0) s_cnt[wid] = (unsigned)__popcll(nanb);
        const unsigned long long ki = in ? key_of(Ci) : ~0ull;
        s_key[tid] = ki;
        __syncthreads();
        unsigned nn = 0;
#pragma unroll
        for (int w = 0; w < kPfSmallBlock / kWave; ++w) nn += s_cnt[w];
        if (nn > 0u) {
            c.error = 1;
            c.done = 1;
            break;
        }
        unsigned rank = 0;  // keys before mine in (key, index) order
#pragma unroll 4
        for (int j = 0; j < N; ++j) {
            const unsigned long long kj = s_key[j];  // (broadcast read)
            rank += (kj < ki || (kj == ki && j < tid)) ? 1u : 0u;
        }
        const long long n = N;
        const double aleph = (double)n * S.q + (1.0 - S.q);
        long long jq = (long long)aleph;
        if (jq < 1) jq = 1;
        if (jq > n - 1) jq = n - 1;
        if (n == 1) jq = 1;
        double gq = aleph - (double)jq;
        gq = gq < 0.0 ? 0.0 : (gq > 1.0 ? 1.0 : gq);
        if (in) {
            if ((long long)rank == jq - 1) s_ab[0] = Ci;
            if ((long long)rank == (n == 1 ? 0 : jq)) s_ab[1] = Ci;
        }
        __syncthreads();
        const double qa = s_ab[0], qb = s_ab[1];
        double eps;
        if (kabc_isfinite(qa) && kabc_isfinite(qb)) eps = qa + gq * (qb - qa);
        else eps = (1.0 - gq) * qa + gq * qb;
        // ---- ok = !(C > ϵ) as the select kernel writes it, idxok ascending (:299-301)
        const bool ok = in && Ci <= eps;
        const unsigned long long okb = __ballot(ok);
        if (lane == 0) s_cnt[wid] = (unsigned)__popcll(okb);
        __syncthreads();
        unsigned nok = 0, before = 0;
#pragma unroll
        for (int w = 0; w < kPfSmallBlock / kWave; ++w) {
            before += (w < wid) ? s_cnt[w] : 0u;
            nok += s_cnt[w];
        }
        if (ok) s_idx[before + (unsigned)__popcll(okb & below)] = tid;
        __syncthreads();
        // ---- every bad particle's rejection loop (:302-325)
        A.iteration = (uint64_t)(c.iters + 1);
        unsigned long long reps = 0, evals = 0, done = 0;
        if (in && !ok) pf_reject_loop<DT>(A, tid, (uint64_t)nok, eps, s_idx, 0u, 1u << 24, reps, evals, done, &Ci);
        const unsigned long long sr = wave_sum(reps), se = wave_sum(evals), sd = wave_sum(done);
        if (lane == 0) {
            s_red[0][wid] = sr;
            s_red[1][wid] = se;
            s_red[2][wid] = sd;
        }
        __syncthreads();  // (and: every row written above is visible to the workgroup's next reads)
        unsigned long long nreps = 0, nev = 0, ndone = 0;
#pragma unroll
        for (int w = 0; w < kPfSmallBlock / kWave; ++w) {
            nreps += s_red[0][w];
            nev += s_red[1][w];
            ndone += s_red[2][w];
        }
        // ---- end of the iteration (:326-333), as pf_iter_end_kernel
        c.cost_evals += nev;
        c.total_reps += nreps;
        c.nreps = nreps;
        const unsigned long long nbad = (unsigned long long)N - nok;
        c.remaining = nbad - ndone;
        if (c.remaining != 0ull) {  // (2^24 proposals did not replace some particle)
            c.error = 9;
            c.done = 1;
            break;
        }
        c.iters += 1;
        c.eps = eps;
        c.eff = (double)nbad / (double)nreps;  // :327 (0/0 = NaN when nothing was bad, as in Julia)
        if (c.eff < S.eff_tol || eps < S.epstol || (S.max_iters >= 0 && c.iters > S.max_iters) || !(nreps > 0ull)) {
            c.done = 1;
            break;
        }
    }
    if (tid == 0) *A.ctrl = c;
}

#ifndef __HIPCC_RTC__  // host side
using PfLaunchFn = void (*)(const PfArgs&, hipStream_t);
using PfLaunch = Launcher<PfArgs>;
inline dim3 pf_geom(const PfArgs& a) { return dim3((unsigned)((a.N + kPfBlock - 1) / kPfBlock)); }
using PfSmallLaunchFn = void (*)(const PfSmallArgs&, hipStream_t);
#endif

}  // namespace kabc
