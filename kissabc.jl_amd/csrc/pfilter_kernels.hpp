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

template <int DT>
__global__ void __launch_bounds__(kPfBlock) pf_attempt_kernel(const PfArgs A) {
    const int64_t i = (int64_t)blockIdx.x * kPfBlock + threadIdx.x;
    constexpr int CAP = DimOf<DT>::cap;
    const int D = DimOf<DT>::get(A.D_rt);
    unsigned long long reps = 0, evals = 0, done = 0;
    if (A.ctrl->done) return;  // uniform
    if (i < A.N && A.pending[i]) {
        const uint64_t nok = (uint64_t)A.sel->ess;
        const double eps = A.sel->eps;
        const uint32_t w = (uint32_t)i;
        const double lpi_i = A.lpi[i];
        // the rejection loop of one bad particle (:306-325).  Its proposals are built from
        // survivors only, which nobody writes during the iteration, so the loop needs nothing
        // from the other bad particles: it runs to its end inside the launch.
        const uint32_t a_end = A.loop_attempts ? (1u << 24) : A.attempt + 1u;
        for (uint32_t attempt = A.attempt; attempt < a_end && !done; ++attempt) {
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
            const int64_t b = A.idxok[pb], c = A.idxok[pc], d = A.idxok[pd];
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
                    A.pending[i] = 0;
                    done = 1;
                }
            }
        }
    }
    const unsigned long long sr = wave_sum(reps), se = wave_sum(evals), sd = wave_sum(done);
    if ((threadIdx.x & 63) == 0 && sr) {
        atomicAdd(&A.ctrl->nreps, sr);
        atomicAdd(&A.ctrl->total_reps, sr);
        if (se) atomicAdd(&A.ctrl->cost_evals, se);
        if (sd) atomicAdd(&A.ctrl->remaining, 0ull - sd);
    }
}

#ifndef __HIPCC_RTC__  // host side
using PfLaunchFn = void (*)(const PfArgs&, hipStream_t);
using PfLaunch = Launcher<PfArgs>;
inline dim3 pf_geom(const PfArgs& a) { return dim3((unsigned)((a.N + kPfBlock - 1) / kPfBlock)); }
#endif

}  // namespace kabc
