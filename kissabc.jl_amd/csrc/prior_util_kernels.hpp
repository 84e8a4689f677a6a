// prior_util_kernels.hpp -- the Factored utility kernels of the C ABI (kabc_factored_logpdf /
// push_p / rand, include/kabc.h): logpdf(d::Factored, x) src/priors.jl:30-36, push_p
// src/types.jl:29-32, rand(rng, d::Factored) src/priors.jl:42-43 for n rows at a run-time D.
// A header so that a prior with USER families (kabc_compile_prior_plugin) gets them compiled at
// run time together with the families' snippets (capi_plugin.hip: model units).
#pragma once

#include "kabc_device.hpp"

namespace kabc {

struct PriorUtilArgs {
    const double* x;
    double* out;
    int64_t n;
    int32_t D;
    int32_t mode;  // 0 logpdf, 1 push_p
    uint64_t seed;
    uint64_t attempt;
    uint32_t first_walker;
    uint32_t domain;
    const PriorDev* prior;     // [D] prepared components (device; any D up to KABC_MAX_DIM_DYN)
    const kabc_prior_t* raw;   // [D] raw components (device)
};

__global__ void __launch_bounds__(256) prior_logpdf_kernel(const PriorUtilArgs A) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= A.n) return;
    double s = 0.0;
    for (int k = 0; k < A.D; ++k) {
        const double xv = A.x[i * A.D + k];
        const PriorDev q = A.prior[k];
        const double v = q.discrete ? kabc_rint(xv) : xv;
        if (A.mode == 1) {
            A.out[i * A.D + k] = v;
        } else {
            // logpdf(Factored, x) is evaluated on x as given (src/priors.jl:30-36)
            // (an MvNormal component reads coordinates 0..k of the row: continuous, x as given)
            const double l = q.kind == KABC_PRIOR_MVNORMAL
                                 ? kabc_mvn_logpdf_comp(kabc_mvn_ptr_from_double(q.p[2]), A.D, k, A.x + i * A.D)
                                 : comp_logpdf(q.kind, q, xv);
            s = (k == 0) ? l : s + l;
        }
    }
    // (a joint user prior: the log-density of the row as given)
    if (A.mode == 0) A.out[i] = joint_logpdf_or(s, A.prior[0].kind, A.x + i * A.D, A.D, A.prior, kabc_log_tab);
}

__global__ void __launch_bounds__(256) prior_rand_kernel(const PriorUtilArgs A) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= A.n) return;
    for (int k = 0; k < A.D; ++k) {
        kabc_slotwin_t win = {A.seed, A.attempt, A.first_walker + (uint32_t)i, A.domain,
                              (uint32_t)k * KABC_SLOTS_PER_DIM};
        A.out[i * A.D + k] = kabc_sample_prior(&A.raw[k], &win);  // (a pointer INTO the array: joint priors)
    }
}


#ifndef __HIPCC_RTC__
inline dim3 prior_util_geom(const PriorUtilArgs& a) { return dim3((unsigned)((a.n + 255) / 256)); }
#endif

}  // namespace kabc
