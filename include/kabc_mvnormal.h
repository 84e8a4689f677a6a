/*
 * kabc_mvnormal.h -- a full-covariance MvNormal(mu, Sigma) prior.
 *
 * The reference takes any `Distribution` as a prior (src/types.jl:30 push_p broadcasts over
 * it, :34-35 `rand(rng, density.prior)`, :52/:85 `logpdf(density.prior, x)`; src/smc.jl:92
 * `prior::Distribution`); Distributions.jl's MvNormal evaluates
 *     logpdf(x) = -(D log 2pi + log det Sigma)/2 - |L^-1 (x - mu)|^2 / 2,   Sigma = L L'
 *     rand      = mu + L randn(D)
 * Here the prior is D components of kind KABC_PRIOR_MVNORMAL that share one prepared block
 *     [ mu (D) | W = L^-1 (D x D, row-major, lower) | L (D x D, row-major, lower) | log L_kk (D) ]
 * and component k contributes, exactly like a univariate Normal (kabc_device.hpp),
 *     z_k = sum_{j<=k} W[k][j] (x_j - mu_j)   (j ascending, mul then add: no contraction)
 *     l_k = -(z_k^2 + log 2pi)/2 - log L_kk
 * so that the left-to-right sum of the components (src/priors.jl:30-36's order, which the rest
 * of the path uses) is the MvNormal log-density.  A draw is x_k = mu_k + sum_{j<=k} L[k][j] z_j
 * with z_j the standard normal that dimension j's own slot window yields (kabc_sampling.h).
 * Shared by the library (host preparation, device evaluation) and the CPU oracle: same
 * operations in the same order on both sides.
 */
#ifndef KABC_MVNORMAL_H
#define KABC_MVNORMAL_H

#include "kabc_math.h"

KABC_HD int kabc_mvn_block_words(int D) { return 2 * D + 2 * D * D; }
KABC_HD const double* kabc_mvn_W(const double* blk, int D) { return blk + D; }
KABC_HD const double* kabc_mvn_L(const double* blk, int D) { return blk + D + D * D; }
KABC_HD const double* kabc_mvn_logdiag(const double* blk, int D) { return blk + D + 2 * D * D; }

/* the block pointer travels in kabc_prior_t.p[2] of the RESOLVED components (the library's /
 * the oracle's internal copies: the caller only supplies p[0] = handle, p[1] = k) */
KABC_HD double kabc_mvn_ptr_to_double(const double* p) { return kabc_from_bits((uint64_t)(uintptr_t)p); }
KABC_HD const double* kabc_mvn_ptr_from_double(double v) { return (const double*)(uintptr_t)kabc_bits(v); }

/* host: Cholesky (row by row), its inverse by forward substitution, log of the diagonal.
 * 0 = ok, 1 = Sigma is not symmetric, 2 = not positive definite. */
static inline int kabc_mvn_prepare(int D, const double* mu, const double* cov, double* blk) {
    double* W = blk + D;
    double* L = blk + D + D * D;
    double* ld = blk + D + 2 * D * D;
    for (int i = 0; i < D; ++i) {
        blk[i] = mu[i];
        for (int j = 0; j < D; ++j) {
            const double a = cov[i * D + j], b = cov[j * D + i];
            const double m = kabc_fabs(a) > kabc_fabs(b) ? kabc_fabs(a) : kabc_fabs(b);
            if (!(kabc_fabs(a - b) <= 1e-12 * m)) return 1;
            L[i * D + j] = 0.0;
            W[i * D + j] = 0.0;
        }
    }
    for (int i = 0; i < D; ++i)
        for (int j = 0; j <= i; ++j) {
            double s = cov[i * D + j];
            for (int k = 0; k < j; ++k) s = s - L[i * D + k] * L[j * D + k];
            if (i == j) {
                if (!(s > 0.0) || !kabc_isfinite(s)) return 2;
                L[i * D + i] = kabc_sqrt(s);
            } else {
                L[i * D + j] = s / L[j * D + j];
            }
        }
    for (int c = 0; c < D; ++c) {
        W[c * D + c] = 1.0 / L[c * D + c];
        for (int i = c + 1; i < D; ++i) {
            double s = 0.0;
            for (int k = c; k < i; ++k) s = s + L[i * D + k] * W[k * D + c];
            W[i * D + c] = -s / L[i * D + i];
        }
    }
    for (int k = 0; k < D; ++k) ld[k] = kabc_log(L[k * D + k]);
    return 0;
}

/* component k of the log-density; xv = the (pushed) coordinates 0..k */
KABC_HD double kabc_mvn_logpdf_comp(const double* blk, int D, int k, const double* xv) {
    const double* W = blk + D + k * D;
    double z = 0.0;
    for (int j = 0; j <= k; ++j) z = z + W[j] * (xv[j] - blk[j]);
    return -(z * z + KABC_LOG_2PI) / 2.0 - blk[D + 2 * D * D + k];
}

#endif /* KABC_MVNORMAL_H */
