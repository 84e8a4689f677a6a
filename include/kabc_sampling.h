/*
 * kabc_sampling.h -- `rand(rng, prior_component)`: how counter-stream blocks
 * become prior draws.  Part of the stream contract (with kabc_philox.h).
 *
 * The reference calls Distributions.jl `rand` per component
 * (src/priors.jl:42-43 via src/types.jl:34-35); Distributions.jl is not in
 * the reference tree and its samplers consume a serial RNG, so only the
 * DISTRIBUTION of each draw is reproducible, not the stream.  The samplers here
 * are textbook algorithms written against a bounded window of counter slots
 * (so a draw is a pure function of (seed, walker, attempt, dimension)):
 *   Gamma   : Marsaglia & Tsang 2000, "A simple method for generating gamma variables"
 *   Poisson : multiplication method for lambda < 10, Hörmann 1993 PTRS otherwise
 *   Beta    : ratio of Gammas;  NegativeBinomial : Gamma-Poisson mixture
 *   truncated Normal : rejection from the parent Normal (bounded tries)
 * tests/test_priors.py checks moments/KS of each against scipy.stats.
 */
#ifndef KABC_SAMPLING_H
#define KABC_SAMPLING_H

#include "kabc.h"
#include "kabc_philox.h"
#include "kabc_sampling_base.h"
#include "kabc_mvnormal.h"

/* rand(rng, p_k) as a Float64 (`op(float, ...)`, src/KissABC.jl:50).  The window
 * covers KABC_SLOTS_PER_DIM slots. */
KABC_HD double kabc_sample_prior(const kabc_prior_t* pr, const kabc_slotwin_t* w) {
    const double p0 = pr->p[0], p1 = pr->p[1];
    switch (pr->kind) {
        case KABC_PRIOR_UNIFORM: {
            double u = kabc_u01(kabc_lo64(kabc_slot(w, 0)));
            return p0 + (p1 - p0) * u;
        }
        case KABC_PRIOR_NORMAL: {
            kabc_u128_t b = kabc_slot(w, 0);
            double z0, z1;
            kabc_normal_pair(kabc_lo64(b), kabc_hi64(b), &z0, &z1);
            return p0 + p1 * z0;
        }
        case KABC_PRIOR_TRUNCNORMAL: {
            const double lo = pr->p[2], hi = pr->p[3];
            for (uint32_t j = 0; j < KABC_SLOTS_PER_DIM; ++j) {
                kabc_u128_t b = kabc_slot(w, j);
                double z0, z1;
                kabc_normal_pair(kabc_lo64(b), kabc_hi64(b), &z0, &z1);
                double x0 = p0 + p1 * z0;
                if (x0 >= lo && x0 <= hi) return x0;
                double x1 = p0 + p1 * z1;
                if (x1 >= lo && x1 <= hi) return x1;
            }
            /* the window [lo,hi] has < 2^-100 mass under the parent: take its nearest end */
            return (kabc_fabs(lo - p0) < kabc_fabs(hi - p0)) ? lo : hi;
        }
        case KABC_PRIOR_BETA: {
            double x = kabc_sample_gamma1(w, 0u, p0);
            double y = kabc_sample_gamma1(w, 64u, p1);
            return x / (x + y);
        }
        case KABC_PRIOR_DISCRETE_UNIFORM: {
            uint64_t n = (uint64_t)(p1 - p0 + 1.0);
            return p0 + (double)kabc_index(kabc_lo64(kabc_slot(w, 0)), n);
        }
        case KABC_PRIOR_NEGBINOMIAL: {
            double lam = kabc_sample_gamma1(w, 0u, p0) * ((1.0 - p1) / p1);
            return kabc_sample_poisson(w, 64u, lam);
        }
        case KABC_PRIOR_EXPONENTIAL: {
            double u = kabc_u01(kabc_lo64(kabc_slot(w, 0)));
            return -p0 * kabc_log(u);
        }
        case KABC_PRIOR_GAMMA: return kabc_sample_gamma1(w, 0u, p0) * p1;
        case KABC_PRIOR_LOGNORMAL: {
            kabc_u128_t b = kabc_slot(w, 0);
            double z0, z1;
            kabc_normal_pair(kabc_lo64(b), kabc_hi64(b), &z0, &z1);
            return kabc_exp(p0 + p1 * z0);
        }
        case KABC_PRIOR_MVNORMAL: {
            /* x_k = mu_k + sum_{j<=k} L[k][j] z_j, z_j = the standard normal dimension j's own
             * window yields (the draw a Normal component would make there).  `pr` is a RESOLVED
             * component: p[1] = k, p[2] = the prepared block, p[3] = D (kabc_mvnormal.h). */
            const int k = (int)p1, D = (int)pr->p[3];
            const double* blk = kabc_mvn_ptr_from_double(pr->p[2]);
            const double* L = kabc_mvn_L(blk, D) + k * D;
            double acc = 0.0;
            for (int j = 0; j <= k; ++j) {
                kabc_slotwin_t wj = *w;
                wj.base = w->base - (uint32_t)(k - j) * KABC_SLOTS_PER_DIM;
                kabc_u128_t b = kabc_slot(&wj, 0);
                double z0, z1;
                kabc_normal_pair(kabc_lo64(b), kabc_hi64(b), &z0, &z1);
                acc = acc + L[j] * z0;
            }
            return blk[k] + acc;
        }
        default:
            /* a user family (kind >= KABC_PRIOR_USER): kabc_user_prior_rand of its snippet, compiled
             * into this translation unit in front of this header (capi_plugin.hip: model units) */
#ifdef KABC_USER_MVPRIOR_RAND
            /* a JOINT user prior (kabc_compile_mvprior_plugin): all D components carry its kind and the
             * draw is one function of the whole vector.  `pr` is a RESOLVED component of a contiguous array
             * (p[3] = D), `w` its window (base = k * KABC_SLOTS_PER_DIM): coordinate k of the draw the
             * snippet makes from the walker's window at base 0 -- the same vector for every k. */
            if (pr->kind >= KABC_PRIOR_USER && KABC_USER_PRIOR_IS_JOINT(pr->kind)) {
                const int k = (int)(w->base / KABC_SLOTS_PER_DIM), D = (int)pr->p[3];
                double tmp[KABC_MAX_DIM_DYN];
                kabc_slotwin_t w0 = *w;
                w0.base = 0u;
                KABC_USER_MVPRIOR_RAND(pr->kind, tmp, D, (pr - k)->p, (int)(sizeof(kabc_prior_t) / sizeof(double)), &w0);
                return tmp[k];
            }
#endif
#ifdef KABC_USER_PRIOR_RAND
            if (pr->kind >= KABC_PRIOR_USER) return KABC_USER_PRIOR_RAND(pr->kind, pr->p, w);
#endif
            return KABC_NAN;
    }
}

/* (built-in families only: whether a user family is discrete is a property of its registration,
 * kabc_compile_prior_plugin, kept by the host side that prepares the components) */
KABC_HD int kabc_prior_is_discrete(int kind) {
    return kind == KABC_PRIOR_DISCRETE_UNIFORM || kind == KABC_PRIOR_NEGBINOMIAL;
}

#endif /* KABC_SAMPLING_H */
