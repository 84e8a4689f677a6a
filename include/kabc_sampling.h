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
#include "kabc_mvnormal.h"

#define KABC_SLOTS_PER_DIM 128u

/* a window of slots of the stream (seed, walker, t, domain) */
typedef struct kabc_slotwin {
    uint64_t seed;
    uint64_t t;
    uint32_t walker;
    uint32_t domain;
    uint32_t base;
} kabc_slotwin_t;

KABC_HD kabc_u128_t kabc_slot(const kabc_slotwin_t* w, uint32_t j) {
    return kabc_stream_block(w->seed, w->walker, w->t, w->base + j, w->domain);
}

/* Gamma(shape a, scale 1) using slots [off, off+64) of the window */
KABC_HD double kabc_sample_gamma1(const kabc_slotwin_t* w, uint32_t off, double a) {
    double boost = 1.0;
    if (a < 1.0) {
        double ub = kabc_u01(kabc_lo64(kabc_slot(w, off + 63u)));
        boost = kabc_exp(kabc_log(ub) / a);
        a += 1.0;
    }
    double d = a - 1.0 / 3.0;
    double c = 1.0 / kabc_sqrt(9.0 * d);
    for (uint32_t j = 0; j < 31u; ++j) {
        kabc_u128_t bn = kabc_slot(w, off + 2u * j);
        kabc_u128_t bu = kabc_slot(w, off + 2u * j + 1u);
        double z0, z1;
        kabc_normal_pair(kabc_lo64(bn), kabc_hi64(bn), &z0, &z1);
        double u = kabc_u01(kabc_lo64(bu));
        double v = 1.0 + c * z0;
        if (v <= 0.0) continue;
        v = v * v * v;
        if (kabc_log(u) < 0.5 * z0 * z0 + d - d * v + d * kabc_log(v)) return d * v * boost;
    }
    return d * boost;
}

/* Poisson(lam) using slots [off, off+64) */
KABC_HD double kabc_sample_poisson(const kabc_slotwin_t* w, uint32_t off, double lam) {
    if (!(lam > 0.0)) return 0.0;
    if (lam < 10.0) {
        double L = kabc_exp(-lam);
        double prod = 1.0;
        double k = 0.0;
        for (uint32_t j = 0; j < 64u; ++j) {
            kabc_u128_t b = kabc_slot(w, off + j);
            prod *= kabc_u01(kabc_lo64(b));
            if (!(prod > L)) return k;
            k += 1.0;
            prod *= kabc_u01(kabc_hi64(b));
            if (!(prod > L)) return k;
            k += 1.0;
        }
        return k;
    }
    double slam = kabc_sqrt(lam);
    double loglam = kabc_log(lam);
    double b = 0.931 + 2.53 * slam;
    double a = -0.059 + 0.02483 * b;
    double invalpha = 1.1239 + 1.1328 / (b - 3.4);
    double vr = 0.9277 - 3.6224 / (b - 2.0);
    for (uint32_t j = 0; j < 64u; ++j) {
        kabc_u128_t blk = kabc_slot(w, off + j);
        double U = kabc_u01(kabc_lo64(blk)) - 0.5;
        double V = kabc_u01(kabc_hi64(blk));
        double us = 0.5 - kabc_fabs(U);
        double k = kabc_floor((2.0 * a / us + b) * U + lam + 0.43);
        if (us >= 0.07 && V <= vr) return k;
        if (k < 0.0 || (us < 0.013 && V > us)) continue;
        if (kabc_log(V) + kabc_log(invalpha) - kabc_log(a / (us * us) + b) <=
            -lam + k * loglam - kabc_lgamma(k + 1.0))
            return k;
    }
    return kabc_floor(lam);
}

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
        default: return KABC_NAN;
    }
}

KABC_HD int kabc_prior_is_discrete(int kind) {
    return kind == KABC_PRIOR_DISCRETE_UNIFORM || kind == KABC_PRIOR_NEGBINOMIAL;
}

#endif /* KABC_SAMPLING_H */
