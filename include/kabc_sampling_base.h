/*
 * kabc_sampling_base.h -- the slot window of the counter stream and the building-block samplers
 * (Gamma, Poisson) the prior samplers of kabc_sampling.h are written with.  Split from
 * kabc_sampling.h so that a USER prior family (kabc_compile_prior_plugin, include/kabc.h) can be
 * compiled between the two: its `rand` may use everything here.
 */
#ifndef KABC_SAMPLING_BASE_H
#define KABC_SAMPLING_BASE_H

#include "kabc.h"
#include "kabc_philox.h"

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

#endif /* KABC_SAMPLING_BASE_H */
