/*
 * kabc_costs.h -- the DeviceCost library.
 *
 * In the reference the `cost` argument of ApproxKernelizedPosterior /
 * ApproxPosterior / smc is an arbitrary Julia closure (src/types.jl:42,55;
 * src/smc.jl:94,176).  A gfx950 kernel cannot call a Julia closure, so on this
 * path a cost is a DeviceCost: an id + parameter/data arrays whose formula is
 * defined ONCE here as a host+device inline.  The same definition is evaluated
 * by the HIP kernels and by the CPU oracle, i.e. it plays the role of the user's
 * closure handed to both the reference path and the accelerated path.
 *
 * Each entry cites the reference workload it restates.  `x` is the push_p'ed
 * parameter vector (discrete coordinates already rounded, src/types.jl:27-32).
 * Stochastic simulators draw from the kabc_cost_rng_t stream they are given.
 */
#ifndef KABC_COSTS_H
#define KABC_COSTS_H

#include "kabc_philox.h"

enum {
    KABC_COST_GAUSS_DIST = 1,         /* ||x - c||_2 ; params c[D]                   */
    KABC_COST_ROSENBROCK = 2,         /* sqrt(sum 100(x[k+1]-x[k]^2)^2 + (1-x[k])^2)  (C3, C5)      */
    KABC_COST_HIER_GAUSS_SIM = 3,     /* hierarchical Gaussian simulator, RMS to data  (C4)         */
    KABC_COST_NORMAL_MEANSTD_SIM = 4, /* README.md:43-49: n draws N(mu,sigma), hypot(dmean, 50 dstd) */
    KABC_COST_DIRAC_SQ = 5,           /* test/runtests.jl:79-80: |x^2 + 1 - target|                 */
    KABC_COST_ABS_DIFF = 6,           /* test/runtests.jl:178:   |x - target|                       */
    KABC_COST_NORM_SHELL = 7,         /* test/runtests.jl:186:   | ||x||_2 - target |               */
    KABC_COST_NOISY_QUAD_DU = 8,      /* test/runtests.jl:108-109: |(n^2+du)(n+0.01 randn) - target| */
    KABC_COST_MIXTURE = 9,            /* test/runtests.jl:145-146: |mu + rand((0.1 randn, randn)) - target| */
    KABC_COST_NOISY_BANANA = 10,      /* test/runtests.jl:242,248: noisy Rosenbrock, optional Inf   */
    KABC_COST_WIENER_RMS = 11,        /* test/runtests.jl:116-126: drifted Wiener RMS curve         */
    KABC_COST__COUNT = 12,
    /* ids >= KABC_COST_USER are user DeviceCosts compiled at run time from a C snippet
     * (kabc_register_cost_plugin, include/kabc.h).  The snippet defines
     *   KABC_HD double kabc_user_cost(const double* x, int D, const double* params,
     *                                 const double* data, int64_t ndata, kabc_cost_rng_t* rng);
     * and may use everything in kabc_math.h / kabc_philox.h (logs inside a cost: prefer
     * kabc_log_t(x, kabc_cost_tab(rng)) to kabc_log(x) -- the kernel's LDS copy of the table).
     * Optional, for CommonLogDensity with an arbitrary `sample_init` (src/types.jl:105-113): the
     * snippet also defines
     *   #define KABC_USER_SAMPLE_INIT 1
     *   KABC_HD void kabc_user_sample_init(double* x, int D, const double* params,
     *                                      const double* data, int64_t ndata, kabc_cost_rng_t* rng);
     * which draws one initial walker from the stream it is given (kabc_cost_rng_normal2 /
     * kabc_cost_rng_uniform2); the model's prior components then carry KABC_PRIOR_USER_INIT. */
    KABC_COST_USER = 100
};

/* does the cost consume random numbers? (host-side bookkeeping only) */
KABC_HD int kabc_cost_is_stochastic(int id) {
    return id == KABC_COST_HIER_GAUSS_SIM || id == KABC_COST_NORMAL_MEANSTD_SIM ||
           id == KABC_COST_NOISY_QUAD_DU || id == KABC_COST_MIXTURE ||
           id == KABC_COST_NOISY_BANANA || id == KABC_COST_WIENER_RMS;
}

KABC_HD double kabc_cost_gauss_dist(const double* x, int D, const double* params) {
    double s = 0.0;
    for (int k = 0; k < D; ++k) {
        double d = x[k] - params[k];
        s += d * d;
    }
    return kabc_sqrt_dist(s);
}

KABC_HD double kabc_cost_rosenbrock(const double* x, int D) {
    double s = 0.0;
    for (int k = 0; k + 1 < D; ++k) {
        double a = x[k + 1] - x[k] * x[k];
        double b = 1.0 - x[k];
        s += 100.0 * a * a + b * b;
    }
    /* s is 0, +Inf, NaN or >= 2^-106: a term (1 - x)^2 with x != 1 is at least 2^-106, and if
     * every x[k] before the last is exactly 1 the only other term is 100 (x[D-1] - 1)^2, zero
     * or >= 2^-100.  So the input scaling of the general square root never applies and the
     * scaling-free kabc_sqrt_pn (9 device instructions fewer, bit-identical on normal
     * arguments) serves, with 0 and Inf passed through. */
    const double r = kabc_sqrt_pn(s);
    return (s == 0.0 || s == KABC_INF) ? s : r;
}

/* theta = (m, s, z_1..z_G), G = D-2 groups of 8 observations each:
 * ybar_g = m + s z_g + randn/sqrt(8);  cost = RMS(ybar - data[0..G)). */
KABC_HD double kabc_cost_hier_gauss_sim(const double* x, int D, const double* data,
                                        kabc_cost_rng_t* rng) {
    int G = D - 2;
    double m = x[0], s = x[1];
    double acc = 0.0;
    for (int g = 0; g < G; g += 2) {
        double z0, z1;
        kabc_cost_rng_normal2(rng, &z0, &z1);
        double y0 = m + s * x[2 + g] + z0 * 0x1.6a09e667f3bcdp-2; /* 1/sqrt(8) */
        double d0 = y0 - data[g];
        acc += d0 * d0;
        if (g + 1 < G) {
            double y1 = m + s * x[3 + g] + z1 * 0x1.6a09e667f3bcdp-2;
            double d1 = y1 - data[g + 1];
            acc += d1 * d1;
        }
    }
    return kabc_sqrt(acc / (double)G);
}

/* ---- prepared costs ----------------------------------------------------------
 * A simulator often spends most of its time on draws that do not depend on the
 * parameters (README.md:45: `randn(1000)` scaled and shifted afterwards).  Such a cost
 * splits into  prepare(params, rng) -> aux[W]  (parameter-independent) and the rest;
 * on the device the AIS producer waves run `prepare` for the sub-steps ahead while the
 * consumer wave -- the serial part of the chain -- only finishes the cost.  The draws
 * are counter-based, so preparing them early (or for a proposal the prior then
 * rejects) changes nothing; kabc_cost_eval computes the same aux in place when none was
 * prepared. */
#define KABC_COST_MAX_AUX 2
KABC_HD int kabc_cost_aux_words(int id) { return id == KABC_COST_NORMAL_MEANSTD_SIM ? 2 : 0; }

/* params = (n, mean(tdata), std(tdata)); x = (mu, sigma).  The n draws are
 * mu + sigma z_j, so mean = mu + sigma mean(z), std = |sigma| std(z).
 * prepare: aux = (mean, standard deviation) of the n standard normals of the stream, from
 *          (sum z_j, sum z_j^2) -- kabc_cost_normal_meanstd_moments.
 *
 * SUMMATION ORDER (part of the contract: it fixes the bits).  The n draws are independent, so
 * the sums are defined the way 64 lanes of a wavefront form them together:
 *   - the ceil(n/2) normal pairs (one Philox block each, slot = pair index) are cut into
 *     KABC_SIM_LANES = 64 contiguous slices of c = ceil(pairs / 64) pairs; slice l adds its
 *     draws to its own partial sums in stream order (z0 then z1 of every pair);
 *   - the 64 partials are combined by the pairwise tree
 *         a[l] += a[l + off]   for l = 0, 2 off, 4 off, ...;   off = 1, 2, 4, 8, 16, 32
 *     which is what lane 0 of an xor-butterfly over the wavefront computes (IEEE addition is
 *     commutative, so both partners of a butterfly step hold the same bits).
 * One thread evaluating the whole cost (init kernels, smc, the CPU oracle) runs the slices one
 * after the other through kabc_cost_normal_meanstd_prepare below; the AIS path runs them on 64
 * lanes (csrc/ais_aux_kernels.hpp) -- same operations, same order, same bits.  The README's
 * AIS(10) ensemble (README.md:31-57) is what this is for: five walkers per half-generation used
 * to mean five busy lanes running 500 pairs each, one after the other. */
#define KABC_SIM_LANES 64
KABC_HD int kabc_sim_pairs(int n) { return (n + 1) / 2; }
KABC_HD int kabc_sim_slice(int n) { return (kabc_sim_pairs(n) + KABC_SIM_LANES - 1) / KABC_SIM_LANES; }
/* slice l of the draws: partial (sum z, sum z^2); rng->slot must be the stream's base slot */
KABC_HD void kabc_cost_normal_meanstd_slice(int n, int l, kabc_cost_rng_t* rng, double* psz,
                                            double* pszz) {
    const int pairs = kabc_sim_pairs(n), c = kabc_sim_slice(n);
    const uint32_t base = rng->slot;
    int hi = (l + 1) * c;
    if (hi > pairs) hi = pairs;
    double sz = 0.0, szz = 0.0;
    for (int p = l * c; p < hi; ++p) {
        double z0, z1;
        rng->slot = base + (uint32_t)p;
        kabc_cost_rng_normal2(rng, &z0, &z1);
        sz += z0;
        szz += z0 * z0;
        if (2 * p + 1 < n) {
            sz += z1;
            szz += z1 * z1;
        }
    }
    rng->slot = base;
    *psz = sz;
    *pszz = szz;
}
/* The prepared words: aux = (mean, standard deviation) of the n standard normals, from their
 * sums (sum z, sum z^2).  These steps of README.md:43-49 do not depend on the walker's state
 * either, so they belong to the prepare step (two divisions and a square root less on the
 * chain of dependent transitions); same operations in the same order as when the cost is
 * evaluated in place. */
KABC_HD void kabc_cost_normal_meanstd_moments(int n, double sz, double szz, double* aux) {
    double dn = (double)n;
    double mz = sz / dn;
    double vz = (szz - dn * mz * mz) / (dn - 1.0);
    if (vz < 0.0) vz = 0.0;
    aux[0] = mz;
    aux[1] = kabc_sqrt(vz);
}
KABC_HD void kabc_cost_normal_meanstd_prepare(const double* params, kabc_cost_rng_t* rng,
                                              double* aux) {
    int n = (int)params[0];
    double a[KABC_SIM_LANES], b[KABC_SIM_LANES];
    for (int l = 0; l < KABC_SIM_LANES; ++l) kabc_cost_normal_meanstd_slice(n, l, rng, &a[l], &b[l]);
    for (int off = 1; off < KABC_SIM_LANES; off <<= 1)
        for (int l = 0; l < KABC_SIM_LANES; l += 2 * off) {
            a[l] = a[l] + a[l + off];
            b[l] = b[l] + b[l + off];
        }
    rng->slot += (uint32_t)kabc_sim_pairs(n);
    kabc_cost_normal_meanstd_moments(n, a[0], b[0], aux);
}
KABC_HD double kabc_cost_normal_meanstd_sim(const double* x, const double* params,
                                            kabc_cost_rng_t* rng) {
    double aux[2];
    if (rng->aux) {
        aux[0] = rng->aux[0];
        aux[1] = rng->aux[rng->aux_stride];
    } else {
        kabc_cost_normal_meanstd_prepare(params, rng, aux);
    }
    const double mz = aux[0], svz = aux[1];
    double mean = x[0] + x[1] * mz;
    double sd = kabc_fabs(x[1]) * svz;
    double a = mean - params[1];
    double b = 50.0 * (sd - params[2]);
    return kabc_sqrt(a * a + b * b);
}

/* runtime dispatch of the prepare step (producers; W = kabc_cost_aux_words(id) > 0).
 * A user cost (KABC_COST_USER) takes part by defining, in its snippet,
 *     #define KABC_USER_AUX_WORDS W          (1 .. KABC_COST_MAX_AUX)
 *     KABC_HD void kabc_user_cost_prepare(const double* params, const double* data,
 *                                         int64_t ndata, kabc_cost_rng_t* rng, double* aux);
 * and by reading rng->aux[j * rng->aux_stride] in kabc_user_cost when rng->aux != NULL
 * (calling kabc_user_cost_prepare itself otherwise). */
KABC_HD void kabc_cost_prepare(int id, const double* params, const double* data, int64_t ndata,
                               kabc_cost_rng_t* rng, double* aux) {
    if (id == KABC_COST_NORMAL_MEANSTD_SIM) kabc_cost_normal_meanstd_prepare(params, rng, aux);
#ifdef KABC_USER_AUX_WORDS
    else if (id >= KABC_COST_USER) kabc_user_cost_prepare(params, data, ndata, rng, aux);
#endif
}

KABC_HD double kabc_cost_dirac_sq(const double* x, const double* params) {
    return kabc_fabs(x[0] * x[0] + 1.0 - params[0]);
}

KABC_HD double kabc_cost_abs_diff(const double* x, const double* params) {
    return kabc_fabs(x[0] - params[0]);
}

KABC_HD double kabc_cost_norm_shell(const double* x, int D, const double* params) {
    double s = 0.0;
    for (int k = 0; k < D; ++k) s += x[k] * x[k];
    return kabc_fabs(kabc_sqrt_dist(s) - params[0]);
}

KABC_HD double kabc_cost_noisy_quad_du(const double* x, const double* params,
                                       kabc_cost_rng_t* rng) {
    double z0, z1;
    kabc_cost_rng_normal2(rng, &z0, &z1);
    double n = x[0], du = x[1];
    return kabc_fabs((n * n + du) * (n + z0 * 0.01) - params[0]);
}

KABC_HD double kabc_cost_mixture(const double* x, const double* params, kabc_cost_rng_t* rng) {
    double z0, z1, u0, u1;
    kabc_cost_rng_normal2(rng, &z0, &z1);
    kabc_cost_rng_uniform2(rng, &u0, &u1);
    double e = (u0 < 0.5) ? z0 * 0.1 : z1;
    return kabc_fabs(x[0] + e - params[0]);
}

/* params[0] = probability of returning +Inf (0 for cc, 0.5 for cc2) */
KABC_HD double kabc_cost_noisy_banana(const double* x, const double* params,
                                      kabc_cost_rng_t* rng) {
    double z0, z1, u0, u1;
    kabc_cost_rng_normal2(rng, &z0, &z1);
    kabc_cost_rng_uniform2(rng, &u0, &u1);
    double a = x[0] + z0 * 0.01 - x[1] * x[1];
    double b = x[1] - 1.0 + z1 * 0.01;
    double c = 50.0 * a * a + b * b;
    return (u0 < params[0]) ? KABC_INF : c;
}

/* data = tdata[0..T]; x = (mu, sigma); one multiplicative jitter per call */
KABC_HD double kabc_cost_wiener_rms(const double* x, const double* data, int64_t ndata,
                                    kabc_cost_rng_t* rng) {
    double u0, u1;
    kabc_cost_rng_uniform2(rng, &u0, &u1);
    double jit = 0.95 + 0.1 * u0;
    double acc = 0.0;
    for (int64_t t = 0; t < ndata; ++t) {
        double dt = (double)t;
        double v = kabc_sqrt(x[0] * x[0] * dt * dt + x[1] * x[1] * dt) * jit;
        acc += kabc_fabs(v - data[t]);
    }
    return acc / (double)ndata;
}

/* runtime dispatch (oracle, host checks).  The HIP kernels dispatch at compile
 * time on the id and call the functions above directly. */
KABC_HD double kabc_cost_eval(int id, const double* x, int D, const double* params,
                              const double* data, int64_t ndata, kabc_cost_rng_t* rng) {
    switch (id) {
        case KABC_COST_GAUSS_DIST: return kabc_cost_gauss_dist(x, D, params);
        case KABC_COST_ROSENBROCK: return kabc_cost_rosenbrock(x, D);
        case KABC_COST_HIER_GAUSS_SIM: return kabc_cost_hier_gauss_sim(x, D, data, rng);
        case KABC_COST_NORMAL_MEANSTD_SIM: return kabc_cost_normal_meanstd_sim(x, params, rng);
        case KABC_COST_DIRAC_SQ: return kabc_cost_dirac_sq(x, params);
        case KABC_COST_ABS_DIFF: return kabc_cost_abs_diff(x, params);
        case KABC_COST_NORM_SHELL: return kabc_cost_norm_shell(x, D, params);
        case KABC_COST_NOISY_QUAD_DU: return kabc_cost_noisy_quad_du(x, params, rng);
        case KABC_COST_MIXTURE: return kabc_cost_mixture(x, params, rng);
        case KABC_COST_NOISY_BANANA: return kabc_cost_noisy_banana(x, params, rng);
        case KABC_COST_WIENER_RMS: return kabc_cost_wiener_rms(x, data, ndata, rng);
        default:
#ifdef KABC_USER_COST_DEFINED
            if (id >= KABC_COST_USER) return kabc_user_cost(x, D, params, data, ndata, rng);
#endif
            return KABC_NAN;
    }
}

/* minimum / exact dimension each cost accepts; 0 = any D >= 1 */
KABC_HD int kabc_cost_dim_ok(int id, int D) {
    switch (id) {
        case KABC_COST_GAUSS_DIST: return D >= 1;
        case KABC_COST_ROSENBROCK: return D >= 2;
        case KABC_COST_HIER_GAUSS_SIM: return D >= 3;
        case KABC_COST_NORMAL_MEANSTD_SIM: return D == 2;
        case KABC_COST_DIRAC_SQ: return D == 1;
        case KABC_COST_ABS_DIFF: return D == 1;
        case KABC_COST_NORM_SHELL: return D >= 1;
        case KABC_COST_NOISY_QUAD_DU: return D == 2;
        case KABC_COST_MIXTURE: return D == 1;
        case KABC_COST_NOISY_BANANA: return D == 2;
        case KABC_COST_WIENER_RMS: return D == 2;
        default: return 0; /* user costs: decided by the plugin registry */
    }
}

#endif /* KABC_COSTS_H */
