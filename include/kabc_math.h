/*
 * kabc_math.h -- the ARITHMETIC CONTRACT of the KissABC MI355X walker-update path.
 *
 * Every backend (the gfx950 HIP kernels, the C-ABI host code, and the CPU oracle
 * under oracle/) computes its transcendental functions through these inline
 * definitions and nothing else.  They use only IEEE-754 binary64 +, -, *, /,
 * sqrt, fma and integer bit manipulation, each of which is correctly rounded on
 * x86-64 and on gfx950, so a walker trajectory is BIT-IDENTICAL on the host and
 * on the device as long as both are compiled with -ffp-contract=off (the build
 * does that; every fused multiply-add below is an explicit kabc_fma()).
 *
 * Why it exists: the reference (KissABC.jl) calls Julia's libm-free openlibm
 * ports through Distributions.jl / Random (src/types.jl:137,156 randexp/abs2,
 * src/transition.jl:3,58 exp/log, src/smc.jl:165-166 randn/log).  Those are not
 * vendored in the reference tree, so bit parity is defined between OUR backends;
 * tests/test_math.py pins each function here against glibc libm / mpmath to
 * <= 2 ulp and the priors against scipy golden vectors.
 *
 * Constants are derived by tools/gen_math_consts.py (mpmath), not copied.
 * Valid C99 and HIP C++.
 */
#ifndef KABC_MATH_H
#define KABC_MATH_H

#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define KABC_HD __host__ __device__ inline
#else
#define KABC_HD static inline
#endif

#define KABC_INF (__builtin_inf())
#define KABC_NAN (__builtin_nan(""))

KABC_HD uint64_t kabc_bits(double x) {
    uint64_t u;
    __builtin_memcpy(&u, &x, 8);
    return u;
}
KABC_HD double kabc_from_bits(uint64_t u) {
    double x;
    __builtin_memcpy(&x, &u, 8);
    return x;
}
KABC_HD double kabc_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
KABC_HD double kabc_sqrt(double x) { return __builtin_sqrt(x); }
KABC_HD double kabc_fabs(double x) { return __builtin_fabs(x); }
/* round to nearest, ties to even == Julia round(Int, x) (src/types.jl:114) */
KABC_HD double kabc_rint(double x) { return __builtin_rint(x); }
KABC_HD double kabc_floor(double x) { return __builtin_floor(x); }
KABC_HD int kabc_isfinite(double x) {
    return ((kabc_bits(x) >> 52) & 0x7ff) != 0x7ff;
}
KABC_HD int kabc_isnan(double x) { return x != x; }

/* x / c with rc = RN(1/c) precomputed (Markstein 1990: q = x*rc, r = x - q*c
 * exactly by fma, q' = q + r*rc).  Equals the correctly rounded IEEE quotient
 * x / c whenever no intermediate over/underflows (tests/test_math_contract.py
 * checks it against `/` on 10^7 random pairs).  A hardware f64 division costs
 * ~14 VALU instructions incl. a quarter-rate reciprocal on gfx950; this costs 3.
 * The contract uses it wherever the reference divides by a loop-invariant
 * (`/ 300`, `/ 3` in src/transition.jl:13,35, `cost / scale` in src/types.jl:137,
 * `(x - mu) / sigma` in the Normal log-density). */
KABC_HD double kabc_div_rc(double x, double c, double rc) {
    const double q = x * rc;
    const double r = kabc_fma(-q, c, x);
    return kabc_fma(r, rc, q);
}

/* fma(a, b, C) with a compile-time constant C.  On gfx950 hipcc materialises every
 * 64-bit literal of a Horner step into a VGPR pair (two v_mov_b32) because it
 * selects the accumulating v_fmac form: 3 VALU per coefficient.  Pinning the
 * constant to an SGPR pair (two SALU s_mov_b32, issued beside the vector pipe)
 * makes it 1 VALU.  Same correctly rounded fma either way. */
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ double kabc_fma_c(double a, double b, double c) {
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c));
    return r;
}
#else
KABC_HD double kabc_fma_c(double a, double b, double c) { return kabc_fma(a, b, c); }
#endif

#define KABC_LN2_HI 0x1.62e42fee00000p-1
#define KABC_LN2_LO 0x1.a39ef35793c76p-33
#define KABC_INV_LN2 0x1.71547652b82fep+0
#define KABC_PIO2 0x1.921fb54442d18p+0
#define KABC_HALF_LOG_2PI 0x1.d67f1c864beb5p-1
#define KABC_LOG_2PI 0x1.d67f1c864beb5p+0
#define KABC_SQRT3 0x1.bb67ae8584caap+0
#define KABC_INV_SQRT3 0x1.279a74590331cp-1

/* R(z)/z for log, Chebyshev fit on z in [0, ((sqrt2-1)/(sqrt2+1))^2] */
KABC_HD double kabc__log_poly(double z) {
    double r = 0x1.0c04595972ab8p-3;
    r = kabc_fma_c(r, z, 0x1.0fbe83c2cbcf3p-3);
    r = kabc_fma_c(r, z, 0x1.3b1c360763b8ap-3);
    r = kabc_fma_c(r, z, 0x1.745cf901605f7p-3);
    r = kabc_fma_c(r, z, 0x1.c71c720160b47p-3);
    r = kabc_fma_c(r, z, 0x1.2492492476c87p-2);
    r = kabc_fma_c(r, z, 0x1.9999999999a38p-2);
    r = kabc_fma_c(r, z, 0x1.5555555555555p-1);
    return r * z;
}

/* natural log; <= 1 ulp; log(+-0) = -inf, log(x<0) = nan */
KABC_HD double kabc_log(double x) {
    uint64_t ix = kabc_bits(x);
    int k = 0;
    if (ix < 0x0010000000000000ULL || (ix >> 63)) {
        if ((ix << 1) == 0) return -KABC_INF;
        if (ix >> 63) return KABC_NAN;
        x *= 0x1p54; /* subnormal */
        k -= 54;
        ix = kabc_bits(x);
    } else if (ix >= 0x7ff0000000000000ULL) {
        return x + x; /* inf, nan */
    }
    /* x = 2^k * m, m in [sqrt(2)/2, sqrt(2)) */
    uint32_t hx = (uint32_t)(ix >> 32);
    hx += 0x3ff00000u - 0x3fe6a09eu;
    k += (int)(hx >> 20) - 0x3ff;
    hx = (hx & 0x000fffffu) + 0x3fe6a09eu;
    x = kabc_from_bits(((uint64_t)hx << 32) | (ix & 0xffffffffULL));
    double f = x - 1.0;
    double hfsq = 0.5 * f * f;
    double s = f / (2.0 + f);
    double z = s * s;
    double R = kabc__log_poly(z);
    double dk = (double)k;
    return s * (hfsq + R) + dk * KABC_LN2_LO - hfsq + f + dk * KABC_LN2_HI;
}

/* kabc_log restricted to positive NORMAL finite x (every argument the walker
 * update produces: u01 variates, the stretch factor Z, Gamma/Poisson internals);
 * bit-identical to kabc_log there, without the zero/negative/subnormal/inf tests. */
KABC_HD double kabc_log_pn(double x) {
    const uint64_t ix = kabc_bits(x);
    uint32_t hx = (uint32_t)(ix >> 32);
    hx += 0x3ff00000u - 0x3fe6a09eu;
    const int k = (int)(hx >> 20) - 0x3ff;
    hx = (hx & 0x000fffffu) + 0x3fe6a09eu;
    x = kabc_from_bits(((uint64_t)hx << 32) | (ix & 0xffffffffULL));
    const double f = x - 1.0;
    const double hfsq = 0.5 * f * f;
    const double s = f / (2.0 + f);
    const double z = s * s;
    const double R = kabc__log_poly(z);
    const double dk = (double)k;
    return s * (hfsq + R) + dk * KABC_LN2_LO - hfsq + f + dk * KABC_LN2_HI;
}

/* log(1+x); <= 2 ulp */
KABC_HD double kabc_log1p(double x) {
    if (x <= -1.0) return x == -1.0 ? -KABC_INF : KABC_NAN;
    if (!kabc_isfinite(x)) return x + x;
    double ax = kabc_fabs(x);
    if (ax < 0x1p-54) return x;
    /* u = 1+x rounded; c = correction term ((1+x) - u) / u */
    double u = 1.0 + x;
    double c = (ax >= 1.0) ? (1.0 - (u - x)) : (x - (u - 1.0));
    uint64_t iu = kabc_bits(u);
    uint32_t hu = (uint32_t)(iu >> 32);
    hu += 0x3ff00000u - 0x3fe6a09eu;
    int k = (int)(hu >> 20) - 0x3ff;
    c = (k < 54) ? c / u : 0.0;
    hu = (hu & 0x000fffffu) + 0x3fe6a09eu;
    u = kabc_from_bits(((uint64_t)hu << 32) | (iu & 0xffffffffULL));
    double f = u - 1.0;
    double hfsq = 0.5 * f * f;
    double s = f / (2.0 + f);
    double z = s * s;
    double R = kabc__log_poly(z);
    double dk = (double)k;
    return s * (hfsq + R) + (dk * KABC_LN2_LO + c) - hfsq + f + dk * KABC_LN2_HI;
}

/* 2^k for k in [-1022, 1023] */
KABC_HD double kabc__pow2i(int k) { return kabc_from_bits((uint64_t)(k + 1023) << 52); }

/* exp(x); <= 1 ulp */
KABC_HD double kabc_exp(double x) {
    if (kabc_isnan(x)) return x;
    if (x > 709.782712893384) return KABC_INF;
    if (x < -745.1332191019412) return 0.0;
    double kf = kabc_rint(x * KABC_INV_LN2);
    int k = (int)kf;
    double hi = kabc_fma(-kf, KABC_LN2_HI, x);
    double r = kabc_fma(-kf, KABC_LN2_LO, hi);
    /* Taylor sum_{j<=13} r^j / j!, |r| <= ln2/2 */
    double p = 0x1.6124613a86d09p-33;
    p = kabc_fma_c(p, r, 0x1.1eed8eff8d898p-29);
    p = kabc_fma_c(p, r, 0x1.ae64567f544e4p-26);
    p = kabc_fma_c(p, r, 0x1.27e4fb7789f5cp-22);
    p = kabc_fma_c(p, r, 0x1.71de3a556c734p-19);
    p = kabc_fma_c(p, r, 0x1.a01a01a01a01ap-16);
    p = kabc_fma_c(p, r, 0x1.a01a01a01a01ap-13);
    p = kabc_fma_c(p, r, 0x1.6c16c16c16c17p-10);
    p = kabc_fma_c(p, r, 0x1.1111111111111p-7);
    p = kabc_fma_c(p, r, 0x1.5555555555555p-5);
    p = kabc_fma_c(p, r, 0x1.5555555555555p-3);
    p = kabc_fma_c(p, r, 0.5);
    p = kabc_fma_c(p, r, 1.0);
    p = kabc_fma_c(p, r, 1.0);
    if (k > 1022) return p * 0x1p1022 * kabc__pow2i(k - 1022);
    if (k < -1021) return p * 0x1p-1021 * kabc__pow2i(k + 1021);
    return p * kabc__pow2i(k);
}

/* sin(2 pi u), cos(2 pi u) for u in [0,1]; abs error <= 2^-52 */
KABC_HD void kabc_sincos2pi(double u, double* sn, double* cs) {
    double t = u * 4.0;
    double j = kabc_rint(t);
    double x = (t - j) * KABC_PIO2; /* |x| <= pi/4 */
    double w = x * x;
    double sp = 0x1.952c77030ad4ap-49;
    sp = kabc_fma_c(sp, w, -0x1.ae7f3e733b81fp-41);
    sp = kabc_fma_c(sp, w, 0x1.6124613a86d09p-33);
    sp = kabc_fma_c(sp, w, -0x1.ae64567f544e4p-26);
    sp = kabc_fma_c(sp, w, 0x1.71de3a556c734p-19);
    sp = kabc_fma_c(sp, w, -0x1.a01a01a01a01ap-13);
    sp = kabc_fma_c(sp, w, 0x1.1111111111111p-7);
    sp = kabc_fma_c(sp, w, -0x1.5555555555555p-3);
    double s = kabc_fma(sp * w, x, x);
    double cp = 0x1.ae7f3e733b81fp-45;
    cp = kabc_fma_c(cp, w, -0x1.93974a8c07c9dp-37);
    cp = kabc_fma_c(cp, w, 0x1.1eed8eff8d898p-29);
    cp = kabc_fma_c(cp, w, -0x1.27e4fb7789f5cp-22);
    cp = kabc_fma_c(cp, w, 0x1.a01a01a01a01ap-16);
    cp = kabc_fma_c(cp, w, -0x1.6c16c16c16c17p-10);
    cp = kabc_fma_c(cp, w, 0x1.5555555555555p-5);
    cp = kabc_fma_c(cp, w, -0.5);
    double c = kabc_fma(cp, w, 1.0);
    int q = ((int)j) & 3;
    double ss = (q & 1) ? c : s;
    double cc = (q & 1) ? s : c;
    if (q == 1 || q == 2) cc = -cc;
    if (q >= 2) ss = -ss;
    *sn = ss;
    *cs = cc;
}

/* log Gamma(x) for x > 0 (Stirling series after shifting x up to >= 16);
 * abs error <= ~4e-15 * max(1, |lgamma|).  x <= 0 -> +inf (poles; never used
 * with negative non-integers on this path). */
KABC_HD double kabc_lgamma(double x) {
    if (kabc_isnan(x)) return x;
    if (x <= 0.0) return KABC_INF;
    if (!kabc_isfinite(x)) return x;
    double p = 1.0;
    double y = x;
    while (y < 16.0) {
        p *= y;
        y += 1.0;
    }
    double iy = 1.0 / y;
    double w = iy * iy;
    double st = -0x1.e4286cb0f5398p-6;
    st = kabc_fma_c(st, w, 0x1.a41a41a41a41ap-8);
    st = kabc_fma_c(st, w, -0x1.f6ab0d9993c7dp-10);
    st = kabc_fma_c(st, w, 0x1.b951e2b18ff23p-11);
    st = kabc_fma_c(st, w, -0x1.3813813813814p-11);
    st = kabc_fma_c(st, w, 0x1.a01a01a01a01ap-11);
    st = kabc_fma_c(st, w, -0x1.6c16c16c16c17p-9);
    st = kabc_fma_c(st, w, 0x1.5555555555555p-4);
    double ly = kabc_log(y);
    double r = (y - 0.5) * ly - y + KABC_HALF_LOG_2PI + st * iy;
    return (p == 1.0) ? r : r - kabc_log(p);
}

/* ---- uniform variates from 64 random bits ------------------------------- */
/* strictly inside (0,1): (k + 1/2) * 2^-52, k = top 52 bits; exact in binary64 */
KABC_HD double kabc_u01(uint64_t r) {
    /* 1 + k 2^-52 in [1, 2) straight from its bit pattern, then two exact additions:
     * (1 + k 2^-52) - 1 = k 2^-52 and k 2^-52 + 2^-53 = (2k + 1) 2^-53 with 2k + 1 < 2^53.
     * Same value as ((double)k + 0.5) * 2^-52 without the 64-bit int -> double conversion
     * (3 instructions instead of 7 on gfx950). */
    const double one_plus = kabc_from_bits(0x3ff0000000000000ULL | (r >> 12));
    return (one_plus - 1.0) + 0x1p-53;
}
/* floor(r * n / 2^64): uniform index in [0, n), bias < n / 2^64 */
KABC_HD uint64_t kabc_index(uint64_t r, uint64_t n) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(r, n);
#else
    return (uint64_t)(((unsigned __int128)r * (unsigned __int128)n) >> 64);
#endif
}
/* kabc_index for n < 2^32 (walker / particle counts): the same floor(r n / 2^64) from
 * two 32x32->64 multiply-adds.  hi32(r) n + ((lo32(r) n) >> 32) < 2^64, so nothing
 * carries out. */
KABC_HD uint32_t kabc_index32(uint64_t r, uint32_t n) {
    const uint64_t t = ((r & 0xffffffffULL) * (uint64_t)n) >> 32;
    return (uint32_t)(((r >> 32) * (uint64_t)n + t) >> 32);
}

/* sqrt for positive NORMAL finite x well inside the exponent range (2^-700 .. 2^700:
 * -2 log u of Box-Muller, squared distances): bit-identical to kabc_sqrt there.  The
 * device code is the Newton sequence hipcc itself emits for a correctly rounded f64
 * sqrt (v_rsq_f64 seed, two coupled Goldschmidt steps, two residual corrections)
 * without its input scaling and zero/inf selects -- 9 instructions fewer. */
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ double kabc_sqrt_pn(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    double h = 0.5 * y;
    const double r = kabc_fma(-h, g, 0.5);
    g = kabc_fma(g, r, g);
    h = kabc_fma(h, r, h);
    const double d0 = kabc_fma(-g, g, x);
    g = kabc_fma(d0, h, g);
    const double d1 = kabc_fma(-g, g, x);
    return kabc_fma(d1, h, g);
}
#else
KABC_HD double kabc_sqrt_pn(double x) { return __builtin_sqrt(x); }
#endif

/* Box-Muller: two independent N(0,1) from two 64-bit words */
KABC_HD void kabc_normal_pair(uint64_t r0, uint64_t r1, double* z0, double* z1) {
    double u1 = kabc_u01(r0);
    double u2 = kabc_u01(r1);
    double rad = kabc_sqrt_pn(-2.0 * kabc_log_pn(u1));
    double s, c;
    kabc_sincos2pi(u2, &s, &c);
    *z0 = rad * c;
    *z1 = rad * s;
}

#endif /* KABC_MATH_H */
