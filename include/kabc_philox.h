/*
 * kabc_philox.h -- counter-based random streams of the walker-update path.
 *
 * Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy
 * as 1, 2, 3", SC'11), restated from the published round function; pinned by
 * the Random123 known-answer vectors in tests/test_math_contract.py.
 *
 * The reference draws from ONE serial Julia RNG (rand/randn/randexp call sites
 * src/transition.jl:3,6,9,13,27-33,38-40,49,54,62; src/types.jl:74,103;
 * src/smc.jl:163-166).  That stream is defined by Julia's stdlib, which is not
 * in the reference tree, and is inherently serial.  Here every draw is a pure
 * function of (seed, walker id, transition counter, slot, domain), so the
 * result does not depend on how walkers are mapped to lanes, workgroups or GPUs.
 *
 *   key     = (seed lo32, seed hi32)
 *   counter = (walker id, transition counter t lo32, slot, domain | t hi24 << 8)
 */
#ifndef KABC_PHILOX_H
#define KABC_PHILOX_H

#include "kabc_math.h"

typedef struct kabc_u128 {
    uint32_t w[4];
} kabc_u128_t;

/* Rounds of the Philox4x32 bijection: part of the stream contract (device kernels and the CPU
 * oracle compile the same value).  10 = the Random123 default (Philox4x32-10); 7 is the smallest
 * count the Random123 paper reports as passing BigCrush ("Crush-resistant"), 30 % fewer
 * instructions per block.  tests/test_math_contract.py pins both against an independent
 * restatement of the round function. */
#ifndef KABC_PHILOX_ROUNDS
#define KABC_PHILOX_ROUNDS 10
#endif

#define KABC_PHILOX_M0 0xD2511F53u
#define KABC_PHILOX_M1 0xCD9E8D57u
#define KABC_PHILOX_W0 0x9E3779B9u
#define KABC_PHILOX_W1 0xBB67AE85u

/* stream domains (counter word 3, low 8 bits) */
#define KABC_DOM_AIS_INIT 1u      /* prior draws of step(init), src/KissABC.jl:50,55 */
#define KABC_DOM_AIS_INIT_COST 2u /* cost RNG during init */
#define KABC_DOM_AIS_MOVE 3u      /* propose + accept draws, src/transition.jl */
#define KABC_DOM_AIS_COST 4u      /* cost RNG inside transition! */
#define KABC_DOM_SMC_INIT 5u      /* src/smc.jl:119 */
#define KABC_DOM_SMC_INIT_COST 6u /* src/smc.jl:120-123 */
#define KABC_DOM_SMC_MOVE 7u      /* src/smc.jl:160-167 */
#define KABC_DOM_SMC_COST 8u      /* src/smc.jl:176 */
#define KABC_DOM_ABCDE_INIT 9u       /* src/smc.jl:349,362 */
#define KABC_DOM_ABCDE_INIT_COST 10u /* src/smc.jl:358,364 */
#define KABC_DOM_ABCDE_MOVE 11u      /* src/smc.jl:392-406 */
#define KABC_DOM_ABCDE_COST 12u      /* src/smc.jl:408 */
#define KABC_DOM_PF_INIT 13u         /* src/smc.jl:280,290 */
#define KABC_DOM_PF_INIT_COST 14u    /* src/smc.jl:287,292 */
#define KABC_DOM_PF_MOVE 15u         /* src/smc.jl:309-319 */
#define KABC_DOM_PF_COST 16u         /* src/smc.jl:321 */

/* a ^ b ^ c: one v_bitop3_b32 on gfx950 instead of two v_xor_b32 */
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ uint32_t kabc_xor3(uint32_t a, uint32_t b, uint32_t c) {
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
}
#else
KABC_HD uint32_t kabc_xor3(uint32_t a, uint32_t b, uint32_t c) { return a ^ b ^ c; }
#endif

KABC_HD kabc_u128_t kabc_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                       uint32_t k0, uint32_t k1) {
#if defined(__clang__)
#pragma unroll
#endif
    for (int r = 0; r < KABC_PHILOX_ROUNDS; ++r) {
        uint64_t p0 = (uint64_t)KABC_PHILOX_M0 * (uint64_t)c0;
        uint64_t p1 = (uint64_t)KABC_PHILOX_M1 * (uint64_t)c2;
        uint32_t n0 = kabc_xor3((uint32_t)(p1 >> 32), c1, k0);
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = kabc_xor3((uint32_t)(p0 >> 32), c3, k1);
        uint32_t n3 = (uint32_t)p0;
        c0 = n0;
        c1 = n1;
        c2 = n2;
        c3 = n3;
        k0 += KABC_PHILOX_W0;
        k1 += KABC_PHILOX_W1;
    }
    kabc_u128_t o;
    o.w[0] = c0;
    o.w[1] = c1;
    o.w[2] = c2;
    o.w[3] = c3;
    return o;
}

/* one 128-bit block of the stream (seed, walker, t, slot, domain) */
KABC_HD kabc_u128_t kabc_stream_block(uint64_t seed, uint32_t walker, uint64_t t, uint32_t slot,
                                      uint32_t domain) {
    return kabc_philox4x32_10(walker, (uint32_t)t, slot,
                              domain | ((uint32_t)(t >> 32) << 8), (uint32_t)seed,
                              (uint32_t)(seed >> 32));
}

KABC_HD uint64_t kabc_lo64(kabc_u128_t b) { return ((uint64_t)b.w[1] << 32) | b.w[0]; }
KABC_HD uint64_t kabc_hi64(kabc_u128_t b) { return ((uint64_t)b.w[3] << 32) | b.w[2]; }

/* RNG service handed to a (stochastic) cost function: sequential slots of one
 * (walker, t, domain) stream.  This is what replaces the cost closure's use of
 * Julia's global RNG (e.g. README.md:45 `randn(1000)`, test/runtests.jl:109). */
/* every hipcc translation unit (library, plugins) gets the prefetch fields below; the plain-C
 * oracle build does not */
#if defined(__HIPCC__) && !defined(KABC_RNG_PREFETCH)
#define KABC_RNG_PREFETCH 1
#endif
typedef struct kabc_cost_rng {
    uint64_t seed;
    uint64_t t;
    uint32_t walker;
    uint32_t domain;
    uint32_t slot;
    /* State-independent part of the cost, prepared ahead of time (kabc_costs.h
     * "prepared costs"): aux[j * aux_stride] is word j; NULL = not prepared, the cost
     * computes it itself from the stream.  Same arithmetic either way. */
    uint32_t aux_stride;
    const double* aux;
    /* copy of kabc_log_tab the normals should look their logs up in (a kernel's LDS copy:
     * a per-lane gather from LDS instead of global memory); NULL = kabc_log_tab */
    const double* logtab;
#ifdef KABC_RNG_PREFETCH
    /* The first pre_n blocks of the stream, already expanded into normal pairs
     * (pre[2 s], pre[2 s + 1] = the pair of block s): a kernel computes them while its
     * loads are in flight -- the draws are counter-based, so when they are computed changes
     * nothing.  Only for costs whose first pre_n blocks ARE normal pairs (kabc_device.hpp
     * cost_pre_blocks).  Compiled into the translation units that define KABC_RNG_PREFETCH. */
    const double* pre;
    uint32_t pre_n;
    /* distance between consecutive words of `pre` (1: a per-thread array; the batch size: one
     * lane's column of an SoA block in LDS).  Whoever sets `pre` sets it. */
    uint32_t pre_stride;
#endif
} kabc_cost_rng_t;

/* the table a cost should hand to the table-driven functions of kabc_math.h (kabc_log_t,
 * kabc_log1p_t, kabc_lgamma_t, kabc_sincos2pi_tab, kabc_normal_pair_tab): the calling kernel's
 * LDS copy when there is one.  kabc_log(x) & co. read the table in global memory -- a dependent
 * L2 round trip per call on the AIS consumer wave; same values either way. */
KABC_HD const double* kabc_cost_tab(const kabc_cost_rng_t* g) {
    return (g && g->logtab) ? g->logtab : kabc_log_tab;
}

KABC_HD kabc_u128_t kabc_cost_rng_next(kabc_cost_rng_t* g) {
    return kabc_stream_block(g->seed, g->walker, g->t, g->slot++, g->domain);
}
/* two N(0,1) per block */
KABC_HD void kabc_cost_rng_normal2(kabc_cost_rng_t* g, double* z0, double* z1) {
#ifdef KABC_RNG_PREFETCH
    if (g->pre && g->slot < g->pre_n) {
        *z0 = g->pre[(2u * g->slot) * g->pre_stride];
        *z1 = g->pre[(2u * g->slot + 1u) * g->pre_stride];
        g->slot++;
        return;
    }
#endif
    kabc_u128_t b = kabc_cost_rng_next(g);
    kabc_normal_pair_tab(kabc_lo64(b), kabc_hi64(b), z0, z1, g->logtab ? g->logtab : kabc_log_tab);
}
/* two U(0,1) per block */
KABC_HD void kabc_cost_rng_uniform2(kabc_cost_rng_t* g, double* u0, double* u1) {
    kabc_u128_t b = kabc_cost_rng_next(g);
    *u0 = kabc_u01(kabc_lo64(b));
    *u1 = kabc_u01(kabc_hi64(b));
}

#endif /* KABC_PHILOX_H */
