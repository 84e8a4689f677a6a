// ais_inst.hip -- instantiates ais_half_kernel<D, COST, prior class, posterior kind> for one DeviceCost id
// (-DKABC_INST_COST=<id>) and every dimension 1..KABC_MAX_DIM the cost accepts.
// One translation unit per cost id so the instantiations compile in parallel.
// Horner constants (kabc_fma_c, include/kabc_math.h) as vector-register operands in these
// kernels: the producers run their own chunk loop with ~80 vector registers to spare, so hipcc
// hoists most of the constants out of it, where the scalar-register form re-materialises each one
// with two s_mov_b32 at every use (scalar registers are what this kernel is short of).  Same
// v_fma_f64, same results; -0.8 % per launch at ntransitions = 100, -3 % at 1.
#define KABC_FMA_C_VGPR
#include "ais_small_kernel.hpp"

#ifndef KABC_INST_COST
#error "compile with -DKABC_INST_COST=<cost id>"
#endif

namespace kabc {

template <int D, int COST, int PC, int PK>
static void launch_half(const AisArgs& a, hipStream_t s, unsigned nchains) {
    const unsigned grid = (unsigned)((a.rows_owned + kBatch - 1) / kBatch);
    if (grid == 0) return;
    hipLaunchKernelGGL((ais_half_kernel<D, COST, PC, PK>), dim3(grid, nchains), dim3(kAisBlock), 0, s, a);
}

// (one translation unit instantiates the dimensions KABC_INST_DLO .. KABC_INST_DHI of some prior
// classes: the Makefile builds three per cost, one of them scheduled differently -- AIS_SCHED there)
#ifndef KABC_INST_DLO
#define KABC_INST_DLO 1
#endif
#ifndef KABC_INST_DHI
#define KABC_INST_DHI KABC_MAX_DIM
#endif
#ifndef KABC_INST_SUFFIX
#define KABC_INST_SUFFIX
#endif
// KABC_INST_PCSEL: 0 every prior class, 1 the NORMAL class only, 2 every class but NORMAL
#ifndef KABC_INST_PCSEL
#define KABC_INST_PCSEL 0
#endif
template <int COST, int D, int PCX>
static AisLaunchFn pick() {
    constexpr bool is_normal = (PCX % kPriorClasses) == kPriorNormal;
    constexpr bool pc_ok = KABC_INST_PCSEL == 0 || (KABC_INST_PCSEL == 1) == is_normal;
    if constexpr (pc_ok && D >= KABC_INST_DLO && D <= KABC_INST_DHI && cost_dim_ok_c(COST, D))
        return &launch_half<D, COST, PCX % kPriorClasses, PCX / kPriorClasses + 1>;
    else return nullptr;
}

template <int COST, int PCX, int... Ds>
static AisLaunchFn row(int D, std::integer_sequence<int, Ds...>) {
    AisLaunchFn f[] = {pick<COST, Ds + 1, PCX>()...};
    return (D >= 1 && D <= (int)sizeof...(Ds)) ? f[D - 1] : nullptr;
}

template <int COST, int... PCXs>
static AisLaunchFn table(int D, int pcx, std::integer_sequence<int, PCXs...>) {
    using Dims = std::make_integer_sequence<int, KABC_MAX_DIM>;
    AisLaunchFn r = nullptr;
    ((pcx == PCXs ? (void)(r = row<COST, PCXs>(D, Dims{})) : (void)0), ...);
    return r;
}

// the one-workgroup kernel of small ensembles (ais_small_kernel.hpp): the classes BOX, NORMAL and
// GENERAL (a SIMPLE prior runs on GENERAL: same bits; the model's own kernel replaces it anyway)
template <int COST, int D, int PCX>
static AisSmallLaunchFn pick_small() {
    constexpr int pc = PCX % kPriorClasses;
    constexpr bool is_normal = pc == kPriorNormal;
    constexpr bool pc_ok = KABC_INST_PCSEL == 0 || (KABC_INST_PCSEL == 1) == is_normal;
    if constexpr (pc_ok && pc != kPriorSimple && D >= KABC_INST_DLO &&
                  D <= KABC_INST_DHI && cost_dim_ok_c(COST, D))
        return &launch_ais_small<D, COST, pc, PCX / kPriorClasses + 1>;
    else return nullptr;
}
template <int COST, int PCX, int... Ds>
static AisSmallLaunchFn row_small(int D, std::integer_sequence<int, Ds...>) {
    AisSmallLaunchFn f[] = {pick_small<COST, Ds + 1, PCX>()...};
    return (D >= 1 && D <= (int)sizeof...(Ds)) ? f[D - 1] : nullptr;
}
template <int COST, int... PCXs>
static AisSmallLaunchFn table_small(int D, int pcx, std::integer_sequence<int, PCXs...>) {
    using Dims = std::make_integer_sequence<int, KABC_MAX_DIM>;
    AisSmallLaunchFn r = nullptr;
    ((pcx == PCXs ? (void)(r = row_small<COST, PCXs>(D, Dims{})) : (void)0), ...);
    return r;
}

#define KABC_CAT2(a, b) a##b
#define KABC_CAT(a, b) KABC_CAT2(a, b)
AisLaunchFn KABC_CAT(KABC_CAT(find_ais_kernel_cost_, KABC_INST_COST), KABC_INST_SUFFIX)(int D, int pcx) {
    return table<KABC_INST_COST>(D, pcx, std::make_integer_sequence<int, kAisVariants>{});
}

AisSmallLaunchFn KABC_CAT(KABC_CAT(find_ais_small_kernel_cost_, KABC_INST_COST), KABC_INST_SUFFIX)(int D, int pcx) {
    return table_small<KABC_INST_COST>(D, pcx, std::make_integer_sequence<int, kAisVariants>{});
}

}  // namespace kabc
