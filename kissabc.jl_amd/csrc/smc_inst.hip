// smc_inst.hip -- instantiates smc_mcmc_kernel<D, COST> for one DeviceCost id
// (-DKABC_INST_COST=<id>) and every dimension the cost accepts.
#include "smc_loop_kernel.hpp"
#include "smc_small_kernel.hpp"

#ifndef KABC_INST_COST
#error "compile with -DKABC_INST_COST=<cost id>"
#endif

namespace kabc {

template <int D, int COST, bool SIMPLE>
static void launch_mcmc(const SmcMcmcArgs& a, hipStream_t s) {
    const unsigned grid = smc_grid(a);
    if (grid == 0) return;
    hipLaunchKernelGGL((smc_mcmc_kernel<D, COST, SIMPLE>), dim3(grid), dim3(kSmcBlock), 0, s, a);
}

template <int COST, int D, bool SIMPLE>
static SmcLaunchFn pick() {
    if constexpr (cost_dim_ok_c(COST, D)) return &launch_mcmc<D, COST, SIMPLE>;
    else return nullptr;
}

template <int COST, int... Ds>
static SmcLaunchFn table(int D, bool simple, std::integer_sequence<int, Ds...>) {
    SmcLaunchFn fs[] = {pick<COST, Ds + 1, true>()...};
    SmcLaunchFn fg[] = {pick<COST, Ds + 1, false>()...};
    if (D < 1 || D > (int)sizeof...(Ds)) return nullptr;
    return simple ? fs[D - 1] : fg[D - 1];
}

template <int COST, int D, bool SIMPLE>
static SmcLoopLaunchFn pick_loop() {
    if constexpr (cost_dim_ok_c(COST, D)) return &launch_smc_loop<D, COST, SIMPLE>;
    else return nullptr;
}

template <int COST, int... Ds>
static SmcLoopLaunchFn loop_table(int D, bool simple, std::integer_sequence<int, Ds...>) {
    SmcLoopLaunchFn fs[] = {pick_loop<COST, Ds + 1, true>()...};
    SmcLoopLaunchFn fg[] = {pick_loop<COST, Ds + 1, false>()...};
    if (D < 1 || D > (int)sizeof...(Ds)) return nullptr;
    return simple ? fs[D - 1] : fg[D - 1];
}

template <int COST, int D, bool SIMPLE>
static SmcSmallLaunchFn pick_small() {
    if constexpr (cost_dim_ok_c(COST, D)) return &launch_smc_small<D, COST, SIMPLE>;
    else return nullptr;
}

template <int COST, int... Ds>
static SmcSmallLaunchFn small_table(int D, bool simple, std::integer_sequence<int, Ds...>) {
    SmcSmallLaunchFn fs[] = {pick_small<COST, Ds + 1, true>()...};
    SmcSmallLaunchFn fg[] = {pick_small<COST, Ds + 1, false>()...};
    if (D < 1 || D > (int)sizeof...(Ds)) return nullptr;
    return simple ? fs[D - 1] : fg[D - 1];
}

#define KABC_CAT2(a, b) a##b
#define KABC_CAT(a, b) KABC_CAT2(a, b)
SmcLaunchFn KABC_CAT(find_smc_kernel_cost_, KABC_INST_COST)(int D, bool simple) {
    return table<KABC_INST_COST>(D, simple, std::make_integer_sequence<int, KABC_MAX_DIM>{});
}

SmcSmallLaunchFn KABC_CAT(find_smc_small_kernel_cost_, KABC_INST_COST)(int D, bool simple) {
    return small_table<KABC_INST_COST>(D, simple, std::make_integer_sequence<int, KABC_MAX_DIM>{});
}

SmcLoopLaunchFn KABC_CAT(find_smc_loop_kernel_cost_, KABC_INST_COST)(int D, bool simple) {
    return loop_table<KABC_INST_COST>(D, simple, std::make_integer_sequence<int, KABC_MAX_DIM>{});
}

}  // namespace kabc
