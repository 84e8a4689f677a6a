// ais_inst.hip -- instantiates ais_half_kernel<D, COST> for one DeviceCost id
// (-DKABC_INST_COST=<id>) and every dimension 1..KABC_MAX_DIM the cost accepts.
// One translation unit per cost id so the instantiations compile in parallel.
#include "ais_kernels.hpp"

#ifndef KABC_INST_COST
#error "compile with -DKABC_INST_COST=<cost id>"
#endif

namespace kabc {

template <int D, int COST, int PC>
static void launch_half(const AisArgs& a, hipStream_t s) {
    const unsigned grid = (unsigned)((a.rows_owned + kBatch - 1) / kBatch);
    if (grid == 0) return;
    hipLaunchKernelGGL((ais_half_kernel<D, COST, PC>), dim3(grid), dim3(kAisBlock), 0, s, a);
}

template <int COST, int D, int PC>
static AisLaunchFn pick() {
    if constexpr (cost_dim_ok_c(COST, D)) return &launch_half<D, COST, PC>;
    else return nullptr;
}

template <int COST, int... Ds>
static AisLaunchFn table(int D, int pc, std::integer_sequence<int, Ds...>) {
    AisLaunchFn fb[] = {pick<COST, Ds + 1, kPriorBox>()...};
    AisLaunchFn fs[] = {pick<COST, Ds + 1, kPriorSimple>()...};
    AisLaunchFn fg[] = {pick<COST, Ds + 1, kPriorGeneral>()...};
    if (D < 1 || D > (int)sizeof...(Ds)) return nullptr;
    return pc == kPriorBox ? fb[D - 1] : pc == kPriorSimple ? fs[D - 1] : fg[D - 1];
}

#define KABC_CAT2(a, b) a##b
#define KABC_CAT(a, b) KABC_CAT2(a, b)
AisLaunchFn KABC_CAT(find_ais_kernel_cost_, KABC_INST_COST)(int D, int pc) {
    return table<KABC_INST_COST>(D, pc, std::make_integer_sequence<int, KABC_MAX_DIM>{});
}

}  // namespace kabc
