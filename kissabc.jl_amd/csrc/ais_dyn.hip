// ais_dyn.hip -- the run-time-dimension AIS and smc kernels (ais_dyn_kernels.hpp, smc_dyn_kernels.hpp) for the
// built-in DeviceCosts.  The four costs that take any number of parameters get their own instantiation (a
// kernel that carries every built-in cost allocates the registers of the hungriest: 292 against 140-146 for
// the AIS kernel, 175 against 91-97 for smc's -- one wavefront per SIMD against three to five); any other id is
// dispatched inside the kernel.
#include "ais_dyn_kernels.hpp"
#include "smc_dyn_kernels.hpp"

namespace kabc {
AisDynLaunchFn find_ais_dyn_kernel(int cost_id) {
    switch (cost_id) {
        case KABC_COST_GAUSS_DIST: return &launch_ais_dyn<KABC_COST_GAUSS_DIST>;
        case KABC_COST_ROSENBROCK: return &launch_ais_dyn<KABC_COST_ROSENBROCK>;
        case KABC_COST_HIER_GAUSS_SIM: return &launch_ais_dyn<KABC_COST_HIER_GAUSS_SIM>;
        case KABC_COST_NORM_SHELL: return &launch_ais_dyn<KABC_COST_NORM_SHELL>;
        default: return &launch_ais_dyn<0>;
    }
}
SmcDynLaunchFn find_smc_dyn_kernel(int cost_id) {
    switch (cost_id) {
        case KABC_COST_GAUSS_DIST: return &launch_smc_dyn<KABC_COST_GAUSS_DIST>;
        case KABC_COST_ROSENBROCK: return &launch_smc_dyn<KABC_COST_ROSENBROCK>;
        case KABC_COST_HIER_GAUSS_SIM: return &launch_smc_dyn<KABC_COST_HIER_GAUSS_SIM>;
        case KABC_COST_NORM_SHELL: return &launch_smc_dyn<KABC_COST_NORM_SHELL>;
        default: return &launch_smc_dyn<0>;
    }
}
}  // namespace kabc
