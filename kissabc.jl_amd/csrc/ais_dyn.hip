// ais_dyn.hip -- the run-time-dimension AIS kernels (ais_dyn_kernels.hpp) for the built-in
// DeviceCosts: one instantiation, the cost is dispatched on its id inside the kernel.
#include "ais_dyn_kernels.hpp"

namespace kabc {
AisDynLaunchFn find_ais_dyn_kernel() { return &launch_ais_dyn<0>; }
}  // namespace kabc
