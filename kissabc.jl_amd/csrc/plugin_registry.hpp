// plugin_registry.hpp -- run-time DeviceCost plugins (see capi_plugin.hip)
#pragma once
#include <stdint.h>

namespace kabc {

struct CostPlugin {
    void* dl;
    int32_t id;
    int32_t (*dim_ok)(int32_t D);
    void* (*ais)(int32_t D, int32_t prior_class);   // -> AisLaunchFn
    void* (*smc)(int32_t D, int32_t simple_prior);  // -> SmcLaunchFn
    void* (*ais_init)(int32_t D);                   // -> void (*)(const InitArgs&, hipStream_t)
    void* (*smc_init)(int32_t D);                   // -> void (*)(const SmcInitArgs&, hipStream_t)
    void* (*abcde_init)(int32_t D);                 // -> AbcdeLaunchFn (may be NULL)
    void* (*abcde_gen)(int32_t D);                  // -> AbcdeLaunchFn (may be NULL)
    void* (*pf_attempt)(int32_t D);                 // -> PfLaunchFn (may be NULL)
    void* (*smc_loop)(int32_t D, int32_t simple_prior);  // -> SmcLoopLaunchFn (may be NULL)
    void* (*ais_dyn)(void);                          // -> AisDynLaunchFn (may be NULL)
    void* (*smc_dyn)(void);                          // -> SmcDynLaunchFn (may be NULL)
};

const CostPlugin* find_plugin(int cost_id);
bool cost_dim_ok_rt(int cost_id, int D);

}  // namespace kabc
