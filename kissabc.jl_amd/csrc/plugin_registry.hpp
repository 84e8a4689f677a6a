// plugin_registry.hpp -- run-time DeviceCost plugins (see capi_plugin.hip)
#pragma once
#include <stdint.h>

namespace kabc {

struct RtcPlugin;  // a cost compiled in-process by hipRTC (capi_plugin.hip)

struct CostPlugin {
    void* dl;        // plugin .so built by hipcc (kabc_register_cost_plugin), else NULL
    RtcPlugin* rtc;  // hipRTC plugin (kabc_compile_cost_plugin), else NULL
    int32_t id;
    int32_t has_sample_init;  // the snippet defines kabc_user_sample_init (KABC_PRIOR_USER_INIT)
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

// One kernel family of a plugin: a host launch function (.so plugins) or a module kernel
// (hipRTC plugins, compiled at first use -- a family and dimension per compilation, 1-3 s).
// `variant`: AIS pcx (prior class + kPriorClasses * (posterior kind - 1)); smc: simple prior 0/1.
enum PluginFamily {
    kPfAis = 0, kPfAisInit, kPfSmc, kPfSmcInit, kPfSmcLoop, kPfAbcdeInit, kPfAbcdeGen, kPfAttempt
};
struct PluginKernel {
    void* host = nullptr;
    void* mod = nullptr;
};
PluginKernel plugin_kernel(const CostPlugin* p, int family, int D, int variant);

}  // namespace kabc
