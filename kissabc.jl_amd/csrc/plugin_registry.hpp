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
    kPfAis = 0, kPfAisInit, kPfSmc, kPfSmcInit, kPfSmcLoop, kPfAbcdeInit, kPfAbcdeGen, kPfAttempt,
    kPfPriorLogpdf, kPfPriorRand,  // (model units only: the Factored utility kernels)
    kPfSmcSmall,                   // (hipRTC units: the one-workgroup smc driver, smc_small_kernel.hpp)
    // (hipRTC units) the run-time-dimension kernels, length(prior) > KABC_MAX_DIM: variant 0 the
    // half-generation / propose+accept kernel, 1 the init kernel (ais_dyn_kernels.hpp, smc_dyn_kernels.hpp);
    // ABCDE / pfilter beyond KABC_MAX_DIM: their own families, instantiated with D = 0
    kPfAisDyn, kPfSmcDyn,
    // (hipRTC units) the one-workgroup AIS driver of small ensembles (ais_small_kernel.hpp); variant = AIS pcx
    kPfAisSmall
};
struct PluginKernel {
    void* host = nullptr;
    void* mod = nullptr;
};
PluginKernel plugin_kernel(const CostPlugin* p, int family, int D, int variant);

// ---- user prior families + model units (capi_plugin.hip) ------------------------------------
// is `kind` a registered user family (kabc_compile_prior_plugin)?  *discrete: push_p rounds it
bool user_prior_info(int kind, int* discrete);
// a JOINT family (kabc_compile_mvprior_plugin): all D components of a prior carry it
bool user_prior_is_joint(int kind);

// A MODEL UNIT is the recipe of a run-time compiled translation unit for a (prior, cost) pair:
//   * the cost's snippet when the cost is a user cost (hipRTC form),
//   * the snippets of the user prior families among the components,
//   * for a SPECIALISED unit (kabc_compile_model / KABC_SPECIALIZE=1): the prior tuple itself as
//     constexpr data (kabc_device.hpp model_logpdf_push),
// plus the kernels compiled from it so far (per device).  model_unit_for returns, in *out, the
// unit the pair must run on (a user family among the components: there are no prebuilt kernels)
// or may run on (a registered specialisation of exactly these components and cost id), or
// nullptr when the prebuilt kernels serve.  A non-OK status: the pair needs a unit that cannot
// be built (no hipRTC, a user family with a hipcc-built cost plugin, ...), message set.
struct ModelUnit;
}  // namespace kabc
#include "kabc.h"
namespace kabc {
// allow_spec = false: only the unit user families make necessary (the Factored utility kernels)
kabc_status_t model_unit_for(const kabc_prior_t* prior, int D, int cost_id, ModelUnit** out,
                             bool allow_spec = true);
bool unit_is_spec(const ModelUnit* u);
// must the model's kernels come from run-time compiled code (user families among the components:
// there are no prebuilt kernels)?  false: a miss of unit_kernel means "take the prebuilt kernel"
bool unit_required(const ModelUnit* u);
// AIS variant (pcx): a unit instantiates the GENERAL prior class only (user families are not
// "simple"; a specialised unit routes that class to the model's own constants).
// A specialised unit an entry point made on its own (the default, KABC_SPECIALIZE unset) NEVER
// blocks here: its kernels are compiled by a worker process; until they are there the call
// returns the kernel of the unit's fallback (the generic unit of the prior's user families) or
// nothing (= the prebuilt kernel), and *spec_state says KABC_SPEC_PENDING -- ask again at the next
// launch boundary.  Everything returned computes the same bits.
PluginKernel unit_kernel(ModelUnit* u, int family, int D, int variant, int* spec_state = nullptr);

}  // namespace kabc
