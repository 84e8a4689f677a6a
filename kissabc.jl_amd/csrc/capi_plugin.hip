// capi_plugin.hip -- run-time DeviceCost plugins.
//
// The reference accepts any Julia closure as `cost` (src/types.jl:42,55;
// src/smc.jl:94).  The device path accepts any cost expressible as a C function
//   double kabc_user_cost(const double* x, int D, const double* params,
//                         const double* data, int64_t ndata, kabc_cost_rng_t* rng);
// The host side (kissabc.jl_amd/costs.py: UserCost) wraps the snippet into a
// translation unit ending in #include "user_plugin.inc", compiles it with hipcc for
// gfx950 into a shared library and registers it here; the AIS / SMC kernels in the
// plugin are the same templates instantiated with COST = KABC_COST_USER.
#include <dlfcn.h>

#include <mutex>
#include <vector>

#include "host_common.hpp"
#include "plugin_registry.hpp"

namespace kabc {

static std::mutex g_mu;
static std::vector<CostPlugin> g_plugins;

const CostPlugin* find_plugin(int cost_id) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (const CostPlugin& p : g_plugins)
        if (p.id == cost_id) return &p;
    return nullptr;
}

bool cost_dim_ok_rt(int cost_id, int D) {
    if (cost_id < KABC_COST_USER) return kabc_cost_dim_ok(cost_id, D) != 0;
    const CostPlugin* p = find_plugin(cost_id);
    return p && p->dim_ok(D) != 0;
}

}  // namespace kabc

using namespace kabc;

extern "C" kabc_status_t kabc_register_cost_plugin(const char* path, int32_t* out_cost_id) {
    if (!path || !out_cost_id) {
        set_error("kabc_register_cost_plugin: NULL argument");
        return KABC_ERR_INVALID_ARG;
    }
    void* dl = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!dl) {
        set_error("cannot load cost plugin %s: %s", path, dlerror());
        return KABC_ERR_INVALID_ARG;
    }
    CostPlugin p;
    p.dl = dl;
    auto abi = (int32_t(*)(void))dlsym(dl, "kabc_plugin_abi");
    p.dim_ok = (int32_t(*)(int32_t))dlsym(dl, "kabc_plugin_dim_ok");
    p.ais = (void* (*)(int32_t, int32_t))dlsym(dl, "kabc_plugin_ais");
    p.smc = (void* (*)(int32_t, int32_t))dlsym(dl, "kabc_plugin_smc");
    p.ais_init = (void* (*)(int32_t))dlsym(dl, "kabc_plugin_ais_init");
    p.smc_init = (void* (*)(int32_t))dlsym(dl, "kabc_plugin_smc_init");
    p.abcde_init = (void* (*)(int32_t))dlsym(dl, "kabc_plugin_abcde_init");
    p.abcde_gen = (void* (*)(int32_t))dlsym(dl, "kabc_plugin_abcde_gen");
    p.pf_attempt = (void* (*)(int32_t))dlsym(dl, "kabc_plugin_pf_attempt");
    p.smc_loop = (void* (*)(int32_t, int32_t))dlsym(dl, "kabc_plugin_smc_loop");
    p.ais_dyn = (void* (*)(void))dlsym(dl, "kabc_plugin_ais_dyn");
    p.smc_dyn = (void* (*)(void))dlsym(dl, "kabc_plugin_smc_dyn");
    if (!abi || !p.dim_ok || !p.ais || !p.smc || !p.ais_init || !p.smc_init) {
        dlclose(dl);
        set_error("%s is not a kabc cost plugin (missing entry points)", path);
        return KABC_ERR_INVALID_ARG;
    }
    if (abi() != KABC_VERSION) {
        dlclose(dl);
        set_error("cost plugin %s was built against kabc %d, this library is %d", path, abi(),
                  KABC_VERSION);
        return KABC_ERR_INVALID_ARG;
    }
    std::lock_guard<std::mutex> lk(g_mu);
    p.id = KABC_COST_USER + (int32_t)g_plugins.size();
    g_plugins.push_back(p);
    *out_cost_id = p.id;
    return KABC_OK;
}
