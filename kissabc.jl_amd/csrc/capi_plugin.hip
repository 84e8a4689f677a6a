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
//
// Two ways in:
//   kabc_compile_cost_plugin  (default) the snippet is compiled IN PROCESS by hipRTC for gfx950:
//       no hipcc, no temporary files; kernels are compiled lazily, one family and dimension at
//       a time (AIS: the half-generation kernel of the requested prior class + the init kernel;
//       smc: init + propose/accept + the persistent loop kernel; ...), 1-3 s each, into code
//       objects loaded with hipModuleLoadData and launched with hipModuleLaunchKernel
//       (launcher.hpp).  libhiprtc.so is loaded at first use.
//   kabc_register_cost_plugin  a plugin .so prebuilt by hipcc from the same snippet +
//       csrc/user_plugin.inc (every family and dimension at once, 15-40 s; also the only form
//       that carries the run-time-dimension kernels for length(prior) > KABC_MAX_DIM).
#include <dlfcn.h>
#include <hip/hiprtc.h>

#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "host_common.hpp"
#include "launcher.hpp"
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

// ---- hipRTC, loaded on demand --------------------------------------------------------
struct Hiprtc {
    void* dl = nullptr;
    decltype(&hiprtcCreateProgram) CreateProgram = nullptr;
    decltype(&hiprtcDestroyProgram) DestroyProgram = nullptr;
    decltype(&hiprtcAddNameExpression) AddNameExpression = nullptr;
    decltype(&hiprtcCompileProgram) CompileProgram = nullptr;
    decltype(&hiprtcGetProgramLogSize) GetProgramLogSize = nullptr;
    decltype(&hiprtcGetProgramLog) GetProgramLog = nullptr;
    decltype(&hiprtcGetCodeSize) GetCodeSize = nullptr;
    decltype(&hiprtcGetCode) GetCode = nullptr;
    decltype(&hiprtcGetLoweredName) GetLoweredName = nullptr;
    decltype(&hiprtcGetErrorString) GetErrorString = nullptr;
    std::string why;
};
static Hiprtc g_rtc;
static Hiprtc* hiprtc() {
    Hiprtc& R = g_rtc;
    static std::once_flag once;
    std::call_once(once, [&R] {
        const char* cand[] = {std::getenv("KABC_HIPRTC_LIB"), "libhiprtc.so.7", "libhiprtc.so",
                              "/opt/rocm/lib/libhiprtc.so"};
        for (const char* c : cand) {
            if (!c || !*c) continue;
            R.dl = dlopen(c, RTLD_NOW | RTLD_LOCAL);
            if (R.dl) break;
            R.why += std::string(c) + ": " + dlerror() + "; ";
        }
        if (!R.dl) return;
        bool ok = true;
#define KABC_SYM(field)                                                  \
    R.field = (decltype(R.field))dlsym(R.dl, "hiprtc" #field);           \
    if (!R.field) {                                                      \
        R.why += "missing symbol hiprtc" #field "; ";                    \
        ok = false;                                                      \
    }
        KABC_SYM(CreateProgram)
        KABC_SYM(DestroyProgram)
        KABC_SYM(AddNameExpression)
        KABC_SYM(CompileProgram)
        KABC_SYM(GetProgramLogSize)
        KABC_SYM(GetProgramLog)
        KABC_SYM(GetCodeSize)
        KABC_SYM(GetCode)
        KABC_SYM(GetLoweredName)
        KABC_SYM(GetErrorString)
#undef KABC_SYM
        if (!ok) {
            dlclose(R.dl);
            R.dl = nullptr;
        }
    });
    return R.dl ? &R : nullptr;
}

// where the headers the kernels are written in live: next to the library
// (<root>/kissabc.jl_amd/lib/libkabc_hip.so -> ../csrc and ../../include), or KABC_RTC_INCLUDE
// ("dir1:dir2")
static std::vector<std::string> rtc_include_dirs() {
    std::vector<std::string> dirs;
    if (const char* e = std::getenv("KABC_RTC_INCLUDE")) {
        std::string v(e);
        size_t a = 0;
        while (a <= v.size()) {
            const size_t b = v.find(':', a);
            const std::string d = v.substr(a, b == std::string::npos ? std::string::npos : b - a);
            if (!d.empty()) dirs.push_back(d);
            if (b == std::string::npos) break;
            a = b + 1;
        }
        return dirs;
    }
    Dl_info info;
    if (dladdr((const void*)&find_plugin, &info) && info.dli_fname) {
        std::string lib(info.dli_fname);
        const size_t sl = lib.rfind('/');
        const std::string dir = sl == std::string::npos ? std::string(".") : lib.substr(0, sl);
        dirs.push_back(dir + "/../csrc");
        dirs.push_back(dir + "/../../include");
    }
    return dirs;
}

struct RtcPlugin {
    std::string src;        // the user's snippet
    std::string dim_cond;   // KABC_USER_DIM_OK(D)
    std::vector<int> dims;
    int pk_mask = 7;
    std::mutex mu;
    std::map<std::string, void*> fns;  // "<device>|<name expression>" -> hipFunction_t
    std::vector<hipModule_t> mods;
};

static bool rtc_dim_listed(const RtcPlugin* R, int D) {
    for (int d : R->dims)
        if (d == D) return true;
    return false;
}

// compile `names` (kernel name expressions) from the snippet + `header` into one code object
static kabc_status_t rtc_compile(const RtcPlugin* R, const char* header, bool fma_c_vgpr,
                                 const std::vector<std::string>& names, std::vector<char>* code,
                                 std::vector<std::string>* lowered) {
    Hiprtc* H = hiprtc();
    if (!H) {
        set_error("hipRTC is not available: %s(set KABC_HIPRTC_LIB, or register a plugin .so built "
                  "by hipcc with kabc_register_cost_plugin)", g_rtc.why.c_str());
        return KABC_ERR_DEVICE;
    }
    std::string text = "// generated by kabc_compile_cost_plugin\n#define KABC_USER_DIM_OK(D) (" +
                       R->dim_cond + ")\n#define KABC_USER_PK_MASK " + std::to_string(R->pk_mask) +
                       "\n#include \"kabc_philox.h\"\n" + R->src + "\n#define KABC_USER_COST_DEFINED 1\n";
    if (fma_c_vgpr) text += "#define KABC_FMA_C_VGPR\n";
    if (header) text += std::string("#include \"") + header + "\"\n";
    hiprtcProgram prog = nullptr;
    if (H->CreateProgram(&prog, text.c_str(), "kabc_user_cost.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) {
        set_error("hiprtcCreateProgram failed");
        return KABC_ERR_DEVICE;
    }
    for (const std::string& n : names) (void)H->AddNameExpression(prog, n.c_str());
    // the flags of csrc/Makefile: the arithmetic contract needs -ffp-contract=off
    std::vector<std::string> opt = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                                    "-fno-fast-math"};
    for (const std::string& d : rtc_include_dirs()) opt.push_back("-I" + d);
    std::vector<const char*> optp;
    for (const std::string& o : opt) optp.push_back(o.c_str());
    const hiprtcResult r = H->CompileProgram(prog, (int)optp.size(), optp.data());
    if (r != HIPRTC_SUCCESS) {
        size_t ls = 0;
        (void)H->GetProgramLogSize(prog, &ls);
        std::string log(ls, '\0');
        if (ls) (void)H->GetProgramLog(prog, &log[0]);
        if (log.size() > 3000) log = log.substr(0, 3000) + "...";
        set_error("user cost failed to compile (%s):\n%s", H->GetErrorString(r), log.c_str());
        (void)H->DestroyProgram(&prog);
        return KABC_ERR_INVALID_ARG;
    }
    if (code) {
        size_t cs = 0;
        (void)H->GetCodeSize(prog, &cs);
        code->resize(cs);
        (void)H->GetCode(prog, code->data());
        for (const std::string& n : names) {
            const char* low = nullptr;
            (void)H->GetLoweredName(prog, n.c_str(), &low);
            lowered->push_back(low ? low : "");
        }
    }
    (void)H->DestroyProgram(&prog);
    return KABC_OK;
}

// the kernel `want` of a family; the whole batch `names` is compiled together on a miss
static void* rtc_kernel(RtcPlugin* R, const char* header, bool fma_c_vgpr,
                        const std::vector<std::string>& names, const std::string& want) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) dev = -1;  // (no device: the compilation still runs, the load fails)
    const std::string pre = std::to_string(dev) + "|";
    std::lock_guard<std::mutex> lk(R->mu);
    auto it = R->fns.find(pre + want);
    if (it != R->fns.end()) return it->second;
    std::vector<char> code;
    std::vector<std::string> lowered;
    if (rtc_compile(R, header, fma_c_vgpr, names, &code, &lowered) != KABC_OK) return nullptr;
    hipModule_t mod = nullptr;
    const hipError_t e = dev < 0 ? hipErrorNoDevice : hipModuleLoadData(&mod, code.data());
    if (e != hipSuccess) {
        set_error("the user cost compiled (%zu bytes of gfx950 code for %s), but hipModuleLoadData failed: %s",
                  code.size(), want.c_str(), hipGetErrorString(e));
        return nullptr;
    }
    R->mods.push_back(mod);
    for (size_t i = 0; i < names.size(); ++i) {
        hipFunction_t f = nullptr;
        if (!lowered[i].empty() && hipModuleGetFunction(&f, mod, lowered[i].c_str()) == hipSuccess)
            R->fns[pre + names[i]] = (void*)f;
    }
    it = R->fns.find(pre + want);
    if (it == R->fns.end()) {
        set_error("kernel %s is missing from the compiled user cost", want.c_str());
        return nullptr;
    }
    return it->second;
}

static constexpr int kPriorClassesRt = 4;  // kPriorClasses of ais_kernels.hpp (static_assert there)

PluginKernel plugin_kernel(const CostPlugin* p, int family, int D, int variant) {
    PluginKernel k;
    if (!p) return k;
    if (!p->rtc) {
        switch (family) {
            case kPfAis: k.host = p->ais ? p->ais(D, variant) : nullptr; break;
            case kPfAisInit: k.host = p->ais_init ? p->ais_init(D) : nullptr; break;
            case kPfSmc: k.host = p->smc ? p->smc(D, variant) : nullptr; break;
            case kPfSmcInit: k.host = p->smc_init ? p->smc_init(D) : nullptr; break;
            case kPfSmcLoop: k.host = p->smc_loop ? p->smc_loop(D, variant) : nullptr; break;
            case kPfAbcdeInit: k.host = p->abcde_init ? p->abcde_init(D) : nullptr; break;
            case kPfAbcdeGen: k.host = p->abcde_gen ? p->abcde_gen(D) : nullptr; break;
            case kPfAttempt: k.host = p->pf_attempt ? p->pf_attempt(D) : nullptr; break;
            default: break;
        }
        return k;
    }
    RtcPlugin* R = p->rtc;
    if (D < 1 || D > KABC_MAX_DIM || !rtc_dim_listed(R, D)) return k;
    const std::string d = std::to_string(D), u = std::to_string((int)KABC_COST_USER);
    const std::string ais_init = "kabc::ais_init_kernel<" + d + ">";
    auto smc_names = [&](int simple) {
        const std::string sb = simple ? "true" : "false";
        return std::vector<std::string>{"kabc::smc_init_kernel<" + d + ">",
                                        "kabc::smc_mcmc_kernel<" + d + ", " + u + ", " + sb + ">",
                                        "kabc::smc_loop_kernel<" + d + ", " + u + ", " + sb + ">"};
    };
    switch (family) {
        case kPfAis: {
            const int pc = variant % kPriorClassesRt, pk = variant / kPriorClassesRt + 1;
            if (!((R->pk_mask >> (pk - 1)) & 1) || pc == 3) return k;  // (NORMAL class: the host falls back to SIMPLE)
            const std::string half = "kabc::ais_half_kernel<" + d + ", " + u + ", " + std::to_string(pc) +
                                     ", " + std::to_string(pk) + ">";
            k.mod = rtc_kernel(R, "ais_kernels.hpp", true, {half, ais_init}, half);
            break;
        }
        case kPfAisInit: k.mod = rtc_kernel(R, "ais_kernels.hpp", true, {ais_init}, ais_init); break;
        case kPfSmcInit: {
            // (requested before the pass kernels of the same run: compile for the class it will use)
            const std::vector<std::string> n = smc_names(variant);
            k.mod = rtc_kernel(R, "smc_loop_kernel.hpp", false, n, n[0]);
            break;
        }
        case kPfSmc: {
            const std::vector<std::string> n = smc_names(variant);
            k.mod = rtc_kernel(R, "smc_loop_kernel.hpp", false, n, n[1]);
            break;
        }
        case kPfSmcLoop: {
            const std::vector<std::string> n = smc_names(variant);
            k.mod = rtc_kernel(R, "smc_loop_kernel.hpp", false, n, n[2]);
            break;
        }
        case kPfAbcdeInit:
        case kPfAbcdeGen: {
            const std::vector<std::string> n = {"kabc::abcde_init_kernel<" + d + ">",
                                                "kabc::abcde_gen_kernel<" + d + ">"};
            k.mod = rtc_kernel(R, "abcde_kernels.hpp", false, n, n[family == kPfAbcdeInit ? 0 : 1]);
            break;
        }
        case kPfAttempt: {
            const std::string n = "kabc::pf_attempt_kernel<" + d + ">";
            k.mod = rtc_kernel(R, "pfilter_kernels.hpp", false, {n}, n);
            break;
        }
        default: break;
    }
    return k;
}

bool cost_dim_ok_rt(int cost_id, int D) {
    if (cost_id < KABC_COST_USER) return kabc_cost_dim_ok(cost_id, D) != 0;
    const CostPlugin* p = find_plugin(cost_id);
    if (p && p->rtc) return D >= 1 && D <= KABC_MAX_DIM && rtc_dim_listed(p->rtc, D);
    return p && p->dim_ok(D) != 0;
}

hipError_t rtc_launch(void* fn, dim3 grid, dim3 block, const void* args, hipStream_t s) {
    void* params[] = {const_cast<void*>(args)};
    return hipModuleLaunchKernel((hipFunction_t)fn, grid.x, grid.y, grid.z, block.x, block.y, block.z, 0, s,
                                 params, nullptr);
}

hipError_t rtc_launch_cooperative(void* fn, unsigned G, unsigned block, const void* args, hipStream_t s) {
    int dev = 0, per_cu = 0, cus = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    e = hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (hipFunction_t)fn, (int)block, 0);
    if (e != hipSuccess) return e;
    e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess) return e;
    if ((long long)G > (long long)per_cu * cus) return hipErrorCooperativeLaunchTooLarge;
    void* params[] = {const_cast<void*>(args)};
    return hipModuleLaunchCooperativeKernel((hipFunction_t)fn, G, 1, 1, block, 1, 1, 0, s, params);
}

}  // namespace kabc

using namespace kabc;

extern "C" kabc_status_t kabc_compile_cost_plugin(const char* src, const int32_t* dims, int32_t ndims,
                                                  int32_t posterior_mask, int32_t* out_cost_id) {
    if (!src || !dims || ndims < 1 || !out_cost_id) {
        set_error("kabc_compile_cost_plugin: NULL / empty argument");
        return KABC_ERR_INVALID_ARG;
    }
    RtcPlugin* R = new RtcPlugin();
    R->src = src;
    R->pk_mask = (posterior_mask & 7) ? (posterior_mask & 7) : 7;
    for (int i = 0; i < ndims; ++i) {
        if (dims[i] < 1 || dims[i] > KABC_MAX_DIM) {
            delete R;
            set_error("kabc_compile_cost_plugin: dims must lie in 1..%d (longer parameter vectors run on "
                      "the run-time-dimension kernels of a plugin .so built by hipcc: "
                      "kabc_register_cost_plugin)", KABC_MAX_DIM);
            return KABC_ERR_UNSUPPORTED;
        }
        R->dims.push_back(dims[i]);
        R->dim_cond += (i ? " || (D) == " : "(D) == ") + std::to_string(dims[i]);
    }
    // the snippet alone first: its errors come back now, with the compiler's message, not at the
    // first sample() / smc() call
    if (kabc_status_t st = rtc_compile(R, nullptr, false, {}, nullptr, nullptr)) {
        delete R;
        return st;
    }
    CostPlugin p = {};
    p.rtc = R;
    p.has_sample_init = R->src.find("KABC_USER_SAMPLE_INIT") != std::string::npos ? 1 : 0;
    std::lock_guard<std::mutex> lk(g_mu);
    p.id = KABC_COST_USER + (int32_t)g_plugins.size();
    g_plugins.push_back(p);
    *out_cost_id = p.id;
    return KABC_OK;
}

extern "C" kabc_status_t kabc_plugin_precompile(int32_t cost_id, int32_t family, int32_t D, int32_t variant) {
    const CostPlugin* p = find_plugin(cost_id);
    if (!p) {
        set_error("kabc_plugin_precompile: %d is not a registered user cost", cost_id);
        return KABC_ERR_INVALID_ARG;
    }
    if (family < kPfAis || family > kPfAttempt) {
        set_error("kabc_plugin_precompile: unknown kernel family %d", family);
        return KABC_ERR_INVALID_ARG;
    }
    set_error("%s", "");
    const PluginKernel k = plugin_kernel(p, family, D, variant);
    if (k.host || k.mod) return KABC_OK;
    if (!get_error()[0])
        set_error("the cost plugin has no kernel of family %d for D = %d, variant %d", family, D, variant);
    return KABC_ERR_UNSUPPORTED;
}

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
    CostPlugin p = {};
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
    if (auto hi = (int32_t(*)(void))dlsym(dl, "kabc_plugin_has_sample_init")) p.has_sample_init = hi();
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
