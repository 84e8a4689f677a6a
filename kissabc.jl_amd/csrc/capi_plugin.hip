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
#include <dirent.h>
#include <dlfcn.h>
#include <fcntl.h>
#include <hip/hiprtc.h>
#include <spawn.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include <atomic>
#include <chrono>

#include <algorithm>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "host_common.hpp"
#include "launcher.hpp"
#include "plugin_registry.hpp"

extern "C" char** environ;

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

// ---- what gets compiled ---------------------------------------------------------------------
// kernels compiled so far from one translation-unit recipe, per device
// a compilation handed to the worker process (rtc_kernel_try)
struct RtcAsyncJob {
    enum State { kDeferred, kPending, kFailed } state = kPending;  // deferred: the workers are busy, ask again
    std::string cpath;
    uint64_t digest = 0;
    std::chrono::steady_clock::time_point t_spawn, t_poll;
    std::string why;  // (failed)
};
struct RtcCache {
    std::mutex mu;
    std::map<std::string, void*> fns;  // "<device>|<name expression>" -> hipFunction_t
    std::vector<hipModule_t> mods;
    std::map<std::string, RtcAsyncJob> jobs;  // by the wanted name expression (process-wide, not per device)
    // compiler options beyond the common ones for the AIS half-generation kernel of this unit
    // (csrc/Makefile AIS_SCHED: the NORMAL prior class up to seven parameters is scheduled for ILP)
    std::vector<std::string> ais_opt;
};

struct RtcPlugin : RtcCache {
    std::string src;        // the user's snippet
    std::string dim_cond;   // KABC_USER_DIM_OK(D)
    std::vector<int> dims;
    int pk_mask = 7;
};

static bool rtc_dim_listed(const RtcPlugin* R, int D) {
    for (int d : R->dims)
        if (d == D) return true;
    return false;
}

// the cost part of a translation unit: the user's snippet and what the kernels need to know of it
static std::string cost_head(const RtcPlugin* R) {
    return "#define KABC_USER_DIM_OK(D) (" + R->dim_cond + ")\n#define KABC_USER_PK_MASK " +
           std::to_string(R->pk_mask) + "\n#include \"kabc_philox.h\"\n" + R->src +
           "\n#define KABC_USER_COST_DEFINED 1\n";
}

// ---- user prior families (kabc_compile_prior_plugin) ------------------------------------------
struct PriorPlugin {
    int32_t kind;
    int32_t discrete;
    std::string src;
    bool joint = false;  // kabc_compile_mvprior_plugin: one density of the whole vector
};
static std::vector<PriorPlugin*> g_prior_plugins;  // kind = KABC_PRIOR_USER + index (g_mu)

static const PriorPlugin* find_prior_plugin(int kind) {
    std::lock_guard<std::mutex> lk(g_mu);
    const int i = kind - KABC_PRIOR_USER;
    return (i >= 0 && i < (int)g_prior_plugins.size()) ? g_prior_plugins[(size_t)i] : nullptr;
}

bool user_prior_info(int kind, int* discrete) {
    const PriorPlugin* p = find_prior_plugin(kind);
    if (!p) return false;
    if (discrete) *discrete = p->discrete;
    return true;
}
bool user_prior_is_joint(int kind) {
    const PriorPlugin* p = find_prior_plugin(kind);
    return p && p->joint;
}

// the prior part of a translation unit: every listed family's snippet under its own function
// names + the dispatch the built-in switch statements fall through to (include/kabc_sampling.h
// kabc_sample_prior, kabc_device.hpp comp_logpdf_general_body)
static std::string priors_head(const std::vector<int>& kinds) {
    if (kinds.empty()) return std::string();
    std::string t = "#include \"kabc_sampling_base.h\"\n";
    std::string lp = "#define KABC_USER_PRIOR_LOGPDF(kind, x, p, tab) (";
    std::string rd = "#define KABC_USER_PRIOR_RAND(kind, p, w) (";
    // joint families (kabc_compile_mvprior_plugin): one density / one draw of the whole vector
    std::string isj = "#define KABC_USER_PRIOR_IS_JOINT(kind) (";
    std::string jlp = "#define KABC_USER_MVPRIOR_LOGPDF(kind, x, D, p, st, tab) (";
    std::string jrd = "#define KABC_USER_MVPRIOR_RAND(kind, out, D, p, st, w) do { ";
    bool any_joint = false;
    for (int k : kinds) {
        const PriorPlugin* pp = find_prior_plugin(k);
        const std::string sk = std::to_string(k);
        if (pp && pp->joint) {
            any_joint = true;
            t += "#define kabc_user_mvprior_logpdf kabc_user_mvprior_logpdf_" + sk +
                 "\n#define kabc_user_mvprior_rand kabc_user_mvprior_rand_" + sk + "\n" + pp->src +
                 "\n#undef kabc_user_mvprior_logpdf\n#undef kabc_user_mvprior_rand\n";
            isj += "(kind) == " + sk + " || ";
            jlp += "(kind) == " + sk + " ? kabc_user_mvprior_logpdf_" + sk + "(x, D, p, st, tab) : ";
            jrd += "if ((kind) == " + sk + ") kabc_user_mvprior_rand_" + sk + "(out, D, p, st, w); ";
            continue;
        }
        t += "#define kabc_user_prior_logpdf kabc_user_prior_logpdf_" + sk +
             "\n#define kabc_user_prior_rand kabc_user_prior_rand_" + sk + "\n" + (pp ? pp->src : std::string()) +
             "\n#undef kabc_user_prior_logpdf\n#undef kabc_user_prior_rand\n";
        lp += "(kind) == " + sk + " ? kabc_user_prior_logpdf_" + sk + "(x, p, tab) : ";
        rd += "(kind) == " + sk + " ? kabc_user_prior_rand_" + sk + "(p, w) : ";
    }
    t += lp + "KABC_NAN)\n" + rd + "KABC_NAN)\n";
    if (any_joint) t += isj + "0)\n" + jlp + "KABC_NAN)\n" + jrd + "} while (0)\n";
    return t;
}

// a double as a C++17 literal with exactly its bits
static std::string lit(double v) {
    if (v != v) return "__builtin_nan(\"\")";
    if (v == HUGE_VAL) return "__builtin_inf()";
    if (v == -HUGE_VAL) return "(-__builtin_inf())";
    char b[64];
    std::snprintf(b, sizeof b, "%a", v);
    return b;
}

// the model part of a specialised unit: the prepared components as constexpr data
static std::string spec_head(const PriorDev* q, int D) {
    std::string t = "#define KABC_MODEL_SPEC 1\nnamespace kabc_mspec {\nconstexpr int D = " + std::to_string(D) + ";\n";
    std::string kind = "constexpr int KIND[D] = {", disc = "constexpr bool DISC[D] = {", P = "constexpr double P[D][4] = {",
                c0 = "constexpr double C0[D] = {", c1 = "constexpr double C1[D] = {", rb = "constexpr double RB[D] = {";
    for (int k = 0; k < D; ++k) {
        const char* sep = k + 1 < D ? ", " : "};\n";
        kind += std::to_string(q[k].kind) + sep;
        disc += std::string(q[k].discrete ? "true" : "false") + sep;
        P += "{" + lit(q[k].p[0]) + ", " + lit(q[k].p[1]) + ", " + lit(q[k].p[2]) + ", " + lit(q[k].p[3]) + "}" + sep;
        c0 += lit(q[k].c0) + sep;
        c1 += lit(q[k].c1) + sep;
        rb += lit(q[k].rb) + sep;
    }
    return t + kind + disc + P + c0 + c1 + rb + "}\n";
}

// ---- compilation + the on-disk cache of code objects -----------------------------------------
static uint64_t fnv1a(const void* data, size_t n, uint64_t h) {
    const unsigned char* p = (const unsigned char*)data;
    for (size_t i = 0; i < n; ++i) {
        h ^= p[i];
        h *= 0x100000001b3ull;
    }
    return h;
}

// hash of every header a unit is compiled against (the include directories' *.h / *.hpp / *.inc)
// and of the compiler: a cached code object of older kernels must never be loaded
static uint64_t toolchain_fingerprint() {
    static uint64_t fp = 0;
    static std::once_flag once;
    std::call_once(once, [] {
        uint64_t h = 0xcbf29ce484222325ull;
        for (const std::string& d : rtc_include_dirs()) {
            std::vector<std::string> names;
            if (DIR* dir = opendir(d.c_str())) {
                while (const dirent* e = readdir(dir)) {
                    const std::string n = e->d_name;
                    const size_t dot = n.rfind('.');
                    const std::string ext = dot == std::string::npos ? "" : n.substr(dot);
                    if (ext == ".h" || ext == ".hpp" || ext == ".inc") names.push_back(n);
                }
                closedir(dir);
            }
            std::sort(names.begin(), names.end());
            for (const std::string& n : names) {
                h = fnv1a(n.data(), n.size() + 1, h);
                if (FILE* f = std::fopen((d + "/" + n).c_str(), "rb")) {
                    char buf[65536];
                    size_t got;
                    while ((got = std::fread(buf, 1, sizeof buf, f)) > 0) h = fnv1a(buf, got, h);
                    std::fclose(f);
                }
            }
        }
        int maj = 0, min = 0;
        if (auto ver = (hiprtcResult(*)(int*, int*))(g_rtc.dl ? dlsym(g_rtc.dl, "hiprtcVersion") : nullptr))
            (void)ver(&maj, &min);
        h = fnv1a(&maj, sizeof maj, h);
        h = fnv1a(&min, sizeof min, h);
        fp = h;
    });
    return fp;
}

static std::string lib_dir() {
    Dl_info info;
    if (dladdr((const void*)&find_plugin, &info) && info.dli_fname) {
        std::string lib(info.dli_fname);
        const size_t sl = lib.rfind('/');
        return sl == std::string::npos ? std::string(".") : lib.substr(0, sl);
    }
    return std::string();
}

// A cache directory is TRUSTED when nobody but this user can have put a file into it: a real directory
// (not a symbolic link), owned by the effective user, without group / world write permission.  Code
// objects found there are handed to hipModuleLoadData and run on the GPU of THIS process, and the
// names are computable (a hash of the unit's text), so a directory another local user may write to
// -- a pre-created /tmp/kabc_rtc_cache_<uid>, a 0777 directory -- is never used.
static bool cache_dir_trusted(const std::string& d) {
    struct stat st;
    if (lstat(d.c_str(), &st) != 0) return false;
    return S_ISDIR(st.st_mode) && st.st_uid == geteuid() && (st.st_mode & (S_IWGRP | S_IWOTH)) == 0;
}
// creates `d` (0700) when it is missing; true: trusted, writable and searchable
static bool cache_dir_usable(const std::string& d) {
    if (d.empty()) return false;
    (void)mkdir(d.c_str(), 0700);  // (EEXIST: judged by what is there)
    return cache_dir_trusted(d) && access(d.c_str(), W_OK | X_OK) == 0;
}

// KABC_RTC_CACHE_DIR, else <library directory>/rtc_cache, else (a read-only install, or a tree that
// belongs to somebody else) $XDG_CACHE_HOME/kabc_rtc_cache or ~/.cache/kabc_rtc_cache, else
// $TMPDIR/kabc_rtc_cache_<uid> (/tmp without TMPDIR); every candidate must be trusted (above), the first that is serves;
// "" or "0" disables.  No trusted directory: no cache, no worker -- the prebuilt kernels stay.
static std::string rtc_cache_dir() {
    if (const char* e = std::getenv("KABC_RTC_CACHE_DIR")) {
        const std::string v(e);
        if (v.empty() || v == "0") return std::string();
        return cache_dir_usable(v) ? v : std::string();
    }
    static std::string dir;
    static std::once_flag once;
    std::call_once(once, [] {
        std::vector<std::string> cand;
        const std::string ld = lib_dir();
        if (!ld.empty()) cand.push_back(ld + "/rtc_cache");
        if (const char* x = std::getenv("XDG_CACHE_HOME")) {
            if (x[0] == '/') cand.push_back(std::string(x) + "/kabc_rtc_cache");
        } else if (const char* hm = std::getenv("HOME")) {
            if (hm[0] == '/') {
                (void)mkdir((std::string(hm) + "/.cache").c_str(), 0700);
                cand.push_back(std::string(hm) + "/.cache/kabc_rtc_cache");
            }
        }
        const char* td = std::getenv("TMPDIR");
        cand.push_back(std::string(td && td[0] == '/' ? td : "/tmp") + "/kabc_rtc_cache_" + std::to_string((long long)geteuid()));
        for (const std::string& c : cand)
            if (cache_dir_usable(c)) {
                dir = c;
                return;
            }
    });
    return dir;
}

// FNV-1a of a byte range with a second offset basis: the check words inside a cache file
static uint64_t fnv1a(const void* data, size_t n, uint64_t h);
static uint64_t code_checksum(const std::vector<char>& code) { return fnv1a(code.data(), code.size(), 0x84222325cbf29ce4ull); }

// a regular file of ours, opened without following a symbolic link
static FILE* open_cache_file_read(const std::string& path) {
    const int fd = open(path.c_str(), O_RDONLY | O_NOFOLLOW | O_CLOEXEC);
    if (fd < 0) return nullptr;
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_uid != geteuid()) {
        close(fd);
        return nullptr;
    }
    FILE* f = fdopen(fd, "rb");
    if (!f) close(fd);
    return f;
}
// a NEW file of ours (0600): never through a link somebody planted, never on top of an existing file
static FILE* open_cache_file_new(const std::string& path) {
    const int fd = open(path.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW | O_CLOEXEC, 0600);
    if (fd < 0) return nullptr;
    FILE* f = fdopen(fd, "wb");
    if (!f) close(fd);
    return f;
}

// File: "KABCRTC2", key digest (of the unit's text, names and options: what the file's NAME is a hash
// of, with another basis), checksum of the code bytes, the lowered names, the code object.  A file
// whose digest is not the expected one, whose checksum does not match its bytes, that is truncated,
// not a regular file of this user or reached through a link is not loaded (the caller compiles, or
// stays on the prebuilt kernels).
static bool cache_load(const std::string& path, size_t nnames, uint64_t key_digest, std::vector<char>* code,
                       std::vector<std::string>* lowered) {
    FILE* f = open_cache_file_read(path);
    if (!f) return false;
    bool ok = false;
    char magic[8];
    uint32_t n = 0;
    uint64_t dg = 0, ck = 0;
    if (std::fread(magic, 1, 8, f) == 8 && std::memcmp(magic, "KABCRTC2", 8) == 0 && std::fread(&dg, 8, 1, f) == 1 &&
        dg == key_digest && std::fread(&ck, 8, 1, f) == 1 && std::fread(&n, 4, 1, f) == 1 && n == nnames) {
        ok = true;
        for (uint32_t i = 0; i < n && ok; ++i) {
            uint32_t len = 0;
            ok = std::fread(&len, 4, 1, f) == 1 && len < 4096;
            std::string s(ok ? len : 0, '\0');
            ok = ok && (len == 0 || std::fread(&s[0], 1, len, f) == len);
            lowered->push_back(s);
        }
        uint64_t cs = 0;
        ok = ok && std::fread(&cs, 8, 1, f) == 1 && cs > 0 && cs < (1ull << 31);
        if (ok) {
            code->resize((size_t)cs);
            ok = std::fread(code->data(), 1, (size_t)cs, f) == (size_t)cs && code_checksum(*code) == ck;
        }
    }
    std::fclose(f);
    if (!ok) {
        code->clear();
        lowered->clear();
    }
    return ok;
}

// The cache directory is bounded (KABC_RTC_CACHE_MB, default 512): before a code object is stored
// the oldest ones go until it fits -- a service that sees thousands of distinct models (every
// distinct prior tuple is a unit of its own) must not fill the disk.  Leftovers of workers that died
// (lock / job files older than ten minutes) go on the same occasion.
// a negative entry (<code object>.err: the compiler's message) counts for an hour
static constexpr time_t kErrTtlSeconds = 3600;
static bool fresh_err_file(const std::string& p) {
    struct stat st;
    if (lstat(p.c_str(), &st) != 0 || !S_ISREG(st.st_mode)) return false;
    if (time(nullptr) - st.st_mtime > kErrTtlSeconds) {
        (void)unlink(p.c_str());
        return false;
    }
    return true;
}

static void cache_trim(const std::string& dir, size_t incoming) {
    double cap_mb = 512.0;
    if (const char* e = std::getenv("KABC_RTC_CACHE_MB")) cap_mb = std::atof(e);
    if (!(cap_mb > 0.0)) return;
    const double cap = cap_mb * 1048576.0;
    struct Ent { std::string path; time_t mtime; double size; };
    std::vector<Ent> co;
    double total = (double)incoming;
    const time_t now = time(nullptr);
    if (DIR* d = opendir(dir.c_str())) {
        while (const dirent* e = readdir(d)) {
            const std::string n = e->d_name;
            if (n.size() < 6 || n.compare(0, 5, "kabc_") != 0) continue;
            const std::string p = dir + "/" + n;
            struct stat st;
            if (lstat(p.c_str(), &st) != 0 || !S_ISREG(st.st_mode)) continue;
            const bool is_co = n.size() > 3 && n.compare(n.size() - 3, 3, ".co") == 0;
            if (is_co) {
                co.push_back({p, st.st_mtime, (double)st.st_size});
                total += (double)st.st_size;
            } else if (now - st.st_mtime > 600 &&
                       (n.find(".lock") != std::string::npos || n.find(".job") != std::string::npos ||
                        n.find(".tmp") != std::string::npos)) {
                (void)unlink(p.c_str());
            } else if (now - st.st_mtime > kErrTtlSeconds && n.size() > 4 && n.compare(n.size() - 4, 4, ".err") == 0) {
                (void)unlink(p.c_str());  // (a failure of long ago is not a verdict: out of memory, a missing hipRTC ...)
            }
        }
        closedir(d);
    }
    if (total <= cap) return;
    std::sort(co.begin(), co.end(), [](const Ent& a, const Ent& b) { return a.mtime < b.mtime; });
    for (const Ent& e : co) {
        if (total <= cap) break;
        if (unlink(e.path.c_str()) == 0) total -= e.size;
    }
}

static bool cache_store(const std::string& dir, const std::string& path, uint64_t key_digest,
                        const std::vector<char>& code, const std::vector<std::string>& lowered) {
    if (!cache_dir_usable(dir)) return false;
    cache_trim(dir, code.size());
    const std::string tmp = path + "." + std::to_string((long long)getpid()) + ".tmp";
    (void)unlink(tmp.c_str());  // (a leftover of an earlier process with this pid)
    FILE* f = open_cache_file_new(tmp);
    if (!f) return false;  // (a read-only install: compile every time)
    const uint32_t n = (uint32_t)lowered.size();
    const uint64_t cs = code.size(), ck = code_checksum(code);
    bool ok = std::fwrite("KABCRTC2", 1, 8, f) == 8 && std::fwrite(&key_digest, 8, 1, f) == 1 &&
              std::fwrite(&ck, 8, 1, f) == 1 && std::fwrite(&n, 4, 1, f) == 1;
    for (const std::string& s : lowered) {
        const uint32_t len = (uint32_t)s.size();
        ok = ok && std::fwrite(&len, 4, 1, f) == 1 && (len == 0 || std::fwrite(s.data(), 1, len, f) == len);
    }
    ok = ok && std::fwrite(&cs, 8, 1, f) == 1 && std::fwrite(code.data(), 1, code.size(), f) == code.size();
    ok = (std::fclose(f) == 0) && ok;
    if (!ok || std::rename(tmp.c_str(), path.c_str()) != 0) {
        (void)std::remove(tmp.c_str());
        return false;
    }
    return true;
}

// everything that determines a code object: the unit's text, the compiler options, the cache path
struct RtcJob {
    std::string text;
    std::vector<std::string> opt;
    std::string cdir, cpath;  // on-disk cache (empty: disabled)
    uint64_t digest = 0;      // of text + names + options (stored inside the cache file, checked at load)
};

static void rtc_prepare(const std::string& head, const char* header, bool fma_c_vgpr,
                        const std::vector<std::string>& names, bool want_cache, RtcJob* J,
                        const std::vector<std::string>* extra_opt = nullptr) {
    J->text = "// generated by libkabc_hip (capi_plugin.hip)\n" + head;
    if (fma_c_vgpr) J->text += "#define KABC_FMA_C_VGPR\n";
    if (header) J->text += std::string("#include \"") + header + "\"\n";
    // the flags of csrc/Makefile: the arithmetic contract needs -ffp-contract=off
    J->opt = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math"};
    if (extra_opt) J->opt.insert(J->opt.end(), extra_opt->begin(), extra_opt->end());
    // the on-disk cache: keyed by everything that determines the code object
    J->cdir = want_cache ? rtc_cache_dir() : std::string();
    J->cpath.clear();
    if (!J->cdir.empty()) {
        uint64_t h = fnv1a(J->text.data(), J->text.size(), toolchain_fingerprint());
        uint64_t h2 = fnv1a(J->text.data(), J->text.size(), 0x9e3779b97f4a7c15ull ^ toolchain_fingerprint());
        for (const std::string& n : names) {
            h = fnv1a(n.data(), n.size() + 1, h);
            h2 = fnv1a(n.data(), n.size() + 1, h2);
        }
        for (const std::string& o : J->opt) h = fnv1a(o.data(), o.size() + 1, h);
        uint64_t dg = fnv1a(J->text.data(), J->text.size(), 0x6c62272e07bb0142ull ^ toolchain_fingerprint());
        for (const std::string& n : names) dg = fnv1a(n.data(), n.size() + 1, dg);
        for (const std::string& o : J->opt) dg = fnv1a(o.data(), o.size() + 1, dg);
        J->digest = dg;
        char nm[64];
        std::snprintf(nm, sizeof nm, "/kabc_%016llx%016llx.co", (unsigned long long)h, (unsigned long long)h2);
        J->cpath = J->cdir + nm;
    }
}

// hipRTC proper: `text` -> one gfx950 code object + the lowered names of `names`
static kabc_status_t rtc_compile_text(const std::string& text, std::vector<std::string> opt,
                                      const std::vector<std::string>& names, std::vector<char>* code,
                                      std::vector<std::string>* lowered) {
    Hiprtc* H = hiprtc();
    if (!H) {
        set_error("hipRTC is not available: %s(set KABC_HIPRTC_LIB, or register a plugin .so built "
                  "by hipcc with kabc_register_cost_plugin)", g_rtc.why.c_str());
        return KABC_ERR_DEVICE;
    }
    hiprtcProgram prog = nullptr;
    if (H->CreateProgram(&prog, text.c_str(), "kabc_rtc_unit.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) {
        set_error("hiprtcCreateProgram failed");
        return KABC_ERR_DEVICE;
    }
    for (const std::string& n : names) (void)H->AddNameExpression(prog, n.c_str());
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
        set_error("run-time compilation failed (%s):\n%s", H->GetErrorString(r), log.c_str());
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

// compile `names` (kernel name expressions) from `head` (snippets, model constants) + `header`
// (the kernel templates) into one code object
static kabc_status_t rtc_compile(const std::string& head, const char* header, bool fma_c_vgpr,
                                 const std::vector<std::string>& names, std::vector<char>* code,
                                 std::vector<std::string>* lowered, const std::vector<std::string>* extra_opt = nullptr) {
    RtcJob J;
    rtc_prepare(head, header, fma_c_vgpr, names, code != nullptr, &J, extra_opt);
    if (!J.cpath.empty() && cache_load(J.cpath, names.size(), J.digest, code, lowered)) return KABC_OK;
    if (kabc_status_t st = rtc_compile_text(J.text, J.opt, names, code, lowered)) return st;
    if (code && !J.cpath.empty()) (void)cache_store(J.cdir, J.cpath, J.digest, *code, *lowered);
    return KABC_OK;
}

// loads a code object on the current device and records its kernels (R->mu held)
static void* rtc_load(RtcCache* R, int dev, const std::vector<char>& code, const std::vector<std::string>& names,
                      const std::vector<std::string>& lowered, const std::string& want) {
    const std::string pre = std::to_string(dev) + "|";
    hipModule_t mod = nullptr;
    const hipError_t e = dev < 0 ? hipErrorNoDevice : hipModuleLoadData(&mod, code.data());
    if (e != hipSuccess) {
        set_error("the kernels compiled (%zu bytes of gfx950 code for %s), but hipModuleLoadData failed: %s",
                  code.size(), want.c_str(), hipGetErrorString(e));
        return nullptr;
    }
    R->mods.push_back(mod);
    for (size_t i = 0; i < names.size(); ++i) {
        hipFunction_t f = nullptr;
        if (!lowered[i].empty() && hipModuleGetFunction(&f, mod, lowered[i].c_str()) == hipSuccess)
            R->fns[pre + names[i]] = (void*)f;
    }
    auto it = R->fns.find(pre + want);
    if (it == R->fns.end()) {
        set_error("kernel %s is missing from the compiled unit", want.c_str());
        return nullptr;
    }
    return it->second;
}

// the kernel `want` of a family on the CURRENT device (every entry point selects its context's
// device before it looks kernels up); the whole batch `names` is compiled together on a miss
static void* rtc_kernel(RtcCache* R, const std::string& head, const char* header, bool fma_c_vgpr,
                        const std::vector<std::string>& names, const std::string& want) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) dev = -1;  // (no device: the compilation still runs, the load fails)
    std::lock_guard<std::mutex> lk(R->mu);
    auto it = R->fns.find(std::to_string(dev) + "|" + want);
    if (it != R->fns.end()) return it->second;
    std::vector<char> code;
    std::vector<std::string> lowered;
    const bool ais = header && std::strcmp(header, "ais_kernels.hpp") == 0 && !R->ais_opt.empty();
    if (rtc_compile(head, header, fma_c_vgpr, names, &code, &lowered, ais ? &R->ais_opt : nullptr) != KABC_OK) return nullptr;
    return rtc_load(R, dev, code, names, lowered, want);
}

// ---- compilation off the caller's thread: a worker PROCESS ----------------------------------
// The default path of a model that can be specialised must never wait for the compiler (2-20 s
// per AIS kernel): the caller starts on the prebuilt kernels, the unit's text goes to
// <library directory>/kabc_rtc_worker (csrc/rtc_worker.c: a detached process that loads this
// library and calls kabc_rtc_worker_main -- no GPU, no thread inside the host application, nothing
// to join at exit; it finishes and fills the on-disk cache even when its parent has gone), and the
// caller looks for the code object in the cache at its next launch boundaries.
struct SpecCounters {
    std::atomic<uint64_t> spawned{0}, loaded{0}, failed{0}, cache_hits{0};
};
static SpecCounters g_spec_counters;

static std::string worker_path() {
    if (const char* e = std::getenv("KABC_RTC_WORKER")) return e;
    const std::string ld = lib_dir();
    return ld.empty() ? std::string() : ld + "/kabc_rtc_worker";
}

static std::string self_lib_path() {
    Dl_info info;
    return (dladdr((const void*)&find_plugin, &info) && info.dli_fname) ? std::string(info.dli_fname) : std::string();
}

static bool async_compile_available() {
    const std::string w = worker_path();
    return !w.empty() && access(w.c_str(), X_OK) == 0 && !rtc_cache_dir().empty() && !self_lib_path().empty();
}

static bool file_exists(const std::string& p) {
    struct stat st;
    return stat(p.c_str(), &st) == 0;
}

static std::string read_small_file(const std::string& p) {
    std::string out;
    if (FILE* f = std::fopen(p.c_str(), "rb")) {
        char buf[2048];
        const size_t n = std::fread(buf, 1, sizeof buf - 1, f);
        out.assign(buf, n);
        std::fclose(f);
    }
    return out;
}

// compilations in flight in this cache directory, whoever started them (fresh lock files): the
// worker processes of ALL processes sharing the cache are bounded by KABC_RTC_WORKERS (default 2)
// -- a test-suite or a service creating a hundred models in a minute must not start a hundred
// compilers
static int workers_in_flight(const std::string& cdir) {
    int n = 0;
    if (DIR* dir = opendir(cdir.c_str())) {
        const time_t now = time(nullptr);
        while (const dirent* e = readdir(dir)) {
            const size_t len = std::strlen(e->d_name);
            if (len < 5 || std::strcmp(e->d_name + len - 5, ".lock") != 0) continue;
            struct stat st;
            if (stat((cdir + "/" + e->d_name).c_str(), &st) == 0 && now - st.st_mtime < 600) ++n;
        }
        closedir(dir);
    }
    return n;
}
static int max_workers() {
    const char* e = std::getenv("KABC_RTC_WORKERS");
    const int v = e ? std::atoi(e) : 2;
    return v < 1 ? 1 : v;
}

// writes the job file and starts the worker; false: could not (message in *why)
static bool spawn_worker(const RtcJob& J, const std::vector<std::string>& names, std::string* why) {
    const std::string lock = J.cpath + ".lock";
    const int lflags = O_CREAT | O_EXCL | O_WRONLY | O_CLOEXEC | O_NOFOLLOW;
    const int lfd = open(lock.c_str(), lflags, 0600);
    if (lfd < 0) {
        struct stat st;
        if (errno == EEXIST && lstat(lock.c_str(), &st) == 0 && time(nullptr) - st.st_mtime < 600)
            return true;  // another process (or handle) is compiling exactly this unit: wait for its result
        (void)unlink(lock.c_str());  // (a worker that died)
        const int l2 = open(lock.c_str(), lflags, 0600);
        if (l2 < 0) {
            if (errno == EEXIST) return true;  // (somebody else replaced the stale lock first: theirs to compile)
            *why = "cannot create " + lock + ": " + std::strerror(errno);
            return false;
        }
        close(l2);
    } else {
        close(lfd);
    }
    const std::string job = J.cpath + ".job";
    (void)unlink(job.c_str());  // (the lock is ours: a job file of this name is a leftover)
    FILE* f = open_cache_file_new(job);
    bool ok = f != nullptr;
    if (ok) {
        std::string hd = "KABCJOB2\n" + J.cpath + "\n" + std::to_string((unsigned long long)J.digest) + "\n" +
                         std::to_string(J.opt.size()) + "\n";
        for (const std::string& o : J.opt) hd += o + "\n";
        hd += std::to_string(names.size()) + "\n";
        for (const std::string& n : names) hd += n + "\n";
        hd += std::to_string(J.text.size()) + "\n";
        ok = std::fwrite(hd.data(), 1, hd.size(), f) == hd.size() &&
             std::fwrite(J.text.data(), 1, J.text.size(), f) == J.text.size();
        ok = (std::fclose(f) == 0) && ok;
    }
    if (!ok) {
        *why = "cannot write " + job;
        (void)unlink(job.c_str());
        (void)unlink(lock.c_str());
        return false;
    }
    // the worker's environment: ours without the variables that would make a tool (profiler,
    // preload) initialise the GPU inside it
    std::vector<char*> envp;
    for (char** e = ::environ; e && *e; ++e) {
        static const char* drop[] = {"LD_PRELOAD=", "HSA_TOOLS_LIB=", "ROCP_", "ROCPROF", "ROCTRACER_", "HIP_PROFILE"};
        bool skip = false;
        for (const char* d : drop) skip = skip || std::strncmp(*e, d, std::strlen(d)) == 0;
        if (!skip) envp.push_back(*e);
    }
    envp.push_back(nullptr);
    const std::string w = worker_path(), lib = self_lib_path();
    char* argv[] = {const_cast<char*>(w.c_str()), const_cast<char*>(lib.c_str()), const_cast<char*>(job.c_str()), nullptr};
    posix_spawn_file_actions_t fa;
    posix_spawn_file_actions_init(&fa);
    posix_spawn_file_actions_addopen(&fa, 0, "/dev/null", O_RDONLY, 0);
    posix_spawn_file_actions_addopen(&fa, 1, "/dev/null", O_WRONLY, 0);
    posix_spawn_file_actions_addopen(&fa, 2, "/dev/null", O_WRONLY, 0);
    pid_t pid = 0;
    const int rc = posix_spawn(&pid, w.c_str(), &fa, nullptr, argv, envp.data());
    posix_spawn_file_actions_destroy(&fa);
    if (rc != 0) {
        *why = "cannot start " + w + ": " + std::strerror(rc);
        (void)unlink(job.c_str());
        (void)unlink(lock.c_str());
        return false;
    }
    // (the worker detaches at once -- it forks and its first process exits: nothing of ours stays
    // a zombie, and the host application's own child handling never sees it)
    int wst = 0;
    while (waitpid(pid, &wst, 0) < 0 && errno == EINTR) {
    }
    g_spec_counters.spawned++;
    return true;
}

// The non-blocking form of rtc_kernel: the kernel when it is loaded already or its code object
// lies in the on-disk cache (loaded now, ~1 ms); otherwise nullptr -- the first such call hands
// the compilation to the worker, later ones look for its result at most every 2 ms.
// *failed: the compilation cannot or did not succeed (never ask again).
static void* rtc_kernel_try(RtcCache* R, const std::string& head, const char* header, bool fma_c_vgpr,
                            const std::vector<std::string>& names, const std::string& want, bool* failed) {
    *failed = false;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) dev = -1;
    std::lock_guard<std::mutex> lk(R->mu);
    auto it = R->fns.find(std::to_string(dev) + "|" + want);
    if (it != R->fns.end()) return it->second;
    const std::vector<std::string>* ais_opt =
        (header && std::strcmp(header, "ais_kernels.hpp") == 0 && !R->ais_opt.empty()) ? &R->ais_opt : nullptr;
    const auto now = std::chrono::steady_clock::now();
    auto jt = R->jobs.find(want);
    if (jt != R->jobs.end()) {
        RtcAsyncJob& job = jt->second;
        if (job.state == RtcAsyncJob::kFailed) {
            *failed = true;
            return nullptr;
        }
        if (now - job.t_poll < std::chrono::milliseconds(job.state == RtcAsyncJob::kDeferred ? 50 : 2)) return nullptr;
        job.t_poll = now;
        if (job.state == RtcAsyncJob::kDeferred) {  // the workers were busy: look again, start it if there is room
            RtcJob J;
            rtc_prepare(head, header, fma_c_vgpr, names, true, &J, ais_opt);
            if (file_exists(J.cpath) || fresh_err_file(J.cpath + ".err") || file_exists(J.cpath + ".lock")) {
                job.state = RtcAsyncJob::kPending;  // (somebody else took it meanwhile)
            } else if (workers_in_flight(J.cdir) < max_workers()) {
                job.t_spawn = now;
                if (spawn_worker(J, names, &job.why)) {
                    job.state = RtcAsyncJob::kPending;
                } else {
                    job.state = RtcAsyncJob::kFailed;
                    g_spec_counters.failed++;
                    *failed = true;
                }
            }
            return nullptr;
        }
        std::vector<char> code;
        std::vector<std::string> lowered;
        if (file_exists(job.cpath) && cache_load(job.cpath, names.size(), job.digest, &code, &lowered)) {
            void* fn = rtc_load(R, dev, code, names, lowered, want);
            if (fn) {
                g_spec_counters.loaded++;
                return fn;  // (the job entry stays: other devices load from the cache the same way)
            }
            job.state = RtcAsyncJob::kFailed;
            job.why = get_error();
        } else if (fresh_err_file(job.cpath + ".err")) {
            job.state = RtcAsyncJob::kFailed;
            job.why = read_small_file(job.cpath + ".err");
        } else if (now - job.t_spawn > std::chrono::seconds(900)) {
            job.state = RtcAsyncJob::kFailed;
            job.why = "the compilation worker did not deliver within 900 s";
        }
        if (job.state == RtcAsyncJob::kFailed) {
            g_spec_counters.failed++;
            *failed = true;
        }
        return nullptr;
    }
    // first request
    RtcJob J;
    rtc_prepare(head, header, fma_c_vgpr, names, true, &J, ais_opt);
    RtcAsyncJob job;
    job.cpath = J.cpath;
    job.digest = J.digest;
    job.t_spawn = job.t_poll = now;
    if (J.cpath.empty()) {
        job.state = RtcAsyncJob::kFailed;
        job.why = "no code-object cache directory";
    } else {
        std::vector<char> code;
        std::vector<std::string> lowered;
        if (cache_load(J.cpath, names.size(), J.digest, &code, &lowered)) {
            if (void* fn = rtc_load(R, dev, code, names, lowered, want)) {
                g_spec_counters.cache_hits++;
                return fn;
            }
            job.state = RtcAsyncJob::kFailed;
            job.why = get_error();
        } else if (fresh_err_file(J.cpath + ".err")) {
            job.state = RtcAsyncJob::kFailed;
            job.why = read_small_file(J.cpath + ".err");
        } else if (!file_exists(J.cpath + ".lock") && workers_in_flight(J.cdir) >= max_workers()) {
            job.state = RtcAsyncJob::kDeferred;
        } else if (!spawn_worker(J, names, &job.why)) {
            job.state = RtcAsyncJob::kFailed;
        }
    }
    if (job.state == RtcAsyncJob::kFailed) {
        g_spec_counters.failed++;
        *failed = true;
    }
    R->jobs.emplace(want, job);
    return nullptr;
}

static constexpr int kPriorClassesRt = 4;  // kPriorClasses of ais_kernels.hpp (static_assert there)

// one kernel family compiled from `head`: `cost` is the COST template argument of the kernels
// (a built-in DeviceCost id, or KABC_COST_USER when `head` carries a user cost), pk_mask the
// posterior kinds the AIS kernel may be asked for
static PluginKernel rtc_family_kernel(RtcCache* R, const std::string& head, int cost, int pk_mask,
                                      int family, int D, int variant, bool* try_failed = nullptr) {
    PluginKernel k;
    // try_failed != nullptr: the non-blocking form (rtc_kernel_try)
    auto rtc_kernel = [try_failed](RtcCache* R_, const std::string& head_, const char* header, bool fma,
                                   const std::vector<std::string>& names, const std::string& want) -> void* {
        if (try_failed) return rtc_kernel_try(R_, head_, header, fma, names, want, try_failed);
        return kabc::rtc_kernel(R_, head_, header, fma, names, want);
    };
    const bool big = D > KABC_MAX_DIM;  // run-time-dimension kernels: the D = 0 / dyn instantiations
    if (big && family != kPfAisDyn && family != kPfSmcDyn && family != kPfAbcdeInit && family != kPfAbcdeGen &&
        family != kPfAttempt && family != kPfPriorLogpdf && family != kPfPriorRand)
        return k;
    const std::string d = std::to_string(big ? 0 : D), u = std::to_string(cost);
    // (the dyn kernels dispatch a built-in cost at run time: their COST argument only says "user cost")
    // (run-time-dimension kernels: the user cost, one of the built-in costs that take any number of parameters
    // -- compile-time dispatch, ais_dyn.hip -- or 0: dispatched inside the kernel)
    const bool any_d = cost == (int)KABC_COST_GAUSS_DIST || cost == (int)KABC_COST_ROSENBROCK ||
                       cost == (int)KABC_COST_HIER_GAUSS_SIM || cost == (int)KABC_COST_NORM_SHELL;
    const std::string udyn = std::to_string(cost == (int)KABC_COST_USER ? (int)KABC_COST_USER : any_d ? cost : 0);
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
            if (!((pk_mask >> (pk - 1)) & 1) || pc == 3) return k;  // (NORMAL class: the host falls back to SIMPLE)
            const std::string half = "kabc::ais_half_kernel<" + d + ", " + u + ", " + std::to_string(pc) +
                                     ", " + std::to_string(pk) + ">";
            k.mod = rtc_kernel(R, head, "ais_kernels.hpp", true, {half, ais_init}, half);
            break;
        }
        case kPfAisSmall: {
            const int pc = variant % kPriorClassesRt, pk = variant / kPriorClassesRt + 1;
            if (!((pk_mask >> (pk - 1)) & 1) || pc == 3 || pc == 1) return k;  // (BOX and GENERAL: capi_ais.hip small_class)
            const std::string n = "kabc::ais_small_kernel<" + d + ", " + u + ", " + std::to_string(pc) + ", " +
                                  std::to_string(pk) + ">";
            k.mod = rtc_kernel(R, head, "ais_small_kernel.hpp", true, {n}, n);
            break;
        }
        case kPfAisInit: k.mod = rtc_kernel(R, head, "ais_kernels.hpp", true, {ais_init}, ais_init); break;
        case kPfSmcInit: {
            // (requested before the pass kernels of the same run: compile for the class it will use)
            const std::vector<std::string> n = smc_names(variant);
            k.mod = rtc_kernel(R, head, "smc_loop_kernel.hpp", false, n, n[0]);
            break;
        }
        case kPfSmc: {
            const std::vector<std::string> n = smc_names(variant);
            k.mod = rtc_kernel(R, head, "smc_loop_kernel.hpp", false, n, n[1]);
            break;
        }
        case kPfSmcLoop: {
            const std::vector<std::string> n = smc_names(variant);
            k.mod = rtc_kernel(R, head, "smc_loop_kernel.hpp", false, n, n[2]);
            break;
        }
        case kPfAbcdeInit:
        case kPfAbcdeGen: {
            const std::vector<std::string> n = {"kabc::abcde_init_kernel<" + d + ">",
                                                "kabc::abcde_gen_kernel<" + d + ">"};
            k.mod = rtc_kernel(R, head, "abcde_kernels.hpp", false, n, n[family == kPfAbcdeInit ? 0 : 1]);
            break;
        }
        case kPfAttempt: {
            const std::string n = "kabc::pf_attempt_kernel<" + d + ">";
            k.mod = rtc_kernel(R, head, "pfilter_kernels.hpp", false, {n}, n);
            break;
        }
        case kPfSmcSmall: {
            const std::string n = "kabc::smc_small_kernel<" + d + ", " + u + ", " + (variant ? "true" : "false") + ">";
            k.mod = rtc_kernel(R, head, "smc_small_kernel.hpp", false, {n}, n);
            break;
        }
        case kPfAisDyn: {
            // variants: 0 / 2 / 3 the half-generation kernel with teams of 16 / 8 / 64 lanes per walker, 1 init
            const std::vector<std::string> n = {"kabc::ais_dyn_half_kernel<" + udyn + ", 16>",
                                                "kabc::ais_dyn_init_kernel<" + udyn + ">",
                                                "kabc::ais_dyn_half_kernel<" + udyn + ", 8>",
                                                "kabc::ais_dyn_half_kernel<" + udyn + ", 64>"};
            if (variant < 0 || variant > 3) return k;
            k.mod = rtc_kernel(R, head, "ais_dyn_kernels.hpp", false, n, n[(size_t)variant]);
            break;
        }
        case kPfSmcDyn: {
            // variants: 0 the thread-per-particle pass, 1 init, 2 / 3 / 4 the pass with teams of 8 / 16 / 64 lanes
            // per particle, 5 the statistics kernel behind a team pass
            const std::vector<std::string> n = {"kabc::smc_dyn_mcmc_kernel<" + udyn + ">",
                                                "kabc::smc_dyn_init_kernel<" + udyn + ">",
                                                "kabc::smc_dyn_team_kernel<" + udyn + ", 8>",
                                                "kabc::smc_dyn_team_kernel<" + udyn + ", 16>",
                                                "kabc::smc_dyn_team_kernel<" + udyn + ", 64>",
                                                "kabc::smc_dyn_part_kernel<" + udyn + ">"};
            if (variant < 0 || variant > 5) return k;
            k.mod = rtc_kernel(R, head, "smc_dyn_kernels.hpp", false, n, n[(size_t)variant]);
            break;
        }
        case kPfPriorLogpdf:
        case kPfPriorRand: {
            const std::vector<std::string> n = {"kabc::prior_logpdf_kernel", "kabc::prior_rand_kernel"};
            k.mod = rtc_kernel(R, head, "prior_util_kernels.hpp", false, n, n[family == kPfPriorLogpdf ? 0 : 1]);
            break;
        }
        default: break;
    }
    return k;
}

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
    if (D < 1 || D > KABC_MAX_DIM_DYN || !rtc_dim_listed(R, D)) return k;
    return rtc_family_kernel(R, cost_head(R), (int)KABC_COST_USER, R->pk_mask, family, D, variant);
}

// ---- model units -----------------------------------------------------------------------------
struct ModelUnit : RtcCache {
    std::string head;
    int cost_tmpl = 0;   // COST template argument
    int pk_mask = 7;
    bool spec = false;
    // a specialisation is found again by exactly its components, D and cost id
    int cost_id = 0;
    std::vector<kabc_prior_t> prior;
    int handle = 0;      // kabc_compile_model registration (0: created by an entry point on its own)
    bool released = false;
    // created by an entry point at first sight of the model (the default): its kernels are compiled
    // by the worker process and taken when they are there (unit_kernel never waits for them)
    bool async = false;
    // the unit the model's kernels MUST come from while / when this one cannot serve (a prior with
    // user families has no prebuilt kernels: their generic unit); nullptr: the prebuilt kernels
    ModelUnit* fallback = nullptr;
};
static std::map<std::string, ModelUnit*> g_units;  // generic units by key (g_mu)
static std::vector<ModelUnit*> g_specs;            // specialised units (g_mu)

bool unit_is_spec(const ModelUnit* u) { return u && u->spec; }
bool unit_required(const ModelUnit* u) { return u && (!u->spec || u->fallback != nullptr); }

static std::vector<int> user_kinds_of(const kabc_prior_t* prior, int D) {
    std::vector<int> ks;
    for (int k = 0; k < D; ++k)
        if (prior[k].kind >= KABC_PRIOR_USER) ks.push_back(prior[k].kind);
    std::sort(ks.begin(), ks.end());
    ks.erase(std::unique(ks.begin(), ks.end()), ks.end());
    return ks;
}

// can (and should) this prior be specialised?  Not a pure box (its kernel class is
// parameter-free already), not an MvNormal (a device block behind a pointer), D within the
// register-resident kernels
static bool spec_eligible(const kabc_prior_t* prior, int D) {
    if (D < 1 || D > KABC_MAX_DIM) return false;
    bool allbox = true, allnormal = true;
    for (int k = 0; k < D; ++k) {
        const int kd = prior[k].kind;
        if (kd == KABC_PRIOR_MVNORMAL || kd == KABC_PRIOR_USER_INIT) return false;
        if (kd >= KABC_PRIOR_USER && user_prior_is_joint(kd)) return false;  // (one density of the vector: nothing per component to fold)
        allbox = allbox && (kd == KABC_PRIOR_UNIFORM || kd == KABC_PRIOR_DISCRETE_UNIFORM);
        allnormal = allnormal && kd == KABC_PRIOR_NORMAL;
    }
    // plain Normals up to seven parameters: the prebuilt NORMAL class (no support tests, no
    // rounding, scheduled for ILP: csrc/Makefile AIS_SCHED / AIS_SPLIT) is the faster kernel --
    // C2 (4096 x 2): 51.6 us per launch against 54.8 for the model's own kernel with either
    // scheduling strategy, 70.3 against 70.6 at 65 536 walkers (profiles/r05_c2_spec_ab.txt); from
    // eight parameters on the model's own wins (0.72 -> 0.78 of the contract roofline)
    if (allnormal && D <= 7) return false;
    return !allbox;
}

static bool same_prior(const std::vector<kabc_prior_t>& a, const kabc_prior_t* b, int D) {
    if ((int)a.size() != D) return false;
    for (int k = 0; k < D; ++k)
        if (a[(size_t)k].kind != b[k].kind || std::memcmp(a[(size_t)k].p, b[k].p, sizeof b[k].p) != 0) return false;
    return true;
}

// the head of a unit for (prior kinds, cost); *cost_tmpl / *pk_mask as rtc_family_kernel wants them
static kabc_status_t unit_head(const kabc_prior_t* prior, int D, int cost_id, bool spec, std::string* head,
                               int* cost_tmpl, int* pk_mask) {
    *cost_tmpl = cost_id;
    *pk_mask = 7;
    head->clear();
    if (cost_id >= KABC_COST_USER) {
        const CostPlugin* cp = find_plugin(cost_id);
        if (!cp || !cp->rtc) {
            set_error("a prior with user families / a specialised model needs its user cost in the hipRTC form "
                      "(kabc_compile_cost_plugin), not a plugin .so built by hipcc (cost id %d)", cost_id);
            return KABC_ERR_UNSUPPORTED;
        }
        *head += cost_head(cp->rtc);
        *cost_tmpl = (int)KABC_COST_USER;
        *pk_mask = cp->rtc->pk_mask;
    }
    *head += priors_head(user_kinds_of(prior, D));
    if (spec) {
        std::vector<PriorDev> q((size_t)D);
        for (int k = 0; k < D; ++k)
            if (!prepare_prior(prior[k], q[(size_t)k])) {
                set_error("invalid prior (kind/parameters) of component %d", k + 1);
                return KABC_ERR_INVALID_ARG;
            }
        *head += spec_head(q.data(), D);
    }
    return KABC_OK;
}

// KABC_SPECIALIZE: unset -- every entry point specialises on its own WITHOUT waiting (worker
// process, see rtc_kernel_try); 1 -- the same, compiling at first sight (blocking: tests, warm-up
// scripts); 0 -- never (registered kabc_compile_model units are ignored too)
// kabc_set_specialize: the embedding host's own word (-1: the environment decides, as above)
static std::atomic<int> g_spec_mode{-1};
static bool specialize_env() {
    const int m = g_spec_mode.load();
    if (m >= 0) return m == KABC_SPECIALIZE_BLOCKING;
    const char* e = std::getenv("KABC_SPECIALIZE");
    return e && *e && *e != '0';
}
static bool specialize_off() {
    const int m = g_spec_mode.load();
    if (m >= 0) return m == KABC_SPECIALIZE_OFF;
    const char* e = std::getenv("KABC_SPECIALIZE");
    return e && *e == '0';
}

static kabc_status_t make_spec_unit(const kabc_prior_t* prior, int D, int cost_id, int handle, bool async,
                                    ModelUnit** out) {
    ModelUnit* u = new ModelUnit();
    if (kabc_status_t st = unit_head(prior, D, cost_id, true, &u->head, &u->cost_tmpl, &u->pk_mask)) {
        delete u;
        return st;
    }
    if (std::getenv("KABC_SPEC_INJECT_ERROR"))  // (tests: a specialisation whose compilation fails)
        u->head += "#error \"KABC_SPEC_INJECT_ERROR\"\n";
    if (const char* e = std::getenv("KABC_SPEC_SCHED"))  // (probe: the scheduling strategy of csrc/Makefile AIS_SCHED)
        if (*e == '1') u->ais_opt = {"-mllvm", "-amdgpu-sched-strategy=max-ilp"};
    u->spec = true;
    u->async = async;
    u->cost_id = cost_id;
    u->prior.assign(prior, prior + D);
    u->handle = handle;
    std::lock_guard<std::mutex> lk(g_mu);
    // (two threads at the first sight of one model: one unit)
    for (ModelUnit* v : g_specs)
        if (!v->released && v->cost_id == cost_id && same_prior(v->prior, prior, D)) {
            delete u;
            *out = v;
            return KABC_OK;
        }
    g_specs.push_back(u);
    *out = u;
    return KABC_OK;
}

// the generic unit of the user families among `prior` (there are no prebuilt kernels for them)
static kabc_status_t generic_unit_for(const kabc_prior_t* prior, int D, int cost_id, const std::vector<int>& uk,
                                      ModelUnit** out) {
    std::string key = "c" + std::to_string(cost_id) + "|";
    for (int k : uk) key += std::to_string(k) + ",";
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_units.find(key);
        if (it != g_units.end()) {
            *out = it->second;
            return KABC_OK;
        }
    }
    if (!hiprtc()) {
        set_error("hipRTC is not available: %sa prior with user families has no prebuilt kernels", g_rtc.why.c_str());
        return KABC_ERR_DEVICE;
    }
    ModelUnit* u = new ModelUnit();
    if (kabc_status_t st = unit_head(prior, D, cost_id, false, &u->head, &u->cost_tmpl, &u->pk_mask)) {
        delete u;
        return st;
    }
    u->cost_id = cost_id;
    std::lock_guard<std::mutex> lk(g_mu);
    auto ins = g_units.emplace(key, u);
    if (!ins.second) delete u;
    *out = ins.first->second;
    return KABC_OK;
}

kabc_status_t model_unit_for(const kabc_prior_t* prior, int D, int cost_id, ModelUnit** out, bool allow_spec) {
    *out = nullptr;
    if (!prior || D < 1) return KABC_OK;
    const std::vector<int> uk = user_kinds_of(prior, D);
    for (int k : uk)
        if (!find_prior_plugin(k)) {
            set_error("prior kind %d is not a registered user family (kabc_compile_prior_plugin)", k);
            return KABC_ERR_INVALID_ARG;
        }
    if (!uk.empty() && D > KABC_MAX_DIM_DYN) {
        set_error("a prior with user families supports length(prior) <= %d (got %d)", KABC_MAX_DIM_DYN, D);
        return KABC_ERR_UNSUPPORTED;
    }
    // 1. the unit the kernels must come from when no specialisation serves
    ModelUnit* generic = nullptr;
    if (!uk.empty())
        if (kabc_status_t st = generic_unit_for(prior, D, cost_id, uk, &generic)) return st;
    *out = generic;
    // 2. a specialisation of exactly this model: registered (kabc_compile_model), or made here
    if (allow_spec && !specialize_off() && spec_eligible(prior, D)) {
        ModelUnit* spec = nullptr;
        {
            std::lock_guard<std::mutex> lk(g_mu);
            for (ModelUnit* u : g_specs)
                if (!u->released && u->cost_id == cost_id && same_prior(u->prior, prior, D)) spec = u;
        }
        if (!spec && hiprtc()) {
            // (a user cost built by hipcc has no text to specialise: unit_head refuses, the other kernels serve)
            const bool sync = specialize_env();
            if (sync || async_compile_available()) {
                const std::string keep = get_error();
                if (make_spec_unit(prior, D, cost_id, 0, !sync, &spec) != KABC_OK) {
                    spec = nullptr;
                    set_error("%s", keep.c_str());
                }
            }
        }
        if (spec) {
            spec->fallback = generic;  // (the same generic unit every time: its key is (cost, kinds))
            *out = spec;
        }
    }
    return KABC_OK;
}

static bool is_init_family(int family) {
    return family == kPfAisInit || family == kPfSmcInit || family == kPfAbcdeInit;
}

PluginKernel unit_kernel(ModelUnit* u, int family, int D, int variant, int* spec_state) {
    if (spec_state) *spec_state = KABC_SPEC_NONE;
    if (!u || D < 1 || D > KABC_MAX_DIM_DYN) return PluginKernel();
    if (!u->spec) return rtc_family_kernel(u, u->head, u->cost_tmpl, u->pk_mask, family, D, variant);
    PluginKernel k;
    if (!u->async) {
        k = rtc_family_kernel(u, u->head, u->cost_tmpl, u->pk_mask, family, D, variant);
        if (spec_state) *spec_state = k.mod ? KABC_SPEC_ACTIVE : KABC_SPEC_FAILED;
    } else if (!is_init_family(family)) {
        // (the one-off init kernels are not worth a compilation: they stay generic / prebuilt)
        bool failed = false;
        const std::string keep = get_error();
        k = rtc_family_kernel(u, u->head, u->cost_tmpl, u->pk_mask, family, D, variant, &failed);
        if (!k.mod) set_error("%s", keep.c_str());  // (a pending or failed background job is not the caller's error)
        if (spec_state) *spec_state = k.mod ? KABC_SPEC_ACTIVE : failed ? KABC_SPEC_FAILED : KABC_SPEC_PENDING;
    }
    if (k.mod || !u->fallback) return k;
    return rtc_family_kernel(u->fallback, u->fallback->head, u->fallback->cost_tmpl, u->fallback->pk_mask, family,
                             D, variant);
}

bool cost_dim_ok_rt(int cost_id, int D) {
    if (cost_id < KABC_COST_USER) return kabc_cost_dim_ok(cost_id, D) != 0;
    const CostPlugin* p = find_plugin(cost_id);
    if (p && p->rtc) return D >= 1 && D <= KABC_MAX_DIM_DYN && rtc_dim_listed(p->rtc, D);
    return p && p->dim_ok(D) != 0;
}

hipError_t rtc_launch(void* fn, dim3 grid, dim3 block, const void* args, hipStream_t s) {
    return rtc_launch_lds(fn, grid, block, args, s, 0u);
}
hipError_t rtc_launch_lds(void* fn, dim3 grid, dim3 block, const void* args, hipStream_t s, unsigned lds_bytes) {
    void* params[] = {const_cast<void*>(args)};
    return hipModuleLaunchKernel((hipFunction_t)fn, grid.x, grid.y, grid.z, block.x, block.y, block.z, lds_bytes, s,
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
        if (dims[i] < 1 || dims[i] > KABC_MAX_DIM_DYN) {
            delete R;
            set_error("kabc_compile_cost_plugin: dims must lie in 1..%d", KABC_MAX_DIM_DYN);
            return KABC_ERR_UNSUPPORTED;
        }
        R->dims.push_back(dims[i]);
        R->dim_cond += (i ? " || (D) == " : "(D) == ") + std::to_string(dims[i]);
    }
    // the snippet alone first: its errors come back now, with the compiler's message, not at the
    // first sample() / smc() call
    if (kabc_status_t st = rtc_compile(cost_head(R), nullptr, false, {}, nullptr, nullptr)) {
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
    if (family < kPfAis || (family > kPfAttempt && family != kPfSmcSmall && family != kPfAisDyn && family != kPfSmcDyn &&
                            family != kPfAisSmall)) {
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

extern "C" kabc_status_t kabc_compile_mvprior_plugin(const char* src, int32_t* out_kind) {
    if (!src || !out_kind) {
        set_error("kabc_compile_mvprior_plugin: NULL argument");
        return KABC_ERR_INVALID_ARG;
    }
    PriorPlugin* P = new PriorPlugin();
    P->src = src;
    P->discrete = 0;
    P->joint = true;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        for (const PriorPlugin* q : g_prior_plugins)  // (the same snippet registered again is the same family)
            if (q->joint && q->src == P->src) {
                *out_kind = q->kind;
                delete P;
                return KABC_OK;
            }
        P->kind = KABC_PRIOR_USER + (int32_t)g_prior_plugins.size();
        g_prior_plugins.push_back(P);
    }
    // the snippet alone first, with both functions called the way the kernels call them
    const std::string sk = std::to_string(P->kind);
    const std::string check = priors_head({P->kind}) +
        "extern \"C\" __global__ void kabc_mvprior_check(double* o, const double* p, const kabc_slotwin_t* w) {\n"
        "    o[0] = KABC_USER_MVPRIOR_LOGPDF(" + sk + ", o + 1, 3, p, 5, kabc_log_tab);\n"
        "    KABC_USER_MVPRIOR_RAND(" + sk + ", o + 4, 3, p, 5, w);\n}\n";
    if (kabc_status_t st = rtc_compile(check, nullptr, false, {}, nullptr, nullptr)) {
        std::lock_guard<std::mutex> lk(g_mu);
        P->src = "#error \"this joint prior failed to compile at registration\"\n";  // (the kind stays taken)
        return st;
    }
    *out_kind = P->kind;
    return KABC_OK;
}

extern "C" kabc_status_t kabc_compile_prior_plugin(const char* src, int32_t discrete, int32_t* out_kind) {
    if (!src || !out_kind) {
        set_error("kabc_compile_prior_plugin: NULL argument");
        return KABC_ERR_INVALID_ARG;
    }
    PriorPlugin* P = new PriorPlugin();
    P->src = src;
    P->discrete = discrete ? 1 : 0;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        // (the same snippet registered again is the same family)
        for (const PriorPlugin* q : g_prior_plugins)
            if (!q->joint && q->src == P->src && q->discrete == P->discrete) {
                *out_kind = q->kind;
                delete P;
                return KABC_OK;
            }
        P->kind = KABC_PRIOR_USER + (int32_t)g_prior_plugins.size();
        g_prior_plugins.push_back(P);
    }
    // the snippet alone first, with both functions called the way the kernels call them: its
    // errors come back now, with the compiler's message
    const std::string sk = std::to_string(P->kind);
    const std::string check = priors_head({P->kind}) +
        "extern \"C\" __global__ void kabc_prior_check(double* o, const double* p, const kabc_slotwin_t* w) {\n"
        "    o[0] = KABC_USER_PRIOR_LOGPDF(" + sk + ", o[1], p, kabc_log_tab);\n"
        "    o[2] = KABC_USER_PRIOR_RAND(" + sk + ", p, w);\n}\n";
    if (kabc_status_t st = rtc_compile(check, nullptr, false, {}, nullptr, nullptr)) {
        std::lock_guard<std::mutex> lk(g_mu);
        P->src = "#error \"this prior family failed to compile at registration\"\n";  // (the kind stays taken)
        return st;
    }
    *out_kind = P->kind;
    return KABC_OK;
}

// the kernel families (and their variants) a model's entry points will ask its specialised unit for
struct SpecReq { int family, variant; };
static std::vector<SpecReq> spec_requests(const kabc_model_t* model, int families, int pk_mask) {
    const int D = model->D;
    if (families == 0) families = KABC_FAMILY_AIS | KABC_FAMILY_SMC;
    bool simple = true;
    for (int k = 0; k < D; ++k) simple = simple && prior_is_simple(model->prior[k].kind);
    std::vector<SpecReq> reqs;
    if (families & KABC_FAMILY_AIS) {
        const int pk_lo = (model->posterior >= 1 && model->posterior <= 3) ? model->posterior : 1;
        const int pk_hi = (model->posterior >= 1 && model->posterior <= 3) ? model->posterior : 3;
        for (int pk = pk_lo; pk <= pk_hi; ++pk)
            if ((pk_mask >> (pk - 1)) & 1) reqs.push_back({kPfAis, 2 + kPriorClassesRt * (pk - 1)});
    }
    if ((families & KABC_FAMILY_AIS_SMALL) && D <= KABC_MAX_DIM) {  // (the one-workgroup driver of small ensembles)
        const int pk_lo = (model->posterior >= 1 && model->posterior <= 3) ? model->posterior : 1;
        const int pk_hi = (model->posterior >= 1 && model->posterior <= 3) ? model->posterior : 3;
        for (int pk = pk_lo; pk <= pk_hi; ++pk)
            if ((pk_mask >> (pk - 1)) & 1) reqs.push_back({kPfAisSmall, 2 + kPriorClassesRt * (pk - 1)});
    }
    if (families & KABC_FAMILY_SMC) reqs.push_back({kPfSmcLoop, simple ? 1 : 0});
    if (families & KABC_FAMILY_ABCDE) reqs.push_back({kPfAbcdeGen, 0});
    if (families & KABC_FAMILY_PFILTER) {
        reqs.push_back({kPfAbcdeInit, 0});
        reqs.push_back({kPfAttempt, 0});
    }
    return reqs;
}

extern "C" kabc_status_t kabc_compile_model(const kabc_model_t* model, int32_t families, int32_t* out_handle) {
    if (!model || !model->prior) {
        set_error("kabc_compile_model: NULL argument");
        return KABC_ERR_INVALID_ARG;
    }
    const int D = model->D;
    if (out_handle) *out_handle = 0;
    if (!spec_eligible(model->prior, D)) return KABC_OK;  // (left to the prebuilt kernels: include/kabc.h)
    if (!cost_dim_ok_rt(model->cost.id, D)) {
        set_error("DeviceCost id %d does not accept D = %d", model->cost.id, D);
        return KABC_ERR_UNSUPPORTED;
    }
    if (!hiprtc()) {
        set_error("hipRTC is not available: %sthe prebuilt kernels remain the path", g_rtc.why.c_str());
        return KABC_ERR_DEVICE;
    }
    for (int k : user_kinds_of(model->prior, D))
        if (!find_prior_plugin(k)) {
            set_error("prior kind %d is not a registered user family (kabc_compile_prior_plugin)", k);
            return KABC_ERR_INVALID_ARG;
        }
    ModelUnit* u = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        for (ModelUnit* v : g_specs)
            if (!v->released && v->cost_id == model->cost.id && same_prior(v->prior, model->prior, D)) u = v;
    }
    if (!u) {
        static int next_handle = 0;
        int h;
        {
            std::lock_guard<std::mutex> lk(g_mu);
            h = ++next_handle;
        }
        if (kabc_status_t st = make_spec_unit(model->prior, D, model->cost.id, h, false, &u)) return st;
    }
    {
        // (a unit an entry point made on its own: from now on it is this registration's, and its
        // kernels are compiled here and now)
        std::lock_guard<std::mutex> lk(g_mu);
        if (u->handle == 0) {
            static int next_auto = 1 << 20;
            u->handle = ++next_auto;
        }
        u->async = false;
    }
    if (out_handle) *out_handle = u->handle;
    // compile (and, with a device, load) the requested families now
    int dev = -1;
    const bool have_dev = hipGetDevice(&dev) == hipSuccess && dev >= 0;
    const std::vector<SpecReq> reqs = spec_requests(model, families, u->pk_mask);
    for (const SpecReq& r : reqs) {
        set_error("%s", "");
        const PluginKernel k = unit_kernel(u, r.family, D, r.variant);
        if (k.mod) continue;
        // without a device the code object is in the on-disk cache now; it is loaded at first use
        if (!have_dev && std::strstr(get_error(), "hipModuleLoadData failed")) continue;
        if (!get_error()[0]) set_error("kabc_compile_model: kernel family %d is not available for this model", r.family);
        return KABC_ERR_UNSUPPORTED;
    }
    set_error("%s", "");
    return KABC_OK;
}

extern "C" kabc_status_t kabc_prefetch_model(const kabc_model_t* model, int32_t families) {
    if (!model || !model->prior) {
        set_error("kabc_prefetch_model: NULL argument");
        return KABC_ERR_INVALID_ARG;
    }
    ModelUnit* u = nullptr;
    if (kabc_status_t st = model_unit_for(model->prior, model->D, model->cost.id, &u)) return st;
    if (!u || !u->spec || !u->async) return KABC_OK;  // (nothing to do ahead: prebuilt, generic or compiled already)
    for (const SpecReq& r : spec_requests(model, families, u->pk_mask))
        if (!is_init_family(r.family)) (void)unit_kernel(u, r.family, model->D, r.variant);
    return KABC_OK;
}

extern "C" kabc_status_t kabc_model_release(int32_t handle) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (ModelUnit* u : g_specs)
        if (u->handle == handle && handle != 0 && !u->released) {
            u->released = true;  // (handles that were created on it keep their kernels)
            return KABC_OK;
        }
    set_error("kabc_model_release: %d is not a registered model", handle);
    return KABC_ERR_INVALID_ARG;
}

extern "C" int32_t kabc_rtc_cache_dir(char* out, int32_t cap) {
    const std::string d = rtc_cache_dir();
    if (out && cap > 0) std::snprintf(out, (size_t)cap, "%s", d.c_str());
    return (int32_t)d.size();
}

extern "C" kabc_status_t kabc_set_specialize(int32_t mode) {
    if (mode < KABC_SPECIALIZE_ENV || mode > KABC_SPECIALIZE_BACKGROUND) {
        set_error("kabc_set_specialize: mode is KABC_SPECIALIZE_ENV (-1), _OFF (0), _BLOCKING (1) or _BACKGROUND (2)");
        return KABC_ERR_INVALID_ARG;
    }
    g_spec_mode.store(mode);
    return KABC_OK;
}

extern "C" kabc_status_t kabc_spec_counters(uint64_t out[4]) {
    if (!out) {
        set_error("kabc_spec_counters: NULL argument");
        return KABC_ERR_INVALID_ARG;
    }
    out[0] = g_spec_counters.spawned.load();
    out[1] = g_spec_counters.loaded.load();
    out[2] = g_spec_counters.failed.load();
    out[3] = g_spec_counters.cache_hits.load();
    return KABC_OK;
}

// The worker process's whole job (csrc/rtc_worker.c calls this after it has detached): read the
// job file spawn_worker wrote, compile, store the code object under the cache path the parent
// polls -- or <cache path>.err with the compiler's message.  No HIP call is made.
extern "C" int32_t kabc_rtc_worker_main(const char* jobfile) {
    if (!jobfile) return 2;
    FILE* f = std::fopen(jobfile, "rb");
    if (!f) return 2;
    auto line = [f](std::string* out) {
        out->clear();
        int c;
        while ((c = std::fgetc(f)) != EOF && c != '\n') out->push_back((char)c);
        return c != EOF || !out->empty();
    };
    std::string magic, cpath, n, dg;
    std::vector<std::string> opt, names;
    std::string text;
    bool ok = line(&magic) && magic == "KABCJOB2" && line(&cpath) && line(&dg) && line(&n);
    const bool have_path = ok;  // (from here on a failure can be reported where the parent looks)
    for (long i = 0, m = ok ? std::atol(n.c_str()) : 0; i < m && ok; ++i) {
        std::string o;
        ok = line(&o);
        opt.push_back(o);
    }
    ok = ok && line(&n);
    for (long i = 0, m = ok ? std::atol(n.c_str()) : 0; i < m && ok; ++i) {
        std::string o;
        ok = line(&o);
        names.push_back(o);
    }
    ok = ok && line(&n);
    if (ok) {
        text.resize((size_t)std::atoll(n.c_str()));
        ok = text.empty() || std::fread(&text[0], 1, text.size(), f) == text.size();
    }
    std::fclose(f);
    int rc = 0;
    // every exit that delivers no code object leaves <cache path>.err (what the waiting handles poll
    // for -- without it they would look every 2 ms for 900 s) and takes the lock away
    auto fail = [&cpath](const char* msg) {
        const std::string tmp = cpath + ".err." + std::to_string((long long)getpid()) + ".tmp";
        (void)unlink(tmp.c_str());
        if (FILE* e = open_cache_file_new(tmp)) {
            std::fwrite(msg, 1, std::strlen(msg), e);
            std::fclose(e);
            if (std::rename(tmp.c_str(), (cpath + ".err").c_str()) != 0) (void)unlink(tmp.c_str());
        }
    };
    if (!ok) {
        rc = 2;
        if (have_path) fail("the compilation worker could not read its job file");
    } else {
        std::vector<char> code;
        std::vector<std::string> lowered;
        const size_t sl = cpath.rfind('/');
        const std::string cdir = sl == std::string::npos ? std::string(".") : cpath.substr(0, sl);
        if (rtc_compile_text(text, opt, names, &code, &lowered) == KABC_OK && !code.empty()) {
            if (!cache_store(cdir, cpath, std::strtoull(dg.c_str(), nullptr, 10), code, lowered)) {
                rc = 3;
                fail("the compilation worker could not store the code object (cache directory full, gone or not trusted)");
            }
        } else {
            rc = 1;
            fail(get_error());
        }
    }
    if (have_path) (void)unlink((cpath + ".lock").c_str());
    (void)unlink(jobfile);
    return rc;
}
