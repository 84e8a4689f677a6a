// capi_common.hip -- context, error plumbing, prior preparation and the small
// Factored utility kernels (logpdf / rand) of the C ABI (include/kabc.h).
#include <cstddef>
#include <map>
#include <mutex>
#include <vector>

#include "ais_kernels.hpp"
#include "host_common.hpp"
#include "plugin_registry.hpp"
#include "prior_util_kernels.hpp"

namespace kabc {

static thread_local char g_err[768];

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
const char* get_error() { return g_err; }

static double std_normal_cdf(double z) { return 0.5 * std::erfc(-z * M_SQRT1_2); }

// ---- MvNormal(mu, Sigma) priors (include/kabc_mvnormal.h) ---------------------------
// Registered once per process; the prepared block is copied to a device at first use there.
struct MvnEntry {
    int D = 0;
    std::vector<double> host;
    std::map<int, double*> dev;  // device id -> block
};
static std::mutex g_mvn_mu;
static std::vector<MvnEntry*> g_mvn;

kabc_status_t resolve_priors(kabc_ctx_t* ctx, const kabc_prior_t* prior, int D, kabc_prior_t* out) {
    if (!ctx || !prior || D < 1 || D > KABC_MAX_DIM_DYN) {
        set_error("invalid prior: NULL or D outside 1..%d", KABC_MAX_DIM_DYN);
        return KABC_ERR_INVALID_ARG;
    }
    int nmv = 0, njoint = 0;
    for (int k = 0; k < D; ++k) {
        out[k] = prior[k];
        nmv += prior[k].kind == KABC_PRIOR_MVNORMAL;
        njoint += (prior[k].kind >= KABC_PRIOR_USER && user_prior_is_joint(prior[k].kind)) ? 1 : 0;
    }
    if (njoint) {
        // a joint user prior (kabc_compile_mvprior_plugin) is all D components or none, one kind; p[3] is
        // the library's: D, for the sampler that is handed one component at a time (kabc_sampling.h)
        bool same = njoint == D;
        for (int k = 1; k < D && same; ++k) same = prior[k].kind == prior[0].kind;
        if (!same) {
            set_error("a joint user prior is all D components of the prior, every one with the kind "
                      "kabc_compile_mvprior_plugin returned; it does not mix with other components");
            return KABC_ERR_INVALID_ARG;
        }
        for (int k = 0; k < D; ++k) out[k].p[3] = (double)D;
        return KABC_OK;
    }
    if (nmv == 0) return KABC_OK;
    if (nmv != D || D > KABC_MAX_DIM) {
        set_error("an MvNormal prior is all D components of the prior (kind KABC_PRIOR_MVNORMAL, one handle, "
                  "p[1] = index), D <= %d; it does not mix with univariate components (the reference's "
                  "Factored takes univariate distributions only, src/priors.jl:10-13)", KABC_MAX_DIM);
        return nmv != D ? KABC_ERR_INVALID_ARG : KABC_ERR_UNSUPPORTED;
    }
    const int h = (int)prior[0].p[0];
    std::lock_guard<std::mutex> lk(g_mvn_mu);
    if (h < 1 || h > (int)g_mvn.size() || g_mvn[h - 1]->D != D) {
        set_error("MvNormal prior: handle %d is not a registered %d-dimensional MvNormal "
                  "(kabc_mvnormal_register)", h, D);
        return KABC_ERR_INVALID_ARG;
    }
    MvnEntry* e = g_mvn[h - 1];
    for (int k = 0; k < D; ++k)
        if ((int)prior[k].p[0] != h || prior[k].p[1] != (double)k) {
            set_error("MvNormal prior: component %d must carry p = (handle %d, %d)", k + 1, h, k);
            return KABC_ERR_INVALID_ARG;
        }
    double* d = nullptr;
    auto it = e->dev.find(ctx->device);
    if (it != e->dev.end()) {
        d = it->second;
    } else {
        KABC_HIP_CHECK(hipSetDevice(ctx->device));
        KABC_HIP_CHECK(hipMalloc(&d, sizeof(double) * e->host.size()));
        KABC_HIP_CHECK(hipMemcpy(d, e->host.data(), sizeof(double) * e->host.size(), hipMemcpyHostToDevice));
        KABC_HIP_CHECK(hipDeviceSynchronize());  // (once per handle and device: the contexts' streams are non-blocking)
        e->dev[ctx->device] = d;
    }
    for (int k = 0; k < D; ++k) {
        out[k].p[2] = kabc_mvn_ptr_to_double(d);
        out[k].p[3] = (double)D;
    }
    return KABC_OK;
}

bool prepare_prior(const kabc_prior_t& pr, PriorDev& q) {
    q.kind = pr.kind;
    q.discrete = kabc_prior_is_discrete(pr.kind);
    for (int j = 0; j < 4; ++j) q.p[j] = pr.p[j];
    q.c0 = q.c1 = 0.0;
    const double a = pr.p[0], b = pr.p[1];
    q.rb = 1.0 / ((pr.kind == KABC_PRIOR_EXPONENTIAL) ? a : b);
    switch (pr.kind) {
        case KABC_PRIOR_MVNORMAL:  // a RESOLVED component (resolve_priors): p[2] = the device block
            q.rb = 0.0;
            return kabc_bits(pr.p[2]) != 0 && pr.p[3] >= 1.0;
        case KABC_PRIOR_USER_INIT:  // no density, no parameters (drawn by the cost plugin)
            q.rb = 0.0;
            return true;
        case KABC_PRIOR_UNIFORM:
            if (!(b > a)) return false;
            q.c0 = -kabc_log(b - a);
            return true;
        case KABC_PRIOR_NORMAL:
        case KABC_PRIOR_LOGNORMAL:
            if (!(b > 0)) return false;
            q.c0 = kabc_log(b);
            return true;
        case KABC_PRIOR_TRUNCNORMAL: {
            if (!(b > 0) || !(pr.p[3] > pr.p[2])) return false;
            q.c0 = kabc_log(b);
            const double zl = (pr.p[2] - a) / b, zh = (pr.p[3] - a) / b;
            const double tp = (zl > 0) ? std_normal_cdf(-zl) - std_normal_cdf(-zh)
                                       : std_normal_cdf(zh) - std_normal_cdf(zl);
            q.c1 = std::log(tp);
            return true;
        }
        case KABC_PRIOR_BETA:
            if (!(a > 0) || !(b > 0)) return false;
            q.c0 = std::lgamma(a) + std::lgamma(b) - std::lgamma(a + b);
            return true;
        case KABC_PRIOR_DISCRETE_UNIFORM:
            if (!(b >= a) || a != kabc_rint(a) || b != kabc_rint(b)) return false;
            q.c0 = -kabc_log(b - a + 1.0);
            return true;
        case KABC_PRIOR_NEGBINOMIAL:
            if (!(a > 0) || !(b > 0) || !(b <= 1)) return false;
            q.c0 = a * kabc_log(b) - std::lgamma(a);
            q.c1 = kabc_log1p(-b);
            return true;
        case KABC_PRIOR_EXPONENTIAL:
            if (!(a > 0)) return false;
            q.c0 = kabc_log(a);
            return true;
        case KABC_PRIOR_GAMMA:
            if (!(a > 0) || !(b > 0)) return false;
            q.c0 = std::lgamma(a) + a * kabc_log(b);
            return true;
        default: {
            // a user family (kabc_compile_prior_plugin): parameters as given, no derived constants
            int disc = 0;
            if (pr.kind < KABC_PRIOR_USER || !user_prior_info(pr.kind, &disc)) return false;
            q.discrete = disc;
            q.rb = 0.0;
            return true;
        }
    }
}

bool prepare_priors(const kabc_prior_t* prior, int D, PriorSet& out) {
    if (!prior || D < 1 || D > KABC_MAX_DIM) return false;
    std::memset(&out, 0, sizeof out);
    for (int k = 0; k < D; ++k)
        if (!prepare_prior(prior[k], out.c[k])) return false;
    return true;
}

}  // namespace kabc

using namespace kabc;

namespace kabc {
// arithmetic-contract probe: one kabc_math.h function per launch (include/kabc.h)
__global__ void math_probe_kernel(int fn, int64_t n, const double* __restrict__ x,
                                  double* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        switch (fn) {
            case 0: out[i] = kabc_log(x[i]); break;
            case 1: out[i] = kabc_exp(x[i]); break;
            case 2: out[i] = kabc_log1p(x[i]); break;
            case 3: out[i] = kabc_lgamma(x[i]); break;
            case 4: {
                double s, c;
                kabc_sincos2pi(x[i], &s, &c);
                out[2 * i] = s;
                out[2 * i + 1] = c;
            } break;
            case 5: out[i] = kabc_sqrt(x[i]); break;
            case 6: out[i] = kabc_rint(x[i]); break;
            case 7: out[i] = kabc_log_pn(x[i]); break;
            case 8: out[i] = kabc_sqrt_pn(x[i]); break;
            case 9: out[i] = kabc_u01(kabc_bits(x[i])); break;
            case 10:
                kabc_normal_pair(kabc_bits(x[2 * i]), kabc_bits(x[2 * i + 1]), &out[2 * i],
                                 &out[2 * i + 1]);
                break;
            case 12: out[i] = kabc_exp_bounded(x[i]); break;
            default:
                out[i] = (double)kabc_index32(kabc_bits(x[2 * i]), (uint32_t)x[2 * i + 1]);
        }
    }
}
}  // namespace kabc

extern "C" {

int32_t kabc_version(void) { return KABC_VERSION; }

int32_t kabc_abi_sizeof(int32_t which) {
    static const int32_t sz[] = {
        (int32_t)sizeof(kabc_prior_t),        (int32_t)sizeof(kabc_cost_t),
        (int32_t)sizeof(kabc_model_t),        (int32_t)sizeof(kabc_stats_t),
        (int32_t)sizeof(kabc_smc_opts_t),     (int32_t)sizeof(kabc_smc_iter_t),
        (int32_t)sizeof(kabc_smc_result_t),   (int32_t)sizeof(kabc_abcde_opts_t),
        (int32_t)sizeof(kabc_abcde_result_t), (int32_t)sizeof(kabc_pfilter_opts_t),
        (int32_t)sizeof(kabc_pfilter_result_t)};
    return (which >= 0 && which < (int32_t)(sizeof sz / sizeof sz[0])) ? sz[which] : -1;
}
int32_t kabc_abi_offsetof(int32_t which, int32_t field) {
#define KABC_OFF(T, f) (int32_t)offsetof(T, f)
    static const std::vector<std::vector<int32_t>> off = {
        {KABC_OFF(kabc_prior_t, kind), KABC_OFF(kabc_prior_t, reserved), KABC_OFF(kabc_prior_t, p)},
        {KABC_OFF(kabc_cost_t, id), KABC_OFF(kabc_cost_t, nparams), KABC_OFF(kabc_cost_t, params),
         KABC_OFF(kabc_cost_t, ndata), KABC_OFF(kabc_cost_t, data)},
        {KABC_OFF(kabc_model_t, prior), KABC_OFF(kabc_model_t, D), KABC_OFF(kabc_model_t, posterior),
         KABC_OFF(kabc_model_t, eps), KABC_OFF(kabc_model_t, cost)},
        {KABC_OFF(kabc_stats_t, proposals), KABC_OFF(kabc_stats_t, cost_evals), KABC_OFF(kabc_stats_t, accepted)},
        {KABC_OFF(kabc_smc_opts_t, nparticles), KABC_OFF(kabc_smc_opts_t, alpha), KABC_OFF(kabc_smc_opts_t, mcmc_retrys),
         KABC_OFF(kabc_smc_opts_t, verbose), KABC_OFF(kabc_smc_opts_t, mcmc_tol), KABC_OFF(kabc_smc_opts_t, epstol),
         KABC_OFF(kabc_smc_opts_t, r_epstol), KABC_OFF(kabc_smc_opts_t, min_r_ess), KABC_OFF(kabc_smc_opts_t, max_stretch),
         KABC_OFF(kabc_smc_opts_t, seed), KABC_OFF(kabc_smc_opts_t, max_iterations)},
        {KABC_OFF(kabc_smc_iter_t, eps), KABC_OFF(kabc_smc_iter_t, ess), KABC_OFF(kabc_smc_iter_t, accepted),
         KABC_OFF(kabc_smc_iter_t, resampled), KABC_OFF(kabc_smc_iter_t, flag), KABC_OFF(kabc_smc_iter_t, mcmc_passes),
         KABC_OFF(kabc_smc_iter_t, reserved)},
        {KABC_OFF(kabc_smc_result_t, theta), KABC_OFF(kabc_smc_result_t, cost), KABC_OFF(kabc_smc_result_t, alive),
         KABC_OFF(kabc_smc_result_t, eps), KABC_OFF(kabc_smc_result_t, iterations), KABC_OFF(kabc_smc_result_t, n_alive),
         KABC_OFF(kabc_smc_result_t, cost_evals), KABC_OFF(kabc_smc_result_t, proposals),
         KABC_OFF(kabc_smc_result_t, iter_log), KABC_OFF(kabc_smc_result_t, iter_log_cap),
         KABC_OFF(kabc_smc_result_t, kernel_ms_mcmc), KABC_OFF(kabc_smc_result_t, mcmc_launches)},
        {KABC_OFF(kabc_abcde_opts_t, nparticles), KABC_OFF(kabc_abcde_opts_t, generations),
         KABC_OFF(kabc_abcde_opts_t, eps_target), KABC_OFF(kabc_abcde_opts_t, alpha),
         KABC_OFF(kabc_abcde_opts_t, proposal_width), KABC_OFF(kabc_abcde_opts_t, earlystop),
         KABC_OFF(kabc_abcde_opts_t, verbose), KABC_OFF(kabc_abcde_opts_t, seed)},
        {KABC_OFF(kabc_abcde_result_t, theta), KABC_OFF(kabc_abcde_result_t, cost), KABC_OFF(kabc_abcde_result_t, reached_eps),
         KABC_OFF(kabc_abcde_result_t, reserved), KABC_OFF(kabc_abcde_result_t, generations_run),
         KABC_OFF(kabc_abcde_result_t, nsims)},
        {KABC_OFF(kabc_pfilter_opts_t, nparticles), KABC_OFF(kabc_pfilter_opts_t, q), KABC_OFF(kabc_pfilter_opts_t, eff_tol),
         KABC_OFF(kabc_pfilter_opts_t, epstol), KABC_OFF(kabc_pfilter_opts_t, proposal_width),
         KABC_OFF(kabc_pfilter_opts_t, max_iters), KABC_OFF(kabc_pfilter_opts_t, verbose),
         KABC_OFF(kabc_pfilter_opts_t, reserved), KABC_OFF(kabc_pfilter_opts_t, seed)},
        {KABC_OFF(kabc_pfilter_result_t, theta), KABC_OFF(kabc_pfilter_result_t, cost), KABC_OFF(kabc_pfilter_result_t, eps),
         KABC_OFF(kabc_pfilter_result_t, eff), KABC_OFF(kabc_pfilter_result_t, iterations),
         KABC_OFF(kabc_pfilter_result_t, nreps), KABC_OFF(kabc_pfilter_result_t, cost_evals)}};
#undef KABC_OFF
    if (which < 0 || which >= (int32_t)off.size()) return -1;
    const std::vector<int32_t>& f = off[(size_t)which];
    return (field >= 0 && field < (int32_t)f.size()) ? f[(size_t)field] : -1;
}
const char* kabc_last_error(void) { return get_error(); }

int32_t kabc_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

kabc_status_t kabc_ctx_create(int32_t device_id, void* stream, kabc_ctx_t** out) {
    if (!out) {
        set_error("kabc_ctx_create: out is NULL");
        return KABC_ERR_INVALID_ARG;
    }
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        set_error("kabc_ctx_create: no HIP device visible (the gfx950 kernels cannot run; "
                  "there is no CPU fallback)");
        return KABC_ERR_DEVICE;
    }
    if (device_id < 0 || device_id >= n) {
        set_error("kabc_ctx_create: device %d out of range [0,%d)", device_id, n);
        return KABC_ERR_INVALID_ARG;
    }
    KABC_HIP_CHECK(hipSetDevice(device_id));
    kabc_ctx_t* c = new kabc_ctx_t();
    c->device = device_id;
    c->own_stream = (stream == nullptr);
    if (stream) {
        c->stream = (hipStream_t)stream;
    } else {
        hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            delete c;
            set_error("hipStreamCreate failed: %s", hipGetErrorString(e));
            return KABC_ERR_DEVICE;
        }
    }
    *out = c;
    return KABC_OK;
}

kabc_status_t kabc_ctx_destroy(kabc_ctx_t* ctx) {
    if (!ctx) return KABC_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto& e : ctx->pool) (void)hipFree(e.second);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return KABC_OK;
}

kabc_status_t kabc_ctx_synchronize(kabc_ctx_t* ctx) {
    if (!ctx) {
        set_error("ctx is NULL");
        return KABC_ERR_INVALID_ARG;
    }
    KABC_HIP_CHECK(hipSetDevice(ctx->device));
    KABC_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return KABC_OK;
}

kabc_status_t kabc_host_alloc(size_t bytes, void** out) {
    if (!out || bytes == 0) {
        set_error("kabc_host_alloc: NULL out pointer or zero size");
        return KABC_ERR_INVALID_ARG;
    }
    *out = nullptr;
    KABC_HIP_CHECK(hipHostMalloc(out, bytes, hipHostMallocPortable));
    return KABC_OK;
}

kabc_status_t kabc_host_free(void* p) {
    if (p) KABC_HIP_CHECK(hipHostFree(p));
    return KABC_OK;
}

static kabc_status_t prior_util(kabc_ctx_t* ctx, const kabc_prior_t* prior, int32_t D, int64_t n,
                                const double* x, double* out, int mode, uint64_t seed,
                                uint64_t attempt, uint32_t first_walker, uint32_t domain) {
    if (!ctx || !prior || !out || n < 0) {
        set_error("invalid argument");
        return KABC_ERR_INVALID_ARG;
    }
    PriorUtilArgs A;
    std::memset(&A, 0, sizeof A);
    if (D < 1 || D > KABC_MAX_DIM_DYN) {
        set_error("invalid prior: D outside 1..%d", KABC_MAX_DIM_DYN);
        return KABC_ERR_INVALID_ARG;
    }
    std::vector<kabc_prior_t> rp((size_t)D);
    if (kabc_status_t st = resolve_priors(ctx, prior, D, rp.data())) return st;
    prior = rp.data();
    std::vector<PriorDev> prep((size_t)D);
    for (int k = 0; k < D; ++k)
        if (!prepare_prior(prior[k], prep[k])) {
            set_error("invalid prior (kind/parameters) of component %d", k + 1);
            return KABC_ERR_INVALID_ARG;
        }
    if (n == 0) return KABC_OK;
    KABC_HIP_CHECK(hipSetDevice(ctx->device));
    // (cost id 0: the utility kernels contain no cost; only user FAMILIES make a unit necessary --
    // a specialisation registered for these components is for the samplers' kernels, not these)
    ModelUnit* unit = nullptr;
    bool has_user = false;
    for (int k = 0; k < D; ++k) has_user = has_user || prior[k].kind >= KABC_PRIOR_USER;
    if (has_user) {
        if (kabc_status_t st = model_unit_for(prior, D, 0, &unit, false)) return st;
    }
    PriorDev* d_prep = nullptr;
    kabc_prior_t* d_raw = nullptr;
    KABC_HIP_CHECK(hipMalloc(&d_prep, sizeof(PriorDev) * D));
    KABC_HIP_CHECK(hipMalloc(&d_raw, sizeof(kabc_prior_t) * D));
    KABC_HIP_CHECK(hipMemcpyAsync(d_prep, prep.data(), sizeof(PriorDev) * D, hipMemcpyHostToDevice, ctx->stream));
    KABC_HIP_CHECK(hipMemcpyAsync(d_raw, prior, sizeof(kabc_prior_t) * D, hipMemcpyHostToDevice, ctx->stream));
    A.prior = d_prep;
    A.raw = d_raw;
    const size_t in_bytes = sizeof(double) * n * D;
    const size_t out_bytes = sizeof(double) * n * ((mode == 0) ? 1 : D);
    double *dx = nullptr, *dout = nullptr;
    if (mode != 2) {
        KABC_HIP_CHECK(hipMalloc(&dx, in_bytes));
        KABC_HIP_CHECK(hipMemcpyAsync(dx, x, in_bytes, hipMemcpyHostToDevice, ctx->stream));
    }
    KABC_HIP_CHECK(hipMalloc(&dout, out_bytes));
    A.x = dx;
    A.out = dout;
    A.n = n;
    A.D = D;
    A.mode = mode;
    A.seed = seed;
    A.attempt = attempt;
    A.first_walker = first_walker;
    A.domain = domain;
    const unsigned grid = (unsigned)((n + 255) / 256);
    if (unit) {  // user families among the components: the kernels compiled with their snippets
        const PluginKernel k = unit_kernel(unit, mode == 2 ? kPfPriorRand : kPfPriorLogpdf, 1, 0);
        if (!k.mod) return KABC_ERR_DEVICE;  // (message set by the compilation / load)
        KABC_HIP_CHECK(rtc_launch(k.mod, dim3(grid), dim3(256), &A, ctx->stream));
    } else if (mode == 2) {
        hipLaunchKernelGGL(prior_rand_kernel, dim3(grid), dim3(256), 0, ctx->stream, A);
    } else {
        hipLaunchKernelGGL(prior_logpdf_kernel, dim3(grid), dim3(256), 0, ctx->stream, A);
    }
    KABC_HIP_CHECK(hipGetLastError());
    KABC_HIP_CHECK(hipMemcpyAsync(out, dout, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    KABC_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (dx) (void)hipFree(dx);
    (void)hipFree(dout);
    (void)hipFree(d_prep);
    (void)hipFree(d_raw);
    return KABC_OK;
}

kabc_status_t kabc_mvnormal_register(const double* mu, const double* cov, int32_t D, int32_t* handle) {
    if (!mu || !cov || !handle || D < 1 || D > KABC_MAX_DIM) {
        set_error("kabc_mvnormal_register: NULL argument or D outside 1..%d", KABC_MAX_DIM);
        return KABC_ERR_INVALID_ARG;
    }
    for (int i = 0; i < D; ++i)
        if (!kabc_isfinite(mu[i])) {
            set_error("kabc_mvnormal_register: mu[%d] is not finite", i);
            return KABC_ERR_INVALID_ARG;
        }
    MvnEntry* e = new MvnEntry();
    e->D = D;
    e->host.resize((size_t)kabc_mvn_block_words(D));
    const int rc = kabc_mvn_prepare(D, mu, cov, e->host.data());
    if (rc) {
        delete e;
        set_error("kabc_mvnormal_register: Sigma is not %s", rc == 1 ? "symmetric" : "positive definite");
        return KABC_ERR_INVALID_ARG;
    }
    std::lock_guard<std::mutex> lk(g_mvn_mu);
    // the same (mu, Sigma) registered again -- priors built in a loop, a sweep's repeated models --
    // is the same entry: nothing accumulates on the host or on the devices
    for (size_t i = 0; i < g_mvn.size(); ++i)
        if (g_mvn[i]->D == D && g_mvn[i]->host.size() == e->host.size() &&
            std::memcmp(g_mvn[i]->host.data(), e->host.data(), sizeof(double) * e->host.size()) == 0) {
            delete e;
            *handle = (int32_t)(i + 1);
            return KABC_OK;
        }
    g_mvn.push_back(e);
    *handle = (int32_t)g_mvn.size();
    return KABC_OK;
}

kabc_status_t kabc_factored_logpdf(kabc_ctx_t* ctx, const kabc_prior_t* prior, int32_t D,
                                   int64_t n, const double* x, double* out) {
    return prior_util(ctx, prior, D, n, x, out, 0, 0, 0, 0, 0);
}
kabc_status_t kabc_factored_push_p(kabc_ctx_t* ctx, const kabc_prior_t* prior, int32_t D,
                                   int64_t n, const double* x, double* out) {
    return prior_util(ctx, prior, D, n, x, out, 1, 0, 0, 0, 0);
}
kabc_status_t kabc_factored_rand(kabc_ctx_t* ctx, const kabc_prior_t* prior, int32_t D,
                                 uint64_t seed, uint32_t domain, int64_t first_walker, int64_t n,
                                 uint64_t attempt, double* out) {
    return prior_util(ctx, prior, D, n, nullptr, out, 2, seed, attempt, (uint32_t)first_walker,
                      domain);
}

kabc_status_t kabc_math_probe(kabc_ctx_t* ctx, int32_t fn, int64_t n, const double* x,
                              double* out) {
    if (!ctx || !x || !out || n < 0 || fn < 0 || fn > 12) {
        set_error("kabc_math_probe: bad argument");
        return KABC_ERR_INVALID_ARG;
    }
    if (n == 0) return KABC_OK;
    KABC_HIP_CHECK(hipSetDevice(ctx->device));
    const int in_w = (fn == 10 || fn == 11) ? 2 : 1, out_w = (fn == 4 || fn == 10) ? 2 : 1;
    double *dx = nullptr, *dout = nullptr;
    KABC_HIP_CHECK(hipMalloc(&dx, sizeof(double) * n * in_w));
    KABC_HIP_CHECK(hipMalloc(&dout, sizeof(double) * n * out_w));
    KABC_HIP_CHECK(hipMemcpyAsync(dx, x, sizeof(double) * n * in_w, hipMemcpyHostToDevice,
                                  ctx->stream));
    const int blocks = (int)((n + 255) / 256 < 65535 ? (n + 255) / 256 : 65535);
    hipLaunchKernelGGL(kabc::math_probe_kernel, dim3(blocks), dim3(256), 0, ctx->stream, fn, n, dx,
                       dout);
    KABC_HIP_CHECK(hipGetLastError());
    KABC_HIP_CHECK(hipMemcpyAsync(out, dout, sizeof(double) * n * out_w, hipMemcpyDeviceToHost,
                                  ctx->stream));
    KABC_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    KABC_HIP_CHECK(hipFree(dx));
    KABC_HIP_CHECK(hipFree(dout));
    return KABC_OK;
}

}  // extern "C"
