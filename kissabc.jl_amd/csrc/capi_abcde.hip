// capi_abcde.hip -- kabc_abcde_run: ABCDE(prior, cost, ϵ_target; ...) of
// src/smc.jl:347-430.  All generations are enqueued without a host round trip; the
// earlystop break is taken on the device (kernels after it are no-ops).
#include <cmath>
#include <vector>

#include <hipcub/hipcub.hpp>

#define KABC_ABCDE_SINGLE_UNIT 1
#include "abcde_kernels.hpp"
#include "host_common.hpp"
#include "plugin_registry.hpp"

namespace kabc {

template <int D>
static void l_init(const AbcdeArgs& a, hipStream_t s) {
    hipLaunchKernelGGL((abcde_init_kernel<D>), dim3((unsigned)((a.N + kAbcdeBlock - 1) / kAbcdeBlock)),
                       dim3(kAbcdeBlock), 0, s, a);
}
template <int D>
static void l_gen(const AbcdeArgs& a, hipStream_t s) {
    hipLaunchKernelGGL((abcde_gen_kernel<D>), dim3((unsigned)((a.N + kAbcdeBlock - 1) / kAbcdeBlock)),
                       dim3(kAbcdeBlock), 0, s, a);
}
template <int... Ds>
static AbcdeLaunchFn pick_init(int D, std::integer_sequence<int, Ds...>) {
    static const AbcdeLaunchFn f[] = {&l_init<Ds + 1>...};
    return f[D - 1];
}
template <int... Ds>
static AbcdeLaunchFn pick_gen(int D, std::integer_sequence<int, Ds...>) {
    static const AbcdeLaunchFn f[] = {&l_gen<Ds + 1>...};
    return f[D - 1];
}

// ---- rank structure (the donor draw was O(N) per particle) ------------------------
constexpr int64_t kRankMinN = 4096;  // below this the two scans are cheaper than building it

__global__ void __launch_bounds__(256) wm_iota_kernel(unsigned* v, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) v[i] = (unsigned)i;
}
// bit b of every element of the sequence entering level b, packed 64 per word
__global__ void __launch_bounds__(256) wm_bits_kernel(const unsigned* seq, int64_t n, int b,
                                                      unsigned long long* bits, int64_t words) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool one = p < n && ((seq[p] >> b) & 1u);
    const unsigned long long m = __ballot(one);
    if ((threadIdx.x & 63) == 0 && (p >> 6) < words) bits[p >> 6] = m;
}
// exclusive prefix of the words' popcounts (ones before each word) and the level's zero count
__global__ void __launch_bounds__(1024) wm_count_kernel(const unsigned long long* bits, int64_t words,
                                                        int64_t n, unsigned* cnt, unsigned* nz) {
    __shared__ unsigned s_w[16];
    __shared__ unsigned s_run;
    if (threadIdx.x == 0) s_run = 0;
    __syncthreads();
    for (int64_t w0 = 0; w0 < words; w0 += 1024) {
        const int64_t w = w0 + threadIdx.x;
        const unsigned c = w < words ? (unsigned)__popcll(bits[w]) : 0u;
        const unsigned incl = wave_scan_incl(c);
        if ((threadIdx.x & 63) == 63) s_w[threadIdx.x >> 6] = incl;
        __syncthreads();
        unsigned off = s_run, tot = 0;
        for (int q = 0; q < 16; ++q) {
            if (q < (int)(threadIdx.x >> 6)) off += s_w[q];
            tot += s_w[q];
        }
        if (w < words) cnt[w] = off + incl - c;
        __syncthreads();
        if (threadIdx.x == 0) s_run += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) *nz = (unsigned)n - s_run;
}

struct RankStructure {
    double* sorted = nullptr;
    unsigned *iota = nullptr, *seq[2] = {nullptr, nullptr}, *cnt = nullptr, *nz = nullptr;
    unsigned long long* bits = nullptr;
    void* tmp = nullptr;
    size_t tmp_bytes = 0;
    int levels = 0;
    int64_t words = 0;
};

// sorted costs + wavelet matrix of the cost-sorted particle order, all on stream s
static hipError_t build_rank(const RankStructure& R, const double* delta, int64_t N, hipStream_t s) {
    size_t tb = R.tmp_bytes;
    hipError_t e = hipcub::DeviceRadixSort::SortPairs(R.tmp, tb, delta, R.sorted, R.iota, R.seq[0],
                                                      (int)N, 0, 64, s);
    int cur = 0;
    const unsigned g256 = (unsigned)((R.words * 64 + 255) / 256);
    for (int b = R.levels - 1; b >= 0 && e == hipSuccess; --b) {
        unsigned long long* bits = R.bits + (size_t)b * R.words;
        hipLaunchKernelGGL(wm_bits_kernel, dim3(g256), dim3(256), 0, s, R.seq[cur], N, b, bits, R.words);
        hipLaunchKernelGGL(wm_count_kernel, dim3(1), dim3(1024), 0, s, bits, R.words, N,
                           R.cnt + (size_t)b * R.words, R.nz + b);
        if (b > 0) {  // stable partition by bit b: the sequence entering the next level
            tb = R.tmp_bytes;
            e = hipcub::DeviceRadixSort::SortKeys(R.tmp, tb, R.seq[cur], R.seq[1 - cur], (int)N, b, b + 1, s);
            cur ^= 1;
        }
    }
    return e == hipSuccess ? hipGetLastError() : e;
}

}  // namespace kabc

using namespace kabc;

extern "C" {

void kabc_abcde_default_opts(kabc_abcde_opts_t* o) {
    if (!o) return;
    o->nparticles = 50;
    o->generations = 20;
    o->eps_target = 0.0;
    o->alpha = 0.0;
    o->proposal_width = 1.0;
    o->earlystop = 0;
    o->verbose = 0;
    o->seed = 0;
}

kabc_status_t kabc_abcde_run(kabc_ctx_t* ctx, const kabc_prior_t* prior, int32_t D,
                             const kabc_cost_t* cost, const kabc_abcde_opts_t* o,
                             kabc_abcde_result_t* res) {
    if (!ctx || !prior || !cost || !o || !res) {
        set_error("kabc_abcde_run: NULL argument");
        return KABC_ERR_INVALID_ARG;
    }
    if (!(o->alpha >= 0 && o->alpha < 1)) {  // @assert 0<=α<1 (:348)
        set_error("α must be in 0 <= α < 1.");
        return KABC_ERR_INVALID_ARG;
    }
    const int64_t N = o->nparticles;
    if (D < 1 || D > KABC_MAX_DIM_DYN) {
        set_error("length(prior) = %d is outside the device path's range 1..%d", D, KABC_MAX_DIM_DYN);
        return KABC_ERR_UNSUPPORTED;
    }
    // length(prior) > KABC_MAX_DIM: the run-time-dimension instantiation (D = 0) of the kernels,
    // prior components as device arrays (built-in DeviceCosts)
    const bool dyn = D > KABC_MAX_DIM;
    if (dyn && cost->id >= KABC_COST_USER) {
        set_error("ABCDE with length(prior) = %d > %d: built-in DeviceCosts only", D, KABC_MAX_DIM);
        return KABC_ERR_UNSUPPORTED;
    }
    if (N < 3 || N >= (1ll << 31)) {  // three distinct indices s, a, b are drawn (:394-401)
        set_error("nparticles must be >= 3 (and < 2^31)");
        return KABC_ERR_INVALID_ARG;
    }
    std::vector<kabc_prior_t> resolved((size_t)D);  // MvNormal components: device block, D
    if (kabc_status_t st = resolve_priors(ctx, prior, D, resolved.data())) return st;
    prior = resolved.data();
    AbcdeArgs A;
    std::memset(&A, 0, sizeof A);
    std::vector<PriorDev> Pdyn((size_t)(dyn ? D : 0));
    bool prior_ok = true;
    if (dyn)
        for (int k = 0; k < D && prior_ok; ++k) prior_ok = prepare_prior(prior[k], Pdyn[k]);
    else
        prior_ok = prepare_priors(prior, D, A.prior);
    if (!prior_ok) {
        set_error("invalid prior parameters");
        return KABC_ERR_INVALID_ARG;
    }
    if (!cost_dim_ok_rt(cost->id, D)) {
        set_error("DeviceCost id %d does not accept D = %d", cost->id, D);
        return KABC_ERR_UNSUPPORTED;
    }
    AbcdeLaunch f_init, f_gen;
    // (run-time compiled kernels are loaded on the CURRENT device)
    KABC_HIP_CHECK(hipSetDevice(ctx->device));
    ModelUnit* unit = nullptr;
    for (int k = 0; k < D && dyn; ++k)
        if (prior[k].kind >= KABC_PRIOR_USER) {
            set_error("a prior with user families supports length(prior) <= %d (got %d)", KABC_MAX_DIM, D);
            return KABC_ERR_UNSUPPORTED;
        }
    if (!dyn)
        if (kabc_status_t st = model_unit_for(prior, D, cost->id, &unit)) return st;
    if (unit) {  // user prior families / a specialised model (plugin_registry.hpp)
        const PluginKernel ki = unit_kernel(unit, kPfAbcdeInit, D, 0), kg = unit_kernel(unit, kPfAbcdeGen, D, 0);
        if (ki.mod && kg.mod) {
            f_init = AbcdeLaunch(ki.mod, &abcde_geom, (unsigned)kAbcdeBlock);
            f_gen = AbcdeLaunch(kg.mod, &abcde_geom, (unsigned)kAbcdeBlock);
        } else if (!unit_is_spec(unit)) {
            return KABC_ERR_DEVICE;
        } else {
            unit = nullptr;
        }
    }
    if (unit) {
    } else if (dyn) {
        f_init = AbcdeLaunch(&l_init<0>);
        f_gen = AbcdeLaunch(&l_gen<0>);
    } else if (const CostPlugin* p = find_plugin(cost->id)) {
        const PluginKernel ki = plugin_kernel(p, kPfAbcdeInit, D, 0), kg = plugin_kernel(p, kPfAbcdeGen, D, 0);
        f_init = ki.host ? AbcdeLaunch((AbcdeLaunchFn)ki.host)
                         : ki.mod ? AbcdeLaunch(ki.mod, &abcde_geom, (unsigned)kAbcdeBlock) : AbcdeLaunch();
        f_gen = kg.host ? AbcdeLaunch((AbcdeLaunchFn)kg.host)
                        : kg.mod ? AbcdeLaunch(kg.mod, &abcde_geom, (unsigned)kAbcdeBlock) : AbcdeLaunch();
        if (!f_init || !f_gen) {
            set_error("cost plugin has no ABCDE kernels for D = %d", D);
            return KABC_ERR_UNSUPPORTED;
        }
    } else {
        f_init = pick_init(D, std::make_integer_sequence<int, KABC_MAX_DIM>{});
        f_gen = pick_gen(D, std::make_integer_sequence<int, KABC_MAX_DIM>{});
    }
    if (!dyn) std::memcpy(A.raw, prior, sizeof(kabc_prior_t) * D);
    KABC_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    std::vector<void*> bufs;
    auto alloc = [&](void** p, size_t bytes) {
        hipError_t e = dev_malloc(p, bytes ? bytes : 8);
        if (e == hipSuccess) bufs.push_back(*p);
        return e;
    };
    struct Free {
        std::vector<void*>& b;
        ~Free() {
            for (void* p : b) (void)hipFree(p);
        }
    } freer{bufs};
    for (int b = 0; b < 2; ++b) {
        KABC_HIP_CHECK(alloc((void**)&A.theta[b], sizeof(double) * N * D));
        KABC_HIP_CHECK(alloc((void**)&A.delta[b], sizeof(double) * N));
        KABC_HIP_CHECK(alloc((void**)&A.lpi[b], sizeof(double) * N));
    }
    double *d_params = nullptr, *d_data = nullptr, *d_out = nullptr, *d_dout = nullptr;
    KABC_HIP_CHECK(alloc((void**)&A.ctrl, sizeof(AbcdeCtrl)));
    KABC_HIP_CHECK(alloc((void**)&d_out, sizeof(double) * N * D));
    KABC_HIP_CHECK(alloc((void**)&d_dout, sizeof(double) * N));
    KABC_HIP_CHECK(hipMemsetAsync(A.ctrl, 0, sizeof(AbcdeCtrl), s));
    PriorDev* d_prior = nullptr;
    kabc_prior_t* d_raw = nullptr;
    if (dyn) {
        KABC_HIP_CHECK(alloc((void**)&d_prior, sizeof(PriorDev) * D));
        KABC_HIP_CHECK(alloc((void**)&d_raw, sizeof(kabc_prior_t) * D));
        KABC_HIP_CHECK(hipMemcpyAsync(d_prior, Pdyn.data(), sizeof(PriorDev) * D, hipMemcpyHostToDevice, s));
        KABC_HIP_CHECK(hipMemcpyAsync(d_raw, prior, sizeof(kabc_prior_t) * D, hipMemcpyHostToDevice, s));
    }
    A.D_rt = D;
    A.dprior = d_prior;
    A.draw = d_raw;
    if (cost->nparams > 0) {
        KABC_HIP_CHECK(alloc((void**)&d_params, sizeof(double) * cost->nparams));
        KABC_HIP_CHECK(hipMemcpyAsync(d_params, cost->params, sizeof(double) * cost->nparams,
                                      hipMemcpyHostToDevice, s));
    }
    if (cost->ndata > 0) {
        KABC_HIP_CHECK(alloc((void**)&d_data, sizeof(double) * cost->ndata));
        KABC_HIP_CHECK(hipMemcpyAsync(d_data, cost->data, sizeof(double) * cost->ndata,
                                      hipMemcpyHostToDevice, s));
    }
    A.cost_params = d_params;
    A.cost_data = d_data;
    A.cost_ndata = cost->ndata;
    A.N = N;
    A.seed = o->seed;
    A.cost_id = cost->id;
    A.earlystop = o->earlystop;
    A.eps_target = o->eps_target;
    A.alpha = o->alpha;
    A.gamma = o->proposal_width * 2.38 / std::sqrt((double)(2 * D));
    A.dom_init = KABC_DOM_ABCDE_INIT;
    A.dom_init_cost = KABC_DOM_ABCDE_INIT_COST;
    f_init(A, s);
    KABC_HIP_CHECK(hipGetLastError());
    // large ensembles: a rank structure per generation replaces the O(N) donor scans
    RankStructure R;
    if (N >= kRankMinN) {
        R.levels = 1;
        while ((1ll << R.levels) < N) ++R.levels;
        R.words = (N + 64) / 64;  // one word past position N (rank queries at p = N)
        KABC_HIP_CHECK(alloc((void**)&R.sorted, sizeof(double) * N));
        KABC_HIP_CHECK(alloc((void**)&R.iota, sizeof(unsigned) * N));
        KABC_HIP_CHECK(alloc((void**)&R.seq[0], sizeof(unsigned) * N));
        KABC_HIP_CHECK(alloc((void**)&R.seq[1], sizeof(unsigned) * N));
        KABC_HIP_CHECK(alloc((void**)&R.bits, sizeof(unsigned long long) * R.levels * R.words));
        KABC_HIP_CHECK(alloc((void**)&R.cnt, sizeof(unsigned) * R.levels * R.words));
        KABC_HIP_CHECK(alloc((void**)&R.nz, sizeof(unsigned) * R.levels));
        size_t t1 = 0, t2 = 0;
        KABC_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, t1, (const double*)nullptr, (double*)nullptr,
                                                          (const unsigned*)nullptr, (unsigned*)nullptr, (int)N,
                                                          0, 64, s));
        KABC_HIP_CHECK(hipcub::DeviceRadixSort::SortKeys(nullptr, t2, (const unsigned*)nullptr,
                                                         (unsigned*)nullptr, (int)N, 0, 1, s));
        R.tmp_bytes = t1 > t2 ? t1 : t2;
        KABC_HIP_CHECK(alloc(&R.tmp, R.tmp_bytes));
        hipLaunchKernelGGL(wm_iota_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, R.iota, N);
        A.sorted_delta = R.sorted;
        A.wm_bits = R.bits;
        A.wm_cnt = R.cnt;
        A.wm_nz = R.nz;
        A.wm_levels = R.levels;
        A.wm_words = R.words;
    }
    for (int64_t g = 0; g < o->generations; ++g) {  // while iters < generations (:372)
        hipLaunchKernelGGL(abcde_extrema_kernel, dim3(1), dim3(1024), 0, s, A);
        // the buffer set flips once per generation until the earlystop break, after which every
        // kernel is a no-op: the host knows which one is current
        if (R.sorted) KABC_HIP_CHECK(build_rank(R, A.delta[g & 1], N, s));
        f_gen(A, s);
        hipLaunchKernelGGL(abcde_flip_kernel, dim3(1), dim3(1), 0, s, A.ctrl);
    }
    KABC_HIP_CHECK(hipGetLastError());
    AbcdeFinalArgs F;
    for (int b = 0; b < 2; ++b) {
        F.theta[b] = A.theta[b];
        F.delta[b] = A.delta[b];
    }
    F.ctrl = A.ctrl;
    F.out = d_out;
    F.dout = d_dout;
    F.N = N;
    F.D = D;
    F.prior = A.prior;
    F.dprior = d_prior;
    hipLaunchKernelGGL(abcde_final_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, F);
    KABC_HIP_CHECK(hipGetLastError());
    AbcdeCtrl hc;
    KABC_HIP_CHECK(hipMemcpyAsync(&hc, A.ctrl, sizeof hc, hipMemcpyDeviceToHost, s));
    if (res->theta)
        KABC_HIP_CHECK(hipMemcpyAsync(res->theta, d_out, sizeof(double) * N * D,
                                      hipMemcpyDeviceToHost, s));
    std::vector<double> hd((size_t)N);
    KABC_HIP_CHECK(hipMemcpyAsync(hd.data(), d_dout, sizeof(double) * N, hipMemcpyDeviceToHost, s));
    KABC_HIP_CHECK(hipStreamSynchronize(s));
    if (hc.error) {
        set_error("ABCDE: the prior never produced a finite (cost, logpdf) pair for some particle");
        return KABC_ERR_RETRY_EXHAUSTED;
    }
    double mx = -INFINITY;
    for (int64_t i = 0; i < N; ++i) {
        if (res->cost) res->cost[i] = hd[i];
        mx = hd[i] > mx ? hd[i] : mx;
    }
    res->reached_eps = (mx <= o->eps_target) ? 1 : 0;  // conv = maximum(Δs) <= ϵ_target
    res->reserved = 0;
    res->generations_run = hc.iters;
    res->nsims = hc.nsims;
    if (o->verbose)
        fprintf(stderr, "ABCDE End: converged = %d nsim = %llu range_eps = (%g, %g)\n",
                res->reached_eps, (unsigned long long)hc.nsims, hc.eps_l, hc.eps_h);
    return KABC_OK;
}

}  // extern "C"
