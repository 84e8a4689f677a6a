// capi_smc.hip -- kabc_smc_run: smc(prior, cost; kwargs...) of src/smc.jl:92-206
// driven from the host, one ε-iteration = one select kernel + 1..(1+mcmc_retrys)
// propose/accept kernels; the host reads one small control record per pass.
#include <cmath>
#include <cstdlib>
#include <vector>

#define KABC_SMC_SINGLE_UNIT 1
#include "ais_aux_kernels.hpp"
#include "host_common.hpp"
#include "plugin_registry.hpp"
#include "smc_loop_kernel.hpp"
#include "smc_small_kernel.hpp"
#include "smc_dyn_kernels.hpp"
#include "smc_dsel_kernels.hpp"

namespace kabc {
SmcDynLaunchFn find_smc_dyn_kernel(int cost_id);  // ais_dyn.hip
}
#include "abcde_kernels.hpp"
#include "pfilter_kernels.hpp"

namespace kabc {

#define KABC_DECL_COST(id) SmcLaunchFn find_smc_kernel_cost_##id(int D, bool simple);
KABC_DECL_COST(1)
KABC_DECL_COST(2)
KABC_DECL_COST(3)
KABC_DECL_COST(4)
KABC_DECL_COST(5)
KABC_DECL_COST(6)
KABC_DECL_COST(7)
KABC_DECL_COST(8)
KABC_DECL_COST(9)
KABC_DECL_COST(10)
KABC_DECL_COST(11)

#define KABC_DECL_LOOP(id) SmcLoopLaunchFn find_smc_loop_kernel_cost_##id(int D, bool simple);
KABC_DECL_LOOP(1)
KABC_DECL_LOOP(2)
KABC_DECL_LOOP(3)
KABC_DECL_LOOP(4)
KABC_DECL_LOOP(5)
KABC_DECL_LOOP(6)
KABC_DECL_LOOP(7)
KABC_DECL_LOOP(8)
KABC_DECL_LOOP(9)
KABC_DECL_LOOP(10)
KABC_DECL_LOOP(11)

#define KABC_DECL_SMALL(id) SmcSmallLaunchFn find_smc_small_kernel_cost_##id(int D, bool simple);
KABC_DECL_SMALL(1)
KABC_DECL_SMALL(2)
KABC_DECL_SMALL(3)
KABC_DECL_SMALL(4)
KABC_DECL_SMALL(5)
KABC_DECL_SMALL(6)
KABC_DECL_SMALL(7)
KABC_DECL_SMALL(8)
KABC_DECL_SMALL(9)
KABC_DECL_SMALL(10)
KABC_DECL_SMALL(11)

SmcSmallLaunch find_smc_small_kernel(int cost_id, int D, bool simple, ModelUnit* unit) {
    if (unit) {  // user prior families / a specialised model (plugin_registry.hpp)
        const PluginKernel k = unit_kernel(unit, kPfSmcSmall, D, simple ? 1 : 0);
        if (k.mod) return SmcSmallLaunch(k.mod, &smc_small_geom, (unsigned)kSmallBlock);
        if (unit_required(unit)) return SmcSmallLaunch();
        // (a specialisation that is not there (yet): the kernels below, same bits)
    }
    switch (cost_id) {
        case 1: return find_smc_small_kernel_cost_1(D, simple);
        case 2: return find_smc_small_kernel_cost_2(D, simple);
        case 3: return find_smc_small_kernel_cost_3(D, simple);
        case 4: return find_smc_small_kernel_cost_4(D, simple);
        case 5: return find_smc_small_kernel_cost_5(D, simple);
        case 6: return find_smc_small_kernel_cost_6(D, simple);
        case 7: return find_smc_small_kernel_cost_7(D, simple);
        case 8: return find_smc_small_kernel_cost_8(D, simple);
        case 9: return find_smc_small_kernel_cost_9(D, simple);
        case 10: return find_smc_small_kernel_cost_10(D, simple);
        case 11: return find_smc_small_kernel_cost_11(D, simple);
        default: {
            // (hipRTC user costs; a plugin .so built by hipcc has no such kernel: the other drivers serve)
            const PluginKernel k = plugin_kernel(find_plugin(cost_id), kPfSmcSmall, D, simple ? 1 : 0);
            return k.mod ? SmcSmallLaunch(k.mod, &smc_small_geom, (unsigned)kSmallBlock) : SmcSmallLaunch();
        }
    }
}

SmcLoopLaunch find_smc_loop_kernel(int cost_id, int D, bool simple, ModelUnit* unit) {
    if (unit) {  // user prior families / a specialised model (plugin_registry.hpp)
        const PluginKernel k = unit_kernel(unit, kPfSmcLoop, D, simple ? 1 : 0);
        if (k.mod) return SmcLoopLaunch(k.mod);
        if (unit_required(unit)) return SmcLoopLaunch();
    }
    switch (cost_id) {
        case 1: return find_smc_loop_kernel_cost_1(D, simple);
        case 2: return find_smc_loop_kernel_cost_2(D, simple);
        case 3: return find_smc_loop_kernel_cost_3(D, simple);
        case 4: return find_smc_loop_kernel_cost_4(D, simple);
        case 5: return find_smc_loop_kernel_cost_5(D, simple);
        case 6: return find_smc_loop_kernel_cost_6(D, simple);
        case 7: return find_smc_loop_kernel_cost_7(D, simple);
        case 8: return find_smc_loop_kernel_cost_8(D, simple);
        case 9: return find_smc_loop_kernel_cost_9(D, simple);
        case 10: return find_smc_loop_kernel_cost_10(D, simple);
        case 11: return find_smc_loop_kernel_cost_11(D, simple);
        default: {
            const PluginKernel k = plugin_kernel(find_plugin(cost_id), kPfSmcLoop, D, simple ? 1 : 0);
            if (k.host) return SmcLoopLaunch((SmcLoopLaunchFn)k.host);
            if (k.mod) return SmcLoopLaunch(k.mod);
            return nullptr;
        }
    }
}

SmcLaunch find_smc_kernel(int cost_id, int D, bool simple, ModelUnit* unit) {
    if (unit) {
        const PluginKernel k = unit_kernel(unit, kPfSmc, D, simple ? 1 : 0);
        if (k.mod) return SmcLaunch(k.mod, &smc_mcmc_geom, (unsigned)kSmcBlock);
        if (unit_required(unit)) return SmcLaunch();
    }
    switch (cost_id) {
        case 1: return find_smc_kernel_cost_1(D, simple);
        case 2: return find_smc_kernel_cost_2(D, simple);
        case 3: return find_smc_kernel_cost_3(D, simple);
        case 4: return find_smc_kernel_cost_4(D, simple);
        case 5: return find_smc_kernel_cost_5(D, simple);
        case 6: return find_smc_kernel_cost_6(D, simple);
        case 7: return find_smc_kernel_cost_7(D, simple);
        case 8: return find_smc_kernel_cost_8(D, simple);
        case 9: return find_smc_kernel_cost_9(D, simple);
        case 10: return find_smc_kernel_cost_10(D, simple);
        case 11: return find_smc_kernel_cost_11(D, simple);
        default: {
            const PluginKernel k = plugin_kernel(find_plugin(cost_id), kPfSmc, D, simple ? 1 : 0);
            if (k.host) return SmcLaunch((SmcLaunchFn)k.host);
            if (k.mod) return SmcLaunch(k.mod, &smc_mcmc_geom, (unsigned)kSmcBlock);
            return nullptr;
        }
    }
}

template <int D>
static void launch_smc_init_d(const SmcInitArgs& a, hipStream_t s) {
    const unsigned grid = smc_grid(a);
    if (grid == 0) return;
    hipLaunchKernelGGL((smc_init_kernel<D>), dim3(grid), dim3(kSmcBlock), 0, s, a);
}
template <int... Ds>
static void launch_smc_init(int D, const SmcInitArgs& a, hipStream_t s,
                            std::integer_sequence<int, Ds...>) {
    using Fn = void (*)(const SmcInitArgs&, hipStream_t);
    static const Fn fns[] = {&launch_smc_init_d<Ds + 1>...};
    fns[D - 1](a, s);
}

struct DevBufs {
    kabc_ctx_t* ctx = nullptr;                         // set: buffers come from / go back to its cache
    std::vector<std::pair<size_t, void*>> held;
    // bytes kept per context: 4 GiB of the 288 (the working set of smc at 2 M particles x 16 is
    // 0.9 GB: with the former 512 MiB every call allocated and freed two 268 MB buffers);
    // KABC_POOL_MB overrides
    static size_t pool_cap() {
        static const size_t cap = [] {
            const char* e = std::getenv("KABC_POOL_MB");
            const double mb = e ? std::atof(e) : 4096.0;
            return (size_t)((mb > 0.0 ? mb : 0.0) * (double)(1 << 20));
        }();
        return cap;
    }
    ~DevBufs() { release(); }
    // hands every buffer back (to the context's cache, else to the driver); the owner's pointers dangle
    void release() {
        for (auto& e : held) {
            if (!e.second) continue;
            if (ctx) {
                std::lock_guard<std::mutex> lk(ctx->pool_mu);
                if (ctx->pool_bytes + e.first <= pool_cap()) {
                    ctx->pool.push_back(e);
                    ctx->pool_bytes += e.first;
                    continue;
                }
            }
            (void)hipFree(e.second);
        }
        held.clear();
    }
    template <class T>
    hipError_t alloc(T** p, size_t n) {
        const size_t bytes = sizeof(T) * (n ? n : 1);
        if (ctx) {  // smallest cached buffer that fits and is not more than twice too large
            std::lock_guard<std::mutex> lk(ctx->pool_mu);
            int best = -1;
            for (int i = 0; i < (int)ctx->pool.size(); ++i)
                if (ctx->pool[i].first >= bytes && ctx->pool[i].first <= 2 * bytes + 4096 &&
                    (best < 0 || ctx->pool[i].first < ctx->pool[best].first))
                    best = i;
            if (best >= 0) {
                held.push_back(ctx->pool[best]);
                *p = (T*)ctx->pool[best].second;
                ctx->pool_bytes -= ctx->pool[best].first;
                ctx->pool.erase(ctx->pool.begin() + best);
                // (a recycled buffer carries the last run's bytes: the same poison as a fresh one)
                return poison_alloc() ? poison_fill((void*)*p, held.back().first) : hipSuccess;
            }
        }
        hipError_t e = dev_malloc(p, bytes);
        if (e == hipSuccess) held.push_back({bytes, (void*)*p});
        return e;
    }
};

}  // namespace kabc

using namespace kabc;

namespace kabc {
// workgroups of the select kernel: one per 2048 particles, at most 16 (32 from 2^17 particles
// on, 64 from 2^19, 128 from 2^21: its passes over the costs stream at the rate of the CUs it
// occupies, its five device-wide barriers cost 2.4 us each at 32 workgroups and 5 at 128 --
// measured at 2 M particles: 118 -> 83 us per call, profiles/r04_smc_large.txt);
// KABC_SMC_SELECT_BLOCKS overrides
static unsigned select_capacity();
static unsigned select_blocks(int64_t N) {
    long g = (long)((N + 2047) / 2048);
    const long cap = N >= (1 << 21) ? 128 : N >= (1 << 19) ? 64 : N >= (1 << 17) ? 32 : 16;
    if (g > cap) g = cap;
    if (const char* e = std::getenv("KABC_SMC_SELECT_BLOCKS")) {
        const long v = std::atol(e);
        if (v >= 1) g = v;
    }
    const long ntile = (long)((N + kSelBlock - 1) / kSelBlock);
    if (g > ntile) g = ntile;
    if (g > kSelMaxBlocks) g = kSelMaxBlocks;
    // at most a quarter of what the device holds of this kernel (128 of 512 on a whole MI355X): four
    // concurrent runs are always co-resident; a partitioned or masked device gets smaller grids
    const long quarter = (long)select_capacity() / 4 > 0 ? (long)select_capacity() / 4 : 1;
    if (g > quarter) g = quarter;
    return g < 1 ? 1u : (unsigned)g;
}
// The kernel's device-wide barrier needs its G <= 128 workgroups resident at the same time.
// They are launched as an ORDINARY grid that is known to fit: G is clamped to what the device
// holds of this kernel (occupancy x CUs), the stream is in order (nothing of this run is on the
// CUs when the kernel starts), and a tenant of another process delays the last workgroups but
// cannot starve them -- its kernels end; the barrier's spin is bounded all the same (5 s, then
// KABC_ERR_DEVICE; sel_grid_barrier).  hipLaunchCooperativeKernel gives the guarantee by
// construction and costs ~21 us per launch, host and device side (131 072 particles x 16: 88 -> 67 us
// per iteration without it, profiles/r04_smc_large.txt); KABC_SMC_COOPERATIVE=1 selects it.
static unsigned select_capacity() {
    // (per device: a process may drive partitioned or different devices)
    static std::mutex mu;
    static std::vector<std::pair<int, unsigned>> caps;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 1u;
    std::lock_guard<std::mutex> lk(mu);
    for (const auto& c : caps)
        if (c.first == dev) return c.second;
    unsigned cap = 1u;
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)smc_select_kernel, kSelBlock, 0) == hipSuccess &&
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess) {
        const long c = (long)per_cu * cus;
        cap = c < 1 ? 1u : (unsigned)c;
    }
    caps.emplace_back(dev, cap);
    return cap;
}
// set while a run is repeated with cooperative launches after an ordinary launch of the select kernel
// did not become co-resident in time (several large runs or another tenant holding the CUs)
static thread_local bool tl_smc_force_coop = false;
static bool select_cooperative() {
    const char* e = std::getenv("KABC_SMC_COOPERATIVE");  // (read per launch: a test switches it)
    return tl_smc_force_coop || (e && e[0] == '1');
}
static hipError_t launch_select(const SmcSelectArgs& sa, unsigned G, hipStream_t s) {
    SmcSelectArgs a = sa;
    const bool coop = select_cooperative();
    // 100 MHz ticks: 0.2 s (then the run is repeated cooperatively) / 5 s; KABC_SMC_BARRIER_TIMEOUT_MS (tests)
    a.barrier_timeout = coop ? 500000000ull : 20000000ull;
    if (const char* e = coop ? nullptr : std::getenv("KABC_SMC_BARRIER_TIMEOUT_MS")) {  // (the ordinary launch's)
        const double ms = std::atof(e);
        if (ms > 0) a.barrier_timeout = (unsigned long long)(ms * 1e5);
    }
    if (G <= 1u || !coop) {
        hipLaunchKernelGGL(smc_select_kernel, dim3(G), dim3(kSelBlock), 0, s, a);
        return hipGetLastError();
    }
    void* args[] = {&a};
    return hipLaunchCooperativeKernel((void*)smc_select_kernel, dim3(G), dim3(kSelBlock), args, 0, s);
}
}  // namespace kabc

extern "C" {

void kabc_smc_default_opts(kabc_smc_opts_t* o) {
    if (!o) return;
    o->nparticles = 100;
    o->alpha = 0.95;
    o->mcmc_retrys = 0;
    o->verbose = 0;
    o->mcmc_tol = 0.015;
    o->epstol = 0.0;
    o->r_epstol = NAN;
    o->min_r_ess = NAN;
    o->max_stretch = 2.0;
    o->seed = 0;
    o->max_iterations = 0;
}

// set while a run is repeated on the kernel-per-phase path after the persistent loop kernel gave up
static thread_local bool tl_smc_no_loop = false;
// kabc_smc_run_dist_mode: what the ranks of the communicator share out (KABC_SMC_DIST_*)
static thread_local int tl_smc_dist_mode = 0;
// kabc_smc_dist_stats: how the calling thread's last sharded run was driven
static thread_local int64_t tl_dist_stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};

static kabc_status_t smc_run_impl(kabc_ctx_t* ctx, kabc_comm_t* comm, const kabc_prior_t* prior,
                                  int32_t D, const kabc_cost_t* cost, const kabc_smc_opts_t* o,
                                  kabc_smc_result_t* res);

kabc_status_t kabc_smc_run(kabc_ctx_t* ctx, const kabc_prior_t* prior, int32_t D,
                           const kabc_cost_t* cost, const kabc_smc_opts_t* o,
                           kabc_smc_result_t* res) {
    return smc_run_impl(ctx, nullptr, prior, D, cost, o, res);
}

kabc_status_t kabc_smc_run_dist_mode(kabc_comm_t* comm, const kabc_prior_t* prior, int32_t D,
                                     const kabc_cost_t* cost, const kabc_smc_opts_t* o, int32_t mode,
                                     kabc_smc_result_t* res) {
    if (!comm) {
        set_error("kabc_smc_run_dist: communicator is NULL");
        return KABC_ERR_INVALID_ARG;
    }
    if (mode != KABC_SMC_DIST_COST_LOOP && mode != KABC_SMC_DIST_PARTICLES) {
        set_error("kabc_smc_run_dist_mode: mode is KABC_SMC_DIST_COST_LOOP or KABC_SMC_DIST_PARTICLES");
        return KABC_ERR_INVALID_ARG;
    }
    tl_smc_dist_mode = mode;
    // The select grid's "ordinary launch, bounded barrier wait, repeat cooperatively" turn is decided by
    // one rank from its own GPU: a rank that took it alone would restart from the initial exchange while
    // its peers go on with pass exchanges -- mismatched collectives.  A sharded run therefore launches
    // its selections cooperatively from the start (co-residency asserted by the runtime, ~21 us per
    // launch: nothing beside the host round trip per pass this mode already has).
    const bool was_coop = tl_smc_force_coop;
    tl_smc_force_coop = true;
    const kabc_status_t st = smc_run_impl(comm->ctx, comm, prior, D, cost, o, res);
    tl_smc_force_coop = was_coop;
    tl_smc_dist_mode = 0;
    return st;
}

void kabc_smc_dist_stats(int64_t out[8]) {
    if (out) std::memcpy(out, tl_dist_stats, sizeof tl_dist_stats);
}

kabc_status_t kabc_smc_run_dist(kabc_comm_t* comm, const kabc_prior_t* prior, int32_t D,
                                const kabc_cost_t* cost, const kabc_smc_opts_t* o,
                                kabc_smc_result_t* res) {
    // KABC_SMC_DIST=particles: the selection sharded as well (every rank must see the same value)
    const char* e = std::getenv("KABC_SMC_DIST");
    const int32_t mode = (e && std::strcmp(e, "particles") == 0) ? KABC_SMC_DIST_PARTICLES : KABC_SMC_DIST_COST_LOOP;
    return kabc_smc_run_dist_mode(comm, prior, D, cost, o, mode, res);
}

static kabc_status_t smc_run_impl(kabc_ctx_t* ctx, kabc_comm_t* comm, const kabc_prior_t* prior,
                                  int32_t D, const kabc_cost_t* cost, const kabc_smc_opts_t* o,
                                  kabc_smc_result_t* res) {
    std::memset(tl_dist_stats, 0, sizeof tl_dist_stats);  // (kabc_smc_dist_stats: of THIS run, whatever becomes of it)
    tl_dist_stats[7] = -1;
    if (!ctx || !prior || !cost || !o || !res) {
        set_error("kabc_smc_run: NULL argument");
        return KABC_ERR_INVALID_ARG;
    }
    const int64_t N = o->nparticles;
    const double alpha = o->alpha;
    const double r_epstol = std::isnan(o->r_epstol) ? std::pow(1.0 - alpha, 1.5) / 50.0 : o->r_epstol;
    const double min_r_ess = std::isnan(o->min_r_ess) ? alpha * alpha : o->min_r_ess;
    // src/smc.jl:107-118, same messages
#define KABC_REQ(cond, msg)          \
    if (!(cond)) {                   \
        set_error(msg);              \
        return KABC_ERR_INVALID_ARG; \
    }
    KABC_REQ(min_r_ess > 0, "min_r_ess must be > 0.")
    KABC_REQ(o->mcmc_retrys >= 0, "mcmc_retrys must be >= 0.")
    KABC_REQ(alpha > 0, "alpha must be > 0.")
    KABC_REQ(r_epstol >= 0, "r_epstol must be >= 0")
    KABC_REQ(o->mcmc_tol >= 0, "mcmc_tol must be >= 0")
    KABC_REQ(o->max_stretch > 1, "max_stretch must be > 1")
#undef KABC_REQ
    if (D < 1 || D > KABC_MAX_DIM_DYN) {
        set_error("length(prior) = %d is outside the device path's range 1..%d", D, KABC_MAX_DIM_DYN);
        return KABC_ERR_UNSUPPORTED;
    }
    // beyond KABC_MAX_DIM: run-time-dimension kernels (smc_dyn_kernels.hpp) on the
    // kernel-per-phase path; the selection / control kernels do not depend on D
    std::vector<kabc_prior_t> resolved((size_t)D);  // MvNormal components: device block, D
    if (kabc_status_t st = resolve_priors(ctx, prior, D, resolved.data())) return st;
    prior = resolved.data();
    const bool dyn = D > KABC_MAX_DIM;
    PriorSet P;
    std::memset(&P, 0, sizeof P);
    std::vector<PriorDev> Pdyn;
    bool prior_ok = true;
    if (dyn) {
        Pdyn.resize((size_t)D);
        for (int k = 0; k < D && prior_ok; ++k) prior_ok = prepare_prior(prior[k], Pdyn[k]);
    } else {
        prior_ok = prepare_priors(prior, D, P);
    }
    if (!prior_ok) {
        set_error("invalid prior parameters");
        return KABC_ERR_INVALID_ARG;
    }
    if (!cost_dim_ok_rt(cost->id, D)) {
        set_error("DeviceCost id %d does not accept D = %d", cost->id, D);
        return KABC_ERR_UNSUPPORTED;
    }
    bool simple = true;
    for (int k = 0; k < D; ++k) simple = simple && prior_is_simple(prior[k].kind);
    // (run-time compiled kernels are loaded on the CURRENT device)
    KABC_HIP_CHECK(hipSetDevice(ctx->device));
    // user prior families among the components, or a specialisation of exactly this model
    ModelUnit* unit = nullptr;
    if (kabc_status_t st = model_unit_for(prior, D, cost->id, &unit)) return st;
    // (a specialisation an entry point made on its own is asked for the kernels of the driver that
    // actually runs, below; here: does a propose / accept kernel exist at all)
    SmcLaunch mcmc = dyn ? SmcLaunch() : find_smc_kernel(cost->id, D, simple, unit_required(unit) ? unit : nullptr);
    bool mcmc_final = !unit || unit_required(unit) || dyn;
    if (!mcmc && !dyn && unit_required(unit)) return KABC_ERR_DEVICE;  // (message set by the compilation / load)
    // length(prior) > KABC_MAX_DIM: the run-time-dimension kernels -- of the unit (user prior families),
    // of the user cost (hipRTC form, or its plugin .so), or the built-in ones
    SmcDynLaunch dyn_fn;
    if (dyn) {
        if (unit) {
            const PluginKernel km = unit_kernel(unit, kPfSmcDyn, D, 0), ki = unit_kernel(unit, kPfSmcDyn, D, 1);
            dyn_fn = SmcDynLaunch(km.mod, ki.mod, unit_kernel(unit, kPfSmcDyn, D, 2).mod, unit_kernel(unit, kPfSmcDyn, D, 3).mod,
                                  unit_kernel(unit, kPfSmcDyn, D, 4).mod, unit_kernel(unit, kPfSmcDyn, D, 5).mod);
            if (!dyn_fn) return KABC_ERR_DEVICE;  // (message set by the compilation / load)
        } else if (cost->id >= KABC_COST_USER) {
            const CostPlugin* pl = find_plugin(cost->id);
            if (pl && pl->rtc) {
                const PluginKernel km = plugin_kernel(pl, kPfSmcDyn, D, 0), ki = plugin_kernel(pl, kPfSmcDyn, D, 1);
                dyn_fn = SmcDynLaunch(km.mod, ki.mod, plugin_kernel(pl, kPfSmcDyn, D, 2).mod, plugin_kernel(pl, kPfSmcDyn, D, 3).mod,
                                      plugin_kernel(pl, kPfSmcDyn, D, 4).mod, plugin_kernel(pl, kPfSmcDyn, D, 5).mod);
            } else if (pl && pl->smc_dyn) {
                dyn_fn = SmcDynLaunch((SmcDynLaunchFn)pl->smc_dyn());
            }
        } else {
            dyn_fn = SmcDynLaunch(find_smc_dyn_kernel(cost->id));
        }
    }
    if (!mcmc && !dyn_fn) {
        set_error("no gfx950 kernel instantiated for cost id %d, D = %d", cost->id, D);
        return KABC_ERR_UNSUPPORTED;
    }
    {
        const double mn = alpha < min_r_ess ? alpha : min_r_ess;
        const int64_t min_n = (int64_t)std::ceil(3.0 * D / mn);
        if (N < min_n) {
            set_error("nparticles must be >= %lld.", (long long)min_n);
            return KABC_ERR_INVALID_ARG;
        }
        if (N >= (1ll << 31)) {
            set_error("nparticles must be < 2^31");
            return KABC_ERR_INVALID_ARG;
        }
    }
    KABC_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    DevBufs bufs;
    bufs.ctx = ctx;
    double *th[2], *X[2], *lp[2], *d_params = nullptr, *d_data = nullptr, *d_out = nullptr,
           *d_Xout = nullptr;
    uint8_t* alive;
    int32_t* cidx;
    SmcCtrl* ctrl;
    unsigned long long* slots;
    kabc_smc_iter_t* d_log = nullptr;
    const int64_t log_cap = res->iter_log ? res->iter_log_cap : 0;
    // Sharded cost loop (kabc_smc_run_dist; the reference's own parallel leg, src/smc.jl:120-123,
    // 168): every rank keeps the whole ensemble and runs the selection redundantly; the
    // propose / prior-MH / COST / accept pass is split by workgroups of 64 particles -- rank r
    // takes [r * wg_per, (r + 1) * wg_per) -- and each pass ends with one grouped in-place
    // all-gather of the rows it produced (theta, X, logprior, the per-workgroup cost statistics
    // and counter lines).  Buffers are padded to `world` equal segments.
    const int world = comm ? comm->world : 1, rank = comm ? comm->rank : 0;
    const int64_t nwg_all = (N + kSmcBlock - 1) / kSmcBlock;
    const int64_t wg_per = (nwg_all + world - 1) / world;
    const int64_t wg_lo = std::min<int64_t>((int64_t)rank * wg_per, nwg_all);
    const int64_t wg_n = std::min<int64_t>(wg_lo + wg_per, nwg_all) - wg_lo;
    const size_t Npad = comm ? (size_t)(wg_per * world) * kSmcBlock : (size_t)N;
    for (int b = 0; b < 2; ++b) {
        KABC_HIP_CHECK(bufs.alloc(&th[b], Npad * D));
        KABC_HIP_CHECK(bufs.alloc(&X[b], Npad));
        KABC_HIP_CHECK(bufs.alloc(&lp[b], Npad));
    }
    KABC_HIP_CHECK(bufs.alloc(&alive, Npad));  // (padded: gathered at the end of a particle-sharded run)
    KABC_HIP_CHECK(bufs.alloc(&cidx, (size_t)N));
    KABC_HIP_CHECK(bufs.alloc(&ctrl, 1));
    KABC_HIP_CHECK(bufs.alloc(&slots, (size_t)kSmcSlots * 8 * world));
    SmcSelScratch* sel_scratch;
    KABC_HIP_CHECK(bufs.alloc(&sel_scratch, 1));
    KABC_HIP_CHECK(hipMemsetAsync(sel_scratch, 0, sizeof(SmcSelScratch), s));
    const unsigned selG = select_blocks(N);
    unsigned long long* part;  // per-workgroup cost statistics for the select kernel
    const int64_t npart = (N + kSmcBlock - 1) / kSmcBlock;
    KABC_HIP_CHECK(bufs.alloc(&part, (size_t)(comm ? wg_per * world : npart) * 4));
    KABC_HIP_CHECK(bufs.alloc(&d_out, (size_t)N * D));
    KABC_HIP_CHECK(bufs.alloc(&d_Xout, (size_t)N));
    if (log_cap > 0) KABC_HIP_CHECK(bufs.alloc(&d_log, (size_t)log_cap));
    KABC_HIP_CHECK(hipMemsetAsync(ctrl, 0, sizeof(SmcCtrl), s));
    KABC_HIP_CHECK(hipMemsetAsync(slots, 0, sizeof(unsigned long long) * kSmcSlots * 8 * world, s));
    // the all-gather at the end of a sharded pass / of the sharded init (buffer set `b`)
    int64_t n_collectives = 0, n_looks = 0, n_spec = 0, n_stalls = 0;
    auto exchange = [&](int b, bool with_slots) -> kabc_status_t {
        ++n_collectives;
        double* bases[5] = {th[b], X[b], lp[b], reinterpret_cast<double*>(part),
                            reinterpret_cast<double*>(slots)};
        const size_t counts[5] = {(size_t)wg_per * kSmcBlock * D, (size_t)wg_per * kSmcBlock,
                                  (size_t)wg_per * kSmcBlock, (size_t)wg_per * 4, (size_t)kSmcSlots * 8};
        return comm_allgather_many(comm, bases, counts, with_slots ? 5 : 4);
    };
    if (cost->nparams > 0) {
        KABC_HIP_CHECK(bufs.alloc(&d_params, (size_t)cost->nparams));
        KABC_HIP_CHECK(hipMemcpyAsync(d_params, cost->params, sizeof(double) * cost->nparams,
                                      hipMemcpyHostToDevice, s));
    }
    if (cost->ndata > 0) {
        KABC_HIP_CHECK(bufs.alloc(&d_data, (size_t)cost->ndata));
        KABC_HIP_CHECK(hipMemcpyAsync(d_data, cost->data, sizeof(double) * cost->ndata,
                                      hipMemcpyHostToDevice, s));
    }
    struct EvPair {  // released on every return path
        hipEvent_t a = nullptr, b = nullptr;
        ~EvPair() {
            if (a) (void)hipEventDestroy(a);
            if (b) (void)hipEventDestroy(b);
        }
    } evp;
    KABC_HIP_CHECK(hipEventCreate(&evp.a));
    KABC_HIP_CHECK(hipEventCreate(&evp.b));
    const hipEvent_t ev0 = evp.a, ev1 = evp.b;
    double mcmc_ms = 0.0;
    int64_t mcmc_timed = 0;

    SmcDynArgs da;
    std::memset(&da, 0, sizeof da);
    if (dyn) {
        PriorDev* d_prior;
        kabc_prior_t* d_raw;
        KABC_HIP_CHECK(bufs.alloc(&d_prior, (size_t)D));
        KABC_HIP_CHECK(bufs.alloc(&d_raw, (size_t)D));
        KABC_HIP_CHECK(bufs.alloc(&da.scratch, (size_t)N * 2 * D));
        KABC_HIP_CHECK(hipMemcpyAsync(d_prior, Pdyn.data(), sizeof(PriorDev) * D, hipMemcpyHostToDevice, s));
        KABC_HIP_CHECK(hipMemcpyAsync(d_raw, prior, sizeof(kabc_prior_t) * D, hipMemcpyHostToDevice, s));
        for (int b = 0; b < 2; ++b) {
            da.theta[b] = th[b];
            da.X[b] = X[b];
            da.lpi[b] = lp[b];
        }
        da.alive = alive;
        da.cidx = cidx;
        da.ctrl = ctrl;
        da.slots = slots;
        da.cost_params = d_params;
        da.cost_data = d_data;
        da.cost_ndata = cost->ndata;
        da.N = N;
        da.seed = o->seed;
        da.max_stretch = o->max_stretch;
        da.D = D;
        da.cost_id = cost->id;
        da.prior = d_prior;
        da.raw = d_raw;
        da.part = part;
        da.p0 = 0;
        da.p1 = N;
        // (a sharded run: every rank draws and costs ALL particles at the start -- the draws are counter-based,
        // the ranks end up with the same ensemble, nothing is exchanged; the passes are shared out)
        dyn_fn(da, s, 1);
        KABC_HIP_CHECK(hipGetLastError());
        if (comm) {
            da.p0 = std::min<int64_t>(wg_lo * kSmcBlock, N);
            da.p1 = std::min<int64_t>((wg_lo + wg_n) * kSmcBlock, N);
            da.slots = slots + (size_t)rank * kSmcSlots * 8;  // this rank's block of counter lines
        }
    }
    // :119-125
    if (!dyn) {
        SmcInitArgs a;
        std::memset(&a, 0, sizeof a);
        a.theta = th[0];
        a.X = X[0];
        a.lpi = lp[0];
        a.alive = alive;
        a.ctrl = ctrl;
        a.cost_params = d_params;
        a.cost_data = d_data;
        a.cost_ndata = cost->ndata;
        a.N = N;
        a.seed = o->seed;
        a.cost_id = cost->id;
        a.prior = P;
        a.part = part;
        if (comm) {  // this rank's workgroups only; everybody is alive at the start (:125)
            a.sharded = 1;
            a.wg0 = wg_lo;
            a.nwg = wg_n;
            KABC_HIP_CHECK(hipMemsetAsync(alive, 1, (size_t)N, s));
            SmcCtrl c0 = {};
            c0.eps = INFINITY;
            c0.eps_prev = INFINITY;
            c0.cost_evals = (unsigned long long)N;
            KABC_HIP_CHECK(hipMemcpyAsync(ctrl, &c0, sizeof c0, hipMemcpyHostToDevice, s));
            KABC_HIP_CHECK(hipStreamSynchronize(s));  // (c0 is on this stack frame)
        }
        std::memcpy(a.raw, prior, sizeof(kabc_prior_t) * D);
        const PluginKernel uk = unit ? unit_kernel(unit, kPfSmcInit, D, simple ? 1 : 0) : PluginKernel();
        if (uk.mod) {
            SmcInitLaunch(uk.mod, &smc_init_geom, (unsigned)kSmcBlock)(a, s);
        } else if (unit_required(unit)) {
            return KABC_ERR_DEVICE;
        } else if (const CostPlugin* pl = find_plugin(cost->id)) {
            using Fn = void (*)(const SmcInitArgs&, hipStream_t);
            const PluginKernel k = plugin_kernel(pl, kPfSmcInit, D, simple ? 1 : 0);
            if (k.host) SmcInitLaunch((Fn)k.host)(a, s);
            else if (k.mod) SmcInitLaunch(k.mod, &smc_init_geom, (unsigned)kSmcBlock)(a, s);
        } else {
            launch_smc_init(D, a, s, std::make_integer_sequence<int, KABC_MAX_DIM>{});
        }
        KABC_HIP_CHECK(hipGetLastError());
        if (comm)
            if (kabc_status_t st = exchange(0, false)) return st;
    }
    SmcSelectArgs sa;
    sa.Xbuf[0] = X[0];
    sa.Xbuf[1] = X[1];
    sa.alive = alive;
    sa.ridx = nullptr;
    sa.cidx = cidx;
    sa.ctrl = ctrl;
    sa.N = N;
    sa.alpha = alpha;
    sa.min_r_ess = min_r_ess;
    sa.mode = 0;
    sa.alive_out = alive;
    sa.part = part;
    sa.npart = npart;
    sa.scratch = sel_scratch;
    sa.stamps = nullptr;
    if (getenv("KABC_SMC_STAMPS")) {
        KABC_HIP_CHECK(bufs.alloc(&sa.stamps, 8));
        KABC_HIP_CHECK(hipMemsetAsync(sa.stamps, 0, 64, s));
    }
    auto do_select = [&](hipStream_t st) -> hipError_t { return launch_select(sa, selG, st); };
    // Particles sharded (KABC_SMC_DIST_PARTICLES): the selection over this rank's own costs, the ranks'
    // contributions all-gathered between its phases (smc_dsel_kernels.hpp).  Collective and synchronous.
    const bool dist_particles = comm && tl_smc_dist_mode == KABC_SMC_DIST_PARTICLES;
    DselArgs dz;
    std::memset(&dz, 0, sizeof dz);
    unsigned dselG = 0, dsel2G = 1;
    if (dist_particles) {
        dz.Xbuf[0] = X[0];
        dz.Xbuf[1] = X[1];
        dz.alive = alive;
        dz.cidx = cidx;
        dz.ctrl = ctrl;
        dz.part = part;
        dz.npart = npart;
        dz.N = N;
        dz.p_lo = std::min<int64_t>(wg_lo * kSmcBlock, N);
        dz.p_hi = std::min<int64_t>((wg_lo + wg_n) * kSmcBlock, N);
        dz.alpha = alpha;
        dz.min_r_ess = min_r_ess;
        dz.rank = rank;
        dz.world = world;
        dz.seg_len = wg_per * kSmcBlock;
        KABC_HIP_CHECK(bufs.alloc(&dz.st, 1));
        KABC_HIP_CHECK(bufs.alloc(&dz.hist, (size_t)world * kSelBins));
        KABC_HIP_CHECK(bufs.alloc(&dz.cand, (size_t)world * kDselCandStride));
        KABC_HIP_CHECK(bufs.alloc(&dz.misc, (size_t)world * 8));
        KABC_HIP_CHECK(bufs.alloc(&dz.seg, (size_t)world * dz.seg_len));
        KABC_HIP_CHECK(bufs.alloc(&dz.sub_cnt, (size_t)kDsel2MaxGrid));
        KABC_HIP_CHECK(hipMemsetAsync(dz.hist, 0, sizeof(unsigned) * world * kSelBins, s));
        KABC_HIP_CHECK(hipMemsetAsync(dz.cand, 0, sizeof(unsigned long long) * world * kDselCandStride, s));
        KABC_HIP_CHECK(hipMemsetAsync(dz.misc, 0, sizeof(unsigned long long) * world * 8, s));
        const int64_t len = dz.p_hi - dz.p_lo;
        const int64_t g = (len + 2 * kSelBlock - 1) / (2 * kSelBlock);  // 2048 particles per workgroup
        dselG = len <= 0 ? 0u : (unsigned)std::min<int64_t>(std::max<int64_t>(g, 1), kDselMaxGrid);
    }
    // the buffers of the one-exchange course (dsel2_*): a slot of an eighth of the rank's particles (the
    // window holds a few percent).  Without sharded particles the "rank" owns the whole ensemble.
    auto dsel2_setup = [&]() -> kabc_status_t {
        if (!dist_particles) {
            dz.Xbuf[0] = X[0];
            dz.Xbuf[1] = X[1];
            dz.alive = alive;
            dz.cidx = cidx;
            dz.ctrl = ctrl;
            dz.part = part;
            dz.npart = npart;
            dz.N = N;
            dz.p_lo = 0;
            dz.p_hi = N;
            dz.alpha = alpha;
            dz.min_r_ess = min_r_ess;
            dz.rank = 0;
            dz.world = 1;
            dz.seg_len = N;
            KABC_HIP_CHECK(bufs.alloc(&dz.st, 1));
            KABC_HIP_CHECK(bufs.alloc(&dz.sub_cnt, (size_t)kDsel2MaxGrid));
        }
        KABC_HIP_CHECK(hipMemsetAsync(dz.st, 0, sizeof(DselState), s));
        dz.spec_cap = std::max<int64_t>(kSelCand, (dz.seg_len / 8 + 1) & ~(int64_t)1);
        dz.spec_stride = kDselSpecKeys + dz.spec_cap;
        KABC_HIP_CHECK(bufs.alloc(&dz.spec, (size_t)dz.world * dz.spec_stride));
        KABC_HIP_CHECK(hipMemsetAsync(dz.spec, 0, sizeof(unsigned long long) * dz.world * dz.spec_stride, s));
        KABC_HIP_CHECK(bufs.alloc(&dz.bin, (size_t)kSelCand + 8));
        KABC_HIP_CHECK(hipMemsetAsync(dz.bin, 0, sizeof(unsigned long long) * 8, s));
        KABC_HIP_CHECK(hipMemsetAsync(dz.bin + 1, 0xff, sizeof(unsigned long long), s));
        const int64_t g2 = (N + 2 * kSelBlock - 1) / (2 * kSelBlock);  // (2048 particles per workgroup)
        // (KABC_DSEL2_G: A/B of the passes' grid -- 128 / 256 / 512 workgroups at 2 M particles: 417 / 412 / 433 us per
        // iteration, at 524 288: 134 / 140 / 138: more workgroups are more atomics on the payload, not more bandwidth)
        int64_t gmax = kDselMaxGrid;
        if (const char* eg = std::getenv("KABC_DSEL2_G")) gmax = std::max(1, std::min(atoi(eg), (int)kDsel2MaxGrid));
        dsel2G = (unsigned)std::min<int64_t>(std::max<int64_t>(g2, 1), gmax);
        return KABC_OK;
    };
    DselState hz;
    std::memset(&hz, 0, sizeof hz);
    auto dsel_look = [&]() -> kabc_status_t {  // the state the deciding kernel left
        ++n_looks;
        KABC_HIP_CHECK(hipGetLastError());
        KABC_HIP_CHECK(hipMemcpyAsync(&hz, dz.st, sizeof hz, hipMemcpyDeviceToHost, s));
        KABC_HIP_CHECK(hipStreamSynchronize(s));
        return KABC_OK;
    };
    auto dsel_gather = [&](void* base, size_t doubles_per_rank) -> kabc_status_t {
        ++n_collectives;
        double* b[1] = {reinterpret_cast<double*>(base)};
        const size_t c[1] = {doubles_per_rank};
        return comm_allgather_many(comm, b, c, 1);
    };
    long dsel_calls = 0, dsel_rounds = 0, dsel_lists = 0, dsel_scans = 0, dsel_resamples = 0;  // (KABC_SMC_STAMPS)
    auto dist_select = [&]() -> kabc_status_t {
        ++dsel_calls;
        // Every kernel tests the selection state before it acts, so the phases of the USUAL course are
        // enqueued without a look in between -- one histogram round when more than 4096 particles may be
        // alive, the candidate list, the ranking, the counts, the compaction -- and the host looks once;
        // what the usual course did not cover (more rounds, the scan above the range) is caught up below.
        auto round = [&]() -> kabc_status_t {
            if (dselG) hipLaunchKernelGGL(dsel_hist_kernel, dim3(dselG), dim3(kSelBlock), 0, s, dz);
            if (kabc_status_t st = dsel_gather(dz.hist, kSelBins / 2)) return st;
            hipLaunchKernelGGL(dsel_narrow_kernel, dim3(1), dim3(kSelBlock), 0, s, dz);
            ++dsel_rounds;
            return KABC_OK;
        };
        auto tail = [&](bool list) -> kabc_status_t {
            if (list) {
                if (dselG) hipLaunchKernelGGL(dsel_collect_kernel, dim3(dselG), dim3(kSelBlock), 0, s, dz);
                if (kabc_status_t st = dsel_gather(dz.cand, kDselCandStride)) return st;
                hipLaunchKernelGGL(dsel_rank_kernel, dim3(1), dim3(kSelBlock), 0, s, dz);
                ++dsel_lists;
            }
            if (dselG) hipLaunchKernelGGL(dsel_count_kernel, dim3(dselG), dim3(kSelBlock), 0, s, dz);
            if (kabc_status_t st = dsel_gather(dz.misc, 8)) return st;
            // (a rank without particles still decides -- ESS, resample -- like the others)
            hipLaunchKernelGGL(dsel_compact_kernel, dim3(dselG ? dselG : 1u), dim3(kSelBlock), 0, s, dz);
            return dsel_look();
        };
        hipLaunchKernelGGL(dsel_begin_kernel, dim3(1), dim3(kSelBlock), 0, s, dz);
        int rounds = 0;
        if (N > (int64_t)kSelCand) {
            if (kabc_status_t st = round()) return st;
            ++rounds;
        }
        if (kabc_status_t st = tail(true)) return st;
        while (!hz.error && hz.state != 3) {
            if (hz.state == 0) {
                if (++rounds > kSelRounds) {  // cannot happen: 7 rounds x 10 bits > 64 bits
                    set_error("smc: the sharded selection did not narrow its key range");
                    return KABC_ERR_INVALID_STATE;
                }
                if (kabc_status_t st = round()) return st;
                if (kabc_status_t st = tail(true)) return st;
            } else if (hz.state == 4) {  // the smallest key above the range is not among the candidates
                ++dsel_scans;
                if (dselG) hipLaunchKernelGGL(dsel_above_kernel, dim3(dselG), dim3(kSelBlock), 0, s, dz);
                if (kabc_status_t st = dsel_gather(dz.misc, 8)) return st;
                hipLaunchKernelGGL(dsel_above_fold_kernel, dim3(1), dim3(64), 0, s, dz);
                if (kabc_status_t st = tail(false)) return st;
            } else {
                set_error("smc: the sharded selection stopped in state %d", (int)hz.state);
                return KABC_ERR_INVALID_STATE;
            }
        }
        if (hz.error) return KABC_OK;  // (in the control block: the caller reads it)
        if (hz.resample) {
            ++dsel_resamples;
            if (kabc_status_t st = dsel_gather(dz.seg, (size_t)dz.seg_len / 2)) return st;
            hipLaunchKernelGGL(dsel_finish_kernel, dim3(256), dim3(256), 0, s, dz);
        }
        hipLaunchKernelGGL(dsel_publish_kernel, dim3(1), dim3(64), 0, s, dz);
        KABC_HIP_CHECK(hipGetLastError());
        return KABC_OK;
    };
    SmcMcmcArgs ma;
    std::memset(&ma, 0, sizeof ma);
    for (int b = 0; b < 2; ++b) {
        ma.theta[b] = th[b];
        ma.X[b] = X[b];
        ma.lpi[b] = lp[b];
    }
    ma.alive = alive;
    ma.cidx = cidx;
    ma.ctrl = ctrl;
    ma.slots = slots;
    ma.cost_params = d_params;
    ma.cost_data = d_data;
    ma.cost_ndata = cost->ndata;
    ma.N = N;
    ma.seed = o->seed;
    ma.max_stretch = o->max_stretch;
    ma.prior = P;
    ma.part = part;
    if (comm) {
        ma.sharded = 1;
        ma.wg0 = wg_lo;
        ma.nwg = wg_n;
        ma.slots = slots + (size_t)rank * kSmcSlots * 8;  // this rank's block of counter lines
    }
    // A prepared built-in cost (README.md:43-49's simulator): its parameter-independent sums for
    // EVERY particle of a pass come from a grid-wide pre-pass, one wavefront per cost evaluation
    // (ais_aux_kernels.hpp), instead of 500 Box-Muller pairs one after the other in the particle's
    // own lane.  That needs a launch per pass: such costs take the kernel-per-phase path.
    const int auxW = dyn ? 0 : aux_prepass_words(cost->id);
    double* d_aux = nullptr;
    AuxArgs xa;
    std::memset(&xa, 0, sizeof xa);
    // One pre-pass launch per batch of iterations instead of one per pass, when an iteration is
    // exactly one pass (no retries) and the ring of prepared passes stays small: README.md:80-84
    // (100 particles) is bound by its four launches per iteration.  kAuxRing = the iterations the
    // host enqueues between two looks at the control block (kBatch below).
    constexpr int kAuxRing = 16;
    const int aux_ring = (auxW && !comm && o->mcmc_retrys == 0 &&
                          (size_t)auxW * (size_t)N * kAuxRing * sizeof(double) <= ((size_t)32 << 20))
                             ? kAuxRing : 1;
    if (auxW) {
        KABC_HIP_CHECK(bufs.alloc(&d_aux, (size_t)auxW * N * aux_ring));
        xa.aux = d_aux + (comm ? wg_lo * kSmcBlock : 0);
        xa.cost_params = d_params;
        xa.cost_data = d_data;
        xa.cost_ndata = cost->ndata;
        xa.row_first = comm ? wg_lo * kSmcBlock : 0;
        xa.rows = comm ? std::min<int64_t>(wg_n * kSmcBlock, N - wg_lo * kSmcBlock) : N;
        if (xa.rows < 0) xa.rows = 0;
        xa.seed = o->seed;
        xa.nt = aux_ring;
        xa.ring = aux_ring > 1 ? aux_ring : 0;
        xa.domain = KABC_DOM_SMC_COST;
        xa.t_dev = &ctrl->pass;
        xa.word_stride = N;
        xa.skip_if = &ctrl->done;
        ma.aux = d_aux;
        ma.aux_ring = aux_ring;
    }
    auto run_pass = [&](hipStream_t st) {  // one propose / accept pass (+ its pre-pass)
        if (dyn) {
            dyn_fn(da, st, 0);
            return;
        }
        if (!mcmc_final) {  // the kernel-per-phase driver runs: the model's own kernel if it is there
            if (SmcLaunch m2 = find_smc_kernel(cost->id, D, simple, unit)) mcmc = m2;
            mcmc_final = true;
        }
        if (auxW && aux_ring == 1) launch_aux_prepass(cost->id, xa, st, 1);
        mcmc(ma, st);
    };
    SmcLoopParams lpz;
    lpz.mcmc_tol = o->mcmc_tol;
    lpz.epstol = o->epstol;
    lpz.r_epstol = r_epstol;
    lpz.max_iterations = o->max_iterations > 0 ? o->max_iterations : 100000;

    SmcCtrl hc;
    std::memset(&hc, 0, sizeof hc);
    kabc_status_t rc = KABC_OK;
    const int R = 1 + o->mcmc_retrys;

    // Path 0: a small ensemble (N <= 256: the reference's default nparticles = 100) in ONE
    // workgroup, the ensemble in LDS, workgroup barriers where the other drivers launch kernels or
    // cross the device (smc_small_kernel.hpp).  A prepared cost's pre-pass stays grid-wide: one
    // launch for the next kAuxRing passes, then one launch of this kernel for those passes.
    // KABC_SMC_SMALL=0, or an explicit choice of one of the other drivers (KABC_SMC_LOOP set),
    // skips it.
    bool looped = false;
    {
        const char* env = std::getenv("KABC_SMC_SMALL");
        const bool allow = !(env && env[0] == '0') && !std::getenv("KABC_SMC_LOOP") && !tl_smc_no_loop &&
                           !comm && !dyn && N <= (int64_t)kSmallBlock && (!auxW || aux_ring > 1);
        SmcSmallLaunch small_fn = allow ? find_smc_small_kernel(cost->id, D, simple, unit) : SmcSmallLaunch();
        if (small_fn) {
            SmcSmallArgs sm;
            std::memset(&sm, 0, sizeof sm);
            for (int b = 0; b < 2; ++b) {
                sm.theta[b] = th[b];
                sm.X[b] = X[b];
                sm.lpi[b] = lp[b];
            }
            sm.alive = alive;
            sm.ctrl = ctrl;
            sm.log = d_log;
            sm.log_cap = log_cap;
            sm.cost_params = d_params;
            sm.cost_data = d_data;
            sm.cost_ndata = cost->ndata;
            sm.N = N;
            sm.seed = o->seed;
            sm.max_stretch = o->max_stretch;
            sm.alpha = alpha;
            sm.min_r_ess = min_r_ess;
            sm.loop = lpz;
            sm.retry_n = R;
            sm.max_passes = auxW ? aux_ring : 0;
            sm.aux = auxW ? d_aux : nullptr;
            sm.aux_ring = aux_ring;
            PriorDev* d_prior;
            KABC_HIP_CHECK(bufs.alloc(&d_prior, (size_t)KABC_MAX_DIM));
            KABC_HIP_CHECK(hipMemcpyAsync(d_prior, &P, sizeof(PriorSet), hipMemcpyHostToDevice, s));
            sm.prior = d_prior;
            KABC_HIP_CHECK(hipEventRecord(ev0, s));
            for (int64_t launches = 0;; ++launches) {
                if (auxW) launch_aux_prepass(cost->id, xa, s, 1);
                small_fn(sm, s);
                KABC_HIP_CHECK(hipGetLastError());
                if (launches == 0) KABC_HIP_CHECK(hipEventRecord(ev1, s));
                KABC_HIP_CHECK(hipMemcpyAsync(&hc, ctrl, sizeof hc, hipMemcpyDeviceToHost, s));
                KABC_HIP_CHECK(hipStreamSynchronize(s));
                if (hc.done) break;
                if (!auxW) {  // (without a ring the kernel only returns when the loop is over)
                    set_error("smc: the small-ensemble kernel returned before the loop ended");
                    return KABC_ERR_DEVICE;
                }
            }
            float ms = 0.f;
            if (hc.pass > 0 && hipEventElapsedTime(&ms, ev0, ev1) == hipSuccess) {
                const unsigned long long first = auxW ? (hc.pass < (unsigned long long)aux_ring ? hc.pass : (unsigned long long)aux_ring) : hc.pass;
                mcmc_ms = ms / (double)first;  // the first launch per pass: there is no separate propose+accept kernel
                mcmc_timed = 1;
            }
            looped = true;
        }
    }

    // Path 1: the whole ε-loop as ONE persistent cooperative kernel (smc_loop_kernel.hpp) --
    // one thread per particle, for ensembles whose alive mask fits in LDS.  KABC_SMC_LOOP=0
    // selects the multi-kernel path below (also taken when the grid cannot be co-resident).
    {
        const char* env = std::getenv("KABC_SMC_LOOP");  // read per call: tests flip it
        const bool allow = !looped && !(env && env[0] == '0') && !tl_smc_no_loop && !comm && !auxW;
        const unsigned G = (unsigned)((N + kLoopBlock - 1) / kLoopBlock);
        SmcLoopLaunch loop_fn =
            (allow && !dyn && G <= (unsigned)kLoopMaxG) ? find_smc_loop_kernel(cost->id, D, simple, unit) : SmcLoopLaunch();
        if (loop_fn) {
            SmcLoopScratch* lsc;
            KABC_HIP_CHECK(bufs.alloc(&lsc, 1));
            KABC_HIP_CHECK(hipMemsetAsync(lsc, 0, sizeof(SmcLoopScratch), s));
            SmcLoopArgs la;
            std::memset(&la, 0, sizeof la);
            for (int b = 0; b < 2; ++b) {
                la.theta[b] = th[b];
                la.X[b] = X[b];
                la.lpi[b] = lp[b];
            }
            la.alive = alive;
            la.ctrl = ctrl;
            la.scratch = lsc;
            la.log = d_log;
            la.log_cap = log_cap;
            la.cost_params = d_params;
            la.cost_data = d_data;
            la.cost_ndata = cost->ndata;
            la.N = N;
            la.seed = o->seed;
            la.max_stretch = o->max_stretch;
            la.alpha = alpha;
            la.min_r_ess = min_r_ess;
            la.loop = lpz;
            la.retry_n = R;
            PriorDev* d_prior;
            KABC_HIP_CHECK(bufs.alloc(&d_prior, (size_t)KABC_MAX_DIM));
            KABC_HIP_CHECK(hipMemcpyAsync(d_prior, &P, sizeof(PriorSet), hipMemcpyHostToDevice, s));
            la.prior = d_prior;
            la.stamps = nullptr;
            if (getenv("KABC_SMC_STAMPS")) {
                KABC_HIP_CHECK(bufs.alloc(&la.stamps, 24));
                KABC_HIP_CHECK(hipMemsetAsync(la.stamps, 0, 192, s));
            }
            KABC_HIP_CHECK(hipEventRecord(ev0, s));
            const hipError_t le = loop_fn(la, G, s);
            if (le == hipSuccess) {
                KABC_HIP_CHECK(hipEventRecord(ev1, s));
                KABC_HIP_CHECK(hipMemcpyAsync(&hc, ctrl, sizeof hc, hipMemcpyDeviceToHost, s));
                KABC_HIP_CHECK(hipStreamSynchronize(s));
                float ms = 0.f;
                if (hc.pass > 0 && hipEventElapsedTime(&ms, ev0, ev1) == hipSuccess) {
                    mcmc_ms = ms / (double)hc.pass;  // the whole loop per pass: there is no
                    mcmc_timed = 1;                  // separate propose+accept kernel here
                }
                looped = true;
                unsigned long long st[24];
                if (la.stamps && hipMemcpy(st, la.stamps, 192, hipMemcpyDeviceToHost) == hipSuccess && st[8])
                    fprintf(stderr, "[kabc smc loop, 10 ns ticks per iteration] publish (records %.0f + next draws %.0f + sync,arrive %.0f) B1 %.0f fold %.0f "
                            "rounds %.0f gather %.0f B2 %.0f eps+mask %.0f mcmc %.0f | iterations %llu "
                            "cand/iter %.1f predicted %llu barriers %.2f/iter | eps+mask split: loads+fold %.0f rank %.0f patch+scan %.0f | mcmc split: philox+select %.0f issue+pre %.0f wait %.0f logpdf %.0f cost+accept %.0f tail %.0f\n",
                            (double)st[21] / st[8], (double)st[22] / st[8],
                            (double)st[0] / st[8], (double)st[1] / st[8], (double)st[2] / st[8],
                            (double)st[3] / st[8], (double)st[4] / st[8], (double)st[5] / st[8],
                            (double)st[6] / st[8], (double)st[7] / st[8], st[8], (double)st[9] / st[8],
                            st[10], (double)st[11] / st[8] / ((double)st[8] + 1) * 2.0, (double)st[12] / st[8],
                            (double)st[13] / st[8], (double)st[14] / st[8], (double)st[16] / st[8], (double)st[17] / st[8],
                            (double)st[18] / st[8], (double)st[19] / st[8], (double)st[20] / st[8], (double)st[7] / st[8]);
            } else if (le != hipErrorCooperativeLaunchTooLarge) {
                set_error("cooperative launch of the smc loop kernel failed: %s", hipGetErrorString(le));
                return KABC_ERR_DEVICE;
            } else {
                (void)hipGetLastError();
            }
        }
    }

    // Path 2: one ε-iteration = select kernel + 1..R propose/accept kernels + a pass-end kernel.
    // The ε-loop is decided on the device (smc_pass_end_kernel / smc_iter_end_kernel);
    // the host enqueues kBatch iterations and then reads the 128-byte control block
    // once.  Kernels enqueued past the end of the loop are no-ops.
    // Path 2, sharded (kabc_smc_run_dist): the same kernels; the host looks at the control block
    // after every pass -- it has to know which buffer set the pass wrote (that is what is
    // gathered) and whether the next pass is still open; with a simulator expensive enough to
    // be worth sharding, a host round trip per pass is noise.
    // One pass per iteration (mcmc_retrys = 0, the reference's default): the buffer set a pass writes is
    // known without asking, so kDistBatch iterations are enqueued -- kernels and collectives -- between two
    // looks at the control block, and the selection is the ONE-exchange course (smc_dsel_kernels.hpp,
    // dsel2_*): begin (+ the previous pass's end), spec, [all-gather,] decide, apply, index -- ordinary
    // launches, no device-wide barrier; a selection that stalls turns everything behind it into no-ops and
    // is repeated after the look (phase by phase when the particles are sharded, else by the select kernel).
    // Sharded particles: two collectives per iteration.  Sharded cost loop, and single-GPU runs of 2^20
    // particles and more: the same course with the whole ensemble as the one rank's (no exchange).
    // KABC_SMC_DIST_LOOKS=1 (sharded runs) / KABC_SMC_SPEC_SELECT=0: the courses below, also taken when
    // retry passes are allowed; KABC_SMC_SPEC_SELECT=1: this course on a single GPU at any size.
    bool blind_done = false;
    {
        const char* envl = std::getenv("KABC_SMC_DIST_LOOKS");
        const char* envs = std::getenv("KABC_SMC_SPEC_SELECT");
        // single GPU: five ordinary launches against the select kernel's one with its device-wide barriers --
        // measured (profiles/r06_spec_select_ab.txt) 59 / 73 / 133 / 418 us per iteration against 49 / 64 /
        // 130 / 436 at 32 768 / 131 072 / 524 288 / 2 M particles: the default from 2^20 particles on
        const bool spec_single = envs ? envs[0] != '0' : N >= ((int64_t)1 << 20);
        const bool spec_ok = comm ? !(envs && envs[0] == '0') : spec_single;
        const bool blind = !looped && R == 1 &&
                           (comm ? !(envl && envl[0] == '1') : (spec_single && !dyn && !auxW && !tl_smc_no_loop));
        const bool sel2 = blind && (dist_particles || spec_ok);  // (else: the select kernel, batched)
        if (sel2)
            if (kabc_status_t st = dsel2_setup()) return st;
        constexpr int kDistBatch = 8;
        unsigned decideG = std::min<unsigned>(dsel2G, (unsigned)kDselMaxGrid);  // (KABC_DSEL2_DECIDE_G: A/B of the deciding kernel's grid)
        if (const char* eg = std::getenv("KABC_DSEL2_DECIDE_G")) decideG = (unsigned)std::max(1, std::min(atoi(eg), (int)kDselMaxGrid));
        Dsel2End e2;
        std::memset(&e2, 0, sizeof e2);
        e2.slots = slots;
        e2.log = d_log;
        e2.log_cap = log_cap;
        e2.P = lpz;
        e2.mcmc_tol = o->mcmc_tol;
        e2.nregions = world;
        int cur_host = 0;      // ctrl->cur as long as the loop runs: one flip per iteration
        bool pending = false;  // a pass whose end has not been folded yet
        auto look = [&]() -> kabc_status_t {
            ++n_looks;
            KABC_HIP_CHECK(hipGetLastError());
            KABC_HIP_CHECK(hipMemcpyAsync(&hc, ctrl, sizeof hc, hipMemcpyDeviceToHost, s));
            if (sel2) KABC_HIP_CHECK(hipMemcpyAsync(&hz, dz.st, sizeof hz, hipMemcpyDeviceToHost, s));
            KABC_HIP_CHECK(hipStreamSynchronize(s));
            return KABC_OK;
        };
        auto timed_pass = [&]() -> kabc_status_t {
            const bool timed = (mcmc_timed == 0);
            if (timed) KABC_HIP_CHECK(hipEventRecord(ev0, s));
            run_pass(s);
            if (timed) {
                KABC_HIP_CHECK(hipEventRecord(ev1, s));
                mcmc_timed = -1;  // (read after the next look)
            }
            return KABC_OK;
        };
        // one iteration with the selection phase by phase (the course of the looked loop below)
        bool stop = false;
        auto looked_iteration = [&]() -> kabc_status_t {
            if (pending) {
                e2.do_pass_end = 1;
                e2.end_only = 1;
                hipLaunchKernelGGL(dsel2_begin_kernel, dim3(1), dim3(kSmcSlots), 0, s, dz, e2);
                pending = false;
                if (kabc_status_t st = look()) return st;
                if (hc.done) {
                    stop = true;
                    return KABC_OK;
                }
            }
            if (dist_particles) {
                if (kabc_status_t st = dist_select()) return st;
            } else {
                KABC_HIP_CHECK(hipMemsetAsync(&dz.st->stalled, 0, sizeof(int32_t), s));
                KABC_HIP_CHECK(do_select(s));
            }
            if (kabc_status_t st = look()) return st;
            if (hc.done) {
                stop = true;
                return KABC_OK;
            }
            if (kabc_status_t st = timed_pass()) return st;
            if (comm)
                if (kabc_status_t st = exchange(1 - hc.cur, true)) return st;
            cur_host = 1 - hc.cur;
            pending = true;
            return KABC_OK;
        };
        // no window without two values of eps: the first two selections phase by phase (unless the whole
        // ensemble fits a slot: then every alive key is a candidate from the start)
        if (sel2 && !(N <= dz.spec_cap && N <= (int64_t)kDselStage))
            for (int i = 0; i < 2 && !stop; ++i) {
                ++n_stalls;
                if (kabc_status_t st = looked_iteration()) return st;
            }
        int kb = kDistBatch;
        while (blind && !stop) {
            for (int it = 0; it < kb; ++it) {
                if (sel2) {
                    e2.do_pass_end = pending ? 1 : 0;
                    e2.end_only = 0;
                    hipLaunchKernelGGL(dsel2_begin_kernel, dim3(1), dim3(kSmcSlots), 0, s, dz, e2);
                    hipLaunchKernelGGL(dsel2_spec_kernel, dim3(dsel2G), dim3(kSelBlock), 0, s, dz);
                    if (dist_particles)
                        if (kabc_status_t st = dsel_gather(dz.spec, (size_t)dz.spec_stride)) return st;
                    hipLaunchKernelGGL(dsel2_decide_kernel, dim3(decideG), dim3(kSelBlock), 0, s, dz);
                    hipLaunchKernelGGL(dsel2_apply_kernel, dim3(dsel2G), dim3(kSelBlock), 0, s, dz);
                    hipLaunchKernelGGL(dsel2_index_kernel, dim3(dsel2G), dim3(kSelBlock), 0, s, dz);
                    ++n_spec;
                } else {
                    if (pending)
                        hipLaunchKernelGGL(smc_pass_end_kernel, dim3(1), dim3(kSmcSlots), 0, s, ctrl, slots, N,
                                           o->mcmc_tol, 1, d_log, log_cap, lpz, world);
                    KABC_HIP_CHECK(do_select(s));
                }
                if (kabc_status_t st = timed_pass()) return st;
                if (comm)
                    if (kabc_status_t st = exchange(1 - cur_host, true)) return st;
                cur_host ^= 1;
                pending = true;
            }
            // the last pass's end, then the look
            if (sel2) {
                e2.do_pass_end = 1;
                e2.end_only = 1;
                hipLaunchKernelGGL(dsel2_begin_kernel, dim3(1), dim3(kSmcSlots), 0, s, dz, e2);
            } else {
                hipLaunchKernelGGL(smc_pass_end_kernel, dim3(1), dim3(kSmcSlots), 0, s, ctrl, slots, N,
                                   o->mcmc_tol, 1, d_log, log_cap, lpz, world);
            }
            pending = false;
            if (kabc_status_t st = look()) return st;
            if (hc.done) break;
            cur_host = hc.cur;
            kb = std::min(2 * kb, kDistBatch);
            if (sel2 && hz.stalled) {
                // every kernel behind the stalled selection was a no-op (the collectives re-gathered what
                // was there): that selection phase by phase, its pass, and on with shorter batches
                ++n_stalls;
                n_spec -= 1;
                if (std::getenv("KABC_SMC_STAMPS") && rank == 0)
                    fprintf(stderr, "[kabc smc] the one-exchange selection of iteration %lld stalled (reason %d)\n",
                            (long long)hz.stall_iteration + 1, (int)hz.stalled);
                if (kabc_status_t st = looked_iteration()) return st;
                kb = 1;
            }
        }
        if (mcmc_timed == -1) {
            float ms = 0.f;
            mcmc_timed = 0;
            if (hipEventElapsedTime(&ms, ev0, ev1) == hipSuccess) {
                mcmc_ms = ms;
                mcmc_timed = 1;
            }
        }
        blind_done = blind;
    }
    while (comm && !looped && !blind_done) {
        if (dist_particles) {
            if (kabc_status_t st = dist_select()) return st;
        } else {
            KABC_HIP_CHECK(do_select(s));
        }
        KABC_HIP_CHECK(hipMemcpyAsync(&hc, ctrl, sizeof hc, hipMemcpyDeviceToHost, s));
        KABC_HIP_CHECK(hipStreamSynchronize(s));
        ++n_looks;
        if (hc.done) break;
        bool ended = false;
        for (int r = 0; r < R && !hc.done && hc.pass_open; ++r) {
            const bool timed = (mcmc_timed == 0 && r == 0);
            if (timed) KABC_HIP_CHECK(hipEventRecord(ev0, s));
            run_pass(s);
            if (timed) KABC_HIP_CHECK(hipEventRecord(ev1, s));
            KABC_HIP_CHECK(hipGetLastError());
            if (kabc_status_t st = exchange(1 - hc.cur, true)) return st;
            ended = (r == R - 1);
            hipLaunchKernelGGL(smc_pass_end_kernel, dim3(1), dim3(kSmcSlots), 0, s, ctrl, slots, N,
                               o->mcmc_tol, ended ? 1 : 0, d_log, log_cap, lpz, world);
            KABC_HIP_CHECK(hipMemcpyAsync(&hc, ctrl, sizeof hc, hipMemcpyDeviceToHost, s));
            KABC_HIP_CHECK(hipStreamSynchronize(s));
            ++n_looks;
            if (timed) {
                float ms = 0.f;
                if (hipEventElapsedTime(&ms, ev0, ev1) == hipSuccess) {
                    mcmc_ms += ms;
                    ++mcmc_timed;
                }
            }
        }
        if (!ended && !hc.done) {
            hipLaunchKernelGGL(smc_iter_end_kernel, dim3(1), dim3(1), 0, s, ctrl, d_log, log_cap, N, lpz);
            KABC_HIP_CHECK(hipMemcpyAsync(&hc, ctrl, sizeof hc, hipMemcpyDeviceToHost, s));
            KABC_HIP_CHECK(hipStreamSynchronize(s));
            ++n_looks;
        }
        if (hc.done) break;
    }
    const int kGroup = 4;                        // retry passes enqueued between host checks
    const int kBatch = (R <= kGroup) ? 16 : 1;   // iterations per host sync
    bool first = true;
    while (!looped && !comm && !blind_done) {
        // (the next kBatch passes at once: pass t = *ctrl.pass + 1 + s in slot t mod kAuxRing)
        if (auxW && aux_ring > 1) launch_aux_prepass(cost->id, xa, s, 1);
        for (int it = 0; it < kBatch; ++it) {
            KABC_HIP_CHECK(do_select(s));
            bool ended = false;  // the iteration's end rode on the last pass_end launch
            for (int r0 = 0; r0 < R; r0 += kGroup) {
                const int r1 = (r0 + kGroup < R) ? r0 + kGroup : R;
                for (int r = r0; r < r1; ++r) {
                    const bool timed = (it == 0 && r == 0);
                    if (timed) KABC_HIP_CHECK(hipEventRecord(ev0, s));
                    run_pass(s);
                    if (timed) KABC_HIP_CHECK(hipEventRecord(ev1, s));
                    ended = (r == R - 1);
                    hipLaunchKernelGGL(smc_pass_end_kernel, dim3(1), dim3(kSmcSlots), 0, s, ctrl,
                                       slots, N, o->mcmc_tol, ended ? 1 : 0, d_log, log_cap, lpz);
                }
                if (r1 < R) {  // many retries allowed: look before enqueueing more
                    KABC_HIP_CHECK(hipMemcpyAsync(&hc, ctrl, sizeof hc, hipMemcpyDeviceToHost, s));
                    KABC_HIP_CHECK(hipStreamSynchronize(s));
                    if (hc.done || !hc.pass_open) break;
                }
            }
            if (!ended)
                hipLaunchKernelGGL(smc_iter_end_kernel, dim3(1), dim3(1), 0, s, ctrl, d_log,
                                   log_cap, N, lpz);
        }
        KABC_HIP_CHECK(hipGetLastError());
        KABC_HIP_CHECK(hipMemcpyAsync(&hc, ctrl, sizeof hc, hipMemcpyDeviceToHost, s));
        KABC_HIP_CHECK(hipStreamSynchronize(s));
        float ms = 0.f;
        if ((first || !hc.done) && hipEventElapsedTime(&ms, ev0, ev1) == hipSuccess) {
            mcmc_ms += ms;
            ++mcmc_timed;
        }
        first = false;
        if (hc.done) break;
    }
    if (hc.error) {
        if (hc.error == 1) {
            set_error("quantiles are undefined in presence of NaNs");
            rc = KABC_ERR_NAN_COST;
        } else if (hc.error == 3) {
            set_error("smc: a device-wide barrier of a cooperative launch timed out after 5 s (the device is wedged)");
            rc = KABC_ERR_DEVICE;
        } else if (hc.error == 4) {
            // more particles share one histogram bin of the costs than the loop kernel's candidate
            // list holds (heavy ties): a limit of that kernel, not of the problem.  The run is
            // repeated below on the kernel-per-phase path -- every draw is counter-based, so the
            // repetition is the same run.
            rc = KABC_ERR_UNSUPPORTED;
        } else {
            set_error("collection must be non-empty");
            rc = KABC_ERR_INVALID_STATE;
        }
    }
    if (rc == KABC_OK) {
        // :200-205
        SmcFinalArgs fa;
        for (int b = 0; b < 2; ++b) {
            fa.theta[b] = th[b];
            fa.X[b] = X[b];
        }
        fa.ctrl = ctrl;
        fa.out = d_out;
        fa.Xout = d_Xout;
        fa.N = N;
        fa.D = D;
        fa.prior = P;
        fa.dprior = dyn ? da.prior : nullptr;
        hipLaunchKernelGGL(smc_finalize_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s,
                           fa);
        KABC_HIP_CHECK(hipGetLastError());
        if (res->theta)
            KABC_HIP_CHECK(hipMemcpyAsync(res->theta, d_out, sizeof(double) * N * D,
                                          hipMemcpyDeviceToHost, s));
        if (res->cost)
            KABC_HIP_CHECK(hipMemcpyAsync(res->cost, d_Xout, sizeof(double) * N,
                                          hipMemcpyDeviceToHost, s));
        if (dist_particles)  // every rank holds the mask of its own range only
            if (kabc_status_t st = dsel_gather(alive, (size_t)wg_per * kSmcBlock / 8)) return st;
        if (res->alive)
            KABC_HIP_CHECK(hipMemcpyAsync(res->alive, alive, (size_t)N, hipMemcpyDeviceToHost, s));
        const int64_t nlog = hc.iteration < log_cap ? hc.iteration : log_cap;
        if (nlog > 0)
            KABC_HIP_CHECK(hipMemcpyAsync(res->iter_log, d_log, sizeof(kabc_smc_iter_t) * nlog,
                                          hipMemcpyDeviceToHost, s));
        KABC_HIP_CHECK(hipStreamSynchronize(s));
        if (o->verbose)  // @show iteration, ϵ, ESS  (src/smc.jl:143)
            for (int64_t i = 0; i < nlog; ++i)
                fprintf(stderr, "(iteration, ϵ, ESS) = (%lld, %.17g, %lld)\n", (long long)(i + 1),
                        res->iter_log[i].eps, (long long)res->iter_log[i].ess);
        res->eps = hc.eps;
        res->iterations = hc.iteration;
        res->n_alive = hc.n_alive;
        res->cost_evals = hc.cost_evals;
        res->proposals = hc.proposals;
        res->kernel_ms_mcmc = mcmc_timed ? mcmc_ms / (double)mcmc_timed : 0.0;
        res->mcmc_launches = (int64_t)hc.pass;
    }
    // (test hook: KABC_SMC_LOOP_GIVE_UP=1 makes every loop-kernel run count as given up)
    if (looped && !tl_smc_no_loop && std::getenv("KABC_SMC_LOOP_GIVE_UP")) hc.error = 4;
    // (test hook: KABC_SMC_SELECT_TIME_OUT=1 makes the first, ordinary-launch run count as timed out)
    if (!looped && !select_cooperative() && hc.error == 0 && std::getenv("KABC_SMC_SELECT_TIME_OUT")) hc.error = 3;
    if (hc.error == 3 && !looped && !select_cooperative()) {
        // an ordinary launch of the select grid did not become co-resident within 0.2 s: the same run
        // with cooperative launches (co-residency asserted by the runtime; ~21 us per launch dearer).
        // Single-rank runs only: a sharded run launches cooperatively from its first selection (below),
        // so no rank can take this turn on its own while its peers go on exchanging passes.
        (void)hipStreamSynchronize(s);
        bufs.release();  // (the repetition allocates its own; the first run's go back to the cache first)
        tl_smc_force_coop = true;
        const kabc_status_t st2 = smc_run_impl(ctx, comm, prior, D, cost, o, res);
        tl_smc_force_coop = false;
        return st2;
    }
    if (hc.error == 4 && looped && !tl_smc_no_loop) {
        (void)hipStreamSynchronize(s);
        bufs.release();
        tl_smc_no_loop = true;
        const kabc_status_t st2 = kabc_smc_run(ctx, prior, D, cost, o, res);
        tl_smc_no_loop = false;
        return st2;
    }
    {
        tl_dist_stats[0] = hc.iteration;
        tl_dist_stats[1] = n_collectives;
        tl_dist_stats[2] = n_looks;
        tl_dist_stats[3] = blind_done && n_spec > 0 ? std::max<int64_t>(hc.iteration - n_stalls, 0) : 0;
        tl_dist_stats[4] = n_stalls;
        tl_dist_stats[5] = (int64_t)hc.pass;
        tl_dist_stats[6] = blind_done ? 1 : 0;
        tl_dist_stats[7] = blind_done ? (dist_particles ? 2 : (comm ? 1 : 0)) : -1;  // collectives of an iteration's usual course
    }
    if (dist_particles && sa.stamps && rank == 0)
        fprintf(stderr, "[kabc smc sharded selection, %d ranks] %ld selections: %ld histogram rounds, %ld candidate "
                        "lists, %ld scans above the range, %ld resamples (workgroups per pass and rank: %u)\n",
                world, dsel_calls, dsel_rounds, dsel_lists, dsel_scans, dsel_resamples, dselG);
    if (sa.stamps) {
        unsigned long long st[8];
        if (hipMemcpy(st, sa.stamps, 64, hipMemcpyDeviceToHost) == hipSuccess && st[7])
            fprintf(stderr, "[kabc smc select stamps, cycles/call @100MHz-ticks] stats %.0f narrow %.0f list %.0f eps %.0f tiles %.0f write %.0f (calls %llu)\n",
                    (double)st[0] / st[7], (double)st[1] / st[7], (double)st[2] / st[7], (double)st[3] / st[7],
                    (double)st[4] / st[7], (double)st[5] / st[7], st[7]);
    }
    return rc;
}


// ---- pfilter(prior, cost, N; ...) -- src/smc.jl:275-340 ---------------------------
void kabc_pfilter_default_opts(kabc_pfilter_opts_t* o) {
    if (!o) return;
    o->nparticles = 100;
    o->q = 0.7;
    o->eff_tol = 0.1;
    o->epstol = -INFINITY;
    o->proposal_width = 0.75;
    o->max_iters = -1;
    o->verbose = 0;
    o->reserved = 0;
    o->seed = 0;
}

int64_t kabc_pfilter_nparticles(int64_t N, double q, int32_t D) {
    const int64_t lowN = 4 * (int64_t)D;  // :276-279
    if ((double)N * q <= (double)lowN) N = (int64_t)std::ceil((double)(lowN + 1) / q);
    return N;
}

}  // extern "C"

namespace {
template <int D>
void pf_l_init(const AbcdeArgs& a, hipStream_t s) {
    hipLaunchKernelGGL((abcde_init_kernel<D>), dim3((unsigned)((a.N + kAbcdeBlock - 1) / kAbcdeBlock)),
                       dim3(kAbcdeBlock), 0, s, a);
}
template <int D>
void pf_l_attempt(const PfArgs& a, hipStream_t s) {
    hipLaunchKernelGGL((pf_attempt_kernel<D>), dim3((unsigned)((a.N + kPfBlock - 1) / kPfBlock)),
                       dim3(kPfBlock), 0, s, a);
}
template <int... Ds>
AbcdeLaunchFn pf_pick_init(int D, std::integer_sequence<int, Ds...>) {
    static const AbcdeLaunchFn f[] = {&pf_l_init<Ds + 1>...};
    return f[D - 1];
}
template <int... Ds>
PfLaunchFn pf_pick_attempt(int D, std::integer_sequence<int, Ds...>) {
    static const PfLaunchFn f[] = {&pf_l_attempt<Ds + 1>...};
    return f[D - 1];
}
template <int D>
void pf_l_small(const PfSmallArgs& a, hipStream_t s) {
    hipLaunchKernelGGL((pf_small_kernel<D>), dim3(1), dim3(kPfSmallBlock), 0, s, a);
}
template <int... Ds>
PfSmallLaunchFn pf_pick_small(int D, std::integer_sequence<int, Ds...>) {
    static const PfSmallLaunchFn f[] = {&pf_l_small<Ds + 1>...};
    return f[D - 1];
}
}  // namespace

extern "C" {

kabc_status_t kabc_pfilter_run(kabc_ctx_t* ctx, const kabc_prior_t* prior, int32_t D,
                               const kabc_cost_t* cost, const kabc_pfilter_opts_t* o,
                               kabc_pfilter_result_t* res) {
    if (!ctx || !prior || !cost || !o || !res) {
        set_error("kabc_pfilter_run: NULL argument");
        return KABC_ERR_INVALID_ARG;
    }
    if (D < 1 || D > KABC_MAX_DIM_DYN) {
        set_error("length(prior) = %d is outside the device path's range 1..%d", D, KABC_MAX_DIM_DYN);
        return KABC_ERR_UNSUPPORTED;
    }
    if (!(o->q > 0 && o->q <= 1) || o->nparticles < 1) {
        set_error("pfilter needs 0 < q <= 1 and N >= 1");
        return KABC_ERR_INVALID_ARG;
    }
    // length(prior) > KABC_MAX_DIM: the run-time-dimension instantiation (D = 0) of the kernels,
    // prior components as device arrays
    std::vector<kabc_prior_t> resolved((size_t)D);  // MvNormal components: device block, D
    if (kabc_status_t st = resolve_priors(ctx, prior, D, resolved.data())) return st;
    prior = resolved.data();
    const bool dyn = D > KABC_MAX_DIM;
    PriorSet P;
    std::memset(&P, 0, sizeof P);
    std::vector<PriorDev> Pdyn((size_t)(dyn ? D : 0));
    bool prior_ok = true;
    if (dyn)
        for (int k = 0; k < D && prior_ok; ++k) prior_ok = prepare_prior(prior[k], Pdyn[k]);
    else
        prior_ok = prepare_priors(prior, D, P);
    if (!prior_ok) {
        set_error("invalid prior parameters");
        return KABC_ERR_INVALID_ARG;
    }
    if (!cost_dim_ok_rt(cost->id, D)) {
        set_error("DeviceCost id %d does not accept D = %d", cost->id, D);
        return KABC_ERR_UNSUPPORTED;
    }
    AbcdeLaunch f_init;
    PfLaunch f_att;
    {
        const CostPlugin* pl = cost->id >= KABC_COST_USER ? find_plugin(cost->id) : nullptr;
        if (dyn && pl && !pl->rtc) {
            set_error("pfilter with length(prior) = %d > %d: built-in DeviceCosts or a user cost in the hipRTC "
                      "form (kabc_compile_cost_plugin)", D, KABC_MAX_DIM);
            return KABC_ERR_UNSUPPORTED;
        }
    }
    // (run-time compiled kernels are loaded on the CURRENT device)
    KABC_HIP_CHECK(hipSetDevice(ctx->device));
    const int64_t N = kabc_pfilter_nparticles(o->nparticles, o->q, D);
    // Up to 256 particles (the reference's default is 100) with a built-in cost: the whole loop in one
    // launch of ONE workgroup (pf_small_kernel; KABC_PF_SMALL=0 or KABC_PF_PASSES=1: the launches per
    // phase).  At this size a model's own kernels would buy nothing: none are asked for.
    bool small = false;
    {
        const char* e = std::getenv("KABC_PF_SMALL");
        const char* pe = std::getenv("KABC_PF_PASSES");
        small = N <= (int64_t)kPfSmallBlock && cost->id < KABC_COST_USER && !(e && e[0] == '0') && !(pe && pe[0] == '1');
    }
    ModelUnit* unit = nullptr;
    if (kabc_status_t st = model_unit_for(prior, D, cost->id, &unit, !small)) return st;
    if (unit && unit_required(unit)) small = false;  // (user prior families: only their unit knows them)
    if (unit) {  // user prior families / a specialised model (plugin_registry.hpp)
        const PluginKernel ki = unit_kernel(unit, kPfAbcdeInit, D, 0), ka = unit_kernel(unit, kPfAttempt, D, 0);
        if (ki.mod) f_init = AbcdeLaunch(ki.mod, &abcde_geom, (unsigned)kAbcdeBlock);
        if (ka.mod) f_att = PfLaunch(ka.mod, &pf_geom, (unsigned)kPfBlock);
        if ((!f_init || !f_att) && unit_required(unit)) return KABC_ERR_DEVICE;
        // (a specialisation that is not there (yet): what is missing comes from below, same bits)
    }
    if (!f_init || !f_att) {
        AbcdeLaunch b_init;
        PfLaunch b_att;
        if (dyn && cost->id < KABC_COST_USER) {
            b_init = AbcdeLaunch(&pf_l_init<0>);
            b_att = PfLaunch(&pf_l_attempt<0>);
        } else if (const CostPlugin* p = find_plugin(cost->id)) {
            const PluginKernel ki = plugin_kernel(p, kPfAbcdeInit, D, 0), ka = plugin_kernel(p, kPfAttempt, D, 0);
            b_init = ki.host ? AbcdeLaunch((AbcdeLaunchFn)ki.host)
                             : ki.mod ? AbcdeLaunch(ki.mod, &abcde_geom, (unsigned)kAbcdeBlock) : AbcdeLaunch();
            b_att = ka.host ? PfLaunch((PfLaunchFn)ka.host)
                            : ka.mod ? PfLaunch(ka.mod, &pf_geom, (unsigned)kPfBlock) : PfLaunch();
            if (!b_init || !b_att) {
                set_error("cost plugin has no pfilter kernels for D = %d", D);
                return KABC_ERR_UNSUPPORTED;
            }
        } else {
            b_init = pf_pick_init(D, std::make_integer_sequence<int, KABC_MAX_DIM>{});
            b_att = pf_pick_attempt(D, std::make_integer_sequence<int, KABC_MAX_DIM>{});
        }
        if (!f_init) f_init = b_init;
        if (!f_att) f_att = b_att;
    }
    KABC_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    DevBufs bufs;
    bufs.ctx = ctx;
    double *th, *Cc, *lpi, *d_params = nullptr, *d_data = nullptr, *d_out, *d_cout;
    uint8_t *ones, *ok, *pending;
    int32_t* cidx;
    SmcCtrl* sel;
    PfCtrl* pctrl;
    AbcdeCtrl* actrl;
    KABC_HIP_CHECK(bufs.alloc(&th, (size_t)N * D));
    KABC_HIP_CHECK(bufs.alloc(&Cc, (size_t)N));
    KABC_HIP_CHECK(bufs.alloc(&lpi, (size_t)N));
    KABC_HIP_CHECK(bufs.alloc(&d_out, (size_t)N * D));
    KABC_HIP_CHECK(bufs.alloc(&d_cout, (size_t)N));
    KABC_HIP_CHECK(bufs.alloc(&ones, (size_t)N));
    KABC_HIP_CHECK(bufs.alloc(&ok, (size_t)N));
    KABC_HIP_CHECK(bufs.alloc(&pending, (size_t)N));
    KABC_HIP_CHECK(bufs.alloc(&cidx, (size_t)N));
    KABC_HIP_CHECK(bufs.alloc(&sel, 1));
    KABC_HIP_CHECK(bufs.alloc(&pctrl, 1));
    KABC_HIP_CHECK(bufs.alloc(&actrl, 1));
    KABC_HIP_CHECK(hipMemsetAsync(ones, 1, (size_t)N, s));
    KABC_HIP_CHECK(hipMemsetAsync(sel, 0, sizeof(SmcCtrl), s));
    KABC_HIP_CHECK(hipMemsetAsync(pctrl, 0, sizeof(PfCtrl), s));
    KABC_HIP_CHECK(hipMemsetAsync(actrl, 0, sizeof(AbcdeCtrl), s));
    PriorDev* d_prior = nullptr;
    kabc_prior_t* d_raw = nullptr;
    if (dyn) {
        KABC_HIP_CHECK(bufs.alloc(&d_prior, (size_t)D));
        KABC_HIP_CHECK(bufs.alloc(&d_raw, (size_t)D));
        KABC_HIP_CHECK(hipMemcpyAsync(d_prior, Pdyn.data(), sizeof(PriorDev) * D, hipMemcpyHostToDevice, s));
        KABC_HIP_CHECK(hipMemcpyAsync(d_raw, prior, sizeof(kabc_prior_t) * D, hipMemcpyHostToDevice, s));
    }
    if (cost->nparams > 0) {
        KABC_HIP_CHECK(bufs.alloc(&d_params, (size_t)cost->nparams));
        KABC_HIP_CHECK(hipMemcpyAsync(d_params, cost->params, sizeof(double) * cost->nparams,
                                      hipMemcpyHostToDevice, s));
    }
    if (cost->ndata > 0) {
        KABC_HIP_CHECK(bufs.alloc(&d_data, (size_t)cost->ndata));
        KABC_HIP_CHECK(hipMemcpyAsync(d_data, cost->data, sizeof(double) * cost->ndata,
                                      hipMemcpyHostToDevice, s));
    }
    // :280-294 (same initial-draw loop as ABCDE, its own stream domains)
    {
        AbcdeArgs a;
        std::memset(&a, 0, sizeof a);
        a.theta[0] = th;
        a.delta[0] = Cc;
        a.lpi[0] = lpi;
        a.ctrl = actrl;
        a.cost_params = d_params;
        a.cost_data = d_data;
        a.cost_ndata = cost->ndata;
        a.N = N;
        a.seed = o->seed;
        a.cost_id = cost->id;
        a.dom_init = KABC_DOM_PF_INIT;
        a.dom_init_cost = KABC_DOM_PF_INIT_COST;
        a.prior = P;
        a.D_rt = D;
        a.dprior = d_prior;
        a.draw = d_raw;
        if (!dyn) std::memcpy(a.raw, prior, sizeof(kabc_prior_t) * D);
        f_init(a, s);
        KABC_HIP_CHECK(hipGetLastError());
        AbcdeCtrl hc;
        KABC_HIP_CHECK(hipMemcpyAsync(&hc, actrl, sizeof hc, hipMemcpyDeviceToHost, s));
        KABC_HIP_CHECK(hipStreamSynchronize(s));
        if (hc.error) {
            set_error("pfilter: the prior never produced a finite (cost, logpdf) pair for some particle");
            return KABC_ERR_RETRY_EXHAUSTED;
        }
    }
    SmcSelectArgs sa;
    sa.Xbuf[0] = Cc;
    sa.Xbuf[1] = Cc;
    sa.alive = ones;
    sa.alive_out = ok;
    sa.ridx = nullptr;
    sa.cidx = cidx;
    sa.ctrl = sel;
    sa.N = N;
    sa.alpha = o->q;
    sa.min_r_ess = 1.0;
    sa.stamps = nullptr;
    sa.mode = 1;
    sa.part = nullptr;  // pfilter's kernels do not produce the partials: select scans C
    sa.npart = 0;
    KABC_HIP_CHECK(bufs.alloc(&sa.scratch, 1));
    KABC_HIP_CHECK(hipMemsetAsync(sa.scratch, 0, sizeof(SmcSelScratch), s));
    const unsigned selG = select_blocks(N);
    auto do_select = [&](hipStream_t st) -> hipError_t { return launch_select(sa, selG, st); };
    PfArgs pa;
    std::memset(&pa, 0, sizeof pa);
    pa.theta = th;
    pa.C = Cc;
    pa.lpi = lpi;
    pa.pending = pending;
    pa.idxok = cidx;
    pa.sel = sel;
    pa.ctrl = pctrl;
    pa.cost_params = d_params;
    pa.cost_data = d_data;
    pa.cost_ndata = cost->ndata;
    pa.N = N;
    pa.seed = o->seed;
    pa.cost_id = cost->id;
    pa.proposal_width = o->proposal_width;
    pa.prior = P;
    pa.D_rt = D;
    pa.dprior = d_prior;
    int64_t iters = 0;
    double eps = 0.0, eff = 0.0;
    SmcCtrl hsel;
    PfCtrl hp;
    const char* pf_env = std::getenv("KABC_PF_PASSES");  // =1: one launch per attempt (the former scheme)
    const bool pf_loop = !(pf_env && pf_env[0] == '1');
    std::memset(&hp, 0, sizeof hp);
    // Default: every bad particle's rejection loop inside one launch, the stop tests on the device,
    // FOUR iterations enqueued per host round trip (kernels of iterations after the last are
    // no-ops); verbose runs look after every iteration, to print it.
    bool batched_done = false;
    if (small && pf_loop) {
        PfSmallArgs sm;
        std::memset(&sm, 0, sizeof sm);
        sm.pf = pa;
        sm.pf.pending = nullptr;
        sm.pf.idxok = nullptr;
        sm.pf.sel = nullptr;
        sm.q = o->q;
        sm.eff_tol = o->eff_tol;
        sm.epstol = o->epstol;
        sm.max_iters = o->max_iters;
        sm.iters_this_launch = o->verbose ? 1 : 0;
        const PfSmallLaunchFn f_small = dyn ? &pf_l_small<0> : pf_pick_small(D, std::make_integer_sequence<int, KABC_MAX_DIM>{});
        while (true) {
            f_small(sm, s);
            KABC_HIP_CHECK(hipGetLastError());
            KABC_HIP_CHECK(hipMemcpyAsync(&hp, pctrl, sizeof hp, hipMemcpyDeviceToHost, s));
            KABC_HIP_CHECK(hipStreamSynchronize(s));
            if (hp.error == 9) {
                set_error("pfilter: a particle was not replaced after 2^24 proposals");
                return KABC_ERR_RETRY_EXHAUSTED;
            }
            if (hp.error) {
                set_error("pfilter: quantile of the costs is undefined (NaN or empty)");
                return KABC_ERR_NAN_COST;
            }
            if (o->verbose)
                fprintf(stderr, "(iters, ϵ, eff) = (%lld, %.17g, %.17g)\n", (long long)hp.iters, hp.eps, hp.eff);
            if (hp.done) break;
        }
        iters = hp.iters;
        eps = hp.eps;
        eff = hp.eff;
        batched_done = true;
    }
    while (pf_loop && !batched_done) {
        const int kIterBatch = o->verbose ? 1 : 4;
        for (int b = 0; b < kIterBatch; ++b) {
            ++iters;
            KABC_HIP_CHECK(do_select(s));
            hipLaunchKernelGGL(pf_mark_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s,
                               pending, ok, pctrl, sel, N);
            pa.iteration = (uint64_t)iters;
            pa.attempt = 0u;
            pa.loop_attempts = 1;
            f_att(pa, s);
            hipLaunchKernelGGL(pf_iter_end_kernel, dim3(1), dim3(1), 0, s, pctrl, sel, N, o->eff_tol,
                               o->epstol, o->max_iters);
        }
        KABC_HIP_CHECK(hipGetLastError());
        KABC_HIP_CHECK(hipMemcpyAsync(&hp, pctrl, sizeof hp, hipMemcpyDeviceToHost, s));
        KABC_HIP_CHECK(hipStreamSynchronize(s));
        if (!select_cooperative() && hp.error == 0 && std::getenv("KABC_SMC_SELECT_TIME_OUT")) hp.error = 3;  // (test hook)
        if (hp.error == 3) {
            if (!select_cooperative()) {  // an ordinary launch that did not become co-resident in time: the same run, cooperatively
                tl_smc_force_coop = true;
                const kabc_status_t st2 = kabc_pfilter_run(ctx, prior, D, cost, o, res);
                tl_smc_force_coop = false;
                return st2;
            }
            set_error("pfilter: a device-wide barrier of a cooperative launch timed out after 5 s (the device is wedged)");
            return KABC_ERR_DEVICE;
        }
        if (hp.error == 9) {
            set_error("pfilter: a particle was not replaced after 2^24 proposals");
            return KABC_ERR_RETRY_EXHAUSTED;
        }
        if (hp.error) {
            set_error("pfilter: quantile of the costs is undefined (NaN or empty)");
            return KABC_ERR_NAN_COST;
        }
        if (o->verbose)
            fprintf(stderr, "(iters, ϵ, eff) = (%lld, %.17g, %.17g)\n", (long long)hp.iters, hp.eps, hp.eff);
        if (hp.done) {
            iters = hp.iters;
            eps = hp.eps;
            eff = hp.eff;
            batched_done = true;
            break;
        }
    }
    while (!batched_done) {
        ++iters;
        KABC_HIP_CHECK(do_select(s));
        hipLaunchKernelGGL(pf_mark_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s,
                           pending, ok, pctrl, sel, N);
        pa.iteration = (uint64_t)iters;
        uint32_t attempt = 0;
        while (true) {
            pa.loop_attempts = 0;  // (KABC_PF_PASSES=1: a launch per attempt, eight per host round trip)
            for (int g = 0; g < 8; ++g) {
                pa.attempt = attempt++;
                f_att(pa, s);
            }
            KABC_HIP_CHECK(hipGetLastError());
            KABC_HIP_CHECK(hipMemcpyAsync(&hp, pctrl, sizeof hp, hipMemcpyDeviceToHost, s));
            KABC_HIP_CHECK(hipMemcpyAsync(&hsel, sel, sizeof hsel, hipMemcpyDeviceToHost, s));
            KABC_HIP_CHECK(hipStreamSynchronize(s));
            if (hsel.error == 3) {
                if (!select_cooperative()) {  // an ordinary launch that did not become co-resident in time: the same run, cooperatively
                    tl_smc_force_coop = true;
                    const kabc_status_t st2 = kabc_pfilter_run(ctx, prior, D, cost, o, res);
                    tl_smc_force_coop = false;
                    return st2;
                }
                set_error("pfilter: a device-wide barrier of a cooperative launch timed out after 5 s (the device is wedged)");
                return KABC_ERR_DEVICE;
            }
            if (hsel.error) {
                set_error("pfilter: quantile of the costs is undefined (NaN or empty)");
                return KABC_ERR_NAN_COST;
            }
            if (hp.remaining == 0) break;
            if (attempt >= (1u << 24)) {
                set_error("pfilter: a particle was not replaced after 2^24 proposals");
                return KABC_ERR_RETRY_EXHAUSTED;
            }
        }
        // NOTE passes enqueued after the last replacement are no-ops (nothing pending)
        eps = hsel.eps;
        const double nbad = (double)(N - hsel.ess);
        eff = nbad / (double)hp.nreps;  // :327 (0/0 = NaN when nothing was bad, as in Julia)
        if (o->verbose)
            fprintf(stderr, "(iters, ϵ, eff) = (%lld, %.17g, %.17g)\n", (long long)iters, eps, eff);
        if (eff < o->eff_tol) break;
        if (eps < o->epstol) break;
        if (o->max_iters >= 0 && iters > o->max_iters) break;  // src/smc.jl:332; < 0 = Inf
        if (!(hp.nreps > 0)) break;  // nothing left to refresh: eff is NaN forever
    }
    SmcFinalArgs fa;
    fa.theta[0] = fa.theta[1] = th;
    fa.X[0] = fa.X[1] = Cc;
    fa.ctrl = sel;
    fa.out = d_out;
    fa.Xout = d_cout;
    fa.N = N;
    fa.D = D;
    fa.prior = P;
    fa.dprior = d_prior;
    hipLaunchKernelGGL(smc_finalize_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, fa);
    KABC_HIP_CHECK(hipGetLastError());
    if (res->theta)
        KABC_HIP_CHECK(hipMemcpyAsync(res->theta, d_out, sizeof(double) * N * D,
                                      hipMemcpyDeviceToHost, s));
    if (res->cost)
        KABC_HIP_CHECK(hipMemcpyAsync(res->cost, d_cout, sizeof(double) * N, hipMemcpyDeviceToHost, s));
    KABC_HIP_CHECK(hipStreamSynchronize(s));
    res->eps = eps;
    res->eff = eff;
    res->iterations = iters;
    res->nreps = hp.total_reps;
    res->cost_evals = hp.cost_evals;
    return KABC_OK;
}

}  // extern "C"
