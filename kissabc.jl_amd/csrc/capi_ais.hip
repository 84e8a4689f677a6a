// capi_ais.hip -- the AIS entry points of the C ABI (include/kabc.h):
// AIS(N) + AISState + step(init) + step(advance) of src/KissABC.jl:21-80,
// executed by the gfx950 kernels in ais_kernels.hpp.
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <thread>
#include <vector>

#include "ais_aux_kernels.hpp"
#include "ais_dyn_kernels.hpp"
#include "ais_small_kernel.hpp"
#include "host_common.hpp"
#include "plugin_registry.hpp"

namespace kabc {

// (three translation units per cost, csrc/Makefile: dimensions 1..7 of the NORMAL prior class,
// 1..7 of the other classes, 8..KABC_MAX_DIM)
#define KABC_DECL_COST(id)                                                \
    AisLaunchFn find_ais_kernel_cost_##id(int D, int pc);                 \
    AisLaunchFn find_ais_kernel_cost_##id##_nrm(int D, int pc);           \
    AisLaunchFn find_ais_kernel_cost_##id##_hi(int D, int pc);            \
    AisSmallLaunchFn find_ais_small_kernel_cost_##id(int D, int pc);      \
    AisSmallLaunchFn find_ais_small_kernel_cost_##id##_nrm(int D, int pc); \
    AisSmallLaunchFn find_ais_small_kernel_cost_##id##_hi(int D, int pc);
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

AisDynLaunchFn find_ais_dyn_kernel(int cost_id);
constexpr int kAisInstSplit = 7;  // = KABC_INST_DHI of the low translation units (csrc/Makefile)
static AisLaunchFn ais_inst_pick(int D, int pc, AisLaunchFn (*lo)(int, int), AisLaunchFn (*nrm)(int, int),
                                 AisLaunchFn (*hi)(int, int)) {
    if (D > kAisInstSplit) return hi(D, pc);
    return (pc % kPriorClasses) == kPriorNormal ? nrm(D, pc) : lo(D, pc);
}

AisLaunch find_ais_kernel(int cost_id, int D, int pc) {
    switch (cost_id) {
        case 1: return ais_inst_pick(D, pc, find_ais_kernel_cost_1, find_ais_kernel_cost_1_nrm, find_ais_kernel_cost_1_hi);
        case 2: return ais_inst_pick(D, pc, find_ais_kernel_cost_2, find_ais_kernel_cost_2_nrm, find_ais_kernel_cost_2_hi);
        case 3: return ais_inst_pick(D, pc, find_ais_kernel_cost_3, find_ais_kernel_cost_3_nrm, find_ais_kernel_cost_3_hi);
        case 4: return ais_inst_pick(D, pc, find_ais_kernel_cost_4, find_ais_kernel_cost_4_nrm, find_ais_kernel_cost_4_hi);
        case 5: return ais_inst_pick(D, pc, find_ais_kernel_cost_5, find_ais_kernel_cost_5_nrm, find_ais_kernel_cost_5_hi);
        case 6: return ais_inst_pick(D, pc, find_ais_kernel_cost_6, find_ais_kernel_cost_6_nrm, find_ais_kernel_cost_6_hi);
        case 7: return ais_inst_pick(D, pc, find_ais_kernel_cost_7, find_ais_kernel_cost_7_nrm, find_ais_kernel_cost_7_hi);
        case 8: return ais_inst_pick(D, pc, find_ais_kernel_cost_8, find_ais_kernel_cost_8_nrm, find_ais_kernel_cost_8_hi);
        case 9: return ais_inst_pick(D, pc, find_ais_kernel_cost_9, find_ais_kernel_cost_9_nrm, find_ais_kernel_cost_9_hi);
        case 10: return ais_inst_pick(D, pc, find_ais_kernel_cost_10, find_ais_kernel_cost_10_nrm, find_ais_kernel_cost_10_hi);
        case 11: return ais_inst_pick(D, pc, find_ais_kernel_cost_11, find_ais_kernel_cost_11_nrm, find_ais_kernel_cost_11_hi);
        default: {
            const PluginKernel k = plugin_kernel(find_plugin(cost_id), kPfAis, D, pc);
            if (k.host) return AisLaunch((AisLaunchFn)k.host);
            if (k.mod) return AisLaunch(k.mod, &ais_half_geom, (unsigned)kAisBlock);
            return nullptr;
        }
    }
}

// the one-workgroup kernel of small ensembles (ais_small_kernel.hpp); pcx's prior class is BOX,
// NORMAL or GENERAL (small_class below)
static AisSmallLaunchFn ais_small_pick(int D, int pc, AisSmallLaunchFn (*lo)(int, int), AisSmallLaunchFn (*nrm)(int, int),
                                       AisSmallLaunchFn (*hi)(int, int)) {
    if (D > kAisInstSplit) return hi(D, pc);
    return (pc % kPriorClasses) == kPriorNormal ? nrm(D, pc) : lo(D, pc);
}
AisSmallLaunch find_ais_small_kernel(int cost_id, int D, int pc) {
#define KABC_SMALL_CASE(id) \
    case id: return AisSmallLaunch(ais_small_pick(D, pc, find_ais_small_kernel_cost_##id, find_ais_small_kernel_cost_##id##_nrm, find_ais_small_kernel_cost_##id##_hi));
    switch (cost_id) {
        KABC_SMALL_CASE(1)
        KABC_SMALL_CASE(2)
        KABC_SMALL_CASE(3)
        KABC_SMALL_CASE(4)
        KABC_SMALL_CASE(5)
        KABC_SMALL_CASE(6)
        KABC_SMALL_CASE(7)
        KABC_SMALL_CASE(8)
        KABC_SMALL_CASE(9)
        KABC_SMALL_CASE(10)
        KABC_SMALL_CASE(11)
        default: {
            const PluginKernel k = plugin_kernel(find_plugin(cost_id), kPfAisSmall, D, pc);
            if (k.mod) return AisSmallLaunch(k.mod, &ais_small_geom, (unsigned)kAisSmallBlock);
            return AisSmallLaunch();
        }
    }
#undef KABC_SMALL_CASE
}

template <int D>
static void launch_init_d(const InitArgs& a, hipStream_t s, unsigned nchains) {
    const unsigned grid = (unsigned)((a.rows_owned + kInitBlock - 1) / kInitBlock);
    if (grid == 0) return;
    hipLaunchKernelGGL((ais_init_kernel<D>), dim3(grid, nchains), dim3(kInitBlock), 0, s, a);
}

template <int... Ds>
static void launch_init_table(int D, const InitArgs& a, hipStream_t s, unsigned nchains,
                              std::integer_sequence<int, Ds...>) {
    using Fn = void (*)(const InitArgs&, hipStream_t, unsigned);
    static const Fn fns[] = {&launch_init_d<Ds + 1>...};
    fns[D - 1](a, s, nchains);
}

bool launch_ais_init(int D, const InitArgs& a, hipStream_t s, unsigned nchains, ModelUnit* unit) {
    if (unit) {  // user prior families / a specialisation compiled ahead: the unit's own init kernel
        const PluginKernel k = unit_kernel(unit, kPfAisInit, D, 0);
        if (k.mod) {
            AisInitLaunch(k.mod, &ais_init_geom, (unsigned)kInitBlock)(a, s, nchains);
            return true;
        }
        if (unit_required(unit)) return false;  // (message set by the compilation / load)
    }
    if (const CostPlugin* p = find_plugin(a.cost_id)) {
        const PluginKernel k = plugin_kernel(p, kPfAisInit, D, 0);
        using Fn = void (*)(const InitArgs&, hipStream_t, unsigned);
        if (k.host) AisInitLaunch((Fn)k.host)(a, s, nchains);
        else if (k.mod) AisInitLaunch(k.mod, &ais_init_geom, (unsigned)kInitBlock)(a, s, nchains);
        else return false;
        return true;
    }
    launch_init_table(D, a, s, nchains, std::make_integer_sequence<int, KABC_MAX_DIM>{});
    return true;
}

}  // namespace kabc

using namespace kabc;

static constexpr int kTraceBufs = 3;

struct kabc_ais {
    kabc_ctx_t* ctx;
    int32_t D, posterior, cost_id;
    double eps;
    kabc_prior_t raw[KABC_MAX_DIM];
    PriorSet prior;
    double* d_cost_params;
    double* d_cost_data;
    int64_t cost_ndata;
    int64_t N;             // total walkers (all ranks)
    int64_t rows[2];       // global rows per half
    int64_t rows_owned[2]; // owned rows per half (all segments)
    // Ownership is block-cyclic over `xk` EXCHANGE CHUNKS: chunk k of a half is the row range
    // [k * world * cper, (k + 1) * world * cper), split into `world` rank segments of cper rows
    // -- so that the all-gather of chunk k is in place and contiguous (count = cper * D) and
    // can run while the kernels of chunk k + 1 compute.  xk = 1 is the plain contiguous
    // split.  Draws are keyed by the global walker id, never by the owner: results do not
    // depend on xk or world.
    struct Seg { int64_t first, count, off; };  // global first row, rows, offset in lp / ll
    std::vector<Seg> seg[2];
    int32_t xk;
    int64_t cper[2];       // rows per rank and chunk (all-gather count / D)
    kabc_comm_t* comm;     // library-owned exchange (kabc_ais_create_dist), else NULL
    // length(prior) > KABC_MAX_DIM: run-time-dimension kernels (ais_dyn_kernels.hpp)
    AisDynLaunch dyn;
    std::vector<kabc_prior_t> raw_dyn;
    std::vector<PriorDev> prior_dyn;
    kabc_prior_t* d_raw;   // [D] raw components (dyn)
    double* d_scratch;     // [max rows_owned][2][D] (dyn)
    int32_t nchains;       // independent ensembles in this handle (kabc_ais_create_batch), else 1
    uint64_t* d_seeds;     // [nchains] (batch handles)
    unsigned long long* d_chain_retries;  // [nchains]
    uint32_t id_base[2];   // global walker id of row 0 of each half
    double* d_half[2];     // global halves [rows[h]][D]
    bool own_halves;
    double* d_lp[2];
    double* d_ll[2];
    DevCounters* d_counters;
    unsigned long long* d_slots;  // [kCounterSlots][8]
    PriorDev* d_prior;            // [KABC_MAX_DIM] prepared components
    uint64_t seed, t;
    int32_t rank, world;
    AisLaunch launch;      // half-generation kernel (a small-ensemble handle resolves it at first need)
    int32_t pc;            // its prior class
    // the one-workgroup driver of small ensembles (ais_small_kernel.hpp): kabc_ais_advance runs every
    // generation of a call in ONE launch; spec_state / spec_variant then describe THIS kernel
    bool small_ok;
    AisSmallLaunch small;
    int32_t small_pcx;     // the prebuilt table's variant (prior class BOX / NORMAL / GENERAL + posterior kind)
    double* d_strace;      // its device trace buffer
    size_t strace_cap;     // bytes
    ModelUnit* unit;       // run-time compiled unit (user prior families / specialised model), else NULL
    // the model's own kernels (the default, non-blocking specialisation: plugin_registry.hpp)
    int32_t spec_state;    // KABC_SPEC_*
    int32_t spec_variant;  // the AIS variant asked of the unit while KABC_SPEC_PENDING
    int64_t launches;      // half-generation launches so far
    int64_t spec_switch_at;  // launches that ran before the switch, -1
    std::chrono::steady_clock::time_point spec_next_poll;
    double box_lp;
    bool initialised;
    // sample-trace streaming: device chunks filled in rotation by the kernels and
    // drained to the caller's buffer on a copy stream while the next chunks compute
    double* d_trace[kTraceBufs];
    int64_t trace_cap_gens;
    hipStream_t copy_stream;
    hipEvent_t ev_filled[kTraceBufs];
    // prepared-cost words of the launch in flight (ais_aux_kernels.hpp)
    double* d_aux;
    size_t aux_cap;  // bytes
    // debug records (tests)
    int32_t* d_dbg;
    int64_t dbg_cap;  // in int32 units
    // timing
    bool timing;
    int32_t timing_stride;   // launches per event pair
    int32_t open_count;      // launches inside the pair that is open (0 = none open)
    std::vector<hipEvent_t> ev;
    std::vector<int32_t> ev_n;  // launches bracketed by pair i
    size_t ev_used;
    kabc_stats_t last;  // counters at the last kabc_ais_advance return
    // exchange diagnostics of a sharded handle (kabc_ais_exchange_us): per timed half-generation
    // three events on the context stream -- e0 the half's kernels start, e1 they have ended, e2 the
    // gathered half is available to the stream -- and a pair per exchange chunk on the stream the
    // gather runs on
    struct XT {
        hipEvent_t e0, e1, e2;
        hipEvent_t x0[KABC_MAX_EXCHANGE_CHUNKS], x1[KABC_MAX_EXCHANGE_CHUNKS];
        bool closed;
    };
    std::vector<XT> xt;
    size_t xt_used;
    int xt_open;  // entry whose e2 is still to be recorded (pipelined exchange), else -1
};

static constexpr size_t kXtHalves = 128;

static kabc_status_t check_handle(const kabc_ais_t* h) {
    if (!h) {
        set_error("AIS handle is NULL");
        return KABC_ERR_INVALID_ARG;
    }
    return KABC_OK;
}

static kabc_status_t ais_alloc(kabc_ais_t* h, const kabc_model_t* m, void* ext0, void* ext1,
                               const uint64_t* seeds) {
    kabc_ctx_t* ctx = h->ctx;
    const int world = h->world;
    const size_t nch = (size_t)h->nchains;
    KABC_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    if (m->cost.nparams > 0) {
        KABC_HIP_CHECK(dev_malloc(&h->d_cost_params, sizeof(double) * m->cost.nparams));
        KABC_HIP_CHECK(hipMemcpyAsync(h->d_cost_params, m->cost.params,
                                      sizeof(double) * m->cost.nparams, hipMemcpyHostToDevice, s));
    }
    if (m->cost.ndata > 0) {
        KABC_HIP_CHECK(dev_malloc(&h->d_cost_data, sizeof(double) * m->cost.ndata));
        KABC_HIP_CHECK(hipMemcpyAsync(h->d_cost_data, m->cost.data, sizeof(double) * m->cost.ndata,
                                      hipMemcpyHostToDevice, s));
    }
    for (int hf = 0; hf < 2; ++hf) {
        if (h->own_halves) {
            // padded to world equal segments so that the in-place all-gather has one count
            const size_t nb = sizeof(double) * (size_t)(h->cper[hf] * h->xk * world) * h->D * nch;
            KABC_HIP_CHECK(dev_malloc(&h->d_half[hf], nb));
            KABC_HIP_CHECK(hipMemsetAsync(h->d_half[hf], 0, nb, s));
        } else {
            h->d_half[hf] = (double*)(hf == 0 ? ext0 : ext1);
        }
        const size_t nb = sizeof(double) * (size_t)(h->rows_owned[hf] > 0 ? h->rows_owned[hf] : 1) * nch;
        KABC_HIP_CHECK(dev_malloc(&h->d_lp[hf], nb));
        KABC_HIP_CHECK(dev_malloc(&h->d_ll[hf], nb));
    }
    if (seeds) {
        KABC_HIP_CHECK(dev_malloc(&h->d_seeds, sizeof(uint64_t) * nch));
        KABC_HIP_CHECK(hipMemcpyAsync(h->d_seeds, seeds, sizeof(uint64_t) * nch, hipMemcpyHostToDevice, s));
        KABC_HIP_CHECK(dev_malloc(&h->d_chain_retries, sizeof(unsigned long long) * nch));
    }
    KABC_HIP_CHECK(dev_malloc(&h->d_counters, sizeof(DevCounters)));
    KABC_HIP_CHECK(hipMemsetAsync(h->d_counters, 0, sizeof(DevCounters), s));
    if (h->dyn) {
        const size_t D = (size_t)h->D;
        const size_t rmax = (size_t)(h->rows_owned[0] > h->rows_owned[1] ? h->rows_owned[0] : h->rows_owned[1]);
        KABC_HIP_CHECK(dev_malloc(&h->d_prior, sizeof(PriorDev) * D));
        KABC_HIP_CHECK(hipMemcpyAsync(h->d_prior, h->prior_dyn.data(), sizeof(PriorDev) * D,
                                      hipMemcpyHostToDevice, s));
        KABC_HIP_CHECK(dev_malloc(&h->d_raw, sizeof(kabc_prior_t) * D));
        KABC_HIP_CHECK(hipMemcpyAsync(h->d_raw, h->raw_dyn.data(), sizeof(kabc_prior_t) * D,
                                      hipMemcpyHostToDevice, s));
        KABC_HIP_CHECK(dev_malloc(&h->d_scratch, sizeof(double) * (rmax ? rmax : 1) * 2 * D));
    } else {
        KABC_HIP_CHECK(dev_malloc(&h->d_prior, sizeof(PriorSet)));
        KABC_HIP_CHECK(hipMemcpyAsync(h->d_prior, &h->prior, sizeof(PriorSet), hipMemcpyHostToDevice, s));
    }
    KABC_HIP_CHECK(dev_malloc(&h->d_slots, sizeof(unsigned long long) * kCounterSlots * 8));
    KABC_HIP_CHECK(hipMemsetAsync(h->d_slots, 0, sizeof(unsigned long long) * kCounterSlots * 8, s));
    KABC_HIP_CHECK(hipStreamSynchronize(s));
    return KABC_OK;
}

// the half-generation kernel of a handle: the unit's (user prior families; a specialisation that is
// there already), else the prebuilt one of the prior's class
static kabc_status_t ais_resolve_half(kabc_ais_t* h) {
    if (h->dyn) return KABC_OK;
    const bool track = !h->small_ok;  // (a small-ensemble handle's spec_state describes its small kernel)
    AisLaunch fn;
    const int pk_off = kPriorClasses * (h->posterior - 1);
    if (h->unit) {
        int st = KABC_SPEC_NONE;
        PluginKernel uk;
        if (track || unit_required(h->unit)) uk = unit_kernel(h->unit, kPfAis, h->D, h->spec_variant, &st);
        if (uk.mod) fn = AisLaunch(uk.mod, &ais_half_geom, (unsigned)kAisBlock);
        if (track) h->spec_state = st;
        // user families: there are no other kernels (message set by the compilation / load)
        if (!fn && unit_required(h->unit)) return KABC_ERR_DEVICE;
    }
    if (!fn) {  // (no unit, or a specialisation that is not there (yet): the prebuilt kernels)
        int pc = h->pc;
        // (A/B runs: what the GENERAL class costs a SIMPLE / NORMAL prior -- the classes give the same bits)
        if (const char* e = std::getenv("KABC_PREBUILT_CLASS"))
            if (e[0] == 'g' && pc != kPriorBox) pc = kPriorGeneral;
        fn = find_ais_kernel(h->cost_id, h->D, pc + pk_off);
        if (!fn && h->pc == kPriorNormal)  // plugins instantiate SIMPLE only
            fn = find_ais_kernel(h->cost_id, h->D, kPriorSimple + pk_off);
    }
    if (!fn) {
        if (!get_error()[0]) set_error("no gfx950 kernel instantiated for cost id %d, D = %d", h->cost_id, h->D);
        return KABC_ERR_UNSUPPORTED;
    }
    h->launch = fn;
    return KABC_OK;
}

static kabc_status_t ais_create_common(kabc_ctx_t* ctx, const kabc_model_t* m, int64_t n_total,
                                       int32_t rank, int32_t world, uint64_t seed, void* ext0,
                                       void* ext1, kabc_comm_t* comm, kabc_ais_t** out,
                                       int32_t nchains = 1, const uint64_t* seeds = nullptr) {
    if (!ctx || !m || !out || !m->prior) {
        set_error("kabc_ais_create: NULL argument");
        return KABC_ERR_INVALID_ARG;
    }
    if (m->D < 1 || m->D > KABC_MAX_DIM_DYN) {
        set_error("length(prior) = %d is outside the device path's range 1..%d", m->D,
                  KABC_MAX_DIM_DYN);
        return KABC_ERR_UNSUPPORTED;
    }
    // the library-side fields of MvNormal components (device block, D): everything below works
    // on the resolved copy
    std::vector<kabc_prior_t> resolved((size_t)m->D);
    if (kabc_status_t st = resolve_priors(ctx, m->prior, m->D, resolved.data())) return st;
    kabc_model_t mres = *m;
    mres.prior = resolved.data();
    m = &mres;
    const bool dyn = m->D > KABC_MAX_DIM;
    if (dyn && nchains != 1) {
        set_error("length(prior) = %d > %d runs on the run-time-dimension kernels: one chain per handle",
                  m->D, KABC_MAX_DIM);
        return KABC_ERR_UNSUPPORTED;
    }
    // src/KissABC.jl:43-48
    if (n_total < m->D + 5) {
        set_error("nparticles = %lld is insufficient, set number of particles in AIS(⋅) atleast to %d",
                  (long long)n_total, m->D + 5);
        return KABC_ERR_INVALID_ARG;
    }
    if (n_total >= (1ll << 31)) {
        set_error("nparticles must be < 2^31");
        return KABC_ERR_INVALID_ARG;
    }
    // the kernels address a partner row by a 32-bit byte offset into its half (include/kabc.h)
    if ((n_total / 2 + 1 + 64ll * world) * m->D * 8 >= (1ll << 32)) {
        set_error("nparticles = %lld with length(prior) = %d: a half-ensemble must stay below 4 GiB",
                  (long long)n_total, m->D);
        return KABC_ERR_UNSUPPORTED;
    }
    if (m->posterior != KABC_POSTERIOR_KERNELIZED && m->posterior != KABC_POSTERIOR_THRESHOLD &&
        m->posterior != KABC_POSTERIOR_COMMON) {
        set_error("unknown posterior kind %d", m->posterior);
        return KABC_ERR_INVALID_ARG;
    }
    if (world < 1 || rank < 0 || rank >= world ||
        (!comm && world > 1 && n_total % (2 * world) != 0)) {
        set_error("sharded AIS needs nparticles divisible by 2*world (got %lld, world %d)",
                  (long long)n_total, world);
        return KABC_ERR_INVALID_ARG;
    }
    if (!cost_dim_ok_rt(m->cost.id, m->D)) {
        set_error("DeviceCost id %d does not accept D = %d", m->cost.id, m->D);
        return KABC_ERR_UNSUPPORTED;
    }
    {  // sample_init drawn by the cost plugin (KABC_PRIOR_USER_INIT): CommonLogDensity only
        int n_user = 0;
        for (int k = 0; k < m->D; ++k) n_user += m->prior[k].kind == KABC_PRIOR_USER_INIT;
        if (n_user) {
            const CostPlugin* pl = find_plugin(m->cost.id);
            if (n_user != m->D || m->posterior != KABC_POSTERIOR_COMMON || dyn || !pl || !pl->has_sample_init) {
                set_error("KABC_PRIOR_USER_INIT: every component must carry it, the model must be a "
                          "CommonLogDensity of at most %d parameters, and its log-density a user cost "
                          "whose snippet defines KABC_USER_SAMPLE_INIT + kabc_user_sample_init",
                          KABC_MAX_DIM);
                return KABC_ERR_INVALID_ARG;
            }
        }
    }
    // prior class of the half-generation kernel (ais_kernels.hpp)
    bool isbox = true, gaussbox = true, allnormal = true;
    for (int k = 0; k < m->D; ++k) {
        const int kd = m->prior[k].kind;
        const bool box = kd == KABC_PRIOR_UNIFORM || kd == KABC_PRIOR_DISCRETE_UNIFORM;
        const bool gauss = kd == KABC_PRIOR_NORMAL || kd == KABC_PRIOR_TRUNCNORMAL;
        isbox = isbox && box;
        gaussbox = gaussbox && (box || gauss);
        allnormal = allnormal && kd == KABC_PRIOR_NORMAL;
    }
    const int pc = isbox ? kPriorBox : allnormal ? kPriorNormal : gaussbox ? kPriorSimple : kPriorGeneral;
    static_assert(kPriorClasses == 4, "capi_plugin.hip decodes pcx with 4 prior classes");
    // (run-time compiled kernels are loaded on the CURRENT device)
    KABC_HIP_CHECK(hipSetDevice(ctx->device));
    // user prior families among the components, or a specialisation of exactly this model
    // (kabc_compile_model): the kernels of that unit, GENERAL class (plugin_registry.hpp)
    ModelUnit* unit = nullptr;
    if (kabc_status_t st = model_unit_for(m->prior, m->D, m->cost.id, &unit)) return st;
    // length(prior) > KABC_MAX_DIM: the run-time-dimension kernels -- of the unit (user prior families:
    // compiled with their snippets), of the user cost (hipRTC form, or its plugin .so), or the built-in ones
    AisDynLaunch dyn_fn;
    if (dyn) {
        if (unit) {
            const PluginKernel kh = unit_kernel(unit, kPfAisDyn, m->D, 0), ki = unit_kernel(unit, kPfAisDyn, m->D, 1);
            dyn_fn = AisDynLaunch(kh.mod, unit_kernel(unit, kPfAisDyn, m->D, 2).mod, unit_kernel(unit, kPfAisDyn, m->D, 3).mod, ki.mod);
            if (!dyn_fn) return KABC_ERR_DEVICE;  // (message set by the compilation / load)
        } else if (m->cost.id >= KABC_COST_USER) {
            const CostPlugin* pl = find_plugin(m->cost.id);
            if (pl && pl->rtc) {
                const PluginKernel kh = plugin_kernel(pl, kPfAisDyn, m->D, 0), ki = plugin_kernel(pl, kPfAisDyn, m->D, 1);
                dyn_fn = AisDynLaunch(kh.mod, plugin_kernel(pl, kPfAisDyn, m->D, 2).mod, plugin_kernel(pl, kPfAisDyn, m->D, 3).mod, ki.mod);
            } else if (pl && pl->ais_dyn) {
                dyn_fn = AisDynLaunch((AisDynLaunchFn)pl->ais_dyn());
            }
        } else {
            dyn_fn = AisDynLaunch(find_ais_dyn_kernel(m->cost.id));
        }
        if (!dyn_fn) {
            if (!get_error()[0])
                set_error("length(prior) = %d > %d: no run-time-dimension kernels for cost id %d (a plugin .so "
                          "built from older headers?)", m->D, KABC_MAX_DIM, m->cost.id);
            return KABC_ERR_UNSUPPORTED;
        }
    }
    const int spec_variant = kPriorGeneral + kPriorClasses * (m->posterior - 1);
    kabc_ais_t* h = new kabc_ais_t();
    h->ctx = ctx;
    h->D = m->D;
    h->posterior = m->posterior;
    h->cost_id = m->cost.id;
    h->eps = m->eps;
    std::memset(h->raw, 0, sizeof h->raw);
    std::memset(&h->prior, 0, sizeof h->prior);
    h->d_raw = nullptr;
    h->d_scratch = nullptr;
    bool prior_ok = true;
    if (dyn) {
        h->raw_dyn.assign(m->prior, m->prior + m->D);
        h->prior_dyn.resize((size_t)m->D);
        for (int k = 0; k < m->D && prior_ok; ++k)
            prior_ok = prepare_prior(h->raw_dyn[k], h->prior_dyn[k]);
    } else {
        std::memcpy(h->raw, m->prior, sizeof(kabc_prior_t) * m->D);
        prior_ok = prepare_priors(h->raw, h->D, h->prior);
    }
    if (!prior_ok) {
        delete h;
        set_error("invalid prior parameters");
        return KABC_ERR_INVALID_ARG;
    }
    h->unit = unit;
    h->pc = pc;
    h->spec_state = KABC_SPEC_NONE;
    h->spec_variant = spec_variant;
    h->launches = 0;
    h->small_ok = false;
    h->d_strace = nullptr;
    h->strace_cap = 0;
    // Small ensembles (both halves fit one workgroup's LDS, ais_small_kernel.hpp): one workgroup per
    // chain runs every generation of a kabc_ais_advance call in one launch (a cost with a grid-wide
    // pre-pass, ais_aux_kernels.hpp: one pre-pass launch per half for all of the call's sub-steps first).
    // Not for sharded handles, caller-lent halves, the run-time-dimension kernels; KABC_AIS_SMALL=0 keeps
    // the launch per half-generation.
    {
        const char* e = std::getenv("KABC_AIS_SMALL");
        const bool off = e && e[0] == '0';
        if (!off && !dyn && !comm && world == 1 && ext0 == nullptr &&
            (n_total + 1) / 2 <= (int64_t)ais_small_rmax(m->D)) {
            const int pk_off = kPriorClasses * (m->posterior - 1);
            // prebuilt classes of the small kernel: BOX, NORMAL up to kAisInstSplit parameters (the two
            // the default path never specialises), GENERAL for everything else -- same bits
            const bool plug = m->cost.id >= KABC_COST_USER;
            const int spc = isbox ? kPriorBox : (allnormal && m->D <= kAisInstSplit && !plug) ? kPriorNormal : kPriorGeneral;
            h->small_pcx = spc + pk_off;
            AisSmallLaunch sfn;
            int st = KABC_SPEC_NONE;
            if (unit) {
                const PluginKernel uk = unit_kernel(unit, kPfAisSmall, m->D, spec_variant, &st);
                if (uk.mod) sfn = AisSmallLaunch(uk.mod, &ais_small_geom, (unsigned)kAisSmallBlock);
            }
            if (!sfn && !(unit && unit_required(unit))) sfn = find_ais_small_kernel(m->cost.id, m->D, h->small_pcx);
            if (sfn) {
                h->small = sfn;
                h->small_ok = true;
                h->spec_state = st;
            } else {
                set_error("%s", "");  // (no small kernel for this model: the launch per half-generation serves)
            }
        }
    }
    h->dyn = dyn_fn;
    if (!h->small_ok || (m->cost.id < KABC_COST_USER && !(unit && unit_required(unit)))) {
        // (a small-ensemble handle of a user cost / user prior families compiles its half-generation
        // kernel only when somebody asks for one: kabc_ais_half_generation)
        if (kabc_status_t st = ais_resolve_half(h)) {
            delete h;
            return st;
        }
    }
    h->spec_switch_at = h->spec_state == KABC_SPEC_ACTIVE ? 0 : -1;
    // BOX class: logpdf inside the box = c0_1 + ... + c0_D, summed left to right
    // exactly as logpdf(d::Factored, x) does (src/priors.jl:30-36)
    h->box_lp = h->prior.c[0].c0;
    for (int k = 1; k < h->D && !dyn; ++k) h->box_lp += h->prior.c[k].c0;
    h->N = n_total;
    h->rows[0] = (n_total + 1) / 2;
    h->rows[1] = n_total / 2;
    h->id_base[0] = 0;
    h->id_base[1] = (uint32_t)h->rows[0];
    h->rank = rank;
    h->world = world;
    h->comm = comm;
    h->nchains = nchains;
    h->d_seeds = nullptr;
    h->d_chain_retries = nullptr;
    // exchange chunks: KABC_EXCHANGE_CHUNKS, else one chunk per residency wave of the half-
    // generation kernel (512 workgroups of 64 walkers fill the 256 CUs; a launch below that
    // takes as long as a full one -- tools/occupancy_probe.py -- so finer chunks would only
    // serialise the compute they are meant to overlap)
    h->xk = 1;
    if (comm) {
        const int64_t per_rank = (h->rows[0] + world - 1) / world;
        int64_t k = world > 1 ? per_rank / (512 * (int64_t)kBatch) : 1;
        if (const char* e = std::getenv("KABC_EXCHANGE_CHUNKS")) k = std::atol(e);
        h->xk = (int32_t)(k < 1 ? 1 : (k > KABC_MAX_EXCHANGE_CHUNKS ? KABC_MAX_EXCHANGE_CHUNKS : k));
    }
    for (int hf = 0; hf < 2; ++hf) {
        // caller-lent buffers: equal shards (n_total % (2 world) == 0 was checked); library-
        // owned exchange: ceil shards, the last segments may hold fewer rows or none
        const int64_t parts = (int64_t)world * h->xk;
        const int64_t cper = comm ? (h->rows[hf] + parts - 1) / parts : h->rows[hf] / world;
        h->cper[hf] = cper;
        h->rows_owned[hf] = 0;
        for (int k = 0; k < h->xk; ++k) {
            int64_t lo = cper * ((int64_t)k * world + rank), hi = lo + cper;
            lo = lo < h->rows[hf] ? lo : h->rows[hf];
            hi = hi < h->rows[hf] ? hi : h->rows[hf];
            h->seg[hf].push_back({lo, hi - lo, h->rows_owned[hf]});
            h->rows_owned[hf] += hi - lo;
        }
    }
    h->seed = seed;
    h->t = 0;
    h->initialised = false;
    h->trace_cap_gens = 0;
    h->copy_stream = nullptr;
    for (int b = 0; b < kTraceBufs; ++b) {
        h->d_trace[b] = nullptr;
        h->ev_filled[b] = nullptr;
    }
    h->d_dbg = nullptr;
    h->dbg_cap = 0;
    h->d_aux = nullptr;
    h->aux_cap = 0;
    h->timing = false;
    h->timing_stride = 1;
    h->open_count = 0;
    h->ev_used = 0;
    h->last = kabc_stats_t{0, 0, 0};
    h->xt_used = 0;
    h->xt_open = -1;
    h->d_cost_params = h->d_cost_data = nullptr;
    h->cost_ndata = m->cost.ndata;
    h->own_halves = (ext0 == nullptr);

    // every early return below releases what was allocated so far
    if (kabc_status_t st = ais_alloc(h, m, ext0, ext1, seeds)) {
        (void)kabc_ais_destroy(h);
        return st;
    }
    *out = h;
    return KABC_OK;
}

static kabc_status_t read_counters(kabc_ais_t* h, DevCounters* c) {
    static thread_local std::vector<unsigned long long> slots(kCounterSlots * 8);
    KABC_HIP_CHECK(hipMemcpyAsync(c, h->d_counters, sizeof(DevCounters), hipMemcpyDeviceToHost,
                                  h->ctx->stream));
    KABC_HIP_CHECK(hipMemcpyAsync(slots.data(), h->d_slots,
                                  sizeof(unsigned long long) * kCounterSlots * 8,
                                  hipMemcpyDeviceToHost, h->ctx->stream));
    KABC_HIP_CHECK(hipStreamSynchronize(h->ctx->stream));
    for (int i = 0; i < kCounterSlots; ++i) {
        c->proposals += slots[i * 8 + 0];
        c->cost_evals += slots[i * 8 + 1];
        c->accepted += slots[i * 8 + 2];
    }
    return KABC_OK;
}

static kabc_status_t check_device_error(kabc_ais_t* h, const DevCounters& c) {
    (void)h;
    if (c.error == 1) {
        set_error("ld_correction is invalid");  // src/types.jl:69
        return KABC_ERR_INVALID_STATE;
    }
    if (c.error == 2) {
        set_error("starting sample invalid.");  // src/types.jl:70
        return KABC_ERR_INVALID_STATE;
    }
    return KABC_OK;
}

extern "C" {

kabc_status_t kabc_ais_create(kabc_ctx_t* ctx, const kabc_model_t* model, int64_t nparticles,
                              uint64_t seed, kabc_ais_t** out) {
    return ais_create_common(ctx, model, nparticles, 0, 1, seed, nullptr, nullptr, nullptr, out);
}

kabc_status_t kabc_ais_create_batch(kabc_ctx_t* ctx, const kabc_model_t* model, int64_t nparticles,
                                    int32_t nchains, const uint64_t* seeds, kabc_ais_t** out) {
    if (nchains < 1 || nchains > 65535 || !seeds) {
        set_error("kabc_ais_create_batch: nchains must be 1..65535 and seeds non-NULL");
        return KABC_ERR_INVALID_ARG;
    }
    return ais_create_common(ctx, model, nparticles, 0, 1, seeds[0], nullptr, nullptr, nullptr, out,
                             nchains, seeds);
}

kabc_status_t kabc_ais_create_sharded(kabc_ctx_t* ctx, const kabc_model_t* model, int64_t n_total,
                                      int32_t rank, int32_t world, uint64_t seed, void* dev_half0,
                                      void* dev_half1, kabc_ais_t** out) {
    if (!dev_half0 || !dev_half1) {
        set_error("kabc_ais_create_sharded: device half buffers must be provided");
        return KABC_ERR_INVALID_ARG;
    }
    return ais_create_common(ctx, model, n_total, rank, world, seed, dev_half0, dev_half1, nullptr,
                             out);
}

kabc_status_t kabc_ais_create_dist(kabc_comm_t* comm, const kabc_model_t* model,
                                   int64_t nparticles, uint64_t seed, kabc_ais_t** out) {
    if (!comm) {
        set_error("kabc_ais_create_dist: communicator is NULL");
        return KABC_ERR_INVALID_ARG;
    }
    return ais_create_common(comm->ctx, model, nparticles, comm->rank, comm->world, seed, nullptr,
                             nullptr, comm, out);
}

// exchange chunk k of a half: [world][cper][D] doubles, this rank's segment at rank * cper
static double* chunk_base(kabc_ais_t* h, int half, int k) {
    return h->d_half[half] + (size_t)k * h->world * h->cper[half] * h->D;
}

static AisDynArgs dyn_args(kabc_ais_t* h, int half, const kabc_ais::Seg& sg) {
    AisDynArgs a;
    std::memset(&a, 0, sizeof a);
    a.x_act = h->d_half[half];
    a.x_comp = h->d_half[1 - half];
    a.lp = h->d_lp[half] + sg.off;
    a.ll = h->d_ll[half] + sg.off;
    a.scratch = h->d_scratch + sg.off * 2 * h->D;
    a.counters = h->d_counters;
    a.slots = h->d_slots;
    a.cost_params = h->d_cost_params;
    a.cost_data = h->d_cost_data;
    a.cost_ndata = h->cost_ndata;
    a.row_first = sg.first;
    a.rows_owned = sg.count;
    a.n_comp = h->rows[1 - half];
    a.seed = h->seed;
    a.id_base = h->id_base[half];
    a.posterior = h->posterior;
    a.cost_id = h->cost_id;
    a.D = h->D;
    a.eps = h->eps;
    a.reps = (h->posterior == KABC_POSTERIOR_COMMON) ? 1.0 : 1.0 / h->eps;
    a.prior = h->d_prior;
    a.raw = h->d_raw;
    return a;
}

// step(init) of one handle, enqueued on its stream (no read-back)
static kabc_status_t ais_init_enqueue(kabc_ais_t* h, int32_t retry_sampling) {
    if (check_handle(h)) return KABC_ERR_INVALID_ARG;
    if (retry_sampling < 0) {
        set_error("retry_sampling must be >= 0");
        return KABC_ERR_INVALID_ARG;
    }
    KABC_HIP_CHECK(hipSetDevice(h->ctx->device));
    hipStream_t s = h->ctx->stream;
    KABC_HIP_CHECK(hipMemsetAsync(h->d_counters, 0, sizeof(DevCounters), s));
    KABC_HIP_CHECK(hipMemsetAsync(h->d_slots, 0, sizeof(unsigned long long) * kCounterSlots * 8, s));
    if (h->d_chain_retries)
        KABC_HIP_CHECK(hipMemsetAsync(h->d_chain_retries, 0, sizeof(unsigned long long) * h->nchains, s));
    // the budget is per ensemble (src/KissABC.jl:52); a shard gets its share
    const unsigned long long budget = (unsigned long long)retry_sampling *
                                      (unsigned long long)(h->rows_owned[0] + h->rows_owned[1]);
    for (int hf = 0; hf < 2; ++hf) {
        for (const kabc_ais::Seg& sg : h->seg[hf]) {
            if (sg.count == 0) continue;
            if (h->dyn) {
                AisDynArgs a = dyn_args(h, hf, sg);
                a.retry_budget = budget;
                h->dyn(a, s, 1);
                KABC_HIP_CHECK(hipGetLastError());
                continue;
            }
            InitArgs a;
            std::memset(&a, 0, sizeof a);
            a.x_act = h->d_half[hf];
            a.lp = h->d_lp[hf] + sg.off;
            a.ll = h->d_ll[hf] + sg.off;
            a.counters = h->d_counters;
            a.cost_params = h->d_cost_params;
            a.cost_data = h->d_cost_data;
            a.cost_ndata = h->cost_ndata;
            a.row_first = sg.first;
            a.rows_owned = sg.count;
            a.seed = h->seed;
            a.id_base = h->id_base[hf];
            a.posterior = h->posterior;
            a.cost_id = h->cost_id;
            a.eps = h->eps;
            a.retry_budget = budget;
            a.prior = h->prior;
            std::memcpy(a.raw, h->raw, sizeof a.raw);
            a.seeds = h->d_seeds;
            a.chain_retries = h->d_chain_retries;
            a.stride_act = h->rows[hf] * h->D;
            a.stride_own = h->rows_owned[hf];
            if (!launch_ais_init(h->D, a, s, (unsigned)h->nchains, h->unit)) {
                if (!get_error()[0]) set_error("no init kernel for cost id %d, D = %d", h->cost_id, h->D);
                return KABC_ERR_DEVICE;
            }
            KABC_HIP_CHECK(hipGetLastError());
        }
    }
    return KABC_OK;
}

static kabc_status_t ais_init_failed() {
    // src/KissABC.jl:58-59
    set_error("Prior leads to ∞ costs too often, tune the prior or increase `retry_sampling`.");
    return KABC_ERR_RETRY_EXHAUSTED;
}

static void ais_mark_initialised(kabc_ais_t* h) {
    h->t = 0;
    h->initialised = true;
    h->last = kabc_stats_t{0, 0, 0};
}

kabc_status_t kabc_ais_init(kabc_ais_t* h, int32_t retry_sampling) {
    if (check_handle(h)) return KABC_ERR_INVALID_ARG;
    if (h->comm && h->comm->single_process) {
        set_error("this handle belongs to a single-process group: use kabc_ais_init_multi");
        return KABC_ERR_INVALID_ARG;
    }
    // A rank-local failure must not strand the other ranks in the collectives below: every rank
    // takes part in them whatever happened locally, and all of them learn about it.
    kabc_status_t local = ais_init_enqueue(h, retry_sampling);
    char local_msg[512] = "";
    if (local) std::snprintf(local_msg, sizeof local_msg, "%s", get_error());
    if (local && !h->comm) return local;
    DevCounters c;
    std::memset(&c, 0, sizeof c);
    if (local == KABC_OK && read_counters(h, &c)) {
        local = KABC_ERR_DEVICE;
        std::snprintf(local_msg, sizeof local_msg, "%s", get_error());
        if (!h->comm) return local;
    }
    uint64_t failed = c.init_failed ? 1u : 0u;
    if (h->comm) {
        // every rank reports the failure of any (the reference's retry budget is per ensemble)
        for (int hf = 0; hf < 2; ++hf)
            for (int k = 0; k < h->xk; ++k)
                if (kabc_status_t st = comm_allgather_inplace(h->comm, chunk_base(h, hf, k),
                                                              (size_t)h->cper[hf] * h->D))
                    if (local == KABC_OK) {
                        local = st;
                        std::snprintf(local_msg, sizeof local_msg, "%s", get_error());
                    }
        uint64_t v[2] = {failed, local != KABC_OK ? 1u : 0u};
        const kabc_status_t st = kabc_comm_allreduce_sum_u64(h->comm, v, 2);
        if (local) {
            set_error("%s", local_msg);
            return local;
        }
        if (st) return st;
        if (v[1]) {
            set_error("kabc_ais_init failed on %llu other rank(s) of the communicator",
                      (unsigned long long)v[1]);
            return KABC_ERR_DEVICE;
        }
        failed = v[0];
    }
    if (failed) return ais_init_failed();
    ais_mark_initialised(h);
    return KABC_OK;
}

static kabc_status_t check_group(kabc_ais_t** hs, int32_t n, const char* who) {
    if (!hs || n < 1 || n > KABC_COMM_MAX_WORLD) {
        set_error("%s: bad handle array", who);
        return KABC_ERR_INVALID_ARG;
    }
    unsigned seen = 0;
    for (int i = 0; i < n; ++i) {
        kabc_ais_t* h = hs[i];
        if (!h || !h->comm || !h->comm->single_process || h->comm->world != n ||
            h->comm->backend != hs[0]->comm->backend || h->comm->grp != hs[0]->comm->grp ||
            h->N != hs[0]->N || h->D != hs[0]->D || h->seed != hs[0]->seed || h->xk != hs[0]->xk) {
            set_error("%s: the handles must be the %d shards of one ensemble, created with "
                      "kabc_ais_create_dist on the communicators of one kabc_comm_init_all call",
                      who, n);
            return KABC_ERR_INVALID_ARG;
        }
        seen |= 1u << h->comm->rank;
    }
    if (seen != (n >= 32 ? ~0u : (1u << n) - 1u)) {
        set_error("%s: every rank must be present exactly once", who);
        return KABC_ERR_INVALID_ARG;
    }
    return KABC_OK;
}

// all-gather of exchange chunk k of a half on the context streams (no pipelining)
static kabc_status_t gather_multi(kabc_ais_t** hs, int32_t n, int half, int k) {
    kabc_comm_t* comms[KABC_COMM_MAX_WORLD];
    double* bases[KABC_COMM_MAX_WORLD];
    for (int i = 0; i < n; ++i) {
        comms[i] = hs[i]->comm;
        bases[i] = chunk_base(hs[i], half, k);
    }
    return comm_allgather_inplace_multi(comms, bases, n, (size_t)hs[0]->cper[half] * hs[0]->D);
}

kabc_status_t kabc_ais_init_multi(kabc_ais_t** hs, int32_t n, int32_t retry_sampling) {
    if (kabc_status_t st = check_group(hs, n, "kabc_ais_init_multi")) return st;
    for (int i = 0; i < n; ++i)
        if (kabc_status_t st = ais_init_enqueue(hs[i], retry_sampling)) return st;
    for (int hf = 0; hf < 2; ++hf)
        for (int k = 0; k < hs[0]->xk; ++k)
            if (kabc_status_t st = gather_multi(hs, n, hf, k)) return st;
    bool failed = false;
    for (int i = 0; i < n; ++i) {
        KABC_HIP_CHECK(hipSetDevice(hs[i]->ctx->device));
        DevCounters c;
        if (read_counters(hs[i], &c)) return KABC_ERR_DEVICE;
        failed = failed || c.init_failed;
    }
    if (failed) return ais_init_failed();
    for (int i = 0; i < n; ++i) ais_mark_initialised(hs[i]);
    return KABC_OK;
}

// ends the open event pair (if any) behind the last launch: a pair never spans a point where the
// host synchronises or leaves the library, so the gaps between calls stay out of the figures
static kabc_status_t timing_close_pair(kabc_ais_t* h) {
    if (h->open_count > 0) {
        KABC_HIP_CHECK(hipEventRecord(h->ev[h->ev_used + 1], h->ctx->stream));
        h->ev_n[h->ev_used / 2] = h->open_count;
        h->ev_used += 2;
        h->open_count = 0;
    }
    return KABC_OK;
}

// A handle that started on the prebuilt kernels while the worker compiles the model's own: look
// for them (a map lookup; a stat() at most every 2 ms) and switch.  Same bits either way.
static void ais_poll_spec(kabc_ais_t* h) {
    const auto now = std::chrono::steady_clock::now();
    if (now < h->spec_next_poll) return;
    h->spec_next_poll = now + std::chrono::milliseconds(2);
    int st = KABC_SPEC_NONE;
    const PluginKernel k = unit_kernel(h->unit, h->small_ok ? kPfAisSmall : kPfAis, h->D, h->spec_variant, &st);
    if (st == KABC_SPEC_ACTIVE && k.mod) {
        if (h->small_ok) h->small = AisSmallLaunch(k.mod, &ais_small_geom, (unsigned)kAisSmallBlock);
        else h->launch = AisLaunch(k.mod, &ais_half_geom, (unsigned)kAisBlock);
        h->spec_state = KABC_SPEC_ACTIVE;
        h->spec_switch_at = h->launches;
    } else if (st == KABC_SPEC_FAILED) {
        h->spec_state = KABC_SPEC_FAILED;
    }
}

// one launch: `ntransitions` transitions for the owned rows of segment `sg` of `half`
static kabc_status_t launch_half_seg(kabc_ais_t* h, int32_t half, const kabc_ais::Seg& sg,
                                     int32_t ntransitions, double* dev_trace_rows) {
    if (sg.count == 0) return KABC_OK;
    hipStream_t s = h->ctx->stream;
    if (h->spec_state == KABC_SPEC_PENDING && !h->small_ok) ais_poll_spec(h);
    if (!h->launch && !h->dyn)
        if (kabc_status_t st = ais_resolve_half(h)) return st;
    h->launches++;
    // debug records: layout [N_owned][nt][6] in the order of the owned rows (half 0 first)
    int32_t* dbg = nullptr;
    if (h->d_dbg) {
        const int64_t off = ((half == 0 ? 0 : h->rows_owned[0]) + sg.off) * (int64_t)ntransitions * 6;
        if (off + sg.count * (int64_t)ntransitions * 6 <= h->dbg_cap) dbg = h->d_dbg + off;
    }
    // timing: one hipEvent pair brackets `timing_stride` consecutive launches (the
    // marker packets cost ~3 us per pair; amortised over the group they stop
    // inflating the per-kernel figure)
    const bool t_on = h->timing && (h->ev_used + 2 <= h->ev.size());
    if (t_on && h->open_count == 0) KABC_HIP_CHECK(hipEventRecord(h->ev[h->ev_used], s));
    if (h->dyn) {
        AisDynArgs a = dyn_args(h, half, sg);
        a.trace = dev_trace_rows ? dev_trace_rows + sg.off * h->D : nullptr;
        a.dbg = dbg;
        a.t0 = h->t;
        a.nt = ntransitions;
        h->dyn(a, s, 0);
    } else {
        AisArgs a;
        std::memset(&a, 0, sizeof a);
        a.x_act = h->d_half[half];
        a.x_comp = h->d_half[1 - half];
        a.lp = h->d_lp[half] + sg.off;
        a.ll = h->d_ll[half] + sg.off;
        a.dbg = dbg;
        a.dbg_nt = ntransitions;
        a.counters = h->d_counters;
        a.slots = h->d_slots;
        a.cost_params = h->d_cost_params;
        a.cost_data = h->d_cost_data;
        a.cost_ndata = h->cost_ndata;
        a.row_first = sg.first;
        a.rows_owned = sg.count;
        a.n_comp = h->rows[1 - half];
        a.seed = h->seed;
        a.id_base = h->id_base[half];
        a.posterior = h->posterior;
        a.eps = h->eps;
        a.reps = (h->posterior == KABC_POSTERIOR_COMMON) ? 1.0 : 1.0 / h->eps;
        a.box_lp = h->box_lp;
        a.prior = h->d_prior;
        a.seeds = h->d_seeds;
        a.stride_act = h->rows[half] * h->D;
        a.stride_comp = h->rows[1 - half] * h->D;
        a.stride_own = h->rows_owned[half];
        a.stride_trace = h->N * h->D;
        {
            static const int ab = [] {
                const char* e = getenv("KABC_ABLATE");
                const int v = e ? atoi(e) : 0;
#ifndef KABC_PROBES
                if (v) fprintf(stderr, "[kabc] KABC_ABLATE=%d ignored: the timing probes are compiled "
                                       "only into libkabc_hip_probes.so (make PROBES=1, KABC_PROBES=1)\n", v);
#endif
                return v;
            }();
            a.ablate = ab;
        }
        // A prepared built-in cost: its parameter-independent words for every (sub-step, row) of
        // the launch come from a grid-wide pre-pass, one wavefront per cost evaluation
        // (ais_aux_kernels.hpp).  The buffer is bounded: beyond it the launch is cut into blocks
        // of sub-steps -- the state lives in memory between launches, so the result is the same.
        const int W = aux_prepass_words(h->cost_id);
        int32_t blk = ntransitions;
        if (W) {
            const size_t per_step = sizeof(double) * (size_t)W * (size_t)sg.count * (size_t)h->nchains;
            size_t cap = (size_t)256 << 20;
            if (const char* e = std::getenv("KABC_AUX_KIB")) {  // tests: force the block path
                const long kib = std::atol(e);
                if (kib > 0) cap = (size_t)kib << 10;
            }
            const size_t fit = cap / per_step;
            blk = (int32_t)(fit < 1 ? 1 : (fit > (size_t)ntransitions ? (size_t)ntransitions : fit));
            if (per_step * (size_t)blk > h->aux_cap) {
                if (h->d_aux) {
                    KABC_HIP_CHECK(hipStreamSynchronize(s));
                    KABC_HIP_CHECK(hipFree(h->d_aux));
                    h->d_aux = nullptr;
                    h->aux_cap = 0;
                }
                KABC_HIP_CHECK(dev_malloc(&h->d_aux, per_step * (size_t)blk));
                h->aux_cap = per_step * (size_t)blk;
            }
        }
        for (int32_t s0 = 0; s0 < ntransitions; s0 += blk) {
            const int32_t nb = ntransitions - s0 < blk ? ntransitions - s0 : blk;
            a.t0 = h->t + (uint64_t)s0;
            a.nt = nb;
            a.dbg_s0 = s0;
            // push_p(x) after the LAST transition is the sample step() returns (src/KissABC.jl:78)
            a.trace = (dev_trace_rows && s0 + nb == ntransitions) ? dev_trace_rows + sg.off * h->D : nullptr;
            if (W) {
                AuxArgs x;
                std::memset(&x, 0, sizeof x);
                x.aux = h->d_aux;
                x.cost_params = h->d_cost_params;
                x.cost_data = h->d_cost_data;
                x.cost_ndata = h->cost_ndata;
                x.row_first = sg.first;
                x.rows = sg.count;
                x.seed = h->seed;
                x.t0 = a.t0;
                x.id_base = h->id_base[half];
                x.nt = nb;
                x.seeds = h->d_seeds;
                x.stride_aux = (int64_t)nb * W * sg.count;
                launch_aux_prepass(h->cost_id, x, s, (unsigned)h->nchains);
                a.aux = h->d_aux;
                a.stride_aux = x.stride_aux;
            }
            h->launch(a, s, (unsigned)h->nchains);
        }
    }
    if (t_on && ++h->open_count >= h->timing_stride)
        if (kabc_status_t st = timing_close_pair(h)) return st;
    KABC_HIP_CHECK(hipGetLastError());
    return KABC_OK;
}

kabc_status_t kabc_ais_half_generation(kabc_ais_t* h, int32_t half, int32_t ntransitions,
                                       void* dev_trace_rows) {
    if (check_handle(h)) return KABC_ERR_INVALID_ARG;
    if (!h->initialised) {
        set_error("kabc_ais_init / kabc_ais_set_state has not been called");
        return KABC_ERR_INVALID_STATE;
    }
    if ((half != 0 && half != 1) || ntransitions < 1) {
        set_error("half must be 0/1 and ntransitions >= 1");
        return KABC_ERR_INVALID_ARG;
    }
    for (const kabc_ais::Seg& sg : h->seg[half])
        if (kabc_status_t st = launch_half_seg(h, half, sg, ntransitions, (double*)dev_trace_rows))
            return st;
    return KABC_OK;
}

kabc_status_t kabc_ais_end_generation(kabc_ais_t* h, int32_t ntransitions) {
    if (check_handle(h)) return KABC_ERR_INVALID_ARG;
    h->t += (uint64_t)ntransitions;
    return KABC_OK;
}

}  // extern "C"

// units (batch, sub-step) of one generation a consumer of the small kernel may have to count
static int64_t ais_small_units_per_gen(const kabc_ais_t* h, int32_t ntransitions) {
    return ((h->rows[0] + kBatch - 1) / kBatch + (h->rows[1] + kBatch - 1) / kBatch) * (int64_t)ntransitions;
}

// `ngenerations` generations of a small ensemble: one launch of one workgroup per chain
// (ais_small_kernel.hpp) per block of generations -- a block ends where the device trace buffer
// (64 MiB) or the kernel's 32-bit unit counters would.  Enqueued on the handle's stream; the trace
// block is copied to out_samples behind its launch.
static kabc_status_t ais_small_run(kabc_ais_t* h, int64_t ngenerations, int32_t ntransitions, double* out_samples) {
    hipStream_t s = h->ctx->stream;
    const int64_t gen_elems = h->N * h->D * h->nchains;  // [chain][N][D] per generation
    const size_t gen_bytes = sizeof(double) * (size_t)gen_elems;
    int64_t block = (1ll << 30) / ais_small_units_per_gen(h, ntransitions);
    if (out_samples) {
        size_t target = (size_t)64 << 20;
        if (const char* e = std::getenv("KABC_TRACE_CHUNK_MIB")) {  // tuning / tests: force several blocks
            const long mib = std::atol(e);
            if (mib > 0) target = (size_t)mib << 20;
        }
        const int64_t fit = (int64_t)(target / gen_bytes);
        if (fit < block) block = fit;
    }
    // a prepared cost's words for every sub-step of a block, from the grid-wide pre-pass: W per (walker,
    // sub-step), bounded like the half-generation path's buffer (KABC_AUX_KIB)
    const int auxW = aux_prepass_words(h->cost_id);
    const size_t aux_gen_bytes = sizeof(double) * (size_t)auxW * (size_t)h->N * (size_t)h->nchains * (size_t)ntransitions;
    if (auxW) {
        size_t cap = (size_t)256 << 20;
        if (const char* e = std::getenv("KABC_AUX_KIB")) {  // tests: force several blocks
            const long kib = std::atol(e);
            if (kib > 0) cap = (size_t)kib << 10;
        }
        const int64_t fit = (int64_t)(cap / aux_gen_bytes);
        if (fit < block) block = fit;
    }
    if (block < 1) block = 1;
    if (block > ngenerations) block = ngenerations;
    if (auxW && aux_gen_bytes * (size_t)block > h->aux_cap) {
        if (h->d_aux) {
            KABC_HIP_CHECK(hipStreamSynchronize(s));
            KABC_HIP_CHECK(hipFree(h->d_aux));
            h->d_aux = nullptr;
            h->aux_cap = 0;
        }
        KABC_HIP_CHECK(dev_malloc(&h->d_aux, aux_gen_bytes * (size_t)block));
        h->aux_cap = aux_gen_bytes * (size_t)block;
    }
    if (out_samples && gen_bytes * (size_t)block > h->strace_cap) {
        if (h->d_strace) {
            KABC_HIP_CHECK(hipStreamSynchronize(s));
            KABC_HIP_CHECK(hipFree(h->d_strace));
            h->d_strace = nullptr;
            h->strace_cap = 0;
        }
        KABC_HIP_CHECK(dev_malloc(&h->d_strace, gen_bytes * (size_t)block));
        h->strace_cap = gen_bytes * (size_t)block;
    }
    for (int64_t g0 = 0; g0 < ngenerations; g0 += block) {
        const int64_t gc = ngenerations - g0 < block ? ngenerations - g0 : block;
        if (h->spec_state == KABC_SPEC_PENDING) ais_poll_spec(h);
        h->launches++;
        AisSmallArgs a;
        std::memset(&a, 0, sizeof a);
        for (int hf = 0; hf < 2; ++hf) {
            a.x[hf] = h->d_half[hf];
            a.lp[hf] = h->d_lp[hf];
            a.ll[hf] = h->d_ll[hf];
            a.rows[hf] = (int32_t)h->rows[hf];
            a.id_base[hf] = h->id_base[hf];
        }
        a.trace = out_samples ? h->d_strace : nullptr;
        // debug records: [N][nt][6] in walker order (those of the block's last generation remain)
        a.dbg = (h->d_dbg && h->N * (int64_t)ntransitions * 6 <= h->dbg_cap) ? h->d_dbg : nullptr;
        a.counters = h->d_counters;
        a.slots = h->d_slots;
        a.cost_params = h->d_cost_params;
        a.cost_data = h->d_cost_data;
        a.cost_ndata = h->cost_ndata;
        a.seed = h->seed;
        a.t0 = h->t;
        a.nt = ntransitions;
        a.ngen = (int32_t)gc;
        a.trace_from = 0;
        a.nchains = h->nchains;
        a.eps = h->eps;
        a.reps = (h->posterior == KABC_POSTERIOR_COMMON) ? 1.0 : 1.0 / h->eps;
        a.box_lp = h->box_lp;
        a.prior = h->d_prior;
        a.seeds = h->d_seeds;
        const bool t_on = h->timing && (h->ev_used + 2 <= h->ev.size());
        if (t_on && h->open_count == 0) KABC_HIP_CHECK(hipEventRecord(h->ev[h->ev_used], s));
        if (auxW) {  // (inside the timed region, like the half-generation path's pre-pass)
            double* base = reinterpret_cast<double*>(h->d_aux);
            for (int hf = 0; hf < 2; ++hf) {
                AuxArgs x;
                std::memset(&x, 0, sizeof x);
                x.aux = base;
                x.cost_params = h->d_cost_params;
                x.cost_data = h->d_cost_data;
                x.cost_ndata = h->cost_ndata;
                x.row_first = 0;
                x.rows = h->rows[hf];
                x.seed = h->seed;
                x.t0 = h->t;
                x.id_base = h->id_base[hf];
                x.nt = (int32_t)(gc * (int64_t)ntransitions);
                x.seeds = h->d_seeds;
                x.stride_aux = (int64_t)x.nt * auxW * h->rows[hf];
                launch_aux_prepass(h->cost_id, x, s, (unsigned)h->nchains);
                a.aux[hf] = base;
                a.stride_aux[hf] = x.stride_aux;
                base += (size_t)x.stride_aux * (size_t)h->nchains;
            }
        }
        h->small(a, s);
        if (t_on && ++h->open_count >= h->timing_stride)
            if (kabc_status_t st = timing_close_pair(h)) return st;
        KABC_HIP_CHECK(hipGetLastError());
        if (out_samples)
            KABC_HIP_CHECK(hipMemcpyAsync(out_samples + g0 * gen_elems, h->d_strace, gen_bytes * (size_t)gc,
                                          hipMemcpyDeviceToHost, s));
        h->t += (uint64_t)gc * (uint64_t)ntransitions;
    }
    return KABC_OK;
}

extern "C" {

kabc_status_t kabc_ais_advance(kabc_ais_t* h, int64_t ngenerations, int32_t ntransitions,
                               double* out_samples, kabc_stats_t* stats) {
    if (check_handle(h)) return KABC_ERR_INVALID_ARG;
    if (h->world != 1 && !h->comm) {
        set_error("kabc_ais_advance drives single-process handles and kabc_ais_create_dist handles; "
                  "handles on caller-lent buffers are driven with kabc_ais_half_generation + the "
                  "caller's all-gather per half");
        return KABC_ERR_INVALID_ARG;
    }
    if (h->comm && h->comm->single_process) {
        set_error("this handle belongs to a single-process group: use kabc_ais_advance_multi");
        return KABC_ERR_INVALID_ARG;
    }
    if (h->comm && out_samples) {
        set_error("a sharded ensemble has no streamed trace: read it with kabc_ais_get_ensemble");
        return KABC_ERR_INVALID_ARG;
    }
    if (ngenerations < 0 || ntransitions < 1) {
        set_error("ngenerations must be >= 0 and ntransitions >= 1");
        return KABC_ERR_INVALID_ARG;
    }
    if (!h->initialised) {
        set_error("kabc_ais_init / kabc_ais_set_state has not been called");
        return KABC_ERR_INVALID_STATE;
    }
    KABC_HIP_CHECK(hipSetDevice(h->ctx->device));
    hipStream_t s = h->ctx->stream;
    const int64_t gen_elems = h->N * h->D * h->nchains;  // [chain][N][D] per generation
    const size_t gen_bytes = sizeof(double) * (size_t)gen_elems;
    // A rank-local failure (a launch that did not go out) must not leave the other ranks
    // blocked in a collective this rank never joins: the kernels stop, the exchanges of the
    // remaining half-generations are still issued, and the ranks agree on the outcome below.
    kabc_status_t local_err = KABC_OK;
    char local_msg[512] = "";
    auto keep = [&](kabc_status_t st) {
        if (st != KABC_OK && local_err == KABC_OK) {
            local_err = st;
            std::snprintf(local_msg, sizeof local_msg, "%s", get_error());
        }
    };
    if (h->small_ok && ngenerations > 0 && ais_small_units_per_gen(h, ntransitions) <= (1ll << 30)) {
        if (kabc_status_t st = ais_small_run(h, ngenerations, ntransitions, out_samples)) return st;
    } else if (!out_samples || ngenerations == 0) {
        for (int64_t g = 0; g < ngenerations; ++g) {
            for (int hf = 0; hf < 2; ++hf) {
                // exchange diagnostics: this half-generation is timed while entries are left
                kabc_ais::XT* xt = (h->comm && h->timing && h->xt_used < h->xt.size()) ? &h->xt[h->xt_used] : nullptr;
                if (!h->comm || h->xk == 1) {
                    if (xt) keep(hipEventRecord(xt->e0, s) == hipSuccess ? KABC_OK : KABC_ERR_DEVICE);
                    if (local_err == KABC_OK) keep(kabc_ais_half_generation(h, hf, ntransitions, nullptr));
                    if (xt) {
                        (void)timing_close_pair(h);  // (the kernel pair ends before the collective)
                        (void)hipEventRecord(xt->e1, s);
                        (void)hipEventRecord(xt->x0[0], s);
                    }
                    // the one collective of the design: rebuild half hf on every rank
                    if (h->comm)
                        keep(comm_allgather_inplace(h->comm, h->d_half[hf], (size_t)h->cper[hf] * h->D));
                    if (xt) {
                        (void)hipEventRecord(xt->x1[0], s);
                        (void)hipEventRecord(xt->e2, s);
                        xt->closed = true;
                        ++h->xt_used;
                    }
                } else {
                    // pipelined: the kernels read the half gathered last (fence), then chunk k
                    // is gathered on the exchange stream while the kernels of chunk k + 1 run
                    keep(comm_exchange_fence(h->comm));
                    if (h->xt_open >= 0) {  // the previous half's gathers have landed for this stream
                        (void)hipEventRecord(h->xt[(size_t)h->xt_open].e2, s);
                        h->xt[(size_t)h->xt_open].closed = true;
                        h->xt_open = -1;
                    }
                    if (xt) (void)hipEventRecord(xt->e0, s);
                    for (int k = 0; k < h->xk; ++k) {
                        if (local_err == KABC_OK)
                            keep(launch_half_seg(h, hf, h->seg[hf][k], ntransitions, nullptr));
                        if (xt && k == h->xk - 1) {
                            (void)timing_close_pair(h);
                            (void)hipEventRecord(xt->e1, s);
                        }
                        keep(comm_exchange_chunk(h->comm, chunk_base(h, hf, k),
                                                 (size_t)h->cper[hf] * h->D, k, xt ? xt->x0[k] : nullptr,
                                                 xt ? xt->x1[k] : nullptr));
                    }
                    if (xt) {
                        h->xt_open = (int)h->xt_used;
                        ++h->xt_used;
                    }
                }
                if (local_err && !h->comm) return local_err;
            }
            if (local_err == KABC_OK) h->t += (uint64_t)ntransitions;
        }
        if (h->comm && h->xk > 1) keep(comm_exchange_fence(h->comm));
        if (h->xt_open >= 0) {
            (void)hipEventRecord(h->xt[(size_t)h->xt_open].e2, s);
            h->xt[(size_t)h->xt_open].closed = true;
            h->xt_open = -1;
        }
        if (h->comm) {
            uint64_t bad = local_err != KABC_OK;
            const kabc_status_t st = kabc_comm_allreduce_sum_u64(h->comm, &bad, 1);
            if (local_err) {
                set_error("%s", local_msg);
                return local_err;
            }
            if (st) return st;
            if (bad) {
                set_error("kabc_ais_advance failed on %llu other rank(s) of the communicator",
                          (unsigned long long)bad);
                return KABC_ERR_DEVICE;
            }
        }
    } else {
        // ---- sample-trace streaming ------------------------------------------------
        // The kernels write the trace into kTraceBufs device chunks in rotation; a drain
        // thread copies each finished chunk to the caller's buffer while this thread keeps
        // the device fed with the next chunks.  The copies are issued from their own
        // thread because hipMemcpyAsync to host memory holds its calling thread until the
        // chunk's kernels have finished (measured: the whole run time, pinned or not); on
        // this thread that would stop kernel submission and idle the device.
        // 4..32 MiB chunks (1/16 of the trace): long enough to amortise a copy's set-up,
        // short enough that the last copy -- the only one no kernel hides -- is a short tail.
        size_t target = gen_bytes * (size_t)ngenerations / 16;
        if (target < (4ull << 20)) target = 4ull << 20;
        if (target > (32ull << 20)) target = 32ull << 20;
        if (const char* e = std::getenv("KABC_TRACE_CHUNK_MIB")) {  // tuning/probing only
            const long mib = std::atol(e);
            if (mib > 0) target = (size_t)mib << 20;
        }
        int64_t chunk = (int64_t)(target / gen_bytes);
        if (chunk < 1) chunk = 1;
        if (chunk > ngenerations) chunk = ngenerations;
        if (chunk > h->trace_cap_gens) {
            for (int b = 0; b < kTraceBufs; ++b) {
                if (h->d_trace[b]) KABC_HIP_CHECK(hipFree(h->d_trace[b]));
                h->d_trace[b] = nullptr;
                KABC_HIP_CHECK(dev_malloc(&h->d_trace[b], gen_bytes * chunk));
            }
            h->trace_cap_gens = chunk;
        }
        if (!h->copy_stream) {
            KABC_HIP_CHECK(hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
            for (int b = 0; b < kTraceBufs; ++b)
                KABC_HIP_CHECK(hipEventCreateWithFlags(&h->ev_filled[b], hipEventDisableTiming));
        }
        const int64_t nchunks = (ngenerations + chunk - 1) / chunk;
        std::mutex mu;
        std::condition_variable cv;
        int64_t filled = 0, drained = 0;  // chunks submitted / chunks copied out
        bool abort_drain = false;
        hipError_t drain_err = hipSuccess;
        std::thread drainer([&] {
            hipError_t e = hipSetDevice(h->ctx->device);
            for (int64_t c = 0; c < nchunks && e == hipSuccess; ++c) {
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return filled > c || abort_drain; });
                    if (abort_drain) break;
                }
                const int b = (int)(c % kTraceBufs);
                const int64_t gc = (c + 1 < nchunks) ? chunk : ngenerations - c * chunk;
                e = hipEventSynchronize(h->ev_filled[b]);
                if (e == hipSuccess)
                    e = hipMemcpyAsync(out_samples + c * chunk * gen_elems, h->d_trace[b],
                                       gen_bytes * gc, hipMemcpyDeviceToHost, h->copy_stream);
                if (e == hipSuccess) e = hipStreamSynchronize(h->copy_stream);
                std::lock_guard<std::mutex> lk(mu);
                drained = c + 1;
                cv.notify_all();
            }
            std::lock_guard<std::mutex> lk(mu);
            drain_err = e;
            drained = nchunks;  // releases the submitter on error
            cv.notify_all();
        });
        kabc_status_t st = KABC_OK;
        hipError_t sub_err = hipSuccess;
        for (int64_t c = 0; c < nchunks && st == KABC_OK && sub_err == hipSuccess; ++c) {
            const int b = (int)(c % kTraceBufs);
            const int64_t gc = (c + 1 < nchunks) ? chunk : ngenerations - c * chunk;
            if (c >= kTraceBufs) {  // the chunk that used this buffer last must be out
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return drained > c - kTraceBufs; });
                if (drain_err != hipSuccess) break;
            }
            for (int64_t g = 0; g < gc && st == KABC_OK; ++g) {
                double* tr0 = h->d_trace[b] + g * gen_elems;
                st = kabc_ais_half_generation(h, 0, ntransitions, tr0);
                if (st == KABC_OK)
                    st = kabc_ais_half_generation(h, 1, ntransitions, tr0 + h->rows[0] * h->D);
                if (st == KABC_OK) h->t += (uint64_t)ntransitions;
            }
            if (st != KABC_OK) break;
            sub_err = hipEventRecord(h->ev_filled[b], s);
            if (sub_err != hipSuccess) break;
            std::lock_guard<std::mutex> lk(mu);
            filled = c + 1;
            cv.notify_all();
        }
        {
            std::lock_guard<std::mutex> lk(mu);
            if (st != KABC_OK || sub_err != hipSuccess) abort_drain = true;
            cv.notify_all();
        }
        drainer.join();
        if (st != KABC_OK) return st;
        KABC_HIP_CHECK(sub_err);
        KABC_HIP_CHECK(drain_err);
    }
    if (kabc_status_t stc = timing_close_pair(h)) return stc;
    DevCounters c;
    if (read_counters(h, &c)) return KABC_ERR_DEVICE;
    kabc_status_t st = check_device_error(h, c);
    if (st) return st;
    if (stats) {
        stats->proposals += c.proposals - h->last.proposals;
        stats->cost_evals += c.cost_evals - h->last.cost_evals;
        stats->accepted += c.accepted - h->last.accepted;
    }
    h->last.proposals = c.proposals;
    h->last.cost_evals = c.cost_evals;
    h->last.accepted = c.accepted;
    return KABC_OK;
}

kabc_status_t kabc_ais_advance_multi(kabc_ais_t** hs, int32_t n, int64_t ngenerations,
                                     int32_t ntransitions, kabc_stats_t* stats) {
    if (kabc_status_t st = check_group(hs, n, "kabc_ais_advance_multi")) return st;
    if (ngenerations < 0 || ntransitions < 1) {
        set_error("ngenerations must be >= 0 and ntransitions >= 1");
        return KABC_ERR_INVALID_ARG;
    }
    for (int i = 0; i < n; ++i) {
        if (!hs[i]->initialised) {
            set_error("kabc_ais_init_multi has not been called");
            return KABC_ERR_INVALID_STATE;
        }
    }
    const int xk = hs[0]->xk;
    kabc_comm_t* comms[KABC_COMM_MAX_WORLD];
    for (int i = 0; i < n; ++i) comms[i] = hs[i]->comm;
    for (int64_t g = 0; g < ngenerations; ++g) {
        for (int hf = 0; hf < 2; ++hf) {
            if (xk == 1) {
                // exchange diagnostics (kabc_ais_exchange_us), as kabc_ais_advance records them: every
                // shard's kernels between e0 and e1 on its stream, the gather between e1 and e2
                kabc_ais::XT* xts[KABC_COMM_MAX_WORLD];
                for (int i = 0; i < n; ++i) {
                    kabc_ais_t* h = hs[i];
                    KABC_HIP_CHECK(hipSetDevice(h->ctx->device));
                    xts[i] = (h->timing && h->xt_used < h->xt.size()) ? &h->xt[h->xt_used] : nullptr;
                    if (xts[i]) KABC_HIP_CHECK(hipEventRecord(xts[i]->e0, h->ctx->stream));
                    if (kabc_status_t st = kabc_ais_half_generation(h, hf, ntransitions, nullptr))
                        return st;
                    if (xts[i]) {
                        if (kabc_status_t stc = timing_close_pair(h)) return stc;
                        KABC_HIP_CHECK(hipEventRecord(xts[i]->e1, h->ctx->stream));
                        KABC_HIP_CHECK(hipEventRecord(xts[i]->x0[0], h->ctx->stream));
                    }
                }
                if (kabc_status_t st = gather_multi(hs, n, hf, 0)) return st;
                for (int i = 0; i < n; ++i) {
                    if (!xts[i]) continue;
                    KABC_HIP_CHECK(hipSetDevice(hs[i]->ctx->device));
                    KABC_HIP_CHECK(hipEventRecord(xts[i]->x1[0], hs[i]->ctx->stream));
                    KABC_HIP_CHECK(hipEventRecord(xts[i]->e2, hs[i]->ctx->stream));
                    xts[i]->closed = true;
                    ++hs[i]->xt_used;
                }
                continue;
            }
            if (kabc_status_t st = comm_exchange_fence_multi(comms, n, false)) return st;
            for (int k = 0; k < xk; ++k) {
                double* bases[KABC_COMM_MAX_WORLD];
                for (int i = 0; i < n; ++i) {
                    KABC_HIP_CHECK(hipSetDevice(hs[i]->ctx->device));
                    if (kabc_status_t st = launch_half_seg(hs[i], hf, hs[i]->seg[hf][k], ntransitions, nullptr))
                        return st;
                    bases[i] = chunk_base(hs[i], hf, k);
                }
                if (kabc_status_t st = comm_exchange_chunk_multi(comms, bases, n,
                                                                 (size_t)hs[0]->cper[hf] * hs[0]->D, k))
                    return st;
            }
        }
        for (int i = 0; i < n; ++i) hs[i]->t += (uint64_t)ntransitions;
    }
    if (xk > 1)
        if (kabc_status_t st = comm_exchange_fence_multi(comms, n, true)) return st;
    for (int i = 0; i < n; ++i) {
        kabc_ais_t* h = hs[i];
        KABC_HIP_CHECK(hipSetDevice(h->ctx->device));
        if (kabc_status_t stc = timing_close_pair(h)) return stc;
        DevCounters c;
        if (read_counters(h, &c)) return KABC_ERR_DEVICE;
        if (kabc_status_t st = check_device_error(h, c)) return st;
        if (stats) {
            stats->proposals += c.proposals - h->last.proposals;
            stats->cost_evals += c.cost_evals - h->last.cost_evals;
            stats->accepted += c.accepted - h->last.accepted;
        }
        h->last.proposals = c.proposals;
        h->last.cost_evals = c.cost_evals;
        h->last.accepted = c.accepted;
    }
    return KABC_OK;
}

kabc_status_t kabc_ais_get_ensemble(kabc_ais_t* h, double* x) {
    if (check_handle(h) || !x) return KABC_ERR_INVALID_ARG;
    KABC_HIP_CHECK(hipSetDevice(h->ctx->device));
    hipStream_t s = h->ctx->stream;
    const size_t W = sizeof(double), nch = (size_t)h->nchains;
    // device pitch per chain: the (padded, for sharded handles) half buffer
    const size_t p0 = W * (h->comm ? h->cper[0] * h->xk * h->world : h->rows[0]) * h->D;
    const size_t p1 = W * (h->comm ? h->cper[1] * h->xk * h->world : h->rows[1]) * h->D;
    KABC_HIP_CHECK(hipMemcpy2DAsync(x, W * h->N * h->D, h->d_half[0], p0, W * h->rows[0] * h->D, nch,
                                    hipMemcpyDeviceToHost, s));
    if (h->rows[1] > 0)
        KABC_HIP_CHECK(hipMemcpy2DAsync(x + h->rows[0] * h->D, W * h->N * h->D, h->d_half[1], p1,
                                        W * h->rows[1] * h->D, nch, hipMemcpyDeviceToHost, s));
    KABC_HIP_CHECK(hipStreamSynchronize(s));
    return KABC_OK;
}

kabc_status_t kabc_ais_get_state(kabc_ais_t* h, double* x, double* logprior, double* loglik,
                                 uint64_t* t) {
    if (check_handle(h)) return KABC_ERR_INVALID_ARG;
    KABC_HIP_CHECK(hipSetDevice(h->ctx->device));
    hipStream_t s = h->ctx->stream;
    // one strided copy per half and array; height = chains (1 for an ordinary handle).  Host
    // layout per chain: owned rows of half 0, then of half 1.
    const size_t nch = (size_t)h->nchains, W = sizeof(double);
    const int64_t n_own = h->rows_owned[0] + h->rows_owned[1];
    int64_t off = 0;
    for (int hf = 0; hf < 2; ++hf) {
        const int64_t n = h->rows_owned[hf];
        // (the pitch between chains of a batch handle; sharded handles hold one chain)
        for (const kabc_ais::Seg& sg : h->seg[hf])
            if (x && sg.count > 0)
                KABC_HIP_CHECK(hipMemcpy2DAsync(x + (off + sg.off) * h->D, W * n_own * h->D,
                                                h->d_half[hf] + sg.first * h->D,
                                                W * h->rows[hf] * h->D, W * sg.count * h->D, nch,
                                                hipMemcpyDeviceToHost, s));
        if (n > 0) {
            if (logprior)
                KABC_HIP_CHECK(hipMemcpy2DAsync(logprior + off, W * n_own, h->d_lp[hf], W * n, W * n,
                                                nch, hipMemcpyDeviceToHost, s));
            if (loglik)
                KABC_HIP_CHECK(hipMemcpy2DAsync(loglik + off, W * n_own, h->d_ll[hf], W * n, W * n,
                                                nch, hipMemcpyDeviceToHost, s));
        }
        off += n;
    }
    KABC_HIP_CHECK(hipStreamSynchronize(s));
    if (t) *t = h->t;
    DevCounters c;
    if (read_counters(h, &c)) return KABC_ERR_DEVICE;
    return check_device_error(h, c);
}

kabc_status_t kabc_ais_set_state(kabc_ais_t* h, const double* x, const double* logprior,
                                 const double* loglik, uint64_t t) {
    if (check_handle(h)) return KABC_ERR_INVALID_ARG;
    if (!x || !logprior || !loglik) {
        set_error("kabc_ais_set_state: NULL buffer");
        return KABC_ERR_INVALID_ARG;
    }
    KABC_HIP_CHECK(hipSetDevice(h->ctx->device));
    hipStream_t s = h->ctx->stream;
    const size_t nch = (size_t)h->nchains, W = sizeof(double);
    const int64_t n_own = h->rows_owned[0] + h->rows_owned[1];
    int64_t off = 0;
    for (int hf = 0; hf < 2; ++hf) {
        const int64_t n = h->rows_owned[hf];
        for (const kabc_ais::Seg& sg : h->seg[hf])
            if (sg.count > 0)
                KABC_HIP_CHECK(hipMemcpy2DAsync(h->d_half[hf] + sg.first * h->D, W * h->rows[hf] * h->D,
                                                x + (off + sg.off) * h->D, W * n_own * h->D,
                                                W * sg.count * h->D, nch, hipMemcpyHostToDevice, s));
        if (n > 0) {
            KABC_HIP_CHECK(hipMemcpy2DAsync(h->d_lp[hf], W * n, logprior + off, W * n_own, W * n, nch,
                                            hipMemcpyHostToDevice, s));
            KABC_HIP_CHECK(hipMemcpy2DAsync(h->d_ll[hf], W * n, loglik + off, W * n_own, W * n, nch,
                                            hipMemcpyHostToDevice, s));
        }
        off += n;
    }
    KABC_HIP_CHECK(hipStreamSynchronize(s));
    h->t = t;
    h->initialised = true;
    return KABC_OK;
}

kabc_status_t kabc_ais_get_stats(kabc_ais_t* h, kabc_stats_t* stats) {
    if (check_handle(h) || !stats) return KABC_ERR_INVALID_ARG;
    KABC_HIP_CHECK(hipSetDevice(h->ctx->device));
    DevCounters c;
    if (read_counters(h, &c)) return KABC_ERR_DEVICE;
    stats->proposals = c.proposals;
    stats->cost_evals = c.cost_evals;
    stats->accepted = c.accepted;
    return check_device_error(h, c);
}

int32_t kabc_ais_driver(const kabc_ais_t* h) { return (h && h->small_ok) ? 1 : 0; }

int64_t kabc_ais_owned(const kabc_ais_t* h, int32_t half) {
    if (!h || (half != 0 && half != 1)) return -1;
    return h->rows_owned[half];
}

int32_t kabc_ais_owned_segments(const kabc_ais_t* h, int32_t half, int64_t* first, int64_t* count,
                                int32_t cap) {
    if (!h || (half != 0 && half != 1) || cap < 0) return -1;
    const int32_t n = (int32_t)h->seg[half].size();
    for (int32_t i = 0; i < n && i < cap; ++i) {
        if (first) first[i] = h->seg[half][i].first;
        if (count) count[i] = h->seg[half][i].count;
    }
    return n;
}

kabc_status_t kabc_ais_spec_state(kabc_ais_t* h, int32_t* state, int64_t* launches_before_switch) {
    if (check_handle(h)) return KABC_ERR_INVALID_ARG;
    if (state) *state = h->spec_state;
    if (launches_before_switch) *launches_before_switch = h->spec_switch_at;
    return KABC_OK;
}

kabc_status_t kabc_ais_set_timing_stride(kabc_ais_t* h, int32_t stride) {
    if (check_handle(h)) return KABC_ERR_INVALID_ARG;
    if (kabc_status_t st = timing_close_pair(h)) return st;
    h->timing_stride = stride > 0 ? stride : 1;
    return KABC_OK;
}

kabc_status_t kabc_ais_set_timing(kabc_ais_t* h, int32_t max_launches) {
    if (check_handle(h)) return KABC_ERR_INVALID_ARG;
    KABC_HIP_CHECK(hipSetDevice(h->ctx->device));
    for (hipEvent_t e : h->ev) (void)hipEventDestroy(e);
    h->ev.clear();
    h->ev_used = 0;
    h->open_count = 0;
    h->timing = max_launches > 0;
    for (int i = 0; i < 2 * max_launches; ++i) {
        hipEvent_t e;
        KABC_HIP_CHECK(hipEventCreate(&e));
        h->ev.push_back(e);
    }
    h->ev_n.assign((size_t)(max_launches > 0 ? max_launches : 0), 0);
    // exchange diagnostics of a sharded handle: up to kXtHalves half-generations
    for (kabc_ais::XT& x : h->xt) {
        (void)hipEventDestroy(x.e0);
        (void)hipEventDestroy(x.e1);
        (void)hipEventDestroy(x.e2);
        for (int k = 0; k < h->xk; ++k) {
            (void)hipEventDestroy(x.x0[k]);
            (void)hipEventDestroy(x.x1[k]);
        }
    }
    h->xt.clear();
    h->xt_used = 0;
    h->xt_open = -1;
    if (h->comm && max_launches > 0) {
        const size_t nh = (size_t)max_launches < kXtHalves ? (size_t)max_launches : kXtHalves;
        for (size_t i = 0; i < nh; ++i) {
            kabc_ais::XT x;
            std::memset(&x, 0, sizeof x);
            KABC_HIP_CHECK(hipEventCreate(&x.e0));
            KABC_HIP_CHECK(hipEventCreate(&x.e1));
            KABC_HIP_CHECK(hipEventCreate(&x.e2));
            for (int k = 0; k < h->xk; ++k) {
                KABC_HIP_CHECK(hipEventCreate(&x.x0[k]));
                KABC_HIP_CHECK(hipEventCreate(&x.x1[k]));
            }
            h->xt.push_back(x);
        }
    }
    return KABC_OK;
}

kabc_status_t kabc_ais_exchange_us(kabc_ais_t* h, double out[4]) {
    if (check_handle(h) || !out) return KABC_ERR_INVALID_ARG;
    out[0] = out[1] = out[2] = 0.0;
    out[3] = (double)h->xk;
    KABC_HIP_CHECK(hipSetDevice(h->ctx->device));
    KABC_HIP_CHECK(hipStreamSynchronize(h->ctx->stream));
    if (h->comm && h->comm->xstream) KABC_HIP_CHECK(hipStreamSynchronize(h->comm->xstream));
    double comp = 0.0, exch = 0.0, expo = 0.0;
    int64_t n = 0;
    for (size_t i = 0; i < h->xt_used; ++i) {
        const kabc_ais::XT& x = h->xt[i];
        if (!x.closed) continue;
        float c = 0.f, e = 0.f, xs = 0.f;
        if (hipEventElapsedTime(&c, x.e0, x.e1) != hipSuccess || hipEventElapsedTime(&e, x.e1, x.e2) != hipSuccess)
            continue;
        bool ok = true;
        for (int k = 0; k < h->xk && ok; ++k) {
            float g = 0.f;
            ok = hipEventElapsedTime(&g, x.x0[k], x.x1[k]) == hipSuccess;
            xs += g;
        }
        if (!ok) continue;
        comp += c;
        expo += e;
        exch += xs;
        ++n;
    }
    for (size_t i = 0; i < h->xt_used; ++i) h->xt[i].closed = false;
    h->xt_used = 0;
    if (n) {
        out[0] = comp / (double)n * 1e3;
        out[1] = exch / (double)n * 1e3;
        out[2] = expo / (double)n * 1e3;
    }
    return KABC_OK;
}

double kabc_ais_kernel_ms(kabc_ais_t* h, int64_t* nlaunches) {
    if (nlaunches) *nlaunches = 0;
    if (!h || h->ev_used == 0) return 0.0;
    (void)hipSetDevice(h->ctx->device);
    if (hipStreamSynchronize(h->ctx->stream) != hipSuccess) return 0.0;
    if (h->open_count > 0) (void)timing_close_pair(h);
    if (hipStreamSynchronize(h->ctx->stream) != hipSuccess) return 0.0;
    double total = 0.0;
    int64_t n = 0;
    for (size_t i = 0; i + 1 < h->ev_used; i += 2) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, h->ev[i], h->ev[i + 1]) == hipSuccess) {
            total += ms;
            n += h->ev_n[i / 2];
        }
    }
    h->ev_used = 0;
    if (nlaunches) *nlaunches = n;
    return n ? total / (double)n : 0.0;
}

kabc_status_t kabc_ais_set_debug(kabc_ais_t* h, int32_t ntransitions) {
    if (check_handle(h)) return KABC_ERR_INVALID_ARG;
    if (h->nchains != 1 && ntransitions > 0) {
        set_error("debug records are for single-chain handles");
        return KABC_ERR_INVALID_ARG;
    }
    KABC_HIP_CHECK(hipSetDevice(h->ctx->device));
    if (h->d_dbg) (void)hipFree(h->d_dbg);
    h->d_dbg = nullptr;
    h->dbg_cap = 0;
    if (ntransitions > 0) {
        h->dbg_cap = (h->rows_owned[0] + h->rows_owned[1]) * (int64_t)ntransitions * 6;
        KABC_HIP_CHECK(dev_malloc(&h->d_dbg, sizeof(int32_t) * h->dbg_cap));
        // (on the handle's stream: the null stream is not ordered against a non-blocking one)
        KABC_HIP_CHECK(hipMemsetAsync(h->d_dbg, 0xff, sizeof(int32_t) * h->dbg_cap, h->ctx->stream));
    }
    return KABC_OK;
}

kabc_status_t kabc_ais_get_debug(kabc_ais_t* h, int32_t* out, int64_t n_int32) {
    if (check_handle(h) || !out) return KABC_ERR_INVALID_ARG;
    if (!h->d_dbg || n_int32 > h->dbg_cap) {
        set_error("debug records not enabled or buffer too large");
        return KABC_ERR_INVALID_ARG;
    }
    KABC_HIP_CHECK(hipSetDevice(h->ctx->device));
    KABC_HIP_CHECK(hipStreamSynchronize(h->ctx->stream));
    KABC_HIP_CHECK(hipMemcpy(out, h->d_dbg, sizeof(int32_t) * n_int32, hipMemcpyDeviceToHost));
    return KABC_OK;
}

kabc_status_t kabc_ais_destroy(kabc_ais_t* h) {
    if (!h) return KABC_OK;
    (void)hipSetDevice(h->ctx->device);
    (void)hipStreamSynchronize(h->ctx->stream);
    for (int hf = 0; hf < 2; ++hf) {
        if (h->own_halves && h->d_half[hf]) (void)hipFree(h->d_half[hf]);
        if (h->d_lp[hf]) (void)hipFree(h->d_lp[hf]);
        if (h->d_ll[hf]) (void)hipFree(h->d_ll[hf]);
    }
    if (h->d_cost_params) (void)hipFree(h->d_cost_params);
    if (h->d_cost_data) (void)hipFree(h->d_cost_data);
    if (h->d_counters) (void)hipFree(h->d_counters);
    if (h->d_slots) (void)hipFree(h->d_slots);
    if (h->d_prior) (void)hipFree(h->d_prior);
    if (h->d_raw) (void)hipFree(h->d_raw);
    if (h->d_scratch) (void)hipFree(h->d_scratch);
    if (h->d_seeds) (void)hipFree(h->d_seeds);
    if (h->d_chain_retries) (void)hipFree(h->d_chain_retries);
    for (int b = 0; b < kTraceBufs; ++b) {
        if (h->d_trace[b]) (void)hipFree(h->d_trace[b]);
        if (h->ev_filled[b]) (void)hipEventDestroy(h->ev_filled[b]);
    }
    if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
    if (h->d_dbg) (void)hipFree(h->d_dbg);
    if (h->d_aux) (void)hipFree(h->d_aux);
    if (h->d_strace) (void)hipFree(h->d_strace);
    for (hipEvent_t e : h->ev) (void)hipEventDestroy(e);
    for (kabc_ais::XT& x : h->xt) {
        (void)hipEventDestroy(x.e0);
        (void)hipEventDestroy(x.e1);
        (void)hipEventDestroy(x.e2);
        for (int k = 0; k < h->xk; ++k) {
            (void)hipEventDestroy(x.x0[k]);
            (void)hipEventDestroy(x.x1[k]);
        }
    }
    delete h;
    return KABC_OK;
}

}  // extern "C"
