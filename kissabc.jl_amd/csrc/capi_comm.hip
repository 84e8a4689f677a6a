// capi_comm.hip -- communicators of the walker-sharded path (include/kabc.h "multi-GPU").
//
// The reference has no device-to-device communication (`MCMCDistributed` is a
// re-exported AbstractMCMC tag, src/KissABC.jl:9,175); the one collective of this design --
// the all-gather that rebuilds the complementary half after every half-generation -- lives
// HERE, behind the C ABI, so that a Julia / C host needs nothing but `ccall`:
//   * RCCL backend: librccl.so is loaded at first use (dlopen; the library itself links only
//     the HIP runtime), ncclAllGather in place on the context stream, i.e. ordered behind
//     the half-generation kernel without any host synchronisation;
//   * P2P backend (single process): a pull kernel reads the peers' fresh rows through
//     peer-mapped pointers -- on MI355X all seven xGMI links of a GPU carry traffic at
//     once and no ring is formed (xGMI is point-to-point, a ring all-gather is bound by one
//     link); ordering by one event per rank.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <chrono>
#include <condition_variable>
#include <mutex>
#include <string>

#include "host_common.hpp"

namespace kabc {

// ---- librccl.so, loaded on demand -------------------------------------------------
struct Rccl {
    void* dl = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t,
                              hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t,
                              hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string why;
};

static Rccl g_rccl;
static Rccl* rccl() {
    Rccl& R = g_rccl;
    static std::once_flag once;
    std::call_once(once, [&R] {
        const char* cand[] = {std::getenv("KABC_RCCL_LIB"), "librccl.so.1", "librccl.so",
                              "/opt/rocm/lib/librccl.so.1"};
        for (const char* c : cand) {
            if (!c || !*c) continue;
            R.dl = dlopen(c, RTLD_NOW | RTLD_LOCAL);
            if (R.dl) break;
            R.why += std::string(c) + ": " + dlerror() + "; ";
        }
        if (!R.dl) return;
#define KABC_SYM(field, name)                                    \
    R.field = (decltype(R.field))dlsym(R.dl, name);              \
    if (!R.field) {                                              \
        R.why += std::string("missing symbol ") + name + "; ";   \
        ok = false;                                              \
    }
        bool ok = true;
        KABC_SYM(GetUniqueId, "ncclGetUniqueId")
        KABC_SYM(CommInitRank, "ncclCommInitRank")
        KABC_SYM(CommInitAll, "ncclCommInitAll")
        KABC_SYM(CommDestroy, "ncclCommDestroy")
        KABC_SYM(AllGather, "ncclAllGather")
        KABC_SYM(AllReduce, "ncclAllReduce")
        KABC_SYM(GroupStart, "ncclGroupStart")
        KABC_SYM(GroupEnd, "ncclGroupEnd")
        KABC_SYM(GetErrorString, "ncclGetErrorString")
#undef KABC_SYM
        if (!ok) {
            dlclose(R.dl);
            R.dl = nullptr;
        }
    });
    return R.dl ? &R : nullptr;
}

static kabc_status_t need_rccl(Rccl** out) {
    Rccl* r = rccl();
    if (!r) {
        set_error("RCCL is not available: %s(set KABC_RCCL_LIB to the path of librccl.so)",
                  g_rccl.why.c_str());
        return KABC_ERR_DEVICE;
    }
    *out = r;
    return KABC_OK;
}

#define KABC_NCCL_CHECK(R, expr)                                                        \
    do {                                                                                \
        ncclResult_t _r = (expr);                                                       \
        if (_r != ncclSuccess) {                                                        \
            kabc::set_error("RCCL error %d (%s) at %s:%d: %s", (int)_r,                 \
                            (R)->GetErrorString(_r), __FILE__, __LINE__, #expr);        \
            return KABC_ERR_DEVICE;                                                     \
        }                                                                               \
    } while (0)

// ---- P2P backend ------------------------------------------------------------------
struct P2PGroup {
    int world = 0;
    int refs = 0;
    kabc_comm* member[KABC_COMM_MAX_WORLD] = {};
    hipEvent_t ev[KABC_COMM_MAX_WORLD] = {};  // ev[r]: rank r's rows of the gathered half are final
    // ranks driven by one host thread each (comm_allgather_many): rendezvous + published bases
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    unsigned long long phase = 0;
    double* pub[KABC_COMM_MAX_WORLD][8] = {};
};

// false: a rank did not show up within 120 s (it failed before its collective: the caller
// returns an error instead of waiting for ever)
static bool p2p_host_barrier(P2PGroup* g) {
    std::unique_lock<std::mutex> lk(g->mu);
    const unsigned long long ph = g->phase;
    if (++g->arrived == g->world) {
        g->arrived = 0;
        ++g->phase;
        g->cv.notify_all();
        return true;
    }
    if (g->cv.wait_for(lk, std::chrono::seconds(120), [&] { return g->phase != ph; })) return true;
    --g->arrived;
    return false;
}

// exchange stream + events of a communicator, created at first use (on the context's device)
static kabc_status_t exchange_setup(kabc_comm* c) {
    if (c->xstream) return KABC_OK;
    KABC_HIP_CHECK(hipSetDevice(c->ctx->device));
    // the stream last: it is the "set up" flag, and a failure half way must not leave it set
    for (int k = 0; k < KABC_MAX_EXCHANGE_CHUNKS; ++k)
        if (!c->ev_chunk[k]) KABC_HIP_CHECK(hipEventCreateWithFlags(&c->ev_chunk[k], hipEventDisableTiming));
    if (!c->ev_done) KABC_HIP_CHECK(hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming));
    KABC_HIP_CHECK(hipStreamCreateWithFlags(&c->xstream, hipStreamNonBlocking));
    return KABC_OK;
}

struct PullArgs {
    const double* src[KABC_COMM_MAX_WORLD];  // rank r's buffer base (peer-mapped)
    double* dst;                             // this rank's buffer base
    unsigned long long count;                // doubles per rank segment
    int self, world;
};

// blockIdx.y walks the world-1 peers, blockIdx.x strides over one segment.  Reads are
// remote (xGMI), writes local: one launch keeps every link of this GPU busy.
__global__ void __launch_bounds__(256) p2p_pull_kernel(const PullArgs A) {
    int r = (int)blockIdx.y;
    r += (r >= A.self);
    const size_t off = (size_t)r * A.count;
    const double* __restrict__ s = A.src[r] + off;
    double* __restrict__ d = A.dst + off;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (((A.count | off) & 1ull) == 0) {  // segments 16-byte aligned: dwordx4 both ways
        const double2* __restrict__ s2 = reinterpret_cast<const double2*>(s);
        double2* __restrict__ d2 = reinterpret_cast<double2*>(d);
        for (const size_t n2 = A.count / 2; i < n2; i += stride) d2[i] = s2[i];
    } else {
        for (; i < A.count; i += stride) d[i] = s[i];
    }
}

static kabc_status_t p2p_launch_pull(kabc_comm** comms, const int* at, double** bases, int n, int d,
                                     size_t count, hipStream_t s) {
    unsigned gx = (unsigned)((count / 2 + 255) / 256);
    gx = gx < 1 ? 1 : (gx > 64 ? 64 : gx);
    PullArgs a;
    std::memset(&a, 0, sizeof a);
    for (int r = 0; r < n; ++r) a.src[r] = bases[at[r]];
    a.dst = bases[at[d]];
    a.count = count;
    a.self = d;
    a.world = n;
    hipLaunchKernelGGL(p2p_pull_kernel, dim3(gx, (unsigned)(n - 1)), dim3(256), 0, s, a);
    KABC_HIP_CHECK(hipGetLastError());
    return KABC_OK;
}

static kabc_status_t p2p_allgather_multi(kabc_comm** comms, double** bases, int n, size_t count) {
    P2PGroup* g = comms[0]->grp;
    // order[r] = index into comms[] of rank r
    int at[KABC_COMM_MAX_WORLD];
    for (int i = 0; i < n; ++i) at[comms[i]->rank] = i;
    for (int r = 0; r < n; ++r) {
        kabc_comm* c = comms[at[r]];
        KABC_HIP_CHECK(hipSetDevice(c->ctx->device));
        KABC_HIP_CHECK(hipEventRecord(g->ev[r], c->ctx->stream));
    }
    if (n == 1 || count == 0) return KABC_OK;
    for (int d = 0; d < n; ++d) {
        kabc_comm* c = comms[at[d]];
        KABC_HIP_CHECK(hipSetDevice(c->ctx->device));
        for (int r = 0; r < n; ++r)
            if (r != d) KABC_HIP_CHECK(hipStreamWaitEvent(c->ctx->stream, g->ev[r], 0));
        if (kabc_status_t st = p2p_launch_pull(comms, at, bases, n, d, count, c->ctx->stream)) return st;
    }
    return KABC_OK;
}

kabc_status_t comm_allgather_inplace(kabc_comm* c, double* base, size_t count) {
    if (c->single_process) {
        set_error("this communicator belongs to a single-process group: use the *_multi entry points");
        return KABC_ERR_INVALID_ARG;
    }
    Rccl* R;
    if (kabc_status_t st = need_rccl(&R)) return st;
    if (count == 0) return KABC_OK;
    KABC_NCCL_CHECK(R, R->AllGather(base + (size_t)c->rank * count, base, count, ncclDouble,
                                    (ncclComm_t)c->nccl, c->ctx->stream));
    return KABC_OK;
}

kabc_status_t comm_allgather_inplace_multi(kabc_comm** comms, double** bases, int n, size_t count) {
    if (comms[0]->backend == KABC_COMM_P2P) return p2p_allgather_multi(comms, bases, n, count);
    Rccl* R;
    if (kabc_status_t st = need_rccl(&R)) return st;
    if (count == 0) return KABC_OK;
    KABC_NCCL_CHECK(R, R->GroupStart());
    for (int i = 0; i < n; ++i) {
        kabc_comm* c = comms[i];
        ncclResult_t r = R->AllGather(bases[i] + (size_t)c->rank * count, bases[i], count,
                                      ncclDouble, (ncclComm_t)c->nccl, c->ctx->stream);
        if (r != ncclSuccess) {
            (void)R->GroupEnd();
            set_error("RCCL error %d (%s) in grouped ncclAllGather", (int)r, R->GetErrorString(r));
            return KABC_ERR_DEVICE;
        }
    }
    KABC_NCCL_CHECK(R, R->GroupEnd());
    return KABC_OK;
}

kabc_status_t comm_allgather_many(kabc_comm* c, double** bases, const size_t* counts, int n) {
    if (n < 1 || n > 8) {
        set_error("comm_allgather_many: 1..8 buffers");
        return KABC_ERR_INVALID_ARG;
    }
    if (c->backend == KABC_COMM_P2P) {
        // one host thread per rank: publish, meet, pull, meet (synchronous: this form exists so
        // that the multi-rank logic can run -- and be tested -- on one GPU)
        P2PGroup* g = c->grp;
        KABC_HIP_CHECK(hipSetDevice(c->ctx->device));
        for (int j = 0; j < n; ++j) g->pub[c->rank][j] = bases[j];
        KABC_HIP_CHECK(hipStreamSynchronize(c->ctx->stream));  // this rank's segments are final
        if (!p2p_host_barrier(g)) {
            set_error("P2P all-gather: a rank of the group did not arrive within 120 s");
            return KABC_ERR_DEVICE;
        }
        kabc_status_t st = KABC_OK;
        for (int j = 0; j < n && st == KABC_OK && c->world > 1; ++j) {
            if (counts[j] == 0) continue;
            PullArgs a;
            std::memset(&a, 0, sizeof a);
            for (int r = 0; r < c->world; ++r) a.src[r] = g->pub[r][j];
            a.dst = bases[j];
            a.count = counts[j];
            a.self = c->rank;
            a.world = c->world;
            unsigned gx = (unsigned)((counts[j] / 2 + 255) / 256);
            gx = gx < 1 ? 1 : (gx > 64 ? 64 : gx);
            hipLaunchKernelGGL(p2p_pull_kernel, dim3(gx, (unsigned)(c->world - 1)), dim3(256), 0,
                               c->ctx->stream, a);
            if (hipGetLastError() != hipSuccess) st = KABC_ERR_DEVICE;
        }
        const hipError_t e = hipStreamSynchronize(c->ctx->stream);
        // nobody overwrites a source before every pull has finished
        if (!p2p_host_barrier(g) && st == KABC_OK) st = KABC_ERR_DEVICE;
        if (st != KABC_OK || e != hipSuccess) {
            set_error("P2P all-gather failed: %s", hipGetErrorString(e));
            return KABC_ERR_DEVICE;
        }
        return KABC_OK;
    }
    Rccl* R;
    if (kabc_status_t st = need_rccl(&R)) return st;
    KABC_NCCL_CHECK(R, R->GroupStart());
    for (int j = 0; j < n; ++j) {
        if (counts[j] == 0) continue;
        const ncclResult_t r = R->AllGather(bases[j] + (size_t)c->rank * counts[j], bases[j], counts[j],
                                            ncclDouble, (ncclComm_t)c->nccl, c->ctx->stream);
        if (r != ncclSuccess) {
            (void)R->GroupEnd();
            set_error("RCCL error %d (%s) in grouped ncclAllGather", (int)r, R->GetErrorString(r));
            return KABC_ERR_DEVICE;
        }
    }
    KABC_NCCL_CHECK(R, R->GroupEnd());
    return KABC_OK;
}

// ---- pipelined exchange ---------------------------------------------------------------
kabc_status_t comm_exchange_chunk(kabc_comm* c, double* base, size_t count, int k, hipEvent_t t0, hipEvent_t t1) {
    if (c->single_process) {
        set_error("this communicator belongs to a single-process group: use the *_multi entry points");
        return KABC_ERR_INVALID_ARG;
    }
    Rccl* R;
    if (kabc_status_t st = need_rccl(&R)) return st;
    if (kabc_status_t st = exchange_setup(c)) return st;
    KABC_HIP_CHECK(hipEventRecord(c->ev_chunk[k], c->ctx->stream));
    KABC_HIP_CHECK(hipStreamWaitEvent(c->xstream, c->ev_chunk[k], 0));
    // (t0 / t1: optional timing events around the gather on the exchange stream)
    if (t0) KABC_HIP_CHECK(hipEventRecord(t0, c->xstream));
    if (count != 0)
        KABC_NCCL_CHECK(R, R->AllGather(base + (size_t)c->rank * count, base, count, ncclDouble,
                                        (ncclComm_t)c->nccl, c->xstream));
    if (t1) KABC_HIP_CHECK(hipEventRecord(t1, c->xstream));
    return KABC_OK;
}

kabc_status_t comm_exchange_fence(kabc_comm* c) {
    if (!c->xstream) return KABC_OK;  // nothing was ever issued on the exchange stream
    KABC_HIP_CHECK(hipEventRecord(c->ev_done, c->xstream));
    KABC_HIP_CHECK(hipStreamWaitEvent(c->ctx->stream, c->ev_done, 0));
    return KABC_OK;
}

kabc_status_t comm_exchange_chunk_multi(kabc_comm** comms, double** bases, int n, size_t count, int k) {
    int at[KABC_COMM_MAX_WORLD];
    for (int i = 0; i < n; ++i) at[comms[i]->rank] = i;
    // "the kernels of chunk k are done" on every rank's context stream
    for (int r = 0; r < n; ++r) {
        kabc_comm* c = comms[at[r]];
        if (kabc_status_t st = exchange_setup(c)) return st;
        KABC_HIP_CHECK(hipSetDevice(c->ctx->device));
        KABC_HIP_CHECK(hipEventRecord(c->ev_chunk[k], c->ctx->stream));
    }
    if (comms[0]->backend == KABC_COMM_P2P) {
        // rank d pulls the peers' segments of the chunk once THEIR kernels of it are done
        for (int d = 0; d < n; ++d) {
            kabc_comm* c = comms[at[d]];
            KABC_HIP_CHECK(hipSetDevice(c->ctx->device));
            for (int r = 0; r < n; ++r)
                KABC_HIP_CHECK(hipStreamWaitEvent(c->xstream, comms[at[r]]->ev_chunk[k], 0));
            if (n > 1 && count > 0)
                if (kabc_status_t st = p2p_launch_pull(comms, at, bases, n, d, count, c->xstream)) return st;
        }
        return KABC_OK;
    }
    Rccl* R;
    if (kabc_status_t st = need_rccl(&R)) return st;
    for (int i = 0; i < n; ++i) {
        KABC_HIP_CHECK(hipSetDevice(comms[i]->ctx->device));
        KABC_HIP_CHECK(hipStreamWaitEvent(comms[i]->xstream, comms[i]->ev_chunk[k], 0));
    }
    if (count == 0) return KABC_OK;
    KABC_NCCL_CHECK(R, R->GroupStart());
    for (int i = 0; i < n; ++i) {
        kabc_comm* c = comms[i];
        ncclResult_t r = R->AllGather(bases[i] + (size_t)c->rank * count, bases[i], count,
                                      ncclDouble, (ncclComm_t)c->nccl, c->xstream);
        if (r != ncclSuccess) {
            (void)R->GroupEnd();
            set_error("RCCL error %d (%s) in grouped ncclAllGather", (int)r, R->GetErrorString(r));
            return KABC_ERR_DEVICE;
        }
    }
    KABC_NCCL_CHECK(R, R->GroupEnd());
    return KABC_OK;
}

kabc_status_t comm_exchange_fence_multi(kabc_comm** comms, int n, bool all_ranks) {
    for (int i = 0; i < n; ++i) {
        kabc_comm* c = comms[i];
        if (!c->xstream) return KABC_OK;  // set up together: none has one
        KABC_HIP_CHECK(hipSetDevice(c->ctx->device));
        KABC_HIP_CHECK(hipEventRecord(c->ev_done, c->xstream));
    }
    for (int i = 0; i < n; ++i) {
        kabc_comm* c = comms[i];
        KABC_HIP_CHECK(hipSetDevice(c->ctx->device));
        for (int j = 0; j < n; ++j)
            if (j == i || all_ranks)
                KABC_HIP_CHECK(hipStreamWaitEvent(c->ctx->stream, comms[j]->ev_done, 0));
    }
    return KABC_OK;
}

}  // namespace kabc

using namespace kabc;

extern "C" {

kabc_status_t kabc_comm_unique_id(uint8_t id[KABC_COMM_ID_BYTES]) {
    static_assert(sizeof(ncclUniqueId) == KABC_COMM_ID_BYTES, "ncclUniqueId size");
    if (!id) {
        set_error("kabc_comm_unique_id: NULL");
        return KABC_ERR_INVALID_ARG;
    }
    Rccl* R;
    if (kabc_status_t st = need_rccl(&R)) return st;
    ncclUniqueId u;
    KABC_NCCL_CHECK(R, R->GetUniqueId(&u));
    std::memcpy(id, &u, sizeof u);
    return KABC_OK;
}

kabc_status_t kabc_comm_init_rank(kabc_ctx_t* ctx, const uint8_t id[KABC_COMM_ID_BYTES],
                                  int32_t rank, int32_t world, kabc_comm_t** out) {
    if (!ctx || !id || !out || world < 1 || world > KABC_COMM_MAX_WORLD || rank < 0 ||
        rank >= world) {
        set_error("kabc_comm_init_rank: bad argument (world must be 1..%d, 0 <= rank < world)",
                  KABC_COMM_MAX_WORLD);
        return KABC_ERR_INVALID_ARG;
    }
    Rccl* R;
    if (kabc_status_t st = need_rccl(&R)) return st;
    KABC_HIP_CHECK(hipSetDevice(ctx->device));
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof u);
    ncclComm_t nc = nullptr;
    KABC_NCCL_CHECK(R, R->CommInitRank(&nc, world, u, rank));
    kabc_comm_t* c = new kabc_comm_t();
    c->ctx = ctx;
    c->rank = rank;
    c->world = world;
    c->backend = KABC_COMM_RCCL;
    c->own_ctx = false;
    c->single_process = false;
    c->nccl = nc;
    c->grp = nullptr;
    c->d_scratch = nullptr;
    c->xstream = nullptr;
    if (hipMalloc(&c->d_scratch, 64 * sizeof(double)) != hipSuccess) {
        (void)R->CommDestroy(nc);
        delete c;
        set_error("kabc_comm_init_rank: hipMalloc failed");
        return KABC_ERR_DEVICE;
    }
    *out = c;
    return KABC_OK;
}

kabc_status_t kabc_comm_init_all(int32_t ndev, const int32_t* dev_ids, int32_t backend,
                                 kabc_ctx_t** ctxs, kabc_comm_t** comms) {
    if (ndev < 1 || ndev > KABC_COMM_MAX_WORLD || !dev_ids || !ctxs || !comms ||
        (backend != KABC_COMM_RCCL && backend != KABC_COMM_P2P)) {
        set_error("kabc_comm_init_all: bad argument (ndev must be 1..%d, backend RCCL or P2P)",
                  KABC_COMM_MAX_WORLD);
        return KABC_ERR_INVALID_ARG;
    }
    for (int i = 0; i < ndev; ++i) {
        ctxs[i] = nullptr;
        comms[i] = nullptr;
    }
    auto fail = [&](kabc_status_t st) {
        for (int i = 0; i < ndev; ++i) {
            if (comms[i]) {
                delete comms[i];
                comms[i] = nullptr;
            }
            if (ctxs[i]) {
                (void)kabc_ctx_destroy(ctxs[i]);
                ctxs[i] = nullptr;
            }
        }
        return st;
    };
    for (int i = 0; i < ndev; ++i)
        if (kabc_status_t st = kabc_ctx_create(dev_ids[i], nullptr, &ctxs[i])) return fail(st);
    ncclComm_t nc[KABC_COMM_MAX_WORLD] = {};
    P2PGroup* grp = nullptr;
    if (backend == KABC_COMM_RCCL) {
        Rccl* R;
        if (kabc_status_t st = need_rccl(&R)) return fail(st);
        int devs[KABC_COMM_MAX_WORLD];
        for (int i = 0; i < ndev; ++i) devs[i] = dev_ids[i];
        ncclResult_t r = R->CommInitAll(nc, ndev, devs);
        if (r != ncclSuccess) {
            set_error("ncclCommInitAll failed: %d (%s)", (int)r, R->GetErrorString(r));
            return fail(KABC_ERR_DEVICE);
        }
    } else {
        grp = new P2PGroup();
        grp->world = ndev;
        grp->refs = ndev;
        for (int i = 0; i < ndev; ++i) {
            hipError_t e = hipSetDevice(dev_ids[i]);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&grp->ev[i], hipEventDisableTiming);
            for (int j = 0; j < ndev && e == hipSuccess; ++j) {
                if (dev_ids[j] == dev_ids[i]) continue;
                int can = 0;
                e = hipDeviceCanAccessPeer(&can, dev_ids[i], dev_ids[j]);
                if (e == hipSuccess && !can) {
                    delete grp;
                    set_error("device %d cannot map the memory of device %d (no peer access)",
                              dev_ids[i], dev_ids[j]);
                    return fail(KABC_ERR_DEVICE);
                }
                if (e == hipSuccess) {
                    e = hipDeviceEnablePeerAccess(dev_ids[j], 0);
                    if (e == hipErrorPeerAccessAlreadyEnabled) {
                        (void)hipGetLastError();
                        e = hipSuccess;
                    }
                }
            }
            if (e != hipSuccess) {
                delete grp;
                set_error("P2P set-up failed: %s", hipGetErrorString(e));
                return fail(KABC_ERR_DEVICE);
            }
        }
    }
    for (int i = 0; i < ndev; ++i) {
        kabc_comm_t* c = new kabc_comm_t();
        c->ctx = ctxs[i];
        c->rank = i;
        c->world = ndev;
        c->backend = backend;
        c->own_ctx = true;
        c->single_process = true;
        c->nccl = nc[i];
        c->grp = grp;
        c->d_scratch = nullptr;
        c->xstream = nullptr;
        if (grp) grp->member[i] = c;
        comms[i] = c;
    }
    return KABC_OK;
}

int32_t kabc_comm_rank(const kabc_comm_t* c) { return c ? c->rank : -1; }
int32_t kabc_comm_world(const kabc_comm_t* c) { return c ? c->world : -1; }
kabc_ctx_t* kabc_comm_ctx(const kabc_comm_t* c) { return c ? c->ctx : nullptr; }

static kabc_status_t host_allreduce(kabc_comm_t* c, void* inout, int32_t n, ncclDataType_t dt,
                                    ncclRedOp_t op) {
    if (!c || !inout || n < 1 || n > 64) {
        set_error("kabc_comm_allreduce: bad argument (1 <= n <= 64)");
        return KABC_ERR_INVALID_ARG;
    }
    if (c->single_process) {
        set_error("host-value reductions are for one-process-per-GPU communicators; a "
                  "single-process host reduces its own values");
        return KABC_ERR_INVALID_ARG;
    }
    Rccl* R;
    if (kabc_status_t st = need_rccl(&R)) return st;
    KABC_HIP_CHECK(hipSetDevice(c->ctx->device));
    hipStream_t s = c->ctx->stream;
    KABC_HIP_CHECK(hipMemcpyAsync(c->d_scratch, inout, 8 * (size_t)n, hipMemcpyHostToDevice, s));
    KABC_NCCL_CHECK(R, R->AllReduce(c->d_scratch, c->d_scratch, (size_t)n, dt, op,
                                    (ncclComm_t)c->nccl, s));
    KABC_HIP_CHECK(hipMemcpyAsync(inout, c->d_scratch, 8 * (size_t)n, hipMemcpyDeviceToHost, s));
    KABC_HIP_CHECK(hipStreamSynchronize(s));
    return KABC_OK;
}

kabc_status_t kabc_comm_allreduce_sum_u64(kabc_comm_t* c, uint64_t* inout, int32_t n) {
    return host_allreduce(c, inout, n, ncclUint64, ncclSum);
}
kabc_status_t kabc_comm_allreduce_max_f64(kabc_comm_t* c, double* inout, int32_t n) {
    return host_allreduce(c, inout, n, ncclDouble, ncclMax);
}
kabc_status_t kabc_comm_barrier(kabc_comm_t* c) {
    uint64_t one = 1;
    return kabc_comm_allreduce_sum_u64(c, &one, 1);
}

kabc_status_t kabc_comm_destroy(kabc_comm_t* c) {
    if (!c) return KABC_OK;
    (void)hipSetDevice(c->ctx->device);
    (void)hipStreamSynchronize(c->ctx->stream);
    if (c->nccl) {
        if (Rccl* R = rccl()) (void)R->CommDestroy((ncclComm_t)c->nccl);
    }
    if (c->grp) {
        P2PGroup* g = c->grp;
        if (g->ev[c->rank]) (void)hipEventDestroy(g->ev[c->rank]);
        g->ev[c->rank] = nullptr;
        g->member[c->rank] = nullptr;
        if (--g->refs == 0) delete g;
    }
    if (c->d_scratch) (void)hipFree(c->d_scratch);
    if (c->xstream) (void)hipStreamSynchronize(c->xstream);
    for (int k = 0; k < KABC_MAX_EXCHANGE_CHUNKS; ++k)
        if (c->ev_chunk[k]) (void)hipEventDestroy(c->ev_chunk[k]);
    if (c->ev_done) (void)hipEventDestroy(c->ev_done);
    if (c->xstream) (void)hipStreamDestroy(c->xstream);
    if (c->own_ctx) (void)kabc_ctx_destroy(c->ctx);
    delete c;
    return KABC_OK;
}

}  // extern "C"
