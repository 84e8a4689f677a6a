// host_common.hpp -- host-side helpers of the C-ABI library (not part of the ABI)
#pragma once
#include <cstdlib>

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <utility>
#include <vector>

#include "kabc_device.hpp"

namespace kabc {

void set_error(const char* fmt, ...);
const char* get_error();

#define KABC_HIP_CHECK(expr)                                                              \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess) {                                                           \
            kabc::set_error("HIP error %d (%s) at %s:%d: %s", (int)_e, hipGetErrorString(_e), \
                            __FILE__, __LINE__, #expr);                                   \
            return KABC_ERR_DEVICE;                                                       \
        }                                                                                 \
    } while (0)

// derived constants of one Factored component.  Host libm supplies the one-off
// normalisers (lgamma, erfc); everything evaluated per walker goes through the
// math contract.  Returns false for invalid parameters.
// hipMalloc for the library's working buffers.  KABC_POISON_ALLOC=1 (tests) fills every buffer --
// fresh or recycled from a context's pool -- with 0xA5 bytes first: fresh device memory usually
// reads as zero, so a kernel that relies on that passes every test until the driver hands out a
// recycled page (several processes starting on one GPU).
inline bool poison_alloc() {
    static const bool on = [] {
        const char* e = std::getenv("KABC_POISON_ALLOC");
        return e && *e && *e != '0';
    }();
    return on;
}
// (hipMemset on the null stream is not ordered against the contexts' non-blocking streams and may
// return before it has run: wait for it, or it lands on top of the run's own initialisation)
inline hipError_t poison_fill(void* p, size_t bytes) {
    hipError_t e = hipMemset(p, 0xA5, bytes);
    return e == hipSuccess ? hipDeviceSynchronize() : e;
}
template <class T>
inline hipError_t dev_malloc(T** p, size_t bytes) {
    hipError_t e = hipMalloc((void**)p, bytes);
    if (e == hipSuccess && poison_alloc()) e = poison_fill((void*)*p, bytes);
    return e;
}

// copy of the caller's components with the library-side fields of MvNormal components filled in
// (device block pointer, D); every entry point resolves before it prepares or copies the prior
kabc_status_t resolve_priors(kabc_ctx_t* ctx, const kabc_prior_t* prior, int D, kabc_prior_t* out);
bool prepare_prior(const kabc_prior_t& pr, PriorDev& q);
bool prepare_priors(const kabc_prior_t* prior, int D, PriorSet& out);

}  // namespace kabc

struct kabc_ctx {
    int device;
    hipStream_t stream;
    bool own_stream;
    // scratch-buffer cache of the run-to-completion entry points (kabc_smc_run, kabc_pfilter_run):
    // a C4 smc run allocates ~15 device buffers; hipMalloc / hipFree per call cost more than a
    // tenth of the run.  Buffers return here instead of to the driver and are handed out again
    // (best fit) by the next call on this context; released by kabc_ctx_destroy.
    std::mutex pool_mu;
    std::vector<std::pair<size_t, void*>> pool;
    size_t pool_bytes = 0;
};

// ---- communicators (capi_comm.hip) ------------------------------------------------
namespace kabc {
struct P2PGroup;  // single-process peer group shared by the communicators of one init_all
}
struct kabc_comm {
    kabc_ctx_t* ctx;
    int32_t rank, world, backend;
    bool own_ctx;
    bool single_process;   // created by kabc_comm_init_all
    void* nccl;            // ncclComm_t (RCCL backend)
    kabc::P2PGroup* grp;   // P2P backend
    void* d_scratch;       // small device buffer for the host-value reductions
    // pipelined exchange (more than one exchange chunk per half): the all-gathers run on their
    // own stream behind one event per chunk; created at first use
    hipStream_t xstream;
    hipEvent_t ev_chunk[KABC_MAX_EXCHANGE_CHUNKS];  // kernels of chunk k of the current half are done
    hipEvent_t ev_done;    // every all-gather issued so far has completed on this rank
};

namespace kabc {
// in-place all-gather of `count` doubles per rank inside `base` ([world][count]) on the
// communicator's stream; one-process-per-GPU RCCL communicators only
kabc_status_t comm_allgather_inplace(kabc_comm* c, double* base, size_t count);
// the same for the n communicators of one kabc_comm_init_all call: bases[i] is the buffer of
// comms[i]; RCCL: one ncclGroup; P2P: every rank pulls the peers' segments after their kernels
kabc_status_t comm_allgather_inplace_multi(kabc_comm** comms, double** bases, int n, size_t count);
// ---- pipelined exchange: chunk k of a half is gathered on the exchange stream while the
// kernels of chunk k + 1 run on the context stream ------------------------------------
// records "the kernels of chunk k are done" on the context stream(s) and gathers the chunk
// ([world][count] doubles at base / bases[i]) on the exchange stream(s) behind it
kabc_status_t comm_exchange_chunk(kabc_comm* c, double* base, size_t count, int k, hipEvent_t t0 = nullptr,
                                  hipEvent_t t1 = nullptr);
kabc_status_t comm_exchange_chunk_multi(kabc_comm** comms, double** bases, int n, size_t count, int k);
// the context stream(s) wait until every gather issued so far has landed (before the next
// half-generation reads the gathered half; `all_ranks`: also until no peer still reads this
// rank's rows -- before the host may touch them)
kabc_status_t comm_exchange_fence(kabc_comm* c);
// In-place all-gather of n buffers at once on the context stream ([world][count[j]] doubles at
// bases[j], this rank's segment at rank * count[j]): the sharded cost loop of smc.  RCCL: one
// group; P2P communicators of a single-process group: every rank is driven by its OWN host
// thread and the threads meet here (host rendezvous, pull kernels, synchronous).
kabc_status_t comm_allgather_many(kabc_comm* c, double** bases, const size_t* counts, int n);
kabc_status_t comm_exchange_fence_multi(kabc_comm** comms, int n, bool all_ranks);
}  // namespace kabc
