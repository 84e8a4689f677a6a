// ais_aux_kernels.hpp -- the parameter-independent part of a simulator cost ("prepared cost",
// include/kabc_costs.h) for EVERY (walker, sub-step) of a half-generation launch, computed by a
// grid-wide pre-pass: ONE WAVEFRONT PER COST EVALUATION, its 64 lanes sharing the evaluation's
// independent draws.
//
// Why: the reference's own usage is a tiny ensemble with an expensive simulator -- README.md:31-57,
// AIS(10), 1000 normals per cost evaluation.  Inside the half-generation kernel that is five
// busy lanes each running 500 Philox + Box-Muller pairs one after the other, 100 sub-steps in a
// row: the MI355X was slower than one host core (profiles/r02_small_ensembles.json).  The draws
// of sub-step s do not depend on the walker's state (counter-based stream keyed by (walker, t)),
// so all nt x rows evaluations of a launch are independent of each other AND of the chain, and
// each splits 64 ways by the summation order the cost's contract defines
// (kabc_cost_normal_meanstd_slice + the pairwise tree).  The half-generation kernel's producers
// then only copy the two words per (walker, sub-step) into their LDS records.
#pragma once

#include "kabc_device.hpp"

namespace kabc {

struct AuxArgs {
    double* aux;             // [chain][nt][W][rows]: word j of (sub-step s, owned row r)
    const double* cost_params;
    const double* cost_data;
    int64_t cost_ndata;
    int64_t row_first;       // first row of the launch's segment inside its half
    int64_t rows;            // rows of the segment
    uint64_t seed;
    uint64_t t0;             // transition counter of sub-step 0
    uint32_t id_base;        // global walker id of row 0 of the half
    int32_t nt;
    const uint64_t* seeds;   // batched chains (blockIdx.y = chain), else NULL
    int64_t stride_aux;      // doubles per chain
    // smc (kernel-per-phase path): the stream domain (0 = KABC_DOM_AIS_COST), the pass counter
    // read from the device (t = *t_dev + 1: the host enqueues passes without knowing which of them
    // still run), the distance between the words of one row (0 = rows) and the run's control block
    uint32_t domain;
    const unsigned long long* t_dev;
    int64_t word_stride;
    const int32_t* skip_if;  // the kernel is a no-op while *skip_if != 0 (smc: the loop is over)
    // smc, several passes ahead in one launch (nt = ring > 1): pass t goes to slot t mod ring of
    // the buffer ([ring][W][word_stride]) -- smc_mcmc_kernel looks its pass up the same way, so
    // no launch has to know how many of the prepared passes have been used (0: slot = sub-step)
    int32_t ring;
};

// does a grid-wide pre-pass exist for this cost?  (built-ins only: a user cost's prepare step is
// one sequential function, it stays with the half-generation kernel's producers)
inline int aux_prepass_words(int cost_id) { return cost_id == KABC_COST_NORMAL_MEANSTD_SIM ? 2 : 0; }
void launch_aux_prepass(int cost_id, const AuxArgs& a, hipStream_t s, unsigned nchains);

}  // namespace kabc
