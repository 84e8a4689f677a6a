// capi_abcde.hip -- kabc_abcde_run: ABCDE(prior, cost, ϵ_target; ...) of
// src/smc.jl:347-430.  All generations are enqueued without a host round trip; the
// earlystop break is taken on the device (kernels after it are no-ops).
#include <cmath>
#include <vector>

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
constexpr int64_t kDonorMinN = 256;  // below this one lane per particle scans inside the generation kernel

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

// ---- the cost order: an LSD radix sort of (cost key, particle index) written for this path ---
// (replaces hipcub::DeviceRadixSort: no device library call is left on the path).  8-bit digits over
// the order-preserving 64-bit key of the cost, three launches per digit:
//   rs_hist    per tile of 2048 keys, the count of every digit value          counts[256][G]
//   rs_scan    exclusive prefix over (digit value, tile) -- one workgroup
//   rs_scatter every key to offset[digit][tile] + its STABLE rank inside the tile: rounds of 256
//              consecutive positions; inside a wavefront the lanes holding the same digit are found
//              with eight ballots (match-any), the leader of each group publishes the group's size,
//              earlier wavefronts of the round and earlier rounds of the tile are added from LDS
// LSD needs every pass stable; nothing here depends on the order atomics retire in.
constexpr int kRsBlock = 256;
constexpr int kRsItems = 8;
constexpr int kRsTile = kRsBlock * kRsItems;
constexpr int kRsBuckets = 256;

__device__ __forceinline__ unsigned long long rs_key_of(double x) {  // total order; -0.0 folds onto +0.0
    const unsigned long long u = kabc_bits(x + 0.0);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ULL);
}
__device__ __forceinline__ double rs_val_of(unsigned long long k) {
    const unsigned long long u = (k >> 63) ? (k & 0x7fffffffffffffffULL) : ~k;
    return kabc_from_bits(u);
}
__global__ void __launch_bounds__(256) rs_keys_kernel(const double* delta, int64_t n, unsigned long long* keys,
                                                      unsigned* vals) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        keys[i] = rs_key_of(delta[i]);
        vals[i] = (unsigned)i;
    }
}
__global__ void __launch_bounds__(256) rs_values_kernel(const unsigned long long* keys, int64_t n, double* sorted) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) sorted[i] = rs_val_of(keys[i]);
}
__global__ void __launch_bounds__(kRsBlock) rs_hist_kernel(const unsigned long long* keys, int64_t n, int shift,
                                                           unsigned* counts, unsigned G) {
    __shared__ unsigned s_h[kRsBuckets];
    s_h[threadIdx.x] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * kRsTile;
#pragma unroll
    for (int r = 0; r < kRsItems; ++r) {
        const int64_t p = base + (int64_t)r * kRsBlock + threadIdx.x;
        if (p < n) atomicAdd(&s_h[(unsigned)(keys[p] >> shift) & 255u], 1u);
    }
    __syncthreads();
    counts[(size_t)threadIdx.x * G + blockIdx.x] = s_h[threadIdx.x];
}
// exclusive prefix, in place, over m = 256 * G counts in (digit value, tile) order
__global__ void __launch_bounds__(1024) rs_scan_kernel(unsigned* counts, int64_t m) {
    __shared__ unsigned s_w[16];
    __shared__ unsigned s_run;
    if (threadIdx.x == 0) s_run = 0;
    __syncthreads();
    for (int64_t i0 = 0; i0 < m; i0 += 1024) {
        const int64_t i = i0 + threadIdx.x;
        const unsigned c = i < m ? counts[i] : 0u;
        const unsigned incl = wave_scan_incl(c);
        if ((threadIdx.x & 63) == 63) s_w[threadIdx.x >> 6] = incl;
        __syncthreads();
        unsigned off = s_run, tot = 0;
        for (int q = 0; q < 16; ++q) {
            if (q < (int)(threadIdx.x >> 6)) off += s_w[q];
            tot += s_w[q];
        }
        if (i < m) counts[i] = off + incl - c;
        __syncthreads();
        if (threadIdx.x == 0) s_run += tot;
        __syncthreads();
    }
}
__global__ void __launch_bounds__(kRsBlock) rs_scatter_kernel(const unsigned long long* kin, const unsigned* vin,
                                                              unsigned long long* kout, unsigned* vout, int64_t n,
                                                              int shift, const unsigned* offsets, unsigned G) {
    __shared__ unsigned s_off[kRsBuckets];                  // next free slot of every digit value
    __shared__ unsigned s_wc[kRsBlock / kWave][kRsBuckets];  // this round's group sizes per wavefront
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wid = tid >> 6;
    s_off[tid] = offsets[(size_t)tid * G + blockIdx.x];
#pragma unroll
    for (int w = 0; w < kRsBlock / kWave; ++w) s_wc[w][tid] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * kRsTile;
    const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (int r = 0; r < kRsItems; ++r) {
        const int64_t p = base + (int64_t)r * kRsBlock + tid;
        const bool in = p < n;
        const unsigned long long key = in ? kin[p] : 0ull;
        const unsigned val = in ? vin[p] : 0u;
        const unsigned d = (unsigned)(key >> shift) & 255u;
        // lanes of this wavefront with the same digit (and in range)
        unsigned long long peers = __ballot(in);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const unsigned long long m = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? m : ~m;
        }
        const unsigned before = (unsigned)__popcll(peers & below);
        if (in && before == 0u) s_wc[wid][d] = (unsigned)__popcll(peers);  // the group's leader
        __syncthreads();
        if (in) {
            unsigned dst = s_off[d] + before;
            for (int w = 0; w < wid; ++w) dst += s_wc[w][d];
            kout[dst] = key;
            vout[dst] = val;
        }
        __syncthreads();
        unsigned t = 0;
#pragma unroll
        for (int w = 0; w < kRsBlock / kWave; ++w) {
            t += s_wc[w][tid];
            s_wc[w][tid] = 0;
        }
        s_off[tid] += t;
        __syncthreads();
    }
}
// stable partition of a wavelet-matrix level by its bit: zeros first, then ones, each in order --
// straight from the level's own rank words (wm_bits / wm_count above)
__global__ void __launch_bounds__(256) wm_partition_kernel(const unsigned* sin, unsigned* sout, int64_t n, int b,
                                                           const unsigned long long* bits, const unsigned* cnt,
                                                           const unsigned* nz) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const unsigned v = sin[p];
    const int64_t w = p >> 6;
    const unsigned l = (unsigned)(p & 63);
    const unsigned long long m = l ? (~0ull >> (64 - l)) : 0ull;
    const unsigned ones = cnt[w] + (unsigned)__popcll(bits[w] & m);
    sout[((v >> b) & 1u) ? (int64_t)*nz + ones : p - ones] = v;
}

// ---- the donor draw of medium ensembles (1536 .. 32 768 particles): TWO launches per generation
// The rank structure above costs ~68 dependent launches per generation (an 8-pass radix sort of
// 128-512 KB of keys, then a wavelet-matrix level per index bit): 242 us per generation at 16 384
// particles, nearly all of it kernel boundaries.  The donor draw
//     s = rand((1:N)[Δs .<= Δs[i]])          (src/smc.jl:392)
// is "the m-th index, in ascending order, among the particles whose cost does not exceed mine, m
// uniform below their count".  With the indices cut into BLOCKS of 256 consecutive particles and
// every block's costs sorted (one workgroup each, a bitonic network in LDS -- no dependency
// between workgroups), one WAVEFRONT answers a particle's draw: every lane bisects the sorted costs
// of its share of the blocks (count of costs <= mine per block), a wavefront prefix over the block
// counts gives the total and the block holding the m-th hit, and the 64 lanes scan that block's
// 256 costs in index order with four ballots.  Same count, same m, same index as the scans and
// the wavelet matrix.
constexpr int kDbBlock = 256;                 // particles per sorted block
constexpr int64_t kDbMinN = 1536;             // below: teams of sixteen lanes over the costs in LDS (measured
                                              // crossover, profiles/r05_abcde_blocks_from.txt: 1000: 1.16 vs 1.23 ms per 50
                                              // generations, 2000: 1.58 vs 1.27, 4000: 2.45 vs 1.33)
constexpr int64_t kDbMaxN = 32768;            // beyond: the wavelet matrix (measured crossover: 32 768: 1.8 vs 3.1 ms per 10 generations, 65 536: 5.5 vs 4.2)
constexpr int kDbMaxPerLane = (int)(kDbMaxN / kDbBlock / kWave);

// sorted[b * 256 + r] = r-th smallest cost of particles [256 b, 256 b + 256) (+Inf past N)
__global__ void __launch_bounds__(kDbBlock) abcde_blocksort_kernel(const AbcdeArgs A, double* sorted) {
    __shared__ double s_v[kDbBlock];
    if (A.ctrl->done) return;
    const int tid = threadIdx.x;
    const int64_t i = (int64_t)blockIdx.x * kDbBlock + tid;
    s_v[tid] = i < A.N ? A.delta[A.ctrl->cur][i] : KABC_INF;
    __syncthreads();
    for (int k = 2; k <= kDbBlock; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            const int p = tid ^ j;
            if (p > tid) {
                const double a = s_v[tid], b = s_v[p];
                const bool up = (tid & k) == 0;
                if (up ? (b < a) : (a < b)) {
                    s_v[tid] = b;
                    s_v[p] = a;
                }
            }
            __syncthreads();
        }
    }
    sorted[i] = s_v[tid];  // (the array is padded to whole blocks)
}

// donor[i] for every particle that needs one: a wavefront per particle
__global__ void __launch_bounds__(256) abcde_donor_blocks_kernel(const AbcdeArgs A, const double* sorted,
                                                                 int32_t* donor) {
    if (A.ctrl->done) return;
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t i = (int64_t)blockIdx.x * (256 / kWave) + (threadIdx.x >> 6);
    if (i >= A.N) return;  // (wave-uniform)
    const int64_t N = A.N;
    const double* __restrict__ DL = A.delta[A.ctrl->cur];
    const double di = DL[i];
    const bool skip = A.earlystop && di <= A.eps_target;                        // :384-386
    const double eps = (di <= A.eps_target) ? A.eps_target : A.ctrl->eps_pop;   // :390
    if (skip || !(di > eps)) {  // (wave-uniform: one particle per wavefront)
        if (lane == 0) donor[i] = (int32_t)i;
        return;
    }
    const int nblocks = (int)((N + kDbBlock - 1) / kDbBlock);
    const int per = (nblocks + kWave - 1) / kWave;  // consecutive blocks per lane (<= kDbMaxPerLane)
    // count of costs <= di in each of this lane's blocks: upper bound by bisection
    int cb[kDbMaxPerLane];
    int c = 0;
#pragma unroll
    for (int q = 0; q < kDbMaxPerLane; ++q) {
        cb[q] = 0;
        const int b = lane * per + q;
        if (q < per && b < nblocks) {
            const double* __restrict__ sb = sorted + (size_t)b * kDbBlock;
            int lo = 0, hi = kDbBlock;  // 257 possible answers: nine steps
#pragma unroll
            for (int step = 0; step < 9; ++step) {
                if (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (sb[mid] <= di) lo = mid + 1;
                    else hi = mid;
                }
            }
            cb[q] = lo;
            c += lo;
        }
    }
    int incl = c;
    for (int off = 1; off < kWave; off <<= 1) {
        const int o = __shfl_up(incl, off, kWave);
        if (lane >= off) incl += o;
    }
    const int total = __shfl(incl, kWave - 1, kWave);  // >= 1: the particle's own cost counts
    const kabc_u128_t B0 = kabc_stream_block(A.seed, (uint32_t)i, (uint64_t)A.ctrl->iters, 0u, KABC_DOM_ABCDE_MOVE);
    const int64_t m = (int64_t)kabc_index(kabc_lo64(B0), (uint64_t)total);
    // the lane whose blocks hold the m-th hit names the block and the hit's rank inside it
    const int excl = incl - c;
    int tb = -1, tm = 0;
    if (m >= excl && m < incl) {
        int r = (int)m - excl;
#pragma unroll
        for (int q = 0; q < kDbMaxPerLane; ++q) {
            if (tb < 0 && q < per) {
                if (r < cb[q]) {
                    tb = lane * per + q;
                    tm = r;
                }
                r -= cb[q];
            }
        }
    }
    const unsigned long long owner = __ballot(tb >= 0);  // exactly one lane
    const int src = __ffsll((long long)owner) - 1;
    tb = __shfl(tb, src, kWave);
    tm = __shfl(tm, src, kWave);
    // the tm-th particle of block tb, in index order, whose cost does not exceed di
    const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    int found = -1;
#pragma unroll
    for (int q = 0; q < kDbBlock / kWave; ++q) {
        const int64_t j = (int64_t)tb * kDbBlock + q * kWave + lane;
        const bool hit = j < N && DL[j] <= di;
        const unsigned long long mk = __ballot(hit);
        const int cnt = __popcll(mk);
        if (found < 0 && tm >= 0 && tm < cnt && hit && __popcll(mk & below) == tm) found = (int)j;
        tm -= cnt;  // (negative from the group after the hit on: no later group matches)
    }
    const unsigned long long fm = __ballot(found >= 0);
    if (fm && lane == __ffsll((long long)fm) - 1) donor[i] = found;
}

struct RankStructure {
    double* sorted = nullptr;
    unsigned *seq[2] = {nullptr, nullptr}, *cnt = nullptr, *nz = nullptr;
    unsigned long long* bits = nullptr;
    unsigned long long* keys[2] = {nullptr, nullptr};
    unsigned* vals[2] = {nullptr, nullptr};
    unsigned* counts = nullptr;  // [256][G]
    unsigned G = 0;
    int levels = 0;
    int64_t words = 0;
};

// sorted costs + wavelet matrix of the cost-sorted particle order, all on stream s
static hipError_t build_rank(const RankStructure& R, const double* delta, int64_t N, hipStream_t s) {
    const unsigned g256 = (unsigned)((N + 255) / 256);
    hipLaunchKernelGGL(rs_keys_kernel, dim3(g256), dim3(256), 0, s, delta, N, R.keys[0], R.vals[0]);
    int cur = 0;
    for (int shift = 0; shift < 64; shift += 8) {
        hipLaunchKernelGGL(rs_hist_kernel, dim3(R.G), dim3(kRsBlock), 0, s, R.keys[cur], N, shift, R.counts, R.G);
        hipLaunchKernelGGL(rs_scan_kernel, dim3(1), dim3(1024), 0, s, R.counts, (int64_t)kRsBuckets * R.G);
        // (the last pass writes the particle order where the wavelet matrix starts from)
        unsigned* vout = shift == 56 ? R.seq[0] : R.vals[1 - cur];
        hipLaunchKernelGGL(rs_scatter_kernel, dim3(R.G), dim3(kRsBlock), 0, s, R.keys[cur], R.vals[cur],
                           R.keys[1 - cur], vout, N, shift, R.counts, R.G);
        cur ^= 1;
    }
    hipLaunchKernelGGL(rs_values_kernel, dim3(g256), dim3(256), 0, s, R.keys[cur], N, R.sorted);
    int sc = 0;
    const unsigned gw = (unsigned)((R.words * 64 + 255) / 256);
    for (int b = R.levels - 1; b >= 0; --b) {
        unsigned long long* bits = R.bits + (size_t)b * R.words;
        hipLaunchKernelGGL(wm_bits_kernel, dim3(gw), dim3(256), 0, s, R.seq[sc], N, b, bits, R.words);
        hipLaunchKernelGGL(wm_count_kernel, dim3(1), dim3(1024), 0, s, bits, R.words, N,
                           R.cnt + (size_t)b * R.words, R.nz + b);
        if (b > 0) {  // stable partition by bit b: the sequence entering the next level
            hipLaunchKernelGGL(wm_partition_kernel, dim3(g256), dim3(256), 0, s, R.seq[sc], R.seq[1 - sc], N, b, bits,
                               R.cnt + (size_t)b * R.words, R.nz + b);
            sc ^= 1;
        }
    }
    return hipGetLastError();
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
    {
        const CostPlugin* pl = cost->id >= KABC_COST_USER ? find_plugin(cost->id) : nullptr;
        if (dyn && pl && !pl->rtc) {
            set_error("ABCDE with length(prior) = %d > %d: built-in DeviceCosts or a user cost in the hipRTC form "
                      "(kabc_compile_cost_plugin)", D, KABC_MAX_DIM);
            return KABC_ERR_UNSUPPORTED;
        }
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
    if (kabc_status_t st = model_unit_for(prior, D, cost->id, &unit)) return st;
    if (unit) {  // user prior families / a specialised model (plugin_registry.hpp)
        const PluginKernel ki = unit_kernel(unit, kPfAbcdeInit, D, 0), kg = unit_kernel(unit, kPfAbcdeGen, D, 0);
        if (ki.mod) f_init = AbcdeLaunch(ki.mod, &abcde_geom, (unsigned)kAbcdeBlock);
        if (kg.mod) f_gen = AbcdeLaunch(kg.mod, &abcde_geom, (unsigned)kAbcdeBlock);
        if ((!f_init || !f_gen) && unit_required(unit)) return KABC_ERR_DEVICE;
        // (a specialisation that is not there (yet): what is missing comes from below, same bits)
    }
    if (!f_init || !f_gen) {
        AbcdeLaunch b_init, b_gen;
        if (dyn && cost->id < KABC_COST_USER) {
            b_init = AbcdeLaunch(&l_init<0>);
            b_gen = AbcdeLaunch(&l_gen<0>);
        } else if (const CostPlugin* p = find_plugin(cost->id)) {
            const PluginKernel ki = plugin_kernel(p, kPfAbcdeInit, D, 0), kg = plugin_kernel(p, kPfAbcdeGen, D, 0);
            b_init = ki.host ? AbcdeLaunch((AbcdeLaunchFn)ki.host)
                             : ki.mod ? AbcdeLaunch(ki.mod, &abcde_geom, (unsigned)kAbcdeBlock) : AbcdeLaunch();
            b_gen = kg.host ? AbcdeLaunch((AbcdeLaunchFn)kg.host)
                            : kg.mod ? AbcdeLaunch(kg.mod, &abcde_geom, (unsigned)kAbcdeBlock) : AbcdeLaunch();
            if (!b_init || !b_gen) {
                set_error("cost plugin has no ABCDE kernels for D = %d", D);
                return KABC_ERR_UNSUPPORTED;
            }
        } else {
            b_init = pick_init(D, std::make_integer_sequence<int, KABC_MAX_DIM>{});
            b_gen = pick_gen(D, std::make_integer_sequence<int, KABC_MAX_DIM>{});
        }
        if (!f_init) f_init = b_init;
        if (!f_gen) f_gen = b_gen;
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
    // (KABC_ABCDE_RANK=wavelet | blocks: force one of the two structures from kRankMinN particles on)
    const char* rank_env = std::getenv("KABC_ABCDE_RANK");
    const bool force_wm = rank_env && rank_env[0] == 'w';
    int64_t blocks_from = kDbMinN;  // (KABC_ABCDE_BLOCKS_FROM: probes)
    if (const char* e = std::getenv("KABC_ABCDE_BLOCKS_FROM")) blocks_from = std::atoll(e) > 0 ? std::atoll(e) : blocks_from;
    const bool use_blocks = N >= blocks_from && !force_wm && N <= kDbMaxN;
    double* d_bsorted = nullptr;
    if (N >= kRankMinN && !use_blocks) {
        R.levels = 1;
        while ((1ll << R.levels) < N) ++R.levels;
        R.words = (N + 64) / 64;  // one word past position N (rank queries at p = N)
        KABC_HIP_CHECK(alloc((void**)&R.sorted, sizeof(double) * N));
        KABC_HIP_CHECK(alloc((void**)&R.seq[0], sizeof(unsigned) * N));
        KABC_HIP_CHECK(alloc((void**)&R.seq[1], sizeof(unsigned) * N));
        KABC_HIP_CHECK(alloc((void**)&R.bits, sizeof(unsigned long long) * R.levels * R.words));
        KABC_HIP_CHECK(alloc((void**)&R.cnt, sizeof(unsigned) * R.levels * R.words));
        KABC_HIP_CHECK(alloc((void**)&R.nz, sizeof(unsigned) * R.levels));
        R.G = (unsigned)((N + kRsTile - 1) / kRsTile);
        for (int b = 0; b < 2; ++b) {
            KABC_HIP_CHECK(alloc((void**)&R.keys[b], sizeof(unsigned long long) * N));
            KABC_HIP_CHECK(alloc((void**)&R.vals[b], sizeof(unsigned) * N));
        }
        KABC_HIP_CHECK(alloc((void**)&R.counts, sizeof(unsigned) * (size_t)kRsBuckets * R.G));
        A.sorted_delta = R.sorted;
        A.wm_bits = R.bits;
        A.wm_cnt = R.cnt;
        A.wm_nz = R.nz;
        A.wm_levels = R.levels;
        A.wm_words = R.words;
    }
    // in between (256 <= N < 1536): the donor draws by teams of sixteen lanes, one launch per
    // generation in front of the generation kernel (KABC_ABCDE_DONOR=0: the kernel's own scans)
    int32_t* d_donor = nullptr;
    {
        const char* e = std::getenv("KABC_ABCDE_DONOR");
        if (!R.sorted && !use_blocks && N >= kDonorMinN && N <= (int64_t)kAbcdeScanMax && !(e && e[0] == '0')) {
            KABC_HIP_CHECK(alloc((void**)&d_donor, sizeof(int32_t) * N));
            A.donor = d_donor;
        }
    }
    const unsigned db_blocks = (unsigned)((N + kDbBlock - 1) / kDbBlock);
    if (use_blocks) {
        KABC_HIP_CHECK(alloc((void**)&d_donor, sizeof(int32_t) * N));
        KABC_HIP_CHECK(alloc((void**)&d_bsorted, sizeof(double) * (size_t)db_blocks * kDbBlock));
        A.donor = d_donor;
    }
    for (int64_t g = 0; g < o->generations; ++g) {  // while iters < generations (:372)
        A.flip_first = g > 0 ? 1 : 0;  // (the flip after generation g - 1 rides on this launch)
        hipLaunchKernelGGL(abcde_extrema_kernel, dim3(1), dim3(1024), 0, s, A);
        if (use_blocks) {
            hipLaunchKernelGGL(abcde_blocksort_kernel, dim3(db_blocks), dim3(kDbBlock), 0, s, A, d_bsorted);
            hipLaunchKernelGGL(abcde_donor_blocks_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, s, A,
                               d_bsorted, d_donor);
        } else if (d_donor)
            hipLaunchKernelGGL(abcde_donor_kernel,
                               dim3((unsigned)((N + kDonorBlock / kDonorTeam - 1) / (kDonorBlock / kDonorTeam))),
                               dim3(kDonorBlock), 0, s, A, d_donor);
        // the buffer set flips once per generation until the earlystop break, after which every
        // kernel is a no-op: the host knows which one is current
        if (R.sorted) KABC_HIP_CHECK(build_rank(R, A.delta[g & 1], N, s));
        f_gen(A, s);
    }
    if (o->generations > 0) hipLaunchKernelGGL(abcde_flip_kernel, dim3(1), dim3(1), 0, s, A.ctrl);  // the last one
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
