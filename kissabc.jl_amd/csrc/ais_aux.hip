// ais_aux.hip -- launcher of the prepared-cost pre-pass (ais_aux_kernels.hpp)
#include "ais_aux_kernels.hpp"

namespace kabc {

constexpr int kAuxBlock = 256;

// KABC_COST_NORMAL_MEANSTD_SIM: aux = (mean, standard deviation) of the n standard normals
__global__ void __launch_bounds__(kAuxBlock) aux_normal_meanstd_kernel(const AuxArgs A) {
    __shared__ __attribute__((aligned(16))) double slogtab[KABC_MATH_TAB_WORDS];
    static_assert(KABC_MATH_TAB_WORDS == 2 * kAuxBlock, "two table words per thread");
    slogtab[threadIdx.x] = kabc_log_tab[threadIdx.x];
    slogtab[threadIdx.x + kAuxBlock] = kabc_log_tab[threadIdx.x + kAuxBlock];
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & (kWave - 1);
    const int64_t item = (int64_t)blockIdx.x * (kAuxBlock / kWave) + wave;
    if (item >= A.rows * A.nt) return;
    if (A.skip_if && *A.skip_if) return;
    const int64_t s = item / A.rows, r = item - s * A.rows;
    const uint64_t seed = A.seeds ? A.seeds[blockIdx.y] : A.seed;
    double* aux = A.aux + (A.seeds ? (int64_t)blockIdx.y * A.stride_aux : 0);
    const uint64_t t0 = A.t_dev ? (uint64_t)*A.t_dev + 1u : A.t0;
    kabc_cost_rng_t rng = {seed, t0 + (uint64_t)s, A.id_base + (uint32_t)(A.row_first + r),
                           A.domain ? A.domain : KABC_DOM_AIS_COST, 0u, 0u, nullptr, slogtab};
    const int n = (int)A.cost_params[0];
    static_assert(KABC_SIM_LANES == kWave, "one slice of the draws per lane");
    double sz, szz;
    kabc_cost_normal_meanstd_slice(n, lane, &rng, &sz, &szz);
    // the pairwise tree of the contract, as an xor-butterfly: lane 0 holds a[0] of the sequential form
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        sz = sz + __shfl_xor(sz, off, kWave);
        szz = szz + __shfl_xor(szz, off, kWave);
    }
    if (lane == 0) {
        const int64_t ws = A.word_stride ? A.word_stride : A.rows;
        double m[2];
        kabc_cost_normal_meanstd_moments(n, sz, szz, m);
        const int64_t sl = A.ring > 0 ? (int64_t)((t0 + (uint64_t)s) % (uint64_t)A.ring) : s;
        aux[(sl * 2 + 0) * ws + r] = m[0];
        aux[(sl * 2 + 1) * ws + r] = m[1];
    }
}

void launch_aux_prepass(int cost_id, const AuxArgs& a, hipStream_t s, unsigned nchains) {
    const int64_t items = a.rows * a.nt;
    if (items <= 0) return;
    const unsigned grid = (unsigned)((items + kAuxBlock / kWave - 1) / (kAuxBlock / kWave));
    if (cost_id == KABC_COST_NORMAL_MEANSTD_SIM)
        hipLaunchKernelGGL(aux_normal_meanstd_kernel, dim3(grid, nchains), dim3(kAuxBlock), 0, s, a);
}

}  // namespace kabc
