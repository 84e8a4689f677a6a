// atomic_scope_probe.hip -- what the loop kernel's window-histogram atomics cost in front of its barrier:
// 128 workgroups x 256 threads, ~30 % of the threads add 1 to one of 1024 bins of a global histogram, then
// wait for their memory operations (s_waitcnt vmcnt(0)) -- with agent-scope atomics (performed at the memory
// side: coherent across the XCDs) and with workgroup-scope atomics into one histogram PER XCD (performed in
// the XCD's L2).  Prints the mean and maximum wait per workgroup in ns (s_memrealtime, 100 MHz).
//   hipcc --offload-arch=gfx950 -O3 tools/atomic_scope_probe.hip -o /tmp/atomic_scope_probe && /tmp/atomic_scope_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ unsigned xcc_id() {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 0xfu;
}

template <int MODE>  // 0 agent scope, one histogram; 1 workgroup scope, a histogram per XCD; 2 no atomics (stores only)
__global__ void __launch_bounds__(256) probe(unsigned* hist, double* rows, unsigned long long* out, int rounds) {
    const unsigned gid = blockIdx.x * 256 + threadIdx.x;
    const unsigned x = xcc_id();
    unsigned long long acc = 0;
    for (int r = 0; r < rounds; ++r) {
        // the pass's own stores: a 128-byte row per thread
        double* row = rows + (size_t)gid * 16;
        for (int k = 0; k < 16; ++k) row[k] = (double)(r + k);
        unsigned h = (gid + r * 7919u) * 2654435761u;
        const bool in_window = (h >> 8) % 10u < 3u;
        const unsigned bin = (h >> 12) & 1023u;
        __syncthreads();
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        if (in_window) {
            if (MODE == 0) __hip_atomic_fetch_add(&hist[bin], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (MODE == 1) __hip_atomic_fetch_add(&hist[x * 1024u + bin], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        acc += __builtin_amdgcn_s_memrealtime() - t0;
    }
    if (threadIdx.x == 0) out[blockIdx.x] = acc;
}

int main() {
    const int G = 128, rounds = 200;
    unsigned* hist;
    double* rows;
    unsigned long long* out;
    hipMalloc(&hist, sizeof(unsigned) * 16 * 1024);
    hipMalloc(&rows, sizeof(double) * 16 * G * 256);
    hipMalloc(&out, sizeof(unsigned long long) * G);
    std::vector<unsigned long long> h(G);
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            hipMemset(hist, 0, sizeof(unsigned) * 16 * 1024);
            if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(G), dim3(256), 0, 0, hist, rows, out, rounds);
            if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(G), dim3(256), 0, 0, hist, rows, out, rounds);
            if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(G), dim3(256), 0, 0, hist, rows, out, rounds);
            hipDeviceSynchronize();
        }
        hipMemcpy(h.data(), out, sizeof(unsigned long long) * G, hipMemcpyDeviceToHost);
        double mean = 0, mx = 0;
        for (int b = 0; b < G; ++b) {
            const double ns = (double)h[b] * 10.0 / rounds;
            mean += ns / G;
            mx = ns > mx ? ns : mx;
        }
        std::vector<unsigned> hh(16 * 1024);
        hipMemcpy(hh.data(), hist, sizeof(unsigned) * 16 * 1024, hipMemcpyDeviceToHost);
        unsigned long long total = 0;
        for (unsigned v : hh) total += v;
        printf("{\"mode\": \"%s\", \"wait_ns_mean\": %.0f, \"wait_ns_max_workgroup\": %.0f, \"counted\": %llu}\n",
               mode == 0 ? "agent scope, one histogram" : mode == 1 ? "workgroup scope, histogram per XCD" : "row stores only",
               mean, mx, total);
    }
    return 0;
}
