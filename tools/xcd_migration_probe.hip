// Do the workgroups of a long-running (cooperative) kernel stay on the XCD they started on when
// several processes share the GPU?  The XCD-aware release barrier of smc_loop_kernel
// (csrc/smc_loop_kernel.hpp) reads HW_REG_XCC_ID once per launch.  Every workgroup polls its XCC id
// and its HW_ID (SE / CU) for `seconds` and counts the changes.
//   hipcc -O2 --offload-arch=gfx950 tools/xcd_migration_probe.hip -o tools/_bin/xcdmig
//   for i in 1 2 3 4; do tools/_bin/xcdmig 8 & done; wait
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ unsigned xcc_id() {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 0xfu;
}
__device__ __forceinline__ unsigned hw_id() {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(x));
    return x;
}

__global__ void __launch_bounds__(256) k(unsigned long long ticks, unsigned long long* out) {
    // out[0] XCD changes, out[1] CU/SE changes, out[2] polls
    if (threadIdx.x != 0) return;
    const unsigned x0 = xcc_id();
    unsigned xl = x0, hl = hw_id() & 0xffff0f00u;  // keep SE / SH / CU fields, drop wave / SIMD slots
    unsigned long long nx = 0, nh = 0, np = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
        const unsigned x = xcc_id(), h = hw_id() & 0xffff0f00u;
        nx += x != xl;
        nh += h != hl;
        xl = x;
        hl = h;
        ++np;
        __builtin_amdgcn_s_sleep(8);
    }
    atomicAdd(&out[0], nx);
    atomicAdd(&out[1], nh);
    atomicAdd(&out[2], np);
    if (xl != x0) atomicAdd(&out[3], 1ull);
}

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 5.0;
    unsigned long long* d;
    CK(hipMalloc(&d, 32));
    CK(hipMemset(d, 0, 32));
    unsigned long long ticks = (unsigned long long)(seconds * 100e6);  // s_memrealtime: 100 MHz
    void* args[] = {&ticks, &d};
    CK(hipLaunchCooperativeKernel((void*)k, dim3(128), dim3(256), args, 0, 0));
    CK(hipDeviceSynchronize());
    unsigned long long h[4];
    CK(hipMemcpy(h, d, 32, hipMemcpyDeviceToHost));
    printf("{\"workgroups\": 128, \"seconds\": %.1f, \"xcd_changes\": %llu, \"cu_changes\": %llu, \"polls\": %llu, \"workgroups_ending_on_another_xcd\": %llu}\n",
           seconds, h[0], h[1], h[2], h[3]);
    return 0;
}
