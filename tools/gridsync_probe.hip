// Cost of a device-wide barrier inside one cooperative launch on gfx950:
// cooperative_groups grid.sync() vs a hand-rolled sense-reversing atomic barrier,
// for the grid shapes a multi-workgroup SMC select would use.
//   hipcc -O2 --offload-arch=gfx950 tools/gridsync_probe.hip -o /tmp/gridsync && /tmp/gridsync
#include <hip/hip_cooperative_groups.h>
#include <hip/hip_runtime.h>
#include <cstdio>
namespace cg = cooperative_groups;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void k_cg(int iters, unsigned* sink) {
    cg::grid_group g = cg::this_grid();
    unsigned acc = 0;
    for (int i = 0; i < iters; ++i) {
        acc += threadIdx.x;
        g.sync();
    }
    if (acc == 0xffffffffu) sink[0] = acc;
}

// sense-reversing barrier: one atomic per workgroup, thread 0 spins on the generation word
__device__ __forceinline__ void grid_barrier(unsigned* count, volatile unsigned* gen, unsigned nblocks) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned g = *gen;
        __threadfence();
        if (atomicAdd(count, 1u) == nblocks - 1u) {
            atomicExch(count, 0u);
            __threadfence();
            atomicAdd((unsigned*)gen, 1u);
        } else {
            while (*gen == g) __builtin_amdgcn_s_sleep(1);
        }
        __threadfence();
    }
    __syncthreads();
}
__global__ void k_manual(int iters, unsigned* bar, unsigned* sink) {
    unsigned acc = 0;
    for (int i = 0; i < iters; ++i) {
        acc += threadIdx.x;
        grid_barrier(&bar[0], (volatile unsigned*)&bar[32], gridDim.x);
    }
    if (acc == 0xffffffffu) sink[0] = acc;
}

int main() {
    unsigned *sink, *bar;
    CK(hipMalloc(&sink, 64));
    CK(hipMalloc(&bar, 256));
    int coop = 0;
    CK(hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, 0));
    printf("{\"cooperative_launch_supported\": %d", coop);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int shapes[][2] = {{32, 1024}, {64, 512}, {128, 256}, {256, 256}};
    for (auto& sh : shapes) {
        int iters = 200;
        unsigned* sk = sink;
        void* a1[] = {&iters, &sk};
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            CK(hipLaunchCooperativeKernel((void*)k_cg, dim3(sh[0]), dim3(sh[1]), a1, 0, 0));
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
        }
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf(", \"cg_sync_us_%dx%d\": %.3f", sh[0], sh[1], ms * 1e3 / iters);
        CK(hipMemset(bar, 0, 256));
        void* a2[] = {&iters, &bar, &sk};
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            CK(hipLaunchCooperativeKernel((void*)k_manual, dim3(sh[0]), dim3(sh[1]), a2, 0, 0));
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
        }
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf(", \"manual_barrier_us_%dx%d\": %.3f", sh[0], sh[1], ms * 1e3 / iters);
    }
    // plain launch gap for comparison: 200 empty dependent kernels
    {
        int iters = 0;
        unsigned* sk = sink;
        CK(hipEventRecord(e0));
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_manual, dim3(32), dim3(1024), 0, 0, iters, bar, sk);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf(", \"empty_kernel_us\": %.3f", ms * 1e3 / 200);
    }
    printf("}\n");
    return 0;
}
