// Dependent-issue latency probe for gfx950: ns per instruction of ONE dependency chain
// per wave, at 1, 2 and 4 waves per SIMD -- what a latency-bound instruction stream
// (Philox rounds, Horner polynomials) achieves at the occupancy the AIS kernel runs at.
//   hipcc -O2 --offload-arch=gfx950 tools/valu_latency.hip -o /tmp/valu_latency && /tmp/valu_latency
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define ITERS 4096
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

#define CHAIN64(NAME, ASM)                                                        \
    __global__ void NAME(double* out, double seed) {                              \
        double a = seed + threadIdx.x;                                            \
        const double b = 1.0000001, c = 1e-9;                                     \
        for (int i = 0; i < ITERS; ++i) {                                         \
            asm volatile(ASM : "+v"(a) : "v"(b), "v"(c));                         \
            asm volatile(ASM : "+v"(a) : "v"(b), "v"(c));                         \
            asm volatile(ASM : "+v"(a) : "v"(b), "v"(c));                         \
            asm volatile(ASM : "+v"(a) : "v"(b), "v"(c));                         \
            asm volatile(ASM : "+v"(a) : "v"(b), "v"(c));                         \
            asm volatile(ASM : "+v"(a) : "v"(b), "v"(c));                         \
            asm volatile(ASM : "+v"(a) : "v"(b), "v"(c));                         \
            asm volatile(ASM : "+v"(a) : "v"(b), "v"(c));                         \
        }                                                                         \
        if (a == 12345.678) out[0] = a;                                           \
    }
#define CHAIN32(NAME, ASM)                                                        \
    __global__ void NAME(double* out, double seed) {                              \
        uint32_t a = (uint32_t)seed + threadIdx.x;                                \
        const uint32_t b = 0xD2511F53u, c = 0x9E3779B9u;                          \
        for (int i = 0; i < ITERS; ++i) {                                         \
            asm volatile(ASM : "+v"(a) : "v"(b), "v"(c));                         \
            asm volatile(ASM : "+v"(a) : "v"(b), "v"(c));                         \
            asm volatile(ASM : "+v"(a) : "v"(b), "v"(c));                         \
            asm volatile(ASM : "+v"(a) : "v"(b), "v"(c));                         \
            asm volatile(ASM : "+v"(a) : "v"(b), "v"(c));                         \
            asm volatile(ASM : "+v"(a) : "v"(b), "v"(c));                         \
            asm volatile(ASM : "+v"(a) : "v"(b), "v"(c));                         \
            asm volatile(ASM : "+v"(a) : "v"(b), "v"(c));                         \
        }                                                                         \
        if (a == 12345u) out[0] = a;                                              \
    }
__global__ void c_mad_u64_u32(double* out, double seed) {
    uint64_t a = (uint64_t)seed + threadIdx.x;
    const uint32_t b = 0xD2511F53u;
    for (int i = 0; i < ITERS; ++i) {
#define MADC asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(a) : "v"(b) : "vcc")
        MADC; MADC; MADC; MADC; MADC; MADC; MADC; MADC;
    }
    if (a == 12345u) out[0] = (double)a;
}
CHAIN32(c_add_f32, "v_add_f32 %0, %0, %1")
CHAIN32(c_xor3, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96")
CHAIN64(c_add_f64, "v_add_f64 %0, %0, %2")
CHAIN64(c_fma_f64, "v_fma_f64 %0, %0, %1, %2")
CHAIN64(c_rcp_f64, "v_rcp_f64 %0, %0")

typedef void (*kern_t)(double*, double);

int main() {
    double* d;
    CK(hipMalloc(&d, 64));
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    struct { const char* name; kern_t k; } ks[] = {{"v_add_f32", c_add_f32}, {"v_bitop3_b32", c_xor3},
        {"v_mad_u64_u32", c_mad_u64_u32}, {"v_add_f64", c_add_f64}, {"v_fma_f64", c_fma_f64},
        {"v_rcp_f64", c_rcp_f64}};
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("{\"cus\": %d, \"ns_per_dependent_instr\": {", cus);
    for (size_t i = 0; i < sizeof ks / sizeof ks[0]; ++i) {
        printf("%s\"%s\": {", i ? ", " : "", ks[i].name);
        const int wps[3] = {1, 2, 4};
        for (int w = 0; w < 3; ++w) {
            const int blocks = cus * wps[w];  // 256-thread workgroups: one wave per SIMD each
            hipLaunchKernelGGL(ks[i].k, dim3(blocks), dim3(256), 0, 0, d, 1.0);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(ks[i].k, dim3(blocks), dim3(256), 0, 0, d, 1.0);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            // per wave: ITERS * 8 dependent instructions per launch
            printf("%s\"waves_per_simd_%d\": %.3f", w ? ", " : "", wps[w], ms * 1e6 / (5.0 * ITERS * 8));
        }
        printf("}");
    }
    printf("}}\n");
    return 0;
}
