// VALU issue-rate probe for gfx950: SIMD cycles per wave64 instruction for the
// instruction classes the AIS producers/consumer are made of.  Each kernel runs 8
// independent dependency chains of one instruction per lane, 8 waves per SIMD, so the
// figure is throughput, not latency.  Reported relative to v_add_f32 (4 cycles/wave64).
//   hipcc -O2 --offload-arch=gfx950 tools/valu_rate.hip -o gpurun_out/valu_rate && gpurun_out/valu_rate
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define ITERS 2048
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

#define KERNEL64(NAME, ASM)                                                                   \
    __global__ void NAME(double* out, double seed) {                                          \
        double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4,   \
               a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;                                         \
        const double b = 1.0000001, c = 1e-9;                                                 \
        for (int i = 0; i < ITERS; ++i) {                                                     \
            asm volatile(ASM : "+v"(a0) : "v"(b), "v"(c));                                    \
            asm volatile(ASM : "+v"(a1) : "v"(b), "v"(c));                                    \
            asm volatile(ASM : "+v"(a2) : "v"(b), "v"(c));                                    \
            asm volatile(ASM : "+v"(a3) : "v"(b), "v"(c));                                    \
            asm volatile(ASM : "+v"(a4) : "v"(b), "v"(c));                                    \
            asm volatile(ASM : "+v"(a5) : "v"(b), "v"(c));                                    \
            asm volatile(ASM : "+v"(a6) : "v"(b), "v"(c));                                    \
            asm volatile(ASM : "+v"(a7) : "v"(b), "v"(c));                                    \
        }                                                                                     \
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678) out[0] = a0;                  \
    }

#define KERNEL32(NAME, ASM)                                                                   \
    __global__ void NAME(double* out, double seed) {                                          \
        uint32_t a0 = (uint32_t)seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3,   \
                 a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;                          \
        const uint32_t b = 0xD2511F53u, c = 0x9E3779B9u;                                      \
        for (int i = 0; i < ITERS; ++i) {                                                     \
            asm volatile(ASM : "+v"(a0) : "v"(b), "v"(c));                                    \
            asm volatile(ASM : "+v"(a1) : "v"(b), "v"(c));                                    \
            asm volatile(ASM : "+v"(a2) : "v"(b), "v"(c));                                    \
            asm volatile(ASM : "+v"(a3) : "v"(b), "v"(c));                                    \
            asm volatile(ASM : "+v"(a4) : "v"(b), "v"(c));                                    \
            asm volatile(ASM : "+v"(a5) : "v"(b), "v"(c));                                    \
            asm volatile(ASM : "+v"(a6) : "v"(b), "v"(c));                                    \
            asm volatile(ASM : "+v"(a7) : "v"(b), "v"(c));                                    \
        }                                                                                     \
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345u) out[0] = a0;                     \
    }

// 64-bit integer accumulator chains for v_mad_u64_u32 (d = s0*s1 + d)
__global__ void k_mad_u64_u32(double* out, double seed) {
    uint64_t a0 = (uint64_t)seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4,
             a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const uint32_t b = 0xD2511F53u;
    uint32_t c = 0x9E3779B9u + threadIdx.x;
    for (int i = 0; i < ITERS; ++i) {
#define MAD(A) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(A) : "v"(b), "v"(c) : "vcc")
        MAD(a0); MAD(a1); MAD(a2); MAD(a3); MAD(a4); MAD(a5); MAD(a6); MAD(a7);
    }
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345u) out[0] = (double)a0;
}

KERNEL32(k_add_f32, "v_add_f32 %0, %0, %1")
KERNEL32(k_xor3, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96")
KERNEL32(k_mul_lo_u32, "v_mul_lo_u32 %0, %0, %1")
KERNEL32(k_mul_hi_u32, "v_mul_hi_u32 %0, %0, %1")
KERNEL32(k_mul_u32_u24, "v_mul_u32_u24 %0, %0, %1")
KERNEL32(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL32(k_cndmask_e64_vcc, "v_cndmask_b32_e64 %0, %0, %1, vcc")
KERNEL32(k_cndmask_sdwa, "v_cndmask_b32_sdwa %0, %0, %1, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD")
KERNEL32(k_addc_u32, "v_addc_co_u32 %0, vcc, %0, %1, vcc")
KERNEL32(k_cndmask_sgpr, "v_cndmask_b32_e64 %0, %0, %1, s[10:11]")
KERNEL32(k_mov_b32, "v_mov_b32 %0, %1")
KERNEL32(k_and_b32, "v_and_b32 %0, %0, %1")
KERNEL32(k_lshl_b32, "v_lshlrev_b32 %0, 1, %0")
KERNEL32(k_add_u32, "v_add_u32 %0, %0, %1")
KERNEL32(k_add_co_u32, "v_add_co_u32 %0, vcc, %0, %1")
KERNEL32(k_cmp_u32, "v_cmp_lt_u32 vcc, %0, %1")
KERNEL32(k_cmp_u32_sgpr, "v_cmp_lt_u32_e64 s[10:11], %0, %1")
KERNEL32(k_fma_f32, "v_fma_f32 %0, %0, %1, %2")
KERNEL32(k_readlane, "v_readlane_b32 s10, %0, 3")
KERNEL64(k_cmp_f64, "v_cmp_lt_f64 vcc, %0, %1")
KERNEL64(k_cmp_class_f64, "v_cmp_class_f64 vcc, %0, 3")
__global__ void k_cvt_f64_u32(double* out, double seed) {
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;
    const uint32_t b = (uint32_t)seed + threadIdx.x;
    for (int i = 0; i < ITERS; ++i) {
#define CVT(A) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(A) : "v"(b))
        CVT(a0); CVT(a1); CVT(a2); CVT(a3); CVT(a4); CVT(a5); CVT(a6); CVT(a7);
    }
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678) out[0] = a0;
}
__global__ void k_cvt_u32_f64(double* out, double seed) {
    uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;
    const double b = seed + threadIdx.x;
    for (int i = 0; i < ITERS; ++i) {
#define CVTU(A) asm volatile("v_cvt_u32_f64 %0, %1" : "=v"(A) : "v"(b))
        CVTU(a0); CVTU(a1); CVTU(a2); CVTU(a3); CVTU(a4); CVTU(a5); CVTU(a6); CVTU(a7);
    }
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345u) out[0] = a0;
}
KERNEL64(k_lshl_add_u64, "v_lshl_add_u64 %0, %0, 1, %1")
KERNEL64(k_lshlrev_b64, "v_lshlrev_b64 %0, 1, %0")
KERNEL64(k_div_fixup_f64, "v_div_fixup_f64 %0, %0, %1, %2")
KERNEL64(k_div_fmas_f64, "v_div_fmas_f64 %0, %0, %1, %2")
KERNEL64(k_div_scale_f64, "v_div_scale_f64 %0, vcc, %0, %1, %2")
KERNEL64(k_max_f64, "v_max_f64 %0, %0, %1")
KERNEL64(k_pk_add_f32, "v_pk_add_f32 %0, %0, %1")
KERNEL64(k_add_f64, "v_add_f64 %0, %0, %2")
KERNEL64(k_mul_f64, "v_mul_f64 %0, %0, %1")
KERNEL64(k_fma_f64, "v_fma_f64 %0, %0, %1, %2")
KERNEL64(k_rcp_f64, "v_rcp_f64 %0, %0")
KERNEL64(k_sqrt_f64, "v_sqrt_f64 %0, %0")
KERNEL64(k_rsq_f64, "v_rsq_f64 %0, %0")
KERNEL64(k_rndne_f64, "v_rndne_f64 %0, %0")
KERNEL64(k_ldexp_f64, "v_ldexp_f64 %0, %0, 1")
KERNEL64(k_mov_b64, "v_mov_b64 %0, %1")

// one select among seven adds: is the VOP2 v_cndmask cost intrinsic or a back-to-back effect?
#define MIXK(NAME, SEL)                                                                       \
    __global__ void NAME(double* out, double seed) {                                          \
        uint32_t a0 = (uint32_t)seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3,   \
                 a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;                          \
        const uint32_t b = 0xD2511F53u, c = 0x9E3779B9u;                                      \
        for (int i = 0; i < ITERS; ++i) {                                                     \
            asm volatile("v_add_f32 %0, %0, %1" : "+v"(a0) : "v"(b), "v"(c));                 \
            asm volatile("v_add_f32 %0, %0, %1" : "+v"(a1) : "v"(b), "v"(c));                 \
            asm volatile("v_add_f32 %0, %0, %1" : "+v"(a2) : "v"(b), "v"(c));                 \
            asm volatile(SEL : "+v"(a3) : "v"(b), "v"(c));                                    \
            asm volatile("v_add_f32 %0, %0, %1" : "+v"(a4) : "v"(b), "v"(c));                 \
            asm volatile("v_add_f32 %0, %0, %1" : "+v"(a5) : "v"(b), "v"(c));                 \
            asm volatile("v_add_f32 %0, %0, %1" : "+v"(a6) : "v"(b), "v"(c));                 \
            asm volatile("v_add_f32 %0, %0, %1" : "+v"(a7) : "v"(b), "v"(c));                 \
        }                                                                                     \
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345u) out[0] = a0;                     \
    }
#define MIXN(NAME, S0, S1, S2, S3, S4, S5, S6, S7)                                           \
    __global__ void NAME(double* out, double seed) {                                          \
        uint32_t a0 = (uint32_t)seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3,   \
                 a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;                          \
        const uint32_t b = 0xD2511F53u, c = 0x9E3779B9u;                                      \
        for (int i = 0; i < ITERS; ++i) {                                                     \
            asm volatile(S0 : "+v"(a0) : "v"(b), "v"(c));                                     \
            asm volatile(S1 : "+v"(a1) : "v"(b), "v"(c));                                     \
            asm volatile(S2 : "+v"(a2) : "v"(b), "v"(c));                                     \
            asm volatile(S3 : "+v"(a3) : "v"(b), "v"(c));                                     \
            asm volatile(S4 : "+v"(a4) : "v"(b), "v"(c));                                     \
            asm volatile(S5 : "+v"(a5) : "v"(b), "v"(c));                                     \
            asm volatile(S6 : "+v"(a6) : "v"(b), "v"(c));                                     \
            asm volatile(S7 : "+v"(a7) : "v"(b), "v"(c));                                     \
        }                                                                                     \
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345u) out[0] = a0;                     \
    }
#define ADDI "v_add_f32 %0, %0, %1"
#define SEL32 "v_cndmask_b32 %0, %0, %1, vcc"
#define SEL64 "v_cndmask_b32_e64 %0, %0, %1, vcc"
MIXN(k_mix2_e32, SEL32, SEL32, ADDI, ADDI, ADDI, ADDI, ADDI, ADDI)
MIXN(k_mix4_e32, SEL32, SEL32, SEL32, SEL32, ADDI, ADDI, ADDI, ADDI)
MIXN(k_mix4_e64, SEL64, SEL64, SEL64, SEL64, ADDI, ADDI, ADDI, ADDI)
MIXK(k_mix_e32, "v_cndmask_b32 %0, %0, %1, vcc")
MIXK(k_mix_e64, "v_cndmask_b32_e64 %0, %0, %1, vcc")

typedef void (*kern_t)(double*, double);

int main(int argc, char** argv) {
    double* d;
    CK(hipMalloc(&d, 64));
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    // default: 8 workgroups of 256 = 32 waves per CU = 8 per SIMD (throughput);
    // argv[1] = waves per SIMD (1 shows what ONE wave with 8 independent chains can issue)
    const int wps = argc > 1 ? atoi(argv[1]) : 8;
    const int blocks = cus * wps;
    struct { const char* name; kern_t k; } ks[] = {
        {"v_add_f32", k_add_f32},       {"v_bitop3_b32", k_xor3},       {"v_cndmask_b32", k_cndmask},
        {"mix_6add_2cndmask_e32", k_mix2_e32}, {"mix_4add_4cndmask_e32", k_mix4_e32},
        {"mix_4add_4cndmask_e64", k_mix4_e64},
        {"mix_7add_1cndmask_e32", k_mix_e32}, {"mix_7add_1cndmask_e64", k_mix_e64},
        {"v_cndmask_b32_e64_vcc", k_cndmask_e64_vcc}, {"v_cndmask_b32_sdwa", k_cndmask_sdwa},
        {"v_addc_co_u32", k_addc_u32},
        {"v_cndmask_b32_e64_sgpr", k_cndmask_sgpr}, {"v_mov_b32", k_mov_b32}, {"v_and_b32", k_and_b32},
        {"v_lshlrev_b32", k_lshl_b32}, {"v_add_u32", k_add_u32}, {"v_add_co_u32", k_add_co_u32},
        {"v_cmp_lt_u32_vcc", k_cmp_u32}, {"v_cmp_lt_u32_sgpr", k_cmp_u32_sgpr}, {"v_fma_f32", k_fma_f32},
        {"v_readlane_b32", k_readlane}, {"v_cmp_lt_f64", k_cmp_f64}, {"v_cmp_class_f64", k_cmp_class_f64},
        {"v_cvt_f64_u32", k_cvt_f64_u32}, {"v_cvt_u32_f64", k_cvt_u32_f64},
        {"v_lshl_add_u64", k_lshl_add_u64}, {"v_lshlrev_b64", k_lshlrev_b64},
        {"v_div_fixup_f64", k_div_fixup_f64}, {"v_div_fmas_f64", k_div_fmas_f64},
        {"v_div_scale_f64", k_div_scale_f64}, {"v_max_f64", k_max_f64}, {"v_pk_add_f32", k_pk_add_f32},
        {"v_mul_u32_u24", k_mul_u32_u24}, {"v_mul_lo_u32", k_mul_lo_u32}, {"v_mul_hi_u32", k_mul_hi_u32},
        {"v_mad_u64_u32", k_mad_u64_u32}, {"v_mov_b64", k_mov_b64},       {"v_add_f64", k_add_f64},
        {"v_mul_f64", k_mul_f64},       {"v_fma_f64", k_fma_f64},       {"v_ldexp_f64", k_ldexp_f64},
        {"v_rndne_f64", k_rndne_f64},   {"v_rcp_f64", k_rcp_f64},       {"v_rsq_f64", k_rsq_f64},
        {"v_sqrt_f64", k_sqrt_f64}};
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    double base = 0;
    printf("{\"cus\": %d, \"clock_mhz\": %d, \"waves_per_simd\": %d, \"rates\": {", cus, p.clockRate / 1000, wps);
    for (size_t i = 0; i < sizeof ks / sizeof ks[0]; ++i) {
        hipLaunchKernelGGL(ks[i].k, dim3(blocks), dim3(256), 0, 0, d, 1.0);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(ks[i].k, dim3(blocks), dim3(256), 0, 0, d, 1.0);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        // wave-instructions per SIMD: 8 waves * ITERS * 8 chains * 5 launches
        const double per_simd = (double)wps * ITERS * 8 * 5;
        const double ns_per_instr = ms * 1e6 / per_simd;
        if (i == 0) base = ns_per_instr;
        printf("%s\"%s\": {\"ns_per_wave_instr_per_simd\": %.3f, \"cycles_if_add_f32_is_4\": %.2f}",
               i ? ", " : "", ks[i].name, ns_per_instr, 4.0 * ns_per_instr / base);
    }
    printf("}}\n");
    return 0;
}
