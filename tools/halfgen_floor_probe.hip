// Floor of ONE half-generation inside a persistent (cooperative) AIS kernel on gfx950, without
// any arithmetic: what the data dependence "every walker of half h+1 may read every row half h
// just wrote" costs when it is paid with a device-wide barrier instead of a kernel boundary.
//   hipcc -O2 --offload-arch=gfx950 tools/halfgen_floor_probe.hip -o /tmp/hgfloor && /tmp/hgfloor
// Variants per grid shape (G workgroups x 256 threads, wave 0 = the "consumer" of a batch of 64):
//   bar     : barrier only (release fence, one atomic, poll, acquire fence)
//   rw      : + the consumer stores its 64-byte row before the barrier and loads one random
//             row of the other half after it (the first partner fetch of the next half)
//   rw_dep2 : + a second, dependent random row (own-row reload then partner)
//   launch  : the same read/write as `rw`, one ordinary kernel launch per half instead
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

struct Bar {
    unsigned long long count;
    unsigned pad0[30];
    unsigned gen;
    unsigned pad1[31];
};

__device__ __forceinline__ void barrier(Bar* b, unsigned G, unsigned nb) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(&b->count, 1ull) + 1ull == (unsigned long long)(nb + 1u) * G) {
            __threadfence();
            atomicExch(&b->gen, nb + 1u);
        }
        volatile unsigned* gen = &b->gen;
        unsigned spins = 0;
        while (*gen < nb + 1u) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 22)) break;  // bounded
        }
        __threadfence();
    }
    __syncthreads();
}

// two-level arrival: `NG` group counters on their own cache lines (group = workgroup mod NG: with
// round-robin dispatch that is the XCD), the last arrival of a group bumps the top counter
struct Bar2 {
    struct { unsigned long long count; unsigned pad[30]; } grp[16];
    unsigned long long top;
    unsigned pad0[30];
    unsigned gen;
    unsigned pad1[31];
};
template <int NG>
__device__ __forceinline__ void barrier2(Bar2* b, unsigned G, unsigned nb) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned g = blockIdx.x % NG;
        const unsigned members = G / NG + (g < G % NG ? 1u : 0u);
        __threadfence();
        if (atomicAdd(&b->grp[g].count, 1ull) + 1ull == (unsigned long long)(nb + 1u) * members) {
            if (atomicAdd(&b->top, 1ull) + 1ull == (unsigned long long)(nb + 1u) * NG) {
                __threadfence();
                atomicExch(&b->gen, nb + 1u);
            }
        }
        volatile unsigned* gen = &b->gen;
        unsigned spins = 0;
        while (*gen < nb + 1u) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 22)) break;
        }
        __threadfence();
    }
    __syncthreads();
}
template <int NG>
__global__ void __launch_bounds__(256) k_bar2(int iters, Bar2* bar, double* sink) {
    double acc = 0;
    for (int it = 0; it < iters; ++it) {
        acc += threadIdx.x;
        barrier2<NG>(bar, gridDim.x, (unsigned)it);
    }
    if (acc == -1.0) sink[0] = acc;
}

// variants of the plain barrier: SLEEP = s_sleep argument between polls (1 = 64 cycles);
// NF > 1: the releasing workgroup writes NF generation words on their own cache lines and a
// workgroup polls word (blockIdx mod NF) -- fewer pollers per line
struct BarF {
    unsigned long long count;
    unsigned pad0[30];
    struct { unsigned gen; unsigned pad[31]; } f[16];
};
template <int SLEEP, int NF>
__device__ __forceinline__ void barrier_v(BarF* b, unsigned G, unsigned nb) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(&b->count, 1ull) + 1ull == (unsigned long long)(nb + 1u) * G) {
            __threadfence();
#pragma unroll
            for (int j = 0; j < NF; ++j) atomicExch(&b->f[j].gen, nb + 1u);
        }
        volatile unsigned* gen = &b->f[blockIdx.x % NF].gen;
        unsigned spins = 0;
        while (*gen < nb + 1u) {
            __builtin_amdgcn_s_sleep(SLEEP);
            if (++spins > (1u << 22)) break;
        }
        __threadfence();
    }
    __syncthreads();
}
template <int SLEEP, int NF>
__global__ void __launch_bounds__(256) k_barv(int iters, BarF* bar, double* sink) {
    double acc = 0;
    for (int it = 0; it < iters; ++it) {
        acc += threadIdx.x;
        barrier_v<SLEEP, NF>(bar, gridDim.x, (unsigned)it);
    }
    if (acc == -1.0) sink[0] = acc;
}
// no fences at all: what the arrival / release protocol alone costs
template <int SLEEP>
__global__ void __launch_bounds__(256) k_bar_nofence(int iters, BarF* b, double* sink) {
    double acc = 0;
    const unsigned G = gridDim.x;
    for (int it = 0; it < iters; ++it) {
        acc += threadIdx.x;
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned nb = (unsigned)it;
            if (__hip_atomic_fetch_add(&b->count, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1ull ==
                (unsigned long long)(nb + 1u) * G)
                __hip_atomic_store(&b->f[0].gen, nb + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned spins = 0;
            while (__hip_atomic_load(&b->f[0].gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nb + 1u) {
                __builtin_amdgcn_s_sleep(SLEEP);
                if (++spins > (1u << 22)) break;
            }
        }
        __syncthreads();
    }
    if (acc == -1.0) sink[0] = acc;
}

__device__ __forceinline__ unsigned mix(unsigned a, unsigned b) {
    unsigned h = a * 0x9E3779B1u + b * 0x85EBCA77u;
    h ^= h >> 15;
    h *= 0xC2B2AE3Du;
    h ^= h >> 13;
    return h;
}

template <int MODE>
__global__ void __launch_bounds__(256) k_persist(double* h0, double* h1, unsigned rows, int iters, Bar* bar,
                                                 double* sink) {
    const unsigned G = gridDim.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned row = blockIdx.x * 64u + lane;
    double x[8];
    for (int k = 0; k < 8; ++k) x[k] = (double)(row + k);
    for (int it = 0; it < iters; ++it) {
        double* act = (it & 1) ? h1 : h0;
        const double* comp = (it & 1) ? h0 : h1;
        if (MODE >= 1 && wave == 0 && row < rows) {
            unsigned p = mix(row, (unsigned)it) % rows;
            if (MODE >= 2) {
                const double2* r = reinterpret_cast<const double2*>(act + (size_t)row * 8);
                double2 a = r[0];
                p = (p + (unsigned)(a.x * 0.0)) % rows;  // dependent chain
            }
            const double2* r = reinterpret_cast<const double2*>(comp + (size_t)p * 8);
            double2 a = r[0], b = r[1], c = r[2], d = r[3];
            x[0] += a.x * 1e-9; x[1] += a.y * 1e-9; x[2] += b.x * 1e-9; x[3] += b.y * 1e-9;
            x[4] += c.x * 1e-9; x[5] += c.y * 1e-9; x[6] += d.x * 1e-9; x[7] += d.y * 1e-9;
            double2* w = reinterpret_cast<double2*>(act + (size_t)row * 8);
            w[0] = make_double2(x[0], x[1]);
            w[1] = make_double2(x[2], x[3]);
            w[2] = make_double2(x[4], x[5]);
            w[3] = make_double2(x[6], x[7]);
        }
        barrier(bar, G, (unsigned)it);
    }
    if (x[0] == -1.0) sink[0] = x[0];
}

__global__ void __launch_bounds__(256) k_half(double* act, const double* comp, unsigned rows, int it) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned row = blockIdx.x * 64u + lane;
    if (wave == 0 && row < rows) {
        const unsigned p = mix(row, (unsigned)it) % rows;
        double2* w = reinterpret_cast<double2*>(act + (size_t)row * 8);
        const double2* r = reinterpret_cast<const double2*>(comp + (size_t)p * 8);
        double2 a = r[0], b = r[1], c = r[2], d = r[3];
        double2 o0 = w[0];
        w[0] = make_double2(o0.x + a.x * 1e-9, o0.y + a.y * 1e-9);
        w[1] = make_double2(b.x, b.y);
        w[2] = make_double2(c.x, c.y);
        w[3] = make_double2(d.x, d.y);
    }
}

template <int MODE>
static int run(const char* name, unsigned G, double* h0, double* h1, Bar* bar, double* sink, hipEvent_t e0,
               hipEvent_t e1) {
    unsigned rows = G * 64u;
    int iters = 400;
    float ms = 0.f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(bar, 0, sizeof(Bar)));
        void* args[] = {&h0, &h1, &rows, &iters, &bar, &sink};
        CK(hipEventRecord(e0));
        CK(hipLaunchCooperativeKernel((void*)k_persist<MODE>, dim3(G), dim3(256), args, 0, 0));
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    printf(", \"%s_%u\": %.3f", name, G, ms * 1e3 / iters);
    return 0;
}

int main() {
    double *h0, *h1, *sink;
    Bar* bar;
    const size_t nb = (size_t)65536 * 8 * 8;
    CK(hipMalloc(&h0, nb));
    CK(hipMalloc(&h1, nb));
    CK(hipMemset(h0, 0, nb));
    CK(hipMemset(h1, 0, nb));
    CK(hipMalloc(&sink, 64));
    CK(hipMalloc(&bar, sizeof(Bar)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("{\"unit\": \"us per half-generation\"");
    for (unsigned G : {32u, 64u, 128u, 256u, 512u}) {
        if (run<0>("bar", G, h0, h1, bar, sink, e0, e1)) return 1;
        if (run<1>("rw", G, h0, h1, bar, sink, e0, e1)) return 1;
        if (run<2>("rw_dep2", G, h0, h1, bar, sink, e0, e1)) return 1;
        {
            Bar2* b2;
            CK(hipMalloc(&b2, sizeof(Bar2)));
            int iters2 = 400;
            float ms2 = 0.f;
            for (int v = 0; v < 2; ++v) {
                for (int rep = 0; rep < 3; ++rep) {
                    CK(hipMemset(b2, 0, sizeof(Bar2)));
                    void* args[] = {&iters2, &b2, &sink};
                    CK(hipEventRecord(e0));
                    CK(hipLaunchCooperativeKernel(v == 0 ? (void*)k_bar2<8> : (void*)k_bar2<16>, dim3(G), dim3(256), args, 0, 0));
                    CK(hipEventRecord(e1));
                    CK(hipEventSynchronize(e1));
                    CK(hipEventElapsedTime(&ms2, e0, e1));
                }
                printf(", \"bar2level%d_%u\": %.3f", v == 0 ? 8 : 16, G, ms2 * 1e3 / iters2);
            }
            CK(hipFree(b2));
        }
        {
            BarF* bf;
            CK(hipMalloc(&bf, sizeof(BarF)));
            int itv = 400;
            float msv = 0.f;
            struct { const char* name; void* fn; } vs[] = {
                {"bar_sleep4", (void*)k_barv<4, 1>}, {"bar_sleep16", (void*)k_barv<16, 1>},
                {"bar_sleep1_8flags", (void*)k_barv<1, 8>}, {"bar_sleep4_8flags", (void*)k_barv<4, 8>},
                {"bar_sleep4_16flags", (void*)k_barv<4, 16>}, {"bar_nofence_sleep1", (void*)k_bar_nofence<1>},
                {"bar_nofence_sleep4", (void*)k_bar_nofence<4>}};
            for (auto& v : vs) {
                for (int rep = 0; rep < 3; ++rep) {
                    CK(hipMemset(bf, 0, sizeof(BarF)));
                    void* args[] = {&itv, &bf, &sink};
                    CK(hipEventRecord(e0));
                    CK(hipLaunchCooperativeKernel(v.fn, dim3(G), dim3(256), args, 0, 0));
                    CK(hipEventRecord(e1));
                    CK(hipEventSynchronize(e1));
                    CK(hipEventElapsedTime(&msv, e0, e1));
                }
                printf(", \"%s_%u\": %.3f", v.name, G, msv * 1e3 / itv);
            }
            CK(hipFree(bf));
        }
        // ordinary launches
        const unsigned rows = G * 64u;
        const int iters = 400;
        float ms = 0.f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            for (int it = 0; it < iters; ++it)
                hipLaunchKernelGGL(k_half, dim3(G), dim3(256), 0, 0, (it & 1) ? h1 : h0, (it & 1) ? h0 : h1, rows, it);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
        }
        printf(", \"launch_%u\": %.3f", G, ms * 1e3 / iters);
    }
    printf("}\n");
    return 0;
}
