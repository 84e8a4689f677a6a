// A device-wide barrier that pays the cross-XCD coherence ONCE PER XCD instead of once per
// workgroup (gfx950: 8 XCDs, one L2 each; memory is the only point the L2s agree on).
//
// The plain barrier (smc_loop_kernel.hpp) gives every workgroup an agent-scope release before
// it arrives (buffer_wbl2 sc1: write the XCD's whole L2 back) and an agent-scope acquire after
// it leaves (buffer_inv sc1: invalidate it): with 128 workgroups that is 16 write-backs and 16
// invalidates per XCD and barrier, and 55 % of the barrier's 5.3 us (tools/halfgen_floor_probe.hip).
// Here:
//   arrive : a workgroup only waits for its own stores (they are in its XCD's L2 then) and
//            counts itself on its XCD's counter; the LAST workgroup of an XCD writes that L2 back
//            (all its workgroups' dirty lines) and counts the XCD on the global counter; the
//            eighth XCD publishes the generation.
//   wait   : every workgroup polls the generation; the FIRST one of each XCD to see it
//            invalidates the XCD's L2 and publishes an XCD-local flag; the others wait for that
//            flag and invalidate only their CU's vector L1 (buffer_inv sc0).
// The XCD of a workgroup is read from HW_REG_XCC_ID; the members of each XCD are counted once at
// the start of the kernel (the dispatcher's placement is not assumed).
//
// This program checks the protocol (every workgroup writes rows that other workgroups read after
// the barrier, 2000 iterations, mismatches counted) and times it against the plain barrier.
//   hipcc -O2 --offload-arch=gfx950 tools/xcd_barrier_probe.hip -o /tmp/xcdbar && /tmp/xcdbar
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

struct Line {
    unsigned long long v;
    unsigned long long pad[15];
};
struct Bar {
    Line count;        // plain barrier: arrivals
    Line gen;          // generation word (plain barrier)
    Line gen2;         // generation word (XCD-aware barrier)
    Line members[16];  // workgroups per XCD
    Line xcount[16];   // arrivals per XCD
    Line gcount;       // XCDs that have arrived
    Line ticket[16];   // first workgroup of an XCD to leave
    Line flag[16];     // the XCD's L2 has been invalidated for generation v
    Line nxcd;         // XCDs that hold at least one workgroup
};

__device__ __forceinline__ unsigned xcc_id() {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 0xfu;
}
__device__ __forceinline__ unsigned long long ld(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool spin_until(const unsigned long long* p, unsigned long long want) {
    unsigned spins = 0;
    while (ld(p) < want) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1u << 22)) return false;  // bounded
    }
    return true;
}

__device__ __forceinline__ void plain_barrier(Bar* b, unsigned G, unsigned nb) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(&b->count.v, 1ull) + 1ull == (unsigned long long)(nb + 1u) * G) {
            __threadfence();
            atomicExch(&b->gen.v, (unsigned long long)(nb + 1u));
        }
        spin_until(&b->gen.v, nb + 1u);
        __threadfence();
    }
    __syncthreads();
}

// barrier number nb (0-based); xcd / members: this workgroup's XCD and its population
__device__ __forceinline__ void xcd_barrier(Bar* b, unsigned nb, unsigned xcd, unsigned members, unsigned nxcd) {
    // every wavefront's stores must be in this XCD's L2 before the workgroup is counted: an explicit
    // vmcnt wait -- __syncthreads() alone waits for LDS / scalar traffic only (the hole the
    // contention soak found in smc_loop_kernel: tools/contention_stress.py)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        if (atomicAdd(&b->xcount[xcd].v, 1ull) + 1ull == (unsigned long long)(nb + 1u) * members) {
            // last workgroup of this XCD: one write-back for all of them
            asm volatile("buffer_wbl2 sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
            if (atomicAdd(&b->gcount.v, 1ull) + 1ull == (unsigned long long)(nb + 1u) * nxcd)
                __hip_atomic_store(&b->gen2.v, (unsigned long long)(nb + 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        spin_until(&b->gen2.v, nb + 1u);
        if (atomicAdd(&b->ticket[xcd].v, 1ull) == (unsigned long long)nb * members) {
            // first workgroup of this XCD to leave: one L2 invalidate for all of them
            asm volatile("buffer_inv sc1" ::: "memory");
            __hip_atomic_store(&b->flag[xcd].v, (unsigned long long)(nb + 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            spin_until(&b->flag[xcd].v, nb + 1u);
        }
    }
    __syncthreads();
    asm volatile("buffer_inv sc0" ::: "memory");  // this CU's vector L1 (every wavefront)
}

__device__ __forceinline__ unsigned mix(unsigned a, unsigned b) {
    unsigned h = a * 0x9E3779B1u + b * 0x85EBCA77u;
    h ^= h >> 15;
    h *= 0xC2B2AE3Du;
    h ^= h >> 13;
    return h;
}

// MODE 0 plain barrier, 1 XCD-aware barrier with the L1-only invalidate (buffer_inv sc0) for the
// workgroups that do not invalidate the L2, 2 XCD-aware release side + buffer_inv sc1 in every
// workgroup.  The rows a thread checks are the SAME 16 per workgroup in every iteration: they
// stay in the CU's vector L1, so a barrier that does not invalidate it reads stale values.  Every iteration: wave 0 writes the workgroup's 64
// rows (value = f(iteration, row)), barrier, every thread reads a random row of another
// workgroup and checks it.
__device__ __forceinline__ void xcd_barrier_inv1(Bar* b, unsigned nb, unsigned xcd, unsigned members, unsigned nxcd) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        if (atomicAdd(&b->xcount[xcd].v, 1ull) + 1ull == (unsigned long long)(nb + 1u) * members) {
            asm volatile("buffer_wbl2 sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
            if (atomicAdd(&b->gcount.v, 1ull) + 1ull == (unsigned long long)(nb + 1u) * nxcd)
                __hip_atomic_store(&b->gen2.v, (unsigned long long)(nb + 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        spin_until(&b->gen2.v, nb + 1u);
        asm volatile("buffer_inv sc1" ::: "memory");
    }
    __syncthreads();
}

// MODE 3 (round 5): the release side without a returning atomic on the critical path.  Every workgroup counts
// itself on its XCD with a fire-and-forget atomic; ONE workgroup per XCD (the first that registered: the
// leader) polls that counter, writes the XCD's L2 back when it is full and counts the XCD on the global
// counter, again without waiting for the result; everybody polls the global counter.  The last workgroup's
// chain is store drain -> atomic -> (leader sees it) -> write-back -> atomic -> (pollers see it) instead of
// store drain -> atomic round trip -> write-back -> atomic round trip -> store -> (pollers see it).
__device__ __forceinline__ void xcd_barrier_leader(Bar* b, unsigned nb, unsigned xcd, unsigned members, unsigned nxcd,
                                                   bool leader) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        (void)__hip_atomic_fetch_add(&b->xcount[xcd].v, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (leader) {
            spin_until(&b->xcount[xcd].v, (unsigned long long)(nb + 1u) * members);
            asm volatile("buffer_wbl2 sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
            (void)__hip_atomic_fetch_add(&b->gcount.v, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        spin_until(&b->gcount.v, (unsigned long long)(nb + 1u) * nxcd);
        asm volatile("buffer_inv sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
}

template <int MODE>
__global__ void __launch_bounds__(256) k_check(double* buf0, double* buf1, unsigned rows, int iters, Bar* bar,
                                               unsigned long long* errors, unsigned* xcd_of) {
    const unsigned G = gridDim.x;
    unsigned xcd = 0, members = 0, nxcd = 0;
    unsigned nb = 0;
    __shared__ int s_leader;
    if (MODE >= 1) {
        xcd = xcc_id();
        if (threadIdx.x == 0) {
            s_leader = 0;
            if (atomicAdd(&bar->members[xcd].v, 1ull) == 0ull) {
                atomicAdd(&bar->nxcd.v, 1ull);
                s_leader = 1;
            }
            xcd_of[blockIdx.x] = xcd;
        }
        plain_barrier(bar, G, nb++);
        members = (unsigned)ld(&bar->members[xcd].v);
        nxcd = (unsigned)ld(&bar->nxcd.v);
    }
    unsigned long long bad = 0;
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (int it = 0; it < iters; ++it) {
        double* w = (it & 1) ? buf1 : buf0;
        if (wave == 0) {
            const unsigned row = blockIdx.x * 64u + lane;
            double2* p = reinterpret_cast<double2*>(w + (size_t)row * 8);
            const double v = (double)mix(row, (unsigned)it);
            p[0] = make_double2(v, v + 1.0);
            p[1] = make_double2(v + 2.0, v + 3.0);
            p[2] = make_double2(v + 4.0, v + 5.0);
            p[3] = make_double2(v + 6.0, v + 7.0);
        }
        if (MODE == 0) plain_barrier(bar, G, nb);
        else if (MODE == 1) xcd_barrier(bar, nb - 1u, xcd, members, nxcd);
        else if (MODE == 2) xcd_barrier_inv1(bar, nb - 1u, xcd, members, nxcd);
        else xcd_barrier_leader(bar, nb - 1u, xcd, members, nxcd, s_leader != 0);
        ++nb;
        const unsigned r = mix(blockIdx.x, threadIdx.x & 15u) % rows;  // L1-resident across iterations
        const double2* q = reinterpret_cast<const double2*>(w + (size_t)r * 8);
        const double2 a = q[0], d = q[3];
        const double v = (double)mix(r, (unsigned)it);
        if (a.x != v || a.y != v + 1.0 || d.x != v + 6.0 || d.y != v + 7.0) ++bad;
    }
    if (bad) atomicAdd(errors, bad);
}

int main() {
    double *b0, *b1;
    Bar* bar;
    unsigned long long* err;
    unsigned* xcd_of;
    const size_t nbuf = (size_t)512 * 64 * 8 * 8;
    CK(hipMalloc(&b0, nbuf));
    CK(hipMalloc(&b1, nbuf));
    CK(hipMalloc(&bar, sizeof(Bar)));
    CK(hipMalloc(&err, 8));
    CK(hipMalloc(&xcd_of, 512 * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("{\"unit\": \"us per iteration (rows written, barrier, random row of another workgroup read and checked)\"");
    for (unsigned G : {32u, 64u, 128u, 256u, 512u}) {
        unsigned rows = G * 64u;
        int iters = 2000;
        for (int mode = 0; mode < 4; ++mode) {
            float ms = 0.f;
            unsigned long long herr = 0;
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipMemset(bar, 0, sizeof(Bar)));
                CK(hipMemset(err, 0, 8));
                CK(hipMemset(b0, 0, nbuf));
                CK(hipMemset(b1, 0, nbuf));
                void* args[] = {&b0, &b1, &rows, &iters, &bar, &err, &xcd_of};
                CK(hipEventRecord(e0));
                CK(hipLaunchCooperativeKernel(mode == 0 ? (void*)k_check<0> : mode == 1 ? (void*)k_check<1> : mode == 2 ? (void*)k_check<2> : (void*)k_check<3>,
                                              dim3(G), dim3(256), args, 0, 0));
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
                CK(hipMemcpy(&herr, err, 8, hipMemcpyDeviceToHost));
            }
            const char* nm = mode == 0 ? "plain" : mode == 1 ? "xcd_l1inv_sc0" : mode == 2 ? "xcd_release_only" : "xcd_leader";
            printf(", \"%s_%u\": %.3f, \"%s_errors_%u\": %llu", nm, G, ms * 1e3 / iters, nm, G, herr);
            if (mode == 1) {
                unsigned h[512], cnt[16] = {};
                CK(hipMemcpy(h, xcd_of, G * 4, hipMemcpyDeviceToHost));
                for (unsigned i = 0; i < G; ++i) cnt[h[i] & 15]++;
                printf(", \"xcd_population_%u\": [%u,%u,%u,%u,%u,%u,%u,%u]", G, cnt[0], cnt[1], cnt[2], cnt[3], cnt[4],
                       cnt[5], cnt[6], cnt[7]);
            }
        }
    }
    printf("}\n");
    return 0;
}
