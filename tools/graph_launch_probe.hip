// Does a hipGraph shorten the boundary between DEPENDENT kernels of one stream on gfx950?
// The half-generation launches of AIS at ntransitions = 1 are 3.6 us of boundary + 4.3 us of kernel
// (profiles/r05_short_launch_ablation.txt).  Same kernel as tools/halfgen_floor_probe.hip's `launch`
// variant (wave 0 of every workgroup stores its row and loads a random row of the other half), G
// workgroups x 256 threads, `iters` launches that each depend on the one before:
//   stream : hipLaunchKernelGGL back to back on one stream
//   graph  : the same sequence captured once (hipStreamBeginCapture) and replayed (hipGraphLaunch)
//   hipcc -O2 --offload-arch=gfx950 tools/graph_launch_probe.hip -o /tmp/glp && /tmp/glp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void __launch_bounds__(256) k_half(double* act, const double* comp, unsigned rows, int it) {
    if (threadIdx.x >= 64) return;
    const unsigned r = blockIdx.x * 64u + threadIdx.x;
    if (r >= rows) return;
    unsigned j = (r * 2654435761u + (unsigned)it * 40503u) % rows;
    const double4* p = reinterpret_cast<const double4*>(comp + (size_t)j * 8);
    double4 a = p[0], b = p[1];
    double4* q = reinterpret_cast<double4*>(act + (size_t)r * 8);
    a.x += 1.0;
    q[0] = a;
    q[1] = b;
}

int main() {
    const int iters = 400;
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("{\"unit\": \"us per dependent launch\"");
    for (unsigned G : {32u, 128u, 512u}) {
        const unsigned rows = G * 64u;
        double *h0, *h1;
        CK(hipMalloc(&h0, (size_t)rows * 64));
        CK(hipMalloc(&h1, (size_t)rows * 64));
        CK(hipMemset(h0, 0, (size_t)rows * 64));
        CK(hipMemset(h1, 0, (size_t)rows * 64));
        auto enqueue = [&]() {
            for (int it = 0; it < iters; ++it)
                hipLaunchKernelGGL(k_half, dim3(G), dim3(256), 0, s, (it & 1) ? h1 : h0, (it & 1) ? h0 : h1, rows, it);
        };
        float best_s = 1e9f, best_g = 1e9f;
        enqueue();
        CK(hipStreamSynchronize(s));
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(e0, s));
            enqueue();
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            best_s = ms < best_s ? ms : best_s;
        }
        hipGraph_t graph;
        hipGraphExec_t exec;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        enqueue();
        CK(hipStreamEndCapture(s, &graph));
        CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        CK(hipGraphLaunch(exec, s));
        CK(hipStreamSynchronize(s));
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(e0, s));
            CK(hipGraphLaunch(exec, s));
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            best_g = ms < best_g ? ms : best_g;
        }
        printf(", \"stream_%u\": %.3f, \"graph_%u\": %.3f", G, best_s * 1e3 / iters, G, best_g * 1e3 / iters);
        CK(hipGraphExecDestroy(exec));
        CK(hipGraphDestroy(graph));
        CK(hipFree(h0));
        CK(hipFree(h1));
    }
    printf("}\n");
    return 0;
}
