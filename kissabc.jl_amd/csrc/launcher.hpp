// launcher.hpp -- one callable type for "enqueue this kernel family with these arguments":
//   * a host launch function (kernels linked into the library, or into a plugin .so built by
//     hipcc -- hipLaunchKernelGGL inside), or
//   * a kernel of a code object compiled at run time by hipRTC (kabc_compile_cost_plugin,
//     capi_plugin.hip): hipModuleLaunchKernel with the family's launch geometry.
// The call sites look the same for both.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>

namespace kabc {

// hipModuleLaunchKernel / hipModuleLaunchCooperativeKernel of a kernel that takes ONE by-value
// argument struct (every kernel of this library does)
hipError_t rtc_launch(void* fn, dim3 grid, dim3 block, const void* args, hipStream_t s);
// the same with dynamic LDS (the run-time-dimension kernels keep their walkers' rows there)
hipError_t rtc_launch_lds(void* fn, dim3 grid, dim3 block, const void* args, hipStream_t s, unsigned lds_bytes);
// cooperative launch of G workgroups after an occupancy check (all co-resident or an error)
hipError_t rtc_launch_cooperative(void* fn, unsigned G, unsigned block, const void* args, hipStream_t s);

template <class Args, class... Extra>
struct Launcher {
    using Fn = void (*)(const Args&, hipStream_t, Extra...);
    using Geom = dim3 (*)(const Args&, Extra...);
    Fn fn = nullptr;      // host launch function
    void* mod = nullptr;  // hipFunction_t of a run-time compiled kernel
    Geom geom = nullptr;
    unsigned block = 0;
    Launcher() = default;
    Launcher(std::nullptr_t) {}
    Launcher(Fn f) : fn(f) {}
    Launcher(void* m, Geom g, unsigned b) : mod(m), geom(g), block(b) {}
    explicit operator bool() const { return fn != nullptr || mod != nullptr; }
    void operator()(const Args& a, hipStream_t s, Extra... e) const {
        if (fn) {
            fn(a, s, e...);
            return;
        }
        const dim3 g = geom(a, e...);
        if (g.x == 0 || g.y == 0) return;
        (void)rtc_launch(mod, g, dim3(block), &a, s);  // (a failure is what hipGetLastError() reports next)
    }
};

}  // namespace kabc
