// host_common.hpp -- host-side helpers of the C-ABI library (not part of the ABI)
#pragma once

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "kabc_device.hpp"

namespace kabc {

void set_error(const char* fmt, ...);
const char* get_error();

#define KABC_HIP_CHECK(expr)                                                              \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess) {                                                           \
            kabc::set_error("HIP error %d (%s) at %s:%d: %s", (int)_e, hipGetErrorString(_e), \
                            __FILE__, __LINE__, #expr);                                   \
            return KABC_ERR_DEVICE;                                                       \
        }                                                                                 \
    } while (0)

// derived constants of one Factored component.  Host libm supplies the one-off
// normalisers (lgamma, erfc); everything evaluated per walker goes through the
// math contract.  Returns false for invalid parameters.
bool prepare_prior(const kabc_prior_t& pr, PriorDev& q);
bool prepare_priors(const kabc_prior_t* prior, int D, PriorSet& out);

}  // namespace kabc

struct kabc_ctx {
    int device;
    hipStream_t stream;
    bool own_stream;
};
