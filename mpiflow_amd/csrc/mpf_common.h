// mpf_common.h - launcher-side helpers shared by the .hip translation units of libmpiflow_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../../include/mpiflow_hip.h"

void mpf_set_error(const char *fmt, ...);

#define MPF_REQUIRE(cond, ...)                                                                         \
    do {                                                                                               \
        if (!(cond)) {                                                                                 \
            mpf_set_error(__VA_ARGS__);                                                                \
            return MPF_ERR_BAD_ARGUMENT;                                                               \
        }                                                                                              \
    } while (0)

#define MPF_HIP(call)                                                                                  \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            mpf_set_error("%s failed: %s", #call, hipGetErrorString(e_));                              \
            return (int)e_;                                                                            \
        }                                                                                              \
    } while (0)

static inline int mpf_launch_status(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        mpf_set_error("launch of %s failed: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

static inline bool mpf_aligned16(const void *p) { return (((uintptr_t)p) & 15) == 0; }
