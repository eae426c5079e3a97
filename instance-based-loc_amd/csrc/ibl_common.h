// Shared internals of libibloc_hip.so (status codes, error channel, launch checks).
#pragma once
#include <cstdint>
#include <cstdio>

#define IBL_OK 0
#define IBL_ERR_ARG (-1)
#define IBL_ERR_HIP (-2)
#define IBL_ERR_INTERNAL (-3)
#define IBL_ERR_UNSUPPORTED (-4)

// records a thread-local message retrievable through ibl_last_error(); returns `code`
int ibl_set_error(int code, const char* fmt, ...);

#ifdef __HIPCC__
#include <hip/hip_runtime.h>
#define IBL_HIP_CHECK(expr)                                                                         \
    do {                                                                                            \
        hipError_t _e = (expr);                                                                     \
        if (_e != hipSuccess)                                                                       \
            return ibl_set_error(IBL_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                                 __FILE__, __LINE__);                                               \
    } while (0)
#define IBL_LAUNCH_CHECK() IBL_HIP_CHECK(hipGetLastError())
#endif

static inline int64_t ibl_align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }
