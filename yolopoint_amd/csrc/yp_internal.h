// Internal helpers shared by the HIP translation units of libyolopoint_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdint>
#include "../../include/yolopoint_hip.h"

void yp_set_error(const char* fmt, ...);

#define YP_CHECK_HIP(expr)                                                         \
    do {                                                                           \
        hipError_t _e = (expr);                                                    \
        if (_e != hipSuccess) {                                                    \
            yp_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),    \
                         __FILE__, __LINE__);                                      \
            return YP_ERR_HIP;                                                     \
        }                                                                          \
    } while (0)

#define YP_REQUIRE(cond, ...)                                                      \
    do {                                                                           \
        if (!(cond)) {                                                             \
            yp_set_error(__VA_ARGS__);                                             \
            return YP_ERR_INVALID;                                                 \
        }                                                                          \
    } while (0)

static inline int yp_dtype_bytes(int dtype) { return dtype == YP_F32 ? 4 : ((dtype == YP_FP8 || dtype == YP_FP8_BF8) ? 1 : 2); }
static inline int yp_cdiv(int a, int b) { return (a + b - 1) / b; }

// launchers living in other translation units (used by the plan)
int yp_conv2d_launch(const YpConvDesc* d, const YpDetectDesc* det, hipStream_t stream);
