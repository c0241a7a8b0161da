// Internal helpers shared by the HIP translation units of libyolopoint_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdint>
#include <atomic>
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

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel instantiation, DEVICE): function attributes are per device, so a process
// that drives several GPUs must raise the limit on each of them (a process-wide flag left every device but the first at 64 KiB).
// One YpLdsAttr (a device bit set) per launcher instantiation; lock-free, idempotent.
struct YpLdsAttr { std::atomic<unsigned long long> done{0}; };
static inline hipError_t yp_set_max_lds(YpLdsAttr& st, const void* fn, int bytes) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (st.done.load(std::memory_order_acquire) & bit) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) st.done.fetch_or(bit, std::memory_order_release);
    return e;
}

static inline int yp_dtype_bytes(int dtype) { return dtype == YP_F32 ? 4 : ((dtype == YP_FP8 || dtype == YP_FP8_BF8) ? 1 : 2); }
static inline int yp_cdiv(int a, int b) { return (a + b - 1) / b; }

// element (n, k) of a packed filter image (yp_pack_weight).  mode 0: forward filter of input channels [c0, c0 + Cj); 1: the dgrad filter of
// that slice (flipped, channel-transposed); 2 / 3: the image-like (<= 4 channel) stem filter for 16-bit / fp32 plans -- 2 pairs adjacent pixels
// (k = (r * S/2 + s/2) * 8 + (s % 2) * 4 + c, what the [H, W/2, 8] view of the packed image multiplies), 3 pads to 4 channels;
// 4 + 2 py + px: the dgrad filter of a 3x3 / stride-2 / pad-1 convolution for the input pixels of parity (py, px) -- only the taps that
// reach such a pixel: din(2y + py, 2x + px) = sum over r' < 1 + py, s' < 1 + px of dout(y + r', x + s') . w[.][.][r][s] with r = 1 (py = 0) or
// r = 2 - 2 r' (py = 1), the same along x: a (1 + py) x (1 + px) stride-1 convolution over dout instead of a 3x3 one over a zero-stuffed tensor.
__device__ __forceinline__ float yp_pack_elem(const float* __restrict__ w, int Cout, int Cin, int R, int S, int c0, int Cj, int mode, int Cout_pad, int n, int k) {
    if (mode >= 4) {
        const int py = (mode - 4) >> 1, px = (mode - 4) & 1, Sp = 1 + px;
        if (n >= Cj || k >= (1 + py) * Sp * Cout_pad) return 0.f;
        const int tap = k / Cout_pad, c = k - tap * Cout_pad;
        const int rp = tap / Sp, sp = tap - rp * Sp;
        const int r = py ? 2 - 2 * rp : 1, s_ = px ? 2 - 2 * sp : 1;
        return c < Cout ? w[(((size_t)c * Cin + c0 + n) * R + r) * S + s_] : 0.f;
    }
    if (mode >= 2) {
        const int Cq = mode == 2 ? 8 : 4, Sq = mode == 2 ? S / 2 : S;
        if (n >= Cout || k >= R * Sq * Cq) return 0.f;
        const int tap = k / Cq, j = k - tap * Cq;
        const int r = tap / Sq, sq = tap - r * Sq;
        const int c = j & 3, s_ = mode == 2 ? 2 * sq + (j >> 2) : sq;
        return c < Cin ? w[(((size_t)n * Cin + c) * R + r) * S + s_] : 0.f;
    }
    const int Cq = mode == 0 ? Cj : Cout_pad, Nreal = mode == 0 ? Cout : Cj;
    if (n >= Nreal || k >= R * S * Cq) return 0.f;
    const int tap = k / Cq, c = k - tap * Cq;
    const int r = tap / S, s_ = tap - r * S;
    if (mode == 0) return w[(((size_t)n * Cin + c0 + c) * R + r) * S + s_];
    return c < Cout ? w[(((size_t)c * Cin + c0 + n) * R + (R - 1 - r)) * S + (S - 1 - s_)] : 0.f;
}

// launchers living in other translation units (used by the plan)
int yp_conv2d_launch(const YpConvDesc* d, const YpDetectDesc* det, hipStream_t stream);

// ---- 8-bit (OCP fp8) helpers shared by csrc/fp8.hip and the BatchNorm kernels that write 1-byte twins of their outputs (csrc/train.hip)
// four floats -> four saturated e4m3 (FMT 0, |max| 448) / e5m2 (FMT 1, |max| 57344) bytes
template <int FMT> __device__ __forceinline__ unsigned yp_fp8_pack4(float a, float b, float c, float d) {
    constexpr float MX = FMT == 0 ? 448.0f : 57344.0f;
    a = fminf(fmaxf(a, -MX), MX); b = fminf(fmaxf(b, -MX), MX); c = fminf(fmaxf(c, -MX), MX); d = fminf(fmaxf(d, -MX), MX);
    int v = 0;
    if constexpr (FMT == 0) {
        v = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, v, false);
        v = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, v, true);
    } else {
        v = __builtin_amdgcn_cvt_pk_bf8_f32(a, b, v, false);
        v = __builtin_amdgcn_cvt_pk_bf8_f32(c, d, v, true);
    }
    return (unsigned)v;
}
// max|x| of a 256-thread workgroup -> ONE atomic max on the float bits (values >= 0), spread over the YP_FP8_AMAX_SLOTS sub-slots of the
// tensor's maximum.  Atomics on one address retire one after the other in L2 (~0.1 us each): a wave-level atomic per 64 lanes on a single
// address cost 160 us per launch, one per workgroup over 16 addresses still 25 us for the 4096 workgroups of a BatchNorm pass.
#define YP_FP8_AMAX_SLOTS 256
__device__ __forceinline__ void yp_block_amax(float mx, float* amax) {
    __shared__ float yp_wmax[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((threadIdx.x & 63) == 0) yp_wmax[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0 && amax != nullptr) {
        const float m = fmaxf(fmaxf(yp_wmax[0], yp_wmax[1]), fmaxf(yp_wmax[2], yp_wmax[3]));
        if (m > 0.f) atomicMax(reinterpret_cast<unsigned*>(amax) + ((blockIdx.x + blockIdx.y) & (YP_FP8_AMAX_SLOTS - 1)), __float_as_uint(m));
    }
}

// Thread t's sum of p[t], p[t + 256], p[t + 512], ... (i < n) in double precision, in that order, with eight loads in flight (the single-
// workgroup folds behind the loss kernels walk 10-25 k partials: one dependent load per addition made them 13-25 us of pure latency).
__device__ __forceinline__ double yp_strided_sum256(const float* __restrict__ p, int n) {
    double a = 0.0;
    int i = threadIdx.x;
    for (; i + 7 * 256 < n; i += 8 * 256) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = p[i + j * 256];
#pragma unroll
        for (int j = 0; j < 8; ++j) a += v[j];
    }
    for (; i < n; i += 256) a += p[i];
    return a;
}
