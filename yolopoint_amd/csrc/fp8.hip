// 8-bit (OCP fp8) support of the training convolutions: quantisation of activations / gradients / filters with per-tensor scales.
//
//   stored = saturate(real / scale) in e4m3 (|max| 448: activations, filters) or e5m2 (|max| 57344: gradients); real = stored * scale.
//   A tensor's recorded maximum is YP_FP8_AMAX_SLOTS (256) floats (the workgroups spread their atomics over them; the maximum is their max).
//   Scales are device scalars that lag one step behind ("delayed scaling"): a quantisation pass uses the scale derived from the absolute
//   maximum the SAME tensor had in the previous step and records this step's maximum (atomic max on the float bits: values are >= 0);
//   yp_fp8_update_scales turns the recorded maxima into the next step's scales in one launch for all tensors.
//
// replaces: nothing in the reference (it trains in 16-bit mixed precision through accelerate, src/train.py:45-46,206); this is BASELINE.json
// configs[4] ("fp8 (CDNA4 fp8 MFMA convs)"), built next to the bf16 path so that the two can be compared step by step.
#include "yp_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

namespace {

__device__ __forceinline__ float wave_max_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// src: 16-bit (f16 / bf16) NHWC view; dst: 1-byte NHWC view with the same logical shape; 8 channels per thread
template <int DT, int FMT>
__global__ __launch_bounds__(256) void quantize_kernel(const char* __restrict__ src, int scs, int sco, unsigned char* __restrict__ dst, int dcs, int dco, size_t M, int C,
                                                       const float* __restrict__ scale, float* __restrict__ amax) {
    using T = typename std::conditional<DT == YP_F16, _Float16, __bf16>::type;
    const int chunks = C / 8;
    const size_t n = M * chunks;
    const float inv = 1.0f / scale[0];
    float mx = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const size_t r = i / chunks;
        const int ch = (int)(i - r * chunks);
        const u32x4 raw = *reinterpret_cast<const u32x4*>(src + (r * scs + sco + ch * 8) * 2);
        const T* e = reinterpret_cast<const T*>(&raw);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { v[j] = (float)e[j]; mx = fmaxf(mx, fabsf(v[j])); }
        u32x2 o;
        o[0] = yp_fp8_pack4<FMT>(v[0] * inv, v[1] * inv, v[2] * inv, v[3] * inv);
        o[1] = yp_fp8_pack4<FMT>(v[4] * inv, v[5] * inv, v[6] * inv, v[7] * inv);
        *reinterpret_cast<u32x2*>(dst + r * dcs + dco + ch * 8) = o;
    }
    yp_block_amax(mx, amax);
}

__global__ __launch_bounds__(256) void update_scales_kernel(float* __restrict__ scale, float* __restrict__ amax, const float* __restrict__ fmax, int n, float margin) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;      // one wavefront per tensor
    if (i >= n) return;
    float m = 0.f;
#pragma unroll
    for (int k = lane; k < YP_FP8_AMAX_SLOTS; k += 64) { m = fmaxf(m, amax[(size_t)i * YP_FP8_AMAX_SLOTS + k]); amax[(size_t)i * YP_FP8_AMAX_SLOTS + k] = 0.f; }
    m = wave_max_f(m);
    if (lane == 0 && m > 0.f) scale[i] = m * margin / fmax[i];
}

// fp32 master filter -> packed e4m3 [Npad + 1][Kpad] (yp_pack_weight's layouts), one table entry per packed copy; real = stored * *scale.
__global__ __launch_bounds__(256) void pack_weight_fp8_kernel(const YpPackEntry8* __restrict__ table, int n_entries) {
    const int bid = blockIdx.x;
    int e = 0;
    for (int lo = 0, hi = n_entries - 1; lo <= hi;) {
        const int mid = (lo + hi) >> 1;
        if (table[mid].blk0 <= bid) { e = mid; lo = mid + 1; } else hi = mid - 1;
    }
    const YpPackEntry8 en = table[e];
    const int Cout = (int)en.Cout, Cin = (int)en.Cin, R = (int)en.R, S = (int)en.S, c0 = (int)en.c0, Cj = (int)en.Cj, mode = (int)en.mode;
    const int Cout_pad = (int)en.Cout_pad, Kpad = (int)en.Kpad, Npad = (int)en.Npad;
    const size_t total = (size_t)(Npad + 1) * Kpad;
    const float inv = 1.0f / en.scale[0];
    const size_t base = ((size_t)(bid - (int)en.blk0) * 256 + threadIdx.x) * 4;       // 4 consecutive k of one row per thread (Kpad % 4 == 0)
    float mx = 0.f;
    if (base < total) {
        const int n = (int)(base / Kpad), k0 = (int)(base - (size_t)n * Kpad);
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = k0 + u;
            const float x = yp_pack_elem(en.w, Cout, Cin, R, S, c0, Cj, mode, Cout_pad, n, k);
            mx = fmaxf(mx, fabsf(x));
            v[u] = x * inv;
        }
        *reinterpret_cast<unsigned*>(reinterpret_cast<unsigned char*>(en.dst) + base) = yp_fp8_pack4<0>(v[0], v[1], v[2], v[3]);
    }
    yp_block_amax(mx, en.amax);
}

}  // namespace

extern "C" int yp_quantize_fp8(YpView src, YpView dst, int src_dtype, int B, int format, const float* scale, float* amax, void* stream) {
    YP_REQUIRE(src.ptr && dst.ptr && scale && B > 0 && (src_dtype == YP_F16 || src_dtype == YP_BF16) && (format == 0 || format == 1), "yp_quantize_fp8: bad arguments");
    YP_REQUIRE(src.C == dst.C && src.H == dst.H && src.W == dst.W && src.C % 8 == 0 && src.cstride % 8 == 0 && src.coff % 8 == 0 && dst.cstride % 8 == 0 && dst.coff % 8 == 0 &&
               src.ups == 0 && dst.ups == 0, "yp_quantize_fp8: views must match and be 8-channel aligned");
    const size_t M = (size_t)B * src.H * src.W;
    size_t g = (M * (src.C / 8) + 255) / 256;
    if (g > 1024) g = 1024;
    hipStream_t st = (hipStream_t)stream;
#define YP_Q(DT, F) quantize_kernel<DT, F><<<(unsigned)g, 256, 0, st>>>((const char*)src.ptr, src.cstride, src.coff, (unsigned char*)dst.ptr, dst.cstride, dst.coff, M, src.C, scale, amax)
    if (src_dtype == YP_F16) { if (format == 0) YP_Q(YP_F16, 0); else YP_Q(YP_F16, 1); }
    else { if (format == 0) YP_Q(YP_BF16, 0); else YP_Q(YP_BF16, 1); }
#undef YP_Q
    YP_CHECK_HIP(hipGetLastError());
    return YP_OK;
}

extern "C" int yp_fp8_update_scales(float* scale, float* amax, const float* fmax, int n, float margin, void* stream) {
    YP_REQUIRE(scale && amax && fmax && n > 0 && margin > 0.f, "yp_fp8_update_scales: bad arguments");
    update_scales_kernel<<<(n + 3) / 4, 256, 0, (hipStream_t)stream>>>(scale, amax, fmax, n, margin);
    YP_CHECK_HIP(hipGetLastError());
    return YP_OK;
}

extern "C" int yp_pack_weight_fp8_batch(const YpPackEntry8* table_dev, int n_entries, int total_blocks, void* stream) {
    YP_REQUIRE(table_dev && n_entries > 0 && total_blocks > 0, "yp_pack_weight_fp8_batch: bad arguments");
    pack_weight_fp8_kernel<<<total_blocks, 256, 0, (hipStream_t)stream>>>(table_dev, n_entries);
    YP_CHECK_HIP(hipGetLastError());
    return YP_OK;
}
