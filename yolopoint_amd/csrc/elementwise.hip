// Layout, pooling, normalisation and Detect-decode kernels (HBM-bound, gfx950).
#include "yp_internal.h"
#include <cfloat>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {

template <int DT> struct Sc;
template <> struct Sc<YP_F16> { using t = _Float16; };
template <> struct Sc<YP_BF16> { using t = __bf16; };
template <> struct Sc<YP_F32> { using t = float; };

// ------------------------------------------------------------------ NCHW fp32 -> NHWC dtype
template <int DT>
__global__ void pack_input_kernel(const float* __restrict__ x, int B, int C, int H, int W,
                                  typename Sc<DT>::t* __restrict__ out, int cs, int co, int Cpad) {
    using T = typename Sc<DT>::t;
    const size_t npix = (size_t)B * H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (size_t)gridDim.x * blockDim.x) {
        const size_t hw = (size_t)H * W;
        const size_t b = i / hw, p = i - b * hw;
        T* o = out + i * cs + co;
        for (int c = 0; c < Cpad; ++c) {
            const float v = c < C ? x[(b * C + c) * hw + p] : 0.f;
            o[c] = (T)v;
        }
    }
}

// fast path of the above for the 4-channel packed image in a 16-bit type (the training graphs' input): 4 consecutive pixels per thread,
// one 16-byte load per colour plane, 32 contiguous bytes stored (the per-channel 2-byte stores of the generic kernel ran at 1 TB/s)
template <int DT>
__global__ __launch_bounds__(256) void pack_input4_kernel(const float* __restrict__ x, int B, int C, size_t hw, typename Sc<DT>::t* __restrict__ out) {
    using T = typename Sc<DT>::t;
    const size_t n4 = (size_t)B * hw / 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const size_t pix = i * 4, b = pix / hw, p = pix - b * hw;
        float4 v[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = c < C ? *reinterpret_cast<const float4*>(x + (b * C + c) * hw + p) : float4{0.f, 0.f, 0.f, 0.f};
        T o[16];
        const float* f0 = reinterpret_cast<const float*>(&v[0]);
        const float* f1 = reinterpret_cast<const float*>(&v[1]);
        const float* f2 = reinterpret_cast<const float*>(&v[2]);
#pragma unroll
        for (int k = 0; k < 4; ++k) { o[4 * k] = (T)f0[k]; o[4 * k + 1] = (T)f1[k]; o[4 * k + 2] = (T)f2[k]; o[4 * k + 3] = (T)0.f; }
        uint4* dst = reinterpret_cast<uint4*>(out + pix * 4);
        dst[0] = *reinterpret_cast<const uint4*>(o);
        dst[1] = *reinterpret_cast<const uint4*>(o + 8);
    }
}

// ------------------------------------------------------------------ NHWC view -> NCHW fp32
template <int DT>
__global__ void unpack_nchw_kernel(const typename Sc<DT>::t* __restrict__ in, int cs, int co, int B, int C,
                                   int H, int W, float* __restrict__ out) {
    const size_t hw = (size_t)H * W;
    const size_t n = (size_t)B * C * hw;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t p = i % hw;
        const size_t bc = i / hw;
        const size_t c = bc % C, b = bc / C;
        out[i] = (float)in[(b * hw + p) * cs + co + c];
    }
}

// ------------------------------------------------------------------ SPPF: 5/9/13 max windows
// Chained MaxPool2d(5,1,2) with -inf padding equals direct 5x5 / 9x9 / 13x13 windows of x.
// One thread per (pixel, 8-channel chunk); rows of the 13x13 window are reduced horizontally
// once and folded into the three nested vertical windows.
template <int DT>
__global__ void sppf_pool_kernel(const char* __restrict__ x, int xcs, int xco, char* __restrict__ y1, int y1cs, int y1co,
                                 char* __restrict__ y2, int y2cs, int y2co, char* __restrict__ y3, int y3cs, int y3co,
                                 int B, int H, int W, int C) {
    using T = typename Sc<DT>::t;
    constexpr int CE = 16 / sizeof(T);
    const int chunks = C / CE;
    const size_t n = (size_t)B * H * W * chunks;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int ch = (int)(i % chunks);
        const size_t pix = i / chunks;
        const int w = (int)(pix % W);
        const int h = (int)((pix / W) % H);
        const int b = (int)(pix / ((size_t)W * H));
        float m5[CE], m9[CE], m13[CE];
#pragma unroll
        for (int j = 0; j < CE; ++j) m5[j] = m9[j] = m13[j] = -FLT_MAX;
        for (int dy = -6; dy <= 6; ++dy) {
            const int yy = h + dy;
            if (yy < 0 || yy >= H) continue;
            float r5[CE], r9[CE], r13[CE];
#pragma unroll
            for (int j = 0; j < CE; ++j) r5[j] = r9[j] = r13[j] = -FLT_MAX;
            for (int dx = -6; dx <= 6; ++dx) {
                const int xx = w + dx;
                if (xx < 0 || xx >= W) continue;
                const u32x4 raw = *reinterpret_cast<const u32x4*>(
                    x + (((size_t)b * H + yy) * W + xx) * xcs * sizeof(T) + (size_t)(xco + ch * CE) * sizeof(T));
                const T* e = reinterpret_cast<const T*>(&raw);
                const int ax = dx < 0 ? -dx : dx;
#pragma unroll
                for (int j = 0; j < CE; ++j) {
                    const float v = (float)e[j];
                    r13[j] = fmaxf(r13[j], v);
                    if (ax <= 4) r9[j] = fmaxf(r9[j], v);
                    if (ax <= 2) r5[j] = fmaxf(r5[j], v);
                }
            }
            const int ay = dy < 0 ? -dy : dy;
#pragma unroll
            for (int j = 0; j < CE; ++j) {
                m13[j] = fmaxf(m13[j], r13[j]);
                if (ay <= 4) m9[j] = fmaxf(m9[j], r9[j]);
                if (ay <= 2) m5[j] = fmaxf(m5[j], r5[j]);
            }
        }
        u32x4 o1, o2, o3;
        T* e1 = reinterpret_cast<T*>(&o1);
        T* e2 = reinterpret_cast<T*>(&o2);
        T* e3 = reinterpret_cast<T*>(&o3);
#pragma unroll
        for (int j = 0; j < CE; ++j) { e1[j] = (T)m5[j]; e2[j] = (T)m9[j]; e3[j] = (T)m13[j]; }
        const size_t po = pix;
        *reinterpret_cast<u32x4*>(y1 + (po * y1cs + y1co + ch * CE) * sizeof(T)) = o1;
        *reinterpret_cast<u32x4*>(y2 + (po * y2cs + y2co + ch * CE) * sizeof(T)) = o2;
        *reinterpret_cast<u32x4*>(y3 + (po * y3cs + y3co + ch * CE) * sizeof(T)) = o3;
    }
}

// LDS-resident version: one workgroup owns one image x one 16-byte channel chunk; the whole H x W
// plane of that chunk sits in LDS and the three MaxPool2d(5,1,2) are applied exactly as the
// reference chains them (separable: 5-wide row max, then 5-tall column max), ping-ponging between
// three LDS planes.  6 passes x 5 LDS reads per pixel instead of 169 global reads.
template <int DT>
__global__ __launch_bounds__(256) void sppf_pool_lds_kernel(const char* __restrict__ x, int xcs, int xco, char* __restrict__ y1,
                                                            int y1cs, int y1co, char* __restrict__ y2, int y2cs, int y2co,
                                                            char* __restrict__ y3, int y3cs, int y3co, int H, int W, int C) {
    using T = typename Sc<DT>::t;
    constexpr int CE = 16 / sizeof(T);
    extern __shared__ __attribute__((aligned(16))) char pool_smem[];
    const int HW = H * W;
    u32x4* p0 = reinterpret_cast<u32x4*>(pool_smem);
    u32x4* p1 = p0 + HW;
    u32x4* p2 = p1 + HW;
    const int chunks = C / CE;
    // workgroup -> (image, chunk) through the XCD-contiguous numbering (workgroup i runs on XCD i % 8, each XCD has its own L2): the eight
    // 16-byte chunks of a 128-byte line are consecutive logical ids, so one L2 fetches the line once and all eight workgroups hit it -- with the
    // plain numbering eight L2s each fetched every line (profiles/r05_layers_traffic.txt: 13.2 MB fetched for a 3.3 MB input)
    const int nb_ = gridDim.x, q_ = nb_ >> 3, r_ = nb_ & 7, xcd_ = blockIdx.x & 7, idx_ = blockIdx.x >> 3;
    const int bid = (xcd_ < r_ ? xcd_ * (q_ + 1) : r_ * (q_ + 1) + (xcd_ - r_) * q_) + idx_;
    const int b = bid / chunks, ch = bid % chunks;
    const size_t pix0 = (size_t)b * HW;
    // 16-bit types: the planes hold ORDER KEYS instead of the values -- a sign-magnitude 16-bit float x becomes the two's-complement integer
    // x ^ (x < 0 ? 0x7fff : 0), whose signed order is the float order (f16 and bf16 alike) -- so that a maximum of 8 channels is four packed
    // 16-bit integer maxima instead of 8 conversions + compares + selects; keys are turned back into values where they are stored.
    typedef short s16x8_t __attribute__((ext_vector_type(8)));
    auto key = [](u32x4 v) -> u32x4 {
        if constexpr (sizeof(T) == 2) {
            s16x8_t h = __builtin_bit_cast(s16x8_t, v);
            h = h ^ ((h >> 15) & (short)0x7fff);
            return __builtin_bit_cast(u32x4, h);
        } else return v;
    };
    auto vmax = [](u32x4 a, u32x4 c) -> u32x4 {
        if constexpr (sizeof(T) == 2) {
            return __builtin_bit_cast(u32x4, __builtin_elementwise_max(__builtin_bit_cast(s16x8_t, a), __builtin_bit_cast(s16x8_t, c)));
        } else {
            const T* ea = reinterpret_cast<const T*>(&a);
            const T* ec = reinterpret_cast<const T*>(&c);
            u32x4 r;
            T* er = reinterpret_cast<T*>(&r);
#pragma unroll
            for (int j = 0; j < CE; ++j) er[j] = (float)ea[j] >= (float)ec[j] ? ea[j] : ec[j];
            return r;
        }
    };
    for (int i = threadIdx.x; i < HW; i += 256)
        p0[i] = key(*reinterpret_cast<const u32x4*>(x + ((pix0 + i) * xcs + xco + ch * CE) * sizeof(T)));
    __syncthreads();
    auto rowpass = [&](const u32x4* src, u32x4* dst) {
        for (int i = threadIdx.x; i < HW; i += 256) {
            const int w = i % W;
            u32x4 m = src[i];
            for (int d = 1; d <= 2; ++d) {
                if (w - d >= 0) m = vmax(m, src[i - d]);
                if (w + d < W) m = vmax(m, src[i + d]);
            }
            dst[i] = m;
        }
        __syncthreads();
    };
    auto colpass = [&](const u32x4* src, u32x4* dst, char* y, int ycs, int yco) {
        for (int i = threadIdx.x; i < HW; i += 256) {
            const int h = i / W;
            u32x4 m = src[i];
            for (int d = 1; d <= 2; ++d) {
                if (h - d >= 0) m = vmax(m, src[i - d * W]);
                if (h + d < H) m = vmax(m, src[i + d * W]);
            }
            dst[i] = m;
            *reinterpret_cast<u32x4*>(y + ((pix0 + i) * ycs + yco + ch * CE) * sizeof(T)) = key(m);       // (the key map is its own inverse)
        }
        __syncthreads();
    };
    rowpass(p0, p1); colpass(p1, p2, y1, y1cs, y1co);     // y1 = m(x)
    rowpass(p2, p1); colpass(p1, p0, y2, y2cs, y2co);     // y2 = m(y1)
    rowpass(p0, p1); colpass(p1, p2, y3, y3cs, y3co);     // y3 = m(y2)
}

// ------------------------------------------------------------------ MaxPool2d(2, 2)  (YOLOPointv52 descriptor branch)
template <int DT>
__global__ void maxpool2_kernel(const char* __restrict__ x, int xcs, int xco, char* __restrict__ y, int ycs, int yco, int B, int Ho, int Wo, int C) {
    using T = typename Sc<DT>::t;
    constexpr int CE = 16 / sizeof(T);
    const int chunks = C / CE;
    const size_t n = (size_t)B * Ho * Wo * chunks;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int ch = (int)(i % chunks);
        const size_t pix = i / chunks;
        const int w = (int)(pix % Wo), h = (int)((pix / Wo) % Ho);
        const size_t b = pix / ((size_t)Wo * Ho);
        float m[CE];
#pragma unroll
        for (int j = 0; j < CE; ++j) m[j] = -FLT_MAX;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const u32x4 raw = *reinterpret_cast<const u32x4*>(x + (((b * 2 * Ho + 2 * h + dy) * 2 * Wo + 2 * w + dx) * xcs + xco + ch * CE) * sizeof(T));
                const T* e = reinterpret_cast<const T*>(&raw);
#pragma unroll
                for (int j = 0; j < CE; ++j) m[j] = fmaxf(m[j], (float)e[j]);
            }
        u32x4 o;
        T* eo = reinterpret_cast<T*>(&o);
#pragma unroll
        for (int j = 0; j < CE; ++j) eo[j] = (T)m[j];
        *reinterpret_cast<u32x4*>(y + (pix * ycs + yco + ch * CE) * sizeof(T)) = o;
    }
}

// ------------------------------------------------------------------ descriptor L2 norm (fp32)
// one wavefront per pixel: lanes stride over channels, butterfly-reduce the sum of squares.
__global__ void l2norm_kernel(const float* __restrict__ in, int ics, int ico, float* __restrict__ out, int ocs, int oco,
                              size_t npix, int C) {
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t p = wave; p < npix; p += nwaves) {
        const float* src = in + p * ics + ico;
        float ss = 0.f;
        for (int c = lane; c < C; c += 64) { const float v = src[c]; ss += v * v; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
        const float nrm = sqrtf(ss);
        float* dst = out + p * ocs + oco;
        for (int c = lane; c < C; c += 64) dst[c] = src[c] / nrm;   // no epsilon: models/YOLOPoint.py:219-220
    }
}

// ------------------------------------------------------------------ Detect decode
struct Anchors { float wh[16]; };
__global__ void detect_decode_kernel(const float* __restrict__ raw, int cs, int co, int B, int na, int no, int ny, int nx,
                                     float stride, Anchors anc, float* __restrict__ x_out, float* __restrict__ z_out,
                                     int rows_total, int row_offset) {
    const size_t n = (size_t)B * na * ny * nx * no;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int o = (int)(i % no);
        size_t r = i / no;
        const int xx = (int)(r % nx); r /= nx;
        const int yy = (int)(r % ny); r /= ny;
        const int a = (int)(r % na);
        const int b = (int)(r / na);
        const float v = raw[(((size_t)b * ny + yy) * nx + xx) * cs + co + a * no + o];
        x_out[i] = v;
        if (z_out != nullptr) {
            const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.4426950408889634f));   // as the fused Detect epilogue
            float z;
            if (o == 0) z = (s * 2.0f - 0.5f + (float)xx) * stride;
            else if (o == 1) z = (s * 2.0f - 0.5f + (float)yy) * stride;
            else if (o == 2) { const float t = s * 2.0f; z = t * t * anc.wh[a * 2]; }
            else if (o == 3) { const float t = s * 2.0f; z = t * t * anc.wh[a * 2 + 1]; }
            else z = s;
            const size_t row = (size_t)row_offset + ((size_t)a * ny + yy) * nx + xx;
            z_out[((size_t)b * rows_total + row) * no + o] = z;
        }
    }
}

inline int grid_for(size_t n, int block) {
    size_t g = (n + block - 1) / block;
    const size_t cap = 256 * 16;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

extern "C" int yp_pack_input(const float* x, int B, int C, int H, int W, YpView out, int dtype, void* stream) {
    YP_REQUIRE(x && out.ptr && B > 0 && C > 0 && H > 0 && W > 0, "yp_pack_input: bad arguments");
    YP_REQUIRE(out.H == H && out.W == W && out.C >= C && out.coff + out.C <= out.cstride, "yp_pack_input: view mismatch");
    hipStream_t st = (hipStream_t)stream;
    const size_t npix = (size_t)B * H * W;
    if (dtype != YP_F32 && out.C == 4 && out.cstride == 4 && out.coff == 0 && C <= 3 && ((size_t)H * W) % 4 == 0 && ((uintptr_t)x & 15) == 0 &&
        ((uintptr_t)out.ptr & 15) == 0) {
        const int g4 = grid_for(npix / 4, 256);
        if (dtype == YP_F16) pack_input4_kernel<YP_F16><<<g4, 256, 0, st>>>(x, B, C, (size_t)H * W, (_Float16*)out.ptr);
        else pack_input4_kernel<YP_BF16><<<g4, 256, 0, st>>>(x, B, C, (size_t)H * W, (__bf16*)out.ptr);
        YP_CHECK_HIP(hipGetLastError());
        return YP_OK;
    }
    const int g = grid_for(npix, 256);
    switch (dtype) {
        case YP_F16: pack_input_kernel<YP_F16><<<g, 256, 0, st>>>(x, B, C, H, W, (_Float16*)out.ptr, out.cstride, out.coff, out.C); break;
        case YP_BF16: pack_input_kernel<YP_BF16><<<g, 256, 0, st>>>(x, B, C, H, W, (__bf16*)out.ptr, out.cstride, out.coff, out.C); break;
        case YP_F32: pack_input_kernel<YP_F32><<<g, 256, 0, st>>>(x, B, C, H, W, (float*)out.ptr, out.cstride, out.coff, out.C); break;
        default: YP_REQUIRE(false, "yp_pack_input: bad dtype %d", dtype);
    }
    YP_CHECK_HIP(hipGetLastError());
    return YP_OK;
}

extern "C" int yp_unpack_nchw(YpView in, int src_dtype, int B, int C, float* out, void* stream) {
    YP_REQUIRE(in.ptr && out && B > 0 && C > 0 && C <= in.C, "yp_unpack_nchw: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const size_t n = (size_t)B * C * in.H * in.W;
    const int g = grid_for(n, 256);
    switch (src_dtype) {
        case YP_F16: unpack_nchw_kernel<YP_F16><<<g, 256, 0, st>>>((const _Float16*)in.ptr, in.cstride, in.coff, B, C, in.H, in.W, out); break;
        case YP_BF16: unpack_nchw_kernel<YP_BF16><<<g, 256, 0, st>>>((const __bf16*)in.ptr, in.cstride, in.coff, B, C, in.H, in.W, out); break;
        case YP_F32: unpack_nchw_kernel<YP_F32><<<g, 256, 0, st>>>((const float*)in.ptr, in.cstride, in.coff, B, C, in.H, in.W, out); break;
        default: YP_REQUIRE(false, "yp_unpack_nchw: bad dtype %d", src_dtype);
    }
    YP_CHECK_HIP(hipGetLastError());
    return YP_OK;
}

extern "C" int yp_sppf_pool(YpView x, YpView y1, YpView y2, YpView y3, int B, int dtype, void* stream) {
    const int ce = dtype == YP_F32 ? 4 : 8;
    YP_REQUIRE(x.ptr && y1.ptr && y2.ptr && y3.ptr && B > 0, "yp_sppf_pool: null view");
    YP_REQUIRE(x.C > 0 && x.C % ce == 0 && y1.C == x.C && y2.C == x.C && y3.C == x.C, "yp_sppf_pool: channel mismatch");
    YP_REQUIRE(y1.H == x.H && y1.W == x.W && y2.H == x.H && y3.H == x.H, "yp_sppf_pool: dims mismatch");
    YP_REQUIRE(x.cstride % ce == 0 && x.coff % ce == 0 && y1.cstride % ce == 0 && y1.coff % ce == 0 && y2.cstride % ce == 0 &&
                   y2.coff % ce == 0 && y3.cstride % ce == 0 && y3.coff % ce == 0, "yp_sppf_pool: slices must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = (size_t)3 * x.H * x.W * 16;
    if (lds <= 96 * 1024) {     // the plane fits in LDS three times: cascaded separable pooling
        const int nb = B * (x.C / ce);
#define YP_SPPF_LDS(DT)                                                                                                      \
    sppf_pool_lds_kernel<DT><<<nb, 256, lds, st>>>((const char*)x.ptr, x.cstride, x.coff, (char*)y1.ptr, y1.cstride, y1.coff, \
                                                   (char*)y2.ptr, y2.cstride, y2.coff, (char*)y3.ptr, y3.cstride, y3.coff, x.H, x.W, x.C)
        switch (dtype) {
            case YP_F16: YP_SPPF_LDS(YP_F16); break;
            case YP_BF16: YP_SPPF_LDS(YP_BF16); break;
            case YP_F32: YP_SPPF_LDS(YP_F32); break;
            default: YP_REQUIRE(false, "yp_sppf_pool: bad dtype %d", dtype);
        }
#undef YP_SPPF_LDS
        YP_CHECK_HIP(hipGetLastError());
        return YP_OK;
    }
    const size_t n = (size_t)B * x.H * x.W * (x.C / ce);
    const int g = grid_for(n, 256);
#define YP_SPPF(DT)                                                                                               \
    sppf_pool_kernel<DT><<<g, 256, 0, st>>>((const char*)x.ptr, x.cstride, x.coff, (char*)y1.ptr, y1.cstride, y1.coff, \
                                            (char*)y2.ptr, y2.cstride, y2.coff, (char*)y3.ptr, y3.cstride, y3.coff, B, x.H, x.W, x.C)
    switch (dtype) {
        case YP_F16: YP_SPPF(YP_F16); break;
        case YP_BF16: YP_SPPF(YP_BF16); break;
        case YP_F32: YP_SPPF(YP_F32); break;
        default: YP_REQUIRE(false, "yp_sppf_pool: bad dtype %d", dtype);
    }
#undef YP_SPPF
    YP_CHECK_HIP(hipGetLastError());
    return YP_OK;
}

extern "C" int yp_l2norm_f32(YpView in, YpView out, int B, int C, void* stream) {
    YP_REQUIRE(in.ptr && out.ptr && B > 0 && C > 0 && C <= in.C && C <= out.C, "yp_l2norm_f32: bad arguments");
    YP_REQUIRE(in.H == out.H && in.W == out.W, "yp_l2norm_f32: dims mismatch");
    const size_t npix = (size_t)B * in.H * in.W;
    const int g = grid_for(npix * 64, 256);
    l2norm_kernel<<<g, 256, 0, (hipStream_t)stream>>>((const float*)in.ptr, in.cstride, in.coff, (float*)out.ptr, out.cstride,
                                                     out.coff, npix, C);
    YP_CHECK_HIP(hipGetLastError());
    return YP_OK;
}

extern "C" int yp_detect_decode(YpView raw, int B, int na, int no, float stride, const float* anchors_px_host, float* x_out,
                                float* z_out, int rows_total, int row_offset, void* stream) {
    YP_REQUIRE(raw.ptr && x_out && anchors_px_host && B > 0 && na > 0 && na <= 8 && no > 5, "yp_detect_decode: bad arguments");
    YP_REQUIRE(raw.C >= na * no, "yp_detect_decode: raw view has %d channels, need %d", raw.C, na * no);
    Anchors anc{};
    for (int i = 0; i < na * 2; ++i) anc.wh[i] = anchors_px_host[i];
    const size_t n = (size_t)B * na * raw.H * raw.W * no;
    const int g = grid_for(n, 256);
    detect_decode_kernel<<<g, 256, 0, (hipStream_t)stream>>>((const float*)raw.ptr, raw.cstride, raw.coff, B, na, no, raw.H, raw.W,
                                                            stride, anc, x_out, z_out, rows_total, row_offset);
    YP_CHECK_HIP(hipGetLastError());
    return YP_OK;
}

extern "C" int yp_maxpool2(YpView x, YpView y, int B, int dtype, void* stream) {
    const int ce = dtype == YP_F32 ? 4 : 8;
    YP_REQUIRE(x.ptr && y.ptr && B > 0 && x.C > 0 && x.C == y.C && x.C % ce == 0, "yp_maxpool2: bad views");
    YP_REQUIRE(x.H == 2 * y.H && x.W == 2 * y.W && x.cstride % ce == 0 && x.coff % ce == 0 && y.cstride % ce == 0 && y.coff % ce == 0, "yp_maxpool2: dims / alignment");
    hipStream_t st = (hipStream_t)stream;
    const size_t n = (size_t)B * y.H * y.W * (x.C / ce);
    const int g = grid_for(n, 256);
    switch (dtype) {
        case YP_F16: maxpool2_kernel<YP_F16><<<g, 256, 0, st>>>((const char*)x.ptr, x.cstride, x.coff, (char*)y.ptr, y.cstride, y.coff, B, y.H, y.W, x.C); break;
        case YP_BF16: maxpool2_kernel<YP_BF16><<<g, 256, 0, st>>>((const char*)x.ptr, x.cstride, x.coff, (char*)y.ptr, y.cstride, y.coff, B, y.H, y.W, x.C); break;
        case YP_F32: maxpool2_kernel<YP_F32><<<g, 256, 0, st>>>((const char*)x.ptr, x.cstride, x.coff, (char*)y.ptr, y.cstride, y.coff, B, y.H, y.W, x.C); break;
        default: YP_REQUIRE(false, "yp_maxpool2: bad dtype %d", dtype);
    }
    YP_CHECK_HIP(hipGetLastError());
    return YP_OK;
}
