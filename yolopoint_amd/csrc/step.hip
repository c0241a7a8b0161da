// The small pieces between the plans of a training step (reference train.py:189-259: optimizer.zero_grad, the loss sum of :232-241,
// loss.backward's accumulation into p.grad, BatchNorm's num_batches_tracked counters): one launch each, addressed through device tables
// built once, so that a steady-state step dispatches no framework kernels at all.
#include "yp_internal.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void fill_zero_kernel(f32x4* __restrict__ p, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) p[i] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// dst_e += src_e for every entry of a table (1024 elements per workgroup; workgroups are numbered across entries, blk0 = an entry's first)
__global__ __launch_bounds__(256) void multi_add_kernel(const YpAddEntry* __restrict__ table, int n_entries) {
    const int bid = blockIdx.x;
    int e = 0;
    for (int lo = 0, hi = n_entries - 1; lo <= hi;) {       // last entry with blk0 <= bid
        const int mid = (lo + hi) >> 1;
        if (table[mid].blk0 <= bid) { e = mid; lo = mid + 1; } else hi = mid - 1;
    }
    const YpAddEntry en = table[e];
    const int64_t base = (int64_t)(bid - (int)en.blk0) * 1024;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int64_t i = base + u * 256 + threadIdx.x;
        if (i < en.n) en.dst[i] = en.mode ? en.src[i] : en.dst[i] + en.src[i];
    }
}

__global__ void counters_add_kernel(int64_t* const* __restrict__ table, int n, int64_t inc) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) *table[i] += inc;
}

__device__ __forceinline__ double wave_sum_f64(double a) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        a += __longlong_as_double(((long long)__shfl_xor((int)(__double_as_longlong(a) >> 32), o, 64) << 32) |
                                  (unsigned)__shfl_xor((int)(__double_as_longlong(a) & 0xffffffffll), o, 64));
    return a;
}

// total = (sum of the detector-group losses + lambda_desc * mean(rows)) + lambda_obj * (obj[0] + obj[1] + obj[2]), times `scale` when it
// is not 1 -- the expression of train.py:232-241 in its order; out[0..3] = total, detector, descriptor, object terms; out[4] = the InfoNCE
// row count the sum was taken over (ALWAYS written: the host's n_rows or the device-side count).
// desc_scale_out (the scalar the InfoNCE backward multiplies its gradients with) = desc_scale.
__global__ __launch_bounds__(256) void loss_combine_kernel(const float* __restrict__ det, int n_det, const float* __restrict__ rows, int n_rows,
                                                           const float* __restrict__ obj, float lambda_desc, float lambda_obj, float scale, float desc_scale,
                                                           float* __restrict__ out, float* __restrict__ desc_scale_out, const int* __restrict__ n_rows_dev,
                                                           float g_desc, double tau) {
    __shared__ double sh[4];
    if (n_rows_dev != nullptr) {           // device-side row count: desc_scale = g_desc / (tau * n), rounded as the host computes it
        n_rows = n_rows_dev[0];
        desc_scale = n_rows > 0 ? g_desc * (float)(1.0 / (tau * (double)n_rows)) : 0.f;
    }
    double a = yp_strided_sum256(rows, n_rows);
    a = wave_sum_f64(a);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float l_desc = n_rows > 0 ? (float)((sh[0] + sh[1] + sh[2] + sh[3]) / n_rows) : 0.f;
        float l_det = 0.f;
        for (int i = 0; i < n_det; ++i) l_det += det[i];
        const float l_obj = obj != nullptr ? (obj[0] + obj[1]) + obj[2] : 0.f;
        float total = (l_det + lambda_desc * l_desc) + lambda_obj * l_obj;
        if (scale != 1.0f) total *= scale;
        out[0] = total; out[1] = l_det; out[2] = l_desc; out[3] = l_obj;
        out[4] = (float)n_rows;          // the InfoNCE row count the step ran with (0: empty sampling pool, no descriptor term)
        if (desc_scale_out != nullptr) *desc_scale_out = desc_scale;
    }
}

}  // namespace

extern "C" int yp_fill_zero(void* p, size_t bytes, void* stream) {
    YP_REQUIRE(p != nullptr && bytes % 16 == 0 && ((size_t)p) % 16 == 0, "yp_fill_zero: 16-byte aligned pointer and size");
    if (bytes == 0) return YP_OK;
    const size_t n16 = bytes / 16;
    const size_t want = (n16 + 255) / 256;
    fill_zero_kernel<<<(unsigned)(want < 2048 ? want : 2048), 256, 0, (hipStream_t)stream>>>((f32x4*)p, n16);
    YP_CHECK_HIP(hipGetLastError());
    return YP_OK;
}

extern "C" int yp_multi_add(const YpAddEntry* table_dev, int n_entries, int total_blocks, void* stream) {
    YP_REQUIRE(table_dev && n_entries > 0 && total_blocks > 0, "yp_multi_add: bad arguments");
    multi_add_kernel<<<total_blocks, 256, 0, (hipStream_t)stream>>>(table_dev, n_entries);
    YP_CHECK_HIP(hipGetLastError());
    return YP_OK;
}

extern "C" int yp_counters_add(int64_t* const* table_dev, int n, int64_t inc, void* stream) {
    YP_REQUIRE(table_dev && n > 0, "yp_counters_add: bad arguments");
    counters_add_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(table_dev, n, inc);
    YP_CHECK_HIP(hipGetLastError());
    return YP_OK;
}

// (named ...5: the output is FIVE floats since round 5; a caller built against the four-float yp_loss_combine fails to link instead of
// receiving a write behind its buffer)
extern "C" int yp_loss_combine5(const float* det_losses, int n_det, const float* nce_rows, int n_rows, const float* obj_sums, float lambda_desc, float lambda_obj,
                                float scale, float desc_scale, float* out5, float* desc_scale_out, const int* n_rows_dev, float g_desc, double tau, void* stream) {
    float* const out4 = out5;
    YP_REQUIRE(out5 && n_det >= 0 && n_rows >= 0 && (n_det == 0 || det_losses) && (n_rows == 0 || nce_rows), "yp_loss_combine5: bad arguments");
    loss_combine_kernel<<<1, 256, 0, (hipStream_t)stream>>>(det_losses, n_det, nce_rows, n_rows, obj_sums, lambda_desc, lambda_obj, scale, desc_scale, out4,
                                                           desc_scale_out, n_rows_dev, g_desc, tau);
    YP_CHECK_HIP(hipGetLastError());
    return YP_OK;
}
