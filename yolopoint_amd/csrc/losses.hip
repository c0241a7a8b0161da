// InfoNCE descriptor loss without the Gram matrix (reference utils/loss_functions.py:484-597, `descriptor_loss_sparse`):
//
//   logit[i][j] = <da[i], db[idx[i][j]]> / tau      (j = 0: the match itself, idx[i][0] = i;  j >= 1: the sampled negatives)
//   loss        = mean_i ( logsumexp_j logit[i][j] - logit[i][0] )
//
// The PyTorch formulation materialises either the gathered negatives [n, negs, D] (the reference: 1.5 GB) or the full Gram matrix
// and its dense gradient (2 x 576 MB at n = 12000).  Here one wavefront owns one anchor row: its descriptor sits in registers, the
// E = negs + 1 rows of db it needs are gathered (each a coalesced D*4-byte read that hits L2 / Infinity Cache: db is 12 MB),
// the dot products are wave reductions.  Backward: d(da) is the same gather with the softmax weights; d(db) is a gather too, over
// the edge list sorted by column (built with the sampling, before the forward passes), so no atomics and no dense [n, n] tensor.
#include "yp_internal.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

template <int VPL> struct Row { float v[VPL]; };
template <int VPL> __device__ __forceinline__ Row<VPL> load_row(const float* __restrict__ base, size_t row, int D, int lane) {
    Row<VPL> r;
    const float* p = base + row * D + lane * VPL;
    if constexpr (VPL == 4) { const float4 q = *reinterpret_cast<const float4*>(p); r.v[0] = q.x; r.v[1] = q.y; r.v[2] = q.z; r.v[3] = q.w; }
    else if constexpr (VPL == 2) { const float2 q = *reinterpret_cast<const float2*>(p); r.v[0] = q.x; r.v[1] = q.y; }
    else {
#pragma unroll
        for (int k = 0; k < VPL; ++k) r.v[k] = p[k];
    }
    return r;
}

// logits[i][j] (scaled by 1/tau) and loss_i = lse_i - logit[i][0]; one wave per anchor, 4 gathers in flight
template <int VPL>
__global__ __launch_bounds__(256) void infonce_fwd_kernel(const float* __restrict__ da, const float* __restrict__ db, const int* __restrict__ idx, int n, int E,
                                                          int D, float inv_tau, float* __restrict__ logits, float* __restrict__ loss) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const Row<VPL> a = load_row<VPL>(da, i, D, lane);
    const int* row = idx + (size_t)i * E;
    float* lrow = logits + (size_t)i * E;
    float mx = -3.0e38f, l0 = 0.f;
    float mine[8];                                   // logit j lives in lane j % 64, slot j / 64 (E <= 512)
#pragma unroll
    for (int q = 0; q < 8; ++q) mine[q] = -3.0e38f;
    for (int j0 = 0; j0 < E; j0 += 4) {
        Row<VPL> b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + u < E ? j0 + u : E - 1;
            b[u] = load_row<VPL>(db, (size_t)row[j], D, lane);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float d = 0.f;
#pragma unroll
            for (int k = 0; k < VPL; ++k) d += a.v[k] * b[u].v[k];
            d = wave_sum(d) * inv_tau;
            const int j = j0 + u;
            if (j < E) {
                mx = fmaxf(mx, d);
                if (j == 0) l0 = d;
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (q == (j >> 6) && lane == (j & 63)) mine[q] = d;
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int j = q * 64 + lane;
        if (j < E) { s += expf(mine[q] - mx); lrow[j] = mine[q]; }
    }
    s = wave_sum(s);
    if (lane == 0) loss[i] = (mx + logf(s)) - l0;
}

// w[i][j] = (softmax_j - [j == 0]) * scale  and  dda[i] = sum_j w[i][j] * db[idx[i][j]]
template <int VPL>
__global__ __launch_bounds__(256) void infonce_bwd_a_kernel(const float* __restrict__ db, const int* __restrict__ idx, const float* __restrict__ logits, int n, int E,
                                                            int D, const float* __restrict__ gscale, float* __restrict__ w, float* __restrict__ dda) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const int* row = idx + (size_t)i * E;
    const float* lrow = logits + (size_t)i * E;
    float* wrow = w + (size_t)i * E;
    float mx = -3.0e38f;
    for (int j = lane; j < E; j += 64) mx = fmaxf(mx, lrow[j]);
    mx = wave_max(mx);
    float s = 0.f;
    for (int j = lane; j < E; j += 64) s += expf(lrow[j] - mx);
    s = wave_sum(s);
    const float scale = gscale[0], inv_s = 1.0f / s;
    for (int j = lane; j < E; j += 64) wrow[j] = (expf(lrow[j] - mx) * inv_s - (j == 0 ? 1.0f : 0.0f)) * scale;    // for the d(db) pass
    float acc[VPL];
#pragma unroll
    for (int k = 0; k < VPL; ++k) acc[k] = 0.f;
    for (int j0 = 0; j0 < E; j0 += 4) {
        Row<VPL> b[4];
        float wj[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + u < E ? j0 + u : E - 1;
            b[u] = load_row<VPL>(db, (size_t)row[j], D, lane);
            wj[u] = j0 + u < E ? (expf(lrow[j] - mx) * inv_s - (j == 0 ? 1.0f : 0.0f)) * scale : 0.f;      // (recomputed: uniform over the wave)
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int k = 0; k < VPL; ++k) acc[k] += wj[u] * b[u].v[k];
    }
    float* o = dda + (size_t)i * D + lane * VPL;
#pragma unroll
    for (int k = 0; k < VPL; ++k) o[k] = acc[k];
}

// ddb[k] = sum over the edges (i, j) with idx[i][j] == k of w[i][j] * da[i]; edges sorted by k: order[offsets[k] .. offsets[k+1])
template <int VPL>
__global__ __launch_bounds__(256) void infonce_bwd_b_kernel(const float* __restrict__ da, const float* __restrict__ w, const int* __restrict__ order,
                                                            const int* __restrict__ offsets, int n, int E, int D, float* __restrict__ ddb) {
    const int lane = threadIdx.x & 63;
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= n) return;
    const int e0 = offsets[k], e1 = offsets[k + 1];
    float acc[VPL];
#pragma unroll
    for (int q = 0; q < VPL; ++q) acc[q] = 0.f;
    for (int e = e0; e < e1; e += 4) {
        Row<VPL> a[4];
        float we[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int ee = e + u < e1 ? e + u : e1 - 1;
            const int edge = order[ee];
            a[u] = load_row<VPL>(da, (size_t)(edge / E), D, lane);
            we[u] = e + u < e1 ? w[edge] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int q = 0; q < VPL; ++q) acc[q] += we[u] * a[u].v[q];
    }
    float* o = ddb + (size_t)k * D + lane * VPL;
#pragma unroll
    for (int q = 0; q < VPL; ++q) o[q] = acc[q];
}

}  // namespace

#define YP_VPL_SWITCH(D, CALL)                                  \
    switch ((D) / 64) {                                         \
        case 1: { constexpr int VPL = 1; CALL; } break;         \
        case 2: { constexpr int VPL = 2; CALL; } break;         \
        case 3: { constexpr int VPL = 3; CALL; } break;         \
        case 4: { constexpr int VPL = 4; CALL; } break;         \
        default: YP_REQUIRE(false, "descriptor width %d: multiples of 64 up to 256", D); \
    }

extern "C" int yp_infonce_fwd(const float* da, const float* db, const int* idx, int n, int E, int D, float inv_tau, float* logits, float* loss_rows,
                              void* stream) {
    YP_REQUIRE(da && db && idx && logits && loss_rows && n > 0 && E > 0 && E <= 512 && D > 0 && D % 64 == 0, "yp_infonce_fwd: bad arguments (E <= 512, D %% 64 == 0)");
    const int grid = (n + 3) / 4;
    YP_VPL_SWITCH(D, (infonce_fwd_kernel<VPL><<<grid, 256, 0, (hipStream_t)stream>>>(da, db, idx, n, E, D, inv_tau, logits, loss_rows)));
    YP_CHECK_HIP(hipGetLastError());
    return YP_OK;
}

extern "C" int yp_infonce_bwd(const float* da, const float* db, const int* idx, const int* order, const int* offsets, const float* logits, int n, int E, int D,
                              const float* grad_scale_dev, float* w_scratch, float* dda, float* ddb, void* stream) {
    YP_REQUIRE(da && db && idx && order && offsets && logits && grad_scale_dev && w_scratch && dda && ddb && n > 0 && E > 0 && D > 0 && D % 64 == 0,
               "yp_infonce_bwd: bad arguments");
    const int grid = (n + 3) / 4;
    hipStream_t st = (hipStream_t)stream;
    YP_VPL_SWITCH(D, (infonce_bwd_a_kernel<VPL><<<grid, 256, 0, st>>>(db, idx, logits, n, E, D, grad_scale_dev, w_scratch, dda)));
    YP_VPL_SWITCH(D, (infonce_bwd_b_kernel<VPL><<<grid, 256, 0, st>>>(da, w_scratch, order, offsets, n, E, D, ddb)));
    YP_CHECK_HIP(hipGetLastError());
    return YP_OK;
}
