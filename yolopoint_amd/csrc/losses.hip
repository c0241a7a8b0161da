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

// The forward WITH the anchor-side gradient in the same gather pass (training: the loss is always back-propagated): the rows db[idx[i][j]]
// are the traffic of both passes (n * E rows of D floats: 4.8 GB for 24 000 anchors x 201 x 256), so
//   dda_u[i] = sum_j softmax_j(l_i) * db[idx[i][j]] - db[idx[i][0]]        (= d loss_i / d da_i * tau, the backward scales it)
// is accumulated while the logits are computed, with a running maximum as in a streaming softmax; the `logits` array receives
// w[i][j] = exp(l_ij - lse_i) - [j == 0] (what the db-side pass needs), lse[i] the row's log-sum-exp.  Replaces infonce_fwd + infonce_bwd_a (356 + 381 us -> ~400 us).
template <int VPL>
__global__ __launch_bounds__(256) void infonce_fwd_grad_kernel(const float* __restrict__ da, const float* __restrict__ db, const int* __restrict__ idx, int n, int E,
                                                               int D, float inv_tau, float* __restrict__ logits, float* __restrict__ loss,
                                                               float* __restrict__ lse, float* __restrict__ dda_u, const int* __restrict__ n_dev) {
    const int lane = threadIdx.x & 63;
    if (n_dev != nullptr) { n = n_dev[0]; db = da + (size_t)n * D; }     // (device-side count: the matched rows are [0, n) of a capacity-sized array)
    for (int i = blockIdx.x * 4 + (threadIdx.x >> 6); i < n; i += gridDim.x * 4) {      // (a capped grid walks the rows: yp_infonce_set_grid_cap)
    const Row<VPL> a = load_row<VPL>(da, i, D, lane);
    const int* row = idx + (size_t)i * E;
    float* lrow = logits + (size_t)i * E;
    float mx = -3.0e38f, s = 0.f, l0 = 0.f;          // running maximum / sum of exp(l - mx) (wave-uniform)
    float acc[VPL], b0[VPL];
#pragma unroll
    for (int k = 0; k < VPL; ++k) acc[k] = b0[k] = 0.f;
    float mine[8];                                   // logit j lives in lane j % 64, slot j / 64 (E <= 512): stored coalesced at the end
#pragma unroll
    for (int q = 0; q < 8; ++q) mine[q] = 0.f;
    for (int j0 = 0; j0 < E; j0 += 4) {
        Row<VPL> b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + u < E ? j0 + u : E - 1;
            b[u] = load_row<VPL>(db, (size_t)row[j], D, lane);
        }
        float d[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < VPL; ++k) t += a.v[k] * b[u].v[k];
            d[u] = wave_sum(t) * inv_tau;
        }
        float cmx = mx;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (j0 + u < E) cmx = fmaxf(cmx, d[u]);
        const float resc = __expf(mx - cmx);           // (first chunk: exp(-huge) = 0 on zero accumulators)
        s *= resc;
#pragma unroll
        for (int k = 0; k < VPL; ++k) acc[k] *= resc;
        mx = cmx;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + u;
            if (j < E) {
                const float e = __expf(d[u] - mx);
                s += e;
#pragma unroll
                for (int k = 0; k < VPL; ++k) acc[k] += e * b[u].v[k];
                if (j == 0) {
                    l0 = d[u];
#pragma unroll
                    for (int k = 0; k < VPL; ++k) b0[k] = b[u].v[k];
                }
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (q == (j >> 6) && lane == (j & 63)) mine[q] = d[u];
            }
        }
    }
    const float ls = mx + logf(s);
    // the row's softmax weights w[i][j] = exp(l_ij - lse_i) - [j == 0] replace the logits (what the db-side pass multiplies da[i] with)
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int j = q * 64 + lane;
        if (j < E) lrow[j] = __expf(mine[q] - ls) - (j == 0 ? 1.0f : 0.0f);
    }
    if (lane == 0) { loss[i] = ls - l0; lse[i] = ls; }
    const float inv_s = 1.0f / s;
    float* o = dda_u + (size_t)i * D + lane * VPL;
#pragma unroll
    for (int k = 0; k < VPL; ++k) o[k] = acc[k] * inv_s - b0[k];
    }
}

// ddb[k] = scale * sum over the edges (i, j) with idx[i][j] == k of w[i][j] * da[i]; edges sorted by k
template <int VPL>
__global__ __launch_bounds__(256) void infonce_bwd_b2_kernel(const float* __restrict__ da, const float* __restrict__ logits, const float* __restrict__ lse,
                                                             const int* __restrict__ order, const int* __restrict__ offsets, int n, int E, int D,
                                                             const float* __restrict__ gscale, float* __restrict__ ddb, const int* __restrict__ n_dev) {
    const int lane = threadIdx.x & 63;
    if (n_dev != nullptr) { n = n_dev[0]; ddb += (size_t)n * D; }         // (ddb = the second half of a [2n][D] gradient whose base was passed)
    for (int k = blockIdx.x * 4 + (threadIdx.x >> 6); k < n; k += gridDim.x * 4) {
    const int e0 = offsets[k], e1 = offsets[k + 1];
    const float scale = gscale[0];
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
            const int i = edge / E;
            a[u] = load_row<VPL>(da, (size_t)i, D, lane);
            we[u] = e + u < e1 ? logits[edge] : 0.f;          // (the weights yp_infonce_fwd_grad left in place of the logits)
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int q = 0; q < VPL; ++q) acc[q] += we[u] * a[u].v[q];
    }
    float* o = ddb + (size_t)k * D + lane * VPL;
#pragma unroll
    for (int q = 0; q < VPL; ++q) o[q] = acc[q] * scale;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Second generation of the two gathers (round 4).  The first one keeps 4 rows of 8-byte-per-lane loads in flight per wave and runs
// beside a backward plan under a workgroup cap (3 waves per SIMD): by Little's law that is ~6 MB in flight on the chip, ~3.9 TB/s of
// Infinity-Cache hits.  Here
//   * a row is read with 16-byte loads by HL = D/4 lanes, so ONE load instruction fetches RPL = 64/HL rows (D = 128: two rows, one per
//     half wave; D = 256: one) and the dot products reduce inside a lane group (5 instead of 6 shuffle steps for two rows at once);
//   * U = 8 load instructions are in flight per wave (16 rows at D = 128, 8 at D = 256);
//   * the E column indices of the anchor (and the logits on their way back) go through a wave-private LDS row: the index of a row is
//     an LDS broadcast read instead of a dependent global load in front of every gather;
//   * every lane group runs its own streaming softmax over the rows it fetched; the groups merge once per anchor.
// The summation order differs from the first generation (groups interleave the rows); it is fixed, so results are bit-reproducible.
// ---------------------------------------------------------------------------------------------------------------------------
template <int HL> __device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = HL / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

template <int HL>
__global__ __launch_bounds__(256) void infonce_fwd_grad2_kernel(const float* __restrict__ da, const float* __restrict__ db, const int* __restrict__ idx, int n, int E,
                                                                float inv_tau, float* __restrict__ logits, float* __restrict__ loss,
                                                                float* __restrict__ lse, float* __restrict__ dda_u, const int* __restrict__ n_dev) {
    constexpr int D = HL * 4, RPL = 64 / HL, U = 8;
    __shared__ int s_idx[4][512];
    __shared__ float s_lg[4][512];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int g = lane / HL, gl = lane % HL;
    if (n_dev != nullptr) { n = n_dev[0]; db = da + (size_t)n * D; }
    int* li = s_idx[wv];
    float* lg = s_lg[wv];
    for (int i = blockIdx.x * 4 + wv; i < n; i += gridDim.x * 4) {
        const float4 a = ld4(da + (size_t)i * D + gl * 4);
        const int* row = idx + (size_t)i * E;
        for (int j = lane; j < E; j += 64) li[j] = row[j];
        __builtin_amdgcn_wave_barrier();
        float mx = -3.0e38f, s = 0.f, l0 = 0.f;
        float4 acc = {0.f, 0.f, 0.f, 0.f}, b0 = {0.f, 0.f, 0.f, 0.f};
        for (int j0 = 0; j0 < E; j0 += RPL * U) {
            float4 b[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int j = j0 + u * RPL + g;
                const int r = li[j < E ? j : E - 1];
                b[u] = ld4(db + (size_t)r * D + gl * 4);
            }
            float d[U];
#pragma unroll
            for (int u = 0; u < U; ++u) d[u] = group_sum<HL>(a.x * b[u].x + a.y * b[u].y + a.z * b[u].z + a.w * b[u].w) * inv_tau;
            float cmx = mx;
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (j0 + u * RPL + g < E) cmx = fmaxf(cmx, d[u]);
            const float resc = __expf(mx - cmx);
            s *= resc; acc.x *= resc; acc.y *= resc; acc.z *= resc; acc.w *= resc;
            mx = cmx;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int j = j0 + u * RPL + g;
                if (j < E) {
                    const float e = __expf(d[u] - mx);
                    s += e;
                    acc.x += e * b[u].x; acc.y += e * b[u].y; acc.z += e * b[u].z; acc.w += e * b[u].w;
                    if (j == 0) { l0 = d[u]; b0 = b[u]; }
                    if (gl == 0) lg[j] = d[u];
                }
            }
        }
        // merge the lane groups' running softmax states (a group that saw no row: mx = -3e38, s = 0 -> factor 0)
#pragma unroll
        for (int o = HL; o < 64; o <<= 1) {
            const float mx2 = __shfl_xor(mx, o, 64), s2 = __shfl_xor(s, o, 64);
            float4 a2;
            a2.x = __shfl_xor(acc.x, o, 64); a2.y = __shfl_xor(acc.y, o, 64); a2.z = __shfl_xor(acc.z, o, 64); a2.w = __shfl_xor(acc.w, o, 64);
            const float m = fmaxf(mx, mx2);
            // (identical expression on both partners -- own term first would differ in the last bit between them: order by group index)
            const bool lo = (lane & o) == 0;
            const float f_lo = __expf((lo ? mx : mx2) - m), f_hi = __expf((lo ? mx2 : mx) - m);
            s = (lo ? s : s2) * f_lo + (lo ? s2 : s) * f_hi;
            acc.x = (lo ? acc.x : a2.x) * f_lo + (lo ? a2.x : acc.x) * f_hi;
            acc.y = (lo ? acc.y : a2.y) * f_lo + (lo ? a2.y : acc.y) * f_hi;
            acc.z = (lo ? acc.z : a2.z) * f_lo + (lo ? a2.z : acc.z) * f_hi;
            acc.w = (lo ? acc.w : a2.w) * f_lo + (lo ? a2.w : acc.w) * f_hi;
            mx = m;
        }
        l0 = __shfl(l0, 0, 64);
        const float ls = mx + logf(s);
        __builtin_amdgcn_wave_barrier();
        float* lrow = logits + (size_t)i * E;
        for (int j = lane; j < E; j += 64) lrow[j] = __expf(lg[j] - ls) - (j == 0 ? 1.0f : 0.0f);
        if (lane == 0) { loss[i] = ls - l0; lse[i] = ls; }
        if (g == 0) {
            const float inv_s = 1.0f / s;
            float4 o4;
            o4.x = acc.x * inv_s - b0.x; o4.y = acc.y * inv_s - b0.y; o4.z = acc.z * inv_s - b0.z; o4.w = acc.w * inv_s - b0.w;
            *reinterpret_cast<float4*>(dda_u + (size_t)i * D + gl * 4) = o4;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ddb[k] = scale * sum over the edges (i, j) with idx[i][j] == k of w[i][j] * da[i]; edges sorted by k.  A wave stages 256 edges of its
// column at a time (anchor row i = edge / E and weight, coalesced) in its LDS rows; lane group g adds the edges e0 + RPL * t + g.
template <int HL>
__global__ __launch_bounds__(256) void infonce_bwd_b3_kernel(const float* __restrict__ da, const float* __restrict__ logits, const int* __restrict__ order,
                                                             const int* __restrict__ offsets, int n, int E, const float* __restrict__ gscale,
                                                             float* __restrict__ ddb, const int* __restrict__ n_dev) {
    constexpr int D = HL * 4, RPL = 64 / HL, U = 8;
    __shared__ int s_i[4][256];
    __shared__ float s_w[4][256];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int g = lane / HL, gl = lane % HL;
    if (n_dev != nullptr) { n = n_dev[0]; ddb += (size_t)n * D; }
    int* li = s_i[wv];
    float* lw = s_w[wv];
    const float scale = gscale[0];
    for (int k = blockIdx.x * 4 + wv; k < n; k += gridDim.x * 4) {
        const int e0 = offsets[k], e1 = offsets[k + 1];
        float4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int c0 = e0; c0 < e1; c0 += 256) {
            const int cn = e1 - c0 < 256 ? e1 - c0 : 256;
            __builtin_amdgcn_wave_barrier();
            for (int q = lane; q < cn; q += 64) {
                const int edge = order[c0 + q];
                li[q] = (int)((unsigned)edge / (unsigned)E);
                lw[q] = logits[edge];
            }
            __builtin_amdgcn_wave_barrier();
            for (int q0 = 0; q0 < cn; q0 += RPL * U) {
                float4 a[U];
                float we[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int q = q0 + u * RPL + g;
                    const int qq = q < cn ? q : cn - 1;
                    a[u] = ld4(da + (size_t)li[qq] * D + gl * 4);
                    we[u] = q < cn ? lw[qq] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < U; ++u) { acc.x += we[u] * a[u].x; acc.y += we[u] * a[u].y; acc.z += we[u] * a[u].z; acc.w += we[u] * a[u].w; }
            }
        }
#pragma unroll
        for (int o = HL; o < 64; o <<= 1) {
            const float x = __shfl_xor(acc.x, o, 64), y = __shfl_xor(acc.y, o, 64), z = __shfl_xor(acc.z, o, 64), w = __shfl_xor(acc.w, o, 64);
            const bool lo = (lane & o) == 0;             // (lower group's term first on both partners: same bits)
            acc.x = lo ? acc.x + x : x + acc.x; acc.y = lo ? acc.y + y : y + acc.y; acc.z = lo ? acc.z + z : z + acc.z; acc.w = lo ? acc.w + w : w + acc.w;
        }
        if (g == 0) {
            float4 o4;
            o4.x = acc.x * scale; o4.y = acc.y * scale; o4.z = acc.z * scale; o4.w = acc.w * scale;
            *reinterpret_cast<float4*>(ddb + (size_t)k * D + gl * 4) = o4;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// The same two gathers over a 16-BIT copy of the descriptor table (rows as bf16: yp_infonce_rows16): the gathered rows are the traffic of these
// kernels (n x E rows of D floats per pass: 9.9 GB at YOLOPoint-l sizes, a fifth of the training step's HBM / fabric bytes), the table is
// what the 16-bit training modes produce from 16-bit activations anyway, and at D = 256 both generations sat at the ~8 TB/s the fabric delivers
// to random 1 KB reads -- half the bytes per row is the lever left.  A row is read with 16-byte loads of EIGHT elements by HL = D / 8 lanes (one
// load instruction fetches 64 / HL rows: 2 at D = 256, 4 at D = 128), sums and softmax state in fp32 as before; the anchor's own row (read
// once) stays fp32.  Used by engine.TrainStep when the compute dtype is bf16 (incl. fp8 mode); the fp32 kernels above remain the ones the
// reference fixtures pin (tests/test_gpu_losses_golden.py) and what fp32 / f16 graphs run.
typedef unsigned int nce_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void ld8h(const unsigned short* p, float (&v)[8]) {
    const nce_u32x4 raw = *reinterpret_cast<const nce_u32x4*>(p);
#pragma unroll
    for (int q = 0; q < 4; ++q) { v[2 * q] = __uint_as_float(raw[q] << 16); v[2 * q + 1] = __uint_as_float(raw[q] & 0xffff0000u); }
}

template <int HL>
__global__ __launch_bounds__(256) void infonce_fwd_grad2h_kernel(const float* __restrict__ da, const unsigned short* __restrict__ dbh, const int* __restrict__ idx, int n, int E,
                                                                 float inv_tau, float* __restrict__ logits, float* __restrict__ loss,
                                                                 float* __restrict__ lse, float* __restrict__ dda_u, const int* __restrict__ n_dev) {
    constexpr int D = HL * 8, RPL = 64 / HL, U = 8;
    __shared__ int s_idx[4][512];
    __shared__ float s_lg[4][512];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int g = lane / HL, gl = lane % HL;
    if (n_dev != nullptr) n = n_dev[0];
    dbh += (size_t)n * D;                              // (the 16-bit table holds both sides: anchors [0, n), matches [n, 2n))
    int* li = s_idx[wv];
    float* lg = s_lg[wv];
    for (int i = blockIdx.x * 4 + wv; i < n; i += gridDim.x * 4) {
        float a[8];
        {
            const float4 a0 = ld4(da + (size_t)i * D + gl * 8), a1 = ld4(da + (size_t)i * D + gl * 8 + 4);
            a[0] = a0.x; a[1] = a0.y; a[2] = a0.z; a[3] = a0.w; a[4] = a1.x; a[5] = a1.y; a[6] = a1.z; a[7] = a1.w;
        }
        const int* row = idx + (size_t)i * E;
        for (int j = lane; j < E; j += 64) li[j] = row[j];
        __builtin_amdgcn_wave_barrier();
        float mx = -3.0e38f, s = 0.f, l0 = 0.f;
        float acc[8], b0[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) { acc[c] = 0.f; b0[c] = 0.f; }
        for (int j0 = 0; j0 < E; j0 += RPL * U) {
            float b[U][8];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int j = j0 + u * RPL + g;
                const int r = li[j < E ? j : E - 1];
                ld8h(dbh + (size_t)r * D + gl * 8, b[u]);
            }
            float d[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float t = 0.f;
#pragma unroll
                for (int c = 0; c < 8; ++c) t += a[c] * b[u][c];
                d[u] = group_sum<HL>(t) * inv_tau;
            }
            float cmx = mx;
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (j0 + u * RPL + g < E) cmx = fmaxf(cmx, d[u]);
            const float resc = __expf(mx - cmx);
            s *= resc;
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[c] *= resc;
            mx = cmx;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int j = j0 + u * RPL + g;
                if (j < E) {
                    const float e = __expf(d[u] - mx);
                    s += e;
#pragma unroll
                    for (int c = 0; c < 8; ++c) acc[c] += e * b[u][c];
                    if (j == 0) {
                        l0 = d[u];
#pragma unroll
                        for (int c = 0; c < 8; ++c) b0[c] = b[u][c];
                    }
                    if (gl == 0) lg[j] = d[u];
                }
            }
        }
#pragma unroll
        for (int o = HL; o < 64; o <<= 1) {
            const float mx2 = __shfl_xor(mx, o, 64), s2 = __shfl_xor(s, o, 64);
            const float m = fmaxf(mx, mx2);
            const bool lo = (lane & o) == 0;
            const float f_lo = __expf((lo ? mx : mx2) - m), f_hi = __expf((lo ? mx2 : mx) - m);
            s = (lo ? s : s2) * f_lo + (lo ? s2 : s) * f_hi;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float a2 = __shfl_xor(acc[c], o, 64);
                acc[c] = (lo ? acc[c] : a2) * f_lo + (lo ? a2 : acc[c]) * f_hi;
            }
            mx = m;
        }
        l0 = __shfl(l0, 0, 64);
        const float ls = mx + logf(s);
        __builtin_amdgcn_wave_barrier();
        float* lrow = logits + (size_t)i * E;
        for (int j = lane; j < E; j += 64) lrow[j] = __expf(lg[j] - ls) - (j == 0 ? 1.0f : 0.0f);
        if (lane == 0) { loss[i] = ls - l0; lse[i] = ls; }
        if (g == 0) {
            const float inv_s = 1.0f / s;
            float4 o0, o1;
            o0.x = acc[0] * inv_s - b0[0]; o0.y = acc[1] * inv_s - b0[1]; o0.z = acc[2] * inv_s - b0[2]; o0.w = acc[3] * inv_s - b0[3];
            o1.x = acc[4] * inv_s - b0[4]; o1.y = acc[5] * inv_s - b0[5]; o1.z = acc[6] * inv_s - b0[6]; o1.w = acc[7] * inv_s - b0[7];
            *reinterpret_cast<float4*>(dda_u + (size_t)i * D + gl * 8) = o0;
            *reinterpret_cast<float4*>(dda_u + (size_t)i * D + gl * 8 + 4) = o1;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

template <int HL>
__global__ __launch_bounds__(256) void infonce_bwd_b3h_kernel(const unsigned short* __restrict__ dah, const float* __restrict__ logits, const int* __restrict__ order,
                                                              const int* __restrict__ offsets, int n, int E, const float* __restrict__ gscale,
                                                              float* __restrict__ ddb, const int* __restrict__ n_dev) {
    constexpr int D = HL * 8, RPL = 64 / HL, U = 8;
    __shared__ int s_i[4][256];
    __shared__ float s_w[4][256];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int g = lane / HL, gl = lane % HL;
    if (n_dev != nullptr) { n = n_dev[0]; ddb += (size_t)n * D; }
    int* li = s_i[wv];
    float* lw = s_w[wv];
    const float scale = gscale[0];
    for (int k = blockIdx.x * 4 + wv; k < n; k += gridDim.x * 4) {
        const int e0 = offsets[k], e1 = offsets[k + 1];
        float acc[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c] = 0.f;
        for (int c0 = e0; c0 < e1; c0 += 256) {
            const int cn = e1 - c0 < 256 ? e1 - c0 : 256;
            __builtin_amdgcn_wave_barrier();
            for (int q = lane; q < cn; q += 64) {
                const int edge = order[c0 + q];
                li[q] = (int)((unsigned)edge / (unsigned)E);
                lw[q] = logits[edge];
            }
            __builtin_amdgcn_wave_barrier();
            for (int q0 = 0; q0 < cn; q0 += RPL * U) {
                float a[U][8];
                float we[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int q = q0 + u * RPL + g;
                    const int qq = q < cn ? q : cn - 1;
                    ld8h(dah + (size_t)li[qq] * D + gl * 8, a[u]);
                    we[u] = q < cn ? lw[qq] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int c = 0; c < 8; ++c) acc[c] += we[u] * a[u][c];
            }
        }
#pragma unroll
        for (int o = HL; o < 64; o <<= 1) {
            const bool lo = (lane & o) == 0;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float x = __shfl_xor(acc[c], o, 64);
                acc[c] = lo ? acc[c] + x : x + acc[c];
            }
        }
        if (g == 0) {
            float4 o0, o1;
            o0.x = acc[0] * scale; o0.y = acc[1] * scale; o0.z = acc[2] * scale; o0.w = acc[3] * scale;
            o1.x = acc[4] * scale; o1.y = acc[5] * scale; o1.z = acc[6] * scale; o1.w = acc[7] * scale;
            *reinterpret_cast<float4*>(ddb + (size_t)k * D + gl * 8) = o0;
            *reinterpret_cast<float4*>(ddb + (size_t)k * D + gl * 8 + 4) = o1;
        }
    }
}

// fp32 rows -> bf16 (round to nearest even), 8 elements per thread
__global__ __launch_bounds__(256) void nce_rows16_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst, size_t n8) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
        const float4 a = ld4(src + i * 8), b = ld4(src + i * 8 + 4);
        const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        nce_u32x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            unsigned lo = __float_as_uint(v[2 * q]), hi = __float_as_uint(v[2 * q + 1]);
            lo += 0x7fffu + ((lo >> 16) & 1u); hi += 0x7fffu + ((hi >> 16) & 1u);
            o[q] = (lo >> 16) | (hi & 0xffff0000u);
        }
        *reinterpret_cast<nce_u32x4*>(dst + i * 8) = o;
    }
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


// ---------------------------------------------------------------------------------------------------------------------------
// Descriptor lookup of the InfoNCE loss: F.grid_sample(desc, uv, 'bilinear', align_corners=True) at P points per image
// (reference utils/loss_functions.py:553-560).  The map is read as it leaves the network (channels innermost), one wavefront per
// point: four coalesced D-float reads forward; backward four D-float atomic adds into the zeroed gradient map (PyTorch's
// grid_sampler_2d_backward walks NCHW per point and took 640 us per call at 12 000 points x 256 channels; this is ~40 us).
// ---------------------------------------------------------------------------------------------------------------------------
struct Taps {
    int off[4];      // pixel index (y*W + x) of the four taps, -1 = outside (zero padding)
    float w[4];
};
__device__ __forceinline__ Taps bilinear_taps(float u, float v, int H, int W) {
    const float ix = ((u + 1.0f) / 2.0f) * (float)(W - 1), iy = ((v + 1.0f) / 2.0f) * (float)(H - 1);
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy;
    const float wx1 = ix - fx, wx0 = (fx + 1.0f) - ix, wy1 = iy - fy, wy0 = (fy + 1.0f) - iy;
    Taps t;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int x = x0 + (k & 1), y = y0 + (k >> 1);
        const bool ok = x >= 0 && x < W && y >= 0 && y < H && fx >= -1.0f && fx <= (float)W && fy >= -1.0f && fy <= (float)H;
        t.off[k] = ok ? y * W + x : -1;
        t.w[k] = ((k & 1) ? wx1 : wx0) * ((k >> 1) ? wy1 : wy0);
    }
    return t;
}

template <int VPL>
__global__ __launch_bounds__(256) void points_sample_fwd_kernel(const float* __restrict__ map, int H, int W, int D, const float* __restrict__ uv, int P, int n,
                                                                float* __restrict__ out, const int* __restrict__ p_dev, int B) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p_dev != nullptr) { P = p_dev[0]; n = B * P; }                      // (device-side points per image: the arrays are compact, [B][P])
    if (i >= n) return;
    const Taps t = bilinear_taps(uv[2 * i], uv[2 * i + 1], H, W);
    const float* base = map + (size_t)(i / P) * H * W * D;
    float acc[VPL];
#pragma unroll
    for (int k = 0; k < VPL; ++k) acc[k] = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (t.off[q] < 0) continue;
        const Row<VPL> r = load_row<VPL>(base, (size_t)t.off[q], D, lane);
#pragma unroll
        for (int k = 0; k < VPL; ++k) acc[k] += t.w[q] * r.v[k];
    }
    float* o = out + (size_t)i * D + lane * VPL;
#pragma unroll
    for (int k = 0; k < VPL; ++k) o[k] = acc[k];
}

template <int VPL>
__global__ __launch_bounds__(256) void points_sample_bwd_kernel(const float* __restrict__ g, int H, int W, int D, const float* __restrict__ uv, int P, int n,
                                                                float* __restrict__ gmap) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const Taps t = bilinear_taps(uv[2 * i], uv[2 * i + 1], H, W);
    float* base = gmap + (size_t)(i / P) * H * W * D;
    const Row<VPL> r = load_row<VPL>(g, (size_t)i, D, lane);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (t.off[q] < 0 || t.w[q] == 0.f) continue;
        float* dst = base + (size_t)t.off[q] * D + lane * VPL;
#pragma unroll
        for (int k = 0; k < VPL; ++k) atomicAdd(dst + k, t.w[q] * r.v[k]);
    }
}

// The same backward WITHOUT atomics (and without the zero-fill of the gradient map): the (point, tap) pairs are sorted by the cell they
// touch once per batch of sample points (yp_points_sample_taps + a key sort, label-only work) and one wavefront per CELL sums its
// contributions in that order -- every cell of the map is written exactly once (zeros where nothing lands), the sums have a fixed order
// (bit-reproducible also when taps of different points overlap), and 98 M same-address-class atomics (1.25 ms at YOLOPoint-l: the chip
// retires ~250 G of them per second) become one streaming pass.
__global__ __launch_bounds__(256) void points_taps_kernel(const float* __restrict__ uv, int P, int n, int H, int W, int* __restrict__ keys,
                                                          const int* __restrict__ p_dev, int B) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (p_dev != nullptr) {                                                // (device-side points per image: entries behind B * P are skipped by the sort)
        P = p_dev[0];
        if (i >= B * P) { keys[4 * i] = keys[4 * i + 1] = keys[4 * i + 2] = keys[4 * i + 3] = 0x7fffffff; return; }
    }
    const Taps t = bilinear_taps(uv[2 * i], uv[2 * i + 1], H, W);
    const int base = (i / P) * H * W;
#pragma unroll
    for (int q = 0; q < 4; ++q) keys[4 * i + q] = (t.off[q] < 0 || t.w[q] == 0.f) ? 0x7fffffff : base + t.off[q];
}

template <int VPL>
__global__ __launch_bounds__(256) void points_sample_bwd_sorted_kernel(const float* __restrict__ g, int H, int W, int D, const float* __restrict__ uv, int ncells,
                                                                       const int* __restrict__ order, const int* __restrict__ offsets,
                                                                       const float* __restrict__ row_scale, int n_scaled, float* __restrict__ gmap,
                                                                       const int* __restrict__ n_scaled_dev) {
    const int lane = threadIdx.x & 63;
    if (n_scaled_dev != nullptr) n_scaled = n_scaled_dev[0];
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= ncells) return;
    float acc[VPL];
#pragma unroll
    for (int k = 0; k < VPL; ++k) acc[k] = 0.f;
    const int e0 = offsets[c], e1 = offsets[c + 1];
    const float rs = row_scale != nullptr ? row_scale[0] : 1.0f;
    for (int e = e0; e < e1; ++e) {
        const int ent = order[e], i = ent >> 2, q = ent & 3;
        const Taps t = bilinear_taps(uv[2 * i], uv[2 * i + 1], H, W);
        const float w = q == 0 ? t.w[0] : (q == 1 ? t.w[1] : (q == 2 ? t.w[2] : t.w[3]));
        Row<VPL> r = load_row<VPL>(g, (size_t)i, D, lane);
        if (i < n_scaled) {              // rows that still carry an unscaled gradient (the anchor half of yp_infonce_fwd_grad): scaled first,
#pragma unroll                           // rounded to fp32, exactly as a separate scaling pass would leave them
            for (int k = 0; k < VPL; ++k) r.v[k] = __fmul_rn(r.v[k], rs);
        }
#pragma unroll
        for (int k = 0; k < VPL; ++k) acc[k] += w * r.v[k];
    }
    float* o = gmap + (size_t)c * D + lane * VPL;
#pragma unroll
    for (int k = 0; k < VPL; ++k) o[k] = acc[k];
}

// ---------------------------------------------------------------------------------------------------------------------------
// Keypoint-detector loss (reference utils/loss_functions.py:600-619): BCE between softmax(semi) over the 65 cell channels and the cell
// labels, summed over channels, masked, averaged over the valid cells.  One wavefront per cell (lane = channel, lane 0 also carries the
// dustbin): softmax, the BCE with PyTorch's log clamp (-100) and its gradient w.r.t. the logits in one pass; per-workgroup partial
// sums (no same-address atomics), folded by a second tiny launch.  Replaces softmax / BCE / mask / sum kernels forward and backward.
// ---------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void detloss_kernel(const float* __restrict__ z, long zb, long zc, long zy, long zx, const float* __restrict__ y, long yb, long yc,
                                                      long yy, long yx, const float* __restrict__ mask, int B, int Hc, int Wc, float* __restrict__ dz,
                                                      float* __restrict__ partial) {
    __shared__ float sh[4][2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long cell = (long)blockIdx.x * 4 + wave, ncell = (long)B * Hc * Wc;
    float lsum = 0.f, m = 0.f;
    if (cell < ncell) {
        const int b = (int)(cell / (Hc * Wc)), rem = (int)(cell - (long)b * Hc * Wc), cy = rem / Wc, cx = rem - cy * Wc;
        const float* zp = z + b * zb + cy * zy + cx * zx;
        const float* yp = y + b * yb + cy * yy + cx * yx;
        m = mask[cell];
        const float z0 = zp[lane * zc], z1 = lane == 0 ? zp[64 * zc] : -3.0e38f;       // lane 0: channels 0 and 64
        const float t0 = yp[lane * yc], t1 = lane == 0 ? yp[64 * yc] : 0.f;
        const float mx = wave_max(fmaxf(z0, z1));
        const float e0 = expf(z0 - mx), e1 = lane == 0 ? expf(z1 - mx) : 0.f;
        const float inv = 1.0f / wave_sum(e0 + e1);
        const float p0 = e0 * inv, p1 = e1 * inv;
        auto bce = [](float p, float t, float& g) {
            const float l = -(t * fmaxf(logf(p), -100.0f) + (1.0f - t) * fmaxf(logf(1.0f - p), -100.0f));
            g = (p - t) / fmaxf((1.0f - p) * p, 1e-12f);
            return l;
        };
        float g0, g1 = 0.f;
        float l = bce(p0, t0, g0);
        if (lane == 0) l += bce(p1, t1, g1);
        lsum = wave_sum(l) * m;
        g0 *= m; g1 *= m;
        const float dot = wave_sum(p0 * g0 + (lane == 0 ? p1 * g1 : 0.f));
        float* dp = dz + b * zb + cy * zy + cx * zx;
        dp[lane * zc] = p0 * (g0 - dot);
        if (lane == 0) dp[64 * zc] = p1 * (g1 - dot);
    }
    if (lane == 0) { sh[wave][0] = lsum; sh[wave][1] = cell < ncell ? m : 0.f; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[2 * blockIdx.x] = sh[0][0] + sh[1][0] + sh[2][0] + sh[3][0];
        partial[2 * blockIdx.x + 1] = sh[0][1] + sh[1][1] + sh[2][1] + sh[3][1];
    }
}

__global__ __launch_bounds__(256) void detloss_fold_kernel(const float* __restrict__ partial, int nblk, float* __restrict__ sums) {
    __shared__ double sh[4][2];
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 256) { a += partial[2 * i]; b += partial[2 * i + 1]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        a += __longlong_as_double(((long long)__shfl_xor((int)(__double_as_longlong(a) >> 32), o, 64) << 32) | (unsigned)__shfl_xor((int)(__double_as_longlong(a) & 0xffffffffll), o, 64));
        b += __longlong_as_double(((long long)__shfl_xor((int)(__double_as_longlong(b) >> 32), o, 64) << 32) | (unsigned)__shfl_xor((int)(__double_as_longlong(b) & 0xffffffffll), o, 64));
    }
    if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6][0] = a; sh[threadIdx.x >> 6][1] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        sums[0] = (float)(sh[0][0] + sh[1][0] + sh[2][0] + sh[3][0]);
        sums[1] = (float)(sh[0][1] + sh[1][1] + sh[2][1] + sh[3][1]);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// The same loss straight from the 2-D label maps (reference train.py:212-231 feeds ComputeDetectorLoss with labels2Dto3D(labels_2D) and
// getMasks(valid_mask), utils/utils.py:184-209 / :103-116): the 65-channel target of a cell is its 64 label pixels plus the dustbin,
// divided by their sum; a cell is valid when all of its 64 mask pixels are.  Neither the [B,65,Hc,Wc] target nor a concatenated batch
// is materialised.  cell_mask_kernel (label-only, runs beside the forward) leaves the cell mask and its sum; detloss2d_kernel then writes
// the FINAL gradient -- the per-cell gradient times gscale / (sum(mask) + 1e-10) -- into the network's gradient buffer.
// ---------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cell_mask_kernel(const float* __restrict__ valid, int B, int H, int W, float* __restrict__ mask, float* __restrict__ partial) {
    __shared__ float sh[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int Hc = H >> 3, Wc = W >> 3;
    const long cell = (long)blockIdx.x * 4 + wave, ncell = (long)B * Hc * Wc;
    float m = 0.f;
    if (cell < ncell) {
        const int b = (int)(cell / (Hc * Wc)), rem = (int)(cell - (long)b * Hc * Wc), cy = rem / Wc, cx = rem - cy * Wc;
        m = valid[((size_t)b * H + cy * 8 + (lane >> 3)) * W + cx * 8 + (lane & 7)];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m *= __shfl_xor(m, o, 64);
        if (lane == 0) mask[cell] = m;
    }
    if (lane == 0) sh[wave] = m;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

__global__ __launch_bounds__(256) void cell_mask_fold_kernel(const float* __restrict__ partial, int nblk, float* __restrict__ msum) {
    __shared__ double sh[4];
    double a = yp_strided_sum256(partial, nblk);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        a += __longlong_as_double(((long long)__shfl_xor((int)(__double_as_longlong(a) >> 32), o, 64) << 32) | (unsigned)__shfl_xor((int)(__double_as_longlong(a) & 0xffffffffll), o, 64));
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) msum[0] = (float)(sh[0] + sh[1] + sh[2] + sh[3]);
}

__global__ __launch_bounds__(256) void detloss2d_kernel(const float* __restrict__ z, long zb, long zc, long zy, long zx, const float* __restrict__ labels, int H, int W,
                                                        const float* __restrict__ mask, const float* __restrict__ msum, float gscale, int B,
                                                        float* __restrict__ dz, long db, long dc, long dy, long dx, float* __restrict__ partial) {
    __shared__ float sh[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int Hc = H >> 3, Wc = W >> 3;
    const long cell = (long)blockIdx.x * 4 + wave, ncell = (long)B * Hc * Wc;
    float lsum = 0.f;
    if (cell < ncell) {
        const int b = (int)(cell / (Hc * Wc)), rem = (int)(cell - (long)b * Hc * Wc), cy = rem / Wc, cx = rem - cy * Wc;
        const float* zp = z + b * zb + cy * zy + cx * zx;
        const float m = mask[cell];
        const float gi = gscale * (1.0f / (msum[0] + 1e-10f));
        const float lab = labels[((size_t)b * H + cy * 8 + (lane >> 3)) * W + cx * 8 + (lane & 7)];
        const float s = wave_sum(lab);
        float dust = 1.0f - s;
        dust = dust < 1.0f ? 0.0f : dust;
        const float tot = s + dust;
        const float z0 = zp[lane * zc], z1 = lane == 0 ? zp[64 * zc] : -3.0e38f;       // lane 0: channels 0 and 64
        const float t0 = lab / tot, t1 = lane == 0 ? dust / tot : 0.f;
        const float mx = wave_max(fmaxf(z0, z1));
        const float e0 = expf(z0 - mx), e1 = lane == 0 ? expf(z1 - mx) : 0.f;
        const float inv = 1.0f / wave_sum(e0 + e1);
        const float p0 = e0 * inv, p1 = e1 * inv;
        auto bce = [](float p, float t, float& g) {
            const float l = -(t * fmaxf(logf(p), -100.0f) + (1.0f - t) * fmaxf(logf(1.0f - p), -100.0f));
            g = (p - t) / fmaxf((1.0f - p) * p, 1e-12f);
            return l;
        };
        float g0, g1 = 0.f;
        float l = bce(p0, t0, g0);
        if (lane == 0) l += bce(p1, t1, g1);
        lsum = wave_sum(l) * m;
        g0 *= m; g1 *= m;
        const float dot = wave_sum(p0 * g0 + (lane == 0 ? p1 * g1 : 0.f));
        float* dp = dz + b * db + cy * dy + cx * dx;
        const float u0 = p0 * (g0 - dot), u1 = p1 * (g1 - dot);          // (the unscaled gradient, rounded as yp_detloss stores it)
        dp[lane * dc] = u0 * gi;
        if (lane == 0) dp[64 * dc] = u1 * gi;
    }
    if (lane == 0) sh[wave] = lsum;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ __launch_bounds__(256) void detloss2d_fold_kernel(const float* __restrict__ partial, int nblk, const float* __restrict__ msum, float* __restrict__ loss) {
    __shared__ double sh[4];
    double a = yp_strided_sum256(partial, nblk);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        a += __longlong_as_double(((long long)__shfl_xor((int)(__double_as_longlong(a) >> 32), o, 64) << 32) | (unsigned)__shfl_xor((int)(__double_as_longlong(a) & 0xffffffffll), o, 64));
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) loss[0] = (float)(sh[0] + sh[1] + sh[2] + sh[3]) * (1.0f / (msum[0] + 1e-10f));
}

// ---------------------------------------------------------------------------------------------------------------------------
// YOLOv5 object loss of one Detect level (reference utils/loss_functions.py:90-176, `ComputeObjectLoss.__call__`; CIoU:
// utils/metrics_yolo.py:202-240), value AND gradient in three launches instead of ~330 tiny PyTorch kernels per level:
//   objloss_init    dp = 0, owner = -1
//   objloss_targets one thread per (target, cell) entry: decode the box, CIoU against the target box with forward-mode dual
//                   numbers (4 inputs); gradients are atomically added to dp (a cell can be claimed twice), the
//                   clamped IoU is kept per entry and the LAST entry of a cell becomes its owner (index_put semantics)
//   objloss_cls     one thread per (entry, class): class BCE of the claimed cells (nc > 1)
//   objloss_cells   one thread per cell: objectness BCE against tobj = IoU of the owner (0 without), gradient into channel 4
// sums[0..2] accumulate the weighted box / obj / cls terms (weights = hyp gain x level balance, folded on the host).
// ---------------------------------------------------------------------------------------------------------------------------
struct Dual {
    float v, d[4];
};
__device__ __forceinline__ Dual dconst(float c) { return Dual{c, {0.f, 0.f, 0.f, 0.f}}; }
__device__ __forceinline__ Dual operator+(const Dual& a, const Dual& b) { return Dual{a.v + b.v, {a.d[0] + b.d[0], a.d[1] + b.d[1], a.d[2] + b.d[2], a.d[3] + b.d[3]}}; }
__device__ __forceinline__ Dual operator-(const Dual& a, const Dual& b) { return Dual{a.v - b.v, {a.d[0] - b.d[0], a.d[1] - b.d[1], a.d[2] - b.d[2], a.d[3] - b.d[3]}}; }
__device__ __forceinline__ Dual operator*(const Dual& a, const Dual& b) {
    return Dual{a.v * b.v, {a.d[0] * b.v + a.v * b.d[0], a.d[1] * b.v + a.v * b.d[1], a.d[2] * b.v + a.v * b.d[2], a.d[3] * b.v + a.v * b.d[3]}};
}
__device__ __forceinline__ Dual operator*(const Dual& a, float c) { return Dual{a.v * c, {a.d[0] * c, a.d[1] * c, a.d[2] * c, a.d[3] * c}}; }
__device__ __forceinline__ Dual operator+(const Dual& a, float c) { return Dual{a.v + c, {a.d[0], a.d[1], a.d[2], a.d[3]}}; }
__device__ __forceinline__ Dual operator/(const Dual& a, const Dual& b) {
    const float r = 1.0f / b.v, q = a.v * r;
    return Dual{q, {(a.d[0] - q * b.d[0]) * r, (a.d[1] - q * b.d[1]) * r, (a.d[2] - q * b.d[2]) * r, (a.d[3] - q * b.d[3]) * r}};
}
// torch.minimum / maximum split the gradient at ties; clamp(min=0) passes it at 0
__device__ __forceinline__ Dual dmin(const Dual& a, const Dual& b) { return a.v < b.v ? a : (b.v < a.v ? b : (a + b) * 0.5f); }
__device__ __forceinline__ Dual dmax(const Dual& a, const Dual& b) { return a.v > b.v ? a : (b.v > a.v ? b : (a + b) * 0.5f); }
__device__ __forceinline__ Dual dclamp0(const Dual& a) { return a.v >= 0.f ? a : dconst(0.f); }
__device__ __forceinline__ Dual datan(const Dual& a) {
    const float g = 1.0f / (1.0f + a.v * a.v);
    return Dual{atanf(a.v), {a.d[0] * g, a.d[1] * g, a.d[2] * g, a.d[3] * g}};
}
__device__ __forceinline__ Dual dsigmoid_in(float x, int k) {       // sigmoid of input k
    const float s = 1.0f / (1.0f + expf(-x));
    Dual r = dconst(s);
    r.d[k] = s * (1.0f - s);
    return r;
}
__device__ __forceinline__ float softplus(float x) { return fmaxf(x, 0.f) + log1pf(expf(-fabsf(x))); }
// BCEWithLogits(x, t, pos_weight = pw) = pw * t * softplus(-x) + (1 - t) * softplus(x)
__device__ __forceinline__ void bce_pw(float x, float t, float pw, float& loss, float& grad) {
    const float s = 1.0f / (1.0f + expf(-x));
    loss = pw * t * softplus(-x) + (1.0f - t) * softplus(x);
    grad = (1.0f - t) * s - pw * t * (1.0f - s);
}

__device__ __forceinline__ float block_sum_256(float v, float* sh) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    const float r = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
    return r;
}

__global__ __launch_bounds__(256) void objloss_init_kernel(float* __restrict__ dp, size_t nf, int* __restrict__ owner, int cells) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i * 4 + 3 < nf) *reinterpret_cast<float4*>(dp + i * 4) = float4{0.f, 0.f, 0.f, 0.f};
    else
        for (size_t k = i * 4; k < nf; ++k) dp[k] = 0.f;
    if (i < (size_t)cells) owner[i] = -1;
}

// (entry counts: `n` from the host, or -- n_dev != nullptr -- read from the device, where yp_build_targets left it: no host sync between
// the target assignment and the loss; the grids are then sized for the capacity and surplus workgroups leave at once)
__global__ __launch_bounds__(256) void objloss_targets_kernel(const float* __restrict__ p, int no, const int* __restrict__ cell, const float* __restrict__ tbox,
                                                              const float* __restrict__ anch, int n, const int* __restrict__ n_dev, float w_box, float* __restrict__ iou_e,
                                                              int* __restrict__ owner, float* __restrict__ dp, float* __restrict__ sums) {
    __shared__ float sh[4];
    if (n_dev != nullptr) n = *n_dev;
    if (blockIdx.x * 256 >= n) return;
    w_box /= (float)n;
    const int e = blockIdx.x * 256 + threadIdx.x;
    float lb = 0.f;
    if (e < n) {
        const int c = cell[e];
        const float* row = p + (size_t)c * no;
        float* drow = dp + (size_t)c * no;
        const float eps = 1e-7f;
        const Dual px = dsigmoid_in(row[0], 0) * 2.0f + (-0.5f), py = dsigmoid_in(row[1], 1) * 2.0f + (-0.5f);
        Dual sw = dsigmoid_in(row[2], 2) * 2.0f, shh = dsigmoid_in(row[3], 3) * 2.0f;
        const Dual pw = sw * sw * anch[e * 2 + 0], ph = shh * shh * anch[e * 2 + 1];
        const float tx = tbox[e * 4 + 0], ty = tbox[e * 4 + 1], tw = tbox[e * 4 + 2], th = tbox[e * 4 + 3];
        const Dual ax1 = px - pw * 0.5f, ax2 = px + pw * 0.5f, ay1 = py - ph * 0.5f, ay2 = py + ph * 0.5f;
        const Dual bx1 = dconst(tx - tw / 2), bx2 = dconst(tx + tw / 2), by1 = dconst(ty - th / 2), by2 = dconst(ty + th / 2);
        const Dual inter = dclamp0(dmin(ax2, bx2) - dmax(ax1, bx1)) * dclamp0(dmin(ay2, by2) - dmax(ay1, by1));
        const Dual uni = pw * ph + tw * th - inter + eps;
        const Dual iou = inter / uni;
        const Dual cw = dmax(ax2, bx2) - dmin(ax1, bx1), ch = dmax(ay2, by2) - dmin(ay1, by1);
        const Dual c2 = cw * cw + ch * ch + eps;
        const Dual dx = bx1 + bx2 - ax1 - ax2, dy = by1 + by2 - ay1 - ay2;
        const Dual rho2 = (dx * dx + dy * dy) * 0.25f;
        const Dual da = dconst(atanf(tw / (th + eps))) - datan(pw / (ph + eps));
        const Dual v = da * da * 0.40528473456935109f;                     // 4 / pi^2
        const float alpha = v.v / (v.v - iou.v + (1.0f + eps));              // (no gradient through alpha)
        const Dual ciou = iou - (rho2 / c2 + v * alpha);
        lb = (1.0f - ciou.v) * w_box;
        iou_e[e] = fmaxf(ciou.v, 0.f);
        atomicMax(owner + c, e);
#pragma unroll
        for (int k = 0; k < 4; ++k) atomicAdd(drow + k, -ciou.d[k] * w_box);
    }
    lb = block_sum_256(lb, sh);
    if (threadIdx.x == 0) atomicAdd(sums + 0, lb);
}

// class BCE of the claimed cells, one thread per (entry, class): a per-entry loop over 80 classes is 80 dependent global loads
__global__ __launch_bounds__(256) void objloss_cls_kernel(const float* __restrict__ p, int no, int nc, const int* __restrict__ cell, const int* __restrict__ tcls, int n,
                                                          const int* __restrict__ n_dev, float cp, float cn, float cls_pw, float w_cls, float* __restrict__ dp,
                                                          float* __restrict__ sums) {
    __shared__ float sh[4];
    if (n_dev != nullptr) n = *n_dev;
    if ((long)blockIdx.x * 256 >= (long)n * nc) return;
    w_cls /= ((float)n * nc);
    const int i = blockIdx.x * 256 + threadIdx.x;
    float lc = 0.f;
    if (i < n * nc) {
        const int e = i / nc, k = i - e * nc;
        const size_t at = (size_t)cell[e] * no + 5 + k;
        float l, g;
        bce_pw(p[at], k == tcls[e] ? cp : cn, cls_pw, l, g);
        lc = l * w_cls;
        atomicAdd(dp + at, g * w_cls);
    }
    lc = block_sum_256(lc, sh);
    if (threadIdx.x == 0) atomicAdd(sums + 2, lc);
}

__global__ __launch_bounds__(256) void objloss_cells_kernel(const float* __restrict__ p, int no, int cells, const int* __restrict__ owner, const float* __restrict__ iou_e,
                                                            float obj_pw, float w_obj, float* __restrict__ dp, float* __restrict__ sums) {
    __shared__ float sh[4];
    const int c = blockIdx.x * 256 + threadIdx.x;
    float lo = 0.f;
    if (c < cells) {
        const int o = owner[c];
        const float t = o >= 0 ? iou_e[o] : 0.f;
        float l, g;
        bce_pw(p[(size_t)c * no + 4], t, obj_pw, l, g);
        lo = l * w_obj;
        dp[(size_t)c * no + 4] = g * w_obj;
    }
    lo = block_sum_256(lo, sh);
    if (threadIdx.x == 0) atomicAdd(sums + 1, lo);
}

// ---------------------------------------------------------------------------------------------------------------------------
// YOLOv5 target assignment on the device (reference utils/loss_functions.py:177-234 `build_targets`): every label [img, cls, x, y, w, h]
// (normalised) is tried against the na anchors of a level (kept when max(w/aw, aw/w, h/ah, ah/h) < anchor_t) and claims its own grid cell
// plus the up-to-two neighbour cells its centre is closest to (the four half-cell tests of :211-217).  The reference builds the entry
// list with boolean-mask indexing (a host synchronisation per level); here one workgroup per level evaluates the 5 * na * nt candidate
// entries, and an ordered block scan gives every live entry the position it has in the reference's list: offset-major, then anchor,
// then label -- the order matters where two entries claim one cell (the later one owns it, index_put semantics).
// Outputs per level l (arrays of capacity cap = 5 * na * nt): cell = ((img*na + a)*ny + gj)*nx + gi, tcls, tbox = (gx - gi, gy - gj, gw, gh),
// anch = the anchor, count[l].
// ---------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void build_targets_kernel(const float* __restrict__ targets, int nt, const float* __restrict__ anchors, int na,
                                                            const int* __restrict__ shapes, float anchor_t, int cap, int* __restrict__ cell_out,
                                                            int* __restrict__ cls_out, float* __restrict__ box_out, float* __restrict__ anch_out,
                                                            int* __restrict__ count) {
    __shared__ int wsum[4];
    __shared__ int running;
    const int l = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int ny = shapes[l * 2], nx = shapes[l * 2 + 1];
    const float* an = anchors + (size_t)l * na * 2;
    int* cell = cell_out + (size_t)l * cap;
    int* cls = cls_out + (size_t)l * cap;
    float* box = box_out + (size_t)l * cap * 4;
    float* anch = anch_out + (size_t)l * cap * 2;
    const int total = 5 * na * nt;
    if (t == 0) running = 0;
    __syncthreads();
    for (int base = 0; base < total; base += 256) {
        const int id = base + t;
        bool live = false;
        int o = 0, a = 0, k = 0;
        float gx = 0.f, gy = 0.f, gw = 0.f, gh = 0.f;
        if (id < total) {
            o = id / (na * nt);
            a = (id / nt) % na;
            k = id % nt;
            const float* tg = targets + (size_t)k * 6;
            gx = tg[2] * (float)nx; gy = tg[3] * (float)ny; gw = tg[4] * (float)nx; gh = tg[5] * (float)ny;
            const float rw = gw / an[a * 2], rh = gh / an[a * 2 + 1];
            live = fmaxf(fmaxf(rw, 1.0f / rw), fmaxf(rh, 1.0f / rh)) < anchor_t;
            // neighbour claims: the centre lies in the lower / upper half of its cell and is not in the border cell
            const float ix = (float)nx - gx, iy = (float)ny - gy;
            if (o == 1) live = live && (fmodf(gx, 1.0f) < 0.5f) && gx > 1.0f;
            else if (o == 2) live = live && (fmodf(gy, 1.0f) < 0.5f) && gy > 1.0f;
            else if (o == 3) live = live && (fmodf(ix, 1.0f) < 0.5f) && ix > 1.0f;
            else if (o == 4) live = live && (fmodf(iy, 1.0f) < 0.5f) && iy > 1.0f;
        }
        // ordered compaction: exclusive prefix count of the live flags over this chunk
        const unsigned long long bal = __ballot(live);
        const int before_lane = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wave] = __popcll(bal);
        __syncthreads();
        int before = running + before_lane;
        for (int w = 0; w < wave; ++w) before += wsum[w];
        if (live) {
            const float ox = o == 1 ? 0.5f : (o == 3 ? -0.5f : 0.f), oy = o == 2 ? 0.5f : (o == 4 ? -0.5f : 0.f);
            int gi = (int)(gx - ox), gj = (int)(gy - oy);              // .long(): truncation
            gi = min(max(gi, 0), nx - 1); gj = min(max(gj, 0), ny - 1);   // clamp_ acts on the view the box offsets are taken from too
            const float* tg = targets + (size_t)k * 6;
            const int b = (int)tg[0];
            cell[before] = ((b * na + a) * ny + gj) * nx + gi;
            cls[before] = (int)tg[1];
            box[before * 4 + 0] = gx - (float)gi; box[before * 4 + 1] = gy - (float)gj; box[before * 4 + 2] = gw; box[before * 4 + 3] = gh;
            anch[before * 2 + 0] = an[a * 2]; anch[before * 2 + 1] = an[a * 2 + 1];
        }
        __syncthreads();
        if (t == 0) running += wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();
    }
    if (t == 0) count[l] = running;
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

// Workgroups of the two gather kernels: one wave per row, four rows per workgroup; max_workgroups > 0 caps the grid (the workgroups then
// walk the rows), which leaves CU slots to kernels of another stream (engine.TrainStep runs this chain beside a backward plan).
// YP_NCE_GEN=1: the first-generation gathers (one row per load instruction, 4 in flight) -- kept for A/B runs and for D = 192
static int nce_generation() {
    const char* e = getenv("YP_NCE_GEN");             // (read per launch: the tests switch it inside one process)
    return e ? atoi(e) : 2;
}
static int nce_grid(int n, int cap) {
    const int g = (n + 3) / 4;
    return cap > 0 && g > cap ? cap : g;
}

extern "C" int yp_infonce_fwd_grad(const float* da, const float* db, const int* idx, int n, int E, int D, float inv_tau, float* logits, float* loss_rows,
                                   float* lse, float* dda_unscaled, const int* n_dev, int max_workgroups, void* stream) {
    YP_REQUIRE(da && (db || n_dev) && idx && logits && loss_rows && lse && dda_unscaled && n > 0 && E > 0 && E <= 512 && D > 0 && D % 64 == 0,
               "yp_infonce_fwd_grad: bad arguments (E <= 512, D %% 64 == 0)");
    const int grid = nce_grid(n, max_workgroups);
    hipStream_t st = (hipStream_t)stream;
    if (nce_generation() >= 2 && (D == 64 || D == 128 || D == 256)) {
        if (D == 64) infonce_fwd_grad2_kernel<16><<<grid, 256, 0, st>>>(da, db, idx, n, E, inv_tau, logits, loss_rows, lse, dda_unscaled, n_dev);
        else if (D == 128) infonce_fwd_grad2_kernel<32><<<grid, 256, 0, st>>>(da, db, idx, n, E, inv_tau, logits, loss_rows, lse, dda_unscaled, n_dev);
        else infonce_fwd_grad2_kernel<64><<<grid, 256, 0, st>>>(da, db, idx, n, E, inv_tau, logits, loss_rows, lse, dda_unscaled, n_dev);
        YP_CHECK_HIP(hipGetLastError());
        return YP_OK;
    }
    YP_VPL_SWITCH(D, (infonce_fwd_grad_kernel<VPL><<<grid, 256, 0, st>>>(da, db, idx, n, E, D, inv_tau, logits, loss_rows, lse, dda_unscaled, n_dev)));
    YP_CHECK_HIP(hipGetLastError());
    return YP_OK;
}

extern "C" int yp_infonce_bwd_db(const float* da, const int* order, const int* offsets, const float* logits, const float* lse, int n, int E, int D,
                                 const float* grad_scale_dev, float* ddb, const int* n_dev, int max_workgroups, void* stream) {
    YP_REQUIRE(da && order && offsets && logits && lse && grad_scale_dev && ddb && n > 0 && E > 0 && D > 0 && D % 64 == 0, "yp_infonce_bwd_db: bad arguments");
    const int grid = nce_grid(n, max_workgroups);
    hipStream_t st = (hipStream_t)stream;
    if (nce_generation() >= 2 && (D == 64 || D == 128 || D == 256)) {
        if (D == 64) infonce_bwd_b3_kernel<16><<<grid, 256, 0, st>>>(da, logits, order, offsets, n, E, grad_scale_dev, ddb, n_dev);
        else if (D == 128) infonce_bwd_b3_kernel<32><<<grid, 256, 0, st>>>(da, logits, order, offsets, n, E, grad_scale_dev, ddb, n_dev);
        else infonce_bwd_b3_kernel<64><<<grid, 256, 0, st>>>(da, logits, order, offsets, n, E, grad_scale_dev, ddb, n_dev);
        YP_CHECK_HIP(hipGetLastError());
        return YP_OK;
    }
    YP_VPL_SWITCH(D, (infonce_bwd_b2_kernel<VPL><<<grid, 256, 0, st>>>(da, logits, lse, order, offsets, n, E, D, grad_scale_dev, ddb, n_dev)));
    YP_CHECK_HIP(hipGetLastError());
    return YP_OK;
}

// 16-bit rows: rows16 = bf16 copy [2n][D] of the table [anchors | matches] (yp_infonce_rows16); arguments otherwise as yp_infonce_fwd_grad /
// yp_infonce_bwd_db.  D in {64, 128, 256}.  With n_dev the row count, and with it the start of the match half, is read on the device.
extern "C" int yp_infonce_rows16(const float* rows, size_t count, void* rows16, void* stream) {
    YP_REQUIRE(rows && rows16 && count > 0 && count % 8 == 0, "yp_infonce_rows16: bad arguments (count %% 8 == 0)");
    size_t g = (count / 8 + 255) / 256;
    if (g > 2048) g = 2048;
    nce_rows16_kernel<<<(unsigned)g, 256, 0, (hipStream_t)stream>>>(rows, (unsigned short*)rows16, count / 8);
    YP_CHECK_HIP(hipGetLastError());
    return YP_OK;
}

extern "C" int yp_infonce_fwd_grad_h(const float* da, const void* rows16, const int* idx, int n, int E, int D, float inv_tau, float* logits, float* loss_rows,
                                     float* lse, float* dda_unscaled, const int* n_dev, int max_workgroups, void* stream) {
    YP_REQUIRE(da && rows16 && idx && logits && loss_rows && lse && dda_unscaled && n > 0 && E > 0 && E <= 512 && (D == 64 || D == 128 || D == 256),
               "yp_infonce_fwd_grad_h: bad arguments (E <= 512, D in {64, 128, 256})");
    const int grid = nce_grid(n, max_workgroups);
    hipStream_t st = (hipStream_t)stream;
    const unsigned short* h = (const unsigned short*)rows16;
    if (D == 64) infonce_fwd_grad2h_kernel<8><<<grid, 256, 0, st>>>(da, h, idx, n, E, inv_tau, logits, loss_rows, lse, dda_unscaled, n_dev);
    else if (D == 128) infonce_fwd_grad2h_kernel<16><<<grid, 256, 0, st>>>(da, h, idx, n, E, inv_tau, logits, loss_rows, lse, dda_unscaled, n_dev);
    else infonce_fwd_grad2h_kernel<32><<<grid, 256, 0, st>>>(da, h, idx, n, E, inv_tau, logits, loss_rows, lse, dda_unscaled, n_dev);
    YP_CHECK_HIP(hipGetLastError());
    return YP_OK;
}

extern "C" int yp_infonce_bwd_db_h(const void* rows16, const int* order, const int* offsets, const float* logits, int n, int E, int D,
                                   const float* grad_scale_dev, float* ddb, const int* n_dev, int max_workgroups, void* stream) {
    YP_REQUIRE(rows16 && order && offsets && logits && grad_scale_dev && ddb && n > 0 && E > 0 && (D == 64 || D == 128 || D == 256), "yp_infonce_bwd_db_h: bad arguments");
    const int grid = nce_grid(n, max_workgroups);
    hipStream_t st = (hipStream_t)stream;
    const unsigned short* h = (const unsigned short*)rows16;
    if (D == 64) infonce_bwd_b3h_kernel<8><<<grid, 256, 0, st>>>(h, logits, order, offsets, n, E, grad_scale_dev, ddb, n_dev);
    else if (D == 128) infonce_bwd_b3h_kernel<16><<<grid, 256, 0, st>>>(h, logits, order, offsets, n, E, grad_scale_dev, ddb, n_dev);
    else infonce_bwd_b3h_kernel<32><<<grid, 256, 0, st>>>(h, logits, order, offsets, n, E, grad_scale_dev, ddb, n_dev);
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

extern "C" int yp_objloss_level(const float* p, int cells, int no, int nc, const int* cell, const float* tbox, const float* anch, const int* tcls, int n, float cp,
                                float cn, float cls_pw, float obj_pw, float w_box, float w_obj, float w_cls, float* iou_scratch, int* owner_scratch, float* dp,
                                float* sums, void* stream) {
    return yp_objloss_level_dev(p, cells, no, nc, cell, tbox, anch, tcls, n, nullptr, cp, cn, cls_pw, obj_pw, w_box, w_obj, w_cls, iou_scratch, owner_scratch, dp,
                                sums, stream);
}

extern "C" int yp_objloss_level_dev(const float* p, int cells, int no, int nc, const int* cell, const float* tbox, const float* anch, const int* tcls, int n,
                                    const int* n_dev, float cp, float cn, float cls_pw, float obj_pw, float w_box, float w_obj, float w_cls, float* iou_scratch,
                                    int* owner_scratch, float* dp, float* sums, void* stream) {
    YP_REQUIRE(p && dp && sums && owner_scratch && cells > 0 && no >= 5 && nc == no - 5 && n >= 0, "yp_objloss_level: bad arguments");
    YP_REQUIRE(n == 0 || (cell && tbox && anch && iou_scratch && (nc <= 1 || tcls)), "yp_objloss_level: target arrays missing");
    YP_REQUIRE(((uintptr_t)dp & 15) == 0, "yp_objloss_level: dp must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const size_t nf = (size_t)cells * no, n4 = (nf + 3) / 4;
    objloss_init_kernel<<<(unsigned)((n4 + 255) / 256), 256, 0, st>>>(dp, nf, owner_scratch, cells);
    if (n > 0) {      // (with n_dev: n is the CAPACITY of the entry arrays, the live count is read on the device)
        objloss_targets_kernel<<<(n + 255) / 256, 256, 0, st>>>(p, no, cell, tbox, anch, n, n_dev, w_box, iou_scratch, owner_scratch, dp, sums);
        if (nc > 1)
            objloss_cls_kernel<<<(unsigned)(((long)n * nc + 255) / 256), 256, 0, st>>>(p, no, nc, cell, tcls, n, n_dev, cp, cn, cls_pw, w_cls, dp, sums);
    }
    objloss_cells_kernel<<<(cells + 255) / 256, 256, 0, st>>>(p, no, cells, owner_scratch, iou_scratch, obj_pw, w_obj / cells, dp, sums);
    YP_CHECK_HIP(hipGetLastError());
    return YP_OK;
}

extern "C" int yp_points_sample_fwd(const float* map_nhwc, int B, int H, int W, int D, const float* uv, int P, float* out, const int* p_dev, void* stream) {
    YP_REQUIRE(map_nhwc && uv && out && B > 0 && H > 0 && W > 0 && P > 0 && D > 0 && D % 64 == 0, "yp_points_sample_fwd: bad arguments (D %% 64 == 0)");
    const int n = B * P, grid = (n + 3) / 4;
    YP_VPL_SWITCH(D, (points_sample_fwd_kernel<VPL><<<grid, 256, 0, (hipStream_t)stream>>>(map_nhwc, H, W, D, uv, P, n, out, p_dev, B)));
    YP_CHECK_HIP(hipGetLastError());
    return YP_OK;
}

extern "C" int yp_points_sample_bwd(const float* g, int B, int H, int W, int D, const float* uv, int P, float* gmap_nhwc, void* stream) {
    YP_REQUIRE(g && uv && gmap_nhwc && B > 0 && H > 0 && W > 0 && P > 0 && D > 0 && D % 64 == 0, "yp_points_sample_bwd: bad arguments (D %% 64 == 0)");
    const int n = B * P, grid = (n + 3) / 4;
    YP_VPL_SWITCH(D, (points_sample_bwd_kernel<VPL><<<grid, 256, 0, (hipStream_t)stream>>>(g, H, W, D, uv, P, n, gmap_nhwc)));
    YP_CHECK_HIP(hipGetLastError());
    return YP_OK;
}

extern "C" int yp_points_sample_taps(const float* uv, int B, int P, int H, int W, int* keys, const int* p_dev, void* stream) {
    YP_REQUIRE(uv && keys && B > 0 && P > 0 && H > 0 && W > 0 && (long)B * H * W < 0x7fffffffL, "yp_points_sample_taps: bad arguments");
    const int n = B * P;
    points_taps_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(uv, P, n, H, W, keys, p_dev, B);
    YP_CHECK_HIP(hipGetLastError());
    return YP_OK;
}

extern "C" int yp_points_sample_bwd_sorted(const float* g, int B, int H, int W, int D, const float* uv, int P, const int* order, const int* offsets,
                                           const float* row_scale_dev, int n_scaled_rows, float* gmap_nhwc, const int* n_scaled_dev, void* stream) {
    YP_REQUIRE(g && uv && order && offsets && gmap_nhwc && B > 0 && H > 0 && W > 0 && P > 0 && D > 0 && D % 64 == 0, "yp_points_sample_bwd_sorted: bad arguments (D %% 64 == 0)");
    const int ncells = B * H * W, grid = (ncells + 3) / 4;
    YP_VPL_SWITCH(D, (points_sample_bwd_sorted_kernel<VPL><<<grid, 256, 0, (hipStream_t)stream>>>(g, H, W, D, uv, ncells, order, offsets, row_scale_dev,
                                                                                                        row_scale_dev != nullptr ? n_scaled_rows : 0, gmap_nhwc,
                                                                                                        row_scale_dev != nullptr ? n_scaled_dev : nullptr)));
    YP_CHECK_HIP(hipGetLastError());
    return YP_OK;
}

extern "C" size_t yp_detloss_workspace_bytes(int B, int Hc, int Wc) { return ((size_t)B * Hc * Wc + 3) / 4 * 2 * sizeof(float); }

extern "C" int yp_detloss(const float* semi, const int64_t* semi_strides, const float* target, const int64_t* target_strides, const float* mask, int B, int Hc,
                          int Wc, float* dsemi, float* sums, void* workspace, size_t workspace_bytes, void* stream) {
    YP_REQUIRE(semi && semi_strides && target && target_strides && mask && dsemi && sums && workspace && B > 0 && Hc > 0 && Wc > 0, "yp_detloss: bad arguments");
    YP_REQUIRE(workspace_bytes >= yp_detloss_workspace_bytes(B, Hc, Wc), "yp_detloss: workspace too small");
    const int nblk = (int)(((size_t)B * Hc * Wc + 3) / 4);
    hipStream_t st = (hipStream_t)stream;
    detloss_kernel<<<nblk, 256, 0, st>>>(semi, semi_strides[0], semi_strides[1], semi_strides[2], semi_strides[3], target, target_strides[0], target_strides[1],
                                         target_strides[2], target_strides[3], mask, B, Hc, Wc, dsemi, (float*)workspace);
    detloss_fold_kernel<<<1, 256, 0, st>>>((const float*)workspace, nblk, sums);
    YP_CHECK_HIP(hipGetLastError());
    return YP_OK;
}


extern "C" size_t yp_cell_mask_workspace_bytes(int B, int H, int W) { return (((size_t)B * (H / 8) * (W / 8) + 3) / 4 + 4) * sizeof(float); }

extern "C" int yp_cell_mask(const float* valid2d, int B, int H, int W, float* mask, float* mask_sum, void* workspace, size_t workspace_bytes, void* stream) {
    YP_REQUIRE(valid2d && mask && mask_sum && workspace && B > 0 && H > 0 && W > 0 && H % 8 == 0 && W % 8 == 0, "yp_cell_mask: bad arguments (H, W multiples of 8)");
    YP_REQUIRE(workspace_bytes >= yp_cell_mask_workspace_bytes(B, H, W), "yp_cell_mask: workspace too small");
    const int nblk = (int)(((size_t)B * (H / 8) * (W / 8) + 3) / 4);
    hipStream_t st = (hipStream_t)stream;
    cell_mask_kernel<<<nblk, 256, 0, st>>>(valid2d, B, H, W, mask, (float*)workspace);
    cell_mask_fold_kernel<<<1, 256, 0, st>>>((const float*)workspace, nblk, mask_sum);
    YP_CHECK_HIP(hipGetLastError());
    return YP_OK;
}

extern "C" int yp_detloss2d(const float* semi, const int64_t* semi_strides, const float* labels2d, const float* mask, const float* mask_sum, float gscale, int B,
                            int H, int W, float* dsemi, const int64_t* dsemi_strides, float* loss, void* workspace, size_t workspace_bytes, void* stream) {
    YP_REQUIRE(semi && semi_strides && labels2d && mask && mask_sum && dsemi && dsemi_strides && loss && workspace && B > 0 && H > 0 && W > 0 && H % 8 == 0 && W % 8 == 0,
               "yp_detloss2d: bad arguments (H, W multiples of 8)");
    YP_REQUIRE(workspace_bytes >= yp_cell_mask_workspace_bytes(B, H, W), "yp_detloss2d: workspace too small");
    const int nblk = (int)(((size_t)B * (H / 8) * (W / 8) + 3) / 4);
    hipStream_t st = (hipStream_t)stream;
    detloss2d_kernel<<<nblk, 256, 0, st>>>(semi, semi_strides[0], semi_strides[1], semi_strides[2], semi_strides[3], labels2d, H, W, mask, mask_sum, gscale, B, dsemi,
                                           dsemi_strides[0], dsemi_strides[1], dsemi_strides[2], dsemi_strides[3], (float*)workspace);
    detloss2d_fold_kernel<<<1, 256, 0, st>>>((const float*)workspace, nblk, mask_sum, loss);
    YP_CHECK_HIP(hipGetLastError());
    return YP_OK;
}

extern "C" int yp_build_targets(const float* targets, int nt, const float* anchors, int nl, int na, const int* shapes_dev, float anchor_t, int cap, int* cell,
                                int* tcls, float* tbox, float* anch, int* count, void* stream) {
    YP_REQUIRE(anchors && shapes_dev && count && nl > 0 && nl <= 8 && na > 0 && na <= 8 && nt >= 0 && cap >= 5 * na * nt && anchor_t > 0.f, "yp_build_targets: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (nt == 0) {
        YP_CHECK_HIP(hipMemsetAsync(count, 0, sizeof(int) * nl, st));
        return YP_OK;
    }
    YP_REQUIRE(targets && cell && tcls && tbox && anch, "yp_build_targets: null output");
    build_targets_kernel<<<nl, 256, 0, st>>>(targets, nt, anchors, na, shapes_dev, anchor_t, cap, cell, tcls, tbox, anch, count);
    YP_CHECK_HIP(hipGetLastError());
    return YP_OK;
}
