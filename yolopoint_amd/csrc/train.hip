// Training-path kernels around the convolutions: batch-statistics BatchNorm + SiLU forward, their
// backward, and the small backward pieces of the graph (upsample, SPPF max-pool, descriptor L2 norm,
// Detect permute).  All HBM-bound row/column passes over NHWC views.
//
// Reference semantics: nn.BatchNorm2d(eps=1e-3, momentum=0.03) in train mode (models/common.py:18-29):
// normalise with the biased batch variance, update running stats with the unbiased one; nn.SiLU;
// Bottleneck residual add (common.py:88-89).  Backward = what autograd derives for those ops
// (train.py:245 `accelerator.backward(loss)`).
#include "yp_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {

template <int DT> struct Sc;
template <> struct Sc<YP_F16> { using t = _Float16; };
template <> struct Sc<YP_BF16> { using t = __bf16; };
template <> struct Sc<YP_F32> { using t = float; };

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
inline int grid_for(size_t n, int block, size_t cap = 256 * 16) {
    size_t g = (n + block - 1) / block;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

constexpr int BN_ROWS = 512;     // rows reduced by one workgroup
template <int DT> __device__ __forceinline__ constexpr size_t esize() { return DT == YP_F32 ? 4 : 2; }

// load 8 consecutive channels of a row as floats (CE8: two 16-byte loads for f32)
template <int DT>
__device__ __forceinline__ void load8(const char* base, size_t elem_off, float (&v)[8]) {
    using T = typename Sc<DT>::t;
    if constexpr (DT == YP_F32) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(base + elem_off * 4);
        const f32x4 b = *reinterpret_cast<const f32x4*>(base + elem_off * 4 + 16);
        v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
    } else {
        const u32x4 raw = *reinterpret_cast<const u32x4*>(base + elem_off * 2);
        const T* e = reinterpret_cast<const T*>(&raw);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (float)e[j];
    }
}
template <int DT>
__device__ __forceinline__ void store8(char* base, size_t elem_off, const float (&v)[8]) {
    using T = typename Sc<DT>::t;
    if constexpr (DT == YP_F32) {
        *reinterpret_cast<f32x4*>(base + elem_off * 4) = f32x4{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(base + elem_off * 4 + 16) = f32x4{v[4], v[5], v[6], v[7]};
    } else {
        u32x4 pk;
        T* e = reinterpret_cast<T*>(&pk);
#pragma unroll
        for (int j = 0; j < 8; ++j) e[j] = (T)v[j];
        *reinterpret_cast<u32x4*>(base + elem_off * 2) = pk;
    }
}

__device__ __forceinline__ float sigmoidf_(float z) { return 1.0f / (1.0f + expf(-z)); }

// ---------------------------------------------------------------------------------------------
// Column reductions over the M = B*H*W rows of an NHWC view.  One workgroup owns BN_ROWS rows and all
// channels: thread = (row lane, 8-channel chunk); partial sums go to part[blk][2][C] (fp32), a second
// kernel folds the partials in double precision.
//   MODE 0: (sum x, sum x^2)                       -> batch statistics
//   MODE 1: (sum dz, sum dz * xhat), dz = dy*act'  -> BatchNorm / SiLU backward
// ---------------------------------------------------------------------------------------------
template <int DT, int MODE>
__global__ __launch_bounds__(256) void col_reduce_kernel(const char* __restrict__ raw, int rcs, int rco, const char* __restrict__ dy,
                                                         int dcs, int dco, size_t Mg, int nbg, int C, const float* __restrict__ mean,
                                                         const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, int act, float* __restrict__ part) {
    // statistics groups (Mg rows each, nbg workgroups each): a workgroup never straddles two groups; MODE 1 reads the group's mean / invstd
    __shared__ float red[2][256][8];
    const int grp = blockIdx.x / nbg, gblk = blockIdx.x - grp * nbg;
    const int chunks = C / 8;
    const int rlanes = 256 / chunks;            // row lanes per workgroup (chunks <= 256)
    const int t = threadIdx.x;
    const int ch = t % chunks, rl = t / chunks;
    float s0[8], s1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s0[j] = s1[j] = 0.f;
    if (rl < rlanes) {
        float mu[8], is[8], ga[8], be[8];
        if constexpr (MODE == 1) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                mu[j] = mean[grp * C + ch * 8 + j]; is[j] = invstd[grp * C + ch * 8 + j];
                ga[j] = gamma[ch * 8 + j]; be[j] = beta[ch * 8 + j];
            }
        }
        const size_t r0 = (size_t)grp * Mg + (size_t)gblk * BN_ROWS;
        const size_t r1 = ((size_t)gblk + 1) * BN_ROWS < Mg ? r0 + BN_ROWS : ((size_t)grp + 1) * Mg;
        for (size_t r = r0 + rl; r < r1; r += rlanes) {
            float x[8];
            load8<DT>(raw, r * rcs + rco + ch * 8, x);
            if constexpr (MODE == 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { s0[j] += x[j]; s1[j] += x[j] * x[j]; }
            } else {
                float g[8];
                load8<DT>(dy, r * dcs + dco + ch * 8, g);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float xh = (x[j] - mu[j]) * is[j];
                    float dz = g[j];
                    if (act == YP_ACT_SILU) {
                        const float z = xh * ga[j] + be[j];
                        const float sg = sigmoidf_(z);
                        dz *= sg * (1.0f + z * (1.0f - sg));
                    }
                    s0[j] += dz; s1[j] += dz * xh;
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[0][t][j] = s0[j]; red[1][t][j] = s1[j]; }
    __syncthreads();
    if (rl == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float a0 = 0.f, a1 = 0.f;
            for (int q = 0; q < rlanes; ++q) { a0 += red[0][q * chunks + ch][j]; a1 += red[1][q * chunks + ch][j]; }
            part[((size_t)0 * C + ch * 8 + j) * gridDim.x + blockIdx.x] = a0;      // part[stat][channel][row block]: a channel's partials lie together
            part[((size_t)1 * C + ch * 8 + j) * gridDim.x + blockIdx.x] = a1;
        }
    }
}

// fold partials -> mean / invstd (+ running statistics), or -> dgamma / dbeta.  One workgroup per channel: threads
// stride over the workgroup partials, then a butterfly reduction in double precision.
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int lo = __shfl_xor((int)(__double_as_longlong(v) & 0xffffffffll), o, 64);
        const int hi = __shfl_xor((int)(__double_as_longlong(v) >> 32), o, 64);
        v += __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
    }
    return v;
}
// (256 threads per channel: with one wavefront the 2048 partial rows of a large layer were 32 dependent round trips)
// part[stat][channel][R row blocks] (R = all groups' rows); this group's rows are [row0, row0 + nblk): contiguous per channel
__device__ __forceinline__ void fold_partials(const float* __restrict__ part, int R, int row0, int nblk, int C, int c, double& s, double& ss) {
    __shared__ double fold_sh[8];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    double a = 0.0, b = 0.0;
    const float* pa = part + (size_t)c * R + row0;
    const float* pb = part + ((size_t)C + c) * R + row0;
#pragma unroll 4
    for (int q = t; q < nblk; q += 256) { a += pa[q]; b += pb[q]; }
    a = wave_sum_d(a);
    b = wave_sum_d(b);
    if (lane == 0) { fold_sh[2 * w] = a; fold_sh[2 * w + 1] = b; }
    __syncthreads();
    s = fold_sh[0] + fold_sh[2] + fold_sh[4] + fold_sh[6];
    ss = fold_sh[1] + fold_sh[3] + fold_sh[5] + fold_sh[7];
}
// WAVE variants (few partial rows, <= 256): one wavefront per channel, four channels per workgroup, no LDS / barrier -- these kernels
// are pure launch latency (237 of them per training step), so the lighter they are the better.
template <bool WAVE>
__device__ __forceinline__ bool fold_dispatch(const float* __restrict__ part, int R, int row0, int nblk, int C, int& c, double& s, double& ss) {
    if constexpr (WAVE) {
        c = blockIdx.x * 4 + (threadIdx.x >> 6);
        if (c >= C) return false;
        const int lane = threadIdx.x & 63;
        double a = 0.0, b = 0.0;
        const float* pa = part + (size_t)c * R + row0;
        const float* pb = part + ((size_t)C + c) * R + row0;
        for (int q = lane; q < nblk; q += 64) { a += pa[q]; b += pb[q]; }
        s = wave_sum_d(a);
        ss = wave_sum_d(b);
        return lane == 0;
    } else {
        c = blockIdx.x;
        fold_partials(part, R, row0, nblk, C, c, s, ss);
        return threadIdx.x == 0;
    }
}
template <bool WAVE>
__global__ __launch_bounds__(256) void bn_stats_finalize_kernel(const float* __restrict__ part, int nblk, int C, double M, float eps, float momentum,
                                                               float* __restrict__ mean, float* __restrict__ invstd, float* running_mean,
                                                               float* running_var, int groups) {
    // `groups` statistics groups of nblk partial rows / M pixels each: mean / invstd [groups][C]; the running statistics take one momentum
    // update per group, in group order (= consecutive forward passes of the module over the groups' samples)
    for (int g = 0; g < groups; ++g) {
        int c;
        double s, ss;
        if (fold_dispatch<WAVE>(part, nblk * groups, g * nblk, nblk, C, c, s, ss)) {
            const double mu = s / M;
            double var = ss / M - mu * mu;
            if (var < 0.0) var = 0.0;
            mean[g * C + c] = (float)mu;
            invstd[g * C + c] = (float)(1.0 / sqrt(var + (double)eps));
            if (running_mean != nullptr) {
                const double unbiased = M > 1.0 ? var * M / (M - 1.0) : var;
                running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mu);
                running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
            }
        }
        if constexpr (!WAVE) __syncthreads();
    }
}
// o0 / o1 get the two column sums; when p0 / p1 are given they receive (or accumulate) them as well
template <bool WAVE>
__global__ __launch_bounds__(256) void pair_finalize_kernel(const float* __restrict__ part, int nblk, int C, float* __restrict__ o0, float* __restrict__ o1,
                                                           float* p0, float* p1, int accumulate, int groups) {
    // per-group sums -> o0 / o1 [groups][C]; p0 / p1 get the sums over all groups
    double ts = 0.0, tss = 0.0;
    int c = 0;
    bool lead = false;
    for (int g = 0; g < groups; ++g) {
        double s, ss;
        if (fold_dispatch<WAVE>(part, nblk * groups, g * nblk, nblk, C, c, s, ss)) {
            lead = true;
            if (o0) o0[g * C + c] = (float)s;
            if (o1) o1[g * C + c] = (float)ss;
            ts += s; tss += ss;
        }
        if constexpr (!WAVE) __syncthreads();
    }
    if (!lead) return;
    if (p0) p0[c] = (accumulate ? p0[c] : 0.f) + (float)ts;
    if (p1) p1[c] = (accumulate ? p1[c] : 0.f) + (float)tss;
}

// ---------------------------------------------------------------------------------------------
// y = act(gamma*(x-mean)*invstd + beta) [+ res]
// ---------------------------------------------------------------------------------------------
template <int DT>
__global__ void bn_apply_kernel(const char* __restrict__ raw, int rcs, int rco, char* __restrict__ out, int ocs, int oco,
                                const char* __restrict__ res, int scs, int sco, size_t M, size_t Mg, int C, const float* __restrict__ mean,
                                const float* __restrict__ invstd, const float* __restrict__ gamma, const float* __restrict__ beta,
                                int act) {
    const int chunks = C / 8;
    const size_t n = M * chunks;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int ch = (int)(i % chunks);
        const size_t r = i / chunks;
        const int gc = (int)(r / Mg) * C;
        float x[8], y[8];
        load8<DT>(raw, r * rcs + rco + ch * 8, x);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = ch * 8 + j;
            float z = (x[j] - mean[gc + c]) * invstd[gc + c] * gamma[c] + beta[c];
            if (act == YP_ACT_SILU) z = z * sigmoidf_(z);
            y[j] = z;
        }
        if (res != nullptr) {
            float rr[8];
            load8<DT>(res, r * scs + sco + ch * 8, rr);
#pragma unroll
            for (int j = 0; j < 8; ++j) y[j] += rr[j];
        }
        store8<DT>(out, r * ocs + oco + ch * 8, y);
    }
}

// dx = gamma*invstd*(dz - dbeta/M - xhat*dgamma/M), dz = dy*act'(z)
template <int DT>
__global__ void bn_bwd_apply_kernel(const char* __restrict__ raw, int rcs, int rco, const char* __restrict__ dy, int dcs, int dco,
                                    char* __restrict__ dx, int xcs, int xco, size_t M, size_t Mg, int C, const float* __restrict__ mean,
                                    const float* __restrict__ invstd, const float* __restrict__ gamma, const float* __restrict__ beta,
                                    int act, const float* __restrict__ dgamma, const float* __restrict__ dbeta) {
    const int chunks = C / 8;
    const size_t n = M * chunks;
    const float invM = 1.0f / (float)Mg;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int ch = (int)(i % chunks);
        const size_t r = i / chunks;
        float x[8], g[8], o[8];
        const int gc = (int)(r / Mg) * C;
        load8<DT>(raw, r * rcs + rco + ch * 8, x);
        load8<DT>(dy, r * dcs + dco + ch * 8, g);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = ch * 8 + j;
            const float xh = (x[j] - mean[gc + c]) * invstd[gc + c];
            float dz = g[j];
            if (act == YP_ACT_SILU) {
                const float z = xh * gamma[c] + beta[c];
                const float sg = sigmoidf_(z);
                dz *= sg * (1.0f + z * (1.0f - sg));
            }
            o[j] = gamma[c] * invstd[gc + c] * (dz - dbeta[gc + c] * invM - xh * dgamma[gc + c] * invM);
        }
        store8<DT>(dx, r * xcs + xco + ch * 8, o);
    }
}

// ---------------------------------------------------------------------------------------------
// Fast paths of the three BatchNorm passes for C/8 a power of two <= 256 (every BN layer of the YOLOPoint family):
// a thread keeps ONE 8-channel chunk for all the rows it visits, so the per-channel parameters sit in registers,
// the index arithmetic is 32-bit shifts and several independent 16-byte loads are in flight per thread.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float fast_sigmoid(float z) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z * -1.4426950408889634f)); }

template <int DT, int MODE>
__global__ __launch_bounds__(256) void col_reduce_fast_kernel(const char* __restrict__ raw, int rcs, int rco, const char* __restrict__ dy,
                                                              int dcs, int dco, unsigned Mg, int nbg, int C, int lg, unsigned rows_per_blk,
                                                              const float* __restrict__ mean, const float* __restrict__ invstd,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta, int act,
                                                              float* __restrict__ part) {
    __shared__ float red[256][17];
    const int t = threadIdx.x;
    const int ch = t & ((1 << lg) - 1), rl = t >> lg, rlanes = 256 >> lg;
    float s0[8], s1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s0[j] = s1[j] = 0.f;
    float mu[8], is[8], ga[8], be[8];
    const int grp = blockIdx.x / nbg, gblk = blockIdx.x - grp * nbg;      // statistics group: Mg rows, nbg workgroups
    if constexpr (MODE == 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            mu[j] = mean[grp * C + ch * 8 + j]; is[j] = invstd[grp * C + ch * 8 + j];
            ga[j] = gamma[ch * 8 + j]; be[j] = beta[ch * 8 + j];
        }
    }
    const unsigned r0 = grp * Mg + gblk * rows_per_blk;
    const unsigned r1 = ((gblk + 1) * rows_per_blk < Mg) ? r0 + rows_per_blk : (grp + 1) * Mg;
    auto body = [&](const float (&x)[8], const float (&g)[8]) {
        if constexpr (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { s0[j] += x[j]; s1[j] += x[j] * x[j]; }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float xh = (x[j] - mu[j]) * is[j];
                float dz = g[j];
                if (act == YP_ACT_SILU) {
                    const float z = xh * ga[j] + be[j];
                    const float sg = fast_sigmoid(z);
                    dz *= sg * (1.0f + z * (1.0f - sg));
                }
                s0[j] += dz; s1[j] += dz * xh;
            }
        }
    };
    unsigned r = r0 + rl;
    for (; r + 3 * rlanes < r1; r += 4 * rlanes) {          // four independent rows in flight
        float x[4][8], g[4][8];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            load8<DT>(raw, (size_t)(r + u * rlanes) * rcs + rco + ch * 8, x[u]);
            if constexpr (MODE == 1) load8<DT>(dy, (size_t)(r + u * rlanes) * dcs + dco + ch * 8, g[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) body(x[u], g[u]);
    }
    for (; r < r1; r += rlanes) {
        float x[8], g[8];
        load8<DT>(raw, (size_t)r * rcs + rco + ch * 8, x);
        if constexpr (MODE == 1) load8<DT>(dy, (size_t)r * dcs + dco + ch * 8, g);
        body(x, g);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[t][j] = s0[j]; red[t][8 + j] = s1[j]; }
    __syncthreads();
    for (int hw = rlanes >> 1; hw > 0; hw >>= 1) {          // tree over the row lanes
        if (rl < hw) {
#pragma unroll
            for (int j = 0; j < 16; ++j) red[t][j] += red[t + (hw << lg)][j];
        }
        __syncthreads();
    }
    if (rl == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            part[((size_t)0 * C + ch * 8 + j) * gridDim.x + blockIdx.x] = red[t][j];
            part[((size_t)1 * C + ch * 8 + j) * gridDim.x + blockIdx.x] = red[t][8 + j];
        }
    }
}

template <int DT, bool BWD>
__global__ __launch_bounds__(256) void bn_rows_fast_kernel(const char* __restrict__ raw, int rcs, int rco, const char* __restrict__ dy, int dcs, int dco,
                                                           char* __restrict__ out, int ocs, int oco, const char* __restrict__ res, int scs, int sco,
                                                           unsigned M, int C, int lg, const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta, int act,
                                                           const float* __restrict__ dgamma, const float* __restrict__ dbeta,
                                                           unsigned char* __restrict__ q8, int qcs, int qco, const float* __restrict__ qscale,
                                                           float* __restrict__ qamax, char* __restrict__ gsum = nullptr, int gcs = 0, int gco = 0, int gacc = 0) {
    // gsum != nullptr (BWD only): the shortcut of a Bottleneck -- out = x + act(bn(raw)) -- hands dy on to the gradient of x unchanged:
    // gsum[row] (+)= dy[row] in this pass, which reads dy anyway (a separate add pass re-read it: yp_add_views, 13 / 39 launches per step)
    // blockIdx.y = statistics group: M rows each, mean / invstd / dgamma / dbeta [groups][C]
    // q8 != nullptr (fp8 training): the result is ALSO written as 1-byte values q8[row * qcs + qco + c] = saturate(result / *qscale) --
    // e4m3 for the forward output (the next Conv's activation), e5m2 for the backward's dx (the dgrad's output gradient) -- and its
    // max|result| recorded into qamax (csrc/fp8.hip): the separate quantisation pass over the tensor disappears.
    const int t = threadIdx.x;
    const int ch = t & ((1 << lg) - 1), rl = t >> lg, rlanes = 256 >> lg;
    float mu[8], is[8], ga[8], be[8], k0[8], k1[8];
    const float invM = 1.0f / (float)M;
    const int gc = blockIdx.y * C;
    {
        const size_t roff = (size_t)blockIdx.y * M;
        raw += roff * rcs * esize<DT>();
        if (out != nullptr) out += roff * ocs * esize<DT>();      // (out == nullptr: only the 1-byte twin is written -- no 16-bit reader)
        if (dy != nullptr) dy += roff * dcs * esize<DT>();
        if (res != nullptr) res += roff * scs * esize<DT>();
        if (q8 != nullptr) q8 += roff * qcs;
        if (gsum != nullptr) gsum += roff * gcs * esize<DT>();
    }
    const float qinv = q8 != nullptr ? 1.0f / qscale[0] : 0.f;
    float qmx = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = ch * 8 + j;
        mu[j] = mean[gc + c]; is[j] = invstd[gc + c]; ga[j] = gamma[c]; be[j] = beta[c];
        if constexpr (BWD) { k0[j] = dbeta[gc + c] * invM; k1[j] = dgamma[gc + c] * invM; }
    }
    const unsigned stride = gridDim.x * rlanes;
    auto body = [&](size_t r, const float (&x)[8], const float (&g)[8]) {
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float xh = (x[j] - mu[j]) * is[j];
            if constexpr (BWD) {
                float dz = g[j];
                if (act == YP_ACT_SILU) {
                    const float z = xh * ga[j] + be[j];
                    const float sg = fast_sigmoid(z);
                    dz *= sg * (1.0f + z * (1.0f - sg));
                }
                o[j] = ga[j] * is[j] * (dz - k0[j] - xh * k1[j]);
            } else {
                float z = xh * ga[j] + be[j];
                if (act == YP_ACT_SILU) z = z * fast_sigmoid(z);
                o[j] = z + g[j];
            }
        }
        if (out != nullptr) store8<DT>(out, r * ocs + oco + ch * 8, o);
        if constexpr (BWD) {
            if (gsum != nullptr) {
                float a8[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) a8[j] = g[j];
                if (gacc) {
                    float old8[8];
                    load8<DT>(gsum, r * gcs + gco + ch * 8, old8);
#pragma unroll
                    for (int j = 0; j < 8; ++j) a8[j] = old8[j] + g[j];       // (same operand order as yp_add_views: dst + src)
                }
                store8<DT>(gsum, r * gcs + gco + ch * 8, a8);
            }
        }
        if (q8 != nullptr) {
#pragma unroll
            for (int j = 0; j < 8; ++j) qmx = fmaxf(qmx, fabsf(o[j]));
            uint2 pk;
            pk.x = yp_fp8_pack4<BWD ? 1 : 0>(o[0] * qinv, o[1] * qinv, o[2] * qinv, o[3] * qinv);
            pk.y = yp_fp8_pack4<BWD ? 1 : 0>(o[4] * qinv, o[5] * qinv, o[6] * qinv, o[7] * qinv);
            *reinterpret_cast<uint2*>(q8 + r * qcs + qco + ch * 8) = pk;
        }
    };
    unsigned r = blockIdx.x * rlanes + rl;
    for (; (size_t)r + stride < M; r += 2 * stride) {       // two independent rows in flight
        float x[2][8], g[2][8];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const size_t rr = (size_t)r + u * stride;
            load8<DT>(raw, rr * rcs + rco + ch * 8, x[u]);
            if constexpr (BWD) load8<DT>(dy, rr * dcs + dco + ch * 8, g[u]);
            else if (res != nullptr) load8<DT>(res, rr * scs + sco + ch * 8, g[u]);
            else {
#pragma unroll
                for (int j = 0; j < 8; ++j) g[u][j] = 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) body((size_t)r + u * stride, x[u], g[u]);
    }
    if (r < M) {
        float x[8], g[8];
        load8<DT>(raw, (size_t)r * rcs + rco + ch * 8, x);
        if constexpr (BWD) load8<DT>(dy, (size_t)r * dcs + dco + ch * 8, g);
        else if (res != nullptr) load8<DT>(res, (size_t)r * scs + sco + ch * 8, g);
        else {
#pragma unroll
            for (int j = 0; j < 8; ++j) g[j] = 0.f;
        }
        body(r, x, g);
    }
    if (q8 != nullptr) yp_block_amax(qmx, qamax);
}

// out[pix, c] (+)= sum of the 2x2 block of `in` it was upsampled to  (backward of nn.Upsample(2,'nearest'))
template <int DT>
__global__ void ups2_bwd_kernel(const char* __restrict__ in, int ics, int ico, char* __restrict__ out, int ocs, int oco, int B, int H,
                                int W, int C, int accumulate) {
    const int chunks = C / 8;
    const size_t n = (size_t)B * H * W * chunks;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int ch = (int)(i % chunks);
        const size_t pix = i / chunks;
        const int w = (int)(pix % W), h = (int)((pix / W) % H);
        const size_t b = pix / ((size_t)W * H);
        float acc[8];
        if (accumulate) load8<DT>(out, pix * ocs + oco + ch * 8, acc);
        else {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        }
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                float v[8];
                load8<DT>(in, ((b * 2 * H + 2 * h + dy) * 2 * W + 2 * w + dx) * ics + ico + ch * 8, v);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += v[j];
            }
        store8<DT>(out, pix * ocs + oco + ch * 8, acc);
    }
}

// dst (+)= src elementwise over an NHWC view (gradient fan-in of a tensor with several consumers)
template <int DT>
__global__ void add_views_kernel(const char* __restrict__ src, int scs, int sco, char* __restrict__ dst, int dcs, int dco, size_t M,
                                 int C, int accumulate) {
    const int chunks = C / 8;
    const size_t n = M * chunks;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int ch = (int)(i % chunks);
        const size_t r = i / chunks;
        float a[8];
        load8<DT>(src, r * scs + sco + ch * 8, a);
        if (accumulate) {
            float d[8];
            load8<DT>(dst, r * dcs + dco + ch * 8, d);
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] += d[j];
        }
        store8<DT>(dst, r * dcs + dco + ch * 8, a);
    }
}

// backward of MaxPool2d(5,1,2): the gradient of each output pixel goes to the first maximum of its window
// (row-major scan, like ATen's CPU kernel).  Pass 1 records, per output element, which of the 25 window
// positions is that maximum; pass 2 gathers: an input element sums dy of the <= 25 outputs that selected it.
template <int DT>
__global__ void maxpool5_argmax_kernel(const char* __restrict__ x, int xcs, int xco, unsigned char* __restrict__ arg, int B, int H, int W, int C) {
    using T = typename Sc<DT>::t;
    const size_t n = (size_t)B * H * W * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const size_t pix = i / C;
        const int ow = (int)(pix % W), oh = (int)((pix / W) % H);
        const size_t b = pix / ((size_t)W * H);
        const T* xb = reinterpret_cast<const T*>(x) + (b * H * W) * xcs + xco + c;
        float best = -3.0e38f;
        int code = 0;
        for (int dy = -2; dy <= 2; ++dy)
            for (int dx = -2; dx <= 2; ++dx) {
                const int yy = oh + dy, xx = ow + dx;
                if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
                const float v = (float)xb[((size_t)yy * W + xx) * xcs];
                if (v > best) { best = v; code = (dy + 2) * 5 + (dx + 2); }
            }
        arg[i] = (unsigned char)code;
    }
}
template <int DT>
__global__ void maxpool5_bwd_kernel(const unsigned char* __restrict__ arg, const char* __restrict__ dy, int dcs, int dco, char* __restrict__ dx,
                                    int gcs, int gco, int B, int H, int W, int C, int accumulate) {
    using T = typename Sc<DT>::t;
    const size_t n = (size_t)B * H * W * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const size_t pix = i / C;
        const int w = (int)(pix % W), h = (int)((pix / W) % H);
        const size_t b = pix / ((size_t)W * H);
        const T* gb = reinterpret_cast<const T*>(dy) + (b * H * W) * dcs + dco + c;
        float acc = 0.f;
        for (int oh = max(h - 2, 0); oh <= min(h + 2, H - 1); ++oh)
            for (int ow = max(w - 2, 0); ow <= min(w + 2, W - 1); ++ow) {
                const int code = (h - oh + 2) * 5 + (w - ow + 2);          // position of (h, w) inside the window of (oh, ow)
                if (arg[((b * H + oh) * W + ow) * C + c] == code) acc += (float)gb[((size_t)oh * W + ow) * dcs];
            }
        T* o = reinterpret_cast<T*>(dx) + pix * gcs + gco + c;
        *o = (T)((accumulate ? (float)*o : 0.f) + acc);
    }
}

// The same two passes with 8 channels per thread (16-byte loads, 8-byte code words): the one-element-per-thread kernels above took
// 18 + 30 us per pool for SPPF's 3.3 MB maps (three pools per step).  Same scan order, same strict comparison, same accumulation order.
template <int DT>
__global__ __launch_bounds__(256) void maxpool5_argmax8_kernel(const char* __restrict__ x, int xcs, int xco, unsigned char* __restrict__ arg, int B, int H, int W, int C) {
    const unsigned chunks = (unsigned)C >> 3;
    const unsigned n = (unsigned)B * H * W * chunks;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const unsigned ch = i % chunks, pix = i / chunks;
        const int ow = (int)(pix % (unsigned)W), oh = (int)((pix / (unsigned)W) % (unsigned)H);
        const unsigned b = pix / (unsigned)(W * H);
        float best[8];
        unsigned code[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { best[j] = -3.0e38f; code[j] = 0; }
        for (int dy = -2; dy <= 2; ++dy) {
            const int yy = oh + dy;
            if (yy < 0 || yy >= H) continue;
#pragma unroll
            for (int dx = -2; dx <= 2; ++dx) {
                const int xx = ow + dx;
                if (xx < 0 || xx >= W) continue;
                float v[8];
                load8<DT>(x, ((size_t)(b * H + yy) * W + xx) * xcs + xco + ch * 8, v);
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (v[j] > best[j]) { best[j] = v[j]; code[j] = (unsigned)((dy + 2) * 5 + (dx + 2)); }
            }
        }
        uint2 pk;
        pk.x = code[0] | (code[1] << 8) | (code[2] << 16) | (code[3] << 24);
        pk.y = code[4] | (code[5] << 8) | (code[6] << 16) | (code[7] << 24);
        *reinterpret_cast<uint2*>(arg + (size_t)pix * C + ch * 8) = pk;
    }
}
template <int DT>
__global__ __launch_bounds__(256) void maxpool5_bwd8_kernel(const unsigned char* __restrict__ arg, const char* __restrict__ dy, int dcs, int dco, char* __restrict__ dx,
                                                            int gcs, int gco, int B, int H, int W, int C, int accumulate) {
    const unsigned chunks = (unsigned)C >> 3;
    const unsigned n = (unsigned)B * H * W * chunks;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const unsigned ch = i % chunks, pix = i / chunks;
        const int w = (int)(pix % (unsigned)W), h = (int)((pix / (unsigned)W) % (unsigned)H);
        const unsigned b = pix / (unsigned)(W * H);
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        for (int oh = max(h - 2, 0); oh <= min(h + 2, H - 1); ++oh)
            for (int ow = max(w - 2, 0); ow <= min(w + 2, W - 1); ++ow) {
                const unsigned code = (unsigned)((h - oh + 2) * 5 + (w - ow + 2));          // position of (h, w) inside the window of (oh, ow)
                const size_t op = (size_t)(b * H + oh) * W + ow;
                const uint2 a = *reinterpret_cast<const uint2*>(arg + op * C + ch * 8);
                float g[8];
                load8<DT>(dy, op * dcs + dco + ch * 8, g);
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if ((((j < 4 ? a.x : a.y) >> (8 * (j & 3))) & 255u) == code) acc[j] += g[j];
            }
        float o[8];
        if (accumulate) {
            load8<DT>(dx, (size_t)pix * gcs + gco + ch * 8, o);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = o[j] + acc[j];
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = 0.f + acc[j];
        }
        store8<DT>(dx, (size_t)pix * gcs + gco + ch * 8, o);
    }
}

// descriptor L2-norm backward: d = x/|x|  ->  dx = (g - d*(d.g)) / |x|   (fp32 views, one wave per pixel)
__global__ void l2norm_bwd_kernel(const float* __restrict__ x, int xcs, int xco, const float* __restrict__ g, int gcs, int gco,
                                  float* __restrict__ dx, int dcs, int dco, size_t npix, int C) {
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t p = wave; p < npix; p += nwaves) {
        const float* xs = x + p * xcs + xco;
        const float* gs = g + p * gcs + gco;
        float ss = 0.f, dg = 0.f;
        for (int c = lane; c < C; c += 64) { const float v = xs[c]; ss += v * v; dg += v * gs[c]; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { ss += __shfl_xor(ss, o, 64); dg += __shfl_xor(dg, o, 64); }
        const float nrm = sqrtf(ss);
        const float k = dg / (nrm * nrm * nrm);
        float* ds = dx + p * dcs + dco;
        for (int c = lane; c < C; c += 64) ds[c] = gs[c] / nrm - xs[c] * k;
    }
}

// fp32 [B,na,ny,nx,no] gradient of the permuted Detect output -> NHWC view [B,ny,nx,na*no (+pad)] in dtype
template <int DT>
__global__ void detect_bwd_pack_kernel(const float* __restrict__ gx, const float* __restrict__ scale, int B, int na, int no, int ny, int nx,
                                       typename Sc<DT>::t* __restrict__ out, int cs, int co, int Cpad) {
    using T = typename Sc<DT>::t;
    const size_t n = (size_t)B * ny * nx * Cpad;
    const float sc = scale != nullptr ? scale[0] : 1.0f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cpad);
        const size_t pix = i / Cpad;
        const size_t hw = (size_t)ny * nx;
        const size_t b = pix / hw, rem = pix - b * hw;
        float v = 0.f;
        if (c < na * no) {
            const int a = c / no, o = c - a * no;
            v = gx[((b * na + a) * hw + rem) * no + o] * sc;
        }
        out[pix * cs + co + c] = (T)v;
    }
}

// NHWC view (optionally read through the 2x nearest upsample) -> channel-major, batch-minor copy
// out[c][h][w][b] (b padded with zeros to Bpad): the operand layout of "wgrad as a convolution", where
// the channel axis becomes the batch and the batch becomes the (contiguous) channel axis.
template <int DT>
__global__ void to_chwb_kernel(const typename Sc<DT>::t* __restrict__ in, int cs, int co, int ups, int B, int H, int W, int C,
                               typename Sc<DT>::t* __restrict__ out, int Bpad) {
    using T = typename Sc<DT>::t;
    const int Hp = H >> ups, Wp = W >> ups;
    const size_t n = (size_t)C * H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);                       // consecutive threads: consecutive channels of one pixel
        const size_t pix = i / C;
        const int w = (int)(pix % W), h = (int)(pix / W);
        T* o = out + (((size_t)c * H + h) * W + w) * Bpad;
        for (int b = 0; b < Bpad; ++b) {
            T v = (T)0.f;
            if (b < B) v = in[(((size_t)b * Hp + (h >> ups)) * Wp + (w >> ups)) * cs + co + c];
            o[b] = v;
        }
    }
}

// fp32 NHWC view -> dtype NHWC view (gradient hand-over from the fp32 head outputs to the 16-bit conv operands)
template <int DT>
__global__ void cast_from_f32_kernel(const float* __restrict__ in, int ics, int ico, char* __restrict__ out, int ocs, int oco, size_t M, int C) {
    const int chunks = C / 8;
    const size_t n = M * chunks;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int ch = (int)(i % chunks);
        const size_t r = i / chunks;
        float v[8];
        load8<YP_F32>(reinterpret_cast<const char*>(in), r * ics + ico + ch * 8, v);
        store8<DT>(out, r * ocs + oco + ch * 8, v);
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------
#define YP_DT_SWITCH(dtype, CALL)                                             \
    switch (dtype) {                                                          \
        case YP_F16: { constexpr int DT = YP_F16; CALL; } break;              \
        case YP_BF16: { constexpr int DT = YP_BF16; CALL; } break;            \
        case YP_F32: { constexpr int DT = YP_F32; CALL; } break;              \
        default: YP_REQUIRE(false, "bad dtype %d", dtype);                    \
    }

static int check_view8(const YpView& v, const char* what) {
    YP_REQUIRE(v.ptr && v.C > 0 && v.C % 8 == 0 && v.cstride % 8 == 0 && v.coff % 8 == 0, "%s: view must be 8-channel aligned", what);
    return YP_OK;
}

// fast-path geometry: log2(C/8) when C/8 is a power of two <= 256 (else -1); partial-sum workgroups for M rows
static int fast_lg(int C) {
    const int chunks = C / 8;
    if (C % 8 || chunks < 1 || chunks > 256 || (chunks & (chunks - 1))) return -1;
    int lg = 0;
    while ((1 << lg) < chunks) ++lg;
    return lg;
}
static int fast_reduce_blocks(size_t M, int lg, unsigned* rows_per_blk, size_t cap = 2048) {
    const unsigned rlanes = 256u >> lg;
    size_t nblk = (M + 4 * rlanes - 1) / (4 * rlanes);       // >= 4 rows per thread
    if (nblk > cap) nblk = cap;
    if (nblk < 1) nblk = 1;
    unsigned rpb = (unsigned)((M + nblk - 1) / nblk);
    rpb = (rpb + rlanes - 1) / rlanes * rlanes;
    *rows_per_blk = rpb;
    return (int)((M + rpb - 1) / rpb);
}

extern "C" size_t yp_bn_workspace_bytes(int B, int H, int W, int C) {
    const size_t M = (size_t)B * H * W;
    size_t nblk = (M + BN_ROWS - 1) / BN_ROWS;
    if (nblk < 2048) nblk = 2048;                              // the fast reduction uses up to 2048 workgroups
    return align_up(nblk * 2 * (size_t)C * sizeof(float), 256);
}

extern "C" int yp_bn_stats_grouped(YpView raw, int dtype, int B, int groups, float eps, float momentum, float* mean, float* invstd, float* running_mean,
                                   float* running_var, void* ws, size_t ws_bytes, void* stream) {
    if (int rc = check_view8(raw, "yp_bn_stats")) return rc;
    YP_REQUIRE(mean && invstd && ws && B > 0 && raw.C <= 2048 && groups >= 1 && groups <= 8 && B % groups == 0, "yp_bn_stats: bad arguments");
    YP_REQUIRE(ws_bytes >= yp_bn_workspace_bytes(B, raw.H, raw.W, raw.C), "yp_bn_stats: workspace too small");
    const size_t M = (size_t)B * raw.H * raw.W, Mg = M / groups;
    int nbg = (int)((Mg + BN_ROWS - 1) / BN_ROWS);                // workgroups per statistics group
    hipStream_t st = (hipStream_t)stream;
    const int lg = fast_lg(raw.C);
    if (lg >= 0 && M < (1ull << 31)) {
        unsigned rpb;
        nbg = fast_reduce_blocks(Mg, lg, &rpb, 2048 / groups);
        YP_DT_SWITCH(dtype, (col_reduce_fast_kernel<DT, 0><<<nbg * groups, 256, 0, st>>>((const char*)raw.ptr, raw.cstride, raw.coff, nullptr, 0, 0, (unsigned)Mg, nbg,
                                                                                          raw.C, lg, rpb, nullptr, nullptr, nullptr, nullptr, 0, (float*)ws)));
    } else {
        YP_DT_SWITCH(dtype, (col_reduce_kernel<DT, 0><<<nbg * groups, 256, 0, st>>>((const char*)raw.ptr, raw.cstride, raw.coff, nullptr, 0, 0, Mg, nbg, raw.C,
                                                                                     nullptr, nullptr, nullptr, nullptr, 0, (float*)ws)));
    }
    if (nbg <= 256) bn_stats_finalize_kernel<true><<<(raw.C + 3) / 4, 256, 0, st>>>((const float*)ws, nbg, raw.C, (double)Mg, eps, momentum, mean, invstd, running_mean, running_var, groups);
    else bn_stats_finalize_kernel<false><<<raw.C, 256, 0, st>>>((const float*)ws, nbg, raw.C, (double)Mg, eps, momentum, mean, invstd, running_mean, running_var, groups);
    YP_CHECK_HIP(hipGetLastError());
    return YP_OK;
}
extern "C" int yp_bn_stats(YpView raw, int dtype, int B, float eps, float momentum, float* mean, float* invstd, float* running_mean,
                           float* running_var, void* ws, size_t ws_bytes, void* stream) {
    return yp_bn_stats_grouped(raw, dtype, B, 1, eps, momentum, mean, invstd, running_mean, running_var, ws, ws_bytes, stream);
}

extern "C" int yp_bn_finalize_grouped(const float* partial, int rows, int groups, int C, double M, float eps, float momentum, float* mean, float* invstd,
                                      float* running_mean, float* running_var, void* stream) {
    YP_REQUIRE(partial && mean && invstd && rows > 0 && C > 0 && M > 0 && groups >= 1 && rows % groups == 0, "yp_bn_finalize: bad arguments");
    const int rg = rows / groups;
    if (rg <= 256) bn_stats_finalize_kernel<true><<<(C + 3) / 4, 256, 0, (hipStream_t)stream>>>(partial, rg, C, M / groups, eps, momentum, mean, invstd, running_mean, running_var, groups);
    else bn_stats_finalize_kernel<false><<<C, 256, 0, (hipStream_t)stream>>>(partial, rg, C, M / groups, eps, momentum, mean, invstd, running_mean, running_var, groups);
    YP_CHECK_HIP(hipGetLastError());
    return YP_OK;
}
extern "C" int yp_bn_finalize(const float* partial, int rows, int C, double M, float eps, float momentum, float* mean, float* invstd, float* running_mean,
                              float* running_var, void* stream) {
    return yp_bn_finalize_grouped(partial, rows, 1, C, M, eps, momentum, mean, invstd, running_mean, running_var, stream);
}

extern "C" int yp_bn_act_apply_grouped(YpView raw, YpView out, YpView res, int dtype, int B, int groups, const float* mean, const float* invstd,
                                       const float* gamma, const float* beta, int act, void* stream) {
    YpView none{};
    return yp_bn_act_apply_grouped_q8(raw, out, res, dtype, B, groups, mean, invstd, gamma, beta, act, none, nullptr, nullptr, stream);
}
extern "C" int yp_bn_act_apply_grouped_q8(YpView raw, YpView out, YpView res, int dtype, int B, int groups, const float* mean, const float* invstd,
                                          const float* gamma, const float* beta, int act, YpView q8, const float* q_scale, float* q_amax, void* stream) {
    if (int rc = check_view8(raw, "yp_bn_act_apply")) return rc;
    // out.ptr == NULL with a twin: the 16-bit result is not stored at all (fp8 training: every consumer of this tensor reads the 1-byte twin)
    const bool twin_only = out.ptr == nullptr && q8.ptr != nullptr;
    if (!twin_only) { if (int rc = check_view8(out, "yp_bn_act_apply")) return rc; }
    YP_REQUIRE((twin_only || out.C == raw.C) && (res.C == 0 || res.C == raw.C) && mean && invstd && gamma && beta && groups >= 1 && B % groups == 0, "yp_bn_act_apply: bad arguments");
    const size_t M = (size_t)B * raw.H * raw.W, Mg = M / groups;
    hipStream_t st = (hipStream_t)stream;
    const int g = grid_for(M * (raw.C / 8), 256);
    const int lg = fast_lg(raw.C);
    YP_REQUIRE(q8.ptr == nullptr || (lg >= 0 && M < (1ull << 31) && q8.C == raw.C && q8.H == raw.H && q8.W == raw.W && q8.cstride % 8 == 0 && q8.coff % 8 == 0 && q_scale),
               "yp_bn_act_apply: the 1-byte twin needs C/8 a power of two <= 256, a matching view and a scale");
    YP_REQUIRE(!twin_only || (lg >= 0 && M < (1ull << 31)), "yp_bn_act_apply: a twin-only output needs the fast path");
    if (lg >= 0 && M < (1ull << 31)) {
        const int gf = grid_for((Mg * (raw.C / 8) + 1) / 2, 256, (size_t)(256 * 16) / groups);      // ~2 rows per thread
        YP_DT_SWITCH(dtype, (bn_rows_fast_kernel<DT, false><<<dim3(gf, groups), 256, 0, st>>>((const char*)raw.ptr, raw.cstride, raw.coff, nullptr, 0, 0, (char*)out.ptr, out.cstride,
                                                                               out.coff, res.C ? (const char*)res.ptr : nullptr, res.cstride, res.coff, (unsigned)Mg, raw.C, lg,
                                                                               mean, invstd, gamma, beta, act, nullptr, nullptr, (unsigned char*)q8.ptr, q8.cstride, q8.coff,
                                                                               q_scale, q_amax)));
    } else {
        YP_DT_SWITCH(dtype, (bn_apply_kernel<DT><<<g, 256, 0, st>>>((const char*)raw.ptr, raw.cstride, raw.coff, (char*)out.ptr, out.cstride, out.coff,
                                                                    res.C ? (const char*)res.ptr : nullptr, res.cstride, res.coff, M, Mg, raw.C, mean,
                                                                    invstd, gamma, beta, act)));
    }
    YP_CHECK_HIP(hipGetLastError());
    return YP_OK;
}
extern "C" int yp_bn_act_apply(YpView raw, YpView out, YpView res, int dtype, int B, const float* mean, const float* invstd,
                               const float* gamma, const float* beta, int act, void* stream) {
    return yp_bn_act_apply_grouped(raw, out, res, dtype, B, 1, mean, invstd, gamma, beta, act, stream);
}

extern "C" int yp_bn_act_bwd_grouped(YpView raw, YpView dy, YpView dx, int dtype, int B, int groups, const float* mean, const float* invstd,
                                     const float* gamma, const float* beta, int act, float* dgamma, float* dbeta, int accumulate_param_grads,
                                     void* ws, size_t ws_bytes, void* stream) {
    YpView none{};
    return yp_bn_act_bwd_grouped_q8(raw, dy, dx, dtype, B, groups, mean, invstd, gamma, beta, act, dgamma, dbeta, accumulate_param_grads, ws, ws_bytes, none, nullptr,
                                    nullptr, stream);
}
static int bn_act_bwd_impl(YpView raw, YpView dy, YpView dx, int dtype, int B, int groups, const float* mean, const float* invstd,
                           const float* gamma, const float* beta, int act, float* dgamma, float* dbeta, int accumulate_param_grads,
                           void* ws, size_t ws_bytes, YpView q8, const float* q_scale, float* q_amax, YpView gsum, int gacc, void* stream);
extern "C" int yp_bn_act_bwd_grouped_q8(YpView raw, YpView dy, YpView dx, int dtype, int B, int groups, const float* mean, const float* invstd,
                                        const float* gamma, const float* beta, int act, float* dgamma, float* dbeta, int accumulate_param_grads,
                                        void* ws, size_t ws_bytes, YpView q8, const float* q_scale, float* q_amax, void* stream) {
    YpView none{};
    return bn_act_bwd_impl(raw, dy, dx, dtype, B, groups, mean, invstd, gamma, beta, act, dgamma, dbeta, accumulate_param_grads, ws, ws_bytes, q8, q_scale, q_amax,
                           none, 0, stream);
}
// gsum.ptr != nullptr: the shortcut gradient gsum (+)= dy rides in the row pass (YP_OP_BN_BWD with p1 / i5..i7)
static int bn_act_bwd_impl(YpView raw, YpView dy, YpView dx, int dtype, int B, int groups, const float* mean, const float* invstd,
                           const float* gamma, const float* beta, int act, float* dgamma, float* dbeta, int accumulate_param_grads,
                           void* ws, size_t ws_bytes, YpView q8, const float* q_scale, float* q_amax, YpView gsum, int gacc, void* stream) {
    if (int rc = check_view8(raw, "yp_bn_act_bwd")) return rc;
    if (int rc = check_view8(dy, "yp_bn_act_bwd")) return rc;
    const bool twin_only = dx.ptr == nullptr && q8.ptr != nullptr;       // (as yp_bn_act_apply: dx only as its 1-byte twin)
    if (!twin_only) { if (int rc = check_view8(dx, "yp_bn_act_bwd")) return rc; }
    YP_REQUIRE(dy.C == raw.C && (twin_only || dx.C == raw.C) && mean && invstd && gamma && beta && dgamma && dbeta && ws && raw.C <= 2048 && groups >= 1 && groups <= 8 && B % groups == 0,
               "yp_bn_act_bwd: bad arguments");
    YP_REQUIRE(ws_bytes >= yp_bn_workspace_bytes(B, raw.H, raw.W, raw.C) + 2 * (size_t)groups * raw.C * 4, "yp_bn_act_bwd: workspace too small");
    const size_t M = (size_t)B * raw.H * raw.W, Mg = M / groups;
    int nbg = (int)((Mg + BN_ROWS - 1) / BN_ROWS);
    hipStream_t st = (hipStream_t)stream;
    // this call's own per-group sums live at the end of the workspace (the parameter gradients may be accumulated)
    float* dg = (float*)((char*)ws + yp_bn_workspace_bytes(B, raw.H, raw.W, raw.C));
    float* db = dg + (size_t)groups * raw.C;
    const int lg = fast_lg(raw.C);
    const bool fastp = lg >= 0 && M < (1ull << 31);
    YP_REQUIRE(q8.ptr == nullptr || (fastp && q8.C == raw.C && q8.H == raw.H && q8.W == raw.W && q8.cstride % 8 == 0 && q8.coff % 8 == 0 && q_scale),
               "yp_bn_act_bwd: the 1-byte twin needs C/8 a power of two <= 256, a matching view and a scale");
    YP_REQUIRE(!twin_only || fastp, "yp_bn_act_bwd: a twin-only output needs the fast path");
    if (fastp) {
        unsigned rpb;
        nbg = fast_reduce_blocks(Mg, lg, &rpb, 2048 / groups);
        YP_DT_SWITCH(dtype, (col_reduce_fast_kernel<DT, 1><<<nbg * groups, 256, 0, st>>>((const char*)raw.ptr, raw.cstride, raw.coff, (const char*)dy.ptr, dy.cstride, dy.coff,
                                                                                          (unsigned)Mg, nbg, raw.C, lg, rpb, mean, invstd, gamma, beta, act, (float*)ws)));
    } else {
        YP_DT_SWITCH(dtype, (col_reduce_kernel<DT, 1><<<nbg * groups, 256, 0, st>>>((const char*)raw.ptr, raw.cstride, raw.coff, (const char*)dy.ptr, dy.cstride,
                                                                                     dy.coff, Mg, nbg, raw.C, mean, invstd, gamma, beta, act, (float*)ws)));
    }
    // per group (sum dz, sum dz*xhat) -> this call's dbeta / dgamma; their sums over the groups are (accumulated) into the parameter gradients
    if (nbg <= 256) pair_finalize_kernel<true><<<(raw.C + 3) / 4, 256, 0, st>>>((const float*)ws, nbg, raw.C, db, dg, dbeta, dgamma, accumulate_param_grads, groups);
    else pair_finalize_kernel<false><<<raw.C, 256, 0, st>>>((const float*)ws, nbg, raw.C, db, dg, dbeta, dgamma, accumulate_param_grads, groups);
    const int g = grid_for(M * (raw.C / 8), 256);
    if (fastp) {
        const int gf = grid_for((Mg * (raw.C / 8) + 1) / 2, 256, (size_t)(256 * 16) / groups);
        YP_DT_SWITCH(dtype, (bn_rows_fast_kernel<DT, true><<<dim3(gf, groups), 256, 0, st>>>((const char*)raw.ptr, raw.cstride, raw.coff, (const char*)dy.ptr, dy.cstride, dy.coff,
                                                                              (char*)dx.ptr, dx.cstride, dx.coff, nullptr, 0, 0, (unsigned)Mg, raw.C, lg, mean, invstd, gamma, beta,
                                                                              act, dg, db, (unsigned char*)q8.ptr, q8.cstride, q8.coff, q_scale, q_amax,
                                                                              (char*)gsum.ptr, gsum.cstride, gsum.coff, gacc)));
    } else {
        if (gsum.ptr != nullptr)
            if (int rc = yp_add_views(dy, gsum, dtype, B, gacc, stream)) return rc;
        YP_DT_SWITCH(dtype, (bn_bwd_apply_kernel<DT><<<g, 256, 0, st>>>((const char*)raw.ptr, raw.cstride, raw.coff, (const char*)dy.ptr, dy.cstride, dy.coff,
                                                                        (char*)dx.ptr, dx.cstride, dx.coff, M, Mg, raw.C, mean, invstd, gamma, beta, act, dg, db)));
    }
    YP_CHECK_HIP(hipGetLastError());
    return YP_OK;
}
extern "C" int yp_bn_act_bwd(YpView raw, YpView dy, YpView dx, int dtype, int B, const float* mean, const float* invstd,
                             const float* gamma, const float* beta, int act, float* dgamma, float* dbeta, int accumulate_param_grads,
                             void* ws, size_t ws_bytes, void* stream) {
    return yp_bn_act_bwd_grouped(raw, dy, dx, dtype, B, 1, mean, invstd, gamma, beta, act, dgamma, dbeta, accumulate_param_grads, ws, ws_bytes, stream);
}

extern "C" int yp_ups2_bwd(YpView in, YpView out, int dtype, int B, int accumulate, void* stream) {
    if (int rc = check_view8(in, "yp_ups2_bwd")) return rc;
    if (int rc = check_view8(out, "yp_ups2_bwd")) return rc;
    YP_REQUIRE(in.C == out.C && in.H == 2 * out.H && in.W == 2 * out.W, "yp_ups2_bwd: dims mismatch");
    const size_t n = (size_t)B * out.H * out.W * (out.C / 8);
    YP_DT_SWITCH(dtype, (ups2_bwd_kernel<DT><<<grid_for(n, 256), 256, 0, (hipStream_t)stream>>>((const char*)in.ptr, in.cstride, in.coff, (char*)out.ptr,
                                                                                               out.cstride, out.coff, B, out.H, out.W, out.C, accumulate)));
    YP_CHECK_HIP(hipGetLastError());
    return YP_OK;
}

extern "C" int yp_add_views(YpView src, YpView dst, int dtype, int B, int accumulate, void* stream) {
    if (int rc = check_view8(src, "yp_add_views")) return rc;
    if (int rc = check_view8(dst, "yp_add_views")) return rc;
    YP_REQUIRE(src.C == dst.C && src.H == dst.H && src.W == dst.W, "yp_add_views: dims mismatch");
    const size_t M = (size_t)B * src.H * src.W;
    YP_DT_SWITCH(dtype, (add_views_kernel<DT><<<grid_for(M * (src.C / 8), 256), 256, 0, (hipStream_t)stream>>>((const char*)src.ptr, src.cstride, src.coff,
                                                                                                              (char*)dst.ptr, dst.cstride, dst.coff, M, src.C, accumulate)));
    YP_CHECK_HIP(hipGetLastError());
    return YP_OK;
}

extern "C" size_t yp_maxpool5_bwd_workspace_bytes(int B, int H, int W, int C) { return align_up((size_t)B * H * W * C, 256); }

extern "C" int yp_maxpool5_bwd(YpView x, YpView dy, YpView dx, int dtype, int B, int accumulate, void* ws, size_t ws_bytes, void* stream) {
    YP_REQUIRE(x.ptr && dy.ptr && dx.ptr && ws && x.C == dy.C && x.C == dx.C && x.H == dy.H && x.H == dx.H && x.W == dy.W, "yp_maxpool5_bwd: bad views");
    YP_REQUIRE(ws_bytes >= yp_maxpool5_bwd_workspace_bytes(B, x.H, x.W, x.C), "yp_maxpool5_bwd: workspace too small");
    const size_t n = (size_t)B * x.H * x.W * x.C;
    hipStream_t st = (hipStream_t)stream;
    const auto al8 = [](const YpView& v) { return v.C % 8 == 0 && v.cstride % 8 == 0 && v.coff % 8 == 0 && ((size_t)v.ptr & 15) == 0; };
    if (al8(x) && al8(dy) && al8(dx) && ((size_t)ws & 7) == 0 && n / 8 < (1ull << 31)) {
        const size_t n8 = n / 8;
        YP_DT_SWITCH(dtype, (maxpool5_argmax8_kernel<DT><<<grid_for(n8, 256), 256, 0, st>>>((const char*)x.ptr, x.cstride, x.coff, (unsigned char*)ws, B, x.H, x.W, x.C)));
        YP_DT_SWITCH(dtype, (maxpool5_bwd8_kernel<DT><<<grid_for(n8, 256), 256, 0, st>>>((const unsigned char*)ws, (const char*)dy.ptr, dy.cstride, dy.coff, (char*)dx.ptr,
                                                                                         dx.cstride, dx.coff, B, x.H, x.W, x.C, accumulate)));
        YP_CHECK_HIP(hipGetLastError());
        return YP_OK;
    }
    YP_DT_SWITCH(dtype, (maxpool5_argmax_kernel<DT><<<grid_for(n, 256), 256, 0, st>>>((const char*)x.ptr, x.cstride, x.coff, (unsigned char*)ws, B, x.H, x.W, x.C)));
    YP_DT_SWITCH(dtype, (maxpool5_bwd_kernel<DT><<<grid_for(n, 256), 256, 0, st>>>((const unsigned char*)ws, (const char*)dy.ptr, dy.cstride, dy.coff, (char*)dx.ptr,
                                                                                   dx.cstride, dx.coff, B, x.H, x.W, x.C, accumulate)));
    YP_CHECK_HIP(hipGetLastError());
    return YP_OK;
}

extern "C" int yp_l2norm_bwd_f32(YpView x, YpView g, YpView dx, int B, int C, void* stream) {
    YP_REQUIRE(x.ptr && g.ptr && dx.ptr && C > 0 && C <= x.C && C <= g.C && C <= dx.C && x.H == g.H && x.W == g.W, "yp_l2norm_bwd_f32: bad views");
    const size_t npix = (size_t)B * x.H * x.W;
    l2norm_bwd_kernel<<<grid_for(npix * 64, 256), 256, 0, (hipStream_t)stream>>>((const float*)x.ptr, x.cstride, x.coff, (const float*)g.ptr, g.cstride, g.coff,
                                                                                  (float*)dx.ptr, dx.cstride, dx.coff, npix, C);
    YP_CHECK_HIP(hipGetLastError());
    return YP_OK;
}

extern "C" int yp_detect_bwd_pack(const float* gx, int B, int na, int no, YpView out, int dtype, const float* scale_dev, void* stream) {
    YP_REQUIRE(gx && out.ptr && B > 0 && na > 0 && no > 0 && out.C >= na * no, "yp_detect_bwd_pack: bad arguments");
    const size_t n = (size_t)B * out.H * out.W * out.C;
    YP_DT_SWITCH(dtype, (detect_bwd_pack_kernel<DT><<<grid_for(n, 256), 256, 0, (hipStream_t)stream>>>(gx, scale_dev, B, na, no, out.H, out.W,
                                                                                                      (typename Sc<DT>::t*)out.ptr, out.cstride, out.coff, out.C)));
    YP_CHECK_HIP(hipGetLastError());
    return YP_OK;
}

extern "C" int yp_to_chwb(YpView in, int dtype, int B, int C, void* out, int Bpad, void* stream) {
    YP_REQUIRE(in.ptr && out && B > 0 && C > 0 && C <= in.C && Bpad >= B, "yp_to_chwb: bad arguments");
    const int H = in.H << in.ups, W = in.W << in.ups;
    const size_t n = (size_t)C * H * W;
    YP_DT_SWITCH(dtype, (to_chwb_kernel<DT><<<grid_for(n, 256), 256, 0, (hipStream_t)stream>>>((const typename Sc<DT>::t*)in.ptr, in.cstride, in.coff, in.ups, B, H, W, C,
                                                                                              (typename Sc<DT>::t*)out, Bpad)));
    YP_CHECK_HIP(hipGetLastError());
    return YP_OK;
}

extern "C" int yp_col_sum(YpView v, int dtype, int B, float* out, int accumulate, void* ws, size_t ws_bytes, void* stream) {
    if (int rc = check_view8(v, "yp_col_sum")) return rc;
    YP_REQUIRE(out && ws && v.C <= 2048 && ws_bytes >= yp_bn_workspace_bytes(B, v.H, v.W, v.C) + (size_t)v.C * 4, "yp_col_sum: bad arguments / workspace");
    const size_t M = (size_t)B * v.H * v.W;
    int nblk = (int)((M + BN_ROWS - 1) / BN_ROWS);
    hipStream_t st = (hipStream_t)stream;
    const int lg = fast_lg(v.C);
    if (lg >= 0 && M < (1ull << 31)) {
        // (the generic kernel took 21-22 us for each Detect level's bias gradient, P5's 3200 rows as long as P3's 51200)
        unsigned rpb;
        nblk = fast_reduce_blocks(M, lg, &rpb);
        YP_DT_SWITCH(dtype, (col_reduce_fast_kernel<DT, 0><<<nblk, 256, 0, st>>>((const char*)v.ptr, v.cstride, v.coff, nullptr, 0, 0, (unsigned)M, nblk, v.C, lg, rpb,
                                                                                 nullptr, nullptr, nullptr, nullptr, 0, (float*)ws)));
    } else {
        YP_DT_SWITCH(dtype, (col_reduce_kernel<DT, 0><<<nblk, 256, 0, st>>>((const char*)v.ptr, v.cstride, v.coff, nullptr, 0, 0, M, nblk, v.C, nullptr, nullptr,
                                                                            nullptr, nullptr, 0, (float*)ws)));
    }
    if (nblk <= 256) pair_finalize_kernel<true><<<(v.C + 3) / 4, 256, 0, st>>>((const float*)ws, nblk, v.C, nullptr, nullptr, out, nullptr, accumulate, 1);
    else pair_finalize_kernel<false><<<v.C, 256, 0, st>>>((const float*)ws, nblk, v.C, nullptr, nullptr, out, nullptr, accumulate, 1);
    YP_CHECK_HIP(hipGetLastError());
    return YP_OK;
}

extern "C" int yp_cast_from_f32(YpView in, YpView out, int dtype, int B, void* stream) {
    if (int rc = check_view8(in, "yp_cast_from_f32")) return rc;
    if (int rc = check_view8(out, "yp_cast_from_f32")) return rc;
    YP_REQUIRE(in.C == out.C && in.H == out.H && in.W == out.W, "yp_cast_from_f32: dims mismatch");
    const size_t M = (size_t)B * in.H * in.W;
    YP_DT_SWITCH(dtype, (cast_from_f32_kernel<DT><<<grid_for(M * (in.C / 8), 256), 256, 0, (hipStream_t)stream>>>((const float*)in.ptr, in.cstride, in.coff,
                                                                                                                 (char*)out.ptr, out.cstride, out.coff, M, in.C)));
    YP_CHECK_HIP(hipGetLastError());
    return YP_OK;
}

template <int DT>
__global__ void pack_weight_kernel(const float* __restrict__ w, int Cout, int Cin, int R, int S, int c0, int Cj, int mode, int Cout_pad,
                                   char* __restrict__ dst, int Kpad, int Npad, const float* __restrict__ bias, float* __restrict__ bias_dst) {
    using sc = typename Sc<DT>::t;
    const size_t total = (size_t)(Npad + 1) * Kpad;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int n = (int)(i / Kpad), k = (int)(i - (size_t)n * Kpad);
        const float v = yp_pack_elem(w, Cout, Cin, R, S, c0, Cj, mode, Cout_pad, n, k);
        reinterpret_cast<sc*>(dst)[i] = (sc)v;
        if (bias_dst != nullptr && i < (size_t)Npad) bias_dst[i] = (bias != nullptr && (int)i < Cout) ? bias[i] : 0.f;
    }
}

extern "C" int yp_pack_weight(const float* w, int Cout, int Cin, int R, int S, int c0, int Cj, int mode, int Cout_pad, void* dst, int Kpad,
                              int Npad, int dtype, const float* bias, float* bias_dst, void* stream) {
    YP_REQUIRE(w && dst && Cout > 0 && Cin > 0 && R > 0 && S > 0 && c0 >= 0 && Cj > 0 && c0 + Cj <= Cin && mode >= 0 && mode <= 7, "yp_pack_weight: bad arguments");
    YP_REQUIRE(mode < 2 || mode > 3 || (Cin <= 4 && c0 == 0 && (mode == 3 || S % 2 == 0)), "yp_pack_weight: modes 2 / 3 pack an image-like filter (<= 4 input channels; 2: even width)");
    YP_REQUIRE(mode < 4 || (R == 3 && S == 3), "yp_pack_weight: modes 4..7 are the parity classes of a 3x3 stride-2 dgrad");
    const bool tr = mode == 1 || mode >= 4;                                // channel-transposed (dgrad) forms: rows = input channels
    const int taps = mode >= 4 ? (1 + ((mode - 4) >> 1)) * (1 + ((mode - 4) & 1)) : R * S;
    const int Cq = mode == 0 ? Cj : (tr ? Cout_pad : 4), Nreal = tr ? Cj : Cout;       // (modes 2 / 3: 4 k slots per filter pixel)
    YP_REQUIRE((!tr || Cout_pad >= Cout) && Kpad >= taps * Cq && Npad >= Nreal, "yp_pack_weight: packed dims %dx%d too small", Npad, Kpad);
    const size_t total = (size_t)(Npad + 1) * Kpad;
    YP_DT_SWITCH(dtype, (pack_weight_kernel<DT><<<grid_for(total, 256), 256, 0, (hipStream_t)stream>>>(w, Cout, Cin, R, S, c0, Cj, mode, Cout_pad, (char*)dst, Kpad, Npad,
                                                                                                     bias, bias_dst)));
    YP_CHECK_HIP(hipGetLastError());
    return YP_OK;
}

// Every packed filter copy of a parameter set in one launch (the optimizer step changes all masters at once): a table of
// yp_pack_weight argument sets; a workgroup converts 1024 consecutive elements of its entry's packed image (153 per-filter launches
// of ~3.3 us each per training step before).
template <int DT>
__global__ __launch_bounds__(256) void pack_weight_batch_kernel(const YpPackEntry* __restrict__ table, int n_entries) {
    using sc = typename Sc<DT>::t;
    const int bid = blockIdx.x;
    int e = 0;
    for (int lo = 0, hi = n_entries - 1; lo <= hi;) {       // last entry with blk0 <= bid
        const int mid = (lo + hi) >> 1;
        if (table[mid].blk0 <= bid) { e = mid; lo = mid + 1; } else hi = mid - 1;
    }
    const YpPackEntry en = table[e];
    const int Cout = (int)en.Cout, Cin = (int)en.Cin, R = (int)en.R, S = (int)en.S, c0 = (int)en.c0, Cj = (int)en.Cj, mode = (int)en.mode;
    const int Cout_pad = (int)en.Cout_pad, Kpad = (int)en.Kpad, Npad = (int)en.Npad;
    // four consecutive k of one packed row per thread (Kpad is a multiple of 4: they share their row), 32-bit index arithmetic (a packed
    // image has < 2^31 elements), one 8-byte store
    const unsigned total = (unsigned)(Npad + 1) * (unsigned)Kpad;
    const unsigned i = (unsigned)(bid - (int)en.blk0) * 1024u + 4u * threadIdx.x;
    if (i >= total) return;
    const int n = (int)(i / (unsigned)Kpad), k0 = (int)(i - (unsigned)n * (unsigned)Kpad);
    sc o[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) o[u] = (sc)yp_pack_elem(en.w, Cout, Cin, R, S, c0, Cj, mode, Cout_pad, n, k0 + u);
    if constexpr (sizeof(sc) == 2) *reinterpret_cast<uint2*>(reinterpret_cast<sc*>(en.dst) + i) = *reinterpret_cast<const uint2*>(o);
    else *reinterpret_cast<uint4*>(reinterpret_cast<sc*>(en.dst) + i) = *reinterpret_cast<const uint4*>(o);
    if (en.bias_dst != nullptr && i < (unsigned)Npad) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i + u < (unsigned)Npad) en.bias_dst[i + u] = (en.bias != nullptr && (int)(i + u) < Cout) ? en.bias[i + u] : 0.f;
    }
}

extern "C" int yp_pack_weight_batch(const YpPackEntry* table_dev, int n_entries, int total_blocks, int dtype, void* stream) {
    YP_REQUIRE(table_dev && n_entries > 0 && total_blocks > 0, "yp_pack_weight_batch: bad arguments");
    YP_DT_SWITCH(dtype, (pack_weight_batch_kernel<DT><<<total_blocks, 256, 0, (hipStream_t)stream>>>(table_dev, n_entries)));
    YP_CHECK_HIP(hipGetLastError());
    return YP_OK;
}

__global__ void wgrad_unpack_kernel(const float* __restrict__ dw, float* __restrict__ grad, int Cout, int Cin, int k, int c0, int creal, int Cout_pad) {
    const int kk = k * k;
    const size_t total = (size_t)Cout * creal * kk;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int tap = (int)(i % kk);
        const int ci = (int)((i / kk) % creal);
        const int co = (int)(i / ((size_t)kk * creal));
        grad[((size_t)co * Cin + c0 + ci) * kk + tap] = dw[((size_t)ci * kk + tap) * Cout_pad + co];
    }
}

extern "C" int yp_wgrad_unpack(const float* dw, float* grad, int Cout, int Cin, int k, int c0, int creal, int Cout_pad, void* stream) {
    YP_REQUIRE(dw && grad && Cout > 0 && Cin > 0 && k > 0 && c0 >= 0 && creal > 0 && c0 + creal <= Cin && Cout_pad >= Cout, "yp_wgrad_unpack: bad arguments");
    wgrad_unpack_kernel<<<grid_for((size_t)Cout * creal * k * k, 256), 256, 0, (hipStream_t)stream>>>(dw, grad, Cout, Cin, k, c0, creal, Cout_pad);
    YP_CHECK_HIP(hipGetLastError());
    return YP_OK;
}

// All weight gradients of a backward pass in one launch: entry e transposes dw_e [rows = creal*k*k][cout_pad] (summed over `split`
// partial copies `pstride` floats apart) into grad_e[co][out_off + row] (row stride out_stride = Cin*k*k) through 32 x 32 LDS
// tiles, so both sides are coalesced (the per-layer kernel above reads with a stride of cout_pad floats; 84 of those launches
// were 570 us of a 6.7 ms backward).  Tiles are numbered across entries; tile0 is an entry's first tile.
__global__ __launch_bounds__(256) void wgrad_unpack_batch_kernel(const YpUnpackEntry* __restrict__ table, int n_entries) {
    __shared__ float tile[32][33];
    const int tid = blockIdx.x;
    int e = 0;
    for (int lo = 0, hi = n_entries - 1; lo <= hi;) {       // last entry with tile0 <= tid
        const int mid = (lo + hi) >> 1;
        if (table[mid].tile0 <= tid) { e = mid; lo = mid + 1; } else hi = mid - 1;
    }
    const YpUnpackEntry en = table[e];
    const int tiles_c = (int)((en.cout + 31) / 32);
    const int local = tid - (int)en.tile0;
    const int r0 = (local / tiles_c) * 32, c0 = (local % tiles_c) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = r0 + ty + 8 * j, c = c0 + tx;
        float v = 0.f;
        if (r < en.rows && c < en.cout) {
            const float* src = en.dw + (size_t)r * en.cout_pad + c;
            for (int p = 0; p < (int)en.split; ++p) v += src[(size_t)p * en.pstride];
        }
        tile[ty + 8 * j][tx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = c0 + ty + 8 * j, r = r0 + tx;
        if (r < en.rows && c < en.cout) en.grad[(size_t)c * en.out_stride + en.out_off + r] = tile[tx][ty + 8 * j];
    }
}

extern "C" int yp_wgrad_unpack_batch(const YpUnpackEntry* table_dev, int n_entries, int total_tiles, void* stream) {
    YP_REQUIRE(table_dev && n_entries > 0 && total_tiles > 0, "yp_wgrad_unpack_batch: bad arguments");
    wgrad_unpack_batch_kernel<<<total_tiles, 256, 0, (hipStream_t)stream>>>(table_dev, n_entries);
    YP_CHECK_HIP(hipGetLastError());
    return YP_OK;
}

// backward of MaxPool2d(2, 2): the gradient of an output pixel goes to the FIRST maximum of its 2x2 window (row-major scan, as
// ATen); windows do not overlap, so every input element is written exactly once (dx (+)= dy or 0).
template <int DT>
__global__ void maxpool2_bwd_kernel(const char* __restrict__ x, int xcs, int xco, const char* __restrict__ dy, int dcs, int dco, char* __restrict__ dx,
                                    int gcs, int gco, int B, int Ho, int Wo, int C, int accumulate) {
    const int chunks = C / 8;
    const size_t n = (size_t)B * Ho * Wo * chunks;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int ch = (int)(i % chunks);
        const size_t pix = i / chunks;
        const int w = (int)(pix % Wo), h = (int)((pix / Wo) % Ho);
        const size_t b = pix / ((size_t)Wo * Ho);
        float g[8], v[4][8];
        load8<DT>(dy, pix * dcs + dco + ch * 8, g);
        size_t ip[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            ip[q] = (b * 2 * Ho + 2 * h + (q >> 1)) * 2 * Wo + 2 * w + (q & 1);
            load8<DT>(x, ip[q] * xcs + xco + ch * 8, v[q]);
        }
        int arg[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int a = 0;
            float best = v[0][j];
#pragma unroll
            for (int q = 1; q < 4; ++q)
                if (v[q][j] > best) { best = v[q][j]; a = q; }
            arg[j] = a;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float o[8];
            if (accumulate) load8<DT>(dx, ip[q] * gcs + gco + ch * 8, o);
            else {
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = 0.f;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] += arg[j] == q ? g[j] : 0.f;
            store8<DT>(dx, ip[q] * gcs + gco + ch * 8, o);
        }
    }
}

extern "C" int yp_maxpool2_bwd(YpView x, YpView dy, YpView dx, int dtype, int B, int accumulate, void* stream) {
    if (int rc = check_view8(x, "yp_maxpool2_bwd")) return rc;
    if (int rc = check_view8(dy, "yp_maxpool2_bwd")) return rc;
    if (int rc = check_view8(dx, "yp_maxpool2_bwd")) return rc;
    YP_REQUIRE(x.C == dy.C && x.C == dx.C && x.H == 2 * dy.H && x.W == 2 * dy.W && dx.H == x.H && dx.W == x.W, "yp_maxpool2_bwd: dims mismatch");
    const size_t n = (size_t)B * dy.H * dy.W * (x.C / 8);
    YP_DT_SWITCH(dtype, (maxpool2_bwd_kernel<DT><<<grid_for(n, 256), 256, 0, (hipStream_t)stream>>>((const char*)x.ptr, x.cstride, x.coff, (const char*)dy.ptr, dy.cstride,
                                                                                                   dy.coff, (char*)dx.ptr, dx.cstride, dx.coff, B, dy.H, dy.W, x.C, accumulate)));
    YP_CHECK_HIP(hipGetLastError());
    return YP_OK;
}

__global__ void zero_fill_kernel(f32x4* __restrict__ p, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) p[i] = f32x4{0.f, 0.f, 0.f, 0.f};
}

namespace {
__global__ __launch_bounds__(256) void sum_slabs_kernel(const float* __restrict__ slabs, float* __restrict__ dst, size_t elems, int n, size_t stride,
                                                        size_t group_stride) {
    // blockIdx.y = slab group: sums its n slabs (`stride` floats apart) in order into dst + blockIdx.y * group_stride
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= elems) return;
    slabs += blockIdx.y * group_stride;
    dst += blockIdx.y * group_stride;
    float4 v = *reinterpret_cast<const float4*>(slabs + i);
    for (int s = 1; s < n; ++s) {
        const float4 u = *reinterpret_cast<const float4*>(slabs + (size_t)s * stride + i);
        v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
    }
    *reinterpret_cast<float4*>(dst + i) = v;
}
}  // namespace

extern "C" int yp_sum_slabs(const float* slabs, float* dst, size_t elems, int n_slabs, void* stream) {
    YP_REQUIRE(slabs && dst && elems > 0 && elems % 4 == 0 && n_slabs > 0 && ((uintptr_t)slabs & 15) == 0 && ((uintptr_t)dst & 15) == 0, "yp_sum_slabs: bad arguments");
    const unsigned gx = (unsigned)((elems / 4 + 255) / 256);
    sum_slabs_kernel<<<gx, 256, 0, (hipStream_t)stream>>>(slabs, dst, elems, n_slabs, elems, 0);
    YP_CHECK_HIP(hipGetLastError());
    return YP_OK;
}

/* The same sum as a fixed two-level tree for many small slabs (few elements, hundreds of slabs: one level leaves the chip to a handful of
 * workgroups): groups of `group` consecutive slabs are summed in order INTO the group's first slab (the slabs are scratch), then the group
 * sums in order into dst.  n_slabs % group == 0.  Bit-reproducible: the tree is fixed. */
extern "C" int yp_sum_slabs_tree(float* slabs, float* dst, size_t elems, int n_slabs, int group, void* stream) {
    YP_REQUIRE(slabs && dst && elems > 0 && elems % 4 == 0 && n_slabs > 0 && group > 0 && n_slabs % group == 0 && ((uintptr_t)slabs & 15) == 0 && ((uintptr_t)dst & 15) == 0,
               "yp_sum_slabs_tree: bad arguments");
    const unsigned gx = (unsigned)((elems / 4 + 255) / 256);
    hipStream_t st = (hipStream_t)stream;
    sum_slabs_kernel<<<dim3(gx, n_slabs / group), 256, 0, st>>>(slabs, slabs, elems, group, elems, (size_t)group * elems);
    sum_slabs_kernel<<<gx, 256, 0, st>>>(slabs, dst, elems, n_slabs / group, (size_t)group * elems, 0);
    YP_CHECK_HIP(hipGetLastError());
    return YP_OK;
}

// Adam over one flat fp32 range: torch.optim.Adam's update (reference train.py:88,252: Adam(lr), amsgrad off, maximize off) --
//   g' = g + wd*p;  m = b1*m + (1-b1)*g';  v = b2*v + (1-b2)*g'^2;  p -= (lr / (1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
// one pass over the four arrays instead of a multi-tensor launch per 30-odd parameters (6 launches, 380 us for 7.6 M parameters).
namespace {
__global__ __launch_bounds__(256) void adam_flat_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, size_t n4,
                                                        float omb1, float b2, float omb2, float eps, float wd, float step_size, float bc2_sqrt) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float4 pp = reinterpret_cast<float4*>(p)[i], mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
        const float4 gg = reinterpret_cast<const float4*>(g)[i];
        float* pa = reinterpret_cast<float*>(&pp);
        float* ma = reinterpret_cast<float*>(&mm);
        float* va = reinterpret_cast<float*>(&vv);
        const float* ga = reinterpret_cast<const float*>(&gg);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float gr = wd != 0.f ? ga[k] + wd * pa[k] : ga[k];
            ma[k] = ma[k] + omb1 * (gr - ma[k]);                         // (torch: exp_avg.lerp_(grad, 1 - beta1); 1 - beta in double on the host)
            va[k] = b2 * va[k] + omb2 * gr * gr;
            const float denom = sqrtf(va[k]) / bc2_sqrt + eps;
            pa[k] -= step_size * (ma[k] / denom);
        }
        reinterpret_cast<float4*>(p)[i] = pp; reinterpret_cast<float4*>(m)[i] = mm; reinterpret_cast<float4*>(v)[i] = vv;
    }
}
}  // namespace

extern "C" int yp_adam_step(float* p, const float* g, float* m, float* v, size_t n, double lr, double beta1, double beta2, double eps, double weight_decay, int step,
                            void* stream) {
    YP_REQUIRE(p && g && m && v && n > 0 && n % 4 == 0 && step >= 1, "yp_adam_step: bad arguments (n must be a multiple of 4)");
    YP_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "yp_adam_step: arrays must be 16-byte aligned");
    const double bc1 = 1.0 - pow(beta1, step), bc2 = 1.0 - pow(beta2, step);
    const size_t n4 = n / 4;
    size_t grid = (n4 + 255) / 256;
    if (grid > 256 * 16) grid = 256 * 16;
    adam_flat_kernel<<<(unsigned)grid, 256, 0, (hipStream_t)stream>>>(p, g, m, v, n4, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, (float)weight_decay,
                                                                          (float)(lr / bc1), (float)sqrt(bc2));
    YP_CHECK_HIP(hipGetLastError());
    return YP_OK;
}

extern "C" int yp_run_op(const YpOpArgs* a, void* stream) {
    YP_REQUIRE(a != nullptr, "yp_run_op: null args");
    const int dt = a->i[0], B = a->i[1];
    switch (a->op) {
        case YP_OP_BN_STATS:
            if (a->i[2] > 0)
                return yp_bn_finalize_grouped((const float*)a->p[1], a->i[2], a->i[3] > 0 ? a->i[3] : 1, a->v[0].C, (double)B * a->v[0].H * a->v[0].W, a->s[0], a->s[1],
                                              a->g[0], a->g[1], a->g[2], a->g[3], stream);
            return yp_bn_stats_grouped(a->v[0], dt, B, a->i[3] > 0 ? a->i[3] : 1, a->s[0], a->s[1], a->g[0], a->g[1], a->g[2], a->g[3], a->p[0], a->n[0], stream);
        case YP_OP_BN_APPLY:
            return yp_bn_act_apply_grouped_q8(a->v[0], a->v[1], a->v[2], dt, B, a->i[3] > 0 ? a->i[3] : 1, a->f[0], a->f[1], a->f[2], a->f[3], a->i[2], a->v[3],
                                              a->g[2], a->g[3], stream);
        case YP_OP_BN_BWD: {
            // p1 != NULL: the shortcut gradient [B, H, W, .] at p1 with channel stride i5 / offset i6 receives (i7 = 1: accumulates) dy in the same pass
            YpView gs{};
            if (a->p[1] != nullptr) { gs = a->v[1]; gs.ptr = a->p[1]; gs.cstride = a->i[5]; gs.coff = a->i[6]; }
            YP_REQUIRE(a->p[1] == nullptr || (a->i[5] % 8 == 0 && a->i[6] % 8 == 0 && a->i[5] >= a->i[6] + a->v[1].C), "YP_OP_BN_BWD: bad shortcut-gradient view");
            return bn_act_bwd_impl(a->v[0], a->v[1], a->v[2], dt, B, a->i[4] > 0 ? a->i[4] : 1, a->f[0], a->f[1], a->f[2], a->f[3], a->i[2], a->g[0], a->g[1],
                                   a->i[3], a->p[0], a->n[0], a->v[3], a->g[2], a->g[3], gs, a->i[7], stream);
        }
        case YP_OP_UPS2_BWD: return yp_ups2_bwd(a->v[0], a->v[1], dt, B, a->i[2], stream);
        case YP_OP_ADD_VIEWS: return yp_add_views(a->v[0], a->v[1], dt, B, a->i[2], stream);
        case YP_OP_MAXPOOL5_BWD: return yp_maxpool5_bwd(a->v[0], a->v[1], a->v[2], dt, B, a->i[2], a->p[0], a->n[0], stream);
        case YP_OP_L2NORM_BWD: return yp_l2norm_bwd_f32(a->v[0], a->v[1], a->v[2], B, a->i[2], stream);
        case YP_OP_DETECT_BWD_PACK: return yp_detect_bwd_pack(a->f[0], B, a->i[2], a->i[3], a->v[0], dt, a->f[1], stream);
        case YP_OP_TO_CHWB: return yp_to_chwb(a->v[0], dt, B, a->i[2], a->p[0], a->i[3], stream);
        case YP_OP_COL_SUM: return yp_col_sum(a->v[0], dt, B, a->g[0], a->i[2], a->p[0], a->n[0], stream);
        case YP_OP_MEMSET0: {
            // a zero-fill KERNEL, not hipMemsetAsync: captured into a hipGraph, the memset node did not stay ordered with the kernel
            // nodes around it when backward graphs were replayed back to back (weight gradients came out inf / huge; tools/probe)
            YP_REQUIRE(a->p[0] != nullptr && a->n[0] % 16 == 0 && ((size_t)a->p[0]) % 16 == 0, "YP_OP_MEMSET0: 16-byte aligned pointer and size");
            const size_t n16 = a->n[0] / 16;
            zero_fill_kernel<<<grid_for(n16, 256, 2048), 256, 0, (hipStream_t)stream>>>((f32x4*)a->p[0], n16);
            YP_CHECK_HIP(hipGetLastError());
            return YP_OK;
        }
        case YP_OP_PACK_NCHW: return yp_pack_input(a->f[0], B, a->i[2], a->v[0].H, a->v[0].W, a->v[0], dt, stream);
        case YP_OP_L2NORM: return yp_l2norm_f32(a->v[0], a->v[1], B, a->i[2], stream);
        case YP_OP_SPPF_POOL: return yp_sppf_pool(a->v[0], a->v[1], a->v[2], a->v[3], B, dt, stream);
        case YP_OP_CAST_F32: return yp_cast_from_f32(a->v[0], a->v[1], dt, B, stream);
        case YP_OP_MAXPOOL2: return yp_maxpool2(a->v[0], a->v[1], B, dt, stream);
        case YP_OP_MAXPOOL2_BWD: return yp_maxpool2_bwd(a->v[0], a->v[1], a->v[2], dt, B, a->i[2], stream);
        case YP_OP_WGRAD_UNPACK: return yp_wgrad_unpack((const float*)a->p[0], a->g[0], a->i[1], a->i[2], a->i[3], a->i[4], a->i[5], a->i[6], stream);
        case YP_OP_WGRAD_UNPACK_BATCH: return yp_wgrad_unpack_batch((const YpUnpackEntry*)a->p[0], a->i[1], a->i[2], stream);
        case YP_OP_QUANT_FP8: return yp_quantize_fp8(a->v[0], a->v[1], dt, B, a->i[2], (const float*)a->p[0], (float*)a->p[1], stream);
        case YP_OP_STEM_WGRAD: return yp_stem_wgrad(a->v[0], a->v[1], dt, B, (float*)a->p[0], (float*)a->p[1], stream);
        case YP_OP_SUM_SLABS: return yp_sum_slabs((const float*)a->p[0], (float*)a->p[1], a->n[0], (int)a->n[1], stream);
        case YP_OP_WGRAD_GROUP: return yp_wgrad_group_run_det(a->p[0], a->i[1], a->i[2], a->i[5], dt, a->i[3], a->i[4], a->i[6] > 0 ? a->i[6] : 64, stream);
        case YP_OP_WGRAD: return yp_conv_wgrad(a->v[0], a->v[1], dt, B, a->i[2], a->i[3] > 0 ? a->i[3] : 1, (float*)a->p[0], stream);
        case YP_OP_PACK_WEIGHT:
            return yp_pack_weight(a->f[0], a->i[1], a->i[2], a->i[3], a->i[4], a->i[5], a->i[6], a->i[7], (int)(a->n[1] >> 32), a->p[0], (int)a->n[0],
                                  (int)(a->n[1] & 0xffffffffu), dt, a->f[1], a->g[0], stream);
    }
    yp_set_error("yp_run_op: unknown opcode %d", a->op);
    return YP_ERR_INVALID;
}
