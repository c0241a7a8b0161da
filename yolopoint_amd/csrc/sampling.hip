// The label-only half of the InfoNCE descriptor loss (reference utils/loss_functions.py:484-552): which cells of the image are matched
// to which cells of its warp, which matches serve as negatives, and the two index structures the atomic-free backward kernels walk.
// The reference does this with ~60 small framework launches (warp the validity mask, reshape it into cells, warp the cell grid, shuffle,
// randint, ...); here it is a handful of kernels around a counter-based generator (Philox 4x32-10), on the device, with the draws of the
// reference's distributions:
//   nce_cells_kernel      per cell: valid <=> all 64 pixels of the cell land, through the inverse homography (nearest, zero padding,
//                         align_corners = True), on mask pixels equal to 1;  uv_b = round(cell centre through the cell-grid homography)
//   nce_select_kernel     per image: `pool` = min(samples, min over images of #valid) cells drawn uniformly without replacement from the
//                         valid ones (the cells with the smallest random keys; listed in cell order), written as normalised sample
//                         coordinates ua | ub
//   nce_negatives_*       rnd[i][j] uniform in [0, n); a draw equal to its own row is replaced by floor(U * #such draws) -- the reference's
//                         redraw from [0, #collisions)
//   csr_*                 (key, item) pairs -> items grouped by key, ascending inside a group, + CSR offsets: counting sort (one integer
//                         atomic per item gives the bucket sizes and an arrival rank; a three-step scan gives the offsets; then every
//                         bucket is put in ascending order, so the result does not depend on the order of arrival)
#include "yp_internal.h"

namespace {

struct U4 { unsigned x, y, z, w; };

__device__ __forceinline__ U4 philox4x32(unsigned long long seed, unsigned long long ctr, unsigned stream) {
    unsigned c0 = (unsigned)ctr, c1 = (unsigned)(ctr >> 32), c2 = stream, c3 = 0u;
    unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return U4{c0, c1, c2, c3};
}

__device__ __forceinline__ float lin11(int i, int n) {      // torch.linspace(-1, 1, n)[i]
    const float step = 2.0f / (float)(n - 1);
    return i < n / 2 ? -1.0f + step * (float)i : 1.0f - step * (float)(n - 1 - i);
}

__global__ __launch_bounds__(256) void nce_cells_kernel(const float* __restrict__ mask, const float* __restrict__ inv_h, int B, int H, int W,
                                                        unsigned char* __restrict__ valid, float* __restrict__ uvb) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int Hc = H >> 3, Wc = W >> 3;
    for (long cell = (long)blockIdx.x * 4 + wave; cell < (long)B * Hc * Wc; cell += (long)gridDim.x * 4) {
    const int b = (int)(cell / (Hc * Wc)), rem = (int)(cell - (long)b * Hc * Wc), cy = rem / Wc, cx = rem - cy * Wc;
    const float* h = inv_h + 9 * b;
    const int y = cy * 8 + (lane >> 3), x = cx * 8 + (lane & 7);
    const float xn = lin11(x, W), yn = lin11(y, H);
    const float w0 = h[0] * xn + h[1] * yn + h[2], w1 = h[3] * xn + h[4] * yn + h[5], w2 = h[6] * xn + h[7] * yn + h[8];
    const float u = w0 / w2, v = w1 / w2;
    const float rx = nearbyintf(((u + 1.0f) / 2.0f) * (float)(W - 1)), ry = nearbyintf(((v + 1.0f) / 2.0f) * (float)(H - 1));
    const bool inb = rx >= 0.0f && rx < (float)W && ry >= 0.0f && ry < (float)H;
    const float m = inb ? mask[((size_t)b * H + (int)ry) * W + (int)rx] : 0.0f;
    const bool all1 = __ballot(m == 1.0f) == ~0ull;
    if (lane == 0) {
        valid[cell] = all1 ? 1 : 0;
        // cell grid homography: inverse(T) @ inv_h @ T with T the pixel -> [-1, 1] map of a Wc x Hc grid (utils/utils.py:333-345)
        const float sx = 2.0f / (float)Wc, sy = 2.0f / (float)Hc, hx = 0.5f * (float)Wc, hy = 0.5f * (float)Hc;
        float M[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r) { M[r][0] = h[3 * r] * sx; M[r][1] = h[3 * r + 1] * sy; M[r][2] = h[3 * r + 2] - h[3 * r] - h[3 * r + 1]; }
        const float px = (float)cx, py = (float)cy;
        const float a0 = (hx * M[0][0] + hx * M[2][0]) * px + (hx * M[0][1] + hx * M[2][1]) * py + (hx * M[0][2] + hx * M[2][2]);
        const float a1 = (hy * M[1][0] + hy * M[2][0]) * px + (hy * M[1][1] + hy * M[2][1]) * py + (hy * M[1][2] + hy * M[2][2]);
        const float a2 = M[2][0] * px + M[2][1] * py + M[2][2];
        uvb[2 * cell] = nearbyintf(a0 / a2);
        uvb[2 * cell + 1] = nearbyintf(a1 / a2);
    }
    }
}

// 1024 threads: sum of one int per thread
__device__ __forceinline__ int block_sum_1024(int v, int* sh) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();                       // (sh may still be read from the previous call)
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    int t = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += sh[i];
    return t;
}

__global__ __launch_bounds__(1024) void nce_select_kernel(const unsigned char* __restrict__ valid, const float* __restrict__ uvb, int B, int Hc, int Wc,
                                                          int samples, unsigned long long seed, float* __restrict__ uab, int* __restrict__ meta) {
    extern __shared__ unsigned keys[];                     // [Nc] random key of every cell of this image
    __shared__ int sh[16];
    __shared__ int sh_cnt[1024];
    const int Nc = Hc * Wc, b = blockIdx.x, t = threadIdx.x;
    // ---- pool = min(samples, min over images of #valid cells): every workgroup counts all images (B * Nc bytes: L2 resident)
    for (int j = t; j < B; j += 1024) sh_cnt[j] = 0;
    __syncthreads();
    for (int j = 0; j < B; ++j) {
        int c = 0;
        for (int i = t; i < Nc; i += 1024) c += valid[(size_t)j * Nc + i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
        if ((t & 63) == 0 && c) atomicAdd(&sh_cnt[j], c);
    }
    __syncthreads();
    int pool = samples;
    for (int j = 0; j < B; ++j) pool = min(pool, sh_cnt[j]);
    if (b == 0 && t == 0) { meta[0] = pool; meta[1] = B * pool; meta[2] = 0; }
    if (pool <= 0) return;
    const unsigned char* vb = valid + (size_t)b * Nc;
    for (int i = t; i < Nc; i += 1024) keys[i] = philox4x32(seed, (unsigned long long)b * Nc + i, 0u).x;
    __syncthreads();
    // ---- the pool-th smallest (key, cell) pair of the valid cells: binary search on the 48-bit value key << 16 | cell (pairs are distinct)
    unsigned long long lo = 0ull, hi = (1ull << 48) - 1ull;
    while (lo < hi) {
        const unsigned long long mid = lo + ((hi - lo) >> 1);
        int c = 0;
        for (int i = t; i < Nc; i += 1024) c += (vb[i] && (((unsigned long long)keys[i] << 16) | (unsigned)i) <= mid) ? 1 : 0;
        c = block_sum_1024(c, sh);
        if (c >= pool) hi = mid; else lo = mid + 1ull;
    }
    const unsigned long long T = lo;
    // ---- the selected cells in cell order -> normalised coordinates (normPts: p / size * 2 - 1)
    int base = 0;
    const float fw = (float)Wc, fh = (float)Hc;
    for (int i0 = 0; i0 < Nc; i0 += 1024) {
        const int i = i0 + t;
        const int f = (i < Nc && vb[i] && (((unsigned long long)keys[i] << 16) | (unsigned)i) <= T) ? 1 : 0;
        int incl = f;                                       // inclusive scan over the workgroup
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(incl, o, 64); if ((t & 63) >= o) incl += u; }
        __syncthreads();
        if ((t & 63) == 63) sh[t >> 6] = incl;
        __syncthreads();
        int before = 0, total = 0;
#pragma unroll
        for (int wv = 0; wv < 16; ++wv) { const int s_ = sh[wv]; if (wv < (t >> 6)) before += s_; total += s_; }
        if (f) {
            const int pos = base + before + incl - 1;
            const float ax = (float)(i % Wc), ay = (float)(i / Wc);
            const float bx = uvb[2 * ((size_t)b * Nc + i)], by = uvb[2 * ((size_t)b * Nc + i) + 1];
            float* pa = uab + 2 * ((size_t)b * pool + pos);
            float* pb = uab + 2 * ((size_t)(B + b) * pool + pos);
            pa[0] = ax / fw * 2.0f - 1.0f; pa[1] = ay / fh * 2.0f - 1.0f;
            pb[0] = bx / fw * 2.0f - 1.0f; pb[1] = by / fh * 2.0f - 1.0f;
        }
        base += total;
    }
}

__device__ __forceinline__ int draw_row(unsigned r, int n) { return (int)(((unsigned long long)r * (unsigned)n) >> 32); }

// draws equal to their own row, counted (integer atomics: one per workgroup)
__global__ __launch_bounds__(256) void nce_negatives_count_kernel(int n, int negs, unsigned long long seed, int* __restrict__ meta, int n_from_meta) {
    __shared__ int sh[4];
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;      // four draws per thread: one Philox call
    if (n_from_meta) n = meta[1];                                  // (n = B * pool as nce_select_kernel left it; the launch covers the capacity)
    const size_t total = (size_t)n * negs;
    int c = 0;
    if (q * 4 < total) {
        const U4 r = philox4x32(seed, q, 1u);
        const unsigned rr[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t e = q * 4 + u;
            if (e < total) c += draw_row(rr[u], n) == (int)(e / negs) ? 1 : 0;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) { const int s_ = sh[0] + sh[1] + sh[2] + sh[3]; if (s_) atomicAdd(&meta[2], s_); }
}

// idx[i][0] = i, idx[i][1 + j] = the j-th negative of match i
__global__ __launch_bounds__(256) void nce_negatives_write_kernel(int n, int negs, unsigned long long seed, const int* __restrict__ meta, int* __restrict__ idx,
                                                                  int n_from_meta) {
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int E = negs + 1;
    if (n_from_meta) {                                             // rows [n, capacity): no edges (keys the counting sort skips)
        const size_t cap_total = (size_t)n * negs;
        const int nn = meta[1];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t e = q * 4 + u;
            if (e < cap_total && e >= (size_t)nn * negs) {
                const int i = (int)(e / negs), j = (int)(e - (size_t)i * negs);
                idx[(size_t)i * E + 1 + j] = 0x7fffffff;
                if (j == 0) idx[(size_t)i * E] = 0x7fffffff;
            }
        }
        n = nn;
    }
    const size_t total = (size_t)n * negs;
    if (q * 4 >= total) return;
    const U4 r = philox4x32(seed, q, 1u);
    const unsigned rr[4] = {r.x, r.y, r.z, r.w};
    const float same = (float)meta[2];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const size_t e = q * 4 + u;
        if (e >= total) break;
        const int i = (int)(e / negs), j = (int)(e - (size_t)i * negs);
        int v = draw_row(rr[u], n);
        if (v == i) v = (int)((float)(philox4x32(seed, e, 2u).x >> 8) * (1.0f / 16777216.0f) * same);
        idx[(size_t)i * E + 1 + j] = v;
        if (j == 0) idx[(size_t)i * E] = i;
    }
}

// ---- counting sort into CSR
__global__ __launch_bounds__(256) void csr_zero_kernel(int* __restrict__ p, int n) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) p[i] = 0;
}

// one atomic per item: its arrival rank inside its bucket (the bucket sizes are what the counters end at)
__global__ __launch_bounds__(256) void csr_rank_kernel(const int* __restrict__ keys, int n_items, int n_buckets, int* __restrict__ count, int* __restrict__ rank) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n_items; i += gridDim.x * 256) {
        const int k = keys[i];
        if (k >= 0 && k < n_buckets) rank[i] = atomicAdd(&count[k], 1);
    }
}

// exclusive scan of the bucket sizes in three steps: sums of 4096-bucket chunks, scan of the chunk sums (one workgroup), scan inside the chunks
constexpr int SCAN_CHUNK = 4096;

__global__ __launch_bounds__(256) void csr_chunk_sum_kernel(const int* __restrict__ count, int n_buckets, int* __restrict__ chunk_sum) {
    __shared__ int sh[4];
    const int base = blockIdx.x * SCAN_CHUNK;
    int s = 0;
    for (int i = threadIdx.x; i < SCAN_CHUNK; i += 256) s += base + i < n_buckets ? count[base + i] : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) chunk_sum[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

__global__ __launch_bounds__(1024) void csr_chunk_scan_kernel(int* __restrict__ chunk_sum, int n_chunks, int* __restrict__ total_out) {
    __shared__ int sh[16];
    const int t = threadIdx.x;
    int run = 0;
    for (int i0 = 0; i0 < n_chunks; i0 += 1024) {          // (one trip up to 4 M buckets)
        const int v = i0 + t < n_chunks ? chunk_sum[i0 + t] : 0;
        int incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(incl, o, 64); if ((t & 63) >= o) incl += u; }
        __syncthreads();
        if ((t & 63) == 63) sh[t >> 6] = incl;
        __syncthreads();
        int before = 0, total = 0;
#pragma unroll
        for (int wv = 0; wv < 16; ++wv) { const int s_ = sh[wv]; if (wv < (t >> 6)) before += s_; total += s_; }
        if (i0 + t < n_chunks) chunk_sum[i0 + t] = run + before + incl - v;
        run += total;
    }
    if (t == 0) *total_out = run;
}

// offsets[b] for the buckets of one chunk (thread t owns buckets [16 t, 16 t + 16) of it)
__global__ __launch_bounds__(256) void csr_offsets_kernel(const int* __restrict__ count, const int* __restrict__ chunk_base, int n_buckets, int* __restrict__ offsets) {
    __shared__ int sh[4];
    const int t = threadIdx.x, base = blockIdx.x * SCAN_CHUNK + t * 16;
    int c[16], s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) { c[i] = base + i < n_buckets ? count[base + i] : 0; s += c[i]; }
    int incl = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(incl, o, 64); if ((t & 63) >= o) incl += u; }
    if ((t & 63) == 63) sh[t >> 6] = incl;
    __syncthreads();
    int before = 0;
#pragma unroll
    for (int wv = 0; wv < 4; ++wv) if (wv < (t >> 6)) before += sh[wv];
    int run = chunk_base[blockIdx.x] + before + incl - s;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        if (base + i < n_buckets) offsets[base + i] = run;
        run += c[i];
    }
}

__global__ __launch_bounds__(256) void csr_scatter_kernel(const int* __restrict__ keys, const int* __restrict__ rank, int n_items, int n_buckets,
                                                          const int* __restrict__ offsets, int* __restrict__ order) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n_items; i += gridDim.x * 256) {
        const int k = keys[i];
        if (k >= 0 && k < n_buckets) order[offsets[k] + rank[i]] = i;
    }
}

// Bitonic sort of 256 ints held four per lane (element e = 4 * lane + r), ascending over e.  Strides 1 and 2 are exchanges between a lane's own
// registers, strides >= 4 one cross-lane read (__shfl_xor) + min / max / select per element: ~520 instructions for the 36 stages, against ~2 400 for
// the rank sort of a 200-entry bucket (every lane compares each of its entries with all the others through LDS).
__device__ __forceinline__ void wave_bitonic256(int (&v)[4], int lane) {
    auto cx = [](int& x, int& y, bool asc) {              // in-lane compare-exchange: (x, y) ascending or descending
        const int mn = x < y ? x : y, mx = x < y ? y : x;
        x = asc ? mn : mx;
        y = asc ? mx : mn;
    };
#pragma unroll
    for (int size = 2; size <= 256; size <<= 1) {
#pragma unroll
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            if (stride >= 4) {
                const int ls = stride >> 2;
                const bool keep_min = ((lane & ls) == 0) == (((4 * lane) & size) == 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int o = __shfl_xor(v[r], ls, 64);
                    const int mn = v[r] < o ? v[r] : o, mx = v[r] < o ? o : v[r];
                    v[r] = keep_min ? mn : mx;
                }
            } else if (stride == 2) {
                const bool asc = ((4 * lane) & size) == 0;            // (size >= 4 here: the direction bit lies in the lane index)
                cx(v[0], v[2], asc);
                cx(v[1], v[3], asc);
            } else {
                if (size == 2) { cx(v[0], v[1], true); cx(v[2], v[3], false); }      // direction bit = bit 1 of r
                else {
                    const bool asc = ((4 * lane) & size) == 0;
                    cx(v[0], v[1], asc);
                    cx(v[2], v[3], asc);
                }
            }
        }
    }
}

// every bucket into ascending item order.  WAVE: one wavefront per bucket (<= 256 entries: the bitonic sort above, in registers; more: rank sort
// through LDS), else one thread per bucket (insertion sort in place; buckets of a few entries)
template <bool WAVE>
__global__ __launch_bounds__(256) void csr_sort_kernel(const int* __restrict__ offsets, int n_buckets, int* __restrict__ order) {
    constexpr int CAP = 2048;
    __shared__ int stage[WAVE ? 4 * CAP : 1];
    if constexpr (WAVE) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        for (int bk = blockIdx.x * 4 + wave; bk < n_buckets; bk += gridDim.x * 4) {
        const int e0 = offsets[bk], k = offsets[bk + 1] - e0;
        if (k <= 1) continue;
        if (k <= 256) {
            // (the buckets of the InfoNCE edge list: ~E = 200 entries.  The rank sort below took 558 us per -s step -- the longest launch of the
            // label stream, in front of the InfoNCE chain the trunk backward waits for -- 4.2 ms at 64 samples)
            int v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = 4 * lane + r < k ? order[e0 + 4 * lane + r] : 0x7fffffff;
            wave_bitonic256(v, lane);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (4 * lane + r < k) order[e0 + 4 * lane + r] = v[r];
            continue;
        }
        if (k <= CAP) {
            int* st = stage + wave * CAP;
            for (int i = lane; i < k; i += 64) st[i] = order[e0 + i];
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_s_waitcnt(0xc07f);            // lgkmcnt(0): the LDS writes of this wave are visible to its other lanes
            for (int i = lane; i < k; i += 64) {
                const int a = st[i];
                int rank = 0;
                for (int j = 0; j < k; ++j) rank += st[j] < a ? 1 : 0;
                order[e0 + rank] = a;                      // (item ids are distinct: ranks are a permutation)
            }
            __builtin_amdgcn_wave_barrier();               // (the next bucket of this wave overwrites its LDS stage)
            continue;
        }
        if (lane != 0) continue;
        for (int i = 1; i < k; ++i) {                      // (oversized bucket: correct, slow, not expected)
            const int a = order[e0 + i];
            int j = i - 1;
            while (j >= 0 && order[e0 + j] > a) { order[e0 + j + 1] = order[e0 + j]; --j; }
            order[e0 + j + 1] = a;
        }
        }
    } else {
        const int bk = blockIdx.x * 256 + threadIdx.x;
        if (bk >= n_buckets) return;
        const int e0 = offsets[bk], k = offsets[bk + 1] - e0;
        for (int i = 1; i < k; ++i) {
            const int a = order[e0 + i];
            int j = i - 1;
            while (j >= 0 && order[e0 + j] > a) { order[e0 + j + 1] = order[e0 + j]; --j; }
            order[e0 + j + 1] = a;
        }
    }
}

}  // namespace

// Grid cap of the large launches in this file (yp_sampling_set_max_workgroups; 0 = none): the label work of a training step runs on a side
// stream beside the forward pass, and thousands of tiny workgroups would take the CU slots the convolutions wait for.
// (a process-wide tuning knob -- results never depend on it; atomic so that a setter on one thread and launches on another do not race)
static std::atomic<int> g_max_wgs{0};
static unsigned capped(size_t want) { const int cap = g_max_wgs.load(std::memory_order_relaxed); return (unsigned)(cap > 0 && want > (size_t)cap ? (size_t)cap : want); }
extern "C" int yp_sampling_set_max_workgroups(int n) {
    YP_REQUIRE(n >= 0, "yp_sampling_set_max_workgroups: n >= 0");
    g_max_wgs.store(n, std::memory_order_relaxed);
    return YP_OK;
}

extern "C" int yp_nce_cells(const float* mask, const float* inv_h, int B, int H, int W, unsigned char* valid, float* uvb, void* stream) {
    YP_REQUIRE(mask && inv_h && valid && uvb && B > 0 && H >= 16 && W >= 16 && H % 8 == 0 && W % 8 == 0, "yp_nce_cells: bad arguments (H, W multiples of 8)");
    const long cells = (long)B * (H / 8) * (W / 8);
    nce_cells_kernel<<<capped((size_t)(cells + 3) / 4), 256, 0, (hipStream_t)stream>>>(mask, inv_h, B, H, W, valid, uvb);
    YP_CHECK_HIP(hipGetLastError());
    return YP_OK;
}

extern "C" int yp_nce_select(const unsigned char* valid, const float* uvb, int B, int Hc, int Wc, int samples, uint64_t seed, float* uab, int* meta, void* stream) {
    YP_REQUIRE(valid && uvb && uab && meta && B > 0 && B <= 1024 && Hc > 0 && Wc > 0 && samples > 0, "yp_nce_select: bad arguments (B <= 1024)");
    const size_t lds = (size_t)Hc * Wc * sizeof(unsigned);
    YP_REQUIRE(Hc * Wc < 65536 && lds <= 144 * 1024, "yp_nce_select: %d x %d cells do not fit the workgroup's LDS", Hc, Wc);
    static YpLdsAttr attr;              // per device
    YP_CHECK_HIP(yp_set_max_lds(attr, (const void*)nce_select_kernel, 144 * 1024));
    nce_select_kernel<<<B, 1024, lds, (hipStream_t)stream>>>(valid, uvb, B, Hc, Wc, samples, (unsigned long long)seed, uab, meta);
    YP_CHECK_HIP(hipGetLastError());
    return YP_OK;
}

extern "C" int yp_nce_negatives(int n, int negs, uint64_t seed, int* meta, int* idx, int n_from_meta, void* stream) {
    YP_REQUIRE(meta && idx && n > 0 && negs > 0, "yp_nce_negatives: bad arguments");
    const size_t quads = ((size_t)n * negs + 3) / 4;
    const unsigned grid = (unsigned)((quads + 255) / 256);
    nce_negatives_count_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(n, negs, (unsigned long long)seed, meta, n_from_meta);
    nce_negatives_write_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(n, negs, (unsigned long long)seed, meta, idx, n_from_meta);
    YP_CHECK_HIP(hipGetLastError());
    return YP_OK;
}

extern "C" size_t yp_csr_workspace_ints(int n_items, int n_buckets) { return (size_t)n_items + n_buckets + (n_buckets + 4095) / 4096 + 8; }

extern "C" int yp_csr_build(const int* keys, int n_items, int n_buckets, int wide_buckets, int* order, int* offsets, int* workspace, void* stream) {
    YP_REQUIRE(keys && order && offsets && workspace && n_items > 0 && n_buckets > 0, "yp_csr_build: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const int n_chunks = (n_buckets + SCAN_CHUNK - 1) / SCAN_CHUNK;
    int* count = workspace;                    // [n_buckets]
    int* chunk = count + n_buckets;            // [n_chunks]
    int* rank = chunk + n_chunks;              // [n_items]
    const int zb = (n_buckets + 255) / 256;
    csr_zero_kernel<<<zb < 1024 ? zb : 1024, 256, 0, st>>>(count, n_buckets);
    csr_rank_kernel<<<capped((size_t)(n_items + 255) / 256), 256, 0, st>>>(keys, n_items, n_buckets, count, rank);
    csr_chunk_sum_kernel<<<n_chunks, 256, 0, st>>>(count, n_buckets, chunk);
    csr_chunk_scan_kernel<<<1, 1024, 0, st>>>(chunk, n_chunks, offsets + n_buckets);
    csr_offsets_kernel<<<n_chunks, 256, 0, st>>>(count, chunk, n_buckets, offsets);
    csr_scatter_kernel<<<capped((size_t)(n_items + 255) / 256), 256, 0, st>>>(keys, rank, n_items, n_buckets, offsets, order);
    if (wide_buckets) csr_sort_kernel<true><<<capped((size_t)(n_buckets + 3) / 4), 256, 0, st>>>(offsets, n_buckets, order);
    else csr_sort_kernel<false><<<(n_buckets + 255) / 256, 256, 0, st>>>(offsets, n_buckets, order);
    YP_CHECK_HIP(hipGetLastError());
    return YP_OK;
}
