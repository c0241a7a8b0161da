// Post-processing kernels: keypoint heat-map decode + greedy grid NMS, batched box NMS,
// descriptor sampling and the mutual-nearest-neighbour matcher.
//
// Everything here is index-selection work whose results must equal the sequential reference
// bit for bit (for distinct scores), so this file is compiled with -ffp-contract=off and the
// float arithmetic follows the reference's operation order.
#include "yp_internal.h"
#include <cfloat>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;

namespace {

inline int grid_for(size_t n, int block, size_t cap = 256 * 16) {
    size_t g = (n + block - 1) / block;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

// =========================================================================================
// keypoint decode: softmax(65) -> drop dustbin -> depth-to-space(8)
// A workgroup owns one row of up to 32 cells; each wavefront reduces one cell at a time (lane c
// holds channel c, the dustbin is read by every lane) and parks the 64 probabilities in an LDS
// tile [8][256] that is then written out as whole 1 KiB heat-map rows.
// =========================================================================================
__global__ __launch_bounds__(256) void kp_decode_kernel(const float* __restrict__ semi, int B, int Hc, int Wc, long sb, long sc,
                                                        long sy, long sx, int mode, float* __restrict__ heat) {
    __shared__ float tile[8][32 * 8 + 1];
    const int groups = (Wc + 31) / 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long nwork = (long)B * Hc * groups;
    for (long wk = blockIdx.x; wk < nwork; wk += gridDim.x) {
        const int gx = (int)(wk % groups);
        const int hc = (int)((wk / groups) % Hc);
        const int b = (int)(wk / ((long)groups * Hc));
        const int w0 = gx * 32;
        const int ncell = min(32, Wc - w0);
        for (int ci = wave; ci < ncell; ci += 4) {
            const float* p = semi + b * sb + hc * sy + (long)(w0 + ci) * sx;
            const float x = p[lane * sc];
            const float d = p[64 * sc];
            float e, ed, s;
            if (mode == 0) {
                float m = x;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
                m = fmaxf(m, d);
                e = expf(x - m);
                ed = expf(d - m);
                s = e;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
                s += ed;
            } else {
                e = expf(x);
                ed = expf(d);
                s = e;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
                s += ed;
                s += 0.00001f;
            }
            tile[lane >> 3][ci * 8 + (lane & 7)] = e / s;
        }
        __syncthreads();
        const int W = Wc * 8;
        const int ncol = ncell * 8;
        for (int i = threadIdx.x; i < 8 * ncol; i += 256) {
            const int r = i / ncol, c = i - r * ncol;
            heat[((long)b * Hc * 8 + hc * 8 + r) * W + w0 * 8 + c] = tile[r][c];
        }
        __syncthreads();
    }
}

// =========================================================================================
// keypoint greedy grid NMS.
// The sequential reference visits candidates by descending score; a visited candidate that is
// still alive is kept and kills every other candidate in its (2r+1)^2 window.  Equivalent
// fix-point: p is KEPT once every higher-priority candidate in its window is SUPPRESSED, and
// SUPPRESSED once a higher-priority candidate in its window is KEPT.  State transitions are
// monotone and final, so any asynchronous evaluation order reaches the same fix-point; rounds
// are separate launches, with an early-out on a per-round undecided counter.
// =========================================================================================
enum : unsigned char { KP_EMPTY = 0, KP_UNDECIDED = 1, KP_KEPT = 2, KP_SUPPRESSED = 3 };

// Threshold + candidate compaction.  A returning atomic on ONE counter retires every ~125 ns on this part (10 000 candidates of a
// 1280x1280 map: 1.2 ms), so a workgroup counts the candidates of its pixel span first, reserves its slice of the list with a single
// atomic and then writes it (the span is read twice; the list order is arbitrary, the NMS fix-point and the final rank sort do not
// depend on it).  grid = (KP_SPANS, B).
constexpr int KP_SPANS = 128;
__global__ __launch_bounds__(256) void kp_threshold_kernel(const float* __restrict__ heat, int B, int HW, float thr, unsigned char* __restrict__ state,
                                                           int* __restrict__ cand, int* __restrict__ ncand) {
    __shared__ int wave_cnt[4];
    __shared__ int base_s;
    const int b = blockIdx.y, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int span = (HW + KP_SPANS - 1) / KP_SPANS;
    const int i0 = blockIdx.x * span, i1 = min(i0 + span, HW);
    const float* hb = heat + (long)b * HW;
    unsigned char* sb = state + (long)b * HW;
    int mine = 0;
    for (int i = i0 + t; i < i1; i += 256) {
        const bool c = hb[i] >= thr;
        sb[i] = c ? KP_UNDECIDED : KP_EMPTY;
        mine += c ? 1 : 0;
    }
    // exclusive prefix of `mine` over the workgroup
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o, 64);
        if (lane >= o) incl += v;
    }
    if (lane == 63) wave_cnt[wave] = incl;
    __syncthreads();
    int before = incl - mine;
    for (int w = 0; w < wave; ++w) before += wave_cnt[w];
    if (t == 0) {
        const int total = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        base_s = total ? atomicAdd(&ncand[b], total) : 0;
    }
    __syncthreads();
    if (mine == 0) return;
    int pos = base_s + before;
    int* cb = cand + (long)b * HW;
    for (int i = i0 + t; i < i1; i += 256)
        if (hb[i] >= thr) cb[pos++] = i;
}

__device__ __forceinline__ bool kp_higher(float sq, int q, float sp, int p) { return sq > sp || (sq == sp && q < p); }

// One round: 16 lanes share a candidate's (2r+1)^2 window (a serial walk of 81 dependent byte loads per thread made a round 33 us).
__global__ __launch_bounds__(256) void kp_round_kernel(const float* __restrict__ heat, int B, int H, int W, int radius, volatile unsigned char* state,
                                                       const int* __restrict__ cand, const int* __restrict__ ncand, const int* prev_left, int* left) {
    if (prev_left != nullptr && *prev_left == 0) return;
    const int HW = H * W;
    const int sub = threadIdx.x & 15;
    const int gid = (blockIdx.x * blockDim.x + threadIdx.x) >> 4, ngroups = (gridDim.x * blockDim.x) >> 4;
    const int side = 2 * radius + 1, win = side * side;
    int undecided = 0;
    for (int b = 0; b < B; ++b) {
        const int nc = ncand[b];
        const float* hb = heat + (long)b * HW;
        volatile unsigned char* sb = state + (long)b * HW;
        for (int i = gid; i < nc; i += ngroups) {                  // (uniform over the 16 lanes of a group)
            const int p = cand[(long)b * HW + i];
            if (sb[p] != KP_UNDECIDED) continue;
            const int py = p / W, px = p - py * W;
            const float sp = hb[p];
            bool killed = false, blocked = false;
            for (int k = sub; k < win; k += 16) {
                const int dy = k / side, dx = k - dy * side;
                const int y = py - radius + dy, x = px - radius + dx;
                if (y < 0 || y >= H || x < 0 || x >= W) continue;
                const int q = y * W + x;
                const unsigned char s_ = sb[q];   // a stale UNDECIDED only delays the decision
                if (s_ == KP_EMPTY || s_ == KP_SUPPRESSED || q == p) continue;
                if (!kp_higher(hb[q], q, sp, p)) continue;
                if (s_ == KP_KEPT) killed = true;
                else blocked = true;              // an undecided higher-priority neighbour
            }
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) {
                killed |= (bool)__shfl_xor((int)killed, o, 64);
                blocked |= (bool)__shfl_xor((int)blocked, o, 64);
            }
            if (sub == 0) {
                if (killed) sb[p] = KP_SUPPRESSED;
                else if (!blocked) sb[p] = KP_KEPT;
                else ++undecided;
            }
        }
    }
    if (undecided) atomicAdd(left, undecided);
}

// kept candidates outside the border strip -> compact list (order arbitrary)
__global__ void kp_collect_kernel(const float* __restrict__ heat, int B, int H, int W, int border, const unsigned char* __restrict__ state,
                                  const int* __restrict__ cand, const int* __restrict__ ncand, int* __restrict__ kept, float* __restrict__ kscore,
                                  int* __restrict__ nkept) {
    const int HW = H * W;
    const int lane = threadIdx.x & 63;
    for (int b = 0; b < B; ++b) {
        const int nc = ncand[b];
        const int nc_up = (nc + 63) & ~63;                              // whole waves stay in the loop (ballots below)
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nc_up; i += gridDim.x * blockDim.x) {
            bool keep = false;
            int p = 0;
            if (i < nc) {
                p = cand[(long)b * HW + i];
                if (state[(long)b * HW + p] == KP_KEPT) {
                    const int y = p / W, x = p - y * W;
                    keep = !(x < border || x >= W - border || y < border || y >= H - border);
                }
            }
            const unsigned long long m = __ballot(keep);               // one atomic per wave, not per point
            if (m == 0) continue;
            int base = 0;
            if (lane == __ffsll((long long)m) - 1) base = atomicAdd(&nkept[b], __popcll(m));
            base = __shfl(base, __ffsll((long long)m) - 1, 64);
            if (keep) {
                const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
                kept[(long)b * HW + pos] = p;
                kscore[(long)b * HW + pos] = heat[(long)b * HW + p];
            }
        }
    }
}

// rank-by-counting sort (score desc, index asc) of the kept list; writes (x, y, conf) rows.
// rank of kept point i = number of kept points of higher priority, counted against tiles of the list staged in LDS; the j range is
// split KP_RANK_SPLIT ways across blockIdx.y (partial counts, summed by kp_rank_write_kernel) so that 20 000 points use the whole chip.
constexpr int KP_RANK_SPLIT = 8;
__global__ __launch_bounds__(256) void kp_rank_kernel(int b, int HW, const int* __restrict__ kept, const float* __restrict__ kscore,
                                                      const int* __restrict__ nkept, int* __restrict__ partial) {
    __shared__ int tp[1024];
    __shared__ float ts[1024];
    const int n = nkept[b];
    const int* kb = kept + (long)b * HW;
    const float* sb = kscore + (long)b * HW;
    const int jspan = ((n + (int)gridDim.y - 1) / (int)gridDim.y + 1023) & ~1023;
    const int ja = blockIdx.y * jspan, jb = min(ja + jspan, n);
    for (int i0 = blockIdx.x * 256; i0 < n; i0 += gridDim.x * 256) {     // (uniform per workgroup: barriers inside)
        const int i = i0 + threadIdx.x;
        const bool live = i < n;
        const int p = live ? kb[i] : 0;
        const float sp = live ? sb[i] : 0.f;
        int rank = 0;
        for (int j0 = ja; j0 < jb; j0 += 1024) {
            __syncthreads();
            for (int j = threadIdx.x; j < 1024 && j0 + j < jb; j += 256) { tp[j] = kb[j0 + j]; ts[j] = sb[j0 + j]; }
            __syncthreads();
            const int m = min(1024, jb - j0);
            int j = 0;
            for (; j + 8 <= m; j += 8) {                      // eight independent LDS reads in flight
#pragma unroll
                for (int u = 0; u < 8; ++u) rank += kp_higher(ts[j + u], tp[j + u], sp, p) ? 1 : 0;
            }
            for (; j < m; ++j) rank += kp_higher(ts[j], tp[j], sp, p) ? 1 : 0;
        }
        if (live) partial[(long)blockIdx.y * n + i] = rank;
    }
}

__global__ void kp_rank_write_kernel(int b, int HW, int W, const int* __restrict__ kept, const float* __restrict__ kscore, const int* __restrict__ nkept,
                                     const int* __restrict__ partial, int split, float* __restrict__ out, int* __restrict__ out_count, int max_out) {
    const int n = nkept[b];
    if (blockIdx.x == 0 && threadIdx.x == 0) out_count[b] = min(n, max_out);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        int rank = 0;
        for (int y = 0; y < split; ++y) rank += partial[(long)y * n + i];
        if (rank < max_out) {
            const int p = kept[(long)b * HW + i];
            float* o = out + ((long)b * max_out + rank) * 3;
            const int yy = p / W;
            o[0] = (float)(p - yy * W);
            o[1] = (float)yy;
            o[2] = kscore[(long)b * HW + i];
        }
    }
}

// =========================================================================================
// box NMS
// =========================================================================================
__device__ __forceinline__ u64 box_key(float conf, unsigned id) {
    // ascending u64 order == (conf descending, id ascending); conf > 0 so its bits are monotone
    return ((u64)(0xFFFFFFFFu - __float_as_uint(conf)) << 32) | id;
}

// (block-aggregated like kp_threshold_kernel: a workgroup counts the candidates of its row span, reserves its slice of the key list
// with one atomic and writes it in a second walk over the rows that passed the objectness test.)  grid = (spans, B): 128 row spans per image,
// 512 for a single image (one frame of the frame pipeline: 128 workgroups walked the 34 MB of [1, 100800, 85] at 0.9 TB/s)
static inline int box_spans(int B) { return B >= 4 ? 128 : (B == 1 ? 512 : 256); }
__global__ __launch_bounds__(256) void box_candidates_kernel(const float* __restrict__ pred, int B, int N, int nc, float conf_thres, int multi_label,
                                                             const unsigned* __restrict__ class_mask, u64* __restrict__ keys, int cap, int* __restrict__ count) {
    __shared__ int wave_cnt[4];
    __shared__ int base_s;
    const int no = nc + 5;
    const int b = blockIdx.y, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int span = (N + (int)gridDim.x - 1) / (int)gridDim.x;
    const int r0 = blockIdx.x * span, r1 = min(r0 + span, N);
    const float* pb = pred + (long)b * N * no;
    auto visit = [&](int row, int& pos, bool write) {
        const float* r = pb + (long)row * no;
        const float obj = r[4];
        if (!(obj > conf_thres)) return;
        if (multi_label) {
            for (int j = 0; j < nc; ++j) {
                const float conf = r[5 + j] * obj;
                if (conf > conf_thres && (class_mask == nullptr || ((class_mask[j >> 5] >> (j & 31)) & 1u))) {     // `classes` filter (general_yolo.py:199-200)
                    if (write && pos < cap) keys[(long)b * cap + pos] = box_key(conf, (unsigned)(row * nc + j));
                    ++pos;
                }
            }
        } else {
            float best = r[5] * obj;
            int bj = 0;
            for (int j = 1; j < nc; ++j) {
                const float conf = r[5 + j] * obj;
                if (conf > best) { best = conf; bj = j; }
            }
            if (best > conf_thres && (class_mask == nullptr || ((class_mask[bj >> 5] >> (bj & 31)) & 1u))) {
                if (write && pos < cap) keys[(long)b * cap + pos] = box_key(best, (unsigned)(row * nc + bj));
                ++pos;
            }
        }
    };
    int mine = 0;
    for (int row = r0 + t; row < r1; row += 256) visit(row, mine, false);
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o, 64);
        if (lane >= o) incl += v;
    }
    if (lane == 63) wave_cnt[wave] = incl;
    __syncthreads();
    int before = incl - mine;
    for (int w = 0; w < wave; ++w) before += wave_cnt[w];
    if (t == 0) {
        const int total = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        base_s = total ? atomicAdd(&count[b], total) : 0;
    }
    __syncthreads();
    if (mine == 0) return;
    int pos = base_s + before;
    for (int row = r0 + t; row < r1; row += 256) visit(row, pos, true);
}

struct BoxF { float x1, y1, x2, y2; };

__device__ __forceinline__ bool iou_gt(const BoxF& a, float aarea, const BoxF& b, float barea, float thr) {
    // torchvision.ops.nms CPU definition: inter / (area_a + area_b - inter) > thr
    const float xx1 = fmaxf(a.x1, b.x1), yy1 = fmaxf(a.y1, b.y1);
    const float xx2 = fminf(a.x2, b.x2), yy2 = fminf(a.y2, b.y2);
    const float w = fmaxf(0.f, xx2 - xx1), h = fmaxf(0.f, yy2 - yy1);
    const float inter = w * h;
    const float ovr = inter / (aarea + barea - inter);
    return ovr > thr;
}

// The same predicate without the division wherever the outcome is not within rounding distance of the threshold: inter/union > thr is
// decided by inter against thr*union with a 1e-6 relative guard band (the correctly rounded quotient and the product are each within 2^-24
// of their exact values); inside the band, or for a degenerate union, the exact expression above decides.  (The greedy pass of a frame is
// ~1 M of these on one CU: the division sequence was most of its time.)
__device__ __forceinline__ bool iou_gt_banded(const BoxF& a, float aarea, const BoxF& b, float barea, float thr) {
    const float xx1 = fmaxf(a.x1, b.x1), yy1 = fmaxf(a.y1, b.y1);
    const float xx2 = fminf(a.x2, b.x2), yy2 = fminf(a.y2, b.y2);
    const float w = fmaxf(0.f, xx2 - xx1), h = fmaxf(0.f, yy2 - yy1);
    const float inter = w * h;
    const float uni = aarea + barea - inter;
    const float tt = thr * uni;
    const bool sure_t = inter > tt * 1.000001f, sure_f = inter < tt * 0.999999f;
    if (uni > 0.f && uni < 3e38f && tt > 1e-30f && (sure_t || sure_f)) return sure_t;
    return inter / uni > thr;
}

constexpr int BOX_THREADS = 1024;
constexpr int BOX_MAX_DET = 2048;
constexpr int BOX_TILE = 4096;                                  // keys of one LDS sort tile
// dynamic LDS of box_sort_nms_kernel: [sort tile | bin prefix | kept boxes | kept area, id, confidence | staged candidate boxes]
constexpr int BOX_LDS_SORT = 0, BOX_LDS_PRE = BOX_TILE * 8, BOX_LDS_KBOX = BOX_LDS_PRE + BOX_TILE * 4, BOX_LDS_KAUX = BOX_LDS_KBOX + BOX_MAX_DET * 16;
constexpr int BOX_LDS_CBOX = BOX_LDS_KAUX + 3 * BOX_MAX_DET * 4;            // boxes of the 1024 sorted candidates being resolved
constexpr int BOX_LDS_BYTES = BOX_LDS_CBOX + BOX_THREADS * 16;

__device__ __forceinline__ unsigned long long box_shfl_xor_u64(unsigned long long v, int m) {
    const unsigned lo = __shfl_xor((unsigned)(v & 0xFFFFFFFFu), m, 64);
    const unsigned hi = __shfl_xor((unsigned)(v >> 32), m, 64);
    return ((unsigned long long)hi << 32) | lo;
}

// Ascending bitonic sort of sk[0..p2) (p2 = E * threads-in-use, a power of two <= 4096) by one 1024-thread workgroup.  Every thread holds E
// consecutive keys in registers: compare-exchange steps with partner distance j < E stay in the thread, j < 64 E go through lane
// shuffles (no barrier), only j >= 64 E is exchanged through LDS (10 of the 78 steps of a 4096-key sort; all 78 took a barrier before).
template <int E>
__device__ __forceinline__ void box_tile_sort(unsigned long long* sk, int p2, int t) {
    using u64_ = unsigned long long;
    const bool act = t * E < p2;
    u64_ v[E];
#pragma unroll
    for (int e = 0; e < E; ++e) v[e] = act ? sk[t * E + e] : ~0ull;
    for (int k = 2; k <= p2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j < E) {                                           // (constant register indices: a runtime v[e ^ j] would live in scratch)
                auto cx = [&](int e, int f) {
                    const bool up = ((t * E + e) & k) == 0;
                    const u64_ a = v[e], c = v[f];
                    if ((a > c) == up) { v[e] = c; v[f] = a; }
                };
                if constexpr (E == 4) {
                    if (j == 2) { cx(0, 2); cx(1, 3); }
                    else { cx(0, 1); cx(2, 3); }
                } else if constexpr (E == 2) {
                    cx(0, 1);
                }
            } else if (j < 64 * E) {
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const int idx = t * E + e;
                    const u64_ c = box_shfl_xor_u64(v[e], j / E);
                    const bool lower = (idx & j) == 0, up = (idx & k) == 0;
                    const u64_ mn = v[e] < c ? v[e] : c, mx = v[e] < c ? c : v[e];
                    v[e] = (lower == up) ? mn : mx;
                }
            } else {
                __syncthreads();                                   // (earlier partner reads are done)
                if (act) {
#pragma unroll
                    for (int e = 0; e < E; ++e) sk[t * E + e] = v[e];
                }
                __syncthreads();
                if (act) {
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        const int idx = t * E + e;
                        const u64_ c = sk[idx ^ j];
                        const bool lower = (idx & j) == 0, up = (idx & k) == 0;
                        const u64_ mn = v[e] < c ? v[e] : c, mx = v[e] < c ? c : v[e];
                        v[e] = (lower == up) ? mn : mx;
                    }
                }
            }
        }
    }
    __syncthreads();
    if (act) {
#pragma unroll
        for (int e = 0; e < E; ++e) sk[t * E + e] = v[e];
    }
    __syncthreads();
}

// one workgroup per image: candidate keys in confidence order, then chunked greedy NMS.
//   n <= 4096 keys: one LDS sort, the NMS reads the sorted keys from LDS.
//   more: the keys are consumed in ROUNDS of whole confidence bins (4096-bin histogram over the confidence word of the keys; a round takes
//   the next bins that together fit the 4096-key tile, so equal confidences stay together), each round sorted in LDS and fed to the SAME
//   greedy pass, which stops at max_det boxes -- usually inside the first round, long before it has seen 30 000 candidates.  (The first
//   version sorted the whole list in global memory and restarted the NMS when the first 4096 did not yield max_det boxes: a frame's
//   6 500 candidates / 300 boxes paid ~90 us for that.)  A single bin that overflows the tile: full bitonic sort through global memory.
__global__ __launch_bounds__(BOX_THREADS) void box_sort_nms_kernel(const float* __restrict__ pred, int N, int nc, float iou_thres, float conf_lo,
                                                                    int agnostic, int max_det, int max_nms, float max_wh,
                                                                    u64* __restrict__ keys_all, int cap, int cap_pow2,
                                                                    int* __restrict__ count, float* __restrict__ out_det,
                                                                    int* __restrict__ out_count) {
    extern __shared__ __attribute__((aligned(16))) char box_lds[];
    u64* sk = reinterpret_cast<u64*>(box_lds + BOX_LDS_SORT);
    int* pre = reinterpret_cast<int*>(box_lds + BOX_LDS_PRE);
    BoxF* kbox = reinterpret_cast<BoxF*>(box_lds + BOX_LDS_KBOX);
    float* karea = reinterpret_cast<float*>(box_lds + BOX_LDS_KAUX);
    unsigned* kid = reinterpret_cast<unsigned*>(box_lds + BOX_LDS_KAUX + BOX_MAX_DET * 4);
    float* kconf = reinterpret_cast<float*>(box_lds + BOX_LDS_KAUX + 2 * BOX_MAX_DET * 4);
    BoxF* cbox = reinterpret_cast<BoxF*>(box_lds + BOX_LDS_CBOX);
    __shared__ u64 supmask[BOX_THREADS / 64];
    __shared__ u64 cmask[64];
    __shared__ int wsum[BOX_THREADS / 64];
    __shared__ int s_nkept, s_bsel, s_cnt;
    constexpr int TILE = BOX_TILE;

    const int b = blockIdx.x;
    const int t = threadIdx.x;
    u64* keys = keys_all + (long)b * cap_pow2;
    const bool overflow = count[b] > cap_pow2;
    const int n = min(count[b], cap_pow2);
    int P = 1;
    while (P < n) P <<= 1;
    const int lane = t & 63, wave = t >> 6;
    const int n_all = min(n, max_nms);
    const int n_up = (n + BOX_THREADS - 1) / BOX_THREADS * BOX_THREADS;          // whole waves stay in the key loops (ballots)
    // the first 8192 keys live in registers (one global round trip for all passes over them)
    constexpr int KR = 8;
    u64 kreg[KR];
#pragma unroll
    for (int q = 0; q < KR; ++q) { const int i = t + q * BOX_THREADS; kreg[q] = i < n ? keys[i] : ~0ull; }
    if (t == 0) s_nkept = 0;
    __syncthreads();
#ifdef YP_PROBE_BOX
    u64* dbg = keys + cap_pow2 - 64; int dbi = 0;
#define BOXTL(v) do { if (t == 0) { dbg[dbi++] = wall_clock64(); dbg[dbi++] = (u64)(v); } } while (0)
#else
#define BOXTL(v) do { } while (0)
#endif
    BOXTL(n);

    auto sort_tile = [&](int m) {                    // sk[0..m) -> ascending (padded with ~0 to a power of two)
        int p2 = 1;
        while (p2 < m) p2 <<= 1;
        for (int i = m + t; i < p2; i += BOX_THREADS) sk[i] = ~0ull;
        __syncthreads();
        if (p2 >= 4 * BOX_THREADS) box_tile_sort<4>(sk, p2, t);
        else if (p2 >= 2 * BOX_THREADS) box_tile_sort<2>(sk, p2, t);
        else box_tile_sort<1>(sk, p2, t);
    };
    // Full sort (fallback): bitonic network; every compare-exchange step whose partner distance j is < 4096 stays inside a 4096-key tile, so
    // those steps run on tiles held in LDS; only the j >= 4096 steps go through global memory.
    auto full_sort = [&]() {
        for (int i = n + t; i < P; i += BOX_THREADS) keys[i] = ~0ull;
        __syncthreads();
        auto tile_pass = [&](int kfirst, int klast) {      // stages kfirst..klast (powers of two), steps j = min(k/2, TILE/2) .. 1, per tile
            for (int t0 = 0; t0 < P; t0 += TILE) {
                const int tn = min(TILE, P - t0);
                for (int i = t; i < tn; i += BOX_THREADS) sk[i] = keys[t0 + i];
                __syncthreads();
                for (int k = kfirst; k <= klast; k <<= 1) {
                    for (int j = min(k >> 1, TILE >> 1); j > 0; j >>= 1) {
                        for (int i = t; i < tn; i += BOX_THREADS) {
                            const int ixj = i ^ j;
                            if (ixj > i) {
                                const u64 a = sk[i], c = sk[ixj];
                                const bool up = ((t0 + i) & k) == 0;
                                if ((a > c) == up) { sk[i] = c; sk[ixj] = a; }
                            }
                        }
                        __syncthreads();
                    }
                }
                for (int i = t; i < tn; i += BOX_THREADS) keys[t0 + i] = sk[i];
                __syncthreads();
            }
        };
        tile_pass(2, min(P, TILE));
        for (int k = TILE << 1; k <= P; k <<= 1) {
            for (int j = k >> 1; j >= TILE; j >>= 1) {
                for (int i = t; i < P; i += BOX_THREADS) {
                    const int ixj = i ^ j;
                    if (ixj > i) {
                        const u64 a = keys[i], c = keys[ixj];
                        const bool up = (i & k) == 0;
                        if ((a > c) == up) { keys[i] = c; keys[ixj] = a; }
                    }
                }
                __syncthreads();
            }
            tile_pass(k, k);
        }
    };

    // ---- greedy NMS over a sorted run of candidates, 64 at a time; continues from the kept list in LDS (s_nkept)
    const int no = nc + 5;
    const float* pb = pred + (long)b * N * no;
    constexpr int NW = BOX_THREADS / 64;
    auto run_nms = [&](const u64* list, int cnt) {
    // The boxes of 1024 sorted candidates at a time are staged in LDS by all threads at once (key -> prediction row -> box: dependent
    // global loads, 1-2 us cold; the next 1024 are in flight while the current ones are resolved).  Fetched per 64-candidate chunk -- even
    // one chunk ahead -- every chunk waited about a microsecond for them.
    auto fetch = [&](int ci) -> BoxF {
        BoxF bx{0.f, 0.f, 0.f, 0.f};
        if (ci < cnt) {
            const unsigned id = (unsigned)(list[ci] & 0xFFFFFFFFu);
            const unsigned row = id / (unsigned)nc, cls = id - row * (unsigned)nc;
            const float* r = pb + (long)row * no;
            const float cx = r[0], cy = r[1], w = r[2], h = r[3];
            const float off = agnostic ? 0.f : (float)cls * max_wh;
            // xywh2xyxy (utils/general_yolo.py:623-630) then "+ c" (:216-217), both in fp32
            bx.x1 = (cx - w / 2) + off; bx.y1 = (cy - h / 2) + off;
            bx.x2 = (cx + w / 2) + off; bx.y2 = (cy + h / 2) + off;
        }
        return bx;
    };
    BoxF staged = fetch(t);
    bool full = false;
    for (int blk0 = 0; blk0 < cnt && !full; blk0 += BOX_THREADS) {
    __syncthreads();                                   // (the previous block's readers are done)
    cbox[t] = staged;
    staged = fetch(blk0 + BOX_THREADS + t);
    __syncthreads();
    const int blk_end = min(cnt, blk0 + BOX_THREADS);
    for (int base = blk0; base < blk_end; base += 64) {
        const int nk0 = s_nkept;
        if (nk0 >= max_det) { full = true; break; }
        const int ci = base + lane;
        const bool valid = ci < cnt;
        const BoxF bx = cbox[ci - blk0];
        const float area = (bx.x2 - bx.x1) * (bx.y2 - bx.y1);
        const u64 key = valid ? list[ci] : 0ull;
        const unsigned id = (unsigned)(key & 0xFFFFFFFFu);
        const float conf = __uint_as_float(0xFFFFFFFFu - (unsigned)(key >> 32));
        // every wave tests the chunk against a strided share of the kept list
        // (four kept boxes per step so that their LDS reads are in flight together; a wave leaves the loop once all its candidates are gone)
        bool sup = false;
        int k = wave;
        for (; k + 3 * NW < nk0; k += 4 * NW) {
            const BoxF b0 = kbox[k], b1 = kbox[k + NW], b2 = kbox[k + 2 * NW], b3 = kbox[k + 3 * NW];
            const float a0 = karea[k], a1 = karea[k + NW], a2 = karea[k + 2 * NW], a3 = karea[k + 3 * NW];
            sup = sup | iou_gt_banded(b0, a0, bx, area, iou_thres) | iou_gt_banded(b1, a1, bx, area, iou_thres) | iou_gt_banded(b2, a2, bx, area, iou_thres) |
                  iou_gt_banded(b3, a3, bx, area, iou_thres);
            if (__ballot(valid && !sup) == 0) break;
        }
        for (; k < nk0 && !sup; k += NW) sup = iou_gt_banded(kbox[k], karea[k], bx, area, iou_thres);
        const u64 m = __ballot(sup && valid);
        if (lane == 0) supmask[wave] = m;
        // ... and four rows of the chunk's own 64 x 64 suppression matrix (row i: the later candidates j > i that box i suppresses), which does
        // not depend on the kept list: the serial part below is then bit arithmetic only (it was one broadcast + IoU + ballot per kept box,
        // ~150 clocks x up to 64 per chunk, on the critical path of every chunk)
#pragma unroll
        for (int r = 0; r < 64 / NW; ++r) {
            const int i = __builtin_amdgcn_readfirstlane(wave * (64 / NW) + r);
            BoxF bi;
            bi.x1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bx.x1), i));
            bi.y1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bx.y1), i));
            bi.x2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bx.x2), i));
            bi.y2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bx.y2), i));
            const float ai = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(area), i));
            const u64 row = __ballot(valid && lane > i && iou_gt_banded(bi, ai, bx, area, iou_thres));
            if (lane == 0) cmask[i] = row;
        }
        __syncthreads();
        if (wave == 0) {
            u64 dead = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) dead |= supmask[w];
            u64 alive = __ballot(valid) & ~dead;
            const u64 mine = cmask[lane];
            const unsigned mlo = (unsigned)(mine & 0xFFFFFFFFu), mhi = (unsigned)(mine >> 32);
            int nk = nk0;
            u64 keep = 0;
            while (alive != 0 && nk < max_det) {
                const int i = __ffsll((long long)alive) - 1;                                       // (uniform: alive lives in scalar registers)
                const u64 mi = ((u64)(unsigned)__builtin_amdgcn_readlane((int)mhi, i) << 32) | (unsigned)__builtin_amdgcn_readlane((int)mlo, i);
                keep |= 1ull << i;
                ++nk;
                alive &= ~(1ull << i);
                alive &= ~mi;
            }
            if ((keep >> lane) & 1ull) {
                const int slot = nk0 + __popcll(keep & ((1ull << lane) - 1ull));
                kbox[slot] = bx; karea[slot] = area; kid[slot] = id; kconf[slot] = conf;
            }
            if (lane == 0) s_nkept = nk;
        }
        __syncthreads();
    }
    }
    };

    const bool probe_sort_only = max_wh < 0.f;           // (probe: negative max_wh = ordering work only)
    bool fallback = false;
    if (n <= TILE) {
#pragma unroll
        for (int q = 0; q < TILE / BOX_THREADS; ++q) { const int i = t + q * BOX_THREADS; if (i < n) sk[i] = kreg[q]; }
        __syncthreads();
        sort_tile(n);
        if (!probe_sort_only) run_nms(sk, n_all);
    } else {
        // confidences lie in (conf_thres, 1]: the key's confidence word (0xFFFFFFFF - float bits) in [~bits(1), ~bits(conf_thres)]
        const unsigned base_h = 0xFFFFFFFFu - __float_as_uint(1.0f);
        const unsigned span_h = (0xFFFFFFFFu - __float_as_uint(conf_lo)) - base_h;
        int shift = 0;
        while ((span_h >> shift) >= (unsigned)TILE) ++shift;
        auto bin_of = [&](u64 key) -> int {
            const unsigned h = (unsigned)(key >> 32);
            return h <= base_h ? 0 : (int)min((h - base_h) >> shift, (unsigned)TILE - 1u);
        };
        // f(i, key) over all keys, whole waves (i may be >= n: key = ~0)
        auto for_keys = [&](auto&& f) {
#pragma unroll
            for (int q = 0; q < KR; ++q)
                if (q * BOX_THREADS < n_up) f(t + q * BOX_THREADS, kreg[q]);
            for (int i = t + KR * BOX_THREADS; i < n_up; i += BOX_THREADS) f(i, i < n ? keys[i] : ~0ull);
        };
        for (int i = t; i < TILE; i += BOX_THREADS) pre[i] = 0;
        __syncthreads();
        for_keys([&](int i, u64 key) { if (i < n) atomicAdd(&pre[bin_of(key)], 1); });
        __syncthreads();
        {   // inclusive prefix over the bins (4 per thread)
            int v[4], sum = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) { v[q] = pre[4 * t + q]; sum += v[q]; }
            int incl = sum;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(incl, o, 64); if (lane >= o) incl += u; }
            if (lane == 63) wsum[wave] = incl;
            __syncthreads();
            int run = incl - sum;
            for (int w = 0; w < wave; ++w) run += wsum[w];
#pragma unroll
            for (int q = 0; q < 4; ++q) { run += v[q]; pre[4 * t + q] = run; }
            __syncthreads();
        }
        BOXTL(2);
        int lo = -1, taken = 0;                           // bins <= lo (taken keys) are consumed
        while (taken < n_all) {
            // the next bins that together fit the tile: hi = the largest bin > lo with pre[hi] - taken <= TILE that holds keys itself
            if (t == 0) s_bsel = -1;
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int bin = 4 * t + q;
                const int upto = pre[bin], own = upto - (bin ? pre[bin - 1] : 0);
                if (bin > lo && own > 0 && upto - taken <= TILE) atomicMax(&s_bsel, bin);
            }
            __syncthreads();
            const int hi = s_bsel;
            if (hi < 0) { fallback = true; break; }      // the next bin alone overflows the tile
            if (t == 0) s_cnt = 0;
            __syncthreads();
            for_keys([&](int i, u64 key) {
                const int bin = bin_of(key);
                const bool sel = i < n && bin > lo && bin <= hi;
                const u64 mk = __ballot(sel);                    // one LDS atomic per wave (64 returning atomics on one address serialise)
                if (mk != 0) {
                    const int leader = __ffsll((long long)mk) - 1;
                    int wbase = 0;
                    if (lane == leader) wbase = atomicAdd(&s_cnt, __popcll(mk));
                    wbase = __shfl(wbase, leader, 64);
                    if (sel) sk[wbase + __popcll(mk & ((1ull << lane) - 1ull))] = key;
                }
            });
            __syncthreads();
            const int m = s_cnt;
            BOXTL(m);
            // (measured and dropped: a counting sort by bin + per-bin insertion sort instead of the network -- 50 us, and planted confidences
            // put more than 16 keys into single bins, which needs the network anyway)
            sort_tile(m);
            BOXTL(1);
            if (!probe_sort_only) run_nms(sk, min(m, n_all - taken));
            BOXTL(s_nkept);
            taken += m;
            lo = hi;
            if (probe_sort_only || s_nkept >= max_det) break;
        }
    }
    if (fallback) {                                       // sort everything through global memory, redo the greedy pass on the whole list
        __syncthreads();
        if (t == 0) s_nkept = 0;
        full_sort();
        if (!probe_sort_only) run_nms(keys, n_all);
    }
    // ---- emit (x1,y1,x2,y2,conf,cls) without the class offset
    const int nk = min(s_nkept, max_det);
    if (t == 0) out_count[b] = overflow ? -1 : nk;   // -1: candidate list truncated, result not exact
    for (int k = t; k < nk; k += BOX_THREADS) {
        const unsigned id = kid[k];
        const unsigned row = id / (unsigned)nc, cls = id - row * (unsigned)nc;
        const float* r = pb + (long)row * no;
        const float cx = r[0], cy = r[1], w = r[2], h = r[3];
        float* o = out_det + ((long)b * max_det + k) * 6;
        o[0] = cx - w / 2; o[1] = cy - h / 2; o[2] = cx + w / 2; o[3] = cy + h / 2;
        o[4] = kconf[k];
        o[5] = (float)cls;
    }
}

// =========================================================================================
// descriptor sampling: bilinear grid_sample(align_corners=True) + L2 renorm, one wave per point
// =========================================================================================
__global__ void desc_sample_kernel(const float* __restrict__ desc, int D, int Hc, int Wc, long sc, long sy, long sx,
                                   const float* __restrict__ pts, int N, int cell, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
    const double Wf = (double)(Wc * cell), Hf = (double)(Hc * cell);
    for (long pi = wave; pi < N; pi += nwaves) {
        // reference: float64 normalisation by the FULL resolution, then .float()
        const float gx = (float)((double)pts[pi * 2] / (Wf / 2.) - 1.);
        const float gy = (float)((double)pts[pi * 2 + 1] / (Hf / 2.) - 1.);
        const float ix = ((gx + 1.f) / 2.f) * (float)(Wc - 1);
        const float iy = ((gy + 1.f) / 2.f) * (float)(Hc - 1);
        const float fx = floorf(ix), fy = floorf(iy);
        const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
        const float wnw = ((float)x1 - ix) * ((float)y1 - iy);
        const float wne = (ix - (float)x0) * ((float)y1 - iy);
        const float wsw = ((float)x1 - ix) * (iy - (float)y0);
        const float wse = (ix - (float)x0) * (iy - (float)y0);
        const bool inx0 = x0 >= 0 && x0 < Wc, inx1 = x1 >= 0 && x1 < Wc;
        const bool iny0 = y0 >= 0 && y0 < Hc, iny1 = y1 >= 0 && y1 < Hc;
        float ss = 0.f;
        for (int c = lane; c < D; c += 64) {
            const float* pc = desc + c * sc;
            float v = 0.f;
            if (inx0 && iny0) v += pc[y0 * sy + x0 * sx] * wnw;
            if (inx1 && iny0) v += pc[y0 * sy + x1 * sx] * wne;
            if (inx0 && iny1) v += pc[y1 * sy + x0 * sx] * wsw;
            if (inx1 && iny1) v += pc[y1 * sy + x1 * sx] * wse;
            out[(long)c * N + pi] = v;
            ss += v * v;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
        const float nrm = sqrtf(ss);
        for (int c = lane; c < D; c += 64) out[(long)c * N + pi] /= nrm;
    }
}

// =========================================================================================
// mutual nearest neighbour matcher.  fp32 MFMA (v_mfma_f32_16x16x4_f32: an exact f32 fma chain)
// computes 64x64 tiles of d1^T d2; distances are formed per element exactly like the reference
// (sqrt(2 - 2*clip(dot,-1,1))) and reduced to per-row / per-column minima through packed
// (dist_bits << 32 | index) keys, whose u64 minimum is "smallest distance, first index" ==
// np.argmin's tie rule.  The N1 x N2 distance matrix is never written.
// =========================================================================================
__device__ __forceinline__ u64 min_u64(u64 a, u64 b) { return a < b ? a : b; }
__device__ __forceinline__ u64 shfl_xor_u64(u64 v, int m) {
    const unsigned lo = __shfl_xor((unsigned)(v & 0xFFFFFFFFu), m, 64);
    const unsigned hi = __shfl_xor((unsigned)(v >> 32), m, 64);
    return ((u64)hi << 32) | lo;
}

__global__ __launch_bounds__(256) void mnn_tile_kernel(const float* __restrict__ d1, int N1, const float* __restrict__ d2, int N2,
                                                       int D, u64* __restrict__ rowmin, u64* __restrict__ colmin) {
    constexpr int BT = 64, KT = 16;
    __shared__ float sA[KT][BT + 4];
    __shared__ float sB[KT][BT + 4];
    const int i0 = blockIdx.y * BT, j0 = blockIdx.x * BT;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wi = (wave & 1) * 32, wj = (wave >> 1) * 32;
    const int p = lane & 15, g = lane >> 4;
    f32x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c) acc[a][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < D; k0 += KT) {
        for (int e = t; e < KT * BT; e += 256) {
            const int kk = e / BT, c = e - kk * BT;
            const int k = k0 + kk;
            sA[kk][c] = (k < D && i0 + c < N1) ? d1[(long)k * N1 + i0 + c] : 0.f;
            sB[kk][c] = (k < D && j0 + c < N2) ? d2[(long)k * N2 + j0 + c] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < KT; kk += 4) {
            float af[2], bf[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) af[a] = sA[kk + g][wi + a * 16 + p];
#pragma unroll
            for (int c = 0; c < 2; ++c) bf[c] = sB[kk + g][wj + c * 16 + p];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int c = 0; c < 2; ++c) acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[a], bf[c], acc[a][c], 0, 0, 0);
        }
        __syncthreads();
    }
    // acc[a][c][r] = dot(row i = i0+wi+a*16+4*g+r, col j = j0+wj+c*16+p)
    u64 rkey[2][4], ckey[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) ckey[c] = ~0ull;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            rkey[a][r] = ~0ull;
            const int i = i0 + wi + a * 16 + 4 * g + r;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int j = j0 + wj + c * 16 + p;
                if (i < N1 && j < N2) {
                    const float dc = fminf(fmaxf(acc[a][c][r], -1.f), 1.f);
                    const float dist = __fsqrt_rn(2.f - 2.f * dc);
                    const u64 bits = (u64)__float_as_uint(dist) << 32;
                    rkey[a][r] = min_u64(rkey[a][r], bits | (unsigned)j);
                    ckey[c] = min_u64(ckey[c], bits | (unsigned)i);
                }
            }
        }
    // rows: reduce over the 16 lanes that share g
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            u64 v = rkey[a][r];
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) v = min_u64(v, shfl_xor_u64(v, o));
            const int i = i0 + wi + a * 16 + 4 * g + r;
            if (p == 0 && i < N1 && v != ~0ull) atomicMin(&rowmin[i], v);
        }
    // columns: reduce over the 4 lane groups
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        u64 v = ckey[c];
        v = min_u64(v, shfl_xor_u64(v, 16));
        v = min_u64(v, shfl_xor_u64(v, 32));
        const int j = j0 + wj + c * 16 + p;
        if (g == 0 && j < N2 && v != ~0ull) atomicMin(&colmin[j], v);
    }
}

__global__ void fill_u64_kernel(u64* p, long n, u64 v) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = v;
}

// single workgroup: mutual check + threshold, compacted in ascending idx1 order
__global__ __launch_bounds__(1024) void mnn_select_kernel(const u64* __restrict__ rowmin, const u64* __restrict__ colmin, int N1,
                                                          float nn_thresh, float* __restrict__ out, int* __restrict__ out_count,
                                                          int max_out) {
    __shared__ int wsum[16];
    __shared__ int s_base;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t == 0) s_base = 0;
    __syncthreads();
    const int cols = min(N1, max_out);
    for (int c0 = 0; c0 < N1; c0 += 1024) {
        const int i = c0 + t;
        bool keep = false;
        int j = 0;
        float dist = 0.f;
        if (i < N1) {
            const u64 rk = rowmin[i];
            j = (int)(rk & 0xFFFFFFFFu);
            dist = __uint_as_float((unsigned)(rk >> 32));
            keep = dist < nn_thresh && (int)(colmin[j] & 0xFFFFFFFFu) == i;
        }
        const u64 bal = __ballot(keep);
        const int within = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wave] = __popcll(bal);
        __syncthreads();
        int before = s_base;
        for (int w = 0; w < wave; ++w) before += wsum[w];
        if (keep) {
            const int pos = before + within;
            if (pos < cols) {
                out[pos] = (float)i;
                out[cols + pos] = (float)j;
                out[2 * cols + pos] = dist;
            }
        }
        __syncthreads();
        if (t == 0) {
            int tot = 0;
            for (int w = 0; w < 16; ++w) tot += wsum[w];
            s_base += tot;
        }
        __syncthreads();
    }
    if (t == 0) *out_count = s_base;
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
inline int next_pow2(int v) { int p = 1; while (p < v) p <<= 1; return p; }

}  // namespace

// -------------------------------------------------------------------------------------------
extern "C" int yp_kp_decode(const float* semi, int B, int Hc, int Wc, int64_t sb, int64_t sc, int64_t sy, int64_t sx, int mode,
                            float* heat, void* stream) {
    YP_REQUIRE(semi && heat && B > 0 && Hc > 0 && Wc > 0 && (mode == 0 || mode == 1), "yp_kp_decode: bad arguments");
    const long nwork = (long)B * Hc * ((Wc + 31) / 32);
    kp_decode_kernel<<<grid_for(nwork, 1), 256, 0, (hipStream_t)stream>>>(semi, B, Hc, Wc, sb, sc, sy, sx, mode, heat);
    YP_CHECK_HIP(hipGetLastError());
    return YP_OK;
}

// workspace layout: state[B*HW] u8 | cand[B*HW] i32 | kept[B*HW] i32 | kscore[B*HW] f32 | ncand[B] | nkept[B] | left[KP_MAX_ROUNDS]
constexpr int KP_ROUND_BATCH = 8;
constexpr int KP_MAX_ROUNDS = 4096;

extern "C" size_t yp_kp_nms_workspace_bytes(int B, int H, int W) {
    const size_t hw = (size_t)B * H * W;
    return align_up(hw, 256) + 3 * align_up(hw * 4, 256) + 2 * align_up((size_t)B * 4, 256) + align_up((size_t)KP_MAX_ROUNDS * 4, 256);
}

// byte offset, inside that workspace, of the per-image count of pixels that passed the confidence threshold (int32 [B]) -- for callers that
// read it back with their own synchronisation instead of re-deriving the layout
extern "C" size_t yp_kp_nms_candidate_count_offset(int B, int H, int W) {
    const size_t hw = (size_t)B * H * W;
    return align_up(hw, 256) + 3 * align_up(hw * 4, 256);
}

// fixed_rounds == 0: rounds in batches of KP_ROUND_BATCH until the undecided counter read back by the host is 0 (one stream
// synchronisation per batch).  fixed_rounds > 0: exactly that many rounds, no host synchronisation; the last round's undecided counter
// goes to *undecided_out (device) for the caller to check at its own synchronisation point.
static int kp_nms_run(const float* heat, int B, int H, int W, float conf_thresh, int radius, int border, float* out_xyc, int32_t* out_count, int max_out,
                      void* workspace, size_t workspace_bytes, int fixed_rounds, int32_t* undecided_out, void* stream) {
    YP_REQUIRE(heat && out_xyc && out_count && workspace, "yp_kp_nms: null pointer");
    YP_REQUIRE(B > 0 && H > 0 && W > 0 && radius >= 0 && border >= 0 && max_out > 0, "yp_kp_nms: bad dims");
    if (workspace_bytes < yp_kp_nms_workspace_bytes(B, H, W)) {
        yp_set_error("yp_kp_nms: workspace %zu < %zu", workspace_bytes, yp_kp_nms_workspace_bytes(B, H, W));
        return YP_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const size_t hw = (size_t)B * H * W;
    char* ws = (char*)workspace;
    unsigned char* state = (unsigned char*)ws; ws += align_up(hw, 256);
    int* cand = (int*)ws; ws += align_up(hw * 4, 256);
    int* kept = (int*)ws; ws += align_up(hw * 4, 256);
    float* kscore = (float*)ws; ws += align_up(hw * 4, 256);
    int* ncand = (int*)ws; ws += align_up((size_t)B * 4, 256);
    int* nkept = (int*)ws; ws += align_up((size_t)B * 4, 256);
    int* left = (int*)ws;
    // zero the counters (ncand, nkept, left are contiguous)
    YP_CHECK_HIP(hipMemsetAsync(ncand, 0, 2 * align_up((size_t)B * 4, 256) + (size_t)KP_MAX_ROUNDS * 4, st));
    kp_threshold_kernel<<<dim3(KP_SPANS, B), 256, 0, st>>>(heat, B, H * W, conf_thresh, state, cand, ncand);
    YP_CHECK_HIP(hipGetLastError());
    int round = 0;
    if (fixed_rounds > 0) {
        YP_REQUIRE(fixed_rounds <= KP_MAX_ROUNDS && undecided_out != nullptr, "yp_kp_nms_async: 1..%d rounds and a device counter", KP_MAX_ROUNDS);
        for (; round < fixed_rounds; ++round)
            kp_round_kernel<<<512, 256, 0, st>>>(heat, B, H, W, radius, state, cand, ncand, round ? left + round - 1 : nullptr, left + round);
        YP_CHECK_HIP(hipGetLastError());
        YP_CHECK_HIP(hipMemcpyAsync(undecided_out, left + round - 1, sizeof(int), hipMemcpyDeviceToDevice, st));
    } else {
        for (;;) {
            for (int k = 0; k < KP_ROUND_BATCH; ++k, ++round) {
                kp_round_kernel<<<512, 256, 0, st>>>(heat, B, H, W, radius, state, cand, ncand, round ? left + round - 1 : nullptr,
                                                     left + round);
            }
            YP_CHECK_HIP(hipGetLastError());
            int h_left = 0;
            YP_CHECK_HIP(hipMemcpyAsync(&h_left, left + round - 1, sizeof(int), hipMemcpyDeviceToHost, st));
            YP_CHECK_HIP(hipStreamSynchronize(st));
            if (h_left == 0) break;
            YP_REQUIRE(round + KP_ROUND_BATCH <= KP_MAX_ROUNDS, "yp_kp_nms: no fix-point after %d rounds", round);
        }
    }
    kp_collect_kernel<<<256, 256, 0, st>>>(heat, B, H, W, border, state, cand, ncand, kept, kscore, nkept);
    // the candidate list is dead after the collect: its storage (HW ints per image) takes the partial ranks, split * kept of them --
    // kept points are pairwise > radius apart, so kept <= ceil(H/(r+1)) * ceil(W/(r+1))
    const long max_kept = (long)((H + radius) / (radius + 1)) * ((W + radius) / (radius + 1));
    int split = (int)((long)H * W / (max_kept > 0 ? max_kept : 1));
    split = split < 1 ? 1 : (split > KP_RANK_SPLIT ? KP_RANK_SPLIT : split);
    for (int b = 0; b < B; ++b) {
        int* partial = cand + (size_t)b * H * W;
        kp_rank_kernel<<<dim3(64, split), 256, 0, st>>>(b, H * W, kept, kscore, nkept, partial);
        kp_rank_write_kernel<<<64, 256, 0, st>>>(b, H * W, W, kept, kscore, nkept, partial, split, out_xyc, out_count, max_out);
    }
    YP_CHECK_HIP(hipGetLastError());
    return YP_OK;
}

extern "C" int yp_kp_nms(const float* heat, int B, int H, int W, float conf_thresh, int radius, int border, float* out_xyc,
                         int32_t* out_count, int max_out, void* workspace, size_t workspace_bytes, void* stream) {
    return kp_nms_run(heat, B, H, W, conf_thresh, radius, border, out_xyc, out_count, max_out, workspace, workspace_bytes, 0, nullptr, stream);
}

extern "C" int yp_kp_nms_async(const float* heat, int B, int H, int W, float conf_thresh, int radius, int border, float* out_xyc,
                               int32_t* out_count, int max_out, void* workspace, size_t workspace_bytes, int rounds, int32_t* undecided_out,
                               void* stream) {
    return kp_nms_run(heat, B, H, W, conf_thresh, radius, border, out_xyc, out_count, max_out, workspace, workspace_bytes, rounds, undecided_out, stream);
}

// workspace layout: count[B] i32 | keys[B*cap_pow2] u64
static int box_cap(int N, int nc, int multi_label) {
    const long full = (long)N * (multi_label ? nc : 1);
    return (int)(full < (1l << 21) ? full : (1l << 21));
}

extern "C" size_t yp_box_nms_workspace_bytes(int B, int N, int nc, int multi_label, int max_nms) {
    (void)max_nms;
    const int cap2 = next_pow2(box_cap(N, nc, multi_label));
    return align_up((size_t)B * 4, 256) + (size_t)B * cap2 * 8;
}

extern "C" int yp_box_nms(const float* pred, int B, int N, int nc, float conf_thres, float iou_thres, int multi_label, int agnostic,
                          int max_det, int max_nms, float max_wh, float* out_det, int32_t* out_count, void* workspace,
                          size_t workspace_bytes, void* stream) {
    return yp_box_nms_classes(pred, B, N, nc, conf_thres, iou_thres, multi_label, agnostic, max_det, max_nms, max_wh, nullptr, out_det, out_count, workspace,
                              workspace_bytes, stream);
}

extern "C" int yp_box_nms_classes(const float* pred, int B, int N, int nc, float conf_thres, float iou_thres, int multi_label, int agnostic,
                                  int max_det, int max_nms, float max_wh, const uint32_t* class_mask, float* out_det, int32_t* out_count, void* workspace,
                                  size_t workspace_bytes, void* stream) {
    YP_REQUIRE(pred && out_det && out_count && workspace, "yp_box_nms: null pointer");
    YP_REQUIRE(B > 0 && N > 0 && nc > 0 && max_det > 0 && max_det <= BOX_MAX_DET && max_nms > 0, "yp_box_nms: bad dims (max_det <= %d)", BOX_MAX_DET);
    YP_REQUIRE(conf_thres >= 0.f && conf_thres <= 1.f && iou_thres >= 0.f && iou_thres <= 1.f, "yp_box_nms: thresholds must be in [0,1]");
    YP_REQUIRE((long)N * nc < (1l << 32), "yp_box_nms: N*nc overflows the candidate id");
    multi_label = multi_label && nc > 1;      // utils/general_yolo.py:158
    if (workspace_bytes < yp_box_nms_workspace_bytes(B, N, nc, multi_label, max_nms)) {
        yp_set_error("yp_box_nms: workspace too small");
        return YP_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const int cap = box_cap(N, nc, multi_label);
    const int cap2 = next_pow2(cap);
    int* count = (int*)workspace;
    u64* keys = (u64*)((char*)workspace + align_up((size_t)B * 4, 256));
    YP_CHECK_HIP(hipMemsetAsync(count, 0, (size_t)B * 4, st));
    box_candidates_kernel<<<dim3(box_spans(B), B), 256, 0, st>>>(pred, B, N, nc, conf_thres, multi_label, class_mask, keys, cap2, count);
    static YpLdsAttr attr;
    YP_CHECK_HIP(yp_set_max_lds(attr, (const void*)box_sort_nms_kernel, BOX_LDS_BYTES));
    box_sort_nms_kernel<<<B, BOX_THREADS, BOX_LDS_BYTES, st>>>(pred, N, nc, iou_thres, conf_thres, agnostic, max_det, max_nms, max_wh, keys, cap, cap2, count,
                                                               out_det, out_count);
    YP_CHECK_HIP(hipGetLastError());
    return YP_OK;
}

// ---------------------------------------------------------------------------------------------
// Keypoints that fall inside a detected box are dropped (reference demo.py:176-196 `filter_points`): the reference
// paints mask[y0:y1, x0:x1] = 0 for rint(xyxy) of every kept detection (numpy slice semantics: a negative bound counts from
// the end, bounds are clamped to the image) and keeps a point when mask[int(y), int(x)] == 1.  Order-preserving compaction
// by one workgroup: per 256-point chunk a ballot / prefix-count gives each surviving point its output slot.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int py_slice_bound(int v, int dim) { return v < 0 ? (v + dim < 0 ? 0 : v + dim) : (v > dim ? dim : v); }

// Pass 1 (all CUs): one point per thread against every box; the verdict goes to out[3 i] (the output buffer doubles as the flag array: the
// compaction below reads a 1024-point block's flags before it writes, and writes slots <= the indices it has read).  On one workgroup
// the 4 000 points x 300 boxes of a frame were 65 us of comparisons on a single CU.
__global__ __launch_bounds__(256) void pts_box_flags_kernel(const float* __restrict__ pts, const int* __restrict__ n_in_dev, int n_in_host,
                                                            const float* __restrict__ boxes, const int* __restrict__ n_boxes_dev, int n_boxes_host,
                                                            int box_stride, int H, int W, float* __restrict__ flags) {
    __shared__ int4 sb[512];
    const int n = n_in_dev ? *n_in_dev : n_in_host;
    if ((int)(blockIdx.x * 256) >= n) return;
    int nb = n_boxes_dev ? *n_boxes_dev : n_boxes_host;
    if (nb > 512) nb = 512;
    for (int i = threadIdx.x; i < nb; i += 256) {
        const float* b = boxes + (size_t)i * box_stride;
        sb[i] = int4{py_slice_bound((int)rintf(b[0]), W), py_slice_bound((int)rintf(b[1]), H), py_slice_bound((int)rintf(b[2]), W),
                     py_slice_bound((int)rintf(b[3]), H)};
    }
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int xi = (int)pts[3 * i], yi = (int)pts[3 * i + 1];
    int inside = 0;
#pragma unroll 4
    for (int k = 0; k < nb; ++k) {
        const int4 q = sb[k];
        inside |= (int)(xi >= q.x) & (int)(xi < q.z) & (int)(yi >= q.y) & (int)(yi < q.w);
    }
    flags[3 * i] = inside ? 1.f : 0.f;
}

// (one workgroup: the compaction keeps the order of the list.  1024 threads, the box bounds as one 16-byte LDS read per box and no
// early exit from the box loop so that the reads pipeline -- the first version walked 4 dependent 4-byte reads per box with a break and
// took 0.9 ms for 8000 points x 300 boxes; this one ~15 us)
__global__ __launch_bounds__(1024) void pts_box_filter_kernel(const float* __restrict__ pts, const int* __restrict__ n_in_dev, int n_in_host,
                                                              const float* __restrict__ boxes, const int* __restrict__ n_boxes_dev, int n_boxes_host,
                                                              int box_stride, int H, int W, float* __restrict__ out, int* __restrict__ out_count,
                                                              int have_flags) {
    __shared__ int4 sb[512];               // slice bounds (x1, y1, x2, y2) of up to 512 boxes
    __shared__ int wave_cnt[16];
    __shared__ int base;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int n = n_in_dev ? *n_in_dev : n_in_host;
    int nb = n_boxes_dev ? *n_boxes_dev : n_boxes_host;
    if (nb > 512) nb = 512;
    if (have_flags) nb = 0;                // (pass 1 has tested the boxes: out[3 i] != 0 = inside one)
    for (int i = t; i < nb; i += 1024) {
        const float* b = boxes + (size_t)i * box_stride;
        sb[i] = int4{py_slice_bound((int)rintf(b[0]), W), py_slice_bound((int)rintf(b[1]), H), py_slice_bound((int)rintf(b[2]), W),
                     py_slice_bound((int)rintf(b[3]), H)};
    }
    if (t == 0) base = 0;
    __syncthreads();
    for (int i0 = 0; i0 < n; i0 += 4096) {                     // four points per thread share every box read
        float x[4], y[4], c[4];
        int xi[4], yi[4];
        int inside[4];
        const int U = min(4, (n - i0 + 1023) / 1024);          // slots of this pass that hold points (uniform over the workgroup)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * 1024 + t;
            x[u] = y[u] = c[u] = 0.f;
            if (i < n) { x[u] = pts[3 * i]; y[u] = pts[3 * i + 1]; c[u] = pts[3 * i + 2]; }
            xi[u] = (int)x[u]; yi[u] = (int)y[u];
            inside[u] = (have_flags && i < n) ? (int)(out[3 * i] != 0.f) : 0;
        }
        __syncthreads();                       // (every flag of this block is read before any slot is written)
        if (U == 1) {
#pragma unroll 4
            for (int k = 0; k < nb; ++k) {                    // branch-free: bitwise combination of the four comparisons
                const int4 q = sb[k];
                inside[0] |= (int)(xi[0] >= q.x) & (int)(xi[0] < q.z) & (int)(yi[0] >= q.y) & (int)(yi[0] < q.w);
            }
        } else {
#pragma unroll 4
            for (int k = 0; k < nb; ++k) {
                const int4 q = sb[k];
#pragma unroll
                for (int u = 0; u < 4; ++u) inside[u] |= (int)(xi[u] >= q.x) & (int)(xi[u] < q.z) & (int)(yi[u] >= q.y) & (int)(yi[u] < q.w);
            }
        }
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * 1024 + t;
            const bool keep = i < n && !inside[u];
            const unsigned long long m = __ballot(keep);
            if (lane == 0) wave_cnt[wave] = __popcll(m);
            __syncthreads();
            int off = base, total = 0;
            for (int w = 0; w < 16; ++w) {
                if (w < wave) off += wave_cnt[w];
                total += wave_cnt[w];
            }
            if (keep) {
                const int slot = off + __popcll(m & ((1ull << lane) - 1ull));
                out[3 * slot] = x[u]; out[3 * slot + 1] = y[u]; out[3 * slot + 2] = c[u];
            }
            __syncthreads();
            if (t == 0) base += total;
            __syncthreads();
        }
    }
    if (t == 0) *out_count = base;
}

extern "C" int yp_pts_box_filter(const float* pts_xyc, const int* n_pts_dev, int n_pts, const float* boxes, const int* n_boxes_dev, int n_boxes,
                                 int box_stride, int H, int W, float* out_xyc, int* out_count, void* stream) {
    YP_REQUIRE(pts_xyc && out_xyc && out_count && (boxes || (n_boxes == 0 && !n_boxes_dev)) && n_pts >= 0 && n_boxes >= 0 && box_stride >= 4 && H > 0 && W > 0,
               "yp_pts_box_filter: bad arguments");
    // (more than a few hundred point x box tests per thread: pass 1 on all CUs, then the order-keeping compaction)
    const int two_pass = n_pts >= 1024 && (n_boxes_dev || n_boxes >= 16) && out_xyc != pts_xyc;
    if (two_pass)
        pts_box_flags_kernel<<<(n_pts + 255) / 256, 256, 0, (hipStream_t)stream>>>(pts_xyc, n_pts_dev, n_pts, boxes, n_boxes_dev, n_boxes, box_stride, H, W, out_xyc);
    pts_box_filter_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(pts_xyc, n_pts_dev, n_pts, boxes, n_boxes_dev, n_boxes, box_stride, H, W, out_xyc, out_count,
                                                               two_pass);
    YP_CHECK_HIP(hipGetLastError());
    return YP_OK;
}

// ---------------------------------------------------------------------------------------------
// Homography adaptation, the aggregation step (reference export_homography.py:94-96,143-145): N views of one image were pushed
// through the network; view v's heat map (times its valid mask) and the mask itself are warped back to the base frame with
// warp_image_batch (utils/utils.py:333-376: normalised [-1,1] grid through inv_homographies[v], bilinear grid_sample,
// align_corners=True, zero padding), summed over the views and divided.  The reference materialises 2 x N warped full-resolution
// maps; here one thread owns one base-frame pixel and walks the N views (8 gathered taps per view), nothing is written but the
// aggregated map.  0 / 0 (a pixel no view covers) stays NaN as in the reference; the >= threshold of the decode rejects it.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float linspace_m1_p1(int i, int n) {      // torch.linspace(-1, 1, n)[i] (symmetric fill)
    if (n == 1) return -1.0f;
    const float step = 2.0f / (float)(n - 1);
    return i < n / 2 ? -1.0f + step * (float)i : 1.0f - step * (float)(n - 1 - i);
}

__global__ __launch_bounds__(256) void homo_combine_kernel(const float* __restrict__ heat, const float* __restrict__ mask, const float* __restrict__ invh, int N,
                                                           int H, int W, float* __restrict__ out, float* __restrict__ out_cover) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const float gx = linspace_m1_p1(x, W), gy = linspace_m1_p1(y, H);
    const size_t plane = (size_t)H * W;
    float sh = 0.f, sm = 0.f;
    for (int v = 0; v < N; ++v) {
        const float* h = invh + v * 9;
        const float X = h[0] * gx + h[1] * gy + h[2], Y = h[3] * gx + h[4] * gy + h[5], Z = h[6] * gx + h[7] * gy + h[8];
        const float ix = ((X / Z + 1.0f) / 2.0f) * (float)(W - 1), iy = ((Y / Z + 1.0f) / 2.0f) * (float)(H - 1);
        const float fx0 = floorf(ix), fy0 = floorf(iy);
        if (!(fx0 >= -1.0f && fx0 <= (float)W && fy0 >= -1.0f && fy0 <= (float)H)) continue;     // all four taps outside (or NaN)
        const int x0 = (int)fx0, y0 = (int)fy0, x1 = x0 + 1, y1 = y0 + 1;
        const float wx1 = ix - fx0, wx0 = (fx0 + 1.0f) - ix, wy1 = iy - fy0, wy0 = (fy0 + 1.0f) - iy;
        const float* hp = heat + v * plane;
        const float* mp = mask + v * plane;
        float ah = 0.f, am = 0.f;
        const bool okx0 = x0 >= 0 && x0 < W, okx1 = x1 >= 0 && x1 < W, oky0 = y0 >= 0 && y0 < H, oky1 = y1 >= 0 && y1 < H;
        if (oky0 && okx0) { const float m = mp[(size_t)y0 * W + x0], w = wx0 * wy0; ah += hp[(size_t)y0 * W + x0] * m * w; am += m * w; }
        if (oky0 && okx1) { const float m = mp[(size_t)y0 * W + x1], w = wx1 * wy0; ah += hp[(size_t)y0 * W + x1] * m * w; am += m * w; }
        if (oky1 && okx0) { const float m = mp[(size_t)y1 * W + x0], w = wx0 * wy1; ah += hp[(size_t)y1 * W + x0] * m * w; am += m * w; }
        if (oky1 && okx1) { const float m = mp[(size_t)y1 * W + x1], w = wx1 * wy1; ah += hp[(size_t)y1 * W + x1] * m * w; am += m * w; }
        sh += ah;
        sm += am;
    }
    out[(size_t)y * W + x] = sh / sm;
    if (out_cover) out_cover[(size_t)y * W + x] = sm;
}

extern "C" int yp_homo_combine(const float* heat, const float* mask, const float* inv_homographies, int N, int H, int W, float* out, float* out_cover,
                               void* stream) {
    YP_REQUIRE(heat && mask && inv_homographies && out && N > 0 && H > 0 && W > 0, "yp_homo_combine: bad arguments");
    dim3 grid((W + 63) / 64, (H + 3) / 4);
    homo_combine_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(heat, mask, inv_homographies, N, H, W, out, out_cover);
    YP_CHECK_HIP(hipGetLastError());
    return YP_OK;
}

extern "C" int yp_desc_sample(const float* desc, int D, int Hc, int Wc, int64_t sc, int64_t sy, int64_t sx, const float* pts_xy, int N,
                              int cell, float* out, void* stream) {
    YP_REQUIRE(desc && out && D > 0 && Hc > 0 && Wc > 0 && cell > 0 && N >= 0, "yp_desc_sample: bad arguments");
    if (N == 0) return YP_OK;
    YP_REQUIRE(pts_xy != nullptr, "yp_desc_sample: null points");
    desc_sample_kernel<<<grid_for((size_t)N * 64, 256), 256, 0, (hipStream_t)stream>>>(desc, D, Hc, Wc, sc, sy, sx, pts_xy, N, cell, out);
    YP_CHECK_HIP(hipGetLastError());
    return YP_OK;
}

extern "C" size_t yp_mnn_workspace_bytes(int N1, int N2) { return align_up((size_t)N1 * 8, 256) + align_up((size_t)N2 * 8, 256); }

extern "C" int yp_mnn_match(const float* d1, int N1, const float* d2, int N2, int D, float nn_thresh, float* out_match,
                            int32_t* out_count, int max_out, void* workspace, size_t workspace_bytes, void* stream) {
    YP_REQUIRE(out_count != nullptr, "yp_mnn_match: null out_count");
    YP_REQUIRE(N1 >= 0 && N2 >= 0 && D > 0 && max_out >= 0, "yp_mnn_match: bad dims");
    YP_REQUIRE(nn_thresh > 0.f, "yp_mnn_match: nn_thresh must be > 0");     // models/model_wrap.py:452
    hipStream_t st = (hipStream_t)stream;
    if (N1 == 0 || N2 == 0) {   // reference returns zeros((3,0))
        YP_CHECK_HIP(hipMemsetAsync(out_count, 0, 4, st));
        return YP_OK;
    }
    YP_REQUIRE(d1 && d2 && out_match && workspace, "yp_mnn_match: null pointer");
    if (workspace_bytes < yp_mnn_workspace_bytes(N1, N2)) {
        yp_set_error("yp_mnn_match: workspace too small");
        return YP_ERR_WORKSPACE;
    }
    u64* rowmin = (u64*)workspace;
    u64* colmin = (u64*)((char*)workspace + align_up((size_t)N1 * 8, 256));
    const long nfill = (long)(align_up((size_t)N1 * 8, 256) / 8) + N2;
    fill_u64_kernel<<<grid_for(nfill, 256), 256, 0, st>>>(rowmin, nfill, ~0ull);
    dim3 grid((N2 + 63) / 64, (N1 + 63) / 64);
    mnn_tile_kernel<<<grid, 256, 0, st>>>(d1, N1, d2, N2, D, rowmin, colmin);
    mnn_select_kernel<<<1, 1024, 0, st>>>(rowmin, colmin, N1, nn_thresh, out_match, out_count, max_out);
    YP_CHECK_HIP(hipGetLastError());
    return YP_OK;
}
