// Weight gradient of a convolution (1x1 stride 1, or 3x3 with pad 1 and stride 1 | 2) straight from the NHWC tensors:
//
//   dW[ci][r][s][co] += sum over (b, y, x) of  x[b, y*st+r-p, x*st+s-p, ci] * dy[b, y, x, co]
//
// The reduction runs over PIXELS, which is the strided axis of both NHWC operands, so neither matches the
// MFMA fragment layout (8 consecutive k per lane).  Instead of materialising pixel-major copies in HBM, both
// tiles are DMA'd into LDS as they lie (rows = pixels, 16-channel blocks of 32 bytes) and the fragments are
// read with gfx950's transposing LDS read: ds_read_b64_tr_b16 hands lane i of each 16-lane group column i of a
// 4 x 16 block whose four rows are addressed by the group's lanes 4j..4j+3 -- i.e. 4 consecutive pixels of one
// channel.  Two such reads give the 8 k values of an MFMA 16x16x32 operand; the same pixel order is used for
// x and dy, so the k permutation inside a fragment cancels in the dot product.
//
// A workgroup owns a [64 ci x 64 co] block of dW for every tap and walks a strided subset of the pixel tiles
// (128 consecutive pixels for 1x1; an 8 x 16 output patch + halo for 3x3 stride 1, 4 x 16 for stride 2), double buffered through
// LDS-DMA; partial sums go to the zero-initialised fp32 dW with hardware fp32 atomics.
//
// replaces: autograd's conv2d weight gradient for reference models/common.py:22-34 (loss.backward(), train.py:245).
#include "yp_internal.h"

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __attribute__((aligned(16))) unsigned int wg_zero16[4] = {0u, 0u, 0u, 0u};

struct WgradArgs {
    const char* x;
    const char* dy;
    float* dw;
    int x_cs, x_co, x_ups, x_H, x_W;      // x buffer: channel stride / offset, 2x-nearest-upsample flag, STORED dims
    int dy_cs, dy_co;
    int B, H, W;                           // output map; the logical input map is (H*stride, W*stride)
    int Cj, Cout_pad;
    int n_co_blk;
    int tiles_x, tiles_y, ntiles, M;
    int skip_store;                        // always 0 in the product (round-3 probe: time the reduction without the final atomics
    int blk;                               // channels per (ci x co) block side: 64, or 128 (1x1 filters with >= 128 channels on both sides)
    int g_blk0, g_nblk, g_split;           // grouped launch: first flat workgroup of this entry, its (ci x co) blocks and pixel split
    float* part;                           // deterministic mode: partial slabs [g_split][Cj][taps][Cout_pad] (plain stores, folded in order by
    int f_chunk0, f_chunks;                // wgrad_fold_kernel: this entry's first 1024-element chunk and chunk count); nullptr: fp32 atomics into dw
    const float* sx;                       // 8-bit operands (wgrad_body8: x e4m3, dy e5m2): device scalars, real = stored * scale; the sums are
    const float* sdy;                      // multiplied by *sx * *sdy before they leave the workgroup
};

__device__ __forceinline__ void wg_glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

template <int DT> __device__ __forceinline__ f32x4 wg_mma(s16x8 a, s16x8 b, f32x4 c) {
    if constexpr (DT == YP_F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

__device__ __forceinline__ s16x8 wg_tr8(const char* lds_lo, const char* lds_hi) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)lds_lo);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)lds_hi);
    return s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// NB = 16-channel fragments per wave and side: the workgroup (2 x 2 waves) owns a [32 NB ci x 32 NB co] block of dW.  NB = 4 (1x1 only):
// 128 x 128 blocks, 64 x 64 per wave -- every fragment read from LDS feeds four MFMAs instead of two and a 128-pixel tile carries 1024 clk
// of MFMA work per wave between its two barriers instead of 256 (the 64 x 64 blocks ran every 1x1 shape at ~210 TFLOP/s, barrier-bound).
template <int DT, int TAPS, int ST, int NB = 2>
__device__ __forceinline__ void wgrad_body(const WgradArgs& a, const int bx, const int first, const int step) {
    constexpr int KS = TAPS == 9 ? 3 : 1;
    constexpr int TH = (TAPS == 9 && ST == 2) ? 4 : 8;     // output rows of a 3x3 tile (16 columns)
    constexpr int HP = 16 * ST + (ST == 1 ? 2 : 1);        // halo pitch: 18 | 33 pixels
    constexpr int HROWS = (TH * ST + (ST == 1 ? 2 : 1)) * HP;   // 10 x 18 | 9 x 33
    constexpr int NPIX = TH * 16;                          // pixels (= k extent) of one tile: 128 | 64
    constexpr int KK = NPIX / 32;                          // MFMA k steps per tile
    constexpr int XR = TAPS == 9 ? (HROWS + 31) / 32 * 32 : 128;   // LDS rows (pixels) of the x image per 16-channel block
    constexpr int XI = XR / 32;                            // DMA instructions per channel block (32 rows x 32 B each)
    constexpr int DYI = NPIX / 32;
    constexpr int NCB = 2 * NB;                            // 16-channel blocks per operand tile
    constexpr int XBYTES = NCB * XR * 32, DYBYTES = NCB * NPIX * 32, STAGE = XBYTES + DYBYTES;
    constexpr int XN = XI * NB / 2, DYN = DYI * NB / 2;    // DMA instructions per wave and tile (x | dy)
    constexpr int NDMA = XN + DYN;
    static_assert(TAPS == 9 || ST == 1, "1x1: stride 1 only");
    static_assert(NB == 2 || TAPS == 1, "128 x 128 blocks: 1x1 filters only (a 3x3 block holds 9 accumulator sets)");

    extern __shared__ __attribute__((aligned(1024))) char wsm[];      // 2 stages of [x: NCB cb][XR][32 B] [dy: NCB cb][NPIX][32 B]
    const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) char*)wsm);

    const int ci0 = (bx / a.n_co_blk) * (32 * NB), co0 = (bx % a.n_co_blk) * (32 * NB);
    const int t = threadIdx.x, l = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int half = l & 1, prow = l >> 1;                 // DMA: this lane moves channels [half*8, +8) of row prow of its 32-row slab

    auto issue = [&](int tile, int stage) {
        const unsigned sb = lds0 + stage * STAGE;
        int b, y0, x0;
        if constexpr (TAPS == 9) {
            int r_ = tile;
            const int tx = r_ % a.tiles_x; r_ /= a.tiles_x;
            const int ty = r_ % a.tiles_y;
            b = r_ / a.tiles_y; y0 = ty * TH; x0 = tx * 16;
        } else { b = 0; y0 = 0; x0 = 0; }
        // ---- x rows
#pragma unroll
        for (int i = 0; i < XN; ++i) {
            const int q = wave + 4 * i;                    // q in [0, NCB*XI): channel block q / XI, row slab q % XI
            const int cb = q / XI, rb = q - cb * XI;
            const int row = rb * 32 + prow;
            const int ch = ci0 + cb * 16 + half * 8;
            bool ok = ch < a.Cj;
            int bb, iy, ix;
            if constexpr (TAPS == 9) {
                const int hy = row / HP, hx = row - hy * HP;
                bb = b; iy = y0 * ST - 1 + hy; ix = x0 * ST - 1 + hx;
                ok = ok && row < HROWS && (unsigned)iy < (unsigned)(a.H * ST) && (unsigned)ix < (unsigned)(a.W * ST);
            } else {
                const int m = tile * 128 + row;
                ok = ok && m < a.M;
                bb = m / (a.H * a.W);
                const int rem = m - bb * (a.H * a.W);
                iy = rem / a.W; ix = rem - iy * a.W;
            }
            const long pix = ((long)bb * a.x_H + (iy >> a.x_ups)) * a.x_W + (ix >> a.x_ups);
            const char* src = a.x + (pix * a.x_cs + a.x_co + ch) * 2;
            wg_glds16(ok ? (const void*)src : (const void*)wg_zero16, sb + (cb * XR + rb * 32) * 32);
        }
        // ---- dy rows
#pragma unroll
        for (int i = 0; i < DYN; ++i) {
            const int q = wave + 4 * i;                    // q in [0, NCB*DYI)
            const int cb = q / DYI, rb = q - cb * DYI;
            const int row = rb * 32 + prow;
            const int ch = co0 + cb * 16 + half * 8;
            bool ok = ch < a.Cout_pad;
            long pix;
            if constexpr (TAPS == 9) {
                const int oy = y0 + (row >> 4), ox = x0 + (row & 15);
                ok = ok && oy < a.H && ox < a.W;
                pix = ((long)b * a.H + oy) * a.W + ox;
            } else {
                pix = (long)tile * 128 + row;
                ok = ok && pix < a.M;
            }
            const char* src = a.dy + (pix * a.dy_cs + a.dy_co + ch) * 2;
            wg_glds16(ok ? (const void*)src : (const void*)wg_zero16, sb + XBYTES + (cb * NPIX + rb * 32) * 32);
        }
    };

    // ---- fragment read addressing (see the file header): lane i of group g addresses row j = i/4 of its 4-row block,
    // channels 4*(i%4)..+3; read h covers pixels g*8 + 4*(h ^ (g&1)) + j of a 32-pixel k step (groups alternate the
    // order of their two 4-row halves so that one read's four row blocks spread over the LDS banks)
    const int li = l & 15, g = l >> 4, j = li >> 2, q4 = li & 3;
    int xoff[2], yoff[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int k32 = g * 8 + ((h ^ (g & 1)) * 4) + j;
        if constexpr (TAPS == 9) xoff[h] = (((k32 >> 4) * ST * HP) + (k32 & 15) * ST) * 32 + q4 * 8;
        else xoff[h] = k32 * 32 + q4 * 8;
        yoff[h] = k32 * 32 + q4 * 8;
    }
    const int wci = wave & 1, wco = wave >> 1;             // wave tile: ci [wci*16*NB, +16*NB), co [wco*16*NB, +16*NB)

    f32x4 acc[TAPS][NB][NB];
#pragma unroll
    for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
        for (int fa = 0; fa < NB; ++fa)
#pragma unroll
            for (int fb = 0; fb < NB; ++fb) acc[tp][fa][fb] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (first < a.ntiles) issue(first, 0);
    int it = 0;
    for (int tile = first; tile < a.ntiles; tile += step, ++it) {
        const bool more = tile + step < a.ntiles;
        if (more) issue(tile + step, (it + 1) & 1);        // that stage was last read two tiles ago: the barrier below covers it
        if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const char* xs = wsm + (it & 1) * STAGE;
        const char* ys = xs + XBYTES;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            s16x8 yf[NB];
#pragma unroll
            for (int fb = 0; fb < NB; ++fb) {
                const char* base = ys + ((wco * NB + fb) * NPIX + kk * 32) * 32;
                yf[fb] = wg_tr8(base + yoff[0], base + yoff[1]);
            }
#pragma unroll
            for (int tp = 0; tp < TAPS; ++tp) {
                const int r = tp / KS, s = tp - r * KS;
                s16x8 xf[NB];
#pragma unroll
                for (int fa = 0; fa < NB; ++fa) {
                    const char* base = xs + (wci * NB + fa) * XR * 32 + (TAPS == 9 ? (kk * 2 * ST * HP + r * HP + s) * 32 : kk * 32 * 32);
                    xf[fa] = wg_tr8(base + xoff[0], base + xoff[1]);
                }
#pragma unroll
                for (int fa = 0; fa < NB; ++fa)
#pragma unroll
                    for (int fb = 0; fb < NB; ++fb) acc[tp][fa][fb] = wg_mma<DT>(xf[fa], yf[fb], acc[tp][fa][fb]);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the compiler may sink the last MFMAs, and the wait for their operands, below a barrier)
        __builtin_amdgcn_s_barrier();                      // everyone is done reading this stage before it is refilled
    }

    // ---- partial sums -> dW[ci][r][s][co]; accumulator lane (p, g): rows (ci) 4g..4g+3, column (co) p.
    // Deterministic mode: this pixel slice's own slab, plain stores (every element of a slab has exactly one writer); the slabs are summed
    // in slice order by wgrad_fold_kernel.  Otherwise fp32 atomics into the zero-initialised dW (order of arrival: run-to-run differences
    // at the 1e-4 level in 16-bit training).
    if (a.part != nullptr) {
        float* slab = a.part + (size_t)first * a.Cj * TAPS * a.Cout_pad;
#pragma unroll
        for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
            for (int fa = 0; fa < NB; ++fa)
#pragma unroll
                for (int fb = 0; fb < NB; ++fb) {
                    const int co = co0 + (wco * NB + fb) * 16 + li;
                    if (co >= a.Cout_pad) continue;
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        const int ci = ci0 + (wci * NB + fa) * 16 + 4 * g + jj;
                        if (ci < a.Cj) slab[((size_t)ci * TAPS + tp) * a.Cout_pad + co] = acc[tp][fa][fb][jj];
                    }
                }
        return;
    }
#pragma unroll
    for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
        for (int fa = 0; fa < NB; ++fa)
#pragma unroll
            for (int fb = 0; fb < NB; ++fb) {
                const int co = co0 + (wco * NB + fb) * 16 + li;
                if (co >= a.Cout_pad) continue;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int ci = ci0 + (wci * NB + fa) * 16 + 4 * g + jj;
                    if (ci < a.Cj && !a.skip_store) atomicAdd(a.dw + ((size_t)ci * TAPS + tp) * a.Cout_pad + co, acc[tp][fa][fb][jj]);
                }
            }
}

// ---------------------------------------------------------------------------------------------------------------------------
// The same reduction on 8-BIT operands (BASELINE.json configs[4]): x = the e4m3 twin of the layer's input, dy = the e5m2 twin of its output
// gradient -- the bytes the forward / dgrad convolutions of the same layer multiply (csrc/fp8.hip: per-tensor scales, real = stored * scale).
// Half the bytes from HBM and through LDS; the MFMA rate is the 16-bit one (v_mfma_f32_16x16x32_fp8_bf8), which these kernels do not reach.
//   LDS: 16-channel blocks of 16-BYTE pixel rows; one DMA instruction moves 64 pixel rows of one channel block (lane l = row l).
//   Fragments: ds_read_b64_tr_b8 -- in each 16-lane group lane i supplies the address of 8 contiguous bytes of row i/2 (bytes 8 (i%2) ..) of
//   an 8-row block and receives column i of those 8 rows (semantics pinned by tools/probe/tr8_probe.hip): ONE read = the 8 k values
//   (consecutive pixels of channel i) of an MFMA 16x16x32 operand; group g takes pixels 8 g .. 8 g + 7 of the 32-pixel k step for both
//   operands, so the k order is the same on both sides.
// Accumulation in fp32 as before; the per-tensor scales multiply the sums once, in the epilogue.
typedef int i32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ long wg_tr8b(const char* lds) {
    const i32x2 v = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) i32x2*)lds);
    return ((long)(unsigned)v[1] << 32) | (unsigned)v[0];
}

template <int TAPS, int ST, int NB = 2>
__device__ __forceinline__ void wgrad_body8(const WgradArgs& a, const int bx, const int first, const int step) {
    constexpr int KS = TAPS == 9 ? 3 : 1;
    constexpr int TH = (TAPS == 9 && ST == 2) ? 4 : 8;
    constexpr int HP = 16 * ST + (ST == 1 ? 2 : 1);
    constexpr int HROWS = (TH * ST + (ST == 1 ? 2 : 1)) * HP;
    constexpr int NPIX = TH * 16;
    constexpr int KK = NPIX / 32;
    constexpr int XR = TAPS == 9 ? (HROWS + 63) / 64 * 64 : 128;   // LDS rows (pixels) of the x image per 16-channel block: 192 | 320 | 128
    constexpr int XI = XR / 64;                            // DMA instructions per channel block (64 rows x 16 B each)
    constexpr int DYI = NPIX / 64;
    constexpr int NCB = 2 * NB;
    constexpr int XBYTES = NCB * XR * 16, DYBYTES = NCB * NPIX * 16, STAGE = XBYTES + DYBYTES;
    constexpr int XN = (XI * NCB + 3) / 4, DYN = (DYI * NCB + 3) / 4;    // DMA instructions per wave and tile (x | dy)
    constexpr int NDMA = XN + DYN;
    static_assert(TAPS == 9 || ST == 1, "1x1: stride 1 only");
    static_assert(NB == 2 || TAPS == 1, "128 x 128 blocks: 1x1 filters only");
    static_assert(DYI >= 1 && (DYI * NCB) % 4 == 0, "dy rows: whole DMA instructions per wave");

    extern __shared__ __attribute__((aligned(1024))) char wsm[];      // 2 stages of [x: NCB cb][XR][16 B] [dy: NCB cb][NPIX][16 B]
    const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) char*)wsm);

    const int ci0 = (bx / a.n_co_blk) * (32 * NB), co0 = (bx % a.n_co_blk) * (32 * NB);
    const int t = threadIdx.x, l = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);

    auto issue = [&](int tile, int stage) {
        const unsigned sb = lds0 + stage * STAGE;
        int b, y0, x0;
        if constexpr (TAPS == 9) {
            int r_ = tile;
            const int tx = r_ % a.tiles_x; r_ /= a.tiles_x;
            const int ty = r_ % a.tiles_y;
            b = r_ / a.tiles_y; y0 = ty * TH; x0 = tx * 16;
        } else { b = 0; y0 = 0; x0 = 0; }
#pragma unroll
        for (int i = 0; i < XN; ++i) {
            int q = wave + 4 * i;                          // q in [0, NCB*XI): channel block q / XI, row slab q % XI
            if (q > NCB * XI - 1) q = NCB * XI - 1;         // (surplus instructions re-fetch the last slab: same bytes)
            const int cb = q / XI, rb = q - cb * XI;
            const int row = rb * 64 + l;
            const int ch = ci0 + cb * 16;
            bool ok = ch < a.Cj;
            int bb, iy, ix;
            if constexpr (TAPS == 9) {
                const int hy = row / HP, hx = row - hy * HP;
                bb = b; iy = y0 * ST - 1 + hy; ix = x0 * ST - 1 + hx;
                ok = ok && row < HROWS && (unsigned)iy < (unsigned)(a.H * ST) && (unsigned)ix < (unsigned)(a.W * ST);
            } else {
                const int m = tile * 128 + row;
                ok = ok && m < a.M;
                bb = m / (a.H * a.W);
                const int rem = m - bb * (a.H * a.W);
                iy = rem / a.W; ix = rem - iy * a.W;
            }
            const long pix = ((long)bb * a.x_H + (iy >> a.x_ups)) * a.x_W + (ix >> a.x_ups);
            const char* src = a.x + (pix * a.x_cs + a.x_co + ch);
            wg_glds16(ok ? (const void*)src : (const void*)wg_zero16, sb + (cb * XR + rb * 64) * 16);
        }
#pragma unroll
        for (int i = 0; i < DYN; ++i) {
            const int q = wave + 4 * i;                    // q in [0, NCB*DYI)
            const int cb = q / DYI, rb = q - cb * DYI;
            const int row = rb * 64 + l;
            const int ch = co0 + cb * 16;
            bool ok = ch < a.Cout_pad;
            long pix;
            if constexpr (TAPS == 9) {
                const int oy = y0 + (row >> 4), ox = x0 + (row & 15);
                ok = ok && oy < a.H && ox < a.W;
                pix = ((long)b * a.H + oy) * a.W + ox;
            } else {
                pix = (long)tile * 128 + row;
                ok = ok && pix < a.M;
            }
            const char* src = a.dy + (pix * a.dy_cs + a.dy_co + ch);
            wg_glds16(ok ? (const void*)src : (const void*)wg_zero16, sb + XBYTES + (cb * NPIX + rb * 64) * 16);
        }
    };

    // ---- fragment read addressing: lane i of group g addresses row (pixel) 8 g + i/2 of the 32-pixel k step, bytes 8 (i%2) .. of its 16
    const int li = l & 15, g = l >> 4;
    const int k32 = g * 8 + (li >> 1);
    int xoff, yoff;
    if constexpr (TAPS == 9) xoff = (((k32 >> 4) * ST * HP) + (k32 & 15) * ST) * 16 + (li & 1) * 8;
    else xoff = k32 * 16 + (li & 1) * 8;
    yoff = k32 * 16 + (li & 1) * 8;
    const int wci = wave & 1, wco = wave >> 1;

    f32x4 acc[TAPS][NB][NB];
#pragma unroll
    for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
        for (int fa = 0; fa < NB; ++fa)
#pragma unroll
            for (int fb = 0; fb < NB; ++fb) acc[tp][fa][fb] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (first < a.ntiles) issue(first, 0);
    int it = 0;
    for (int tile = first; tile < a.ntiles; tile += step, ++it) {
        const bool more = tile + step < a.ntiles;
        if (more) issue(tile + step, (it + 1) & 1);
        if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const char* xs = wsm + (it & 1) * STAGE;
        const char* ys = xs + XBYTES;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            long yf[NB];
#pragma unroll
            for (int fb = 0; fb < NB; ++fb) yf[fb] = wg_tr8b(ys + ((wco * NB + fb) * NPIX + kk * 32) * 16 + yoff);
#pragma unroll
            for (int tp = 0; tp < TAPS; ++tp) {
                const int r = tp / KS, s = tp - r * KS;
                long xf[NB];
#pragma unroll
                for (int fa = 0; fa < NB; ++fa)
                    xf[fa] = wg_tr8b(xs + (wci * NB + fa) * XR * 16 + (TAPS == 9 ? (kk * 2 * ST * HP + r * HP + s) * 16 : kk * 32 * 16) + xoff);
#pragma unroll
                for (int fa = 0; fa < NB; ++fa)
#pragma unroll
                    for (int fb = 0; fb < NB; ++fb) acc[tp][fa][fb] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_bf8(xf[fa], yf[fb], acc[tp][fa][fb], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }

    const float sc = a.sx[0] * a.sdy[0];
    if (a.part != nullptr) {
        float* slab = a.part + (size_t)first * a.Cj * TAPS * a.Cout_pad;
#pragma unroll
        for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
            for (int fa = 0; fa < NB; ++fa)
#pragma unroll
                for (int fb = 0; fb < NB; ++fb) {
                    const int co = co0 + (wco * NB + fb) * 16 + li;
                    if (co >= a.Cout_pad) continue;
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        const int ci = ci0 + (wci * NB + fa) * 16 + 4 * g + jj;
                        if (ci < a.Cj) slab[((size_t)ci * TAPS + tp) * a.Cout_pad + co] = acc[tp][fa][fb][jj] * sc;
                    }
                }
        return;
    }
#pragma unroll
    for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
        for (int fa = 0; fa < NB; ++fa)
#pragma unroll
            for (int fb = 0; fb < NB; ++fb) {
                const int co = co0 + (wco * NB + fb) * 16 + li;
                if (co >= a.Cout_pad) continue;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int ci = ci0 + (wci * NB + fa) * 16 + 4 * g + jj;
                    if (ci < a.Cj && !a.skip_store) atomicAdd(a.dw + ((size_t)ci * TAPS + tp) * a.Cout_pad + co, acc[tp][fa][fb][jj] * sc);
                }
            }
}

template <int TAPS, int ST, int NB = 2>
__global__ __launch_bounds__(256) void wgrad8_kernel(const WgradArgs a) {
    wgrad_body8<TAPS, ST, NB>(a, blockIdx.x, blockIdx.y, gridDim.y);
}

template <int TAPS, int ST, int NB = 2>
__global__ __launch_bounds__(256) void wgrad8_group_kernel(const WgradArgs* __restrict__ table, int n_entries) {
    const int bid = blockIdx.x;
    int e = 0;
    for (int lo = 0, hi = n_entries - 1; lo <= hi;) {
        const int mid = (lo + hi) >> 1;
        if (table[mid].g_blk0 <= bid) { e = mid; lo = mid + 1; } else hi = mid - 1;
    }
    const WgradArgs a = table[e];
    const int local = bid - a.g_blk0;
    wgrad_body8<TAPS, ST, NB>(a, local % a.g_nblk, local / a.g_nblk, a.g_split);
}

template <int DT, int TAPS, int ST, int NB = 2>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradArgs a) {
    wgrad_body<DT, TAPS, ST, NB>(a, blockIdx.x, blockIdx.y, gridDim.y);
}

// All weight gradients of one filter class (1x1 | 3x3 stride 1 | 3x3 stride 2) of a backward pass in ONE launch: a device table of
// argument sets, flat workgroup ids mapped to (entry, channel block, pixel-split slice).  81 + 20 + 12 launches per training step
// become 3 per backward pass; the small 1x1 gradients no longer leave most of the chip idle between launches.
template <int DT, int TAPS, int ST, int NB = 2>
__global__ __launch_bounds__(256) void wgrad_group_kernel(const WgradArgs* __restrict__ table, int n_entries) {
    const int bid = blockIdx.x;                                   // (everything below depends on blockIdx only: scalar loads, SGPR arguments)
    int e = 0;
    for (int lo = 0, hi = n_entries - 1; lo <= hi;) {             // last entry with g_blk0 <= bid
        const int mid = (lo + hi) >> 1;
        if (table[mid].g_blk0 <= bid) { e = mid; lo = mid + 1; } else hi = mid - 1;
    }
    const WgradArgs a = table[e];
    const int local = bid - a.g_blk0;
    wgrad_body<DT, TAPS, ST, NB>(a, local % a.g_nblk, local / a.g_nblk, a.g_split);
}

// dW = slab[0] + slab[1] + ... + slab[split-1], in that order, for every entry of a grouped launch (1024 elements per workgroup).
__global__ __launch_bounds__(256) void wgrad_fold_kernel(const WgradArgs* __restrict__ table, int n_entries, int taps) {
    const int bid = blockIdx.x;
    int e = 0;
    for (int lo = 0, hi = n_entries - 1; lo <= hi;) {
        const int mid = (lo + hi) >> 1;
        if (table[mid].f_chunk0 <= bid) { e = mid; lo = mid + 1; } else hi = mid - 1;
    }
    const WgradArgs a = table[e];
    const size_t n = (size_t)a.Cj * taps * a.Cout_pad;            // (a multiple of 8: Cout_pad is)
    const size_t i = ((size_t)(bid - a.f_chunk0) * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    // (eight slab reads in flight per thread; the additions keep the slab order)
    const float* p = a.part + i;
    f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
    int s = 1;
    for (; s + 8 <= a.g_split; s += 8) {
        f32x4 u[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) u[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + (size_t)(s + j) * n));
#pragma unroll
        for (int j = 0; j < 8; ++j) { v[0] += u[j][0]; v[1] += u[j][1]; v[2] += u[j][2]; v[3] += u[j][3]; }
    }
    for (; s < a.g_split; ++s) {
        const f32x4 u = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + (size_t)s * n));
        v[0] += u[0]; v[1] += u[1]; v[2] += u[2]; v[3] += u[3];
    }
    *reinterpret_cast<f32x4*>(a.dw + i) = v;
}

template <int DT, int TAPS, int ST, int NB = 2>
hipError_t launch_wgrad(const WgradArgs& a, dim3 grid, hipStream_t st) {
    constexpr int TH = (TAPS == 9 && ST == 2) ? 4 : 8;
    constexpr int HP = 16 * ST + (ST == 1 ? 2 : 1);
    constexpr int HROWS = (TH * ST + (ST == 1 ? 2 : 1)) * HP;
    constexpr int XR = TAPS == 9 ? (HROWS + 31) / 32 * 32 : 128;
    constexpr size_t lds = (size_t)2 * (2 * NB * XR * 32 + 2 * NB * TH * 16 * 32);
    auto kern = wgrad_kernel<DT, TAPS, ST, NB>;
    static YpLdsAttr attr;        // per instantiation, per device
    if (hipError_t e = yp_set_max_lds(attr, (const void*)kern, (int)lds); e != hipSuccess) return e;
    kern<<<grid, 256, lds, st>>>(a);
    return hipGetLastError();
}

template <int TAPS, int ST, int NB = 2>
constexpr size_t wgrad8_lds() {
    constexpr int TH = (TAPS == 9 && ST == 2) ? 4 : 8;
    constexpr int HP = 16 * ST + (ST == 1 ? 2 : 1);
    constexpr int HROWS = (TH * ST + (ST == 1 ? 2 : 1)) * HP;
    constexpr int XR = TAPS == 9 ? (HROWS + 63) / 64 * 64 : 128;
    return (size_t)2 * (2 * NB * XR * 16 + 2 * NB * TH * 16 * 16);
}

template <int TAPS, int ST, int NB = 2>
hipError_t launch_wgrad8(const WgradArgs& a, dim3 grid, hipStream_t st) {
    constexpr size_t lds = wgrad8_lds<TAPS, ST, NB>();
    auto kern = wgrad8_kernel<TAPS, ST, NB>;
    static YpLdsAttr attr;        // per instantiation, per device
    if (hipError_t e = yp_set_max_lds(attr, (const void*)kern, (int)lds); e != hipSuccess) return e;
    kern<<<grid, 256, lds, st>>>(a);
    return hipGetLastError();
}

static hipError_t dispatch_wgrad8(int k, int stride, const WgradArgs& a, dim3 grid, hipStream_t st) {
    if (k == 1) return a.blk == 128 ? launch_wgrad8<1, 1, 4>(a, grid, st) : launch_wgrad8<1, 1>(a, grid, st);
    return stride == 2 ? launch_wgrad8<9, 2>(a, grid, st) : launch_wgrad8<9, 1>(a, grid, st);
}

template <int DT>
hipError_t dispatch_wgrad(int k, int stride, const WgradArgs& a, dim3 grid, hipStream_t st) {
    if (k == 1) return a.blk == 128 ? launch_wgrad<DT, 1, 1, 4>(a, grid, st) : launch_wgrad<DT, 1, 1>(a, grid, st);
    return stride == 2 ? launch_wgrad<DT, 9, 2>(a, grid, st) : launch_wgrad<DT, 9, 1>(a, grid, st);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Weight gradient of the STEM (reference models/YOLOPoint.py:156: Conv(3, c1, k=6, s=2, p=2)) straight from the packed image
// [B][H][W][4] (three channels + a zero one, 8 bytes per pixel) and dy [B][H/2][W/2][Cout_pad]:
//
//   dW[ci][r][s][co] = sum over (b, y, x) of  img[b, 2y+r-2, 2x+s-2, ci] * dy[b, y, x, co]
//
// Same transposing-LDS-read scheme as above with the image as the 16-wide operand: for one output pixel and one filter row r, the four
// pixels 2x+4h-2 .. 2x+4h+1 (h = 0 | 1) are 32 contiguous bytes = 16 "channels" (tap column 4h + 0..3, image channel 0..3) of one
// transposing read row.  12 operand rows (6 filter rows x 2 halves; the half h = 1 carries two tap columns that do not exist, 6 and 7,
// which are computed and dropped) x Cout_pad/16 dy blocks per 32-pixel k step; wave w owns operand rows 3w..3w+2.  A workgroup walks
// 8 x 16 output patches (20 x 36 image pixels, DMA'd as they lie) and leaves its sums as one slab, plain stores; the slabs are summed
// in order by yp_sum_slabs (bit-reproducible).  Replaces the pixel-major copies of both tensors + a split-K convolution over them:
// 670 us -> ~70 us per training step of YOLOPoint-s at 16 x 640 x 640.
// ---------------------------------------------------------------------------------------------------------------------------
struct StemWgradArgs {
    const char* x;
    const char* dy;
    float* slabs;
    int H, W, Ho, Wo;                      // image / output map
    int dy_cs, dy_co, Cout_pad;
    int tiles_x, tiles_y, ntiles;
};

template <int DT, int NCB>
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const StemWgradArgs a) {
    constexpr int TH = 8, PW = 36, PROWS = 20, UNITS = PROWS * (PW / 2);       // image patch: 20 rows x 36 pixels = 360 16-byte units
    constexpr int XBYTES = 8192, NPIX = TH * 16, KK = NPIX / 32, DYBYTES = NCB * NPIX * 32, STAGE = XBYTES + DYBYTES, NDMA = 2 + NCB;
    extern __shared__ __attribute__((aligned(1024))) char wsm[];
    const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) char*)wsm);
    const int t = threadIdx.x, l = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int half = l & 1, prow = l >> 1;

    auto issue = [&](int tile, int stage) {
        const unsigned sb = lds0 + stage * STAGE;
        int r_ = tile;
        const int tx = r_ % a.tiles_x; r_ /= a.tiles_x;
        const int ty = r_ % a.tiles_y;
        const int b = r_ / a.tiles_y, y0 = ty * TH, x0 = tx * 16;
#pragma unroll
        for (int i = 0; i < 2; ++i) {                       // image patch: unit u = row * 18 + pixel pair, laid out densely (pitch 288 bytes)
            const int slab = i * 4 + wave;
            const int u = slab * 64 + l;
            const int row = u / (PW / 2), cu = u - row * (PW / 2);
            const int iy = 2 * y0 - 2 + row, ix = 2 * x0 - 2 + 2 * cu;
            const bool ok = u < UNITS && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            const char* src = a.x + (((long)b * a.H + iy) * a.W + ix) * 8;
            wg_glds16(ok ? (const void*)src : (const void*)wg_zero16, sb + slab * 1024);
        }
#pragma unroll
        for (int i = 0; i < NCB; ++i) {                     // dy: [16-channel block][128 pixels][32 bytes]
            const int q = wave + 4 * i;
            const int cb = q >> 2, rb = q & 3;
            const int row = rb * 32 + prow;
            const int ch = cb * 16 + half * 8;
            const int oy = y0 + (row >> 4), ox = x0 + (row & 15);
            const bool ok = ch < a.Cout_pad && oy < a.Ho && ox < a.Wo;
            const long pix = ((long)b * a.Ho + oy) * a.Wo + ox;
            const char* src = a.dy + (pix * a.dy_cs + a.dy_co + ch) * 2;
            wg_glds16(ok ? (const void*)src : (const void*)wg_zero16, sb + XBYTES + (cb * NPIX + rb * 32) * 32);
        }
    };

    // fragment read addressing as in wgrad_body: lane i of group g addresses row j = i/4 of its 4-row block, 8-byte quarter i%4
    const int li = l & 15, g = l >> 4, j = li >> 2, q4 = li & 3;
    int xoff[2], yoff[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int k32 = g * 8 + ((h ^ (g & 1)) * 4) + j;
        xoff[h] = ((2 * (k32 >> 4)) * PW + 2 * (k32 & 15)) * 8 + q4 * 8;
        yoff[h] = k32 * 32 + q4 * 8;
    }
    f32x4 acc[3][NCB];
#pragma unroll
    for (int fi = 0; fi < 3; ++fi)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) acc[fi][cb] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int first = blockIdx.x, step = gridDim.x;
    if (first < a.ntiles) issue(first, 0);
    int it = 0;
    for (int tile = first; tile < a.ntiles; tile += step, ++it) {
        const bool more = tile + step < a.ntiles;
        if (more) issue(tile + step, (it + 1) & 1);
        if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const char* xs = wsm + (it & 1) * STAGE;
        const char* ys = xs + XBYTES;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            s16x8 yf[NCB];
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                const char* base = ys + (cb * NPIX + kk * 32) * 32;
                yf[cb] = wg_tr8(base + yoff[0], base + yoff[1]);
            }
#pragma unroll
            for (int fi = 0; fi < 3; ++fi) {
                const int f = wave * 3 + fi, r = f >> 1, hh = f & 1;
                const char* base = xs + ((4 * kk + r) * PW + 4 * hh) * 8;
                const s16x8 xf = wg_tr8(base + xoff[0], base + xoff[1]);
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) acc[fi][cb] = wg_mma<DT>(xf, yf[cb], acc[fi][cb]);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    // accumulator lane (li, g): image-side row 4g + jj = (tap column 4hh + g, image channel jj), dy channel cb*16 + li.
    // slab [ci][6][6][Cout_pad]: every element has exactly one writer
    float* slab = a.slabs + (size_t)blockIdx.x * (4 * 36 * a.Cout_pad);
#pragma unroll
    for (int fi = 0; fi < 3; ++fi) {
        const int f = wave * 3 + fi, r = f >> 1, sc = 4 * (f & 1) + g;
        if (sc >= 6) continue;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            const int co = cb * 16 + li;
            if (co >= a.Cout_pad) continue;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) slab[((size_t)(jj * 6 + r) * 6 + sc) * a.Cout_pad + co] = acc[fi][cb][jj];
        }
    }
}

template <int DT, int NCB>
static hipError_t launch_stem_wgrad(const StemWgradArgs& a, int grid, hipStream_t st) {
    constexpr size_t lds = (size_t)2 * (8192 + NCB * 128 * 32);
    auto kern = stem_wgrad_kernel<DT, NCB>;
    static YpLdsAttr attr;        // per instantiation, per device
    if (hipError_t e = yp_set_max_lds(attr, (const void*)kern, (int)lds); e != hipSuccess) return e;
    kern<<<grid, 256, lds, st>>>(a);
    return hipGetLastError();
}

}  // namespace

static int stem_wgrad_grid(int B, int Ho, int Wo) {
    const int ntiles = B * yp_cdiv(Ho, 8) * yp_cdiv(Wo, 16);
    return ntiles < 512 ? ntiles : 512;
}

extern "C" int yp_stem_wgrad_slabs(int B, int H, int W) { return (B > 0 && H > 0 && W > 0) ? stem_wgrad_grid(B, H / 2, W / 2) : 0; }

extern "C" int yp_stem_wgrad(YpView x, YpView dy, int dtype, int B, float* slabs, float* dw, void* stream) {
    YP_REQUIRE(dtype == YP_F16 || dtype == YP_BF16, "yp_stem_wgrad: 16-bit element types only");
    YP_REQUIRE(x.ptr && dy.ptr && slabs && dw && B > 0 && ((uintptr_t)slabs & 15) == 0 && ((uintptr_t)dw & 15) == 0, "yp_stem_wgrad: null / unaligned buffer");
    YP_REQUIRE(x.C == 4 && x.cstride == 4 && x.coff == 0 && x.ups == 0 && x.H % 2 == 0 && x.W % 2 == 0, "yp_stem_wgrad: x must be the packed 4-channel image");
    YP_REQUIRE(dy.H == x.H / 2 && dy.W == x.W / 2 && dy.C % 8 == 0 && dy.C > 0 && dy.C <= 80 && dy.cstride % 8 == 0 && dy.coff % 8 == 0 && dy.ups == 0,
               "yp_stem_wgrad: dy %dx%dx%d does not match a 6x6 / stride 2 / pad 2 convolution of a %dx%d image", dy.H, dy.W, dy.C, x.H, x.W);
    StemWgradArgs a{};
    a.x = (const char*)x.ptr; a.dy = (const char*)dy.ptr; a.slabs = slabs;
    a.H = x.H; a.W = x.W; a.Ho = dy.H; a.Wo = dy.W;
    a.dy_cs = dy.cstride; a.dy_co = dy.coff; a.Cout_pad = dy.C;
    a.tiles_x = yp_cdiv(dy.W, 16); a.tiles_y = yp_cdiv(dy.H, 8); a.ntiles = B * a.tiles_x * a.tiles_y;
    const int grid = stem_wgrad_grid(B, dy.H, dy.W);
    const int ncb = yp_cdiv(dy.C, 16);
    hipStream_t st = (hipStream_t)stream;
    hipError_t e;
#define YP_S(DT) (ncb == 1 ? launch_stem_wgrad<DT, 1>(a, grid, st) : ncb == 2 ? launch_stem_wgrad<DT, 2>(a, grid, st) : ncb == 3 ? launch_stem_wgrad<DT, 3>(a, grid, st) \
                  : ncb == 4 ? launch_stem_wgrad<DT, 4>(a, grid, st) : launch_stem_wgrad<DT, 5>(a, grid, st))
    e = dtype == YP_F16 ? YP_S(YP_F16) : YP_S(YP_BF16);
#undef YP_S
    if (e != hipSuccess) { yp_set_error("yp_stem_wgrad: launch failed: %s", hipGetErrorString(e)); return YP_ERR_HIP; }
    // (512 slabs of a few thousand floats: a fixed two-level tree, 16 slabs per first-level group)
    if (grid % 16 == 0) return yp_sum_slabs_tree(slabs, dw, (size_t)4 * 36 * dy.C, grid, 16, stream);
    return yp_sum_slabs(slabs, dw, (size_t)4 * 36 * dy.C, grid, stream);
}

// argument set + launch geometry of one weight gradient (shared by the single and the grouped entry points)
// block: channels per side of a workgroup's dW block -- 64, 128 (1x1 filters), or 0 = 128 where it applies (1x1, >= 128 channels on both sides)
// (the larger block pays off when a workgroup still walks >= 12 pixel tiles after the pixel split -- with few tiles per workgroup its
// longer fill / drain and the single workgroup per CU cost more than the better MFMA : LDS ratio wins: YOLOPoint-s at 8 samples per GPU)
static bool wgrad_big_ok(YpView x, YpView dy, int B, int k) {
    if (k != 1 || x.C < 128 || dy.C < 128) return false;
    const long ntiles = ((long)B * dy.H * dy.W + 127) / 128;
    const int nblk = yp_cdiv(x.C, 128) * yp_cdiv(dy.C, 128);
    const int split = yp_cdiv(128, nblk);
    return ntiles >= 12l * split;
}
static int wgrad_make_args(YpView x, YpView dy, int dtype, int B, int k, int stride, float* dw, WgradArgs* out, int* nblk_out, int* split_out, int target = 256,
                           int block = 0) {
    YP_REQUIRE(block == 0 || block == 64 || (block == 128 && k == 1), "yp_conv_wgrad: block is 64, or 128 for 1x1 filters");
    const int blk = block == 0 ? (wgrad_big_ok(x, dy, B, k) ? 128 : 64) : block;
    const bool q8 = dtype == YP_FP8;        // x = e4m3 bytes, dy = e5m2 bytes (the twins of csrc/fp8.hip)
    YP_REQUIRE(dtype == YP_F16 || dtype == YP_BF16 || q8, "yp_conv_wgrad: 16-bit element types, or YP_FP8 (x e4m3, dy e5m2)");
    YP_REQUIRE(!q8 || (x.C % 16 == 0 && x.cstride % 16 == 0 && x.coff % 16 == 0 && dy.C % 16 == 0 && dy.cstride % 16 == 0 && dy.coff % 16 == 0),
               "yp_conv_wgrad: 8-bit views must be 16-channel aligned");
    YP_REQUIRE((k == 1 && stride == 1) || (k == 3 && (stride == 1 || stride == 2)), "yp_conv_wgrad: 1x1 (stride 1) or 3x3 (pad 1, stride 1 | 2) filters only");
    YP_REQUIRE(x.ptr && dy.ptr && dw && B > 0, "yp_conv_wgrad: null buffer");
    YP_REQUIRE(x.C > 0 && x.C % 8 == 0 && x.cstride % 8 == 0 && x.coff % 8 == 0 && dy.C > 0 && dy.C % 8 == 0 && dy.cstride % 8 == 0 && dy.coff % 8 == 0,
               "yp_conv_wgrad: views must be 8-channel aligned");
    const int Hi = x.H << x.ups, Wi = x.W << x.ups;
    YP_REQUIRE(x.ups >= 0 && x.ups <= 1 && dy.ups == 0 && (Hi + stride - 1) / stride == dy.H && (Wi + stride - 1) / stride == dy.W && Hi % stride == 0 && Wi % stride == 0,
               "yp_conv_wgrad: x %dx%d<<%d vs dy %dx%d at stride %d", x.H, x.W, x.ups, dy.H, dy.W, stride);
    const long M = (long)B * dy.H * dy.W;
    YP_REQUIRE(M < (1l << 30), "yp_conv_wgrad: too many pixels");
    WgradArgs a{};
    a.x = (const char*)x.ptr; a.dy = (const char*)dy.ptr; a.dw = dw;
    a.x_cs = x.cstride; a.x_co = x.coff; a.x_ups = x.ups; a.x_H = x.H; a.x_W = x.W;
    a.dy_cs = dy.cstride; a.dy_co = dy.coff;
    a.B = B; a.H = dy.H; a.W = dy.W; a.Cj = x.C; a.Cout_pad = dy.C; a.M = (int)M;
    a.blk = blk;
    a.n_co_blk = yp_cdiv(dy.C, blk);
    a.skip_store = false;
    const int nblk = yp_cdiv(x.C, blk) * a.n_co_blk;
    if (k == 3) { a.tiles_x = yp_cdiv(dy.W, 16); a.tiles_y = yp_cdiv(dy.H, stride == 2 ? 4 : 8); a.ntiles = B * a.tiles_x * a.tiles_y; }
    else a.ntiles = yp_cdiv((int)M, 128);
    // pixel split: one workgroup per CU.  Every workgroup ends in 64*64*taps fp32 atomics and the chip retires only ~250 G of them
    // per second (measured in round 3: skipping the flush halves the total), so more, smaller workgroups lose: 768 -> 256
    // workgroups took the 23 distinct YOLOPoint-s shapes from 863 to 612 us.  A single 64x64 block of a 3x3 filter (nblk = 1)
    // is best at ~96 when it has few tiles per workgroup anyway.
    // (grouped launches run dozens of entries side by side: 128 workgroups per entry keep the chip full with half the slab / flush traffic --
    // 13.7 -> 13.0 ms per training step)
    int split = yp_cdiv(target, nblk);
    int cap = (k == 3 && nblk == 1 && a.ntiles <= 800) ? 96 : 256;
    if (split > cap) split = cap;
    if (split > a.ntiles) split = a.ntiles;
    if (split < 1) split = 1;
    a.g_nblk = nblk; a.g_split = split;
    *out = a; *nblk_out = nblk; *split_out = split;
    return YP_OK;
}

extern "C" int yp_conv_wgrad(YpView x, YpView dy, int dtype, int B, int k, int stride, float* dw, void* stream) {
    WgradArgs a;
    int nblk, split;
    if (int rc = wgrad_make_args(x, dy, dtype, B, k, stride, dw, &a, &nblk, &split)) return rc;
    const dim3 grid(nblk, split);
    const hipError_t e = dtype == YP_F16 ? dispatch_wgrad<YP_F16>(k, stride, a, grid, (hipStream_t)stream) : dispatch_wgrad<YP_BF16>(k, stride, a, grid, (hipStream_t)stream);
    if (e != hipSuccess) { yp_set_error("yp_conv_wgrad: launch failed: %s", hipGetErrorString(e)); return YP_ERR_HIP; }
    return YP_OK;
}

// 8-bit operands: x = e4m3 twin, dy = e5m2 twin of the same logical views, sx / sdy their per-tensor scales (device scalars)
extern "C" int yp_conv_wgrad_q8(YpView x8, YpView dy8, const float* sx, const float* sdy, int B, int k, int stride, float* dw, void* stream) {
    YP_REQUIRE(sx != nullptr && sdy != nullptr, "yp_conv_wgrad_q8: null scale");
    WgradArgs a;
    int nblk, split;
    if (int rc = wgrad_make_args(x8, dy8, YP_FP8, B, k, stride, dw, &a, &nblk, &split)) return rc;
    a.sx = sx; a.sdy = sdy;
    const hipError_t e = dispatch_wgrad8(k, stride, a, dim3(nblk, split), (hipStream_t)stream);
    if (e != hipSuccess) { yp_set_error("yp_conv_wgrad_q8: launch failed: %s", hipGetErrorString(e)); return YP_ERR_HIP; }
    return YP_OK;
}

extern "C" size_t yp_wgrad_group_entry_bytes(void) { return sizeof(WgradArgs); }

extern "C" int yp_wgrad_group_pack(const YpView* xs, const YpView* dys, float* const* dws, int n, int dtype, int B, int k, int stride, int block, void* table_host,
                                   int* total_blocks) {
    return yp_wgrad_group_pack_det(xs, dys, dws, nullptr, n, dtype, B, k, stride, block, table_host, total_blocks, nullptr);
}

extern "C" int yp_wgrad_block(YpView x, YpView dy, int B, int k) { return wgrad_big_ok(x, dy, B, k) ? 128 : 64; }

extern "C" size_t yp_wgrad_partial_elems(YpView x, YpView dy, int dtype, int B, int k, int stride, int block) {
    WgradArgs a;
    int nblk, split;
    float dummy;
    if (wgrad_make_args(x, dy, dtype, B, k, stride, &dummy, &a, &nblk, &split, 128, block) != YP_OK) return 0;
    return (size_t)split * a.Cj * (k * k) * a.Cout_pad;
}

extern "C" int yp_wgrad_group_pack_det(const YpView* xs, const YpView* dys, float* const* dws, float* const* parts, int n, int dtype, int B, int k, int stride,
                                       int block, void* table_host, int* total_blocks, int* fold_chunks) {
    YP_REQUIRE(block == 64 || block == 128, "yp_wgrad_group_pack: a grouped launch runs ONE block size (64 | 128: yp_wgrad_block of its entries)");
    YP_REQUIRE(xs && dys && dws && table_host && total_blocks && n > 0 && (parts == nullptr || fold_chunks != nullptr), "yp_wgrad_group_pack: bad arguments");
    WgradArgs* t = (WgradArgs*)table_host;
    int blk0 = 0, chunk0 = 0;
    for (int i = 0; i < n; ++i) {
        int nblk, split;
        if (int rc = wgrad_make_args(xs[i], dys[i], dtype, B, k, stride, dws[i], &t[i], &nblk, &split, 128, block)) return rc;
        t[i].g_blk0 = blk0;
        blk0 += nblk * split;
        if (parts != nullptr) {
            YP_REQUIRE(parts[i] != nullptr && ((uintptr_t)parts[i] & 15) == 0 && ((uintptr_t)dws[i] & 15) == 0, "yp_wgrad_group_pack_det: slabs / dW must be 16-byte aligned");
            t[i].part = parts[i];
            t[i].f_chunk0 = chunk0;
            t[i].f_chunks = yp_cdiv(t[i].Cj * k * k * t[i].Cout_pad, 1024);
            chunk0 += t[i].f_chunks;
        }
    }
    *total_blocks = blk0;
    if (fold_chunks) *fold_chunks = chunk0;
    return YP_OK;
}

// The grouped table for 8-bit entries: yp_wgrad_group_pack_det with dtype = YP_FP8 plus one (sx, sdy) pair of device scalars per entry
extern "C" int yp_wgrad_group_pack_q8(const YpView* xs, const YpView* dys, float* const* dws, float* const* parts, const float* const* sx, const float* const* sdy,
                                      int n, int B, int k, int stride, int block, void* table_host, int* total_blocks, int* fold_chunks) {
    YP_REQUIRE(sx != nullptr && sdy != nullptr, "yp_wgrad_group_pack_q8: null scale tables");
    if (int rc = yp_wgrad_group_pack_det(xs, dys, dws, parts, n, YP_FP8, B, k, stride, block, table_host, total_blocks, fold_chunks)) return rc;
    WgradArgs* t = (WgradArgs*)table_host;
    for (int i = 0; i < n; ++i) {
        YP_REQUIRE(sx[i] != nullptr && sdy[i] != nullptr, "yp_wgrad_group_pack_q8: null scale of entry %d", i);
        t[i].sx = sx[i]; t[i].sdy = sdy[i];
    }
    return YP_OK;
}

template <int TAPS, int ST, int NB = 2>
static hipError_t launch_wgrad8_group(const WgradArgs* table, int n, int blocks, hipStream_t st) {
    constexpr size_t lds = wgrad8_lds<TAPS, ST, NB>();
    auto kern = wgrad8_group_kernel<TAPS, ST, NB>;
    static YpLdsAttr attr;        // per instantiation, per device
    if (hipError_t e = yp_set_max_lds(attr, (const void*)kern, (int)lds); e != hipSuccess) return e;
    kern<<<blocks, 256, lds, st>>>(table, n);
    return hipGetLastError();
}

template <int DT, int TAPS, int ST, int NB = 2>
static hipError_t launch_wgrad_group(const WgradArgs* table, int n, int blocks, hipStream_t st) {
    constexpr int TH = (TAPS == 9 && ST == 2) ? 4 : 8;
    constexpr int HP = 16 * ST + (ST == 1 ? 2 : 1);
    constexpr int HROWS = (TH * ST + (ST == 1 ? 2 : 1)) * HP;
    constexpr int XR = TAPS == 9 ? (HROWS + 31) / 32 * 32 : 128;
    constexpr size_t lds = (size_t)2 * (2 * NB * XR * 32 + 2 * NB * TH * 16 * 32);
    auto kern = wgrad_group_kernel<DT, TAPS, ST, NB>;
    static YpLdsAttr attr;        // per instantiation, per device
    if (hipError_t e = yp_set_max_lds(attr, (const void*)kern, (int)lds); e != hipSuccess) return e;
    kern<<<blocks, 256, lds, st>>>(table, n);
    return hipGetLastError();
}

extern "C" int yp_wgrad_group_run(const void* table_dev, int n, int total_blocks, int dtype, int k, int stride, int block, void* stream) {
    return yp_wgrad_group_run_det(table_dev, n, total_blocks, 0, dtype, k, stride, block, stream);
}

extern "C" int yp_wgrad_group_run_det(const void* table_dev, int n, int total_blocks, int fold_chunks, int dtype, int k, int stride, int block, void* stream) {
    YP_REQUIRE(table_dev && n > 0 && total_blocks > 0 && fold_chunks >= 0 && (dtype == YP_F16 || dtype == YP_BF16 || dtype == YP_FP8), "yp_wgrad_group_run: bad arguments");
    YP_REQUIRE(block == 64 || (block == 128 && k == 1), "yp_wgrad_group_run: block is 64, or 128 for 1x1 filters");
    YP_REQUIRE((k == 1 && stride == 1) || (k == 3 && (stride == 1 || stride == 2)), "yp_wgrad_group_run: 1x1 (stride 1) or 3x3 (stride 1 | 2)");
    const WgradArgs* t = (const WgradArgs*)table_dev;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e;
#define YP_G(DT) (k == 1 ? (block == 128 ? launch_wgrad_group<DT, 1, 1, 4>(t, n, total_blocks, st) : launch_wgrad_group<DT, 1, 1>(t, n, total_blocks, st)) : (stride == 2 ? launch_wgrad_group<DT, 9, 2>(t, n, total_blocks, st) : launch_wgrad_group<DT, 9, 1>(t, n, total_blocks, st)))
    if (dtype == YP_FP8)
        e = k == 1 ? (block == 128 ? launch_wgrad8_group<1, 1, 4>(t, n, total_blocks, st) : launch_wgrad8_group<1, 1>(t, n, total_blocks, st))
                   : (stride == 2 ? launch_wgrad8_group<9, 2>(t, n, total_blocks, st) : launch_wgrad8_group<9, 1>(t, n, total_blocks, st));
    else e = dtype == YP_F16 ? YP_G(YP_F16) : YP_G(YP_BF16);
#undef YP_G
    if (e != hipSuccess) { yp_set_error("yp_wgrad_group_run: launch failed: %s", hipGetErrorString(e)); return YP_ERR_HIP; }
    if (fold_chunks > 0) {           // deterministic mode: sum the slices' slabs in order
        wgrad_fold_kernel<<<fold_chunks, 256, 0, st>>>(t, n, k * k);
        YP_CHECK_HIP(hipGetLastError());
    }
    return YP_OK;
}
