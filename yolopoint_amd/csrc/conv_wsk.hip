// Wave-private split-K implicit-GEMM convolution for the SHORT-M layers (P4 / P5 of the pyramid at small batch): tile ids 71..73.
//
// Why another kernel (reference: the Conv / Bottleneck / C3 / SPPF layers of models/common.py:22-34,79-89,123-135,213-229 at 40x40 and
// 20x20).  At batch 8 these layers have M = B*Ho*Wo = 3 200 .. 12 800 output pixels: 100-400 tiles of
// 64 x 64 for 256 CUs, i.e. ONE four-wave workgroup per CU whose k loop is a serial chain per k tile -- barrier -> DMA issue -> LDS
// read -> 4 MFMAs per wave: ~600 clocks for 64 clocks of matrix work (tools/probe/wg_census.py).  A second wave per
// SIMD is what hides such a chain, and the only place left to find independent work is K:
//
//   * every WAVE of the workgroup takes the k tiles t = w, w + NW, w + 2 NW, ... and multiplies them against the WHOLE BM x 64 output
//     tile (16 MFMAs per 8 fragment reads for 64 x 64, instead of 4 per 4);
//   * each wave stages its own operands in its own two-slot LDS ring (LDS-DMA, global_load_lds_dwordx4): nothing in the main loop is
//     shared between waves, so there is NO barrier in it -- a wave waits only for its own DMA (counted vmcnt) and the 8 (4) waves of the
//     workgroup drift apart and fill each other's stalls;
//   * the NW partial tiles are summed through the (now idle) ring memory in a fixed order (reduce-scatter: every wave adds up and
//     finishes 1/NW of the tile: bias -> SiLU -> (+ residual) -> 16-byte NHWC stores), so the result is bit-reproducible; the fp32
//     summation order differs from the sequential kernels (NW interleaved partial sums).
//
// Bound: the per-CU L2 -> LDS path (~64 B/clk): a 64 x 64 tile moves (64 + 64) x 64 B = 8 KB per 32-deep k tile = 128 clocks against
// 68 clocks of MFMA work per SIMD-quad -- the kernel is built to sit on that bound instead of on the latency chain.
// Addressing is the FAST path of conv_igemm.hip (16-bit types, every source a multiple of 32 channels, zero tails behind the buffers):
// two sources, 2x nearest upsampled views, any stride / filter size, split destinations, residual.
#include "conv_common.h"

#ifdef YP_PROBE_WSK
// probe build: shader-clock (s_memtime) and 100 MHz wall-clock stamps of wave `yp_wsk_tl_wave` of workgroup 0 at the phase boundaries
__device__ unsigned long long yp_wsk_tl[32];
__device__ int yp_wsk_tl_wave = 0;
extern "C" int yp_debug_wsk_timeline(unsigned long long* out_host) { return (int)hipMemcpyFromSymbol(out_host, HIP_SYMBOL(yp_wsk_tl), sizeof(unsigned long long) * 32); }
extern "C" int yp_debug_wsk_timeline_wave(int wv) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(yp_wsk_tl_wave), &wv, sizeof(int)); }
#define YPW_TL(i) do { if (tl_hit) { yp_wsk_tl[i] = __builtin_readcyclecounter(); } } while (0)
#else
#define YPW_TL(i) do {} while (0)
#endif

namespace {

// exact m / d for m * d < 2^32 with mg = ceil(2^32 / d), d >= 2 (the launcher refuses other shapes)
__device__ __forceinline__ int yp_div_magic(int m, unsigned mg) { return (int)__umulhi((unsigned)m, mg); }

}  // namespace

template <int DT, bool OUT_F32, int BM, int NW, bool CONTIG>
__global__ __launch_bounds__(64 * NW) void conv_wsk_kernel(const ConvKArgs a) {
    using E = Elem<DT>;
    using frag_t = typename E::frag;
    static_assert(DT == YP_F16 || DT == YP_BF16, "16-bit element types");
    constexpr int EB = 2, BK = 32, ROWB = 64, BN = 64;
    constexpr int SA = BM / 16, SB = BN / 16;            // 1-KiB DMA slots (16 rows x 64 B) of one k tile: pixels, filter rows
    constexpr int SUB = (SA + SB) * 1024;                // LDS image of one k tile
    constexpr int NS = 2;                                // slots of a wave's ring
    constexpr int FMT = BM / 16, FNT = BN / 16;
    constexpr int UNITS = 2 * FMT, UPW = UNITS / NW;     // output units of 16 pixels x 32 channels (8 consecutive channels per lane)
    static_assert(UNITS % NW == 0 && UPW >= 1 && (UPW == 1 || UPW % 2 == 0), "every wave finishes the same number of output units");
    static_assert(NW * FNT * FMT * 1024 <= 160 * 1024 && NW * NS * SUB <= 160 * 1024, "LDS budget");

    extern __shared__ __attribute__((aligned(1024))) char smem[];

    const int nblk = gridDim.x;
    const int logical = yp_xcd_remap(blockIdx.x, nblk);
    const int tile_n = logical % a.tiles_n, tile_m = logical / a.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
#ifdef YP_PROBE_WSK
    const bool tl_hit = blockIdx.x == 0 && lane == 0 && w == __builtin_amdgcn_readfirstlane(yp_wsk_tl_wave);
#endif
    YPW_TL(0);
    const int lrow = lane >> 2;
    const int jl = (lane & 3) ^ ((0x3300 >> ((lane >> 4) * 4)) & 3);      // logical 16-byte chunk this lane fetches (source-side swizzle)

    // ---- per-slot gather state of the pixel rows (fixed for the whole k range)
    int hi0[SA], wi0[SA], bb[SA];
#pragma unroll
    for (int i = 0; i < SA; ++i) {
        const int m = m0 + i * 16 + lrow;
        if (m < a.M) {
            const int b = yp_div_magic(m, a.mg_howo);
            const int rem = m - b * a.HoWo;
            const int ho = yp_div_magic(rem, a.mg_wo);
            const int wo = rem - ho * a.Wo;
            hi0[i] = ho * a.sh - a.ph;
            wi0[i] = wo * a.sw - a.pw;
            bb[i] = b;
        } else {
            hi0[i] = -(1 << 28); wi0[i] = 0; bb[i] = 0;
        }
    }
    // filter rows: LDS row f*16 + g*4 + r holds channel n0 + g*16 + f*4 + r, so that a lane's four fragments interleave to 16
    // consecutive channels of its pixel
    unsigned b_off[SB];
#pragma unroll
    for (int i = 0; i < SB; ++i) {
        const int g_ = lrow >> 2, r_ = lrow & 3;
        const int n = n0 + g_ * 16 + i * 4 + r_;
        b_off[i] = ((n < a.Npad) ? (unsigned)n * (unsigned)a.Kpad * EB : a.wgt_zrow) + (unsigned)jl * 16u;
    }

    YP_PIN2(const char*, in0); YP_PIN2(const char*, in1); YP_PIN2(const char*, wgt);
    YP_PIN2(int, in0_cs); YP_PIN2(int, in1_cs); YP_PIN2(int, in0_co); YP_PIN2(int, in1_co); YP_PIN2(int, in0_C);
    YP_PIN2(int, in0_ups); YP_PIN2(int, in1_ups); YP_PIN2(int, in0_H); YP_PIN2(int, in1_H); YP_PIN2(int, in0_W); YP_PIN2(int, in1_W);
    YP_PIN2(int, Hi); YP_PIN2(int, Wi); YP_PIN2(int, Cin); YP_PIN2(int, S); YP_PIN2(int, invS);
    YP_PIN2(unsigned, in0_zoff); YP_PIN2(unsigned, in1_zoff);

    const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) char*)smem);
    const unsigned ring = lds0 + (unsigned)w * (NS * SUB);

    // ---- this wave's k tiles.  CONTIG: the contiguous range [kt0, kt0 + nkw) (a wave stays inside one filter tap / source as long as
    // possible: the per-lane offsets are recomputed once per run); else w, w + NW, ... (neighbouring waves fetch neighbouring 64-byte
    // halves of the same cache lines at about the same time).  (s_tap, s_c0) = filter tap and channel position of its NEXT tile
    const int nk_all = a.Kreal / BK;
    int kt0, nkw, kstep;
    if constexpr (CONTIG) {
        kt0 = (int)(((long)nk_all * w) / NW);
        nkw = (int)(((long)nk_all * (w + 1)) / NW) - kt0;
        kstep = 1;
    } else {
        kt0 = w;
        nkw = w < nk_all ? (nk_all - w + NW - 1) / NW : 0;
        kstep = NW;
    }
    int s_tap = 0, s_c0 = kt0 * BK;
    while (s_c0 >= Cin) { s_c0 -= Cin; ++s_tap; }
    int cur_key = -1;
    unsigned seg_voff[SA];
#pragma unroll
    for (int i = 0; i < SA; ++i) seg_voff[i] = 0;

    // prep(kt): scalar / per-lane addressing of this wave's next k tile (a new (filter tap, source tensor) run recomputes the per-lane pixel
    // offsets, otherwise only the scalar base moves); fire(j, dst): DMA instruction j of that tile (pixel slots first, then filter slots)
    const char* cur_sb = in0;
    const char* cur_wk = wgt;
    auto prep = [&](int kt) {
        const bool s0 = s_c0 < in0_C;
        const int key = s_tap * 2 + (s0 ? 0 : 1);
        if (key != cur_key) {
            cur_key = key;
            const unsigned zoff = s0 ? in0_zoff : in1_zoff;
            const int kr = (s_tap * invS) >> 16;
            const int ks = s_tap - kr * S;
            const int cs = s0 ? in0_cs : in1_cs;
            const int ups = s0 ? in0_ups : in1_ups;
            const int Hp = s0 ? in0_H : in1_H;
            const int Wp = s0 ? in0_W : in1_W;
            const bool zs = s0 && a.in0_zs;
            const int csb = cs * EB;
            const unsigned lanec = (unsigned)jl * 16u;
#pragma unroll
            for (int i = 0; i < SA; ++i) {
                const int hi = hi0[i] + kr, wi = wi0[i] + ks;
                const bool ok = (unsigned)hi < (unsigned)Hi && (unsigned)wi < (unsigned)Wi && !(zs && ((hi | wi) & 1));
                const int pix = (bb[i] * Hp + (hi >> ups)) * Wp + (wi >> ups);
                // (an out-of-image lane reads base + zoff: the zero tail behind the buffer, one pixel + 64 elements long)
                seg_voff[i] = ok ? (unsigned)(pix * csb) + lanec : zoff;
            }
        }
        const int c_in_src = s0 ? s_c0 : s_c0 - in0_C;
        cur_sb = (s0 ? in0 : in1) + (size_t)((s0 ? in0_co : in1_co) + c_in_src) * EB;
        cur_wk = wgt + (size_t)kt * (BK * EB);
        s_c0 += kstep * BK;
        while (s_c0 >= Cin) { s_c0 -= Cin; ++s_tap; }
    };
#ifdef YP_PROBE_WSK
    const int probe = a.probe;      // 1: no MFMA, 2: no steady-state DMA, 4: no reduction / epilogue, 8: no pixel DMA, 16: no filter DMA
#else
    constexpr int probe = 0;
#endif
    auto fire = [&](int j, unsigned dst) {
        if ((probe & 8) && j < SA) return;
        if ((probe & 16) && j >= SA) return;
        if (j < SA) yp_glds16_s(cur_sb, seg_voff[j < SA ? j : 0], dst + j * 1024);
        else yp_glds16_s(cur_wk, b_off[j >= SA ? j - SA : 0], dst + j * 1024);
    };

    const int p = lane & 15, g = lane >> 4;
    const int swr = (0x3300 >> ((p >> 2) * 4)) & 3;
    const int rd = p * ROWB + ((g ^ swr) << 4);

    f32x4 acc[FNT][FMT];
#pragma unroll
    for (int f = 0; f < FNT; ++f)
#pragma unroll
        for (int fm = 0; fm < FMT; ++fm) acc[f][fm] = f32x4{0.f, 0.f, 0.f, 0.f};

    // bias of the channels this lane finishes (older than every DMA: the counted waits below stay valid).  One unit per wave: its 8
    // channels (half h = w & 1); several: all 16 of the lane group (the halves alternate with the unit index)
    constexpr int NBIAS = UPW == 1 ? 8 : 16;
    float bias[NBIAS];
    {
        const int nb0 = n0 + g * 16 + (UPW == 1 ? (w & 1) * 8 : 0);
#pragma unroll
        for (int q = 0; q < NBIAS / 4; ++q) {
            const int nb = nb0 + 4 * q;
            f32x4 b4 = f32x4{0.f, 0.f, 0.f, 0.f};
            if (a.bias != nullptr && nb < a.Cout) b4 = *reinterpret_cast<const f32x4*>(a.bias + nb);
            bias[4 * q] = b4[0]; bias[4 * q + 1] = b4[1]; bias[4 * q + 2] = b4[2]; bias[4 * q + 3] = b4[3];
        }
    }

    // Pipeline: tiles it+1 and it+2 are in flight while tile `it` is multiplied.  The fragments of tile `it` are in registers before its
    // MFMAs issue, so its slot is refilled (tile it+2) BETWEEN those MFMAs: the DMA issue slots hide in the matrix pipe's shadow.
    constexpr int NG = SA + SB, MF = FNT * FMT;
    if (nkw > 0) {
        prep(kt0);
#pragma unroll
        for (int j = 0; j < NG; ++j) fire(j, ring);
    }
    if (nkw > 1) {
        prep(kt0 + kstep);
#pragma unroll
        for (int j = 0; j < NG; ++j) fire(j, ring + SUB);
    }
    auto load_frags = [&](int it, frag_t (&wf)[FNT], frag_t (&xf)[FMT]) {
        const char* s = smem + w * (NS * SUB) + (it & 1) * SUB + rd;
#pragma unroll
        for (int f = 0; f < FNT; ++f) wf[f] = *reinterpret_cast<const frag_t*>(s + (SA + f) * 1024);
#pragma unroll
        for (int fm = 0; fm < FMT; ++fm) xf[fm] = *reinterpret_cast<const frag_t*>(s + fm * 1024);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // every fragment is in registers: the slot may be overwritten
    };
    YPW_TL(1);
    int it = 0;
    for (; it + 2 < nkw; ++it) {           // steady state: tile it+2 is issued between the MFMAs of tile `it`
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NG) : "memory");          // tile `it` has landed, tile it+1 stays in flight
        if (it < 8) YPW_TL(2 + it);
        frag_t wf[FNT], xf[FMT];
        load_frags(it, wf, xf);
        if (it == 2) YPW_TL(20);
        const unsigned dst = ring + (unsigned)(it & 1) * SUB;
        prep(kt0 + (it + 2) * kstep);
        if (it == 2) YPW_TL(21);
#pragma unroll
        for (int q = 0; q < MF; ++q) {
            if (q % 2 == 0 && q / 2 < NG) { if (!(probe & 2)) fire(q / 2, dst); __builtin_amdgcn_sched_barrier(0); }
            if (probe & 1) { asm volatile("" ::"v"(wf[q / FMT]), "v"(xf[q % FMT])); continue; }
            acc[q / FMT][q % FMT] = E::mma(wf[q / FMT], xf[q % FMT], acc[q / FMT][q % FMT]);
            if (q % 2 == 1 && q / 2 < NG) __builtin_amdgcn_sched_barrier(0);
        }
        if (it == 2) YPW_TL(22);
    }
    for (; it < nkw; ++it) {               // the last two tiles: nothing left to issue
        if (it + 1 < nkw) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NG) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        frag_t wf[FNT], xf[FMT];
        load_frags(it, wf, xf);
#pragma unroll
        for (int q = 0; q < MF; ++q) acc[q / FMT][q % FMT] = E::mma(wf[q / FMT], xf[q % FMT], acc[q / FMT][q % FMT]);
    }

    YPW_TL(12);
    // ---- reduce-scatter of the NW partial tiles through the ring memory (every DMA of every wave has landed and been read)
    if (probe & 4) { if (acc[0][0][0] == 123.456f) a.out[0] = 1; return; }
    __syncthreads();
    YPW_TL(13);
    f32x4* red = reinterpret_cast<f32x4*>(smem);
#pragma unroll
    for (int f = 0; f < FNT; ++f)
#pragma unroll
        for (int fm = 0; fm < FMT; ++fm) red[((w * FNT + f) * FMT + fm) * 64 + lane] = acc[f][fm];
    __syncthreads();
    YPW_TL(14);
#pragma unroll
    for (int u = 0; u < UPW; ++u) {
        const int unit = w * UPW + u;
        const int fm = unit >> 1;
        const int h = UPW == 1 ? (w & 1) : (u & 1);        // (UPW is even when > 1)
        float v[8];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int f = 2 * h + q;
            f32x4 sum = red[((0 * FNT + f) * FMT + fm) * 64 + lane];
#pragma unroll
            for (int ws = 1; ws < NW; ++ws) {
                const f32x4 x = red[((ws * FNT + f) * FMT + fm) * 64 + lane];
                sum[0] += x[0]; sum[1] += x[1]; sum[2] += x[2]; sum[3] += x[3];
            }
            v[4 * q] = sum[0]; v[4 * q + 1] = sum[1]; v[4 * q + 2] = sum[2]; v[4 * q + 3] = sum[3];
        }
        const int m = m0 + fm * 16 + p;
        const int nc = n0 + g * 16 + h * 8;
        if (m >= a.M || nc >= a.Cout) continue;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float x = v[j] + bias[(UPW == 1 ? 0 : (u & 1) * 8) + j];
            if (a.act == YP_ACT_SILU) x = yp_silu(x);
            v[j] = x;
        }
        yp_store_chunk<DT, OUT_F32, 8>(a, m, nc, v);
    }
    YPW_TL(15);
}

namespace {

template <int DT, bool OUT_F32, int BM, int NW, bool CONTIG>
hipError_t launch_wsk(const ConvKArgs& a, int nblk, hipStream_t st) {
    constexpr size_t ring = (size_t)NW * 2 * (BM / 16 + 4) * 1024, red = (size_t)NW * 4 * (BM / 16) * 1024;
    constexpr size_t lds = ring > red ? ring : red;
    auto kern = conv_wsk_kernel<DT, OUT_F32, BM, NW, CONTIG>;
    static YpLdsAttr attr;        // per instantiation, per device
    if (hipError_t e = yp_set_max_lds(attr, (const void*)kern, (int)lds); e != hipSuccess) return e;
    kern<<<nblk, 64 * NW, lds, st>>>(a);
    return hipGetLastError();
}

template <int DT, bool OUT_F32>
hipError_t dispatch_wsk(int tile, const ConvKArgs& a, int nblk, hipStream_t st) {
    switch (tile) {
        case 71: return launch_wsk<DT, OUT_F32, 64, 8, false>(a, nblk, st);
        case 72: return launch_wsk<DT, OUT_F32, 64, 4, false>(a, nblk, st);
        case 73: return launch_wsk<DT, OUT_F32, 128, 4, false>(a, nblk, st);
        case 74: return launch_wsk<DT, OUT_F32, 64, 8, true>(a, nblk, st);
        case 75: return launch_wsk<DT, OUT_F32, 64, 4, true>(a, nblk, st);
        case 76: return launch_wsk<DT, OUT_F32, 128, 4, true>(a, nblk, st);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace

bool yp_wsk_tile_dims(int tile, int* bm, int* bn) {
    int m = 0;
    switch (tile) {
        case 71: case 72: case 74: case 75: m = 64; break;
        case 73: case 76: m = 128; break;
        default: return false;
    }
    if (bm) *bm = m;
    if (bn) *bn = 64;
    return true;
}

hipError_t yp_wsk_launch(int tile, int dtype, bool out_f32, const ConvKArgs& a0, int nblk, hipStream_t st) {
    ConvKArgs a = a0;
#ifdef YP_PROBE_WSK
    { const char* e = getenv("YP_WSK_PROBE"); a.probe = e ? atoi(e) : 0; }
#endif
    if (dtype == YP_F16) return out_f32 ? dispatch_wsk<YP_F16, true>(tile, a, nblk, st) : dispatch_wsk<YP_F16, false>(tile, a, nblk, st);
    if (dtype == YP_BF16) return out_f32 ? dispatch_wsk<YP_BF16, true>(tile, a, nblk, st) : dispatch_wsk<YP_BF16, false>(tile, a, nblk, st);
    return hipErrorInvalidValue;
}
