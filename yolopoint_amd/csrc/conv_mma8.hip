// Third-generation implicit-GEMM convolution for gfx950 (MI355X): 8 wavefronts per workgroup, 32x32x16 MFMA, 128-byte k rows,
// two wave groups in ping-pong.  Serves the compute-bound layers (channels % 64 == 0, M = B*Ho*Wo in the tens of thousands):
// reference models/common.py:22-34 (Conv), :79-89 (Bottleneck.cv2), autograd's dgrad of both (train.py:245).
//
// Why a second kernel.  The 4-wave kernel of conv_igemm.hip moves (BM + BN) * 64 bytes through the L1 -> LDS path per 32-deep k tile;
// that path delivers ~64 B/clk per CU, the four SIMDs retire 2 * BM * BN * 32 FLOP in BM * BN * 64 / 4069 clocks, so a 128 x 128 tile
// needs 64 B/clk at the full MFMA rate -- it cannot pass ~50 %, and measured 15-30 %.  Here:
//   * tile 256 pixels x 256 channels (32 B/clk at the full MFMA rate), k tile = 64 elements = one 128-byte line per row and DMA lane group;
//   * per wave 128 x 64 outputs = 4 x 2 accumulator tiles of 32 x 32 (128 registers): 6 fragment reads per 8 MFMAs of 32 clocks each;
//   * the waves of a workgroup form two groups (waves 0-3 / 4-7: one wave of each group per SIMD).  A k tile is four PHASES (one 16-deep
//     MFMA step each); a phase is a LOAD segment (6 ds_read_b128, 2-3 LDS-DMA instructions of the next k tile, lgkmcnt(0)) and an MFMA segment
//     (8 MFMAs under s_setprio 1), separated by raw s_barriers; group 1 runs one segment behind group 0, so on every SIMD one wave
//     multiplies while the other one loads -- the DMA issue slots (60-100 clocks each) and the LDS latency never sit in front of an MFMA;
//   * per-tap address arithmetic (one segment = the k tiles of one filter tap and source tensor) is emitted inside an MFMA segment.
// LDS image of a stage: (BP + BC) rows of 128 bytes (pixels, then filter rows).  One DMA instruction brings 8 rows; bank conflicts of the
// 16-byte fragment reads (lanes 0-31: 32 rows, lanes 32-63: the next 8 k elements) are removed by chunk ^= (row >> 1) & 7, applied
// on the DMA source side and by the readers.  Filter rows are permuted when fetched so that a lane's accumulators are 16 * CT consecutive
// channels of one pixel (MFMA row 8j + 4h + i of channel tile ct = channel h*16*CT + ct*16 + 4j + i): 16-byte NHWC stores.
#include "conv_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int DT> struct Mma32;
template <> struct Mma32<YP_F16> {
    using frag = f16x8;
    static __device__ __forceinline__ f32x16 mma(frag a, frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};
template <> struct Mma32<YP_BF16> {
    using frag = bf16x8;
    static __device__ __forceinline__ f32x16 mma(frag a, frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};

template <int DT, bool OUT_F32, int BP, int BC, int WP, int WC, int NS, bool STATS>
__global__ __launch_bounds__(512) void conv_mma8_kernel(const ConvKArgs a) {
    using MM = Mma32<DT>;
    using frag_t = typename MM::frag;
    constexpr int ROWB = 128, BK = 64, EB = 2;
    constexpr int TP = BP / (2 * WP), TC = BC / WC, PT = TP / 32, CT = TC / 32;
    constexpr int NLP = BP / 64, NLW = BC / 64, NL = NLP + NLW;           // DMA instructions per wave per k tile
    constexpr int STAGE = (BP + BC) * ROWB;
    static_assert(WP * WC == 4 && PT >= 1 && CT >= 1 && BP % 64 == 0 && BC % 64 == 0 && NS >= 2 && NS <= 4, "unsupported tile");
    static_assert((NS - 1) * NL <= 60, "vmcnt immediate range");

    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) char*)smem);

    const int logical = yp_xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = logical % a.tiles_n, tile_m = logical / a.tiles_n;
    const int m0 = tile_m * BP, n0 = tile_n * BC;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int g = wave >> 2, wi = wave & 3, wp = wi % WP, wc = wi / WP;

    // ---- DMA lane constants: an instruction covers rows 8j .. 8j+7 (j = wave + 8i: its parity is the wave's), lane -> row lane/8, chunk lane%8
    const int lrow = lane >> 3;
    const unsigned lanec = (unsigned)(((lane & 7) ^ (4 * (wave & 1) + (lane >> 4))) << 4);

    YP_PIN2(const char*, in0); YP_PIN2(const char*, in1); YP_PIN2(const char*, wgt);
    YP_PIN2(int, in0_cs); YP_PIN2(int, in1_cs); YP_PIN2(int, in0_co); YP_PIN2(int, in1_co); YP_PIN2(int, in0_C);
    YP_PIN2(int, in0_ups); YP_PIN2(int, in1_ups); YP_PIN2(int, in0_H); YP_PIN2(int, in1_H); YP_PIN2(int, in0_W); YP_PIN2(int, in1_W);
    YP_PIN2(int, Hi); YP_PIN2(int, Wi); YP_PIN2(int, Cin); YP_PIN2(int, S); YP_PIN2(int, invS);
    YP_PIN2(unsigned, in0_zoff); YP_PIN2(unsigned, in1_zoff); YP_PIN2(int, in0_zs);

    int hi0[NLP], wi0[NLP], bb[NLP];
#pragma unroll
    for (int i = 0; i < NLP; ++i) {
        const int m = m0 + 8 * (wave + 8 * i) + lrow;
        if (m < a.M) {
            const int b = m / a.HoWo;
            const int rem = m - b * a.HoWo;
            const int ho = rem / a.Wo;
            const int wo = rem - ho * a.Wo;
            hi0[i] = ho * a.sh - a.ph;
            wi0[i] = wo * a.sw - a.pw;
            bb[i] = b;
        } else {
            hi0[i] = -(1 << 28);
            wi0[i] = 0;
            bb[i] = 0;
        }
    }
    unsigned w_off[NLW];
#pragma unroll
    for (int i = 0; i < NLW; ++i) {
        const int rl = 8 * (wave + 8 * i) + lrow;                // LDS filter row -> output channel (see the header comment)
        const int wcx = rl / TC, q = rl % TC;
        const int ct = q >> 5, rho = q & 31;
        const int n = n0 + wcx * TC + ((rho >> 2) & 1) * (16 * CT) + ct * 16 + (rho >> 3) * 4 + (rho & 3);
        w_off[i] = ((n < a.Npad) ? (unsigned)n * (unsigned)a.Kpad * EB : a.wgt_zrow) + lanec;
    }

    // ---- k-tile issue state.  A SEGMENT is the run of k tiles inside one (filter tap, source tensor): the per-lane pixel offsets are
    // constant there and the channel position rides in the scalar base pointer.
    int s_tap = 0, s_c0 = 0, seg_left = 0, kt_prep = 0;
    const char* seg_base = in0;
    unsigned seg_voff[NLP];
#pragma unroll
    for (int i = 0; i < NLP; ++i) seg_voff[i] = 0;
    const char* cur_p = in0;
    const char* cur_w = wgt;
    auto prepare = [&]() {                  // bases / offsets of k tile kt_prep -> cur_p, cur_w, seg_voff
        if (seg_left == 0) {
            const int kr = (s_tap * invS) >> 16;
            const int ks = s_tap - kr * S;
            const bool s0 = s_c0 < in0_C;
            const char* base = s0 ? in0 : in1;
            const int cs = s0 ? in0_cs : in1_cs;
            const int ups = s0 ? in0_ups : in1_ups;
            const int Hp = s0 ? in0_H : in1_H;
            const int Wp = s0 ? in0_W : in1_W;
            const bool zs = s0 && in0_zs;
            const int c_in_src = s0 ? s_c0 : s_c0 - in0_C;
            const unsigned zoff = s0 ? in0_zoff : in1_zoff;
            seg_left = ((s0 ? in0_C : Cin - in0_C) - c_in_src) / BK;
            seg_base = base + (size_t)((s0 ? in0_co : in1_co) + c_in_src) * EB;
            const int csb = cs * EB;
#pragma unroll
            for (int i = 0; i < NLP; ++i) {
                const int hi = hi0[i] + kr, wi_ = wi0[i] + ks;
                const bool ok = (unsigned)hi < (unsigned)Hi && (unsigned)wi_ < (unsigned)Wi && !(zs && ((hi | wi_) & 1));
                const int pix = (bb[i] * Hp + (hi >> ups)) * Wp + (wi_ >> ups);
                seg_voff[i] = ok ? (unsigned)(pix * csb) + lanec : zoff;     // (out-of-image rows read the zero tail behind the buffer)
            }
        }
        cur_p = seg_base;
        cur_w = wgt + (size_t)kt_prep * (BK * EB);
        seg_base += BK * EB;
        --seg_left;
        s_c0 += BK;
        if (s_c0 >= Cin) { s_c0 -= Cin; ++s_tap; }
        ++kt_prep;
    };
    // DMA instructions [lo, hi) of the prepared k tile -> ring stage `stage` (pixel rows first, then filter rows)
    auto issue_range = [&](int stage, auto lo_c, auto hi_c) {
        constexpr int lo = decltype(lo_c)::value, hi = decltype(hi_c)::value;
        const unsigned sbase = lds0 + (unsigned)stage * STAGE;
#pragma unroll
        for (int idx = lo; idx < hi; ++idx) {
            if (idx < NLP) yp_glds16_s(cur_p, seg_voff[idx < NLP ? idx : 0], sbase + (unsigned)(wave + 8 * idx) * 1024u);
            else yp_glds16_s(cur_w, w_off[idx >= NLP ? idx - NLP : 0], sbase + (unsigned)(BP * ROWB) + (unsigned)(wave + 8 * (idx - NLP)) * 1024u);
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, (NL + 2) / 3>;
    using I2 = std::integral_constant<int, (2 * NL + 2) / 3>;
    using I3 = std::integral_constant<int, NL>;

    // ---- fragment read offsets: lane -> row lane%32, logical chunk 2p + lane/32 of phase p; physical chunk = logical ^ ((row >> 1) & 7)
    const int lr = lane & 31, hh = lane >> 5;
    const int rd_lane = lr * ROWB + ((hh ^ ((lr >> 1) & 7)) << 4);
    const char* const p_rd = smem + (g * (BP / 2) + wp * TP) * ROWB;
    const char* const w_rd = smem + (BP + wc * TC) * ROWB;

    f32x16 acc[CT][PT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int pt = 0; pt < PT; ++pt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ct][pt][r] = 0.f;

    const int nk = a.Kreal / BK;

    // ---- prologue: tiles 0 .. NS-2 in flight, tile NS-1 prepared
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) {
        if (s < nk) { prepare(); issue_range(s, I0{}, I3{}); }
    }
    if (NS - 1 < nk) prepare();
    // tile 0 has landed (up to NS-2 younger tiles stay in flight)
    if (nk >= NS - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * NL) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (g == 1) __builtin_amdgcn_s_barrier();                     // group 1 runs one segment behind group 0

    int stage = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const char* const ps = p_rd + stage * STAGE;
        const char* const ws = w_rd + stage * STAGE;
        int nstage = stage + NS - 1;
        if (nstage >= NS) nstage -= NS;
        const bool more = kt + NS - 1 < nk;                       // tile kt+NS-1 exists: its DMA is issued during this tile's load segments
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            // ---- load segment
            frag_t wf[CT], pf[PT];
            const int off = rd_lane ^ (p << 5);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) wf[ct] = *reinterpret_cast<const frag_t*>(ws + ct * 32 * ROWB + off);
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) pf[pt] = *reinterpret_cast<const frag_t*>(ps + pt * 32 * ROWB + off);
            if (more) {
                if (p == 0) issue_range(nstage, I0{}, I1{});
                if (p == 1) issue_range(nstage, I1{}, I2{});
                if (p == 2) issue_range(nstage, I2{}, I3{});
            }
            if (p == 3) {
                // tile kt+1 must have landed before its first read (next load segment, after the barrier below publishes it)
                int younger = nk - kt - 2;
                if (younger > NS - 2) younger = NS - 2;
                if (younger <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NS > 2 ? NL : 0) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NS > 3 ? 2 * NL : 0) : "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // ---- MFMA segment
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) acc[ct][pt] = MM::mma(wf[ct], pf[pt], acc[ct][pt]);
            __builtin_amdgcn_s_setprio(0);
            if (p == 3 && kt + NS < nk) prepare();                 // (address arithmetic of the next tile to issue: VALU beside the MFMAs)
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        stage = stage + 1 == NS ? 0 : stage + 1;
    }
    if (g == 0) __builtin_amdgcn_s_barrier();                     // (group 0 waits out group 1's last segment: equal barrier counts)

    // ---- epilogue: lane = pixel lr of each pixel tile, 16 * CT consecutive channels starting at nb
    const int nb = n0 + wc * TC + hh * (16 * CT);
    if constexpr (STATS) {
        // BatchNorm statistics of the raw output from the accumulators (training forward): per lane the sums over its PT pixels, then a
        // recursive-halving reduction over the 32 pixel lanes (fixed order: bit-reproducible); one partial row per wave-row of
        // 32 * PT pixels: stats[(rb*2 + {0,1})*Cout + c], rb = (tile_m * 2 + g) * WP + wp.  Rows behind M were multiplied from zero pages.
        constexpr int NV = 16 * CT;
        float sv[NV], sq[NV];
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) { const float v = acc[j >> 4][pt][j & 15]; s += v; q += v * v; }
            sv[j] = s; sq[j] = q;
        }
        int mych = 0;
#pragma unroll
        for (int st = 0; st < 5; ++st) {
            const int o = 1 << st;
            const int n = (NV >> st) > 1 ? (NV >> st) : 1;          // values still held per lane before this step (compile-time after unrolling)
            const bool up = (lr & o) != 0;
            if (n > 1) {
                const int hn = n / 2;
#pragma unroll
                for (int j = 0; j < NV / 2; ++j) {
                    if (j < hn) {
                        const float keep_s = up ? sv[hn + j] : sv[j], give_s = up ? sv[j] : sv[hn + j];
                        const float keep_q = up ? sq[hn + j] : sq[j], give_q = up ? sq[j] : sq[hn + j];
                        sv[j] = keep_s + __shfl_xor(give_s, o, 64);
                        sq[j] = keep_q + __shfl_xor(give_q, o, 64);
                    }
                }
                mych += up ? hn : 0;
            } else {
                sv[0] += __shfl_xor(sv[0], o, 64);
                sq[0] += __shfl_xor(sq[0], o, 64);
            }
        }
        const bool owner = NV >= 32 || (lr & 16) == 0;            // (NV = 16: lanes lr and lr ^ 16 hold the same total)
        const int c = nb + mych;
        const int rb = (tile_m * 2 + g) * WP + wp;
        if (owner && c < a.Cout && (size_t)rb * (32 * PT) < (size_t)a.M) {
            a.stats[((size_t)rb * 2 + 0) * a.Cout + c] = sv[0];
            a.stats[((size_t)rb * 2 + 1) * a.Cout + c] = sq[0];
        }
    }
    float bias[16 * CT];
#pragma unroll
    for (int q = 0; q < 4 * CT; ++q) {
        f32x4 b4 = f32x4{0.f, 0.f, 0.f, 0.f};
        if (a.bias != nullptr && nb + 4 * q < a.Cout) b4 = *reinterpret_cast<const f32x4*>(a.bias + nb + 4 * q);
        bias[4 * q] = b4[0]; bias[4 * q + 1] = b4[1]; bias[4 * q + 2] = b4[2]; bias[4 * q + 3] = b4[3];
    }
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        const int m = m0 + g * (BP / 2) + wp * TP + pt * 32 + lr;
        if (m >= a.M) continue;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
            for (int h8 = 0; h8 < 2; ++h8) {
                const int nc = nb + ct * 16 + h8 * 8;
                if (nc >= a.Cout) continue;
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float x = acc[ct][pt][h8 * 8 + j] + bias[ct * 16 + h8 * 8 + j];
                    if (a.act == YP_ACT_SILU) x = yp_silu(x);
                    v[j] = x;
                }
                yp_store_chunk<DT, OUT_F32, 8>(a, m, nc, v);
            }
        }
    }
}

namespace {

template <int DT, bool OUT_F32, int BP, int BC, int WP, int WC, int NS, bool STATS>
hipError_t launch_mma8(const ConvKArgs& a, int nblk, hipStream_t st) {
    constexpr size_t lds = (size_t)NS * (BP + BC) * 128;
    static_assert(lds <= 160 * 1024, "LDS budget");
    auto kern = conv_mma8_kernel<DT, OUT_F32, BP, BC, WP, WC, NS, STATS>;
    static bool attr_set = false;        // per instantiation
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    kern<<<nblk, 512, lds, st>>>(a);
    return hipGetLastError();
}

template <int DT, bool OUT_F32, bool STATS>
hipError_t dispatch_mma8(int tile, const ConvKArgs& a, int nblk, hipStream_t st) {
    switch (tile) {
        case 41: return launch_mma8<DT, OUT_F32, 256, 256, 1, 4, 2, STATS>(a, nblk, st);
        case 42: return launch_mma8<DT, OUT_F32, 256, 128, 2, 2, 3, STATS>(a, nblk, st);
        case 43: return launch_mma8<DT, OUT_F32, 128, 256, 1, 4, 3, STATS>(a, nblk, st);
        case 44: return launch_mma8<DT, OUT_F32, 128, 128, 1, 4, 2, STATS>(a, nblk, st);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace

bool yp_mma8_tile_dims(int tile, int* bp, int* bc, int* stat_rows_px) {
    int p = 0, c = 0, r = 0;
    switch (tile) {
        case 41: p = 256; c = 256; r = 128; break;
        case 42: p = 256; c = 128; r = 64; break;
        case 43: p = 128; c = 256; r = 64; break;
        case 44: p = 128; c = 128; r = 64; break;
        default: return false;
    }
    if (bp) *bp = p;
    if (bc) *bc = c;
    if (stat_rows_px) *stat_rows_px = r;
    return true;
}

hipError_t yp_mma8_launch(int tile, int dtype, bool out_f32, bool stats, const ConvKArgs& a, int nblk, hipStream_t st) {
    if (dtype == YP_F16) {
        if (stats) return dispatch_mma8<YP_F16, false, true>(tile, a, nblk, st);
        return out_f32 ? dispatch_mma8<YP_F16, true, false>(tile, a, nblk, st) : dispatch_mma8<YP_F16, false, false>(tile, a, nblk, st);
    }
    if (dtype == YP_BF16) {
        if (stats) return dispatch_mma8<YP_BF16, false, true>(tile, a, nblk, st);
        return out_f32 ? dispatch_mma8<YP_BF16, true, false>(tile, a, nblk, st) : dispatch_mma8<YP_BF16, false, false>(tile, a, nblk, st);
    }
    return hipErrorInvalidValue;
}
