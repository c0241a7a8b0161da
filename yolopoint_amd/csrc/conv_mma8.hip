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
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifdef YP_PROBE8
// probe build: waves 0 and 4 of workgroup 0 record the shader clock at the segment boundaries of k tile 6 (YP8_TS(i))
__device__ unsigned long long yp8_timeline[8][32];
extern "C" int yp_debug_mma8_timeline(unsigned long long* out_host) { return (int)hipMemcpyFromSymbol(out_host, HIP_SYMBOL(yp8_timeline), sizeof(unsigned long long) * 256); }
#define YP8_TS_DECL unsigned long long yp8_x[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long yp8_ts[24]; _Pragma("unroll") for (int i_ = 0; i_ < 24; ++i_) yp8_ts[i_] = 0; const bool yp8_rec = blockIdx.x == 0 && wave0_ >= 0
#define YP8_TS(i) do { if (yp8_rec && kt == 6) { __builtin_amdgcn_sched_barrier(0); yp8_ts[i] = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#define YP8_TSX(i) do { if (blockIdx.x == 0) { __builtin_amdgcn_sched_barrier(0); yp8_x[i] = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#define YP8_TS_FLUSH2() do { if (blockIdx.x == 0 && lane == 0) { _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) yp8_timeline[wave][24 + i_] = yp8_x[i_]; } } while (0)
#define YP8_TS_FLUSH() do { if (yp8_rec && lane == 0) { _Pragma("unroll") for (int i_ = 0; i_ < 24; ++i_) yp8_timeline[wave][i_] = yp8_ts[i_]; } } while (0)
#else
#define YP8_TS_DECL do {} while (0)
#define YP8_TS(i) do {} while (0)
#define YP8_TS_FLUSH() do {} while (0)
#define YP8_TSX(i) do {} while (0)
#define YP8_TS_FLUSH2() do {} while (0)
#endif

template <int DT> struct Mma32;
template <> struct Mma32<YP_F16> {
    using frag = f16x8;
    static __device__ __forceinline__ f32x16 mma(frag a, frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};
template <> struct Mma32<YP_BF16> {
    using frag = bf16x8;
    static __device__ __forceinline__ f32x16 mma(frag a, frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
// 8-bit inputs (OCP fp8) at the fp8 rate: the block-scaled K = 64 form v_mfma_scale_f32_32x32x64_f8f6f4 (the only fp8 MFMA that issues at
// twice the 16-bit rate; the non-scaled 32x32x16 / 16x16x32 fp8 forms run at the bf16 rate) with UNIT block scales (E8M0 127 = 2^0): the
// per-tensor dequantisation scales multiply the accumulators in the epilogue, exactly as in the 4-wave kernel.  A fragment is 32 bytes per
// lane (two 16-byte LDS reads); filter and activation fragments are cut the same way, so the k order inside a step cancels.
//   YP_FP8: filter e4m3 x activation e4m3 (forward)      YP_FP8_BF8: filter e4m3 x activation e5m2 (dgrad: the activation is dy)
typedef int i32x8 __attribute__((ext_vector_type(8)));
template <> struct Mma32<YP_FP8> {
    using frag = i32x8;
    static __device__ __forceinline__ f32x16 mma(frag a, frag b, f32x16 c) {
        return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    }
};
template <> struct Mma32<YP_FP8_BF8> {
    using frag = i32x8;
    static __device__ __forceinline__ f32x16 mma(frag a, frag b, f32x16 c) {
        return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 1, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    }
};

// SCHED: the main-loop schedule.
//   0: ping-pong, phases of one 16-deep step (8 MFMAs for the 256 x 256 tile), the DMA of the next k tile in the load segments of phases 0..2
//   7: ping-pong, TWO phases per k tile (16-bit: two steps each; 8-bit: one K = 64 step each); every wave fetches pixel rows of its OWN group
//      only: the filter-row instructions go out in the load segment of phase 0, the pixel-row instructions in the load segment of phase 1
//      (awaited behind that phase's MFMAs: only the issuing group reads them, one segment later than the filter rows are needed)
//   8: free-running, ONE barrier per k tile, fragments double-buffered in registers (see the loop)
// Measured and dropped (same layer, same session; all within +-5 % of schedule 0): DMA inside the MFMA segments,
// phases of two steps with the DMA split between a load and an MFMA segment, 64-byte rows with a 4-stage ring, one barrier per k tile
// without the early publication of schedule 8.  tools/probe/mfma_lds_probe.hip reproduces the plateau outside the kernel.
//
// NW = 4 (round 5): ONE wavefront per SIMD, a wave = BP pixels x 64 channels (PT = BP / 32 accumulator tiles x 2: up to 256 accumulator
// registers of the 512 a lone wave owns), free-running schedule only.  No second wave competes for the matrix pipe, so the waves of a
// workgroup reach the k-tile barrier together (the ~350-clock arrival skew of the 8-wave form is what its 0.73 loop efficiency loses),
// and BP is any multiple of 32: the host picks the row count that fills the 256 CUs (224 rows: 229 workgroups for M = 51 200 instead of 200).
template <int DT, bool OUT_F32, int BP, int BC, int WP, int WC, int NS, bool STATS, int SCHED = 0, int NW = 8>
__global__ __launch_bounds__(NW * 64) void conv_mma8_kernel(const ConvKArgs a) {
    using MM = Mma32<DT>;
    using frag_t = typename MM::frag;
    constexpr int EB = Elem<DT>::BYTES;                                   // 2, or 1 (OCP fp8: a 128-byte row holds 128 k elements = two K = 64 MFMA steps)
    constexpr bool Q8 = EB == 1;
    constexpr int ROWB = 128, BK = ROWB / EB;
    constexpr int RPI = 1024 / ROWB, CPR = ROWB / 16, KS = ROWB / (Q8 ? 64 : 32);      // rows per DMA instruction, 16-byte chunks per row, MFMA steps per k tile
    static_assert(!Q8 || SCHED == 7, "8-bit inputs: the two-phase schedule");
    constexpr int NG = NW / 4;                                            // wave groups (one wave of each group per SIMD)
    static_assert(NW == 8 || (NW == 4 && SCHED == 8), "four waves: the free-running schedule");
    constexpr int TP = BP / (NG * WP), TC = BC / WC, PT = TP / 32, CT = TC / 32;
    constexpr int NLP = BP / (NW * RPI), NLW = BC / (NW * RPI), NL = NLP + NLW;         // DMA instructions per wave per k tile
    constexpr int STAGE = (BP + BC) * ROWB;
    constexpr int PH = (SCHED == 7) ? KS / 2 : 1, NPH = KS / PH;          // MFMA steps per phase, phases per k tile
    constexpr bool OWNP = SCHED == 7;                                     // pixel-row DMA of a wave covers its own group's rows only                  // 16-deep MFMA steps per phase, phases per k tile
    static_assert(WP * WC == 4 && PT >= 1 && CT >= 1 && TP % 32 == 0 && BP % (NW * RPI) == 0 && BC % (NW * RPI) == 0 && NS >= 2 && NS <= 4, "unsupported tile");
    static_assert((NS - 1) * NL <= 60, "vmcnt immediate range");

    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) char*)smem);

    const int wave0_ = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    YP8_TS_DECL;
    YP8_TSX(0);
    const int logical = yp_xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = logical % a.tiles_n, tile_m = logical / a.tiles_n;
    const int m0 = tile_m * BP, n0 = tile_n * BC;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int g = wave >> 2, wi = wave & 3, wp = wi % WP, wc = wi / WP;

#ifdef YP_PROBE8
    const int probe = __builtin_amdgcn_readfirstlane(a.probe);
#else
    constexpr int probe = 0;
#endif
    // ---- DMA lane constants: an instruction covers rows 8j .. 8j+7 (j = wave + 8i: its parity is the wave's), lane -> row lane/8, chunk lane%8
    const int lrow = lane / CPR;
    const unsigned lanec = (unsigned)(((lane & 7) ^ (4 * (wave & 1) + (lane >> 4))) << 4);

    YP_PIN2(const char*, in0); YP_PIN2(const char*, in1); YP_PIN2(const char*, wgt);
    YP_PIN2(int, in0_cs); YP_PIN2(int, in1_cs); YP_PIN2(int, in0_co); YP_PIN2(int, in1_co); YP_PIN2(int, in0_C);
    YP_PIN2(int, in0_ups); YP_PIN2(int, in1_ups); YP_PIN2(int, in0_H); YP_PIN2(int, in1_H); YP_PIN2(int, in0_W); YP_PIN2(int, in1_W);
    YP_PIN2(int, Hi); YP_PIN2(int, Wi); YP_PIN2(int, Cin); YP_PIN2(int, S); YP_PIN2(int, invS);
    YP_PIN2(unsigned, in0_zoff); YP_PIN2(unsigned, in1_zoff); YP_PIN2(int, in0_zs);

    int hi0[NLP], wi0[NLP], bb[NLP];
#pragma unroll
    for (int i = 0; i < NLP; ++i) {
        const int m = m0 + RPI * (OWNP ? g * (BP / (2 * RPI)) + wi + 4 * i : wave + NW * i) + lrow;
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
    // Pixel byte offsets of filter tap (0, 0) per source (+ the lane's chunk): for sources read at their own resolution a tap only adds the
    // wave-uniform (kr * W + ks) * pixel pitch, so a segment change costs two adds, two compares and a select per row -- the full
    // (b, y, x) -> offset arithmetic (integer multiplies; ~850 clocks per wave and filter tap when it sat in the k loop) runs once here.
    // Upsampled / zero-stuffed sources keep the general formula (`slow`).
    const bool slow = (in0_ups | in1_ups) != 0;
    unsigned pb0[NLP], pb1[NLP];
#pragma unroll
    for (int i = 0; i < NLP; ++i) {
        pb0[i] = (unsigned)((bb[i] * in0_H + hi0[i]) * in0_W + wi0[i]) * (unsigned)(in0_cs * EB) + lanec;
        pb1[i] = (unsigned)((bb[i] * in1_H + hi0[i]) * in1_W + wi0[i]) * (unsigned)(in1_cs * EB) + lanec;
    }
    // Tap validity per pixel row as a bit mask (bit kr * S + ks: the tap reads inside the image): a segment change is then four independent VALU
    // instructions per row.  (The compare / select form with its scalar hops measured ~1 500 clocks per change on a wave alone on its SIMD --
    // every fourth k tile of a 256-channel 3x3 layer, 7 % of the kernel.)
    unsigned okm[NLP];
#pragma unroll
    for (int i = 0; i < NLP; ++i) okm[i] = 0;
    if (!slow) {
        // rows kr in [max(0, -hi0), min(R, Hi - hi0)) and columns ks in [max(0, -wi0), min(S, Wi - wi0)) are inside: two bit ranges per pixel row in
        // closed form, the tap mask is their outer product (R * S <= 32, host-checked)
        const int R_ = a.R;
        auto range = [](int lo, int hi) -> unsigned { return hi > lo ? (((1u << hi) - 1u) & ~((1u << lo) - 1u)) : 0u; };
        unsigned rowm[NLP], colm[NLP];
#pragma unroll
        for (int i = 0; i < NLP; ++i) {
            const int rlo = min(max(0, -hi0[i]), R_), rhi = max(min(R_, Hi - hi0[i]), 0);
            const int clo = min(max(0, -wi0[i]), S), chi = max(min(S, Wi - wi0[i]), 0);
            rowm[i] = range(rlo, rhi);
            colm[i] = range(clo, chi);
        }
#pragma unroll 1
        for (int kr = 0; kr < R_; ++kr) {
#pragma unroll
            for (int i = 0; i < NLP; ++i) okm[i] |= (0u - ((rowm[i] >> kr) & 1u)) & (colm[i] << (kr * S));
        }
    }
    unsigned w_off[NLW];
#pragma unroll
    for (int i = 0; i < NLW; ++i) {
        const int rl = RPI * (wave + NW * i) + lrow;             // LDS filter row -> output channel (see the header comment)
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
            if (!slow) {
                // offset = valid ? pixel base + tap offset : the zero tail behind the buffer; branch-free: zoff + ((pb + d - zoff) & -valid)
                const unsigned dz = (unsigned)((kr * Wp + ks) * csb) - zoff;
                if (s0) {
#pragma unroll
                    for (int i = 0; i < NLP; ++i) seg_voff[i] = zoff + ((pb0[i] + dz) & (unsigned)__builtin_amdgcn_sbfe((int)okm[i], s_tap, 1));
                } else {
#pragma unroll
                    for (int i = 0; i < NLP; ++i) seg_voff[i] = zoff + ((pb1[i] + dz) & (unsigned)__builtin_amdgcn_sbfe((int)okm[i], s_tap, 1));
                }
            } else {
#pragma unroll
                for (int i = 0; i < NLP; ++i) {
                    const int hi = hi0[i] + kr, wi_ = wi0[i] + ks;
                    const bool ok = (unsigned)hi < (unsigned)Hi && (unsigned)wi_ < (unsigned)Wi && !(zs && ((hi | wi_) & 1));
                    const int pix = (bb[i] * Hp + (hi >> ups)) * Wp + (wi_ >> ups);
                    seg_voff[i] = ok ? (unsigned)(pix * csb) + lanec : zoff;
                }
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
    // DMA instruction idx of the prepared k tile -> ring stage `stage` (pixel rows first, then filter rows)
    auto issue_one = [&](int stage, int idx) {
        const unsigned sbase = lds0 + (unsigned)stage * STAGE;
        if (idx < NLP) yp_glds16_s(cur_p, seg_voff[idx < NLP ? idx : 0], sbase + (unsigned)(OWNP ? g * (BP / (2 * RPI)) + wi + 4 * idx : wave + NW * idx) * 1024u);
        else yp_glds16_s(cur_w, w_off[idx >= NLP ? idx - NLP : 0], sbase + (unsigned)(BP * ROWB) + (unsigned)(wave + NW * (idx - NLP)) * 1024u);
    };
    // schedule 0: the instructions [dma_lo(p), dma_lo(p + 1)) go out in the load segment of phase p (none in the last phase)
    auto dma_lo = [](int slot) -> int { return slot >= 3 ? NL : (slot * NL + 2) / 3; };

    // ---- fragment read offsets: lane -> row lane%32, logical chunk 2p + lane/32 of phase p; physical chunk = logical ^ ((row >> 1) & 7)
    const int lr = lane & 31, hh = lane >> 5;
    // (8-bit: step ks reads chunks 4ks + 2hh + {0, 1}; 16-bit: step p reads chunk 2p + hh)
    const int rd_lane = lr * ROWB + (((Q8 ? 2 * hh : hh) ^ ((lr >> 1) & 7)) << 4);
    auto ld_frag = [&](const char* base, int step) -> frag_t {
        if constexpr (Q8) {
            const int off = rd_lane ^ (step << 6);
            frag_t f;
            reinterpret_cast<u32x4*>(&f)[0] = *reinterpret_cast<const u32x4*>(base + off);
            reinterpret_cast<u32x4*>(&f)[1] = *reinterpret_cast<const u32x4*>(base + (off ^ 16));
            return f;
        } else {
            return *reinterpret_cast<const frag_t*>(base + (rd_lane ^ (step << 5)));
        }
    };
    const char* const p_rd = smem + (g * (BP / NG) + wp * TP) * ROWB;
    const char* const w_rd = smem + (BP + wc * TC) * ROWB;

    f32x16 acc[CT][PT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int pt = 0; pt < PT; ++pt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ct][pt][r] = 0.f;

    const int nk = a.Kreal / BK;

    YP8_TSX(1);
    // ---- prologue: tiles 0 .. NS-2 in flight, tile NS-1 prepared
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) {
        if (s < nk) {
            prepare();
#pragma unroll
            for (int idx = 0; idx < NL; ++idx) issue_one(s, idx);
        }
    }
    if (NS - 1 < nk) prepare();
    // tile 0 has landed (up to NS-2 younger tiles stay in flight)
    if (nk >= NS - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * NL) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    YP8_TSX(2);
    if constexpr (SCHED == 8) {
        // ---- free-running schedule: ONE barrier per k tile, no wave groups.  Measured (tools/probe/lds_read_probe.hip): a wave alone on its
        // SIMD gets one ds_read_b128 per ~29 clocks (6 reads: 176 clocks), two waves reading together get the full 256 B/clk -- and in the
        // ping-pong schedules every wave's own instruction stream (24 reads ~700 clocks + 8 DMA issues ~560 + 32 MFMAs 1024 + eight barrier
        // round trips) is the period of the k tile, ~3000 clocks however the segments are arranged.  Here both waves of a SIMD run the same
        // code and interleave freely on the matrix pipe (2 x 32 MFMAs = 2048 clocks per tile); what a wave does besides its MFMAs (reads for
        // the next step into the second fragment set, DMA issues between MFMAs) hides under the partner's MFMAs.  The barrier sits between
        // steps 2 and 3: tile t+1 (DMA parts issued in step 3 of tile t-1 and steps 0, 1 of tile t: at least one step of slack) is awaited and
        // published there, every wave's last reads of tile t (the fragments of step 3) have retired there (WAR for the DMA that refills the
        // stage from step 3 on), and step 3 loads the first fragments of tile t+1 beside its MFMAs -- no wave ever waits for LDS or DMA with
        // nothing to multiply except in the barrier itself.
        static_assert(NS == 2 && KS == 4 && NL >= 3, "SCHED 8: two stages, four 16-deep steps per k tile");
        constexpr int NM = CT * PT;
        constexpr int DA = NL / 4, DB = DA + (NL - DA + 1) / 2;               // DMA parts of a tile: [0, DA) | [DA, DB) | [DB, NL)
        frag_t wf[2][CT], pf[2][PT];
        auto load = [&](int stage_, int step, frag_t (&w_)[CT], frag_t (&p_)[PT]) {
            const char* const ps = p_rd + stage_ * STAGE;
            const char* const ws = w_rd + stage_ * STAGE;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) w_[ct] = ld_frag(ws + ct * 32 * ROWB, step);
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) p_[pt] = ld_frag(ps + pt * 32 * ROWB, step);
        };
        // (prologue above: tile 0 issued and awaited, tile 1 prepared)  part A of tile 1 now, then the first fragments
        if (1 < nk) {
#pragma unroll
            for (int idx = 0; idx < DA; ++idx) issue_one(1, idx);
        }
        load(0, 0, wf[0], pf[0]);
        int stage = 0;
        for (int kt = 0; kt < nk; ++kt) {
            const bool more1 = kt + 1 < nk, more2 = kt + 2 < nk;
            YP8_TS(0);
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int lo = p == 0 ? DA : (p == 1 ? DB : 0), hi = p == 0 ? DB : (p == 1 ? NL : (p == 3 ? DA : 0));
                const int nd = hi - lo;
                if constexpr (NW == 4) {
                    // One wave per SIMD: nothing else fills the matrix pipe while this wave issues LDS reads (one ds_read_b128 per ~29 clocks
                    // from a lone wave) or DMA instructions (60-80 clocks each), so they go out ONE PER MFMA, under the 32 clocks the matrix
                    // pipe spends on it: the step's CT + PT fragment reads for the NEXT step first (w0, p0 .. p(PT-1), w1 ..: the order in which
                    // the next step's MFMAs consume them), then this step's DMA part.
                    constexpr int NR = CT + PT;
                    const int nops = NR + nd;
                    const char* const rs_p = p_rd + (p < 3 ? stage : stage ^ 1) * STAGE;
                    const char* const rs_w = w_rd + (p < 3 ? stage : stage ^ 1) * STAGE;
                    const bool rd_on = p < 3 || more1;
                    const bool dma_on = (p == 3 ? more2 : more1) && !(probe & 2);
                    YP8_TS(1 + 4 * p);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < NM; ++i) {
                        const int ct = i / PT, pt = i % PT;
                        if (!(probe & 1)) acc[ct][pt] = MM::mma(wf[p & 1][ct], pf[p & 1][pt], acc[ct][pt]);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int j = 0; j < nops; ++j) {
                            if (j * NM / nops != i) continue;
                            if (j < NR) {
                                if (rd_on) {
                                    const int step = (p + 1) & 3;
                                    if (j == 0) wf[(p + 1) & 1][0] = ld_frag(rs_w, step);
                                    else if (j <= PT) pf[(p + 1) & 1][j - 1] = ld_frag(rs_p + (j - 1) * 32 * ROWB, step);
                                    else wf[(p + 1) & 1][j - PT] = ld_frag(rs_w + (j - PT) * 32 * ROWB, step);
                                }
                            } else if (dma_on) {
                                issue_one(p == 3 ? stage : stage ^ 1, lo + j - NR);
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else {
                if (p < 3) load(stage, p + 1, wf[(p + 1) & 1], pf[(p + 1) & 1]);
                else if (more1) load(stage ^ 1, 0, wf[0], pf[0]);
                YP8_TS(1 + 4 * p);
                __builtin_amdgcn_sched_barrier(0);
                const int every = nd > 0 ? (NM / nd > 0 ? NM / nd : 1) : NM + 1;
                __builtin_amdgcn_s_setprio(1);                     // (tools/probe/mfma_lds_probe.hip mode 7 vs 5: -8 % with the MFMA cluster prioritised)
#pragma unroll
                for (int i = 0; i < NM; ++i) {
                    const int ct = i / PT, pt = i % PT;
                    if (!(probe & 1)) acc[ct][pt] = MM::mma(wf[p & 1][ct], pf[p & 1][pt], acc[ct][pt]);
                    if ((i + 1) % every == 0 && (i + 1) / every <= nd) {
                        __builtin_amdgcn_sched_barrier(0);
                        if ((p == 3 ? more2 : more1) && !(probe & 2)) issue_one(p == 3 ? stage : stage ^ 1, lo + (i + 1) / every - 1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(0);
                }
                YP8_TS(2 + 4 * p);
                if (p == 2) {
                    if (more2 && !(probe & 8)) prepare();          // (tile kt+2: its part A goes out in step 3, behind the barrier)
                    YP8_TS(18);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    YP8_TS(11);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    YP8_TS(19);
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                    YP8_TS(12);
                }
            }
            YP8_TS(17);
            stage ^= 1;
        }
    } else {
    if (g == 1) __builtin_amdgcn_s_barrier();                     // group 1 runs one segment behind group 0

    int stage = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const char* const ps = p_rd + stage * STAGE;
        const char* const ws = w_rd + stage * STAGE;
        int nstage = stage + NS - 1;
        if (nstage >= NS) nstage -= NS;
        const bool more = kt + NS - 1 < nk;                       // tile kt+NS-1 exists: its DMA is issued during this tile
#pragma unroll
        for (int p = 0; p < NPH; ++p) {
            // ---- load segment
            YP8_TS(5 * p);
            frag_t wf[PH][CT], pf[PH][PT];
#pragma unroll
            for (int kk = 0; kk < PH; ++kk) {
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) wf[kk][ct] = ld_frag(ws + ct * 32 * ROWB, p * PH + kk);
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) pf[kk][pt] = ld_frag(ps + pt * 32 * ROWB, p * PH + kk);
            }
            if (SCHED == 7 && more && !(probe & 2)) {
#pragma unroll
                for (int idx = 0; idx < NL; ++idx)
                    if (p == 0 ? idx >= NLP : idx < NLP) issue_one(nstage, idx);
            }
            if (SCHED == 0 && more && !(probe & 2)) {
#pragma unroll
                for (int idx = 0; idx < NL; ++idx)
                    if (idx >= dma_lo(p) && idx < dma_lo(p + 1)) issue_one(nstage, idx);
            }
            YP8_TS(5 * p + 1);
            if (SCHED == 7 && p == NPH - 1) {
                if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLP) : "memory");       // the filter rows of tile kt+1 (older than this segment's pixel rows)
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            if (SCHED != 7 && p == NPH - 1) {
                // tile kt+1 must have landed before its first read (next load segment, after the barrier below publishes it)
                int younger = nk - kt - 2;
                if (younger > NS - 2) younger = NS - 2;
                if (younger <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NS > 2 ? NL : 0) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NS > 3 ? 2 * NL : 0) : "memory");
            }
            // (the reads of a tile's LAST phase retire before the barrier: behind it the other group's DMA may refill this stage;
            // in the other phases the LDS latency overlaps the barrier wait)
            if (p == NPH - 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            YP8_TS(5 * p + 2);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            YP8_TS(5 * p + 3);
            // ---- MFMA segment
            if (!(probe & 16)) __builtin_amdgcn_s_setprio(1);
            constexpr int NM = PH * CT * PT;
            // (MFMAs have no side effects: nothing but data dependences keeps them between the two barriers.  The fragments pass through an
            // empty asm behind barrier a and the accumulators through one in front of barrier b -- without them the 8-bit instantiation
            // had BOTH phases' MFMAs sunk behind the second phase's loads: all fragments live at once, spilled)
#pragma unroll
            for (int kk = 0; kk < PH; ++kk) {
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) asm volatile("" : "+v"(wf[kk][ct]));
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) asm volatile("" : "+v"(pf[kk][pt]));
            }
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                const int kk = i / (CT * PT), ct = (i / PT) % CT, pt = i % PT;
                if (!(probe & 1)) acc[ct][pt] = MM::mma(wf[kk][ct], pf[kk][pt], acc[ct][pt]);
            }
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) asm volatile("" : "+v"(acc[ct][pt]));
            __builtin_amdgcn_s_setprio(0);
            YP8_TS(5 * p + 4);
            if (SCHED == 7 && p == NPH - 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // own group's pixel rows of tile kt+1
            if (p == NPH - 1 && kt + NS < nk && !(probe & 8)) prepare();           // (address arithmetic of the next tile to issue: VALU beside the MFMAs)
            if (p == NPH - 1) YP8_TS(20);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        stage = stage + 1 == NS ? 0 : stage + 1;
    }
    if (g == 0) __builtin_amdgcn_s_barrier();                     // (group 0 waits out group 1's last segment: equal barrier counts)
    }

    YP8_TS_FLUSH();
    if constexpr (Q8) {                       // back to real units: the dequantisation scales of the activation and of the filter
        const float osc = a.scale_in[0] * a.scale_w[0];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) acc[ct][pt] *= osc;
    }
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
        auto xorl = [](float x, int st) {                           // (st is a constant after unrolling: the DPP forms of conv_common.h)
            return st == 0 ? yp_xor_lane<1>(x) : st == 1 ? yp_xor_lane<2>(x) : st == 2 ? yp_xor_lane<4>(x) : st == 3 ? yp_xor_lane<8>(x) : yp_xor_lane<16>(x);
        };
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
                        sv[j] = keep_s + xorl(give_s, st);
                        sq[j] = keep_q + xorl(give_q, st);
                    }
                }
                mych += up ? hn : 0;
            } else {
                sv[0] += xorl(sv[0], st);
                sq[0] += xorl(sq[0], st);
            }
        }
        const bool owner = NV >= 32 || (lr & 16) == 0;            // (NV = 16: lanes lr and lr ^ 16 hold the same total)
        const int c = nb + mych;
        const int rb = (tile_m * NG + g) * WP + wp;
        if (owner && c < a.Cout && (size_t)rb * (32 * PT) < (size_t)a.M) {
#ifndef YP_PROBE_NOSTATSTORE
            a.stats[((size_t)0 * a.Cout + c) * a.stats_rows + rb] = sv[0];
            a.stats[((size_t)1 * a.Cout + c) * a.stats_rows + rb] = sq[0];
#else
            if (sv[0] == 1.2345e-30f) { a.stats[0] = sv[0]; a.stats[1] = sq[0]; }
#endif
        }
    }
    float bias[16 * CT];
#pragma unroll
    for (int q = 0; q < 4 * CT; ++q) {
        f32x4 b4 = f32x4{0.f, 0.f, 0.f, 0.f};
        if (a.bias != nullptr && nb + 4 * q < a.Cout) b4 = *reinterpret_cast<const f32x4*>(a.bias + nb + 4 * q);
        bias[4 * q] = b4[0]; bias[4 * q + 1] = b4[1]; bias[4 * q + 2] = b4[2]; bias[4 * q + 3] = b4[3];
    }
    // (the bias has ARRIVED on every path into the store loop: without this common use the compiler re-waits -- vmcnt(0), behind the stores
    // issued meanwhile -- wherever a lane-masked block `m >= M` could have skipped the first use)
    yp_pin_arrived(bias);
    YP8_TSX(4);
    YpResRaw<DT, 8> res_next[CT][2];                // (residual launches: the next pixel's chunks, in flight across this pixel's stores)
    auto epilogue = [&](auto res_c) {
    constexpr int RES = decltype(res_c)::value;
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        const int m = m0 + g * (BP / NG) + wp * TP + pt * 32 + lr;
        if constexpr (RES == 1) {
            // Residual launches: NO control flow between the pixel's fetches and its stores (rows / chunks outside the tensor fetch a valid
            // address and skip only the store), and the NEXT pixel's residual is fetched before this pixel's stores go out: the wait for it then
            // leaves those stores in flight (loads and stores retire through one in-order counter).
            auto fetch = [&](int pt_, YpResRaw<DT, 8> (&r)[CT][2]) {
                const int m_ = m0 + g * (BP / NG) + wp * TP + pt_ * 32 + lr;
                const YpOutRow row_ = yp_out_row<DT, OUT_F32>(a, m_ < a.M ? m_ : a.M - 1);
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                    for (int h8 = 0; h8 < 2; ++h8) {
                        const int nc = nb + ct * 16 + h8 * 8;
                        r[ct][h8] = yp_res_fetch<DT, 8>(row_.rp, nc < a.Cout ? nc : 0);
                    }
            };
            YpResRaw<DT, 8> raw[CT][2];
            if (pt == 0) fetch(0, res_next);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int h8 = 0; h8 < 2; ++h8) raw[ct][h8] = res_next[ct][h8];
            if (pt + 1 < PT) fetch(pt + 1, res_next);
            const YpOutRow row = yp_out_row<DT, OUT_F32>(a, m < a.M ? m : a.M - 1);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
                for (int h8 = 0; h8 < 2; ++h8) {
                    const int nc = nb + ct * 16 + h8 * 8;
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        float x = acc[ct][pt][h8 * 8 + j] + bias[ct * 16 + h8 * 8 + j];
                        if (a.act == YP_ACT_SILU) x = yp_silu(x);
                        v[j] = x;
                    }
                    yp_res_add<DT, 8>(raw[ct][h8], v);
                    if (m < a.M && nc < a.Cout) yp_store_chunk_at<DT, OUT_F32, 8, 0>(a, row, nc, v);
                }
            }
        } else {
        if (m >= a.M) continue;
        const YpOutRow row = yp_out_row<DT, OUT_F32>(a, m);
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
                if (!(probe & 32) || v[0] == 1.2345e-30f) yp_store_chunk_at<DT, OUT_F32, 8, 0>(a, row, nc, v);
            }
        }
        }
    }
    };
    // 16-bit output without residual / pixel remap, waves of TP x 64 outputs: the wave's block goes through the (now idle) pipeline LDS so that a
    // store instruction writes 8 whole 128-byte lines (lane -> row lane / 8, 16-byte chunk lane % 8) instead of 16-byte pieces of 32 lines
    // (one wave per SIMD: epilogue 18.4 k -> 15.8 k clocks at 224 rows).  Same fp32 arithmetic, same single rounding: bit-identical outputs.
    bool staged = false;
    if constexpr (!OUT_F32 && WP == 1 && CT == 2) {
        if (!a.has_res && !a.out_sub) {
            staged = true;
            using sc = typename Elem<DT>::scalar;
            __syncthreads();                                          // (every wave is done with the last k tile's fragments)
            char* const my = smem + wave * (TP * 128);                // wave-private: TP rows x 128 bytes (64 channels); NW * TP = 4 * BP rows in all
#pragma unroll
            for (int pt = 0; pt < PT; ++pt)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                    for (int h8 = 0; h8 < 2; ++h8) {
                        u32x4 pk;
                        sc* e = reinterpret_cast<sc*>(&pk);
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            float x = acc[ct][pt][h8 * 8 + j] + bias[ct * 16 + h8 * 8 + j];
                            if (a.act == YP_ACT_SILU) x = yp_silu(x);
                            e[j] = (sc)x;
                        }
                        const int c = hh * (2 * CT) + ct * 2 + h8;    // 16-byte chunk of the pixel's 64 channels
                        *reinterpret_cast<u32x4*>(my + (pt * 32 + lr) * 128 + ((c ^ (lr & 7)) << 4)) = pk;
                    }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // (own writes only: the block is wave-private)
            const int rr = lane >> 3, cc = lane & 7;
            const int nc = n0 + wc * TC + cc * 8;
            const bool second = nc >= a.split;
            char* const obase = second ? a.out2 : a.out;
            const size_t ocs = second ? (size_t)a.out2_cs : (size_t)a.out_cs;
            const long oco = second ? (long)a.out2_co - (long)a.split + nc : (long)a.out_co + nc;
            constexpr int NB = 8;                                     // rows-of-8 per batch: NB reads in flight, then NB stores (TP / 16: no faster)
#pragma unroll
            for (int j0 = 0; j0 < TP / 8; j0 += NB) {
                u32x4 d[NB];
#pragma unroll
                for (int j = 0; j < NB; ++j)
                    if (j0 + j < TP / 8) d[j] = *reinterpret_cast<const u32x4*>(my + ((j0 + j) * 8 + rr) * 128 + ((cc ^ rr) << 4));
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    if (j0 + j >= TP / 8) continue;
                    const int m = m0 + g * (BP / NG) + wp * TP + (j0 + j) * 8 + rr;
                    if (m < a.M && nc < a.Cout && !((probe & 32) && d[j][0] != 0x12345678u))
                        *reinterpret_cast<u32x4*>(obase + ((size_t)m * ocs + oco) * 2) = d[j];
                }
            }
        }
    }
    if (!staged) YP_RES_DISPATCH(a, epilogue);
    YP8_TSX(5);
    YP8_TS_FLUSH2();
}

namespace {

template <int DT, bool OUT_F32, int BP, int BC, int WP, int WC, int NS, bool STATS, int SCHED = 0, int NW = 8>
hipError_t launch_mma8(const ConvKArgs& a, int nblk, hipStream_t st) {
    constexpr size_t lds = (size_t)NS * (BP + BC) * 128;
    static_assert(lds <= 160 * 1024, "LDS budget");
    auto kern = conv_mma8_kernel<DT, OUT_F32, BP, BC, WP, WC, NS, STATS, SCHED, NW>;
    static YpLdsAttr attr;        // per instantiation, per device
    if (hipError_t e = yp_set_max_lds(attr, (const void*)kern, (int)lds); e != hipSuccess) return e;
    kern<<<nblk, NW * 64, lds, st>>>(a);
    return hipGetLastError();
}

template <int DT, bool OUT_F32, bool STATS>
hipError_t dispatch_mma8(int tile, const ConvKArgs& a, int nblk, hipStream_t st) {
    switch (tile) {
        case 41: return launch_mma8<DT, OUT_F32, 256, 256, 1, 4, 2, STATS>(a, nblk, st);
        case 42: return launch_mma8<DT, OUT_F32, 256, 128, 2, 2, 3, STATS>(a, nblk, st);
        case 43: return launch_mma8<DT, OUT_F32, 128, 256, 1, 4, 3, STATS>(a, nblk, st);
        case 44: return launch_mma8<DT, OUT_F32, 128, 128, 1, 4, 2, STATS>(a, nblk, st);
        case 57: return launch_mma8<DT, OUT_F32, 256, 256, 1, 4, 2, STATS, 7>(a, nblk, st);
        case 58: return launch_mma8<DT, OUT_F32, 256, 256, 1, 4, 2, STATS, 8>(a, nblk, st);
        // one wave per SIMD (NW = 4), BP x 256 tiles with BP = 256 / 224  (192 and 160 rows were built and measured: 111 / 431 us against 69 on the
        // 256 -> 256 3x3 layer at 40 x 40 x 32 -- the DMA path per MFMA grows as the tile shrinks -- and removed; a wave of 128 x 128 outputs, 8 fragment
        // reads per 16 MFMAs, needs 256 accumulator + 64 double-buffered fragment registers beside the addressing state and spills: 1 290 us)
        case 61: return launch_mma8<DT, OUT_F32, 256, 256, 1, 4, 2, STATS, 8, 4>(a, nblk, st);
        case 62: return launch_mma8<DT, OUT_F32, 224, 256, 1, 4, 2, STATS, 8, 4>(a, nblk, st);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace

bool yp_mma8_tile_dims(int tile, int* bp, int* bc, int* stat_rows_px) {
    int p = 0, c = 0, r = 0;
    switch (tile) {
        case 41: case 57: case 58: p = 256; c = 256; r = 128; break;
        case 42: p = 256; c = 128; r = 64; break;
        case 43: p = 128; c = 256; r = 64; break;
        case 44: p = 128; c = 128; r = 64; break;
        case 61: p = 256; c = 256; r = 256; break;
        case 62: p = 224; c = 256; r = 224; break;
        default: return false;
    }
    if (bp) *bp = p;
    if (bc) *bc = c;
    if (stat_rows_px) *stat_rows_px = r;
    return true;
}

hipError_t yp_mma8_launch(int tile, int dtype, bool out_f32, bool stats, const ConvKArgs& a0, int nblk, hipStream_t st) {
    ConvKArgs a = a0;
#ifdef YP_PROBE8
    { const char* e = getenv("YP_MMA8_PROBE"); a.probe = e ? atoi(e) : 0; }
#endif
    if (dtype == YP_F16) {
        if (stats) return dispatch_mma8<YP_F16, false, true>(tile, a, nblk, st);
        return out_f32 ? dispatch_mma8<YP_F16, true, false>(tile, a, nblk, st) : dispatch_mma8<YP_F16, false, false>(tile, a, nblk, st);
    }
    if (dtype == YP_BF16) {
        if (stats) return dispatch_mma8<YP_BF16, false, true>(tile, a, nblk, st);
        return out_f32 ? dispatch_mma8<YP_BF16, true, false>(tile, a, nblk, st) : dispatch_mma8<YP_BF16, false, false>(tile, a, nblk, st);
    }
    if ((dtype == YP_FP8 || dtype == YP_FP8_BF8) && tile == 57 && !out_f32) {        // 8-bit inputs: the 256 x 256 two-phase kernel only
        if (dtype == YP_FP8) return stats ? launch_mma8<YP_FP8, false, 256, 256, 1, 4, 2, true, 7>(a, nblk, st) : launch_mma8<YP_FP8, false, 256, 256, 1, 4, 2, false, 7>(a, nblk, st);
        if (!stats) return launch_mma8<YP_FP8_BF8, false, 256, 256, 1, 4, 2, false, 7>(a, nblk, st);
    }
    return hipErrorInvalidValue;
}
