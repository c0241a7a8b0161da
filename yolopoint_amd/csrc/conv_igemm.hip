// Fused implicit-GEMM convolution for gfx950 (MI355X), NHWC, MFMA 16x16.
//
//   out[m, n] = [res[m, n] +] act( sum_k X[m, k] * Wt[n, k] + bias[n] )
//   m = (b, ho, wo) output pixel, n = output channel, k = (r, s, c) filter tap x input channel.
//
// There is no im2col buffer: the loader gathers 16-byte channel chunks of the input pixel each
// (m, tap) touches straight into LDS (zero-filled where the tap falls into the padding) — the
// "A" side of the GEMM exists only as an addressing rule.  Two input sources are supported so
// that torch.cat((a, b), 1) feeding a conv is never materialised, and a source can be read
// through a 2x nearest upsample (pixel (y,x) -> (y>>1, x>>1)).
//
// MFMA operand roles are swapped w.r.t. the textbook GEMM: the weight tile is the MFMA "A"
// operand (rows = output channels) and the pixel tile the "B" operand (columns = pixels).  The
// accumulator layout is then D[channel = 4*(lane/16)+reg][pixel = lane%16]: each lane owns 4
// consecutive channels of one pixel per fragment.  Weight rows are permuted when they are
// written to LDS so that the FN fragments of a wave interleave to 4*FN *consecutive* channels
// per lane -> the epilogue stores (and the residual loads) are 16-byte vectors of one pixel,
// contiguous in NHWC.
//
// Reference call sites replaced: models/common.py:22-34 (Conv), :88-89 (Bottleneck add),
// :135,:229 (cat), models/YOLOPoint.py:186,195 (head convs), models/yolo.py:51 (Detect.m).
#include "yp_internal.h"
#include "conv_common.h"
#ifndef YP_PART
#define YP_PART 0      // 0: one translation unit (probe builds); 1 / 2: see "Two translation units" above the host code
#endif

struct __attribute__((packed, aligned(4))) YpF4U { f32x4 v; };      // a 16-byte vector at a 4-byte-aligned address (Detect rows: 85 floats)

// Probe builds (-DYP_TIMELINE, tools/probe/timeline.py): workgroup 0 / lane 0 records the shader clock at phase boundaries.
#ifdef YP_TIMELINE
__device__ long long yp_timeline[64];
__device__ int yp_tl_block = 0;                            // the workgroup whose phase clocks are recorded (yp_debug_timeline_block)
__device__ int yp_probe_mode = 0;                          // elimination experiments on the generic kernel: 1 = no MFMA / fragment reads, 2 = pixel DMA
                                                           // reads the zero page, 4 = filter DMA reads the zero row
extern "C" int yp_debug_probe_mode(int m) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(yp_probe_mode), &m, sizeof(int)); }
#define YP_PROBE_MODE() __builtin_amdgcn_readfirstlane(yp_probe_mode)
__device__ unsigned long long yp_wg_times[3 * 16384];      // per workgroup: {entry, epilogue-stores-issued} on the 100 MHz wall clock, HW_ID | XCC_ID << 32
extern "C" int yp_debug_timeline(long long* out_host) { return (int)hipMemcpyFromSymbol(out_host, HIP_SYMBOL(yp_timeline), sizeof(long long) * 64); }
extern "C" int yp_debug_timeline_block(int b) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(yp_tl_block), &b, sizeof(int)); }
extern "C" int yp_debug_wg_times(unsigned long long* out_host, int nwg) { return (int)hipMemcpyFromSymbol(out_host, HIP_SYMBOL(yp_wg_times), sizeof(unsigned long long) * 3 * nwg); }
// (the selected block is read ONCE, at YP_TL(0) = kernel entry: a load of the __device__ variable inside the k loop is a VMEM operation
// that the compiler waits for with vmcnt(0) -- it drained the hand-counted DMA pipeline every iteration and made the probe build
// 1.5x slower than the production one)
#define YP_TL(i) do { if ((i) == 0) yp_tl_hit = (threadIdx.x == 0 && (int)blockIdx.x == __builtin_amdgcn_readfirstlane(yp_tl_block));                    \
        if (yp_tl_hit) yp_timeline[i] = __builtin_readcyclecounter();                                                                                        \
        if (((i) == 0 || (i) >= 41) && threadIdx.x == 0 && blockIdx.x < 16384) { yp_wg_times[blockIdx.x * 3 + ((i) == 0 ? 0 : 1)] = wall_clock64();        \
            if ((i) == 0) yp_wg_times[blockIdx.x * 3 + 2] = (unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) |               \
                                                               ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32); } } while (0)
#define YP_TL_DECL bool yp_tl_hit = false
#else
#define YP_TL(i) do {} while (0)
#define YP_TL_DECL do {} while (0)
#define YP_PROBE_MODE() 0
#endif


template <int DT, bool OUT_F32, int LPG, int RES = -1, typename AccFn>
__device__ __forceinline__ void yp_epilogue_pixel(const ConvKArgs& a, int m, int nb, const float (&bias)[LPG], AccFn acc) {
    constexpr int CW = LPG < 8 ? LPG : 8;
    const YpOutRow row = yp_out_row<DT, OUT_F32>(a, m);
#pragma unroll
    for (int h = 0; h < LPG / CW; ++h) {
        const int nc = nb + h * CW;
        if (nc >= a.Cout) continue;
        float v[CW];
#pragma unroll
        for (int j = 0; j < CW; ++j) {
            float x = acc(h * CW + j) + bias[h * CW + j];
            if (a.act == YP_ACT_SILU) x = yp_silu(x);
            v[j] = x;
        }
        yp_store_chunk_at<DT, OUT_F32, CW, RES>(a, row, nc, v);
    }
}

// Detect-head decode of one element (reference models/yolo.py:53-68): channel n = a*no + o of pixel
// (b, rem = y*nx + x) goes to x_out[b, a, y, x, o] untouched and, decoded, to z[b, row_off + (a*ny + y)*nx + x, o]:
//   xy = (2*sigmoid - 0.5 + grid) * stride, wh = (2*sigmoid)^2 * anchor_px, the rest = sigmoid.
// The anchor of a lane's (dynamic) anchor index, selected from the table held in registers.  Indexing a.det_anchor[] per lane is a GLOBAL load out
// of the kernel-argument segment whose wait, `vmcnt(0)`, also waits for every store issued before it: one write round trip per loop iteration of
// the Detect epilogue until round 5.
struct YpAnchors {
    float t[16];
    __device__ __forceinline__ void load(const ConvKArgs& a) {
#pragma unroll
        for (int i = 0; i < 16; ++i) t[i] = a.det_anchor[i];          // constant indices: scalar loads
    }
    __device__ __forceinline__ void pick(int an, float& aw, float& ah) const {
        aw = t[0]; ah = t[1];
#pragma unroll
        for (int k = 1; k < 8; ++k) { aw = an == k ? t[2 * k] : aw; ah = an == k ? t[2 * k + 1] : ah; }
    }
};
__device__ __forceinline__ void yp_detect_store(const ConvKArgs& a, const YpAnchors& anchors, int b, int rem, int y, int x, int n, float v) {
    const int no = a.det_no;
    const int an = (n * a.det_invno) >> 16, o = n - an * no;
    const size_t cell = (size_t)(b * a.det_na + an) * a.HoWo + rem;
    a.det_x[cell * no + o] = v;
    if (a.det_z != nullptr) {
        const float s = yp_sigmoid(v);
        float z, aw, ah;
        anchors.pick(an, aw, ah);
        if (o == 0) z = (s * 2.0f - 0.5f + (float)x) * a.det_stride;
        else if (o == 1) z = (s * 2.0f - 0.5f + (float)y) * a.det_stride;
        else if (o == 2) { const float t2 = s * 2.0f; z = t2 * t2 * aw; }
        else if (o == 3) { const float t2 = s * 2.0f; z = t2 * t2 * ah; }
        else z = s;
        const size_t row = (size_t)a.det_row_off + (size_t)an * a.HoWo + rem;
        a.det_z[((size_t)b * a.det_rows_total + row) * no + o] = z;
    }
}

template <int LPG>
__device__ __forceinline__ void yp_load_bias(const ConvKArgs& a, int nb, float (&bias)[LPG]) {
#pragma unroll
    for (int q = 0; q < LPG / 4; ++q) {     // Cout is a multiple of 8: groups of 4 channels are all-or-nothing
        f32x4 b4 = f32x4{0.f, 0.f, 0.f, 0.f};
        if (a.bias != nullptr && nb + 4 * q < a.Cout) b4 = *reinterpret_cast<const f32x4*>(a.bias + nb + 4 * q);
        bias[4 * q] = b4[0]; bias[4 * q + 1] = b4[1]; bias[4 * q + 2] = b4[2]; bias[4 * q + 3] = b4[3];
    }
}

// Recursive-halving reduction of per-lane channel sums over the 16 pixel lanes of a lane group (the BatchNorm-statistics epilogues): at
// offset O a lane keeps one half of its N channels and hands the other half to lane ^ O; with one channel left the remaining offsets are
// plain butterfly adds.  N and O are compile-time values, so every array index is static and sv / sq stay in registers (the run-time
// form of the same loop kept `n` in a variable: the compiler indexed the arrays dynamically and put them in SCRATCH memory -- 80-144
// bytes per lane in all 56 STATS instantiations of the generic and the halo kernel).  Same operations in the same order: same bits.
template <int N, int O>
struct YpHalve {
    template <int LPG>
    static __device__ __forceinline__ void run(float (&sv)[LPG], float (&sq)[LPG], int p, int& mych) {
        if constexpr (O < 16) {
            const bool up = (p & O) != 0;
            if constexpr (N > 1) {
                constexpr int hn = N / 2;
#pragma unroll
                for (int j = 0; j < hn; ++j) {
                    const float keep_s = up ? sv[hn + j] : sv[j], give_s = up ? sv[j] : sv[hn + j];
                    const float keep_q = up ? sq[hn + j] : sq[j], give_q = up ? sq[j] : sq[hn + j];
                    sv[j] = keep_s + yp_xor_lane<O>(give_s);
                    sq[j] = keep_q + yp_xor_lane<O>(give_q);
                }
                mych += up ? hn : 0;
                YpHalve<hn, O * 2>::run(sv, sq, p, mych);
            } else {
                sv[0] += yp_xor_lane<O>(sv[0]);
                sq[0] += yp_xor_lane<O>(sq[0]);
                YpHalve<1, O * 2>::run(sv, sq, p, mych);
            }
        }
    }
};

#ifdef YP_PROBE_BNFUSE
// Probe build (make probebn; tools/probe/bnfuse_bench.py): the 1x1 convolution applies x = silu(scale[c] * y + shift[c]) to its input tile in
// LDS between the DMA's landing and the fragment reads -- the consumer-side BatchNorm + SiLU of review item 4 (rounds 2-5), as a measurement.
__device__ const float* yp_xf_scale = nullptr;
__device__ const float* yp_xf_shift = nullptr;
extern "C" int yp_debug_set_xform(const float* scale, const float* shift) {
    if (hipMemcpyToSymbol(HIP_SYMBOL(yp_xf_scale), &scale, sizeof(scale)) != hipSuccess) return -1;
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(yp_xf_shift), &shift, sizeof(shift));
}
#endif

// 16 zero bytes: out-of-image taps / padded k / padded filter rows are fetched from here, so every
// LDS-DMA lane always has a valid source and no predication or LDS pre-clearing is needed.
__device__ __attribute__((aligned(16))) unsigned int yp_zero16[4] = {0u, 0u, 0u, 0u};


// Pipeline: NS LDS stages; k tile t+NS-1 is in flight (LDS-DMA, global_load_lds_dwordx4: no VGPR
// round trip) while tile t is multiplied.  One raw s_barrier per k tile; waits are counted
// s_waitcnt vmcnt(N) so younger tiles stay in flight across the barrier.
//
// LDS image of one stage: (BM + BN) dense 64-byte rows (pixels first, then filter rows).  One DMA
// instruction moves 1 KiB = 16 rows x 4 chunks, lane l -> row l/4, physical chunk l%4 (the hardware
// places lanes linearly).  Bank conflicts of the 16-byte fragment reads are removed by an XOR
// swizzle applied on the SOURCE side: physical chunk j of row r holds logical chunk j ^ swz(r),
// swz(r) = {0,0,3,3}[(r/4)%4]; readers apply the same involution.
//
// KPB > 0 selects the second-generation main loop (16-bit element types, FAST addressing): a ring stage holds KPB consecutive k tiles,
// so there is ONE barrier per KPB tiles instead of one per tile, the MFMA fragments are double-buffered in registers (the LDS reads of
// sub-tile q+1 are in flight while the MFMAs of sub-tile q issue) and the refill of a ring slot is issued right behind the barrier
// that frees it.  Measured on the first-generation loop (tools/probe/wg_census.py): one k tile of a 128 x 128 workgroup took 0.56 us
// whatever the ring depth -- 1340 clocks for 256 clocks of MFMA work, a serial barrier -> DMA issue -> LDS read -> MFMA chain per tile.
//
// WSK ("waves split k", with KPB = 4): the four waves of the workgroup each take ONE of the four k tiles of a stage and multiply it
// against the WHOLE BM x BN tile (16 MFMAs per wave and barrier for 64 x 64 instead of 4: short-M / deep-K layers keep small tiles for
// the sake of enough workgroups, and were issue-bound on ~50 instructions per 4 MFMAs); the four partial tiles are summed through LDS
// before the unchanged epilogue.  The k order of the fp32 sums changes (4 interleaved partial sums), so results differ from the
// sequential loop in the last bits.
template <int DT, bool OUT_F32, bool FAST, bool DETECT, int BM, int BN, int WAVES_M, int WAVES_N, int NS, bool STATS = false, int KPB = 0, bool WSK = false>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvKArgs a) {
    using E = Elem<DT>;
    using frag_t = typename E::frag;
    constexpr int EB = E::BYTES;
    constexpr int CE = 16 / EB;          // elements per 16-byte chunk
    constexpr int BK = 64 / EB;          // k elements per tile (64 bytes of k per row)
    constexpr int ROWB = 64;
    constexpr int SLOTS_A = BM / 16, SLOTS_B = BN / 16, SLOTS = SLOTS_A + SLOTS_B;
    constexpr int NLA = SLOTS_A / 4;                 // DMA instructions per wave per tile (pixels)
    constexpr int NLB = (SLOTS_B + 3) / 4;           //                                   (filter rows)
    constexpr int NL = NLA + NLB;
    constexpr int KP = KPB > 0 ? KPB : 1;            // k tiles per ring stage (= per barrier)
    constexpr int SUB = SLOTS * 1024;                // LDS image of one k tile
    constexpr int STAGE = KP * SUB;
    static_assert(KPB == 0 || (FAST && DT != YP_F32 && KPB % 2 == 0), "the second-generation loop serves the 16-bit FAST path; KPB even");
    static_assert(!WSK || (KPB == 4 && !STATS), "waves-split-k: one k tile of a 4-tile stage per wave");
    constexpr int TM = BM / WAVES_M, TN = BN / WAVES_N;
    constexpr int FM = TM / 16, FN = TN / 16;
    constexpr int LPG = 4 * FN;          // consecutive channels owned by one lane
    static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
    static_assert(BM % 64 == 0 && FM >= 1 && FN >= 1 && NS >= 2 && NS <= 8, "unsupported tile");
    static_assert(KP * NL * (NS - 1) <= 60, "vmcnt immediate range");

    extern __shared__ __attribute__((aligned(1024))) char smem[];      // NS * STAGE bytes (dynamic: deep rings exceed 64 KiB)
    YP_TL_DECL;
    YP_TL(0);

    // ---- XCD-aware tile mapping: workgroup b runs on XCD b%8; give each XCD a contiguous run of
    // logical tiles so that tiles sharing input pixels / filter rows hit the same L2.
    const int nblk = gridDim.x;
    int logical;
    {
        const int bid = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    // (exact magic division where the host could guarantee it -- yp_conv2d_launch: a.mg_* != 0 -- instead of the ~35-instruction run-time
    // division sequences: the setup of a launch is 0.8-1.7 us of a 4-6 us kernel on the short layers)
    const int tile_m = a.mg_tn ? (int)__umulhi((unsigned)logical, a.mg_tn) : logical / a.tiles_n;
    const int tile_n = logical - tile_m * a.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    YP_TL(50);                                                    // (first kernel arguments have arrived, tile decoded)

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int lrow = lane >> 2;                                   // row inside a 16-row DMA slot
    const int jl = (lane & 3) ^ ((0x3300 >> ((lane >> 4) * 4)) & 3);   // logical chunk this lane fetches

    // ---- per-slot gather state (fixed for the whole k loop)
    int hi0[NLA], wi0[NLA], bb[NLA];
#pragma unroll
    for (int i = 0; i < NLA; ++i) {
        const int m = m0 + (wave + 4 * i) * 16 + lrow;
        if (m < a.M) {
            const int b = a.mg_howo ? (int)__umulhi((unsigned)m, a.mg_howo) : m / a.HoWo;
            const int rem = m - b * a.HoWo;
            const int ho = a.mg_howo ? (int)__umulhi((unsigned)rem, a.mg_wo) : rem / a.Wo;
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
    YP_TL(51);                                                    // (pixel rows decoded)
    const char* b_src[NLB];
    unsigned b_off[NLB];
    int b_slot[NLB];
#pragma unroll
    for (int i = 0; i < NLB; ++i) {
        int sl = __builtin_amdgcn_readfirstlane(wave) + 4 * i;
        if (sl > SLOTS_B - 1) sl = SLOTS_B - 1;       // surplus instructions re-fetch the last slot (same bytes)
        b_slot[i] = SLOTS_A + sl;
        const int rho = sl * 16 + lrow;               // LDS row = MFMA row order
        // MFMA row (wn, f, g, r) holds channel wn*TN + g*LPG + f*4 + r, so that a lane's FN fragments
        // interleave to LPG consecutive channels in the epilogue
        const int wn = rho / TN, q = rho % TN;
        const int f = q >> 4, g = (q & 15) >> 2, r = q & 3;
        const int n = n0 + wn * TN + g * LPG + f * 4 + r;
        b_src[i] = (n < a.Npad) ? a.wgt + ((size_t)n * a.Kpad + jl * CE) * EB : nullptr;
        b_off[i] = ((n < a.Npad) ? (unsigned)n * (unsigned)a.Kpad * EB : a.wgt_zrow) + jl * 16;
    }
    // ---- split-K: this workgroup reduces k tiles [kt0, kt1)
    const int nk_all = (a.Kreal + BK - 1) / BK;
    int kt0 = 0, kt1 = nk_all;
    if (a.ksplit > 1) {                   // (keeps the two 64-bit divisions out of the setup of every ordinary launch)
        kt0 = (int)(((long)nk_all * blockIdx.y) / a.ksplit);
        kt1 = (int)(((long)nk_all * (blockIdx.y + 1)) / a.ksplit);
    }
    // ---- per-lane filter-tap state for logical chunk jl: k = kt*BK + jl*CE = tap*Cin + kc, tap = r*S + s
    int kc, tap;
    {
        const int k = kt0 * BK + jl * CE;
        if (k < a.Cin) { tap = 0; kc = k; }          // the common case (first k tile, Cin >= 64 bytes of channels): no division
        else { tap = k / a.Cin; kc = k - tap * a.Cin; }
    }
    const char* const zero = reinterpret_cast<const char*>(yp_zero16);

    // Pin the per-source kernel arguments in SGPRs.  Without this, clang turns `s0 ? a.in0_x : a.in1_x`
    // into a VECTOR load from a selected kernarg address — a VMEM operation inside the k loop that
    // would both stall it and corrupt the hand-counted vmcnt waits below.
#define YP_PIN(T, name) T name = a.name; asm volatile("" : "+s"(name))
    YP_PIN(const char*, in0); YP_PIN(const char*, in1);
    YP_PIN(int, in0_cs); YP_PIN(int, in1_cs); YP_PIN(int, in0_co); YP_PIN(int, in1_co); YP_PIN(int, in0_C);
    YP_PIN(int, in0_ups); YP_PIN(int, in1_ups); YP_PIN(int, in0_H); YP_PIN(int, in1_H); YP_PIN(int, in0_W); YP_PIN(int, in1_W);
    YP_PIN(int, Hi); YP_PIN(int, Wi); YP_PIN(int, Cin); YP_PIN(int, S); YP_PIN(int, invS); YP_PIN(int, dt); YP_PIN(int, dc);
#undef YP_PIN
    YP_TL(52);                                                    // (filter offsets, arguments pinned)
    const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) char*)smem);
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    // FAST path state: every lane of the workgroup is in the same filter tap (Cin % BK == 0), so the tap
    // decode, the source select and the channel offset live on the scalar unit.
    int s_tap = 0, s_c0 = 0;
    if (kt0 > 0) { s_tap = (kt0 * BK) / a.Cin; s_c0 = kt0 * BK - s_tap * a.Cin; }
    YP_PIN2(unsigned, in0_zoff); YP_PIN2(unsigned, in1_zoff); YP_PIN2(const char*, wgt);
    YP_PIN2(int, RS);
    const int probe_mode = YP_PROBE_MODE();
    unsigned seg_voff[NLA];            // FAST path issue state: per-lane pixel offsets of the current segment,
    const char* seg_base = in0;        // its scalar base (advances 64 bytes per k tile), the tiles left in it, its zero-page offset
    int seg_left = 0;
    unsigned seg_zoff = 0;
#pragma unroll
    for (int i = 0; i < NLA; ++i) seg_voff[i] = 0;
    auto issue_tile = [&](int kt, unsigned lds_off) {
        const unsigned sbase = lds0 + lds_off;
        if constexpr (FAST) {
            // A SEGMENT = the run of k tiles inside one (filter tap, source tensor): there the per-lane pixel offsets are constant and
            // only the channel position moves, which rides in the scalar base pointer.  The per-lane work (tap decode, bounds test,
            // pixel offset: ~35 instructions per slot) is done once per segment, a k tile inside a segment costs two scalar adds per
            // DMA instruction -- a 1x1 convolution is ONE segment per source.  (Before: ~95 instructions per k tile around 4 MFMAs; with
            // every load redirected to the zero page and the MFMAs removed the kernel took as long as with them: it was issue-bound.)
            const bool past = KPB > 0 && kt >= kt1;        // a k tile behind this workgroup's k range (stage padding): all-zero operands
            if (seg_left == 0) {
                const int kr = (s_tap * invS) >> 16;
                const int ks = s_tap - kr * S;
                const bool s0 = s_c0 < in0_C;
                const char* base = s0 ? in0 : in1;
                const int cs = s0 ? in0_cs : in1_cs;
                const int ups = s0 ? in0_ups : in1_ups;
                const int Hp = s0 ? in0_H : in1_H;
                const int Wp = s0 ? in0_W : in1_W;
                const bool zs = s0 && a.in0_zs;
                const int c_in_src = s0 ? s_c0 : s_c0 - in0_C;
                seg_zoff = s0 ? in0_zoff : in1_zoff;
                seg_left = ((s0 ? in0_C : Cin - in0_C) - c_in_src) / BK;
                seg_base = base + (size_t)((s0 ? in0_co : in1_co) + c_in_src) * EB;
                const int csb = cs * EB;
                const unsigned lanec = (unsigned)jl * 16u;
#pragma unroll
                for (int i = 0; i < NLA; ++i) {
                    const int hi = hi0[i] + kr, wi = wi0[i] + ks;
                    const bool ok = (unsigned)hi < (unsigned)Hi && (unsigned)wi < (unsigned)Wi && !(zs && ((hi | wi) & 1));
                    const int pix = (bb[i] * Hp + (hi >> ups)) * Wp + (wi >> ups);
                    // (an out-of-image lane reads seg_base + zoff: the channel position moves it at most one pixel of channels into the
                    // buffer's zero tail, which is one pixel + 64 elements long)
                    seg_voff[i] = ok ? (unsigned)(pix * csb) + lanec : seg_zoff;
                }
            }
            const bool zero_pix = past;
#ifndef YP_PROBE_NODMA
            {
#pragma unroll
            for (int i = 0; i < NLA; ++i) yp_glds16_s(seg_base, zero_pix ? seg_zoff : seg_voff[i], sbase + (wave_u + 4 * i) * 1024);
            }
#endif
            const char* wk = past ? wgt : wgt + (size_t)kt * (BK * EB);
#ifndef YP_PROBE_NODMA
            {
#pragma unroll
            for (int i = 0; i < NLB; ++i) yp_glds16_s(wk, past ? a.wgt_zrow + (unsigned)jl * 16u : b_off[i], sbase + b_slot[i] * 1024);
            }
#endif
            if (!past) {
                seg_base += BK * EB;
                --seg_left;
                s_c0 += BK;
                if (s_c0 >= Cin) { s_c0 -= Cin; ++s_tap; }
            }
            return;
        }
        const int kr = tap / S;                       // generic path: any filter size (wgrad "filters" are Ho x Wo)
        const int ks = tap - kr * S;
        const bool tapok = tap < RS;
        const bool s0 = kc < in0_C;
        const char* base = s0 ? in0 : in1;
        const int cs = s0 ? in0_cs : in1_cs;
        const int cc = s0 ? (in0_co + kc) : (in1_co + kc - in0_C);
        const int ups = s0 ? in0_ups : in1_ups;
        const int Hp = s0 ? in0_H : in1_H;
        const int Wp = s0 ? in0_W : in1_W;
        const bool zs = s0 && a.in0_zs;
#pragma unroll
        for (int i = 0; i < NLA; ++i) {
            const int hi = hi0[i] + kr * a.dil_h, wi = wi0[i] + ks * a.dil_w;
            const bool ok = tapok && (unsigned)hi < (unsigned)Hi && (unsigned)wi < (unsigned)Wi && !(zs && ((hi | wi) & 1));
            const long pix = ((long)bb[i] * Hp + (hi >> ups)) * Wp + (wi >> ups);
            const char* src = base + (pix * cs + cc) * EB;       // only dereferenced when ok
            yp_glds16(ok ? src : zero, sbase + (wave_u + 4 * i) * 1024);
        }
#pragma unroll
        for (int i = 0; i < NLB; ++i) {
            const char* src = b_src[i] + (size_t)kt * BK * EB;
            yp_glds16(b_src[i] != nullptr ? src : zero, sbase + b_slot[i] * 1024);
        }
        // advance to the next k tile: k += BK  ->  (tap, kc) += (BK / Cin, BK % Cin) with carry
        kc += dc;
        const bool carry = kc >= Cin;
        kc -= carry ? Cin : 0;
        tap += dt + (carry ? 1 : 0);
    };

    const int wm = wave % WAVES_M, wn = wave / WAVES_M;
    const int p = lane & 15, g = lane >> 4;
    const int swr = (0x3300 >> ((p >> 2) * 4)) & 3;               // read-side swizzle of row p (+16k)
    // byte offset of this lane's fragment inside a row, for k step kk
    auto koff = [&](int kk) -> int {
        if constexpr (DT == YP_F32) return ((kk ^ swr) << 4) + g * 4;   // chunk = k step, element g
        else return ((g ^ swr) << 4);                                    // chunk = lane group (8 elements)
    };
    const int a_rd = (wm * TM + p) * ROWB;
    const int b_rd = (BM + wn * TN + p) * ROWB;

    f32x4 acc[FN][FM];
#pragma unroll
    for (int f = 0; f < FN; ++f)
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) acc[f][fm] = f32x4{0.f, 0.f, 0.f, 0.f};

    // bias is fetched before the pipeline starts (older than every DMA, so the counted waits stay valid)
    const int nb = n0 + wn * TN + g * LPG;
    float bias[LPG];
    yp_load_bias<LPG>(a, nb, bias);

    const int nk = kt1 - kt0;              // tiles are numbered relative to kt0 below; the filter offset uses kt0 + kt
    YP_TL(1);
    if constexpr (WSK) {
        constexpr int FNT = BN / 16, FMT = BM / 16;                // every wave multiplies the whole tile
        constexpr int NLS = KP * NL;
        static_assert(4 * FNT * FMT * 1024 <= NS * STAGE, "the partial tiles are summed through the pipeline LDS");
        const int nks = (nk + KP - 1) / KP;
        f32x4 pacc[FNT][FMT];
#pragma unroll
        for (int fa = 0; fa < FNT; ++fa)
#pragma unroll
            for (int fb = 0; fb < FMT; ++fb) pacc[fa][fb] = f32x4{0.f, 0.f, 0.f, 0.f};
        auto issue_stage = [&](int ks) {
            const unsigned base = (unsigned)(ks % NS) * STAGE;
#pragma unroll
            for (int j = 0; j < KP; ++j) issue_tile(kt0 + ks * KP + j, base + j * SUB);
        };
        const int ko = koff(0);
        const int rd_p = p * ROWB + ko;
#pragma unroll
        for (int sidx = 0; sidx < NS - 1; ++sidx)
            if (sidx < nks) issue_stage(sidx);
        for (int ks = 0; ks < nks; ++ks) {
            int younger = nks - 1 - ks;
            if (younger > NS - 2) younger = NS - 2;
            if (younger <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLS) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NS > 3 ? 2 * NLS : NLS) : "memory");
            static_assert(NS <= 4, "waves-split-k: rings of up to 4 stages");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (reads of stage ks-1 retired: its slot is refilled behind the barrier)
            __builtin_amdgcn_s_barrier();
            if (ks < 20) YP_TL(2 + ks);
            if (ks + NS - 1 < nks) issue_stage(ks + NS - 1);          // into the slot every wave finished reading before this barrier
            const char* sub = smem + (ks % NS) * STAGE + wave_u * SUB + rd_p;
            frag_t wf[FNT], xf[FMT];
#pragma unroll
            for (int fa = 0; fa < FNT; ++fa) wf[fa] = *reinterpret_cast<const frag_t*>(sub + (BM + fa * 16) * ROWB);
#pragma unroll
            for (int fb = 0; fb < FMT; ++fb) xf[fb] = *reinterpret_cast<const frag_t*>(sub + fb * 16 * ROWB);
#pragma unroll
            for (int fa = 0; fa < FNT; ++fa)
#pragma unroll
                for (int fb = 0; fb < FMT; ++fb) pacc[fa][fb] = E::mma(wf[fa], xf[fb], pacc[fa][fb]);
        }
        // ---- sum the four partial tiles: every wave parks its tile in LDS, wave (wm, wn) collects the fragments its epilogue owns
        __syncthreads();
        f32x4* red = reinterpret_cast<f32x4*>(smem);
#pragma unroll
        for (int fa = 0; fa < FNT; ++fa)
#pragma unroll
            for (int fb = 0; fb < FMT; ++fb) red[((wave_u * FNT + fa) * FMT + fb) * 64 + lane] = pacc[fa][fb];
        __syncthreads();
#pragma unroll
        for (int f = 0; f < FN; ++f)
#pragma unroll
            for (int fm = 0; fm < FM; ++fm) {
                const int fa = wn * FN + f, fb = wm * FM + fm;
                f32x4 v = red[((0 * FNT + fa) * FMT + fb) * 64 + lane];
#pragma unroll
                for (int w = 1; w < 4; ++w) {
                    const f32x4 u = red[((w * FNT + fa) * FMT + fb) * 64 + lane];
                    v[0] += u[0]; v[1] += u[1]; v[2] += u[2]; v[3] += u[3];
                }
                acc[f][fm] = v;
            }
        if constexpr (DETECT) __syncthreads();        // the Detect epilogue stages its tile in the same LDS
    } else if constexpr (KPB > 0) {
        // ---- second-generation loop.  Stage ks = k tiles [ks*KP, (ks+1)*KP) (zero tiles behind the end), ring slot ks % NS.
        constexpr int NLS = KP * NL;                              // DMA instructions per wave per stage
        const int nks = (nk + KP - 1) / KP;
        auto issue_stage = [&](int ks) {
            const unsigned base = (unsigned)(ks % NS) * STAGE;
#pragma unroll
            for (int j = 0; j < KP; ++j) issue_tile(kt0 + ks * KP + j, base + j * SUB);
        };
        auto wait_stages = [&](int younger) {                     // all but the `younger` most recent stages have landed
            if (younger <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLS) : "memory");
            else if (younger == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NS > 2 ? 2 * NLS : 0) : "memory");
            else if (younger == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NS > 3 ? 3 * NLS : 0) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NS > 4 ? 4 * NLS : 0) : "memory");
        };
        static_assert(NS <= 6, "wait_stages covers rings of up to 6 stages");
        const int ko = koff(0);
        auto load_frags = [&](const char* sub, frag_t (&wf)[FN], frag_t (&xf)[FM]) {
#pragma unroll
            for (int f = 0; f < FN; ++f) wf[f] = *reinterpret_cast<const frag_t*>(sub + b_rd + f * 16 * ROWB + ko);
#pragma unroll
            for (int fm = 0; fm < FM; ++fm) xf[fm] = *reinterpret_cast<const frag_t*>(sub + a_rd + fm * 16 * ROWB + ko);
        };
        auto mma_all = [&](const frag_t (&wf)[FN], const frag_t (&xf)[FM]) {
#pragma unroll
            for (int f = 0; f < FN; ++f)
#pragma unroll
                for (int fm = 0; fm < FM; ++fm) acc[f][fm] = E::mma(wf[f], xf[fm], acc[f][fm]);
        };
#pragma unroll
        for (int sidx = 0; sidx < NS; ++sidx)
            if (sidx < nks) issue_stage(sidx);
        wait_stages((nks < NS ? nks : NS) - 1);
        __builtin_amdgcn_s_barrier();
        frag_t wfA[FN], xfA[FM], wfB[FN], xfB[FM];
        load_frags(smem, wfA, xfA);
        for (int ks = 0; ks < nks; ++ks) {
            const char* st = smem + (ks % NS) * STAGE;
            // the KP sub-tiles of this stage, fragments ping-ponging between the A and B register sets (KP is even or 1)
#pragma unroll
            for (int j = 0; j < KP; ++j) {
                const bool cur_is_a = (j & 1) == 0;
                auto step = [&](frag_t (&wc)[FN], frag_t (&xc)[FM], frag_t (&wn)[FN], frag_t (&xn)[FM]) {
                    if (j + 1 < KP) {
                        load_frags(st + (j + 1) * SUB, wn, xn);                 // same stage: landed and published at the last barrier
                    } else if (ks + 1 < nks) {
                        // stage boundary: stage ks+1 must have landed for every wave, and every wave must be done READING stage ks
                        // (its last fragments are in registers: lgkmcnt(0)) before the slot is refilled
                        int younger = nks - 2 - ks;
                        if (younger > NS - 2) younger = NS - 2;
                        wait_stages(younger);
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_barrier();
                        if (ks < 20) YP_TL(2 + ks);
                        load_frags(smem + ((ks + 1) % NS) * STAGE, wn, xn);
                        if (ks + NS < nks) issue_stage(ks + NS);                 // into the slot of stage ks
                    }
                    mma_all(wc, xc);
                };
                if (cur_is_a) step(wfA, xfA, wfB, xfB); else step(wfB, xfB, wfA, xfA);
            }
        }
    } else {
#ifdef YP_PROBE_BNFUSE
    // (scale, shift) of every input channel into LDS behind the ring, before the pipeline starts (plain loads: older than every DMA)
    float* xf_tab = reinterpret_cast<float*>(smem + NS * STAGE);
    const float* xfs = yp_xf_scale;
    const float* xfh = yp_xf_shift;
    const bool xf_on = xfs != nullptr && FAST && E::BYTES == 2 && a.RS == 1;
    if (xf_on) {
        for (int i = t; i < a.Kpad; i += 256) { xf_tab[2 * i] = i < a.Cin ? xfs[i] : 0.f; xf_tab[2 * i + 1] = i < a.Cin ? xfh[i] : 0.f; }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
#endif
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s < nk) issue_tile(kt0 + s, s * STAGE);

    for (int kt = 0; kt < nk; ++kt) {
        // tile kt must have landed; up to NS-2 younger tiles may stay in flight
        int younger = nk - 1 - kt;
        if (younger > NS - 2) younger = NS - 2;
        if (younger <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL) : "memory");
        else if (NS <= 4 || younger == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NL) : "memory");
        else if (younger == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NS > 4 ? 3 * NL : 0) : "memory");
        else if (younger == 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NS > 5 ? 4 * NL : 0) : "memory");
        else if (younger == 5) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NS > 6 ? 5 * NL : 0) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NS > 7 ? 6 * NL : 0) : "memory");
#ifdef YP_PROBE_BNFUSE
        if constexpr (E::BYTES == 2 && FAST) {
            if (xf_on) {
                // this wave's own pixel slots of tile kt (its DMA instructions have landed: the counted wait above): lane l holds row l/4,
                // physical chunk l%4 = logical chunk jl = channels kt * 32 + 8 jl .. + 7
                using sc_t = typename E::scalar;
                char* sp = smem + (kt % NS) * STAGE;
                const float* tb = xf_tab + 2 * ((kt0 + kt) * BK + jl * 8);
                float cs[8], ch[8];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v4 = *reinterpret_cast<const f32x4*>(tb + 4 * q);
                    cs[2 * q] = v4[0]; ch[2 * q] = v4[1]; cs[2 * q + 1] = v4[2]; ch[2 * q + 1] = v4[3];
                }
#pragma unroll
                for (int i = 0; i < NLA; ++i) {
                    char* q_ = sp + (wave_u + 4 * i) * 1024 + lane * 16;
                    u32x4 raw = *reinterpret_cast<const u32x4*>(q_);
                    sc_t* e = reinterpret_cast<sc_t*>(&raw);
#pragma unroll
                    for (int j = 0; j < 8; ++j) e[j] = (sc_t)yp_silu((float)e[j] * cs[j] + ch[j]);
                    *reinterpret_cast<u32x4*>(q_) = raw;
                }
            }
        }
#endif
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (this wave's reads of tile kt-1 have retired before its stage is refilled)
#ifndef YP_PROBE_NOBAR
        __builtin_amdgcn_s_barrier();
#endif
        if (kt < 24) YP_TL(2 + kt);
        // every wave has finished reading tile kt-1: its stage can be refilled
        if (kt + NS - 1 < nk) issue_tile(kt0 + kt + NS - 1, ((kt + NS - 1) % NS) * STAGE);
        const char* s = smem + (kt % NS) * STAGE;
        if (probe_mode & 1) continue;
#pragma unroll
        for (int kk = 0; kk < BK / E::KM; ++kk) {
            frag_t wf[FN], xf[FM];
            const int ko = koff(kk);
#pragma unroll
            for (int f = 0; f < FN; ++f) wf[f] = *reinterpret_cast<const frag_t*>(s + b_rd + f * 16 * ROWB + ko);
#pragma unroll
            for (int fm = 0; fm < FM; ++fm) xf[fm] = *reinterpret_cast<const frag_t*>(s + a_rd + fm * 16 * ROWB + ko);
#pragma unroll
            for (int f = 0; f < FN; ++f)
#pragma unroll
                for (int fm = 0; fm < FM; ++fm) acc[f][fm] = E::mma(wf[f], xf[fm], acc[f][fm]);
        }
    }
    }

    if constexpr (E::BYTES == 1) {          // 8-bit inputs: back to real units (dequantisation scales of activation and filter)
        const float osc = a.scale_in[0] * a.scale_w[0];
#pragma unroll
        for (int f = 0; f < FN; ++f)
#pragma unroll
            for (int fm = 0; fm < FM; ++fm) acc[f][fm] *= osc;
    }
    YP_TL(40);
    // ---- epilogue: bias -> activation -> (+ residual) -> store, 16-byte vectors per pixel
    yp_pin_arrived(bias);
    auto epilogue = [&](auto res_c) {
    constexpr int RES = decltype(res_c)::value;
#pragma unroll
    for (int fm = 0; fm < FM; ++fm) {
        const int m = m0 + wm * TM + fm * 16 + p;
        if (m >= a.M) continue;
        if constexpr (!DETECT) {
            if (a.split_slabs != nullptr) {   // deterministic split-K: this k slice's own slab, summed in slice order by yp_sum_slabs
                float* slab = a.split_slabs + (size_t)blockIdx.y * a.split_stride;
#pragma unroll
                for (int j = 0; j < LPG; ++j)
                    if (nb + j < a.Cout) slab[(size_t)m * a.out_cs + a.out_co + nb + j] = acc[j >> 2][fm][j & 3];
            } else if (a.atomic_out) {   // split-K partial sums (wgrad): fp32 atomics into a zero-initialised buffer
#pragma unroll
                for (int j = 0; j < LPG; ++j)
                    if (nb + j < a.Cout) atomicAdd(reinterpret_cast<float*>(a.out) + (size_t)m * a.out_cs + a.out_co + nb + j, acc[j >> 2][fm][j & 3]);
            } else {
                yp_epilogue_pixel<DT, OUT_F32, LPG, RES>(a, m, nb, bias, [&](int cj) { return acc[cj >> 2][fm][cj & 3]; });
            }
        }
    }
    };
    YP_RES_DISPATCH(a, epilogue);
    if constexpr (STATS) {
        // BatchNorm statistics of the raw output straight from the accumulators (training forward of a 1x1 Conv: the separate reduction
        // pass re-read the tensor, 5.8 us per layer): per lane sum / sum of squares over its pixels, 16-lane butterfly over the pixel
        // lanes, LDS atomics across the waves of a 64-pixel row block, one partial row per row block: stats[(rb*2 + {0,1})*Cout + c].
        float sv[LPG], sq[LPG];
#pragma unroll
        for (int j = 0; j < LPG; ++j) sv[j] = sq[j] = 0.f;
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) {
            const int m = m0 + wm * TM + fm * 16 + p;
            if (m < a.M) {
#pragma unroll
                for (int j = 0; j < LPG; ++j) { const float v = acc[j >> 2][fm][j & 3]; sv[j] += v; sq[j] += v * v; }
            }
        }
        // reduce over the 16 pixel lanes by recursive halving: at offset o a lane keeps one half of its channels and hands the other half
        // to lane ^ o (LPG/2 + LPG/4 + ... exchanges instead of 4 * LPG); once a single channel is left the remaining offsets are plain
        // butterfly adds.  Lane p ends up with the total of channel `mych` of its group's LPG channels.
        int mych = 0;
        YpHalve<LPG, 1>::run(sv, sq, p, mych);
        constexpr int HALVES = BM / 64;
        constexpr int WPH = WAVES_M / HALVES > 0 ? WAVES_M / HALVES : 1;       // waves sharing a 64-pixel row block (per channel range)
        constexpr int OWNERS = LPG < 16 ? LPG : 16;           // lanes p < OWNERS of a 16-lane group each own one channel total
        float* red = reinterpret_cast<float*>(smem);           // [HALVES][WPH][2][BN] in the now idle pipeline LDS; plain stores and a
        __syncthreads();                                       // fixed summation order: the statistics are bit-reproducible
        const int half = (wm * TM) / 64, wih = wm % WPH;
        if (p < OWNERS) {
            const int cl = wn * TN + g * LPG + mych;
            red[((half * WPH + wih) * 2 + 0) * BN + cl] = sv[0];
            red[((half * WPH + wih) * 2 + 1) * BN + cl] = sq[0];
        }
        __syncthreads();
        for (int i = t; i < HALVES * 2 * BN; i += 256) {
            const int h = i / (2 * BN), w2 = (i / BN) & 1, cl = i % BN;
            const int c = n0 + cl;
            const int rb = m0 / 64 + h;
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < WPH; ++w) v += red[((h * WPH + w) * 2 + w2) * BN + cl];
#ifndef YP_PROBE_NOSTATSTORE
            if (c < a.Cout && rb * 64 < a.M) a.stats[((size_t)w2 * a.Cout + c) * a.stats_rows + rb] = v;
#else
            if (v == 1.2345e-30f) a.stats[0] = v;
#endif
        }
    }
    YP_TL(41);
    if constexpr (DETECT) {
        // The (pixel x channel) tile is staged through the now idle pipeline LDS so that each wave writes one
        // pixel's channels as a contiguous run of x_out / z (both are o-contiguous per (pixel, anchor)).
        constexpr int PITCH = BN + (BN < 128 ? 1 : 0);
        static_assert(BM * PITCH * 4 <= NS * STAGE, "detect staging tile must fit the pipeline LDS");
        float* tile = reinterpret_cast<float*>(smem);
        YpAnchors anchors;
        anchors.load(a);
        __syncthreads();                  // every wave has consumed the last k tile (its DMAs were drained above)
#pragma unroll
        for (int fm = 0; fm < FM; ++fm)
#pragma unroll
            for (int j = 0; j < LPG; ++j)
                tile[(wm * TM + fm * 16 + p) * PITCH + wn * TN + g * LPG + j] = acc[j >> 2][fm][j & 3] + bias[j];
        __syncthreads();
        const int nreal = a.det_na * a.det_no, no = a.det_no;
        constexpr int GPP = BN / 4;           // 4-channel groups per pixel
        for (int idx = t; idx < BM * GPP; idx += 256) {
            const int pix = idx / GPP, grp = idx - pix * GPP;
            const int m = m0 + pix;
            if (m >= a.M) break;
            const int n = n0 + grp * 4;
            if (n >= nreal) continue;
            const int b = m / a.HoWo, rem = m - b * a.HoWo;
            const int y = rem / a.Wo, x = rem - y * a.Wo;
            const float* src = tile + pix * PITCH + grp * 4;
            const int an = (n * a.det_invno) >> 16, o = n - an * no;
            if (o + 3 < no) {                 // the 4 channels stay inside one anchor: two 16-byte stores
                const float v0 = src[0], v1 = src[1], v2 = src[2], v3 = src[3];
                const size_t cell = (size_t)(b * a.det_na + an) * a.HoWo + rem;
                float* xp = a.det_x + cell * no + o;
                // ONE 16-byte store per lane (rows of no = 85 floats are only 4-byte aligned: the unaligned dwordx4 form; four dword stores
                // per lane wrote every cache line of the run four times)
                reinterpret_cast<YpF4U*>(xp)->v = f32x4{v0, v1, v2, v3};
                if (a.det_z != nullptr) {
                    float zz[4];
                    const float vv[4] = {v0, v1, v2, v3};
                    float aw, ah;
                    anchors.pick(an, aw, ah);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float sg = yp_sigmoid(vv[j]);
                        const int oj = o + j;
                        float z;
                        if (oj == 0) z = (sg * 2.0f - 0.5f + (float)x) * a.det_stride;
                        else if (oj == 1) z = (sg * 2.0f - 0.5f + (float)y) * a.det_stride;
                        else if (oj == 2) { const float t2 = sg * 2.0f; z = t2 * t2 * aw; }
                        else if (oj == 3) { const float t2 = sg * 2.0f; z = t2 * t2 * ah; }
                        else z = sg;
                        zz[j] = z;
                    }
                    const size_t row = (size_t)a.det_row_off + (size_t)an * a.HoWo + rem;
                    float* zp = a.det_z + ((size_t)b * a.det_rows_total + row) * no + o;
                    reinterpret_cast<YpF4U*>(zp)->v = f32x4{zz[0], zz[1], zz[2], zz[3]};
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (n + j < nreal) yp_detect_store(a, anchors, b, rem, y, x, n + j, src[j]);
            }
        }
    }
}


// ==========================================================================================
// 3x3 convolution (stride 1 or 2, pad 1) with LDS halo reuse.
//
// A workgroup owns an 8 x 16 tile of output pixels x BN output channels.  Per 32-channel chunk of
// the input, the (8*S+2) x (16*S+2) input halo of that tile is DMA'd into LDS ONCE and serves all
// nine filter taps (the generic kernel re-fetches every input pixel once per tap); the filter is
// streamed one filter ROW (3 taps x BN rows x 32 channels) per pipeline step through a 3-stage ring.
// One raw barrier per step = 24..48 MFMAs per wave between barriers; DMA issue costs no VALU work
// (per-lane byte offsets are precomputed once, the channel / tap offset rides in the SGPR base).
//
// A 16-pixel MFMA fragment is 16 consecutive x positions of one output row, so for tap (r, s) its
// LDS rows are consecutive: stride 1: (y+r)*18 + (x+s); stride 2: the halo row stores even input
// columns at [0,17) and odd ones at [17,33) (pitch 34), so (2y+r)*34 + {0,17,1}[s] + x.
// Same source-side XOR swizzle as the generic kernel (keyed on the LDS row index).
// ==========================================================================================
// halo of a TH x 16 output tile: TH*S + (3 - S) rows; pitch 18 pixels (stride 1) / 2 x 17 de-interleaved columns (stride 2)
template <int STRIDE, int TH> struct Halo { static constexpr int HH = TH * STRIDE + (3 - STRIDE), HP = STRIDE == 1 ? 18 : 34; };

template <int DT, bool OUT_F32, int STRIDE, int BN, int WAVES_M, int TH = 8, bool STATS = false>
__global__ __launch_bounds__(256) void conv3x3_halo_kernel(const ConvKArgs a) {
    using E = Elem<DT>;
    using frag_t = typename E::frag;
    static_assert(E::BYTES <= 2, "halo kernel: 16-bit or 8-bit element types");
    constexpr int EB = E::BYTES, BK = 64 / EB;          // 64 bytes of k per chunk: 32 16-bit channels, 64 8-bit ones (Elem<YP_FP8>)
    constexpr int TW = 16;
    constexpr int WAVES_N = 4 / WAVES_M;
    constexpr int FM = TH / WAVES_M;                 // output rows (16-pixel fragments) per wave
    constexpr int TN = BN / WAVES_N, FN = TN / 16, LPG = 4 * FN;
    constexpr int HH = Halo<STRIDE, TH>::HH, HP = Halo<STRIDE, TH>::HP;
    constexpr int HROWS = HH * HP, HSLOTS = (HROWS + 15) / 16, NH = (HSLOTS + 3) / 4, HBYTES = HSLOTS * 1024;
    constexpr int WSLOTS_TAP = BN / 16, WSLOTS = 3 * WSLOTS_TAP, NW = (WSLOTS + 3) / 4, WBYTES = WSLOTS * 1024;
    static_assert(FN >= 1 && FM >= 1 && NW + NH <= 60, "unsupported tile");

    extern __shared__ __attribute__((aligned(1024))) char hsm[];      // [halo 0][halo 1][filter row 0][1][2]
    const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) char*)hsm);
    const int hbufs = a.Cin > BK ? 2 : 1;            // a single 32-channel chunk needs no second halo buffer (host sizes the LDS alike)
    const unsigned ldsW = lds0 + hbufs * HBYTES;

    // ---- tile decode (channel tiles fastest: neighbours share the input halo in L2)
    YP_TL_DECL;
    YP_TL(0);
    int bid = yp_xcd_remap(blockIdx.x, gridDim.x);
    const int tn = bid % a.tiles_n; bid /= a.tiles_n;
    const int tx = bid % a.tiles_x; bid /= a.tiles_x;
    const int ty = bid % a.tiles_y;
    const int b = bid / a.tiles_y;
    const int y0 = ty * TH, x0 = tx * TW, n0 = tn * BN;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int lrow = lane >> 2;
    const int jl = (lane & 3) ^ ((0x3300 >> ((lane >> 4) * 4)) & 3);

    YP_PIN2(const char*, in0); YP_PIN2(const char*, wgt);
    YP_PIN2(int, in0_cs); YP_PIN2(int, in0_co); YP_PIN2(int, Cin); YP_PIN2(int, Hi); YP_PIN2(int, Wi);
    YP_PIN2(unsigned, in0_zoff);

    // ---- per-lane DMA byte offsets, computed once
    unsigned hoff[NH];
    int hslot[NH];
#pragma unroll
    for (int i = 0; i < NH; ++i) {
        int sl = wave + 4 * i;
        if (sl > HSLOTS - 1) sl = HSLOTS - 1;
        hslot[i] = sl;
        const int rho = sl * 16 + lrow;
        const int hy = rho / HP, rem = rho - hy * HP;
        int hx;
        bool valid = rho < HROWS;
        if constexpr (STRIDE == 1) hx = rem;
        else { hx = rem < 17 ? 2 * rem : 2 * (rem - 17) + 1; valid = valid && rem < 33; }
        const int iy = y0 * STRIDE - 1 + hy, ix = x0 * STRIDE - 1 + hx;
        valid = valid && (unsigned)iy < (unsigned)Hi && (unsigned)ix < (unsigned)Wi;
        hoff[i] = valid ? (unsigned)(((b * Hi + iy) * Wi + ix) * in0_cs * EB) + (unsigned)jl * 16u : in0_zoff;
        // chunk 0's DMA leaves as soon as its offset exists: the rest of the setup overlaps the first round trip
        yp_glds16_s(in0 + (size_t)in0_co * EB, hoff[i], lds0 + hslot[i] * 1024);
    }
    unsigned woff[NW];
    int wslot[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        int sl = wave + 4 * i;
        if (sl > WSLOTS - 1) sl = WSLOTS - 1;
        wslot[i] = sl;
        const int tap_s = sl / WSLOTS_TAP, rs = sl - tap_s * WSLOTS_TAP;
        const int rho = rs * 16 + lrow;
        const int wn_ = rho / TN, q = rho % TN;
        const int f = q >> 4, g_ = (q & 15) >> 2, r_ = q & 3;
        const int n = n0 + wn_ * TN + g_ * LPG + f * 4 + r_;
        woff[i] = ((n < a.Npad) ? (unsigned)n * (unsigned)a.Kpad * EB : a.wgt_zrow) + (unsigned)jl * 16u + (unsigned)(tap_s * Cin * EB);
    }
    auto issueW = [&](int c, int r) {      // filter row r of chunk c -> ring stage r
        const char* wk = wgt + ((size_t)(r * 3) * Cin + (size_t)c * BK) * EB;
#pragma unroll
        for (int i = 0; i < NW; ++i) yp_glds16_s(wk, woff[i], ldsW + r * WBYTES + wslot[i] * 1024);
    };
    auto issueH = [&](int c) {
        const char* hk = in0 + (size_t)(in0_co + c * BK) * EB;
#pragma unroll
        for (int i = 0; i < NH; ++i) yp_glds16_s(hk, hoff[i], lds0 + (c & 1) * HBYTES + hslot[i] * 1024);
    };

    const int wm = wave % WAVES_M, wn = wave / WAVES_M;
    const int p = lane & 15, g = lane >> 4;
    const int swr = (0x3300 >> ((p >> 2) * 4)) & 3;
    const int w_rd = (wn * TN + p) * 64 + ((g ^ swr) << 4);          // + (s*BN + f*16)*64 per fragment
    int x_row[FM];                                                     // halo row index of (local row, x = p) for r = s = 0
#pragma unroll
    for (int fm = 0; fm < FM; ++fm) x_row[fm] = ((wm * FM + fm) * STRIDE) * HP + p;

    f32x4 acc[FN][FM];
#pragma unroll
    for (int f = 0; f < FN; ++f)
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) acc[f][fm] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nb = n0 + wn * TN + g * LPG;
    float bias[LPG];
    yp_load_bias<LPG>(a, nb, bias);      // older than every DMA: the counted waits below stay valid

    const int nchunks = Cin / BK;
    const int nsteps = 3 * nchunks;
    YP_TL(1);
    issueW(0, 0);        // (halo chunk 0 is already in flight; filter row 1 stays the youngest DMA, as the counted waits assume)
    issueW(0, 1);
    for (int c = 0; c < nchunks; ++c) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int st = 3 * c + r;
            // DMAs younger than filter row `st`: the next filter row, and (r != 0) the next chunk's halo
            const bool moreW = st + 1 < nsteps;
            const bool moreH = (r != 0) && (c + 1 < nchunks);
            if (moreW && moreH) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NW + NH) : "memory");
            else if (moreH) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NH) : "memory");
            else if (moreW) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (WAR: the reads of the stage / halo buffer refilled behind this barrier have retired)
            __builtin_amdgcn_s_barrier();
            if (st < 30) YP_TL(2 + st);
            if (st + 2 < nsteps) { const int s2 = st + 2; issueW(s2 / 3, s2 % 3); }
            if (r == 0 && c + 1 < nchunks) issueH(c + 1);
            const char* hb = hsm + (c & 1) * HBYTES;
            const char* wb = hsm + hbufs * HBYTES + r * WBYTES;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                constexpr int XC1[3] = {0, 1, 2};
                constexpr int XC2[3] = {0, 17, 1};
                const int xc = (STRIDE == 1 ? XC1[s] : XC2[s]) + r * HP;
                frag_t wf[FN], xf[FM];
#pragma unroll
                for (int f = 0; f < FN; ++f) wf[f] = *reinterpret_cast<const frag_t*>(wb + w_rd + (s * BN + f * 16) * 64);
#pragma unroll
                for (int fm = 0; fm < FM; ++fm) {
                    const int rho = x_row[fm] + xc;
                    const int sw = (0x3300 >> (((rho >> 2) & 3) * 4)) & 3;
                    xf[fm] = *reinterpret_cast<const frag_t*>(hb + rho * 64 + ((g ^ sw) << 4));
                }
#pragma unroll
                for (int f = 0; f < FN; ++f)
#pragma unroll
                    for (int fm = 0; fm < FM; ++fm) acc[f][fm] = E::mma(wf[f], xf[fm], acc[f][fm]);
            }
        }
    }

    YP_TL(40);
    if constexpr (E::BYTES == 1) {          // 8-bit inputs: back to real units (dequantisation scales of activation and filter)
        const float osc = a.scale_in[0] * a.scale_w[0];
#pragma unroll
        for (int f = 0; f < FN; ++f)
#pragma unroll
            for (int fm = 0; fm < FM; ++fm) acc[f][fm] *= osc;
    }
    yp_pin_arrived(bias);
    auto epilogue = [&](auto res_c) {
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) {
            const int oy = y0 + wm * FM + fm, ox = x0 + p;
            if (oy >= a.Ho || ox >= a.Wo) continue;
            const int m = (b * a.Ho + oy) * a.Wo + ox;
            yp_epilogue_pixel<DT, OUT_F32, LPG, decltype(res_c)::value>(a, m, nb, bias, [&](int cj) { return acc[cj >> 2][fm][cj & 3]; });
        }
    };
    YP_RES_DISPATCH(a, epilogue);
    if constexpr (STATS) {
        // BatchNorm statistics of the raw output from the accumulators (training forward of a 3x3 Conv; see the generic kernel): one
        // partial row per pixel tile, stats[(tile*2 + {0,1})*Cout + c] -- the reduction pass that re-read the tensor disappears.
        float sv[LPG], sq[LPG];
#pragma unroll
        for (int j = 0; j < LPG; ++j) sv[j] = sq[j] = 0.f;
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) {
            const int oy = y0 + wm * FM + fm, ox = x0 + p;
            if (oy < a.Ho && ox < a.Wo) {
#pragma unroll
                for (int j = 0; j < LPG; ++j) { const float v = acc[j >> 2][fm][j & 3]; sv[j] += v; sq[j] += v * v; }
            }
        }
        int mych = 0;
        YpHalve<LPG, 1>::run(sv, sq, p, mych);
        constexpr int OWNERS = LPG < 16 ? LPG : 16;
        float* red = reinterpret_cast<float*>(hsm);            // [WAVES_M][2][BN] in the idle pipeline LDS; fixed summation order
        __syncthreads();
        if (p < OWNERS) {
            const int cl = wn * TN + g * LPG + mych;
            red[(wm * 2 + 0) * BN + cl] = sv[0];
            red[(wm * 2 + 1) * BN + cl] = sq[0];
        }
        __syncthreads();
        const int tile = (b * a.tiles_y + ty) * a.tiles_x + tx;
        for (int i = t; i < 2 * BN; i += 256) {
            const int w2 = i / BN, cl = i % BN, c = n0 + cl;
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < WAVES_M; ++w) v += red[(w * 2 + w2) * BN + cl];
#ifndef YP_PROBE_NOSTATSTORE
            if (c < a.Cout) a.stats[((size_t)w2 * a.Cout + c) * a.stats_rows + tile] = v;
#else
            if (v == 1.2345e-30f) a.stats[0] = v;
#endif
        }
    }
    YP_TL(41);
}


// ==========================================================================================
// Fused Bottleneck (reference models/common.py:79-89): out = [x +] act2(conv3x3(act1(conv1x1(x)))) with
// C = Cin = hidden = Cout in {32, 64, 128}.  Same 8 x 16 output tile as the halo kernel; the hidden tensor
// never leaves the workgroup:
//   phase A  the 10 x 18 halo of x (all C channels, NCH = C/32 chunks) and the 1x1 filter are DMA'd into LDS
//            in ONE burst (no stage reuse -> a single counted wait + barrier per chunk); the four waves
//            compute hidden[192 halo rows][C] (3 pixel fragments x C/16 channel fragments per wave), apply
//            bias + activation, zero the rows that fall outside the image (the 3x3's padding acts on the
//            HIDDEN tensor) and store it as 16-bit rows in exactly the swizzled 64-byte-row format the halo
//            DMA would have produced;
//   phase B  the halo kernel's tap loop, reading the halo from that resident LDS copy (no halo DMA at all);
//            filter rows stream through the 3-stage ring, which re-uses the LDS of phase A's operands.
// HBM traffic: x once (+ the shortcut re-read of the tile centre, an L2 hit) and out once -- the hidden
// tensor's write + 9-tap read of the two-launch form disappear, as does one launch.
// ==========================================================================================
// NWAVES = 8 (two wavefronts per SIMD, 512 threads): the phases of a tile are latency chains of ONE wave per SIMD (measured at C = 128, 40 x 40
// x 8: 27.9 k clocks per tile -- setup 2.2 k, phase A DMA + MFMAs 4.8 k, bias + SiLU + hidden write 5.6 k, twelve filter-row steps at ~0.9 k
// each with 0.38 k of MFMA work per wave, epilogue 3.7 k -- and a launch is ONE round of 240 workgroups); with eight waves every phase has half
// the work per wave and a second wave per SIMD to issue while the first waits for LDS.  Same fragments, same k order: bit-identical.
template <int DT, int C, int BN, int WAVES_M, bool POST = false, int NWAVES = 4>
__global__ __launch_bounds__(64 * NWAVES) void bottleneck_halo_kernel(const ConvKArgs a) {
    using E = Elem<DT>;
    using frag_t = typename E::frag;
    using sc = typename E::scalar;
    static_assert(E::BYTES == 2, "fused bottleneck: 16-bit element types");
    constexpr int EB = 2, BK = 32;
    constexpr int TH = 8, TW = 16;
    constexpr int WAVES_N = NWAVES / WAVES_M;
    constexpr int FM = TH / WAVES_M;
    constexpr int TN = BN / WAVES_N, FN = TN / 16, LPG = 4 * FN;
    constexpr int HH = 10, HP = 18;
    constexpr int HROWS = HH * HP, HSLOTS = 12, NH = (HSLOTS + NWAVES - 1) / NWAVES, HBYTES = HSLOTS * 1024;
    constexpr int NCH = C / BK;                         // 32-channel chunks of x / of the hidden tensor
    constexpr int NSUB = C / 32;                        // 32-channel sub-tiles of the hidden tensor (2 fragments each)
    constexpr int WASLOTS = C / 16, NWA = (WASLOTS + NWAVES - 1) / NWAVES, WABYTES = WASLOTS * 1024;
    constexpr int ASTAGE = HBYTES + WABYTES;
    constexpr int WSLOTS_TAP = BN / 16, WSLOTS = 3 * WSLOTS_TAP, NW = (WSLOTS + NWAVES - 1) / NWAVES, WBYTES = WSLOTS * 1024;
    constexpr int NLA = NH + NWA;                       // DMA instructions per wave per phase-A chunk
    static_assert(HROWS <= HSLOTS * 16 && BN <= C && FN >= 1 && FM >= 1, "unsupported tile");
    // POST: the C3 tail.  out = act(W3 . cat(bottleneck output, in1) + b3), N3 = K3 = 2C; the bottleneck output never leaves LDS.
    constexpr int N3 = 2 * C, FN3 = N3 / 16, LPG3 = 4 * FN3, KCH3 = N3 / BK;      // one wave column: every wave holds all N3 channels
    constexpr int W3BYTES = KCH3 * N3 * 64, NW3 = KCH3 * (N3 / 16) / 4, NU = NCH * 8 / 4;
    static_assert(!POST || (BN == C && WAVES_M == 4 && NWAVES == 4), "C3 tail: all channels of a pixel in one workgroup");
    constexpr int NPF = (HSLOTS + NWAVES - 1) / NWAVES;      // phase A: pixel fragments (16 halo rows each) per wave -- fragment wave + NWAVES * i

    extern __shared__ __attribute__((aligned(1024))) char hsm[];      // [hidden | bottleneck output][ring: phase A operands | filter rows | W3][POST: in1 tile]
    const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) char*)hsm);
    constexpr int RING = NCH * HBYTES;
    constexpr int OPA = NCH * ASTAGE, OPB = 3 * WBYTES;
    constexpr int RINGBYTES = POST ? (OPA > OPB ? (OPA > W3BYTES ? OPA : W3BYTES) : (OPB > W3BYTES ? OPB : W3BYTES)) : (OPA > OPB ? OPA : OPB);
    const unsigned ldsR = lds0 + RING;
    const unsigned ldsU = ldsR + RINGBYTES;

    YP_TL_DECL;
    YP_TL(0);
    int bid = yp_xcd_remap(blockIdx.x, gridDim.x);
    const int tn = bid % a.tiles_n; bid /= a.tiles_n;
    const int tx = bid % a.tiles_x; bid /= a.tiles_x;
    const int ty = bid % a.tiles_y;
    const int b = bid / a.tiles_y;
    const int y0 = ty * TH, x0 = tx * TW, n0 = tn * BN;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int lrow = lane >> 2;
    const int jl = (lane & 3) ^ ((0x3300 >> ((lane >> 4) * 4)) & 3);

    YP_PIN2(const char*, in0); YP_PIN2(const char*, wgt); YP_PIN2(const char*, pre_wgt);
    YP_PIN2(int, in0_cs); YP_PIN2(int, in0_co); YP_PIN2(int, Hi); YP_PIN2(int, Wi); YP_PIN2(int, pre_Kpad);
    YP_PIN2(unsigned, in0_zoff);

    // ---- per-lane DMA byte offsets
    unsigned hoff[NH];
    int hslot[NH];
#pragma unroll
    for (int i = 0; i < NH; ++i) {
        int sl = wave + NWAVES * i;
        if (sl > HSLOTS - 1) sl = HSLOTS - 1;           // (8 waves: surplus instructions re-fetch the last slot, same bytes)
        hslot[i] = sl;
        const int rho = sl * 16 + lrow;
        const int hy = rho / HP, hx = rho - hy * HP;
        const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
        const bool valid = rho < HROWS && (unsigned)iy < (unsigned)Hi && (unsigned)ix < (unsigned)Wi;
        hoff[i] = valid ? (unsigned)(((b * Hi + iy) * Wi + ix) * in0_cs * EB) + (unsigned)jl * 16u : in0_zoff;
    }
    unsigned waoff[NWA];
    int waslot[NWA];
#pragma unroll
    for (int i = 0; i < NWA; ++i) {
        int sl = wave + NWAVES * i;
        if (sl > WASLOTS - 1) sl = WASLOTS - 1;
        waslot[i] = sl;
        const int rho = sl * 16 + lrow;                 // LDS row -> hidden channel (lanes own 8 consecutive channels per sub-tile)
        const int sub = rho >> 5, q = rho & 31;
        const int n = sub * 32 + ((q & 15) >> 2) * 8 + (q >> 4) * 4 + (q & 3);
        waoff[i] = (unsigned)n * (unsigned)pre_Kpad * EB + (unsigned)jl * 16u;
    }
    unsigned woff[NW];
    int wslot[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        int sl = wave + NWAVES * i;
        if (sl > WSLOTS - 1) sl = WSLOTS - 1;
        wslot[i] = sl;
        const int tap_s = sl / WSLOTS_TAP, rs = sl - tap_s * WSLOTS_TAP;
        const int rho = rs * 16 + lrow;
        const int wn_ = rho / TN, q = rho % TN;
        const int f = q >> 4, g_ = (q & 15) >> 2, r_ = q & 3;
        const int n = n0 + wn_ * TN + g_ * LPG + f * 4 + r_;
        woff[i] = ((n < a.Npad) ? (unsigned)n * (unsigned)a.Kpad * EB : a.wgt_zrow) + (unsigned)jl * 16u + (unsigned)(tap_s * C * EB);
    }

    const int p = lane & 15, g = lane >> 4;
    const int swr = (0x3300 >> ((p >> 2) * 4)) & 3;

    // biases of both convolutions: ordinary loads issued before any DMA (the counted waits stay valid)
    float b1[NSUB][8];
#pragma unroll
    for (int sb = 0; sb < NSUB; ++sb) {
        f32x4 lo = f32x4{0.f, 0.f, 0.f, 0.f}, hi = lo;
        if (a.pre_bias != nullptr) {
            lo = *reinterpret_cast<const f32x4*>(a.pre_bias + sb * 32 + g * 8);
            hi = *reinterpret_cast<const f32x4*>(a.pre_bias + sb * 32 + g * 8 + 4);
        }
        b1[sb][0] = lo[0]; b1[sb][1] = lo[1]; b1[sb][2] = lo[2]; b1[sb][3] = lo[3];
        b1[sb][4] = hi[0]; b1[sb][5] = hi[1]; b1[sb][6] = hi[2]; b1[sb][7] = hi[3];
    }
    const int wm = wave % WAVES_M, wn = wave / WAVES_M;
    const int nb = n0 + wn * TN + g * LPG;
    float bias[LPG];
    yp_load_bias<LPG>(a, nb, bias);

    YP_TL(1);
    if constexpr (POST) {
        // in1 (the C3's cv2 branch) at the tile's 128 centre pixels: issued first, consumed last
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            const int sl = wave + 4 * i;                 // NCH chunks x 8 slots of 16 pixel rows
            const int cc = sl >> 3, rho = (sl & 7) * 16 + lrow;
            const int oy = y0 + (rho >> 4), ox = x0 + (rho & 15);
            const bool ok = oy < a.Ho && ox < a.Wo;
            const unsigned off = ok ? (unsigned)((((b * a.Ho + oy) * a.Wo + ox) * a.in1_cs + a.in1_co + cc * BK) * EB) + (unsigned)jl * 16u : a.in1_zoff;
            yp_glds16_s(a.in1, off, ldsU + sl * 1024);
        }
    }
    // ---- phase A: everything in flight at once
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const char* hk = in0 + (size_t)(in0_co + c * BK) * EB;
        const char* wk = pre_wgt + (size_t)(c * BK) * EB;
#pragma unroll
        for (int i = 0; i < NH; ++i) yp_glds16_s(hk, hoff[i], ldsR + c * ASTAGE + hslot[i] * 1024);
#pragma unroll
        for (int i = 0; i < NWA; ++i) yp_glds16_s(wk, waoff[i], ldsR + c * ASTAGE + HBYTES + waslot[i] * 1024);
    }
    f32x4 hacc[NSUB][2][NPF];
#pragma unroll
    for (int sb = 0; sb < NSUB; ++sb)
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int i = 0; i < NPF; ++i) hacc[sb][f][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int a_rd = p * 64 + ((g ^ swr) << 4);          // fragment row p, k group g (both operand images use the same swizzle key)
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        if (c == NCH - 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (c == NCH - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLA) : "memory");
        else if (c == NCH - 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NLA) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NLA) : "memory");
        __builtin_amdgcn_s_barrier();
        YP_TL(2 + c);
        const char* xb = hsm + RING + c * ASTAGE;
        const char* wb = xb + HBYTES;
        frag_t xf[NPF];
#pragma unroll
        for (int i = 0; i < NPF; ++i) {                  // (a wave without an i-th fragment re-reads the last one; its result is not stored)
            const int fr = wave + NWAVES * i < HSLOTS ? wave + NWAVES * i : HSLOTS - 1;
            xf[i] = *reinterpret_cast<const frag_t*>(xb + fr * 1024 + a_rd);
        }
#pragma unroll
        for (int sb = 0; sb < NSUB; ++sb) {
            frag_t wf[2];
#pragma unroll
            for (int f = 0; f < 2; ++f) wf[f] = *reinterpret_cast<const frag_t*>(wb + (sb * 32 + f * 16) * 64 + a_rd);
#pragma unroll
            for (int f = 0; f < 2; ++f)
#pragma unroll
                for (int i = 0; i < NPF; ++i)
                    if (NWAVES * i + NWAVES <= HSLOTS || wave + NWAVES * i < HSLOTS) hacc[sb][f][i] = E::mma(wf[f], xf[i], hacc[sb][f][i]);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();            // every wave is done with phase A's operands (its reads have retired): the ring may be overwritten
    YP_TL(8);
#pragma unroll
    for (int sb = 0; sb < NSUB; ++sb)        // make the compiler's own wait for the bias loads land here, not behind the prefetch below
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(b1[sb][j]));

    auto issueW = [&](int c, int r) {        // filter row r of chunk c -> ring stage r
        const char* wk = wgt + ((size_t)(r * 3) * C + (size_t)c * BK) * EB;
#pragma unroll
        for (int i = 0; i < NW; ++i) yp_glds16_s(wk, woff[i], ldsR + r * WBYTES + wslot[i] * 1024);
    };
    // (Measured and dropped, round 5: a deeper filter ring -- up to six stages fit the ring region at C = 128 -- changes nothing (tile 11:
    // 14.4 vs 14.4 us, tile 18: 12.6 vs 12.6): the tap loop is bound by its LDS fragment reads (18 ds_read_b128 per 24 MFMAs per wave and
    // step; a wave alone on its SIMD issues one every ~29 clocks), not by the filter DMA.  What helps is a second wave per SIMD: NWAVES = 8.)
    issueW(0, 0);
    issueW(0, 1);

    // ---- hidden = act(hacc + b1), zero outside the image, 16-bit, into the resident halo image
#pragma unroll
    for (int i = 0; i < NPF; ++i) {
        if (wave + NWAVES * i >= HSLOTS) break;          // (wave-uniform)
        const int rho = (wave + NWAVES * i) * 16 + p;
        const int hy = rho / HP, hx = rho - hy * HP;
        const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
        const bool inside = (unsigned)iy < (unsigned)Hi && (unsigned)ix < (unsigned)Wi;      // rows >= HROWS are never read
#pragma unroll
        for (int sb = 0; sb < NSUB; ++sb) {
            u32x4 pk;
            sc* e = reinterpret_cast<sc*>(&pk);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float v = hacc[sb][j >> 2][i][j & 3] + b1[sb][j];
                if (a.pre_act == YP_ACT_SILU) v = yp_silu(v);
                e[j] = (sc)(inside ? v : 0.0f);
            }
            *reinterpret_cast<u32x4*>(hsm + sb * HBYTES + rho * 64 + ((g ^ swr) << 4)) = pk;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    YP_TL(9);

    // ---- phase B: 3x3 over the resident hidden halo
    const int w_rd = (wn * TN + p) * 64 + ((g ^ swr) << 4);
    int x_row[FM];
#pragma unroll
    for (int fm = 0; fm < FM; ++fm) x_row[fm] = (wm * FM + fm) * HP + p;
    f32x4 acc[FN][FM];
#pragma unroll
    for (int f = 0; f < FN; ++f)
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) acc[f][fm] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int nsteps = 3 * NCH;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int st = 3 * c + r;
            // WAR on the ring: the refill issued behind this barrier overwrites the stage that step st-1 READ.  The loop is fully unrolled
            // and an MFMA is not a memory operation, so the compiler may sink step st-1's last MFMAs -- and the lgkmcnt wait for their
            // operands -- below the barrier: a ds_read could then still be in flight when another wave's DMA lands in its stage
            // (rare wrong tiles at high occupancy: YOLOPoint-s bs 8 640x640 differed run to run).  Retire the reads before the barrier.
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (st + 1 < nsteps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            YP_TL(10 + st);
            if (st + 2 < nsteps) { const int s2 = st + 2; issueW(s2 / 3, s2 % 3); }
            const char* hb = hsm + c * HBYTES;
            const char* wb = hsm + RING + r * WBYTES;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int xc = s + r * HP;
                frag_t wf[FN], xf[FM];
#pragma unroll
                for (int f = 0; f < FN; ++f) wf[f] = *reinterpret_cast<const frag_t*>(wb + w_rd + (s * BN + f * 16) * 64);
#pragma unroll
                for (int fm = 0; fm < FM; ++fm) {
                    const int rho = x_row[fm] + xc;
                    const int sw = (0x3300 >> (((rho >> 2) & 3) * 4)) & 3;
                    xf[fm] = *reinterpret_cast<const frag_t*>(hb + rho * 64 + ((g ^ sw) << 4));
                }
#pragma unroll
                for (int f = 0; f < FN; ++f)
#pragma unroll
                    for (int fm = 0; fm < FM; ++fm) acc[f][fm] = E::mma(wf[f], xf[fm], acc[f][fm]);
            }
        }
    }

    YP_TL(40);
    if constexpr (!POST) {
        yp_pin_arrived(bias);
        auto epilogue = [&](auto res_c) {
#pragma unroll
            for (int fm = 0; fm < FM; ++fm) {
                const int oy = y0 + wm * FM + fm, ox = x0 + p;
                if (oy >= a.Ho || ox >= a.Wo) continue;
                const int m = (b * a.Ho + oy) * a.Wo + ox;
                yp_epilogue_pixel<DT, false, LPG, decltype(res_c)::value>(a, m, nb, bias, [&](int cj) { return acc[cj >> 2][fm][cj & 3]; });
            }
        };
        YP_RES_DISPATCH(a, epilogue);
        YP_TL(41);
    } else {
        // ---- C3 tail.  The ring is free: fetch W3 (all of it), meanwhile finish the bottleneck output into LDS (it replaces the
        // hidden tensor, same swizzled 64-byte-row format, rows = the tile's 128 centre pixels).
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        {
            const char* w3 = a.post_wgt;
#pragma unroll
            for (int i = 0; i < NW3; ++i) {
                const int sl = wave + 4 * i;             // KCH3 chunks x N3/16 slots
                const int kc = sl / (N3 / 16), rs = sl - kc * (N3 / 16);
                const int rho = rs * 16 + lrow;          // LDS row -> channel (lanes own LPG3 consecutive channels)
                const int n = ((rho & 15) >> 2) * LPG3 + (rho >> 4) * 4 + (rho & 3);
                const unsigned off = (unsigned)n * (unsigned)a.post_Kpad * EB + (unsigned)(kc * BK * EB) + (unsigned)jl * 16u;
                yp_glds16_s(w3, off, ldsR + sl * 1024);
            }
        }
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) {
            const int ly = wm * FM + fm;
            const int oy = y0 + ly, ox = x0 + p;
            const bool inside = oy < a.Ho && ox < a.Wo;
            const int m = (b * a.Ho + oy) * a.Wo + ox;
            const int rho = ly * 16 + p;
#pragma unroll
            for (int h = 0; h < LPG / 8; ++h) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float x = acc[(h * 8 + j) >> 2][fm][(h * 8 + j) & 3] + bias[h * 8 + j];
                    if (a.act == YP_ACT_SILU) x = yp_silu(x);
                    v[j] = x;
                }
                const int nc = nb + h * 8;
                if (a.has_res && inside) {
                    const u32x4 raw = *reinterpret_cast<const u32x4*>(a.res + ((size_t)m * a.res_cs + a.res_co + nc) * EB);
                    const sc* e = reinterpret_cast<const sc*>(&raw);
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] += (float)e[j];
                }
                u32x4 pk;
                sc* e = reinterpret_cast<sc*>(&pk);
#pragma unroll
                for (int j = 0; j < 8; ++j) e[j] = (sc)v[j];
                *reinterpret_cast<u32x4*>(hsm + (nc >> 5) * 8192 + rho * 64 + ((((nc & 31) >> 3) ^ swr) << 4)) = pk;
            }
        }
        float b3[LPG3];
#pragma unroll
        for (int q = 0; q < LPG3 / 4; ++q) {
            f32x4 b4 = f32x4{0.f, 0.f, 0.f, 0.f};
            if (a.post_bias != nullptr) b4 = *reinterpret_cast<const f32x4*>(a.post_bias + g * LPG3 + 4 * q);
            b3[4 * q] = b4[0]; b3[4 * q + 1] = b4[1]; b3[4 * q + 2] = b4[2]; b3[4 * q + 3] = b4[3];
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        yp_pin_arrived(b3);
        __builtin_amdgcn_s_barrier();
        YP_TL(41);
        f32x4 acc3[FN3][FM];
#pragma unroll
        for (int f = 0; f < FN3; ++f)
#pragma unroll
            for (int fm = 0; fm < FM; ++fm) acc3[f][fm] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int rd = p * 64 + ((g ^ swr) << 4);
#pragma unroll
        for (int kc = 0; kc < KCH3; ++kc) {
            const char* xb = kc < NCH ? hsm + kc * 8192 : hsm + RING + RINGBYTES + (kc - NCH) * 8192;
            const char* wb = hsm + RING + kc * (N3 * 64);
            frag_t xf[FM];
#pragma unroll
            for (int fm = 0; fm < FM; ++fm) xf[fm] = *reinterpret_cast<const frag_t*>(xb + (wm * FM + fm) * 1024 + rd);
#pragma unroll
            for (int f = 0; f < FN3; ++f) {
                const frag_t wf = *reinterpret_cast<const frag_t*>(wb + f * 1024 + rd);
#pragma unroll
                for (int fm = 0; fm < FM; ++fm) acc3[f][fm] = E::mma(wf, xf[fm], acc3[f][fm]);
            }
        }
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) {
            const int oy = y0 + wm * FM + fm, ox = x0 + p;
            if (oy >= a.Ho || ox >= a.Wo) continue;
            const size_t m = (size_t)(b * a.Ho + oy) * a.Wo + ox;
            char* op = a.out + (m * a.out_cs + a.out_co + g * LPG3) * EB;
#pragma unroll
            for (int h = 0; h < LPG3 / 8; ++h) {
                if (g * LPG3 + h * 8 >= a.post_N) continue;
                u32x4 pk;
                sc* e = reinterpret_cast<sc*>(&pk);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float x = acc3[(h * 8 + j) >> 2][fm][(h * 8 + j) & 3] + b3[h * 8 + j];
                    if (a.post_act == YP_ACT_SILU) x = yp_silu(x);
                    e[j] = (sc)x;
                }
                *reinterpret_cast<u32x4*>(op + h * 16) = pk;
            }
        }
        YP_TL(42);
    }
}


// ==========================================================================================
// Persistent form of the fused Bottleneck + C3 tail for C = 32 (Bottleneck1.m.0 > cv3 of YOLOPoint-s: M = 204 800 pixels at batch 8, 1 600
// tiles of 8 x 16 pixels).  bottleneck_halo_kernel<DT, 32, 32, 4, true> ran one tile per workgroup: every tile fetched the three filters again
// (28 KB through L2 -> LDS against 36 KB of activations in + out) and stood at seven barriers, each behind a DMA wait -- ~12 us per tile,
// four resident workgroups per CU, 26 us per launch for 52 MB (1.7 TB/s).  Here a workgroup keeps W1 (2 KB), W2 (18 KB) and W3 (8 KB) in LDS
// for its whole life, walks its tiles, and the halo of x and the cv2-branch tile of the NEXT tile are in flight (LDS-DMA into the other
// half of a double buffer) while the current one is multiplied: one counted wait and four barriers per tile, none of them behind a fresh
// DMA.  The shortcut (res = x) is read from the resident halo instead of from global memory (an ordinary load would make the compiler's
// vmcnt wait drain the prefetch).  Same fragments, same MFMA order, same roundings as the one-tile kernel: bit-identical
// (tests/test_gpu_blocks.py::test_persistent_bottleneck_c32_equals_the_one_tile_kernel).
// LDS: hidden | B-out 12 KB, x halo 2 x 12 KB, in1 tile 2 x 8 KB, W1 2 KB, W2 18 KB, W3 8 KB = 80 KB -> two workgroups per CU.
// ==========================================================================================
template <int DT>
__global__ __launch_bounds__(256, 2) void bneck32_persist_kernel(const ConvKArgs a) {
    using E = Elem<DT>;
    using frag_t = typename E::frag;
    using sc = typename E::scalar;
    static_assert(E::BYTES == 2, "fused bottleneck: 16-bit element types");
    constexpr int EB = 2, BK = 32, C = 32, BN = 32;
    constexpr int TH = 8, TW = 16, FM = 2, FN = 2, LPG = 8;
    constexpr int HH = 10, HP = 18, HROWS = HH * HP, NH = 3, HBYTES = 12 * 1024;
    constexpr int WBYTES = 6 * 1024;                     // one filter row of the 3x3: 3 taps x 2 slots of 16 output channels
    constexpr int N3 = 64, FN3 = 4, LPG3 = 16, KCH3 = 2;
    constexpr int UBYTES = 8 * 1024;
    constexpr int OFF_HID = 0, OFF_X = HBYTES, OFF_U = OFF_X + 2 * HBYTES, OFF_WA = OFF_U + 2 * UBYTES, OFF_WB = OFF_WA + 2048, OFF_W3 = OFF_WB + 3 * WBYTES;
    static_assert(OFF_W3 + 8192 == 80 * 1024, "LDS layout");

    extern __shared__ __attribute__((aligned(1024))) char hsm[];
    const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) char*)hsm);

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int lrow = lane >> 2;
    const int jl = (lane & 3) ^ ((0x3300 >> ((lane >> 4) * 4)) & 3);
    const int p = lane & 15, g = lane >> 4;
    const int swr = (0x3300 >> ((p >> 2) * 4)) & 3;

    YP_PIN2(const char*, in0); YP_PIN2(const char*, in1);
    YP_PIN2(int, in0_cs); YP_PIN2(int, in0_co); YP_PIN2(int, in1_cs); YP_PIN2(int, in1_co); YP_PIN2(int, Hi); YP_PIN2(int, Wi);
    YP_PIN2(unsigned, in0_zoff); YP_PIN2(unsigned, in1_zoff);

    // ---- the three filters, once per workgroup
    {
        int sl = wave < 2 ? wave : 1;                    // W1: 2 slots (rows = hidden channels in the lane-owns-8-consecutive order)
        int rho = sl * 16 + lrow;
        int q = rho & 31;
        int n = ((q & 15) >> 2) * 8 + (q >> 4) * 4 + (q & 3);
        yp_glds16_s(a.pre_wgt, (unsigned)n * (unsigned)a.pre_Kpad * EB + (unsigned)jl * 16u, lds0 + OFF_WA + sl * 1024);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const char* wk = a.wgt + (size_t)(r * 3) * C * EB;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                sl = wave + 4 * i;
                if (sl > 5) sl = 5;
                const int tap_s = sl >> 1, rs = sl & 1;
                rho = rs * 16 + lrow;
                const int f = rho >> 4, g_ = (rho & 15) >> 2, r_ = rho & 3;
                n = g_ * LPG + f * 4 + r_;
                const unsigned off = ((n < a.Npad) ? (unsigned)n * (unsigned)a.Kpad * EB : a.wgt_zrow) + (unsigned)jl * 16u + (unsigned)(tap_s * C * EB);
                yp_glds16_s(wk, off, lds0 + OFF_WB + r * WBYTES + sl * 1024);
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            sl = wave + 4 * i;                           // W3: KCH3 chunks x 4 slots
            const int kc = sl >> 2, rs = sl & 3;
            rho = rs * 16 + lrow;
            n = ((rho & 15) >> 2) * LPG3 + (rho >> 4) * 4 + (rho & 3);
            yp_glds16_s(a.post_wgt, (unsigned)n * (unsigned)a.post_Kpad * EB + (unsigned)(kc * BK * EB) + (unsigned)jl * 16u, lds0 + OFF_W3 + sl * 1024);
        }
    }
    // ---- biases (ordinary loads, once)
    float b1[8];
    {
        f32x4 lo = f32x4{0.f, 0.f, 0.f, 0.f}, hi = lo;
        if (a.pre_bias != nullptr) { lo = *reinterpret_cast<const f32x4*>(a.pre_bias + g * 8); hi = *reinterpret_cast<const f32x4*>(a.pre_bias + g * 8 + 4); }
        b1[0] = lo[0]; b1[1] = lo[1]; b1[2] = lo[2]; b1[3] = lo[3]; b1[4] = hi[0]; b1[5] = hi[1]; b1[6] = hi[2]; b1[7] = hi[3];
    }
    const int wm = wave;                                 // WAVES_M = 4: wave w owns tile rows 2 w, 2 w + 1
    const int nb = g * LPG;
    float bias[LPG];
    yp_load_bias<LPG>(a, nb, bias);
    float b3[LPG3];
#pragma unroll
    for (int q = 0; q < LPG3 / 4; ++q) {
        f32x4 b4 = f32x4{0.f, 0.f, 0.f, 0.f};
        if (a.post_bias != nullptr) b4 = *reinterpret_cast<const f32x4*>(a.post_bias + g * LPG3 + 4 * q);
        b3[4 * q] = b4[0]; b3[4 * q + 1] = b4[1]; b3[4 * q + 2] = b4[2]; b3[4 * q + 3] = b4[3];
    }

    const int ntiles = a.stats_rows;                     // (host: B * tiles_y * tiles_x)
    auto decode = [&](int tile, int& b, int& y0, int& x0) {
        int bid = tile;
        const int tx = bid % a.tiles_x; bid /= a.tiles_x;
        const int ty = bid % a.tiles_y;
        b = bid / a.tiles_y; y0 = ty * TH; x0 = tx * TW;
    };
    auto issue_inputs = [&](int tile, int buf) {
        int b, y0, x0;
        decode(tile, b, y0, x0);
        const char* hk = in0 + (size_t)in0_co * EB;
#pragma unroll
        for (int i = 0; i < NH; ++i) {
            const int rho = (wave + 4 * i) * 16 + lrow;
            const int hy = rho / HP, hx = rho - hy * HP;
            const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
            const bool valid = rho < HROWS && (unsigned)iy < (unsigned)Hi && (unsigned)ix < (unsigned)Wi;
            const unsigned off = valid ? (unsigned)(((b * Hi + iy) * Wi + ix) * in0_cs * EB) + (unsigned)jl * 16u : in0_zoff;
            yp_glds16_s(hk, off, lds0 + OFF_X + buf * HBYTES + (wave + 4 * i) * 1024);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int sl = wave + 4 * i;                 // 8 slots of 16 pixel rows
            const int rho = sl * 16 + lrow;
            const int oy = y0 + (rho >> 4), ox = x0 + (rho & 15);
            const bool ok = oy < a.Ho && ox < a.Wo;
            const unsigned off = ok ? (unsigned)((((b * a.Ho + oy) * a.Wo + ox) * in1_cs + in1_co) * EB) + (unsigned)jl * 16u : in1_zoff;
            yp_glds16_s(in1, off, lds0 + OFF_U + buf * UBYTES + sl * 1024);
        }
    };

    int tile, tile_end, tile_step;
    yp_xcd_walk(blockIdx.x, gridDim.x, ntiles, tile, tile_end, tile_step);
    if (tile < tile_end) issue_inputs(tile, 0);
    const int a_rd = p * 64 + ((g ^ swr) << 4);
    const int w_rd = p * 64 + ((g ^ swr) << 4);          // (one wave column: wn = 0)
    int buf = 0;
    for (; tile < tile_end; tile += tile_step, buf ^= 1) {
        int b, y0, x0;
        decode(tile, b, y0, x0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                    // this tile's inputs (and, the first time, the filters) have landed; every wave is done with the previous tile
        if (tile + tile_step < tile_end) issue_inputs(tile + tile_step, buf ^ 1);

        // ---- phase A: hidden = act1(W1 . x + b1) on the 192 halo rows (3 fragments of 16 rows per wave)
        f32x4 hacc[2][3];
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int i = 0; i < 3; ++i) hacc[f][i] = f32x4{0.f, 0.f, 0.f, 0.f};
        {
            const char* xb = hsm + OFF_X + buf * HBYTES;
            const char* wb = hsm + OFF_WA;
            frag_t xf[3], wf[2];
#pragma unroll
            for (int i = 0; i < 3; ++i) xf[i] = *reinterpret_cast<const frag_t*>(xb + (wave * 3 + i) * 1024 + a_rd);
#pragma unroll
            for (int f = 0; f < 2; ++f) wf[f] = *reinterpret_cast<const frag_t*>(wb + (f * 16) * 64 + a_rd);
#pragma unroll
            for (int f = 0; f < 2; ++f)
#pragma unroll
                for (int i = 0; i < 3; ++i) hacc[f][i] = E::mma(wf[f], xf[i], hacc[f][i]);
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int rho = (wave * 3 + i) * 16 + p;
            const int hy = rho / HP, hx = rho - hy * HP;
            const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
            const bool inside = (unsigned)iy < (unsigned)Hi && (unsigned)ix < (unsigned)Wi;
            u32x4 pk;
            sc* e = reinterpret_cast<sc*>(&pk);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float v = hacc[j >> 2][i][j & 3] + b1[j];
                if (a.pre_act == YP_ACT_SILU) v = yp_silu(v);
                e[j] = (sc)(inside ? v : 0.0f);
            }
            *reinterpret_cast<u32x4*>(hsm + OFF_HID + rho * 64 + ((g ^ swr) << 4)) = pk;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();

        // ---- phase B: 3x3 over the resident hidden halo, all nine taps of the filter resident
        f32x4 acc[FN][FM];
#pragma unroll
        for (int f = 0; f < FN; ++f)
#pragma unroll
            for (int fm = 0; fm < FM; ++fm) acc[f][fm] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const char* hb = hsm + OFF_HID;
            const char* wb = hsm + OFF_WB + r * WBYTES;
#pragma unroll
            for (int s_ = 0; s_ < 3; ++s_) {
                const int xc = s_ + r * HP;
                frag_t wf[FN], xf[FM];
#pragma unroll
                for (int f = 0; f < FN; ++f) wf[f] = *reinterpret_cast<const frag_t*>(wb + w_rd + (s_ * BN + f * 16) * 64);
#pragma unroll
                for (int fm = 0; fm < FM; ++fm) {
                    const int rho = (wm * FM + fm) * HP + p + xc;
                    const int sw = (0x3300 >> (((rho >> 2) & 3) * 4)) & 3;
                    xf[fm] = *reinterpret_cast<const frag_t*>(hb + rho * 64 + ((g ^ sw) << 4));
                }
#pragma unroll
                for (int f = 0; f < FN; ++f)
#pragma unroll
                    for (int fm = 0; fm < FM; ++fm) acc[f][fm] = E::mma(wf[f], xf[fm], acc[f][fm]);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                    // every wave has read its last hidden fragment: the bottleneck output replaces it

        // ---- bottleneck output = act(acc + bias) + x (the shortcut: the halo's centre pixel, still resident), 16-bit, into LDS
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) {
            const int ly = wm * FM + fm;
            const int oy = y0 + ly, ox = x0 + p;
            const bool inside = oy < a.Ho && ox < a.Wo;
            const int rho = ly * 16 + p;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float x = acc[j >> 2][fm][j & 3] + bias[j];
                if (a.act == YP_ACT_SILU) x = yp_silu(x);
                v[j] = x;
            }
            if (a.has_res && inside) {
                const int hr = (ly + 1) * HP + p + 1;
                const int sw = (0x3300 >> (((hr >> 2) & 3) * 4)) & 3;
                const u32x4 raw = *reinterpret_cast<const u32x4*>(hsm + OFF_X + buf * HBYTES + hr * 64 + ((g ^ sw) << 4));
                const sc* e = reinterpret_cast<const sc*>(&raw);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] += (float)e[j];
            }
            u32x4 pk;
            sc* e = reinterpret_cast<sc*>(&pk);
#pragma unroll
            for (int j = 0; j < 8; ++j) e[j] = (sc)v[j];
            *reinterpret_cast<u32x4*>(hsm + OFF_HID + rho * 64 + ((g ^ swr) << 4)) = pk;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();

        // ---- C3 tail: out = act(W3 . cat(bottleneck output, in1) + b3)
        f32x4 acc3[FN3][FM];
#pragma unroll
        for (int f = 0; f < FN3; ++f)
#pragma unroll
            for (int fm = 0; fm < FM; ++fm) acc3[f][fm] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int rd = p * 64 + ((g ^ swr) << 4);
#pragma unroll
        for (int kc = 0; kc < KCH3; ++kc) {
            const char* xb = kc == 0 ? hsm + OFF_HID : hsm + OFF_U + buf * UBYTES;
            const char* wb = hsm + OFF_W3 + kc * (N3 * 64);
            frag_t xf[FM];
#pragma unroll
            for (int fm = 0; fm < FM; ++fm) xf[fm] = *reinterpret_cast<const frag_t*>(xb + (wm * FM + fm) * 1024 + rd);
#pragma unroll
            for (int f = 0; f < FN3; ++f) {
                const frag_t wf = *reinterpret_cast<const frag_t*>(wb + f * 1024 + rd);
#pragma unroll
                for (int fm = 0; fm < FM; ++fm) acc3[f][fm] = E::mma(wf, xf[fm], acc3[f][fm]);
            }
        }
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) {
            const int oy = y0 + wm * FM + fm, ox = x0 + p;
            if (oy >= a.Ho || ox >= a.Wo) continue;
            const size_t m = (size_t)(b * a.Ho + oy) * a.Wo + ox;
            char* op = a.out + (m * a.out_cs + a.out_co + g * LPG3) * EB;
#pragma unroll
            for (int h = 0; h < LPG3 / 8; ++h) {
                if (g * LPG3 + h * 8 >= a.post_N) continue;
                u32x4 pk;
                sc* e = reinterpret_cast<sc*>(&pk);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float x = acc3[(h * 8 + j) >> 2][fm][(h * 8 + j) & 3] + b3[h * 8 + j];
                    if (a.post_act == YP_ACT_SILU) x = yp_silu(x);
                    e[j] = (sc)x;
                }
                *reinterpret_cast<u32x4*>(op + h * 16) = pk;
            }
        }
    }
}


// ==========================================================================================
// Stem: Conv(inp_ch<=4 -> Cout, k=6, s=2, p=2) straight from the caller's NCHW fp32 image
// (reference models/YOLOPoint.py:156 + the implicit layout change of `model(inp)`).
// A workgroup owns 8 x 16 output pixels x all Cout channels.  Its 20 x 36 input halo is read from the
// three fp32 planes (coalesced along x), converted and stored in LDS as 4-channel pixels, i.e. 16-byte
// pixel PAIRS; with stride 2 the pair a tap (r, s' = s/2) needs for output x is pair (x + s') of halo row
// (2y + r): consecutive lanes read consecutive 16-byte pairs.  K = 6 rows x 3 pairs x 8 = 144 (5 MFMA k
// steps, the last half empty); the filter ([Cout][6][3][8], the layout the packer already produces for
// the generic path) stays in registers.  No separate pack kernel, no NHWC copy of the image in HBM.
// ==========================================================================================
template <int DT, int FN>
__global__ __launch_bounds__(256) void stem_conv_kernel(const float* __restrict__ x, int B, int C, int H, int W, const char* __restrict__ wgt,
                                                        int Kpad, const float* __restrict__ bias, int act, char* __restrict__ out, int out_cs,
                                                        int out_co, int Cout, int tiles_x, int tiles_y) {
    using E = Elem<DT>;
    using frag_t = typename E::frag;
    using sc = typename E::scalar;
    constexpr int TH = 8, TW = 64, HH = 2 * TH + 4, HPAIR = TW + 2;      // halo: 20 rows x 66 pixel pairs (132 columns)
    constexpr int XF = TW / 16;                                          // x fragments per output row
    constexpr int LPG = 4 * FN;
    __shared__ __attribute__((aligned(16))) sc halo[HH * HPAIR * 8];
    int bid = blockIdx.x;
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int b = bid / tiles_y;
    const int Ho = H / 2, Wo = W / 2;
    const int y0 = ty * TH, x0 = tx * TW;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    // ---- halo fill.  An item is 4 consecutive input columns (= 2 pixel pairs) of one halo row, ALL channels: a thread loads
    // the (up to) 4 planes as unaligned 16-byte vectors -- every load of the workgroup is issued before the first is
    // consumed: one HBM round trip -- interleaves them to 4-channel pixels and writes two 16-byte pairs to LDS.
    typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
    const size_t plane = (size_t)H * W;
    constexpr int QUADS = HPAIR / 2, ITEMS = HH * QUADS, NIT = (ITEMS + 255) / 256;
    f32x4 v[NIT][4];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int item = t + 256 * it;
        const int row = item / QUADS, q = item - row * QUADS;
        const int iy = 2 * y0 - 2 + row, ix = 2 * x0 - 2 + 4 * q;
        const bool rowok = item < ITEMS && (unsigned)iy < (unsigned)H;
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
            v[it][ch] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (rowok && ch < C) {
                const float* src = x + ((size_t)b * C + ch) * plane + (size_t)iy * W + ix;
                if (ix >= 0 && ix + 3 < W) v[it][ch] = *reinterpret_cast<const f32x4u*>(src);
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if ((unsigned)(ix + e) < (unsigned)W) v[it][ch][e] = src[e];
                }
            }
        }
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int item = t + 256 * it;
        if (item >= ITEMS) continue;
        const int row = item / QUADS, q = item - row * QUADS;
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            u32x4 pk;
            sc* e = reinterpret_cast<sc*>(&pk);
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) { e[ch] = (sc)v[it][ch][2 * pr]; e[4 + ch] = (sc)v[it][ch][2 * pr + 1]; }
            *reinterpret_cast<u32x4*>(&halo[(row * HPAIR + 2 * q + pr) * 8]) = pk;
        }
    }
    // ---- filter fragments: lane (n = lane%16, g = lane/16) holds W[n][(4*kk + g)*8 .. +8] for kk = 0..4
    const int p = lane & 15, g = lane >> 4;
    frag_t wf[FN][5];
#pragma unroll
    for (int f = 0; f < FN; ++f) {
        // MFMA row (f, g', r) <-> channel g'*LPG + f*4 + r  (same interleave as the other kernels)
        const int n = (p >> 2) * LPG + f * 4 + (p & 3);
#pragma unroll
        for (int kk = 0; kk < 5; ++kk) {
            const int q = 4 * kk + g;
            if (q < 18 && n < Cout) wf[f][kk] = *reinterpret_cast<const frag_t*>(wgt + ((size_t)n * Kpad + q * 8) * 2);
            else {
#pragma unroll
                for (int j = 0; j < 8; ++j) wf[f][kk][j] = (sc)0.f;
            }
        }
    }
    float bv[LPG];
#pragma unroll
    for (int j = 0; j < LPG; ++j) bv[j] = (bias != nullptr && g * LPG + j < Cout) ? bias[g * LPG + j] : 0.f;
    __syncthreads();
#pragma unroll
    for (int fm = 0; fm < 2; ++fm) {
        const int yl = wave * 2 + fm;
        const int oy = y0 + yl;
#pragma unroll
        for (int xf_ = 0; xf_ < XF; ++xf_) {
            f32x4 acc[FN];
#pragma unroll
            for (int f = 0; f < FN; ++f) acc[f] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < 5; ++kk) {
                const int q = 4 * kk + g;
                const int r = q / 3, sp = q - r * 3;
                frag_t xv;
                if (q < 18) xv = *reinterpret_cast<const frag_t*>(&halo[((2 * yl + r) * HPAIR + xf_ * 16 + p + sp) * 8]);
                else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) xv[j] = (sc)0.f;
                }
#pragma unroll
                for (int f = 0; f < FN; ++f) acc[f] = E::mma(wf[f][kk], xv, acc[f]);
            }
            const int ox = x0 + xf_ * 16 + p;
            if (oy >= Ho || ox >= Wo) continue;
            const size_t m = ((size_t)b * Ho + oy) * Wo + ox;
            constexpr int CW = (LPG % 8 == 0) ? 8 : 4;
#pragma unroll
            for (int h = 0; h < LPG / CW; ++h) {
                const int nc = g * LPG + h * CW;
                if (nc >= Cout) continue;
                sc o[CW];
#pragma unroll
                for (int j = 0; j < CW; ++j) {
                    const int cj = h * CW + j;
                    float v = acc[cj >> 2][cj & 3] + bv[cj];
                    if (act == YP_ACT_SILU) v = yp_silu(v);
                    o[j] = (sc)v;
                }
                char* op = out + (m * out_cs + out_co + nc) * 2;
                if constexpr (CW == 8) *reinterpret_cast<u32x4*>(op) = *reinterpret_cast<const u32x4*>(o);
                else *reinterpret_cast<u32x2*>(op) = *reinterpret_cast<const u32x2*>(o);
            }
        }
    }
}

// ==========================================================================================
// Fused stem + Conv2 (reference models/YOLOPoint.py:156-157): out = act2(conv3x3/s2(act1(stem6x6/s2(image)))) with 32 stem channels.
// The stem's output is the largest activation of the network (8 x 320 x 320 x 32 halves = 52 MB at batch 8, 640 x 640): written once and
// read once, it is 104 of the 169 MB the first two launches move.  Here it never leaves the workgroup:
//   phase A  the image halo of a TH x 16 tile of Conv2 outputs -- (4 TH + 6) rows x 72 columns of the fp32 planes -- is read, converted
//            and stored as 4-channel pixel PAIRS exactly as stem_conv_kernel does; the stem outputs the tile needs ((2 TH + 1) rows x 33
//            columns) are computed by MFMA in fragments of 16 pixels that follow the stride-2 halo layout of conv3x3_halo_kernel (a halo
//            row = the 17 odd-side columns, then the 16 even-side ones: fragments of 16 consecutive LDS rows read every other pixel pair),
//            get bias + activation, are zeroed outside the stem's output image (Conv2's padding acts on the stem OUTPUT) and are stored
//            as 16-bit rows in the swizzled 64-byte-row format the halo DMA would have produced;
//   phase B  conv3x3_halo_kernel's stride-2 tap arithmetic over that resident halo; wave w computes channels [16 w, 16 w + 16) of every pixel of
//            the tile with its nine filter fragments in registers (no filter ring in LDS).
// A workgroup loops over tiles (two resident workgroups per CU), so the filter fragments are fetched once.  HBM traffic: the image once
// (+ halo overlap) and Conv2's output once.  Measured (batch 8, 640 x 640, f16): 49-52 us against 35 + 26 for the two launches; what is left
// is the stem's activation arithmetic (26 M SiLU on the VALU, two transcendentals each), which the two-launch form hid behind its
// memory time.  (A first version with `__launch_bounds__(256, 4)` spilled 9 registers: the scratch set-up alone cost ~20 us per launch.)
// ==========================================================================================
//   phase C  (POST: the 1x1 convolution behind Conv2 -- C3.cv1 + C3.cv2 of Bottleneck1 as one 64 -> 64 filter, a.post_*)  Conv2's activated
//            output of the tile goes to LDS as 128-byte pixel rows (it overwrites the hidden halo), wave w multiplies its 16 output
//            channels (two filter fragments in registers) against every pixel and writes through the ordinary split-destination store:
//            Conv2's output -- read by nothing else -- never reaches HBM either (-52 MB and one launch at batch 8).
template <int DT, int TH, bool POST = false>
__global__ __launch_bounds__(256, 2) void stem_conv2_kernel(const ConvKArgs a) {
    using E = Elem<DT>;
    using frag_t = typename E::frag;
    using sc = typename E::scalar;
    static_assert(E::BYTES == 2, "fused stem: 16-bit element types");
    constexpr int EB = 2, TW = 16, BN = 64, CH = 32;
    constexpr int FM = TH, LPG = 4;                                    // phase B: wave w computes channels [16 w, 16 w + 16) of all TH x 16 pixels
    constexpr int HH = 2 * TH + 1, HP = 34, HROWS = HH * HP, HSLOTS = (HROWS + 15) / 16, HBYTES = HSLOTS * 1024;
    constexpr int IH = 4 * TH + 6, IP = 36, QUADS = IP / 2, ITEMS = IH * QUADS, NIT = (ITEMS + 255) / 256, IBYTES = IH * IP * 16;
    constexpr int FN1 = CH / 16, LPG1 = 4 * FN1;                      // stem: 32 channels = 2 fragments, a lane holds 8 consecutive channels
    constexpr int NFRAG = 2 * HH + (HH + 15) / 16;                     // per halo row: slots 0..15 and 17..32; then the slot-16 pixels of all rows
    static_assert(LPG1 == 8 && BN == 64, "unsupported tile");

    extern __shared__ __attribute__((aligned(1024))) char hsm[];      // [hidden halo][image halo]
    sc* img = reinterpret_cast<sc*>(hsm + HBYTES);

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int p = lane & 15, g = lane >> 4;

    // ---- Conv2's filter stays in registers: wave w multiplies its 16 channels against every pixel of the tile, so it needs 9 fragments
    // (lane (n = p, g): W2[16 w + p][tap][8 g .. 8 g + 8)) and no LDS ring at all -- LDS holds the two halos only, four workgroups per CU
    frag_t wf2[9];
    {
        const int n = wave * 16 + p;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            if (n < a.Npad) wf2[tap] = *reinterpret_cast<const frag_t*>(a.wgt + ((size_t)n * a.Kpad + tap * CH + g * 8) * EB);
            else {
#pragma unroll
                for (int j = 0; j < 8; ++j) wf2[tap][j] = (sc)0.f;
            }
        }
    }
    // ---- stem filter fragments + biases (once per workgroup: it loops over tiles -- with one tile per workgroup the 19 filter
    // fragments per lane were 250 MB of L2 reads per launch, more than the image and the output together)
    frag_t wf1[FN1][5];
#pragma unroll
    for (int f = 0; f < FN1; ++f) {
        const int n = (p >> 2) * LPG1 + f * 4 + (p & 3);
#pragma unroll
        for (int kk = 0; kk < 5; ++kk) {
            const int q = 4 * kk + g;
            if (q < 18) wf1[f][kk] = *reinterpret_cast<const frag_t*>(a.stem_wgt + ((size_t)n * a.stem_Kpad + q * 8) * 2);
            else {
#pragma unroll
                for (int j = 0; j < 8; ++j) wf1[f][kk][j] = (sc)0.f;
            }
        }
    }
    float bv1[LPG1];
#pragma unroll
    for (int j = 0; j < LPG1; ++j) bv1[j] = a.stem_bias != nullptr ? a.stem_bias[g * LPG1 + j] : 0.f;
    const int nb = wave * 16 + g * LPG;
    float bias[LPG];
    yp_load_bias<LPG>(a, nb, bias);
    // POST: the pointwise filter behind Conv2 (64 input channels = two k steps) and its bias, in registers like wf2
    frag_t wf3[2];
    float b3[LPG];
    if constexpr (POST) {
        const int n = wave * 16 + p;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (n < a.post_Npad) wf3[ks] = *reinterpret_cast<const frag_t*>(a.post_wgt + ((size_t)n * a.post_Kpad + ks * 32 + g * 8) * EB);
            else {
#pragma unroll
                for (int j = 0; j < 8; ++j) wf3[ks][j] = (sc)0.f;
            }
        }
        const f32x4 b4 = (a.post_bias != nullptr && nb < a.post_N) ? *reinterpret_cast<const f32x4*>(a.post_bias + nb) : f32x4{0.f, 0.f, 0.f, 0.f};
        b3[0] = b4[0]; b3[1] = b4[1]; b3[2] = b4[2]; b3[3] = b4[3];
    }
    const int H1 = a.Hi, W1 = a.Wi;
    const int ntiles = a.stats_rows;                  // (host: B * tiles_y * tiles_x)
    // ---- image halo of a tile: rows 4 y0 - 4 .. , columns 4 x0 - 4 .. (72 of them), all channels.  The loads of tile i + 1 are issued as soon as
    // tile i's values have been converted into LDS (same registers), so they are in flight during the three MFMA phases of tile i: with two
    // workgroups per CU the load latency of a tile (HBM under load: 2-4 k clocks of a ~17 k clock tile) was exposed about half the time.
    typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
    const int H = a.stem_H, W = a.stem_W, C = a.stem_C;
    const size_t plane = (size_t)H * W;
    f32x4 v[NIT][4];
    auto load_tile = [&](int tl) {
        int bid = tl;
        const int tx = bid % a.tiles_x; bid /= a.tiles_x;
        const int ty = bid % a.tiles_y;
        const int b = bid / a.tiles_y;
        const int y0 = ty * TH, x0 = tx * TW;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int item = t + 256 * it;
            const int row = item / QUADS, q = item - row * QUADS;
            const int iy = 4 * y0 - 4 + row, ix = 4 * x0 - 4 + 4 * q;
            const bool rowok = item < ITEMS && (unsigned)iy < (unsigned)H;
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) {
                v[it][ch] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (rowok && ch < C) {
                    const float* src = a.stem_x + ((size_t)b * C + ch) * plane + (size_t)iy * W + ix;
                    if (ix >= 0 && ix + 3 < W) v[it][ch] = *reinterpret_cast<const f32x4u*>(src);
                    else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if ((unsigned)(ix + e) < (unsigned)W) v[it][ch][e] = src[e];
                    }
                }
            }
        }
    };
    int tile_first, tile_end, tile_step;
    yp_xcd_walk(blockIdx.x, gridDim.x, ntiles, tile_first, tile_end, tile_step);
    if (tile_first < tile_end) load_tile(tile_first);
    for (int tile = tile_first; tile < tile_end; tile += tile_step) {
    int bid = tile;
    const int tx = bid % a.tiles_x; bid /= a.tiles_x;
    const int ty = bid % a.tiles_y;
    const int b = bid / a.tiles_y;
    const int y0 = ty * TH, x0 = tx * TW;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int item = t + 256 * it;
        if (item >= ITEMS) continue;
        const int row = item / QUADS, q = item - row * QUADS;
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            u32x4 pk;
            sc* e = reinterpret_cast<sc*>(&pk);
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) { e[ch] = (sc)v[it][ch][2 * pr]; e[4 + ch] = (sc)v[it][ch][2 * pr + 1]; }
            *reinterpret_cast<u32x4*>(&img[(row * IP + 2 * q + pr) * 8]) = pk;
        }
    }
    if (tile + tile_step < tile_end) load_tile(tile + tile_step);
    __syncthreads();

    // ---- phase A: the stem outputs of the halo, fragment by fragment
    // (two fragments per step: four independent MFMA chains, and the activation arithmetic of one pair overlaps the MFMAs of the next)
    for (int fid0 = wave; fid0 < NFRAG; fid0 += 8) {
        int hy[2], slot[2], pair[2];
        bool lane_ok[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            int fid = fid0 + 4 * u;
            lane_ok[u] = fid < NFRAG;
            if (fid >= NFRAG) fid = fid0;
            if (fid < 2 * HH) {
                hy[u] = fid >> 1;
                if (fid & 1) { slot[u] = 17 + p; pair[u] = 2 * p + 1; } else { slot[u] = p; pair[u] = 2 * p; }
            } else {
                hy[u] = p + 16 * (fid - 2 * HH);
                lane_ok[u] = lane_ok[u] && hy[u] < HH;
                if (hy[u] >= HH) hy[u] = HH - 1;
                slot[u] = 16; pair[u] = 32;
            }
        }
        f32x4 acc1[2][FN1];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int f = 0; f < FN1; ++f) acc1[u][f] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 5; ++kk) {
            const int q = 4 * kk + g;
            const int r = q / 3, sp = q - r * 3;
            frag_t xv[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (q < 18) xv[u] = *reinterpret_cast<const frag_t*>(&img[((2 * hy[u] + r) * IP + pair[u] + sp) * 8]);
                else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) xv[u][j] = (sc)0.f;
                }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int f = 0; f < FN1; ++f) acc1[u][f] = E::mma(wf1[f][kk], xv[u], acc1[u][f]);
        }
        const int nfr = fid0 + 4 < NFRAG ? 2 : 1;           // (wave-uniform: the activation arithmetic is the expensive part of a fragment)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (u >= nfr) break;
            const int sy = 2 * y0 - 1 + hy[u];
            const int sx = slot[u] < 17 ? 2 * (x0 + slot[u]) - 1 : 2 * (x0 + slot[u] - 17);
            const bool inside = (unsigned)sy < (unsigned)H1 && (unsigned)sx < (unsigned)W1;
            u32x4 pk;
            sc* e = reinterpret_cast<sc*>(&pk);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float x = acc1[u][j >> 2][j & 3] + bv1[j];
                if (a.stem_act == YP_ACT_SILU) x = yp_silu(x);
                e[j] = (sc)(inside ? x : 0.0f);
            }
            const int rho = hy[u] * HP + slot[u];
            const int sw = (0x3300 >> (((rho >> 2) & 3) * 4)) & 3;
            if (lane_ok[u]) *reinterpret_cast<u32x4*>(hsm + rho * 64 + ((g ^ sw) << 4)) = pk;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (the hidden-halo writes; NOT vmcnt: the next tile's image loads stay in flight)
    __builtin_amdgcn_s_barrier();

    // ---- phase B: 3x3 / stride 2 over the resident halo
    f32x4 acc[FM];
#pragma unroll
    for (int fm = 0; fm < FM; ++fm) acc[fm] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            constexpr int XC2[3] = {0, 17, 1};
            const int xc = XC2[s] + r * HP + p;
#pragma unroll
            for (int fm = 0; fm < FM; ++fm) {
                const int rho = (fm * 2) * HP + xc;
                const int sw = (0x3300 >> (((rho >> 2) & 3) * 4)) & 3;
                const frag_t xf = *reinterpret_cast<const frag_t*>(hsm + rho * 64 + ((g ^ sw) << 4));
                acc[fm] = E::mma(wf2[r * 3 + s], xf, acc[fm]);
            }
        }
    }
    if constexpr (POST) {
        // ---- phase C: y = act2(Conv2) as 128-byte pixel rows in LDS (16-byte chunks swizzled by the pixel), then the 64 -> 64 pointwise filter
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                     // every wave has read its last hidden-halo fragment
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) {
            const int pix = fm * 16 + p;
            u32x2 pk;
            sc* e = reinterpret_cast<sc*>(&pk);
#pragma unroll
            for (int cj = 0; cj < LPG; ++cj) {
                float x = acc[fm][cj] + bias[cj];
                if (a.act == YP_ACT_SILU) x = yp_silu(x);
                e[cj] = (sc)x;
            }
            *reinterpret_cast<u32x2*>(hsm + pix * 128 + (((2 * wave + (g >> 1)) ^ (pix & 7)) << 4) + (g & 1) * 8) = pk;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        f32x4 acc3[FM];
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) acc3[fm] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int fm = 0; fm < FM; ++fm) {
                const int pix = fm * 16 + p;
                const frag_t yf = *reinterpret_cast<const frag_t*>(hsm + pix * 128 + (((4 * ks + g) ^ (pix & 7)) << 4));
                acc3[fm] = E::mma(wf3[ks], yf, acc3[fm]);
            }
        yp_pin_arrived(b3);
        auto epilogue3 = [&](auto res_c) {
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) {
            const int oy = y0 + fm, ox = x0 + p;
            if (oy >= a.Ho || ox >= a.Wo || nb >= a.Cout) continue;
            const int m = (b * a.Ho + oy) * a.Wo + ox;
            float v[LPG];
#pragma unroll
            for (int cj = 0; cj < LPG; ++cj) {
                float x = acc3[fm][cj] + b3[cj];
                if (a.post_act == YP_ACT_SILU) x = yp_silu(x);
                v[cj] = x;
            }
            yp_store_chunk<DT, false, LPG, decltype(res_c)::value>(a, m, nb, v);
        }
        };
        YP_RES_DISPATCH(a, epilogue3);
    } else {
    yp_pin_arrived(bias);
    auto epilogue = [&](auto res_c) {
#pragma unroll
    for (int fm = 0; fm < FM; ++fm) {
        const int oy = y0 + fm, ox = x0 + p;
        if (oy >= a.Ho || ox >= a.Wo) continue;
        const int m = (b * a.Ho + oy) * a.Wo + ox;
        yp_epilogue_pixel<DT, false, LPG, decltype(res_c)::value>(a, m, nb, bias, [&](int cj) { return acc[fm][cj]; });
    }
    };
    YP_RES_DISPATCH(a, epilogue);
    }
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
namespace {

struct TileCfg { int id, bm, bn; };
// 4-stage rings.  (8-stage rings -- 7 k tiles in flight -- were measured on every 1x1 layer shape of YOLOPoint-s and were
// never faster: the k loop is not the latency chain that bounds the short layers.  The kernel keeps NS a parameter.)
// (A 64 x 256 tile for the fused Detect convolution -- every channel of a pixel in one workgroup, so that an anchor's rows form
// one contiguous 16-byte-aligned run per tile, written with 16-byte stores -- measured 62-87 us vs 43-47 us for the 64 x 32 / 64 x 64
// tiles on the 80 x 80 level and was dropped.)
constexpr TileCfg kTiles[] = {{1, 128, 32}, {2, 128, 64}, {3, 128, 128}, {4, 64, 64}, {5, 64, 32},
                              {21, 128, 32}, {22, 128, 64}, {23, 128, 128}, {24, 64, 64}, {25, 64, 32}, {26, 64, 64}, {27, 128, 128},
                              {31, 64, 64}, {33, 64, 32}
#ifdef YP_TIMELINE
    , {6, 64, 64}, {7, 128, 128}, {8, 128, 64}      // probe build only: the same tiles with 8-stage rings
#endif
};

template <int DT, bool OUT_F32, bool FAST, bool DETECT, int BM, int BN, int WAVES_M, int WAVES_N, int NS, bool STATS = false, int KPB = 0, bool WSK = false>
hipError_t launch_tile(const ConvKArgs& a, int nblk, hipStream_t st) {
#ifdef YP_PROBE_BNFUSE
    constexpr size_t lds0 = (size_t)NS * (KPB > 0 ? KPB : 1) * (BM / 16 + BN / 16) * 1024;
    const size_t lds = lds0 + (KPB == 0 ? (size_t)a.Kpad * 8 : 0);
    auto kern = conv_igemm_kernel<DT, OUT_F32, FAST, DETECT, BM, BN, WAVES_M, WAVES_N, NS, STATS, KPB, WSK>;
    { static YpLdsAttr attr; if (hipError_t e = yp_set_max_lds(attr, (const void*)kern, 96 * 1024); e != hipSuccess) return e; }
#else
    constexpr size_t lds = (size_t)NS * (KPB > 0 ? KPB : 1) * (BM / 16 + BN / 16) * 1024;
    auto kern = conv_igemm_kernel<DT, OUT_F32, FAST, DETECT, BM, BN, WAVES_M, WAVES_N, NS, STATS, KPB, WSK>;
    if constexpr (lds > 65536) {
        static YpLdsAttr attr;        // per instantiation, per device
        if (hipError_t e = yp_set_max_lds(attr, (const void*)kern, (int)lds); e != hipSuccess) return e;
    }
#endif
    kern<<<dim3(nblk, a.ksplit), 256, lds, st>>>(a);
    return hipGetLastError();
}

template <int DT, bool OUT_F32, bool FAST, bool DETECT = false, bool STATS = false>
hipError_t launch_cfg(int tile, const ConvKArgs& a, int nblk, hipStream_t st) {
    constexpr bool V2OK = FAST && (DT == YP_F16 || DT == YP_BF16) && !STATS;
    switch (tile) {
        case 1: return launch_tile<DT, OUT_F32, FAST, DETECT, 128, 32, 4, 1, 4, STATS>(a, nblk, st);
        case 2: return launch_tile<DT, OUT_F32, FAST, DETECT, 128, 64, 4, 1, 4, STATS>(a, nblk, st);
        case 3: return launch_tile<DT, OUT_F32, FAST, DETECT, 128, 128, 2, 2, 4, STATS>(a, nblk, st);
        case 4: return launch_tile<DT, OUT_F32, FAST, DETECT, 64, 64, 2, 2, 4, STATS>(a, nblk, st);
        case 5: return launch_tile<DT, OUT_F32, FAST, DETECT, 64, 32, 4, 1, 4, STATS>(a, nblk, st);
        // second-generation main loop (ids 21..27: tiles as 1..5, KPB k tiles per barrier, register double-buffered fragments)
        case 21: if constexpr (V2OK) return launch_tile<DT, OUT_F32, FAST, DETECT, 128, 32, 4, 1, 3, STATS, 2>(a, nblk, st); else return hipErrorInvalidValue;
        case 22: if constexpr (V2OK) return launch_tile<DT, OUT_F32, FAST, DETECT, 128, 64, 4, 1, 3, STATS, 2>(a, nblk, st); else return hipErrorInvalidValue;
        case 23: if constexpr (V2OK) return launch_tile<DT, OUT_F32, FAST, DETECT, 128, 128, 2, 2, 3, STATS, 2>(a, nblk, st); else return hipErrorInvalidValue;
        case 24: if constexpr (V2OK) return launch_tile<DT, OUT_F32, FAST, DETECT, 64, 64, 2, 2, 3, STATS, 2>(a, nblk, st); else return hipErrorInvalidValue;
        case 25: if constexpr (V2OK) return launch_tile<DT, OUT_F32, FAST, DETECT, 64, 32, 4, 1, 3, STATS, 2>(a, nblk, st); else return hipErrorInvalidValue;
        case 26: if constexpr (V2OK) return launch_tile<DT, OUT_F32, FAST, DETECT, 64, 64, 2, 2, 3, STATS, 4>(a, nblk, st); else return hipErrorInvalidValue;
        case 27: if constexpr (V2OK) return launch_tile<DT, OUT_F32, FAST, DETECT, 128, 128, 2, 2, 2, STATS, 4>(a, nblk, st); else return hipErrorInvalidValue;
        // waves-split-k (ids 31..33): 64 x 64 / 128 x 64 / 64 x 32 tiles, each wave one k tile of a 4-tile stage against the whole tile
        case 31: if constexpr (V2OK) return launch_tile<DT, OUT_F32, FAST, DETECT, 64, 64, 2, 2, 3, STATS, 4, true>(a, nblk, st); else return hipErrorInvalidValue;
        // (a 128 x 64 tile -- 32 accumulator fragments per wave -- does not fit: the compiler runs out of scalar registers for the DMA operands)
        case 33: if constexpr (V2OK) return launch_tile<DT, OUT_F32, FAST, DETECT, 64, 32, 4, 1, 4, STATS, 4, true>(a, nblk, st); else return hipErrorInvalidValue;
#ifdef YP_TIMELINE
        case 6: if constexpr (!DETECT && !STATS && FAST) return launch_tile<DT, OUT_F32, FAST, DETECT, 64, 64, 2, 2, 8, STATS>(a, nblk, st); else return hipErrorInvalidValue;
        case 7: if constexpr (!DETECT && !STATS && FAST) return launch_tile<DT, OUT_F32, FAST, DETECT, 128, 128, 2, 2, 8, STATS>(a, nblk, st); else return hipErrorInvalidValue;
        case 8: if constexpr (!DETECT && !STATS && FAST) return launch_tile<DT, OUT_F32, FAST, DETECT, 128, 64, 4, 1, 8, STATS>(a, nblk, st); else return hipErrorInvalidValue;
#endif
        default: return hipErrorInvalidValue;
    }
}


template <int DT, bool OUT_F32, int STRIDE, int BN, int WAVES_M, int TH, bool STATS = false>
hipError_t launch_halo(const ConvKArgs& a, int nblk, hipStream_t st) {
    constexpr int HSLOTS = (Halo<STRIDE, TH>::HH * Halo<STRIDE, TH>::HP + 15) / 16;
    constexpr size_t lds2 = (size_t)2 * HSLOTS * 1024 + (size_t)3 * 3 * (BN / 16) * 1024;
    const size_t lds = lds2 - (a.Cin > 32 ? 0 : (size_t)HSLOTS * 1024);
    auto kern = conv3x3_halo_kernel<DT, OUT_F32, STRIDE, BN, WAVES_M, TH, STATS>;
    static YpLdsAttr attr;        // per instantiation, per device
    if (hipError_t e = yp_set_max_lds(attr, (const void*)kern, (int)lds2); e != hipSuccess) return e;
    kern<<<nblk, 256, lds, st>>>(a);
    return hipGetLastError();
}

// th = 8: 8 x 16 output tiles (ids 10..12); th = 4: 4 x 16 tiles (ids 13..15) -- twice the workgroups, half the halo in LDS
template <int DT, bool OUT_F32, bool STATS = false>
hipError_t dispatch_halo(int stride, int bn, int th, const ConvKArgs& a, int nblk, hipStream_t st) {
#define YP_HALO(S, B, WM, T) launch_halo<DT, OUT_F32, S, B, WM, T, STATS>(a, nblk, st)
    if (th == 8) {
        if (stride == 1) return bn == 128 ? YP_HALO(1, 128, 2, 8) : (bn == 64 ? YP_HALO(1, 64, 4, 8) : YP_HALO(1, 32, 4, 8));
        return bn == 128 ? YP_HALO(2, 128, 2, 8) : (bn == 64 ? YP_HALO(2, 64, 4, 8) : YP_HALO(2, 32, 4, 8));
    }
    if (stride == 1) return bn == 128 ? YP_HALO(1, 128, 2, 4) : (bn == 64 ? YP_HALO(1, 64, 4, 4) : YP_HALO(1, 32, 4, 4));
    return bn == 128 ? YP_HALO(2, 128, 2, 4) : (bn == 64 ? YP_HALO(2, 64, 4, 4) : YP_HALO(2, 32, 4, 4));
#undef YP_HALO
}

template <int DT, int C, int BN, int WAVES_M, bool POST, int NWAVES = 4>
hipError_t launch_bneck(const ConvKArgs& a, int nblk, hipStream_t st) {
    constexpr size_t hid = (size_t)(C / 32) * 12 * 1024;
    constexpr size_t opa = (size_t)(C / 32) * (12 + C / 16) * 1024, opb = (size_t)3 * 3 * (BN / 16) * 1024;
    constexpr size_t w3 = POST ? (size_t)(2 * C / 32) * (2 * C) * 64 : 0, ub = POST ? (size_t)(C / 32) * 8192 : 0;
    constexpr size_t ring = opa > opb ? (opa > w3 ? opa : w3) : (opb > w3 ? opb : w3);
    constexpr size_t lds = hid + ring + ub;
    static_assert(lds <= 160 * 1024, "LDS budget");
    auto kern = bottleneck_halo_kernel<DT, C, BN, WAVES_M, POST, NWAVES>;
    static YpLdsAttr attr;        // per instantiation, per device
    if (hipError_t e = yp_set_max_lds(attr, (const void*)kern, (int)lds); e != hipSuccess) return e;
    kern<<<nblk, 64 * NWAVES, lds, st>>>(a);
    return hipGetLastError();
}

template <int DT>
hipError_t launch_bneck32_persist(const ConvKArgs& a, int ntiles, hipStream_t st) {
    constexpr size_t lds = 80 * 1024;
    auto kern = bneck32_persist_kernel<DT>;
    static YpLdsAttr attr;        // per instantiation, per device
    if (hipError_t e = yp_set_max_lds(attr, (const void*)kern, (int)lds); e != hipSuccess) return e;
    const int nb = ntiles < 2 * 256 ? ntiles : 2 * 256;       // (two resident workgroups per CU)
    kern<<<nb, 256, lds, st>>>(a);
    return hipGetLastError();
}

template <int DT>
hipError_t dispatch_bneck(int c, int bn, bool post, bool waves8, const ConvKArgs& a, int nblk, hipStream_t st) {
    if (post) return c == 32 ? launch_bneck<DT, 32, 32, 4, true>(a, nblk, st) : launch_bneck<DT, 64, 64, 4, true>(a, nblk, st);
    if (waves8) {                // tile ids 17 / 18 / 19: eight wavefronts (two per SIMD), BN = 32 / 64 / 128
        if (c == 32) return launch_bneck<DT, 32, 32, 8, false, 8>(a, nblk, st);
        if (c == 64) return bn == 32 ? launch_bneck<DT, 64, 32, 8, false, 8>(a, nblk, st) : launch_bneck<DT, 64, 64, 4, false, 8>(a, nblk, st);
        if (bn == 32) return launch_bneck<DT, 128, 32, 8, false, 8>(a, nblk, st);
        if (bn == 64) return launch_bneck<DT, 128, 64, 4, false, 8>(a, nblk, st);
        return launch_bneck<DT, 128, 128, 4, false, 8>(a, nblk, st);
    }
    if (c == 32) return launch_bneck<DT, 32, 32, 4, false>(a, nblk, st);
    if (c == 64) return bn == 32 ? launch_bneck<DT, 64, 32, 4, false>(a, nblk, st) : launch_bneck<DT, 64, 64, 4, false>(a, nblk, st);
    if (bn == 32) return launch_bneck<DT, 128, 32, 4, false>(a, nblk, st);
    if (bn == 64) return launch_bneck<DT, 128, 64, 4, false>(a, nblk, st);
    return launch_bneck<DT, 128, 128, 2, false>(a, nblk, st);
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// Two translation units from this one file (compile time: 285 kernel instantiations, ~4.5 minutes in one piece).  -DYP_PART=1: the generic
// kernel's instantiations + the host API; -DYP_PART=2: the halo family (conv3x3_halo / bottleneck_halo / bneck32_persist / stem_conv2 /
// stem_conv) behind the plain functions below; YP_PART undefined (probe builds): everything, as before.  A kernel template is only compiled
// where it is instantiated, so the split is a matter of where the launch helpers are CALLED.
// ---------------------------------------------------------------------------------------------
constexpr int YP_STEM2_TH = 4;          // output rows per tile of the fused stem (host tile count and kernel template agree on it)
hipError_t yp_halo_dispatch(int dtype, bool of32, bool stats, int stride, int bn, int th, const ConvKArgs& a, int nblk, hipStream_t st);
hipError_t yp_bneck_dispatch(int dtype, bool persist, int c, int bn, bool post, bool waves8, const ConvKArgs& a, int nblk, hipStream_t st);
hipError_t yp_stem2_dispatch(int dtype, bool post3, const ConvKArgs& a, int nbs, hipStream_t st);
#if YP_PART != 1
hipError_t yp_halo_dispatch(int dtype, bool of32, bool stats, int stride, int bn, int th, const ConvKArgs& a, int nblk, hipStream_t st) {
    if (stats) {
        if (of32) return hipErrorInvalidValue;
        return dtype == YP_F16 ? dispatch_halo<YP_F16, false, true>(stride, bn, th, a, nblk, st)
             : dtype == YP_BF16 ? dispatch_halo<YP_BF16, false, true>(stride, bn, th, a, nblk, st)
             : dtype == YP_FP8 ? dispatch_halo<YP_FP8, false, true>(stride, bn, th, a, nblk, st) : hipErrorInvalidValue;
    }
    if (dtype == YP_FP8) return dispatch_halo<YP_FP8, false>(stride, bn, th, a, nblk, st);
    if (dtype == YP_FP8_BF8) return dispatch_halo<YP_FP8_BF8, false>(stride, bn, th, a, nblk, st);
    if (dtype == YP_F16) return of32 ? dispatch_halo<YP_F16, true>(stride, bn, th, a, nblk, st) : dispatch_halo<YP_F16, false>(stride, bn, th, a, nblk, st);
    if (dtype == YP_BF16) return of32 ? dispatch_halo<YP_BF16, true>(stride, bn, th, a, nblk, st) : dispatch_halo<YP_BF16, false>(stride, bn, th, a, nblk, st);
    return hipErrorInvalidValue;
}
hipError_t yp_bneck_dispatch(int dtype, bool persist, int c, int bn, bool post, bool waves8, const ConvKArgs& a, int nblk, hipStream_t st) {
    if (dtype != YP_F16 && dtype != YP_BF16) return hipErrorInvalidValue;
    if (persist) return dtype == YP_F16 ? launch_bneck32_persist<YP_F16>(a, nblk, st) : launch_bneck32_persist<YP_BF16>(a, nblk, st);
    return dtype == YP_F16 ? dispatch_bneck<YP_F16>(c, bn, post, waves8, a, nblk, st) : dispatch_bneck<YP_BF16>(c, bn, post, waves8, a, nblk, st);
}
hipError_t yp_stem2_dispatch(int dtype, bool post3, const ConvKArgs& a, int nbs, hipStream_t st) {
    constexpr int TH = YP_STEM2_TH;
    constexpr int HSL = ((2 * TH + 1) * 34 + 15) / 16;
    constexpr size_t lds = (size_t)HSL * 1024 + (size_t)(4 * TH + 6) * 36 * 16;
    static_assert((size_t)HSL * 1024 >= (size_t)TH * 16 * 128, "phase C's pixel rows fit the hidden halo");
    hipError_t e = hipSuccess;
#define YP_STEM2(DTC, P3) { static YpLdsAttr attr; e = yp_set_max_lds(attr, (const void*)stem_conv2_kernel<DTC, TH, P3>, (int)lds); \
                            if (e == hipSuccess) { stem_conv2_kernel<DTC, TH, P3><<<nbs, 256, lds, st>>>(a); e = hipGetLastError(); } }
    if (dtype == YP_F16) { if (post3) YP_STEM2(YP_F16, true) else YP_STEM2(YP_F16, false) }
    else if (dtype == YP_BF16) { if (post3) YP_STEM2(YP_BF16, true) else YP_STEM2(YP_BF16, false) }
    else e = hipErrorInvalidValue;
#undef YP_STEM2
    return e;
}
#endif

#if YP_PART != 2
namespace {
int pick_tile(int M, int N) {
    auto blocks = [&](int bm, int bn) { return (long)yp_cdiv(M, bm) * yp_cdiv(N, bn); };
    const long fill = 2 * 256;   // >= 2 workgroups per CU before a bigger tile is worth it
    if (N <= 32) return blocks(128, 32) >= fill ? 1 : 5;
    if (N >= 128 && blocks(128, 128) >= fill) return 3;
    if (blocks(128, 64) >= fill) return 2;
    return 4;
}

}  // namespace

extern "C" int yp_conv_kpad(int K, int dtype) {
    const int g = dtype == YP_F32 ? 32 : ((dtype == YP_FP8 || dtype == YP_FP8_BF8) ? 128 : 64);   // 128 bytes of k per packed row granule
    return yp_cdiv(K, g) * g;
}

// (yp_conv_bn_partial_rows: walk the dispatch decisions of a launch and report the number of bn_partial rows instead of launching)
static thread_local int* g_bn_rows_query = nullptr;

int yp_conv2d_launch(const YpConvDesc* d, const YpDetectDesc* det, hipStream_t stream) {
    YP_REQUIRE(d != nullptr, "yp_conv2d: null descriptor");
    YP_REQUIRE(d->dtype == YP_F16 || d->dtype == YP_BF16 || d->dtype == YP_F32 || d->dtype == YP_FP8 || d->dtype == YP_FP8_BF8, "yp_conv2d: bad dtype %d", d->dtype);
    const bool q8 = d->dtype == YP_FP8 || d->dtype == YP_FP8_BF8;
    const int ce = d->dtype == YP_F32 ? 4 : (q8 ? 16 : 8);
    YP_REQUIRE(!q8 || (d->scale_in != nullptr && d->scale_w != nullptr && det == nullptr && d->pre_weight == nullptr && !d->out_f32 && d->ksplit <= 1 &&
                       !d->atomic_accumulate && d->tail_zero),
               "yp_conv2d: 8-bit inputs need scale_in / scale_w, tail_zero buffers and a plain 16-bit-output convolution");
    // fused C3 tail (post_weight): in1 is the second input of the TAIL convolution and `out` its destination; the 3x3 itself maps
    // in0.C -> in0.C channels
    const bool post = d->post_weight != nullptr && d->stem_weight == nullptr;      // (with stem_weight, post_weight is the fused stem's pointwise stage)
    const int Cin = post ? d->in0.C : d->in0.C + d->in1.C;
    const int Cout = post ? d->in0.C : d->out.C + d->out2.C;
    YP_REQUIRE(d->in0.ptr && (d->out.ptr || det) && d->weight, "yp_conv2d: null buffer");
    YP_REQUIRE(d->in0.C > 0 && d->in0.C % ce == 0 && d->in1.C % ce == 0, "yp_conv2d: input channels (%d,%d) must be multiples of %d", d->in0.C, d->in1.C, ce);
    YP_REQUIRE(d->in0.cstride % ce == 0 && d->in0.coff % ce == 0, "yp_conv2d: in0 slice not 16-byte aligned");
    YP_REQUIRE(d->in1.C == 0 || (d->in1.ptr && d->in1.cstride % ce == 0 && d->in1.coff % ce == 0), "yp_conv2d: in1 slice not 16-byte aligned");
    YP_REQUIRE(d->out.C > 0 && d->out.C % 8 == 0 && d->out.cstride % 8 == 0 && d->out.coff % 8 == 0, "yp_conv2d: output slice (C=%d cs=%d co=%d) must be multiples of 8", d->out.C, d->out.cstride, d->out.coff);
    YP_REQUIRE(!post || (d->pre_weight != nullptr && det == nullptr), "yp_conv2d: the fused C3 tail (post_weight) extends the fused Bottleneck (pre_weight)");
    YP_REQUIRE(d->out2.C == 0 || (d->out2.ptr && d->out2.C % 8 == 0 && d->out2.cstride % 8 == 0 && d->out2.coff % 8 == 0 && d->out2.H == d->Ho && d->out2.W == d->Wo && d->res.C == 0), "yp_conv2d: bad second output view");
    YP_REQUIRE(d->res.C == 0 || (d->res.ptr && d->res.C == Cout && d->res.cstride % 8 == 0 && d->res.coff % 8 == 0 && !d->out_f32), "yp_conv2d: bad residual view");
    YP_REQUIRE(d->B > 0 && d->Ho > 0 && d->Wo > 0 && d->R > 0 && d->S > 0, "yp_conv2d: bad dims");
    YP_REQUIRE(d->in0.ups >= 0 && d->in0.ups <= 1 && d->in1.ups >= 0 && d->in1.ups <= 1, "yp_conv2d: ups must be 0/1");
    YP_REQUIRE((d->in0.H << d->in0.ups) == d->Hi && (d->in0.W << d->in0.ups) == d->Wi, "yp_conv2d: in0 %dx%d<<%d != logical %dx%d", d->in0.H, d->in0.W, d->in0.ups, d->Hi, d->Wi);
    YP_REQUIRE(d->in1.C == 0 || ((d->in1.H << d->in1.ups) == d->Hi && (d->in1.W << d->in1.ups) == d->Wi), "yp_conv2d: in1 dims mismatch");
    YP_REQUIRE(d->out.H == d->Ho && d->out.W == d->Wo, "yp_conv2d: out view dims mismatch");
    const int dil_h = d->dil_h > 0 ? d->dil_h : 1, dil_w = d->dil_w > 0 ? d->dil_w : 1;
    {   // the caller may ask for fewer output rows / columns than the full convolution yields (wgrad of a strided conv)
        const int fullH = (d->Hi + 2 * d->pad_h - dil_h * (d->R - 1) - 1) / d->stride_h + 1;
        const int fullW = (d->Wi + 2 * d->pad_w - dil_w * (d->S - 1) - 1) / d->stride_w + 1;
        // (out_phase: a (1 + py) x (1 + px)-tap launch over dout with as many output pixels as dout has -- the taps behind the last row / column
        // read zeros through the kernels' bounds checks)
        YP_REQUIRE(d->out_phase != 0 || (d->Ho <= fullH && d->Wo <= fullW && (dil_h > 1 || dil_w > 1 || (d->Ho == fullH && d->Wo == fullW))),
                   "yp_conv2d: Ho/Wo (%d,%d) inconsistent with input, filter, stride, pad, dilation (%d,%d)", d->Ho, d->Wo, fullH, fullW);
    }
    YP_REQUIRE(d->out_phase == 0 || (d->out_phase >= 1 && d->out_phase <= 4 && d->out2.C == 0 && det == nullptr && d->bn_partial == nullptr && d->ksplit <= 1 &&
                                     !d->atomic_accumulate && d->pre_weight == nullptr && d->stride_h == 1 && d->stride_w == 1 && d->pad_h == 0 && d->pad_w == 0 &&
                                     d->Ho == d->Hi && d->Wo == d->Wi && d->R <= 2 && d->S <= 2 && dil_h == 1 && dil_w == 1),
               "yp_conv2d: out_phase is a plain stride-1, pad-0 launch of at most 2x2 taps with Ho x Wo = Hi x Wi (no out2 / Detect / bn_partial / ksplit / prologue)");
    const int ksplit = d->ksplit > 1 ? d->ksplit : 1;
    YP_REQUIRE(ksplit == 1 || (d->out_f32 && d->act == YP_ACT_NONE && d->res.C == 0 && d->out2.C == 0 && d->bias == nullptr && ksplit <= 4096), "yp_conv2d: split-K needs a plain fp32 accumulation target");
    YP_REQUIRE(!d->in0_zero_stuffed || d->in0.ups == 1, "yp_conv2d: a zero-stuffed input is addressed through ups = 1");
    const int Kreal = d->R * d->S * Cin;
    YP_REQUIRE(d->Kpad >= Kreal && d->Kpad % (d->dtype == YP_F32 ? 16 : (q8 ? 64 : 32)) == 0, "yp_conv2d: Kpad %d invalid for K=%d", d->Kpad, Kreal);
    YP_REQUIRE(d->Npad >= Cout, "yp_conv2d: Npad %d < Cout %d", d->Npad, Cout);
    const long Ml = (long)d->B * d->Ho * d->Wo;
    YP_REQUIRE(Ml < (1l << 30), "yp_conv2d: too many output pixels");

    ConvKArgs a{};
    a.in0 = (const char*)d->in0.ptr; a.in1 = (const char*)(d->in1.C ? d->in1.ptr : d->in0.ptr);
    a.wgt = (const char*)d->weight; a.bias = d->bias; a.res = (const char*)d->res.ptr; a.out = (char*)d->out.ptr;
    a.in0_cs = d->in0.cstride; a.in0_co = d->in0.coff; a.in0_C = d->in0.C; a.in0_ups = d->in0.ups; a.in0_H = d->in0.H; a.in0_W = d->in0.W;
    a.in1_cs = d->in1.cstride; a.in1_co = d->in1.coff; a.in1_ups = d->in1.ups; a.in1_H = d->in1.H; a.in1_W = d->in1.W;
    a.res_cs = d->res.cstride; a.res_co = d->res.coff; a.has_res = d->res.C != 0;
    a.out_cs = d->out.cstride; a.out_co = d->out.coff;
    a.out_sub = d->out_phase != 0; a.out_py = (d->out_phase - 1) >> 1; a.out_px = (d->out_phase - 1) & 1;
    a.out2 = (char*)d->out2.ptr; a.out2_cs = d->out2.cstride; a.out2_co = d->out2.coff; a.split = d->out.C;
    a.Hi = d->Hi; a.Wi = d->Wi; a.Wo = d->Wo; a.HoWo = d->Ho * d->Wo;
    a.Cin = Cin; a.Cout = Cout; a.Kreal = Kreal; a.Kpad = d->Kpad; a.Npad = d->Npad;
    a.R = d->R; a.S = d->S; a.RS = d->R * d->S; a.invS = (65536 + d->S - 1) / d->S;
    { const int bk = d->dtype == YP_F32 ? 16 : (q8 ? 64 : 32); a.dt = bk / Cin; a.dc = bk % Cin; }
    a.scale_in = d->scale_in; a.scale_w = d->scale_w;
    a.sh = d->stride_h; a.sw = d->stride_w; a.ph = d->pad_h; a.pw = d->pad_w;
    a.act = d->act; a.M = (int)Ml;
    a.dil_h = dil_h; a.dil_w = dil_w; a.in0_zs = d->in0_zero_stuffed ? 1 : 0; a.ksplit = ksplit; a.atomic_out = ksplit > 1 || d->atomic_accumulate;
    a.split_slabs = ksplit > 1 ? d->split_slabs : nullptr; a.split_stride = d->split_stride;
    YP_REQUIRE(a.split_slabs == nullptr || (d->split_stride >= (int64_t)Ml * d->out.cstride && !d->atomic_accumulate), "yp_conv2d: split_stride must cover one partial output");

    int tile = ((d->tile >= 1 && d->tile <= 8) || (d->tile >= 21 && d->tile <= 27) || (d->tile == 31 || d->tile == 33)) ? d->tile : pick_tile(a.M, Cout);
    const TileCfg* tc = nullptr;
    for (const auto& c : kTiles) if (c.id == tile) tc = &c;
    YP_REQUIRE(tc != nullptr, "yp_conv2d: unknown tile id %d", tile);
    a.tiles_n = yp_cdiv(Cout, tc->bn);
    const int nblk = yp_cdiv(a.M, tc->bm) * a.tiles_n;
    // ceil(2^32 / d) divides every n with n * d < 2^32 exactly by one multiply-high: pixel index -> (b, y, x) and workgroup -> (tile_m, tile_n)
    if ((unsigned long long)(a.M + 255) * (unsigned long long)a.HoWo < (1ull << 32) && a.HoWo >= 2 && a.Wo >= 2) {
        a.mg_howo = (unsigned)(((1ull << 32) + a.HoWo - 1) / a.HoWo);
        a.mg_wo = (unsigned)(((1ull << 32) + a.Wo - 1) / a.Wo);
    }
    if (a.tiles_n >= 2 && (unsigned long long)(nblk + 8) * (unsigned long long)a.tiles_n < (1ull << 32)) a.mg_tn = (unsigned)(((1ull << 32) + a.tiles_n - 1) / a.tiles_n);

    // FAST DMA addressing: wave-uniform taps (k tile never straddles a tap or the source boundary), 16 zero
    // bytes behind each input buffer and a zero filter row behind the packed weights, 32-bit byte offsets.
    const int bk = d->dtype == YP_F32 ? 16 : (q8 ? 64 : 32);
    const int eb = yp_dtype_bytes(d->dtype);
    const size_t in0_bytes = (size_t)d->B * d->in0.H * d->in0.W * d->in0.cstride * eb;
    const size_t in1_bytes = d->in1.C ? (size_t)d->B * d->in1.H * d->in1.W * d->in1.cstride * eb : 0;
    const size_t wgt_bytes = (size_t)d->Npad * d->Kpad * eb;
    const bool fast = d->tail_zero && Cin % bk == 0 && d->in0.C % bk == 0 && dil_h == 1 && dil_w == 1 && d->S <= 8 && d->R * d->S <= 64 && in0_bytes < (1ull << 31) && in1_bytes < (1ull << 31) &&
                      wgt_bytes < (1ull << 31);
    a.in0_zoff = (unsigned)in0_bytes; a.in1_zoff = (unsigned)in1_bytes; a.wgt_zrow = (unsigned)wgt_bytes;

    hipError_t e;
    const bool of32 = d->out_f32 != 0;
    {   // 8-wave 32x32x16 kernels (conv_mma8.hip), tile ids 41..44: the compute-bound 16-bit layers (every source a multiple of 64 channels)
        int bp8 = 0, bc8 = 0, srows = 0;
        if (yp_mma8_tile_dims(d->tile, &bp8, &bc8, &srows)) {
            const int cg = q8 ? 128 : 64;        // channels per 128-byte k row; the 8-bit (block-scaled MFMA) instantiation exists for tile 57 only
            YP_REQUIRE(fast && d->dtype != YP_F32 && (!q8 || d->tile == 57) && Cin % cg == 0 && d->in0.C % cg == 0 && d->Kpad % cg == 0 && det == nullptr &&
                       d->pre_weight == nullptr && ksplit == 1 && !a.atomic_out && d->R * d->S <= 32,
                       "yp_conv2d: tile %d (8-wave kernel) needs a 16-bit (8-bit: tile 57) fast-path convolution with channel counts %% %d == 0 (Cin %d, in0.C %d)", d->tile, cg, Cin, d->in0.C);
            a.tiles_n = yp_cdiv(Cout, bc8);
            const int nblk8 = yp_cdiv(a.M, bp8) * a.tiles_n;
            const bool stats = d->bn_partial != nullptr;
            if (stats) {
                YP_REQUIRE(!of32 && d->bias == nullptr && d->act == YP_ACT_NONE && d->res.C == 0 && d->out2.C == 0,
                           "yp_conv2d: bn_partial needs a plain convolution (no bias / activation / residual / second output)");
                a.stats = d->bn_partial;
                a.stats_rows = yp_cdiv(a.M, srows);
                if (g_bn_rows_query != nullptr) { *g_bn_rows_query = a.stats_rows; return YP_OK; }
            }
            e = yp_mma8_launch(d->tile, d->dtype, of32, stats, a, nblk8, stream);
            if (e != hipSuccess) { yp_set_error("yp_conv2d: 8-wave kernel launch failed: %s", hipGetErrorString(e)); return YP_ERR_HIP; }
            return YP_OK;
        }
    }
#ifdef YP_WITH_WSK      // (probe build only: make probewsk)
    {   // wave-private split-K kernels (conv_wsk.hip), tile ids 71..73: short-M 16-bit fast-path layers
        int bm = 0, bn = 0;
        if (yp_wsk_tile_dims(d->tile, &bm, &bn)) {
            YP_REQUIRE(fast && (d->dtype == YP_F16 || d->dtype == YP_BF16) && det == nullptr && d->pre_weight == nullptr && ksplit == 1 && !a.atomic_out &&
                       d->bn_partial == nullptr && Kreal % 32 == 0,
                       "yp_conv2d: tile %d (wave-private split-K kernel) needs a plain 16-bit fast-path convolution", d->tile);
            a.tiles_n = yp_cdiv(Cout, bn);
            const int nblkw = yp_cdiv(a.M, bm) * a.tiles_n;
            // (pixel index -> (b, y, x) by exact magic division: short-M layers only)
            YP_REQUIRE((unsigned long long)(a.M + 127) * (unsigned long long)a.HoWo < (1ull << 32) && a.HoWo >= 2 && a.Wo >= 2,
                       "yp_conv2d: tile %d serves M * Ho * Wo < 2^32 (M = %d, Ho * Wo = %d)", d->tile, a.M, a.HoWo);
            a.mg_howo = (unsigned)(((1ull << 32) + a.HoWo - 1) / a.HoWo);
            a.mg_wo = (unsigned)(((1ull << 32) + a.Wo - 1) / a.Wo);
            e = yp_wsk_launch(d->tile, d->dtype, of32, a, nblkw, stream);
            if (e != hipSuccess) { yp_set_error("yp_conv2d: wave-private split-K kernel launch failed: %s", hipGetErrorString(e)); return YP_ERR_HIP; }
            return YP_OK;
        }
    }
#endif
    if (det != nullptr) {
        YP_REQUIRE(det->na > 0 && det->na <= 8 && det->no > 5 && det->na * det->no <= Cout && det->x_out != nullptr, "yp_conv2d_detect: bad detect descriptor");
        YP_REQUIRE(d->act == YP_ACT_NONE && d->res.C == 0 && d->out2.C == 0 && fast && d->pre_weight == nullptr, "yp_conv2d_detect: plain fast-path convolution required");
        a.det_na = det->na; a.det_no = det->no; a.det_invno = (65536 + det->no - 1) / det->no;
        YP_REQUIRE((long)Cout * (a.det_invno * det->no - 65536) < 65536, "yp_conv2d_detect: channel decode out of range");
        a.det_rows_total = det->rows_total; a.det_row_off = det->row_offset; a.det_stride = det->stride;
        for (int i = 0; i < 16; ++i) a.det_anchor[i] = det->anchors_px[i];
        a.det_x = det->x_out; a.det_z = det->z_out;
        switch (d->dtype) {
            case YP_F16: e = launch_cfg<YP_F16, true, true, true>(tile, a, nblk, stream); break;
            case YP_BF16: e = launch_cfg<YP_BF16, true, true, true>(tile, a, nblk, stream); break;
            default: e = launch_cfg<YP_F32, false, true, true>(tile, a, nblk, stream); break;
        }
        if (e != hipSuccess) { yp_set_error("yp_conv2d_detect: launch failed: %s", hipGetErrorString(e)); return YP_ERR_HIP; }
        return YP_OK;
    }
    // 3x3 / pad 1 / stride 1|2, single un-upsampled source, 16-bit types, Cin % 32 == 0: LDS halo-reuse kernel
    // (tile ids 10..12 force it with BN = 32/64/128; tile 0 picks BN by the channel count; ids 1..5 force the generic kernel)
    YP_REQUIRE(!q8 || fast, "yp_conv2d: 8-bit inputs need the fast addressing path (channels %% 64 == 0, tail_zero)");
    const bool halo_base = fast && d->dtype != YP_F32 && d->R == 3 && d->S == 3 && d->pad_h == 1 && d->pad_w == 1 &&
                           d->stride_h == d->stride_w && (d->stride_h == 1 || d->stride_h == 2) && d->in0.ups == 0 && ksplit == 1 && !d->atomic_accumulate &&
                           in0_bytes + (size_t)d->in0.cstride * eb + 64 < (1ull << 31);
    const bool halo_ok = halo_base && (d->in1.C == 0 || post);
    if (d->stem_x != nullptr) {          // fused stem + this 3x3 / stride-2 convolution: the stem's output lives in LDS
        const bool post3 = d->post_weight != nullptr;          // ... and the 64 -> 64 pointwise convolution behind it (out / out2 are ITS destinations)
        YP_REQUIRE(halo_base && d->stride_h == 2 && !of32 && d->in0.C == 32 && d->in1.C == 0 && d->res.C == 0 && (d->out2.C == 0 || post3) && det == nullptr &&
                   d->bn_partial == nullptr && d->pre_weight == nullptr && d->out_phase == 0 && Cout <= 64 && Cout % 8 == 0,
                   "yp_conv2d: the fused stem needs a plain 16-bit 3x3 / stride 2 / pad 1 convolution of 32 -> <= 64 channels");
        YP_REQUIRE(!post3 || (Cout == 64 && d->post_Kpad >= 64 && d->post_Kpad % 32 == 0 && d->post_Npad >= 64 && d->Npad >= 64),
                   "yp_conv2d: the pointwise stage behind the fused stem maps 64 -> 64 channels (packed [>= 64][>= 64])");
        YP_REQUIRE(d->stem_weight != nullptr && d->stem_Kpad >= 144 && d->stem_C >= 1 && d->stem_C <= 4 && d->Hi % 2 == 0 && d->Wi % 2 == 0,
                   "yp_conv2d: bad fused-stem arguments");
        YP_REQUIRE(d->tile == 0, "yp_conv2d: tile %d does not apply to the fused stem", d->tile);
        a.stem_x = d->stem_x; a.stem_wgt = (const char*)d->stem_weight; a.stem_bias = d->stem_bias; a.stem_Kpad = d->stem_Kpad; a.stem_act = d->stem_act;
        a.stem_C = d->stem_C; a.stem_H = 2 * d->Hi; a.stem_W = 2 * d->Wi;
        constexpr int TH = YP_STEM2_TH;
        a.tiles_n = 1;
        a.tiles_x = yp_cdiv(d->Wo, 16);
        a.tiles_y = yp_cdiv(d->Ho, TH);
        a.Ho = d->Ho;
        const int ntl = d->B * a.tiles_y * a.tiles_x;
        a.stats_rows = ntl;                 // (the tile count: workgroups loop over tiles)
        const int nbs = ntl < 2 * 256 ? ntl : 2 * 256;     // (two resident workgroups per CU: 216 VGPRs)
        if (post3) {
            a.post_wgt = (const char*)d->post_weight; a.post_bias = d->post_bias; a.post_Kpad = d->post_Kpad; a.post_Npad = d->post_Npad;
            a.post_act = d->post_act; a.post_N = Cout;
        }
        e = yp_stem2_dispatch(d->dtype, post3, a, nbs, stream);
        if (e != hipSuccess) { yp_set_error("yp_conv2d: fused stem launch failed: %s", hipGetErrorString(e)); return YP_ERR_HIP; }
        return YP_OK;
    }
    if (d->pre_weight != nullptr) {      // fused Bottleneck: 1x1 prologue + 3x3, hidden tensor in LDS
        const int Cc = d->in0.C;
        YP_REQUIRE(halo_ok && d->stride_h == 1 && !of32 && d->out2.C == 0, "yp_conv2d: the pointwise prologue needs a 16-bit 3x3 / stride 1 / pad 1 convolution with tail_zero buffers");
        YP_REQUIRE(Cout == Cc && (Cc == 32 || Cc == 64 || Cc == 128), "yp_conv2d: pointwise prologue: in = hidden = out channels must be 32, 64 or 128 (got %d -> %d)", Cc, Cout);
        YP_REQUIRE(d->pre_Npad >= Cc && d->pre_Kpad >= Cc && d->pre_Kpad % 32 == 0, "yp_conv2d: bad packed prologue filter %dx%d", d->pre_Npad, d->pre_Kpad);
        // tile 16: the persistent form (bneck32_persist_kernel: C = 32 with the C3 tail, shortcut = the input itself); tile 0 takes it when a
        // workgroup would walk at least two tiles
        const bool persist_ok = post && Cc == 32 && (d->res.C == 0 || (d->res.ptr == d->in0.ptr && d->res.cstride == d->in0.cstride && d->res.coff == d->in0.coff));
        const bool waves8 = d->tile >= 17 && d->tile <= 19 && !post;      // 17 / 18 / 19: the 8-wave form with BN = 32 / 64 / 128
        YP_REQUIRE(d->tile == 0 || (d->tile >= 10 && d->tile <= 12) || (d->tile == 16 && persist_ok) || waves8, "yp_conv2d: tile %d does not apply to the fused bottleneck", d->tile);
        int bn = Cc < 64 ? Cc : 64;
        if (d->tile == 10 || d->tile == 17) bn = 32; else if (d->tile == 11 || d->tile == 18) bn = 64; else if (d->tile == 12 || d->tile == 19) bn = 128;
        YP_REQUIRE(bn <= Cc, "yp_conv2d: fused bottleneck tile of %d channels > %d", bn, Cc);
        a.pre_wgt = (const char*)d->pre_weight; a.pre_bias = d->pre_bias; a.pre_Kpad = d->pre_Kpad; a.pre_act = d->pre_act;
        if (post) {
            YP_REQUIRE(Cc <= 64 && d->in1.C == Cc && d->in1.ups == 0 && d->in1.H == d->Ho && d->in1.W == d->Wo && d->out.C == 2 * Cc,
                       "yp_conv2d: fused C3 tail needs hidden channels <= 64, in1 = the other %d-channel branch at the output size, out.C = %d", Cc, 2 * Cc);
            YP_REQUIRE(d->post_Npad >= 2 * Cc && d->post_Kpad >= 2 * Cc && d->post_Kpad % 32 == 0 && (d->tile == 0 || d->tile == 16 || d->tile == (Cc == 32 ? 10 : 11)),
                       "yp_conv2d: bad packed C3-tail filter %dx%d / tile %d", d->post_Npad, d->post_Kpad, d->tile);
            bn = Cc;
            a.post_wgt = (const char*)d->post_weight; a.post_bias = d->post_bias; a.post_Kpad = d->post_Kpad; a.post_Npad = d->post_Npad;
            a.post_act = d->post_act; a.post_N = 2 * Cc;
        }
        a.tiles_n = yp_cdiv(Cout, bn);
        a.tiles_x = yp_cdiv(d->Wo, 16);
        a.tiles_y = yp_cdiv(d->Ho, 8);
        a.Ho = d->Ho;
        const int nb3 = d->B * a.tiles_y * a.tiles_x * a.tiles_n;
        if (persist_ok && (d->tile == 16 || (d->tile == 0 && nb3 >= 4 * 256))) {
            a.stats_rows = nb3;                 // (the tile count: workgroups loop over tiles)
            e = yp_bneck_dispatch(d->dtype, true, Cc, bn, post, waves8, a, nb3, stream);
        } else
        e = yp_bneck_dispatch(d->dtype, false, Cc, bn, post, waves8, a, nb3, stream);
        if (e != hipSuccess) {
            yp_set_error("yp_conv2d: fused bottleneck launch failed: %s", hipGetErrorString(e));
            return YP_ERR_HIP;
        }
        return YP_OK;
    }
    if (halo_ok && (d->tile == 0 || (d->tile >= 10 && d->tile <= 15))) {
        int bn = Cout <= 32 ? 32 : (Cout <= 64 ? 64 : 128);
        const int tsel = d->tile >= 13 ? d->tile - 3 : d->tile;
        if (tsel == 10) bn = 32; else if (tsel == 11) bn = 64; else if (tsel == 12) bn = 128;
        const int th = d->tile >= 13 ? 4 : 8;
        a.tiles_n = yp_cdiv(Cout, bn);
        a.tiles_x = yp_cdiv(d->Wo, 16);
        a.tiles_y = yp_cdiv(d->Ho, th);
        a.Ho = d->Ho;
        const int nb3 = d->B * a.tiles_y * a.tiles_x * a.tiles_n;
        if (d->bn_partial != nullptr) {      // BatchNorm statistics in the epilogue: one partial row per pixel tile (B * tiles_y * tiles_x rows)
            YP_REQUIRE(!of32 && d->bias == nullptr && d->act == YP_ACT_NONE && d->res.C == 0 && d->out2.C == 0 && d->in1.C == 0,
                       "yp_conv2d: bn_partial needs a plain convolution (no bias / activation / residual / second output)");
            a.stats = d->bn_partial;
            a.stats_rows = d->B * a.tiles_y * a.tiles_x;
            if (g_bn_rows_query != nullptr) { *g_bn_rows_query = a.stats_rows; return YP_OK; }
            e = yp_halo_dispatch(d->dtype, false, true, d->stride_h, bn, th, a, nb3, stream);
        } else e = yp_halo_dispatch(d->dtype, (d->dtype == YP_F16 || d->dtype == YP_BF16) && of32, false, d->stride_h, bn, th, a, nb3, stream);
        if (e != hipSuccess) {
            yp_set_error("yp_conv2d: halo kernel launch failed: %s", hipGetErrorString(e));
            return YP_ERR_HIP;
        }
        return YP_OK;
    }
    YP_REQUIRE(d->tile < 10 || d->tile > 15, "yp_conv2d: tile %d (3x3 halo kernel) does not apply to this convolution", d->tile);
    if (d->bn_partial != nullptr) {          // BatchNorm statistics in the epilogue (generic kernel, fast addressing, 16-bit or fp32 store)
        YP_REQUIRE(fast && !of32 && ksplit <= 1 && !a.atomic_out && d->bias == nullptr && d->act == YP_ACT_NONE && d->res.C == 0 && d->out2.C == 0,
                   "yp_conv2d: bn_partial needs the plain fast path (tail_zero, no bias / activation / residual / split / second output)");
        a.stats = d->bn_partial;
        a.stats_rows = (int)(((size_t)a.M + 63) / 64);
        if (g_bn_rows_query != nullptr) { *g_bn_rows_query = a.stats_rows; return YP_OK; }
        switch (d->dtype) {
            case YP_F16: e = launch_cfg<YP_F16, false, true, false, true>(tile, a, nblk, stream); break;
            case YP_BF16: e = launch_cfg<YP_BF16, false, true, false, true>(tile, a, nblk, stream); break;
            case YP_FP8: e = launch_cfg<YP_FP8, false, true, false, true>(tile, a, nblk, stream); break;
            case YP_FP8_BF8: e = hipErrorInvalidValue; break;
            default: e = launch_cfg<YP_F32, false, true, false, true>(tile, a, nblk, stream); break;
        }
        if (e != hipSuccess) {
            yp_set_error("yp_conv2d: launch failed: %s", hipGetErrorString(e));
            return YP_ERR_HIP;
        }
        return YP_OK;
    }
#define YP_DISPATCH(DT)                                                                                             \
    (of32 ? (fast ? launch_cfg<DT, true, true>(tile, a, nblk, stream) : launch_cfg<DT, true, false>(tile, a, nblk, stream)) \
          : (fast ? launch_cfg<DT, false, true>(tile, a, nblk, stream) : launch_cfg<DT, false, false>(tile, a, nblk, stream)))
    switch (d->dtype) {
        case YP_F16: e = YP_DISPATCH(YP_F16); break;
        case YP_BF16: e = YP_DISPATCH(YP_BF16); break;
        case YP_FP8: e = launch_cfg<YP_FP8, false, true>(tile, a, nblk, stream); break;
        case YP_FP8_BF8: e = launch_cfg<YP_FP8_BF8, false, true>(tile, a, nblk, stream); break;
        default: e = fast ? launch_cfg<YP_F32, false, true>(tile, a, nblk, stream) : launch_cfg<YP_F32, false, false>(tile, a, nblk, stream); break;
    }
#undef YP_DISPATCH
    if (e != hipSuccess) {
        yp_set_error("yp_conv2d: launch failed: %s", hipGetErrorString(e));
        return YP_ERR_HIP;
    }
    return YP_OK;
}

extern "C" int yp_conv2d(const YpConvDesc* d, void* stream) { return yp_conv2d_launch(d, nullptr, (hipStream_t)stream); }
extern "C" int yp_conv_bn_partial_rows(const YpConvDesc* d, int* rows) {
    YP_REQUIRE(d != nullptr && rows != nullptr && d->bn_partial != nullptr, "yp_conv_bn_partial_rows: a descriptor with bn_partial is required");
    *rows = -1;
    g_bn_rows_query = rows;
    const int rc = yp_conv2d_launch(d, nullptr, nullptr);
    g_bn_rows_query = nullptr;
    if (rc == YP_OK && *rows < 0) { yp_set_error("yp_conv_bn_partial_rows: this convolution does not write bn_partial"); return YP_ERR_INVALID; }
    return rc;
}
extern "C" int yp_conv2d_detect(const YpConvDesc* d, const YpDetectDesc* det, void* stream) {
    YP_REQUIRE(det != nullptr, "yp_conv2d_detect: null detect descriptor");
    return yp_conv2d_launch(d, det, (hipStream_t)stream);
}

#endif   // YP_PART != 2

#if YP_PART != 1
extern "C" int yp_stem_conv(const float* x_nchw, int B, int C, int H, int W, const void* weight, int Kpad, const float* bias, int act, YpView out,
                            int dtype, void* stream) {
    YP_REQUIRE(x_nchw && weight && out.ptr && B > 0 && C > 0 && C <= 4 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "yp_stem_conv: bad arguments");
    YP_REQUIRE(dtype == YP_F16 || dtype == YP_BF16, "yp_stem_conv: 16-bit compute types only");
    YP_REQUIRE(out.H == H / 2 && out.W == W / 2 && out.C % 16 == 0 && out.C <= 64 && out.cstride % 8 == 0 && out.coff % 8 == 0 && Kpad >= 144, "yp_stem_conv: output view / filter mismatch");
    const int tiles_x = yp_cdiv(W / 2, 64), tiles_y = yp_cdiv(H / 2, 8);
    const int nblk = B * tiles_x * tiles_y;
    hipStream_t st = (hipStream_t)stream;
#define YP_STEM(DT, FN) stem_conv_kernel<DT, FN><<<nblk, 256, 0, st>>>(x_nchw, B, C, H, W, (const char*)weight, Kpad, bias, act, (char*)out.ptr, out.cstride, out.coff, out.C, tiles_x, tiles_y)
    const int fn = out.C / 16;
    if (dtype == YP_F16) { if (fn == 1) YP_STEM(YP_F16, 1); else if (fn == 2) YP_STEM(YP_F16, 2); else if (fn == 3) YP_STEM(YP_F16, 3); else YP_STEM(YP_F16, 4); }
    else { if (fn == 1) YP_STEM(YP_BF16, 1); else if (fn == 2) YP_STEM(YP_BF16, 2); else if (fn == 3) YP_STEM(YP_BF16, 3); else YP_STEM(YP_BF16, 4); }
#undef YP_STEM
    YP_CHECK_HIP(hipGetLastError());
    return YP_OK;
}
#endif   // YP_PART != 1
