// Pieces shared by the convolution translation units (conv_igemm.hip, conv_mma8.hip): element types, kernel arguments,
// epilogue store helpers, the XCD-aware tile remap and the LDS-DMA issue helpers.
#pragma once
#include "yp_internal.h"
#include <type_traits>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

template <int DT> struct Elem;
template <> struct Elem<YP_F16> {
    using frag = f16x8;
    using scalar = _Float16;
    static constexpr int BYTES = 2, OBYTES = 2, KPL = 8, KM = 32;
    static __device__ __forceinline__ f32x4 mma(frag a, frag b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
};
template <> struct Elem<YP_BF16> {
    using frag = bf16x8;
    using scalar = __bf16;
    static constexpr int BYTES = 2, OBYTES = 2, KPL = 8, KM = 32;
    static __device__ __forceinline__ f32x4 mma(frag a, frag b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};
// 8-bit inputs (OCP fp8, one byte per element), bf16 results.  A 64-byte k tile row holds 64 elements; a lane's 16-byte fragment feeds
// TWO 16x16x32 MFMAs (its low and its high 8 bytes): filter and activation fragments are cut the same way, so the k permutation inside
// the tile cancels in the dot product and the DMA / LDS layout / swizzle of the 16-bit kernels carries over byte for byte.
//   YP_FP8:     filter e4m3 x activation e4m3 (forward)        YP_FP8_BF8: filter e4m3 x activation e5m2 (dgrad: the activation is dy)
template <> struct Elem<YP_FP8> {
    using frag = u32x4;
    using scalar = __bf16;
    static constexpr int BYTES = 1, OBYTES = 2, KPL = 16, KM = 64;
    static __device__ __forceinline__ f32x4 mma(frag a, frag b, f32x4 c) {
        const long a0 = ((long)a[1] << 32) | a[0], a1 = ((long)a[3] << 32) | a[2], b0 = ((long)b[1] << 32) | b[0], b1 = ((long)b[3] << 32) | b[2];
        c = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a0, b0, c, 0, 0, 0);
        return __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a1, b1, c, 0, 0, 0);
    }
};
template <> struct Elem<YP_FP8_BF8> {
    using frag = u32x4;
    using scalar = __bf16;
    static constexpr int BYTES = 1, OBYTES = 2, KPL = 16, KM = 64;
    static __device__ __forceinline__ f32x4 mma(frag a, frag b, f32x4 c) {
        const long a0 = ((long)a[1] << 32) | a[0], a1 = ((long)a[3] << 32) | a[2], b0 = ((long)b[1] << 32) | b[0], b1 = ((long)b[3] << 32) | b[2];
        c = __builtin_amdgcn_mfma_f32_16x16x32_fp8_bf8(a0, b0, c, 0, 0, 0);
        return __builtin_amdgcn_mfma_f32_16x16x32_fp8_bf8(a1, b1, c, 0, 0, 0);
    }
};
template <> struct Elem<YP_F32> {
    using frag = float;
    using scalar = float;
    static constexpr int BYTES = 4, OBYTES = 4, KPL = 1, KM = 4;
    static __device__ __forceinline__ f32x4 mma(frag a, frag b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
};

#define YP_PIN2(T, name) T name = a.name; asm volatile("" : "+s"(name))


struct ConvKArgs {
    const char* in0;
    const char* in1;
    const char* wgt;
    const float* bias;
    const char* res;
    char* out;
    int in0_cs, in0_co, in0_C, in0_ups, in0_H, in0_W;
    int in1_cs, in1_co, in1_ups, in1_H, in1_W;
    int res_cs, res_co, has_res;
    int out_cs, out_co;
    int out_sub, out_py, out_px;   // out_sub: `out` / `res` are the parity-(py, px) pixels of a [2 Ho][2 Wo] tensor (YpConvDesc.out_phase)
    char* out2; int out2_cs, out2_co, split;      // channels >= split go to out2 (split == Cout: unused)
    int Hi, Wi, Wo, HoWo;
    int Cin, Cout, Kreal, Kpad, Npad;
    int R, S, RS, invS, dt, dc, sh, sw, ph, pw;
    int dil_h, dil_w;              // filter dilation (wgrad-as-convolution of a strided conv)
    int in0_zs;                    // in0 is a zero-stuffed view: logical (2H x 2W), odd rows/cols are zero (dgrad of stride 2)
    int ksplit, atomic_out;        // split-K over blockIdx.y with fp32 atomicAdd epilogue
    float* split_slabs;            // ... or (non-null) a plain fp32 store of slice y's partial tile to split_slabs + y * split_stride
    long split_stride;
    unsigned in0_zoff, in1_zoff, wgt_zrow;   // FAST path: byte offsets of the 16 zero bytes behind each input / the zero filter row
    int act;
    int M, tiles_n;
    int tiles_x, tiles_y, Ho;      // 3x3 halo kernel: 8x16 output tiles
    const char* pre_wgt;           // fused Bottleneck: 1x1 prologue filter / bias
    const float* pre_bias;
    int pre_Kpad, pre_act;
    const char* post_wgt;          // fused C3 tail: 1x1 conv over cat(bottleneck output, in1) -> out
    const float* post_bias;
    int post_Kpad, post_Npad, post_act, post_N;
    unsigned post_zrow;
    // fused Detect decode (EPI_DETECT instantiations)
    int det_na, det_no, det_invno, det_rows_total, det_row_off;
    float det_stride;
    float det_anchor[16];
    float* det_x;
    float* det_z;
    float* stats;                  // STATS instantiations: per-row-block column sums [2][Cout][stats_rows] (BatchNorm statistics; a channel's
    int stats_rows;                // partials lie together: the fold that follows reads them as contiguous runs)
    const float* scale_in;         // 8-bit input types: the accumulators are multiplied by *scale_in * *scale_w (device scalars: the
    const float* scale_w;          // dequantisation scales of the activation and of the filter) before the epilogue
    unsigned mg_howo, mg_wo;       // ceil(2^32 / HoWo), ceil(2^32 / Wo) when M * HoWo < 2^32 (exact magic division), else 0
    unsigned mg_tn;                // ceil(2^32 / tiles_n) (generic kernel: workgroup -> (tile_m, tile_n)), 0: divide
    const float* stem_x;           // fused stem + 3x3 / stride-2 convolution (stem_conv2_kernel): the caller's NCHW fp32 image, the stem's paired-pixel
    const char* stem_wgt;          // filter / bias, the image's dims and channel count
    const float* stem_bias;
    int stem_Kpad, stem_act, stem_C, stem_H, stem_W;
    int probe;                     // -DYP_PROBE8 builds of conv_mma8.hip: elimination experiments (1 no MFMA, 2 no steady-state DMA, 4 L2-resident pixels)
};

// sigmoid as v_mul, v_exp_f32, v_add, v_rcp_f32 (rel. error ~1e-7); a plain 1/(1+expf(-x)) is ~25 instructions
__device__ __forceinline__ float yp_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f)); }

__device__ __forceinline__ float yp_silu(float x) {
    // x * sigmoid(x) as v_mul, v_exp_f32, v_add, v_rcp_f32, v_mul (rel. error ~1e-7).  NOT __frcp_rn / a plain divide: those
    // expand to the 12-instruction IEEE division sequence, and every output element of every convolution passes through here.
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
}

// Epilogue tail shared by the convolution kernels: (+ residual) -> convert -> store CW consecutive
// channels [nc, nc+CW) of output pixel m as 8/16-byte vectors, into `out` or (nc >= split) `out2`.
// RES: 1 = the launch adds a residual, 0 = it does not, -1 = test a.has_res here.  Callers hoist the test around their store loops
// (YP_RES_DISPATCH): with the residual load under a branch INSIDE the loop the compiler must place `s_waitcnt vmcnt(0)` at the join in front of
// every store -- loads and stores share the counter, so each store waited for the previous one's write acknowledgement (~700 clocks): the
// epilogue of the 256 x 256 kernels was 19-32 k clocks with the stores and 2.6-4.8 k without (round 5, tools/probe/mma8_timeline.py).
// pixel (b, y, x) of the Ho x Wo launch grid -> (b, 2y + py, 2x + px) of the tensor behind `out` / `res` (a.out_sub launches)
__device__ __forceinline__ int yp_out_pixel(const ConvKArgs& a, int m) {
    if (a.out_sub) {
        const int b = m / a.HoWo, rem = m - b * a.HoWo, y = rem / a.Wo, x = rem - y * a.Wo;
        m = (b * 2 * (a.HoWo / a.Wo) + 2 * y + a.out_py) * (2 * a.Wo) + 2 * x + a.out_px;
    }
    return m;
}

// The residual's CW channels [nc, nc+CW) of output pixel m, as fetched (16-bit types: 8 / 16 bytes; fp32: CW floats): issued for ALL chunks of a
// pixel before the first of them is used (yp_epilogue_pixel, RES = 1), so that a wave waits once per pixel, not once per chunk.
template <int DT, int CW>
struct YpResRaw { u32x4 q[(DT == YP_F32 ? CW * 4 : CW * 2) / 16 > 0 ? (DT == YP_F32 ? CW * 4 : CW * 2) / 16 : 1]; };
template <int DT, int CW>
__device__ __forceinline__ YpResRaw<DT, CW> yp_res_fetch(const char* row_rp, int nc) {
    constexpr int EB = Elem<DT>::OBYTES;
    const char* rp = row_rp + (size_t)nc * EB;
    YpResRaw<DT, CW> r;
    if constexpr (CW * EB >= 16) {
#pragma unroll
        for (int i = 0; i < CW * EB / 16; ++i) r.q[i] = *reinterpret_cast<const u32x4*>(rp + 16 * i);
    } else {
        const u32x2 lo = *reinterpret_cast<const u32x2*>(rp);
        r.q[0] = u32x4{lo[0], lo[1], 0u, 0u};
    }
    return r;
}
template <int DT, int CW>
__device__ __forceinline__ void yp_res_add(const YpResRaw<DT, CW>& r, float (&v)[CW]) {
    if constexpr (DT == YP_F32) {
        const float* e = reinterpret_cast<const float*>(&r);
#pragma unroll
        for (int j = 0; j < CW; ++j) v[j] += e[j];
    } else {
        using sc = typename Elem<DT>::scalar;
        const sc* e = reinterpret_cast<const sc*>(&r);
#pragma unroll
        for (int j = 0; j < CW; ++j) v[j] += (float)e[j];
    }
}

// Row pointers of one output pixel: computed ONCE per pixel (pixel remap of out_sub launches, 64-bit multiplies), a chunk then adds its channel
// offset and picks its destination with two selects -- branch-free.  (The per-chunk form evaluated both destinations' index arithmetic under
// exec masks for every chunk: ~190 instructions per 16-byte store.)
struct YpOutRow { char* p1; char* p2; const char* rp; };
template <int DT, bool OUT_F32>
__device__ __forceinline__ YpOutRow yp_out_row(const ConvKArgs& a, int m) {
    constexpr int OB = (OUT_F32 || DT == YP_F32) ? 4 : 2, EB = Elem<DT>::OBYTES;
    const size_t mo = (size_t)yp_out_pixel(a, m);
    YpOutRow r;
    r.p1 = a.out + (mo * a.out_cs + a.out_co) * OB;
    r.p2 = a.out2 + ((long)(mo * a.out2_cs + a.out2_co) - (long)a.split) * OB;      // (+ nc * OB for nc >= split; never dereferenced when out2 is unused)
    r.rp = a.res + (mo * a.res_cs + a.res_co) * EB;
    return r;
}

template <int DT, bool OUT_F32, int CW, int RES = -1>
__device__ __forceinline__ void yp_store_chunk_at(const ConvKArgs& a, const YpOutRow& row, int nc, float (&v)[CW]) {
    using E = Elem<DT>;
    constexpr int EB = E::OBYTES;        // (results and the residual they add are 16-bit for the 8-bit input types)
    if (RES == 1 || (RES < 0 && a.has_res)) {
        const char* rp = row.rp + (size_t)nc * EB;
        if constexpr (DT == YP_F32) {
#pragma unroll
            for (int j = 0; j < CW; j += 4) {
                const f32x4 r4 = *reinterpret_cast<const f32x4*>(rp + j * 4);
                v[j] += r4[0]; v[j + 1] += r4[1]; v[j + 2] += r4[2]; v[j + 3] += r4[3];
            }
        } else {
            using sc = typename E::scalar;
            if constexpr (CW == 8) {
                const u32x4 raw = *reinterpret_cast<const u32x4*>(rp);
                const sc* e = reinterpret_cast<const sc*>(&raw);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] += (float)e[j];
            } else {
                const u32x2 raw = *reinterpret_cast<const u32x2*>(rp);
                const sc* e = reinterpret_cast<const sc*>(&raw);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] += (float)e[j];
            }
        }
    }
    if constexpr (OUT_F32 || DT == YP_F32) {
        char* op = (nc >= a.split ? row.p2 : row.p1) + (size_t)nc * 4;
#pragma unroll
        for (int j = 0; j < CW; j += 4) *reinterpret_cast<f32x4*>(op + j * 4) = f32x4{v[j], v[j + 1], v[j + 2], v[j + 3]};
    } else {
        using sc = typename E::scalar;
        char* op = (nc >= a.split ? row.p2 : row.p1) + (size_t)nc * 2;
        if constexpr (CW == 8) {
            u32x4 pk;
            sc* e = reinterpret_cast<sc*>(&pk);
#pragma unroll
            for (int j = 0; j < 8; ++j) e[j] = (sc)v[j];
            *reinterpret_cast<u32x4*>(op) = pk;
        } else {
            u32x2 pk;
            sc* e = reinterpret_cast<sc*>(&pk);
#pragma unroll
            for (int j = 0; j < 4; ++j) e[j] = (sc)v[j];
            *reinterpret_cast<u32x2*>(op) = pk;
        }
    }
}

template <int DT, bool OUT_F32, int CW, int RES = -1>
__device__ __forceinline__ void yp_store_chunk(const ConvKArgs& a, int m, int nc, float (&v)[CW]) {
    yp_store_chunk_at<DT, OUT_F32, CW, RES>(a, yp_out_row<DT, OUT_F32>(a, m), nc, v);
}

// "These registers have ARRIVED" on every path that follows: a common use of values fetched with global loads (the bias), placed in front of
// a store loop whose bodies run under lane masks (`if (m >= M) continue`).  Without it the compiler re-waits for the load at the first use inside
// every masked block -- `s_waitcnt vmcnt(0)`, which also waits for every store issued meanwhile: the stores of a wave ran one write round trip
// (~700 clocks) apart in every convolution kernel until round 5.
template <int N>
__device__ __forceinline__ void yp_pin_arrived(float (&v)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("" : "+v"(v[i]));
}

// body(std::integral_constant<int, RES>) with RES = 1 / 0 by the launch's has_res (see yp_store_chunk)
#define YP_RES_DISPATCH(a_, body_) do { if ((a_).has_res) body_(std::integral_constant<int, 1>{}); else body_(std::integral_constant<int, 0>{}); } while (0)

// bias -> activation -> yp_store_chunk for the LPG consecutive channels [nb, nb+LPG) a lane owns at pixel m;
// acc(j) returns the accumulator of lane-local channel j.
// Workgroup b runs on XCD b % 8 (round-robin dispatch) and every XCD has its own L2: give each XCD a contiguous run of
// logical tile ids so that tiles which share input pixels / filter rows (neighbouring ids) hit the same L2.
__device__ __forceinline__ int yp_xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// The persistent kernels' tile walk.  Workgroup b (on XCD b % 8) visits first, first + step, ... < end inside ITS XCD's contiguous run of tile
// ids: the tiles in flight on one XCD at any moment are ~nwg / 8 CONSECUTIVE tiles (several neighbouring tile rows of one image), so the halo
// lines two neighbouring tiles share are fetched into that XCD's L2 once.  (The plain walk b, b + nwg, ... puts neighbouring tiles on eight
// different L2s: the fused stem stage fetched 102.8 MB for a 39.3 MB image, profiles/r05_layers_traffic.txt.)
__device__ __forceinline__ void yp_xcd_walk(int bid, int nwg, int ntiles, int& first, int& end, int& step) {
    if ((nwg & 7) != 0) { first = bid; end = ntiles; step = nwg; return; }
    const int q = ntiles >> 3, r = ntiles & 7;
    const int xcd = bid & 7, j = bid >> 3;
    const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    first = start + j;
    end = start + q + (xcd < r ? 1 : 0);
    step = nwg >> 3;
}

// One LDS-DMA instruction: 64 lanes x 16 B -> 1 KiB at LDS byte address `lds_dst` (wave-uniform, in
// an SGPR), lane l landing at lds_dst + 16*l.  Issued through inline asm so that the compiler's
// LDS-DMA alias tracking does not put s_waitcnt vmcnt(0) in front of the fragment reads; the
// pipeline below counts these loads itself (they are the only VMEM operations inside the k loop).
__device__ __forceinline__ void yp_glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

// Same, with the source given as a wave-uniform 64-bit base (SGPR pair) + per-lane 32-bit byte offset.
__device__ __forceinline__ void yp_glds16_s(const void* sbase, unsigned voff, unsigned lds_dst) {
    // (the scalar operands go through readfirstlane: a no-op where the compiler already holds them in SGPRs, and the guarantee the "s"
    // constraints need where register pressure made it keep a uniform value in a VGPR -- "illegal VGPR to SGPR copy" otherwise)
    const unsigned long long b = (unsigned long long)sbase;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    const void* sb = (const void*)(((unsigned long long)hi << 32) | lo);
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_dst);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sb), "s"(dst)
                 : "memory");
}

// 8-wave 32x32x16 kernels (conv_mma8.hip), tile ids 41..44
bool yp_mma8_tile_dims(int tile, int* bp, int* bc, int* stat_rows_px);
hipError_t yp_mma8_launch(int tile, int dtype, bool out_f32, bool stats, const ConvKArgs& a, int nblk, hipStream_t st);

// wave-private split-K kernels for the short-M layers (conv_wsk.hip), tile ids 71..73
bool yp_wsk_tile_dims(int tile, int* bm, int* bn);
hipError_t yp_wsk_launch(int tile, int dtype, bool out_f32, const ConvKArgs& a, int nblk, hipStream_t st);

// x of lane ^ O.  O = 1, 2, 4, 8 stay inside a row of 16 lanes: DPP moves on the VALU (quad_perm, row_shl / row_shr under bank masks,
// row_ror) instead of ds_bpermute round trips through the LDS crossbar (what __shfl_xor compiles to); larger O: __shfl_xor.
// Used by the BatchNorm-statistics epilogues (their butterflies over the pixel lanes were two thirds of the epilogue's cost).
template <int O>
__device__ __forceinline__ float yp_xor_lane(float x) {
    const int v = __float_as_int(x);
    if constexpr (O == 1) return __int_as_float(__builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true));        // quad_perm [1,0,3,2]
    else if constexpr (O == 2) return __int_as_float(__builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    else if constexpr (O == 4) {
        int t = __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xA, false);      // row_shr:4 into banks 1, 3 (lanes 4-7, 12-15 <- 0-3, 8-11)
        t = __builtin_amdgcn_update_dpp(t, v, 0x104, 0xF, 0x5, false);          // row_shl:4 into banks 0, 2 (lanes 0-3, 8-11 <- 4-7, 12-15)
        return __int_as_float(t);
    } else if constexpr (O == 8) return __int_as_float(__builtin_amdgcn_update_dpp(0, v, 0x128, 0xF, 0xF, true));   // row_ror:8
    else return __shfl_xor(x, O, 64);
}
