/*
 * yolopoint_hip.h — C ABI of libyolopoint_hip.so (MI355X / gfx950 only).
 *
 * The reference (UniBwTAS/YOLOPoint) has no FFI layer: its hot path is Python calling
 * stock ATen/torchvision ops.  Each entry point below replaces one of those call sites;
 * the citation after "replaces:" is the reference file:line (relative to /root/reference).
 *
 * Conventions
 *   - every function returns 0 (YP_OK) or a negative YP_ERR_* code; yp_last_error() gives
 *     a thread-local human-readable message.  Nothing throws, nothing aborts.
 *   - all pointers are DEVICE pointers unless the parameter name ends in _host.
 *   - no hidden allocation on the data path: the caller owns every buffer; kernels that
 *     need scratch take a workspace sized by the matching *_workspace_bytes() query.
 *   - every launch function takes the hipStream_t to enqueue on (passed as void*).
 *   - activations are NHWC ("pixels x channels"); a YpView describes a channel slice of
 *     an NHWC buffer so that torch.cat (reference models/common.py:135,229 and
 *     models/YOLOPoint.py:216,232,236,239,242) and nn.Upsample (models/YOLOPoint.py:192)
 *     never materialise: producers write into a slice, consumers read (y>>1,x>>1).
 */
#ifndef YOLOPOINT_HIP_H
#define YOLOPOINT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    YP_OK = 0,
    YP_ERR_INVALID = -1,        /* bad argument / unsupported shape            */
    YP_ERR_HIP = -2,            /* a HIP runtime call failed                    */
    YP_ERR_UNSUPPORTED = -3,    /* valid request the library cannot serve       */
    YP_ERR_WORKSPACE = -4,      /* workspace too small                          */
    YP_ERR_OVERFLOW = -5        /* an output list was truncated at max_out      */
};

/* arithmetic type of activations / packed weights */
enum { YP_F16 = 0, YP_BF16 = 1, YP_F32 = 2,
       /* 8-bit OCP floating point INPUTS of a convolution (one byte per element; results are bf16, the accumulation fp32):
        * YP_FP8 = filter e4m3 x activation e4m3 (forward), YP_FP8_BF8 = filter e4m3 x activation e5m2 (dgrad: the "activation" is dy).
        * Per-tensor scales: real value = stored value * scale (YpConvDesc.scale_in / scale_w, yp_quantize_fp8). */
       YP_FP8 = 3, YP_FP8_BF8 = 4 };
/* fused activation */
enum { YP_ACT_NONE = 0, YP_ACT_SILU = 1 };

const char* yp_last_error(void);
int yp_version(void);
/* number of HIP devices visible, 0 when none (never fails, usable on a CPU-only host) */
int yp_device_count(void);

/* ------------------------------------------------------------------------------------
 * NHWC channel-slice view.
 *   ptr      base of the NHWC buffer (pixel 0, channel 0)
 *   H, W     physical spatial size of the buffer
 *   cstride  channels per pixel in the buffer (multiple of 8)
 *   coff     first channel of this view (multiple of 8)
 *   C        channels in this view (multiple of 8; pad channels hold zeros)
 *   ups      0 or 1: when read as a conv INPUT the logical size is (H<<ups, W<<ups) and
 *            pixel (y,x) reads physical (y>>ups, x>>ups)  == nn.Upsample(2,'nearest')
 * ---------------------------------------------------------------------------------- */
typedef struct YpView {
    void* ptr;
    int32_t H, W;
    int32_t cstride, coff, C;
    int32_t ups;
} YpView;

/* ------------------------------------------------------------------------------------
 * Fused implicit-GEMM convolution (MFMA, no im2col buffer).
 *   out = [res +] act( conv(cat(in0,in1), W) + bias )
 * replaces: models/common.py:22-34 (Conv: Conv2d+BatchNorm2d+SiLU, BN folded by
 *           utils/torch_utils_yolo.py:194-214), :88-89 (Bottleneck residual add),
 *           :135/:229 + models/YOLOPoint.py:216,232,236,239,242 (torch.cat),
 *           models/YOLOPoint.py:186,195 (ConvDet/ConvDesc), models/yolo.py:46,51 (Detect.m)
 *   weight   packed [Npad][Kpad] row-major in `dtype`, k = (r*S + s)*Cin + c (yp_conv_pack_*)
 *   bias     fp32 [Npad] or NULL
 *   Cin      = in0.C + in1.C (in1.C == 0 for a single source)
 *   Cout     = out.C (multiple of 8; rows >= real Cout are zero in `weight`)
 *   out_f32  write fp32 instead of `dtype` (head outputs: logits / descriptors)
 *   out2     optional second destination: output channels [out.C, out.C + out2.C) go to `out2`
 *            (two convolutions of the same input fused into one launch, e.g. C3.cv1 + C3.cv2,
 *            reference models/common.py:135); out2.C == 0 when unused
 * ---------------------------------------------------------------------------------- */
typedef struct YpConvDesc {
    YpView in0, in1, out, res;       /* res.C == 0: no residual                       */
    YpView out2;                     /* out2.C == 0: single destination               */
    const void* weight;
    const float* bias;
    int32_t dtype;                   /* YP_F16 | YP_BF16 | YP_F32                     */
    int32_t out_f32;
    int32_t B;
    int32_t Hi, Wi;                  /* logical input size                             */
    int32_t Ho, Wo;
    int32_t R, S;                    /* filter taps                                    */
    int32_t stride_h, stride_w, pad_h, pad_w;
    int32_t Kpad, Npad;              /* packed weight dims                             */
    int32_t act;                     /* YP_ACT_*                                       */
    int32_t tile;                    /* 0 = auto, else forced tile config id (testing) */
    int32_t dil_h, dil_w;            /* filter dilation, 0 = 1 (generic kernel only)   */
    int32_t in0_zero_stuffed;        /* in0 (ups = 1) is a zero-stuffed tensor: only even logical rows/cols
                                      * carry data (gradient of a stride-2 convolution w.r.t. its input)       */
    int32_t ksplit;                  /* > 1: split the reduction over that many workgroups per tile; fp32
                                      * atomicAdd into `out` (must be zero-initialised, out_f32, no bias/act)  */
    int32_t atomic_accumulate;       /* 1: accumulate into `out` with fp32 atomics even when ksplit <= 1       */
    int32_t tail_zero;               /* 1: each input buffer is followed by >= (cstride + 64) zero elements and
                                      * `weight` by one zero row [Kpad] -> the kernels may use their fast 32-bit
                                      * DMA addressing (padding taps are fetched from those zeros)              */
    /* Pointwise prologue (fused Bottleneck, reference models/common.py:79-89): when pre_weight != NULL the 3x3 /
     * stride-1 convolution reads act(conv1x1(in0, pre_weight) + pre_bias) instead of in0; the hidden tensor lives
     * only in LDS.  Requires in0.C == out.C == hidden channels in {32, 64, 128}, a 16-bit dtype, tail_zero, no
     * in1 / out2 / out_f32; `res` (normally in0 itself) is the shortcut.  tile: 0 auto, 10/11/12 = 32/64/128
     * output channels per workgroup. */
    const void* pre_weight;          /* packed [pre_Npad][pre_Kpad] 1x1 filter                                */
    const float* pre_bias;
    int32_t pre_Kpad, pre_Npad;
    int32_t pre_act;
    int32_t post_act;
    /* C3 tail (reference models/common.py:135 `cv3(cat(m(cv1(x)), cv2(x)))`) fused behind the fused Bottleneck: when post_weight !=
     * NULL, `in1` is the cv2 branch (in0.C channels at the output size), `out` the C3 output (2 * in0.C channels) and the kernel
     * writes out = post_act(conv1x1(cat(bottleneck output, in1), post_weight) + post_bias); the bottleneck output stays in LDS.
     * Requires pre_weight and in0.C <= 64. */
    const void* post_weight;         /* packed [post_Npad][post_Kpad] 1x1 filter over 2 * in0.C input channels */
    const float* post_bias;
    int32_t post_Kpad, post_Npad;
    /* BatchNorm statistics in the epilogue (training forward of Conv = conv -> BN -> SiLU, reference models/common.py:22-34): when
     * non-NULL the generic kernel also writes, per block of 64 output pixels rb, the column sums of the raw output and of its square:
     * bn_partial[(0*C + c)*R + rb] and [(1*C + c)*R + rb], C = out.C, R = ceil(B*Ho*Wo / 64) row blocks (a channel's partials lie
     * together: yp_bn_finalize reads them as contiguous runs).
     * Needs tail_zero, a 16-bit or fp32 store of the same dtype, no bias / activation / residual / out2 / ksplit.  The 3x3 halo kernels
     * (tile ids 10..15) write one row per pixel tile instead: R = B * ceil(Ho/8 | Ho/4) * ceil(Wo/16); the 8-wave kernels one per 64 or 128
     * pixels.  Size the buffer for B * ceil(Ho/4) * ceil(Wo/16) rows (>= every variant's count); R of the variant a descriptor selects is
     * yp_conv_bn_partial_rows -- the fold must be given exactly that count (it is the stride of the layout). */
    float* bn_partial;
    /* Deterministic split-K (ksplit > 1): when non-NULL, k slice y writes its partial output -- laid out like `out` -- to
     * split_slabs + y * split_stride floats with plain stores, and the caller sums the slices in order (yp_sum_slabs); NULL: fp32 atomics
     * into the zero-initialised `out` (order of arrival). */
    float* split_slabs;
    int64_t split_stride;
    /* dtype YP_FP8 / YP_FP8_BF8: device scalars, the dequantisation scales of in0 / in1 (one common scale) and of the packed filter; the
     * accumulators are multiplied by *scale_in * *scale_w before bias / BatchNorm statistics / the 16-bit store.  Channels % 64 == 0,
     * Kpad % 64 == 0 (bytes = elements), tail_zero, no out_f32 / split / Detect / pointwise prologue; generic kernel only. */
    const float* scale_in;
    const float* scale_w;
    /* out_phase = 1 + 2 py + px: `out` (and `res`) are the pixels of parity (py, px) of a [2 out.H][2 out.W] tensor -- output pixel (b, y, x)
     * lands at (b, 2y + py, 2x + px); out.ptr is that tensor's base.  0: dense.  The dgrad of a 3x3 / stride-2 convolution is four such
     * launches with yp_pack_weight modes 4..7 ((1 + py) x (1 + px) taps each, pad 0, out dims = dout's) instead of one 3x3 launch over a
     * zero-stuffed tensor: a quarter of the multiply-adds.  No out2 / Detect / bn_partial / ksplit / pointwise prologue. */
    int32_t out_phase;
    int32_t reserved_;
    /* Fused stem (reference models/YOLOPoint.py:156-157, `Conv1` then `Conv2`): when stem_x != NULL this 3x3 / stride-2 / pad-1 convolution over 32
     * channels reads act(stem(x)) -- the 6x6 / stride-2 / pad-2 stem of yp_stem_conv applied to the caller's NCHW fp32 image -- instead of in0: the
     * stem's output (the largest activation of the network) lives in LDS only.  in0 then only DESCRIBES that tensor (H, W, C = 32, any non-null
     * ptr); stem_weight / stem_bias / stem_Kpad / stem_act are yp_stem_conv's arguments, stem_C the image channels (<= 4).  16-bit types, out.C <= 64,
     * no residual / second output / Detect / bn_partial.
     * With stem_weight AND post_weight set, post_weight / post_bias / post_Kpad / post_Npad / post_act describe a 64 -> 64 POINTWISE convolution
     * behind this one (reference models/common.py:133-135: `cv1` and `cv2` of the C3 block that follows Conv2, as one filter): the launch then
     * writes out (+ out2: the split destination of the pointwise stage, out.C + out2.C = 64) and this convolution's own 64-channel output
     * is never materialised either (bias / act stay this convolution's, applied before the pointwise stage). */
    const float* stem_x;
    const void* stem_weight;
    const float* stem_bias;
    int32_t stem_Kpad, stem_act, stem_C, stem_reserved_;
} YpConvDesc;

/* Number of bn_partial rows the launch described by `d` (its tile id included) writes; rows are batch-major, so with `groups` statistics
 * groups the rows of group g are [g*rows/groups, (g+1)*rows/groups) -- for the generic kernel only when B/groups*Ho*Wo is a multiple of 64. */
int yp_conv_bn_partial_rows(const YpConvDesc* d, int* rows);

/* Weight gradient of the stem Conv(3, c1, k=6, s=2, p=2) (reference models/YOLOPoint.py:156; autograd's conv2d weight gradient in
 * loss.backward(), train.py:245) from the packed image x [B][H][W][4] (16-bit, channel 3 zero) and dy [B][H/2][W/2][C <= 80]:
 * dw[ci][r][s][co] fp32, ci < 4, C = dy.C columns.  slabs: yp_stem_wgrad_slabs(B, H, W) * 144 * dy.C floats of scratch (one partial
 * sum per workgroup, folded in order: bit-reproducible). */
int yp_stem_wgrad_slabs(int B, int H, int W);
int yp_stem_wgrad(YpView x, YpView dy, int dtype, int B, float* slabs, float* dw, void* stream);

/* dst[i] = slabs[0][i] + slabs[1][i] + ... + slabs[n_slabs-1][i], summed in that order (the fold of a deterministic split-K). */
int yp_sum_slabs(const float* slabs, float* dst, size_t elems, int n_slabs, void* stream);
/* The same total as a fixed two-level tree (for hundreds of small slabs): every `group` consecutive slabs are summed in order INTO the
 * group's first slab (the slabs are scratch and are overwritten), then the group sums in order into dst.  n_slabs % group == 0. */
int yp_sum_slabs_tree(float* slabs, float* dst, size_t elems, int n_slabs, int group, void* stream);

int yp_conv2d(const YpConvDesc* d, void* stream);

/* Detect-head convolution with the decode fused into its epilogue: the 1x1 conv of one detection level writes
 * the permuted raw tensor x_out [B,na,ny,nx,no] and (eval mode) its rows of the decoded prediction
 * z_out [B,rows_total,no] directly; `d->out` only supplies the level's geometry (ptr may be NULL, out_f32 = 1,
 * no activation / residual / out2).  replaces: models/yolo.py:51-68 in one launch. */
typedef struct YpDetectDesc {
    int32_t na, no;
    float stride;
    float anchors_px[16];           /* na x (w, h) in pixels */
    float* x_out;
    float* z_out;                   /* NULL in train mode */
    int32_t rows_total, row_offset;
} YpDetectDesc;
int yp_conv2d_detect(const YpConvDesc* d, const YpDetectDesc* det, void* stream);
/* Stem convolution (k=6, s=2, p=2, <= 4 input channels, Cout in {16,32,48,64}) straight from an NCHW fp32 image:
 * out = act(conv(x) + bias) as NHWC `dtype` (f16 / bf16).  `weight` is the packed filter of the 16-bit stem
 * ([Cout][6][3][8] rows of pitch Kpad, see yolopoint_amd/plan.py).  Replaces yp_pack_input + yp_conv2d for
 * reference models/YOLOPoint.py:156,200. */
int yp_stem_conv(const float* x_nchw, int B, int C, int H, int W, const void* weight, int Kpad, const float* bias, int act,
                 YpView out, int dtype, void* stream);
/* K padding granule the packer must use for `dtype` */
int yp_conv_kpad(int K, int dtype);

/* ------------------------------------------------------------------------------------
 * Layout / elementwise kernels
 * ---------------------------------------------------------------------------------- */
/* NCHW fp32 image -> NHWC `dtype`, channels zero-padded to out.cstride.
 * replaces: the implicit layout of `model(inp)` input, demo.py:129-135 / train.py:208 */
int yp_pack_input(const float* x_nchw, int B, int C, int H, int W, YpView out, int dtype, void* stream);
/* NHWC view (`dtype` or fp32 when src_f32) -> dense NCHW fp32 [B,C,H,W] */
int yp_unpack_nchw(YpView in, int src_dtype, int B, int C, float* out_nchw, void* stream);
/* SPPF pyramid: three chained MaxPool2d(5,1,2) of view `x`, written to y1,y2,y3 (same dims).
 * replaces: models/common.py:220-229 */
int yp_sppf_pool(YpView x, YpView y1, YpView y2, YpView y3, int B, int dtype, void* stream);
/* MaxPool2d(kernel 2, stride 2): y[h,w] = max of x[2h..2h+1, 2w..2w+1].  replaces: models/YOLOPoint.py:289,311 (YOLOPointv52) */
int yp_maxpool2(YpView x, YpView y, int B, int dtype, void* stream);
/* per-pixel channel L2 normalisation of an fp32 NHWC view, no epsilon, in place or out of place.
 * replaces: models/YOLOPoint.py:219-220 */
int yp_l2norm_f32(YpView in, YpView out, int B, int C, void* stream);
/* Detect head decode of one level.
 *   raw      fp32 NHWC view [B,ny,nx, na*no (+pad)], channel = a*no + o
 *   x_out    fp32 [B,na,ny,nx,no]   (the permuted raw tensor the reference returns)
 *   z_out    fp32 [B,rows_total,no] (decoded, written at row_offset) or NULL in train mode
 *   anchors_px  host array na*2: anchor (w,h) in pixels (= anchors*stride)
 * replaces: models/yolo.py:49-70 */
int yp_detect_decode(YpView raw, int B, int na, int no, float stride, const float* anchors_px_host,
                     float* x_out, float* z_out, int rows_total, int row_offset, void* stream);

/* ------------------------------------------------------------------------------------
 * Training path: batch-statistics BatchNorm + SiLU forward / backward and the small backward
 * pieces of the graph.  The convolution gradients reuse yp_conv2d: dgrad = convolution with the
 * flipped, channel-transposed filter (stride 2 through in0_zero_stuffed); wgrad = convolution of
 * the yp_to_chwb copies (batch <-> channel transposed) with dilation + ksplit.
 * replaces: nn.BatchNorm2d(eps=1e-3, momentum=0.03) train mode + nn.SiLU (models/common.py:18-29)
 *           and their autograd backward (train.py:245).
 * ---------------------------------------------------------------------------------- */
size_t yp_bn_workspace_bytes(int B, int H, int W, int C);
/* batch mean / 1/sqrt(biased var + eps) of an NHWC view over its B*H*W rows; updates running stats
 * (momentum, unbiased variance) in place when the pointers are non-NULL */
int yp_bn_stats(YpView raw, int dtype, int B, float eps, float momentum, float* mean, float* invstd,
                float* running_mean, float* running_var, void* workspace, size_t workspace_bytes, void* stream);

/* mean / invstd (+ running statistics) from the per-row-block partial sums a convolution wrote (YpConvDesc.bn_partial): the second half
 * of yp_bn_stats without its reduction pass.  partial [2][C][rows]; M = the number of pixels the sums cover. */
int yp_bn_finalize(const float* partial, int rows, int C, double M, float eps, float momentum, float* mean, float* invstd, float* running_mean,
                   float* running_var, void* stream);
/* out = act(gamma*(raw-mean)*invstd + beta) [+ res] */
int yp_bn_act_apply(YpView raw, YpView out, YpView res, int dtype, int B, const float* mean, const float* invstd,
                    const float* gamma, const float* beta, int act, void* stream);
/* backward of yp_bn_act_apply w.r.t. raw (dx), gamma, beta.  workspace >= yp_bn_workspace_bytes + 8*C bytes */
int yp_bn_act_bwd(YpView raw, YpView dy, YpView dx, int dtype, int B, const float* mean, const float* invstd,
                  const float* gamma, const float* beta, int act, float* dgamma, float* dbeta, int accumulate_param_grads,
                  void* workspace, size_t workspace_bytes, void* stream);

/* The same four passes with `groups` statistics groups: the B samples are `groups` consecutive sets of B/groups samples, each normalised with
 * its own batch statistics -- what the reference computes when it calls the module once per set (src/train.py:208,220: model(img), then
 * model(img_warp)), in ONE launch per pass.  mean / invstd: [groups][C]; the running statistics take one momentum update per group, in group
 * order; dgamma / dbeta sum over the groups; yp_bn_act_bwd_grouped needs workspace >= yp_bn_workspace_bytes + 8*groups*C bytes.
 * yp_bn_finalize_grouped: `rows` partial rows and M pixels in total, rows/groups and M/groups per group.  groups <= 8, B % groups == 0. */
int yp_bn_stats_grouped(YpView raw, int dtype, int B, int groups, float eps, float momentum, float* mean, float* invstd, float* running_mean,
                        float* running_var, void* workspace, size_t workspace_bytes, void* stream);
int yp_bn_finalize_grouped(const float* partial, int rows, int groups, int C, double M, float eps, float momentum, float* mean, float* invstd,
                           float* running_mean, float* running_var, void* stream);
int yp_bn_act_apply_grouped(YpView raw, YpView out, YpView res, int dtype, int B, int groups, const float* mean, const float* invstd,
                            const float* gamma, const float* beta, int act, void* stream);
int yp_bn_act_bwd_grouped(YpView raw, YpView dy, YpView dx, int dtype, int B, int groups, const float* mean, const float* invstd,
                          const float* gamma, const float* beta, int act, float* dgamma, float* dbeta, int accumulate_param_grads,
                          void* workspace, size_t workspace_bytes, void* stream);

/* One Adam step over a flat fp32 range (all parameters of the model laid out back to back, their gradients / moments likewise):
 * torch.optim.Adam's update rule -- the reference's optimizer (src/train.py:88 Adam(lr), :252 optimizer.step()) -- amsgrad / maximize off;
 * step = 1 for the first update (bias corrections 1 - beta^step, computed in double on the host).  n % 4 == 0, 16-byte aligned arrays. */
int yp_adam_step(float* p, const float* g, float* m, float* v, size_t n, double lr, double beta1, double beta2, double eps, double weight_decay, int step,
                 void* stream);

/* fp8 training: the same two passes also writing a 1-byte twin of their result (q8: a view with the shape of `raw`; e4m3 of the forward
 * output = the next Conv's activation; e5m2 of dx = the dgrad's output gradient), quantised with *q_scale, max|result| recorded into
 * q_amax (256 floats, see yp_quantize_fp8) -- the separate quantisation pass disappears.  q8.ptr == NULL: exactly the functions above. */
int yp_bn_act_apply_grouped_q8(YpView raw, YpView out, YpView res, int dtype, int B, int groups, const float* mean, const float* invstd,
                               const float* gamma, const float* beta, int act, YpView q8, const float* q_scale, float* q_amax, void* stream);
int yp_bn_act_bwd_grouped_q8(YpView raw, YpView dy, YpView dx, int dtype, int B, int groups, const float* mean, const float* invstd,
                             const float* gamma, const float* beta, int act, float* dgamma, float* dbeta, int accumulate_param_grads,
                             void* workspace, size_t workspace_bytes, YpView q8, const float* q_scale, float* q_amax, void* stream);

/* out (+)= 2x2 block sums of `in` (backward of nn.Upsample(2,'nearest'), models/YOLOPoint.py:192) */
int yp_ups2_bwd(YpView in, YpView out, int dtype, int B, int accumulate, void* stream);
/* dst (+)= src (gradient fan-in) */
int yp_add_views(YpView src, YpView dst, int dtype, int B, int accumulate, void* stream);
/* backward of MaxPool2d(5,1,2) (models/common.py:220): dx (+)= dy routed to each window's first maximum */
size_t yp_maxpool5_bwd_workspace_bytes(int B, int H, int W, int C);
int yp_maxpool5_bwd(YpView x, YpView dy, YpView dx, int dtype, int B, int accumulate, void* workspace, size_t workspace_bytes, void* stream);
/* backward of the descriptor L2 normalisation (models/YOLOPoint.py:219-220), fp32 views */
int yp_l2norm_bwd_f32(YpView x, YpView g, YpView dx, int B, int C, void* stream);
/* backward of MaxPool2d(2,2) (YOLOPointv52 descriptor branch, reference models/YOLOPoint.py:311): dx (+)= dy at the first maximum of each window */
int yp_maxpool2_bwd(YpView x, YpView dy, YpView dx, int dtype, int B, int accumulate, void* stream);
/* fp32 gradient of the permuted Detect output [B,na,ny,nx,no] -> NHWC view in `dtype` (inverse of models/yolo.py:53), multiplied by
 * the device scalar *scale_dev first (NULL: 1) -- the upstream factor of the object loss in train.py:240, folded into this pass */
int yp_detect_bwd_pack(const float* gx, int B, int na, int no, YpView out, int dtype, const float* scale_dev, void* stream);
/* NHWC view (optionally through its 2x upsample) -> [C][H][W][Bpad] copy, batch zero-padded (wgrad operand layout) */
int yp_to_chwb(YpView in, int dtype, int B, int C, void* out, int Bpad, void* stream);
/* out[c] (+)= sum over the B*H*W rows of view[., c]  (bias gradients); workspace as yp_bn_workspace_bytes */
/* fp32 NHWC view -> `dtype` NHWC view */
int yp_cast_from_f32(YpView in, YpView out, int dtype, int B, void* stream);
int yp_col_sum(YpView v, int dtype, int B, float* out, int accumulate, void* workspace, size_t workspace_bytes, void* stream);
/* Weight gradient of a convolution (k = 1 stride 1, or k = 3 with pad 1 and stride 1 | 2) straight from the NHWC tensors, no
 * transposed copies: dw[ci][r][s][co] += sum_{b,y,x} x[b, y*stride+r-p, x*stride+s-p, ci] * dy[b, y, x, co]   (fp32 atomics:
 * `dw` [x.C][k][k][dy.C] must be zero-initialised).  x may be read through its 2x nearest upsample (x.ups = 1).
 * 16-bit dtypes.  replaces: autograd's conv2d weight gradient (reference train.py:245 loss.backward()). */
int yp_conv_wgrad(YpView x, YpView dy, int dtype, int B, int k, int stride, float* dw, void* stream);

/* Every weight gradient of ONE filter class (k, stride as yp_conv_wgrad, and one workgroup block size) of a backward pass in a single
 * launch.  yp_wgrad_group_pack fills a host table of n entries of yp_wgrad_group_entry_bytes() bytes each (argument checks as
 * yp_conv_wgrad) and returns the launch size; the caller copies the table to device memory and replays
 * yp_wgrad_group_run(table_dev, n, total_blocks, ...).
 * block: a workgroup owns a [block ci x block co] piece of dW -- 64, or 128 for 1x1 filters with >= 128 channels on both sides and enough
 * pixels per workgroup (every LDS fragment feeds four MFMAs instead of two; yp_wgrad_block tells which one yp_conv_wgrad itself would
 * take for a pair of views at batch B). */
size_t yp_wgrad_group_entry_bytes(void);
int yp_wgrad_block(YpView x, YpView dy, int B, int k);
int yp_wgrad_group_pack(const YpView* xs, const YpView* dys, float* const* dws, int n, int dtype, int B, int k, int stride, int block, void* table_host,
                        int* total_blocks);
int yp_wgrad_group_run(const void* table_dev, int n, int total_blocks, int dtype, int k, int stride, int block, void* stream);
/* Deterministic variant: every pixel slice of an entry writes its own partial slab (plain stores, parts[i] = device buffer of
 * yp_wgrad_partial_elems(...) floats, 16-byte aligned) and a second launch sums the slabs in slice order into dW -- bit-reproducible
 * gradients, no floating-point atomics, dW needs no clearing.  pack_det also returns the fold launch size for run_det. */
size_t yp_wgrad_partial_elems(YpView x, YpView dy, int dtype, int B, int k, int stride, int block);
int yp_wgrad_group_pack_det(const YpView* xs, const YpView* dys, float* const* dws, float* const* parts, int n, int dtype, int B, int k, int stride,
                            int block, void* table_host, int* total_blocks, int* fold_chunks);
int yp_wgrad_group_run_det(const void* table_dev, int n, int total_blocks, int fold_chunks, int dtype, int k, int stride, int block, void* stream);
/* 8-bit operands (BASELINE.json configs[4]; no reference counterpart -- the reference trains 16-bit, src/train.py:45-46,206): x8 = the e4m3
 * twin of the layer's input, dy8 = the e5m2 twin of its output gradient (1-byte NHWC views, 16-channel aligned: the bytes the forward /
 * dgrad convolutions of the same layer multiply, yp_quantize_fp8), *sx / *sdy their per-tensor scales (device scalars, real = stored x scale).
 * dw as yp_conv_wgrad (fp32, the sums multiplied by *sx x *sdy).  Grouped form: yp_wgrad_group_pack_q8 = yp_wgrad_group_pack_det with
 * dtype YP_FP8 and one scale pair per entry (parts may be NULL: fp32 atomics); run with yp_wgrad_group_run_det(..., dtype = YP_FP8, ...);
 * yp_wgrad_partial_elems / yp_wgrad_block take dtype YP_FP8 for such views. */
int yp_conv_wgrad_q8(YpView x8, YpView dy8, const float* sx, const float* sdy, int B, int k, int stride, float* dw, void* stream);
int yp_wgrad_group_pack_q8(const YpView* xs, const YpView* dys, float* const* dws, float* const* parts, const float* const* sx, const float* const* sdy,
                           int n, int B, int k, int stride, int block, void* table_host, int* total_blocks, int* fold_chunks);

/* dw[ci][r][s][co] (fp32, Cout_pad channels per tap: the layout yp_conv_wgrad / the wgrad-as-convolution path produce)
 * -> grad[co][c0+ci][r][s] for ci < creal, co < Cout: the reference layout of conv.weight.grad */
int yp_wgrad_unpack(const float* dw, float* grad, int Cout, int Cin, int k, int c0, int creal, int Cout_pad, void* stream);

/* Every weight gradient of a backward pass in ONE launch (tiled LDS transposes, optional sum over `split` partial copies).
 * Entry: dw [rows = creal*k*k][cout_pad] fp32 (+ p*pstride floats for partial p) -> grad[co*out_stride + out_off + row] for
 * co < cout; out_stride = Cin*k*k, out_off = c0*k*k; tiles of 32 x 32 are numbered across the table: tile0 = the entry's first,
 * it owns ceil(rows/32)*ceil(cout/32) of them.  The table lives in device memory, sorted by tile0. */
typedef struct YpUnpackEntry {
    const float* dw;
    float* grad;
    int64_t rows, cout, cout_pad, out_stride, out_off, tile0, split, pstride;
} YpUnpackEntry;
int yp_wgrad_unpack_batch(const YpUnpackEntry* table_dev, int n_entries, int total_tiles, void* stream);
/* fp32 OIHW master filter w[Cout][Cin][R][S] -> the packed [Npad + 1][Kpad] `dtype` filter yp_conv2d reads (zero padded,
 * zero row last), so a training step re-derives its 16-bit filters on the device without host work:
 *   mode 0  forward filter of input-channel slice [c0, c0+Cj):  dst[n][(r*S+s)*Cj + c]       = w[n][c0+c][r][s]
 *   mode 1  dgrad filter (flipped, channel-transposed):         dst[c][(r*S+s)*Cout_pad + n] = w[n][c0+c][R-1-r][S-1-s]
 *   mode 2  image-like filter (Cin <= 4, S even) for the 16-bit stem, which reads the packed image as [H][W/2][8] (two pixels of
 *           4 channels):                                        dst[n][(r*S/2 + s/2)*8 + (s%2)*4 + c] = w[n][c][r][s]  (c < Cin, else 0)
 *   mode 3  the same filter padded to 4 channels (fp32 plans):  dst[n][(r*S+s)*4 + c]        = w[n][c][r][s]
 *   mode 4 + 2 py + px  (3x3 filters) the dgrad filter of a stride-2 / pad-1 convolution for the input pixels of parity (py, px):
 *           dst[c][(r'*(1+px) + s')*Cout_pad + n] = w[n][c0+c][r][s],  r = 1 (py = 0) | 2 - 2 r' (py = 1), s likewise  (YpConvDesc.out_phase)
 * bias (may be NULL) is copied, zero padded, to bias_dst[Npad].
 * replaces: the conv.weight casts autocast performs every forward (reference train.py:206) + autograd's filter transposes */
int yp_pack_weight(const float* w, int Cout, int Cin, int R, int S, int c0, int Cj, int mode, int Cout_pad, void* dst, int Kpad,
                   int Npad, int dtype, const float* bias, float* bias_dst, void* stream);

/* yp_pack_weight for a whole parameter set in ONE launch: a device table of argument sets (same meaning as yp_pack_weight's);
 * blocks of 1024 packed elements are numbered across the table, blk0 = an entry's first block, it owns
 * ceil((Npad + 1) * Kpad / 1024) of them; sorted by blk0. */
typedef struct YpPackEntry {
    const float* w;
    void* dst;
    const float* bias;
    float* bias_dst;
    int64_t Cout, Cin, R, S, c0, Cj, mode, Cout_pad, Kpad, Npad, blk0;
} YpPackEntry;
int yp_pack_weight_batch(const YpPackEntry* table_dev, int n_entries, int total_blocks, int dtype, void* stream);

/* ---- 8-bit (OCP fp8) training convolutions: BASELINE.json configs[4].  real = stored * scale, per tensor; scales are device scalars.
 * yp_quantize_fp8: 16-bit NHWC view -> 1-byte NHWC view (format 0 = e4m3, saturating at 448; 1 = e5m2, saturating at 57344) with the
 * CURRENT *scale, and max|src| recorded into amax (may be NULL): amax points at 256 floats, the recorded maximum is their maximum (the
 * workgroups spread their atomic max over them).  yp_fp8_update_scales: for n tensors at once (amax: n x 256 floats),
 * scale[i] = max(amax[i][:]) * margin / fmax[i] where that maximum is > 0 (else kept), amax[i][:] = 0 -- the next step's scales
 * ("delayed scaling").
 * yp_pack_weight_fp8_batch: yp_pack_weight_batch for e4m3 packed filters (blocks of 1024 packed elements; each entry quantises with
 * its *scale and records max|w| into *amax). */
int yp_quantize_fp8(YpView src, YpView dst, int src_dtype, int B, int format, const float* scale, float* amax, void* stream);
int yp_fp8_update_scales(float* scale, float* amax, const float* fmax, int n, float margin, void* stream);
typedef struct YpPackEntry8 {
    const float* w;
    void* dst;
    const float* scale;
    float* amax;
    int64_t Cout, Cin, R, S, c0, Cj, mode, Cout_pad, Kpad, Npad, blk0;
} YpPackEntry8;
int yp_pack_weight_fp8_batch(const YpPackEntry8* table_dev, int n_entries, int total_blocks, void* stream);

/* InfoNCE descriptor loss (reference utils/loss_functions.py:484-597) without the gathered-negatives / Gram-matrix tensors.
 *   da, db [n][D] fp32 (D a multiple of 64, <= 256): sampled descriptors of the image / the warped image
 *   idx [n][E] int32: column 0 = the row itself (the match), columns 1.. = the sampled negatives (rows of db); E <= 512
 *   forward:  logits[i][j] = <da[i], db[idx[i][j]]> * inv_tau,  loss_rows[i] = logsumexp_j logits[i][j] - logits[i][0]
 *   backward: with the upstream gradient of mean(loss_rows) folded into *grad_scale_dev (= g * inv_tau / n, device scalar):
 *             dda, ddb [n][D]; `order` / `offsets` are the edge list (edge = i*E + j) sorted by idx value and its CSR offsets
 *             [n+1] (built once with the sampling); w_scratch [n][E] fp32. */
int yp_infonce_fwd(const float* da, const float* db, const int* idx, int n, int E, int D, float inv_tau, float* logits, float* loss_rows, void* stream);
/* Training variant: the forward and the anchor-side gradient in ONE gather pass (a streaming softmax over the E logits of each anchor):
 * dda_unscaled[i] = sum_j softmax_j * db[idx[i][j]] - db[idx[i][0]]; the caller multiplies by dL/dloss / (tau * n).  lse[i] = the row's
 * log-sum-exp; `logits` receives the softmax weights w[i][j] = exp(l_ij - lse_i) - [j == 0], not the logits.  yp_infonce_bwd_db then gives
 * ddb (already scaled by *grad_scale_dev) from those weights. */
int yp_infonce_fwd_grad(const float* da, const float* db, const int* idx, int n, int E, int D, float inv_tau, float* logits, float* loss_rows, float* lse,
                       float* dda_unscaled, const int* n_dev, int max_workgroups, void* stream);
int yp_infonce_bwd_db(const float* da, const int* order, const int* offsets, const float* logits, const float* lse, int n, int E, int D,
                      const float* grad_scale_dev, float* ddb, const int* n_dev, int max_workgroups, void* stream);
/* The two gathers over a 16-bit copy of the descriptor table (bf16 rows: half the gathered bytes; sums / softmax state in fp32).  rows16 =
 * yp_infonce_rows16(rows = the fp32 table [2n][D]: anchors then matches, count = 2 n D elements); yp_infonce_fwd_grad_h reads the anchor's own
 * row from the fp32 table `da` and gathers the match rows from rows16 + n D; yp_infonce_bwd_db_h gathers the anchor rows from rows16.
 * D in {64, 128, 256}.  For the 16-bit training modes (engine.TrainStep); the fp32 forms above are what the reference fixtures pin. */
int yp_infonce_rows16(const float* rows, size_t count, void* rows16, void* stream);
int yp_infonce_fwd_grad_h(const float* da, const void* rows16, const int* idx, int n, int E, int D, float inv_tau, float* logits, float* loss_rows, float* lse,
                          float* dda_unscaled, const int* n_dev, int max_workgroups, void* stream);
int yp_infonce_bwd_db_h(const void* rows16, const int* order, const int* offsets, const float* logits, int n, int E, int D, const float* grad_scale_dev,
                        float* ddb, const int* n_dev, int max_workgroups, void* stream);
/* max_workgroups (both): 0 = one workgroup per four rows.  > 0 caps the grid, the workgroups then walk the rows -- for a caller that runs
 * these gathers on one stream beside other kernels on another (they are latency-bound and would otherwise take every CU slot): the training
 * step passes 768 (3 per CU), which is worth 4 % of the step at -s; 256 starves the gathers themselves. */
int yp_infonce_bwd(const float* da, const float* db, const int* idx, const int* order, const int* offsets, const float* logits, int n, int E, int D,
                   const float* grad_scale_dev, float* w_scratch, float* dda, float* ddb, void* stream);

/* The descriptor lookup in front of the InfoNCE loss: F.grid_sample(desc, uv, bilinear, align_corners=True, zero padding) at P
 * points per image (reference utils/loss_functions.py:553-560).  map / gmap: [B,H,W,D] fp32, channels innermost (the layout the
 * network emits; D % 64 == 0, D <= 256); uv [B*P,2] normalised (x, y); out / g [B*P, D].  gmap must be ZEROED by the caller; the
 * backward accumulates with atomics. */
int yp_points_sample_fwd(const float* map_nhwc, int B, int H, int W, int D, const float* uv, int P, float* out, const int* p_dev, void* stream);
int yp_points_sample_bwd(const float* g, int B, int H, int W, int D, const float* uv, int P, float* gmap_nhwc, void* stream);
/* The backward of the lookup without atomics and without a zeroed map: yp_points_sample_taps writes, per (point, tap), the cell it touches
 * (b*H*W + y*W + x, INT_MAX for taps outside the map / of weight 0) into keys [B*P*4]; the caller sorts (key, 4*point + tap) pairs by key
 * (stable) and builds CSR offsets [B*H*W + 1] over the cells -- label-only work, once per batch of sample points; yp_points_sample_bwd_sorted
 * then writes EVERY row of gmap [B][H][W][D]: the sum of its contributions in sorted order (bit-reproducible), zeros where nothing lands.
 * Device-side counts (the trailing *_dev pointers of these entry points, all may be NULL): the InfoNCE sampling leaves its counts on the
 * device (yp_nce_select's meta: [0] = points per image P, [1] = matched rows n); with them the host passes CAPACITIES (P = samples per image,
 * n = B * samples) for the launch sizes and the kernels read the real counts -- arrays stay compact ([B][P] points, [2n][D] descriptors with the
 * second half at row n), entries behind the counts are skipped.  yp_infonce_fwd_grad with n_dev: db may be NULL, = da + n * D;
 * yp_infonce_bwd_db with n_dev: pass the BASE of the [2n][D] gradient as ddb.  No host synchronisation in the loss stage.
 * row_scale_dev (may be NULL): rows [0, n_scaled_rows) of g are multiplied by this device scalar before they are used (the anchor half of
 * yp_infonce_fwd_grad's gradient, which still lacks dL/dloss / (tau * n)) -- no separate scaling pass.
 * replaces: the scatter of grid_sampler_2d_backward in loss.backward() (utils/loss_functions.py:553-560, train.py:245). */
int yp_points_sample_taps(const float* uv, int B, int P, int H, int W, int* keys, const int* p_dev, void* stream);
int yp_points_sample_bwd_sorted(const float* g, int B, int H, int W, int D, const float* uv, int P, const int* order, const int* offsets,
                                const float* row_scale_dev, int n_scaled_rows, float* gmap_nhwc, const int* n_scaled_dev, void* stream);

/* YOLOv5 object loss of ONE Detect level, value and gradient (reference utils/loss_functions.py:90-176 ComputeLoss.__call__ body
 * of the per-level loop; CIoU: utils/metrics_yolo.py:202-240).  p / dp: [cells, no] fp32 with cells = B*na*ny*nx and no = 5 + nc;
 * the n (target, cell) entries of the level (reference build_targets, :178-234) arrive flattened: cell[e] = ((b*na+a)*ny+gj)*nx+gi,
 * tbox[e] = (dx, dy, w, h) in grid units, anch[e] = the anchor (w, h), tcls[e] = class.  w_box / w_obj / w_cls = hyp gain (x level
 * balance for obj).  Adds the weighted box / obj / cls terms to sums[0..2] (device, caller-zeroed) and writes d(sum)/dp to dp.
 * tobj follows index_put semantics on duplicates: the entry with the highest index owns the cell.
 * iou_scratch: n floats; owner_scratch: cells ints. */
int yp_objloss_level(const float* p, int cells, int no, int nc, const int* cell, const float* tbox, const float* anch, const int* tcls, int n, float cp,
                     float cn, float cls_pw, float obj_pw, float w_box, float w_obj, float w_cls, float* iou_scratch, int* owner_scratch, float* dp,
                     float* sums, void* stream);
/* The same with the entry count read on the device (n_dev != NULL: `n` is then the capacity of the entry arrays): what follows
 * yp_build_targets without a host synchronisation. */
int yp_objloss_level_dev(const float* p, int cells, int no, int nc, const int* cell, const float* tbox, const float* anch, const int* tcls, int n,
                         const int* n_dev, float cp, float cn, float cls_pw, float obj_pw, float w_box, float w_obj, float w_cls, float* iou_scratch,
                         int* owner_scratch, float* dp, float* sums, void* stream);

/* YOLOv5 target assignment on the device (reference utils/loss_functions.py:177-234 ComputeLoss.build_targets): labels [nt,6]
 * (image, class, xc, yc, w, h normalised) x anchors [nl][na][2] (grid units) x shapes_dev [nl][2] (ny, nx; device ints) -> per level
 * the entry list in the reference's order (offset-major, anchor, label): cell [nl][cap], tcls [nl][cap], tbox [nl][cap][4],
 * anch [nl][cap][2], count [nl]; cap >= 5 * na * nt.  No host synchronisation (the reference's boolean-mask indexing has one per level). */
int yp_build_targets(const float* targets, int nt, const float* anchors, int nl, int na, const int* shapes_dev, float anchor_t, int cap, int* cell,
                     int* tcls, float* tbox, float* anch, int* count, void* stream);


/* Keypoint-detector loss (reference utils/loss_functions.py:600-619 ComputeDetectorLoss): sums[0] = sum over cells of
 * mask * sum_c BCE(softmax(semi)_c, target_c) (PyTorch's BCE: logs clamped at -100), sums[1] = sum of mask; dsemi (same strides as
 * semi) = d sums[0] / d semi.  semi / target: fp32 [B,65,Hc,Wc] with element strides (b, c, y, x); mask: contiguous [B,Hc,Wc].
 * loss = sums[0] / (sums[1] + 1e-10), so its gradient is dsemi / (sums[1] + 1e-10). */
size_t yp_detloss_workspace_bytes(int B, int Hc, int Wc);
int yp_detloss(const float* semi, const int64_t* semi_strides, const float* target, const int64_t* target_strides, const float* mask, int B, int Hc, int Wc,
               float* dsemi, float* sums, void* workspace, size_t workspace_bytes, void* stream);

/* The same loss straight from the 2-D label maps, as train.py:212-231 feeds it (labels2Dto3D(labels_2D) / getMasks(valid_mask),
 * utils/utils.py:184-209 / :103-116) -- neither the [B,65,Hc,Wc] target nor the cell mask is built by framework kernels:
 *   yp_cell_mask   valid2d [B,1,H,W] fp32 -> mask [B,H/8,W/8] (product of a cell's 64 pixels) and mask_sum[0] = its sum.  Label-only.
 *   yp_detloss2d   semi as above; labels2d [B,1,H,W]: the target of a cell is its 64 pixels and the dustbin 1 - sum (0 when < 1),
 *                  divided by their sum.  loss[0] = sum(mask * BCE) / (mask_sum + 1e-10); dsemi (its own strides: the network's
 *                  gradient buffer) = d loss / d semi * gscale -- the final gradient, no scaling pass.
 * Both take a workspace of yp_cell_mask_workspace_bytes(B, H, W). */
size_t yp_cell_mask_workspace_bytes(int B, int H, int W);
int yp_cell_mask(const float* valid2d, int B, int H, int W, float* mask, float* mask_sum, void* workspace, size_t workspace_bytes, void* stream);
int yp_detloss2d(const float* semi, const int64_t* semi_strides, const float* labels2d, const float* mask, const float* mask_sum, float gscale, int B, int H,
                 int W, float* dsemi, const int64_t* dsemi_strides, float* loss, void* workspace, size_t workspace_bytes, void* stream);

/* Process-wide tuning knob: caps the grids of the large launches of yp_nce_cells / yp_csr_build (their workgroups then walk the items).
 * 0 (default) = no cap.  engine.TrainStep sets it: its label work runs on a side stream beside the forward pass, where thousands of tiny
 * workgroups would take the CU slots the convolutions wait for.  Results do not depend on it. */
int yp_sampling_set_max_workgroups(int n);
/* The label-only half of the InfoNCE loss (reference utils/loss_functions.py:484-552: warp the validity mask back, keep the cells whose
 * 64 pixels are all valid, map the cell grid through the inverse homography, shuffle, draw the negatives) as device kernels around a
 * counter-based generator (Philox 4x32-10 keyed by `seed`) -- csrc/sampling.hip:
 *   yp_nce_cells      mask [B,1,H,W] fp32, inv_h [B,3,3] (normalised coordinates) -> valid [B*Hc*Wc] bytes, uvb [B*Hc*Wc][2] = the rounded
 *                     cell coordinates in the warped image (Hc = H/8, Wc = W/8)
 *   yp_nce_select     pool = min(samples, min over images of #valid) cells per image, uniformly without replacement, in cell order:
 *                     uab [2B][pool][2] normalised (x, y) (rows [0,B): the cells, rows [B,2B): their matches; capacity 2*B*samples*2 floats);
 *                     meta[0] = pool, meta[1] = n = B*pool, meta[2] = 0
 *   yp_nce_negatives  idx [n][1+negs] int32: column 0 the row itself, then uniform draws from [0, n); a draw equal to its row is replaced by
 *                     floor(U * #such draws) (the reference's redraw from [0, #collisions)); meta[2] receives that count.  n_from_meta: `n` is
 *                     the CAPACITY of idx, the row count is meta[1] (no host read-back); rows [meta[1], n) get keys the sort skips
 *   yp_csr_build      keys [n_items] (values outside [0, n_buckets) are skipped) -> order: the item ids grouped by key, ascending inside a
 *                     group; offsets [n_buckets + 1]; workspace: yp_csr_workspace_ints(n_items, n_buckets) ints.  wide_buckets: groups of hundreds (one wavefront
 *                     each) instead of a few (one thread each).  The InfoNCE backward walks (idx.flatten() -> n buckets) and the descriptor
 *                     lookup backward (yp_points_sample_taps keys -> B*H*W buckets) with it. */
int yp_nce_cells(const float* mask, const float* inv_h, int B, int H, int W, unsigned char* valid, float* uvb, void* stream);
int yp_nce_select(const unsigned char* valid, const float* uvb, int B, int Hc, int Wc, int samples, uint64_t seed, float* uab, int* meta, void* stream);
int yp_nce_negatives(int n, int negs, uint64_t seed, int* meta, int* idx, int n_from_meta, void* stream);
size_t yp_csr_workspace_ints(int n_items, int n_buckets);
int yp_csr_build(const int* keys, int n_items, int n_buckets, int wide_buckets, int* order, int* offsets, int* workspace, void* stream);

/* The pieces between the plans of a training step (reference train.py:189-259), one launch each (csrc/step.hip):
 *   yp_fill_zero     optimizer.zero_grad() over the gradient arena (16-byte aligned pointer and size)
 *   yp_multi_add     loss.backward()'s accumulation into p.grad for a whole table of (dst, src, n) at once: dst += src (mode 0) or
 *                    dst = src (mode 1); blocks of 1024 elements are numbered across the table, blk0 = an entry's first, sorted by blk0
 *   yp_counters_add  BatchNorm's num_batches_tracked (+= inc for every int64 counter of a device pointer table)
 *   yp_loss_combine5 train.py:232-241: out5[0] = (sum(det_losses[0..n_det)) + lambda_desc * mean(nce_rows[0..n_rows))) + lambda_obj *
 *                    sum(obj_sums[0..3)), times `scale` when it is not 1; out5[1..3] = the detector / descriptor / object terms;
 *                    out5[4] = the InfoNCE row count the mean was taken over, as a float (always written; 0: an image had no valid cell
 *                    under its warp -- the step ran without a descriptor term, where the reference's mean over nothing would have been NaN);
 *                    *desc_scale_out = desc_scale (the device scalar yp_infonce_bwd_db reads; NULL: not written).  n_rows_dev (may be NULL):
 *                    the row count lives on the device (yp_nce_select's meta[1]); desc_scale is then g_desc / (tau * n) computed there.
 *                    (Round 6: renamed from yp_loss_combine, whose output had four floats, so that a stale caller fails to link.) */
typedef struct YpAddEntry {
    float* dst;
    const float* src;
    int64_t n, mode, blk0;
} YpAddEntry;
int yp_fill_zero(void* p, size_t bytes, void* stream);
int yp_multi_add(const YpAddEntry* table_dev, int n_entries, int total_blocks, void* stream);
int yp_counters_add(int64_t* const* table_dev, int n, int64_t inc, void* stream);
int yp_loss_combine5(const float* det_losses, int n_det, const float* nce_rows, int n_rows, const float* obj_sums, float lambda_desc, float lambda_obj, float scale,
                     float desc_scale, float* out5, float* desc_scale_out, const int* n_rows_dev, float g_desc, double tau, void* stream);

/* One generic launch record: `op` selects one of the functions above, the slots carry its arguments in the
 * order documented next to each opcode.  Lets training plans replay any mix of launches (yp_plan_add_op). */
enum {
    YP_OP_BN_STATS = 10,      /* v0=raw; i0=dtype i1=B; s0=eps s1=momentum; g0=mean g1=invstd g2=running_mean g3=running_var; p0=ws n0=ws_bytes;
                                 i2=rows > 0: finalize only (yp_bn_finalize) from p1 = the partial sums a convolution wrote; i3=groups (0 = 1) */
    YP_OP_BN_APPLY = 11,      /* v0=raw v1=out v2=res; i0=dtype i1=B i2=act i3=groups; f0=mean f1=invstd f2=gamma f3=beta; v3=1-byte twin (ptr NULL: none) g2=its scale g3=its amax */
    YP_OP_BN_BWD = 12,        /* v0=raw v1=dy v2=dx; i0=dtype i1=B i2=act i3=accumulate i4=groups; f0..f3 as above; g0=dgamma g1=dbeta; p0=ws n0=ws_bytes; v3 / g2 / g3: 1-byte twin of dx as for BN_APPLY;
                               * p1 != NULL: the shortcut gradient of a Bottleneck (x + cv2(cv1(x)), models/common.py:79-89) -- a tensor of dy's geometry at p1, channel stride i5,
                               * channel offset i6 -- receives dy (i7 = 1: accumulates it) in the same row pass */
    YP_OP_UPS2_BWD = 13,      /* v0=in v1=out; i0=dtype i1=B i2=accumulate */
    YP_OP_ADD_VIEWS = 14,     /* v0=src v1=dst; i0=dtype i1=B i2=accumulate */
    YP_OP_MAXPOOL5_BWD = 15,  /* v0=x v1=dy v2=dx; i0=dtype i1=B i2=accumulate; p0=ws n0=ws_bytes */
    YP_OP_L2NORM_BWD = 16,    /* v0=x v1=g v2=dx; i1=B i2=C */
    YP_OP_DETECT_BWD_PACK = 17, /* f0=gx f1=scale (device scalar, may be NULL); v0=out; i0=dtype i1=B i2=na i3=no */
    YP_OP_TO_CHWB = 18,       /* v0=in; i0=dtype i1=B i2=C i3=Bpad; p0=out */
    YP_OP_COL_SUM = 19,       /* v0=view; i0=dtype i1=B i2=accumulate; g0=out; p0=ws n0=ws_bytes */
    YP_OP_MEMSET0 = 20,       /* p0=ptr n0=bytes */
    YP_OP_PACK_NCHW = 21,     /* f0=x_nchw; v0=out; i0=dtype i1=B i2=C */
    YP_OP_L2NORM = 22,        /* v0=in v1=out; i1=B i2=C */
    YP_OP_SPPF_POOL = 23,     /* v0=x v1..v3=y1..y3; i0=dtype i1=B */
    YP_OP_CAST_F32 = 24,      /* v0=in (fp32) v1=out; i0=dtype i1=B */
    YP_OP_MAXPOOL2 = 25,      /* v0=x v1=y; i0=dtype i1=B */
    YP_OP_WGRAD = 27,         /* v0=x v1=dy; p0=dw; i0=dtype i1=B i2=k i3=stride */
    YP_OP_MAXPOOL2_BWD = 29,  /* v0=x v1=dy v2=dx; i0=dtype i1=B i2=accumulate */
    YP_OP_WGRAD_UNPACK = 28,  /* p0=dw [Cj][k][k][Cout_pad] fp32 -> g0=grad OIHW [Cout][Cin][k][k] fp32, input-channel slice [c0, c0+creal):
                               * i1=Cout i2=Cin i3=k i4=c0 i5=creal i6=Cout_pad */
    YP_OP_WGRAD_UNPACK_BATCH = 30, /* p0=device table of YpUnpackEntry; i1=entries i2=total tiles */
    YP_OP_SUM_SLABS = 32,     /* p0=slabs p1=dst; n0=elements per slab (multiple of 4) n1=slabs: dst = slab0 + slab1 + ... in order */
    YP_OP_QUANT_FP8 = 34,     /* v0=src (16-bit) v1=dst (1-byte); i0=src dtype i1=B i2=format (0 e4m3 | 1 e5m2); p0=scale p1=amax: yp_quantize_fp8 */
    YP_OP_STEM_WGRAD = 33,    /* v0=image v1=dy; i0=dtype i1=B; p0=slabs p1=dw */
    YP_OP_WGRAD_GROUP = 31,   /* p0=device table (yp_wgrad_group_pack[_det | _q8]); i0=dtype (YP_FP8: 8-bit entries) i1=entries i2=total blocks i3=k i4=stride i5=fold chunks (0: atomics) i6=block (0 = 64) */
    YP_OP_PACK_WEIGHT = 26    /* f0=w f1=bias; g0=bias_dst; p0=dst; i0=dtype i1=Cout i2=Cin i3=R i4=S i5=c0 i6=Cj i7=mode; n0=Kpad n1=Npad | Cout_pad<<32 */
};
typedef struct YpOpArgs {
    int32_t op, pad_;
    YpView v[4];
    const float* f[4];
    float* g[4];
    void* p[2];
    size_t n[2];
    int32_t i[8];
    float s[4];
} YpOpArgs;
int yp_run_op(const YpOpArgs* a, void* stream);

/* ------------------------------------------------------------------------------------
 * Post-processing
 * ---------------------------------------------------------------------------------- */
/* Keypoint heat-map decode: softmax over 65 channels, drop dustbin, depth-to-space(8).
 *   semi: fp32, element (b,c,y,x) at semi[b*sb + c*sc + y*sy + x*sx]  (any layout)
 *   mode 0: torch softmax (max-subtracted)              replaces utils/utils.py:232-262
 *   mode 1: exp(x)/(sum+1e-5) without max subtraction   replaces demo.py:140-150
 *   heat: fp32 [B, 8*Hc, 8*Wc] */
int yp_kp_decode(const float* semi, int B, int Hc, int Wc, int64_t sb, int64_t sc, int64_t sy, int64_t sx,
                 int mode, float* heat, void* stream);

/* Greedy grid NMS of heat-map peaks, bit-exact with the sequential reference for distinct scores.
 *   heat [B,H,W] fp32; candidates are pixels with heat >= conf_thresh; a kept point suppresses every
 *   lower-scored candidate inside its (2r+1)^2 window; kept points within `border` px of the image
 *   edge are dropped afterwards; survivors are returned sorted by score (desc, ties by index asc).
 *   out_xyc [B,max_out,3] fp32 (x, y, conf); out_count [B] int32
 * replaces: utils/utils.py:118-182 (nms_fast) + :465-485 (getPtsFromHeatmap) */
size_t yp_kp_nms_workspace_bytes(int B, int H, int W);
/* byte offset, inside that workspace, of the per-image count (int32 [B]) of pixels that passed conf_thresh in the last yp_kp_nms* call */
size_t yp_kp_nms_candidate_count_offset(int B, int H, int W);
int yp_kp_nms(const float* heat, int B, int H, int W, float conf_thresh, int radius, int border,
              float* out_xyc, int32_t* out_count, int max_out, void* workspace, size_t workspace_bytes,
              void* stream);
/* The same without a host synchronisation: exactly `rounds` fix-point rounds are enqueued (rounds after convergence exit at once) and
 * the number of candidates still undecided after the last one is written to *undecided_out (device memory).  The result is the
 * reference's iff that counter is 0: the caller checks it at its own synchronisation point and falls back to yp_kp_nms otherwise. */
int yp_kp_nms_async(const float* heat, int B, int H, int W, float conf_thresh, int radius, int border, float* out_xyc, int32_t* out_count,
                    int max_out, void* workspace, size_t workspace_bytes, int rounds, int32_t* undecided_out, void* stream);

/* Batched box NMS on decoded predictions.
 *   pred [B,N,5+nc] fp32 (xywh, obj, cls...).  Output rows (x1,y1,x2,y2,conf,cls) sorted by conf desc.
 *   out_det [B,max_det,6] fp32; out_count [B] int32
 * replaces: utils/general_yolo.py:124-235 incl. torchvision.ops.nms (:218) */
size_t yp_box_nms_workspace_bytes(int B, int N, int nc, int multi_label, int max_nms);
int yp_box_nms(const float* pred, int B, int N, int nc, float conf_thres, float iou_thres,
               int multi_label, int agnostic, int max_det, int max_nms, float max_wh,
               float* out_det, int32_t* out_count, void* workspace, size_t workspace_bytes, void* stream);
/* The same with the reference's `classes` argument (utils/general_yolo.py:128,199-200): class_mask = device bit mask, bit j of word
 * j/32 set when class j is kept (NULL: every class). */
int yp_box_nms_classes(const float* pred, int B, int N, int nc, float conf_thres, float iou_thres,
                       int multi_label, int agnostic, int max_det, int max_nms, float max_wh, const uint32_t* class_mask,
                       float* out_det, int32_t* out_count, void* workspace, size_t workspace_bytes, void* stream);


/* Homography adaptation aggregate (reference export_homography.py:94-96,143-145 with warp_image_batch utils/utils.py:333-376):
 * heat / mask: [N,H,W] fp32 per-view heat maps (flattenDetection output) and valid masks; inv_homographies [N,9] row-major in
 * normalised [-1,1] coordinates.  out[H,W] = sum_v warp(heat_v*mask_v) / sum_v warp(mask_v) (bilinear, align_corners, zero pad;
 * NaN where no view covers a pixel, as in the reference); out_cover (optional) = the denominator. */
int yp_homo_combine(const float* heat, const float* mask, const float* inv_homographies, int N, int H, int W, float* out, float* out_cover,
                    void* stream);

/* Drop the keypoints that fall inside a detected box: pts_xyc [n,3] (x, y, conf; order kept) against boxes [n_boxes, box_stride]
 * (x1,y1,x2,y2,...), bounds = rint(xyxy) with numpy slice semantics on an H x W mask.  Counts may live on the device
 * (n_pts_dev / n_boxes_dev non-NULL override the host values, which then only bound the launch) so that the whole frame
 * pipeline runs without a host sync.  At most 512 boxes are honoured.  replaces: demo.py:176-196 (filter_points) */
int yp_pts_box_filter(const float* pts_xyc, const int* n_pts_dev, int n_pts, const float* boxes, const int* n_boxes_dev, int n_boxes,
                      int box_stride, int H, int W, float* out_xyc, int* out_count, void* stream);
/* Bilinear descriptor sampling (grid_sample align_corners=True with the reference's
 * full-resolution normalisation) + L2 renormalisation.
 *   desc: fp32, element (c,y,x) at desc[c*sc + y*sy + x*sx]; pts [N,2] fp32 (x,y) in pixels;
 *   out [D,N] fp32 column-per-point (reference layout)
 * replaces: evaluations/descriptor_evaluation.py:148-181, demo.py:200-215 */
int yp_desc_sample(const float* desc, int D, int Hc, int Wc, int64_t sc, int64_t sy, int64_t sx,
                   const float* pts_xy, int N, int cell, float* out, void* stream);

/* Mutual nearest-neighbour matcher on unit descriptors.
 *   d1 [D,N1], d2 [D,N2] fp32 column-per-point.  dist = sqrt(2-2*clip(d1^T d2,-1,1)).
 *   out_match [3, min(N1,max_out)] fp32 rows (idx1, idx2, dist) compacted in idx1 order.
 * replaces: models/model_wrap.py:434-476 == demo.py:300-341 */
size_t yp_mnn_workspace_bytes(int N1, int N2);
int yp_mnn_match(const float* d1, int N1, const float* d2, int N2, int D, float nn_thresh,
                 float* out_match, int32_t* out_count, int max_out, void* workspace,
                 size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------
 * Execution plan: an immutable, caller-owned list of the launches above, replayed with
 * one call (optionally through a captured hipGraph).  Buffers referenced by the
 * descriptors must stay alive and fixed for the life of the plan.
 * replaces: the Python module-by-module dispatch of models/YOLOPoint.py:198-246
 * ---------------------------------------------------------------------------------- */
typedef struct YpPlan YpPlan;
int yp_plan_create(YpPlan** plan);
int yp_plan_destroy(YpPlan* plan);
int yp_plan_add_conv(YpPlan* plan, const YpConvDesc* d);
int yp_plan_add_conv_detect(YpPlan* plan, const YpConvDesc* d, const YpDetectDesc* det);
int yp_plan_add_sppf_pool(YpPlan* plan, YpView x, YpView y1, YpView y2, YpView y3, int B, int dtype);
int yp_plan_add_l2norm(YpPlan* plan, YpView in, YpView out, int B, int C);
int yp_plan_add_detect_decode(YpPlan* plan, YpView raw, int B, int na, int no, float stride,
                              const float* anchors_px_host, float* x_out, float* z_out,
                              int rows_total, int row_offset);
int yp_plan_add_op(YpPlan* plan, const YpOpArgs* a);
int yp_plan_num_ops(const YpPlan* plan);
/* Replace view `slot` (0..3) of generic op `op` (added with yp_plan_add_op) before yp_plan_instantiate_graph.  Used by the training graph to clear
 * the 16-bit output view of a BatchNorm op (YP_OP_BN_APPLY v[1], YP_OP_BN_BWD v[2]) whose readers all take the 1-byte twin: with a NULL 16-bit
 * view and a twin, yp_bn_act_apply_grouped_q8 / yp_bn_act_bwd_grouped_q8 store the twin only. */
int yp_plan_patch_op_view(YpPlan* plan, int op, int slot, YpView v);
/* true data dependencies of op `op`: the earlier ops it must wait for.  When every op has a
 * dependency list, yp_plan_instantiate_graph replaces the captured chain's edges by these, so
 * independent branches run concurrently.  Eager yp_plan_run ignores them (single stream order). */
int yp_plan_set_deps(YpPlan* plan, int op, const int* deps, int ndeps);
int yp_plan_graph_is_parallel(const YpPlan* plan);
/* Two-lane schedule: an op on YP_LANE_SIDE starts after everything added before it but runs BESIDE the ops added after it (second
 * stream; a parallel branch once the plan is a hipGraph); an op on YP_LANE_JOIN first waits for every side op.  The end of the
 * plan always joins.  The caller guarantees that no later main-lane op touches what a side op reads or writes. */
enum { YP_LANE_MAIN = 0, YP_LANE_SIDE = 1, YP_LANE_JOIN = 2 };
int yp_plan_set_lane(YpPlan* plan, int op, int lane);
/* A callback op: when the replay reaches it, fn(user, stream) runs on the host and may enqueue launches of its own on `stream` (the op's lane).
 * Replaces the host-side sequencing of reference demo.py:138-160 (forward -> numpy decode -> NMS): the keypoint post-processing needs the
 * keypoint head only, so the front end (yolopoint_amd/frontend.py) hangs it into the forward's side lane.  A plan with callback ops cannot
 * be captured into a hipGraph (yp_plan_instantiate_graph refuses); it replays eagerly.  fn returns YP_OK or an error code. */
/* A stream of the library's per-device pool that really runs BESIDE `main_stream` (tested once per caller stream: the runtime maps all
 * streams of a process onto a few hardware queues, and two streams on one queue serialise).  slot 0: the plans' side lane; slot 1: an
 * auxiliary stream for the caller (engine.TrainStep's loss / label stream).  The streams live as long as the process.
 * (no reference counterpart: the reference runs everything on PyTorch's current stream) */
int yp_stream_pick(void* main_stream, int slot, void** out_stream);
/* Forget the picks made for `main_stream` on the current device ((void*)-1: for every stream): to be called when the caller destroys that
 * stream -- a later stream at the same address would otherwise inherit companions tested against another queue assignment. */
int yp_stream_forget(void* main_stream);

typedef int (*yp_plan_callback_t)(void* user, void* stream);
int yp_plan_add_callback(YpPlan* plan, yp_plan_callback_t fn, void* user);
/* capture the op list into a hipGraph on `stream` (call once, after the last add) */
int yp_plan_instantiate_graph(YpPlan* plan, void* stream);
/* enqueue all ops (graph launch when instantiated) */
int yp_plan_run(YpPlan* plan, void* stream);
/* enqueue all ops eagerly and time each with hip events; ms_out[num_ops] (host) */
int yp_plan_profile(YpPlan* plan, void* stream, float* ms_out_host);
/* enqueue the plan `iters` times and return the mean wall ms per iteration measured with hip
 * events on `stream` (events recorded on the launch stream itself) */
int yp_plan_time(YpPlan* plan, void* stream, int iters, float* ms_per_iter_host);

#ifdef __cplusplus
}
#endif
#endif /* YOLOPOINT_HIP_H */
