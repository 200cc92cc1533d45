/*
 * creste_hip.h -- C ABI of libcreste_hip.so: the MI355X (gfx950) kernels behind the CREStE
 * perception -> costmap -> IRL hot path.
 *
 * The reference (ut-amrl/creste_public) has no FFI of its own: every op below replaces a
 * sequence of stock PyTorch ops inside the reference's nn.Modules; each entry point cites the
 * reference file:line whose arithmetic it reproduces.  The host-side Python mirror
 * (creste_public_amd/creste/...) binds these symbols with ctypes; INTEGRATION.md shows the stub.
 *
 * Conventions
 *   - plain pointers and sizes only; all pointers are DEVICE pointers owned by the caller
 *     (PyTorch allocations, `tensor.data_ptr()`); the library allocates no persistent memory.
 *   - activations are fp32 NHWC ("channels-last"): element (n,y,x,c) of a tensor with pixel
 *     stride `cs` lives at ((n*H + y)*W + x)*cs + c.  A pixel stride larger than the channel
 *     count addresses a channel slice of a wider buffer (zero-copy concat).
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*), re-entrant, and
 *     keeps no global state besides immutable kernel handles.
 *   - return value: 0 on success, negative on error (never throws);
 *     creste_last_error() returns a thread-local description of the last failure.
 */
#ifndef CRESTE_HIP_H
#define CRESTE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CRESTE_OK 0
#define CRESTE_ERR_ARG (-1)
#define CRESTE_ERR_HIP (-2)
#define CRESTE_ERR_NOCONV (-3) /* value iteration hit max_sweeps */

#define CRESTE_ACT_NONE 0
#define CRESTE_ACT_RELU 1
#define CRESTE_ACT_SWISH 2 /* x * sigmoid(x) (efficientnet_pytorch Swish) */

#define CRESTE_PREC_F32 0  /* v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulate */
#define CRESTE_PREC_BF16 1 /* v_mfma_f32_32x32x16_bf16: bf16 operands, fp32 accumulate   */
#define CRESTE_PREC_BF16X3 2 /* fp32 operands split into bf16 hi+lo, hi*hi+hi*lo+lo*hi on the bf16
                                MFMA, fp32 accumulate: ~2^-16 product error at 5.3x the fp32 MFMA rate */
#define CRESTE_PREC_BF16X6 3 /* fp32 operands split into THREE bf16 pieces (24 bits = the whole fp32
                                significand), the 6 piece products >= 2^-16 summed on the bf16 MFMA:
                                fp32-equivalent products (dropped terms <= 2^-24) at 2.7x the fp32 MFMA rate */
#define CRESTE_PREC_F16X3 4  /* fp32 operands, rescaled by a per-tensor / per-output-channel power of two, split
                                into fp16 hi+lo (22 significand bits), hi*hi+hi*lo+lo*hi on
                                v_mfma_f32_32x32x16_f16, fp32 accumulate: product error <= 2^-21 (below the
                                fp32 accumulation round-off of a K>=16 dot product) at 5.3x the fp32 MFMA
                                rate; needs creste_conv_desc.a_amax / w_unscale */
#define CRESTE_ALGO_DIRECT 0   /* implicit GEMM over the K*K taps (every shape) */
#define CRESTE_ALGO_WINOGRAD 1 /* F(2x2,3x3): stride-1 3x3 convs, see creste_conv_wino_* below */
/* creste_conv_desc.flags.  V_VALID (CRESTE_ALGO_WINOGRAD4): `work` already holds the transformed input of THIS input
 * tensor (same N, H, W, Cin, padding, precision) from a previous call on the same stream -- several convs that read one
 * tensor (the three BEV heads' first conv, inpainting.py:141-146) run the input transform once. */
#define CRESTE_CONV_V_VALID 1
/* EMIT_NEXT_V (CRESTE_ALGO_WINOGRAD4, CRESTE_PREC_BF16X6, pad 1, no residual / row mask / out_amax): this conv is the first of
 * a conv3x3 -> conv3x3 pair (reference Up.conv, effnet.py:15-28; DeconvHead.up1, inpainting.py:52-68).  Its output
 * act(conv + bias) is NOT written: `out` must point to the `work` buffer of the SECOND conv (3x3, stride 1, pad 1, Cin == this
 * Cout, same N / H / W, same precision), and the output transform writes that conv's transformed input there -- bit for bit
 * what the second conv's own input transform would have produced from the materialised tensor.  The second call passes
 * CRESTE_CONV_V_VALID (its `in` is not read).  out_cs / out_co are ignored. */
#define CRESTE_CONV_EMIT_NEXT_V 2
/* CRESTE_ALGO_WINOGRAD4, the two halves of `Upsample(x2, bilinear) -> conv3x3` run as FOUR PHASE convolutions on the
 * low-resolution map (reference DeconvHead.up2, inpainting.py:56-60; creste_upconv2x_ring_fix_f32 below):
 * REPLICATE_PAD: window pixels outside the image take the nearest border pixel instead of zero;
 * PHASE2X: Cout = 4 x C'; channel n of low-resolution pixel (y, x) is written as channel n % C' of pixel
 *   (2 y + (n / C' >> 1), 2 x + (n / C' & 1)) of `out` [N, 2 Ho, 2 Wo, out_cs] (bias per n); the outermost ring of that image
 *   is written WITHOUT the activation -- creste_upconv2x_ring_fix_f32 subtracts the taps the high-resolution conv's zero
 *   padding excludes and applies it. */
#define CRESTE_CONV_REPLICATE_PAD 4
#define CRESTE_CONV_PHASE2X 8
#define CRESTE_ALGO_WINOGRAD4 2 /* F(4x4,3x3): the same convs, transformed input materialised, see creste_conv_wino4_* */

const char* creste_last_error(void);
int creste_abi_version(void);

/* ---------------------------------------------------------------------------------------------
 * Dense convolution as implicit GEMM on the matrix cores, fused epilogue.
 * Replaces nn.Conv2d (+ folded eval-mode BatchNorm2d) (+ ReLU / swish) (+ residual add) as used by
 *   reference creste/models/blocks/effnet.py:16-23,74,87-89 (Up / 1x1), conv.py:21-29,48-55,63-85,
 *   inpainting.py:52-68,82-103, depth.py:126, distillation.py:179, and the third-party
 *   EfficientNet-B0 1x1 expand/project convs and ResNet-18 BasicBlocks (call sites effnet.py:83,
 *   inpainting.py:96-103).
 *   out[m, co] = act( sum_{ky,kx,ci} in[n, oy*s - pad_t + ky, ox*s - pad_l + kx, ci] * a_scale[n,ci]
 *                     * w[co, ky, kx, ci]  + bias[co] + res[m, co] ) * row_mask[m]
 * `wpk` is the packed weight produced by creste_conv_pack_weight_f32 (GEMM B operand,
 * [cout_pad][KH*KW*cin_pad], zero padded), NOT the torch OIHW tensor. */
typedef struct creste_conv_desc {
  const float* in;       /* [N,H,W,in_cs] */
  const void* wpk;       /* packed weights (f32 or bf16 according to `prec`) */
  const float* bias;     /* [Cout] or NULL */
  const float* res;      /* [N,Ho,Wo,res_cs] residual added before the activation, or NULL */
  const float* a_scale;  /* [N,Cin] per-sample input-channel gate (squeeze-excite), or NULL */
  const float* row_mask; /* [N*Ho*Wo] multiplied into every output row after the activation, or NULL */
  float* out;            /* [N,Ho,Wo,out_cs], written at channel offset out_co */
  void* work;            /* CRESTE_ALGO_WINOGRAD / _WINOGRAD4: caller-owned workspace of creste_conv_wino[4]_workspace_bytes(); else NULL */
  int32_t N, H, W, Cin, in_cs;
  int32_t Ho, Wo, Cout, out_cs, out_co, res_cs;
  int32_t KH, KW, stride, pad_t, pad_l;
  int32_t act;  /* CRESTE_ACT_* */
  int32_t prec; /* CRESTE_PREC_* */
  int32_t algo; /* CRESTE_ALGO_* : how `wpk` was packed and which kernels run */
  int32_t flags; /* CRESTE_CONV_* bits, 0 by default */
  /* Dynamic-range bookkeeping of the fp16-split engine (CRESTE_PREC_F16X3); all three may be NULL otherwise.
   * a_amax   device float: an UPPER BOUND of max|in| over the slice read; the kernel scales
   *          the operand by a power of two so that the bound lands in [2^14, 2^15) -- exact, undone in the
   *          epilogue.  |a_scale| must be <= 1 (the squeeze-excite gate is a sigmoid).
   * out_amax device float, zero-initialised by the caller: atomically raised to max|out| by every
   *          workgroup (any precision) -- the next layer's a_amax without another pass over the tensor.
   * w_unscale [Cout] inverse of the per-output-channel power-of-two weight scale chosen by
   *          creste_conv_pack_weight (same call's `w_unscale` output). */
  const float* a_amax;
  float* out_amax;
  const float* w_unscale;
  /* CRESTE_ALGO_WINOGRAD4 only, or NULL: the conv's input is cat([in (Cin - up_C channels), bilinear_up2x(up_src)], C)
   * -- reference `Up.forward` (effnet.py:16-23: nn.Upsample(scale_factor=2, mode="bilinear", align_corners=False) + cat) and
   * DeconvHead.up2 (inpainting.py:81, no skip: `in` may be NULL when up_C == Cin) -- formed inside the input transform
   * with the expression of creste_upsample_concat_nhwc_f32 (bit-identical to materialising the tensor first).
   * up_src [N, up_H, up_W, up_cs] with H == 2 up_H, W == 2 up_W, up_C % 4 == 0. */
  const float* up_src;
  int32_t up_H, up_W, up_C, up_cs;
  /* Optional (training): per-channel partial sums of the values this call WRITES -- [rows][2][Cout] fp32, row r =
   * (sum x, sum x^2) over one workgroup's pixels, rows = creste_conv_stat_rows(d) -- so that the training-mode BatchNorm that
   * follows the conv (reference train_pefree.py / train_ssc.py: nn.Conv2d -> nn.BatchNorm2d) gets its batch statistics without
   * another pass over the tensor: creste_bn_train_forward_stats_f32.  Deterministic (one row per workgroup, no atomics).
   * Built for CRESTE_ALGO_WINOGRAD4 and for the stride-1 1x1 convs of the bf16 split modes; no residual / row mask. */
  float* out_stats;
} creste_conv_desc;

int creste_conv2d_nhwc(const creste_conv_desc* d, void* stream);
/* Three 1x1 convolutions with 128 output channels each, every one followed by a folded eval-mode BatchNorm and ReLU, as ONE launch:
 * the distillation head MultiLayerConv(kernels [1,1,1], dims [Cin, 128, 128, 128]) of reference creste/models/distillation.py:179
 * (blocks/conv.py:5-32).  out[p, out_co + n] = relu(W3 relu(W2 relu(W1 in[p] + b1) + b2) + b3): the hidden activations stay in the
 * accumulator registers of the wave that owns the pixel, split-operand bf16 products as creste_conv2d_nhwc (CRESTE_PREC_BF16X6 /
 * _BF16X3; Cin a multiple of 32).  wimg: the three weight matrices as MFMA operand tiles, [Cin / 16 + 16 steps][pieces][2][128][8] bf16 -- first layer in
 * channel order, layers two and three in the channel order of the accumulator layout (step 2t + j, k-octet h: channels
 * 32t + 16j + {4h + e, 8 + 4h + e}, e < 4); creste_conv1x1_chain3_weight_bytes(Cin, prec) bytes.  bias [3][128] fp32. */
int64_t creste_conv1x1_chain3_weight_bytes(int Cin, int prec);
int creste_conv1x1_chain3_f32(const float* in, int in_cs, int64_t P, int Cin, const void* wimg, const float* bias, int prec, float* out,
                              int out_cs, int out_co, void* stream);
/* Second half of `Upsample(x2, bilinear, align_corners=False) -> conv3x3(pad 1)` run as phase convolutions on the low-resolution
 * map (reference DeconvHead.up2, inpainting.py:56-60): after creste_conv2d_nhwc with CRESTE_CONV_REPLICATE_PAD | CRESTE_CONV_PHASE2X
 * and the composed 4 x Cout kernels, the outermost ring of `out` [N, 2H, 2W, out_cs] still contains the taps the high-resolution
 * conv's ZERO padding excludes and has no activation applied.  For every ring pixel and every tap whose source lies outside the
 * upsampled image this subtracts sum_ci w_ring[ky][kx][ci][co] * u~(source) (u~: the bilinear formula on the replicate-padded
 * x, exact-fp32 products) and applies `act`.  x [N, H, W, x_cs] is the conv's low-resolution input, w_ring [3][3][Cin][Cout] the
 * ORIGINAL 3x3 kernel (BatchNorm scale folded).  Cin % 64 == 0. */
int creste_upconv2x_ring_fix_f32(const float* x, int x_cs, int N, int H, int W, int Cin, const float* w_ring, int Cout, int act,
                                 float* out, int out_cs, int out_co, void* stream);
/* rows of creste_conv_desc.out_stats this call would write (host query; `out_stats` itself is not read), or -1 when the
 * kernel this descriptor dispatches to does not produce statistics */
int creste_conv_stat_rows(const creste_conv_desc* d);
/* 1 when (precision, kernel, stride) is built: F32 covers everything; BF16 / BF16X3 / BF16X6 cover stride-1 1x1 and
 * 3x3; F16X3 covers stride-1 1x1, 3x3, 5x5, 7x7 and stride-2 1x1, 3x3, 7x7 (every dense conv of the reference path:
 * effnet.py / inpainting.py / conv.py). */
int creste_conv_supported(int prec, int KH, int KW, int stride);
/* ---- Winograd F(2x2,3x3) path of the stride-1 3x3 convs (CRESTE_ALGO_WINOGRAD), same operator as above: the reference's
 * nn.Conv2d(k=3, padding=1) + BatchNorm + ReLU of effnet.py:16-23 (Up blocks), inpainting.py:52-68 (DeconvHead) and
 * depth.py:126.  2.25x fewer matrix-core products per output than the direct form, with the SAME fp32-equivalent
 * split-operand products (CRESTE_PREC_BF16X6) and fp32 accumulation: per 2x2 output tile and channel the 4x4 input
 * window d is transformed (V = B^T d B, +-1 coefficients, formed by the GEMM kernel's own loader), the 16 transform
 * positions are 16 independent [tiles x Cin] x [Cin x Cout] GEMMs against the pre-transformed weights U = G g G^T
 * (formed in float64 at pack time, BatchNorm scale folded), and the outputs Y = A^T M A get the usual epilogue
 * (bias + residual + activation + row mask + running |max|) in a second kernel that reads the fp32 products M from
 * `work`.  creste_conv_wino_supported: 1 when this shape / precision is built. */
int creste_conv_wino_supported(int prec, int KH, int KW, int stride, int Cin, int Cout);
int64_t creste_conv_wino_weight_bytes(int Cout, int Cin, int prec);
int creste_conv_wino_pack_weight(const float* w_oihw, const float* scale, void* wpk, int Cout, int Cin, int prec,
                                 void* stream);
/* bytes of `work` for an output of N x Ho x Wo x Cout: 16 positions x N*ceil(Ho/2)*ceil(Wo/2) tiles x Cout floats */
int64_t creste_conv_wino_workspace_bytes(int N, int Ho, int Wo, int Cout);
/* ---- Winograd F(4x4,3x3) path (CRESTE_ALGO_WINOGRAD4), same operator and reference call sites as above: 4x fewer
 * matrix-core products per output than the direct form (36 per 4x4 tile instead of 144), same fp32-equivalent
 * split-operand products and fp32 accumulation, transforms in fp32 (weights: float64).  Three launches: the input
 * transform writes V = B^T d B (6x6 windows at stride 4), already split into bf16 pieces and laid out as the GEMM's LDS
 * image, into `work`; the 36 per-position GEMMs stream V and the packed weights by LDS-DMA and write the fp32 products M
 * into `work`; the output transform Y = A^T M A applies the usual epilogue.  Relative rms error against float64 on the
 * 496-channel layers 1.3e-6 (F(2x2): 1.1e-6, direct: 1.8e-7).  `work`: creste_conv_wino4_workspace_bytes(), 16-byte
 * aligned. */
int creste_conv_wino4_supported(int prec, int KH, int KW, int stride, int Cin, int Cout);
int64_t creste_conv_wino4_weight_bytes(int Cout, int Cin, int prec);
int creste_conv_wino4_pack_weight(const float* w_oihw, const float* scale, void* wpk, int Cout, int Cin, int prec,
                                  void* stream);
int64_t creste_conv_wino4_workspace_bytes(int N, int Ho, int Wo, int Cin, int Cout, int prec);
/* Measurement aid: with the probe enabled every CRESTE_ALGO_WINOGRAD4 call records HIP events around its GEMM kernel on the
 * call's stream; creste_conv_wino4_gemm_last_ms waits for the last one (bench.py: roofline.gemm_kernel). */
int creste_conv_wino4_gemm_probe(int enable);
int creste_conv_wino4_gemm_last_ms(float* ms);
/* Size in BYTES of the packed weight for (Cout,Cin,KH,KW) at precision `prec`. */
int64_t creste_conv_packed_weight_bytes(int Cout, int Cin, int KH, int KW, int prec);
/* Pack a torch OIHW fp32 weight (device pointer, contiguous) into the GEMM layout, optionally
 * scaling output channel co by scale[co] (the folded BatchNorm gamma/sqrt(var+eps)). */
int creste_conv_pack_weight(const float* w_oihw, const float* scale, void* wpk, int Cout, int Cin,
                            int KH, int KW, int prec, void* stream);
/* CRESTE_PREC_F16X3 packing: as above, but every output channel is first scaled by the power of two that
 * brings max|w[co]*scale[co]| into [2^7, 2^8) (fp16 has 5 exponent bits: small BN-folded weights would
 * lose their low piece to subnormals); w_unscale[co] receives the inverse factor for the conv epilogue. */
int creste_conv_pack_weight_f16(const float* w_oihw, const float* scale, void* wpk, float* w_unscale, int Cout,
                                int Cin, int KH, int KW, void* stream);
/* amax[0] = max(amax[0], max |x|) over an NHWC slice ([pixels][cs] rows, C channels read per row): the
 * stand-alone producer of a_amax for tensors that did not come out of creste_conv2d_nhwc. */
int creste_absmax_nhwc_f32(const float* x, int64_t pixels, int C, int cs, float* amax, void* stream);

/* Depthwise KxK conv + bias + activation (EfficientNet MBConv `_depthwise_conv` + `_bn1` + swish,
 * third-party; static asymmetric "same" padding).  w is [KH*KW][C] (tap-major), BN pre-folded. */
int creste_dwconv2d_nhwc_f32(const float* in, const float* w, const float* bias, float* out, int N,
                             int H, int W, int C, int Ho, int Wo, int K, int stride, int pad_t,
                             int pad_l, int act, void* stream);

/* Squeeze-excite gate of an MBConv block: gate[n,c] = sigmoid(W2 * swish(W1 * mean_hw(x) + b1) + b2).
 * `partial` is caller workspace of at least N*creste_se_partial_count(H*W, C)*C floats; x may be NULL when
 * `partial` already holds the sums (creste_dwconv_se_nhwc_f32). */
int creste_se_partial_count(int HW, int C);   /* partial-sum rows per image for an HW x C tensor */
/* Depthwise conv + bias + activation that also leaves the squeeze-excite partial channel sums of its
 * output in `partial` ([N][creste_se_partial_count(Ho*Wo, C)][C]); follow with creste_se_gate_f32(x = NULL).
 * out_amax (optional, zero-initialised device float) is raised to max|out| as in creste_conv_desc. */
int creste_dwconv_se_nhwc_f32(const float* in, const float* w, const float* bias, float* out,
                              float* partial, float* out_amax, int N, int H, int W, int C, int Ho, int Wo,
                              int K, int stride, int pad_t, int pad_l, int act, void* stream);
/* The same operator (swish only) from an LDS tile (csrc/mbconv.hip) -- the deep MBConv blocks' many-channel / small-map
 * shapes; partial is [N][creste_dwconv_se_tile_partial_count(..)][C] -> creste_se_gate_partial_f32. */
int creste_dwconv_se_tile_partial_count(int Ho, int Wo, int C, int K, int stride);   /* < 0: not built */
int creste_dwconv_se_tile_f32(const float* in, const float* w, const float* bias, float* out, float* partial,
                              float* out_amax, int N, int H, int W, int C, int Ho, int Wo, int K, int stride,
                              int pad_t, int pad_l, void* stream);
/* plain depthwise conv (+ bias, may be NULL) (+ swish) on the same LDS-tile kernel: the training path's forward and,
 * with the taps flipped and pad' = K-1-pad, its stride-1 input gradient */
int creste_dwconv_tile_f32(const float* in, const float* w, const float* bias, float* out, int N, int H, int W, int C,
                           int Ho, int Wo, int K, int stride, int pad_t, int pad_l, int act, void* stream);
int creste_se_gate_f32(const float* x, float* partial, const float* w1, const float* b1,
                       const float* w2, const float* b2, float* gate, int N, int HW, int C, int Cse,
                       void* stream);
/* the gate from `nchunk` partial-sum rows per image that another kernel left in `partial` ([N][nchunk][C]) */
int creste_se_gate_partial_f32(const float* partial, int nchunk, const float* w1, const float* b1, const float* w2,
                               const float* b2, float* gate, int N, int HW, int C, int Cse, void* stream);

/* MBConv front half in one pass (csrc/mbconv.hip): out = swish(dw_KxK/stride(swish(x * w_expand + b_expand)) + b_dw)
 * -- the expand 1x1 conv, its folded BatchNorm + swish, the depthwise conv (zero padding of the EXPANDED map: pad_t /
 * pad_l, bottom / right implied by Ho, Wo), its folded BatchNorm + swish, and the squeeze-excite partial channel sums of
 * the output; the expanded tensor never reaches HBM.  Replaces, for the early EfficientNet-B0 blocks, the
 * `_expand_conv -> _bn0 -> swish -> _depthwise_conv -> _bn1 -> swish` chain of efficientnet_pytorch 0.7.1
 * MBConvBlock.forward (reference call site creste/models/blocks/effnet.py:83).  Exact fp32 FMAs in every precision
 * mode.  x: NHWC slice (channel stride x_cs); w_expand [Cin][Cexp], w_dw [K*K][Cexp] (both BN-folded), out dense
 * [N,Ho,Wo,Cexp]; partial: [N][creste_mbconv_partial_count(..)][Cexp] -> creste_se_gate_partial_f32.
 * Built for K 3|5, stride 1|2, Cin 16|24|40, Cexp <= 256 (multiple of 4); anything else is CRESTE_ERR_ARG. */
/* Encoder stem + the depthwise half of the first MBConv block in one pass: out = swish(dw3x3(swish(conv3x3/2(x) +
 * b_stem)) + b_dw) for the 4-channel NHWC RGB-D image x [N,H,W,4] (`_conv_stem`/`_bn0` then block 0's
 * `_depthwise_conv`/`_bn1`, reference effnet.py:41-44,83); w_stem [(ky*3+kx)*4+ci][C1], w_dw [9][C1] (BN-folded);
 * pad_t/pad_l: the stem's static 'same' padding, dpad_*: the depthwise conv's; out dense [N,H1,W1,C1]; partial:
 * [N][creste_stem_dw_partial_count(..)][C1] -> creste_se_gate_partial_f32. */
int creste_stem_dw_partial_count(int N, int H1, int W1, int C1);   /* < 0: not built */
int creste_stem_dw_f32(const float* x, int N, int H, int W, const float* w_stem, const float* b_stem, int pad_t,
                       int pad_l, const float* w_dw, const float* b_dw, int dpad_t, int dpad_l, float* out,
                       float* partial, float* out_amax, int C1, int H1, int W1, void* stream);
int creste_mbconv_partial_count(int N, int Ho, int Wo, int Cin, int Cexp, int K, int stride);   /* < 0: not built */
int creste_mbconv_expand_dw_f32(const float* x, int N, int H, int W, int Cin, int x_cs, const float* w_expand,
                                const float* b_expand, const float* w_dw, const float* b_dw, float* out,
                                float* partial, float* out_amax, int Cexp, int Ho, int Wo, int K, int stride,
                                int pad_t, int pad_l, void* stream);

/* out[..., 0:C2] = skip ; out[..., C2:C2+C1] = bilinear_upsample(x1) (align_corners=False, PyTorch
 * source-index rule src = rs*(dst+0.5)-0.5 clamped at 0).  reference effnet.py:25-28
 * (nn.Upsample + torch.cat) and inpainting.py:57-58.  skip may be NULL (C2 = 0).  out_amax: optional
 * zero-initialised device float raised to max|written values| (creste_conv_desc.out_amax). */
int creste_upsample_concat_nhwc_f32(const float* x1, int N, int H1, int W1, int C1, int x1_cs,
                                    const float* skip, int C2, int skip_cs, float* out, int Ho,
                                    int Wo, int out_cs, int out_co, float rh, float rw, float* out_amax,
                                    void* stream);

/* 2x2/2 max pool over rows [0, Ho) of the pooled map (F.max_pool2d; vin.py:104-109 crops the
 * pooled map to its first H//2 rows -> pass Ho = H//4). */
int creste_maxpool2_nhwc_f32(const float* in, int N, int H, int W, int C, int in_cs, float* out,
                             int Ho, int Wo, int out_cs, float* out_amax, void* stream);
/* The same with a ds x ds window and stride ds, ds in {1, 2, 4} (reward_cfg.ds of vin.py:104-106; ds = 1 is the plain
 * front-half crop: configurations whose MDP grid equals the BEV grid's front half). */
int creste_maxpool_nhwc_f32(const float* in, int N, int H, int W, int C, int in_cs, float* out,
                            int Ho, int Wo, int out_cs, int ds, float* out_amax, void* stream);

/* Bookkeeping launches, so that one inference forward consists of entry points of this library only (what
 * creste_hip_model_infer replays): dst[0..n) = value (32-bit words); *out = max(*a, *b) on device floats (the operand
 * bound of a concatenated tensor from its parts' bounds). */
int creste_fill_u32(void* dst, uint32_t value, int64_t n, void* stream);
int creste_max2_f32(const float* a, const float* b, float* out, void* stream);

/* `workgroups` single-wavefront workgroups that each do nothing for `microseconds` of the device's constant 100 MHz clock
 * (at most 100 000 us, 1 .. 1 000 000 workgroups): the probe with which the host finds out whether two of its streams
 * really run side by side -- HIP maps streams onto a few hardware queues (two streams on one queue run their kernels in
 * issue order) and the queues onto fewer dispatch pipes (a kernel waits while another queue of its pipe is still handing
 * out the workgroups of a large grid); creste_public_amd/ops.py: concurrent_stream.  The reference has no counterpart: it
 * runs one stream. */
int creste_spin_us(int microseconds, int workgroups, void* stream);

/* y = act(x*scale[c] + shift[c]): an eval-mode BatchNorm that FOLLOWS a ReLU (MultiScaleFCN trunk,
 * reference conv.py:118-128: conv -> ReLU -> BN -> ReLU) and so cannot be folded into the conv. */
int creste_affine_act_nhwc_f32(const float* x, int x_cs, const float* scale, const float* shift,
                               float* out, int out_cs, int64_t P, int C, int act, void* stream);

/* Bilinear resize of single-channel planes [N,H,W] into rows [0,Ho) of [N,Hd,Wo] planes
 * (F.interpolate(size=...), reference vin.py:121-125 `traversability_preds_full`). */
int creste_resize_plane_f32(const float* in, int N, int H, int W, float* out, int Ho, int Wo, int Hd,
                            float rh, float rw, void* stream);

/* layout changes at the module boundary (the reference's tensors are NCHW); out may be a channel
 * slice of a wider NHWC buffer (pixel stride out_cs). */
int creste_nchw_to_nhwc_f32(const float* in, float* out, int out_cs, int N, int C, int H, int W,
                            void* stream);
int creste_nhwc_to_nchw_f32(const float* in, int in_cs, float* out, int N, int C, int H, int W,
                            void* stream);

/* LiDAR scan -> sparse depth image (input preparation).  reference
 * creste/utils/projection.py:64-155 (pixels_to_depth): p_cam = lidar2cam[:3,:] @ [x,y,z,1] in float64,
 * pixel = trunc(clip(p_cam[:2]/p_cam[2])), keep z_cam > 0 and in-image, depth[v,u] = max (or min) over
 * the points of z_cam*scale, 0 where empty.  points [B,NP,point_stride>=3] fp32, lidar2cam [B][mat_stride>=12]
 * float64 row-major 3x4 (a 4x4 works with mat_stride 16), depth [B] planes of HxW at depth_batch_stride
 * elements (e.g. channel 3 of an NCHW rgbd tensor: stride 4*H*W). */
int creste_lidar_depth_image_f32(const float* points, int point_stride, const double* lidar2cam,
                                 int mat_stride, int B, int64_t NP, int H, int W, int reduce_min,
                                 double scale, float* depth, int64_t depth_batch_stride, void* stream);

/* The whole of the reference's `pixels_to_depth` for one scan, in its float64 (creste/utils/projection.py:64-155, every
 * `return_keys` entry :139-153): points [NP, point_stride>=3] fp32 or float64 (points_f64), lidar2cam row-major 3x4 / 4x4
 * float64 -> uv [NP,2] int32 = trunc(clip(xy/z, int32 range)) of EVERY point (:90-95; may be null), mask [NP] bytes =
 * z_cam > 0 and in-image (:97-104; may be null), reduced [H*W] float64 = max | min over the kept points of a pixel, 0 =
 * empty (torch_scatter.scatter, :124-131), last_write [H*W] fp32 = depth of the LAST kept point on the pixel (numpy fancy
 * assignment, :116-118; may be null).  work: 16*H*W bytes of scratch, 8-byte aligned. */
int creste_lidar_pixels_to_depth_f64(const void* points, int points_f64, int point_stride, const double* lidar2cam,
                                     int64_t NP, int H, int W, int reduce_min, int* uv, unsigned char* mask,
                                     double* reduced, float* last_write, void* work, void* stream);

/* Depth-bin logits -> metric depth + argmax bin.  reference depth.py:61-100,129-130 and
 * depth_utils.py:300-313: depth_m = sum_c softmax(logits)_c * bin_values[c] / 1000. */
int creste_depth_expectation_f32(const float* logits, int cs, int64_t P, int nbins,
                                 const float* bin_values, float* depth_m, int64_t* bins,
                                 void* stream);

/* pixel -> LiDAR point, range mask and z-embedding MLP.  reference splat_projection.py:37-49
 * (c=[u*d, v*d, d, 1]; xyz = p2p @ c as an fma chain k=0..3), :98-104,:152-157 (z MLP 1->2Z->Z,
 * ReLU,ReLU), :169 (min_bound <= xyz < max_bound).  zfeat is written at channel offset z_co of a
 * [B*P, z_cs] buffer. */
int creste_pixel_geometry_f32(const float* depth, const float* p2p, int B, int Hs, int Ws,
                              const float* bounds6, const float* w1, const float* b1,
                              const float* w2, const float* b2, int zhid, int zdim, float* xyz,
                              float* mask, float* zfeat, int z_cs, int z_co, void* stream);

/* The same launch + the first kernel of the splat's binning plan (reference splat_projection.py:185-187: map = lidar2map @
 * xyz, / voxel size): also writes bev_coords [B*P][2] and every point's base-cell key into the first B*P ints of `splat_work`
 * (creste_bev_splat_workspace_bytes(B, Hs*Ws, GH, GW) bytes); creste_bev_splat_plan_keyed_f32 finishes the plan.  Built for the
 * shipped 1 -> 64 -> 32 z-MLP (CRESTE_ERR_ARG otherwise: use creste_pixel_geometry_f32 + creste_bev_splat_plan_f32). */
int creste_pixel_geometry_keyed_f32(const float* depth, const float* p2p, int B, int Hs, int Ws,
                                    const float* bounds6, const float* w1, const float* b1, const float* w2,
                                    const float* b2, int zhid, int zdim, float* xyz, float* mask, float* zfeat,
                                    int z_cs, int z_co, float off_x, float off_y, float vox_x, float vox_y, int GH,
                                    int GW, float* coords, void* splat_work, void* stream);

/* Depth-guided camera->BEV bilinear voxel pooling (mean).  reference splat_projection.py:185-187
 * (lidar2map, /voxel), :293-352 (floor, 4 taps, scatter-add of weights and weighted features,
 * / clamp(density, 1)).  Gather formulation: points are binned by base cell, each BEV cell then
 * sums its (<=4 source cells') points -- no float atomics, every output cell written once.
 *   xyz     [B,P,3]   LiDAR points (from creste_pixel_geometry_f32)
 *   feats   [B,P,F]   fused, already range-masked features (pixel stride feats_cs)
 *   coords  [B,P,2]   OUT un-floored (X,Y) voxel coordinates (bev_coords)
 *   bev     [B,GH,GW,F] OUT, dens [B,GH,GW] OUT
 *   work    caller workspace, creste_bev_splat_workspace_bytes(B,P,GH,GW) bytes */
int64_t creste_bev_splat_workspace_bytes(int B, int P, int GH, int GW);
int creste_bev_splat_f32(const float* xyz, const float* feats, int feats_cs, int B, int P, int F,
                         float off_x, float off_y, float vox_x, float vox_y, int GH, int GW,
                         float min_weight, float* coords, float* bev, float* dens, void* work,
                         void* stream);
/* The same with the reference's `scatter_mode` (splat_projection.py:334-352): MEAN as above; SUM without the
 * normalisation; MAX = torch_scatter's scatter(..., reduce='max') of w*f per tap folded with torch.maximum
 * against the zero volume, i.e. max(0, max over taps and points of w*f).  dens is the tap-weight sum in all modes.
 * Several cameras (Camera2MapMulti.NC > 1, :227-234) are one call with P = NC*H*W: the reference concatenates the
 * cameras' points of a frame before the splat. */
#define CRESTE_SPLAT_MEAN 0
#define CRESTE_SPLAT_SUM 1
#define CRESTE_SPLAT_MAX 2
int creste_bev_splat_mode_f32(const float* xyz, const float* feats, int feats_cs, int B, int P, int F,
                              float off_x, float off_y, float vox_x, float vox_y, int GH, int GW,
                              float min_weight, int mode, float* coords, float* bev, float* dens, void* work,
                              void* stream);

/* creste_bev_splat_mode_f32 in its two halves.  The binning PLAN (voxel coordinates, base-cell keys, CSR of the cells,
 * per-cell records sorted by point id) depends on xyz alone (splat_projection.py:185-187 and the index half of :293-333), so
 * the model enqueues it right after the pixel geometry, ahead of the 288 -> 96 fusion conv that produces the features; the
 * GATHER then reads the plan from the same `work` (same B, P, GH, GW, same stream order) and writes bev / dens once. */
int creste_bev_splat_plan_f32(const float* xyz, int B, int P, float off_x, float off_y, float vox_x, float vox_y, int GH,
                              int GW, float* coords, void* work, void* stream);
int creste_bev_splat_plan_keyed_f32(int B, int P, int GH, int GW, const float* coords, void* work, void* stream);
int creste_bev_splat_gather_f32(const float* feats, int feats_cs, int B, int P, int F, int GH, int GW, float min_weight,
                                int mode, float* bev, float* dens, void* work, void* stream);

/* Value iteration on the 8-connected grid MDP.  reference vin.py:36-46 (transition kernel),
 * :48-80 (Jacobi sweeps, hard-max backup, batch-global convergence test, final q + softmax).
 *   r [B,H,W] -> v [B,H,W], q [B,8,H,W], policy [B,8,H,W]; *sweeps_out (device int) = sweep count.
 *   work: creste_value_iteration_workspace_bytes(B,H,W) bytes.
 * Asynchronous on `stream` like every other entry point: the whole iteration is ONE persistent launch that decides
 * convergence on the device (hipGraph-capturable).  A solve that hits max_sweeps therefore cannot come back as a return
 * code: it leaves *sweeps_out = -(sweeps run) and the last iterate in v / q / policy.  The persistent launch's workgroups exchange
 * their halos on the device, so they must all become resident.  Every wait is bounded: a launch that cannot get its workgroups
 * resident in time (the device held by long-running full-chip kernels of another stream) leaves *sweeps_out = INT32_MIN and invalid
 * outputs -- redo such a solve with creste_value_iteration_chunked_f32 (launch per chunk, no co-residency needed).  Beside the
 * conv kernels of this library's own backbone on another stream the persistent form was never seen to give up (0 of 276 solves,
 * scripts/irl_early.py): its workgroups take the CUs as the other kernels' workgroups retire.  Only grids too large for all their
 * 32 x 32 tiles to be resident at once (beyond about 2000 x 2000 cells per sample at batch 1) fall back to one launch per
 * chunk with host peeks, which can return CRESTE_ERR_NOCONV. */
int64_t creste_value_iteration_workspace_bytes(int B, int H, int W);
int creste_value_iteration_f32(const float* r, int B, int H, int W, float discount, float threshold,
                               int max_sweeps, float* v, float* q, float* policy, int32_t* sweeps_out,
                               void* work, void* stream);
/* The same solve in the launch-per-chunk form whatever the grid size (host-synchronous peeks, no co-residency needed): what a
 * caller retries with when the persistent launch reported *sweeps_out = INT32_MIN (its workgroups could not all become resident
 * beside another stream's kernels).  Same arguments, same results. */
int creste_value_iteration_chunked_f32(const float* r, int B, int H, int W, float discount, float threshold,
                                       int max_sweeps, float* v, float* q, float* policy, int32_t* sweeps_out,
                                       void* work, void* stream);

/* Expected state-visitation frequency + greedy rollout.  reference lfd.py:156-277,
 * train_utils.py:765-803.
 *   policy [B,8,H,W]; expert_xy [B,T_expert,2] (full-resolution BEV row,col as float); fov [H,W] u8.
 *   T = rollout horizon (action_horizon), T_expert = poses per expert trajectory: the start state is the
 *   earliest expert pose in the field of view, the terminal state the LAST expert pose (lfd.py:171-177) -- the
 *   reference indexes them independently of the horizon.
 *   -> exp_svf [B,H,W], state_preds [B,T,2] int64, state_grid [B,H,W].
 *   sharp_policy: workspace [B,8,H,W] floats (the temperature-sharpened policy). */
int creste_expected_svf_f32(const float* policy, const float* expert_xy, const uint8_t* fov, int B,
                            int H, int W, int T, int T_expert, float ds, float temperature, int sharpen,
                            int zero_terminal, float* sharp_policy, float* exp_svf,
                            int64_t* state_preds, float* state_grid, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Training-mode primitives (csrc/train.hip): what the IRL reward network (reference conv.py:88-161,
 * trained by train_traversability.py:66-105 through loss_utils.py:1118-1259) needs beyond the forward
 * engines.  NHWC fp32, explicit pixel strides; deterministic reductions.
 *
 * The gradient penalty's second-order term is evaluated as a TANGENT (JVP) forward followed by a
 * backward through the (primal, tangent) pair:  d/dtheta <u, grad_x R> = d/dtheta JVP_x R [u].
 * --------------------------------------------------------------------------------------------- */

/* gw[Cout][Cin][K][K] (torch OIHW) (+)= sum_pixels gy[p][co] * x[p + tap][ci] for a stride-1 conv with
 * pad = (K-1)/2 (F.conv2d backward w.r.t. weight).  work: creste_conv_wgrad_workspace_bytes(). */
int64_t creste_conv_wgrad_workspace_bytes(int N, int H, int W, int Cin, int Cout, int K);
int creste_conv_wgrad_f32(const float* x, int x_cs, const float* gy, int gy_cs, float* gw, int N, int H, int W,
                          int Cin, int Cout, int K, int pad, int accumulate, void* work, void* stream);
/* wt[Cin][cout_pad][K][K] = flipped, channel-transposed copy of w[Cout][Cin][K][K] (zero rows for
 * co >= Cout): packing `wt` with creste_conv_pack_weight and running creste_conv2d_nhwc on gy gives the
 * input gradient of the stride-1 conv (F.conv2d backward w.r.t. input). */
int creste_conv_flip_weight_f32(const float* w, float* wt, int Cout, int Cin, int K, int cout_pad, void* stream);

/* Training-mode BatchNorm2d over P = N*H*W pixels x C channels (C <= 256).
 *   forward : batch mean / biased variance -> mean, invstd = 1/sqrt(var+eps); running stats updated as
 *             nn.BatchNorm2d (momentum, unbiased variance; pass NULL to skip); y = gamma*xh+beta (+ReLU).
 *   tangent : yd = gamma*invstd*((xd - m(xd)) - xh*m(xh*xd)); mom_t[2][C] keeps the two moments.
 *   backward: cotangents gy (of y) and/or gyd (of yd) -> gx, gxd, g_gamma, g_beta ((+)= with
 *             accumulate); mom_b[5][C] scratch.  gyd == NULL is the ordinary first-order backward.
 *   out_amax / gx_amax (optional): device float raised to max|y| / max|gx| (zero it first) -- the operand bound
 *             the f16x3 convs consuming the tensor need (creste_conv_desc.a_amax), without a separate pass.
 *   work: creste_bn_workspace_bytes(C). */
int64_t creste_bn_workspace_bytes(int C);
/* creste_bn_train_forward_f32 without its statistics pass: mean / variance from the producing conv's partial sums
 * (creste_conv_desc.out_stats, [stat_rows][2][C]: summed per channel in float64, fixed order), then the same apply pass. */
int creste_bn_train_forward_stats_f32(const float* x, int x_cs, int64_t P, int C, const float* gamma, const float* beta,
                                      float eps, float momentum, float* running_mean, float* running_var, float* mean,
                                      float* invstd, float* var_scratch, float* y, int y_cs, int relu, float* out_amax,
                                      const float* stat_partial, int stat_rows, void* stream);
int creste_bn_train_forward_f32(const float* x, int x_cs, int64_t P, int C, const float* gamma, const float* beta,
                                float eps, float momentum, float* running_mean, float* running_var, float* mean,
                                float* invstd, float* var_scratch, float* y, int y_cs, int relu /* 0 none, 1 ReLU, 2 swish */,
                                float* out_amax, void* work, void* stream);
int creste_bn_train_tangent_f32(const float* x, int x_cs, const float* xd, int xd_cs, int64_t P, int C,
                                const float* gamma, const float* mean, const float* invstd, float* mom_t,
                                float* yd, int yd_cs, void* work, void* stream);
int creste_bn_train_backward_f32(const float* x, int x_cs, const float* xd, int xd_cs, const float* gy, int gy_cs,
                                 const float* gyd, int gyd_cs, int64_t P, int C, const float* gamma,
                                 const float* mean, const float* invstd, const float* mom_t, float* mom_b,
                                 float* gx, int gx_cs, float* gxd, int gxd_cs, float* g_gamma, float* g_beta,
                                 int accumulate, float* gx_amax, void* work, void* stream);
/* BatchNorm + fused ReLU backward: as above, with the cotangents masked by the ReLU's mask -- recomputed from x with
 * the forward's own expression (gamma * xh + beta > 0), so no separate mask pass reads y / writes a masked cotangent. */
int creste_bn_relu_train_backward_f32(const float* x, int x_cs, const float* xd, int xd_cs, const float* gy, int gy_cs,
                                 const float* gyd, int gyd_cs, int64_t P, int C, const float* gamma, const float* beta,
                                 const float* mean, const float* invstd, const float* mom_t, float* mom_b,
                                 float* gx, int gx_cs, float* gxd, int gxd_cs, float* g_gamma, float* g_beta,
                                 int accumulate, float* gx_amax, void* work, void* stream);
/* First-order backward of BatchNorm + a fused activation (act 1: ReLU, 2: swish = z * sigmoid(z); the forward is
 * creste_bn_train_forward_f32 with relu = act): gy is multiplied by act'(z), z = gamma * xh + beta recomputed from x.
 * Replaces efficientnet_pytorch's BatchNorm2d -> MemoryEfficientSwish pairs in training mode (reference call site
 * creste/models/blocks/effnet.py:83) without keeping z in memory. */
int creste_bn_act_train_backward_f32(int act, const float* x, int x_cs, const float* gy, int gy_cs, int64_t P, int C,
                                     const float* gamma, const float* beta, const float* mean, const float* invstd,
                                     float* mom_b, float* gx, int gx_cs, float* g_gamma, float* g_beta, int accumulate,
                                     float* gx_amax, void* work, void* stream);

/* op 0: o = max(a, 0) | op 1: o = a > 0 ? b : 0 (ReLU backward / tangent with a = the ReLU output) |
 * op 2: o = a + b.  [P][C] with pixel strides. */
int creste_pointwise2_f32(int op, const float* a, int a_cs, const float* b, int b_cs, float* o, int o_cs,
                          int64_t P, int C, void* stream);
/* 2x2/2 max-pool that also records the argmax (0..3 = dy*2+dx, first maximum wins as in ATen);
 * route: backward == 0 gathers `in` (full res) at idx -> pooled `out` (tangent); backward == 1 scatters
 * pooled `in` to the full-res `out` (zeros elsewhere).  H, W are the FULL-resolution extents. */
int creste_maxpool2_idx_f32(const float* in, int in_cs, int N, int H, int W, int C, float* out, int out_cs,
                            uint8_t* idx, void* stream);
int creste_maxpool2_route_f32(int backward, const float* in, int in_cs, const uint8_t* idx, float* out,
                              int out_cs, int N, int H, int W, int C, void* stream);
/* transpose of creste_upsample_concat's bilinear resize: gy [N,Ho,Wo,C] -> gx [N,H1,W1,C]. */
int creste_upsample_bwd_nhwc_f32(const float* gy, int gy_cs, int Ho, int Wo, float* gx, int gx_cs, int N, int H1,
                                 int W1, int C, float rh, float rw, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Encoder backward primitives and stage-1 distillation losses (csrc/train_backbone.hip): reference
 * train_pefree.py:71-99 -> distillation.py:145-207 -> blocks/effnet.py + efficientnet_pytorch MBConv.
 * --------------------------------------------------------------------------------------------- */

/* conv weight gradient for any stride / static asymmetric padding (stem, ResNet strided convs, heads):
 * gw[Cout][Cin][K][K] (+)= sum over output pixels q of gy[q][co] * x[q*stride + tap - pad][ci]. */
int64_t creste_conv_wgrad_strided_workspace_bytes(int N, int Ho, int Wo, int Cin, int Cout, int K);
int creste_conv_wgrad_strided_f32(const float* x, int x_cs, const float* gy, int gy_cs, float* gw, int N, int H,
                                  int W, int Ho, int Wo, int Cin, int Cout, int K, int stride, int pad_t, int pad_l,
                                  int accumulate, void* work, void* stream);
/* Same gradient with f16x3 operands (fp32 accumulation): x_amax / gy_amax are device floats bounding |x| and
 * |gy| (creste_absmax_nhwc_f32 or a tracked out_amax); same workspace. */
int creste_conv_wgrad_f16x3(const float* x, int x_cs, const float* gy, int gy_cs, float* gw, const float* x_amax,
                            const float* gy_amax, int N, int H, int W, int Ho, int Wo, int Cin, int Cout, int K,
                            int stride, int pad_t, int pad_l, int accumulate, void* work, void* stream);
/* Same gradient at the fp32-equivalent bf16x6 operand grade (three bf16 pieces per operand, six piece products on
 * the bf16 MFMA, fp32 accumulation; no operand bounds needed): the stride-1 "same" 3x3 convs with >= 64 channels on
 * both sides; any other shape runs creste_conv_wgrad_strided_f32.  Same workspace. */
int creste_conv_wgrad_bf16x6(const float* x, int x_cs, const float* gy, int gy_cs, float* gw, int N, int H, int W,
                             int Ho, int Wo, int Cin, int Cout, int K, int stride, int pad_t, int pad_l,
                             int accumulate, void* work, void* stream);
/* The same gradient through the F(4x4,3x3) transform (csrc/conv_wino4.hip): dU_p = sum over tiles of (A dY A^T)_p (x)
 * (B^T d B)_p per transform position, dW = G^T dU G -- 36 x SEG GEMMs over the tiles on the forward's GEMM kernel, 4x
 * fewer matrix products than the direct form at the same bf16x6 grade (fp32 transforms: rel. error ~3x the direct
 * kernel's).  Stride-1 same-size 3x3 convs, >= 128 channels on both sides; own workspace. */
int creste_conv_wgrad_wino4_supported(int K, int stride, int H, int W, int Ho, int Wo, int Cin, int Cout);
int64_t creste_conv_wgrad_wino4_workspace_bytes(int N, int H, int W, int Cin, int Cout);
int creste_conv_wgrad_wino4(const float* x, int x_cs, const float* gy, int gy_cs, float* gw, int N, int H, int W, int Cin,
                            int Cout, int pad_t, int pad_l, int accumulate, void* work, void* stream);
/* depthwise conv backward (weights tap-major [K*K][C] as in creste_dwconv2d_nhwc_f32): input gradient and
 * per-tap weight gradient gw_taps[K*K][C] (+)=. */
int creste_dwconv_dgrad_f32(const float* gy, const float* w, float* gx, int N, int H, int W, int C, int Ho, int Wo,
                            int K, int stride, int pad_t, int pad_l, void* stream);
int64_t creste_dwconv_wgrad_workspace_bytes(int C, int K);
int creste_dwconv_wgrad_f32(const float* x, const float* gy, float* gw_taps, int N, int H, int W, int C, int Ho,
                            int Wo, int K, int stride, int pad_t, int pad_l, int accumulate, void* work, void* stream);
/* op 0 swish(a) | 1 swish'(a)*b | 2 a*g[n][c] | 3 b + a*g[n][c] | 4 b*g[n][c] + r[n][c]/HW.
 * g is [N][g_c] with g_c == C (squeeze-excite gate) or 1 (per-sample scalar: drop-connect); out_amax (optional):
 * device float raised to max|o|. */
int creste_train_pointwise_f32(int op, const float* a, int a_cs, const float* b, int b_cs, const float* g, int g_c,
                               const float* r, float* o, int o_cs, int64_t HW, int64_t P, int C, float* out_amax,
                               void* stream);
/* out[n][c] = scale * sum_hw a[n,p,c] * (b ? b[n,p,c] : 1)  (squeeze-excite pool and its gate gradient). */
int64_t creste_sample_reduce_workspace_bytes(int N, int C);
int creste_sample_reduce_f32(const float* a, int a_cs, const float* b, int b_cs, float* out, int N, int64_t HW, int C,
                             float scale, void* work, void* stream);
/* squeeze-excite bottleneck: hpre = W1 s + b1, hact = swish(hpre), gate = sigmoid(W2 hact + b2); backward gives
 * gz = d/d(W2 hact + b2), ghpre = d/d hpre, gs = d/d s; weight gradients via creste_fc_wgrad_f32. */
int creste_se_fc_forward_f32(const float* s, const float* w1, const float* b1, const float* w2, const float* b2,
                             float* hpre, float* hact, float* gate, int N, int C, int Cse, void* stream);
int creste_se_fc_backward_f32(const float* gg, const float* gate, const float* hpre, const float* w1, const float* w2,
                              float* gz, float* ghpre, float* gs, int N, int C, int Cse, void* stream);
int creste_fc_wgrad_f32(const float* go, const float* in, float* gw, float* gb, int N, int O, int I, int accumulate,
                        void* stream);
/* CrossEntropyDepth (loss_utils.py:477-527, UD bins): out3 = {loss, accuracy, n_valid}; g_logits (optional) =
 * weight * d loss / d logits.  MSELoss (:606-647, +-inf labels masked): out2 = {loss, count}.
 * work: creste_loss_workspace_bytes(). */
int64_t creste_loss_workspace_bytes(void);
int creste_depth_ce_loss_f32(const float* logits, int cs, const float* gt_mm, int64_t P, int num_bins, float depth_min,
                             float depth_max, float weight, float* g_logits, int g_cs, float* out3, void* work,
                             void* stream);
int creste_mse_loss_f32(const float* pred, int p_cs, const float* gt, int g_cs, int64_t P, int C, float weight,
                        float* g_pred, int o_cs, float* out2, void* work, void* stream);

/* BEV-stage objectives, fused (csrc/losses.hip).  work: creste_loss_workspace_bytes().  Deterministic reductions.
 * creste_bev_ce_loss_f32 -- reference loss_utils.py:379-474 (`CrossEntropy`): pred NHWC [P][cs] with C classes, label
 *   from gt NCHW [B,Cg,H,W] (class_dim >= 0: the class index stored in that channel; < 0: argmax_c gt/(sum+eps)), only
 *   pixels with fov != 0, optional class weights and ignore_index (< -99999: none) as torch.nn.CrossEntropyLoss(weight,
 *   ignore_index, reduction='mean').  out4 = (loss, accuracy over labelled (!= 0) pixels, sum of weights, labelled
 *   count); g_pred [P][g_cs] = grad_scale * dloss/dpred (zeros outside the mask).
 * creste_smooth_l1_loss_f32 -- kind 0: elevation regression (:576-603): pred NHWC [P][cs] (2 channels), gt NCHW
 *   [B,2,H,W], channel 1 relative to channel 0 unless `absolute`, non-finite labels masked; kind 1: metric depth
 *   (:530-573): pred [P] metres, gt [P] millimetres, valid where the label falls into one of num_bins uniform bins of
 *   [depth_min, depth_max].  out2 = (mean smooth-L1 over valid elements, valid count); g_pred likewise. */
int creste_bev_ce_loss_f32(const float* pred, int cs, int C, const float* gt, int Cg, int64_t HW, int64_t P,
                           const uint8_t* fov, const float* class_weights, int class_dim, int ignore_index, float eps,
                           float grad_scale, float* g_pred, int g_cs, float* out4, void* work, void* stream);
int creste_smooth_l1_loss_f32(int kind, const float* pred, int cs, const float* gt, int64_t HW, int64_t P, int absolute,
                              float beta, float depth_min, float depth_max, int num_bins, float grad_scale,
                              float* g_pred, int g_cs, float* out2, void* work, void* stream);

/* Label bookkeeping of the supervised pixel-contrastive loss on the device (csrc/labels.hip; reference
 * creste/utils/utils.py:59-77 `remap_labels_in_batch`, creste/utils/train_utils.py:324-352 `extract_max_per_class`,
 * creste/utils/loss_utils.py:203-286).
 *   label_minmax : out2 = (min, max) of n int64 labels (sizes the presence table).
 *   remap_labels : gt [B,HW] int64 in [0,L) -> out: per sample, label -> (index in that sample's sorted unique labels)
 *                  + running offset (offset += present non-ignore labels), ignore_idx kept; table: B*(L+2) ints of work;
 *                  *nclass (device int) = largest new label + 1.
 *   group_by_class: the cells with label in [0,K), != ignore_idx and fov != 0 (fov may be NULL), grouped by class in
 *                  ascending class order, row-major order inside a class: counts [K], offsets [K+1], class_list [n]
 *                  (cell indices); work: creste_group_by_class_workspace_bytes(n, K).
 *   pick_cells   : cell[s] = class_list[offsets[sel_cls[s]] + sel_rank[s]] -- the host draws (class, rank) pairs with
 *                  the reference's generator (torch.randperm per over-full class, in class order).
 *   gather_rows / scatter_rows: rows [S][Z] <-> grid[cell[s]][0..Z) of an NHWC map with pixel stride cs (scatter into a
 *                  zero-filled gradient map; the cells of one pick are distinct). */
int creste_label_minmax_i64(const int64_t* labels, int64_t n, int64_t* out2, void* stream);
int creste_remap_labels_i64(const int64_t* gt, int B, int64_t HW, int ignore_idx, int L, int* table, int64_t* out,
                            int* nclass, void* stream);
int64_t creste_group_by_class_workspace_bytes(int64_t n, int K);
int creste_group_by_class_i64(const int64_t* labels, const uint8_t* fov, int64_t n, int K, int ignore_idx, int* counts,
                              int* offsets, int* class_list, void* work, void* stream);
int creste_pick_cells_i32(const int* class_list, const int* offsets, const int* sel_cls, const int* sel_rank, int S,
                          int* cell, void* stream);
int creste_gather_rows_f32(const float* grid, int cs, int Z, const int* cell, int64_t S, float* rows, void* stream);
int creste_scatter_rows_f32(const float* rows, const int* cell, int64_t S, int Z, float* grid, int cs, void* stream);

/* Backward of creste_bev_splat_f32 (reference autograd through splat_projection.py:262-354):
 * coords / bev / dens are the forward's outputs, feats its (range-masked) input.  g_feats [B*P][gf_cs] and
 * g_xyz [B*P][3] (LiDAR x, y; z gets 0) are gathers -- no atomics.  cell_work: B*GH*GW floats.  g_dens may be
 * NULL. */
int creste_bev_splat_bwd_f32(const float* coords, const float* feats, int feats_cs, const float* g_bev,
                             const float* g_dens, const float* bev, const float* dens, int B, int P, int F, int GH,
                             int GW, float vox_x, float vox_y, float min_weight, float* g_feats, int gf_cs,
                             float* g_xyz, float* cell_work, void* stream);
/* The same for every scatter mode of creste_bev_splat_mode_f32 (CRESTE_SPLAT_MEAN / _SUM / _MAX; splat_projection.py:
 * 334-352): 'max' routes a cell/channel cotangent to the entries that attain a positive maximum (w*f == bev, the
 * forward's own arithmetic); exact ties are a measure-zero case the reference leaves to torch_scatter's argmax. */
int creste_bev_splat_mode_bwd_f32(const float* coords, const float* feats, int feats_cs, const float* g_bev,
                                  const float* g_dens, const float* bev, const float* dens, int B, int P, int F, int GH,
                                  int GW, float vox_x, float vox_y, float min_weight, int mode, float* g_feats, int gf_cs,
                                  float* g_xyz, float* cell_work, void* stream);
/* Backward of creste_depth_expectation_f32: g_logits (+)= g_depth * softmax * (bin/1000 - depth). */
int creste_depth_expectation_bwd_f32(const float* logits, int cs, int64_t P, int C, const float* bin_values,
                                     const float* g_depth, float* g_logits, int g_cs, int accumulate, void* stream);

/* gz [N,(Ho-1)*s+1,(Wo-1)*s+1,C] = gy with s-1 zeros between neighbours: creste_conv2d_nhwc on gz with the
 * creste_conv_flip_weight_f32 kernel (stride 1, pad K-1-pad) is the input gradient of a stride-s conv. */
int creste_zero_insert_nhwc_f32(const float* gy, int gy_cs, float* gz, int N, int Ho, int Wo, int C, int stride,
                                void* stream);

/* Backward of creste_pixel_geometry_f32: cotangents of xyz [B*P][3] and of the z features (a slice of a wider
 * buffer, pixel stride gz_cs) -> g_depth [B*P]; gq [B*P][zdim], ghp [B*P][zhid], hbuf [B*P][zhid], zbuf [B*P] are the
 * per-pixel factors of the z-MLP's parameter gradients (W2: gq^T hbuf, w1: ghp^T zbuf, b2: sum gq, b1: sum ghp). */
int creste_pixel_geometry_bwd_f32(const float* depth, const float* p2p, int B, int Hs, int Ws, const float* w1,
                                  const float* b1, const float* w2, const float* b2, int zhid, int zdim,
                                  const float* g_xyz, const float* g_zf, int gz_cs, float* g_depth, float* gq,
                                  float* ghp, float* hbuf, float* zbuf, void* stream);

/* Multi-positive contrastive loss (reference models/losses/supcon_loss.py:56-115) without the N x M similarity
 * matrix: feats [N][D] local L2-normalised features, all_feats [M][D] the all-gathered ones (== feats on one rank),
 * int64 labels, optional per-row weights row_weights [N] (= class_weights[label]), self_offset = N * rank (the
 * column of row i's own sample).
 * forward: *loss = mean_i w_i * (logsumexp_{j != self} z_ij - mean_{j in P_i} z_ij), rows without positives count 0;
 * backward (same workspace, after forward): g_feats [N][D], g_all [M][D] = grad_scale * d loss / d (feats, all_feats).
 * D in {8,16,32,64}; D = 16 / 32 / 64 run on the matrix cores (csrc/supcon_mfma.hip: fp16 hi+lo operands rescaled by
 * exact powers of two from a device |max|, fp32 accumulation -- any finite feature magnitude).
 * work: creste_multipos_con_workspace_bytes(N, M, D). */
int64_t creste_multipos_con_workspace_bytes(int N, int M, int D);
int creste_multipos_con_forward_f32(const float* feats, const float* all_feats, const int64_t* labels,
                                    const int64_t* all_labels, const float* row_weights, int N, int M, int D,
                                    int self_offset, float temperature, float* loss, void* work, void* stream);
int creste_multipos_con_backward_f32(const float* feats, const float* all_feats, const int64_t* labels,
                                     const int64_t* all_labels, const float* row_weights, int N, int M, int D,
                                     int self_offset, float temperature, float grad_scale, void* work, float* g_feats,
                                     float* g_all, void* stream);

/* Candidate-trajectory scoring against a costmap (the consumer of the path's output).
 * reference loss_utils.py:1054-1116 (polyline rasterisation: max_steps = max over the call of ceil(segment length)
 * samples of torch.linspace(0,1,max_steps) per segment + the last pose, clamp, truncate, each cell once) and
 * :1197-1258 (trajectory reward = sum of the costmap over its visited cells).  Candidates: the Ackermann sampler of
 * scripts/traversability/planner_utils/control.py:12-118 (creste_public_amd/planner.py).
 *   xy [N,T,2] (row, col) in full-resolution BEV cells, divided by map_ds inside; costmap [.,H,W] with trajectory n
 *   reading map (map_index ? map_index[n] : n) at stride map_stride floats (0 = one shared map);
 *   -> scores [N]; visit [N,H,W] 0/1 (may be NULL); n_cells [N] visited-cell counts (may be NULL).  work: 1 int.
 *   costmap may be NULL when only the visitation maps are wanted (scores = 0). */
int creste_trajectory_scores_f32(const float* xy, int N, int T, float map_ds, int H, int W, const float* costmap,
                                 const int* map_index, int64_t map_stride, float* scores, float* visit,
                                 int* n_cells, int* work, void* stream);
/* The same for a BATCH of reference calls in one launch: group[n] in [0, n_groups) names the call trajectory n
 * belongs to; max_steps is taken per group (the expert set and every sample's counterfactual set of MaxEntIRLLoss,
 * loss_utils.py:1139,1160-1170).  work: n_groups ints. */
int creste_trajectory_scores_grouped_f32(const float* xy, int N, int T, float map_ds, int H, int W,
                                         const float* costmap, const int* map_index, int64_t map_stride,
                                         const int* group, int n_groups, float* scores, float* visit, int* n_cells,
                                         int* work, void* stream);

/* Visitation bookkeeping of MaxEntIRLLoss (reference loss_utils.py:1139-1186) in one launch: svf = normalise(fov *
 * visit_expert), policy_svf = normalise(fov * exp_svf_raw) (L1, +1e-5), and for samples with counterfactual maps
 * (visit_cf rows cf_ptr[b]..cf_ptr[b+1]): cf_total[b] = normalise(sum of those rows), exp_svf[b] = alpha * cf_total[b]
 * + (1 - alpha) * policy_svf[b]; otherwise exp_svf[b] = policy_svf[b], cf_total[b] = 0.  fov / visit_cf / cf_ptr may be
 * NULL.  All maps [B][HW] fp32. */
int creste_irl_visitation_mix_f32(const float* exp_svf_raw, const uint8_t* fov, const float* visit_expert,
                                  const float* visit_cf, const int* cf_ptr, float alpha, int B, int64_t HW, float* svf,
                                  float* exp_svf, float* cf_total, float* policy_svf, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Python-free deployment entry (csrc/plan_runtime.cpp).  reference scripts/runtime/compile.py:160-210 traces
 * TraversabilityModel(solve_mdp=False) / TerrainNet with torch.jit.trace and saves a self-contained module that the
 * C++ runtime stack loads without Python.  Here the artefact is a PLAN file written by
 * creste_public_amd.deploy.export_plan: the memory arena, the constant blocks (packed BN-folded weights; raw bytes, no
 * pickle) and the recorded sequence of entry-point calls of one forward at fixed shapes.
 *   load   : allocates the arena on the current device, uploads the constants.  flags & 1: replay through a hipGraph
 *            captured on the first infer call of a stream.
 *   infer  : inputs[i] = DEVICE pointer to input i (rgbd [B,1,4,H,W] fp32, p2p [B,1,4,4] fp32; contiguous) or NULL
 *            when the caller wrote it in place (creste_hip_model_input's pointer); asynchronous on `stream`.
 *   output : index -> name (the reference's output-dict key, terrainnet.py:272-350 / vin.py:119-133), device pointer
 *            into the arena, dtype (0 = f32, 1 = i64, 2 = u8/bool), rank, shape[6], stride[6] in ELEMENTS; valid after
 *            the stream drained, overwritten by the next infer.
 *   streams: a plan of a forward that the Python path pipelines (batches of >= 12 frames run as two half-batch forwards
 *            on two streams) carries every call's stream and the fork / join / buffer-ordering edges; the runtime replays
 *            them on side streams of its own (num_streams - 1 of them) behind `stream`, joined back before infer's last
 *            call is issued -- the caller still synchronises `stream` only.
 * Results are bit-identical to the Python host path (the same launches on the same arena layout). */
int creste_hip_model_load(const char* path, int flags, void** handle);
int creste_hip_model_free(void* handle);
const char* creste_hip_model_info(void* handle);
int creste_hip_model_num_inputs(void* handle);
int creste_hip_model_num_outputs(void* handle);
int creste_hip_model_num_streams(void* handle);
int creste_hip_model_input(void* handle, int index, const char** name, void** ptr, int* dtype, int* ndim,
                           int64_t* shape, int64_t* stride);
int creste_hip_model_output(void* handle, int index, const char** name, void** ptr, int* dtype, int* ndim,
                            int64_t* shape, int64_t* stride);
int creste_hip_model_infer(void* handle, const void* const* inputs, int n_inputs, void* stream);
/* tooling helper of the plan exporter: synchronous device -> host copy of raw bytes */
int creste_hip_memcpy_d2h(void* dst_host, const void* src_dev, int64_t nbytes);

#ifdef __cplusplus
}
#endif
#endif /* CRESTE_HIP_H */
