/*
 * hvr_hip -- C ABI of the MI355X (gfx950) kernels behind the HVRNet video-detection
 * forward path.  This is the drop-in boundary: every entry point below names the
 * reference interface it replaces (paths relative to the youthHan/HVRNet tree).
 *
 * Conventions (mirroring the reference's caller-allocates pybind ABI,
 * mmdet/ops/roi_align/src/roi_align_cuda.cpp:27-53, mmdet/ops/roi_align/roi_align.py:23):
 *   - every pointer is a DEVICE pointer owned by the caller unless marked "host";
 *     the library never allocates, frees or retains memory;
 *   - work is enqueued on `stream` (hipStream_t passed as void*); nothing synchronises;
 *   - return 0 on success, a negative HVR_E* code otherwise; hvr_last_error() gives the
 *     message of the calling thread's last failure (the reference printf()s "wrong roi
 *     size" and carries on, roi_align_cuda.cpp:39-42 -- this library never does that);
 *   - dtype: HVR_F32 = 0, HVR_BF16 = 1, HVR_F16 = 2, HVR_F16S = 3 (element type of activations /
 *     weights; all accumulation, softmax, box and score arithmetic is f32) -- the precision ladder:
 *       HVR_BF16  bf16 operands, the benchmark dtype BASELINE.json names (v_mfma_f32_16x16x32_bf16);
 *       HVR_F16   IEEE half operands: the same rate and bytes, 8 x finer mantissa, range 65504;
 *       HVR_F16S  "split half": a logical element x is stored as hi = half(x), lo = half(x - hi) (22 significant
 *                 bits for |x| >= 2^-3, absolute error < 2^-24 below) and a product is three half MFMAs -- hi*hi + hi*lo +
 *                 lo*hi -- accumulated in f32: f32-grade results at 1/3 of the half MFMA rate.  Memory layout of a row
 *                 of C elements (C % 32 == 0): C / 32 groups of [32 hi halves][32 lo halves] = 4 bytes per logical
 *                 element, both planes of a 32-element K-step in one 128-byte line; leading dimensions stay in logical
 *                 elements, bases are 128-byte aligned, column offsets multiples of 32.  hvr_cast converts to and from it.  Taken by hvr_gemm, hvr_conv2d_nhwc, hvr_relation_fwd,
 *                 hvr_relation_probs, hvr_transpose_pad and hvr_cast; the other entry points return HVR_EUNSUPPORTED;
 *       HVR_F32   exact f32 (v_mfma_f32_16x16x4_f32, 1/16 of the half rate): the parity mode.
 *     The reference computes in f32 (no fp16 key in configs/faster_rcnn_r101_{selsa,hrnmp}_c5.py); its optional
 *     mixed-precision islands are mmdet/core/fp16/decorators.py:9-160.
 *   - re-entrant; the only global mutable state is the one-time per-device kernel attribute setup (atomic, idempotent) and ONE
 *     process-wide tuning knob, hvr_rpn_wide_frames (an atomic int: which of two bit-identical kernel forms hvr_rpn_proposals
 *     launches; a captured graph keeps the form chosen at capture time).
 *   - environment switches, read once per process (round 5 retired the tuning knobs of rounds 1-4: tile-shape hints, slice sizes,
 *     group sizes, thresholds and the variants that measured slower are gone).  What is left selects between kernels that compute
 *     the same thing, for A/B runs and for the tests that compare them:
 *       HVR_BIGTILE=0         no 288 x 256 tiles (csrc/bigtile.hip): the tile engine's shapes everywhere (bit-identical)
 *       HVR_EXPAND=0          no row-panel expand kernel (csrc/expand.hip): tile engine / big tiles
 *       HVR_CONV3=0           no weights-resident 3x3 kernel for Cin = 64 (csrc/conv3x3.hip): tile engine (bit-identical)
 *       HVR_CONV_SPLITK       few-row products given a workspace (stream mode's one-frame launches): 0 = unsplit; 1 (default) = K
 *                             sliced across the waves of a workgroup where the shape allows (csrc/kpar.hip: one launch, no
 *                             partials), else across workgroups + a reduce launch; 2 = always the latter
 *       HVR_SPLIT_NORMALIZE   split-half relation: 0 = block weights folded into the apply pass, 1 = one normalising sweep + plain
 *                             product (default: by query-row count)
 *       HVR_REL_GROUPED       hvr_relation_fwd_grouped: 0 = one hvr_relation_fwd per group, 1 = grouped scores + per-group apply,
 *                             2 (default) = + the grouped apply launch
 *       HVR_NMS_MASK          (set) hvr_nms always takes the bit-mask kernels, never the greedy small-cap kernel
 *       HVR_RPN_WIDE=<frames> initial value of hvr_rpn_wide_frames
 *     and, in the Python host layer, HVR_RPN_SIDE=0 (RPN branch on the caller's stream instead of a side stream).
 */
#ifndef HVR_HIP_H_
#define HVR_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HVR_OK 0
#define HVR_EINVAL (-1)       /* bad argument (shape, alignment, null pointer) */
#define HVR_EUNSUPPORTED (-2) /* valid request outside the implemented envelope */
#define HVR_ELAUNCH (-3)      /* HIP reported a launch error */
#define HVR_EWORKSPACE (-4)   /* workspace too small */

#define HVR_F32 0
#define HVR_BF16 1
#define HVR_F16 2
#define HVR_F16S 3

#define HVR_LAYOUT_NCHW 0 /* reference layout */
#define HVR_LAYOUT_NHWC 1 /* native layout of this library */

int hvr_abi_version(void);   /* 2 since the descriptors of hvr_gemm / hvr_conv2d_nhwc grew the split-K scratch fields; 3: HVR_F16 / HVR_F16S; 4: hvr_tail_next_desc carries alpha / beta (split-half operands); 5: hvr_relation_fwd_grouped; 6: hvr_gemm_splitk_batched / hvr_unpack_conv_wgrads_multi */
const char* hvr_last_error(void);

/* ------------------------------------------------------------------------------------
 * Dense contraction with fused epilogue:  C[M][N] = act(A[M][K] . B[N][K]^T + bias + R)
 * Replaces the ATen calls behind nn.Linear / 1x1 Conv2d on the path:
 *   selsa_bbox_head.py:156,159,185-186,220-221,237 ; hrnmp_bbox_head.py:283,286,345-346,827-906.
 * K must be a multiple of 128 bytes / sizeof(elem); N a multiple of 4; rows 16-byte aligned.
 * staging: 0 = register-staged loads, 1 = direct global->LDS DMA.
 * ---------------------------------------------------------------------------------- */
typedef struct hvr_gemm_desc {
  const void* A; const void* B; void* C;
  int32_t M, N, K;
  int64_t lda, ldb, ldc;   /* in elements */
  const float* bias;       /* [N] f32 or NULL */
  const void* resid;       /* [M][ldr] (operand dtype) or NULL */
  int64_t ldr;
  int32_t relu;            /* apply max(x, 0) last */
  int32_t out_f32;         /* store C as f32 regardless of dtype */
  int32_t dtype;
  int32_t staging;
  int32_t tile_hint;       /* 0 = library picks the tile shape; k > 0 forces shape k-1 (tuning) */
  void* ws; size_t ws_bytes;   /* hvr_gemm: few-row split-K scratch (hvr_gemm_fewrow_workspace_bytes) or NULL / 0 */
  float alpha;             /* HVR_F16S only: C = act(alpha * (A . B^T) + bias + R), alpha a power of two (0 = 1).  A split-half
                              value below 2^-3 keeps an absolute, not a relative, error bound (its lo half is a subnormal), so
                              callers store small-magnitude operands scaled up -- this build's host layer keeps weights x 2^6 and
                              activations x 2^4 (hvr_cast_scaled) -- and pass the inverse here */
  float beta;              /* HVR_F16S only: factor on the bias (0 = 1): C = act(alpha * (A . B^T) + beta * bias + R) */
} hvr_gemm_desc;
/* Few-row products with an epilogue (one frame's 300 proposals through fc_new_1: 300 x 1024 x 12544 is 48 tiles of 128 x 64
 * walking 196 K-steps each): bytes of caller-owned scratch with which hvr_gemm cuts K into grid.y slices of f32 partial tiles
 * and applies bias / residual / ReLU in one reduce launch (bf16 operands and output).  0 = not split.  Without the scratch the
 * product runs unsplit (same result up to the f32 summation order).  The same rule as hvr_conv2d_splitk_workspace_bytes. */
size_t hvr_gemm_fewrow_workspace_bytes(const hvr_gemm_desc* d);
int hvr_gemm(const hvr_gemm_desc* d, void* stream);
/* The same product for outputs with few tiles and a long K (weight gradients: dW = dZ^T X has K = pixels): the K loop is cut
 * into slices that run as one launch, each writing an f32 partial into `ws`, and a second kernel sums them in a fixed order.
 * f32 output only (dtype f32, or out_f32 with bf16 operands), no bias / residual / ReLU.  The library picks the slice count
 * (1 = falls back to hvr_gemm's single pass; hvr_gemm_splitk_workspace_bytes then returns 0). */
size_t hvr_gemm_splitk_workspace_bytes(int M, int N, int K, int dtype);
int hvr_gemm_splitk(const hvr_gemm_desc* d, void* ws, size_t ws_bytes, void* stream);
/* `count` such products of ONE shape in one launch (round 6: the weight gradients of a ResNet stage's identical Bottlenecks -- the reference
 * differentiates each nn.Conv2d on its own, mmdet/models/backbones/resnet.py:220-266; at three frames per rank a single dW = dZ^T X leaves
 * most of the chip idle and costs three launches): d describes problem 0, problem g reads A + g * stride_a and B + g * stride_b ELEMENTS and
 * writes C + g * stride_c floats; f32 output, no epilogue.  The library cuts K into slices while count x tiles is below a round of the
 * chip; the sliced form needs the workspace below and one contiguous output (ldc = N, stride_c = M * N). */
size_t hvr_gemm_splitk_batched_workspace_bytes(int M, int N, int K, int dtype, int count);
int hvr_gemm_splitk_batched(const hvr_gemm_desc* d, int count, int64_t stride_a, int64_t stride_b, int64_t stride_c, void* ws, size_t ws_bytes,
                            void* stream);

/* ------------------------------------------------------------------------------------
 * NHWC convolution as implicit GEMM (frozen BatchNorm folded into w / bias by the caller)
 * with fused bias + residual + ReLU.  Replaces nn.Conv2d + BatchNorm2d(eval) + ReLU (+ the
 * Bottleneck residual add): mmdet/models/backbones/resnet.py:220-266,
 * mmdet/models/shared_heads/res_layer.py:67-74, mmdet/models/anchor_heads/rpn_head.py:30-35.
 *   x [B][H][W][Cin], w [Cout][KH][KW][Cin], y [B][OH][OW][Cout]; Cin % (128/sizeof elem) == 0.
 *   zero: >= 16 readable zero bytes on the device (source of padding taps).
 * ---------------------------------------------------------------------------------- */
typedef struct hvr_conv_desc {
  const void* x; const void* w; void* y;
  int32_t B, H, W, Cin, Cout, KH, KW, stride, pad, dil;
  const float* bias;
  const void* resid;       /* [B][OH][OW][Cout] or NULL */
  int32_t relu, out_f32, dtype, staging, tile_hint;
  const void* zero;
  void* ws; size_t ws_bytes;   /* split-K workspace (see hvr_conv2d_splitk_workspace_bytes) or NULL / 0 */
  float alpha, beta;       /* HVR_F16S only: factors on the accumulators / the bias, see hvr_gemm_desc (0 = 1) */
} hvr_conv_desc;
int hvr_conv2d_nhwc(const hvr_conv_desc* d, void* stream);
/* Few-row convolutions (one 600x1000 frame gives the stride-16 stages 2 394 output pixels: 76 tiles of 128 x 64 for 256 CUs,
 * each walking the whole K = 9 Cin loop alone -- the reference's steady-state loop, tools/test.py:214-250, runs the backbone
 * on ONE new frame per output frame): bytes of caller-owned scratch with which hvr_conv2d_nhwc cuts the K loop into slices
 * (grid.y), each slice writing an f32 partial tile, followed by one reduce launch that adds bias / residual, applies the
 * ReLU and rounds once to bf16.  0 = this descriptor is not split (enough tiles, short K, f32, or a dedicated kernel takes it);
 * the call then ignores ws.  With ws == NULL or ws_bytes smaller than this the conv runs unsplit (same result up to the f32
 * summation order of the K slices).  Shapes with a short K loop (K <= 2 560, at most 320 tiles of 64 x 64) do not use the
 * workspace: their K-steps are dealt to the four waves of one workgroup per tile and summed in the LDS (csrc/kpar.hip) --
 * same contract, one launch.  HVR_CONV_SPLITK=0 in the environment turns both forms off, 2 the second one. */
size_t hvr_conv2d_splitk_workspace_bytes(const hvr_conv_desc* d);
/* Which kernel hvr_conv2d_nhwc would run for this descriptor, without launching anything (pointers are only inspected
 * for alignment): 0 = MFMA tile engine (gemm.hip), 1 = row-panel kernel for the Bottleneck's channel-expanding 1x1 conv
 * + residual (expand.hip: bf16, 1x1 stride 1, Cin = 64 / 128 / 256, a residual, Cout >= 2 Cin in whole 64-channel chunks,
 * >= 128 output pixels; tile_hint 13 forces it where it applies, HVR_EXPAND=0 in the environment turns the automatic
 * choice off), 2 = persistent 3x3 kernel with LDS-resident weights for bf16 64 -> 64 channels, stride 1, pad 1, no
 * residual (conv3x3.hip: layer 1's conv2; HVR_CONV3=0 turns it off; any tile_hint keeps the tile engine),
 * 3 = big-tile kernel (bigtile.hip: 288 x 256 output tiles, one workgroup per CU; bf16, bias / ReLU epilogue, Cout a multiple
 * of 256, K >= 512 -- taken by default when its grid covers most of the chip (>= 192 tiles: res5's and the RPN's convs on a
 * 15-frame batch), and from 96 tiles on with tile_hint 16, the hint of a caller that keeps several windows in flight and wants
 * CU-time rather than latency; bit-identical to the tile engine; HVR_BIGTILE=0 turns it off);
 * tile_hint 18 (round 6): TWO-LEVEL accumulation for exact-f32 / split-half operands whose K is a multiple of 256 elements -- every 8 K-steps'
 * products sum in a block accumulator that is then added to the running total (tile engine, 128 x 128 shape), so a long-K conv's f32 rounding noise
 * stops growing with the number of MFMAs chained on one accumulator (the RPN's 3x3 conv, K = 9 216: its share of the final boxes' distance to the
 * f64 reference halves, profiles/r06_noise_two_level.txt); other formats / shapes run as with hint 0;
 * negative = the descriptor would be rejected. */
int hvr_conv2d_path(const hvr_conv_desc* d);

/* Closing 1x1 conv of a Bottleneck whose identity path is a projection -- the first block of every stage
 * (mmdet/models/backbones/resnet.py:248-264 with `downsample`, built at resnet.py:283-296):
 *     y = relu( h W3^T + x_s Wd^T + bias )
 * in ONE pass: the downsample conv (1x1, stride s) becomes a second K segment of the expand product, so its [B][OH][OW][Cout]
 * output is neither written nor re-read as a residual.  h [B][OH][OW][C1] = conv2's output, x [B][H2][W2][C2] = the block
 * input, x_s its pixels (oy * s, ox * s); w [Cout][C1 + C2] = [W3 | Wd] rows with both BatchNorm scales folded in;
 * bias [Cout] = shift3 + shiftd (f32).  bf16; C1 + C2 in {128, 384} (stages 1 and 2 of the R-101: 64 + 64, 128 + 256) on the
 * row-panel kernel, any other C1, C2 in whole 64-channel K-steps (stage 3: 256 + 512 at stride 2, res5: 512 + 1024) on the tile engine,
 * Cout a multiple of 128.  hvr_bottleneck_tail_supported: 1 when this descriptor runs, 0 otherwise (the caller then issues the
 * two convs separately through hvr_conv2d_nhwc). */
typedef struct hvr_tail_desc {
  const void* h; const void* x; const void* w; void* y;
  int32_t B, OH, OW, C1, H2, W2, C2, stride2, Cout;
  const float* bias;
  int32_t relu, dtype;
} hvr_tail_desc;
int hvr_bottleneck_tail(const hvr_tail_desc* d, void* stream);
int hvr_bottleneck_tail_supported(const hvr_tail_desc* d);

/* The same closing 1x1 AND the next block's opening (reducing) 1x1 + bn1 + ReLU (resnet.py:224-232 of block i + 1) in one
 * pass over the pixels: the workgroup that produced all Cout channels of a pixel multiplies them by wn while they are still
 * in registers, so the next block's conv1 never re-reads y.
 *   y  = relu(h W3^T [+ x_s Wd^T] + bias [+ resid])          written as before (the next block's residual / shortcut input)
 *   hn = relu(y wn^T + bias_n)                                [B][OH][OW][Cn]
 * tail.C2 == 0 (tail.x ignored): an identity block, resid [B][OH][OW][Cout] is its input map; tail.C2 > 0: resid must be
 * NULL.  wn [Cn][Cout] bf16, bias_n f32 [Cn].  (Cout, Cn) in {(256, 64), (512, 128)}: stages 1 and 2 of the R-101;
 * hn is computed from the bf16-rounded y, exactly as a separate conv would read it. */
typedef struct hvr_tail_next_desc {
  hvr_tail_desc tail;
  const void* resid;
  const void* wn; const float* bias_n; void* hn;
  int32_t Cn;
  float alpha, beta;   /* HVR_F16S only (identity form, tail.C2 == 0, (Cout, Cn) = (256, 64)): the factors of hvr_gemm_desc on BOTH
                          products -- y = relu(alpha h W3^T + beta bias + resid), hn = relu(alpha y wn^T + beta bias_n); 0 = 1 */
} hvr_tail_next_desc;
int hvr_bottleneck_tail_next(const hvr_tail_next_desc* d, void* stream);
int hvr_bottleneck_tail_next_supported(const hvr_tail_next_desc* d);

/* 7x7/2 stem: gathers img (NCHW f32, the reference's input layout, resnet.py:522-524) into
 * patch rows [B*OH*OW][KP] with k = (ky*7+kx)*3 + c and zeros for k >= 147 (KP = 192). */
int hvr_im2col_stem(const float* img, void* cols, int B, int H, int W, int KP, int dtype, void* stream);
/* Fused bf16 stem: conv1 7x7/2 (BN folded) + ReLU + MaxPool2d(3,2,1) (resnet.py:456-466,522-526) in one kernel.
 * img NCHW f32 [B][3][H][W]; wpk bf16 [64][7][32] with wpk[n][ky][kx*4+c] = w[n][c][ky][kx] (zeros for c = 3 or
 * kx = 7); bias f32 [64]; out bf16 NHWC [B][PH][PW][64], PH = ((H-1)/2+1 - 1)/2 + 1. */
int hvr_stem_fused(const float* img, const void* wpk, const float* bias, void* out, int B, int H, int W, void* stream);
/* the same in the other half formats: dtype HVR_BF16 / HVR_F16 (wpk and out in that format) or HVR_F16S -- split half: wpk is
 * [2][64][7][32] half of the weights x 2^6 (the kernel takes the factor back): plane 0 = half(64 w), plane 1 = half(64 w - plane 0); out [B][PH][PW][64] in the split-half layout */
int hvr_stem_fused_dtype(const float* img, const void* wpk, const float* bias, void* out, int B, int H, int W, int dtype, void* stream);
/* nn.MaxPool2d(3, 2, 1) on NHWC (resnet.py:466,526) */
int hvr_maxpool3x3s2_nhwc(const void* x, void* y, int B, int H, int W, int C, int dtype, void* stream);

/* ------------------------------------------------------------------------------------
 * Relation core:  O = softmax(scale * Q K^T, over keys) V      (single head, D up to any
 * multiple of the K-step).  Replaces torch.bmm / scalar mul / nn.Softmax / torch.mm in
 * selsa_bbox_head.py:166-182 and hrnmp_bbox_head.py:293-342 without materialising the
 * f32 Mq x Mk logits.  Q [Mq][ldq], K [Mk][ldk], V [Mk][ldv], O [Mq][ldo], all `dtype`.
 * ---------------------------------------------------------------------------------- */
size_t hvr_relation_workspace_bytes(int Mq, int Mk, int D, int dtype);
int hvr_relation_fwd(const void* Q, int64_t ldq, const void* K, int64_t ldk, const void* V, int64_t ldv,
                     void* O, int64_t ldo, int Mq, int Mk, int D, float scale, int dtype, int staging,
                     void* ws, size_t ws_bytes, void* stream);

/* The same for `groups` INDEPENDENT problems of one shape in one call -- the clips (windows) a caller has in flight, each with its
 * own Q / K / V / O: group g's operands start gs* ELEMENTS behind group 0's (a batched head keeps the groups' rows back to back in
 * one matrix: gs = rows x ld).  The reference runs one clip at a time (tools/test.py:214-250 -> hnmb_rcnn.py:195-222), so a stage of
 * W clips is W calls of the lines above; here the 352 x 256 score tiles of all groups are one list walked by persistent workgroups
 * (csrc/relation_bt.hip: HVR_BF16, HVR_F16, HVR_F16S) and, for those formats and >= 3 window-sized groups, the apply pass is one
 * launch of 288 x 256 tiles over the whole key axis (csrc/relation_apply_bt.hip).  Per group the result is hvr_relation_fwd's up to
 * the association of the f32 sums -- split half: and up to the half rounding of block-weighted probabilities below 2^-26 of a row's
 * largest -- (exact != 0: bit for bit: the scores launch stays grouped, the apply pass runs per group); shapes the grouped kernels
 * do not take run as `groups` hvr_relation_fwd calls.  Workspace: hvr_relation_grouped_workspace_bytes. */
size_t hvr_relation_grouped_workspace_bytes(int groups, int Mq, int Mk, int D, int dtype);
int hvr_relation_fwd_grouped(const void* Q, int64_t ldq, int64_t gsq, const void* K, int64_t ldk, int64_t gsk,
                             const void* V, int64_t ldv, int64_t gsv, void* O, int64_t ldo, int64_t gso, int groups,
                             int Mq, int Mk, int D, float scale, int dtype, int staging, int exact, void* ws, size_t ws_bytes, void* stream);

/* Backward of the relation core (training path; the reference gets it from autograd through
 * torch.bmm / nn.Softmax / torch.mm, selsa_bbox_head.py:166-182).  Two pieces that are not plain GEMMs:
 *   hvr_relation_probs : P = softmax(scale * Q K^T) as a [Mq][ldp] matrix in `dtype` (ldp = keys padded to 128, padding
 *                        columns zero), the f32 logits never materialised;
 *   hvr_relation_dscore: dS = scale * P * (dP - rowsum(dO * O))  (softmax backward, logit scale folded in).
 * With them  dV = P^T dO,  dP = dO V^T,  dQ = dS K,  dK = dS^T Q  are hvr_gemm / hvr_transpose_pad calls
 * (hvrnet_amd/ops.py: RelationFunction). */
size_t hvr_relation_probs_workspace_bytes(int Mq, int Mk);
int hvr_relation_probs(const void* Q, int64_t ldq, const void* K, int64_t ldk, void* P, int64_t ldp, int Mq, int Mk, int D,
                       float scale, int dtype, int staging, void* ws, size_t ws_bytes, void* stream);
int hvr_relation_dscore(const void* P, const void* dP, const void* dO, int64_t ldgo, const void* O, int64_t ldo, void* dS,
                        int Mq, int64_t ldp, int D, float scale, int dtype, void* stream);

/* Head training helpers.  The reference gets all of these from autograd:
 *   hvr_relu_bwd : dZ = dY where Y > 0 (ReLU fused into a GEMM epilogue; Y is that epilogue's output);
 *   hvr_colsum   : db[n] = sum_m dY[m][n]  (bias gradient, f32);
 *   hvr_det_loss : BBoxHead.loss for class-agnostic boxes (mmdet/models/bbox_heads/bbox_head.py:100-130 with
 *                  losses/cross_entropy_loss.py:9-20, losses/smooth_l1_loss.py:9-18, losses/accuracy.py:4-21):
 *                  out3 = (loss_cls, loss_bbox, acc) and dlogits = d(w_cls*loss_cls + w_bbox*loss_bbox)/d logits for a
 *                  logit matrix [R][ldl] holding ncls class logits at cls_off and 4 box deltas at reg_off. */
int hvr_relu_bwd(const void* dY, const void* Y, void* dZ, int64_t n, int dtype, void* stream);
size_t hvr_colsum_workspace_bytes(int M, int N);
int hvr_colsum(const void* dY, float* db, int M, int N, int64_t ld, int dtype, void* ws, size_t ws_bytes, void* stream);
int hvr_det_loss(const float* logits, int ldl, int cls_off, int reg_off, int ncls, const int64_t* labels, const float* label_weights,
                 const float* bbox_targets, const float* bbox_weights, int R, float beta, float w_cls, float w_bbox, float* out3,
                 float* dlogits, void* stream);

/* Conv backward helpers (training path; the reference differentiates nn.Conv2d + frozen BatchNorm through autograd,
 * mmdet/models/backbones/resnet.py:220-266).  Input and weight gradients are tile-engine products:
 *   dX   = hvr_conv2d_nhwc(dZ, weights rotated 180 degrees with the channel axes swapped)      (stride-1 KxK convs)
 *   dW^T = hvr_gemm(dZ^T, cols^T)   with cols = hvr_im2col_nhwc(X): [B*OH*OW][KH*KW*Cin], stride 1, zero padding
 * hvr_scale_rows multiplies row r of a [R][C] matrix by scale[r]: the frozen BatchNorm scale folded into the weights on
 * the way in and into their gradient on the way out. */
int hvr_im2col_nhwc(const void* x, void* cols, int B, int H, int W, int Cin, int KH, int KW, int pad, int dil, int dtype, void* stream);
/* The K-contiguous operands of the weight-gradient product in one pass each (bf16 / half; round 5): colsT [KH*KW*Cin][ldt] = the
 * TRANSPOSED patch matrix (row tap*Cin + c holds that input channel of every output pixel, columns B*OH*OW .. ldt-1 zero), and
 * dZ = dY where Y > 0 together with dZt [C][ldt] = dZ^T -- instead of hvr_im2col_nhwc / hvr_relu_bwd each followed by
 * hvr_transpose_pad. */
int hvr_im2col_t(const void* x, void* colsT, int64_t ldt, int B, int H, int W, int Cin, int KH, int KW, int pad, int dil, int dtype, void* stream);
int hvr_relu_bwd_t(const void* dY, const void* Y, void* dZ, void* dZt, int64_t ldt, int R, int C, int dtype, void* stream);
int hvr_scale_rows(const void* w, const float* scale, void* out, int R, int64_t C, int dtype, void* stream);
/* The two layout + scale passes of a trainable conv in one kernel each: the f32 master weight [Cout][Cin][KH][KW] times the
 * frozen BatchNorm scale, permuted to the conv kernel's [Cout][KH][KW][Cin] and rounded to the compute dtype; and the way back
 * for the f32 weight gradient (accumulate != 0: added to `out`, i.e. written straight into the parameter's gradient buffer). */
int hvr_pack_conv_weight(const float* w, const float* scale, void* out, int Cout, int Cin, int KH, int KW, int out_dtype, void* stream);
int hvr_unpack_conv_wgrad(const float* dw, const float* scale, float* out, int Cout, int Cin, int KH, int KW, int accumulate, void* stream);
/* The weight side of a whole training iteration in two launches (round 5): `items_dev` is a table IN DEVICE MEMORY, one entry per trainable
 * conv / linear layer (pointers are stable: the parameters live in one flat buffer, dist_train.FlatParams).  hvr_pack_conv_weights_multi is
 * hvr_pack_conv_weight for every entry (first = index of the entry's first element in the concatenated work list, ascending; total = their
 * sum); hvr_transpose_multi writes dst[c][r] = src[r][c] for every entry as 64 x 64 tiles of 16-bit words (C % 8 == 0, ldd and dcols % 8 == 0,
 * columns R .. dcols - 1 of dst zero; first_tile ascending, tiles = their sum): the transposed (1x1 / linear) and the rotated (KxK, one entry per
 * filter tap) operands of the input-gradient products, from the packed weights. */
typedef struct { const float* w; const float* scale; void* out; int64_t first; int32_t Cout, Cin, KK, pad_; } hvr_pack_item;
typedef struct { const void* src; void* dst; int64_t lds, ldd; int32_t R, C, first_tile, tiles_c, dcols, pad_; } hvr_transpose_item;   /* dcols: columns of dst written per row (R .. dcols - 1: zeros) */
int hvr_pack_conv_weights_multi(const hvr_pack_item* items_dev, int n, int64_t total, int out_dtype, void* stream);
/* hvr_unpack_conv_wgrad for a table of layers in one launch (the outputs of hvr_gemm_splitk_batched): entry i's f32 product dw [Cout][KK][Cin]
 * times scale -> the parameter-layout gradient out [Cout][Cin][KK], added to it when accumulate != 0; first / total as above. */
typedef struct { float* out; const float* scale; const float* dw; int64_t first; int32_t Cout, Cin, KK, pad_; } hvr_unpack_item;
int hvr_unpack_conv_wgrads_multi(const hvr_unpack_item* items_dev, int n, int64_t total, int accumulate, void* stream);
int hvr_transpose_multi(const hvr_transpose_item* items_dev, int n, int tiles, void* stream);

/* Optimizer step on a flat f32 buffer (the training configs: SGD lr 5e-4, momentum 0.9, weight decay 1e-4, gradient
 * clipping max_norm 35, configs/faster_rcnn_r101_selsa_c5.py:215-222; torch.optim.SGD semantics, dampening 0).  The two
 * scalings the reference applies to the summed gradient first are folded in: grad_scale = 1 / world_size
 * (mmdet/core/utils/dist_utils.py:24-25) and the clip_grad_norm_ coefficient min(1, max_norm / (norm + 1e-6)), norm taken
 * over the scaled gradient, on the device (max_norm <= 0: no clipping).  first_step: the momentum buffer is initialised
 * with the first update, as torch does. */
size_t hvr_sgd_workspace_bytes(void);
int hvr_sgd_step(float* param, const float* grad, float* momentum_buf, int64_t n, float lr, float momentum, float weight_decay,
                 float grad_scale, float max_norm, float* ws, size_t ws_bytes, int first_step, void* stream);

/* ------------------------------------------------------------------------------------
 * RoIAlign (legacy "+1" convention).  Replaces roi_align_cuda.forward / .backward:
 *   mmdet/ops/roi_align/src/roi_align_cuda.cpp:27-80, roi_align_kernel.cu:63-141,187-282.
 * rois [K][5] f32 = (batch_idx, x1, y1, x2, y2).  layout NCHW: feat [B][C][H][W] ->
 * out [K][C][PH][PW]; layout NHWC: feat [B][H][W][C] -> out [K][PH][PW][C].
 * Backward is f32 only (as the reference) and ACCUMULATES into grad_feat (caller zeroes).
 * ---------------------------------------------------------------------------------- */
int hvr_roi_align_fwd(const void* feat, const float* rois, void* out, int B, int C, int H, int W, int K,
                      int PH, int PW, float spatial_scale, int sample_num, int dtype, int layout, void* stream);
int hvr_roi_align_bwd(const float* grad_out, const float* rois, float* grad_feat, int B, int C, int H, int W,
                      int K, int PH, int PW, float spatial_scale, int sample_num, int layout, void* stream);

/* ------------------------------------------------------------------------------------
 * Greedy NMS.  Replaces nms_cpu.nms / nms_cuda.nms (mmdet/ops/nms/src/nms_cpu.cpp:5-71,
 * nms_kernel.cu:24-136) with no host round trip.  dets [n][5] f32 (x1,y1,x2,y2,score);
 * ge_semantics != 0 suppresses at IoU >= thr (the CPU reference), 0 at IoU > thr (the CUDA
 * reference).  keep [n] int64 receives the surviving indices in ascending input order,
 * *n_keep their count (both device memory).  n <= 8192.
 * ---------------------------------------------------------------------------------- */
size_t hvr_nms_workspace_bytes(int n);
int hvr_nms(const float* dets, int n, float thr, int ge_semantics, int64_t* keep, int32_t* n_keep,
            void* ws, size_t ws_bytes, void* stream);
/* The same, stopping after the first max_keep survivors in score order: `nms(proposals, thr)[:nms_post]` of
 * mmdet/models/anchor_heads/rpn_head.py:95-97 without pricing the IoUs of boxes behind the cut.  keep: ascending input
 * order of those survivors.  One workgroup that evaluates only the rows of surviving boxes (max_keep x n IoUs, not n^2/2);
 * max_keep <= 1024, larger caps take the mask + sweep path of hvr_nms and stop the sweep at the cap. */
int hvr_nms_first(const float* dets, int n, float thr, int ge_semantics, int max_keep, int64_t* keep, int32_t* n_keep,
                  void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------
 * RPN proposal generation for T frames in one call.  Replaces RPNHead.get_bboxes_single
 * (mmdet/models/anchor_heads/rpn_head.py:55-104) + AnchorGenerator.grid_anchors
 * (mmdet/core/anchor/anchor_generator.py:66-83) + delta2bbox (core/bbox/transforms.py:78-110):
 * sigmoid, top nms_pre, decode + clip, NMS(thr, >=), first nms_post, top max_num by score.
 *   cls [T][H][W][A] f32 logits, reg [T][H][W][A*4] f32 (NHWC conv outputs),
 *   base_anchors host [A][4] f32, means/stds host [4] f32.
 *   proposals [T][max_num][5] f32, counts [T] int32 (rows beyond counts[t] are untouched).
 * ---------------------------------------------------------------------------------- */
typedef struct hvr_rpn_desc {
  const float* cls; const float* reg;
  int32_t T, H, W, A, anchor_stride;
  const float* base_anchors; const float* means; const float* stds;  /* host */
  float img_h, img_w, wh_ratio_clip;
  int32_t nms_pre, nms_post, max_num;
  float nms_thr;
  float* proposals; int32_t* counts;
  int32_t cls_pitch, reg_pitch;  /* channels per pixel of the cls / reg maps (0 = A / 4A); lets both be
                                    channel slices of one fused [T][H][W][5A(+pad)] conv output */
} hvr_rpn_desc;
size_t hvr_rpn_workspace_bytes(int T, int H, int W, int A, int nms_pre);
int hvr_rpn_proposals(const hvr_rpn_desc* d, void* ws, size_t ws_bytes, void* stream);
/* Calls with T <= frames take the chip-wide kernels (histogram / counting-sort selection, banded suppression mask + one-wave
 * sweep: many workgroups per frame) instead of one workgroup per frame; the proposals are the same bit for bit.  Default 4
 * (HVR_RPN_WIDE=<frames> in the environment overrides it; 0 = never).  frames < 0 only queries.  Returns the previous value.
 * PROCESS-WIDE (an atomic): every thread and device of the process sees the new value from its next call on. */
int hvr_rpn_wide_frames(int frames);

/* ------------------------------------------------------------------------------------
 * RCNN detection read-out for one key frame.  Replaces BBoxHead.get_det_bboxes
 * (mmdet/models/bbox_heads/bbox_head.py:132-169; hrnmp_bbox_head.py:1009-1052 per branch):
 *   hvr_det_decode:     softmax over classes + delta2bbox + clip (+ / scale_factor)
 *   hvr_multiclass_nms: mmdet/core/post_processing/bbox_nms.py:6-66 (class-agnostic boxes)
 * logits [R][ldl] f32 with class logits at cls_off.. and 4 deltas at reg_off..;
 * rois [R][5]; scores [R][ncls]; boxes [R][4]; dets [max_num][5]; labels [max_num] int64
 * (0-based foreground class); *n_out device int32; rows [*n_out, max_num) of dets / labels come back zeroed (R > 0:
 * the caller's buffers need no initialisation).  R <= 512.
 * scale_factor <= 0 means rescale=False.  img_w <= 0 means no clipping.
 * ---------------------------------------------------------------------------------- */
int hvr_det_decode(const float* logits, int ldl, int cls_off, int reg_off, int ncls, const float* rois, int R,
                   const float* means, const float* stds, float wh_ratio_clip, float img_h, float img_w,
                   float scale_factor, float* scores, float* boxes, void* stream);
size_t hvr_multiclass_nms_workspace_bytes(int R, int ncls);
int hvr_multiclass_nms(const float* boxes, const float* scores, int R, int ncls, float score_thr, float iou_thr,
                       int max_num, float* dets, int64_t* labels, int32_t* n_out, void* ws, size_t ws_bytes,
                       void* stream);

/* ---- layout / dtype plumbing at the API boundary ---- */
/* any pair of the four dtypes; pairs other than f32 <-> bf16 move 8 elements per thread: n % 8 == 0 and 16-byte aligned buffers
 * (split half: n % 32 == 0 -- a contiguous tensor whose last dimension is a multiple of 32 -- and 128-byte alignment) */
int hvr_cast(const void* in, void* out, int64_t n, int from_dtype, int to_dtype, void* stream);
/* out = convert(in * scale), scale > 0 (a power of two for exact results): how the host layer moves between true values and the
 * scaled split-half tensors it keeps (activations x 2^4, weights x 2^6: hvrnet_amd/native.py) */
int hvr_cast_scaled(const void* in, void* out, int64_t n, int from_dtype, int to_dtype, float scale, void* stream);
/* to_nhwc != 0: [B][C][HW] -> [B][HW][C]; else the inverse */
int hvr_permute_nchw_nhwc(const void* in, void* out, int B, int C, int HW, int to_nhwc, int from_dtype,
                          int to_dtype, void* stream);
/* out[C][ldt] = in[R][ldx]^T, zero-filled for columns >= R */
int hvr_transpose_pad(const void* in, void* out, int R, int C, int64_t ldx, int64_t ldt, int dtype, void* stream);

/* Training targets: ground truth -> assigned, sampled boxes -> loss targets, all on the device.
 *   hvr_max_iou_assign : MaxIoUAssigner.assign (mmdet/core/bbox/assigners/max_iou_assigner.py:48-173, IoU of
 *                        mmdet/core/bbox/geometry.py:46-60) for float thresholds, gt_max_assign_all=True and no ignore
 *                        boxes: gt_inds[i] = -1 ignore / 0 background / g+1 assigned to gt g; max_overlaps[i].
 *                        `valid` (nullable, one byte per box) marks the boxes that take part: anchor_target_single's
 *                        inside_flags (mmdet/core/anchor/anchor_target.py:104-112) without compacting the anchors;
 *                        boxes outside get gt_inds -1 and max_overlaps -1.  n == 0 or k == 0 is HVR_EINVAL where the
 *                        reference raises ValueError('No gt or bboxes').
 *   hvr_sample_pos_neg : BaseSampler.sample (mmdet/core/bbox/samplers/base_sampler.py:32-78) over boxes classed by
 *                        cls[i] (> 0 positive, == 0 negative, < 0 neither).  From each class the boxes with the
 *                        smallest keys[i] win (ties: lower index); RandomSampler = uniform random keys
 *                        (random_sampler.py:37-53 shuffles on the host instead), OHEMHNLSampler.get_ohem_weights =
 *                        keys -loss (ohem_hnl_sampler.py:50-113).  inds[0..counts[0]) positives, then counts[1]
 *                        negatives, each ascending -- SamplingResult's order (sampling_result.py:9-12 after .unique()).
 *   hvr_box_targets    : regression / classification targets of the sampled boxes (transforms.py:6-31 bbox2delta).
 *                        scatter = 1: [n]-row outputs, row inds[j]  (anchor_target.py:121-155 after `unmap`);
 *                        scatter = 0: [num]-row outputs, row j      (bbox_target.py:35-62).  gt_labels null: label 1.
 *   hvr_rpn_loss       : AnchorHead.loss_single for one level, sigmoid objectness (anchor_head.py:141-160 with
 *                        losses/cross_entropy_loss.py:23-37 and losses/smooth_l1_loss.py:9-18) on the fused RPN head
 *                        output o [rows][ldo] (A logits then 4A deltas per position): out2 = (loss_rpn_cls,
 *                        loss_rpn_bbox), d_o = d(sum of both)/d o; avg_factor = max(counts[0],1) + max(counts[1],1).
 *   hvr_ce_rows        : per-row softmax cross entropy (`reduction_override='none'`, selsa_rcnn.py:209-218).
 *   hvr_det_loss_sampled: hvr_det_loss in the OHEM form (selsa_rcnn.py:224-232): rows outside cat(pos_inds, neg_inds)
 *                        carry zero weights; smooth-L1 and the accuracy are averaged over sel_counts[0]+sel_counts[1]. */
size_t hvr_max_iou_assign_workspace_bytes(int n, int k);
int hvr_max_iou_assign(const float* boxes, int ldb, int n, const float* gts, int k, const uint8_t* valid, float pos_iou_thr,
                       float neg_iou_lo, float neg_iou_hi, float min_pos_iou, int64_t* gt_inds, float* max_overlaps, void* ws,
                       size_t ws_bytes, void* stream);
int hvr_sample_pos_neg(const int64_t* cls, const float* keys, int n, int num, int num_expected_pos, float neg_pos_ub, int64_t* inds,
                       int32_t* counts, void* stream);
int hvr_box_targets(const float* boxes, int ldb, int n, const float* gts, const int64_t* gt_labels, const int64_t* gt_inds,
                    const int64_t* inds, const int32_t* counts, int num, const float* means4, const float* stds4, float pos_weight,
                    int scatter, int64_t* labels, float* label_weights, float* bbox_targets, float* bbox_weights, void* stream);
int hvr_rpn_loss(const float* o, int ldo, int A, int rows, const int64_t* labels, const float* label_weights, const float* bbox_targets,
                 const float* bbox_weights, const int32_t* counts, float beta, float* out2, float* d_o, void* stream);
int hvr_ce_rows(const float* logits, int ldl, int cls_off, int ncls, const int64_t* labels, int R, float* loss, void* stream);
/* Hard-proposal mining of the HVR head (mmdet/models/bbox_heads/hrnmp_bbox_head.py:357-414 `hardest_proposal_mining`): for
 * every query row of the scaled affinity matrix aff [Mq][ld] (f32), with row label labels[r] and key labels all_labels[k]:
 *   out4[r][0] argmax over keys with a different label  (the reference's masked_fill(-inf) + topk(1), `inds_for_pos_sm`)
 *   out4[r][1] argmin over keys with the same label     (masked_fill(+inf) + topk(1, largest=False), `inds_for_pos_nsm`)
 *   out4[r][2], out4[r][3] the two largest among keys with a different label (topk(2), `inds_for_bg`, used for label-0 rows)
 * ties to the lower index; a row without candidates gets index 0 (0, 1). */
/* STAND-IN for the triplet loss over the mined triples: the reference calls TripletNonLocalLoss(margin).compute_loss(q, k, labels,
 * [anchor_idx, pos_idx, neg_idx]) (hrnmp_bbox_head.py:555-561) from a pytorch_metric_learning fork that is NOT in the reference tree.
 * This is the library's published TripletMarginLoss with anchors from q and positives / negatives from k:
 * d(x,y) = ||x - y + 1e-6||_2, l_i = max(d(q_a,k_p) - d(q_a,k_n) + margin, 0), loss = sum l_i / max(#{l_i > 0}, 1).
 * out2 = (loss, number of active triples); dq [Mq][D] / dk [Mk][D] (f32, nullable) = d loss / d q, d k.  ws: 3 n floats. */
int hvr_triplet_margin(const void* q, int64_t ldq, const void* k, int64_t ldk, int D, int Mq, int Mk, const int64_t* anchor_idx,
                       const int64_t* pos_idx, const int64_t* neg_idx, int n, float margin, int dtype, float* ws, size_t ws_bytes,
                       float* out2, float* dq, float* dk, void* stream);
int hvr_mining_argreduce(const float* aff, int Mq, int Mk, int64_t ld, const int64_t* labels, const int64_t* all_labels, int64_t* out4,
                         void* stream);
int hvr_det_loss_sampled(const float* logits, int ldl, int cls_off, int reg_off, int ncls, const int64_t* labels,
                         const float* label_weights, const float* bbox_targets, const float* bbox_weights, int R,
                         const int32_t* sel_counts, float beta, float* out3, float* dlogits, void* stream);

/* Frame ingest: decoded BGR uint8 frame src [src_h][src_pitch bytes] (3 bytes per pixel) -> dst f32 [3][pad_h][pad_w]:
 * bilinear resize to new_w x new_h (OpenCV's 8-bit INTER_LINEAR fixed-point arithmetic, which mmcv.imrescale calls), optional
 * BGR->RGB, (x - mean) / std, zero padding -- the reference's Resize -> Normalize -> Pad -> ImageToTensor
 * (mmdet/datasets/pipelines/transforms.py:111-124,260-269,308-313) in one kernel instead of three CPU passes per frame. */
int hvr_ingest_frame(const uint8_t* src, int src_h, int src_w, int64_t src_pitch, float* dst, int new_h, int new_w, int pad_h, int pad_w,
                     const float* mean3, const float* std3, int to_rgb, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HVR_HIP_H_ */
