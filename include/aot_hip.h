/*
 * aot_hip.h -- C ABI of libaot_hip.so, the MI355X (gfx950) kernels behind the
 * AOT/DeAOT per-frame inference path of yoxu515/aot-benchmark.
 *
 * The reference has no FFI of its own (it is pure PyTorch); the entry points below
 * are one call per fused stage of its hot path and each one names the reference code
 * it replaces (paths relative to the reference repo).  Conventions:
 *
 *   - every pointer is DEVICE memory owned by the caller (16-byte aligned, fp32),
 *     `stream` is a hipStream_t passed as void*;
 *   - activations are NHWC / token-major:  [H*W, C] row-major with an explicit row
 *     stride `ld*` in floats, so slices of wider buffers can be read and written in place;
 *   - no allocation, no synchronisation, no host callbacks inside: every call is
 *     asynchronous on `stream` and hipGraph-capturable; no global mutable state;
 *   - return value: 0 on success, AOT_ERR_* (<0) on a rejected argument, or the positive
 *     hipError_t of a failed launch.  Nothing throws.
 */
#ifndef AOT_HIP_H
#define AOT_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define AOT_OK 0
#define AOT_ERR_BADARG (-1)
#define AOT_ERR_UNSUPPORTED (-2)

#define AOT_ACT_NONE 0
#define AOT_ACT_RELU 1
#define AOT_ACT_RELU6 2
#define AOT_ACT_GELU 3 /* exact-erf GELU */
#define AOT_ACT_SILU 4 /* conv/linear epilogue only */

/* library identification: "aot_hip <version> gfx950" */
const char* aot_hip_version(void);

/* Implicit-GEMM convolution / linear layer on the fp32 MFMA (v_mfma_f32_32x32x2_f32):
 *   out[m, n] = act( sum_k A[m, k] * w[k, n] + bias[n] + res[m % res_rows, n] )
 * with A the on-the-fly im2col of B NHWC images ([B*H*W, lda]), m = (b, oy, ox), k = (ky*KW + kx)*Cin + c.
 * w is [KH*KW*Cin, ldb] row-major (ldb >= Cout, multiple of 4; FrozenBN already folded in by the host);
 * wt (optional) is the same weight with k-contiguous rows, [Cout, ldwt] (ldwt >= KH*KW*Cin): when it is given and
 * Cin % 32 == 0 the LDS-direct tile kernel (csrc/gemm_lds.hip) runs, otherwise the register-staged one
 * (csrc/gemm_conv.hip); w may then be NULL (the LDS-direct kernels read wt only: AOT_ERR_UNSUPPORTED if the shape would need another
 * kernel) -- the training graph's products of two activations hand over one operand layout, not two.  bias / res may be NULL; res_rows = 0 means one residual row per output row, otherwise the
 * residual is a [res_rows, ldr] map shared by the images (row m % res_rows).  A linear layer is the 1x1 case with
 * B = 1, H = 1, W = M.  `scratch` (optional, scratch_floats floats) enables split-K for shapes with too few tiles
 * (partials summed in slice order: deterministic).  cfg = -1: kernel chosen for the lowest latency of this launch on
 * its own (one clip at a time); cfg = -2: chosen for the lowest cost when several clips keep the chip busy (the lean
 * tile kernel wherever it applies); cfg >= 0 forces a kernel configuration (tuning / tests).  Same results up to
 * summation order.  Requires Cin % 4 == 0 and lda % 4 == 0.
 * Replaces: every nn.Conv2d + FrozenBatchNorm2d + ReLU of networks/encoders/resnet.py:34-54,
 * 140-157 and mobilenetv2.py (1x1 / 3x3), encoder_projector (models/aot.py:19-21,81-84),
 * the nn.Linear layers of networks/layers/transformer.py:321-359 and attention.py:76-79,119,
 * and the FPN convs of networks/decoders/fpn.py:34-58. */
int aot_conv2d_nhwc_f32(const float* in, const float* w, const float* wt, const float* bias, const float* res,
                        float* out, float* scratch, long scratch_floats, int B, int H, int W, int Cin,
                        int OH, int OW, int Cout, int KH, int KW, int stride, int pad, int dil,
                        int lda, int ldb, int ldwt, int ldc, int ldr, int res_rows, int act, int cfg,
                        void* stream);

/* Second, parity-gated kernel family for the same operation: fp32-EQUIVALENT arithmetic on the bf16 matrix cores.  Every fp32
 * number is exactly the sum of three truncated bf16 numbers (8 + 8 + 8 significand bits); of the nine partial products of
 * a * w the six of order <= 2 are formed by v_mfma_f32_32x32x16_bf16 (each exact in its fp32 accumulator) and summed in fp32:
 * the dropped terms are <= 3 * 2^-24 of the product, one fp32 rounding.  aot_pack_bf16x6_f32 splits a weight w [K, ldb]
 * (K % 32 == 0; the layout aot_conv2d_nhwc_f32 takes) once into w6 = three bf16 planes in the kernel's tile order,
 * 3 * K * cout_pad * 2 bytes, cout_pad = Cout rounded up to 64.  aot_conv2d_bf16x6_f32 is aot_conv2d_nhwc_f32 on that weight
 * (same arguments and epilogue; needs Cin % 32 == 0; activations are split on the fly).  `tile` selects the member: 0 = by shape
 * (the register-staged 64x64 kernel with the weight fragments straight from global memory, 66, everywhere except KxK layers with
 * >= 128 output channels whose 128x128 tiles fill exactly one dispatch round -- those take the register-staged 128x128 form, 129);
 * 66 / 129 force one (tests, tuning).  Both form the same six products in the same order per accumulator and accumulate over k in
 * the same order: bit-identical results.  (The LDS-DMA members of rounds 3-5 -- tiles 64 / 128 / 65 / 1 and the form on pre-split
 * activation planes -- were superseded in round 5 and removed from the library in round 6.)  An engine opts in
 * (build_engine(..., mfma='bf16x6'); bench.py times this arithmetic by default since round 4), and results are reported under their own dtype string.
 * Replaces the same reference code as aot_conv2d_nhwc_f32. */
int aot_pack_bf16x6_f32(const float* w, void* w6, int K, int Cout, int ldb, int cout_pad, void* stream);
int aot_conv2d_bf16x6_f32(const float* in, const void* w6, int cout_pad, const float* bias, const float* res, float* out,
                          int B, int H, int W, int Cin, int OH, int OW, int Cout, int KH, int KW, int stride, int pad, int dil,
                          int lda, int ldc, int ldr, int res_rows, int act, int tile, void* stream);
/* Split-K forms of the same product for layers whose tiles alone do not fill the chip (the stride-16 maps): (K / 32) % |ksplit| == 0,
 * every k-slice writes its raw
 * partial tile to a slab of `scratch` ([ksplit][M][Cout] floats, scratch_floats = its size), one more launch sums the slabs in
 * slice order and applies bias / residual / activation.  ksplit > 1: the 128x128 LDS-DMA tile whose two waves per SIMD alternate
 * between a load phase and an MFMA phase (gemm_x6pp_kernel<., true>; the K = 2304 layers); ksplit < -1: |ksplit| slices on the 64x64
 * register-staged kernel with direct weight fragments (gemm_x6rd_kernel<., true>; the K = 1024 linear of the LSTT at one lane).
 * Replaces the same reference code as aot_conv2d_nhwc_f32. */
int aot_conv2d_bf16x6k_f32(const float* in, const void* w6, int cout_pad, const float* bias, const float* res, float* out,
                           int B, int H, int W, int Cin, int OH, int OW, int Cout, int KH, int KW, int stride, int pad, int dil,
                           int lda, int ldc, int ldr, int res_rows, int act, int ksplit, float* scratch, long scratch_floats,
                           void* stream);
/* The ResNet stem in the same family (round 5): a KxK convolution of B NHWC images with FOUR channels (the image padded to r, g, b, 0 by
 * aot_nchw_to_nhwc_f32): one 16-byte chunk of an im2col row is one filter tap, a k-step is eight taps.  w6 = aot_pack_bf16x6_f32 of the
 * weight [Kp, ldb] with rows k = 4 * tap + channel, Kp = ceil(KH * KW / 8) * 32, zero rows past KH * KW * 4.  in [B*H*W, 4], out
 * [B*OH*OW, ldc].  Replaces conv1 + bn1 + relu of networks/encoders/resnet.py:140-143. */
int aot_conv2d_c4_bf16x6_f32(const float* in, const void* w6, int cout_pad, const float* bias, float* out, int B, int H, int W,
                             int OH, int OW, int Cout, int KH, int KW, int stride, int pad, int dil, int ldc, int act, void* stream);

/* Plain bf16 form of the same operation for the TRAINING path (`--amp` of the reference's trainer, trainer.py:123-125,460-487 --
 * there fp16 autocast + GradScaler; BASELINE config 5: bf16): both operands rounded to bf16 (round to nearest even), ONE
 * v_mfma_f32_32x32x16_bf16 product, fp32 accumulation and output.  aot_pack_bf16_f32 rounds a weight w [K, ldb] (K % 32 == 0)
 * into one plane of the tile order above (K * cout_pad * 2 bytes); aot_conv2d_bf16_f32 takes fp32 activations and rounds them in
 * registers (v_cvt_pk_bf16_f32).  Same arguments and epilogue as aot_conv2d_bf16x6_f32 (64x64 tile).  ksplit > 1 (1x1 only, K / 32
 * divisible by it; the weight gradients dW = dY^T X of the training path, whose K is the row count): every k-slice writes its
 * partial tile to a slab of `scratch` [ksplit][M][Cout] floats and one more launch sums the slabs in order (+ bias / residual / act). */
int aot_pack_bf16_f32(const float* w, void* wq, int K, int Cout, int ldb, int cout_pad, void* stream);
int aot_conv2d_bf16_f32(const float* in, const void* wq, int cout_pad, const float* bias, const float* res, float* out,
                        int B, int H, int W, int Cin, int OH, int OW, int Cout, int KH, int KW, int stride, int pad, int dil,
                        int lda, int ldc, int ldr, int res_rows, int act, int ksplit, float* scratch, void* stream);

/* Depthwise KxK convolution over B NHWC maps ([B*H*W, C]), w is [KH*KW, C], optional bias, fused activation.
 * Replaces: GNActDWConv2d.conv / DWConv2d.conv (networks/layers/basic.py:19-25,33,41-47,54)
 * and the depthwise 3x3 of MobileNetV2 InvertedResidual (mobilenetv2.py:93-98). */
int aot_dwconv2d_nhwc_f32(const float* in, const float* w, const float* bias, float* out, int B,
                          int H, int W, int C, int OH, int OW, int KH, int KW,
                          int stride, int pad, int dil, int act, void* stream);

/* 3x3 stride-2 pad-1 max pooling, NHWC (resnet.py:79,144). */
int aot_maxpool3x3s2_nhwc_f32(const float* in, float* out, int H, int W, int C, int OH, int OW,
                              void* stream);

/* [C,H,W] planar image -> [H*W, Cpad] interleaved, channels >= C zero filled (input layout
 * change in front of the stem conv; the reference keeps NCHW throughout). */
int aot_nchw_to_nhwc_f32(const float* in, float* out, int C, int H, int W, int Cpad, void* stream);
/* [H*W, C] (row stride ld) -> [C,H,W] planar: the layout callers of the reference's surface expect back (feature maps of
 * AOT.encode_image, aot.py:81-84; logits of decode_id_logits, aot.py:86-92). */
int aot_nhwc_to_nchw_f32(const float* in, float* out, int C, int H, int W, int ld, void* stream);

/* LayerNorm over the last dim (eps inside sqrt, biased variance, as torch):
 *   y = LN(x)*gamma + beta ;  if (add && y2)  y2 = y + add[row % add_rows]   (positional embedding shared by the lanes
 *   of a batch, transformer.py:322; add_rows = 0: one add row per row)
 * Replaces nn.LayerNorm in transformer.py:321,329,355 and LSTT.decoder_norms (:124-135). */
int aot_layernorm_f32(const float* x, const float* gamma, const float* beta, float* y,
                      const float* add, float* y2, int M, int C, int ldx, int ldy, int ldadd,
                      int ldy2, int add_rows, float eps, void* stream);

/* GroupNorm over B lanes of [M, C] NHWC maps (lane b = rows [b*M, (b+1)*M)).  Statistics: ONE launch, deterministic
 * two-level fp64 reduction -- nsplit partial-sum workgroups per (lane, group); the last one to arrive (device-scope
 * ticket) adds the partials in index order and writes `stats` = [B][G][2] doubles (mean, rstd).  `scratch` must hold
 * B*G*nsplit*2 doubles; `ticket` B*G unsigned ints that are ZERO before the first call (the kernel leaves them zero).
 * Apply: y = act(GN(x)) (+ add) (act: 0 none, 1 relu, 3 exact-erf GELU; add: optional map added after the activation,
 * one row per row, or -- add_rows > 0 -- a [add_rows, ldadd] map shared by the lanes, row % add_rows).
 * Replaces nn.GroupNorm in GNActDWConv2d (basic.py:18,31-32), ConvGN (basic.py:82-85) + F.relu_ (fpn.py:41-56) and
 * GroupNorm1D (basic.py:6-12). */
int aot_groupnorm_stats_f32(const float* x, double* scratch, double* stats, unsigned* ticket, int B, int M,
                            int C, int G, int ldx, float eps, int nsplit, void* stream);
int aot_groupnorm_apply_f32(const float* x, const double* stats, const float* gamma,
                            const float* beta, float* y, const float* add, int B, int M, int C, int G,
                            int ldx, int ldy, int ldadd, int add_rows, int act, void* stream);
/* GroupNorm-apply + activation + 5x5 depthwise conv (stride 1, pad 2, no bias) in one launch, for 32-channel groups
 * (C == 32*G): x [B*H*W, ldx] with the statistics of aot_groupnorm_stats_f32, w [25, C] -> out [B*H*W, ldo].
 * Replaces GNActDWConv2d.forward after the statistics (gn -> GELU -> conv, networks/layers/basic.py:27-35). */
int aot_gn_act_dwconv5_f32(const float* x, const double* stats, const float* gamma, const float* beta,
                           const float* w, float* out, int B, int H, int W, int C, int G, int ldx, int ldo,
                           int act, void* stream);

/* The statistics pass folded into the producing GEMM (round 5): aot_linear_gn_bf16x6_f32 is a linear layer of the bf16x6 family
 * (out = act(x W + bias (+ res)), x [M, lda], w6 from aot_pack_bf16x6_f32, Cout % 32 == 0) whose tile end ALSO writes the GroupNorm
 * partial sums of `out` for 32-channel groups: gn_part [2 * ceil(M / 64)][Cout / 32][2] floats = (sum, sum of squares) of every
 * 32-row x 32-column block (gn_part_floats = the buffer's size).  aot_gn_act_dwconv5p_f32 is aot_gn_act_dwconv5_f32 (one lane) that
 * takes those P = 2 * ceil(M / 64) partial rows instead of finished statistics: every workgroup adds its group's partials in index
 * order in double and forms (mean, rstd) as aot_groupnorm_stats_f32 does -- deterministic, no statistics launch, no extra pass over
 * the map.  Replaces linear1 + GNActDWConv2d of the LSTT's feed-forward (networks/layers/transformer.py:355-362, basic.py:15-35). */
int aot_linear_gn_bf16x6_f32(const float* in, const void* w6, int cout_pad, const float* bias, const float* res, float* out,
                             int M, int K, int Cout, int lda, int ldc, int ldr, int res_rows, int act, float* gn_part,
                             long gn_part_floats, void* stream);
int aot_gn_act_dwconv5p_f32(const float* x, const float* part, int P, const float* gamma, const float* beta, const float* w,
                            float* out, int H, int W, int C, int G, int ldx, int ldo, int act, float eps, void* stream);

/* A split-K linear layer with Cout == 256 whose reduce launch also writes LayerNorm(out) to a second map (round 6): out = act(x W + b
 * (+ res)) as aot_conv2d_bf16x6k_f32 (ksplit < 0: the 64x64 kernel), ln_out [M, ld_ln] = LayerNorm(out) * gamma + beta with the
 * arithmetic of aot_layernorm_f32 (bit-identical to that launch on the stored result): a row of 256 channels is one wave of the
 * reduce.  linear2 (+ residual) of an LSTT block followed by the stack's output norm (networks/layers/transformer.py:124-135, 359-362). */
int aot_linear_bf16x6k_ln_f32(const float* in, const void* w6, int cout_pad, const float* bias, const float* res, float* out, int M, int K,
                              int Cout, int lda, int ldc, int ldr, int res_rows, int act, int ksplit, float* scratch, long scratch_floats,
                              const float* ln_gamma, const float* ln_beta, float* ln_out, int ld_ln, float eps, void* stream);

/* n <= 4 independent linear layers of ONE shape in one launch of the bf16x6 family (round 6): out[g] = act(in[g] W[g] + bias[g] (+ res[g])),
 * g = blockIdx.y.  in / w6 / bias / res / out are HOST arrays of n device pointers, read at launch time (bias, res: NULL, or arrays
 * whose entries are all set or all NULL); w6[g] from aot_pack_bf16x6_f32 with one common cout_pad.  A linear layer on the stride-16
 * map has 108 tiles for 256 CUs: the three layers' linear_V of the memory update (networks/layers/transformer.py:364-367 via
 * aot_engine.py:307-338) and the value / gate projections of a GPM block's self-propagation (transformer.py:643-653) run side by side. */
int aot_linear_group_bf16x6_f32(int n, const float* const* in, const void* const* w6, int cout_pad, const float* const* bias,
                                const float* const* res, float* const* out, int M, int K, int Cout, int lda, int ldc, int ldr,
                                int res_rows, int act, void* stream);

/* LayerNorm folded into the consuming GEMM (round 6; SURVEY 8b `aot_layernorm_linear`): out = act(LayerNorm(x) W + b (+ res)) in one
 * launch of the bf16x6 family, the normalised map never materialised.  x [M, lda] un-normalised, K % 32 == 0; the CALLER folds the
 * affine part once per model: w6 = aot_pack_bf16x6_f32 of W' = diag(gamma) W, bias = beta W + b, colsum [Cout] = the column sums of W'.
 * The kernel owes (x - mean) * rstd per row, and both statistics ride along its k-loop (no pass over the rows in front of it): with
 * c = the row's first element, d = x - c is what gets split into the bf16 planes, sum d and sum d^2 are added up by the staging
 * threads, and the tile end applies rstd * (acc - (mean - c) * colsum) -- a shifted one-pass variance, |mean - c| being of the
 * order of the row's spread.  eps as nn.LayerNorm (biased variance).  gn_part / gn_part_floats optional (NULL, 0): the GroupNorm
 * partials of `out` exactly as aot_linear_gn_bf16x6_f32.  Replaces norm1 -> linear_Q|K|V of the self-attention and norm3 -> linear1
 * of the LSTT block (networks/layers/transformer.py:321-323, 355-359 in the reference). */
int aot_layernorm_linear_bf16x6_f32(const float* in, const void* w6, int cout_pad, const float* bias, const float* colsum, const float* res,
                                    float* out, int M, int K, int Cout, int lda, int ldc, int ldr, int res_rows, int act, float eps,
                                    float* gn_part, long gn_part_floats, void* stream);

/* Multi-head softmax attention over a key/value bank, flash style (no S materialised):
 *   out[n, h*d:(h+1)*d] = softmax_t( (q[n,h]/scale_div) . k[t,h] ) @ v[t,h]          (d == 32)
 * B independent lanes (object groups of one frame, or clips): lane b owns query rows [b*Nq, (b+1)*Nq) of q / out
 * (row strides ldq / ldo) and the key/value rows [b*kv_brows, b*kv_brows + T) of k / v (one bank per lane; kv_brows >= T
 * when B > 1).  H heads of width 32.  `T_dev` (optional) is a device int overriding T so a captured graph can follow a
 * growing bank.  The four waves of a workgroup split the key range between them and merge through LDS; nsplit > 1
 * additionally splits the bank over workgroups: `part` must then hold nsplit*B*Nq*(H*32 + 2*H) floats, receives the
 * un-normalised partial (O, m, l) of every split, and aot_attn_merge_f32 (with Nq := B*Nq) must follow on the same
 * stream to produce `out`.  Exact fp32: QK^T and PV on v_mfma_f32_32x32x2_f32.
 * Replaces MultiheadAttention.forward's core, networks/layers/attention.py:82-117 (long-term
 * attention over the memory bank and self-attention). */
int aot_attn_f32(const float* q, const float* k, const float* v, float* out, float* part, int B, long kv_brows,
                 int Nq, int T, const int* T_dev, int H, int d, int ldq, int ldk, int ldv, int ldo,
                 float scale_div, int nsplit, void* stream);
/* Merge of the nsplit partials written by aot_attn_f32 / aot_gated_attn_f32 (nsplit > 1) into out [Nq, ldo] (second half of
 * the softmax(QK^T)V of attention.py:92-117 / 672-707 when the bank is cut over workgroups):
 * C output channels in H groups that carry one (m, l) each; optional gate [Nq, ldg] multiplies the result. */
int aot_attn_merge_f32(const float* part, const float* gate, float* out, int Nq, int H, int C, int ldg,
                       int ldo, int nsplit, void* stream);

/* bf16x6 member of the attention family (fp32-equivalent arithmetic on the bf16 matrix cores, like
 * aot_conv2d_bf16x6_f32: every fp32 operand is exactly the sum of three truncated bf16 numbers, six of the nine partial
 * products are kept, each exact in its fp32 accumulator).  The memory bank is kept PRE-SPLIT and tile-major:
 *   kv [B][cap_rows / 32][H][K: 3 planes x 2 sub-steps x 64 lanes x 8 | V: the same] bf16   (12 KB per 32-row tile and head;
 *   every operand fetch of a wave is one contiguous KB; V transposed, rows in the order the score tile leaves P in)
 * cap_rows (a multiple of 32) = rows of one lane's bank.  aot_attn_pack_x6_f32 splits rows [b*src_brows, +rows) of k / v
 * (fp32, C = H*32 columns, row strides ldk / ldv) into lane b's planes at bank rows slot*rows .. (slot_dev: optional device
 * int overriding slot, so that a replayed graph can append); the planes must start zeroed (rows past the bank length are
 * multiplied by weights that are exactly 0).  aot_attn_x6_f32 is aot_attn_f32 on such a bank: same grid, key split,
 * `part` format and merge (aot_attn_merge_f32 must follow when nsplit > 1).
 * Replaces MultiheadAttention.forward's core, networks/layers/attention.py:82-117, and the bank append of
 * networks/engines/aot_engine.py:329-338 for the packed copy. */
int aot_attn_pack_x6_f32(const float* k, const float* v, void* kv, int B, long rows, int C, long src_brows, int ldk, int ldv,
                         long cap_rows, const int* slot_dev, int slot, void* stream);
int aot_attn_x6_f32(const float* q, const void* kv, float* out, float* part, int B, long cap_rows, int Nq, int T,
                    const int* T_dev, int H, int d, int ldq, int ldo, float scale_div, int nsplit, void* stream);

/* The gated-propagation form in the bf16x6 family (twin of aot_gated_attn_f32; dqk = 128, dv = 1024): K and V are kept in two
 * packed buffers of the layout above -- planes [B][cap_rows / 32][C / 32][3 planes x 2 sub-steps x 64 lanes x 8] bf16 -- filled
 * by aot_attn_pack_x6_part_f32 (transpose = 0: K-style, lane = row of the tile; transpose = 1: V-style, lane = channel, rows in
 * the score tile's C/D order).  Same split / merge protocol as aot_gated_attn_f32 (H := 4 groups; with nsplit > 1 pass the gate
 * to aot_attn_merge_f32).  Replaces GatedPropagation.forward's core, networks/layers/attention.py:672-707, and the packed
 * copy of the bank append of networks/engines/deaot_engine.py:20-56. */
int aot_attn_pack_x6_part_f32(const float* x, void* planes, int B, long rows, int C, long src_brows, int ldx, long cap_rows,
                              const int* slot_dev, int slot, int transpose, void* stream);
int aot_gated_attn_x6_f32(const float* q, const void* kp, const void* vp, const float* gate, float* out, float* part, int B,
                          long cap_rows, int Nq, int T, const int* T_dev, int dqk, int dv, int ldq, int ldg, int ldo,
                          float scale_div, int nsplit, void* stream);

/* Top-k sparse form of aot_attn_f32 (MultiheadAttention with top_k > 0, networks/layers/attention.py:102-105, a
 * default-off long-video knob): per query row and head only the top_k largest scores enter the softmax and the
 * value sum.  `scores` is caller-owned scratch of H*Nq*((T+3)&~3) floats (the materialised score matrix).
 * 0 < top_k < T required (top_k >= T is the dense softmax: call aot_attn_f32).  Among scores EQUAL to the k-th
 * largest the choice is arbitrary, as in torch.topk. */
int aot_attn_topk_f32(const float* q, const float* k, const float* v, float* out, float* scores, int Nq, int T,
                      int H, int d, int ldq, int ldk, int ldv, int ldo, float scale_div, int top_k, void* stream);

/* Top-k sparse form of aot_gated_attn_f32 (GatedPropagation with top_k > 0, networks/layers/attention.py:689-693): one head
 * of width d = 128, value / gate / out of width dv (a multiple of 4, <= 2048).  `scores` is caller-owned scratch of
 * Nq*((T+3)&~3) floats.  0 < top_k < T required.  Among scores EQUAL to the k-th largest the first ones in key order are
 * taken (torch.topk leaves that choice open); the summation order is fixed, so the result is reproducible. */
int aot_gated_attn_topk_f32(const float* q, const float* k, const float* v, const float* gate, float* out, float* scores,
                            int Nq, int T, int d, int dv, int ldq, int ldk, int ldv, int ldg, int ldo, float scale_div,
                            int top_k, void* stream);

/* Gated-propagation attention of DeAOT, single head: out = softmax((q/scale_div) k^T) v  (* gate), with
 * q [Nq, dqk=128], k [T, 128], v [T, dv] (dv a multiple of 256; 1024 = [V | ID_V]), gate/out [Nq, dv]; B lanes laid
 * out as in aot_attn_f32.  Same split/merge protocol as aot_attn_f32 with H := dv/256 groups; with nsplit > 1 pass the
 * gate to aot_attn_merge_f32 instead.  Replaces GatedPropagation.forward's core, attention.py:672-707. */
int aot_gated_attn_f32(const float* q, const float* k, const float* v, const float* gate, float* out,
                       float* part, int B, long kv_brows, int Nq, int T, const int* T_dev, int dqk, int dv,
                       int ldq, int ldk, int ldv, int ldg, int ldo, float scale_div, int nsplit, void* stream);

/* Short-term (windowed) attention of AOT, fused: window dot products, relative-position key
 * bias (grouped 1x1 conv on the UNSCALED q), border masking, softmax over the (2*max_dis+1)^2
 * window, aggregation of v plus relative_emb_v.  q,k,v,out are token-major [h*w, ld*] with H
 * heads of width 32.  The wave-uniform tables are laid out for 64-byte scalar loads, WS = 2*max_dis+1:
 *   relk_t [H][WS][32][16]: relk_t[hd][dy][c][dx] = sqrt(d) * relative_emb_k.weight[hd*WS*WS + dy*WS + dx][c]
 *                           (the kernel holds q/scale_div only; scale_div must equal sqrt(d))
 *   relk_b [H][WS][16]    : relative_emb_k.bias[hd*WS*WS + dy*WS + dx]
 *   relv_t [H][WS][32][16]: relative_emb_v[hd][c][dy*WS + dx]           (dx = 15 is padding)
 * B lanes: lane b owns rows [b*h*w, (b+1)*h*w) of q / out and rows [b*kv_brows, ...) of k / v.
 * Replaces MultiheadLocalAttentionV2.forward + local2global + pad_and_unfold
 * (attention.py:308-428) i.e. what spatial_correlation_sampler computes, minus `projection`. */
int aot_local_attn_f32(const float* q, const float* k, const float* v, const float* relk_t,
                       const float* relk_b, const float* relv_t, float* out, int B, long kv_brows, int h, int w,
                       int H, int d, int max_dis, int ldq, int ldk, int ldv, int ldo, float scale_div,
                       void* stream);

/* Short-term gated propagation of DeAOT (single head): window scores on q = k [h*w, 128] with the relative-position
 * key bias, softmax over the 15x15 window, aggregation of v [h*w, dv] (dv % 32 == 0) and multiplication by the
 * gate [h*w, dv] (may be NULL).  relk_t [15][128][16] (sqrt(128)-scaled, layout of aot_local_attn_f32 with one
 * head), relk_b [15][16]; prob is a [B][225, h*w] scratch map; B lanes as in aot_local_attn_f32.  Three launches
 * (scores, softmax, aggregate).
 * Replaces LocalGatedPropagation.forward up to `agg_value * u`, attention.py:789-855. */
int aot_local_gated_f32(const float* q, const float* k, const float* v, const float* gate, const float* relk_t,
                        const float* relk_b, float* prob, float* out, int B, long kv_brows, int h, int w, int dqk,
                        int dv, int max_dis, int ldq, int ldk, int ldv, int ldg, int ldo, float scale_div,
                        void* stream);

/* Swin window attention (W-MSA / SW-MSA, 7x7 windows, heads of width 32), fused with the reference's pad / roll /
 * window_partition / window_reverse / crop: qkv [B*H*W, ld] = [q | k | v] (C each) is the output of the qkv Linear on the
 * LayerNormed tokens of B images stacked along the rows (round 6: one launch for the whole batch), qkv_bias [3C] is its bias
 * (the value padded tokens take), rpb_table [169, nH], out [B*H*W, ldo] (pre-projection).  shift = 0 or 3.  Replaces WindowAttention.forward + the index plumbing of
 * SwinTransformerBlock.forward, networks/encoders/swin/swin_transformer.py:159-199, 262-312. */
int aot_swin_window_attn_f32(const float* qkv, const float* qkv_bias, const float* rpb_table, float* out, int B, int H, int W,
                             int C, int nH, int window, int shift, int ld, int ldo, float scale, void* stream);
/* PatchMerging gather (swin_transformer.py:338-356): x [H*W, ldx] -> out [ceil(H/2)*ceil(W/2), 4C], zero padded. */
int aot_patch_merge_f32(const float* x, float* out, int H, int W, int C, int ldx, void* stream);

/* Identity-bank embedding of a label map: out[(Y,X), c] = bias[c] + sum_{ky,kx} table[label(16Y+ky-pad,
 * 16X+kx-pad), ky, kx, c] over in-image taps; labels outside [0, nlabel) or non-integer add nothing.
 * mask is [H,W] float label ids, table [nlabel, K, K, C]; sumtab [nlabel, C] (optional) = sum of table over the
 * K*K taps, used when a token's whole window carries one label.
 * group_size > 0: lane g is object group group0+g of the SAME label map (AOTInferEngine.separate_mask,
 * aot_engine.py:515-534): labels (group0+g)*group_size+1 .. +group_size map to 1 .. group_size, everything else to
 * background 0; rows [g*OH*OW, ...) of out.  group_size = 0 (G = 1): the labels are used as they are.
 * Fused memory update (nfuse <= 4): fuse_out[i][row] = id_emb[row] + fuse_add[i][row] (host arrays of device
 * pointers; row strides ldadd / ldfout) -- the `V + id_emb` of every LSTT layer (transformer.py:364-367) in the same
 * launch; out may then be NULL.
 * Replaces one_hot_mask (utils/image.py:69-74) + patch_wise_id_bank conv (models/aot.py:50-63,76-79). */
int aot_idbank_f32(const float* mask, const float* table, const float* sumtab, const float* bias, float* out,
                   int G, int group_size, int group0, int H, int W, int OH, int OW, int K, int stride, int pad,
                   int C, int nlabel, int ldo, const float* const* fuse_add, float* const* fuse_out, int nfuse,
                   int ldadd, int ldfout, void* stream);

/* Bilinear resize of B NHWC maps with torch's fp32 source-index arithmetic; out = resize(in) (+ add); `add` is one map
 * per lane, or (add_shared) one [OH*OW, ldadd] map added to every lane.  Replaces F.interpolate(mode='bilinear') in
 * fpn.py:44-55. */
int aot_bilinear_nhwc_f32(const float* in, const float* add, float* out, int B, int IH, int IW, int OH,
                          int OW, int C, int ldi, int ldadd, int ldo, int align_corners, int add_shared,
                          void* stream);
/* aot_groupnorm_apply_f32 + aot_bilinear_nhwc_f32 in one launch (round 6): out = bilinear(act(GroupNorm(in))) (+ add), the four taps
 * normalised on the fly with the apply kernel's own arithmetic -- bit-identical to the pair, the normalised map never written.  stats
 * [B][G][2] doubles (mean, rstd); (C / G) % 4 == 0.  The FPN head's conv_16x / conv_8x blocks, whose GroupNorm output feeds only the next
 * upsampling (networks/decoders/fpn.py:40-44, 50-51). */
/* aot_groupnorm_apply_f32 + a 1x1 convolution with Cout <= 32 in one launch (round 6): out = act(gn_act(GroupNorm(in)) W + bias), the
 * normalisation applied to the A operand as it is loaded -- bit-identical to the pair.  in [B*M, lda] (B lanes of M rows), stats
 * [B][G][2] doubles, w [K, ldb] k-major as aot_conv2d_nhwc_f32 takes it.  The FPN head's conv_4x block -> conv_out
 * (networks/decoders/fpn.py:56-58). */
int aot_gn_conv1x1_f32(const float* in, const double* stats, const float* gamma, const float* beta, const float* w, const float* bias,
                       float* out, int B, int M, int K, int Cout, int G, int lda, int ldb, int ldc, int gn_act, int act, void* stream);
int aot_gn_bilinear_nhwc_f32(const float* in, const double* stats, const float* gamma, const float* beta, const float* add, float* out,
                             int B, int IH, int IW, int OH, int OW, int C, int G, int ldi, int ldadd, int ldo, int align_corners,
                             int add_shared, int act, void* stream);

/* Logit finalisation for the G object groups (lanes) of a frame: logits [G*IH*IW, ldi] stride-4 NHWC.  Per group the
 * channels of unused identities (group g holds objects g*(C-1)+1 .. min((g+1)*(C-1), obj_total)) are set to -1e10, the
 * masked maps are written planar to out4 [G][C,IH,IW] (may be NULL) and bilinearly resized to OH x OW.  G == 1: out is
 * [C,OH,OW].  G > 1: the resized logits are merged in the same launch by the reference's soft aggregation (softmax per
 * group, background = product of the background probabilities, clamp to [1e-5, 1-1e-5], logit): out is
 * [1 + G*(C-1), OH, OW].  Replaces aot_engine.py:367-378 and AOTInferEngine.soft_logit_aggregation (:565-582). */
int aot_logits_finalize_f32(const float* logits, float* out4, float* out, int G, int IH, int IW, int C,
                            int ldi, int OH, int OW, int obj_total, int align_corners, void* stream);

/* The frame tail for ONE object group and ONE augmentation in one launch (round 5): from the decoder's stride-4 logits
 * [IH*IW, ldi] straight to (a) label_out [OH*OW] = argmax of the softmax of the bilinearly resized, id-masked logits (what
 * aot_logits_finalize_f32 + aot_fuse_probs_f32 give: same arithmetic in the same order, bit-identical labels), (b) label_in
 * [LH*LW] (may be NULL) = that label map resized to the engine's input size by nearest neighbour (aot_label_resize_f32), the
 * mask the memory update reads, and (c) out4 [C, IH, IW] (may be NULL) = the planar masked stride-4 logits (pred_id_logits).  The
 * output-size logits (C planes of OH x OW floats) are never written.
 * Replaces aot_engine.py:367-378, evaluator.py:332-352 (softmax -> mean over one augmentation -> argmax) and :375-381. */
int aot_frame_tail_f32(const float* logits, float* out4, float* label_out, float* label_in, int IH, int IW, int C, int ldi,
                       int OH, int OW, int LH, int LW, int obj_total, int align_corners, void* stream);

/* out = a + b over n floats (n % 4 == 0) -- V + id_emb in fuse_key_value_id (transformer.py:364-367). */
int aot_add_f32(const float* a, const float* b, float* out, long n, void* stream);

/* Row-block copy between token-major buffers, B lanes: dst[(b*dst_brows + slot*rows + r), 0..C) = src[(b*src_brows + r), 0..C)
 * for r < rows (row strides lds / ldd; src_brows = 0: the same source block for every lane).  `slot_dev` (optional) is a
 * device int overriding `slot`, so that a captured graph can append to whichever bank slot is next.
 * Replaces the torch.cat of update_long_term_memory (networks/engines/aot_engine.py:291-305: the frame's K / V are written
 * into the next slot of the pre-allocated bank instead of re-copying the whole bank) and the per-lane copy of the shared
 * image embedding in front of the LSTT (aot_engine.py:606-616). */
int aot_copy_rows_f32(const float* src, float* dst, int B, long rows, int C, long src_brows, long dst_brows, int lds, int ldd,
                      const int* slot_dev, int slot, void* stream);

/* ---- evaluator-side steps: what the reference does on either side of the engine per frame ---- */

/* Frame preparation: cubic resize of an interleaved H x W x 3 image (uint8 or float32 in [0,255], row stride ld_src
 * elements) to OH x OW with OpenCV's INTER_CUBIC arithmetic (a = -0.75, half-pixel centres, replicated border; a plain copy
 * when the size is unchanged), optional horizontal flip of the RESULT, then ((v / 255) - mean[c]) / std[c] with the
 * reference's float32/float64 rounding sequence; dst is the engine input [3, OH, OW] planar.
 * Replaces MultiRestrictSize's cv2.resize + flip and MultiToTensor, dataloaders/video_transforms.py:655-680,703-711. */
int aot_preprocess_f32(const void* src, int src_is_u8, int H, int W, int ld_src, float* dst, int OH, int OW,
                       int flip, const double* mean3, const double* std3, void* stream);

/* Test-time-augmentation fusion: logits [A][nc][OH*OW] (one engine per augmentation, all decoded at the original
 * size); bit a of flipmask = augmentation a is horizontally flipped.  Per augmentation: un-flip, softmax over the nc
 * channels, argmax -> aug_labels[a] (optional); mean of the probabilities over A -> fused_prob (optional) -> argmax ->
 * fused_label.  new_label (optional): pixels where it is non-zero override every label map (new objects).
 * Replaces networks/managers/evaluator.py:325-372 (flip_tensor, softmax, mean, argmax, keep-merge). */
int aot_fuse_probs_f32(const float* logits, const float* new_label, float* fused_label, float* aug_labels,
                       float* fused_prob, int A, int nc, int OH, int OW, int flipmask, void* stream);

/* Label feedback: optional horizontal flip, then F.interpolate(mode="nearest") of a [H, W] float label map to the engine's
 * input size [OH, OW] (torch's legacy nearest index rule).  Replaces evaluator.py:375-386,399-408. */
int aot_label_resize_f32(const float* src, float* dst, int H, int W, int OH, int OW, int flip, void* stream);

/* ---- training-side stages (first slice; networks/layers/loss.py, utils/ema.py, trainer.py:116-118,501-503) ----
 * logits [B, C, P] planar (the reference's [B,C,H,W], P = H*W), labels [B, P] fp32 class ids, 255 = ignore; C <= 16. */

/* CrossEntropyLoss of the reference (loss.py:137-188) for B samples: loss_px [B,P] = per-pixel cross entropy (0 on ignored
 * pixels); loss[b] = mean of the top_k largest per-pixel losses of sample b (hard-example mining, top_k > 0; thr [2B] receives
 * for the backward pass the order key of the k-th largest, thr[b], and the share of the pixels tied with it that entered the
 * mean, thr[B + b] as float bits) or, with top_k = 0, the mean over the valid pixels (cnt[b] = their
 * number). */
int aot_ce_loss_f32(const float* logits, const float* labels, float* loss_px, float* loss, unsigned* thr, float* cnt, int B,
                    int C, long P, long top_k, void* stream);
/* What autograd derives from loss.py:137-188: grad [B,C,P] = gscale[b] * (softmax - onehot) on the pixels that entered loss[b] (thr given: per-pixel loss > the k-th
 * largest, and the pixels tied with it weighted by their share; thr = NULL: every valid pixel), 0 elsewhere.  gscale[b] = upstream gradient / k (or / cnt[b]). */
int aot_ce_loss_bwd_f32(const float* logits, const float* labels, const float* loss_px, const unsigned* thr,
                        const float* gscale, float* grad, int B, int C, long P, void* stream);

/* SoftJaccordLoss of the reference (loss.py:119-137 = tversky_loss(alpha = beta = 1), :29-55): loss[b] = mean over the classes
 * present in sample b of 1 - I / (I + FP + FN + eps) on the softmax probabilities of the valid pixels.  part is scratch of
 * B*nchunk*16*3 doubles, sums [B,16,3] (I, sum p, sum g per class) is kept for the backward pass. */
int aot_soft_jaccard_f32(const float* logits, const float* labels, double* part, double* sums, float* loss, int B, int C, long P,
                         int nchunk, float eps, void* stream);
/* What autograd derives from loss.py:26-52,119-137: grad [B,C,P] of sum_b gout[b] * loss[b] w.r.t. the logits (through the softmax). */
int aot_soft_jaccard_bwd_f32(const float* logits, const float* labels, const double* sums, const float* gout, float* grad, int B,
                             int C, long P, float eps, void* stream);

/* One tensor of torch.optim.AdamW (decoupled weight decay, no amsgrad) at 1-based step `step`; gscale multiplies the
 * gradient first (the clip_grad_norm_ factor, trainer.py:501-503). */
int aot_adamw_step_f32(float* p, const float* g, float* m, float* v, long n, float lr, float weight_decay, float beta1,
                       float beta2, float eps, int step, float gscale, void* stream);
/* utils/ema.py:63-66: shadow -= one_minus_decay * (shadow - param). */
int aot_ema_update_f32(float* shadow, const float* param, long n, float one_minus_decay, void* stream);
/* out[0] += sum(x^2) in fp64 (one workgroup, fixed order): the gradient-norm reduction of clip_grad_norm_
 * (trainer.py:501-503). */
int aot_sumsq_accum_f64(const float* x, long n, double* out, void* stream);
/* The same three steps over FLAT training state -- every trainable tensor a range of one parameter / gradient / moment buffer
 * (what DistributedDataParallel's gradient buckets and torch's fused optimisers do for the reference's trainer,
 * trainer.py:59-74,116-118,501-503) -- so that a step is three launches whatever the number of tensors:
 * aot_sumsq_flat_f64: out[0] = sum(x^2) in fp64, per-block partials (part [nblk]) summed in block order by the last arriver
 *   (`ticket`: one zeroed unsigned, re-armed by the kernel) -- deterministic;
 * aot_adamw_flat_f32: torch.optim.AdamW on tensor s = elements [seg_off[s], seg_off[s+1]) with hyp[s] = {lr, weight decay,
 *   1 - beta1^step, sqrt(1 - beta2^step)}; lr < 0 skips the tensor (`p.grad is None`); the clip_grad_norm_ factor
 *   min(1, max_norm / (sqrt(sumsq[0]) + 1e-6)) is taken from the device (max_norm <= 0 or sumsq NULL: none): no host sync. */
int aot_sumsq_flat_f64(const float* x, long n, double* part, int nblk, unsigned* ticket, double* out, void* stream);
int aot_adamw_flat_f32(float* p, const float* g, float* m, float* v, long n, const long* seg_off, const float* hyp, int nseg,
                       float beta1, float beta2, float eps, const double* sumsq, float max_norm, void* stream);

/* ---- training path, differentiable primitives (csrc/train_bwd.hip) ----
 * What `loss.backward()` (networks/managers/trainer.py:460-519) derives for the training engine's forward
 * (networks/engines/aot_engine.py:33-108) decomposes into these primitives and their adjoints; the host side wraps them as
 * torch.autograd.Function (networks/layers/train_ops.py).  Correctness-first kernels, fixed summation order. */

/* C[b][m][n] (ldc) = alpha * sum_k A[b][m][k] B[b][k][n] (+ bias[n]) (accumulate: C += ...) for operands with arbitrary ELEMENT
 * strides (sab, sam, sak / sbb, sbk, sbn): nn.Linear and 1x1 convs (transformer.py:321-359, fpn.py:34-58), the QK^T / PV
 * products of attention.py:92-117,672-707 per head, and -- on transposed views -- all of their gradients.  Exact fp32
 * (v_mfma_f32_32x32x2_f32, k-ordered). */
int aot_matmul_strided_f32(const float* a, const float* b, const float* bias, float* c, int batch, int M, int N, int K, long sab,
                           long sam, long sak, long sbb, long sbk, long sbn, long scb, int ldc, float alpha, int accumulate,
                           void* stream);
/* cols [B*OH*OW, KH*KW*C] = im2col of B NHWC maps [B*H*W, C] (C % 4 == 0; k = (ky*KW + kx)*C + c, zeros outside the image), and its
 * adjoint dx [B*H*W, C] = col2im(cols) as a gather.  KxK convolutions = im2col + matmul: fpn.py:41-56 (3x3), the identity bank
 * (models/aot.py:50-63: 17x17 / stride 16 on the one-hot or probability map). */
int aot_im2col_f32(const float* x, float* cols, int B, int H, int W, int C, int OH, int OW, int KH, int KW, int stride, int pad,
                   int dil, void* stream);
int aot_col2im_f32(const float* cols, float* dx, int B, int H, int W, int C, int OH, int OW, int KH, int KW, int stride, int pad,
                   int dil, void* stream);
/* Adjoint of aot_dwconv2d_nhwc_f32 (no bias, no activation): dx from dy and w [KH*KW, C]; dw [KH*KW, C] from dy and x.
 * basic.py:19-25,41-47 (5x5), mobilenetv2.py:93-98 (3x3, stride 1 / 2, dilation). */
int aot_dwconv2d_bwd_data_f32(const float* dy, const float* w, float* dx, int B, int H, int W, int C, int OH, int OW, int KH, int KW,
                              int stride, int pad, int dil, void* stream);
int aot_dwconv2d_bwd_weight_f32(const float* dy, const float* x, float* dw, int B, int H, int W, int C, int OH, int OW, int KH,
                                int KW, int stride, int pad, int dil, void* stream);
/* y = act(x) and dx = dy * act'(x) over n floats (AOT_ACT_*): F.relu / relu6 (mobilenetv2.py:44-45), F.gelu (basic.py:32), silu
 * (attention.py:585-586). */
int aot_act_f32(const float* x, float* y, long n, int act, void* stream);
int aot_act_bwd_f32(const float* x, const float* dy, float* dx, long n, int act, void* stream);
/* nn.LayerNorm backward over rows of contiguous [M, C]: dx, and xhat = (x - mean) * rstd for the parameter gradients. */
int aot_layernorm_bwd_f32(const float* x, const float* dy, const float* gamma, float* dx, float* xhat, int M, int C, float eps,
                          void* stream);
/* nn.GroupNorm backward over B lanes of contiguous [M, C] maps with the forward's statistics (aot_groupnorm_stats_f32): dx, xhat. */
int aot_groupnorm_bwd_f32(const float* x, const float* dy, const double* stats, const float* gamma, float* dx, float* xhat, int B,
                          int M, int C, int G, void* stream);
/* The same in two well-filled launches for long maps: per-chunk fp64 partials `part` [B*G][nchunk][2], summed in chunk order by the
 * last arriver of `ticket` [B*G] (zeroed unsigned, re-armed) into m12 [B*G][2]; then dx / xhat elementwise (C % 4 == 0). */
int aot_groupnorm_bwd2_f32(const float* x, const float* dy, const double* stats, const float* gamma, float* dx, float* xhat,
                           double* part, unsigned* ticket, float* m12, int B, int M, int C, int G, int nchunk, void* stream);
/* dgamma[c] = sum_r dy[r][c] * xhat[r][c], dbeta[c] = sum_r dy[r][c] over R rows of [R, C] (both norms). */
int aot_norm_param_grads_f32(const float* dy, const float* xhat, float* dgamma, float* dbeta, long R, int C, void* stream);
/* The same column reductions for long inputs (a batched training step reaches 1e5 rows): grid of (32-channel group, row chunk),
 * fp64 partials `part` [2][nchunk][C], summed in chunk order by the last arriver of each group's `ticket` (ceil(C/32) zeroed
 * unsigned, re-armed by the kernel) -- deterministic, one launch.  xhat / dgamma may be NULL (bias gradient = column sums:
 * what autograd derives for nn.Linear / nn.Conv2d biases, trainer.py:460-519). */
int aot_col_reduce_f32(const float* dy, const float* xhat, float* dgamma, float* dbeta, long R, int C, double* part,
                       unsigned* ticket, int nchunk, void* stream);
/* out [R, Cout] = the columns idx[0..Cout) (int32) of x [R, Cin]: the identity shuffle / un-shuffle of label maps and logits
 * (trainer.py:457, aot_engine.py:364-367: einsum with a 0 / 1 matrix there); its adjoint is the gather by the inverse permutation. */
int aot_gather_cols_f32(const float* x, const int* idx, float* out, long R, int Cin, int Cout, void* stream);
/* Operand copies of the weight-gradient GEMM dW = dy^T x (what autograd derives for nn.Linear, trainer.py:460-519) in one launch:
 * transpose != 0: dst [C, ldd] = src [R, lds]^T with columns R..Rpad-1 zero; transpose == 0: dst [Rpad, ldd] = the rows of src
 * followed by zero rows (C % 4 == 0). */
int aot_transpose_pad_f32(const float* src, float* dst, long R, int C, long lds, long ldd, long Rpad, int transpose, void* stream);
/* The general form: dst [Rpad, ldd] (columns < Cpad written) = the 2-D VIEW src[r * s_r + c * s_c] (element strides, any values >= 0:
 * row-major, transposed, broadcast), zeros outside R x C -- the operands of the attention products QK^T / PV of the training graph
 * and of their gradients (attention.py:99-110, 672-707 under autograd: torch.matmul on views), made dense and padded to the
 * GEMM kernels' reduction granule in one launch. */
int aot_copy2d_pad_f32(const float* src, float* dst, long R, long C, long s_r, long s_c, long Rpad, long Cpad, long ldd, void* stream);
/* y = softmax(x) over rows of length T (entries at -inf give 0) and dx = y * (dy - sum(dy * y)): attention.py:107,359,703,846. */
int aot_softmax_rows_f32(const float* x, float* y, long rows, int T, void* stream);
int aot_softmax_rows_bwd_f32(const float* y, const float* dy, float* dx, long rows, int T, void* stream);
/* Adjoint of aot_bilinear_nhwc_f32 (contiguous maps): dx [B*IH*IW, C] from dy [B*OH*OW, C]. */
int aot_bilinear_bwd_nhwc_f32(const float* dy, float* dx, int B, int IH, int IW, int OH, int OW, int C, int align_corners,
                              void* stream);
/* Window <-> dense layouts of the windowed attentions (local2global, attention.py:378-417,863-903) for G maps of N = h*w tokens:
 * gather: win [G][N][(2R+1)^2] = dense [G][N][N] at the window's keys (`fill` outside the image); scatter: the reverse (`fill`
 * outside the window).  Each is the other's adjoint (with fill = 0). */
int aot_window_gather_f32(const float* dense, float* win, int G, int h, int w, int max_dis, float fill, void* stream);
int aot_window_scatter_f32(const float* win, float* dense, int G, int h, int w, int max_dis, float fill, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AOT_HIP_H */
