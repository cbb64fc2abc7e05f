/*
 * diffpure_hip.h - C ABI of libdiffpure_hip.so, the MI355X (gfx950) kernels behind the DiffPure
 * purification hot path.
 *
 * The reference has no FFI of its own for this path: its device work is `torch.nn` calls that end
 * in cuDNN / cuBLAS / ATen kernels.  Each entry point below therefore replaces one *call site* of
 * the reference (cited as file:line under /root/reference) and is what a maintainer would bind
 * (ctypes / pybind / TORCH_LIBRARY - see INTEGRATION.md).
 *
 * Conventions
 *   - plain pointers to DEVICE memory, plain sizes; no torch types.
 *   - activations are NHWC fp32 ("pixel-major": [B][H][W][C]); weights are pre-packed by the host
 *     (see each function).  All float4-vectorised paths need the channel counts named below to be
 *     multiples of 4 and the base pointers 16-byte aligned.
 *   - `stream` is a hipStream_t passed as void*; NULL = the default stream.  Nothing synchronises.
 *   - return 0 on success, non-zero on error; dp_last_error() gives the message (thread-local).
 *   - no hidden allocations: scratch is passed in by the caller.
 */
#ifndef DIFFPURE_HIP_H
#define DIFFPURE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- library ------------------------------------------------------------------------------ */
int dp_abi_version(void);
const char* dp_last_error(void);

/* Tuning switches (csrc/dp_tune.h lists them: kernel-variant selectors, bit-identical in their results - all but
 * DIFFPURE_BATCH_INVARIANT, which decides whether a convolution's split of K may depend on the batch bucket: see
 * dp_conv2d_nhwc_h2_workspace).  Each is read
 * from the environment variable of the same name ONCE, on first use, and never again; dp_set_tuning() is the only way to
 * change one afterwards (probes, A/B tests).  Unknown name -> non-zero.  No reference counterpart (the reference tunes
 * through cudnn.benchmark, eval_sde_adv.py:303). */
int dp_set_tuning(const char* name, int value);
int dp_get_tuning(const char* name, int* value);

/* Per-launch timing of the convolution kernels with hipEvents recorded on the launch stream (bench.py's roofline leg).
 * dp_prof_enable(1) opens a recording window (previous records are discarded), dp_prof_enable(0) closes it (records are
 * kept); dp_prof_collect() synchronises the recorded events and returns, per launch kind, total milliseconds, launches
 * algorithmic FLOPs (2*M*N*K) and algorithmic HBM bytes (every operand and the output once), plus the number of launches that were NOT recorded because the window exceeded the
 * record buffer (65 536 launches) - callers must report a non-zero `dropped`. */
enum { DP_PROF_3X3_PP = 0,      /* 3x3 convolutions on the 256-wide tile kernels (8-wave / one-wave-per-SIMD / ping-pong: the dominant kernel) */
       DP_PROF_1X1 = 1,         /* 1x1 convolutions / linear layers, any kernel */
       DP_PROF_3X3_OTHER = 2,   /* 3x3 convolutions on the other tile variants (stem, head, split-K levels, small shapes) */
       DP_PROF_1X1_PP = 3,      /* 1x1 convolutions that ran on the ping-pong kernel (a rocprofv3 per-kernel total covers kinds 0 + 3) */
       DP_PROF_GN_APPLY = 4,    /* GroupNorm-apply launches (dp_gn_apply / dp_gn_apply_h16): the HBM-bound second kernel of a step; flop = 0,
                                   bytes = every input element once + every output element once */
       DP_PROF_3X3_DH = 5,      /* ABI 8: 3x3 convolutions on the 4-wave half-height kernel (conv_igemm_dh, un-split launches) - kinds 0 / 3 are then */
       DP_PROF_1X1_DH = 6,      /*        1x1 ones.  One kind per kernel NAME, so that a rocprofv3 per-kernel row covers exactly (0 + 3) or (5 + 6) */
       DP_PROF_KINDS = 7 };
int dp_prof_enable(int on);
int dp_prof_collect(double* ms, long long* n, double* flop, double* bytes, long long* dropped);

/* ---- convolution / linear: implicit GEMM on MFMA -------------------------------------------
 * Replaces nn.Conv2d / nn.Conv1d(k=1) / nn.Linear / NIN at
 *   guided_diffusion/unet.py:196 (in_layers conv), :224 (out_layers conv), :229-234 (skip),
 *   :295 (qkv), :302 (proj_out), :478-484, :623 (stem / head), :213-216 (emb linear),
 *   score_sde/models/layerspp.py:224,233,235 (Conv_0/1/2), :226 (Dense_0),
 *   score_sde/models/layers.py:552-555 (NIN), score_sde/models/ncsnpp.py:87-92 (temb MLP).
 *
 * out[m][n] = scale * ( res[m][n] + bias[n] + temb[b(m)][n] + sum_k A[m][k] * w[k][n] )
 *   m = (b, oy, ox) over B*H*W output pixels (stride 1, "same" zero padding, KH=KW in {1,3});
 *   k = ((ky*KW + kx) * Cin + ci), Cin = C1 + C2: the input is the channel concatenation of x1
 *   (C1 channels) and optional x2 (C2 channels) - the th.cat of unet.py:667 / ncsnpp.py:325 is
 *   never materialised;  w is [K][ldw] row-major (ldw >= N, ldw % 4 == 0, pad columns zero);
 *   bias [N] or NULL; temb [B][temb_stride] or NULL (temb_stride 0 broadcasts one row);
 *   res [M][ldr] or NULL; out [M][ldo].
 * Arithmetic: fp32 MFMA (v_mfma_f32_32x32x2_f32, exact fp32 products and accumulation).
 * out_fmt: 0 = `out` is fp32 [M][ldo]; 1 = `out` is PLAIN fp16 [M][ldo] (final value rounded to nearest; the stem convolution of a
 *   network whose residual stream is fp16, unet.py:478-484 under use_fp16).  colstats are those of the unrounded values.
 * colstats (optional): [ceil(M/512)*8][2][N] floats (one record per 64 rows, ROUNDED UP to whole 512-row tiles: a tile writes all
 *   the records of its rows, also those wholly beyond a ragged M); for every record of 64 consecutive output rows the
 *   per-column sum and sum of squares of the FINAL values, reduced inside the epilogue in an order that
 *   does not depend on the tile shape the dispatcher picks (bit-identical for any batch sharding).
 *   *tile_rows returns the record size (64).  dp_gn_finalize_cols turns the records into the GroupNorm
 *   statistics of the tensor without re-reading it.
 */
int dp_conv2d_nhwc(const float* x1, int C1, const float* x2, int C2,
                   int B, int H, int W, int KH, int KW,
                   const float* w, int ldw, int N,
                   const float* bias, const float* temb, int temb_stride,
                   const float* res, int ldr, float scale,
                   void* out, int ldo, int out_fmt,
                   float* colstats, int* tile_rows, void* stream);

/* The STEM of a score network (ABI 8): 3x3, stride 1, "same" zero padding, Cin = 3 -> N channels - guided_diffusion/unet.py:478-484
 * (3 -> 256 at 256^2), score_sde/models/ncsnpp.py:232-236 (3 -> 128 at 32^2) - as a WRITE-BOUND kernel: K = 27 is one 32-wide k-tile
 * of the fp16 matrix cores and the layer moves 2.15 GB of fp16 output against 50 MB of input at B = 64 (csrc/stem.hip).
 *   x [B][H][W][3] fp32 (the SDE state; no border - padding is resolved in the gather);
 *   w: [2][N][32] fp16 - the (hi, lo) split of the fp32 weights, hi = fp16(w), lo = fp16(w - hi), row n, k = (ky*3+kx)*3 + ci,
 *      columns 27..31 zero (diffpure_amd/ops.py:pack_stem_weight);  bias [N] fp32 or NULL;
 *   out [B*H*W][N] fp32 (out_fmt 0) or plain fp16 (out_fmt 1; rounded to nearest, the column records are those of the unrounded values).
 * Arithmetic: x and w as 22-bit (hi, lo) pairs, x_lo*w_hi + x_hi*w_lo + x_hi*w_hi in three v_mfma_f32_32x32x16_f16 passes, fp32
 * accumulation ("f16x3").  colstats / tile_rows as in dp_conv2d_nhwc (64-row records; M % 64 == 0 here, so no padding records are
 * written).  dp_conv2d_stem_ok: does the kernel serve this shape (Cin == 3, N in {128, 256}, B*H*W % 64 == 0)? - a function of the
 * layer and of nothing else a sharding of the batch could change. */
int dp_conv2d_stem_ok(int Cin, int B, int H, int W, int N);
int dp_conv2d_stem(const float* x, int Cin, int B, int H, int W, const void* w, int N, const float* bias,
                   void* out, int out_fmt, float* colstats, int* tile_rows, void* stream);

/* Same contract on the fp16 matrix cores with fp32-class accuracy ("f16x3"): every operand is a
 * (hi, lo) pair of fp16 numbers and every product is three v_mfma_f32_32x32x16_f16 passes
 * (a_lo*w_hi + a_hi*w_lo + a_hi*w_hi) into one fp32 accumulator.
 * "h2" format: channels in blocks of 8, each block = 8 fp16 hi followed by 8 fp16 lo (32 bytes):
 *   activations x: [B][H+2][W+2][C/8][2][8] fp16 WITH A ONE-PIXEL ZERO BORDER (so that the loader
 *   has no bounds tests), produced by dp_gn_apply(out_fmt=1); C % 32 == 0;
 *   weights w: [N][K/8][2][8] fp16 in the kernel's reduction order
 *   k' = (c32*KS*KS + (ky*KS+kx))*32 + ci%32, c32 = ci/32 (channel slices outermost, taps innermost:
 *   the 9 taps re-read the same halo strip from L1/L2), produced by dp_pack_h2 from the fp32 panel
 *   [N][K'] (diffpure_amd/ops.py:pack_conv_weight_h2).
 *   H, W are the OUTPUT (= interior) sizes; bias/temb/res/out are fp32 as above.
 * a_fmt / passes select the activation operand and the arithmetic per product (all accumulate in fp32):
 *   a_fmt 0 = x in h2 form (above), passes 3  = a_lo*w_hi + a_hi*w_lo + a_hi*w_hi   ("f16x3": 22-bit operands)
 *                                   passes 12 = a_lo*w_hi + a_hi*w_hi               ("f16x2w": weights rounded to fp16;
 *                                                                                     precision study only)
 *   a_fmt 1 = x in "h1" form: PLAIN fp16 [B][H+2][W+2][C] with the same zero border (dp_gn_apply(out_fmt=2)), i.e. the
 *             activations are rounded to fp16 once; weights stay h2:
 *                                   passes 2  = a_hi*w_lo + a_hi*w_hi               ("f16x2": weights kept to 22 bits)
 *                                   passes 1  = a_hi*w_hi                           ("f16": the arithmetic of the reference's
 *             own use_fp16 torso, /root/reference/configs/imagenet.yml:18, guided_diffusion/unet.py:626-632, with fp32
 *             accumulation and fp32 GroupNorm)
 *   w_fmt 0 = weights in h2 form (above); w_fmt 1 (with a_fmt 1, passes 1) = PLAIN fp16 weights in the same k' order, stored
 *             in blocks of 32 rows x 8 elements: element (n, k') at index (((n/32) * (K/8) + k'/8) * 32 + n%32) * 8 + k'%8,
 *             N rounded up to 32 rows (zero rows) - the B fragment of one 32x32x16 MFMA is 1 KB of contiguous memory
 *             (diffpure_amd/ops.py:order_conv_weight_w16): "f16" when rounded to nearest once at load, "f16sr" when the panel is re-rounded STOCHASTICALLY from
 *             the fp32 masters before every network call (dp_round_weights) - the rounding error of the weights then
 *             changes from call to call and averages out over the solver steps like the activation rounding does,
 *             instead of accumulating coherently as a fixed perturbation of the model.
 *   Measured on the full loops (tests/probes/precision_loops.py, 100 EM steps, max-abs on purified pixels vs the exact
 *   fp32 engine): f16x3 3.9e-6, f16x2 1.3e-4, f16x2w 1.0e-3, f16 1.0e-3 (guided 256^2); 2.3e-6 / 9.4e-5 / 8.6e-4 / 8.6e-4
 *   (NCSN++): rounding the WEIGHTS is what costs accuracy (a fixed perturbation of the model, coherent over the steps),
 *   rounding the activations averages out; "f16sr": 2.2e-4 (guided) / 1.6e-4 (NCSN++) against the reference modules
 *   (tests/probes/sr_weights_probe.py, tests/test_gpu_loops.py).
 *   out_fmt 0 = `out` is fp32 [M][ldo]; out_fmt 1 = `out` is PLAIN fp16 [M][ldo] (the final fp32 value - bias, temb, residual
 *             and scale applied - rounded to nearest even; ldo even, N % 4 == 0).  For a tensor whose only consumer is a
 *             GroupNorm-apply pass that emits an fp16 operand anyway (the first convolution of every ResBlock:
 *             guided_diffusion/unet.py:244-264 in_layers -> out_layers[0]): half the store bytes here, half the load bytes
 *             there (dp_gn_apply_h16).  `colstats` are the sums of the UNROUNDED values in both formats.
 *   res_fmt 0 = `res` is fp32 [M][ldr]; res_fmt 1 = `res` is PLAIN fp16 [M][ldr] (ldr even).  Together with out_fmt 1 this is the
 *             fp16 RESIDUAL STREAM of the fp16 x fp16 modes (ABI 6): the block outputs h of the UNet travel as fp16, which is the
 *             reference's own arithmetic for the ImageNet model (`use_fp16: True`, configs/imagenet.yml:18; convert_to_fp16 casts
 *             the torso, guided_diffusion/unet.py:626-632, fp16_util.py:23-40) - here with fp32 accumulation, fp32 epilogue
 *             arithmetic and fp32 GroupNorm statistics on top.
 *   seg1 / seg2 (ABI 6, with w_fmt 1): 1x1 "skip" K-SEGMENTS.  After the KS*KS*C reduction over x the k-loop continues over the
 *             segC1 (+ segC2) channels of up to two PLAIN fp16 NHWC tensors [B][H][W][segC] (no border; segC % 32 == 0), whose
 *             weight columns FOLLOW in the same panel: w is [N32][KS*KS*C + segC1 + segC2] in the block layout above, and
 *             out += [seg1 | seg2] . w[:, KS*KS*C:].  This folds a ResBlock's 1x1 skip_connection (unet.py:223-230, 262-264;
 *             layerspp.py:268-272: Conv_2 / NIN shortcut) over its raw input - the channel concatenation of the two skip
 *             sources - into its second 3x3 convolution: the skip tensor is never written or re-read (`bias` then carries the sum
 *             of both biases).  Every fp16 x fp16 tile variant the dispatcher can pick for such a launch has a segment loader (the
 *             8-wave kernel, the one-wave-per-SIMD kernel, the generic tiles incl. split-K) and all give identical bits, so whether a
 *             layer is fused is a property of the LAYER, never of the batch: dp_conv2d_nhwc_h2_takes_segments(). */
int dp_conv2d_nhwc_h2(const void* x, int C, int B, int H, int W, int KS,
                      const void* w, int N,
                      const float* bias, const float* temb, int temb_stride,
                      const void* res, int ldr, float scale,
                      void* out, int ldo, float* colstats, int* tile_rows,
                      void* work, long long work_bytes, int passes, int a_fmt, int w_fmt, int out_fmt, int res_fmt,
                      const void* seg1, int segC1, const void* seg2, int segC2, void* stream);
/* 1 when an fp16 x fp16 launch (a_fmt 1, w_fmt 1, passes 1) of this layer shape may carry 1x1 K-segments of segC1 (+ segC2) channels
 * (whole 32-channel slices everywhere), else 0.  A function of the layer shape only. */
int dp_conv2d_nhwc_h2_takes_segments(int H, int W, int KS, int C, int N, int segC1, int segC2);
/* Scratch the call above needs (0 for most launches): split-K - partial sums per k-range, then one reduction + epilogue pass.
 *   Shape rule: the low-resolution levels (H*W <= 64), with a factor that depends on the layer shape only.
 *   Batch rule (ABI 8, default): launches of fewer than 128 tiles of 128 x 256 with long reductions - the big levels at small per-GPU
 *   batches, e.g. the reference's own 4 images per GPU (run_scripts/imagenet/run_in_rand_inf.sh:16) - are split by a power of two chosen
 *   from the tile count, i.e. per (layer shape, batch bucket); a sample's low-order bits then depend on the bucket.  The tuning switch
 *   DIFFPURE_BATCH_INVARIANT=1 (dp_set_tuning / environment) keeps the shape rule alone: bit-identical results for any batch sharding. */
long long dp_conv2d_nhwc_h2_workspace(int B, int H, int W, int KS, int C, int N);
/* ---- fused block boundary of the <= 64-pixel levels (ABI 8; csrc/boundary.hip) -------------------------------------------------
 * Replaces, per ResBlock boundary at 8x8 / 4x4 (score_sde/models/layerspp.py:242-274, guided_diffusion/unet.py:244-264), the launch
 * chain  convolution partials -> split-K reduction + epilogue -> GroupNorm statistics -> GroupNorm-apply  by  partials -> ONE launch.
 *
 * dp_conv2d_nhwc_h2_partials: the convolution of dp_conv2d_nhwc_h2 (same operand formats, same kernels) for a layer that is reduced
 *   with split-K (dp_conv2d_nhwc_h2_workspace() > 0: H*W <= 64), stopping after the partial sums: work[s][B*H*W][N] fp32,
 *   s = 0 .. *n_parts - 1.
 * dp_splitk_epilogue: the reduction + epilogue on its own (what dp_conv2d_nhwc_h2 runs behind its partial sums): out = scale * (res +
 *   temb[b] + bias + sum_s work[s]), column records as dp_conv2d_nhwc_h2 - for a consumer the fused form does not serve.
 * dp_splitk_gn: reduction + epilogue + GroupNorm (+FiLM) (+SiLU) of cat(out, x2) in one launch.  One workgroup per (sample, 128- or
 *   256-channel block of whole groups) keeps the sample's slab in registers.
 *     out (optional): the stream tensor [B*H*W][N], fp32 (out_fmt 0) or plain fp16 (out_fmt 1; out_fmt also names the format the
 *       VALUE THAT IS NORMALISED has - the fp16-rounded one for a fp16 stream - when out itself is not wanted);
 *     colstats (optional, H*W == 64): [B][2][N], the 64-row column records of out (one per sample);
 *     x2 (optional): second source [B][H*W][C2] fp32 (x2_fmt 0) / plain fp16 (1): the other half of a skip concatenation;
 *     stats (optional): [B][G][2] (mean, rstd) of cat(out, x2) - statistics of the UNROUNDED values, formed as dp_gn_finalize forms them;
 *     y: the operand [B][H+2][W+2][N + C2] plain fp16, zero border written; y_raw (optional): the un-normalised cat(out, x2) in the
 *       same form (input of a 1x1 shortcut convolution).
 *   dp_splitk_gn_ok: H*W in {64, 16}, channel blocks of whole groups, (N + C2) / G a multiple of 4.  A function of the layer shape only. */
int dp_conv2d_nhwc_h2_splits_by_shape(int H, int W, int KS, int C, int N);     /* 1: split by the shape-only rule (<= 64-pixel levels) */
int dp_conv2d_nhwc_h2_partials(const void* x, int C, int B, int H, int W, int KS, const void* w, int N, void* work, long long work_bytes,
                               int passes, int a_fmt, int w_fmt, const void* seg1, int segC1, const void* seg2, int segC2, int* n_parts,
                               void* stream);
int dp_splitk_epilogue(const float* work, int n_parts, int B, int H, int W, int N, const float* bias, const float* temb, int temb_stride,
                       const void* res, int res_fmt, float scale, void* out, int out_fmt, float* colstats, int* tile_rows, void* stream);
int dp_splitk_gn_ok(int H, int W, int N, int C2, int G);
int dp_splitk_gn(const float* work, int n_parts, int B, int H, int W, int N, const float* bias, const float* temb, int temb_stride,
                 const void* res, int res_fmt, float scale, void* out, int out_fmt, float* colstats, const void* x2, int x2_fmt, int C2,
                 int G, float eps, const float* gamma, const float* beta, const float* fscale, const float* fshift, int film_stride,
                 int act, float* stats, void* y, void* y_raw, void* stream);

/* fp32 [rows][ld] (first `cols` columns, cols % 8 == 0) -> h2 [rows][cols/8][2][8] fp16. */
int dp_pack_h2(const float* src, long long rows, int cols, int ld, void* dst, void* stream);
/* fp32 -> fp16 over a flat buffer of n elements (n % 8 == 0; every fp16 weight panel of a network lives in ONE buffer so
 * that this is one launch per network call).  stochastic = 0: round to nearest even ("f16").  stochastic = 1 ("f16sr"):
 * unbiased stochastic rounding - a value between two fp16 neighbours goes to either with probability proportional to its
 * distance from the other - with Philox4x32-10 bits keyed by (seed, key, element index): the same (seed, key) always gives
 * the same panel (results stay reproducible and independent of batch sharding), different keys give independent roundings
 * (the solver passes the step index). */
int dp_round_weights(const float* src, void* dst, long long n, int stochastic, unsigned long long seed, long long key,
                     void* stream);

/* ---- strided batched GEMM (attention cores, forward and backward) --------------------------------
 * Replaces the einsums at unet.py:355-359 / :389-396 and layerspp.py:82,86 (and their autograd).
 * C[z][m][n] = alpha * sum_k Aop[z][m][k] * Bop[z][k][n];  z = zb*ZH + zh, and each operand's batch
 * offset is zb*s?b + zh*s?h (elements).
 * transA = 0: A stored [M][lda]; 1: A stored [K][lda].  transB = 0: B stored [K][ldb]; 1: [N][ldb].
 * The contiguous extent of each operand (K or M for A, N or K for B) and lda/ldb must be % 4 == 0. */
int dp_gemm_strided(const float* A, int lda, long long sAb, long long sAh, int transA,
                    const float* B, int ldb, long long sBb, long long sBh, int transB,
                    float* C, int ldc, long long sCb, long long sCh,
                    int M, int N, int K, int ZB, int ZH, float alpha, void* stream);

/* The same contract on the fp16 matrix cores (round 5): the fp32 operands are rounded to fp16 (nearest) on their way into LDS, one
 * fp16 MFMA pass per product, fp32 accumulation, fp32 result - the arithmetic the input-gradient convolutions of the fp16 x fp16
 * precision modes already use.  Serves the attention BACKWARD of those modes (the recomputed scores and dV, dP, dQ, dK: torch autograd
 * through unet.py:345-362 / layerspp.py:75-91 under the reference's use_fp16 torso runs them in fp16 as well).  Shapes:
 * dp_gemm_strided_h16_ok(M, N, K) (M % 128 == 0, N % 128 == 0, K % 32 == 0); anything else stays on dp_gemm_strided.
 * a_fmt / b_fmt: 0 = the operand is fp32 in memory, 1 = it is plain fp16 already (q, k, v inside the fp16 qkv tensor of the taped forward:
 * read in place, no up-conversion pass); lda / ldb and the batch strides count elements of the operand's own type (fp16: multiples of 8). */
int dp_gemm_strided_h16_ok(int M, int N, int K);
int dp_gemm_strided_h16(const void* A, int a_fmt, int lda, long long sAb, long long sAh, int transA,
                        const void* B, int b_fmt, int ldb, long long sBb, long long sBh, int transB,
                        float* C, int ldc, long long sCb, long long sCh,
                        int M, int N, int K, int ZB, int ZH, float alpha, void* stream);

/* Row softmax in place, rows x cols fp32 (unet.py:358 `th.softmax(weight.float(), -1)`,
 * layerspp.py:84). */
int dp_softmax_rows(float* x, long long rows, int cols, void* stream);

/* ---- GroupNorm (+FiLM) (+SiLU) (+2x resample) -----------------------------------------------
 * Replaces GroupNorm32 + SiLU + Upsample/Downsample at unet.py:193-195, :218-221, :245-250,
 * :258-261, :293, :621-622 and nn.GroupNorm + act + naive_{up,down}sample_2d at
 * layerspp.py:217,230,243-258,268, :66,77, ncsnpp.py:225-227,372.
 *
 * Step 1  dp_gn_stats:   partial[b][s][g] = (sum, sumsq) over pixel slab s of (x1|x2) (float2)
 * Step 2  dp_gn_finalize: stats[b][g] = (mean, rstd) in float, combined in double
 * Step 3  dp_gn_apply:   y = resample( act( ((x-mean)*rstd*gamma+beta) * (1+fscale) + fshift ) )
 *   C = C1 + C2, C % (4*G) == 0; nsplit chosen by the caller (scratch = B*nsplit*G*2 floats).
 *   fscale/fshift: [B][film_stride] rows or NULL (film_stride 0 broadcasts one row).
 *   gamma == NULL skips the normalisation (pure act/resample of x: the x-branch of a
 *   resampling ResBlock, unet.py:249 / layerspp.py:249,256).
 *   act: 0 none, 1 SiLU.  resample: 0 none, 1 nearest x2 (out 2H x 2W), 2 mean 2x2 (out H/2 x W/2),
 *   3 / 4 = FIR x2 up / down-sampling of score_sde's `fir: True` networks (up_or_down_sampling.py:203-265, i.e.
 *   upfirdn2d, op/upfirdn2d_kernel.cu:107-207, for a separable 4-tap filter): fir4 = the 1-D taps normalised to sum 1
 *   (HOST pointer, read at call time; NULL for the other modes), zeros outside the image.
 *   out_fmt: 0 = fp32 NHWC [B][Ho][Wo][C]; 1 = "h2" split-fp16 with a one-pixel zero border,
 *   [B][Ho+2][Wo+2][C] (see dp_conv2d_nhwc_h2); 2 = "h1" plain fp16 with the same border (a_fmt 1 there).
 *   y_raw (optional, out_fmt=1 and resample=0 only): second output = the UN-normalised input
 *   cat(x1, x2) in the same bordered h2 form, the operand of a 1x1 skip convolution
 *   (unet.py:229-234 / layerspp.py:235) - written in the same pass that reads it.
 */
int dp_gn_stats(const float* x1, int C1, const float* x2, int C2, int B, int HW, int G,
                int nsplit, float* partial, void* stream);
int dp_gn_finalize(const float* partial, int B, int nsplit, int G, long long count, float eps,
                   float* stats, void* stream);
/* Step 1+2 without reading the tensor: statistics of cat(x1, x2) from the column partials the producing
 * convolutions wrote (dp_conv2d_nhwc[_h2] colstats).  HW % tile_rows == 0 for every source. */
int dp_gn_finalize_cols(const float* cs1, int C1, int tile_rows1, const float* cs2, int C2, int tile_rows2,
                        int B, int HW, int G, float eps, float* stats, void* stream);
int dp_gn_apply(const float* x1, int C1, const float* x2, int C2, int B, int H, int W, int G,
                const float* stats, const float* gamma, const float* beta,
                const float* fscale, const float* fshift, int film_stride,
                int act, int resample, int out_fmt, void* y, void* y_raw, const float* fir4, void* stream);
/* Step 3 over PLAIN fp16 tensors (ABI 6; replaces dp_gn_apply_f16in): x1 / x2 are [B][H][W][C1 | C2] fp16 without a border - a
 * first convolution's fp16 output (dp_conv2d_nhwc_h2 out_fmt 1) or, in the fp16 x fp16 modes, the residual stream itself (the
 * reference's own use_fp16 arithmetic keeps h in fp16: guided_diffusion/unet.py:626-632) and the skip tensors of the up path
 * (the th.cat of unet.py:667 is two source pointers).  Same contract as dp_gn_apply otherwise: optional normalisation (gamma ==
 * NULL: none) + FiLM + SiLU, resample 0 | 1 (nearest x2) | 2 (mean 2x2) - the FIR modes stay on the fp32 stream -, optional raw
 * second output (out_fmt 2 only): with resample 0 the un-normalised input in the same bordered operand form (input of a 1x1 skip
 * convolution); with resample 1 | 2 (ABI 8) the RESAMPLED un-normalised input as a plain tensor [B][Ho][Wo][C] - the identity skip of an
 * up / down ResBlock (unet.py:245-250), formed from the values the pass holds in registers instead of by a second launch.  out_fmt 2 = the zero-bordered "h1" operand [B][Ho+2][Wo+2][C]; out_fmt 3 = a plain fp16 tensor [B][Ho][Wo][C]
 * (the resampled identity skip of an up / down ResBlock, unet.py:245-250: the residual of its second convolution, res_fmt 1).
 * C1 % 8 == 0, C % 8 == 0.  Same arithmetic per element as dp_gn_apply(out_fmt 2) on the up-converted tensors (identical bytes):
 * 4 HBM bytes per element instead of 6. */
int dp_gn_apply_h16(const void* x1, int C1, const void* x2, int C2, int B, int H, int W, int G, const float* stats,
                    const float* gamma, const float* beta, const float* fscale, const float* fshift, int film_stride,
                    int act, int resample, int out_fmt, void* y, void* y_raw, void* stream);

/* ---- small elementwise pieces ----------------------------------------------------------------*/
/* y = x * sigmoid(x)   (nn.SiLU on the embedding vector, unet.py:211, layerspp.py:265) */
int dp_silu(const float* x, float* y, long long n, void* stream);
/* out = a*x + b*y      (forward diffusion, runners/diffpure_sde.py:223) */
int dp_axpby(const float* x, float a, const float* y, float b, float* out, long long n, void* stream);
/* emb[i][:] = sinusoid(t[i] * freqs[:half]); cos_first=1: [cos|sin] (guided nn.py:111-129),
 * 0: [sin|cos] (score_sde layers.py:515-529).  freqs is the host-computed table. */
int dp_timestep_embedding(const float* t, int n, const float* freqs, int half, int cos_first,
                          float* emb, void* stream);

/* ---- solver steps -----------------------------------------------------------------------------
 * One fixed step of the reverse VP-SDE (Euler-Maruyama) or of the probability-flow ODE (Euler),
 * fused over all pixels.  Replaces RevVPSDE.f/.g (runners/diffpure_sde.py:86-147) + torchsde's
 * Euler.step, and VPODE.ode_fn (runners/diffpure_ode.py:90-122) + torchdiffeq's Euler step:
 *   score = score_div ? (-eps)/score_coef : score_coef*eps
 *   drift = neg_half_beta*x - gg*score          (gg = g^2 for the SDE, 0.5*g^2 for the ODE)
 *   x_new = x + (-drift)*h + g*(z*sqrt_h)       (g = 0: no noise is drawn or read)
 * x: [B][HW][C] state; eps: network output [B][HW][eps_ld], first C channels used; x_out may alias
 * x.  Noise z: read from `noise` ([B][HW][C]) if non-NULL, else Philox4x32-10 + Box-Muller keyed
 * by (seed, global sample index = sample0 + b, step, element) so that any sharding of the batch
 * over GPUs draws the same stream.  All scalars are computed by the host in float32 exactly as the
 * reference computes them. */
int dp_em_step(const float* x, const float* eps, int eps_ld, int B, int HW, int C,
               float neg_half_beta, float gg, float score_coef, int score_div,
               float h, float g, float sqrt_h,
               const float* noise, unsigned long long seed, long long sample0, int step,
               float* x_out, void* stream);
/* Standard normals from the same Philox stream (used for the forward-diffusion noise `e`,
 * diffpure_sde.py:217; step = -1 by convention). out: [B][per_sample]. per_sample % 4 == 0. */
int dp_philox_normal(float* out, int B, long long per_sample, unsigned long long seed,
                     long long sample0, int step, void* stream);
/* DDPM ancestral step with learned-range variance and clipped x0
 * (gaussian_diffusion.py:266-334, :438-446). out6: [B][HW][2C] (eps | v). z as in dp_em_step. */
int dp_ddpm_step(const float* x, const float* out6, int B, int HW, int C,
                 float sqrt_recip_ac, float sqrt_recipm1_ac, float coef1, float coef2,
                 float min_log, float max_log, int nonzero,
                 const float* noise, unsigned long long seed, long long sample0, int step,
                 float* x_out, void* stream);

/* ---- backward (adjoint-ODE, dL/dx only; runners/diffpure_ode.py:229-238 odeint_adjoint) ----------
 * Input gradient of dp_gn_apply's operator (see there for the forward):
 *   da = resample^T(dy); du = da*act'(u); dxh = du*(1+fscale)*gamma;
 *   dx = rstd*(dxh - mean_g(dxh) - xh*mean_g(dxh*xh)).
 * H, W = INPUT resolution of the forward; dy: [B][Ho][Wo][C] fp32 (Ho, Wo per `resample`).
 * x_fmt (ABI 7): 0 = x1 / x2 are fp32 [B][H][W][C1 | C2]; 1 = PLAIN fp16 - the taped forward of the fp16 x fp16 modes keeps the fp16
 * residual stream (round 5: the adjoint re-runs exactly the network the forward solve evaluated), so the tape holds fp16 tensors and
 * the backward reads them as stored; dy, the sums and dx stay fp32.
 * dp_gn_bwd_stats: slab partials [B][nsplit][G][2] -> sums [B][G][2] = the two group means.
 * dp_gn_bwd_apply: dx1 [B][H][W][C1] (+ dx2 [B][H][W][C2]) fp32, or with out_fmt=1 (C2 == 0) dx1 in
 * the zero-bordered h2 operand format, out_fmt=2 in the zero-bordered plain-fp16 ("h1") operand format, ready for the
 * next dgrad convolution (three-pass / one-pass fp16 matrix path).  add1 / add2 (ABI 7; optional, fp32 output only): a second
 * gradient arriving at the same tensors - the skip branch of a ResBlock (unet.py:262-264 / layerspp.py:272-274) - is added in the same
 * pass, dx += add_scale * add, as dp_gn_bwd_fused does for the small feature maps. */
int dp_gn_bwd_stats(const void* x1, int C1, const void* x2, int C2, int x_fmt, int B, int H, int W, int G,
                    const float* stats, const float* gamma, const float* beta,
                    const float* fscale, const float* fshift, int film_stride, int act, int resample,
                    const float* fir4, const float* dy, int nsplit, float* partial, float* sums, void* stream);
int dp_gn_bwd_apply(const void* x1, int C1, const void* x2, int C2, int x_fmt, int B, int H, int W, int G,
                    const float* stats, const float* gamma, const float* beta,
                    const float* fscale, const float* fshift, int film_stride, int act, int resample,
                    const float* fir4, const float* dy, const float* sums, int out_fmt, void* dx1, float* dx2,
                    const float* add1, const float* add2, float add_scale, void* stream);
/* The same operator in ONE pass over x and dy (ABI 6), for small feature maps - CIFAR-10 sizes: one workgroup per (sample, block of
 * whole groups) keeps its pixels in registers between the group sums and the update, so x and dy are read once instead of twice and
 * three launches become one.  dp_gn_bwd_fused_ok(): does the tensor shape fit (a function of the shape only; resample 0 | 1 | 2 - the
 * FIR modes keep the three-launch form)?  add1 / add2 (optional, fp32 output only): a second gradient arriving at the same tensors
 * (the skip branch of a ResBlock, unet.py:262-264 / layerspp.py:272-274) is added in the same pass (dx += add_scale * add), replacing a dp_add / dp_axpby launch. */
int dp_gn_bwd_fused_ok(int H, int W, int C1, int C2, int G, int resample);
int dp_gn_bwd_fused(const void* x1, int C1, const void* x2, int C2, int x_fmt, int B, int H, int W, int G,
                    const float* stats, const float* gamma, const float* beta,
                    const float* fscale, const float* fshift, int film_stride, int act, int resample,
                    const float* dy, int out_fmt, void* dx1, float* dx2, const float* add1, const float* add2, float add_scale, void* stream);
/* Adjoint of the plain 2x resamplers: mode 1 (forward was nearest x2): dx[Ho/2][Wo/2] = sum of the
 * 2x2 dy block; mode 2 (forward was mean 2x2): dx[2Ho][2Wo] = 0.25 * dy[y/2][x/2]; modes 3 / 4 (forward was the
 * FIR x2 up / down of dp_gn_apply with taps fir4[4]): the transposed stencils
 *   up:   dx[i]  = 2 (k0 dy[2i-1] + k1 dy[2i] + k2 dy[2i+1] + k3 dy[2i+2])      per axis, dx [Ho/2][Wo/2]
 *   down: dx[2n] = k2 dy[n] + k0 dy[n-1],  dx[2n+1] = k1 dy[n] + k3 dy[n+1]      per axis, dx [2Ho][2Wo]
 * (the adjoint of up_or_down_sampling.py:203-265 that torch autograd derives for the reference's upfirdn2d).
 * fir4 may be NULL for modes 1 / 2; dp_gn_bwd_stats / _apply take the same `resample` + `fir4` pair. */
int dp_resample_bwd(const float* dy, int B, int Ho, int Wo, int C, int mode, const float* fir4, float* dx, void* stream);
/* Softmax backward in place on dP given P: dS = P * (dP - sum_j dP_j P_j) per row. */
int dp_softmax_bwd_rows(const float* p, float* dp, long long rows, int cols, void* stream);
/* out = a + b (gradient accumulation at fan-out points). n % 4 == 0. */
int dp_add(const float* a, const float* b, float* out, long long n, void* stream);

/* Fused attention for the inference path (no probabilities kept): out[b,t,h*d+:] = softmax(q k^T / sqrt(d)) v per
 * head on the fp16 matrix cores, flash-style (the T x T scores stay in registers).  layout 0 = 'legacy' (heads x [q|k|v],
 * QKVAttentionLegacy unet.py:345-362), 1 = 'split' ([Q|K|V]); head dimension 64 (T % 64 == 0) or 256 (T % 128 == 0).
 * qkv_fmt 0: qkv [B,T,3C] fp32, split-fp16 operands, three MFMA passes per product (fp32-class accuracy);
 *            work: 3 * B * T * C * 4 bytes of 16-byte-aligned scratch (packed Q, K and transposed V).
 * qkv_fmt 1: qkv [B,T,3C] plain fp16 (the qkv convolution's out_fmt 1), ONE fp16 MFMA pass per product - the arithmetic of the
 *            reference's use_fp16 attention (unet.py:358-361); Q and K are read in place, work (B * T * C * 2 bytes) holds V^T.
 * out_fmt 0: out is fp32 [B][T][C].  out_fmt 1: the tokens are the pixels of an image of width W (T % W == 0) and out is the
 * zero-bordered plain-fp16 operand [B][T/W + 2][W + 2][C] of the proj_out 1x1 convolution (a_fmt 1 of dp_conv2d_nhwc_h2;
 * AttentionBlock.proj_out, unet.py:312): the kernel writes the interior pixels (rounded to nearest), the CALLER zeroes
 * the border. */
int dp_attention_fused(const void* qkv, int qkv_fmt, int B, int T, int C, int n_heads, int layout, void* out, int out_fmt, int W,
                       void* work, void* stream);

/* ---- the steps either side of the purifier ---------------------------------------
 * y = (bilinear(x) + shift) * scale, PyTorch semantics of F.interpolate(mode='bilinear',
 * align_corners=False), with a free choice of layouts (in_nhwc / out_nhwc: 0 = NCHW, 1 = NHWC).
 * Replaces, in SDE_Adv_Model.forward (/root/reference/eval_sde_adv.py):
 *   :74-75 interpolate 224->256 + :78 (x - 0.5) * 2 + the NCHW->NHWC repack   (shift -0.5, scale 2)
 *   :81-82 interpolate 256->224 + :89 (x_re + 1) * 0.5 + the NHWC->NCHW repack (shift  1 , scale 0.5)
 * x: [B,C,Hi,Wi] or [B,Hi,Wi,C] fp32; y: [B,C,Ho,Wo] or [B,Ho,Wo,C] fp32. */
int dp_resize_affine(const float* x, int B, int C, int Hi, int Wi, int in_nhwc, float shift, float scale, float* y,
                     int Ho, int Wo, int out_nhwc, void* stream);
/* Its adjoint, dx = scale * bilinear^T(dy) (what autograd runs for those lines on the adaptive-attack path),
 * gather form, deterministic. dy has the forward's OUTPUT shape/layout, dx the forward's INPUT shape/layout. */
int dp_resize_affine_bwd(const float* dy, int B, int C, int Ho, int Wo, int out_nhwc, float scale, float* dx, int Hi,
                         int Wi, int in_nhwc, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DIFFPURE_HIP_H */
