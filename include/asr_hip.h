/*
 * asr_hip.h -- C ABI of libasr_hip.so: hand-written gfx950 (MI355X / CDNA4) kernels for the
 * Augmented Super-Resolution hot path of nicoloalbergoni/DeepLabV3Plus-Augmented-SuperResolution.
 *
 * The reference is 100 % Python on TensorFlow 2.7 / tensorflow-addons 0.15 and has no FFI or
 * operator-plugin interface; each entry point below replaces one (group of) stock TF/TFA/Keras
 * op call site(s), cited as file:line relative to the reference checkout.  A Python
 * maintainer binds them with ctypes.
 *
 * Conventions
 *  - every function returns int: ASR_OK (0) or a negative ASR_ERR_*; asr_last_error() returns a
 *    thread-local message for the last failing call on this thread;
 *  - all array pointers are CALLER-OWNED DEVICE pointers (hipMalloc / torch ROCm tensors); the
 *    library allocates nothing and keeps no unsynchronised global state -- its only caches (a kernel's dynamic-LDS
 *    allowance, a device's CU count) are per-device atomics, and no environment variable changes which kernel runs --
 *    so it is callable from several host threads / streams / devices concurrently; scratch space is passed in explicitly;
 *  - every call takes an explicit stream (hipStream_t passed as void*) and is asynchronous
 *    with respect to the host;
 *  - layouts are dense row-major float32, images NHWC; "ld*" arguments are the element stride
 *    between consecutive pixels (>= channels) so outputs can land inside concat buffers;
 *  - projective transforms are 8 floats [a0,a1,a2,b0,b1,b2,c0,c1] exactly as
 *    ImageProjectiveTransformV3 takes them (output pixel (x,y) reads input at
 *    ((a0 x + a1 y + a2)/k, (b0 x + b1 y + b2)/k), k = c0 x + c1 y + 1), BILINEAR, out-of-range
 *    taps read 0.
 */
#ifndef ASR_HIP_H
#define ASR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ASR_ABI_VERSION 3

#define ASR_OK 0
#define ASR_ERR_INVALID_ARG (-1)
#define ASR_ERR_UNSUPPORTED (-2)
#define ASR_ERR_HIP (-3)
#define ASR_ERR_WORKSPACE (-4)

typedef void* asr_stream_t; /* hipStream_t */

const char* asr_last_error(void);
int asr_abi_version(void);
const char* asr_target_arch(void);

/* ------------------------------------------------------------------------------------------
 * Augmentation / warps
 * ------------------------------------------------------------------------------------------ */

/* One ImageProjectiveTransformV3 pass.  Replaces tfa.image.rotate / tfa.image.translate
 * (superresolution_scripts/augmentation_utils.py:22-25, superresolution.py:61-64,142-147).
 * src [n or 1, h_in, w_in, c] (src_batched = 0: one shared source), transforms [n or 1, 8]
 * (tf_batched = 0: one shared transform), dst [n, h_out, w_out, c]. */
int asr_warp_affine_f32(const float* src, float* dst, const float* transforms, int n, int src_batched,
                        int tf_batched, int h_in, int w_in, int h_out, int w_out, int c, asr_stream_t stream);

/* The same pass with interpolation "NEAREST" (tfa.image.rotate / translate of the label maps,
 * check_robustness.py:45-50): dst = src(round(in_y), round(in_x)) with std::round, 0 outside. */
int asr_warp_affine_nearest_f32(const float* src, float* dst, const float* transforms, int n, int src_batched,
                        int tf_batched, int h_in, int w_in, int h_out, int w_out, int c, asr_stream_t stream);

/* copies[i] = translate(rotate(image, rot_tf[i]), trans_tf[i]) -- tile + two bilinear resamplings
 * fused (superresolution_scripts/augmentation_utils.py:12-25 create_augmented_copies; :46-54
 * chunked variant).  image [h,w,c] (c = 1 or 3), copies [n,h,w,c], rot_tf / trans_tf [n,8]. */
int asr_augment_copies_f32(const float* image, float* copies, const float* rot_tf, const float* trans_tf, int n,
                           int h, int w, int c, asr_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Super-resolution solver (superresolution_scripts/superresolution.py, optimizer.py)
 * x [batch,H,W]; y / resid [batch,n,h,w]; transforms [batch,n,8]; H = f*h, W = f*w, f even.
 * ------------------------------------------------------------------------------------------ */

/* x = tf.image.resize(y[:, 0], (H, W)) -- superresolution.py:112-113. */
int asr_sr_init_target_f32(const float* y, float* x, int batch, int n, int H, int W, int h, int w,
                           asr_stream_t stream);

/* resid[b,i] = resize(translate(rotate(tile(x_b), rot_tf), trans_tf), (h,w))[i] - y[b,i]
 * -- the forward model of loss_function, superresolution.py:58-72. */
int asr_sr_forward_residual_f32(const float* x, const float* y, const float* rot_tf, const float* trans_tf,
                                float* resid, int batch, int n, int H, int W, int h, int w, asr_stream_t stream);

/* One optimiser step: gradient of the loss (TensorFlow's registered gradients: SquaredDifference,
 * ResizeBilinearGrad, ImageProjectiveTransformV3 gradient = inverse warp with inv_*_tf, Tile
 * sum; TV / L2 / L1 priors, superresolution.py:71-98,133) followed by the Keras Adam / AMSGrad
 * update (optimizer.py:37-41, superresolution.py:134-135).  alphas [batch] =
 * lr_t * sqrt(1 - beta2^t) / (1 - beta1^t) per image.  x_new must not alias x.  grad_out
 * (optional) receives the raw gradient; with x_new == NULL only grad_out is produced. */
int asr_sr_backward_adam_f32(const float* x, float* x_new, const float* resid, const float* inv_rot_tf,
                             const float* inv_trans_tf, float* m, float* v, float* vhat, const float* alphas,
                             float* grad_out, int batch, int n, int H, int W, int h, int w, float lambda_df,
                             float lambda_tv, float lambda_l2, float lambda_l1, float one_minus_beta1,
                             float one_minus_beta2, float epsilon, int amsgrad, asr_stream_t stream);

/* terms[b] = {sum resid^2, TV(x), sum x^2, sum |x|} in float64 -- the pieces of the scalar loss
 * of superresolution.py:71-98 (reporting only). */
int asr_sr_loss_terms_f64(const float* x, const float* resid, double* terms, int batch, int n, int H, int W, int h,
                          int w, asr_stream_t stream);

/* Bytes of caller-owned workspace asr_sr_solve_f32 needs (residuals, the ping-pong x, the running data-term sum and the
 * zero-bordered gradient planes of one chunk of copies; library default chunking, see asr_sr_config.plane_chunk). */
size_t asr_sr_solve_workspace_bytes(int batch, int n, int H, int W, int h, int w);

/* The whole loop of augmented_superresolution (superresolution.py:120-135): num_iter x
 * {forward residual, backward + Adam}.  x holds the initial target on entry and the result on
 * exit; m / v / vhat [batch,H,W] are the optimiser slots (zero for a fresh variable); alphas
 * [num_iter, batch] (device); last_loss_terms [batch,4] float64 (optional) receives the loss
 * pieces of the last iteration, evaluated before its update like the reference's `loss`. */
int asr_sr_solve_f32(float* x, const float* y, const float* rot_tf, const float* trans_tf, const float* inv_rot_tf,
                     const float* inv_trans_tf, float* m, float* v, float* vhat, const float* alphas, int num_iter,
                     double* last_loss_terms, void* workspace, size_t workspace_bytes, int batch, int n, int H, int W,
                     int h, int w, float lambda_df, float lambda_tv, float lambda_l2, float lambda_l1,
                     float one_minus_beta1, float one_minus_beta2, float epsilon, int amsgrad, asr_stream_t stream);

/* --- the other optimisers and priors of the reference's sweeps -----------------------------------------------
 * optimizer.py:21-41 lets sweep_all.yaml pick Adadelta / Adagrad / Adamax / SGD besides Adam, and
 * superresolution.py:8-23,81-82 swaps the TV prior for bilateral TV (use_BTV).  asr_sr_config selects them for
 * the *_cfg entry points below; the *_adam_f32 / asr_sr_solve_f32 / asr_sr_loss_terms_f64 functions above are the
 * {ASR_OPT_ADAM, ASR_PRIOR_TV} case.  Update rules are the dense CPU kernels of tensorflow 2.7
 * (core/kernels/training_ops.cc) that the Keras optimisers dispatch to; `alphas` carries the per-step scalar:
 *
 *   optimizer          slots used (zero-initialised unless noted)      c0          c1          c2        alphas[it,b]
 *   ASR_OPT_ADAM       m, v, vhat (flag = amsgrad)                     1 - beta1   1 - beta2   epsilon   lr_t sqrt(1-beta2^t)/(1-beta1^t)
 *   ASR_OPT_SGD        m = momentum accumulator (flag = nesterov)      momentum    -           -         lr_t
 *   ASR_OPT_ADAGRAD    v = accumulator (init initial_accumulator_value) -          -           epsilon   lr_t
 *   ASR_OPT_ADADELTA   v = accum, m = accum_update                     rho         1 - rho     epsilon   lr_t
 *   ASR_OPT_ADAMAX     m, v                                            1 - beta1   beta2       epsilon   lr_t / (1 - beta1^t)
 */
#define ASR_OPT_ADAM 0
#define ASR_OPT_SGD 1
#define ASR_OPT_ADAGRAD 2
#define ASR_OPT_ADADELTA 3
#define ASR_OPT_ADAMAX 4
#define ASR_PRIOR_TV 0
#define ASR_PRIOR_BTV 1

typedef struct asr_sr_config {
    int optimizer;     /* ASR_OPT_* */
    int flag;          /* Adam: amsgrad; SGD: nesterov */
    float c0, c1, c2;  /* see the table above */
    int prior;         /* ASR_PRIOR_TV: tf.image.image_gradients TV; ASR_PRIOR_BTV: bilateral_tv */
    float btv_alpha;   /* bilateral_tv(alpha=0.6, ...) */
    int btv_shift;     /* bilateral_tv(shift_factor=2): pairs (h, v), h in [-s, s], v in [0, s]; 1 <= s <= 4 */
    int plane_chunk;   /* asr_sr_solve_*: copies whose per-copy gradient planes are alive at once (a K_gt + K_bwd launch pair
                        * per chunk, the data-term sum carried across in copy order: results do not depend on it).
                        * 0 = library default (all copies at once while the planes stay under 1 GiB per call -- the fastest
                        * form on every measured shape --, an even split beyond); >= n = all copies at once. */
} asr_sr_config;

/* asr_sr_backward_adam_f32 with the update rule and prior of `cfg` (host pointer, read during the call). */
int asr_sr_backward_cfg_f32(const float* x, float* x_new, const float* resid, const float* inv_rot_tf,
                            const float* inv_trans_tf, float* m, float* v, float* vhat, const float* alphas,
                            float* grad_out, int batch, int n, int H, int W, int h, int w, float lambda_df,
                            float lambda_tv, float lambda_l2, float lambda_l1, const asr_sr_config* cfg,
                            asr_stream_t stream);

/* asr_sr_loss_terms_f64 with terms[b][1] = the prior selected by cfg (TV or bilateral TV). */
int asr_sr_loss_terms_cfg_f64(const float* x, const float* resid, double* terms, int batch, int n, int H, int W,
                              int h, int w, const asr_sr_config* cfg, asr_stream_t stream);

/* asr_sr_solve_workspace_bytes for asr_sr_solve_cfg_f32 with this cfg (its plane_chunk decides the size). */
size_t asr_sr_solve_workspace_bytes_cfg(int batch, int n, int H, int W, int h, int w, const asr_sr_config* cfg);

/* asr_sr_solve_f32 with the update rule and prior of `cfg`. */
int asr_sr_solve_cfg_f32(float* x, const float* y, const float* rot_tf, const float* trans_tf, const float* inv_rot_tf,
                         const float* inv_trans_tf, float* m, float* v, float* vhat, const float* alphas, int num_iter,
                         double* last_loss_terms, void* workspace, size_t workspace_bytes, int batch, int n, int H, int W,
                         int h, int w, float lambda_df, float lambda_tv, float lambda_l2, float lambda_l1,
                         const asr_sr_config* cfg, asr_stream_t stream);

/* out[b] = max / mean over copies of rotate(translate(resize(y[b,i], (H,W)), trans_tf), rot_tf)
 * -- max_superresolution / mean_superresolution, superresolution.py:139-161 (trans_tf built
 * from -shifts, rot_tf from -angles).  out [batch,H,W]. */
int asr_realign_max_f32(const float* y, float* out, const float* trans_tf, const float* rot_tf, int batch, int n,
                        int H, int W, int h, int w, asr_stream_t stream);
int asr_realign_mean_f32(const float* y, float* out, const float* trans_tf, const float* rot_tf, int batch, int n,
                         int H, int W, int h, int w, asr_stream_t stream);

/* Both of the above in one pass over the copies (the reference computes max- and mean-SR of the same
 * copies, SR_single_class.py:103-110): out_max / out_mean [batch,H,W], bit-identical to the two calls. */
int asr_realign_max_mean_f32(const float* y, float* out_max, float* out_mean, const float* trans_tf,
                             const float* rot_tf, int batch, int n, int H, int W, int h, int w, asr_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Output processing, thresholding, IoU
 * ------------------------------------------------------------------------------------------ */

/* out_minmax[s] = {min, max} of segment s (tf.reduce_min / tf.reduce_max,
 * augmentation_utils.py:100-101, superres_utils.py:134). */
int asr_minmax_f32(const float* x, float* out_minmax, int64_t per_segment, int segments, asr_stream_t stream);

/* Activation(last_activation) over the class axis (model.py:124-125): kind 1 = softmax, 2 = sigmoid.
 * out may alias logits. */
int asr_class_activation_f32(const float* logits, float* out, int64_t pixels, int classes, int kind,
                             asr_stream_t stream);

/* create_mask: argmax over the class axis, first maximum wins (utils.py:115-119). */
int asr_argmax_i32(const float* logits, int32_t* out, int64_t pixels, int classes, asr_stream_t stream);

/* argmax OPM: class_id where argmax == class_id else 0, as float32 (augmentation_utils.py:106-113). */
int asr_opm_argmax_f32(const float* logits, float* class_mask, int64_t pixels, int classes, int class_id,
                       asr_stream_t stream);

/* slice_max OPM: class logit and max over the other classes (augmentation_utils.py:82-93). */
int asr_opm_slice_max_f32(const float* logits, float* class_mask, float* max_mask, int64_t pixels, int classes,
                          int class_id, asr_stream_t stream);

/* slice OPM: class logit min-max normalised with the copy's global min / max over all classes
 * (augmentation_utils.py:95-104 + superres_utils.py:56-62).  minmax_ws: [copies,2] floats. */
int asr_opm_slice_f32(const float* logits, float* class_mask, float* minmax_ws, int copies, int64_t pixels_per_copy,
                      int classes, int class_id, float new_min, float new_max, asr_stream_t stream);

/* threshold_image (superres_utils.py:118-139): out = image >= th_mask ? th_value : 0, or
 * (th_mask == NULL) image > th_factor * max(image) ? th_value : 0 per segment.
 * minmax_ws: [segments,2] floats (needed when th_mask == NULL). */
int asr_threshold_f32(const float* image, const float* th_mask, float* minmax_ws, int32_t* out, int64_t per_segment,
                      int segments, float th_factor, int th_value, asr_stream_t stream);

/* single_class_IOU integer counts (utils.py:180-204): counts[s] = {inter_c, union_c, inter_bg,
 * union_bg}; with include_bg the truth is first remapped (truth != class_id -> 0).  Void (255)
 * pixels are not excluded, exactly like the reference. */
int asr_iou_counts_i32(const int32_t* truth, const int32_t* pred, int64_t* counts, int64_t per_segment, int segments,
                       int class_id, int include_bg, asr_stream_t stream);

/* The same counts for num_preds prediction masks [num_preds, pixels] against ONE label map [pixels] -- the four masks of an
 * image (standard, ASR, max-SR, mean-SR) against its ground truth, SR_single_class.py:109-120 -- without replicating it. */
int asr_iou_counts_shared_truth_i32(const int32_t* truth, const int32_t* preds, int64_t* counts, int64_t pixels, int num_preds,
                                    int class_id, int include_bg, asr_stream_t stream);

/* min_max_normalization of whole stacks with their own global extrema, as load_SR_data applies it to the argmax / slice_max
 * masks of an image (superres_utils.py:56-62, 183-206): per segment out = new_min + ((x - min) * (new_max - new_min)) /
 * (max - min, or 1 when they are equal).  minmax_ws: [segments, 2] floats of scratch (receives the extrema). */
int asr_minmax_normalize_f32(const float* x, float* out, float* minmax_ws, int64_t per_segment, int segments, float new_min,
                             float new_max, asr_stream_t stream);

/* Standard-output mask of ONE image (generate_standard_output.py:52-65: the model built with final_upsample=True,
 * model.py:108-111, then create_mask and the class filter): bilinear half-pixel upsample of the logits [h_in, w_in, classes]
 * to h_out x w_out, argmax over the classes (first maximum), mask = class_id where it wins, else 0 -- in one pass. */
int asr_standard_mask_i32(const float* logits, int32_t* mask, int h_in, int w_in, int classes, int h_out, int w_out, int class_id,
                          asr_stream_t stream);

/* Per-label pixel counts for the multi-class Mean_IOU (utils.py:151-177, compute_IoU(class_id=None)):
 * counts[seg][0][l] = |truth == l|, counts[seg][1][l] = |pred == l|, counts[seg][2][l] = |truth == l and pred == l|,
 * l = 0..255 (int64, zeroed by the call); IoU_l = c2 / (c0 + c1 - c2). */
int asr_class_counts_i32(const int32_t* truth, const int32_t* pred, int64_t* counts, int64_t per_segment, int segments,
                         asr_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * DeepLabV3+ (Xception-65, OS16) layers -- model.py.  BatchNorm is folded by the caller.
 * ------------------------------------------------------------------------------------------ */

/* Packed-weight size (floats) and packing for the MFMA GEMM: w_kn [k,n] row-major (a Keras
 * HWIO kernel reshaped to [kh*kw*cin, cout]) -> private layout [ceil32(k)/4][ceil128(n)][4]. */
size_t asr_pwconv_packed_floats(int k, int n);
int asr_pwconv_pack_weights_f32(const float* w_kn, float* w_packed, int k, int n, asr_stream_t stream);

/* Conv2D 1x1 (+ folded BN bias, optional ReLU, optional residual Add) on FP32 MFMA:
 * y[r, :n] = act(x[row(r), :k] @ W + bias) + residual[r, :n].  Replaces every pointwise
 * Conv2D + BatchNormalization (+ ReLU / Add) group of model.py (:195-231, :244-247, :303-304,
 * :403-417, :500-506).  sub_stride > 1: rows are gathered at (b, s*oy, s*ox) of an
 * h_in x w_in map -- the stride-2 1x1 shortcut of _conv2d_same (model.py:529-541).
 * relu: 0 = linear, 1 = ReLU, 2 = ReLU6 (the expand convs of _inverted_res_block, model.py:434-440; the same
 * encoding is used by every `relu` / `post_relu` argument below).  k % 4 == 0, ldx % 4 == 0. */
int asr_pwconv_mfma_f32(const float* x, const float* w_packed, const float* bias, const float* residual, float* y,
                        int64_t m, int k, int n, int ldx, int ldy, int ldres, int relu, int sub_stride, int h_in,
                        int w_in, asr_stream_t stream);

/* Split-f16 variant of asr_pwconv_mfma_f32 (same arguments, n > 64): every f32 operand is split into
 * f16 hi + lo on the way into LDS and x*w is evaluated as hi*hi + hi*lo + lo*hi on
 * v_mfma_f32_32x32x16_f16 with f32 accumulation -- f32-grade results (error ~1e-6 of sum |x||w|) at
 * 3/16 of the FP32-MFMA matrix time.  Weights are packed by asr_pwconv_pack_weights_f16x3
 * (asr_pwconv_packed_floats_f16x3 floats: two half planes [ceil32(k)/8][ceil128(n)][8]). */
size_t asr_pwconv_packed_floats_f16x3(int k, int n);
int asr_pwconv_pack_weights_f16x3(const float* w_kn, float* w_packed, int k, int n, asr_stream_t stream);
int asr_pwconv_mfma_f16x3(const float* x, const float* w_packed, const float* bias, const float* residual, float* y,
                          int64_t m, int k, int n, int ldx, int ldy, int ldres, int relu, int sub_stride, int h_in,
                          int w_in, asr_stream_t stream);

/* Conv2D 3x3 as implicit GEMM on FP32 MFMA (cin % 32 == 0): entry_flow_conv1_2, model.py:155-159. */
int asr_conv3x3_mfma_f32(const float* x, const float* w_packed, const float* bias, float* y, int batch, int h_in,
                         int w_in, int cin, int cout, int stride, int pad, int dil, int h_out, int w_out, int ldx,
                         int ldy, int relu, asr_stream_t stream);

/* The same 3x3 implicit GEMM on the split-f16 matrix path (asr_pwconv_mfma_f16x3): w_packed from
 * asr_pwconv_pack_weights_f16x3 on the [9 * cin, cout] matrix; cin % 32 == 0. */
int asr_conv3x3_mfma_f16x3(const float* x, const float* w_packed, const float* bias, float* y, int batch, int h_in,
                         int w_in, int cin, int cout, int stride, int pad, int dil, int h_out, int w_out, int ldx,
                         int ldy, int relu, asr_stream_t stream);

/* --- split-f16 activations between the two halves of a separable conv ------------------------------------------
 * The depthwise output of _SepConv_BN (model.py:478-495) is consumed only by its pointwise conv.  Written directly in
 * the A-operand format of the split-f16 GEMM -- per pixel and per chunk of 32 channels one 128-byte line
 * [hi(32 halfs) | lo(32 halfs)], hi = f16(v), lo = f16(v - hi): the very split asr_pwconv_mfma_f16x3 applies to the
 * f32 value on its way into LDS, and the same number of bytes -- it lets the GEMM take both operands by LDS-DMA
 * (no staging registers, no conversion work) on a 256 x 256 tile.  Same hi / lo halves, same three products; the 256 x 256
 * kernel sums each 32-deep K-step in one v_mfma_f32_16x16x32_f16, so it differs from asr_dwconv3x3_nhwc_f32 followed by
 * asr_pwconv_mfma_f16x3 by f32 summation-order noise only (both within 4e-6 * sum |x||w| of the exact product).
 *
 * asr_dwconv3x3_nhwc_split_f16: asr_dwconv3x3_nhwc_f32 with y in that format; ldy_chunks = ceil(c / 32) chunks per
 * pixel, channels c .. 32 * ldy_chunks - 1 are written as zeros; needs (stride 1, rate 1|2) or (stride 2, rate 1) and
 * h_out a multiple of 16 (32 above 32 rows); y_split 128-byte aligned.
 * asr_pwconv_mfma_f16x3_presplit: asr_pwconv_mfma_f16x3 on such an operand (m rows = pixels, ldx_chunks chunks per
 * row); ceil128(n) must be a multiple of 256; no sub_stride. */
int asr_dwconv3x3_nhwc_split_f16(const float* x, const float* w, const float* bias, void* y_split, int batch, int h_in,
                                 int w_in, int c, int stride, int rate, int pad_top, int pad_left, int h_out, int w_out,
                                 int ldx, int ldy_chunks, int pre_relu, int post_relu, asr_stream_t stream);
int asr_pwconv_mfma_f16x3_presplit(const void* x_split, const float* w_packed, const float* bias, const float* residual,
                                   float* y, int64_t m, int k, int n, int ldx_chunks, int ldy, int ldres, int relu,
                                   asr_stream_t stream);

/* 1 when asr_aspp_dwconv3_nhwc_{f32,split_f16} can run an h x w plane at these three rates (a residue class of the plane
 * modulo gcd(rates) must fit the CU's LDS in one of the kernel's column groupings), else 0: the caller then runs the three
 * branches (model.py:214-221) as separate asr_dwconv3x3_nhwc_f32 launches.  Host arithmetic only -- no device call, no
 * stream; the same function the launchers use, so a plan built on its answer never meets ASR_ERR_UNSUPPORTED for geometry. */
int asr_aspp_dwconv3_supported(int h, int w, int rate0, int rate1, int rate2);

/* asr_aspp_dwconv3_nhwc_f32 with its three outputs as split-f16 operands (ldy_chunks = c / 32; c % 32 == 0). */
int asr_aspp_dwconv3_nhwc_split_f16(const float* x, const float* w3, const float* bias3, void* y0, void* y1, void* y2,
                                    int batch, int h, int w, int c, int rate0, int rate1, int rate2, int ldx,
                                    int ldy_chunks, int pre_relu, int post_relu, asr_stream_t stream);

/* Conv2D 3x3 for tiny cin, weights HWIO [3,3,cin,cout]: entry_flow_conv1_1, model.py:150-153
 * ('same' with stride 2 on an even input pads bottom/right only: pad_top = pad_left = 0). */
int asr_conv3x3_direct_f32(const float* x, const float* w, const float* bias, float* y, int batch, int h_in, int w_in,
                           int cin, int cout, int stride, int pad_top, int pad_left, int h_out, int w_out, int ldx,
                           int ldy, int relu, asr_stream_t stream);

/* The same layer (cin = 3, cout = 32 only) as an implicit GEMM on split-f16 MFMA (K = 27 padded to 32; hi*hi + hi*lo +
 * lo*hi with f32 accumulation: f32-grade results, like asr_pwconv_mfma_f16x3).  Same arguments. */
int asr_conv3x3_stem_f16x3(const float* x, const float* w, const float* bias, float* y, int batch, int h_in, int w_in,
                           int cin, int cout, int stride, int pad_top, int pad_left, int h_out, int w_out, int ldx,
                           int ldy, int relu, asr_stream_t stream);

/* entry_flow_conv1_1 + BN + ReLU + entry_flow_conv1_2 + BN + ReLU (model.py:150-155) in one kernel on split-f16 MFMA:
 * x [batch,h_in,w_in,3] (even sizes) -> y [batch,h_in/2,w_in/2,64].  w1 [3,3,3,32] HWIO and b1 [32], b2 [64] with the BNs
 * folded; w2_packed = asr_pwconv_pack_weights_f16x3 of the [288, 64] matrix (rows (ky, kx, cin)).  The 32-channel
 * intermediate stays in LDS. */
int asr_entry_stem_f16x3(const float* x, const float* w1, const float* b1, const void* w2_packed, const float* b2, float* y,
                         int batch, int h_in, int w_in, int ldx, int ldy, asr_stream_t stream);

/* A whole _SepConv_BN (model.py:463-508) in one kernel for the high-resolution entry-flow layers: [ReLU ->] depthwise
 * 3x3 (stride 1, rate 1, 'same') + BN [-> ReLU] -> pointwise 1x1 + BN [-> ReLU]; cin in {64, 128}, cout = 128.  The
 * depthwise output goes to LDS as split-f16 lines and is consumed there by the MFMA GEMM (weights resident in LDS);
 * bit-identical to asr_dwconv3x3_nhwc_f32 followed by asr_pwconv_mfma_f16x3.  w_dw [3,3,cin], w_pw_packed from
 * asr_pwconv_pack_weights_f16x3 of the [cin, cout] matrix. */
int asr_sepconv_fused_f16x3(const float* x, const float* w_dw, const float* bias_dw, const void* w_pw_packed,
                            const float* bias_pw, float* y, int batch, int h, int w, int cin, int cout, int ldx, int ldy,
                            int pre_relu, int dw_relu, int out_relu, asr_stream_t stream);

/* DepthwiseConv2D 3x3 (+ ZeroPadding2D, folded BN, ReLU before and/or after): the depthwise half
 * of _SepConv_BN, model.py:478-495, and of _inverted_res_block, model.py:442-449 (post_relu = 2: ReLU6).
 * w [3,3,c] with the BN scale folded, bias [c].
 * mode: 0 = auto, 1 = direct (any stride / rate), 2 = register-window streaming ((stride 1, rate 1|2) or (stride 2, rate 1)). */
int asr_dwconv3x3_nhwc_f32(const float* x, const float* w, const float* bias, float* y, int batch, int h_in,
                           int w_in, int c, int stride, int rate, int pad_top, int pad_left, int h_out, int w_out,
                           int ldx, int ldy, int pre_relu, int post_relu, int mode, asr_stream_t stream);

/* The three dilated depthwise convs of the ASPP (aspp1/2/3_depthwise + BN + ReLU, model.py:212-221)
 * fused: each residue class of the plane modulo g = gcd(rates) -- on which the dilated taps close -- is staged in LDS
 * once and read by all three rates (input read from HBM once, any plane size).  w3 [3,3,3,c] (branch major),
 * bias3 [3,c]; stride 1, 'same' padding; ceil(h/g) * ceil(w/g) * 128 bytes must fit the 160 KB LDS (ASR_ERR_UNSUPPORTED
 * otherwise: run asr_dwconv3x3_nhwc_f32 per branch). */
int asr_aspp_dwconv3_nhwc_f32(const float* x, const float* w3, const float* bias3, float* y0, float* y1, float* y2,
                              int batch, int h, int w, int c, int rate0, int rate1, int rate2, int ldx, int ldy,
                              int pre_relu, int post_relu, asr_stream_t stream);

/* GlobalAveragePooling2D(keepdims=True): y[b, :c] = mean over hw pixels (model.py:196-197). */
int asr_gap_f32(const float* x, float* y, int batch, int hw, int c, int ldx, asr_stream_t stream);

/* Resizing(bilinear) / tf.image.resize, half-pixel centres (model.py:109-110, 204-205, 241-242). */
int asr_resize_bilinear_f32(const float* x, float* y, int batch, int h_in, int w_in, int c, int h_out, int w_out,
                            int ldx, int ldy, asr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* ASR_HIP_H */
