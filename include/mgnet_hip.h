/*
 * mgnet_hip.h -- C-ABI of libmgnet_hip.so: the MI355X (gfx950) kernels of MGNet's training hot path.
 *
 * The reference (uulm-mrm/MGNet) is pure Python on torch ops and has no FFI of its own; each entry
 * point below replaces the chain of torch ops behind ONE reference function (cited per symbol), and
 * is what a maintainer would bind with ctypes from that function.
 *
 * Conventions (all symbols):
 *   - plain pointers and sizes only; no torch / C++ types; `stream` is a hipStream_t passed as void*
 *   - every data pointer is a DEVICE pointer to a contiguous NCHW fp32 tensor unless stated otherwise
 *   - returns 0 on success, a negative MGN_E* code on error; never throws, never allocates, never syncs
 *   - stream-ordered, re-entrant; scratch space is caller-provided (`*_workspace_bytes` query per op)
 */
#ifndef MGNET_HIP_H
#define MGNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MGN_OK 0
#define MGN_EINVAL (-22)     /* bad shape / null pointer / unsupported option value            */
#define MGN_ENOSPC (-28)     /* workspace too small                                            */
#define MGN_ENOTSUP (-95)    /* option exists in the reference but has no kernel yet           */
#define MGN_ELAUNCH (-5)     /* hipLaunchKernel reported an error                              */

#define MGN_MAX_SCALES 4

/* mgn_reproj_cfg.frame_layout */
#define MGN_FRAMES_PLANAR_F32 0
#define MGN_FRAMES_CTX_RGBX_F32 1
#define MGN_FRAMES_RGBX_U8 2

/* library identification: "mgnet_hip <version> gfx950" */
const char* mgn_version(void);

/* ------------------------------------------------------------------------------------------------
 * Self-supervised photometric reprojection loss
 *   replaces mgnet/modeling/loss.py:111-154  MultiViewPhotometricLoss.forward  (+ its autograd backward)
 *   and everything it calls in mgnet/geometry (camera.py:107-182, camera_utils.py:24-55, pose.py:40-95,
 *   pose_utils.py:9-59, depth.py:11-51, image.py:42-69) and loss.py:156-294.
 *
 * Configuration supported by the kernels: the reference defaults (mgnet/config.py:109-117: automask_loss=True,
 *   photometric_reduce_op="min", padding_mode="zeros", ssim_loss_weight>0) and automask_loss=False with "min" or "mean".
 *   padding_mode "zeros" (default), "border", "reflection".  ssim_loss_weight = 0 (loss.py:196-197: the photometric maps are the
 *   3-channel L1 maps): "min" runs over channels and sources and needs a mask, "mean" is the ordinary formula and must not have one --
 *   the other two combinations are MGN_EINVAL (the reference's boolean indexing raises IndexError); automask_loss=True with "mean" is MGN_EINVAL (the reference asserts, loss.py:105-109).
 *
 * Inputs
 *   inv_depth[n_scales] : [B,1,H,W] fp32 each (all scales already at full resolution, mg_net.py:804-807)
 *   img, prev, next     : [B,3,H,W] fp32 in [0,1]   (image_orig, image_prev_orig, image_next_orig); other layouts: cfg.frame_layout
 *   mask                : [B,1,H,W] uint8 (torch.bool storage) or NULL = all ones (loss.py:236-237)
 *   cam                 : camera matrices, fp32; `cam_stride` floats between images and `cam_ld` floats between
 *                         rows, so that camera_matrix[B,4,4] (cam_stride=16, cam_ld=4) is consumed in place
 *                         (loss.py:122 takes [:, :3, :3])
 *   pose                : [B,2,6] fp32 (tx,ty,tz,rx,ry,rz) for prev and next (loss.py:117-119)
 * Outputs
 *   losses              : device fp32[2] = { photometric_loss_weight * L_p , smoothing_loss_weight * L_s }
 *   d_pose              : device fp32[B,2,6] = d losses[0] / d pose              (only if want_grad)
 *   g_inv[n_scales]     : [B,1,H,W] fp32, UNSCALED photometric gradient wrt inv_depth (only if want_grad);
 *                         mgn_reproj_loss_bwd turns it into the final gradient in place
 *   dbg_minmap          : optional [n_scales][B,1,H,W] per-pixel min photometric map (tests), or NULL
 *
 * The forward launch computes the loss AND (want_grad!=0) the pixel-wise photometric gradient in the same
 * pass over the inputs; mgn_reproj_loss_bwd is a light streaming kernel that applies the upstream gradients
 * and adds the smoothness term.  Algorithmic HBM traffic: fwd 49 B/px read + 12 B/px written,
 * bwd 37 B/px read + 12 B/px written (+12 B/px read of g_inv) = 110 B/px per training step.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int B, H, W;
    int n_scales;             /* 1..MGN_MAX_SCALES */
    float ssim_loss_weight;   /* 0.85 */
    float photometric_loss_weight; /* 1.0 */
    float smoothing_loss_weight;   /* 0.001 */
    int automask_loss;        /* 1 (reference default): the un-warped context frames compete in the per-pixel min (loss.py:139-144); 0: off */
    int photometric_reduce_op;/* 0 = "min", 1 = "mean" (loss.py:242-246; "mean" only with automask_loss = 0, loss.py:105-109) */
    int padding_mode;         /* F.grid_sample padding of the warp (camera_utils.py:24-55): 0 = "zeros", 1 = "border", 2 = "reflection" */
    int rows_per_wave;        /* 0 = choose automatically; else rows each wavefront owns (>=4) */
    int frame_layout;         /* MGN_FRAMES_*: how img / prev / next are laid out (same results for all three):
                                 0 PLANAR_F32   fp32 [B,3,H,W] planes, the reference's tensors (mg_net.py:320-335 `uint8.float() / 255`)
                                 1 CTX_RGBX_F32 prev / next fp32 [B,H,W,4] RGBx (a 4-channel channels_last tensor, 4th channel
                                                ignored; mgn_u8_frames_to_f32_nhwc4 produces it), img planar: one 16-byte gather
                                                per bilinear corner instead of three 4-byte ones
                                 2 RGBX_U8      img, prev, next uint8 [B,H,W,4] RGBX (mgn_u8_frames_to_rgbx produces it from the
                                                [3,H,W] uint8 frames the step receives): one 4-byte gather per corner, 4 B/px per
                                                frame; the kernels convert with the exactly rounded byte / 255 */
    void* prof_begin;         /* optional hipEvent_t recorded on `stream` right before the dominant kernel */
    void* prof_end;           /* optional hipEvent_t recorded right after it (bench.py's roofline leg); NULL = off */
} mgn_reproj_cfg;

int mgn_reproj_workspace_bytes(const mgn_reproj_cfg* cfg, size_t* bytes);

int mgn_reproj_loss_fwd(const mgn_reproj_cfg* cfg,
                        const float* const* inv_depth, const void* img, const void* prev, const void* next,
                        const uint8_t* mask, const float* cam, int cam_stride, int cam_ld, const float* pose,
                        int want_grad, float* losses, float* d_pose, float* const* g_inv, float* dbg_minmap,
                        void* workspace, size_t workspace_bytes, void* stream);

/* grad_losses: device fp32[2] = upstream gradients of {loss_photometric, loss_smoothness}.
 * g_inv[i] (written by the forward with want_grad=1 and the SAME workspace) is overwritten with
 * d(grad_losses . losses)/d inv_depth[i].  d_pose_out[B,2,6] = grad_losses[0] * d_pose. */
int mgn_reproj_loss_bwd(const mgn_reproj_cfg* cfg,
                        const float* const* inv_depth, const void* img, const uint8_t* mask,
                        const float* grad_losses, const float* d_pose, float* const* g_inv, float* d_pose_out,
                        const void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * In-place activated batch norm with cross-rank statistics
 *   replaces inplace_abn.InPlaceABNSync (68 call sites: mgnet/modeling/res_net.py:35,49,59,103 and
 *   layers.py:63,71,117,209,242,253,291; always momentum=0.01, group=WORLD, activation leaky_relu(0.01)|identity).
 *   y = act((|weight|+eps) * (x-mean)/sqrt(var+eps) + bias); backward recovers x_hat from the OUTPUT y.
 *
 * x/y/dy/dx: [M = N*H*W, C] channels-last activations, dtype 0 = fp32, 1 = bf16; C % (16/sizeof) == 0, C <= 1024.
 * The cross-rank exchange is the caller's (RCCL via torch.distributed): all_gather of `stats` between
 * mgn_iabn_stats and mgn_iabn_combine (forward), all_reduce of `sums` between mgn_iabn_bwd_reduce and
 * mgn_iabn_bwd_apply (backward) -- the same two exchanges inplace_abn performs.
 *   forward : stats -> [all_gather] -> combine (Chan's formula; running stats; scale/offset; saved={mean,rstd}) -> apply
 *   backward: bwd_reduce (sums[2][C] = {sum dz, sum dz*x_hat}; d_bias = sums[0], d_weight = sign(weight)*sums[1])
 *             -> [all_reduce sums] -> bwd_apply
 * activation: 0 = identity, 1 = leaky_relu(slope).
 * ---------------------------------------------------------------------------------------------- */
int mgn_iabn_workspace_bytes(long M, int C, int dtype, size_t* bytes);
int mgn_iabn_stats(const void* x, int dtype, long M, int C, float* stats /*[3][C]: count, mean, M2*/,
                   void* workspace, size_t workspace_bytes, void* stream);
/* single-rank training forward: statistics AND the coefficients of one process in ONE launch (= stats + combine with
 * n_ranks = 1; coef[4][C] = scale, offset, mean, rstd; running statistics updated when non-null) */
/* the same outputs as mgn_iabn_stats (stats[3][C], may be NULL) and / or mgn_iabn_train_coeffs (coef[4][C] + running statistics,
 * may be NULL) from the per-tile partial sums a convolution left behind (mgn_conv3x3_win: partials[rows][C][2] of (r - shift),
 * (r - shift)^2 over its rounded outputs; shift as given to the convolution): the statistics pass over the activation becomes a
 * read of rows*C*8 bytes.  M = N*H*W of the activation; C % 4 == 0. */
/* optional first stage for very many partial rows (the stems' 32768 pixel tiles): out [rows_out][C][2] = the rows k, k + rows_out, ...
 * added in fp64; mgn_iabn_coeffs_from_partials then runs on `out` (rows_out <= 1024). */
int mgn_iabn_partials_reduce(const float* partials, int rows, int C, int rows_out, float* out, void* stream);
int mgn_iabn_coeffs_from_partials(const float* partials, int rows, int C, long M, const float* shift, const float* weight,
                                  const float* bias, float eps, float momentum, float* running_mean, float* running_var,
                                  float* coef /*[4][C] or NULL*/, float* stats /*[3][C] or NULL*/, void* stream);
int mgn_iabn_train_coeffs(const void* x, int dtype, long M, int C, const float* weight, const float* bias, float eps,
                          float momentum, float* running_mean, float* running_var, float* coef /*[4][C]*/,
                          void* workspace, size_t workspace_bytes, void* stream);
int mgn_iabn_combine(const float* gathered /*[n_ranks][3][C]*/, int n_ranks, int C, const float* weight, const float* bias,
                     float eps, float momentum, float* running_mean /*nullable*/, float* running_var,
                     float* scale, float* offset, float* saved /*[2][C]: mean, rstd*/, void* stream);
int mgn_iabn_eval_coeffs(int C, const float* weight, const float* bias, const float* running_mean,
                         const float* running_var, float eps, float* scale, float* offset, void* stream);
int mgn_iabn_apply(const void* x, void* y /*may alias x*/, int dtype, long M, int C, const float* scale,
                   const float* offset, int activation, float slope, void* stream);
int mgn_iabn_bwd_reduce(const void* y, const void* dy, int dtype, long M, int C, const float* weight, const float* bias,
                        float eps, int activation, float slope, float* sums /*[2][C]*/,
                        float* dwb /*nullable [2][C]: d_weight, d_bias of this rank*/,
                        void* workspace, size_t workspace_bytes, void* stream);
int mgn_iabn_bwd_apply(const void* y, const void* dy, void* dx /*may alias dy*/, int dtype, long M, int C,
                       const float* weight, const float* bias, const float* saved, const float* sums,
                       float total_count, float eps, int activation, float slope, void* stream);

/* "from x" forms of the two backward passes: `x` is the norm's INPUT (the producing conv's output, kept instead of the
 * normalised map) and z = scale * x + offset is recomputed (scale/offset: rows 0, 1 of the coefficient block).  Used by
 * the fused norm + add + ReLU tail of the residual blocks (mgn_abn_add_relu_fwd). */
int mgn_iabn_bwd_reduce_x(const void* x, const void* dy, int dtype, long M, int C, const float* weight, const float* bias,
                          const float* scale, const float* offset, float eps, int activation, float slope, float* sums, float* dwb,
                          void* ws, size_t ws_bytes, void* stream);
/* block tail relu(norm_identity(x) + shortcut) (res_net.py:62-79): mgn_iabn_bwd_reduce_x with the ReLU mask folded in.  g = gradient of the
 * tail's output, yrelu = that output (16-bit); writes dm = g * (yrelu > 0) -- the gradient of both summands: the shortcut's gradient and
 * what mgn_iabn_bwd_apply_x reads -- and reduces it in the same pass (replaces mgn_relu_mask_bwd + the re-read of its result).
 * relu_bits (nullable): the mask as one byte per 8 values, bit k = (yrelu[k] > 0), as mgn_abn_add_relu_fwd writes it; when given it is
 * read instead of yrelu (1/16 of the bytes; yrelu may then be NULL).  Same dm bit for bit. */
int mgn_iabn_bwd_reduce_x_relu(const void* x, const void* g, const void* yrelu /*nullable with relu_bits*/, const void* relu_bits /*nullable*/,
                               void* dm, long M, int C, const float* weight, const float* bias, const float* scale, const float* offset,
                               float eps, float* sums, float* dwb, void* ws, size_t ws_bytes, void* stream);
int mgn_iabn_bwd_apply_x(const void* x, const void* dy, void* dx, int dtype, long M, int C, const float* weight, const float* bias,
                         const float* scale, const float* offset, const float* saved, const float* sums, float total_count, float eps,
                         int activation, float slope, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Full-model gradient clipping + Adam over flat fp32 buckets
 *   replaces tools/train_net.py:108-154: torch.optim.Adam wrapped in FullModelGradientClippingOptimizer
 *   (clip_grad_norm_(all params, CLIP_VALUE=0.01, L2) then Adam.step), parameter groups of mgnet/solver/build.py:9-116.
 * Buckets are padded so that every tensor starts at a multiple of mgn_optim_chunk() elements; chunk_lr / chunk_wd hold
 * the learning rate / weight decay of the tensor that owns each chunk (device arrays, n / chunk entries).
 *   mgn_sqnorm     : partials[0..n_partials) = block sums of g^2 (call once per bucket, consecutive partial ranges)
 *   mgn_clip_coef  : coef_and_norm = { min(1, max_norm/(norm+1e-6)), norm } with norm = sqrt(sum partials)*grad_scale
 *   mgn_adam_step  : torch.optim.Adam (no amsgrad) on g*grad_scale*coef, bias correction for `step` (1-based)
 * ---------------------------------------------------------------------------------------------- */
int mgn_optim_chunk(void);
int mgn_sqnorm(const float* g, long n, float* partials, int max_partials, int* n_partials, void* stream);
int mgn_clip_coef(const float* partials, int n_partials, float max_norm, float grad_scale, float* coef_and_norm,
                  void* stream);
int mgn_adam_step(float* p, const float* g, float* m, float* v, long n, const float* chunk_lr, const float* chunk_wd,
                  float beta1, float beta2, float eps, int step, const float* clip_coef, float grad_scale, void* stream);
/* the same update with the bias corrections read from DEVICE memory: hyper = { 1/(1-beta1^step), 1/sqrt(1-beta2^step) }.
 * No per-step host scalar is left in the launch, so it can be captured once in a hipGraph and replayed every step. */
int mgn_adam_step_dev(float* p, const float* g, float* m, float* v, long n, const float* chunk_lr, const float* chunk_wd,
                      float beta1, float beta2, float eps, const float* hyper, const float* clip_coef, float grad_scale,
                      void* stream);
/* The other optimizers tools/train_net.py:129-154 can build, on the same flat buckets / tables / clip coefficient: kind 0 = Adam (above),
 * 1 = AdamW (torch.optim.AdamW: decoupled decay p <- p (1 - lr wd), v = second moment), 2 = SGD (torch.optim.SGD with momentum `beta1`,
 * dampening 0, optional Nesterov; m = momentum buffer, v unused / may be NULL).  hyper as for mgn_adam_step_dev (ignored by SGD). */
int mgn_optim_step_dev(int kind, float* p, const float* g, float* m, float* v, long n, const float* chunk_lr, const float* chunk_wd,
                       float beta1, float beta2, float eps, int nesterov, const float* hyper, const float* clip_coef, float grad_scale,
                       void* stream);
/* Dynamic loss scaling for fp16 activations -- torch.cuda.amp.GradScaler as used by detectron2's AMPTrainer
 * (tools/train_net.py:162), evaluated on the device.  The gradients in the buckets are `S` times the true ones
 * (the caller multiplied the loss by scaler_state[0] before backward).  After mgn_sqnorm:
 *   coef_norm_found = { clip coefficient / S, true gradient norm, found_inf (1 | 0) }   (coef[0] = 0 when found_inf)
 *   scaler_state    = { S, clean steps since the last growth, optimizer steps taken }: found_inf -> S / 2 (>= 1), else
 *                     step count + 1 and S * 2 after `growth_interval` clean steps (GradScaler defaults: 65536, 2000)
 *   hyper           = Adam bias corrections for the (possibly unchanged) step count, consumed by mgn_adam_step_dev,
 *                     which leaves parameters and moments untouched when coef_norm_found[2] != 0 (3 floats needed there). */
int mgn_clip_coef_scaled(const float* partials, int n_partials, float max_norm, float grad_scale, float beta1, float beta2,
                         int growth_interval, float* scaler_state, float* hyper, float* coef_norm_found, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Convolutions as implicit GEMM on the bf16 matrix cores
 *   replaces F.conv2d (cuDNN/MIOpen) behind detectron2.layers.Conv2d / nn.Conv2d in mgnet/modeling/res_net.py:28-60,
 *   96-104 and layers.py:53-72,110-118,146-149,201-210,234-256,283-311.
 * in  : [N, IH, IW, Cin]  bf16 (channels-last);  w : [Cout, KH, KW, Cin] bf16;  out : [N, OH, OW, Cout] bf16 | fp32
 * mgn_conv_igemm: out[n,oh,ow,co] = bias[co] + sum_{kh,kw,ci} in[n, t(oh,kh), t(ow,kw), ci] * w[co,kh,kw,ci], optional ReLU,
 *   t(o,k) = o*stride + k - pad, and when up > 1 the tap only contributes where t is divisible by `up` (then t /= up):
 *   with (w flipped+transposed, stride=1, pad=K-1-pad, up=forward stride) this is the data gradient (computed per
 *   output parity class over the taps that meet a non-zero, not over the zero-upsampled tensor).
 *   Cin must be a multiple of 32 (MGN_ENOTSUP otherwise: the 3/9-channel 7x7 stems).
 * mgn_conv_wgrad: dw[co,kh,kw,ci] = sum_{n,oh,ow} dout[n,oh,ow,co] * in[n, oh*stride+kh-pad, ow*stride+kw-pad, ci]
 *   (pixels are split over blocks; per-split partial tiles go to the workspace with plain stores and a second kernel
 *    sums them in a fixed order -- deterministic, no atomics; Cin, Cout multiples of 8)
 * ---------------------------------------------------------------------------------------------- */
int mgn_conv_igemm(const void* in, const void* w, void* out, const float* bias, int N, int IH, int IW, int Cin, int OH,
                   int OW, int Cout, int KH, int KW, int stride, int pad, int up, int relu, int out_f32,
                   const void* residual /* bf16 [N,OH,OW,Cout] added before rounding (the gradient of a second branch of the
                                           same tensor, e.g. the ResNet shortcut), or NULL */,
                   void* stream);
int mgn_conv_wgrad(const void* dout, const void* in, float* dw, int N, int IH, int IW, int Cin, int OH, int OW, int Cout,
                   int KH, int KW, int stride, int pad, int oihw_cin /* >0: dw is [Cout][oihw_cin][KH][KW] */,
                   void* workspace, size_t workspace_bytes, void* stream);
int mgn_conv_wgrad_workspace_bytes(int N, int OH, int OW, int Cin, int Cout, int KH, int KW, size_t* bytes);
/* Deferred split-K reduction: mgn_conv_wgrad_partial is mgn_conv_wgrad without the final reduction -- the per-split partial tiles
 * stay in `workspace` (the caller keeps it alive) and desc8 (HOST, 8 x int64) receives {partial, 0, splits, Cout, taps, Cin, oihw,
 * cin_real}; MGN_ENOTSUP for the stem shapes (they keep their own reduction).  mgn_conv_wgrad_reduce_batch then performs the
 * reductions of MANY weight gradients in ONE launch from a device table of 10 x int64 per entry: {partial, dst (fp32 gradient in the
 * layout oihw/cin_real select), splits, Cout, taps, Cin, oihw, cin_real, first block, blocks along the (tap, ci) axis =
 * mgn_conv_wgrad_reduce_blocks(splits, taps, Cin)}, entry k owning blocks [first_k, first_k + Cout_k * gy_k).  Same sums in the same
 * fixed order as mgn_conv_wgrad.
 * The gradient reducer (mgnet_amd/engine/reducer.py) batches a bucket's convolutions this way: ~70 launches per step become ~5. */
int mgn_conv_wgrad_partial(const void* dout, const void* in, int N, int IH, int IW, int Cin, int OH, int OW, int Cout, int KH, int KW,
                           int stride, int pad, int oihw_cin, void* workspace, size_t workspace_bytes, long long* desc8, void* stream);
int mgn_conv_wgrad_reduce_batch(const void* table_dev, int n_entries, long total_blocks, void* stream);
int mgn_conv_wgrad_reduce_blocks(int splits, int taps, int Cin);   /* blocks per output channel of one entry (table column 9); < 0: error */
/* The FeatureFusionModule's 1x1 convolution over torch.cat([fsp, fcp], dim=1) (layers.py:316-317) WITHOUT the concatenated map -- the
 * three decoders each copied 2 x 67 MB into it and split its gradient again (mgn_concat2 / mgn_split2, 0.28 ms per step):
 *   mgn_conv1x1_cat   out = conv1x1(in0 | in1, w): in0, in1 [N,H,W,128], w [Cout][256] (layout mode 0), Cout % 256 == 0; the rows of the
 *                     streaming kernel's LDS tile arrive as two half rows from the two maps;
 *   mgn_conv1x1_split (out0 | out1) = conv1x1(in, w): in [N,H,W,256] = the output gradient, w [256][256] (layout mode 1): the data
 *                     gradient, its two channel halves written to maps of their own;
 *   mgn_conv_wgrad_cat weight gradient against (in0 | in1), Cin % 256 == 0: dw != NULL reduces at once (like mgn_conv_wgrad), desc8 !=
 *                     NULL leaves the split partials for mgn_conv_wgrad_reduce_batch (like mgn_conv_wgrad_partial); exactly one of them.
 * Same arithmetic in the same order as the plain kernels on the concatenated map: bit-identical results (tests/test_conv_gpu.py).
 * MGN_ENOTSUP: other channel counts or maps too small for the streaming kernel (the caller concatenates). */
int mgn_conv1x1_cat(const void* in0, const void* in1, const void* w, void* out, int N, int H, int W, int Cin, int Cout, void* stream);
int mgn_conv1x1_split(const void* in, const void* w, void* out0, void* out1, int N, int H, int W, int Cin, int Cout, void* stream);
int mgn_conv_wgrad_cat(const void* dout, const void* in0, const void* in1, float* dw, int N, int H, int W, int Cin, int Cout,
                       void* workspace, size_t workspace_bytes, long long* desc8, void* stream);
/* 3x3 / stride 1 / pad 1 convolution (forward, or data gradient on flipped+transposed weights) with Cin % 32 == 0 and
 * Cout % 128 == 0 as a WINDOWED implicit GEMM (csrc/conv_win.hip): a block owns a patch of patch_rows (8 | 16) x 32 output pixels
 * and keeps the input window in LDS for all nine taps.  Same tensors as mgn_conv_igemm (which dispatches here for the
 * 128/256/512-channel layers of res_net.py:28-60 and layers.py:53-72,110-118,201-210,283-311); MGN_ENOTSUP for other shapes. */
int mgn_conv3x3_win(const void* in, const void* w, void* out, int N, int H, int W, int Cin, int Cout, const void* residual,
                    int patch_rows,
                    float* stat_partials /* NULL, or [N * ceil(H/patch_rows) * ceil(W/32)][Cout][2]: per-patch sums of (r - shift) and
                                            (r - shift)^2 over the ROUNDED outputs r, the input of mgn_iabn_coeffs_from_partials: the
                                            statistics pass of the InPlaceABNSync that follows the conv (res_net.py:35,49,59) */,
                    const float* stat_shift /* [Cout] or NULL (= 0): e.g. the layer's running_mean */, void* stream);
/* the same with the epilogue out = act(conv + bias + residual) (bias [Cout] fp32 or NULL; act 0 none | 1 ReLU | 2 leaky ReLU with
 * `slope`): not together with the statistics rows.  mgn_conv_igemm_act dispatches here. */
int mgn_conv3x3_win_act(const void* in, const void* w, void* out, int N, int H, int W, int Cin, int Cout, const void* residual,
                        int patch_rows, float* stat_partials, const float* stat_shift, const float* bias, int act, float slope, void* stream);
/* out = act(conv(in, w) + bias + residual): mgn_conv_igemm (up = 1, 16-bit output) with the activation as a parameter (0 none | 1 ReLU |
 * 2 leaky ReLU with `slope`) and the bias / activation epilogue in every forward kernel it dispatches to except the 1x1 streaming
 * kernel (those layers take the generic kernel).  Inference (mg_net.py:375-425, the reference's InPlaceABN -> ABN swap for deployment,
 * tools/onnx_trt_export.py:19): `conv -> InPlaceABNSync(eval)` is a fixed affine + activation, so the caller folds the scale into the
 * weights, passes the shift as `bias`, and a residual block's `+ shortcut -> ReLU` as `residual` / act = 1: no norm pass at all. */
int mgn_conv_igemm_act(const void* in, const void* w, void* out, const float* bias, int N, int IH, int IW, int Cin, int OH, int OW,
                       int Cout, int KH, int KW, int stride, int pad, int act, float slope, const void* residual, void* stream);
/* Data gradient of a 3x3 / stride 2 / pad 1 convolution (res_net.py:28-60 with stride 2: conv1 of the down-sampling BasicBlocks) as a
 * windowed implicit GEMM over the LOW-resolution gradient (csrc/conv_up2.hip): in = d(conv output) [N,H,W,Cin], w = the flipped /
 * transposed weights [Cout][3][3][Cin] (mgn_weight_layout mode 1), out = d(conv input) [N,OH,OW,Cout] with OH in {2H-1, 2H}, OW alike;
 * the same sums as mgn_conv_igemm(stride 1, pad 1, up 2), which dispatches here (Cin % 32 == 0, Cout % 64 == 0; MGN_ENOTSUP otherwise).
 * residual: 16-bit [N,OH,OW,Cout] added before rounding (the shortcut's gradient), or NULL. */
int mgn_conv3x3_up2_win(const void* in, const void* w, void* out, int N, int H, int W, int Cin, int Cout, int OH, int OW,
                        int ksize /* 3, or 1: the 1x1 / stride 2 / pad 0 shortcut conv (res_net.py `downsample`), w = [Cout][1][1][Cin]:
                                     only the even output pixels receive a product, the rest is zeros (+ residual) */,
                        const void* residual,
                        int residual_lowres /* 1 (ksize 3): `residual` is [N,H,W,Cout] at the LOW resolution and is added to the even output
                                               pixels only -- the data gradient of the block's 1x1 / stride-2 shortcut conv as the plain
                                               1x1 product it is, without its zero-filled full-resolution form */,
                        void* stream);
/* Convolution + the batch statistics of its output in one launch (forward of conv -> InPlaceABNSync, res_net.py:35,49,59,
 * layers.py:63,71): mgn_conv_stat_rows says how many partial rows the kernel mgn_conv_igemm would pick for this layer leaves behind
 * (0: that kernel has no statistics epilogue -- run mgn_iabn_train_coeffs over the output instead; *shifted = 1: the sums are taken
 * around stat_shift, else around 0); mgn_conv_igemm_stats is mgn_conv_igemm (no bias / ReLU / residual, 16-bit output) that also
 * fills stat_partials [rows][Cout][2] = sums of r, r^2 over the ROUNDED outputs, the input of mgn_iabn_coeffs_from_partials.
 * Kernels with the epilogue: the windowed 3x3 kernel (csrc/conv_win.hip) and the 64-channel row-march kernel (conv3x3_c64). */
int mgn_conv_stat_rows(int N, int IH, int IW, int Cin, int OH, int OW, int Cout, int KH, int KW, int stride, int pad, int* shifted);
int mgn_conv_igemm_stats(const void* in, const void* w, void* out, int N, int IH, int IW, int Cin, int OH, int OW, int Cout, int KH,
                         int KW, int stride, int pad, float* stat_partials, const float* stat_shift, void* stream);
/* ------------------------------------------------------------------------------------------------
 * Launch plan: one training step recorded once and replayed from C
 *   replaces, per step, the host side of the loop tools/train_net.py:232-234 hands to detectron2's trainer
 *   (`trainer.train()` -> run_step: forward, losses, backward, optimizer): between mgn_plan_begin and
 *   mgn_plan_end every kernel launch of this library is executed AND written down (kernel, grid, block, LDS bytes, stream,
 *   arguments by value, each classified as pointer-to-const / pointer / opaque); the host (mgnet_amd/engine/plan.py)
 *   derives the cross-stream dependencies from the memory the launches touch, hands the schedule back with
 *   mgn_plan_compile and replays it with one mgn_plan_run per step segment.  Memory must be static across replays
 *   (the recording runs inside a private allocator pool); values the host changes per step live in device tables.
 *   One recording at a time per process; replay is single-threaded per plan.
 * ------------------------------------------------------------------------------------------------ */
#define MGN_PLAN_OP_LAUNCH 0   /* a = node index (kernel launch or profiling mark)        */
#define MGN_PLAN_OP_RECORD 1   /* a = event index, recorded on `stream`                   */
#define MGN_PLAN_OP_WAIT 2     /* `stream` waits for event a                              */
#define MGN_PLAN_OP_BREAK 3    /* return to the host (collectives, torch ops it replays)  */
typedef struct mgn_plan_node_info_t {
    int type;                 /* 0 kernel launch, 1 profiling mark (hipEvent pair around the dominant kernel) */
    int which;                /* profiling mark: 0 begin, 1 end */
    void* stream;             /* hipStream_t */
    const void* func;         /* host address of the kernel */
    const char* name;         /* kernel name (owned by the runtime) */
    unsigned grid[3], block[3];
    size_t shmem;
    int nargs, nbytes;        /* arguments; size of the by-value argument blob */
    const unsigned char* blob;
} mgn_plan_node_info_t;
int mgn_plan_begin(void);
int mgn_plan_recorded(void);                       /* nodes recorded so far, -1 outside a recording */
const void* mgn_plan_current(void);                /* the plan being recorded (node queries while it grows), or NULL */
int mgn_plan_end(void** plan);
int mgn_plan_abort(void);
int mgn_plan_node_count(const void* plan);
int mgn_plan_node_info(const void* plan, int i, mgn_plan_node_info_t* out);
int mgn_plan_node_args(const void* plan, int i, int max_args, int* offsets, int* sizes, int* kinds /*0 opaque, 1 const pointer, 2 pointer*/);
int mgn_plan_node_ro(const void* plan, int i, int max_args, unsigned long long* ro /*[2 * nargs]: read-only pointer words of struct arguments*/);
int mgn_plan_node_ro_family(const void* plan, int i, int max_args, int* family /*[nargs]: 0 no declaration, 1 convolution-kernel structs (honoured by default), 2 other*/);
int mgn_plan_compile(void* plan, int n_ops, const int* types, const int* a, void* const* streams, int n_events, int prof_slots);
int mgn_plan_run(void* plan, int from_op, int prof_slot);   /* -> index of the BREAK it stopped at, or the op count */
int mgn_plan_set_stream(void* plan, int node, void* stream);   /* replay a node on another stream (the schedule must order it accordingly) */
int mgn_plan_prof_elapsed(void* plan, int slot, float* ms);
/* Measuring the step from inside un-profiled replays (tools/critical_path.py):
 * mgn_plan_trace(on): the launches of the following replays carry a hipEvent pair bound to the dispatch itself (no marker packets);
 * mgn_plan_trace_read after such a replay has completed: per node (n = node count), ms relative to the begin of node `ref`:
 *   t_a = elapsed(ref.begin_event, node.begin_event), t_b = elapsed(ref.begin_event, node.end_event), dur = elapsed(node.begin_event,
 *   node.end_event); NaN for marks and skipped nodes;
 * mgn_plan_set_skip: what-if replays -- skip 1: the node is not launched (results meaningless, timing = the step without it);
 *   skip 2: the node is launched twice back to back (its cost as an increase, data intact for pure kernels); skip 0: as recorded. */
int mgn_plan_trace(void* plan, int on);
int mgn_plan_trace_read(void* plan, int ref, int n, float* t_a, float* t_b, float* dur);
int mgn_plan_set_skip(void* plan, int node, int skip);
/* race hunting: random idle kernels (1 .. max_us us, probability permille / 1000 per launch) in front of a replay's launches; a complete set
 * of cross-stream edges gives the same bits under any timing (tests/test_plan_gpu.py).  permille 0 = off. */
int mgn_plan_set_jitter(void* plan, unsigned long long seed, int permille, int max_us);
/* debugging: after node `node` of every following replay an order-independent checksum of [ptr, ptr + nbytes) is ADDED to *out (8 bytes of
 * device memory the caller zeroes between replays) on the node's stream -- two replays of one step compared buffer by buffer; node | 1 << 24: the range is COPIED to out (nbytes) instead;
 * node < 0 clears the probes */
int mgn_plan_probe(void* plan, int node, const void* ptr, size_t nbytes, void* out);
int mgn_plan_free(void* plan);

/* The 7x7 / stride 2 / pad 3 stems with 64 output channels (res_net.py:96-104 BasicStem conv1; the 9-channel pose-net stem,
 * res_net.py:169-181) on the channel-padded input of mgn_prep_input (Cin = 4 | 8 | 16) and the layout-mode-2 weights of
 * mgn_weight_layout: persistent windowed kernel with the weights in registers (csrc/conv_stem.hip); mgn_conv_igemm dispatches
 * here.  Cin = 4 (3 real channels, IW even) is the backbone stem at its dense reduction: a kernel row is one run of 8 column
 * slots x 4 channels (K = 224 for 147 real products, against 392 on the 8-channel input).  stat_partials: NULL or [mgn_conv_stem7_blocks(...)][64][2] sums of r, r^2 over the rounded outputs.  MGN_ENOTSUP for
 * other shapes (mgn_conv_stem7_blocks == 0). */
int mgn_conv_stem7(const void* in, const void* w_packed, void* out, int N, int IH, int IW, int Cin, int OH, int OW, int Cout,
                   float* stat_partials, void* stream);
int mgn_conv_stem7_f16(const void* in, const void* w_packed, void* out, int N, int IH, int IW, int Cin, int OH, int OW, int Cout,
                       float* stat_partials, void* stream);
int mgn_conv_stem7_blocks(int N, int IH, int IW, int Cin, int OH, int OW, int Cout);
/* patch height mgn_conv_igemm picks for a 3x3 / stride 1 / pad 1 layer of this shape: 16, 8, or 0 (= it uses another kernel) */
int mgn_conv_win_patch_rows(int N, int OH, int OW, int Cin, int Cout);
/* fp32 OIHW master weights -> bf16 kernel layout. mode 0: [Cout][KH][KW][Cin]; 1: [Cin][KH][KW][Cout] with flipped taps
 * (data gradient); 2: packed-tap stem layout [Cout][ceil(KH*KW*Cp/32)*32] for a Cp-channel (8|16) padded input; Cp = 4 (7x7 only):
 * the dense-row layout [Cout][224], k = kh*32 + (kw+1)*4 + c */
int mgn_weight_layout(const float* w_oihw, void* out_bf16, int Cout, int Cin, int KH, int KW, int mode, int Cp,
                      int cout_pad /* > Cout: output channels zero-padded to cout_pad in the layout (few-class predictors) */, void* stream);
/* every conv weight of a model in one launch (after the optimizer step): table_dev = n_entries rows of 8 x int64 on the
 * device {src fp32 OIHW pointer, dst bf16 pointer, n_out elements, first block (prefix sum of ceil(n_out/256)), Cout | cout_pad << 32,
 * Cin, KH << 32 | KW, mode << 32 | Cp}, rows sorted by first block; total_blocks = sum of ceil(n_out/256). */
int mgn_weight_layout_batch(const void* table_dev, int n_entries, long total_blocks, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Panoptic-head losses fused with the bilinear (align_corners=True) upsampling of the low-resolution head outputs
 *   mgn_upce_*     replace mg_net.py:597-610 (F.interpolate x8 of the logits) + loss.py:45-81 OhemCE / :9-42 DeepLabCE:
 *                  per-pixel weighted CE map (fp32 [B,H,W]) and sums3 = {count(ce>thr), sum(ce|ce>thr), sum(ce)};
 *                  the caller applies the OHEM rule (count > n_min ? mean of {ce>thr} : mean of the n_min largest).
 *                  backward: sel3 = {tau, tie_weight, scale}: dL/dce = scale*(ce>tau ? 1 : ce==tau ? tie_weight : 0);
 *                  dlogits [B,h,w,Kp] fp32 must be zero-initialised.
 *   mgn_ins_loss_* replace mg_net.py:676-715: centre map (low-res AFTER sigmoid, fp32) and offset map (low-res bf16)
 *                  upsampled on the fly; out4 = {sum w(c-t)^2 / sum w, sum w|o*s-t| / sum w, sum w_c, sum w_o};
 *                  backward dco [B,h,w,4] fp32 (zero-initialised) = {d centre_lr, d offset_lr[0], d offset_lr[1], 0}.
 * Low-res maps: bf16 (logits/offset) with channel stride 1 and element strides (sb, sh, sw) multiples of 8.
 * The three backward entry points (mgn_upce_bwd, mgn_ins_loss_bwd, mgn_upsample1_bwd) compute the bilinear ADJOINT per pixel
 * tile (32 x 8 for mgn_upce_bwd, 32 x 16 for the others).  footprints == NULL: the tiles add their low-res footprints with float atomics (the destination must be
 * zero-initialised; sums are order-dependent in the last bits).  footprints != NULL (mgn_adjoint_footprint_floats floats; channels =
 * K | 3 | 1): every tile stores its footprint into its own slot and a second kernel sums, per low-res element, the slots
 * that cover it in a fixed order -- bit-reproducible, the destination need not be initialised.
 * ---------------------------------------------------------------------------------------------- */
int mgn_upce_partials(int B, int H, int W);
int mgn_adjoint_footprint_floats(int which /* 0: mgn_upce_bwd, 1: mgn_ins_loss_bwd, 2: mgn_upsample1_bwd */, int B, int h, int w, int H,
                                 int W, int channels, size_t* floats);
/* single-channel fp32 bilinear (align_corners=True) upsampling [B,1,h,w] -> [B,1,H,W] and its adjoint (dlr zero-initialised);
 * replaces F.interpolate at mg_net.py:804-807 (depth head, x8/x16/x32). The adjoint needs an upsampling factor >= 7. */
int mgn_upsample1_fwd(const float* lr, int B, int h, int w, int H, int W, float* out, void* stream);
int mgn_upsample1_bwd(const float* dfull, int B, int h, int w, int H, int W, float* dlr_zeroed, float* footprints, void* stream);
int mgn_upce_fwd(const void* logits_bf16, long sb, long sh, long sw, int B, int h, int w, int H, int W, int K,
                 const long* labels, const float* weights, int ignore, float thr, float* ce_map, float* partials,
                 float* sums3, void* stream);
int mgn_upce_bwd(const void* logits_bf16, long sb, long sh, long sw, int B, int h, int w, int H, int W, int K, int Kp,
                 const long* labels, const float* weights, int ignore, const float* ce_map, const float* sel3,
                 const float* gout, float* dlogits, float* footprints, void* stream);
/* OhemCE / DeepLabCE selection on the device (replaces the full torch.sort of loss.py:67-81 and its host-side branch):
 * from the per-pixel loss map and sums3 of mgn_upce_fwd -> sel3 = {tau, tie_weight, scale} for mgn_upce_bwd and the loss.
 * count(ce > thr) > n_sel: mean of {ce > thr}; otherwise (or force_topk: DeepLabCE hard-pixel mining) the mean of the n_sel
 * largest values, found by a radix select (no sort, no host synchronisation). */
int mgn_ohem_select_workspace_bytes(long n, size_t* bytes);
int mgn_ohem_select(const float* ce_map, long n, const float* sums3, float thr, long n_sel, int force_topk, float* sel3,
                    float* loss, void* workspace, size_t workspace_bytes, void* stream);
int mgn_ins_loss_fwd(const float* center_lr, long csb, long csh, long csw, const void* offset_lr_bf16, long osb, long osh,
                     long osw, int B, int h, int w, int H, int W, const float* ct, const float* cw, const float* ot,
                     const float* ow, float oscale, float* partials, float* out4, void* stream);
int mgn_ins_loss_bwd(const float* center_lr, long csb, long csh, long csw, const void* offset_lr_bf16, long osb, long osh,
                     long osw, int B, int h, int w, int H, int W, const float* ct, const float* cw, const float* ot,
                     const float* ow, float oscale, const float* out4, const float* gout2, float* dco, float* footprints,
                     void* stream);

/* ------------------------------------------------------------------------------------------------
 * Network input assembly -- replaces mg_net.py:250-264 (`.float()/255`, mean/std normalisation of image, image_prev,
 * image_next and their channel concatenation for PoseCNN).
 * frames_u8: HOST array of n_frames (1..3) device pointers to [B,3,H,W] uint8; pixel_mean3/std3: HOST floats in the 0..1
 * domain (cfg value / 255); out: [B,H,W,Cp] bf16 channels-last, Cp = 4 (one frame), 8 or 16, channels 3f..3f+2 = frame f, rest zero.
 * mgn_conv_igemm / mgn_conv_wgrad accept such Cin = 8 | 16 inputs ("packed taps": k = tap*Cin + c, weights
 * [Cout][ceil(KH*KW*Cin/32)*32] bf16, dw [Cout][KH*KW*Cin] fp32) and, for the 7x7 / stride 2 / pad 3 / 64-channel stem on even widths
 * only, Cin = 4 ("dense rows": k = kh*32 + (kw+1)*4 + c, weights [Cout][224]).
 * ---------------------------------------------------------------------------------------------- */
int mgn_prep_input(const void* const* frames_u8, int n_frames, int B, int H, int W, const float* pixel_mean3,
                   const float* pixel_std3, void* out_bf16, int Cp, void* stream);

/* uint8 frames -> fp32 `x / divisor` stacked into one batch tensor -- replaces the `.float() / 255` of the un-jittered frames
 * of the photometric loss and their stacking (mg_net.py:320-335).  frames_u8: HOST array of n_frames (<= 16) device pointers
 * (16-byte aligned) to n_per_frame bytes each (n_per_frame % 16 == 0); out: [n_frames, n_per_frame] fp32.  IEEE division. */
int mgn_u8_frames_to_f32(const void* const* frames_u8, int n_frames, long n_per_frame, float divisor, float* out, void* stream);
/* the same conversion of n_frames [3,H,W] uint8 frames (hw = H*W, a multiple of 4) into ONE pixel-interleaved RGBx batch
 * [n_frames][H][W][4] fp32 (4th channel 0): the context-frame layout MGN_FRAMES_CTX_RGBX_F32 of mgn_reproj_cfg.frame_layout */
int mgn_u8_frames_to_f32_nhwc4(const void* const* frames_u8, int n_frames, long hw, float divisor, float* out, void* stream);
/* n_frames (<= 48: three frame sets of a batch of 16) [3,H,W] uint8 frames (4-byte aligned, hw = H*W a multiple of 4) -> ONE
 * pixel-interleaved uint8 batch [n_frames][H][W][4] (R,G,B,0): the layout MGN_FRAMES_RGBX_U8 of the reprojection loss.  The un-jittered
 * frames of the photometric loss (mg_net.py:320-335) stay bytes until the loss kernel converts them in registers: 7 B/px of traffic
 * here instead of the 15 B/px of mgn_u8_frames_to_f32, and 4 B/px per frame in the loss instead of 12. */
int mgn_u8_frames_to_rgbx(const void* const* frames_u8, int n_frames, long hw, void* out_u8, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Multi-scale + flip inference: the per-pass tensor algebra of mg_net.py:427-520 `forward_multi_scale_flip`.
 *
 * mgn_msc_input: `x = F.interpolate(norm_images, scale_factor=scale, bilinear, align_corners=True)` (+ `torch.flip(x, dims=(3,))`,
 *   mg_net.py:451-455) written as the network input: norm_nchw [N,3,H,W] fp32 -> out [N,h,w,8] bf16 (out_f16 = 0) or fp16 (1),
 *   channels-last, channels 3..7 zero (the stem kernels' layout, like mgn_prep_input); out_f16 = 2: [N,h,w,3] fp32 for an fp32 trunk
 *   (SOLVER.AMP.ENABLED False, the reference's PseudoLabelGeneration yamls); h, w = floor(H * scale), floor(W * scale).
 * mgn_msc_accumulate: one head output of one pass added to its running average (mg_net.py:462-512):
 *   lr      low-resolution head output [N,C,h,w] with element strides sn, sc, sh, sw; dtype 0 = fp32, 1 = bf16, 2 = fp16; C <= 32
 *   acc     [N,C,H,W] fp32 contiguous: the running sum; first != 0: written, not added to (no zero fill needed)
 *   mode    0  softmax over C of the upsampled logits       (sem_seg, :464-470)
 *           1  the upsampled map itself                     (center, :471-476)
 *           2  (upsampled * stride) / scale, channel 1 (x) negated when flip   (offset, :477-493)
 *           3  1 / max(upsampled, 1e-6)                     (depth = inv2depth, :499-507)
 *   flip    the pass ran on the mirrored frame: output column x takes the upsampled map's column W-1-x (:487-490)
 *   divide  > 0 on the last pass: acc = (acc + v) / divide  (the averages of :514-520, rounded like sum / n)
 *   Upsampling = F.interpolate(bilinear, align_corners=True) from [h,w] to [H,W].
 * ---------------------------------------------------------------------------------------------- */
int mgn_msc_input(const float* norm_nchw, int N, int H, int W, int h, int w, int flip, int out_f16, void* out_nhwc8, void* stream);
int mgn_msc_accumulate(const void* lr, int dtype, long sn, long sc, long sh, long sw, int N, int C, int h, int w, int H, int W, int mode,
                       int flip, int first, float stride, float scale, float divide, float* acc, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Uncertainty weighting of the task losses (mg_net.py:360-372), all tasks in one launch:
 *     weighted[k] = tau_k * exp(-log_vars[k]) * raw_losses[k][0] + 0.5 * log_vars[k],  tau_k = 1 if bit k of tau_one_mask else 0.5
 *     uncertainty[k] = exp(log_vars[k])                      (the "<loss>_uncertainty" scalar of the event storage, :369)
 * raw_losses / grads: HOST arrays of n (<= MGN_MAX_TASKS) DEVICE pointers to fp32 scalars (the loss kernels' outputs where they lie;
 * a null grads[k] = that output was not used).  Backward: d_raw[k] = tau_k exp(-lv_k) g_k, d_log_vars[k] = (0.5 - tau_k exp(-lv_k)
 * raw_k) g_k for k < n and 0 for n <= k < n_log_vars.
 * ---------------------------------------------------------------------------------------------- */
#define MGN_MAX_TASKS 8
int mgn_uncertainty_fwd(const float* const* raw_losses, int n, const float* log_vars, unsigned tau_one_mask, float* weighted, float* uncertainty,
                        void* stream);
int mgn_uncertainty_bwd(const float* const* raw_losses, const float* const* grads, int n, int n_log_vars, const float* log_vars,
                        unsigned tau_one_mask, float* d_raw, float* d_log_vars, void* stream);

/* Small host -> device table upload (per-step learning-rate tables of the optimizer, descriptor tables of the batched launches) as a
 * kernel that reads PINNED host memory directly: dst_dev[0 .. nbytes) = src_pinned_host[0 .. nbytes), nbytes a multiple of 4, both 4-byte
 * aligned; stream-ordered (the host buffer must stay unchanged until the launch has run).  Replaces the `tensor.to(device,
 * non_blocking=True)` of those tables (detectron2's / torch.optim's per-step host values reach the device the same way). */
int mgn_copy_from_host(void* dst_dev, const void* src_pinned_host, size_t nbytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Tails of the prediction heads: between a 1x1 predictor (output channels padded to P = 32, channels-last, 16-bit) and its loss.
 * mgn_head_act_fwd: y[B,C,h,w] fp32 = f(x_padded[B,h,w,P][..., :C].float()); kind 0: identity, 1: sigmoid (centre heat map,
 *   mg_net.py:694), 2: sigmoid / 0.5 (inverse depth, :819-823).
 * mgn_head_act_bwd: dx_padded[B,h,w,P] 16-bit = f'(.) * gscale * g for channels < C, ZERO for the padding channels (what the
 *   predictor's data / weight gradient kernels consume); g is fp32 with element strides sb / sc / sp per image / channel / pixel
 *   (an NCHW map: C*h*w, h*w, 1; an NHWC table of the loss kernels with pitch Kp: h*w*Kp, 1, Kp); y = the forward's output (kind != 0).
 * Replaces per head and step: .float(), sigmoid, / 0.5 and their autograd twins + the zero-fill and strided copy of the channel slice.
 * ---------------------------------------------------------------------------------------------- */
int mgn_head_act_fwd(const void* x_padded, int B, int h, int w, int P, int C, int kind, int is_f16, float* y, void* stream);
int mgn_head_act_bwd(const float* g, long sb, long sc, long sp, const float* y, int B, int h, int w, int P, int C, int kind, int is_f16,
                     float gscale, void* dx_padded, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Peer-to-peer exchange of the SyncBN statistics between the ranks of ONE node (one process per GPU), csrc/p2p.hip.
 * Replaces the all_gather (forward: 3*C floats) and all_reduce (backward: 2*C floats) that inplace_abn.InPlaceABNSync issues per
 * layer (68 sites: res_net.py:35,49,59,103, layers.py:63,71,117,209,242,253,291) -- 136 latency-bound collectives per step -- by one
 * small kernel each on the compute stream: every rank PUSHES its row into a mailbox in every peer's memory (xGMI is point-to-point),
 * raises a flag there (release, system scope) and waits for the peers' flags in its own mailbox (acquire).
 *   mgn_p2p_alloc / _free     this rank's mailbox: fine-grained device memory of mgn_p2p_mailbox_bytes(), zeroed (MGN_ENOTSUP if the
 *                             runtime refuses such an allocation)
 *   mgn_p2p_export / _open / _close   64-byte IPC handle of the own mailbox (to be sent to the peers by any means, e.g.
 *                             torch.distributed.all_gather_object) / mapping of a peer's mailbox into this process
 *   mgn_p2p_exchange          mailboxes: HOST array of `world` device pointers in rank order (own at [rank]); channel < MGN_P2P_CHANNELS
 *                             = one per stream that issues exchanges; seq = 1, 2, 3, ... per channel, the same on every rank; payload:
 *                             n <= MGN_P2P_SLOT_FLOATS fp32; reduce = 0: out[world][n] = every rank's payload (all_gather), 1: out[n] =
 *                             sum over ranks in rank order (all_reduce; bit-identical on every rank).  payload and out must not overlap.
 *                             seq_counters (optional): DEVICE array [MGN_P2P_CHANNELS][MGN_P2P_MAX_WORLD] of uint32, zeroed once; when
 *                             given, the kernel counts the channel's exchanges itself and `seq` is ignored (no per-launch host value:
 *                             a recorded step can be replayed, mgn_plan_*).
 *                             status: int in HOST-VISIBLE memory (pinned), set to 1 if a peer did not post within timeout_s; the
 *                             output is then filled with NaN (a timed-out exchange makes the step's losses non-finite at once).
 * Every rank must issue the same exchanges in the same order per channel, and exchanges of one channel must be stream-ordered.
 * ---------------------------------------------------------------------------------------------- */
#define MGN_P2P_CHANNELS 4
#define MGN_P2P_SLOT_FLOATS 3072
#define MGN_P2P_MAX_WORLD 8
#define MGN_P2P_HANDLE_BYTES 64
size_t mgn_p2p_mailbox_bytes(void);
int mgn_p2p_alloc(void** mailbox);
int mgn_p2p_free(void* mailbox);
int mgn_p2p_export(void* mailbox, void* handle64);
int mgn_p2p_open(const void* handle64, void** peer_mailbox);
int mgn_p2p_close(void* peer_mailbox);
int mgn_p2p_exchange(void* const* mailboxes, int world, int rank, int channel, unsigned seq, unsigned* seq_counters, const float* payload,
                     int n, int reduce, float* out, int* status, float timeout_s, void* stream);

/* The stem's activated batch norm folded into the pooling (BasicStem, res_net.py:82-110: conv -> InPlaceABNSync -> max_pool):
 * forward pools y = act(scale * x + offset) evaluated on the fly (bf16-rounded like mgn_iabn_apply stores it; the
 * normalised map is never written), backward = mgn_iabn_bwd_reduce on (pooled, d pooled) for the channel sums, then
 * mgn_abn_maxpool_bwd: gather of d y per 2x2 input patch + the norm's dx formula with z recomputed from the saved conv
 * output x.  scale/offset/rstd: rows 0, 1, 3 of the coefficient block of mgn_iabn_train_coeffs / mgn_iabn_combine;
 * sums: [2,C] of mgn_iabn_bwd_reduce (all-reduced over ranks by the caller); total_count = pixels of x over all ranks. */
int mgn_abn_maxpool_fwd(const void* x_bf16, const float* scale, const float* offset, int activation, float slope, void* y_bf16,
                        uint8_t* argmax, int N, int IH, int IW, int C, void* stream);
int mgn_abn_maxpool_bwd(const void* x_bf16, const void* dpool_bf16, const uint8_t* argmax, void* dx_bf16, const float* scale,
                        const float* offset, const float* weight, const float* bias, const float* rstd, const float* sums,
                        float total_count, float eps, int activation, float slope, int N, int IH, int IW, int C, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Panoptic training targets on the device (the producer of the hot path's target maps)
 *   replaces mgnet/data/target_generator.py:54-158  PanopticDeepLabTargetGenerator.__call__,
 *   the `rgb2id` decode before it (mgnet/data/dataset_mapper.py:178) and the per-class part of the
 *   reprojection mask (dataset_mapper.py:214-216).
 * Inputs
 *   panoptic   : pan_rgb == 0: int32 [B,H,W] segment ids;  pan_rgb == 1: uint8 [B,H,W,3] label image as read from the
 *                PNG (id = R + 256 G + 65536 B)
 *   seg_ids    : int32 [B, max_segments], the ids of segments_info, ASCENDING and unique per image (host sorts)
 *   seg_attr   : int32 [B, max_segments] in the same order: category_id (0..255) | iscrowd << 8 | is_thing << 9
 *   seg_count  : int32 [B] number of valid table rows per image (<= max_segments <= MGN_TARGETS_MAX_SEGMENTS)
 *   gauss      : fp32 [(6 sigma + 3)^2], the Gaussian patch of target_generator.py:45-50 rounded to fp32
 * Outputs (dtypes and shapes of the reference's dict entries, stacked over the batch)
 *   sem_seg int64 [B,H,W]; center fp32 [B,H,W]; offset fp32 [B,2,H,W] (dy, dx); sem_seg_weights fp32 [B,H,W];
 *   center_weights, offset_weights fp32 [B,1,H,W]; reprojection_mask uint8 [B,H,W] (optional, NULL = skip):
 *   0 where sem_seg is one of the classes in depth_ignore_mask; center_points fp64 [B,max_segments,2] (optional; (cy,cx)
 *   per table row, NaN where the row has no centre); seg_area int64 [B,max_segments] (optional)
 * Integer statistics + a gather formulation of the heat map: results do not depend on the launch shape.
 * Algorithmic HBM bytes: 2 reads of the labels + one write of every map = 41 B/px (int32 labels, with mask).
 * ---------------------------------------------------------------------------------------------- */
#define MGN_TARGETS_MAX_SEGMENTS 1024
typedef struct {
    int B, H, W;
    int pan_rgb;                  /* 0: int32 ids, 1: uint8 RGB label image                                           */
    int ignore_label;             /* sem_seg value of pixels outside every segment (0..255)                           */
    int sigma;                    /* Gaussian sigma (INPUT.GAUSSIAN_SIGMA, config.py:51), 1..64                        */
    int first_thing_id;           /* thing_ids[0]: sem_seg < first_thing_id gets center weight 1 (:146)               */
    int ignore_stuff_in_offset;
    int small_instance_area;
    int small_instance_weight;
    int ignore_crowd_in_semantic;
    int legacy_promotion;         /* 1: offsets = f32(center) - f32(coord) (NumPy < 2 value-based casting, the only
                                     NumPy the reference's np.bool runs on); 0: f32(center64 - coord) (NEP 50)         */
    int max_segments;             /* row stride of the segment tables                                                 */
    uint32_t depth_ignore_mask[8];/* bit c set: class c is excluded from the photometric loss                         */
} mgn_targets_cfg;

int mgn_panoptic_targets_workspace_bytes(const mgn_targets_cfg* cfg, size_t* bytes);
int mgn_panoptic_targets(const mgn_targets_cfg* cfg, const void* panoptic, const int32_t* seg_ids, const int32_t* seg_attr,
                         const int32_t* seg_count, const float* gauss, int64_t* sem_seg, float* center, float* offset,
                         float* sem_seg_weights, float* center_weights, float* offset_weights, uint8_t* reprojection_mask,
                         double* center_points, int64_t* seg_area, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Panoptic fusion of the inference path (the consumer of the hot path's head outputs)
 *   replaces mgnet/postprocessing/panoptic_post_proc.py:9-147  get_panoptic_prediction
 *   (+ _group_instances_and_fuse_logits), one image per call like the reference (mg_net.py:375).
 * Inputs  : sem_seg int64 [H,W] (argmax labels), center_heatmap fp32 [H,W], offsets fp32 [2,H,W] (dy, dx); not modified
 *           (the reference adds the pixel grid into `offsets` and scatters into `sem_seg` in place).
 * Outputs : panoptic int64 [H,W]; info int32[2] = { centres that survived the NMS, 1 if that exceeded
 *           MGN_PANOPTIC_MAX_CENTERS (the list is cut there; the reference's own 65535 sentinel breaks at the same point) }
 * Centre order = row-major (torch.nonzero), nearest centre = first minimum of the fp32 L2 distance, class vote = first
 * maximum: the instance ids equal the reference's.  No host synchronisation.
 * ---------------------------------------------------------------------------------------------- */
#define MGN_PANOPTIC_MAX_CENTERS 65534
typedef struct {
    int H, W;
    int num_thing_classes;   /* len(thing ids)                                              */
    int last_stuff_id;       /* max contiguous stuff id; sem_seg > last_stuff_id = thing     */
    int label_divisor;       /* panoptic id = class * label_divisor + instance               */
    int stuff_area;          /* MODEL.POST_PROCESSING.STUFF_AREA (config.py:122)             */
    int void_label;          /* -1 (mg_net.py:166)                                           */
    float threshold;         /* CENTER_THRESHOLD (config.py:123)                             */
    int nms_kernel;          /* NMS_KERNEL (config.py:124), odd                              */
} mgn_panoptic_cfg;

int mgn_panoptic_post_workspace_bytes(const mgn_panoptic_cfg* cfg, size_t* bytes);
int mgn_panoptic_post(const mgn_panoptic_cfg* cfg, const int64_t* sem_seg, const float* center_heatmap, const float* offsets,
                      int64_t* panoptic, int32_t* info, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Instance predictions from a panoptic prediction (TEST.EVAL_INSTANCE)
 *   replaces mgnet/postprocessing/instance_post_proc.py:11-72  get_instance_predictions
 * sem_logits fp32 [C,H,W] (the per-image semantic logits, softmax is taken inside), center_heatmap fp32 [H,W], panoptic int64
 * [H,W]; a segment is a thing when (id / label_divisor) is a set bit of thing_mask.  Outputs (device), segments in ascending id
 * order (np.unique): labels int64[n], classes int32[n], scores fp32[n] = mean softmax probability of the class over the mask *
 * center_heatmap[int(mean y), int(mean x)], boxes fp32[n][4] = x_min, y_min, x_max + 1, y_max + 1; info int32[2] = {n, overflow};
 * every array holds MGN_INSTANCE_MAX entries.  mgn_instance_masks fills masks uint8 [n][H][W] (zero-initialised by the caller).
 * Integer accumulation (32.32 fixed point for the probabilities): results do not depend on the launch shape.
 * ---------------------------------------------------------------------------------------------- */
#define MGN_INSTANCE_MAX 4096
typedef struct {
    int H, W, C;
    int label_divisor;
    uint64_t thing_mask;      /* bit c set: class c is a thing (thing_ids of the dataset metadata); C <= 64 */
} mgn_instance_cfg;
int mgn_instance_post_workspace_bytes(const mgn_instance_cfg* cfg, size_t* bytes);
int mgn_instance_post(const mgn_instance_cfg* cfg, const float* sem_logits, const float* center_heatmap, const int64_t* panoptic,
                      int64_t* labels, int32_t* classes, float* scores, float* boxes, int32_t* info, void* workspace,
                      size_t workspace_bytes, void* stream);
int mgn_instance_masks(const mgn_instance_cfg* cfg, const int64_t* panoptic, const int64_t* labels, int n, uint8_t* masks_zeroed,
                       void* stream);
/* Pseudo-label images -- replaces the id arithmetic of tools/generate_pseudo_labels.py:100-118: a panoptic
 * prediction in train ids (int64, void = -1) -> the dataset's `instanceIds` image (uint16): stuff -> id_map[class], things ->
 * id_map[class] * label_divisor + instance.  id_map256: uint8[256] on the device (trainId -> id, zeros elsewhere). */
int mgn_pseudo_label_ids(const int64_t* panoptic, long n_pixels, int label_divisor, const uint8_t* id_map256, uint16_t* out,
                         void* stream);

/* ------------------------------------------------------------------------------------------------
 * Depth post-processing with DGC metric rescaling
 *   replaces mgnet/postprocessing/depth_post_proc.py:11-185  get_depth_prediction (+ _get_scale_recovery,
 *   _get_surface_normal with nei = 1, _get_ground_mask with 5 degrees) and Camera.reconstruct (geometry/camera.py:107-141).
 * depth fp32 [H,W] (H, W >= 3); panoptic int64 [H,W] or NULL (has_panoptic = 0: the ground mask comes from the surface
 * normals); depth_out fp32 [H,W]; xyz fp32 [3,H,W] (camera-frame points, required when use_dgc_scaling); scale: device
 * fp32[1] = real_camera_height / median ground height (NaN when no pixel is ground; 1 without DGC).
 * The median is torch.median's (the lower middle element), found by a 4-pass radix select, not a sort.
 * ---------------------------------------------------------------------------------------------- */
#define MGN_DEPTH_MAX_FILTER_IDS 16
typedef struct {
    int H, W;
    int use_dgc_scaling;          /* MODEL.POST_PROCESSING.USE_DGC_SCALING (config.py:126)                   */
    int has_panoptic;
    float fx, fy, cx, cy;         /* camera_matrix[0,0], [1,1], [0,2], [1,2]                                 */
    float real_camera_height;
    int n_filter;                 /* number of valid filter_ids                                              */
    int64_t road_class_id;        /* panoptic id of the road class (trainId * label_divisor, mg_net.py:173-180) */
    int64_t filter_ids[MGN_DEPTH_MAX_FILTER_IDS]; /* panoptic ids whose depth is set to 0 / points to NaN (:62-68) */
} mgn_depth_post_cfg;

int mgn_depth_post_workspace_bytes(const mgn_depth_post_cfg* cfg, size_t* bytes);
int mgn_depth_post(const mgn_depth_post_cfg* cfg, const float* depth, const int64_t* panoptic, float* depth_out, float* xyz,
                   float* scale, void* workspace, size_t workspace_bytes, void* stream);

/* Depth metrics of one frame -- replaces the arithmetic of mgnet/evaluation/depth_evaluation.py:70-112
 * (DepthEvaluator.process): mask = min_depth < label < max_depth inside rows [crop_y0, crop_y1) x columns [crop_x0, crop_x1)
 * (the Eigen crop, or the whole frame), optional median scaling (np.median of both maps: two radix selects each), clamp,
 * then out9 (device fp64) = { abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3, scale ratio, number of masked pixels }.
 * prediction, label: fp32 [H,W].  fp64 accumulation in a fixed order. */
int mgn_depth_metrics_workspace_bytes(int H, int W, size_t* bytes);
int mgn_depth_metrics(const float* prediction, const float* label, int H, int W, float min_depth, float max_depth,
                      int use_gt_scale, int crop_y0, int crop_y1, int crop_x0, int crop_x1, double* out9, void* workspace,
                      size_t workspace_bytes, void* stream);

/* 3x3 / stride 2 / pad 1 max pooling of the ResNet stems (res_net.py:109) on channels-last bf16 [N,IH,IW,C] (C % 8 == 0);
 * argmax: 1 byte per output element (winning tap 0..8); backward is a deterministic gather. */
int mgn_maxpool3x3s2_fwd(const void* x_bf16, void* y_bf16, uint8_t* argmax, int N, int IH, int IW, int C, void* stream);
int mgn_maxpool3x3s2_bwd(const void* dy_bf16, const uint8_t* argmax, void* dx_bf16, int N, int IH, int IW, int C, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Channel-attention vectors (mgnet/modeling/layers.py:248-267 AttentionRefinementModule.channel_attention,
 * :297-322 FeatureFusionModule.channel_attention): a 1x1 conv on the pooled [N, K] vectors (+ InPlaceABNSync over the N
 * samples of ONE process, + activation) as one launch.  fp32; W is the conv's fp32 master weight [C][K] (= [C,K,1,1]).
 * act: 0 none, 1 ReLU, 2 sigmoid.  bn_weight == NULL: no norm.  Training keeps xhat [N][C] and rstd [C] for the backward.
 * N <= 64, N*K*4 <= 96 KB.  One block per 16 output channels; the input gradient is summed over the blocks in a fixed order.
 * ---------------------------------------------------------------------------------------------- */
int mgn_vec_linear_fwd(const float* in, const float* W, int N, int K, int C, int act, const float* bn_weight,
                       const float* bn_bias, float* running_mean, float* running_var, int training, float momentum,
                       float eps, float* out, float* xhat, float* rstd, void* stream);
int mgn_vec_linear_bwd_workspace_bytes(int N, int K, int C, size_t* bytes);
int mgn_vec_linear_bwd(const float* dout, const float* out, const float* in, const float* W, int N, int K, int C, int act,
                       const float* bn_weight, const float* xhat, const float* rstd, float eps, float din_scale /* din is
                       multiplied by it (1/HW of the pool) */, float* dW, float* din, float* dbn_weight, float* dbn_bias,
                       void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Element-wise / broadcast / pooling glue of the blocks, channels-last bf16 [N, H*W, C], C % 8 == 0
 *   mgn_add_relu_fwd / mgn_relu_mask_bwd : res_net.py:77-78 (out + shortcut, relu_)
 *   mgn_colsum                          : out[n,c] = scale * sum_r x[n,r,c] (* x2[n,r,c]); layers.py:170-184 global
 *                                         average pool (scale = 1/HW) and the d/d attention reduction; deterministic
 *   mgn_bcast_rows                      : dx[n,r,c] = g[n,c] * scale (adjoint of the pool)
 *   mgn_scale_channels                  : y = x * s[n,c] (mode 0, layers.py:262-267) | x * (1 + s[n,c]) (mode 1, :315-322)
 *                                         [+ add[n,c]] [+ addt[n,r,c]]
 *   mgn_nearest_fwd / _bwd              : F.interpolate(mode="nearest") (layers.py:90, :217) and its adjoint
 *   mgn_concat2 / mgn_split2            : torch.cat([a, b], dim=1) (layers.py:316) and the split of its gradient
 * ---------------------------------------------------------------------------------------------- */
int mgn_add_relu_fwd(const void* a, const void* b, void* y, long n_elems, void* stream);
/* y = a + b (+ c) on 16-bit tensors, fp32 sum, one rounding: the gradient of a tensor consumed by three branches (the backbone
 * features feed the semantic, instance and depth heads, mg_net.py:290-311) in one pass; c may be NULL */
int mgn_sum3(const void* a, const void* b, const void* c, void* y, long n_elems, void* stream);
/* BasicBlock tail (res_net.py:62-79) with the second InPlaceABNSync(identity) folded in:
 * y = relu(bf16(scale[c] * x + offset[c]) + shortcut), x [M,C] bf16 = conv2 output (kept for the backward), C % 8 == 0.
 * relu_bits (nullable): [M*C/8] bytes, bit k of byte i = (y[8 i + k] > 0) on the rounded 16-bit output -- the ReLU mask the backward
 * (mgn_iabn_bwd_reduce_x_relu) then reads instead of y. */
int mgn_abn_add_relu_fwd(const void* x, const float* scale, const float* offset, const void* shortcut, void* y, void* relu_bits /*nullable*/,
                         long M, int C, void* stream);
int mgn_relu_mask_bwd(const void* dy, const void* y, void* dx, long n_elems, void* stream);
int mgn_colsum(const void* x, const void* x2 /*nullable*/, int N, long HW, int C, float scale, float* out, float* workspace,
               size_t workspace_bytes /* >= N*64*C*4 */, void* stream);
int mgn_bcast_rows(const float* g, int N, long HW, int C, float scale, void* dx, void* stream);
int mgn_scale_channels(const void* x, const float* s, int N, long HW, int C, int mode, const float* add /* nullable:
                       y += add[n,c] (the pooled branch of the attention backward) */, const void* addt /* nullable: y += addt[n,r,c], a
                       16-bit tensor of x's shape: `arm(x) + last` of the decoder, layers.py:87, in the same pass */, void* y, void* stream);

/* InPlaceABNSync followed by a channel-attention module (mgnet/modeling/layers.py:221-267 AttentionRefinementModule, :270-322
 * FeatureFusionModule: `fm = conv+norm(x); fm * sigmoid(attention(avg_pool(fm)))` / `fm + fm * ...`), the norm's and the attention's
 * passes over the activation fused (csrc/eltwise.hip): forward 2 R + 2 W instead of 3 R + 2 W, backward 4 R + 1 W instead of 7 R + 2 W.
 * x, z, g, dy: 16-bit [N, HW, C] channels-last, C % 8 == 0; act 0 identity / 1 leaky_relu(slope).
 * mgn_abn_apply_pool: z = act(scale[c] x + offset[c]) (z may be x: in place) and pooled[n][c] = pool_scale * sum_rows(rounded z);
 *   workspace >= N * 256 * C floats.
 * mgn_att_abn_bwd_stats: S[5][N][C] (quantity-major) = sums over the image's rows of {g z, g m, g m xh, m, m xh}, m = act'(z), xh = (act^-1(z) - bias) /
 *   (|weight| + eps); workspace >= N * 64 * 5 * C floats.
 * mgn_att_abn_bwd_sums: sums[2][C] = {sum dz, sum dz xh} of the norm's backward for dz = (g base + dpool) m, base = s (mode 0) | 1 + s
 *   (mode 1), dpool[N][C] = the pooled branch's gradient per element (NULL = 0); dwb[2][C] = {d weight, d bias} or NULL.
 * mgn_att_abn_bwd_apply: dy = (|w| + eps) rstd (dz - sums[0] inv_n) - (act^-1(z) - bias) rstd sums[1] inv_n  (sums global over ranks). */
int mgn_abn_apply_pool(const void* x, void* z, const float* scale, const float* offset, int act, float slope, int N, long HW, int C,
                       float pool_scale, float* pooled, float* workspace, size_t workspace_bytes, void* stream);
int mgn_att_abn_bwd_stats(const void* g, const void* z, const float* weight, const float* bias, float eps, int act, float slope, int N, long HW,
                          int C, float* S, float* workspace, size_t workspace_bytes, void* stream);
int mgn_att_abn_bwd_sums(const float* S, const float* s, const float* dpool, int mode, int N, int C, const float* weight, float* sums, float* dwb,
                         void* stream);
int mgn_att_abn_bwd_apply(const void* g, const void* z, void* dy, const float* s, const float* dpool, int mode, const float* weight,
                          const float* bias, const float* rstd, const float* sums, float inv_n, float eps, int act, float slope, int N, long HW,
                          int C, void* stream);
int mgn_nearest_fwd(const void* x, int N, int h, int w, int H, int W, int C, void* y, void* stream);
int mgn_nearest_bwd(const void* dy, int N, int h, int w, int H, int W, int C, void* dx, void* stream);
int mgn_concat2(const void* a, const void* b, long rows, int Ca, int Cb, void* y, void* stream);
int mgn_split2(const void* dy, long rows, int Ca, int Cb, void* da, void* db, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Stand-alone geometry stages (mgnet/geometry public functions used outside the fused loss) -- csrc/geometry.hip
 * Every function takes ONE affine map per image, A [B,9] row-major and t [B,3] (device fp32), formed by the caller
 * from intrinsics and poses (B x 12 numbers).  Backward entry points also write `partials` [rows,12] fp32 with
 * rows = mgn_geometry_partial_rows(B,H,W): rows of image b are contiguous (rows/B each); their sum is
 * (dL/dA [9], dL/dt [3]) of that image.  padding_mode: 0 = "zeros" only (others MGN_ENOTSUP).
 *
 * mgn_view_synthesis_*  replaces mgnet/geometry/camera_utils.py:24-55 view_synthesis:
 *     out[b,c,v,u] = grid_sample(ref[b,c], (X/Z, Y/Z)),  (X,Y,z) = depth[b,v,u] * (A_b.[u,v,1]) + t_b, Z = clamp(z,1e-5)
 *     with A = K_ref.R.Kinv_cam, t = K_ref.trans, (R,trans) = ref_cam.Tcw o cam.Twc  (camera.py:107-182)
 *     bwd: d_depth [B,1,H,W] and the partials of (dA, dt); ref carries no gradient (as in loss.py: the images are data)
 * mgn_reconstruct_*     replaces camera.py:107-141 Camera.reconstruct (+ pose.py:77-83 for frame "w"):
 *     points[b,k,v,u] = depth * (A_k.[u,v,1]) + t_k,  A = R_wc.Kinv, t = trans_wc
 * mgn_project_*         replaces camera.py:143-182 Camera.project (+ pose.py:77-83):
 *     coords[b,v,u,:] = (2 (X/Z)/(W-1) - 1, 2 (Y/Z)/(H-1) - 1), (X,Y,z) = A.P + t,  A = K.R_cw, t = K.trans_cw
 * ---------------------------------------------------------------------------------------------- */
int mgn_geometry_partial_rows(int B, int H, int W, size_t* rows);
int mgn_view_synthesis_fwd(const float* ref, const float* depth, const float* A, const float* t, int B, int C, int H, int W,
                           int padding_mode, float* out, void* stream);
int mgn_view_synthesis_bwd(const float* ref, const float* depth, const float* A, const float* t, const float* g_out, int B,
                           int C, int H, int W, int padding_mode, float* d_depth, float* partials, void* stream);
int mgn_reconstruct_fwd(const float* depth, const float* A, const float* t, int B, int H, int W, float* points, void* stream);
int mgn_reconstruct_bwd(const float* depth, const float* A, const float* t, const float* g_points, int B, int H, int W,
                        float* d_depth, float* partials, void* stream);
int mgn_project_fwd(const float* points, const float* A, const float* t, int B, int H, int W, float* coords, void* stream);
int mgn_project_bwd(const float* points, const float* A, const float* t, const float* g_coords, int B, int H, int W,
                    float* d_points, float* partials, void* stream);

/* ------------------------------------------------------------------------------------------------
 * IEEE fp16 activations -- the reference's AMP format (configs/MGNet-*.yaml SOLVER.AMP.ENABLED -> torch.cuda.amp: fp16 +
 * GradScaler, tools/train_net.py:162).  Every entry point above that reads or writes 16-bit activations has a twin with the
 * suffix _f16, identical in signature and semantics, in which "bf16" reads "IEEE fp16" (weight layouts included:
 * v_mfma_f32_32x32x16_f16 instead of ..._bf16, round-to-nearest-even conversions, overflow -> inf).  For the mgn_iabn_*
 * functions the dtype code 1 then means fp16.  Built from the same sources (csrc/x_f16.hip = csrc/x.hip with csrc/h16.h
 * switched to fp16).
 * ---------------------------------------------------------------------------------------------- */
int mgn_weight_layout_f16(const float* w_oihw, void* out_h16, int Cout, int Cin, int KH, int KW, int mode, int Cp,
    int cout_pad /* > Cout: output channels zero-padded to cout_pad in the layout (few-class predictors) */, void*
    stream);
int mgn_weight_layout_batch_f16(const void* table_dev, int n_entries, long total_blocks, void* stream);
int mgn_conv_igemm_f16(const void* in, const void* w, void* out, const float* bias, int N, int IH, int IW, int Cin,
    int OH, int OW, int Cout, int KH, int KW, int stride, int pad, int up, int relu, int out_f32, const void*
    residual /* bf16 [N,OH,OW,Cout] added before rounding (the gradient of a second branch of the same tensor, e.g.
    the ResNet shortcut), or NULL */, void* stream);
int mgn_conv_wgrad_f16(const void* dout, const void* in, float* dw, int N, int IH, int IW, int Cin, int OH, int OW,
    int Cout, int KH, int KW, int stride, int pad, int oihw_cin /* >0: dw is [Cout][oihw_cin][KH][KW] */, void*
    workspace, size_t workspace_bytes, void* stream);
int mgn_conv_igemm_stats_f16(const void* in, const void* w, void* out, int N, int IH, int IW, int Cin, int OH, int OW, int Cout, int KH,
    int KW, int stride, int pad, float* stat_partials, const float* stat_shift, void* stream);
int mgn_conv3x3_win_f16(const void* in, const void* w, void* out, int N, int H, int W, int Cin, int Cout, const void* residual,
    int patch_rows, float* stat_partials, const float* stat_shift, void* stream);
int mgn_conv3x3_up2_win_f16(const void* in, const void* w, void* out, int N, int H, int W, int Cin, int Cout, int OH, int OW,
    int ksize, const void* residual, int residual_lowres, void* stream);
int mgn_conv_wgrad_partial_f16(const void* dout, const void* in, int N, int IH, int IW, int Cin, int OH, int OW, int Cout, int KH, int KW,
    int stride, int pad, int oihw_cin, void* workspace, size_t workspace_bytes, long long* desc8, void* stream);
int mgn_conv3x3_win_act_f16(const void* in, const void* w, void* out, int N, int H, int W, int Cin, int Cout, const void* residual,
                            int patch_rows, float* stat_partials, const float* stat_shift, const float* bias, int act, float slope, void* stream);
int mgn_conv_igemm_act_f16(const void* in, const void* w, void* out, const float* bias, int N, int IH, int IW, int Cin, int OH, int OW,
                           int Cout, int KH, int KW, int stride, int pad, int act, float slope, const void* residual, void* stream);
int mgn_conv1x1_cat_f16(const void* in0, const void* in1, const void* w, void* out, int N, int H, int W, int Cin, int Cout, void* stream);
int mgn_conv1x1_split_f16(const void* in, const void* w, void* out0, void* out1, int N, int H, int W, int Cin, int Cout, void* stream);
int mgn_conv_wgrad_cat_f16(const void* dout, const void* in0, const void* in1, float* dw, int N, int H, int W, int Cin, int Cout,
                           void* workspace, size_t workspace_bytes, long long* desc8, void* stream);
int mgn_abn_apply_pool_f16(const void* x, void* z, const float* scale, const float* offset, int act, float slope, int N, long HW, int C,
                           float pool_scale, float* pooled, float* workspace, size_t workspace_bytes, void* stream);
int mgn_att_abn_bwd_stats_f16(const void* g, const void* z, const float* weight, const float* bias, float eps, int act, float slope, int N, long HW,
                              int C, float* S, float* workspace, size_t workspace_bytes, void* stream);
int mgn_att_abn_bwd_apply_f16(const void* g, const void* z, void* dy, const float* s, const float* dpool, int mode, const float* weight,
                              const float* bias, const float* rstd, const float* sums, float inv_n, float eps, int act, float slope, int N, long HW,
                              int C, void* stream);
int mgn_add_relu_fwd_f16(const void* a, const void* b, void* y, long n_elems, void* stream);
int mgn_sum3_f16(const void* a, const void* b, const void* c, void* y, long n_elems, void* stream);
int mgn_abn_add_relu_fwd_f16(const void* x, const float* scale, const float* offset, const void* shortcut, void* y,
    void* relu_bits /*nullable*/, long M, int C, void* stream);
int mgn_relu_mask_bwd_f16(const void* dy, const void* y, void* dx, long n_elems, void* stream);
int mgn_colsum_f16(const void* x, const void* x2 /*nullable*/, int N, long HW, int C, float scale, float* out,
    float* workspace, size_t workspace_bytes /* >= N*64*C*4 */, void* stream);
int mgn_bcast_rows_f16(const float* g, int N, long HW, int C, float scale, void* dx, void* stream);
int mgn_scale_channels_f16(const void* x, const float* s, int N, long HW, int C, int mode, const float* add /*
    nullable: y += add[n,c] (the pooled branch of the attention backward) */, const void* addt /* nullable */, void* y, void* stream);
int mgn_nearest_fwd_f16(const void* x, int N, int h, int w, int H, int W, int C, void* y, void* stream);
int mgn_nearest_bwd_f16(const void* dy, int N, int h, int w, int H, int W, int C, void* dx, void* stream);
int mgn_abn_maxpool_fwd_f16(const void* x_h16, const float* scale, const float* offset, int activation, float slope,
    void* y_h16, uint8_t* argmax, int N, int IH, int IW, int C, void* stream);
int mgn_abn_maxpool_bwd_f16(const void* x_h16, const void* dpool_h16, const uint8_t* argmax, void* dx_h16, const
    float* scale, const float* offset, const float* weight, const float* bias, const float* rstd, const float* sums,
    float total_count, float eps, int activation, float slope, int N, int IH, int IW, int C, void* stream);
int mgn_maxpool3x3s2_fwd_f16(const void* x_h16, void* y_h16, uint8_t* argmax, int N, int IH, int IW, int C, void*
    stream);
int mgn_maxpool3x3s2_bwd_f16(const void* dy_h16, const uint8_t* argmax, void* dx_h16, int N, int IH, int IW, int C,
    void* stream);
int mgn_upce_fwd_f16(const void* logits_h16, long sb, long sh, long sw, int B, int h, int w, int H, int W, int K,
    const long* labels, const float* weights, int ignore, float thr, float* ce_map, float* partials, float* sums3,
    void* stream);
int mgn_upce_bwd_f16(const void* logits_h16, long sb, long sh, long sw, int B, int h, int w, int H, int W, int K,
    int Kp, const long* labels, const float* weights, int ignore, const float* ce_map, const float* sel3, const
    float* gout, float* dlogits, float* footprints, void* stream);
int mgn_ins_loss_fwd_f16(const float* center_lr, long csb, long csh, long csw, const void* offset_lr_h16, long osb,
    long osh, long osw, int B, int h, int w, int H, int W, const float* ct, const float* cw, const float* ot, const
    float* ow, float oscale, float* partials, float* out4, void* stream);
int mgn_ins_loss_bwd_f16(const float* center_lr, long csb, long csh, long csw, const void* offset_lr_h16, long osb,
    long osh, long osw, int B, int h, int w, int H, int W, const float* ct, const float* cw, const float* ot, const
    float* ow, float oscale, const float* out4, const float* gout2, float* dco, float* footprints, void* stream);
int mgn_prep_input_f16(const void* const* frames_u8, int n_frames, int B, int H, int W, const float* pixel_mean3,
    const float* pixel_std3, void* out_h16, int Cp, void* stream);
int mgn_iabn_stats_f16(const void* x, int dtype, long M, int C, float* stats /*[3][C]: count, mean, M2*/, void*
    workspace, size_t workspace_bytes, void* stream);
int mgn_iabn_train_coeffs_f16(const void* x, int dtype, long M, int C, const float* weight, const float* bias, float
    eps, float momentum, float* running_mean, float* running_var, float* coef /*[4][C]*/, void* workspace, size_t
    workspace_bytes, void* stream);
int mgn_iabn_apply_f16(const void* x, void* y /*may alias x*/, int dtype, long M, int C, const float* scale, const
    float* offset, int activation, float slope, void* stream);
int mgn_iabn_bwd_reduce_f16(const void* y, const void* dy, int dtype, long M, int C, const float* weight, const
    float* bias, float eps, int activation, float slope, float* sums /*[2][C]*/, float* dwb /*nullable [2][C]:
    d_weight, d_bias of this rank*/, void* workspace, size_t workspace_bytes, void* stream);
int mgn_iabn_bwd_reduce_x_f16(const void* x, const void* dy, int dtype, long M, int C, const float* weight, const
    float* bias, const float* scale, const float* offset, float eps, int activation, float slope, float* sums,
    float* dwb, void* ws, size_t ws_bytes, void* stream);
int mgn_iabn_bwd_reduce_x_relu_f16(const void* x, const void* g, const void* yrelu, const void* relu_bits, void* dm, long M, int C,
    const float* weight, const float* bias, const float* scale, const float* offset, float eps, float* sums, float* dwb, void* ws,
    size_t ws_bytes, void* stream);
int mgn_iabn_bwd_apply_f16(const void* y, const void* dy, void* dx /*may alias dy*/, int dtype, long M, int C, const
    float* weight, const float* bias, const float* saved, const float* sums, float total_count, float eps, int
    activation, float slope, void* stream);
int mgn_iabn_bwd_apply_x_f16(const void* x, const void* dy, void* dx, int dtype, long M, int C, const float* weight,
    const float* bias, const float* scale, const float* offset, const float* saved, const float* sums, float
    total_count, float eps, int activation, float slope, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MGNET_HIP_H */
