/* nerf_mi355x.h -- C ABI of libnerf_mi355x.so: the NeRF volume-rendering hot path on MI355X (gfx950).
 *
 * The reference (rkin100g/Nerf-Replication) is pure Python on PyTorch ATen ops: there is no native
 * interface on this path to mirror.  The entry points below are therefore
 * what a reference-side binding (ctypes) would call from inside the reference's
 * own plugin surface:
 *     src/models/nerf/renderer/volume_renderer.py:290-432   Renderer.render
 *     src/models/nerf/network.py:199-258                    Network.forward
 * Each function cites the reference lines whose arithmetic it replaces.
 *
 * Conventions: every pointer is a DEVICE pointer to contiguous row-major fp32 unless stated
 * otherwise; inputs are borrowed, outputs are caller-allocated; `stream` is a hipStream_t (NULL =
 * default stream); calls only enqueue work (no host sync, graph-capturable); the return value is 0
 * on success or a negative nerf_status, never an exception; nerf_last_error() gives the message of
 * the calling thread's last failure.  n_rays == 0 is a successful no-op everywhere.
 */
#ifndef NERF_MI355X_H
#define NERF_MI355X_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Version 2 (round 3).  Contracts that changed since version 1 (entry names and signatures did not):
 *   - the TrainSave / TrainGrad buffers (nerf_train_save_floats / nerf_train_grad_floats) hold pad32(P) rows per region, with the
 *     ReLU sign-bit blocks, the live-tile flags / list / count and a "rows skipped" stamp appended;
 *     nerf_mlp_backward (both precisions) takes its ReLU masks from those sign bits (written by the SAVE forwards only);
 *   - both precisions: the density entries skip the colour branch (its rows and sign bits are not stored / not read), and lanes
 *     past the last point write the pad32 padding rows of every region;
 *   - with dead-tile skipping (default; NERF_DEAD_TILE_SKIP=0 in the environment disables it) the g_z rows of tiles whose incoming
 *     gradient is zero throughout are left unwritten in `gsave` (point mode zero-fills the g_zv region, which
 *     nerf_viewdirs_backward sums over);
 *   - nerf_mlp_forward_rays_save_for_compositing may omit the rows of density-free tiles; it stamps the buffer, and a backward pass
 *     that runs without the live-tile list on such a buffer poisons grads[alpha_linear.bias] with NaN instead of reading them;
 *   - nerf_composite_backward refuses more than 192 samples per ray. */
#define NERF_ABI_VERSION 2

enum nerf_status {
  NERF_OK = 0,
  NERF_ERR_INVALID_ARG = -1,   /* null pointer, negative size, unsupported sample counts */
  NERF_ERR_WORKSPACE = -2,     /* workspace too small */
  NERF_ERR_HIP = -3,           /* a HIP runtime call or kernel launch failed */
  NERF_ERR_UNSUPPORTED = -4    /* precision / mode not built */
};

enum nerf_precision {
  NERF_PREC_F32 = 0,           /* exact fp32: v_mfma_f32_32x32x2_f32 (the parity path) */
  NERF_PREC_F16 = 1,           /* fp16 activations+weights, fp32 accumulate: v_mfma_f32_32x32x16_f16
                                  (BASELINE config 5; PSNR-level tolerance, not the parity path) */
  NERF_PREC_F32X = 2,          /* fp32-accurate on the fp16 matrix cores: every operand split into hi + 2^-11 lo
                                  fp16 parts, 3 MFMAs per product, fp32 accumulate (~2^-22 operand error) */
  NERF_PREC_F16S = 3           /* NERF_PREC_F16's arithmetic on the other MFMA shape, v_mfma_f32_16x16x32_f16 (same operands and
                                  accumulation width; only the order of the fp32 partial sums differs); its own packed layout */
};

/* Renderer constants the reference effectively hard-codes (volume_renderer.py:14-24). */
#define NERF_N_SAMPLES 64
#define NERF_N_IMPORTANCE 128

int32_t nerf_abi_version(void);

/* Build self-description: 0 for a product build.  Timing experiments (tools/ab_bench.py) compile kernels with
   switches that change their numerics; such a library only builds with -DNERF_TIMING_BUILD and reports it here.
   A binder must refuse a non-zero value (nerf_replication_amd/_lib.py does). */
enum nerf_build_flag { NERF_BUILD_TIMING = 1, NERF_BUILD_WRONG_NUMERICS = 2 };
int32_t nerf_build_flags(void);
const char* nerf_last_error(void);

/* Size in bytes of one packed sub-model (coarse or fine) for a precision; -1 if unknown. */
int64_t nerf_packed_model_bytes(int32_t precision);

/* Permute the 24 parameter tensors of one NeRF sub-model into the kernel's weight stream
 * (csrc/nerf_layout.h).  `params` is a HOST array of 24 DEVICE pointers in the reference's
 * state_dict order (network.py:22-47): pts_linears.0..7 {weight,bias}, views_linears.0,
 * feature_linear, alpha_linear, rgb_linear; weights are nn.Linear [out,in] row-major.
 * Runs on the device; call again whenever the parameters change (e.g. after an optimizer step). */
int32_t nerf_pack_model(const float* const params[24], void* packed, int32_t precision, void* stream);

/* Positional encoding alone: x [n,3] -> out [n, 3+6*n_freqs] in the reference's channel order.
 * Replaces freq.py:31-32 (Encoder.embed); n_freqs is 10 (xyz) or 4 (view dir). Uses the same
 * device sincos as the fused MLP; exposed for parity tests. */
int32_t nerf_positional_encoding(const float* x, int64_t n, int32_t n_freqs, float* out, void* stream);

/* Network.forward(inputs[n,s,3], viewdirs[n,3], valid_mask=None, model) -> raw [n,s,4] = (r,g,b,sigma)
 * pre-activation.  Replaces network.py:216-256: PE of points and directions, the batchify(512) loop
 * over NeRF.forward (network.py:49-74), and the reshape.  `packed` selects coarse or fine model. */
int32_t nerf_mlp_forward(const float* pts, const float* viewdirs, int64_t n_rays, int32_t n_samples,
                         const void* packed, float* raw, int32_t precision, void* stream);

/* Same network on points generated on the fly: pts = rays_o + rays_d * t (volume_renderer.py:63,
 * :267), viewdirs = rays_d / ||rays_d|| (:314).  t of (ray i, sample s) is
 * tvals[i*t_ray_stride + s]; t_ray_stride = 0 shares one table (the deterministic coarse linspace). */
int32_t nerf_mlp_forward_rays(const float* rays_o, const float* rays_d, const float* tvals,
                              int64_t t_ray_stride, int64_t n_rays, int32_t n_samples,
                              const void* packed, float* raw, int32_t precision, void* stream);

/* Density-only form of nerf_mlp_forward_rays: raw[..., 3] = sigma, bit for bit the value nerf_mlp_forward_rays writes there.
 * In a hierarchical render (N_importance > 0) the reference reads nothing else of the coarse network's output
 * (volume_renderer.py:335 `density_coarse = outputs[...,3]`; rgb and depth are composited from the fine outputs, :414-437), so
 * nerf_render_forward runs its coarse pass through this entry.  The network stops after the sigma head (feature_linear,
 * views_linears.0 and rgb_linear are not evaluated: 982 528 instead of 1 186 816 FLOP per point) and the rgb columns of `raw`
 * are written as 0, in every precision. */
int32_t nerf_mlp_forward_rays_density(const float* rays_o, const float* rays_d, const float* tvals,
                                      int64_t t_ray_stride, int64_t n_rays, int32_t n_samples,
                                      const void* packed, float* raw, int32_t precision, void* stream);

/* nerf_mlp_forward_rays for outputs that go to nerf_composite and nowhere else (the fine pass of nerf_render_forward): the
 * sigma column is bit for bit nerf_mlp_forward_rays'; the rgb columns are too, EXCEPT that they may be 0 for points whose
 * sigma is <= 0 -- compositing gives those samples alpha = 1 - exp(-relu(sigma) delta) = 0, hence weight 0, and multiplies their
 * colour by exactly zero (volume_renderer.py:67-96, :414-437), so rgb and depth out of nerf_composite are bit-identical either
 * way.  A 32-point tile (32 consecutive samples) without a single sigma > 0 skips the colour branch (17 % of its FLOP), in
 * every precision; how many tiles do is scene-dependent. */
int32_t nerf_mlp_forward_rays_for_compositing(const float* rays_o, const float* rays_d, const float* tvals,
                                              int64_t t_ray_stride, int64_t n_rays, int32_t n_samples,
                                              const void* packed, float* raw, int32_t precision, void* stream);

/* ---- training path (BASELINE config 3; fp32 only) ------------------------------------------------
 * Forward with activation save: as nerf_mlp_forward_rays, and additionally stores, row-major per
 * point (P = n_rays*n_samples; floats): pe [P,64] and dpe [P,32] (encodings in the reference's channel
 * order, zero padded), h0..h7 [P,256] each (post-ReLU, network.py:55-56), feature [P,256] (:62),
 * views [P,128] (post-ReLU, :66-67) -- in that order, nerf_train_save_floats(P) floats in total.
 * These are the tensors autograd would keep for network.py:49-74.  precision: NERF_PREC_F32 or NERF_PREC_F32X
 * (`packed` must be the stream of that precision); the backward kernels are fp32 either way. */
int64_t nerf_train_save_floats(int64_t n_points);
int32_t nerf_mlp_forward_rays_save(const float* rays_o, const float* rays_d, const float* tvals,
                                   int64_t t_ray_stride, int64_t n_rays, int32_t n_samples,
                                   const void* packed, float* raw, float* save, int32_t precision, void* stream);

/* Backward of the MLP (network.py:49-74 under autograd) for the points of nerf_mlp_forward_rays_save.
 * `packed_bwd` is the transposed weight stream of nerf_pack_model_bwd (nerf_packed_bwd_bytes(precision) bytes;
 * precision NERF_PREC_F32: exact fp32 MFMA chain, NERF_PREC_F32X: split-fp16 chain; weight gradients are fp32 MFMA);
 * `draw` [P,4] is d loss / d raw; `save` the forward's activation store; `gsave` scratch of
 * nerf_train_grad_floats(P) floats (receives every layer's pre-activation gradient).  Adds the 24
 * parameter gradients (state_dict order, nn.Linear layouts; the caller zeroes them) and, if `g_t` [P]
 * is given, writes d loss / d t through the points (x = o + d t, positional encoding included) -- the
 * path by which the coarse network is trained.
 * The ReLU masks come from the sign-bit blocks the SAVE forwards append behind the activation rows (nerf_train_save_floats
 * covers them; both precisions write the same layout, so a `save` of either forward is accepted as long as it was written by the
 * matching entry: full / for-compositing / density).  Both `save` and `gsave` regions hold their rows padded to a
 * multiple of 32 points (offsets are derived from the padded count, see csrc/nerf_mlp_f32.hip.inc TrainSave). */
int64_t nerf_train_grad_floats(int64_t n_points);
/* Dead-tile skipping (n_points a multiple of 32; exact): a tile of 32 consecutive points whose `draw` rows are
 * all zero has zero g_z rows, adds nothing to any parameter gradient and has zero g_t / g_x -- compositing writes such rows
 * wherever relu(sigma) = 0.  nerf_mlp_backward* drop those tiles from the chain launch and from every weight-gradient launch
 * (their rows in `gsave` are then left unwritten).  The number of live tiles of the call is left as an int32 at float offset
 * nerf_train_live_count_offset(n_points) of `gsave` (-1: the call ran without a list: ragged n_points, or
 * NERF_DEAD_TILE_SKIP=0 in the environment, which turns the skipping off for A/B comparisons). */
int64_t nerf_train_live_count_offset(int64_t n_points);
int64_t nerf_packed_bwd_bytes(int32_t precision);
int32_t nerf_pack_model_bwd(const float* const params[24], void* packed_bwd, int32_t precision, void* stream);
int32_t nerf_mlp_backward(const float* rays_o, const float* rays_d, const float* tvals, int64_t t_ray_stride,
                          int64_t n_rays, int32_t n_samples, const void* packed_bwd, const float* draw,
                          const float* save, float* gsave, float* g_t, float* const grads[24], int32_t precision,
                          void* stream);

/* nerf_mlp_forward_rays_save for a fine pass whose `raw` goes to nerf_composite and whose `draw` will come from
 * nerf_composite_backward (training.RenderFunction): a 32-point tile without a single sigma > 0 stops after the sigma head
 * (rgb = 0 there, as nerf_mlp_forward_rays_for_compositing) and stores nothing past its h6 row (NERF_PREC_F32) / its h7 row
 * (NERF_PREC_F32X: no feature / views rows, no views bits).  CONTRACT: the `draw`
 * later given to nerf_mlp_backward with this `save` must be zero wherever sigma <= 0 -- nerf_composite_backward guarantees it --
 * so that the backward pass, which skips tiles with a zero incoming gradient, never reads those rows.  With
 * NERF_DEAD_TILE_SKIP=0 in the environment (or a point count that is not a multiple of 32) this is nerf_mlp_forward_rays_save. */
int32_t nerf_mlp_forward_rays_save_for_compositing(const float* rays_o, const float* rays_d, const float* tvals,
                                                   int64_t t_ray_stride, int64_t n_rays, int32_t n_samples,
                                                   const void* packed, float* raw, float* save, int32_t precision, void* stream);

/* Density-only twins for the COARSE pass of a training step.  In the reference's step only the coarse sigma is ever used
 * (it places the fine samples; the coarse colour is never composited, volume_renderer.py:385-397), so
 * d loss / d raw_coarse has identically zero rgb columns and the gradients of rgb_linear, views_linears.0 and
 * feature_linear of the coarse sub-model are exactly zero.  These entries skip that branch (both precisions): the forward
 * stops after the sigma head (raw = (0, 0, 0, sigma); feature / views rows are not stored), the backward starts at
 * g_h7 = w_alpha * g_sigma and leaves the three colour gradients as zeroed by the caller.  `draw`'s rgb columns are not
 * read.  `save` from the density forward must go to the density backward. */
int32_t nerf_mlp_forward_rays_save_density(const float* rays_o, const float* rays_d, const float* tvals,
                                           int64_t t_ray_stride, int64_t n_rays, int32_t n_samples,
                                           const void* packed, float* raw, float* save, int32_t precision, void* stream);
int32_t nerf_mlp_backward_density(const float* rays_o, const float* rays_d, const float* tvals, int64_t t_ray_stride,
                                  int64_t n_rays, int32_t n_samples, const void* packed_bwd, const float* draw,
                                  const float* save, float* gsave, float* g_t, float* const grads[24], int32_t precision,
                                  void* stream);

/* Point-mode twins of the two calls above, for Network.forward itself under autograd (network.py:199-258: explicit
 * `inputs` [n_rays, n_samples, 3] and `viewdirs` [n_rays, 3] used AS GIVEN, no normalisation -- not o + d t):
 * nerf_mlp_forward_points_save = nerf_mlp_forward + the activation store; nerf_mlp_backward_points adds the 24
 * parameter gradients and, if `g_pts` [P,3] is given, writes d loss / d inputs (through the positional encoding).
 * d loss / d viewdirs comes from nerf_viewdirs_backward below (a separate, tiny launch: no caller of the reference needs it). */
int32_t nerf_mlp_forward_points_save(const float* pts, const float* viewdirs, int64_t n_rays, int32_t n_samples,
                                     const void* packed, float* raw, float* save, int32_t precision, void* stream);
int32_t nerf_mlp_backward_points(const float* pts, int64_t n_rays, int32_t n_samples, const void* packed_bwd,
                                 const float* draw, const float* save, float* gsave, float* g_pts,
                                 float* const grads[24], int32_t precision, void* stream);
/* d loss / d viewdirs [n_rays,3] of the Network.forward call whose backward just filled `gsave` (either chain): the ray's
 * direction reaches views_linears.0 through its 27-channel encoding, shared by the ray's samples.  `w_views` is the raw
 * views_linears.0.weight [128,283] (nn.Linear layout), `viewdirs` the forward's [n_rays,3]. */
int32_t nerf_viewdirs_backward(const float* gsave, int64_t n_rays, int32_t n_samples, const float* w_views,
                               const float* viewdirs, float* g_viewdirs, void* stream);

/* Adjoint of nerf_composite (autograd of volume_renderer.py:414-432 with :67-96): g_rgb [n,3], g_depth [n]
 * (nullable) -> g_raw [n,S,4] and, if given, g_t [n,S] (the direct dependence of the image on the sample
 * depths through delta_k = t_{k+1}-t_k and the depth sum).  S <= 192.  Rows of g_raw are exactly zero wherever
 * sigma <= 0 (relu: alpha = 0, weight 0) -- what the dead-tile skipping of nerf_mlp_backward* feeds on. */
int32_t nerf_composite_backward(const float* raw, const float* tvals, int64_t t_ray_stride, int64_t n_rays,
                                int32_t n_samples, int32_t white_bkgd, const float* g_rgb, const float* g_depth,
                                float* g_raw, float* g_t, void* stream);

/* Adjoint of nerf_sample_fine (autograd of volume_renderer.py:126-154, :247-264, :349-356): gradient of the
 * merged depths g_t_sorted [n,192] -> g_raw_coarse [n,64,4] (channel 3 only; the reference does NOT detach
 * the coarse weights, so the coarse network trains through the sample positions). */
int32_t nerf_sample_fine_backward(const float* raw_coarse, const float* t_coarse, const float* u, int64_t n_rays,
                                  const float* t_sorted, const float* g_t_sorted, float* g_raw_coarse, void* stream);

/* Fused gradient clipping + Adam over up to 48 tensors in one launch.  Replaces
 * clip_grad_value_(parameters, 40) (src/train/trainers/trainer.py:59) followed by torch.optim.Adam.step()
 * as configured in src/train/optimizer.py:21-24 (betas (0.9, 0.999), no amsgrad; weight decay added to the
 * gradient).  The pointer arrays are HOST arrays of DEVICE pointers; `step` is the 1-based step count
 * (bias corrections are computed on the host in double); clip_value <= 0 disables clipping. */
int32_t nerf_adam_step(int32_t n_tensors, float* const params[], const float* const grads[], float* const exp_avg[],
                       float* const exp_avg_sq[], const int64_t numel[], float lr, float beta1, float beta2, float eps,
                       float weight_decay, float clip_value, int64_t step, void* stream);

/* Weight / bias gradient of one nn.Linear (or a column block of it): for o < n_out, i < n_in
 *     dw[o*ldw + wc0 + i] += sum_p dz[p*ldz + zc0 + o] * hin[p*ldh + hc0 + i],   db[o] += sum_p dz[p*ldz + zc0 + o]
 * i.e. autograd's grad_weight = grad_out^T @ input, grad_bias = grad_out.sum(0) for network.py:22-47;
 * `dw` (and `db`, optional) accumulate with float atomics and must be zeroed by the caller; dw is in the
 * nn.Linear [out, in] layout (ldw = in_features), so concatenated inputs (skip, view) are two calls with
 * different wc0.  n_out, n_in <= 256. */
int32_t nerf_wgrad(const float* dz, int64_t ldz, int32_t zc0, int32_t n_out, const float* hin, int64_t ldh,
                   int32_t hc0, int32_t n_in, float* dw, int64_t ldw, int32_t wc0, float* db, int64_t n_points,
                   void* stream);

/* Hierarchical sampling + merge.  raw_coarse [n,64,4] (sigma = channel 3, pre-ReLU), t_coarse [64],
 * u [128] -> t_sorted [n,192] (ascending union of coarse and fine depths), optional t_fine [n,128].
 * Replaces ReLU of the coarse density (volume_renderer.py:335-338), weights_computation (:67-96),
 * fine_sample_points' deterministic branch (:126-154, :247-264) incl. its index clamp to 61, and the
 * cat + torch.sort of depths (:349-353); the sorted POINTS (:354-356) are regenerated as o + d*t.
 * If `valid_sorted` [n,192] (uint8) is non-NULL the fast_sampling branch is evaluated too: ESS (coarse
 * weight < weights_threshold around the sample, strict for rays with max sigma > 0.5) and ERT (coarse
 * transmittance < ert_threshold) masks and the empty-ray test (:116-123, :132-133, :158-193), merged
 * through the sort with the always-valid coarse samples (:359-369). */
int32_t nerf_sample_fine(const float* raw_coarse, const float* t_coarse, const float* u,
                         int64_t n_rays, float* t_sorted, float* t_fine, uint8_t* valid_sorted,
                         float weights_threshold, float ert_threshold, void* stream);

/* Final activations + alpha compositing.  raw [n,S,4], t per (ray,sample) as above ->
 * rgb [n,3], depth [n], optional weights [n,S].  Replaces volume_renderer.py:414-432:
 * sigmoid(rgb), relu(sigma), weights_computation, the two sums and the white background. */
int32_t nerf_composite(const float* raw, const float* tvals, int64_t t_ray_stride, int64_t n_rays,
                       int32_t n_samples, int32_t white_bkgd, float* rgb, float* depth,
                       float* weights, void* stream);

/* Pinhole ray generation on the device.  Replaces src/datasets/nerf/blender.py:102-127
 * (float64 math, float32 result, unit directions): pixel id -> u = id % W, v = id / W,
 * dirs = [(u-W/2)/f, -(v-H/2)/f, -1], rays_d = normalize(R dirs), rays_o = t.  `c2w` is a HOST array,
 * row-major 3x4.  Pixels are pixel_begin .. pixel_begin+n_pixels-1 (row-major image order, a rank's
 * tile) or, if `pixel_ids` (DEVICE int64 [n_pixels]) is given, that list (a training batch). */
int32_t nerf_generate_rays(const double c2w[12], int32_t H, int32_t W, double focal, int64_t pixel_begin,
                           int64_t n_pixels, const int64_t* pixel_ids, float* rays_o, float* rays_d, void* stream);

/* Evaluator sums, src/evaluators/nerf.py: sums2[0] = sum over all values of
 * (clip(pred,0,1) - clip(gt,0,1))^2 (:96-100), sums2[1] = sum of the evaluator's psnr_metric
 * integrand (:23-30) whose uint8 subtraction and squaring wrap mod 256.  `sums2` is a
 * DEVICE double[2], zeroed by the call; mean = sum / n_values, PSNR = 10 log10(peak^2 / mean). */
int32_t nerf_image_metrics(const float* pred, const float* gt, int64_t n_values, double* sums2, void* stream);

/* SSIM sum of src/evaluators/nerf.py:49-77: skimage.metrics.structural_similarity on the uint8 images
 * (win_size 7, channel_axis 2: uniform 7x7 window, sample covariance, K1 .01, K2 .03, data_range 255).
 * pred, gt: [H,W,3] float in [0,1] (clipped and truncated to uint8 as the evaluator does); *sum1 (DEVICE
 * double, zeroed by the call) = sum of the SSIM map over the (H-6)x(W-6) interior and 3 channels;
 * SSIM = sum / ((H-6)(W-6)3).  skimage is not vendored/pinned by the reference (requirements.txt): the
 * published algorithm is restated. */
int32_t nerf_image_ssim(const float* pred, const float* gt, int32_t H, int32_t W, double* sum1, void* stream);

/* Bytes of scratch nerf_render_forward needs for n_rays rays. */
int64_t nerf_render_workspace_bytes(int64_t n_rays, int32_t n_importance, int32_t fast_sampling);

/* Renderer.render for already-flattened rays [n,3]: coarse pass, hierarchical sampling, fine pass,
 * compositing (volume_renderer.py:306-432).  n_importance is 0 (coarse only) or 128.  t_coarse [64]
 * and u [128] are the host-built torch.linspace tables (bit-sensitive).
 * fast_sampling != 0 selects the ESS/ERT masked fine pass (off in every reference config):
 * only merged samples that survive the masks go through the fine network (device-side compaction),
 * the rest contribute raw = 0 exactly as network.py:238-253 does.  Outputs rgb [n,3], depth [n]. */
int32_t nerf_render_forward(const float* rays_o, const float* rays_d, int64_t n_rays,
                            const void* packed_coarse, const void* packed_fine,
                            const float* t_coarse, const float* u, int32_t n_importance,
                            int32_t white_bkgd, int32_t precision, int32_t fast_sampling,
                            float weights_threshold, void* workspace,
                            int64_t workspace_bytes, float* rgb, float* depth, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NERF_MI355X_H */
