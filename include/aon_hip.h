/* aon_hip.h -- C ABI of libaon_hip.so, the MI355X (gfx950) NeRF volume-rendering hot path.
 *
 * The reference (zubair-irshad/articulated-object-nerf) is 100 % Python and has no FFI of its own: the boundary
 * it offers is `NeRF.forward(rays, randomized, white_bkgd, near, far)` (models/vanilla_nerf/model.py:147-199)
 * and the free functions of models/vanilla_nerf/helper.py.  Each entry point below replaces one of those
 * functions (cited per declaration); the Python host side in articulated-object-nerf_amd/ binds them with ctypes and
 * re-exposes the reference's names and signatures (INTEGRATION.md shows the binding a maintainer would add).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to contiguous row-major fp32 unless the parameter is marked `host`;
 *   - nothing is allocated, freed or synchronised inside: outputs and workspace are caller-owned, launches are
 *     enqueued on `stream` (a hipStream_t passed as void*; NULL = the default stream) and return immediately;
 *   - inputs are never written;
 *   - return value: 0 = success, negative = failure (AON_E_* below, or -(hipError_t) - 1000 for a HIP error);
 *     aon_last_error() returns a thread-local human-readable message for the last failure.
 */
#ifndef AON_HIP_H
#define AON_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AON_ABI_VERSION 5   /* 5: + aon_adam_step, aon_code_library_fwd / _bwd, aon_art_pack_step, aon_vanilla_pack_step, aon_stream_form, aon_declare_stream_form; a packed pointer this process never
                               packed or declared is refused (AON_E_INVALID / HIP "invalid value") instead of being taken to have the default form */

#define AON_OK 0
#define AON_E_INVALID (-1)    /* null pointer, negative size, unsupported geometry */
#define AON_E_WORKSPACE (-2)  /* workspace too small */
#define AON_E_HIP_BASE (-1000) /* HIP error e is reported as AON_E_HIP_BASE - e */

/* activation applied to the raw network outputs inside aon_composite */
#define AON_ACT_NONE 0     /* inputs already activated: helper.volumetric_rendering as is */
#define AON_ACT_VANILLA 1  /* rgb = sigmoid(raw), sigma = relu(raw)                 model.py:186-187 */
#define AON_ACT_ARTICULATED 2 /* rgb = sigmoid(raw)*1.002-0.001, sigma = softplus(raw-1)  model_autodecoder.py:321-323 */

int aon_abi_version(void);
const char* aon_last_error(void);

/* ---- R1+R2  datasets/ray_utils.py:71-90 (get_ray_directions) + :118-159 (get_rays, output_view_dirs=True) ----
 * Generates the rays of row-major pixels [pix_begin, pix_end) of an HxW pinhole frame: no +0.5 pixel centre,
 * OpenGL camera (looks down -z).  c2w: HOST pointer to 12 floats, row-major (3,4).  Writes rays_o (n,3) and the
 * unit-norm viewdirs (n,3); rays_d (n,3) may be NULL (the reference's rays_d is the same storage as viewdirs). */
int aon_raygen(const float* c2w_host, int H, int W, float focal, int64_t pix_begin, int64_t pix_end,
               float* rays_o, float* viewdirs, float* rays_d, void* stream);

/* ---- R13  the training losses as two launches (round 5) ----
 * helper.py:17-22 (img2mse, mse2psnr), model.py:271-273 (loss0 + loss1), model_autodecoder.py:460-466 (+ reg_scale * the sum of the
 * means of |code| over the latent codes, each ONE row: torch.norm(code, dim=0) of a (1,D) code is |code|).
 * fwd: stats (8 floats, device) = {loss0, loss1, reg, loss, psnr0, psnr1, 0, 0}; loss (1 float, device) = fl(fl(loss1 + loss0) + reg);
 *      fp64 sums in a fixed order, each output rounded once.  rgb_coarse may be NULL (num_levels = 1): loss0 = 0.
 *      latents_host / latent_len_host: HOST arrays of 3 device pointers / lengths, NULL entries (or NULL arrays) = no such code.
 * bwd: d_rgb_* (n,3) = fl(fl(grad_loss / 3n) * fl(2 (rgb - target))), d_latents[k] = fl(x * fl(fl(fl(grad_loss * reg_scale) / len) / |x|)),
 *      0 where x = 0 -- the operations torch autograd runs for the same lines; grad_loss: device scalar. */
int aon_train_loss_fwd(const float* rgb_coarse, const float* rgb_fine, const float* target, int64_t n, const float* const* latents_host,
                       const int* latent_len_host, float reg_scale, float* stats, float* loss, void* stream);
int aon_train_loss_bwd(const float* rgb_coarse, const float* rgb_fine, const float* target, int64_t n, const float* const* latents_host,
                       const int* latent_len_host, float reg_scale, const float* grad_loss, float* d_rgb_coarse, float* d_rgb_fine,
                       float* const* d_latents_host, void* stream);

/* ---- the end of a training step on one flat parameter arena (round 6) ----
 * aon_adam_step: torch.optim.Adam(lr, betas=(beta1, beta2), eps) as configured by the reference (model.py:386-389, model_autodecoder.py:604-606;
 * weight_decay = 0, amsgrad = False) over `n` contiguous fp32 elements in ONE launch: params, grads and both moments are flat buffers of the
 * same layout (the Python side makes every nn.Parameter, its .grad and its Adam state views into them: aon_amd/arena.py).  `step` is the
 * count AFTER this update (torch's state["step"]), lr the value the harness's rule set for this step (model.py:391-419).  The element-wise
 * operations and their order are those of torch's single-tensor implementation (exp_avg.lerp_, exp_avg_sq.mul_().addcmul_(), sqrt / bias
 * correction + eps, addcdiv_), fp32 with one rounding each; the bias corrections are evaluated in double on the host. */
int aon_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, double lr, double beta1, double beta2, double eps,
                  int64_t step, void* stream);

/* CodeLibraryArticulated.forward for the reference's batch of ONE object in ONE state (models/code_library.py:36-53; sapien_multi.py:362-479
 * delivers instance_id / articulation_id of shape (1,)): out[t] (dims[t],) = tables[t][ids[t][0]] for the three tables (shape, appearance,
 * articulation) in one launch; an out-of-range id gives a NaN row (nn.Embedding raises; a kernel cannot).  All *_host arguments are HOST
 * arrays of 3 entries; ids[t]: device pointer to ONE int64.
 * _bwd: the dense table gradients nn.Embedding's autograd produces, g_tables[t] (rows[t], dims[t]) = 0 except row ids[t][0] = g_rows[t]. */
int aon_code_library_fwd(const float* const* tables_host, const int64_t* const* ids_host, const int* rows_host, const int* dims_host, float* const* out_host,
                         void* stream);
int aon_code_library_bwd(const float* const* g_rows_host, const int64_t* const* ids_host, const int* rows_host, const int* dims_host, float* const* g_tables_host,
                         void* stream);

/* Everything a training step of a TWO-level articulated model packs, in one call (round 6): aon_pack_art_mlp_deg + aon_art_prepare_deg +
 * aon_pack_art_mlp_bwd_deg for the coarse and the fine network -- every element computed by the same code on the same operands, the same
 * bytes in all six buffers -- as THREE launches instead of ten: the four fp64 products W' = W_v0[:, :256] W_b (each network's, for its
 * forward and for its transposed stream) as one, both networks' per-call blocks + forward streams as one, both transposed streams as one.
 * packed_bwd_* may be NULL (then one pack launch per network and buffer, as the separate calls).  Buffer sizes: aon_art_packed_bytes /
 * aon_art_small_bytes / aon_art_bwd_packed_bytes; all 16-byte aligned.  The form (folded / literal) is the process default at the time of
 * the call, recorded for all six buffers. */
int aon_art_pack_step(const float* const* params_coarse_host, const float* const* params_fine_host, const float* shape, const float* appearance,
                      const float* articulation, int min_deg_point, int max_deg_point, int deg_view, void* packed_coarse, void* small_coarse,
                      void* packed_bwd_coarse, void* packed_fine, void* small_fine, void* packed_bwd_fine, void* stream);

/* The vanilla counterpart: aon_pack_vanilla_mlp_deg + aon_pack_vanilla_mlp_bwd_deg for the coarse and the fine network (models/vanilla_nerf/
 * model.py:147-199 holds two NeRFMLPs), the same bytes in all four buffers, with the eight fp64 fold products (W', b' of each network, for
 * its forward and for its transposed stream) as ONE launch in front.  packed_bwd_* may be NULL.  Buffer sizes: aon_mlp_packed_bytes /
 * aon_bwd_packed_bytes; 16-byte aligned. */
int aon_vanilla_pack_step(const float* const* params_coarse_host, const float* const* params_fine_host, int min_deg_point, int max_deg_point,
                          int deg_view, void* packed_coarse, void* packed_bwd_coarse, void* packed_fine, void* packed_bwd_fine, void* stream);

/* get_ray_directions alone (ray_utils.py:71-90): directions (H*W,3), un-normalised camera-space. */
int aon_ray_directions(int H, int W, float focal, float* directions, void* stream);

/* get_rays on caller-supplied camera-space directions (ray_utils.py:118-159): directions (n,3) -> rays_o,
 * viewdirs (unit), rays_d (may be NULL). */
int aon_get_rays(const float* directions, const float* c2w_host, int64_t n, float* rays_o, float* viewdirs,
                 float* rays_d, void* stream);

/* The `radii` output of get_rays(..., output_radii=True) (ray_utils.py:138-143), the call form of every reference
 * dataset (sapien.py:102,145; sapien_multi.py:301,343): directions (H,W,3) row-major -> radii (H*W,).  H >= 3. */
int aon_ray_radii(const float* directions, const float* c2w_host, int H, int W, float* radii, void* stream);

/* helper.cast_rays (helper.py:25-26): coords (n,S,3) = origins[:,None,:] + t_vals[...,None] * directions[:,None,:] */
int aon_cast_rays(const float* t_vals, const float* origins, const float* directions, int64_t n_rays, int S,
                  float* coords, void* stream);

/* ---- R3  helper.sample_along_rays (helper.py:106-133, lindisp=False) + helper.cast_rays (:25-26) ----
 * S = num_samples + 1 t-values per ray.  t_rand (n,S) replaces the reference's torch.rand draw when the
 * caller wants randomized=True; NULL = deterministic.  coords (n,S,3) may be NULL. */
int aon_sample_along_rays(const float* rays_o, const float* rays_d, int64_t n_rays, int S, float near_, float far_,
                          const float* t_rand, float* t_vals, float* coords, void* stream);

/* ---- R4  helper.pos_enc (helper.py:136-140), stage-level ----  x (n,3) -> out (n, 3 + 6*(max_deg-min_deg)) */
int aon_pos_enc(const float* x, int64_t n, int min_deg, int max_deg, float* out, void* stream);

/* ---- R5  NeRFMLP weights (models/vanilla_nerf/model.py:39-93) ----
 * params: HOST array of 24 DEVICE pointers to the unmodified nn.Linear storages ((out,in) row-major), order:
 *   pts_linears.{0..7}.{weight,bias}, views_linear.0.{weight,bias}, bottleneck_layer.{weight,bias},
 *   density_layer.{weight,bias}, rgb_layer.{weight,bias}.
 * Re-run whenever the parameters change; `packed` needs aon_mlp_packed_bytes() bytes, 16-byte aligned. */
int64_t aon_mlp_packed_bytes(void);
int aon_pack_vanilla_mlp(const float* const* params_host, void* packed, void* stream);

/* ---- Round 5: bottleneck_layer folded into views_linear[0] ----
 * The reference's bottleneck_layer has NO activation and feeds views_linear[0] directly (models/vanilla_nerf/model.py:109-114,
 * model_autodecoder.py:223-230), so
 *     W_v0[:, :256] (W_b h + b_b) + W_v0[:, 256:] c + b_v0  ==  (W_v0[:, :256] W_b) h + W_v0[:, 256:] c + (W_v0[:, :256] b_b + b_v0)
 * and a from-scratch kernel can run ONE 256 -> 128 layer where the literal graph runs 256 -> 256 then 256 -> 128: 65,536 of the
 * vanilla network's 593,408 multiply-adds per sample (11.0 %; 9.5 % of the articulated network's executed 692,480), in the forward,
 * the backward data chain and the weight gradients alike.  With the switch on (default) every aon_pack_* / aon_art_prepare* call
 * builds the FOLDED form: W' = W_v0[:, :256] W_b and the vanilla b' are evaluated in fp64 from the fp32 parameters and rounded once (the
 * articulated per-call block: the fp64 sum W_v0[:, :256] b_b is added to the fp32 value of b_v0 + the appearance-latent term and the
 * result rounded again -- two roundings, the second half an ulp of the bias); the training
 * forward writes no bottleneck rows; the backward computes dW' = dZ_v0 H7^T, db' and un-folds them exactly as autograd's chain rule
 * does -- dW_b = W_v0[:, :256]^T dW', db_b = W_v0[:, :256]^T db', dW_v0[:, :256] = dW' W_b^T + db' (x) b_b (fp64 accumulation) -- so
 * the 24 / 40 parameter gradients keep the reference's shapes and meaning.  Buffer sizes do not depend on the switch.
 * The form is a property of each packed buffer / per-call block, fixed when it was made and remembered per pointer: flipping the
 * switch later does not change how an existing buffer is run, and a call that is handed buffers of two forms returns AON_E_INVALID
 * (HIP "invalid value").  A buffer this process neither packed nor declared (a device-side copy of a packed buffer) has NO form and every
 * call refuses it the same way (round 6; rounds 5 assumed the current default): its owner states the form of the copy with
 * aon_declare_stream_form(copy, aon_stream_form(original)).  The Python binding keeps the form next to the tensor and re-declares it on every
 * call, so an address the caching allocator hands out again cannot carry the form of an earlier tenant.
 * aon_set_bottleneck_fold(0): the literal two-layer form (rounds 1-4), for A/B measurements and bit-level comparisons. */
int aon_set_bottleneck_fold(int on);
int aon_get_bottleneck_fold(void);
int aon_stream_is_folded(const void* packed);   /* 1 / 0: the form `packed` (stream or per-call block) would be run in */
int aon_stream_form(const void* packed);        /* 1 folded, 0 literal, -1 never packed / declared by this process */
int aon_declare_stream_form(const void* packed, int form);   /* form 0 / 1: states the form of a COPY of a packed buffer (or re-states a known one) */

/* As aon_pack_vanilla_mlp for a NeRFMLP(min_deg_point, max_deg_point, deg_view) of default widths and depths whose encodings have
 * at most 10 / 4 frequency levels (pts_linears.0: (256, 3 + 6 L), pts_linears.5: (256, 256 + 3 + 6 L), views_linear.0: (128, 256 +
 * 3 + 6 deg_view)): same stream size, zero weight in the slots of the missing levels.  Consumed by aon_render_fwd_ex with the same
 * degrees in aon_render_opts, or by aon_mlp_fwd_enc on encodings in the padded 63 / 27-column layout. */
int aon_pack_vanilla_mlp_deg(const float* const* params_host, int min_deg_point, int max_deg_point, int deg_view, void* packed, void* stream);
/* ... and its transposed twin for the backward data chain (aon_pack_vanilla_mlp_bwd): with these two streams and the degrees in
 * aon_render_opts, aon_render_fwd_train_ex / aon_render_bwd_ex train such a network on the fused kernels too -- the forward on
 * encodings in the padded layout, the three encoding-fed weight gradients (pts_linears.0, pts_linears.5, views_linear.0) written in
 * the network's own (3 + 6 L)-wide column order. */
int aon_pack_vanilla_mlp_bwd_deg(const float* const* params_host, int min_deg_point, int max_deg_point, int deg_view, void* packed_bwd, void* stream);

/* ---- R3(cast)+R4+R5  cast_rays + pos_enc + NeRFMLP.forward fused (model.py:175-181 -> :95-120) ----
 * raw (n*S,4) = (raw_rgb[3], raw_density) per sample, before the sigmoid/relu of model.py:186-187. */
int aon_mlp_fwd(const void* packed, const float* rays_o, const float* rays_d, const float* viewdirs,
                const float* t_vals, int64_t n_rays, int S, float* raw, void* stream);

/* ---- R5  NeRFMLP.forward(x, condition) on caller-encoded inputs (model.py:95-120), stage-level ----
 * samples_enc (n,S,63), viewdirs_enc (n,27). */
int aon_mlp_fwd_enc(const void* packed, const float* samples_enc, const float* viewdirs_enc, int64_t n_rays, int S,
                    float* raw, void* stream);

/* ---- R8  helper.volumetric_rendering (helper.py:157-195) ----
 * rgb / sigma are addressed as rgb[g*rgb_stride + c], sigma[g*sigma_stride] for sample g = ray*S + s, so both
 * the reference's separate (n,S,3)/(n,S,1) tensors (strides 3,1) and the fused kernel's packed raw (n*S,4)
 * (rgb = raw, sigma = raw+3, strides 4,4) can be composited.  weights (n,S) may be NULL.
 * depth: NaN -> +inf (helper.py:182); the clamp to the batch's own [min,max] (:183) is an identity. */
int aon_composite(const float* rgb, int rgb_stride, const float* sigma, int sigma_stride, const float* t_vals,
                  const float* dirs, int64_t n_rays, int S, int white_bkgd, int act, float* comp_rgb, float* acc,
                  float* depth, float* weights, void* stream);

/* ---- R6+R7  helper.sorted_piecewise_constant_pdf (helper.py:203-243) and helper.sample_pdf (:246-252) ----
 * The reference geometry: 64 bins, 63 weights, 128 new samples, 65 coarse t's -> 193 sorted t's (any other: aon_sample_pdf_n).
 *   bins     (n,64) or NULL (then bins = mid-points of t_coarse, model.py:163)
 *   weights  pointer to the first of ray 0's 63 weights; w_stride floats between rays (63 dense, or 65 with
 *            weights = coarse_weights + 1 for the reference's weights[..., 1:-1], model.py:166)
 *   t_coarse (n,65); may be NULL when only `samples` is requested
 *   u        the uniform draws: (128,) shared by all rays when u_stride == 0 (randomized=False: pass
 *            torch.linspace(0, 1-2^-32, 128), helper.py:229), else (n,128) with u_stride = 128
 *   samples  (n,128) unsorted draws, or NULL;   t_fine (n,193) sorted union with t_coarse, or NULL */
int aon_sample_pdf(const float* bins, const float* weights, int64_t w_stride, const float* t_coarse, const float* u,
                   int64_t u_stride, int64_t n_rays, float* samples, float* t_fine, void* stream);

/* ---- R8 + R6/R7 of the coarse level in one kernel: NeRF.forward, model.py:160-173 (helper.volumetric_rendering on the 65
 * coarse samples, then helper.sample_pdf on weights[..., 1:-1] over the mid-points of t_coarse) ----
 *   raw      (n*65,4) packed (raw rgb, raw sigma) records as the MLP entry points write them, 16-byte aligned
 *   t_coarse (n,65), dirs (n,3), u / u_stride as aon_sample_pdf
 * Outputs as aon_composite (weights (n,65) may be NULL: they never leave the registers) plus t_fine (n,193).  Same bits
 * as aon_composite followed by aon_sample_pdf.  The whole-path entry points use this kernel for the coarse level;
 * aon_set_coarse_fusion(0) makes them run the two stage kernels instead (measurements, equality tests). */
int aon_composite_pdf(const float* raw, const float* t_coarse, const float* dirs, int64_t n_rays, int white_bkgd, int act,
                      const float* u, int64_t u_stride, float* comp_rgb, float* acc, float* depth, float* weights,
                      float* t_fine, void* stream);
int aon_set_coarse_fusion(int on);

/* ---- R9  NeRF.forward (model.py:147-199): the whole path in one call ----
 * num_levels 1 (coarse only) or 2.  t_rand (n,65) / u as above, NULL t_rand = randomized False (then u must be
 * the deterministic (128,) vector with u_stride 0).  Outputs: per level comp_rgb (n,3), acc (n,), depth (n,);
 * the *_f pointers may be NULL when num_levels == 1.  The call is chunked internally so that any workspace of
 * at least aon_render_workspace_bytes(1) works; aon_render_workspace_bytes(n) avoids chunking. */
int64_t aon_render_workspace_bytes(int64_t n_rays);
int aon_render_fwd(const void* packed_coarse, const void* packed_fine, const float* rays_o, const float* rays_d,
                   const float* viewdirs, int64_t n_rays, float near_, float far_, int white_bkgd, int num_levels,
                   const float* t_rand, const float* u, int64_t u_stride, float* rgb_c, float* acc_c,
                   float* depth_c, float* rgb_f, float* acc_f, float* depth_f, void* workspace,
                   int64_t workspace_bytes, void* stream);

/* ---- R10/R11  articulated network: NeRFMLP.forward(pos, condition, latents) (models/vanilla_nerf/
 * model_autodecoder.py:172-239, deformation_mlp=True, enc_after=True) and NeRF_AE_Art.forward (:278-337) ----
 * params: HOST array of 40 DEVICE pointers, order: deformations_linear.{0..3}.{weight,bias}, deformation_layer.{w,b},
 *   pts_linears.{0..7}.{w,b}, views_linear.{0..3}.{w,b}, bottleneck_layer.{w,b}, density_layer.{w,b}, rgb_layer.{w,b}.
 * aon_pack_art_mlp: once per parameter update -> `packed` (aon_art_packed_bytes()).
 * aon_art_prepare : once per (parameters, latents) -> `small` (aon_art_small_bytes()): the three latents
 *   (latents["density"] (128), ["color"] (128), ["articulation"] (32); model_autodecoder.py:172-178) are broadcast to
 *   every sample by the reference, so their weight columns are folded into effective bias vectors here.
 * aon_art_mlp_fwd    : cast_rays + deformation MLP + pos_enc + trunk + view branch, raw (n*S,4) as aon_mlp_fwd.
 * aon_art_mlp_fwd_pos: the same on caller-supplied sample positions pos (n,S,3) and encoded view dirs (n,27),
 *   i.e. exactly NeRFMLP.forward(pos, condition, latents).
 * aon_art_render_fwd : NeRF_AE_Art.forward; as aon_render_fwd with rgb = sigmoid*1.002-0.001, sigma = softplus(raw-1). */
int64_t aon_art_packed_bytes(void);
int64_t aon_art_small_bytes(void);
int aon_pack_art_mlp(const float* const* params_host, void* packed, void* stream);
/* Round 4: the articulated network at other encoding degrees (NeRFMLP(min_deg_point, max_deg_point, deg_view), model_autodecoder.py:60-170;
 * at most 10 position and 4 view frequency levels, default widths).  The streams keep the kernels' 63 / 27-wide slots with zero weight in
 * the levels the network lacks (row strides of pts_linears.0 / .5 and views_linear.0 follow P = 3 + 6 L, V = 3 + 6 Lv), and the small
 * block carries the ten encoding scales 2^(min_deg_point + l) the kernels multiply the DEFORMED point by (0 for a missing level), so
 * every aon_art_* call works unchanged on streams / blocks made by the _deg forms; aon_art_render_bwd_ex and aon_art_wgrad_deg take the
 * degrees (aon_render_opts / arguments) for the layout of the gradients they write.  The plain forms are these with (0, 10, 4).
 * WARNING: nothing cross-checks the degrees a stream / block was packed with against the degrees handed to aon_art_render_bwd_ex
 * (aon_render_opts, NULL = (0, 10, 4)) or aon_art_wgrad_deg: they fix the ROW STRIDES of the three concatenating layers' gradients
 * (P + 128, 256 + P + 128, 256 + V + 128 columns), so passing larger degrees than the gradient buffers were allocated for writes out
 * of bounds, and smaller ones leave columns unwritten.  Pass the same three numbers to pack, prepare, backward and size your buffers
 * by them.  aon_art_small_bytes() grew from 18,976 to 19,024 bytes in round 4 (the ten encoding scales): query it, do not hard-code it. */
int aon_pack_art_mlp_deg(const float* const* params_host, int min_deg_point, int max_deg_point, int deg_view, void* packed, void* stream);
int aon_art_prepare_deg(const float* const* params_host, const float* shape, const float* appearance, const float* articulation,
                        int min_deg_point, int max_deg_point, int deg_view, void* small, void* stream);
int aon_art_prepare(const float* const* params_host, const float* shape, const float* appearance,
                    const float* articulation, void* small, void* stream);
int aon_art_mlp_fwd(const void* packed, const void* small, const float* rays_o, const float* rays_d,
                    const float* viewdirs, const float* t_vals, int64_t n_rays, int S, float* raw, void* stream);
int aon_art_mlp_fwd_pos(const void* packed, const void* small, const float* pos, const float* viewdirs_enc,
                        int64_t n_rays, int S, float* raw, void* stream);
int aon_art_render_fwd(const void* packed_coarse, const void* small_coarse, const void* packed_fine,
                       const void* small_fine, const float* rays_o, const float* rays_d, const float* viewdirs,
                       int64_t n_rays, float near_, float far_, int white_bkgd, int num_levels, const float* t_rand,
                       const float* u, int64_t u_stride, float* rgb_c, float* acc_c, float* depth_c, float* rgb_f,
                       float* acc_f, float* depth_f, void* workspace, int64_t workspace_bytes, void* stream);

/* ---- R14  backward of the vanilla path (what autograd does for loss.backward(), model.py:264-273) ----
 * Gradients reach only the MLP parameters (t_samples is detached, helper.py:249).  Per level (S = 65 or 193):
 *   aon_mlp_fwd_train   as aon_mlp_fwd, and additionally stores the layer activations as STEP-MAJOR planes (round 3; rounds 1-2
 *                       were feature-major): rows = aon_train_plane_rows() feature rows (a multiple of 4), Np = 128 *
 *                       ceil(n*S / 128) samples, rows * Np floats; feature row f of sample s lives at float offset
 *                           ((s / 32) * (rows / 4) + f / 4) * 128 + (s % 32) * 4 + f % 4
 *                       i.e. [step of 32 samples][f / 4][sample in step][f % 4]: a 16-byte unit holds four consecutive
 *                       feature rows of one sample, the kernels store / fetch whole units (csrc/aon_mlp_core.h); and the
 *                       ReLU decisions of the nine activated layers as bit masks (aon_train_mask_bytes(Np) bytes).
 *   aon_composite_bwd   (g_rgb (n,3), optional g_acc (n,), g_depth (n,)) -> d_raw (n*S,4) = dL/d(raw rgb, raw sigma);
 *                       the caller zero-fills d_raw up to Np rows (padded samples must carry zero gradient); S <= 512; evaluated
 *                       in fp64 on the fp32 inputs, every output rounded once.
 *   aon_mlp_bwd_chain   data-gradient chain through the MLP (ReLU derivatives from `masks`); writes the pre-activation
 *                       gradient planes `dplanes` (same layout / row map as `planes`, every sample written); needs the
 *                       transposed stream of aon_pack_vanilla_mlp_bwd.
 *   aon_vanilla_wgrad   all 24 parameter gradients (order of aon_pack_vanilla_mlp, full nn.Linear shapes, overwritten)
 *                       from planes x dplanes: one grouped launch over all layers + one over the heads + one second stage
 *                       (csrc/aon_wgrad.h); workspace >= aon_wgrad_workspace_bytes(); deterministic (no atomics).
 *                       packed_bwd (round 5): the transposed stream aon_mlp_bwd_chain ran with -- its form says whether the planes
 *                       carry bottleneck rows, and a folded stream's buffer holds the raw W_v0[:, :256], W_b, b_b the un-folding
 *                       of (dW', db') reads.  NULL = planes of the literal form.  aon_art_wgrad* likewise (form only). */
int64_t aon_train_plane_rows(void);
int64_t aon_train_mask_bytes(int64_t Np);
int64_t aon_bwd_packed_bytes(void);
int64_t aon_wgrad_workspace_bytes(void);
int aon_pack_vanilla_mlp_bwd(const float* const* params_host, void* packed_bwd, void* stream);
int aon_mlp_fwd_train(const void* packed, const float* rays_o, const float* rays_d, const float* viewdirs,
                      const float* t_vals, int64_t n_rays, int S, float* raw, float* planes, void* masks, void* stream);
int aon_composite_bwd(const float* raw, const float* t_vals, const float* dirs, const float* g_rgb, const float* g_acc,
                      const float* g_depth, int64_t n_rays, int S, int white_bkgd, int act, float* d_raw, void* stream);
int aon_mlp_bwd_chain(const void* packed_bwd, const void* packed_fwd, const float* d_raw, const void* masks,
                      float* dplanes, int64_t Np, void* stream);
int aon_vanilla_wgrad(const float* planes, const float* dplanes, const float* d_raw, int64_t Np,
                      float* const* grads_host, void* workspace, int64_t workspace_bytes, void* stream, const void* packed_bwd);
/* Host-only: the plan aon_vanilla_wgrad / aon_art_wgrad would run for a level of Np samples on `cus` compute units.  Round 4: the
 * steps of all jobs form one work line priced in cost units, and each of the G = min(cus, 304, steps) workgroups owns the steps that
 * start in its 1/G of the line (csrc/aon_wgrad.h), so a workgroup runs up to a few segments of consecutive jobs.  Per job six ints:
 * kind, first workgroup owning a step of it, number of such workgroups, steps of the job (Np / 32), partial offset in floats,
 * partials.  Returns the number of jobs (<= max_jobs) or a negative status; `ws_bytes` = workspace bytes used.  No GPU needed (tests). */
int aon_wgrad_plan(int articulated, int64_t Np, int cus, int32_t* jobs6, int max_jobs, int64_t* ws_bytes);
/* Host-only: begin_end[0..1] = the steps [begin, end) of job `job` that workgroup `workgroup` owns under that plan (the kernel's own
 * arithmetic restated on the host).  Returns 1 when the workgroup writes partials for the job (end may equal begin in tiny problems:
 * a zero partial), 0 when it does not, negative on bad arguments. */
int aon_wgrad_plan_segment(int articulated, int64_t Np, int cus, int job, int workgroup, int32_t* begin_end);
/* Measurement aid: while a device buffer of >= 2 x 304 int64 is set, every grouped weight-gradient launch on any stream writes
 * [2 w] / [2 w + 1] = the 100 MHz wall clock at entry / exit of workgroup w (tools/kernel_bench.py --wgrad-probe); NULL = off. */
int aon_set_wgrad_probe(void* device_buffer);
/* Measurement aid: `nlayers` (<= 20) identical weight-gradient jobs of one kind (0: 256x256, 1: 128x128, 2: 256x64, 3: 128x256, 4: 128x32;
 * csrc/aon_wgrad.h) on arbitrary rows of two plane buffers, the grouped kernel only, partials left in the workspace. */
int aon_wgrad_kind_bench(int kind, int nlayers, const float* planes, const float* dplanes, int rows, int64_t Np, void* workspace,
                         int64_t workspace_bytes, void* stream);

/* ---- R14 for the articulated network (training_step, model_autodecoder.py:395-477) ----
 * Same four stages as the vanilla backward, on the articulated row map (aon_art_train_plane_rows() rows).  Gradients reach
 * the 40 parameters AND the three latents.  aon_art_bwd_chain also writes dxp (Np,4) = dL/d(deformed position) per sample,
 * consumed by aon_art_wgrad for deformation_layer's gradient.  aon_art_wgrad needs the forward's parameters and latents
 * again: the latent columns of the weights get db (x) latent, the latents get W[:, latent cols]^T db. */
int64_t aon_art_train_plane_rows(void);
int64_t aon_art_train_mask_bytes(int64_t Np);
int64_t aon_art_bwd_packed_bytes(void);
int aon_pack_art_mlp_bwd(const float* const* params_host, void* packed_bwd, void* stream);
int aon_pack_art_mlp_bwd_deg(const float* const* params_host, int min_deg_point, int max_deg_point, int deg_view, void* packed_bwd, void* stream);
int aon_art_mlp_fwd_train(const void* packed, const void* small, const float* rays_o, const float* rays_d,
                          const float* viewdirs, const float* t_vals, int64_t n_rays, int S, float* raw, float* planes,
                          void* masks, void* stream);
int aon_art_bwd_chain(const void* packed_bwd, const void* small, const float* d_raw, const void* masks,
                      const float* planes, float* dplanes, float* dxp, int64_t Np, void* stream);
int aon_art_wgrad(const float* planes, const float* dplanes, const float* d_raw, const float* dxp, int64_t Np,
                  const float* const* params_host, const float* shape, const float* appearance,
                  const float* articulation, float* const* grads_host, float* g_shape, float* g_appearance,
                  float* g_articulation, void* workspace, int64_t workspace_bytes, void* stream, const void* packed_bwd);
int aon_art_wgrad_deg(const float* planes, const float* dplanes, const float* d_raw, const float* dxp, int64_t Np,
                      const float* const* params_host, const float* shape, const float* appearance,
                      const float* articulation, float* const* grads_host, float* g_shape, float* g_appearance,
                      float* g_articulation, void* workspace, int64_t workspace_bytes, void* stream, int min_deg_point,
                      int max_deg_point, int deg_view, const void* packed_bwd);

/* ---- R14, the training step in two calls ----
 * aon_render_fwd_train = NeRF.forward under grad mode (model.py:147-199 as called by training_step :264): both levels,
 * same arguments and outputs as aon_render_fwd, plus everything the backward needs (per-level t values, raw outputs,
 * activation planes, ReLU bits, coarse weights) left in `workspace` (>= aon_train_workspace_bytes(n_rays, articulated,
 * num_levels), 256-byte aligned; NOT to be touched until the backward ran: 15 GB at 4096 articulated rays and two levels, a
 * quarter of that with one).  aon_render_bwd = loss.backward() through that forward; its temporaries (gradient planes,
 * d_raw, weight-gradient partials: 11 GB) live in a separate `scratch` (>= aon_train_scratch_bytes(...), 256-byte aligned)
 * that only has to exist during the call, so a live graph pins the forward's workspace alone.  aon_render_bwd
 * (model.py:271-282): given dL/d(comp_rgb, acc, depth) of each level (HOST arrays of num_levels device pointers; the acc /
 * depth arrays or their entries may be NULL) it writes all 24 parameter gradients per level (order of
 * aon_pack_vanilla_mlp, full nn.Linear shapes, overwritten).  packed_bwd_* from aon_pack_vanilla_mlp_bwd, packed_fwd_* the
 * forward streams (head weights).  Exact-fp32 engine.  The articulated twins (model_autodecoder.py:278-337, :395-477) take
 * the per-call small blocks of aon_art_prepare, the 40 parameters per level and the three latents, and also return the
 * latent gradients summed over the levels.
 * With two levels the backward of each level (independent of the other) runs on its own library-owned stream so that one
 * level's kernels fill the CUs the other's tail rounds leave idle; both are ordered after everything already enqueued on
 * `stream` and `stream` continues only after both (events) -- from the caller's view the call is enqueued on `stream`.
 * aon_set_bwd_overlap(0) keeps everything on `stream` (default 1).  The level streams are only used when the merged chain launch is off
 * (aon_set_bwd_merge(0)); 2 additionally puts the head reductions of the merged form on side streams beside each level's grouped
 * weight-gradient kernel (round 4 measured it slower, 34.5 against 34.1 ms per step, profiles/r04_backward_schedules.txt; round 6, with
 * today's kernels: 30.31 against 30.42 ms for the ARTICULATED network -- whose calls therefore treat 1 as 2 -- and no gain for the vanilla
 * one, which keeps 1). */
int aon_set_bwd_overlap(int on);
/* Likewise the training FORWARD of two levels runs two ray halves (split on a multiple of 128 rays) on the two library streams: a
 * half's levels depend on each other only through its own inverse CDF, so one half's fine level fills the CUs the other half's
 * coarse level leaves idle in its last partial round of workgroups.  Same bits as the one-stream form.  aon_set_fwd_overlap(0)
 * keeps everything on `stream` (default 1 = two halves; k >= 2 = k ranges alternating on the two streams, for measurements). */
int aon_set_fwd_overlap(int on);
/* Round 4, default 1 and in front of the above: the training forward of two levels as THREE persistent launches on `stream` --
 * coarse(A) | fine(A) + coarse(B) | fine(B) for two ray ranges A, B split on a multiple of 128 rays chosen to minimise the rounds of
 * workgroups (4096 x (65 + 193) samples on 256 CUs: 33 rounds instead of 9 + 25) -- the middle launch carrying two networks'
 * passes (model.py:149-197 / model_autodecoder.py:297-335: a range's fine level needs only its own coarse weights).  Same bits as
 * the one-launch-per-level form.  aon_set_fwd_merge(0) falls back to aon_set_fwd_overlap's forms; 2 (tests) merges whenever the batch has
 * two 128-ray ranges, whether or not rounds are saved. */
int aon_set_fwd_merge(int on);
/* Round 4, default 1: the data-gradient chains of the two levels (independent of each other) run as ONE persistent launch of two
 * segments on `stream` -- 8,256 passes = 33 rounds of 256 workgroups at 4096 x (65 + 193) samples where two launches cost 9 + 25 --
 * followed by the two levels' weight gradients on the two library streams (aon_set_bwd_overlap).  Same bits.  0: one chain launch
 * per level, each on its level's stream (round 3). */
int aon_set_bwd_merge(int on);
/* Round 5, default 1 (merged form only): the head / bias reductions that need nothing from the chain -- density head on the layer-7
 * output, rgb head, the sums of d_raw: 60 % of the head kernel's bytes -- are launched on a library side stream behind the merged
 * chain launch, whose last round of workgroups is a quarter full (8,256 passes on 256 compute units at 4096 x (65 + 193) samples):
 * they run on compute units that would idle for one pass.  Ordered by events on both sides like the level streams; same bits
 * (the partials are summed by the same second stage).  0: every head reduction behind the chain, on `stream` (round 4). */
int aon_set_bwd_early_heads(int on);
/* Round 5, default 1: whole-path calls (aon_render_fwd*, aon_art_render_fwd*, the training forwards) of a network in its folded form compute
 * b' + W_v0[:, 256:283] viewenc(ray) -- a constant of the RAY, the head of the first view layer's accumulation chains (articulated: b' is
 * the per-call effective bias) -- once per ray in a small kernel and start that layer's accumulators from it, instead of running the
 * 27 -> 128 view-encoding chunk for every sample:
 * 56 MFMAs, 12 sines and a 16 KiB weight chunk fewer per 128-sample pass, the same fused multiply-adds in the same order (bit-equal to
 * the chunk form, which the stage-level calls keep).  0: the chunk form everywhere (the A/B partner). */
int aon_set_view_bias(int on);
int aon_get_view_bias(void);
/* The per-ray kernel on its own: view_bias (n_rays,128) = b' + W_v0[:, 256:] pos_enc(viewdirs, 0, 4) of a FOLDED vanilla stream
 * (model.py:110-116; W_v0 = views_linear.0.weight, b' = W_v0[:, :256] b_bottleneck + b_v0). */
int aon_view_bias(const void* packed, const float* viewdirs, int64_t n_rays, float* view_bias, void* stream);
int64_t aon_train_workspace_bytes(int64_t n_rays, int articulated, int num_levels);
int64_t aon_train_scratch_bytes(int64_t n_rays, int articulated, int num_levels);
int aon_render_fwd_train(const void* packed_coarse, const void* packed_fine, const float* rays_o, const float* rays_d,
                         const float* viewdirs, int64_t n_rays, float near_, float far_, int white_bkgd, int num_levels,
                         const float* t_rand, const float* u, int64_t u_stride, float* rgb_c, float* acc_c, float* depth_c,
                         float* rgb_f, float* acc_f, float* depth_f, void* workspace, int64_t workspace_bytes, void* stream);
int aon_render_bwd(const void* packed_bwd_coarse, const void* packed_fwd_coarse, const void* packed_bwd_fine,
                   const void* packed_fwd_fine, const float* rays_d, int64_t n_rays, int white_bkgd, int num_levels,
                   const float* const* g_rgb_host, const float* const* g_acc_host, const float* const* g_depth_host,
                   float* const* grads_coarse_host, float* const* grads_fine_host, void* workspace,
                   int64_t workspace_bytes, void* scratch, int64_t scratch_bytes, void* stream);
int aon_art_render_fwd_train(const void* packed_coarse, const void* small_coarse, const void* packed_fine,
                             const void* small_fine, const float* rays_o, const float* rays_d, const float* viewdirs,
                             int64_t n_rays, float near_, float far_, int white_bkgd, int num_levels, const float* t_rand,
                             const float* u, int64_t u_stride, float* rgb_c, float* acc_c, float* depth_c, float* rgb_f,
                             float* acc_f, float* depth_f, void* workspace, int64_t workspace_bytes, void* stream);
int aon_art_render_bwd(const void* packed_bwd_coarse, const void* small_coarse, const void* packed_bwd_fine,
                       const void* small_fine, const float* rays_d, int64_t n_rays, int white_bkgd, int num_levels,
                       const float* const* g_rgb_host, const float* const* g_acc_host, const float* const* g_depth_host,
                       const float* const* params_coarse_host, const float* const* params_fine_host, const float* shape,
                       const float* appearance, const float* articulation, float* const* grads_coarse_host,
                       float* const* grads_fine_host, float* g_shape, float* g_appearance, float* g_articulation,
                       void* workspace, int64_t workspace_bytes, void* scratch, int64_t scratch_bytes, void* stream);

/* ---- Constructor arguments of NeRF / NeRF_AE_Art beyond their defaults (round 3) ----
 * NeRF.__init__(num_coarse_samples, num_fine_samples, noise_std, lindisp)          models/vanilla_nerf/model.py:124-135
 * NeRF_AE_Art.__init__(..., rgb_padding, density_bias)                             models/vanilla_nerf/model_autodecoder.py:241-257
 * The entry points above are these with the reference's defaults; the *_ex forms take the struct (NULL = defaults).  Scalars are
 * passed as the reference's fp32 tensor arithmetic sees its Python numbers (the host side computes them in double precision and
 * rounds once, as torch does when a Python scalar meets an fp32 tensor):
 *   num_coarse_samples  level 0 evaluates num_coarse_samples + 1 t values (helper.py:115); >= 2, <= 1023
 *   num_fine_samples    draws of the inverse CDF (helper.py:224-230); level 1 evaluates num_coarse_samples + 1 + num_fine_samples
 *                       (<= 512 in the training entry points)
 *   lindisp             helper.py:116-117: t = 1 / (inv_near (1 - s) + inv_far s) with inv_near = fp32(1.0 / near), inv_far =
 *                       fp32(1.0 / far) -- the reference evaluates 1.0 / near in Python double precision
 *   noise_c, noise_f    model.py:183-184 (`noise_std > 0 and randomized`): the caller's torch.rand_like(raw_sigma) draws of level
 *                       0 (n, num_coarse_samples + 1) and level 1 (n, num_coarse_samples + 1 + num_fine_samples); raw_sigma +=
 *                       noise * noise_std before the activation.  NULL = no noise at that level.
 *   rgb_scale, rgb_shift, sigma_bias   articulated activations (model_autodecoder.py:321-323): rgb = sigmoid(raw) * rgb_scale -
 *                       rgb_shift, sigma = softplus(raw_sigma + sigma_bias); fp32(1 + 2 rgb_padding), fp32(rgb_padding),
 *                       fp32(density_bias).  Ignored by the vanilla entry points.
   min_deg_point, max_deg_point, deg_view   NeRF.__init__'s encoding degrees (model.py:126-128) on the FUSED inference kernels, for a
 *                       default-size NeRFMLP with max_deg_point - min_deg_point <= 10 and deg_view <= 4: the stream of
 *                       aon_pack_vanilla_mlp_deg leaves the 63 / 27-wide input slots of the missing levels at zero weight, and
 *                       aon_render_fwd_ex computes the encodings outside the MLP kernel in that padded layout (pos_enc stage
 *                       kernel, +252 B/sample of HBM traffic) and runs the MLP on them; aon_render_fwd_train_ex / aon_render_bwd_ex
 *                       likewise (streams of aon_pack_vanilla_mlp_deg / _bwd_deg).  The articulated calls refuse other degrees.
 * Geometries other than 64 / 128 run the coarse level as two kernels (compositing, then aon_sample_pdf_n). */
typedef struct aon_render_opts {
  int32_t num_coarse_samples;   /* 64 */
  int32_t num_fine_samples;     /* 128 */
  int32_t lindisp;              /* 0 */
  float inv_near, inv_far;      /* read when lindisp != 0 */
  float noise_std;              /* 0 */
  const float* noise_c;         /* NULL */
  const float* noise_f;         /* NULL */
  float rgb_scale, rgb_shift, sigma_bias;   /* 1.002f, 0.001f, -1.0f */
  int32_t min_deg_point, max_deg_point, deg_view;   /* 0, 10, 4 */
} aon_render_opts;
void aon_render_opts_init(aon_render_opts* opts);   /* the reference's defaults */

/* helper.sample_along_rays with lindisp (helper.py:116-117); as aon_sample_along_rays otherwise */
int aon_sample_along_rays_ex(const float* rays_o, const float* rays_d, int64_t n_rays, int S, float near_, float far_, int lindisp,
                             float inv_near, float inv_far, const float* t_rand, float* t_vals, float* coords, void* stream);
/* aon_composite with the activation scalars and the density noise of `opts` (noise_c is the (n,S) noise of THIS call; the sample
 * counts and lindisp fields are ignored) */
int aon_composite_ex(const float* rgb, int rgb_stride, const float* sigma, int sigma_stride, const float* t_vals, const float* dirs,
                     int64_t n_rays, int S, int white_bkgd, int act, const aon_render_opts* opts, float* comp_rgb, float* acc,
                     float* depth, float* weights, void* stream);
/* helper.sorted_piecewise_constant_pdf / helper.sample_pdf (helper.py:203-252) for ANY sizes: num_bins bins (n,num_bins) -- or
 * NULL = mid-points of t_coarse, then num_t == num_bins + 1 -- num_bins - 1 weights per ray (w_stride floats apart), num_samples
 * draws u ((num_samples,) shared when u_stride == 0, else (n, >= num_samples)), num_t coarse t's; outputs samples (n,num_samples)
 * and / or t_fine (n, num_t + num_samples).  Bit-exact against torch's CPU kernels like aon_sample_pdf (ATen's summation order
 * for any length, double running sum).  2 <= num_bins, 3 * num_bins + pow2ceil(num_t + num_samples) <= 16384. */
int aon_sample_pdf_n(const float* bins, const float* weights, int64_t w_stride, const float* t_coarse, const float* u,
                     int64_t u_stride, int64_t n_rays, int num_bins, int num_samples, int num_t, float* samples, float* t_fine,
                     void* stream);
int64_t aon_render_workspace_bytes_ex(int64_t n_rays, const aon_render_opts* opts);
int aon_render_fwd_ex(const void* packed_coarse, const void* packed_fine, const float* rays_o, const float* rays_d,
                      const float* viewdirs, int64_t n_rays, float near_, float far_, int white_bkgd, int num_levels,
                      const float* t_rand, const float* u, int64_t u_stride, float* rgb_c, float* acc_c, float* depth_c,
                      float* rgb_f, float* acc_f, float* depth_f, void* workspace, int64_t workspace_bytes, void* stream,
                      const aon_render_opts* opts);
int aon_art_render_fwd_ex(const void* packed_coarse, const void* small_coarse, const void* packed_fine, const void* small_fine,
                          const float* rays_o, const float* rays_d, const float* viewdirs, int64_t n_rays, float near_, float far_,
                          int white_bkgd, int num_levels, const float* t_rand, const float* u, int64_t u_stride, float* rgb_c,
                          float* acc_c, float* depth_c, float* rgb_f, float* acc_f, float* depth_f, void* workspace,
                          int64_t workspace_bytes, void* stream, const aon_render_opts* opts);
/* training twins: the SAME opts (sizes, noise pointers, activation scalars) must be handed to the forward, the backward and the
 * two size queries of one step */
int64_t aon_train_workspace_bytes_ex(int64_t n_rays, int articulated, int num_levels, const aon_render_opts* opts);
int64_t aon_train_scratch_bytes_ex(int64_t n_rays, int articulated, int num_levels, const aon_render_opts* opts);
int aon_render_fwd_train_ex(const void* packed_coarse, const void* packed_fine, const float* rays_o, const float* rays_d,
                            const float* viewdirs, int64_t n_rays, float near_, float far_, int white_bkgd, int num_levels,
                            const float* t_rand, const float* u, int64_t u_stride, float* rgb_c, float* acc_c, float* depth_c,
                            float* rgb_f, float* acc_f, float* depth_f, void* workspace, int64_t workspace_bytes, void* stream,
                            const aon_render_opts* opts);
int aon_render_bwd_ex(const void* packed_bwd_coarse, const void* packed_fwd_coarse, const void* packed_bwd_fine,
                      const void* packed_fwd_fine, const float* rays_d, int64_t n_rays, int white_bkgd, int num_levels,
                      const float* const* g_rgb_host, const float* const* g_acc_host, const float* const* g_depth_host,
                      float* const* grads_coarse_host, float* const* grads_fine_host, void* workspace,
                      int64_t workspace_bytes, void* scratch, int64_t scratch_bytes, void* stream, const aon_render_opts* opts);
int aon_art_render_fwd_train_ex(const void* packed_coarse, const void* small_coarse, const void* packed_fine,
                                const void* small_fine, const float* rays_o, const float* rays_d, const float* viewdirs,
                                int64_t n_rays, float near_, float far_, int white_bkgd, int num_levels, const float* t_rand,
                                const float* u, int64_t u_stride, float* rgb_c, float* acc_c, float* depth_c, float* rgb_f,
                                float* acc_f, float* depth_f, void* workspace, int64_t workspace_bytes, void* stream,
                                const aon_render_opts* opts);
int aon_art_render_bwd_ex(const void* packed_bwd_coarse, const void* small_coarse, const void* packed_bwd_fine,
                          const void* small_fine, const float* rays_d, int64_t n_rays, int white_bkgd, int num_levels,
                          const float* const* g_rgb_host, const float* const* g_acc_host, const float* const* g_depth_host,
                          const float* const* params_coarse_host, const float* const* params_fine_host, const float* shape,
                          const float* appearance, const float* articulation, float* const* grads_coarse_host,
                          float* const* grads_fine_host, float* g_shape, float* g_appearance, float* g_articulation,
                          void* workspace, int64_t workspace_bytes, void* scratch, int64_t scratch_bytes, void* stream,
                          const aon_render_opts* opts);

/* ---- NeRFMLP of ANY constructor geometry (models/vanilla_nerf/model.py:40-120), round 3 ----
 * The entry points above run the reference's default NeRFMLP (8 x 256, skip 4, 1 x 128, degrees 10 / 4) on fused kernels compiled
 * for it.  Every other geometry runs here: layer-wise fp32 MFMA GEMMs on the unmodified nn.Linear storages (csrc/aon_gmlp.hip).
 * aon_mlp_geometry mirrors NeRFMLP.__init__'s arguments (model.py:40-54); pos_size = ((max_deg_point - min_deg_point) * 2 + 1) *
 * input_ch, view_pos_size = (deg_view * 2 + 1) * input_ch_view.  params: HOST array of 2 * (netdepth + netdepth_condition + 3)
 * DEVICE pointers: pts_linears.{0..netdepth-1}.{weight,bias}, views_linear.{0..netdepth_condition-1}.{weight,bias},
 * bottleneck_layer.{w,b}, density_layer.{w,b}, rgb_layer.{w,b} (the vanilla order for the default depths).
 * A geometry whose LAST trunk layer would concatenate the encoding (netdepth - 1 > 0 and (netdepth - 1) % skip_layer == 0) is
 * rejected: the reference's own forward fails on it (density_layer is built for netwidth inputs, model.py:90 vs :103-104).
 *   aon_gmlp_fwd            NeRFMLP.forward(x, condition): samples_enc (n,S,pos_size), viewdirs_enc (n,view_pos_size) ->
 *                           raw_rgb (n*S, num_rgb_channels), raw_density (n*S, num_density_channels)
 *   aon_grender_fwd         NeRF.forward (model.py:147-199) with this NeRFMLP at both levels and the sampler options of
 *                           aon_render_opts; needs input_ch = input_ch_view = 3, 3 rgb channels, 1 density channel.  Chunked
 *                           internally to the workspace given (>= aon_grender_workspace_bytes(geom, 1, opts)).
 *   aon_grender_fwd_train / aon_grender_bwd   the training step in two calls, as aon_render_fwd_train / aon_render_bwd: the
 *                           forward keeps every layer's output in `workspace` (aon_grender_train_workspace_bytes), the backward
 *                           writes all parameter gradients per level (order and shapes of `params`, overwritten); its
 *                           temporaries live in `scratch` (aon_grender_train_scratch_bytes).  Deterministic (no atomics).
 * The aon_grender_* calls take the encoding degrees from aon_mlp_geometry; the degree fields of aon_render_opts are ignored there
 * (any value is accepted: they are validated against the fused kernels' 10 / 4-level limits by the fused entry points only). */
typedef struct aon_mlp_geometry {
  int32_t min_deg_point, max_deg_point, deg_view;
  int32_t netdepth, netwidth, netdepth_condition, netwidth_condition, skip_layer;
  int32_t input_ch, input_ch_view, num_rgb_channels, num_density_channels;
} aon_mlp_geometry;
void aon_mlp_geometry_init(aon_mlp_geometry* geom);   /* (0, 10, 4, 8, 256, 1, 128, 4, 3, 3, 3, 1) */
int aon_gmlp_param_count(const aon_mlp_geometry* geom);   /* number of pointers in `params`, or a negative status */
int64_t aon_gmlp_workspace_bytes(const aon_mlp_geometry* geom, int64_t n_samples);
int aon_gmlp_fwd(const aon_mlp_geometry* geom, const float* const* params_host, const float* samples_enc, const float* viewdirs_enc,
                 int64_t n_rays, int S, float* raw_rgb, float* raw_density, void* workspace, int64_t workspace_bytes, void* stream);
int64_t aon_grender_workspace_bytes(const aon_mlp_geometry* geom, int64_t n_rays, const aon_render_opts* opts);
int aon_grender_fwd(const aon_mlp_geometry* geom, const float* const* params_coarse_host, const float* const* params_fine_host,
                    const float* rays_o, const float* rays_d, const float* viewdirs, int64_t n_rays, float near_, float far_,
                    int white_bkgd, int num_levels, const float* t_rand, const float* u, int64_t u_stride, float* rgb_c, float* acc_c,
                    float* depth_c, float* rgb_f, float* acc_f, float* depth_f, void* workspace, int64_t workspace_bytes, void* stream,
                    const aon_render_opts* opts);
int64_t aon_grender_train_workspace_bytes(const aon_mlp_geometry* geom, int64_t n_rays, int num_levels, const aon_render_opts* opts);
int64_t aon_grender_train_scratch_bytes(const aon_mlp_geometry* geom, int64_t n_rays, int num_levels, const aon_render_opts* opts);
int aon_grender_fwd_train(const aon_mlp_geometry* geom, const float* const* params_coarse_host, const float* const* params_fine_host,
                          const float* rays_o, const float* rays_d, const float* viewdirs, int64_t n_rays, float near_, float far_,
                          int white_bkgd, int num_levels, const float* t_rand, const float* u, int64_t u_stride, float* rgb_c,
                          float* acc_c, float* depth_c, float* rgb_f, float* acc_f, float* depth_f, void* workspace,
                          int64_t workspace_bytes, void* stream, const aon_render_opts* opts);
int aon_grender_bwd(const aon_mlp_geometry* geom, const float* const* params_coarse_host, const float* const* params_fine_host,
                    const float* rays_d, int64_t n_rays, int white_bkgd, int num_levels, const float* const* g_rgb_host,
                    const float* const* g_acc_host, const float* const* g_depth_host, float* const* grads_coarse_host,
                    float* const* grads_fine_host, void* workspace, int64_t workspace_bytes, void* scratch, int64_t scratch_bytes,
                    void* stream, const aon_render_opts* opts);

/* ---- measurement aid (no reference counterpart) ----
 * Between aon_profile_begin() and aon_profile_end() every launch of the path's kernels made through this library is
 * bracketed by HIP events recorded on the LAUNCH stream, by kernel class.  aon_profile_end() waits for those events
 * and returns the totals of the dominant class -- the fused MLP forward: summed kernel time (ms), launches, samples
 * (network evaluations); aon_profile_class() then reads the totals of any class of that same interval
 * (units: samples for the MLP / backward-chain / weight-gradient classes, rays for the per-ray kernels;
 * a weight-gradient "launch" is one aon_*_wgrad call = all layers of one level). */
#define AON_PROF_MLP_FWD 0       /* fused encode + MLP forward (inference and training) */
#define AON_PROF_BWD_CHAIN 1     /* backward data-gradient chain */
#define AON_PROF_WGRAD 2         /* weight-gradient GEMMs + partial reductions of one level */
#define AON_PROF_COMPOSITE 3     /* alpha compositing (R8) */
#define AON_PROF_SAMPLE_PDF 4    /* inverse-CDF sampling + merge (R6/R7) */
#define AON_PROF_COMPOSITE_BWD 5 /* compositing backward */
#define AON_PROF_COMPOSITE_PDF 6 /* coarse compositing fused with the inverse CDF + merge (R8 + R6/R7) */
#define AON_PROF_SAMPLE_T 7      /* stratified t of the coarse level (R3) inside the whole-path calls */
#define AON_PROF_NUM_CLASSES 8
int aon_profile_begin(void);
int aon_profile_end(double* mlp_ms, int64_t* mlp_launches, int64_t* mlp_samples);
int aon_profile_class(int kernel_class, double* ms, int64_t* launches, int64_t* units);

#ifdef __cplusplus
}
#endif
#endif /* AON_HIP_H */
