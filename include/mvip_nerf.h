/*
 * mvip_nerf.h — C ABI of libmvipnerf.so, the MI355X (gfx950) implementation of MVIP-NeRF's
 * volume-rendering + SDS hot path.
 *
 * The reference has no FFI on this path (it is stock PyTorch ops called from Python, SURVEY.md
 * §8b), so each entry point below names the reference Python call site it replaces
 * (paths relative to the reference checkout).  A maintainer binds these with ctypes; the binding
 * is shown in INTEGRATION.md and shipped in mvip_nerf_amd/_lib.py.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to fp32 data unless the name ends in _host or the type
 *     says otherwise; tensors are dense row-major with the shapes given in the comments;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); all work is enqueued
 *     on it, nothing synchronises, nothing allocates, no global mutable state: re-entrant per
 *     stream and capturable into a hipGraph;
 *   - return value: MVIP_OK (0) or a negative MVIP_E* code; mvip_strerror() names it.  Launch
 *     errors are read back with hipGetLastError() and reported as MVIP_ELAUNCH.
 *   - "nullable" arguments switch an optional input/output off.
 */
#ifndef MVIP_NERF_H
#define MVIP_NERF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MVIP_OK        0
#define MVIP_EINVAL   -1   /* bad size / null pointer / unsupported configuration */
#define MVIP_ELAUNCH  -2   /* hipLaunch / runtime error (see mvip_last_hip_error) */
#define MVIP_EUNSUP   -3   /* shape outside what the kernels are built for */

#define MVIP_ABI_VERSION 5      /* 2: `prec` on the SDS operand producers / contractions, operand-sink entry points;
                                   3: prec = 2 (two products for fp16-exact weights), mvip_packed_weights_two_product,
                                      mvip_build_is_experiment;
                                   4: mvip_mlp_*_f16x3_w16 (two-waves-per-SIMD split-precision forward) added,
                                      mvip_mlp_forward_rays16_persistent removed;
                                   5: LayerNorm statistics from the producing GEMM's epilogue (mvip_gemm_ln_segments,
                                      mvip_gemm_f16x3_ws_ln, mvip_layernorm_split_planes_stats), GroupNorm moments
                                      from the unsplit convolution's epilogue (mvip_conv3x3_f16x3_tile_moments) */

int         mvip_abi_version(void);
int         mvip_build_is_experiment(void); /* 1: compiled with a -DMVIP_EXPERIMENT_* macro (timing build, WRONG results) */
const char *mvip_strerror(int code);
const char *mvip_last_hip_error(void);      /* text of the last HIP error seen by this thread */
int         mvip_device_info(int *n_cu, int *lds_bytes, char *arch_out, int arch_cap);

/* ------------------------------------------------------------------------------------------
 * a1  get_rays                         DS_NeRF/run_nerf_helpers.py:249-260
 * c2w [3,4].  Writes the (y0..y0+h, x0..x0+w) window of the H x W frame (the `patch` crop of
 * DS_NeRF/run.py:1174-1177; pass 0,0,H,W for the full frame): rays_o, rays_d [h,w,3].
 */
int mvip_get_rays(const float *c2w, int H, int W, float focal, int y0, int x0, int h, int w,
                  float *rays_o, float *rays_d, void *stream);

/* a2  ray-row assembly of render()     DS_NeRF/run.py:1182-1207
 * rays_o, rays_d [B,3] -> rows [B,11] = (o, d, near, far, d/|d|).  viewdirs_src nullable: when
 * given ([B,3]) viewdirs come from it instead of rays_d (the c2w_staticcam case, run.py:1185). */
int mvip_ray_rows(const float *rays_o, const float *rays_d, const float *viewdirs_src,
                  float near, float far, int64_t B, float *rows, void *stream);

/* a1+a2 fused: rows straight from a pose; selects pixels by an optional int64 index list
 * (`sel` [B] flat y*W+x indices, nullable -> all H*W pixels in raster order).  Replaces
 * get_rays + the masked gather of DS_NeRF/run.py:869-884 + the row assembly. */
int mvip_ray_rows_from_pose(const float *c2w, int H, int W, float focal, float near, float far,
                            const int64_t *sel, int64_t B, float *rows, void *stream);

/* a3  stratified depths                DS_NeRF/run.py:1759-1781
 * rows [B,ncols] (near, far in columns 6,7); t_vals [S] = torch.linspace(0,1,S) (passed in so
 * it is bit-identical to torch's); t_rand [B,S] uniforms, nullable (perturb == 0). z [B,S]. */
int mvip_stratified_z(const float *rows, int ncols, int64_t B, int S, const float *t_vals,
                      int lindisp, const float *t_rand, float *z, void *stream);

/* a4  Embedder.embed                   DS_NeRF/run_nerf_helpers.py:22-52
 * x [N,3] -> y [N, 3+6L]  (x, then per octave: sin(x*2^k) xyz, cos(x*2^k) xyz). */
int mvip_posenc(const float *x, int64_t N, int L, float *y, void *stream);

/* ------------------------------------------------------------------------------------------
 * a5/a6  NeRF.forward behind run_network   DS_NeRF/run_nerf_helpers.py:104-127, run.py:1108-1124
 *
 * The 8x256 MLP (D=8, W=256, skip at 4, 63+27 encoded inputs, use_viewdirs) evaluated by ONE
 * kernel per call: encoding, all 11 linear layers and the activations stay on chip; weights are
 * streamed from a packed image (2.3 MB) through LDS.
 *
 * mvip_mlp_packed_floats(): size of the packed image in floats.
 * mvip_mlp_pack(): params_host is a HOST array of 24 device pointers in state-dict order
 *   pts_linears.{0..7}.{weight,bias}, views_linears.0.{weight,bias}, feature_linear.{weight,bias},
 *   alpha_linear.{weight,bias}, rgb_linear.{weight,bias}   (run_nerf_helpers.py:86-100).
 * mvip_mlp_unpack_grads(): the inverse mapping (packed-layout image -> 24 tensors).
 */
int64_t mvip_mlp_packed_floats(void);
int mvip_mlp_pack(const float *const *params_host, float *packed, void *stream);
/* image for precision 1; packed_f32 = the image mvip_mlp_pack() wrote for the same parameters
 * (its fp32 bias / head section is shared). */
int mvip_mlp_pack_f16x3(const float *const *params_host, float *image, const float *packed_f32,
                        void *stream);
int mvip_mlp_forward_rays_f16x3(const float *image, const float *rows, const float *z, int64_t B, int S,
                                float *raw, void *stream);
int mvip_mlp_forward_points_f16x3(const float *image, const float *pts, const float *dirs, int64_t P,
                                  float *raw, void *stream);

/* Round 5: the split-precision forward with TWO waves per SIMD (csrc/mlp_fwd16_f16x3.hip: 16 points per wave on
 * v_mfma_f32_16x16x32_f16, one weight ring per 8-wave workgroup) -- the kernel behind NeRF.forward under no_grad at
 * inference_precision = 1 (run_network, DS_NeRF/run.py:1108-1124; DS_NeRF/run_nerf_helpers.py:104-127).  Its own image
 * (mvip_mlp_packed_floats() floats: fp16 hi / lo A fragments in 16x16x32 order + section B of the fp32 image); values equal
 * mvip_mlp_forward_*_f16x3's up to fp32 summation order.  Forward only. */
int mvip_mlp_pack_f16x3_w16(const float *const *params_host, const float *packed_f32, float *image, void *stream);
int mvip_mlp_forward_rays_f16x3_w16(const float *image, const float *rows, const float *z, int64_t B, int S,
                                    float *raw, void *stream);
int mvip_mlp_forward_points_f16x3_w16(const float *image, const float *pts, const float *dirs, int64_t P,
                                      float *raw, void *stream);
/* ... and its stash-writing TRAINING form (train_precision = 1): raw + the activation stash of mvip_mlp_stash_floats(B*S) floats
 * that mvip_mlp_backward_stash(precision = 1) consumes (same layout and values as mvip_mlp_forward_rays_stash(precision = 1) writes,
 * up to fp32 summation order).  Renders under loss.backward(), DS_NeRF/run.py:948-974, :1030. */
int mvip_mlp_forward_rays_stash_f16x3_w16(const float *image, const float *rows, const float *z, int64_t B, int S,
                                          float *raw, float *stash, void *stream);

/* Forward from geometry: rows [B,11], z [B,S] -> raw [B,S,4]; points are o + d*z, view dirs
 * are rows[:,8:11] (run.py:1783, :1787).
 * precision 0: exact fp32 MFMA, `packed` from mvip_mlp_pack().
 * precision 1: "f16x3" split precision -- fp16 MFMA on W = Wh+Wl, X = Xh+Xl with the three leading
 *   products (Wh.Xh + Wh.Xl + Wl.Xh) accumulated in fp32; `packed` must be the image written by
 *   mvip_mlp_pack_f16x3() (same size).  Forward only; ~1e-6 relative deviation from precision 0. */
int mvip_mlp_forward_rays(const float *packed, const float *rows, const float *z, int64_t B, int S,
                          float *raw, int precision, void *stream);

/* Forward from explicit points: pts [P,3], dirs [P,3] (already normalised) -> raw [P,4]; this is
 * network_query_fn / run_network for arbitrary inputs (run.py:1530-1533). */
int mvip_mlp_forward_points(const float *packed, const float *pts, const float *dirs, int64_t P,
                            float *raw, int precision, void *stream);

/* Inference-only forward with two waves per SIMD (csrc/mlp_fwd16.hip): 16 points per wave on
 * v_mfma_f32_16x16x4_f32, exact fp32, same results to rounding as mvip_mlp_forward_* with precision 0.
 * packed16 (mvip_mlp_packed_floats() floats) = the same weights in the 16-point block order, built from the 24
 * parameter tensors and the small-vector section of an up-to-date mvip_mlp_pack image. */
int mvip_mlp_pack16(const float *const *params_host, const float *packed, float *packed16, void *stream);
int mvip_mlp_forward_rays16(const float *packed16, const float *rows, const float *z, int64_t B, int S,
                            float *raw, void *stream);
int mvip_mlp_forward_points16(const float *packed16, const float *pts, const float *dirs, int64_t P,
                              float *raw, void *stream);

/* ------------------------------------------------------------------------------------------
 * a10 (row g)  render_rays as TWO launches per chunk (DS_NeRF/run.py:1703-1847), no-grad renders of the native 8x256
 * networks with 64 coarse + <= 64 fine samples: the same device functions as the stand-alone entry points, fused behind
 * the network so that no raw / weights / depth tensor of the coarse pass and no z tensor of its samples ever exists.
 * Every output is BIT-IDENTICAL to the unfused chain (tests/test_render.py).
 * mvip_render_coarse_fused: rows [B,11] -> stratified depths (t_vals [64] = linspace(0,1,64); t_rand [B,64] or NULL) ->
 *   coarse network (packed16 of mvip_mlp_pack16) -> raw2outputs (noise [B,64] or NULL; flags MVIP_COMP_*) ->
 *   inverse-CDF resampling with Nf <= 64 uniforms (u [B,Nf], or one row of Nf when u_is_row) -> sort(cat[z, z_samples]).
 *   Outputs rgb0 [B,3], disp0 [B], acc0 [B], z_merged [B,64+Nf], z_std [B]; depth0 [B], weights0 [B,64], alpha0 [B,64]
 *   optional (NULL = not wanted).
 * mvip_render_fine_fused: network at the 128 depths z [B,128] -> raw2outputs; outputs as mvip_composite_forward plus
 *   raw [B,128,4] (NULL = not wanted), alpha optional. */
int mvip_render_coarse_fused(const float *packed16, const float *rows, int64_t B, const float *t_vals, int lindisp,
                             const float *t_rand, const float *noise, const float *u, int u_is_row, int Nf, int flags,
                             float *rgb0, float *disp0, float *acc0, float *depth0, float *weights0, float *alpha0,
                             float *z_merged, float *z_std, void *stream);
int mvip_render_fine_fused(const float *packed16, const float *rows, const float *z, int64_t B, const float *noise,
                           int flags, float *raw, float *rgb, float *disp, float *acc, float *depth, float *weights,
                           float *alpha, void *stream);

/* Backward: d_raw [P,4] -> the 24 parameter gradients.  grads_host is a HOST array of 24 device
 * pointers (state-dict order, natural [out][in] shapes) that are ACCUMULATED into with fp32
 * atomics (zero them, or pass .grad buffers).  Inputs are re-encoded and activations recomputed
 * on chip in tiles of `tile_points` (>=128); `workspace` must hold
 * mvip_mlp_backward_workspace_bytes(tile_points) bytes.  Gradients w.r.t. pts/dirs are not
 * produced (the reference never needs them: SURVEY.md 8b "Autograd"). */
int64_t mvip_mlp_backward_workspace_bytes(int64_t tile_points);
int mvip_mlp_backward_rays(const float *packed, const float *rows, const float *z, int64_t B, int S,
                           const float *d_raw, float *const *grads_host, void *workspace,
                           int64_t tile_points, int precision, void *stream);
int mvip_mlp_backward_points(const float *packed, const float *pts, const float *dirs, int64_t P,
                             const float *d_raw, float *const *grads_host, void *workspace,
                             int64_t tile_points, int precision, void *stream);
/* Training fast path: the forward also writes every activation to `stash`
 * (mvip_mlp_stash_floats(P) floats, ~9.9 KB per point) and the backward consumes it instead of
 * recomputing -- 2.9 instead of 3.9 forward-equivalents per training step when HBM has room
 * (288 GB: a 2.6 M-point step needs 26 GB). */
int64_t mvip_mlp_stash_floats(int64_t P);
int mvip_mlp_forward_rays_stash(const float *packed, const float *rows, const float *z, int64_t B,
                                int S, float *raw, float *stash, int precision, void *stream);
int mvip_mlp_forward_points_stash(const float *packed, const float *pts, const float *dirs, int64_t P,
                                  float *raw, float *stash, int precision, void *stream);
/* mvip_mlp_forward_rays_stash (precision 0) on the two-waves-per-SIMD kernel: `packed16` from mvip_mlp_pack16, the
 * stash layout and size are the same, so mvip_mlp_backward_stash (given the 32-point `packed` image) consumes it. */
int mvip_mlp_forward_rays_stash16(const float *packed16, const float *rows, const float *z, int64_t B, int S,
                                  float *raw, float *stash, void *stream);
int mvip_mlp_backward_stash(const float *packed, const float *stash, int64_t P, const float *d_raw,
                            float *const *grads_host, void *workspace, int64_t tile_points,
                            int precision, void *stream);
/* packed-layout image -> 24 tensors (inverse of mvip_mlp_pack; += when accumulate != 0). */
int mvip_mlp_unpack_grads(const float *grad_packed, float *const *grads_host, int accumulate,
                          void *stream);

/* ------------------------------------------------------------------------------------------
 * a7  raw2outputs                      DS_NeRF/run_nerf_helpers.py:350-404
 * raw [B,S,4], z [B,S], rows [B,ncols] (direction in columns 3..5), noise [B,S] nullable
 * (already multiplied by raw_noise_std).  Outputs: rgb [B,3], disp/acc/depth [B], weights [B,S],
 * alpha [B,S] nullable.  flags: bit0 white_bkgd.
 */
#define MVIP_COMP_WHITE   1
#define MVIP_COMP_DETACHW 2   /* detach_weights: no gradient from rgb_map into the weights */
int mvip_composite_forward(const float *raw, const float *z, const float *rows, int ncols,
                           const float *noise, int64_t B, int S, int flags, float *rgb, float *disp,
                           float *acc, float *depth, float *weights, float *alpha, void *stream);
/* Backward of the above.  Upstream gradients (each nullable = zero): g_rgb [B,3], g_disp, g_acc,
 * g_depth [B], g_weights [B,S], g_alpha [B,S].  Output d_raw [B,S,4]. */
int mvip_composite_backward(const float *raw, const float *z, const float *rows, int ncols,
                            const float *noise, int64_t B, int S, int flags, const float *g_rgb,
                            const float *g_disp, const float *g_acc, const float *g_depth,
                            const float *g_weights, const float *g_alpha, float *d_raw, void *stream);

/* a8+a9  sample_pdf + sort(cat)        DS_NeRF/run_nerf_helpers.py:304-347, run.py:1809-1816, :1836
 * z [B,Nc] coarse depths, weights [B,Nc] coarse weights (the kernel forms the Nc-1 midpoints
 * and uses weights[:,1:-1] itself), u [B,Nf] uniforms or, when u_is_row != 0, one row [Nf]
 * shared by all rays (det: torch.linspace(0,1,Nf)).  Outputs: z_samples [B,Nf], z_merged
 * [B,Nc+Nf] ascending, z_std [B] (population std of z_samples), inds int64 [B,Nf] nullable
 * (= #{cdf <= u}, searchsorted right=True), cdf [B,Nc-1] nullable.
 * z_std contract: two-pass (mean, then squared deviations) fp32 wave-tree sums; 1/Nf and the square root are
 * v_rcp_f32 / v_sqrt_f32 (<= 1 ulp each, 1/Nf exact for powers of two): within 3e-6 relative of
 * torch.std(z_samples, -1, unbiased=False) evaluated in fp64, not correctly rounded
 * (tests/test_hip_kernels.py::test_z_std_contract_non_power_of_two).
 * Nc = Nf = 64 (the reference configuration) runs two rays per wavefront; every output equals the one-ray-per-wavefront
 * kernel's bit for bit (a NaN equals a NaN), also for rows with NaN / negative / unsorted entries, which take the general route
 * (MVIP_SAMPLE_PAIR=0 selects the one-ray kernel for A/B; tests/test_hip_kernels.py::test_rays_per_wave_routes_are_bit_identical). */
int mvip_sample_pdf_merge(const float *z, const float *weights, const float *u, int u_is_row,
                          int64_t B, int Nc, int Nf, float *z_samples, float *z_merged,
                          float *z_std, int64_t *inds, float *cdf, void *stream);
/* Standalone sample_pdf for arbitrary bins/weights (bins [B,Nb], weights [B,Nb-1]). */
int mvip_sample_pdf(const float *bins, const float *weights, const float *u, int u_is_row,
                    int64_t B, int Nb, int Nf, float *samples, int64_t *inds, float *cdf,
                    void *stream);

/* ------------------------------------------------------------------------------------------
 * a12  depth2xyz_torch                 DS_NeRF/run.py:1909-1922
 * depth [H,W] -> points [H,W,3]: x=(w-cx)*z/fx, y=(h-cy)*z/fy, z.  Backward: g_points -> d_depth. */
int mvip_depth2xyz(const float *depth, int H, int W, float fx, float fy, float cx, float cy,
                   float *points, void *stream);
int mvip_depth2xyz_backward(const float *g_points, int H, int W, float fx, float fy, float cx,
                            float cy, float *d_depth, void *stream);

/* a12  depth2normal_geo                DS_NeRF/run.py:1924-1940
 * points PLANAR [3,H,W]; k odd window (31).  normals [3,H,W]: n = (A^T A)^-1 A^T 1 over the
 * zero-padded k x k window.  `moments` [9,H,W] receives the window sums (kept for the backward);
 * `scratch` is [9,H,W] for the forward and [18,H,W] for the backward. */
int mvip_normal_fit_forward(const float *points, int H, int W, int k, float *moments, float *scratch,
                            float *normals, void *stream);
int mvip_normal_fit_backward(const float *points, const float *moments, const float *normals,
                             const float *g_normals, int H, int W, int k, float *scratch,
                             float *d_points, void *stream);

/* ------------------------------------------------------------------------------------------
 * a13/a14  elementwise core of the SDS step   DS_NeRF/guidance/sd_utils.py:406-413, :29-37
 * latents = sqrt(abar)*x0 + sqrt(1-abar)*noise                       (scheduler.add_noise)
 * grad    = nan_to_num((1-abar) * (e_u + s*(e_c - e_u) - noise))     (CFG + SDS weight)
 * all [n] fp32. */
int mvip_sds_add_noise(const float *x0, const float *noise, float sqrt_abar, float sqrt_1m_abar,
                       int64_t n, float *latents, void *stream);
int mvip_sds_grad(const float *eps_uncond, const float *eps_cond, const float *noise,
                  float guidance_scale, float w, int64_t n, int accumulate, float *grad,
                  void *stream);

/* Posterior sample of the VAE encoder and its adjoint (the pipeline's _encode_vae_image: scaling_factor *
 * latent_dist.sample(), reached from DS_NeRF/guidance/sd_utils.py:207): moments [N][2C][HW] = (mean | logvar),
 * noise / out / d_out [N][C][HW], d_moments [N][2C][HW];  out = sf * (mean + exp(0.5 clamp(logvar, -30, 20)) * noise).
 * One launch each instead of the ~7 / ~15 elementwise launches of the torch expression and its autograd backward. */
int mvip_vae_sample(const float *moments, const float *noise, float scaling_factor, int64_t N, int64_t C,
                    int64_t HW, float *out, void *stream);
int mvip_vae_sample_backward(const float *moments, const float *noise, const float *d_out, float scaling_factor,
                             int64_t N, int64_t C, int64_t HW, float *d_moments, void *stream);

/* Sinusoidal timestep embedding in front of the UNet's time MLP (diffusers get_timestep_embedding with
 * flip_sin_to_cos=True; the unet(...) call of DS_NeRF/guidance/sd_utils.py:390-403): out [N][2 half] =
 * (cos(t_n f_k) | sin(t_n f_k)), t [N] on the device (graph-replayable), freqs [half] computed once by the caller. */
int mvip_timestep_sincos(const float *t, const float *freqs, int64_t N, int64_t half, float *out, void *stream);

/* The same two kernels with the timestep-dependent scalars read from device memory
 * (scal = {sqrt(abar), sqrt(1-abar), 1-abar}), so that ONE captured hipGraph of the SDS step
 * serves every timestep. */
int mvip_sds_add_noise_dev(const float *x0, const float *noise, const float *scal, int64_t n,
                           float *latents, void *stream);
int mvip_sds_grad_dev(const float *eps_uncond, const float *eps_cond, const float *noise,
                      float guidance_scale, const float *scal, int64_t n, int accumulate, float *grad,
                      void *stream);

/* ------------------------------------------------------------------------------------------
 * a14-a16  bilinear resize, align_corners = False: F.interpolate(pred_rgb, (512, 512), mode='bilinear') in front of
 * vae.encode (DS_NeRF/guidance/sd_utils.py:282-284, :449-452) and its adjoint.  x [planes, H, W] -> y [planes, OH, OW]
 * with torch's source-index convention (scale = in/out, src = scale*(dst+0.5)-0.5 clamped at 0); the backward gathers
 * (deterministic), dx [planes, H, W] = J^T dy. */
int mvip_resize_bilinear(const float *x, int64_t planes, int64_t H, int64_t W, int64_t OH, int64_t OW, float *y,
                         void *stream);
int mvip_resize_bilinear_backward(const float *dy, int64_t planes, int64_t H, int64_t W, int64_t OH, int64_t OW,
                                  float *dx, void *stream);

/* ------------------------------------------------------------------------------------------
 * a14-a16  GroupNorm (+ fused SiLU) of the SDS networks: the `norm -> silu -> conv` prologue of every
 * ResNet block inside vae.encode / unet (call sites DS_NeRF/guidance/sd_utils.py:148, :162, :189,
 * :212; the blocks themselves live in `diffusers`, absent from the reference tree -- published
 * SD-1.5 architecture, parity checked against torch.nn.functional.group_norm in fp32).
 * x, y, dy, dx: [N, C, HW] contiguous (NCHW); gamma, beta: [C] or NULL; dtype 0 = fp32, 1 = fp16
 * (storage type of x/y/dy/dx/gamma/beta; arithmetic fp32, statistics fp64).
 *   forward : y = act((x - mean_g) * rstd_g * gamma[c] + beta[c]),  act = SiLU if `silu` else identity;
 *             mean, rstd [N, G] fp32 are written for the backward.
 *   backward: dx only (the SDS networks are frozen; gradients flow to the rendered image).
 * workspace: mvip_groupnorm_workspace_bytes(N, C, HW) bytes, 16-byte aligned. */
int64_t mvip_groupnorm_workspace_bytes(int64_t N, int64_t C, int64_t HW);
int mvip_groupnorm_forward(const void *x, const void *gamma, const void *beta, int64_t N, int64_t C,
                           int64_t HW, int G, float eps, int silu, int dtype, void *y, float *mean,
                           float *rstd, void *workspace, void *stream);
int mvip_groupnorm_backward(const void *x, const void *dy, const void *gamma, const void *beta,
                            const float *mean, const float *rstd, int64_t N, int64_t C, int64_t HW, int G,
                            int silu, int dtype, void *dx, void *workspace, void *stream);
/* The same backward with the two passes that used to follow it folded in (the VAE encoder's data-gradient chain,
 * vae.encode under autograd at DS_NeRF/guidance/sd_utils.py:207): dx = backward(...) + dx_add (dx_add [N, C, HW] or NULL: the
 * gradient arriving over a ResNet block's identity shortcut, which autograd would add in a separate pass), and
 * maxima[mvip_groupnorm_backward_maxima(N, C, HW)] (or NULL) receives one max|dx| per workgroup (plain stores; NaN / Inf
 * skipped as mvip_absmax_scale skips them).  mvip_absmax_scale_from_maxima turns them into the scale2 = {s, 1/s} that
 * mvip_absmax_scale(dx) would compute -- the same power of two, without re-reading dx. */
int64_t mvip_groupnorm_backward_maxima(int64_t N, int64_t C, int64_t HW);
int mvip_groupnorm_backward_fused(const void *x, const void *dy, const void *gamma, const void *beta,
                                  const float *mean, const float *rstd, int64_t N, int64_t C, int64_t HW, int G,
                                  int silu, int dtype, const void *dx_add, void *dx, float *maxima, void *workspace,
                                  void *stream);
int mvip_absmax_scale_from_maxima(const float *maxima, int64_t count, float *scale2, void *stream);
/* statistics only (mean, rstd [N, G]); the normalised tensor is then produced by another kernel.  mean == rstd == NULL:
 * only the fp64 moment partials are left in `workspace`, for mvip_groupnorm_split_planes_moments. */
int mvip_groupnorm_stats(const void *x, int64_t N, int64_t C, int64_t HW, int G, float eps, int dtype,
                         float *mean, float *rstd, void *workspace, void *stream);

/* ------------------------------------------------------------------------------------------
 * a14-a16  3x3 / stride 1 / pad 1 convolution of the SDS networks (the ResNet-block convolutions inside
 * vae.encode / unet: DS_NeRF/guidance/sd_utils.py:148, :162, :189, :212; layers from the published SD-1.5
 * architecture) as an implicit GEMM on the fp16 matrix cores in split precision: both operands are
 * split into fp16 hi + lo terms and Wh.Xh + Wh.Xl + Wl.Xh is accumulated in fp32 (~1e-6 relative).
 *
 * Activations travel in "split planes": xs[N][Cin/16][2][2][H][W][8] fp16 (N*Cin*H*W*4 bytes), written
 * by mvip_groupnorm_split_planes (GroupNorm + optional SiLU fused) or mvip_split_planes (x * scale2[0]).
 * Weights are packed once per layer by mvip_conv3x3_pack into mvip_conv3x3_packed_bytes(Cout, Cin)
 * bytes; transpose = 1 packs the data-gradient operator (then the conv maps dY [N,Cout,H,W] -> dX
 * [N,Cin,H,W]: call mvip_conv3x3_f16x3 with Cin and Cout swapped).
 * scale2 = {s, 1/s, scratch, scratch} (device, 16 bytes): power-of-two scale from the absolute maximum
 * (mvip_absmax_scale; zero_words2 nullable: two caller-owned 32-bit scratch words, zero on entry and on exit, make it
 * ONE launch -- maximum by atomics, the last workgroup writes the scale -- instead of zero + reduce + scale).
 *   y[n,co,h,w] = conv(x, W)[n,co,h,w] / (s_w s_x) + bias[co] + chan_add[n,co] + residual[n,co,h,w]
 * (bias, chan_add, residual, x_scale2 may be NULL).  Supported: Cout % 32 == 0, Cin % 16 == 0,
 * (H % 8 == 0 and W % 32 == 0) or (H % 16 == 0 and W % 16 == 0) or H == W == 8 (mvip_conv3x3_supported); anything
 * else returns MVIP_EINVAL.
 * mvip_conv3x3_f16x3_ws takes a caller-owned workspace of mvip_conv3x3_workspace_bytes(N, Cin, Cout, H, W) bytes (0 for
 * most shapes; workspace may then be NULL): layers whose (pixel tile, 32-row block) grid would occupy a fraction of the
 * chip -- the UNet's 1280-channel levels at 16x16 and 8x8, whose weights are 59-118 MB per layer -- are split over the
 * input channels, each workgroup writing raw partial sums, and a second launch adds the splits in index order
 * (deterministic) and applies the epilogue.  Results equal mvip_conv3x3_f16x3's up to fp32 summation order. */
int mvip_conv3x3_supported(int64_t Cout, int64_t Cin, int64_t H, int64_t W);
int64_t mvip_conv3x3_packed_bytes(int64_t Cout, int64_t Cin);
int mvip_conv3x3_pack(const float *weight, int64_t Cout, int64_t Cin, int transpose, void *packed, void *stream);
/* Pack-time query (round 4): *two_product_host = 1 when every lo fragment of a packed WEIGHT image is zero, i.e. the
 * (power-of-two scaled) weights are exact fp16 values -- true for every weight the reference loads
 * (DS_NeRF/guidance/sd_utils.py:69-74: `revision="fp16"`, cast up to fp32 in the default mode).  Such an image may be
 * contracted with prec = 2.  fragment_bytes = the image's size without its 256-byte tail: Cout * Cin * 36
 * (mvip_conv3x3_pack) or M * K * 4 (mvip_gemm_pack_a).  Synchronises `stream` and reads one word back: not for the step. */
int mvip_packed_weights_two_product(const void *packed, int64_t fragment_bytes, int *two_product_host, void *stream);
int mvip_absmax_scale(const float *x, int64_t n, float *scale2, void *zero_words2, void *stream);
int mvip_split_planes(const float *x, int64_t N, int64_t C, int64_t HW, const float *scale2, void *xs,
                      int prec, void *stream);
int mvip_groupnorm_split_planes(const float *x, const float *gamma, const float *beta, const float *mean,
                                const float *rstd, int64_t N, int64_t C, int64_t HW, int G, int silu, void *xs,
                                int prec, void *stream);
/* mvip_groupnorm_split_planes reading the statistics from the moment partials of mvip_groupnorm_stats(mean = rstd = NULL)
 * (`moments` = that call's workspace): every workgroup reduces the partials of the <= 5 groups its 16 channels touch in
 * the order mvip_groupnorm_stats itself uses, so the planes are bit-identical -- one launch less per GroupNorm when
 * mean / rstd are not kept for a backward.  C / G >= 4. */
int mvip_groupnorm_split_planes_moments(const float *x, const float *gamma, const float *beta, const void *moments,
                                        float eps, int64_t N, int64_t C, int64_t HW, int G, int silu, void *xs,
                                        int prec, void *stream);
/* The same, also writing mean / rstd [N, G] (the bits mvip_groupnorm_stats itself would write) for a backward that needs them
 * (vae.encode under autograd, DS_NeRF/guidance/sd_utils.py:207): no separate finalize launch in the forward (round 4). */
int mvip_groupnorm_split_planes_moments_out(const float *x, const float *gamma, const float *beta, const void *moments,
                                            float eps, int64_t N, int64_t C, int64_t HW, int G, int silu, void *xs,
                                            float *mean_out, float *rstd_out, int prec, void *stream);
int mvip_conv3x3_f16x3(const void *xs, const void *packed, const float *bias, const float *chan_add,
                       const float *residual, const float *x_scale2, int64_t N, int64_t Cin, int64_t Cout,
                       int64_t H, int64_t W, float *y, int prec, void *stream);
int64_t mvip_conv3x3_workspace_bytes(int64_t N, int64_t Cin, int64_t Cout, int64_t H, int64_t W);
/* Channel-split launches can leave the GroupNorm moments of their OUTPUT (the norm -> silu -> conv chains of the ResNet
 * blocks): mvip_conv3x3_f16x3_ws_moments's reduction launch also writes `row_moments` [N][Cout][2] fp64 = {sum, sum of
 * squares} of every output row -- the layout mvip_groupnorm_stats(mean = rstd = NULL) leaves in its workspace for
 * H * W <= 4096, so mvip_groupnorm_split_planes_moments(y, ..., row_moments, ...) needs no pass over y for its statistics.
 * mvip_conv3x3_row_moments_doubles: N * Cout * 2 when the shape qualifies (channel-split launch; H * W = 64, 256, 1024 or
 * 4096), else 0 (then only mvip_conv3x3_f16x3_ws applies).  y is bit-identical to mvip_conv3x3_f16x3_ws's. */
int64_t mvip_conv3x3_row_moments_doubles(int64_t N, int64_t Cin, int64_t Cout, int64_t H, int64_t W);
int mvip_conv3x3_f16x3_ws_moments(const void *xs, const void *packed, const float *bias, const float *chan_add,
                                  const float *residual, const float *x_scale2, int64_t N, int64_t Cin, int64_t Cout,
                                  int64_t H, int64_t W, float *y, void *workspace, void *row_moments, int prec,
                                  void *stream);
/* Unsplit launches (mvip_conv3x3_workspace_bytes == 0, images wider than 8 pixels): the same moments from partials the
 * convolution's epilogue leaves per (image, channel, pixel tile, wave) -- fp32 sums of 64 finished values, added in fp64 in
 * index order by a second, small launch (N * Cout rows) -- instead of a pass over y (DS_NeRF/guidance/sd_utils.py:330-352,
 * :390-403: the resnet chains of the VAE encoder and the UNet).  tile_scratch: mvip_conv3x3_tile_moments_scratch_bytes(...)
 * bytes (0 = this shape cannot: use mvip_conv3x3_f16x3_ws / _ws_moments); moments: mvip_groupnorm_workspace_bytes(N, Cout,
 * H * W) bytes in mvip_groupnorm_stats' workspace layout (what mvip_groupnorm_split_planes_moments reads). */
int64_t mvip_conv3x3_tile_moments_scratch_bytes(int64_t N, int64_t Cin, int64_t Cout, int64_t H, int64_t W);
int mvip_conv3x3_f16x3_tile_moments(const void *xs, const void *packed, const float *bias, const float *chan_add,
                                    const float *residual, const float *x_scale2, int64_t N, int64_t Cin, int64_t Cout,
                                    int64_t H, int64_t W, float *y, void *tile_scratch, void *moments, int prec,
                                    void *stream);
int mvip_conv3x3_f16x3_ws(const void *xs, const void *packed, const float *bias, const float *chan_add,
                          const float *residual, const float *x_scale2, int64_t N, int64_t Cin, int64_t Cout,
                          int64_t H, int64_t W, float *y, void *workspace, int prec, void *stream);

/* The same split-precision machinery as a plain GEMM (1x1 convolutions and the products of the VAE
 * mid-block attention, vae.encode at DS_NeRF/guidance/sd_utils.py:207):
 *   Y[n][m][p] = sum_k A[m][k] X[n][k][p] / (s_a s_x) + bias[m] + chan_add[n][m] + residual[n][m][p]
 * A[m][k] = src[m*sm + k*sk] (src a dense block of M*K floats) is packed by mvip_gemm_pack_a into
 * mvip_gemm_packed_bytes(M, K) bytes; X travels as split planes [N][K/16][2][2][P][8], written by
 * mvip_split_planes_strided from x[n*sn + k*sc + p*sp] (times scale2[0] if given).
 * M % 32 == 0, K % 32 == 0, P % 256 == 0. */
int64_t mvip_gemm_packed_bytes(int64_t M, int64_t K);
int mvip_gemm_pack_a(const float *src, int64_t M, int64_t K, int64_t sm, int64_t sk, void *packed, void *stream);
int mvip_split_planes_strided(const float *x, int64_t N, int64_t C, int64_t HW, int64_t sn, int64_t sc, int64_t sp,
                              const float *scale2, void *xs, int prec, void *stream);
/* The same planes for the 2x nearest-neighbour up-sampled image [N][C][2H][2W] of x [N][C][H][W] (Upsample2D of the UNet:
 * F.interpolate(scale_factor=2, mode='nearest') in front of a 3x3 convolution), without materialising it. */
int mvip_split_planes_upsample2(const float *x, int64_t N, int64_t C, int64_t H, int64_t W, const float *scale2, void *xs,
                                int prec, void *stream);
/* Row softmax P = softmax(scale * S) over the last axis of S [rows][cols] (cols <= 8192) and its adjoint
 * dS = scale * P * (dP - rowsum(dP * P)): the score matrix of AutoencoderKL's single-head mid-block attention inside
 * vae.encode (DS_NeRF/guidance/sd_utils.py:207) and its data gradient. */
int mvip_softmax_rows(const float *s, int64_t rows, int64_t cols, float scale, float *p, void *stream);
int mvip_softmax_rows_backward(const float *p, const float *dp, int64_t rows, int64_t cols, float scale, float *ds,
                               void *stream);
/* General convolution as this GEMM -- the stride-2 down-samplers of the UNet and the VAE encoder and the layers with 3,
 * 4, 8 or 9 channels (conv_in, conv_out, quant_conv; DS_NeRF/guidance/sd_utils.py:207, :240), which do not fit the 3x3
 * stride-1 kernel's operand tiles:
 *   X_col[k = ci*KH*KW + ky*KW + kx][p = oy*OW + ox] = x[n][ci][oy*stride + ky - pad_top][ox*stride + kx - pad_left]
 * is written directly as split planes [N][KP/16][2][2][PP][8] (KP >= Cin*KH*KW a multiple of 32, PP >= OH*OW a multiple
 * of 256, zero filled), so y = mvip_gemm_f16x3(xs, pack(weight as [Cout][Cin*KH*KW]), ...) with K = KP, P = PP.
 * Data gradient: col[N][KP][PP] = mvip_gemm_f16x3 with the transposed weight on dY's split planes, then
 * mvip_col2im gathers dx[n][ci][iy][ix] (deterministic). */
int mvip_im2col_split_planes(const float *x, int64_t N, int64_t Cin, int64_t H, int64_t W, int KH, int KW, int stride,
                             int pad_top, int pad_left, int64_t OH, int64_t OW, int64_t KP, int64_t PP,
                             const float *scale2, void *xs, int prec, void *stream);
int mvip_col2im(const float *col, int64_t N, int64_t Cin, int64_t H, int64_t W, int KH, int KW, int stride, int pad_top,
                int pad_left, int64_t OH, int64_t OW, int64_t KP, int64_t PP, float *dx, void *stream);
int mvip_gemm_f16x3(const void *xs, const void *packed, const float *bias, const float *chan_add,
                    const float *residual, const float *x_scale2, int64_t N, int64_t K, int64_t M, int64_t P,
                    float *y, int prec, void *stream);
/* The feed-forward's first projection with the GEGLU in the epilogue (the [N][2R][P] intermediate never exists):
 * out [N][R][P] = (W_v x + b_v) * gelu(W_g x + b_g) for columns < L, zero beyond; `packed` / `bias` hold the M2 = 2R
 * rows interleaved in 32-row tiles (value rows of tile t, then the gate rows of tile t); M2 % 64 == 0.  scale2 and
 * zero_word as in mvip_geglu. */
int mvip_gemm_geglu_f16x3(const void *xs, const void *packed, const float *bias, const float *x_scale2, int64_t N,
                          int64_t K, int64_t M2, int64_t P, int64_t L, float *out, float *scale2, void *zero_word,
                          int prec, void *stream);
/* the same with the workgroup tile forced (timing switch; identical arithmetic per output element up to the k order
 * inside a stage, which is the same): cfg 0 = by shape (what mvip_gemm_f16x3 does), 1 = 32/64 rows x 256 columns,
 * 2 = 128 x 256 (M % 128 == 0), 3 = 128 x 128 (M % 128 == 0), 4 = 64 x 128 (M % 64 == 0). */
int mvip_gemm_f16x3_cfg(const void *xs, const void *packed, const float *bias, const float *chan_add,
                        const float *residual, const float *x_scale2, int64_t N, int64_t K, int64_t M, int64_t P,
                        float *y, int cfg, int prec, void *stream);
/* mvip_gemm_f16x3 with a caller-owned workspace of mvip_gemm_workspace_bytes(N, K, M, P) bytes (0 for most shapes;
 * workspace may then be NULL): launches with fewer workgroups than CUs and a long contraction (the UNet's 1280-channel
 * transformer blocks at 16x16 and 8x8) are split over K, each workgroup writing raw partial sums, and a second launch
 * adds the splits in index order and applies the epilogue. */
int64_t mvip_gemm_workspace_bytes(int64_t N, int64_t K, int64_t M, int64_t P);
int mvip_gemm_f16x3_ws(const void *xs, const void *packed, const float *bias, const float *chan_add,
                       const float *residual, const float *x_scale2, int64_t N, int64_t K, int64_t M, int64_t P,
                       float *y, void *workspace, int prec, void *stream);
/* The same GEMM leaving, besides y, the LayerNorm statistics of y over its M rows (the unet(...) call,
 * DS_NeRF/guidance/sd_utils.py:390-403: BasicTransformerBlock's norm1 / norm2 / norm3 read the residual stream a
 * projection has just written): ln_part[((n * S + g) * 2 + {0, 1}) * P + p] = fp64 sum / sum of squares of the finished
 * rows of segment g for column p, S = mvip_gemm_ln_segments(N, K, M, P, prec) segments of M / S rows each
 * (N * S * 2 * P doubles).  mvip_gemm_ln_segments returns 0 for shapes whose launch cannot leave them (split-K
 * launches, the square-tile kernel); mvip_gemm_f16x3_ws_ln then returns MVIP_EINVAL and the caller keeps
 * mvip_layernorm_split_planes' own statistics pass. */
int64_t mvip_gemm_ln_segments(int64_t N, int64_t K, int64_t M, int64_t P, int prec);
int mvip_gemm_f16x3_ws_ln(const void *xs, const void *packed, const float *bias, const float *chan_add,
                          const float *residual, const float *x_scale2, int64_t N, int64_t K, int64_t M, int64_t P,
                          float *y, void *workspace, void *ln_part, int prec, void *stream);

/* ------------------------------------------------------------------------------------------
 * a14-a16  transformer blocks of the SD UNet (unet(...) at DS_NeRF/guidance/sd_utils.py:390-403 and :240;
 * the block structure is the published SD-1.5 one: LayerNorm -> 8-head self-attention -> LayerNorm ->
 * cross-attention onto the 77 prompt tokens -> LayerNorm -> GEGLU feed-forward).  Activations stay
 * channel-major [N][C][LP] (LP = tokens padded to a multiple of 256), so every linear layer is
 * mvip_gemm_f16x3 with the weight as the A operand and no layout transposes; these entry points produce
 * its operands and run the attention itself, all in split precision (fp16 hi/lo operands, three products,
 * fp32 accumulation: fp32-grade results).
 *
 * mvip_attention_f16x3: flash-style multi-head attention, nothing of size [Lq, Lk] touches memory.
 *   qs, ks : split planes [N][heads*NCH][2][2][Lq | LkP][8] (mvip_split_planes_strided), NCH = ceil(D/16),
 *            head h = chunks h*NCH.., channels beyond D zero; scaled by q_scale2[0] / k_scale2[0];
 *   vp     : mvip_attention_pack_v output (mvip_attention_v_bytes bytes), scaled by v_scale2[0];
 *   out    : fp32 [N][heads*D][LqP], columns < Lq written;
 *   out[n, h*D + d, i] = sum_j softmax_j(softmax_scale q_i . k_j)[j < Lk] v_j[d].
 *   D in {40, 80, 160} (mvip_attention_supported), Lq % 32 == 0, LkP % 64 == 0, LkP >= Lk.
 *   flags (tuning switches, same results): bit 0 = 64-key LDS tiles for D = 40, bit 1 = never use the 256-query
 *   workgroup.
 * mvip_attention_pack_v: v[n*sn + (h*DP + d)*sr + key*sk], d < D <= DP, key < Lk -> A fragments whose k order
 *   is the accumulator-row order of the score tile (so the probabilities feed the second product from
 *   registers), zero padded to LkP.
 * mvip_absmax_scale_sections: x [outer][sections][len] -> scale2[s] = {2^k, 2^-k, -, -} per section
 *   (|x|max 2^k in [2^9, 2^10)), sections <= 63; ONE launch for the q / k / v thirds of a fused projection (the last
 *   workgroup writes the scales).
 *   zero_words64: 64 caller-owned 32-bit scratch words that are ZERO on entry and left zero on exit (allocate
 *   zeroed once per stream; the maxima are collected there with atomics, which saves a zeroing launch per use). */
int mvip_attention_supported(int64_t D);
int64_t mvip_attention_v_bytes(int64_t N, int64_t heads, int64_t D, int64_t LkP);
int mvip_attention_pack_v(const float *v, int64_t N, int64_t heads, int64_t D, int64_t DP, int64_t Lk, int64_t LkP,
                          int64_t sn, int64_t sr, int64_t sk, const float *scale2, void *vp, int prec, void *stream);
int mvip_absmax_scale_sections(const float *x, int64_t outer, int64_t sections, int64_t len, float *scale2,
                               void *zero_words64, void *stream);
int mvip_attention_f16x3(const void *qs, const void *ks, const void *vp, const float *q_scale2, const float *k_scale2,
                         const float *v_scale2, int64_t N, int64_t heads, int64_t D, int64_t Lq, int64_t LqP,
                         int64_t Lk, int64_t LkP, float softmax_scale, int flags, float *out, int prec, void *stream);
/* `prec` (round 3) on the operand producers and the contractions of this section and the next: 0 = split precision
 * ("f16x3": fp16 hi + lo halves of both operands, three products, fp32 accumulate -- fp32-grade results, the default of
 * the fp32 networks); 1 = the reference's --fp16 mode (DS_NeRF/guidance/sd_utils.py:66, DS_NeRF/run.py:251): ONE fp16
 * product per step, fp32 accumulate -- producers write the hi planes only, contractions fetch hi planes / hi weight
 * fragments only (a third of the matrix work, half the operand traffic; ~1e-3 relative, fp16-grade).  Operand buffers have
 * the same size and layout in both modes.
 * 2 (round 4; the CONTRACTIONS that take a packed weight image: mvip_conv3x3_f16x3*, mvip_gemm_f16x3*,
 * mvip_gemm_geglu_f16x3*, mvip_gemm_f16x3_sinks, mvip_gemm_f16x3_planes_ws) = TWO products, Wh.Xh + Wh.Xl: the caller
 * asserts that the image's lo fragments are zero (mvip_packed_weights_two_product said 1).  The third product of prec 0
 * would add exact zeros, so every output bit equals prec 0's, with a third less matrix work and half the weight-operand
 * bytes.  Activations are split and fetched as for prec 0 (producers take 0 or 1, and treat 2 as 0); a launch shape
 * without a two-product instantiation silently runs the three-product kernel (same result).  With prec = 1 a shape
 * without a single-product instantiation returns MVIP_EUNSUP (its lo planes were never written). */
/* Contractions that hand each other OPERANDS (round 3; same call sites: the unet(...) call of
 * DS_NeRF/guidance/sd_utils.py:390-403 / :240).  Between two contractions of a transformer block the reference
 * materialises an fp32 tensor; rounds 1-2 of this library followed it with an absolute-maximum pass and a split pass.
 * Here the PRODUCER's epilogue writes the consumer's operand format, scaled by a power of two that the caller fixes
 * BEFORE the launch from a rigorous bound of the result (|W x| <= |x|max * max_row ||W||_1; |LayerNorm(x)| <=
 * sqrt(C) |gamma|max + |beta|max; |softmax(..) V| <= |V|max; |a gelu(g)| <= |a| |g|): no overflow by construction, and the
 * fp16 hi + lo pair keeps ~2^-25 of the scaled range, i.e. fp32-grade accuracy relative to the tensor's maximum for
 * bounds up to ~2^15 too wide.
 * mvip_gemm_f16x3_sinks: Y = W X + bias with the M rows cut into nsec <= 3 consecutive sections of sec_rows[i] rows
 *   (multiples of 64).  Section i leaves as (Y * sec_scale[i]) in format sec_kind[i]:
 *     1  split planes [N][sec_rows/16][2][2][P][8 halves] -- the B operand of mvip_gemm_f16x3 / the Q or K operand of
 *        mvip_attention_f16x3_sink;
 *     2  attention V fragments [N][heads][v_dt][P/16][2][64][8 halves], sec_rows = heads * v_dt * 32 (head h's rows at
 *        h * v_dt * 32 .., zero rows beyond its D channels); only as the LAST section (computed with the MFMA operands
 *        swapped, i.e. transposed, in a launch of its own).
 * mvip_gemm_geglu_f16x3_sink: mvip_gemm_geglu_f16x3 whose product leaves as planes [N][(M2/2)/16][2][2][P][8] * out_scale.
 * mvip_attention_f16x3_sink: mvip_attention_f16x3 on operands written by mvip_gemm_f16x3_sinks (q_stride / k_stride tokens
 *   per plane, v_groups 16-key groups per V block) whose result leaves as the output projection's operand planes
 *   [N][heads*D/16][2][2][LqP][8 halves] scaled by v_scale2[0]; columns >= Lq are not written. */
int mvip_gemm_f16x3_sinks(const void *xs, const void *packed, const float *bias, const float *x_scale2, int64_t N,
                          int64_t K, int64_t M, int64_t P, int nsec, const int64_t *sec_rows, const int *sec_kind,
                          void *const *sec_ptr, const float *sec_scale, int v_dt, int prec, void *stream);
/* (W X + bias + residual) * out_scale as ONE section of split planes [N][M/16][2][2][P][8 halves]: the second
 * feed-forward projection handing the finished residual stream to proj_out as its operand.  Split-K as in
 * mvip_gemm_f16x3_ws (workspace of mvip_gemm_workspace_bytes(N, K, M, P) bytes, NULL when that is 0); M % 32 == 0. */
int mvip_gemm_f16x3_planes_ws(const void *xs, const void *packed, const float *bias, const float *residual,
                              const float *x_scale2, int64_t N, int64_t K, int64_t M, int64_t P, void *out_planes,
                              float out_scale, void *workspace, int prec, void *stream);
int mvip_gemm_geglu_f16x3_sink(const void *xs, const void *packed, const float *bias, const float *x_scale2, int64_t N,
                               int64_t K, int64_t M2, int64_t P, int64_t L, void *out_planes, float out_scale,
                               int prec, void *stream);
int mvip_attention_f16x3_sink(const void *qs, const void *ks, const void *vp, const float *q_scale2, const float *k_scale2,
                              const float *v_scale2, int64_t N, int64_t heads, int64_t D, int64_t Lq, int64_t LqP,
                              int64_t Lk, int64_t LkP, int64_t q_stride, int64_t k_stride, int64_t v_groups,
                              float softmax_scale, int flags, void *out_planes, int prec, void *stream);
/* LayerNorm over the channel axis of x [N][C][LP] for tokens < L, times out_scale (a power of two), written as
 * split planes [N][C/16][2][2][LP][8] (zero for tokens >= L).  C % 64 == 0, LP % 256 == 0; workspace of
 * mvip_layernorm_workspace_bytes(N, C, LP) bytes (fp64 partial moments), 8-byte aligned. */
int64_t mvip_layernorm_workspace_bytes(int64_t N, int64_t C, int64_t LP);
int mvip_layernorm_split_planes(const float *x, const float *gamma, const float *beta, int64_t N, int64_t C,
                                int64_t L, int64_t LP, float eps, float out_scale, void *workspace, void *xs,
                                int prec, void *stream);
/* The normalise + split half alone, from statistics `part` = [N][segments][2][LP] fp64 (sum, sum of squares) over
 * `segments` disjoint channel ranges covering C -- what mvip_gemm_f16x3_ws_ln leaves: one launch, no pass over x for
 * its moments.  C % 16 == 0. */
int mvip_layernorm_split_planes_stats(const float *x, const float *gamma, const float *beta, const void *part,
                                      int64_t segments, int64_t N, int64_t C, int64_t L, int64_t LP, float eps,
                                      float out_scale, void *xs, int prec, void *stream);
/* GEGLU: y [N][2R][LP] -> out [N][R][LP] = y[:, :R] * gelu(y[:, R:]) (erf form) for tokens < L, zero beyond;
 * scale2 = {2^k, 2^-k, -, -} from the result's absolute maximum; zero_word: one scratch word as in
 * mvip_absmax_scale_sections (zero on entry, zero on exit). */
int mvip_geglu(const float *y, int64_t N, int64_t R, int64_t L, int64_t LP, float *out, float *scale2, void *zero_word,
               void *stream);
/* y [NB][M] = act(x [NB][K]) W[M][K]^T + b  (act_in: 0 identity, 1 SiLU), NB <= 8: the timestep-embedding MLP and
 * the ResNet blocks' time projections (published SD-1.5 UNet), exact fp32, one wavefront per output feature. */
int mvip_linear_small(const float *x, const float *W, const float *b, int64_t NB, int64_t M, int64_t K, int act_in,
                      float *y, void *stream);
/* The same for several layers sharing the input x (the UNet's 22 ResNet time projections all read silu(temb)): W [M][K]
 * and b [M] hold the layers' rows one after the other, layer_off (n_layers + 1 device ints) their first rows, and layer
 * l's result is its own contiguous [NB][C_l] block at y + layer_off[l] * NB. */
int mvip_linear_small_grouped(const float *x, const float *W, const float *b, int64_t NB, int64_t M, int64_t K, int act_in,
                              const int *layer_off, int64_t n_layers, float *y, void *stream);

/* ------------------------------------------------------------------------------------------
 * encodings of the reference's second model NeRF_TCNN
 * (DS_NeRF/run_nerf_helpers_tcnn.py:36-46 hash grid, :63-69 spherical harmonics, :91-101 forward), which
 * the reference takes from tiny-cuda-nn (absent from the reference tree; published algorithm restated).
 * x [P,3] raw coordinates, mapped to [0,1] by (x + bound)/(2 bound) when bound > 0; table [n_entries,2];
 * levels [16][4] 32-bit words {scale (fp32 bits), resolution, offset, size} in DEVICE memory;
 * features / d_features [32][P] level-major (row 2l+f = feature f of level l); d_table is accumulated
 * into (zero it first).  mvip_sh4: dirs [P,3] in [-1,1] -> out [16][P]. */
int mvip_hashgrid_forward(const float *x, const float *table, const void *levels, int64_t P, float bound,
                          float *features, void *stream);
int mvip_hashgrid_backward(const float *x, const float *d_features, const void *levels, int64_t P, float bound,
                           float *d_table, void *stream);
int mvip_sh4(const float *dirs, int64_t P, float *out, void *stream);
/* Opt-in table gradient with tiny-cuda-nn's own arithmetic for the scattered contributions (half-precision pair
 * atomics): scale2 = {s, 1/s} from mvip_absmax_scale(d_features); d_table (fp32) and d_table_h2 ([n_entries] fp16
 * pairs) zeroed by the caller; gradient = d_table + d_table_h2 * (16 * scale2[1]). */
int mvip_hashgrid_backward_half2(const float *x, const float *d_features, const void *levels, int64_t P, float bound,
                                 const float *scale2, float *d_table, void *d_table_h2, void *stream);

/* Fused no-grad NeRF_TCNN.forward (run_nerf_helpers_tcnn.py:88-112): x [P,3], dirs [P,3] -> raw [P,4] =
 * (colour 0..2, sigma), in exact fp32 on the matrix pipe.  sigma_params = [64x32 | 16x64] floats, colour_params =
 * [64x32 | 64x64 | 16x64] floats ([out][in] row-major, layer order = the tiny-cuda-nn `params` vectors);
 * mvip_hashgrid_mlp_pack writes the kernel's operand image (mvip_hashgrid_mlp_packed_floats() floats). */
int64_t mvip_hashgrid_mlp_packed_floats(void);
int mvip_hashgrid_mlp_pack(const float *sigma_params, const float *colour_params, float *img, void *stream);
int mvip_hashgrid_nerf_forward(const float *x, const float *dirs, const float *table, const void *levels,
                               const float *img, int64_t P, float bound, float *raw, void *stream);

/* Weight gradient of one bias-free layer of that model's MLPs (what autograd derives for the reference's
 * tcnn.Network calls, run_nerf_helpers_tcnn.py:96-108): dW[M][N] = sum_p dY[m][p] X[n][p], operands channel-major
 * [C][P], 1 <= M, N <= 64, P % 64 == 0.  Writes mvip_skinny_wgrad_slabs(P) partial [M][N] slabs; their sum in
 * index order is dW. */
int64_t mvip_skinny_wgrad_slabs(int64_t P);
int mvip_skinny_wgrad(const float *dY, const float *X, int64_t M, int64_t N, int64_t P, float *slabs, void *stream);
/* Forward / data gradient of the same layers: Y[M][P] = act(W X), W[m][n] = w[m*w_sm + n*w_sn] (the data gradient passes
 * the same weight with the two strides swapped), X [N][P] channel-major, 1 <= M, N <= 64, P % 4 == 0, X / Y 16-byte
 * aligned; relu != 0 applies max(., 0) (the activation of tcnn's hidden layers).  Exact fp32 on the matrix pipe, one
 * streaming pass: (N + M) x 4 bytes per point. */
int mvip_skinny_linear(const float *w, int64_t w_sm, int64_t w_sn, const float *X, int64_t M, int64_t N, int64_t P,
                       int relu, float *Y, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MVIP_NERF_H */
