/*
 * nerf_sampling_hip.h -- C ABI of libnerf_sampling_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for ONE hot path of MarcinKadziolka/nerf-sampling: the ray-batch render
 * operator (ray generation, ray-sphere intersection, positional encoding, DepthNet and
 * radiance-field MLP forward, alpha compositing).  The reference has no native code, so
 * each entry point below names the reference *Python* function (file:line relative to the
 * reference tree) whose arithmetic it replaces; a ctypes binding for every one of them is in
 * nerf_sampling_amd/_lib.py and the reference-side stub is shown in INTEGRATION.md.
 *
 * Conventions
 *   - every pointer marked "dev" is a device (HBM) pointer, fp32 unless stated; caller-owned
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls are
 *     asynchronous on it and re-entrant across streams (weight handles are read-only after packing and
 *     own no per-call scratch; per-call intermediates live in the caller's workspace)
 *   - return value: 0 = ok, negative = NS_E_* below (nothing was launched), never throws
 *   - no torch types, no global state except lazily-queried device properties
 */
#ifndef NERF_SAMPLING_HIP_H
#define NERF_SAMPLING_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NS_OK 0
#define NS_E_INVALID (-1)     /* bad argument (null pointer, non-positive size, ...)   */
#define NS_E_UNSUPPORTED (-2) /* network shape outside what the kernels are built for  */
#define NS_E_HIP (-3)         /* a HIP runtime call failed; see ns_last_error()        */
#define NS_E_NOMEM (-4)

/* operand precision of the MFMA kernels (accumulation is always fp32) */
#define NS_DTYPE_F32 0  /* v_mfma_f32_32x32x2_f32, exact-fp32 parity path (k-major engine)             */
#define NS_DTYPE_BF16 1 /* v_mfma_f32_16x16x32_bf16, both networks (output-sub-block-major engine)      */
#define NS_DTYPE_F16 2  /* v_mfma_f32_16x16x32_f16, same engine                                         */
#define NS_DTYPE_F16X3 3 /* split fp16 operands (x = hi + lo), three v_mfma_f32_16x16x32_f16 per product term: fp32-grade
                          * results (the fp32 parity gates hold) at ~1/3 of the fp16 rate; operands must fit fp16's
                          * range (|w| < 65504, checked at pack time)                                    */
#define NS_DTYPE_F16M 4  /* DepthNet only, the production 10 x 256 trunk: MIXED fp16 operands -- the first
                          * NS_F16M_SPLIT_LAYERS trunk layers split as in F16X3 (that is where an fp16 DepthNet loses its depth:
                          * 90 % of the error's variance), the rest plain F16; 1.6 x the fp16 kernel's MFMAs instead of 3 x */
#define NS_F16M_SPLIT_LAYERS 3

/* sample placement modes, utils.py:220-244 */
#define NS_MODE_DEPTH_ONLY 0
#define NS_MODE_UNIFORM 1
#define NS_MODE_GAUSSIAN 2

const char* ns_last_error(void);
int ns_version(void);
/* number of compute units of the current device (0 if no device) */
int ns_device_cu_count(void);
/* Diagnostic switches of the kernel dispatch (tests compare code paths inside one process; no effect on results):
 *   "generic_kernels" 0/1 -- 1: the production network (8 x 256, skips = [4]) runs the generic compiler-scheduled kernels
 *                            instead of the generated instruction streams;
 *   "prod_tiles" 0/4/5    -- tiles per wave of the 16-bit production kernel, 0 = chosen per launch;
 *   "hier_chain" 0/1      -- 1: ns_render_rays_hierarchical keeps raw [R,N,4] in HBM and composites with ns_raw2outputs instead
 *                            of in the MLP kernels' epilogues.
 * Initial values come from the environment (NS_OB16_GENERIC, NS_OB16_TILES), read once at first use.              */
int ns_debug_set(const char* name, int value);

/* ---- a1  get_rays + prepare_rays  (run_nerf_helpers.py:187-202, nerf_utils.py:156-188) ----
 * Pixel rows [row0,row1) of an HxW pinhole camera, row-major.  c2w is 12 HOST floats (3x4,
 * row-major).  Any output pointer may be NULL.  ray_batch is [R,11] =
 * [o(3) d(3) near far viewdir(3)], viewdir = d/|d|; R = (row1-row0)*W.                      */
int ns_get_rays(int H, int W, float fx, float fy, float cx, float cy, const float* c2w_host,
                int row0, int row1, float near_, float far_, float* rays_o_dev, float* rays_d_dev,
                float* viewdirs_dev, float* ray_batch_dev, void* stream);

/* ---- a2  find_intersection_points_with_sphere / solve_quadratic_equation (utils.py:159-217)
 * t [R,2] (minus-sqrt root first) and points [R,2,3]; NaN where the line misses the sphere.  */
int ns_sphere_intersect(const float* o_dev, const float* d_dev, int64_t R, float radius,
                        float* t_dev, float* pts_dev, void* stream);
/* element-wise quadratic roots: out [2,n] */
int ns_solve_quadratic(const float* a_dev, const float* b_dev, const float* c_dev, int64_t n,
                       float* out_dev, void* stream);

/* ---- a3  Embedder.embed (run_nerf_helpers.py:15-63): x [M,d] -> [M, d*(1+2L)] ------------- */
int ns_posenc(const float* x_dev, int64_t M, int d, int n_freqs, float* out_dev, void* stream);

/* ---- packed network weights ------------------------------------------------------------
 * Host-side packers: take the reference state-dict tensors (HOST fp32, row-major
 * [out,in] weights as nn.Linear stores them) and build the device-resident MFMA weight
 * stream.  Handles own device memory; destroy with ns_weights_destroy.                      */
typedef struct ns_weights ns_weights; /* opaque */

/* NeRF(D, W, input_ch=63, input_ch_views=27, skips=[skip], use_viewdirs=True)
 * (run_nerf_helpers.py:67-134).  w/b: arrays of D + 4 host pointers in the order
 * pts_linears.0..D-1, feature_linear, alpha_linear, views_linears.0, rgb_linear.
 * skip = index i after which the embedded input is re-concatenated (4), or -1 for none.
 * feature_linear (no activation) is composed with views_linears.0 at pack time, in fp64.   */
int ns_pack_nerf(int D, int W, int skip, const float* const* w, const float* const* b, int dtype,
                 ns_weights** out);
/* The full signature of the reference's NeRF (run_nerf_helpers.py:67-134): any `skips` list (bit i of skip_mask set
 * <=> i in skips; every skip must precede the last layer, D <= 32) and both head variants.  use_viewdirs != 0: as
 * ns_pack_nerf (w/b: D + 4 tensors; raw has 4 channels whatever output_ch says, :126-131).  use_viewdirs == 0: the
 * head is output_linear (W -> output_ch, :132-133; w/b: D + 1 tensors, pts_linears.0..D-1 then output_linear), the
 * network takes no view directions, and raw has output_ch channels (create_nerf builds 5 with N_importance > 0,
 * nerf_utils.py:405-406).  W: any width from 2 to 256 (netwidth, nerf_utils.py:409-423): the kernels are instantiated
 * for 128 and 256 and the packer zero-pads every tensor to the next of the two -- a padded unit is relu(0) = 0 and feeds
 * zero columns, so the real units sum the same products plus exact zeros.                                           */
int ns_pack_nerf_ex(int D, int W, uint32_t skip_mask, int use_viewdirs, int output_ch,
                    const float* const* w, const float* const* b, int dtype, ns_weights** out);
/* channels of the raw output of a packed NeRF (4, or output_ch without view directions) */
int ns_nerf_out_channels(const ns_weights* net);
/* DepthNet(hidden_sizes, cat_hidden_sizes, multires=10) (depth_net.py:10-169).  w/b: 3*n_branch + n_trunk + 1
 * host pointers in the order origin_layers.0.., direction_layers.0.., intersection_layers.0.., cat_layers.0,2,..,
 * to_depth.0.  hidden_sizes [n_branch]: widths of the three skip branches (any); cat_sizes [n_trunk]: trunk widths,
 * each <= 256 (class defaults [128]*6 / [128,128,128,128,256] and the production 10 x 256 both qualify).
 * The skip branches are affine (the reference constructs nn.LeakyReLU(x) and never applies it, depth_net.py:140,148,
 * 156), so the packer composes them with the first trunk layer, in fp64, into one 252 -> cat_sizes[0] layer; trunk
 * layers are zero-padded to one width in {128, 256}.                                                          */
int ns_pack_depthnet_ex(int n_branch, const int* hidden_sizes, int n_trunk, const int* cat_sizes,
                        const float* const* w, const float* const* b, int dtype, ns_weights** out);
/* the uniform case: hidden_sizes = cat_hidden_sizes = [width] * n_layers */
int ns_pack_depthnet(int n_layers, int width, const float* const* w, const float* const* b,
                     int dtype, ns_weights** out);
/* The pack-time folds on their own (HOST only, no device touched; tests check them against the literal chain):
 * F_out [c0, 252] and bias_out [c0] of the folded DepthNet front end on cat[gamma(o), gamma(d), gamma(isect)]
 * (w/b: the 3*n_branch branch tensors followed by cat_layers.0);                                              */
int ns_fold_depthnet_front(int n_branch, const int* hidden_sizes, int c0, const float* const* w,
                           const float* const* b, float* F_out, float* bias_out);
/* w_out [W/2, W+27], b_out [W/2] of views_linears[0] o feature_linear (run_nerf_helpers.py:119-125)             */
int ns_fold_nerf_views(int W, const float* w_feature, const float* b_feature, const float* w_views,
                       const float* b_views, float* w_out, float* b_out);
void ns_weights_destroy(ns_weights* w);
/* bytes of the device weight stream (for roofline accounting) */
int64_t ns_weights_stream_bytes(const ns_weights* w);

/* ---- a4  DepthNet.forward (depth_net.py:117-169): (o,d) [R,3] -> z [R] in [near,far] -------- */
int ns_depthnet_forward(const ns_weights* net, const float* o_dev, const float* d_dev, int64_t R,
                        float near_, float far_, float sphere_radius, float* z_dev, void* stream);

/* ---- a5  sample_points_around_mean (utils.py:220-244) ---------------------------------------
 * mean [R]; noise [R,N-1] standard-normal draws (GAUSSIAN only, else NULL); outputs z [R,N]
 * and pts [R,N,3] (either may be NULL).  DEPTH_ONLY ignores N (treated as 1).               */
int ns_place_samples(int mode, const float* o_dev, const float* d_dev, const float* mean_dev,
                     const float* noise_dev, int64_t R, int N, float std_, float* pts_dev,
                     float* z_dev, void* stream);

/* ---- a6+a7  Trainer.run_network + NeRF.forward (Trainer.py:789-806, helpers :67-134) --------
 * pts [R,N,3], viewdirs [R,3] -> raw [R,N,C] = (rgb pre-sigmoid, sigma pre-relu[, ...]), C = ns_nerf_out_channels(net)
 * (4 for every network with view directions).  viewdirs_dev is ignored (may be NULL) for a network packed with
 * use_viewdirs == 0.  If pts_dev is NULL the points are formed in-kernel as o + d*z from o,d [R,3], z [R,N].     */
int ns_nerf_forward(const ns_weights* net, const float* pts_dev, const float* o_dev,
                    const float* d_dev, const float* z_dev, const float* viewdirs_dev, int64_t R,
                    int N, float* raw_dev, void* stream);
/* NeRF.forward on an already embedded input x [M,90] (x [M,63] without view directions) -> [M,C] */
int ns_nerf_forward_embedded(const ns_weights* net, const float* x_dev, int64_t M, float* raw_dev,
                             void* stream);

/* ---- a8  raw2alpha + DepthNetTrainer.raw2outputs (nerf_utils.py:27-42, sampling_trainer.py
 * :153-230).  raw [R,N,4], z [R,N], rays_d [R,3]; noise [R,N] already multiplied by
 * raw_noise_std, or NULL.  Outputs (any may be NULL): rgb [R,3], disp/acc/depth [R],
 * alphas/weights [R,N].  (density is raw[...,3], a view, and has no output here.)
 * N == 1 reproduces the reference exactly: its dists/alphas/weights are then EMPTY ([R,0]), so
 * rgb = sigmoid(raw rgb), acc = depth = 0, disp = 1e10 and alphas/weights are not written.    */
int ns_raw2outputs(const float* raw_dev, const float* z_dev, const float* rays_d_dev,
                   const float* noise_dev, int64_t R, int N, int white_bkgd, float* rgb_dev,
                   float* disp_dev, float* acc_dev, float* depth_dev, float* alphas_dev,
                   float* weights_dev, void* stream);
/* the same with the per-ray rgb / disp outputs at caller-chosen strides (in floats; rgb of ray r at
 * rgb_dev + r*rgb_stride, >= 3; disp at disp_dev + r*disp_stride, >= 1): lets a renderer write straight
 * into an interleaved [R,4] = (r,g,b,disp) frame shard, the unit the multi-GPU all-gather moves
 * (no reference counterpart: the reference returns separate tensors and has no multi-GPU path)      */
int ns_raw2outputs_strided(const float* raw_dev, const float* z_dev, const float* rays_d_dev,
                           const float* noise_dev, int64_t R, int N, int white_bkgd, float* rgb_dev,
                           int64_t rgb_stride, float* disp_dev, int64_t disp_stride, float* acc_dev,
                           float* depth_dev, float* alphas_dev, float* weights_dev, void* stream);

/* ---- a11 vanilla hierarchical pieces (Trainer.py:579-710, run_nerf_helpers.py:250-293) ------ */
/* stratified coarse depths: near/far [R]; t_rand [R,N] uniform draws or NULL (perturb==0)    */
int ns_coarse_z(const float* near_dev, const float* far_dev, int64_t R, int N, int lindisp,
                const float* t_rand_dev, float* z_dev, void* stream);
/* inverse-CDF sampling: bins [R,Nb], weights [R,Nb-1], u [R,Nf] or NULL (= linspace(0,1,Nf)) */
int ns_sample_pdf(const float* bins_dev, const float* weights_dev, int64_t R, int Nb, int Nf,
                  const float* u_dev, float* samples_dev, void* stream);
/* z_mid + sample_pdf(weights[1:-1]) + sort(cat[z, samples]) in one pass: z [R,Nc],
 * weights [R,Nc] -> z_out [R,Nc+Nf] ascending (Trainer.py:672-685)                         */
int ns_importance_z(const float* z_dev, const float* weights_dev, int64_t R, int Nc, int Nf,
                    const float* u_dev, float* z_out_dev, void* stream);
/* ascending sort of every row of x [R,N] (torch.sort(x,-1).values), N <= 2048 */
int ns_sort_rows(const float* x_dev, int64_t R, int N, float* out_dev, void* stream);
/* pts = o + d*z : o,d [R,3], z [R,N] -> [R,N,3] */
int ns_points_along_rays(const float* o_dev, const float* d_dev, const float* z_dev, int64_t R,
                         int N, float* pts_dev, void* stream);
/* per-ray argmax of weights [R,N] and the gathered z / weight / sigmoid(raw rgb)
 * (nerf_utils.py:813-819); any output may be NULL                                           */
int ns_argmax_gather(const float* weights_dev, const float* z_dev, const float* raw_dev, int64_t R,
                     int N, float* max_z_dev, float* max_w_dev, float* max_rgb_dev, void* stream);

/* ---- a9  the DepthNet branch of render_rays_test as one call (nerf_utils.py:836-865) --------
 * rays come either from (o,d,viewdirs) device arrays or, if o_dev is NULL, are generated in
 * place from the camera (rows [row0,row1) of an HxW image).  Runs DepthNet -> placement ->
 * NeRF MLP -> compositing on `stream` with intermediates in the caller-provided workspace.
 * Outputs rgb [R,3], disp [R] always; z/weights/pts (per-sample extras) only if non-NULL.   */
typedef struct ns_render_args {
  const ns_weights* depthnet;
  const ns_weights* nerf;
  /* rays: explicit ... */
  const float* o_dev;
  const float* d_dev;
  const float* viewdirs_dev;
  int64_t R;
  /* ... or camera (used when o_dev == NULL) */
  int H, W, row0, row1;
  float fx, fy, cx, cy;
  float c2w[12];
  /* sampling set-up */
  int mode;    /* NS_MODE_* */
  int N;       /* trainer.n_depth_samples */
  float std_;  /* trainer.distance */
  const float* noise_dev; /* [R,N-1], GAUSSIAN only */
  float near_, far_, sphere_radius;
  int white_bkgd;
  /* workspace: ns_render_workspace_bytes(R,N) bytes of device memory */
  void* workspace_dev;
  /* outputs */
  float* rgb_dev;
  float* disp_dev;
  float* z_dev;       /* [R,N] or NULL */
  float* weights_dev; /* [R,N] or NULL */
  float* pts_dev;     /* [R,N,3] or NULL */
  /* optional hipEvent_t pair recorded on `stream` immediately before / after the NeRF-MLP kernel
   * (the dominant kernel), so a harness can time it inside its own timed region; NULL = none   */
  void* ev_mlp_begin;
  void* ev_mlp_end;
  /* strides (in floats) of the rgb / disp outputs; 0 = packed (3 / 1).  rgb_stride = disp_stride = 4 with
   * disp_dev = rgb_dev + 3 writes an interleaved [R,4] shard directly                                  */
  int64_t rgb_stride;
  int64_t disp_stride;
  /* PSNR guard (optional, NS_MODE_UNIFORM only; NULL = off).  The reference composites the LAST sample of a ray with
   * dist = 1e10 (sampling_trainer.py:176-180): alpha_last = step(sigma_last), so with 16-bit operands a sigma_last near
   * zero flips a ray's colour.  With a second handle of the SAME network packed NS_DTYPE_F16X3 (fp32-grade) here, the
   * last sample of every ray is evaluated a second time through it (R of the R*N samples, ~5 % of the frame) and its
   * sigma replaces the 16-bit one before compositing.  Pair it with an F16X3 DepthNet handle for fp32-grade depths.  */
  const ns_weights* nerf_guard;
  /* 0: the guard re-evaluates the last sample of EVERY ray.  > 0 (ns_render_rays_fused, N <= 64): only of the rays whose own
   * 16-bit sigma of that sample lies within this distance of zero -- the only ones whose step can flip: a sigma beyond it
   * composites to alpha = 0 or 1 either way -- found by the kernel itself, re-evaluated afterwards on a compacted list
   * (three small launches; the count never leaves the device) and their pixels re-added from the kernel's partial sums, bit
   * for bit what the every-ray guard gives wherever |sigma16 - sigma32| < guard_threshold.  The five-launch chain
   * (ns_render_rays_depthnet) ignores it and guards every ray.                                                            */
  float guard_threshold;
} ns_render_args;
int64_t ns_render_workspace_bytes(int64_t R, int N);
int ns_render_rays_depthnet(const ns_render_args* args, void* stream);
/* The same operator as ONE kernel per ray tile (the reference's chain nerf_utils.py:836-865):
 * rays -> DepthNet -> [sample placement + radiance-field MLP + raw2outputs in one persistent kernel].  Sample depths are
 * evaluated in-kernel from the DepthNet depth (no z array), raw stays in the CU and is composited in the MLP kernel's
 * epilogue by the same wave scan ns_raw2outputs runs: rgb / disp (and z / weights / pts when asked for) are BIT-IDENTICAL
 * to ns_render_rays_depthnet.  Three launches per call (ray generation, DepthNet, the fused kernel); HBM traffic is the
 * rays, 4 B + 16 B per ray and the weight streams.  Supported (ns_render_fused_supported != 0): mode NS_MODE_UNIFORM,
 * a bf16 / f16 NeRF handle with view directions, N a power of two in [2, 64] or a multiple of 64 up to 512 (a ray is then
 * several 64-sample chunks composited side by side); anything else returns NS_E_UNSUPPORTED and
 * is served by ns_render_rays_depthnet.  Workspace: ns_render_fused_workspace_bytes(R) bytes, 256-byte aligned.        */
int ns_render_fused_supported(const ns_weights* nerf, int mode, int N);
int64_t ns_render_fused_workspace_bytes(int64_t R);
int ns_render_rays_fused(const ns_render_args* args, void* stream);

/* ---- a11 as one call: sample_as_in_NeRF (nerf_utils.py:497-611) = coarse pass, inverse-CDF importance
 * sampling, sorted merge, fine pass.  Rays explicit or generated from the camera (o_dev == NULL).
 * Outputs of the FINE pass: rgb [R,3], disp [R] always; z / weights [R,Nc+Nf] and raw [R,Nc+Nf,4] if
 * non-NULL.  t_rand [R,Nc] (stratified jitter) and u [R,Nf] (inverse-CDF draws) are NULL for the
 * deterministic perturb == 0 path.                                                               */
typedef struct ns_hier_args {
  const ns_weights* coarse;
  const ns_weights* fine; /* NULL: the coarse network is used for both passes */
  const float* o_dev;
  const float* d_dev;
  const float* viewdirs_dev;
  int64_t R;
  int H, W, row0, row1;
  float fx, fy, cx, cy;
  float c2w[12];
  int Nc, Nf;
  int lindisp, white_bkgd;
  float near_, far_;
  const float* t_rand_dev;
  const float* u_dev;
  void* workspace_dev; /* ns_hier_workspace_bytes(R, Nc, Nf) bytes, 256-byte aligned */
  float* rgb_dev;
  float* disp_dev;
  float* z_dev;
  float* weights_dev;
  float* raw_dev;
  void* ev_mlp_begin; /* optional hipEvent_t pair around the FINE-pass MLP kernel */
  void* ev_mlp_end;
  int64_t rgb_stride; /* as in ns_render_args; 0 = packed */
  int64_t disp_stride;
  void* ev_coarse_begin; /* optional hipEvent_t pair around the COARSE-pass MLP kernel */
  void* ev_coarse_end;
} ns_hier_args;
int64_t ns_hier_workspace_bytes(int64_t R, int Nc, int Nf);
int ns_render_rays_hierarchical(const ns_hier_args* args, void* stream);

/* ---- training step of the DepthNet (Trainer.core_optimization_loop, Trainer.py:506-544) -------------
 * Batches are N_rand = 1024 rays, so these are small fp32 kernels (exact-fp32 MFMA products), not the
 * fused inference path.  One strided GEMM serves nn.Linear forward (y = x W^T + b), grad-input
 * (dx = dy W) and grad-weight (dW = dy^T x):
 *   C[i,j] (+)= sum_k A[i*sa0 + k*sa1] * B[j*sb0 + k*sb1] (+ bias[j]),  C row-major with leading dim ldc */
int ns_gemm_strided(const float* A_dev, int64_t sa0, int64_t sa1, const float* B_dev, int64_t sb0,
                    int64_t sb1, const float* bias_dev, float* C_dev, int64_t ldc, int M, int N, int K,
                    int accumulate, void* stream);
/* The same GEMM with fused epilogues (a training step is launch-bound: every elementwise kernel beside a GEMM costs as much
 * as a small GEMM): act != 0 applies an activation to the result (0 none, 1 relu, 2 leaky 0.01, 3 sigmoid: a forward
 * layer); dact != 0 multiplies the result by the derivative of activation `dact` evaluated from its OUTPUT
 * dact_ref[i*ld_ref + j] (the grad-input GEMM of the layer above carries the backward through this layer's activation);
 * a_rowsum [M], if non-NULL, receives sum_k A[i,k] (the bias gradient when A = dy^T, i.e. of the grad-weight GEMM).      */
int ns_gemm_fused(const float* A_dev, int64_t sa0, int64_t sa1, const float* B_dev, int64_t sb0,
                  int64_t sb1, const float* bias_dev, float* C_dev, int64_t ldc, int M, int N, int K,
                  int accumulate, int act, int dact, const float* dact_ref_dev, int64_t ld_ref,
                  float* a_rowsum_dev, void* stream);
/* Up to four such GEMMs in ONE launch (the three skip branches of the DepthNet run the same layer side by side); all
 * problems of a launch share their operand layout (sa1 == 1 or not, sb1 == 1 or not).  problems_host: HOST array.     */
typedef struct ns_gemm_problem {
  const float* A_dev; int64_t sa0, sa1;
  const float* B_dev; int64_t sb0, sb1;
  const float* bias_dev;
  float* C_dev; int64_t ldc;
  int M, N, K;
  int accumulate, act, dact;
  const float* dact_ref_dev; int64_t ld_ref;
  float* a_rowsum_dev;
} ns_gemm_problem;
int ns_gemm_fused_batched(const ns_gemm_problem* problems_host, int count, void* stream);
/* out[j] = sum_i X[i*ld + j]  (bias gradient) */
int ns_colsum(const float* X_dev, int64_t ld, int M, int N, float* out_dev, void* stream);
/* activations in place: act 0 none, 1 ReLU, 2 LeakyReLU(0.01), 3 sigmoid; backward scales dy by act'(.)
 * evaluated from the activation OUTPUT y                                                            */
int ns_act_forward(float* y_dev, int64_t n, int act, void* stream);
int ns_act_backward(float* dy_dev, const float* y_dev, int64_t n, int act, void* stream);
/* gradient of Embedder.embed w.r.t. its input: x [M,d], de [M,d(1+2L)] -> dx [M,d] */
int ns_posenc_backward(const float* x_dev, const float* de_dev, int64_t M, int d, int n_freqs,
                       float* dx_dev, void* stream);
/* gradient of pts = o + d*z w.r.t. z: dpts [R,N,3], d [R,3] -> dz [R,N] */
int ns_points_backward(const float* dpts_dev, const float* d_dev, int64_t R, int N, float* dz_dev,
                       void* stream);
/* torch.optim.Adam update (no weight decay / amsgrad), step counted from 1 */
int ns_adam_step(float* p_dev, const float* g_dev, float* m_dev, float* v_dev, int64_t n, float lr,
                 float beta1, float beta2, float eps, int step, void* stream);
/* the same update with the step count read from *step_dev (>= 1) and, when lr_dev is not NULL, the learning rate from
 * *lr_dev: nothing in the launch changes from step to step, so a captured hipGraph of the whole training step
 * (forward, backward, update) can be replayed; ns_add_i32 advances the counter on the stream                       */
int ns_adam_step_dev(float* p_dev, const float* g_dev, float* m_dev, float* v_dev, int64_t n, float lr,
                     const float* lr_dev, float beta1, float beta2, float eps, const int* step_dev,
                     void* stream);
int ns_add_i32(int* x_dev, int delta, void* stream);
/* ns_adam_step_dev for EVERY parameter tensor of the optimiser in one launch: table_dev = n_tensors rows {p, g, m, v, n}
 * in device memory, max_n = the largest n (82 launches per DepthNet step become one)                                  */
typedef struct ns_adam_tensor {
  float* p;
  const float* g;
  float* m;
  float* v;
  int64_t n;
} ns_adam_tensor;
int ns_adam_step_multi_dev(const ns_adam_tensor* table_dev, int n_tensors, int64_t max_n, float lr,
                           const float* lr_dev, float beta1, float beta2, float eps, const int* step_dev,
                           void* stream);

/* ---- timing helpers (hipEvent_t as void*) used by bench.py for the live roofline figure --------- */
int ns_event_create(void** ev);
void ns_event_destroy(void* ev);
int ns_event_record(void* ev, void* stream);
/* work submitted to `stream` after this call waits for `ev` (hipStreamWaitEvent): lets a copy stream start a frame
 * chunk's device-to-host copies exactly when the NEXT chunk's NeRF-MLP kernel starts (ev = its ev_mlp_begin)      */
int ns_stream_wait_event(void* stream, void* ev);
/* milliseconds between two recorded events; blocks until `end` has completed */
int ns_event_elapsed_ms(void* begin, void* end, float* ms);

#ifdef __cplusplus
}
#endif
#endif /* NERF_SAMPLING_HIP_H */
