/* libsvolsdf_hip.so -- C-ABI of the MI355X-native S-VolSDF volume-rendering hot path.
 *
 * The reference (cvlab-stonybrook/s-volsdf) is pure Python: it has no FFI of its own, so each entry point
 * below names the reference function (file:line, relative to the reference root) whose arithmetic it
 * replaces.  The binding a maintainer adds on the reference side is the ctypes stub shown in INTEGRATION.md
 * (s-volsdf_amd/svs_hip/lib.py is that stub, s-volsdf_amd/volsdf/ the drop-in classes that call it).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to float32 / int32 data unless stated otherwise; tensors are
 *     contiguous row-major; `hip_stream` is a hipStream_t (NULL = default stream);
 *   - calls are asynchronous (stream-ordered); none allocates, frees or synchronises; all outputs and
 *     workspaces are caller-allocated (size queries: *_bytes);
 *   - return 0 on success, < 0 for an argument/shape error, > 0 = hipError_t of a failed launch;
 *     svs_last_error_string() describes the last failure on the calling thread.
 */
#ifndef SVOLSDF_HIP_H
#define SVOLSDF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI version.  101 (round 6): svs_set_deterministic / svs_get_deterministic, svs_conv2d_mfma*, svs_featurenet_fpn2 added.  100 -> 101 also marks the argument
 * lists that changed during round 5 without a version signal: svs_sdf_bwd_a lost `a2max`, svs_sdf_bwd_b gained `ubuf`, and in
 * fp16x2 svs_sdf_outputs writes records BEHIND gbuf's 8 blocks -- size gbuf with svs_sdf_gbuf_bytes(), not svs_sdf_hbuf_bytes().
 * A host binding should refuse a library whose version differs from the one it was written against (svs_hip/lib.py does). */
int svs_version(void);
const char* svs_last_error_string(void);

/* Deterministic accumulation of the weight gradients (process-wide, off by default; returns the previous setting).
 * The reference's CPU path (torch autograd on one thread, volsdf/vsdf.py:21) gives the same gradient for the same inputs
 * every time; svs_wgrad / svs_wgrad_multi / svs_lin8_row0_grad add their workgroups' partial sums with float atomics, in
 * arrival order.  With the switch on, the workgroups that add into one accumulator do so in launch order (csrc/svs_ticket.h):
 * one fixed summation order, bit-identical results run to run, at the cost of serialised flushes.  Launches that add into the
 * same accumulator must then be ordered by the caller (one stream). */
int svs_set_deterministic(int on);
int svs_get_deterministic(void);

/* ---- a1  rays ---------------------------------------------------------------------------------------
 * rend_util.get_camera_params + lift (volsdf/utils/rend_util.py:60-95,143-156) and the depth_scale of
 * VolSDFNetwork.forward (volsdf/model/network.py:213-217).
 * uv (n_rays,2) pixel (x,y); pose (4,4) camera-to-world; intrinsics (4,4).
 * -> ray_dirs (n_rays,3) unit, cam_loc (3), depth_scale (n_rays) */
int svs_rays_from_uv(const float* uv, const float* pose, const float* intrinsics, int n_rays, float* ray_dirs,
                     float* cam_loc, float* depth_scale, void* hip_stream);

/* The eikonal points of a train-mode forward (volsdf/model/network.py:258-266; BG model: network_bg.py:128-138):
 * points[0..n_rays) = uniform_points (the host's uniform draws in the bounding sphere, (n_rays,3)),
 * points[n_rays + r] = cam_loc + z_eik[r] * ray_dirs[r] (the sampler's extra depth per ray, ray_sampler.py:200-208).
 * points: DEVICE float[2 n_rays][3], the tail of the point list the fused SDF launch evaluates. */
int svs_eikonal_points(const float* uniform_points, const float* cam_loc, const float* z_eik, const float* ray_dirs,
                       int n_rays, float* points, void* hip_stream);
/* The small per-step inputs of VolOpt.train_step (volsdf/vsdf.py:196-204: `model_input[...].cuda()`, and the train-mode
 * draws of the sampler): `bytes` (a multiple of 4) from a PINNED HOST buffer (hipHostMalloc / torch pin_memory: mapped
 * into the device's address space) to device memory by a kernel on hip_stream; both 16-byte aligned.  The host must not
 * rewrite the buffer before the launch has run (the callers keep a ring of staging buffers and an event per slot). */
int svs_stage_in(const void* pinned_host, void* device, size_t bytes, void* hip_stream);
/* BG model (volsdf/model/network_bg.py:60-62): z (n_rays, n) -> head (n_rays, n - 1) = z[:, :-1] dense, last (n_rays) =
 * z[:, -1] (the sphere exit depth), one launch. */
int svs_split_last(const float* z, int n_rays, int n, float* head, float* last, void* hip_stream);

/* ---- a5/a6  weight packing --------------------------------------------------------------------------
 * Weight-norm materialisation w = g*v/||v|| (nn.utils.weight_norm, volsdf/model/network.py:64-65) and the
 * permutation into the MFMA consumption order.  weight_v/weight_g/bias: HOST arrays of 9 (SDF) / 5 (radiance)
 * device pointers in layer order, shapes as in the checkpoint (`implicit_network.lin{l}.weight_v` ...);
 * weight_g == NULL for networks without weight-norm.  workspace: svs_pack_workspace_bytes().
 * which: 0 SDF forward, 1 SDF full (forward + feature head + input-gradient pass), 2 SDF training backward,
 *        3 radiance forward, 4 radiance backward (5..8: the background networks, see below), 9 SDF forward for
 *        svs_sdf_vals16 (fp16x2 only; experimental build).  Like every entry point, svs_pack_stream allocates and copies
 * nothing on its own: the stream's chunk table (<= 2.2 KB) travels in the kernel arguments.
 * precision: how the kernels that consume the stream evaluate the layer products (the same value is passed to them):
 *   SVS_MMA_F32   v_mfma_f32_32x32x2_f32 on float32 operands (exact float32 products);
 *   SVS_MMA_F16X2 v_mfma_f32_32x32x16_f16 on operands split into two fp16 pieces, a = hi + mid, three products
 *                 per term (22-bit significands, float32 accumulation: float32-class accuracy at 2.8x the speed;
 *                 operands must stay below 65504) -- forward AND backward: the activation blocks the training backward
 *                 keeps in memory hold both pieces, parameter gradients agree with float64 autograd to < 1e-5 of a
 *                 tensor's largest entry (the float32-MFMA kernels: 3e-6);
 *   SVS_MMA_F16X2_HALF  the same kernels with the gradient-only blocks of the backward (ghat, u, a2, abar, zbar, fbar)
 *                 stored as ONE fp16 piece under a per-point scale and one-product weight-gradient GEMMs: half the
 *                 backward's block bytes, parameter gradients 2e-4 ... 8e-4 off (a mixed-precision training mode; the
 *                 forward is identical).  Packs the same streams as SVS_MMA_F16X2.
 * Stream sizes do not depend on the precision. */
enum { SVS_MMA_F32 = 0, SVS_MMA_F16X2 = 1, SVS_MMA_F16X2_HALF = 2 };
size_t svs_stream_bytes(int which);
int svs_pack_stream(int which, int precision, const float* const* weight_v, const float* const* weight_g,
                    const float* const* bias, float* workspace, float* stream_out, void* hip_stream);
size_t svs_pack_workspace_bytes(void);

/* ---- a5  SDF MLP ------------------------------------------------------------------------------------
 * Sample positions of a launch = the ray samples cam + z*dir (n_rays x S, row-major, may be 0 rays) followed by
 * n_points explicit points (may be 0).  cam_stride 0 = one camera centre, 3 = one per ray.
 *
 * svs_sdf_vals: ImplicitNetwork.get_sdf_vals (volsdf/model/network.py:125-131) -> sdf (P).
 *   sphere clamp min(sdf, scale*(radius-|x|)) on points [0, clamp_n) (clamp_n < 0: all) when radius > 0.
 *   gate: optional device flags, one per group of gate_points points (a multiple of 128; <= 0: one group), gate_stride
 *   ints apart: the points of a group are skipped when its flag is 0 (sampler rounds, one flag per convergence
 *   group of rays). */
int svs_sdf_vals(const float* points, int n_points, const float* cam, int cam_stride, const float* dirs, const float* z,
                 int S, int n_rays, const float* stream, int precision, float sphere_radius, float sphere_scale,
                 int clamp_n, float* sdf, const int* gate, int gate_points, int gate_stride, void* hip_stream);
#ifdef SVS_EXPERIMENTAL_KERNELS   /* built only with SVS_BUILD_EXPERIMENTS=1 (s-volsdf_amd/build.py); not part of the default library */
/* The same evaluation by the 16-point-wave kernel (two waves per SIMD, v_mfma_f32_16x16x32_f16, csrc/svs_mlp_w16.hip):
 * `stream` packed with which = 9; fp16x2 arithmetic; results agree with svs_sdf_vals to float32 rounding of the
 * accumulation order. */
int svs_sdf_vals16(const float* points, int n_points, const float* cam, int cam_stride, const float* dirs, const float* z,
                   int S, int n_rays, const float* stream, float sphere_radius, float sphere_scale, int clamp_n, float* sdf,
                   const int* gate, int gate_points, int gate_stride, void* hip_stream);
/* ... and by the K-split-pair kernel (two waves per SIMD on the same 32 points, csrc/svs_mlp_h2p.hip): the stream of
 * svs_sdf_vals (which = 0 or 1, fp16x2).  Both variants are experiments kept for A/B runs. */
int svs_sdf_vals_pair(const float* points, int n_points, const float* cam, int cam_stride, const float* dirs, const float* z,
                      int S, int n_rays, const float* stream, float sphere_radius, float sphere_scale, int clamp_n, float* sdf,
                      const int* gate, int gate_points, int gate_stride, void* hip_stream);
#endif
/* svs_sdf_outputs: ImplicitNetwork.get_outputs (network.py:105-123) and .gradient (:90-103):
 *   sdf (P), grad = d sdf/dx (P,3), feat_tiles (svs_feat_tiles_bytes; wave-tile layout, may be NULL),
 *   hbuf (svs_sdf_hbuf_bytes): the activations h_1..h_8 kept for the gradient pass / training backward;
 *   gbuf (svs_sdf_gbuf_bytes, may be NULL): ghat_l = g(h_{l+1}) * softplus'(a_l), l = 0..7, of the gradient pass, and -- fp16x2
 *   precisions -- behind the 8 blocks of every tile their records [1.0 x 32][max_r |ghat_l| of point 0..31] (the layout of
 *   "Records" below), which svs_sdf_bwd_b reads; clamp_mask (P bytes, may be NULL): 1 where the sphere clamp is active -- both
 *   are only needed by the training backward. */
size_t svs_sdf_hbuf_bytes(int n_points_total);
size_t svs_feat_tiles_bytes(int n_points_total);
int svs_sdf_outputs(const float* points, int n_points, const float* cam, int cam_stride, const float* dirs,
                    const float* z, int S, int n_rays, const float* stream, int precision, float sphere_radius,
                    float sphere_scale, int clamp_n, float* sdf, float* grad, float* feat_tiles, float* hbuf,
                    float* gbuf, unsigned char* clamp_mask, void* hip_stream);
/* feature vectors in row-major (P,256), for callers outside the fused pipeline */
int svs_tiles_to_rows(const float* tiles, int n_points, int precision, float* rows, void* hip_stream);

/* ---- a6  radiance MLP -------------------------------------------------------------------------------
 * RenderingNetwork.forward, mode 'idr' (volsdf/model/network.py:170-190): rgb (P,3) =
 * sigmoid(MLP(cat[x, PE1(view), normal, feature])).  view_dirs: (n_rays,3) when view_S == S, (P,3) when 0. */
int svs_rgb_eval(const float* points, int n_points, const float* cam, int cam_stride, const float* dirs, const float* z,
                 int S, int n_rays, const float* normals, const float* view_dirs, int view_S, const float* feat_tiles,
                 const float* stream, int precision, float* rgb, float* rbuf, void* hip_stream);
/* rbuf (svs_rgb_rbuf_bytes, may be NULL): post-ReLU activations + the 16 extra input rows, kept for svs_rgb_bwd */
size_t svs_rgb_rbuf_bytes(int n_points_total);

/* ---- a2/a3/a4  error-bounded sampler ------------------------------------------------------------------
 * ErrorBoundSampler.get_z_vals / get_error_bound, UniformSampler.get_z_vals
 * (volsdf/model/ray_sampler.py:22-43, 67-219, 221-229).  One call sequence per batch:
 *   svs_sampler_init;  for round i < max_iters: svs_sdf_vals(samples -> samples_sdf, gate = ctl.active[i]),
 *   svs_sampler_round(phase 0), svs_sampler_round(phase 1);   max_iters == 0: svs_sampler_round(phase 2).
 * Buffers: z, sdf (n_rays, svs_sampler_cap()); samples, samples_sdf (n_rays, svs_sampler_max_new());
 * beta, far (n_rays); ctl (svs_sampler_ctl_bytes(n_rays, group_rays): per convergence group of group_rays rays
 * (<= 0: all rays form one group) int conv_flag[8], active[8], final_round -- the reference tests convergence per
 * forward call, i.e. per chunk of split_n_pixels rays; one launch can cover many chunks); err_flag (1 int,
 * set on a bounding-sphere miss).  Random draws (train mode) are inputs: jitter (n_rays,n_eval),
 * u_final (n_rays,n_final), extra_idx (n_extra int), eik_idx (n_rays int); NULL = eval mode.
 * beta0 = |*beta_param| + beta_min is read on the device (volsdf/model/density.py:28-30). */
size_t svs_sampler_ctl_bytes(int n_rays, int group_rays);
int svs_sampler_ctl_stride(void);   /* ints per group record; active[i] is int 8 + i of a record */
int svs_sampler_cap(void);
int svs_sampler_max_new(void);
int svs_sampler_init(const float* cam, int cam_stride, const float* dirs, int n_rays, int n_eval, float near_, float far_,
                     int sphere_far, float sphere_radius, const float* jitter, float inv_4log, int max_iters,
                     float* samples, float* beta, float* far_out, void* ctl, int group_rays, int* err_flag,
                     void* hip_stream);
int svs_sampler_round(int phase, int n_rays, int round, int max_iters, int n_eval, int n_final, int n_extra,
                      const float* beta_param, float beta_min, float eps, int beta_iters, float add_tiny, float near_,
                      const float* far_, float* z, float* sdf, float* beta, float* samples, const float* samples_sdf,
                      void* ctl, int group_rays, const float* u_final, const int* extra_idx, const int* eik_idx, float* z_final,
                      float* z_eik, int* dbg_samples_idx, int* dbg_inds, float* dbg_cdf, float* dbg_weights,
                      void* hip_stream);

/* ---- a8  alpha compositing ----------------------------------------------------------------------------
 * VolSDFNetwork.volume_rendering (volsdf/model/network.py:281-295) and the reductions of :237-256,270-276.
 * z (R,S), sdf (R*S), rgb (R*S,3), normals (R*S,3) or NULL, depth_scale (R) -> weights (R,S), rgb_values (R,3),
 * depth_values (R), depth_vals (R,S), normal_map (R,3) or NULL. */
int svs_composite(int n_rays, int n_samples, const float* z, const float* sdf, const float* rgb, const float* normals,
                  const float* depth_scale, const float* beta_param, float beta_min, float* weights, float* rgb_values,
                  float* depth_values, float* depth_vals, float* normal_map, void* hip_stream);

/* a8 backward: gradients of a scalar loss through the compositing (hand-derived reverse pass of network.py:281-295
 * and :237-243).  d_weights / d_depth_values may be NULL.  -> d_sdf (R*S), d_rgb (R*S,3), d_beta_param (1) =
 * d loss / d density.beta; d_beta_ray (R) is workspace. */
int svs_composite_bwd(int n_rays, int n_samples, const float* z, const float* sdf, const float* rgb,
                      const float* depth_scale, const float* beta_param, float beta_min, const float* d_rgb_values,
                      const float* d_weights, const float* d_depth_values, float* d_sdf, float* d_rgb,
                      float* d_beta_ray, float* d_beta_param, void* hip_stream);

/* ---- a9 (config 4)  inverted-sphere background model, VolSDFNetworkBG (volsdf/model/network_bg.py) -----------
 * precision = SVS_MMA_F16X2 / SVS_MMA_F16X2_HALF (csrc/svs_bg_h2.hip; selects the format of the blocks kept for / written by
 * the backward) or SVS_MMA_F32 (csrc/svs_bg_f32.hip: float32 MFMAs, every block a float32 block; same buffer sizes).  Streams: svs_pack_stream which = 5 (bg implicit forward), 6 (bg implicit backward), 7 (bg radiance
 * forward), 8 (bg radiance backward); weight arrays of 9 / 2 device pointers, weight_g = NULL (no weight-norm).
 *   svs_bg_points     UniformSampler(1,0,n_bg,far=1) * (1/radius), flipped (ray_sampler.py:22-43,215-216;
 *                     network_bg.py:79-82) + depth2pts_outside (:182-214): jitter (n_rays,n_bg) train draws or NULL
 *                     -> z_bg (n_rays,n_bg), pts (n_rays*n_bg,4), depth_real (n_rays,n_bg)
 *   svs_bg_sdf_eval   bg_implicit_network (:85-88): pts (P,4) -> out0 (P) = output[:,0] (density = |out0|),
 *                     feat_tiles; training: hbuf (svs_sdf_hbuf_bytes) + ghat7 + pebuf (svs_block_bytes(P,1) each), all
 *                     or none
 *   svs_bg_rgb_eval   bg_rendering_network, mode 'nerf' (:91-93): view_dirs (n_rays,3) with view_S points per ray
 *                     (or (P,3) with view_S = 0) -> rgb (P,3); rbuf (svs_bg_rbuf_bytes, training) or NULL
 *   svs_composite_bg  volume_rendering / bg_volume_rendering and the composition (:76-125,147-180) */
int svs_bg_points(const float* cam, int cam_stride, const float* dirs, int n_rays, int n_bg, const float* jitter,
                  float radius, float* z_bg, float* pts, float* depth_real, void* hip_stream);
int svs_bg_sdf_eval(const float* pts, int n_points, const float* stream, int precision, float* out0, float* feat_tiles,
                    float* hbuf, float* ghat7, float* pebuf, void* hip_stream);
size_t svs_bg_rbuf_bytes(int n_points);
int svs_bg_rgb_eval(int n_points, const float* view_dirs, int view_S, const float* feat_tiles, const float* stream,
                    int precision, float* rgb, float* rbuf, void* hip_stream);
/* training backward of the background networks (per-point scaling as in the fg sweeps; absmax: 3 floats, caller
 * zeroes once per step: [0] max |abar|, [1] max |zbar|, [2] max |feat_bar|):
 *   svs_bg_rgb_bwd  d_rgb (P,3), rgb (P,3), rbuf, stream (which = 8) -> zbuf (svs_block_bytes(P,2): zbar_0, zbar_1 per
 *                   tile; ZERO-INITIALISED by the caller once), feat_bar (svs_block_bytes(P,1))
 *   svs_bg_sdf_bwd  d_out0 (P), feat_bar, hbuf, ghat7, stream (which = 6) -> abuf (8 blocks/tile), sbar_out (padded P);
 *                   n_points must be a multiple of 32 (n_rays * 32 is)
 * Weight gradients: svs_wgrad_multi with A = abuf / feat_bar / zbuf blocks and B = pebuf / hbuf / rbuf blocks;
 * svs_lin8_row0_grad with ubuf = NULL; svs_unpack_wgrad maps 3 (bg lin4) and 4 (bg radiance lin0). */
int svs_bg_rgb_bwd(int n_points, const float* d_rgb, const float* rgb, const float* rbuf, const float* stream, int precision,
                   float* zbuf, float* feat_bar, float* absmax, void* hip_stream);
int svs_bg_sdf_bwd(int n_points, const float* d_out0, const float* feat_bar, const float* hbuf, const float* ghat7,
                   const float* stream, int precision, float* abuf, float* sbar_out, float* absmax, void* hip_stream);
int svs_composite_bg(int n_rays, int n_samples, int n_bg, const float* z, const float* z_max, const float* sdf,
                     const float* rgb, const float* normals, const float* depth_scale, const float* beta_param,
                     float beta_min, const float* z_bg, const float* bg_out0, const float* bg_rgb, const float* bg_depth,
                     float* weights, float* bg_trans, float* bg_weights, float* rgb_values, float* depth_values,
                     float* depth_values_all, float* depth_vals, float* normal_map, void* hip_stream);
/* backward of svs_composite_bg: d_weights / d_depth_values (gradient of the fg depth, network_bg.py:109-110) /
 * d_depth_all (gradient of depth_values_all, :105-107 -- what VolSDFLoss's sparsity term reads, loss.py:72-73; needs
 * bg_depth) may be NULL; d_beta_ray (n_rays) is workspace */
int svs_composite_bg_bwd(int n_rays, int n_samples, int n_bg, const float* z, const float* z_max, const float* sdf,
                         const float* rgb, const float* depth_scale, const float* beta_param, float beta_min,
                         const float* z_bg, const float* bg_out0, const float* bg_rgb, const float* d_rgb_values,
                         const float* d_weights, const float* d_depth_values, const float* d_depth_all,
                         const float* bg_depth, float* d_sdf, float* d_rgb,
                         float* d_bg_out0, float* d_bg_rgb, float* d_beta_ray, float* d_beta_param, void* hip_stream);

/* ---- a12  weight-gradient contraction of the training backward ---------------------------------------------------
 * dW[256][ldw] += sum over points of A(:,p) B(:,p)^T for one or two operand pairs stored as wave-tile activation
 * blocks (128*64 floats per 32 points; s* = floats between consecutive blocks).  b_extra: optional 16 extra B rows
 * per block (dW columns
 * 256..271, ldw >= 272: the radiance MLP's first layer).  db[256] += row sums of pair 0's A (bias gradient).
 * dW / db are accumulated with float atomics: the caller zeroes them.
 * precision SVS_MMA_F16X2 / _HALF: the gradient-like operands (A of pair 0, B of pair 1) are scaled blocks: every point
 * carries a power-of-two scale, kept in the block's 64-float RECORD (rec0: records of a0's blocks, rec1: of b1's; [tile][64]
 * floats, see the buffer layout below); the contraction re-scales them to one power of two per launch derived from
 * *absmax (device float: their maximum magnitude, published by the fp16x2 sweeps below; NULL = operands are unscaled and
 * no records are read).  SVS_MMA_F16X2 multiplies both fp16 pieces of both operands (three products), _HALF the hi
 * pieces. */
typedef struct svs_wgrad_job {
  const float *a0, *b0; long long sa0, sb0;   /* pair 0 */
  const float *a1, *b1; long long sa1, sb1;   /* pair 1, a1 == NULL: one pair */
  const float* b_extra; long long s_extra;    /* optional 16 extra B rows of pair 0 (1024 floats per tile) */
  int n_points, ldw;
  float *dW, *db;                             /* db may be NULL */
  const float* absmax;                        /* fp16x2 scaling, may be NULL */
  const float *rec0, *rec1;                   /* fp16x2 with absmax: scale records of a0's / b1's blocks, 64 floats per tile */
} svs_wgrad_job;
/* The weight gradients of several layers (<= 20 jobs, one with b_extra counts twice) in one call.  SVS_MMA_F16X2: ONE
 * launch, ~one workgroup per CU in total, split between the jobs in proportion to their work (so each layer flushes
 * ~256/n_jobs partial sums instead of 256); SVS_MMA_F32: one launch per job. */
int svs_wgrad_multi(const svs_wgrad_job* jobs, int n_jobs, int precision, void* hip_stream);
/* single job */
int svs_wgrad(const float* a0, const float* b0, long long sa0, long long sb0, const float* a1, const float* b1,
              long long sa1, long long sb1, const float* b_extra, long long s_extra, int n_points, int precision,
              const float* absmax, const float* rec0, const float* rec1, float* dW, int ldw, float* db, void* hip_stream);

/* ---- a12  training backward of the fused MLPs (hand-written reverse mode; the reference uses torch.autograd,
 * loss.backward() at volsdf/vsdf.py:215, incl. the double backward through network.py:115-121) -------------------
 * Buffers are wave-tile activation blocks; svs_block_bytes(n_points, k) = bytes of k blocks per 32-point tile INCLUDING
 * their records (below).
 * Layout of the multi-block buffers (hbuf, gbuf, ubuf, a2buf, abuf, rbuf, zbuf): [block][wave tile][128*64 floats], the tile
 * count T padded to whole workgroups (T = 4 * ceil(n_points / 128)): block b of tile t starts at float (b * T + t) *
 * 8192 -- what a launch touches at one time and what one weight-gradient job reads is contiguous (for svs_wgrad:
 * pointer = buffer + b * T * 8192, stride 8192).  The radiance buffers use the same layout: rbuf = [4 blocks][tile]
 * followed by the 1024-float extras of every tile, zbuf = [5 blocks][tile]; the background networks' two small
 * buffers (bg rbuf, bg zbuf) are [tile][block].
 * Records (fp16x2): a buffer of k scaled blocks per tile (ubuf, a2buf, abuf, zbuf, feat_bar) is followed by k * T records
 * of 64 floats ([scale of point 0..31][max |value| of point 0..31]), in the order of the slots: the record of block b of
 * tile t is at float k * T * 8192 + (b * T + t) * 64 (also for bg zbuf, whose slots are [tile][block]).
 *   svs_rgb_bwd : d_rgb (P,3), rgb (P,3), rbuf, radiance backward stream (svs_pack_stream which=4)
 *                 -> zbuf (5 blocks/tile, ZERO-INITIALISED by the caller once), feat_bar (1 block/tile), d_normals (P,3)
 *   svs_sdf_bwd_a: second-order sweep (forward mode: u_0 = J_PE nbar, u_{l+1} = (W_l u_l) s'(a_l)).  points/rays as in
 *                 svs_sdf_outputs; d_grad (P,3) = dL/d(d sdf/dx); hbuf, gbuf from svs_sdf_outputs; stream = SDF training
 *                 stream (which=2) -> ubuf (9 blocks/tile), pebuf (1), and with SVS_MMA_F32 the second-order source blocks
 *                 a2_l = (W_l u_l) ghat_l 100 (1 - s'(a_l)) -> a2buf (8).  fp16x2 (since round 5): gbuf is not read and a2buf
 *                 not written (both may be NULL); the second half of the record of u_{l+1} holds max_r |(W_l u_l)| 100 (1 - s')
 *   svs_sdf_bwd_b: backprop.  d_sdf (P) or NULL, feat_bar for the first n_feat_points points (multiple of 32)
 *                 -> abuf (8 blocks/tile), sbar_out (32 floats per tile).  SVS_MMA_F32 reads a2buf (ubuf may be NULL); fp16x2
 *                 re-forms a2_l = u_{l+1} ghat_l 100 (1 - s') / s' from ubuf (with pass A's records) and gbuf (with the records
 *                 svs_sdf_outputs wrote behind it: max_r |ghat_l| per point; svs_sdf_gbuf_bytes) -- a2buf may be NULL: 8 KB
 *                 per point less HBM traffic than storing a2 in pass A and reading it here
 *   svs_lin8_row0_grad: out257[0..255] += dL/dW8[0,:], out257[256] += dL/db8[0] (caller zeroes)
 *   precision SVS_MMA_F16X2 (streams packed with the same precision): every point carries its own power-of-two
 *                 scale through the sweeps (gradients are far below fp16's range); all buffers hold true float32
 *                 values.  Extra arguments, NULL for SVS_MMA_F32: absmax (3 floats, caller zeroes once per step:
 *                 [0] max |abar|,|u|, [1] max |zbar|, [2] max |feat_bar| -- the scales of svs_wgrad).
 *   svs_unpack_wgrad: kernel-order dW (from svs_wgrad) -> parameter gradients incl. weight-norm backward
 *                 (w = g v/|v|, network.py:64-65).  map: 0 identity, 1 SDF lin4 (skip splice, 1/sqrt2),
 *                 2 radiance lin0, 3 bg lin4, 4 bg radiance lin0.  row_off: first parameter row covered by dWk (SDF lin8: 1, with row0 = out257). */
size_t svs_block_bytes(int n_points, int blocks_per_tile);
size_t svs_rgb_zbuf_bytes(int n_points);
size_t svs_sdf_ubuf_bytes(int n_points);
size_t svs_sdf_gbuf_bytes(int n_points);      /* gbuf of svs_sdf_outputs: 8 blocks per tile + their records */
int svs_rgb_bwd(int n_points, const float* d_rgb, const float* rgb, const float* rbuf, const float* stream, int precision,
                float* zbuf, float* feat_bar, float* d_normals, float* absmax, void* hip_stream);
int svs_sdf_bwd_a(const float* points, int n_points, const float* cam, int cam_stride, const float* dirs, const float* z,
                  int S, int n_rays, const float* d_grad, const unsigned char* clamp_mask, const float* hbuf,
                  const float* gbuf, const float* stream, int precision, float* ubuf, float* a2buf, float* pebuf,
                  float* absmax, void* hip_stream);
int svs_sdf_bwd_b(int n_points, const float* d_sdf, const unsigned char* clamp_mask, const float* feat_bar,
                  int n_feat_points, const float* hbuf, const float* gbuf, const float* a2buf, const float* ubuf,
                  const float* stream, int precision, float* abuf, float* sbar_out, float* absmax, void* hip_stream);
int svs_lin8_row0_grad(const float* hbuf, const float* ubuf, const float* sbar, int n_points, int precision, float* out257,
                       void* hip_stream);
int svs_unpack_wgrad(const float* dWk, const float* dbk, int ldw, int map, int rows, int cols, int row_off,
                     const float* weight_v, const float* weight_g, const float* row0, float* grad_v, float* grad_g,
                     float* grad_b, void* hip_stream);
/* the same for all layers of a step (<= 24 jobs) in ONE launch */
typedef struct svs_unpack_job {
  const float *dWk, *dbk;
  int ldw, map, rows, cols, row_off;
  const float *weight_v, *weight_g, *row0;
  float *grad_v, *grad_g, *grad_b;
} svs_unpack_job;
int svs_unpack_wgrad_multi(const svs_unpack_job* jobs, int n_jobs, void* hip_stream);

/* ---- a12  optimiser step ---------------------------------------------------------------------------------------
 * clip_grad_norm_(1.0) + NaN/Inf guard + Adam of VolOpt.train_step (volsdf/vsdf.py:214-219,454-463,101-102) on flat
 * float32 buffers, no host sync.  step: 1-based Adam step count; when step_counter (device int) is not NULL the launch
 * increments that counter and uses the new value instead, so that a captured launch sequence (hipGraph) replays with
 * an advancing step.  lr / betas / eps are float64 like torch's hyper-parameters (1 - beta, the bias corrections and
 * lr / bias_correction1 are formed in float64 and rounded once, as torch does).  A non-finite gradient is zeroed and
 * the Adam step still runs (torch 1.9 zero_grad semantics).  info (2 floats, may be NULL): gradient norm before
 * clipping, gradient dropped (0/1). */
size_t svs_adam_workspace_bytes(void);
int svs_clip_guard_adam(float* params, float* grads, float* exp_avg, float* exp_avg_sq, long long n, int step,
                        int* step_counter, double max_norm, double lr, double beta1, double beta2, double eps,
                        void* workspace, float* info, void* hip_stream);

/* ---- a10  MVS prior lookup ----------------------------------------------------------------------------
 * VolOpt.cost_mapping (volsdf/vsdf.py:382-452).  Points: xyz (n_points,3) or, when xyz == NULL, cam + z*dir with
 * z (n_points/S, S).  view_params: HOST float array, 17 per view: fx, fy, cx, cy, sk, c2w rows (3x4).
 * cost / z_near / z_far: HOST arrays of device pointers per view: probability volume (D,H,W), depth hypotheses
 * [0] and [-1] (H,W); dims: HOST int array D,H,W per view.  same_view: index of the rendered view (-> pi).
 * img_w, img_h: SceneDataset resolution used for the normalisation (vsdf.py:397,414-415).
 * same_view_dev (device int, may be NULL): overrides same_view at run time, so that a captured launch sequence
 * (hipGraph) can be replayed for another rendered view.
 * -> pj (n_points), pi (n_points), valid (n_points, uint8) */
int svs_cost_lookup(const float* xyz, const float* cam, const float* dirs, const float* z, int S, int n_points,
                    int n_views, int same_view, int inverse_depth, float img_w, float img_h, const float* view_params,
                    const float* const* cost, const float* const* z_near, const float* const* z_far, const int* dims,
                    float* pj, float* pi, unsigned char* valid, const int* same_view_dev, void* hip_stream);

/* ---- a11  loss ----------------------------------------------------------------------------------------
 * VolSDFLoss.forward (volsdf/model/loss.py:80-114) and the gradient of the total w.r.t. the model outputs.
 * rgb_target = ground_truth['rgb'], or 'rgb_smooth' with annealed = 1 (loss.py:103-105); pi/pj NULL = no MVS terms.
 * losses[5] = rgb, eikonal, mvs, sparse, total.  n_rays_norm / n_eik_norm (0 = n_rays / n_eik): denominators of the
 * means when a batch is processed in several ray groups (the groups' losses and gradients then simply add).
 * anneal_dev (device, 2 floats, may be NULL): {annealed, anneal_sparse} read at run time instead of the two by-value
 * arguments (captured launch sequences replayed while the annealing advances). */
int svs_loss(int n_rays, int n_samples, int n_eik, const float* rgb_values, const float* rgb_target,
             const float* grad_theta, const float* weights, const float* pi, const float* pj, const float* depth_values,
             float rgb_weight, float eikonal_weight, float mvs_weight, float sparse_weight, float gce, float confi,
             int annealed, float anneal_sparse, int n_rays_norm, int n_eik_norm, float* losses, float* d_rgb_values,
             float* d_grad_theta, float* d_weights, float* d_depth_values, double* workspace, const float* anneal_dev,
             void* hip_stream);
size_t svs_loss_workspace_bytes(int n_rays, int n_eik);

/* ---- f1  FeatureNet convolutions (models/CasMVSNet.py:24-55,338-439) ---------------------------------------------
 * out (Cout,Ho,Wo) = [add +] relu?(conv2d(in (Cin,H,W), weight) + bias), k in {1,3,5}, padding k/2, stride in {1,2};
 * BatchNorm(eval) folded into weight / bias by the caller.  weight is PACKED for scalar loads: [ceil(Cout/8)][Cin][k][k][8]
 * floats, entry [g][ci][ky][kx][c] = W[8 g + c][ci][ky][kx], zero for 8 g + c >= Cout, 32-byte aligned (the caller packs
 * once per weight tensor; svs_hip/costvol.py: conv2d_pack).  add: optional (Cout,Ho,Wo) tensor added
 * AFTER the activation; add_upsample2 != 0: add is (Cout,Ho/2,Wo/2) and enters nearest-up-sampled by 2 (the FPN's
 * top-down path, :413-431). */
int svs_conv2d(const float* in, const float* weight, const float* bias, const float* add, int add_upsample2, float* out,
               int Cin, int Cout, int H, int W, int k, int stride, int relu, void* hip_stream);
/* The whole 'fpn' FeatureNet (models/CasMVSNet.py:338-439, num_stage 3) for one image (3,H,W), H and W multiples of 4,
 * enqueued by one call: weights[i] / biases[i] (biases[i] may be null), i = conv0.0, conv0.1, conv1.0, conv1.1, conv1.2,
 * conv2.0, conv2.1, conv2.2 (BatchNorm folded), out1, inner1, out2, inner2, out3; weights packed as for svs_conv2d.
 * stage1 (4b,H/4,W/4), stage2 (2b,H/2,W/2), stage3 (b,H,W), b = base_channels. */
size_t svs_featurenet_fpn_workspace_bytes(int base_channels, int H, int W);
int svs_featurenet_fpn(const float* image, int H, int W, int base_channels, const float* const* weights,
                       const float* const* biases, float* workspace, float* stage1, float* stage2, float* stage3,
                       void* hip_stream);
/* The 3x3 (stride 1; Cin in {8,16,32}) and 5x5 (stride 2; Cin in {8,16}) layers of the pyramid, Cout <= 32, on the fp16 matrix
 * cores with two-piece fp16 operands and float32 accumulation (csrc/svs_conv2d_mfma.hip: the float32 accuracy class, 2e-7
 * relative per layer): out (Cout,Ho,Wo) = relu?(conv2d(in (Cin,H,W), k x k, padding k / 2, stride) + bias).  The weights
 * (Cout,Cin,k,k) float32, BatchNorm folded, are packed once per tensor by svs_conv2d_mfma_pack into svs_conv2d_mfma_wfrag_bytes
 * bytes of MFMA A fragments.  svs_featurenet_fpn2 = svs_featurenet_fpn with an optional table of 13 fragment pointers: layer i
 * runs on the matrix cores where wfrags[i] is not null (and the shape is supported), on the float32 kernels otherwise. */
int svs_conv2d_mfma_supported(int Cin, int Cout, int k, int stride);
size_t svs_conv2d_mfma_wfrag_bytes(int Cin, int Cout, int k);
int svs_conv2d_mfma_pack(const float* weight, int Cin, int Cout, int k, void* wfrag, void* hip_stream);
int svs_conv2d_mfma(const float* in, const void* wfrag, const float* bias, float* out, int Cin, int Cout, int H, int W, int k,
                    int stride, int relu, void* hip_stream);
/* The lateral step of the FPN (models/CasMVSNet.py:425-431) fused into the 3x3 layer behind it (out3): the layer's 32-channel
 * input X[c][y][x] = sum_ci W1[c][ci] lat_in[ci][y][x] + lat_bias[c] + lat_add[c][y / 2][x / 2] (ci < 8) is formed while the
 * kernel converts its input windows -- the float32 1x1 kernel's operations in its order, bit for bit -- and never exists in
 * memory.  lat_weight: (32,8,1,1) in svs_conv2d's packed layout; wfrag: (Cout <= 16, 32, 3, 3) packed by svs_conv2d_mfma_pack;
 * H, W even.  svs_featurenet_fpn2 uses it for inner2 + out3 when base_channels = 8 and wfrags[12] is given. */
int svs_conv2d_mfma_lateral(const float* lat_in, const float* lat_weight, const float* lat_bias, const float* lat_add,
                            const void* wfrag, const float* bias, float* out, int Cout, int H, int W, int relu, void* hip_stream);
int svs_featurenet_fpn2(const float* image, int H, int W, int base_channels, const float* const* weights,
                        const float* const* biases, const void* const* wfrags, float* workspace, float* stage1, float* stage2,
                        float* stage3, void* hip_stream);

/* ---- a13/a14  homography warp + variance --------------------------------------------------------------------
 * homo_warping (models/CasMVSNet.py:280-315) for every source view fused with the variance aggregation of
 * DepthNet.forward (:611-642): variance (C,D,H,W) = sum(f^2)/V - (sum(f)/V)^2 over the reference feature (C,H,W)
 * and the n_src warped source features.  Source features are passed channel-last (H,W,C) -- svs_chw_to_hwc.
 * rot_trans: HOST float array, 12 per source view: rows of (src_proj @ inv(ref_proj))[:3,:3], then [:3,3], with
 * proj = K[:3,:3] @ E[:3,:4] (:622-625).  depth_values (D,H,W) per-pixel hypotheses.  C in {8,16,32}.
 * raw_warp != 0: write the warped volume of source 0 instead (homo_warping on its own). */
int svs_chw_to_hwc(const float* in, float* out, int C, int H, int W, void* hip_stream);
int svs_warp_variance(const float* ref_feature, const float* const* src_features_hwc, const float* rot_trans, int n_src,
                      int C, int D, int H, int W, const float* depth_values, float* variance, int raw_warp,
                      void* hip_stream);

/* ---- a15  3-D regularisation blocks ----------------------------------------------------------------------------
 * Conv3d / Deconv3d blocks of CostRegNet (models/CasMVSNet.py:107-186,441-472) with BatchNorm(eval) folded:
 * out = [skip +] relu?(conv(in, weight) + bias).  weight: [Cin][27][Cout], tap = (kd*3+kh)*3+kw, BN scale folded
 * in; bias: BN shift (NULL for the final `prob` conv).  transposed: ConvTranspose3d(k3,s2,p1,output_padding 1). */
int svs_conv3d(const float* in, const float* weight, const float* bias, const float* skip, float* out, int Cin, int Cout,
               int Di, int Hi, int Wi, int stride, int transposed, int relu, void* hip_stream);
/* The stride-1 convolutions with Cin in {8,16,32} and Cout <= 16 (conv0, conv2, prob: 79 % of the U-Net's MACs at
 * stage 1) on the fp16 matrix cores with two-piece split operands (float32-class accuracy).
 * wfrag (svs_conv3d_mfma_wfrag_bytes(Cin)): the folded weights as fp16 hi / mid A fragments of
 * v_mfma_f32_16x16x32_f16: [k-step s][piece][lane][8]: output channel lane & 15, k = 32 s + 8 (lane >> 4) + j =
 * tap * Cin + cin (zero for tap > 26 and channels >= Cout). */
size_t svs_conv3d_mfma_wfrag_bytes(int Cin);
int svs_conv3d_mfma(const float* in, const void* wfrag, const float* bias, const float* skip, float* out, int Cin,
                    int Cout, int D, int H, int W, int relu, void* hip_stream);

/* conv0 of CostRegNet (C -> 8 at full resolution, models/CasMVSNet.py:444,460; 68 / 52 / 35 % of the U-Net's MACs at
 * stage 1 / 2 / 3) fused with the producer of its input.  The variance volume travels between the two kernels as a
 * "split volume": fp16 hi and mid parts (v = hi + mid to float32 accuracy) in channel-last 16-byte units with a zero
 * border, [D+2][Hp][2 pieces][C/8][Wp] units, voxel (z,y,x) at (z+1,y+1,x+1), Hp = 4 ceil(H/4) + 2,
 * Wp = 32 ceil(W/32) + 4.  svs_split_volume_dims returns its size in bytes (dims[0..1] = Hp, Wp); the caller zero-fills
 * the buffer once, the producers write the interior only.
 *   svs_warp_variance_split  = svs_warp_variance writing that form (DepthNet.forward steps 1-2, CasMVSNet.py:611-642);
 *   svs_split_volume_pack    = the same form from a float32 (C,D,H,W) volume (tests, other callers);
 *   svs_conv3d_pair          = relu?(conv3d(volume, 3x3x3, stride 1, padding 1) + bias), Cin in {8,16,32}, Cout <= 8:
 *     the 16 rows of v_mfma_f32_16x16x32_f16 are 8 output channels x 2 neighbouring x positions, a column is that pair
 *     of positions, K = 3 kd x 3 kh x 4 input columns x Cin (27 of 36 products useful; 27 of 54 with channels only).
 *     wfrag (svs_conv3d_pair_wfrag_bytes(Cin)): [k-step s][piece][lane][8] fp16, row m = lane & 15 = (channel m & 7,
 *     position m >> 3), k = 32 s + 8 (lane >> 4) + j = (((kd*3 + kh)*4 + t)*(Cin/8) + g)*8 + c8: the folded weight of
 *     tap (kd, kh, kw = t - (m >> 3)) and input channel 8 g + c8, zero if kw is outside 0..2 or m & 7 >= Cout. */
size_t svs_split_volume_dims(int C, int D, int H, int W, int* dims);
/* The final `prob` layer (Cin -> 1, CasMVSNet.py:458,471): float32 fused multiply-adds on the vector ALUs (a one-row
 * matrix-core tile would be 1/16 full).  weight [Cin][27][1]. */
int svs_conv3d_c1(const float* in, const float* weight, const float* bias, const float* skip, float* out, int Cin, int D,
                  int H, int W, int relu, void* hip_stream);
int svs_split_volume_pack(const float* in, void* split, int C, int D, int H, int W, void* hip_stream);
int svs_warp_variance_split(const float* ref_feature, const float* const* src_features_hwc, const float* rot_trans,
                            int n_src, int C, int D, int H, int W, const float* depth_values, void* split,
                            void* hip_stream);
size_t svs_conv3d_pair_wfrag_bytes(int Cin);
int svs_conv3d_pair(const void* split, const void* wfrag, const float* bias, float* out, int Cin, int Cout, int D, int H,
                    int W, int relu, void* hip_stream);

/* Every other 3x3x3 layer of the U-Net (stride 2, transposed, the 32/64-channel levels) as an implicit GEMM on the
 * same matrix-core instruction with both operands split: Cin in {8,16,32,64}, Cout <= 64
 * (svs_conv3d_gemm_supported).  One wave = 16 output voxels along x times all output channels; the transposed form
 * runs as 8 GEMMs, one per output parity class, over only the taps that class uses.
 * wfrag: svs_conv3d_gemm_wfrag_bytes(Cin, Cout, transposed), filled once per layer by svs_conv3d_gemm_pack from the
 * folded [Cin][27][Cout] weights (layout [class][k-step][m-tile][hi,mid][lane][8 fp16], k = tap-in-class * Cin + cin). */
int svs_conv3d_gemm_supported(int Cin, int Cout);
size_t svs_conv3d_gemm_wfrag_bytes(int Cin, int Cout, int transposed);
int svs_conv3d_gemm_pack(const float* weight, int Cin, int Cout, int transposed, void* wfrag, void* hip_stream);
int svs_conv3d_gemm(const float* in, const void* wfrag, const float* bias, const float* skip, float* out, int Cin, int Cout,
                    int Di, int Hi, int Wi, int stride, int transposed, int relu, void* hip_stream);
/* conv1 of CostRegNet (8 -> 2b channels, stride 2, from the full-resolution volume: CasMVSNet.py:445,461): a lane group
 * owns whole (kz,ky) input rows, one 8-byte load per (row, channel) serves the taps kx = 1, 2 and the neighbour lane's
 * value the tap kx = 0.  Cout <= 16, Wi even.  wfrag (svs_conv3d_s2c8_wfrag_bytes()): [9 k-steps][piece][lane][8] fp16,
 * row lane & 15 = output channel, k = 32 s + 8 g + j with s = kx*3 + q, g = lane >> 4: tap ((g+4q)/3, (g+4q)%3, kx),
 * input channel j; zero for g + 4q > 8. */
size_t svs_conv3d_s2c8_wfrag_bytes(void);
/* split_out != NULL (Cout = 16, no skip): the output leaves as a split volume (svs_split_volume_dims(16, Do, Ho, Wo)) instead
 * of float32 to `out` -- the input form of svs_conv3d_rows (conv2). */
int svs_conv3d_s2c8(const float* in, const void* wfrag, const float* bias, const float* skip, float* out, void* split_out,
                    int Cout, int Di, int Hi, int Wi, int relu, void* hip_stream);
/* conv2 of CostRegNet (2b -> 2b at half resolution, CasMVSNet.py:446,462) from a split volume: slice ring + LDS-DMA as
 * svs_conv3d_pair, the 16 MFMA rows = output channels.  Cin = 16, Cout <= 16.  wfrag (svs_conv3d_rows_wfrag_bytes(Cin)):
 * [k-step s][piece][lane][8] fp16, row lane & 15 = output channel, k-step s / lane group lane >> 4 = combination
 * 4 s + (lane >> 4) = tap * (Cin/8) + g of the (kd,kh,kw)-major tap list: weight of that tap, input channel 8 g + c8. */
size_t svs_conv3d_rows_wfrag_bytes(int Cin);
int svs_conv3d_rows(const void* split, const void* wfrag, const float* bias, float* out, int Cin, int Cout, int D, int H,
                    int W, int relu, void* hip_stream);

/* ---- a14 tail  softmax over D, depth regression, photometric confidence (models/CasMVSNet.py:648-663) ---------
 * reg, depth_values (D,H,W) -> prob (D,H,W), depth (H,W), conf (H,W), index (H,W int, may be NULL). */
int svs_prob_depth_conf(const float* reg, const float* depth_values, int D, int H, int W, float* prob, float* depth,
                        float* conf, int* index, void* hip_stream);

/* ---- a16  depth hypotheses (models/CasMVSNet.py:519-595,733-751) -------------------------------------------------
 * prev_depth == NULL: D planes dmin..dmax (inverse != 0: uniform in 1/depth); else the previous stage's depth
 * (Hp,Wp) -> +-(D/2)*pix_interval around its bilinear resize to the image, resized to (D, H_img/scale, W_img/scale). */
int svs_depth_hypotheses(const float* prev_depth, int Hp, int Wp, int H_img, int W_img, int D, int scale, float dmin,
                         float dmax, float pix_interval, int inverse, float* out, void* hip_stream);

/* ---- f3  depth-map fusion (helpers/utils.py:75-132, runner.py:301-386) -------------------------------------------
 * svs_fuse_view: reproject_with_depth + check_geometric_consistency of ONE reference view against n_src <= 16 source
 * views, then the aggregation of filter_depth: geo_mask_sum, depth_est_averaged = (sum of the masked reprojected
 * depths + ref depth) / (geo_mask_sum + 1) (float64, as numpy promotes it), photo_mask = confidence > conf,
 * geo_mask = geo_mask_sum >= thres_view, final = photo & geo [& extra_mask].
 * mats: DEVICE float64, svs_fuse_mats_per_src() = 68 per source view, row-major:
 *   inv(K_ref) (9), E_src @ inv(E_ref) (16), K_src (9), inv(K_src) (9), E_ref @ inv(E_src) (16), K_ref (9)
 *   -- formed by the host in float32 exactly as the reference forms them, then widened.
 * The source depth is sampled like cv2.remap(INTER_LINEAR, BORDER_CONSTANT 0): 5-bit fixed-point coordinates.
 * Optional per-source outputs (all four or none, (n_src,H,W)): the tuple check_geometric_consistency returns.
 * svs_fuse_points: the surviving pixels in row-major order -> world xyz float32 (count,3) and, if ref_img (H,W,3
 * float32 in [0,1]) is given, uint8 colours (count,3) = trunc(img * 255).  mats: inv(K_ref) (9), inv(E_ref) (16).
 * offset_ws: H*W ints; xyz / rgb sized for H*W points; *count (device int) receives the number written. */
int svs_fuse_mats_per_src(void);
int svs_fuse_view(const float* ref_depth, const float* confidence, const float* const* src_depths, const double* mats,
                  int n_src, int H, int W, float conf, double filter_dist, float filter_diff, int thres_view,
                  const uint8_t* extra_mask, double* depth_avg, uint8_t* photo_mask, uint8_t* geo_mask, uint8_t* final_mask,
                  uint8_t* src_mask, float* src_depth_reproj, float* src_x, float* src_y, void* hip_stream);
int svs_fuse_points(const double* depth_avg, const uint8_t* final_mask, const float* ref_img, const double* mats, int H, int W,
                    int* offset_ws, float* xyz, uint8_t* rgb, int* count, void* hip_stream);

/* ---- f4  Chamfer evaluator on point clouds (evals/eval_dtu.py:100-176) ---------------------------------------------
 * All clouds are (n,3) float64 (what open3d hands the reference).  One structure serves both neighbour problems:
 * points sorted by uniform-grid cell + a hash from cell to its run; grid_ws: svs_cloud_grid_bytes(n_points of the
 * cloud the grid is built over); origin: HOST double[3], <= every coordinate of both clouds, (max - origin)/cell < 2^21.
 * svs_cloud_nn: sklearn NearestNeighbors(n_neighbors=1).fit(ref).kneighbors(query) (:150-152,:174-175): dist
 *   (float64, same arithmetic as the kd-tree) and idx (may be NULL).  Exact for neighbours closer than max_radius;
 *   a result >= max_radius is an upper bound (+inf / -1 if none met) -- the protocol drops distances >= max_dist.
 * svs_cloud_downsample_*: the greedy radius down-sampling (:104-118) = the lexicographically-first maximal
 *   independent set of the radius graph in index order.  _begin builds the grid (cell = radius) and clears state;
 *   every _round settles more points (state: 0 undecided, 1 kept, 2 dropped; *undecided = points still open after
 *   the round); repeat until *undecided == 0.
 * svs_cloud_obs_filter: bounding-box-plus-patch test and ObsMask lookup (:124-135).  bb: HOST float[6] (BB rows),
 *   obs_mask: uint8 (d0,d1,d2) C order.  inbound[i] = inside the padded box; in_obs[i] = inbound & grid-inbound & mask.
 * svs_cloud_plane_side: (P . [x,y,z,1]) > 0 (:165-167), plane: HOST double[4].
 * svs_cloud_compact: rows with mask != 0, order kept; offset_ws: n ints; *count (device int).
 * svs_cloud_mean_below: mean of dist[dist < max_dist] (:153,:176) -> mean_count[0], and the number of terms
 *   -> mean_count[1]; deterministic two-level sum.  workspace: svs_cloud_mean_workspace_bytes(). */
size_t svs_cloud_grid_bytes(int n_points);
int svs_cloud_nn(const double* ref, int n_ref, const double* query, int n_query, const double* origin, double cell,
                 double max_radius, void* grid_ws, double* dist, int* idx, void* hip_stream);
int svs_cloud_downsample_begin(const double* pts, int n, const double* origin, double radius, void* grid_ws, uint8_t* state,
                               void* hip_stream);
int svs_cloud_downsample_round(const double* origin, int n, double radius, void* grid_ws, uint8_t* state, int* undecided,
                               void* hip_stream);
int svs_cloud_obs_filter(const double* pts, int n, const float* bb, double res, double patch, const uint8_t* obs_mask,
                         int d0, int d1, int d2, uint8_t* inbound, uint8_t* in_obs, void* hip_stream);
int svs_cloud_plane_side(const double* pts, int n, const double* plane, uint8_t* above, void* hip_stream);
int svs_cloud_compact(const double* pts, const uint8_t* mask, int n, int* offset_ws, double* out, int* count, void* hip_stream);
size_t svs_cloud_mean_workspace_bytes(void);
/* svs_cloud_bounds: lo_hi[0..2] = per-axis minimum, lo_hi[3..5] = per-axis maximum of pts (n >= 1; DEVICE double[6]) -- the
 * origin and extent of the search grids (the evaluator took them from two column reductions of torch: 2.6 ms per 4 M-point
 * cloud, as long as the neighbour search itself).  workspace: svs_cloud_bounds_workspace_bytes(). */
size_t svs_cloud_bounds_workspace_bytes(void);
int svs_cloud_bounds(const double* pts, int n, double* workspace, double* lo_hi, void* hip_stream);
int svs_cloud_mean_below(const double* dist, int n, double max_dist, double* workspace, double* mean_count, void* hip_stream);

/* Evaluator --mode mesh (evals/eval_dtu.py:14-23 sample_single_tri, :62-90): the points the script samples on every
 * triangle of the predicted mesh before it proceeds as in point-cloud mode.  tri: DEVICE double[n_tri][11] =
 * [n1, n2, v1(3), v2(3), p0(3)] per triangle of non-zero area (:70-83; integer-valued n1 = floor(l1 / thr), n2).
 * svs_mesh_sample_count: counts[t] = number of grid entries ((i + 0.5) / max(n1, 1e-7), (j + 0.5) / max(n2, 1e-7)),
 *   i = 0..n1, j = 0..n2, whose coordinates sum to less than 1 (:22).
 * svs_mesh_sample_points: out[offsets[t] + m] = v1 c0 + v2 c1 + p0 (:23) for the m-th such entry in row-major order;
 *   offsets = exclusive prefix sum of counts, out: DEVICE double[sum(counts)][3].  Each product and sum is rounded
 *   separately, as numpy's. */
int svs_mesh_sample_count(const double* tri, int n_tri, long long* counts, void* hip_stream);
int svs_mesh_sample_points(const double* tri, int n_tri, const long long* offsets, double* out, void* hip_stream);

/* ---- numeric-contract self tests (used by tests/test_gpu_parity.py) --------------------------------------- */
int svs_selftest_exp(const float* x, float* y_exp, float* y_expm1, int n, void* hip_stream);
int svs_selftest_arith(const float* a, const float* b, float* quotient, float* sqrt_abs_a, int n, void* hip_stream);
int svs_selftest_cumsum(const float* x, float* y, float* total, int rows, int m, void* hip_stream);
/* total[r] = torch.sum(x[r, :m], -1) of a float32 row in ATen's cascade_sum order (the order of the reference's
 * `pdf / torch.sum(pdf, -1, keepdim=True)`, volsdf/model/ray_sampler.py:149,161, and of `(dists ** 2.).sum(-1)`, :77);
 * svs_selftest_exp evaluates the restated Sleef_expf8_u10 / Sleef_expm1f8_u10 (torch.exp pinned / torch.expm1,
 * ray_sampler.py:130-146,225-227, volsdf/model/density.py:26).  1 <= m <= 16000. */
int svs_selftest_rowsum(const float* x, float* total, int rows, int m, void* hip_stream);

/* ---- host: the training pixels of a step (volsdf/datasets/scene_dataset.py:275-279 `change_sampling_idx`, called by
 * volsdf/vsdf.py:234 after every step) ------------------------------------------------------------------------------
 * out[0..k) = torch.randperm(n)[:k] for the CPU generator whose serialised state (torch.get_rng_state(): 5056 HOST bytes)
 * is rng_state, which is advanced exactly as torch.randperm(n) advances it (n - 1 draws of at::mt19937): the first k
 * iterations of ATen's forward Fisher-Yates shuffle, the remaining draws skipped without being formed -- O(k + n / 624)
 * instead of a shuffle of all n pixels.  HOST pointers; n < 2^32 / 20 (ATen's small-n algorithm). */
int svs_randperm_prefix(unsigned char* rng_state, size_t state_bytes, long long n, long long k, long long* out);

/* ---- launch plans: the device part of a step enqueued by one call --------------------------------------------
 * The launch sequence of VolOpt.train_step (volsdf/vsdf.py:196-235: forward, prior lookup, loss, backward) contains no host
 * decision; the host records it once per configuration with a stream capture and hands the captured hipGraph_t to
 * svs_plan_build, which reads its nodes and edges (kernel, memcpy, memset and empty nodes; anything else: SVS_EINVAL) into
 * a plan: the nodes in a topological order, each on one of <= 12 HIP streams (chains of the dependency graph), one event
 * per edge that crosses streams.  The critical chain (every node hands its stream to the successor with the longest way
 * to the end) is the first chain; side_streams: optional hipStream_t handles of the caller for the 2nd, 3rd ... chain
 * (which hardware queue a stream maps to is decided when it is created -- a caller whose eager schedule runs well hands in
 * the streams of that schedule); chains beyond them get streams the plan creates.  svs_plan_run enqueues the plan with plain launches -- first chain on `hip_stream`, the
 * others on streams the plan owns, ordered behind what `hip_stream` held when the call was made and joined back into it
 * before the call returns -- and does not synchronise.  The graph is never instantiated or launched; it must outlive the
 * plan (the kernel arguments stay in its nodes).  svs_plan_info: counts[8] = nodes, kernels, copies, memsets, empty
 * nodes, streams, events, side streams that start at the entry. */
int svs_plan_build(void* hip_graph, void* const* side_streams, int n_side_streams, void** plan_out);
int svs_plan_run(void* plan, void* hip_stream);
int svs_plan_info(void* plan, int* counts);
/* one line per node in issue order: "<position> s<stream> kernel <name> grid .. | memcpy .. | memset .. | empty", followed by
 * " w<event>" per event waited for in front of it and " r<event>" when an event is recorded behind it (HOST text buffer) */
int svs_plan_describe(void* plan, char* text, size_t capacity);
int svs_plan_destroy(void* plan);

#ifdef __cplusplus
}
#endif
#endif /* SVOLSDF_HIP_H */
