/*
 * goslam_neus.h -- C ABI of the mapping hot path (hash-grid NeuS renderer) in libgoslam_hip.so.
 *
 * Replaces the reference's use of tiny-cuda-nn (`tcnn.Encoding`, `tcnn.Network`;
 * reference src/InstantNeuS.py:62,192) and the PyTorch op chains of
 * `Renderer.render_batch_ray` (src/render.py:73-175) and `InstantNeuS.forward`
 * (src/InstantNeuS.py:295-370).  Conventions as in goslam_hip.h: device pointers unless the name
 * ends in `_host`, dense row-major tensors, hipStream_t as void*, no sync / no allocation.
 */
#ifndef GOSLAM_NEUS_H
#define GOSLAM_NEUS_H

#include "goslam_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Hash-grid geometry shared by host and device (tiny-cuda-nn HashGrid as configured at
 * src/InstantNeuS.py:44-52: 16 levels x 2 features, T=2^19, base 16, scale 1.447269237440378). */
#define GS_GRID_LEVELS 16
#define GS_GRID_FEATS 2
typedef struct {
  float scale[GS_GRID_LEVELS];
  uint32_t resolution[GS_GRID_LEVELS];
  uint32_t size[GS_GRID_LEVELS];     /* entries in the level's table */
  uint32_t offset[GS_GRID_LEVELS];   /* first entry of the level in the flat table */
  uint32_t hashed[GS_GRID_LEVELS];   /* 1 if the level is hashed, 0 if dense */
  uint32_t total;                    /* total entries (x GS_GRID_FEATS parameters) */
} gs_grid_meta;

/* Fill `meta_host` for the InstantNeuS configuration (tcnn grid.h constructor arithmetic). */
int gs_grid_meta_default(gs_grid_meta* meta_host);

/* Renderer.render_batch_ray sample placement (src/render.py:99-171): ray/AABB far bound,
 * stratified + near-surface samples, per-ray sort (merge of the two sorted runs), dists.
 *   rays_o/rays_d f32 [n,3]; gt_depth f32 [n] or NULL (then n_surface is ignored);
 *   bound f32 [3,2] (device); t_samples f32 [n_samples] = torch.linspace(0,1,n_samples),
 *   t_surface f32 [n_surface] likewise; perturb f32 [n_samples] = the shared
 *   torch.rand(N_samples) vector (:159) or NULL; gt_max = gt_depth.max() as a host scalar, or -- when
 *   gt_max_dev != NULL -- read from that device scalar instead (no host round trip per batch); or -- gt_max_dev == NULL
 *   and gt_max == -INFINITY, n_samples / n_surface <= 64 -- taken over gt_depth inside the launch (NaN if any depth is
 *   NaN, as torch.max): no reduction launch in front of this one;
 *   -> z_vals, dists f32 [n, n_samples + n_surface].                                          */
int gs_render_sample(const float* rays_o, const float* rays_d, const float* gt_depth,
                     const float* bound, const float* t_samples, const float* t_surface,
                     const float* perturb, float gt_max, const float* gt_max_dev, float* z_vals, float* dists,
                     int n, int n_samples, int n_surface, gs_stream_t stream);

/* tcnn.Encoding.__call__ (src/InstantNeuS.py:62,86): x f32 [n,3] in [0,1], grid f16
 * [total*2] -> out f16 [n,32]; optional dy_dx f32 [n,32,3] (analytic d out / d x).            */
int gs_grid_encode(const float* x, const void* grid, void* out, float* dy_dx, int n,
                   gs_stream_t stream);

/* Backward passes of tcnn.Encoding (the autograd the reference drives at src/InstantNeuS.py:134-148: the encoding is
 * differentiated w.r.t. x with create_graph=True on EVERY forward, and that gradient is differentiated again by
 * loss.backward()).  x f32 [n,3] in [0,1]; dy [n,32] f16 or f32 (dy_dtype), read as dy * dy_scale; grid f16
 * [total*2] (may be NULL when neither dx nor ddy is requested).
 *   v == NULL -- first order (tcnn kernel_grid_backward / kernel_grid_backward_input):
 *       grid_grad += sum_points w_corner(x) * dy     (atomically accumulated; zero it first; NULL = skip)
 *       dx f32 [n,3] = sum_c dy_c * d y_c / d x      (NULL = skip)
 *   v f32 [n,3] = d L / d (dx) -- second order (tcnn kernel_grid_backward_input_backward_*):
 *       ddy f32 [n,32] = (d y / d x) . v             (d L / d dy; NULL = skip)
 *       grid_grad += dy * sum_d v_d * d w_corner / d x_d
 *       dx f32 [n,3] = sum_c dy_c * (d^2 y_c / dx dx) v   (trilinear interpolation: mixed terms only)
 * grid_grad dtype GS_F32, or GS_F16 = tcnn's own mode (one packed fp16 atomic per table entry); every
 * contribution is multiplied by grid_grad_scale (tcnn's loss scale; the caller divides it out).       */
int gs_grid_backward(const float* x, const void* grid, const void* dy, int dy_dtype, float dy_scale,
                     const float* v, void* grid_grad, int grid_grad_dtype, float grid_grad_scale,
                     float* dx, float* ddy, int n, gs_stream_t stream);

/* tcnn.Network.__call__ (src/InstantNeuS.py:192,201): FullyFusedMLP n_in(->80, padded with
 * ones)->64->64->n_out(->16), ReLU, no bias; mlp f16 [64*80+64*64+16*64] row-major [out,in] per
 * layer; x f16 [n,n_in] -> out f16 [n,n_out].  Workspace only needed when n_in != 80.         */
size_t gs_mlp_workspace_bytes(int n, int n_in);
int gs_mlp_forward(const void* x, const void* mlp, void* out, int n, int n_in, int n_out,
                   void* workspace, size_t workspace_bytes, gs_stream_t stream);

/* InstantNeuS.forward (src/InstantNeuS.py:295-370) for one chunk of rays, forward only:
 * hash-grid encode + SDF linear + analytic SDF gradient + NeuS alpha + colour MLP + compositing.
 *   grid f16 [total*2], sdf_w f32 [32,35], sdf_b f32 [32], color_B f32 [3,33], mlp f16 [10240];
 *   inv_s = clip(exp(10*variance), 1e-6, 1e6) as a host scalar, or -- when inv_s_dev != NULL -- read from that device
 *   scalar instead (a training loop then never reads the variance parameter back to the host);
 *   bound_host / rt_bound_host: HOST f32 [3,2] (static bound for normalisation, realtime bound
 *   for the in-bound mask; 6 floats each, passed by value to the kernels).  rt_bound_dev (optional, DEVICE f32
 *   [3,2]): when non-NULL the kernels read the realtime bound from it instead of rt_bound_host, so a captured
 *   hipGraph of the launch follows InstantNeuS.update_bound (src/InstantNeuS.py:255-257,310) without a re-capture.
 * Outputs f32: color [n,3], depth [n], depth_var [n], normal [n,3], weight_sum [n], sdf [n,s],
 *   z_mid [n,s] (= z_vals + dists/2), grad_err_ray [n] (per-ray sum of (|grad|-1)^2 * mask; the
 *   caller divides the total by n*s); optional per-point alpha f32 [n,s], rgb f16 [n,s,3],
 *   grad f32 [n,s,3], mask u8 [n,s], mlp_in f16 [n,s,80] (NULL = keep in the workspace; the
 *   training path saves them for gs_neus_backward_*); enc_aux_out f16 [16,n*s,8] (optional): per level and point
 *   the record [enc0, enc1, d enc0 / dx (3), d enc1 / dx (3)] of the in-bound points -- handed to
 *   gs_neus_backward_points* as `enc_aux`, the backward streams it instead of gathering the table again.
 *   grad_err_scale multiplies grad_err_ray (1 = the raw per-ray sums the training step reduces; 1 / (n s) makes
 *   sum(grad_err_ray) the `gradient_error` of InstantNeuS.py:360-370 directly); sdf_variance_out f32 [n] (optional)
 *   is filled with sdf_variance_value (`sdf_variance` of the same dict).
 * The workspace needs no initial state (the in-bound flags are one byte per wave, each written by its wave).        */
size_t gs_neus_forward_workspace_bytes(int n, int s);
/* gs_neus_forward gathers the hashed levels LEVEL-MAJOR (one 2 MB level at a time per XCD, 16-byte records, then the
 * per-point stage) for batches of at least this many sample points, and per point below it.  Returns the previous value;
 * points < 0 only queries.  Default 524288 (measured: the two orders meet at ~442 K points on MI355X, profiles/r05_level_major_crossover.json).  Same results either way up to the
 * fp32 summation order of d sdf / d x.                                                                            */
int gs_neus_level_major_min_points(int points);
int gs_neus_forward(const float* rays_o, const float* rays_d, const float* z_vals,
                    const float* dists, const void* grid, const float* sdf_w, const float* sdf_b,
                    const float* color_B, const void* mlp, float inv_s, const float* inv_s_dev,
                    const float* bound_host, const float* rt_bound_host, const float* rt_bound_dev,
                    float* color, float* depth, float* depth_var, float* normal,
                    float* weight_sum, float* sdf, float* z_mid, float* grad_err_ray,
                    float* alpha_out, void* rgb_out, float* grad_out, uint8_t* mask_out,
                    void* mlp_in_out, void* enc_aux_out, float grad_err_scale, float* sdf_variance_out,
                    float sdf_variance_value, int n, int s,
                    void* workspace, size_t workspace_bytes, gs_stream_t stream);

/* Backward of InstantNeuS.forward, stage 1 (per ray): from the upstream gradients of the ray
 * outputs -- d_color [n,3], d_depth [n], d_depth_var [n], d_normal [n,3], d_weight_sum [n] --
 * and the saved per-point alpha / rgb / z_mid / grad / mask, produce d_alpha f32 [n,s] (w.r.t. the
 * unmasked alpha), d_rgb f32 [n,s,3] and the normal-output part of d_grad f32 [n,s,3].  s <= 128.  */
int gs_neus_backward_rays(const float* alpha, const void* rgb, const float* z_mid, const float* grad,
                          const uint8_t* mask, const float* d_color, const float* d_depth,
                          const float* d_depth_var, const float* d_normal, const float* d_weight_sum,
                          float* d_alpha, float* d_rgb, float* d_grad, int n, int s, gs_stream_t stream);

/* Backward of the fused colour MLP (tcnn FullyFusedMLP 67(->80)->64->64->3(->16), ReLU, sigmoid output) in ONE
 * kernel: recomputes H1, H2 as gs_mlp_forward does, then dX and the three weight gradients on MFMA.
 *   x f16 [n,80] (the saved MLP input rows), d_rgb f32 [n,3] (gradient w.r.t. the sigmoid outputs), rgb f16
 *   [n,3] (the saved outputs; NULL = no output activation, d_rgb is the gradient w.r.t. the raw network outputs,
 *   which is tcnn.Network's own contract: `output_activation: none`, src/InstantNeuS.py:184-192), loss_scale (tcnn: 128) multiplies every gradient that travels in fp16.
 *   wpack f16 [40][64][8]: MFMA A-fragments A[l][e] = M[32 mt + (l&31)][k(ks, l>>5, e)] of, in this
 *   order, M = W1 (mt<2, ks<5), W2 (mt<2, ks<4), W3^T (mt<2, ks=0), W2^T (mt<2, ks<4), W1^T zero-padded to 96
 *   rows (mt<3, ks<4), each block mt-major.  k(ks, hf, e) = 16 ks + 8 hf + e where the contraction runs over inputs /
 *   outputs (W1, W3^T); where it runs over HIDDEN neurons (W2, W2^T, W1^T) it is the order in which an accumulator
 *   tile of the previous GEMM holds them: k = 32 (ks >> 1) + 8 (2 (ks & 1) + (e >> 2)) + 4 hf + (e & 3) -- the kernel
 *   feeds one GEMM's accumulator registers to the next as its B fragments (gs_map_step_prep gathers this layout).
 * Outputs: dx f16 [n,80] = loss_scale * dL/dx;  partial f32 [gs_mlp_backward_blocks(n)][10240]: per-workgroup
 *   partial sums of loss_scale * (dW1 [64,80] | dW2 [64,64] | dW3 [16,64]) in tcnn's parameter layout -- sum
 *   over the first axis and divide by loss_scale.                                                        */
int gs_mlp_backward_blocks(int n);
int gs_mlp_backward(const void* x, const void* wpack, const float* d_rgb, const void* rgb, float loss_scale,
                    void* dx, float* partial, int n, gs_stream_t stream);

/* Stage 2 (per sample point): chain d_alpha, d_sdf (loss on the sdf samples), d_grad (stage 1 +
 * the eikonal term d_gerr_ray[ray] * d(|grad|-1)^2 * mask + the colour MLP's input gradient dX[:,33:36]),
 * d_feat = dX[:,36:67] and d_emb = dX[:,0:33] through NeuS alpha, the SDF linear layer and the hash
 * grid, INCLUDING the second-order path through d sdf / d x (tiny-cuda-nn's double backward).
 *   dX [n*s,80] is the colour MLP's input gradient (computed by the caller's MLP backward): f32, or f16
 *   holding dx_scale * gradient (the loss-scaled output of an fp16 GEMM).
 *   The per-point rows d_out, lin_in, dw0, d_arg, pts are f32 or f16 (row_dtype); the gradient-valued ones
 *   (d_out, dw0, d_arg) are multiplied by row_scale before they are stored (loss scale for fp16 rows).
 *   f16 rows are padded to GEMM-friendly widths -- lin_in / dw0 / d_arg 40, pts 8 (d_out stays 32); the pad
 *   columns are written as zeros, except pts[:,3] = 1 (so that pts^T @ rows also yields column sums).
 *   row_stride = 0: five separate contiguous buffers; row_stride = R (f16 only, R % 8 == 0): consecutive points'
 *   rows are R elements apart in every buffer, i.e. the five pointers address column blocks of ONE [n*s, R]
 *   matrix whose Gram matrix then delivers every dense-parameter gradient in a single GEMM.
 * Outputs: grid_grad [total*2] (atomically accumulated; zero it first) -- dtype GS_F32, or GS_F16 =
 * tiny-cuda-nn's mode: fp16 table gradient, both features of an entry added with one packed atomic,
 * every contribution pre-multiplied by grid_grad_scale (tcnn's loss scale, 128; the caller divides it
 * out) -- d_out f32 [n*s,32]
 * and lin_in f32 [n*s,35] (d sdf_layer.weight = d_out^T @ lin_in, d bias = colsum(d_out)),
 * dw0 f32 [n*s,35] (extra per-point contribution to sdf_layer.weight row 0: colsum),
 * d_arg f32 [n*s,33] (d color_B = pts^T @ d_arg), pts f32 [n*s,3], d_inv_s f32 [1] (atomic; zero it). */
int gs_neus_backward_points(const float* rays_o, const float* rays_d, const float* z_vals,
                            const float* dists, const void* grid, const float* sdf_w,
                            const float* color_B, float inv_s, const float* inv_s_dev, const float* bound_host,
                            const float* sdf, const float* grad, const uint8_t* mask,
                            const float* d_alpha, const float* d_sdf, const float* d_grad,
                            const void* dX, int dx_dtype, float dx_scale, const float* d_gerr_ray,
                            void* grid_grad, int grid_grad_dtype, float grid_grad_scale, void* d_out,
                            void* lin_in, void* dw0, void* d_arg, void* pts, int row_dtype, float row_scale,
                            int row_stride, float* d_inv_s, int n, int s, const void* enc_aux, gs_stream_t stream);
/* The same backward with the table gradient of the HASHED levels accumulated WITHOUT global atomics (bin-and-reduce:
 * workgroups write (13-bit index, 2 x fp16) records to their own segments of per-(level, bin) queues in `bin_ws`, then
 * one workgroup per bin sums its queue in fp32 LDS accumulators and writes its 8192 entries once; neus_bwd.hip).
 * grid_grad is the loss-scaled f16 table gradient (zero it first; the dense levels and any overflow records still
 * arrive as packed atomics).  `bin_ws`: gs_neus_bin_workspace_bytes(n * s) bytes of scratch (no initial state).
 * `sdf_wt` (optional, f32 [16][2][32]): sdf_w's encoding columns transposed, sdf_wt[l][f][o] = sdf_w[o][3 + 2 l + f]
 * (gs_map_step_prep writes it), which turns the kernel's strided weight reads into contiguous ones.                */
size_t gs_neus_bin_workspace_bytes(int n_points);
int gs_neus_backward_points_binned(const float* rays_o, const float* rays_d, const float* z_vals,
                                   const float* dists, const void* grid, const float* sdf_w, const float* color_B,
                                   float inv_s, const float* inv_s_dev, const float* bound_host, const float* sdf,
                                   const float* grad, const uint8_t* mask, const float* d_alpha, const float* d_sdf,
                                   const float* d_grad, const void* dX, int dx_dtype, float dx_scale,
                                   const float* d_gerr_ray, void* grid_grad, float grid_grad_scale, void* d_out,
                                   void* lin_in, void* dw0, void* d_arg, void* pts, int row_dtype, float row_scale,
                                   int row_stride, float* d_inv_s, int n, int s, void* bin_ws, size_t bin_ws_bytes,
                                   const float* sdf_wt, const void* enc_aux, gs_stream_t stream);

/* The mapper's ray draw for all frames of one joint iteration (src/nerf_func.py:115-181 build_rays after its random pick;
 * src/mapping.py:222-240,262-283 calls it per visited keyframe): ray t = (k, r) of drawn frame k takes the (rank[t] + 1)-th
 * valid pixel p of bank frame f = frame_pos[k] -- the first p with cums[f][p] >= rank[t] + 1, cums = the running sum of the
 * frame's mask over its hw = H * W pixels (row-major), rank[t] < cums[f][hw - 1] -- and writes
 *   rays_d[t] = [(u - cx) / fx, (v - cy) / fy, 1] @ rot_t[f]   (u = p % width, v = p / width; rot_t[f] = c2w[:3,:3]^T)
 *   rays_o[t] = trans[f],  out_color[t] = color[f * hw + p],  out_depth[t] = depth[f * hw + p].
 * rank i64 [n_frames_drawn * n_rays] (the reference's torch.randint draws), frame_pos i32 [n_frames_drawn], cums i32
 * [F][hw], color f32 [F * hw][3], depth f32 [F * hw], rot_t f32 [F][3][3], trans f32 [F][3]; outputs f32.           */
int gs_ray_draw(const long long* rank, const int* frame_pos, const int* cums, const float* color, const float* depth,
                const float* rot_t, const float* trans, int n_frames_drawn, int n_rays, int hw, int width, float fx,
                float fy, float cx, float cy, float* rays_o, float* rays_d, float* out_color, float* out_depth,
                gs_stream_t stream);

/* The mapper's loss without the eikonal term (src/mapping.py:96-132 + InstantNeuS.compute_sdf_error,
 * src/InstantNeuS.py:372-400) and its gradient, one launch.  Rays with rays_depth <= 0 are masked out.
 *   loss_rays[r] = ( w_color |c - c*|_1 / 3 + |d - d*| uw + w_sdf (e_r + f_r) ) / counts[0]
 *   with uw = 1/sqrt(depth_var + 1e-10) if `uncertainty` (treated as a constant, as the reference detaches it),
 *   e_r / f_r the per-ray SDF error / free-space terms; the loss is sum_r loss_rays[r]; counts[0] = number of
 *   valid rays over ALL ranks (device scalar).  d_color [n,3], d_depth [n], d_sdf [n,s] are d(sum loss)/d(.).
 * color [n,3], depth [n], depth_var [n], sdf [n,s], z_vals [n,s] (the sample depths InstantNeuS returns),
 * rays_color [n,3], rays_depth [n]; all f32; s <= 128.                                                      */
int gs_mapping_loss(const float* color, const float* depth, const float* depth_var, const float* sdf,
                    const float* z_vals, const float* rays_color, const float* rays_depth, const float* counts,
                    float truncation, float sparse_factor, float w_color, float w_sdf, int uncertainty,
                    float* d_color, float* d_depth, float* d_sdf, float* loss_rays, int n, int s,
                    gs_stream_t stream);

/* The mapper's optimiser step (src/mapping.py:55-58,135-137: clip_grad_norm_(35) over all trained parameters, then
 * AdamW with lr 1e-2 for the hash table and 1e-3 for the networks) on ONE flat fp32 parameter buffer laid out
 * [hash table (n16 entries) | dense parameters (n - n16)], in two launches.
 *   gs_map_grad_sqnorm: sqnorm_out[0] += sum g^2 (zero it first) over the table gradient g16 (fp16 holding
 *     gradient / inv_scale16, tiny-cuda-nn's loss-scaled form; n16 elements) and the dense gradients g32 (fp32).
 *   gs_map_adamw: coef = min(1, max_norm / (sqrt(sqnorm[0]) + 1e-6)) (sqnorm == NULL: no clipping), then for every
 *     element g' = g * coef, p *= 1 - lr wd, m = b1 m + (1 - b1) g', v = b2 v + (1 - b2) g'^2,
 *     p -= lr / (1 - b1^step) * m / (sqrt(v) / sqrt(1 - b2^step) + eps)   (torch.optim.AdamW), lr = lr16 on the table
 *     and lr32 on the dense range; p16 (optional, fp16 [n]) receives the fp16 working copy of the new parameters.
 *   All buffers 16-byte aligned, n16 % 8 == 0, step >= 1.                                                      */
int gs_map_grad_sqnorm(const void* g16, size_t n16, float inv_scale16, const float* g32, size_t n32,
                       float* sqnorm_out, gs_stream_t stream);
int gs_map_adamw(float* p, float* m, float* v, void* p16, const void* g16, size_t n16, float inv_scale16,
                 const float* g32, size_t n, float lr16, float lr32, float beta1, float beta2, float eps,
                 float weight_decay, int step, const float* sqnorm, float max_norm, gs_stream_t stream);
/* gs_map_adamw_seg: the same update with the two ranges given separately -- (p, m, v, p16, g16, n16) one contiguous
 * run of table entries: the whole table, or the 1/G slice a rank owns when the optimiser state is sharded over the G
 * ranks of a node (reduce-scatter of the table gradient -> this call on the slice -> all-gather of p16); (pd, md, vd,
 * p16d, g32, n32) the dense parameters.  `step_dev` (device int32 >= 1; may be NULL) overrides `step`, so a captured
 * hipGraph replays with the right bias corrections.  n16 need not be a multiple of 8; table pointers 16-byte aligned. */
int gs_map_adamw_seg(float* p, float* m, float* v, void* p16, const void* g16, size_t n16, float inv_scale16,
                     float* pd, float* md, float* vd, void* p16d, const float* g32, size_t n32, float lr16, float lr32,
                     float beta1, float beta2, float eps, float weight_decay, int step, const int* step_dev,
                     const float* sqnorm, float max_norm, gs_stream_t stream);

/* The scalar / reduction arithmetic around the mapper step's kernels in two launches (map_opt.hip):
 *   gs_map_step_prep: counts_out = counts_in if given, else [#rays with depth > 0, n, max depth] of rays_depth [n];
 *     inv_s_out[0] = clamp(exp(variance[0] * scale_factor), 1e-6, 1e6); d_gerr_out[0:n] = w_eikonal / (counts[1] * samples);
 *     d_invs[0] = sqnorm[0] = 0; step_dev[0] += 1; sdf_wt_out[l][f][o] = sdf_w[o][3 + 2 l + f] (both optional);
 *     mlp_wpack_out[i] = frag_index[i] == 10240 ? 0 : mlp16[frag_index[i]] for i < 20480 (all three optional): the
 *     40 A-fragments gs_mlp_backward takes, gathered from the fp16 parameter vector (frag_index: int32 [20480]).
 *   gs_map_step_post: g32 [mlp 10240 | sdf_w 32x35 | sdf_b 32 | color_B 3x33 | variance 1 | loss 1] from the chunked Gram
 *     product of the per-point rows laid out [d_out 32 | x y z 1 .. 8 | lin_in 40 | dw0 40 | d_arg 40] (gram_chunks f32
 *     [nchunk,40,160] = rows[:, :40]^T rows per chunk, summed over chunks, x inv_loss_scale), the MLP
 *     backward's workgroup partials (f32 [nb,10240]), d inv_s, and loss = sum(loss_rays) + w_eikonal * sum(gerr) /
 *     (counts[1] * samples).                                                                                        */
/* rows[:, :40]^T rows of the backward's per-point rows (f16 [n_rows,160], n_rows % 16 == 0, zero rows as padding) on the
 * matrix cores: partial f32 [gs_map_gram_blocks(n_rows)][40][160], one slab per workgroup, only the entries
 * gs_map_step_post reads are written (rows 0..31 x columns 32..95, rows 32..39 x all columns) -- pass it as that
 * function's gram_chunks with nchunk = gs_map_gram_blocks(n_rows).  Replaces a batched library GEMM.               */
int gs_map_gram_blocks(int n_rows);
int gs_map_gram(const void* rows, int n_rows, float* partial, gs_stream_t stream);
int gs_map_step_prep(const float* rays_depth, int n, const float* variance, float scale_factor, float w_eikonal,
                     int samples, const float* counts_in, float* counts_out, float* inv_s_out, float* d_gerr_out,
                     float* d_invs, float* sqnorm, int* step_dev, const float* sdf_w, float* sdf_wt_out,
                     const void* mlp16, const int* frag_index, void* mlp_wpack_out, gs_stream_t stream);
int gs_map_step_post(const float* gram_chunks, int nchunk, float inv_loss_scale, const float* mlp_partial, int nb,
                     const float* d_invs, const float* variance, const float* inv_s, float scale_factor,
                     const float* loss_rays, const float* gerr, int n, float w_eikonal, int samples, const float* counts,
                     float* g32, gs_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GOSLAM_NEUS_H */
