/*
 * goslam_hip.h -- C ABI of the MI355X (gfx950) hot-path library `libgoslam_hip.so`.
 *
 * Every entry point replaces one export of the reference's `droid_backends` pybind module
 * (reference: src/lib/droid.cpp:237-250) or one tiny-cuda-nn call made by
 * src/InstantNeuS.py; the reference file:line each one stands in for is cited per function.
 *
 * Conventions
 *   - all pointers are DEVICE pointers unless the name ends in `_host`;
 *   - tensors are dense row-major ("contiguous") with the shapes given in the comments;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); no entry point
 *     synchronises the device, allocates device memory, or throws;
 *   - return value: GS_OK (0) or a negative gs_status; gs_last_error() gives a message for
 *     the calling thread;
 *   - index tensors are int64 (torch.long) exactly as the reference passes them.
 */
#ifndef GOSLAM_HIP_H
#define GOSLAM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* gs_stream_t;

typedef enum {
  GS_OK = 0,
  GS_ERR_INVALID_ARG = -1,
  GS_ERR_WORKSPACE = -2,
  GS_ERR_LAUNCH = -3,
  GS_ERR_UNSUPPORTED = -4
} gs_status;

typedef enum { GS_F16 = 0, GS_F32 = 1, GS_F64 = 2 } gs_dtype;

const char* gs_version(void);
const char* gs_last_error(void);

/* Kernel timer (measurement mode for bench.py's roofline entries; no counterpart in the reference).  Between
 * gs_timing_begin(stream) and gs_timing_end() every kernel this THREAD launches through the library is followed by a
 * HIP event on `stream` (which must be the stream the kernels are launched on); gs_timing_read(name, ...) synchronises
 * and returns the summed duration [ms] and launch count of the kernels recorded under `name` (the launch names used
 * in error messages: "neus_point", "grid_bin_reduce", "conv3x3_pp", ...); gs_timing_names writes the comma-separated
 * names seen.  Nothing is recorded while the stream is being captured into a graph.                                  */
int gs_timing_begin(gs_stream_t stream);
int gs_timing_end(void);
int gs_timing_read(const char* name, double* total_ms, int* count);
int gs_timing_names(char* buf, int buf_bytes);

/* ------------------------------------------------------------------ correlation ---- */

/* droid_backends.corr_index_forward (src/lib/droid.cpp:149-158, correlation_kernels.cu:19-70,
 * 126-155).  volume [n,h1,w1,h2,w2] (dtype), coords f32 [n,2,h1,w1] -> corr [n,2r+1,2r+1,h1,w1]
 * (dtype).  Arithmetic is carried in `dtype` in the reference's accumulation order.        */
int gs_corr_index_forward(const void* volume, const float* coords, void* corr,
                          int n, int h1, int w1, int h2, int w2, int radius, int dtype,
                          gs_stream_t stream);

/* droid_backends.corr_index_backward (droid.cpp:160-171, correlation_kernels.cu:73-124,157-185).
 * volume_grad [n,h1,w1,h2,w2] must be zero-initialised by the caller.                        */
int gs_corr_index_backward(const float* coords, const void* corr_grad, void* volume_grad,
                           int n, int h1, int w1, int h2, int w2, int radius, int dtype,
                           gs_stream_t stream);

/* CorrBlock.__call__ (src/modules/corr.py:43-53) in ONE launch: the 4 pyramid levels
 * vol[l] [n,h1,w1,h2>>l,w2>>l] are sampled at coords/2^l (coords f32 [n,h1,w1,2], the layout
 * FactorGraph hands over) and written as corr [n, 4*(2r+1)^2, h1, w1] (level-major channels);
 * channels_last != 0 stores the same logical tensor with NHWC strides ([n,h1,w1,196] in memory).
 * layout: GS_CORR_ROWMAJOR = the reference's planes; GS_CORR_TILE8 (fp16 + channels_last only) =
 * levels 0-1 as written by gs_corr_volume_pyramid(layout = GS_CORR_TILE8), see below.               */
#define GS_CORR_ROWMAJOR 0
#define GS_CORR_TILE8 1
int gs_corr_lookup_pyramid(const void* vol0, const void* vol1, const void* vol2, const void* vol3,
                           const float* coords, void* corr,
                           int n, int h1, int w1, int h2, int w2, int radius, int dtype,
                           int channels_last, int layout, gs_stream_t stream);
/* The same lookup (fp16 volumes, radius 3) FUSED with corr_encoder[0] (src/droid_net.py:75-77: Conv2d(196, 128, 1) +
 * ReLU): y[n,h1,w1, 0:128] (pixels y_stride elements apart, fp16) = relu(W @ lookup + bias) -- the 196-channel features
 * never reach HBM.  wpad: fp16 [128][208] = the 1x1 weight [128][196] with rows zero-padded to 208; bias f32 [128].
 * Equals gs_corr_lookup_pyramid (bit-exact features) followed by gs_conv1x1 up to the fp32 summation order.      */
int gs_corr_lookup_enc(const void* vol0, const void* vol1, const void* vol2, const void* vol3, const float* coords,
                       const void* wpad, const float* bias, void* y, int y_stride, int n, int h1, int w1, int h2,
                       int w2, int layout, gs_stream_t stream);
/* The `_slots` forms read the planes of edge e from slot `slot[e]` (int64 [n] in device memory, NULL = e) of a volume
 * POOL vol[l] = [capacity, h1, w1, plane_l]: FactorGraph keeps the correlation volumes of its edges in such a pool, so
 * that adding edges (`CorrBlock.cat`, src/factor_graph.py:118 -- a copy of every volume held, 61 MB per edge at 60 x 80)
 * and dropping edges (`self.corr[~mask]`, :150 -- another) move no volume at all.  Same arithmetic as the plain forms.
 * gs_corr_lookup_pyramid_slots serves fp16 + channels_last only.                                                      */
int gs_corr_lookup_enc_slots(const void* vol0, const void* vol1, const void* vol2, const void* vol3, const int64_t* slot,
                             const float* coords, const void* wpad, const float* bias, void* y, int y_stride, int n,
                             int h1, int w1, int h2, int w2, int layout, gs_stream_t stream);
int gs_corr_lookup_pyramid_slots(const void* vol0, const void* vol1, const void* vol2, const void* vol3,
                                 const int64_t* slot, const float* coords, void* corr, int n, int h1, int w1, int h2,
                                 int w2, int radius, int dtype, int channels_last, int layout, gs_stream_t stream);

/* CorrBlock.__init__ + CorrBlock.corr (src/modules/corr.py:26-41,67-76): all-pairs volume of
 * fp16 feature maps fmap1[e], fmap2[e] ([n,128,h,w], both divided by 4) plus the 3 average-pooled
 * levels, each pooled level computed from the fp16-rounded level below (avg_pool2d on half).
 * Outputs vol[l] f16 [n,h,w,h>>l,w>>l].  Requires w % 8 == 0, w <= 96, h >= 8.
 * layout GS_CORR_TILE8 (w % 16 == 0): a private layout for the lookup's benefit -- the planes of levels
 * 0 and 1 are stored as 8x8-element (128-byte = one L2 line) tiles, element (y,x) of a plane at
 * ((y>>3) * ceil(wl/8) + (x>>3)) * 64 + (y&7) * 8 + (x&7), plane size gs_corr_level_elems(); an 8x8
 * lookup window then touches <= 4 lines instead of ~9.  Levels 2-3 stay row-major.  Rows beyond
 * h>>l inside the last tile row are never read and left unwritten.                                 */
size_t gs_corr_volume_workspace_bytes(int n, int dim, int h, int w);
size_t gs_corr_level_elems(int h, int w, int level, int layout);   /* elements per plane of one level */
int gs_corr_volume_pyramid(const void* fmap1, const void* fmap2, void* vol0, void* vol1, void* vol2,
                           void* vol3, int n, int dim, int h, int w, int layout,
                           void* workspace, size_t workspace_bytes, gs_stream_t stream);
/* ... written into slots out_slot[e] (int64 [n] in device memory, NULL = e) of a volume pool (see the lookups above) */
int gs_corr_volume_pyramid_slots(const void* fmap1, const void* fmap2, void* vol0, void* vol1, void* vol2,
                                 void* vol3, const int64_t* out_slot, int n, int dim, int h, int w, int layout,
                                 void* workspace, size_t workspace_bytes, gs_stream_t stream);

/* droid_backends.altcorr_forward (droid.cpp:173-184, altcorr_kernel.cu:27-149,290-319).
 * fmap1 [b,h1,w1,c], fmap2 [b,h2,w2,c] (channels-last, c in {64,128,256}), coords f32
 * [b,s,h1,w1,2] -> corr [b,s,(2r+1)^2,h1,w1]; dtype f16 or f32, fp32 accumulation.             */
int gs_altcorr_forward(const void* fmap1, const void* fmap2, const float* coords, void* corr,
                       int b, int s, int h1, int w1, int h2, int w2, int c, int radius, int dtype,
                       gs_stream_t stream);

/* droid_backends.altcorr_backward (droid.cpp:186-199, altcorr_kernel.cu:151-283,322-354): fp32 only, as the
 * reference instantiates it.  fmap1 [b,h1,w1,c], fmap2 [b,h2,w2,c], coords [b,s,h1,w1,2], corr_grad
 * [b,s,49,h1,w1] -> fmap1_grad (written), fmap2_grad (atomically accumulated; zero it first).  coords get
 * no gradient (the reference returns zeros).  c <= 256.                                              */
int gs_altcorr_backward(const float* fmap1, const float* fmap2, const float* coords, const float* corr_grad,
                        float* fmap1_grad, float* fmap2_grad, int b, int s, int h1, int w1, int h2, int w2,
                        int c, int radius, gs_stream_t stream);

/* AltCorrBlock.__call__ for one chunk of edges (src/modules/corr.py:95-145 as driven by FactorGraph.update_lowmem,
 * src/factor_graph.py:283-300): all four pyramid levels of altcorr_forward (altcorr_kernel.cu:27-149) in ONE launch,
 * features indexed by ii / jj inside the kernel (the reference gathers `pyramid[0][:, ii]`, `pyramid[i][:, jj]` per level)
 * and coords / 2^l formed in registers.
 *   pyr0..pyr3 f16 [N, h >> l, w >> l, c] channels-last (c = 128), coords f32 [e, h, w, 2], ii / jj i64 [e] (rows of the
 *   pyramid) -> out f16 [e, h, w, 196] (= the [e,196,h,w] tensor of the reference in channels-last memory order;
 *   channel = 49 l + 7 ix + iy).  fp16 products, fp32 accumulation and blend, one rounding: within half an fp16 ulp of
 *   gs_altcorr_forward on fp16 features.  h, w >= 8.                                                         */
int gs_altcorr_pyramid(const void* pyr0, const void* pyr1, const void* pyr2, const void* pyr3, const float* coords,
                       const int64_t* ii, const int64_t* jj, void* out, int e, int h, int w, int c, int radius,
                       gs_stream_t stream);

/* ------------------------------------------------------------------- geometry ------ */

/* DepthVideo.reproject -> pops.projective_transform(jacobian=False)
 * (src/depth_video.py:207-217, src/geom/projective_ops.py:114-144).
 * poses f32 [nbuf,7], disps f32 [nbuf,h,w], intrinsics f32 [nbuf,4], ii/jj i64 [n]
 * -> coords f32 [n,h,w,2], valid f32 [n,h,w,1].                                              */
int gs_reproject(const float* poses, const float* disps, const float* intrinsics,
                 const int64_t* ii, const int64_t* jj, float* coords, float* valid,
                 int n, int h, int w, gs_stream_t stream);

/* droid_backends.projmap (droid.cpp:133-140, droid_kernels.cu:427-516,1463-1488).
 * intrinsics f32 [4]; coords f32 [n,h,w,3] (channel 2 left 0), valid f32 [n,h,w,1].          */
int gs_projmap(const float* poses, const float* disps, const float* intrinsics,
               const int64_t* ii, const int64_t* jj, float* coords, float* valid,
               int n, int h, int w, gs_stream_t stream);

/* droid_backends.frame_distance (droid.cpp:120-131, droid_kernels.cu:518-657,1438-1460).    */
int gs_frame_distance(const float* poses, const float* disps, const float* intrinsics,
                      const int64_t* ii, const int64_t* jj, float* dist,
                      int n, int h, int w, float beta, gs_stream_t stream);

/* droid_backends.iproj (droid.cpp:141-147, droid_kernels.cu:779-850,1518-1541).
 * poses f32 [n,7], disps f32 [n,h,w] -> points f32 [n,h,w,3].                                */
int gs_iproj(const float* poses, const float* disps, const float* intrinsics, float* points,
             int n, int h, int w, gs_stream_t stream);

/* droid_backends.depth_filter (droid.cpp:186-196 region, droid_kernels.cu:661-775,1491-1515).
 * disps f32 [num,h,w], ix i64 [n], thresh f32 [n] -> counter f32 [n,h,w] (zeroed here).      */
int gs_depth_filter(const float* poses, const float* disps, const float* intrinsics,
                    const int64_t* ix, const float* thresh, float* counter,
                    int n, int num, int h, int w, gs_stream_t stream);

/* ------------------------------------------- update-operator gate fusions ---- */

/* ConvGRU gates of src/modules/gru.py:20-33 around MIOpen's convolutions; all NHWC fp16.
 * gs_gru_gate_zr: zr_pre f16 [n,hw,256] = fused convz|convr output WITHOUT bias, bias_zr f32 [256],
 *   glo_zr f32 [n,256] (the 1x1 global-context terms), hx f16 [n,hw,ldx] whose first 128 channels
 *   hold `net` on entry and r*net on exit, z_out f16 [n,hw,128].
 * gs_gru_gate_q : q_pre f16 [n,hw,128] (convq output without bias), bias_q f32 [128], glo_q f32
 *   [n,128], z / net f16 [n,hw,128] -> net_out = (1-z)*net + z*tanh(q_pre + bias + glo).
 * inp_pre (may be NULL): f16 [n,hw,384] = the z | r | q convolutions restricted to the context
 *   features `inp`, which are constant while an edge lives; by linearity they are hoisted out of the
 *   update loop and added to the pre-activations here (zr_pre / q_pre then cover net, corr, flow).  */
int gs_gru_gate_zr(const void* zr_pre, const float* bias_zr, const float* glo_zr, const void* inp_pre, void* hx,
                   void* z_out, int n, int hw, int ldx, gs_stream_t stream);
int gs_gru_gate_q(const void* q_pre, const float* bias_q, const float* glo_q, const void* inp_pre, const void* z,
                  const void* net, void* net_out, int n, int hw, gs_stream_t stream);
/* y[row, 0:channels] = act(x[row, 0:channels] + bias) for NHWC fp16 data viewed as [rows, channels]
 * (channels % 8 == 0); x / y rows are x_stride / y_stride elements apart, so y may be x itself (in
 * place) and either side may be a channel slice of a wider NHWC tensor (replaces torch.cat / split).  bias may be
 * NULL (plain strided copy when act == 0).  act: 0 none, 1 ReLU, 2 sigmoid.
 * Epilogue of the bias-free MIOpen convolutions of src/droid_net.py:69-140.                       */
int gs_bias_act(const void* x, const float* bias, void* y, int rows, int channels, int x_stride, int y_stride,
                int act, gs_stream_t stream);
/* FactorGraph.update glue (src/factor_graph.py:201-207): out [n,h,w,4] fp16 (NHWC; logical [n,4,h,w]) =
 * clamp([coords1 - pixel grid, target - coords1], -64, 64), coords1 / target f32 [n,h,w,2].            */
int gs_motion_features(const float* coords1, const float* target, void* out, int n, int h, int w,
                       gs_stream_t stream);
/* The per-chunk glue of FactorGraph.update_lowmem (src/factor_graph.py:283-312), one launch each instead of ~11 torch
 * launches per chunk of 13 source keyframes:
 *   gs_lowmem_gather: rows sel[r] (i64 [n_sel], edge indices) of coords1 / target f32 [E,h,w,2] and net f16 [E,h,w,128]
 *     (NHWC) -> coords_out f32 [n_sel,h,w,2] (= coords1[:, v]), motion_out f16 [n_sel,h,w,4] (= clamp([coords1 - grid,
 *     target - coords1], +-64): gs_motion_features of the chunk), net_out f16 [n_sel,h,w,128] (= self.net[:, v]).
 *   gs_lowmem_scatter: target[sel[r]] = coords + delta, weight_all[sel[r]] = weight (all f32 [.,h,w,2]),
 *     net[sel[r]] = net_new (f16 NHWC) -- the three masked assignments of :308-310.  sel must not repeat an index.     */
int gs_lowmem_gather(const float* coords1, const float* target, const void* net, const int64_t* sel,
                     float* coords_out, void* motion_out, void* net_out, int n_sel, int h, int w, gs_stream_t stream);
int gs_lowmem_scatter(const float* coords, const float* delta, const float* weight, const void* net_new,
                      const int64_t* sel, float* target, float* weight_all, void* net, int n_sel, int h, int w,
                      gs_stream_t stream);

/* FactorGraph.update glue (src/factor_graph.py:222-223,244-247): target [n,h,w,2] = coords1 + delta, plus
 * the [n,2,h,w] copies of target and weight that droid_backends.ba takes (ba_target / ba_weight point at
 * the first of these n edges inside the caller's [E_all,2,h,w] buffers).  All f32.                    */
int gs_ba_inputs(const float* coords1, const float* delta, const float* weight, float* target,
                 float* ba_target, float* ba_weight, int n, int h, int w, gs_stream_t stream);
/* 1x1 convolution + bias + activation as one memory-bound MFMA GEMM: y[p, 0:n_out] = act(W x[p, 0:k_in] + b)
 * over `rows` NHWC fp16 pixels (corr_encoder[0] 196->128 ReLU, src/droid_net.py:75; GraphAgg upmask
 * 128->576, src/droid_net.py:45).  x / y rows are x_stride / y_stride elements apart.  k_in % 4 == 0,
 * k_in <= 208; n_out % 32 == 0.  wpack: fp16 [n_out/32][KS][64][8] A-fragments with KS = 8 (k_in <= 128)
 * or 13, wpack[nb][ks][l][e] = W[32 nb + (l & 31)][16 ks + 8 (l >> 5) + e] (0 for k >= k_in).
 * bias may be NULL; act: 0 none, 1 ReLU.                                                            */
int gs_conv1x1(const void* x, int x_stride, int k_in, const void* wpack, const float* bias, int act, void* y,
               int y_stride, int n_out, long long rows, gs_stream_t stream);
/* act(conv7x7(x) + bias), padding 3, stride 1, for a FOUR-channel NHWC fp16 map [n,h,w,4] -> 128 channels (pixels of
 * y `ys` halves apart, 128 written): the flow encoder's first layer over the motion features (reference
 * src/droid_net.py:79 `Conv2d(4, 128, 7, padding=3)` + ReLU).  fp32 accumulation starting from the bias, one rounding to
 * fp16, then ReLU if `relu`.  K is laid out as 7 kernel rows x 8 taps x 4 channels (the 8th tap is zero) = 14 MFMA
 * k-steps; wpack: fp16 [2][2][14][64][8],
 * wpack[nh][t][s][l][e] = Wk[64 nh + 32 t + (l & 31)][16 s + 8 (l >> 5) + e] with Wk[o][32 ky + 4 kx + c] = W[o][c][ky][kx]
 * (0 for kx = 7).  rt = image rows per workgroup (0: chosen from the map size); w <= 1024.              */
int gs_conv7x7_c4(const void* x, const void* wpack, const float* bias, void* y, int ys, int n, int h, int w, int relu,
                  int rt, gs_stream_t stream);
/* 3x3 convolution, padding 1, stride 1, no bias: NHWC fp16 [n,h,w,c_in] (pixels x_stride elements apart) ->
 * NHWC fp16 [n,h,w,n_out] (y_stride apart), fp32 accumulation, as an implicit GEMM on MFMA.  Covers the large
 * convolutions of the update operator: ConvGRU convz|convr and convq (src/modules/gru.py:10-12), the first layers of
 * the delta / weight / agg heads (src/droid_net.py:83,88,40), corr_encoder[2] and flow_encoder[2] (:76,:80).
 * Row-stacked tiling: the n images are tiled as one image of n*h rows, tiles of 512 / tw rows x tw columns (tw in
 * {8, 16}: pick the width that divides w best, 40 -> 8, 80 -> 16) run across image boundaries and the vertical taps
 * are masked per pixel at y == 0 / y == h-1, so no tile padding is spent on h; the images must be contiguous (image
 * stride = h * w * x_stride resp. y_stride).  512-thread workgroups whose two wave groups alternate a fragment-read
 * phase and an MFMA phase one phase apart ("ping-pong"), all staging by LDS-DMA (global_load_lds) with counted waits
 * across raw barriers, an XOR-swizzled bank-conflict-free patch layout.  xcd_order != 0: workgroup ids are decoded so
 * that the output-channel blocks of a tile run on one XCD (shared L2).
 * n_out % 64 == 0 (BN = 128 output channels per workgroup, or 64 when n_out is only a multiple of 64), c_in % 32 == 0.
 * wpack: fp16 [n_out/BN][c_in/32][9][4][BN][8],
 * wpack[nb][ck][3 ky + kx][kg][r][e] = W[BN nb + r][32 ck + 8 kg + e][ky][kx] (gs_conv3x3_wpack_elems halves). */
size_t gs_conv3x3_wpack_elems(int c_in, int n_out);
int gs_conv3x3_pp(const void* x, int x_stride, int c_in, const void* wpack, int tw, void* y, int y_stride, int n_out,
                  int n, int h, int w, int xcd_order, gs_stream_t stream);
/* ConvGRU (src/modules/gru.py:20-33) with the gate arithmetic fused into the 3x3 convolutions' epilogues
 * (results equal gs_conv3x3_pp + gs_gru_gate_zr / gs_gru_gate_q to one fp16 ulp on < 1e-4 of the elements; the 256 + 128
 * channels of pre-activations never travel to HBM and back).
 *   gs_conv3x3_gru_zr: hx [n,h,w,hx_stride] fp16, first c_in channels = [net(128) | rest]; wpack = gs_conv3x3_pp image of
 *     the fused convz|convr weight [256, c_in, 3, 3]; bias_zr f32 [256]; glo_zr f32 [n,256]; inp_pre fp16
 *     [n,h,w,384] or NULL.  Writes z_out = sigmoid(.) [n,h,w,128] and rnet_out = r * net [n,h,w,128]; hx is not modified.
 *   gs_conv3x3_gru_q: input [rnet (128 ch, dense) | x_rest (c_rest channels, pixels x_rest_stride apart)]; wpack = image
 *     of convq [128, 128 + c_rest, 3, 3] (c_rest % 32 == 0); writes net_out = (1 - z) net + z tanh(.).   */
/* gs_conv3x3_pp + bias + ReLU in one kernel; y / y_stride may address a channel slice of a wider NHWC tensor
 * (corr_encoder[2] / flow_encoder[2] -> the GRU input buffer, src/droid_net.py:76,80; agg.conv2, :41).             */
int gs_conv3x3_bias_relu(const void* x, int x_stride, int c_in, const void* wpack, const float* bias, void* y,
                         int y_stride, int n_out, int n, int h, int w, gs_stream_t stream);
/* gs_conv3x3_gru_zr2: the same with the input given as [net (128 channels, dense rows) | x_rest (c_rest channels, pixels
 * x_rest_stride apart)] -- no copy of net into the input buffer before the step (as gs_conv3x3_gru_q).          */
int gs_conv3x3_gru_zr2(const void* net, const void* x_rest, int x_rest_stride, int c_rest, const void* wpack,
                       const float* bias_zr, const float* glo_zr, const void* inp_pre, void* z_out, void* rnet_out,
                       int n, int h, int w, gs_stream_t stream);
int gs_conv3x3_gru_zr(const void* hx, int hx_stride, int c_in, const void* wpack, const float* bias_zr,
                      const float* glo_zr, const void* inp_pre, void* z_out, void* rnet_out, int n, int h, int w,
                      gs_stream_t stream);
int gs_conv3x3_gru_q(const void* rnet, const void* x_rest, int x_rest_stride, int c_rest, const void* wpack,
                     const float* bias_q, const float* glo_q, const void* inp_pre, const void* z, const void* net,
                     void* net_out, int n, int h, int w, gs_stream_t stream);
/* 3x3 convolution (padding 1) from 128 channels to n_out in {1,2} channels, NHWC fp16 in, fp32 out
 * [n,h,w,n_out]: the flow-revision / confidence heads delta[2], weight[2] (src/droid_net.py:83-92) and
 * GraphAgg's eta[0] (src/droid_net.py:43).  x rows are x_stride elements apart (a channel slice of a
 * wider tensor is fine).  If in_bias != NULL or in_relu, the operand is relu?(x + in_bias) applied on
 * the fly (the producer convolution's epilogue).  wpack: fp16 [8][64][8] MFMA A-fragments,
 * wpack[ks][l][e] = W[o][16 ks + 8 (l>>5) + e][ky][kx] with (l & 31) = (3 ky + kx) n_out + o (0 beyond
 * 9 n_out).  out = out_scale * epi(half(conv + bias)); epilogue: 0 none, 1 sigmoid (rounded to fp16),
 * 2 softplus (fp32).                                                                              */
int gs_conv3x3_head(const void* x, int x_stride, const float* in_bias, int in_relu, const void* wpack,
                    const float* bias, int n_out, int epilogue, float out_scale, float* out, int n, int h, int w,
                    gs_stream_t stream);
/* GraphAgg's scatter_mean over source keyframes (src/droid_net.py:57-60, torch_scatter):
 * out[s, p, :] = mean over k in [seg_offsets[s], seg_offsets[s+1]) of act(x[seg_edges[k], p, :]), NHWC
 * fp16 [*, hw, channels] (channels % 8 == 0, x rows x_stride elements apart), fp32 accumulation.
 * act = relu?(. + in_bias) rounded to fp16 when in_bias != NULL or in_relu (the producer
 * convolution's epilogue applied on the fly), identity otherwise.                                   */
int gs_segment_mean(const void* x, int x_stride, const float* in_bias, int in_relu, const int* seg_offsets,
                    const int* seg_edges, void* out, int n_seg, int hw, int channels, gs_stream_t stream);
/* Frame encoders (src/modules/extractor.py:27-57 ResidualBlock.forward, :113-126 BasicEncoder.forward): the
 * elementwise tail of every convolution of `fnet` (norm_fn='instance') / `cnet` (norm_fn='none') in three launches
 * (one without the norm) instead of torch's batch_norm_collect_statistics + calc_invstd + transform_input + clamp
 * + add + clamp (and the bias add of the convolution before them).  x, skip, y: NHWC fp16 [n, hw, channels], y may
 * alias x or skip; bias: fp16 [channels] or NULL.
 *   x = bias ? half(x + bias_c) : x                           the convolution's bias, added to its fp16 output
 *   t = instance_norm ? half((x - mean_c) * invstd_c) : x     InstanceNorm2d(affine=False): per image and channel,
 *                                                             biased variance, fp32 statistics, eps inside the sqrt
 *   t = relu_in  ? max(t, 0) : t
 *   t = skip     ? half(skip + t) : t
 *   y = relu_out ? max(t, 0) : t
 * workspace: gs_norm_act_workspace_bytes bytes (not needed when instance_norm == 0).  stat_chunks > 0: the statistics pass
 * is skipped -- gs_enc_conv, given the same workspace as `stats_ws`, already wrote stat_chunks = gs_enc_conv_stat_chunks
 * (h_out, w_out, c_out) partial-sum slabs per image in its epilogue (workspace: gs_norm_act_workspace_bytes_chunks). */
size_t gs_norm_act_workspace_bytes(int n, int hw, int channels);
size_t gs_norm_act_workspace_bytes_chunks(int n, int chunks, int channels);
/* The frame encoders' convolutions (src/modules/extractor.py:61-126: BasicEncoder.conv1 7x7 / stride 2, the residual
 * blocks' 3x3 convolutions with stride 1 / 2, their strided 1x1 skips, conv2 1x1), NHWC fp16 -> NHWC fp16 with fp32
 * accumulation and ONE fp16 rounding (+ a second one after the fp16 bias add when `bias` f16 [c_out] is given -- the
 * rounding points of a library convolution followed by its bias kernel).  Built shapes (ksize, c_in, c_out, stride):
 * (7,4,32,2) -- the stem on a dense 4-channel RGB0 image --, (3,32,32,1), (3,32,64,2), (3,64,64,1), (3,64,128,2),
 * (3,128,128,1), (1,32,64,2), (1,64,128,2), (1,128,128,1), (1,128,256,1); padding = ksize / 2; anything else returns
 * GS_ERR_UNSUPPORTED.  wpack: MFMA A-fragments in lane order, f16 [ksize^2][c_in/16][c_out/32][64][8] with element
 * [t][s][m][l][e] = W[32 m + (l & 31)][16 s + 8 (l >> 5) + e][t / ksize][t % ksize]; the stem: [7][2][64][8] with
 * [dy][s][l][e] = W[l & 31][e % 4][dy][4 s + 2 (l >> 5) + e / 4] (0 for channel 3 and tap 7).
 * stats_ws (optional): the gs_norm_act workspace of the InstanceNorm that follows -- the epilogue leaves per workgroup
 * and channel the sums of d and d^2, d = half(conv + stat_bias) - stat_bias (stat_bias f16 [c_out], optional), there, and
 * gs_norm_act is then called with the SAME bias and stat_chunks = gs_enc_conv_stat_chunks(h_out, w_out, c_out).     */
size_t gs_enc_conv_wpack_elems(int ksize, int c_in, int c_out);
int gs_enc_conv_stat_chunks(int h_out, int w_out, int c_out);
int gs_enc_conv(const void* x, int x_stride, int c_in, const void* wpack, const void* bias, void* y, int y_stride,
                int c_out, int ksize, int stride, int n, int h, int w, const void* stat_bias, void* stats_ws,
                gs_stream_t stream);
int gs_norm_act(const void* x, const void* bias, const void* skip, void* y, int n, int hw, int channels,
                int instance_norm, int relu_in, int relu_out, float eps, void* workspace, size_t workspace_bytes,
                int stat_chunks, gs_stream_t stream);
/* ConvGRU global context (src/modules/gru.py:22-27): glo = mean_hw(sigmoid(w_pre + w_bias) * net),
 * then the three 1x1 convolutions convz_glo | convr_glo (-> gzr [n,256]) and convq_glo (-> gq [n,128]).
 * w_pre = bias-free 1x1 conv of net, NHWC fp16 [n,hw,128]; wz/wr/wq fp16 [128 out,128 in]; outputs f32
 * holding fp16-rounded values (what autocast produces).                                           */
size_t gs_gru_glo_workspace_bytes(int n);
int gs_gru_glo(const void* w_pre, const float* w_bias, const void* net, const void* wz, const void* wr,
               const void* wq, const float* bz, const float* br, const float* bq, float* gzr, float* gq,
               int n, int hw, void* workspace, size_t workspace_bytes, gs_stream_t stream);

/* The same with the 1x1 convolution w(net) inside (MFMA; gru.py:22 `self.w`): w_pre never exists in memory, net is
 * read once.  net: NHWC fp16 [n,hw,*] with pixels net_stride halves apart (first 128 channels used); w_pack = gs_conv1x1's
 * weight image of w ([4][8][64][8] halves); the pre-activation is rounded once, fp16(conv + w_bias).  Deterministic.  */
size_t gs_gru_glo_fused_workspace_bytes(int n, int hw);
int gs_gru_glo_fused(const void* net, int net_stride, const void* w_pack, const float* w_bias, const void* wz,
                     const void* wr, const void* wq, const float* bz, const float* br, const float* bq, float* gzr,
                     float* gq, int n, int hw, void* workspace, size_t workspace_bytes, gs_stream_t stream);

/* FactorGraph.update's damping rows (src/factor_graph.py:228,244): damping_buf[index[k]] = eta[inv[k]] where
 * inv[k] >= 0, then out[k] = scale * damping_buf[index[k]] + eps, rows of hw floats; eta [*,hw] (may be NULL when no
 * inv[k] >= 0), inv int32 [n_rows], index int64 [n_rows] (distinct frames), out [n_rows,hw].                  */
int gs_damping_rows(const float* eta, const int* inv, const int64_t* index, float* damping_buf, float* out,
                    int n_rows, int hw, float scale, float eps, gs_stream_t stream);

/* DepthVideo.upsample -> cvx_upsample (src/depth_video.py:194-196, src/droid_net.py:9-23):
 * out[ix[n]] (f32 [*,8h,8w]) = convex 8x upsampling of disps[ix[n]] (f32 [*,h,w]) with the softmax
 * of mask f16 [m,576,h,w] (logical NCHW; mask_channels_last != 0: NHWC strides).  ix i64 [m] or
 * NULL (identity).                                                                            */
int gs_cvx_upsample(const float* disps, const void* mask, const int64_t* ix, float* out,
                    int m, int h, int w, int mask_channels_last, gs_stream_t stream);
/* GraphAgg's upmask convolution (Conv2d(128, 576, 1), src/droid_net.py:45,62) fused with the convex upsampling above:
 * x fp16 [m*h*w, x_stride] (NHWC agg features, first 128 channels), weight fp16 [576][128], bias f32 [576];
 * out[ix[n]] = cvx_upsample(disps[ix[n]], half(W x + b)).  The 576-channel mask is never materialised.              */
int gs_upmask_upsample(const void* x, int x_stride, const void* weight, const float* bias, const float* disps,
                       const int64_t* ix, float* out, int m, int h, int w, gs_stream_t stream);

/* ------------------------------------------------------ dense bundle adjustment ---- */

/* Workspace size for gs_ba (bytes).  n_edges = len(ii), n_poses = t1-t0, n_depth = rows of
 * eta (= |unique(cat(arange(t0,t1), ii))|; pass an upper bound when unknown), nbuf = poses
 * rows, hw = h*w.                                                                           */
size_t gs_ba_workspace_bytes(int n_edges, int n_poses, int n_depth, int nbuf, int hw);

/* droid_backends.ba (droid.cpp:88-117, droid_kernels.cu:1314-1434 incl. the host-side
 * SparseBlock / schur_block / Eigen LLT of :1117-1311, all on the device here).
 *   poses f32 [nbuf,7] (in/out), disps f32 [nbuf,h,w] (in/out), intrinsics f32 [4],
 *   disps_sens f32 [nbuf,h,w], targets/weights f32 [n_edges,2,h,w], eta f32 [n_depth,h,w]
 *   (ignored when motion_only), ii/jj i64 [n_edges];
 *   dx f32 [t1-t0,6] and dz f32 [n_depth,h*w] receive the last iteration's update.
 * A Cholesky failure leaves dx = 0 for that iteration as the reference does (:1202-1210).
 * status_out (optional, device int32[4]): [0] = #depth keyframes found, [1] = 1 if it
 * differs from n_depth, [2] = #Cholesky failures, [3] reserved.                             */
int gs_ba(float* poses, float* disps, const float* intrinsics, const float* disps_sens,
          const float* targets, const float* weights, const float* eta,
          const int64_t* ii, const int64_t* jj,
          int t0, int t1, int iterations, float lm, float ep, int motion_only,
          int n_edges, int n_depth, int nbuf, int h, int w,
          float* dx, float* dz, int32_t* status_out,
          void* workspace, size_t workspace_bytes, gs_stream_t stream);

/* gs_ba with flags.  GS_BA_REUSE_TABLES (1): `workspace` still holds the index tables (unique keyframes, CSR of
 * out-edges, Schur entry lists) an earlier call built for the SAME ii / jj / t0 / t1 / n_depth / nbuf / map size and
 * nothing has written to it since -- they depend on nothing else, so the one-workgroup table kernel (26 us, once per
 * call) is skipped.  FactorGraph.update issues 6 calls per keyframe on one edge set.                       */
#define GS_BA_REUSE_TABLES 1
int gs_ba_ex(float* poses, float* disps, const float* intrinsics, const float* disps_sens,
             const float* targets, const float* weights, const float* eta,
             const int64_t* ii, const int64_t* jj,
             int t0, int t1, int iterations, float lm, float ep, int motion_only,
             int n_edges, int n_depth, int nbuf, int h, int w,
             float* dx, float* dz, int32_t* status_out,
             void* workspace, size_t workspace_bytes, int flags, gs_stream_t stream);

/* Edge proposal with greedy non-maximum suppression on the device: FactorGraph.add_proximity_factors
 * (src/factor_graph.py:384-450) and Backend.ba's selection incl. the loop-closure rule (src/backend.py:31-94), in two
 * launches around a device-side stable sort; the frame-distance matrix never leaves HBM.
 *   raw f32 [(t-i0) x (t-j0)]: frame distances of keyframe pairs (i0 + row, j0 + col).
 *   gs_edge_prep: d_work = raw with the non-candidates (i - rad < j, raw > cut) at +inf, +inf in the (2 nms + 1)^2
 *     windows of the existing edges ex_i / ex_j (i64, n_existing; pairs outside the window are ignored) and of the
 *     local-window edges (i, j), (j, i) for j in [max(i - rad, jmin), i) -- which are written to es (i64 [cap,2], with
 *     (i, i) first per keyframe if `stereo`) in the reference's order; count[0] = number of edges written.
 *   (the caller sorts d_work ascending, stable: sorted_vals f32, order i64)
 *   gs_edge_greedy: visits the candidates with sorted value <= thresh in order; a candidate that an earlier pick has
 *     not suppressed appends (i, j), (j, i) -- in `loop` mode instead the members (si != sj) of its 3x3 neighbourhood
 *     with raw <= thresh, and only if more than 4 of the 9 are -- and suppresses its window; stops as soon as
 *     count > max_factors.  count[0] is updated.  At most 512 x 512 candidate pairs.                       */
int gs_edge_prep(const float* raw, float* d_work, const long long* ex_i, const long long* ex_j, int n_existing,
                 long long* es, int* count, int cap, int i0, int j0, int t, int rad, int nms, float cut, int stereo,
                 int jmin, gs_stream_t stream);
int gs_edge_greedy(const float* raw, const float* sorted_vals, const long long* order, long long* es, int* count,
                   int cap, int i0, int j0, int t, int nms, float thresh, int max_factors, int loop, gs_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GOSLAM_HIP_H */
