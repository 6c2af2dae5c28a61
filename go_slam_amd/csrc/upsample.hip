// Convex 8x upsampling of the disparity maps (reference src/droid_net.py:9-23 cvx_upsample via
// DepthVideo.upsample, src/depth_video.py:194-196): softmax over the 9 neighbours of every one of
// the 64 sub-pixels, then the weighted sum of the 3x3 coarse neighbourhood.
//
// The reference's formulation (view + softmax + unfold + mul + sum + permute + index_put) costs
// ~1.1 ms per update at 25 keyframes of 60x80 on MI355X, 17 % of an update; here one wave serves
// one coarse pixel (lane = sub-pixel), reads its 9 logits with 9 coalesced 128-B loads when the
// mask is NHWC (the layout the update operator's 1x1 conv emits), and writes 8 rows of 8 floats.
// The softmax weights are rounded to the mask dtype exactly like torch.softmax on a half tensor.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void cvx_upsample_kernel(const float* __restrict__ disps,
                                                           const _Float16* __restrict__ mask,
                                                           const int64_t* __restrict__ ix, float* __restrict__ out,
                                                           int m, int h, int w, long cs, long ps) {
  const int lane = threadIdx.x & 63;
  const long pix = (long)blockIdx.x * 4 + (threadIdx.x >> 6);     // n * h*w + p
  const int hw = h * w;
  if (pix >= (long)m * hw) return;
  const int n = (int)(pix / hw), p = (int)(pix - (long)n * hw);
  const int y = p / w, x = p - y * w;
  const long frame = ix ? ix[n] : n;
  const _Float16* mk = mask + (long)n * 576 * hw + (long)p * ps;
  float lg[9], mx = -INFINITY;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    lg[k] = (float)mk[(long)(k * 64 + lane) * cs];
    mx = fmaxf(mx, lg[k]);
  }
  float den = 0.f;
#pragma unroll
  for (int k = 0; k < 9; ++k) { lg[k] = __expf(lg[k] - mx); den += lg[k]; }
  const float* d = disps + frame * hw;
  const float inv_den = __builtin_amdgcn_rcpf(den);
  float acc = 0.f;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;              // F.unfold(3x3, padding 1): zero padded
    const float nb = (yy >= 0 && yy < h && xx >= 0 && xx < w) ? d[yy * w + xx] : 0.0f;
    const float wk = (float)(_Float16)(lg[k] * inv_den);           // softmax output is fp16
    acc += wk * nb;
  }
  const int i = lane >> 3, j = lane & 7;
  out[frame * (long)hw * 64 + (long)(8 * y + i) * (8 * w) + 8 * x + j] = acc;
}

// NHWC masks (the update operator's layout): 8 lanes per coarse pixel, lane c owns row c of the 8x8
// sub-pixel block, so every tap is one 16-byte load per lane (a wave reads 8 pixels x 128 B per
// instruction instead of 128 B in 2-byte pieces) and the 8 results leave as two 16-byte stores.
typedef _Float16 half8u __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void cvx_upsample_nhwc_kernel(const float* __restrict__ disps,
                                                                const _Float16* __restrict__ mask,
                                                                const int64_t* __restrict__ ix,
                                                                float* __restrict__ out, int m, int h, int w) {
  const int lane = threadIdx.x & 63;
  const long pix = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + (lane >> 3);     // n * h*w + p
  const int hw = h * w;
  if (pix >= (long)m * hw) return;
  const int c = lane & 7;
  const int n = (int)(pix / hw), p = (int)(pix - (long)n * hw);
  const int y = p / w, x = p - y * w;
  const long frame = ix ? ix[n] : n;
  const _Float16* mk = mask + pix * 576 + 8 * c;
  half8u lg[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) lg[k] = *reinterpret_cast<const half8u*>(mk + k * 64);
  const float* d = disps + frame * hw;
  float nb[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;              // F.unfold(3x3, padding 1): zero padded
    nb[k] = (yy >= 0 && yy < h && xx >= 0 && xx < w) ? d[yy * w + xx] : 0.0f;
  }
  float res[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float e[9], mx = -INFINITY, den = 0.f, acc = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) { e[k] = (float)lg[k][j]; mx = fmaxf(mx, e[k]); }
#pragma unroll
    for (int k = 0; k < 9; ++k) { e[k] = __expf(e[k] - mx); den += e[k]; }
    const float inv_den = __builtin_amdgcn_rcpf(den);             // one v_rcp instead of nine IEEE division sequences
#pragma unroll
    for (int k = 0; k < 9; ++k) acc += (float)(_Float16)(e[k] * inv_den) * nb[k];   // softmax output is fp16
    res[j] = acc;
  }
  float4* o = reinterpret_cast<float4*>(out + frame * (long)hw * 64 + (long)(8 * y + c) * (8 * w) + 8 * x);
  o[0] = make_float4(res[0], res[1], res[2], res[3]);
  o[1] = make_float4(res[4], res[5], res[6], res[7]);
}

// ---- GraphAgg's upmask 1x1 convolution (128 -> 576, src/droid_net.py:45,62) FUSED with the convex upsampling ------------
// Unfused, the 576-channel mask is written (138 MB for 25 keyframes of 60x80) only to be read back by the kernel above.
// Here wave w of an 8-wave workgroup computes the logits of sub-pixel ROW w of the 8x8 block: 8 sub-pixels x 9 taps = 72
// mask channels (rows 8 k + i of its weight slice = channel 64 k + 8 w + i; 3 MFMA row tiles, the last one a quarter
// full) for 32 coarse pixels per step on the matrix cores, weights resident in 96 VGPRs.  One tap = exactly one
// 8-row group of the 32x32 accumulator layout, so lane (pixel r, half h) ends up with all 9 taps of sub-pixels 4 h .. 4 h + 3:
// softmax and the 3x3 weighted sum are lane-local and the result is ONE 16-byte store.  Two waves per SIMD: one wave's
// softmax arithmetic (the bulk: 36 exponentials per lane and step) overlaps the other's MFMAs -- the first version
// (4 waves x 144 channels, one wave per SIMD) was bound by exactly that arithmetic, 74 us.
// Same arithmetic and rounding points as gs_conv1x1 -> fp16 mask -> gs_cvx_upsample (logits and softmax weights in fp16).
typedef _Float16 up_h8 __attribute__((ext_vector_type(8)));
typedef float up_f16v __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512, 1) void upmask_upsample_kernel(const _Float16* __restrict__ x, int xs,
                                                                 const _Float16* __restrict__ wgt,
                                                                 const float* __restrict__ bias,
                                                                 const float* __restrict__ disps,
                                                                 const int64_t* __restrict__ ix, float* __restrict__ out,
                                                                 int m, int h, int w) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;       // wv = sub-pixel row sy
  const int r = lane & 31, kgl = lane >> 5;
  const int hw = h * w;
  const long total = (long)m * hw;
  up_h8 wa[3][8];
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const int rowp = 32 * t + r;                                  // 8 k + i
    const int ch = (rowp >> 3) * 64 + 8 * wv + (rowp & 7);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const up_h8 z = {0, 0, 0, 0, 0, 0, 0, 0};
      wa[t][s] = rowp < 72 ? *reinterpret_cast<const up_h8*>(wgt + (size_t)ch * 128 + 16 * s + 8 * kgl) : z;
    }
  }
  float bv[9][4];                                                 // bias of tap k, sub-pixel column 4 kgl + e
#pragma unroll
  for (int k = 0; k < 9; ++k)
#pragma unroll
    for (int e = 0; e < 4; ++e) bv[k][e] = bias[64 * k + 8 * wv + 4 * kgl + e];
  const long gstride = (long)gridDim.x * 32;
  up_h8 bnext[8];
  long fnext;                                        // the frame of the next block's pixel: requested with its rows
  const int64_t* ixp = ix ? ix : reinterpret_cast<const int64_t*>(disps);   // (no index list: any readable words, unused)
  {
    const long g0 = (long)blockIdx.x * 32;
    const long pq0 = (g0 + r < total) ? g0 + r : (g0 < total ? g0 : 0);
    const int n0 = (int)(pq0 / hw);
    { const long f = ixp[n0]; fnext = ix ? f : (long)n0; }
    const _Float16* xr0 = x + (size_t)pq0 * xs + 8 * kgl;
#pragma unroll
    for (int s = 0; s < 8; ++s) bnext[s] = *reinterpret_cast<const up_h8*>(xr0 + 16 * s);
  }
  for (long g0 = (long)blockIdx.x * 32; g0 < total; g0 += gstride) {
    const long pix = g0 + r;
    const bool valid = pix < total;
    const long pq = valid ? pix : g0;
    const int n = (int)(pq / hw), p = (int)(pq - (long)n * hw);
    const long frame = fnext;                        // (the frame index is the head of a two-step chain: it arrives with the rows)
    up_h8 bcur[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) bcur[s] = bnext[s];
    {
      const long g1 = g0 + gstride;
      const long pq1 = (g1 + r < total) ? g1 + r : (g1 < total ? g1 : pq);
      const int n1 = (int)(pq1 / hw);
      { const long f = ixp[n1]; fnext = ix ? f : (long)n1; }
      const _Float16* xr1 = x + (size_t)pq1 * xs + 8 * kgl;
#pragma unroll
      for (int s = 0; s < 8; ++s) bnext[s] = *reinterpret_cast<const up_h8*>(xr1 + 16 * s);
    }
    // the pixel's 3 x 3 disparities are requested BEFORE the mask's MFMAs (frame index -> nine loads is a chain of two
    // round trips that sat exposed between the MFMAs and the softmax), behind no branch: coordinates clamped, zeros selected
    const int y = p / w, xq = p - y * w;
    const float* d = disps + frame * hw;
    float nb[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const int yy = y + k / 3 - 1, xx = xq + k % 3 - 1;          // F.unfold(3x3, padding 1): zero padded
      const bool in = yy >= 0 && yy < h && xx >= 0 && xx < w;
      const float v = d[min(max(yy, 0), h - 1) * w + min(max(xx, 0), w - 1)];
      nb[k] = in ? v : 0.0f;
    }
    up_f16v acc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[t][e] = 0.0f;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
#pragma unroll
      for (int t = 0; t < 3; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[t][s], bcur[s], acc[t], 0, 0, 0);
    }
    float res[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float lg[9], mx = -INFINITY, den = 0.f, a = 0.f;
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        // row' = 8 k + i, i = 4 kgl + e -> tile k / 4, accumulator slot 4 (k % 4) + e
        lg[k] = (float)(_Float16)(acc[k >> 2][4 * (k & 3) + e] + bv[k][e]);                     // the fp16 mask value
        mx = fmaxf(mx, lg[k]);
      }
#pragma unroll
      for (int k = 0; k < 9; ++k) { lg[k] = __expf(lg[k] - mx); den += lg[k]; }
      const float inv_den = __builtin_amdgcn_rcpf(den);
#pragma unroll
      for (int k = 0; k < 9; ++k) a += (float)(_Float16)(lg[k] * inv_den) * nb[k];             // softmax output is fp16
      res[e] = a;
    }
    if (valid)
      *reinterpret_cast<float4*>(out + frame * (long)hw * 64 + (long)(8 * y + wv) * (8 * w) + 8 * xq + 4 * kgl) =
          make_float4(res[0], res[1], res[2], res[3]);
  }
}

}  // namespace

extern "C" int gs_cvx_upsample(const float* disps, const void* mask, const int64_t* ix, float* out, int m, int h,
                               int w, int mask_channels_last, gs_stream_t stream) {
  GS_REQUIRE(disps && mask && out, "cvx_upsample: null pointer");
  GS_REQUIRE(m >= 0 && h > 0 && w > 0, "cvx_upsample: bad shape");
  if (m == 0) return GS_OK;
  const long hw = (long)h * w;
  const long cs = mask_channels_last ? 1 : hw, ps = mask_channels_last ? 576 : 1;
  const long total = (long)m * hw;
  if (mask_channels_last) {
    cvx_upsample_nhwc_kernel<<<(unsigned)((total + 31) / 32), 256, 0, (hipStream_t)stream>>>(
        disps, (const _Float16*)mask, ix, out, m, h, w);
    GS_CHECK_LAUNCH("cvx_upsample");
    return GS_OK;
  }
  cvx_upsample_kernel<<<(unsigned)((total + 3) / 4), 256, 0, (hipStream_t)stream>>>(
      disps, (const _Float16*)mask, ix, out, m, h, w, cs, ps);
  GS_CHECK_LAUNCH("cvx_upsample");
  return GS_OK;
}

extern "C" int gs_upmask_upsample(const void* x, int x_stride, const void* weight, const float* bias, const float* disps,
                                  const int64_t* ix, float* out, int m, int h, int w, gs_stream_t stream) {
  GS_REQUIRE(x && weight && bias && disps && out, "upmask_upsample: null pointer");
  GS_REQUIRE(m >= 0 && h > 0 && w > 0 && x_stride >= 128 && x_stride % 8 == 0, "upmask_upsample: bad shape");
  GS_REQUIRE((8 * w) % 4 == 0, "upmask_upsample: bad width");
  if (m == 0) return GS_OK;
  const long total = (long)m * h * w;
  long blocks = (total + 31) / 32;
  if (blocks > 256) blocks = 256;                     // one long-lived workgroup per CU: its 147 KB of weights are loaded once
  upmask_upsample_kernel<<<(unsigned)blocks, 512, 0, (hipStream_t)stream>>>(
      (const _Float16*)x, x_stride, (const _Float16*)weight, bias, disps, ix, out, m, h, w);
  GS_CHECK_LAUNCH("upmask_upsample");
  return GS_OK;
}
