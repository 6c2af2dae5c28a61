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
