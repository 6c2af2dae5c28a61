// The mapper's loss (reference src/mapping.py:96-132 with InstantNeuS.compute_sdf_error,
// src/InstantNeuS.py:372-400) and its gradient w.r.t. the rendered quantities, in one launch.
//
// As PyTorch code the loss is ~60 small elementwise / reduction launches forward and ~80 more in autograd;
// at the reference's own batch (4096 rays) that was ~1 ms of a 2.6 ms optimisation step.  Every term is a
// masked sum over rays divided by a (global) count, so one wave per ray evaluates the ray's colour, depth and
// SDF terms, reduces its 72 samples with DPP, and writes the analytic gradients d_color, d_depth, d_sdf that
// the render backward consumes.  Counts arrive as device scalars (they may come out of an all-reduce).
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void map_loss_kernel(
    const float* __restrict__ color, const float* __restrict__ depth, const float* __restrict__ dvar,
    const float* __restrict__ sdf, const float* __restrict__ z_vals, const float* __restrict__ rays_color,
    const float* __restrict__ rays_depth, const float* __restrict__ counts /* [0] = global valid rays */,
    float trunc, float sparse, float w_color, float w_sdf, int uncertainty, float* __restrict__ d_color,
    float* __restrict__ d_depth, float* __restrict__ d_sdf, float* __restrict__ loss_rays, int n, int s) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n) return;
  const float gt = rays_depth[r];
  const bool valid = gt > 0.0f;
  const float inv_nv = 1.0f / counts[0];
  // ---- SDF terms over the ray's samples (two per lane: s <= 128)
  float pred[2], bnd[2];
  bool front[2], sm[2], live[2];
  float cnt = 0.0f;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int k = lane + 64 * q;
    live[q] = k < s;
    const float z = live[q] ? z_vals[(size_t)r * s + k] : 0.0f;
    pred[q] = live[q] ? sdf[(size_t)r * s + k] : 0.0f;
    bnd[q] = gt - z;
    front[q] = live[q] && valid && (z < gt - trunc);
    sm[q] = live[q] && valid && (fabsf(bnd[q]) <= trunc);
    cnt += (front[q] ? 1.0f : 0.0f) + (sm[q] ? 1.0f : 0.0f);
  }
  const float nvs = gs_wave_sum(cnt) + 1e-8f;
  float num = 0.0f, g[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const float diff = pred[q] - bnd[q];
    float gq = 0.0f;
    if (sm[q]) {
      num += fabsf(diff);
      gq += (diff > 0.0f) ? 1.0f : ((diff < 0.0f) ? -1.0f : 0.0f);
    }
    if (front[q]) {
      const float arg = -sparse * pred[q];
      const float a = expf(fminf(arg, 10.0f)) - 1.0f;
      const float m = fmaxf(a, diff);
      if (m >= 0.0f) {                                   // clamp(min=0) passes the gradient where m >= 0
        num += m;
        const float da = (arg <= 10.0f) ? -sparse * (a + 1.0f) : 0.0f;    // d/dpred exp(clamp(-sparse pred, max=10))
        if (a > diff) gq += da;
        else if (a < diff) gq += 1.0f;
        else gq += 0.5f * (da + 1.0f);                                       // torch.max splits ties evenly
      }
    }
    g[q] = gq;
  }
  const float sdf_ray = gs_wave_sum(num) / nvs;          // e_ray + f_ray
  const float gs_scale = valid ? w_sdf * inv_nv / nvs : 0.0f;
#pragma unroll
  for (int q = 0; q < 2; ++q)
    if (live[q]) d_sdf[(size_t)r * s + lane + 64 * q] = g[q] * gs_scale;
  // ---- colour / depth terms (lane 0)
  if (lane == 0) {
    float lc = 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float d = color[r * 3 + c] - rays_color[r * 3 + c];
      lc += fabsf(d);
      const float sg = (d > 0.0f) ? 1.0f : ((d < 0.0f) ? -1.0f : 0.0f);
      d_color[r * 3 + c] = valid ? w_color * sg * inv_nv / 3.0f : 0.0f;
    }
    const float dd = depth[r] - gt;
    const float uw = uncertainty ? 1.0f / sqrtf(dvar[r] + 1e-10f) : 1.0f;
    const float sgd = (dd > 0.0f) ? 1.0f : ((dd < 0.0f) ? -1.0f : 0.0f);
    d_depth[r] = valid ? sgd * uw * inv_nv : 0.0f;
    loss_rays[r] = valid ? (w_color * lc / 3.0f + fabsf(dd) * uw + w_sdf * sdf_ray) * inv_nv : 0.0f;
  }
}

}  // namespace

extern "C" int gs_mapping_loss(const float* color, const float* depth, const float* depth_var, const float* sdf,
                               const float* z_vals, const float* rays_color, const float* rays_depth,
                               const float* counts, float truncation, float sparse_factor, float w_color, float w_sdf,
                               int uncertainty, float* d_color, float* d_depth, float* d_sdf, float* loss_rays, int n,
                               int s, gs_stream_t stream) {
  GS_REQUIRE(color && depth && depth_var && sdf && z_vals && rays_color && rays_depth && counts && d_color && d_depth &&
                 d_sdf && loss_rays, "mapping_loss: null pointer");
  GS_REQUIRE(n >= 0 && s > 0 && s <= 128, "mapping_loss: 1 <= samples per ray <= 128 (got %d)", s);
  if (n == 0) return GS_OK;
  GS_TIMING_PRE();
  map_loss_kernel<<<gs_cdiv(n, 4), 256, 0, (hipStream_t)stream>>>(color, depth, depth_var, sdf, z_vals, rays_color,
                                                                 rays_depth, counts, truncation, sparse_factor, w_color,
                                                                 w_sdf, uncertainty, d_color, d_depth, d_sdf, loss_rays,
                                                                 n, s);
  GS_CHECK_LAUNCH("mapping_loss");
  return GS_OK;
}
