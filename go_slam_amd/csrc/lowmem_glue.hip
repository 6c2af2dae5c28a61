// The per-chunk glue of FactorGraph.update_lowmem (reference src/factor_graph.py:283-312: for each block of 13 source
// keyframes `coords1[:, v]`, `self.net[:, v]`, the motion features of the chunk, and after the update operator the three
// boolean-mask assignments `self.net[:, v] = net`, `self.target[:, v] = coords1[:, v] + delta`, `self.weight[:, v] =
// weight`).  As torch ops these are ~11 launches per chunk -- index_select x 3, cat / permute / clamp / cast / layout copy
// of the motion features, add, index_put x 3 -- i.e. ~180 of the 746 launches of a 200-keyframe step and ~1.5 ms of it
// (profiles/r05_global_ba_stress_kernel_stats.md).  Here: ONE gather launch and ONE scatter launch per chunk.
//
//   gs_lowmem_gather   rows sel[r] of coords1 / target / net (NHWC fp16) -> the chunk's coords (fp32), its motion features
//                      clamp([coords1 - grid, target - coords1], +-64) as the NHWC fp16 tensor the flow encoder reads
//                      (the arithmetic of gs_motion_features), and its recurrent state
//   gs_lowmem_scatter  target[sel[r]] = coords + delta, weight[sel[r]] = weight, net[sel[r]] = the operator's new state
//
// HBM-bound row copies: a workgroup owns 64 pixels of one edge -- 64 x 8 B of coordinates per tensor (lanes 0..63) and the
// 64 x 256 B of the 128-channel state as 1024 contiguous 16-byte pieces (four per thread).
#include "common.h"

namespace {

typedef _Float16 half4g __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void lowmem_gather_kernel(const float2* __restrict__ coords1,
                                                            const float2* __restrict__ target,
                                                            const uint4* __restrict__ net, const int64_t* __restrict__ sel,
                                                            float2* __restrict__ c_out, _Float16* __restrict__ motion,
                                                            uint4* __restrict__ net_out, int hw, int w) {
  const int r = blockIdx.y;
  const size_t e = (size_t)sel[r];
  const int p0 = blockIdx.x * 64;
  const int np = min(64, hw - p0);
  // the state's four 16-byte pieces per thread are requested FIRST and behind no branch (index clamped, the stores stay
  // predicated): each behind its own `if`, the compiler waited for one before it issued the next -- the copy of 23 MB per
  // call ran as four dependent round trips per thread, ~1 TB/s
  const uint4* src = net + (e * hw + p0) * 16;             // 128 halves = 16 pieces of 16 B per pixel
  uint4* dst = net_out + ((size_t)r * hw + p0) * 16;
  uint4 v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = src[min((int)threadIdx.x + 256 * k, np * 16 - 1)];
  const int pc = p0 + min((int)threadIdx.x, np - 1);
  const float2 c_in = coords1[e * hw + pc], tg_in = target[e * hw + pc];
  if (threadIdx.x < 64) {   // (wave 0, no lane predicate: lanes past the tile's last pixel / piece repeat the last one -- the
                            // same values to the same addresses; a predicated store would pull its load back behind the branch)
    const int p = pc;
    const float2 c = c_in, tg = tg_in;
    c_out[(size_t)r * hw + p] = c;
    const float gx = (float)(p % w), gy = (float)(p / w);
    half4g o;
    o[0] = (_Float16)fminf(fmaxf(c.x - gx, -64.0f), 64.0f);
    o[1] = (_Float16)fminf(fmaxf(c.y - gy, -64.0f), 64.0f);
    o[2] = (_Float16)fminf(fmaxf(tg.x - c.x, -64.0f), 64.0f);
    o[3] = (_Float16)fminf(fmaxf(tg.y - c.y, -64.0f), 64.0f);
    reinterpret_cast<half4g*>(motion)[(size_t)r * hw + p] = o;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) dst[min((int)threadIdx.x + 256 * k, np * 16 - 1)] = v[k];
}

__global__ __launch_bounds__(256) void lowmem_scatter_kernel(const float2* __restrict__ coords, const float2* __restrict__ delta,
                                                             const float2* __restrict__ weight,
                                                             const uint4* __restrict__ net_new,
                                                             const int64_t* __restrict__ sel, float2* __restrict__ target,
                                                             float2* __restrict__ weight_all, uint4* __restrict__ net,
                                                             int hw) {
  const int r = blockIdx.y;
  const size_t e = (size_t)sel[r];
  const int p0 = blockIdx.x * 64;
  const int np = min(64, hw - p0);
  const uint4* src = net_new + ((size_t)r * hw + p0) * 16;  // (loads first and behind no branch: see lowmem_gather_kernel)
  uint4* dst = net + (e * hw + p0) * 16;
  uint4 v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = src[min((int)threadIdx.x + 256 * k, np * 16 - 1)];
  const size_t ic = (size_t)r * hw + p0 + min((int)threadIdx.x, np - 1);
  const float2 c = coords[ic], d = delta[ic], wv = weight[ic];
  if (threadIdx.x < 64) {
    const size_t o = e * hw + p0 + min((int)threadIdx.x, np - 1);
    target[o] = make_float2(c.x + d.x, c.y + d.y);
    weight_all[o] = wv;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) dst[min((int)threadIdx.x + 256 * k, np * 16 - 1)] = v[k];
}

}  // namespace

extern "C" int gs_lowmem_gather(const float* coords1, const float* target, const void* net, const int64_t* sel,
                                float* coords_out, void* motion_out, void* net_out, int n_sel, int h, int w,
                                gs_stream_t stream) {
  GS_REQUIRE(coords1 && target && net && sel && coords_out && motion_out && net_out, "lowmem_gather: null pointer");
  GS_REQUIRE(n_sel >= 0 && h > 0 && w > 0, "lowmem_gather: bad shape");
  if (n_sel == 0) return GS_OK;
  GS_REQUIRE(n_sel <= 65535, "lowmem_gather: %d edges exceed the grid limit", n_sel);
  const int hw = h * w;
  lowmem_gather_kernel<<<dim3(gs_cdiv(hw, 64), n_sel), 256, 0, (hipStream_t)stream>>>(
      (const float2*)coords1, (const float2*)target, (const uint4*)net, sel, (float2*)coords_out, (_Float16*)motion_out,
      (uint4*)net_out, hw, w);
  GS_CHECK_LAUNCH("lowmem_gather");
  return GS_OK;
}

extern "C" int gs_lowmem_scatter(const float* coords, const float* delta, const float* weight, const void* net_new,
                                 const int64_t* sel, float* target, float* weight_all, void* net, int n_sel, int h, int w,
                                 gs_stream_t stream) {
  GS_REQUIRE(coords && delta && weight && net_new && sel && target && weight_all && net, "lowmem_scatter: null pointer");
  GS_REQUIRE(n_sel >= 0 && h > 0 && w > 0, "lowmem_scatter: bad shape");
  if (n_sel == 0) return GS_OK;
  GS_REQUIRE(n_sel <= 65535, "lowmem_scatter: %d edges exceed the grid limit", n_sel);
  const int hw = h * w;
  lowmem_scatter_kernel<<<dim3(gs_cdiv(hw, 64), n_sel), 256, 0, (hipStream_t)stream>>>(
      (const float2*)coords, (const float2*)delta, (const float2*)weight, (const uint4*)net_new, sel, (float2*)target,
      (float2*)weight_all, (uint4*)net, hw);
  GS_CHECK_LAUNCH("lowmem_scatter");
  return GS_OK;
}
