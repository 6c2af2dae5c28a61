// The mapper's ray draw (reference src/nerf_func.py:115-181 `build_rays`, called once per visited keyframe in every joint
// iteration, src/mapping.py:222-240,262-283) for ALL frames of an iteration in one launch.
//
// The random pick stays the reference's own `torch.randint(N_f, (n_rays,))` calls (one per frame, in frame order: a seeded
// run consumes the generator as the reference does); what follows each pick in `build_rays` -- rank -> pixel through the
// mask, pixel -> (u, v), d = K^-1 [u, v, 1] R^T, o = t, colour and depth of the pixel -- was ~20 small launches per
// iteration in neus/rays.RayBank (searchsorted, index arithmetic, stack, bmm, two gathers, casts): host-bound at ~8 us
// each beside a mapper step of 0.7 ms.  Here: one thread per ray, a binary search over its frame's running mask sum
// (19 dependent 4-byte loads for 480 x 640), the direction arithmetic with the reference's operation order
// ((u - cx) / fx with a correctly rounded division), three multiply-adds per output component.
#include "common.h"
#include "../../include/goslam_neus.h"

namespace {

__global__ __launch_bounds__(256) void ray_draw_kernel(const long long* __restrict__ rank, const int* __restrict__ frame_pos,
                                                       const int* __restrict__ cums, const float* __restrict__ color,
                                                       const float* __restrict__ depth, const float* __restrict__ rot_t,
                                                       const float* __restrict__ trans, int nf, int n_rays, int hw,
                                                       int width, float fx, float fy, float cx, float cy,
                                                       float* __restrict__ rays_o, float* __restrict__ rays_d,
                                                       float* __restrict__ out_color, float* __restrict__ out_depth) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= nf * n_rays) return;
  const int k = t / n_rays;
  const int f = frame_pos[k];
  const int want = (int)rank[t] + 1;                 // the (rank + 1)-th valid pixel: first p with cums[p] >= rank + 1
  const int* c = cums + (size_t)f * hw;
  int lo = 0, hi = hw - 1;                           // (the caller guarantees rank < N_f = cums[hw - 1])
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (c[mid] >= want) hi = mid; else lo = mid + 1;
  }
  const int p = lo;
  const float x = (float)(p % width), y = (float)(p / width);
  const float d0 = (x - cx) / fx, d1 = (y - cy) / fy;           // dirs = [(x - cx) / fx, (y - cy) / fy, 1]
  const float* R = rot_t + (size_t)f * 9;                       // rays_d = dirs @ c2w[:3, :3]^T
  const float* T = trans + (size_t)f * 3;
  const size_t g = (size_t)f * hw + p;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    rays_d[(size_t)t * 3 + j] = fmaf(1.0f, R[6 + j], fmaf(d1, R[3 + j], d0 * R[j]));
    rays_o[(size_t)t * 3 + j] = T[j];
    out_color[(size_t)t * 3 + j] = color[g * 3 + j];
  }
  out_depth[t] = depth[g];
}

}  // namespace

extern "C" int gs_ray_draw(const long long* rank, const int* frame_pos, const int* cums, const float* color,
                           const float* depth, const float* rot_t, const float* trans, int n_frames_drawn, int n_rays,
                           int hw, int width, float fx, float fy, float cx, float cy, float* rays_o, float* rays_d,
                           float* out_color, float* out_depth, gs_stream_t stream) {
  GS_REQUIRE(rank && frame_pos && cums && color && depth && rot_t && trans && rays_o && rays_d && out_color && out_depth,
             "ray_draw: null pointer");
  GS_REQUIRE(n_frames_drawn >= 0 && n_rays >= 0 && hw > 0 && width > 0 && hw % width == 0, "ray_draw: bad shape");
  GS_REQUIRE((long long)n_frames_drawn * n_rays < (1ll << 31), "ray_draw: too many rays");
  const int total = n_frames_drawn * n_rays;
  if (total == 0) return GS_OK;
  GS_TIMING_PRE();
  ray_draw_kernel<<<gs_cdiv(total, 256), 256, 0, (hipStream_t)stream>>>(rank, frame_pos, cums, color, depth, rot_t, trans,
                                                                        n_frames_drawn, n_rays, hw, width, fx, fy, cx, cy,
                                                                        rays_o, rays_d, out_color, out_depth);
  GS_CHECK_LAUNCH("ray_draw");
  return GS_OK;
}
