// Per-pixel geometry kernels: reproject, projmap, frame_distance, iproj, depth_filter.
// Reference: src/geom/projective_ops.py:26-144 and src/lib/droid_kernels.cu:427-850.
//
// All five are HBM/latency bound (a few loads and ~100 flops per pixel).  One lane owns one
// pixel; grids are (pixel chunks, edges) so even a single edge spreads over many CUs instead
// of the reference's one 256-thread block per edge.  The relative pose of an edge is
// recomputed per lane from 14 uniform floats (the compiler keeps them in SGPRs) rather than
// via a thread-0 prologue + barrier.  Compiled with -ffp-contract=off so that masks, floors
// and counts agree bit-for-bit with the op-by-op fp32 oracle.
#include "common.h"

namespace {

constexpr float kMinDepthKernel = 0.25f;   // droid_kernels.cu:26
constexpr float kMinDepthPy = 0.2f;        // geom/projective_ops.py:4

__device__ __forceinline__ void load_pose(const float* poses, int k, float* t, float* q) {
  const float* p = poses + (size_t)k * 7;
  t[0] = p[0]; t[1] = p[1]; t[2] = p[2];
  q[0] = p[3]; q[1] = p[4]; q[2] = p[5]; q[3] = p[6];
}

// ---- DepthVideo.reproject (projective_ops.py:114-144, jacobian=False) -------------------
__global__ __launch_bounds__(256) void reproject_kernel(
    const float* __restrict__ poses, const float* __restrict__ disps, const float* __restrict__ intr,
    const int64_t* __restrict__ ii, const int64_t* __restrict__ jj, float* __restrict__ coords,
    float* __restrict__ valid, int hw, int wd) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  const int e = blockIdx.y;
  if (p >= hw) return;
  const int ix = (int)ii[e], jx = (int)jj[e];
  float tij[3], qij[4];
  if (ix == jx) {   // stereo pair override (projective_ops.py:124)
    tij[0] = -0.1f; tij[1] = 0.f; tij[2] = 0.f;
    qij[0] = 0.f; qij[1] = 0.f; qij[2] = 0.f; qij[3] = 1.f;
  } else {          // Gij = Gj * Gi^-1 with lietorch's group ops
    float ti[3], qi[4], tj[3], qj[4];
    load_pose(poses, ix, ti, qi);
    load_pose(poses, jx, tj, qj);
    float qinv[4] = {-qi[0], -qi[1], -qi[2], qi[3]};
    float r[3], tinv[3];
    gs_act_so3(qinv, ti, r);
    tinv[0] = -r[0]; tinv[1] = -r[1]; tinv[2] = -r[2];
    gs_act_so3(qj, tinv, r);
    tij[0] = tj[0] + r[0]; tij[1] = tj[1] + r[1]; tij[2] = tj[2] + r[2];
    gs_quat_mul(qj, qinv, qij);
  }
  const float fxi = intr[ix * 4 + 0], fyi = intr[ix * 4 + 1], cxi = intr[ix * 4 + 2], cyi = intr[ix * 4 + 3];
  const float fxj = intr[jx * 4 + 0], fyj = intr[jx * 4 + 1], cxj = intr[jx * 4 + 2], cyj = intr[jx * 4 + 3];
  const float u = (float)(p % wd), v = (float)(p / wd);
  float X0[4] = {(u - cxi) / fxi, (v - cyi) / fyi, 1.0f, disps[(size_t)ix * hw + p]};
  float X1[4];
  gs_act_se3(tij, qij, X0, X1);
  const float Z = (X1[2] < 0.5f * kMinDepthPy) ? 1.0f : X1[2];
  const size_t o = (size_t)e * hw + p;
  reinterpret_cast<float2*>(coords)[o] = make_float2(fxj * (X1[0] / Z) + cxj, fyj * (X1[1] / Z) + cyj);
  valid[o] = ((X1[2] > kMinDepthPy) && (X0[2] > kMinDepthPy)) ? 1.0f : 0.0f;
}

// ---- projmap (droid_kernels.cu:427-516) --------------------------------------------------
__global__ __launch_bounds__(256) void projmap_kernel(
    const float* __restrict__ poses, const float* __restrict__ disps, const float* __restrict__ intr,
    const int64_t* __restrict__ ii, const int64_t* __restrict__ jj, float* __restrict__ coords,
    float* __restrict__ valid, int hw, int wd) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  const int e = blockIdx.y;
  if (p >= hw) return;
  const int ix = (int)ii[e], jx = (int)jj[e];
  float ti[3], qi[4], tj[3], qj[4], tij[3], qij[4];
  load_pose(poses, ix, ti, qi);
  load_pose(poses, jx, tj, qj);
  gs_rel_se3(ti, qi, tj, qj, tij, qij);
  const float fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];
  const float u = (float)(p % wd), v = (float)(p / wd);
  float Xi[4] = {(u - cx) / fx, (v - cy) / fy, 1.0f, disps[(size_t)ix * hw + p]};
  float Xj[4];
  gs_act_se3(tij, qij, Xi, Xj);
  float cu = u, cv = v;
  if (Xj[2] > 0.01f) {
    cu = fx * (Xj[0] / Xj[2]) + cx;
    cv = fy * (Xj[1] / Xj[2]) + cy;
  }
  const size_t o = (size_t)e * hw + p;
  coords[o * 3 + 0] = cu;
  coords[o * 3 + 1] = cv;
  coords[o * 3 + 2] = 0.0f;
  valid[o] = (Xj[2] > kMinDepthKernel) ? 1.0f : 0.0f;
}

// ---- frame_distance (droid_kernels.cu:518-657): one workgroup per frame pair -------------
__global__ __launch_bounds__(256) void frame_distance_kernel(
    const float* __restrict__ poses, const float* __restrict__ disps, const float* __restrict__ intr,
    const int64_t* __restrict__ ii, const int64_t* __restrict__ jj, float* __restrict__ dist,
    int hw, int wd, float beta) {
  const int e = blockIdx.x;
  const int ix = (int)ii[e], jx = (int)jj[e];
  float ti[3], qi[4], tj[3], qj[4], tij[3], qij[4];
  load_pose(poses, ix, ti, qi);
  load_pose(poses, jx, tj, qj);
  gs_rel_se3(ti, qi, tj, qj, tij, qij);
  const float fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];
  float accum = 0.f, valid = 0.f;
  const float* dsp = disps + (size_t)ix * hw;
  for (int p = threadIdx.x; p < hw; p += 256) {
    const float u = (float)(p % wd), v = (float)(p / wd);
    float Xi[4] = {(u - cx) / fx, (v - cy) / fy, 1.0f, dsp[p]};
    float Xj[4];
    gs_act_se3(tij, qij, Xi, Xj);
    float du = fx * (Xj[0] / Xj[2]) + cx - u;
    float dv = fy * (Xj[1] / Xj[2]) + cy - v;
    float d = sqrtf(du * du + dv * dv);
    if (Xj[2] > kMinDepthKernel) { accum += beta * d; valid += beta; }
    // translation-only flow
    float Yx = Xi[0] + Xi[3] * tij[0];
    float Yy = Xi[1] + Xi[3] * tij[1];
    float Yz = Xi[2] + Xi[3] * tij[2];
    du = fx * (Yx / Yz) + cx - u;
    dv = fy * (Yy / Yz) + cy - v;
    d = sqrtf(du * du + dv * dv);
    if (Yz > kMinDepthKernel) { accum += (1.0f - beta) * d; valid += (1.0f - beta); }
  }
  __shared__ float red[2][4];
  accum = gs_wave_sum(accum);
  valid = gs_wave_sum(valid);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[0][wave] = accum; red[1][wave] = valid; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float a = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    const float vl = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    const float total = (float)hw * (beta + (1.0f - beta));
    dist[e] = (vl / (total + 1e-8f) < 0.75f) ? 1000.0f : a / vl;
  }
}

// ---- iproj (droid_kernels.cu:779-850) ----------------------------------------------------
__global__ __launch_bounds__(256) void iproj_kernel(
    const float* __restrict__ poses, const float* __restrict__ disps, const float* __restrict__ intr,
    float* __restrict__ points, int hw, int wd) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  const int n = blockIdx.y;
  if (p >= hw) return;
  float t[3], q[4];
  load_pose(poses, n, t, q);
  const float fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];
  const float u = (float)(p % wd), v = (float)(p / wd);
  float Xi[4] = {(u - cx) / fx, (v - cy) / fy, 1.0f, disps[(size_t)n * hw + p]};
  float Xj[4];
  gs_act_se3(t, q, Xi, Xj);
  float* o = points + ((size_t)n * hw + p) * 3;
  o[0] = Xj[0] / Xj[3];
  o[1] = Xj[1] / Xj[3];
  o[2] = Xj[2] / Xj[3];
}

// ---- depth_filter (droid_kernels.cu:661-775) ---------------------------------------------
// One lane per pixel loops over the 6 neighbours itself, so the count is a plain register
// sum (the reference issues one float atomicAdd per neighbour hit).
__global__ __launch_bounds__(256) void depth_filter_kernel(
    const float* __restrict__ poses, const float* __restrict__ disps, const float* __restrict__ intr,
    const int64_t* __restrict__ inds, const float* __restrict__ thresh, float* __restrict__ counter,
    int num, int hw, int ht, int wd) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (p >= hw) return;
  const int ix = (int)inds[b];
  const float th = thresh[b];
  const float fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];
  const float ui = (float)(p % wd), vi = (float)(p / wd);
  const float di = disps[(size_t)ix * hw + p];
  float ti[3], qi[4];
  load_pose(poses, ix, ti, qi);
  float count = 0.f;
#pragma unroll
  for (int neigh = 0; neigh < 6; ++neigh) {
    const int jx = (neigh < 3) ? ix - neigh - 1 : ix + neigh;
    if (jx < 0 || jx >= num) continue;
    float tj[3], qj[4], tij[3], qij[4];
    load_pose(poses, jx, tj, qj);
    gs_rel_se3(ti, qi, tj, qj, tij, qij);
    float Xi[4] = {(ui - cx) / fx, (vi - cy) / fy, 1.0f, di};
    float Xj[4];
    gs_act_se3(tij, qij, Xi, Xj);
    const float uj = fx * (Xj[0] / Xj[2]) + cx;
    const float vj = fy * (Xj[1] / Xj[2]) + cy;
    const float dj = Xj[3] / Xj[2];
    const float fu = floorf(uj), fv = floorf(vj);
    if (fu >= 0.f && fv >= 0.f && fu < (float)(wd - 1) && fv < (float)(ht - 1)) {
      const int u0 = (int)fu, v0 = (int)fv;
      const float* dj_map = disps + (size_t)jx * hw + (size_t)v0 * wd + u0;
      const double inv = 1.0 / (double)dj;
      const double t = (double)th;
      if (fabs(inv - 1.0 / (double)dj_map[0]) < t) count += 1.0f;
      else if (fabs(inv - 1.0 / (double)dj_map[1]) < t) count += 1.0f;
      else if (fabs(inv - 1.0 / (double)dj_map[wd]) < t) count += 1.0f;
      else if (fabs(inv - 1.0 / (double)dj_map[wd + 1]) < t) count += 1.0f;
    }
  }
  counter[(size_t)b * hw + p] = count;
}

}  // namespace

#define GS_GEOM_COMMON(name)                                                              \
  GS_REQUIRE(n >= 0 && h > 0 && w > 0, name ": bad shape");                               \
  if (n == 0) return GS_OK;                                                               \
  hipStream_t st = (hipStream_t)stream;                                                   \
  const int hw = h * w;

extern "C" int gs_reproject(const float* poses, const float* disps, const float* intrinsics,
                            const int64_t* ii, const int64_t* jj, float* coords, float* valid, int n, int h,
                            int w, gs_stream_t stream) {
  GS_REQUIRE(poses && disps && intrinsics && ii && jj && coords && valid, "reproject: null pointer");
  GS_GEOM_COMMON("reproject");
  GS_REQUIRE(n <= 65535, "reproject: n=%d exceeds grid.y limit", n);
  reproject_kernel<<<dim3(gs_cdiv(hw, 256), n), 256, 0, st>>>(poses, disps, intrinsics, ii, jj, coords, valid, hw, w);
  GS_CHECK_LAUNCH("reproject");
  return GS_OK;
}

extern "C" int gs_projmap(const float* poses, const float* disps, const float* intrinsics, const int64_t* ii,
                          const int64_t* jj, float* coords, float* valid, int n, int h, int w,
                          gs_stream_t stream) {
  GS_REQUIRE(poses && disps && intrinsics && ii && jj && coords && valid, "projmap: null pointer");
  GS_GEOM_COMMON("projmap");
  GS_REQUIRE(n <= 65535, "projmap: n=%d exceeds grid.y limit", n);
  projmap_kernel<<<dim3(gs_cdiv(hw, 256), n), 256, 0, st>>>(poses, disps, intrinsics, ii, jj, coords, valid, hw, w);
  GS_CHECK_LAUNCH("projmap");
  return GS_OK;
}

extern "C" int gs_frame_distance(const float* poses, const float* disps, const float* intrinsics,
                                 const int64_t* ii, const int64_t* jj, float* dist, int n, int h, int w,
                                 float beta, gs_stream_t stream) {
  GS_REQUIRE(poses && disps && intrinsics && ii && jj && dist, "frame_distance: null pointer");
  GS_GEOM_COMMON("frame_distance");
  frame_distance_kernel<<<n, 256, 0, st>>>(poses, disps, intrinsics, ii, jj, dist, hw, w, beta);
  GS_CHECK_LAUNCH("frame_distance");
  return GS_OK;
}

extern "C" int gs_iproj(const float* poses, const float* disps, const float* intrinsics, float* points, int n,
                        int h, int w, gs_stream_t stream) {
  GS_REQUIRE(poses && disps && intrinsics && points, "iproj: null pointer");
  GS_GEOM_COMMON("iproj");
  GS_REQUIRE(n <= 65535, "iproj: n=%d exceeds grid.y limit", n);
  iproj_kernel<<<dim3(gs_cdiv(hw, 256), n), 256, 0, st>>>(poses, disps, intrinsics, points, hw, w);
  GS_CHECK_LAUNCH("iproj");
  return GS_OK;
}

extern "C" int gs_depth_filter(const float* poses, const float* disps, const float* intrinsics,
                               const int64_t* ix, const float* thresh, float* counter, int n, int num, int h,
                               int w, gs_stream_t stream) {
  GS_REQUIRE(poses && disps && intrinsics && ix && thresh && counter, "depth_filter: null pointer");
  GS_GEOM_COMMON("depth_filter");
  GS_REQUIRE(n <= 65535 && num > 0, "depth_filter: bad n/num");
  depth_filter_kernel<<<dim3(gs_cdiv(hw, 256), n), 256, 0, st>>>(poses, disps, intrinsics, ix, thresh, counter, num, hw, h, w);
  GS_CHECK_LAUNCH("depth_filter");
  return GS_OK;
}
