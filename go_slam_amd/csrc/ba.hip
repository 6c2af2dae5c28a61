// Dense bundle adjustment on the device (reference: src/lib/droid_kernels.cu:176-424, 854-1434).
//
// The reference runs one 256-thread block per EDGE, spills per-pixel E/C/b rows, and then
// assembles the Schur complement on the HOST (argsort/CSR, O(P^2 deg^2) pair loop, Eigen
// sparse LLT) with ~20 blocking D2H copies per Gauss-Newton iteration.  Here the whole
// iteration stays on the GPU and is organised around the SOURCE keyframe instead of the edge:
// every depth-related quantity (C, w, Q, sum_e Eii) is indexed by a pixel of the source
// keyframe, so a lane that owns pixel p of keyframe k and loops over k's out-edges holds all
// of them in registers -- the reference's four `accum_cuda` passes disappear.
//
//   ba_prep_kernel     1 workgroup : kx = unique(cat(arange(t0,t1), ii)), CSR of out-edges per
//                                    depth keyframe, Schur entry lists + pair prefix (once/call)
//   ba_accum_kernel    (chunks, M) : Jacobians; per-edge 12x12 Hessian blocks reduced per wave
//                                    (DPP) and added to the dense fp64 system with hardware
//                                    fp64 atomics; Q, w, Ei, Eij rows written for the next stage
//   ba_schur_kernel    persistent  : S = E Q E^T block pairs and E Q w, subtracted in fp64
//   (chol.hip)                     : LM damping, blocked fp64 Cholesky, triangular solves
//   ba_update_kernel   (chunks, M) : dz = Q (w - E^T dx) with the reference's `<= 0` skip,
//                                    disparity update, SE3 retraction of the poses
//
// No host synchronisation, no allocation: everything lives in the caller's workspace.
#include "common.h"

// chol.hip
int gs_chol_solve_launch(double* H, double* b, int n, float lm, float ep, float* dx_out, int32_t* fail_flag,
                         int32_t* fail_count, int32_t* sync, hipStream_t st);

namespace {

constexpr float kMinDepth = 0.25f;   // droid_kernels.cu:26
constexpr float kAlpha = 0.05f;      // droid_kernels.cu:1396
constexpr int kPrepThreads = 1024;
constexpr int kAccSplit = 4;         // ba_accum: workgroups that share one depth keyframe's out-edge list (round 6)
constexpr int kMaxBuf = 4096;        // frames addressable by the prep kernel's LDS tables

struct BaWs {
  int32_t* hdr;       // [16] 0:M_dev 1:mismatch 2:chol_fail(this iter) 3:chol_fail_total 4:n_pairs
  int32_t* kinv;      // [nbuf]   frame -> depth row or -1
  int32_t* kx;        // [M]      depth row -> frame
  int32_t* edge_m;    // [E]      depth row of ii[e]
  int32_t* row_ptr;   // [M+1]    CSR over out-edges
  int32_t* csr_edge;  // [E]
  int32_t* ent_ptr;   // [M+1]    Schur entries (pose in window) per depth row
  int32_t* ent_src;   // [M+E]    row into Ei (m) or M + e into Eij
  int32_t* ent_pose;  // [M+E]    pose index in [0,P)
  int32_t* pair_ptr;  // [M+1]    prefix of n_m^2
  float* Q;           // [M,HW]
  float* W;           // [M,HW]
  float* Ei;          // [M,6,HW]
  float* Eij;         // [E,6,HW]
  float* part;        // [kAccSplit,M,8,HW] per-pixel partial sums (C, w, Ei) of keyframes whose edge list is split
  double* H;          // [6P,6P] lower triangle used
  double* b;          // [6P]
  size_t total;
};

BaWs carve(void* base, int E, int P, int M, int nbuf, int hw) {
  BaWs w;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += gs_align(bytes); return (char*)base + o; };
  w.hdr = (int32_t*)take(16 * 4);
  w.kinv = (int32_t*)take((size_t)nbuf * 4);
  w.kx = (int32_t*)take((size_t)(M + 1) * 4);
  w.edge_m = (int32_t*)take((size_t)(E + 1) * 4);
  w.row_ptr = (int32_t*)take((size_t)(M + 2) * 4);
  w.csr_edge = (int32_t*)take((size_t)(E + 1) * 4);
  w.ent_ptr = (int32_t*)take((size_t)(M + 2) * 4);
  w.ent_src = (int32_t*)take((size_t)(M + E + 1) * 4);
  w.ent_pose = (int32_t*)take((size_t)(M + E + 1) * 4);
  w.pair_ptr = (int32_t*)take((size_t)(M + 2) * 4);
  w.Q = (float*)take((size_t)M * hw * 4);
  w.W = (float*)take((size_t)M * hw * 4);
  w.Ei = (float*)take((size_t)M * 6 * hw * 4);
  w.Eij = (float*)take((size_t)E * 6 * hw * 4);
  w.part = (float*)take((size_t)kAccSplit * M * 8 * hw * 4);
  const size_t n = (size_t)6 * P;
  w.H = (double*)take((n * n + n) * 8);   // H then b, contiguous so one memset clears both
  w.b = w.H + n * n;
  w.total = off;
  return w;
}

// ---------------------------------------------------------------------------------------
// block-wide exclusive scan of an LDS int array of length n (n <= 8*kPrepThreads)
// ---------------------------------------------------------------------------------------
__device__ void block_exclusive_scan(int* data, int n, int* scratch /*[kPrepThreads]*/, int* total) {
  const int tid = threadIdx.x;
  const int per = (n + kPrepThreads - 1) / kPrepThreads;
  const int beg = min(tid * per, n), end = min(beg + per, n);
  int s = 0;
  for (int i = beg; i < end; ++i) s += data[i];
  scratch[tid] = s;
  __syncthreads();
  for (int off = 1; off < kPrepThreads; off <<= 1) {
    int v = (tid >= off) ? scratch[tid - off] : 0;
    __syncthreads();
    scratch[tid] += v;
    __syncthreads();
  }
  int run = scratch[tid] - s;   // exclusive prefix of this thread's segment
  for (int i = beg; i < end; ++i) {
    int v = data[i];
    data[i] = run;
    run += v;
  }
  if (tid == kPrepThreads - 1) *total = scratch[tid];
  __syncthreads();
}

__global__ __launch_bounds__(kPrepThreads) void ba_prep_kernel(
    const int64_t* __restrict__ ii, const int64_t* __restrict__ jj, int E, int nbuf, int t0, int t1,
    int M_host, BaWs w) {
  __shared__ int flag[kMaxBuf];      // per frame: member of kx -> exclusive rank
  __shared__ int cnt_a[kMaxBuf];     // per depth row: out-degree -> row_ptr
  __shared__ int cnt_b[kMaxBuf];     // per depth row: #entries -> ent_ptr
  __shared__ int scratch[kPrepThreads];
  __shared__ int total;
  const int tid = threadIdx.x;

  for (int k = tid; k < nbuf; k += kPrepThreads) flag[k] = (k >= t0 && k < t1) ? 1 : 0;
  __syncthreads();
  for (int e = tid; e < E; e += kPrepThreads) {
    const int i = (int)ii[e];
    if (i >= 0 && i < nbuf) flag[i] = 1;
  }
  __syncthreads();
  // remember membership before the scan overwrites it
  for (int k = tid; k < nbuf; k += kPrepThreads) cnt_a[k] = flag[k];
  __syncthreads();
  block_exclusive_scan(flag, nbuf, scratch, &total);
  const int M_dev = total;
  const int M = min(M_dev, M_host);
  for (int k = tid; k < nbuf; k += kPrepThreads) {
    const bool mem = cnt_a[k] != 0;
    const int r = flag[k];
    const int row = (mem && r < M) ? r : -1;
    w.kinv[k] = row;
    if (row >= 0) w.kx[row] = k;
  }
  if (tid == 0) {
    w.hdr[0] = M_dev;
    w.hdr[1] = (M_dev != M_host) ? 1 : 0;
    w.hdr[2] = 0;
    w.hdr[3] = 0;
  }
  __syncthreads();   // flag[] now holds ranks; cnt_a reused below
  for (int m = tid; m < kMaxBuf; m += kPrepThreads) { cnt_a[m] = 0; cnt_b[m] = 0; }
  __syncthreads();
  // self entries: depth rows whose frame lies in the optimisation window
  for (int k = tid; k < nbuf; k += kPrepThreads) {
    const int row = w.kinv[k];
    if (row >= 0 && k >= t0 && k < t1) cnt_b[row] = 1;
  }
  __syncthreads();
  for (int e = tid; e < E; e += kPrepThreads) {
    const int i = (int)ii[e], j = (int)jj[e];
    const int row = (i >= 0 && i < nbuf) ? w.kinv[i] : -1;
    w.edge_m[e] = row;
    if (row >= 0) {
      atomicAdd(&cnt_a[row], 1);
      if (j >= t0 && j < t1) atomicAdd(&cnt_b[row], 1);
    }
  }
  __syncthreads();
  // pair counts n_m^2 (into flag[]), then the three prefix arrays.  Rows in [M, M_host) only
  // exist when the caller's n_depth disagrees with the graph (hdr[1]); they stay empty.
  for (int m = tid; m < M_host; m += kPrepThreads) flag[m] = cnt_b[m] * cnt_b[m];
  __syncthreads();
  block_exclusive_scan(cnt_a, M_host, scratch, &total);
  if (tid == 0) w.row_ptr[M_host] = total;
  block_exclusive_scan(cnt_b, M_host, scratch, &total);
  if (tid == 0) w.ent_ptr[M_host] = total;
  block_exclusive_scan(flag, M_host, scratch, &total);
  if (tid == 0) { w.pair_ptr[M_host] = total; w.hdr[4] = total; }
  __syncthreads();
  for (int m = tid; m < M_host; m += kPrepThreads) {
    w.row_ptr[m] = cnt_a[m];
    w.ent_ptr[m] = cnt_b[m];
    w.pair_ptr[m] = flag[m];
  }
  __syncthreads();
  // fill CSR + entry lists in ascending edge order (deterministic)
  for (int m = tid; m < M; m += kPrepThreads) {
    int rp = cnt_a[m], ep = cnt_b[m];
    const int k = w.kx[m];
    if (k >= t0 && k < t1) { w.ent_src[ep] = m; w.ent_pose[ep] = k - t0; ++ep; }
    for (int e = 0; e < E; ++e) {
      if (w.edge_m[e] == m) {
        w.csr_edge[rp++] = e;
        const int j = (int)jj[e];
        if (j >= t0 && j < t1) { w.ent_src[ep] = M_host + e; w.ent_pose[ep] = j - t0; ++ep; }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// Jacobians + per-edge Hessian blocks + per-pixel depth terms
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ba_accum_kernel(
    const float* __restrict__ poses, const float* __restrict__ disps, const float* __restrict__ intr,
    const float* __restrict__ disps_sens, const float* __restrict__ targets, const float* __restrict__ weights,
    const float* __restrict__ eta, const int64_t* __restrict__ jj, int t0, int t1, int hw, int wd,
    int motion_only, int reset_fail, BaWs w) {
  __shared__ float red[4][92];
  // (a call that reuses an earlier call's tables skips ba_prep_kernel, which is where the failure counters are cleared)
  if (reset_fail && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) { w.hdr[2] = 0; w.hdr[3] = 0; }
  typedef const __attribute__((address_space(4))) int* cip;
  typedef const __attribute__((address_space(4))) int64_t* clp;
  typedef const __attribute__((address_space(4))) float* cfp;
  const int m = blockIdx.y;
  if (m >= ((cip)w.hdr)[0]) return;   // only when n_depth over-states the graph (status word 1)
  // Round 6: the out-edges of keyframe m are dealt to up to kAccSplit workgroups (edge e_beg + s, e_beg + s + kAccSplit, ...):
  // the launch is ONE resident round whose length was the longest edge list (32 us at the bench window's <= 6 out-edges,
  // 86 us in the live frontend, where the window's inactive edges make lists of ~20).  The per-EDGE sums go to the fp64
  // system as before; the per-PIXEL sums over a keyframe's edges (C, w, Ei) of a split list leave as partials and are
  // added in split order by ba_accum_finish_kernel (deterministic: no atomics); an unsplit list finishes here.
  // The launcher splits only graphs whose lists are long on average (edges >= 6 x depth keyframes: the live window with its
  // inactive edges 62 + 6 us instead of 86, the 200-keyframe graph); at the bench window's 3 edges per keyframe the split
  // kernel + the finishing launch (24.5 + 5.6 us) only equal the unsplit kernel (32 us), so gridDim.z is 1 there.
  const int split = blockIdx.z, nsp = gridDim.z;
  const int nsplit = min(nsp, max(((cip)w.row_ptr)[m + 1] - ((cip)w.row_ptr)[m], 1));
  if (split >= nsplit) return;
  const int tid = threadIdx.x;
  const int p = blockIdx.x * 256 + tid;
  const bool active = p < hw;
  const int pc = active ? p : hw - 1;
  const int k = ((cip)w.kx)[m];
  const int P = t1 - t0;
  const int n6 = 6 * P;
  const float fx = ((cfp)intr)[0], fy = ((cfp)intr)[1], cx = ((cfp)intr)[2], cy = ((cfp)intr)[3];
  const float u = (float)(pc % wd), v = (float)(pc / wd);
  const float disp = disps[(size_t)k * hw + pc];
  float Xi[4] = {(u - cx) / fx, (v - cy) / fy, 1.0f, disp};
  float ti[3], qi[4];
  {
    cfp pp = (cfp)(poses + (size_t)k * 7);
    ti[0] = pp[0]; ti[1] = pp[1]; ti[2] = pp[2];
    qi[0] = pp[3]; qi[1] = pp[4]; qi[2] = pp[5]; qi[3] = pp[6];
  }
  const int pi = k - t0;
  const bool i_in = (pi >= 0) && (pi < P);

  float Csum = 0.f, wsum = 0.f;
  float Ei[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  // The edge list, the edges' targets and the target keyframes' poses are uniform over the workgroup and unchanged
  // during the launch: read through the constant address space they are SCALAR loads (tens of ns from the scalar cache).
  // As vector loads -- what a plain pointer gives once the kernel has stored anything -- list entry -> edge -> jj -> pose
  // was a chain of three dependent ~1 us round trips in front of every edge's arithmetic, ~10 us of this 33 us kernel.
  const int e_beg = ((cip)w.row_ptr)[m], e_end = ((cip)w.row_ptr)[m + 1];
  for (int idx = e_beg + split; idx < e_end; idx += nsp) {
    const int e = ((cip)w.csr_edge)[idx];
    const int jx = (int)((clp)jj)[e];
    // (the edge's four per-pixel inputs: requested here, in front of the pose arithmetic, and unconditionally -- pc is in
    // range; behind `close || !active` each was its own load -> wait)
    const size_t tb = ((size_t)e * 2) * hw + pc;
    const float w_u = weights[tb], w_v = weights[tb + hw], t_u = targets[tb], t_v = targets[tb + hw];
    float tij[3], qij[4];
    if (jx == k) {   // stereo frames (droid_kernels.cu:219-229)
      tij[0] = -0.1f; tij[1] = 0.f; tij[2] = 0.f;
      qij[0] = 0.f; qij[1] = 0.f; qij[2] = 0.f; qij[3] = 1.f;
    } else {
      cfp pj = (cfp)(poses + (size_t)jx * 7);
      float tj[3] = {pj[0], pj[1], pj[2]};
      float qj[4] = {pj[3], pj[4], pj[5], pj[6]};
      gs_rel_se3(ti, qi, tj, qj, tij, qij);
    }
    float Xj[4];
    gs_act_se3(tij, qij, Xi, Xj);
    const float x = Xj[0], y = Xj[1], h = Xj[3];
    const bool close = Xj[2] < kMinDepth;
    const float d = close ? 0.0f : 1.0f / Xj[2];
    const float d2 = d * d;
    float wu = (close || !active) ? 0.0f : 0.001f * w_u;
    float wv = (close || !active) ? 0.0f : 0.001f * w_v;
    const float ru = t_u - (fx * d * x + cx);
    const float rv = t_v - (fy * d * y + cy);

    float hij[78];
#pragma unroll
    for (int l = 0; l < 78; ++l) hij[l] = 0.f;
    float vi[6], vj[6], Eii_e[6], Eij_e[6];
    float Jx[12];
    float* Ji = &Jx[0];
    float* Jj = &Jx[6];

    // ---- x-coordinate row (droid_kernels.cu:312-342)
    Jj[0] = fx * (h * d);
    Jj[1] = fx * 0.0f;
    Jj[2] = fx * (-x * h * d2);
    Jj[3] = fx * (-x * y * d2);
    Jj[4] = fx * (1.0f + x * x * d2);
    Jj[5] = fx * (-y * d);
    float Jz = fx * (tij[0] * d - tij[2] * (x * d2));
    float Ce = wu * Jz * Jz;
    float be = wu * ru * Jz;
    if (jx == k) wu = 0.0f;
    gs_adj_se3(tij, qij, Jj, Ji);
#pragma unroll
    for (int n = 0; n < 6; ++n) Ji[n] = -Ji[n];
    {
      int l = 0;
#pragma unroll
      for (int n = 0; n < 12; ++n)
#pragma unroll
        for (int mm = 0; mm <= n; ++mm) { hij[l] = fmaf(wu * Jx[n], Jx[mm], hij[l]); ++l; }
    }
#pragma unroll
    for (int n = 0; n < 6; ++n) {
      vi[n] = wu * ru * Ji[n];
      vj[n] = wu * ru * Jj[n];
      Eii_e[n] = wu * Jz * Ji[n];
      Eij_e[n] = wu * Jz * Jj[n];
    }
    // ---- y-coordinate row (:345-375)
    Jj[0] = fy * 0.0f;
    Jj[1] = fy * (h * d);
    Jj[2] = fy * (-y * h * d2);
    Jj[3] = fy * (-1.0f - y * y * d2);
    Jj[4] = fy * (x * y * d2);
    Jj[5] = fy * (x * d);
    Jz = fy * (tij[1] * d - tij[2] * (y * d2));
    Ce = Ce + wv * Jz * Jz;
    be = be + wv * rv * Jz;
    if (jx == k) wv = 0.0f;
    gs_adj_se3(tij, qij, Jj, Ji);
#pragma unroll
    for (int n = 0; n < 6; ++n) Ji[n] = -Ji[n];
    {
      int l = 0;
#pragma unroll
      for (int n = 0; n < 12; ++n)
#pragma unroll
        for (int mm = 0; mm <= n; ++mm) { hij[l] = fmaf(wv * Jx[n], Jx[mm], hij[l]); ++l; }
    }
#pragma unroll
    for (int n = 0; n < 6; ++n) {
      vi[n] = vi[n] + wv * rv * Ji[n];
      vj[n] = vj[n] + wv * rv * Jj[n];
      Eii_e[n] = Eii_e[n] + wv * Jz * Ji[n];
      Eij_e[n] = Eij_e[n] + wv * Jz * Jj[n];
    }

    if (!motion_only) {
      Csum += Ce;
      wsum += be;
#pragma unroll
      for (int n = 0; n < 6; ++n) Ei[n] += Eii_e[n];
      if (active) {
        float* eo = w.Eij + ((size_t)e * 6) * hw + p;
#pragma unroll
        for (int n = 0; n < 6; ++n) eo[(size_t)n * hw] = Eij_e[n];
      }
    }

    // ---- reduce the edge's 78 + 12 sums over the workgroup, add into the fp64 system
    const int pj_ = jx - t0;
    const bool j_in = (pj_ >= 0) && (pj_ < P);
    if (i_in || j_in) {   // uniform across the workgroup
      const int wave = tid >> 6, lane = tid & 63;
      // 78 + 6 + 6 sums per wave: one reduce-scatter (~90 lane exchanges) instead of 90 butterflies (~1000 instructions)
      float rs[90];
#pragma unroll
      for (int l = 0; l < 78; ++l) rs[l] = hij[l];
#pragma unroll
      for (int n = 0; n < 6; ++n) { rs[78 + n] = vi[n]; rs[84 + n] = vj[n]; }
      gs_wave_reduce_scatter<90>(rs, lane);
      const int ix = gs_bitrev6(lane);
      red[wave][ix] = rs[0];
      if (64 + ix < 90) red[wave][64 + ix] = rs[1];
      __syncthreads();
      if (tid < 90) {
        const double s = (double)((red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]));
        if (tid < 78) {
          // decode packed lower index l -> (n, mm), n >= mm, over the stacked [Ji, Jj]
          int n = 0;
          while ((n + 1) * (n + 2) / 2 <= tid) ++n;
          const int mm = tid - n * (n + 1) / 2;
          if (n < 6) {                       // Hii, lower part
            if (i_in) gs_atomic_add_f64(&w.H[(size_t)(6 * pi + n) * n6 + (6 * pi + mm)], s);
          } else if (mm >= 6) {              // Hjj, lower part
            if (j_in) gs_atomic_add_f64(&w.H[(size_t)(6 * pj_ + n - 6) * n6 + (6 * pj_ + mm - 6)], s);
          } else if (i_in && j_in && pi != pj_) {   // Hji[n-6][mm] == Hij[mm][n-6]
            if (pj_ > pi) gs_atomic_add_f64(&w.H[(size_t)(6 * pj_ + n - 6) * n6 + (6 * pi + mm)], s);
            else          gs_atomic_add_f64(&w.H[(size_t)(6 * pi + mm) * n6 + (6 * pj_ + n - 6)], s);
          }
        } else if (tid < 84) {
          if (i_in) gs_atomic_add_f64(&w.b[6 * pi + (tid - 78)], s);
        } else {
          if (j_in) gs_atomic_add_f64(&w.b[6 * pj_ + (tid - 84)], s);
        }
      }
      __syncthreads();
    }
  }

  if (!motion_only && active) {
    if (nsplit > 1) {          // this workgroup's share of the keyframe's per-pixel sums
      float* po = w.part + (((size_t)split * gridDim.y + m) * 8) * hw + p;     // (gridDim.y = the M the workspace was carved for)
      po[0] = Csum;
      po[(size_t)hw] = wsum;
#pragma unroll
      for (int n = 0; n < 6; ++n) po[(size_t)(2 + n) * hw] = Ei[n];
      return;
    }
    const size_t o = (size_t)m * hw + p;
    const float sens = disps_sens[(size_t)k * hw + p];
    const float mk = (sens > 0.0f) ? 1.0f : 0.0f;
    const float C = Csum + mk * kAlpha + (1.0f - mk) * eta[o];
    const float ww = wsum - mk * kAlpha * (disp - sens);
    w.Q[o] = 1.0f / C;
    w.W[o] = ww;
    if (i_in) {
      float* eo = w.Ei + ((size_t)m * 6) * hw + p;
#pragma unroll
      for (int n = 0; n < 6; ++n) eo[(size_t)n * hw] = Ei[n];
    }
  }
}

// the per-pixel sums of the keyframes whose edge list ba_accum_kernel split: partials added in split order, then the same
// depth prior / damping / Q = 1 / C as the unsplit path
__global__ __launch_bounds__(256) void ba_accum_finish_kernel(const float* __restrict__ disps,
                                                              const float* __restrict__ disps_sens,
                                                              const float* __restrict__ eta, int t0, int t1, int hw, int nsp,
                                                              BaWs w) {
  const int m = blockIdx.y;
  if (m >= w.hdr[0]) return;
  const int nsplit = min(nsp, max(w.row_ptr[m + 1] - w.row_ptr[m], 1));
  if (nsplit <= 1) return;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= hw) return;
  float acc[8];
#pragma unroll
  for (int n = 0; n < 8; ++n) acc[n] = 0.0f;
  for (int s2 = 0; s2 < nsplit; ++s2) {
    const float* pi_ = w.part + (((size_t)s2 * gridDim.y + m) * 8) * hw + p;
#pragma unroll
    for (int n = 0; n < 8; ++n) acc[n] += pi_[(size_t)n * hw];
  }
  const int k = w.kx[m];
  const size_t o = (size_t)m * hw + p;
  const float disp = disps[(size_t)k * hw + p];
  const float sens = disps_sens[(size_t)k * hw + p];
  const float mk = (sens > 0.0f) ? 1.0f : 0.0f;
  const float C = acc[0] + mk * kAlpha + (1.0f - mk) * eta[o];
  const float ww = acc[1] - mk * kAlpha * (disp - sens);
  w.Q[o] = 1.0f / C;
  w.W[o] = ww;
  const int pi = k - t0;
  if (pi >= 0 && pi < t1 - t0) {
    float* eo = w.Ei + ((size_t)m * 6) * hw + p;
#pragma unroll
    for (int n = 0; n < 6; ++n) eo[(size_t)n * hw] = acc[2 + n];
  }
}

// ---------------------------------------------------------------------------------------
// Schur complement: for every depth row m and ordered entry pair (a,b) of m with
// pose_a >= pose_b:  H[pa,pb] -= sum_pix (E_a Q) E_b^T ;  a==b also does b[pa] -= E_a Q w.
// Persistent grid over the flat pair index (its length only exists on the device).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ba_schur_kernel(int M, int hw, int P, BaWs w) {
  __shared__ float red[4][44];
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  // (the pair tables are uniform over the workgroup and unchanged during the launch: through the constant address space
  // the binary search and the entity look-ups are scalar loads -- as vector loads they were ~9 dependent round trips in
  // front of every pair's pixel loop)
  typedef const __attribute__((address_space(4))) int* cip;
  cip pair_ptr = (cip)w.pair_ptr, ent_ptr = (cip)w.ent_ptr, ent_pose = (cip)w.ent_pose, ent_src = (cip)w.ent_src;
  const int n_pairs = pair_ptr[M];
  const int n6 = 6 * P;
  for (int pidx = blockIdx.x; pidx < n_pairs; pidx += gridDim.x) {
    // binary search: largest m with pair_ptr[m] <= pidx
    int lo = 0, hi = M;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (pair_ptr[mid] <= pidx) lo = mid; else hi = mid;
    }
    const int m = lo;
    const int eb = ent_ptr[m];
    const int nm = ent_ptr[m + 1] - eb;
    const int r = pidx - pair_ptr[m];
    const int a = r / nm, b = r - a * nm;
    const int pa = ent_pose[eb + a], pb = ent_pose[eb + b];
    if (pa < pb) continue;   // covered by the transposed pair
    const int sa = ent_src[eb + a], sb = ent_src[eb + b];
    const float* Ea = (sa < M) ? w.Ei + (size_t)sa * 6 * hw : w.Eij + (size_t)(sa - M) * 6 * hw;
    const float* Eb = (sb < M) ? w.Ei + (size_t)sb * 6 * hw : w.Eij + (size_t)(sb - M) * 6 * hw;
    const float* Qm = w.Q + (size_t)m * hw;
    const float* Wm = w.W + (size_t)m * hw;
    const bool diag = (a == b);
    float S[36];
#pragma unroll
    for (int i = 0; i < 36; ++i) S[i] = 0.f;
    float vv[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // Four pixels per trip, all 4 x 14 loads requested before the first product (round 6): a pair's 4800 pixels were 19
    // dependent round trips of one pixel per lane (19.7 us per launch at the bench window, 44 us in the live frontend:
    // one resident round of workgroups, each as long as ITS pair's latency chain).  Same products in the same order.
    constexpr int UP = 4;
    for (int p0 = tid; p0 < hw; p0 += UP * 256) {
      float q[UP], ea[UP][6], eb_[UP][6], ww[UP];
#pragma unroll
      for (int u = 0; u < UP; ++u) {
        const int p = p0 + 256 * u;
        const bool ok = p < hw;
        const int pc = ok ? p : tid;
        q[u] = Qm[pc];
#pragma unroll
        for (int n = 0; n < 6; ++n) {
          ea[u][n] = Ea[(size_t)n * hw + pc];
          eb_[u][n] = Eb[(size_t)n * hw + pc];
        }
        ww[u] = diag ? Wm[pc] : 0.0f;
      }
#pragma unroll
      for (int u = 0; u < UP; ++u) {
        if (p0 + 256 * u < hw) {
#pragma unroll
          for (int n = 0; n < 6; ++n) ea[u][n] = ea[u][n] * q[u];
#pragma unroll
          for (int n = 0; n < 6; ++n)
#pragma unroll
            for (int c = 0; c < 6; ++c) S[n * 6 + c] = fmaf(ea[u][n], eb_[u][c], S[n * 6 + c]);
          if (diag) {
#pragma unroll
            for (int n = 0; n < 6; ++n) vv[n] = fmaf(ea[u][n], ww[u], vv[n]);
          }
        }
      }
    }
    {
      float rs[42];                                     // 36 + 6 sums: one reduce-scatter per wave
#pragma unroll
      for (int i = 0; i < 36; ++i) rs[i] = S[i];
#pragma unroll
      for (int n = 0; n < 6; ++n) rs[36 + n] = vv[n];
      gs_wave_reduce_scatter<42>(rs, lane);
      const int ix = gs_bitrev6(lane);
      if (ix < 42) red[wave][ix] = rs[0];
    }
    __syncthreads();
    if (tid < 36) {
      const int rr = tid / 6, cc = tid - rr * 6;
      const double s = (double)((red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]));
      if (pa > pb || rr >= cc)   // strictly-lower block: all 36; diagonal block: lower part
        gs_atomic_add_f64(&w.H[(size_t)(6 * pa + rr) * n6 + (6 * pb + cc)], -s);
    } else if (diag && tid < 42) {
      const double s = (double)((red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]));
      gs_atomic_add_f64(&w.b[6 * pa + (tid - 36)], -s);
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------
// Back-substitution + retractions.  Rows m < M handle a 256-pixel chunk of depth row m;
// the extra row blockIdx.y == M retracts the poses (motion_only launches only that row).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ba_update_kernel(
    float* __restrict__ poses, float* __restrict__ disps, const float* __restrict__ dx,
    float* __restrict__ dz, int M, int t0, int t1, int hw, BaWs w) {
  const int tid = threadIdx.x;
  const int P = t1 - t0;
  if ((int)blockIdx.y == M) {
    if (blockIdx.x != 0) return;
    for (int kk = t0 + tid; kk < t1; kk += 256) {   // droid_kernels.cu:898-931
      float* pp = poses + (size_t)kk * 7;
      float t[3] = {pp[0], pp[1], pp[2]};
      float q[4] = {pp[3], pp[4], pp[5], pp[6]};
      float xi[6], tn[3], qn[4];
#pragma unroll
      for (int n = 0; n < 6; ++n) xi[n] = dx[(size_t)(kk - t0) * 6 + n];
      gs_retr_se3(xi, t, q, tn, qn);
      pp[0] = tn[0]; pp[1] = tn[1]; pp[2] = tn[2];
      pp[3] = qn[0]; pp[4] = qn[1]; pp[5] = qn[2]; pp[6] = qn[3];
    }
    return;
  }
  const int m = blockIdx.y;
  const int p = blockIdx.x * 256 + tid;
  if (p >= hw) return;
  if (m >= w.hdr[0]) {                     // eta had more rows than the graph has depth keyframes (status word 1):
    dz[(size_t)m * hw + p] = 0.0f;         // the surplus rows of dz are defined (zero), not left as allocated
    return;
  }
  const int eb = w.ent_ptr[m], ee = w.ent_ptr[m + 1];
  float dw = 0.f;
  for (int idx = eb; idx < ee; ++idx) {
    const int pn = w.ent_pose[idx];
    if (pn <= 0 || pn >= P) continue;   // droid_kernels.cu:1105 (first window pose is skipped)
    const int s = w.ent_src[idx];
    const float* En = (s < M) ? w.Ei + (size_t)s * 6 * hw : w.Eij + (size_t)(s - M) * 6 * hw;
    float acc = 0.f;
#pragma unroll
    for (int n = 0; n < 6; ++n) acc += En[(size_t)n * hw + p] * dx[(size_t)pn * 6 + n];
    dw += acc;
  }
  const size_t o = (size_t)m * hw + p;
  const float z = w.Q[o] * (w.W[o] - dw);
  dz[o] = z;
  const size_t dk = (size_t)w.kx[m] * hw + p;
  disps[dk] = disps[dk] + z;            // droid_kernels.cu:933-946
}

__global__ void ba_status_kernel(int32_t* status_out, BaWs w) {
  if (threadIdx.x < 4) status_out[threadIdx.x] = (threadIdx.x == 2) ? w.hdr[3] : (threadIdx.x == 3 ? w.hdr[4] : w.hdr[threadIdx.x]);
}

}  // namespace

extern "C" size_t gs_ba_workspace_bytes(int n_edges, int n_poses, int n_depth, int nbuf, int hw) {
  if (n_edges < 0 || n_poses < 0 || n_depth < 0 || nbuf < 0 || hw < 0) return 0;
  return carve(nullptr, n_edges, n_poses, n_depth, nbuf, hw).total + 256;
}

extern "C" int gs_ba_ex(float* poses, float* disps, const float* intrinsics, const float* disps_sens,
                        const float* targets, const float* weights, const float* eta, const int64_t* ii,
                        const int64_t* jj, int t0, int t1, int iterations, float lm, float ep, int motion_only,
                        int n_edges, int n_depth, int nbuf, int h, int w, float* dx, float* dz,
                        int32_t* status_out, void* workspace, size_t workspace_bytes, int flags, gs_stream_t stream) {
  GS_REQUIRE(poses && disps && intrinsics && disps_sens && targets && weights && ii && jj && dx && workspace,
             "ba: null pointer");
  GS_REQUIRE(motion_only || (eta && dz), "ba: eta/dz required unless motion_only");
  GS_REQUIRE(h > 0 && w > 0 && n_edges >= 0 && iterations >= 0, "ba: bad shape");
  GS_REQUIRE(t0 >= 0 && t1 > t0 && t1 <= nbuf, "ba: window [%d,%d) outside buffer of %d frames", t0, t1, nbuf);
  GS_REQUIRE(nbuf <= kMaxBuf, "ba: buffer of %d frames exceeds the supported %d", nbuf, kMaxBuf);
  GS_REQUIRE(n_depth > 0 && n_depth <= nbuf && n_depth <= 65534, "ba: n_depth=%d out of range", n_depth);
  const int hw = h * w, P = t1 - t0, M = n_depth;
  const size_t need = gs_ba_workspace_bytes(n_edges, P, M, nbuf, hw);
  if (workspace_bytes < need) {
    gs_set_error("ba: workspace too small (%zu < %zu bytes)", workspace_bytes, need);
    return GS_ERR_WORKSPACE;
  }
  void* base = (void*)gs_align((size_t)workspace);
  BaWs ws = carve(base, n_edges, P, M, nbuf, hw);
  hipStream_t st = (hipStream_t)stream;
  const int n6 = 6 * P;

  // GS_BA_REUSE_TABLES: the index tables in `workspace` are those of an earlier call with the same ii / jj / t0 / t1 /
  // n_depth / nbuf / map size (the caller's promise): they depend on nothing else, so ba_prep_kernel is skipped
  const bool reuse = (flags & 1) != 0;
  if (!reuse) {
    ba_prep_kernel<<<1, kPrepThreads, 0, st>>>(ii, jj, n_edges, nbuf, t0, t1, M, ws);
    GS_CHECK_LAUNCH("ba_prep");
  }
  const int chunks = gs_cdiv(hw, 256);
  for (int it = 0; it < iterations; ++it) {
    if (hipMemsetAsync(ws.H, 0, ((size_t)n6 * n6 + n6) * sizeof(double), st) != hipSuccess) {
      gs_set_error("ba: memset failed");
      return GS_ERR_LAUNCH;
    }
    const int nsp = (!motion_only && n_edges >= 6 * M) ? kAccSplit : 1;      // (see ba_accum_kernel)
    ba_accum_kernel<<<dim3(chunks, M, nsp), 256, 0, st>>>(poses, disps, intrinsics, disps_sens, targets, weights,
                                                          eta, jj, t0, t1, hw, w, motion_only,
                                                          (reuse && it == 0) ? 1 : 0, ws);
    GS_CHECK_LAUNCH("ba_accum");
    if (!motion_only) {
      if (nsp > 1) {
        ba_accum_finish_kernel<<<dim3(chunks, M), 256, 0, st>>>(disps, disps_sens, eta, t0, t1, hw, nsp, ws);
        GS_CHECK_LAUNCH("ba_accum_finish");
      }
      ba_schur_kernel<<<2048, 256, 0, st>>>(M, hw, P, ws);
      GS_CHECK_LAUNCH("ba_schur");
    }
    int rc = gs_chol_solve_launch(ws.H, ws.b, n6, lm, ep, dx, &ws.hdr[2], &ws.hdr[3], &ws.hdr[8], st);
    if (rc != GS_OK) return rc;
    if (motion_only)
      ba_update_kernel<<<dim3(1, 1), 256, 0, st>>>(poses, disps, dx, dz, 0, t0, t1, hw, ws);
    else
      ba_update_kernel<<<dim3(chunks, M + 1), 256, 0, st>>>(poses, disps, dx, dz, M, t0, t1, hw, ws);
    GS_CHECK_LAUNCH("ba_update");
  }
  if (status_out) {
    ba_status_kernel<<<1, 64, 0, st>>>(status_out, ws);
    GS_CHECK_LAUNCH("ba_status");
  }
  return GS_OK;
}

extern "C" int gs_ba(float* poses, float* disps, const float* intrinsics, const float* disps_sens,
                     const float* targets, const float* weights, const float* eta, const int64_t* ii,
                     const int64_t* jj, int t0, int t1, int iterations, float lm, float ep, int motion_only,
                     int n_edges, int n_depth, int nbuf, int h, int w, float* dx, float* dz,
                     int32_t* status_out, void* workspace, size_t workspace_bytes, gs_stream_t stream) {
  return gs_ba_ex(poses, disps, intrinsics, disps_sens, targets, weights, eta, ii, jj, t0, t1, iterations, lm, ep,
                  motion_only, n_edges, n_depth, nbuf, h, w, dx, dz, status_out, workspace, workspace_bytes, 0, stream);
}
