// Backward of the NeuS renderer for the mapper's training step (reference: the autograd graph of
// src/InstantNeuS.py:295-370 incl. autograd.grad(create_graph=True) at :141-148 and tiny-cuda-nn's
// grid backward / double-backward kernels).
//
//   neus_ray_bwd_kernel    one wave per ray: recompute transmittance with a prefix scan, form
//                          dL/dw_k, and turn it into dL/dalpha_k with a reverse suffix scan
//                          (T_j, j>k, all depend on alpha_k); emits d_rgb and the normal-output
//                          share of d_grad.
//   neus_point_bwd_kernel  one lane per sample point: NeuS-alpha chain, SDF linear layer, and the
//                          hash grid -- both the value path (d enc * w_corner) and the second-order
//                          path through d sdf / d x (d dy_dx * +-scale * w_other) are scattered with
//                          fp32 atomics into the table gradient; small dense parameter gradients
//                          are emitted as per-point rows and reduced by GEMM / column sums on the
//                          host side of the C ABI's caller.
//
// Table gradient WITHOUT global atomics (gs_neus_backward_points_binned; the production mode).  Every flavour of
// global atomic on MI355X retires at 21 G requests/s however the addresses are spread (scratch/atomic_bench.hip,
// profiles/r03_pmc_neus.json: pk_add_f16 / add_f32, sc1 or not, 0.5 ... 25 MB tables, XCD-private regions), and the PMC
// counters show all of them forwarded to the memory side; the 95 M requests of a 32768-ray step cost 4.5 ms of this
// kernel's 5.0.  The same read-modify-writes run 7x faster as plain L2 traffic and far faster still in LDS, so for the
// 11 HASHED levels (93 % of the requests; 2^19 entries each) the scatter becomes bin-and-reduce:
//   pass 1  neus_point_bwd_kernel<true>: each hashed level is cut into 64 bins of 8192 entries (bin = index >> 13).  A
//           workgroup stages its (13-bit index, 2 x fp16 value) records per bin in LDS (double-buffered, so one barrier
//           per level) and copies every staged bin out as ONE contiguous run into a segment of the bin's queue that
//           belongs to this workgroup alone -- no reservation, no global atomic, no zeroing: the segment's fill count
//           (one byte) is written with it.
//   pass 2  grid_bin_reduce_kernel: one workgroup per bin sums the bin's records in LDS and writes the bin's 8192 entries
//           of the table gradient ONCE with plain stores (adding what is already there: any overflow records' atomics).
//           The sums are INTEGER: an fp16 value is a multiple of 2^-24, so every record is converted exactly to 64-bit
//           fixed point (x 2^24) and added with ds_add_u64 -- the bin's sum is exact and independent of the order, and
//           is rounded to fp16 once.  Why not float: scratch/lds_atomic_bench.hip measures ds_add_f32 / ds_pk_add_f16
//           at 0.33 lanes per clock and CU against 5.3 for integer LDS atomics (the first version of this pass, with
//           ds_add_f32: 2.2 ms; a barrier-synchronised non-atomic version: 5.3 ms).
// The dense levels 0-4 (0.5 M entries, heavily pre-reduced inside the wave) stay on packed atomics.  Overflow of a
// staging bin or of a queue falls back to the atomic, so every input is handled.  Side effect: the hashed levels'
// gradient is summed in fp32 (each record rounded once to fp16) instead of through thousands of fp16 read-modify-writes.
#include "common.h"
#include "neus_common.h"
#include <math.h>

namespace {

__device__ __forceinline__ float wave_incl_prod(float v, int lane) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const float u = __shfl_up(v, off, 64);
    if (lane >= off) v = v * u;
  }
  return v;
}

// inclusive suffix sum over lanes (lane k gets sum_{j>=k} v_j)
__device__ __forceinline__ float wave_incl_suffix_sum(float v, int lane) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const float u = __shfl_down(v, off, 64);
    if (lane + off < 64) v = v + u;
  }
  return v;
}

__global__ __launch_bounds__(256) void neus_ray_bwd_kernel(
    const float* __restrict__ alpha, const _Float16* __restrict__ rgb, const float* __restrict__ zmid,
    const float* __restrict__ grad, const uint8_t* __restrict__ mask, const float* __restrict__ d_color,
    const float* __restrict__ d_depth, const float* __restrict__ d_dvar, const float* __restrict__ d_normal,
    const float* __restrict__ d_wsum, float* __restrict__ d_alpha, float* __restrict__ d_rgb,
    float* __restrict__ d_grad, int n, int s) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n) return;
  const size_t b = (size_t)r * s;
  // two samples per lane: k0 = lane, k1 = lane + 64   (s <= 128)
  float a[2], z[2], T[2], w[2], mk[2], g[2][3], c[2][3];
  bool on[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int k = lane + 64 * h;
    on[h] = k < s;
    a[h] = on[h] ? alpha[b + k] : 0.0f;
    z[h] = on[h] ? zmid[b + k] : 0.0f;
    mk[h] = (on[h] && mask[b + k]) ? 1.0f : 0.0f;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      g[h][d] = on[h] ? grad[(b + k) * 3 + d] : 0.0f;
      c[h][d] = on[h] ? (float)rgb[(b + k) * 3 + d] : 0.0f;
    }
  }
  float Trun = 1.0f;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const float t = on[h] ? (1.0f - a[h] + 1e-7f) : 1.0f;
    const float incl = wave_incl_prod(t, lane);
    float excl = __shfl_up(incl, 1, 64);
    if (lane == 0) excl = 1.0f;
    T[h] = Trun * excl;
    w[h] = a[h] * T[h];
    Trun = Trun * __shfl(incl, 63, 64);
  }
  const float wsum = gs_wave_sum(w[0] + w[1]);
  const float dep = gs_wave_sum(z[0] * w[0] + z[1] * w[1]);
  const float dc[3] = {d_color[r * 3 + 0], d_color[r * 3 + 1], d_color[r * 3 + 2]};
  const float dn[3] = {d_normal[r * 3 + 0], d_normal[r * 3 + 1], d_normal[r * 3 + 2]};
  const float ddv = d_dvar[r], dws = d_wsum[r];
  // depth also enters depth_var: d depth_var / d depth = -2 sum_k w_k (z_k - depth)
  const float dd_tot = d_depth[r] - 2.0f * ddv * (dep - dep * wsum);
  float dLdw[2], u[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const float dz = z[h] - dep;
    float v = (dc[0] * c[h][0] + dc[1] * c[h][1] + dc[2] * c[h][2]) + dd_tot * z[h] + dws + ddv * dz * dz;
    v += (dn[0] * g[h][0] + dn[1] * g[h][1] + dn[2] * g[h][2]) * mk[h];
    dLdw[h] = on[h] ? v : 0.0f;
    u[h] = dLdw[h] * w[h];
  }
  // R_k = sum_{j>k} u_j  (suffix over the 128 virtual positions, second half first)
  float R[2];
  {
    const float inc1 = wave_incl_suffix_sum(u[1], lane);
    R[1] = inc1 - u[1];
    const float tot1 = __shfl(inc1, 0, 64);
    const float inc0 = wave_incl_suffix_sum(u[0], lane);
    R[0] = (inc0 - u[0]) + tot1;
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int k = lane + 64 * h;
    if (!on[h]) continue;
    const float da = dLdw[h] * T[h] - R[h] / (1.0f - a[h] + 1e-7f);
    d_alpha[b + k] = da * mk[h];                       // stored alpha = alpha * mask
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      d_rgb[(b + k) * 3 + d] = dc[d] * w[h] * mk[h];   // masked-out points have rgb == 0 (no grad)
      d_grad[(b + k) * 3 + d] = dn[d] * w[h] * mk[h];
    }
  }
}

// ---------------------------------------------------------------------------------------
struct BwdArgs {
  const float* rays_o; const float* rays_d; const float* z_vals; const float* dists;
  const _Float16* grid; const float* sdf_w; const float* sdf_wt; const float* color_B;
  float inv_s; const float* inv_s_dev; float bound[6];
  const float* sdf; const float* grad; const uint8_t* mask;
  const float* d_alpha; const float* d_sdf; const float* d_grad; const void* dX;
  const float* d_gerr_ray;
  const _Float16* enc_aux;     // [16][n*s][8] f16 records of the forward (encoding, d enc / d x), or nullptr: gather again
  float* grid_grad; _Float16* grid_grad16; float grad_scale16; void* d_out; void* lin_in; void* dw0; void* d_arg; void* pts; float* d_inv_s;
  int rows16; float row_scale; int dx16; float dx_inv_scale; int row_stride16;
  int n, s;
  // binned table gradient (BINNED instantiation): record queues [hashed level * 64 + bin][workgroup][ST_SLOTS] and the
  // segments' fill counts [hashed level * 64 + bin][workgroup]
  uint16_t* q_idx; uint32_t* q_val; uint8_t* q_cnt;
#ifdef GS_BWD_STAMP
  unsigned long long* stamp;   // [16] phase times summed over the workgroups' thread 0 (100 MHz ticks); -DGS_BWD_STAMP builds only
#endif
};
#ifdef GS_BWD_STAMP
#define BSTAMP(i) do { const long long t_now = wall_clock64(); t_acc[i] += t_now - t_last; t_last = t_now; } while (0)
#else
#define BSTAMP(i) do { } while (0)
#endif

constexpr int BIN_SHIFT = 13, BIN_ENTRIES = 1 << BIN_SHIFT;   // 8192 entries per bin
constexpr int BINS_PER_LEVEL = 64;                            // hashed levels hold 2^19 entries
constexpr int ST_SLOTS = 48;                                  // records per (workgroup, level, bin): staging AND queue segment
                                                              // (expected 256 x 8 / 64 = 32 at most; beyond: atomics)

// 40 consecutive entries of row i of dX: ALL loads are issued before the first conversion.  (One chunk at a time --
// load, convert, next chunk, each behind the dtype branch -- every conversion waits for its own load with nothing else in
// flight: five serialized memory round trips at the head of every wave and five more at its tail, of a wave that lives
// ~22 us; found in the ISA after the same pattern turned up in the MLP backward's prefetch.)
__device__ __forceinline__ void load_dx40(const BwdArgs& A, int i, int c, float* out) {
  if (A.dx16) {
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    const h8* r = reinterpret_cast<const h8*>(reinterpret_cast<const _Float16*>(A.dX) + (size_t)i * 80 + c);
    h8 v[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) v[k] = r[k];
#pragma unroll
    for (int k = 0; k < 5; ++k)
#pragma unroll
      for (int e = 0; e < 8; ++e) out[8 * k + e] = (float)v[k][e] * A.dx_inv_scale;
  } else {
    const float4* r = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(A.dX) + (size_t)i * 80 + c);
    float4 v[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) v[k] = r[k];
#pragma unroll
    for (int k = 0; k < 10; ++k) { out[4 * k] = v[k].x; out[4 * k + 1] = v[k].y; out[4 * k + 2] = v[k].z; out[4 * k + 3] = v[k].w; }
  }
}

// fp16 per-point rows leave through a wave-private LDS tile: a lane-per-point store of 2-byte values puts 64
// unrelated addresses into every store instruction (138 of them per point), which the write path handles
// at a fraction of HBM speed; the tile turns them into contiguous 16-byte pieces of the wave's 64 rows.
constexpr int ROW_TS = 41;                       // tile row stride in dwords (lin 20 | w0 20 | pad)
__device__ __forceinline__ void wave_sync_lds() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// store `ndw` dwords per point (a multiple of 4) from tile[point][off .. off+ndw) to rows of `ndw` dwords
// `stride16` = distance between consecutive points' rows in 16-byte units (0: rows are contiguous)
__device__ __forceinline__ void tile_flush(const uint32_t* __restrict__ tile, int off, int ndw, void* __restrict__ base,
                                           size_t p0, size_t np, int lane, int stride16) {
  const int ppp = ndw >> 2;                       // 16-byte pieces per point
  const int st = stride16 ? stride16 : ppp;
  uint4* dst = reinterpret_cast<uint4*>(base) + p0 * st;
  for (int idx = lane; idx < 64 * ppp; idx += 64) {
    const int pt = idx / ppp, part = idx - pt * ppp;
    if (p0 + pt < np) {
      const uint32_t* t = tile + pt * ROW_TS + off + 4 * part;
      dst[(size_t)pt * st + part] = make_uint4(t[0], t[1], t[2], t[3]);
    }
  }
}
__device__ __forceinline__ uint32_t pack2h(float a, float b) {
  return (uint32_t)__builtin_bit_cast(unsigned short, (_Float16)a) |
         ((uint32_t)__builtin_bit_cast(unsigned short, (_Float16)b) << 16);
}

// per-point rows (reduced by the caller's split-K GEMMs) in fp32 or fp16
__device__ __forceinline__ void st_row(void* base, bool h16, size_t idx, float v) {
  if (h16) reinterpret_cast<_Float16*>(base)[idx] = (_Float16)v;
  else reinterpret_cast<float*>(base)[idx] = v;
}

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
template <bool BINNED, bool AUX>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void neus_point_bwd_kernel(BwdArgs A, gs_grid_meta m) {
  __shared__ float red[4];
  __shared__ uint32_t row_tiles[4][64 * ROW_TS];
  __shared__ uint32_t st_val[BINNED ? 2 : 1][BINNED ? BINS_PER_LEVEL : 1][BINNED ? ST_SLOTS : 1];
  __shared__ uint16_t st_idx[BINNED ? 2 : 1][BINNED ? BINS_PER_LEVEL : 1][BINNED ? ST_SLOTS : 1];
  __shared__ uint32_t st_cnt[2][BINS_PER_LEVEL];
  if (BINNED) {
    if (threadIdx.x < 2 * BINS_PER_LEVEL) (&st_cnt[0][0])[threadIdx.x] = 0u;
    __syncthreads();
  }
#ifdef GS_BWD_STAMP
  long long t_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, t_last = wall_clock64();
#endif
  int hord = 0;                                   // ordinal of the current level among the hashed ones
  // weights are read through the constant address space: uniform, unchanged during the launch -> scalar loads
  typedef const __attribute__((address_space(4))) float* cfp;
  cfp cB = (cfp)A.color_B;
  uint32_t* tile = row_tiles[threadIdx.x >> 6];
  _Float16* tile_h = reinterpret_cast<_Float16*>(tile + (threadIdx.x & 63) * ROW_TS);    // this lane's row
  const size_t wave_p0 = (size_t)blockIdx.x * 256 + (threadIdx.x >> 6) * 64;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const int np = A.n * A.s;
  const bool valid = idx < np;
  const int i = valid ? idx : np - 1;
  const int ray = i / A.s;
  // ---- every input of the point is REQUESTED here, before anything is computed from it (see load_dx40: the round-5 form
  // interleaved loads, branches and first uses, and its head was a chain of ~10 dependent memory round trips)
  const uint8_t mk = A.mask[i];
  const float dist = A.dists[i], zv = A.z_vals[i], sdf = A.sdf[i];
  const float d_sdf_in = A.d_sdf[i], da_in = A.d_alpha[i], gerr_ray = A.d_gerr_ray[ray];
  float g[3], dg_in[3], dir[3], org[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    g[d] = A.grad[i * 3 + d];
    dg_in[d] = A.d_grad[i * 3 + d];
    dir[d] = A.rays_d[ray * 3 + d];
    org[d] = A.rays_o[ray * 3 + d];
  }
  const float inv_s_ = A.inv_s_dev ? *A.inv_s_dev : A.inv_s;
  // colour-MLP input gradient row dX[i, 32:72] (normal / feature part; the embedding part is re-read at the
  // end so that it does not occupy registers across the level loop)
  float dxh[40];
  load_dx40(A, i, 32, dxh);
  const bool on = valid && mk != 0;
  const float zm = zv + dist / 2.0f;
  float pt[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) pt[d] = org[d] + dir[d] * zm;
  if (valid && !A.rows16) {
    const size_t o3 = (size_t)i * 3;
    st_row(A.pts, false, o3 + 0, pt[0]); st_row(A.pts, false, o3 + 1, pt[1]); st_row(A.pts, false, o3 + 2, pt[2]);
  }
  // Every lane runs the whole body (wave-level run reduction below needs uniform control flow);
  // lanes that are out of bound / past the end carry zero upstream gradients and store nothing.
  const float live = on ? 1.0f : 0.0f;
  // ---- total gradient w.r.t. sdf and grad ----------------------------------------------------
  float d_sdf = d_sdf_in * live;
  float dg[3];
  const float gn = sqrtf((g[0] * g[0] + g[1] * g[1]) + g[2] * g[2]);
  const float eik = (gn > 0.f) ? gerr_ray * 2.0f * (gn - 1.0f) / gn : 0.0f;
#pragma unroll
  for (int d = 0; d < 3; ++d) dg[d] = (dg_in[d] + eik * g[d] + dxh[1 + d]) * live;
  float d_invs_local = 0.f;
  {   // NeuS alpha (InstantNeuS.py:276-293)
    const float da = da_in * live;
    const float cosv = (dir[0] * g[0] + dir[1] * g[1]) + dir[2] * g[2];
    const float c = -fmaxf(-cosv, 0.0f);
    const float est_next = sdf + c * dist / 2.0f, est_prev = sdf - c * dist / 2.0f;
    const float p = 1.0f / (1.0f + expf(-(est_prev * inv_s_)));
    const float q = 1.0f / (1.0f + expf(-(est_next * inv_s_)));
    const float raw = (p - q + 1e-5f) / (p + 1e-5f);
    if (da != 0.0f && raw >= 0.0f && raw <= 1.0f) {    // torch.clip passes the gradient on [min, max]
      const float dp = da * q / ((p + 1e-5f) * (p + 1e-5f));
      const float dq = -da / (p + 1e-5f);
      const float dprev = dp * p * (1.0f - p), dnext = dq * q * (1.0f - q);
      d_invs_local = dprev * est_prev + dnext * est_next;
      d_sdf += (dprev + dnext) * inv_s_;
      const float dc = (dnext - dprev) * inv_s_ * dist / 2.0f;
      if (cosv < 0.0f) {
#pragma unroll
        for (int d = 0; d < 3; ++d) dg[d] += dc * dir[d];
      }
    }
  }
  // ---- SDF network -------------------------------------------------------------------------------
  float p_[3], view[3], dG[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const float span = A.bound[2 * d + 1] - A.bound[2 * d];
    float qn = (pt[d] - A.bound[2 * d]) / span * 2.0f - 1.0f;
    const float inside = (qn >= -1.0f && qn <= 1.0f) ? 1.0f : 0.0f;
    qn = fminf(fmaxf(qn, -1.0f), 1.0f);
    p_[d] = qn;
    view[d] = (qn + 1.0f) / 2.0f;
    dG[d] = dg[d] * inside * 2.0f / span;
  }
  const bool r16 = A.rows16 != 0;
  const float rs = A.row_scale;
  const size_t o32 = (size_t)i * 32, o35 = (size_t)i * 35;
  float dov[32];
  dov[0] = d_sdf;
#pragma unroll
  for (int o = 1; o < 32; ++o) dov[o] = dxh[4 + (o - 1)] * live;
  if (r16) {                                     // lin | w0 rows are built in this lane's tile row
#pragma unroll
    for (int d = 0; d < 3; ++d) { tile_h[d] = (_Float16)(p_[d] * live); tile_h[40 + d] = (_Float16)(dG[d] * rs); }
#pragma unroll
    for (int d = 35; d < 40; ++d) { tile_h[d] = (_Float16)0.0f; tile_h[40 + d] = (_Float16)0.0f; }
  } else if (valid) {
#pragma unroll
    for (int o = 0; o < 32; ++o) st_row(A.d_out, false, o32 + o, dov[o] * rs);
#pragma unroll
    for (int d = 0; d < 3; ++d) { st_row(A.lin_in, false, o35 + d, p_[d] * live); st_row(A.dw0, false, o35 + d, dG[d] * rs); }
  }
  BSTAMP(0);
#pragma unroll 1
  for (int l = 0; l < GS_GRID_LEVELS; ++l) {
    const float scale = m.scale[l];
    float f[3];
    uint32_t gi[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const float pos = fmaf(scale, view[d], 0.5f);
      const float fl = floorf(pos);
      gi[d] = (uint32_t)(int)fl;
      f[d] = pos - fl;
    }
    const size_t off = (size_t)m.offset[l];
    uint32_t cidx[8];
    float v[8][2];
    float wc[8], e0 = 0.f, e1 = 0.f;
    half8 rec;
    if constexpr (AUX) {   // the forward kept this level's encoding and d enc / d x: one coalesced 16-byte load, no gathers
      // (requesting the next level's record here, one level ahead, was measured: 5 % slower at both batch sizes)
      rec = *reinterpret_cast<const half8*>(A.enc_aux + ((size_t)l * (size_t)np + i) * 8);
      grid_corners(m, l, gi, cidx);
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        float w = 1.0f;
#pragma unroll
        for (int d = 0; d < 3; ++d) w = w * (((c >> d) & 1) ? f[d] : (1.0f - f[d]));
        wc[c] = w;
      }
      e0 = on ? (float)rec[0] : 0.0f;          // (out-of-bound points have no record: select, never multiply)
      e1 = on ? (float)rec[1] : 0.0f;
    } else {
      grid_corners(m, l, gi, cidx);
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const uint32_t raw = *reinterpret_cast<const uint32_t*>(A.grid + (off + cidx[c]) * 2);
        v[c][0] = (float)__builtin_bit_cast(_Float16, (uint16_t)(raw & 0xffffu));
        v[c][1] = (float)__builtin_bit_cast(_Float16, (uint16_t)(raw >> 16));
      }
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        float w = 1.0f;
#pragma unroll
        for (int d = 0; d < 3; ++d) w = w * (((c >> d) & 1) ? f[d] : (1.0f - f[d]));
        wc[c] = w;
        e0 = fmaf(w, v[c][0], e0);
        e1 = fmaf(w, v[c][1], e1);
      }
    }
    // value path: d enc_f = sum_o d_out[o] W[o][3+2l+f]
    cfp wl = (cfp)(A.sdf_w + 3 + 2 * l);
    // Two partial sums per feature, over the even and the odd outputs: {even, odd} += {d_out[o], d_out[o + 1]} * {w[o],
    // w[o + 1]} is ONE v_pk_fma_f32 whose operands are register pairs as they lie -- two consecutive d_out registers and two
    // consecutive scalar weights.  (One chain per feature, round 5's form, makes d_out[o] the broadcast operand: the
    // compiler duplicates it into a pair with two v_mov per product, 64 per level and wave -- the level loop's largest
    // single item in the ISA, and with [level][output][feature] weights 54 s_mov + 32 v_writelane on top.)
    typedef float f2v __attribute__((ext_vector_type(2)));
    f2v acc0 = {0.f, 0.f}, acc1 = {0.f, 0.f};       // (written as 2-vectors: left to itself the vectoriser pairs the two FEATURES)
    if (A.sdf_wt) {
      // transposed copy [level][feature][output]: the level's 64 weights are contiguous -- scalar loads of whole pairs
      // instead of 32 strided ones (the elimination probes put this projection at 0.4 of the kernel's 2.7 ms)
      // (constant address space: the weights do not change during the launch, so the uniform reads become s_load --
      // through a plain global pointer the compiler issues one VECTOR load per weight, 64 per level and wave)
      cfp wt = (cfp)(A.sdf_wt + 64 * l);
#pragma unroll
      for (int o = 0; o < 32; o += 2) {
        const f2v d = {dov[o], dov[o + 1]};
        acc0 = __builtin_elementwise_fma(d, f2v{wt[o], wt[o + 1]}, acc0);
        acc1 = __builtin_elementwise_fma(d, f2v{wt[32 + o], wt[32 + o + 1]}, acc1);
      }
    } else {
#pragma unroll
      for (int o = 0; o < 32; o += 2) {
        const f2v d = {dov[o], dov[o + 1]};
        acc0 = __builtin_elementwise_fma(d, f2v{wl[o * 35], wl[(o + 1) * 35]}, acc0);
        acc1 = __builtin_elementwise_fma(d, f2v{wl[o * 35 + 1], wl[(o + 1) * 35 + 1]}, acc1);
      }
    }
    const float de0 = acc0[0] + acc0[1], de1 = acc1[0] + acc1[1];
    // gradient path: grad_d = (W0[d] + 1/2 sum g_lf dydx_lf,d) * inside * 2/span
    const float g0 = (float)(_Float16)wl[0], g1 = (float)(_Float16)wl[1];
    float dy0[3], dy1[3];
    float gacc[8][2];
#pragma unroll
    for (int c = 0; c < 8; ++c) { gacc[c][0] = de0 * wc[c]; gacc[c][1] = de1 * wc[c]; }
#pragma unroll
    for (int gd = 0; gd < 3; ++gd) {
      const int o0 = (gd == 0) ? 1 : 0, o1 = (gd == 2) ? 1 : 2;
      float a0 = 0.f, a1 = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float w = scale;
        w = w * ((k & 1) ? f[o0] : (1.0f - f[o0]));
        w = w * ((k & 2) ? f[o1] : (1.0f - f[o1]));
        const int cl = ((k & 1) << o0) | (((k >> 1) & 1) << o1);
        const int cr = cl | (1 << gd);
        if constexpr (!AUX) {
          a0 = fmaf(w, v[cr][0] - v[cl][0], a0);
          a1 = fmaf(w, v[cr][1] - v[cl][1], a1);
        }
        const float s0 = 0.5f * dG[gd] * g0 * w, s1 = 0.5f * dG[gd] * g1 * w;
        gacc[cr][0] += s0; gacc[cl][0] -= s0;
        gacc[cr][1] += s1; gacc[cl][1] -= s1;
      }
      if constexpr (AUX) {
        a0 = on ? (float)rec[2 + gd] : 0.0f;
        a1 = on ? (float)rec[5 + gd] : 0.0f;
      }
      dy0[gd] = a0;
      dy1[gd] = a1;
    }
    {
      const float l0 = (float)(_Float16)e0 * live, l1 = (float)(_Float16)e1 * live;
      const float v0 = rs * 0.5f * ((dG[0] * dy0[0] + dG[1] * dy0[1]) + dG[2] * dy0[2]);
      const float v1 = rs * 0.5f * ((dG[0] * dy1[0] + dG[1] * dy1[1]) + dG[2] * dy1[2]);
      if (r16) {
        tile_h[3 + 2 * l] = (_Float16)l0; tile_h[4 + 2 * l] = (_Float16)l1;
        tile_h[43 + 2 * l] = (_Float16)v0; tile_h[44 + 2 * l] = (_Float16)v1;
      } else if (valid) {
        st_row(A.lin_in, false, o35 + 3 + 2 * l, l0);
        st_row(A.lin_in, false, o35 + 3 + 2 * l + 1, l1);
        st_row(A.dw0, false, o35 + 3 + 2 * l, v0);
        st_row(A.dw0, false, o35 + 3 + 2 * l + 1, v1);
      }
    }
    BSTAMP(1);
    if (BINNED && m.hashed[l]) {
      // ---- pass 1 of bin-and-reduce: stage this workgroup's records of level l per bin, then copy them to the queues
      _Float16* tab16 = A.grid_grad16 + off * 2;
      const int buf = hord & 1;
      const bool act = lvl_prereduce(gacc, gi, on, lane) && on;
      BSTAMP(2);
      // the 8 slot reservations go out back to back and are waited for once (an LDS atomic with return is a ~100-cycle
      // round trip: reserved one corner at a time they are 8 dependent round trips per level and wave)
      uint32_t slot_[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        slot_[c] = 0u;
        if (act && (gacc[c][0] != 0.0f || gacc[c][1] != 0.0f)) slot_[c] = atomicAdd(&st_cnt[buf][cidx[c] >> BIN_SHIFT], 1u);
      }
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        if (act && (gacc[c][0] != 0.0f || gacc[c][1] != 0.0f)) {
          const uint32_t e = cidx[c], bin = e >> BIN_SHIFT;
          const uint32_t packed = pack2h(gacc[c][0] * A.grad_scale16, gacc[c][1] * A.grad_scale16);
          const uint32_t slot = slot_[c];
          if (slot < (uint32_t)ST_SLOTS) {
            st_idx[buf][bin][slot] = (uint16_t)(e & (BIN_ENTRIES - 1));
            st_val[buf][bin][slot] = packed;
          } else {                                  // staging bin full: this record goes out as an atomic
            __builtin_amdgcn_global_atomic_fadd_v2f16((__attribute__((address_space(1))) half2a*)(tab16 + (size_t)e * 2),
                                                      __builtin_bit_cast(half2a, packed));
          }
        }
      }
      BSTAMP(3);
      __syncthreads();
      BSTAMP(4);
      // wave w copies bins 16 w .. 16 w + 15 to this workgroup's segments, FOUR bins per step (16 lanes each, <= 3
      // strides of 16 records): 4 steps per level instead of 16 serial bin copies -- the elimination probes put
      // staging + copy + barrier at 1.0 of the kernel's 2.8 ms.  Staging buffer `buf` is appended to again two hashed
      // levels from now, i.e. after the NEXT level's barrier, which every wave reaches only after this copy.
      {
        // (round 5: the four fill counts, then all twelve record pairs, are requested from LDS before the first store
        // goes out -- one count -> records -> stores chain per bin group was 8 dependent LDS round trips per level and
        // wave; the segment offset is 32-bit arithmetic: the launcher checks that the queues hold < 2^32 records)
        const int wv = threadIdx.x >> 6, sub = lane >> 4, s16 = lane & 15;
        const uint32_t nblk = gridDim.x;
        constexpr int NJ = BINS_PER_LEVEL / 16, NK = (ST_SLOTS + 15) / 16;
        uint32_t cn[NJ], rv[NJ][NK];
        uint16_t ri[NJ][NK];
#pragma unroll
        for (int j = 0; j < NJ; ++j) cn[j] = st_cnt[buf][16 * wv + 4 * j + sub];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int b = 16 * wv + 4 * j + sub;
#pragma unroll
          for (int k = 0; k < NK; ++k) {               // (slots past the fill count hold stale records: read, never stored)
            ri[j][k] = st_idx[buf][b][s16 + 16 * k];
            rv[j][k] = st_val[buf][b][s16 + 16 * k];
          }
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const uint32_t b = 16 * wv + 4 * j + sub;
          const uint32_t c = cn[j] < (uint32_t)ST_SLOTS ? cn[j] : (uint32_t)ST_SLOTS;
          const uint32_t seg = ((uint32_t)hord * BINS_PER_LEVEL + b) * nblk + blockIdx.x;
          const uint32_t at = seg * ST_SLOTS + s16;
#pragma unroll
          for (int k = 0; k < NK; ++k) {
            if (s16 + 16 * k < c) {
              A.q_idx[(size_t)(at + 16 * k)] = ri[j][k];
              A.q_val[(size_t)(at + 16 * k)] = rv[j][k];
            }
          }
          if (s16 == 0) A.q_cnt[seg] = (uint8_t)c;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();              // every lane has read its bins' counts before they are reset
        if (lane < 16) st_cnt[buf][16 * wv + lane] = 0u;
      }
      ++hord;
      BSTAMP(5);
    } else {
      lvl_scatter(A.grid_grad ? A.grid_grad + off * 2 : nullptr, A.grid_grad16 ? A.grid_grad16 + off * 2 : nullptr,
                  A.grad_scale16, cidx, gacc, gi, on, lane);
      BSTAMP(6);
    }
  }
  // ---- colour embedding sin(pts @ B): d arg = d emb * cos(arg)
  const size_t np_all = (size_t)np;
  if (r16) {
    // the embedding part of the dX row: all five loads at once, and d arg formed in registers BEFORE the row flushes (a
    // load consumed behind the flushes' predicated stores is waited for with vmcnt(0), i.e. behind their write
    // acknowledgements; the round-5 form fetched and waited for 8 columns at a time, five round trips per wave)
    uint32_t dap[20];
    {
      float dxe[40];
      load_dx40(A, i, 0, dxe);
#pragma unroll
      for (int c = 0; c < 40; c += 2) {
        float da[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int cc = c + e < 33 ? c + e : 32;
          const float arg = (pt[0] * cB[cc] + pt[1] * cB[33 + cc]) + pt[2] * cB[66 + cc];
          da[e] = c + e < 33 ? dxe[c + e] * emb_cos(arg) * live * rs : 0.0f;
        }
        dap[c >> 1] = pack2h(da[0], da[1]);
      }
    }
    // lin | w0 rows are complete: flush them, then reuse the tile for d_out, d_arg and pts
    wave_sync_lds();
    tile_flush(tile, 0, 20, A.lin_in, wave_p0, np_all, lane, A.row_stride16);
    tile_flush(tile, 20, 20, A.dw0, wave_p0, np_all, lane, A.row_stride16);
    wave_sync_lds();
    uint32_t* trow = tile + lane * ROW_TS;
#pragma unroll
    for (int o = 0; o < 16; ++o) trow[o] = pack2h(dov[2 * o] * rs, dov[2 * o + 1] * rs);
#pragma unroll
    for (int e = 0; e < 20; ++e) trow[16 + e] = dap[e];
    trow[36] = pack2h(pt[0], pt[1]); trow[37] = pack2h(pt[2], 1.0f); trow[38] = 0u; trow[39] = 0u;   // (x, y, z, 1): the 1 yields column sums
    wave_sync_lds();
    tile_flush(tile, 0, 16, A.d_out, wave_p0, np_all, lane, A.row_stride16);
    tile_flush(tile, 16, 20, A.d_arg, wave_p0, np_all, lane, A.row_stride16);
    tile_flush(tile, 36, 4, A.pts, wave_p0, np_all, lane, A.row_stride16);
  } else if (valid) {
    float dxe[40];
    load_dx40(A, i, 0, dxe);
#pragma unroll
    for (int c = 0; c < 33; ++c) {
      const float arg = (pt[0] * cB[c] + pt[1] * cB[33 + c]) + pt[2] * cB[66 + c];
      st_row(A.d_arg, false, (size_t)i * 33 + c, dxe[c] * emb_cos(arg) * live * rs);
    }
  }
  BSTAMP(7);
#ifdef GS_BWD_STAMP
  if (threadIdx.x == 0 && A.stamp)
    for (int q = 0; q < 8; ++q) atomicAdd(A.stamp + q, (unsigned long long)t_acc[q]);
#endif
  // one atomic per workgroup for d inv_s
  const float ws = gs_wave_sum(d_invs_local);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ws;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float t = (red[0] + red[1]) + (red[2] + red[3]);
    if (t != 0.0f) atomicAdd(A.d_inv_s, t);
  }
}

// pass 2 of bin-and-reduce: one workgroup per (hashed level, bin); see the file header
__device__ __forceinline__ long long fix24(uint32_t h) {       // fp16 bits -> value * 2^24, exact (|v| <= 65504 < 2^16)
  const float v = fminf(fmaxf((float)__builtin_bit_cast(_Float16, (uint16_t)h), -65504.0f), 65504.0f);   // (inf: saturate)
  return (long long)(v * 16777216.0f);
}

__global__ __launch_bounds__(1024) void grid_bin_reduce_kernel(const uint16_t* __restrict__ q_idx,
                                                               const uint32_t* __restrict__ q_val,
                                                               const uint8_t* __restrict__ q_cnt, int nblk,
                                                               int cnt_in_lds, _Float16* __restrict__ tab16,
                                                               gs_grid_meta m) {
  extern __shared__ unsigned long long acc[];          // [BIN_ENTRIES][2] 64-bit fixed point (two's complement)
                                                       // | (cnt_in_lds) the bin's nblk fill counts
  __shared__ int nonfinite;                            // a NaN / inf record was seen: fix24 saturates, so a second scan
                                                       // writes those records' own bits (a diverged step stays visible)
  if (threadIdx.x == 0) nonfinite = 0;
  const int q = blockIdx.x, h = q / BINS_PER_LEVEL, b = q - h * BINS_PER_LEVEL;
  int l = 0;
  for (int k = 0, seen = 0; k < GS_GRID_LEVELS; ++k)
    if (m.hashed[k]) { if (seen == h) l = k; ++seen; }
  const int tid = threadIdx.x;
  for (int e = tid; e < 2 * BIN_ENTRIES / 2; e += 1024) reinterpret_cast<uint4*>(acc)[e] = make_uint4(0u, 0u, 0u, 0u);
  const uint8_t* cnt = q_cnt + (size_t)q * nblk;
  const uint16_t* qi = q_idx + (size_t)q * nblk * ST_SLOTS;
  const uint32_t* qv = q_val + (size_t)q * nblk * ST_SLOTS;
  // The fill counts go to LDS first: read from memory inside the loop below, every segment cost TWO dependent round
  // trips (count, then records) in a rolled loop of nblk / 85 steps -- 108 steps at 32768 rays, the bulk of this
  // kernel's 533 us.  With the counts at hand four segments' record loads are issued back to back.
  uint8_t* lcnt = reinterpret_cast<uint8_t*>(acc + 2 * BIN_ENTRIES);
  if (cnt_in_lds) {
    for (int i0 = tid; i0 < nblk; i0 += 1024 * 4) {
      uint8_t c4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) c4[u] = i0 + 1024 * u < nblk ? cnt[i0 + 1024 * u] : (uint8_t)0;
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (i0 + 1024 * u < nblk) lcnt[i0 + 1024 * u] = c4[u];
    }
  }
  // a segment (one pass-1 workgroup's records for this bin: ST_SLOTS = 48 slots, `cnt` of them filled) is read by 12
  // threads, 4 records each: one 8-byte index load + one 16-byte value load
  constexpr int TPS = ST_SLOTS / 4;                    // threads per segment
  constexpr int SPB = 1020 / TPS;                      // segments per workgroup step (85)
  const int sub = tid / TPS, part = tid - sub * TPS;
  __syncthreads();                                     // (zeroed accumulators, counts in LDS)
  if (sub < SPB) {
    constexpr int UN = 4;                              // segments in flight per thread
    const int s0 = 4 * part;
    for (int sg0 = sub; sg0 < nblk; sg0 += UN * SPB) {
      int c[UN];
      uint2 ix[UN];
      uint4 vv[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int sg = sg0 + u * SPB;
        c[u] = sg < nblk ? (int)(cnt_in_lds ? lcnt[sg] : cnt[sg]) : 0;
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        if (s0 < c[u]) {
          const size_t at = (size_t)(sg0 + u * SPB) * ST_SLOTS + s0;
          ix[u] = *reinterpret_cast<const uint2*>(qi + at);
          vv[u] = *reinterpret_cast<const uint4*>(qv + at);
        }
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        if (s0 < c[u]) {
          const uint32_t iw[4] = {ix[u].x & 0xffffu, ix[u].x >> 16, ix[u].y & 0xffffu, ix[u].y >> 16};
          const uint32_t vw[4] = {vv[u].x, vv[u].y, vv[u].z, vv[u].w};
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            if (s0 + k < c[u]) {
              atomicAdd(&acc[2 * iw[k]], (unsigned long long)fix24(vw[k] & 0xffffu));
              atomicAdd(&acc[2 * iw[k] + 1], (unsigned long long)fix24(vw[k] >> 16));
              if ((vw[k] & 0x7c00u) == 0x7c00u || (vw[k] & 0x7c000000u) == 0x7c000000u) nonfinite = 1;
            }
          }
        }
      }
    }
  }
  __syncthreads();
  // the bin's entries: existing content (zero, or the atomics of overflow records) + the exact sums, rounded once
  uint32_t* out = reinterpret_cast<uint32_t*>(tab16) + (size_t)m.offset[l] + (size_t)b * BIN_ENTRIES;
  for (int e = tid; e < BIN_ENTRIES / 4; e += 1024) {
    const uint4 old = reinterpret_cast<const uint4*>(out)[e];
    const uint32_t ow[4] = {old.x, old.y, old.z, old.w};
    uint32_t nw[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float o0 = (float)__builtin_bit_cast(_Float16, (uint16_t)(ow[k] & 0xffffu));
      const float o1 = (float)__builtin_bit_cast(_Float16, (uint16_t)(ow[k] >> 16));
      const float a0 = (float)((double)(long long)acc[2 * (4 * e + k)] * (1.0 / 16777216.0));
      const float a1 = (float)((double)(long long)acc[2 * (4 * e + k) + 1] * (1.0 / 16777216.0));
      nw[k] = pack2h(o0 + a0, o1 + a1);
    }
    reinterpret_cast<uint4*>(out)[e] = make_uint4(nw[0], nw[1], nw[2], nw[3]);
  }
  if (nonfinite) {      // rare slow path: entries that received a NaN / inf record become that value, as with atomics
    __syncthreads();
    uint16_t* out16 = reinterpret_cast<uint16_t*>(out);
    for (int sg = tid; sg < nblk; sg += 1024) {
      const int c = cnt[sg];
      for (int k = 0; k < c; ++k) {
        const uint32_t v = qv[(size_t)sg * ST_SLOTS + k], i = qi[(size_t)sg * ST_SLOTS + k];
        if ((v & 0x7c00u) == 0x7c00u) out16[2 * i] = (uint16_t)(v & 0xffffu);
        if ((v & 0x7c000000u) == 0x7c000000u) out16[2 * i + 1] = (uint16_t)(v >> 16);
      }
    }
  }
}

gs_grid_meta host_meta() {
  gs_grid_meta m;
  gs_grid_meta_default(&m);
  return m;
}

size_t bin_workgroups(size_t np) { return (np + 255) / 256; }

}  // namespace

extern "C" size_t gs_neus_bin_workspace_bytes(int n_points) {
  const gs_grid_meta m = host_meta();
  size_t nh = 0;
  for (int l = 0; l < GS_GRID_LEVELS; ++l) nh += m.hashed[l] ? 1 : 0;
  const size_t nq = nh * BINS_PER_LEVEL, nblk = bin_workgroups((size_t)(n_points > 0 ? n_points : 0));
  return gs_align(nq * nblk) + nq * nblk * ST_SLOTS * 6 + 512;       // fill counts | values | indices
}

extern "C" int gs_neus_backward_rays(const float* alpha, const void* rgb, const float* z_mid, const float* grad,
                                     const uint8_t* mask, const float* d_color, const float* d_depth,
                                     const float* d_depth_var, const float* d_normal, const float* d_weight_sum,
                                     float* d_alpha, float* d_rgb, float* d_grad, int n, int s, gs_stream_t stream) {
  GS_REQUIRE(alpha && rgb && z_mid && grad && mask && d_color && d_depth && d_depth_var && d_normal && d_weight_sum &&
                 d_alpha && d_rgb && d_grad, "neus_backward_rays: null pointer");
  GS_REQUIRE(n >= 0 && s > 0 && s <= 128, "neus_backward_rays: 1 <= samples per ray <= 128 (got %d)", s);
  if (n == 0) return GS_OK;
  GS_TIMING_PRE();
  neus_ray_bwd_kernel<<<gs_cdiv(n, 4), 256, 0, (hipStream_t)stream>>>(alpha, (const _Float16*)rgb, z_mid, grad, mask,
                                                                     d_color, d_depth, d_depth_var, d_normal,
                                                                     d_weight_sum, d_alpha, d_rgb, d_grad, n, s);
  GS_CHECK_LAUNCH("neus_backward_rays");
  return GS_OK;
}

static int backward_points_impl(const float* rays_o, const float* rays_d, const float* z_vals,
                                       const float* dists, const void* grid, const float* sdf_w,
                                       const float* color_B, float inv_s, const float* inv_s_dev, const float* bound_host,
                                       const float* sdf,
                                       const float* grad, const uint8_t* mask, const float* d_alpha,
                                       const float* d_sdf, const float* d_grad, const void* dX, int dx_dtype,
                                       float dx_scale, const float* d_gerr_ray,
                                       void* grid_grad, int grid_grad_dtype, float grid_grad_scale, void* d_out,
                                       void* lin_in, void* dw0, void* d_arg, void* pts, int row_dtype,
                                       float row_scale, int row_stride, float* d_inv_s, int n, int s,
                                       void* bin_ws, size_t bin_ws_bytes, const float* sdf_wt, const void* enc_aux,
                                       gs_stream_t stream) {
  GS_REQUIRE(row_stride == 0 || (row_dtype == GS_F16 && row_stride >= 40 && row_stride % 8 == 0),
             "neus_backward_points: row_stride needs f16 rows, >= 40, a multiple of 8");
  GS_REQUIRE(dx_dtype == GS_F32 || dx_dtype == GS_F16, "neus_backward_points: dX dtype f32 or f16");
  GS_REQUIRE(row_dtype == GS_F32 || row_dtype == GS_F16, "neus_backward_points: row dtype f32 or f16");
  GS_REQUIRE(dx_scale > 0.0f && row_scale > 0.0f, "neus_backward_points: scales must be positive");
  GS_REQUIRE(grid_grad_dtype == GS_F32 || grid_grad_dtype == GS_F16, "neus_backward_points: grid_grad dtype f32 or f16");
  GS_REQUIRE(rays_o && rays_d && z_vals && dists && grid && sdf_w && color_B && bound_host && sdf && grad && mask &&
                 d_alpha && d_sdf && d_grad && dX && d_gerr_ray && grid_grad && d_out && lin_in && dw0 && d_arg && pts && d_inv_s,
             "neus_backward_points: null pointer");
  GS_REQUIRE(n >= 0 && s > 0, "neus_backward_points: bad shape");
  if (n == 0) return GS_OK;
  BwdArgs A;
  A.rays_o = rays_o; A.rays_d = rays_d; A.z_vals = z_vals; A.dists = dists;
  A.grid = (const _Float16*)grid; A.sdf_w = sdf_w; A.sdf_wt = sdf_wt; A.color_B = color_B; A.inv_s = inv_s; A.inv_s_dev = inv_s_dev;
  for (int k = 0; k < 6; ++k) A.bound[k] = bound_host[k];
  A.sdf = sdf; A.grad = grad; A.mask = mask; A.d_alpha = d_alpha; A.d_sdf = d_sdf; A.d_grad = d_grad; A.dX = dX;
  A.d_gerr_ray = d_gerr_ray;
  A.enc_aux = (const _Float16*)enc_aux;
  A.grid_grad = grid_grad_dtype == GS_F32 ? (float*)grid_grad : nullptr;
  A.grid_grad16 = grid_grad_dtype == GS_F16 ? (_Float16*)grid_grad : nullptr;
  A.grad_scale16 = grid_grad_scale;
  A.d_out = d_out; A.lin_in = lin_in; A.dw0 = dw0; A.d_arg = d_arg; A.pts = pts;
  A.rows16 = row_dtype == GS_F16; A.row_scale = row_scale; A.row_stride16 = row_stride / 8; A.dx16 = dx_dtype == GS_F16; A.dx_inv_scale = 1.0f / dx_scale;
  A.d_inv_s = d_inv_s; A.n = n; A.s = s;
  A.q_idx = nullptr; A.q_val = nullptr; A.q_cnt = nullptr;
#ifdef GS_BWD_STAMP
  A.stamp = nullptr;
#endif
  const gs_grid_meta m = host_meta();
  if (!bin_ws) {
    if (enc_aux) neus_point_bwd_kernel<false, true><<<gs_cdiv(n * s, 256), 256, 0, (hipStream_t)stream>>>(A, m);
    else neus_point_bwd_kernel<false, false><<<gs_cdiv(n * s, 256), 256, 0, (hipStream_t)stream>>>(A, m);
    GS_CHECK_LAUNCH("neus_backward_points");
    return GS_OK;
  }
  // ---- binned table gradient: queues carved from the caller's workspace
  GS_REQUIRE(grid_grad_dtype == GS_F16, "neus_backward_points_binned: the table gradient must be f16 (loss-scaled)");
  int nh = 0;
  for (int l = 0; l < GS_GRID_LEVELS; ++l) {
    if (!m.hashed[l]) continue;
    GS_REQUIRE(m.size[l] == (uint32_t)BINS_PER_LEVEL * BIN_ENTRIES, "neus_backward_points_binned: hashed level %d has %u entries", l, m.size[l]);
    ++nh;
  }
  const size_t nq = (size_t)nh * BINS_PER_LEVEL;
  const size_t need = gs_neus_bin_workspace_bytes(n * s);
  if (bin_ws_bytes < need) {
    gs_set_error("neus_backward_points_binned: workspace too small (%zu < %zu bytes)", bin_ws_bytes, need);
    return GS_ERR_WORKSPACE;
  }
  const size_t nblk = bin_workgroups((size_t)n * s);
  // (pass 1 addresses the queues with 32-bit record offsets: 26 GB of queues, ~32 M sample points, is the limit)
  GS_REQUIRE(nq * nblk * ST_SLOTS < ((size_t)1 << 32), "neus_backward_points_binned: %d x %d sample points exceed the record queues' 32-bit offsets", n, s);
  char* base = (char*)gs_align((size_t)bin_ws);
  A.q_cnt = (uint8_t*)base;                                       // [nq][nblk]
  A.q_val = (uint32_t*)(base + gs_align(nq * nblk));              // [nq][nblk][ST_SLOTS]
  A.q_idx = (uint16_t*)((char*)A.q_val + nq * nblk * ST_SLOTS * 4);
#ifdef GS_BWD_STAMP
  static unsigned long long* stamp_dev = nullptr;
  if (!stamp_dev) (void)hipMalloc((void**)&stamp_dev, 16 * sizeof(unsigned long long));
  (void)hipMemsetAsync(stamp_dev, 0, 16 * sizeof(unsigned long long), (hipStream_t)stream);
  A.stamp = stamp_dev;
#endif
  GS_TIMING_PRE();
  if (enc_aux) neus_point_bwd_kernel<true, true><<<(unsigned)nblk, 256, 0, (hipStream_t)stream>>>(A, m);
  else neus_point_bwd_kernel<true, false><<<(unsigned)nblk, 256, 0, (hipStream_t)stream>>>(A, m);
  GS_CHECK_LAUNCH("neus_backward_points_binned");
#ifdef GS_BWD_STAMP
  {
    unsigned long long h[16];
    (void)hipStreamSynchronize((hipStream_t)stream);
    (void)hipMemcpy(h, stamp_dev, sizeof(h), hipMemcpyDeviceToHost);
    fprintf(stderr, "[bwd stamp] %zu workgroups; us per workgroup (thread 0): prolog %.2f | per step: value-path %.2f prereduce %.2f stage %.2f barrier %.2f copy-out %.2f | dense levels %.2f | tail %.2f\n",
            nblk, h[0] / 100.0 / nblk, h[1] / 100.0 / nblk, h[2] / 100.0 / nblk, h[3] / 100.0 / nblk, h[4] / 100.0 / nblk,
            h[5] / 100.0 / nblk, h[6] / 100.0 / nblk, h[7] / 100.0 / nblk);
  }
#endif
  static GsLdsLimit limit;
  const size_t acc_bytes = (size_t)2 * BIN_ENTRIES * sizeof(unsigned long long);
  const size_t cnt_bytes = gs_align(nblk, 16);                        // the bin's fill counts, if they fit beside the sums
  const int cnt_in_lds = acc_bytes + cnt_bytes + 256 <= (size_t)160 * 1024;
  const size_t lds = acc_bytes + (cnt_in_lds ? cnt_bytes : 0);
  if (int rc = limit.raise((const void*)grid_bin_reduce_kernel, lds, "grid_bin_reduce")) return rc;
  grid_bin_reduce_kernel<<<(unsigned)nq, 1024, lds, (hipStream_t)stream>>>(A.q_idx, A.q_val, A.q_cnt, (int)nblk,
                                                                           cnt_in_lds, A.grid_grad16, m);
  GS_CHECK_LAUNCH("grid_bin_reduce");
  return GS_OK;
}

extern "C" int gs_neus_backward_points(const float* rays_o, const float* rays_d, const float* z_vals,
                                       const float* dists, const void* grid, const float* sdf_w,
                                       const float* color_B, float inv_s, const float* inv_s_dev, const float* bound_host,
                                       const float* sdf,
                                       const float* grad, const uint8_t* mask, const float* d_alpha,
                                       const float* d_sdf, const float* d_grad, const void* dX, int dx_dtype,
                                       float dx_scale, const float* d_gerr_ray,
                                       void* grid_grad, int grid_grad_dtype, float grid_grad_scale, void* d_out,
                                       void* lin_in, void* dw0, void* d_arg, void* pts, int row_dtype,
                                       float row_scale, int row_stride, float* d_inv_s, int n, int s,
                                       const void* enc_aux, gs_stream_t stream) {
  return backward_points_impl(rays_o, rays_d, z_vals, dists, grid, sdf_w, color_B, inv_s, inv_s_dev, bound_host, sdf, grad,
                              mask, d_alpha, d_sdf, d_grad, dX, dx_dtype, dx_scale, d_gerr_ray, grid_grad, grid_grad_dtype,
                              grid_grad_scale, d_out, lin_in, dw0, d_arg, pts, row_dtype, row_scale, row_stride, d_inv_s, n,
                              s, nullptr, 0, nullptr, enc_aux, stream);
}

extern "C" int gs_neus_backward_points_binned(const float* rays_o, const float* rays_d, const float* z_vals,
                                              const float* dists, const void* grid, const float* sdf_w,
                                              const float* color_B, float inv_s, const float* inv_s_dev,
                                              const float* bound_host, const float* sdf, const float* grad,
                                              const uint8_t* mask, const float* d_alpha, const float* d_sdf,
                                              const float* d_grad, const void* dX, int dx_dtype, float dx_scale,
                                              const float* d_gerr_ray, void* grid_grad, float grid_grad_scale,
                                              void* d_out, void* lin_in, void* dw0, void* d_arg, void* pts,
                                              int row_dtype, float row_scale, int row_stride, float* d_inv_s, int n,
                                              int s, void* bin_ws, size_t bin_ws_bytes, const float* sdf_wt,
                                              const void* enc_aux, gs_stream_t stream) {
  GS_REQUIRE(bin_ws, "neus_backward_points_binned: null workspace");
  return backward_points_impl(rays_o, rays_d, z_vals, dists, grid, sdf_w, color_B, inv_s, inv_s_dev, bound_host, sdf, grad,
                              mask, d_alpha, d_sdf, d_grad, dX, dx_dtype, dx_scale, d_gerr_ray, grid_grad, GS_F16,
                              grid_grad_scale, d_out, lin_in, dw0, d_arg, pts, row_dtype, row_scale, row_stride, d_inv_s, n,
                              s, bin_ws, bin_ws_bytes, sdf_wt, enc_aux, stream);
}
