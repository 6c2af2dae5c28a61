// Mapping hot path: hash-grid NeuS renderer (reference: src/render.py:73-175,
// src/InstantNeuS.py:12-370; tiny-cuda-nn HashGrid + FullyFusedMLP restated per SURVEY App. B).
//
//   render_sample_kernel   one lane per ray: ray/AABB far bound, stratified + near-surface
//                          samples, merge of the two sorted runs (= the reference's torch.sort)
//   (the reference forces the first 100 points valid when no point is in bound, InstantNeuS.py:311-312: a
//    one-workgroup second pass of neus_point_kernel)
//   neus_point_kernel      one lane per sample point: 16-level hash-grid gathers (8 corners x
//                          2 fp16 features = one 4-byte load each; the table is L2/MALL resident),
//                          the SDF linear row streamed level by level, the ANALYTIC d sdf/d x
//                          from the same 8 corner values (no autograd graph, no second gather),
//                          NeuS alpha, and the 80-wide fp16 colour-MLP input row
//   neus_mlp_kernel        FullyFusedMLP 80->64->64->16 on MFMA (v_mfma_f32_32x32x16_f16):
//                          H^T = W X^T so that weights are the A operand (kept in VGPRs for the
//                          whole launch) and the point rows are read as B fragments straight
//                          from their [point][k] rows; activations between layers go through a
//                          wave-private LDS tile; sigmoid; fp16 rgb
//   neus_ray_kernel        one lane per ray: alpha compositing (exclusive cumprod), colour,
//                          depth, variance, normal, weight sum, per-ray eikonal partial sum
#include "common.h"
#include "neus_common.h"
#include <math.h>
#include <atomic>

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));   // (a native vector: what the non-temporal builtins take)

// ---------------------------------------------------------------------------------------
// sample placement
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void render_sample_kernel(
    const float* __restrict__ rays_o, const float* __restrict__ rays_d, const float* __restrict__ gt_depth,
    const float* __restrict__ bound, const float* __restrict__ t_samples, const float* __restrict__ t_surface,
    const float* __restrict__ perturb, float gt_max_host, const float* __restrict__ gt_max_dev,
    float* __restrict__ z_vals, float* __restrict__ dists, int n, int ns, int nsurf) {
  const float gt_max = gt_max_dev ? *gt_max_dev : gt_max_host;
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= n) return;
  const float o[3] = {rays_o[r * 3 + 0], rays_o[r * 3 + 1], rays_o[r * 3 + 2]};
  const float d[3] = {rays_d[r * 3 + 0], rays_d[r * 3 + 1], rays_d[r * 3 + 2]};
  // far_bb = min_dim max((b0-o)/d, (b1-o)/d) + 0.01        (render.py:112-118)
  float far_bb = INFINITY;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float t0 = (bound[2 * k + 0] - o[k]) / d[k];
    const float t1 = (bound[2 * k + 1] - o[k]) / d[k];
    // torch.max / torch.min propagate NaN
    const float tm = (t0 != t0 || t1 != t1) ? NAN : fmaxf(t0, t1);
    far_bb = (tm != tm || far_bb != far_bb) ? NAN : fminf(far_bb, tm);
  }
  far_bb = far_bb + 0.01f;
  const bool has_depth = gt_depth != nullptr;
  const float gd = has_depth ? gt_depth[r] : 0.0f;
  float nearv, farv;
  if (has_depth) {
    nearv = gd * 0.01f;
    farv = fminf(fmaxf(far_bb, 0.0f), gt_max * 1.2f);     // torch.clamp(far_bb, 0, max)
    if (far_bb != far_bb) farv = far_bb;
  } else {
    nearv = 0.01f;
    farv = far_bb;
    nsurf = 0;
  }
  const int total = ns + nsurf;
  float* zo = z_vals + (size_t)r * total;
  float* dd = dists + (size_t)r * total;
  const float span = farv - nearv;
  auto zu = [&](int j) { return nearv + span * t_samples[j]; };           // :147
  auto zs = [&](int j) -> float {                                         // :150-166
    float z = zu(j);
    if (perturb) {
      const float lo = (j == 0) ? z : 0.5f * (zu(j - 1) + z);
      const float hi = (j == ns - 1) ? z : 0.5f * (z + zu(j + 1));
      z = lo + (hi - lo) * perturb[j];
    }
    return z;
  };
  auto zf = [&](int j) -> float {                                         // :126-141
    const float t = t_surface[j];
    if (gd > 0.0f) {
      const float snr = (1.0f - 0.1f) * gd, sfar = (1.0f + 0.1f) * gd;
      return (snr + (sfar - snr) * t) * 1.0f + (0.001f + (gt_max - 0.001f) * t) * (1.0f - 1.0f);
    }
    const float vd = gd * 0.0f;                                           // gt_depth * valid_mask
    const float snr = (1.0f - 0.1f) * vd, sfar = (1.0f + 0.1f) * vd;
    return (snr + (sfar - snr) * t) * 0.0f + (0.001f + (gt_max - 0.001f) * t) * (1.0f - 0.0f);
  };
  // merge the two monotone runs (= the reference's torch.sort of their concatenation, render.py:168-171).  Both are
  // ascending except in two degenerate cases, where they are walked backwards: far < near (a ray whose box exit lies
  // behind the camera: far clamps to 0) and, for rays without depth, a batch maximum below 0.001.  WITHOUT near-surface
  // samples (no depth image) the reference does not sort at all (:162): a descending run stays descending.
  const bool rev_a = span < 0.0f && nsurf > 0, rev_b = !(gd > 0.0f) && gt_max < 0.001f;
  auto zsa = [&](int j) { return zs(rev_a ? ns - 1 - j : j); };
  auto zfa = [&](int j) { return zf(rev_b ? nsurf - 1 - j : j); };
  int a = 0, b = 0;
  float prev = 0.f;
  for (int k = 0; k < total; ++k) {
    float z;
    if (b >= nsurf) z = zsa(a++);
    else if (a >= ns) z = zfa(b++);
    else {
      const float za = zsa(a), zb = zfa(b);
      if (za <= zb) { z = za; ++a; } else { z = zb; ++b; }
    }
    zo[k] = z;
    if (k > 0) dd[k - 1] = z - prev;
    prev = z;
  }
  dd[total - 1] = span / (float)ns;    // mean over identical columns of (far-near)/N_samples (:149)
}

// Wave-per-ray variant (n_samples, n_surface <= 64): lane j owns stratified sample j and
// near-surface sample j; the merge position of each is its index plus its rank in the other run
// (ties resolved like the sequential merge: stratified first), so every lane writes its own
// slot and all global traffic is coalesced.  Values are bit-identical to the sequential kernel.
__global__ __launch_bounds__(256) void render_sample_wave_kernel(
    const float* __restrict__ rays_o, const float* __restrict__ rays_d, const float* __restrict__ gt_depth,
    const float* __restrict__ bound, const float* __restrict__ t_samples, const float* __restrict__ t_surface,
    const float* __restrict__ perturb, float gt_max_host, const float* __restrict__ gt_max_dev,
    float* __restrict__ z_vals, float* __restrict__ dists, int n, int ns, int nsurf) {
  __shared__ float sa[4][64], sb[4][64], sz[4][128];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float gt_max = gt_max_dev ? *gt_max_dev : gt_max_host;
  if (!gt_max_dev && gt_depth && gt_max_host == -INFINITY) {
    // the batch maximum (render.py:121,140: `gt_depth.max()`, NaN if any depth is NaN) taken HERE, by every workgroup for
    // itself -- n floats from L2 -- instead of by a reduction launch in front of this one (5 of a render batch's 167 us)
    __shared__ float smax[4];
    __shared__ int snan[4];
    float m = -INFINITY;
    int isnan_ = 0;
    for (int i0 = threadIdx.x * 4; i0 < n; i0 += 1024) {
      float v[4];
      if (i0 + 4 <= n) {
        const float4 q = *reinterpret_cast<const float4*>(gt_depth + i0);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = i0 + u < n ? gt_depth[i0 + u] : -INFINITY;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) { m = fmaxf(m, v[u]); isnan_ |= (v[u] != v[u]) ? 1 : 0; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      m = fmaxf(m, __shfl_xor(m, o));
      isnan_ |= __shfl_xor(isnan_, o);
    }
    if (lane == 0) { smax[wave] = m; snan[wave] = isnan_; }
    __syncthreads();
    m = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
    gt_max = (snan[0] | snan[1] | snan[2] | snan[3]) ? NAN : m;
  }
  const int r = blockIdx.x * 4 + wave;
  if (r >= n) return;
  float far_bb = INFINITY;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float o = rays_o[r * 3 + k], d = rays_d[r * 3 + k];
    const float t0 = (bound[2 * k + 0] - o) / d;
    const float t1 = (bound[2 * k + 1] - o) / d;
    const float tm = (t0 != t0 || t1 != t1) ? NAN : fmaxf(t0, t1);
    far_bb = (tm != tm || far_bb != far_bb) ? NAN : fminf(far_bb, tm);
  }
  far_bb = far_bb + 0.01f;
  const bool has_depth = gt_depth != nullptr;
  const float gd = has_depth ? gt_depth[r] : 0.0f;
  float nearv, farv;
  if (has_depth) {
    nearv = gd * 0.01f;
    farv = fminf(fmaxf(far_bb, 0.0f), gt_max * 1.2f);
    if (far_bb != far_bb) farv = far_bb;
  } else {
    nearv = 0.01f;
    farv = far_bb;
    nsurf = 0;
  }
  const int total = ns + nsurf;
  const float span = farv - nearv;
  auto zu = [&](int j) { return nearv + span * t_samples[j]; };
  // (descending runs -- far < near, or no-depth rays under a batch maximum below 0.001 -- are walked backwards: the
  // merge below then equals the reference's sort in those degenerate cases too; without near-surface samples the
  // reference does not sort, render.py:162, and the run is left as it is)
  const bool rev_a = span < 0.0f && nsurf > 0, rev_b = !(gd > 0.0f) && gt_max < 0.001f;
  float za = INFINITY, zb = INFINITY;
  if (lane < ns) {
    const int j = rev_a ? ns - 1 - lane : lane;
    float z = zu(j);
    if (perturb) {
      const float lo = (j == 0) ? z : 0.5f * (zu(j - 1) + z);
      const float hi = (j == ns - 1) ? z : 0.5f * (z + zu(j + 1));
      z = lo + (hi - lo) * perturb[j];
    }
    za = z;
  }
  if (lane < nsurf) {
    const float t = t_surface[rev_b ? nsurf - 1 - lane : lane];
    if (gd > 0.0f) {
      const float snr = (1.0f - 0.1f) * gd, sfar = (1.0f + 0.1f) * gd;
      zb = (snr + (sfar - snr) * t) * 1.0f + (0.001f + (gt_max - 0.001f) * t) * (1.0f - 1.0f);
    } else {
      const float vd = gd * 0.0f;
      const float snr = (1.0f - 0.1f) * vd, sfar = (1.0f + 0.1f) * vd;
      zb = (snr + (sfar - snr) * t) * 0.0f + (0.001f + (gt_max - 0.001f) * t) * (1.0f - 0.0f);
    }
  }
  sa[wave][lane] = za;
  sb[wave][lane] = zb;
  __syncthreads();
  int ra = 0, rb = 0;      // #surface < za ; #stratified <= zb
  for (int k = 0; k < nsurf; ++k) ra += (sb[wave][k] < za) ? 1 : 0;
  for (int k = 0; k < ns; ++k) rb += (sa[wave][k] <= zb) ? 1 : 0;
  if (lane < ns) sz[wave][lane + ra] = za;
  if (lane < nsurf) sz[wave][lane + rb] = zb;
  __syncthreads();
  float* zo = z_vals + (size_t)r * total;
  float* dd = dists + (size_t)r * total;
  for (int k = lane; k < total; k += 64) {
    const float z = sz[wave][k];
    zo[k] = z;
    dd[k] = (k + 1 < total) ? sz[wave][k + 1] - z : span / (float)ns;
  }
}

// ---------------------------------------------------------------------------------------
// hash grid
// ---------------------------------------------------------------------------------------
// One level: value (2 features) and d value / d x (3 x 2), tcnn accumulation order.
__device__ __forceinline__ void grid_level(const gs_grid_meta& m, int l, const _Float16* __restrict__ grid,
                                           const float x[3], float val[2], float dv[3][2], bool want_grad) {
  const float scale = m.scale[l];
  float f[3];
  uint32_t g[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const float pos = fmaf(scale, x[d], 0.5f);
    const float fl = floorf(pos);
    g[d] = (uint32_t)(int)fl;
    f[d] = pos - fl;
  }
  const _Float16* tab = grid + (size_t)m.offset[l] * 2;
  float v[8][2];
  uint32_t cidx[8];
  grid_corners(m, l, g, cidx);
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const uint32_t raw = *reinterpret_cast<const uint32_t*>(tab + (size_t)cidx[c] * 2);
    v[c][0] = (float)__builtin_bit_cast(_Float16, (uint16_t)(raw & 0xffffu));
    v[c][1] = (float)__builtin_bit_cast(_Float16, (uint16_t)(raw >> 16));
  }
  val[0] = 0.f; val[1] = 0.f;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    float w = 1.0f;
#pragma unroll
    for (int d = 0; d < 3; ++d) w = w * (((c >> d) & 1) ? f[d] : (1.0f - f[d]));
    val[0] = fmaf(w, v[c][0], val[0]);
    val[1] = fmaf(w, v[c][1], val[1]);
  }
  if (want_grad) {
#pragma unroll
    for (int gd = 0; gd < 3; ++gd) {
      const int o0 = (gd == 0) ? 1 : 0, o1 = (gd == 2) ? 1 : 2;   // the two other dims, ascending
      float a0 = 0.f, a1 = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float w = scale;
        w = w * ((k & 1) ? f[o0] : (1.0f - f[o0]));
        w = w * ((k & 2) ? f[o1] : (1.0f - f[o1]));
        const int cl = ((k & 1) << o0) | (((k >> 1) & 1) << o1);
        const int cr = cl | (1 << gd);
        a0 = fmaf(w, v[cr][0] - v[cl][0], a0);
        a1 = fmaf(w, v[cr][1] - v[cl][1], a1);
      }
      dv[gd][0] = a0;
      dv[gd][1] = a1;
    }
  }
}

__global__ __launch_bounds__(256) void grid_encode_kernel(const float* __restrict__ x,
                                                          const _Float16* __restrict__ grid,
                                                          _Float16* __restrict__ out, float* __restrict__ dy_dx,
                                                          int n, gs_grid_meta m) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float xi[3] = {x[i * 3 + 0], x[i * 3 + 1], x[i * 3 + 2]};
#pragma unroll 1
  for (int l = 0; l < GS_GRID_LEVELS; ++l) {
    float val[2], dv[3][2];
    grid_level(m, l, grid, xi, val, dv, dy_dx != nullptr);
    out[(size_t)i * 32 + 2 * l + 0] = (_Float16)val[0];
    out[(size_t)i * 32 + 2 * l + 1] = (_Float16)val[1];
    if (dy_dx) {
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int d = 0; d < 3; ++d) dy_dx[((size_t)i * 32 + 2 * l + f) * 3 + d] = dv[d][f];
    }
  }
}

// ---------------------------------------------------------------------------------------
// fused MLP on MFMA.  H^T[neuron][point] = W[neuron][k] X^T[k][point]
//   A operand (32 neurons x 16 k): lane l holds W[32*mt + (l&31)][16*ks + 8*(l>>5) .. +7]
//   B operand (16 k x 32 points) : lane l holds X[32*nt + (l&31)][16*ks + 8*(l>>5) .. +7]
//   C (32 neurons x 32 points)   : lane l, reg r -> neuron 32*mt + (r&3) + 8*(r>>2) + 4*(l>>5),
//                                  point 32*nt + (l&31)
// ---------------------------------------------------------------------------------------
constexpr int HS = 72;   // LDS row stride (halfs) of the wave-private activation tile [64][64]

__device__ __forceinline__ half8 ld8(const _Float16* p) { return *reinterpret_cast<const half8*>(p); }

template <bool RELU>
__device__ __forceinline__ void store_act(_Float16* hs, const float16v& c, int mt, int nt, int lane) {
  const int point = 32 * nt + (lane & 31);
#pragma unroll
  for (int q = 0; q < 4; ++q) {       // (round, then ReLU as a signed 16-bit max on the bit patterns: see store_act_xs)
    typedef float float4q __attribute__((ext_vector_type(4)));
    typedef short short4q __attribute__((ext_vector_type(4)));
    const float4q v4 = {c[q * 4], c[q * 4 + 1], c[q * 4 + 2], c[q * 4 + 3]};
    half4 pk = __builtin_convertvector(v4, half4);
    if (RELU) pk = __builtin_bit_cast(half4, __builtin_elementwise_max(__builtin_bit_cast(short4q, pk), short4q{0, 0, 0, 0}));
    *reinterpret_cast<half4*>(hs + point * HS + 32 * mt + 8 * q + 4 * (lane >> 5)) = pk;
  }
}


// The colour MLP of ONE wave's 64 points, rows in a wave-private LDS tile xs[64][XS] (fp16, 80 columns used):
// what neus_mlp_kernel does per tile, with the weights fetched per layer from L1 / L2 (10 KB, shared by every wave of
// the launch) instead of living in 88 registers for a whole launch, and the activations written back into the SAME
// tile (row stride XS).  Returns the 16 output rows x 2 x 32 points accumulators (neurons 0..3 = regs 0..3 of lanes 0..31).
constexpr int XS = 80;   // halves per row (160 B: four 40 KB workgroups fit a CU; 88 would make the b128 reads conflict-free but only three fit)

template <bool RELU>
__device__ __forceinline__ void store_act_xs(_Float16* xs, const float16v& c, int mt, int nt, int lane) {
  const int point = 32 * nt + (lane & 31);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    // round first, then ReLU as a signed 16-bit max with 0 on the bit patterns (rounding is monotone and keeps the sign:
    // the same values as max-then-round, -0 becomes +0): v_cvt_pk + v_pk_max_i16 per pair, where fmaxf on floats costs a
    // canonicalising v_max + the max per VALUE (257 v_max_f32 in this kernel's ISA)
    typedef float float4q __attribute__((ext_vector_type(4)));
    typedef short short4q __attribute__((ext_vector_type(4)));
    const float4q v4 = {c[q * 4], c[q * 4 + 1], c[q * 4 + 2], c[q * 4 + 3]};
    half4 pk = __builtin_convertvector(v4, half4);
    if (RELU) pk = __builtin_bit_cast(half4, __builtin_elementwise_max(__builtin_bit_cast(short4q, pk), short4q{0, 0, 0, 0}));
    *reinterpret_cast<half4*>(xs + point * XS + 32 * mt + 8 * q + 4 * (lane >> 5)) = pk;
  }
}

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ void wave_mlp64(_Float16* xs, const _Float16* __restrict__ w, int lane, float16v (&co)[2]) {
  const _Float16* W1 = w;                 // [64][80]
  const _Float16* W2 = w + 64 * 80;       // [64][64]
  const _Float16* W3 = w + 64 * 80 + 64 * 64;   // [16][64]
  const int r = lane & 31, kh = 8 * (lane >> 5);
  float16v c[2][2];
  // ---- layer 1
  {
    half8 a[2][5];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int ks = 0; ks < 5; ++ks) a[mt][ks] = ld8(W1 + (32 * mt + r) * 80 + 16 * ks + kh);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int e = 0; e < 16; ++e) c[mt][nt][e] = 0.f;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int ks = 0; ks < 5; ++ks) {
        const half8 b = ld8(xs + (32 * nt + r) * XS + 16 * ks + kh);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) c[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mt][ks], b, c[mt][nt], 0, 0, 0);
      }
  }
  wave_lds_sync();                        // every lane has read its input rows: the tile now takes the activations
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) store_act_xs<true>(xs, c[mt][nt], mt, nt, lane);
  wave_lds_sync();
  // ---- layer 2
  {
    half8 a[2][4], bfr[2][4];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) a[mt][ks] = ld8(W2 + (32 * mt + r) * 64 + 16 * ks + kh);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) bfr[nt][ks] = ld8(xs + (32 * nt + r) * XS + 16 * ks + kh);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int e = 0; e < 16; ++e) c[mt][nt][e] = 0.f;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
          c[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mt][ks], bfr[nt][ks], c[mt][nt], 0, 0, 0);
  }
  wave_lds_sync();
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) store_act_xs<true>(xs, c[mt][nt], mt, nt, lane);
  wave_lds_sync();
  // ---- output layer (16 rows, padded to the 32-row tile with zero weights)
  {
    half8 a3[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
      a3[ks] = (r < 16) ? ld8(W3 + r * 64 + 16 * ks + kh) : z;
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
      for (int e = 0; e < 16; ++e) co[nt][e] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const half8 b = ld8(xs + (32 * nt + r) * XS + 16 * ks + kh);
        co[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a3[ks], b, co[nt], 0, 0, 0);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// per-point stage
// ---------------------------------------------------------------------------------------
struct NeusArgs {
  const float* rays_o; const float* rays_d; const float* z_vals; const float* dists;
  const _Float16* grid; const float* sdf_w; const float* sdf_b; const float* color_B;
  float inv_s; const float* inv_s_dev;     // inv_s_dev != nullptr: read the scalar from device memory instead
  float bound[6]; float rt_bound[6];
  const float* rt_bound_dev;               // != nullptr: the realtime bound lives in device memory (replayed graphs see
                                           // InstantNeuS.update_bound without a re-capture)
  int n, s;
};

__device__ __forceinline__ bool point_of(const NeusArgs& A, int idx, float pt[3], float dir[3], float& zm, float& dist) {
  const int ray = idx / A.s;
  const float z = A.z_vals[idx];
  dist = A.dists[idx];
  zm = z + dist / 2.0f;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    dir[d] = A.rays_d[ray * 3 + d];
    pt[d] = A.rays_o[ray * 3 + d] + dir[d] * zm;
  }
  float rb[6];
  if (A.rt_bound_dev) {   // uniform, unchanged during the launch: constant address space -> s_load
    typedef const __attribute__((address_space(4))) float* cfp;
    cfp q = (cfp)(uintptr_t)A.rt_bound_dev;
#pragma unroll
    for (int k = 0; k < 6; ++k) rb[k] = q[k];
  } else {
#pragma unroll
    for (int k = 0; k < 6; ++k) rb[k] = A.rt_bound[k];
  }
  return (pt[0] < rb[1]) && (pt[0] > rb[0]) && (pt[1] < rb[3]) && (pt[1] > rb[2]) && (pt[2] < rb[5]) && (pt[2] > rb[4]);
}

// normalized_3d_coordinate (InstantNeuS.py:12-32) with the STATIC bound, clamped to [-1,1]; `view` = the grid's [0,1] input
__device__ __forceinline__ void normalise_point(const NeusArgs& A, const float pt[3], float p[3], float view[3],
                                                float inside[3], float span[3]) {
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    span[d] = A.bound[2 * d + 1] - A.bound[2 * d];
    float q = (pt[d] - A.bound[2 * d]) / span[d] * 2.0f - 1.0f;
    inside[d] = (q >= -1.0f && q <= 1.0f) ? 1.0f : 0.0f;
    q = fminf(fmaxf(q, -1.0f), 1.0f);
    p[d] = q;
    view[d] = (q + 1.0f) / 2.0f;
  }
}

// ---------------------------------------------------------------------------------------
// LEVEL-MAJOR encode of the hashed levels (round 5).  The 11 hashed levels are 2 MB tables each (2^19 entries x 4 B);
// a point-major kernel walks all 16 levels per point, so every XCD streams the whole 25 MB table through its 4 MB L2
// and half of the gathers miss (profiles/r04_pmc_neus.json: 57.5 M line misses per 32768-ray launch, 64 B moved for 4-8
// useful: 3.1x the algorithmic traffic).  Here a work item is (hashed level, 256-point chunk) and the launch order makes
// EVERY XCD work through its own eighth of the chunks ONE LEVEL AT A TIME (block b runs on XCD b % 8 -- observed, used for
// speed only: any placement gives the same records): the level's 2 MB stay in that XCD's L2 while its chunks pass, and
// the misses shrink to the compulsory 2 MB per level and XCD.  Per (level, in-bound point) the kernel leaves ONE 16-byte
// record [enc0, enc1 (fp16) | this level's contribution to d sdf / d view (3 x fp32)], written with non-temporal stores so
// that the stream does not evict the table; neus_point_kernel consumes the records in level order with the same
// arithmetic as before (the SDF linear layer is evaluated from the same fp16 encodings in the same order: sdf is
// bit-identical; the gradient sums one rounded fma chain per level instead of one chain over all levels).
// Training: the backward's record [enc0, enc1, d enc / dx (6)] fp16 (enc_aux) is written here as well.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void neus_encode_levels_kernel(NeusArgs A, gs_grid_meta m, u32x4* __restrict__ rec,
                                                                 _Float16* __restrict__ enc_aux, int first_hashed,
                                                                 int chunks_per_xcd) {
  const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int g = j / chunks_per_xcd, ci = j - g * chunks_per_xcd;
  const int l = first_hashed + g;
  const int idx = (ci * 8 + x) * 256 + threadIdx.x;
  const int np = A.n * A.s;
  if (idx >= np) return;
  float pt[3], dir[3], zm, dist;
  if (!point_of(A, idx, pt, dir, zm, dist)) return;      // (out-of-bound points have no record: nobody reads one)
  float p[3], view[3], inside[3], span[3];
  normalise_point(A, pt, p, view, inside, span);
  float val[2], dv[3][2];
  grid_level(m, l, A.grid, view, val, dv, true);
  typedef const __attribute__((address_space(4))) float* cfp;    // uniform, unchanged during the launch -> s_load
  cfp wl = (cfp)(uintptr_t)(A.sdf_w + 3 + 2 * l);
  const float g0 = (float)(_Float16)wl[0], g1 = (float)(_Float16)wl[1];     // dL/d enc arrives as fp16
  u32x4 r;
  r.x = (uint32_t)__builtin_bit_cast(unsigned short, (_Float16)val[0]) |
        ((uint32_t)__builtin_bit_cast(unsigned short, (_Float16)val[1]) << 16);
  r.y = __float_as_uint(fmaf(g1, dv[0][1], g0 * dv[0][0]));
  r.z = __float_as_uint(fmaf(g1, dv[1][1], g0 * dv[1][0]));
  r.w = __float_as_uint(fmaf(g1, dv[2][1], g0 * dv[2][0]));
  __builtin_nontemporal_store(r, rec + (size_t)g * (size_t)np + idx);
  if (enc_aux) {
    half8 a;
    a[0] = (_Float16)val[0]; a[1] = (_Float16)val[1];
    a[2] = (_Float16)dv[0][0]; a[3] = (_Float16)dv[1][0]; a[4] = (_Float16)dv[2][0];
    a[5] = (_Float16)dv[0][1]; a[6] = (_Float16)dv[1][1]; a[7] = (_Float16)dv[2][1];
    __builtin_nontemporal_store(a, reinterpret_cast<half8*>(enc_aux + ((size_t)l * (size_t)np + idx) * 8));
  }
}

// Since round 4 the colour MLP runs in this kernel's tail (wave_mlp64): the 80-wide input row of every point goes into a
// wave-private LDS tile and the wave evaluates its 64 points on the matrix cores -- at render time the rows never reach
// HBM (47 MB written + read per 4096-ray batch before) and a launch is gone; the training path still saves them
// (`mlp_in` != nullptr) for gs_mlp_backward.  All 64 lanes stay until the end (the MFMAs are wave-wide): out-of-bound
// and past-the-end lanes contribute zero rows.
// `flags` (one byte per wave, every one written by the main pass -- nothing to zero beforehand): whether ANY of the wave's
// points lies in the realtime bound; the reference forces the first 100 points valid when none does
// (InstantNeuS.py:311-312), which a second launch of this kernel with force_pass = 1 and ONE workgroup handles: it ORs
// the flags and returns at once unless all are 0 (a 3 us launch instead of the full-size neus_count_kernel pass over
// all points that used to precede the main pass).
// 4 waves per SIMD (amdgpu_waves_per_eu): 124 VGPRs without scratch, 4 x 40 KB workgroups = a CU's LDS exactly.  Left to
// itself the compiler takes 188 VGPRs (2 waves per SIMD, more gathers in flight per wave) -- measured on one box: 156 vs
// 130 us for the 4096-ray batch (4608 waves: 2.25 rounds of 2048 resident waves vs 1.125 of 4096), render 19.4 -> 22.1
// M rays/s with everything else equal.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void neus_point_kernel(
    NeusArgs A, gs_grid_meta m, uint8_t* __restrict__ flags, int force_pass, float* __restrict__ sdf_out,
    float* __restrict__ zmid_out, float* __restrict__ alpha_out, float* __restrict__ grad_out,
    uint8_t* __restrict__ mask_out, _Float16* __restrict__ mlp_in, _Float16* __restrict__ enc_aux,
    const _Float16* __restrict__ mlp_w, _Float16* __restrict__ rgb_out, const u32x4* __restrict__ rec, int first_hashed) {
  // `rec` != nullptr: the hashed levels (l >= first_hashed) were encoded level-major by neus_encode_levels_kernel -- this
  // kernel streams their 16-byte records and gathers only the dense levels (2 MB in all, L2-resident, a wave's 64
  // consecutive samples of a ray share most cells).  rec == nullptr (the one-workgroup force pass, whose points were out of
  // bound and have no records): every level is gathered here.
  __shared__ __attribute__((aligned(16))) _Float16 xs_all[4 * 64 * XS];
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const int np = A.n * A.s;
  const bool valid = idx < np;
  const int lane = threadIdx.x & 63;
  _Float16* xs = xs_all + (threadIdx.x >> 6) * 64 * XS;
  half8* xrow = reinterpret_cast<half8*>(xs + lane * XS);
  float pt[3], dir[3], zm = 0.f, dist = 0.f;
  bool in = false;
  if (force_pass) {                                 // one workgroup: did any wave of the main pass see a point in bound?
    const int nw = (np + 63) >> 6;
    int any = 0;                                    // (every wave scans all flags itself: no LDS, no barrier)
    for (int i0 = lane * 16; i0 < nw; i0 += 8 * 1024) {    // eight 16-byte loads in flight (36 dependent trips at 32768
      uint4 v[8];                                          // rays otherwise: 20 us for a pass that has nothing to do)
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + 1024 * u;
        v[u] = i + 16 <= nw ? *reinterpret_cast<const uint4*>(flags + i) : make_uint4(0u, 0u, 0u, 0u);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + 1024 * u;
        any |= (v[u].x | v[u].y | v[u].z | v[u].w) != 0u;
        if (i < nw && i + 16 > nw)
          for (int k = i; k < nw; ++k) any |= flags[k];
      }
    }
    if (__ballot(any != 0) != 0ull) return;         // (uniform) some point was in bound: nothing to force
  }
  if (valid) {
    in = point_of(A, idx, pt, dir, zm, dist);
    if (force_pass && idx < 100) in = true;         // InstantNeuS.py:311-312
    zmid_out[idx] = zm;
    mask_out[idx] = in ? 1 : 0;
  }
  {
    const unsigned long long anyin = __ballot(in);
    if (!force_pass && lane == 0) flags[idx >> 6] = anyin != 0ull ? 1 : 0;        // every wave writes its own slot
  }
  if (!in) {                                        // sdf = 100, grad = feat = rgb = 0, alpha * mask = 0
    const half8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
    if (valid) {
      sdf_out[idx] = 100.0f;
      alpha_out[idx] = 0.0f;
      grad_out[idx * 3 + 0] = 0.f; grad_out[idx * 3 + 1] = 0.f; grad_out[idx * 3 + 2] = 0.f;
      if (mlp_in) {   // keep the saved MLP input row finite: the training path runs GEMMs over ALL rows (0 * NaN = NaN)
        half8* dst = reinterpret_cast<half8*>(mlp_in + (size_t)idx * 80);
#pragma unroll
        for (int q = 0; q < 10; ++q) dst[q] = zero;
      }
    }
#pragma unroll
    for (int q = 0; q < 10; ++q) xrow[q] = zero;
  } else {
    float p[3], view[3], inside[3], span[3];
    normalise_point(A, pt, p, view, inside, span);
    // Linear(35 -> 32): xyz part first, then one level (2 inputs) at a time
    float out[32];
  #pragma unroll
    for (int o = 0; o < 32; ++o) {
      float acc = 0.f;
  #pragma unroll
      for (int d = 0; d < 3; ++d) acc = fmaf(A.sdf_w[o * 35 + d], p[d], acc);
      out[o] = acc;
    }
    float gview[3] = {0.f, 0.f, 0.f};
    // levels gathered here: all of them, or the dense ones in front of the level-major records
    const int l_rec = rec ? first_hashed : GS_GRID_LEVELS;
  #pragma unroll 1
    for (int l = 0; l < l_rec; ++l) {
      const float* wl = A.sdf_w + 3 + 2 * l;
      float val[2], dv[3][2];
      grid_level(m, l, A.grid, view, val, dv, true);
      const float e0 = (float)(_Float16)val[0], e1 = (float)(_Float16)val[1];   // encoding output is fp16
      if (enc_aux) {  // training: what the backward needs of this level's 8 corner values -- the encoding and d enc / d x --
                      // as one 16-byte record [level][point][8], so that it streams them instead of gathering again
        half8 a;
        a[0] = (_Float16)val[0]; a[1] = (_Float16)val[1];
        a[2] = (_Float16)dv[0][0]; a[3] = (_Float16)dv[1][0]; a[4] = (_Float16)dv[2][0];
        a[5] = (_Float16)dv[0][1]; a[6] = (_Float16)dv[1][1]; a[7] = (_Float16)dv[2][1];
        *reinterpret_cast<half8*>(enc_aux + ((size_t)l * (size_t)np + idx) * 8) = a;
      }
      const float g0 = (float)(_Float16)wl[0], g1 = (float)(_Float16)wl[1];     // dL/d enc arrives as fp16
  #pragma unroll
      for (int d = 0; d < 3; ++d) gview[d] += fmaf(g1, dv[d][1], g0 * dv[d][0]);        // (as the level-major records)
  #pragma unroll
      for (int o = 0; o < 32; ++o) out[o] = fmaf(wl[o * 35 + 1], e1, fmaf(wl[o * 35], e0, out[o]));
    }
    if (rec) {
      // encoded level-major: one coalesced 16-byte record per point and hashed level.  The record of level l + 1 is
      // REQUESTED before level l's 64 multiply-adds (two register sets, the loop body twice: a rotation through copies
      // would wait for the load it has just issued): consumed where it is loaded, every level was one exposed memory
      // round trip -- eleven per wave at 4 waves per SIMD (round 5's counters: this kernel's waves waiting 48 %).
      const u32x4* rp = rec + idx;
      auto request = [&](int l) {      // (past the last level: the last level again, never consumed)
        return __builtin_nontemporal_load(rp + (size_t)(min(l, GS_GRID_LEVELS - 1) - first_hashed) * (size_t)np);
      };
      auto consume = [&](const u32x4& r, int l) {
        const float* wl = A.sdf_w + 3 + 2 * l;
        const float e0 = (float)__builtin_bit_cast(_Float16, (unsigned short)(r.x & 0xffffu));
        const float e1 = (float)__builtin_bit_cast(_Float16, (unsigned short)(r.x >> 16));
        gview[0] += __uint_as_float(r.y); gview[1] += __uint_as_float(r.z); gview[2] += __uint_as_float(r.w);
  #pragma unroll
        for (int o = 0; o < 32; ++o) out[o] = fmaf(wl[o * 35 + 1], e1, fmaf(wl[o * 35], e0, out[o]));
      };
      int l = first_hashed;
      u32x4 ra = request(l), rb;
      if ((GS_GRID_LEVELS - l) & 1) {   // an odd count: one level in front, so that the pairs below have no branch inside
        rb = request(l + 1);            // (behind a branch the compiler sinks a request down to its first use)
        consume(ra, l);
        ra = rb;
        ++l;
      }
  #pragma unroll 1
      for (; l < GS_GRID_LEVELS; l += 2) {
        rb = request(l + 1);
        consume(ra, l);
        ra = request(l + 2);
        consume(rb, l + 1);
      }
    }
  #pragma unroll
    for (int o = 0; o < 32; ++o) out[o] = out[o] + A.sdf_b[o];
    const float sdf = out[0];
    float grad[3];
  #pragma unroll
    for (int d = 0; d < 3; ++d) grad[d] = (A.sdf_w[d] + gview[d] / 2.0f) * inside[d] * 2.0f / span[d];
    // NeuS alpha (InstantNeuS.py:276-293), cos_anneal_ratio = 1
    const float true_cos = (dir[0] * grad[0] + dir[1] * grad[1]) + dir[2] * grad[2];
    const float iter_cos = -(fmaxf(-true_cos * 0.5f + 0.5f, 0.0f) * 0.0f + fmaxf(-true_cos, 0.0f) * 1.0f);
    const float est_next = sdf + iter_cos * dist / 2.0f;
    const float est_prev = sdf - iter_cos * dist / 2.0f;
    const float inv_s_ = A.inv_s_dev ? *A.inv_s_dev : A.inv_s;
    const float prev_cdf = 1.0f / (1.0f + expf(-(est_prev * inv_s_)));
    const float next_cdf = 1.0f / (1.0f + expf(-(est_next * inv_s_)));
    float alpha = (prev_cdf - next_cdf + 1e-5f) / (prev_cdf + 1e-5f);
    alpha = fminf(fmaxf(alpha, 0.0f), 1.0f);
    sdf_out[idx] = sdf;
    alpha_out[idx] = alpha;
    grad_out[idx * 3 + 0] = grad[0]; grad_out[idx * 3 + 1] = grad[1]; grad_out[idx * 3 + 2] = grad[2];
    // colour-MLP input row: sin(pts @ B) (33) | normals (3) | feat (31) | ones (13), emitted eight
    // columns (16 B) at a time so the row never sits in registers as a whole
    half8* dst = mlp_in ? reinterpret_cast<half8*>(mlp_in + (size_t)idx * 80) : nullptr;
#pragma unroll
    for (int q = 0; q < 10; ++q) {
      half8 pk;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int c = q * 8 + k;
        float v;
        if (c < 33) v = emb_sin((pt[0] * A.color_B[c] + pt[1] * A.color_B[33 + c]) + pt[2] * A.color_B[66 + c]);
        else if (c < 36) v = grad[c - 33];
        else if (c < 67) v = out[1 + (c - 36)];
        else v = 1.0f;
        pk[k] = (_Float16)v;
      }
      if (dst) dst[q] = pk;
      xrow[q] = pk;
    }
  }
  // ---- the colour MLP of this wave's 64 points (tcnn FullyFusedMLP 67(->80)->64->64->3, ReLU, then sigmoid)
  wave_lds_sync();
  float16v co[2];
  wave_mlp64(xs, mlp_w, lane, co);
  const int wave_p0 = blockIdx.x * 256 + (threadIdx.x >> 6) * 64;
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int on = __shfl((int)in, (32 * nt + lane) & 63, 64);
    const int pnt = wave_p0 + 32 * nt + lane;
    if (lane < 32 && pnt < np) {
#pragma unroll
      for (int o = 0; o < 3; ++o) {
        float v = (float)(_Float16)co[nt][o];      // network output is fp16
        v = 1.0f / (1.0f + expf(-v));
        rgb_out[(size_t)pnt * 3 + o] = on ? (_Float16)v : (_Float16)0.0f;
      }
    }
  }
}

__global__ __launch_bounds__(256) void neus_mlp_kernel(const _Float16* __restrict__ x, int ld_x,
                                                       const _Float16* __restrict__ w, const uint8_t* __restrict__ mask,
                                                       _Float16* __restrict__ out, int n_out, int ld_out, int np,
                                                       int apply_sigmoid, int n_tiles) {
  __shared__ _Float16 lds[4][64 * HS];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  _Float16* hs = lds[wave];
  const _Float16* W1 = w;                 // [64][80]
  const _Float16* W2 = w + 64 * 80;       // [64][64]
  const _Float16* W3 = w + 64 * 80 + 64 * 64;   // [16][64]
  const int r = lane & 31, kh = 8 * (lane >> 5);
  // weights live in registers for the whole launch
  half8 a1[2][5], a2[2][4], a3[4];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) a1[mt][ks] = ld8(W1 + (32 * mt + r) * 80 + 16 * ks + kh);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) a2[mt][ks] = ld8(W2 + (32 * mt + r) * 64 + 16 * ks + kh);
  }
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    a3[ks] = (r < 16) ? ld8(W3 + r * 64 + 16 * ks + kh) : z;
  }
  // every wave of the workgroup runs the same number of iterations (barriers below); tiles past
  // the end are computed on clamped rows and never stored
  for (int t0 = blockIdx.x * 4; t0 < n_tiles; t0 += gridDim.x * 4) {
    const int tile = t0 + wave;
    const int p0 = tile * 64;
    // ---- layer 1: B fragments straight from the [point][80] rows
    float16v c[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int e = 0; e < 16; ++e) c[mt][nt][e] = 0.f;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int pnt = max(0, min(p0 + 32 * nt + r, np - 1));
      const _Float16* xr = x + (size_t)pnt * ld_x + kh;
#pragma unroll
      for (int ks = 0; ks < 5; ++ks) {
        const half8 b = ld8(xr + 16 * ks);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
          c[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[mt][ks], b, c[mt][nt], 0, 0, 0);
      }
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) store_act<true>(hs, c[mt][nt], mt, nt, lane);
    __syncthreads();
    // ---- layer 2
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int e = 0; e < 16; ++e) c[mt][nt][e] = 0.f;
    half8 bfr[2][4];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) bfr[nt][ks] = ld8(hs + (32 * nt + r) * HS + 16 * ks + kh);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
          c[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2[mt][ks], bfr[nt][ks], c[mt][nt], 0, 0, 0);
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) store_act<true>(hs, c[mt][nt], mt, nt, lane);
    __syncthreads();
    // ---- output layer (16 rows, padded to the 32-row tile with zero weights)
    float16v co[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
      for (int e = 0; e < 16; ++e) co[nt][e] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const half8 b = ld8(hs + (32 * nt + r) * HS + 16 * ks + kh);
        co[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a3[ks], b, co[nt], 0, 0, 0);
      }
    }
    __syncthreads();
    // neurons 0..3 sit in regs 0..3 of lanes 0..31
    if (lane < 32) {
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const int pnt = p0 + 32 * nt + lane;
        if (pnt < np) {
          const bool on = mask ? (mask[pnt] != 0) : true;
          for (int o = 0; o < n_out; ++o) {
            float v = (float)(_Float16)co[nt][o];          // network output is fp16
            if (apply_sigmoid) v = 1.0f / (1.0f + expf(-v));
            out[(size_t)pnt * ld_out + o] = on ? (_Float16)v : (_Float16)0.0f;
          }
        }
      }
    }
  }
}

// generic-width input (tcnn.Network API): pad [n, n_in] fp16 rows to 80 with ones
__global__ __launch_bounds__(256) void mlp_pad_kernel(const _Float16* __restrict__ x, _Float16* __restrict__ xp,
                                                      int n, int n_in) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)n * 80) return;
  const int c = (int)(i % 80);
  const size_t row = i / 80;
  xp[i] = (c < n_in) ? x[row * n_in + c] : (_Float16)1.0f;
}

// ---------------------------------------------------------------------------------------
// per-ray compositing (InstantNeuS.py:343-358)
// ---------------------------------------------------------------------------------------
// One wave per ray: lanes own samples (64 per pass), the exclusive transmittance product is a
// wave prefix scan, the per-ray sums are wave reductions; every load is coalesced.
__device__ __forceinline__ float wave_incl_prod(float v, int lane) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const float u = __shfl_up(v, off, 64);
    if (lane >= off) v = v * u;
  }
  return v;
}

__global__ __launch_bounds__(256) void neus_ray_kernel(const float* __restrict__ alpha, const _Float16* __restrict__ rgb,
                                                       const float* __restrict__ zmid, const float* __restrict__ grad,
                                                       const uint8_t* __restrict__ mask, float* __restrict__ color,
                                                       float* __restrict__ depth, float* __restrict__ depth_var,
                                                       float* __restrict__ normal, float* __restrict__ weight_sum,
                                                       float* __restrict__ gerr, float gerr_scale,
                                                       float* __restrict__ sdf_var_out, float sdf_var_value,
                                                       int n, int s) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n) return;
  const size_t b = (size_t)r * s;
  float wsum = 0.f, col[3] = {0.f, 0.f, 0.f}, dep = 0.f, nrm[3] = {0.f, 0.f, 0.f}, ge = 0.f;
  float Trun = 1.0f;
  for (int base = 0; base < s; base += 64) {
    const int k = base + lane;
    const bool on = k < s;
    const float a = on ? alpha[b + k] : 0.0f;
    const float t = on ? (1.0f - a + 1e-7f) : 1.0f;
    const float incl = wave_incl_prod(t, lane);
    float excl = __shfl_up(incl, 1, 64);
    if (lane == 0) excl = 1.0f;
    const float w = a * (Trun * excl);
    Trun = Trun * __shfl(incl, 63, 64);
    if (on) {
      wsum += w;
      dep += zmid[b + k] * w;
      const float mk = mask[b + k] ? 1.0f : 0.0f;
      const float gx = grad[(b + k) * 3 + 0], gy = grad[(b + k) * 3 + 1], gz = grad[(b + k) * 3 + 2];
      col[0] += (float)rgb[(b + k) * 3 + 0] * w;
      col[1] += (float)rgb[(b + k) * 3 + 1] * w;
      col[2] += (float)rgb[(b + k) * 3 + 2] * w;
      nrm[0] += gx * w * mk; nrm[1] += gy * w * mk; nrm[2] += gz * w * mk;
      const float nr = sqrtf((gx * gx + gy * gy) + gz * gz) - 1.0f;
      ge += nr * nr * mk;
    }
  }
  wsum = gs_wave_sum(wsum);
  dep = gs_wave_sum(dep);
  ge = gs_wave_sum(ge);
#pragma unroll
  for (int d = 0; d < 3; ++d) { col[d] = gs_wave_sum(col[d]); nrm[d] = gs_wave_sum(nrm[d]); }
  // variance needs the final depth: redo the (cheap) scan
  float var = 0.f;
  Trun = 1.0f;
  for (int base = 0; base < s; base += 64) {
    const int k = base + lane;
    const bool on = k < s;
    const float a = on ? alpha[b + k] : 0.0f;
    const float t = on ? (1.0f - a + 1e-7f) : 1.0f;
    const float incl = wave_incl_prod(t, lane);
    float excl = __shfl_up(incl, 1, 64);
    if (lane == 0) excl = 1.0f;
    const float w = a * (Trun * excl);
    Trun = Trun * __shfl(incl, 63, 64);
    if (on) {
      const float dz = zmid[b + k] - dep;
      var += dz * dz * w;
    }
  }
  var = gs_wave_sum(var);
  if (lane == 0) {
    color[r * 3 + 0] = col[0]; color[r * 3 + 1] = col[1]; color[r * 3 + 2] = col[2];
    normal[r * 3 + 0] = nrm[0]; normal[r * 3 + 1] = nrm[1]; normal[r * 3 + 2] = nrm[2];
    depth[r] = dep;
    depth_var[r] = var;
    weight_sum[r] = wsum;
    gerr[r] = ge * gerr_scale;
    if (sdf_var_out) sdf_var_out[r] = sdf_var_value;
  }
}

gs_grid_meta host_meta() {
  gs_grid_meta m;
  gs_grid_meta_default(&m);
  return m;
}

}  // namespace

extern "C" int gs_grid_meta_default(gs_grid_meta* m) {
  if (!m) return GS_ERR_INVALID_ARG;
  // tcnn grid.h: grid_scale / grid_resolution / params_in_level in fp32
  const float log2_s = log2f(1.447269237440378f);
  uint32_t total = 0;
  for (int l = 0; l < GS_GRID_LEVELS; ++l) {
    const float scale = exp2f((float)l * log2_s) * 16.0f - 1.0f;
    const uint32_t res = (uint32_t)ceilf(scale) + 1u;
    const double dense = (double)res * res * res;
    uint64_t n = dense > 4294967295.0 ? 4294967295ull : (uint64_t)dense;
    n = (n + 7) / 8 * 8;
    if (n > (1u << 19)) n = 1u << 19;
    m->scale[l] = scale;
    m->resolution[l] = res;
    m->size[l] = (uint32_t)n;
    m->offset[l] = total;
    m->hashed[l] = dense > (double)n ? 1u : 0u;
    total += (uint32_t)n;
  }
  m->total = total;
  return GS_OK;
}

extern "C" int gs_render_sample(const float* rays_o, const float* rays_d, const float* gt_depth, const float* bound,
                                const float* t_samples, const float* t_surface, const float* perturb, float gt_max,
                                const float* gt_max_dev, float* z_vals, float* dists, int n, int n_samples,
                                int n_surface, gs_stream_t stream) {
  GS_REQUIRE(rays_o && rays_d && bound && t_samples && z_vals && dists, "render_sample: null pointer");
  GS_REQUIRE(n >= 0 && n_samples > 0 && n_surface >= 0, "render_sample: bad shape");
  GS_REQUIRE(n_surface == 0 || t_surface, "render_sample: t_surface required");
  GS_REQUIRE(gt_max_dev || !gt_depth || gt_max != -INFINITY || (n_samples <= 64 && n_surface <= 64 && (((size_t)gt_depth) & 15) == 0),
             "render_sample: the in-launch batch maximum needs n_samples, n_surface <= 64 and a 16-byte aligned gt_depth");
  if (n == 0) return GS_OK;
  GS_TIMING_PRE();
  if (n_samples <= 64 && n_surface <= 64)
    render_sample_wave_kernel<<<gs_cdiv(n, 4), 256, 0, (hipStream_t)stream>>>(
        rays_o, rays_d, gt_depth, bound, t_samples, t_surface, perturb, gt_max, gt_max_dev, z_vals, dists, n, n_samples,
        gt_depth ? n_surface : 0);
  else
    render_sample_kernel<<<gs_cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(
        rays_o, rays_d, gt_depth, bound, t_samples, t_surface, perturb, gt_max, gt_max_dev, z_vals, dists, n, n_samples,
        gt_depth ? n_surface : 0);
  GS_CHECK_LAUNCH("render_sample");
  return GS_OK;
}

extern "C" int gs_grid_encode(const float* x, const void* grid, void* out, float* dy_dx, int n, gs_stream_t stream) {
  GS_REQUIRE(x && grid && out, "grid_encode: null pointer");
  GS_REQUIRE(n >= 0, "grid_encode: bad n");
  if (n == 0) return GS_OK;
  grid_encode_kernel<<<gs_cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(x, (const _Float16*)grid, (_Float16*)out, dy_dx,
                                                                       n, host_meta());
  GS_CHECK_LAUNCH("grid_encode");
  return GS_OK;
}

extern "C" size_t gs_mlp_workspace_bytes(int n, int n_in) { return n_in == 80 ? 0 : gs_align((size_t)n * 80 * 2) + 256; }

extern "C" int gs_mlp_forward(const void* x, const void* mlp, void* out, int n, int n_in, int n_out,
                              void* workspace, size_t workspace_bytes, gs_stream_t stream) {
  GS_REQUIRE(x && mlp && out, "mlp_forward: null pointer");
  GS_REQUIRE(n >= 0 && n_in > 0 && n_in <= 80 && n_out > 0 && n_out <= 4,
             "mlp_forward: only the InstantNeuS shape (<=80 inputs, 64x64 hidden, <=4 used outputs) is built");
  if (n == 0) return GS_OK;
  hipStream_t st = (hipStream_t)stream;
  const _Float16* xp = (const _Float16*)x;
  if (n_in != 80) {
    const size_t need = gs_mlp_workspace_bytes(n, n_in);
    if (!workspace || workspace_bytes < need) {
      gs_set_error("mlp_forward: workspace too small (%zu < %zu)", workspace_bytes, need);
      return GS_ERR_WORKSPACE;
    }
    _Float16* pad = (_Float16*)gs_align((size_t)workspace);
    mlp_pad_kernel<<<(unsigned)(((size_t)n * 80 + 255) / 256), 256, 0, st>>>((const _Float16*)x, pad, n, n_in);
    GS_CHECK_LAUNCH("mlp_pad");
    xp = pad;
  }
  const int n_tiles = gs_cdiv(n, 64);
  const int grid = n_tiles < 4 ? 1 : (gs_cdiv(n_tiles, 4) < 1024 ? gs_cdiv(n_tiles, 4) : 1024);
  neus_mlp_kernel<<<grid, 256, 0, st>>>(xp, 80, (const _Float16*)mlp, nullptr, (_Float16*)out, n_out, n_out, n, 0,
                                        n_tiles);
  GS_CHECK_LAUNCH("mlp_forward");
  return GS_OK;
}

namespace {
struct NeusWs {
  uint8_t* flags; float* alpha; float* grad; uint8_t* mask; _Float16* mlp_in; _Float16* rgb; u32x4* rec; size_t total;
};
// `rec_levels`: hashed levels whose level-major records get room (0: the point kernel gathers every level itself)
NeusWs carve_neus(void* base, int n, int s, int rec_levels) {
  NeusWs w;
  size_t off = 0;
  const size_t np = (size_t)n * s;
  auto take = [&](size_t bytes) { size_t o = off; off += gs_align(bytes); return (char*)base + o; };
  w.flags = (uint8_t*)take((np + 63) / 64 + 64);
  w.alpha = (float*)take(np * 4);
  w.grad = (float*)take(np * 12);
  w.mask = (uint8_t*)take(np);
  w.mlp_in = (_Float16*)take(np * 80 * 2);
  w.rgb = (_Float16*)take(np * 3 * 2 + 64);
  w.rec = rec_levels > 0 ? (u32x4*)take(np * 16 * (size_t)rec_levels) : nullptr;   // 16 B per point and hashed level
  w.total = off;
  return w;
}
}  // namespace

// crossover of the two gather orders, in sample points (gs_neus_level_major_min_points sets it: tests run both orders on
// small batches, tools measure the crossover)
static std::atomic<int> g_level_major_min_points{512 * 1024};
static int level_major_min_points() { return g_level_major_min_points.load(std::memory_order_relaxed); }
extern "C" int gs_neus_level_major_min_points(int points) {
  const int old = g_level_major_min_points.load(std::memory_order_relaxed);
  if (points >= 0) g_level_major_min_points.store(points, std::memory_order_relaxed);
  return old;
}

static int hashed_levels(const gs_grid_meta& meta, int* first_hashed) {
  int first = GS_GRID_LEVELS;
  for (int l = GS_GRID_LEVELS - 1; l >= 0 && meta.hashed[l]; --l) first = l;      // (the hashed levels are the finest)
  if (first_hashed) *first_hashed = first;
  return GS_GRID_LEVELS - first;
}

// Level-major records (16 B per point and HASHED level: 11 of the 16 levels here) are only reserved for batches that use
// them (>= gs_neus_level_major_min_points sample points): 176 B per point = 415 MB at 32768 x 72, nothing for the mapper's
// own 4096-ray batches (round 5 reserved 256 B per point for every batch: 75 MB at 4096 rays).
extern "C" size_t gs_neus_forward_workspace_bytes(int n, int s) {
  if (n < 0 || s < 0) return 0;
  const bool lm = (long long)n * s >= level_major_min_points();
  return carve_neus(nullptr, n, s, lm ? hashed_levels(host_meta(), nullptr) : 0).total + 256;
}

extern "C" int gs_neus_forward(const float* rays_o, const float* rays_d, const float* z_vals, const float* dists,
                               const void* grid, const float* sdf_w, const float* sdf_b, const float* color_B,
                               const void* mlp, float inv_s, const float* inv_s_dev, const float* bound_host,
                               const float* rt_bound_host, const float* rt_bound_dev,
                               float* color, float* depth, float* depth_var, float* normal, float* weight_sum,
                               float* sdf, float* z_mid, float* grad_err_ray, float* alpha_out, void* rgb_out,
                               float* grad_out, uint8_t* mask_out, void* mlp_in_out, void* enc_aux_out, float grad_err_scale,
                               float* sdf_variance_out, float sdf_variance_value, int n, int s,
                               void* workspace, size_t workspace_bytes, gs_stream_t stream) {
  GS_REQUIRE(rays_o && rays_d && z_vals && dists && grid && sdf_w && sdf_b && color_B && mlp && bound_host &&
                 rt_bound_host, "neus_forward: null input");
  GS_REQUIRE(color && depth && depth_var && normal && weight_sum && sdf && z_mid && grad_err_ray,
             "neus_forward: null output");
  GS_REQUIRE(n >= 0 && s > 0, "neus_forward: bad shape");
  if (n == 0) return GS_OK;
  const gs_grid_meta meta = host_meta();
  int first_hashed = GS_GRID_LEVELS;
  int nh = hashed_levels(meta, &first_hashed);
  const size_t need = carve_neus(nullptr, n, s, 0).total + 256;
  if (!workspace || workspace_bytes < need) {
    gs_set_error("neus_forward: workspace too small (%zu < %zu)", workspace_bytes, need);
    return GS_ERR_WORKSPACE;
  }
  // Level-major pays where the table traffic is the bound: measured on MI355X (profiles/r05_level_major_crossover.json,
  // r05_gather_replay.json), 32768 rays x 72 samples: 1149 -> 993 us, the gathers alone 1.07 -> 0.51 ms in the gather-only
  // replay; at 4096 rays a level's chunks do not fill one residency round per XCD, both orders are latency-bound and the
  // extra launch + the record round trip cost more than the misses (146 -> 162 us).  The two meet at ~442 K points; below
  // gs_neus_level_major_min_points (524288) the point kernel gathers every level itself (rec == nullptr), as the
  // one-workgroup force pass always does -- and so does a call whose workspace has no room for the records (a caller
  // that sized it before the crossover was lowered).
  if ((long long)n * s < level_major_min_points()) nh = 0;
  if (nh > 0 && workspace_bytes < carve_neus(nullptr, n, s, nh).total + 256) nh = 0;
  NeusWs ws = carve_neus((void*)gs_align((size_t)workspace), n, s, nh);
  hipStream_t st = (hipStream_t)stream;
  NeusArgs A;
  A.rays_o = rays_o; A.rays_d = rays_d; A.z_vals = z_vals; A.dists = dists;
  A.grid = (const _Float16*)grid; A.sdf_w = sdf_w; A.sdf_b = sdf_b; A.color_B = color_B;
  A.inv_s = inv_s;
  A.inv_s_dev = inv_s_dev;
  for (int k = 0; k < 6; ++k) { A.bound[k] = bound_host[k]; A.rt_bound[k] = rt_bound_host[k]; }
  A.rt_bound_dev = rt_bound_dev;
  A.n = n; A.s = s;
  const int np = n * s;
  float* alpha = alpha_out ? alpha_out : ws.alpha;
  float* grad = grad_out ? grad_out : ws.grad;
  _Float16* rgb = rgb_out ? (_Float16*)rgb_out : ws.rgb;
  if (mask_out) ws.mask = mask_out;
  GS_TIMING_PRE();
  // level-major encode of the hashed levels (records in the workspace), then the per-point stage
  if (nh > 0) {
    const int cpx = gs_cdiv(gs_cdiv(np, 256), 8);
    neus_encode_levels_kernel<<<8 * cpx * nh, 256, 0, st>>>(A, meta, ws.rec, (_Float16*)enc_aux_out, first_hashed, cpx);
    GS_CHECK_LAUNCH("neus_encode_levels");
  }
  // (the colour MLP runs in the point kernel's tail; the MLP input rows reach memory only when the caller saves them)
  neus_point_kernel<<<gs_cdiv(np, 256), 256, 0, st>>>(A, meta, ws.flags, 0, sdf, z_mid, alpha, grad,
                                                      ws.mask,
                                                      (_Float16*)mlp_in_out, (_Float16*)enc_aux_out, (const _Float16*)mlp,
                                                      rgb, nh > 0 ? ws.rec : nullptr, first_hashed);
  GS_CHECK_LAUNCH("neus_point");
  neus_point_kernel<<<1, 128, 0, st>>>(A, meta, ws.flags, 1, sdf, z_mid, alpha, grad, ws.mask,   // points 0..127
                                       (_Float16*)mlp_in_out, (_Float16*)enc_aux_out, (const _Float16*)mlp, rgb, nullptr,
                                       first_hashed);
  GS_CHECK_LAUNCH("neus_force100");
  neus_ray_kernel<<<gs_cdiv(n, 4), 256, 0, st>>>(alpha, rgb, z_mid, grad, ws.mask, color, depth, depth_var, normal,
                                                   weight_sum, grad_err_ray, grad_err_scale, sdf_variance_out,
                                                   sdf_variance_value, n, s);
  GS_CHECK_LAUNCH("neus_ray");
  return GS_OK;
}
