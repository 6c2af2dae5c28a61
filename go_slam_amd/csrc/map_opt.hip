// The mapper's optimiser step on ONE flat parameter buffer (reference src/mapping.py:55-58,135-137:
// clip_grad_norm_(35) over all trained parameters, then AdamW with two learning rates) in two launches:
//
//   map_grad_sqnorm_kernel   sum of squared gradients over [hash-table gradient | dense-parameter gradients] into a
//                            device scalar (the table gradient may stay in tiny-cuda-nn's loss-scaled fp16 form -- it is
//                            unscaled on the fly, never converted in a pass of its own);
//   map_adamw_kernel         clip coefficient from that scalar, unscale, AdamW (decoupled weight decay, bias-corrected
//                            moments, torch.optim.AdamW's formulas), and the fp16 working copy of the parameters that
//                            the next forward reads -- instead of foreach-norm + stack + norm + clamp + foreach-mul +
//                            fused AdamW (x2 groups) + a 12.6 M-entry fp32 -> fp16 cast + zero_grad.
//
// Traffic per step: table 12.6 M entries x (2 B grad + 4 B p + 4 B m + 4 B v read, 4 + 4 + 4 + 2 written) = 28 B
// -> 353 MB, one pass; HBM-bound (~60 us on MI355X).
#include "common.h"
#include <math.h>

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

// grads: element i < n16 comes from g16 (fp16, value * inv_scale16), element i >= n16 from g32[i - n16]
__global__ __launch_bounds__(256) void map_grad_sqnorm_kernel(const _Float16* __restrict__ g16, size_t n16,
                                                              float inv_scale16, const float* __restrict__ g32,
                                                              size_t n32, float* __restrict__ out) {
  const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256;
  float acc = 0.0f;
  const size_t n8 = n16 / 8;
  const half8* g8 = reinterpret_cast<const half8*>(g16);
  size_t i0 = tid;
  {                                                     // eight 16-byte loads in flight per thread (one at a time made
    float a4[4] = {0.f, 0.f, 0.f, 0.f};                 // the 25 MB pass latency-bound: 12.6 us for 12 dependent trips)
    for (; i0 + 7 * stride < n8; i0 += 8 * stride) {
      half8 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = g8[i0 + u * stride];
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float f = (float)v[u][k] * inv_scale16;
          a4[u & 3] = fmaf(f, f, a4[u & 3]);
        }
    }
    acc = (a4[0] + a4[1]) + (a4[2] + a4[3]);
  }
  for (size_t i = i0; i < n8; i += stride) {
    const half8 v = g8[i];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float f = (float)v[k] * inv_scale16;
      acc = fmaf(f, f, acc);
    }
  }
  for (size_t i = n8 * 8 + tid; i < n16; i += stride) {
    const float f = (float)g16[i] * inv_scale16;
    acc = fmaf(f, f, acc);
  }
  for (size_t i = tid; i < n32; i += stride) {
    const float f = g32[i];
    acc = fmaf(f, f, acc);
  }
  acc = gs_wave_sum(acc);
  __shared__ float part[4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) part[wv] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, part[0] + part[1] + part[2] + part[3]);
}

struct AdamArgs {
  // segment 16: parameters whose gradient arrives as loss-scaled fp16 (the hash table, or ONE RANK'S SLICE of it)
  float* p; float* m; float* v; _Float16* p16;
  const _Float16* g16; size_t n16;
  // segment 32: parameters with fp32 gradients (colour MLP, SDF layer, colour embedding, variance)
  float* pd; float* md; float* vd; _Float16* p16d;
  const float* g32; size_t n32;
  float inv_scale16;
  float lr16, lr32;              // learning rates of the two segments (volume / network parameters)
  float beta1, beta2, eps, wd, bc1, bc2_sqrt;
  const int* step_dev;           // != NULL: the step count lives on the device (graph replays), bc1 / bc2_sqrt from it
  const float* sqnorm; float max_norm;
};

__device__ __forceinline__ void adam_one(float& p, float& m, float& v, float g, float lr, const AdamArgs& A) {
  p = p * (1.0f - lr * A.wd);                          // decoupled weight decay
  m = A.beta1 * m + (1.0f - A.beta1) * g;              // = m.lerp(g, 1 - beta1)
  v = A.beta2 * v + (1.0f - A.beta2) * g * g;
  const float denom = sqrtf(v) / A.bc2_sqrt + A.eps;
  p = p - (lr / A.bc1) * (m / denom);
}

__global__ __launch_bounds__(256) void map_adamw_kernel(AdamArgs A) {
  const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256;
  if (A.step_dev) {
    const float t = (float)*A.step_dev;
    A.bc1 = 1.0f - powf(A.beta1, t);
    A.bc2_sqrt = sqrtf(1.0f - powf(A.beta2, t));
  }
  // torch.nn.utils.clip_grad_norm_: coef = max_norm / (total_norm + 1e-6), clamped to 1
  float coef = 1.0f;
  if (A.sqnorm) {
    const float c = A.max_norm / (sqrtf(*A.sqnorm) + 1e-6f);
    coef = c < 1.0f ? c : 1.0f;
  }
  const float s16 = A.inv_scale16 * coef;
  const size_t n8 = A.n16 / 8;
  for (size_t i = tid; i < n8; i += stride) {
    const half8 g = reinterpret_cast<const half8*>(A.g16)[i];
    float4 p0 = reinterpret_cast<float4*>(A.p)[2 * i], p1 = reinterpret_cast<float4*>(A.p)[2 * i + 1];
    float4 m0 = reinterpret_cast<float4*>(A.m)[2 * i], m1 = reinterpret_cast<float4*>(A.m)[2 * i + 1];
    float4 v0 = reinterpret_cast<float4*>(A.v)[2 * i], v1 = reinterpret_cast<float4*>(A.v)[2 * i + 1];
    float pp[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
    float mm[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
    float vv[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    half8 h;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      adam_one(pp[k], mm[k], vv[k], (float)g[k] * s16, A.lr16, A);
      h[k] = (_Float16)pp[k];
    }
    reinterpret_cast<float4*>(A.p)[2 * i] = make_float4(pp[0], pp[1], pp[2], pp[3]);
    reinterpret_cast<float4*>(A.p)[2 * i + 1] = make_float4(pp[4], pp[5], pp[6], pp[7]);
    reinterpret_cast<float4*>(A.m)[2 * i] = make_float4(mm[0], mm[1], mm[2], mm[3]);
    reinterpret_cast<float4*>(A.m)[2 * i + 1] = make_float4(mm[4], mm[5], mm[6], mm[7]);
    reinterpret_cast<float4*>(A.v)[2 * i] = make_float4(vv[0], vv[1], vv[2], vv[3]);
    reinterpret_cast<float4*>(A.v)[2 * i + 1] = make_float4(vv[4], vv[5], vv[6], vv[7]);
    if (A.p16) reinterpret_cast<half8*>(A.p16)[i] = h;
  }
  for (size_t i = n8 * 8 + tid; i < A.n16; i += stride) {      // (tail of a segment that is not a multiple of 8)
    float p = A.p[i], m = A.m[i], v = A.v[i];
    adam_one(p, m, v, (float)A.g16[i] * s16, A.lr16, A);
    A.p[i] = p; A.m[i] = m; A.v[i] = v;
    if (A.p16) A.p16[i] = (_Float16)p;
  }
  for (size_t i = tid; i < A.n32; i += stride) {
    float p = A.pd[i], m = A.md[i], v = A.vd[i];
    adam_one(p, m, v, A.g32[i] * coef, A.lr32, A);
    A.pd[i] = p; A.md[i] = m; A.vd[i] = v;
    if (A.p16d) A.p16d[i] = (_Float16)p;
  }
}

}  // namespace

extern "C" int gs_map_grad_sqnorm(const void* g16, size_t n16, float inv_scale16, const float* g32, size_t n32,
                                  float* sqnorm_out, gs_stream_t stream) {
  GS_REQUIRE(sqnorm_out && (g16 || n16 == 0) && (g32 || n32 == 0), "map_grad_sqnorm: null pointer");
  GS_REQUIRE(((size_t)g16 & 15) == 0, "map_grad_sqnorm: g16 must be 16-byte aligned");
  if (n16 + n32 == 0) return GS_OK;
  const size_t work = n16 / 8 + n32;
  unsigned blocks = (unsigned)((work + 255) / 256);
  if (blocks > 256) blocks = 256;       // every workgroup ends with an atomic on ONE address: 2048 of them cost ~25 us,
                                        // 512 still ~6 of this kernel's 12
  if (blocks == 0) blocks = 1;
  GS_TIMING_PRE();
  map_grad_sqnorm_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>((const _Float16*)g16, n16, inv_scale16, g32, n32,
                                                                   sqnorm_out);
  GS_CHECK_LAUNCH("map_grad_sqnorm");
  return GS_OK;
}

static int launch_adamw(AdamArgs A, int step, gs_stream_t stream) {
  if (A.n16 + A.n32 == 0) return GS_OK;
  if (!A.step_dev) {
    A.bc1 = 1.0f - powf(A.beta1, (float)step);
    A.bc2_sqrt = sqrtf(1.0f - powf(A.beta2, (float)step));
  } else {
    A.bc1 = A.bc2_sqrt = 1.0f;
  }
  const size_t work = A.n16 / 8 + A.n32;
  unsigned blocks = (unsigned)((work + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  if (blocks == 0) blocks = 1;
  GS_TIMING_PRE();
  map_adamw_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(A);
  GS_CHECK_LAUNCH("map_adamw");
  return GS_OK;
}

extern "C" int gs_map_adamw(float* p, float* m, float* v, void* p16, const void* g16, size_t n16, float inv_scale16,
                            const float* g32, size_t n, float lr16, float lr32, float beta1, float beta2, float eps,
                            float weight_decay, int step, const float* sqnorm, float max_norm, gs_stream_t stream) {
  GS_REQUIRE(p && m && v && (g16 || n16 == 0) && (g32 || n == n16), "map_adamw: null pointer");
  GS_REQUIRE(n16 <= n && n16 % 8 == 0 && step >= 1, "map_adamw: bad sizes (n16 must be a multiple of 8) or step");
  GS_REQUIRE((((size_t)p | (size_t)m | (size_t)v | (size_t)g16 | (size_t)p16) & 15) == 0,
             "map_adamw: buffers must be 16-byte aligned");
  AdamArgs A;
  A.p = p; A.m = m; A.v = v; A.p16 = (_Float16*)p16; A.g16 = (const _Float16*)g16; A.n16 = n16;
  A.pd = p + n16; A.md = m + n16; A.vd = v + n16; A.p16d = p16 ? (_Float16*)p16 + n16 : nullptr;
  A.g32 = g32; A.n32 = n - n16;
  A.inv_scale16 = inv_scale16; A.lr16 = lr16; A.lr32 = lr32;
  A.beta1 = beta1; A.beta2 = beta2; A.eps = eps; A.wd = weight_decay;
  A.step_dev = nullptr; A.sqnorm = sqnorm; A.max_norm = max_norm;
  return launch_adamw(A, step, stream);
}

// The two segments given separately: `*16` = one contiguous run of table entries (the whole table, or the slice a rank
// owns when the optimiser state is sharded over the ranks of a node), `*d` = the dense parameters.  `step_dev` (device
// int32, >= 1) replaces `step` when non-NULL, so a captured graph replays with the right bias corrections.
extern "C" int gs_map_adamw_seg(float* p, float* m, float* v, void* p16, const void* g16, size_t n16,
                                float inv_scale16, float* pd, float* md, float* vd, void* p16d, const float* g32,
                                size_t n32, float lr16, float lr32, float beta1, float beta2, float eps,
                                float weight_decay, int step, const int* step_dev, const float* sqnorm, float max_norm,
                                gs_stream_t stream) {
  GS_REQUIRE((n16 == 0 || (p && m && v && g16)) && (n32 == 0 || (pd && md && vd && g32)), "map_adamw_seg: null pointer");
  GS_REQUIRE(step_dev || step >= 1, "map_adamw_seg: step must be >= 1");
  GS_REQUIRE((((size_t)p | (size_t)m | (size_t)v | (size_t)g16 | (size_t)p16) & 15) == 0,
             "map_adamw_seg: table-segment buffers must be 16-byte aligned");
  AdamArgs A;
  A.p = p; A.m = m; A.v = v; A.p16 = (_Float16*)p16; A.g16 = (const _Float16*)g16; A.n16 = n16;
  A.pd = pd; A.md = md; A.vd = vd; A.p16d = (_Float16*)p16d; A.g32 = g32; A.n32 = n32;
  A.inv_scale16 = inv_scale16; A.lr16 = lr16; A.lr32 = lr32;
  A.beta1 = beta1; A.beta2 = beta2; A.eps = eps; A.wd = weight_decay;
  A.step_dev = step_dev; A.sqnorm = sqnorm; A.max_norm = max_norm;
  return launch_adamw(A, step, stream);
}

// ---- the small arithmetic AROUND the mapper step's big kernels, two launches instead of ~45 ---------------------------------
// A 4096-ray step is ~1.2 ms of GPU time, of which ~0.25 ms were ~50 one-block torch kernels (reductions over the ray
// batch, scalar arithmetic on 1-element tensors, zero-fills, slices of the Gram matrix, copies into the flat gradient):
//   map_step_prep_kernel   counts [valid rays, rays, max depth] (unless the caller provides all-rank counts), inv_s =
//                          clamp(exp(variance * scale)), the per-ray eikonal upstream gradient w_eik / (rays * samples),
//                          zeroed accumulators (d inv_s, squared gradient norm), step count + 1;
//   map_step_post_kernel   Gram-matrix chunks -> d sdf_layer.weight / bias, d color_B; MLP-backward workgroup partials ->
//                          d MLP; d variance; the step's loss (sum of the per-ray terms + eikonal mean) -- all written
//                          into the flat dense-gradient buffer [mlp 10240 | sdf_w 1120 | sdf_b 32 | cB 99 | var 1 | loss].
namespace {

__global__ __launch_bounds__(1024) void map_step_prep_kernel(const float* __restrict__ rays_depth, int n,
                                                             const float* __restrict__ variance, float scale_factor,
                                                             float w_eik, int s, const float* __restrict__ counts_in,
                                                             float* __restrict__ counts_out, float* __restrict__ inv_s_out,
                                                             float* __restrict__ d_gerr_out, float* __restrict__ d_invs,
                                                             float* __restrict__ sqnorm, int* __restrict__ step_dev,
                                                             const float* __restrict__ sdf_w, float* __restrict__ sdf_wt,
                                                             const _Float16* __restrict__ mlp16,
                                                             const int* __restrict__ frag_index,
                                                             _Float16* __restrict__ mlp_wpack) {
  __shared__ float red_c[16], red_m[16], bc[3];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (counts_in) {
    if (tid < 3) bc[tid] = counts_in[tid];
  } else {
    float c = 0.f, mx = -INFINITY;
    for (int i0 = tid; i0 < n; i0 += 1024 * 8) {       // eight loads in flight (a count and a maximum: any order)
      float d8[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) d8[u] = i0 + 1024 * u < n ? rays_depth[i0 + 1024 * u] : -INFINITY;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        c += d8[u] > 0.f ? 1.0f : 0.0f;
        mx = fmaxf(mx, d8[u]);
      }
    }
    c = gs_wave_sum(c);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    if (lane == 0) { red_c[wv] = c; red_m[wv] = mx; }
    __syncthreads();
    if (tid == 0) {
      float cs = 0.f, ms = -INFINITY;
      for (int k = 0; k < 16; ++k) { cs += red_c[k]; ms = fmaxf(ms, red_m[k]); }
      bc[0] = cs; bc[1] = (float)n; bc[2] = n > 0 ? ms : 0.0f;
    }
  }
  __syncthreads();
  if (tid < 3) counts_out[tid] = bc[tid];
  if (tid == 0) {
    const float raw = expf(variance[0] * scale_factor);
    inv_s_out[0] = fminf(fmaxf(raw, 1e-6f), 1e6f);
    d_invs[0] = 0.0f;
    sqnorm[0] = 0.0f;
    step_dev[0] += 1;
  }
  const float dg = w_eik / (bc[1] * (float)s);
  for (int i = tid; i < n; i += 1024) d_gerr_out[i] = dg;
  if (sdf_w && sdf_wt) {                               // [16][2][32] <- sdf_w [32][35] columns 3..34
    const int lf = tid >> 5, o = tid & 31;
    sdf_wt[tid] = sdf_w[o * 35 + 3 + lf];
  }
  if (mlp16 && frag_index && mlp_wpack) {              // the MLP backward's 40 A-fragments (was: cat + index + fill launches)
    // 20 entries per thread: all indices first, then all gathers (rolled, the two dependent loads per entry cost 40
    // memory round trips in this one-workgroup kernel: 5 -> 19 us)
    int j[20];
    _Float16 v[20];
#pragma unroll
    for (int u = 0; u < 20; ++u) j[u] = frag_index[tid + 1024 * u];
#pragma unroll
    for (int u = 0; u < 20; ++u) v[u] = mlp16[j[u] < 10240 ? j[u] : 0];
#pragma unroll
    for (int u = 0; u < 20; ++u) mlp_wpack[tid + 1024 * u] = j[u] < 10240 ? v[u] : (_Float16)0.0f;
  }
}

struct PostArgs {
  const float* gram; int nchunk;        // [nchunk][40][160] partial products rows[:, :40]^T rows of the per-point rows
  float inv_ls;                         // 1 / loss scale of the fp16 rows
  const float* mlp_partial; int nb;     // [nb][10240] workgroup partials of the MLP weight gradient (loss-scaled)
  const float* d_invs; const float* variance; const float* inv_s; float scale_factor;
  const float* loss_rays; const float* gerr; int n; float w_eik; int s; const float* counts;
  float* g32;                           // [10240 + 1120 + 32 + 99 + 1 + 1]
};

// workgroup = 32 outputs x 8 slices of the reduction axis (MLP workgroup partials, Gram chunks): the serial sums this
// replaces took 98 us on 45 workgroups
__global__ __launch_bounds__(256) void map_step_post_kernel(PostArgs A) {
  constexpr int N_MLP = 10240, N_W = 32 * 35, N_B = 32, N_CB = 3 * 33;
  constexpr int N_DENSE = N_W + N_B + N_CB;                 // outputs that come from the Gram matrix
  constexpr int B_MLP = N_MLP / 32, B_DENSE = (N_DENSE + 31) / 32;
  __shared__ float red[8][32];
  const int o = threadIdx.x & 31, sl = threadIdx.x >> 5;
  if ((int)blockIdx.x == B_MLP + B_DENSE) {                 // last workgroup: the loss and d variance
    float a = 0.f, b = 0.f;
    // eight loads of each array in flight, added in the order of the rolled loop (which paid one memory round trip per
    // element: 128 dependent trips at 32768 rays, 44 of this kernel's 54 us)
    for (int i0 = threadIdx.x; i0 < A.n; i0 += 256 * 8) {
      float va[8], vb[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + 256 * u;
        va[u] = i < A.n ? A.loss_rays[i] : 0.0f;
        vb[u] = i < A.n ? A.gerr[i] : 0.0f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (i0 + 256 * u < A.n) { a += va[u]; b += vb[u]; }
    }
    a = gs_wave_sum(a);
    b = gs_wave_sum(b);
    const float v = a + A.w_eik * b / (A.counts[1] * (float)A.s);
    if ((threadIdx.x & 63) == 0) red[0][threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
      A.g32[N_MLP + N_DENSE + 1] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
      const float raw = expf(A.variance[0] * A.scale_factor);   // d variance through inv_s = clamp(exp(variance * scale))
      A.g32[N_MLP + N_DENSE] = (raw >= 1e-6f && raw <= 1e6f) ? A.d_invs[0] * A.scale_factor * A.inv_s[0] : 0.0f;
    }
    return;
  }
  float t = 0.f;
  int out = -1;
  if ((int)blockIdx.x < B_MLP) {
    out = blockIdx.x * 32 + o;
    // 8 loads in flight per round trip, added in the same order as one by one (a rolled loop pays one L2 latency per
    // partial: 32 of them per thread in a launch of 330 small workgroups)
    for (int k0 = sl; k0 < A.nb; k0 += 64) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int k = k0 + 8 * u;
        v[u] = k < A.nb ? A.mlp_partial[(size_t)k * N_MLP + out] : 0.0f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (k0 + 8 * u < A.nb) t += v[u];
    }
  } else {
    const int e = ((int)blockIdx.x - B_MLP) * 32 + o;
    if (e < N_DENSE) {
      out = N_MLP + e;
      // per-point row layout [d_out 0:32 | x y z 1 32:36 (.. 40) | lin_in 40:80 | dw0 80:120 | d_arg 120:160]; the chunks
      // hold rows[:, :40]^T rows: G[r][c], r < 40
      int r0, c0, r1 = -1, c1 = 0;
      if (e < N_W) {                                        // d sdf_layer.weight [32][35] = d_out^T lin_in; row 0 += colsum(dw0)
        const int oo = e / 35, c = e - oo * 35;
        r0 = oo; c0 = 40 + c;
        if (oo == 0) { r1 = 35; c1 = 80 + c; }
      } else if (e < N_W + N_B) {                           // d bias = colsum(d_out) (row 35 = the ones column)
        r0 = 35; c0 = e - N_W;
      } else {                                              // d color_B [3][33] = pts^T d_arg
        const int q = e - N_W - N_B, d = q / 33, c = q - d * 33;
        r0 = 32 + d; c0 = 120 + c;
      }
      for (int k0 = sl; k0 < A.nchunk; k0 += 64) {
        float v0[8], v1[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int k = k0 + 8 * u;
          const float* g = A.gram + (size_t)(k < A.nchunk ? k : 0) * 40 * 160;
          v0[u] = g[r0 * 160 + c0];
          v1[u] = r1 >= 0 ? g[r1 * 160 + c1] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (k0 + 8 * u >= A.nchunk) continue;
          t += v0[u];
          if (r1 >= 0) t += v1[u];
        }
      }
    }
  }
  red[sl][o] = t;
  __syncthreads();
  if (sl == 0 && out >= 0) {
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) v += red[k][o];
    A.g32[out] = v * A.inv_ls;
  }
}

}  // namespace

// ---- dense-parameter gradients: the per-point rows' partial Gram products on the matrix cores ---------------------------
// The backward's per-point rows [np, 160] fp16 ([d_out 32 | x y z 1 .. 8 | lin_in 40 | dw0 40 | d_arg 40], loss-scaled)
// carry every dense-parameter gradient as rows[:, :40]^T rows.  Until round 4 this product was the one library kernel of
// the fused mapper step (a batched hipBLASLt GEMM over 2048 / 8192-row chunks, 45 us at 4096 rays, 227 us at 32768).
// Here: the contraction runs over POINTS, so an MFMA operand (8 consecutive k per lane) is a COLUMN of the row-major
// rows -- each wave streams 16 rows at a time (5 coalesced 16-byte loads per lane, requested one group ahead) through a
// private LDS tile and gathers its five 32-column fragments F0..F4 from it with 2-byte reads (row stride 336 B: the two
// half-waves, 8 rows apart, fall into different banks).  Only the six 32x32 products the gradients need are formed:
// F0^T F1, F0^T F2 (d_out^T lin_in) and F1^T F0, F1^T F2, F1^T F3, F1^T F4 (the [x y z 1] rows: column sums and
// pts^T d_arg).  The four waves' accumulators are added in LDS in a fixed order; one partial [40][160] per workgroup
// goes to gs_map_step_post, which sums the partials as it summed the GEMM's chunks.  Memory-bound: 320 B per point read once.
namespace {
constexpr int GRAM_TS = 168;                      // LDS tile row stride in halves (336 B: 16-byte aligned, 84 dwords)
typedef _Float16 gram_h8 __attribute__((ext_vector_type(8)));
typedef float gram_f16v __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512) void map_gram_kernel(const _Float16* __restrict__ rows, int ngroups,
                                                       float* __restrict__ partial) {
  constexpr int NW = 8;                            // waves per workgroup: two per SIMD, so one wave's LDS gather overlaps the other's MFMAs
  __shared__ __attribute__((aligned(16))) _Float16 tile[NW][16 * GRAM_TS];
  __shared__ float red[6][1024];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  // groups of 16 rows: this workgroup's contiguous share, dealt to its four waves round-robin
  const int per = (ngroups + (int)gridDim.x - 1) / (int)gridDim.x;
  const int g_lo = (int)blockIdx.x * per, g_hi = min(ngroups, g_lo + per);
  gram_f16v acc[6];
#pragma unroll
  for (int t = 0; t < 6; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.0f;
  _Float16* tl = tile[wv];
  // piece p of a group (320 pieces of 16 B): row p / 20, 16-byte part p % 20
  int prow[5], ppart[5];
#pragma unroll
  for (int u = 0; u < 5; ++u) {
    const int p = lane + 64 * u;
    prow[u] = p / 20;
    ppart[u] = p - prow[u] * 20;
  }
  const int r = lane & 31, kh = lane >> 5;
  struct Group { uint4 p0, p1, p2, p3, p4; };      // (by value: five named registers sets, never an indexed array)
  const int o0 = prow[0] * 20 + ppart[0], o1 = prow[1] * 20 + ppart[1], o2 = prow[2] * 20 + ppart[2],
            o3 = prow[3] * 20 + ppart[3], o4 = prow[4] * 20 + ppart[4];
  const int l0 = prow[0] * GRAM_TS + ppart[0] * 8, l1 = prow[1] * GRAM_TS + ppart[1] * 8,
            l2 = prow[2] * GRAM_TS + ppart[2] * 8, l3 = prow[3] * GRAM_TS + ppart[3] * 8,
            l4 = prow[4] * GRAM_TS + ppart[4] * 8;
  auto fetch = [&](int g) -> Group {
    const uint4* src = reinterpret_cast<const uint4*>(rows + (size_t)g * 16 * 160);
    Group v;
    v.p0 = src[o0]; v.p1 = src[o1]; v.p2 = src[o2]; v.p3 = src[o3]; v.p4 = src[o4];
    return v;
  };
  auto consume = [&](const Group v) {               // one 16-row group: registers -> LDS tile -> fragments -> 6 MFMAs
    *reinterpret_cast<uint4*>(tl + l0) = v.p0;
    *reinterpret_cast<uint4*>(tl + l1) = v.p1;
    *reinterpret_cast<uint4*>(tl + l2) = v.p2;
    *reinterpret_cast<uint4*>(tl + l3) = v.p3;
    *reinterpret_cast<uint4*>(tl + l4) = v.p4;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    gram_h8 f[5];
#pragma unroll
    for (int t = 0; t < 5; ++t)
#pragma unroll
      for (int j = 0; j < 8; ++j) f[t][j] = tl[(8 * kh + j) * GRAM_TS + 32 * t + r];
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[0], f[1], acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[0], f[2], acc[1], 0, 0, 0);
    acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[1], f[0], acc[2], 0, 0, 0);
    acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[1], f[2], acc[3], 0, 0, 0);
    acc[4] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[1], f[3], acc[4], 0, 0, 0);
    acc[5] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[1], f[4], acc[5], 0, 0, 0);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();                // every lane has read the tile before the next group overwrites it
  };
  int g = g_lo + wv;
  Group cur = Group();
  if (g < g_hi) cur = fetch(g);
#pragma unroll 1
  for (; g < g_hi; g += NW) {
    Group nx = cur;
    if (g + NW < g_hi) nx = fetch(g + NW);          // the next group's rows are in flight during this group's products
    consume(cur);
    cur = nx;
  }
  // merge the waves (fixed order: deterministic), C layout: reg e -> m = (e & 3) + 8 (e >> 2) + 4 kh, n = r
  for (int w = 0; w < NW; ++w) {
    if (wv == w) {
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int m = (e & 3) + 8 * (e >> 2) + 4 * kh;
          float* dst = &red[t][m * 32 + r];
          *dst = (w == 0 ? 0.0f : *dst) + acc[t][e];
        }
    }
    __syncthreads();
  }
  // tile t = (row block tm, column block tn) of the [40][160] partial; rows >= 40 of the second row block are padding
  float* out = partial + (size_t)blockIdx.x * 40 * 160;
  for (int idx = tid; idx < 6 * 1024; idx += 512) {
    const int t = idx >> 10, m = (idx >> 5) & 31, n = idx & 31;
    const int tm = t < 2 ? 0 : 1;
    const int tn = t == 0 ? 1 : (t == 1 ? 2 : (t == 2 ? 0 : t - 1));
    const int gm = 32 * tm + m;
    if (gm < 40) out[gm * 160 + 32 * tn + n] = red[t][m * 32 + n];
  }
}
}  // namespace

extern "C" int gs_map_gram_blocks(int n_rows) {
  const int ngroups = n_rows / 16;
  int nb = ngroups / 64;                          // >= 64 groups (1024 rows) per workgroup
  if (nb < 1) nb = 1;
  return nb > 256 ? 256 : nb;
}

extern "C" int gs_map_gram(const void* rows, int n_rows, float* partial, gs_stream_t stream) {
  GS_REQUIRE(rows && partial, "map_gram: null pointer");
  GS_REQUIRE(n_rows > 0 && n_rows % 16 == 0, "map_gram: the row count must be a positive multiple of 16 (pad with zero rows)");
  GS_REQUIRE(((size_t)rows & 15) == 0, "map_gram: rows must be 16-byte aligned");
  const int nb = gs_map_gram_blocks(n_rows);
  GS_TIMING_PRE();
  map_gram_kernel<<<nb, 512, 0, (hipStream_t)stream>>>((const _Float16*)rows, n_rows / 16, partial);
  GS_CHECK_LAUNCH("map_gram");
  return GS_OK;
}

extern "C" int gs_map_step_prep(const float* rays_depth, int n, const float* variance, float scale_factor, float w_eikonal,
                                int samples, const float* counts_in, float* counts_out, float* inv_s_out, float* d_gerr_out,
                                float* d_invs, float* sqnorm, int* step_dev, const float* sdf_w, float* sdf_wt_out,
                                const void* mlp16, const int* frag_index, void* mlp_wpack_out, gs_stream_t stream) {
  GS_REQUIRE((rays_depth || counts_in) && variance && counts_out && inv_s_out && d_gerr_out && d_invs && sqnorm && step_dev,
             "map_step_prep: null pointer");
  GS_REQUIRE(n >= 0 && samples > 0, "map_step_prep: bad shape");
  GS_TIMING_PRE();
  map_step_prep_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(rays_depth, n, variance, scale_factor, w_eikonal, samples,
                                                           counts_in, counts_out, inv_s_out, d_gerr_out, d_invs, sqnorm,
                                                           step_dev, sdf_w, sdf_wt_out, (const _Float16*)mlp16, frag_index,
                                                           (_Float16*)mlp_wpack_out);
  GS_CHECK_LAUNCH("map_step_prep");
  return GS_OK;
}

extern "C" int gs_map_step_post(const float* gram_chunks, int nchunk, float inv_loss_scale, const float* mlp_partial,
                                int nb, const float* d_invs, const float* variance, const float* inv_s, float scale_factor,
                                const float* loss_rays, const float* gerr, int n, float w_eikonal, int samples,
                                const float* counts, float* g32, gs_stream_t stream) {
  GS_REQUIRE(gram_chunks && mlp_partial && d_invs && variance && inv_s && loss_rays && gerr && counts && g32,
             "map_step_post: null pointer");
  GS_REQUIRE(nchunk > 0 && nb > 0 && n >= 0 && samples > 0, "map_step_post: bad shape");
  PostArgs A;
  A.gram = gram_chunks; A.nchunk = nchunk; A.inv_ls = inv_loss_scale; A.mlp_partial = mlp_partial; A.nb = nb;
  A.d_invs = d_invs; A.variance = variance; A.inv_s = inv_s; A.scale_factor = scale_factor; A.loss_rays = loss_rays;
  A.gerr = gerr; A.n = n; A.w_eik = w_eikonal; A.s = samples; A.counts = counts; A.g32 = g32;
  const int blocks = 10240 / 32 + (32 * 35 + 32 + 99 + 31) / 32 + 1;
  GS_TIMING_PRE();
  map_step_post_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(A);
  GS_CHECK_LAUNCH("map_step_post");
  return GS_OK;
}
