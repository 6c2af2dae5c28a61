// On-the-fly ("alt") correlation lookup for the global / loop-closure BA path
// (reference: src/lib/altcorr_kernel.cu:27-149, called per pyramid level from
// src/modules/corr.py:112-131).
//
// The reference uses 32-thread blocks that stage f1/f2 through shared memory with a
// __syncthreads per tap.  Here one wave64 serves one source pixel at a time: lane = (tx, cc) with
// tx = window column (8) and cc = channel chunk (8 x C/8 channels).  A window ROW of fmap2 is
// 8 x C contiguous channels-last values, so a single wave-wide load instruction fetches a whole
// row fully coalesced (2 KB of fp16 / 4 KB of fp32); the partial dot products are reduced over the
// 8 chunk lanes with three DPP steps, the bilinear blend needs one cross-lane move per row, and
// the 49 outputs of 16 consecutive pixels are gathered in LDS so that global stores are 64-byte
// segments instead of the 4-byte scatters a per-pixel write would give.  fp32 accumulation
// throughout (v_dot2_f32_f16 for fp16 inputs).
#include "common.h"

namespace {

constexpr int PIXW = 16;     // pixels per wave strip

typedef _Float16 half2v __attribute__((ext_vector_type(2)));

template <typename T, int CPL>   // CPL = channels per lane
__device__ __forceinline__ void load_chunk(const T* p, T (&dst)[CPL]) {
  constexpr int BYTES = CPL * sizeof(T);
  static_assert(BYTES % 16 == 0, "chunk must be a multiple of 16 bytes");
  const uint4* src = reinterpret_cast<const uint4*>(p);
  uint4* d = reinterpret_cast<uint4*>(dst);
#pragma unroll
  for (int i = 0; i < BYTES / 16; ++i) d[i] = src[i];
}

template <int CPL>
__device__ __forceinline__ float dot_chunk(const float (&a)[CPL], const float (&b)[CPL]) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < CPL; ++i) s = fmaf(a[i], b[i], s);
  return s;
}

template <int CPL>
__device__ __forceinline__ float dot_chunk(const _Float16 (&a)[CPL], const _Float16 (&b)[CPL]) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < CPL; i += 2) {
    half2v x = {a[i], a[i + 1]}, y = {b[i], b[i + 1]};
    s = __builtin_amdgcn_fdot2(x, y, s, false);
  }
  return s;
}

template <typename T, int C>
__global__ __launch_bounds__(256) void altcorr_forward_kernel(
    const T* __restrict__ fmap1, const T* __restrict__ fmap2, const float* __restrict__ coords,
    T* __restrict__ corr, int S, int H1, int W1, int H2, int W2, int r) {
  constexpr int CPL = C / 8;
  __shared__ float tile[4][49][PIXW + 1];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int tx = lane >> 3, cc = lane & 7;
  const int bs = blockIdx.y;                 // b * S + s
  const int b = bs / S;
  const int hw1 = H1 * W1;
  const int p0 = (blockIdx.x * 4 + wave) * PIXW;
  if (p0 >= hw1) return;
  const int npix = min(PIXW, hw1 - p0);
  const int rd = 2 * r + 1;                  // == 7 (checked by the launcher)
  const T* f1b = fmap1 + (size_t)b * hw1 * C;
  const T* f2b = fmap2 + (size_t)b * H2 * W2 * C;

  for (int pi = 0; pi < npix; ++pi) {
    const int p = p0 + pi;
    const float2 cxy = reinterpret_cast<const float2*>(coords)[(size_t)bs * hw1 + p];
    const float fx0 = floorf(cxy.x), fy0 = floorf(cxy.y);
    const float dx = cxy.x - fx0, dy = cxy.y - fy0;
    const int x0 = (int)fminf(fmaxf(fx0, -1.0e6f), 1.0e6f) - r + tx;
    const int y0 = (int)fminf(fmaxf(fy0, -1.0e6f), 1.0e6f) - r;
    T a[CPL];
    load_chunk<T, CPL>(f1b + (size_t)p * C + cc * CPL, a);
    const bool xin = (x0 >= 0) && (x0 < W2);
    float prev = 0.f, prev_n = 0.f;
#pragma unroll 1
    for (int iy = 0; iy <= rd; ++iy) {
      const int y = y0 + iy;
      float s = 0.f;
      if (xin && y >= 0 && y < H2) {
        T bv[CPL];
        load_chunk<T, CPL>(f2b + ((size_t)y * W2 + x0) * C + cc * CPL, bv);
        s = dot_chunk<CPL>(a, bv);
      }
      // sum the 8 channel chunks (lanes tx*8 .. tx*8+7)
      s = s + gs_dpp<0xB1>(s);
      s = s + gs_dpp<0x4E>(s);
      s = s + gs_dpp<0x141>(s);
      const float s_n = __shfl_down(s, 8, 64);          // tap to the right (tx + 1)
      if (iy > 0 && tx < rd && cc == 0) {
        // channel (iy-1) + rd*tx: the four taps around it, reference accumulation order
        float o = prev * ((1.0f - dy) * (1.0f - dx));
        o = fmaf(prev_n, (1.0f - dy) * dx, o);
        o = fmaf(s, dy * (1.0f - dx), o);
        o = fmaf(s_n, dy * dx, o);
        tile[wave][(iy - 1) + rd * tx][pi] = o;
      }
      prev = s;
      prev_n = s_n;
    }
  }
  __syncthreads();
  // the strip's 49 x npix outputs, 64-byte segments per channel row
  T* out = corr + (size_t)bs * 49 * hw1 + p0;
  for (int i = lane; i < 49 * PIXW; i += 64) {
    const int c = i / PIXW, pi = i % PIXW;
    if (pi < npix) out[(size_t)c * hw1 + pi] = (T)tile[wave][c][pi];
  }
}

template <typename T>
int launch(const void* f1, const void* f2, const float* coords, void* corr, int B, int S, int H1, int W1, int H2,
           int W2, int C, int r, hipStream_t st) {
  const int hw1 = H1 * W1;
  dim3 grid(gs_cdiv(hw1, 4 * PIXW), B * S);
  if (C == 128)
    altcorr_forward_kernel<T, 128><<<grid, 256, 0, st>>>((const T*)f1, (const T*)f2, coords, (T*)corr, S, H1, W1, H2, W2, r);
  else if (C == 256)
    altcorr_forward_kernel<T, 256><<<grid, 256, 0, st>>>((const T*)f1, (const T*)f2, coords, (T*)corr, S, H1, W1, H2, W2, r);
  else if (C == 64)
    altcorr_forward_kernel<T, 64><<<grid, 256, 0, st>>>((const T*)f1, (const T*)f2, coords, (T*)corr, S, H1, W1, H2, W2, r);
  else {
    gs_set_error("altcorr_forward: C=%d not built (64, 128, 256)", C);
    return GS_ERR_UNSUPPORTED;
  }
  GS_CHECK_LAUNCH("altcorr_forward");
  return GS_OK;
}

}  // namespace

extern "C" int gs_altcorr_forward(const void* fmap1, const void* fmap2, const float* coords, void* corr, int b,
                                  int s, int h1, int w1, int h2, int w2, int c, int radius, int dtype,
                                  gs_stream_t stream) {
  GS_REQUIRE(fmap1 && fmap2 && coords && corr, "altcorr_forward: null pointer");
  GS_REQUIRE(b >= 0 && s > 0 && h1 > 0 && w1 > 0 && h2 > 0 && w2 > 0, "altcorr_forward: bad shape");
  GS_REQUIRE(radius == 3, "altcorr_forward: only radius 3 (the reference's value) is built");
  if (b == 0) return GS_OK;
  GS_REQUIRE((long)b * s <= 65535, "altcorr_forward: b*s=%ld exceeds grid.y limit", (long)b * s);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == GS_F32) return launch<float>(fmap1, fmap2, coords, corr, b, s, h1, w1, h2, w2, c, radius, st);
  if (dtype == GS_F16) return launch<_Float16>(fmap1, fmap2, coords, corr, b, s, h1, w1, h2, w2, c, radius, st);
  gs_set_error("altcorr_forward: unsupported dtype %d", dtype);
  return GS_ERR_UNSUPPORTED;
}

// ---- backward (reference: src/lib/altcorr_kernel.cu:151-283, fp32 only, training path) ---------
// d fmap1[p] = sum_{s,iy,ix} g * fmap2[window(iy,ix)],  d fmap2[window(iy,ix)] += g * fmap1[p]  with
// g = the four bilinear-weighted corr_grad taps that touched that window cell in the forward pass;
// coords receive no gradient (the reference returns zeros).  One wave per source pixel: lane L first
// computes g for window cell L (8x8 cells), then the lanes switch to channels -- a window cell's C
// floats are one coalesced load and one coalesced run of atomics.
namespace {
__global__ __launch_bounds__(256) void altcorr_backward_kernel(const float* __restrict__ f1, const float* __restrict__ f2,
                                                               const float* __restrict__ coords,
                                                               const float* __restrict__ cg, float* __restrict__ g1,
                                                               float* __restrict__ g2, int S, int H1, int W1, int H2,
                                                               int W2, int C, size_t npix) {
  __shared__ float gs[4][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const size_t pw = (size_t)blockIdx.x * 4 + wv;           // b*H1*W1 + p
  if (pw >= npix) return;
  const int hw1 = H1 * W1;
  const int b = (int)(pw / hw1), p = (int)(pw % hw1);
  const int iy = lane >> 3, ix = lane & 7;
  const float* f1p = f1 + pw * (size_t)C;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};                     // C <= 256: channels lane, lane+64, ...
  for (int s = 0; s < S; ++s) {
    const float* cp = coords + ((size_t)(b * S + s) * hw1 + p) * 2;
    const float x2 = cp[0], y2 = cp[1];
    const float fx = floorf(x2), fy = floorf(y2);
    const float dx = x2 - fx, dy = y2 - fy;
    const float* gp = cg + ((size_t)(b * S + s) * 49) * hw1 + p;      // channel stride hw1
    float g = 0.0f;
    if (iy > 0 && ix > 0) g += gp[(size_t)((iy - 1) + 7 * (ix - 1)) * hw1] * dy * dx;
    if (iy > 0 && ix < 7) g += gp[(size_t)((iy - 1) + 7 * ix) * hw1] * dy * (1.0f - dx);
    if (iy < 7 && ix > 0) g += gp[(size_t)(iy + 7 * (ix - 1)) * hw1] * (1.0f - dy) * dx;
    if (iy < 7 && ix < 7) g += gp[(size_t)(iy + 7 * ix) * hw1] * (1.0f - dy) * (1.0f - dx);
    gs[wv][lane] = g;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int y0 = (int)fminf(fmaxf(fy, -1.0e6f), 1.0e6f) - 3, x0 = (int)fminf(fmaxf(fx, -1.0e6f), 1.0e6f) - 3;
    for (int cell = 0; cell < 64; ++cell) {
      const int h2 = y0 + (cell >> 3), w2 = x0 + (cell & 7);
      if (h2 < 0 || h2 >= H2 || w2 < 0 || w2 >= W2) continue;       // wave-uniform
      const float gc = gs[wv][cell];
      const size_t o2 = ((size_t)(b * H2 + h2) * W2 + w2) * C;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int c = lane + 64 * k;
        if (c < C) {
          acc[k] = fmaf(gc, f2[o2 + c], acc[k]);
          atomicAdd(g2 + o2 + c, gc * f1p[c]);
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = lane + 64 * k;
    if (c < C) g1[pw * (size_t)C + c] = acc[k];
  }
}
}  // namespace

extern "C" int gs_altcorr_backward(const float* fmap1, const float* fmap2, const float* coords, const float* corr_grad,
                                   float* fmap1_grad, float* fmap2_grad, int b, int s, int h1, int w1, int h2, int w2,
                                   int c, int radius, gs_stream_t stream) {
  GS_REQUIRE(fmap1 && fmap2 && coords && corr_grad && fmap1_grad && fmap2_grad, "altcorr_backward: null pointer");
  GS_REQUIRE(b >= 0 && s > 0 && h1 > 0 && w1 > 0 && h2 > 0 && w2 > 0, "altcorr_backward: bad shape");
  GS_REQUIRE(radius == 3, "altcorr_backward: only radius 3 (the reference's value) is built");
  GS_REQUIRE(c > 0 && c <= 256, "altcorr_backward: channels %d > 256", c);
  if (b == 0) return GS_OK;
  const size_t npix = (size_t)b * h1 * w1;
  altcorr_backward_kernel<<<(unsigned)((npix + 3) / 4), 256, 0, (hipStream_t)stream>>>(
      fmap1, fmap2, coords, corr_grad, fmap1_grad, fmap2_grad, s, h1, w1, h2, w2, c, npix);
  GS_CHECK_LAUNCH("altcorr_backward");
  return GS_OK;
}
