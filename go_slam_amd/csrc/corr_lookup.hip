// Correlation-volume window lookup (reference: src/lib/correlation_kernels.cu:19-124 and the
// 4-level loop of src/modules/corr.py:43-53).
//
// Two kernels:
//  * corr_pyramid_coop_kernel (further down) -- the production path (fp16 volume, NHWC output, all 4
//    levels fused, row-major or tile8 volume layout): 8 lanes per source pixel, one window row per lane.
//  * the lane-per-pixel kernels (corr_pyramid_kernel, corr_index_*): fp32/fp64 volumes, planar output,
//    the reference's single-level ABI and its backward.  One lane owns one source pixel p=(y,x) of one
//    edge: its correlation slice volume[n][y][x][:][:] is private to the lane, so nothing is shared
//    through LDS; the lane streams the 8 rows of its 8x8 window (one unaligned 128-bit load per row for
//    fp16, clamped into the plane + funnel shift at the borders) and emits the 49 bilinear taps.
// The bilinear blend is carried in the volume's dtype in the reference's order (i outer, j inner,
// `corr += s * T(w)`) with at::Half's compute-in-fp32-round-to-fp16 operator semantics, which makes
// fp16 results bit-identical to the reference's fp16-accumulated values; the file is compiled with
// -ffp-contract=off.
#include "common.h"

namespace {

// float -> int with the huge/NaN cases pinned far outside any map (the reference's
// static_cast<int> saturates; a plain cast could wrap `xs + 8` back into range).
__device__ __forceinline__ int safe_int(float f) { return (int)fminf(fmaxf(f, -1.0e6f), 1.0e6f); }

// The reference instantiates its kernels with at::Half, whose operators compute in fp32 and
// round the result back to fp16 (c10/util/Half-inl.h) -- i.e. `a + b` is half(float(a)+float(b)),
// a double rounding that a native v_add_f16 does not reproduce in rare tie cases.  Emulate it.
// `pin` keeps hipcc from folding an fp32 op and the following fp32->fp16 conversion into one
// single-rounding v_fma_mixlo_f16 (it does so even under -ffp-contract=off).
__device__ __forceinline__ float pin(float v) { asm volatile("" : "+v"(v)); return v; }
template <typename T> __device__ __forceinline__ T mul_r(T a, T b) { return a * b; }
template <typename T> __device__ __forceinline__ T add_r(T a, T b) { return a + b; }
template <> __device__ __forceinline__ _Float16 mul_r(_Float16 a, _Float16 b) { return (_Float16)pin((float)a * (float)b); }
template <> __device__ __forceinline__ _Float16 add_r(_Float16 a, _Float16 b) { return (_Float16)pin((float)a + (float)b); }
// scalar_t(w): the fp32 weight product is rounded to fp32 first, then converted
template <typename T> __device__ __forceinline__ T wcast(float w) { return (T)pin(w); }

template <typename T> struct VecRow;   // 8 consecutive taps of a window row
template <> struct VecRow<_Float16> { typedef struct __attribute__((packed, aligned(2))) { _Float16 v[8]; } type; };
template <> struct VecRow<float>    { typedef struct __attribute__((packed, aligned(4))) { float v[8]; } type; };
template <> struct VecRow<double>   { typedef struct __attribute__((packed, aligned(8))) { double v[8]; } type; };


// 8x8 window s[i][j] = slice[ys+j][xs+i] (0 outside the plane).
template <typename T>
__device__ __forceinline__ void load_window(const T* __restrict__ slice, int h2, int w2, int xs, int ys, T (&s)[8][8]) {
  const bool xin = (xs >= 0) && (xs + 8 <= w2);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int y1 = ys + j;
    const bool yin = (y1 >= 0) && (y1 < h2);
    if (yin && xin) {
      typename VecRow<T>::type row = *reinterpret_cast<const typename VecRow<T>::type*>(slice + (size_t)y1 * w2 + xs);
#pragma unroll
      for (int i = 0; i < 8; ++i) s[i][j] = row.v[i];
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int x1 = xs + i;
        const bool in = yin && (x1 >= 0) && (x1 < w2);
        s[i][j] = in ? slice[(size_t)(in ? y1 : 0) * w2 + (in ? x1 : 0)] : (T)0;
      }
    }
  }
}

// fp16 (the production dtype): no per-element border path.  Every row is ONE unaligned 128-bit load
// from the start column clamped into the plane (planes are >= 8 wide at every pyramid level the
// fused kernels accept), then a 128-bit funnel shift by the clamp distance moves the taps into place
// and shifts zeros in for the columns outside the plane.  On the coarse levels (20x15, 10x7 planes)
// most windows straddle a border, and the element path cost 64 two-byte loads per lane there.
template <>
__device__ __forceinline__ void load_window<_Float16>(const _Float16* __restrict__ slice, int h2, int w2, int xs,
                                                      int ys, _Float16 (&s)[8][8]) {
  typedef unsigned __int128 u128;
  if (w2 < 8) {   // never on the fused path; keeps the ABI entry total
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int x1 = xs + i, y1 = ys + j;
        const bool in = (y1 >= 0) && (y1 < h2) && (x1 >= 0) && (x1 < w2);
        s[i][j] = in ? slice[(size_t)(in ? y1 : 0) * w2 + (in ? x1 : 0)] : (_Float16)0;
      }
    return;
  }
  const int xc = min(max(xs, 0), w2 - 8);
  const int d = min(max(xs - xc, -8), 8);       // s[i] = row[i + d]
  const bool none = (d <= -8) || (d >= 8);
  const int sh = 16 * (d < 0 ? -d : d);
  // all eight rows are requested before the first one is used, from row indices clamped into the plane (rows outside it
  // are zeroed by a select afterwards): one load per row behind its own `if (yin)` was eight dependent round trips per
  // level and lane -- 32 per lookup, most of the single-edge lookup's 27 us in MotionFilter.track
  typedef struct __attribute__((packed, aligned(2))) { u128 q; } U;
  u128 raw[8];
#pragma unroll
  for (int j = 0; j < 8; ++j)
    raw[j] = reinterpret_cast<const U*>(slice + (size_t)min(max(ys + j, 0), h2 - 1) * w2 + xc)->q;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int y1 = ys + j;
    const bool yin = (y1 >= 0) && (y1 < h2) && !none;
    u128 v = yin ? raw[j] : (u128)0;
    if (d > 0) v >>= sh;
    else if (d < 0) v <<= sh;
    const unsigned long long lo = (unsigned long long)v, hi = (unsigned long long)(v >> 64);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const unsigned short bits = (unsigned short)((i < 4 ? lo : hi) >> (16 * (i & 3)));
      s[i][j] = __builtin_bit_cast(_Float16, bits);
    }
  }
}

// ---- packed fp16 arithmetic (the production dtype on the fused NHWC kernel) ------------------------
// at::Half's "compute in fp32, round to fp16" equals a native fp16 operation for a single + or x
// (fp32's 24-bit significand >= 2*11 + 2, so the double rounding is innocuous), hence v_pk_mul_f16 /
// v_pk_add_f16 reproduce the reference bit for bit while blending TWO taps per instruction and
// skipping ~10 conversions per tap (the scalar emulation above costs ~17 VALU instructions per tap).
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ half2v pin2(half2v v) { asm volatile("" : "+v"(v)); return v; }
__device__ __forceinline__ half2v as_h2(unsigned int u) { return __builtin_bit_cast(half2v, u); }

// ---- wave-cooperative fused lookup (fp16, NHWC): 8 lanes per pixel, one window ROW per lane --------
// PMC on the lane-per-pixel kernel (tools/profile_lookup_only.py; TA_BUSY ~100 %, TCP_PENDING_STALL ~100 %
// of the kernel's cycles, VALU < 10 %) shows it is bound by the L1/TA miss machinery, not by HBM bytes or
// ALU work: every load instruction presents 64 unrelated addresses, i.e. 64 tag lookups and up to 64
// outstanding misses, and the 8 rows of a window are fetched by 8-16 separate instructions that all
// hit the same few lines.  Here the 8 rows of a window are loaded by 8 adjacent lanes with ONE
// instruction (two for a straddled tile8 row), so the TA sees the window as 2-4 whole 128-byte lines;
// the row below comes from the neighbouring lane (ds_bpermute), each lane blends the 7 taps of its
// row two at a time (packed fp16, see above), and the 196 channels of 8 consecutive pixels are
// assembled in LDS and stored as one contiguous 3136-byte run.
template <bool TILED>
__device__ __forceinline__ void load_row_coop(const _Float16* __restrict__ plane, int h2, int w2, int xs, int y1,
                                              unsigned int (&R)[4]) {
  typedef unsigned __int128 u128;
  u128 v = 0;
  const bool yin = (y1 >= 0) && (y1 < h2);
  if constexpr (TILED) {
    const int ntx = w2 >> 3;
    const int tx0 = xs >> 3, ox = xs & 7;
    const _Float16* row = plane + ((size_t)(y1 >> 3) * ntx) * 64 + (y1 & 7) * 8;
    u128 lo = 0, hi = 0;
    if (yin && tx0 >= 0 && tx0 < ntx) lo = *reinterpret_cast<const u128*>(row + (size_t)tx0 * 64);
    if (yin && ox != 0 && tx0 + 1 >= 0 && tx0 + 1 < ntx) hi = *reinterpret_cast<const u128*>(row + (size_t)(tx0 + 1) * 64);
    v = lo;
    if (ox != 0) v = (lo >> (16 * ox)) | (hi << (128 - 16 * ox));
  } else if (w2 >= 8) {
    const int xc = min(max(xs, 0), w2 - 8);
    const int d = min(max(xs - xc, -8), 8);
    if (yin && d > -8 && d < 8) {
      typedef struct __attribute__((packed, aligned(2))) { u128 q; } U;
      v = reinterpret_cast<const U*>(plane + (size_t)y1 * w2 + xc)->q;
      if (d > 0) v >>= 16 * d;
      else if (d < 0) v <<= -16 * d;
    }
  } else {   // planes narrower than a window (unit-test sizes only)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int x1 = xs + i;
      const bool in = yin && (x1 >= 0) && (x1 < w2);
      const unsigned short bits = in ? __builtin_bit_cast(unsigned short, plane[(size_t)(in ? y1 : 0) * w2 + (in ? x1 : 0)]) : 0;
      v |= (u128)bits << (16 * i);
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) R[k] = (unsigned int)(v >> (32 * k));
}

template <int L, bool TILED>
__device__ __forceinline__ void level_coop(const _Float16* __restrict__ vol, size_t pix, bool valid, int h2, int w2,
                                           float2 c, int j, _Float16* __restrict__ tile_px) {
  const int h2l = h2 >> L, w2l = w2 >> L;
  const float sc = 1.0f / (float)(1 << L);
  const float x0 = c.x * sc, y0 = c.y * sc;
  const float fx0 = floorf(x0), fy0 = floorf(y0);
  const float dx = x0 - fx0, dy = y0 - fy0;
  const int xs = safe_int(fx0) - 3, ys = safe_int(fy0) - 3;
  unsigned int R[4] = {0, 0, 0, 0};
  if (valid) {
    if constexpr (TILED && L <= 1) {
      const size_t plane = (size_t)((w2l + 7) >> 3) * ((h2l + 7) >> 3) * 64;
      load_row_coop<true>(vol + pix * plane, h2l, w2l, xs, ys + j, R);
    } else {
      load_row_coop<false>(vol + pix * (size_t)(h2l * w2l), h2l, w2l, xs, ys + j, R);
    }
  }
  unsigned int Rn[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) Rn[k] = (unsigned int)__shfl_down((int)R[k], 1);      // row j+1 (lane j = 7: unused)
  const _Float16 w_nw = wcast<_Float16>(dx * dy);
  const _Float16 w_ne = wcast<_Float16>(dx * (1.0f - dy));
  const _Float16 w_sw = wcast<_Float16>((1.0f - dx) * dy);
  const _Float16 w_se = wcast<_Float16>((1.0f - dx) * (1.0f - dy));
  const half2v Wnw = {w_nw, w_nw}, Wne = {w_ne, w_ne}, Wsw = {w_sw, w_sw}, Wse = {w_se, w_se};
  _Float16 o[8];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    // taps (i, j), i = 2k, 2k+1:  ((s[i][j] w_se + s[i][j+1] w_sw) + s[i+1][j] w_ne) + s[i+1][j+1] w_nw
    const unsigned int q = (k < 3) ? __builtin_amdgcn_alignbit(R[k + 1], R[k], 16) : (R[3] >> 16);
    const unsigned int qn = (k < 3) ? __builtin_amdgcn_alignbit(Rn[k + 1], Rn[k], 16) : (Rn[3] >> 16);
    half2v acc = pin2(as_h2(R[k]) * Wse);
    acc = pin2(acc + pin2(as_h2(Rn[k]) * Wsw));
    acc = pin2(acc + pin2(as_h2(q) * Wne));
    acc = pin2(acc + pin2(as_h2(qn) * Wnw));
    o[2 * k] = acc[0];
    o[2 * k + 1] = acc[1];
  }
  if (j < 7) {
#pragma unroll
    for (int i = 0; i < 7; ++i) tile_px[49 * L + i * 7 + j] = o[i];
  }
}

constexpr int COOP_PXW = 8;      // pixels per wave: one pass of 8 (more, shorter waves measured fastest: 118 us vs 130 us at 64)

template <bool TILED, int PXW = COOP_PXW>
__global__ __launch_bounds__(256) void corr_pyramid_coop_kernel(
    const _Float16* __restrict__ v0, const _Float16* __restrict__ v1, const _Float16* __restrict__ v2,
    const _Float16* __restrict__ v3, const float* __restrict__ coords, _Float16* __restrict__ corr, int hw1, int h2,
    int w2, const long* __restrict__ slot) {
  __shared__ __attribute__((aligned(16))) _Float16 tiles[4][8 * 196];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int n = blockIdx.y;
  const size_t vn = slot ? (size_t)slot[n] : (size_t)n;          // the edge's planes: slot vn of a volume pool
  const int pw0 = (blockIdx.x * 4 + wv) * PXW;
  if (pw0 >= hw1) return;
  const int pp = lane >> 3, j = lane & 7;
  _Float16* tile = tiles[wv];
#pragma unroll 1
  for (int pass = 0; pass < PXW / 8; ++pass) {
    const int pb = pw0 + 8 * pass;
    if (pb >= hw1) break;
    const int p = pb + pp;
    const bool valid = p < hw1;
    const size_t pix = (size_t)n * hw1 + (valid ? p : pb);
    const size_t vpix = vn * hw1 + (valid ? p : pb);
    const float2 c = reinterpret_cast<const float2*>(coords)[pix];
    _Float16* tp = tile + pp * 196;
    level_coop<0, TILED>(v0, vpix, valid, h2, w2, c, j, tp);
    level_coop<1, TILED>(v1, vpix, valid, h2, w2, c, j, tp);
    level_coop<2, TILED>(v2, vpix, valid, h2, w2, c, j, tp);
    level_coop<3, TILED>(v3, vpix, valid, h2, w2, c, j, tp);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int npiece = min(8, hw1 - pb) * 49;                    // 8-byte pieces of the contiguous output run
    uint2* dst = reinterpret_cast<uint2*>(corr + ((size_t)n * hw1 + pb) * 196);
    const uint2* src = reinterpret_cast<const uint2*>(tile);
#pragma unroll
    for (int r = 0; r < 7; ++r) {
      const int idx = r * 64 + lane;
      if (idx < npiece) dst[idx] = src[idx];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// ---- lookup fused with corr_encoder[0] (src/droid_net.py:75-77: Conv2d(196, 128, 1) + ReLU on the looked-up features) --
// The cooperative lookup already assembles the 196 channels of 8 pixels per wave in LDS; here the four waves of a
// workgroup put their 4 x 8 pixels into ONE [32 pixels][208] tile (k padded to 13 MFMA steps, pad columns zero) and each
// wave multiplies it by ITS 32 of the 128 output channels on the matrix cores (v_mfma_f32_32x32x16_f16, weights = A
// operand, resident in 52 VGPRs for the whole workgroup: 6 passes = 192 pixels), adds the bias, applies ReLU and the
// [32 pixels][128] fp16 result leaves through LDS as one contiguous 8 KB run.  The 196-channel features (141 MB per
// update at the bench shape, written once and read once) never exist in HBM, and one launch is gone.
// Arithmetic = lookup (bit-exact, as above) -> fp16 features -> fp32-accumulated dot products -> + bias -> ReLU -> fp16,
// i.e. what gs_corr_lookup_pyramid + gs_conv1x1 compute, up to the summation order inside a dot product.
constexpr int ENC_K = 208;            // 196 padded to 13 k-steps of 16
constexpr int ENC_PASSES = 6;         // passes of 32 pixels per workgroup.  Measured at 75 x 60 x 80 (5 workgroups per CU):
                                      // 3: 120, 4: 122, 5: 131, 6: 119, 7: 128, 8: 135, 9: 144, 10: 150, 12: 123, 16: 127 us --
                                      // 8 left a second round of workgroups 11 % full, and chunks that are whole map rows
                                      // (5, 10 passes at w = 80) start every workgroup in the same column of its plane
typedef _Float16 enc_h8 __attribute__((ext_vector_type(8)));
typedef float enc_f16v __attribute__((ext_vector_type(16)));

template <bool TILED>
__global__ __launch_bounds__(256) void corr_lookup_enc_kernel(
    const _Float16* __restrict__ v0, const _Float16* __restrict__ v1, const _Float16* __restrict__ v2,
    const _Float16* __restrict__ v3, const float* __restrict__ coords, const _Float16* __restrict__ wpad,
    const float* __restrict__ bias, _Float16* __restrict__ y, int ys, int hw1, int h2, int w2,
    const long* __restrict__ slot) {
  __shared__ __attribute__((aligned(16))) _Float16 atile[32 * ENC_K];      // [pixel][k]
  __shared__ __attribute__((aligned(16))) _Float16 otile[32 * 128];        // [pixel][out channel]
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int n = blockIdx.y;
  const size_t vn = slot ? (size_t)slot[n] : (size_t)n;          // the edge's planes: slot vn of a volume pool
  const int p00 = blockIdx.x * 32 * ENC_PASSES;
  const int pp = lane >> 3, j = lane & 7;
  const int r = lane & 31, kgl = lane >> 5;
  // this wave's weights: A[m = 32 wv + r][k = 16 s + 8 kgl + e]
  enc_h8 wa[ENC_K / 16];
#pragma unroll
  for (int s = 0; s < ENC_K / 16; ++s)
    wa[s] = *reinterpret_cast<const enc_h8*>(wpad + (size_t)(32 * wv + r) * ENC_K + 16 * s + 8 * kgl);
  float bv[16];
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int e = 0; e < 4; ++e) bv[4 * g + e] = bias[32 * wv + 8 * g + 4 * kgl + e];
  for (int i = threadIdx.x; i < 32 * (ENC_K - 196); i += 256)              // pad columns: zero, never written again
    atile[(i / (ENC_K - 196)) * ENC_K + 196 + i % (ENC_K - 196)] = (_Float16)0.0f;
  __syncthreads();
#pragma unroll 1
  for (int pass = 0; pass < ENC_PASSES; ++pass) {
    const int pb = p00 + 32 * pass;
    if (pb >= hw1) break;                                     // (workgroup-uniform)
    {
      const int p = pb + 8 * wv + pp;
      const bool valid = p < hw1;
      const size_t pix = (size_t)n * hw1 + (valid ? p : pb);
      const size_t vpix = vn * hw1 + (valid ? p : pb);
      const float2 c = reinterpret_cast<const float2*>(coords)[pix];
      _Float16* tp = atile + (8 * wv + pp) * ENC_K;
      level_coop<0, TILED>(v0, vpix, valid, h2, w2, c, j, tp);
      level_coop<1, TILED>(v1, vpix, valid, h2, w2, c, j, tp);
      level_coop<2, TILED>(v2, vpix, valid, h2, w2, c, j, tp);
      level_coop<3, TILED>(v3, vpix, valid, h2, w2, c, j, tp);
    }
    __syncthreads();
    enc_f16v acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = bv[e];
#pragma unroll
    for (int s = 0; s < ENC_K / 16; ++s) {
      const enc_h8 b = *reinterpret_cast<const enc_h8*>(atile + r * ENC_K + 16 * s + 8 * kgl);   // B[k][col = pixel r]
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[s], b, acc, 0, 0, 0);
    }
    // D[row = channel 8 g + 4 kgl + e][col = pixel r]
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      typedef _Float16 h4 __attribute__((ext_vector_type(4)));
      h4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (_Float16)fmaxf(acc[4 * g + e], 0.0f);
      *reinterpret_cast<h4*>(otile + r * 128 + 32 * wv + 8 * g + 4 * kgl) = o;
    }
    __syncthreads();
    const int npx = min(32, hw1 - pb);
    for (int i = threadIdx.x; i < npx * 16; i += 256) {       // 16-byte pieces: 16 per pixel
      const int px = i >> 4, part = i & 15;
      *reinterpret_cast<uint4*>(y + ((size_t)n * hw1 + pb + px) * ys + 8 * part) =
          *reinterpret_cast<const uint4*>(otile + px * 128 + 8 * part);
    }
    // (the next pass writes atile only after this pass's reads -- ordered by the barrier above -- and otile only after
    // its own first barrier, which every wave reaches after these stores' LDS reads)
  }
}

// Sample one level for one pixel; writes 49 taps to out[(i*7+j)*plane] (plane == 0: `out` is a
// 49-entry register array).
template <typename T>
__device__ __forceinline__ void lookup_r3(const T* __restrict__ slice, int h2, int w2,
                                          float x0, float y0, T* __restrict__ out, size_t plane) {
  const float fx0 = floorf(x0), fy0 = floorf(y0);
  const float dx = x0 - fx0, dy = y0 - fy0;
  const int xs = safe_int(fx0) - 3, ys = safe_int(fy0) - 3;
  const T w_nw = wcast<T>(dx * dy);
  const T w_ne = wcast<T>(dx * (1.0f - dy));
  const T w_sw = wcast<T>((1.0f - dx) * dy);
  const T w_se = wcast<T>((1.0f - dx) * (1.0f - dy));

  T s[8][8];   // s[i][j]: i = x offset, j = y offset
  load_window<T>(slice, h2, w2, xs, ys, s);
#pragma unroll
  for (int i = 0; i < 7; ++i) {
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      // contributions arrive in loop order (i,j), (i,j+1), (i+1,j), (i+1,j+1)
      T c = mul_r(s[i][j], w_se);
      c = add_r(c, mul_r(s[i][j + 1], w_sw));
      c = add_r(c, mul_r(s[i + 1][j], w_ne));
      c = add_r(c, mul_r(s[i + 1][j + 1], w_nw));
      out[plane ? (size_t)(i * 7 + j) * plane : (size_t)(i * 7 + j)] = c;
    }
  }
}

// NHWC emission of one level: the pixel's 196 channels are contiguous (392 B, 8-byte aligned),
// so taps are packed four at a time into 8-byte stores (49 stores per pixel instead of 196
// two-byte ones, i.e. 4x fewer write requests at the L2); up to 3 taps carry over to the next level.
template <typename T, int L>
__device__ __forceinline__ void emit_nhwc(const T (&lv)[49], T (&carry)[4], T* __restrict__ out) {
  constexpr int pend = (49 * L) % 4;
  constexpr int total = pend + 49;
  constexpr int npk = total / 4;
  T tmp[52];
#pragma unroll
  for (int k = 0; k < pend; ++k) tmp[k] = carry[k];
#pragma unroll
  for (int k = 0; k < 49; ++k) tmp[pend + k] = lv[k];
  T* base = out + 49 * L - pend;
#pragma unroll
  for (int q = 0; q < npk; ++q) {
    struct alignas(4 * sizeof(T)) Pack { T v[4]; } pk;
#pragma unroll
    for (int k = 0; k < 4; ++k) pk.v[k] = tmp[4 * q + k];
    *reinterpret_cast<Pack*>(base + 4 * q) = pk;
  }
#pragma unroll
  for (int k = 0; k < total % 4; ++k) carry[k] = tmp[4 * npk + k];
}

template <typename T, int L>
__device__ __forceinline__ void level_nhwc(const T* __restrict__ vol, size_t pix, int h2, int w2, float2 c,
                                           T (&carry)[4], T* __restrict__ out) {
  const int h2l = h2 >> L, w2l = w2 >> L;
  const float sc = 1.0f / (float)(1 << L);
  T lv[49];
  lookup_r3<T>(vol + pix * (size_t)(h2l * w2l), h2l, w2l, c.x * sc, c.y * sc, lv, 0);
  emit_nhwc<T, L>(lv, carry, out);
}

// ---- fused 4-level pyramid lookup: coords [n,h1,w1,2] -> corr [n,4*49,h1,w1] ------------
// NHWC=true writes the same logical tensor with channels-last strides ([n,h1,w1,196] in
// memory), the layout MIOpen's NHWC convolutions of the update operator consume directly.
template <typename T, bool NHWC>
__global__ __launch_bounds__(256) void corr_pyramid_kernel(
    const T* __restrict__ v0, const T* __restrict__ v1, const T* __restrict__ v2, const T* __restrict__ v3,
    const float* __restrict__ coords, T* __restrict__ corr, int hw1, int h2, int w2) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  const int n = blockIdx.y;
  if (p >= hw1) return;
  const size_t pix = (size_t)n * hw1 + p;
  const float2 c = reinterpret_cast<const float2*>(coords)[pix];
  if constexpr (NHWC) {
    T* out = corr + pix * 196;
    T carry[4];
    level_nhwc<T, 0>(v0, pix, h2, w2, c, carry, out);
    level_nhwc<T, 1>(v1, pix, h2, w2, c, carry, out);
    level_nhwc<T, 2>(v2, pix, h2, w2, c, carry, out);
    level_nhwc<T, 3>(v3, pix, h2, w2, c, carry, out);   // (3*49) % 4 + 49 = 52: no remainder
  } else {
    T* out = corr + (size_t)n * 196 * hw1 + p;
    const T* vols[4] = {v0, v1, v2, v3};
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      const int h2l = h2 >> l, w2l = w2 >> l;
      const float sc = 1.0f / (float)(1 << l);   // coords / 2**l (exact)
      lookup_r3<T>(vols[l] + pix * (size_t)(h2l * w2l), h2l, w2l, c.x * sc, c.y * sc,
                   out + (size_t)l * 49 * hw1, (size_t)hw1);
    }
  }
}

// ---- single level, reference ABI: coords [n,2,h1,w1] -> corr [n,7,7,h1,w1] ---------------
template <typename T>
__global__ __launch_bounds__(256) void corr_index_r3_kernel(
    const T* __restrict__ vol, const float* __restrict__ coords, T* __restrict__ corr,
    int hw1, int h2, int w2) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  const int n = blockIdx.y;
  if (p >= hw1) return;
  const float x0 = coords[((size_t)n * 2 + 0) * hw1 + p];
  const float y0 = coords[((size_t)n * 2 + 1) * hw1 + p];
  lookup_r3<T>(vol + ((size_t)n * hw1 + p) * (size_t)(h2 * w2), h2, w2, x0, y0,
               corr + (size_t)n * 49 * hw1 + p, (size_t)hw1);
}

// ---- generic radius (not used by GO-SLAM, kept for ABI completeness) ---------------------
template <typename T>
__global__ __launch_bounds__(256) void corr_index_generic_kernel(
    const T* __restrict__ vol, const float* __restrict__ coords, T* __restrict__ corr,
    int hw1, int h2, int w2, int r) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  const int n = blockIdx.y;
  if (p >= hw1) return;
  const int rd = 2 * r + 1;
  const float x0 = coords[((size_t)n * 2 + 0) * hw1 + p];
  const float y0 = coords[((size_t)n * 2 + 1) * hw1 + p];
  const float fx0 = floorf(x0), fy0 = floorf(y0);
  const float dx = x0 - fx0, dy = y0 - fy0;
  const T* slice = vol + ((size_t)n * hw1 + p) * (size_t)(h2 * w2);
  T* out = corr + (size_t)n * rd * rd * hw1 + p;
  for (int c = 0; c < rd * rd; ++c) out[(size_t)c * hw1] = (T)0;
  for (int i = 0; i < rd + 1; ++i) {
    for (int j = 0; j < rd + 1; ++j) {
      const int x1 = safe_int(fx0) - r + i, y1 = safe_int(fy0) - r + j;
      if (y1 >= 0 && y1 < h2 && x1 >= 0 && x1 < w2) {
        const T s = slice[(size_t)y1 * w2 + x1];
        if (i > 0 && j > 0) { T* o = out + (size_t)((i - 1) * rd + (j - 1)) * hw1; *o = add_r(*o, mul_r(s, wcast<T>(dx * dy))); }
        if (i > 0 && j < rd) { T* o = out + (size_t)((i - 1) * rd + j) * hw1; *o = add_r(*o, mul_r(s, wcast<T>(dx * (1.0f - dy)))); }
        if (i < rd && j > 0) { T* o = out + (size_t)(i * rd + (j - 1)) * hw1; *o = add_r(*o, mul_r(s, wcast<T>((1.0f - dx) * dy))); }
        if (i < rd && j < rd) { T* o = out + (size_t)(i * rd + j) * hw1; *o = add_r(*o, mul_r(s, wcast<T>((1.0f - dx) * (1.0f - dy)))); }
      }
    }
  }
}

// ---- backward wrt the volume (training only in the reference) ----------------------------
template <typename T>
__global__ __launch_bounds__(256) void corr_index_backward_kernel(
    const float* __restrict__ coords, const T* __restrict__ corr_grad, T* __restrict__ vol_grad,
    int hw1, int h2, int w2, int r) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  const int n = blockIdx.y;
  if (p >= hw1) return;
  const int rd = 2 * r + 1;
  const float x0 = coords[((size_t)n * 2 + 0) * hw1 + p];
  const float y0 = coords[((size_t)n * 2 + 1) * hw1 + p];
  const float fx0 = floorf(x0), fy0 = floorf(y0);
  const float dx = x0 - fx0, dy = y0 - fy0;
  T* slice = vol_grad + ((size_t)n * hw1 + p) * (size_t)(h2 * w2);
  const T* g = corr_grad + (size_t)n * rd * rd * hw1 + p;
  for (int i = 0; i < rd + 1; ++i) {
    for (int j = 0; j < rd + 1; ++j) {
      const int x1 = safe_int(fx0) - r + i, y1 = safe_int(fy0) - r + j;
      if (y1 >= 0 && y1 < h2 && x1 >= 0 && x1 < w2) {
        T acc = (T)0;
        if (i > 0 && j > 0) acc = add_r(acc, mul_r(g[(size_t)((i - 1) * rd + (j - 1)) * hw1], wcast<T>(dx * dy)));
        if (i > 0 && j < rd) acc = add_r(acc, mul_r(g[(size_t)((i - 1) * rd + j) * hw1], wcast<T>(dx * (1.0f - dy))));
        if (i < rd && j > 0) acc = add_r(acc, mul_r(g[(size_t)(i * rd + (j - 1)) * hw1], wcast<T>((1.0f - dx) * dy)));
        if (i < rd && j < rd) acc = add_r(acc, mul_r(g[(size_t)(i * rd + j) * hw1], wcast<T>((1.0f - dx) * (1.0f - dy))));
        T* o = slice + (size_t)y1 * w2 + x1;
        *o = add_r(*o, acc);
      }
    }
  }
}

template <typename T>
int launch_index_forward(const void* volume, const float* coords, void* corr, int n, int h1, int w1,
                         int h2, int w2, int r, hipStream_t st) {
  const int hw1 = h1 * w1;
  dim3 grid(gs_cdiv(hw1, 256), n), block(256);
  if (r == 3)
    corr_index_r3_kernel<T><<<grid, block, 0, st>>>((const T*)volume, coords, (T*)corr, hw1, h2, w2);
  else
    corr_index_generic_kernel<T><<<grid, block, 0, st>>>((const T*)volume, coords, (T*)corr, hw1, h2, w2, r);
  GS_CHECK_LAUNCH("corr_index_forward");
  return GS_OK;
}

template <typename T>
int launch_index_backward(const float* coords, const void* g, void* vg, int n, int h1, int w1, int h2,
                          int w2, int r, hipStream_t st) {
  const int hw1 = h1 * w1;
  dim3 grid(gs_cdiv(hw1, 256), n), block(256);
  corr_index_backward_kernel<T><<<grid, block, 0, st>>>(coords, (const T*)g, (T*)vg, hw1, h2, w2, r);
  GS_CHECK_LAUNCH("corr_index_backward");
  return GS_OK;
}

template <typename T>
int launch_pyramid(const void* v0, const void* v1, const void* v2, const void* v3, const float* coords,
                   void* corr, int n, int h1, int w1, int h2, int w2, int nhwc, int layout, hipStream_t st,
                   const long* slot = nullptr) {
  const int hw1 = h1 * w1;
  dim3 grid(gs_cdiv(hw1, 256), n), block(256);
  if constexpr (sizeof(T) == 2) {
    if (nhwc) {          // production path: wave-cooperative kernel, either volume layout
      GS_REQUIRE(layout != GS_CORR_TILE8 || w2 % 16 == 0, "corr_lookup_pyramid: tile8 needs w2 %% 16 == 0");
      dim3 cgrid(gs_cdiv(hw1, 4 * COOP_PXW), n);
      const _Float16 *a = (const _Float16*)v0, *b = (const _Float16*)v1, *c = (const _Float16*)v2, *d = (const _Float16*)v3;
      if (layout == GS_CORR_TILE8)
        corr_pyramid_coop_kernel<true><<<cgrid, block, 0, st>>>(a, b, c, d, coords, (_Float16*)corr, hw1, h2, w2, slot);
      else
        corr_pyramid_coop_kernel<false><<<cgrid, block, 0, st>>>(a, b, c, d, coords, (_Float16*)corr, hw1, h2, w2, slot);
      GS_CHECK_LAUNCH("corr_lookup_pyramid");
      return GS_OK;
    }
  }
  if (layout == GS_CORR_TILE8 || slot) {
    gs_set_error("corr_lookup_pyramid: the tile8 layout / a volume pool is served by the fp16 channels_last kernel only");
    return GS_ERR_UNSUPPORTED;
  }
  if (nhwc)
    corr_pyramid_kernel<T, true><<<grid, block, 0, st>>>((const T*)v0, (const T*)v1, (const T*)v2, (const T*)v3,
                                                         coords, (T*)corr, hw1, h2, w2);
  else
    corr_pyramid_kernel<T, false><<<grid, block, 0, st>>>((const T*)v0, (const T*)v1, (const T*)v2, (const T*)v3,
                                                          coords, (T*)corr, hw1, h2, w2);
  GS_CHECK_LAUNCH("corr_lookup_pyramid");
  return GS_OK;
}

}  // namespace

extern "C" int gs_corr_index_forward(const void* volume, const float* coords, void* corr, int n, int h1,
                                     int w1, int h2, int w2, int radius, int dtype, gs_stream_t stream) {
  GS_REQUIRE(volume && coords && corr, "corr_index_forward: null pointer");
  GS_REQUIRE(n >= 0 && h1 > 0 && w1 > 0 && h2 > 0 && w2 > 0 && radius >= 0, "corr_index_forward: bad shape");
  if (n == 0) return GS_OK;
  GS_REQUIRE(n <= 65535, "corr_index_forward: n=%d exceeds grid.y limit", n);
  hipStream_t st = (hipStream_t)stream;
  switch (dtype) {
    case GS_F16: return launch_index_forward<_Float16>(volume, coords, corr, n, h1, w1, h2, w2, radius, st);
    case GS_F32: return launch_index_forward<float>(volume, coords, corr, n, h1, w1, h2, w2, radius, st);
    case GS_F64: return launch_index_forward<double>(volume, coords, corr, n, h1, w1, h2, w2, radius, st);
  }
  gs_set_error("corr_index_forward: unsupported dtype %d", dtype);
  return GS_ERR_UNSUPPORTED;
}

extern "C" int gs_corr_index_backward(const float* coords, const void* corr_grad, void* volume_grad, int n,
                                      int h1, int w1, int h2, int w2, int radius, int dtype,
                                      gs_stream_t stream) {
  GS_REQUIRE(coords && corr_grad && volume_grad, "corr_index_backward: null pointer");
  GS_REQUIRE(n >= 0 && h1 > 0 && w1 > 0 && h2 > 0 && w2 > 0 && radius >= 0, "corr_index_backward: bad shape");
  if (n == 0) return GS_OK;
  GS_REQUIRE(n <= 65535, "corr_index_backward: n=%d exceeds grid.y limit", n);
  hipStream_t st = (hipStream_t)stream;
  switch (dtype) {
    case GS_F16: return launch_index_backward<_Float16>(coords, corr_grad, volume_grad, n, h1, w1, h2, w2, radius, st);
    case GS_F32: return launch_index_backward<float>(coords, corr_grad, volume_grad, n, h1, w1, h2, w2, radius, st);
    case GS_F64: return launch_index_backward<double>(coords, corr_grad, volume_grad, n, h1, w1, h2, w2, radius, st);
  }
  gs_set_error("corr_index_backward: unsupported dtype %d", dtype);
  return GS_ERR_UNSUPPORTED;
}

extern "C" int gs_corr_lookup_enc(const void* vol0, const void* vol1, const void* vol2, const void* vol3,
                                  const float* coords, const void* wpad, const float* bias, void* y, int y_stride, int n,
                                  int h1, int w1, int h2, int w2, int layout, gs_stream_t stream) {
  return gs_corr_lookup_enc_slots(vol0, vol1, vol2, vol3, nullptr, coords, wpad, bias, y, y_stride, n, h1, w1, h2, w2, layout,
                                  stream);
}

extern "C" int gs_corr_lookup_enc_slots(const void* vol0, const void* vol1, const void* vol2, const void* vol3,
                                        const int64_t* slot, const float* coords, const void* wpad, const float* bias,
                                        void* y, int y_stride, int n, int h1, int w1, int h2, int w2, int layout,
                                        gs_stream_t stream) {
  GS_REQUIRE(layout == GS_CORR_ROWMAJOR || layout == GS_CORR_TILE8, "corr_lookup_enc: unknown layout %d", layout);
  GS_REQUIRE(vol0 && vol1 && vol2 && vol3 && coords && wpad && bias && y, "corr_lookup_enc: null pointer");
  GS_REQUIRE(n >= 0 && h1 > 0 && w1 > 0 && (h2 >> 3) > 0 && (w2 >> 3) > 0, "corr_lookup_enc: bad shape");
  GS_REQUIRE(y_stride >= 128 && y_stride % 8 == 0, "corr_lookup_enc: y_stride must be >= 128 and a multiple of 8");
  GS_REQUIRE(layout == GS_CORR_ROWMAJOR || (w2 % 16 == 0), "corr_lookup_enc: tile8 needs w2 %% 16 == 0");
  if (n == 0) return GS_OK;
  GS_REQUIRE(n <= 65535, "corr_lookup_enc: n=%d exceeds grid.y limit", n);
  const int hw1 = h1 * w1;
  dim3 grid(gs_cdiv(hw1, 32 * ENC_PASSES), n);
  if (layout == GS_CORR_TILE8)
    corr_lookup_enc_kernel<true><<<grid, 256, 0, (hipStream_t)stream>>>(
        (const _Float16*)vol0, (const _Float16*)vol1, (const _Float16*)vol2, (const _Float16*)vol3, coords,
        (const _Float16*)wpad, bias, (_Float16*)y, y_stride, hw1, h2, w2, (const long*)slot);
  else
    corr_lookup_enc_kernel<false><<<grid, 256, 0, (hipStream_t)stream>>>(
        (const _Float16*)vol0, (const _Float16*)vol1, (const _Float16*)vol2, (const _Float16*)vol3, coords,
        (const _Float16*)wpad, bias, (_Float16*)y, y_stride, hw1, h2, w2, (const long*)slot);
  GS_CHECK_LAUNCH("corr_lookup_enc");
  return GS_OK;
}

extern "C" int gs_corr_lookup_pyramid(const void* vol0, const void* vol1, const void* vol2, const void* vol3,
                                      const float* coords, void* corr, int n, int h1, int w1, int h2, int w2,
                                      int radius, int dtype, int channels_last, int layout, gs_stream_t stream) {
  return gs_corr_lookup_pyramid_slots(vol0, vol1, vol2, vol3, nullptr, coords, corr, n, h1, w1, h2, w2, radius, dtype,
                                      channels_last, layout, stream);
}

extern "C" int gs_corr_lookup_pyramid_slots(const void* vol0, const void* vol1, const void* vol2, const void* vol3,
                                            const int64_t* slot, const float* coords, void* corr, int n, int h1, int w1,
                                            int h2, int w2, int radius, int dtype, int channels_last, int layout,
                                            gs_stream_t stream) {
  GS_REQUIRE(layout == GS_CORR_ROWMAJOR || layout == GS_CORR_TILE8, "corr_lookup_pyramid: unknown layout %d", layout);
  GS_REQUIRE(vol0 && vol1 && vol2 && vol3 && coords && corr, "corr_lookup_pyramid: null pointer");
  GS_REQUIRE(radius == 3, "corr_lookup_pyramid: only radius 3 (the reference's value) is supported");
  GS_REQUIRE(n >= 0 && h1 > 0 && w1 > 0 && (h2 >> 3) > 0 && (w2 >> 3) > 0, "corr_lookup_pyramid: bad shape");
  if (n == 0) return GS_OK;
  GS_REQUIRE(n <= 65535, "corr_lookup_pyramid: n=%d exceeds grid.y limit", n);
  hipStream_t st = (hipStream_t)stream;
  switch (dtype) {
    case GS_F16: return launch_pyramid<_Float16>(vol0, vol1, vol2, vol3, coords, corr, n, h1, w1, h2, w2, channels_last, layout, st, (const long*)slot);
    case GS_F32: return launch_pyramid<float>(vol0, vol1, vol2, vol3, coords, corr, n, h1, w1, h2, w2, channels_last, layout, st, (const long*)slot);
  }
  gs_set_error("corr_lookup_pyramid: unsupported dtype %d", dtype);
  return GS_ERR_UNSUPPORTED;
}
