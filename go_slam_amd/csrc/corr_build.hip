// All-pairs correlation volume + 4-level average-pool pyramid in one pass
// (reference: src/modules/corr.py:26-41 CorrBlock.__init__ and :67-76 CorrBlock.corr, which run a
// batched fp16 GEMM and then three avg_pool2d passes that re-read the 46 MB/edge volume).
//
// The op is HBM-WRITE bound (2.656*HW^2 bytes vs 256*HW^2 flops per edge, AI ~ 96 flop/B), so the
// kernel is organised around writing every byte exactly once:
//   corr_prep_kernel   : [n,128,HW] -> channels-last [n,HW,128] fp16, scaled by 1/4 (corr.py:71-72),
//                        so both MFMA operands are K-contiguous 16-byte fragments (both maps, one launch)
//   corr_volume_kernel : one workgroup = 64 source pixels (p1) x 4 full rows of the target map
//                        (p2 = 4*w columns).  v_mfma_f32_32x32x16_f16 with A = f2 rows, B = f1 rows
//                        (so each lane ends up holding 4 consecutive p2 of one p1 -> 8-byte LDS
//                        writes); the fp16 tile is staged in LDS, written to level 0 with 16-byte
//                        coalesced stores, and the 2x2 / 4x4 / 8x8 pooled levels are produced from
//                        that LDS tile (each level from the fp16-rounded level below, exactly as
//                        avg_pool2d on a half tensor does) -- levels 0 and 1 are never read back.
//   corr_pool3_kernel  : the 8x8 level from level 2 (0.3 % of the bytes)
#include "common.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float float16v __attribute__((ext_vector_type(16)));

constexpr int KDIM = 128;
constexpr int BM = 64;          // source pixels per workgroup: every target-row fragment read from L2 feeds two MFMAs
                                // (at 32 x 8 rows the operand reads from L2 were 3x the bytes written, ~10 TB/s)
constexpr int MT = BM / 32;     // 32-pixel MFMA column blocks per workgroup
constexpr int ROWS = 4;         // target rows per workgroup (covers the 2x2 and 4x4 pooling blocks; 52 KB of LDS ->
                                // 3 workgroups per CU; the 8x8 level is pooled from level 2 by corr_pool3_kernel)
constexpr int MAXT = 3;         // n-tiles (32 columns) per wave: 4*w/32/4 <= 3  <=>  w <= 96

// Pixels a feature map is padded to in the workspace: whole 64-pixel source blocks and whole target-row blocks
// (+ one 32-pixel tile when a target-row block is not a whole number of tiles -- w % 8 == 4, EuRoC's 40 x 60 maps: the
// last n-tile of a block then reads up to 16 pixels past it)
__host__ __device__ inline int padded_pixels(int h, int w) {
  const int a = (h * w + BM - 1) / BM * BM, b = (h + ROWS - 1) / ROWS * ROWS * w;
  return ((a > b ? a : b) + 31) / 32 * 32 + (((ROWS * w) & 31) ? 32 : 0);
}

// [n,128,HW] -> MFMA-fragment order, x 1/4, for both feature maps in one launch (blockIdx.z = map * n + edge).
// Fragment order: [32-pixel tile][k-step ks (16 channels)][lane][8 halfs], lane = 32 * hi + r holding channels
// 16 ks + 8 hi .. + 8 of pixel 32 tile + r -- exactly the A / B operand of v_mfma_f32_32x32x16_f16, so that a wave
// fetches one operand of one k-step with ONE fully coalesced 1 KB load straight into the fragment registers.
// Pixels in [HW, padded) are written as zeros.  A workgroup handles 64 pixels x 128 channels: 128-byte row
// segments in (4 B per lane), 512-byte runs out (16 B per lane).
__global__ __launch_bounds__(256) void corr_prep_kernel(const _Float16* __restrict__ in1, const _Float16* __restrict__ in2,
                                                        _Float16* __restrict__ out1, _Float16* __restrict__ out2,
                                                        int n, int hw, int padded) {
  typedef _Float16 half2v __attribute__((ext_vector_type(2)));
  __shared__ _Float16 t[KDIM][64 + 2];
  const int which = blockIdx.z / n, e = blockIdx.z - which * n;
  const _Float16* in = (which ? in2 : in1) + (size_t)e * KDIM * hw;
  _Float16* out = (which ? out2 : out1) + (size_t)e * padded * KDIM;
  const int p0 = blockIdx.x * 64;
  const int pp = (threadIdx.x & 31) * 2, cr = threadIdx.x >> 5;
  const bool pair_ok = ((hw & 1) == 0) && (p0 + pp + 1 < hw);
  if ((hw & 1) == 0) {
    // even maps (every map the tracker builds): a lane's pixel pair is inside the map or outside it as a whole, so all
    // sixteen of its loads go out unconditionally from a clamped address and are zeroed afterwards (behind the `pair_ok`
    // branch each load was waited for before the next was issued: sixteen dependent round trips per workgroup)
    const int pq = min(p0 + pp, hw - 2);
    const bool in_map = p0 + pp < hw;
    half2v v[KDIM / 8];
#pragma unroll
    for (int k = 0; k < KDIM / 8; ++k) v[k] = *reinterpret_cast<const half2v*>(in + (size_t)(cr + 8 * k) * hw + pq);
#pragma unroll
    for (int k = 0; k < KDIM / 8; ++k) *reinterpret_cast<half2v*>(&t[cr + 8 * k][pp]) = in_map ? v[k] : half2v{0, 0};
  } else
#pragma unroll 4
  for (int c = cr; c < KDIM; c += 8) {
    const _Float16* src = in + (size_t)c * hw + p0 + pp;
    half2v v = {0, 0};
    if (pair_ok) v = *reinterpret_cast<const half2v*>(src);
    else {
      if (p0 + pp < hw) v[0] = src[0];
      if (p0 + pp + 1 < hw) v[1] = src[1];
    }
    *reinterpret_cast<half2v*>(&t[c][pp]) = v;
  }
  __syncthreads();
  const int px = threadIdx.x & 63;
  const int p = p0 + px;
  if (p >= padded) return;
  for (int chunk = threadIdx.x >> 6; chunk < 16; chunk += 4) {
    half8 o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = t[8 * chunk + k][px] / (_Float16)4.0f;
    *reinterpret_cast<half8*>(out + ((((size_t)(p >> 5) * 8 + (chunk >> 1)) * 64) + 32 * (chunk & 1) + (p & 31)) * 8) = o;
  }
}

__device__ __forceinline__ half8 ld8(const _Float16* p) { return *reinterpret_cast<const half8*>(p); }

__global__ __launch_bounds__(256, 3) void corr_volume_kernel(
    const _Float16* __restrict__ f1t, const _Float16* __restrict__ f2t, _Float16* __restrict__ v0,
    _Float16* __restrict__ v1, _Float16* __restrict__ v2, int h, int w, int tiled, int ntiles, int padded,
    const long* __restrict__ oslot) {
  extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
  const int hw = h * w;
  const int BN = ROWS * w;                 // columns of the tile: a multiple of 32 when w % 8 == 0, of 16 when w % 8 == 4
  const int ntile = (BN + 31) / 32;        // (then the last n-tile is half full; its upper 16 columns are never stored)
  const int LD0 = 32 * ntile + 8;          // LDS row strides (halfs)
  const int LD1 = (ROWS / 2) * (w / 2) + 8;   // multiple of 8 halves when w % 16 == 0: 16-byte LDS stores
  _Float16* c0 = lds;                      // [BM][LD0]
  _Float16* c1 = c0 + BM * LD0;            // [BM][LD1]
  // Workgroups are dealt to the 8 XCDs round-robin by launch index, and each XCD has its own L2.  The tiles are
  // therefore renumbered so that one XCD runs CONSECUTIVE tiles (target-row block fastest, then source block, then
  // edge): the 2 / 4 / 3.2 workgroups that write pieces of the same 128-byte line of levels 0 / 1 / 2 then meet in
  // one L2 and the line goes to HBM once and whole, instead of as 64 / 32 / 40-byte masked writes from different L2s.
  const int per_xcd = gridDim.x >> 3;                  // the launch pads the grid to a multiple of 8
  const int tile = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (tile >= ntiles) return;
  const int gx = (h + ROWS - 1) / ROWS, gy = (hw + BM - 1) / BM;
  const int e = tile / (gx * gy);
  // where edge e's planes go: slot oslot[e] of the caller's volume pool (gs_corr_volume_pyramid_slots), else edge e
  const size_t eo = oslot ? (size_t)oslot[e] : (size_t)e;
  const int p1_0 = ((tile / gx) % gy) * BM;
  const int y2_0 = (tile % gx) * ROWS;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = lane & 31;
  const _Float16* A = f2t + (size_t)e * padded * KDIM;   // target pixels, fragment order
  const _Float16* B = f1t + (size_t)e * padded * KDIM;   // source pixels, fragment order

  // A operand (target pixels): the 8 k-step fragments of a 32-pixel tile are 8 coalesced 1 KB loads straight into
  // the fragment registers (corr_prep wrote the map in fragment order) -- no LDS stage, no wave barriers.  The first
  // tile is issued before anything else so that its round trip overlaps the B tile's.
  // (target pixel of lane (hi, r) in n-tile nt: q0 + 32 nt + r.  q0 = y2_0 * w is a multiple of 32 when w % 8 == 0 -- the
  // lane then reads lane-th piece of stored tile q0 / 32 + nt, one coalesced 1 KB run per wave -- and of 16 when w % 8 == 4:
  // the n-tile straddles two stored tiles, two 256-byte runs per half-wave)
  const int q0 = y2_0 * w;
  half8 areg[8];
  auto frag = [&](int nt, int ks) {
    const int q = q0 + 32 * nt + r;
    return ld8(A + (((size_t)(q >> 5) * 8 + ks) * 64 + (lane & 32) + (q & 31)) * 8);
  };
  if (wave < ntile) {
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) areg[ks] = frag(wave, ks);
  }
  // B operand (source pixels): the workgroup's 64 x 128 tile (two 8 KB fragment-ordered blocks, contiguous) is
  // copied to LDS once for the 4 waves; its fragments are re-read per tile rather than held in 64 VGPRs.  The
  // copy aliases the output tile c0, which is only written after the MFMA phase.
  _Float16* bstage = lds;
  {
    const _Float16* src = B + (size_t)p1_0 * KDIM;
    for (int idx = threadIdx.x; idx < BM * 16; idx += 256)
      *reinterpret_cast<half8*>(bstage + 8 * idx) = ld8(src + 8 * idx);
    __syncthreads();
  }
  float16v acc[MAXT][MT];
#pragma unroll
  for (int t = 0; t < MAXT; ++t)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][mt][i] = 0.f;
  // Rolling prefetch: as soon as the two MFMAs of k-step ks have consumed areg[ks], the same registers receive the
  // next tile's fragment for that k-step, so 8 loads stay in flight through the whole phase.
#pragma unroll
  for (int t = 0; t < MAXT; ++t) {
    const int nt = wave + 4 * t;
    if (nt < ntile) {
      const bool more = (t + 1 < MAXT) && (nt + 4 < ntile);
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const half8 af = areg[ks];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const half8 bfr = *reinterpret_cast<const half8*>(bstage + ((mt * 8 + ks) * 64 + lane) * 8);
          acc[t][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bfr, acc[t][mt], 0, 0, 0);
        }
        if (more) areg[ks] = frag(nt + 4, ks);
      }
    }
  }
  __syncthreads();                                     // every wave is done with the B tile before c0 is written
  // D[p2][p1]: lane -> p1 = 32*mt + (lane&31); reg q*4+k -> p2 = 32*nt + 8*q + 4*(lane>>5) + k
#pragma unroll
  for (int t = 0; t < MAXT; ++t) {
    const int nt = wave + 4 * t;
    if (nt < ntile) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        _Float16* row = c0 + (size_t)(32 * mt + r) * LD0 + 32 * nt + 4 * (lane >> 5);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          half4 pk;
#pragma unroll
          for (int k = 0; k < 4; ++k) pk[k] = (_Float16)acc[t][mt][q * 4 + k];
          *reinterpret_cast<half4*>(row + 8 * q) = pk;
        }
      }
    }
  }
  __syncthreads();
  const int rows_valid = min(ROWS, h - y2_0);          // target rows of this tile inside the map
  const int m_valid = min(BM, hw - p1_0);
  // ---- level 0: [e][p1][y2][x2], the tile is rows_valid*w contiguous halfs per p1
  if (tiled) {
    // tile8 layout: the plane is cut into 8x8-element (128-byte) tiles, tile (ty,tx) at ((ty*ntx+tx)*64,
    // row-major inside).  This workgroup's 4 target rows are the upper or lower half of tile row y2_0/8;
    // four consecutive lanes write one tile's 64 contiguous bytes.
    const int ntx = w >> 3;
    const size_t plane = (size_t)ntx * ((h + 7) >> 3) * 64;
    const int vec = ntx * ROWS;
    const int ybase = y2_0 & 7;
    for (int i = threadIdx.x; i < m_valid * vec; i += 256) {
      const int m = i / vec, d = i - m * vec;
      const int tx = d / ROWS, y = d - tx * ROWS;
      if (y < rows_valid) {
        const half8 v = *reinterpret_cast<const half8*>(c0 + (size_t)m * LD0 + y * w + 8 * tx);
        *reinterpret_cast<half8*>(v0 + (eo * hw + p1_0 + m) * plane + ((size_t)(y2_0 >> 3) * ntx + tx) * 64 +
                                  8 * (ybase + y)) = v;
      }
    }
  } else {
    const int seg = rows_valid * w;                    // halfs per p1 row (multiple of 4 since w%4==0)
    const int vec = seg / 8;                           // 16-byte vectors (seg % 8 == 0 when w % 8 == 0)
    if ((seg & 7) == 0) {
      for (int i = threadIdx.x; i < m_valid * vec; i += 256) {
        const int m = i / vec, j = i - m * vec;
        const half8 v = *reinterpret_cast<const half8*>(c0 + (size_t)m * LD0 + 8 * j);
        *reinterpret_cast<half8*>(v0 + (eo * hw + p1_0 + m) * hw + (size_t)y2_0 * w + 8 * j) = v;
      }
    } else {
      const int vec4 = seg / 4;
      for (int i = threadIdx.x; i < m_valid * vec4; i += 256) {
        const int m = i / vec4, j = i - m * vec4;
        const half4 v = *reinterpret_cast<const half4*>(c0 + (size_t)m * LD0 + 4 * j);
        *reinterpret_cast<half4*>(v0 + (eo * hw + p1_0 + m) * hw + (size_t)y2_0 * w + 4 * j) = v;
      }
    }
  }
  // ---- level 1 (2x2 average of the fp16 level-0 values; ((a+b)+c)+d in fp32, x0.25, round)
  const int h1 = h >> 1, w1 = w >> 1, h2 = h >> 2, w2 = w >> 2;
  {
    const int r1 = ROWS / 2;
    if ((w1 & 7) == 0) {
      // 8 outputs per thread: two 32-byte LDS row segments in, one 16-byte LDS store and one 16-byte global
      // store out (the element-wise form below issues 2-byte global stores)
      const int pc = w1 >> 3;                        // 16-byte pieces per level-1 row
      // four threads per source pixel share its r1 * pc pieces (no integer divisions); in the tile8 layout the two
      // rows of a piece column are adjacent lanes, because there they are 32 contiguous bytes of one tile
      static_assert(BM * 4 == 256 && ROWS == 4, "level-1/2 thread mapping");
      const int m = threadIdx.x >> 2;
      for (int j = threadIdx.x & 3; j < r1 * pc; j += 4) {
        const int yy = tiled ? (j & 1) : (j >= pc), px = tiled ? (j >> 1) : (j - yy * pc);
        const _Float16* s = c0 + (size_t)m * LD0 + (2 * yy) * w + 16 * px;
        const half8 a0 = *reinterpret_cast<const half8*>(s), a1 = *reinterpret_cast<const half8*>(s + 8);
        const half8 b0 = *reinterpret_cast<const half8*>(s + w), b1 = *reinterpret_cast<const half8*>(s + w + 8);
        half8 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          o[k] = (_Float16)(((((float)a0[2 * k] + (float)a0[2 * k + 1]) + (float)b0[2 * k]) + (float)b0[2 * k + 1]) * 0.25f);
          o[4 + k] = (_Float16)(((((float)a1[2 * k] + (float)a1[2 * k + 1]) + (float)b1[2 * k]) + (float)b1[2 * k + 1]) * 0.25f);
        }
        *reinterpret_cast<half8*>(c1 + (size_t)m * LD1 + yy * w1 + 8 * px) = o;
        const int gy = (y2_0 >> 1) + yy;
        if (m < m_valid && gy < h1) {
          if (tiled) {
            const size_t plane1 = (size_t)pc * ((h1 + 7) >> 3) * 64;
            *reinterpret_cast<half8*>(v1 + (eo * hw + p1_0 + m) * plane1 + ((size_t)(gy >> 3) * pc + px) * 64 + (gy & 7) * 8) = o;
          } else {
            *reinterpret_cast<half8*>(v1 + (eo * hw + p1_0 + m) * ((size_t)h1 * w1) + (size_t)gy * w1 + 8 * px) = o;
          }
        }
      }
    } else {
    for (int i = threadIdx.x; i < BM * r1 * w1; i += 256) {
      const int m = i / (r1 * w1), rem = i - m * (r1 * w1);
      const int yy = rem / w1, xx = rem - yy * w1;
      const _Float16* s = c0 + (size_t)m * LD0 + (2 * yy) * w + 2 * xx;
      const float a = (float)s[0], b = (float)s[1], c = (float)s[w], d = (float)s[w + 1];
      const _Float16 o = (_Float16)((((a + b) + c) + d) * 0.25f);
      c1[(size_t)m * LD1 + yy * w1 + xx] = o;
      const int gy = (y2_0 >> 1) + yy;
      if (m < m_valid && gy < h1) {
        if (tiled) {
          const int ntx1 = w1 >> 3;
          const size_t plane1 = (size_t)ntx1 * ((h1 + 7) >> 3) * 64;
          v1[(eo * hw + p1_0 + m) * plane1 + ((size_t)(gy >> 3) * ntx1 + (xx >> 3)) * 64 + (gy & 7) * 8 + (xx & 7)] = o;
        } else {
          v1[(eo * hw + p1_0 + m) * ((size_t)h1 * w1) + (size_t)gy * w1 + xx] = o;
        }
      }
    }
    }
  }
  __syncthreads();
  // ---- level 2 from the fp16 level-1 tile (one row per workgroup)
  const int gy2 = y2_0 >> 2;
  if ((w2 & 3) == 0 && (w1 & 7) == 0) {
    // 4 outputs per thread: two 16-byte LDS row segments in, one 8-byte global store out
    const int m = threadIdx.x >> 2;
    if (m < m_valid && gy2 < h2) {
      for (int j = threadIdx.x & 3; j < (w2 >> 2); j += 4) {
        const _Float16* s = c1 + (size_t)m * LD1 + 8 * j;
        const half8 a = *reinterpret_cast<const half8*>(s), b = *reinterpret_cast<const half8*>(s + w1);
        half4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k)
          o[k] = (_Float16)(((((float)a[2 * k] + (float)a[2 * k + 1]) + (float)b[2 * k]) + (float)b[2 * k + 1]) * 0.25f);
        *reinterpret_cast<half4*>(v2 + (eo * hw + p1_0 + m) * ((size_t)h2 * w2) + (size_t)gy2 * w2 + 4 * j) = o;
      }
    }
  } else {
    for (int i = threadIdx.x; i < BM * w2; i += 256) {
      const int m = i / w2, xx = i - m * w2;
      const _Float16* s = c1 + (size_t)m * LD1 + 2 * xx;
      const float a = (float)s[0], b = (float)s[1], c = (float)s[w1], d = (float)s[w1 + 1];
      if (m < m_valid && gy2 < h2)
        v2[(eo * hw + p1_0 + m) * ((size_t)h2 * w2) + (size_t)gy2 * w2 + xx] = (_Float16)((((a + b) + c) + d) * 0.25f);
    }
  }
}

// Level 3 = 2x2 average of the fp16 level-2 values (same rounding chain as avg_pool2d on a half tensor): 0.3 % of the
// pyramid's bytes, read back from L2 right after the volume kernel wrote them.
__global__ __launch_bounds__(256) void corr_pool3_kernel(const _Float16* __restrict__ v2, _Float16* __restrict__ v3,
                                                         long planes, int h2, int w2, int hw, const long* __restrict__ oslot) {
  const int h3 = h2 >> 1, w3 = w2 >> 1;
  const long total = planes * h3 * w3;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    long pl = i / (h3 * w3);
    const int rem = (int)(i - pl * (h3 * w3));
    if (oslot) { const long e = pl / hw; pl = oslot[e] * hw + (pl - e * hw); }      // the edge's slot in the volume pool
    const int y = rem / w3, x = rem - y * w3;
    const _Float16* s = v2 + pl * ((long)h2 * w2) + (long)(2 * y) * w2 + 2 * x;
    const float a = (float)s[0], b = (float)s[1], c = (float)s[w2], d = (float)s[w2 + 1];
    v3[pl * ((long)h3 * w3) + rem] = (_Float16)((((a + b) + c) + d) * 0.25f);
  }
}

}  // namespace

extern "C" size_t gs_corr_volume_workspace_bytes(int n, int dim, int h, int w) {
  if (n < 0 || dim != KDIM || h <= 0 || w <= 0) return 0;
  return 2 * gs_align((size_t)n * padded_pixels(h, w) * KDIM * 2) + 256;
}

extern "C" size_t gs_corr_level_elems(int h, int w, int level, int layout) {
  if (h <= 0 || w <= 0 || level < 0 || level > 3) return 0;
  const int hl = h >> level, wl = w >> level;
  if (layout == GS_CORR_TILE8 && level <= 1) return (size_t)((wl + 7) >> 3) * ((hl + 7) >> 3) * 64;
  return (size_t)hl * wl;
}

extern "C" int gs_corr_volume_pyramid(const void* fmap1, const void* fmap2, void* vol0, void* vol1, void* vol2,
                                      void* vol3, int n, int dim, int h, int w, int layout, void* workspace,
                                      size_t workspace_bytes, gs_stream_t stream) {
  return gs_corr_volume_pyramid_slots(fmap1, fmap2, vol0, vol1, vol2, vol3, nullptr, n, dim, h, w, layout, workspace,
                                      workspace_bytes, stream);
}

extern "C" int gs_corr_volume_pyramid_slots(const void* fmap1, const void* fmap2, void* vol0, void* vol1, void* vol2,
                                            void* vol3, const int64_t* out_slot, int n, int dim, int h, int w, int layout,
                                            void* workspace, size_t workspace_bytes, gs_stream_t stream) {
  GS_REQUIRE(layout == GS_CORR_ROWMAJOR || layout == GS_CORR_TILE8, "corr_volume_pyramid: unknown layout %d", layout);
  GS_REQUIRE(layout == GS_CORR_ROWMAJOR || w % 16 == 0, "corr_volume_pyramid: the tile8 layout needs w %% 16 == 0");
  GS_REQUIRE(fmap1 && fmap2 && vol0 && vol1 && vol2 && vol3, "corr_volume_pyramid: null pointer");
  GS_REQUIRE(dim == KDIM, "corr_volume_pyramid: feature dim %d (DROID uses 128)", dim);
  GS_REQUIRE(n >= 0 && h >= 8 && w >= 8, "corr_volume_pyramid: bad shape");
  GS_REQUIRE(w % 4 == 0 && w <= 32 * MAXT, "corr_volume_pyramid: map width %d must be a multiple of 4 and <= %d",
             w, 32 * MAXT);
  if (n == 0) return GS_OK;
  const size_t need = gs_corr_volume_workspace_bytes(n, dim, h, w);
  if (!workspace || workspace_bytes < need) {
    gs_set_error("corr_volume_pyramid: workspace too small (%zu < %zu)", workspace_bytes, need);
    return GS_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const int hw = h * w;
  _Float16* f1t = (_Float16*)gs_align((size_t)workspace);
  const int padded = padded_pixels(h, w);
  _Float16* f2t = f1t + gs_align((size_t)n * padded * KDIM * 2) / 2;
  corr_prep_kernel<<<dim3(gs_cdiv(padded, 64), 1, 2 * n), 256, 0, st>>>((const _Float16*)fmap1, (const _Float16*)fmap2,
                                                                        f1t, f2t, n, hw, padded);
  GS_CHECK_LAUNCH("corr_prep");
  const int BN = ROWS * w;
  size_t lds = (size_t)(BM * ((BN + 31) / 32 * 32 + 8) + BM * ((ROWS / 2) * (w / 2) + 8)) * 2;     // c0 and c1
  const size_t btile = (size_t)BM * KDIM * 2;                              // the B tile aliases the output tile
  if (lds < btile) lds = btile;
  static GsLdsLimit limit;
  if (int rc = limit.raise((const void*)corr_volume_kernel, 160 * 1024, "corr_volume")) return rc;
  const long ntiles = (long)gs_cdiv(h, ROWS) * gs_cdiv(hw, BM) * n;
  GS_REQUIRE(ntiles < (1L << 30), "corr_volume_pyramid: map too large");
  const int grid = (int)((ntiles + 7) / 8 * 8);
  corr_volume_kernel<<<grid, 256, lds, st>>>(f1t, f2t, (_Float16*)vol0, (_Float16*)vol1, (_Float16*)vol2, h, w,
                                             layout == GS_CORR_TILE8, (int)ntiles, padded, (const long*)out_slot);
  GS_CHECK_LAUNCH("corr_volume");
  if ((h >> 3) > 0 && (w >> 3) > 0) {
    const long planes = (long)n * hw, total = planes * (h >> 3) * (w >> 3);
    corr_pool3_kernel<<<(int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192), 256, 0, st>>>(
        (const _Float16*)vol2, (_Float16*)vol3, planes, h >> 2, w >> 2, hw, (const long*)out_slot);
    GS_CHECK_LAUNCH("corr_pool3");
  }
  return GS_OK;
}
