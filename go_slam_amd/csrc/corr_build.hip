// All-pairs correlation volume + 4-level average-pool pyramid in one pass
// (reference: src/modules/corr.py:26-41 CorrBlock.__init__ and :67-76 CorrBlock.corr, which run a
// batched fp16 GEMM and then three avg_pool2d passes that re-read the 46 MB/edge volume).
//
// The op is HBM-WRITE bound (2.656*HW^2 bytes vs 256*HW^2 flops per edge, AI ~ 96 flop/B), so the
// kernel is organised around writing every byte exactly once:
//   corr_prep_kernel   : [n,128,HW] -> channels-last [n,HW,128] fp16, scaled by 1/4 (corr.py:71-72),
//                        so both MFMA operands are K-contiguous 16-byte fragments (both maps, one launch)
//   corr_volume_kernel : one workgroup = 32 source pixels (p1) x 8 full rows of the target map
//                        (p2 = 8*w columns).  v_mfma_f32_32x32x16_f16 with A = f2 rows, B = f1 rows
//                        (so each lane ends up holding 4 consecutive p2 of one p1 -> 8-byte LDS
//                        writes); the fp16 tile is staged in LDS, written to level 0 with 16-byte
//                        coalesced stores, and the 2x2 / 4x4 / 8x8 pooled levels are produced from
//                        that LDS tile (each level from the fp16-rounded level below, exactly as
//                        avg_pool2d on a half tensor does) -- the volume is never read back.
#include "common.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float float16v __attribute__((ext_vector_type(16)));

constexpr int KDIM = 128;
constexpr int BM = 32;          // source pixels per workgroup (32: 55 KB of LDS -> 2 workgroups per CU, so one's
                                // store phase overlaps the other's MFMA phase; 64 left a single workgroup per CU)
constexpr int MT = BM / 32;     // 32-pixel MFMA column blocks per workgroup
constexpr int ROWS = 8;         // target rows per workgroup (covers one 8x8 pooling block row)
constexpr int MAXT = 5;         // n-tiles (32 columns) per wave: 8*w/32/4 <= 5  <=>  w <= 80

// [n,128,HW] -> [n,HW,128], x 1/4, for both feature maps in one launch (blockIdx.z = map * n + edge).  A workgroup
// transposes 64 pixels x 128 channels: 128-byte row segments in (4 B per lane), whole 256-byte pixel rows out (16 B per lane).
__global__ __launch_bounds__(256) void corr_prep_kernel(const _Float16* __restrict__ in1, const _Float16* __restrict__ in2,
                                                        _Float16* __restrict__ out1, _Float16* __restrict__ out2,
                                                        int n, int hw) {
  typedef _Float16 half2v __attribute__((ext_vector_type(2)));
  __shared__ _Float16 t[KDIM][64 + 2];
  const int which = blockIdx.z / n, e = blockIdx.z - which * n;
  const _Float16* in = (which ? in2 : in1) + (size_t)e * KDIM * hw;
  _Float16* out = (which ? out2 : out1) + (size_t)e * hw * KDIM;
  const int p0 = blockIdx.x * 64;
  const int pp = (threadIdx.x & 31) * 2, cr = threadIdx.x >> 5;
  const bool pair_ok = ((hw & 1) == 0) && (p0 + pp + 1 < hw);
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
  const int c8 = (threadIdx.x & 15) * 8;
  for (int px = threadIdx.x >> 4; px < 64; px += 16) {
    if (p0 + px >= hw) break;
    half8 o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = t[c8 + k][px] / (_Float16)4.0f;
    *reinterpret_cast<half8*>(out + (size_t)(p0 + px) * KDIM + c8) = o;
  }
}

__device__ __forceinline__ half8 ld8(const _Float16* p) { return *reinterpret_cast<const half8*>(p); }

__global__ __launch_bounds__(256, 3) void corr_volume_kernel(
    const _Float16* __restrict__ f1t, const _Float16* __restrict__ f2t, _Float16* __restrict__ v0,
    _Float16* __restrict__ v1, _Float16* __restrict__ v2, _Float16* __restrict__ v3, int h, int w, int tiled) {
  extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
  const int hw = h * w;
  const int BN = ROWS * w;                 // columns of the tile (multiple of 32)
  const int LD0 = BN + 8;                  // LDS row strides (halfs)
  const int LD1 = (ROWS / 2) * (w / 2) + 8;   // multiple of 8 halves when w % 16 == 0: 16-byte LDS stores
  const int LD2 = (ROWS / 4) * (w / 4) + 2;
  _Float16* c0 = lds;                      // [BM][LD0]
  _Float16* c1 = c0 + BM * LD0;            // [BM][LD1]
  _Float16* c2 = c0;                       // [BM][LD2]: level 2 is produced after the last read of c0 (barrier below)
  const int e = blockIdx.z;
  // target-row block fastest: workgroups that run together then write the SAME 32 source pixels' planes side by side
  // (32 x 9.6 KB contiguous per 8 workgroups at 60 x 80) instead of 1.3 KB pieces 9.6 KB apart across the whole volume
  const int p1_0 = blockIdx.y * BM;
  const int y2_0 = blockIdx.x * ROWS;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = lane & 31;
  const int ntile = BN / 32;
  const _Float16* A = f2t + (size_t)e * hw * KDIM;   // rows p2
  const _Float16* B = f1t + (size_t)e * hw * KDIM;   // rows p1

  // A operand, first tile: issued before anything else so that its round trip overlaps the B tile's (see below)
  const int srow = lane >> 4, schunk = lane & 15;
  half8 areg[8];
  auto issue_tile = [&](int t) {
    const int nt = wave + 4 * t;
    if (nt < ntile) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int p2 = min(y2_0 * w + 32 * nt + 4 * i + srow, hw - 1);
        areg[i] = ld8(A + (size_t)p2 * KDIM + 8 * schunk);
      }
    }
  };
  issue_tile(0);
  // B operand (source pixels): the workgroup's BM x 128 tile through LDS as well (coalesced loads, one copy for the 4
  // waves; see the A operand below).  Its fragments are re-read from LDS per tile rather than held in 32 VGPRs.
  _Float16* bstage = lds + 4 * (32 * KDIM);
  {
    for (int idx = threadIdx.x; idx < BM * 16; idx += 256) {
      const int row = idx >> 4, chunk = idx & 15;
      const int p1 = min(p1_0 + row, hw - 1);
      *reinterpret_cast<half8*>(bstage + row * KDIM + 8 * (chunk ^ (row & 15))) = ld8(B + (size_t)p1 * KDIM + 8 * chunk);
    }
    __syncthreads();
  }
  float16v acc[MAXT][MT];
#pragma unroll
  for (int t = 0; t < MAXT; ++t)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][mt][i] = 0.f;
  // A operand (target pixels): a fragment lane wants 16 bytes of ITS row, so loading fragments straight from memory
  // makes every load instruction touch 32 rows x 32 B (32 cache lines for 1 KB; the texture path, not HBM, bounded the
  // kernel at 0.17 of the write roofline).  Instead a wave fetches its 32 x 128 tile as 8 fully coalesced 1 KB loads
  // (lane -> row 4 i + lane / 16, 16-byte chunk lane % 16), parks it in a wave-private 8 KB LDS stage (chunk XOR row:
  // conflict-free both ways; the stage aliases the output tile c0, which is only written after the MFMA phase) and
  // reads the fragments back with ds_read_b128.  Tile t + 1 is in flight while tile t is multiplied.  (One register
  // buffer: with 3 workgroups per CU -- 168 VGPRs, 52 KB of LDS -- the other workgroups cover what is left of the latency.)
  _Float16* stage = lds + wave * (32 * KDIM);
#pragma unroll
  for (int t = 0; t < MAXT; ++t) {
    const int nt = wave + 4 * t;
    if (nt < ntile) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int R = 4 * i + srow;
        *reinterpret_cast<half8*>(stage + R * KDIM + 8 * (schunk ^ (R & 15))) = areg[i];
      }
    }
    if (t + 1 < MAXT) issue_tile(t + 1);               // the registers are free again: next tile flies during the MFMAs
    if (nt < ntile) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const int sw = 8 * ((2 * ks + (lane >> 5)) ^ (r & 15));
        const half8 af = *reinterpret_cast<const half8*>(stage + r * KDIM + sw);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const half8 bfr = *reinterpret_cast<const half8*>(bstage + (32 * mt + r) * KDIM + sw);
          acc[t][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bfr, acc[t][mt], 0, 0, 0);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
  __syncthreads();                                     // every wave is done with its stage before c0 is written
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
    // row-major inside).  This workgroup's 8 target rows are exactly tile row y2_0/8; consecutive lanes
    // write consecutive 16-byte pieces of it (piece d = 8*tx + y).
    const int ntx = w >> 3;
    const size_t plane = (size_t)ntx * ((h + 7) >> 3) * 64;
    const int vec = ntx * 8;
    for (int i = threadIdx.x; i < m_valid * vec; i += 256) {
      const int m = i / vec, d = i - m * vec;
      const int tx = d >> 3, y = d & 7;
      if (y < rows_valid) {
        const half8 v = *reinterpret_cast<const half8*>(c0 + (size_t)m * LD0 + y * w + 8 * tx);
        *reinterpret_cast<half8*>(v0 + ((size_t)e * hw + p1_0 + m) * plane + ((size_t)(y2_0 >> 3) * ntx) * 64 + 8 * d) = v;
      }
    }
  } else {
    const int seg = rows_valid * w;                    // halfs per p1 row (multiple of 4 since w%4==0)
    const int vec = seg / 8;                           // 16-byte vectors (seg % 8 == 0 when w % 8 == 0)
    if ((seg & 7) == 0) {
      for (int i = threadIdx.x; i < m_valid * vec; i += 256) {
        const int m = i / vec, j = i - m * vec;
        const half8 v = *reinterpret_cast<const half8*>(c0 + (size_t)m * LD0 + 8 * j);
        *reinterpret_cast<half8*>(v0 + ((size_t)e * hw + p1_0 + m) * hw + (size_t)y2_0 * w + 8 * j) = v;
      }
    } else {
      const int vec4 = seg / 4;
      for (int i = threadIdx.x; i < m_valid * vec4; i += 256) {
        const int m = i / vec4, j = i - m * vec4;
        const half4 v = *reinterpret_cast<const half4*>(c0 + (size_t)m * LD0 + 4 * j);
        *reinterpret_cast<half4*>(v0 + ((size_t)e * hw + p1_0 + m) * hw + (size_t)y2_0 * w + 4 * j) = v;
      }
    }
  }
  // ---- level 1 (2x2 average of the fp16 level-0 values; ((a+b)+c)+d in fp32, x0.25, round)
  const int h1 = h >> 1, w1 = w >> 1, h2 = h >> 2, w2 = w >> 2, h3 = h >> 3, w3 = w >> 3;
  {
    const int r1 = ROWS / 2;
    if ((w1 & 7) == 0) {
      // 8 outputs per thread: two 32-byte LDS row segments in, one 16-byte LDS store and one 16-byte global
      // store out (the element-wise form below issues 2-byte global stores)
      const int pc = w1 >> 3;                        // 16-byte pieces per level-1 row
      for (int i = threadIdx.x; i < BM * r1 * pc; i += 256) {
        const int m = i / (r1 * pc), rem = i - m * (r1 * pc);
        const int yy = rem / pc, px = rem - yy * pc;
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
            *reinterpret_cast<half8*>(v1 + ((size_t)e * hw + p1_0 + m) * plane1 + ((size_t)(gy >> 3) * pc + px) * 64 + (gy & 7) * 8) = o;
          } else {
            *reinterpret_cast<half8*>(v1 + ((size_t)e * hw + p1_0 + m) * ((size_t)h1 * w1) + (size_t)gy * w1 + 8 * px) = o;
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
          v1[((size_t)e * hw + p1_0 + m) * plane1 + ((size_t)(gy >> 3) * ntx1 + (xx >> 3)) * 64 + (gy & 7) * 8 + (xx & 7)] = o;
        } else {
          v1[((size_t)e * hw + p1_0 + m) * ((size_t)h1 * w1) + (size_t)gy * w1 + xx] = o;
        }
      }
    }
    }
  }
  __syncthreads();
  {
    const int r2 = ROWS / 4;
    for (int i = threadIdx.x; i < BM * r2 * w2; i += 256) {
      const int m = i / (r2 * w2), rem = i - m * (r2 * w2);
      const int yy = rem / w2, xx = rem - yy * w2;
      const _Float16* s = c1 + (size_t)m * LD1 + (2 * yy) * w1 + 2 * xx;
      const float a = (float)s[0], b = (float)s[1], c = (float)s[w1], d = (float)s[w1 + 1];
      const _Float16 o = (_Float16)((((a + b) + c) + d) * 0.25f);
      c2[(size_t)m * LD2 + yy * w2 + xx] = o;
      const int gy = (y2_0 >> 2) + yy;
      if (m < m_valid && gy < h2 && xx < w2)
        v2[((size_t)e * hw + p1_0 + m) * ((size_t)h2 * w2) + (size_t)gy * w2 + xx] = o;
    }
  }
  __syncthreads();
  {
    for (int i = threadIdx.x; i < BM * w3; i += 256) {
      const int m = i / w3, xx = i - m * w3;
      const _Float16* s = c2 + (size_t)m * LD2 + 2 * xx;
      const float a = (float)s[0], b = (float)s[1], c = (float)s[w2], d = (float)s[w2 + 1];
      const _Float16 o = (_Float16)((((a + b) + c) + d) * 0.25f);
      const int gy = y2_0 >> 3;
      if (m < m_valid && gy < h3)
        v3[((size_t)e * hw + p1_0 + m) * ((size_t)h3 * w3) + (size_t)gy * w3 + xx] = o;
    }
  }
}

}  // namespace

extern "C" size_t gs_corr_volume_workspace_bytes(int n, int dim, int h, int w) {
  if (n < 0 || dim != KDIM || h <= 0 || w <= 0) return 0;
  return 2 * gs_align((size_t)n * h * w * KDIM * 2) + 256;
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
  GS_REQUIRE(layout == GS_CORR_ROWMAJOR || layout == GS_CORR_TILE8, "corr_volume_pyramid: unknown layout %d", layout);
  GS_REQUIRE(layout == GS_CORR_ROWMAJOR || w % 16 == 0, "corr_volume_pyramid: the tile8 layout needs w %% 16 == 0");
  GS_REQUIRE(fmap1 && fmap2 && vol0 && vol1 && vol2 && vol3, "corr_volume_pyramid: null pointer");
  GS_REQUIRE(dim == KDIM, "corr_volume_pyramid: feature dim %d (DROID uses 128)", dim);
  GS_REQUIRE(n >= 0 && h >= 8 && w >= 8, "corr_volume_pyramid: bad shape");
  GS_REQUIRE(w % 8 == 0 && w <= 16 * MAXT, "corr_volume_pyramid: map width %d must be a multiple of 8 and <= %d",
             w, 16 * MAXT);
  if (n == 0) return GS_OK;
  GS_REQUIRE(n <= 32767, "corr_volume_pyramid: n=%d exceeds the grid.z limit", n);
  const size_t need = gs_corr_volume_workspace_bytes(n, dim, h, w);
  if (!workspace || workspace_bytes < need) {
    gs_set_error("corr_volume_pyramid: workspace too small (%zu < %zu)", workspace_bytes, need);
    return GS_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const int hw = h * w;
  _Float16* f1t = (_Float16*)gs_align((size_t)workspace);
  _Float16* f2t = f1t + gs_align((size_t)n * hw * KDIM * 2) / 2;
  corr_prep_kernel<<<dim3(gs_cdiv(hw, 64), 1, 2 * n), 256, 0, st>>>((const _Float16*)fmap1, (const _Float16*)fmap2, f1t,
                                                                    f2t, n, hw);
  GS_CHECK_LAUNCH("corr_prep");
  const int BN = ROWS * w;
  size_t lds = (size_t)(BM * (BN + 8) + BM * ((ROWS / 2) * (w / 2) + 8)) * 2;     // c0 (+ c2 inside it) and c1
  const size_t stages = (size_t)(4 * 32 + BM) * KDIM * 2;                   // operand stages (alias the output tile)
  if (lds < stages) lds = stages;
  static GsLdsLimit limit;
  if (int rc = limit.raise((const void*)corr_volume_kernel, 160 * 1024, "corr_volume")) return rc;
  GS_REQUIRE(gs_cdiv(hw, BM) <= 65535, "corr_volume_pyramid: map too large");
  dim3 grid(gs_cdiv(h, ROWS), gs_cdiv(hw, BM), n);
  corr_volume_kernel<<<grid, 256, lds, st>>>(f1t, f2t, (_Float16*)vol0, (_Float16*)vol1, (_Float16*)vol2,
                                             (_Float16*)vol3, h, w, layout == GS_CORR_TILE8);
  GS_CHECK_LAUNCH("corr_volume");
  return GS_OK;
}
