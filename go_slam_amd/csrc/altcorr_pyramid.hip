// The whole on-the-fly ("alt") correlation lookup of one chunk of edges in ONE launch, on the matrix cores (round 5).
//
// Replaces, per chunk of FactorGraph.update_lowmem (reference src/factor_graph.py:255-321 -> src/modules/corr.py:112-131):
// per pyramid level two feature gathers (`pyramid[0][:, ii]`, `pyramid[i][:, jj]`), the coordinate scaling, a launch of
// altcorr_forward_kernel (src/lib/altcorr_kernel.cu:27-149), a cast, a permute and the final cat -- ~40 launches and, in the
// round-4 trace of the 200-keyframe step, the #1 kernel (altcorr_forward_kernel: one wave per source pixel on v_dot2, 64 x
// 90 us = 5.8 ms = 0.10 of the fp32 vector roof).
//
// MI355X-first formulation.  The windows of neighbouring source pixels overlap almost entirely (a reprojection flow is
// smooth), so a 4 x 4 tile of source pixels needs the dot products of its 16 feature vectors with the UNION of their 8 x 8
// windows in fmap2 -- a [16 x 128] x [128 x (rows x 16)] product.  A wave owns one tile of one edge and walks the four
// levels:
//   * A operand: the tile's 16 rows of fmap1 (channels-last, 128 fp16) live in 16 VGPRs for the whole wave
//     (v_mfma_f32_16x16x32_f16: lane (q, m) holds channels 32 ks + 8 q .. + 7 of pixel m);
//   * per level the integer window origins (x0, y0) = floor(coords / 2^l) - 3 of the 16 pixels are reduced to their bounding
//     box; when the box is at most 16 columns x 16 rows wide (x0 / y0 spread <= 8: any flow whose local stretch is below
//     ~3.6x) every window ROW of the union is one N-tile: the B operand of a k-step is ONE 16-byte load per lane straight
//     from the channels-last map (16 positions x 64 contiguous bytes), zero outside the image, no LDS staging; 4 MFMAs per
//     row.  Each accumulator value that belongs to its pixel's own 8 x 8 window goes to a wave-private LDS tile
//     S[pixel][8][8] (4 KB) -- the other ~3/4 of the 16 x (rows x 16) products are the price of the dense formulation;
//   * windows too far apart for the box (depth discontinuities, wild coordinates): the level falls back, inside the same
//     wave, to the per-pixel path (lane = (window column, 16-channel chunk), v_dot2_f32_f16, DPP sums) that fills the same S;
//   * the bilinear blend of the four neighbouring taps (the reference's accumulation order) reads S and writes the level's
//     49 fp16 outputs per pixel into a wave-private stage [16][196]; after the fourth level the 392-byte pixel rows leave
//     as coalesced runs into out[edge][y][x][196] -- the NHWC layout the update operator's corr_encoder[0] reads.
// Features are indexed by ii / jj INSIDE the kernel (no gathered copies), and the launch order keeps all tiles of an edge on
// one XCD (block b -> XCD b % 8 is an observed property used for speed only) so that the edge's fmap2 pyramid (408 KB at
// 30 x 40) is fetched into one L2 instead of eight.
// Arithmetic: exact fp16 products, fp32 accumulation (matrix-core order on the main path, dot2 chain on the fallback),
// fp32 blend, one rounding to fp16 -- within half an fp16 ulp of altcorr_forward_kernel<half> either way.
#include "common.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float float4v __attribute__((ext_vector_type(4)));

constexpr int AC = 128;            // channels
constexpr int SP = 65;             // S: dwords per pixel (8 x 8 taps + 1: pixels fall into different banks)
constexpr int SS = 99;             // stage: dwords per pixel (196 halves = 98 dwords + 1)

struct AltPyrArgs {
  const _Float16* pyr[4];          // [N][H >> l][W >> l][128] fp16, channels-last
  const float* coords;             // [E][H][W][2]
  const int64_t* ii; const int64_t* jj;
  _Float16* out;                   // [E][H][W][196]
  int E, H, W;
};

__device__ __forceinline__ void wave_sync_lds_() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ int grp16_min(int v) {
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) v = min(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ int grp16_max(int v) {
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) v = max(v, __shfl_xor(v, o, 64));
  return v;
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void altcorr_pyramid_kernel(AltPyrArgs A,
                                                                                                        int tiles_x,
                                                                                                        int tiles,
                                                                                                        int blocks_per_edge) {
  __shared__ float s_all[4][16 * SP];
  __shared__ uint32_t stage_all[4][16 * SS];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int m = lane & 15, q = lane >> 4;
  // all tiles of an edge on one XCD: XCD x serves the edges e = 8 k + x
  const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
  const int e = (jb / blocks_per_edge) * 8 + xcd;
  const int tile = (jb % blocks_per_edge) * 4 + wv;
  if (e >= A.E || tile >= tiles) return;                  // (wave-uniform; no workgroup barrier below)
  float* S = s_all[wv];
  uint32_t* stage = stage_all[wv];
  _Float16* stage_h = reinterpret_cast<_Float16*>(stage);
  const int ty = tile / tiles_x, tx_ = tile - ty * tiles_x;
  // lanes beyond the image border stand in for the nearest pixel inside (the bounding box is unchanged, nothing is stored)
  const int py = min(ty * 4 + (m >> 2), A.H - 1), px = min(tx_ * 4 + (m & 3), A.W - 1);
  const int hw = A.H * A.W;
  const int p = py * A.W + px;
  const size_t fi = (size_t)A.ii[e], fj = (size_t)A.jj[e];
  const _Float16* f1p = A.pyr[0] + (fi * hw + p) * AC;
  half8 a[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) a[ks] = *reinterpret_cast<const half8*>(f1p + 32 * ks + 8 * q);
  const float2 c0 = reinterpret_cast<const float2*>(A.coords)[(size_t)e * hw + p];
#pragma unroll 1
  for (int l = 0; l < 4; ++l) {
    const int H2 = A.H >> l, W2 = A.W >> l;
    const float inv = 1.0f / (float)(1 << l);             // (coords / 2^l: exact)
    const float cx = c0.x * inv, cy = c0.y * inv;
    const float fx0 = floorf(cx), fy0 = floorf(cy);
    const float dx = cx - fx0, dy = cy - fy0;
    const int x0 = (int)fminf(fmaxf(fx0, -1.0e6f), 1.0e6f) - 3;
    const int y0 = (int)fminf(fmaxf(fy0, -1.0e6f), 1.0e6f) - 3;
    const int xb = grp16_min(x0), yb = grp16_min(y0);
    const int ncols = grp16_max(x0) - xb + 8, nrows = grp16_max(y0) - yb + 8;
    const _Float16* f2 = A.pyr[l] + fj * (size_t)(H2 * W2) * AC;
    if (ncols <= 16 && nrows <= 16) {                     // (wave-uniform: the four lane groups hold the same 16 pixels)
      // ---- the union window on the matrix cores: one window row (16 positions) per N-tile
      int xr[4], yr[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        xr[r] = __shfl(x0, 4 * q + r, 64) - xb;
        yr[r] = __shfl(y0, 4 * q + r, 64) - yb;
      }
      const int x = xb + m;
      const bool xin = x >= 0 && x < W2;
      const half8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
      auto load_row = [&](int j, half8 (&b)[4]) {
        const int y = yb + j;
        if (xin && y >= 0 && y < H2) {
          const _Float16* src = f2 + ((size_t)y * W2 + x) * AC + 8 * q;
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) b[ks] = *reinterpret_cast<const half8*>(src + 32 * ks);
        } else {
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) b[ks] = z8;
        }
      };
      half8 bc[4];
      load_row(0, bc);
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        if (j < nrows) {                                  // (uniform)
          half8 bn[4];
          if (j + 1 < nrows) load_row(j + 1, bn);         // the next row's operands are in flight behind this row's MFMAs
          float4v acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[ks], bc[ks], acc, 0, 0, 0);
          // C: column = lane & 15 = window position m of row j, row = 4 q + r = pixel
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int iy = j - yr[r], tx = m - xr[r];
            if ((unsigned)iy < 8u && (unsigned)tx < 8u) S[(4 * q + r) * SP + iy * 8 + tx] = acc[r];
          }
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) bc[ks] = bn[ks];
        }
      }
    } else {
      // ---- windows too far apart: per pixel, lane = (window column tx, 16-channel chunk cc)
      const int tx = lane >> 3, cc = lane & 7;
#pragma unroll 1
      for (int mi = 0; mi < 16; ++mi) {
        const int x0m = __shfl(x0, mi, 64), y0m = __shfl(y0, mi, 64), pm = __shfl(p, mi, 64);
        const _Float16* ap = A.pyr[0] + (fi * hw + pm) * AC + 16 * cc;
        const half8 a0 = *reinterpret_cast<const half8*>(ap), a1 = *reinterpret_cast<const half8*>(ap + 8);
        const int x = x0m + tx;
        const bool xin = x >= 0 && x < W2;
#pragma unroll 1
        for (int iy = 0; iy < 8; ++iy) {
          const int y = y0m + iy;
          float s = 0.f;
          if (xin && y >= 0 && y < H2) {
            const _Float16* bp = f2 + ((size_t)y * W2 + x) * AC + 16 * cc;
            const half8 b0 = *reinterpret_cast<const half8*>(bp), b1 = *reinterpret_cast<const half8*>(bp + 8);
#pragma unroll
            for (int k = 0; k < 8; k += 2) {
              const half2v u = {a0[k], a0[k + 1]}, v = {b0[k], b0[k + 1]};
              s = __builtin_amdgcn_fdot2(u, v, s, false);
            }
#pragma unroll
            for (int k = 0; k < 8; k += 2) {
              const half2v u = {a1[k], a1[k + 1]}, v = {b1[k], b1[k + 1]};
              s = __builtin_amdgcn_fdot2(u, v, s, false);
            }
          }
          s = s + gs_dpp<0xB1>(s);                        // sum of the 8 channel chunks (lanes 8 tx .. 8 tx + 7)
          s = s + gs_dpp<0x4E>(s);
          s = s + gs_dpp<0x141>(s);
          if (cc == 0) S[mi * SP + iy * 8 + tx] = s;
        }
      }
    }
    wave_sync_lds_();
    // ---- bilinear blend (altcorr_kernel.cu:95-130: the four taps around an output, in the reference's order);
    // lane (q, m): pixel m, window columns q and q + 4; output channel = 49 l + 7 tx + iy
    {
      const float w00 = (1.0f - dy) * (1.0f - dx), w01 = (1.0f - dy) * dx, w10 = dy * (1.0f - dx), w11 = dy * dx;
      const float* Sm = S + m * SP;
      _Float16* dst = stage_h + m * (2 * SS) + 49 * l;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int tx = q + 4 * h;
        if (tx < 7) {
          float prev = Sm[tx], prev_n = Sm[tx + 1];
#pragma unroll
          for (int iy = 1; iy < 8; ++iy) {
            const float s = Sm[iy * 8 + tx], s_n = Sm[iy * 8 + tx + 1];
            float o = prev * w00;
            o = fmaf(prev_n, w01, o);
            o = fmaf(s, w10, o);
            o = fmaf(s_n, w11, o);
            dst[7 * tx + (iy - 1)] = (_Float16)o;
            prev = s;
            prev_n = s_n;
          }
        }
      }
    }
    wave_sync_lds_();                                     // S is overwritten by the next level
  }
  // ---- the tile's 16 x 196 outputs: 392-byte pixel rows, pixels of a tile row adjacent in memory
  uint32_t* out32 = reinterpret_cast<uint32_t*>(A.out) + (size_t)e * hw * 98;
#pragma unroll 1
  for (int i = lane; i < 16 * 98; i += 64) {
    const int mi = i / 98, dw = i - mi * 98;
    const int oy = ty * 4 + (mi >> 2), ox = tx_ * 4 + (mi & 3);
    if (oy < A.H && ox < A.W) out32[(size_t)(oy * A.W + ox) * 98 + dw] = stage[mi * SS + dw];
  }
}

}  // namespace

extern "C" int gs_altcorr_pyramid(const void* pyr0, const void* pyr1, const void* pyr2, const void* pyr3,
                                  const float* coords, const int64_t* ii, const int64_t* jj, void* out, int e, int h, int w,
                                  int c, int radius, gs_stream_t stream) {
  GS_REQUIRE(pyr0 && pyr1 && pyr2 && pyr3 && coords && ii && jj && out, "altcorr_pyramid: null pointer");
  GS_REQUIRE(e >= 0 && h >= 8 && w >= 8, "altcorr_pyramid: bad shape (four levels need maps of at least 8 x 8)");
  GS_REQUIRE(radius == 3, "altcorr_pyramid: only radius 3 (the reference's value) is built");
  if (c != AC) {
    gs_set_error("altcorr_pyramid: C=%d not built (128: the reference's feature width)", c);
    return GS_ERR_UNSUPPORTED;
  }
  if (e == 0) return GS_OK;
  AltPyrArgs A;
  A.pyr[0] = (const _Float16*)pyr0; A.pyr[1] = (const _Float16*)pyr1;
  A.pyr[2] = (const _Float16*)pyr2; A.pyr[3] = (const _Float16*)pyr3;
  A.coords = coords; A.ii = ii; A.jj = jj; A.out = (_Float16*)out;
  A.E = e; A.H = h; A.W = w;
  const int tiles_x = gs_cdiv(w, 4), tiles = tiles_x * gs_cdiv(h, 4);
  const int bpe = gs_cdiv(tiles, 4);
  const long blocks = 8L * gs_cdiv(e, 8) * bpe;
  GS_REQUIRE(blocks <= 0x7fffffffL, "altcorr_pyramid: %ld workgroups exceed the grid limit", blocks);
  GS_TIMING_PRE();
  altcorr_pyramid_kernel<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(A, tiles_x, tiles, bpe);
  GS_CHECK_LAUNCH("altcorr_pyramid");
  return GS_OK;
}
