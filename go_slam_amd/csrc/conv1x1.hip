// 1x1 convolutions of the update operator as a memory-bound MFMA GEMM with a fused epilogue:
// corr_encoder[0] (196 -> 128, ReLU; reference src/droid_net.py:75) and GraphAgg's upmask head
// (128 -> 576, bias only; src/droid_net.py:45).
//
// y[p, :] = act(W x[p, :] + b) for NHWC fp16 pixels p.  Arithmetic intensity is ~100 FLOP/B, far
// below the MFMA/HBM balance point, so the kernel is organised around the byte stream: every pixel
// row is read once (K halves), every output row written once (N halves), bias + activation happen in
// the accumulator registers (MIOpen's CK kernel + a separate bias/ReLU pass moved the output twice
// more).  W^T is the MFMA A operand: pre-packed per (32-channel block, 16-wide k-step) into 1 KB
// fragments, copied to LDS once per workgroup (53 KB for 196->128, 147 KB for 128->576) and read back
// conflict-free; pixels are the B operand.  A fragment lane wants 16 bytes of ITS pixel row, so loading
// fragments straight from memory makes every load instruction touch 32 rows (32 cache lines per KB:
// the texture path, not HBM, was the bound).  When the rows are dense (x_stride == k_in) a wave's 32
// pixels are ONE contiguous chunk: it is fetched with fully coalesced 16-byte loads, parked in a
// wave-private LDS stage (rows padded to a stride of 4 banks mod 64: conflict-free ds_read_b128) and
// the fragments are read from there; the stage doubles as the epilogue's transpose tile.  Strided
// inputs keep the direct loads.  Workgroups (8 waves) are persistent and stride over 32-pixel blocks.
#include "common.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float float16v __attribute__((ext_vector_type(16)));
struct __attribute__((packed, aligned(8))) half8_a8 { half8 v; };   // pixel rows of 196 halves are 8-B aligned

// 16-byte piece `c` of a dense chunk of `avail` halves; the last piece of a chunk whose length is 4 mod 8 is 8 bytes
__device__ __forceinline__ half8 ld_piece(const _Float16* base, int c, size_t avail) {
  if ((size_t)c * 8 + 8 <= avail) return *reinterpret_cast<const half8*>(base + (size_t)c * 8);
  const half4 t = *reinterpret_cast<const half4*>(base + (size_t)c * 8);
  half8 v = {t[0], t[1], t[2], t[3], 0, 0, 0, 0};
  return v;
}

constexpr int C1_WAVES = 8;
constexpr int C1_THREADS = 64 * C1_WAVES;

// halves per row of the wave stage: an ODD number of 16-byte pieces >= the row, so that rows r, r + 1, ... r + 15 start
// in 16 different 16-byte slots modulo 256 B (conflict-free ds_read_b128 service groups)
__host__ __device__ inline int c1_stage_stride(int K) {
  const int pieces = (K + 7) / 8;
  return 8 * (pieces | 1);
}

template <int KS>   // k-steps of 16 input channels
__global__ __launch_bounds__(C1_THREADS) void conv1x1_kernel(const _Float16* __restrict__ x, int ldx, int K,
                                                      const _Float16* __restrict__ wpack,
                                                      const float* __restrict__ bias, int act,
                                                      _Float16* __restrict__ y, int ldy, int NBT, int NBS,
                                                      size_t rows, int staged) {
  // blockIdx.y picks a slice of NBS 32-channel blocks: wide outputs (576) are split so that a slice's
  // weights (<= 52 KB) leave room for 3 workgroups per CU; the pixel rows are re-read per slice from L2
  extern __shared__ _Float16 wl[];            // [NB][KS][64][8]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int nbase = blockIdx.y * NBS;
  const int NB = min(NBS, NBT - nbase);
  {
    const int nfrag16 = NB * KS * 64;         // 16-byte pieces
    const half8* src = reinterpret_cast<const half8*>(wpack) + (size_t)nbase * KS * 64;
    for (int i = tid; i < nfrag16; i += C1_THREADS) reinterpret_cast<half8*>(wl)[i] = src[i];
  }
  __syncthreads();
  constexpr int TS = 72;                      // tile row stride in halves (144 B: 16-B aligned, staggered banks)
  const int SS = c1_stage_stride(K);          // stage row stride in halves
  const int wave_lds = staged ? (32 * SS > 32 * TS ? 32 * SS : 32 * TS) : 32 * TS;
  _Float16* tile = wl + (size_t)NBS * KS * 512 + (size_t)wv * wave_lds;     // stage and transpose tile share it
  // this slice's bias in LDS (per-lane global loads of it inside the block loop cost a memory round trip per use)
  float* sbias = reinterpret_cast<float*>(wl + (size_t)NBS * KS * 512 + (size_t)C1_WAVES * wave_lds);
  for (int i = tid; i < NB * 32; i += C1_THREADS) sbias[i] = bias ? bias[nbase * 32 + i] : 0.0f;
  __syncthreads();
  const int kh = 8 * (lane >> 5);
  const size_t nblk = (rows + 31) / 32;
  const int U = K >> 2;                       // 8-byte units per row
  const float invU = 1.0f / (float)U;
  // the wave's 32 rows = one contiguous chunk of 32 * K halves (fewer at the tail): 16-byte pieces, coalesced.
  // (Requesting the NEXT block's pieces before this block's MFMAs -- 52 more VGPRs -- measured no faster.)
  constexpr int NLOAD = (KS * 16 * 32 / 8 + 63) / 64;                     // >= ceil(32 K / 8 / 64)
  half8 pc[NLOAD];
  auto issue = [&](size_t blk) {
    const size_t first = blk * 32 * (size_t)K;                            // in halves; 16-byte aligned (32 K * 2 B)
    const size_t avail = (rows - blk * 32 < 32 ? rows - blk * 32 : 32) * (size_t)K;
    const int npiece = (int)((avail + 7) >> 3);
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) {
      const int c = i * 64 + lane;
      pc[i] = ld_piece(x + first, c < npiece ? c : npiece - 1, avail);
    }
  };
  const size_t blk_first = (size_t)blockIdx.x * C1_WAVES + wv, blk_step = (size_t)gridDim.x * C1_WAVES;
  for (size_t blk = blk_first; blk < nblk; blk += blk_step) {
    const size_t px = blk * 32 + (lane & 31);
    const bool valid = px < rows;
    half8 b[KS];
    if (staged) {
      issue(blk);
      const size_t avail = (rows - blk * 32 < 32 ? rows - blk * 32 : 32) * (size_t)K;
      const int npiece = (int)((avail + 7) >> 3);
#pragma unroll
      for (int i = 0; i < NLOAD; ++i) {
        const int c = i * 64 + lane;
        if (c < npiece) {
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {                                // two 8-byte units; K % 4 == 0: none straddles a row
            const int u = 2 * c + hh;
            const int r = (int)(((float)u + 0.5f) * invU);
            const int cu = u - r * U;
            if (r < 32) {
              half4 t4;
              t4[0] = pc[i][4 * hh]; t4[1] = pc[i][4 * hh + 1]; t4[2] = pc[i][4 * hh + 2]; t4[3] = pc[i][4 * hh + 3];
              *reinterpret_cast<half4*>(tile + r * SS + 4 * cu) = t4;
            }
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const _Float16* sr = tile + (lane & 31) * SS + kh;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int k0 = 16 * ks + kh;
        half8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (valid) {
          if (k0 + 8 <= K) v = *reinterpret_cast<const half8*>(sr + 16 * ks);
          else if (k0 + 4 <= K) {
            const half4 t = *reinterpret_cast<const half4*>(sr + 16 * ks);
            v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
          }
        }
        b[ks] = v;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();                                    // fragments are in registers: the tile is free
    } else {
      const _Float16* xr = x + (valid ? px : 0) * (size_t)ldx + kh;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int k0 = 16 * ks + kh;
        half8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (valid) {
          if (k0 + 8 <= K) v = reinterpret_cast<const half8_a8*>(xr + 16 * ks)->v;
          else if (k0 + 4 <= K) {               // K % 8 == 4 (196): the last 4 channels
            const half4 t = *reinterpret_cast<const half4*>(xr + 16 * ks);
            v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
          }
        }
        b[ks] = v;
      }
    }
    for (int nb0 = 0; nb0 < NB; nb0 += 4) {
      float16v c[4];
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 16; ++e) c[q][e] = 0.0f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (nb0 + q < NB) {
            const half8 a = *reinterpret_cast<const half8*>(wl + ((size_t)((nb0 + q) * KS + ks) * 64 + lane) * 8);
            c[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b[ks], c[q], 0, 0, 0);
          }
        }
      }
      // epilogue: bias + activation in registers, then 64 channels at a time through a wave-private
      // LDS tile so that the global stores are 128 B per pixel row (8 lanes x 16 B), not 8-byte
      // fragments of 32 different rows per instruction (which ran the store path at ~1.2 TB/s)
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        const int nbp = nb0 + 2 * pr;
        if (nbp >= NB) break;
        const int nvalid = min(2, NB - nbp);          // 32-channel blocks in this pair
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
          const int q = 2 * pr + qq;
          if (qq >= nvalid) continue;
#pragma unroll
          for (int g = 0; g < 4; ++g) {               // 4 consecutive channels per register group
            const int chl = qq * 32 + 8 * g + 4 * (lane >> 5);
            half4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float f = c[q][4 * g + e] + sbias[nbp * 32 + chl + e];
              if (act == 1) f = fmaxf(f, 0.0f);
              o[e] = (_Float16)f;
            }
            *reinterpret_cast<half4*>(tile + (lane & 31) * TS + chl) = o;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int pxr = it * 8 + (lane >> 3), piece = lane & 7;
          const size_t gp = blk * 32 + pxr;
          if (gp < rows && piece * 8 < nvalid * 32) {
            const half8 v = *reinterpret_cast<const half8*>(tile + pxr * TS + piece * 8);
            *reinterpret_cast<half8*>(y + gp * (size_t)ldy + (nbase + nbp) * 32 + piece * 8) = v;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
}

template <int KS>
int launch1x1(const void* x, int ldx, int K, const void* wpack, const float* bias, int act, void* y, int ldy, int N,
              size_t rows, hipStream_t st) {
  const int NBT = N / 32;
  int NBS = NBT;
  while ((size_t)NBS * KS * 1024 > 52 * 1024 && NBS > 1) NBS = (NBS + 1) / 2;
  NBS = (NBS + 3) / 4 * 4 > NBT ? NBT : (NBS + 3) / 4 * 4;       // whole groups of 4 blocks (one accumulator pass)
  if ((size_t)NBS * KS * 1024 > 60 * 1024) NBS = NBS > 4 ? NBS - 4 : NBS;
  const int nslice = (NBT + NBS - 1) / NBS;
  // dense, 16-byte aligned rows: coalesced loads through the wave stages (see the kernel)
  int staged = (ldx == K && ((size_t)x & 15) == 0) ? 1 : 0;
  size_t wave_lds = 32 * 72;
  if (staged && (size_t)32 * c1_stage_stride(K) > wave_lds) wave_lds = (size_t)32 * c1_stage_stride(K);
  size_t lds = (size_t)NBS * KS * 1024 + C1_WAVES * wave_lds * sizeof(_Float16) + (size_t)NBS * 32 * sizeof(float);
  if (lds > 160 * 1024) {                       // cannot happen for the update operator's layers; keep the direct loads
    staged = 0;
    lds = (size_t)NBS * KS * 1024 + C1_WAVES * 32 * 72 * sizeof(_Float16) + (size_t)NBS * 32 * sizeof(float);
  }
  GS_REQUIRE(lds <= 160 * 1024, "conv1x1: %d x %d weights need %zu bytes of LDS", N, K, lds);
  static GsLdsLimit limit;
  if (int rc = limit.raise((const void*)conv1x1_kernel<KS>, 160 * 1024, "conv1x1")) return rc;
  const size_t nblk = (rows + 31) / 32;
  size_t grid = (size_t)256 / nslice;           // one 8-wave workgroup per CU and slice
  if (grid < 32) grid = 32;
  if (grid > (nblk + C1_WAVES - 1) / C1_WAVES) grid = (nblk + C1_WAVES - 1) / C1_WAVES;
  conv1x1_kernel<KS><<<dim3((unsigned)grid, nslice), C1_THREADS, lds, st>>>(
      (const _Float16*)x, ldx, K, (const _Float16*)wpack, bias, act, (_Float16*)y, ldy, NBT, NBS, rows, staged);
  GS_CHECK_LAUNCH("conv1x1");
  return GS_OK;
}

}  // namespace

extern "C" int gs_conv1x1(const void* x, int x_stride, int k_in, const void* wpack, const float* bias, int act,
                          void* y, int y_stride, int n_out, long long rows, gs_stream_t stream) {
  GS_REQUIRE(x && wpack && y, "conv1x1: null pointer");
  GS_REQUIRE(k_in > 0 && k_in % 4 == 0 && k_in <= 208, "conv1x1: k_in must be a multiple of 4, <= 208");
  GS_REQUIRE(x_stride >= k_in && x_stride % 4 == 0, "conv1x1: x_stride must be >= k_in and a multiple of 4");
  GS_REQUIRE(n_out > 0 && n_out % 32 == 0, "conv1x1: n_out must be a multiple of 32");
  GS_REQUIRE(y_stride >= n_out && y_stride % 8 == 0, "conv1x1: y_stride must be >= n_out and a multiple of 8");
  GS_REQUIRE(act == 0 || act == 1, "conv1x1: act in {0 none, 1 relu}");
  GS_REQUIRE(rows >= 0, "conv1x1: bad row count");
  if (rows == 0) return GS_OK;
  hipStream_t st = (hipStream_t)stream;
  const int ks = (k_in + 15) / 16;
  if (ks <= 8) return launch1x1<8>(x, x_stride, k_in, wpack, bias, act, y, y_stride, n_out, (size_t)rows, st);
  return launch1x1<13>(x, x_stride, k_in, wpack, bias, act, y, y_stride, n_out, (size_t)rows, st);
}
