// 3x3 / pad 1 / stride 1 convolution, NHWC fp16 -> NHWC fp16 (fp32 accumulation), as an implicit GEMM on MFMA:
// the five large convolutions of the update operator (GRU z|r 320->256 and q 320->128, the merged heads 128->384,
// corr_encoder[2] 128->128, the hoisted context gates 128->384; reference src/modules/gru.py:10-12,
// src/droid_net.py:76,83-92,40) -- 1.27 TFLOP of an update's 1.6.
//
// Measured on MI355X (75 edges, 60x80 maps): 885 / 896 / 917 / 859 TFLOP/s on the four layers against MIOpen's 795 / 706 /
// 792 / 658; the host mirror uses it wherever the 16x16 tiles fit the map (go_slam_amd/droid_net.py, CONV3X3_IMPL).
// ROUND-1 KERNELS, kept as the A/B reference of the production kernel (conv3x3_pp.hip, which is 1.2-1.3x faster:
// tools/conv3x3_bench.py, profiles/r02_conv3x3_bench.json): conv3x3_kernel<KC, LP>, conv3x3_stacked_kernel<KC, TW, LP>;
// LP = lane-permuted (bank-conflict-free) fragments, with them optionally the XCD-aware block order
// (GOSLAM_CONV3X3_LANEPERM / _XCD; both hardware-verified in round 2, worth 1-4 %).
//
// Organisation:
//  * a workgroup (4 waves) owns a 16x16-pixel tile x 128 output channels; K = 9 taps x C runs in chunks of 32
//    input channels;
//  * per chunk the 18x18-pixel input patch (20.7 KB) is staged in LDS ONCE and serves all 9 taps -- a generic
//    implicit GEMM re-gathers it per tap; the tap is just an offset into the patch;
//  * weights are pre-packed per (128-channel block, chunk, tap) into the exact LDS image (8 KB), double-buffered in
//    LDS with the next tap's image in flight in registers while the current tap is on the matrix cores;
//  * weights are the MFMA A operand (rows = output channels), pixels the B operand: a wave's 64-channel x 128-pixel
//    register tile needs 2 + 4 LDS fragment reads per 8 MFMAs (24 B/clk/wave), and the accumulator layout gives each
//    lane 4 consecutive channels of one pixel -- packed 8-byte LDS writes in the epilogue;
//  * patch and weight planes are laid out [8-channel group][pixel | channel][8]: a weight-fragment read is 32
//    consecutive 16-byte vectors per half-wave (conflict-free); a pixel-fragment read is two 16-pixel rows 18 slots
//    apart, which ds_read_b128's service groups turn into a 2-way conflict -- removed by the LP instantiations;
//  * the epilogue goes through a wave-private LDS tile so that global stores are 128 B per pixel (8 lanes x 16 B).
// LDS: 37 KB (KC = 32) / 74 KB (KC = 64) per workgroup; 2 workgroups per CU at 210-234 VGPRs, no spills.
#include "common.h"
#include "conv3x3_common.h"
#include <stdlib.h>

namespace {

constexpr int TH = 16, TW = 16;            // output tile (pixels)
constexpr int PH = TH + 2, PW = TW + 2;    // input patch
constexpr int NP = PH * PW;                // 324 patch pixels
constexpr int BN = 128;                    // output channels per workgroup
constexpr int TS = 72;                     // epilogue tile row stride in halves (144 B, staggers the banks)

// KC = input channels per chunk (32: 37 KB of LDS, a barrier every 16 MFMAs per wave; 64: 74 KB, every 32 MFMAs)
template <int KC, bool LP>
__global__ __launch_bounds__(256, 2) void conv3x3_kernel(const _Float16* __restrict__ x, int xs, int C,
                                                         const half8* __restrict__ wpack, _Float16* __restrict__ y,
                                                         int ys, int H, int W, int tiles_x, int tiles_y, int NB,
                                                         int xcd) {
  constexpr int KG = KC / 8;                             // 8-channel groups per chunk
  constexpr int WTAP = KG * BN;                          // 16-byte vectors of one tap's weight image
  constexpr int WPT = WTAP / 256;                        // ... per thread
  constexpr int NPP = LP ? NP + 1 : NP;                  // plane stride; +1 staggers the 4 planes over the write banks
  extern __shared__ half8 smem[];                        // patch [KG][NPP] | weights [2][KG][BN]
  half8* patch = smem;
  half8* wbuf = smem + KG * NPP;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wm = wv & 1, wn = wv >> 1;                    // pixel half (tile rows 8 wm ..), channel half (64 wn ..)
  const int r = lane & 31, kgl = lane >> 5;
  int t = blockIdx.x, nb_ = blockIdx.y;
  if constexpr (LP) decode_block(blockIdx.x, gridDim.x / NB, NB, xcd, t, nb_);   // LP: 1-D grid of tiles * NB ids
  const int tx0 = (t % tiles_x) * TW;
  t /= tiles_x;
  const int ty0 = (t % tiles_y) * TH;
  const int img = t / tiles_y;
  const int nb = nb_;
  const int nchunk = C / KC;
  const _Float16* ximg = x + (size_t)img * H * W * xs;
  const half8* wsrc = wpack + (size_t)nb * nchunk * 9 * WTAP;

  // patch offsets of this lane's pixel in the 4 pixel fragments (32 pixels = 2 tile rows each)
  int pb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if constexpr (LP) {
      int ty, tx;
      tile_pixel<TW, LP>(wm, i, r, ty, tx);
      pb[i] = ty * PW + tx;
    } else {
      pb[i] = (wm * 8 + 2 * i + (r >> 4)) * PW + (r & 15);
    }
  }

  float16v acc[2][4];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][i][e] = 0.0f;

  const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int ck = 0; ck < nchunk; ++ck) {
    // ---- stage the input patch of this channel chunk (4 lanes read the 64 contiguous bytes of a pixel)
    for (int it = tid; it < KG * NP; it += 256) {
      const int kg = it & (KG - 1), p = it / KG;
      const int py = p / PW, px = p - py * PW;
      const int gy = ty0 + py - 1, gx = tx0 + px - 1;
      half8 v = zero8;
      if (gy >= 0 && gy < H && gx >= 0 && gx < W)
        v = *reinterpret_cast<const half8*>(ximg + ((size_t)gy * W + gx) * xs + ck * KC + kg * 8);
      patch[kg * NPP + p] = v;
    }
    // ---- and the first tap's weights
    const half8* wck = wsrc + (size_t)ck * 9 * WTAP;
#pragma unroll
    for (int q = 0; q < WPT; ++q) wbuf[tid + 256 * q] = wck[tid + 256 * q];
    __syncthreads();
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      half8 nx[WPT];
      if (tap < 8) {                                      // next tap's image: global -> registers, in flight
#pragma unroll
        for (int q = 0; q < WPT; ++q) nx[q] = wck[(tap + 1) * WTAP + tid + 256 * q];
      }
      __builtin_amdgcn_sched_barrier(0);                  // keep the prefetch at the top of the tap (the scheduler
                                                          // otherwise sinks it to its use and exposes the latency)
      const half8* wb = wbuf + (tap & 1) * WTAP;
      const int toff = (tap / 3) * PW + (tap % 3);
#pragma unroll
      for (int grp = 0; grp < KC / 32; ++grp) {           // 32 channels at a time: 12 fragment reads, then 16 MFMAs
        half8 a[2][2], b[2][4];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const int kg = 4 * grp + 2 * s + kgl;
          a[s][0] = wb[kg * BN + wn * 64 + r];
          a[s][1] = wb[kg * BN + wn * 64 + 32 + r];
#pragma unroll
          for (int i = 0; i < 4; ++i) b[s][i] = patch[kg * NPP + pb[i] + toff];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            acc[0][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s][0], b[s][i], acc[0][i], 0, 0, 0);
            acc[1][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s][1], b[s][i], acc[1][i], 0, 0, 0);
          }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (tap < 8) {
        half8* wnext = wbuf + ((tap + 1) & 1) * WTAP;     // last read during tap - 1: every wave is past that barrier
#pragma unroll
        for (int q = 0; q < WPT; ++q) wnext[tid + 256 * q] = nx[q];
      }
      __syncthreads();                                    // tap 8: also frees the patch and wbuf[0] for the next chunk
    }
  }

  // ---- epilogue: [32 pixels][64 channels] at a time through a wave-private LDS tile (aliases the patch; all waves
  // are past the last barrier, and a wave only touches its own 4.6 KB)
  _Float16* tile = reinterpret_cast<_Float16*>(smem) + wv * 32 * TS;   // 18 KB in all: inside the patch for any KC
  _Float16* yimg = y + (size_t)img * H * W * ys + nb * BN + wn * 64;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {                       // C layout: row (channel) = 8 g + 4 (lane >> 5) + e, col = pixel
        half4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (_Float16)acc[j][i][4 * g + e];
        *reinterpret_cast<half4*>(tile + r * TS + j * 32 + 8 * g + 4 * kgl) = o;
      }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int pxr = it * 8 + (lane >> 3), piece = lane & 7;
      int gy, gx;
      if constexpr (LP) {
        int ty, tx;
        tile_pixel<TW, LP>(wm, i, pxr, ty, tx);
        gy = ty0 + ty;
        gx = tx0 + tx;
      } else {
        gy = ty0 + wm * 8 + 2 * i + (pxr >> 4);
        gx = tx0 + (pxr & 15);
      }
      if (gy < H && gx < W) {
        const half8 v = *reinterpret_cast<const half8*>(tile + pxr * TS + piece * 8);
        *reinterpret_cast<half8*>(yimg + ((size_t)gy * W + gx) * ys + piece * 8) = v;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Row-stacked variant: the n images are treated as ONE image of n*H rows (they are contiguous in NHWC memory), so
// 256-pixel tiles of TH = 256 / TW rows run across image boundaries and only the very last tile of the batch is
// partial in y -- no tile padding for H = 60, 40 or 30.  What a tile must not do is let a pixel of image k see rows
// of image k +- 1 through the vertical taps: the B fragments of tap row 0 are zeroed for pixels with y == 0, those of
// tap row 2 for pixels with y == H - 1 (a per-lane select, 4 dwords per fragment).  TW in {8, 16, 32} picks the tile
// width that divides the map width best (40 -> 8, 80 -> 16).
template <int KC, int TW, bool LP>
__global__ __launch_bounds__(256, 2) void conv3x3_stacked_kernel(const _Float16* __restrict__ x, int xs, int C,
                                                                 const half8* __restrict__ wpack,
                                                                 _Float16* __restrict__ y, int ys, int H, int W,
                                                                 int rows, int tiles_x, int NB, int xcd) {
  constexpr int TH_ = 256 / TW, PW_ = TW + 2, NP_ = (TH_ + 2) * PW_;
  constexpr int NPP = LP ? NP_ + 1 : NP_;
  constexpr int KG = KC / 8, WTAP = KG * BN, WPT = WTAP / 256;
  extern __shared__ half8 smem[];                        // patch [KG][NPP] | weights [2][KG][BN]
  half8* patch = smem;
  half8* wbuf = smem + KG * NPP;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wm = wv & 1, wn = wv >> 1;
  const int r = lane & 31, kgl = lane >> 5;
  int tx0, g0, nb;                                        // g0: first stacked row (img * H + y) of the tile
  if constexpr (LP) {
    int tix;
    decode_block(blockIdx.x, gridDim.x / NB, NB, xcd, tix, nb);
    tx0 = (tix % tiles_x) * TW;
    g0 = (tix / tiles_x) * TH_;
  } else {
    tx0 = (blockIdx.x % tiles_x) * TW;
    g0 = (blockIdx.x / tiles_x) * TH_;
    nb = blockIdx.y;
  }
  const int nchunk = C / KC;
  const half8* wsrc = wpack + (size_t)nb * nchunk * 9 * WTAP;

  int pb[4];
  bool top[4], bot[4];                                    // this lane's pixel is on the first / last row of its image
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int ty, tx;
    if constexpr (LP) {
      tile_pixel<TW, LP>(wm, i, r, ty, tx);
    } else {
      const int m = wm * 128 + i * 32 + r;
      ty = m / TW;
      tx = m % TW;
    }
    pb[i] = ty * PW_ + tx;
    const int yy = (g0 + ty) % H;
    top[i] = yy == 0;
    bot[i] = yy == H - 1;
  }

  float16v acc[2][4];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][i][e] = 0.0f;

  const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int ck = 0; ck < nchunk; ++ck) {
    for (int it = tid; it < KG * NP_; it += 256) {
      const int kg = it & (KG - 1), p = it / KG;
      const int pr = p / PW_, pc = p - pr * PW_;
      const int gv = g0 + pr - 1, gx = tx0 + pc - 1;      // stacked row: rows of neighbouring images are loaded as
      half8 v = zero8;                                    // they are and masked per pixel below
      if (gv >= 0 && gv < rows && gx >= 0 && gx < W)
        v = *reinterpret_cast<const half8*>(x + ((size_t)gv * W + gx) * xs + ck * KC + kg * 8);
      patch[kg * NPP + p] = v;
    }
    const half8* wck = wsrc + (size_t)ck * 9 * WTAP;
#pragma unroll
    for (int q = 0; q < WPT; ++q) wbuf[tid + 256 * q] = wck[tid + 256 * q];
    __syncthreads();
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      half8 nx[WPT];
      if (tap < 8) {
#pragma unroll
        for (int q = 0; q < WPT; ++q) nx[q] = wck[(tap + 1) * WTAP + tid + 256 * q];
      }
      __builtin_amdgcn_sched_barrier(0);
      const half8* wb = wbuf + (tap & 1) * WTAP;
      const int dy = tap / 3;
      const int toff = dy * PW_ + (tap % 3);
#pragma unroll
      for (int grp = 0; grp < KC / 32; ++grp) {
        half8 a[2][2], b[2][4];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const int kg = 4 * grp + 2 * s + kgl;
          a[s][0] = wb[kg * BN + wn * 64 + r];
          a[s][1] = wb[kg * BN + wn * 64 + 32 + r];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            half8 v = patch[kg * NPP + pb[i] + toff];
            if (dy == 0 && top[i]) v = zero8;             // the row above belongs to the previous image
            if (dy == 2 && bot[i]) v = zero8;             // the row below belongs to the next image
            b[s][i] = v;
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            acc[0][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s][0], b[s][i], acc[0][i], 0, 0, 0);
            acc[1][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s][1], b[s][i], acc[1][i], 0, 0, 0);
          }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (tap < 8) {
        half8* wnext = wbuf + ((tap + 1) & 1) * WTAP;
#pragma unroll
        for (int q = 0; q < WPT; ++q) wnext[tid + 256 * q] = nx[q];
      }
      __syncthreads();
    }
  }

  _Float16* tile = reinterpret_cast<_Float16*>(smem) + wv * 32 * TS;
  _Float16* yb = y + nb * BN + wn * 64;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        half4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (_Float16)acc[j][i][4 * g + e];
        *reinterpret_cast<half4*>(tile + r * TS + j * 32 + 8 * g + 4 * kgl) = o;
      }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int pxr = it * 8 + (lane >> 3), piece = lane & 7;
      int gv, gx;
      if constexpr (LP) {
        int ty, tx;
        tile_pixel<TW, LP>(wm, i, pxr, ty, tx);
        gv = g0 + ty;
        gx = tx0 + tx;
      } else {
        const int m = wm * 128 + i * 32 + pxr;
        gv = g0 + m / TW;
        gx = tx0 + m % TW;
      }
      if (gv < rows && gx < W) {
        const half8 v = *reinterpret_cast<const half8*>(tile + pxr * TS + piece * 8);
        *reinterpret_cast<half8*>(yb + ((size_t)gv * W + gx) * ys + piece * 8) = v;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// Opt-in tuning switches, read once.  The lane permutation (bank-conflict-free B reads; see frag_lane) and the
// XCD-aware workgroup order (see decode_block; only with the lane permutation) are parity-checked by emulation but had
// not run on hardware at the end of round 1, so both are OFF unless GOSLAM_CONV3X3_LANEPERM=1 / GOSLAM_CONV3X3_XCD=1.
static bool env_flag(const char* name) {
  const char* e = getenv(name);
  return e && e[0] == '1';
}
static bool lane_perm_enabled() {
  static const bool on = env_flag("GOSLAM_CONV3X3_LANEPERM");
  return on;
}
static int xcd_remap_enabled() {
  static const int on = env_flag("GOSLAM_CONV3X3_XCD") ? 1 : 0;
  return on;
}

template <int KC, int TW, bool LP>
int launch3x3s(const void* x, int x_stride, int c_in, const void* wpack, void* y, int y_stride, int n_out, int n, int h,
               int w, hipStream_t st) {
  constexpr int NP_ = (256 / TW + 2) * (TW + 2) + (LP ? 1 : 0);
  constexpr size_t lds = (size_t)((KC / 8) * NP_ + 2 * (KC / 8) * BN) * sizeof(half8);
  static GsLdsLimit limit;
  if (int rc = limit.raise((const void*)conv3x3_stacked_kernel<KC, TW, LP>, lds, "conv3x3_stacked")) return rc;
  const long long rows = (long long)n * h;
  GS_REQUIRE(rows * w < (1ll << 31), "conv3x3_stacked: too many pixels");
  const int tiles_x = gs_cdiv(w, TW), tiles_y = gs_cdiv((int)rows, 256 / TW);
  const int NB = n_out / BN;
  const dim3 grid = LP ? dim3((unsigned)(tiles_x * tiles_y * NB)) : dim3((unsigned)(tiles_x * tiles_y), NB);
  conv3x3_stacked_kernel<KC, TW, LP><<<grid, 256, lds, st>>>((const _Float16*)x, x_stride, c_in, (const half8*)wpack,
                                                             (_Float16*)y, y_stride, h, w, (int)rows, tiles_x, NB,
                                                             xcd_remap_enabled());
  GS_CHECK_LAUNCH("conv3x3_stacked");
  return GS_OK;
}

template <int KC, bool LP>
int launch3x3(const void* x, int x_stride, int c_in, const void* wpack, void* y, int y_stride, int n_out, int n, int h,
              int w, hipStream_t st) {
  constexpr size_t lds = (size_t)((KC / 8) * (NP + (LP ? 1 : 0)) + 2 * (KC / 8) * BN) * sizeof(half8);
  static GsLdsLimit limit;
  if (int rc = limit.raise((const void*)conv3x3_kernel<KC, LP>, lds, "conv3x3")) return rc;
  const int tiles_x = gs_cdiv(w, TW), tiles_y = gs_cdiv(h, TH);
  const long long blocks = (long long)n * tiles_x * tiles_y;
  GS_REQUIRE(blocks < (1ll << 31), "conv3x3: too many tiles");
  const int NB = n_out / BN;
  GS_REQUIRE(blocks * NB < (1ll << 31), "conv3x3: too many workgroups");
  const dim3 grid = LP ? dim3((unsigned)(blocks * NB)) : dim3((unsigned)blocks, NB);
  conv3x3_kernel<KC, LP><<<grid, 256, lds, st>>>((const _Float16*)x, x_stride, c_in, (const half8*)wpack,
                                                      (_Float16*)y, y_stride, h, w, tiles_x, tiles_y, NB,
                                                      xcd_remap_enabled());
  GS_CHECK_LAUNCH("conv3x3");
  return GS_OK;
}

}  // namespace

extern "C" size_t gs_conv3x3_wpack_elems(int c_in, int n_out) { return (size_t)9 * c_in * n_out; }

extern "C" int gs_conv3x3(const void* x, int x_stride, int c_in, const void* wpack, int kc, void* y, int y_stride,
                          int n_out, int n, int h, int w, gs_stream_t stream) {
  GS_REQUIRE(x && wpack && y, "conv3x3: null pointer");
  GS_REQUIRE(kc == 32 || kc == 64, "conv3x3: kc (channels per chunk of the packed weights) must be 32 or 64");
  GS_REQUIRE(c_in > 0 && c_in % kc == 0, "conv3x3: c_in must be a multiple of kc = %d", kc);
  GS_REQUIRE(n_out > 0 && n_out % BN == 0, "conv3x3: n_out must be a multiple of %d", BN);
  GS_REQUIRE(x_stride >= c_in && x_stride % 8 == 0, "conv3x3: x_stride must be >= c_in and a multiple of 8");
  GS_REQUIRE(y_stride >= n_out && y_stride % 8 == 0, "conv3x3: y_stride must be >= n_out and a multiple of 8");
  GS_REQUIRE(n >= 0 && h > 0 && w > 0, "conv3x3: bad shape");
  if (n == 0) return GS_OK;
  hipStream_t st = (hipStream_t)stream;
  if (lane_perm_enabled()) {
    if (kc == 32) return launch3x3<32, true>(x, x_stride, c_in, wpack, y, y_stride, n_out, n, h, w, st);
    return launch3x3<64, true>(x, x_stride, c_in, wpack, y, y_stride, n_out, n, h, w, st);
  }
  if (kc == 32) return launch3x3<32, false>(x, x_stride, c_in, wpack, y, y_stride, n_out, n, h, w, st);
  return launch3x3<64, false>(x, x_stride, c_in, wpack, y, y_stride, n_out, n, h, w, st);
}

extern "C" int gs_conv3x3_stacked(const void* x, int x_stride, int c_in, const void* wpack, int kc, int tw, void* y,
                                  int y_stride, int n_out, int n, int h, int w, gs_stream_t stream) {
  GS_REQUIRE(x && wpack && y, "conv3x3_stacked: null pointer");
  GS_REQUIRE(kc == 32 || kc == 64, "conv3x3_stacked: kc must be 32 or 64");
  GS_REQUIRE(tw == 8 || tw == 16 || tw == 32, "conv3x3_stacked: tile width must be 8, 16 or 32");
  GS_REQUIRE(c_in > 0 && c_in % kc == 0, "conv3x3_stacked: c_in must be a multiple of kc = %d", kc);
  GS_REQUIRE(n_out > 0 && n_out % BN == 0, "conv3x3_stacked: n_out must be a multiple of %d", BN);
  GS_REQUIRE(x_stride >= c_in && x_stride % 8 == 0, "conv3x3_stacked: x_stride must be >= c_in and a multiple of 8");
  GS_REQUIRE(y_stride >= n_out && y_stride % 8 == 0, "conv3x3_stacked: y_stride must be >= n_out and a multiple of 8");
  GS_REQUIRE(n >= 0 && h > 0 && w > 0, "conv3x3_stacked: bad shape");
  if (n == 0) return GS_OK;
  hipStream_t st = (hipStream_t)stream;
#define GS_S(KC_, TW_, LP_) return launch3x3s<KC_, TW_, LP_>(x, x_stride, c_in, wpack, y, y_stride, n_out, n, h, w, st)
  if (lane_perm_enabled()) {
    if (kc == 32) {
      if (tw == 8) GS_S(32, 8, true);
      if (tw == 16) GS_S(32, 16, true);
      GS_S(32, 32, false);                                // 1 x 32 fragments are conflict-free as they are
    }
    if (tw == 8) GS_S(64, 8, true);
    if (tw == 16) GS_S(64, 16, true);
    GS_S(64, 32, false);
  }
  if (kc == 32) {
    if (tw == 8) GS_S(32, 8, false);
    if (tw == 16) GS_S(32, 16, false);
    GS_S(32, 32, false);
  }
  if (tw == 8) GS_S(64, 8, false);
  if (tw == 16) GS_S(64, 16, false);
  GS_S(64, 32, false);
#undef GS_S
}
