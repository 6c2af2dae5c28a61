// 7x7 / pad 3 convolution of a 4-channel NHWC fp16 map to 128 channels with bias (+ ReLU) fused: the first layer of the
// update operator's flow encoder (reference src/droid_net.py:79-83, `flow_encoder[0]` = Conv2d(4, 128, 7, padding=3)
// over the motion features of every edge, every update).  K = 7 x 7 x 4 = 196 is far too short for a tiled implicit
// GEMM to amortise anything (the library kernel this replaces ran 116 us + a 30 us bias/ReLU pass at 75 x 60 x 80);
// the layer is bound by its OUTPUT (256 B per pixel written, 8 B read).  So:
//
//   * K is laid out as 7 kernel rows x (8 taps x 4 channels) = 224 = 14 MFMA k-steps of 16; the 8th tap of a row has zero
//     weights.  One k-group of a fragment = 2 neighbouring pixels x 4 channels = 16 contiguous bytes of the input
//     patch, so the pixel operand is read straight from the NHWC patch in LDS (two ds_read_b64) -- no im2col anywhere;
//   * the WEIGHTS are the MFMA A operand and live in registers for the whole workgroup: a wave owns 64 output channels
//     (2 fragments x 14 k-steps x 4 VGPRs = 112 VGPRs, loaded once from a pre-packed image in lane order);
//   * a workgroup (4 waves = 2 channel halves x 2 pixel-fragment parities) takes a strip of `rt` full-width rows:
//     patch = (rt + 6) x (W + 8) pixels x 8 B; the strip's pixels are consumed as fragments of 32 CONSECUTIVE pixels of
//     the flattened strip (no column padding for any map width; the 32 lanes of a half-wave read 256 contiguous bytes);
//   * accumulators start at the bias; ReLU + fp16 rounding in registers; the 32 px x 64 ch fragment pair goes through a
//     wave-private LDS transpose so that every lane stores 16 B and a pixel's 128 B half-row is written by 8 lanes.
//
// Algorithmic traffic per pixel: 8 B in + 256 B out.  MFMA work: 2 x 224 x 128 flop per pixel (14 % of it on zero taps).
#include "common.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float float16v __attribute__((ext_vector_type(16)));

constexpr int kKSteps = 14;                 // 7 kernel rows x 2 (taps 0-3, taps 4-7)
constexpr int kScratchRow = 72;             // halves per pixel row of the transpose scratch: 64 + 8 (144 B, 16 B-aligned)

struct C7Args {
  const half4* x;            // [n, h, w] pixels of 4 halves
  const half8* wpack;        // [2 channel halves][2 fragments][14 k-steps][64 lanes] x 8 halves
  const float* bias;         // [128]
  _Float16* y; int ys;       // output, `ys` halves per pixel
  int n, h, w, rt, strips;   // strips per image = ceil(h / rt)
  int relu;
};

__global__ __launch_bounds__(256, 2) void conv7x7_c4_kernel(C7Args A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int nh = wv & 1, mp = wv >> 1;
  const int img = blockIdx.x / A.strips, strip = blockIdx.x - img * A.strips;
  const int row0 = strip * A.rt;
  const int rows = min(A.rt, A.h - row0);
  const int W = A.w, Ws = W + 8;
  const int prow = A.rt + 6;

  half4* patch = reinterpret_cast<half4*>(smem);                                  // [prow][Ws]
  float* sbias = reinterpret_cast<float*>(smem + (size_t)prow * Ws * 8);          // [128]
  _Float16* scratch = reinterpret_cast<_Float16*>(sbias + 128) + wv * 32 * kScratchRow;

  // input patch: rows row0 - 3 .. row0 + rt + 2, columns -3 .. W + 4, zero outside the image.  Batches of 8 pixels per
  // thread, branch-free (clamped address + validity bit), all loads of a batch in flight before the first LDS store.
  // vmcnt retires in order, so the FIRST batch is issued before the 28 weight loads and stored while those still fly.
  const half4* ximg = A.x + (size_t)img * A.h * W;
  const half4 zero4 = {0, 0, 0, 0};
  const int ptotal = prow * Ws;
  half4 pv[8];
  unsigned inside = 0u;
  auto patch_issue = [&](int base) {
    inside = 0u;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = base + k * 256 + tid;
      const int pr = i / Ws, pc = i - pr * Ws;
      const int gy = row0 - 3 + pr, gx = pc - 3;
      if (i < ptotal && gy >= 0 && gy < A.h && gx >= 0 && gx < W) inside |= 1u << k;
      const int cy = min(max(gy, 0), A.h - 1), cx = min(max(gx, 0), W - 1);
      pv[k] = ximg[(size_t)cy * W + cx];
    }
  };
  auto patch_store = [&](int base) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = base + k * 256 + tid;
      if (i < ptotal) patch[i] = ((inside >> k) & 1u) ? pv[k] : zero4;
    }
  };
  patch_issue(0);

  // weights of this wave's 64 channels: 28 fragments, resident
  half8 wf[2][kKSteps];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int s = 0; s < kKSteps; ++s) wf[t][s] = A.wpack[((nh * 2 + t) * kKSteps + s) * 64 + lane];

  patch_store(0);
  for (int base = 2048; base < ptotal; base += 2048) {
    patch_issue(base);
    patch_store(base);
  }
  if (tid < 128) sbias[tid] = A.bias[tid];
  __syncthreads();

  const int npix = rows * W;
  const int nfrag = (npix + 31) >> 5;
  const int col = lane & 31, kg = lane >> 5;
  _Float16* yimg = A.y + ((size_t)img * A.h + row0) * W * A.ys + nh * 64;

  for (int f = mp; f < nfrag; f += 2) {
    int p = f * 32 + col;
    if (p >= npix) p = npix - 1;                         // tail lanes recompute the last pixel (not stored)
    const int py = p / W, px = p - py * W;
    const half4* src = patch + py * Ws + px + 2 * kg;

    float16v acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 b = *reinterpret_cast<const float4*>(sbias + nh * 64 + t * 32 + 8 * q + 4 * kg);
        acc[t][4 * q + 0] = b.x; acc[t][4 * q + 1] = b.y; acc[t][4 * q + 2] = b.z; acc[t][4 * q + 3] = b.w;
      }
#pragma unroll
    for (int s = 0; s < kKSteps; ++s) {
      const half4* q = src + (s >> 1) * Ws + 4 * (s & 1);
      const half4 lo = q[0], hi = q[1];
      const half8 pf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[0][s], pf, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[1][s], pf, acc[1], 0, 0, 0);
    }
    // D[channel i][pixel j]: lane holds pixel j = lane % 32, channels 8 q + 4 (lane / 32) + e of each fragment
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        half4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = acc[t][4 * q + e];
          _Float16 hv = (_Float16)v;
          if (A.relu && hv < (_Float16)0) hv = (_Float16)0;
          o[e] = hv;
        }
        *reinterpret_cast<half4*>(scratch + col * kScratchRow + t * 32 + 8 * q + 4 * kg) = o;
      }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int j = it * 8 + (lane >> 3), c8 = lane & 7;
      const half8 v = *reinterpret_cast<const half8*>(scratch + j * kScratchRow + c8 * 8);
      const int pp = f * 32 + j;
      if (pp < npix) *reinterpret_cast<half8*>(yimg + (size_t)pp * A.ys + c8 * 8) = v;
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
}

GsLdsLimit g_c7_lds;

}  // namespace

// x: [n, h, w, 4] fp16; wpack: gs_conv7x7_c4 weight image (include/goslam_hip.h); bias: [128] fp32;
// y: [n, h, w] pixels of `ys` halves, 128 written per pixel.  rt = rows per workgroup (0 = choose).
extern "C" int gs_conv7x7_c4(const void* x, const void* wpack, const float* bias, void* y, int ys, int n, int h, int w,
                             int relu, int rt, gs_stream_t stream) {
  GS_REQUIRE(n >= 0 && h > 0 && w > 0 && w <= 1024, "conv7x7_c4: bad shape (w <= 1024)");
  if (n == 0) return GS_OK;
  GS_REQUIRE(x && wpack && bias && y, "conv7x7_c4: null pointer");
  GS_REQUIRE(ys >= 128 && ys % 8 == 0 && ((size_t)y & 15) == 0 && ((size_t)x & 7) == 0 && ((size_t)wpack & 15) == 0,
             "conv7x7_c4: output rows must be 16-byte aligned, >= 128 halves");
  if (rt <= 0) {
    // rows per strip: one round of workgroups (2 per CU x 256 CUs) covers the launch when it can -- measured best at
    // all three map sizes (75 edges: 60 x 80 -> 10 rows, 85 x 150 -> 15, 48 x 64 -> 8) -- with at least ~256 pixels each
    const int slots = 512;
    const int per_image = n >= slots ? 1 : slots / n;
    rt = (h + per_image - 1) / per_image;
    const int min_rows = (256 + w - 1) / w;
    if (rt < min_rows) rt = min_rows;
    if (rt > h) rt = h;
    while (rt > 1 && (size_t)(rt + 6) * (w + 8) * 8 > 96 * 1024) rt = (rt + 1) / 2;
  }
  if (rt > h) rt = h;
  const size_t lds = (size_t)(rt + 6) * (w + 8) * 8 + 128 * 4 + (size_t)4 * 32 * kScratchRow * 2;
  GS_REQUIRE(lds <= 160 * 1024, "conv7x7_c4: strip does not fit LDS (lower rt)");
  if (int rc = g_c7_lds.raise((const void*)conv7x7_c4_kernel, lds, "conv7x7_c4")) return rc;
  C7Args A;
  A.x = (const half4*)x; A.wpack = (const half8*)wpack; A.bias = bias; A.y = (_Float16*)y; A.ys = ys;
  A.n = n; A.h = h; A.w = w; A.rt = rt; A.strips = (h + rt - 1) / rt; A.relu = relu;
  const long long blocks = (long long)n * A.strips;
  GS_REQUIRE(blocks < (1ll << 31), "conv7x7_c4: too many workgroups");
  conv7x7_c4_kernel<<<(unsigned)blocks, 256, lds, (hipStream_t)stream>>>(A);
  GS_CHECK_LAUNCH("conv7x7_c4");
  return GS_OK;
}
