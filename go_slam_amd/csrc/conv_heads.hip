// 3x3 convolutions from 128 channels to 1 or 2 channels: the update operator's flow-revision and
// confidence heads (reference src/droid_net.py:83-92, delta[2] / weight[2]) and GraphAgg's damping
// head (src/droid_net.py:43, eta[0]).
//
// MIOpen runs these as implicit GEMMs with N = 2 (70 us each at 75x60x80: the MFMA tile is >90 %
// padding) plus separate bias / sigmoid / permute / float passes.  Here the convolution is split by
// linearity into (a) a per-pixel [128] x [128 x 9*O] product -- one MFMA 32x32x16 column block, the 9*O
// tap weights are the A operand and stay in VGPRs -- whose results go to an LDS tile, and (b) a 9-tap
// gather-sum over the tile.  The producer's bias + ReLU is applied to the operand on the fly
// (consumer-side fusion: the 128-channel pre-activation is read once, straight from the merged head
// convolution's output), and the epilogue writes the final fp32 [n,h,w,O] tensor.  HBM-bound: 256 B
// per pixel read (x 1.33 halo re-read at 6-row tiles, served by L2).
#include "common.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));

constexpr int HEAD_ROWS = 6;      // output rows per workgroup

template <int O>
__global__ __launch_bounds__(256) void conv3x3_head_kernel(
    const _Float16* __restrict__ x, int ldx, const float* __restrict__ in_bias, int in_relu,
    const _Float16* __restrict__ wpack, const float* __restrict__ bias, int epilogue, float out_scale,
    float* __restrict__ out, int h, int w) {
  constexpr int NT = 9 * O;             // tap-output columns actually used (<= 32)
  constexpr int S = NT | 1;             // odd LDS row stride: conflict-free column writes
  extern __shared__ float cs[];         // [(HEAD_ROWS + 2) * w][S]
  const int n = blockIdx.y;
  const int r0 = blockIdx.x * HEAD_ROWS;
  const int in_lo = max(r0 - 1, 0), in_hi = min(r0 + HEAD_ROWS, h - 1);
  const int npx = (in_hi - in_lo + 1) * w;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int kh = 8 * (lane >> 5);

  half8 a[8];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) a[ks] = *reinterpret_cast<const half8*>(wpack + ((size_t)ks * 64 + lane) * 8);
  // the producer's bias lives in LDS (two 16-byte reads per k-step), not in 64 registers: the second operand buffer
  // of the prefetch below needs them, and 3 workgroups per CU need <= 168 VGPRs
  __shared__ __attribute__((aligned(16))) float sb[128];
  if (tid < 128) sb[tid] = in_bias ? in_bias[tid] : 0.0f;
  __syncthreads();
  const bool touch = (in_bias != nullptr) || in_relu;

  // One 32-pixel block = 8 fragment loads, MFMAs, 16 LDS writes.  All of a workgroup's blocks are resident at once
  // (750 workgroups on 768 slots at the bench shape), so the kernel's duration IS a wave's chain of blocks: the next
  // block's loads are issued before this block's MFMAs (two register buffers) instead of after its LDS writes.
  auto load_block = [&](int blk, half8 (&v)[8]) {
    const int px = blk * 32 + (lane & 31);
    const _Float16* xr = x + ((size_t)(n * h + in_lo) * w + (px < npx ? px : 0)) * ldx + kh;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) v[ks] = *reinterpret_cast<const half8*>(xr + 16 * ks);
  };
  auto do_block = [&](int blk, half8 (&v)[8]) {
    const int px = blk * 32 + (lane & 31);
    float16v c;
#pragma unroll
    for (int e = 0; e < 16; ++e) c[e] = 0.0f;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      half8 t = v[ks];
      if (touch) {
        typedef float float4v __attribute__((ext_vector_type(4)));
        const float4v b0 = *reinterpret_cast<const float4v*>(sb + ks * 16 + kh);
        const float4v b1 = *reinterpret_cast<const float4v*>(sb + ks * 16 + kh + 4);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float f = (float)t[e] + (e < 4 ? b0[e & 3] : b1[e & 3]);
          if (in_relu) f = fmaxf(f, 0.0f);
          t[e] = (_Float16)f;
        }
      }
      c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks], t, c, 0, 0, 0);
    }
    if (px < npx) {
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        if (row < NT) cs[px * S + row] = c[reg];
      }
    }
  };
  if (wv * 32 < npx) {
    half8 va[8], vb[8];
    int blk = wv;
    load_block(blk, va);
    while (true) {
      const bool more1 = (blk + 4) * 32 < npx;
      if (more1) load_block(blk + 4, vb);
      do_block(blk, va);
      if (!more1) break;
      const bool more2 = (blk + 8) * 32 < npx;
      if (more2) load_block(blk + 8, va);
      do_block(blk + 4, vb);
      if (!more2) break;
      blk += 8;
    }
  }
  __syncthreads();

  const int nrow = min(HEAD_ROWS, h - r0);
  for (int idx = tid; idx < nrow * w; idx += 256) {
    const int r = r0 + idx / w, cc = idx % w;
    float acc[O];
#pragma unroll
    for (int o = 0; o < O; ++o) acc[o] = 0.0f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int rr = r + ky - 1;
      if (rr < 0 || rr >= h) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int c2 = cc + kx - 1;
        if (c2 < 0 || c2 >= w) continue;
        const float* p = cs + ((rr - in_lo) * w + c2) * S + (ky * 3 + kx) * O;
#pragma unroll
        for (int o = 0; o < O; ++o) acc[o] += p[o];
      }
    }
    float* op = out + ((size_t)(n * h + r) * w + cc) * O;
#pragma unroll
    for (int o = 0; o < O; ++o) {
      float v = (float)(_Float16)(acc[o] + bias[o]);                 // the convolution's fp16 output
      if (epilogue == 1) v = (float)(_Float16)gs_sigmoid(v);                          // sigmoid (fp16 op)
      else if (epilogue == 2) v = (v > 20.0f) ? v : log1pf(__expf(v));                // softplus (fp32 op)
      op[o] = v * out_scale;
    }
  }
}

template <int O>
int launch_head(const void* x, int ldx, const float* in_bias, int in_relu, const void* wpack, const float* bias,
                int epilogue, float out_scale, float* out, int n, int h, int w, hipStream_t st) {
  constexpr int S = (9 * O) | 1;
  const size_t lds = (size_t)(HEAD_ROWS + 2) * w * S * sizeof(float);
  GS_REQUIRE(lds <= 160 * 1024 - 512, "conv3x3_head: image width %d needs %zu bytes of LDS", w, lds);   // 512: sb
  static GsLdsLimit limit;
  if (int rc = limit.raise((const void*)conv3x3_head_kernel<O>, lds, "conv3x3_head")) return rc;
  conv3x3_head_kernel<O><<<dim3(gs_cdiv(h, HEAD_ROWS), n), 256, lds, st>>>(
      (const _Float16*)x, ldx, in_bias, in_relu, (const _Float16*)wpack, bias, epilogue, out_scale, out, h, w);
  GS_CHECK_LAUNCH("conv3x3_head");
  return GS_OK;
}

}  // namespace

extern "C" int gs_conv3x3_head(const void* x, int x_stride, const float* in_bias, int in_relu, const void* wpack,
                               const float* bias, int n_out, int epilogue, float out_scale, float* out, int n,
                               int h, int w, gs_stream_t stream) {
  GS_REQUIRE(x && wpack && bias && out, "conv3x3_head: null pointer");
  GS_REQUIRE(x_stride >= 128 && x_stride % 8 == 0, "conv3x3_head: x_stride must be >= 128 and a multiple of 8");
  GS_REQUIRE(n_out == 1 || n_out == 2, "conv3x3_head: n_out must be 1 or 2");
  GS_REQUIRE(epilogue >= 0 && epilogue <= 2, "conv3x3_head: epilogue in {0 none, 1 sigmoid, 2 softplus}");
  GS_REQUIRE(n >= 0 && h > 0 && w > 0, "conv3x3_head: bad shape");
  if (n == 0) return GS_OK;
  hipStream_t st = (hipStream_t)stream;
  return n_out == 1 ? launch_head<1>(x, x_stride, in_bias, in_relu, wpack, bias, epilogue, out_scale, out, n, h, w, st)
                    : launch_head<2>(x, x_stride, in_bias, in_relu, wpack, bias, epilogue, out_scale, out, n, h, w, st);
}
