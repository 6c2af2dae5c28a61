// Fused gate arithmetic of the update operator's ConvGRU (reference src/modules/gru.py:20-33).
// The convolutions stay MIOpen; these two kernels replace the ~17 elementwise passes
// and one 448-channel torch.cat between them:
//   gate_zr : z = sigmoid(convz(hx) + bz + glo_z),  r = sigmoid(convr(hx) + br + glo_r)
//             from ONE fused 448->256 convolution output; writes z and overwrites the first 128
//             channels of the NHWC buffer hx (= [net | inp | corr | flow]) with r * net, which makes
//             hx the input of convq without another cat.
//   gate_q  : q = tanh(convq(hx') + bq + glo_q);  net' = (1 - z) * net + z * q
// All tensors are NHWC fp16 (channel fastest), 8 channels (16 B) per lane, fp32 arithmetic, one
// rounding to fp16 at the end.
#include "common.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float sigm(float x) { return gs_sigmoid(x); }

// zr_pre [n,hw,256]; bias [256] f32; glo [n,256] f32; hx [n,hw,ldx] (first 128 ch = net, in/out); z [n,hw,128]
__global__ __launch_bounds__(256) void gru_gate_zr_kernel(const _Float16* __restrict__ zr_pre,
                                                          const float* __restrict__ bias,
                                                          const float* __restrict__ glo,
                                                          const _Float16* __restrict__ inp_pre,
                                                          _Float16* __restrict__ hx,
                                                          _Float16* __restrict__ z_out, int hw, int ldx,
                                                          size_t total /* n*hw*16 */) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const int c8 = (int)(t & 15);                 // which group of 8 channels (0..15)
  const size_t pix = t >> 4;                    // n*hw + p
  const int n = (int)(pix / hw);
  const half8 zp = *reinterpret_cast<const half8*>(zr_pre + pix * 256 + c8 * 8);
  const half8 rp = *reinterpret_cast<const half8*>(zr_pre + pix * 256 + 128 + c8 * 8);
  half8* hp = reinterpret_cast<half8*>(hx + pix * ldx + c8 * 8);
  const half8 net = *hp;
  half8 zi = {0, 0, 0, 0, 0, 0, 0, 0}, ri = zi;
  if (inp_pre) {   // hoisted convolution over the constant context features (z | r | q channel blocks)
    zi = *reinterpret_cast<const half8*>(inp_pre + pix * 384 + c8 * 8);
    ri = *reinterpret_cast<const half8*>(inp_pre + pix * 384 + 128 + c8 * 8);
  }
  half8 zo, rn;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int c = c8 * 8 + k;
    const float z = sigm((float)zp[k] + (float)zi[k] + bias[c] + glo[(size_t)n * 256 + c]);
    const float r = sigm((float)rp[k] + (float)ri[k] + bias[128 + c] + glo[(size_t)n * 256 + 128 + c]);
    zo[k] = (_Float16)z;
    rn[k] = (_Float16)(r * (float)net[k]);
  }
  *reinterpret_cast<half8*>(z_out + pix * 128 + c8 * 8) = zo;
  *hp = rn;
}

// q_pre, z, net, net_out [n,hw,128]; bias [128]; glo [n,128]
__global__ __launch_bounds__(256) void gru_gate_q_kernel(const _Float16* __restrict__ q_pre,
                                                         const float* __restrict__ bias,
                                                         const float* __restrict__ glo,
                                                         const _Float16* __restrict__ inp_pre,
                                                         const _Float16* __restrict__ z, const _Float16* __restrict__ net,
                                                         _Float16* __restrict__ net_out, int hw, size_t total) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const int c8 = (int)(t & 15);
  const size_t pix = t >> 4;
  const int n = (int)(pix / hw);
  const size_t o = pix * 128 + c8 * 8;
  const half8 qp = *reinterpret_cast<const half8*>(q_pre + o);
  const half8 zz = *reinterpret_cast<const half8*>(z + o);
  const half8 nn = *reinterpret_cast<const half8*>(net + o);
  half8 qi = {0, 0, 0, 0, 0, 0, 0, 0};
  if (inp_pre) qi = *reinterpret_cast<const half8*>(inp_pre + pix * 384 + 256 + c8 * 8);
  half8 out;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int c = c8 * 8 + k;
    const float a = (float)qp[k] + (float)qi[k] + bias[c] + glo[(size_t)n * 128 + c];
    const float q = gs_tanh(a);                                    // tanh (|err| ~1e-7, result goes to fp16)
    const float zf = (float)zz[k];
    out[k] = (_Float16)((1.0f - zf) * (float)nn[k] + zf * q);
  }
  *reinterpret_cast<half8*>(net_out + o) = out;
}

}  // namespace

extern "C" int gs_gru_gate_zr(const void* zr_pre, const float* bias_zr, const float* glo_zr, const void* inp_pre,
                              void* hx, void* z_out, int n, int hw, int ldx, gs_stream_t stream) {
  GS_REQUIRE(zr_pre && bias_zr && glo_zr && hx && z_out, "gru_gate_zr: null pointer");
  GS_REQUIRE(n >= 0 && hw > 0 && ldx >= 128 && ldx % 8 == 0, "gru_gate_zr: bad shape");
  if (n == 0) return GS_OK;
  const size_t total = (size_t)n * hw * 16;
  gru_gate_zr_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(
      (const _Float16*)zr_pre, bias_zr, glo_zr, (const _Float16*)inp_pre, (_Float16*)hx, (_Float16*)z_out, hw, ldx,
      total);
  GS_CHECK_LAUNCH("gru_gate_zr");
  return GS_OK;
}

extern "C" int gs_gru_gate_q(const void* q_pre, const float* bias_q, const float* glo_q, const void* inp_pre,
                             const void* z, const void* net, void* net_out, int n, int hw, gs_stream_t stream) {
  GS_REQUIRE(q_pre && bias_q && glo_q && z && net && net_out, "gru_gate_q: null pointer");
  GS_REQUIRE(n >= 0 && hw > 0, "gru_gate_q: bad shape");
  if (n == 0) return GS_OK;
  const size_t total = (size_t)n * hw * 16;
  gru_gate_q_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(
      (const _Float16*)q_pre, bias_q, glo_q, (const _Float16*)inp_pre, (const _Float16*)z, (const _Float16*)net,
      (_Float16*)net_out, hw, total);
  GS_CHECK_LAUNCH("gru_gate_q");
  return GS_OK;
}

// ---- bias + activation epilogue for the update operator's other convolutions ----------------
// MIOpen's NHWC fp16 convolutions are launched without bias; this one pass adds the bias, applies
// ReLU / sigmoid (PyTorch runs add_ and relu_ as two passes) and can write straight into a channel
// slice of a wider NHWC tensor (the GRU's 448-channel input), which removes the torch.cat pass.
namespace {
template <int ACT, bool BIAS>   // ACT: 0 none, 1 relu, 2 sigmoid
__global__ __launch_bounds__(256) void bias_act_kernel(const _Float16* __restrict__ x, const float* __restrict__ bias,
                                                       _Float16* __restrict__ y, int c8n /* C/8 */, int ldx, int ldy,
                                                       size_t total /* rows*C/8 */) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const int c8 = (int)(t % c8n);
  const size_t row = t / c8n;
  half8 v = *reinterpret_cast<const half8*>(x + row * ldx + c8 * 8);
  if (BIAS || ACT) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float f = (float)v[k];
      if (BIAS) f += bias[c8 * 8 + k];
      if (ACT == 1) f = fmaxf(f, 0.0f);
      if (ACT == 2) f = sigm(f);
      v[k] = (_Float16)f;
    }
  }
  *reinterpret_cast<half8*>(y + row * ldy + c8 * 8) = v;
}

// mean over the edges of each segment (GraphAgg's scatter_mean over source keyframes), CSR form;
// optionally applies the producer convolution's bias + ReLU on the fly and reads a channel slice
__global__ __launch_bounds__(256) void segment_mean_kernel(const _Float16* __restrict__ x, int ldx,
                                                           const float* __restrict__ in_bias, int in_relu,
                                                           const int* __restrict__ off, const int* __restrict__ edges,
                                                           _Float16* __restrict__ out, int hw, int c8n, size_t total) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const int c8 = (int)(t % c8n);
  const size_t sp = t / c8n;                 // seg * hw + p
  const int seg = (int)(sp / hw);
  const int p = (int)(sp - (size_t)seg * hw);
  const int b = off[seg], e = off[seg + 1];
  float bias[8], acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { bias[j] = in_bias ? in_bias[c8 * 8 + j] : 0.0f; acc[j] = 0.0f; }
  const bool touch = (in_bias != nullptr) || in_relu;
  for (int k0 = b; k0 < e; k0 += 4) {           // 4 independent row loads in flight
    half8 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = min(k0 + u, e - 1);
      v[u] = *reinterpret_cast<const half8*>(x + ((size_t)edges[k] * hw + p) * ldx + c8 * 8);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (k0 + u >= e) continue;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float f = (float)v[u][j];
        if (touch) {
          f += bias[j];
          if (in_relu) f = fmaxf(f, 0.0f);
          f = (float)(_Float16)f;              // the activation the reference materialises in fp16
        }
        acc[j] += f;
      }
    }
  }
  const float inv = e > b ? 1.0f / (float)(e - b) : 0.0f;
  half8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (_Float16)(acc[j] * inv);
  reinterpret_cast<half8*>(out)[t] = o;
}

// partial[n, chunk, 128] = sum over the chunk's pixels of sigmoid(w_pre + bias) * net
__global__ __launch_bounds__(256) void glo_pool_kernel(const _Float16* __restrict__ w_pre, const float* __restrict__ bias,
                                                       const _Float16* __restrict__ net, float* __restrict__ partial,
                                                       int hw, int nchunk) {
  const int n = blockIdx.y, chunk = blockIdx.x;
  const int c8 = threadIdx.x & 15, lane = threadIdx.x >> 4;          // 16 channel groups x 16 pixel lanes
  const int p0 = (int)((long)hw * chunk / nchunk), p1 = (int)((long)hw * (chunk + 1) / nchunk);
  float b[8], acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { b[k] = bias[c8 * 8 + k]; acc[k] = 0.0f; }
  for (int p = p0 + lane; p < p1; p += 16) {
    const size_t o = ((size_t)n * hw + p) * 128 + c8 * 8;
    const half8 w = *reinterpret_cast<const half8*>(w_pre + o);
    const half8 h = *reinterpret_cast<const half8*>(net + o);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      // the reference rounds sigmoid(.) and the product to fp16 (autocast elementwise ops)
      const _Float16 g = (_Float16)sigm((float)(_Float16)((float)w[k] + b[k]));
      acc[k] += (float)(_Float16)((float)g * (float)h[k]);
    }
  }
  __shared__ float red[16][129];
#pragma unroll
  for (int k = 0; k < 8; ++k) red[lane][c8 * 8 + k] = acc[k];
  __syncthreads();
  if (threadIdx.x < 128) {
    float s = 0.0f;
#pragma unroll
    for (int l = 0; l < 16; ++l) s += red[l][threadIdx.x];
    partial[((size_t)n * nchunk + chunk) * 128 + threadIdx.x] = s;
  }
}

// The same partial sums WITHOUT the w_pre round trip: the 1x1 convolution w(net) (128 -> 128) runs on MFMA inside this
// kernel (weights = A operand, 32 KB of fragments in LDS; pixels = B operand straight from their rows, as in conv1x1.hip)
// and sigmoid(. + bias) * net is pooled in its epilogue -- net is read once from HBM (the second read, in row layout,
// hits L1/L2), nothing but 512 B of partial sums per 32 pixels is written: 23 MB instead of 92 + 92 + 92 + 92 MB moved
// at 75 edges x 60 x 80.  A wave owns 32 consecutive pixels of ONE edge (blocks never straddle edges);
// partial[(edge, block), 128] in a fixed order -> deterministic.  Rounding: fp16(conv + bias) once (as autocast's
// conv2d does), fp16 sigmoid, fp16 product, fp32 sums.
typedef _Float16 half4g __attribute__((ext_vector_type(4)));
typedef float float16g __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void glo_conv_pool_kernel(const _Float16* __restrict__ net, int ldx,
                                                            const _Float16* __restrict__ wpack,
                                                            const float* __restrict__ bias, float* __restrict__ partial,
                                                            int hw, int bpi, size_t nblk) {
  extern __shared__ _Float16 wl[];            // [4][8][64][8] weight fragments, then 4 x [32][72] wave tiles
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  for (int i = tid; i < 4 * 8 * 64; i += 256) reinterpret_cast<half8*>(wl)[i] = reinterpret_cast<const half8*>(wpack)[i];
  __syncthreads();
  constexpr int TS = 72;
  _Float16* tile = wl + 4 * 8 * 512 + wv * 32 * TS;
  float* sbias = reinterpret_cast<float*>(wl + 4 * 8 * 512 + 4 * 32 * TS);      // [128]: per-lane global loads of the
  if (tid < 128) sbias[tid] = bias[tid];                                        // bias inside the block loop cost a
  __syncthreads();                                                              // memory round trip per use
  const int kh = 8 * (lane >> 5);
  const int piece = lane & 7;
  for (size_t blk = (size_t)blockIdx.x * 4 + wv; blk < nblk; blk += (size_t)gridDim.x * 4) {
    const size_t img = blk / bpi;
    const int b0 = (int)(blk - img * bpi) * 32;
    const int pin = b0 + (lane & 31);
    const _Float16* xr = net + (img * hw + (pin < hw ? pin : hw - 1)) * (size_t)ldx + kh;
    half8 bfr[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) bfr[ks] = *reinterpret_cast<const half8*>(xr + 16 * ks);
    // the same rows again in row layout for the epilogue's product (L1 / L2 hits), requested now, used after the MFMAs
    half8 hrow[2][4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int pxr = b0 + it * 8 + (lane >> 3);
      const _Float16* hr = net + (img * hw + (pxr < hw ? pxr : hw - 1)) * (size_t)ldx + piece * 8;
      hrow[0][it] = *reinterpret_cast<const half8*>(hr);
      hrow[1][it] = *reinterpret_cast<const half8*>(hr + 64);
    }
    float16g c[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int e = 0; e < 16; ++e) c[q][e] = 0.0f;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const half8 a = *reinterpret_cast<const half8*>(wl + ((size_t)(q * 8 + ks) * 64 + lane) * 8);
        c[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bfr[ks], c[q], 0, 0, 0);
      }
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
#pragma unroll
      for (int qq = 0; qq < 2; ++qq)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int chl = qq * 32 + 8 * g + 4 * (lane >> 5);
          half4g o;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            o[e] = (_Float16)sigm((float)(_Float16)(c[2 * pr + qq][4 * g + e] + sbias[pr * 64 + chl + e]));
          *reinterpret_cast<half4g*>(tile + (lane & 31) * TS + chl) = o;
        }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int pxr = it * 8 + (lane >> 3);
        const half8 g8 = *reinterpret_cast<const half8*>(tile + pxr * TS + piece * 8);
        const half8 h8 = hrow[pr][it];
        const float live = (b0 + pxr < hw) ? 1.0f : 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += live * (float)(_Float16)((float)g8[k] * (float)h8[k]);
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        acc[k] += __shfl_xor(acc[k], 8);
        acc[k] += __shfl_xor(acc[k], 16);
        acc[k] += __shfl_xor(acc[k], 32);
      }
      if (lane < 8) {
        float* dst = partial + blk * 128 + pr * 64 + lane * 8;
        *reinterpret_cast<float4*>(dst) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        *reinterpret_cast<float4*>(dst + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// glo = mean -> the three 1x1 "global context" convolutions (128x128 mat-vecs) of the ConvGRU
__global__ __launch_bounds__(384) void glo_heads_kernel(const float* __restrict__ partial, int nchunk, float inv_hw,
                                                        const _Float16* __restrict__ wz, const _Float16* __restrict__ wr,
                                                        const _Float16* __restrict__ wq, const float* __restrict__ bz,
                                                        const float* __restrict__ br, const float* __restrict__ bq,
                                                        float* __restrict__ gzr, float* __restrict__ gq) {
  const int n = blockIdx.x, t = threadIdx.x;
  __shared__ float g[128];
  __shared__ float part3[3][128];
  {
    // 384 threads: thread (third, channel) sums every third chunk; thirds combined in a fixed order
    const int third = t >> 7, ch = t & 127;
    float s = 0.0f;
    // 10 loads in flight per round trip, summed in the same order as one by one (a rolled loop paid one L2 latency per chunk:
    // 50 of them in a 75-workgroup launch = most of the kernel's 17 us)
    for (int c0 = third; c0 < nchunk; c0 += 30) {
      float v[10];
#pragma unroll
      for (int u = 0; u < 10; ++u) {
        const int c = c0 + 3 * u;
        v[u] = c < nchunk ? partial[((size_t)n * nchunk + c) * 128 + ch] : 0.0f;
      }
#pragma unroll
      for (int u = 0; u < 10; ++u)
        if (c0 + 3 * u < nchunk) s += v[u];
    }
    part3[third][ch] = s;
  }
  __syncthreads();
  if (t < 128) g[t] = (float)(_Float16)(((part3[0][t] + part3[1][t]) + part3[2][t]) * inv_hw);
  __syncthreads();
  const int head = t >> 7, o = t & 127;
  const _Float16* w = (head == 0 ? wz : head == 1 ? wr : wq) + (size_t)o * 128;
  float s = (head == 0 ? bz : head == 1 ? br : bq)[o];
#pragma unroll 4
  for (int k8 = 0; k8 < 16; ++k8) {
    const half8 v = *reinterpret_cast<const half8*>(w + k8 * 8);
#pragma unroll
    for (int k = 0; k < 8; ++k) s = fmaf((float)v[k], g[k8 * 8 + k], s);
  }
  s = (float)(_Float16)s;
  if (head < 2) gzr[(size_t)n * 256 + head * 128 + o] = s;
  else gq[(size_t)n * 128 + o] = s;
}
}  // namespace

extern "C" int gs_bias_act(const void* x, const float* bias, void* y, int rows, int channels, int x_stride,
                           int y_stride, int act, gs_stream_t stream) {
  GS_REQUIRE(x && y, "bias_act: null pointer");
  GS_REQUIRE(rows >= 0 && channels > 0 && channels % 8 == 0, "bias_act: channels must be a multiple of 8");
  GS_REQUIRE(x_stride >= channels && x_stride % 8 == 0, "bias_act: x_stride must be >= channels and a multiple of 8");
  GS_REQUIRE(y_stride >= channels && y_stride % 8 == 0, "bias_act: y_stride must be >= channels and a multiple of 8");
  GS_REQUIRE(act >= 0 && act <= 2, "bias_act: act in {0 none, 1 relu, 2 sigmoid}");
  if (rows == 0) return GS_OK;
  const size_t total = (size_t)rows * (channels / 8);
  const unsigned grid = (unsigned)((total + 255) / 256);
  hipStream_t st = (hipStream_t)stream;
  const _Float16* xi = (const _Float16*)x;
  _Float16* yo = (_Float16*)y;
  const int c8n = channels / 8;
#define GS_BA(A, B) bias_act_kernel<A, B><<<grid, 256, 0, st>>>(xi, bias, yo, c8n, x_stride, y_stride, total)
  if (bias) { if (act == 0) GS_BA(0, true); else if (act == 1) GS_BA(1, true); else GS_BA(2, true); }
  else      { if (act == 0) GS_BA(0, false); else if (act == 1) GS_BA(1, false); else GS_BA(2, false); }
#undef GS_BA
  GS_CHECK_LAUNCH("bias_act");
  return GS_OK;
}

extern "C" int gs_segment_mean(const void* x, int x_stride, const float* in_bias, int in_relu, const int* seg_offsets,
                               const int* seg_edges, void* out, int n_seg, int hw, int channels, gs_stream_t stream) {
  GS_REQUIRE(x && seg_offsets && seg_edges && out, "segment_mean: null pointer");
  GS_REQUIRE(n_seg >= 0 && hw > 0 && channels > 0 && channels % 8 == 0, "segment_mean: channels must be a multiple of 8");
  GS_REQUIRE(x_stride >= channels && x_stride % 8 == 0, "segment_mean: x_stride must be >= channels, multiple of 8");
  if (n_seg == 0) return GS_OK;
  const size_t total = (size_t)n_seg * hw * (channels / 8);
  segment_mean_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(
      (const _Float16*)x, x_stride, in_bias, in_relu, seg_offsets, seg_edges, (_Float16*)out, hw, channels / 8, total);
  GS_CHECK_LAUNCH("segment_mean");
  return GS_OK;
}

constexpr int GLO_CHUNKS = 32;   // pixel chunks per edge: 2400 workgroups at E = 75 (8 chunks left the GPU a third empty)
extern "C" size_t gs_gru_glo_workspace_bytes(int n) { return (size_t)(n > 0 ? n : 0) * GLO_CHUNKS * 128 * sizeof(float); }

extern "C" int gs_gru_glo(const void* w_pre, const float* w_bias, const void* net, const void* wz, const void* wr,
                          const void* wq, const float* bz, const float* br, const float* bq, float* gzr, float* gq,
                          int n, int hw, void* workspace, size_t workspace_bytes, gs_stream_t stream) {
  GS_REQUIRE(w_pre && w_bias && net && wz && wr && wq && bz && br && bq && gzr && gq, "gru_glo: null pointer");
  GS_REQUIRE(n >= 0 && hw > 0, "gru_glo: bad shape");
  if (n == 0) return GS_OK;
  GS_REQUIRE(workspace && workspace_bytes >= gs_gru_glo_workspace_bytes(n), "gru_glo: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int nchunk = GLO_CHUNKS;
  glo_pool_kernel<<<dim3(nchunk, n), 256, 0, st>>>((const _Float16*)w_pre, w_bias, (const _Float16*)net,
                                                   (float*)workspace, hw, nchunk);
  GS_CHECK_LAUNCH("gru_glo pool");
  glo_heads_kernel<<<n, 384, 0, st>>>((const float*)workspace, nchunk, 1.0f / (float)hw, (const _Float16*)wz,
                                      (const _Float16*)wr, (const _Float16*)wq, bz, br, bq, gzr, gq);
  GS_CHECK_LAUNCH("gru_glo heads");
  return GS_OK;
}

extern "C" size_t gs_gru_glo_fused_workspace_bytes(int n, int hw) {
  if (n <= 0 || hw <= 0) return 0;
  return (size_t)n * ((hw + 31) / 32) * 128 * sizeof(float);
}

// gs_gru_glo with the w convolution inside: w_pack = gs_conv1x1's weight image of gru.w ([4][8][64][8] halves)
extern "C" int gs_gru_glo_fused(const void* net, int net_stride, const void* w_pack, const float* w_bias, const void* wz,
                                const void* wr, const void* wq, const float* bz, const float* br, const float* bq,
                                float* gzr, float* gq, int n, int hw, void* workspace, size_t workspace_bytes,
                                gs_stream_t stream) {
  GS_REQUIRE(net && w_pack && w_bias && wz && wr && wq && bz && br && bq && gzr && gq, "gru_glo_fused: null pointer");
  GS_REQUIRE(n >= 0 && hw > 0 && net_stride >= 128 && net_stride % 8 == 0, "gru_glo_fused: bad shape");
  GS_REQUIRE((((size_t)net | (size_t)w_pack) & 15) == 0, "gru_glo_fused: net / w_pack must be 16-byte aligned");
  if (n == 0) return GS_OK;
  GS_REQUIRE(workspace && workspace_bytes >= gs_gru_glo_fused_workspace_bytes(n, hw), "gru_glo_fused: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int bpi = (hw + 31) / 32;
  const size_t nblk = (size_t)n * bpi;
  const size_t lds = (size_t)(4 * 8 * 512 + 4 * 32 * 72) * sizeof(_Float16) + 128 * sizeof(float);   // 32 + 18 + 0.5 KB
  static GsLdsLimit limit;
  if (int rc = limit.raise((const void*)glo_conv_pool_kernel, lds, "gru_glo_fused")) return rc;
  size_t grid = 256 * 3;
  if (grid > (nblk + 3) / 4) grid = (nblk + 3) / 4;
  glo_conv_pool_kernel<<<(unsigned)grid, 256, lds, st>>>((const _Float16*)net, net_stride, (const _Float16*)w_pack, w_bias,
                                                         (float*)workspace, hw, bpi, nblk);
  GS_CHECK_LAUNCH("gru_glo_fused pool");
  glo_heads_kernel<<<n, 384, 0, st>>>((const float*)workspace, bpi, 1.0f / (float)hw, (const _Float16*)wz,
                                      (const _Float16*)wr, (const _Float16*)wq, bz, br, bq, gzr, gq);
  GS_CHECK_LAUNCH("gru_glo_fused heads");
  return GS_OK;
}

// ---- FactorGraph.update glue around the operator (reference src/factor_graph.py:201-207, 222-247) ----
namespace {
// motion features: clamp([coords1 - coords0, target - coords1], +-64) as the NHWC fp16 tensor the flow
// encoder consumes (the reference builds it with sub, sub, cat, permute, clamp + the autocast cast)
__global__ __launch_bounds__(256) void motion_features_kernel(const float2* __restrict__ coords1,
                                                              const float2* __restrict__ target,
                                                              _Float16* __restrict__ out, int hw, int w, size_t total) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const int p = (int)(t % hw);
  const float gx = (float)(p % w), gy = (float)(p / w);
  const float2 c = coords1[t], tg = target[t];
  typedef _Float16 half4m __attribute__((ext_vector_type(4)));
  half4m o;
  o[0] = (_Float16)fminf(fmaxf(c.x - gx, -64.0f), 64.0f);
  o[1] = (_Float16)fminf(fmaxf(c.y - gy, -64.0f), 64.0f);
  o[2] = (_Float16)fminf(fmaxf(tg.x - c.x, -64.0f), 64.0f);
  o[3] = (_Float16)fminf(fmaxf(tg.y - c.y, -64.0f), 64.0f);
  reinterpret_cast<half4m*>(out)[t] = o;
}

// damping rows of a BA call (reference src/factor_graph.py:228,244): damping_buf[uniq] = eta, then
// out = scale * damping_buf[index] + eps -- index_put + gather + mul + add in one pass.  inv[k] = the row of `eta` that
// belongs to frame index[k], or -1 if the operator produced none for it this update (its buffered value is used).
__global__ __launch_bounds__(256) void damping_rows_kernel(const float* __restrict__ eta, const int* __restrict__ inv,
                                                           const int64_t* __restrict__ index,
                                                           float* __restrict__ damping_buf, float* __restrict__ out,
                                                           int hw, float scale, float eps, size_t total) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const int k = (int)(t / hw), p = (int)(t - (size_t)k * hw);
  const int m = inv[k];
  float* slot = damping_buf + (size_t)index[k] * hw + p;
  float v;
  if (m >= 0) { v = eta[(size_t)m * hw + p]; *slot = v; }
  else v = *slot;
  out[t] = scale * v + eps;
}

// target = coords1 + delta (kept as [E,h,w,2] state) and the BA-layout copies target/weight [E,2,h,w]
__global__ __launch_bounds__(256) void ba_inputs_kernel(const float2* __restrict__ coords1, const float2* __restrict__ delta,
                                                        const float2* __restrict__ weight, float2* __restrict__ target,
                                                        float* __restrict__ ba_target, float* __restrict__ ba_weight,
                                                        int hw, size_t total) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const size_t e = t / hw;
  const int p = (int)(t - e * hw);
  const float2 c = coords1[t], d = delta[t], wt = weight[t];
  const float2 tg = make_float2(c.x + d.x, c.y + d.y);
  target[t] = tg;
  float* bt = ba_target + e * 2 * hw + p;
  bt[0] = tg.x; bt[hw] = tg.y;
  float* bw = ba_weight + e * 2 * hw + p;
  bw[0] = wt.x; bw[hw] = wt.y;
}
}  // namespace

extern "C" int gs_motion_features(const float* coords1, const float* target, void* out, int n, int h, int w,
                                  gs_stream_t stream) {
  GS_REQUIRE(coords1 && target && out, "motion_features: null pointer");
  GS_REQUIRE(n >= 0 && h > 0 && w > 0, "motion_features: bad shape");
  if (n == 0) return GS_OK;
  const size_t total = (size_t)n * h * w;
  motion_features_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(
      (const float2*)coords1, (const float2*)target, (_Float16*)out, h * w, w, total);
  GS_CHECK_LAUNCH("motion_features");
  return GS_OK;
}

extern "C" int gs_ba_inputs(const float* coords1, const float* delta, const float* weight, float* target,
                            float* ba_target, float* ba_weight, int n, int h, int w, gs_stream_t stream) {
  GS_REQUIRE(coords1 && delta && weight && target && ba_target && ba_weight, "ba_inputs: null pointer");
  GS_REQUIRE(n >= 0 && h > 0 && w > 0, "ba_inputs: bad shape");
  if (n == 0) return GS_OK;
  const size_t total = (size_t)n * h * w;
  ba_inputs_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(
      (const float2*)coords1, (const float2*)delta, (const float2*)weight, (float2*)target, ba_target, ba_weight,
      h * w, total);
  GS_CHECK_LAUNCH("ba_inputs");
  return GS_OK;
}

extern "C" int gs_damping_rows(const float* eta, const int* inv, const int64_t* index, float* damping_buf, float* out,
                               int n_rows, int hw, float scale, float eps, gs_stream_t stream) {
  GS_REQUIRE(inv && index && damping_buf && out, "damping_rows: null pointer");
  GS_REQUIRE(n_rows >= 0 && hw > 0, "damping_rows: bad shape");
  if (n_rows == 0) return GS_OK;
  const size_t total = (size_t)n_rows * hw;
  damping_rows_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(eta, inv, index, damping_buf, out,
                                                                                       hw, scale, eps, total);
  GS_CHECK_LAUNCH("damping_rows");
  return GS_OK;
}
