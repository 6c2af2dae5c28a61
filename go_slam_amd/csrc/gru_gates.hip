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

__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + __expf(-x)); }

// zr_pre [n,hw,256]; bias [256] f32; glo [n,256] f32; hx [n,hw,ldx] (first 128 ch = net, in/out); z [n,hw,128]
__global__ __launch_bounds__(256) void gru_gate_zr_kernel(const _Float16* __restrict__ zr_pre,
                                                          const float* __restrict__ bias,
                                                          const float* __restrict__ glo, _Float16* __restrict__ hx,
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
  half8 zo, rn;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int c = c8 * 8 + k;
    const float z = sigm((float)zp[k] + bias[c] + glo[(size_t)n * 256 + c]);
    const float r = sigm((float)rp[k] + bias[128 + c] + glo[(size_t)n * 256 + 128 + c]);
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
  half8 out;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int c = c8 * 8 + k;
    const float a = (float)qp[k] + bias[c] + glo[(size_t)n * 128 + c];
    const float q = 1.0f - 2.0f / (1.0f + __expf(2.0f * a));      // tanh (|err| ~1e-7, result goes to fp16)
    const float zf = (float)zz[k];
    out[k] = (_Float16)((1.0f - zf) * (float)nn[k] + zf * q);
  }
  *reinterpret_cast<half8*>(net_out + o) = out;
}

}  // namespace

extern "C" int gs_gru_gate_zr(const void* zr_pre, const float* bias_zr, const float* glo_zr, void* hx, void* z_out,
                              int n, int hw, int ldx, gs_stream_t stream) {
  GS_REQUIRE(zr_pre && bias_zr && glo_zr && hx && z_out, "gru_gate_zr: null pointer");
  GS_REQUIRE(n >= 0 && hw > 0 && ldx >= 128 && ldx % 8 == 0, "gru_gate_zr: bad shape");
  if (n == 0) return GS_OK;
  const size_t total = (size_t)n * hw * 16;
  gru_gate_zr_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(
      (const _Float16*)zr_pre, bias_zr, glo_zr, (_Float16*)hx, (_Float16*)z_out, hw, ldx, total);
  GS_CHECK_LAUNCH("gru_gate_zr");
  return GS_OK;
}

extern "C" int gs_gru_gate_q(const void* q_pre, const float* bias_q, const float* glo_q, const void* z,
                             const void* net, void* net_out, int n, int hw, gs_stream_t stream) {
  GS_REQUIRE(q_pre && bias_q && glo_q && z && net && net_out, "gru_gate_q: null pointer");
  GS_REQUIRE(n >= 0 && hw > 0, "gru_gate_q: bad shape");
  if (n == 0) return GS_OK;
  const size_t total = (size_t)n * hw * 16;
  gru_gate_q_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(
      (const _Float16*)q_pre, bias_q, glo_q, (const _Float16*)z, (const _Float16*)net, (_Float16*)net_out, hw, total);
  GS_CHECK_LAUNCH("gru_gate_q");
  return GS_OK;
}

// ---- bias + activation epilogue for the update operator's other convolutions ----------------
// MIOpen's NHWC fp16 convolutions are launched without bias; this one in-place pass adds the
// bias and applies ReLU / sigmoid (PyTorch runs add_ and relu_ as two separate passes).
namespace {
template <int ACT>   // 0 none, 1 relu, 2 sigmoid
__global__ __launch_bounds__(256) void bias_act_kernel(_Float16* __restrict__ x, const float* __restrict__ bias,
                                                       int c8n /* C/8 */, size_t total /* rows*C/8 */) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const int c8 = (int)(t % c8n);
  half8* p = reinterpret_cast<half8*>(x) + t;
  half8 v = *p;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    float f = (float)v[k] + bias[c8 * 8 + k];
    if (ACT == 1) f = fmaxf(f, 0.0f);
    if (ACT == 2) f = sigm(f);
    v[k] = (_Float16)f;
  }
  *p = v;
}
}  // namespace

extern "C" int gs_bias_act(void* x, const float* bias, int rows, int channels, int act, gs_stream_t stream) {
  GS_REQUIRE(x && bias, "bias_act: null pointer");
  GS_REQUIRE(rows >= 0 && channels > 0 && channels % 8 == 0, "bias_act: channels must be a multiple of 8");
  GS_REQUIRE(act >= 0 && act <= 2, "bias_act: act in {0 none, 1 relu, 2 sigmoid}");
  if (rows == 0) return GS_OK;
  const size_t total = (size_t)rows * (channels / 8);
  const unsigned grid = (unsigned)((total + 255) / 256);
  hipStream_t st = (hipStream_t)stream;
  if (act == 0) bias_act_kernel<0><<<grid, 256, 0, st>>>((_Float16*)x, bias, channels / 8, total);
  else if (act == 1) bias_act_kernel<1><<<grid, 256, 0, st>>>((_Float16*)x, bias, channels / 8, total);
  else bias_act_kernel<2><<<grid, 256, 0, st>>>((_Float16*)x, bias, channels / 8, total);
  GS_CHECK_LAUNCH("bias_act");
  return GS_OK;
}
