// Backward of the fused colour MLP (reference: tiny-cuda-nn FullyFusedMLP backward, reached through
// `tcnn.Network` at src/InstantNeuS.py:192-217; 67(->80) -> 64 -> 64 -> 3(->16), ReLU, sigmoid output).
//
// One kernel replaces the forward recompute (2 GEMMs), 3 data-gradient GEMMs, 3 weight-gradient split-K
// batched GEMMs and ~12 elementwise passes over [points, 64] tensors of the hipBLASLt formulation:
//   H1 = relu(W1 X), H2 = relu(W2 H1)                          (recomputed exactly as neus_mlp_kernel does)
//   dH2 = (W3^T dpre) * [H2 > 0],  dH1 = (W2^T dH2) * [H1 > 0],  dX = W1^T dH1
//   dW3 += dpre H2^T,  dW2 += dH2 H1^T,  dW1 += dH1 X^T
// all on v_mfma_f32_32x32x16_f16 with 32 points per wave-block as the N (data GEMMs) or K (weight GEMMs)
// dimension.  Every operand is needed twice, once [point][neuron] (B operand of the next data GEMM) and
// once [neuron][point] (A/B operand of a weight GEMM, whose reduction runs over points), so each
// accumulator tile is written to two wave-private LDS tiles; the 40 KB of pre-packed weight fragments
// (W1, W2, W3^T, W2^T, W1^T) sit in LDS for the whole launch; the weight-gradient accumulators (192
// registers) stay in the register file across the wave's blocks and are reduced once per workgroup.
// Gradients travel in fp16 multiplied by the caller's loss scale (tcnn: 128), as in the reference.
#include "common.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float float16v __attribute__((ext_vector_type(16)));

constexpr int HS2 = 72;            // [point][neuron] tile stride (halves)
constexpr int TS = 40;             // [row][point] tile stride (halves): 32 points + pad, 80-byte rows
constexpr int OFF_HS = 0;          // 32 x 72
constexpr int OFF_XT = 2304;       // 96 x 40 (rows 80..95 stay zero); doubles as the dX staging tile [32][88]
constexpr int OFF_TA = OFF_XT + 3840;   // 64 x 40: H1^T, then dH1^T
constexpr int OFF_TB = OFF_TA + 2560;   // 64 x 40: H2^T, then dH2^T
constexpr int OFF_DP = OFF_TB + 2560;   // 32 x 40: dpre^T (rows 3..31 stay zero)
constexpr int WAVE_LDS = OFF_DP + 1280; // 12544 halves = 25088 B per wave
constexpr int NFRAG = 40;          // W1 (10) | W2 (8) | W3^T (2) | W2^T (8) | W1^T (12), 1 KB each
constexpr int F_W1 = 0, F_W2 = 10, F_W3T = 18, F_W2T = 20, F_W1T = 28;
constexpr int NPARAM = 64 * 80 + 64 * 64 + 16 * 64;   // 10240, tcnn's layout

__device__ __forceinline__ half8 ld8(const _Float16* p) { return *reinterpret_cast<const half8*>(p); }
__device__ __forceinline__ void wsync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ void zero16(float16v& c) {
#pragma unroll
  for (int e = 0; e < 16; ++e) c[e] = 0.f;
}

// accumulator tile (rows = neurons 32*mt.., cols = the wave's 32 points) -> fp16 values `f(c, reg)`, stored
// both as [point][neuron] (8-byte groups) and as [neuron][point] (2-byte, consecutive lanes consecutive)
template <typename F>
__device__ __forceinline__ void store_tiles(const float16v& c, int mt, int lane, _Float16* pn, _Float16* np_, F f) {
  const int pt = lane & 31, hf = lane >> 5;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    half4 pk;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const _Float16 h = f(c[4 * q + k], 4 * q + k);
      pk[k] = h;
      if (np_) np_[(32 * mt + 8 * q + 4 * hf + k) * TS + pt] = h;
    }
    if (pn) *reinterpret_cast<half4*>(pn + pt * HS2 + 32 * mt + 8 * q + 4 * hf) = pk;
  }
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void neus_mlp_bwd_kernel(const _Float16* __restrict__ X, const _Float16* __restrict__ wpack,
                         const float* __restrict__ d_rgb, const _Float16* __restrict__ rgb, float loss_scale,
                         _Float16* __restrict__ dX, float* __restrict__ partial, int np, int nblk) {
  extern __shared__ __attribute__((aligned(16))) _Float16 sm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  _Float16* wl = sm;
  _Float16* wb = sm + NFRAG * 512 + wave * WAVE_LDS;
  _Float16 *hs = wb + OFF_HS, *XT = wb + OFF_XT, *TA = wb + OFF_TA, *TB = wb + OFF_TB, *DP = wb + OFF_DP;
  for (int i = tid; i < NFRAG * 64; i += 256) reinterpret_cast<half8*>(wl)[i] = reinterpret_cast<const half8*>(wpack)[i];
  {
    const half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = lane; i < WAVE_LDS / 8; i += 64) reinterpret_cast<half8*>(wb)[i] = z;
  }
  __syncthreads();
  const int r = lane & 31, hf = lane >> 5, kh = 8 * hf;
  auto A = [&](int frag) { return ld8(wl + ((size_t)frag * 64 + lane) * 8); };

  float16v cw1[2][3], cw2[2][2], cw3[2];
#pragma unroll
  for (int a = 0; a < 2; ++a) {
#pragma unroll
    for (int b = 0; b < 3; ++b) zero16(cw1[a][b]);
#pragma unroll
    for (int b = 0; b < 2; ++b) zero16(cw2[a][b]);
    zero16(cw3[a]);
  }

  // A wave runs its blocks one after the other, alone on its SIMD (the 192 weight-gradient accumulators): nothing
  // hides a global round trip.  The NEXT block's inputs (X fragments, d_rgb, rgb) are therefore requested at the head
  // of the current block and consumed one iteration later.
  half8 xn[5];
  float drn[3], yn[3];
  auto request = [&](int blk) {
    const int pr = blk * 32 + r;
    const size_t pcn = (size_t)(pr < np ? pr : np - 1);
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) xn[ks] = ld8(X + pcn * 80 + 16 * ks + kh);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      drn[k] = d_rgb[pcn * 3 + k];
      yn[k] = rgb ? (float)rgb[pcn * 3 + k] : 0.0f;
    }
  };
  const int blk0 = blockIdx.x * 4 + wave, bstride = gridDim.x * 4;
  if (blk0 < nblk) request(blk0);
  for (int blk = blk0; blk < nblk; blk += bstride) {
    const int p0 = blk * 32;
    const bool valid = p0 + r < np;
    // ---- X: B fragments from the [point][80] rows, and X^T into its tile
    half8 xb[5];
    float dr[3], yv[3];
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) xb[ks] = xn[ks];
#pragma unroll
    for (int k = 0; k < 3; ++k) { dr[k] = drn[k]; yv[k] = yn[k]; }
    if (blk + bstride < nblk) request(blk + bstride);
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) {
#pragma unroll
      for (int e = 0; e < 8; ++e) XT[(16 * ks + kh + e) * TS + r] = xb[ks][e];
    }
    // ---- layer 1 (recompute), H1 -> hs [pt][n] and TA [n][pt]
    unsigned m1 = 0, m2 = 0;                       // ReLU masks, bit 16*mt + reg
    {
      float16v c[2];
      zero16(c[0]); zero16(c[1]);
#pragma unroll
      for (int ks = 0; ks < 5; ++ks)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) c[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A(F_W1 + mt * 5 + ks), xb[ks], c[mt], 0, 0, 0);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
        store_tiles(c[mt], mt, lane, hs, TA, [&](float v, int reg) {
          if (v > 0.0f) m1 |= 1u << (16 * mt + reg);
          return (_Float16)fmaxf(v, 0.0f);
        });
    }
    wsync();
    // ---- layer 2 (recompute), H2 -> TB [n][pt]
    {
      half8 b[4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) b[ks] = ld8(hs + r * HS2 + 16 * ks + kh);
      float16v c[2];
      zero16(c[0]); zero16(c[1]);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) c[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A(F_W2 + mt * 4 + ks), b[ks], c[mt], 0, 0, 0);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
        store_tiles(c[mt], mt, lane, (_Float16*)nullptr, TB, [&](float v, int reg) {
          if (v > 0.0f) m2 |= 1u << (16 * mt + reg);
          return (_Float16)fmaxf(v, 0.0f);
        });
    }
    // ---- dpre = d_rgb * y (1 - y) * loss_scale (sigmoid'), as the B fragment of W3^T dpre and as dpre^T rows
    half8 bdp = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hf == 0) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float y = yv[k];
        const float dact = rgb ? y * (1.0f - y) : 1.0f;     // rgb == NULL: d_rgb is w.r.t. the raw outputs
        const float v = valid ? dr[k] * dact * loss_scale : 0.0f;
        bdp[k] = (_Float16)v;
        DP[k * TS + r] = bdp[k];
      }
    }
    wsync();
    // ---- dW3 += dpre^T-rows x H2  (K = the 32 points)
#pragma unroll
    for (int kst = 0; kst < 2; ++kst) {
      const half8 a = ld8(DP + r * TS + 16 * kst + kh);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
        cw3[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, ld8(TB + (32 * nt + r) * TS + 16 * kst + kh), cw3[nt], 0, 0, 0);
    }
    // ---- dH2 = (W3^T dpre) * [H2 > 0]
    float16v cd[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      zero16(cd[mt]);
      cd[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A(F_W3T + mt), bdp, cd[mt], 0, 0, 0);
    }
    wsync();                                       // dW3 has read H2^T: TB may be overwritten
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
      store_tiles(cd[mt], mt, lane, hs, TB, [&](float v, int reg) {
        return (_Float16)(((m2 >> (16 * mt + reg)) & 1u) ? v : 0.0f);
      });
    wsync();
    // ---- dW2 += dH2 x H1^T-rows
#pragma unroll
    for (int kst = 0; kst < 2; ++kst) {
      half8 bh[2];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) bh[nt] = ld8(TA + (32 * nt + r) * TS + 16 * kst + kh);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const half8 a = ld8(TB + (32 * mt + r) * TS + 16 * kst + kh);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) cw2[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bh[nt], cw2[mt][nt], 0, 0, 0);
      }
    }
    // ---- dH1 = (W2^T dH2) * [H1 > 0]
    {
      half8 b[4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) b[ks] = ld8(hs + r * HS2 + 16 * ks + kh);
      zero16(cd[0]); zero16(cd[1]);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) cd[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A(F_W2T + mt * 4 + ks), b[ks], cd[mt], 0, 0, 0);
    }
    wsync();                                       // dW2 has read H1^T, dH1 has read hs
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
      store_tiles(cd[mt], mt, lane, hs, TA, [&](float v, int reg) {
        return (_Float16)(((m1 >> (16 * mt + reg)) & 1u) ? v : 0.0f);
      });
    wsync();
    // ---- dW1 += dH1 x X^T-rows
#pragma unroll
    for (int kst = 0; kst < 2; ++kst) {
      half8 bx[3];
#pragma unroll
      for (int nt = 0; nt < 3; ++nt) bx[nt] = ld8(XT + (32 * nt + r) * TS + 16 * kst + kh);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const half8 a = ld8(TA + (32 * mt + r) * TS + 16 * kst + kh);
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) cw1[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bx[nt], cw1[mt][nt], 0, 0, 0);
      }
    }
    // ---- dX = W1^T dH1  (rows = the 80 input features)
    {
      half8 b[4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) b[ks] = ld8(hs + r * HS2 + 16 * ks + kh);
      float16v cx[3];
#pragma unroll
      for (int mt = 0; mt < 3; ++mt) {
        zero16(cx[mt]);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) cx[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A(F_W1T + mt * 4 + ks), b[ks], cx[mt], 0, 0, 0);
      }
      wsync();                                     // dW1 has read X^T: its tile becomes the [32][88] staging tile
#pragma unroll
      for (int mt = 0; mt < 3; ++mt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int f = 32 * mt + 8 * q + 4 * hf;
          if (f < 80) {
            half4 pk;
#pragma unroll
            for (int k = 0; k < 4; ++k) pk[k] = (_Float16)cx[mt][4 * q + k];
            *reinterpret_cast<half4*>(XT + r * 88 + f) = pk;
          }
        }
    }
    wsync();
#pragma unroll
    for (int it = 0; it < 5; ++it) {               // 32 points x 10 pieces of 16 bytes, contiguous in dX
      const int idx = it * 64 + lane;
      const int pp = idx / 10, part = idx - pp * 10;
      if (p0 + pp < np) *reinterpret_cast<half8*>(dX + (size_t)(p0 + pp) * 80 + 8 * part) = ld8(XT + pp * 88 + 8 * part);
    }
    wsync();
    {                                              // rows 70.4+ of the staging tile overlap nothing that must stay
      // zero, but the staging area [0, 2816) covered X^T rows 0..70: they are rewritten next block; rows 80..95
      // (offset 3200..3839) were never touched and are still zero.
    }
  }

  // ---- reduce the weight-gradient accumulators over the workgroup's waves, write this workgroup's partial.
  // The four waves add their 192 accumulators into the LDS image one after the other with plain read-modify-writes
  // (inside a wave every (register, lane) pair owns its own address): float LDS atomics retire 0.33 lanes per clock on
  // this part -- 49 152 of them were 62 us, half of the kernel at 4096 rays -- and their order was not reproducible.
  __syncthreads();
  float* red = reinterpret_cast<float*>(sm + NFRAG * 512);
  for (int i = tid; i < NPARAM; i += 256) red[i] = 0.0f;
  __syncthreads();
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int row = (reg & 3) + 8 * (reg >> 2) + 4 * hf;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
          for (int nt = 0; nt < 3; ++nt) {
            const int f = 32 * nt + r;
            if (f < 80) red[(32 * mt + row) * 80 + f] += cw1[mt][nt][reg];
          }
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) red[5120 + (32 * mt + row) * 64 + 32 * nt + r] += cw2[mt][nt][reg];
        }
        if (row < 16) {
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) red[9216 + row * 64 + 32 * nt + r] += cw3[nt][reg];
        }
      }
    }
    __syncthreads();
  }
  float* out = partial + (size_t)blockIdx.x * NPARAM;
  for (int i = tid; i < NPARAM; i += 256) out[i] = red[i];
}

}  // namespace

extern "C" int gs_mlp_backward_blocks(int n) {
  if (n <= 0) return 0;
  const int nblk = (n + 31) / 32;
  const int want = 256;                            // one persistent workgroup per CU (140 KB of LDS each)
  return nblk < 4 * want ? (nblk + 3) / 4 : want;
}

extern "C" int gs_mlp_backward(const void* x, const void* wpack, const float* d_rgb, const void* rgb, float loss_scale,
                               void* dx, float* partial, int n, gs_stream_t stream) {
  GS_REQUIRE(x && wpack && d_rgb && dx && partial, "mlp_backward: null pointer");
  GS_REQUIRE(n >= 0 && loss_scale > 0.0f, "mlp_backward: bad arguments");
  if (n == 0) return GS_OK;
  const int nblk = (n + 31) / 32;
  const int grid = gs_mlp_backward_blocks(n);
  const size_t lds = (size_t)(NFRAG * 512 + 4 * WAVE_LDS) * sizeof(_Float16);
  static GsLdsLimit limit;
  if (int rc = limit.raise((const void*)neus_mlp_bwd_kernel, lds, "mlp_backward")) return rc;
  GS_TIMING_PRE();
  neus_mlp_bwd_kernel<<<grid, 256, lds, (hipStream_t)stream>>>((const _Float16*)x, (const _Float16*)wpack, d_rgb,
                                                              (const _Float16*)rgb, loss_scale, (_Float16*)dx, partial,
                                                              n, nblk);
  GS_CHECK_LAUNCH("mlp_backward");
  return GS_OK;
}
