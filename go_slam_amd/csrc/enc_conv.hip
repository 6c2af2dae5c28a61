// The frame encoders' convolutions (reference src/modules/extractor.py:61-126, run once per INPUT frame by
// MotionFilter.track, src/motion_filter.py:41-83): the 7x7 stride-2 stem 3 -> 32, 3x3 convolutions at 32 / 64 / 128
// channels with stride 1 or 2, the strided 1x1 skips and the 1x1 projection -- SURVEY 8 f2.  Until round 4 these were
// MIOpen's NHWC fp16 solvers; they are small (16.5 GFLOP and <= 5 MB of activations per frame in all), so what a layer
// costs is launch + latency, not MFMA or HBM throughput, and the design follows from that:
//
//   * implicit GEMM on v_mfma_f32_32x32x16_f16 WITHOUT any staging: a wave owns 32 consecutive output pixels of one
//     output row x ONE 32-channel tile (16 accumulator registers: many light waves per CU hide the load latency, and the
//     60 x 80 layers still spread over 700+ waves; the channel tiles of a pixel group sit in one workgroup and share the
//     pixel operand through L1).  The pixel operand of a k-step is 8 consecutive input
//     channels of one tap of the lane's pixel = ONE 16-byte load from the NHWC map (zero for taps outside the image);
//     every input value is read 9 times, but out of L1 / L2 -- the whole layer-1 activation is 4.9 MB.  The weights
//     are the A operand, pre-packed in fragment order ([tap][k-step][32-channel tile][lane][8]), so a fragment is one
//     coalesced 1 KB load that every wave of the launch shares through L1 / L2;
//   * the stem reads a 4-channel (RGB0) fp16 image: K = 7 kernel rows x (8 taps x 4 channels), the 8th tap zero; one
//     k-group of a fragment = 2 neighbouring taps x 4 channels = two 8-byte loads (a pair can straddle the image border);
//   * fp32 accumulators -> one fp16 rounding (+ the fp16 bias add of the 1x1 projection, the rounding points of
//     conv -> bias-add as two fp16 tensors) -> wave-private LDS transpose -> 16-byte stores of contiguous channel runs.
//
// The instance-norm / ReLU / residual tail of every convolution stays in gs_norm_act (instnorm.hip).
#include "common.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float float16v __attribute__((ext_vector_type(16)));

struct EncArgs {
  float* stats;                       // [n][workgroups per image][cout][2] sums of (v - bias) and (v - bias)^2, or nullptr
  const _Float16* stat_bias;          // [cout] fp16: v = half(half(conv) + bias), the tensor torch normalises
  int cout;
  const _Float16* x; int xs;          // input NHWC, `xs` halves per pixel
  const half8* wpack;                 // [taps][k-steps][m-tiles][64] fragments
  const _Float16* bias;               // [cout] fp16 or nullptr
  _Float16* y; int ys;                // output NHWC, `ys` halves per pixel
  int n, h, w, ho, wo;
};

// one fp32 accumulator tile (32 channels x 32 pixels) -> fp16 NHWC rows through a wave-private LDS tile [32 px][40]
// this lane's 16 channels of a per-channel fp16 vector (channels c0 + 8 g + 4 kh + e): four 8-byte loads, REQUESTED
// at the head of the kernel.  (Read where they are used -- one 2-byte load per accumulator register, each converted at
// once -- the epilogue of every wave was 32 dependent cache round trips: a third of a 12-18 us launch.)
struct EncBias { half4 q[4]; };
__device__ __forceinline__ EncBias load_bias16(const _Float16* p, const void* any, int c0, int lane) {
  // (a null vector is read from `any` -- readable, >= 64 bytes, values unused -- so that the loads sit behind no branch
  // and stay in flight across the convolution's loop)
  const _Float16* q = p ? p + c0 : reinterpret_cast<const _Float16*>(any);
  EncBias b;
#pragma unroll
  for (int g = 0; g < 4; ++g) b.q[g] = *reinterpret_cast<const half4*>(q + 8 * g + 4 * (lane >> 5));
  return b;
}

__device__ __forceinline__ void store_tile(const float16v& acc, _Float16* tile, const EncArgs& A, int img, int oy, int ox0,
                                           int c0, int lane, float* red, const EncBias& bias, const EncBias& sbias) {
  constexpr int TS = 40;
  const int r = lane & 31, kh = lane >> 5;
  if (A.stats) {
    // InstanceNorm statistics in the convolution's epilogue (one launch less per normalisation, and the statistics pass
    // over the tensor is gone): per channel the sums of d and d^2 over this wave's pixels, d = v - bias with v =
    // half(half(conv) + bias) the fp16 value torch normalises -- shifting by the bias keeps d centred on the
    // convolution's own (small) mean, so plain fp32 sums stay well conditioned and the workgroups' partial sums merge
    // by ADDITION (instnorm_final_sums_kernel; Chan merges cost a division per chunk and channel).
    const int nvalid = min(32, A.wo - ox0);
    float sv[32];                                  // [0..15] = d, [16..31] = d^2 of this lane's pixel, per accumulator register
#pragma unroll
    for (int e16 = 0; e16 < 16; ++e16) {
      const float v = (float)(_Float16)acc[e16];
      float d = v;
      if (A.stat_bias) {
        const float bsh = (float)sbias.q[e16 >> 2][e16 & 3];
        d = (float)(_Float16)(v + bsh) - bsh;
      }
      if (r >= nvalid) d = 0.0f;
      sv[e16] = d;
      sv[16 + e16] = d * d;
    }
    // sums over the 32 pixel lanes of each half-wave as a reduce-scatter (31 lane exchanges instead of 32 butterflies of
    // 5): afterwards lane L holds the total of value index bitrev5(L & 31)
    gs_rs_step<32, 16>(sv, lane);
    gs_rs_step<16, 8>(sv, lane);
    gs_rs_step<8, 4>(sv, lane);
    gs_rs_step<4, 2>(sv, lane);
    gs_rs_step<2, 1>(sv, lane);
    {
      const int idx = (int)(__brev((unsigned)(lane & 31)) >> 27);
      const int e16 = idx & 15;
      const int ch = c0 + 8 * (e16 >> 2) + 4 * kh + (e16 & 3);
      red[2 * ch + (idx >> 4)] = sv[0];       // this wave's own slab: 64 lanes, 64 distinct (channel, moment) slots
    }
  }
#pragma unroll
  for (int g = 0; g < 4; ++g) {                   // C layout: channel = 8 g + 4 kh + e, pixel = r
    half4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v = (float)(_Float16)acc[4 * g + e];
      if (A.bias) v = (float)(_Float16)(v + (float)bias.q[g][e]);
      o[e] = (_Float16)v;
    }
    *reinterpret_cast<half4*>(tile + r * TS + 8 * g + 4 * kh) = o;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  _Float16* yrow = A.y + ((size_t)img * A.ho + oy) * A.wo * A.ys + c0;
#pragma unroll
  for (int it = 0; it < 2; ++it) {                // 32 px x 4 pieces of 16 B
    const int idx = lane + 64 * it;
    const int px = idx >> 2, pc = idx & 3;
    if (ox0 + px < A.wo)
      *reinterpret_cast<half8*>(yrow + (size_t)(ox0 + px) * A.ys + pc * 8) =
          *reinterpret_cast<const half8*>(tile + px * TS + pc * 8);
  }
}

// KS x KS convolution (KS in {1, 3}), padding KS / 2, stride STRIDE, CIN input channels; `mtiles` = c_out / 32
// workgroup-level tail of the statistics: zero / publish the per-channel LDS accumulators
// (one slab of 2 * cout sums per wave, written with plain stores and added in wave order: the statistics -- and with
// them the encoder features -- are bitwise reproducible; floating-point LDS atomics from four waves were not)
__device__ __forceinline__ void red_init(float* red, int n2, int slab) {
  for (int w = 0; w < 4; ++w)
    for (int i = threadIdx.x; i < n2; i += 256) red[w * slab + i] = 0.0f;
  __syncthreads();
}
__device__ __forceinline__ void red_publish(const float* red, const EncArgs& A, int img, int slab) {
  __syncthreads();
  float* dst = A.stats + ((size_t)img * gridDim.x + blockIdx.x) * A.cout * 2;
  for (int i = threadIdx.x; i < 2 * A.cout; i += 256)
    dst[i] = ((red[i] + red[slab + i]) + red[2 * slab + i]) + red[3 * slab + i];
}

template <int KS, int CIN, int STRIDE>
__global__ __launch_bounds__(256) void enc_conv_kernel(EncArgs A, int mtiles) {
  constexpr int KSTEPS = CIN / 16, PAD = KS / 2;
  __shared__ __attribute__((aligned(16))) _Float16 tiles[4][32 * 40];
  constexpr int SLAB = 2 * 256;
  __shared__ float red[4 * SLAB];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int groups_x = (A.wo + 31) / 32;
  const int img = blockIdx.y;                     // a workgroup never straddles two images (its statistics are per image)
  const int wave_id = blockIdx.x * 4 + wv;
  const bool active = wave_id < A.ho * groups_x * mtiles;
  if (A.stats) red_init(red, 2 * A.cout, SLAB);
  const int mt = wave_id % mtiles;
  const int pg = wave_id / mtiles;
  const int gx = pg % groups_x;
  const int oy = active ? pg / groups_x : 0;
  const int r = lane & 31, kh = lane >> 5;
  const int ox0 = gx * 32, ox = ox0 + r;
  float16v acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
  const EncBias bias = load_bias16(A.bias, A.wpack, 32 * mt, lane), sbias = load_bias16(A.stat_bias, A.wpack, 32 * mt, lane);
  const _Float16* ximg = A.x + (size_t)img * A.h * A.w * A.xs;
  const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int dy = 0; dy < KS; ++dy) {
    const int iy = oy * STRIDE + dy - PAD;
    const bool rowok = active && iy >= 0 && iy < A.h;
#pragma unroll
    for (int dx = 0; dx < KS; ++dx) {
      const int ix = ox * STRIDE + dx - PAD;
      const bool ok = rowok && ix >= 0 && ix < A.w && ox < A.wo;
      const _Float16* px = ximg + ((size_t)(rowok ? iy : 0) * A.w + (ok ? ix : 0)) * A.xs + 8 * kh;
      const half8* wt = A.wpack + ((size_t)(dy * KS + dx) * KSTEPS * mtiles + mt) * 64 + lane;
      half8 b[KSTEPS], a[KSTEPS];
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
        const half8 t = *reinterpret_cast<const half8*>(px + 16 * ks);      // (px is clamped into the image: no branch
        b[ks] = ok ? t : zero8;                                              //  around the load, the taps' loads overlap)
        a[ks] = wt[(size_t)ks * mtiles * 64];
      }
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks], b[ks], acc, 0, 0, 0);
    }
  }
  if (active) store_tile(acc, tiles[wv], A, img, oy, ox0, 32 * mt, lane, red + wv * SLAB, bias, sbias);
  if (A.stats) red_publish(red, A, img, SLAB);
}

// the stem: 7 x 7, stride 2, padding 3, 4 (RGB0) -> 32 channels.  K = 7 kernel rows x 32 (8 taps x 4 channels, tap 7 = 0)
__global__ __launch_bounds__(256) void enc_stem_kernel(EncArgs A) {
  __shared__ __attribute__((aligned(16))) _Float16 tiles[4][32 * 40];
  constexpr int SLAB = 2 * 32;
  __shared__ float red[4 * SLAB];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int groups_x = (A.wo + 31) / 32;
  const int img = blockIdx.y;
  const int wave_id = blockIdx.x * 4 + wv;
  const bool active = wave_id < A.ho * groups_x;
  if (A.stats) red_init(red, 2 * 32, SLAB);
  const int gx = wave_id % groups_x;
  const int oy = active ? wave_id / groups_x : 0;
  const int r = lane & 31, kh = lane >> 5;
  const int ox0 = gx * 32, ox = ox0 + r;
  float16v acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
  const EncBias bias = load_bias16(A.bias, A.wpack, 0, lane), sbias = load_bias16(A.stat_bias, A.wpack, 0, lane);
  const half4* ximg = reinterpret_cast<const half4*>(A.x) + (size_t)img * A.h * A.w;
  const half4 zero4 = {0, 0, 0, 0};
  // all 28 pixel loads and 14 weight fragments of the wave are requested before the first product, from addresses
  // clamped into the image (taps outside it are zeroed by a select afterwards): as `inside ? load : 0` every k-step's
  // loads were waited for before the next step's were issued -- 14 dependent round trips in a 12 us kernel
  half4 px0[14], px1[14];
  half8 wf[14];
#pragma unroll
  for (int dy = 0; dy < 7; ++dy) {
    const int iy = 2 * oy + dy - 3;
    const half4* xrow = ximg + (size_t)min(max(iy, 0), A.h - 1) * A.w;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int t0 = 4 * ks + 2 * kh;                // this half-wave's two taps of the k-step
      const int ix0 = 2 * ox + t0 - 3, ix1 = ix0 + 1;
      px0[dy * 2 + ks] = xrow[min(max(ix0, 0), A.w - 1)];
      px1[dy * 2 + ks] = xrow[min(max(ix1, 0), A.w - 1)];
      wf[dy * 2 + ks] = A.wpack[(dy * 2 + ks) * 64 + lane];
    }
  }
#pragma unroll
  for (int dy = 0; dy < 7; ++dy) {
    const int iy = 2 * oy + dy - 3;
    const bool rowok = active && iy >= 0 && iy < A.h && ox < A.wo;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int t0 = 4 * ks + 2 * kh;
      const int ix0 = 2 * ox + t0 - 3, ix1 = ix0 + 1;
      const half4 p0 = (rowok && ix0 >= 0 && ix0 < A.w) ? px0[dy * 2 + ks] : zero4;
      const half4 p1 = (rowok && ix1 >= 0 && ix1 < A.w) ? px1[dy * 2 + ks] : zero4;     // (tap 7: weights are zero)
      const half8 b = {p0[0], p0[1], p0[2], p0[3], p1[0], p1[1], p1[2], p1[3]};
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[dy * 2 + ks], b, acc, 0, 0, 0);
    }
  }
  if (active) store_tile(acc, tiles[wv], A, img, oy, ox0, 0, lane, red + wv * SLAB, bias, sbias);
  if (A.stats) red_publish(red, A, img, SLAB);
}

template <int KS, int CIN, int STRIDE>
int launch_enc(const EncArgs& A, int c_out, hipStream_t st) {
  const int mtiles = c_out / 32;
  const long long waves = (long long)A.ho * ((A.wo + 31) / 32) * mtiles;     // per image
  GS_REQUIRE(waves < (1ll << 31) && A.n <= 65535, "enc_conv: too many pixel groups / images");
  enc_conv_kernel<KS, CIN, STRIDE><<<dim3((unsigned)((waves + 3) / 4), (unsigned)A.n), 256, 0, st>>>(A, mtiles);
  GS_CHECK_LAUNCH("enc_conv");
  return GS_OK;
}

}  // namespace

extern "C" size_t gs_enc_conv_wpack_elems(int ksize, int c_in, int c_out) {
  if (ksize == 7) return (size_t)14 * 64 * 8;
  return (size_t)ksize * ksize * c_in * c_out;
}

// workgroups per image = partial-sum slabs per image the epilogue statistics produce
extern "C" int gs_enc_conv_stat_chunks(int h_out, int w_out, int c_out) {
  return (h_out * ((w_out + 31) / 32) * (c_out / 32) + 3) / 4;
}

extern "C" int gs_enc_conv(const void* x, int x_stride, int c_in, const void* wpack, const void* bias, void* y, int y_stride,
                           int c_out, int ksize, int stride, int n, int h, int w, const void* stat_bias, void* stats_ws,
                           gs_stream_t stream) {
  GS_REQUIRE(x && wpack && y, "enc_conv: null pointer");
  GS_REQUIRE(n >= 0 && h > 0 && w > 0, "enc_conv: bad shape");
  GS_REQUIRE(stride == 1 || stride == 2, "enc_conv: stride must be 1 or 2");
  GS_REQUIRE(x_stride >= c_in && (x_stride % 8 == 0 || ksize == 7) && y_stride >= c_out && y_stride % 8 == 0,
             "enc_conv: strides must cover the channels and be multiples of 8");
  GS_REQUIRE((((size_t)x | (size_t)y | (size_t)wpack) & 15) == 0, "enc_conv: buffers must be 16-byte aligned");
  GS_REQUIRE((((size_t)bias | (size_t)stat_bias) & 7) == 0, "enc_conv: bias vectors must be 8-byte aligned");
  GS_REQUIRE(!stats_ws || c_out <= 256, "enc_conv: epilogue statistics support at most 256 output channels");
  if (n == 0) return GS_OK;
  EncArgs A;
  A.x = (const _Float16*)x; A.xs = x_stride; A.wpack = (const half8*)wpack; A.bias = (const _Float16*)bias;
  A.y = (_Float16*)y; A.ys = y_stride; A.n = n; A.h = h; A.w = w;
  A.stats = stats_ws ? (float*)gs_align((size_t)stats_ws) : nullptr;      // (where gs_norm_act expects the partial sums)
  A.stat_bias = (const _Float16*)stat_bias; A.cout = c_out;
  const int pad = ksize / 2;
  A.ho = (h + 2 * pad - ksize) / stride + 1;
  A.wo = (w + 2 * pad - ksize) / stride + 1;
  hipStream_t st = (hipStream_t)stream;
  if (ksize == 7) {
    GS_REQUIRE(c_in == 4 && c_out == 32 && stride == 2 && x_stride == 4,
               "enc_conv: the 7x7 stem is built for a dense 4-channel (RGB0) input, 32 outputs, stride 2");
    const long long waves = (long long)A.ho * ((A.wo + 31) / 32);
    GS_REQUIRE(waves < (1ll << 31) && A.n <= 65535, "enc_conv: too many pixel groups / images");
    enc_stem_kernel<<<dim3((unsigned)((waves + 3) / 4), (unsigned)A.n), 256, 0, st>>>(A);
    GS_CHECK_LAUNCH("enc_stem");
    return GS_OK;
  }
#define ENC_CASE(KS_, CI_, CO_, S_) \
  if (ksize == KS_ && c_in == CI_ && c_out == CO_ && stride == S_) return launch_enc<KS_, CI_, S_>(A, c_out, st);
  ENC_CASE(3, 32, 32, 1)
  ENC_CASE(3, 32, 64, 2)
  ENC_CASE(3, 64, 64, 1)
  ENC_CASE(3, 64, 128, 2)
  ENC_CASE(3, 128, 128, 1)
  ENC_CASE(1, 32, 64, 2)
  ENC_CASE(1, 64, 128, 2)
  ENC_CASE(1, 128, 128, 1)
  ENC_CASE(1, 128, 256, 1)
#undef ENC_CASE
  gs_set_error("enc_conv: no kernel for %dx%d, %d -> %d channels, stride %d (the encoder's layer shapes only)", ksize, ksize,
               c_in, c_out, stride);
  return GS_ERR_UNSUPPORTED;
}
