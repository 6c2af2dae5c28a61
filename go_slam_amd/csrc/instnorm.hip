// InstanceNorm2d (affine = False) + ReLU + residual add + ReLU of the frame encoders, NHWC fp16, three launches
// (reference: src/modules/extractor.py:4-57 ResidualBlock.forward and :93-126 BasicEncoder.forward; the feature
// encoder `fnet` is built with norm_fn='instance', the context encoder `cnet` with norm_fn='none').
//
// torch runs `relu(norm(conv(x)))` as batch_norm_collect_statistics + batch_norm_calc_invstd +
// batch_norm_transform_input + clamp, and the block's `relu(skip + y)` as an add and another clamp: 5-7 launches of
// 3-11 us each on tensors of 1.2-4.9 MB, 200 launches per input frame in all.  Here:
//   instnorm_stats_kernel : per (image, 256-pixel chunk): sums of (x - k) and (x - k)^2 per channel, k = the chunk's
//                           first pixel -> (count, mean, M2) of the chunk
//   instnorm_final_kernel : per image: the chunks merged with Chan's formula -> mean and 1 / sqrt(var + eps) per
//                           channel (fp32, biased variance, as torch's native kernels)
//   norm_act_kernel       : x = half(x + bias)              [bias]     (the convolution is called WITHOUT its bias: torch adds
//                                                                      it in a separate kernel anyway; the statistics see x + bias)
//                           t = half((x - mean) * invstd)   [norm]     -- every rounding point is the one of the fp16
//                           t = max(t, 0)                   [relu_in]     tensors the reference materialises
//                           t = half(skip + t)              [skip]
//                           t = max(t, 0)                   [relu_out]
// Memory-bound: one read of x (+ skip) and one write of y; the statistics pass reads x once more (L2-resident: the
// convolution just wrote it).
#include "common.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

constexpr int IN_CHUNK = 256;      // pixels per statistics workgroup

struct Moments { float n, mean, m2; };

__device__ __forceinline__ Moments merge(const Moments a, const Moments b) {   // Chan et al.
  if (b.n == 0.0f) return a;
  if (a.n == 0.0f) return b;
  Moments r;
  r.n = a.n + b.n;
  const float d = b.mean - a.mean;
  const float f = b.n / r.n;
  r.mean = a.mean + d * f;
  r.m2 = a.m2 + b.m2 + d * d * a.n * f;
  return r;
}

// partial: [n][nblk][c] Moments
__global__ __launch_bounds__(256) void instnorm_stats_kernel(const _Float16* __restrict__ x,
                                                             const _Float16* __restrict__ bias, int hw, int c,
                                                             Moments* __restrict__ partial) {
  __shared__ float ssum[256 * 16];
  const int img = blockIdx.y, blk = blockIdx.x, nblk = gridDim.x;
  const int tid = threadIdx.x;
  const int c8n = c >> 3, lanes = 256 / c8n;
  const int cg = tid % c8n, pl = tid / c8n;
  const int p0 = blk * IN_CHUNK, p1 = min(p0 + IN_CHUNK, hw);
  const _Float16* xi = x + (size_t)img * hw * c;
  // shifted sums with ONE shift per channel and workgroup (the chunk's first pixel): the threads' sums then simply
  // add -- no pairwise Chan merges (a division each) inside the workgroup
  float k[8], s1[8], s2[8];
  half8 bv;
  {
    const half8 v0 = *reinterpret_cast<const half8*>(xi + (size_t)p0 * c + cg * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      bv[j] = bias ? bias[cg * 8 + j] : (_Float16)0.0f;
      k[j] = bias ? (float)(_Float16)((float)v0[j] + (float)bv[j]) : (float)v0[j];
      s1[j] = 0.0f;
      s2[j] = 0.0f;
    }
  }
  for (int p = p0 + pl; p < p1; p += 4 * lanes) {             // 4 loads in flight per thread
    half8 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      v[u] = *reinterpret_cast<const half8*>(xi + (size_t)min(p + u * lanes, p1 - 1) * c + cg * 8);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (p + u * lanes >= p1) continue;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float f = (float)v[u][j];
        if (bias) f = (float)(_Float16)(f + (float)bv[j]);       // the fp16 tensor conv + bias
        const float d = f - k[j];
        s1[j] += d;
        s2[j] = fmaf(d, d, s2[j]);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {                                // [pl][channel][2]
    ssum[(pl * c + cg * 8 + j) * 2 + 0] = s1[j];
    ssum[(pl * c + cg * 8 + j) * 2 + 1] = s2[j];
  }
  __syncthreads();
  for (int col = tid; col < 2 * c; col += 256) {               // column = (channel, which sum): add the pixel lanes
    float a = 0.0f;
    for (int l = 0; l < lanes; ++l) a += ssum[l * 2 * c + col];
    ssum[col] = a;                                             // own column of row 0: nobody else reads or writes it
  }
  __syncthreads();
  if (tid < c) {
    const float n = (float)(p1 - p0), a1 = ssum[2 * tid], a2 = ssum[2 * tid + 1];
    // the shift of channel tid: recomputed from the chunk's first pixel
    float kk = (float)xi[(size_t)p0 * c + tid];
    if (bias) kk = (float)(_Float16)(kk + (float)bias[tid]);
    Moments m;
    m.n = n;
    m.mean = kk + a1 / n;
    m.m2 = fmaxf(a2 - a1 * a1 / n, 0.0f);
    partial[((size_t)img * nblk + blk) * c + tid] = m;
  }
}

// One workgroup per image merges the chunks' moments (Chan) and writes mean and 1 / sqrt(var + eps) per channel.  A
// kernel of its own rather than "the last workgroup to finish": that needs a device-scope fence in every workgroup, and on
// this part (8 XCDs, one L2 each) such a fence writes the L2's dirty lines back -- the convolution output the
// statistics were just read from -- which made the statistics pass 17 us instead of 3.
__global__ __launch_bounds__(256) void instnorm_final_kernel(const Moments* __restrict__ partial, int nblk, int c, float eps,
                                                             float* __restrict__ final_) {
  __shared__ Moments sm[256];
  const int img = blockIdx.x, tid = threadIdx.x;
  for (int c0 = 0; c0 < c; c0 += 256) {                       // (c <= 256: one pass)
    const int cc = min(c - c0, 256);
    const int per = 256 / cc;                                 // threads per channel
    const int ch = tid % cc, part = tid / cc;
    Moments acc = {0.0f, 0.0f, 0.0f};
    if (part < per) {
      // 8 independent loads per round trip (a rolled loop pays one L2 latency per chunk)
      for (int b = part; b < nblk; b += 8 * per) {
        Moments m[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int bb = b + u * per;
          m[u] = partial[((size_t)img * nblk + min(bb, nblk - 1)) * c + c0 + ch];
          if (bb >= nblk) m[u].n = 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = merge(acc, m[u]);
      }
    }
    sm[tid] = acc;
    __syncthreads();
    if (tid < cc) {
      Moments r = sm[tid];
      for (int q = 1; q < per; ++q) r = merge(r, sm[q * cc + tid]);
      const float var = r.n > 0.0f ? r.m2 / r.n : 0.0f;
      final_[((size_t)img * c + c0 + tid) * 2 + 0] = r.mean;
      final_[((size_t)img * c + c0 + tid) * 2 + 1] = 1.0f / sqrtf(var + eps);
    }
    __syncthreads();
  }
}

// The same for partial SUMS [n][nblk][c][2] of d = v - shift and d^2 (gs_enc_conv's epilogue statistics, shift = the
// convolution's bias): plain additions, eight loads in flight per thread, then mean = shift + S1 / N and the biased
// variance (S2 - S1^2 / N) / N.
constexpr int FS_NT = 1024;      // one workgroup per image merges <= 600 partial sums per channel: with 256 threads that was
                                 // 9-10 dependent round trips of 64 chunks (6.5 us, 15 times per input frame); 1024 threads
                                 // take 256 chunks per round trip.  The order of the additions is fixed (thread `part` adds
                                 // its chunks in ascending order, thread 0..c-1 then adds the parts in ascending order).
__global__ __launch_bounds__(FS_NT) void instnorm_final_sums_kernel(const float* __restrict__ sums, int nblk, int c, int hw,
                                                                    const _Float16* __restrict__ shift, float eps,
                                                                    float* __restrict__ final_) {
  __shared__ float sm[FS_NT][2];
  const int img = blockIdx.x, tid = threadIdx.x;
  const int per = FS_NT / c;                                  // threads per channel (c in {32, 64, 128, 256})
  const int ch = tid % c, part = tid / c;
  float a1 = 0.0f, a2 = 0.0f;
  const float2* src = reinterpret_cast<const float2*>(sums) + (size_t)img * nblk * c + ch;
  for (int b = part; b < nblk; b += 8 * per) {
    float2 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int bb = b + u * per;
      v[u] = src[(size_t)min(bb, nblk - 1) * c];
      if (bb >= nblk) v[u] = make_float2(0.0f, 0.0f);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) { a1 += v[u].x; a2 += v[u].y; }
  }
  sm[tid][0] = a1;
  sm[tid][1] = a2;
  __syncthreads();
  if (tid < c) {
    for (int q = 1; q < per; ++q) { a1 += sm[q * c + tid][0]; a2 += sm[q * c + tid][1]; }
    // E[d^2] - E[d]^2 in fp64: on a near-constant map the two terms agree to ~7 digits and their fp32 difference would be
    // of the order of eps itself (the sums are of d = v - bias, already centred on the convolution's own mean)
    const double n = (double)hw;
    const double mean_d = (double)a1 / n;
    const double var = fmax((double)a2 / n - mean_d * mean_d, 0.0);
    final_[((size_t)img * c + tid) * 2 + 0] = (shift ? (float)shift[tid] : 0.0f) + (float)mean_d;
    final_[((size_t)img * c + tid) * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
}
__global__ __launch_bounds__(256) void norm_act_kernel(const _Float16* __restrict__ x, const _Float16* __restrict__ skip,
                                                       _Float16* __restrict__ y, const _Float16* __restrict__ bias,
                                                       const float* __restrict__ final_, int hw, int c, int relu_in,
                                                       int relu_out, size_t total) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const int c8n = c >> 3;
  const int cg = (int)(t % c8n);
  const int img = (int)(t / ((size_t)hw * c8n));
  // every operand is requested before the first is used, behind no branch (an absent operand reads a valid address and is
  // ignored): x, then the bias, then the statistics, then the skip tensor were up to four dependent round trips of a
  // 4 us kernel that runs fifteen times per input frame
  typedef float float4v __attribute__((ext_vector_type(4)));
  half8 v = reinterpret_cast<const half8*>(x)[t];
  const half8 sk = reinterpret_cast<const half8*>(skip ? skip : x)[t];
  const half8 bv = *reinterpret_cast<const half8*>(bias ? bias + cg * 8 : x);
  float4v f0 = {0.f, 1.f, 0.f, 1.f}, f1 = f0, f2 = f0, f3 = f0;
  if (final_) {                           // (a kernel argument: the branch is uniform and the four loads stay together)
    const float4v* fp = reinterpret_cast<const float4v*>(final_ + ((size_t)img * c + cg * 8) * 2);
    f0 = fp[0]; f1 = fp[1]; f2 = fp[2]; f3 = fp[3];
  }
  if (bias) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (_Float16)((float)v[j] + (float)bv[j]);
  }
  if (final_) {
    const float f[16] = {f0[0], f0[1], f0[2], f0[3], f1[0], f1[1], f1[2], f1[3], f2[0], f2[1], f2[2], f2[3], f3[0], f3[1], f3[2], f3[3]};
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (_Float16)(((float)v[j] - f[2 * j]) * f[2 * j + 1]);
  }
  if (relu_in) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = v[j] > (_Float16)0.0f ? v[j] : (_Float16)0.0f;
  }
  if (skip) {
    const half8 s = sk;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (_Float16)((float)s[j] + (float)v[j]);
  }
  if (relu_out) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = v[j] > (_Float16)0.0f ? v[j] : (_Float16)0.0f;
  }
  reinterpret_cast<half8*>(y)[t] = v;
}

}  // namespace

extern "C" size_t gs_norm_act_workspace_bytes_chunks(int n, int chunks, int channels) {
  if (n <= 0 || chunks <= 0 || channels <= 0) return 0;
  const size_t nblk = (size_t)chunks;
  return gs_align((size_t)n * nblk * channels * sizeof(Moments)) + gs_align((size_t)n * channels * 2 * sizeof(float)) + 256;
}

extern "C" size_t gs_norm_act_workspace_bytes(int n, int hw, int channels) {
  if (n <= 0 || hw <= 0 || channels <= 0) return 0;
  const size_t nblk = (size_t)gs_cdiv(hw, IN_CHUNK);
  return gs_align((size_t)n * nblk * channels * sizeof(Moments)) + gs_align((size_t)n * channels * 2 * sizeof(float)) + 256;
}

extern "C" int gs_norm_act(const void* x, const void* bias, const void* skip, void* y, int n, int hw, int channels,
                           int instance_norm,
                           int relu_in, int relu_out, float eps, void* workspace, size_t workspace_bytes,
                           int stat_chunks, gs_stream_t stream) {
  GS_REQUIRE(x && y, "norm_act: null pointer");
  GS_REQUIRE(n >= 0 && hw > 0, "norm_act: bad shape");
  GS_REQUIRE(channels == 32 || channels == 64 || channels == 128 || channels == 256 || !instance_norm,
             "norm_act: instance norm supports 32, 64, 128 or 256 channels (got %d)", channels);
  GS_REQUIRE(channels > 0 && channels % 8 == 0, "norm_act: channels must be a multiple of 8");
  GS_REQUIRE((((size_t)x | (size_t)y | (size_t)skip | (size_t)bias) & 15) == 0, "norm_act: tensors must be 16-byte aligned");
  if (n == 0) return GS_OK;
  hipStream_t st = (hipStream_t)stream;
  const float* fin = nullptr;
  if (instance_norm) {
    GS_REQUIRE(channels <= 256, "norm_act: at most 256 channels");
    // stat_chunks > 0: the producing convolution (gs_enc_conv) already left that many chunk moments per image at the
    // start of the workspace -- only the merge and the apply pass run
    const size_t need = stat_chunks > 0 ? gs_norm_act_workspace_bytes_chunks(n, stat_chunks, channels)
                                        : gs_norm_act_workspace_bytes(n, hw, channels);
    if (!workspace || workspace_bytes < need) {
      gs_set_error("norm_act: workspace too small (%zu < %zu)", workspace_bytes, need);
      return GS_ERR_WORKSPACE;
    }
    const int nblk = stat_chunks > 0 ? stat_chunks : gs_cdiv(hw, IN_CHUNK);
    char* base = (char*)gs_align((size_t)workspace);
    Moments* partial = (Moments*)base;
    float* final_ = (float*)((char*)partial + gs_align((size_t)n * nblk * channels * sizeof(Moments)));
    GS_REQUIRE(n <= 65535, "norm_act: too many images");
    if (stat_chunks <= 0) {
      instnorm_stats_kernel<<<dim3(nblk, n), 256, 0, st>>>((const _Float16*)x, (const _Float16*)bias, hw, channels, partial);
      GS_CHECK_LAUNCH("instnorm_stats");
    }
    if (stat_chunks > 0)
      instnorm_final_sums_kernel<<<n, FS_NT, 0, st>>>((const float*)partial, nblk, channels, hw, (const _Float16*)bias, eps, final_);
    else
      instnorm_final_kernel<<<n, 256, 0, st>>>(partial, nblk, channels, eps, final_);
    GS_CHECK_LAUNCH("instnorm_final");
    fin = final_;
  }
  const size_t total = (size_t)n * hw * (channels / 8);
  norm_act_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>((const _Float16*)x, (const _Float16*)skip,
                                                                   (_Float16*)y, (const _Float16*)bias, fin, hw, channels,
                                                                   relu_in, relu_out, total);
  GS_CHECK_LAUNCH("norm_act");
  return GS_OK;
}
