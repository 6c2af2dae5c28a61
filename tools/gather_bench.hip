// Gather-rate ceiling of one MI355X for the hash-grid access pattern ("L2 / Infinity-Cache gather rate" is the
// bound of NeuS render / train; reference src/InstantNeuS.py:35-94 -> tcnn HashGrid, 16 levels x 8 corners x one 4-byte
// (2 x fp16) entry per sample point).  Stand-alone:  hipcc --offload-arch=gfx950 -O3 tools/gather_bench.hip -o gather_bench
// Prints ONE JSON object: for tables of 0.5 / 4 / 25.2 MB (a coarse hashed level, a few levels, the whole table) the
// rate of independent random 4-byte loads ("random4"), of the grid's own corner pattern ("corner8": per point 4 random
// bases, each read as the two ADJACENT entries x, x+1 -- tcnn's hash multiplies x by 1, so the x-neighbours of a cell
// are neighbours in memory), and of one coalesced streaming pass ("stream") as the HBM/L2 reference.  bench.py's
// `roofline_other` entries of neus_point_kernel / neus_point_bwd_kernel are quoted against these numbers
// (profiles/r04_gather_bench.json).  Measurement tool, not part of libgoslam_hip.so.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "hip error %s at line %d\n", hipGetErrorString(e_), __LINE__); \
      return 1;                                                                    \
    }                                                                              \
  } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

// MODE 0: `per_thread` independent random 4-byte loads.  MODE 1: per_thread / 2 random bases, entries (b, b ^ 1) each.
// MODE 2: the lanes of a wave read the SAME random neighbourhood (consecutive samples of one ray fall into the same
// coarse cell): base = f(wave, k) + small per-lane offset -- what the coarse levels of a real batch look like.
template <int MODE>
__global__ __launch_bounds__(256) void gather_kernel(const uint32_t* __restrict__ tab, uint32_t entries,
                                                     int per_thread, uint32_t* __restrict__ out) {
  const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
  uint32_t acc = 0;
  if (MODE == 0) {
#pragma unroll 8
    for (int k = 0; k < per_thread; ++k) acc += tab[mix(tid * 977u + k * 0x9e3779b9u) % entries];
  } else if (MODE == 1) {
#pragma unroll 4
    for (int k = 0; k < per_thread / 2; ++k) {
      const uint32_t b = mix(tid * 977u + k * 0x9e3779b9u) % entries;
      acc += tab[b] + tab[b ^ 1u];
    }
  } else {
    const uint32_t wave = tid >> 6, lane = tid & 63u;
#pragma unroll 4
    for (int k = 0; k < per_thread / 2; ++k) {
      const uint32_t b = (mix(wave * 977u + k * 0x9e3779b9u) + (lane >> 3) * 17u) % entries;
      acc += tab[b] + tab[b ^ 1u];
    }
  }
  out[tid] = acc;
}

__global__ __launch_bounds__(256) void stream_kernel(const uint4* __restrict__ tab, size_t n16, uint32_t* __restrict__ out) {
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
    const uint4 v = tab[i];
    acc += v.x ^ v.y ^ v.z ^ v.w;
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int MODE>
static float time_gather(const uint32_t* tab, uint32_t entries, int blocks, int per_thread, uint32_t* out) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  gather_kernel<MODE><<<blocks, 256>>>(tab, entries, per_thread, out);
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < 5; ++i) gather_kernel<MODE><<<blocks, 256>>>(tab, entries, per_thread, out);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0.f;
  hipEventElapsedTime(&ms, a, b);
  hipEventDestroy(a);
  hipEventDestroy(b);
  return ms / 5;
}

int main() {
  const size_t max_bytes = 64u << 20;
  uint32_t* tab;
  CK(hipMalloc(&tab, max_bytes));
  CK(hipMemset(tab, 1, max_bytes));
  // points per launch: the 4096-ray batch (294,912 points = 1152 workgroups) and the 32768-ray batch (2.36 M points)
  const int per_thread = 128;       // 16 levels x 8 corners
  uint32_t* out;
  CK(hipMalloc(&out, (size_t)9216 * 256 * 4));
  const double mb[] = {0.5, 4.0, 25.2};
  const int blocks_list[] = {1152, 9216};
  printf("{\"device\": \"MI355X (gfx950)\", \"loads_per_point\": %d, \"results\": [\n", per_thread);
  bool first = true;
  for (int bi = 0; bi < 2; ++bi) {
    const int blocks = blocks_list[bi];
    const double loads = (double)blocks * 256 * per_thread;
    for (int s = 0; s < 3; ++s) {
      const uint32_t entries = (uint32_t)(mb[s] * 1e6 / 4);
      const float t0 = time_gather<0>(tab, entries, blocks, per_thread, out);
      const float t1 = time_gather<1>(tab, entries, blocks, per_thread, out);
      const float t2 = time_gather<2>(tab, entries, blocks, per_thread, out);
      printf("%s {\"points\": %d, \"table_MB\": %.1f, \"random4\": {\"ms\": %.4f, \"Ggathers_per_s\": %.1f, \"GBps_4B\": %.1f}, "
             "\"corner8\": {\"ms\": %.4f, \"Ggathers_per_s\": %.1f, \"GBps_4B\": %.1f}, "
             "\"corner8_wave_coherent\": {\"ms\": %.4f, \"Ggathers_per_s\": %.1f, \"GBps_4B\": %.1f}}",
             first ? "" : ",\n", blocks * 256, mb[s], t0, loads / t0 / 1e6, loads * 4 / t0 / 1e6, t1, loads / t1 / 1e6,
             loads * 4 / t1 / 1e6, t2, loads / t2 / 1e6, loads * 4 / t2 / 1e6);
      first = false;
    }
  }
  printf("\n],\n \"stream\": [\n");
  for (int s = 0; s < 4; ++s) {
    const double smb[] = {0.5, 4.0, 25.2, 64.0};
    const size_t n16 = (size_t)(smb[s] * 1e6 / 16) > (max_bytes / 16) ? max_bytes / 16 : (size_t)(smb[s] * 1e6 / 16);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const int reps = 20;
    stream_kernel<<<2048, 256>>>((const uint4*)tab, n16, out);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) stream_kernel<<<2048, 256>>>((const uint4*)tab, n16, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0.f;
    hipEventElapsedTime(&ms, a, b);
    ms /= reps;
    printf("%s {\"table_MB\": %.1f, \"ms\": %.4f, \"GBps\": %.1f}", s ? ",\n" : "", n16 * 16.0 / 1e6, ms, n16 * 16.0 / ms / 1e6);
    hipEventDestroy(a);
    hipEventDestroy(b);
  }
  printf("\n]}\n");
  return 0;
}
