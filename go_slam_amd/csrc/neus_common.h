// Hash-grid helpers shared by the NeuS kernels (neus.hip, neus_bwd.hip, grid_autograd.hip): tiny-cuda-nn HashGrid
// indexing and the wave-merged table-gradient scatter.
#pragma once
#include "common.h"
#include "../../include/goslam_neus.h"

namespace {

// sin / cos of a Fourier-feature argument (InstantNeuS.py:64,72,177,196: `torch.sin(x @ _B)`, _B = 25 * randn -- arguments
// of a few hundred radians).  The forward rounds the value to fp16 at once (it is a colour-MLP input) and the backward's
// d arg row is fp16 as well, so libm's sinf / cosf -- ~40 VALU instructions each on its small-argument path, 33 per point:
// a quarter of the forward point stage's arithmetic -- buy nothing: two-term Cody-Waite reduction to [-pi, pi]
// (k * 6.28125 is exact for |k| < 2^16, the remainder of 2 pi follows in a second fma: reduction error < 3e-7 rad for
// |x| < 1e4) and the hardware's v_sin_f32 / v_cos_f32 (argument in revolutions): 6 instructions, |error| < 2e-6 against
// libm on [-1e3, 1e3] (tests/test_neus_gpu.py::test_embedding_sine_*), i.e. below 1 / 100 of an fp16 ulp at 1.
__device__ __forceinline__ float emb_reduce(float x) {
  const float k = __builtin_rintf(x * 0.15915494309189535f);
  float r = fmaf(-k, 6.28125f, x);
  r = fmaf(-k, 1.9353071795864769e-3f, r);
  return r * 0.15915494309189535f;                  // revolutions, |.| <= 1/2 (+ rounding)
}
__device__ __forceinline__ float emb_sin(float x) { return __builtin_amdgcn_sinf(emb_reduce(x)); }
__device__ __forceinline__ float emb_cos(float x) { return __builtin_amdgcn_cosf(emb_reduce(x)); }

__device__ __forceinline__ uint32_t grid_index(const gs_grid_meta& m, int l, uint32_t cx, uint32_t cy, uint32_t cz) {
  const uint32_t size = m.size[l];
  uint32_t idx;
  if (m.hashed[l]) {
    idx = (cx * 1u) ^ (cy * 2654435761u) ^ (cz * 805459861u);
  } else {
    const uint32_t res = m.resolution[l];
    idx = cx + cy * res + cz * res * res;
  }
  // tcnn: `index % hashmap_size`.  A hashed level holds 2^19 entries (a mask), and a dense level's index is below its
  // size except at the far corner of the box -- the generic 32-bit modulo (~30 VALU instructions, 8 per level and
  // point, a third of a level's arithmetic) only runs when it has to.  Same result in every case.
  if ((size & (size - 1u)) == 0u) return idx & (size - 1u);
  return idx < size ? idx : idx % size;
}

// The 8 corner indices of one cell, corner c = (c & 1, (c >> 1) & 1, (c >> 2) & 1) -- the same values as eight
// grid_index calls (uint32 arithmetic is a ring: (cy + 1) * P == cy * P + P), with the level's kind decided ONCE instead of
// inside every corner: a hashed level costs 2 multiplications + 2 additions + 12 xors + 8 masks instead of 16
// multiplications (v_mul_lo_u32 is a quarter-rate instruction) + 16 xors + 8 masks + 12 corner additions and five uniform
// branches per corner; a dense level is the base index plus the uniform strides 1 / res / res^2.
__device__ __forceinline__ void grid_corners(const gs_grid_meta& m, int l, const uint32_t (&g)[3], uint32_t (&cidx)[8]) {
  const uint32_t size = m.size[l];
  if (m.hashed[l]) {
    const uint32_t y0 = g[1] * 2654435761u, y1 = y0 + 2654435761u;
    const uint32_t z0 = g[2] * 805459861u, z1 = z0 + 805459861u;
    const uint32_t x0 = g[0], x1 = g[0] + 1u;
    const uint32_t yz[4] = {y0 ^ z0, y1 ^ z0, y0 ^ z1, y1 ^ z1};
#pragma unroll
    for (int c = 0; c < 8; ++c) cidx[c] = ((c & 1) ? x1 : x0) ^ yz[c >> 1];
  } else {
    const uint32_t res = m.resolution[l], res2 = res * res;
    const uint32_t base = g[0] + g[1] * res + g[2] * res2;
#pragma unroll
    for (int c = 0; c < 8; ++c) cidx[c] = base + (uint32_t)(c & 1) + (((c >> 1) & 1) ? res : 0u) + (((c >> 2) & 1) ? res2 : 0u);
  }
  // tcnn: `index % hashmap_size` (see grid_index)
  if ((size & (size - 1u)) == 0u) {
#pragma unroll
    for (int c = 0; c < 8; ++c) cidx[c] &= size - 1u;
  } else {
#pragma unroll
    for (int c = 0; c < 8; ++c) cidx[c] = cidx[c] < size ? cidx[c] : cidx[c] % size;
  }
}

// Scatter one level's 8 corners x 2 features.  Lanes are consecutive samples of a ray, so
// neighbours often sit in the SAME cell (always at the coarse levels): runs of equal cells are
// pre-reduced inside the wave with a segmented scan and only the run's last lane issues atomics --
// at the coarse levels this removes ~10x of the (memory-side, heavily contended) atomic traffic.
// `tab16` != nullptr: tiny-cuda-nn's own accumulation mode -- the table gradient is fp16, both features
// of an entry go out as ONE packed atomic (global_atomic_pk_add_f16), pre-multiplied by the loss scale
// (tcnn: 128) so that small contributions stay above fp16's subnormal range; half the atomic count.
typedef _Float16 half2a __attribute__((ext_vector_type(2)));
// Runs of consecutive lanes in the same cell: inclusive segmented sum of the 8 x 2 corner contributions; returns
// whether this lane is the LAST of its run (the one that holds the run's total and emits it).
// value of the lane below (lane 0: its own) -- DPP wave_shr:1, one full-rate VALU move instead of the address arithmetic +
// ds_bpermute + LDS-return wait of __shfl_up(v, 1): the neighbour test and the scan's first (and at the fine levels only)
// step use 20 of them per level
__device__ __forceinline__ int wave_shr1(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ float wave_shr1(float v) {
  return __builtin_bit_cast(float, wave_shr1(__builtin_bit_cast(int, v)));
}

__device__ __forceinline__ bool lvl_prereduce(float (&gacc)[8][2], const uint32_t (&gi)[3], bool on, int lane) {
  const uint32_t p0 = (uint32_t)wave_shr1((int)gi[0]), p1 = (uint32_t)wave_shr1((int)gi[1]), p2 = (uint32_t)wave_shr1((int)gi[2]);
  const int on_prev = wave_shr1((int)on);
  const bool same = (lane > 0) && on && on_prev && p0 == gi[0] && p1 == gi[1] && p2 == gi[2];
  const unsigned long long same_mask = __ballot(same);
  bool tail = true;
  if (same_mask != 0ull) {                            // wave-uniform
    const unsigned long long starts = ~same_mask;     // bit set where a run begins
    const unsigned long long below = starts & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull));
    const int run_start = 63 - __builtin_clzll(below);
    // (the scan stops as soon as no lane reaches back `off` lanes inside its run -- i.e. after ceil(log2(longest run)) of
    // the 6 doubling steps: skipping a step in which no lane takes anything changes nothing, and at the fine levels, where
    // runs are 2-3 lanes long, it saves 64-80 of the 96 cross-lane moves the backward's ISA count shows per level)
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const bool take = (lane - off) >= run_start;
      if (__ballot(take) == 0ull) break;            // wave-uniform
      // all sixteen lane exchanges first, then ONE predicated block of adds (round 6: exchange + select + add per value
      // was 243 v_cndmask per level in the ISA -- half of this function's vector instructions)
      float u[8][2];
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        u[c][0] = off == 1 ? wave_shr1(gacc[c][0]) : __shfl_up(gacc[c][0], off, 64);
        u[c][1] = off == 1 ? wave_shr1(gacc[c][1]) : __shfl_up(gacc[c][1], off, 64);
      }
      if (take) {
#pragma unroll
        for (int c = 0; c < 8; ++c) { gacc[c][0] += u[c][0]; gacc[c][1] += u[c][1]; }
      }
    }
    tail = (lane == 63) || (((starts >> (lane + 1)) & 1ull) != 0ull);
  }
  return tail;
}

__device__ __forceinline__ void lvl_scatter(float* __restrict__ tab, _Float16* __restrict__ tab16, float scale16,
                                            const uint32_t (&cidx)[8], float (&gacc)[8][2],
                                            const uint32_t (&gi)[3], bool on, int lane) {
  const bool tail = lvl_prereduce(gacc, gi, on, lane);
  if (tab16) {
    // The atomics are executed memory-side: every (lane, address) pair that is not merged in the TA costs a
    // fabric transaction, and 128 of them per point are what bound this kernel.  The two x-neighbours of a
    // cell (corners c, c^1) are ADJACENT table entries whenever the level is dense or x is even (the hash
    // multiplies x by 1), so lanes 2i / 2i+1 issue them in the SAME instruction -- first for point 2i, then
    // for point 2i+1 -- which lets the hardware merge the pair into one 64-byte request.
    const bool act = on && tail;
#pragma unroll
    for (int cp = 0; cp < 4; ++cp) {
      const int c0 = 2 * cp, c1 = 2 * cp + 1;
      const half2a v0 = {(_Float16)(gacc[c0][0] * scale16), (_Float16)(gacc[c0][1] * scale16)};
      const half2a v1 = {(_Float16)(gacc[c1][0] * scale16), (_Float16)(gacc[c1][1] * scale16)};
      const int u0 = __builtin_bit_cast(int, v0), u1 = __builtin_bit_cast(int, v1);
      const int a0 = (act && (gacc[c0][0] != 0.0f || gacc[c0][1] != 0.0f)) ? 1 : 0;
      const int a1 = (act && (gacc[c1][0] != 0.0f || gacc[c1][1] != 0.0f)) ? 1 : 0;
      const int i0 = (int)cidx[c0], i1 = (int)cidx[c1];
      const bool odd = lane & 1;
      // round A: point 2i -- even lane its own c0, odd lane the even neighbour's c1
      {
        const int ni = __builtin_amdgcn_mov_dpp(i1, 0xA0, 0xf, 0xf, true);     // quad_perm [0,0,2,2]
        const int nu = __builtin_amdgcn_mov_dpp(u1, 0xA0, 0xf, 0xf, true);
        const int na = __builtin_amdgcn_mov_dpp(a1, 0xA0, 0xf, 0xf, true);
        const int idx = odd ? ni : i0, uu = odd ? nu : u0, aa = odd ? na : a0;
        if (aa)
          __builtin_amdgcn_global_atomic_fadd_v2f16(
              (__attribute__((address_space(1))) half2a*)(tab16 + (size_t)(uint32_t)idx * 2), __builtin_bit_cast(half2a, uu));
      }
      // round B: point 2i+1 -- odd lane its own c1, even lane the odd neighbour's c0
      {
        const int ni = __builtin_amdgcn_mov_dpp(i0, 0xF5, 0xf, 0xf, true);     // quad_perm [1,1,3,3]
        const int nu = __builtin_amdgcn_mov_dpp(u0, 0xF5, 0xf, 0xf, true);
        const int na = __builtin_amdgcn_mov_dpp(a0, 0xF5, 0xf, 0xf, true);
        const int idx = odd ? i1 : ni, uu = odd ? u1 : nu, aa = odd ? a1 : na;
        if (aa)
          __builtin_amdgcn_global_atomic_fadd_v2f16(
              (__attribute__((address_space(1))) half2a*)(tab16 + (size_t)(uint32_t)idx * 2), __builtin_bit_cast(half2a, uu));
      }
    }
  } else if (on && tail) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      float* gp = tab + (size_t)cidx[c] * 2;
      if (gacc[c][0] != 0.0f) atomicAdd(gp, gacc[c][0]);
      if (gacc[c][1] != 0.0f) atomicAdd(gp + 1, gacc[c][1]);
    }
  }
}

}  // namespace
