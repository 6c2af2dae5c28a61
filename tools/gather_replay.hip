// Gather-only REPLAY of the hash grid's own index stream: the ceiling `neus_point_kernel` / `neus_encode_levels_kernel`
// are quoted against (bench.py `roofline_other`, profiles/r05_gather_replay.json).
//
// Round 4's ceiling (tools/gather_bench.hip, "corner8") drew independent random points; the production kernel beat it by
// up to 1.8x, because a wave's 64 lanes are consecutive samples of ONE ray (coherent cells, merged x-neighbours): a
// comparator that does not do the kernel's work is no ceiling.  Here the comparator IS the kernel's memory work and
// nothing else: `gr_index_stream` writes the 16 x 8 table indices of every in-bound sample point of a real batch (the
// production index arithmetic: csrc/neus_common.h grid_index, tcnn grid.h restated) and the replay kernels issue
// exactly those 4-byte loads, summed into one word per lane -- no interpolation weights, no SDF layer, no colour MLP --
//   gr_replay_point_major   lane per point, 16 levels x 8 loads: the access order of the round-4 forward;
//   gr_replay_level_major   work item = (level, 256-point chunk) in the XCD-consecutive order of
//                           neus_encode_levels_kernel (levels [l0, 16)); the dense levels below l0 are replayed
//                           point-major (gr_replay_point_major with l1 = l0), as neus_point_kernel gathers them.
// An index is read with one coalesced 4-byte load per gather (idx[level][corner][point]); that stream is traffic the real
// kernels do not have, so the replay errs on the slow side -- a production kernel may approach, not beat, what its own
// loads cost here by more than that share.
//
// Build (tools/gather_replay.py does it):  hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/gather_replay.hip -o tools/gather_replay.so
// Measurement tool, not part of libgoslam_hip.so.
#include "../go_slam_amd/csrc/neus_common.h"

namespace {

__global__ __launch_bounds__(256) void index_stream_kernel(const float* __restrict__ view, const uint8_t* __restrict__ mask,
                                                           uint32_t* __restrict__ idx, int np, gs_grid_meta m) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= np) return;
  const bool on = mask[i] != 0;
  const float x[3] = {view[i * 3 + 0], view[i * 3 + 1], view[i * 3 + 2]};
  for (int l = 0; l < GS_GRID_LEVELS; ++l) {
    uint32_t g[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) g[d] = (uint32_t)(int)floorf(fmaf(m.scale[l], x[d], 0.5f));
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const uint32_t e = grid_index(m, l, g[0] + (c & 1), g[1] + ((c >> 1) & 1), g[2] + ((c >> 2) & 1));
      idx[((size_t)l * 8 + c) * np + i] = on ? m.offset[l] + e : 0xffffffffu;
    }
  }
}

__global__ __launch_bounds__(256) void replay_point_major_kernel(const uint32_t* __restrict__ idx,
                                                                 const uint32_t* __restrict__ tab, int np, int l0, int l1,
                                                                 uint32_t* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= np) return;
  uint32_t acc = 0;
  uint32_t e[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) e[c] = idx[((size_t)l0 * 8 + c) * np + i];
  if (e[0] == 0xffffffffu) { out[i] = 0; return; }   // out of bound: the real kernels skip the point altogether
#pragma unroll 1
  for (int l = l0; l < l1; ++l) {
    uint32_t en[8];                             // the next level's indices are requested before this level's gathers are
    const int ln = l + 1 < l1 ? l + 1 : l;      // consumed: the index stream never adds a dependent round trip per level
#pragma unroll
    for (int c = 0; c < 8; ++c) en[c] = idx[((size_t)ln * 8 + c) * np + i];
#pragma unroll
    for (int c = 0; c < 8; ++c) acc += tab[e[c]];
#pragma unroll
    for (int c = 0; c < 8; ++c) e[c] = en[c];
  }
  out[i] = acc;
}

__global__ __launch_bounds__(256) void replay_level_major_kernel(const uint32_t* __restrict__ idx,
                                                                 const uint32_t* __restrict__ tab, int np, int l0,
                                                                 int chunks_per_xcd, uint32_t* __restrict__ out) {
  const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int g = j / chunks_per_xcd, ci = j - g * chunks_per_xcd;
  const int l = l0 + g;
  const int i = (ci * 8 + x) * 256 + threadIdx.x;
  if (i >= np) return;
  uint32_t e[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) e[c] = idx[((size_t)l * 8 + c) * np + i];
  if (e[0] == 0xffffffffu) return;
  uint32_t acc = 0;
#pragma unroll
  for (int c = 0; c < 8; ++c) acc += tab[e[c]];
  __builtin_nontemporal_store(acc, out + (size_t)g * np + i);
}

}  // namespace

extern "C" int gr_index_stream(const float* view, const uint8_t* mask, uint32_t* idx, int np, const gs_grid_meta* meta,
                               void* stream) {
  index_stream_kernel<<<(np + 255) / 256, 256, 0, (hipStream_t)stream>>>(view, mask, idx, np, *meta);
  return (int)hipGetLastError();
}

extern "C" int gr_replay_point_major(const uint32_t* idx, const uint32_t* tab, int np, int l0, int l1, uint32_t* out,
                                     void* stream) {
  replay_point_major_kernel<<<(np + 255) / 256, 256, 0, (hipStream_t)stream>>>(idx, tab, np, l0, l1, out);
  return (int)hipGetLastError();
}

extern "C" int gr_replay_level_major(const uint32_t* idx, const uint32_t* tab, int np, int l0, uint32_t* out,
                                     void* stream) {
  const int cpx = (((np + 255) / 256) + 7) / 8;
  const int nl = GS_GRID_LEVELS - l0;
  if (nl <= 0) return 0;
  replay_level_major_kernel<<<8 * cpx * nl, 256, 0, (hipStream_t)stream>>>(idx, tab, np, l0, cpx, out);
  return (int)hipGetLastError();
}
