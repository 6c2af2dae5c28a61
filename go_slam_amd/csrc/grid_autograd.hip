// Generic autograd kernels of the tiny-cuda-nn HashGrid encoding, for the `tinycudann` drop-in
// (go_slam_amd/neus/tcnn_compat.py).  The reference differentiates the encoding twice on EVERY forward
// (src/InstantNeuS.py:134-148: enable_grad -> pts.requires_grad_ -> autograd.grad(create_graph=True)), so the
// drop-in needs tcnn's three grid backward passes, not only the fused NeuS-specific backward of neus_bwd.hip:
//
//   first order  (tcnn kernel_grid_backward + kernel_grid_backward_input):
//       grid_grad[corner]  += w_corner(x) * dy                       (d L / d params)
//       dx                  = sum_c dy_c * d y_c / d x               (d L / d x)
//   second order (tcnn kernel_grid_backward_input_backward_{grid,input,dLdoutput}), v = d L / d (dx):
//       ddy_c               = d y_c / d x . v                        (d L / d dy)
//       grid_grad[corner]  += dy * sum_d v_d * d w_corner / d x_d    (d L / d params)
//       dx                  = sum_c dy_c * (d^2 y_c / d x d x) v     (d L / d x; trilinear => cross terms only)
//
// One lane per point, levels streamed (16 x 8 corner gathers of one 4-byte entry = both fp16 features).  The
// table gradient goes out through the same wave-merged scatter as the fused path (neus_common.h: runs of lanes
// in one cell are pre-reduced; fp32 atomics, or tcnn's own fp16 mode with one packed atomic per entry).
#include "common.h"
#include "neus_common.h"
#include <math.h>

namespace {

struct GridBwdArgs {
  const float* x; const _Float16* grid;
  const void* dy; int dy16; float dy_scale;
  const float* v;
  float* gg32; _Float16* gg16; float gg_scale;
  float* dx; float* ddy;
  int n;
};

template <bool SECOND>
__global__ __launch_bounds__(256) void grid_backward_kernel(GridBwdArgs A, gs_grid_meta m) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const bool on = i < A.n;
  const int ii = on ? i : 0;
  const float xi[3] = {A.x[ii * 3 + 0], A.x[ii * 3 + 1], A.x[ii * 3 + 2]};
  float vv[3] = {0.f, 0.f, 0.f};
  if (SECOND) {
#pragma unroll
    for (int d = 0; d < 3; ++d) vv[d] = A.v[ii * 3 + d];
  }
  const bool want_gg = A.gg32 != nullptr || A.gg16 != nullptr;
  const bool want_vals = A.dx != nullptr || A.ddy != nullptr;
  const float s32 = A.gg32 ? A.gg_scale : 1.0f;      // the fp16 scatter applies its scale itself
  float dxa[3] = {0.f, 0.f, 0.f};
#pragma unroll 1
  for (int l = 0; l < GS_GRID_LEVELS; ++l) {
    const float scale = m.scale[l];
    float f[3];
    uint32_t g[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const float pos = fmaf(scale, xi[d], 0.5f);
      const float fl = floorf(pos);
      g[d] = (uint32_t)(int)fl;
      f[d] = pos - fl;
    }
    uint32_t cidx[8];
    grid_corners(m, l, g, cidx);
    float dyl[2];
    if (A.dy16) {
      const _Float16* p = reinterpret_cast<const _Float16*>(A.dy) + (size_t)ii * 32 + 2 * l;
      dyl[0] = (float)p[0] * A.dy_scale; dyl[1] = (float)p[1] * A.dy_scale;
    } else {
      const float* p = reinterpret_cast<const float*>(A.dy) + (size_t)ii * 32 + 2 * l;
      dyl[0] = p[0] * A.dy_scale; dyl[1] = p[1] * A.dy_scale;
    }
    if (!on) { dyl[0] = 0.f; dyl[1] = 0.f; }
    // per-dimension interpolation factors of the 8 corners: wd[d][bit]
    float wd[3][2];
#pragma unroll
    for (int d = 0; d < 3; ++d) { wd[d][0] = 1.0f - f[d]; wd[d][1] = f[d]; }

    if (want_gg) {
      float gacc[8][2];
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const int b0 = c & 1, b1 = (c >> 1) & 1, b2 = (c >> 2) & 1;
        float w;
        if (!SECOND) {
          w = wd[0][b0] * wd[1][b1] * wd[2][b2];
        } else {        // directional derivative of the corner weight along v
          const float s0 = b0 ? scale : -scale, s1 = b1 ? scale : -scale, s2 = b2 ? scale : -scale;
          w = vv[0] * s0 * (wd[1][b1] * wd[2][b2]) + vv[1] * s1 * (wd[0][b0] * wd[2][b2]) +
              vv[2] * s2 * (wd[0][b0] * wd[1][b1]);
        }
        gacc[c][0] = w * dyl[0] * s32;
        gacc[c][1] = w * dyl[1] * s32;
      }
      float* tab = A.gg32 ? A.gg32 + (size_t)m.offset[l] * 2 : nullptr;
      _Float16* tab16 = A.gg16 ? A.gg16 + (size_t)m.offset[l] * 2 : nullptr;
      lvl_scatter(tab, tab16, A.gg_scale, cidx, gacc, g, on, lane);
    }
    if (want_vals) {
      const _Float16* tabv = A.grid + (size_t)m.offset[l] * 2;
      float val[8][2];
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const uint32_t raw = *reinterpret_cast<const uint32_t*>(tabv + (size_t)cidx[c] * 2);
        val[c][0] = (float)__builtin_bit_cast(_Float16, (uint16_t)(raw & 0xffffu));
        val[c][1] = (float)__builtin_bit_cast(_Float16, (uint16_t)(raw >> 16));
      }
      // first derivatives d y_f / d x_gd (same accumulation order as the forward kernel's dy_dx)
      float dv[3][2];
#pragma unroll
      for (int gd = 0; gd < 3; ++gd) {
        const int o0 = (gd == 0) ? 1 : 0, o1 = (gd == 2) ? 1 : 2;
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float w = scale;
          w = w * wd[o0][k & 1];
          w = w * wd[o1][(k >> 1) & 1];
          const int cl = ((k & 1) << o0) | (((k >> 1) & 1) << o1);
          const int cr = cl | (1 << gd);
          a0 = fmaf(w, val[cr][0] - val[cl][0], a0);
          a1 = fmaf(w, val[cr][1] - val[cl][1], a1);
        }
        dv[gd][0] = a0;
        dv[gd][1] = a1;
      }
      if (!SECOND) {
#pragma unroll
        for (int d = 0; d < 3; ++d) dxa[d] += dyl[0] * dv[d][0] + dyl[1] * dv[d][1];
      } else {
        if (A.ddy && on) {
          A.ddy[(size_t)i * 32 + 2 * l + 0] = vv[0] * dv[0][0] + vv[1] * dv[1][0] + vv[2] * dv[2][0];
          A.ddy[(size_t)i * 32 + 2 * l + 1] = vv[0] * dv[0][1] + vv[1] * dv[1][1] + vv[2] * dv[2][1];
        }
        if (A.dx) {
          // mixed second derivatives: d^2 y_f / d x_a d x_b = scale^2 sum_k w_third[k] *
          //   (val[a=1,b=1] - val[a=1,b=0] - val[a=0,b=1] + val[a=0,b=0])
#pragma unroll
          for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = a + 1; b < 3; ++b) {
              const int t = 3 - a - b;
              float h0 = 0.f, h1 = 0.f;
#pragma unroll
              for (int k = 0; k < 2; ++k) {
                const int base = k << t;
                const int c11 = base | (1 << a) | (1 << b), c10 = base | (1 << a), c01 = base | (1 << b), c00 = base;
                const float w = scale * scale * wd[t][k];
                h0 = fmaf(w, (val[c11][0] - val[c10][0]) - (val[c01][0] - val[c00][0]), h0);
                h1 = fmaf(w, (val[c11][1] - val[c10][1]) - (val[c01][1] - val[c00][1]), h1);
              }
              const float hab = dyl[0] * h0 + dyl[1] * h1;
              dxa[a] += hab * vv[b];
              dxa[b] += hab * vv[a];
            }
        }
      }
    }
  }
  if (A.dx && on) {
#pragma unroll
    for (int d = 0; d < 3; ++d) A.dx[(size_t)i * 3 + d] = dxa[d];
  }
}

gs_grid_meta host_meta_bwd() {
  gs_grid_meta m;
  gs_grid_meta_default(&m);
  return m;
}

}  // namespace

extern "C" int gs_grid_backward(const float* x, const void* grid, const void* dy, int dy_dtype, float dy_scale,
                                const float* v, void* grid_grad, int grid_grad_dtype, float grid_grad_scale,
                                float* dx, float* ddy, int n, gs_stream_t stream) {
  GS_REQUIRE(x && dy, "grid_backward: null pointer");
  GS_REQUIRE(dy_dtype == GS_F16 || dy_dtype == GS_F32, "grid_backward: dy must be f16 or f32");
  GS_REQUIRE(!grid_grad || grid_grad_dtype == GS_F16 || grid_grad_dtype == GS_F32,
             "grid_backward: grid_grad must be f16 or f32");
  GS_REQUIRE(grid || (!dx && !ddy), "grid_backward: dx / ddy need the grid values");
  GS_REQUIRE(v || !ddy, "grid_backward: ddy is a second-order output (needs v)");
  GS_REQUIRE(n >= 0, "grid_backward: bad n");
  if (n == 0) return GS_OK;
  static const gs_grid_meta meta = host_meta_bwd();
  GridBwdArgs A;
  A.x = x; A.grid = (const _Float16*)grid; A.dy = dy; A.dy16 = dy_dtype == GS_F16; A.dy_scale = dy_scale; A.v = v;
  A.gg32 = (grid_grad && grid_grad_dtype == GS_F32) ? (float*)grid_grad : nullptr;
  A.gg16 = (grid_grad && grid_grad_dtype == GS_F16) ? (_Float16*)grid_grad : nullptr;
  A.gg_scale = grid_grad_scale; A.dx = dx; A.ddy = ddy; A.n = n;
  const unsigned blocks = (unsigned)((n + 255) / 256);
  if (v)
    grid_backward_kernel<true><<<blocks, 256, 0, (hipStream_t)stream>>>(A, meta);
  else
    grid_backward_kernel<false><<<blocks, 256, 0, (hipStream_t)stream>>>(A, meta);
  GS_CHECK_LAUNCH("grid_backward");
  return GS_OK;
}
