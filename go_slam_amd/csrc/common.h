// Shared device/host helpers for libgoslam_hip.so (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/goslam_hip.h"

#define GS_WAVE 64

// ---------------------------------------------------------------- host side errors -----
void gs_set_error(const char* fmt, ...);

#define GS_REQUIRE(cond, ...)                      \
  do {                                             \
    if (!(cond)) {                                 \
      gs_set_error(__VA_ARGS__);                   \
      return GS_ERR_INVALID_ARG;                   \
    }                                              \
  } while (0)

// Kernel timer (capi.hip; gs_timing_begin / _end / _read in include/goslam_hip.h): while a thread has it switched on,
// every GS_CHECK_LAUNCH records a HIP event on the timed stream, so consecutive events bracket each kernel.  Off (the
// default) it is one thread-local load.
extern thread_local int gs_timing_on;
void gs_timing_mark(const char* name);

// placed in front of a launch whose duration must not include whatever the caller enqueued before it
#define GS_TIMING_PRE()                      \
  do {                                       \
    if (gs_timing_on) gs_timing_mark("");    \
  } while (0)

#define GS_CHECK_LAUNCH(name)                                                   \
  do {                                                                          \
    /* the launch's status is read BEFORE the timer's own runtime calls can overwrite the thread's last error */ \
    hipError_t e_ = hipGetLastError();                                          \
    if (gs_timing_on) gs_timing_mark(name);                                     \
    if (e_ != hipSuccess) {                                                     \
      gs_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));       \
      return GS_ERR_LAUNCH;                                                     \
    }                                                                           \
  } while (0)

// Raises a kernel's dynamic-LDS limit the first time it is launched on EACH device of the process (the attribute is
// per device; tracking on cuda:0 and mapping on cuda:1 in one process is a supported configuration).  Idempotent, so
// two threads racing on the first launch are harmless.
#include <atomic>
struct GsLdsLimit {
  // dynamic-LDS limit raised so far for one kernel, per device (devices 0-63; a later call that needs more raises again)
  std::atomic<unsigned int> raised[64];
  GsLdsLimit() { for (auto& r : raised) r.store(0u, std::memory_order_relaxed); }
  int raise(const void* fn, size_t bytes, const char* what) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    std::atomic<unsigned int>& slot = raised[dev & 63];
    if (dev < 64 && slot.load(std::memory_order_acquire) >= bytes) return GS_OK;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) {
      gs_set_error("%s: cannot raise the dynamic LDS limit to %zu bytes", what, bytes);
      return GS_ERR_LAUNCH;
    }
    unsigned int cur = slot.load(std::memory_order_relaxed);
    while (cur < bytes && !slot.compare_exchange_weak(cur, (unsigned int)bytes, std::memory_order_release)) {}
    return GS_OK;
  }
};

static inline int gs_cdiv(int a, int b) { return (a + b - 1) / b; }
static inline size_t gs_align(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// ---------------------------------------------------------------- SE3 on the device ----
// Pose layout [tx,ty,tz,qx,qy,qz,qw] (world->camera), see SURVEY App. A.  All of these are
// written so that, compiled with -ffp-contract=off, they round exactly like the op-by-op fp32
// restatement in oracle/se3.py.

__device__ __forceinline__ void gs_act_so3(const float* q, const float* X, float* Y) {
  float uv0 = 2.0f * (q[1] * X[2] - q[2] * X[1]);
  float uv1 = 2.0f * (q[2] * X[0] - q[0] * X[2]);
  float uv2 = 2.0f * (q[0] * X[1] - q[1] * X[0]);
  Y[0] = X[0] + q[3] * uv0 + (q[1] * uv2 - q[2] * uv1);
  Y[1] = X[1] + q[3] * uv1 + (q[2] * uv0 - q[0] * uv2);
  Y[2] = X[2] + q[3] * uv2 + (q[0] * uv1 - q[1] * uv0);
}

__device__ __forceinline__ void gs_act_se3(const float* t, const float* q, const float* X, float* Y) {
  float R[3];
  gs_act_so3(q, X, R);
  Y[0] = R[0] + X[3] * t[0];
  Y[1] = R[1] + X[3] * t[1];
  Y[2] = R[2] + X[3] * t[2];
  Y[3] = X[3];
}

// dual adjoint applied to a 6-covector
__device__ __forceinline__ void gs_adj_se3(const float* t, const float* q, const float* X, float* Y) {
  float qinv[4] = {-q[0], -q[1], -q[2], q[3]};
  gs_act_so3(qinv, &X[0], &Y[0]);
  gs_act_so3(qinv, &X[3], &Y[3]);
  float u[3], v[3];
  u[0] = t[2] * X[1] - t[1] * X[2];
  u[1] = t[0] * X[2] - t[2] * X[0];
  u[2] = t[1] * X[0] - t[0] * X[1];
  gs_act_so3(qinv, u, v);
  Y[3] = Y[3] + v[0];
  Y[4] = Y[4] + v[1];
  Y[5] = Y[5] + v[2];
}

__device__ __forceinline__ void gs_rel_se3(const float* ti, const float* qi, const float* tj,
                                           const float* qj, float* tij, float* qij) {
  qij[0] = -qj[3] * qi[0] + qj[0] * qi[3] - qj[1] * qi[2] + qj[2] * qi[1];
  qij[1] = -qj[3] * qi[1] + qj[1] * qi[3] - qj[2] * qi[0] + qj[0] * qi[2];
  qij[2] = -qj[3] * qi[2] + qj[2] * qi[3] - qj[0] * qi[1] + qj[1] * qi[0];
  qij[3] = qj[3] * qi[3] + qj[0] * qi[0] + qj[1] * qi[1] + qj[2] * qi[2];
  float r[3];
  gs_act_so3(qij, ti, r);
  tij[0] = tj[0] - r[0];
  tij[1] = tj[1] - r[1];
  tij[2] = tj[2] - r[2];
}

__device__ __forceinline__ void gs_quat_mul(const float* a, const float* b, float* o) {
  o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  o[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
  o[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
  o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
}

__device__ __forceinline__ void gs_exp_so3(const float* phi, float* q) {
  float theta_sq = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2];
  float theta_p4 = theta_sq * theta_sq;
  float theta = sqrtf(theta_sq);
  float imag, real;
  if (theta_sq < 1e-8f) {
    imag = 0.5f - (1.0f / 48.0f) * theta_sq + (1.0f / 3840.0f) * theta_p4;
    real = 1.0f - (1.0f / 8.0f) * theta_sq + (1.0f / 384.0f) * theta_p4;
  } else {
    imag = sinf(0.5f * theta) / theta;
    real = cosf(0.5f * theta);
  }
  q[0] = imag * phi[0];
  q[1] = imag * phi[1];
  q[2] = imag * phi[2];
  q[3] = real;
}

__device__ __forceinline__ void gs_cross_inplace(const float* a, float* b) {
  float x0 = a[1] * b[2] - a[2] * b[1];
  float x1 = a[2] * b[0] - a[0] * b[2];
  float x2 = a[0] * b[1] - a[1] * b[0];
  b[0] = x0; b[1] = x1; b[2] = x2;
}

__device__ __forceinline__ void gs_exp_se3(const float* xi, float* t, float* q) {
  gs_exp_so3(xi + 3, q);
  float tau[3] = {xi[0], xi[1], xi[2]};
  float phi[3] = {xi[3], xi[4], xi[5]};
  float theta_sq = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2];
  float theta = sqrtf(theta_sq);
  t[0] = tau[0]; t[1] = tau[1]; t[2] = tau[2];
  if (theta > 1e-4f) {
    float a = (1.0f - cosf(theta)) / theta_sq;
    gs_cross_inplace(phi, tau);
    t[0] = t[0] + a * tau[0];
    t[1] = t[1] + a * tau[1];
    t[2] = t[2] + a * tau[2];
    float b = (theta - sinf(theta)) / (theta * theta_sq);
    gs_cross_inplace(phi, tau);
    t[0] = t[0] + b * tau[0];
    t[1] = t[1] + b * tau[1];
    t[2] = t[2] + b * tau[2];
  }
}

// left retraction exp(xi) * (t,q)
__device__ __forceinline__ void gs_retr_se3(const float* xi, const float* t, const float* q,
                                            float* t1, float* q1) {
  float dt[3], dq[4];
  gs_exp_se3(xi, dt, dq);
  gs_quat_mul(dq, q, q1);
  float r[3];
  gs_act_so3(dq, t, r);
  t1[0] = r[0] + dt[0];
  t1[1] = r[1] + dt[1];
  t1[2] = r[2] + dt[2];
}

// ---------------------------------------------------------------- wave64 reductions ----
// Sum across the 64 lanes of a wave: 4 DPP steps inside each 16-lane row (quad_perm xor 1,
// xor 2, row_half_mirror, row_mirror) then the four row totals are combined through scalar
// registers.  Every lane returns the full sum; the order is fixed => deterministic.
template <int CTRL>
__device__ __forceinline__ float gs_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

__device__ __forceinline__ float gs_wave_sum(float v) {
  v = v + gs_dpp<0xB1>(v);   // quad_perm [1,0,3,2]
  v = v + gs_dpp<0x4E>(v);   // quad_perm [2,3,0,1]
  v = v + gs_dpp<0x141>(v);  // row_half_mirror
  v = v + gs_dpp<0x140>(v);  // row_mirror
  float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
  float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
  float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
  float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
  return (r0 + r1) + (r2 + r3);
}

// sigmoid / tanh of an fp32 value whose result is rounded to fp16 by the caller: v_exp_f32 + v_rcp_f32 (1 ulp) instead
// of the IEEE division sequence (v_div_scale / v_div_fmas / v_div_fixup + Newton steps: ~10 instructions per value,
// measured 17 us of the 53 us global-context kernel and a tenth of the GRU convolutions' epilogues).  After the fp16
// rounding the result differs from the exactly divided one for ~1 element in 10^4 (by one fp16 ulp).
__device__ __forceinline__ float gs_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float gs_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }

// Sums of N per-lane values over the wave, "reduce-scatter" form: instead of N full butterflies (gs_wave_sum: 11
// instructions per value) the value array is halved at every lane-bit step -- a lane keeps the even or the odd element of
// each pair and sends the other one to its partner -- so N values cost ~N exchanges in all.  On return lane L holds, in
// v[j], the wave total of value index 64 j + bitrev6(L) (j < ceil(N / 64); indices >= N are padding).  Fixed order =>
// deterministic.
template <int N, int MASK>
__device__ __forceinline__ void gs_rs_step(float* v, int lane) {
  constexpr int NN = (N + 1) / 2;
  const bool up = (lane & MASK) != 0;
#pragma unroll
  for (int i = 0; i < NN; ++i) {
    const float a = v[2 * i];
    const float b = (2 * i + 1 < N) ? v[2 * i + 1] : 0.0f;
    const float send = up ? a : b;
    const float keep = up ? b : a;
    v[i] = keep + __shfl_xor(send, MASK, 64);
  }
}
template <int N>
__device__ __forceinline__ void gs_wave_reduce_scatter(float (&v)[N], int lane) {
  constexpr int n1 = (N + 1) / 2, n2 = (n1 + 1) / 2, n3 = (n2 + 1) / 2, n4 = (n3 + 1) / 2, n5 = (n4 + 1) / 2;
  gs_rs_step<N, 32>(v, lane);
  gs_rs_step<n1, 16>(v, lane);
  gs_rs_step<n2, 8>(v, lane);
  gs_rs_step<n3, 4>(v, lane);
  gs_rs_step<n4, 2>(v, lane);
  gs_rs_step<n5, 1>(v, lane);
}
__device__ __forceinline__ int gs_bitrev6(int lane) { return (int)(__brev((unsigned)lane) >> 26); }

__device__ __forceinline__ double gs_wave_sum_f64(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__device__ __forceinline__ void gs_atomic_add_f64(double* p, double v) {
  // global_atomic_add_f64 (hardware fp64 atomic on gfx950), relaxed, device scope
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
