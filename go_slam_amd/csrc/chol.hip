// Dense fp64 Cholesky solve of the reduced camera system, on the device.
// Stands in for Eigen::SimplicialLLT on the host (reference: src/lib/droid_kernels.cu:1192-1213):
//   L = A;  diag(L) += ep + lm * diag(L);  LLT;  x = solve(b);  failure (pivot <= 0) => x = 0.
//
// The (6P x 6P) matrix lives in HBM/L2 as a dense row-major fp64 array whose LOWER triangle is
// valid.  Right-looking blocked factorisation, NB = 32:
//   chol_panel_kernel   : every workgroup (one wave) re-factors the 32x32 diagonal block in LDS
//                         (cheaper than a dependent launch) and solves 64 rows of the panel
//   chol_trail_kernel   : 64x64 tiles of the trailing matrix, A22 -= L21 L21^T, operands in LDS
//   chol_solve_kernel   : one workgroup, blocked forward + backward substitution, writes fp32 dx
// 2 launches per panel; 6P = 150 (frontend window) is 5 panels, 6P = 1200 (global BA) 38.
#include "common.h"

namespace {

constexpr int NB = 32;     // panel width
constexpr int PR = 64;     // panel rows per workgroup (one wave)
constexpr int TT = 64;     // trailing tile edge
constexpr int SB = 64;     // substitution block

__global__ void chol_damp_kernel(double* __restrict__ A, int n, double lm, double ep, int32_t* fail_flag) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i == 0) *fail_flag = 0;
  if (i < n) {
    const double d = A[(size_t)i * n + i];
    A[(size_t)i * n + i] = d + (ep + lm * d);
  }
}

__device__ __forceinline__ double readlane_f64(double v, int lane) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned int lo = __builtin_amdgcn_readlane((int)(u & 0xffffffffu), lane);
  const unsigned int hi = __builtin_amdgcn_readlane((int)(u >> 32), lane);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// Factor the diagonal block [k0,k0+nb) (every workgroup redundantly -- cheaper than a dependent
// launch) and compute L21 = A21 * L11^-T for this workgroup's PR rows.
// The 32x32 factorisation is wave-synchronous and register-resident: lane i holds row i
// (32 doubles), pivots and column entries are broadcast with v_readlane, no LDS, no barriers
// (~1.5 us instead of 32 x 3 barrier-separated LDS sweeps).
__global__ __launch_bounds__(64) void chol_panel_kernel(double* __restrict__ A, int n, int k0,
                                                        int32_t* fail_flag) {
  __shared__ double D[NB][NB + 1];
  const int lane = threadIdx.x;
  const int nb = min(NB, n - k0);
  double a[NB];
  {
    const int r = lane & (NB - 1);
    const bool live = (lane < NB) && (r < nb);
    const double* Ar = A + (size_t)(k0 + (live ? r : 0)) * n + k0;
#pragma unroll
    for (int c = 0; c < NB; ++c) {
      double v = (live && c <= r && c < nb) ? Ar[c] : 0.0;
      if (!live && c == r) v = 1.0;            // identity padding keeps the recurrence well-defined
      a[c] = v;
    }
  }
  bool bad = false;
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const double piv = readlane_f64(a[j], j);
    if (j < nb && !(piv > 0.0) && piv == piv) bad = true;   // pivot <= 0 (NaN falls through like Eigen)
    const double dj = sqrt(piv);
    if (lane == j) a[j] = dj;
    else if (lane > j) a[j] = a[j] / dj;
#pragma unroll
    for (int c = j + 1; c < NB; ++c) {
      const double lcj = readlane_f64(a[j], c);
      if (lane >= c) a[c] -= a[j] * lcj;
    }
  }
  if (lane < NB) {
#pragma unroll
    for (int c = 0; c < NB; ++c) D[lane][c] = a[c];
  }
  __syncthreads();
  if (blockIdx.x == 0) {
    if (lane < nb) {
      double* Ar = A + (size_t)(k0 + lane) * n + k0;
#pragma unroll
      for (int c = 0; c < NB; ++c)
        if (c <= lane) Ar[c] = a[c];
    }
    if (lane == 0 && bad) *fail_flag = 1;
  }
  // panel rows
  const int row = k0 + nb + blockIdx.x * PR + lane;
  if (row < n) {
    double x[NB];
    double* Ar = A + (size_t)row * n + k0;
#pragma unroll
    for (int c = 0; c < NB; ++c) x[c] = (c < nb) ? Ar[c] : 0.0;
#pragma unroll
    for (int c = 0; c < NB; ++c) {
      if (c < nb) {
        double s = x[c];
#pragma unroll
        for (int t = 0; t < NB; ++t)
          if (t < c) s -= x[t] * D[c][t];
        x[c] = s / D[c][c];
      }
    }
#pragma unroll
    for (int c = 0; c < NB; ++c)
      if (c < nb) Ar[c] = x[c];
  }
}

// A22[r][c] -= sum_k L21[r][k] L21[c][k] over lower tiles of the trailing matrix.
__global__ __launch_bounds__(256) void chol_trail_kernel(double* __restrict__ A, int n, int k0, int nb) {
  __shared__ double Lr[TT][NB + 1];
  __shared__ double Lc[TT][NB + 1];
  // decode (ti,tj), tj <= ti, from the flat lower-triangular tile index
  const int t = blockIdx.x;
  int ti = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
  while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
  while (ti * (ti + 1) / 2 > t) --ti;
  const int tj = t - ti * (ti + 1) / 2;
  const int s0 = k0 + nb;
  const int r0 = s0 + ti * TT, c0 = s0 + tj * TT;
  const int tid = threadIdx.x;
  for (int idx = tid; idx < TT * NB; idx += 256) {
    const int r = idx / NB, k = idx % NB;
    Lr[r][k] = (r0 + r < n && k < nb) ? A[(size_t)(r0 + r) * n + (k0 + k)] : 0.0;
    Lc[r][k] = (c0 + r < n && k < nb) ? A[(size_t)(c0 + r) * n + (k0 + k)] : 0.0;
  }
  __syncthreads();
  const int tr = (tid / 16) * 4, tc = (tid % 16) * 4;
  double acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;
#pragma unroll 4
  for (int k = 0; k < NB; ++k) {
    double a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = Lr[tr + i][k]; b[i] = Lc[tc + i][k]; }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = fma(a[i], b[j], acc[i][j]);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + tr + i;
    if (r >= n) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = c0 + tc + j;
      if (c <= r) A[(size_t)r * n + c] -= acc[i][j];
    }
  }
}

// Blocked forward (L y = b) and backward (L^T x = y) substitution by one workgroup.  The
// sequential part of each 64-wide diagonal block is wave-synchronous and register-resident:
// forward, lane i holds ROW i of the block and the solved entries are broadcast with v_readlane;
// backward, lane j holds COLUMN j (coalesced loads) -- 64 steps of {readlane, fma} instead of 64
// barrier-separated LDS sweeps.  The off-diagonal updates use all 256 threads.
__global__ __launch_bounds__(256) void chol_solve_kernel(const double* __restrict__ L, double* __restrict__ b,
                                                         int n, float* __restrict__ dx,
                                                         int32_t* fail_flag, int32_t* fail_count) {
  __shared__ double y[SB];
  const int tid = threadIdx.x;
  if (*fail_flag) {   // reference: zero update on failure (droid_kernels.cu:1207-1210)
    for (int i = tid; i < n; i += 256) dx[i] = 0.0f;
    if (tid == 0) *fail_count += 1;
    return;
  }
  const int nblk = (n + SB - 1) / SB;
  // ---- forward
  for (int kb = 0; kb < nblk; ++kb) {
    const int k0 = kb * SB, nb = min(SB, n - k0);
    if (tid < 64) {
      const int i = tid;
      const bool live = i < nb;
      double a[SB];
      const double* Lr = L + (size_t)(k0 + (live ? i : 0)) * n + k0;
#pragma unroll
      for (int c = 0; c < SB; ++c) a[c] = (live && c < i) ? Lr[c] : 0.0;
      const double inv_dg = live ? 1.0 / Lr[i] : 1.0;
      double bv = live ? b[k0 + i] : 0.0;
#pragma unroll
      for (int j = 0; j < SB; ++j) {
        const double yj = readlane_f64(bv * inv_dg, j);
        if (i == j) bv = yj;
        else if (i > j) bv -= a[j] * yj;
      }
      y[i] = bv;
      if (live) b[k0 + i] = bv;
    }
    __syncthreads();
    // b[r] -= L[r][k0:k0+nb] . y  for rows below the block
    for (int r = k0 + nb + tid; r < n; r += 256) {
      const double* Lr = L + (size_t)r * n + k0;
      double s = 0.0;
      for (int c = 0; c < nb; ++c) s = fma(Lr[c], y[c], s);
      b[r] -= s;
    }
    __syncthreads();
  }
  // ---- backward
  for (int kb = nblk - 1; kb >= 0; --kb) {
    const int k0 = kb * SB, nb = min(SB, n - k0);
    if (tid < 64) {
      const int j = tid;
      const bool live = j < nb;
      double c_[SB];                      // column j of the block: L[k0+i][k0+j], i > j
#pragma unroll
      for (int i = 0; i < SB; ++i) c_[i] = (live && i > j && i < nb) ? L[(size_t)(k0 + i) * n + k0 + j] : 0.0;
      const double inv_dg = live ? 1.0 / L[(size_t)(k0 + j) * n + k0 + j] : 1.0;
      double yv = live ? b[k0 + j] : 0.0;
#pragma unroll
      for (int i = SB - 1; i >= 0; --i) {
        const double xi = readlane_f64(yv * inv_dg, i);
        if (j == i) yv = xi;
        else if (j < i) yv -= c_[i] * xi;
      }
      y[j] = yv;
      if (live) { b[k0 + j] = yv; dx[k0 + j] = (float)yv; }
    }
    __syncthreads();
    // b[c] -= sum_r L[k0+r][c] * x[r]  for columns left of the block
    for (int c = tid; c < k0; c += 256) {
      double s = 0.0;
      for (int r = 0; r < nb; ++r) s = fma(L[(size_t)(k0 + r) * n + c], y[r], s);
      b[c] -= s;
    }
    __syncthreads();
  }
}


// ---- single-launch path for the frontend window (6P <= 192) -------------------------------
// The multi-kernel path above spends its time in launch-to-launch dependencies: 6P = 150 is 5 panel
// + 4 trailing launches + the solve, ~230 us for 1.1 MFLOP.  Here ONE workgroup keeps the packed
// lower triangle (n(n+1)/2 doubles, 90 KB at n = 150; CDNA4's 160 KB LDS holds n <= 192) in LDS
// and does damping, the blocked factorisation, both substitutions and the failure fallback in one
// launch; the only global traffic is one read of H and b and the fp32 dx.
constexpr int SMALL_N = 192;
#ifdef CHOL_TIMING   // scratch/chol_bench.hip only: per-phase wall-clock stamps (100 MHz)
__device__ long long g_chol_t[64];
#define CHOL_STAMP(i) do { if (threadIdx.x == 0) g_chol_t[i] = wall_clock64(); } while (0)
#else
#define CHOL_STAMP(i) do { } while (0)
#endif

__device__ __forceinline__ int tri(int r, int c) { return ((r * (r + 1)) >> 1) + c; }

// sqrt and reciprocal sqrt of a pivot from v_rsq_f64 + two coupled Goldschmidt steps (full fp64
// accuracy, ~12 dependent FMAs instead of the ~35-instruction sqrt + divide sequences).
__device__ __forceinline__ void sqrt_rsqrt(double x, double& s, double& rs) {
  const double r = __builtin_amdgcn_rsq(x);
  double g = x * r, h = 0.5 * r;
  double e = fma(-g, h, 0.5);
  g = fma(g, e, g); h = fma(h, e, h);
  e = fma(-g, h, 0.5);
  g = fma(g, e, g); h = fma(h, e, h);
  s = g; rs = h + h;
}

__global__ __launch_bounds__(256) void chol_small_kernel(const double* __restrict__ H, const double* __restrict__ bg,
                                                         int n, double lm, double ep, float* __restrict__ dx,
                                                         int32_t* fail_flag, int32_t* fail_count) {
  extern __shared__ double sm[];
  double* Lp = sm;                                 // packed lower triangle, row r at r(r+1)/2
  double* bv = sm + ((n * (n + 1)) >> 1);          // right-hand side / solution
  double* invd = bv + n;                           // 1 / L[j][j]
  __shared__ double ys[NB];
  __shared__ int s_bad;
  const int tid = threadIdx.x, lane = tid & 63;
  if (tid == 0) s_bad = 0;
  CHOL_STAMP(0);
  // lower triangle only, 16 independent loads in flight per thread: a lone workgroup is bound by
  // global-load latency, not bandwidth (a one-load-per-iteration loop cost ~90 us here)
  const int npk = (n * (n + 1)) >> 1;
  for (int base = 0; base < npk; base += 256 * 16) {
    double v[16];
    int rr[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int idx = base + u * 256 + tid;
      int r = (int)((sqrtf(8.0f * (float)idx + 1.0f) - 1.0f) * 0.5f);
      while (((r + 1) * (r + 2)) >> 1 <= idx) ++r;
      while ((r * (r + 1)) >> 1 > idx) --r;
      const int c = idx - ((r * (r + 1)) >> 1);
      rr[u] = (c == r) ? 1 : 0;
      v[u] = idx < npk ? H[(size_t)r * n + c] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int idx = base + u * 256 + tid;
      if (idx < npk) Lp[idx] = rr[u] ? v[u] + (ep + lm * v[u]) : v[u];
    }
  }
  for (int i = tid; i < n; i += 256) bv[i] = bg[i];
  __syncthreads();
  CHOL_STAMP(1);

  for (int k0 = 0; k0 < n; k0 += NB) {
    const int nb = min(NB, n - k0);
    const int rem = n - k0 - nb;
    CHOL_STAMP(2 + 4 * (k0 / NB));
    // (a) diagonal block: wave 0, register-resident, lane r holds row r
    if (tid < 64) {
      const int r = lane & (NB - 1);
      const bool live = (lane < NB) && (r < nb);
      double a[NB];
      const double* Ar = Lp + tri(k0 + (live ? r : 0), k0);
#pragma unroll
      for (int c = 0; c < NB; ++c) {
        double v = (live && c <= r) ? Ar[c] : 0.0;
        if (!live && c == r) v = 1.0;
        a[c] = v;
      }
      bool bad = false;
      double my_inv = 1.0;
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const double piv = readlane_f64(a[j], j);
        if (j < nb && !(piv > 0.0) && piv == piv) bad = true;
        double dj, inv;
        sqrt_rsqrt(piv, dj, inv);
        if (lane == j) { a[j] = dj; my_inv = inv; }
        else if (lane > j) a[j] = a[j] * inv;
#pragma unroll
        for (int c = j + 1; c < NB; ++c) {
          const double lcj = readlane_f64(a[j], c);
          if (lane >= c) a[c] = fma(-a[j], lcj, a[c]);
        }
      }
      if (live) {
        double* Aw = Lp + tri(k0 + r, k0);
#pragma unroll
        for (int c = 0; c < NB; ++c)
          if (c <= r) Aw[c] = a[c];
        invd[k0 + r] = my_inv;
      }
      if (bad && lane == 0) s_bad = 1;
    }
    __syncthreads();
    CHOL_STAMP(3 + 4 * (k0 / NB));
    // (b) panel rows: L21 = A21 L11^-T, one row per thread
    if (tid < rem) {
      double* Ar = Lp + tri(k0 + nb + tid, k0);
      double x[NB];
#pragma unroll
      for (int c = 0; c < NB; ++c) x[c] = (c < nb) ? Ar[c] : 0.0;
#pragma unroll
      for (int c = 0; c < NB; ++c) {
        if (c < nb) {
          const double* Dc = Lp + tri(k0 + c, k0);
          double sacc = x[c];
#pragma unroll
          for (int t = 0; t < NB; ++t)
            if (t < c) sacc = fma(-x[t], Dc[t], sacc);
          x[c] = sacc * invd[k0 + c];
        }
      }
#pragma unroll
      for (int c = 0; c < NB; ++c)
        if (c < nb) Ar[c] = x[c];
    }
    __syncthreads();
    CHOL_STAMP(4 + 4 * (k0 / NB));
    // (c) trailing update A22 -= L21 L21^T on 4x4 register tiles of the lower triangle
    if (rem > 0) {
      const int T = (rem + 3) >> 2, ntile = (T * (T + 1)) >> 1;
      const int s0 = k0 + nb;
      for (int t = tid; t < ntile; t += 256) {
        int ti = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
        while (((ti + 1) * (ti + 2)) >> 1 <= t) ++ti;
        while ((ti * (ti + 1)) >> 1 > t) --ti;
        const int tj = t - ((ti * (ti + 1)) >> 1);
        const int r0 = s0 + 4 * ti, c0 = s0 + 4 * tj;
        const double* pr[4];
        const double* pc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          pr[i] = Lp + tri(min(r0 + i, n - 1), k0);
          pc[i] = Lp + tri(min(c0 + i, n - 1), k0);
        }
        double acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;
        for (int k = 0; k < nb; ++k) {
          double a[4], b[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) { a[i] = pr[i][k]; b[i] = pc[i][k]; }
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = fma(a[i], b[j], acc[i][j]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int r = r0 + i;
          if (r >= n) continue;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int c = c0 + j;
            if (c <= r) Lp[tri(r, c)] -= acc[i][j];
          }
        }
      }
    }
    __syncthreads();
  }

  if (s_bad) {   // reference: zero update on failure (droid_kernels.cu:1207-1210)
    for (int i = tid; i < n; i += 256) dx[i] = 0.0f;
    if (tid == 0) { *fail_flag = 1; *fail_count += 1; }
    return;
  }
  if (tid == 0) *fail_flag = 0;
  CHOL_STAMP(40);

  // ---- forward substitution L y = b
  for (int k0 = 0; k0 < n; k0 += NB) {
    const int nb = min(NB, n - k0);
    if (tid < 64) {
      const int i = lane & (NB - 1);
      const bool live = (lane < NB) && (i < nb);
      double a[NB];
      const double* Lr = Lp + tri(k0 + (live ? i : 0), k0);
#pragma unroll
      for (int c = 0; c < NB; ++c) a[c] = (live && c < i) ? Lr[c] : 0.0;
      const double inv_dg = live ? invd[k0 + i] : 1.0;
      double v = live ? bv[k0 + i] : 0.0;
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const double yj = readlane_f64(v * inv_dg, j);
        if (lane == j) v = yj;
        else if (lane > j && lane < NB) v = fma(-a[j], yj, v);
      }
      if (lane < NB) ys[lane] = live ? v : 0.0;
      if (live) bv[k0 + i] = v;
    }
    __syncthreads();
    for (int r = k0 + nb + tid; r < n; r += 256) {
      const double* Lr = Lp + tri(r, k0);
      double sacc = 0.0;
      for (int c = 0; c < nb; ++c) sacc = fma(Lr[c], ys[c], sacc);
      bv[r] -= sacc;
    }
    __syncthreads();
  }
  CHOL_STAMP(41);
  // ---- backward substitution L^T x = y
  const int last = ((n - 1) / NB) * NB;
  for (int k0 = last; k0 >= 0; k0 -= NB) {
    const int nb = min(NB, n - k0);
    if (tid < 64) {
      const int j = lane & (NB - 1);
      const bool live = (lane < NB) && (j < nb);
      double c_[NB];                      // column j of the block: L[k0+i][k0+j], i > j
#pragma unroll
      for (int i = 0; i < NB; ++i) c_[i] = (live && i > j && i < nb) ? Lp[tri(k0 + i, k0 + j)] : 0.0;
      const double inv_dg = live ? invd[k0 + j] : 1.0;
      double v = live ? bv[k0 + j] : 0.0;
#pragma unroll
      for (int i = NB - 1; i >= 0; --i) {
        const double xi = readlane_f64(v * inv_dg, i);
        if (lane == i) v = xi;
        else if (lane < i) v = fma(-c_[i], xi, v);
      }
      if (lane < NB) ys[lane] = live ? v : 0.0;
      if (live) { bv[k0 + j] = v; dx[k0 + j] = (float)v; }
    }
    __syncthreads();
    for (int c = tid; c < k0; c += 256) {
      double sacc = 0.0;
      for (int r = 0; r < nb; ++r) sacc = fma(Lp[tri(k0 + r, c)], ys[r], sacc);
      bv[c] -= sacc;
    }
    __syncthreads();
  }
  CHOL_STAMP(42);
}

}  // namespace

int gs_chol_solve_launch(double* H, double* b, int n, float lm, float ep, float* dx_out, int32_t* fail_flag,
                         int32_t* fail_count, hipStream_t st) {
  if (n <= SMALL_N) {
    const size_t lds = ((size_t)n * (n + 1) / 2 + 2 * (size_t)n) * sizeof(double);
    static bool attr_set = false;
    if (!attr_set) {
      const size_t cap = ((size_t)SMALL_N * (SMALL_N + 1) / 2 + 2 * SMALL_N) * sizeof(double);
      if (hipFuncSetAttribute((const void*)chol_small_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)cap) != hipSuccess) {
        gs_set_error("chol: cannot raise the dynamic LDS limit to %zu bytes", cap);
        return GS_ERR_LAUNCH;
      }
      attr_set = true;
    }
    chol_small_kernel<<<1, 256, lds, st>>>(H, b, n, (double)lm, (double)ep, dx_out, fail_flag, fail_count);
    GS_CHECK_LAUNCH("chol_small");
    return GS_OK;
  }
  chol_damp_kernel<<<gs_cdiv(n, 256), 256, 0, st>>>(H, n, (double)lm, (double)ep, fail_flag);
  GS_CHECK_LAUNCH("chol_damp");
  for (int k0 = 0; k0 < n; k0 += NB) {
    const int nb = (n - k0 < NB) ? (n - k0) : NB;
    const int rem = n - k0 - nb;
    const int pgrid = rem > 0 ? gs_cdiv(rem, PR) : 1;
    chol_panel_kernel<<<pgrid, 64, 0, st>>>(H, n, k0, fail_flag);
    GS_CHECK_LAUNCH("chol_panel");
    if (rem > 0) {
      const int T = gs_cdiv(rem, TT);
      chol_trail_kernel<<<T * (T + 1) / 2, 256, 0, st>>>(H, n, k0, nb);
      GS_CHECK_LAUNCH("chol_trail");
    }
  }
  chol_solve_kernel<<<1, 256, 0, st>>>(H, b, n, dx_out, fail_flag, fail_count);
  GS_CHECK_LAUNCH("chol_solve");
  return GS_OK;
}
