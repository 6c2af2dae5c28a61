// Correctness + timing of the device Cholesky solves (developer tool, not part of the product):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include -I go_slam_amd/csrc -o /tmp/chol_bench tools/chol_bench.hip && /tmp/chol_bench 150 294 ...
// For every n: max residual |(A + D) x - b| of the single-launch path (chol_small_kernel for n <= 192, chol_mid_kernel
// above) and of the multi-kernel blocked path, their agreement, and event-timed microseconds per solve; for n <= 192
// also chol_small_kernel's phase stamps.
#define CHOL_TIMING 1
int g_chol_force_blocked = 0;
int g_chol_two_launches = 0;        // 1: the multi-kernel path in its round-5 form (panel launch, then trailing-update launch)
#include "../go_slam_amd/csrc/chol.hip"
#include <cstdio>
#include <vector>
#include <cmath>
void gs_set_error(const char* fmt, ...) {}
thread_local int gs_timing_on = 0;
void gs_timing_mark(const char*) {}

static double residual(const std::vector<double>& A, const std::vector<double>& b, const std::vector<float>& x, int n) {
  double worst = 0;
  for (int i = 0; i < n; ++i) {
    double s = 0;
    for (int j = 0; j < n; ++j) { double a = A[(size_t)i * n + j]; if (i == j) a = a + (0.1 + 1e-4 * a); s += a * x[j]; }
    worst = fmax(worst, fabs(s - b[i]));
  }
  return worst;
}

int main(int argc, char** argv) {
  for (int ai = 1; ai < (argc > 1 ? argc : 2); ++ai) {
    const int n = argc > 1 ? atoi(argv[ai]) : 150;
    std::vector<double> A((size_t)n * n), b(n), M((size_t)n * n);
    srand(1);
    for (auto& v : M) v = (rand() / (double)RAND_MAX) - 0.5;
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) {
      double s = 0; for (int k = 0; k < n; ++k) s += M[(size_t)i * n + k] * M[(size_t)j * n + k];
      A[(size_t)i * n + j] = s + (i == j ? n : 0);
    }
    for (int i = 0; i < n; ++i) b[i] = i * 0.01 - 0.3;
    double *dA, *db; float* dx; int32_t* flags;
    hipMalloc(&dA, A.size() * 8); hipMalloc(&db, n * 8); hipMalloc(&dx, n * 4); hipMalloc(&flags, 64);
    hipMemset(flags, 0, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<float> xs[3];
    float us[3] = {0, 0, 0};
    for (int mode = 0; mode < 3; ++mode) {
      g_chol_force_blocked = mode;
      float best = 1e9f;
      for (int rep = 0; rep < 6; ++rep) {
        hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
        hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        gs_chol_solve_launch(dA, db, n, 1e-4f, 0.1f, dx, flags, flags + 1, flags + 8, 0);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
      }
      us[mode] = best * 1e3f;
      xs[mode].resize(n);
      hipMemcpy(xs[mode].data(), dx, n * 4, hipMemcpyDeviceToHost);
    }
    int32_t fl[2]; hipMemcpy(fl, flags, 8, hipMemcpyDeviceToHost);
    double dmax = 0, xmax = 0;
    for (int i = 0; i < n; ++i) { dmax = fmax(dmax, fabs((double)xs[0][i] - xs[1][i])); xmax = fmax(xmax, fabs((double)xs[0][i])); }
    double dmax2 = 0;
    for (int i = 0; i < n; ++i) dmax2 = fmax(dmax2, fabs((double)xs[2][i] - xs[1][i]));
    printf("n %4d: product dispatch %7.1f us (residual %.3e) | multi-kernel %7.1f us (residual %.3e) | persistent (%d groups) %7.1f us (residual %.3e) | max |x - x_multi| %.3e / %.3e of %.3e, fail flag %d count %d\n",
           n, us[0], residual(A, b, xs[0], n), us[1], residual(A, b, xs[1], n), g_chol_coop_groups, us[2], residual(A, b, xs[2], n), dmax, dmax2, xmax, fl[0], fl[1]);
    {   // the multi-kernel path in its two-launch form (round 5) against the one-launch-per-panel form that ships
      g_chol_force_blocked = 1;
      g_chol_two_launches = 1;
      float best = 1e9f;
      for (int rep = 0; rep < 6; ++rep) {
        hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
        hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        gs_chol_solve_launch(dA, db, n, 1e-4f, 0.1f, dx, flags, flags + 1, flags + 8, 0);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
      }
      g_chol_two_launches = 0;
      std::vector<float> x(n); hipMemcpy(x.data(), dx, n * 4, hipMemcpyDeviceToHost);
      double d2 = 0;
      for (int i = 0; i < n; ++i) d2 = fmax(d2, fabs((double)x[i] - xs[1][i]));
      printf("        multi-kernel, two launches per panel (round 5): %7.1f us (residual %.3e), max |x - x_one_launch| %.3e\n",
             best * 1e3f, residual(A, b, x, n), d2);
    }
    if (n > 450) {                                    // where the one-launch form's panel wave spends a step (workgroup 0)
      long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      hipMemcpyToSymbol(HIP_SYMBOL(g_step_t), z, sizeof(z));
      g_chol_force_blocked = 1;
      hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
      hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice);
      gs_chol_solve_launch(dA, db, n, 1e-4f, 0.1f, dx, flags, flags + 1, flags + 8, 0);
      hipDeviceSynchronize();
      hipMemcpyFromSymbol(z, HIP_SYMBOL(g_step_t), sizeof(z));
      const double steps = (n - 1) / 32;
      printf("        chol_step panel wave, us per step: loads + staging %.2f | pending updates %.2f | diagonal factor %.2f | row solve %.2f | stores issued %.2f\n",
             z[0] / 100.0 / steps, z[1] / 100.0 / steps, z[2] / 100.0 / steps, z[3] / 100.0 / steps, z[4] / 100.0 / steps);
    }
    if (n > 450) {                                    // the persistent path against its workgroup count
      g_chol_force_blocked = 2;
      const int keep = g_chol_coop_groups;
      for (int G : {16, 32, 64, 128, 256}) {
        g_chol_coop_groups = G;
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
          hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
          hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice);
          hipDeviceSynchronize();
          hipEventRecord(e0);
          gs_chol_solve_launch(dA, db, n, 1e-4f, 0.1f, dx, flags, flags + 1, flags + 8, 0);
          hipEventRecord(e1);
          hipEventSynchronize(e1);
          float ms; hipEventElapsedTime(&ms, e0, e1);
          if (rep > 0 && ms < best) best = ms;
        }
        std::vector<float> x(n); hipMemcpy(x.data(), dx, n * 4, hipMemcpyDeviceToHost);
        long long t[64];
        hipMemcpyFromSymbol(t, HIP_SYMBOL(g_chol_t), sizeof(t));
        printf("        persistent, %3d workgroups: %7.1f us (residual %.3e) | workgroup 0: A-phase %.1f  barrier %.1f  B-phase %.1f | back: misc %.1f triangle %.1f update %.1f barrier %.1f us\n",
               G, best * 1e3f, residual(A, b, x, n), t[50] / 100.0, t[51] / 100.0, t[52] / 100.0, t[53] / 100.0, t[54] / 100.0, t[55] / 100.0, t[56] / 100.0);
      }
      g_chol_coop_groups = keep;
    }
    if (n <= 192) {
      g_chol_force_blocked = 0;
      hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
      hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice);
      gs_chol_solve_launch(dA, db, n, 1e-4f, 0.1f, dx, flags, flags + 1, flags + 8, 0);
      hipDeviceSynchronize();
      long long t[64];
      hipMemcpyFromSymbol(t, HIP_SYMBOL(g_chol_t), sizeof(t));
      auto usf = [&](int a, int c) { return (t[c] - t[a]) / 100.0; };
      printf("        chol_small phases: load %.1f  factor %.1f  backward %.1f  total %.1f us; thread 0: diag %.1f rows %.1f panelupd %.1f barrier %.1f far %.1f\n",
             usf(0, 1), usf(1, 2), usf(2, 3), usf(0, 3), t[10] / 100.0, t[11] / 100.0, t[12] / 100.0, t[13] / 100.0, t[14] / 100.0);
    }
    if (n > 192 && n <= 312) {
      g_chol_force_blocked = 0;
      hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
      hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice);
      gs_chol_solve_launch(dA, db, n, 1e-4f, 0.1f, dx, flags, flags + 1, flags + 8, 0);
      hipDeviceSynchronize();
      long long t[64];
      hipMemcpyFromSymbol(t, HIP_SYMBOL(g_chol_t), sizeof(t));
      auto usf = [&](int a, int c) { return (t[c] - t[a]) / 100.0; };
      printf("        chol_mid phases: damp %.1f |", usf(0, 20) < 0 ? 0.0 : 0.0);
      int prev = 20;
      for (int sgi = 0; sgi < 4 && t[21 + 4 * sgi] > t[20]; ++sgi) {
        const int q = 21 + 4 * sgi;
        printf(" stage %d: load %.1f factor %.1f global-update %.1f write-back %.1f |", sgi, usf(prev, q), usf(q, q + 1), usf(q + 1, q + 2), usf(q + 2, q + 3));
        prev = q + 3;
      }
      printf(" tail: load %.1f factor %.1f backward %.1f | heads: L21^T x %.1f stages %.1f | total %.1f us\n", usf(prev, 40), usf(40, 41), usf(41, 42), usf(42, 43), usf(43, 44), usf(20, 44));
    }
    // an indefinite matrix must give dx = 0 and raise the flag on both paths
    std::vector<double> Aneg = A;
    Aneg[(size_t)(n / 2) * n + n / 2] = -1e6;
    for (int mode = 0; mode < 3; ++mode) {
      g_chol_force_blocked = mode;
      hipMemcpy(dA, Aneg.data(), A.size() * 8, hipMemcpyHostToDevice);
      hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice);
      gs_chol_solve_launch(dA, db, n, 1e-4f, 0.1f, dx, flags, flags + 1, flags + 8, 0);
      hipDeviceSynchronize();
      std::vector<float> x(n); hipMemcpy(x.data(), dx, n * 4, hipMemcpyDeviceToHost);
      hipMemcpy(fl, flags, 8, hipMemcpyDeviceToHost);
      double nz = 0; for (float v : x) nz = fmax(nz, fabs((double)v));
      printf("        indefinite (%s): max |dx| %.1e, fail flag %d\n", mode == 0 ? "product dispatch" : mode == 1 ? "multi-kernel" : "persistent", nz, fl[0]);
    }
    hipFree(dA); hipFree(db); hipFree(dx); hipFree(flags);
  }
  return 0;
}
