// Phase timing of chol_small_kernel (developer tool, not part of the product):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include -I go_slam_amd/csrc -o /tmp/chol_bench tools/chol_bench.hip && /tmp/chol_bench 150
#define CHOL_TIMING 1
#include "../go_slam_amd/csrc/chol.hip"
#include <cstdio>
#include <vector>
#include <cmath>
void gs_set_error(const char* fmt, ...) {}
int main(int argc, char** argv) {
  int n = argc > 1 ? atoi(argv[1]) : 150;
  std::vector<double> A((size_t)n * n), b(n);
  srand(1);
  std::vector<double> M((size_t)n * n);
  for (auto& v : M) v = (rand() / (double)RAND_MAX) - 0.5;
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) {
    double s = 0; for (int k = 0; k < n; ++k) s += M[i * n + k] * M[j * n + k];
    A[i * n + j] = s + (i == j ? n : 0);
  }
  for (int i = 0; i < n; ++i) b[i] = i * 0.01;
  double *dA, *db; float* dx; int32_t* flags;
  hipMalloc(&dA, A.size() * 8); hipMalloc(&db, n * 8); hipMalloc(&dx, n * 4); hipMalloc(&flags, 8);
  hipMemset(flags, 0, 8);
  for (int rep = 0; rep < 5; ++rep) {
    hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice);
    gs_chol_solve_launch(dA, db, n, 1e-4f, 0.1f, dx, flags, flags + 1, 0);
    hipDeviceSynchronize();
  }
  long long t[64];
  hipMemcpyFromSymbol(t, HIP_SYMBOL(g_chol_t), sizeof(t));
  auto us = [&](int a, int b) { return (t[b] - t[a]) / 100.0; };
  printf("load %.1f  factor %.1f  backward %.1f  total %.1f us\n", us(0,1), us(1,2), us(2,3), us(0,3));
  printf("  thread0: diag %.1f  rows %.1f  panelupd %.1f  barrier %.1f  far %.1f us\n", t[10]/100.0, t[11]/100.0, t[12]/100.0, t[13]/100.0, t[14]/100.0);
  std::vector<float> x(n); hipMemcpy(x.data(), dx, n * 4, hipMemcpyDeviceToHost);
  // residual check
  double worst = 0;
  for (int i = 0; i < n; ++i) { double s = 0; for (int j = 0; j < n; ++j) { double a = A[i*n+j]; if (i==j) a = a + (0.1 + 1e-4 * a); s += a * x[j]; } worst = fmax(worst, fabs(s - b[i])); }
  printf("max residual %.3e\n", worst);
  return 0;
}
