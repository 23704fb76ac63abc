// Ablation probe of the complex GEMM kernel (diffquantum_amd/csrc/zgemm.hip) at d = 4096: build once per
// DQ_ZGEMM_ABLATE value (tools/zgemm_probe.sh) and compare the per-launch times of the 4M and 3M forms.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../diffquantum_amd/csrc/zgemm.hip"

int main(int argc, char** argv) {
  const int d = argc > 1 ? atoi(argv[1]) : 4096;
  const size_t n = (size_t)d * d;
  std::vector<double2> h(n);
  for (size_t i = 0; i < n; ++i) { h[i].x = (double)((i * 2654435761u) % 1000) / 1000.0 - 0.5; h[i].y = (double)((i * 40503u) % 997) / 997.0 - 0.5; }
  double2 *A, *B, *C;
  (void)hipMalloc(&A, n * 16); (void)hipMalloc(&B, n * 16); (void)hipMalloc(&C, n * 16);
  (void)hipMemcpy(A, h.data(), n * 16, hipMemcpyHostToDevice);
  (void)hipMemcpy(B, h.data(), n * 16, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int alg = 0; alg <= 1; ++alg)
    for (int op = 0; op < 4; op += 3) {
      dq::ZgemmArgs<double> g;
      g.A = A; g.B = B; g.C = C; g.M = g.N = g.K = d; g.lda = g.ldb = g.ldc = d; g.sA = g.sB = g.sC = 0; g.batch = 1;
      g.alpha = 1.0; g.beta = 0.0; g.opA = op >> 1; g.opB = op & 1;
      g.herm = 0; g.ws = nullptr; g.ws_elems = 0;
      (void)dq::zgemm_launch<double>(g, 0, 128, alg);
      (void)hipDeviceSynchronize();
      (void)hipEventRecord(e0, 0);
      const int reps = 6;
      for (int r = 0; r < reps; ++r) (void)dq::zgemm_launch<double>(g, 0, 128, alg);
      (void)hipEventRecord(e1, 0);
      (void)hipEventSynchronize(e1);
      float ms = 0;
      (void)hipEventElapsedTime(&ms, e0, e1);
      const double ex = (alg ? 6.0 : 8.0) * d * (double)d * d;
      printf("ablate %d alg %s op %d%d: %.3f ms/launch = %.1f TF/s executed\n", DQ_ZGEMM_ABLATE, alg ? "3M" : "4M", op >> 1, op & 1,
             ms / reps, ex / (ms / reps * 1e-3) / 1e12);
    }
  return 0;
}
