// Probe of the linear-form complex GEMM (diffquantum_amd/csrc/zgemm_lin.hip) against the general kernel (zgemm.hip) at
// d = 4096: same product both ways (A^T B, A^H B, anti-Hermitian-output A^H B), max |difference|, per-launch times.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "../diffquantum_amd/csrc/zgemm.hip"
#include "../diffquantum_amd/csrc/zgemm_lin.hip"

static double maxdiff(const double2* a, const double2* b, size_t n, std::vector<double2>& ha, std::vector<double2>& hb) {
  (void)hipMemcpy(ha.data(), a, n * 16, hipMemcpyDeviceToHost);
  (void)hipMemcpy(hb.data(), b, n * 16, hipMemcpyDeviceToHost);
  double m = 0, s = 0;
  for (size_t i = 0; i < n; ++i) { m = fmax(m, fmax(fabs(ha[i].x - hb[i].x), fabs(ha[i].y - hb[i].y))); s = fmax(s, fabs(hb[i].x)); }
  printf("   max|new - old| = %.3e (max |old| = %.3e)\n", m, s);
  return m;
}

int main(int argc, char** argv) {
  const int d = argc > 1 ? atoi(argv[1]) : 4096;
  const int reps = argc > 2 ? atoi(argv[2]) : 6;
  const size_t n = (size_t)d * d;
  std::vector<double2> h(n), ha(n), hb(n);
  for (size_t i = 0; i < n; ++i) { h[i].x = (double)((i * 2654435761u) % 1000) / 1000.0 - 0.5; h[i].y = (double)((i * 40503u) % 997) / 997.0 - 0.5; }
  double2 *A, *B, *C0, *C1, *ws;
  (void)hipMalloc(&A, n * 16); (void)hipMalloc(&B, n * 16); (void)hipMalloc(&C0, n * 16); (void)hipMalloc(&C1, n * 16);
  const size_t ws_elems = (size_t)256 * 2 * 128 * 128;
  (void)hipMalloc(&ws, ws_elems * 16);
  (void)hipMemcpy(A, h.data(), n * 16, hipMemcpyHostToDevice);
  for (size_t i = 0; i < n; ++i) { h[i].x = (double)((i * 7919u) % 991) / 991.0 - 0.5; h[i].y = (double)((i * 104729u) % 983) / 983.0 - 0.5; }
  (void)hipMemcpy(B, h.data(), n * 16, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  int bad = 0;
  for (int cfg = 0; cfg < 3; ++cfg) {  // 0: A^T B (old kernel has no plain transpose: compare with conj(A)^H... see below), 1: A^H B, 2: herm A^H B
    dq::ZgemmArgs<double> g;
    g.A = A; g.B = B; g.M = g.N = g.K = d; g.lda = g.ldb = g.ldc = d; g.sA = g.sB = g.sC = 0; g.batch = 1;
    g.alpha = cfg == 1 ? -1.0 : 1.0; g.beta = 0.0; g.opA = 1; g.opB = 0;
    g.herm = cfg == 2 ? -1 : 0; g.ws = g.herm ? ws : nullptr; g.ws_elems = g.herm ? ws_elems : 0;
    const int conj = cfg != 0;
    float ms_old = 0, ms_new = 0;
    if (cfg != 0) {
      g.C = C0;
      (void)dq::zgemm_launch<double>(g, 0, 128, 1);
      (void)hipDeviceSynchronize();
      (void)hipEventRecord(e0, 0);
      for (int r = 0; r < reps; ++r) (void)dq::zgemm_launch<double>(g, 0, 128, 1);
      (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
      (void)hipEventElapsedTime(&ms_old, e0, e1);
    }
    g.C = C1;
    hipError_t err = dq::zgemm_lin_launch(g, conj, 0, 256);
    hipError_t e2 = hipDeviceSynchronize();
    if (err != hipSuccess || e2 != hipSuccess) { printf("launch failed: %s %s\n", hipGetErrorString(err), hipGetErrorString(e2)); return 1; }
    (void)hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) (void)dq::zgemm_lin_launch(g, conj, 0, 256);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms_new, e0, e1);
    const double ex = 6.0 * d * (double)d * d * (cfg == 2 ? 0.515625 : 1.0);
    printf("cfg %d (%s): old %.3f ms, new %.3f ms/launch = %.1f TF/s executed (old %.1f)\n", cfg,
           cfg == 0 ? "A^T B" : cfg == 1 ? "-A^H B" : "herm A^H B", ms_old / reps, ms_new / reps,
           ex / (ms_new / reps * 1e-3) / 1e12, ms_old > 0 ? ex / (ms_old / reps * 1e-3) / 1e12 : 0.0);
    if (cfg != 0) bad += maxdiff(C1, C0, n, ha, hb) > 1e-9;
    else {
      // A^T B = conj(conj(A)^H B): check a few entries on the host instead
      (void)hipMemcpy(ha.data(), C1, n * 16, hipMemcpyDeviceToHost);
      std::vector<double2> hA(n), hB(n);
      (void)hipMemcpy(hA.data(), A, n * 16, hipMemcpyDeviceToHost);
      (void)hipMemcpy(hB.data(), B, n * 16, hipMemcpyDeviceToHost);
      double m = 0;
      for (int t = 0; t < 64; ++t) {
        const int i = (t * 977 + 13) % d, j = (t * 3571 + 101) % d;
        double sr = 0, si = 0;
        for (int k = 0; k < d; ++k) {
          const double2 a = hA[(size_t)k * d + i], b = hB[(size_t)k * d + j];
          sr += a.x * b.x - a.y * b.y; si += a.x * b.y + a.y * b.x;
        }
        m = fmax(m, fmax(fabs(sr - ha[(size_t)i * d + j].x), fabs(si - ha[(size_t)i * d + j].y)));
      }
      printf("   max|new - host| over 64 entries = %.3e\n", m);
      bad += m > 1e-9;
    }
  }
  printf(bad ? "PROBE FAILED\n" : "PROBE OK\n");
  return bad;
}
