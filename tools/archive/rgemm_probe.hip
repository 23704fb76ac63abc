// Ablation probe of the real GEMM kernel (diffquantum_amd/csrc/rgemm.hip) at d = 4096: build once per
// DQ_RGEMM_ABLATE value (tools/rgemm_probe.sh) and compare the per-launch times.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DDQ_RGEMM_ABLATE=<bits> tools/rgemm_probe.hip -o tools/_build/rgemm_probe_<bits>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../diffquantum_amd/csrc/rgemm.hip"

namespace dq {
int zgemm_pick_tile(int M, int N, int batch) { (void)batch; return (M >= 128 && N >= 128) ? 128 : 64; }
}

int main(int argc, char** argv) {
  const int d = argc > 1 ? atoi(argv[1]) : 4096;
  const size_t n = (size_t)d * d;
  std::vector<double> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = (double)((i * 2654435761u) % 1000) / 1000.0 - 0.5;
  double *A, *B, *C;
  hipMalloc(&A, n * 8); hipMalloc(&B, n * 8); hipMalloc(&C, n * 8);
  hipMemcpy(A, h.data(), n * 8, hipMemcpyHostToDevice);
  hipMemcpy(B, h.data(), n * 8, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int cfg = 1; cfg <= 3; ++cfg) {
    dq::RgemmArgs<double> g;
    g.A = A; g.B = B; g.C = C; g.M = g.N = g.K = d; g.lda = g.ldb = g.ldc = d; g.sA = g.sB = g.sC = 0; g.batch = 1;
    g.alpha = 1.0; g.beta = 0.0; g.opA = 1; g.opB = 0; g.sym = 0; g.ws = nullptr; g.ws_elems = 0;
    g.glds_ok = getenv("NO_GLDS") ? 0 : 1;
    dq::rgemm_launch<double>(g, 0, cfg, 256);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    const int reps = 10;
    for (int r = 0; r < reps; ++r) dq::rgemm_launch<double>(g, 0, cfg, 256);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("ablate %d cfg %d: %.3f ms/launch = %.1f TF/s\n", DQ_RGEMM_ABLATE, cfg, ms / reps, 2.0 * d * d * d / (ms / reps * 1e-3) / 1e12);
  }
  // symmetric-output forms on the 16-wave tile (what the real path issues): one-product and two-product launches
  double* ws; const size_t ws_elems = (size_t)256 * 4 * 128 * 128;
  hipMalloc(&ws, ws_elems * 8);
  for (int two = 0; two <= 1; ++two) {
    dq::RgemmArgs<double> g;
    g.A = A; g.B = B; g.C = C; g.M = g.N = g.K = d; g.lda = g.ldb = g.ldc = d; g.sA = g.sB = g.sC = 0; g.batch = 1;
    g.alpha = 1.0; g.beta = 0.0; g.opA = 1; g.opB = 0; g.sym = 1; g.two = two; g.ws = ws; g.ws_elems = ws_elems;
    g.glds_ok = getenv("NO_GLDS") ? 0 : 1;
    dq::rgemm_launch<double>(g, 0, 3, 256);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    const int reps = 10;
    for (int r = 0; r < reps; ++r) dq::rgemm_launch<double>(g, 0, 3, 256);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double fl = 2.0 * d * 128.0 * 128.0 * 528 * (two ? 2 : 1);
    printf("symmetric output, %s: %.3f ms/launch = %.1f TF/s executed\n", two ? "two products" : "one product", ms / reps,
           fl / (ms / reps * 1e-3) / 1e12);
  }
  return 0;
}
