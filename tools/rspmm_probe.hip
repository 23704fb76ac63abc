// Probe of the sparse-generator x dense pass (out = ca Add + X V, X = n-qubit Ising diagonal + X_k flips: n + 1 nonzeros
// per row) at d = 4096: the library's one-wave-per-row kernel (realpath.hip rsp_mm_kernel) against column-slab variants whose
// working set stays in one XCD's L2.  Prints per-launch times and checks the variants against the library kernel bit for bit.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "../diffquantum_amd/csrc/realpath.hip"

#ifndef PROBE_ONLY_OLD
#include "rspmm_variants.inc"
#endif

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 12;
  const int reps = argc > 2 ? atoi(argv[2]) : 20;
  const int d = 1 << n, K = n;
  const size_t nn = (size_t)d * d;
  // CSR: row i = (i, diag), (i ^ (1 << b), control b)
  std::vector<long> rowptr(d + 1);
  std::vector<int> col((size_t)d * (n + 1)), kid((size_t)d * (n + 1));
  std::vector<double> val((size_t)d * (n + 1)), u(K);
  for (int k = 0; k < K; ++k) u[k] = 0.3 + 0.1 * k;
  for (int i = 0; i < d; ++i) {
    rowptr[i] = (long)i * (n + 1);
    size_t e = rowptr[i];
    col[e] = i; kid[e] = -1; val[e] = sin(0.37 * i) * 3.0; ++e;
    for (int b = 0; b < n; ++b, ++e) { col[e] = i ^ (1 << (n - 1 - b)); kid[e] = b; val[e] = 1.0; }
  }
  rowptr[d] = (long)d * (n + 1);
  std::vector<double> hv(nn), ha(nn);
  for (size_t i = 0; i < nn; ++i) { hv[i] = (double)((i * 2654435761u) % 1000) / 1000.0 - 0.5; ha[i] = (double)((i * 40503u) % 997) / 997.0 - 0.5; }
  double *V, *Add, *O0, *O1, *du, *dval; long* drp; int *dcol, *dkid;
  (void)hipMalloc(&V, nn * 8); (void)hipMalloc(&Add, nn * 8); (void)hipMalloc(&O0, nn * 8); (void)hipMalloc(&O1, nn * 8);
  (void)hipMalloc(&du, K * 8); (void)hipMalloc(&dval, val.size() * 8); (void)hipMalloc(&drp, (d + 1) * 8);
  (void)hipMalloc(&dcol, col.size() * 4); (void)hipMalloc(&dkid, kid.size() * 4);
  (void)hipMemcpy(V, hv.data(), nn * 8, hipMemcpyHostToDevice); (void)hipMemcpy(Add, ha.data(), nn * 8, hipMemcpyHostToDevice);
  (void)hipMemcpy(du, u.data(), K * 8, hipMemcpyHostToDevice); (void)hipMemcpy(dval, val.data(), val.size() * 8, hipMemcpyHostToDevice);
  (void)hipMemcpy(drp, rowptr.data(), (d + 1) * 8, hipMemcpyHostToDevice);
  (void)hipMemcpy(dcol, col.data(), col.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dkid, kid.data(), kid.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const double dt = 1e-3, ca = 0.25;
  auto timeit = [&](const char* name, auto&& launch, double* out) {
    launch(out); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) launch(out);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    hipError_t err = hipGetLastError();
    printf("%-40s %.4f ms/launch  (%.0f GB/s algorithmic: 3 x 8 d^2 B)%s\n", name, ms / reps, 3.0 * nn * 8 / (ms / reps * 1e-3) / 1e9,
           err == hipSuccess ? "" : hipGetErrorString(err));
  };
  timeit("library rsp_mm_kernel (one wave per row)", [&](double* o) {
    (void)dq::rsp_mm_launch<double>(o, V, Add, ca, (long)nn, d, drp, dcol, dkid, dval, du, K, 1, dt, 0, 0); }, O0);
  timeit("library rsp_mm_slab_kernel", [&](double* o) {
    (void)dq::rsp_mm_launch<double>(o, V, Add, ca, (long)nn, d, drp, dcol, dkid, dval, du, K, 1, dt, 1, 0); }, O1);
  {  // the fused dot pass: one wave per row (part = null) against the XCD-slab form (DQ_RSPD_NQG loads in flight per group)
    const long nnz = rowptr[d];
    double *Xm, *nz0, *nz1, *part;
    (void)hipMalloc(&Xm, nn * 8); (void)hipMalloc(&nz0, nnz * 8); (void)hipMalloc(&nz1, nnz * 8); (void)hipMalloc(&part, nnz * 8 * 8);
    (void)hipMemcpy(Xm, ha.data(), nn * 8, hipMemcpyHostToDevice);
    timeit("library rsp_mm_dot_kernel (one wave per row)", [&](double* o) {
      (void)dq::rsp_mm_dot_launch<double>(o, V, Add, ca, Xm, (long)nn, d, drp, dcol, dkid, dval, du, K, 1, dt, nnz, nz0, 0, nullptr, 0); }, O0);
    timeit("library rsp_mm_dot_slab_kernel", [&](double* o) {
      (void)dq::rsp_mm_dot_launch<double>(o, V, Add, ca, Xm, (long)nn, d, drp, dcol, dkid, dval, du, K, 1, dt, nnz, nz1, 0, part, 0); }, O1);
    std::vector<double> a(nnz), b(nnz);
    (void)hipMemcpy(a.data(), nz0, nnz * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(b.data(), nz1, nnz * 8, hipMemcpyDeviceToHost);
    double m = 0, sc = 0;
    for (long e = 0; e < nnz; ++e) { m = fmax(m, fabs(a[e] - b[e])); sc = fmax(sc, fabs(a[e])); }
    printf("    dot sums: max|slab - row| = %.3e (max %.3e)\n", m, sc);
    timeit("library rsp_mm_slab_kernel again (reference for the pair)", [&](double* o) {
      (void)dq::rsp_mm_launch<double>(o, V, Add, ca, (long)nn, d, drp, dcol, dkid, dval, du, K, 1, dt, 1, 0); }, O1);
    timeit("library rsp_mm_kernel -> O0 again", [&](double* o) {
      (void)dq::rsp_mm_launch<double>(o, V, Add, ca, (long)nn, d, drp, dcol, dkid, dval, du, K, 1, dt, 0, 0); }, O0);
  }
#ifndef PROBE_ONLY_OLD
  {  // pitch experiment: the same pass with a row pitch of d + PAD doubles (is the power-of-two pitch costing L2 sets?)
    for (int pad : {0, 16, 64, 272}) {
      const long ld = d + pad;
      double *Vp, *Ap, *Op;
      (void)hipMalloc(&Vp, (size_t)ld * d * 8); (void)hipMalloc(&Ap, (size_t)ld * d * 8); (void)hipMalloc(&Op, (size_t)ld * d * 8);
      (void)hipMemcpy2D(Vp, ld * 8, V, (size_t)d * 8, (size_t)d * 8, d, hipMemcpyDeviceToDevice);
      (void)hipMemcpy2D(Ap, ld * 8, Add, (size_t)d * 8, (size_t)d * 8, d, hipMemcpyDeviceToDevice);
      char name[96];
      snprintf(name, sizeof name, "row x XCD slab, row pitch d + %d", pad);
      timeit(name, [&](double* o) { probe::rowrange_ld_launch<13>(o, Vp, Ap, ca, ld, d, drp, dcol, dkid, dval, du, K, dt); }, Op);
      (void)hipFree(Vp); (void)hipFree(Ap); (void)hipFree(Op);
    }
  }
  std::vector<double> h0(nn), h1(nn);
  (void)hipMemcpy(h0.data(), O0, nn * 8, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int v = 0; v < probe_nvariants(); ++v) {
    (void)hipMemset(O1, 0xff, nn * 8);
    timeit(probe_variant_name(v), [&](double* o) { probe_variant_launch(v, o, V, Add, ca, (long)nn, d, drp, dcol, dkid, dval, du, K, dt); }, O1);
    (void)hipMemcpy(h1.data(), O1, nn * 8, hipMemcpyDeviceToHost);
    double m = 0; size_t nd = 0;
    for (size_t i = 0; i < nn; ++i) { const double df = fabs(h1[i] - h0[i]); if (!(df == 0)) ++nd; if (df > m || df != df) m = df; }
    printf("    max|variant - library| = %.3e (%zu entries differ)\n", m, nd);
    bad += !(m < 1e-12);
  }
  printf(bad ? "PROBE FAILED\n" : "PROBE OK\n");
  return bad;
#else
  return 0;
#endif
}
