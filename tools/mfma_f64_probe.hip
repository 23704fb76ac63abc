// Probe for v_mfma_f64_16x16x4_f64 on gfx950: (1) verifies the A/B/C lane maps the
// ZGEMM kernel relies on, (2) decodes the blgp negate bits, (3) measures the
// back-to-back issue rate (cycles per MFMA, chip TFLOP/s) that prices the fp64
// roofline. Stand-alone: hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

typedef double double4_t __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

template <int BLGP>
__global__ void layout_kernel(const double* A, const double* B, double* C) {
  // A: 16x4 row-major, B: 4x16 row-major, C: 16x16 row-major
  int l = threadIdx.x;
  double a = A[(l & 15) * 4 + (l >> 4)];
  double b = B[(l >> 4) * 16 + (l & 15)];
  double4_t c = {1.0, 2.0, 3.0, 4.0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, BLGP);
  for (int j = 0; j < 4; ++j) C[((l >> 4) + 4 * j) * 16 + (l & 15)] = c[j];
}

template <int NACC>
__global__ void rate_kernel(double* out, int iters, unsigned long long* cyc) {
  double a = 1.0 + threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-3;
  double4_t acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = {0.0, 0.0, 0.0, 0.0};
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i)
      acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC>
__global__ void rate_kernel_varied(double* out, int iters, unsigned long long* cyc) {
  // distinct A/B registers per accumulator (DGEMM-like register tile 4x(NACC/4))
  double a[4], b[NACC / 4 > 0 ? NACC / 4 : 1];
  for (int i = 0; i < 4; ++i) a[i] = 1.0 + (threadIdx.x + i) * 1e-3;
  for (int i = 0; i < NACC / 4; ++i) b[i] = 1.0 - (threadIdx.x + i) * 1e-3;
  double4_t acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = {0.0, 0.0, 0.0, 0.0};
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i)
      acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i & 3], b[i >> 2], acc[i], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC>
__global__ void rate_kernel_4x4(double* out, int iters, unsigned long long* cyc) {
  double a = 1.0 + threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-3;
  double acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i)
      acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC>
__global__ void rate_kernel_valu(double* out, int iters, unsigned long long* cyc) {
  double a = 1.0 + threadIdx.x * 1e-9, b = 1e-9 * threadIdx.x;
  double acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = i;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_fma(acc[i], a, b);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int BLGP>
static void run_layout(const char* name) {
  std::vector<double> A(64), B(64), C(256), R(256);
  for (int i = 0; i < 64; ++i) { A[i] = (i * 7 % 11) - 5; B[i] = (i * 5 % 13) - 6; }
  double *dA, *dB, *dC;
  CK(hipMalloc(&dA, 64 * 8)); CK(hipMalloc(&dB, 64 * 8)); CK(hipMalloc(&dC, 256 * 8));
  CK(hipMemcpy(dA, A.data(), 64 * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(dB, B.data(), 64 * 8, hipMemcpyHostToDevice));
  layout_kernel<BLGP><<<1, 64>>>(dA, dB, dC);
  CK(hipMemcpy(C.data(), dC, 256 * 8, hipMemcpyDeviceToHost));
  // candidates: c0 +/- A*B with sign patterns
  const int sa[4] = {1, -1, 1, -1}, sc[4] = {1, 1, -1, -1};
  for (int v = 0; v < 4; ++v) {
    double err = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
      double s = 0; for (int k = 0; k < 4; ++k) s += A[i * 4 + k] * B[k * 16 + j];
      double c0 = (double)(i / 4 + 1);  // row = (l>>4) + 4*j -> j = row/4... see note below
      // c init was {1,2,3,4} per reg j, row = (l>>4)+4*j, so c0 = row/4 + 1
      double r = sc[v] * c0 + sa[v] * s;
      err = fmax(err, fabs(r - C[i * 16 + j]));
    }
    printf("layout blgp=%d (%s): hypothesis D = %+d*C %+d*A*B -> max err %.3g\n", BLGP, name, sc[v], sa[v], err);
  }
  CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC));
}

template <int NACC>
static void run_rate(int blocks, int threads, int iters) {
  double* out; unsigned long long* cyc;
  CK(hipMalloc(&out, (size_t)blocks * threads * 8)); CK(hipMalloc(&cyc, blocks * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  rate_kernel<NACC><<<blocks, threads>>>(out, 16, cyc);  // warm
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  rate_kernel<NACC><<<blocks, threads>>>(out, iters, cyc);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h(blocks);
  CK(hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost));
  double waves = (double)blocks * threads / 64;
  double nmfma = waves * iters * NACC;
  double flops = nmfma * 2.0 * 16 * 16 * 4;
  // s_memtime ticks at 100 MHz constant clock on gfx9; report both
  printf("rate NACC=%d blocks=%d threads=%d iters=%d: %.3f ms, %.2f TFLOP/s, memtime ticks/blk0=%llu (%.1f ticks per MFMA per wave)\n",
         NACC, blocks, threads, iters, ms, flops / ms * 1e-9, h[0], (double)h[0] / (iters * NACC));
  CK(hipFree(out)); CK(hipFree(cyc));
}

template <int NACC, int KIND>
static void run_rate2(int blocks, int threads, int iters) {
  double* out; unsigned long long* cyc;
  CK(hipMalloc(&out, (size_t)blocks * threads * 8)); CK(hipMalloc(&cyc, blocks * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto launch = [&](int it) {
    if (KIND == 0) rate_kernel_varied<NACC><<<blocks, threads>>>(out, it, cyc);
    if (KIND == 1) rate_kernel_4x4<NACC><<<blocks, threads>>>(out, it, cyc);
    if (KIND == 2) rate_kernel_valu<NACC><<<blocks, threads>>>(out, it, cyc);
  };
  launch(16); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); launch(iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  CK(hipGetLastError());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h(blocks);
  CK(hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost));
  double waves = (double)blocks * threads / 64;
  double n = waves * iters * NACC;
  double fl = KIND == 0 ? 2048.0 : (KIND == 1 ? 512.0 : 128.0);
  const char* nm = KIND == 0 ? "mfma16x16x4-varied" : (KIND == 1 ? "mfma4x4x4_4b" : "v_fma_f64");
  printf("rate2 %s NACC=%d blocks=%d threads=%d: %.3f ms, %.2f TFLOP/s, ticks/instr/wave=%.1f\n",
         nm, NACC, blocks, threads, ms, n * fl / ms * 1e-9, (double)h[0] / (iters * (double)NACC));
  CK(hipFree(out)); CK(hipFree(cyc));
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device %s CUs=%d clock=%d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
  run_layout<0>("none");
  run_layout<1>("bit0");
  run_layout<2>("bit1");
  run_layout<4>("bit2");
  int cus = p.multiProcessorCount;
  run_rate<1>(cus, 256, 20000);
  run_rate<2>(cus, 256, 10000);
  run_rate<4>(cus, 256, 5000);
  run_rate<8>(cus, 256, 4000);
  run_rate<8>(cus, 512, 4000);
  run_rate<8>(cus * 2, 256, 4000);
  run_rate<8>(cus * 4, 256, 4000);
  run_rate<8>(cus * 8, 256, 2000);
  run_rate<4>(cus * 8, 256, 4000);
  run_rate<16>(cus, 256, 2000);
  run_rate<16>(cus, 512, 2000);
  run_rate<16>(cus, 1024, 2000);
  run_rate2<16, 0>(cus, 256, 2000);
  run_rate2<16, 0>(cus, 512, 2000);
  run_rate2<16, 0>(cus, 1024, 2000);
  run_rate2<32, 0>(cus, 256, 1000);
  run_rate2<32, 0>(cus, 512, 1000);
  run_rate2<8, 1>(cus, 256, 8000);
  run_rate2<8, 1>(cus, 1024, 8000);
  run_rate2<8, 2>(cus, 256, 20000);
  run_rate2<8, 2>(cus, 512, 20000);
  run_rate2<8, 2>(cus, 1024, 20000);
  run_rate2<8, 2>(cus * 2, 1024, 20000);
  return 0;
}
