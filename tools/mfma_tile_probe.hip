// Issue rate of a realistic f64 MFMA register tile (4x2 tiles, distinct A/B operand registers, 4M and 3M
// accumulator sets), VGPR-form (launch_bounds(.,2)), with and without LDS fragment reads in the loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double double4_t __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int NSET, int LDSREAD>
__global__ void __launch_bounds__(512, 2) tile_kernel(double* out, int iters, unsigned long long* cyc, const double* src) {
  __shared__ double2 lds[2 * 8 * 129 * 2];
  for (int i = threadIdx.x; i < 2 * 8 * 129 * 2; i += 512) lds[i] = make_double2(src[i & 255], src[(i + 7) & 255]);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double4_t acc[NSET][4][2];
  for (int s = 0; s < NSET; ++s) for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) acc[s][i][j] = {0, 0, 0, 0};
  double2 a[4], b[2];
  for (int i = 0; i < 4; ++i) a[i] = lds[(lane >> 4) * 129 + (wave & 1) * 64 + i * 16 + (lane & 15)];
  for (int j = 0; j < 2; ++j) b[j] = lds[8 * 129 + (lane >> 4) * 129 + (wave >> 1) * 32 + j * 16 + (lane & 15)];
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (LDSREAD) {
      const int kk = (it & 1) * 4 + (lane >> 4);
      for (int i = 0; i < 4; ++i) a[i] = lds[kk * 129 + (wave & 1) * 64 + i * 16 + (lane & 15)];
      for (int j = 0; j < 2; ++j) b[j] = lds[8 * 129 + kk * 129 + (wave >> 1) * 32 + j * 16 + (lane & 15)];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        acc[0][i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i].x, b[j].x, acc[0][i][j], 0, 0, 0);
        acc[1][i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i].y, b[j].y, acc[1][i][j], 0, 0, 0);
        if (NSET == 2) {
          acc[0][i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i].y, b[j].y, acc[0][i][j], 0, 0, 1);
          acc[1][i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i].y, b[j].x, acc[1][i][j], 0, 0, 0);
        } else {
          acc[2][i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i].x + a[i].y, b[j].x + b[j].y, acc[2][i][j], 0, 0, 0);
        }
      }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int q = 0; q < NSET; ++q) for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) s += acc[q][i][j][0] + acc[q][i][j][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NSET, int LDSREAD>
int run(const char* name, int blocks, int iters, const double* src) {
  double* out; unsigned long long* cyc;
  CK(hipMalloc(&out, (size_t)blocks * 512 * 8)); CK(hipMalloc(&cyc, blocks * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  tile_kernel<NSET, LDSREAD><<<blocks, 512>>>(out, 8, cyc, src);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  tile_kernel<NSET, LDSREAD><<<blocks, 512>>>(out, iters, cyc, src);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h(blocks);
  CK(hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost));
  const int per_it = (NSET == 2 ? 32 : 24);
  double nm = (double)blocks * 8 * iters * per_it;
  printf("%-28s blocks=%d: %.3f ms, %.2f TFLOP/s executed, %.1f ticks per MFMA per SIMD (2 waves/SIMD)\n", name, blocks, ms,
         nm * 2048 / ms * 1e-9, (double)h[0] / (iters * per_it * 2.0));
  return 0;
}

int main() {
  double* src; CK(hipMalloc(&src, 256 * 8));
  std::vector<double> h(256); for (int i = 0; i < 256; ++i) h[i] = 1.0 + i * 1e-3;
  CK(hipMemcpy(src, h.data(), 256 * 8, hipMemcpyHostToDevice));
  run<2, 0>("4M tile, regs only", 256, 2000, src);
  run<2, 1>("4M tile + LDS frag reads", 256, 2000, src);
  run<3, 0>("3M tile, regs only", 256, 2000, src);
  run<3, 1>("3M tile + LDS frag reads", 256, 2000, src);
  return 0;
}
