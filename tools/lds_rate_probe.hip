// LDS read-rate probe for the access pattern of the state-column sweeps (diffquantum_amd/csrc/statecol.hip): one 512-thread
// workgroup per CU on a 64 KB image, thread t reads rows r * 512 + (t ^ x) with ds_read_b128 for nine XOR offsets x -- with and
// without the sweeps' arithmetic (2 or 6 f64 FMAs per read), as bursts of 8 reads drained before the next burst or as a ring that
// keeps DEPTH reads in flight.  Prints bytes per clock and CU (shader clock by s_memrealtime is not available: wall time x 2.4 GHz
// and the kernel's own clock64 count both printed).  Not part of the library.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/lds_rate_probe.hip -o tools/_build/lds_rate_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef double d2 __attribute__((ext_vector_type(2)));

template <int OFF>
__device__ __forceinline__ void rd_first(d2& q, unsigned a) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q) : "v"(a), "n"(OFF)); }
template <int OFF>
__device__ __forceinline__ void rd_next(d2& q, unsigned a) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "+v"(q) : "v"(a), "n"(OFF)); }
template <int N>
__device__ __forceinline__ void wt(d2& q) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(q) : "n"(N)); }

template <int R0, int R1, class F>
__device__ __forceinline__ void rows(F&& f) {
  if constexpr (R0 < R1) {
    f(std::integral_constant<int, R0>{});
    rows<R0 + 1, R1>(f);
  }
}

// MODE 0: burst of 8 reads, counted waits, FMAS fmas per read, next burst after the last use
// MODE 1: ring with DEPTH reads in flight
template <int MODE, int FMAS, int DEPTH, int NT>
__global__ void __launch_bounds__(NT) probe(double* out, int iters, unsigned long long* cyc) {
  extern __shared__ __align__(16) unsigned char raw[];
  constexpr int RPT = 8;
  double2* img = (double2*)raw;
  const int t = threadIdx.x;
  for (int r = 0; r < RPT; ++r) img[r * NT + t] = make_double2(1.0 + t * 1e-6, 2.0 - r * 1e-6);
  __syncthreads();
  const unsigned base = (unsigned)(unsigned long long)(const __attribute__((address_space(3))) void*)img;
  double acc[2 * RPT];
#pragma unroll
  for (int i = 0; i < 2 * RPT; ++i) acc[i] = 0.0;
  const double w = 1e-9 * (t + 1);
  d2 v[RPT];
  const long long t0 = clock64();
  if (MODE == 0) {
#pragma clang loop unroll(disable)
    for (int it = 0; it < iters; ++it) {
#pragma clang loop unroll(disable)
      for (int k = 0; k < 9; ++k) {
        const unsigned a = base + (unsigned)((t ^ ((1 << k) & (NT - 1) ? (1 << k) : 1)) << 4);
        rows<0, RPT>([&](auto rc) { constexpr int r = decltype(rc)::value; rd_first<r * NT * 16>(v[r], a); });
        rows<0, RPT>([&](auto rc) {
          constexpr int r = decltype(rc)::value;
          wt<RPT - 1 - r>(v[r]);
          if (FMAS >= 2) {
            acc[2 * r] = __builtin_fma(w, v[r].y, acc[2 * r]);
            acc[2 * r + 1] = __builtin_fma(-w, v[r].x, acc[2 * r + 1]);
          }
          if (FMAS >= 6) {
            acc[2 * r] = __builtin_fma(v[r].x, acc[2 * r + 1], acc[2 * r]);
            acc[2 * r + 1] = __builtin_fma(v[r].y, w, acc[2 * r + 1]);
            acc[2 * r] = __builtin_fma(v[r].y, w, acc[2 * r]);
            acc[2 * r + 1] = __builtin_fma(v[r].x, w, acc[2 * r + 1]);
          }
          if (FMAS == 0) asm volatile("" : "+v"(v[r]));
        });
      }
    }
  } else {
    unsigned a = base + (unsigned)((t ^ 1) << 4);
    rows<0, DEPTH>([&](auto rc) { constexpr int r = decltype(rc)::value; rd_first<r * NT * 16>(v[r], a); });
    rows<DEPTH, RPT>([&](auto rc) { v[decltype(rc)::value] = 0.0; });
#pragma clang loop unroll(disable)
    for (int it = 0; it < iters * 9; ++it) {
      const int k = (it + 1) % 9;
      const unsigned an = base + (unsigned)((t ^ ((1 << k) & (NT - 1) ? (1 << k) : 1)) << 4);
      rows<0, RPT>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        wt<DEPTH - 1>(v[r]);
        if (FMAS >= 2) {
          acc[2 * r] = __builtin_fma(w, v[r].y, acc[2 * r]);
          acc[2 * r + 1] = __builtin_fma(-w, v[r].x, acc[2 * r + 1]);
        }
        if (FMAS >= 6) {
          acc[2 * r] = __builtin_fma(v[r].x, acc[2 * r + 1], acc[2 * r]);
          acc[2 * r + 1] = __builtin_fma(v[r].y, w, acc[2 * r + 1]);
          acc[2 * r] = __builtin_fma(v[r].y, w, acc[2 * r]);
          acc[2 * r + 1] = __builtin_fma(v[r].x, w, acc[2 * r + 1]);
        }
        constexpr int kk = r + DEPTH;
        if constexpr (kk < RPT) rd_next<kk * NT * 16>(v[kk], a);
        else rd_next<(kk - RPT) * NT * 16>(v[kk - RPT], an);
      });
      a = an;
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
  }
  const long long t1 = clock64();
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < 2 * RPT; ++i) s += acc[i];
#pragma unroll
  for (int r = 0; r < RPT; ++r) s += v[r].x;
  out[(size_t)blockIdx.x * NT + t] = s;
  if (t == 0) atomicAdd(cyc, (unsigned long long)(t1 - t0));
}

template <int MODE, int FMAS, int DEPTH, int NT>
int run(const char* name, double* out, unsigned long long* cyc, int iters) {
  const size_t lds = (size_t)NT * 8 * 16;
  CK(hipFuncSetAttribute((const void*)probe<MODE, FMAS, DEPTH, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipMemset(cyc, 0, 8));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((probe<MODE, FMAS, DEPTH, NT>), dim3(256), dim3(NT), lds, 0, out, iters, cyc);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long c = 0;
    CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    const double bytes_cu = (double)iters * 9 * 8 * NT * 16;   // per CU
    if (rep == 1)
      printf("%-34s threads %4d: %.3f ms  -> %6.1f B/clk/CU by clock64 (%.0f cycles), %6.1f B/clk/CU at 2.4 GHz wall, %5.1f TB/s chip\n", name,
             NT, ms, bytes_cu / ((double)c / 256.0), (double)c / 256.0, bytes_cu / (ms * 1e-3 * 2.4e9), bytes_cu * 256 / (ms * 1e-3) * 1e-12);
  }
  return 0;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 2000;
  double* out;
  unsigned long long* cyc;
  CK(hipMalloc(&out, sizeof(double) * 256 * 1024));
  CK(hipMalloc(&cyc, 8));
  run<0, 0, 8, 512>("burst 8, no arithmetic", out, cyc, iters);
  run<0, 2, 8, 512>("burst 8, 2 fma per read", out, cyc, iters);
  run<0, 6, 8, 512>("burst 8, 6 fma per read", out, cyc, iters);
  run<1, 0, 8, 512>("ring depth 8, no arithmetic", out, cyc, iters);
  run<1, 0, 4, 512>("ring depth 4, no arithmetic", out, cyc, iters);
  run<1, 0, 2, 512>("ring depth 2, no arithmetic", out, cyc, iters);
  run<1, 2, 8, 512>("ring depth 8, 2 fma per read", out, cyc, iters);
  run<1, 2, 4, 512>("ring depth 4, 2 fma per read", out, cyc, iters);
  run<1, 2, 2, 512>("ring depth 2, 2 fma per read", out, cyc, iters);
  run<1, 6, 8, 512>("ring depth 8, 6 fma per read", out, cyc, iters);
  run<1, 6, 4, 512>("ring depth 4, 6 fma per read", out, cyc, iters);
  run<1, 6, 2, 512>("ring depth 2, 6 fma per read", out, cyc, iters);
  run<0, 0, 8, 256>("burst 8, no arithmetic", out, cyc, iters);
  run<1, 0, 8, 256>("ring depth 8, no arithmetic", out, cyc, iters);
  run<1, 2, 8, 256>("ring depth 8, 2 fma per read", out, cyc, iters);
  run<1, 6, 8, 256>("ring depth 8, 6 fma per read", out, cyc, iters);
  return 0;
}
