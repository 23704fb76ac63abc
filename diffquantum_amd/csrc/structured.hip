// Structured (matrix-free) propagator engine: SURVEY.md section 8-F1 / BASELINE config 5 regime (n >= 14 qubits,
// where a dense 2^n x 2^n Hamiltonian cannot exist: 550 GB at n = 18 in complex64).
//
//   H(t) = H0 + sum_k u_k(t) H_k,   every H_* = sum_m w_m P_m,  P_m a Pauli string (I/X/Y/Z per qubit)
//
// keeps the reference's "exponential of the SUM" step  psi <- exp(-i dt H(t)) psi  (diffqc.cc:192-198,
// sim_plain.py:137-146; NOT the split product formula of the dead code diffqc.cc:137-170), evaluated as the
// degree-m Taylor polynomial in Horner form  z_m = psi, z_{j-1} = psi + (A z_j)/j, A = -i dt H, with m chosen from
// theta = |dt| * sum |weights| >= ||A|| so that theta^(m+1)/(m+1)! <= tol. Every A z is a matrix-free pass:
//   (P psi)[i] = i^{nY} (-1)^{popc((i ^ x) & z)} psi[i ^ x]     (x = X/Y positions, z = Z/Y positions)
// Diagonal strings are folded into one real vector per step. HBM-bound by construction: the unit of work is one
// pass over the state batch (2 * 2^n * B * sizeof(T) bytes).
//
// Kernel layout (CDNA4): states are [2^n][B], batch contiguous, so any set of rows is coalesced. A workgroup
// stages a sub-cube of 2^p rows (p "local" qubits, any positions) x 16 batch columns in LDS; every string whose
// x-mask lies inside the local set is then served from LDS (XOR of the local row index). Strings are grouped
// by local set on the host (<= p distinct flipped qubits per group); one H application = one pass per group.
// The same tiling computes the control-gradient overlaps <zbar, P_m z>.
//
// Gradient = exact reverse mode of the Horner recursion (not a first-order GRAPE approximation):
//   zbar_0 = lambda_{t+1};  zbar_j = (A^H zbar_{j-1})/j;  lambda_t = sum_j zbar_j;
//   dL/du_k[t] = sum_j (dt/j) sum_{m in k} w_m Im( i^{nY_m} <zbar_{j-1}, P'_m z_j> )
// with psi_t rebuilt as T_m(-A_t) psi_{t+1} between checkpoints.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/diffqc_hip.h"
#include "dq_internal.h"

namespace {

thread_local std::string s_err;
struct SErr { int code; std::string msg; };
[[noreturn]] void sfail(int code, const std::string& m) { throw SErr{code, m}; }
#define SHIP(expr)                                                                         \
  do {                                                                                     \
    hipError_t e__ = (expr);                                                               \
    if (e__ != hipSuccess) {                                                               \
      (void)hipGetLastError();                                                             \
      sfail(e__ == hipErrorOutOfMemory ? DQ_ERR_NOMEM : DQ_ERR_HIP,                        \
            std::string(#expr) + " failed: " + hipGetErrorString(e__));                    \
    }                                                                                      \
  } while (0)

struct SBuf {
  void* p = nullptr;
  size_t bytes = 0;
  SBuf() = default;
  SBuf(const SBuf&) = delete;
  SBuf& operator=(const SBuf&) = delete;
  SBuf(SBuf&& o) noexcept : p(o.p), bytes(o.bytes) { o.p = nullptr; o.bytes = 0; }
  ~SBuf() { if (p) (void)hipFree(p); }
  void ensure(size_t n) {
    if (n <= bytes) return;
    if (p) (void)hipFree(p);
    p = nullptr; bytes = 0;
    SHIP(hipMalloc(&p, n));
    bytes = n;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr; bytes = 0;
  }
};

struct SHostBuf {  // pinned host memory (checkpoints of the first level when the HBM has room for one segment only)
  void* p = nullptr;
  size_t bytes = 0;
  SHostBuf() = default;
  SHostBuf(const SHostBuf&) = delete;
  SHostBuf& operator=(const SHostBuf&) = delete;
  SHostBuf(SHostBuf&& o) noexcept : p(o.p), bytes(o.bytes) { o.p = nullptr; o.bytes = 0; }
  ~SHostBuf() { if (p) (void)hipHostFree(p); }
  bool ensure(size_t n) {  // false (and no exception) when the host will not pin that much
    if (n <= bytes) return true;
    if (p) (void)hipHostFree(p);
    p = nullptr; bytes = 0;
    if (hipHostMalloc(&p, n, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); p = nullptr; return false; }
    bytes = n;
    return true;
  }
};

// how much host memory the first-level checkpoints may pin: 60 % of what this process may still use (cgroup limit or MemAvailable)
double s_host_budget() {
  double lim = 1e30;
  for (const char* f : {"/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory/memory.limit_in_bytes"}) {
    if (FILE* fp = fopen(f, "r")) {
      double v = 0;
      if (fscanf(fp, "%lf", &v) == 1 && v > 0) lim = std::min(lim, v);
      fclose(fp);
    }
  }
  if (FILE* fp = fopen("/proc/meminfo", "r")) {
    char key[64]; double kb = 0; char unit[16];
    while (fscanf(fp, "%63s %lf %15s", key, &kb, unit) >= 2)
      if (strcmp(key, "MemAvailable:") == 0) { lim = std::min(lim, kb * 1024.0); break; }
    fclose(fp);
  }
  return lim > 1e29 ? 0.0 : 0.6 * lim;
}

bool s_is_device_ptr(const void* p) {
  if (!p) return false;
  hipPointerAttribute_t at;
  memset(&at, 0, sizeof(at));
  if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }
  return at.type == hipMemoryTypeDevice || at.type == hipMemoryTypeManaged;
}

constexpr int SBC = 16;        // batch columns per tile (128 B rows for complex64, 256 B for complex128)
constexpr int STHREADS = 1024;  // 16 waves per 64 KB tile -> 2 workgroups = 32 waves per CU (latency hiding)
constexpr int SMAXT = 64;      // strings per group

struct DTerm {  // device view of one Pauli string inside a group
  unsigned xloc;   // x-mask compressed onto the group's local bits (XOR on the local row index)
  unsigned xmask;  // full masks for the sign
  unsigned zmask;
  int slot;        // index into the per-step coefficient table / overlap output
};

template <class T> struct LocalBits { static constexpr int P = 9; };   // 512 rows x 16 cols x 8 B  = 64 KB
template <> struct LocalBits<double> { static constexpr int P = 8; };  // 256 rows x 16 cols x 16 B = 64 KB

__device__ __forceinline__ unsigned deposit_bits(unsigned v, unsigned mask) {  // software pdep
  unsigned out = 0;
  while (mask) {
    const unsigned low = mask & (0u - mask);
    if (v & 1u) out |= low;
    v >>= 1;
    mask ^= low;
  }
  return out;
}

template <class T>
struct SApply {
  typedef typename dq::Cx<T>::type T2;
  const T2* y;      // input state
  const T2* prev;   // partial sums of earlier groups (may be null)
  const T2* add;    // addend of the final pass (may be null)
  T2* out;
  T2* acc;          // optional: acc = acc_in + out (lambda_t = lambda_{t+1} + sum_j zbar_j)
  const T2* acc_in; // the sum so far: acc itself, or another buffer for the first term (no copy of lambda_{t+1} into lambda_t)
  const T* diag;    // per-step real diagonal (first group only) or null
  const DTerm* terms;
  const T2* coef;   // per-slot complex coefficient for this step (weight * u_k * i^{nY})
  int nterms;
  int n, p;         // qubits, local bits
  unsigned lmask;
  int B;
  T sr, si;         // complex scale of the bracket
  // fused overlaps (OVL kernels): part[blk][m] = sum over the tile of conj(y[i ^ x_m]) * (-1)^popc(i & z_m) * zo[i],
  // i.e. <y, P'_m zo> with the roles of soverlap_kernel swapped (y is the tile here, zo is read at the thread's own rows)
  const T2* zo;
  const DTerm* oterms;
  int n_ovl;
  double* part;     // [nblocks * waves][n_ovl][2]
  const int* omap;  // DMA form: applied string m -> its index among the overlap strings, or -1 (the strings' partner rows are shared)
  T2* out2;         // chain form (schain_dma_kernel): the next application's partial sums, see there
  const T2* coef2 = nullptr;  // chain form: coefficients of the application out2 STARTS when that one belongs to the next step (null: coef);
                              // its diagonal then comes in as `diag` (the link reads the diagonal for out2 only)
  const T2* zo2;    // reverse chain form (srchain_dma_kernel): cotangent operand of the NEXT stage's overlaps, and their weight
  T ovl2_scale;     //   relative to this stage's (the two stages' sums share one accumulator set)
};

template <class T2, int V>
struct alignas(sizeof(T2) * V) CVec {
  T2 v[V];
};

// out = add + (sr + i si) * ( prev + diag .* y + sum_m coef_m sign_m P'_m y )
// Which optional operands exist is a compile-time property (FLAGS bit 0 prev, 1 add, 2 acc, 3 diag): a runtime
// "load or zero" per element makes hipcc branch around every load and drain vmcnt each time (measured 3.9 -> 2.7 TB/s).
// V = batch columns per lane (2 when the batch is even: 16-byte accesses for complex64, half the address math).
constexpr int SOVL_TCH = 12;  // control strings per register chunk of the overlap sums (soverlap_kernel, fused sapply_kernel)

// Wave-wide sum without the LDS pipe (ds_bpermute shuffles of 2 x 12 sums per tile cost more than the tile's HBM time share):
// DPP adds inside each row of 16 lanes (xor 1, xor 2, half mirror, mirror), then the four row sums through v_readlane.
template <int CTRL>
__device__ __forceinline__ float s_dpp(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ double s_dpp(double x) {
  const long long b = __double_as_longlong(x);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, false);
  return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double s_lane(float x, int l) { return (double)__int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), l)); }
__device__ __forceinline__ double s_lane(double x, int l) {
  const long long b = __double_as_longlong(x);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), l), hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
  return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
template <class T>
__device__ __forceinline__ double s_wave_sum(T x) {
  x += s_dpp<0xB1>(x);   // quad_perm [1,0,3,2]
  x += s_dpp<0x4E>(x);   // quad_perm [2,3,0,1]
  x += s_dpp<0x141>(x);  // row_half_mirror
  x += s_dpp<0x140>(x);  // row_mirror
  return (s_lane(x, 0) + s_lane(x, 16)) + (s_lane(x, 32) + s_lane(x, 48));
}

template <class T, int FLAGS, int V, bool OVL = false>
__global__ void __launch_bounds__(STHREADS) sapply_kernel(SApply<T> a) {
  constexpr bool HAS_PREV = FLAGS & 1, HAS_ADD = FLAGS & 2, HAS_ACC = FLAGS & 4, HAS_DIAG = FLAGS & 8;
  typedef typename dq::Cx<T>::type T2;
  typedef CVec<T2, V> VT;
  constexpr int PMAX = LocalBits<T>::P;
  constexpr int CPL = SBC / V;              // lanes across one tile row
  constexpr int RSTEP = STHREADS / CPL;     // rows covered by one sweep of the workgroup
  constexpr int RPT = ((1 << PMAX) + RSTEP - 1) / RSTEP;
  __shared__ T2 tile[(1 << PMAX) * SBC];
  __shared__ unsigned rowidx[1 << PMAX];
  __shared__ DTerm tt[SMAXT];
  __shared__ T2 tc[SMAXT];
  __shared__ DTerm to[OVL ? SOVL_TCH : 1];
  const int R = 1 << a.p;
  const unsigned full = (a.n >= 32) ? 0xffffffffu : ((1u << a.n) - 1u);
  const unsigned base = deposit_bits(blockIdx.x, full & ~a.lmask);
  const int c0 = blockIdx.y * SBC;
  for (int r = threadIdx.x; r < R; r += STHREADS) rowidx[r] = base | deposit_bits((unsigned)r, a.lmask);
  for (int m = threadIdx.x; m < a.nterms; m += STHREADS) { tt[m] = a.terms[m]; tc[m] = a.coef[a.terms[m].slot]; }
  if (OVL)
    for (int m = threadIdx.x; m < a.n_ovl; m += STHREADS) to[m] = a.oterms[m];
  __syncthreads();
  const int c = (threadIdx.x % CPL) * V;
  const int rbase = threadIdx.x / CPL;
  const bool col_ok = (c0 + c) < a.B;   // B is a multiple of V, so a lane's V columns are valid together
  VT zero;
#pragma unroll
  for (int w = 0; w < V; ++w) { zero.v[w].x = 0; zero.v[w].y = 0; }
  for (int r = rbase; r < R; r += RSTEP) {
    VT v = zero;
    if (col_ok) v = *(const VT*)(a.y + (size_t)rowidx[r] * a.B + c0 + c);
    *(VT*)(tile + r * SBC + c) = v;
  }
  __syncthreads();
  if (!OVL && !col_ok) return;
  // every global operand of this thread's rows is requested before the LDS work starts
  VT pv[RPT], ad[RPT], ac[RPT], zo[OVL ? RPT : 1];
  T dg[RPT];
  T px[OVL ? SOVL_TCH : 1], py[OVL ? SOVL_TCH : 1];
  if (OVL) {
#pragma unroll
    for (int mm = 0; mm < SOVL_TCH; ++mm) { px[mm] = 0; py[mm] = 0; }
  }
  if (col_ok) {
#pragma unroll
  for (int q = 0; q < RPT; ++q) {
    const int r = min(rbase + q * RSTEP, R - 1);  // clamped: loads stay unconditional
    const size_t off = (size_t)rowidx[r] * a.B + c0 + c;
    if (HAS_PREV) pv[q] = *(const VT*)(a.prev + off);
    if (HAS_ADD) ad[q] = *(const VT*)(a.add + off);
    if (HAS_ACC) ac[q] = *(const VT*)(a.acc_in + off);
    if (HAS_DIAG) dg[q] = a.diag[rowidx[r]];
    if (OVL) zo[q] = *(const VT*)(a.zo + off);
  }
#pragma unroll
  for (int q = 0; q < RPT; ++q) {
    const int r = rbase + q * RSTEP;
    if (r >= R) break;
    const unsigned i = rowidx[r];
    const size_t off = (size_t)i * a.B + c0 + c;
    T sx[V], sy[V];
#pragma unroll
    for (int w = 0; w < V; ++w) { sx[w] = 0; sy[w] = 0; }
    if (HAS_PREV) {
#pragma unroll
      for (int w = 0; w < V; ++w) { sx[w] = pv[q].v[w].x; sy[w] = pv[q].v[w].y; }
    }
    if (HAS_DIAG) {
      const VT v = *(const VT*)(tile + r * SBC + c);
#pragma unroll
      for (int w = 0; w < V; ++w) { sx[w] = fma(dg[q], v.v[w].x, sx[w]); sy[w] = fma(dg[q], v.v[w].y, sy[w]); }
    }
    for (int m = 0; m < a.nterms; ++m) {
      const VT v = *(const VT*)(tile + (r ^ tt[m].xloc) * SBC + c);
      T2 cf = tc[m];
      if (tt[m].zmask != 0 && (__popc((i ^ tt[m].xmask) & tt[m].zmask) & 1)) { cf.x = -cf.x; cf.y = -cf.y; }  // uniform branch
#pragma unroll
      for (int w = 0; w < V; ++w) {
        sx[w] = fma(cf.x, v.v[w].x, sx[w]); sx[w] = fma(-cf.y, v.v[w].y, sx[w]);
        sy[w] = fma(cf.x, v.v[w].y, sy[w]); sy[w] = fma(cf.y, v.v[w].x, sy[w]);
      }
    }
    VT o;
#pragma unroll
    for (int w = 0; w < V; ++w) {
      o.v[w].x = a.sr * sx[w] - a.si * sy[w];
      o.v[w].y = a.sr * sy[w] + a.si * sx[w];
      if (HAS_ADD) { o.v[w].x += ad[q].v[w].x; o.v[w].y += ad[q].v[w].y; }
    }
    *(VT*)(a.out + off) = o;
    if (HAS_ACC) {
      VT w2;
#pragma unroll
      for (int w = 0; w < V; ++w) { w2.v[w].x = ac[q].v[w].x + o.v[w].x; w2.v[w].y = ac[q].v[w].y + o.v[w].y; }
      *(VT*)(a.acc + off) = w2;
    }
    if (OVL) {
#pragma unroll
      for (int mm = 0; mm < SOVL_TCH; ++mm)
        if (mm < a.n_ovl) {  // uniform
          const DTerm t = to[mm];
          const VT v = *(const VT*)(tile + (r ^ t.xloc) * SBC + c);
          T qx = 0, qy = 0;
#pragma unroll
          for (int w = 0; w < V; ++w) {
            qx += v.v[w].x * zo[q].v[w].x + v.v[w].y * zo[q].v[w].y;   // conj(y[i ^ x]) * zo[i]
            qy += v.v[w].x * zo[q].v[w].y - v.v[w].y * zo[q].v[w].x;
          }
          if (t.zmask != 0 && (__popc(i & t.zmask) & 1)) { qx = -qx; qy = -qy; }
          px[mm] += qx; py[mm] += qy;
        }
    }
  }
  }
  if (OVL) {  // one sum per wave and string (no workgroup barrier); soverlap_finish_kernel adds waves and workgroups in order
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t blk = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (STHREADS / 64) + wave;
#pragma unroll
    for (int mm = 0; mm < SOVL_TCH; ++mm)
      if (mm < a.n_ovl) {
        const double tx = s_wave_sum<T>(px[mm]), ty = s_wave_sum<T>(py[mm]);
        if (lane == 0) {
          a.part[(blk * a.n_ovl + mm) * 2 + 0] = tx;
          a.part[(blk * a.n_ovl + mm) * 2 + 1] = ty;
        }
      }
  }
}

// ---- LDS-DMA forms of the passes (round 3: complex64, 2^9-row tiles; round 4: complex128 too, 2^8-row tiles), batch a multiple of 16 ----
// The kernels above run ONE tile per workgroup: load -> barrier -> LDS work -> store, with whatever a second workgroup on the CU
// overlaps (3.6-4.6 TB/s at B = 512; soverlap_kernel, at 80 VGPRs alone on its CU, 2.8 TB/s).  Here a workgroup is persistent and
// walks its tiles with TWO tile buffers (128 KB of LDS): while it works on tile t, tile t + 1 streams into the other buffer by
// LDS-DMA (global_load_lds_dwordx4: no staging registers; one wave instruction moves 8 rows x 128 B = 1 KiB -- four per wave and tile --, every lane with
// its own global address), so the HBM pipe of the CU never drains.  One barrier per tile: a wave reaches it only after its LDS
// work on the previous tile, so the buffer the next DMA overwrites is free.  vmcnt is in issue order: the wait in front of the
// barrier leaves exactly the previous tile's stores in flight (they are younger than the DMA it waits for).
// (inline asm, as in zgemm_lin.hip: for the builtin hipcc drains vmcnt in front of every LDS read)
// (addresses as uniform base pointer in SGPRs + one 32-bit byte offset per row, shared by every operand of the row: per-thread 64-bit
// pointers for eight arrays cost 16 VGPRs and spilled the reverse chain kernel; the DMA forms therefore need states below 4 GB)
// (s_nop 4 in front of every one of these: under SGPR pressure hipcc restores the base pointer with v_readlane_b32 right before the asm
// block, and a VMEM instruction that reads an SGPR written by a VALU instruction needs 5 wait states -- the hazard recognizer does not
// look inside inline asm.  Without it the reverse chain kernel read a half-restored pointer and faulted.)
// (m0 is saved and restored inside the block: it is a reserved register that cannot be named as a clobber, and the compiler may
// hold its own value there -- s_set_gpr_idx / LDS-DMA builtins elsewhere in a kernel)
#define S_DMA16(base_, off_, lds_)                                                                                               \
  do {                                                                                                                           \
    unsigned m0_keep_;                                                                                                           \
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0" \
                 : "=&s"(m0_keep_)                                                                                               \
                 : "v"(off_), "s"(base_), "s"(lds_)                                                                              \
                 : "memory");                                                                                                    \
  } while (0)

constexpr int SD_P = 9, SD_ROWS = 1 << SD_P, SD_RSTEP = 128, SD_RPT = SD_ROWS / SD_RSTEP;  // 1024 threads = 128 rows x 8 column pairs; 4 rows per thread
static_assert(STHREADS == 1024 && SBC == 16, "the DMA forms assume 1024 threads and 16 batch columns per tile");

// tile units.  A group whose local bits are exactly the low row bits (a tile = 2^9 consecutive rows) walks chunk-fastest: the workgroups
// that run at the same time then cover whole rows of a few tiles, not the same 128-byte column chunk of 256 tiles (2-unit pass at
// B = 512: 621 -> 554 us).  Every other group (rows of a tile spread out; the interleaved groups of Ising-type inputs) walks
// tile-fastest: neighbouring tiles are neighbouring rows of the same regions, which the concurrent workgroups then share
// (n = 18, interleaved groups, ms per step at B = 512 / 64: tile-fastest 23.6 / 3.07, chunk-fastest 24.6 / 3.23).
struct SDmaWalk {
  unsigned nchunks, nunits, ntiles, tile_fast;
  __host__ __device__ unsigned tile(unsigned u) const { return tile_fast ? u % ntiles : u / nchunks; }
  __host__ __device__ unsigned chunk(unsigned u) const { return tile_fast ? u / ntiles : u % nchunks; }
};
inline SDmaWalk sdma_walk(unsigned ntiles, unsigned nchunks, unsigned lmask) {
  const bool low_block = (lmask & (lmask + 1u)) == 0;   // the local bits are exactly the low row bits: a tile = consecutive rows
  return SDmaWalk{nchunks, ntiles * nchunks, ntiles, low_block ? 0u : 1u};
}

// Register operands of a tile (prev / add / acc rows, diagonal entries) are requested by inline asm as well: vmcnt counts in issue
// order, and a load the compiler knows about makes it wait with vmcnt(0) at the first use -- i.e. for the next tile's DMA, which
// was issued behind it (the first version of this kernel did exactly that and ran no faster than the one-tile kernel).  Order per
// tile: operand requests -> DMA of tile t + 1 -> LDS gathers of tile t (no vector-memory wait) -> vmcnt(4): operands in, the four
// DMA instructions still in flight -> combine, store.
typedef float sd_f4 __attribute__((ext_vector_type(4)));
#define S_GLOAD16(dst_, base_, off_) asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=&v"(dst_) : "v"(off_), "s"(base_) : "memory")
#define S_GLOAD4(dst_, base_, off_) asm volatile("s_nop 4\n\tglobal_load_dword %0, %1, %2" : "=&v"(dst_) : "v"(off_), "s"(base_) : "memory")
// S_PIN: for the compiler the asm load DEFINES its destination at issue; the empty asm after the vmcnt wait makes every use depend
// on the wait.  Between the two the registers must stay where they are: the kernels below are kept free of VGPR spills (hipcc
// -Rpass-analysis=kernel-resource-usage: ScratchSize 0 for every sapply_dma / schain / srchain instance that is launched).
#define S_PIN(x_) asm volatile("" : "+v"(x_))
// (-DSD_NT_STORES: nontemporal stores; measured on one box: B = 64 2.69 against 2.74 ms per step, B = 512 no difference -- left off)
template <class V>
__device__ __forceinline__ void s_store16(void* p, V v) {
#ifdef SD_NT_STORES
  __builtin_nontemporal_store(v, (V*)p);
#else
  *(V*)p = v;
#endif
}
#define S_STORE16(p_, v_) s_store16((p_), (v_))

// SIMPLE: every string of the group is a pure X-string (no Z factor: no row-dependent sign, real coefficient) and there are at most
// SD_SIMPLE_T of them -- offsets and coefficients sit in scalar registers, a gather is one v_xor + ds_read_b128 + two packed FMAs
// (the general loop issues ~3x the instructions per tile, and at 16 waves per CU the passes were bound by instruction issue:
// ~900 wave-instructions per tile and wave = 6 us of the 9.7 us a tile took).
// Per-dtype shape of the DMA forms (round 4: complex128 too).  Both move 16 B per lane and hold 64 KB tiles: complex64 = 2^9 rows x 16
// columns, a thread owns 2 columns of 4 rows (8 threads per row); complex128 = 2^8 rows x 16 columns, a thread owns 1 column of 4 rows
// (16 threads per row).  One LDS-DMA wave instruction (64 lanes x 16 B) fills 8 rows of 128 B or 4 rows of 256 B.
typedef double sd_d2 __attribute__((ext_vector_type(2)));
template <class T> struct SDm;
template <> struct SDm<float> { typedef sd_f4 V; static constexpr int P = 9, TPR = 8, CPT = 2, EB = 8; };
template <> struct SDm<double> { typedef sd_d2 V; static constexpr int P = 8, TPR = 16, CPT = 1, EB = 16; };
static_assert(SDm<float>::P == LocalBits<float>::P && SDm<double>::P == LocalBits<double>::P, "DMA tiles = the groups' tiles");
#define S_GLOAD8(dst_, base_, off_) asm volatile("s_nop 4\n\tglobal_load_dwordx2 %0, %1, %2" : "=&v"(dst_) : "v"(off_), "s"(base_) : "memory")
__device__ __forceinline__ void sd_gload_real(float& dst, const float* base, unsigned off) { S_GLOAD4(dst, base, off); }
__device__ __forceinline__ void sd_gload_real(double& dst, const double* base, unsigned off) { S_GLOAD8(dst, base, off); }
__device__ __forceinline__ float sd_readlane(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }
__device__ __forceinline__ double sd_readlane(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ float sd_uniform(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); }
__device__ __forceinline__ double sd_uniform(double v) {
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
// (sr + i si) * w for the complex numbers a lane's 16 bytes hold
__device__ __forceinline__ sd_f4 sd_rot(float sr, float si, sd_f4 w) {
  sd_f4 o;
  o.x = sr * w.x - si * w.y; o.y = sr * w.y + si * w.x;
  o.z = sr * w.z - si * w.w; o.w = sr * w.w + si * w.z;
  return o;
}
__device__ __forceinline__ sd_d2 sd_rot(double sr, double si, sd_d2 w) {
  sd_d2 o;
  o.x = sr * w.x - si * w.y; o.y = sr * w.y + si * w.x;
  return o;
}
// Re / Im of conj(v) * z summed over the lane's complex numbers
__device__ __forceinline__ float sd_cre(sd_f4 v, sd_f4 z) { return (v.x * z.x + v.y * z.y) + (v.z * z.z + v.w * z.w); }
__device__ __forceinline__ float sd_cim(sd_f4 v, sd_f4 z) { return (v.x * z.y - v.y * z.x) + (v.z * z.w - v.w * z.z); }
__device__ __forceinline__ double sd_cre(sd_d2 v, sd_d2 z) { return v.x * z.x + v.y * z.y; }
__device__ __forceinline__ double sd_cim(sd_d2 v, sd_d2 z) { return v.x * z.y - v.y * z.x; }
// s += cf * v (complex coefficient)
__device__ __forceinline__ void sd_cfma(float2 cf, sd_f4 v, sd_f4& s) {
  s.x = fmaf(cf.x, v.x, s.x); s.x = fmaf(-cf.y, v.y, s.x);
  s.y = fmaf(cf.x, v.y, s.y); s.y = fmaf(cf.y, v.x, s.y);
  s.z = fmaf(cf.x, v.z, s.z); s.z = fmaf(-cf.y, v.w, s.z);
  s.w = fmaf(cf.x, v.w, s.w); s.w = fmaf(cf.y, v.z, s.w);
}
__device__ __forceinline__ void sd_cfma(double2 cf, sd_d2 v, sd_d2& s) {
  s.x = fma(cf.x, v.x, s.x); s.x = fma(-cf.y, v.y, s.x);
  s.y = fma(cf.x, v.y, s.y); s.y = fma(cf.y, v.x, s.y);
}

constexpr int SD_SIMPLE_T = 10;

// OVL (SIMPLE only, at most SD_OVL_T strings): the pass also forms the overlaps <y, P_m zo> of the group's control strings -- they are
// among the applied strings, so the partner rows of y are gathered once for both; zo is one more register operand per row.
constexpr int SD_OVL_T = 10;

template <class T, int FLAGS, bool SIMPLE, bool OVL = false>
__global__ void __launch_bounds__(STHREADS) sapply_dma_kernel(SApply<T> a, SDmaWalk wk) {
  static_assert(!OVL || SIMPLE, "fused overlaps exist for the pure X-string form only");
  constexpr bool HAS_PREV = FLAGS & 1, HAS_ADD = FLAGS & 2, HAS_ACC = FLAGS & 4, HAS_DIAG = FLAGS & 8;
  constexpr bool NO_OUT = FLAGS & 16;   // the result is only accumulated (acc), not kept: the last stage of the reverse sweep's chain (zbar_m is read by nobody)
  static_assert(!NO_OUT || HAS_ACC, "a pass without `out` must have `acc`");
  constexpr int NSTORE = SD_RPT * ((NO_OUT ? 0 : 1) + (HAS_ACC ? 1 : 0));  // vector-memory stores a thread issues per tile
  typedef typename dq::Cx<T>::type T2;
  typedef SDm<T> SD;
  typedef typename SD::V V;
  constexpr int ROWS = 1 << SD::P, RSTEP = STHREADS / SD::TPR, EB = SD::EB;
  static_assert(ROWS / RSTEP == SD_RPT, "4 rows per thread in both forms");
  extern __shared__ __align__(16) unsigned char sd_raw[];
  T2* tiles = (T2*)sd_raw;  // [2][512 * 16]
  __shared__ DTerm tt[SMAXT];
  __shared__ T2 tc[SMAXT];
  const unsigned full = (a.n >= 32) ? 0xffffffffu : ((1u << a.n) - 1u);
  if (!SIMPLE)
    for (int m = threadIdx.x; m < a.nterms; m += STHREADS) { tt[m] = a.terms[m]; tc[m] = a.coef[a.terms[m].slot]; }
  unsigned xl[SIMPLE ? SD_SIMPLE_T : 1];  // byte offset of the partner row (XOR on the row's LDS address), per string: uniform
  T cr[SIMPLE ? SD_SIMPLE_T : 1];
  unsigned omask = 0;  // applied strings whose overlap is wanted
  T px[OVL ? SD_OVL_T : 1], py[OVL ? SD_OVL_T : 1];
  if (OVL) {
#pragma unroll
    for (int m = 0; m < SD_OVL_T; ++m) {
      px[m] = (T)0; py[m] = (T)0;
      if (m < a.nterms && a.omap[m] >= 0) omask |= 1u << m;
    }
    omask = __builtin_amdgcn_readfirstlane(omask);
  }
  if (SIMPLE) {
#pragma unroll
    for (int m = 0; m < SD_SIMPLE_T; ++m) {
      xl[m] = 0; cr[m] = (T)0;
      if (m < a.nterms) {  // uniform (a group may have no strings at all: its table is then unset)
        const DTerm t = a.terms[m];
        xl[m] = __builtin_amdgcn_readfirstlane(t.xloc * (unsigned)(SBC * EB));
        cr[m] = sd_uniform(a.coef[t.slot].x);
      }
    }
  }
  const int c = (threadIdx.x % SD::TPR) * SD::CPT;
  const int rbase = threadIdx.x / SD::TPR;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned dep[SD_RPT];
#pragma unroll
  for (int q = 0; q < SD_RPT; ++q) dep[q] = deposit_bits((unsigned)(rbase + q * RSTEP), a.lmask);
  const unsigned lds0 = (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) void*)tiles;
  auto issue = [&](unsigned u, int buf) {  // tile unit u -> buffer buf (this wave's 4 pieces of 1 KiB)
    const unsigned base = deposit_bits(wk.tile(u), full & ~a.lmask);
    const int c0 = (int)wk.chunk(u) * SBC;
#pragma unroll
    for (int q = 0; q < SD_RPT; ++q) {
      const unsigned off = ((base | dep[q]) * (unsigned)a.B + (unsigned)(c0 + c)) * (unsigned)EB;
      const unsigned ld = lds0 + (unsigned)buf * (ROWS * SBC * EB) + (unsigned)(wave * (64 / SD::TPR) + q * RSTEP) * (SBC * EB);
      S_DMA16(a.y, off, ld);
    }
  };
  unsigned u = blockIdx.x;
  if (u < wk.nunits) issue(u, 0);
  int buf = 0;
  bool first = true;
  for (; u < wk.nunits; u += gridDim.x, buf ^= 1, first = false) {
    if (first) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NSTORE) : "memory");
    const unsigned base = deposit_bits(wk.tile(u), full & ~a.lmask);
    const int c0 = (int)wk.chunk(u) * SBC;
    V pv[SD_RPT], ad[SD_RPT], ac[SD_RPT], zo[OVL ? SD_RPT : 1];
    T dg[SD_RPT];
#pragma unroll
    for (int q = 0; q < SD_RPT; ++q) {
      const unsigned i = base | dep[q];
      const unsigned off = (i * (unsigned)a.B + (unsigned)(c0 + c)) * (unsigned)EB;   // byte offset: the same for every operand
      if (OVL) S_GLOAD16(zo[q], a.zo, off);
      if (HAS_PREV) S_GLOAD16(pv[q], a.prev, off);
      if (HAS_ADD) S_GLOAD16(ad[q], a.add, off);
      if (HAS_ACC) S_GLOAD16(ac[q], a.acc_in, off);
      if (HAS_DIAG) sd_gload_real(dg[q], a.diag, (i) * (unsigned)sizeof(T));
    }
    const bool more = u + gridDim.x < wk.nunits;  // uniform
    if (more) issue(u + gridDim.x, buf ^ 1);
    const T2* tile = tiles + (size_t)buf * (ROWS * SBC);
    V s[SD_RPT];   // (re, im) of the thread's two columns, per row
#pragma unroll
    for (int q = 0; q < SD_RPT; ++q) s[q] = (V)((T)0);
    if (OVL) {  // zo is needed inside the gathers: the operands are waited for here (the next tile's DMA stays in flight)
      if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SD_RPT) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int q = 0; q < SD_RPT; ++q) {
        S_PIN(zo[q]);
        if (HAS_PREV) { S_PIN(pv[q]); s[q] = pv[q]; }  // the sums start from the earlier groups' partial sums: no registers of their own
      }
    }
    if (SIMPLE) {
      const unsigned char* tb = (const unsigned char*)tile;
      const unsigned ra = (unsigned)(rbase * SBC + c) * (unsigned)EB;  // own row and columns; the rows q * 128 above it are bits 14-15
#pragma unroll
      for (int m = 0; m < SD_SIMPLE_T; ++m)
        if (m < a.nterms) {  // uniform
          // (opaque per tile: otherwise hipcc hoists all strings x rows partner addresses out of the tile loop -- 48 to 64 VGPRs)
          asm volatile("" : "+s"(xl[m]));
#pragma unroll
          for (int q = 0; q < SD_RPT; ++q) {
            const V v = *(const V*)(tb + ((ra | (unsigned)(q * RSTEP * SBC * EB)) ^ xl[m]));
            s[q] += cr[m] * v;
            if (OVL && m < SD_OVL_T && (omask >> m & 1u)) {  // uniform; conj(y[i ^ x]) * zo[i]
              px[m < SD_OVL_T ? m : 0] += sd_cre(v, zo[q]);
              py[m < SD_OVL_T ? m : 0] += sd_cim(v, zo[q]);
            }
          }
        }
    } else {
      for (int m = 0; m < a.nterms; ++m) {
        const DTerm t = tt[m];
        const T2 cf0 = tc[m];
#pragma unroll
        for (int q = 0; q < SD_RPT; ++q) {
          const int r = rbase + q * RSTEP;
          const V v = *(const V*)(tile + (r ^ t.xloc) * SBC + c);
          T2 cf = cf0;
          if (t.zmask != 0 && (__popc(((base | dep[q]) ^ t.xmask) & t.zmask) & 1)) { cf.x = -cf.x; cf.y = -cf.y; }
          sd_cfma(cf, v, s[q]);
        }
      }
    }
    // the operands: everything but the next tile's four DMA instructions has to be in
    if (FLAGS != 0) {
      if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SD_RPT) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int q = 0; q < SD_RPT; ++q) {
        if (HAS_PREV && !OVL) S_PIN(pv[q]);
        if (HAS_ADD) S_PIN(ad[q]);
        if (HAS_ACC) S_PIN(ac[q]);
        if (HAS_DIAG) S_PIN(dg[q]);
      }
    }
#pragma unroll
    for (int q = 0; q < SD_RPT; ++q) {
      const int r = rbase + q * RSTEP;
      const unsigned off = ((base | dep[q]) * (unsigned)a.B + (unsigned)(c0 + c)) * (unsigned)EB;
      V w = s[q];
      if (HAS_DIAG) {
        const V v = *(const V*)(tile + r * SBC + c);
        w += dg[q] * v;
      }
      if (HAS_PREV && !OVL) w += pv[q];
      V o = sd_rot(a.sr, a.si, w);
      if (HAS_ADD) o += ad[q];
      if (!NO_OUT) S_STORE16((unsigned char*)a.out + off, o);
      if (HAS_ACC) S_STORE16((unsigned char*)a.acc + off, ac[q] + o);
    }
  }
  if (OVL) {  // one sum per wave and string over ALL tiles of the workgroup; soverlap_finish_kernel adds them in a fixed order
    const int lane = threadIdx.x & 63;
    const size_t blk = (size_t)blockIdx.x * (STHREADS / 64) + wave;
#pragma unroll
    for (int m = 0; m < SD_OVL_T; ++m)
      if (omask >> m & 1u) {
        const double tx = s_wave_sum<T>(px[m]), ty = s_wave_sum<T>(py[m]);
        if (lane == 0) {
          const int mm = a.omap[m];
          a.part[(blk * a.n_ovl + mm) * 2 + 0] = tx;
          a.part[(blk * a.n_ovl + mm) * 2 + 1] = ty;
        }
      }
  }
}

// ---- chain form: two applications share one pass ---------------------------------------------------------------------------------
// A Horner chain z_{j-1} = in + s_j (H z_j) over two string groups A, B walks the groups in alternating order: A B | B A | A B ...
// The pass that FINISHES an application on group g (z' = add + s (prev + H_g y): prev holds D y + H_other y) has z' complete, and the
// next application may start on the same group g with the tile it has: after a barrier the tile's rows are overwritten in place
// with z', and out2 = D z' + H_g z' goes out as the next pass's `prev`.  Per application 5 state-units (y, prev, add in; z', out2
// out) instead of 2 + 4, and m + 1 tile walks for m applications instead of 2 m.  Pure X-string groups only (scalar tables).
template <class T>
__global__ void __launch_bounds__(STHREADS) schain_dma_kernel(SApply<T> a, SDmaWalk wk) {
  constexpr int NSTORE = 2 * SD_RPT;
  typedef typename dq::Cx<T>::type T2;
  typedef SDm<T> SD;
  typedef typename SD::V V;
  constexpr int ROWS = 1 << SD::P, RSTEP = STHREADS / SD::TPR, EB = SD::EB;
  static_assert(ROWS / RSTEP == SD_RPT, "4 rows per thread in both forms");
  extern __shared__ __align__(16) unsigned char sd_raw[];
  T2* tiles = (T2*)sd_raw;  // [2][512 * 16]
  const unsigned full = (a.n >= 32) ? 0xffffffffu : ((1u << a.n) - 1u);
  unsigned xl[SD_SIMPLE_T];
  // the strings' coefficients, one per lane (lane m: string m), read back with v_readlane inside the gathers: crl for the application
  // the pass finishes, cr2l for the one it starts (the next step's at a handover).  Two sets of scalars beside the offsets did not
  // fit the scalar file: 735 / 1603 spilled SGPRs (complex64 / complex128) where the one-set kernel had 27 / 29.
  T crl = (T)0, cr2l = (T)0;
  {
    const int ln = threadIdx.x & 63;
    if (ln < a.nterms && ln < SD_SIMPLE_T) {
      const int slot = a.terms[ln].slot;
      crl = a.coef[slot].x;
      cr2l = (a.coef2 ? a.coef2 : a.coef)[slot].x;
    }
  }
#pragma unroll
  for (int m = 0; m < SD_SIMPLE_T; ++m) {
    xl[m] = 0;
    if (m < a.nterms) xl[m] = __builtin_amdgcn_readfirstlane(a.terms[m].xloc * (unsigned)(SBC * EB));
  }
  const bool has_diag = a.diag != nullptr;  // uniform
  const int c = (threadIdx.x % SD::TPR) * SD::CPT;
  const int rbase = threadIdx.x / SD::TPR;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned dep[SD_RPT];
#pragma unroll
  for (int q = 0; q < SD_RPT; ++q) dep[q] = deposit_bits((unsigned)(rbase + q * RSTEP), a.lmask);
  const unsigned lds0 = (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) void*)tiles;
  auto issue = [&](unsigned u, int buf) {
    const unsigned base = deposit_bits(wk.tile(u), full & ~a.lmask);
    const int c0 = (int)wk.chunk(u) * SBC;
#pragma unroll
    for (int q = 0; q < SD_RPT; ++q) {
      const unsigned off = ((base | dep[q]) * (unsigned)a.B + (unsigned)(c0 + c)) * (unsigned)EB;
      const unsigned ld = lds0 + (unsigned)buf * (ROWS * SBC * EB) + (unsigned)(wave * (64 / SD::TPR) + q * RSTEP) * (SBC * EB);
      S_DMA16(a.y, off, ld);
    }
  };
  const unsigned ra = (unsigned)(rbase * SBC + c) * (unsigned)EB;
  unsigned u = blockIdx.x;
  if (u < wk.nunits) issue(u, 0);
  int buf = 0;
  bool first = true;
  for (; u < wk.nunits; u += gridDim.x, buf ^= 1, first = false) {
    if (first) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NSTORE) : "memory");
    const unsigned base = deposit_bits(wk.tile(u), full & ~a.lmask);
    const int c0 = (int)wk.chunk(u) * SBC;
    V pv[SD_RPT], ad[SD_RPT];
    T dg[SD_RPT];
#pragma unroll
    for (int q = 0; q < SD_RPT; ++q) {
      const unsigned i = base | dep[q];
      const unsigned off = (i * (unsigned)a.B + (unsigned)(c0 + c)) * (unsigned)EB;   // byte offset: the same for every operand
      S_GLOAD16(pv[q], a.prev, off);
      S_GLOAD16(ad[q], a.add, off);
      dg[q] = (T)0;
      if (has_diag) sd_gload_real(dg[q], a.diag, (i) * (unsigned)sizeof(T));
    }
    const bool more = u + gridDim.x < wk.nunits;  // uniform
    if (more) issue(u + gridDim.x, buf ^ 1);
    unsigned char* tb = (unsigned char*)(tiles + (size_t)buf * (ROWS * SBC));
    V s[SD_RPT];
    auto gather = [&](T cfl) {
#pragma unroll
      for (int q = 0; q < SD_RPT; ++q) s[q] = (V)((T)0);
#pragma unroll
      for (int m = 0; m < SD_SIMPLE_T; ++m)
        if (m < a.nterms) {  // uniform
          asm volatile("" : "+s"(xl[m]));
          const T cf = sd_readlane(cfl, m);
#pragma unroll
          for (int q = 0; q < SD_RPT; ++q) {
            const V v = *(const V*)(tb + ((ra | (unsigned)(q * RSTEP * SBC * EB)) ^ xl[m]));
            s[q] += cf * v;
          }
        }
    };
    gather(crl);  // H_g y
    if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SD_RPT) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    V z[SD_RPT];
#pragma unroll
    for (int q = 0; q < SD_RPT; ++q) {
      S_PIN(pv[q]); S_PIN(ad[q]);
      if (has_diag) S_PIN(dg[q]);
      const unsigned off = ((base | dep[q]) * (unsigned)a.B + (unsigned)(c0 + c)) * (unsigned)EB;
      const V w = s[q] + pv[q];
      z[q] = sd_rot(a.sr, a.si, w) + ad[q];
      S_STORE16((unsigned char*)a.out + off, z[q]);
    }
    // every wave is done with the rows of y: the tile becomes z' in place (LDS-only waits: the DMA and the stores stay in flight)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
    for (int q = 0; q < SD_RPT; ++q) *(V*)(tb + (ra | (unsigned)(q * RSTEP * SBC * EB))) = z[q];
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    gather(cr2l);  // H_g z' (at a handover: the next step's H_g)
#pragma unroll
    for (int q = 0; q < SD_RPT; ++q) {
      const unsigned off = ((base | dep[q]) * (unsigned)a.B + (unsigned)(c0 + c)) * (unsigned)EB;
      S_STORE16((unsigned char*)a.out2 + off, s[q] + dg[q] * z[q]);
    }
  }
}

constexpr size_t SD_LDS = (size_t)2 * 65536;   // two 64 KB tile buffers (2^9 rows x 16 complex64 columns = 2^8 rows x 16 complex128 columns)
template <class T>
void schain_dma_launch(const SApply<T>& a, SDmaWalk wk, int nblk, hipStream_t st) {
  (void)hipFuncSetAttribute((const void*)schain_dma_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SD_LDS);
  hipLaunchKernelGGL(schain_dma_kernel<T>, dim3((unsigned)nblk), dim3(STHREADS), SD_LDS, st, a, wk);
}

// ---- chain form of the reverse sweep ----------------------------------------------------------------------------------------------
// zbar_j = s_j H zbar_{j-1} (s_j = i dt / j), lambda_t = lambda_{t+1} + sum_j zbar_j, overlaps <zbar_{j-1}, P_m z_j> per stage j and
// control string m.  Same walk A B | B A | ...: the pass that finishes stage j on group g (out = s_j (prev + H_g y), acc = acc_in +
// out, stage-j overlaps of g's strings against zo = z_j) rewrites its tile with zbar_j and starts stage j + 1 there (out2 = D zbar_j
// + H_g zbar_j; stage-(j+1) overlaps of g's strings against zo2 = z_{j+1}, weighted ovl2_scale = j / (j+1) into the SAME sums, which
// soverlap_finish_kernel then scales by dt / j).  8 state-units per stage instead of 3 + 6, m + 1 walks instead of 2 m.
template <class T>
__global__ void __launch_bounds__(STHREADS) srchain_dma_kernel(SApply<T> a, SDmaWalk wk) {
  constexpr int NSTORE = 3 * SD_RPT;
  typedef typename dq::Cx<T>::type T2;
  typedef SDm<T> SD;
  typedef typename SD::V V;
  constexpr int ROWS = 1 << SD::P, RSTEP = STHREADS / SD::TPR, EB = SD::EB;
  static_assert(ROWS / RSTEP == SD_RPT, "4 rows per thread in both forms");
  extern __shared__ __align__(16) unsigned char sd_raw[];
  T2* tiles = (T2*)sd_raw;  // [2][512 * 16]
  const unsigned full = (a.n >= 32) ? 0xffffffffu : ((1u << a.n) - 1u);
  unsigned xl[SD_OVL_T];
  T cr[SD_OVL_T];
  unsigned omask = 0;
  T px[SD_OVL_T], py[SD_OVL_T];
#pragma unroll
  for (int m = 0; m < SD_OVL_T; ++m) {
    xl[m] = 0; cr[m] = (T)0; px[m] = (T)0; py[m] = (T)0;
    if (m < a.nterms) {
      const DTerm t = a.terms[m];
      xl[m] = __builtin_amdgcn_readfirstlane(t.xloc * (unsigned)(SBC * EB));
      cr[m] = sd_uniform(a.coef[t.slot].x);
      if (a.omap[m] >= 0) omask |= 1u << m;
    }
  }
  omask = __builtin_amdgcn_readfirstlane(omask);
  const bool has_diag = a.diag != nullptr;  // uniform
  const int c = (threadIdx.x % SD::TPR) * SD::CPT;
  const int rbase = threadIdx.x / SD::TPR;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned dep[SD_RPT];
#pragma unroll
  for (int q = 0; q < SD_RPT; ++q) dep[q] = deposit_bits((unsigned)(rbase + q * RSTEP), a.lmask);
  const unsigned lds0 = (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) void*)tiles;
  auto issue = [&](unsigned u, int buf) {
    const unsigned base = deposit_bits(wk.tile(u), full & ~a.lmask);
    const int c0 = (int)wk.chunk(u) * SBC;
#pragma unroll
    for (int q = 0; q < SD_RPT; ++q) {
      const unsigned off = ((base | dep[q]) * (unsigned)a.B + (unsigned)(c0 + c)) * (unsigned)EB;
      const unsigned ld = lds0 + (unsigned)buf * (ROWS * SBC * EB) + (unsigned)(wave * (64 / SD::TPR) + q * RSTEP) * (SBC * EB);
      S_DMA16(a.y, off, ld);
    }
  };
  const unsigned ra = (unsigned)(rbase * SBC + c) * (unsigned)EB;
  unsigned u = blockIdx.x;
  if (u < wk.nunits) issue(u, 0);
  int buf = 0;
  bool first = true;
  for (; u < wk.nunits; u += gridDim.x, buf ^= 1, first = false) {
    if (first) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NSTORE) : "memory");
    const unsigned base = deposit_bits(wk.tile(u), full & ~a.lmask);
    const int c0 = (int)wk.chunk(u) * SBC;
    V s[SD_RPT], ac[SD_RPT], zo[SD_RPT];
#pragma unroll
    for (int q = 0; q < SD_RPT; ++q) {
      const unsigned i = base | dep[q];
      const unsigned off = (i * (unsigned)a.B + (unsigned)(c0 + c)) * (unsigned)EB;   // byte offset: the same for every operand
      S_GLOAD16(zo[q], a.zo, off);
      S_GLOAD16(s[q], a.prev, off);
      S_GLOAD16(ac[q], a.acc_in, off);
    }
    const bool more = u + gridDim.x < wk.nunits;  // uniform
    if (more) issue(u + gridDim.x, buf ^ 1);
    unsigned char* tb = (unsigned char*)(tiles + (size_t)buf * (ROWS * SBC));
    if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SD_RPT) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int q = 0; q < SD_RPT; ++q) {
      S_PIN(zo[q]); S_PIN(s[q]); S_PIN(ac[q]);
    }
    // stage j on the tile of zbar_{j-1}: H_g y on top of prev, overlaps against z_j
#pragma unroll
    for (int m = 0; m < SD_OVL_T; ++m)
      if (m < a.nterms) {  // uniform
        asm volatile("" : "+s"(xl[m]));
#pragma unroll
        for (int q = 0; q < SD_RPT; ++q) {
          const V v = *(const V*)(tb + ((ra | (unsigned)(q * RSTEP * SBC * EB)) ^ xl[m]));
          s[q] += cr[m] * v;
          if (omask >> m & 1u) {
            px[m] += sd_cre(v, zo[q]);
            py[m] += sd_cim(v, zo[q]);
          }
        }
      }
    // the next stage's cotangent rows take the registers of this stage's (requested now, needed after the tile is rewritten; the
    // eight stores below are younger, the next tile's DMA older: by then it has had the whole first stage to arrive)
    V zn[SD_RPT];
    T dg[SD_RPT];
#pragma unroll
    for (int q = 0; q < SD_RPT; ++q) {
      S_GLOAD16(zn[q], a.zo2, ((base | dep[q]) * (unsigned)a.B + (unsigned)(c0 + c)) * (unsigned)EB);
      dg[q] = (T)0;
      if (has_diag) sd_gload_real(dg[q], a.diag, ((base | dep[q])) * (unsigned)sizeof(T));
    }
    V z[SD_RPT];
#pragma unroll
    for (int q = 0; q < SD_RPT; ++q) {
      const unsigned off = ((base | dep[q]) * (unsigned)a.B + (unsigned)(c0 + c)) * (unsigned)EB;
      z[q] = sd_rot(a.sr, a.si, s[q]);
      S_STORE16((unsigned char*)a.out + off, z[q]);
      S_STORE16((unsigned char*)a.acc + off, ac[q] + z[q]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
    for (int q = 0; q < SD_RPT; ++q) *(V*)(tb + (ra | (unsigned)(q * RSTEP * SBC * EB))) = z[q];
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // stage j + 1 starts on the same tile, now zbar_j
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * SD_RPT) : "memory");
#pragma unroll
    for (int q = 0; q < SD_RPT; ++q) {
      S_PIN(zn[q]);
      if (has_diag) S_PIN(dg[q]);
      s[q] = dg[q] * z[q]; zn[q] *= a.ovl2_scale;
    }
#pragma unroll
    for (int m = 0; m < SD_OVL_T; ++m)
      if (m < a.nterms) {  // uniform
        asm volatile("" : "+s"(xl[m]));
#pragma unroll
        for (int q = 0; q < SD_RPT; ++q) {
          const V v = *(const V*)(tb + ((ra | (unsigned)(q * RSTEP * SBC * EB)) ^ xl[m]));
          s[q] += cr[m] * v;
          if (omask >> m & 1u) {
            px[m] += sd_cre(v, zn[q]);
            py[m] += sd_cim(v, zn[q]);
          }
        }
      }
#pragma unroll
    for (int q = 0; q < SD_RPT; ++q) {
      const unsigned off = ((base | dep[q]) * (unsigned)a.B + (unsigned)(c0 + c)) * (unsigned)EB;
      S_STORE16((unsigned char*)a.out2 + off, s[q]);
    }
  }
  const int lane = threadIdx.x & 63;
  const size_t blk = (size_t)blockIdx.x * (STHREADS / 64) + wave;
#pragma unroll
  for (int m = 0; m < SD_OVL_T; ++m)
    if (omask >> m & 1u) {
      const double tx = s_wave_sum<T>(px[m]), ty = s_wave_sum<T>(py[m]);
      if (lane == 0) {
        const int mm = a.omap[m];
        a.part[(blk * a.n_ovl + mm) * 2 + 0] = tx;
        a.part[(blk * a.n_ovl + mm) * 2 + 1] = ty;
      }
    }
}

template <class T>
void srchain_dma_launch(const SApply<T>& a, SDmaWalk wk, int nblk, hipStream_t st) {
  (void)hipFuncSetAttribute((const void*)srchain_dma_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SD_LDS);
  hipLaunchKernelGGL(srchain_dma_kernel<T>, dim3((unsigned)nblk), dim3(STHREADS), SD_LDS, st, a, wk);
}

// can this pass take the DMA form?  (whole tiles of the dtype's size, batch a multiple of 16, states below 4 GB: 32-bit byte offsets)
template <class T>
bool sdma_ok(int p, int B, int nterms, int mode, int n) {
  return mode != 0 && p == SDm<T>::P && (B % SBC) == 0 && nterms <= SMAXT &&
         ((double)B * (double)SDm<T>::EB * (double)(1ull << n) < 4294967296.0);
}
inline int sdma_blocks(unsigned nunits, int n_cu) {  // one persistent workgroup per CU (128 KB of LDS), equal rounds
  unsigned nb = (unsigned)n_cu < nunits ? (unsigned)n_cu : nunits;
  const unsigned rounds = (nunits + nb - 1) / nb;
  nb = (nunits + rounds - 1) / rounds;
  return (int)(nb < 1 ? 1 : nb);
}
template <class T, int F>
void sapply_dma_launch_f(const SApply<T>& a, SDmaWalk wk, int nblk, hipStream_t st, bool simple) {
  const size_t lds = SD_LDS;
  if (simple) {
    (void)hipFuncSetAttribute((const void*)sapply_dma_kernel<T, F, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((sapply_dma_kernel<T, F, true>), dim3((unsigned)nblk), dim3(STHREADS), lds, st, a, wk);
  } else if constexpr (!(sizeof(T) == sizeof(double) && F == 15)) {  // (that one form is not built: see apply_groups)
    (void)hipFuncSetAttribute((const void*)sapply_dma_kernel<T, F, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((sapply_dma_kernel<T, F, false>), dim3((unsigned)nblk), dim3(STHREADS), lds, st, a, wk);
  }
}
template <class T, int F>
void sapply_dma_launch_ovl(const SApply<T>& a, SDmaWalk wk, int nblk, hipStream_t st) {
  const size_t lds = SD_LDS;
  (void)hipFuncSetAttribute((const void*)sapply_dma_kernel<T, F, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((sapply_dma_kernel<T, F, true, true>), dim3((unsigned)nblk), dim3(STHREADS), lds, st, a, wk);
}
inline bool sapply_ovl_flags_ok(int f);
template <class T>
void sapply_dma_launch(const SApply<T>& a, SDmaWalk wk, int nblk, hipStream_t st, bool simple) {
  const int f = (a.prev ? 1 : 0) | (a.add ? 2 : 0) | (a.acc ? 4 : 0) | (a.diag ? 8 : 0);
  if (a.zo) {  // fused overlaps: the caller has checked simple && sapply_ovl_flags_ok(f)
    if (!a.out) {   // (accumulate only: built for the operand set of the reverse chain's last stage)
      if (f != 5) sfail(DQ_ERR_STATE, "internal: a pass without `out` in an operand set that is not built");
      sapply_dma_launch_ovl<T, 5 | 16>(a, wk, nblk, st);
      return;
    }
    switch (f) {
      case 0: sapply_dma_launch_ovl<T, 0>(a, wk, nblk, st); break;
      case 1: sapply_dma_launch_ovl<T, 1>(a, wk, nblk, st); break;
      case 4: sapply_dma_launch_ovl<T, 4>(a, wk, nblk, st); break;
      case 5: sapply_dma_launch_ovl<T, 5>(a, wk, nblk, st); break;
      case 8: sapply_dma_launch_ovl<T, 8>(a, wk, nblk, st); break;
      default: sapply_dma_launch_ovl<T, 12>(a, wk, nblk, st); break;
    }
    return;
  }
  switch (f) {
    case 0: sapply_dma_launch_f<T, 0>(a, wk, nblk, st, simple); break;
    case 1: sapply_dma_launch_f<T, 1>(a, wk, nblk, st, simple); break;
    case 2: sapply_dma_launch_f<T, 2>(a, wk, nblk, st, simple); break;
    case 3: sapply_dma_launch_f<T, 3>(a, wk, nblk, st, simple); break;
    case 4: sapply_dma_launch_f<T, 4>(a, wk, nblk, st, simple); break;
    case 5: sapply_dma_launch_f<T, 5>(a, wk, nblk, st, simple); break;
    case 6: sapply_dma_launch_f<T, 6>(a, wk, nblk, st, simple); break;
    case 7: sapply_dma_launch_f<T, 7>(a, wk, nblk, st, simple); break;
    case 8: sapply_dma_launch_f<T, 8>(a, wk, nblk, st, simple); break;
    case 9: sapply_dma_launch_f<T, 9>(a, wk, nblk, st, simple); break;
    case 10: sapply_dma_launch_f<T, 10>(a, wk, nblk, st, simple); break;
    case 11: sapply_dma_launch_f<T, 11>(a, wk, nblk, st, simple); break;
    case 12: sapply_dma_launch_f<T, 12>(a, wk, nblk, st, simple); break;
    case 13: sapply_dma_launch_f<T, 13>(a, wk, nblk, st, simple); break;
    case 14: sapply_dma_launch_f<T, 14>(a, wk, nblk, st, simple); break;
    default: sapply_dma_launch_f<T, 15>(a, wk, nblk, st, simple); break;
  }
}
template <class T, int F>
void sapply_launch_f(const SApply<T>& a, dim3 grid, hipStream_t st) {
  if ((a.B & 1) == 0) hipLaunchKernelGGL((sapply_kernel<T, F, 2>), grid, dim3(STHREADS), 0, st, a);
  else hipLaunchKernelGGL((sapply_kernel<T, F, 1>), grid, dim3(STHREADS), 0, st, a);
}
template <class T, int F>
void sapply_launch_ovl(const SApply<T>& a, dim3 grid, hipStream_t st) {
  if ((a.B & 1) == 0) hipLaunchKernelGGL((sapply_kernel<T, F, 2, true>), grid, dim3(STHREADS), 0, st, a);
  else hipLaunchKernelGGL((sapply_kernel<T, F, 1, true>), grid, dim3(STHREADS), 0, st, a);
}
// operand sets of the reverse-mode apply (no addend): first / middle / last group, with or without the diagonal
inline bool sapply_ovl_flags_ok(int f) { return f == 0 || f == 1 || f == 4 || f == 5 || f == 8 || f == 12; }
template <class T>
void sapply_launch(const SApply<T>& a, dim3 grid, hipStream_t st) {
  const int f = (a.prev ? 1 : 0) | (a.add ? 2 : 0) | (a.acc ? 4 : 0) | (a.diag ? 8 : 0);
  if (a.n_ovl > 0) {
    switch (f) {
      case 0: sapply_launch_ovl<T, 0>(a, grid, st); return;
      case 1: sapply_launch_ovl<T, 1>(a, grid, st); return;
      case 4: sapply_launch_ovl<T, 4>(a, grid, st); return;
      case 5: sapply_launch_ovl<T, 5>(a, grid, st); return;
      case 8: sapply_launch_ovl<T, 8>(a, grid, st); return;
      case 12: sapply_launch_ovl<T, 12>(a, grid, st); return;
      default: sfail(DQ_ERR_HIP, "fused overlap pass asked for an operand set it is not built for");
    }
  }
  switch (f) {
    case 0: sapply_launch_f<T, 0>(a, grid, st); break;
    case 1: sapply_launch_f<T, 1>(a, grid, st); break;
    case 2: sapply_launch_f<T, 2>(a, grid, st); break;
    case 3: sapply_launch_f<T, 3>(a, grid, st); break;
    case 4: sapply_launch_f<T, 4>(a, grid, st); break;
    case 5: sapply_launch_f<T, 5>(a, grid, st); break;
    case 6: sapply_launch_f<T, 6>(a, grid, st); break;
    case 7: sapply_launch_f<T, 7>(a, grid, st); break;
    case 8: sapply_launch_f<T, 8>(a, grid, st); break;
    case 9: sapply_launch_f<T, 9>(a, grid, st); break;
    case 10: sapply_launch_f<T, 10>(a, grid, st); break;
    case 11: sapply_launch_f<T, 11>(a, grid, st); break;
    case 12: sapply_launch_f<T, 12>(a, grid, st); break;
    case 13: sapply_launch_f<T, 13>(a, grid, st); break;
    case 14: sapply_launch_f<T, 14>(a, grid, st); break;
    default: sapply_launch_f<T, 15>(a, grid, st); break;
  }
}

template <class T>
struct SOverlap {
  typedef typename dq::Cx<T>::type T2;
  const T2* za;   // cotangent (read at its own row)
  const T2* zb;   // state (tile in LDS, read at row ^ x)
  const DTerm* terms;
  int nterms;
  int n, p;
  unsigned lmask;
  int B;
  unsigned ntiles;  // 2^(n-p) row tiles
  int tpw;          // tiles walked by one workgroup
  double* part;     // [nblocks][nterms][2]
};

// part[blk][m] = sum over the workgroup's tiles of conj(za[i][c]) * sign_m(i) * zb[i ^ x_m][c]
// A workgroup walks `tpw` tiles; per-string partial sums stay in registers (T) across them, so the wave/block
// reduction (fp64 across waves and workgroups, fixed order) is paid once per workgroup, not once per tile.
template <class T, int V>
__global__ void __launch_bounds__(STHREADS) soverlap_kernel(SOverlap<T> a) {
  typedef typename dq::Cx<T>::type T2;
  typedef CVec<T2, V> VT;
  constexpr int PMAX = LocalBits<T>::P;
  constexpr int CPL = SBC / V;
  constexpr int RSTEP = STHREADS / CPL;
  constexpr int RPT = ((1 << PMAX) + RSTEP - 1) / RSTEP;
  constexpr int TCH = 12;  // strings per register chunk
  __shared__ T2 tile[(1 << PMAX) * SBC];
  __shared__ unsigned rowidx[1 << PMAX];
  __shared__ DTerm tt[SMAXT];
  __shared__ double red[2][TCH][STHREADS / 64];
  const int R = 1 << a.p;
  const unsigned full = (a.n >= 32) ? 0xffffffffu : ((1u << a.n) - 1u);
  const int c0 = blockIdx.y * SBC;
  const int c = (threadIdx.x % CPL) * V;
  const int rbase = threadIdx.x / CPL;
  const bool col_ok = (c0 + c) < a.B;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int blk = blockIdx.y * gridDim.x + blockIdx.x;
  VT zero;
#pragma unroll
  for (int w = 0; w < V; ++w) { zero.v[w].x = 0; zero.v[w].y = 0; }
  for (int m = threadIdx.x; m < a.nterms; m += STHREADS) tt[m] = a.terms[m];
  for (int m0 = 0; m0 < a.nterms; m0 += TCH) {
    const int nm = min(TCH, a.nterms - m0);
    T px[TCH], py[TCH];
#pragma unroll
    for (int mm = 0; mm < TCH; ++mm) { px[mm] = 0; py[mm] = 0; }
    for (int ti = 0; ti < a.tpw; ++ti) {
      const unsigned tix = blockIdx.x * a.tpw + ti;
      if (tix >= a.ntiles) break;   // uniform
      const unsigned base = deposit_bits(tix, full & ~a.lmask);
      __syncthreads();              // previous tile fully consumed
      for (int r = threadIdx.x; r < R; r += STHREADS) rowidx[r] = base | deposit_bits((unsigned)r, a.lmask);
      __syncthreads();
      for (int r = rbase; r < R; r += RSTEP) {
        VT v = zero;
        if (col_ok) v = *(const VT*)(a.zb + (size_t)rowidx[r] * a.B + c0 + c);
        *(VT*)(tile + r * SBC + c) = v;
      }
      // the cotangent rows of this thread stay in registers for all strings (one global read per pass)
      VT u[RPT];
      unsigned ui[RPT];
#pragma unroll
      for (int q = 0; q < RPT; ++q) {
        const int r = min(rbase + q * RSTEP, R - 1);
        ui[q] = rowidx[r];
        u[q] = zero;
        if (col_ok && rbase + q * RSTEP < R) u[q] = *(const VT*)(a.za + (size_t)ui[q] * a.B + c0 + c);
      }
      __syncthreads();
#pragma unroll
      for (int mm = 0; mm < TCH; ++mm)
        if (mm < nm) {               // uniform
          const DTerm t = tt[m0 + mm];
#pragma unroll
          for (int q = 0; q < RPT; ++q) {
            const int r = rbase + q * RSTEP;
            if (r < R) {
              const VT v = *(const VT*)(tile + (r ^ t.xloc) * SBC + c);
              T qx = 0, qy = 0;
#pragma unroll
              for (int w = 0; w < V; ++w) {
                qx += u[q].v[w].x * v.v[w].x + u[q].v[w].y * v.v[w].y;   // conj(u) * v
                qy += u[q].v[w].x * v.v[w].y - u[q].v[w].y * v.v[w].x;
              }
              if (t.zmask != 0 && (__popc((ui[q] ^ t.xmask) & t.zmask) & 1)) { qx = -qx; qy = -qy; }
              px[mm] += qx; py[mm] += qy;
            }
          }
        }
    }
    // wave reduction in T (32-bit shuffles for complex64), fp64 across waves / workgroups
#pragma unroll
    for (int mm = 0; mm < TCH; ++mm)
      if (mm < nm) {
        T vx = px[mm], vy = py[mm];
        for (int off = 32; off > 0; off >>= 1) { vx += __shfl_down(vx, off, 64); vy += __shfl_down(vy, off, 64); }
        if (lane == 0) { red[0][mm][wave] = (double)vx; red[1][mm][wave] = (double)vy; }
      }
    __syncthreads();
    if ((int)threadIdx.x < nm) {
      double tx = 0, ty = 0;
      for (int w = 0; w < STHREADS / 64; ++w) { tx += red[0][threadIdx.x][w]; ty += red[1][threadIdx.x][w]; }
      a.part[((size_t)blk * a.nterms + m0 + threadIdx.x) * 2 + 0] = tx;
      a.part[((size_t)blk * a.nterms + m0 + threadIdx.x) * 2 + 1] = ty;
    }
    __syncthreads();
  }
}

// part[blk][m] as soverlap_kernel (blk = blockIdx.x; at most SOVL_TCH strings).  The cotangent rows of tile t + 1 are requested one
// tile ahead as well (registers), so neither operand's latency is exposed.  Both dtypes since round 5 (the shapes of SDm<T>: a lane's
// 16 bytes are two complex64 columns or one complex128 column of a row).
template <class T>
__global__ void __launch_bounds__(STHREADS) soverlap_dma_kernel(SOverlap<T> a, SDmaWalk wk) {
  typedef typename dq::Cx<T>::type T2;
  typedef SDm<T> SD;
  typedef typename SD::V V;
  constexpr int ROWS = 1 << SD::P, RSTEP = STHREADS / SD::TPR, EB = SD::EB;
  static_assert(ROWS / RSTEP == SD_RPT, "4 rows per thread in both forms");
  extern __shared__ __align__(16) unsigned char sd_raw[];
  __shared__ DTerm tt[SOVL_TCH];
  __shared__ double red[2][SOVL_TCH][STHREADS / 64];
  const unsigned full = (a.n >= 32) ? 0xffffffffu : ((1u << a.n) - 1u);
  for (int m = threadIdx.x; m < a.nterms; m += STHREADS) tt[m] = a.terms[m];
  const int c = (threadIdx.x % SD::TPR) * SD::CPT;
  const int rbase = threadIdx.x / SD::TPR;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  unsigned dep[SD_RPT];
#pragma unroll
  for (int q = 0; q < SD_RPT; ++q) dep[q] = deposit_bits((unsigned)(rbase + q * RSTEP), a.lmask);
  const unsigned lds0 = (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) void*)sd_raw;
  auto issue = [&](unsigned u, int buf) {
    const unsigned base = deposit_bits(wk.tile(u), full & ~a.lmask);
    const int c0 = (int)wk.chunk(u) * SBC;
#pragma unroll
    for (int q = 0; q < SD_RPT; ++q) {
      const unsigned off = ((base | dep[q]) * (unsigned)a.B + (unsigned)(c0 + c)) * (unsigned)EB;
      const unsigned ld = lds0 + (unsigned)buf * (ROWS * SBC * EB) + (unsigned)(wave * (64 / SD::TPR) + q * RSTEP) * (SBC * EB);
      S_DMA16(a.zb, off, ld);
    }
  };
  auto load_u = [&](unsigned u, V* dst) {
    const unsigned base = deposit_bits(wk.tile(u), full & ~a.lmask);
    const int c0 = (int)wk.chunk(u) * SBC;
#pragma unroll
    for (int q = 0; q < SD_RPT; ++q) dst[q] = *(const V*)(a.za + (size_t)(base | dep[q]) * a.B + c0 + c);
  };
  T px[SOVL_TCH], py[SOVL_TCH];
#pragma unroll
  for (int mm = 0; mm < SOVL_TCH; ++mm) { px[mm] = 0; py[mm] = 0; }
  V un[SD_RPT];
  unsigned u = blockIdx.x;
  if (u < wk.nunits) { issue(u, 0); load_u(u, un); }
  int buf = 0;
  for (; u < wk.nunits; u += gridDim.x, buf ^= 1) {
    V uc[SD_RPT];
#pragma unroll
    for (int q = 0; q < SD_RPT; ++q) uc[q] = un[q];  // (waits for this tile's cotangent rows: requested a tile ago)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");  // tile u has landed in every wave's pieces
    if (u + gridDim.x < wk.nunits) { issue(u + gridDim.x, buf ^ 1); load_u(u + gridDim.x, un); }
    const T2* tile = (const T2*)sd_raw + (size_t)buf * (ROWS * SBC);
    const unsigned base = deposit_bits(wk.tile(u), full & ~a.lmask);
#pragma unroll
    for (int mm = 0; mm < SOVL_TCH; ++mm)
      if (mm < a.nterms) {  // uniform
        const DTerm t = tt[mm];
#pragma unroll
        for (int q = 0; q < SD_RPT; ++q) {
          const int r = rbase + q * RSTEP;
          const V v = *(const V*)(tile + (r ^ t.xloc) * SBC + c);
          T qx = sd_cre(uc[q], v), qy = sd_cim(uc[q], v);   // conj(u) * v over the lane's complex numbers
          if (t.zmask != 0 && (__popc(((base | dep[q]) ^ t.xmask) & t.zmask) & 1)) { qx = -qx; qy = -qy; }
          px[mm] += qx; py[mm] += qy;
        }
      }
  }
#pragma unroll
  for (int mm = 0; mm < SOVL_TCH; ++mm)
    if (mm < a.nterms) {
      T vx = px[mm], vy = py[mm];
      for (int off = 32; off > 0; off >>= 1) { vx += __shfl_down(vx, off, 64); vy += __shfl_down(vy, off, 64); }
      if (lane == 0) { red[0][mm][wave] = (double)vx; red[1][mm][wave] = (double)vy; }
    }
  __syncthreads();
  if ((int)threadIdx.x < a.nterms) {
    double tx = 0, ty = 0;
    for (int w = 0; w < STHREADS / 64; ++w) { tx += red[0][threadIdx.x][w]; ty += red[1][threadIdx.x][w]; }
    a.part[((size_t)blockIdx.x * a.nterms + threadIdx.x) * 2 + 0] = tx;
    a.part[((size_t)blockIdx.x * a.nterms + threadIdx.x) * 2 + 1] = ty;
  }
}

template <class T>
void soverlap_dma_launch(const SOverlap<T>& o, SDmaWalk wk, int nblk, hipStream_t st) {
  const size_t lds = SD_LDS;
  (void)hipFuncSetAttribute((const void*)soverlap_dma_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(soverlap_dma_kernel<T>, dim3((unsigned)nblk), dim3(STHREADS), lds, st, o, wk);
}

// grad[k(slot)] += fac * Im( phase_slot * sum_blk part[blk][m] ) * weight_slot      (fixed-order sum: deterministic)
__global__ void __launch_bounds__(256) soverlap_finish_kernel(const double* part, int nblk, int nterms, const int* slots,
                                                              const int* owner, const double* wre, const double* wim,
                                                              double fac, double* grad_row) {
  // one block per string; fixed strided order + tree -> deterministic
  __shared__ double rx[256], ry[256];
  const int m = blockIdx.x;
  double sx = 0, sy = 0;
  for (int b = threadIdx.x; b < nblk; b += 256) { sx += part[((size_t)b * nterms + m) * 2]; sy += part[((size_t)b * nterms + m) * 2 + 1]; }
  rx[threadIdx.x] = sx; ry[threadIdx.x] = sy;
  __syncthreads();
  for (int h = 128; h > 0; h >>= 1) {
    if ((int)threadIdx.x < h) { rx[threadIdx.x] += rx[threadIdx.x + h]; ry[threadIdx.x] += ry[threadIdx.x + h]; }
    __syncthreads();
  }
  if (threadIdx.x != 0) return;
  sx = rx[0]; sy = ry[0];
  const int s = slots[m];
  const int k = owner[s];
  if (k < 0) return;
  // (wre + i wim) = weight * i^{nY};  Im((wre + i wim)(sx + i sy))
  const double im = wre[s] * sy + wim[s] * sx;
  atomicAdd(&grad_row[k], fac * im);  // few adds per step; order fixed within a launch, launches are stream-ordered
}

// D[i] = sum_m w_m (-1)^{popc(i & z_m)}
template <class T>
__global__ void sdiag_build_kernel(T* D, unsigned d, const unsigned* z, const double* w, int nt) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= d) return;
  double s = 0.0;
  for (int m = 0; m < nt; ++m) s += (__popc(i & z[m]) & 1) ? -w[m] : w[m];
  D[i] = (T)s;
}

// D_t[i] = D0[i] + sum_k u[k] Dk[k][i]
template <class T>
__global__ void sdiag_step_kernel(T* D, const T* D0, const T* Dk, const double* u, const int* kidx, int nk, unsigned d) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= d) return;
  T s = D0 ? D0[i] : (T)0;
  for (int j = 0; j < nk; ++j) s += (T)u[kidx[j]] * Dk[(size_t)j * d + i];
  D[i] = s;
}

// per-step complex coefficients: coef[slot] = (wre + i wim)[slot] * (owner < 0 ? 1 : u[owner])
template <class T>
__global__ void scoef_kernel(typename dq::Cx<T>::type* coef, const double* wre, const double* wim, const int* owner,
                             const double* u, int nslots) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nslots) return;
  const double f = owner[s] < 0 ? 1.0 : (u ? u[owner[s]] : 0.0);
  coef[s].x = (T)(wre[s] * f);
  coef[s].y = (T)(wim[s] * f);
}

struct HostTerm { int owner; unsigned x, z; double w; };  // owner: -1 drift, 0..K-1 control, -2 observable
struct Group {
  unsigned lmask = 0;
  int p = 0;
  std::vector<DTerm> apply;   // off-diagonal strings applied in this pass
  std::vector<DTerm> ovl;     // strings whose overlaps are taken in this pass (control strings, incl. diagonal ones)
  bool simple = false;        // every applied string is a pure X-string (z = 0: real coefficient, no sign) and they are few
  bool ovl_fusable = false;   // simple, and every overlap string is one of the (at most SD_OVL_T) applied strings: DMA-form fusion
  SBuf d_omap;                // applied string -> index among the overlap strings, or -1
  int last_nblk = 0;          // workgroup-wave partial sums the last fused pass wrote per string
  SBuf d_apply, d_ovl, d_ovl_slots;
};

}  // namespace

struct dq_sctx {
  int dtype = DQ_C64;
  virtual ~dq_sctx() {}
};

namespace {

template <class T>
struct SCtx : dq_sctx {
  typedef typename dq::Cx<T>::type T2;
  int n = 0, K = 0, dev = 0;
  unsigned d = 0;
  hipStream_t st = nullptr;
  bool have_terms = false, have_obs = false;
  std::vector<HostTerm> hterms;     // slots of H (drift + controls), then observable slots
  int n_hslots = 0, n_slots = 0;
  std::vector<Group> hgroups, mgroups;
  SBuf d_wre, d_wim, d_owner, d_coef, d_coefM;
  SBuf D0, Dk, Dkidx, Dt, DM;       // diagonals: drift, per-control, step, observable
  // Handover between STEPS (round 5): the pass that finishes step t's Horner chain also starts step t + 1's (it holds psi_{t+1}'s tile:
  // out2 = D_{t+1} psi_{t+1} + H_g^{t+1} psi_{t+1}) -- a 5-unit link where a 4-unit finish and a 2-unit start pass ran (the start pass
  // at 3.1 TB/s: 10.9 us per tile against 16 us for 2.5 x the bytes).  Step t + 1's coefficients and diagonal are made into the second
  // set before step t's last pass and the sets change places behind it.
  SBuf d_coef2, Dt2;
  int step_handover = 1;
  int reverse_keep_last = 0;   // 1: the reverse chain's last stage stores zbar_m as well (nobody reads it: the form before round 5)
  void swap_step_sets() { std::swap(d_coef.p, d_coef2.p); std::swap(d_coef.bytes, d_coef2.bytes); std::swap(Dt.p, Dt2.p); std::swap(Dt.bytes, Dt2.bytes); }
  std::vector<int> diag_controls;
  bool has_D0 = false, has_DM = false;
  double wsum0 = 0;                 // sum |w| of the drift strings
  std::vector<double> wsumk;        // per control
  double tol = 0, checkpoint_bytes = 0;
  int force_degree = 0;
  // The step polynomial in Horner form z_m = psi, z_{j-1} = psi + rho[j] A z_j: rho[j] = 1 / j is Taylor's; the Chebyshev-economized
  // one (dq_internal.h XsPoly: sum g[j] A^j / j!, one to four degrees fewer at the same sup-norm error on the spectrum i [-theta,
  // theta]), normalised to constant term 1 (p / g[0]: |1 - g[0]| is below the truncation bound, and the bound asked of the cut is
  // HALF the tolerance so that both together stay inside it), has rho[j] = g[j] / (j g[j-1]).  The reverse mode of the recursion carries
  // the same factors (zbar_j = conj(rho[j]) A^H zbar_{j-1}; overlaps weighted dt rho[j]).
  int economize = 1;
  // the polynomial's error is the same function of the spectrum at every step and may add up coherently over a chain: the per-step
  // bound is min(tol, chain_tol / steps) (steps: of the call, or plan_steps when that is larger)
  double chain_tol = sizeof(T) == 8 ? 1e-11 : 2e-5;
  double last_taylor_degree = 0;
  std::vector<double> rho;
  double last_degree = 0, last_theta = 0, last_groups = 0, last_cstride = 0, last_passes = 0, last_bytes = 0, last_dma_passes = 0;
  SBuf u_dev, grad_dev, e_dev, part, tmp[2], lam[2], psi[2], zbar[2];
  std::vector<SBuf> zs, ck, oparts;  // oparts: per-group partial sums of the fused overlap passes
  double host_checkpoint_bytes = 0;  // > 0: cap of the pinned host memory the kept states may take (0: 60 % of what the process may still use)
  int host_checkpoints = 1;  // 1: the kept states of the two-level plan may live in pinned host memory when the HBM holds one segment only
  double last_host_ck = 0;
  std::vector<SHostBuf> ckh;
  int two_level = 1;      // 1: when the states of a chain do not all fit, keep every s-th and rebuild a segment at a time WITH its Horner iterates (below)
  double last_two_level = 0;
  std::vector<SBuf> segpsi;
  std::vector<std::vector<SBuf>> segz;
  int plan_steps = 0;     // checkpoint stride as for a chain of this many steps (a timed slice of a long chain then pays the long chain's reversals)
  int interleave_groups = 2;  // v > 0: two groups of single-qubit flips share the low and the high qubits (set before dq_sctx_set_terms)
  int reverse_chain = 1;  // 1: the reverse sweep in the chain form too (srchain_dma_kernel) when chain applies and the overlaps are fused
  int chain = 1;          // 1: Horner chains over two pure X-string groups in the chain form (schain_dma_kernel); 0: one application = one pass per group
  int fuse_overlaps = 0;  // 1: always (the one-tile kernels pay a workgroup per CU for it); 0: only in the DMA form, where it is free; -1: never
  int dma = 1, n_cu = 256;  // dma: LDS-DMA double-buffered passes where the shapes allow (complex64, 2^9-row tiles, batch % 16 == 0); 0 = off
  // profile of the streaming passes
  bool prof_on = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> recs;
  std::vector<double> rec_bytes;
  double prof_ms = 0, prof_bytes = 0;
  long long prof_launches = 0;

  SCtx() {
    dtype = sizeof(T) == 8 ? DQ_C128 : DQ_C64;
    tol = sizeof(T) == 8 ? 1.1102230246251565e-16 : 5.9604644775390625e-08;
  }
  // host checkpoints move on a stream of their own: down from a staged copy while the chain goes on, up one segment ahead
  hipStream_t copy_st = nullptr;
  hipEvent_t ev_staged = nullptr, ev_down = nullptr, ev_up = nullptr, ev_used = nullptr;
  SBuf stage;
  void copy_stream_ready() {
    if (copy_st) return;
    SHIP(hipStreamCreateWithFlags(&copy_st, hipStreamNonBlocking));
    for (hipEvent_t* e : {&ev_staged, &ev_down, &ev_up, &ev_used}) SHIP(hipEventCreateWithFlags(e, hipEventDisableTiming));
  }
  ~SCtx() override {
    for (auto& r : recs) { (void)hipEventDestroy(r.first); (void)hipEventDestroy(r.second); }
    if (copy_st) {
      (void)hipStreamSynchronize(copy_st);
      for (hipEvent_t e : {ev_staged, ev_down, ev_up, ev_used}) if (e) (void)hipEventDestroy(e);
      (void)hipStreamDestroy(copy_st);
    }
  }
  size_t state_bytes(int B) const { return (size_t)d * B * sizeof(T2); }

  template <class F>
  void timed(double bytes, F&& f) {
    if (!prof_on) { f(); SHIP(hipGetLastError()); return; }
    hipEvent_t e0, e1;
    SHIP(hipEventCreate(&e0)); SHIP(hipEventCreate(&e1));
    SHIP(hipEventRecord(e0, st));
    f();
    SHIP(hipGetLastError());
    SHIP(hipEventRecord(e1, st));
    recs.push_back({e0, e1});
    rec_bytes.push_back(bytes);
  }
  void collect() {
    if (recs.empty()) return;
    SHIP(hipStreamSynchronize(st));
    for (size_t i = 0; i < recs.size(); ++i) {
      float ms = 0;
      SHIP(hipEventElapsedTime(&ms, recs[i].first, recs[i].second));
      prof_ms += ms; prof_bytes += rec_bytes[i]; prof_launches++;
      (void)hipEventDestroy(recs[i].first); (void)hipEventDestroy(recs[i].second);
    }
    recs.clear(); rec_bytes.clear();
  }
};

unsigned compress_bits(unsigned v, unsigned mask) {  // software pext (host)
  unsigned out = 0, bit = 1;
  while (mask) {
    const unsigned low = mask & (0u - mask);
    if (v & low) out |= bit;
    bit <<= 1;
    mask ^= low;
  }
  return out;
}

int popc(unsigned v) { return __builtin_popcount(v); }

// Partition strings into groups whose flipped qubits fit in p local bits (first fit, widest strings first).
template <class T>
void build_groups(SCtx<T>* c, const std::vector<int>& slots, std::vector<Group>& groups, bool with_overlaps) {
  const int P = std::min(LocalBits<T>::P, c->n);
  groups.clear();
  std::vector<int> offd;
  for (int s : slots) if (c->hterms[s].x != 0) offd.push_back(s);
  std::stable_sort(offd.begin(), offd.end(), [&](int a, int b) { return popc(c->hterms[a].x) > popc(c->hterms[b].x); });
  std::vector<std::vector<int>> members;
  for (int s : offd) {
    const unsigned x = c->hterms[s].x;
    if (popc(x) > P) sfail(DQ_ERR_UNSUPPORTED, "a Pauli string flips more qubits than fit in one LDS tile");
    int g = -1;
    for (size_t i = 0; i < groups.size(); ++i)
      if (popc(groups[i].lmask | x) <= P && (int)members[i].size() < SMAXT) { g = (int)i; break; }
    if (g < 0) { groups.emplace_back(); members.emplace_back(); g = (int)groups.size() - 1; }
    groups[g].lmask |= x;
    members[g].push_back(s);
  }
  if (groups.empty()) { groups.emplace_back(); members.emplace_back(); }
  // Two groups of single-qubit flips (Ising-type inputs): deal the qubits to the groups in turn instead of low qubits to one and high
  // qubits to the other.  A tile of a group that owns all high qubits touches 2^9 different 2 MB regions per operand (its rows lie
  // 2 MB apart) and ran 3.2-3.6 TB/s where the low group's tile (one region) ran 5-6.  Option interleave_groups = v: 0 off, 1 quarters
  // (low/high halves of each group), v >= 2 blocks of v - 1 qubits in turn; n = 18 fwd+adjoint ms per step at B = 512 / 64:
  // off 26.9 / 3.06, quarters 26.5 / 2.96, singles (2, default) 25.3 / 2.95, pairs 26.3 / 2.90, triples 26.9 / 3.00.
  if (c->interleave_groups && groups.size() == 2) {
    bool single = true;
    unsigned used = 0;
    for (int s : offd) { single = single && popc(c->hterms[s].x) == 1; used |= c->hterms[s].x; }
    const int k = popc(used);
    const int nA = (k + 1) / 2, nB = k - nA;
    if (single && nA <= P && (int)offd.size() <= 2 * SMAXT) {
      const int a0 = nA / 2, b0 = (nB + 1) / 2;
      unsigned mA = 0, mB = 0;
      int i = 0;
      for (int q = 0; q < 32; ++q)
        if (used & (1u << q)) {
          bool toA = i < a0 || (i >= a0 + b0 && i < a0 + b0 + (nA - a0));
          if (c->interleave_groups > 1) toA = ((i / (c->interleave_groups - 1)) & 1) == 0;   // blocks of v - 1 qubits dealt in turn
          if (toA && popc(mA) >= nA) toA = false;       // (a group never gets more than its share: tiles stay 2^P rows)
          if (!toA && popc(mB) >= nB) toA = true;
          (toA ? mA : mB) |= 1u << q;
          ++i;
        }
      std::vector<std::vector<int>> mem2(2);
      for (int s : offd) mem2[(c->hterms[s].x & mA) ? 0 : 1].push_back(s);
      if ((int)mem2[0].size() <= SMAXT && (int)mem2[1].size() <= SMAXT && popc(mA) <= P && popc(mB) <= P) {
        groups[0].lmask = mA; groups[1].lmask = mB;
        members = mem2;
      }
    }
  }
  const unsigned full = (c->n >= 32) ? 0xffffffffu : ((1u << c->n) - 1u);
  for (size_t g = 0; g < groups.size(); ++g) {
    // fill the local set up to P bits (lowest free qubits) so that tiles have a fixed size
    unsigned lm = groups[g].lmask;
    for (int q = 0; q < c->n && popc(lm) < P; ++q) if (!(lm & (1u << q))) lm |= (1u << q);
    groups[g].lmask = lm & full;
    groups[g].p = popc(groups[g].lmask);
    if (groups[g].p > P) sfail(DQ_ERR_STATE, "internal: a string group has more local qubits than a tile holds");
    for (int s : members[g]) {
      DTerm t;
      t.xmask = c->hterms[s].x; t.zmask = c->hterms[s].z; t.slot = s;
      t.xloc = compress_bits(t.xmask, groups[g].lmask);
      groups[g].apply.push_back(t);
      if (with_overlaps && c->hterms[s].owner >= 0) groups[g].ovl.push_back(t);
    }
  }
  if (with_overlaps) {  // diagonal control strings: overlaps in group 0 (x = 0 is local to every tiling)
    for (int s : slots)
      if (c->hterms[s].x == 0 && c->hterms[s].owner >= 0) {
        size_t g = 0;
        while (g < groups.size() && (int)groups[g].ovl.size() >= SMAXT) ++g;
        if (g == groups.size()) sfail(DQ_ERR_UNSUPPORTED, "too many diagonal control strings");
        DTerm t;
        t.xmask = 0; t.zmask = c->hterms[s].z; t.slot = s; t.xloc = 0;
        groups[g].ovl.push_back(t);
      }
  }
  for (auto& g : groups) {
    g.simple = (int)g.apply.size() <= SD_SIMPLE_T;
    for (auto& t : g.apply) if (t.zmask != 0) g.simple = false;
    {
      std::vector<int> omap(std::max<size_t>(1, g.apply.size()), -1);
      size_t found = 0;
      for (size_t o = 0; o < g.ovl.size(); ++o)
        for (size_t m = 0; m < g.apply.size(); ++m)
          if (g.apply[m].slot == g.ovl[o].slot) { omap[m] = (int)o; ++found; break; }
      g.ovl_fusable = g.simple && !g.ovl.empty() && found == g.ovl.size() && (int)g.apply.size() <= SD_OVL_T;
      g.d_omap.ensure(omap.size() * sizeof(int));
      SHIP(hipMemcpy(g.d_omap.p, omap.data(), omap.size() * sizeof(int), hipMemcpyHostToDevice));
    }
    g.d_apply.ensure(std::max<size_t>(1, g.apply.size()) * sizeof(DTerm));
    if (!g.apply.empty()) SHIP(hipMemcpy(g.d_apply.p, g.apply.data(), g.apply.size() * sizeof(DTerm), hipMemcpyHostToDevice));
    g.d_ovl.ensure(std::max<size_t>(1, g.ovl.size()) * sizeof(DTerm));
    g.d_ovl_slots.ensure(std::max<size_t>(1, g.ovl.size()) * sizeof(int));
    if (!g.ovl.empty()) {
      SHIP(hipMemcpy(g.d_ovl.p, g.ovl.data(), g.ovl.size() * sizeof(DTerm), hipMemcpyHostToDevice));
      std::vector<int> sl;
      for (auto& t : g.ovl) sl.push_back(t.slot);
      SHIP(hipMemcpy(g.d_ovl_slots.p, sl.data(), sl.size() * sizeof(int), hipMemcpyHostToDevice));
    }
  }
}

template <class T>
void build_diag(SCtx<T>* c, SBuf& out, const std::vector<int>& slots, bool& present) {
  std::vector<unsigned> z;
  std::vector<double> w;
  for (int s : slots) { z.push_back(c->hterms[s].z); w.push_back(c->hterms[s].w); }
  present = !slots.empty();
  if (!present) return;
  SBuf dz, dw;
  dz.ensure(z.size() * sizeof(unsigned)); dw.ensure(w.size() * sizeof(double));
  SHIP(hipMemcpy(dz.p, z.data(), z.size() * sizeof(unsigned), hipMemcpyHostToDevice));
  SHIP(hipMemcpy(dw.p, w.data(), w.size() * sizeof(double), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(sdiag_build_kernel<T>, dim3((c->d + 255) / 256), dim3(256), 0, c->st, (T*)out.p, c->d,
                     (const unsigned*)dz.p, (const double*)dw.p, (int)z.size());
  SHIP(hipGetLastError());
  SHIP(hipStreamSynchronize(c->st));
}

template <class T>
void set_terms(SCtx<T>* c, int nt, const int* owner, const unsigned* x, const unsigned* z, const double* w) {
  SHIP(hipSetDevice(c->dev));
  if (nt < 0 || (nt > 0 && (!owner || !x || !z || !w))) sfail(DQ_ERR_INVALID, "bad term arrays");
  const unsigned full = (c->n >= 32) ? 0xffffffffu : ((1u << c->n) - 1u);
  c->hterms.clear();
  std::vector<HostTerm> H, M;
  for (int i = 0; i < nt; ++i) {
    if (owner[i] < -2 || owner[i] >= c->K) sfail(DQ_ERR_INVALID, "term owner out of range");
    if ((x[i] | z[i]) & ~full) sfail(DQ_ERR_INVALID, "term acts on a qubit >= n_qubits");
    if (!std::isfinite(w[i])) sfail(DQ_ERR_INVALID, "non-finite term weight");
    HostTerm t{owner[i], x[i], z[i], w[i]};
    (owner[i] == -2 ? M : H).push_back(t);
  }
  c->hterms = H;
  c->n_hslots = (int)H.size();
  c->hterms.insert(c->hterms.end(), M.begin(), M.end());
  c->n_slots = (int)c->hterms.size();
  // coefficient tables: weight * i^{nY}
  std::vector<double> wre(c->n_slots), wim(c->n_slots);
  std::vector<int> own(c->n_slots);
  c->wsum0 = 0;
  c->wsumk.assign(c->K, 0.0);
  for (int s = 0; s < c->n_slots; ++s) {
    const HostTerm& t = c->hterms[s];
    const int ny = popc(t.x & t.z) & 3;
    const double re[4] = {1, 0, -1, 0}, im[4] = {0, 1, 0, -1};
    wre[s] = t.w * re[ny]; wim[s] = t.w * im[ny];
    own[s] = t.owner;
    // bound of the step's norm (Taylor degree): the off-diagonal strings by the sum of their weights; the Z-strings of a source
    // by the largest entry of the diagonal they add up to (below) -- for an Ising drift far less than the sum of |J| and |h|
    if (t.x == 0) continue;
    if (t.owner == -1) c->wsum0 += std::fabs(t.w);
    else if (t.owner >= 0) c->wsumk[t.owner] += std::fabs(t.w);
  }
  const size_t ns = std::max(1, c->n_slots);
  c->d_wre.ensure(ns * 8); c->d_wim.ensure(ns * 8); c->d_owner.ensure(ns * 4);
  c->d_coef.ensure(ns * sizeof(typename SCtx<T>::T2)); c->d_coefM.ensure(ns * sizeof(typename SCtx<T>::T2));
  if (c->n_slots) {
    SHIP(hipMemcpy(c->d_wre.p, wre.data(), c->n_slots * 8, hipMemcpyHostToDevice));
    SHIP(hipMemcpy(c->d_wim.p, wim.data(), c->n_slots * 8, hipMemcpyHostToDevice));
    SHIP(hipMemcpy(c->d_owner.p, own.data(), c->n_slots * 4, hipMemcpyHostToDevice));
  }
  // groups
  std::vector<int> hs, ms;
  for (int s = 0; s < c->n_hslots; ++s) hs.push_back(s);
  for (int s = c->n_hslots; s < c->n_slots; ++s) ms.push_back(s);
  build_groups(c, hs, c->hgroups, true);
  build_groups(c, ms, c->mgroups, false);
  // diagonals
  std::vector<int> d0, dm;
  std::vector<std::vector<int>> dk(c->K);
  for (int s = 0; s < c->n_slots; ++s) {
    const HostTerm& t = c->hterms[s];
    if (t.x != 0) continue;
    if (t.owner == -1) d0.push_back(s);
    else if (t.owner == -2) dm.push_back(s);
    else dk[t.owner].push_back(s);
  }
  c->D0.ensure((size_t)c->d * sizeof(T)); c->DM.ensure((size_t)c->d * sizeof(T)); c->Dt.ensure((size_t)c->d * sizeof(T));
  build_diag(c, c->D0, d0, c->has_D0);
  build_diag(c, c->DM, dm, c->has_DM);
  c->diag_controls.clear();
  for (int k = 0; k < c->K; ++k) if (!dk[k].empty()) c->diag_controls.push_back(k);
  c->Dk.ensure(std::max<size_t>(1, c->diag_controls.size()) * c->d * sizeof(T));
  c->Dkidx.ensure(std::max<size_t>(1, c->diag_controls.size()) * sizeof(int));
  for (size_t j = 0; j < c->diag_controls.size(); ++j) {
    SBuf one; bool pr;
    one.ensure((size_t)c->d * sizeof(T));
    build_diag(c, one, dk[c->diag_controls[j]], pr);
    SHIP(hipMemcpy((char*)c->Dk.p + j * (size_t)c->d * sizeof(T), one.p, (size_t)c->d * sizeof(T), hipMemcpyDeviceToDevice));
  }
  if (!c->diag_controls.empty())
    SHIP(hipMemcpy(c->Dkidx.p, c->diag_controls.data(), c->diag_controls.size() * sizeof(int), hipMemcpyHostToDevice));
  {  // largest |entry| of the drift's diagonal and of every diagonal control's: their share of the norm bound
    std::vector<T> h((size_t)c->d);
    auto absmax = [&](const void* dev) {
      SHIP(hipMemcpy(h.data(), dev, (size_t)c->d * sizeof(T), hipMemcpyDeviceToHost));
      double m = 0.0;
      for (T v : h) m = std::max(m, (double)std::fabs(v));
      return m;
    };
    if (c->has_D0) c->wsum0 += absmax(c->D0.p);
    for (size_t j = 0; j < c->diag_controls.size(); ++j)
      c->wsumk[c->diag_controls[j]] += absmax((const char*)c->Dk.p + j * (size_t)c->d * sizeof(T));
  }
  // observable coefficients are static: weight * i^{nY}
  if (c->n_slots) {
    hipLaunchKernelGGL(scoef_kernel<T>, dim3((c->n_slots + 63) / 64), dim3(64), 0, c->st,
                       (typename SCtx<T>::T2*)c->d_coefM.p, (const double*)c->d_wre.p, (const double*)c->d_wim.p,
                       (const int*)c->d_owner.p, (const double*)nullptr, c->n_slots);
    // (control slots get 0 here: u = null; only the observable's slots are ever read from d_coefM)
  }
  SHIP(hipStreamSynchronize(c->st));
  c->have_terms = true;
  c->have_obs = !ms.empty() || !dm.empty();
  c->last_groups = (double)c->hgroups.size();
}

// one application  out = add + scale * (H_like y)   over the groups of `groups`
// zo != null: every pass also forms its group's control overlaps <y, P'_m zo> (part[g]: per-workgroup partial sums; the
// caller runs soverlap_finish_kernel on them) -- the partner rows of y are in the tile anyway, zo costs one more read
template <class T>
void apply_groups(SCtx<T>* c, std::vector<Group>& groups, const typename SCtx<T>::T2* coef, const T* diag,
                  const typename SCtx<T>::T2* y, const typename SCtx<T>::T2* add, typename SCtx<T>::T2* out,
                  typename SCtx<T>::T2* acc, double sr, double si, int B, const typename SCtx<T>::T2* zo = nullptr,
                  std::vector<SBuf>* parts = nullptr, const typename SCtx<T>::T2* acc_in = nullptr) {
  typedef typename SCtx<T>::T2 T2;
  const size_t sb = c->state_bytes(B);
  const int ng = (int)groups.size();
  const T2* prev = nullptr;
  for (int g = 0; g < ng; ++g) {
    const bool last = (g == ng - 1);
    SApply<T> a;
    a.zo = nullptr; a.oterms = nullptr; a.n_ovl = 0; a.part = nullptr; a.out2 = nullptr;
    a.y = y; a.prev = prev;
    a.add = last ? add : nullptr;
    a.acc = last ? acc : nullptr;
    a.acc_in = last ? (acc_in ? acc_in : acc) : nullptr;
    a.diag = (g == 0) ? diag : nullptr;
    a.terms = (const DTerm*)groups[g].d_apply.p;
    a.coef = coef;
    a.nterms = (int)groups[g].apply.size();
    a.n = c->n; a.p = groups[g].p; a.lmask = groups[g].lmask; a.B = B;
    a.sr = last ? (T)sr : (T)1; a.si = last ? (T)si : (T)0;
    if (!last) c->tmp[g & 1].ensure(sb);
    T2* dst = last ? out : (T2*)c->tmp[g & 1].p;
    a.out = dst;
    dim3 grid(c->d >> a.p, (B + SBC - 1) / SBC);
    // (complex128: the general -- not pure X-string -- form with all four register operands keeps the one-tile kernel: its DMA form
    // needs more than the 128 registers a wave of a 1024-thread workgroup has, and the asm loads allow no spill)
    const bool dma = sdma_ok<T>(a.p, B, a.nterms, c->dma, c->n) &&
                     !(sizeof(T) == sizeof(double) && !groups[g].simple && prev && a.add && a.acc && a.diag);
    a.omap = nullptr;
    if (zo && !groups[g].ovl.empty()) {
      a.zo = zo;
      a.oterms = (const DTerm*)groups[g].d_ovl.p;
      a.n_ovl = (int)groups[g].ovl.size();
      a.omap = (const int*)groups[g].d_omap.p;
      (*parts)[g].ensure((size_t)grid.x * grid.y * (STHREADS / 64) * a.n_ovl * 16);
      a.part = (double*)(*parts)[g].p;
      groups[g].last_nblk = (int)(grid.x * grid.y) * (STHREADS / 64);
    }
    const double units = 2.0 + (prev ? 1 : 0) + (a.add ? 1 : 0) + (a.acc ? 2 : 0) + (a.zo ? 1 : 0);
    if (dma && (!a.zo || groups[g].ovl_fusable)) {
      SDmaWalk wk = sdma_walk(grid.x, grid.y, a.lmask);
      const int nblk = sdma_blocks(wk.nunits, c->n_cu);
      if (a.zo) groups[g].last_nblk = nblk * (STHREADS / 64);
      c->last_dma_passes += 1;
      c->timed(units * sb, [&] { sapply_dma_launch(a, wk, nblk, c->st, groups[g].simple); });
    } else {
      c->timed(units * sb, [&] { sapply_launch<T>(a, grid, c->st); });
    }
    c->last_passes += 1;
    c->last_bytes += units * sb;
    prev = dst;
  }
}

template <class T>
void step_coefficients(SCtx<T>* c, int t, bool next_set = false) {
  typedef typename SCtx<T>::T2 T2;
  const double* u_row = (const double*)c->u_dev.p + (size_t)t * c->K;
  if (next_set) { c->d_coef2.ensure(c->d_coef.bytes); c->Dt2.ensure(c->Dt.bytes); }
  if (c->n_hslots)
    hipLaunchKernelGGL(scoef_kernel<T>, dim3((c->n_hslots + 63) / 64), dim3(64), 0, c->st, (T2*)(next_set ? c->d_coef2.p : c->d_coef.p),
                       (const double*)c->d_wre.p, (const double*)c->d_wim.p, (const int*)c->d_owner.p, u_row, c->n_hslots);
  if (c->has_D0 || !c->diag_controls.empty())
    hipLaunchKernelGGL(sdiag_step_kernel<T>, dim3((c->d + 255) / 256), dim3(256), 0, c->st, (T*)(next_set ? c->Dt2.p : c->Dt.p),
                       c->has_D0 ? (const T*)c->D0.p : (const T*)nullptr, (const T*)c->Dk.p, u_row,
                       (const int*)c->Dkidx.p, (int)c->diag_controls.size(), c->d);
  SHIP(hipGetLastError());
}

template <class T>
const T* step_diag(SCtx<T>* c) {
  return (c->has_D0 || !c->diag_controls.empty()) ? (const T*)c->Dt.p : (const T*)nullptr;
}

int taylor_degree(double theta, double tol, int force) {
  if (force > 0) return force;
  for (int m = 1; m <= 60; ++m) {
    double v = 1.0;
    for (int j = 1; j <= m + 1; ++j) v *= theta / j;
    if (v <= tol) return m;
  }
  sfail(DQ_ERR_INVALID, "step generator norm too large for the Taylor apply (reduce dt)");
}

template <class T>
std::vector<double> s_fetch_u(SCtx<T>* c, const double* u, int S) {
  const size_t n = (size_t)S * c->K;
  std::vector<double> h(n);
  if (n) {
    if (s_is_device_ptr(u)) { SHIP(hipMemcpy(h.data(), u, n * 8, hipMemcpyDeviceToHost)); }
    else memcpy(h.data(), u, n * 8);
    for (double v : h) if (!std::isfinite(v)) sfail(DQ_ERR_INVALID, "pulse table contains non-finite values");
  }
  c->u_dev.ensure(std::max<size_t>(n, 1) * 8);
  if (n) SHIP(hipMemcpy(c->u_dev.p, h.data(), n * 8, hipMemcpyHostToDevice));
  return h;
}

template <class T>
int plan_degree(SCtx<T>* c, const std::vector<double>& uh, int S, double dt) {
  double th = 0.0;
  for (int t = 0; t < S; ++t) {
    double h = c->wsum0;
    for (int k = 0; k < c->K; ++k) h += std::fabs(uh[(size_t)t * c->K + k]) * c->wsumk[k];
    th = std::max(th, std::fabs(dt) * h);
  }
  c->last_theta = th;
  const double tol = std::min(c->tol, c->chain_tol / (double)std::max(std::max(S, c->plan_steps), 1));
  int m = taylor_degree(th, tol, c->force_degree);
  c->last_taylor_degree = m;
  c->rho.assign((size_t)m + 2, 0.0);
  for (int j = 1; j <= m + 1; ++j) c->rho[j] = 1.0 / j;
  if (c->force_degree <= 0 && c->economize && th <= 2.0) {
    dq::XsPoly pol;
    const int L = dq::xs_poly_make(th, 0.5 * tol, 1, &pol);
    if (L > 0 && L < m) {
      m = L;
      c->rho.assign((size_t)m + 2, 0.0);
      for (int j = 1; j <= m; ++j) c->rho[j] = pol.g[j] / ((double)j * pol.g[j - 1]);
    }
  }
  c->last_degree = m;
  return m;
}

// ---- Horner chain in the chain form -------------------------------------------------------------------------------------------------
template <class T>
bool chain_ok(SCtx<T>* c, int B) {
  if (!c->chain || c->hgroups.size() != 2) return false;
  for (auto& g : c->hgroups)
    if (!g.simple || g.apply.empty() || !sdma_ok<T>(g.p, B, (int)g.apply.size(), c->dma, c->n)) return false;
  return true;
}

// one pass of the chain over group g: which = 0 start (out = D y + H_g y), 1 link (out = add + s (prev + H_g y); out2 = D out + H_g out),
// 2 finish (out = add + s (prev + H_g y))
template <class T>
void chain_pass(SCtx<T>* c, int g, int which, const typename SCtx<T>::T2* y, const typename SCtx<T>::T2* prev,
                const typename SCtx<T>::T2* add, typename SCtx<T>::T2* out, typename SCtx<T>::T2* out2, double sr, double si, int B,
                bool next_step = false) {   // next_step (links only): the application out2 starts is the NEXT step's (second coefficient set)
  const size_t sb = c->state_bytes(B);
  Group& G = c->hgroups[g];
  SApply<T> a;
  a.zo = nullptr; a.oterms = nullptr; a.n_ovl = 0; a.part = nullptr; a.omap = nullptr; a.acc = nullptr; a.acc_in = nullptr;
  a.y = y; a.prev = prev; a.add = add; a.out = out; a.out2 = out2;
  a.diag = (which == 2) ? nullptr : step_diag(c);
  a.terms = (const DTerm*)G.d_apply.p;
  a.coef = (const typename SCtx<T>::T2*)c->d_coef.p;
  if (next_step && which == 1) {
    a.coef2 = (const typename SCtx<T>::T2*)c->d_coef2.p;
    a.diag = step_diag(c) ? (const T*)c->Dt2.p : (const T*)nullptr;
  }
  a.nterms = (int)G.apply.size();
  a.n = c->n; a.p = G.p; a.lmask = G.lmask; a.B = B;
  a.sr = which == 0 ? (T)1 : (T)sr; a.si = which == 0 ? (T)0 : (T)si;
  dim3 grid(c->d >> a.p, (B + SBC - 1) / SBC);
  SDmaWalk wk = sdma_walk(grid.x, grid.y, a.lmask);
  const int nblk = sdma_blocks(wk.nunits, c->n_cu);
  const double units = which == 0 ? 2.0 : (which == 1 ? 5.0 : 4.0);
  if (which == 1) c->timed(units * sb, [&] { schain_dma_launch(a, wk, nblk, c->st); });
  else c->timed(units * sb, [&] { sapply_dma_launch(a, wk, nblk, c->st, true); });
  c->last_passes += 1;
  c->last_dma_passes += 1;
  c->last_bytes += units * sb;
}

// A chain may hand its last tile walk to the chain that follows on its result: the finishing pass then also STARTS that chain
// (out2 = D z + H_g z on the group g it is on, in tmp[pb]) -- the reversal of psi_t hands over to the chain of psi_t's Horner iterates.
struct ChainLink { int g = 0, pb = 0; bool set = false; };

// z_{j-1} = in + (0, -sign dt / j) H z_j  for j = jhi .. jlo  (z_jhi = zhi); dst(j - 1) names where z_{j-1} goes.
// from: the start of this chain was done by the chain before (zhi = its result); to: start the next chain in this one's last pass.
template <class T, class Dst>
void taylor_chain(SCtx<T>* c, int jhi, int jlo, double dt, double sign, const typename SCtx<T>::T2* in,
                  const typename SCtx<T>::T2* zhi, int B, Dst dst, const ChainLink* from = nullptr, ChainLink* to = nullptr,
                  bool to_next_step = false) {   // to_next_step: the chain `to` starts is the next step's (SCtx::step_handover)
  typedef typename SCtx<T>::T2 T2;
  const size_t sb = c->state_bytes(B);
  c->tmp[0].ensure(sb); c->tmp[1].ensure(sb);
  int g = 0, pb = 0;
  if (from && from->set) { g = from->g; pb = from->pb; }
  else chain_pass<T>(c, g, 0, zhi, nullptr, nullptr, (T2*)c->tmp[pb].p, nullptr, 1.0, 0.0, B);
  const T2* zj = zhi;
  for (int j = jhi; j >= jlo; --j) {
    g ^= 1;  // the application is finished on the other group ...
    T2* znew = dst(j - 1);
    if (j > jlo || to) {  // ... which also starts the next one (of this chain, or of the chain that follows)
      chain_pass<T>(c, g, 1, zj, (const T2*)c->tmp[pb].p, in, znew, (T2*)c->tmp[pb ^ 1].p, 0.0, -sign * dt * c->rho[j], B,
                    to_next_step && j == jlo);
      pb ^= 1;
    } else {
      chain_pass<T>(c, g, 2, zj, (const T2*)c->tmp[pb].p, in, znew, nullptr, 0.0, -sign * dt * c->rho[j], B);
    }
    zj = znew;
  }
  if (to) { to->g = g; to->pb = pb; to->set = true; }
}

// The reverse sweep of one step in the chain form (srchain_dma_kernel): lambda_new = lambda_old + sum_j zbar_j and the control
// overlaps of all stages into grad_row.  z[j] are the forward Horner iterates (z[m] = psi_t).
template <class T>
void reverse_chain(SCtx<T>* c, int m, double dt, int B, const std::vector<const typename SCtx<T>::T2*>& z,
                   const typename SCtx<T>::T2* lam_old, typename SCtx<T>::T2* lam_new, double* grad_row) {
  typedef typename SCtx<T>::T2 T2;
  const size_t sb = c->state_bytes(B);
  c->tmp[0].ensure(sb); c->tmp[1].ensure(sb);
  if (c->oparts.size() < c->hgroups.size()) c->oparts.resize(c->hgroups.size());
  auto pass = [&](int g, int which, const T2* y, const T2* prev, const T2* acc_in, const T2* zo, const T2* zo2, T2* out, T2* out2,
                  double si, double ovl2) {
    Group& G = c->hgroups[g];
    SApply<T> a;
    a.add = nullptr; a.oterms = (const DTerm*)G.d_ovl.p; a.n_ovl = (int)G.ovl.size(); a.omap = (const int*)G.d_omap.p;
    a.y = y; a.prev = prev; a.out = out; a.out2 = out2; a.zo = zo; a.zo2 = zo2; a.ovl2_scale = (T)ovl2;
    a.acc = which == 0 ? nullptr : lam_new; a.acc_in = acc_in;
    a.diag = (which == 2) ? nullptr : step_diag(c);
    a.terms = (const DTerm*)G.d_apply.p;
    a.coef = (const T2*)c->d_coef.p;
    a.nterms = (int)G.apply.size();
    a.n = c->n; a.p = G.p; a.lmask = G.lmask; a.B = B;
    a.sr = which == 0 ? (T)1 : (T)0; a.si = which == 0 ? (T)0 : (T)si;
    dim3 grid(c->d >> a.p, (B + SBC - 1) / SBC);
    SDmaWalk wk = sdma_walk(grid.x, grid.y, a.lmask);
    const int nblk = sdma_blocks(wk.nunits, c->n_cu);
    c->oparts[g].ensure((size_t)nblk * (STHREADS / 64) * std::max(1, a.n_ovl) * 16);
    a.part = (double*)c->oparts[g].p;
    G.last_nblk = nblk * (STHREADS / 64);
    const double units = which == 0 ? 3.0 : (which == 1 ? 8.0 : (out ? 6.0 : 5.0));
    if (which == 1) c->timed(units * sb, [&] { srchain_dma_launch(a, wk, nblk, c->st); });
    else c->timed(units * sb, [&] { sapply_dma_launch(a, wk, nblk, c->st, true); });
    c->last_passes += 1; c->last_dma_passes += 1; c->last_bytes += units * sb;
  };
  auto finish = [&](int g, double fac) {
    Group& G = c->hgroups[g];
    if (G.ovl.empty()) return;
    hipLaunchKernelGGL(soverlap_finish_kernel, dim3((unsigned)G.ovl.size()), dim3(256), 0, c->st, (const double*)c->oparts[g].p,
                       G.last_nblk, (int)G.ovl.size(), (const int*)G.d_ovl_slots.p, (const int*)c->d_owner.p,
                       (const double*)c->d_wre.p, (const double*)c->d_wim.p, fac, grad_row);
    SHIP(hipGetLastError());
  };
  int g = 0, pb = 0;
  pass(g, 0, lam_old, nullptr, nullptr, z[1], nullptr, (T2*)c->tmp[pb].p, nullptr, 0.0, 0.0);   // stage 1 starts on group 0
  finish(g, dt * c->rho[1]);
  const T2* zprev = lam_old;
  for (int j = 1; j <= m; ++j) {
    g ^= 1;
    c->zbar[j & 1].ensure(sb);
    T2* zb_new = (T2*)c->zbar[j & 1].p;
    const T2* acc_in = j == 1 ? lam_old : (const T2*)lam_new;
    if (j < m) {
      pass(g, 1, zprev, (const T2*)c->tmp[pb].p, acc_in, z[j], z[j + 1], zb_new, (T2*)c->tmp[pb ^ 1].p, dt * c->rho[j], c->rho[j + 1] / c->rho[j]);
      pb ^= 1;
    } else {   // (the last stage: zbar_m goes into lambda_t and nowhere else -- no store of it: 5 units instead of 6)
      pass(g, 2, zprev, (const T2*)c->tmp[pb].p, acc_in, z[j], nullptr, c->reverse_keep_last ? zb_new : (T2*)nullptr, nullptr, dt * c->rho[j], 0.0);
    }
    finish(g, dt * c->rho[j]);
    zprev = zb_new;
  }
}

// z_out = T_m(sign * A_t) z_in  (Horner); if zs_store != null keeps the iterates z_m..z_1 there (z_j in zs_store[j])
template <class T>
void taylor_apply(SCtx<T>* c, int m, double dt, double sign, const typename SCtx<T>::T2* in, typename SCtx<T>::T2* out,
                  int B, std::vector<SBuf>* store, ChainLink* to = nullptr, const ChainLink* from = nullptr, bool to_next_step = false) {
  typedef typename SCtx<T>::T2 T2;
  const size_t sb = c->state_bytes(B);
  // z_m = in;  z_{j-1} = in + (A z_j)/j,  A = -i dt H * sign  ->  scale = (0, -sign*dt/j)
  if (m >= 1 && chain_ok(c, B)) {
    taylor_chain<T>(c, m, 1, dt, sign, in, in, B, [&](int jm1) -> T2* {
      if (jm1 == 0) return out;
      if (store) { (*store)[jm1].ensure(sb); return (T2*)(*store)[jm1].p; }
      c->zbar[(jm1 + 1) & 1].ensure(sb);
      return (T2*)c->zbar[(jm1 + 1) & 1].p;
    }, from, to, to_next_step);
    return;
  }
  if (from && from->set) sfail(DQ_ERR_STATE, "internal: a step handover reached the pass-per-group form");
  const T2* zj = in;
  for (int j = m; j >= 1; --j) {
    T2* dst;
    if (j == 1) dst = out;
    else if (store) { (*store)[j - 1].ensure(sb); dst = (T2*)(*store)[j - 1].p; }
    else { c->zbar[j & 1].ensure(sb); dst = (T2*)c->zbar[j & 1].p; }
    apply_groups(c, c->hgroups, (const T2*)c->d_coef.p, step_diag(c), zj, in, dst, (T2*)nullptr, 0.0, -sign * dt * c->rho[j], B);
    zj = dst;
  }
  if (m == 0) SHIP(hipMemcpyAsync(out, in, sb, hipMemcpyDeviceToDevice, c->st));
}

template <class T>
void s_check(SCtx<T>* c) {
  SHIP(hipSetDevice(c->dev));
  if (!c->have_terms) sfail(DQ_ERR_STATE, "dq_sctx_set_terms has not been called");
}

template <class T>
void s_propagate(SCtx<T>* c, const double* u, int S, double dt, const void* psi_in, void* psi_out, int B) {
  typedef typename SCtx<T>::T2 T2;
  s_check(c);
  if (S < 0 || B < 0 || !std::isfinite(dt)) sfail(DQ_ERR_INVALID, "bad n_steps / batch / dt");
  if (B == 0) return;
  if (!psi_in || !psi_out || (S > 0 && c->K > 0 && !u)) sfail(DQ_ERR_INVALID, "null pointer");
  const size_t sb = c->state_bytes(B);
  std::vector<double> uh = s_fetch_u(c, u, S);
  const int m = plan_degree(c, uh, S, dt);
  c->psi[0].ensure(sb); c->psi[1].ensure(sb);
  SHIP(hipMemcpyAsync(c->psi[0].p, psi_in, sb, hipMemcpyDefault, c->st));
  c->last_passes = 0; c->last_bytes = 0; c->last_dma_passes = 0;
  int cur = 0;
  const bool xstep = c->step_handover != 0 && m >= 1 && chain_ok(c, B);   // (SCtx::step_handover)
  ChainLink hand;
  for (int t = 0; t < S; ++t) {
    if (!hand.set) step_coefficients(c, t);
    const bool over = xstep && t + 1 < S;
    if (over) step_coefficients(c, t + 1, true);
    ChainLink next;
    taylor_apply(c, m, dt, 1.0, (const T2*)c->psi[cur].p, (T2*)c->psi[cur ^ 1].p, B, nullptr, over ? &next : nullptr, &hand, over);
    if (over) c->swap_step_sets();
    hand = next;
    cur ^= 1;
  }
  SHIP(hipMemcpyAsync(psi_out, c->psi[cur].p, sb, hipMemcpyDefault, c->st));
  SHIP(hipStreamSynchronize(c->st));
}

// lam = 2 M psi, energies
template <class T>
void s_energy(SCtx<T>* c, const typename SCtx<T>::T2* psi, typename SCtx<T>::T2* lam, int B, double* energies, double* total) {
  typedef typename SCtx<T>::T2 T2;
  apply_groups(c, c->mgroups, (const T2*)c->d_coefM.p, c->has_DM ? (const T*)c->DM.p : (const T*)nullptr, psi,
               (const T2*)nullptr, lam, (T2*)nullptr, 2.0, 0.0, B);
  c->e_dev.ensure(8 * ((size_t)B + 1));
  c->part.ensure(8 * (size_t)dq::coldot_chunks((int)c->d) * B);
  SHIP(dq::coldot_launch<T>(psi, B, lam, B, (int)c->d, B, 0.5, (double*)c->e_dev.p, (double*)c->part.p, c->st));
  SHIP(dq::sum_launch((const double*)c->e_dev.p, B, (double*)c->e_dev.p + B, c->st));
  if (energies) SHIP(hipMemcpyAsync(energies, c->e_dev.p, 8 * (size_t)B, hipMemcpyDefault, c->st));
  if (total) SHIP(hipMemcpyAsync(total, (double*)c->e_dev.p + B, 8, hipMemcpyDefault, c->st));
}

template <class T>
void s_energy_grad(SCtx<T>* c, const double* u, int S, double dt, const void* psi0, int B, double* loss, double* grad_u,
                   void* psi_final, double* energies) {
  typedef typename SCtx<T>::T2 T2;
  s_check(c);
  if (!c->have_obs) sfail(DQ_ERR_STATE, "no observable strings (owner -2) were given to dq_sctx_set_terms");
  if (S < 0 || B <= 0 || !std::isfinite(dt)) sfail(DQ_ERR_INVALID, "bad n_steps / batch / dt");
  if (!psi0 || (S > 0 && c->K > 0 && !u)) sfail(DQ_ERR_INVALID, "null pointer");
  const size_t sb = c->state_bytes(B);
  const int K = c->K;
  std::vector<double> uh = s_fetch_u(c, u, S);
  const int m = plan_degree(c, uh, S, dt);
  // (a call that failed half-way may have left checkpoint copies in flight on the copy stream: none may land in `stage` or leave
  // the pinned buffers while this call takes its first snapshots)
  if (c->copy_st) SHIP(hipStreamSynchronize(c->copy_st));
  double budget = c->checkpoint_bytes;
  if (budget <= 0) {
    // default: 70 % of the free HBM plus everything the checkpoint pools of this context hold already -- ALL of it is at the plan's
    // disposal, because the pools the plan does not use are released below before anything new is allocated (a context that ran
    // the two-level plan keeps ~m s states in segz / segpsi; a one-level plan sized from that memory must not allocate on top of it)
    size_t fr = 0, tot = 0;
    SHIP(hipMemGetInfo(&fr, &tot));
    double held = (double)c->stage.bytes;
    for (const auto& b : c->ck) held += (double)b.bytes;
    for (const auto& b : c->segpsi) held += (double)b.bytes;
    for (const auto& v : c->segz) for (const auto& b : v) held += (double)b.bytes;
    budget = 0.70 * ((double)fr + held);
  }
  const long ncp_max = std::max(1L, (long)(budget / (double)sb));
  const long Splan = std::max(S, c->plan_steps);
  int cstride = (int)std::max(1L, (Splan + ncp_max - 1) / ncp_max);
  // Two levels.  With stride > 1 a step's psi_t costs a reversal chain (5m + 1 state-units) and its Horner iterates another chain
  // (5m - 4).  Where memory allows, keep every s-th state instead and, a segment at a time, recompute the segment forward once
  // WITH the iterates of every step stored (m states per step): the recomputation costs what the reversal did, the iterates chain
  // is gone (n = 18, m = 5: 113 -> 92 units per step).  s = sqrt(steps / m) minimises the memory: steps / s + m s states.
  bool two_level = false, host_ck = false;
  if (c->two_level && cstride > 1 && m >= 2) {
    const long s2 = std::max(2L, (long)std::lround(std::sqrt((double)Splan / m)));
    if ((Splan + s2 - 1) / s2 + (long)m * s2 <= ncp_max) { two_level = true; cstride = (int)s2; }
    else if (c->host_checkpoints && (long)m * s2 + 1 <= ncp_max) {
      // the segment fits, the kept states do not: they go to pinned host memory (one state down per s steps on the way forward,
      // one up per segment on the way back: 1 / s of a step's traffic over the host link) -- if the host pins that much
      const int need = S > 0 ? (S + (int)s2 - 1) / (int)s2 : 0;
      if ((int)c->ckh.size() < need) c->ckh.resize(need);
      double pinned = 0.0;
      for (const auto& b : c->ckh) pinned += (double)b.bytes;
      // (no pinning beyond the budget: a plan the host cannot hold falls back to one level -- slower, never an out-of-memory kill)
      const double hb = c->host_checkpoint_bytes > 0 ? c->host_checkpoint_bytes : s_host_budget() + pinned;
      bool ok = (double)need * (double)sb <= hb;
      for (int i = 0; i < need && ok; ++i) ok = c->ckh[i].ensure(sb);
      if (ok) { two_level = true; host_ck = true; cstride = (int)s2; c->copy_stream_ready(); c->stage.ensure(sb); }
    }
  }
  c->last_two_level = two_level ? 1 : 0;
  c->last_host_ck = host_ck ? 1 : 0;
  const int ncp = S > 0 ? (S + cstride - 1) / cstride : 0;
  c->last_cstride = cstride;
  // release the pools this plan does not use (or uses less of) before it allocates its own
  if (host_ck) c->ck.clear();
  else if ((int)c->ck.size() > ncp) c->ck.resize(ncp);
  if (!host_ck) c->stage.release();
  if (!two_level) { c->segpsi.clear(); c->segz.clear(); }
  else {
    if ((int)c->segpsi.size() > cstride + 1) c->segpsi.resize(cstride + 1);
    if ((int)c->segz.size() > cstride + 1) c->segz.resize(cstride + 1);
  }
  if (!host_ck) {
    if ((int)c->ck.size() < ncp) c->ck.resize(ncp);
    for (int i = 0; i < ncp; ++i) c->ck[i].ensure(sb);
  }
  c->psi[0].ensure(sb); c->psi[1].ensure(sb); c->lam[0].ensure(sb); c->lam[1].ensure(sb);
  if ((int)c->zs.size() < m + 1) c->zs.resize(m + 1);
  c->grad_dev.ensure(std::max<size_t>((size_t)S * K, 1) * 8);
  SHIP(hipMemsetAsync(c->grad_dev.p, 0, std::max<size_t>((size_t)S * K, 1) * 8, c->st));
  c->last_passes = 0; c->last_bytes = 0; c->last_dma_passes = 0;

  // ---- forward ----
  SHIP(hipMemcpyAsync(c->psi[0].p, psi0, sb, hipMemcpyDefault, c->st));
  int cur = 0;
  const bool xstep = c->step_handover != 0 && m >= 1 && chain_ok(c, B);   // the last pass of a step's chain starts the next step's
  ChainLink hand;
  for (int t = 0; t < S; ++t) {
    if (t % cstride == 0) {
      if (host_ck) {  // snapshot on the chain's stream (a device copy), the way down on the copy stream behind it
        if (t > 0) SHIP(hipStreamWaitEvent(c->st, c->ev_down, 0));               // the previous state has left the staging buffer
        SHIP(hipMemcpyAsync(c->stage.p, c->psi[cur].p, sb, hipMemcpyDeviceToDevice, c->st));
        SHIP(hipEventRecord(c->ev_staged, c->st));
        SHIP(hipStreamWaitEvent(c->copy_st, c->ev_staged, 0));
        SHIP(hipMemcpyAsync(c->ckh[t / cstride].p, c->stage.p, sb, hipMemcpyDeviceToHost, c->copy_st));
        SHIP(hipEventRecord(c->ev_down, c->copy_st));
      } else SHIP(hipMemcpyAsync(c->ck[t / cstride].p, c->psi[cur].p, sb, hipMemcpyDeviceToDevice, c->st));
    }
    if (!hand.set) step_coefficients(c, t);   // (else: made by step t - 1 into the second set, which is the first by now)
    const bool over = xstep && t + 1 < S;
    if (over) step_coefficients(c, t + 1, true);
    ChainLink next;
    taylor_apply(c, m, dt, 1.0, (const T2*)c->psi[cur].p, (T2*)c->psi[cur ^ 1].p, B, nullptr, over ? &next : nullptr, &hand, over);
    if (over) c->swap_step_sets();
    hand = next;
    cur ^= 1;
  }
  s_energy(c, (const T2*)c->psi[cur].p, (T2*)c->lam[0].p, B, energies, loss);
  if (psi_final) SHIP(hipMemcpyAsync(psi_final, c->psi[cur].p, sb, hipMemcpyDefault, c->st));

  // ---- backward ----
  int lcur = 0;
  // the reverse sweep of step t: lambda_{t+1} -> lambda_t and dL/du[t]; z[m] = psi_t, z[j] its Horner iterates
  auto reverse_step = [&](int t, const std::vector<const T2*>& z) {
    // reverse: zbar_0 = lambda_{t+1}; lambda_t = sum_j zbar_j; overlaps <zbar_{j-1}, P z_j>
    T2* lam_new = (T2*)c->lam[lcur ^ 1].p;   // = lambda_{t+1} + zbar_1 + ... + zbar_m: the first term reads lambda_{t+1} where it is
    if (m < 1) SHIP(hipMemcpyAsync(lam_new, c->lam[lcur].p, sb, hipMemcpyDeviceToDevice, c->st));
    const T2* zb_prev = (const T2*)c->lam[lcur].p;
    // option fuse_overlaps = 1: the reverse apply of a group also forms that group's overlaps (zbar_{j-1} is its tile, z_j read
    // at the own rows) -- one state-unit less per group and Horner stage, but the fused kernel's accumulators cost the second
    // workgroup per CU (64-VGPR limit): 256 -> 231-242 steps/s at n = 18, B = 64.  Default: separate overlap passes.
    // With the DMA form (one workgroup per CU, 128 VGPRs to spend) the fusion is free of that cost: taken whenever every group's
    // control strings are among its applied pure X-strings (fuse_overlaps = -1 forbids it).
    bool fuse = c->fuse_overlaps > 0;
    if (c->fuse_overlaps == 0 && c->dma != 0) {
      bool all = true;
      for (auto& g : c->hgroups)
        if (!sdma_ok<T>(g.p, B, (int)g.apply.size(), c->dma, c->n) || (!g.ovl.empty() && !g.ovl_fusable)) all = false;
      fuse = all;
    }
    for (size_t g = 0; g < c->hgroups.size(); ++g) {
      if ((int)c->hgroups[g].ovl.size() > SOVL_TCH) fuse = false;
      const int f = (g > 0 ? 1 : 0) | (g + 1 == c->hgroups.size() ? 4 : 0) | (g == 0 && step_diag(c) ? 8 : 0);
      if (!c->hgroups[g].ovl.empty() && !sapply_ovl_flags_ok(f)) fuse = false;
    }
    if (c->oparts.size() < c->hgroups.size()) c->oparts.resize(c->hgroups.size());
    bool rchain = fuse && m >= 1 && c->reverse_chain && chain_ok(c, B);
    for (auto& g : c->hgroups) if (!g.ovl.empty() && !g.ovl_fusable) rchain = false;
    if (rchain) reverse_chain<T>(c, m, dt, B, z, (const T2*)c->lam[lcur].p, lam_new, (double*)c->grad_dev.p + (size_t)t * K);
    for (int j = 1; j <= m && !rchain; ++j) {
      for (auto& g : c->hgroups) {
        if (fuse || g.ovl.empty()) continue;
        SOverlap<T> o;
        o.za = zb_prev; o.zb = z[j];
        o.terms = (const DTerm*)g.d_ovl.p; o.nterms = (int)g.ovl.size();
        o.n = c->n; o.p = g.p; o.lmask = g.lmask; o.B = B;
        o.ntiles = c->d >> g.p;
        const unsigned ychunks = (B + SBC - 1) / SBC;
        o.tpw = (int)std::max(1u, std::min(8u, (o.ntiles * ychunks) / 512u));  // keep >= ~512 workgroups (2 per CU)
        dim3 grid((o.ntiles + o.tpw - 1) / o.tpw, ychunks);
        const bool use_dma = sdma_ok<T>(o.p, B, o.nterms, c->dma, c->n) && o.nterms <= SOVL_TCH;   // (both dtypes since round 5)
        SDmaWalk wk = sdma_walk(o.ntiles, ychunks, o.lmask);
        const int nblk = use_dma ? sdma_blocks(wk.nunits, c->n_cu) : (int)(grid.x * grid.y);
        c->part.ensure((size_t)nblk * o.nterms * 16);
        o.part = (double*)c->part.p;
        c->timed(2.0 * sb, [&] {
          if (use_dma) { c->last_dma_passes += 1; soverlap_dma_launch(o, wk, nblk, c->st); }
          else if ((B & 1) == 0) hipLaunchKernelGGL((soverlap_kernel<T, 2>), grid, dim3(STHREADS), 0, c->st, o);
          else hipLaunchKernelGGL((soverlap_kernel<T, 1>), grid, dim3(STHREADS), 0, c->st, o);
        });
        c->last_passes += 1; c->last_bytes += 2.0 * sb;
        hipLaunchKernelGGL(soverlap_finish_kernel, dim3(o.nterms), dim3(256), 0, c->st, (const double*)c->part.p, nblk, o.nterms,
                           (const int*)g.d_ovl_slots.p, (const int*)c->d_owner.p, (const double*)c->d_wre.p,
                           (const double*)c->d_wim.p, dt * c->rho[j], (double*)c->grad_dev.p + (size_t)t * K);
        SHIP(hipGetLastError());
      }
      // zbar_j = (A^H zbar_{j-1})/j = (+i dt/j) H zbar_{j-1}; accumulated into lambda_t
      c->zbar[j & 1].ensure(sb);
      T2* zb_new = (T2*)c->zbar[j & 1].p;
      apply_groups(c, c->hgroups, (const T2*)c->d_coef.p, step_diag(c), zb_prev, (const T2*)nullptr, zb_new, lam_new, 0.0,
                   dt * c->rho[j], B, fuse ? z[j] : (const T2*)nullptr, &c->oparts, j == 1 ? (const T2*)c->lam[lcur].p : (const T2*)nullptr);
      if (fuse)
        for (size_t gi = 0; gi < c->hgroups.size(); ++gi) {
          auto& g = c->hgroups[gi];
          if (g.ovl.empty()) continue;
          const int nblk = g.last_nblk;
          hipLaunchKernelGGL(soverlap_finish_kernel, dim3((unsigned)g.ovl.size()), dim3(256), 0, c->st,
                             (const double*)c->oparts[gi].p, nblk, (int)g.ovl.size(), (const int*)g.d_ovl_slots.p,
                             (const int*)c->d_owner.p, (const double*)c->d_wre.p, (const double*)c->d_wim.p, dt * c->rho[j],
                             (double*)c->grad_dev.p + (size_t)t * K);
          SHIP(hipGetLastError());
        }
      zb_prev = zb_new;
    }
    lcur ^= 1;
  };
  if (two_level) {
    if ((int)c->segpsi.size() < cstride) c->segpsi.resize(cstride);
    if ((int)c->segz.size() < cstride) c->segz.resize(cstride);
    if (host_ck && ncp > 0) {  // the last segment's first state comes back up (behind the copies that went down: same stream)
      SHIP(hipMemcpyAsync(c->stage.p, c->ckh[ncp - 1].p, sb, hipMemcpyHostToDevice, c->copy_st));
      SHIP(hipEventRecord(c->ev_up, c->copy_st));
    }
    for (int seg = ncp - 1; seg >= 0; --seg) {
      const int t0 = seg * cstride, t1 = std::min(S, t0 + cstride);
      if (host_ck) {  // take the state that was brought up, then let the one of the segment before follow while this one is worked on
        c->segpsi[0].ensure(sb);
        SHIP(hipStreamWaitEvent(c->st, c->ev_up, 0));
        SHIP(hipMemcpyAsync(c->segpsi[0].p, c->stage.p, sb, hipMemcpyDeviceToDevice, c->st));
        SHIP(hipEventRecord(c->ev_used, c->st));
        if (seg > 0) {
          SHIP(hipStreamWaitEvent(c->copy_st, c->ev_used, 0));
          SHIP(hipMemcpyAsync(c->stage.p, c->ckh[seg - 1].p, sb, hipMemcpyHostToDevice, c->copy_st));
          SHIP(hipEventRecord(c->ev_up, c->copy_st));
        }
      }
      auto psi_of = [&](int t) -> T2* { return (t == t0 && !host_ck) ? (T2*)c->ck[seg].p : (T2*)c->segpsi[t - t0].p; };
      ChainLink shand;
      for (int t = t0; t < t1; ++t) {  // the segment once more, iterates kept (the last step needs no psi_{t1}: iterates only)
        auto& zs = c->segz[t - t0];
        if ((int)zs.size() < m + 1) zs.resize(m + 1);
        if (!shand.set) step_coefficients(c, t);
        if (t + 1 < t1) {
          c->segpsi[t + 1 - t0].ensure(sb);
          // (handed over to the next step unless that one is the segment's last and has no chain of its own to start: m < 2)
          const bool over = xstep && (t + 2 < t1 || m >= 2);
          if (over) step_coefficients(c, t + 1, true);
          ChainLink next;
          taylor_apply(c, m, dt, 1.0, (const T2*)psi_of(t), psi_of(t + 1), B, &zs, over ? &next : nullptr, &shand, over);
          if (over) c->swap_step_sets();
          shand = next;
        } else if (chain_ok(c, B)) {
          if (m >= 2)
            taylor_chain<T>(c, m, 2, dt, 1.0, (const T2*)psi_of(t), (const T2*)psi_of(t), B, [&](int jm1) -> T2* {
              zs[jm1].ensure(sb);
              return (T2*)zs[jm1].p;
            }, &shand);
        } else {
          const T2* zj = psi_of(t);
          for (int j = m; j >= 2; --j) {
            zs[j - 1].ensure(sb);
            apply_groups(c, c->hgroups, (const T2*)c->d_coef.p, step_diag(c), zj, (const T2*)psi_of(t), (T2*)zs[j - 1].p, (T2*)nullptr, 0.0,
                         -dt * c->rho[j], B);
            zj = (const T2*)zs[j - 1].p;
          }
        }
      }
      for (int t = t1 - 1; t >= t0; --t) {
        step_coefficients(c, t);
        std::vector<const T2*> z(m + 1);
        z[m] = psi_of(t);
        for (int j = 1; j < m; ++j) z[j] = (const T2*)c->segz[t - t0][j].p;
        reverse_step(t, z);
      }
    }
  }
  for (int t = S - 1; t >= 0 && !two_level; --t) {
    step_coefficients(c, t);
    // psi_t: checkpoint, or T_m(-A_t) psi_{t+1}
    const T2* psi_t;
    ChainLink handover;
    if (t % cstride == 0) psi_t = (const T2*)c->ck[t / cstride].p;
    else {
      // (in the chain form the reversal's last pass also starts the chain of psi_t's Horner iterates: one tile walk less)
      taylor_apply(c, m, dt, -1.0, (const T2*)c->psi[cur].p, (T2*)c->psi[cur ^ 1].p, B, nullptr, m >= 2 ? &handover : nullptr);
      cur ^= 1;
      psi_t = (const T2*)c->psi[cur].p;
    }
    // Horner iterates z_m = psi_t, ..., z_1 (z_0 = psi_{t+1} is not needed)
    std::vector<const T2*> z(m + 1);
    z[m] = psi_t;
    if (m >= 2 && chain_ok(c, B)) {
      taylor_chain<T>(c, m, 2, dt, 1.0, psi_t, psi_t, B, [&](int jm1) -> T2* {
        c->zs[jm1].ensure(sb);
        z[jm1] = (const T2*)c->zs[jm1].p;
        return (T2*)c->zs[jm1].p;
      }, &handover);
    } else {
      const T2* zj = psi_t;
      for (int j = m; j >= 2; --j) {
        c->zs[j - 1].ensure(sb);
        T2* dst = (T2*)c->zs[j - 1].p;
        apply_groups(c, c->hgroups, (const T2*)c->d_coef.p, step_diag(c), zj, psi_t, dst, (T2*)nullptr, 0.0, -dt * c->rho[j], B);
        z[j - 1] = dst;
        zj = dst;
      }
    }
    reverse_step(t, z);
    if (t % cstride == 0 && t > 0) {
      // next iteration rebuilds psi_{t-1} from psi_t = this checkpoint
      SHIP(hipMemcpyAsync(c->psi[cur].p, c->ck[t / cstride].p, sb, hipMemcpyDeviceToDevice, c->st));
    }
  }
  if (grad_u && S > 0 && K > 0) SHIP(hipMemcpyAsync(grad_u, c->grad_dev.p, (size_t)S * K * 8, hipMemcpyDefault, c->st));
  SHIP(hipStreamSynchronize(c->st));
}

template <class F>
int s_guard(F&& f) {
  try { f(); return DQ_OK; }
  catch (const SErr& e) { s_err = e.msg; return e.code; }
  catch (const std::exception& e) { s_err = e.what(); return DQ_ERR_HIP; }
}
template <class F>
int s_with(dq_sctx* c, F&& f) {
  return s_guard([&] {
    if (!c) sfail(DQ_ERR_INVALID, "null context");
    if (c->dtype == DQ_C128) f(static_cast<SCtx<double>*>(c));
    else f(static_cast<SCtx<float>*>(c));
  });
}

}  // namespace

extern "C" {

const char* dq_sctx_last_error(void) { return s_err.c_str(); }

int dq_sctx_create(dq_sctx** out, int n_qubits, int n_controls, int device_id, int dtype) {
  return s_guard([&] {
    if (!out) sfail(DQ_ERR_INVALID, "null output");
    *out = nullptr;
    if (n_qubits < 1 || n_qubits > 30) sfail(DQ_ERR_INVALID, "n_qubits must be in [1, 30]");
    if (n_controls < 0 || n_controls > 4096) sfail(DQ_ERR_INVALID, "n_controls must be in [0, 4096]");
    if (dtype != DQ_C128 && dtype != DQ_C64) sfail(DQ_ERR_INVALID, "dtype must be DQ_C128 or DQ_C64");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
      (void)hipGetLastError();
      sfail(DQ_ERR_NO_DEVICE, "no HIP device available (this library has no CPU path)");
    }
    if (device_id < 0 || device_id >= ndev) sfail(DQ_ERR_INVALID, "device_id out of range");
    SHIP(hipSetDevice(device_id));
    auto init = [&](auto* c) {
      c->n = n_qubits; c->d = 1u << n_qubits; c->K = n_controls; c->dev = device_id;
      int ncu = 0;
      if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device_id) == hipSuccess && ncu > 0) c->n_cu = ncu;
      else (void)hipGetLastError();
      *out = c;
    };
    if (dtype == DQ_C128) init(new SCtx<double>()); else init(new SCtx<float>());
  });
}

int dq_sctx_destroy(dq_sctx* c) {
  if (!c) return DQ_OK;
  return s_with(c, [&](auto* t) { (void)hipSetDevice(t->dev); (void)hipStreamSynchronize(t->st); delete t; });
}

int dq_sctx_set_stream(dq_sctx* c, void* s) {
  return s_with(c, [&](auto* t) { SHIP(hipSetDevice(t->dev)); SHIP(hipStreamSynchronize(t->st)); t->st = (hipStream_t)s; });
}

int dq_sctx_set_terms(dq_sctx* c, int n_terms, const int* owner, const unsigned* xmask, const unsigned* zmask,
                      const double* weight) {
  return s_with(c, [&](auto* t) { set_terms(t, n_terms, owner, xmask, zmask, weight); });
}

int dq_sctx_set_option(dq_sctx* c, const char* name, double v) {
  return s_with(c, [&](auto* t) {
    if (!name) sfail(DQ_ERR_INVALID, "null option name");
    const std::string s(name);
    if (s == "tol") { if (!(v > 0)) sfail(DQ_ERR_INVALID, "tol must be > 0"); t->tol = v; }
    else if (s == "force_degree") t->force_degree = (int)v;
    else if (s == "polynomial") t->economize = v != 0 ? 1 : 0;   // 1 (default): Chebyshev-economized step polynomial, 0: Taylor
    else if (s == "chain_tol") { if (!(v > 0)) sfail(DQ_ERR_INVALID, "chain_tol must be > 0"); t->chain_tol = v; }
    else if (s == "checkpoint_bytes") t->checkpoint_bytes = v;
    else if (s == "two_level") t->two_level = v != 0 ? 1 : 0;
    else if (s == "host_checkpoints") t->host_checkpoints = v != 0 ? 1 : 0;
    else if (s == "host_checkpoint_bytes") { if (v < 0) sfail(DQ_ERR_INVALID, "host_checkpoint_bytes must be >= 0"); t->host_checkpoint_bytes = v; }
    else if (s == "chain") t->chain = v != 0 ? 1 : 0;
    else if (s == "step_handover") t->step_handover = v != 0 ? 1 : 0;
    else if (s == "reverse_keep_last") t->reverse_keep_last = v != 0 ? 1 : 0;
    else if (s == "reverse_chain") t->reverse_chain = v != 0 ? 1 : 0;
    else if (s == "interleave_groups") t->interleave_groups = v > 0 ? (int)v : 0;
    else if (s == "plan_steps") { if (v < 0 || v > 1e9) sfail(DQ_ERR_INVALID, "plan_steps must be 0..1e9"); t->plan_steps = (int)v; }
    else if (s == "fuse_overlaps") t->fuse_overlaps = v > 0 ? 1 : (v < 0 ? -1 : 0);
    else if (s == "dma") t->dma = v != 0 ? 1 : 0;
    else if (s == "dma_workgroups") { if (v < 1 || v > 65535) sfail(DQ_ERR_INVALID, "dma_workgroups must be 1..65535"); t->n_cu = (int)v; }
    else if (s == "profile") { t->collect(); t->prof_on = v != 0; if (v < 0) { t->prof_ms = 0; t->prof_bytes = 0; t->prof_launches = 0; } }
    else sfail(DQ_ERR_INVALID, "unknown option '" + s + "'");
  });
}

int dq_sctx_get_info(dq_sctx* c, const char* name, double* v) {
  return s_with(c, [&](auto* t) {
    if (!name || !v) sfail(DQ_ERR_INVALID, "null argument");
    const std::string s(name);
    if (s == "degree") *v = t->last_degree;
    else if (s == "taylor_degree") *v = t->last_taylor_degree;
    else if (s == "theta") *v = t->last_theta;
    else if (s == "groups") *v = (double)t->hgroups.size();
    else if (s == "checkpoint_stride") *v = t->last_cstride;
    else if (s == "passes") *v = t->last_passes;
    else if (s == "dma_passes") *v = t->last_dma_passes;
    else if (s == "two_level") *v = t->last_two_level;
    else if (s == "host_checkpoints") *v = t->last_host_ck;
    else if (s == "pass_bytes") *v = t->last_bytes;
    else if (s == "pinned_host_bytes") {  // pinned host memory the kept states of the two-level plan hold now
      double h = 0.0;
      for (const auto& b : t->ckh) h += (double)b.bytes;
      *v = h;
    }
    else if (s == "checkpoint_pool_bytes") {  // device memory the checkpoint pools hold now (kept states, segment states + iterates, staging)
      double h = (double)t->stage.bytes;
      for (const auto& b : t->ck) h += (double)b.bytes;
      for (const auto& b : t->segpsi) h += (double)b.bytes;
      for (const auto& z : t->segz) for (const auto& b : z) h += (double)b.bytes;
      *v = h;
    }
    else if (s == "prof_ms") { t->collect(); *v = t->prof_ms; }
    else if (s == "prof_bytes") { t->collect(); *v = t->prof_bytes; }
    else if (s == "prof_launches") { t->collect(); *v = (double)t->prof_launches; }
    else if (s == "dtype") *v = t->dtype;
    else sfail(DQ_ERR_INVALID, "unknown info '" + s + "'");
  });
}

int dq_sctx_propagate(dq_sctx* c, const double* u, int n_steps, double dt, const void* psi_in, void* psi_out, int batch) {
  return s_with(c, [&](auto* t) { s_propagate(t, u, n_steps, dt, psi_in, psi_out, batch); });
}

int dq_sctx_energy_grad(dq_sctx* c, const double* u, int n_steps, double dt, const void* psi0, int batch, double* loss,
                        double* grad_u, void* psi_final, double* energies) {
  return s_with(c, [&](auto* t) { s_energy_grad(t, u, n_steps, dt, psi0, batch, loss, grad_u, psi_final, energies); });
}

/* out = H(u_row) in  (which = 0)  or  out = M in  (which = 1): one matrix-free application, for tests. */
int dq_sctx_apply(dq_sctx* c, int which, const double* u_row, const void* in, void* out, int batch) {
  return s_with(c, [&](auto* t) {
    typedef typename std::remove_pointer<decltype(t)>::type::T2 T2;
    typedef decltype(((T2*)nullptr)->x) T;
    s_check(t);
    if (!in || !out || batch <= 0) sfail(DQ_ERR_INVALID, "bad arguments");
    const size_t sb = t->state_bytes(batch);
    t->psi[0].ensure(sb); t->psi[1].ensure(sb);
    SHIP(hipMemcpyAsync(t->psi[0].p, in, sb, hipMemcpyDefault, t->st));
    if (which == 0) {
      std::vector<double> uh = s_fetch_u(t, u_row, 1);
      step_coefficients(t, 0);
      apply_groups(t, t->hgroups, (const T2*)t->d_coef.p, step_diag(t), (const T2*)t->psi[0].p, (const T2*)nullptr,
                   (T2*)t->psi[1].p, (T2*)nullptr, 1.0, 0.0, batch);
    } else {
      apply_groups(t, t->mgroups, (const T2*)t->d_coefM.p, t->has_DM ? (const T*)t->DM.p : (const T*)nullptr,
                   (const T2*)t->psi[0].p, (const T2*)nullptr, (T2*)t->psi[1].p, (T2*)nullptr, 1.0, 0.0, batch);
    }
    SHIP(hipMemcpyAsync(out, t->psi[1].p, sb, hipMemcpyDefault, t->st));
    SHIP(hipStreamSynchronize(t->st));
  });
}

}  // extern "C"
