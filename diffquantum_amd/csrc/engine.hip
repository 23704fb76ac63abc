// Host-side engine of libdiffqc_hip.so: context, workspace, the matrix-exponential plan
// (scaled Paterson-Stockmeyer Taylor polynomial + squarings, built only from ZGEMMs), the
// forward propagator chain and the adjoint (reverse-mode) sweep; plus the extern "C" ABI of
// include/diffqc_hip.h.
//
// What it replaces in the reference (/root/reference):
//   set_H globals                 diffqc.cc:21-25,43-73        -> dq_ctx + dq_set_hamiltonians
//   dH assembly                   diffqc.cc:192-197            -> assemble_launch (batched over steps)
//   dH.exp()                      diffqc.cc:198 (Eigen Pade)   -> expm_forward (Taylor/PS on f64 MFMA)
//   psi = U psi, time loop        diffqc.cc:190-200            -> dq_propagate (psi is dim x batch)
//   sim_plain.py:135-150          scipy expm + np.matmul       -> same
//   gradient                      (reference: MC estimator sim_plain.py:156-231; no adjoint)
//                                                              -> dq_energy_grad (exact discrete adjoint)
// No CPU fallback exists anywhere in this file.
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

thread_local std::string g_err;

struct DqError {
  int code;
  std::string msg;
};

[[noreturn]] void fail(int code, const std::string& msg) { throw DqError{code, msg}; }

#define HIPCK(expr)                                                                              \
  do {                                                                                           \
    hipError_t e__ = (expr);                                                                     \
    if (e__ != hipSuccess) {                                                                     \
      (void)hipGetLastError();                                                                   \
      fail(e__ == hipErrorOutOfMemory ? DQ_ERR_NOMEM : DQ_ERR_HIP,                               \
           std::string(#expr) + " failed: " + hipGetErrorString(e__));                           \
    }                                                                                            \
  } while (0)

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  DevBuf(DevBuf&& o) noexcept : p(o.p), bytes(o.bytes) { o.p = nullptr; o.bytes = 0; }
  ~DevBuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
  }
  void ensure(size_t n) {
    if (n <= bytes) return;
    release();
    HIPCK(hipMalloc(&p, n));
    bytes = n;
  }
  double2* z() const { return (double2*)p; }
  double* d() const { return (double*)p; }
};

bool is_device_ptr(const void* p) {
  if (!p) return false;
  hipPointerAttribute_t at;
  memset(&at, 0, sizeof(at));
  hipError_t e = hipPointerGetAttributes(&at, p);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return at.type == hipMemoryTypeDevice || at.type == hipMemoryTypeManaged;
}

// ---- exp(A) plan ---------------------------------------------------------------------------
// U = p(A/2^s)^(2^s) with p the degree-m Taylor polynomial of exp, written as a short TAPE of two op kinds
//   LIN: dst = cI*I + sum_j c_j * src_j                       (streaming pass)
//   MUL: dst = a * b + cI*I + sum_j c_j * src_j                (one ZGEMM, the linear part pre-stored in dst)
// over matrix variables (variable 0 is A itself). Two evaluation schemes produce tapes:
//   * Paterson-Stockmeyer blocks of size q (m = (r+1) q): P_j = A^j, R_r = B_r, R_i = P_q R_{i+1} + B_i with
//     B_i = sum_j c_{iq+j} P_j : (q-1) + r products;
//   * the degree-8 scheme of Bader, Blanes & Casas (Mathematics 7(12):1174, 2019) with 3 products:
//       A2 = A A;  A4 = A2 (x1 A + x2 A2);  A8 = (x3 A2 + A4)(x4 I + x5 A + x6 A2 + x7 A4);
//       T8 = y0 I + y1 A + y2 A2 + A8      (coefficients checked symbolically: all nine Taylor terms exact)
// followed by s squarings. The scaling 2^-s is folded into the coefficients. The degree is the cheapest whose
// Taylor remainder theta^(m+1)/(m+1)! is below tol (default 2^-53) for theta >= ||A/2^s||_1.
// The same tape drives the reverse mode (tape_backward): for C = X Y, Xbar += Cbar Y^H and Ybar += X^H Cbar.
struct TapeOp {
  int kind;  // 0 = LIN, 1 = MUL
  int herm;  // MUL: +1 / -1 if the product is Hermitian / anti-Hermitian whenever A is anti-Hermitian (even / odd
             // polynomial in A with real coefficients), 0 otherwise -> (anti-)Hermitian output form of the ZGEMM
  int apar;  // MUL: the same parity of the LEFT operand (it can then be read k-major: linear-form ZGEMM)
  int dst, a, b;
  int nlin;
  int lsrc[6];
  double lc[6];
  double cI;
};

struct ExpmPlan {
  int m = 0, s = 0, nvars = 0, out = 0, nprod = 0;
  int rq = 0;  // > 0: real-symmetric plan (cos/sin polynomials in Y = X^2 up to Y^rq); `ops` is empty then
  int spx = 0; // real plan evaluated with the sparse generator: powers X_{m+1} = X X_m by sparse x dense products
  int xr = 0;  // spx plan whose generator is XOR-regular: column form in LDS (xorexp.hip), no stored powers
  int zxr = 0; // complex Hermitian XOR-regular generator: U = sum_{l <= m} A^l / l! row by row in LDS (zxr_* kernels); no tape
  double cc[8] = {0}, sc[8] = {0};  // cos X = sum cc[j] Y^j,  sin X = X sum sc[j] Y^j
  std::vector<TapeOp> ops;
  int products() const { return nprod; }
};

// candidate schemes ordered by cost: {degree, PS block size q (0 = BBC scheme), products}
const int kCand[][3] = {{2, 1, 1}, {4, 2, 2}, {8, 0, 3}, {6, 2, 3}, {9, 3, 4}, {12, 3, 5}, {16, 4, 6}, {20, 4, 7}};
constexpr int kNCand = 8;

double taylor_remainder(double theta, int m) {
  double v = 1.0;
  for (int j = 1; j <= m + 1; ++j) v *= theta / j;
  return v;
}

TapeOp op_lin(int dst, double cI) {
  TapeOp o;
  memset(&o, 0, sizeof(o));
  o.kind = 0; o.dst = dst; o.a = o.b = -1; o.cI = cI;
  return o;
}
TapeOp op_mul(int dst, int a, int b, double cI = 0.0) {
  TapeOp o;
  memset(&o, 0, sizeof(o));
  o.kind = 1; o.dst = dst; o.a = a; o.b = b; o.cI = cI;
  return o;
}
void add_term(TapeOp& o, int src, double c) {
  if (c == 0.0) return;
  o.lsrc[o.nlin] = src;
  o.lc[o.nlin] = c;
  o.nlin++;
}

ExpmPlan make_plan(int m, int q, int s) {
  ExpmPlan p;
  p.m = m;
  p.s = s;
  const double sg = std::ldexp(1.0, -s);
  int nv = 1;  // variable 0 = A
  if (q == 0) {  // Bader-Blanes-Casas degree 8
    const double r177 = std::sqrt(177.0);
    const double x3 = 2.0 / 3.0;
    const double x1 = x3 * (1.0 + r177) / 88.0, x2 = x3 * (1.0 + r177) / 352.0;
    const double x4 = (-271.0 + 29.0 * r177) / (315.0 * x3), x5 = 11.0 * (-1.0 + r177) / (1260.0 * x3);
    const double x6 = 11.0 * (-9.0 + r177) / (5040.0 * x3), x7 = (89.0 - r177) / (5040.0 * x3 * x3);
    const double y2 = (857.0 - 58.0 * r177) / 630.0;
    const double s1 = sg, s2 = sg * sg;
    const int v1 = nv++, t1 = nv++, v2 = nv++, t2 = nv++, t3 = nv++, u0 = nv++;
    p.ops.push_back(op_mul(v1, 0, 0));                       // v1 = A^2            (A'^2 = s2 v1)
    TapeOp o = op_lin(t1, 0.0);                              // t1 = s2 (x1 A' + x2 A'^2)
    add_term(o, 0, s2 * x1 * s1); add_term(o, v1, s2 * x2 * s2);
    p.ops.push_back(o);
    p.ops.push_back(op_mul(v2, v1, t1));                     // v2 = A4'
    o = op_lin(t2, 0.0);                                     // t2 = x3 A'^2 + A4'
    add_term(o, v1, x3 * s2); add_term(o, v2, 1.0);
    p.ops.push_back(o);
    o = op_lin(t3, x4);                                      // t3 = x4 I + x5 A' + x6 A'^2 + x7 A4'
    add_term(o, 0, x5 * s1); add_term(o, v1, x6 * s2); add_term(o, v2, x7);
    p.ops.push_back(o);
    o = op_mul(u0, t2, t3, 1.0);                             // U0 = t2 t3 + I + A' + y2 A'^2
    add_term(o, 0, s1); add_term(o, v1, y2 * s2);
    p.ops.push_back(o);
    p.out = u0;
  } else {  // Paterson-Stockmeyer
    const int r = m / q - 1;
    double c[32];
    double f = 1.0, pw = 1.0;
    for (int j = 0; j <= m; ++j) {
      if (j > 0) { f *= j; pw *= sg; }
      c[j] = pw / f;
    }
    int P[8];
    P[1] = 0;
    for (int j = 2; j <= q; ++j) {
      P[j] = nv++;
      p.ops.push_back(op_mul(P[j], P[j - 1], P[1]));
    }
    int R[8];
    for (int i = r; i >= 0; --i) R[i] = nv++;
    TapeOp o = op_lin(R[r], c[r * q]);
    for (int j = 1; j <= q; ++j) add_term(o, P[j], c[r * q + j]);
    p.ops.push_back(o);
    for (int i = r - 1; i >= 0; --i) {
      o = op_mul(R[i], P[q], R[i + 1], c[i * q]);
      for (int j = 1; j <= q - 1; ++j) add_term(o, P[j], c[i * q + j]);
      p.ops.push_back(o);
    }
    p.out = R[0];
  }
  for (int j = 0; j < s; ++j) {
    const int nx = nv++;
    p.ops.push_back(op_mul(nx, p.out, p.out));
    p.out = nx;
  }
  p.nvars = nv;
  // parity of every variable as a polynomial in A: +1 even (Hermitian for anti-Hermitian A), -1 odd (anti-Hermitian),
  // 0 mixed. Products of two pure-parity polynomials (they commute) are pure again.
  std::vector<int> par(nv, 0);
  par[0] = -1;
  for (auto& o : p.ops) {
    int lp = 2;  // parity of the linear part (2 = empty)
    auto merge = [&](int x) { lp = (lp == 2 || lp == x) ? x : 0; };
    if (o.cI != 0.0) merge(+1);
    for (int j = 0; j < o.nlin; ++j) merge(par[o.lsrc[j]]);
    if (o.kind == 0) {
      par[o.dst] = lp == 2 ? 0 : lp;
    } else {
      int pp = par[o.a] * par[o.b];
      if (lp != 2 && lp != pp) pp = 0;
      par[o.dst] = pp;
      o.herm = pp;
      o.apar = par[o.a];
    }
    p.nprod += o.kind;
  }
  return p;
}

ExpmPlan choose_plan(double theta, double tol, int force_degree, int force_s) {
  if (!(theta >= 0.0) || !std::isfinite(theta)) fail(DQ_ERR_INVALID, "non-finite generator norm");
  int bm = 0, bq = 0, bs = 0, best_cost = 1 << 30;
  for (int s = 0; s <= 60; ++s) {
    if (force_s >= 0 && s != force_s) continue;
    const double th = std::ldexp(theta, -s);
    for (int i = 0; i < kNCand; ++i) {
      const int m = kCand[i][0];
      if (force_degree > 0 && m != force_degree) continue;
      if (force_degree <= 0 && taylor_remainder(th, m) > tol) continue;
      const int cost = kCand[i][2] + s;
      if (cost < best_cost) { best_cost = cost; bm = m; bq = kCand[i][1]; bs = s; }
      break;  // candidates are ordered by cost within one s
    }
    if (force_s < 0 && best_cost <= s + 1) break;  // larger s cannot win any more
  }
  if (bm == 0) fail(DQ_ERR_INVALID, "no exp(A) plan for the requested degree/squarings");
  return make_plan(bm, bq, bs);
}

// ---- profiling -----------------------------------------------------------------------------
struct ProfRec {
  int cls;
  hipEvent_t e0, e1;
  double flops, bytes;
};

}  // namespace

struct dq_ctx {
  int dtype = DQ_C128;
  virtual ~dq_ctx() {}
};

namespace {

template <class T>
struct Ctx : dq_ctx {
  typedef typename dq::Cx<T>::type T2;
  int d = 0, K = 0, dev = 0, n_cu = 256;
  size_t max_lds = 160 * 1024;  // dynamic LDS a workgroup may ask for on this device (the column forms fall back when it is too small)
  hipStream_t st = nullptr;
  // cache of the real path's per-chunk propagator intermediates (X, Y_1..Y_{q-1}, P, U): what the backward sweep would
  // otherwise rebuild with q + 1 products per step; filled in the forward pass as far as the budget reaches
  struct PCacheEntry { size_t off = 0; int c0 = -1, n = 0, rq = 0; bool valid = false; };
  std::vector<PCacheEntry> pcache;
  DevBuf pc_arena;                  // one allocation; entries are offsets into it (filled front to back each call)
  size_t pc_used = 0;
  void* redirect[96] = {nullptr};   // pool slots currently backed by a cache entry (set per chunk by build_propagators)
  int pc_mode = 0, pc_tc = 1;       // 0 off, 1 store (forward), 2 load (backward)
  long pc_limit = 0;                // steps [0, pc_limit) are cached
  double cache_bytes = 0.0;         // option: < 0 off, 0 auto (half of the free HBM), > 0 bytes
  double last_cached = 0;
  // time-sliced operator form (dq_slice_*): what dq_slice_forward left in place for dq_slice_backward
  std::vector<double> slice_u;
  double slice_dt = 0.0;
  int slice_tc = 0;
  bool slice_valid = false;
  DevBuf rws, zws;  // split-K workspaces of the symmetric real GEMM / the (anti-)Hermitian complex GEMM
  DevBuf H0, Hs, M;
  DevBuf H0r, Hsr;  // real parts (real-symmetric fast path)
  bool have_H = false, have_M = false, hermitian = false, real_h = false;
  // sparse control Hamiltonians (real path; realpath.hip): the nonzeros of all H_k in one CSR by row, and the same
  // entries grouped by control; sparse_mode: -1 auto (total nonzeros <= sparse_fill * dim), 0 off
  DevBuf zta, ztb;  // transposed operand copies for the linear-form complex GEMM
  DevBuf sp_rowptr, sp_col, sp_kid, sp_val, sp_kptr, sp_kent, sp_nzval, sp_nzacc, sp_part;
  // XOR-regular generator (xorexp.hip): table found by build_sparse_controls; xor_mode: -1 auto, 0 off
  DevBuf xr_sgn, xr_dval, xr_part, xr_pow;
  dq::XrTab xr_tab;
  bool xr_ok = false;
  dq::XrTab zxr_tab;    // complex Hermitian generators (phases +-1 / +-i): table of the zxr_* kernels
  bool zxr_ok = false;
  int xor_mode = -1;
  double last_xr = 0;
  // state-column form (statecol.hip): whole chains on state columns held in LDS; xs_mode: -1 auto, 0 off, 1 whenever supported
  DevBuf xs_psi, xs_lam, xs_park, xs_part, xs_wt, xs_dgs, m_diag;
  bool m_is_diag = false;  // the observable is diagonal (set_observable): Lambda_S = 2 M Psi_S elementwise on the state columns
  int xs_mode = -1, xs_variant = 0;  // variant 0: 512 threads x 8 rows at dim 4096, no partner prefetch (measured best with the branch-free sweeps)
  int xs_min_dim = 512;
  double xs_batch_ratio = 8.0;  // auto: batch <= ratio x dim (beyond it the operator form, whose cost does not grow with the batch, wins)
  double last_xs = 0;
  // adjoint sweep of the state-column form: -1 auto / 1 = the quadrature form (xs_backward_q_kernel) whenever a rule the kernels are
  // built for keeps its error bound below xs_quad_tol at this chain's theta, 0 = always the exact derivative of the polynomial
  int xs_adjoint = -1;
  // Matrix-free routing (structured.hip): a Pauli-sum generator whose dimension the state-column kernels do not take (complex128
  // above dim 4096, complex64 above 8192) runs on the matrix-free passes INSIDE this context -- the strings are read off the
  // XOR-regular table, the caller keeps the dense API (dq_set_hamiltonians accepts any dim: diffqc.cc:43-67).  mf_mode: -1 auto,
  // 0 never, 1 whenever the table is a sum of Pauli strings (tests at small dim)
  dq_sctx* mf = nullptr;
  int mf_mode = -1;
  bool mf_dirty = true, mf_possible = false, mf_has_obs = false;
  double last_mf = 0, mf_checkpoint_bytes = 0;
  std::vector<unsigned> tab_sgn_h;             // host copies of the table's sign words and offset-0 diagonals (find_*xor_regular)
  std::vector<double> tab_dval_h;
  int xs_workgroups = 0;                       // > 0: at most that many workgroups per launch (tests of the several-columns-per-workgroup loop)
  double xs_part_bytes = 268435456.0;          // scratch of the adjoint sweep's per-step partial sums: the chain is swept in segments that fit
  double xs_quad_tol = sizeof(T) == 8 ? 9.094947017729282e-13 : 1e-7;  // complex128: 2^-40, relative to ||dA_k|| ||lambda|| ||psi|| per step; complex64: at fp32 rounding (the
                                                                       // 4-point rule then serves steps up to theta = 0.37 instead of 0.05)
  int xs_wide_mode = -1;                       // energy gradients of batches wider than the dimension on dim columns (do_energy_grad_xs_wide): -1 by cost (default), 0 off, 1 on
  double last_xs_wide = 0;
  int xs_compose_mode = -1;                    // products of a chain's propagators (composed forward, time slices) from state columns: -1 auto, 0 off
  double last_xs_segments_fwd = 0;             // time segments of the last such product (0: built from explicit propagators)
  int xs_economize = 1;                        // step polynomial of the state-column kernels: 1 = Chebyshev-economized (default), 0 = Taylor
  // the polynomial's error is the same function of the spectrum at every step, so over a chain it can add up coherently: its per-step
  // bound is min(expm_tol, xs_chain_tol / steps) -- no effect in fp64 below 90000 steps, one degree more in fp32 beyond ~330 steps
  double xs_chain_tol = sizeof(T) == sizeof(double) ? 1e-11 : 2e-5;
  dq::XsPoly xs_poly;                          // its level factors (set by xs_degree for the chain at hand)
  double last_xs_taylor_degree = 0;            // the Taylor degree the same tolerance would have needed
  double last_xs_segments = 0;                 // launches the last adjoint sweep was split into
  double last_xs_nodes = 0;                    // interior nodes of the rule the last adjoint sweep ran (0: the exact form)
  // centred diagonal of the state-column kernels (dq_internal.h, xs_center_launch): 1 (default) on, 0 off
  int xs_center = 1;
  double last_xs_centered = 0;                 // 1 when the last state-column call ran in the centred frame
  double last_xs_theta_loose = 0;              // the step bound from the sources' norms (what the degree came from before round 5)
  DevBuf xs_cen;
  // range of the drift's diagonal when it is the only offset-0 source (xs_prepare then needs no read-back): (max + min) / 2, (max - min) / 2
  bool xs_drift_range_ok = false;
  double xs_drift_mid = 0, xs_drift_half = 0;
  std::vector<double> uh_last;                 // host copy of the last pulse table (fetch_u)
  int fuse_levels = 1;  // sparse-generator reverse mode: gradient levels fused into the X Xbar passes (0: separate kernel)
  long sp_nnz = 0, sp_nnz_c = 0;  // entries in the CSR / those of them that belong to controls
  bool sparse_h = false;   // a CSR exists (controls, and the drift too if sparse_x)
  bool sparse_x = false;   // the drift H0 is in the CSR as well (entries with kid = -1): X_t itself is sparse
  int sparse_mode = -1, sparse_gen_mode = -1;
  double sparse_fill = 32.0;
  int real_mode = -1, rgemm_cfg = 0, real_min_dim = 128, rgemm_glds = 1;
  long plan_steps = 0;
  int small_dim = 512, time_parallel = -1;  // see do_energy_grad  // real_mode: -1 auto (dim >= real_min_dim), 0 off, q >= 1 forced
  double last_real_q = 0, last_spx = 0;
  double nH0 = 0.0;
  std::vector<double> nHk;
  // options
  double checkpoint_bytes = 0.0, chunk_bytes = 0.0, expm_tol = 0.0, herm_tol = 0.0;
  int slab_spmm = 1, transpose_fallback = 1;
  int force_degree = 0, force_s = -1, force_tile = 0, zgemm_alg = 2, compose_mode = -1, skew_trick = 1;
  // info
  double last_degree = 0, last_squarings = 0, last_chunk = 0, last_cstride = 0, last_theta = 0, last_products = 0, last_composed = 0, last_opback = 0, last_tpar = 0;
  // workspace
  std::vector<DevBuf> pool;
  size_t pool_tc = 0;
  DevBuf u_dev, grad_dev, part, scratch, e_dev, psi[3], lam[2], tmp_in[4];  // tmp_in[3]: staged Psi_0 of the operator-form sweep
  std::vector<DevBuf> ck;
  // profiling
  bool prof_on = false;
  std::vector<ProfRec> recs;
  std::vector<hipEvent_t> ev_free;
  dq_profile prof_acc;

  Ctx() : pool(96) {
    memset(&prof_acc, 0, sizeof(prof_acc));
    dtype = sizeof(T) == 8 ? DQ_C128 : DQ_C64;
    expm_tol = sizeof(T) == 8 ? 1.1102230246251565e-16 : 5.9604644775390625e-08;
    herm_tol = sizeof(T) == 8 ? 1e-12 : 1e-5;
  }
  ~Ctx() override {
    if (mf) (void)dq_sctx_destroy(mf);
    for (auto& r : recs) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    for (auto e : ev_free) (void)hipEventDestroy(e);
  }

  bool sparse_on() const { return sparse_h && sparse_mode != 0; }
  // sparse-generator scheme: forced (1), or auto (-1) where a sparse x dense pass (nnz/row partner rows per output row,
  // HBM/L2-bound) beats a symmetric-output MFMA product: measured 0.225 vs 1.05 ms at d = 4096 with 13 nonzeros per row,
  // break-even near d = 80 nnz/row (at d = 256 the MFMA product wins 6x)
  bool sparse_gen_on() const {
    if (!(sparse_on() && sparse_x) || sparse_gen_mode == 0) return false;
    if (sparse_gen_mode > 0 || (double)d * d >= 64.0 * (double)sp_nnz) return true;
    // (XOR-regular generators: the column form, xorexp.hip, beats the MFMA products wherever its kernels exist)
    return xr_ok && xor_mode != 0 && dq::xr_supported(d, 1, xr_tab.R, xr_tab.nd, sizeof(T));
  }
  // (the column forms need up to 4 real / 2 complex images of a column in LDS plus tables: a device or partition mode that
  // grants less falls back to the row-wise / tape schemes instead of failing at launch)
  bool column_lds_ok() const {
    const size_t images = std::max<size_t>((size_t)4 * sizeof(T) * d, sizeof(T) == 8 ? (size_t)32 * d : 0);
    return images + (size_t)d + 1024 <= max_lds;
  }
  bool zxr_on() const {
    return sizeof(T) == 8 && zxr_ok && xor_mode != 0 && real_mode != 0 && force_degree <= 0 && force_s < 0 && column_lds_ok() &&
           dq::zxr_supported(d, 1, zxr_tab.R, zxr_tab.nd);
  }
  bool xr_on(int q) const {
    return xr_ok && xor_mode != 0 && sparse_gen_on() && dq::xr_supported(d, q, xr_tab.R, xr_tab.nd, sizeof(T)) && column_lds_ok();
  }
  size_t nn() const { return (size_t)d * d; }
  size_t mat_bytes() const { return nn() * sizeof(T2); }

  hipEvent_t get_event() {
    if (!ev_free.empty()) { hipEvent_t e = ev_free.back(); ev_free.pop_back(); return e; }
    hipEvent_t e;
    HIPCK(hipEventCreate(&e));
    return e;
  }
  template <class F>
  void timed(int cls, double flops, double bytes, F&& launch) {
    if (!prof_on) {
      HIPCK(launch());
      return;
    }
    ProfRec r{cls, get_event(), get_event(), flops, bytes};
    HIPCK(hipEventRecord(r.e0, st));
    HIPCK(launch());
    HIPCK(hipEventRecord(r.e1, st));
    recs.push_back(r);
  }
  void prof_collect() {
    if (recs.empty()) return;
    HIPCK(hipStreamSynchronize(st));
    for (auto& r : recs) {
      float ms = 0.f;
      HIPCK(hipEventElapsedTime(&ms, r.e0, r.e1));
      prof_acc.launches[r.cls] += 1;
      prof_acc.ms[r.cls] += ms;
      prof_acc.flops[r.cls] += r.flops;
      prof_acc.bytes[r.cls] += r.bytes;
      ev_free.push_back(r.e0);
      ev_free.push_back(r.e1);
    }
    recs.clear();
  }

  // pool slot of `tc` stacked d x d matrices
  T2* mat(int slot, size_t tc) {
    if (redirect[slot]) return (T2*)redirect[slot];  // this chunk's variable lives in the propagator cache
    pool[slot].ensure(tc * mat_bytes());
    return (T2*)pool[slot].p;
  }

  // herm: 0 general output; +1 / -1 the product is known to be Hermitian / anti-Hermitian (upper tiles + mirror)
  // opA: 0 = as stored, 1 = conjugate transpose; 2 / 3 / 4 = as stored AND known to be complex symmetric / Hermitian /
  // anti-Hermitian (a hint: the kernel may then read A k-major -- ZgemmArgs::hintA)
  void zgemm(int opA, int opB, int M, int N, int Kd, double alpha, const T2* A, long lda, long sA,
             const T2* B, long ldb, long sB, double beta, T2* C, long ldc, long sC, int batch, int herm = 0) {
    dq::ZgemmArgs<T> g;
    g.A = A; g.B = B; g.C = C;
    g.M = M; g.N = N; g.K = Kd;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.sA = sA; g.sB = sB; g.sC = sC;
    g.batch = batch;
    g.alpha = alpha; g.beta = beta;
    g.opA = opA == 1 ? 1 : 0; g.opB = opB;
    g.hintA = opA >= 2 ? opA - 1 : 0;
    g.herm = herm;
    g.ws = nullptr;
    g.ws_elems = 0;
    if (herm && batch == 1 && M >= 1024) {  // K-split tail of the triangle: at most one partial tile per workgroup slot
      const size_t elems = (size_t)n_cu * 2 * 128 * 128;
      zws.ensure(elems * sizeof(T2));
      g.ws = (T2*)zws.p;
      g.ws_elems = elems;
    }
    // Operands without a k-major form (A as stored with no symmetry hint; B^H) get one by a transposing pre-pass into scratch,
    // so that the product runs on the linear-form kernel too (zgemm_lin.hip): 0.12 ms per 268 MB matrix against the ~0.5 ms the
    // 16-wave kernel saves on a d = 4096 product. Only for big single products (fp64, whole 128-tiles, K >= 1024).
    if (sizeof(T) == 8 && zgemm_alg == 2 && transpose_fallback && (force_tile == 0 || force_tile == 128) && batch == 1 &&
        M % 128 == 0 && N % 128 == 0 && Kd % 16 == 0 && Kd >= 1024 && (long)(M / 128) * (N / 128) >= 64 &&
        ((g.opA == 0 && g.hintA == 0) || g.opB == 1)) {
      if (g.opB == 1) {  // B stored [N][K]: B^H -> [K][N]
        ztb.ensure((size_t)Kd * N * sizeof(T2));
        const T2* Bsrc = B;
        const long ldb0 = ldb;
        T2* Bt = (T2*)ztb.p;
        timed(DQ_PROF_LINCOMB, 0.0, 2.0 * sizeof(T2) * (double)Kd * N,
              [&] { return dq::ztranspose_launch<T>(Bt, N, 0, Bsrc, ldb0, 0, N, Kd, 1, 1, st); });
        g.B = Bt; g.ldb = N; g.opB = 0;
      }
      if (g.opA == 0 && g.hintA == 0) {  // A stored [M][K]: A^T -> [K][M], read k-major without conjugation (hint 1: A^T of A^T)
        zta.ensure((size_t)M * Kd * sizeof(T2));
        const T2* Asrc = A;
        const long lda0 = lda;
        T2* At = (T2*)zta.p;
        timed(DQ_PROF_LINCOMB, 0.0, 2.0 * sizeof(T2) * (double)M * Kd,
              [&] { return dq::ztranspose_launch<T>(At, M, 0, Asrc, lda0, 0, M, Kd, 0, 1, st); });
        g.A = At; g.lda = M; g.hintA = 5;  // 5: "this buffer holds A^T": the launcher reads it k-major as it stands
      }
    }
    // algorithmic flops of the product that is delivered (the herm form executes about half of them)
    const double fl = 8.0 * M * N * (double)Kd * batch;
    const double by = 2.0 * sizeof(T) * ((double)M * Kd + (double)Kd * N + (double)M * N * (beta != 0.0 ? 2 : 1)) * batch;
    const int ft = force_tile, alg = zgemm_alg, ncu = n_cu;
    timed(herm ? DQ_PROF_ZHERM : DQ_PROF_ZGEMM, fl, by, [&] { return dq::zgemm_launch<T>(g, st, ft, alg, ncu); });
  }
  // batched d x d product on pool matrices
  void mm(int opA, int opB, const T2* A, const T2* B, T2* C, double beta, int batch, int herm = 0) {
    const long s = (long)nn();
    zgemm(opA, opB, d, d, d, 1.0, A, d, s, B, d, s, beta, C, d, s, batch, herm);
  }
  // real product of `batch` stacked d x d matrices; A must be symmetric (it is read as A^T: the linear staging form)
  void rmm(const T* A, const T* B, T* C, int batch, bool sym_out, bool two = false, double alpha = 1.0,
           bool accumulate = false) {
    dq::RgemmArgs<T> g;
    g.A = A; g.B = B; g.C = C;
    g.M = g.N = g.K = d;
    g.lda = g.ldb = g.ldc = d;
    g.sA = g.sB = g.sC = (long)nn();
    g.batch = batch;
    g.alpha = alpha; g.beta = accumulate ? 1.0 : 0.0;
    g.opA = 1; g.opB = 0;
    g.sym = sym_out ? 1 : 0;
    g.two = (sym_out && two) ? 1 : 0;
    g.glds_ok = rgemm_glds;
    rgemm_workspace(g);
    const double fl = sym_out ? rsym_flops(d, d, batch) * (two ? 2.0 : 1.0) : 2.0 * d * (double)d * d * batch;
    const double by = sizeof(T) * 3.0 * nn() * batch;
    const int cfg = rgemm_cfg, ncu = n_cu;
    timed(sym_out ? DQ_PROF_RSYM : DQ_PROF_RGEMM, fl, by, [&] { return dq::rgemm_launch<T>(g, st, cfg, ncu); });
  }
  // executed flops of the symmetric-output form: tiles on/above the diagonal only
  double rsym_flops(int n, int k, int batch) const {
    const int t = rgemm_cfg == 32 || rgemm_cfg == 64 ? rgemm_cfg : (rgemm_cfg ? 128 : dq::zgemm_pick_tile(n, n, batch));
    const double nt = (n + t - 1) / t;
    return 2.0 * k * (double)t * t * (nt * (nt + 1) / 2) * batch;
  }
  // workspace for the K-split tail tiles of a large symmetric product (at most one partial tile per workgroup slot)
  void rgemm_workspace(dq::RgemmArgs<T>& g) {
    g.ws = nullptr;
    g.ws_elems = 0;
    if (!g.sym || g.batch != 1 || g.M < 1024) return;
    const size_t elems = (size_t)n_cu * 4 * 128 * 128;
    rws.ensure(elems * sizeof(T));
    g.ws = (T*)rws.p;
    g.ws_elems = elems;
  }
  T* rmat(int slot, size_t tc) {
    if (redirect[slot]) return (T*)redirect[slot];
    pool[slot].ensure(tc * nn() * sizeof(T));
    return (T*)pool[slot].p;
  }
  void lincomb(T2* out, int n_in, const T2* const* x, const double* c, double cI, int batch, bool acc) {
    dq::LincombArgs<T> a;
    memset(&a, 0, sizeof(a));
    a.out = out;
    a.n_in = n_in;
    for (int j = 0; j < n_in; ++j) { a.x[j] = x[j]; a.c[j] = c[j]; }  // n_in may be 0 (x, c null): out = cI * I
    a.cI = cI;
    a.d = d;
    a.stride = (long)nn();
    a.batch = batch;
    a.accumulate = acc ? 1 : 0;
    const double by = 2.0 * sizeof(T) * nn() * (n_in + 1 + (acc ? 1 : 0)) * batch;
    timed(DQ_PROF_LINCOMB, 2.0 * 2 * nn() * n_in * batch, by, [&] { return dq::lincomb_launch<T>(a, st); });
  }
};


// pool slot map: variable v of the tape lives in slot v, its adjoint in slot SLOT_ADJ + v
enum { SLOT_ADJ = 44, SLOT_MAXV = 44, SLOT_TP_L = 88, SLOT_TP_Z = 89, SLOT_TREE = 90, SLOT_TP_Q = 94, SLOT_TP_R = 95 };

// Forward exp of `n` stacked matrices sitting in slot 0 (variable 0 = A). Returns the device pointer of U.
template <class T>
typename dq::Cx<T>::type* expm_forward(Ctx<T>* c, const ExpmPlan& p, int n, bool a_antiherm = false) {
  typedef typename dq::Cx<T>::type T2;
  if (p.nvars > SLOT_MAXV) fail(DQ_ERR_INVALID, "exp(A) plan needs too many temporaries");
  for (const TapeOp& o : p.ops) {
    T2* dst = c->mat(o.dst, n);
    const bool has_lin = o.nlin > 0 || o.cI != 0.0;
    if (o.kind == 0 || has_lin) {
      const T2* x[6];
      for (int j = 0; j < o.nlin; ++j) x[j] = c->mat(o.lsrc[j], n);
      c->lincomb(dst, o.nlin, x, o.lc, o.cI, n, false);
    }
    // a_antiherm (the chain: A = -i dt H with Hermitian H): even / odd powers of A are Hermitian / anti-Hermitian
    // (and the left operand's parity says whether it is Hermitian (opA hint 3) or anti-Hermitian (4) as stored)
    if (o.kind == 1)
      c->mm(a_antiherm && o.apar ? (o.apar > 0 ? 3 : 4) : 0, 0, c->mat(o.a, n), c->mat(o.b, n), dst, has_lin ? 1.0 : 0.0, n,
            a_antiherm ? o.herm : 0);
  }
  c->last_degree = p.m;
  c->last_squarings = p.s;
  c->last_products = p.products();
  return c->mat(p.out, n);
}

// Where the caller must put Ubar (cotangent of U) before expm_backward.
template <class T>
typename dq::Cx<T>::type* ubar_slot(Ctx<T>* c, const ExpmPlan& p, int n) {
  return c->mat(SLOT_ADJ + p.out, n);
}

// Reverse mode through expm_forward (whose variables are still in the pool).
// Pairing: dL = Re tr(Xbar^H dX). For C = X Y: Xbar += Cbar Y^H, Ybar += X^H Cbar; for a linear term
// C = ... + c X: Xbar += c Cbar. Linear contributions are queued per variable and applied in one streaming
// pass when that variable's adjoint is needed. Returns the device pointer of Abar (n stacked matrices).
// skew_only: the caller needs only the anti-Hermitian part of Abar and guarantees A^H = -A (the chain: the gradient
// pairs Abar with -i dt H_k). Then the two products of the reverse mode of v = A A,
//   Abar += vbar A^H + A^H vbar = -(vbar A + A vbar),  have skew part  X^H - X  with  X = herm(vbar) A :
// ONE ZGEMM and two transpose-combine streaming passes instead of two ZGEMMs.
template <class T>
typename dq::Cx<T>::type* expm_backward(Ctx<T>* c, const ExpmPlan& p, int n, bool skew_only = false) {
  typedef typename dq::Cx<T>::type T2;
  struct Pending { const T2* x; double coef; };
  std::vector<std::vector<Pending>> pend(p.nvars);
  std::vector<char> init(p.nvars, 0);
  init[p.out] = 1;
  auto adj = [&](int v) { return c->mat(SLOT_ADJ + v, n); };
  auto flush = [&](int v) {
    auto& q = pend[v];
    size_t i = 0;
    while (i < q.size()) {
      const T2* x[6];
      double cf[6];
      int k = 0;
      for (; k < 6 && i < q.size(); ++k, ++i) { x[k] = q[i].x; cf[k] = q[i].coef; }
      c->lincomb(adj(v), k, x, cf, 0.0, n, init[v] != 0);
      init[v] = 1;
    }
    q.clear();
  };
  for (int oi = (int)p.ops.size() - 1; oi >= 0; --oi) {
    const TapeOp& o = p.ops[oi];
    flush(o.dst);
    if (!init[o.dst]) continue;  // this variable does not reach the output
    const T2* db = adj(o.dst);
    if (o.kind == 1 && skew_only && o.a == 0 && o.b == 0) {
      T2* S = c->mat(SLOT_TREE + 2, n);
      T2* X = c->mat(SLOT_TREE + 3, n);
      const double by = 2.0 * sizeof(T) * c->nn() * n;
      c->timed(DQ_PROF_LINCOMB, 4.0 * c->nn() * n, 3.0 * by, [&] {
        return dq::tcombine_launch<T>(S, db, c->d, 0.5, 0.5, 0, (long)c->nn(), n, c->st);
      });
      c->mm(3, 0, S, c->mat(0, n), X, 0.0, n);  // S = herm(vbar): Hermitian as stored
      flush(0);  // pending linear contributions first, so that `accumulate` below is well defined
      c->timed(DQ_PROF_LINCOMB, 4.0 * c->nn() * n, (init[0] ? 4.0 : 3.0) * by, [&] {
        return dq::tcombine_launch<T>(adj(0), X, c->d, -1.0, 1.0, init[0] ? 1 : 0, (long)c->nn(), n, c->st);
      });
      init[0] = 1;
    } else if (o.kind == 1) {
      c->mm(0, 1, db, c->mat(o.b, n), adj(o.a), init[o.a] ? 1.0 : 0.0, n);
      init[o.a] = 1;
      c->mm(1, 0, c->mat(o.a, n), db, adj(o.b), init[o.b] ? 1.0 : 0.0, n);
      init[o.b] = 1;
    }
    for (int j = 0; j < o.nlin; ++j) pend[o.lsrc[j]].push_back({db, o.lc[j]});
  }
  flush(0);
  if (!init[0]) HIPCK(hipMemsetAsync(adj(0), 0, (size_t)n * c->mat_bytes(), c->st));
  return adj(0);
}

// ---- real-symmetric fast path -----------------------------------------------------------------
// H0, H_k real  =>  A_t = i X_t with X_t = -dt (H0 + sum_k u_k H_k) real symmetric, and
//   U_t = exp(A_t) = cos X_t + i sin X_t = C + i S,   C = sum_j cc_j Y^j,  S = X P,  P = sum_j sc_j Y^j,  Y = X^2
// (the Taylor polynomial of degree 2q+1). All products are real products of commuting symmetric matrices:
// q + 1 symmetric-output real GEMMs (d^3 flop each) instead of 3 complex ones (8 d^3 each).
// Larger steps: the polynomial is evaluated at X / 2^s and squared s times in the real algebra,
//   (C + iS)^2 = (C^2 - S^2) + i (2 C S):   C' = C C - S S,  S' = 2 C S     (3 symmetric-output products per squaring)
// -- the real form of the complex tape's U <- U U, with the same 2^s amplification of the rounding error.
// Pool slots (real unless noted): X 0, Y_j j (j = 1..q <= 6), P 7, C 8, S 9, U 10 (complex), squaring level i = 1..s:
// C_i 12 + 2(i-1), S_i 13 + 2(i-1); adjoints at SLOT_ADJ + slot, G = Ubar at SLOT_ADJ + 10 (complex), second
// cotangent pair of the squaring reverse at SLOT_ADJ + 12/13.
enum { RS_X = 0, RS_P = 7, RS_C = 8, RS_S = 9, RS_U = 10, RS_LVL = 12, RQ_MAX = 6, RS_SMAX = 14 };

// sparse-generator scheme (spx plans, s = 0): the power X_m (m = 2 .. 2q + 1 <= 13) lives in the squaring-level slots
inline int spx_slot(int m) { return m == 1 ? (int)RS_X : (int)RS_LVL + (m - 2); }

ExpmPlan make_real_plan(int q, int s = 0) {
  ExpmPlan p;
  p.rq = q;
  p.m = 2 * q + 1;
  p.s = s;
  p.out = RS_U;
  p.nvars = RS_U + 1;
  p.nprod = q + 1 + 3 * s;
  double f = 1.0;  // (2j)!
  for (int j = 0; j <= q; ++j) {
    if (j > 0) f *= (2.0 * j - 1.0) * (2.0 * j);
    const double sg = (j & 1) ? -1.0 : 1.0;
    p.cc[j] = sg / f;
    p.sc[j] = sg / (f * (2.0 * j + 1.0));
  }
  return p;
}

template <class T>
void rlincomb2(Ctx<T>* c, T* out0, T* out1, int n_in, const T* const* x, const double* c0, const double* c1, double cI0,
               double cI1, int batch, bool acc) {
  dq::RLincombArgs<T> a;
  memset(&a, 0, sizeof(a));
  a.out[0] = out0;
  a.out[1] = out1;
  a.n_out = out1 ? 2 : 1;
  a.n_in = n_in;
  for (int j = 0; j < n_in; ++j) {
    a.x[j] = x[j];
    a.c[0][j] = c0[j];
    a.c[1][j] = c1 ? c1[j] : 0.0;
  }
  a.cI[0] = cI0;
  a.cI[1] = cI1;
  a.d = c->d;
  a.stride = (long)c->nn();
  a.batch = batch;
  a.accumulate = acc ? 1 : 0;
  const double by = (double)sizeof(T) * c->nn() * (n_in + a.n_out * (acc ? 2 : 1)) * batch;
  c->timed(DQ_PROF_LINCOMB, 2.0 * c->nn() * n_in * a.n_out * batch, by, [&] { return dq::rlincomb_launch<T>(a, c->st); });
}

inline dq::XrPoly xr_poly(const ExpmPlan& p) {
  dq::XrPoly pol;
  memset(&pol, 0, sizeof(pol));
  pol.q = p.rq;
  for (int j = 0; j <= p.rq && j <= dq::XR_QMAX; ++j) { pol.cc[j] = p.cc[j]; pol.sc[j] = p.sc[j]; }
  return pol;
}
// assemble X and evaluate U = cos X + i sin X for `n` stacked steps; returns the complex propagators
template <class T>
typename dq::Cx<T>::type* real_forward(Ctx<T>* c, const ExpmPlan& p, int c0, int n, double dt) {
  typedef typename dq::Cx<T>::type T2;
  const int q = p.rq;
  T* X = c->rmat(RS_X, n);
  const double by = (double)sizeof(T) * c->nn() * (c->K + 2.0) * n;
  const double dts = std::ldexp(dt, -p.s);  // X = -(dt / 2^s) H
  c->last_xr = 0;
  if (p.spx && p.xr) {
    // XOR-regular sparse generator: U column by column in LDS (xorexp.hip) -- no dense X, no stored powers, no combination pass
    T2* U = c->mat(RS_U, n);
    const dq::XrPoly pol = xr_poly(p);
    const double rows = (double)c->xr_tab.R + 1.0;
    c->timed(DQ_PROF_SPMM, 2.0 * rows * c->nn() * 2.0 * q * n, (double)sizeof(T2) * c->nn() * n, [&] {
      return dq::xr_forward_launch(U, (long)c->nn(), c->d, c->xr_tab, c->u_dev.d() + (size_t)c0 * c->K, c->K, dts, pol, n, c->st);
    });
    c->last_degree = p.m;
    c->last_squarings = 0;
    c->last_products = 2 * q;
    c->last_real_q = q;
    c->last_spx = 1;
    c->last_xr = 1;
    return U;
  }
  if (c->sparse_on()) {
    // dense pass over H0 only (or zeros, if the drift's nonzeros are in the CSR too), then the nonzeros are scattered in
    if (c->sparse_x) {
      c->timed(DQ_PROF_ASSEMBLE, 0.0, (double)sizeof(T) * c->nn() * n,
               [&] { return hipMemsetAsync(X, 0, sizeof(T) * c->nn() * (size_t)n, c->st); });
    } else {
      c->timed(DQ_PROF_ASSEMBLE, 2.0 * c->nn() * n, (double)sizeof(T) * c->nn() * 2.0 * n, [&] {
        return dq::rassemble_launch<T>(X, (const T*)c->H0r.p, (const T*)c->Hsr.p, c->u_dev.d(), 0, (long)c->nn(), n, dts, c->st);
      });
    }
    c->timed(DQ_PROF_ASSEMBLE, 2.0 * c->sp_nnz * n, 3.0 * sizeof(double) * c->sp_nnz * n, [&] {
      return dq::rsp_add_launch<T>(X, (long)c->nn(), (const long*)c->sp_rowptr.p, (const int*)c->sp_col.p,
                                   (const int*)c->sp_kid.p, (const double*)c->sp_val.p,
                                   c->u_dev.d() + (size_t)c0 * c->K, c->K, c->d, n, dts, c->st);
    });
  } else {
    c->timed(DQ_PROF_ASSEMBLE, 2.0 * c->nn() * (c->K + 1.0) * n, by, [&] {
      return dq::rassemble_launch<T>(X, (const T*)c->H0r.p, (const T*)c->Hsr.p, c->u_dev.d() + (size_t)c0 * c->K, c->K,
                                     (long)c->nn(), n, dts, c->st);
    });
  }
  if (p.spx) {
    // sparse generator: X_{m+1} = X X_m (m = 1 .. 2q) by sparse x dense products; C and S are linear combinations
    const int np = 2 * q + 1;
    const T* Xp[2 * RQ_MAX + 2];
    Xp[1] = X;
    const double bym = (double)sizeof(T) * c->nn() * 2.0 * n;  // algorithmic: V read once, the product written once
    for (int m = 1; m < np; ++m) {
      T* Xn = c->rmat(spx_slot(m + 1), n);
      c->timed(DQ_PROF_SPMM, 2.0 * c->sp_nnz * c->d * n, bym, [&] {
        return dq::rsp_mm_launch<T>(Xn, Xp[m], (const T*)nullptr, 0.0, (long)c->nn(), c->d, (const long*)c->sp_rowptr.p,
                                    (const int*)c->sp_col.p, (const int*)c->sp_kid.p, (const double*)c->sp_val.p,
                                    c->u_dev.d() + (size_t)c0 * c->K, c->K, n, dts, c->slab_spmm, c->st);
      });
      Xp[m + 1] = Xn;
    }
    // U = cos X + i sin X = (cc_0 I + sum_j cc_j X_2j) + i (sum_j sc_j X_2j+1): ONE pass over the stored powers X_1 .. X_2q+1
    // that writes the interleaved complex U directly (C and S are never stored: nothing else reads them on this scheme)
    T2* U = c->mat(RS_U, n);
    if (np > 8) {  // more powers than one linear-combination launch takes (q >= 4): C and S separately, then U = C + i S
      T* Cm = c->rmat(RS_C, n);
      T* S = c->rmat(RS_S, n);
      const T* ev[RQ_MAX + 1];
      const T* od[RQ_MAX + 1];
      for (int j = 1; j <= q; ++j) ev[j - 1] = Xp[2 * j];
      for (int j = 0; j <= q; ++j) od[j] = Xp[2 * j + 1];
      rlincomb2<T>(c, Cm, nullptr, q, ev, p.cc + 1, nullptr, p.cc[0], 0.0, n, false);
      rlincomb2<T>(c, S, nullptr, q + 1, od, p.sc, nullptr, 0.0, 0.0, n, false);
      const long ne = (long)c->nn() * n;
      c->timed(DQ_PROF_LINCOMB, 0.0, 4.0 * sizeof(T) * ne, [&] { return dq::pack_u_launch<T>(U, Cm, S, ne, c->st); });
    } else {
      dq::RLincombArgs<T> a;
      memset(&a, 0, sizeof(a));
      a.out[0] = (T*)U;
      a.n_out = 2;
      a.pack = 1;
      a.n_in = np;
      for (int m = 1; m <= np; ++m) {
        a.x[m - 1] = Xp[m];
        a.c[0][m - 1] = (m & 1) ? 0.0 : p.cc[m / 2];
        a.c[1][m - 1] = (m & 1) ? p.sc[(m - 1) / 2] : 0.0;
      }
      a.cI[0] = p.cc[0];
      a.d = c->d;
      a.stride = (long)c->nn();
      a.batch = n;
      const double by = (double)sizeof(T) * c->nn() * (np + 2.0) * n;
      c->timed(DQ_PROF_LINCOMB, 2.0 * c->nn() * np * n, by, [&] { return dq::rlincomb_launch<T>(a, c->st); });
    }
    c->last_degree = p.m;
    c->last_squarings = 0;
    c->last_products = 2 * q;
    c->last_real_q = q;
    c->last_spx = 1;
    return U;
  }
  c->last_spx = 0;
  const T* Y[RQ_MAX + 1];
  for (int j = 1; j <= q; ++j) {
    T* Yj = c->rmat(j, n);
    c->rmm(j == 1 ? X : Y[1], j == 1 ? X : Y[j - 1], Yj, n, true);
    Y[j] = Yj;
  }
  T* P = c->rmat(RS_P, n);
  T* Cm = c->rmat(RS_C, n);
  rlincomb2(c, Cm, P, q, Y + 1, p.cc + 1, p.sc + 1, p.cc[0], p.sc[0], n, false);
  T* S = c->rmat(RS_S, n);
  c->rmm(X, P, S, n, true);
  for (int i = 1; i <= p.s; ++i) {  // (C + iS) <- (C + iS)^2
    T* Cn = c->rmat(RS_LVL + 2 * (i - 1), n);
    T* Sn = c->rmat(RS_LVL + 2 * (i - 1) + 1, n);
    c->rmm(Cm, Cm, Cn, n, true, false, 1.0, false);
    c->rmm(S, S, Cn, n, true, false, -1.0, true);
    c->rmm(Cm, S, Sn, n, true, false, 2.0, false);
    Cm = Cn;
    S = Sn;
  }
  T2* U = c->mat(RS_U, n);
  const long ne = (long)c->nn() * n;
  c->timed(DQ_PROF_LINCOMB, 0.0, 4.0 * sizeof(T) * ne, [&] { return dq::pack_u_launch<T>(U, Cm, S, ne, c->st); });
  c->last_degree = p.m;
  c->last_squarings = p.s;
  c->last_products = p.nprod;
  c->last_real_q = q;
  return U;
}

// Reverse mode through real_forward (whose variables are still in the pool); G = Ubar sits in ubar_slot.
// Every variable is a symmetric function of X, so only the symmetric part of each cotangent matters:
//   W = V1 V2 (V1 != V2):  V1bar += sym(Wbar V2), V2bar += sym(Wbar V1);     W = V V:  Vbar += 2 sym(Wbar V)
// with sym(Z) = (Z + Z^T)/2 = (Wbar V + V Wbar)/2 for symmetric Wbar, V: one two-product symmetric-output launch per
// cotangent (the MFMA work of one full real GEMM), 2q in all.
// Returns Xbar (n stacked real symmetric matrices): dL/du_k = <Xbar, -dt H_k>.
// skip_x: the two products into Xbar are left out (sparse controls: chunk_vjp contracts Sbar, P, Ybar_1, X directly);
// sb_out / yb_out then receive the cotangents of S and Y_1, and the return value is null.
template <class T>
const T* real_backward(Ctx<T>* c, const ExpmPlan& p, int n, bool skip_x = false, const T** sb_out = nullptr,
                       const T** yb_out = nullptr) {
  typedef typename dq::Cx<T>::type T2;
  const int q = p.rq;
  const T2* G = c->mat(SLOT_ADJ + RS_U, n);
  const T* X = c->rmat(RS_X, n);
  const T* P = c->rmat(RS_P, n);
  const T* Y[RQ_MAX + 1];
  T* Yb[RQ_MAX + 1];
  for (int j = 1; j <= q; ++j) { Y[j] = c->rmat(j, n); Yb[j] = c->rmat(SLOT_ADJ + j, n); }
  T* Xb = c->rmat(SLOT_ADJ + RS_X, n);
  T* Pb = c->rmat(SLOT_ADJ + RS_P, n);
  // sym2(W, V, out, a): out (+)= a (W V + V W)  -- one symmetric-output launch with both products (no full-matrix
  // round trip, no symmetrisation pass); a = 1/2 gives sym(W V), a = 1 gives 2 sym(W V)
  auto sym2 = [&](const T* W, const T* V, T* out, double a, bool acc) { c->rmm(W, V, out, n, true, true, a, acc); };
  // cotangents of (C_s, S_s) from G, then back through the squarings; two slot pairs alternate so that the cotangents of
  // the polynomial's (C_0, S_0) end up in SLOT_ADJ + RS_C / RS_S
  T* pairA[2] = {c->rmat(SLOT_ADJ + RS_C, n), c->rmat(SLOT_ADJ + RS_S, n)};
  T* pairB[2] = {c->rmat(SLOT_ADJ + RS_LVL, n), c->rmat(SLOT_ADJ + RS_LVL + 1, n)};
  T** cur = (p.s & 1) ? pairB : pairA;
  c->timed(DQ_PROF_LINCOMB, 2.0 * c->nn() * n, 6.0 * sizeof(T) * c->nn() * n,
           [&] { return dq::split_g_launch<T>(G, cur[0], cur[1], c->d, (long)c->nn(), n, c->st); });
  for (int i = p.s; i >= 1; --i) {
    // C_i = C C - S S,  S_i = 2 C S  with (C, S) = level i-1:
    //   Cbar = 2 sym(Cbar_i C) + 2 sym(Sbar_i S),   Sbar = -2 sym(Cbar_i S) + 2 sym(Sbar_i C)
    const T* Cp = i == 1 ? c->rmat(RS_C, n) : c->rmat(RS_LVL + 2 * (i - 2), n);
    const T* Sp = i == 1 ? c->rmat(RS_S, n) : c->rmat(RS_LVL + 2 * (i - 2) + 1, n);
    T** nxt = cur == pairA ? pairB : pairA;
    sym2(cur[0], Cp, nxt[0], 1.0, false);
    sym2(cur[1], Sp, nxt[0], 1.0, true);
    sym2(cur[0], Sp, nxt[1], -1.0, false);
    sym2(cur[1], Cp, nxt[1], 1.0, true);
    cur = nxt;
  }
  T* Cb = cur[0];
  T* Sb = cur[1];
  // S = X P
  if (!skip_x) sym2(Sb, P, Xb, 0.5, false);
  sym2(Sb, X, Pb, 0.5, false);
  // C = cc0 I + sum cc_j Y_j,  P = sc0 I + sum sc_j Y_j
  for (int j = 1; j <= q; ++j) {
    const T* in[2] = {Cb, Pb};
    const double cf[2] = {p.cc[j], p.sc[j]};
    rlincomb2<T>(c, Yb[j], nullptr, 2, in, cf, nullptr, 0.0, 0.0, n, false);
  }
  // Y_j = Y_1 Y_{j-1}
  for (int j = q; j >= 3; --j) {
    sym2(Yb[j], Y[j - 1], Yb[1], 0.5, true);
    sym2(Yb[j], Y[1], Yb[j - 1], 0.5, true);
  }
  if (q >= 2) sym2(Yb[2], Y[1], Yb[1], 1.0, true);  // Y_2 = Y_1 Y_1
  if (skip_x) {
    if (sb_out) *sb_out = Sb;
    if (yb_out) *yb_out = Yb[1];
    return nullptr;
  }
  sym2(Yb[1], X, Xb, 1.0, true);                     // Y_1 = X X
  return Xb;
}

// Reverse mode of the sparse-generator scheme (spx plans), down to the control gradients of the chunk:
//   Xbar_{2q+1} = sc_q Sbar;   Xbar_m = coef_m (Cbar | Sbar) + X Xbar_{m+1}   (m = 2q .. 2: sparse x dense with an AXPY)
//   dL/du_k = scale * < Xbar_1 + sum_{m=1}^{2q} Xbar_{m+1} X_m , H_k >  over nnz(H_k)     (Xbar_1 only at those entries)
template <class T>
void real_backward_spx(Ctx<T>* c, const ExpmPlan& p, int c0, int n, double dt, double* grad_out) {
  typedef typename dq::Cx<T>::type T2;
  const int q = p.rq, np = 2 * q + 1, K = c->K;
  const T2* G = c->mat(SLOT_ADJ + RS_U, n);
  if (p.xr) {
    // column form: rows of G in, control gradients out (the powers of the column are rebuilt in registers)
    const dq::XrPoly pol = xr_poly(p);
    c->xr_part.ensure(sizeof(double) * (size_t)n * dq::xr_blocks(c->d, n) * dq::XR_SLOTS);
    c->xr_pow.ensure(sizeof(T) * (size_t)n * dq::xr_pow_elems(c->d, q, n));
    const double rows = (double)c->xr_tab.R + 1.0;
    c->timed(DQ_PROF_SPMM, 2.0 * rows * c->nn() * (4.0 * q + 1.0) * n, (double)sizeof(T2) * c->nn() * n, [&] {
      return dq::xr_reverse_launch(G, (long)c->nn(), c->d, c->xr_tab, c->u_dev.d() + (size_t)c0 * K, K, dt, pol, n,
                                   (double*)c->xr_part.p, (const unsigned*)c->xr_sgn.p, (T*)c->xr_pow.p, grad_out, 0, c->st);
    });
    return;
  }
  T* Cb = c->rmat(SLOT_ADJ + RS_C, n);
  T* Sb = c->rmat(SLOT_ADJ + RS_S, n);
  c->timed(DQ_PROF_LINCOMB, 2.0 * c->nn() * n, 6.0 * sizeof(T) * c->nn() * n,
           [&] { return dq::split_g_launch<T>(G, Cb, Sb, c->d, (long)c->nn(), n, c->st); });
  const T* Xp[2 * RQ_MAX + 2];
  const T* Xb[2 * RQ_MAX + 2];
  for (int m = 1; m <= np; ++m) Xp[m] = c->rmat(spx_slot(m), n);
  auto coef = [&](int m) { return (m & 1) ? p.sc[(m - 1) / 2] : p.cc[m / 2]; };
  {  // Xbar_{2q+1} = sc_q Sbar
    T* top = c->rmat(SLOT_ADJ + spx_slot(np), n);
    const T* in[1] = {Sb};
    const double cf[1] = {coef(np)};
    rlincomb2<T>(c, top, nullptr, 1, in, cf, nullptr, 0.0, 0.0, n, false);
    Xb[np] = top;
  }
  const double dts = dt;  // spx plans have no squarings
  // Each reverse pass Xbar_m = coef_m (Cbar|Sbar) + X Xbar_{m+1} streams the partner rows of Xbar_{m+1}; the gradient's level
  // <Xbar_{m+1}[j,:], X_m[i,:]> over X's pattern rides along (rsp_mm_dot_kernel). One more dot-only pass pairs Xbar_2 with
  // X_1, and rsp_finish_kernel adds Xbar_1 = sc_0 Sbar + X Xbar_2 at the controls' nonzeros. (fuse_levels = 0: the
  // separate level-contraction kernel, kept for comparison.)
  const size_t nz = (size_t)std::max<long>(c->sp_nnz, 1);
  c->sp_nzval.ensure(sizeof(double) * nz * n);
  const double bym = (double)sizeof(T) * c->nn() * 4.0 * n;  // algorithmic: Xbar_{m+1}, Cbar|Sbar, X_m read, Xbar_m written
  if (c->fuse_levels) {
    c->sp_nzacc.ensure(sizeof(double) * nz * n);
    double* nzacc = (double*)c->sp_nzacc.p;
    c->sp_part.ensure(sizeof(double) * nz * n * 8);  // per-slab dot products of the XCD-slab form
    double* part = c->slab_spmm ? (double*)c->sp_part.p : nullptr;
    bool started = false;
    for (int m = np - 1; m >= 1; --m) {
      T* dst = m >= 2 ? c->rmat(SLOT_ADJ + spx_slot(m), n) : nullptr;  // level 1: dot products only
      const T* add = (m & 1) ? Sb : Cb;
      const double ca = coef(m);
      const T* src = Xb[m + 1];
      const T* xm = Xp[m];
      const int accf = started ? 1 : 0;
      c->timed(DQ_PROF_SPMM, 4.0 * c->sp_nnz * c->d * n, m >= 2 ? bym : 0.5 * bym, [&] {
        return dq::rsp_mm_dot_launch<T>(dst, src, add, ca, xm, (long)c->nn(), c->d, (const long*)c->sp_rowptr.p,
                                        (const int*)c->sp_col.p, (const int*)c->sp_kid.p, (const double*)c->sp_val.p,
                                        c->u_dev.d() + (size_t)c0 * K, K, n, dts, c->sp_nnz, nzacc, accf, part, c->st);
      });
      started = true;
      if (m >= 2) Xb[m] = dst;
    }
    c->timed(DQ_PROF_GRADC, 2.0 * c->sp_nnz * n, 16.0 * c->sp_nnz * n, [&] {
      return dq::rsp_finish_launch<T>(nzacc, Sb, Xb[2], coef(1), (long)c->nn(), c->d, (const long*)c->sp_rowptr.p,
                                      (const int*)c->sp_col.p, (const int*)c->sp_kid.p, (const double*)c->sp_val.p,
                                      c->u_dev.d() + (size_t)c0 * K, K, n, dts, c->sp_nnz, (double*)c->sp_nzval.p, c->st);
    });
  } else {
    for (int m = np - 1; m >= 2; --m) {
      T* dst = c->rmat(SLOT_ADJ + spx_slot(m), n);
      const T* add = (m & 1) ? Sb : Cb;
      const double ca = coef(m);
      const T* src = Xb[m + 1];
      c->timed(DQ_PROF_SPMM, 2.0 * c->sp_nnz * c->d * n, 0.75 * bym, [&] {
        return dq::rsp_mm_launch<T>(dst, src, add, ca, (long)c->nn(), c->d, (const long*)c->sp_rowptr.p,
                                    (const int*)c->sp_col.p, (const int*)c->sp_kid.p, (const double*)c->sp_val.p,
                                    c->u_dev.d() + (size_t)c0 * K, K, n, dts, c->slab_spmm, c->st);
      });
      Xb[m] = dst;
    }
    // levels m = 1 .. 2q pair Xbar_{m+1} with X_m; the direct term is Xbar_1 = sc_0 Sbar + X Xbar_2 at the controls' nonzeros
    const T* Gl[2 * RQ_MAX + 2];
    const T* Xl[2 * RQ_MAX + 2];
    for (int m = 1; m < np; ++m) { Gl[m - 1] = Xb[m + 1]; Xl[m - 1] = Xp[m]; }
    const double byl = (double)sizeof(T) * c->d * (np - 1) * ((double)c->d + 0.5 * c->sp_nnz) * n;
    c->timed(DQ_PROF_GRADC, 2.0 * (np - 1) * 0.5 * c->sp_nnz * c->d * n, byl, [&] {
      return dq::rsp_levels_launch<T>(Gl, Xl, np - 1, Sb, Xb[2], coef(1), (long)c->nn(), c->d, (const long*)c->sp_rowptr.p,
                                      (const int*)c->sp_col.p, (const int*)c->sp_kid.p, (const double*)c->sp_val.p,
                                      c->u_dev.d() + (size_t)c0 * K, K, n, dts, c->sp_nnz, (double*)c->sp_nzval.p, c->st);
    });
  }
  c->timed(DQ_PROF_GRADC, (double)c->sp_nnz * n, 8.0 * c->sp_nnz * n, [&] {
    return dq::rsp_reduce_launch((const double*)c->sp_nzval.p, c->sp_nnz, (const long*)c->sp_kptr.p, (const long*)c->sp_kent.p,
                                 K, n, -dt, grad_out, c->st);
  });
}

#define TT typename dq::Cx<T>::type

template <class T>
void stage_h2d(Ctx<T>* c, void* dst, const void* src, size_t bytes) {
  HIPCK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, c->st));
}

// returns a device pointer for an input that may live on the host
template <class T>
const TT* in_dev(Ctx<T>* c, const void* p, size_t bytes, DevBuf& tmp) {
  if (is_device_ptr(p)) return (const TT*)p;
  tmp.ensure(bytes);
  stage_h2d(c, tmp.p, p, bytes);
  return (const TT*)tmp.p;
}

template <class T>
void out_copy(Ctx<T>* c, void* dst, const void* src_dev, size_t bytes) {
  if (!dst || dst == src_dev) return;
  HIPCK(hipMemcpyAsync(dst, src_dev, bytes, hipMemcpyDefault, c->st));
}

template <class T>
double norm1_dev(Ctx<T>* c, const TT* X) {
  c->scratch.ensure(sizeof(double) * ((size_t)c->d + (size_t)dq::colabs_scratch_doubles(c->d)));
  HIPCK(dq::colabs_launch<T>(X, c->d, c->scratch.d(), c->scratch.d() + c->d, c->st));
  std::vector<double> h(c->d);
  HIPCK(hipMemcpyAsync(h.data(), c->scratch.p, sizeof(double) * h.size(), hipMemcpyDeviceToHost, c->st));
  HIPCK(hipStreamSynchronize(c->st));
  double mx = 0.0;
  for (double v : h) {
    if (!std::isfinite(v)) fail(DQ_ERR_INVALID, "matrix contains non-finite entries");
    mx = std::max(mx, v);
  }
  return mx;
}

template <class T>
double herm_defect_dev(Ctx<T>* c, const TT* X) {
  c->scratch.ensure(sizeof(double) * (size_t)std::max(c->d, 1));
  HIPCK(dq::herm_defect_launch<T>(X, c->d, c->scratch.d(), c->st));
  double h = 0.0;
  HIPCK(hipMemcpyAsync(&h, c->scratch.p, sizeof(double), hipMemcpyDeviceToHost, c->st));
  HIPCK(hipStreamSynchronize(c->st));
  return h;
}

template <class T>
void check_ctx(Ctx<T>* c) {
  if (!c) fail(DQ_ERR_INVALID, "null context");
  HIPCK(hipSetDevice(c->dev));
}

// number of steps whose exp() is built in one batched pass
template <class T>
int time_chunk(Ctx<T>* c, int n_steps) {
  const double per_step = 24.0 * (double)c->mat_bytes();  // tape variables + their adjoints (nominal)
  double budget = c->chunk_bytes;
  if (budget <= 0.0) {
    // default: 8 GiB; at dimensions where one d x d product leaves most of the chip idle (d <= 1024) batching over time
    // is what fills it, so a tenth of the free HBM (plus what the pool already holds) if that is more
    budget = 8.0 * 1024 * 1024 * 1024;
    if ((long)((c->d + 127) / 128) * ((c->d + 127) / 128) < c->n_cu) {
      size_t fr = 0, tot = 0;
      HIPCK(hipMemGetInfo(&fr, &tot));
      double held = 0.0;
      for (const auto& b : c->pool) held += (double)b.bytes;
      budget = std::max(budget, 0.10 * ((double)fr + held));
    }
  }
  long tc = (long)(budget / per_step);
  tc = std::max(1L, std::min<long>(tc, n_steps));
  tc = std::min<long>(tc, 16384);
  return (int)tc;
}

template <class T>
double theta_bound(Ctx<T>* c, const double* u_row, double dt) {
  double h = c->nH0;
  for (int k = 0; k < c->K; ++k) h += std::fabs(u_row[k]) * c->nHk[k];
  return std::fabs(dt) * h;
}

template <class T>
std::vector<double> fetch_u(Ctx<T>* c, const double* u, int S) {
  const size_t n = (size_t)S * c->K;
  std::vector<double> h(n);
  if (n == 0) return h;
  c->u_dev.ensure(std::max<size_t>(n, 1) * sizeof(double));
  if (is_device_ptr(u)) {
    // a device table: the engine's copy is made on the device, in stream order (no second trip through the host and no second
    // synchronisation: each costs a short call ~45 us of idle GPU -- configs[1]'s 0.9 ms forward call has three of them)
    HIPCK(hipMemcpyAsync(c->u_dev.p, u, n * sizeof(double), hipMemcpyDeviceToDevice, c->st));
    HIPCK(hipMemcpyAsync(h.data(), u, n * sizeof(double), hipMemcpyDeviceToHost, c->st));
    HIPCK(hipStreamSynchronize(c->st));
  } else {
    memcpy(h.data(), u, n * sizeof(double));
  }
  for (double v : h)
    if (!std::isfinite(v)) fail(DQ_ERR_INVALID, "pulse table contains non-finite values");
  if (!is_device_ptr(u)) {
    HIPCK(hipMemcpyAsync(c->u_dev.p, h.data(), n * sizeof(double), hipMemcpyHostToDevice, c->st));
    HIPCK(hipStreamSynchronize(c->st));  // h may be pageable
  }
  c->uh_last = h;
  return h;
}

template <class T>
ExpmPlan chunk_plan(Ctx<T>* c, const std::vector<double>& uh, int c0, int n, double dt) {
  double th = 0.0;
  for (int t = c0; t < c0 + n; ++t) th = std::max(th, theta_bound(c, uh.data() + (size_t)t * c->K, dt));
  c->last_theta = th;
  c->last_real_q = 0;
  if (c->real_h && c->real_mode != 0) {
    if (c->real_mode > 0) {
      ExpmPlan p = make_real_plan(std::min(c->real_mode, (int)RQ_MAX));
      p.spx = c->sparse_gen_on() ? 1 : 0;
      p.xr = p.spx && c->xr_on(p.rq) ? 1 : 0;
      return p;
    }
    if (c->sparse_gen_on() && c->force_degree <= 0 && c->force_s < 0 && c->d >= c->real_min_dim) {
      // sparse generator: every product is a sparse x dense pass, the polynomial alone (no squarings) whenever one fits
      for (int q = 1; q <= RQ_MAX; ++q)
        if (taylor_remainder(th, 2 * q + 1) <= c->expm_tol) {
          ExpmPlan p = make_real_plan(q, 0);
          p.spx = 1;
          p.xr = c->xr_on(q) ? 1 : 0;
          p.nprod = 2 * q;
          return p;
        }
    }
    if (c->force_degree <= 0 && c->force_s < 0 && c->d >= c->real_min_dim) {  // (those options address the complex tape)
      // cheapest (q, s): q + 1 products for the polynomial at theta / 2^s plus 3 per squaring
      int bq = 0, bs = 0, best = 1 << 30;
      for (int sq = 0; sq <= RS_SMAX; ++sq) {
        const double ths = std::ldexp(th, -sq);
        for (int q = 1; q <= RQ_MAX; ++q)
          if (taylor_remainder(ths, 2 * q + 1) <= c->expm_tol) {
            if (q + 1 + 3 * sq < best) { best = q + 1 + 3 * sq; bq = q; bs = sq; }
            break;
          }
        if (best <= 3 * (sq + 1)) break;  // more squarings cannot win any more
      }
      if (bq > 0) return make_real_plan(bq, bs);
    }
  }
  if (!c->real_h && c->zxr_on()) {
    for (int L = 1; L <= 16; ++L)
      if (taylor_remainder(th, L) <= c->expm_tol && dq::zxr_supported(c->d, L, c->zxr_tab.R, c->zxr_tab.nd)) {
        ExpmPlan p;
        p.zxr = 1;
        p.m = L;
        p.out = RS_U;
        p.nvars = RS_U + 1;
        return p;
      }
  }
  return choose_plan(th, c->expm_tol, c->force_degree, c->force_s);
}

// (the complex column form exists for complex128 engines; the table is never set for others)
inline hipError_t zxr_forward_any(double2* U, long stride, int d, const dq::XrTab& tab, const double* u, int K, double dt, int L, int n,
                                  hipStream_t st) {
  return dq::zxr_forward_launch(U, stride, d, tab, u, K, dt, L, n, st);
}
inline hipError_t zxr_forward_any(float2*, long, int, const dq::XrTab&, const double*, int, double, int, int, hipStream_t) {
  return hipErrorInvalidValue;
}
inline hipError_t zxr_reverse_any(const double2* G, long stride, int d, const dq::XrTab& tab, const double* u, int K, double dt, int L,
                                  int n, double* part, void* xpow, double* grad, hipStream_t st) {
  return dq::zxr_reverse_launch(G, stride, d, tab, u, K, dt, L, n, part, nullptr, (double2*)xpow, grad, 0, st);
}
inline hipError_t zxr_reverse_any(const float2*, long, int, const dq::XrTab&, const double*, int, double, int, int, double*, void*,
                                  double*, hipStream_t) {
  return hipErrorInvalidValue;
}

// assemble + exp for steps [c0, c0+n): returns pointer to n stacked propagators
template <class T>
TT* build_propagators(Ctx<T>* c, const ExpmPlan& p, int c0, int n, double dt) {
  typedef TT T2;
  for (auto& r : c->redirect) r = nullptr;
  if (p.rq > 0) {
    // Cached chunks keep X, Y_1..Y_{q-1}, P (real) and U (complex) in their cache entry: the forward pass computes them
    // there directly (the pool slots are redirected for the duration of the chunk), the backward pass reads them there --
    // no copies, no rebuild.
    // (sparse-generator plans keep the powers X_1 .. X_2q that the gradient contraction pairs with, and U)
    const bool cacheable = p.s == 0 && c->pc_mode != 0 && c0 + n <= c->pc_limit;
    const size_t idx = (size_t)(c0 / std::max(c->pc_tc, 1));
    const size_t mat = c->nn() * sizeof(T) * (size_t)n;  // one real variable of the chunk
    const int q = p.rq;
    const int nreal = p.xr ? 0 : (p.spx ? 2 * q : q + 1);  // (column form: nothing but U is kept)
    auto real_slot = [&](int i) {
      if (p.spx) return spx_slot(i + 1);                      // X_1 .. X_2q
      return i == 0 ? (int)RS_X : (i < q ? i : (int)RS_P);    // X, Y_1 .. Y_{q-1}, P
    };
    auto point_at = [&](char* base) {
      for (int i = 0; i < nreal; ++i) c->redirect[real_slot(i)] = base + (size_t)i * mat;
      c->redirect[RS_U] = base + (size_t)nreal * mat;
    };
    const int tag = p.xr ? -(100 + q) : (p.spx ? -q : q);  // which scheme filled the entry
    if (cacheable && c->pc_mode == 2 && idx < c->pcache.size()) {
      auto& e = c->pcache[idx];
      if (e.valid && e.c0 == c0 && e.n == n && e.rq == tag) {
        point_at((char*)c->pc_arena.p + e.off);
        c->last_degree = p.m; c->last_squarings = 0; c->last_products = p.nprod; c->last_real_q = q;
        c->last_spx = p.spx;
        c->last_xr = p.xr;
        return c->mat(RS_U, n);
      }
    }
    if (cacheable && c->pc_mode == 1 && c->pc_used + (size_t)(nreal + 2) * mat <= c->pc_arena.bytes) {
      if (c->pcache.size() <= idx) c->pcache.resize(idx + 1);
      auto& e = c->pcache[idx];
      e.off = c->pc_used;
      c->pc_used += (size_t)(nreal + 2) * mat;
      point_at((char*)c->pc_arena.p + e.off);
      e.c0 = c0; e.n = n; e.rq = tag; e.valid = true;
      c->last_cached += n;
    }
    return real_forward(c, p, c0, n, dt);
  }
  c->last_xr = 0;
  if (p.zxr) {
    // complex Hermitian Pauli-string generator: the rows of U = exp(-i dt H) built in LDS (no assembly, no tape, no product)
    T2* U = c->mat(RS_U, n);
    const double rows = (double)c->zxr_tab.R + 1.0;
    c->timed(DQ_PROF_SPMM, 8.0 * rows * c->nn() * (p.m - 1.0) * n, (double)sizeof(T2) * c->nn() * n, [&] {
      return zxr_forward_any(U, (long)c->nn(), c->d, c->zxr_tab, c->u_dev.d() + (size_t)c0 * c->K, c->K, dt, p.m, n, c->st);
    });
    c->last_degree = p.m;
    c->last_squarings = 0;
    c->last_products = 0;
    c->last_real_q = 0;
    c->last_spx = 0;
    c->last_xr = 2;
    return U;
  }
  T2* A = c->mat(0, n);
  const double by = 2.0 * sizeof(T) * c->nn() * (c->K + 2.0) * n;
  c->timed(DQ_PROF_ASSEMBLE, 2.0 * 2 * c->nn() * (c->K + 1.0) * n, by, [&] {
    return dq::assemble_launch<T>(A, (const T2*)c->H0.p, (const T2*)c->Hs.p, c->u_dev.d() + (size_t)c0 * c->K, c->K,
                                  (long)c->nn(), n, dt, c->st);
  });
  return expm_forward(c, p, n, c->hermitian);
}

template <class T>
void require_H(Ctx<T>* c) {
  if (!c->have_H) fail(DQ_ERR_STATE, "dq_set_hamiltonians has not been called");
}

// Operator form of a chain: the product U_{S-1} ... U_0 of the propagators of steps [0, S) of the staged pulse table, by a
// log-depth tree of batched products per time chunk (8 d^3 flop per step, O(log n) launches instead of n) and a running
// product across chunks. Returns the d x d product (a pool slot), or null for S = 0.
template <class T>
TT* xs_compose_segments(Ctx<T>* c, const std::vector<double>& uh, int S, double dt, int* n_out);

// X = m stacked d x d matrices (time order) -> their product X[m-1] ... X[0] by a tree of batched products; returns where it is
template <class T>
TT* tree_product(Ctx<T>* c, TT* X, int m, bool first_level_symmetric) {
  typedef TT T2;
  const int d = c->d;
  int lvl = 0;
  while (m > 1) {
    const int pairs = m / 2;
    T2* Y = c->mat(SLOT_TREE + (lvl & 1), (size_t)(pairs + (m & 1)));
    const long s2 = 2 * (long)c->nn();
    // (level 0 multiplies the propagators themselves: complex symmetric on the real path, U = cos X + i sin X)
    c->zgemm(lvl == 0 && first_level_symmetric ? 2 : 0, 0, d, d, d, 1.0, X + c->nn(), d, s2, X, d, s2, 0.0, Y, d, (long)c->nn(), pairs);
    if (m & 1)
      HIPCK(hipMemcpyAsync(Y + (size_t)pairs * c->nn(), X + (size_t)(m - 1) * c->nn(), c->mat_bytes(),
                           hipMemcpyDeviceToDevice, c->st));
    X = Y;
    m = pairs + (m & 1);
    ++lvl;
  }
  return X;
}

template <class T>
TT* compose_chain(Ctx<T>* c, const std::vector<double>& uh, int S, double dt, int tc, bool allow_state_columns = false) {
  typedef TT T2;
  const int d = c->d;
  T2* running = nullptr;
  c->last_xs_segments_fwd = 0;
  if (allow_state_columns) {   // (not where the explicit propagators are read again: the operator-form sweep of dq_energy_grad)
    // Pauli-sum generators: no step exponential at all -- the product of the chain's propagators is what the state-column forward
    // kernel makes of the identity's columns (d columns x S steps: the cost of a forward pass over d states), cut into time segments
    // when d columns alone do not fill the chip; the segments' propagators are then multiplied as the steps' would be
    int nseg = 0;
    if (T2* X = xs_compose_segments(c, uh, S, dt, &nseg)) {
      T2* P = tree_product(c, X, nseg, false);
      running = c->mat(SLOT_TREE + 2, 1);
      HIPCK(hipMemcpyAsync(running, P, c->mat_bytes(), hipMemcpyDeviceToDevice, c->st));
      return running;
    }
  }
  for (int c0 = 0; c0 < S; c0 += tc) {
    const int n = std::min(tc, S - c0);
    const ExpmPlan p = chunk_plan(c, uh, c0, n, dt);
    T2* X = build_propagators(c, p, c0, n, dt);
    X = tree_product(c, X, n, p.rq > 0);
    if (!running) {
      running = c->mat(SLOT_TREE + 2, 1);
      HIPCK(hipMemcpyAsync(running, X, c->mat_bytes(), hipMemcpyDeviceToDevice, c->st));
    } else {
      T2* nxt = c->mat(running == c->mat(SLOT_TREE + 2, 1) ? SLOT_TREE + 3 : SLOT_TREE + 2, 1);
      c->zgemm(n == 1 && p.rq > 0 ? 2 : 0, 0, d, d, d, 1.0, X, d, 0, running, d, 0, 0.0, nxt, d, 0, 1);
      running = nxt;
    }
  }
  return running;
}

// Forward chain on the states in psi[cur] (d x B): psi <- U_{S-1} ... U_0 psi, no checkpoints. Returns the buffer index
// holding the result. compose: multiply the propagators together (compose_chain) and apply the product ONCE -- cheaper
// as soon as the batch is wider than the dimension.
template <class T>
int forward_chain(Ctx<T>* c, const std::vector<double>& uh, int S, double dt, int B, int cur, bool compose, bool allow_state_columns = true) {
  typedef TT T2;
  const int d = c->d;
  const int tc = time_chunk(c, std::max(S, 1));
  c->last_chunk = tc;
  if (compose) {
    T2* running = compose_chain(c, uh, S, dt, tc, allow_state_columns);
    if (running) {
      c->zgemm(0, 0, d, B, d, 1.0, running, d, 0, (const T2*)c->psi[cur].p, B, 0, 0.0, (T2*)c->psi[cur ^ 1].p, B, 0, 1);
      cur ^= 1;
    }
    return cur;
  }
  for (int c0 = 0; c0 < S; c0 += tc) {
    const int n = std::min(tc, S - c0);
    const ExpmPlan p = chunk_plan(c, uh, c0, n, dt);
    T2* U = build_propagators(c, p, c0, n, dt);
    for (int t = 0; t < n; ++t) {
      c->zgemm(p.rq > 0 ? 2 : 0, 0, d, B, d, 1.0, U + (size_t)t * c->nn(), d, 0, (const T2*)c->psi[cur].p, B, 0, 0.0,
               (T2*)c->psi[cur ^ 1].p, B, 0, 1);
      cur ^= 1;
    }
  }
  return cur;
}

// ---- state-column form (statecol.hip) -------------------------------------------------------------------------------------
#include "engine_statecol.inc"
#include "engine_matrix_free.inc"

template <class T>
void do_propagate(Ctx<T>* c, const double* u, int S, double dt, const void* psi_in, void* psi_out, int B) {
  typedef TT T2;
  check_ctx(c);
  require_H(c);
  if (S < 0 || B < 0) fail(DQ_ERR_INVALID, "n_steps and batch must be >= 0");
  if (!std::isfinite(dt)) fail(DQ_ERR_INVALID, "dt must be finite");
  if (B == 0) return;
  if (!psi_in || !psi_out) fail(DQ_ERR_INVALID, "null state pointer");
  if (S > 0 && !u && c->K > 0) fail(DQ_ERR_INVALID, "null pulse table");
  const int d = c->d;
  const size_t sb = (size_t)d * B * sizeof(T2);
  reset_call_info(c);
  if (mf_route(c, S, false, false)) {
    const int rc = dq_sctx_propagate(c->mf, u, S, dt, psi_in, psi_out, B);
    if (rc != DQ_OK) fail(rc, std::string("matrix-free routing: ") + dq_sctx_last_error());
    mf_after(c);
    return;
  }
  std::vector<double> uh = fetch_u(c, u, S);
  c->psi[0].ensure(sb);
  c->psi[1].ensure(sb);
  c->last_xs = 0;
  c->last_xs_segments_fwd = 0;
  // A batch wider than the dimension: sweeping the B states costs ~ S B d L row-sweeps, the product of the chain's propagators from
  // the identity's columns (compose_chain) S d d L plus the products that combine its segments and apply it (8 d^2 (B + G d) flop).
  // Rates at d = 256 .. 4096: 2.3 ps per row and sweep, 60 TF/s on the products; the composed form has to win by 30 %.
  bool prefer_compose = false;
  if (S > 0 && B > d && c->compose_mode != 0 && c->xs_compose_mode != 0 && c->xs_mode < 0) {   // (state_columns = 1 asks for the swept states)
    const double Lest = 6.0, segs = std::max(1.0, 5120.0 / d);
    prefer_compose = (double)S * (B - d) * Lest > 0.0754 * d * ((double)B + segs * d);
  }
  if (const int Lx = (S > 0 && !prefer_compose) ? xs_degree(c, uh, S, dt, B) : 0) {
    const XsLaunch<T> x = xs_prepare(c, S, dt, B, Lx, false);
    const T2* p0 = in_dev(c, psi_in, sb, c->psi[1]);
    xs_forward(c, x, p0, S, B, (T2*)c->psi[0].p);
    out_copy(c, psi_out, c->psi[0].p, sb);
    HIPCK(hipStreamSynchronize(c->st));
    return;
  }
  HIPCK(hipMemcpyAsync(c->psi[0].p, psi_in, sb, hipMemcpyDefault, c->st));
  // composing propagators beats applying them one by one as soon as the batch is wider than the dimension
  const bool compose = c->compose_mode == 1 || (c->compose_mode < 0 && (B > d || d <= c->small_dim));
  const int cur = forward_chain(c, uh, S, dt, B, 0, compose);
  c->last_composed = compose ? 1.0 : 0.0;
  out_copy(c, psi_out, c->psi[cur].p, sb);
  HIPCK(hipStreamSynchronize(c->st));
}

template <class T>
void do_energy(Ctx<T>* c, const TT* psi_dev, int B, TT* lam_out /*2 M psi*/, double* energies, double* total) {
  typedef TT T2;
  const int d = c->d;
  c->zgemm(3, 0, d, B, d, 2.0, (const T2*)c->M.p, d, 0, psi_dev, B, 0, 0.0, lam_out, B, 0, 1);  // M Hermitian (checked at set)
  c->e_dev.ensure(sizeof(double) * ((size_t)B + 1));
  c->part.ensure(sizeof(double) * (size_t)dq::coldot_chunks(d) * B);
  c->timed(DQ_PROF_OTHER, 4.0 * d * B, 4.0 * sizeof(T) * d * B,
           [&] { return dq::coldot_launch<T>(psi_dev, B, lam_out, B, d, B, 0.5, c->e_dev.d(), c->part.d(), c->st); });
  c->timed(DQ_PROF_OTHER, B, 8.0 * B, [&] { return dq::sum_launch(c->e_dev.d(), B, c->e_dev.d() + B, c->st); });
  if (energies) out_copy(c, energies, c->e_dev.p, sizeof(double) * (size_t)B);
  if (total) out_copy(c, total, c->e_dev.d() + B, sizeof(double));
}

// Time-parallel form of the operator backward sweep for one chunk of n steps (U = n stacked propagators, W = W at the
// chunk's end). With the suffix products Q_j = U_{n-1} ... U_j (Q_n = I):  W_{j+1} = Q_{j+1}^H W Q_{j+1}, hence
//     G_j = W_{j+1} U_j = Q_{j+1}^H (W Q_j),      W_start = Q_0^H (W Q_0)
// -- every G_j of the chunk from batched products instead of 2n dependent launches (at d <= 1024 one d x d product
// neither fills the chip nor hides its own latency). The suffix products come from a two-level scan: blocks of ~sqrt(n)
// steps (aligned to the chunk's end), local suffixes L_j built block-parallel in b - 1 batched launches, block totals
// chained (nblk - 1 products), Q_j = R_{k-1} L_j per block. About 4n products for the chunk in ~2 sqrt(n) + 4 launches.
// Returns the new W (anti-Hermitian output form if herm_w).
template <class T>
typename dq::Cx<T>::type* tp_sweep(Ctx<T>* c, const typename dq::Cx<T>::type* U, typename dq::Cx<T>::type* G,
                                   typename dq::Cx<T>::type* W, int n, bool herm_w) {
  typedef typename dq::Cx<T>::type T2;
  const int d = c->d;
  const long nn = (long)c->nn();
  int b = (int)std::ceil(std::sqrt((double)n));
  b = std::max(1, std::min(b, n));
  const int nblk = (n + b - 1) / b;
  const int len_first = n - (nblk - 1) * b;  // the earliest block (k = nblk - 1) may be shorter
  T2* L = c->mat(SLOT_TP_L, n);
  T2* Q = c->mat(SLOT_TP_Q, (size_t)n + 1);
  T2* Z = c->mat(SLOT_TP_Z, n);
  T2* R = c->mat(SLOT_TP_R, nblk);
  // block k (k = 0 latest in time) covers j in (n-1-(k+1)b, n-1-kb] cut at 0; its end element is e_k = n-1-kb
  // 1. L[e_k] = U[e_k]
  {
    const int j0 = n - 1 - (nblk - 1) * b;  // smallest end index
    HIPCK(hipMemcpy2DAsync(L + (size_t)j0 * nn, (size_t)b * c->mat_bytes(), U + (size_t)j0 * nn, (size_t)b * c->mat_bytes(),
                           c->mat_bytes(), (size_t)nblk, hipMemcpyDeviceToDevice, c->st));
  }
  // 2. L[e_k - r] = L[e_k - r + 1] U[e_k - r], batched over the blocks that are longer than r
  for (int r = 1; r < b; ++r) {
    const int nb_r = (r < len_first) ? nblk : nblk - 1;
    if (nb_r <= 0) break;
    const int j0 = n - 1 - (nb_r - 1) * b - r;  // smallest j of this launch
    c->zgemm(0, 0, d, d, d, 1.0, L + (size_t)(j0 + 1) * nn, d, (long)b * nn, U + (size_t)j0 * nn, d, (long)b * nn, 0.0,
             L + (size_t)j0 * nn, d, (long)b * nn, nb_r);
  }
  // 3. block suffixes R_k = U_{n-1} ... U_{start_k}:  R_0 = L[start_0],  R_k = R_{k-1} L[start_k]
  auto start_of = [&](int k) { return std::max(0, n - (k + 1) * b); };
  HIPCK(hipMemcpyAsync(R, L + (size_t)start_of(0) * nn, c->mat_bytes(), hipMemcpyDeviceToDevice, c->st));
  for (int k = 1; k < nblk; ++k)
    c->zgemm(0, 0, d, d, d, 1.0, R + (size_t)(k - 1) * nn, d, 0, L + (size_t)start_of(k) * nn, d, 0, 0.0,
             R + (size_t)k * nn, d, 0, 1);
  // 4. Q_j = L_j in the latest block, R_{k-1} L_j in block k >= 1;  Q_n = I
  {
    const int s0 = start_of(0);
    HIPCK(hipMemcpyAsync(Q + (size_t)s0 * nn, L + (size_t)s0 * nn, (size_t)(n - s0) * c->mat_bytes(),
                         hipMemcpyDeviceToDevice, c->st));
    for (int k = 1; k < nblk; ++k) {
      const int sk = start_of(k), len = (n - k * b) - sk;
      c->zgemm(0, 0, d, d, d, 1.0, R + (size_t)(k - 1) * nn, d, 0, L + (size_t)sk * nn, d, nn, 0.0, Q + (size_t)sk * nn, d,
               nn, len);
    }
    c->lincomb(Q + (size_t)n * nn, 0, nullptr, nullptr, 1.0, 1, false);
  }
  // 5. Z_j = W Q_j;  6. G_j = Q_{j+1}^H Z_j;  7. W <- Q_0^H Z_0
  c->zgemm(0, 0, d, d, d, 1.0, W, d, 0, Q, d, nn, 0.0, Z, d, nn, n);
  c->zgemm(1, 0, d, d, d, 1.0, Q + nn, d, nn, Z, d, nn, 0.0, G, d, nn, n);
  T2* Wn = c->mat(W == c->mat(SLOT_TREE, 1) ? SLOT_TREE + 1 : SLOT_TREE, 1);
  c->zgemm(1, 0, d, d, d, 1.0, Q, d, 0, Z, d, 0, 0.0, Wn, d, 0, 1, herm_w ? -1 : 0);
  return Wn;
}

// Propagator-intermediate cache (real path): budget and the steps it covers. With plan_steps the covered share is that
// of the planned chain, so a slice pays the same share of rebuilds as the whole chain. Leaves pc_mode = 1 (store) when
// the cache is usable for this chain.
template <class T>
void setup_pcache(Ctx<T>* c, const std::vector<double>& uh, int S, double dt, int tc, bool opback) {
  for (auto& e : c->pcache) e.valid = false;
  c->slice_valid = false;  // whatever dq_slice_forward left in the arena is gone
  c->pc_mode = 0;
  c->pc_limit = 0;
  c->last_cached = 0;
  c->pc_tc = tc;
  if (!c->real_h || c->real_mode == 0 || c->cache_bytes < 0 || S == 0) return;
  double budget = c->cache_bytes;
  if (budget == 0.0) {
    size_t fr = 0, tot = 0;
    HIPCK(hipMemGetInfo(&fr, &tot));
    // operator form: no state checkpoints compete for the HBM. Capped at 96 GiB by default: larger single allocations
    // cost seconds on first use (measured: 80 GB free of charge, 200 GB 4 s); cache_bytes raises it.
    budget = std::min((opback ? 0.80 : 0.50) * ((double)fr + (double)c->pc_arena.bytes), 96.0 * 1024 * 1024 * 1024);
  }
  const ExpmPlan p0 = chunk_plan(c, uh, 0, std::min(tc, S), dt);
  if (p0.rq == 0 || p0.s > 0) return;
  // per step: X, Y_1..Y_{q-1}, P (real) + U (complex); sparse-generator scheme: X_1..X_2q + U
  const double per_step = ((p0.xr ? 0.0 : (p0.spx ? 2.0 * p0.rq : p0.rq + 1.0)) + 2.0) * c->nn() * sizeof(T);
  const double S_plan = (double)std::max<long>(S, c->plan_steps);
  const double share = std::min(1.0, budget / (per_step * S_plan));
  // the sparse-generator scheme rebuilds with 2q cheap passes: holding tens of GB for a few percent of the steps is not
  // worth it -- only a chain that is mostly cacheable (a rank's slice of a time-sliced chain) uses the cache
  if (p0.spx && share < 0.5) return;
  c->pc_limit = (long)(share * S);
  // one allocation (grows only); if the device refuses it, halve the covered steps until it fits
  while (c->pc_limit > 0) {
    try {
      c->pc_arena.ensure((size_t)(per_step * (double)c->pc_limit) + 4096);
      break;
    } catch (const DqError&) {
      c->pc_limit /= 2;
    }
  }
  if (c->pc_limit <= 0) return;  // no room at all: run without the cache
  c->pc_used = 0;
  c->pc_mode = 1;
}

// whatever happens in a call, later calls must not see a half-filled cache as live
template <class T>
struct PcOff {
  Ctx<T>* c;
  ~PcOff() {
    c->pc_mode = 0;
    for (auto& r : c->redirect) r = nullptr;
  }
};

// Terminal costate Lambda_S (into lam[0]) of the states in psi_S, and the scalar outputs.
// targets == null: energy objective, Lambda_S = 2 M Psi_S; else fidelity objective, c_b = <phi_b|psi_b>, F_b = |c_b|^2,
// Lambda_S = -2 phi .* c, L = B - sum F (`energies` then receives the fidelities).
template <class T>
void terminal_costate(Ctx<T>* c, const TT* psi_S, int B, const void* targets, double* loss, double* energies) {
  typedef TT T2;
  const int d = c->d;
  const size_t sb = (size_t)d * B * sizeof(T2);
  if (!targets) {
    do_energy(c, psi_S, B, (T2*)c->lam[0].p, energies, loss);
    return;
  }
  const T2* phi = in_dev(c, targets, sb, c->tmp_in[3]);
  c->e_dev.ensure(sizeof(double) * (3 * (size_t)B + 2));
  c->part.ensure(sizeof(double) * 2 * (size_t)dq::coldot_chunks(d) * B);
  double* fid = c->e_dev.d();
  double* cvals = c->e_dev.d() + B + 1;
  c->timed(DQ_PROF_OTHER, 8.0 * d * B, 6.0 * sizeof(T) * d * B, [&] {
    return dq::fidelity_costate_launch<T>(phi, psi_S, d, B, -2.0, c->part.d(), cvals, fid, (T2*)c->lam[0].p, c->st);
  });
  c->timed(DQ_PROF_OTHER, B, 8.0 * B, [&] { return dq::sum_launch(fid, B, fid + B, c->st); });
  if (energies) out_copy(c, energies, fid, sizeof(double) * (size_t)B);
  if (loss) {
    double fsum = 0.0;
    HIPCK(hipMemcpyAsync(&fsum, fid + B, sizeof(double), hipMemcpyDeviceToHost, c->st));
    HIPCK(hipStreamSynchronize(c->st));
    const double l = (double)B - fsum;
    HIPCK(hipMemcpyAsync(loss, &l, sizeof(double), hipMemcpyDefault, c->st));
    HIPCK(hipStreamSynchronize(c->st));
  }
}

// Only the anti-Hermitian part of W reaches the gradient: dU_t = U_t K with K anti-Hermitian (U_t unitary up to the
// Taylor remainder), and Re tr((W U)^H U K) = Re tr((U^H W U)^H K) vanishes for Hermitian W. With W anti-Hermitian,
// W_t = U_t^H (W_{t+1} U_t) is anti-Hermitian for ANY U_t: that product computes the upper triangle only.
// (Off when a polynomial degree is forced: the gradient is then the exact derivative of an under-resolved polynomial.)
template <class T>
bool skew_w_form(Ctx<T>* c) {
  return c->skew_trick != 0 && c->force_degree <= 0 && c->real_mode <= 0;
}

// W_S = Lambda_S Psi_S^H (Lambda_S in lam[0]), reduced to its anti-Hermitian part if skew_w. Returns the pool matrix.
template <class T>
TT* terminal_w(Ctx<T>* c, const TT* psi_S, int B, bool skew_w) {
  typedef TT T2;
  const int d = c->d;
  T2* W = c->mat(SLOT_TREE, 1);
  c->zgemm(0, 1, d, d, B, 1.0, (const T2*)c->lam[0].p, B, 0, psi_S, B, 0, 0.0, W, d, 0, 1);
  if (skew_w) {
    T2* Ws = c->mat(SLOT_TREE + 1, 1);
    c->timed(DQ_PROF_LINCOMB, 4.0 * c->nn(), 3.0 * 2.0 * sizeof(T) * c->nn(),
             [&] { return dq::tcombine_launch<T>(Ws, W, d, 0.5, -0.5, 0, (long)c->nn(), 1, c->st); });
    W = Ws;
  }
  return W;
}

// Operator-form backward sweep over one chunk of n built propagators U (descending in time): G_t = W_{t+1} U_t into the
// cotangent slots G, W_t = U_t^H G_t. W must be one of the pool matrices SLOT_TREE / SLOT_TREE + 1; returns the new W.
template <class T>
TT* opback_chunk(Ctx<T>* c, const TT* U, TT* G, TT* W, int n, bool skew_w) {
  typedef TT T2;
  const int d = c->d;
  // auto: chunks of >= 4 steps at dimensions where one d x d product leaves most of the chip idle (d <= 1024)
  const bool tpar = c->time_parallel == 1 ? n >= 2
                                          : (c->time_parallel < 0 && n >= 4 && (d + 127) / 128 * ((d + 127) / 128) < c->n_cu);
  c->last_tpar = tpar ? 1.0 : 0.0;
  if (tpar) return tp_sweep(c, U, G, W, n, skew_w);
  for (int t = n - 1; t >= 0; --t) {
    const T2* Ut = U + (size_t)t * c->nn();
    T2* Gt = G + (size_t)t * c->nn();
    c->zgemm(skew_w ? 4 : 0, 0, d, d, d, 1.0, W, d, 0, Ut, d, 0, 0.0, Gt, d, 0, 1);  // G_t = W_{t+1} U_t (W anti-Hermitian as stored)
    T2* Wn = c->mat(W == c->mat(SLOT_TREE, 1) ? SLOT_TREE + 1 : SLOT_TREE, 1);
    c->zgemm(1, 0, d, d, d, 1.0, Ut, d, 0, Gt, d, 0, 0.0, Wn, d, 0, 1, skew_w ? -1 : 0);  // W_t = U_t^H G_t
    W = Wn;
  }
  return W;
}

// Reverse mode through the chunk's exponentials (their variables are in the pool, the cotangents G_t of the propagators
// in ubar_slot) and the contractions with the control Hamiltonians: grad_dev[c0 .. c0 + n)[K].
template <class T>
void chunk_vjp(Ctx<T>* c, const ExpmPlan& p, int c0, int n, double dt) {
  typedef TT T2;
  const int K = c->K;
  const int nblk = dq::grad_contract_blocks((long)c->nn());
  c->part.ensure(sizeof(double) * (size_t)n * nblk * std::max(K, 1));
  if (p.zxr) {
    const T2* G = ubar_slot(c, p, n);
    c->xr_part.ensure(sizeof(double) * (size_t)n * dq::xr_blocks(c->d, n) * dq::XR_SLOTS);
    c->xr_pow.ensure(2 * sizeof(double) * (size_t)n * dq::zxr_pow_elems(c->d, p.m, n));
    const double rows = (double)c->zxr_tab.R + 1.0;
    c->timed(DQ_PROF_SPMM, 8.0 * rows * c->nn() * (2.0 * p.m - 1.0) * n, (double)sizeof(T2) * c->nn() * n, [&] {
      return zxr_reverse_any(G, (long)c->nn(), c->d, c->zxr_tab, c->u_dev.d() + (size_t)c0 * K, K, dt, p.m, n, (double*)c->xr_part.p,
                             c->xr_pow.p, c->grad_dev.d() + (size_t)c0 * K, c->st);
    });
    return;
  }
  if (p.rq > 0 && p.spx) {
    real_backward_spx(c, p, c0, n, dt, c->grad_dev.d() + (size_t)c0 * K);
    return;
  }
  if (p.rq > 0 && c->sparse_on()) {
    const T *Sb = nullptr, *Yb = nullptr;
    real_backward(c, p, n, true, &Sb, &Yb);
    c->sp_nzval.ensure(sizeof(double) * (size_t)std::max<long>(c->sp_nnz, 1) * n);
    // bytes: the two cotangent rows once, the partner rows once per nonzero (mostly L2 hits: an upper bound)
    const double bys = (double)sizeof(T) * c->d * (2.0 * c->d + 2.0 * c->sp_nnz) * n;
    c->timed(DQ_PROF_GRADC, 4.0 * c->sp_nnz * c->d * n, bys, [&] {
      return dq::rsp_contract_launch<T>(Sb, c->rmat(RS_P, n), Yb, c->rmat(RS_X, n), (long)c->nn(), c->d,
                                        (const long*)c->sp_rowptr.p, (const int*)c->sp_col.p, (const double*)c->sp_val.p,
                                        c->sp_nnz, (const long*)c->sp_kptr.p, (const long*)c->sp_kent.p, K, n,
                                        -std::ldexp(dt, -p.s), (double*)c->sp_nzval.p,
                                        c->grad_dev.d() + (size_t)c0 * K, c->st);
    });
    return;
  }
  if (p.rq > 0) {
    const T* Xbar = real_backward(c, p, n);
    const double byr = (double)sizeof(T) * c->nn() * (K + ((K + 15) / 16)) * n;
    c->timed(DQ_PROF_GRADC, 2.0 * c->nn() * K * n, byr, [&] {
      return dq::rgrad_contract_launch<T>(Xbar, (long)c->nn(), (const T*)c->Hsr.p, K, (long)c->nn(), n, -std::ldexp(dt, -p.s),
                                          c->part.d(), c->grad_dev.d() + (size_t)c0 * K, c->st);
    });
    return;
  }
  const T2* Abar = expm_backward(c, p, n, c->skew_trick != 0);
  const double by = 2.0 * sizeof(T) * c->nn() * (K + ((K + 7) / 8)) * n;
  c->timed(DQ_PROF_GRADC, 3.0 * c->nn() * K * n, by, [&] {
    return dq::grad_contract_launch<T>(Abar, (long)c->nn(), (const T2*)c->Hs.p, K, (long)c->nn(), n, dt, c->part.d(),
                                       c->grad_dev.d() + (size_t)c0 * K, c->st);
  });
}

#include "engine_statecol_grad.inc"

template <class T>
void do_energy_grad(Ctx<T>* c, const double* u, int S, double dt, const void* psi0, int B, double* loss,
                    double* grad_u, void* psi_final, double* energies, const void* targets = nullptr) {
  // targets == null: energy objective L = sum_b <psi_b|M|psi_b>;  else fidelity objective
  // L = sum_b (1 - |<phi_b|psi_b(T)>|^2) with phi = targets ([dim][batch]); `energies` then receives the fidelities.
  typedef TT T2;
  check_ctx(c);
  require_H(c);
  if (!targets && !c->have_M) fail(DQ_ERR_STATE, "dq_set_observable has not been called");
  if (!c->hermitian)
    fail(DQ_ERR_UNSUPPORTED,
         "adjoint sweep needs Hermitian H0/H_k (it rebuilds psi_t = U_t^H psi_{t+1} between checkpoints)");
  if (S < 0 || B <= 0) fail(DQ_ERR_INVALID, "n_steps must be >= 0 and batch > 0");
  if (!std::isfinite(dt)) fail(DQ_ERR_INVALID, "dt must be finite");
  if (!psi0) fail(DQ_ERR_INVALID, "null state pointer");
  if (S > 0 && !u && c->K > 0) fail(DQ_ERR_INVALID, "null pulse table");
  const int d = c->d, K = c->K;
  const size_t sb = (size_t)d * B * sizeof(T2);
  reset_call_info(c);
  if (mf_route(c, S, true, targets != nullptr)) {   // a Pauli-sum generator beyond the state-column kernels' dimensions
    const int rc = dq_sctx_energy_grad(c->mf, u, S, dt, psi0, B, loss, grad_u, psi_final, energies);
    if (rc != DQ_OK) fail(rc, std::string("matrix-free routing: ") + dq_sctx_last_error());
    mf_after(c);
    return;
  }
  // Operator-form backward sweep for batch >= dim. The cotangent of U_t is G_t = Lambda_{t+1} Psi_t^H; with
  // W_t = Lambda_t Psi_t^H (d x d) and Psi_t = U_t^H Psi_{t+1}, Lambda_t = U_t^H Lambda_{t+1}:
  //     G_t = W_{t+1} U_t,     W_t = U_t^H G_t,     W_S = Lambda_S Psi_S^H
  // -- two d x d x d products per step instead of two d x d x B ones, and NO intermediate states: no checkpoints, no
  // psi_t rebuilds, memory independent of the chain length. (Unitary conjugations: the drift of W over S steps is
  // S * eps, 2e-13 relative at S = 2000.) The forward pass then needs no checkpoints either and, for batch > dim,
  // composes the propagators (forward_chain).
  // Small dimensions (d <= small_dim, default 512) take the operator form for ANY batch: a d x d x B product per step is
  // launch-latency bound there, while the operator form batches over the time chunk (tree-composed forward,
  // time-parallel backward below).
  // (state form: G and Lambda cost 2 x 3 x B/d units per step, operator form 3 + 1.5 -> the crossover is at B = 3d/4)
  // A forced (possibly under-resolved) polynomial keeps the state form in auto mode: the operator form uses
  // U_t^{-1} = U_t^H, exact only up to the Taylor remainder.
  const bool forced_plan = c->force_degree > 0 || c->real_mode > 0;
  const bool opback = S > 0 && (c->compose_mode == 1 ||
                                (c->compose_mode < 0 && !forced_plan && (4L * B >= 3L * d || d <= c->small_dim)));
  const bool compose = c->compose_mode == 1 || (c->compose_mode < 0 && (B > d || d <= c->small_dim));
  c->last_composed = (opback && compose) ? 1.0 : 0.0;
  c->last_opback = opback ? 1.0 : 0.0;
  std::vector<double> uh = fetch_u(c, u, S);
  c->last_xs = 0;
  c->last_xs_segments_fwd = 0;
  c->last_xs_wide = 0;
  if (S > 0 && B > d && !targets && c->xs_compose_mode != 0 && (c->xs_wide_mode > 0 || (c->xs_wide_mode < 0 && c->xs_mode < 0)) && c->compose_mode != 0) {   // (state_columns = 1 asks for the swept states)
    // (-1: by cost -- fwd + adjoint sweeps over B - d columns saved against the Gram product, three d^3 products and the final
    // application; rates as in do_propagate)
    // (measured at d = 4096, B = 32768, S = 20, tools/wide_adjoint_timing.py: 77 ms against 137 ms without the final application,
    // about equal with it: a pairing sweep counts 1.4 plain ones, the products run at 90 TF/s algorithmic, 15 % margin)
    bool take = c->xs_wide_mode > 0;
    const bool need_apply = psi_final != nullptr || energies != nullptr;
    if (!take) take = (double)S * (B - d) * 22.8 > 0.0444 * d * ((need_apply ? 2.0 : 1.0) * B + 8.0 * d);
    if (take)
      if (const int Lw = xs_degree(c, uh, S, dt, d, true)) {
        do_energy_grad_xs_wide(c, uh, S, dt, psi0, B, loss, grad_u, psi_final, energies, Lw);
        return;
      }
  }
  if (const int Lx = S > 0 ? xs_degree(c, uh, S, dt, B) : 0) {
    do_energy_grad_xs(c, S, dt, psi0, B, loss, grad_u, psi_final, energies, targets, Lx);
    return;
  }
  for (int i = 0; i < 2; ++i) c->psi[i].ensure(sb);
  c->lam[0].ensure(sb);
  c->grad_dev.ensure(std::max<size_t>((size_t)S * K, 1) * sizeof(double));
  HIPCK(hipMemcpyAsync(c->psi[0].p, psi0, sb, hipMemcpyDefault, c->st));
  int cur = 0;
  const int tc = time_chunk(c, std::max(S, 1));
  c->last_chunk = tc;
  PcOff<T> pc_off{c};

  // ---- forward ----
  int cstride = 1;
  if (opback) {
    c->last_cstride = 0;  // no state checkpoints
    setup_pcache(c, uh, S, dt, tc, opback);
    cur = forward_chain(c, uh, S, dt, B, 0, compose, false);   // (explicit propagators: they fill the cache the operator-form sweep reads)
  } else {
    // checkpoint stride from the memory budget
    double budget = c->checkpoint_bytes;
    if (budget <= 0.0) {
      // default: 70 % of what is free now (the checkpoints already held by this context count as free); the rest
      // covers the workspace pool (~25 d x d matrices) and the caller's own buffers
      size_t fr = 0, tot = 0;
      HIPCK(hipMemGetInfo(&fr, &tot));
      double held = 0.0;
      for (const auto& b : c->ck) held += (double)b.bytes;
      budget = 0.70 * ((double)fr + held);
    }
    long ncp_max = std::max(1L, (long)(budget / (double)sb));
    // plan_steps: size the stride for a longer chain than this call's (bench: a slice of the full configuration runs
    // with the checkpoint/rebuild ratio of the whole chain)
    const long S_plan = std::max<long>(S, c->plan_steps);
    cstride = (int)std::max(1L, (S_plan + ncp_max - 1) / ncp_max);
    const int ncp = S > 0 ? (S + cstride - 1) / cstride : 0;
    c->last_cstride = cstride;
    if ((int)c->ck.size() < ncp) c->ck.resize(ncp);
    for (int i = 0; i < ncp; ++i) c->ck[i].ensure(sb);
    c->psi[2].ensure(sb);
    c->lam[1].ensure(sb);
    setup_pcache(c, uh, S, dt, tc, opback);  // (after the checkpoints: it takes half of what they left)
    for (int c0 = 0; c0 < S; c0 += tc) {
      const int n = std::min(tc, S - c0);
      const ExpmPlan p = chunk_plan(c, uh, c0, n, dt);
      const T2* U = build_propagators(c, p, c0, n, dt);
      for (int t = 0; t < n; ++t) {
        const int tt = c0 + t;
        if (tt % cstride == 0) {
          const size_t bytes = sb;
          void* dst = c->ck[tt / cstride].p;
          const void* src = c->psi[cur].p;
          c->timed(DQ_PROF_OTHER, 0.0, 2.0 * bytes,
                   [&] { return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, c->st); });
        }
        c->zgemm(p.rq > 0 ? 2 : 0, 0, d, B, d, 1.0, U + (size_t)t * c->nn(), d, 0, (const T2*)c->psi[cur].p, B, 0, 0.0,
                 (T2*)c->psi[cur ^ 1].p, B, 0, 1);
        cur ^= 1;
      }
    }
  }

  // ---- terminal costate Lambda_S (into lam[0]) and the outputs; psi[cur] = Psi_S ----
  terminal_costate(c, (const T2*)c->psi[cur].p, B, targets, loss, energies);
  if (psi_final) out_copy(c, psi_final, c->psi[cur].p, sb);

  // ---- backward ----
  if (c->pc_mode == 1) c->pc_mode = 2;
  if (S > 0) {
    const int last_c0 = ((S - 1) / tc) * tc;
    const T2* psi_next = (const T2*)c->psi[cur].p;
    int lcur = 0;
    const bool skew_w = opback && skew_w_form(c);
    T2* W = opback ? terminal_w(c, psi_next, B, skew_w) : nullptr;  // operator form: W_{t+1}
    for (int c0 = last_c0; c0 >= 0; c0 -= tc) {
      const int n = std::min(tc, S - c0);
      const ExpmPlan p = chunk_plan(c, uh, c0, n, dt);
      const T2* U = build_propagators(c, p, c0, n, dt);
      T2* G = ubar_slot(c, p, n);
      if (opback) W = opback_chunk(c, U, G, W, n, skew_w);
      for (int t = opback ? -1 : n - 1; t >= 0; --t) {
        const int tt = c0 + t;
        const T2* Ut = U + (size_t)t * c->nn();
        T2* Gt = G + (size_t)t * c->nn();
        const T2* psi_t;
        if (tt % cstride == 0) {
          psi_t = (const T2*)c->ck[tt / cstride].p;
        } else {
          // pick a scratch buffer that does not hold psi_next
          T2* dst = nullptr;
          for (int i = 0; i < 3; ++i)
            if ((const T2*)c->psi[i].p != psi_next) { dst = (T2*)c->psi[i].p; break; }
          c->zgemm(1, 0, d, B, d, 1.0, Ut, d, 0, psi_next, B, 0, 0.0, dst, B, 0, 1);  // psi_t = U_t^H psi_{t+1}
          psi_t = dst;
        }
        // G_t = Lambda_{t+1} Psi_t^H    (cotangent of U_t)
        c->zgemm(0, 1, d, d, B, 1.0, (const T2*)c->lam[lcur].p, B, 0, psi_t, B, 0, 0.0, Gt, d, 0, 1);
        // Lambda_t = U_t^H Lambda_{t+1}
        c->zgemm(1, 0, d, B, d, 1.0, Ut, d, 0, (const T2*)c->lam[lcur].p, B, 0, 0.0, (T2*)c->lam[lcur ^ 1].p, B, 0, 1);
        lcur ^= 1;
        psi_next = psi_t;
      }
      chunk_vjp(c, p, c0, n, dt);
    }
  }
  c->pc_mode = 0;
  if (grad_u && S > 0 && K > 0) out_copy(c, grad_u, c->grad_dev.p, sizeof(double) * (size_t)S * K);
  HIPCK(hipStreamSynchronize(c->st));
}

// ---- time-sliced operator form ---------------------------------------------------------------
// In operator form the cost of a gradient evaluation does not depend on the batch: the forward pass is the product
// Q = U_{S-1} ... U_0 applied once, the backward sweep runs on the d x d matrix W. Both split over TIME: a rank that owns
// the steps [a, b) builds only their propagators, contributes their product P = U_{b-1} ... U_a, and -- given W at time b,
// W_b = R^H W_S R with R = U_{S-1} ... U_b the product of the later slices -- sweeps only its own steps. The three calls
// below are those phases of dq_energy_grad; the host composes the slice products and conjugates W_S between them
// (dq_zgemm), with one all-gather of the P's, one all-reduce of W_S (the batch shards' sum) and one all-gather of the
// gradient rows (diffquantum_amd/distributed.py). The propagator cache filled by the forward call serves the backward call.
template <class T>
void do_slice_forward(Ctx<T>* c, const double* u, int n, double dt, void* P_out) {
  typedef TT T2;
  check_ctx(c);
  require_H(c);
  if (!c->hermitian) fail(DQ_ERR_UNSUPPORTED, "the operator form needs Hermitian H0/H_k (it uses U_t^{-1} = U_t^H)");
  if (n < 0) fail(DQ_ERR_INVALID, "n_steps must be >= 0");
  if (!std::isfinite(dt)) fail(DQ_ERR_INVALID, "dt must be finite");
  if (!P_out) fail(DQ_ERR_INVALID, "null output");
  if (n > 0 && !u && c->K > 0) fail(DQ_ERR_INVALID, "null pulse table");
  std::vector<double> uh = fetch_u(c, u, n);
  const int tc = time_chunk(c, std::max(n, 1));
  c->last_chunk = tc;
  PcOff<T> pc_off{c};
  //  decide the state-column product first: when it is taken no propagator is ever built here, so no arena is reserved
  // (up to 80 % of the free HBM, seconds on first allocation) and the backward call goes straight to its own column kernels
  const bool xs_product = n > 0 && c->xs_compose_mode != 0 && xs_degree(c, uh, n, dt, c->d, true) > 0;
  if (!xs_product) setup_pcache(c, uh, n, dt, tc, true);
  else c->pc_limit = 0;
  // (Pauli-sum generators: the slice's product from state-column segments; the backward call then finds no cached propagators and
  // builds its own -- 0.2 ms per step on the column kernels against the 5.8 ms product per step this saves at d = 4096)
  T2* P = compose_chain(c, uh, n, dt, tc, true);
  if (!P) {
    P = c->mat(SLOT_TREE + 2, 1);
    c->lincomb(P, 0, nullptr, nullptr, 1.0, 1, false);  // empty slice: the identity
  }
  out_copy(c, P_out, P, c->mat_bytes());
  HIPCK(hipStreamSynchronize(c->st));
  c->slice_u = uh;
  c->slice_dt = dt;
  c->slice_tc = tc;
  c->slice_valid = true;
}

template <class T>
void do_slice_terminal(Ctx<T>* c, const void* Q, const void* psi0, int B, const void* targets, double* loss,
                       double* energies, void* psi_final, void* W_out) {
  typedef TT T2;
  check_ctx(c);
  if (!targets && !c->have_M) fail(DQ_ERR_STATE, "dq_set_observable has not been called");
  if (B <= 0) fail(DQ_ERR_INVALID, "batch must be > 0");
  if (!Q || !psi0 || !W_out) fail(DQ_ERR_INVALID, "null argument");
  const int d = c->d;
  const size_t sb = (size_t)d * B * sizeof(T2);
  for (int i = 0; i < 2; ++i) c->psi[i].ensure(sb);
  c->lam[0].ensure(sb);
  const T2* Qd = in_dev(c, Q, c->mat_bytes(), c->tmp_in[0]);
  const T2* p0 = in_dev(c, psi0, sb, c->psi[0]);
  T2* psi_S = (T2*)c->psi[1].p;
  c->zgemm(0, 0, d, B, d, 1.0, Qd, d, 0, p0, B, 0, 0.0, psi_S, B, 0, 1);  // Psi_S = Q Psi_0
  terminal_costate(c, psi_S, B, targets, loss, energies);
  if (psi_final) out_copy(c, psi_final, psi_S, sb);
  const T2* W = terminal_w(c, psi_S, B, skew_w_form(c));
  out_copy(c, W_out, W, c->mat_bytes());
  HIPCK(hipStreamSynchronize(c->st));
}

template <class T>
void do_slice_backward(Ctx<T>* c, const double* u, int n, double dt, const void* W_end, double* grad_u, void* W_start) {
  typedef TT T2;
  check_ctx(c);
  require_H(c);
  if (!c->hermitian) fail(DQ_ERR_UNSUPPORTED, "the operator form needs Hermitian H0/H_k (it uses U_t^{-1} = U_t^H)");
  if (n < 0) fail(DQ_ERR_INVALID, "n_steps must be >= 0");
  if (!std::isfinite(dt)) fail(DQ_ERR_INVALID, "dt must be finite");
  if (!W_end) fail(DQ_ERR_INVALID, "null W");
  if (n > 0 && !u && c->K > 0) fail(DQ_ERR_INVALID, "null pulse table");
  const int K = c->K;
  std::vector<double> uh = fetch_u(c, u, n);
  // the forward call's cache is live only for the very same slice
  const bool same = c->slice_valid && c->slice_dt == dt && c->slice_u == uh;
  const int tc = same ? c->slice_tc : time_chunk(c, std::max(n, 1));
  c->last_chunk = tc;
  PcOff<T> pc_off{c};
  if (same) {
    c->pc_mode = c->pc_limit > 0 ? 2 : 0;
  } else {
    for (auto& e : c->pcache) e.valid = false;
    c->slice_valid = false;
    c->pc_mode = 0;
  }
  c->grad_dev.ensure(std::max<size_t>((size_t)n * K, 1) * sizeof(double));
  const bool skew_w = skew_w_form(c);
  T2* W = c->mat(SLOT_TREE, 1);
  HIPCK(hipMemcpyAsync(W, W_end, c->mat_bytes(), hipMemcpyDefault, c->st));
  if (n > 0) {
    const int last_c0 = ((n - 1) / tc) * tc;
    for (int c0 = last_c0; c0 >= 0; c0 -= tc) {
      const int m = std::min(tc, n - c0);
      const ExpmPlan p = chunk_plan(c, uh, c0, m, dt);
      const T2* U = build_propagators(c, p, c0, m, dt);
      T2* G = ubar_slot(c, p, m);
      W = opback_chunk(c, U, G, W, m, skew_w);
      chunk_vjp(c, p, c0, m, dt);
    }
  }
  if (grad_u && n > 0 && K > 0) out_copy(c, grad_u, c->grad_dev.p, sizeof(double) * (size_t)n * K);
  if (W_start) out_copy(c, W_start, W, c->mat_bytes());
  HIPCK(hipStreamSynchronize(c->st));
  c->slice_valid = false;  // one backward per forward
}

// ---- explicit propagators in and out (SURVEY.md section 8-F2 in its literal form: "each GPU builds S/G propagators and
// all-gathers them"): the step propagators U_t = exp(-i dt (H0 + sum_k u[t][k] H_k)) of a set of pulse rows as plain
// [n][dim][dim] matrices, and the pull-back of cotangents G_t of those propagators to the pulse rows,
//     dL/du[t][k] = Re <G_t, dU_t/du_k>      (pairing Re tr(G^H dU); the exp's exact reverse mode + the control contractions).
// With dq_zgemm these two calls let a host run the STATE-form chain on propagators that were built elsewhere (another rank):
// psi_{t+1} = U_t psi_t, lambda_t = U_t^H lambda_{t+1}, G_t = lambda_{t+1} psi_t^H (diffquantum_amd/distributed.py:
// propagator_sharded_energy_grad). The reference builds every U_t where it uses it (diffqc.cc:192-198, sim_plain.py:137-146).
template <class T>
void do_build_propagators(Ctx<T>* c, const double* u, int n, double dt, void* U_out) {
  typedef TT T2;
  check_ctx(c);
  require_H(c);
  if (n < 0) fail(DQ_ERR_INVALID, "n_steps must be >= 0");
  if (!std::isfinite(dt)) fail(DQ_ERR_INVALID, "dt must be finite");
  if (n > 0 && !U_out) fail(DQ_ERR_INVALID, "null output");
  if (n > 0 && !u && c->K > 0) fail(DQ_ERR_INVALID, "null pulse table");
  if (n == 0) return;
  std::vector<double> uh = fetch_u(c, u, n);
  const int tc = time_chunk(c, n);
  c->last_chunk = tc;
  PcOff<T> pc_off{c};
  for (auto& e : c->pcache) e.valid = false;
  c->slice_valid = false;
  c->pc_mode = 0;
  for (int c0 = 0; c0 < n; c0 += tc) {
    const int m = std::min(tc, n - c0);
    const ExpmPlan p = chunk_plan(c, uh, c0, m, dt);
    const T2* U = build_propagators(c, p, c0, m, dt);
    out_copy(c, (char*)U_out + (size_t)c0 * c->mat_bytes(), U, (size_t)m * c->mat_bytes());
  }
  HIPCK(hipStreamSynchronize(c->st));
}

template <class T>
void do_propagator_vjp(Ctx<T>* c, const double* u, int n, double dt, const void* G_in, double* grad_u) {
  typedef TT T2;
  check_ctx(c);
  require_H(c);
  if (!c->hermitian) fail(DQ_ERR_UNSUPPORTED, "the gradient path needs Hermitian H0/H_k");
  if (n < 0) fail(DQ_ERR_INVALID, "n_steps must be >= 0");
  if (!std::isfinite(dt)) fail(DQ_ERR_INVALID, "dt must be finite");
  if (n > 0 && (!G_in || !grad_u)) fail(DQ_ERR_INVALID, "null argument");
  if (n > 0 && !u && c->K > 0) fail(DQ_ERR_INVALID, "null pulse table");
  const int K = c->K;
  if (n == 0 || K == 0) return;
  std::vector<double> uh = fetch_u(c, u, n);
  const int tc = time_chunk(c, n);
  c->last_chunk = tc;
  PcOff<T> pc_off{c};
  for (auto& e : c->pcache) e.valid = false;
  c->slice_valid = false;
  c->pc_mode = 0;
  c->grad_dev.ensure((size_t)n * K * sizeof(double));
  for (int c0 = 0; c0 < n; c0 += tc) {
    const int m = std::min(tc, n - c0);
    const ExpmPlan p = chunk_plan(c, uh, c0, m, dt);
    if (!p.xr && !p.zxr) build_propagators(c, p, c0, m, dt);  // the polynomial's variables, for the reverse mode (column form: none)
    else for (auto& r : c->redirect) r = nullptr;
    T2* G = ubar_slot(c, p, m);
    HIPCK(hipMemcpyAsync(G, (const char*)G_in + (size_t)c0 * c->mat_bytes(), (size_t)m * c->mat_bytes(), hipMemcpyDefault, c->st));
    chunk_vjp(c, p, c0, m, dt);
  }
  out_copy(c, grad_u, c->grad_dev.p, sizeof(double) * (size_t)n * K);
  HIPCK(hipStreamSynchronize(c->st));
}

// The reference's "parameter-shift in time" Monte-Carlo estimator as one device-resident operation
// (compute_energy_grad_MC, sim_plain.py:156-231): evolve 0 -> s on the first time grid, apply the 2K gates
// (I +- i r H_k)/sqrt(1 + r^2), evolve all 2K shifted states s -> T on the second grid as ONE batch, measure M, and
// return ps[b][k] = (1 + r^2)/(2 r) (<M>_- - <M>_+)  (sim_plain.py:220). The chain rule to the spectral coefficients
// (du_k(s)/dv, sim_plain.py:170-184,227) stays on the host.
template <class T>
void do_mc_shift(Ctx<T>* c, const double* u_a, int n_a, double dt_a, const double* u_b, int n_b, double dt_b,
                 const void* psi0, int B, double r, double* ps, void* phi_out) {
  typedef TT T2;
  check_ctx(c);
  require_H(c);
  if (!c->have_M) fail(DQ_ERR_STATE, "dq_set_observable has not been called");
  if (n_a < 0 || n_b < 0 || B <= 0) fail(DQ_ERR_INVALID, "step counts must be >= 0 and batch > 0");
  if (!std::isfinite(dt_a) || !std::isfinite(dt_b) || !(r > 0.0) || !std::isfinite(r))
    fail(DQ_ERR_INVALID, "dt must be finite and r > 0");
  if (!psi0 || !ps) fail(DQ_ERR_INVALID, "null pointer");
  if (c->K > 0 && ((n_a > 0 && !u_a) || (n_b > 0 && !u_b))) fail(DQ_ERR_INVALID, "null pulse table");
  const int d = c->d, K = c->K;
  const int B2 = 2 * K * B;
  const size_t sb = (size_t)d * B * sizeof(T2), sb2 = (size_t)d * std::max(B2, 1) * sizeof(T2);
  const bool small = d <= c->small_dim;
  // phase 1: phi = chain_a psi0
  std::vector<double> ua = fetch_u(c, u_a, n_a);
  for (int i = 0; i < 2; ++i) c->psi[i].ensure(std::max(sb, sb2));
  HIPCK(hipMemcpyAsync(c->psi[0].p, psi0, sb, hipMemcpyDefault, c->st));
  int cur = forward_chain(c, ua, n_a, dt_a, B, 0, c->compose_mode == 1 || (c->compose_mode < 0 && (B > d || small)));
  if (phi_out) out_copy(c, phi_out, c->psi[cur].p, sb);
  if (K == 0) { HIPCK(hipStreamSynchronize(c->st)); return; }
  // phase 2: H_k phi for all k (one batched product), the 2K shifted states per trajectory
  c->lam[0].ensure((size_t)K * sb);
  c->lam[1].ensure(sb2);
  c->zgemm(0, 0, d, B, d, 1.0, (const T2*)c->Hs.p, d, (long)c->nn(), (const T2*)c->psi[cur].p, B, 0, 0.0, (T2*)c->lam[0].p, B,
           (long)d * B, K);
  c->timed(DQ_PROF_OTHER, 8.0 * d * B * K, (double)sizeof(T2) * d * B * (1.0 + 3.0 * K), [&] {
    return dq::mc_shift_launch<T>((const T2*)c->psi[cur].p, (const T2*)c->lam[0].p, d, B, K, r, (T2*)c->lam[1].p, c->st);
  });
  // phase 3: evolve the shifted batch s -> T and measure
  std::vector<double> ub = fetch_u(c, u_b, n_b);
  HIPCK(hipMemcpyAsync(c->psi[0].p, c->lam[1].p, sb2, hipMemcpyDeviceToDevice, c->st));
  cur = forward_chain(c, ub, n_b, dt_b, B2, 0, c->compose_mode == 1 || (c->compose_mode < 0 && (B2 > d || small)));
  std::vector<double> en(B2);
  do_energy(c, (const T2*)c->psi[cur].p, B2, (T2*)c->lam[1].p, nullptr, nullptr);
  HIPCK(hipMemcpyAsync(en.data(), c->e_dev.p, sizeof(double) * (size_t)B2, hipMemcpyDeviceToHost, c->st));
  HIPCK(hipStreamSynchronize(c->st));
  std::vector<double> out((size_t)B * K);
  const double f = (1.0 + r * r) / (2.0 * r);
  for (int b = 0; b < B; ++b)
    for (int k = 0; k < K; ++k) out[(size_t)b * K + k] = f * (en[((size_t)b * K + k) * 2 + 1] - en[((size_t)b * K + k) * 2]);
  HIPCK(hipMemcpyAsync(ps, out.data(), sizeof(double) * out.size(), hipMemcpyDefault, c->st));
  HIPCK(hipStreamSynchronize(c->st));
}

// Is the sparse generator XOR-regular (xorexp.hip)?  Group the CSR's entries by (column ^ row, source): every group off the
// diagonal must hold one entry per row, all of one magnitude (signs free); groups at offset 0 may hold anything.  Fills
// c->xr_tab (sign words and diagonal values in device memory) and sets c->xr_ok.
template <class T>
void find_xor_regular(Ctx<T>* c, const std::vector<long>& rowptr, long nnz) {
  c->xr_ok = false;
  const int d = c->d;
  if (nnz <= 0 || d < 2 || (d & (d - 1)) != 0) return;
  std::vector<int> col((size_t)nnz), kid((size_t)nnz);
  std::vector<double> val((size_t)nnz);
  HIPCK(hipMemcpyAsync(col.data(), c->sp_col.p, sizeof(int) * (size_t)nnz, hipMemcpyDeviceToHost, c->st));
  HIPCK(hipMemcpyAsync(kid.data(), c->sp_kid.p, sizeof(int) * (size_t)nnz, hipMemcpyDeviceToHost, c->st));
  HIPCK(hipMemcpyAsync(val.data(), c->sp_val.p, sizeof(double) * (size_t)nnz, hipMemcpyDeviceToHost, c->st));
  HIPCK(hipStreamSynchronize(c->st));
  dq::XrTab tab;
  memset(&tab, 0, sizeof(tab));
  struct Grp { int x, kid, j; long count; double mag; };
  std::vector<Grp> groups;  // off-diagonal groups in order of first appearance (row 0 names them all, if regular)
  std::vector<int> dsrc;    // sources with diagonal entries
  std::vector<unsigned> sgn((size_t)d, 0u);
  std::vector<double> dval;
  bool neg = false;
  for (int i = 0; i < d; ++i)
    for (long e = rowptr[i]; e < rowptr[i + 1]; ++e) {
      const int x = col[e] ^ i;
      if (x == 0) {
        size_t jd = 0;
        while (jd < dsrc.size() && dsrc[jd] != kid[e]) ++jd;
        if (jd == dsrc.size()) {
          if ((int)dsrc.size() == dq::XR_DMAX) return;
          dsrc.push_back(kid[e]);
          dval.resize(dsrc.size() * (size_t)d, 0.0);
        }
        dval[jd * (size_t)d + i] += val[e];
        continue;
      }
      size_t g = 0;
      while (g < groups.size() && !(groups[g].x == x && groups[g].kid == kid[e])) ++g;
      if (g == groups.size()) {
        if (i != 0 || (int)groups.size() == dq::XR_RMAX) return;  // a diagonal that row 0 lacks, or too many
        groups.push_back({x, kid[e], (int)g, 0, std::fabs(val[e])});
      }
      if (std::fabs(val[e]) != groups[g].mag || groups[g].count != i) return;  // (count == i: exactly one entry per row so far)
      groups[g].count++;
      if (val[e] < 0.0) { sgn[i] |= 1u << g; neg = true; }
    }
  for (const Grp& g : groups)
    if (g.count != d || g.mag == 0.0) return;
  for (size_t a = 0; a < groups.size(); ++a)  // one diagonal per offset and source; symmetric signs (H_k symmetric: a sanity check)
    for (int i = 0; i < d; ++i)
      if (((sgn[i] >> a) & 1u) != ((sgn[i ^ groups[a].x] >> a) & 1u)) return;
  tab.R = (int)groups.size();
  tab.nd = (int)dsrc.size();
  for (int j = 0; j < tab.R; ++j) { tab.xoff[j] = groups[j].x; tab.kid[j] = groups[j].kid; tab.cmag[j] = groups[j].mag; }
  for (int j = 0; j < tab.nd; ++j) tab.dkid[j] = dsrc[j];
  c->xr_sgn.ensure(sizeof(unsigned) * (size_t)d);
  c->xr_dval.ensure(sizeof(double) * std::max<size_t>(dval.size(), 1));
  HIPCK(hipMemcpyAsync(c->xr_sgn.p, sgn.data(), sizeof(unsigned) * (size_t)d, hipMemcpyHostToDevice, c->st));
  if (!dval.empty())
    HIPCK(hipMemcpyAsync(c->xr_dval.p, dval.data(), sizeof(double) * dval.size(), hipMemcpyHostToDevice, c->st));
  HIPCK(hipStreamSynchronize(c->st));
  tab.sgn = neg ? (const unsigned*)c->xr_sgn.p : nullptr;
  tab.dval = (const double*)c->xr_dval.p;
  dq::xr_parity_signs(tab, neg ? sgn.data() : nullptr, d);
  dq::xr_diag_strings(tab, dval.empty() ? nullptr : dval.data(), d);
  c->xr_tab = tab;
  c->xr_ok = true;
  c->tab_sgn_h = sgn;
  c->tab_dval_h = dval;
  {   // the drift's diagonal alone (no diagonal control): its range, for the centred frame of the state-column kernels
    bool only_drift = !dval.empty();
    int jd0 = -1;
    for (int jd = 0; jd < tab.nd; ++jd) {
      if (tab.dkid[jd] >= 0) only_drift = false;
      else jd0 = jd;
    }
    c->xs_drift_range_ok = false;
    if (only_drift && jd0 >= 0 && tab.nd == 1) {
      double mn = dval[0], mx = dval[0];
      for (int i = 1; i < d; ++i) {
        mn = std::min(mn, dval[(size_t)jd0 * d + i]);
        mx = std::max(mx, dval[(size_t)jd0 * d + i]);
      }
      c->xs_drift_mid = 0.5 * (mx + mn);
      c->xs_drift_half = 0.5 * (mx - mn);
      c->xs_drift_range_ok = true;
    }
  }
}

// Complex Hermitian H0 / H_k that are sums of Pauli strings with X and Y factors (XOR-regular pattern, every XOR-diagonal of one
// magnitude and either real, +-c, or imaginary, +-i c; a real diagonal from the drift alone): table for the zxr_* kernels.
// The nonzeros are collected on the device (count, fill) and analysed on the host like find_xor_regular does for real ones.
template <class T>
void find_zxor_regular(Ctx<T>* c) {
  c->zxr_ok = false;
  typedef TT T2;
  if (!c->hermitian || c->real_h || c->K <= 0) return;   // (complex64 engines too: the state-column form takes the table)
  const int d = c->d, K = c->K, NS = K + 1;
  if (d < 256 || (d & (d - 1)) != 0) return;   // (any power of two: dimensions beyond the column kernels' go to the matrix-free passes)
  DevBuf cnt_dev, base_dev, col_dev, re_dev, im_dev;
  cnt_dev.ensure(sizeof(int) * (size_t)NS * d);
  HIPCK(dq::zsp_count_launch((const T2*)c->H0.p, d, 1, (int*)cnt_dev.p, c->st));
  HIPCK(dq::zsp_count_launch((const T2*)c->Hs.p, d, K, (int*)cnt_dev.p + d, c->st));
  std::vector<int> cnt((size_t)NS * d);
  HIPCK(hipMemcpyAsync(cnt.data(), cnt_dev.p, sizeof(int) * cnt.size(), hipMemcpyDeviceToHost, c->st));
  HIPCK(hipStreamSynchronize(c->st));
  std::vector<long> base((size_t)NS * d);
  long nnz = 0;
  for (size_t q = 0; q < cnt.size(); ++q) { base[q] = nnz; nnz += cnt[q]; }
  if (nnz == 0 || nnz > (long)(dq::XR_RMAX + dq::XR_DMAX) * d) return;  // up to 32 XOR-diagonals + 16 offset-0 ones
  base_dev.ensure(sizeof(long) * base.size());
  col_dev.ensure(sizeof(int) * (size_t)nnz);
  re_dev.ensure(sizeof(double) * (size_t)nnz);
  im_dev.ensure(sizeof(double) * (size_t)nnz);
  HIPCK(hipMemcpyAsync(base_dev.p, base.data(), sizeof(long) * base.size(), hipMemcpyHostToDevice, c->st));
  HIPCK(dq::zsp_fill_launch((const T2*)c->H0.p, d, 1, (const long*)base_dev.p, (int*)col_dev.p, (double*)re_dev.p,
                            (double*)im_dev.p, c->st));
  HIPCK(dq::zsp_fill_launch((const T2*)c->Hs.p, d, K, (const long*)base_dev.p + d, (int*)col_dev.p, (double*)re_dev.p,
                            (double*)im_dev.p, c->st));
  std::vector<int> col((size_t)nnz);
  std::vector<double> re((size_t)nnz), im((size_t)nnz);
  HIPCK(hipMemcpyAsync(col.data(), col_dev.p, sizeof(int) * (size_t)nnz, hipMemcpyDeviceToHost, c->st));
  HIPCK(hipMemcpyAsync(re.data(), re_dev.p, sizeof(double) * (size_t)nnz, hipMemcpyDeviceToHost, c->st));
  HIPCK(hipMemcpyAsync(im.data(), im_dev.p, sizeof(double) * (size_t)nnz, hipMemcpyDeviceToHost, c->st));
  HIPCK(hipStreamSynchronize(c->st));
  struct Grp { int x, kid, phase; long count; double mag; };
  std::vector<Grp> groups;
  std::vector<unsigned> sgn((size_t)d, 0u);
  std::vector<int> dsrc;      // sources with (real) diagonal entries: the drift, Z-type controls
  std::vector<double> dval;
  bool have_imag = false;
  for (int sidx = 0; sidx < NS; ++sidx)      // source 0 = drift (kid -1), source k + 1 = control k
    for (int i = 0; i < d; ++i) {
      const long e0 = base[(size_t)sidx * d + i];
      for (long e = e0; e < e0 + cnt[(size_t)sidx * d + i]; ++e) {
        const int x = col[e] ^ i, kid = sidx - 1;
        if (x == 0) {
          if (im[e] != 0.0) return;  // (Hermitian: cannot happen beyond rounding; be strict)
          size_t jd = 0;
          while (jd < dsrc.size() && dsrc[jd] != kid) ++jd;
          if (jd == dsrc.size()) {
            if ((int)dsrc.size() == dq::XR_DMAX) return;
            dsrc.push_back(kid);
            dval.resize(dsrc.size() * (size_t)d, 0.0);
          }
          dval[jd * (size_t)d + i] += re[e];
          continue;
        }
        if (re[e] != 0.0 && im[e] != 0.0) return;  // phases +-1 / +-i only
        const int ph = im[e] != 0.0 ? 1 : 0;
        const double v = ph ? im[e] : re[e];
        size_t g = 0;
        while (g < groups.size() && !(groups[g].x == x && groups[g].kid == kid)) ++g;
        if (g == groups.size()) {
          if (i != 0 || (int)groups.size() == dq::XR_RMAX) return;
          groups.push_back({x, kid, ph, 0, std::fabs(v)});
        }
        if (ph != groups[g].phase || std::fabs(v) != groups[g].mag || groups[g].count != i) return;
        groups[g].count++;
        if (v < 0.0) sgn[i] |= 1u << g;
        have_imag = have_imag || ph;
      }
    }
  if (groups.empty() || !have_imag) return;
  for (size_t a = 0; a < groups.size(); ++a) {
    if (groups[a].count != d) return;
    for (int i = 0; i < d; ++i) {  // Hermitian: real diagonals have symmetric signs, imaginary ones antisymmetric
      const unsigned s0 = (sgn[i] >> a) & 1u, s1 = (sgn[i ^ groups[a].x] >> a) & 1u;
      if (groups[a].phase ? s0 == s1 : s0 != s1) return;
    }
  }
  dq::XrTab tab;
  memset(&tab, 0, sizeof(tab));
  tab.R = (int)groups.size();
  tab.nd = (int)dsrc.size();
  for (int j = 0; j < tab.nd; ++j) tab.dkid[j] = dsrc[j];
  for (int j = 0; j < tab.R; ++j) {
    tab.xoff[j] = groups[j].x; tab.kid[j] = groups[j].kid; tab.cmag[j] = groups[j].mag; tab.phase[j] = groups[j].phase;
  }
  c->xr_sgn.ensure(sizeof(unsigned) * (size_t)d);
  c->xr_dval.ensure(sizeof(double) * std::max<size_t>(dval.size(), 1));
  HIPCK(hipMemcpyAsync(c->xr_sgn.p, sgn.data(), sizeof(unsigned) * (size_t)d, hipMemcpyHostToDevice, c->st));
  if (!dval.empty())
    HIPCK(hipMemcpyAsync(c->xr_dval.p, dval.data(), sizeof(double) * dval.size(), hipMemcpyHostToDevice, c->st));
  HIPCK(hipStreamSynchronize(c->st));
  tab.sgn = (const unsigned*)c->xr_sgn.p;
  tab.dval = (const double*)c->xr_dval.p;
  dq::xr_parity_signs(tab, sgn.data(), d);
  dq::xr_diag_strings(tab, dval.empty() ? nullptr : dval.data(), d);
  c->zxr_tab = tab;
  c->zxr_ok = true;
  c->tab_sgn_h = sgn;
  c->tab_dval_h = dval;
  {   // the drift's diagonal alone (no diagonal control): its range, for the centred frame of the state-column kernels
    bool only_drift = !dval.empty();
    int jd0 = -1;
    for (int jd = 0; jd < tab.nd; ++jd) {
      if (tab.dkid[jd] >= 0) only_drift = false;
      else jd0 = jd;
    }
    c->xs_drift_range_ok = false;
    if (only_drift && jd0 >= 0 && tab.nd == 1) {
      double mn = dval[0], mx = dval[0];
      for (int i = 1; i < d; ++i) {
        mn = std::min(mn, dval[(size_t)jd0 * d + i]);
        mx = std::max(mx, dval[(size_t)jd0 * d + i]);
      }
      c->xs_drift_mid = 0.5 * (mx + mn);
      c->xs_drift_half = 0.5 * (mx - mn);
      c->xs_drift_range_ok = true;
    }
  }
}

// CSR of the Hamiltonians' nonzeros (real path), if they are sparse enough to pay: at most sparse_fill nonzeros per row
// summed over the sources (X-type Paulis: one per control). Two tiers: controls AND drift sparse (sparse_x: the drift's
// entries join the CSR with kid = -1, so X_t itself is sparse and the polynomial runs on sparse x dense products), or the
// controls alone (dense drift: scatter assembly + the last reverse-mode level as row dot products). Otherwise dense passes.
// Row i of the CSR holds its drift entries first (if any), then those of control 0, control 1, ...
template <class T>
void build_sparse_controls(Ctx<T>* c) {
  c->sparse_h = false;
  c->sparse_x = false;
  c->xr_ok = false;
  c->sp_nnz = 0;
  const int d = c->d, K = c->K;
  if (!c->real_h || K <= 0) return;
  const int NS = K + 1;  // source 0 = drift, source k + 1 = control k
  DevBuf cnt_dev, base_dev;
  cnt_dev.ensure(sizeof(int) * (size_t)NS * d);
  HIPCK(dq::rsp_count_launch<T>((const T*)c->H0r.p, d, 1, (int*)cnt_dev.p, c->st));
  HIPCK(dq::rsp_count_launch<T>((const T*)c->Hsr.p, d, K, (int*)cnt_dev.p + d, c->st));
  std::vector<int> cnt((size_t)NS * d);
  HIPCK(hipMemcpyAsync(cnt.data(), cnt_dev.p, sizeof(int) * cnt.size(), hipMemcpyDeviceToHost, c->st));
  HIPCK(hipStreamSynchronize(c->st));
  long nnz_c = 0, nnz_0 = 0;
  for (int i = 0; i < d; ++i) nnz_0 += cnt[i];
  for (size_t q = d; q < cnt.size(); ++q) nnz_c += cnt[q];
  const double cap = std::min(c->sparse_fill * d, 0.5 * (double)d * d);  // (and never more than half full: tiny dimensions)
  if ((double)nnz_c > cap) return;
  const bool with_drift = (double)(nnz_c + nnz_0) <= cap;
  const int s0 = with_drift ? 0 : 1;  // first source in the CSR
  const long nnz = nnz_c + (with_drift ? nnz_0 : 0);
  std::vector<long> rowptr((size_t)d + 1, 0), base((size_t)NS * d, 0), kptr((size_t)K + 1, 0), kent((size_t)std::max<long>(nnz_c, 1));
  long at = 0;
  for (int i = 0; i < d; ++i) {
    rowptr[i] = at;
    for (int sidx = s0; sidx < NS; ++sidx) {
      base[(size_t)sidx * d + i] = at;
      at += cnt[(size_t)sidx * d + i];
    }
  }
  rowptr[d] = at;
  for (int k = 0; k < K; ++k) {
    long nk = 0;
    for (int i = 0; i < d; ++i) nk += cnt[(size_t)(k + 1) * d + i];
    kptr[k + 1] = kptr[k] + nk;
  }
  {  // the control entries grouped by control (fixed order: by row, then column), for the per-control reduction
    std::vector<long> fill(kptr.begin(), kptr.end() - 1);
    for (int i = 0; i < d; ++i)
      for (int k = 0; k < K; ++k)
        for (int q = 0; q < cnt[(size_t)(k + 1) * d + i]; ++q) kent[fill[k]++] = base[(size_t)(k + 1) * d + i] + q;
  }
  const size_t ne = (size_t)std::max<long>(nnz, 1);
  c->sp_rowptr.ensure(sizeof(long) * rowptr.size());
  c->sp_kptr.ensure(sizeof(long) * kptr.size());
  c->sp_kent.ensure(sizeof(long) * kent.size());
  c->sp_col.ensure(sizeof(int) * ne);
  c->sp_kid.ensure(sizeof(int) * ne);
  c->sp_val.ensure(sizeof(double) * ne);
  base_dev.ensure(sizeof(long) * base.size());
  HIPCK(hipMemcpyAsync(c->sp_rowptr.p, rowptr.data(), sizeof(long) * rowptr.size(), hipMemcpyHostToDevice, c->st));
  HIPCK(hipMemcpyAsync(c->sp_kptr.p, kptr.data(), sizeof(long) * kptr.size(), hipMemcpyHostToDevice, c->st));
  HIPCK(hipMemcpyAsync(c->sp_kent.p, kent.data(), sizeof(long) * kent.size(), hipMemcpyHostToDevice, c->st));
  HIPCK(hipMemcpyAsync(base_dev.p, base.data(), sizeof(long) * base.size(), hipMemcpyHostToDevice, c->st));
  if (with_drift)
    HIPCK(dq::rsp_fill_launch<T>((const T*)c->H0r.p, d, 1, (const long*)base_dev.p, (int*)c->sp_col.p, (int*)c->sp_kid.p,
                                 (double*)c->sp_val.p, -1, c->st));
  HIPCK(dq::rsp_fill_launch<T>((const T*)c->Hsr.p, d, K, (const long*)base_dev.p + d, (int*)c->sp_col.p, (int*)c->sp_kid.p,
                               (double*)c->sp_val.p, 0, c->st));
  HIPCK(hipStreamSynchronize(c->st));  // the host vectors and the temporaries go out of scope
  c->sp_nnz = nnz;
  c->sp_nnz_c = nnz_c;
  c->sparse_h = true;
  c->sparse_x = with_drift;
  if (with_drift) find_xor_regular(c, rowptr, nnz);
}

template <class T>
void do_set_hamiltonians(Ctx<T>* c, const void* H0, const void* Hs) {
  typedef TT T2;
  check_ctx(c);
  if (!H0 || (c->K > 0 && !Hs)) fail(DQ_ERR_INVALID, "null Hamiltonian pointer");
  c->have_H = false;
  c->slice_valid = false;  // whatever a slice forward call cached belongs to the old Hamiltonians
  for (auto& e : c->pcache) e.valid = false;
  c->H0.ensure(c->mat_bytes());
  c->Hs.ensure(std::max<size_t>(c->K, 1) * c->mat_bytes());
  HIPCK(hipMemcpyAsync(c->H0.p, H0, c->mat_bytes(), hipMemcpyDefault, c->st));
  if (c->K > 0) HIPCK(hipMemcpyAsync(c->Hs.p, Hs, (size_t)c->K * c->mat_bytes(), hipMemcpyDefault, c->st));
  HIPCK(hipStreamSynchronize(c->st));
  c->nH0 = norm1_dev(c, (const T2*)c->H0.p);
  double defect = herm_defect_dev(c, (const T2*)c->H0.p);
  double scale = std::max(1.0, c->nH0);
  for (int k = 0; k < c->K; ++k) {
    const T2* Hk = (const T2*)c->Hs.p + (size_t)k * c->nn();
    c->nHk[k] = norm1_dev(c, Hk);
    defect = std::max(defect, herm_defect_dev(c, Hk));
    scale = std::max(scale, c->nHk[k]);
  }
  c->hermitian = defect <= c->herm_tol * scale;
  // real-symmetric fast path: all imaginary parts exactly zero -> keep (symmetrised) real copies
  c->real_h = false;
  if (c->hermitian) {
    c->H0r.ensure(c->nn() * sizeof(T));
    c->Hsr.ensure(std::max<size_t>(c->K, 1) * c->nn() * sizeof(T));
    c->scratch.ensure(sizeof(double) * (size_t)std::max(c->d, 1));
    HIPCK(hipMemsetAsync(c->scratch.p, 0, sizeof(double), c->st));
    HIPCK(dq::real_part_launch<T>((const T2*)c->H0.p, (T*)c->H0r.p, c->d, 1, c->scratch.d(), c->st));
    if (c->K > 0) HIPCK(dq::real_part_launch<T>((const T2*)c->Hs.p, (T*)c->Hsr.p, c->d, c->K, c->scratch.d(), c->st));
    double im = 1.0;
    HIPCK(hipMemcpyAsync(&im, c->scratch.p, sizeof(double), hipMemcpyDeviceToHost, c->st));
    HIPCK(hipStreamSynchronize(c->st));
    c->real_h = im == 0.0;
    if (!c->real_h) { c->H0r.release(); c->Hsr.release(); }
  }
  build_sparse_controls(c);
  find_zxor_regular(c);
  c->mf_dirty = true;
  c->have_H = true;
}

template <class T>
void do_set_observable(Ctx<T>* c, const void* M) {
  typedef TT T2;
  check_ctx(c);
  if (!M) fail(DQ_ERR_INVALID, "null observable pointer");
  c->have_M = false;
  c->M.ensure(c->mat_bytes());
  HIPCK(hipMemcpyAsync(c->M.p, M, c->mat_bytes(), hipMemcpyDefault, c->st));
  HIPCK(hipStreamSynchronize(c->st));
  const double nm = norm1_dev(c, (const T2*)c->M.p);
  if (herm_defect_dev(c, (const T2*)c->M.p) > c->herm_tol * std::max(1.0, nm))
    fail(DQ_ERR_INVALID, "observable must be Hermitian");
  {  // diagonal observables (Ising cost functions): the state-column form applies them elementwise
    c->m_diag.ensure(sizeof(double) * (size_t)c->d);
    c->scratch.ensure(sizeof(double) * (size_t)std::max(c->d, 1));
    HIPCK(hipMemsetAsync(c->scratch.p, 0, sizeof(double), c->st));
    HIPCK(dq::xs_offdiag_launch((const T2*)c->M.p, c->d, c->scratch.d(), c->m_diag.d(), c->st));
    double off = 1.0;
    HIPCK(hipMemcpyAsync(&off, c->scratch.p, sizeof(double), hipMemcpyDeviceToHost, c->st));
    HIPCK(hipStreamSynchronize(c->st));
    c->m_is_diag = off == 0.0;
  }
  c->mf_dirty = true;
  c->have_M = true;
}

template <class T>
void do_zgemm(Ctx<T>* c, int opA, int opB, int M, int N, int K, double alpha, const void* A, long lda, long strideA,
              const void* B, long ldb, long strideB, double beta, void* C, long ldc, long strideC, int batch, int herm) {
  typedef TT T2;
  check_ctx(c);
  if (M < 0 || N < 0 || K < 0 || batch < 0) fail(DQ_ERR_INVALID, "negative dimension");
  if (herm < -1 || herm > 1) fail(DQ_ERR_INVALID, "herm must be 0, +1 (Hermitian product) or -1 (anti-Hermitian)");
  if (herm && M != N) fail(DQ_ERR_INVALID, "the (anti-)Hermitian output form needs M == N");
  if (opA < 0 || opA > 4 || (opB & ~1))
    fail(DQ_ERR_INVALID, "opA must be 0 (N), 1 (conjugate transpose) or 2 / 3 / 4 (N, known complex symmetric / Hermitian / "
                         "anti-Hermitian); opB 0 or 1");
  if (opA >= 2 && M != K) fail(DQ_ERR_INVALID, "a symmetry hint needs a square A");
  if (M == 0 || N == 0 || batch == 0) return;
  if (!A || !B || !C) fail(DQ_ERR_INVALID, "null matrix pointer");
  const long ar = opA == 1 ? K : M, ac = opA == 1 ? M : K;  // stored shape of A
  const long br = opB ? N : K, bc = opB ? K : N;
  if (lda < ac || ldb < bc || ldc < N) fail(DQ_ERR_INVALID, "leading dimension smaller than the row length");
  auto span = [&](long rows, long ld, long cols, long stride) -> size_t {
    if (rows == 0 || cols == 0) return 0;
    return (size_t)((batch - 1) * stride + (rows - 1) * ld + cols) * sizeof(T2);
  };
  if (batch > 1 && strideC < (M - 1) * ldc + N) fail(DQ_ERR_INVALID, "overlapping output batch stride");
  const size_t ba = span(ar, lda, ac, strideA), bb = span(br, ldb, bc, strideB), bcz = span(M, ldc, N, strideC);
  const T2* dA = ba ? in_dev(c, A, ba, c->tmp_in[0]) : (const T2*)c->tmp_in[0].p;
  const T2* dB = bb ? in_dev(c, B, bb, c->tmp_in[1]) : (const T2*)c->tmp_in[1].p;
  T2* dC;
  const bool c_dev = is_device_ptr(C);
  if (c_dev) dC = (T2*)C;
  else {
    c->tmp_in[2].ensure(bcz);
    stage_h2d(c, c->tmp_in[2].p, C, bcz);  // beta*C and untouched gaps
    dC = (T2*)c->tmp_in[2].p;
  }
  c->zgemm(opA, opB, M, N, K, alpha, dA, lda, strideA, dB, ldb, strideB, beta, dC, ldc, strideC, batch, herm);
  if (!c_dev) out_copy(c, C, dC, bcz);
  HIPCK(hipStreamSynchronize(c->st));
}

template <class T>
void do_rgemm(Ctx<T>* c, int opA, int opB, int M, int N, int K, double alpha, const void* A, long lda, long strideA,
              const void* B, long ldb, long strideB, double beta, void* C, long ldc, long strideC, int batch, int sym) {
  check_ctx(c);
  if (M < 0 || N < 0 || K < 0 || batch < 0) fail(DQ_ERR_INVALID, "negative dimension");
  if ((opA | opB) & ~1) fail(DQ_ERR_INVALID, "op must be 0 (N) or 1 (transpose)");
  if (sym && M != N) fail(DQ_ERR_INVALID, "symmetric-output form needs M == N");
  if (sym == 2 && K != M) fail(DQ_ERR_INVALID, "the two-product form A B + B A needs square operands");
  if (M == 0 || N == 0 || batch == 0) return;
  if (!A || !B || !C) fail(DQ_ERR_INVALID, "null matrix pointer");
  if (sym) { opA = 1; opB = 0; }
  const long ar = opA == 1 ? K : M, ac = opA == 1 ? M : K;  // stored shape of A
  const long br = opB ? N : K, bc = opB ? K : N;
  if (lda < ac || ldb < bc || ldc < N) fail(DQ_ERR_INVALID, "leading dimension smaller than the row length");
  auto span = [&](long rows, long ld, long cols, long stride) -> size_t {
    if (rows == 0 || cols == 0) return 0;
    return (size_t)((batch - 1) * stride + (rows - 1) * ld + cols) * sizeof(T);
  };
  if (batch > 1 && strideC < (M - 1) * ldc + N) fail(DQ_ERR_INVALID, "overlapping output batch stride");
  const size_t ba = span(ar, lda, ac, strideA), bb = span(br, ldb, bc, strideB), bcz = span(M, ldc, N, strideC);
  const T* dA = ba ? (const T*)in_dev(c, A, ba, c->tmp_in[0]) : (const T*)c->tmp_in[0].p;
  const T* dB = bb ? (const T*)in_dev(c, B, bb, c->tmp_in[1]) : (const T*)c->tmp_in[1].p;
  T* dC;
  const bool c_dev = is_device_ptr(C);
  if (c_dev) dC = (T*)C;
  else {
    c->tmp_in[2].ensure(bcz);
    stage_h2d(c, c->tmp_in[2].p, C, bcz);
    dC = (T*)c->tmp_in[2].p;
  }
  dq::RgemmArgs<T> g;
  g.A = dA; g.B = dB; g.C = dC;
  g.M = M; g.N = N; g.K = K;
  g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.sA = strideA; g.sB = strideB; g.sC = strideC;
  g.batch = batch;
  g.alpha = alpha; g.beta = beta;
  g.opA = opA; g.opB = opB;
  g.sym = sym ? 1 : 0;
  g.two = sym == 2 ? 1 : 0;
  g.glds_ok = c->rgemm_glds;
  c->rgemm_workspace(g);
  const int cfg = c->rgemm_cfg, ncu = c->n_cu;
  c->timed(sym ? DQ_PROF_RSYM : DQ_PROF_RGEMM, sym ? c->rsym_flops(M, K, batch) * (sym == 2 ? 2.0 : 1.0) : 2.0 * M * N * (double)K * batch,
           sizeof(T) * ((double)M * K + (double)K * N + (double)M * N) * batch,
           [&] { return dq::rgemm_launch<T>(g, c->st, cfg, ncu); });
  if (!c_dev) out_copy(c, C, dC, bcz);
  HIPCK(hipStreamSynchronize(c->st));
}

template <class T>
double stacked_theta(Ctx<T>* c, const TT* P, int batch) {
  double th = 0.0;
  for (int z = 0; z < batch; ++z) th = std::max(th, norm1_dev(c, P + (size_t)z * c->nn()));
  c->last_theta = th;
  return th;
}

template <class T>
void do_expm(Ctx<T>* c, const void* A, void* U, int batch) {
  typedef TT T2;
  check_ctx(c);
  if (batch < 0) fail(DQ_ERR_INVALID, "negative batch");
  if (batch == 0) return;
  if (!A || !U) fail(DQ_ERR_INVALID, "null matrix pointer");
  T2* P1 = c->mat(0, batch);
  HIPCK(hipMemcpyAsync(P1, A, (size_t)batch * c->mat_bytes(), hipMemcpyDefault, c->st));
  const ExpmPlan p = choose_plan(stacked_theta(c, P1, batch), c->expm_tol, c->force_degree, c->force_s);
  const T2* Ud = expm_forward(c, p, batch);
  out_copy(c, U, Ud, (size_t)batch * c->mat_bytes());
  HIPCK(hipStreamSynchronize(c->st));
}

template <class T>
void do_expm_vjp(Ctx<T>* c, const void* A, const void* Ubar, void* Abar, int batch) {
  typedef TT T2;
  check_ctx(c);
  if (batch < 0) fail(DQ_ERR_INVALID, "negative batch");
  if (batch == 0) return;
  if (!A || !Ubar || !Abar) fail(DQ_ERR_INVALID, "null matrix pointer");
  T2* P1 = c->mat(0, batch);
  HIPCK(hipMemcpyAsync(P1, A, (size_t)batch * c->mat_bytes(), hipMemcpyDefault, c->st));
  const ExpmPlan p = choose_plan(stacked_theta(c, P1, batch), c->expm_tol, c->force_degree, c->force_s);
  (void)expm_forward(c, p, batch);
  T2* G = ubar_slot(c, p, batch);
  HIPCK(hipMemcpyAsync(G, Ubar, (size_t)batch * c->mat_bytes(), hipMemcpyDefault, c->st));
  const T2* Ab = expm_backward(c, p, batch);
  out_copy(c, Abar, Ab, (size_t)batch * c->mat_bytes());
  HIPCK(hipStreamSynchronize(c->st));
}

template <class T>
void do_set_option(Ctx<T>* c, const std::string& s, double v) {
  if (s == "checkpoint_bytes") c->checkpoint_bytes = v;
  else if (s == "chunk_bytes") { if (v < 0) fail(DQ_ERR_INVALID, "chunk_bytes must be >= 0 (0 = default)"); c->chunk_bytes = v; }
  else if (s == "expm_tol") { if (!(v > 0)) fail(DQ_ERR_INVALID, "expm_tol must be > 0"); c->expm_tol = v; }
  else if (s == "force_degree") c->force_degree = (int)v;
  else if (s == "force_squarings") c->force_s = (int)v;
  else if (s == "force_tile") {
    const int t = (int)v;
    if (t != 0 && t != 16 && t != 32 && t != 64 && t != 128) fail(DQ_ERR_INVALID, "force_tile must be 0, 16, 32, 64 or 128");
    c->force_tile = t;
  } else if (s == "compose") {  // -1 auto (batch > dim), 0 never (no composed route at all: explicit tree, state-column product, wide-batch gradient), 1 always
                                // (for a Pauli-sum generator the product comes from state-column segments unless state_columns_compose = 0)
    c->compose_mode = (int)v < 0 ? -1 : ((int)v ? 1 : 0);
  } else if (s == "real_path") {  // -1 auto (real H, no forced complex tape), 0 never, q in 1..6 forced
    const int q = (int)v;
    if (q < -1 || q > RQ_MAX) fail(DQ_ERR_INVALID, "real_path must be -1 (auto), 0 (off) or a polynomial size 1..6");
    c->real_mode = q;
  } else if (s == "cache_bytes") {  // real path: HBM for propagator intermediates kept from the forward pass (< 0 off, 0 auto)
    c->cache_bytes = v;
  } else if (s == "small_dim") {  // auto mode: dimensions up to this take the operator form for any batch width
    c->small_dim = (int)v;
  } else if (s == "time_parallel") {  // operator backward sweep of a chunk: -1 auto (>= 4 steps), 0 sequential, 1 always
    c->time_parallel = (int)v < 0 ? -1 : ((int)v ? 1 : 0);
  } else if (s == "plan_steps") {  // checkpoint stride as if the chain had this many steps (0 = this call's length)
    if (v < 0) fail(DQ_ERR_INVALID, "plan_steps must be >= 0");
    c->plan_steps = (long)v;
  } else if (s == "rgemm_glds") {  // 1: direct global -> LDS staging in the real GEMM where shapes/alignment allow
    c->rgemm_glds = v != 0 ? 1 : 0;
  } else if (s == "real_min_dim") {  // auto mode: below this dimension the chain is launch-bound and the complex tape
    c->real_min_dim = (int)v;         // (fewer launches) is kept
  } else if (s == "rgemm_cfg") {
    const int q = (int)v;
    if (q != 0 && q != 1 && q != 2 && q != 3 && q != 32 && q != 64)
      fail(DQ_ERR_INVALID, "rgemm_cfg must be 0, 1, 2, 3, 32 or 64");
    c->rgemm_cfg = q;
  } else if (s == "sparse_controls") {  // real path: -1 auto (use the controls' sparsity when they are sparse), 0 never
    c->sparse_mode = (int)v != 0 ? -1 : 0;
  } else if (s == "fuse_levels") {  // sparse generator: gradient levels inside the reverse passes (1) or a separate kernel (0)
    c->fuse_levels = v != 0 ? 1 : 0;
  } else if (s == "sparse_generator") {  // drift and controls sparse: -1 auto (dim >= 64 nonzeros-per-row), 0 off, 1 whenever possible
    c->sparse_gen_mode = v < 0 ? -1 : (v > 0 ? 1 : 0);
  } else if (s == "transpose_fallback") {  // transposing pre-pass for big complex products whose operands have no k-major form
    c->transpose_fallback = v != 0 ? 1 : 0;
  } else if (s == "xor_regular") {  // column-form kernels for XOR-regular sparse generators: -1 auto (default), 0 off
    c->xor_mode = v != 0 ? -1 : 0;
  } else if (s == "state_columns") {  // state-column form (statecol.hip): -1 auto (default), 0 off, 1 whenever the kernels exist
    c->xs_mode = v < 0 ? -1 : (v > 0 ? 1 : 0);
  } else if (s == "state_columns_variant") {  // bit 0: 1024 threads x 4 rows at dim 4096 (default 512 x 8); bit 1: prefetch of the next diagonal's partners in the forward kernel
    c->xs_variant = (int)v & 3;
  } else if (s == "state_columns_center") {  // 1 (default): the kernels run on the diagonal minus the midpoint of its range (lower step bound), 0: as given
    c->xs_center = v != 0 ? 1 : 0;
  } else if (s == "state_columns_adjoint") {  // -1 auto / 1: quadrature form of the adjoint sweep when its error bound allows; 0: exact polynomial derivative
    c->xs_adjoint = v < 0 ? -1 : (v > 0 ? 1 : 0);
  } else if (s == "state_columns_wide_adjoint") {  // energy gradients of batches wider than dim on dim columns: -1 by cost (default), 0 off, 1 on
    c->xs_wide_mode = v < 0 ? -1 : (v > 0 ? 1 : 0);
  } else if (s == "state_columns_compose") {  // products of a chain's propagators (composed forward pass, time slices) from the state-column kernel: -1 auto, 0 off
    c->xs_compose_mode = v < 0 ? -1 : (v > 4096 ? 4096 : (int)v);   // (> 0: that many segments)
  } else if (s == "state_columns_chain_tol") {  // bound of the step polynomial's error summed over the chain (per step: min(expm_tol, this / steps))
    if (!(v > 0)) fail(DQ_ERR_INVALID, "state_columns_chain_tol must be > 0");
    c->xs_chain_tol = v;
  } else if (s == "state_columns_polynomial") {  // 1 (default): Chebyshev-economized step polynomial; 0: Taylor
    c->xs_economize = v != 0 ? 1 : 0;
  } else if (s == "matrix_free") {  // -1 auto (Pauli-sum generators beyond the state-column kernels' dimensions), 0 never, 1 whenever the table is a string sum
    c->mf_mode = v < 0 ? -1 : (v > 0 ? 1 : 0);
  } else if (s == "matrix_free_checkpoint_bytes") {  // checkpoint budget of the routed passes (0: 70 % of the free HBM)
    if (v < 0) fail(DQ_ERR_INVALID, "matrix_free_checkpoint_bytes must be >= 0");
    c->mf_checkpoint_bytes = v;
  } else if (s == "state_columns_workgroups") {  // 0 = as many as fit the chip; > 0 caps the launch (tests: several columns per workgroup)
    if (v < 0 || v > 65535) fail(DQ_ERR_INVALID, "state_columns_workgroups must be 0..65535");
    c->xs_workgroups = (int)v;
  } else if (s == "state_columns_part_bytes") {  // scratch budget of the adjoint sweep's partial sums (the chain is swept in segments that fit)
    if (!(v >= 1)) fail(DQ_ERR_INVALID, "state_columns_part_bytes must be >= 1");
    c->xs_part_bytes = v;
  } else if (s == "state_columns_quad_tol") {  // bound of the rule's error per step, relative to ||dA_k|| ||lambda|| ||psi||
    if (!(v > 0)) fail(DQ_ERR_INVALID, "state_columns_quad_tol must be > 0");
    c->xs_quad_tol = v;
  } else if (s == "state_columns_batch_ratio") {  // auto: taken while batch <= ratio x dim
    if (!(v > 0)) fail(DQ_ERR_INVALID, "state_columns_batch_ratio must be > 0");
    c->xs_batch_ratio = v;
  } else if (s == "slab_spmm") {  // sparse x dense passes in XCD-slab form (1, default) or one wave per whole row (0)
    c->slab_spmm = v != 0 ? 1 : 0;
  } else if (s == "skew_trick") {
    c->skew_trick = v != 0 ? 1 : 0;
  } else if (s == "zgemm_alg") {
    const int a = (int)v;
    if (a < 0 || a > 2) fail(DQ_ERR_INVALID, "zgemm_alg must be 0 (4M), 1 (3M) or 2 (3M + linear-form kernel)");
    c->zgemm_alg = a;
  } else fail(DQ_ERR_INVALID, "unknown option '" + s + "'");
}

template <class T>
double do_get_info(Ctx<T>* c, const std::string& s) {
  if (s == "degree") return c->last_degree;
  if (s == "squarings") return c->last_squarings;
  if (s == "products") return c->last_products;
  if (s == "time_chunk") return c->last_chunk;
  if (s == "checkpoint_stride") return c->last_cstride;
  if (s == "theta") return c->last_theta;
  if (s == "hermitian") return c->hermitian ? 1.0 : 0.0;
  if (s == "zgemm_alg") return sizeof(T) == 8 ? c->zgemm_alg : 1;
  if (s == "composed") return c->last_composed;
  if (s == "operator_backward") return c->last_opback;
  if (s == "time_parallel") return c->last_tpar;
  if (s == "cached_steps") return c->last_cached;
  if (s == "slice_skew") return skew_w_form(c) ? 1.0 : 0.0;
  if (s == "sparse_controls") return c->sparse_on() ? 1.0 : 0.0;
  if (s == "control_nonzeros") return c->sparse_h ? (double)c->sp_nnz_c : -1.0;
  if (s == "sparse_generator") return c->last_spx;
  if (s == "xor_regular") return c->last_xr;
  if (s == "state_columns") return c->last_xs;
  if (s == "matrix_free") return c->last_mf;
  if (s.rfind("matrix_free_", 0) == 0 && c->mf) {   // "matrix_free_passes", "matrix_free_dma_passes", ...: the routed context's own info
    double v = 0;
    if (dq_sctx_get_info(c->mf, s.c_str() + 12, &v) == DQ_OK) return v;
    fail(DQ_ERR_INVALID, std::string("matrix-free routing: ") + dq_sctx_last_error());
  }
  if (s == "state_columns_quad_nodes") return c->last_xs_nodes;
  if (s == "state_columns_centered") return c->last_xs_centered;
  if (s == "state_columns_theta_uncentered") return c->last_xs_theta_loose;
  if (s == "state_columns_taylor_degree") return c->last_xs_taylor_degree;
  if (s == "state_columns_compose_segments") return c->last_xs_segments_fwd;
  if (s == "state_columns_wide_adjoint") return c->last_xs_wide;
  if (s == "state_columns_segments") return c->last_xs_segments;
  if (s == "xor_regular_found") return c->xr_ok ? (double)c->xr_tab.R : (c->zxr_ok ? (double)c->zxr_tab.R : -1.0);
  if (s == "sparse_drift") return c->sparse_h && c->sparse_x ? 1.0 : 0.0;
  if (s == "real_hamiltonians") return c->real_h ? 1.0 : 0.0;
  if (s == "real_path") return c->last_real_q;
  if (s == "norm_H0") return c->nH0;
  if (s == "dim") return c->d;
  if (s == "n_controls") return c->K;
  if (s == "dtype") return c->dtype;
  fail(DQ_ERR_INVALID, "unknown info '" + s + "'");
}

#undef TT

template <class F>
int guarded(F&& f) {
  try {
    f();
    return DQ_OK;
  } catch (const DqError& e) {
    g_err = e.msg;
    return e.code;
  } catch (const std::exception& e) {
    g_err = e.what();
    return DQ_ERR_HIP;
  }
}

// run f on the typed context behind the type-erased handle
template <class F>
int with_ctx(dq_ctx* c, F&& f) {
  return guarded([&] {
    if (!c) fail(DQ_ERR_INVALID, "null context");
    if (c->dtype == DQ_C128) f(static_cast<Ctx<double>*>(c));
    else f(static_cast<Ctx<float>*>(c));
  });
}

}  // namespace

extern "C" {

const char* dq_version(void) { return "diffqc_hip 0.4 (gfx950; complex128 + complex64; real-symmetric fast path; time-sliced operator form)"; }
const char* dq_last_error(void) { return g_err.c_str(); }

int dq_step_polynomial(double theta, double tol, int economize, int* degree, double* factors) {
  return guarded([&] {
    if (!degree || !factors) fail(DQ_ERR_INVALID, "null output");
    if (!(theta >= 0.0) || !(tol > 0.0)) fail(DQ_ERR_INVALID, "theta must be >= 0 and tol > 0");
    dq::XsPoly p;
    *degree = dq::xs_poly_make(theta, tol, economize, &p);
    for (int m = 0; m <= dq::XS_LMAX; ++m) factors[m] = p.g[m];
  });
}

int dq_quadrature_rule(double theta, double tol, int degree, int economize, int basis, int* nodes, double* w0, double* w1, double* c1,
                       double* c2) {
  return guarded([&] {
    if (!nodes || !w0 || !w1 || !c1 || !c2) fail(DQ_ERR_INVALID, "null output");
    if (!(theta >= 0.0) || !(tol > 0.0) || degree < 1 || degree > dq::XS_LMAX) fail(DQ_ERR_INVALID, "theta >= 0, tol > 0, 1 <= degree <= 24");
    dq::XsPoly p;
    dq::xs_poly_of_degree(theta, degree, economize, &p);
    dq::XsQuad qd;
    *nodes = 0;
    if (!dq::xs_quad_make(theta, tol, 2, degree, economize, p, &qd, basis ? 1 : 0)) return;
    *nodes = qd.ni;
    *w0 = qd.w0;
    for (int m = 0; m <= dq::XS_LMAX; ++m) {
      w1[m] = qd.w1[m];
      for (int i = 0; i < 2; ++i) {
        c1[2 * m + i] = qd.c1[m][i];
        c2[2 * m + i] = qd.c2[m][i];
      }
    }
  });
}

int dq_device_count(int* n) {
  return guarded([&] {
    if (!n) fail(DQ_ERR_INVALID, "null output");
    int k = 0;
    hipError_t e = hipGetDeviceCount(&k);
    if (e != hipSuccess) { (void)hipGetLastError(); k = 0; }
    *n = k;
  });
}

int dq_create_ex(dq_ctx** out, int dim, int n_controls, int device_id, int dtype) {
  return guarded([&] {
    if (!out) fail(DQ_ERR_INVALID, "null output");
    *out = nullptr;
    if (dim < 1 || dim > (1 << 15)) fail(DQ_ERR_INVALID, "dim must be in [1, 32768]");
    if (n_controls < 0 || n_controls > 4096) fail(DQ_ERR_INVALID, "n_controls must be in [0, 4096]");
    if (dtype != DQ_C128 && dtype != DQ_C64) fail(DQ_ERR_INVALID, "dtype must be DQ_C128 or DQ_C64");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
      (void)hipGetLastError();
      fail(DQ_ERR_NO_DEVICE, "no HIP device available (this library has no CPU path)");
    }
    if (device_id < 0 || device_id >= ndev) fail(DQ_ERR_INVALID, "device_id out of range");
    HIPCK(hipSetDevice(device_id));
    auto init = [&](auto* c) {
      c->d = dim;
      c->K = n_controls;
      c->dev = device_id;
      int ncu = 0;
      if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device_id) == hipSuccess && ncu > 0) c->n_cu = ncu;
      else (void)hipGetLastError();
      int lds = 0;
      if (hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, device_id) == hipSuccess && lds > 0) c->max_lds = (size_t)lds;
      else (void)hipGetLastError();
      c->nHk.assign(n_controls, 0.0);
      *out = c;
    };
    if (dtype == DQ_C128) init(new Ctx<double>());
    else init(new Ctx<float>());
  });
}

int dq_create(dq_ctx** out, int dim, int n_controls, int device_id) {
  return dq_create_ex(out, dim, n_controls, device_id, DQ_C128);
}

int dq_destroy(dq_ctx* c) {
  if (!c) return DQ_OK;
  return with_ctx(c, [&](auto* t) {
    (void)hipSetDevice(t->dev);
    (void)hipStreamSynchronize(t->st);
    delete t;
  });
}

int dq_set_stream(dq_ctx* c, void* s) {
  return with_ctx(c, [&](auto* t) {
    check_ctx(t);
    HIPCK(hipStreamSynchronize(t->st));
    t->st = (hipStream_t)s;
  });
}

int dq_synchronize(dq_ctx* c) {
  return with_ctx(c, [&](auto* t) {
    check_ctx(t);
    HIPCK(hipStreamSynchronize(t->st));
  });
}

int dq_set_hamiltonians(dq_ctx* c, const void* H0, const void* Hs) {
  return with_ctx(c, [&](auto* t) { do_set_hamiltonians(t, H0, Hs); });
}

int dq_set_observable(dq_ctx* c, const void* M) {
  return with_ctx(c, [&](auto* t) { do_set_observable(t, M); });
}

int dq_set_option(dq_ctx* c, const char* name, double v) {
  return with_ctx(c, [&](auto* t) {
    check_ctx(t);
    if (!name) fail(DQ_ERR_INVALID, "null option name");
    do_set_option(t, std::string(name), v);
  });
}

int dq_get_info(dq_ctx* c, const char* name, double* v) {
  return with_ctx(c, [&](auto* t) {
    check_ctx(t);
    if (!name || !v) fail(DQ_ERR_INVALID, "null argument");
    *v = do_get_info(t, std::string(name));
  });
}

int dq_propagate(dq_ctx* c, const double* u, int n_steps, double dt, const void* psi_in, void* psi_out, int batch) {
  return with_ctx(c, [&](auto* t) { do_propagate(t, u, n_steps, dt, psi_in, psi_out, batch); });
}

int dq_energy(dq_ctx* c, const void* psi, int batch, double* energies, double* total) {
  return with_ctx(c, [&](auto* t) {
    typedef typename std::remove_pointer<decltype(t)>::type::T2 T2;
    check_ctx(t);
    if (!t->have_M) fail(DQ_ERR_STATE, "dq_set_observable has not been called");
    if (batch <= 0 || !psi) fail(DQ_ERR_INVALID, "need batch > 0 and a state pointer");
    const size_t sb = (size_t)t->d * batch * sizeof(T2);
    const T2* p = in_dev(t, psi, sb, t->tmp_in[0]);
    t->lam[0].ensure(sb);
    do_energy(t, p, batch, (T2*)t->lam[0].p, energies, total);
    HIPCK(hipStreamSynchronize(t->st));
  });
}

int dq_energy_grad(dq_ctx* c, const double* u, int n_steps, double dt, const void* psi0, int batch, double* loss,
                   double* grad_u, void* psi_final, double* energies) {
  return with_ctx(c, [&](auto* t) { do_energy_grad(t, u, n_steps, dt, psi0, batch, loss, grad_u, psi_final, energies); });
}

int dq_gram(dq_ctx* c, const void* psi0, int batch, void* R_out) {
  return with_ctx(c, [&](auto* t) { do_gram(t, psi0, batch, R_out); });
}

int dq_energy_grad_columns(dq_ctx* c, const double* u, int n_steps, double dt, const void* R, const void* P, int col0, int n_cols,
                           double* loss, double* grad_u) {
  return with_ctx(c, [&](auto* t) { do_energy_grad_columns(t, u, n_steps, dt, R, P, col0, n_cols, loss, grad_u); });
}

int dq_fidelity_grad(dq_ctx* c, const double* u, int n_steps, double dt, const void* psi0, const void* targets, int batch,
                     double* loss, double* grad_u, void* psi_final, double* fidelities) {
  return with_ctx(c, [&](auto* t) {
    if (!targets) fail(DQ_ERR_INVALID, "null target states");
    do_energy_grad(t, u, n_steps, dt, psi0, batch, loss, grad_u, psi_final, fidelities, targets);
  });
}

int dq_mc_shift_grad(dq_ctx* c, const double* u_a, int n_a, double dt_a, const double* u_b, int n_b, double dt_b,
                     const void* psi0, int batch, double r, double* ps, void* phi_out) {
  return with_ctx(c, [&](auto* t) { do_mc_shift(t, u_a, n_a, dt_a, u_b, n_b, dt_b, psi0, batch, r, ps, phi_out); });
}

int dq_slice_forward(dq_ctx* c, const double* u, int n_steps, double dt, void* P_out) {
  return with_ctx(c, [&](auto* t) { do_slice_forward(t, u, n_steps, dt, P_out); });
}

int dq_slice_terminal(dq_ctx* c, const void* Q, const void* psi0, int batch, const void* targets, double* loss,
                      double* energies, void* psi_final, void* W_out) {
  return with_ctx(c, [&](auto* t) { do_slice_terminal(t, Q, psi0, batch, targets, loss, energies, psi_final, W_out); });
}

int dq_slice_backward(dq_ctx* c, const double* u, int n_steps, double dt, const void* W_end, double* grad_u,
                      void* W_start) {
  return with_ctx(c, [&](auto* t) { do_slice_backward(t, u, n_steps, dt, W_end, grad_u, W_start); });
}

int dq_build_propagators(dq_ctx* c, const double* u, int n_steps, double dt, void* U_out) {
  return with_ctx(c, [&](auto* t) { do_build_propagators(t, u, n_steps, dt, U_out); });
}

int dq_propagator_vjp(dq_ctx* c, const double* u, int n_steps, double dt, const void* G, double* grad_u) {
  return with_ctx(c, [&](auto* t) { do_propagator_vjp(t, u, n_steps, dt, G, grad_u); });
}

int dq_zgemm(dq_ctx* c, int opA, int opB, int M, int N, int K, double alpha, const void* A, long lda, long strideA,
             const void* B, long ldb, long strideB, double beta, void* C, long ldc, long strideC, int batch) {
  return dq_zgemm_ex(c, opA, opB, M, N, K, alpha, A, lda, strideA, B, ldb, strideB, beta, C, ldc, strideC, batch, 0);
}

int dq_zgemm_ex(dq_ctx* c, int opA, int opB, int M, int N, int K, double alpha, const void* A, long lda, long strideA,
                const void* B, long ldb, long strideB, double beta, void* C, long ldc, long strideC, int batch, int herm) {
  return with_ctx(c, [&](auto* t) {
    do_zgemm(t, opA, opB, M, N, K, alpha, A, lda, strideA, B, ldb, strideB, beta, C, ldc, strideC, batch, herm);
  });
}

int dq_rgemm(dq_ctx* c, int opA, int opB, int M, int N, int K, double alpha, const void* A, long lda, long strideA,
             const void* B, long ldb, long strideB, double beta, void* C, long ldc, long strideC, int batch, int sym) {
  return with_ctx(c, [&](auto* t) {
    do_rgemm(t, opA, opB, M, N, K, alpha, A, lda, strideA, B, ldb, strideB, beta, C, ldc, strideC, batch, sym);
  });
}

int dq_expm(dq_ctx* c, const void* A, void* U, int batch) {
  return with_ctx(c, [&](auto* t) { do_expm(t, A, U, batch); });
}

int dq_expm_vjp(dq_ctx* c, const void* A, const void* Ubar, void* Abar, int batch) {
  return with_ctx(c, [&](auto* t) { do_expm_vjp(t, A, Ubar, Abar, batch); });
}

int dq_profile_enable(dq_ctx* c, int on) {
  return with_ctx(c, [&](auto* t) {
    check_ctx(t);
    t->prof_collect();
    t->prof_on = on != 0;
  });
}

int dq_profile_reset(dq_ctx* c) {
  return with_ctx(c, [&](auto* t) {
    check_ctx(t);
    t->prof_collect();
    memset(&t->prof_acc, 0, sizeof(t->prof_acc));
  });
}

int dq_profile_get(dq_ctx* c, dq_profile* out) {
  return with_ctx(c, [&](auto* t) {
    check_ctx(t);
    if (!out) fail(DQ_ERR_INVALID, "null output");
    t->prof_collect();
    *out = t->prof_acc;
  });
}

}  // extern "C"
