// Internal declarations shared by the HIP translation units of libdiffqc_hip.so.
// Not part of the C-ABI (that is include/diffqc_hip.h).
// T is the real scalar (double for complex128, float for complex64): the reference's
// `typedef double Scalar` / commented `typedef float Scalar` (diffqc.cc:13-19).
#pragma once
#include <hip/hip_runtime.h>

namespace dq {

template <class T>
struct Cx;
template <>
struct Cx<double> {
  typedef double2 type;
};
template <>
struct Cx<float> {
  typedef float2 type;
};

// linear index over the tiles on and above the diagonal (row-major over the triangle: row i holds nt - i tiles);
// shared by the symmetric-output forms of the real and the complex GEMM
__device__ __forceinline__ void tri_tile(int pid, int nt, int& tile_m, int& tile_n) {
  const double w = 2.0 * nt + 1.0;
  int i = (int)((w - sqrt(w * w - 8.0 * pid)) * 0.5);
  if (i < 0) i = 0;
  while (i > 0 && i * nt - i * (i - 1) / 2 > pid) --i;  // first tile of row i <= pid
  while ((i + 1) * nt - (i + 1) * i / 2 <= pid) ++i;
  tile_m = i;
  tile_n = i + (pid - (i * nt - i * (i - 1) / 2));
}

template <class T>
struct ZgemmArgs {
  typedef typename Cx<T>::type T2;
  const T2* A;
  const T2* B;
  T2* C;
  int M, N, K;          // op(A) is M x K, op(B) is K x N, C is M x N
  long lda, ldb, ldc;   // leading dimensions of the STORED matrices, in complex elements
  long sA, sB, sC;      // batch strides, in complex elements (0 = shared operand)
  int batch;
  double alpha, beta;
  int opA, opB;         // 0 = as stored, 1 = conjugate transpose
  // hintA (with opA == 0): what the caller knows about the stored A -- 0 nothing, 1 complex symmetric (A^T = A),
  // 2 Hermitian, 3 anti-Hermitian; 5 (engine-internal): the buffer already holds A^T ([K][M], lda = M).  A hint lets the launcher read A k-major (A = A^T, A^H, -A^H) and take the
  // linear-form kernel (zgemm_lin.hip); results do not depend on it for operands that have the stated property.
  int hintA = 0;
  // herm: 0 = general output; +1 / -1 = the caller asserts that the product is Hermitian / anti-Hermitian (e.g. X^H W X
  // for (anti-)Hermitian W): only the tiles on and above the diagonal are computed, the mirror image is +-conj (M == N)
  int herm;
  T2* ws;            // optional workspace for the K-split tail tiles of the herm form (may be null)
  size_t ws_elems;   // its capacity in complex elements
  int n_full, split_f;  // filled in by zgemm_launch
};

// n_cu: compute units of the device (tail splitting of the herm form)
// alg: 0 = 4M, 1 = 3M, 2 = 3M with the linear-form kernel (zgemm_lin.hip) wherever operand forms and shapes allow (fp64)
template <class T>
hipError_t zgemm_launch(const ZgemmArgs<T>& g, hipStream_t st, int force_tile, int alg, int n_cu = 256);
// linear-form complex GEMM (zgemm_lin.hip): op(A) = A^T / A^H read k-major, B as stored; whole 128-tiles only
bool zgemm_lin_eligible(const ZgemmArgs<double>& g);
hipError_t zgemm_lin_launch(ZgemmArgs<double> g, int conj_a, hipStream_t st, int n_cu);
int zgemm_pick_tile(int M, int N, int batch);

// ---- elementwise / reduction kernels (elementwise.hip) ------------------------------
// A[z] = -i*dt*(H0 + sum_k u[z*K + k] * Hs[k])   (diffqc.cc:192-197, sim_plain.py:137-143)
template <class T>
hipError_t assemble_launch(typename Cx<T>::type* A, const typename Cx<T>::type* H0, const typename Cx<T>::type* Hs,
                           const double* u, int K, long n_elem, int batch, double dt, hipStream_t st);

// out[z] = (accumulate ? out[z] : 0) + cI * I + sum_{j<n_in} c[j] * X_j[z]      (d x d, batched)
template <class T>
struct LincombArgs {
  typedef typename Cx<T>::type T2;
  T2* out;
  const T2* x[6];
  double c[6];
  int n_in;
  double cI;
  int d;
  long stride;  // batch stride in complex elements (same for all operands)
  int batch;
  int accumulate;
};
template <class T>
hipError_t lincomb_launch(const LincombArgs<T>& a, hipStream_t st);

// out[z] = (accumulate ? out[z] : 0) + a * X[z] + b * X[z]^H      (d x d, batched, out != X)
template <class T>
hipError_t tcombine_launch(typename Cx<T>::type* out, const typename Cx<T>::type* X, int d, double a, double b,
                           int accumulate, long stride, int batch, hipStream_t st);

// out[z] (cols x rows, ldo) = in[z]^T (in: rows x cols, ldi), conjugated if conj != 0
template <class T>
hipError_t ztranspose_launch(typename Cx<T>::type* out, long ldo, long so, const typename Cx<T>::type* in, long ldi, long si,
                             int rows, int cols, int conj, int batch, hipStream_t st);

// part[z][blk][k] = sum over the block's elements of Im(conj(Abar[z][e]) * Hs[k][e]) (accumulated in fp64);
// grad[z][k] = scale * sum_blk part   (deterministic two-stage reduction)
template <class T>
hipError_t grad_contract_launch(const typename Cx<T>::type* Abar, long strideA, const typename Cx<T>::type* Hs, int K,
                                long n_elem, int batch, double scale, double* part, double* grad, hipStream_t st);
int grad_contract_blocks(long n_elem);
// grad[z][k] = scale * sum over blocks of part[z][blk][k], fixed summation order
hipError_t grad_finish_launch(const double* part, int nblk, int K, int batch, double scale, double* grad, hipStream_t st);

// e[b] = scale * sum_i Re(conj(X[i][b]) * Y[i][b]);  X, Y are d x B (ld in complex elements); fp64 accumulation
// `part` is scratch of coldot_chunks(d) * B doubles (two-stage, fixed order)
template <class T>
hipError_t coldot_launch(const typename Cx<T>::type* X, long ldx, const typename Cx<T>::type* Y, long ldy, int d, int B,
                         double scale, double* e, double* part, hipStream_t st);
int coldot_chunks(int d);
// fidelity objective: c[b] = <phi_b|psi_b>, fid[b] = |c[b]|^2, lam = scale * phi .* c (lam may be null)
// part: coldot_chunks(d) * B * 2 doubles; cvals: 2 B doubles
template <class T>
hipError_t fidelity_costate_launch(const typename Cx<T>::type* phi, const typename Cx<T>::type* psi, int d, int B,
                                   double scale, double* part, double* cvals, double* fid,
                                   typename Cx<T>::type* lam, hipStream_t st);
// shifted states (phi +- i r H_k phi) / sqrt(1 + r^2) of the reference's Monte-Carlo estimator (sim_plain.py:194-199);
// HP is [K][d][B], out is [d][B * K * 2] (column (b K + k) 2 + s, s = 0 plus / 1 minus)
template <class T>
hipError_t mc_shift_launch(const typename Cx<T>::type* phi, const typename Cx<T>::type* HP, int d, int B, int K, double r,
                           typename Cx<T>::type* out, hipStream_t st);
// out[0] = sum_b e[b]  (single block, deterministic)
hipError_t sum_launch(const double* e, int n, double* out, hipStream_t st);
// colsum[j] = sum_i |X[i][j]|  (for the 1-norm), d x d; part: colabs_scratch_doubles(d) doubles (two-stage, fixed order)
template <class T>
hipError_t colabs_launch(const typename Cx<T>::type* X, int d, double* colsum, double* part, hipStream_t st);
int colabs_scratch_doubles(int d);
// max |X - X^H| over the matrix -> out[0] (Hermiticity check)
template <class T>
hipError_t herm_defect_launch(const typename Cx<T>::type* X, int d, double* out, hipStream_t st);

// ---- real-symmetric fast path (rgemm.hip, realpath.hip) -------------------------------------
// Real GEMM C = alpha op(A) op(B) + beta C (op: 0 = N, 1 = T), row-major, strides in elements. sym != 0: the
// product is known to be symmetric (commuting symmetric operands): only tiles on/above the diagonal are computed,
// the mirror image is written too (M == N required; opA/opB ignored; beta != 0 needs a symmetric C on entry).
template <class T>
struct RgemmArgs {
  const T* A;
  const T* B;
  T* C;
  int M, N, K;
  long lda, ldb, ldc;
  long sA, sB, sC;
  int batch;
  double alpha, beta;
  int opA, opB;
  int sym;
  int two;          // with sym: C = alpha (A B + B A) + beta C for symmetric A, B (K == M == N)
  T* ws;            // optional workspace of the symmetric form (split-K partial tiles of the tail), may be null
  size_t ws_elems;  // its capacity in elements
  int n_full, split_f;  // filled in by rgemm_launch
  int glds_ok;          // allow the LDS-DMA staged variant when shapes and alignment permit (1 = yes)
};
// cfg: tile configuration (0 = auto); n_cu: compute units of the device
template <class T>
hipError_t rgemm_launch(const RgemmArgs<T>& g, hipStream_t st, int cfg, int n_cu);

// Hr = (Re H + Re H^T)/2 for `batch` stacked d x d matrices; imag_max[0] = max |Im H| (must be zeroed by the caller)
template <class T>
hipError_t real_part_launch(const typename Cx<T>::type* H, T* Hr, int d, int batch, double* imag_max, hipStream_t st);
// X[z] = -dt (H0r + sum_k u[z*K + k] Hsr[k])          (diffqc.cc:192-197 without the factor i)
template <class T>
hipError_t rassemble_launch(T* X, const T* H0r, const T* Hsr, const double* u, int K, long n_elem, int batch, double dt,
                            hipStream_t st);
// out_o[z] = (accumulate ? out_o[z] : 0) + cI[o] I + sum_j c[o][j] x_j[z]   for o < n_out (inputs read once)
template <class T>
struct RLincombArgs {
  T* out[2];
  const T* x[8];
  double c[2][8];
  double cI[2];
  int n_in, n_out;
  int d;
  long stride;
  int batch;
  int accumulate;
  int pack;  // 1: the two results are written as ONE interleaved complex matrix out[0] = (result 0) + i (result 1)
             // (U = C + i S without storing C and S; n_out must be 2, accumulate 0, out[1] unused)
};
template <class T>
hipError_t rlincomb_launch(const RLincombArgs<T>& a, hipStream_t st);
// U = C + i S   (n real elements each)
template <class T>
hipError_t pack_u_launch(typename Cx<T>::type* U, const T* C, const T* S, long n, hipStream_t st);
// Cbar = (Re G + Re G^T)/2, Sbar = (Im G + Im G^T)/2
template <class T>
hipError_t split_g_launch(const typename Cx<T>::type* G, T* Cbar, T* Sbar, int d, long stride, int batch, hipStream_t st);
// grad[z][k] = scale * sum_e Xbar[z][e] Hsr[k][e]  (two-stage, fixed order; part: batch * grad_contract_blocks * K)
template <class T>
hipError_t rgrad_contract_launch(const T* Xbar, long strideX, const T* Hsr, int K, long n_elem, int batch, double scale,
                                 double* part, double* grad, hipStream_t st);

// ---- sparse control Hamiltonians of the real path (realpath.hip) ----
// cnt[k*d + i] = nonzeros in row i of Hsr[k];  fill: the entries of (k, i), in column order, at base[k*d + i]
template <class T>
hipError_t rsp_count_launch(const T* Hsr, int d, int K, int* cnt, hipStream_t st);
template <class T>
hipError_t rsp_fill_launch(const T* Hsr, int d, int K, const long* base, int* col, int* kid, double* val, int kid0,
                           hipStream_t st);
// X[z][i][col] += -dt u[z*K + kid] val over the CSR (rowptr: d + 1 entries), z < batch
template <class T>
hipError_t rsp_add_launch(T* X, long strideX, const long* rowptr, const int* col, const int* kid, const double* val,
                          const double* u, int K, int d, int batch, double dt, hipStream_t st);
// grad[z][k] = scale * sum_{e in control k} val[e] (<Sb[z][i_e,:], P[z][j_e,:]> + 2 <Yb[z][i_e,:], X[z][j_e,:]>)
// (nzval: batch * nnz doubles of scratch; kptr/kent: the entries grouped by control)
template <class T>
hipError_t rsp_contract_launch(const T* Sb, const T* P, const T* Yb, const T* X, long stride, int d, const long* rowptr,
                               const int* col, const double* val, long nnz, const long* kptr, const long* kent, int K,
                               int batch, double scale, double* nzval, double* grad, hipStream_t st);

// ---- sparse generator (drift and controls sparse; realpath.hip) ----
// out[z][i][:] = ca * Add[z][i][:] + sum_{e in row i} w_e V[z][col_e][:],  w_e = -dt (kid_e < 0 ? 1 : u[z*K + kid_e]) val_e
// (Add may be null). slab: allow the XCD-slab form (16-B column blocks in 8 slabs) when d and the alignments permit
template <class T>
hipError_t rsp_mm_launch(T* out, const T* V, const T* Add, double ca, long stride, int d, const long* rowptr, const int* col,
                         const int* kid, const double* val, const double* u, int K, int batch, double dt, int slab,
                         hipStream_t st);
// nzval[z][e] = val_e (direct_e + sum_l <G[l][z][i,:], Xp[l][z][j,:]>) for the control entries (0 for drift entries);
// direct_e = dc D[z][i][j] + sum_{e' in row i} w_e' G2[z][col_e'][j] if D is given
template <class T>
hipError_t rsp_levels_launch(const T* const* G, const T* const* Xp, int nl, const T* D, const T* G2, double dc, long stride,
                             int d, const long* rowptr, const int* col, const int* kid, const double* val, const double* u,
                             int K, int batch, double dt, long nnz, double* nzval, hipStream_t st);
// rsp_mm_launch with the gradient level fused in: nzacc[z][e] (+)= <V[z][col_e,:], Xm[z][i,:]> for every entry e of row i
// (out may be null: dot products only). part: batch * 8 * nnz doubles of scratch for the XCD-slab form, or null
template <class T>
hipError_t rsp_mm_dot_launch(T* out, const T* V, const T* Add, double ca, const T* Xm, long stride, int d, const long* rowptr,
                             const int* col, const int* kid, const double* val, const double* u, int K, int batch, double dt,
                             long nnz, double* nzacc, int accumulate, double* part, hipStream_t st);
// nzval[z][e] = val_e (nzacc[z][e] + dc D[z][j][i] + sum_{p in row j} w_p G2[z][col_p][i]) for control entries (i, j), else 0
template <class T>
hipError_t rsp_finish_launch(const double* nzacc, const T* D, const T* G2, double dc, long stride, int d, const long* rowptr,
                             const int* col, const int* kid, const double* val, const double* u, int K, int batch, double dt,
                             long nnz, double* nzval, hipStream_t st);
hipError_t rsp_reduce_launch(const double* nzval, long nnz, const long* kptr, const long* kent, int K, int batch, double scale,
                             double* grad, hipStream_t st);

// ---- xorexp.hip: sparse generator with an XOR-regular pattern: besides offset-0 diagonals, row i holds exactly the entries
// (i, i ^ xoff[j]), j < R, each XOR-diagonal from one control and of constant magnitude ----
constexpr int XR_RMAX = 32;  // off-diagonal XOR-diagonals (table capacity; sign words are 32 bits)
constexpr int XR_RSMALL = 16; // reverse kernels come with 16 or 32 per-diagonal accumulators (register budget)
constexpr int XR_DMAX = 16;  // diagonals at offset 0 (one per source matrix that has diagonal entries; any per-row values)
constexpr int XR_SLOTS = XR_RMAX + XR_DMAX;
constexpr int XR_QMAX = 6;   // cos / sin polynomial: powers X_1 .. X_{2q+1}
struct XrTab {
  int R, nd;
  int xoff[XR_RMAX];
  int kid[XR_RMAX];      // the control the diagonal belongs to (-1: the drift)
  double cmag[XR_RMAX];  // |value| along the diagonal
  int phase[XR_RMAX];    // complex tables (zxr_* kernels): 0 = entries +-cmag, 1 = entries +-i cmag; real tables: 0
  int dkid[XR_DMAX];     // source of an offset-0 diagonal: control index, or -1 for the drift
  const unsigned* sgn;   // [d] bit j set: entry (i, i ^ xoff[j]) is -cmag[j]; null when no entry is negative
  const double* dval;    // [nd][d] values of the offset-0 diagonals
  // The signs as parities (every diagonal one Pauli string): bit j of sgn[i] = bit j of s0mask ^ parity(i & zmask[j]).  parity = 1 when
  // that holds for every diagonal (xr_parity_signs); the state-column kernels take their signs from these two (no sign words).
  unsigned zmask[XR_RMAX];
  unsigned s0mask;
  int parity;
  // The offset-0 diagonals of CONTROLS as Z-strings: dstring = 1 when every source jd with dkid[jd] >= 0 is one Pauli-Z string,
  // dval[jd][i] = dmag[jd] (-1)^parity(i & dzmask[jd]) (the reference's ZZ-edge controls, demo_maxcut.py:71-74): the adjoint kernels
  // then contract their per-row sums with signs made from the masks instead of reading nd x d table values per step (xr_diag_strings)
  unsigned dzmask[XR_DMAX];
  double dmag[XR_DMAX];
  int dstring;
};
// fills zmask / s0mask / parity from the host copy of the sign words (sgn_h may be null: no negative entry)
void xr_parity_signs(XrTab& tab, const unsigned* sgn_h, int d);
// fills dzmask / dmag / dstring from the host copy of the offset-0 diagonals ([nd][d])
void xr_diag_strings(XrTab& tab, const double* dval_h, int d);
struct XrPoly {
  int q;
  double cc[XR_QMAX + 1], sc[XR_QMAX + 1];  // cos X = sum cc_j X^2j, sin X = sum sc_j X^(2j+1)
};
bool xr_supported(int d, int q, int R, int nd, size_t elem);  // elem = sizeof(real type)
int xr_blocks(int d, int batch);  // workgroups per step (scratch of xr_reverse_launch: batch * xr_blocks(d, batch) * XR_SLOTS doubles)
hipError_t xr_forward_launch(double2* U, long stride, int d, const XrTab& tab, const double* u, int K, double dt,
                             const XrPoly& pol, int batch, hipStream_t st);
hipError_t xr_forward_launch(float2* U, long stride, int d, const XrTab& tab, const double* u, int K, double dt, const XrPoly& pol,
                             int batch, hipStream_t st);
hipError_t xr_reverse_launch(const double2* G, long stride, int d, const XrTab& tab, const double* u, int K, double dt,
                             const XrPoly& pol, int batch, double* part, const unsigned* zero_sgn, double* xpow, double* grad,
                             int accumulate, hipStream_t st);
hipError_t xr_reverse_launch(const float2* G, long stride, int d, const XrTab& tab, const double* u, int K, double dt,
                             const XrPoly& pol, int batch, double* part, const unsigned* zero_sgn, float* xpow, double* grad,
                             int accumulate, hipStream_t st);
size_t xr_pow_elems(int d, int q, int batch);
// Complex Hermitian XOR-regular generators (phases +-1 / +-i along the XOR-diagonals, real diagonal from the drift only):
// U = sum_{m <= L} A^m / m!, A = -i dt H, row by row in LDS; reverse mode from the rows of Ubar.  complex128, d = 256 .. 4096.
bool zxr_supported(int d, int L, int R, int nd);
hipError_t zsp_count_launch(const double2* H, int d, int K, int* cnt, hipStream_t st);  // cnt[k * d + i] = nonzeros of row i of H[k]
hipError_t zsp_fill_launch(const double2* H, int d, int K, const long* base, int* col, double* re, double* im, hipStream_t st);
hipError_t zsp_count_launch(const float2* H, int d, int K, int* cnt, hipStream_t st);
hipError_t zsp_fill_launch(const float2* H, int d, int K, const long* base, int* col, double* re, double* im, hipStream_t st);
hipError_t zxr_forward_launch(double2* U, long stride, int d, const XrTab& tab, const double* u, int K, double dt, int L, int batch,
                              hipStream_t st);
// xpow: batch * zxr_pow_elems(d, L, batch) double2 of scratch; part as for xr_reverse_launch
size_t zxr_pow_elems(int d, int L, int batch);
hipError_t zxr_reverse_launch(const double2* G, long stride, int d, const XrTab& tab, const double* u, int K, double dt, int L,
                              int batch, double* part, const unsigned* zero_sgn, double2* xpow, double* grad, int accumulate,
                              hipStream_t st);  // per step: scratch of the reverse kernel (xpow), in doubles


// ---- statecol.hip: whole chains on state columns held in LDS (XOR-regular generators; real or complex tables) ----
// States are stored column by column ([B][d]).  wt [S][R]: per-step weights of the XOR-diagonals; dgs: the rows' diagonal
// weights, one vector or one per step (xs_weights_launch).
struct XsOrder {                   // the order in which the sweeps walk the table (statecol.hip, XsCtl)
  unsigned char perm[XR_RMAX];     // sorted position -> index in the table
  int n0, n1;                      // LDS partners that keep / permute the thread's row list
  int nloc[8];                     // partners in the thread's own registers, by offset / threads
  int n0p1, n1p1, nlocp1[8];       // how many of each class have phase 1 (entries +-i c): every class is ordered phase 0 first
  unsigned char rsn[20];           // per group (class 0 / 1 by phase: 0..3; register class xh by phase: 4 + 2 (xh - 1) + phase): how many of
                                   // its diagonals, ordered LAST, carry a row factor of the sign (parity mask with bits at or above the thread count)
  unsigned long long gmask;        // bit g: loop g of the walk has at least one diagonal.  g = perm 4 + phase 2 + rs for the LDS classes,
                                   // 8 + (xh - 1) 4 + phase 2 + rs for the register classes: a signed complex table has 36 loops of which
                                   // a handful run, and an empty one is a bit test instead of its count arithmetic (round 5: the
                                   // headers of empty loops were a fifth of the X / Y-drive sweeps)
};
bool xs_supported(int d, int L, int R, int nd, size_t elem);  // elem = sizeof(real type)
// The step polynomial p(A) = sum_{m <= L} g[m] A^m / m! of the state-column kernels.  g = 1 is the Taylor polynomial; the economized
// form is the Chebyshev expansion of exp on the spectral interval i [-theta, theta] of the anti-Hermitian A,
//     exp(i theta x) = J_0(theta) + 2 sum_k i^k J_k(theta) T_k(x),       |x| <= 1,
// cut at degree L (error at most 2 sum_{k > L} |J_k(theta)|, against theta^(L+1) / (L+1)! for Taylor: a factor 2^L smaller, so one
// to four degrees fewer for the same tolerance) and rewritten in powers of A: the kernels keep their recurrence v_m = A v_{m-1} / m and
// accumulate g[m] v_m.
constexpr int XS_LMAX = 24;   // largest degree of the state-column kernels
struct XsPoly {
  double g[XS_LMAX + 1];
};
// time segments of a forward launch (xs_forward_segments_launch): cols = columns per segment (0: none), total = steps of the table
// phase: the launch's scalar factor (see xs_center_launch) -- seg_phase[2 g], [2 g + 1] for the columns of time segment g, or (pr, pi)
constexpr int XS_SEGMAX = 16;   // segments whose factors travel in the kernel arguments (more: a device table)
struct XsSeg {
  int cols, total;
  const double* seg_phase;
  double pr, pi;
  double ph[2 * XS_SEGMAX];   // cols > 0 and seg_phase == null: (re, im) of segment g at [2 g], [2 g + 1]
};
// the lowest degree whose error bound stays below tol for steps of size <= theta (0: none up to XS_LMAX) and its level factors;
// economize = 0: Taylor (g = 1).  Steps of theta > 2 keep Taylor (the power form of a Chebyshev sum loses digits as theta grows).
int xs_poly_make(double theta, double tol, int economize, XsPoly* p);
void xs_poly_of_degree(double theta, int L, int economize, XsPoly* p);   // the polynomial of a given degree
// Quadrature form of the adjoint sweep (xs_backward_q_kernel): a Gauss-Lobatto rule on [0, 1] with ni interior nodes s_i.  The node
// values exp(-s_i A) lambda are polynomials of the same degree on the shorter interval s_i theta, with their own level factors g_i[m]:
// c1[m][i] = g_i[m] s_i^m, c2[m][i] = -w_i g_i[m] s_i^m, w0 = minus the weight of the end point 0, w1[m] = -g[m] x the weight of the
// end point 1 (g: the factors of the whole step).
// basis = 1 (difference basis, ni = 2; the kernels that keep the rule's second vector in LDS as fp32): the SAME rule with its two node
// values replaced by u_1 = lambda(s_near) - a_1 lambda_t and u_2 = lambda(s_far) - a_0 lambda_t - beta u_1, a_i = g_i[0] / g[0] and
// beta such that u_2 has no first-order term: |u_1| ~ 0.28 theta |lambda|, |u_2| ~ 0.16 theta^2 |lambda|.  The tables keep their
// places and shapes -- phase 1 accumulates vector i with c1[m][i] (now: u_1, u_2; both rows m = 0 are 0 and c1[1][1] = 0),
// q_m = w1[m] lambda_t + c2[m][0] u_1 + c2[m][1] u_2 (+ w0 lambda_{t+1} at m = 0) -- so the kernel's arithmetic is the node form's.
constexpr int XS_NIMAX = 3;
struct XsQuad {
  int ni;
  int basis;
  double w0;
  double g[XS_LMAX + 1];    // the level factors of the whole step (XsPoly)
  double w1[XS_LMAX + 1];
  double c1[XS_LMAX + 1][XS_NIMAX], c2[XS_LMAX + 1][XS_NIMAX];
};
// the rule for a step of size theta (bound of ||A||): the fewest interior nodes whose error bound e (2 theta)^(2 ni + 2) stays below
// tol (relative to ||dA_k|| ||lambda|| ||psi||); false when no rule the kernels are built for reaches it (ni_max <= XS_NIMAX).
// L, economize, poly: the step polynomial the node tables are built for (xs_poly_make).
bool xs_quad_make(double theta, double tol, int ni_max, int L, int economize, const XsPoly& poly, XsQuad* qd, int basis = 0);
bool xs_quad_supported(const XrTab& tab, int ni, int d, int variant, size_t elem = sizeof(double));  // the table / rule / shape combinations xs_backward_q_kernel runs
bool xs_quad_basis(const XrTab& tab, int d, int variant, size_t elem);   // the basis the kernel of this table class / shape takes (true: the difference basis, XsQuad::basis = 1)
// variant: bit 0 = 1024 threads x 4 rows per workgroup at d = 4096 (default 512 x 8); bit 1 = prefetch of the next diagonal's partners
int xs_blocks(int d, size_t elem, int B, int n_cu, int variant);   // workgroups of a launch over B columns
int xs_waves(int d, int variant);
int xs_slots(const XrTab& tab);
int xs_rm(const XrTab& tab);       // per-thread accumulators of the adjoint kernel: 16, or 32 for more than 16 XOR-diagonals
size_t xs_park_elems(int d, int L, int nblk);                 // complex elements of the backward kernel's scratch
size_t xs_part_doubles(int d, int S, int nblk, const XrTab& tab, int variant);
int xs_lds_diagonals(int d, const XrTab& tab, int variant);   // XOR-diagonals whose partners are LDS reads (the others: registers)
// out[0] = max off-diagonal |entry| of M (zero it first), mdiag[i] = Re M[i][i];  lam = 2 diag(m) psi, e[b] = <psi_b|diag(m)|psi_b>
hipError_t xs_offdiag_launch(const double2* M, int d, double* out, double* mdiag, hipStream_t st);
hipError_t xs_offdiag_launch(const float2* M, int d, double* out, double* mdiag, hipStream_t st);
hipError_t xs_diag_costate_launch(const double2* psi, const double* mdiag, double2* lam, double* e, int d, int B, hipStream_t st);
hipError_t xs_diag_costate_launch(const float2* psi, const double* mdiag, float2* lam, double* e, int d, int B, hipStream_t st);
size_t xs_dgs_doubles(int d, int S, const XrTab& tab);       // doubles of the diagonal-weight table
// nt: threads per column of the kernels that will read the table (xs_threads(d, variant)): a vector is stored thread-major for them
hipError_t xs_weights_launch(const XrTab& tab, const double* u, int K, int S, int d, double dt, int nt, double* wt, double* dgs,
                             long* dgs_step, hipStream_t st);
int xs_threads(int d, int variant);
// Centred diagonal (round 5).  A step's generator A = -i (Dg + offdiagonal), Dg = the rows' diagonal weights (dgs).  Subtracting the
// midpoint c of Dg's range changes exp(A) by the scalar e^{-i c} only, and leaves a generator whose norm is half the RANGE of Dg plus
// the off-diagonal row sum -- for a drift like the reference demo's sum_e (I - Z_i Z_j) >= 0 (demo_maxcut.py:44-58) half of what the
// uncentred bound gives, one polynomial degree = one sweep in every pass.  xs_center_launch rewrites every row of the table in place
// as Dg - c and leaves c and the half-range per row (rows = S with diagonal controls, else 1); the kernels run in that frame and the
// launches apply the product of the steps' scalars once: the forward launch multiplies its outputs by (pr, pi) = e^{-i sum c}, the
// adjoint launch its inputs by e^{+i sum c} (psi and lambda alike: their pairings do not see the frame).
hipError_t xs_center_launch(double* dgs, int d, int rows, double* cen, double* half, hipStream_t st);
// every entry of one vector of the table minus c (the drift-only table: its midpoint is known on the host, dq_set_hamiltonians)
hipError_t xs_shift_launch(double* dgs, int d, double c, hipStream_t st);
hipError_t xs_forward_launch(const double2* in, double2* out, int d, const XrTab& tab, const double* wt, const double* dgs, long dgs_step,
                             int S, int B, int L, int nblk, int variant, const XsPoly* poly, double pr, double pi, hipStream_t st);
hipError_t xs_forward_launch(const float2* in, float2* out, int d, const XrTab& tab, const double* wt, const double* dgs, long dgs_step,
                             int S, int B, int L, int nblk, int variant, const XsPoly* poly, double pr, double pi, hipStream_t st);
// the propagators of n_seg consecutive time segments (columns of the identity through the forward kernel): out[(g d + i) d ..] =
// column i of segment g's propagator, i.e. block g is the propagator's transpose
hipError_t xs_forward_segments_launch(double2* out, int d, const XrTab& tab, const double* wt, const double* dgs, long dgs_step,
                                      int S_total, int seg_steps, int n_seg, int L, int nblk, int variant, const XsPoly* poly,
                                      const double* seg_phase, hipStream_t st);   // seg_phase: [n_seg][2] in HOST memory, or null
hipError_t xs_forward_segments_launch(float2* out, int d, const XrTab& tab, const double* wt, const double* dgs, long dgs_step,
                                      int S_total, int seg_steps, int n_seg, int L, int nblk, int variant, const XsPoly* poly,
                                      const double* seg_phase, hipStream_t st);
// psi: psi_S in, psi_0 out;  lam: lambda_S in, lambda_0 out;  grad[s][k] (+)= dL/du_k[s]
hipError_t xs_backward_launch(double2* psi, double2* lam, int d, const XrTab& tab, const double* wt, const double* dgs, long dgs_step,
                              int S, int B, int L, int nblk, double2* park, double* part, int K, double dt, double* grad,
                              int accumulate, int variant, const XsQuad* quad, const XsPoly* poly, double pr, double pi, hipStream_t st);  // quad != null: the quadrature form (park unused); poly = null: Taylor; (pr, pi): the factor the inputs take (centred diagonal)
hipError_t xs_backward_launch(float2* psi, float2* lam, int d, const XrTab& tab, const double* wt, const double* dgs, long dgs_step,
                              int S, int B, int L, int nblk, float2* park, double* part, int K, double dt, double* grad, int accumulate,
                              int variant, const XsQuad* quad, const XsPoly* poly, double pr, double pi, hipStream_t st);

}  // namespace dq
