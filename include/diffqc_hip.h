/* diffqc_hip.h -- C-ABI of libdiffqc_hip.so, the MI355X (gfx950) propagator engine.
 *
 * Drop-in boundary for the Schroedinger-propagator hot path of YilingQiao/diffquantum:
 * these entry points are what a binding of the reference's `diffqc` pybind11 module
 * (/root/reference/diffqc.cc:210-228) or of `SimulatorPlain.trotter`
 * (/root/reference/sim_plain.py:119-153) calls instead of Eigen's `dH.exp() * psi0`
 * (diffqc.cc:198) / `scipy.linalg.expm` + `np.matmul` (sim_plain.py:145-146).
 * INTEGRATION.md shows the reference-side stubs.
 *
 * Conventions
 *   - Complex data is interleaved (re, im) pairs, row-major: the memory of a numpy complex128 array /
 *     std::complex<double>[] (DQ_C128 contexts) or complex64 / std::complex<float>[] (DQ_C64 contexts);
 *     such buffers are passed as void*. Pulse tables, gradients, energies and the loss are always fp64.
 *   - A batch of states Psi is [dim][batch] (batch index contiguous): column b is trajectory b.
 *   - The pulse table u is [n_steps][n_controls] ("step-major"): row t holds u_k(t_0 + t*dt),
 *     the LEFT-endpoint sample the reference uses (diffqc.cc:195,199; sim_plain.py:140,150).
 *   - Every pointer argument may be host memory or device (HIP) memory; the library detects
 *     which (hipPointerGetAttributes) and only copies host buffers.
 *   - Every function returns a DQ_* status; dq_last_error() gives the message of the last
 *     failure on the calling thread. Nothing here falls back to a CPU implementation: without
 *     a gfx950 device dq_create fails with DQ_ERR_NO_DEVICE.
 */
#ifndef DIFFQC_HIP_H
#define DIFFQC_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define DQ_OK 0
#define DQ_ERR_INVALID 1      /* bad argument / shape (Python binding raises ValueError) */
#define DQ_ERR_STATE 2        /* call order: Hamiltonians / observable not set */
#define DQ_ERR_HIP 3          /* a HIP runtime call failed */
#define DQ_ERR_NO_DEVICE 4    /* no usable GPU */
#define DQ_ERR_UNSUPPORTED 5  /* e.g. adjoint of a non-Hermitian generator */
#define DQ_ERR_NOMEM 6

typedef struct dq_ctx dq_ctx;

/* Element type of every complex buffer of a context (the reference's `typedef double Scalar` /
 * commented `typedef float Scalar`, diffqc.cc:13-19). */
#define DQ_C128 0 /* (double re, double im) pairs: numpy complex128 / std::complex<double> */
#define DQ_C64 1  /* (float re, float im) pairs:   numpy complex64  / std::complex<float>  */

/* Kernel classes of the profile (dq_profile_get). */
#define DQ_PROF_ZGEMM 0     /* batched complex GEMM (f64 MFMA)            */
#define DQ_PROF_ASSEMBLE 1  /* A_t = -i dt (H0 + sum u_k H_k)             */
#define DQ_PROF_LINCOMB 2   /* polynomial linear combinations             */
#define DQ_PROF_GRADC 3     /* control-gradient contractions              */
#define DQ_PROF_OTHER 4     /* reductions, copies                         */
#define DQ_PROF_RGEMM 5     /* batched real GEMM of the real-symmetric fast path (f64 MFMA), full output   */
#define DQ_PROF_RSYM 6      /* the same kernel in its symmetric-output form (tiles on/above the diagonal)  */
#define DQ_PROF_ZHERM 7     /* complex GEMM in its (anti-)Hermitian-output form (tiles on/above the diagonal) */
#define DQ_PROF_SPMM 8      /* sparse generator x dense matrix passes (drift and controls sparse)                */
#define DQ_PROF_STATECOL_FWD 9  /* state-column form (statecol.hip), forward chains: whole chains on state columns held in LDS. For these
                                   two classes `bytes` counts the LDS bytes the sweeps gather (HBM traffic: the states in and out) */
#define DQ_PROF_STATECOL_BWD 10 /* state-column form, adjoint sweeps (reversal, forward terms, Horner with the gradient pairings) */
#define DQ_PROF_NCLASS 11

typedef struct dq_profile {
  long long launches[DQ_PROF_NCLASS];
  double ms[DQ_PROF_NCLASS];     /* sum of HIP-event durations of the class's launches       */
  double flops[DQ_PROF_NCLASS];  /* real flops per class. ZGEMM: the algorithmic 8*M*N*K (the 3M form executes 6*M*N*K);
                                    RGEMM: 2*M*N*K; RSYM: the EXECUTED 2*K*128*128*(tiles on/above the diagonal),
                                    about half of the 2*M*N*K of the full product it stands for; ZHERM: the
                                    algorithmic 8*M*N*K of the product delivered (about half of it is executed) */
  double bytes[DQ_PROF_NCLASS];  /* algorithmic bytes moved (streaming classes)              */
} dq_profile;

const char* dq_version(void);
const char* dq_last_error(void);
int dq_device_count(int* n);
/* Host-only (no GPU call): the step polynomial p(A) = sum_{m <= degree} factors[m] A^m / m! the state-column kernels use for
 * exp(A), A anti-Hermitian with ||A|| <= theta, at sup-norm error <= tol on the spectrum -- the Chebyshev-economized form
 * (economize = 1; one to four degrees below Taylor's) or the Taylor polynomial (0: factors = 1).  factors: 25 doubles;
 * degree = 0 when no degree up to 24 reaches tol.  (The reference's step is Eigen's `dH.exp()`, diffqc.cc:198, and
 * `scipy.linalg.expm(dH)`, sim_plain.py:145 -- scaling-and-squaring Pade forms on the CPU: any polynomial within tol of exp on the
 * spectrum is the same step to that tolerance.) */
int dq_step_polynomial(double theta, double tol, int economize, int* degree, double* factors);
/* Host-only: the tables of the quadrature form of the state-column adjoint sweep for a step of size <= theta and a step polynomial of
 * the given degree -- the 4-point Gauss-Lobatto rule of dL/du_k[t] = Re int_0^1 <lambda(s), dA_k psi(s)> ds (nodes = 2 interior nodes;
 * 0 when the rule's error bound 6.6e-7 (2 theta)^6 exceeds tol).  basis 0: node values, phase 1 accumulates vector i with c1[2 m + i]
 * (lambda(s_i) = sum_m c1 z_m) and q_m = w1[m] lambda_t + sum_i c2[2 m + i] lambda(s_i) (+ w0 lambda_{t+1} at m = 0); basis 1: the same
 * rule in the difference basis of the kernels that keep its second vector in LDS as fp32 (u_1, u_2 of order theta, theta^2; same
 * formulas, transformed tables).  w1, c1, c2: 25, 50, 50 doubles.  tests/np_state_column.py holds the numpy twin. */
int dq_quadrature_rule(double theta, double tol, int degree, int economize, int basis, int* nodes, double* w0, double* w1, double* c1,
                       double* c2);

/* Context = one Hilbert-space dimension + one set of Hamiltonians on one GPU.
 * Replaces the process-global store g_H0/g_Hs of diffqc.cc:21-25 with an explicit handle. */
int dq_create(dq_ctx** ctx, int dim, int n_controls, int device_id);                 /* DQ_C128 */
int dq_create_ex(dq_ctx** ctx, int dim, int n_controls, int device_id, int dtype);
int dq_destroy(dq_ctx* ctx);
/* Use the caller's HIP stream (e.g. torch.cuda.current_stream().cuda_stream); NULL = default stream. */
int dq_set_stream(dq_ctx* ctx, void* hip_stream);
int dq_synchronize(dq_ctx* ctx);

/* H0: [dim][dim] complex; Hs: [n_controls][dim][dim] complex.  (set_H, diffqc.cc:43-67) */
int dq_set_hamiltonians(dq_ctx* ctx, const void* H0, const void* Hs);
/* Observable M ([dim][dim], Hermitian) of the energy loss <psi|M|psi> (sim_plain.py:281). */
int dq_set_observable(dq_ctx* ctx, const void* M);

/* Options (dq_set_option; unknown names -> DQ_ERR_INVALID). Defaults pick the fastest scheme for the shape at hand.
 *   expm_tol           Taylor remainder bound of the step exponential (2^-53 fp64, 2^-24 fp32)
 *   force_degree, force_squarings   pin the complex tape's polynomial degree / number of squarings
 *   real_path          -1 auto / 0 off / q = 1..6 forced. H0 and all H_k real: exp(-i dt H) = cos X + i sin X with
 *                      X = -dt H real symmetric, evaluated with q + 1 real symmetric-output products (Taylor degree 2q + 1)
 *                      and differentiated with 2q; larger steps are scaled by 2^-s and squared in the real algebra (three
 *                      symmetric products per squaring); auto = cheapest (q, s) meeting expm_tol
 *   real_min_dim       auto real path only from this dimension up (128; below it the chain is launch-bound)
 *   compose            -1 auto / 0 / 1: operator form. Forward: the step propagators are multiplied together by a
 *                      log-depth tree of batched products and applied once (auto: batch > dim or dim <= small_dim);
 *                      backward: the sweep runs on W = Lambda Psi^H (dim x dim) and keeps no state checkpoints
 *                      (auto: batch >= 3/4 dim or dim <= small_dim)
 *   small_dim          512: dimensions up to it take the operator form for any batch width
 *   time_parallel      -1 auto / 0 / 1: operator backward sweep of a time chunk from suffix products (~4n batched
 *                      products instead of 2n dependent launches; auto: chunks of >= 4 steps at dim <= 1024)
 *   skew_trick         1: only the anti-Hermitian parts of the generator cotangent / of W are carried (saves products)
 *   sparse_controls    real path: 1 (default) = when the control Hamiltonians are sparse (at most 32 nonzeros per row summed
 *                      over the controls -- local Paulis have one each), assembly scatters their nonzeros instead of
 *                      streaming K dense matrices, and the last level of the reverse mode together with the control
 *                      contractions <Xbar, H_k> becomes nnz(H) row dot products instead of two two-product symmetric
 *                      launches (4 d^3 flop); 0 = always the dense passes. dq_get_info("sparse_controls") tells which
 *   sparse_generator   real path: -1 (default, auto: dim >= 64 x the nonzeros per row, where a sparse x dense pass beats an MFMA
 *                      product) / 1 (whenever possible) = when the DRIFT is sparse as well (same bound, drift included: e.g. a diagonal
 *                      Ising drift with local-Pauli controls), X_t itself is sparse and the whole polynomial runs on
 *                      sparse x dense passes: powers X_{m+1} = X X_m, cos/sin as linear combinations of the powers, reverse
 *                      mode Xbar_m = coef (Cbar | Sbar) + X Xbar_{m+1}, gradient = row dot products over nnz(H_k) level by
 *                      level -- no real MFMA product left in the step exponential (steps that need squarings keep the
 *                      dense scheme); 0 = off. dq_get_info("sparse_generator") / ("sparse_drift") tell what is in use
 *   fuse_levels        sparse generator: 1 (default) = the gradient's level dot products are taken inside the reverse passes
 *                      (the partner rows stream through the registers anyway), 0 = a separate contraction kernel (tests)
 *   cache_bytes        real path: HBM for the propagator intermediates (X, Y_j, P, U) that the forward pass leaves in place
 *                      for the backward sweep, which then skips its q + 1 rebuild products for those steps; < 0 off,
 *                      0 = 80 % of the free HBM (operator form) / 50 % (after the state form's checkpoints), at most 96 GiB
 *   checkpoint_bytes   state checkpoints of the state-form sweep (batch < dim); 0 = 70 % of the free HBM
 *   plan_steps         size the checkpoint stride for a chain of this many steps even if the call is shorter
 *   chunk_bytes        workspace for the steps whose propagators are built in one batched pass; 0 = 8 GiB, or 10 % of
 *                      the free HBM if that is more and dim <= 1024
 *   zgemm_alg          0 = four real MFMAs per complex tile step, 1 = the three-multiplication "3M" form (128x128 tile),
 *                      2 (default) = 3M and, for whole 128-tiles at dim >= 1024 whose left operand can be read k-major
 *                      (A^H, or an A the engine knows to be complex symmetric / (anti-)Hermitian), the 16-wave
 *                      linear-form kernel (csrc/zgemm_lin.hip)
 *   transpose_fallback 1 (default): big single complex products whose operands have no k-major form (A as stored without a symmetry
 *                      hint, B^H) get one by a transposing pre-pass into scratch, so that they run on the linear-form kernel too
 *   slab_spmm          1 (default): the sparse x dense passes of the sparse-generator scheme in XCD-slab form (16-B lanes, one
 *                      column slab per XCD; dim a multiple of 1024 in fp64 / 2048 in fp32), 0 = one wave per whole row
 *   xor_regular        -1 / 1 (default): a sparse generator whose pattern is XOR-regular (besides its diagonal, row i holds exactly
 *                      the entries (i, i ^ x_j) for the same offsets x_j in every row, each XOR-diagonal from one source matrix and
 *                      of one magnitude: sums of real Pauli strings, e.g. an Ising drift with X_k controls) takes the column-form
 *                      kernels of csrc/xorexp.hip (dim 256 .. 4096, polynomial q <= 6): U and the control gradients
 *                      column by column in LDS, no dense intermediate; 0 = the row-wise sparse x dense passes.  Complex Hermitian
 *                      generators of the same kind (X and Y drives: XOR-diagonals with entries +-c or +-i c, real diagonals: X, Y and Z
 *                      drives; complex128) take the complex form of the same kernels instead of the tape (info "xor_regular" = 2).
 *                      dq_get_info("xor_regular_found") = number of XOR-diagonals found (-1: not of that class),
 *                      ("xor_regular") = 1 if the last call ran the column form
 *   state_columns      -1 (default, auto) / 0 / 1: the STATE-COLUMN form (csrc/statecol.hip) for the same class of generators (real
 *                      or complex tables, both dtypes, up to 32 XOR-diagonals, dim 256 .. 4096, and 8192 in complex64): no propagator is built -- every state
 *                      column (dim complex numbers: 64 KB at dim 4096) stays in LDS for the WHOLE chain, a step is L sweeps of
 *                      partner reads (the Taylor polynomial of exp(-i dt H(t)) applied to the column), the adjoint sweep runs the
 *                      same way on (psi, lambda) with psi_t re-derived as U_t^H psi_{t+1} and leaves dL/du as per-diagonal sums.
 *                      Cost ~ batch x dim x diagonals per step instead of dim^3: auto takes it from dim 512 up while batch <=
 *                      state_columns_batch_ratio (8) x dim and no option pins another scheme (compose >= 0, real_path != -1,
 *                      force_degree, force_squarings, xor_regular = 0, sparse_generator = 0); 1 = whenever the kernels exist.
 *                      dq_get_info("state_columns") = 1 if the last call ran it ("xor_regular" then reports the table's kind, 1 / 2)
 *   state_columns_variant   kernel configuration: bit 0 = "wide" workgroups (dim 4096: 1024 threads x 4 rows instead of 512 x 8;
 *                      dim <= 1024: 256 threads per column instead of one or two waves -- +20 % at dim 512 with narrow batches);
 *                      bit 1 = partner prefetch in the forward kernel; default 0
 *   state_columns_adjoint   -1 (default) / 1: the adjoint sweep of the state-column form in its QUADRATURE form whenever a rule the
 *                      kernels are built for keeps its error bound below state_columns_quad_tol at the chain's largest step: with
 *                      phi(s) = exp(-s A_t) phi_{t+1}, dL/du_k[t] = Re int_0^1 <lambda(s), dA_k psi(s)> ds by a 4-point Gauss-Lobatto
 *                      rule whose node values are sums of the Taylor terms the propagation of lambda produces anyway (2 L sweeps per
 *                      step, no scratch; up to 16 XOR-diagonals, bound e (2 theta)^6 <= tol: theta <= 0.05 at the default);
 *                      0 = always the exact derivative of the step polynomial (3 L - 1 sweeps, forward terms parked in a scratch).
 *                      dq_get_info("state_columns_quad_nodes") = interior nodes of the rule the last adjoint sweep used (0: exact form)
 *   state_columns_polynomial   1 (default) = Chebyshev-economized step polynomial (dq_step_polynomial: one to four sweeps fewer per
 *                      step at the same tolerance), 0 = Taylor.  dq_get_info("degree") = the degree in use,
 *                      ("state_columns_taylor_degree") = the degree Taylor needs at the same tolerance
 *   state_columns_compose   -1 (default): where the forward pass applies the PRODUCT of the chain's propagators (batch wider than the
 *                      dimension, dim <= small_dim, compose = 1) and H is a sum of Pauli strings, that product comes from the
 *                      state-column forward kernel run on the identity's columns -- dim columns x n_steps steps, cut into time segments
 *                      (as many as fill the chip) whose propagators are multiplied afterwards -- instead of one step exponential and
 *                      one dim^3 product per step; dq_propagate also prefers this form over sweeping the states when its cost model
 *                      says so (batch > dim and enough steps).  0: explicit propagators; v > 0: that many segments.
 *                      dq_get_info("state_columns_compose_segments") = segments of the last call (0: not used)
 *   state_columns_wide_adjoint   -1 (default: by a cost model) / 1: dq_energy_grad of a batch WIDER than the dimension (energy objective,
 *                      Pauli-sum H) does not sweep the B states: L = sum_b <psi_b|M|psi_b> = tr(M P R P^H) with R = Psi0 Psi0^H and P the
 *                      chain's product (state_columns_compose), and dL/du comes from the state-column adjoint sweep over the dim columns
 *                      of (psi_S, lambda_S) = (P R, 2 M P) -- three dim^3 products and the Gram product per call instead of B - dim more
 *                      columns per step; final states / per-state energies, when asked for, cost one more dim x dim x B product.
 *                      0: always sweep the states.  dq_get_info("state_columns_wide_adjoint") = 1 if the last call took it
 *   state_columns_chain_tol   bound of the step polynomial's error summed over the chain (default 1e-11 in complex128, 2e-5 in
 *                      complex64): the per-step bound is min(expm_tol, state_columns_chain_tol / n_steps) -- the polynomial's error is
 *                      the same function of the spectrum at every step and may add up coherently
 *   state_columns_part_bytes  scratch of the adjoint sweep's per-step partial sums (default 256 MiB): a longer chain is swept in segments
 *                      that fit (the sweep restarts at any step: psi and lambda are in/out); dq_get_info("state_columns_segments")
 *   state_columns_workgroups  0 (default): as many workgroups as fit the chip; > 0 caps a launch (tests of the several-columns-per-
 *                      workgroup loop at small batches)
 *   state_columns_quad_tol  bound of the rule's error per step relative to ||dA_k|| ||lambda|| ||psi|| (default 2^-40)
 *   force_tile, rgemm_cfg, rgemm_glds   kernel configurations (tests, tuning)
 * dq_get_info names: "degree", "squarings", "products", "theta", "time_chunk", "checkpoint_stride", "hermitian",
 * "real_hamiltonians", "real_path" (q used by the last call, 0 = complex tape), "composed", "operator_backward",
 * "time_parallel", "cached_steps", "slice_skew", "sparse_controls", "control_nonzeros" (-1 = dense), "sparse_generator", "sparse_drift", "xor_regular", "xor_regular_found", "state_columns", "state_columns_quad_nodes", "zgemm_alg", "norm_H0", "dim",
 * "n_controls", "dtype". */
int dq_set_option(dq_ctx* ctx, const char* name, double value);
int dq_get_info(dq_ctx* ctx, const char* name, double* value);

/* Forward chain: psi_out = prod_{t=n_steps-1..0} exp(-i dt (H0 + sum_k u[t][k] H_k)) psi_in.
 * The time loop of trotter(), diffqc.cc:190-200 / sim_plain.py:135-150, for `batch` states at once.
 * dt may be negative (diffqc.cc:183).  psi_out may equal psi_in. */
int dq_propagate(dq_ctx* ctx, const double* u, int n_steps, double dt, const void* psi_in, void* psi_out, int batch);

/* energies[b] = Re <psi_b| M |psi_b>, total = sum_b.  Either output may be NULL. */
int dq_energy(dq_ctx* ctx, const void* psi, int batch, double* energies, double* total);

/* Forward chain + adjoint sweep. loss = sum_b <psi_b(T)|M|psi_b(T)>,
 * grad_u[t][k] = d loss / d u[t][k] (exact derivative of the discrete chain).
 * psi_final ([dim][batch]) and energies ([batch]) are optional (NULL to skip). */
int dq_energy_grad(dq_ctx* ctx, const double* u, int n_steps, double dt, const void* psi0, int batch, double* loss,
                   double* grad_u, void* psi_final, double* energies);

/* ---- Column-sharded form of the wide-batch energy gradient (BASELINE configs[3]: 32768 states over 8 GPUs; round 5).
 * The loss of the reference's loop (diffqc.cc:190-200, sim_plain.py:281) summed over a batch is L = tr(M P R P^H) with P the chain's
 * product and R = sum_b psi0_b psi0_b^H: it depends on the states through the d x d Gram matrix alone.  A rank
 *   1. dq_gram:                 R_rank = Psi0_rank Psi0_rank^H of ITS states ([dim][dim] complex, out to host or device memory);
 *   2. all-reduces R over the ranks (the caller's collective: 16 dim^2 bytes);
 *   3. dq_energy_grad_columns:  runs the adjoint sweep on the columns [col0, col0 + n_cols) of (P R, 2 M P): loss = its share
 *      1/2 Re sum_i <lambda_i, y_i>, grad_u[t][k] = its share of dL/du.  P = NULL: the call builds the chain's product itself
 *      (the state-column kernels on the identity's columns: a forward pass over dim columns, replicated on every rank); P given
 *      ([dim][dim], U_{S-1} ... U_0): made elsewhere -- e.g. one time slice per rank by dq_slice_forward, all-gathered and multiplied
 *      (dq_zgemm), which splits that forward pass over the ranks as well;
 *   4. all-reduces (loss, grad_u).
 * The sums over all ranks equal dq_energy_grad of the whole batch on one GPU.  Needs dq_set_observable and a generator the
 * state-column kernels take (else DQ_ERR_UNSUPPORTED); n_cols = 0 is allowed (more ranks than columns). */
int dq_gram(dq_ctx* ctx, const void* psi0, int batch, void* R_out);
int dq_energy_grad_columns(dq_ctx* ctx, const double* u, int n_steps, double dt, const void* R, const void* P, int col0, int n_cols,
                           double* loss, double* grad_u);

/* Fidelity objective for a batch of (initial, target) pairs -- the batched form of train_fidelity's
 * M = |psi1><psi1| (sim_plain.py:436-455): loss = sum_b (1 - |<target_b|psi_b(T)>|^2), grad_u as above,
 * fidelities[b] = |<target_b|psi_b(T)>|^2.  targets is [dim][batch] like psi0. */
int dq_fidelity_grad(dq_ctx* ctx, const double* u, int n_steps, double dt, const void* psi0, const void* targets,
                     int batch, double* loss, double* grad_u, void* psi_final, double* fidelities);

/* The reference's Monte-Carlo "parameter shift in time" estimator (compute_energy_grad_MC, sim_plain.py:156-231) as one
 * device-resident operation: phi = chain_a psi0 (n_a steps of dt_a: the interval 0 -> s on its own grid,
 * sim_plain.py:190), the 2K shifted states (I +- i r H_k) phi / sqrt(1 + r^2) (sim_plain.py:194-199) evolved together as
 * one batch through chain_b (s -> T, sim_plain.py:200,210), measured with M, and
 *     ps[b][k] = (1 + r^2) / (2 r) * (<M>_minus - <M>_plus)      (sim_plain.py:220; r = 1/2 in the reference).
 * The host multiplies by du_k(s)/dcoeff (sim_plain.py:170-184,227). phi_out ([dim][batch]) is optional. */
int dq_mc_shift_grad(dq_ctx* ctx, const double* u_a, int n_a, double dt_a, const double* u_b, int n_b, double dt_b,
                     const void* psi0, int batch, double r, double* ps, void* phi_out);

/* ---- Time-sliced operator form: the phases of dq_energy_grad, for a chain split over TIME across GPUs
 * (the reference's time loop diffqc.cc:190-200 / sim_plain.py:135-150 is sequential).
 * In operator form (batch >= 3/4 dim) the work of a gradient evaluation is d x d matrix products per step, independent
 * of the batch; a rank that owns the steps [a, b) of the chain
 *   1. dq_slice_forward : builds their propagators and returns the slice product  P = U_{b-1} ... U_a  ([dim][dim] complex;
 *                         the identity for an empty slice); real path: the propagator intermediates stay cached;
 *   2. (host) all-gathers the P's, forms Q = P_{G-1} ... P_0 and R = the product of the LATER slices (dq_zgemm);
 *   3. dq_slice_terminal: Psi_S = Q Psi_0 for its own batch shard, loss / energies (fidelities with `targets`, as
 *                         dq_fidelity_grad), and  W_S = Lambda_S Psi_S^H  ([dim][dim]; its anti-Hermitian part when
 *                         dq_get_info("slice_skew") is 1) -- the host sums W_S over the ranks (the only batch reduction);
 *   4. (host)             W_b = R^H W_S R  (dq_zgemm_ex; anti-Hermitian output form if slice_skew);
 *   5. dq_slice_backward: sweeps its own steps from W_end = W_b:  G_t = W_{t+1} U_t,  W_t = U_t^H G_t, reverse mode of
 *                         exp and the control contractions -> grad_u[n_steps][n_controls] (its rows of dL/du, already
 *                         summed over the WHOLE batch); W_start (optional) receives W_a.
 * u holds the slice's own rows of the pulse table. With one slice covering the chain these three calls compute
 * exactly what dq_energy_grad computes. */
int dq_slice_forward(dq_ctx* ctx, const double* u, int n_steps, double dt, void* P_out);
int dq_slice_terminal(dq_ctx* ctx, const void* Q, const void* psi0, int batch, const void* targets, double* loss,
                      double* energies, void* psi_final, void* W_out);
int dq_slice_backward(dq_ctx* ctx, const double* u, int n_steps, double dt, const void* W_end, double* grad_u,
                      void* W_start);

/* ---- Explicit propagators in and out: SURVEY.md section 8-F2 in its literal form ("each GPU builds S/G propagators and
 * all-gathers them"), for the STATE-form regime (batch per GPU < 3/4 dim) where every rank needs every U_t for its own columns.
 * dq_build_propagators: U_out[t] = exp(-i dt (H0 + sum_k u[t][k] H_k)) for the n pulse rows given ([n][dim][dim] complex, host or
 *   device) -- the reference builds the same matrix inside its time loop (diffqc.cc:192-198 `dH.exp()`, sim_plain.py:137-145).
 * dq_propagator_vjp: pulls cotangents G[t] of those propagators back to the pulse rows with the exp's exact reverse mode and the
 *   control contractions: grad_u[t][k] = Re <G[t], dU_t/du_k> (pairing Re tr(G^H dU)); needs Hermitian H0/H_k.
 * With dq_zgemm (psi_{t+1} = U_t psi_t, lambda_t = U_t^H lambda_{t+1}, G_t = lambda_{t+1} psi_t^H summed over the ranks' columns)
 * a host distributes the exponentials of a chain over G ranks: diffquantum_amd/distributed.py:propagator_sharded_energy_grad. */
int dq_build_propagators(dq_ctx* ctx, const double* u, int n_steps, double dt, void* U_out);
int dq_propagator_vjp(dq_ctx* ctx, const double* u, int n_steps, double dt, const void* G, double* grad_u);

/* Building blocks, exported for parity tests and reuse. */
/* C = alpha*op(A)*op(B) + beta*C, batched; op: 0 = N, 1 = conjugate transpose; strides in complex elements.
 * opA may also be 2 / 3 / 4: A as stored (like 0), square, and KNOWN by the caller to be complex symmetric (A^T = A) /
 * Hermitian / anti-Hermitian -- a hint that lets the kernel read A k-major (linear-form kernel); the result is the same
 * as with opA = 0 for operands that have the stated property.  For an operand that has it only up to a defect (rounding, or the
 * engine's herm_tol), the result is that of op(A') with A' = A^T / A^H / -A^H: it differs from opA = 0 by the product of the
 * defect with B, and whether the hint takes effect depends on the shape (whole 128-tiles, K >= 1024, batch 1) -- the hint is
 * never checked. */
int dq_zgemm(dq_ctx* ctx, int opA, int opB, int M, int N, int K, double alpha, const void* A, long lda, long strideA,
             const void* B, long ldb, long strideB, double beta, void* C, long ldc, long strideC, int batch);
/* The same with herm = +1 / -1: the caller asserts that the product is Hermitian / anti-Hermitian (e.g. X^H W X for an
 * (anti-)Hermitian W; M == N): only the tiles on and above the diagonal are computed, the mirror image is +-conj, the
 * diagonal is made exactly real / imaginary. beta != 0 then needs an (anti-)Hermitian C on entry. */
int dq_zgemm_ex(dq_ctx* ctx, int opA, int opB, int M, int N, int K, double alpha, const void* A, long lda, long strideA,
                const void* B, long ldb, long strideB, double beta, void* C, long ldc, long strideC, int batch, int herm);
/* Real GEMM of the real-symmetric fast path: C = alpha*op(A)*op(B) + beta*C, batched; op: 0 = N, 1 = transpose;
 * element type double (DQ_C128 contexts) or float (DQ_C64); strides in elements. sym != 0: the caller asserts that
 * the product is symmetric (commuting symmetric operands; M == N): only the tiles on and above the diagonal are
 * computed and mirrored. sym == 2: C = alpha*(A*B + B*A) + beta*C for symmetric A, B (M == N == K; beta != 0 needs a
 * symmetric C), same tiles, both products in one launch. */
int dq_rgemm(dq_ctx* ctx, int opA, int opB, int M, int N, int K, double alpha, const void* A, long lda, long strideA,
             const void* B, long ldb, long strideB, double beta, void* C, long ldc, long strideC, int batch, int sym);
/* U[z] = exp(A[z]) for `batch` dim x dim matrices. */
int dq_expm(dq_ctx* ctx, const void* A, void* U, int batch);
/* Abar[z] = adjoint Frechet derivative: <Ubar, dexp_A(E)> = <Abar, E> for all E (real inner product Re tr(X^H Y)). */
int dq_expm_vjp(dq_ctx* ctx, const void* A, const void* Ubar, void* Abar, int batch);

/* Per-class HIP-event timing of every kernel launch (off by default). */
int dq_profile_enable(dq_ctx* ctx, int on);
int dq_profile_reset(dq_ctx* ctx);
int dq_profile_get(dq_ctx* ctx, dq_profile* out);

/* ---- Structured (matrix-free) engine for large qubit counts -----------------------
 * H(t) = H0 + sum_k u_k(t) H_k with every operator a real-weighted sum of Pauli strings; the dense 2^n x 2^n
 * matrices of dq_set_hamiltonians never exist. Same step as the dense engine -- psi <- exp(-i dt H(t)) psi, the
 * exponential of the SUM (diffqc.cc:192-198, sim_plain.py:137-146) -- evaluated as a Taylor polynomial of
 * matrix-free H.psi passes; same [2^n][batch] state layout, same pulse-table / gradient conventions.
 * A string is (owner, xmask, zmask, weight): owner -1 = drift H0, 0..K-1 = control H_k, -2 = observable M;
 * bit q of xmask set <=> X or Y on the qubit whose basis-index bit is q; bit q of zmask set <=> Z or Y. */
typedef struct dq_sctx dq_sctx;
const char* dq_sctx_last_error(void);
int dq_sctx_create(dq_sctx** ctx, int n_qubits, int n_controls, int device_id, int dtype);
int dq_sctx_destroy(dq_sctx* ctx);
int dq_sctx_set_stream(dq_sctx* ctx, void* hip_stream);
int dq_sctx_set_terms(dq_sctx* ctx, int n_terms, const int* owner, const unsigned* xmask, const unsigned* zmask,
                      const double* weight);
/* options: "tol" (bound of the step polynomial's error per step), "force_degree" (Taylor of that degree), "checkpoint_bytes",
 *   "profile" (1 on, 0 off, -1 reset);
 *   "polynomial" (1, default: the Chebyshev-economized step polynomial of dq_step_polynomial in Horner form, normalised to constant
 *                term 1, taken when its degree is below Taylor's; 0: Taylor), "chain_tol" (bound of the polynomial's error summed
 *                over the chain: per step min(tol, chain_tol / max(n_steps, plan_steps)); 1e-11 in complex128, 2e-5 in complex64),
 *   "plan_steps" (checkpoint stride as for a chain of this many steps: a timed slice of a long chain pays the long chain's reversals),
 *   "two_level" (1, default: when not every state fits but steps / s + m s states do (m = Taylor degree, s = sqrt(steps / m)), every
 *                s-th state is kept and a segment at a time is recomputed with the Horner iterates of its steps stored; 0: one level),
 *   "host_checkpoints" (1, default: when the HBM holds one segment (m s states) but not the kept states as well, those live in pinned
 *                host memory -- one state down per s steps, one up per segment; 0: never),
 *   "host_checkpoint_bytes" (0, default: those kept states may pin up to 60 % of the host memory the process may still use -- cgroup
 *                limit or MemAvailable; > 0: that many bytes.  A chain whose kept states exceed it runs the one-level plan instead),
 *   "dma" (1, default: complex64 passes over 2^9-row tiles with batch % 16 == 0 run as persistent LDS-DMA kernels; 0: one tile per workgroup),
 *   "dma_workgroups" (persistent workgroups per pass; default = the device's CU count),
 *   "chain" (1, default: a Horner chain over two pure X-string groups finishes one application and starts the next in one pass),
 *   "reverse_chain" (1, default: so does the reverse sweep, overlaps of both stages included),
 *   "step_handover" (1, default: the pass that finishes a step's Horner chain also starts the next step's, with that step's
 *   coefficients and diagonal from a second set; 0: a finish pass and a start pass), "reverse_keep_last" (0, default: the reverse
 *   chain's last stage stores no zbar_m -- nobody reads it; 1: the form with the store),
 *   "interleave_groups" (set before dq_sctx_set_terms; 2, default: two groups of single-qubit flips get the qubits in turn, so that
 *                    both tiles touch few 2 MB regions; 0: low qubits to one group, high qubits to the other),
 *   "fuse_overlaps" (1: the reverse-mode apply pass of a group also forms that group's control overlaps, in every kernel form;
 *                    0, default: only in the DMA form, where it costs no occupancy; -1: never).
 * info: "degree", "taylor_degree" (Taylor's at the same bound), "theta", "groups", "checkpoint_stride", "passes", "dma_passes", "two_level", "host_checkpoints", "pass_bytes", "checkpoint_pool_bytes" (device bytes the checkpoint pools hold now), "pinned_host_bytes" (pinned host memory of the kept states), "prof_ms", "prof_bytes", "prof_launches", "dtype" */
int dq_sctx_set_option(dq_sctx* ctx, const char* name, double value);
int dq_sctx_get_info(dq_sctx* ctx, const char* name, double* value);
int dq_sctx_propagate(dq_sctx* ctx, const double* u, int n_steps, double dt, const void* psi_in, void* psi_out,
                      int batch);
int dq_sctx_energy_grad(dq_sctx* ctx, const double* u, int n_steps, double dt, const void* psi0, int batch,
                        double* loss, double* grad_u, void* psi_final, double* energies);
/* out = H(u_row) in (which = 0) or M in (which = 1): one matrix-free application (parity tests). */
int dq_sctx_apply(dq_sctx* ctx, int which, const double* u_row, const void* in, void* out, int batch);

#ifdef __cplusplus
}
#endif
#endif /* DIFFQC_HIP_H */
