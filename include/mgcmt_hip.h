/*
 * mgcmt_hip.h — C-ABI of libmgcmt_hip.so: the MI355X (gfx950) multigrid V-cycle hot path of
 * AndyMN/MultigridCMT behind plain pointers and sizes.
 *
 * The reference is pure Python (no FFI of its own); the boundary it offers is the class API of
 * MGCMTSolver / MGCMTStencilMaker / MGCMTProcessor.  Each entry point below names the reference
 * function(s) it replaces (paths relative to the reference root).  The Python classes in
 * multigridcmt_amd/ bind these with ctypes (see INTEGRATION.md for the binding a maintainer of
 * the reference would add).
 *
 * Conventions
 *   - every function returns 0 on success, a negative mgcmt_status otherwise; the message of the
 *     last failure on the calling thread is mgcmt_last_error().  No C++ exception crosses the ABI.
 *   - host buffers are C-contiguous IEEE fp64; the library never keeps a host pointer after return.
 *   - a plan belongs to one GPU and is not thread-safe; all work of a call is enqueued on the
 *     `stream` argument (a hipStream_t passed as void*, NULL = the default stream).
 *   - grids: a 2-D level is rows x cols with vector index k = i*cols + j (the reference's
 *     kronsum/kron ordering, MGCMTStencilMaker.py:23-24,53); a 1-D level is 1 x n.
 *   - operators are sums of Kronecker products of tridiagonal factors,
 *         A = sum_m X_m (x) Y_m  - shift * I,
 *     X_m over rows, Y_m over columns, each factor given as three arrays (lower, diagonal, upper).
 *     laplacian(n,"1d") is one term (X = [1], Y = tridiag(1,-2,1)/h^2, MGCMTStencilMaker.py:17-21);
 *     laplacian(n,"2d") = I(x)L + L(x)I is two (MGCMTStencilMaker.py:23-24).  Galerkin coarse
 *     operators R*A*P (MGCMTSolver.py:318) keep this form with X_m <- R1 X_m P1, Y_m <- R1 Y_m P1,
 *     which the library computes at plan creation.
 */
#ifndef MGCMT_HIP_H
#define MGCMT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MGCMT_ABI_VERSION 6
#define MGCMT_MAX_TERMS 4
#define MGCMT_HALO_ROWS 16 /* rows of halo kept above and below every vector of a 2-D level (a 1-D level keeps one: its
                              "row" is the whole vector); how many of them a sharded cycle fills: mgcmt_plan_level_halo */

typedef enum mgcmt_status {
  MGCMT_OK = 0,
  MGCMT_ERR_INVALID = -1,   /* bad argument (sizes not powers of two, level out of range, ...) */
  MGCMT_ERR_HIP = -2,       /* a HIP runtime call failed */
  MGCMT_ERR_NOMEM = -3,
  MGCMT_ERR_UNSUPPORTED = -4
} mgcmt_status;

/* smoother kinds; MGCMTSolver.py:182-246 and the (commented-out) gseidelrb :248-279 */
typedef enum mgcmt_smoother {
  MGCMT_WJACOBI = 0,   /* wjacobi  :182-208  v <- v + w D^-1 (f - A v)                              */
  MGCMT_GS_LEX = 1,    /* gseidel  :210-227  forward lexicographic Gauss-Seidel (index order)        */
  MGCMT_SOR_LEX = 2,   /* sor      :229-246  incl. its (D-L)^-1 right-hand-side term (:241)          */
  MGCMT_GS_MC = 3      /* multicolour GS/SOR: 1-D odd/even, 2-D colours (i%2,j%2) in the order
                          (0,1),(1,0),(0,0),(1,1) = red-black on a 5-point operator                 */
} mgcmt_smoother;

/* vector slots of a level */
typedef enum mgcmt_slot { MGCMT_SLOT_V = 0, MGCMT_SLOT_F = 1, MGCMT_SLOT_T = 2, MGCMT_SLOT_W = 3 } mgcmt_slot;

/* which operator of the plan: A, or the mass operator M of rqmin (MGCMTSolver.py:17-57) */
typedef enum mgcmt_opsel { MGCMT_OP_A = 0, MGCMT_OP_M = 1 } mgcmt_opsel;

typedef struct mgcmt_plan mgcmt_plan;

typedef struct mgcmt_plan_desc {
  int32_t dim;          /* 1 or 2 */
  int32_t nterms;       /* Kronecker terms of A (1-D: 1) */
  int64_t g;            /* fine grid size per direction (power of two) */
  int64_t lowest;       /* grid size at which the direct solve happens (vcycle's lowest_level) */
  const double* xfac;   /* [nterms][3][g] row factors (lower, diag, upper); NULL for dim == 1 */
  const double* yfac;   /* [nterms][3][g] column factors */
  int32_t m_nterms;     /* Kronecker terms of the mass operator M, 0 = none */
  const double* m_xfac; /* as xfac, for M */
  const double* m_yfac;
  int32_t nvec;         /* number of simultaneous vectors (columns of vcycle_matrix), >= 1 */
  int32_t device;       /* HIP device ordinal */
  int64_t row_begin;    /* rows [row_begin,row_end) of the fine level owned by this plan ...        */
  int64_t row_end;      /* ... (0,g) on one GPU; multiples of 2^(strip_levels-1) when sharded       */
  int32_t strip_levels; /* how many of the finest levels are row strips; the rest are whole grids   */
  int32_t reserved;
} mgcmt_plan_desc;

const char* mgcmt_last_error(void);
int mgcmt_abi_version(void);
int mgcmt_device_count(int* count);
int mgcmt_device_name(int device, char* buf, int buflen);

/* hierarchy: levels, Galerkin factors (R*A*P, MGCMTSolver.py:318), vector storage */
int mgcmt_plan_create(const mgcmt_plan_desc* desc, mgcmt_plan** out);
int mgcmt_plan_destroy(mgcmt_plan* plan);
int mgcmt_plan_num_levels(const mgcmt_plan* plan, int* levels);
int mgcmt_plan_level_shape(const mgcmt_plan* plan, int level, int64_t* rows, int64_t* cols, int64_t* row_begin);
/* host copy of a level's factors, [nterms][3][n] with n = global rows (which=0) or cols (which=1) */
int mgcmt_plan_get_factors(const mgcmt_plan* plan, int op, int level, int which, double* out, int64_t capacity);
/* device address of the interior (row 0, col 0) of vector `vec` in `slot` of `level`; rows -halo..-1 and
 * rows..rows+halo-1 are addressable halo rows (halo: mgcmt_plan_level_halo) */
int mgcmt_vec_ptr(const mgcmt_plan* plan, int level, int slot, int vec, void** device_ptr);
/* halo_rows: rows kept above and below every vector of `level` (MGCMT_HALO_ROWS on 2-D levels, 1 on 1-D levels);
 * exchanged_rows: how many of them — the ones next to the strip — a sharded cycle fills with the neighbours' rows and its
 * passes read: 8 on levels with a 5-point operator, 10 on the 9-point (Galerkin) levels, whose two four-colour sweeps plus
 * restriction reach nine rows beyond a strip.  Either output may be NULL. */
int mgcmt_plan_level_halo(const mgcmt_plan* plan, int level, int* halo_rows, int* exchanged_rows);

/* shift(s) mu of (A - mu I): vcycle's shift= (MGCMTSolver.py:287-288), vcycle_matrix's shifts= (:385-388) */
int mgcmt_set_shifts(mgcmt_plan* plan, const double* shifts, int k, void* stream);

int mgcmt_upload(mgcmt_plan* plan, int level, int slot, int vec, const double* host, int64_t count, void* stream);
int mgcmt_download(mgcmt_plan* plan, int level, int slot, int vec, double* host, int64_t count, void* stream);
int mgcmt_fill(mgcmt_plan* plan, int level, int slot, int vec, double value, void* stream);
/* vector := 0 including its halo rows (a sharded cycle restarts the coarse iterate without an exchange) */
int mgcmt_zero(mgcmt_plan* plan, int level, int slot, int vec, void* stream);
int mgcmt_copy(mgcmt_plan* plan, int level, int src_slot, int src_vec, int dst_slot, int dst_vec, void* stream);
int mgcmt_sync(void* stream);

/* Page-locked host memory for the host side of mgcmt_upload / mgcmt_download (and mgcmt_csr_upload / _download).
 * A transfer to or from pageable memory is staged through a ring of pinned chunks by several host threads
 * (csrc/transfer.hip); one whose host pointer lies in a buffer of mgcmt_host_alloc — or in any other memory the HIP
 * runtime knows as page-locked — is a single DMA.  The reference returns fresh arrays from every call
 * (MGCMTSolver.py:326 `return v`); the drop-in classes return theirs in recycled buffers of this allocator
 * (multigridcmt_amd/hostmem.py), so a result that is fed back as the next call's v0 never touches pageable memory. */
int mgcmt_host_alloc(int64_t bytes, void** host_ptr);
int mgcmt_host_free(void* host_ptr);

/* smoothers on V,F of `level` for vectors 0..k-1 (MGCMTSolver.py:182-246) */
int mgcmt_smooth(mgcmt_plan* plan, int level, int kind, int nu, double omega, int k, void* stream);
/* F[level+1] <- R (F - (A - mu I) V), V[level+1] <- 0   (MGCMTSolver.py:315-316) */
int mgcmt_residual_restrict(mgcmt_plan* plan, int level, int k, void* stream);
/* V[level] <- V[level] + P V[level+1]                    (MGCMTSolver.py:323-324) */
int mgcmt_prolong_correct(mgcmt_plan* plan, int level, int k, void* stream);
/* V[last] <- (A_last - mu I)^-1 F[last]                  (spsolve, MGCMTSolver.py:305-308) */
int mgcmt_coarse_solve(mgcmt_plan* plan, int level, int k, void* stream);
/* one V-cycle from `level` down: vcycle (:281-329) for k == 1, vcycle_matrix (:375-436, incl. the
 * Gram-Schmidt at every non-coarsest level, :434) with MGCMT_CYCLE_GRAM_SCHMIDT in cycle_flags.  nu_coarse is the
 * sweep count on the levels below `level` (the reference does not forward nu1/nu2, so 4: :320,:426).
 * MGCMT_CYCLE_ZERO_START: the caller vouches that V[level] is zero, the way the reference's eigen-drivers start every
 * cycle (1DPotMatrixVcycle.py:70, v0 = zeros): the first pass then neither reads nor needs V cleared (ABI 4; the
 * argument was the 0 / 1 Gram-Schmidt switch before, which keeps its meaning). */
#define MGCMT_CYCLE_GRAM_SCHMIDT 1
#define MGCMT_CYCLE_ZERO_START 2
int mgcmt_vcycle(mgcmt_plan* plan, int level, int nu1, int nu2, int nu_coarse, int kind, double omega, int k,
                 int cycle_flags, void* stream);
/* twogrid (:331-371): exact solve of (R A P - mu I) on level+1 */
int mgcmt_twogrid(mgcmt_plan* plan, int level, int nu1, int nu2, int kind, double omega, int k, void* stream);

/* dst <- (Op - with_shift*mu I) src        (sparse `*` / .dot, e.g. MGCMTSolver.py:19,315) */
int mgcmt_apply(mgcmt_plan* plan, int op, int level, int src_slot, int src_vec, int dst_slot, int dst_vec,
                int with_shift, void* stream);
/* coarse <- R fine  /  fine <- P coarse    (restriction_matrix * k, MGCMTSolver.py:113,116) */
int mgcmt_restrict(mgcmt_plan* plan, int level, int src_slot, int src_vec, int dst_slot, int dst_vec, void* stream);
int mgcmt_prolong(mgcmt_plan* plan, int level, int src_slot, int src_vec, int dst_slot, int dst_vec, int accumulate,
                  void* stream);

/* vector algebra (MGCMTProcessor.py:10-73; np.dot / np.linalg.norm call sites of MGCMTSolver.py) */
int mgcmt_dot(mgcmt_plan* plan, int level, int slot_a, int vec_a, int slot_b, int vec_b, double* host_out, void* stream);
int mgcmt_axpy(mgcmt_plan* plan, int level, double alpha, int x_slot, int x_vec, int y_slot, int y_vec, void* stream);
/* all inner products <v_a, v_b> of nv <= 6 vectors (slots[a], vecs[a]) in ONE pass over the data; host_out is the
 * symmetric nv x nv Gram matrix, row-major.  The Rayleigh-Ritz steps of rqmin (MGCMTSolver.py:44-50: the entries of
 * its 2x2 matrices R and RM) and of the eigen-drivers need such sets; synchronises like mgcmt_dot. */
int mgcmt_gram(mgcmt_plan* plan, int level, int nv, const int* slots, const int* vecs, double* host_out, void* stream);
/* For the columns q < k of `slot`: rq_out[q] = <v_q, A v_q> / <v_q, v_q> and res_out[q] = ||(A - mu_q I) v_q||_2 with the
 * plan's current shifts mu — the two numbers the reference's eigen-drivers print per column and iteration
 * (2DPotMatrixVcycle.py:100-105: np.dot(v, hamiltonian.dot(v)) and np.linalg.norm(shifted_matrix.dot(v))).  One
 * k-column operator application (into slot W) and one reduction pass per column; ONE synchronisation for all of them.
 * Either output may be NULL. */
int mgcmt_rayleigh_residual(mgcmt_plan* plan, int level, int slot, int k, double* rq_out, double* res_out, void* stream);
/* The inner products of the 2 x 2 Rayleigh-Ritz problem on span{x, w} (rqmin's pencil, MGCMTSolver.py:44-50, when <x, A x>
 * is known from the previous step): out5 = <x,x>, <x,w>, <w,w>, <x,A w>, <w,A w> with the UNSHIFTED operator of `level`.
 * On 2-D levels with a 5-point operator (with or without a product potential) ONE pass reads x and w once and stores
 * nothing; elsewhere A w goes to the scratch vector (mgcmt_apply) and one mgcmt_gram pass follows.  The scratch vector must
 * differ from x and w.  Synchronises. */
int mgcmt_ritz_pair(mgcmt_plan* plan, int level, int x_slot, int x_vec, int w_slot, int w_vec, int scratch_slot, int scratch_vec,
                    double* out5, void* stream);
/* rqmin (MGCMTSolver.py:17-57): nu steps of Rayleigh-quotient minimisation on `level` for the plan's pair (A, M) (M = I
 * without a mass operator), entirely on the device — per step TWO passes over the data (p = -g + beta p_old formed on the
 * fly, A and M applied to x and p in registers, the eight inner products of the 2 x 2 pencil of :33-46; then x + delta p, its
 * gradient g = 2 (A x - rho M x) and the next step's inner products), the pencil solved in closed form by one workgroup,
 * no host round trip.  vecs[0] of `slot` holds the start vector and receives the result; vecs[1..5] are five more, distinct
 * vectors of the slot used as work space.  robust != 0: a degenerate pencil ends the minimisation instead of producing
 * infinities (the repaired variants).  rho_out (may be NULL: nothing synchronises then): the Rayleigh quotient of the
 * result (:53). */
int mgcmt_rqmin(mgcmt_plan* plan, int level, int slot, const int* vecs, int nu, int robust, double* rho_out, void* stream);
/* One line minimisation of the Rayleigh quotient of the plan's pair (A, M) on `level` along a direction the caller
 * supplies: the 2 x 2 problem of rqmin (MGCMTSolver.py:33-50) on span{x, w} — x_out = x + delta w — followed by
 * g = 2 (A x_out - rho M x_out) (:52-54), both on the device, the state words of mgcmt_rqmin reused; no host round trip.
 * With w the preconditioned residual (a V-cycle applied to g) this is the iteration of the 2-D square-well driver.
 * Vectors are {slot, vec} pairs, all distinct: x (read), w (read; NULL: only rho and g of x are computed, x_out unused),
 * x_out, g, and, with a non-identity mass operator, a work vector (NULL otherwise).  record >= 0: the Rayleigh quotient
 * of x_out is also stored as number `record` (< MGCMT_RQ_HISTORY) of the plan's device-side history, read back — with
 * the only synchronisation — by mgcmt_rq_history. */
#define MGCMT_RQ_HISTORY 4096
int mgcmt_rq_line_step(mgcmt_plan* plan, int level, const int* x, const int* w, const int* x_out, const int* g, const int* work, int robust,
                       int record, void* stream);
int mgcmt_rq_history(mgcmt_plan* plan, int first, int count, double* out, void* stream);
/* vcycle_rqmg (MGCMTSolver.py:99-122) from the finest level: rqmin with nu1 steps, the iterate restricted (:113), the same on
 * the Galerkin pair (R A P, R M P) of every coarser level of the plan (the coarsest one only minimises), the interpolated
 * coarse iterates added on the way up (:116-118) and nu2 more steps per level; vectors as for mgcmt_rqmin, on every level.
 * One launch sequence without a host round trip, replayed as a HIP graph from its second call. */
int mgcmt_vcycle_rqmg(mgcmt_plan* plan, int slot, const int* vecs, int nu1, int nu2, int robust, double* rho_out, void* stream);
/* dst = sum_t coeffs[t] * (slots[t], vecs[t]), 1 <= nterms <= 4; dst may be one of the inputs (the updates
 * x <- x + delta p, MGCMTSolver.py:52, and the residual A x - rho M x, :22-23, in one pass each) */
int mgcmt_lincomb(mgcmt_plan* plan, int level, int nterms, const double* coeffs, const int* slots, const int* vecs, int dst_slot,
                  int dst_vec, void* stream);
int mgcmt_scale(mgcmt_plan* plan, int level, double alpha, int slot, int vec, void* stream);
/* Block operations of the blocked Rayleigh-Ritz eigen-solver (the 2-vector problem of rqmin, MGCMTSolver.py:44-50, carried
 * to blocks of trial vectors; SURVEY par. 8(f)4).  block_gram: out[i * nb + j] = <A_i, B_j> for na <= 12 and nb <= 4
 * vectors in one pass, synchronises.  block_combine: OUT_j = sum_i coeffs[i * nout + j] IN_i for nin <= 12 inputs and
 * nout <= 4 distinct outputs, every input read once; an output may be one of the inputs. */
int mgcmt_block_gram(mgcmt_plan* plan, int level, int na, const int* a_slots, const int* a_vecs, int nb, const int* b_slots,
                     const int* b_vecs, double* out, void* stream);
int mgcmt_block_combine(mgcmt_plan* plan, int level, int nin, const int* in_slots, const int* in_vecs, int nout, const int* out_slots,
                        const int* out_vecs, const double* coeffs, void* stream);
/* in-place Gram-Schmidt of vectors 0..k-1 of `slot`: modified != 0 -> MGS (:44-50), else CGS (:34-42) */
int mgcmt_gramschmidt(mgcmt_plan* plan, int level, int slot, int k, int modified, void* stream);
/* columns scaled to unit 2-norm (normalize, :52-63) */
int mgcmt_normalize(mgcmt_plan* plan, int level, int slot, int k, void* stream);

/* Building blocks of a sharded cycle (multigridcmt_amd/distributed.py exchanges halo rows between them).
 * mgcmt_fused_pass: ONE fused row-streaming pass on `level`, V <- nsweep sweeps of `kind` (MGCMT_WJACOBI or
 * MGCMT_GS_MC) applied to V, optionally preceded by V += P V[level+1] (mode 1, MGCMTSolver.py:323-324) or
 * followed by F[level+1] <- R (F - (A - mu I) V) (mode 2, :315); adding 4 to mode 0 or 2 declares the incoming V
 * to be zero (it is then neither read nor required to have been cleared, :316); adding 8 to mode 2 suppresses the
 * store of the smoothed V ("recompute instead of store": a later mode-1 pass with (npre << 4) added re-runs those
 * npre sweeps from the untouched V before it adds the correction — also with 4 when that V is the zero iterate).
 * The pass reads the exchanged halo rows (mgcmt_plan_level_halo) of V
 * and F (and of V[level+1] in mode 1) around a strip.  mgcmt_fused_max_sweeps: sweeps one pass can take on that
 * level (0 = the level is not covered by the fused kernels). */
int mgcmt_fused_pass(mgcmt_plan* plan, int level, int kind, int nsweep, double omega, int mode, int k, void* stream);
int mgcmt_fused_max_sweeps(const mgcmt_plan* plan, int level, int kind, int* max_sweeps);
/* how many pre-smoothing sweeps a mode-1 pass with `nsweep` post-smoothing sweeps can recompute on that level */
int mgcmt_fused_max_recompute(const mgcmt_plan* plan, int level, int kind, int nsweep, int* max_recompute);
/* How the library recognised the operator of `level` (which decides the kernels the level runs on): MGCMT_OPK_GENERAL
 * (Kronecker terms with variable factors), MGCMT_OPK_FIVE_POINT (constant 5-point / 3-point: the scaled, shifted
 * Laplacian of MGCMTStencilMaker.py:15-25), MGCMT_OPK_FIVE_DIAG (the same plus a product potential on the diagonal),
 * MGCMT_OPK_NINE_CONST (Galerkin coarsenings R*A*P, MGCMTSolver.py:318, of a constant operator), MGCMT_OPK_NINE_VAR
 * (the same plus one term with variable factors: the coarsened product potential). */
#define MGCMT_OPK_GENERAL 0
#define MGCMT_OPK_FIVE_POINT 1
#define MGCMT_OPK_FIVE_DIAG 2
#define MGCMT_OPK_NINE_CONST 3
#define MGCMT_OPK_NINE_VAR 4
int mgcmt_level_operator_kind(const mgcmt_plan* plan, int level, int* kind);

/* ---- multi-GPU: row strips with neighbour halo exchange --------------------------------------------
 * A strip plan (mgcmt_plan_desc.row_begin/row_end/strip_levels) gets a communicator; every exchange below is then
 * enqueued by the library itself.  mgcmt_comm_init: RCCL (ncclSend/ncclRecv groups and ncclAllGather on HIP streams,
 * no host synchronisation inside a cycle; librccl is loaded on first use).  Rank 0 obtains the 128-byte id with
 * mgcmt_comm_unique_id and the host program distributes it (any channel).  mgcmt_comm_init_external: the host
 * program supplies the transport as callbacks (gloo, MPI, staging through host memory ...); the library synchronises
 * its stream before each call, `ptr`s are device addresses of this plan, counts are in doubles, a callback returns 0 on
 * success and must have completed the transfer when it returns. */
#define MGCMT_UNIQUE_ID_BYTES 128
typedef struct mgcmt_p2p_op {
  void* ptr;
  int64_t count;
  int32_t peer;
  int32_t is_send;
} mgcmt_p2p_op;
typedef int (*mgcmt_p2p_fn)(void* user, int nops, const mgcmt_p2p_op* ops);                  /* one batch, all at once */
typedef int (*mgcmt_allgather_fn)(void* user, const void* send, void* recv, int64_t count);  /* recv = nranks * count */
typedef int (*mgcmt_allreduce_fn)(void* user, double* host_inout, int n);                    /* sum over ranks */
int mgcmt_comm_unique_id(void* id_out /* MGCMT_UNIQUE_ID_BYTES */);
int mgcmt_comm_init(mgcmt_plan* plan, int rank, int nranks, const void* unique_id);
int mgcmt_comm_init_external(mgcmt_plan* plan, int rank, int nranks, mgcmt_p2p_fn p2p, mgcmt_allgather_fn allgather,
                             mgcmt_allreduce_fn allreduce, void* user);
int mgcmt_comm_destroy(mgcmt_plan* plan);
typedef enum mgcmt_comm_option {
  MGCMT_COMM_OPT_OVERLAP = 0, /* default 1 (RCCL): the exchange of a pass's boundary rows runs on a second stream beside
                                 the launch that produces the interior rows */
  MGCMT_COMM_OPT_SPLIT = 1,   /* default 1: boundary rows are produced by their own launches first on strips of >= 2^22 points (where the
                                 exchange they take off the critical path outweighs two more launches); 2: on every strip; 0: never */
  MGCMT_COMM_OPT_SELF_RING = 2, /* default 0; 1 on a ONE-rank communicator: the rank acts as its own upper and lower neighbour in
                                 every exchange of a cycle: the split launches, the transport and the stream overlap run
                                 for real on one GPU (on a whole-grid plan the rows it receives are never read) */
  MGCMT_COMM_OPT_EMULATE_OF = 3 /* default 0; N > 1 on a ONE-rank communicator in self-ring mode whose plan is the strip of
                                 some rank R of an N-rank job: the cycle then does that rank's work — strip passes, the
                                 exchanges (with itself), a gather of N strips' worth of data, the redundant coarse
                                 sub-cycle — so that one GPU can time one rank's share (bench.py --emulate-rank).  The halo
                                 rows hold the rank's own rows, so the numbers are not the N-rank job's */
} mgcmt_comm_option;
int mgcmt_comm_set_option(mgcmt_plan* plan, int option, int value);
/* halo rows (the exchanged ones, mgcmt_plan_level_halo) of the slots in slot_mask (bit s = slot s) on strip level `level` <-
 * the chain neighbours' boundary rows, one batch; vectors 0..k-1 with k in bits 16-23 of slot_mask (0 = vector 0 only).
 * Adding 0x100 on a ONE-rank communicator treats the strip as a ring (it is its own upper and lower neighbour): a
 * self-test of the transport. */
int mgcmt_halo_exchange(mgcmt_plan* plan, int level, int slot_mask, void* stream);
/* strips of vectors 0..k-1 of (level, slot) of all ranks -> the whole-grid finest level of `coarse` (dst_slot) on every rank */
int mgcmt_gather_coarse(mgcmt_plan* plan, int level, int slot, mgcmt_plan* coarse, int dst_slot, int k, void* stream);
/* host_inout[0..n) <- sum over ranks (norms, inner products: the np.dot / np.linalg.norm call sites); synchronises */
int mgcmt_allreduce_sum(mgcmt_plan* plan, double* host_inout, int n, void* stream);
/* one V(nu1,nu2) cycle (MGCMTSolver.py:281-329) of the sharded hierarchy on vectors 0..k-1 (each with its own shift):
 * `plan` holds the finest levels as row strips, `coarse` the first whole-grid level and everything below it on every
 * rank.  flags: the caller vouches that the halo rows of the fine level's V (nothing but mgcmt_sharded_vcycle wrote V since
 * the previous cycle) / F (since the previous cycle with the same right-hand side) are still the neighbours' rows, which
 * saves their exchange; MGCMT_SHARDED_GRAM_SCHMIDT: vcycle_matrix (:375-436) — modified Gram-Schmidt of the k columns on
 * every non-coarsest level on the way up (:434), on strip levels with ONE all-reduce of the column's coefficients per
 * column (SURVEY par. 8e). */
#define MGCMT_SHARDED_V_HALO_VALID 1
#define MGCMT_SHARDED_F_HALO_VALID 2
#define MGCMT_SHARDED_GRAM_SCHMIDT 4
int mgcmt_sharded_vcycle(mgcmt_plan* plan, mgcmt_plan* coarse, int nu1, int nu2, int nu_coarse, int kind, double omega, int k,
                         int flags, void* stream);

/* ---- general sparse, complex128 operators -------------------------------------------------------
 * The k.p Hamiltonians of ThesisProblem.py:38-40,80,101 / PotWellSolver.py:54-233: an arbitrary square CSR matrix of
 * complex numbers (values = interleaved re, im), cycled as ONE 1-D grid of its full length with the reference's 1-D
 * transfer operators (MGCMTStencilMaker.py:27-78).  The plan builds the Galerkin hierarchy R*A*P (MGCMTSolver.py:318)
 * on the device; vectors are complex (interleaved), slots V, F, T as for mgcmt_plan.  n and lowest are powers of two,
 * lowest <= 64 (the coarsest level is solved by a dense LU with row pivoting, :305-308). */
typedef struct mgcmt_csr_plan mgcmt_csr_plan;
int mgcmt_csr_plan_create(int device, int64_t n, int64_t lowest, const int64_t* indptr, const int32_t* indices, const double* values,
                          mgcmt_csr_plan** out);
int mgcmt_csr_plan_destroy(mgcmt_csr_plan* plan);
int mgcmt_csr_num_levels(const mgcmt_csr_plan* plan, int* levels);
/* rows, entries and the chunk length of the lexicographic sweeps (rows solved together as a first-order recurrence) */
int mgcmt_csr_level_info(const mgcmt_csr_plan* plan, int level, int64_t* n, int64_t* nnz, int32_t* lex_chunk_rows);
/* host copy of a level's matrix (indptr[n+1], indices[nnz], values[2*nnz]): what R*A*P gave on the device */
int mgcmt_csr_get_matrix(const mgcmt_csr_plan* plan, int level, int64_t* indptr, int32_t* indices, double* values);
int mgcmt_csr_upload(mgcmt_csr_plan* plan, int level, int slot, const double* re_im, int64_t count, void* stream);
int mgcmt_csr_download(mgcmt_csr_plan* plan, int level, int slot, double* re_im, int64_t count, void* stream);
/* dst <- (A_level - shift I) src */
int mgcmt_csr_apply(mgcmt_csr_plan* plan, int level, int src_slot, int dst_slot, double shift, void* stream);
/* nu sweeps of MGCMT_WJACOBI / MGCMT_GS_LEX / MGCMT_SOR_LEX on V, F of `level` for (A_level - shift I) */
int mgcmt_csr_smooth(mgcmt_csr_plan* plan, int level, int kind, int nu, double omega, double shift, void* stream);
/* one V-cycle (MGCMTSolver.py:281-329) from level 0; nu_coarse: sweeps below the top level (the reference: 4, :320) */
int mgcmt_csr_vcycle(mgcmt_csr_plan* plan, int nu1, int nu2, int nu_coarse, int kind, double omega, double shift, void* stream);

/* plan options: MGCMT_OPT_FUSED (default 1) selects the fused row-streaming kernels on large constant-
 * coefficient levels; 0 forces the one-launch-per-operation kernels everywhere (A/B checks) */
typedef enum mgcmt_option {
  MGCMT_OPT_FUSED = 0,
  MGCMT_OPT_FUSED_ROWS = 1, /* tuning: rows per wave chunk of this plan's fused passes, 0 = auto */
  MGCMT_OPT_RECOMPUTE = 3,  /* default 1: on levels of >= 2^22 points down-leg passes do not store the pre-smoothed iterate and
                               up-leg passes recompute it; 2: on every fused level; 0: never */
  MGCMT_OPT_GRAPH = 2,      /* default 1: mgcmt_vcycle replays its launch sequence as a HIP graph from the second call on */
  MGCMT_OPT_LEX_WAVE = 5,   /* lexicographic Gauss-Seidel / SOR sweeps of constant-coefficient 2-D levels of >= 16 x 16 points run as a
                               pipeline of waves over the whole chip: 2 = bands of 63 rows swept as a wavefront
                               (kernels_lexband.hip), 1 = skewed blocks of 64 columns with a scan per row (kernels_lexwave.hip);
                               0: one workgroup per vector everywhere */
  MGCMT_OPT_LEX_CHAIN = 6,  /* default 1: the nu Gauss-Seidel sweeps of a smoothing step run chained in ONE launch of the scan pipeline
                               (sweep s + 1 follows sweep s a few rows behind); 0: one launch per sweep.  Same arithmetic, same bits */
  MGCMT_OPT_MGS_BLOCK = 7,  /* default 1: modified Gram-Schmidt of 2..12 columns too long for one workgroup (a value > 1: of at least
                               that many points) as two passes over the data (Gram matrix,
                               its Cholesky factor R, Q = A R^-1 — what MGCMTProcessor.py:44-50 computes in exact arithmetic),
                               with the column-by-column kernels taking over on the device where the columns' condition
                               number would let the difference (cond^2 eps) show; 0: column by column always */
  MGCMT_OPT_TAIL = 4        /* default 1: the 2-D levels of at most 32 x 32 points below a cycle's top level, coarse solve
                               included, run as ONE launch (needs MGCMT_OPT_FUSED; not with Gram-Schmidt): a dense product
                               with the tail's matrix — the sub-cycle is linear in its right-hand side for fixed shift,
                               smoother and sweep counts; the matrix is formed once per shift set by running the tail on
                               the unit vectors —; 2: the LDS-resident launch of ~45 barrier-separated phases itself (the
                               dense form's arithmetic in another summation order); 0: one launch per pass */
} mgcmt_option;
int mgcmt_plan_set_option(mgcmt_plan* plan, int option, int value);

/* timing of the dominant kernel for bench.py: runs `reps` fine-level smoother sweeps between two
 * HIP events on `stream` and returns the elapsed milliseconds */
int mgcmt_time_smoother(mgcmt_plan* plan, int level, int kind, int nu, double omega, int reps, double* ms_out,
                        void* stream);

/* the same for ONE fused pass of the cycle's timed region (mode as mgcmt_fused_pass: e.g. 2|8 = the fine level's
 * restrict-without-store down pass, 1|(npre<<4) = its recompute + correct + post-smooth up pass): average milliseconds
 * of `reps` launches between two HIP events on `stream` (one untimed launch first) */
int mgcmt_time_fused_pass(mgcmt_plan* plan, int level, int kind, int nsweep, double omega, int mode, int reps, double* ms_out,
                          void* stream);

/* diagnostics of the last lexicographic wave-pipeline sweep (kernels_lexwave.hip): out[0] = blocks started, out[1] = error
 * word (a block timed out), and — only in a build with -DMGCMT_LEXWAVE_DEBUG — per block four words {ticks of the 100 MHz
 * clock spent in the block, rows, slow-path entries, start tick}. */
int mgcmt_lex_wave_stats(mgcmt_plan* plan, uint32_t* out, int64_t capacity);

/* empirical HBM ceilings for bench.py: streams the plan's level-`level` vectors (slots V, F -> T) with a
 * plain grid-stride kernel; kind 0 copy (16 B/point), 1 triad (24 B/point), 2 read-only (8 B/point); kinds 3/4/5 use
 * the fused kernels' access pattern instead (128-column windows marching down `blocks` rows): read 1 stream (8 B),
 * read 2 (16 B), read 2 + write 1 (24 B); kinds 6/7/8 the same with overlapping, unaligned windows (124 of 128 kept),
 * 9/10/11 with the fused kernels' own geometry (112 of 128 kept: line-aligned stores, loads straddling half lines),
 * 12/13/14 with 96 of 128 kept (everything line-aligned), 15/16/17 with 120 of 128 (64-byte-aligned); kinds 20..23 are
 * issue-rate probes — `blocks` workgroups of ONE wave each run 20000 trips of 64 instructions: a chain of dependent double
 * FMAs (20), eight independent chains (21), dependent 32-bit vector adds (22), dependent scalar adds (23); returns the
 * average milliseconds per launch */
int mgcmt_bandwidth_probe(mgcmt_plan* plan, int level, int kind, int blocks, int reps, double* ms_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MGCMT_HIP_H */
