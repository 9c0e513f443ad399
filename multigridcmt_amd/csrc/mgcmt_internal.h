// Internal declarations shared by the kernel files and the plan/driver of libmgcmt_hip.so.
// Not part of the ABI (the ABI is include/mgcmt_hip.h).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mgcmt_hip.h"

namespace mgcmt {

constexpr int kHalo = MGCMT_HALO_ROWS;
constexpr int kMaxTerms = MGCMT_MAX_TERMS;
constexpr int kMaxVec = 32;  // upper bound on simultaneous vectors of one launch

// Operator of one level as the kernels see it:  A = sum_m X_m (x) Y_m  (shift applied separately).
// X[m] / Y[m] point at element 0 of the `lower` array; `diag` is at +ldx, `upper` at +2*ldx.
// Row factors are stored with the level's halo count of entries before element 0 and after element nr-1 (the local
// strip's halo rows; zero outside the global grid), column factors span the whole row.
struct KOp {
  int nterms;
  int five_point;  // 1: c0/cn/cw below are valid (constant 5-point / 3-point operator)
  const double* X[kMaxTerms];
  const double* Y[kMaxTerms];
  long ldx, ldy;
  double c0, cn, cw;  // centre, row-neighbour, column-neighbour coefficient (five_point only)
  // nine_const: every factor is Toeplitz except its last diagonal entry (Galerkin levels of a constant
  // operator): interior coefficients c9[di+1][dj+1], own-row coefficients on the last row, centre-column
  // coefficients on the last column, and the corner's diagonal
  int nine_const;
  double c9[3][3], c9row[3], c9col[3], c9corner;
  // nine_var: some terms are Toeplitz-but-last (their sum is c9 / c9row / c9col / c9corner as above), ONE term has
  // arbitrary tridiagonal factors vX, vY ([lower | diag | upper], leading dimensions ldx / ldy) — the Galerkin levels of
  // a constant operator plus a product potential (square well, PotWellSolver.py:150-153 carried to 2-D)
  int nine_var;
  const double* vX;
  const double* vY;
  // five_diag: a constant 5-point part (c0/cn/cw) plus ndiag (1 or 2) terms whose factors are both diagonal — a
  // product potential p(i) q(j) on top of a scaled Laplacian (square well): dX[m][row], dY[m][col] are the diagonals
  int five_diag, ndiag;
  const double* dX[2];
  const double* dY[2];
  // 1-D levels: all terms folded into ONE tridiagonal  T = sum_m x_m Y_m  (X_m is 1 x 1): tri = [lower | diag | upper], n
  // numbers each; tri_const: Toeplitz except for its last diagonal entry (the Laplacian and its Galerkin coarsenings)
  int one_d, tri_const;
  const double* tri;
  double t_lo, t_di, t_up, t_last;
};

// A batch of vectors on one level: interior pointer of vector 0, elements between vectors.
struct KVec {
  double* p;
  long stride;
};

struct KGrid {
  long nr, nc;       // local rows (strip), columns
  int coarsen_rows;  // 1 for 2-D (rows are coarsened too), 0 for 1-D
};

// ---- kernel launchers (kernels_*.hip) --------------------------------------------------------
// All take the number of vectors k (grid z) and a device pointer to the k shifts.

void launch_apply(hipStream_t s, KGrid g, KOp op, KVec src, KVec dst, const double* shifts, int k);
// <x,x>, <x,w>, <w,w>, <x,A w>, <w,A w> in one pass over x and w (nothing stored): true when the level's operator is
// covered (2-D 5-point, with or without a product potential); out[0..4] on the device, partials: 5 * 8192 doubles
bool launch_ritz_pair(hipStream_t s, KGrid g, KOp op, const double* x, const double* w, double* partials, double* out);
void launch_final_sums(hipStream_t s, int nq, int nblocks, const double* partials, double* out);
void launch_wjacobi(hipStream_t s, KGrid g, KOp op, KVec vin, KVec f, KVec vout, const double* shifts, double omega, int k);
void launch_mc_colour(hipStream_t s, KGrid g, KOp op, KVec v, KVec f, const double* shifts, double omega, int ca, int cb, int k);
void launch_residual(hipStream_t s, KGrid g, KOp op, KVec v, KVec f, KVec r, const double* shifts, int k);
void launch_restrict(hipStream_t s, KGrid fine, KGrid coarse, KVec r, KVec rc, int k);
void launch_prolong(hipStream_t s, KGrid fine, KGrid coarse, KVec e, KVec v, int accumulate, int k);
// generalised lexicographic sweep (in place):
//   v_k <- (alpha d_k v_k + beta f_k - wU sum_{j>k} a_kj v_j - wL sum_{j<k} a_kj v_j^new) / d_k
void launch_lex_sweep(hipStream_t s, KGrid g, KOp op, KVec v, KVec f, const double* shifts, double alpha, double beta,
                      double wU, double wL, int k);

// the same sweep as a pipeline of waves over the whole chip (kernels_lexwave.hip; constant-coefficient operators on
// grids of at least 16 x 16 points).  carry: k * lex_wave_blocks(g) * g.nr * 4 eight-byte granules; sync: 2 words; sync[1]
// != 0 after the sweep: a block gave up waiting (reported by the next synchronising call)
bool lex_wave_supported(const KGrid& g, const KOp& op);
long lex_wave_blocks(const KGrid& g);
long lex_wave_carry(const KGrid& g, int k, int nsweeps);  // granules of `carry` for k vectors and nsweeps chained sweeps
void launch_lex_wave(hipStream_t s, KGrid g, KOp op, KVec v, KVec f, const double* shifts, double alpha, double beta, double wU,
                     double wL, int k, double* carry, unsigned* sync, int nsweeps, double gamma);
// ... and as a wavefront over bands of 63 rows (kernels_lexband.hip; same operators and grids).  carry: k *
// lex_band_count(g) * lex_band_stride(g) eight-byte granules; sync as above
long lex_band_count(const KGrid& g);
long lex_band_stride(const KGrid& g);
void launch_lex_band(hipStream_t s, KGrid g, KOp op, KVec v, KVec f, const double* shifts, double alpha, double beta, double wU,
                     double wL, int k, double* carry, unsigned* sync);

// fused row-streaming passes (fused_kernel.h, kernels_fused*.hip): vin -> vout with nsweep sweeps of weighted
// Jacobi or multicolour Gauss-Seidel.  mode & 3: 0 plain, 1 prolong+correct first (coarse = correction), 2
// residual+restrict last (coarse = right-hand side); mode & 4: vin is zero; mode & 8: vout is not written (mode 2);
// npre: pre-smoothing sweeps recomputed in front of the correction (mode 1)
bool fused_supported(const KGrid& g, const KOp& op);
// the 1-D form (kernels_fused1d.hip): a wave takes a window of 128 points through every stage in registers
bool fused1d_supported(const KGrid& g, const KOp& op);
int fused1d_max_sweeps(int multicolour);
int fused1d_max_recompute(int multicolour, int nsweep);
void launch_fused1d(hipStream_t s, KGrid g, KOp op, KVec vin, KVec f, KVec vout, KVec coarse, long coarse_nc, const double* shifts,
                    double omega, int multicolour, int nsweep, int mode, int npre, int k);
int fused_max_sweeps(const KOp& op, int multicolour);
int fused_max_recompute(const KOp& op, int multicolour, int nsweep);
// rows_override: rows per wave chunk (0 = automatic); [out_lo, out_hi): the local rows this launch produces (even
// bounds; the whole strip is 0 .. g.nr) — a sharded pass runs its boundary rows first so that their exchange overlaps
void launch_fused(hipStream_t s, KGrid g, KOp op, KVec vin, KVec f, KVec vout, KVec coarse, long coarse_nc, const double* shifts,
                  double omega, int multicolour, int nsweep, int mode, int npre, long row_lo, long row_hi, long last_row, int k,
                  long rows_override = 0, long out_lo = 0, long out_hi = -1, long out_lo2 = 0, long out_hi2 = 0);

// vector algebra; scalar results / inputs live in device memory so nothing syncs with the host
void launch_fill(hipStream_t s, double* p, long n, double value);
// y += (alpha_scale * alpha_dev[0] / (den_dev ? den_dev[0] : 1)) * x ;  x /= (use_sqrt ? sqrt(s_dev[0]) : s_dev[0])
void launch_axpy_dev(hipStream_t s, long n, const double* alpha_dev, const double* den_dev, double alpha_scale, const double* x, double* y);
void launch_scale_dev(hipStream_t s, long n, const double* s_dev, int use_sqrt, double* x);
void launch_axpy(hipStream_t s, long n, double alpha, const double* x, double* y);
void launch_scale(hipStream_t s, long n, double alpha, double* x);
// streaming probes: kind 0 copy (a -> out), 1 triad (a + s b -> out), 2 read-only (a)
void launch_probe(hipStream_t s, int kind, long n, const double* a, const double* b, double* out, int blocks);
// the fused kernels' access pattern alone (128-column windows marching down `rows` rows): kind 3 read 1 stream,
// 4 read 2 streams, 5 read 2 + write 1; 6/7/8: the same with overlapping unaligned windows (124 of 128 columns kept)
void launch_probe_march(hipStream_t s, long nr, long nc, long rows, int nstreams, int do_write, int wout, const double* a, const double* b, double* out);
// out[q] = <x, y_q> for q < nq (y_q = y + q*ystride), deterministic two-pass reduction; `partials`
// holds at least nq * reduce_blocks(n) doubles
int reduce_blocks(long n);
void launch_dots(hipStream_t s, long n, const double* x, const double* y, long ystride, int nq, double* partials, double* out);
// Gram-Schmidt building blocks that consume the first reduction pass directly (no separate final pass / no scalar
// round trip): x /= sqrt(sum partials);  a_j -= (<q,a_j>/<q,q>) q for nj columns in one launch
void launch_dot_partials(hipStream_t s, long n, const double* x, const double* y, long ystride, int nq, double* partials, const double* gate = nullptr);
void launch_scale_by_norm(hipStream_t s, long n, const double* partials, double* x);
void launch_project_out(hipStream_t s, long n, const double* partials, const double* q, double* a_first, long astride, int nj);
constexpr int kGramMaxVectors = 6;
constexpr int kBlockMaxA = 12, kBlockMaxB = 4;  // block operations: up to 12 x 4 vectors (three blocks of four trial vectors)
void launch_probe_issue(hipStream_t s, int what, int iters, int blocks, double* sink);  // kinds 20..23 of mgcmt_bandwidth_probe
void launch_block_gram(hipStream_t s, long n, const double* const* a, int na, const double* const* b, int nb, double* partials, double* out);
void launch_block_combine(hipStream_t s, long n, const double* const* in, int nin, double* const* out, int nout, const double* c);
void launch_gram(hipStream_t s, long n, const double* const* v, int nv, double* partials, double* out);
void launch_lincomb(hipStream_t s, long n, const double* const* v, const double* c, int nt, double* dst);
bool mgs_small_fits(long n);

// Rayleigh-quotient minimisation as two passes per step (kernels_rq.hip); state: rq_state_words() doubles on the device,
// partials: at least 8 * 4096 doubles.  init: 1 = the initial pair (p = 0: rho and g of the start vector), 2 = the first
// step (p = -g, p_old not read), 0 = a step.  launch_rq_pass1 ends with the step's scalars (delta, rho of x + delta p);
// launch_rq_pass2 returns the number of partial sums per result for launch_rq_scalars2 (rho, beta), which the caller
// runs after <g, M g> is in state[rq_word_gmg()] when M is not the identity.
int rq_state_words();
int rq_word_rho();
int rq_word_gmg();
void launch_rq_pass1(hipStream_t s, KGrid g, KOp A, KOp Mo, int m_identity, const double* x, const double* gv, const double* pold, double* pnew,
                     double* state, int init, int robust, double* partials);
int launch_rq_pass2(hipStream_t s, KGrid g, KOp A, KOp Mo, int m_identity, const double* x, const double* p, double* xnew, double* gout, double* state,
                    int init, double* partials);
bool launch_rq_gmg(hipStream_t s, KGrid g, KOp Mo, const double* gv, double* partials, int nblocks);
// the whole rqmin call (initial pair + nu steps) in one single-workgroup launch where the level is small enough; false: not taken
bool launch_rq_small(hipStream_t s, KGrid g, KOp A, KOp Mo, int m_identity, double* x, double* p, double* gv, double* state, int nu, int robust);
// m_identity: 1 = M is the identity, 0 = <g, M g> is in state[rq_word_gmg()], 2 = it is result 3 of the partial sums (launch_rq_gmg)
void launch_rq_scalars2(hipStream_t s, const double* partials, int nblocks, double* state, int m_identity, int init);

// the levels of at most 32 x 32 points of a V-cycle in one launch (kernels_tail.hip)
constexpr int kTailMaxLevels = 6;
struct TailArgs {
  int g0, nlev, nterms;       // entry grid (g0 x g0), number of levels, Kronecker terms per level
  const double* X[kTailMaxLevels][kMaxTerms];
  const double* Y[kTailMaxLevels][kMaxTerms];
  long ldx[kTailMaxLevels], ldy[kTailMaxLevels];
  const double* f_in;         // entry level right-hand side (vector 0), vstride between vectors
  double* v_out;              // entry level result
  long vstride;
  const double* inv;          // explicit inverse of the coarsest (A - mu I), one n x n block per vector
  long inv_stride;
  const double* shifts;
  double omega;
  int kind, nu;               // smoother and sweeps (pre and post) on every tail level
  int unit_rhs, unit_q;       // launch_tail_matrix: block b runs on the unit vector e_b with vector unit_q's shift and inverse
};
bool tail_fits(long g0, int nlev, int nterms);
void launch_tail(hipStream_t s, const TailArgs& a, int k);
// The tail is linear in its right-hand side for fixed (shift, smoother, nu, omega): its matrix, formed once per shift set
// by running the tail on the unit vectors, turns the ~45 barrier-separated phases into one dense product.
bool tail_dense_fits(long g0);
void launch_tail_matrix(hipStream_t s, TailArgs a, int q, double* mt);
void launch_tail_dense(hipStream_t s, long g0, const double* mt, long mt_stride, const double* f_in, double* v_out, long vstride, int k);
void launch_mgs_small(hipStream_t s, long n, double* a0, long stride, int k);
// nb_in: partial sums per result in partials_in (0 = reduce_blocks(n), what the previous step left; 1 = already summed)
void launch_mgs_step(hipStream_t s, long n, const double* partials_in, double* u, long stride, int m, double* partials_out, int nb_in = 0,
                     const double* gate = nullptr);
// modified Gram-Schmidt of up to mgs_block_max() long columns in two passes over the data (Gram matrix, its factor, Q = A R^-1),
// with a gate word (cf[mgs_block_gate_word()]: 0 = done) for the column-by-column launches behind it: kernels_blas.hip
int mgs_block_max();
int mgs_block_words();
int mgs_block_gate_word();
void launch_mgs_blocked(hipStream_t s, long n, double* a0, long stride, int k, double* partials, double* cf);
// its three launches one by one (a sharded plan all-reduces the Gram matrix between the first two): per-block partial sums
// (returns the number of blocks), the factor from `nblocks` partial sums per entry, Q = A R^-1
int launch_mgs_gram(hipStream_t s, long n, const double* a0, long stride, int k, double* partials);
void launch_mgs_factor(hipStream_t s, const double* partials, int nblocks, int k, double* cf);
void launch_mgs_apply(hipStream_t s, long n, double* a0, long stride, int k, const double* cf);

// banded LU of (A - mu I) on the coarsest level and its solves (one workgroup per vector)
struct KBand {
  long n;        // unknowns
  int kl;        // half bandwidth
  int width;     // stored entries per row: 3*kl + 1
  double* ab;    // [k][n][width]
  int* piv;      // [k][n]
  long ab_stride, piv_stride;
};
void launch_band_assemble(hipStream_t s, KGrid g, KOp op, const double* shifts, KBand b, int k);
void launch_band_factor(hipStream_t s, KBand b, int k);
void launch_band_solve(hipStream_t s, KBand b, KVec rhs, KVec x, int k);
// explicit inverse (n <= 1024 unknowns) and the dense solve x = inv * rhs that replaces the substitutions
void launch_band_invert(hipStream_t s, KBand b, double* inv, long inv_stride, int k);
void launch_dense_solve(hipStream_t s, long n, const double* inv, long inv_stride, KVec rhs, KVec x, int k);

}  // namespace mgcmt
