// Row-streaming fused kernels — the V-cycle's hot kernels on large 2-D levels: constant 5-point
// operators (the scaled / shifted Laplacian of MGCMTStencilMaker.py:15-25) and the separable 9-point
// Galerkin operators R*A*P of the coarser levels (MGCMTSolver.py:318).
//
// One launch does, in a single pass over the level (read v, read f, write v'):
//     [prolong + correct]  ->  NSWEEP smoothing sweeps  ->  [residual + full-weighting restriction]
// i.e. MGCMTSolver.py:323-326 or :313-315 fused.  Algorithmic traffic per fine point: 24 B for the
// pass (+2 B for the coarse array), whatever NSWEEP is, instead of 24 B per sweep plus 2 x 18 B for
// the transfers.
//
// Mapping (wave64, no LDS, no barriers): a wave owns a window of 128 columns (2 per lane, 16-byte
// loads/stores) and marches down a chunk of rows_per_chunk rows.  Every sweep is a pipeline stage
// that lags one row behind the previous one; a stage keeps a three-row window of its input in
// registers and gets the lateral neighbours from the adjacent lanes (wave shuffles).  Values at the
// wave's edges become invalid one column per stage, so a window overlaps its neighbours by `halo`
// columns per side and by S(+1) rows at the chunk ends; those are recomputed redundantly (they come
// from L2, not HBM: vertically adjacent chunks are scheduled back-to-back on the same XCD).  The pass
// is out of place (v -> v'), so neighbouring windows never see half-updated data.
//
// Stages: weighted Jacobi = one stage per sweep; red-black Gauss-Seidel = a red ((i+j) odd) and a
// black stage per sweep — the same update order as the multicolour smoother of kernels_stencil.hip
// restricted to a 5-point operator.  Rows/columns outside the global grid are held at zero
// (Dirichlet ghosts), rows in [row_lo,row_hi) outside the chunk belong to neighbours.
#pragma once
#include "mgcmt_internal.h"

namespace mgcmt {

namespace fused {

#ifndef MGCMT_FUSED_DEPTH
#define MGCMT_FUSED_DEPTH 3
#endif
#ifndef MGCMT_FUSED_WAVES
#define MGCMT_FUSED_WAVES 1
#endif
constexpr int kDepth = MGCMT_FUSED_DEPTH;         // rows per prefetch batch (two batches of registers); 1, 3 or 6
constexpr int kWavesPerBlock = MGCMT_FUSED_WAVES;
#ifndef MGCMT_FUSED_DEPTH9
#define MGCMT_FUSED_DEPTH9 1
#endif
constexpr int kDepth9 = MGCMT_FUSED_DEPTH9;  // 9-point policies carry wider windows: shallower batches keep two or three waves per SIMD
// Register sets of prefetched rows in rotation: one is consumed while the other NS - 1 are in flight.  The 9-point
// passes have single-row batches, and with two sets a wave has ONE row of loads outstanding while it works — on the
// Galerkin levels that is a memory-latency bound (waves waiting 50-60 % of their cycles); more sets of one row cost
// 6 registers each (a right-hand-side pair and a correction value) instead of the 3 x 12 of a deeper batch.
#ifndef MGCMT_FUSED_SETS9
#define MGCMT_FUSED_SETS9 2
#endif
constexpr int kSets9 = MGCMT_FUSED_SETS9;
// the largest number of sets <= want that divides the batches of a loop body (the rotation must close at the latch)
constexpr int sets_for(int batches, int want) {
  int n = want < 2 ? 2 : want;
  while (batches % n != 0) --n;
  return n;
}

struct FusedArgs {
  const double* vin;
  const double* f;
  double* vout;
  const double* ec;  // coarse correction to interpolate and add first (PROLONG)
  double* rc;        // coarse right-hand side written last (RESTRICT)
  long nr, nc;       // local rows / columns
  long row_lo, row_hi;  // local rows that lie inside the global grid
  long vstride, cstride;
  long cnc;
  double c0, cn, cw;                 // constant 5-point operator
  double c9[3][3], c9row[3], c9col[3], c9corner;  // Galerkin levels of a constant operator (Op9c)
  long last_row;                     // local index of the global last row
  const double* X[kMaxTerms];        // separable 9-point operator: row / column factors (KOp layout)
  const double* Y[kMaxTerms];
  long ldx, ldy;
  const double* shifts;
  double omega;
  int n_row_chunks, n_col_groups;
  int rows_per_chunk;  // rows a wave marches over (even), plus the overlap
  int out_lo, out_hi;  // local rows this launch produces (even bounds; 0 .. nr for a whole pass)
  int out_lo2, out_hi2;  // a second row range produced by the same launch (a strip's two edges in one launch); empty: out_lo2 == out_hi2
  int n_row_chunks1;   // chunks of the first range (the rest cover the second)
  int rows_override;   // tuning: rows per chunk, 0 = automatic (host side only)
  int xcd_balanced;    // workgroup -> tile mapping: 1 = equal shares of the tile list per XCD, 0 = whole column groups per XCD
};

#ifndef MGCMT_FUSED_NT_STORE
#ifdef __HIP__
#define MGCMT_FUSED_NT_STORE 1  // measured: 2-4 % faster passes; the written vector is not re-read by this pass
#else
#define MGCMT_FUSED_NT_STORE 0
#endif
#endif
#ifndef MGCMT_FUSED_NT_F
#define MGCMT_FUSED_NT_F 0
#endif
#if MGCMT_FUSED_NT_STORE || MGCMT_FUSED_NT_F
typedef double v2d_t __attribute__((ext_vector_type(2)));
#endif

// 16-byte accesses; the streamed-once operands optionally bypass the caches' retention (non-temporal)
__device__ __forceinline__ double2 load2(const double* p) { return *reinterpret_cast<const double2*>(p); }
__device__ __forceinline__ double2 load2_stream(const double* p) {
#if MGCMT_FUSED_NT_F
  const v2d_t t = __builtin_nontemporal_load(reinterpret_cast<const v2d_t*>(p));
  return make_double2(t.x, t.y);
#else
  return load2(p);
#endif
}
__device__ __forceinline__ void store2_stream(double* p, double x, double y) {
#if MGCMT_FUSED_NT_STORE
  v2d_t t;
  t.x = x;
  t.y = y;
  __builtin_nontemporal_store(t, reinterpret_cast<v2d_t*>(p));
#else
  *reinterpret_cast<double2*>(p) = make_double2(x, y);
#endif
}

constexpr int round_even(int x) { return (x + 1) & ~1; }

// Neighbour-lane reads of the marching loop.  MGCMT_FUSED_DPP = 1: wave-wide DPP shifts (two v_mov_b32_dpp per
// double, no LDS round trip; a lane without a neighbour reads zero).  = 0: ds_bpermute through two precomputed
// lane addresses (a lane without a neighbour reads itself).  The two differ only in lanes 0 / 63, whose columns lie
// in the window overlap and are never stored.
#ifndef MGCMT_FUSED_DPP
#define MGCMT_FUSED_DPP 1
#endif
__device__ __forceinline__ double lane_fetch_addr(int byte_addr, double v) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_ds_bpermute(byte_addr, (int)(unsigned)u);
  const unsigned hi = (unsigned)__builtin_amdgcn_ds_bpermute(byte_addr, (int)(unsigned)(u >> 32));
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
#if MGCMT_FUSED_DPP && defined(__HIP_DEVICE_COMPILE__)
template <int CTRL>
__device__ __forceinline__ double lane_shift(double v) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)u, CTRL, 0xf, 0xf, true);
  const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTRL, 0xf, 0xf, true);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
// value of the left neighbour (lane - 1): wave_shr:1; of the right neighbour (lane + 1): wave_shl:1; the _ROT forms
// wrap around the wave (wave_ror:1 / wave_rol:1; the address versions take (lane -/+ 1) mod 64)
#define MGCMT_FETCH_LEFT(addr, v) lane_shift<0x138>(v)
#define MGCMT_FETCH_RIGHT(addr, v) lane_shift<0x130>(v)
#define MGCMT_FETCH_LEFT_ROT(addr, v) lane_shift<0x13C>(v)
#define MGCMT_FETCH_RIGHT_ROT(addr, v) lane_shift<0x134>(v)
#else
#define MGCMT_FETCH_LEFT(addr, v) lane_fetch_addr(addr, v)
#define MGCMT_FETCH_RIGHT(addr, v) lane_fetch_addr(addr, v)
#define MGCMT_FETCH_LEFT_ROT(addr, v) lane_fetch_addr(addr, v)
#define MGCMT_FETCH_RIGHT_ROT(addr, v) lane_fetch_addr(addr, v)
#endif

// Arithmetic note: multiply-adds are written as explicit fma() (fewer VALU issues; measured -10 % on a 4096^2
// cycle) rather than left to -ffp-contract, so that every instantiation rounds identically: the "recompute instead
// of store" passes must reproduce the stored values bit for bit.

// ---- operator policies -------------------------------------------------------------------------
// A policy evaluates, for one of the lane's two columns (COL 0 = ja, 1 = ja+1), the off-diagonal sum,
// the diagonal of (A - mu I) and its reciprocal from the 3 x 3 neighbourhood n / c / s = row above /
// own row / row below, each ordered west, centre, east.

// constant 5-point operator: c0 on the diagonal, cn for the row neighbours, cw for the column ones
struct Op5 {
  static constexpr bool kNine = false;
  static constexpr bool kBigBody = false;
  double d, invd, cn, cw;
  __device__ __forceinline__ void init(const FusedArgs& a, int q, long, long) {
    d = a.c0 - a.shifts[q];
    invd = 1.0 / d;
    cn = a.cn;
    cw = a.cw;
  }
  static constexpr int kRowValues = 0;  // per-row operator data staged through LDS (none)
  static constexpr bool kSpecialRow = false;
  __device__ __forceinline__ double fetch_row(const FusedArgs&, long, int) const { return 0.0; }
  __device__ __forceinline__ void set_row(const FusedArgs&, int, const double*) {}
  __device__ __forceinline__ bool special_row(int) const { return false; }
  template <int COL>
  __device__ __forceinline__ void eval(const double* n, const double* c, const double* s, double& off, double& dg, double& inv) const {
    off = fma(cn, n[1] + s[1], cw * (c[0] + c[2]));
    dg = d;
    inv = invd;
  }
  template <int COL>
  __device__ __forceinline__ void fix_special(const double*, double&, double&, double&) const {}
};

// Galerkin coarsenings of a constant operator: every factor is Toeplitz except its last diagonal entry
// (MGCMTSolver.py:318 with the one-sided P/R of MGCMTStencilMaker.py:27-78), so the 9 coefficients are
// constants with corrections on the last row, the last column and the corner.  No coefficient loads: the interior
// coefficients are wave-uniform (scalar registers); the last COLUMN only changes the centre-column coefficients of
// the lane that owns it, which are per-lane values set up once; the last ROW is one marching step per stage and is
// patched by a wave-uniform branch (fix_special).
struct Op9c {
  static constexpr bool kNine = true;
  static constexpr bool kBigBody = false;
  static constexpr bool kSpecialRow = true;
  double cnw, cn_, cne, cw_, ce_, csw, cs_, cse;  // interior (uniform)
  double vnb, vsb;                                  // north / south coefficients of column ja+1 (per lane)
  double dga, dgb, inva, invb;                      // diagonal of (A - mu I) and its reciprocal: column ja (uniform), ja+1 (per lane)
  double rw, re;                                    // last row: additions to the west / east coefficients
  double dgra, dgrb, invra, invrb;                  // last row: diagonals
  int last_row_index;
  __device__ __forceinline__ void init(const FusedArgs& a, int q, long ja, long nc) {
    cnw = a.c9[0][0]; cn_ = a.c9[0][1]; cne = a.c9[0][2];
    cw_ = a.c9[1][0]; ce_ = a.c9[1][2];
    csw = a.c9[2][0]; cs_ = a.c9[2][1]; cse = a.c9[2][2];
    const double mu = a.shifts[q];
    const bool lastb = ja + 1 == nc - 1;
    vnb = lastb ? a.c9col[0] : cn_;
    vsb = lastb ? a.c9col[2] : cs_;
    dga = a.c9[1][1] - mu;
    dgb = (lastb ? a.c9col[1] : a.c9[1][1]) - mu;
    inva = 1.0 / dga;
    invb = 1.0 / dgb;
    rw = a.c9row[0] - cw_;
    re = a.c9row[2] - ce_;
    dgra = a.c9row[1] - mu;
    dgrb = (lastb ? a.c9corner : a.c9row[1]) - mu;
    invra = 1.0 / dgra;
    invrb = 1.0 / dgrb;
    last_row_index = (int)a.last_row;
  }
  static constexpr int kRowValues = 0;
  __device__ __forceinline__ double fetch_row(const FusedArgs&, long, int) const { return 0.0; }
  __device__ __forceinline__ void set_row(const FusedArgs&, int, const double*) {}
  __device__ __forceinline__ bool special_row(int row) const { return row == last_row_index; }
  template <int COL>
  __device__ __forceinline__ void eval(const double* n, const double* c, const double* s, double& off, double& dg, double& inv) const {
    const double vn = COL == 1 ? vnb : cn_, vs = COL == 1 ? vsb : cs_;
    off = fma(cse, s[2], fma(vs, s[1], fma(csw, s[0], fma(ce_, c[2], fma(cw_, c[0], fma(cne, n[2], fma(vn, n[1], cnw * n[0])))))));
    dg = COL == 1 ? dgb : dga;
    inv = COL == 1 ? invb : inva;
  }
  template <int COL>
  __device__ __forceinline__ void fix_special(const double* c, double& off, double& dg, double& inv) const {
    off = fma(re, c[2], fma(rw, c[0], off));
    dg = COL == 1 ? dgrb : dgra;
    inv = COL == 1 ? invrb : invra;
  }
};

// 1 / d for the variable-coefficient policy, where it is needed per point and stage: the hardware estimate refined
// by two Newton steps (full double accuracy to the last bit or two) instead of the ~12-instruction IEEE division.
__device__ __forceinline__ double fast_reciprocal(double d) {
#if defined(__HIP_DEVICE_COMPILE__)
  double r = __builtin_amdgcn_rcp(d);
  r = fma(fma(-d, r, 1.0), r, r);
  r = fma(fma(-d, r, 1.0), r, r);
  return r;
#else
  return 1.0 / d;
#endif
}

// general separable 9-point operator  sum_m X_m (x) Y_m  (variable coefficients: potentials, PotWellSolver.py:150-153
// style wells): the lane keeps the column factors of its two columns in registers.  The row factors (3M numbers per
// row, the same for every lane) travel with the prefetched rows — lane k < 3M loads value k of the row — and are
// parked in a small LDS ring indexed by the row, from which every stage reads the row it is updating (LDS reads are
// counted by lgkmcnt, so they never stall the vmcnt-ordered stream of row loads).
template <int M>
struct Op9 {
  static constexpr bool kNine = true;
  static constexpr bool kBigBody = true;  // its 12-step loop bodies are too long to exist twice (checked + steady state)
  static constexpr bool kSpecialRow = false;
  static constexpr int kRowValues = 3 * M;
  double mu;
  double ya[M][3], yb[M][3];  // lower, diag, upper of Y_m at columns ja, ja+1
  double x[M][3];             // lower, diag, upper of X_m at the row being updated
  __device__ __forceinline__ void init(const FusedArgs& a, int q, long ja, long nc) {
    mu = a.shifts[q];
    const long j = ja < 0 ? 0 : (ja > nc - 2 ? nc - 2 : ja);
#pragma unroll
    for (int m = 0; m < M; ++m)
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        ya[m][p] = a.Y[m][p * a.ldy + j];
        yb[m][p] = a.Y[m][p * a.ldy + j + 1];
      }
  }
  // value `lane` (= 3*m + part) of row `rl` (already clamped into the allocation)
  __device__ __forceinline__ double fetch_row(const FusedArgs& a, long rl, int lane) const {
    const int k = lane < 3 * M ? lane : 3 * M - 1;
    return a.X[k / 3][(k % 3) * a.ldx + rl];
  }
  __device__ __forceinline__ void set_row(const FusedArgs&, int, const double* ring_row) {
#pragma unroll
    for (int m = 0; m < M; ++m)
#pragma unroll
      for (int p = 0; p < 3; ++p) x[m][p] = ring_row[3 * m + p];
  }
  __device__ __forceinline__ bool special_row(int) const { return false; }
  template <int COL>
  __device__ __forceinline__ void fix_special(const double*, double&, double&, double&) const {}
  template <int COL>
  __device__ __forceinline__ void eval(const double* n, const double* c, const double* s, double& off, double& dg, double& inv) const {
    double o = 0.0, dd = 0.0;
#pragma unroll
    for (int m = 0; m < M; ++m) {
      const double* y = COL == 0 ? ya[m] : yb[m];
      const double rn = fma(y[2], n[2], fma(y[1], n[1], y[0] * n[0]));
      const double rc = fma(y[2], c[2], y[0] * c[0]);
      const double rs = fma(y[2], s[2], fma(y[1], s[1], y[0] * s[0]));
      o = fma(x[m][2], rs, fma(x[m][1], rc, fma(x[m][0], rn, o)));
      dd = fma(x[m][1], y[1], dd);
    }
    off = o;
    dg = dd - mu;
    inv = fast_reciprocal(dg);
  }
};

// A constant part as in Op9c (Toeplitz-but-last terms: the Galerkin coarsenings of the scaled Laplacian) plus ONE term
// X (x) Y with arbitrary tridiagonal factors (the coarsened product potential of a square well): the eight constant
// off-diagonal coefficients stay scalars, the variable term costs 11 multiply-adds and the diagonal one more — 29
// against the 39 + 3 of treating all three terms as variable (Op9<3>), which is what the Galerkin levels of BASELINE
// config 5 ran on (vector-ALU bound: 0.93 of the 1.46 ms of an 8192^2 cycle).  Column factors of the lane's two
// columns in registers, the row's three factors through the LDS ring like Op9's.
struct Op9cv {
  static constexpr bool kNine = true;
  static constexpr bool kBigBody = false;
  static constexpr bool kSpecialRow = true;
  static constexpr int kRowValues = 3;
  Op9c base;
  double ya[3], yb[3];  // lower, diag, upper of Y at columns ja, ja+1
  double x[3];          // lower, diag, upper of X at the row being updated
  __device__ __forceinline__ void init(const FusedArgs& a, int q, long ja, long nc) {
    base.init(a, q, ja, nc);
    const long j = ja < 0 ? 0 : (ja > nc - 2 ? nc - 2 : ja);
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      ya[p] = a.Y[0][p * a.ldy + j];
      yb[p] = a.Y[0][p * a.ldy + j + 1];
      x[p] = 0.0;
    }
  }
  __device__ __forceinline__ double fetch_row(const FusedArgs& a, long rl, int lane) const {
    const int k = lane < 3 ? lane : 2;
    return a.X[0][k * a.ldx + rl];
  }
  __device__ __forceinline__ void set_row(const FusedArgs&, int, const double* ring_row) {
#pragma unroll
    for (int p = 0; p < 3; ++p) x[p] = ring_row[p];
  }
  __device__ __forceinline__ bool special_row(int row) const { return base.special_row(row); }
  template <int COL>
  __device__ __forceinline__ void eval(const double* n, const double* c, const double* s, double& off, double& dg, double& inv) const {
    double o, dd, unused;
    base.template eval<COL>(n, c, s, o, dd, unused);
    const double* y = COL == 0 ? ya : yb;
    const double rn = fma(y[2], n[2], fma(y[1], n[1], y[0] * n[0]));
    const double rc = fma(y[2], c[2], y[0] * c[0]);
    const double rs = fma(y[2], s[2], fma(y[1], s[1], y[0] * s[0]));
    off = fma(x[2], rs, fma(x[1], rc, fma(x[0], rn, o)));
    dg = fma(x[1], y[1], dd);
    inv = fast_reciprocal(dg);
  }
  // the last row of the constant part (its X factors' last diagonal entry differs); the variable term's row factors
  // come from the arrays and need no patch
  template <int COL>
  __device__ __forceinline__ void fix_special(const double* c, double& off, double& dg, double& inv) const {
    double unused;
    base.template fix_special<COL>(c, off, dg, unused);
    dg = fma(x[1], COL == 0 ? ya[1] : yb[1], dg);
    inv = fast_reciprocal(dg);
  }
};

// constant 5-point operator plus MD product potentials p_m(i) q_m(j) on the diagonal (a square well on a scaled
// Laplacian, PotWellSolver.py:150-153 carried to 2-D): the off-diagonal part is Op5's three scalars, the lane keeps
// q_m of its two columns, p_m of a row travels through the LDS ring like Op9's row factors.
template <int MD>
struct Op5V {
  static constexpr bool kNine = false;
  static constexpr bool kBigBody = false;
  static constexpr bool kSpecialRow = false;
  static constexpr int kRowValues = MD;
  double d0, cn, cw;
  double qa[MD], qb[MD], p[MD];
  __device__ __forceinline__ void init(const FusedArgs& a, int q, long ja, long nc) {
    d0 = a.c0 - a.shifts[q];
    cn = a.cn;
    cw = a.cw;
    const long j = ja < 0 ? 0 : (ja > nc - 2 ? nc - 2 : ja);
#pragma unroll
    for (int m = 0; m < MD; ++m) {
      qa[m] = a.Y[m][j];
      qb[m] = a.Y[m][j + 1];
      p[m] = 0.0;
    }
  }
  __device__ __forceinline__ double fetch_row(const FusedArgs& a, long rl, int lane) const { return a.X[lane < MD ? lane : MD - 1][rl]; }
  __device__ __forceinline__ void set_row(const FusedArgs&, int, const double* ring_row) {
#pragma unroll
    for (int m = 0; m < MD; ++m) p[m] = ring_row[m];
  }
  __device__ __forceinline__ bool special_row(int) const { return false; }
  template <int COL>
  __device__ __forceinline__ void eval(const double* n, const double* c, const double* s, double& off, double& dg, double& inv) const {
    off = fma(cn, n[1] + s[1], cw * (c[0] + c[2]));
    double d = d0;
#pragma unroll
    for (int m = 0; m < MD; ++m) d = fma(p[m], COL == 0 ? qa[m] : qb[m], d);
    dg = d;
    inv = fast_reciprocal(d);
  }
  template <int COL>
  __device__ __forceinline__ void fix_special(const double*, double&, double&, double&) const {}
};

enum { kJacobi = 0, kRedBlack = 1, kFourColour = 2 };

// FLAGS of a pass: what is fused around the NSWEEP smoothing sweeps
enum {
  kProlong = 1,   // V += P V[coarse] first (after the recomputed pre-smoothing, if any)
  kRestrict = 2,  // F[coarse] = R (F - (A - mu I) V') last
  kZeroIn = 4,    // the incoming V is identically zero: it is not read (nor needs clearing)
  kNoStore = 8,   // do not write V' (only the restricted residual is wanted; the up-leg recomputes V')
  kPreShift = 4   // bits 4-5: NPRE sweeps recomputed in front of the correction (kProlong only)
};

constexpr int stages_of(int kind, int sweeps) { return kind == kJacobi ? sweeps : (kind == kRedBlack ? 2 * sweeps : 4 * sweeps); }

template <class OP, int KIND, int NSWEEP, int FLAGS>
struct FusedShape {
  static constexpr bool PROLONG = (FLAGS & kProlong) != 0, RESTRICT = (FLAGS & kRestrict) != 0;
  static constexpr int NPRE = (FLAGS >> kPreShift) & 3;
  static constexpr int SPRE = stages_of(KIND, NPRE);           // recomputed pre-smoothing stages
  static constexpr int S = SPRE + stages_of(KIND, NSWEEP);     // pipeline stages
  static constexpr int E = RESTRICT ? 1 : 0;
  // Lateral overlap of neighbouring windows.  Values go stale one column per stage from the window edges inwards
  // (plus one for an interpolated correction, plus two for the restriction), so S + 2 columns would do; the overlap
  // is a fixed 8 columns instead (16 for the passes with more than six stages), because then every wave STORES 112
  // (96) columns = seven (six) whole, 128-byte-aligned cache lines.  Measured with the access-pattern probe
  // (mgcmt_bandwidth_probe kinds 5 / 8): windows whose stores start and end inside cache lines run a
  // 2-read-1-write stream at 4.2 TB/s, line-aligned ones at 5.5 TB/s.
  static constexpr int need = S + (RESTRICT ? 2 : (PROLONG ? 1 : 0));
  // A pass that does not store V' has no store alignment to protect: with few stages it overlaps by 4 columns only
  // (probe kinds 10 / 16: two read streams run at 5.3 TB/s with 112 of 128 columns kept, 5.7 with 120).
#ifndef MGCMT_FUSED_NARROW_NOSTORE
#define MGCMT_FUSED_NARROW_NOSTORE 1
#endif
  static constexpr int halo = (MGCMT_FUSED_NARROW_NOSTORE && (FLAGS & kNoStore) != 0 && need <= 4) ? 4 : (need <= 8 ? 8 : 16);
  static constexpr int wout = 128 - 2 * halo;
  static_assert(need <= halo, "too many pipeline stages for the window overlap");
  // The colour smoothers update in place: ONE rotating window of `body` rows serves all their stages (a stage only
  // changes its own colour, and points of one colour are never neighbours, so reading a row that the same stage
  // has already passed is harmless).  Weighted Jacobi needs every stage's input intact: one 3-row window per stage.
  static constexpr bool inplace = KIND != kJacobi;
  // ... except for the interpolated correction, which changes every colour of a row: the stages behind it run one
  // row later (XL), so that the row is corrected after the last pre-correction stage has read it
  static constexpr int XL = (inplace && SPRE > 0) ? 1 : 0;
  static constexpr int body = (inplace && S + E + 2 + XL > 6) ? 12 : 6;  // marching steps per loop iteration
  static_assert(!inplace || S + E + 2 + XL <= body, "in-place window too short");
};

// kZeroIn: the error equation on a coarser level starts from zero (MGCMTSolver.py:316).
// kNoStore + NPRE: "recompute instead of store" — the down-leg pass writes only the restricted residual, the
// up-leg pass re-runs the same pre-smoothing sweeps from the untouched V before it adds the correction
// (identical arithmetic, hence identical values): 8 B per point less written and 8 B less read per level.
//
// Instruction economy (these kernels keep the vector ALUs 40-60 % busy, so instructions are time):
//  * the marching loop body is six steps long and every step knows its position T in it at compile time, so the
//    three-row stage windows and the right-hand-side delay line are ROTATING register files indexed by T — a row
//    is written once and never moved;
//  * no per-stage masks: columns outside the grid carry omega = 0 (their value stays at the zero it was loaded
//    as), rows outside the grid skip the stage under a wave-uniform branch;
//  * row/column indices are 32-bit (wave-uniform tests stay on the scalar unit), only row offsets are 64-bit and
//    advance incrementally.
constexpr int mod3(int x) { return ((x % 3) + 3) % 3; }
constexpr int modn(int x, int n) { return ((x % n) + n) % n; }
template <int N>
struct StepIndex {
  static constexpr int value = N;
};
template <bool B>
struct Checked {
  static constexpr bool value = B;
};
// f(StepIndex<T0>{}), ..., f(StepIndex<T0 + N - 1>{})
template <int T0, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (N > 0) {
    f(StepIndex<T0>{});
    static_for<T0 + 1, N - 1>(f);
  }
}

// (Asking the compiler for more waves per SIMD through the second launch-bound argument was measured: with 3 or 4 it
// spills the windows to scratch and the cycle takes 2-3x as long.  Second session, with the real allocations known
// (§4.1 of DESIGN.md): the four-colour 9-point passes sit at 176 - 184 registers = 2 waves per SIMD, and asked for 3
// they come out at 168 with 12 registers spilled — 8192^2 pass 0.306 -> 0.341 ms, 4096^2 red-black cycle 0.402 ->
// 0.474 ms.  The register counts the compiler picks on its own are right.)
template <class OP, int KIND, int NSWEEP, int FLAGS>
__global__ void __launch_bounds__(64 * kWavesPerBlock) k_fused(FusedArgs a) {
  using Shape = FusedShape<OP, KIND, NSWEEP, FLAGS>;
  constexpr int S = Shape::S, E = Shape::E, HALO = Shape::halo, WOUT = Shape::wout, SPRE = Shape::SPRE;
  constexpr bool PROLONG = Shape::PROLONG, RESTRICT = Shape::RESTRICT;
  constexpr bool ZERO_IN = (FLAGS & kZeroIn) != 0, STORE_V = (FLAGS & kNoStore) == 0;
  constexpr bool NINE = OP::kNine;
  constexpr int WN = NINE ? S + E : 1;  // windows that also keep the lateral neighbours
  constexpr int D = NINE ? kDepth9 : kDepth;  // rows per prefetch batch
  constexpr bool INPLACE = Shape::inplace;
  constexpr int XL = Shape::XL;
  constexpr int B = Shape::body;              // marching steps per loop iteration
  constexpr int NS = NINE ? sets_for(B / D, kSets9) : 2;  // register sets of D prefetched rows each
  static_assert(B % D == 0 && NS >= 2 && (B / D) % NS == 0, "the loop body must hold a whole number of rotations of the prefetch sets");
  constexpr int FL = S + E + 1 + XL;          // delay of the right-hand side between its load and its last use
  constexpr bool FRING = FL <= B;             // short enough for a rotating file of B; otherwise a shifting one
  constexpr int FN = FRING ? B : FL;

  // workgroup -> (column group, row chunk).  Blocks b and b+8 run on the same XCD (own L2): an XCD gets a
  // CONTIGUOUS range of column groups, so the half cache lines two neighbouring windows share (their 8-column
  // overlap is 64 bytes on each side) and the overlap rows of vertically adjacent chunks come from that L2.
#ifdef MGCMT_FUSED_XCD_INTERLEAVED
  const int b = blockIdx.x;
  const int xcd = b & 7, seq = b >> 3;
  const int chunk = seq % a.n_row_chunks;
  const int group = xcd + 8 * (seq / a.n_row_chunks);
#else
  const int b = blockIdx.x;
  const int xcd = b & 7, seq = b >> 3;
  int group, chunk;
  if (a.xcd_balanced) {
    // Every XCD gets the same number of (column group, row chunk) items (+-1): a contiguous range of the chunk-major
    // list, i.e. whole rows of chunks plus a partial one at either end, so neighbours in a row still share its L2.
    // With whole column groups per XCD the last XCDs run short on narrow levels (19 groups: 3,3,3,3,3,3,1,0) and the
    // launch lasts as long as the fullest XCD: 2048^2 passes -19 % (Jacobi) / -39 % (four-colour), 4096^2 -11 %,
    // 8192^2 -6 % with this mapping; at 16384^2 (147 groups: 3 % imbalance) the column bands are 6 % faster and stay —
    // also against column bands with equal shares (the group at either end of a band split by rows between two XCDs,
    // walked chunk by chunk: 16384^2 cycle 2.98 -> 3.07 ms).
    const long total = (long)a.n_col_groups * a.n_row_chunks;
    const long lo = xcd * total / 8, hi = (xcd + 1) * total / 8;
    if (lo + seq >= hi) return;
    const int item = (int)(lo + seq);
    chunk = item / a.n_col_groups;
    group = item - chunk * a.n_col_groups;
  } else {
    const int per_xcd = (a.n_col_groups + 7) >> 3;
    group = xcd * per_xcd + seq % per_xcd;
    chunk = seq / per_xcd;
    if (seq % per_xcd + xcd * per_xcd >= a.n_col_groups || chunk >= a.n_row_chunks) return;
  }
#endif
  if (group >= a.n_col_groups) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lane_up = (lane > 0 ? lane - 1 : 0) << 2, lane_dn = (lane < 63 ? lane + 1 : 63) << 2;  // left / right neighbour
  (void)lane_up;  // (only the ds_bpermute form of the lane reads takes the addresses)
  (void)lane_dn;
  const int strip = group * kWavesPerBlock + wave;
  const int nc = (int)a.nc, cnc = (int)a.cnc;
  const int row_lo = (int)a.row_lo, row_hi = (int)a.row_hi;
  if (strip * WOUT >= nc) return;  // wave-uniform
  const int q = blockIdx.y;

  const int ja = strip * WOUT - HALO + 2 * lane;  // this lane's columns: ja (even), ja + 1
  const bool col_in = ja >= 0 && ja < nc;
  const bool col_out = col_in && 2 * lane >= HALO && 2 * lane < HALO + WOUT;
  const int jc = ja >> 1;  // coarse column of the pair
  const bool ccol_in = jc >= 0 && jc < cnc;
  const double lanemask = col_in ? 1.0 : 0.0;    // multiplies what is loaded for columns outside the grid
  const double cmask = ccol_in ? 1.0 : 0.0;
  const double omega = col_in ? a.omega : 0.0;   // ... and they are never updated: Dirichlet ghosts stay zero

  const double* __restrict__ vin = a.vin + q * a.vstride;
  const double* __restrict__ fin = a.f + q * a.vstride;
  double* __restrict__ vout = a.vout + q * a.vstride;
  const double* __restrict__ ec = PROLONG ? a.ec + q * a.cstride : nullptr;
  double* __restrict__ rc = RESTRICT ? a.rc + q * a.cstride : nullptr;

  const bool second = chunk >= a.n_row_chunks1;  // (wave-uniform)
  const int r_begin = second ? a.out_lo2 + (chunk - a.n_row_chunks1) * a.rows_per_chunk : a.out_lo + chunk * a.rows_per_chunk;
  const int range_hi = second ? a.out_hi2 : a.out_hi;
  const int r_end = r_begin + a.rows_per_chunk < range_hi ? r_begin + a.rows_per_chunk : range_hi;
  const int rstart = r_begin - (S + E);
  const int rstop = r_end + S + XL + 2 * E;  // rows [rstart, rstop) are marched over
  // Dependency cone of the chunk's outputs: stage s is only NEEDED on rows >= cone0 + s (its products on the rows
  // before that feed nothing the chunk stores — the march starts S + E rows early for stage 0's sake, and the later a
  // stage, the more of those early rows it can skip: S^2 stage-rows per chunk, more than the useful ones on the short
  // chunks of small levels).  The end of the march needs no such test: it stops when the last stage is done.
  const int cone0 = r_begin - (S - 1) - (RESTRICT ? 1 : 0);

  auto row_ok = [&](int row) { return row >= row_lo && row < row_hi; };

  // Loads are unconditional (addresses clamped into the allocation, values masked when they are
  // consumed): with no branch around a load the compiler's vmcnt bookkeeping stays exact.
  const int ja_ld = ja < 0 ? 0 : (ja > nc - 2 ? nc - 2 : ja);
  const int jc_ld = jc < 0 ? 0 : (jc > cnc - 1 ? cnc - 1 : jc);
  const int crow_lo = (row_lo >> 1) - 1, crow_hi = (row_hi - 1) >> 1;  // coarse rows that may be read

  OP op;
  op.init(a, q, ja, nc);

  // NS register sets of D rows each, used in rotation: one is consumed while the others are in flight (refilled
  // in order), so every load has (NS - 1) * D rows of work between its issue and its use.
  struct Row {
    double2 v, f;
    double e;
    double xr;  // one of the row's operator values (policies with kRowValues > 0)
  };
  constexpr int RV = OP::kRowValues;
  constexpr int kRing = 16;  // rows of operator values kept in LDS; a stage lags at most S + E + 1 <= 10 rows
  __shared__ double s_ring[kWavesPerBlock][RV > 0 ? kRing * RV : 1];
  double* ring = s_ring[wave];
  Row sets[NS][D];
  // rows are fetched in order; the row is clamped into the allocation with min/max (no control flow near a load)
  int frow = rstart;
  auto fetch = [&](Row& r) __attribute__((always_inline)) {
    const int hi = row_hi - 1;
    const int t = frow < hi ? frow : hi;
    const int frl = t > row_lo ? t : row_lo;
    const long fbase = (long)frl * nc;
    if (ZERO_IN) r.v = make_double2(0.0, 0.0);
    else r.v = load2(vin + fbase + ja_ld);
    r.f = load2_stream(fin + fbase + ja_ld);
    r.e = 0.0;
    if (PROLONG) {
      const int I0 = frl >> 1;
      const int I1 = I0 < crow_hi ? I0 : crow_hi;
      const int I = I1 > crow_lo ? I1 : crow_lo;
      r.e = ec[I * cnc + jc_ld];
    }
    r.xr = RV > 0 ? op.fetch_row(a, frl, lane) : 0.0;
    ++frow;
  };
#pragma unroll
  for (int n = 0; n + 1 < NS; ++n)
#pragma unroll
    for (int u = 0; u < D; ++u) fetch(sets[n][u]);

  // stage windows: w[s] is the input of stage s+1; w[S] (RESTRICT) the input of the residual stage.  Three rows
  // each, rotating: at loop position T stage s finds the row above / the row it updates / the row below (just
  // arrived) in slots mod3(T-s+1) / mod3(T-s+2) / mod3(T-s).  wl / wr: values left of ja / right of ja+1.
  constexpr int NW = INPLACE ? 1 : S + E, NWN = INPLACE ? 1 : WN;
  double wa[NW][3], wb[NW][3], wl[NWN][3], wr[NWN][3];
  double fa[FN], fb[FN];
#pragma unroll
  for (int s = 0; s < NW; ++s)
#pragma unroll
    for (int r = 0; r < 3; ++r) wa[s][r] = wb[s][r] = 0.0;
#pragma unroll
  for (int s = 0; s < NWN; ++s)
#pragma unroll
    for (int r = 0; r < 3; ++r) wl[s][r] = wr[s][r] = 0.0;
  // the in-place window: row (row - k) lives in slot modn(T - k, B)
  constexpr int NR = INPLACE ? B : 1, NRN = (INPLACE && NINE) ? B : 1;
  double Wa[NR], Wb[NR], Wl[NRN], Wr[NRN];
#pragma unroll
  for (int r = 0; r < NR; ++r) Wa[r] = Wb[r] = 0.0;
#pragma unroll
  for (int r = 0; r < NRN; ++r) Wl[r] = Wr[r] = 0.0;
#pragma unroll
  for (int s = 0; s < FN; ++s) fa[s] = fb[s] = 0.0;

  // coarse correction values of the last SPRE+2 fine rows: er[k] belongs to fine row (row - k); el[k] is the
  // value of the lane to the left
  constexpr int NE = SPRE + 2 + XL;
  double er[NE], el[NE];
#pragma unroll
  for (int k = 0; k < NE; ++k) er[k] = el[k] = 0.0;
  if (PROLONG) {
    // coarse row of fine row rstart-1 (used when rstart is even); row (row_lo>>1)-1 is the halo row
    const int I = (rstart - 1) >> 1;
    if (ccol_in && I >= crow_lo && I <= crow_hi) er[0] = ec[I * cnc + jc];
    el[0] = MGCMT_FETCH_LEFT(lane_up, er[0]);
  }
  double racc = 0.0;  // running full-weighting sum of the current coarse row (RESTRICT)
  unsigned okbits = 0;  // bit k: row (row - k) lies inside the grid (wave-uniform shift register)
  // Row parities are compile-time: chunks start on even rows (rows_per_chunk is even, so are the strip bounds)
  // and the loop advances six rows per iteration, so the row of stage s at loop position T has the parity of
  // T - (s + 1) - (S + E).

  // one marching step at loop position T: consume the prefetched row `row`, run every stage one row further
  // CHK = false: every row any stage touches in this step lies inside the grid and inside the chunk's output
  // range (the steady state of a long march) — no row tests at all
  auto step = [&](auto pos, auto chk, const int row, const Row& in) __attribute__((always_inline)) {
    constexpr int T = decltype(pos)::value;
    constexpr bool CHK = decltype(chk)::value;
    if (RV > 0 && lane < RV) ring[(row & (kRing - 1)) * RV + lane] = in.xr;  // this row's operator values
    const bool rok = CHK ? row_ok(row) : true;
    if (CHK) okbits = (okbits << 1) | (rok ? 1u : 0u);
    double ina = 0.0, inb = 0.0;
    if (!ZERO_IN) {
      const double m = rok ? lanemask : 0.0;
      ina = in.v.x * m;
      inb = in.v.y * m;
    }

    if (PROLONG) {
#pragma unroll
      for (int k = NE - 1; k > 0; --k) {
        er[k] = er[k - 1];
        el[k] = el[k - 1];
      }
      er[0] = in.e * (rok ? cmask : 0.0);
      el[0] = MGCMT_FETCH_LEFT(lane_up, er[0]);
    }
    // V += P e on fine row r (values va, vb of this lane's columns): odd fine column takes c[J], even takes
    // (c[J-1] + c[J]) / 2; an even fine row takes the mean of coarse rows I-1 and I
    // (lag = row - r; the correction of a row outside the grid is built from zeros, so it needs no row test)
    auto correct = [&](int lag, double& va, double& vb) __attribute__((always_inline)) {
      const double e_cur = er[lag], e_prev = er[lag + 1];
      double ca = 0.5 * (el[lag] + e_cur), cb = e_cur;
      if (((T - lag - (S + E)) & 1) == 0) {
        ca = 0.5 * (0.5 * (el[lag + 1] + e_prev) + ca);
        cb = 0.5 * (e_prev + cb);
      }
      const double m = (!CHK || ((okbits >> lag) & 1u)) ? lanemask : 0.0;
      va = fma(m, ca, va);
      vb = fma(m, cb, vb);
    };
    if (PROLONG && SPRE == 0) correct(0, ina, inb);

    // right-hand side of row (row - k): slot modn(T - k, B) of the rotating file, or entry k of the shifting one
    if (FRING) {
      fa[T] = in.f.x;
      fb[T] = in.f.y;
    } else {
#pragma unroll
      for (int s = FN - 1; s > 0; --s) {
        fa[s] = fa[s - 1];
        fb[s] = fb[s - 1];
      }
      fa[0] = in.f.x;
      fb[0] = in.f.y;
    }

    if constexpr (INPLACE) {
      // ---- colour smoothers: one rotating window, updated in place --------------------------------------------
      constexpr int s0 = modn(T, B);
      Wa[s0] = ina;
      Wb[s0] = inb;
      if (NINE) {
        Wl[s0] = MGCMT_FETCH_LEFT(lane_up, inb);
        Wr[s0] = MGCMT_FETCH_RIGHT(lane_dn, ina);
      }
#pragma unroll
      for (int s = 0; s <= S; ++s) {
        if (s == S && !RESTRICT) break;
        const int lag = s + 1 + ((XL > 0 && s >= SPRE) ? XL : 0);  // this stage works on row (row - lag)
        const int cs = modn(T - lag + 1, B), cc = modn(T - lag, B), cn = modn(T - lag - 1, B);  // rows below / own / above
        const int cl = NINE ? cc : 0, nl = NINE ? cn : 0, sl = NINE ? cs : 0;
        const int rs = row - lag;
        const double ca = Wa[cc], cb = Wb[cc];
        const double fva = FRING ? fa[modn(T - lag, B)] : fa[lag], fvb = FRING ? fb[modn(T - lag, B)] : fb[lag];
        auto eval_a = [&](double& off, double& dg, double& inv) __attribute__((always_inline)) {
          if constexpr (NINE) {
            const double n[3] = {Wl[nl], Wa[cn], Wb[cn]}, c[3] = {Wl[cl], ca, cb}, so[3] = {Wl[sl], Wa[cs], Wb[cs]};
            op.template eval<0>(n, c, so, off, dg, inv);
            if (OP::kSpecialRow && CHK && op.special_row(rs)) op.template fix_special<0>(c, off, dg, inv);
          } else {
            const double left = MGCMT_FETCH_LEFT(lane_up, cb);
            const double n[3] = {0.0, Wa[cn], 0.0}, c[3] = {left, ca, cb}, so[3] = {0.0, Wa[cs], 0.0};
            op.template eval<0>(n, c, so, off, dg, inv);
          }
        };
        auto eval_b = [&](double& off, double& dg, double& inv) __attribute__((always_inline)) {
          if constexpr (NINE) {
            const double n[3] = {Wa[cn], Wb[cn], Wr[nl]}, c[3] = {ca, cb, Wr[cl]}, so[3] = {Wa[cs], Wb[cs], Wr[sl]};
            op.template eval<1>(n, c, so, off, dg, inv);
            if (OP::kSpecialRow && CHK && op.special_row(rs)) op.template fix_special<1>(c, off, dg, inv);
          } else {
            const double right = MGCMT_FETCH_RIGHT(lane_dn, ca);
            const double n[3] = {0.0, Wb[cn], 0.0}, c[3] = {ca, cb, right}, so[3] = {0.0, Wb[cs], 0.0};
            op.template eval<1>(n, c, so, off, dg, inv);
          }
        };
        if (s < S) {
          const int rs_par = (T - lag - (S + E)) & 1;  // parity of rs
          bool upd_a, upd_b;
          if (KIND == kRedBlack) {
            const bool red = (s & 1) == 0;
            upd_a = red == (rs_par != 0);  // column ja is even: it is red ((i+j) odd) iff the row is odd
            upd_b = !upd_a;
          } else {
            // colours (i%2, j%2) in the order (0,1),(1,0),(0,0),(1,1)
            const int c = s & 3;
            const int cra = (c == 1 || c == 3) ? 1 : 0, ccb = (c == 0 || c == 3) ? 1 : 0;
            const bool row_on = rs_par == cra;
            upd_a = row_on && ccb == 0;
            upd_b = row_on && ccb == 1;
          }
          // wave-uniform; outside the grid (a few steps of the first and last chunks) the value stays zero
          if ((upd_a || upd_b) && (!CHK || (((okbits >> lag) & 1u) != 0 && rs >= cone0 + s))) {
            op.set_row(a, rs, ring + (rs & (kRing - 1)) * RV);
            if (upd_a) {
              double off, dg, inv;
              eval_a(off, dg, inv);
              const double na = fma(omega, (fva - fma(dg, ca, off)) * inv, ca);
              Wa[cc] = na;
              if (NINE) Wr[cl] = MGCMT_FETCH_RIGHT(lane_dn, na);
            }
            if (upd_b) {
              double off, dg, inv;
              eval_b(off, dg, inv);
              const double nb = fma(omega, (fvb - fma(dg, cb, off)) * inv, cb);
              Wb[cc] = nb;
              if (NINE) Wl[cl] = MGCMT_FETCH_LEFT(lane_up, nb);
            }
          }
          if (PROLONG && SPRE > 0 && s == SPRE - 1) {
            // the recomputed pre-smoothing ends here: correct the row ABOVE the one just finished (nothing before
            // the correction reads it again)
            const int ck = modn(T - (lag + 1), B), ckl = NINE ? ck : 0;
            correct(lag + 1, Wa[ck], Wb[ck]);
            if (NINE) {
              Wl[ckl] = MGCMT_FETCH_LEFT(lane_up, Wb[ck]);
              Wr[ckl] = MGCMT_FETCH_RIGHT(lane_dn, Wa[ck]);
            }
          }
          if (s == S - 1) {
            const int rout = rs;
            if (STORE_V && col_out && (!CHK || (rout >= r_begin && rout < r_end))) store2_stream(vout + (long)rout * nc + ja, Wa[cc], Wb[cc]);
          }
        } else {
          // residual of row rs and its full-weighting restriction
          double ra = 0.0, rb = 0.0;
          if (!CHK || (((okbits >> lag) & 1u) != 0 && rs >= r_begin)) {
            op.set_row(a, rs, ring + (rs & (kRing - 1)) * RV);
            double offa, offb, dga, dgb, inva, invb;
            eval_a(offa, dga, inva);
            eval_b(offb, dgb, invb);
            ra = lanemask * (fva - fma(dga, ca, offa));
            rb = lanemask * (fvb - fma(dgb, cb, offb));
          }
          const double rnext = MGCMT_FETCH_RIGHT(lane_dn, ra);  // residual at column ja + 2
          const double h = 0.25 * ra + 0.5 * rb + 0.25 * rnext;
          if (((T - lag - (S + E)) & 1) == 0) {
            const int I = (rs >> 1) - 1;  // coarse row closed by fine row rs = 2I + 2
            if (col_out && ccol_in && (!CHK || (2 * I >= r_begin && 2 * I < r_end))) rc[I * cnc + jc] = racc + 0.25 * h;
            racc = 0.25 * h;
          } else {
            racc += 0.5 * h;
          }
        }
      }
    } else {
      // ---- weighted Jacobi: a pipeline of stages, each with its own three-row window ------------------------------
      double oa = ina, ob = inb;  // output of the previous stage = next input row
  #pragma unroll
      for (int s = 0; s <= S; ++s) {
        if (s == S && !RESTRICT) break;
        // the window of stage s (s == S: the residual stage) takes the row the previous stage has just finished
        const int sn = mod3(T - s), sa = mod3(T - s + 1), sc = mod3(T - s + 2);
        wa[s][sn] = oa;
        wb[s][sn] = ob;
        const int sw = NINE ? s : 0;
        if (NINE) {
          wl[sw][sn] = MGCMT_FETCH_LEFT(lane_up, ob);
          wr[sw][sn] = MGCMT_FETCH_RIGHT(lane_dn, oa);
        }
        const int rs = row - (s + 1);  // the row this stage completes now
        const double ca = wa[s][sc], cb = wb[s][sc];
        const double fva = FRING ? fa[modn(T - (s + 1), B)] : fa[s + 1], fvb = FRING ? fb[modn(T - (s + 1), B)] : fb[s + 1];
        // neighbourhood of column ja / ja+1, handed to the operator policy
        auto eval_a = [&](double& off, double& dg, double& inv) __attribute__((always_inline)) {
          if constexpr (NINE) {
            const double n[3] = {wl[sw][sa], wa[s][sa], wb[s][sa]}, c[3] = {wl[sw][sc], ca, cb}, so[3] = {wl[sw][sn], wa[s][sn], wb[s][sn]};
            op.template eval<0>(n, c, so, off, dg, inv);
            if (OP::kSpecialRow && CHK && op.special_row(rs)) op.template fix_special<0>(c, off, dg, inv);
          } else {
            const double left = MGCMT_FETCH_LEFT(lane_up, cb);
            const double n[3] = {0.0, wa[s][sa], 0.0}, c[3] = {left, ca, cb}, so[3] = {0.0, wa[s][sn], 0.0};
            op.template eval<0>(n, c, so, off, dg, inv);
          }
        };
        auto eval_b = [&](double& off, double& dg, double& inv) __attribute__((always_inline)) {
          if constexpr (NINE) {
            const double n[3] = {wa[s][sa], wb[s][sa], wr[sw][sa]}, c[3] = {ca, cb, wr[sw][sc]}, so[3] = {wa[s][sn], wb[s][sn], wr[sw][sn]};
            op.template eval<1>(n, c, so, off, dg, inv);
            if (OP::kSpecialRow && CHK && op.special_row(rs)) op.template fix_special<1>(c, off, dg, inv);
          } else {
            const double right = MGCMT_FETCH_RIGHT(lane_dn, ca);
            const double n[3] = {0.0, wb[s][sa], 0.0}, c[3] = {ca, cb, right}, so[3] = {0.0, wb[s][sn], 0.0};
            op.template eval<1>(n, c, so, off, dg, inv);
          }
        };

        if (s < S) {
          const int rs_par = (T - (s + 1) - (S + E)) & 1;  // parity of rs
          bool upd_a = true, upd_b = true;
          if (KIND == kRedBlack) {
            const bool red = (s & 1) == 0;
            const bool row_odd = rs_par != 0;
            upd_a = red == row_odd;  // column ja is even: it is red ((i+j) odd) iff the row is odd
            upd_b = !upd_a;
          } else if (KIND == kFourColour) {
            // colours (i%2, j%2) in the order (0,1),(1,0),(0,0),(1,1)
            const int c = s & 3;
            const int cra = (c == 1 || c == 3) ? 1 : 0, ccb = (c == 0 || c == 3) ? 1 : 0;
            const bool row_on = rs_par == cra;
            upd_a = row_on && ccb == 0;
            upd_b = row_on && ccb == 1;
          }
          double na = ca, nb = cb;
          // wave-uniform; outside the grid (a few steps of the first and last chunks) the value stays zero
          if ((upd_a || upd_b) && (!CHK || (((okbits >> (s + 1)) & 1u) != 0 && rs >= cone0 + s))) {
            op.set_row(a, rs, ring + (rs & (kRing - 1)) * RV);
            if (upd_a) {
              double off, dg, inv;
              eval_a(off, dg, inv);
              na = fma(omega, (fva - fma(dg, ca, off)) * inv, ca);
            }
            if (upd_b) {
              double off, dg, inv;
              eval_b(off, dg, inv);
              nb = fma(omega, (fvb - fma(dg, cb, off)) * inv, cb);
            }
          }
          oa = na;
          ob = nb;
          if (PROLONG && SPRE > 0 && s == SPRE - 1) correct(s + 1, oa, ob);  // the recomputed pre-smoothing ends here
          if (s == S - 1) {
            const int rout = row - S;
            if (STORE_V && col_out && (!CHK || (rout >= r_begin && rout < r_end))) store2_stream(vout + (long)rout * nc + ja, oa, ob);
          }
        } else {
          // residual of row rs and its full-weighting restriction
          double ra = 0.0, rb = 0.0;
          if (!CHK || (((okbits >> (s + 1)) & 1u) != 0 && rs >= r_begin)) {
            op.set_row(a, rs, ring + (rs & (kRing - 1)) * RV);
            double offa, offb, dga, dgb, inva, invb;
            eval_a(offa, dga, inva);
            eval_b(offb, dgb, invb);
            ra = lanemask * (fva - fma(dga, ca, offa));
            rb = lanemask * (fvb - fma(dgb, cb, offb));
          }
          const double rnext = MGCMT_FETCH_RIGHT(lane_dn, ra);  // residual at column ja + 2
          const double h = 0.25 * ra + 0.5 * rb + 0.25 * rnext;
          if (((T - (s + 1) - (S + E)) & 1) == 0) {
            const int I = (rs >> 1) - 1;  // coarse row closed by fine row rs = 2I + 2
            if (col_out && ccol_in && (!CHK || (2 * I >= r_begin && 2 * I < r_end))) rc[I * cnc + jc] = racc + 0.25 * h;
            racc = 0.25 * h;
          } else {
            racc += 0.5 * h;
          }
        }
      }
    }
  };

  // B steps per iteration in B/D batches: refill the set consumed last, then process the oldest one.
  auto body = [&](auto chk, const int base) __attribute__((always_inline)) {
    static_for<0, B / D>([&](auto g) __attribute__((always_inline)) {
      constexpr int G = decltype(g)::value;
      Row* cur = sets[G % NS];
      Row* nxt = sets[(G + NS - 1) % NS];
#pragma unroll
      for (int u = 0; u < D; ++u) fetch(nxt[u]);
      static_for<0, D>([&](auto u) __attribute__((always_inline)) {
        constexpr int U = decltype(u)::value;
        step(StepIndex<G * D + U>{}, chk, base + G * D + U, cur[U]);
      });
    });
  };
  // Iterations whose B steps need no row test: every stage row (down to base - (S+E+1)) inside the grid, every
  // output row (V': row - S, coarse F: row - S - 3) inside the chunk.  They form the middle of the march and run
  // in a loop of their own (an `if` inside one loop would make the two bodies meet at the loop latch, where the
  // compiler then copies the whole register state and drains the prefetch); the checked body serves the few
  // iterations before and after, the outer two-trip loop only exists so that its code is emitted once.
  constexpr bool kFastBody = !OP::kBigBody || B == 6;
  const int fast_lo = r_begin + S + XL + 3;
  int fast_hi = (r_end + S + XL < row_hi ? r_end + S + XL : row_hi) - (B - 1);
  // ... and, for policies that patch the operator's last row, no stage on that row: the steady-state body then has
  // no trace of the patch (left in, the compiler turns its uniform branch into selects on every evaluation)
  if (OP::kSpecialRow && fast_hi > (int)a.last_row - (B - 1)) fast_hi = (int)a.last_row - (B - 1);
  int base = rstart;
  if constexpr (kFastBody) {
#pragma nounroll
    for (int phase = 0; phase < 2; ++phase) {
      int stop = rstop;
      if (phase == 0) {
        stop = rstart + ((fast_lo - rstart + B - 1) / B) * B;  // first loop position at or after fast_lo
        if (stop > rstop) stop = rstop;
      }
#pragma nounroll
      for (; base < stop; base += B) body(Checked<true>{}, base);
      if (phase == 0) {
#pragma nounroll
        for (; base < fast_hi; base += B) {
          body(Checked<false>{}, base);
          okbits = ~0u;  // all of these rows were inside the grid
        }
      }
    }
  } else {
#pragma nounroll
    for (; base < rstop; base += B) body(Checked<true>{}, base);
  }
}


#ifndef MGCMT_FUSED_MIN_ROWS
#define MGCMT_FUSED_MIN_ROWS 4
#endif
constexpr long kFusedMinRows = MGCMT_FUSED_MIN_ROWS;  // shortest chunk: a wave's march has a fixed cost per row step

template <class OP, int KIND, int NSWEEP, int FLAGS>
void launch_one(hipStream_t s, FusedArgs a, int k) {
  using Shape = FusedShape<OP, KIND, NSWEEP, FLAGS>;
  const long strips = (a.nc + Shape::wout - 1) / Shape::wout;
  const long groups = (strips + kWavesPerBlock - 1) / kWavesPerBlock;
  const long groups8 = (groups + 7) / 8 * 8;
  a.n_col_groups = (int)groups;
  // Rows per chunk.  Every wave does the same amount of work, so the best launch is ONE round: as many
  // chunks as fit on the chip at this kernel's occupancy (a second, partly filled round costs a whole
  // wave lifetime at low bandwidth), each as long as possible (long marches amortise the pipeline fill
  // and the overlap rows).  The occupancy query can be one block per CU optimistic (MI355X_MICROARCH.md,
  // "Residency"), hence the 10 % margin.
  static int resident_blocks = 0;  // per instantiation: workgroups the chip holds at once
  if (resident_blocks == 0) {
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_fused<OP, KIND, NSWEEP, FLAGS>, 64 * kWavesPerBlock, 0) != hipSuccess ||
        hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess || per_cu < 1)
      resident_blocks = 768;
    else
      resident_blocks = per_cu * prop.multiProcessorCount;
  }
  const long nrows1 = a.out_hi - a.out_lo, nrows2 = a.out_hi2 > a.out_lo2 ? a.out_hi2 - a.out_lo2 : 0;
  const long nrows = nrows1 + nrows2;  // rows this launch produces
  if (nrows1 <= 0) return;
  long rows = a.rows_override;
  if (rows <= 0) {
#ifndef MGCMT_FUSED_FILL
#define MGCMT_FUSED_FILL 0.9  // re-measured after the mapping change: 0.8 / 0.9 / 1.0 -> 3.00-3.02 / 2.92-2.93 / 2.94-2.97 ms per 16384^2 cycle
#endif
    long chunks = (long)(MGCMT_FUSED_FILL * resident_blocks) / (groups * k);
    if (chunks < 1) chunks = 1;
    rows = (nrows + chunks - 1) / chunks;
    // (second session: with the dependency-cone skip a Jacobi chunk of 2 rows beats 4 on small levels — 256^2 cycle 0.057 ->
    // 0.053 ms, 1024^2 0.107 -> 0.104 —, the colour smoothers' longer pipelines still want 4: 4096^2 red-black 0.403 vs 0.412)
    const long min_rows = (KIND == kJacobi && kFusedMinRows > 2) ? 2 : kFusedMinRows;
    if (rows < min_rows) rows = min_rows;
  }
  if (rows > nrows1) rows = nrows1;
  rows = (rows + 1) & ~1L;
  a.rows_per_chunk = (int)rows;
  a.n_row_chunks1 = (int)((nrows1 + rows - 1) / rows);
  a.n_row_chunks = a.n_row_chunks1 + (int)((nrows2 + rows - 1) / rows);
  // equal shares where whole column groups would leave the XCDs more than MGCMT_FUSED_XCD_IMBALANCE % apart — on levels up
  // to 8192 columns; the 16384^2 level keeps its column bands (measured 6 % faster there)
#ifndef MGCMT_FUSED_XCD_IMBALANCE
#define MGCMT_FUSED_XCD_IMBALANCE 4
#endif
#ifndef MGCMT_FUSED_XCD_MAXGROUPS
#define MGCMT_FUSED_XCD_MAXGROUPS 128
#endif
  a.xcd_balanced = (groups8 * 100 > groups * (100 + MGCMT_FUSED_XCD_IMBALANCE) && groups < MGCMT_FUSED_XCD_MAXGROUPS) ? 1 : 0;
  const unsigned blocks = (unsigned)(groups8 * a.n_row_chunks);
  hipLaunchKernelGGL((k_fused<OP, KIND, NSWEEP, FLAGS>), dim3(blocks, (unsigned)k), dim3(64 * kWavesPerBlock), 0, s, a);
}


// every variant of one (operator policy, smoother kind, sweeps per pass) combination
template <class OP, int KIND, int NSWEEP>
void launch_variant(hipStream_t s, const FusedArgs& a, int flags, int k) {
  constexpr bool kRecompute = KIND != kFourColour;  // four-colour sweeps are four stages each: no room to recompute
  switch (flags) {
    case 0: return launch_one<OP, KIND, NSWEEP, 0>(s, a, k);
    case kZeroIn: return launch_one<OP, KIND, NSWEEP, kZeroIn>(s, a, k);
    case kRestrict: return launch_one<OP, KIND, NSWEEP, kRestrict>(s, a, k);
    case kRestrict | kZeroIn: return launch_one<OP, KIND, NSWEEP, kRestrict | kZeroIn>(s, a, k);
    case kRestrict | kNoStore: return launch_one<OP, KIND, NSWEEP, kRestrict | kNoStore>(s, a, k);
    case kRestrict | kNoStore | kZeroIn: return launch_one<OP, KIND, NSWEEP, kRestrict | kNoStore | kZeroIn>(s, a, k);
    case kProlong: return launch_one<OP, KIND, NSWEEP, kProlong>(s, a, k);
    default: break;
  }
  if constexpr (kRecompute) {
    switch (flags) {
      case kProlong | (1 << kPreShift): return launch_one<OP, KIND, NSWEEP, kProlong | (1 << kPreShift)>(s, a, k);
      case kProlong | kZeroIn | (1 << kPreShift): return launch_one<OP, KIND, NSWEEP, kProlong | kZeroIn | (1 << kPreShift)>(s, a, k);
      default: break;
    }
    // two recomputed sweeps: as long as all stages fit the (wide) window overlap
    if constexpr (stages_of(KIND, 2) + stages_of(KIND, NSWEEP) + 1 <= 16) {
      switch (flags) {
        case kProlong | (2 << kPreShift): return launch_one<OP, KIND, NSWEEP, kProlong | (2 << kPreShift)>(s, a, k);
        case kProlong | kZeroIn | (2 << kPreShift): return launch_one<OP, KIND, NSWEEP, kProlong | kZeroIn | (2 << kPreShift)>(s, a, k);
        default: break;
      }
    }
  }
}

}  // namespace fused

// per-policy entry points (one translation unit each: kernels_fused_op5.hip, _op9c.hip, _op9.hip)
void launch_fused_op5(hipStream_t s, const fused::FusedArgs& a, int multicolour, int nsweep, int flags, int k);
void launch_fused_op9c(hipStream_t s, const fused::FusedArgs& a, int multicolour, int nsweep, int flags, int k);
void launch_fused_op9(hipStream_t s, const fused::FusedArgs& a, int multicolour, int nsweep, int flags, int k);    // two terms
void launch_fused_op9m3(hipStream_t s, const fused::FusedArgs& a, int multicolour, int nsweep, int flags, int k);  // three terms
void launch_fused_op5v(hipStream_t s, const fused::FusedArgs& a, int multicolour, int nsweep, int flags, int k);   // 5-point + product potential
void launch_fused_op9cv(hipStream_t s, const fused::FusedArgs& a, int multicolour, int nsweep, int flags, int k);  // constant 9-point + one variable term

}  // namespace mgcmt
