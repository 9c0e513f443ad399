// Dispatcher of the fused row-streaming passes (kernel template: fused_kernel.h; instantiations per operator
// policy: kernels_fused_op5.hip, kernels_fused_op9c.hip, kernels_fused_op9.hip).
#include "fused_kernel.h"

namespace mgcmt {

#ifndef MGCMT_FUSED_MIN_COLS
#define MGCMT_FUSED_MIN_COLS 16
#endif
constexpr long kFusedMinCols = MGCMT_FUSED_MIN_COLS;

// Levels the fused kernels cover: 2-D, constant 5-point operators, their Galerkin coarsenings, and general
// separable operators of two or three Kronecker terms (Laplacian plus potential).  Narrow levels run too (one partly filled wave per
// chunk): a fused pass there replaces four to nine tiny launches, which is what small levels cost.
bool fused_supported(const KGrid& g, const KOp& op) {
  if (!g.coarsen_rows) return fused1d_supported(g, op);
  return g.coarsen_rows && g.nr >= 4 && g.nc >= kFusedMinCols && (g.nc & 1) == 0 && (g.nr & 1) == 0 && (op.five_point || op.five_diag || op.nine_const || op.nine_var || op.nterms == 2 || op.nterms == 3);
}

// sweeps one pass can fuse.  A 9-point four-colour sweep is four stages: two of them plus the restriction read nine
// rows above the chunk — strips (sharded levels) therefore exchange ten halo rows on 9-point levels (exchanged_rows,
// plan.hip) and fuse two sweeps like the single plan (they fused one while a strip had eight halo rows).
int fused_max_sweeps(const KOp& op, int multicolour) {
  if (op.one_d) return fused1d_max_sweeps(multicolour);  // (no row pipeline in 1-D: the reference's default nu = 4 is one pass)
  return 2;
}

// sweeps of pre-smoothing an up-leg pass with `nsweep` post-smoothing sweeps can recompute in front of the
// correction (0: none): all stages plus the correction must fit the window overlap (16 columns at most) and
// the exchanged halo rows of a strip
int fused_max_recompute(const KOp& op, int multicolour, int nsweep) {
  if (op.one_d) return fused1d_max_recompute(multicolour, nsweep);
  if (!op.five_point && !op.five_diag && multicolour) return 0;  // four-colour sweeps: four stages each
  const int per_sweep = multicolour ? 2 : 1;
  int n = (8 - per_sweep * nsweep) / per_sweep;
  return n < 0 ? 0 : (n > 2 ? 2 : n);
}

// One fused pass: vin -> vout with `nsweep` sweeps.  mode & 3: 0 plain, 1 prolong+correct first (coarse = the
// correction), 2 residual+restriction last (coarse = the coarse right-hand side); mode & 4: vin is zero;
// mode & 8: vout is not written (mode 2 only); npre: pre-smoothing sweeps recomputed before the correction (mode 1).
void launch_fused(hipStream_t s, KGrid g, KOp op, KVec vin, KVec f, KVec vout, KVec coarse, long coarse_nc, const double* shifts,
                  double omega, int multicolour, int nsweep, int mode, int npre, long row_lo, long row_hi, long last_row, int k,
                  long rows_override, long out_lo, long out_hi, long out_lo2, long out_hi2) {
  if (op.one_d) return launch_fused1d(s, g, op, vin, f, vout, coarse, coarse_nc, shifts, omega, multicolour, nsweep, mode, npre, k);
  fused::FusedArgs a{};
  a.rows_override = (int)rows_override;
  a.out_lo = (int)out_lo;
  a.out_hi = (int)(out_hi < 0 ? g.nr : out_hi);
  a.out_lo2 = (int)out_lo2;
  a.out_hi2 = (int)out_hi2;
  a.vin = vin.p;
  a.f = f.p;
  a.vout = vout.p;
  a.ec = (mode & 3) == 1 ? coarse.p : nullptr;
  a.rc = (mode & 3) == 2 ? coarse.p : nullptr;
  a.nr = g.nr;
  a.nc = g.nc;
  a.row_lo = row_lo;
  a.row_hi = row_hi;
  a.vstride = vin.stride;
  a.cstride = coarse.stride;
  a.cnc = coarse_nc;
  a.c0 = op.c0;
  a.cn = op.cn;
  a.cw = op.cw;
  for (int m = 0; m < kMaxTerms; ++m) {
    a.X[m] = op.X[m];
    a.Y[m] = op.Y[m];
  }
  a.ldx = op.ldx;
  a.ldy = op.ldy;
  a.shifts = shifts;
  a.omega = omega;
  const int flags = (mode & 15) | (npre << fused::kPreShift);
  if (op.five_point) {
    launch_fused_op5(s, a, multicolour, nsweep, flags, k);
  } else if (op.five_diag) {
    for (int m = 0; m < op.ndiag; ++m) {
      a.X[m] = op.dX[m];
      a.Y[m] = op.dY[m];
    }
    launch_fused_op5v(s, a, multicolour, nsweep, flags, k);
  } else if (op.nine_const) {
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) a.c9[i][j] = op.c9[i][j];
      a.c9row[i] = op.c9row[i];
      a.c9col[i] = op.c9col[i];
    }
    a.c9corner = op.c9corner;
    a.last_row = last_row;
    launch_fused_op9c(s, a, multicolour, nsweep, flags, k);
  } else if (op.nine_var) {
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) a.c9[i][j] = op.c9[i][j];
      a.c9row[i] = op.c9row[i];
      a.c9col[i] = op.c9col[i];
    }
    a.c9corner = op.c9corner;
    a.last_row = last_row;
    a.X[0] = op.vX;
    a.Y[0] = op.vY;
    launch_fused_op9cv(s, a, multicolour, nsweep, flags, k);
  } else if (op.nterms == 3) {
    launch_fused_op9m3(s, a, multicolour, nsweep, flags, k);
  } else {
    launch_fused_op9(s, a, multicolour, nsweep, flags, k);
  }
}

}  // namespace mgcmt
