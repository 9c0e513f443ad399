// Fused passes of 1-D levels (the reference's own problems are 1-D: 1DPotMatrixVcycle.py:68-75, RQMin.py, every
// UnitTests KAT): one launch = [recomputed pre-smoothing] -> [V += P e] -> nu sweeps -> [coarse F = R (F - (A - mu I) V')],
// i.e. MGCMTSolver.py:323-326 or :313-315 in ONE pass over the level — 24 B per point (+ 4 B for the coarse array)
// whatever nu is, where the one-launch-per-operation kernels move 24 B per sweep plus 2 x 18 B for the transfers and
// pay a launch each.
//
// Mapping.  A 1-D level has no second direction to march along, so nothing is pipelined: a wave loads a WINDOW of 128
// consecutive points (2 per lane: 16-byte, fully coalesced accesses), runs every stage of the pass on it in registers —
// lateral neighbours from the adjacent lanes by wave-wide DPP shifts, no LDS, no barriers — and stores the middle.
// Values at the window's edges go stale one point per Jacobi sweep (two per red-black sweep, one more for the
// interpolated correction, two for the restriction), so consecutive windows overlap by `halo` points per side, 8, 16,
// 24 or 32 — whole 128-byte lines, so that every wave stores whole lines — and recompute them (the overlap is read from
// L1 / L2: neighbouring windows belong to neighbouring waves of one workgroup).
//
// Operators of a 1-D level (MGCMTStencilMaker.py:17-21 and its Galerkin coarsenings, MGCMTSolver.py:318): a constant
// tridiagonal whose last diagonal entry may differ (the one-sided P / R of the reference leave exactly that) — four
// scalars — or a general tridiagonal (a potential on the diagonal, 1DPotMatrixVcycle.py:16) read from the level's
// combined factor array, 48 B per point more.  Colours of the multicolour smoother in 1-D: odd points, then even ones.
#include <cstdlib>

#include "fused_kernel.h"

namespace mgcmt {

namespace fused1d {

using fused::lane_fetch_addr;
#if MGCMT_FUSED_DPP && defined(__HIP_DEVICE_COMPILE__)
using fused::lane_shift;
#endif

struct Args {
  const double* vin;
  const double* f;
  double* vout;
  const double* ec;  // coarse correction (prolong)
  double* rc;        // coarse right-hand side (restrict)
  long n, cn;        // points of the level / of the coarse level
  long vstride, cstride;
  double cl, c0, cu, clast;  // constant operator: lower, diagonal, upper, last diagonal entry
  const double* tri;         // general operator: [lower | diag | upper], n numbers each
  const double* shifts;
  double omega;
  int nsweep, npre;
  int prolong, restrict_, zero_in, store;
  int halo;                 // points of overlap per side: 8, 16, 24 or 32
  long nwindows, wave_stride;  // windows of the level; windows between two trips of a wave
};

// neighbour-lane reads: the 2-D kernels' forms (DPP shifts on the GPU; a lane without a neighbour lies in the overlap)
#define MGCMT_1D_FROM_LEFT(v) MGCMT_FETCH_LEFT(lane_up, v)    // lane - 1's value
#define MGCMT_1D_FROM_RIGHT(v) MGCMT_FETCH_RIGHT(lane_dn, v)  // lane + 1's value

template <int KIND, bool VAR>
__global__ void __launch_bounds__(256) k_fused1d(Args a) {
  const int lane = threadIdx.x & 63;
  const int lane_up = (lane > 0 ? lane - 1 : 0) << 2, lane_dn = (lane < 63 ? lane + 1 : 63) << 2;  // (the ds_bpermute form's addresses)
  (void)lane_up;
  (void)lane_dn;
  const long wave0 = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int q = blockIdx.y;
  const double* __restrict__ vin = a.vin + q * a.vstride;
  const double* __restrict__ fin = a.f + q * a.vstride;
  double* __restrict__ vout = a.vout + q * a.vstride;
  const double* __restrict__ ec = a.prolong ? a.ec + q * a.cstride : nullptr;
  double* __restrict__ rc = a.restrict_ ? a.rc + q * a.cstride : nullptr;
  const double mu = a.shifts[q];
  const long n = a.n;
  const int wout = 128 - 2 * a.halo;
  const bool lane_out = 2 * lane >= a.halo && 2 * lane < a.halo + wout;
  for (long w = wave0; w < a.nwindows; w += a.wave_stride) {
    const long ja = w * wout - a.halo + 2 * lane;  // this lane's points: ja (even), ja + 1
    const bool in = ja >= 0 && ja < n;             // (n is even: both or neither)
    const long jl = ja < 0 ? 0 : (ja > n - 2 ? n - 2 : ja);  // clamped: loads are unconditional, values masked
    const double m = in ? 1.0 : 0.0;
    const double omega = in ? a.omega : 0.0;       // points outside the grid are Dirichlet ghosts: zero, never updated
    double va = 0.0, vb = 0.0;
    if (!a.zero_in) {
      const double2 t = fused::load2(vin + jl);
      va = t.x * m;
      vb = t.y * m;
    }
    const double2 ft = fused::load2(fin + jl);
    const double fa = ft.x, fb = ft.y;
    // operator at the two points: lower, diagonal of (A - mu I) and its reciprocal, upper
    double la, da, ua, lb, db, ub;
    if (VAR) {
      const double2 tl = fused::load2(a.tri + jl), td = fused::load2(a.tri + n + jl), tu = fused::load2(a.tri + 2 * n + jl);
      la = tl.x; lb = tl.y;
      da = td.x - mu; db = td.y - mu;
      ua = tu.x; ub = tu.y;
    } else {
      la = lb = a.cl;
      ua = ub = a.cu;
      da = a.c0 - mu;
      db = (ja + 1 == n - 1 ? a.clast : a.c0) - mu;
    }
    const double ia = 1.0 / da, ib = 1.0 / db;
    double e = 0.0;
    if (a.prolong) {
      const long J = jl >> 1;
      e = ec[J < a.cn ? J : a.cn - 1] * m;
    }
    auto sweep = [&]() __attribute__((always_inline)) {
      if (KIND == 0) {  // weighted Jacobi: both points from the old values
        const double left = MGCMT_1D_FROM_LEFT(vb), right = MGCMT_1D_FROM_RIGHT(va);
        const double na = fma(omega, (fa - fma(da, va, fma(la, left, ua * vb))) * ia, va);
        const double nb = fma(omega, (fb - fma(db, vb, fma(lb, va, ub * right))) * ib, vb);
        va = na;
        vb = nb;
      } else {  // odd points first (from the even ones), then the even points from the new odd ones
        const double right = MGCMT_1D_FROM_RIGHT(va);
        vb = fma(omega, (fb - fma(db, vb, fma(lb, va, ub * right))) * ib, vb);
        const double left = MGCMT_1D_FROM_LEFT(vb);
        va = fma(omega, (fa - fma(da, va, fma(la, left, ua * vb))) * ia, va);
      }
    };
    for (int s = 0; s < a.npre; ++s) sweep();
    if (a.prolong) {
      // V += P e: odd fine point 2J+1 takes e[J], even point 2J the mean of e[J-1] and e[J] (e[-1] = 0)
      const double eprev = MGCMT_1D_FROM_LEFT(e);
      va = fma(m, 0.5 * (eprev + e), va);
      vb = fma(m, e, vb);
    }
    for (int s = 0; s < a.nsweep; ++s) sweep();
    if (a.store && lane_out && in) fused::store2_stream(vout + ja, va, vb);
    if (a.restrict_) {
      const double left = MGCMT_1D_FROM_LEFT(vb), right = MGCMT_1D_FROM_RIGHT(va);
      const double ra = m * (fa - fma(da, va, fma(la, left, ua * vb)));
      const double rb = m * (fb - fma(db, vb, fma(lb, va, ub * right)));
      const double rnext = MGCMT_1D_FROM_RIGHT(ra);  // residual at point ja + 2 (zero beyond the grid)
      if (lane_out && in) rc[ja >> 1] = 0.25 * ra + 0.5 * rb + 0.25 * rnext;
    }
  }
}

}  // namespace fused1d

// 1-D levels the fused passes cover: every even length from 16 points on (a short level is one partly filled window — one
// launch where the simple kernels take nu + 3; MGCMT_FUSED1D_MIN: A/B measurements)
#ifndef MGCMT_FUSED1D_MIN
#define MGCMT_FUSED1D_MIN 16
#endif
bool fused1d_supported(const KGrid& g, const KOp& op) {
  return !g.coarsen_rows && g.nr == 1 && g.nc >= MGCMT_FUSED1D_MIN && (g.nc & 1) == 0 && (op.tri_const || op.tri != nullptr);
}

// stages' reach per side: one point per Jacobi sweep, two per red-black sweep, one for the correction, two for the restriction
static int reach(int multicolour, int sweeps, int prolong, int restrict_) { return (multicolour ? 2 : 1) * sweeps + (prolong ? 1 : 0) + (restrict_ ? 2 : 0); }

int fused1d_max_sweeps(int multicolour) { return multicolour ? 6 : 12; }

int fused1d_max_recompute(int multicolour, int nsweep) {
  // everything must fit 32 points of overlap; beyond 16 the redundant half of every window costs more than the store saves
  int n = 0;
  while (n < 4 && reach(multicolour, n + 1 + nsweep, 1, 0) <= 16) ++n;
  return n;
}

void launch_fused1d(hipStream_t s, KGrid g, KOp op, KVec vin, KVec f, KVec vout, KVec coarse, long coarse_nc, const double* shifts,
                    double omega, int multicolour, int nsweep, int mode, int npre, int k) {
  fused1d::Args a{};
  a.vin = vin.p;
  a.f = f.p;
  a.vout = vout.p;
  a.prolong = (mode & 3) == 1;
  a.restrict_ = (mode & 3) == 2;
  a.ec = a.prolong ? coarse.p : nullptr;
  a.rc = a.restrict_ ? coarse.p : nullptr;
  a.n = g.nc;
  a.cn = coarse_nc;
  a.vstride = vin.stride;
  a.cstride = coarse.stride;
  a.cl = op.t_lo;
  a.c0 = op.t_di;
  a.cu = op.t_up;
  a.clast = op.t_last;
  a.tri = op.tri;
  a.shifts = shifts;
  a.omega = omega;
  a.nsweep = nsweep;
  a.npre = npre;
  a.zero_in = (mode & 4) ? 1 : 0;
  a.store = (mode & 8) ? 0 : 1;
  const int need = reach(multicolour, nsweep + npre, a.prolong, a.restrict_);
  a.halo = need <= 8 ? 8 : (need <= 16 ? 16 : (need <= 24 ? 24 : 32));
  const long wout = 128 - 2 * a.halo;
  a.nwindows = (g.nc + wout - 1) / wout;
  // one wave per window up to a few rounds of the chip, then grid-stride trips (MGCMT_FUSED1D_MAX_WAVES: A/B measurements)
  static const long max_waves = [] {
    const char* e = getenv("MGCMT_FUSED1D_MAX_WAVES");
    const long v = e ? atol(e) : 0;
    return v > 0 ? v : 256L * 32 * 4;
  }();
  long waves = a.nwindows < max_waves ? a.nwindows : max_waves;
  const unsigned blocks = (unsigned)((waves + 3) / 4);
  a.wave_stride = (long)blocks * 4;
  const dim3 grid(blocks, (unsigned)k), block(256);
  const bool var = !op.tri_const;
  if (multicolour) {
    if (var) hipLaunchKernelGGL((fused1d::k_fused1d<1, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((fused1d::k_fused1d<1, false>), grid, block, 0, s, a);
  } else {
    if (var) hipLaunchKernelGGL((fused1d::k_fused1d<0, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((fused1d::k_fused1d<0, false>), grid, block, 0, s, a);
  }
}

}  // namespace mgcmt
