// Rayleigh-quotient minimisation (MGCMTSolver.py:17-57) as TWO passes over the data per step and no host round trip.
//
// The reference's step:  p <- -g + beta p;  the 2 x 2 pencil on span{x, p} from eight inner products with A x, A p, M x,
// M p (:33-46);  x <- x + delta p (:50-52);  rho = <x,Ax>/<x,Mx> (:53);  g <- 2 (A x - rho M x) (:55).  Taken literally
// that is six operator applications, three Gram passes and four vector updates per step with three host round trips
// (the round-2 form, ~300 B per point and step).  Here:
//   pass 1  reads x, g, p_old: forms p = -g + beta p_old on the fly (beta is a device scalar), applies A and M to x and p
//           in registers, accumulates the eight inner products, writes p                                  (32 B per point)
//   scalars one workgroup: sums the per-block partial sums in a fixed order, solves the 2 x 2 generalised eigenproblem in
//           closed form, delta = y1/y0 of the smaller eigenvalue's vector; rho of x + delta p from the same eight numbers
//   pass 2  reads x, p: forms x' = x + delta p on the fly, applies A and M to it, g' = 2 (A x' - rho M x'); writes x', g';
//           accumulates <x',Ax'>, <x',Mx'> (the rho that is reported, :53) and <g',g'>                     (32 B per point)
//   scalars rho, beta = <g',Mg'> / <g,Mg> for the next step (:31).  With M != I, <g',Mg'> takes one more application.
// x and p are ping-ponged between two vectors each: a pass recomputes the combination at the neighbours' points, so it
// must not overwrite what they still read.
//
// 2-D levels with even sizes take the row march of kernels_stencil.hip (a thread owns two adjacent columns and walks down
// a chunk of rows with the 3 x 4 neighbourhoods of both vectors in registers); 1-D levels and odd shapes a
// one-thread-per-point form with the same arithmetic.
#include <cstdint>
#include <cstdlib>

#include "mgcmt_internal.h"

namespace mgcmt {

namespace {

// state words (doubles in device memory, one block per plan)
enum {
  kS_xAx = 0, kS_xAp, kS_pAx, kS_pAp, kS_xMx, kS_xMp, kS_pMx, kS_pMp,  // pass 1
  kDelta = 8, kRho, kBeta, kGMGprev, kGMG, kStop, kXAXn, kXMXn, kGG, kRhoLin,
  kRqStateWords = 32
};

struct Row4 {
  double w, a, b, e;  // columns j-1, j, j+1, j+2 (zero outside the grid)
};

// operator forms: M = 0 the identity; 1 .. 4 that many Kronecker terms with general tridiagonal factors; kFive + ND a
// constant 5-point operator plus ND (0 .. 2) product potentials p_m(i) q_m(j) on the diagonal (KOp.five_point /
// five_diag: the finest level of a scaled Laplacian or of a square-well Hamiltonian — six multiply-adds per point instead
// of the 36 + of three general terms, and two column values in registers instead of 18)
constexpr int kFive = 5;
template <int M>
struct Fac {
  static constexpr int T = (M >= 1 && M <= 4) ? M : 1;
  double yl[T][2], yd[T][2], yu[T][2];
  double qa[2], qb[2];  // kFive + ND: q_m at the two columns
};

template <int M>
__device__ __forceinline__ void load_fac(const KOp& op, long j, Fac<M>& f) {
  if constexpr (M >= kFive) {
#pragma unroll
    for (int m = 0; m < M - kFive; ++m) {
      f.qa[m] = op.dY[m][j];
      f.qb[m] = op.dY[m][j + 1];
    }
  } else {
#pragma unroll
    for (int m = 0; m < M; ++m)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const double* Y = op.Y[m] + j + c;
        f.yl[m][c] = Y[0];
        f.yd[m][c] = Y[op.ldy];
        f.yu[m][c] = Y[2 * op.ldy];
      }
  }
}

// A Kronecker term applied along a march: (X (x) Y) v at row i is xl (Y v_{i-1}) + xd (Y v_i) + xu (Y v_{i+1}) — the row
// transforms Y v_r (three multiply-adds per point and term) are computed ONCE, when row r is loaded, and carried down
// the march with the rows themselves; combining three of them takes three more.  Six multiply-adds per point and term
// where evaluating the 9-point stencil of every term at every point takes twelve: on the Galerkin levels — every term
// a full 9-point stencil, M too — these passes are bound by the vector ALUs, not by memory (pass 1 at 4096^2 with three
// terms in A and one in M: 140 us for 0.54 GB).
template <int M>
struct RowT {
  static constexpr int T = (M >= 1 && M <= 4) ? M : 1;
  double a[T], b[T];  // per term: (Y v)(r, j), (Y v)(r, j + 1)
};

template <int M>
__device__ __forceinline__ RowT<M> transform(const Fac<M>& f, const Row4& r) {
  RowT<M> t;
  if constexpr (M >= 1 && M <= 4) {
    // (explicit multiply-adds: the library is built with -ffp-contract=off)
#pragma unroll
    for (int m = 0; m < M; ++m) {
      t.a[m] = fma(f.yu[m][0], r.b, fma(f.yd[m][0], r.a, f.yl[m][0] * r.w));
      t.b[m] = fma(f.yu[m][1], r.e, fma(f.yd[m][1], r.b, f.yl[m][1] * r.a));
    }
  } else {
    t.a[0] = t.b[0] = 0.0;  // (the identity and the 5-point forms work on the rows themselves)
  }
  return t;
}

// (Op v) at (i, j) and (i, j+1) from the rows above / at / below (n, c, s) and their transforms; M == 0: the identity
template <int M>
__device__ __forceinline__ void apply2(const KOp& op, const Fac<M>& f, long i, const Row4& n, const Row4& c, const Row4& s, const RowT<M>& tn,
                                       const RowT<M>& tc, const RowT<M>& ts, double& ra, double& rb) {
  if (M == 0) {
    ra = c.a;
    rb = c.b;
    return;
  }
  if constexpr (M >= kFive) {
    double da = op.c0, db = op.c0;
#pragma unroll
    for (int m = 0; m < M - kFive; ++m) {
      const double pm = op.dX[m][i];
      da = fma(pm, f.qa[m], da);
      db = fma(pm, f.qb[m], db);
    }
    ra = fma(da, c.a, fma(op.cn, n.a + s.a, op.cw * (c.w + c.b)));
    rb = fma(db, c.b, fma(op.cn, n.b + s.b, op.cw * (c.a + c.e)));
    return;
  }
  double va = 0.0, vb = 0.0;
  constexpr int TERMS = M < kFive ? M : 0;
#pragma unroll
  for (int m = 0; m < TERMS; ++m) {
    const double* X = op.X[m] + i;
    const double xl = X[0], xd = X[op.ldx], xu = X[2 * op.ldx];
    va = fma(xu, ts.a[m], fma(xd, tc.a[m], fma(xl, tn.a[m], va)));
    vb = fma(xu, ts.b[m], fma(xd, tc.b[m], fma(xl, tn.b[m], vb)));
  }
  ra = va;
  rb = vb;
}

constexpr int kRqThreads = 256;
constexpr int kRqSums = 8;

// per block: nq partial sums into partials[q * nblocks + block] (fixed order: deterministic)
template <int NQ>
__device__ __forceinline__ void block_partials(const double* acc, double (*s_part)[kRqThreads / 64], double* __restrict__ partials, int nblocks, int block) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    double t = acc[q];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) t += __shfl_down(t, d);
    if (lane == 0) s_part[q][wave] = t;
  }
  __syncthreads();
  if ((int)threadIdx.x < NQ) {
    double t = 0.0;
    const int nw = blockDim.x >> 6;
    for (int k = 0; k < nw; ++k) t += s_part[threadIdx.x][k];
    partials[(long)threadIdx.x * nblocks + block] = t;
  }
}

// ---- row march (2-D, even sizes) ------------------------------------------------------------------------------

// v = cu * u + cw * w at columns j-1 .. j+2 of row i (u alone when w == nullptr).  (The neighbour columns by wave-wide DPP
// shifts instead of two more loads per row and vector — only the wave's first and last lane loading theirs — was measured
// in round 3: 8.82 against 8.27 ms per 8192^2 vcycle_rqmg cycle; the loads from L1 are cheaper than the divergent edge
// lanes and the loop the compiler then no longer unrolls.)
__device__ __forceinline__ Row4 load4(const double* __restrict__ u, const double* __restrict__ w, double cu, double cw, long i, long nc, long j, long jw,
                                      long je, bool hw, bool he) {
  const double2 c = *reinterpret_cast<const double2*>(u + i * nc + j);
  Row4 r{hw ? u[i * nc + jw] : 0.0, c.x, c.y, he ? u[i * nc + je] : 0.0};
  if (!w && cu != 1.0) {
    r.w *= cu;
    r.a *= cu;
    r.b *= cu;
    r.e *= cu;
  }
  if (w) {
    const double2 d = *reinterpret_cast<const double2*>(w + i * nc + j);
    const double dw = hw ? w[i * nc + jw] : 0.0, de = he ? w[i * nc + je] : 0.0;
    r.w = fma(cw, dw, cu * r.w);
    r.a = fma(cw, d.x, cu * r.a);
    r.b = fma(cw, d.y, cu * r.b);
    r.e = fma(cw, de, cu * r.e);
  }
  return r;
}

template <int MA, int MM>
__global__ void __launch_bounds__(kRqThreads) k_rq_pass1(KGrid g, KOp A, KOp Mo, const double* __restrict__ x, const double* __restrict__ gv,
                                                        const double* __restrict__ pold, double* __restrict__ pnew, const double* __restrict__ state,
                                                        int init, int rows, double* __restrict__ partials, int nblocks) {
  __shared__ double s_part[kRqSums][kRqThreads / 64];
  const long j = 2 * ((long)blockIdx.x * blockDim.x + threadIdx.x);
  const long nc = g.nc;
  const long i0 = (long)blockIdx.y * rows;
  const long i1 = i0 + rows < g.nr ? i0 + rows : g.nr;
  double acc[kRqSums];
#pragma unroll
  for (int q = 0; q < kRqSums; ++q) acc[q] = 0.0;
  if (j < nc) {
    const bool hw = j > 0, he = j + 2 < nc;
    const long jw = hw ? j - 1 : j, je = he ? j + 2 : j + 1;
    const double beta = init ? 0.0 : state[kBeta];
    const double* po = init >= 2 ? nullptr : pold;  // the first step takes p = -g (MGCMTSolver.py:29-30): p_old is not read
    Fac<MA> fa;
    Fac<MM> fm;
    load_fac<MA>(A, j, fa);
    load_fac<MM>(Mo, j, fm);
    const Row4 zero{0.0, 0.0, 0.0, 0.0};
    const double cg = init == 3 ? 1.0 : -1.0;  // (init 3: the direction is gv itself, mgcmt_rq_line_step)
    auto loadp = [&](long i) { return init == 1 ? zero : load4(gv, po, cg, beta, i, nc, j, jw, je, hw, he); };
    Row4 xn = load4(x, nullptr, 1.0, 0.0, i0 - 1, nc, j, jw, je, hw, he), xc = load4(x, nullptr, 1.0, 0.0, i0, nc, j, jw, je, hw, he);
    Row4 pn = loadp(i0 - 1), pc = loadp(i0);
    RowT<MA> axn = transform<MA>(fa, xn), axc = transform<MA>(fa, xc), apn = transform<MA>(fa, pn), apc = transform<MA>(fa, pc);
    RowT<MM> mxn = transform<MM>(fm, xn), mxc = transform<MM>(fm, xc), mpn = transform<MM>(fm, pn), mpc = transform<MM>(fm, pc);
#pragma unroll 2
    for (long i = i0; i < i1; ++i) {
      const Row4 xs = load4(x, nullptr, 1.0, 0.0, i + 1, nc, j, jw, je, hw, he);
      const Row4 ps = loadp(i + 1);
      const RowT<MA> axs = transform<MA>(fa, xs), aps = transform<MA>(fa, ps);
      const RowT<MM> mxs = transform<MM>(fm, xs), mps = transform<MM>(fm, ps);
      double axa, axb, apa, apb, mxa, mxb, mpa, mpb;
      apply2<MA>(A, fa, i, xn, xc, xs, axn, axc, axs, axa, axb);
      apply2<MA>(A, fa, i, pn, pc, ps, apn, apc, aps, apa, apb);
      apply2<MM>(Mo, fm, i, xn, xc, xs, mxn, mxc, mxs, mxa, mxb);
      apply2<MM>(Mo, fm, i, pn, pc, ps, mpn, mpc, mps, mpa, mpb);
      acc[kS_xAx] = fma(xc.b, axb, fma(xc.a, axa, acc[kS_xAx]));
      acc[kS_xAp] = fma(xc.b, apb, fma(xc.a, apa, acc[kS_xAp]));
      acc[kS_pAx] = fma(pc.b, axb, fma(pc.a, axa, acc[kS_pAx]));
      acc[kS_pAp] = fma(pc.b, apb, fma(pc.a, apa, acc[kS_pAp]));
      acc[kS_xMx] = fma(xc.b, mxb, fma(xc.a, mxa, acc[kS_xMx]));
      acc[kS_xMp] = fma(xc.b, mpb, fma(xc.a, mpa, acc[kS_xMp]));
      acc[kS_pMx] = fma(pc.b, mxb, fma(pc.a, mxa, acc[kS_pMx]));
      acc[kS_pMp] = fma(pc.b, mpb, fma(pc.a, mpa, acc[kS_pMp]));
      if (init == 0 || init == 2) *reinterpret_cast<double2*>(pnew + i * nc + j) = make_double2(pc.a, pc.b);
      xn = xc;
      xc = xs;
      pn = pc;
      pc = ps;
      axn = axc;
      axc = axs;
      apn = apc;
      apc = aps;
      mxn = mxc;
      mxc = mxs;
      mpn = mpc;
      mpc = mps;
    }
  }
  block_partials<kRqSums>(acc, s_part, partials, nblocks, (int)(blockIdx.y * gridDim.x + blockIdx.x));
}

template <int MA, int MM>
__global__ void __launch_bounds__(kRqThreads) k_rq_pass2(KGrid g, KOp A, KOp Mo, const double* __restrict__ x, const double* __restrict__ p,
                                                        double* __restrict__ xnew, double* __restrict__ gout, const double* __restrict__ state, int init,
                                                        int rows, double* __restrict__ partials, int nblocks) {
  __shared__ double s_part[3][kRqThreads / 64];
  const long j = 2 * ((long)blockIdx.x * blockDim.x + threadIdx.x);
  const long nc = g.nc;
  const long i0 = (long)blockIdx.y * rows;
  const long i1 = i0 + rows < g.nr ? i0 + rows : g.nr;
  double acc[3] = {0.0, 0.0, 0.0};
  if (j < nc) {
    const bool hw = j > 0, he = j + 2 < nc;
    const long jw = hw ? j - 1 : j, je = he ? j + 2 : j + 1;
    const double delta = state[kDelta], rho = state[kRhoLin];
    Fac<MA> fa;
    Fac<MM> fm;
    load_fac<MA>(A, j, fa);
    load_fac<MM>(Mo, j, fm);
    const double* pp = init == 1 ? nullptr : p;
    Row4 xn = load4(x, pp, 1.0, delta, i0 - 1, nc, j, jw, je, hw, he), xc = load4(x, pp, 1.0, delta, i0, nc, j, jw, je, hw, he);
    RowT<MA> axn = transform<MA>(fa, xn), axc = transform<MA>(fa, xc);
    RowT<MM> mxn = transform<MM>(fm, xn), mxc = transform<MM>(fm, xc);
#pragma unroll 2
    for (long i = i0; i < i1; ++i) {
      const Row4 xs = load4(x, pp, 1.0, delta, i + 1, nc, j, jw, je, hw, he);
      const RowT<MA> axs = transform<MA>(fa, xs);
      const RowT<MM> mxs = transform<MM>(fm, xs);
      double axa, axb, mxa, mxb;
      apply2<MA>(A, fa, i, xn, xc, xs, axn, axc, axs, axa, axb);
      apply2<MM>(Mo, fm, i, xn, xc, xs, mxn, mxc, mxs, mxa, mxb);
      const double ga = 2.0 * (axa - rho * mxa), gb = 2.0 * (axb - rho * mxb);
      if (init != 1) *reinterpret_cast<double2*>(xnew + i * nc + j) = make_double2(xc.a, xc.b);  // (the initial pair: x' = x stays where it is)
      *reinterpret_cast<double2*>(gout + i * nc + j) = make_double2(ga, gb);
      acc[0] = fma(xc.b, axb, fma(xc.a, axa, acc[0]));
      acc[1] = fma(xc.b, mxb, fma(xc.a, mxa, acc[1]));
      acc[2] = fma(gb, gb, fma(ga, ga, acc[2]));
      xn = xc;
      xc = xs;
      axn = axc;
      axc = axs;
      mxn = mxc;
      mxc = mxs;
    }
  }
  block_partials<3>(acc, s_part, partials, nblocks, (int)(blockIdx.y * gridDim.x + blockIdx.x));
}

// <g, M g> for M != I without storing M g: the march over g alone, result 3 of the pass-2 partial sums (same grid)
template <int MM>
__global__ void __launch_bounds__(kRqThreads) k_rq_gmg(KGrid g, KOp Mo, const double* __restrict__ gv, int rows, double* __restrict__ partials, int nblocks) {
  __shared__ double s_part[1][kRqThreads / 64];
  const long j = 2 * ((long)blockIdx.x * blockDim.x + threadIdx.x);
  const long nc = g.nc;
  const long i0 = (long)blockIdx.y * rows;
  const long i1 = i0 + rows < g.nr ? i0 + rows : g.nr;
  double acc[1] = {0.0};
  if (j < nc) {
    const bool hw = j > 0, he = j + 2 < nc;
    const long jw = hw ? j - 1 : j, je = he ? j + 2 : j + 1;
    Fac<MM> fm;
    load_fac<MM>(Mo, j, fm);
    Row4 gn = load4(gv, nullptr, 1.0, 0.0, i0 - 1, nc, j, jw, je, hw, he), gc = load4(gv, nullptr, 1.0, 0.0, i0, nc, j, jw, je, hw, he);
    RowT<MM> tn = transform<MM>(fm, gn), tc = transform<MM>(fm, gc);
#pragma unroll 2
    for (long i = i0; i < i1; ++i) {
      const Row4 gs = load4(gv, nullptr, 1.0, 0.0, i + 1, nc, j, jw, je, hw, he);
      const RowT<MM> ts = transform<MM>(fm, gs);
      double ma, mb;
      apply2<MM>(Mo, fm, i, gn, gc, gs, tn, tc, ts, ma, mb);
      acc[0] = fma(gc.b, mb, fma(gc.a, ma, acc[0]));
      gn = gc;
      gc = gs;
      tn = tc;
      tc = ts;
    }
  }
  block_partials<1>(acc, s_part, partials + 3L * nblocks, nblocks, (int)(blockIdx.y * gridDim.x + blockIdx.x));
}

// ---- one thread per point (1-D levels, odd shapes): the same arithmetic through direct neighbour loads ---------------

// (Op v)(i, j) with v = cu u + cw w (u alone when w == nullptr); identity: v(i, j)
__device__ __forceinline__ double apply_point(const KOp& op, int identity, const double* __restrict__ u, const double* __restrict__ w, double cu, double cw,
                                              long nc, long i, long j) {
  auto val = [&](long off) { return w ? cu * u[off] + cw * w[off] : cu * u[off]; };
  const long c = i * nc + j;
  if (identity) return val(c);
  const bool hw = j > 0, he = j + 1 < nc;
  const double vc = val(c), n = val(c - nc), s = val(c + nc);
  const double wv = hw ? val(c - 1) : 0.0, e = he ? val(c + 1) : 0.0;
  const double nw = hw ? val(c - nc - 1) : 0.0, ne = he ? val(c - nc + 1) : 0.0;
  const double sw = hw ? val(c + nc - 1) : 0.0, se = he ? val(c + nc + 1) : 0.0;
  double off = 0.0, diag = 0.0;
  for (int m = 0; m < op.nterms; ++m) {
    const double* X = op.X[m] + i;
    const double* Y = op.Y[m] + j;
    const double xl = X[0], xd = X[op.ldx], xu = X[2 * op.ldx];
    const double yl = Y[0], yd = Y[op.ldy], yu = Y[2 * op.ldy];
    const double rn = yl * nw + yd * n + yu * ne;
    const double rc = yl * wv + yu * e;
    const double rs = yl * sw + yd * s + yu * se;
    off += xl * rn + xd * rc + xu * rs;
    diag += xd * yd;
  }
  return off + diag * vc;
}

__global__ void __launch_bounds__(kRqThreads) k_rq_pass1_point(KGrid g, KOp A, KOp Mo, int m_identity, const double* __restrict__ x,
                                                              const double* __restrict__ gv, const double* __restrict__ pold, double* __restrict__ pnew,
                                                              const double* __restrict__ state, int init, double* __restrict__ partials, int nblocks) {
  __shared__ double s_part[kRqSums][kRqThreads / 64];
  const long n = g.nr * g.nc;
  const double beta = init ? 0.0 : state[kBeta];
  double acc[kRqSums];
#pragma unroll
  for (int q = 0; q < kRqSums; ++q) acc[q] = 0.0;
  for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (long)gridDim.x * blockDim.x) {
    const long i = k / g.nc, j = k - i * g.nc;
    const double* po = init >= 2 ? nullptr : pold;
    const double cg = init == 3 ? 1.0 : -1.0;
    const double xc = x[k], pc = init == 1 ? 0.0 : (po ? -gv[k] + beta * po[k] : cg * gv[k]);
    const double ax = apply_point(A, 0, x, nullptr, 1.0, 0.0, g.nc, i, j), mx = apply_point(Mo, m_identity, x, nullptr, 1.0, 0.0, g.nc, i, j);
    const double ap = init == 1 ? 0.0 : apply_point(A, 0, gv, po, cg, beta, g.nc, i, j);
    const double mp = init == 1 ? 0.0 : apply_point(Mo, m_identity, gv, po, cg, beta, g.nc, i, j);
    acc[kS_xAx] += xc * ax;
    acc[kS_xAp] += xc * ap;
    acc[kS_pAx] += pc * ax;
    acc[kS_pAp] += pc * ap;
    acc[kS_xMx] += xc * mx;
    acc[kS_xMp] += xc * mp;
    acc[kS_pMx] += pc * mx;
    acc[kS_pMp] += pc * mp;
    if (init == 0 || init == 2) pnew[k] = pc;
  }
  block_partials<kRqSums>(acc, s_part, partials, nblocks, (int)blockIdx.x);
}

__global__ void __launch_bounds__(kRqThreads) k_rq_pass2_point(KGrid g, KOp A, KOp Mo, int m_identity, const double* __restrict__ x,
                                                              const double* __restrict__ p, double* __restrict__ xnew, double* __restrict__ gout,
                                                              const double* __restrict__ state, int init, double* __restrict__ partials, int nblocks) {
  __shared__ double s_part[3][kRqThreads / 64];
  const long n = g.nr * g.nc;
  const double delta = state[kDelta], rho = state[kRhoLin];
  const double* pp = init == 1 ? nullptr : p;
  double acc[3] = {0.0, 0.0, 0.0};
  for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (long)gridDim.x * blockDim.x) {
    const long i = k / g.nc, j = k - i * g.nc;
    const double xc = pp ? x[k] + delta * pp[k] : x[k];
    const double ax = apply_point(A, 0, x, pp, 1.0, delta, g.nc, i, j), mx = apply_point(Mo, m_identity, x, pp, 1.0, delta, g.nc, i, j);
    const double gg = 2.0 * (ax - rho * mx);
    if (init != 1) xnew[k] = xc;
    gout[k] = gg;
    acc[0] += xc * ax;
    acc[1] += xc * mx;
    acc[2] += gg * gg;
  }
  block_partials<3>(acc, s_part, partials, nblocks, (int)blockIdx.x);
}

// ---- scalars --------------------------------------------------------------------------------------------------

// results' per-block partial sums -> s_out[q] (q < nq <= 8), by a block of kScalarThreads.  The sums of a big level's
// pass are 4096 per result, fresh in HBM: a wave per result adding them up lane-strided is a chain of dependent
// ~2 us loads (measured 20 - 27 us per scalar kernel behind the 8192^2 and 4096^2 passes, 0.5 ms of an 8.2 ms cycle).
// Here the waves split each result between them and a lane issues ALL its loads before it adds the first: one memory
// latency.  Fixed order of additions for a given number of blocks (lane-strided, shuffle tree, pieces in order).
constexpr int kScalarThreads = 1024;
constexpr int kScalarDepth = 16;  // loads in flight per lane: 16 waves x 64 lanes x 16 = 16384 = 4 results of 4096 sums
__device__ __forceinline__ void reduce_results(const double* __restrict__ partials, int nblocks, int nq, double* s_out) {
  __shared__ double s_piece[kRqSums][kScalarThreads / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int nwaves = kScalarThreads / 64;
  const int slots = nq <= 4 ? 4 : kRqSums;  // results, padded to a divisor of the waves
  const int per = nwaves / slots;           // waves per result
  const int q = wave % slots, piece = wave / slots;
  const int span = ((nblocks + per - 1) / per + 63) & ~63;  // sums per wave: whole rounds of the lanes
  {
    const double* src = partials + (long)(q < nq ? q : 0) * nblocks;
    const int lo = piece * span, hi = q >= nq ? 0 : (lo + span < nblocks ? lo + span : nblocks);  // (a wave without a result: nothing)
    double acc = 0.0;
    for (int base = lo + lane; base < hi; base += 64 * kScalarDepth) {
      double v[kScalarDepth];
#pragma unroll
      for (int k = 0; k < kScalarDepth; ++k) {
        const int idx = base + 64 * k;
        v[k] = idx < hi ? src[idx] : 0.0;
      }
#pragma unroll
      for (int k = 0; k < kScalarDepth; ++k) acc += v[k];
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) acc += __shfl_down(acc, d);
    if (lane == 0 && q < nq) s_piece[q][piece] = acc;
  }
  __syncthreads();
  if ((int)threadIdx.x < nq) {
    double t = 0.0;
    for (int k = 0; k < per; ++k) t += s_piece[threadIdx.x][k];
    s_out[threadIdx.x] = t;
  }
  __syncthreads();
}

// after pass 1: the 2 x 2 pencil R y = lambda RM y (MGCMTSolver.py:33-49), delta = y1 / y0 of the smaller eigenvalue
// (:49-50), and rho of x + delta p by bilinearity (what pass 2 needs before it has formed x').  init: delta = 0.
// robust (the repaired variants, solver.py): a degenerate pencil ends the minimisation on this level (delta = 0 from
// here on) instead of producing infinities.
// (one thread) the step's scalars from the eight sums s
__device__ void rq_step_scalars(const double* s, double* __restrict__ state, int init, int robust) {
  for (int q = 0; q < kRqSums; ++q) state[q] = s[q];
  const double r00 = s[kS_xAx], r01 = s[kS_xAp], r10 = s[kS_pAx], r11 = s[kS_pAp];
  const double m00 = s[kS_xMx], m01 = s[kS_xMp], m10 = s[kS_pMx], m11 = s[kS_pMp];
  if (init == 1) {
    state[kDelta] = 0.0;
    state[kRhoLin] = r00 / m00;
    state[kStop] = 0.0;
    return;
  }
  double delta = 0.0;
  bool stop = state[kStop] != 0.0;
  if (!stop && robust) {
    const double ms01 = 0.5 * (m01 + m10);
    const double scale = fabs(m00) > 1e-300 ? fabs(m00) : 1e-300;
    auto fin = [](double v) { return fabs(v) <= 1.7e308; };  // (false for NaN and the infinities)
    const bool finite = fin(r00) && fin(r01) && fin(r10) && fin(r11) && fin(m00) && fin(ms01) && fin(m11);
    if (!finite || m11 <= 1e-28 * scale || (m00 * m11 - ms01 * ms01) <= 1e-14 * m00 * m11) stop = true;
  }
  if (!stop) {
    // det(R - l RM) = a l^2 + b l + c; the smaller root (a > 0 for a definite RM), stable form
    const double a = m00 * m11 - m01 * m10;
    const double b = -(r00 * m11 + m00 * r11) + (r01 * m10 + m01 * r10);
    const double c = r00 * r11 - r01 * r10;
    double disc = b * b - 4.0 * a * c;
    if (disc < 0.0) disc = 0.0;
    const double q = -0.5 * (b + (b >= 0.0 ? sqrt(disc) : -sqrt(disc)));
    const double l1 = q / a, l2 = q != 0.0 ? c / q : l1;
    const double lam = l1 < l2 ? l1 : l2;  // (np.argmin of the two eigenvalues, :49)
    // eigenvector from the row (p, .): delta = y1 / y0 = -(r10 - l m10) / (r11 - l m11).  That row, not (x, .): its
    // numerator is p . (A x - l M x), a product with the gradient, where the other row's r00 - l m00 cancels two numbers
    // of the size of <x, A x> down to rounding noise once x has converged (a 2-point level after one step) — and p may be
    // of ANY size, so the rows cannot be compared by magnitude.  The row (x, .) only when the pencil leaves no choice.
    const double n2 = r10 - lam * m10, d2 = r11 - lam * m11;
    delta = -n2 / d2;
    if (!(fabs(delta) <= 1.7e308)) delta = -(r00 - lam * m00) / (r01 - lam * m01);
    // p = 0 (a gradient that came out as exact zeros: x is an eigenvector to the last bit, as on a 2-point level after one
    // step): the minimiser over span{x} is x — where the reference's eig would return NaNs, nothing is updated
    if (!(fabs(delta) <= 1.7e308)) delta = 0.0;
    if (robust && !(fabs(delta) < 1e300)) {  // y0 = 0: the minimiser is p itself — the reference's ratio is infinite
      stop = true;
      delta = 0.0;
    }
  }
  state[kStop] = stop ? 1.0 : 0.0;
  state[kDelta] = delta;
  const double rl = (r00 + delta * (r01 + r10) + delta * delta * r11) / (m00 + delta * (m01 + m10) + delta * delta * m11);
  state[kRhoLin] = fabs(rl) <= 1.7e308 ? rl : r00 / m00;
}


__global__ void __launch_bounds__(kScalarThreads) k_rq_scalars1(const double* __restrict__ partials, int nblocks, double* __restrict__ state, int init,
                                                           int robust) {
  __shared__ double s[kRqSums];
  reduce_results(partials, nblocks, kRqSums, s);
  if (threadIdx.x != 0) return;
  rq_step_scalars(s, state, init, robust);
}

// after pass 2 (and, with M != I, after <g, M g> has been put into state[kGMG] by a dot product): rho (:53), beta (:31)
// (one thread) rho and the next step's beta from <x',Ax'>, <x',Mx'>, <g',g'> (s[0..2]) and <g',Mg'> (gmg_in unless M = I)
__device__ void rq_gradient_scalars(const double* s, double gmg_in, double* __restrict__ state, int m_identity, int init) {
  state[kXAXn] = s[0];
  state[kXMXn] = s[1];
  state[kGG] = s[2];
  state[kRho] = s[0] / s[1];
  const double gmg = m_identity == 1 ? s[2] : gmg_in;
  // the first step takes p = -g (:29-30): beta = 0; afterwards <g,Mg> / <g_old,Mg_old>
  const double prev = state[kGMGprev];
  state[kBeta] = (init == 1 || !(prev != 0.0)) ? 0.0 : gmg / prev;  // (a previous gradient of exact zeros: restart from -g)
  state[kGMGprev] = gmg;
  state[kGMG] = gmg;
}


__global__ void __launch_bounds__(kScalarThreads) k_rq_scalars2(const double* __restrict__ partials, int nblocks, double* __restrict__ state, int m_identity,
                                                           int init) {
  __shared__ double s[4];
  reduce_results(partials, nblocks, m_identity == 2 ? 4 : 3, s);  // m_identity == 2: <g, M g> is result 3 of the partial sums (k_rq_gmg)
  if (threadIdx.x != 0) return;
  rq_gradient_scalars(s, m_identity == 2 ? s[3] : state[kGMG], state, m_identity, init);
}

// ---- a whole rqmin call in ONE launch (levels of at most kRqSmallMax points) -------------------------------------------
// The reference's own problems (RQMin.py: n = 64; every UnitTests case) and the bottom levels of an RQMG cycle are a few
// hundred points: the launches are all the time there is.  One workgroup holds the level, a point per thread, and runs
// initial pair + nu steps start to end: vectors updated in place in global memory (they stay in the CU's L1 / L2), the
// sums reduced in LDS, the scalars of both passes computed by thread 0 into an LDS copy of the state block — the same
// arithmetic as the pass kernels, block barriers where those have kernel boundaries.  Measured per step (MI355X, 3-term
// operator, M != I): ~10 us up to 32^2, where the passes and their scalar kernels take ~25; with four points per thread
// (64^2) 60 us — slower than the passes, hence one point per thread.
//
// (Tried and dropped: the scalars of a pass computed by the LAST workgroup of the pass itself — partial sums stored
// write-through, a ticket counter, `sc1` loads — instead of one-block kernels behind it: 126 launches fewer per 8192^2
// cycle, but the cycle went from 7.7 - 7.9 to 8.2 - 8.4 ms on the same box.  Thousands of workgroups finishing together
// serialise on the one ticket word (k_rq_gmg at 2048 blocks: 23 -> 40 us) and on the small levels ticket + `sc1` loads
// cost what the launch did (6.4 -> 12.6 us).  With __threadfence() in place of the write-through hand-off the
// fine-level passes ran 3.1x slower: an agent-scope release per block writes back the XCD's L2.)
constexpr int kRqSmallThreads = 1024;
constexpr int kRqSmallMax = kRqSmallThreads;

template <int NQ>
__device__ __forceinline__ void small_reduce(const double* acc, double (*s_part)[kRqSmallThreads / 64], double* s_sum) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    double t = acc[q];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) t += __shfl_down(t, d);
    if (lane == 0) s_part[q][wave] = t;
  }
  __syncthreads();
  if ((int)threadIdx.x < NQ) {
    double t = 0.0;
    for (int k = 0; k < kRqSmallThreads / 64; ++k) t += s_part[threadIdx.x][k];
    s_sum[threadIdx.x] = t;
  }
  __syncthreads();
}

__global__ void __launch_bounds__(kRqSmallThreads) k_rq_small(KGrid g, KOp A, KOp Mo, int m_identity, double* x, double* p, double* gv, double* state,
                                                            int nu, int robust) {
  __shared__ double s_part[kRqSums][kRqSmallThreads / 64];
  __shared__ double s_sum[kRqSums];
  __shared__ double s_state[kRqStateWords];
  const long n = g.nr * g.nc;
  if (threadIdx.x < kRqStateWords) s_state[threadIdx.x] = state[threadIdx.x];
  __syncthreads();
  for (int it = -1; it < nu; ++it) {  // (uniform trip counts: every thread reaches every barrier)
    const int init = it < 0 ? 1 : (it == 0 ? 2 : 0);
    // pass 1: p = -g + beta p, then the eight products of {x, p} with {A, M}{x, p}
    if (init != 1) {
      const double beta = init == 2 ? 0.0 : s_state[kBeta];
      for (long k = threadIdx.x; k < n; k += kRqSmallThreads) p[k] = init == 2 ? -gv[k] : -gv[k] + beta * p[k];
      __syncthreads();
    }
    double acc[kRqSums];
#pragma unroll
    for (int q = 0; q < kRqSums; ++q) acc[q] = 0.0;
    for (long k = threadIdx.x; k < n; k += kRqSmallThreads) {
      const long i = k / g.nc, j = k - i * g.nc;
      const double xc = x[k], pc = init == 1 ? 0.0 : p[k];
      const double ax = apply_point(A, 0, x, nullptr, 1.0, 0.0, g.nc, i, j), mx = apply_point(Mo, m_identity, x, nullptr, 1.0, 0.0, g.nc, i, j);
      const double ap = init == 1 ? 0.0 : apply_point(A, 0, p, nullptr, 1.0, 0.0, g.nc, i, j);
      const double mp = init == 1 ? 0.0 : apply_point(Mo, m_identity, p, nullptr, 1.0, 0.0, g.nc, i, j);
      acc[kS_xAx] += xc * ax;
      acc[kS_xAp] += xc * ap;
      acc[kS_pAx] += pc * ax;
      acc[kS_pAp] += pc * ap;
      acc[kS_xMx] += xc * mx;
      acc[kS_xMp] += xc * mp;
      acc[kS_pMx] += pc * mx;
      acc[kS_pMp] += pc * mp;
    }
    small_reduce<kRqSums>(acc, s_part, s_sum);
    if (threadIdx.x == 0) rq_step_scalars(s_sum, s_state, init, robust);
    __syncthreads();
    // pass 2: x += delta p, g = 2 (A x - rho M x), <x, A x>, <x, M x>, <g, g> and (M != I) <g, M g>
    if (init != 1) {
      const double delta = s_state[kDelta];
      for (long k = threadIdx.x; k < n; k += kRqSmallThreads) x[k] = x[k] + delta * p[k];
      __syncthreads();
    }
    const double rho = s_state[kRhoLin];
    double acc2[4] = {0.0, 0.0, 0.0, 0.0};
    for (long k = threadIdx.x; k < n; k += kRqSmallThreads) {
      const long i = k / g.nc, j = k - i * g.nc;
      const double xc = x[k];
      const double ax = apply_point(A, 0, x, nullptr, 1.0, 0.0, g.nc, i, j), mx = apply_point(Mo, m_identity, x, nullptr, 1.0, 0.0, g.nc, i, j);
      const double gg = 2.0 * (ax - rho * mx);
      gv[k] = gg;
      acc2[0] += xc * ax;
      acc2[1] += xc * mx;
      acc2[2] += gg * gg;
    }
    if (!m_identity) {
      __syncthreads();
      for (long k = threadIdx.x; k < n; k += kRqSmallThreads) {
        const long i = k / g.nc, j = k - i * g.nc;
        acc2[3] += gv[k] * apply_point(Mo, 0, gv, nullptr, 1.0, 0.0, g.nc, i, j);
      }
    }
    small_reduce<4>(acc2, s_part, s_sum);
    if (threadIdx.x == 0) rq_gradient_scalars(s_sum, s_sum[3], s_state, m_identity ? 1 : 0, init);
    __syncthreads();
  }
  if (threadIdx.x < kRqStateWords) state[threadIdx.x] = s_state[threadIdx.x];
}

// the step after the first real one must use beta = <g1,Mg1>/<g0,Mg0>: scalars2 of the init pair stores <g0,Mg0> and beta
// = 0 (first step: p = -g); scalars2 of step 1 then finds kGMGprev = <g0,Mg0>.
__global__ void k_rq_store(double* __restrict__ state, int word, const double* __restrict__ value) { state[word] = value[0]; }

template <int MA>
void launch_pass1_ma(hipStream_t s, int mm, dim3 grid, dim3 block, KGrid g, const KOp& A, const KOp& Mo, const double* x, const double* gv,
                     const double* pold, double* pnew, const double* state, int init, int rows, double* partials, int nblocks) {
  if (mm == 0) hipLaunchKernelGGL((k_rq_pass1<MA, 0>), grid, block, 0, s, g, A, Mo, x, gv, pold, pnew, state, init, rows, partials, nblocks);
  else hipLaunchKernelGGL((k_rq_pass1<MA, 1>), grid, block, 0, s, g, A, Mo, x, gv, pold, pnew, state, init, rows, partials, nblocks);
}

template <int MA>
void launch_pass2_ma(hipStream_t s, int mm, dim3 grid, dim3 block, KGrid g, const KOp& A, const KOp& Mo, const double* x, const double* p,
                     double* xnew, double* gout, const double* state, int init, int rows, double* partials, int nblocks) {
  if (mm == 0) hipLaunchKernelGGL((k_rq_pass2<MA, 0>), grid, block, 0, s, g, A, Mo, x, p, xnew, gout, state, init, rows, partials, nblocks);
  else hipLaunchKernelGGL((k_rq_pass2<MA, 1>), grid, block, 0, s, g, A, Mo, x, p, xnew, gout, state, init, rows, partials, nblocks);
}

// which instantiation serves A (see Fac): the 5-point forms where the plan recognised one, the general terms otherwise
int operator_form(const KOp& A) {
  if (A.five_point && A.cn != 0.0) return kFive;
  if (A.five_diag && A.ndiag >= 1 && A.ndiag <= 2) return kFive + A.ndiag;
  return A.nterms < 1 ? 1 : (A.nterms > 4 ? 4 : A.nterms);
}

// rows per block of a march: 32 on big levels (two overlap rows per chunk: 6 % more loads); shorter chunks where the level
// would otherwise not fill the chip — a block's march is one dependent load per row step, ~1 us each (measured: 29 - 50 us
// per pass on every level from 64^2 to 2048^2 with 32-row chunks).  At most 4096 blocks: the partial sums' array.
long march_rows(const KGrid& g, unsigned gx) {
  static const long min_blocks = [] {  // (MGCMT_RQ_MIN_BLOCKS: A/B measurements)
    const char* e = getenv("MGCMT_RQ_MIN_BLOCKS");
    const long v = e ? atol(e) : 0;
    return v > 0 && v <= 4096 ? v : 2048L;
  }();
  long rows = 32;
  while ((long)gx * ((g.nr + rows - 1) / rows) > 4096) rows *= 2;
  while (rows > 2 && (long)gx * ((g.nr + rows - 1) / rows) < min_blocks) rows /= 2;
  return rows;
}

bool march_ok(const KGrid& g, const KOp& A, const KOp& Mo, int m_identity, const double* a, const double* b, const double* c, const double* d,
              const double* e) {
  const uintptr_t all = (uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)d | (uintptr_t)e;
  return g.coarsen_rows && g.nr >= 2 && g.nc >= 2 && (g.nc & 1) == 0 && (all & 15) == 0 && A.nterms >= 1 && A.nterms <= 4 &&
         (m_identity || Mo.nterms == 1);
}

}  // namespace

int rq_state_words() { return kRqStateWords; }
int rq_word_rho() { return kRho; }
int rq_word_gmg() { return kGMG; }

// partials: at least 8 * 4096 doubles
void launch_rq_pass1(hipStream_t s, KGrid g, KOp A, KOp Mo, int m_identity, const double* x, const double* gv, const double* pold, double* pnew,
                     double* state, int init, int robust, double* partials) {
  int nblocks;
  if (march_ok(g, A, Mo, m_identity, x, gv, pold, pnew, x)) {
    const dim3 b(g.nc >= 512 ? kRqThreads : 64, 1, 1);
    const unsigned gx = (unsigned)((g.nc / 2 + b.x - 1) / b.x);
    const long rows = march_rows(g, gx);
    const dim3 grid(gx, (unsigned)((g.nr + rows - 1) / rows), 1);
    nblocks = (int)(grid.x * grid.y);
    const int mm = m_identity ? 0 : 1;
    switch (operator_form(A)) {
      case 1: launch_pass1_ma<1>(s, mm, grid, b, g, A, Mo, x, gv, pold, pnew, state, init, (int)rows, partials, nblocks); break;
      case 2: launch_pass1_ma<2>(s, mm, grid, b, g, A, Mo, x, gv, pold, pnew, state, init, (int)rows, partials, nblocks); break;
      case 3: launch_pass1_ma<3>(s, mm, grid, b, g, A, Mo, x, gv, pold, pnew, state, init, (int)rows, partials, nblocks); break;
      case 4: launch_pass1_ma<4>(s, mm, grid, b, g, A, Mo, x, gv, pold, pnew, state, init, (int)rows, partials, nblocks); break;
      case kFive: launch_pass1_ma<kFive>(s, mm, grid, b, g, A, Mo, x, gv, pold, pnew, state, init, (int)rows, partials, nblocks); break;
      case kFive + 1: launch_pass1_ma<kFive + 1>(s, mm, grid, b, g, A, Mo, x, gv, pold, pnew, state, init, (int)rows, partials, nblocks); break;
      default: launch_pass1_ma<kFive + 2>(s, mm, grid, b, g, A, Mo, x, gv, pold, pnew, state, init, (int)rows, partials, nblocks); break;
    }
  } else {
    const long n = g.nr * g.nc;
    long blocks = (n + kRqThreads - 1) / kRqThreads;
    if (blocks > 1024) blocks = 1024;
    nblocks = (int)blocks;
    hipLaunchKernelGGL(k_rq_pass1_point, dim3((unsigned)blocks), dim3(kRqThreads), 0, s, g, A, Mo, m_identity, x, gv, pold, pnew, state, init, partials, nblocks);
  }
  hipLaunchKernelGGL(k_rq_scalars1, dim3(1), dim3(kScalarThreads), 0, s, partials, nblocks, state, init, robust);
}

// pass 2 without its scalars (the caller may have to put <g, M g> into the state first)
int launch_rq_pass2(hipStream_t s, KGrid g, KOp A, KOp Mo, int m_identity, const double* x, const double* p, double* xnew, double* gout, double* state,
                    int init, double* partials) {
  int nblocks;
  if (march_ok(g, A, Mo, m_identity, x, p, xnew, gout, x)) {
    const dim3 b(g.nc >= 512 ? kRqThreads : 64, 1, 1);
    const unsigned gx = (unsigned)((g.nc / 2 + b.x - 1) / b.x);
    const long rows = march_rows(g, gx);
    const dim3 grid(gx, (unsigned)((g.nr + rows - 1) / rows), 1);
    nblocks = (int)(grid.x * grid.y);
    const int mm = m_identity ? 0 : 1;
    switch (operator_form(A)) {
      case 1: launch_pass2_ma<1>(s, mm, grid, b, g, A, Mo, x, p, xnew, gout, state, init, (int)rows, partials, nblocks); break;
      case 2: launch_pass2_ma<2>(s, mm, grid, b, g, A, Mo, x, p, xnew, gout, state, init, (int)rows, partials, nblocks); break;
      case 3: launch_pass2_ma<3>(s, mm, grid, b, g, A, Mo, x, p, xnew, gout, state, init, (int)rows, partials, nblocks); break;
      case 4: launch_pass2_ma<4>(s, mm, grid, b, g, A, Mo, x, p, xnew, gout, state, init, (int)rows, partials, nblocks); break;
      case kFive: launch_pass2_ma<kFive>(s, mm, grid, b, g, A, Mo, x, p, xnew, gout, state, init, (int)rows, partials, nblocks); break;
      case kFive + 1: launch_pass2_ma<kFive + 1>(s, mm, grid, b, g, A, Mo, x, p, xnew, gout, state, init, (int)rows, partials, nblocks); break;
      default: launch_pass2_ma<kFive + 2>(s, mm, grid, b, g, A, Mo, x, p, xnew, gout, state, init, (int)rows, partials, nblocks); break;
    }
  } else {
    const long n = g.nr * g.nc;
    long blocks = (n + kRqThreads - 1) / kRqThreads;
    if (blocks > 1024) blocks = 1024;
    nblocks = (int)blocks;
    hipLaunchKernelGGL(k_rq_pass2_point, dim3((unsigned)blocks), dim3(kRqThreads), 0, s, g, A, Mo, m_identity, x, p, xnew, gout, state, init, partials, nblocks);
  }
  return nblocks;
}

// <g, M g> into result 3 of pass 2's partial sums (march levels with a one-term M): true when launched — launch_rq_scalars2
// then takes m_identity = 2; false: the caller computes it (application + dot product) into state[rq_word_gmg()]
bool launch_rq_gmg(hipStream_t s, KGrid g, KOp Mo, const double* gv, double* partials, int nblocks) {
  if (!(g.coarsen_rows && g.nr >= 2 && g.nc >= 2 && (g.nc & 1) == 0 && (((uintptr_t)gv) & 15) == 0 && Mo.nterms == 1)) return false;
  const dim3 b(g.nc >= 512 ? kRqThreads : 64, 1, 1);
  const unsigned gx = (unsigned)((g.nc / 2 + b.x - 1) / b.x);
  const long rows = march_rows(g, gx);
  const dim3 grid(gx, (unsigned)((g.nr + rows - 1) / rows), 1);
  if ((int)(grid.x * grid.y) != nblocks) return false;  // (pass 2 took the point form: another block count)
  hipLaunchKernelGGL((k_rq_gmg<1>), grid, b, 0, s, g, Mo, gv, (int)rows, partials, nblocks);
  return true;
}

void launch_rq_scalars2(hipStream_t s, const double* partials, int nblocks, double* state, int m_identity, int init) {
  hipLaunchKernelGGL(k_rq_scalars2, dim3(1), dim3(kScalarThreads), 0, s, partials, nblocks, state, m_identity, init);
}

// the whole call in one launch where the level is small enough (k_rq_small): x holds start vector and result, p and gv
// are work space; false: not taken (the caller runs the passes)
bool launch_rq_small(hipStream_t s, KGrid g, KOp A, KOp Mo, int m_identity, double* x, double* p, double* gv, double* state, int nu, int robust) {
  const long n = g.nr * g.nc;
  static const long max_points = [] {  // (MGCMT_RQ_SMALL_MAX: A/B measurements)
    const char* e = getenv("MGCMT_RQ_SMALL_MAX");
    const long v = e ? atol(e) : 0;
    return v > 0 && v < kRqSmallMax ? v : (long)kRqSmallMax;
  }();
  if (n > max_points || A.nterms < 1 || (!m_identity && Mo.nterms < 1)) return false;
  hipLaunchKernelGGL(k_rq_small, dim3(1), dim3(kRqSmallThreads), 0, s, g, A, Mo, m_identity, x, p, gv, state, nu, robust);
  return true;
}

void launch_rq_store(hipStream_t s, double* state, int word, const double* value) { hipLaunchKernelGGL(k_rq_store, dim3(1), dim3(1), 0, s, state, word, value); }

}  // namespace mgcmt
