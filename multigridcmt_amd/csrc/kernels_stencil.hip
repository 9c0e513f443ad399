// Baseline (one launch per operation) stencil kernels of the multigrid path, fp64, gfx950.
//
// Every level is a rows x cols grid stored row-major (k = i*cols + j) with MGCMT_HALO_ROWS halo
// rows above and below, so a kernel never branches on the row direction: rows -1 and nr hold
// either zeros (global Dirichlet boundary, MGCMTStencilMaker.py:17-21) or a neighbour strip's
// rows.  The column direction is predicated.  The operator is A = sum_m X_m (x) Y_m with
// tridiagonal factors (see include/mgcmt_hip.h); the shift mu of (A - mu I) is read per vector
// from `shifts` (MGCMTSolver.py:287-288, :385-388).
//
// These kernels are the simple, obviously-correct forms; kernels_fused.hip holds the LDS-staged
// row-streaming kernels that the V-cycle uses on large levels.
#include <cstdint>

#include "mgcmt_internal.h"

namespace mgcmt {

namespace {

struct Point {
  double off;   // sum over the 8 (or 4, or 2) neighbours of a_kj v_j
  double diag;  // a_kk without the shift
};

// Neighbour sum and diagonal of the level operator at (i, j) of vector `v`.
__device__ __forceinline__ Point eval_point(const KOp& op, const double* __restrict__ v, long nc, long i, long j) {
  const double* c = v + i * nc + j;
  const bool hw = j > 0, he = j + 1 < nc;
  Point r;
  if (op.five_point) {
    const double w = hw ? c[-1] : 0.0, e = he ? c[1] : 0.0;
    double acc = op.cw * (w + e);
    if (op.cn != 0.0) acc += op.cn * (c[-nc] + c[nc]);
    r.off = acc;
    r.diag = op.c0;
    return r;
  }
  const double n = c[-nc], s = c[nc];
  const double w = hw ? c[-1] : 0.0, e = he ? c[1] : 0.0;
  const double nw = hw ? c[-nc - 1] : 0.0, ne = he ? c[-nc + 1] : 0.0;
  const double sw = hw ? c[nc - 1] : 0.0, se = he ? c[nc + 1] : 0.0;
  double off = 0.0, diag = 0.0;
  for (int m = 0; m < op.nterms; ++m) {
    const double* X = op.X[m] + i;
    const double* Y = op.Y[m] + j;
    const double xl = X[0], xd = X[op.ldx], xu = X[2 * op.ldx];
    const double yl = Y[0], yd = Y[op.ldy], yu = Y[2 * op.ldy];
    const double rn = yl * nw + yd * n + yu * ne;
    const double rc = yl * w + yu * e;
    const double rs = yl * sw + yd * s + yu * se;
    off += xl * rn + xd * rc + xu * rs;
    diag += xd * yd;
  }
  r.off = off;
  r.diag = diag;
  return r;
}

// dst = (A - mu I) src
__global__ void k_apply(KGrid g, KOp op, KVec src, KVec dst, const double* __restrict__ shifts) {
  const long j = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long i = (long)blockIdx.y * blockDim.y + threadIdx.y;
  if (i >= g.nr || j >= g.nc) return;
  const int q = blockIdx.z;
  const double mu = shifts ? shifts[q] : 0.0;
  const double* v = src.p + q * src.stride;
  const Point p = eval_point(op, v, g.nc, i, j);
  dst.p[q * dst.stride + i * g.nc + j] = p.off + (p.diag - mu) * v[i * g.nc + j];
}

// dst = (A - mu I) src for the 5-point operators of a finest 2-D level (constant, or constant plus a product potential
// p(i) q(j) on the diagonal — the square-well Hamiltonian of PotWellSolver.py:150-153 carried to 2-D) as a row march: a
// thread owns two adjacent columns (16-byte accesses) and walks down kApplyRows rows with the rows above / at / below in
// registers, so every value is read from memory once (plus two overlap rows per chunk) and the pass moves its 16 bytes
// per point at streaming speed.  The one-thread-per-point k_apply evaluates all Kronecker terms with their factor
// loads at every point: 0.92 ms per 8192^2 square-well application against 0.2 ms of traffic — a fifth of a
// Rayleigh-quotient iteration of BASELINE config 5.  Same arithmetic order as k_apply's five_point branch.
#ifndef MGCMT_APPLY_ROWS
#define MGCMT_APPLY_ROWS 32
#endif
constexpr int kApplyRows = MGCMT_APPLY_ROWS;
template <int ND>
__global__ void __launch_bounds__(256) k_apply_march(KGrid g, KOp op, KVec src, KVec dst, const double* __restrict__ shifts) {
  const long j = 2 * ((long)blockIdx.x * blockDim.x + threadIdx.x);
  if (j >= g.nc) return;
  const int q = blockIdx.z;
  const long nc = g.nc;
  const long i0 = (long)blockIdx.y * kApplyRows;
  const long i1 = i0 + kApplyRows < g.nr ? i0 + kApplyRows : g.nr;
  const double mu = shifts ? shifts[q] : 0.0;
  const double* __restrict__ v = src.p + q * src.stride;
  double* __restrict__ out = dst.p + q * dst.stride;
  const bool hw = j > 0, he = j + 2 < nc;
  const long jw = hw ? j - 1 : j, je = he ? j + 2 : j + 1;  // (clamped: the value is discarded)
  const double cn = op.cn, cw = op.cw, d0 = op.c0 - mu;
  double qa[ND > 0 ? ND : 1], qb[ND > 0 ? ND : 1];
#pragma unroll
  for (int m = 0; m < ND; ++m) {
    qa[m] = op.dY[m][j];
    qb[m] = op.dY[m][j + 1];
  }
  auto row2 = [&](long i) { return *reinterpret_cast<const double2*>(v + i * nc + j); };
  double2 n = row2(i0 - 1), c = row2(i0);  // row -1 is a halo row: zeros at the global boundary, a neighbour strip's row otherwise
  double w = hw ? v[i0 * nc + jw] : 0.0, e = he ? v[i0 * nc + je] : 0.0;
#pragma unroll 4
  for (long i = i0; i < i1; ++i) {
    const double2 sr = row2(i + 1);
    const double wn = v[(i + 1) * nc + jw], en = v[(i + 1) * nc + je];
    double da = d0, db = d0;
#pragma unroll
    for (int m = 0; m < ND; ++m) {
      const double pm = op.dX[m][i];
      da += pm * qa[m];
      db += pm * qb[m];
    }
    double acc_a = cw * (w + c.y), acc_b = cw * (c.x + e);
    acc_a += cn * (n.x + sr.x);
    acc_b += cn * (n.y + sr.y);
    *reinterpret_cast<double2*>(out + i * nc + j) = make_double2(acc_a + da * c.x, acc_b + db * c.y);
    n = c;
    c = sr;
    w = hw ? wn : 0.0;
    e = he ? en : 0.0;
  }
}

// The five inner products of the 2 x 2 Rayleigh-Ritz problem on span{x, w} (MGCMTSolver.py:44-50 with <x, A x> known) in
// ONE march over x and w: A w is formed in registers row by row as in k_apply_march and never stored.  Per block five
// partial sums (fixed order: deterministic); launch_final_sums adds them up.
template <int ND>
__global__ void __launch_bounds__(256) k_ritz_march(KGrid g, KOp op, const double* __restrict__ x, const double* __restrict__ v, int rows,
                                                   double* __restrict__ partials, int nblocks) {
  __shared__ double s_part[5][4];
  const long j = 2 * ((long)blockIdx.x * blockDim.x + threadIdx.x);
  const long nc = g.nc;
  const long i0 = (long)blockIdx.y * rows;
  const long i1 = i0 + rows < g.nr ? i0 + rows : g.nr;
  double acc[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
  if (j < nc) {
    const bool hw = j > 0, he = j + 2 < nc;
    const long jw = hw ? j - 1 : j, je = he ? j + 2 : j + 1;
    const double cn = op.cn, cw = op.cw, d0 = op.c0;
    double qa[ND > 0 ? ND : 1], qb[ND > 0 ? ND : 1];
#pragma unroll
    for (int m = 0; m < ND; ++m) {
      qa[m] = op.dY[m][j];
      qb[m] = op.dY[m][j + 1];
    }
    auto row2 = [&](const double* p, long i) { return *reinterpret_cast<const double2*>(p + i * nc + j); };
    double2 n = row2(v, i0 - 1), c = row2(v, i0);
    double w = hw ? v[i0 * nc + jw] : 0.0, e = he ? v[i0 * nc + je] : 0.0;
#pragma unroll 2
    for (long i = i0; i < i1; ++i) {
      const double2 sr = row2(v, i + 1);
      const double2 xr = row2(x, i);
      const double wn = v[(i + 1) * nc + jw], en = v[(i + 1) * nc + je];
      double da = d0, db = d0;
#pragma unroll
      for (int m = 0; m < ND; ++m) {
        const double pm = op.dX[m][i];
        da += pm * qa[m];
        db += pm * qb[m];
      }
      double ha = cw * (w + c.y), hb = cw * (c.x + e);
      ha += cn * (n.x + sr.x);
      hb += cn * (n.y + sr.y);
      ha += da * c.x;
      hb += db * c.y;
      acc[0] += xr.x * xr.x + xr.y * xr.y;
      acc[1] += xr.x * c.x + xr.y * c.y;
      acc[2] += c.x * c.x + c.y * c.y;
      acc[3] += xr.x * ha + xr.y * hb;
      acc[4] += c.x * ha + c.y * hb;
      n = c;
      c = sr;
      w = hw ? wn : 0.0;
      e = he ? en : 0.0;
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int q = 0; q < 5; ++q) {
    double t = acc[q];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) t += __shfl_down(t, d);
    if (lane == 0) s_part[q][wave] = t;
  }
  __syncthreads();
  if (threadIdx.x < 5) {
    double t = 0.0;
    const int nw = blockDim.x >> 6;
    for (int k = 0; k < nw; ++k) t += s_part[threadIdx.x][k];
    partials[(long)threadIdx.x * nblocks + (long)blockIdx.y * gridDim.x + blockIdx.x] = t;
  }
}

// The same march for any operator sum_m X_m (x) Y_m of a 2-D level (Galerkin operators R*A*P and mass operators R*M*P
// of the Rayleigh-quotient multigrid, MGCMTSolver.py:99-122 — six applications per rqmin step and level): the factors
// of the thread's two columns in registers, the row's factors by wave-uniform loads, the 3 x 4 neighbourhood in a
// rotating register window.  The expressions are eval_point's, term by term, so the values are k_apply's bit for bit.
template <int M>
__global__ void __launch_bounds__(256) k_apply_march_terms(KGrid g, KOp op, KVec src, KVec dst, const double* __restrict__ shifts) {
  const long j = 2 * ((long)blockIdx.x * blockDim.x + threadIdx.x);
  if (j >= g.nc) return;
  const int q = blockIdx.z;
  const long nc = g.nc;
  const long i0 = (long)blockIdx.y * kApplyRows;
  const long i1 = i0 + kApplyRows < g.nr ? i0 + kApplyRows : g.nr;
  const double mu = shifts ? shifts[q] : 0.0;
  const double* __restrict__ v = src.p + q * src.stride;
  double* __restrict__ out = dst.p + q * dst.stride;
  const bool hw = j > 0, he = j + 2 < nc;
  const long jw = hw ? j - 1 : j, je = he ? j + 2 : j + 1;
  double yl[M][2], yd[M][2], yu[M][2];
#pragma unroll
  for (int m = 0; m < M; ++m)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const double* Y = op.Y[m] + j + c;
      yl[m][c] = Y[0];
      yd[m][c] = Y[op.ldy];
      yu[m][c] = Y[2 * op.ldy];
    }
  struct Row4 {
    double w, a, b, e;  // columns j-1, j, j+1, j+2 (zero outside the grid)
  };
  auto load4 = [&](long i) {
    const double2 c = *reinterpret_cast<const double2*>(v + i * nc + j);
    const double w = v[i * nc + jw], e = v[i * nc + je];
    return Row4{hw ? w : 0.0, c.x, c.y, he ? e : 0.0};
  };
  Row4 n = load4(i0 - 1), c = load4(i0);  // row -1 / nr: halo rows (zeros at the global boundary, a neighbour strip's rows otherwise)
#pragma unroll 2
  for (long i = i0; i < i1; ++i) {
    const Row4 s = load4(i + 1);
    double offa = 0.0, offb = 0.0, da = 0.0, db = 0.0;
#pragma unroll
    for (int m = 0; m < M; ++m) {
      const double* X = op.X[m] + i;
      const double xl = X[0], xd = X[op.ldx], xu = X[2 * op.ldx];
      {
        const double rn = yl[m][0] * n.w + yd[m][0] * n.a + yu[m][0] * n.b;
        const double rc = yl[m][0] * c.w + yu[m][0] * c.b;
        const double rs = yl[m][0] * s.w + yd[m][0] * s.a + yu[m][0] * s.b;
        offa += xl * rn + xd * rc + xu * rs;
        da += xd * yd[m][0];
      }
      {
        const double rn = yl[m][1] * n.a + yd[m][1] * n.b + yu[m][1] * n.e;
        const double rc = yl[m][1] * c.a + yu[m][1] * c.e;
        const double rs = yl[m][1] * s.a + yd[m][1] * s.b + yu[m][1] * s.e;
        offb += xl * rn + xd * rc + xu * rs;
        db += xd * yd[m][1];
      }
    }
    *reinterpret_cast<double2*>(out + i * nc + j) = make_double2(offa + (da - mu) * c.a, offb + (db - mu) * c.b);
    n = c;
    c = s;
  }
}

// weighted Jacobi, out of place:  v' = v + w (f - (A - mu I) v) / d     (MGCMTSolver.py:193-206)
__global__ void k_wjacobi(KGrid g, KOp op, KVec vin, KVec f, KVec vout, const double* __restrict__ shifts, double omega) {
  const long j = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long i = (long)blockIdx.y * blockDim.y + threadIdx.y;
  if (i >= g.nr || j >= g.nc) return;
  const int q = blockIdx.z;
  const double mu = shifts[q];
  const double* v = vin.p + q * vin.stride;
  const Point p = eval_point(op, v, g.nc, i, j);
  const double d = p.diag - mu;
  const double vc = v[i * g.nc + j];
  const double res = f.p[q * f.stride + i * g.nc + j] - (p.off + d * vc);
  vout.p[q * vout.stride + i * g.nc + j] = vc + omega * (res / d);
}

// one colour (ca, cb) = (i%2, j%2) of the multicolour Gauss-Seidel / SOR sweep, in place
__global__ void k_mc_colour(KGrid g, KOp op, KVec vv, KVec f, const double* __restrict__ shifts, double omega, int ca, int cb) {
  const long j = 2 * ((long)blockIdx.x * blockDim.x + threadIdx.x) + cb;
  const long i = 2 * ((long)blockIdx.y * blockDim.y + threadIdx.y) + ca;
  if (i >= g.nr || j >= g.nc) return;
  const int q = blockIdx.z;
  const double mu = shifts[q];
  double* v = vv.p + q * vv.stride;
  const Point p = eval_point(op, v, g.nc, i, j);
  const double d = p.diag - mu;
  const double vc = v[i * g.nc + j];
  const double res = f.p[q * f.stride + i * g.nc + j] - (p.off + d * vc);
  v[i * g.nc + j] = vc + omega * (res / d);
}

// r = f - (A - mu I) v                                                   (MGCMTSolver.py:315)
__global__ void k_residual(KGrid g, KOp op, KVec vv, KVec f, KVec r, const double* __restrict__ shifts) {
  const long j = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long i = (long)blockIdx.y * blockDim.y + threadIdx.y;
  if (i >= g.nr || j >= g.nc) return;
  const int q = blockIdx.z;
  const double mu = shifts[q];
  const double* v = vv.p + q * vv.stride;
  const Point p = eval_point(op, v, g.nc, i, j);
  r.p[q * r.stride + i * g.nc + j] = f.p[q * f.stride + i * g.nc + j] - (p.off + (p.diag - mu) * v[i * g.nc + j]);
}

// full weighting: rows/cols 2I..2I+2 with weights (1/4, 1/2, 1/4)      (MGCMTStencilMaker.py:57-78)
__global__ void k_restrict(KGrid fine, KGrid coarse, KVec r, KVec rc) {
  const long J = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long I = (long)blockIdx.y * blockDim.y + threadIdx.y;
  if (I >= coarse.nr || J >= coarse.nc) return;
  const int q = blockIdx.z;
  const double* p = r.p + q * r.stride;
  const long nc = fine.nc;
  const long j0 = 2 * J;
  const bool h2 = j0 + 2 < nc;
  double out;
  if (fine.coarsen_rows) {
    const double* a = p + (2 * I) * nc + j0;
    const double* b = a + nc;
    const double* c = b + nc;  // row 2I+2: the halo row when 2I+2 == nr
    const double ra = 0.25 * a[0] + 0.5 * a[1] + (h2 ? 0.25 * a[2] : 0.0);
    const double rb = 0.25 * b[0] + 0.5 * b[1] + (h2 ? 0.25 * b[2] : 0.0);
    const double rcw = 0.25 * c[0] + 0.5 * c[1] + (h2 ? 0.25 * c[2] : 0.0);
    out = 0.25 * ra + 0.5 * rb + 0.25 * rcw;
  } else {
    const double* a = p + I * nc + j0;
    out = 0.25 * a[0] + 0.5 * a[1] + (h2 ? 0.25 * a[2] : 0.0);
  }
  rc.p[q * rc.stride + I * coarse.nc + J] = out;
}

// linear / bilinear interpolation: odd fine index takes c[(k-1)/2], even takes the mean of
// c[k/2-1] and c[k/2] (c[-1] = 0)                                        (MGCMTStencilMaker.py:27-54)
__global__ void k_prolong(KGrid fine, KGrid coarse, KVec e, KVec vv, int accumulate) {
  const long j = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long i = (long)blockIdx.y * blockDim.y + threadIdx.y;
  if (i >= fine.nr || j >= fine.nc) return;
  const int q = blockIdx.z;
  const double* c = e.p + q * e.stride;
  const long cnc = coarse.nc;
  const long J = j >> 1;
  const bool jodd = j & 1;
  double val;
  if (fine.coarsen_rows) {
    const long I = i >> 1;
    const double* r1 = c + I * cnc;
    const double a1 = jodd ? r1[J] : 0.5 * (r1[J] + (J > 0 ? r1[J - 1] : 0.0));
    if (i & 1) {
      val = a1;
    } else {
      const double* r0 = r1 - cnc;  // coarse row I-1: the halo row when I == 0
      const double a0 = jodd ? r0[J] : 0.5 * (r0[J] + (J > 0 ? r0[J - 1] : 0.0));
      val = 0.5 * (a0 + a1);
    }
  } else {
    const double* r1 = c + i * cnc;
    val = jodd ? r1[J] : 0.5 * (r1[J] + (J > 0 ? r1[J - 1] : 0.0));
  }
  double* dst = vv.p + q * vv.stride + i * fine.nc + j;
  *dst = accumulate ? *dst + val : val;
}

// the same for 2-D levels of even size: a thread owns the 2 x 2 fine points of coarse point (I, J) — four coarse values
// (rows I - 1, I; columns J - 1, J) in, two 16-byte accesses per fine row out; the arithmetic per point is k_prolong's
// (one thread per point there: 8-byte accesses and four coarse loads each — 3.6 TB/s on an 8192^2 level; 1.2 GB per launch)
__global__ void __launch_bounds__(256) k_prolong_quad(KGrid fine, KGrid coarse, KVec e, KVec vv, int accumulate) {
  const long J = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long I = (long)blockIdx.y * blockDim.y + threadIdx.y;
  const long cnc = coarse.nc;
  if (I >= coarse.nr || J >= cnc) return;
  const int q = blockIdx.z;
  const double* r1 = e.p + q * e.stride + I * cnc;
  const double* r0 = r1 - cnc;  // coarse row I-1: the halo row when I == 0
  const double c11 = r1[J], c10 = J > 0 ? r1[J - 1] : 0.0;
  const double c01 = r0[J], c00 = J > 0 ? r0[J - 1] : 0.0;
  const double a1e = 0.5 * (c11 + c10), a1o = c11;  // row I at the even / odd fine column
  const double a0e = 0.5 * (c01 + c00), a0o = c01;  // row I - 1
  double2 top = make_double2(0.5 * (a0e + a1e), 0.5 * (a0o + a1o));  // fine row 2I
  double2 bot = make_double2(a1e, a1o);                              // fine row 2I + 1
  double* d0 = vv.p + q * vv.stride + (2 * I) * fine.nc + 2 * J;
  double* d1 = d0 + fine.nc;
  if (accumulate) {
    const double2 t = *reinterpret_cast<const double2*>(d0), b = *reinterpret_cast<const double2*>(d1);
    top = make_double2(t.x + top.x, t.y + top.y);
    bot = make_double2(b.x + bot.x, b.y + bot.y);
  }
  *reinterpret_cast<double2*>(d0) = top;
  *reinterpret_cast<double2*>(d1) = bot;
}

inline dim3 grid2d(long nc, long nr, int k, dim3 b) {
  return dim3((unsigned)((nc + b.x - 1) / b.x), (unsigned)((nr + b.y - 1) / b.y), (unsigned)k);
}

inline dim3 block_for(long nr) { return nr == 1 ? dim3(256, 1, 1) : dim3(64, 4, 1); }

}  // namespace

void launch_apply(hipStream_t s, KGrid g, KOp op, KVec src, KVec dst, const double* shifts, int k) {
  const bool aligned = (((uintptr_t)src.p | (uintptr_t)dst.p) & 15) == 0 && (src.stride & 1) == 0 && (dst.stride & 1) == 0;
  if (g.coarsen_rows && g.nr >= 2 && g.nc >= 2 && (g.nc & 1) == 0 && aligned && ((op.five_point && op.cn != 0.0) || (op.five_diag && op.ndiag <= 2))) {
    const dim3 b(g.nc >= 512 ? 256 : 64, 1, 1);
    const dim3 grid((unsigned)((g.nc / 2 + b.x - 1) / b.x), (unsigned)((g.nr + kApplyRows - 1) / kApplyRows), (unsigned)k);
    if (op.five_point) hipLaunchKernelGGL(k_apply_march<0>, grid, b, 0, s, g, op, src, dst, shifts);
    else if (op.ndiag == 1) hipLaunchKernelGGL(k_apply_march<1>, grid, b, 0, s, g, op, src, dst, shifts);
    else hipLaunchKernelGGL(k_apply_march<2>, grid, b, 0, s, g, op, src, dst, shifts);
    return;
  }
  if (g.coarsen_rows && g.nr >= 2 && g.nc >= 2 && (g.nc & 1) == 0 && aligned && op.nterms >= 1 && op.nterms <= 4 && !op.five_diag) {
    const dim3 b(g.nc >= 512 ? 256 : 64, 1, 1);
    const dim3 grid((unsigned)((g.nc / 2 + b.x - 1) / b.x), (unsigned)((g.nr + kApplyRows - 1) / kApplyRows), (unsigned)k);
    switch (op.nterms) {
      case 1: hipLaunchKernelGGL(k_apply_march_terms<1>, grid, b, 0, s, g, op, src, dst, shifts); break;
      case 2: hipLaunchKernelGGL(k_apply_march_terms<2>, grid, b, 0, s, g, op, src, dst, shifts); break;
      case 3: hipLaunchKernelGGL(k_apply_march_terms<3>, grid, b, 0, s, g, op, src, dst, shifts); break;
      default: hipLaunchKernelGGL(k_apply_march_terms<4>, grid, b, 0, s, g, op, src, dst, shifts); break;
    }
    return;
  }
  const dim3 b = block_for(g.nr);
  hipLaunchKernelGGL(k_apply, grid2d(g.nc, g.nr, k, b), b, 0, s, g, op, src, dst, shifts);
}

bool launch_ritz_pair(hipStream_t s, KGrid g, KOp op, const double* x, const double* w, double* partials, double* out) {
  const bool aligned = (((uintptr_t)x | (uintptr_t)w) & 15) == 0;
  if (!(g.coarsen_rows && g.nr >= 2 && g.nc >= 2 && (g.nc & 1) == 0 && aligned && ((op.five_point && op.cn != 0.0) || (op.five_diag && op.ndiag <= 2))))
    return false;
  const dim3 b(g.nc >= 512 ? 256 : 64, 1, 1);
  const unsigned gx = (unsigned)((g.nc / 2 + b.x - 1) / b.x);
  // row chunks as short as the partial-sum buffer allows (5 x 8192 doubles), not shorter than an application's
  long rows = kApplyRows;
  while ((long)gx * ((g.nr + rows - 1) / rows) > 8192) rows *= 2;
  const dim3 grid(gx, (unsigned)((g.nr + rows - 1) / rows), 1);
  const int nblocks = (int)(grid.x * grid.y);
  if (op.five_point) hipLaunchKernelGGL(k_ritz_march<0>, grid, b, 0, s, g, op, x, w, (int)rows, partials, nblocks);
  else if (op.ndiag == 1) hipLaunchKernelGGL(k_ritz_march<1>, grid, b, 0, s, g, op, x, w, (int)rows, partials, nblocks);
  else hipLaunchKernelGGL(k_ritz_march<2>, grid, b, 0, s, g, op, x, w, (int)rows, partials, nblocks);
  launch_final_sums(s, 5, nblocks, partials, out);
  return true;
}

void launch_wjacobi(hipStream_t s, KGrid g, KOp op, KVec vin, KVec f, KVec vout, const double* shifts, double omega, int k) {
  const dim3 b = block_for(g.nr);
  hipLaunchKernelGGL(k_wjacobi, grid2d(g.nc, g.nr, k, b), b, 0, s, g, op, vin, f, vout, shifts, omega);
}

void launch_mc_colour(hipStream_t s, KGrid g, KOp op, KVec v, KVec f, const double* shifts, double omega, int ca, int cb, int k) {
  const long rows = (g.nr - ca + 1) / 2, cols = (g.nc - cb + 1) / 2;
  if (rows <= 0 || cols <= 0) return;
  const dim3 b = block_for(g.nr);
  hipLaunchKernelGGL(k_mc_colour, grid2d(cols, rows, k, b), b, 0, s, g, op, v, f, shifts, omega, ca, cb);
}

void launch_residual(hipStream_t s, KGrid g, KOp op, KVec v, KVec f, KVec r, const double* shifts, int k) {
  const dim3 b = block_for(g.nr);
  hipLaunchKernelGGL(k_residual, grid2d(g.nc, g.nr, k, b), b, 0, s, g, op, v, f, r, shifts);
}

void launch_restrict(hipStream_t s, KGrid fine, KGrid coarse, KVec r, KVec rc, int k) {
  const dim3 b = block_for(coarse.nr);
  hipLaunchKernelGGL(k_restrict, grid2d(coarse.nc, coarse.nr, k, b), b, 0, s, fine, coarse, r, rc);
}

void launch_prolong(hipStream_t s, KGrid fine, KGrid coarse, KVec e, KVec v, int accumulate, int k) {
  const bool quads = fine.coarsen_rows && fine.nr == 2 * coarse.nr && fine.nc == 2 * coarse.nc && (((uintptr_t)v.p) & 15) == 0 && (v.stride & 1) == 0;
  if (quads) {
    const dim3 b(64, 4, 1);
    hipLaunchKernelGGL(k_prolong_quad, grid2d(coarse.nc, coarse.nr, k, b), b, 0, s, fine, coarse, e, v, accumulate);
    return;
  }
  const dim3 b = block_for(fine.nr);
  hipLaunchKernelGGL(k_prolong, grid2d(fine.nc, fine.nr, k, b), b, 0, s, fine, coarse, e, v, accumulate);
}

}  // namespace mgcmt
