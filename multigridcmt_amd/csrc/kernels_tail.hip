// The tail of a V-cycle in ONE launch: every level of at most 32 x 32 points, from the first such level down to the
// coarsest grid and back up (MGCMTSolver.py:310-326 for those levels, the direct solve of :305-308 as a product
// with the explicit inverse), held in LDS by one 1024-thread workgroup per vector.  On these levels a launch costs
// more than its work: five launches (two fused passes per level and the coarse solve, ~43 us under graph replay)
// become one of ~30 us, paced by its ~45 barrier-separated phases.  (Measured: starting the tail at 64 x 64 does not
// pay — one CU's LDS bandwidth then costs what the launches did.)  The arithmetic is that of the one-launch-per-operation kernels (kernels_stencil.hip): the general
// separable operator sum_m X_m (x) Y_m from the level's factor arrays, weighted Jacobi or four-colour Gauss-Seidel.
#include "mgcmt_internal.h"

namespace mgcmt {

namespace {

constexpr int kTailThreads = 1024;
constexpr int kTailMaxGrid = 32;
// 2 x sum g_l^2 (V, F) + g_0^2 (scratch) + factors: 2*1365 + 1024 + 4 terms * 6 * 63 < 5400 doubles (42 KiB)
constexpr int kTailLds = 5400;

struct Layout {
  int g0, nlev, nterms;
  __device__ __forceinline__ int g(int l) const { return g0 >> l; }
  __device__ __forceinline__ int s2(int l) const {  // sum of g_i^2, i < l
    int t = 0;
    for (int i = 0; i < l; ++i) t += (g0 >> i) * (g0 >> i);
    return t;
  }
  __device__ __forceinline__ int s1(int l) const {  // sum of g_i, i < l
    int t = 0;
    for (int i = 0; i < l; ++i) t += g0 >> i;
    return t;
  }
  __device__ __forceinline__ int v(int l) const { return s2(l); }
  __device__ __forceinline__ int f(int l) const { return s2(nlev) + s2(l); }
  __device__ __forceinline__ int t() const { return 2 * s2(nlev); }
  // factor `which` (0 = X, 1 = Y) of term m on level l: three parts (lower, diag, upper) of g_l numbers each
  __device__ __forceinline__ int fac(int l, int m, int which) const { return 2 * s2(nlev) + g0 * g0 + s1(l) * nterms * 6 + (m * 2 + which) * 3 * g(l); }
};

struct Point {
  double off, diag;
};

// one level as the point loops see it: everything that depends on the level only, computed once per phase
struct View {
  int g, nterms;
  double* V;
  const double* F;
  const double* X[kMaxTerms];  // lower | diag | upper, g numbers each
  const double* Y[kMaxTerms];
};

__device__ __forceinline__ View make_view(double* sm, const Layout& L, int l) {
  View w;
  w.g = L.g(l);
  w.nterms = L.nterms;
  w.V = sm + L.v(l);
  w.F = sm + L.f(l);
  const int base = L.fac(l, 0, 0);
#pragma unroll
  for (int m = 0; m < kMaxTerms; ++m) {
    w.X[m] = sm + base + (m * 2 + 0) * 3 * w.g;
    w.Y[m] = sm + base + (m * 2 + 1) * 3 * w.g;
  }
  return w;
}

// neighbour sum and diagonal at (i, j) of a g x g level held in LDS (zero Dirichlet ghosts), as eval_point of
// kernels_stencil.hip
__device__ __forceinline__ Point tail_point(const View& w, const double* v, int i, int j) {
  const int g = w.g;
  const bool hn = i > 0, hs = i + 1 < g, hw = j > 0, he = j + 1 < g;
  const double* c = v + i * g + j;
  const double n = hn ? c[-g] : 0.0, s = hs ? c[g] : 0.0, wv = hw ? c[-1] : 0.0, e = he ? c[1] : 0.0;
  const double nw = hn && hw ? c[-g - 1] : 0.0, ne = hn && he ? c[-g + 1] : 0.0;
  const double sw = hs && hw ? c[g - 1] : 0.0, se = hs && he ? c[g + 1] : 0.0;
  Point r{0.0, 0.0};
#pragma unroll
  for (int m = 0; m < kMaxTerms; ++m) {
    if (m < w.nterms) {
      const double* X = w.X[m] + i;
      const double* Y = w.Y[m] + j;
      const double xl = X[0], xd = X[g], xu = X[2 * g];
      const double yl = Y[0], yd = Y[g], yu = Y[2 * g];
      const double rn = yl * nw + yd * n + yu * ne;
      const double rc = yl * wv + yu * e;
      const double rs = yl * sw + yd * s + yu * se;
      r.off += xl * rn + xd * rc + xu * rs;
      r.diag += xd * yd;
    }
  }
  return r;
}

// nu sweeps on a level (V in LDS, in place for the caller): weighted Jacobi through the scratch array, or four-colour
// Gauss-Seidel in the order (0,1),(1,0),(0,0),(1,1) of kernels_stencil.hip
__device__ void tail_smooth(const View& w, double* scratch, int kind, int nu, double omega, double mu) {
  const int g = w.g, n = g * g;
  const int shift = 31 - __builtin_clz(g);  // g is a power of two
  double* V = w.V;
  const double* F = w.F;
  if (kind == MGCMT_WJACOBI) {
    double* cur = V;
    double* nxt = scratch;
    for (int it = 0; it < nu; ++it) {
      for (int p = threadIdx.x; p < n; p += kTailThreads) {
        const int i = p >> shift, j = p & (g - 1);
        const Point pt = tail_point(w, cur, i, j);
        const double d = pt.diag - mu;
        const double vc = cur[p];
        nxt[p] = vc + omega * ((F[p] - (pt.off + d * vc)) / d);
      }
      __syncthreads();
      double* t = cur;
      cur = nxt;
      nxt = t;
    }
    if (cur != V) {
      for (int p = threadIdx.x; p < n; p += kTailThreads) V[p] = cur[p];
      __syncthreads();
    }
    return;
  }
  const int h = g >> 1, hshift = shift - 1;  // g >= 2
  for (int it = 0; it < nu; ++it) {
    for (int c = 0; c < 4; ++c) {
      const int ca = (c == 1 || c == 3) ? 1 : 0, cb = (c == 0 || c == 3) ? 1 : 0;
      for (int p = threadIdx.x; p < h * h; p += kTailThreads) {
        const int i = 2 * (p >> hshift) + ca, j = 2 * (p & (h - 1)) + cb;
        const Point pt = tail_point(w, V, i, j);
        const double d = pt.diag - mu;
        const double vc = V[i * g + j];
        V[i * g + j] = vc + omega * ((F[i * g + j] - (pt.off + d * vc)) / d);
      }
      __syncthreads();
    }
  }
}

__global__ void __launch_bounds__(kTailThreads) k_tail(TailArgs a) {
  __shared__ double sm[kTailLds];
  // unit_rhs: block b runs the tail on the unit vector e_b with vector `unit_q`'s shift and inverse — the columns of the
  // tail's matrix (launch_tail_matrix)
  const int q = a.unit_rhs ? a.unit_q : blockIdx.x;
  const int out_q = blockIdx.x;
  const Layout L{a.g0, a.nlev, a.nterms};
  const double mu = a.shifts[q];
  // operator factors of every tail level
  for (int l = 0; l < L.nlev; ++l) {
    const int g = L.g(l);
    for (int m = 0; m < L.nterms; ++m)
      for (int x = threadIdx.x; x < 3 * g; x += kTailThreads) {
        const int part = x / g, i = x - part * g;
        sm[L.fac(l, m, 0) + x] = a.X[l][m][part * a.ldx[l] + i];
        sm[L.fac(l, m, 1) + x] = a.Y[l][m][part * a.ldy[l] + i];
      }
  }
  {  // entry level: right-hand side from memory, zero start value (the error equation, MGCMTSolver.py:316)
    const int n = L.g0 * L.g0;
    const double* f = a.unit_rhs ? nullptr : a.f_in + q * a.vstride;
    for (int p = threadIdx.x; p < n; p += kTailThreads) {
      sm[L.f(0) + p] = f ? f[p] : (p == out_q ? 1.0 : 0.0);
      sm[L.v(0) + p] = 0.0;
    }
  }
  __syncthreads();
  double* scratch = sm + L.t();
  for (int l = 0; l + 1 < L.nlev; ++l) {
    const View w = make_view(sm, L, l);
    const int g = w.g, gc = g >> 1;
    const int shift = 31 - __builtin_clz(g);
    tail_smooth(w, scratch, a.kind, a.nu, a.omega, mu);
    // residual into the scratch array, then full weighting (rows / columns 2I..2I+2, weights 1/4 1/2 1/4)
    double* R = scratch;
    const double* V = w.V;
    const double* F = w.F;
    for (int p = threadIdx.x; p < g * g; p += kTailThreads) {
      const int i = p >> shift, j = p & (g - 1);
      const Point pt = tail_point(w, V, i, j);
      R[p] = F[p] - (pt.off + (pt.diag - mu) * V[p]);
    }
    __syncthreads();
    double* Fc = sm + L.f(l + 1);
    double* Vc = sm + L.v(l + 1);
    for (int p = threadIdx.x; p < gc * gc; p += kTailThreads) {
      const int I = p >> (shift - 1), J = p & (gc - 1);
      const int j0 = 2 * J;
      const bool h2 = j0 + 2 < g;
      double rows[3];
      for (int r = 0; r < 3; ++r) {
        const int i = 2 * I + r;
        if (i < g) {
          const double* x = R + i * g + j0;
          rows[r] = 0.25 * x[0] + 0.5 * x[1] + (h2 ? 0.25 * x[2] : 0.0);
        } else {
          rows[r] = 0.0;
        }
      }
      Fc[p] = 0.25 * rows[0] + 0.5 * rows[1] + 0.25 * rows[2];
      Vc[p] = 0.0;
    }
    __syncthreads();
  }
  {  // coarsest grid: x = (A - mu I)^-1 f with the explicit inverse (row i contiguous)
    // (16 lanes share a row; the partial sums meet in a fixed order, so the result does not depend on scheduling)
    const int l = L.nlev - 1, n = L.g(l) * L.g(l);
    const double* inv = a.inv + q * a.inv_stride;
    const double* F = sm + L.f(l);
    double* Vl = sm + L.v(l);
    const int sub = threadIdx.x & 15;
    for (int i0 = 0; i0 < n; i0 += kTailThreads / 16) {  // (every thread makes the same number of trips)
      const int i = i0 + (threadIdx.x >> 4);
      double acc = 0.0;
      if (i < n) {
        const double* row = inv + (long)i * n;
        for (int j = sub; j < n; j += 16) acc += row[j] * F[j];
      }
      acc += __shfl_xor(acc, 8);
      acc += __shfl_xor(acc, 4);
      acc += __shfl_xor(acc, 2);
      acc += __shfl_xor(acc, 1);
      if (sub == 0 && i < n) Vl[i] = acc;
    }
    __syncthreads();
  }
  for (int l = L.nlev - 2; l >= 0; --l) {
    const View w = make_view(sm, L, l);
    const int g = w.g, gc = g >> 1;
    const int shift = 31 - __builtin_clz(g);
    double* V = w.V;
    const double* C = sm + L.v(l + 1);
    // V += P C: odd fine index takes c[(k-1)/2], even the mean of c[k/2-1] and c[k/2] (c[-1] = 0)
    for (int p = threadIdx.x; p < g * g; p += kTailThreads) {
      const int i = p >> shift, j = p & (g - 1);
      const int I = i >> 1, J = j >> 1;
      const bool jodd = j & 1;
      const double* r1 = C + I * gc;
      const double a1 = jodd ? r1[J] : 0.5 * (r1[J] + (J > 0 ? r1[J - 1] : 0.0));
      double val = a1;
      if (!(i & 1)) {
        double a0 = 0.0;
        if (I > 0) {
          const double* r0 = r1 - gc;
          a0 = jodd ? r0[J] : 0.5 * (r0[J] + (J > 0 ? r0[J - 1] : 0.0));
        }
        val = 0.5 * (a0 + a1);
      }
      V[p] += val;
    }
    __syncthreads();
    tail_smooth(w, scratch, a.kind, a.nu, a.omega, mu);
  }
  {
    const int n = L.g0 * L.g0;
    double* v = a.v_out + out_q * a.vstride;
    for (int p = threadIdx.x; p < n; p += kTailThreads) v[p] = sm[L.v(0) + p];
  }
}

// v = T f with the tail's matrix stored by columns (column j = the tail applied to e_j: mt[j * n + i]): each block owns 16
// outputs and reads their 128-byte pieces of all n columns — 64 column groups of 16 threads, the partial sums combined
// through LDS in a fixed order.  n = g0^2 <= 1024, a multiple of 16.
constexpr int kDenseOut = 16, kDenseGroups = 64;
__global__ void __launch_bounds__(kDenseOut* kDenseGroups) k_tail_dense(int n, const double* __restrict__ mt, long mt_stride, const double* __restrict__ f_in,
                                                                       double* __restrict__ v_out, long vstride) {
  __shared__ double s_part[kDenseGroups][kDenseOut + 1];
  const int q = blockIdx.y;
  const int ii = threadIdx.x & (kDenseOut - 1), jj = threadIdx.x / kDenseOut;
  const int i = blockIdx.x * kDenseOut + ii;
  const double* m = mt + q * mt_stride;
  const double* f = f_in + q * vstride;
  double acc = 0.0;
  for (int j = jj; j < n; j += kDenseGroups) acc = fma(m[(long)j * n + i], f[j], acc);
  s_part[jj][ii] = acc;
  __syncthreads();
  if (jj == 0) {
    double t = 0.0;
    for (int g = 0; g < kDenseGroups; ++g) t += s_part[g][ii];
    v_out[q * vstride + i] = t;
  }
}

}  // namespace

bool tail_fits(long g0, int nlev, int nterms) {
  if (g0 > kTailMaxGrid || nlev < 2 || nlev > kTailMaxLevels || nterms > kMaxTerms) return false;
  long s2 = 0, s1 = 0;
  for (int l = 0; l < nlev; ++l) {
    s2 += (g0 >> l) * (g0 >> l);
    s1 += g0 >> l;
  }
  return 2 * s2 + g0 * g0 + s1 * nterms * 6 <= kTailLds;
}

void launch_tail(hipStream_t s, const TailArgs& a, int k) { hipLaunchKernelGGL(k_tail, dim3((unsigned)k), dim3(kTailThreads), 0, s, a); }

// columns of the tail's matrix for vector q's shift: mt[j * n + i] = (tail e_j)[i], n = g0^2
void launch_tail_matrix(hipStream_t s, TailArgs a, int q, double* mt) {
  const int n = a.g0 * a.g0;
  a.unit_rhs = 1;
  a.unit_q = q;
  a.f_in = nullptr;
  a.v_out = mt;
  a.vstride = n;
  hipLaunchKernelGGL(k_tail, dim3((unsigned)n), dim3(kTailThreads), 0, s, a);
}

bool tail_dense_fits(long g0) { return g0 * g0 <= 1024 && (g0 * g0) % kDenseOut == 0; }

void launch_tail_dense(hipStream_t s, long g0, const double* mt, long mt_stride, const double* f_in, double* v_out, long vstride, int k) {
  const int n = (int)(g0 * g0);
  hipLaunchKernelGGL(k_tail_dense, dim3((unsigned)(n / kDenseOut), (unsigned)k), dim3(kDenseOut * kDenseGroups), 0, s, n, mt, mt_stride, f_in, v_out, vstride);
}

}  // namespace mgcmt
