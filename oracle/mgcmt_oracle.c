/*
 * mgcmt_oracle.c — matrix-free CPU restatement of the reference's multigrid path in plain C.
 *
 * TEST INFRASTRUCTURE — NOT PRODUCT CODE.  Built by oracle/Makefile into oracle/_build/libmgcmt_oracle.so
 * and called (through oracle/structured.py) only by tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py.  Nothing under multigridcmt_amd/ links or loads it.
 *
 * What it restates (paths relative to the reference root):
 *   laplacian / interpolation / restriction          MGCMTStencilMaker.py:15-78
 *   wjacobi / gseidel / sor                          MGCMTSolver.py:182-246
 *   vcycle (Galerkin R*A*P, shift kept apart)        MGCMTSolver.py:281-329
 * on operators written as  A = sum_m X_m (x) Y_m  with tridiagonal factors (X over rows, Y over
 * columns; 1-D problems have one row).  R*A*P of such an operator is sum_m (R1 X_m P1) (x) (R1 Y_m P1)
 * with the 1-D full weighting R1 = (1/4,1/2,1/4) on fine 2I..2I+2 and P1 = 2 R1^T, so every level keeps
 * the form.  Parity: pinned through tests/test_oracle_structured.py, which checks every function here
 * against oracle/sparse_ref.py (itself pinned by the reference's golden vectors) on assembled matrices.
 *
 * Factors are [nterms][3][n] arrays: lower, diagonal, upper.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
  int dim;      /* 1 or 2 */
  long nr, nc;  /* rows (1 for dim 1), columns */
  int nterms;
  const double* X; /* [nterms][3][nr], NULL when nr == 1 */
  const double* Y; /* [nterms][3][nc] */
  double shift;
  int five;          /* constant 5-point (3-point) operator: c0, cn, cw valid */
  double c0, cn, cw;
} mgo_level;

void mgo_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

int mgo_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

static inline double xf(const mgo_level* L, int m, int part, long i) {
  if (L->nr == 1) return part == 1 ? 1.0 : 0.0;
  return L->X[((long)m * 3 + part) * L->nr + i];
}
static inline double yf(const mgo_level* L, int m, int part, long j) { return L->Y[((long)m * 3 + part) * L->nc + j]; }

static inline double at(const mgo_level* L, const double* v, long i, long j) {
  if (i < 0 || i >= L->nr || j < 0 || j >= L->nc) return 0.0;
  return v[i * L->nc + j];
}

/* coefficient of the level operator (without shift) coupling (i,j) to (i+di, j+dj) */
static inline double coef(const mgo_level* L, long i, long j, int di, int dj) {
  double c = 0.0;
  for (int m = 0; m < L->nterms; ++m) c += xf(L, m, di + 1, i) * yf(L, m, dj + 1, j);
  return c;
}

/* constant-coefficient detection (the scaled Laplacian of MGCMTStencilMaker.py:15-25 on the finest level) */
static void detect_five(mgo_level* L) {
  double c[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  L->five = 0;
  for (int m = 0; m < L->nterms; ++m) {
    double x[3], y[3];
    for (int p = 0; p < 3; ++p) {
      x[p] = xf(L, m, p, L->nr > 1 ? 1 - (p == 2) : 0);
      y[p] = yf(L, m, p, L->nc > 1 ? 1 - (p == 2) : 0);
      if (L->nr == 1) x[p] = p == 1 ? 1.0 : 0.0;
    }
    for (long i = 0; i < L->nr && L->nr > 1; ++i)
      if ((i > 0 && xf(L, m, 0, i) != x[0]) || xf(L, m, 1, i) != x[1] || (i + 1 < L->nr && xf(L, m, 2, i) != x[2])) return;
    for (long j = 0; j < L->nc; ++j)
      if ((j > 0 && yf(L, m, 0, j) != y[0]) || yf(L, m, 1, j) != y[1] || (j + 1 < L->nc && yf(L, m, 2, j) != y[2])) return;
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) c[a][b] += x[a] * y[b];
  }
  if (c[0][0] != 0 || c[0][2] != 0 || c[2][0] != 0 || c[2][2] != 0 || c[0][1] != c[2][1] || c[1][0] != c[1][2]) return;
  L->five = 1;
  L->c0 = c[1][1];
  L->cn = L->nr > 1 ? c[0][1] : 0.0;
  L->cw = c[1][0];
}

/* (A v)(i,j) split into the diagonal coefficient and the off-diagonal sum */
static inline void row_parts(const mgo_level* L, const double* v, long i, long j, double* diag, double* off) {
  if (L->five) {
    const long nc = L->nc;
    const double* c = v + i * nc + j;
    double o = L->cw * ((j > 0 ? c[-1] : 0.0) + (j + 1 < nc ? c[1] : 0.0));
    if (L->cn != 0.0) o += L->cn * ((i > 0 ? c[-nc] : 0.0) + (i + 1 < L->nr ? c[nc] : 0.0));
    *diag = L->c0 - L->shift;
    *off = o;
    return;
  }
  double d = 0.0, o = 0.0;
  for (int di = -1; di <= 1; ++di) {
    if (L->nr == 1 && di != 0) continue;
    for (int dj = -1; dj <= 1; ++dj) {
      const double c = coef(L, i, j, di, dj);
      if (di == 0 && dj == 0) d = c;
      else if (c != 0.0) o += c * at(L, v, i + di, j + dj);
    }
  }
  *diag = d - L->shift;
  *off = o;
}

/* one tridiagonal factor through the Galerkin product R1 * T * P1 */
void mgo_galerkin(const double* fine, long n, double* coarse) {
  const long nc = n / 2;
  const double rw[3] = {0.25, 0.5, 0.25}, pw[3] = {0.5, 1.0, 0.5};
  memset(coarse, 0, sizeof(double) * 3 * nc);
  for (long I = 0; I < nc; ++I)
    for (int dJ = -1; dJ <= 1; ++dJ) {
      const long J = I + dJ;
      if (J < 0 || J >= nc) continue;
      double acc = 0.0;
      for (int t = 0; t < 3; ++t) {
        const long a = 2 * I + t;
        if (a >= n) continue;
        for (int s = -1; s <= 1; ++s) {
          const long b = a + s;
          if (b < 0 || b >= n) continue;
          const long o = b - 2 * J;
          if (o < 0 || o > 2) continue;
          acc += rw[t] * fine[(long)(s + 1) * n + a] * pw[o];
        }
      }
      coarse[(long)(dJ + 1) * nc + I] = acc;
    }
}

void mgo_apply(const mgo_level* L, const double* v, double* out) {
#pragma omp parallel for schedule(static)
  for (long i = 0; i < L->nr; ++i)
    for (long j = 0; j < L->nc; ++j) {
      double d, o;
      row_parts(L, v, i, j, &d, &o);
      out[i * L->nc + j] = o + d * v[i * L->nc + j];
    }
}

/* MGCMTSolver.py:182-208 */
void mgo_wjacobi(const mgo_level* L, double* v, const double* f, double* tmp, int nu, double omega) {
  for (int it = 0; it < nu; ++it) {
#pragma omp parallel for schedule(static)
    for (long i = 0; i < L->nr; ++i)
      for (long j = 0; j < L->nc; ++j) {
        double d, o;
        row_parts(L, v, i, j, &d, &o);
        const long k = i * L->nc + j;
        tmp[k] = v[k] + omega * ((f[k] - (o + d * v[k])) / d);
      }
    memcpy(v, tmp, sizeof(double) * L->nr * L->nc);
  }
}

/* lower / upper parts of row k = (i,j) in index order */
static inline void row_lu(const mgo_level* L, const double* v, long i, long j, double* diag, double* low, double* up) {
  double d = 0.0, lo = 0.0, u = 0.0;
  for (int di = -1; di <= 1; ++di) {
    if (L->nr == 1 && di != 0) continue;
    for (int dj = -1; dj <= 1; ++dj) {
      const double c = coef(L, i, j, di, dj);
      if (di == 0 && dj == 0) d = c;
      else if (c != 0.0) {
        if (di < 0 || (di == 0 && dj < 0)) lo += c * at(L, v, i + di, j + dj);
        else u += c * at(L, v, i + di, j + dj);
      }
    }
  }
  *diag = d - L->shift;
  *low = lo;
  *up = u;
}

/* MGCMTSolver.py:210-227: forward Gauss-Seidel in index order */
void mgo_gseidel(const mgo_level* L, double* v, const double* f, int nu) {
  for (int it = 0; it < nu; ++it)
    for (long i = 0; i < L->nr; ++i)
      for (long j = 0; j < L->nc; ++j) {
        double d, lo, u;
        row_lu(L, v, i, j, &d, &lo, &u);
        v[i * L->nc + j] = (f[i * L->nc + j] - lo - u) / d;
      }
}

/* MGCMTSolver.py:229-246: v <- (D-wL)^-1((1-w)D + wU) v + w (D-L)^-1 f  (note (D-L), :241).
 * With L = -strict_lower(A), U = -strict_upper(A):  (D - wL) x = y  <=>  d x_k + w sum_{j<k} a_kj x_j = y_k. */
void mgo_sor(const mgo_level* L, double* v, const double* f, double* tmp, int nu, double omega) {
  const long n = L->nr * L->nc;
  double* gvec = (double*)calloc((size_t)n, sizeof(double));
  for (long i = 0; i < L->nr; ++i) /* (D - L) g = f */
    for (long j = 0; j < L->nc; ++j) {
      double d, lo, u;
      row_lu(L, gvec, i, j, &d, &lo, &u);
      gvec[i * L->nc + j] = (f[i * L->nc + j] - lo) / d;
    }
  for (int it = 0; it < nu; ++it) {
    /* y = ((1-w) D + w U) v = (1-w) d v - w sum_{j>k} a_kj v_j ; then forward solve into tmp */
    for (long i = 0; i < L->nr; ++i)
      for (long j = 0; j < L->nc; ++j) {
        double d, lo_new, u_old, dummy;
        row_lu(L, v, i, j, &d, &dummy, &u_old);
        row_lu(L, tmp, i, j, &d, &lo_new, &dummy); /* lower part sees the new values already in tmp */
        const long k = i * L->nc + j;
        tmp[k] = ((1.0 - omega) * d * v[k] - omega * u_old - omega * lo_new) / d;
      }
    for (long k = 0; k < n; ++k) v[k] = tmp[k] + omega * gvec[k];
  }
  free(gvec);
}

/* multicolour Gauss-Seidel / SOR: colours (i%2, j%2) in the order (0,1),(1,0),(0,0),(1,1) */
void mgo_multicolour(const mgo_level* L, double* v, const double* f, int nu, double omega) {
  static const int order[4][2] = {{0, 1}, {1, 0}, {0, 0}, {1, 1}};
  for (int it = 0; it < nu; ++it)
    for (int c = 0; c < 4; ++c) {
      const int ca = order[c][0], cb = order[c][1];
#pragma omp parallel for schedule(static)
      for (long i = ca; i < L->nr; i += 2)
        for (long j = cb; j < L->nc; j += 2) {
          double d, o;
          row_parts(L, v, i, j, &d, &o);
          const long k = i * L->nc + j;
          v[k] = v[k] + omega * ((f[k] - (o + d * v[k])) / d);
        }
    }
}

void mgo_residual(const mgo_level* L, const double* v, const double* f, double* r) {
#pragma omp parallel for schedule(static)
  for (long i = 0; i < L->nr; ++i)
    for (long j = 0; j < L->nc; ++j) {
      double d, o;
      row_parts(L, v, i, j, &d, &o);
      r[i * L->nc + j] = f[i * L->nc + j] - (o + d * v[i * L->nc + j]);
    }
}

/* MGCMTStencilMaker.py:57-78 for one level: (1/4,1/2,1/4) per coarsened direction */
void mgo_restrict(int dim, long nr, long nc, const double* fine, double* coarse) {
  const double w[3] = {0.25, 0.5, 0.25};
  const long cr = dim == 2 ? nr / 2 : 1, cc = nc / 2;
#pragma omp parallel for schedule(static)
  for (long I = 0; I < cr; ++I)
    for (long J = 0; J < cc; ++J) {
      double acc = 0.0;
      if (dim == 2) {
        for (int a = 0; a < 3; ++a)
          for (int b = 0; b < 3; ++b) {
            const long i = 2 * I + a, j = 2 * J + b;
            if (i < nr && j < nc) acc += w[a] * w[b] * fine[i * nc + j];
          }
      } else {
        for (int b = 0; b < 3; ++b)
          if (2 * J + b < nc) acc += w[b] * fine[2 * J + b];
      }
      coarse[I * cc + J] = acc;
    }
}

/* MGCMTStencilMaker.py:27-54 for one level; fine += P coarse when accumulate */
void mgo_prolong(int dim, long nr, long nc, const double* coarse, double* fine, int accumulate) {
  const long cc = nc / 2, cr = dim == 2 ? nr / 2 : 1;
#pragma omp parallel for schedule(static)
  for (long i = 0; i < nr; ++i)
    for (long j = 0; j < nc; ++j) {
      /* 1-D weights: odd index k takes c[(k-1)/2]; even k takes (c[k/2-1] + c[k/2])/2 */
      long Js[2];
      double wj[2];
      int nj = 0;
      if (j & 1) { Js[0] = j / 2; wj[0] = 1.0; nj = 1; }
      else { Js[0] = j / 2; wj[0] = 0.5; Js[1] = j / 2 - 1; wj[1] = 0.5; nj = 2; }
      long Is[2];
      double wi[2];
      int ni = 0;
      if (dim == 2) {
        if (i & 1) { Is[0] = i / 2; wi[0] = 1.0; ni = 1; }
        else { Is[0] = i / 2; wi[0] = 0.5; Is[1] = i / 2 - 1; wi[1] = 0.5; ni = 2; }
      } else { Is[0] = 0; wi[0] = 1.0; ni = 1; }
      double acc = 0.0;
      for (int a = 0; a < ni; ++a)
        for (int b = 0; b < nj; ++b)
          if (Is[a] >= 0 && Is[a] < cr && Js[b] >= 0 && Js[b] < cc) acc += wi[a] * wj[b] * coarse[Is[a] * cc + Js[b]];
      if (accumulate) fine[i * nc + j] += acc;
      else fine[i * nc + j] = acc;
    }
}

/* dense LU with partial pivoting for the coarsest level (spsolve, MGCMTSolver.py:305-308) */
int mgo_direct_solve(const mgo_level* L, const double* f, double* x) {
  const long n = L->nr * L->nc;
  double* a = (double*)calloc((size_t)n * n, sizeof(double));
  if (!a) return -1;
  for (long i = 0; i < L->nr; ++i)
    for (long j = 0; j < L->nc; ++j)
      for (int di = -1; di <= 1; ++di)
        for (int dj = -1; dj <= 1; ++dj) {
          const long ii = i + di, jj = j + dj;
          if (ii < 0 || ii >= L->nr || jj < 0 || jj >= L->nc) continue;
          if (L->nr == 1 && di != 0) continue;
          double c = coef(L, i, j, di, dj);
          if (di == 0 && dj == 0) c -= L->shift;
          a[(i * L->nc + j) * n + ii * L->nc + jj] = c;
        }
  for (long k = 0; k < n; ++k) x[k] = f[k];
  for (long k = 0; k < n; ++k) {
    long p = k;
    for (long r = k + 1; r < n; ++r)
      if (fabs(a[r * n + k]) > fabs(a[p * n + k])) p = r;
    if (p != k) {
      for (long c = 0; c < n; ++c) { double t = a[k * n + c]; a[k * n + c] = a[p * n + c]; a[p * n + c] = t; }
      double t = x[k]; x[k] = x[p]; x[p] = t;
    }
    for (long r = k + 1; r < n; ++r) {
      const double m = a[r * n + k] / a[k * n + k];
      if (m == 0.0) continue;
      for (long c = k; c < n; ++c) a[r * n + c] -= m * a[k * n + c];
      x[r] -= m * x[k];
    }
  }
  for (long k = n - 1; k >= 0; --k) {
    double s = x[k];
    for (long c = k + 1; c < n; ++c) s -= a[k * n + c] * x[c];
    x[k] = s / a[k * n + k];
  }
  free(a);
  return 0;
}

static void smooth(const mgo_level* L, int kind, double* v, const double* f, double* tmp, int nu, double omega) {
  switch (kind) {
    case 0: mgo_wjacobi(L, v, f, tmp, nu, omega); break;
    case 1: mgo_gseidel(L, v, f, nu); break;
    case 2: if (omega == 1.0) mgo_gseidel(L, v, f, nu); else { memset(tmp, 0, sizeof(double) * L->nr * L->nc); mgo_sor(L, v, f, tmp, nu, omega); } break;
    default: mgo_multicolour(L, v, f, nu, omega); break;
  }
}

/* MGCMTSolver.py:281-329.  X/Y are the fine-level factors; coarser factors are built on the fly. */
int mgo_vcycle(int dim, long g, long lowest, int nterms, const double* X, const double* Y, double shift, int kind, double omega,
               int nu1, int nu2, int nu_coarse, double* v, const double* f) {
  mgo_level L;
  L.dim = dim;
  L.nr = dim == 2 ? g : 1;
  L.nc = g;
  L.nterms = nterms;
  L.X = X;
  L.Y = Y;
  L.shift = shift;
  detect_five(&L);
  const long n = L.nr * L.nc;
  if (g == lowest) {
    double* x = (double*)malloc(sizeof(double) * n);
    int rc = mgo_direct_solve(&L, f, x);
    memcpy(v, x, sizeof(double) * n);
    free(x);
    return rc;
  }
  const long gc = g / 2, ncoarse = (dim == 2 ? gc : 1) * gc;
  double* tmp = (double*)malloc(sizeof(double) * n);
  double* r = (double*)malloc(sizeof(double) * n);
  double* rc_ = (double*)malloc(sizeof(double) * ncoarse);
  double* e = (double*)calloc((size_t)ncoarse, sizeof(double));
  double* Xc = dim == 2 ? (double*)malloc(sizeof(double) * nterms * 3 * gc) : NULL;
  double* Yc = (double*)malloc(sizeof(double) * nterms * 3 * gc);
  for (int m = 0; m < nterms; ++m) {
    mgo_galerkin(Y + (long)m * 3 * g, g, Yc + (long)m * 3 * gc);
    if (dim == 2) mgo_galerkin(X + (long)m * 3 * g, g, Xc + (long)m * 3 * gc);
  }
  smooth(&L, kind, v, f, tmp, nu1, omega);
  mgo_residual(&L, v, f, r);
  mgo_restrict(dim, L.nr, L.nc, r, rc_);
  int rc = mgo_vcycle(dim, gc, lowest, nterms, Xc, Yc, shift, kind, omega, nu_coarse, nu_coarse, nu_coarse, e, rc_);
  mgo_prolong(dim, L.nr, L.nc, e, v, 1);
  smooth(&L, kind, v, f, tmp, nu2, omega);
  free(tmp); free(r); free(rc_); free(e); free(Xc); free(Yc);
  return rc;
}

/* helpers for the Python front end: single-level calls on explicit factors */
#define LEVEL(L) mgo_level L; L.dim = dim; L.nr = nr; L.nc = nc; L.nterms = nterms; L.X = X; L.Y = Y; L.shift = shift; detect_five(&L)

void mgo_apply_level(int dim, long nr, long nc, int nterms, const double* X, const double* Y, double shift, const double* v, double* out) {
  LEVEL(L);
  mgo_apply(&L, v, out);
}

void mgo_smooth_level(int dim, long nr, long nc, int nterms, const double* X, const double* Y, double shift, int kind, double omega, int nu,
                      double* v, const double* f) {
  LEVEL(L);
  double* tmp = (double*)calloc((size_t)(nr * nc), sizeof(double));
  smooth(&L, kind, v, f, tmp, nu, omega);
  free(tmp);
}

void mgo_residual_level(int dim, long nr, long nc, int nterms, const double* X, const double* Y, double shift, const double* v,
                        const double* f, double* r) {
  LEVEL(L);
  mgo_residual(&L, v, f, r);
}
