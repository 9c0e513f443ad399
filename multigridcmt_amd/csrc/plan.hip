// Plan (level hierarchy, Galerkin factors, vector storage) and the C-ABI of libmgcmt_hip.so.
// Host code only; every kernel it enqueues is in kernels_*.hip.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "plan_internal.h"

using namespace mgcmt;

namespace {
thread_local std::string g_last_error;
}

namespace mgcmt {
int fail(int code, const std::string& msg) {
  g_last_error = msg;
  return code;
}
}  // namespace mgcmt

namespace {

bool is_pow2(int64_t x) { return x > 0 && (x & (x - 1)) == 0; }

// Galerkin product of one factor: R1 * T * P1 with R1 = full weighting (1/4,1/2,1/4 on fine
// 2I..2I+2) and P1 = 2 R1^T (MGCMTStencilMaker.py:27-78, MGCMTSolver.py:318).  Stays tridiagonal.
Tri galerkin(const Tri& f) {
  static const double rw[3] = {0.25, 0.5, 0.25};
  static const double pw[3] = {0.5, 1.0, 0.5};
  Tri c;
  c.n = f.n / 2;
  c.a.assign(3 * c.n, 0.0);
  for (int64_t I = 0; I < c.n; ++I) {
    for (int dJ = -1; dJ <= 1; ++dJ) {
      const int64_t J = I + dJ;
      if (J < 0 || J >= c.n) continue;
      double acc = 0.0;
      for (int t = 0; t < 3; ++t) {
        const int64_t a = 2 * I + t;
        if (a >= f.n) continue;
        for (int s = -1; s <= 1; ++s) {
          const int64_t b = a + s;
          if (b < 0 || b >= f.n) continue;
          const int64_t o = b - 2 * J;
          if (o < 0 || o > 2) continue;
          acc += rw[t] * f.at(a, b) * pw[o];
        }
      }
      c.a[(dJ + 1) * c.n + I] = acc;
    }
  }
  return c;
}

Tri identity_tri(int64_t n) {
  Tri t;
  t.n = n;
  t.a.assign(3 * n, 0.0);
  for (int64_t i = 0; i < n; ++i) t.a[n + i] = 1.0;
  return t;
}

int upload_op(const HostOp& h, const Level& L, int dim, DevOp* d) {
  KOp& k = d->k;
  k = KOp{};
  k.nterms = h.nterms;
  k.ldx = L.nr + 2 * L.halo;
  k.ldy = L.gc;
  for (int m = 0; m < h.nterms; ++m) {
    std::vector<double> xs(3 * k.ldx, 0.0);
    for (int part = 0; part < 3; ++part)
      for (int64_t i = -L.halo; i < L.nr + L.halo; ++i) {
        const int64_t gi = L.r0 + i;
        if (gi >= 0 && gi < L.gr) xs[part * k.ldx + (i + L.halo)] = h.X[m].a[part * L.gr + gi];
      }
    double *dx = nullptr, *dy = nullptr;
    MG_HIP(hipMalloc((void**)&dx, xs.size() * sizeof(double)));
    d->owned.push_back(dx);
    MG_HIP(hipMemcpy(dx, xs.data(), xs.size() * sizeof(double), hipMemcpyHostToDevice));
    MG_HIP(hipMalloc((void**)&dy, 3 * L.gc * sizeof(double)));
    d->owned.push_back(dy);
    MG_HIP(hipMemcpy(dy, h.Y[m].a.data(), 3 * L.gc * sizeof(double), hipMemcpyHostToDevice));
    k.X[m] = dx + L.halo;
    k.Y[m] = dy;
  }
  if (dim == 1 && h.nterms > 0) {
    // 1-D: X_m is 1 x 1, so the operator is ONE tridiagonal, sum_m x_m Y_m — folded here for the fused 1-D passes
    const int64_t n = L.gc;
    std::vector<double> t(3 * n, 0.0);
    for (int m = 0; m < h.nterms; ++m) {
      const double x = h.X[m].di(0);
      for (int64_t i = 0; i < 3 * n; ++i) t[i] += x * h.Y[m].a[i];
    }
    double* dt = nullptr;
    MG_HIP(hipMalloc((void**)&dt, t.size() * sizeof(double)));
    d->owned.push_back(dt);
    MG_HIP(hipMemcpy(dt, t.data(), t.size() * sizeof(double), hipMemcpyHostToDevice));
    k.one_d = 1;
    k.tri = dt;
    bool constant = n >= 3;
    for (int64_t i = 0; constant && i < n; ++i) {
      if (i > 0 && t[i] != t[1]) constant = false;                          // lower (entry 0 is outside the matrix)
      if (i + 1 < n && (t[n + i] != t[n] || t[2 * n + i] != t[2 * n])) constant = false;  // diagonal but the last, upper (the last entry is outside)
    }
    if (constant) {
      k.tri_const = 1;
      k.t_lo = t[1];
      k.t_di = t[n];
      k.t_up = t[2 * n];
      k.t_last = t[2 * n - 1];
    }
  }
  // constant-coefficient 5-point (2-D) / 3-point (1-D) detection: every factor Toeplitz and the
  // corner coefficients zero -> the kernels take three scalars instead of the factor arrays
  auto toeplitz = [](const Tri& t, double* lo, double* di, double* up) {
    *di = t.di(0);
    *lo = t.n > 1 ? t.lo(1) : 0.0;
    *up = t.n > 1 ? t.up(0) : 0.0;
    for (int64_t i = 0; i < t.n; ++i) {
      if (t.di(i) != *di) return false;
      if (i > 0 && t.lo(i) != *lo) return false;
      if (i + 1 < t.n && t.up(i) != *up) return false;
    }
    return true;
  };
  double c[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  bool all = h.nterms > 0;
  for (int m = 0; m < h.nterms && all; ++m) {
    double x[3], y[3];
    if (!toeplitz(h.X[m], &x[0], &x[1], &x[2]) || !toeplitz(h.Y[m], &y[0], &y[1], &y[2])) {
      all = false;
      break;
    }
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) c[a][b] += x[a] * y[b];
  }
  if (all && c[0][0] == 0 && c[0][2] == 0 && c[2][0] == 0 && c[2][2] == 0 && c[0][1] == c[2][1] && c[1][0] == c[1][2] &&
      (dim == 2 || c[0][1] == 0)) {
    k.five_point = 1;
    k.c0 = c[1][1];
    k.cn = c[0][1];
    k.cw = c[1][0];
  }
  // constant 5-point part plus ONE product potential on the diagonal: the Toeplitz terms form a 5-point operator,
  // the remaining term has diagonal factors only
  if (!k.five_point && dim == 2 && h.nterms >= 2) {
    auto diagonal_only = [](const Tri& t) {
      for (int64_t i = 0; i < t.n; ++i)
        if ((i > 0 && t.lo(i) != 0.0) || (i + 1 < t.n && t.up(i) != 0.0)) return false;
      return true;
    };
    double c5[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    int nd = 0, dm[2] = {0, 0};
    bool ok = true;
    for (int m = 0; m < h.nterms && ok; ++m) {
      double x[3], y[3];
      if (toeplitz(h.X[m], &x[0], &x[1], &x[2]) && toeplitz(h.Y[m], &y[0], &y[1], &y[2])) {
        for (int a = 0; a < 3; ++a)
          for (int b = 0; b < 3; ++b) c5[a][b] += x[a] * y[b];
      } else if (diagonal_only(h.X[m]) && diagonal_only(h.Y[m]) && nd < 1) {
        dm[nd++] = m;
      } else {
        ok = false;
      }
    }
    if (ok && nd == 1 && c5[0][0] == 0 && c5[0][2] == 0 && c5[2][0] == 0 && c5[2][2] == 0 && c5[0][1] == c5[2][1] && c5[1][0] == c5[1][2]) {
      k.five_diag = 1;
      k.ndiag = nd;
      k.c0 = c5[1][1];
      k.cn = c5[0][1];
      k.cw = c5[1][0];
      for (int t = 0; t < nd; ++t) {
        k.dX[t] = k.X[dm[t]] + k.ldx;  // the diagonal row of the factor arrays ([lower | diag | upper])
        k.dY[t] = k.Y[dm[t]] + k.ldy;
      }
    }
  }
  // Galerkin levels of a constant operator: Toeplitz factors whose last diagonal entry differs
  if (!k.five_point && !k.five_diag && dim == 2 && h.nterms > 0) {
    auto toeplitz_but_last = [](const Tri& t, double* lo, double* di, double* up, double* last) {
      if (t.n < 3) return false;
      *di = t.di(0);
      *lo = t.lo(1);
      *up = t.up(0);
      *last = t.di(t.n - 1);
      for (int64_t i = 0; i < t.n; ++i) {
        if (i + 1 < t.n && t.di(i) != *di) return false;
        if (i > 0 && t.lo(i) != *lo) return false;
        if (i + 1 < t.n && t.up(i) != *up) return false;
      }
      return true;
    };
    bool ok = true;
    double c9[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, crow[3] = {0, 0, 0}, ccol[3] = {0, 0, 0}, corner = 0;
    int nconst = 0, nvar = 0, var_term = -1;
    for (int m = 0; m < h.nterms && ok; ++m) {
      double x[3], y[3], xl, yl;
      if (!(toeplitz_but_last(h.X[m], &x[0], &x[1], &x[2], &xl) && toeplitz_but_last(h.Y[m], &y[0], &y[1], &y[2], &yl))) {
        // a term with variable factors: one of them may ride on top of the constant part (nine_var)
        ++nvar;
        var_term = m;
        ok = nvar <= 1;
        continue;
      }
      ++nconst;
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) c9[a][b] += x[a] * y[b];
      for (int b = 0; b < 3; ++b) crow[b] += xl * y[b];   // last row: X's diagonal entry is the modified one
      for (int a = 0; a < 3; ++a) ccol[a] += x[a] * yl;   // last column: Y's diagonal entry is the modified one
      corner += xl * yl;
      crow[1] += 0.0;
    }
    if (ok && nconst >= 1) {
      // on the last row the centre coefficient of the last column is the corner; crow[1] is the centre elsewhere
      if (nvar == 0) {
        k.nine_const = 1;
      } else {
        k.nine_var = 1;
        k.vX = k.X[var_term];
        k.vY = k.Y[var_term];
      }
      for (int a = 0; a < 3; ++a) {
        for (int b = 0; b < 3; ++b) k.c9[a][b] = c9[a][b];
        k.c9row[a] = crow[a];
        k.c9col[a] = ccol[a];
      }
      k.c9corner = corner;
    }
  }
  return MGCMT_OK;
}

}  // namespace
namespace mgcmt {
int ensure_slot(mgcmt_plan* p, int l, int slot) {
  Level& L = p->levels[l];
  if (L.base[slot]) return MGCMT_OK;
  const size_t bytes = (size_t)L.stride * p->nvec * sizeof(double);
  hipError_t e = hipMalloc((void**)&L.base[slot], bytes);
  if (e != hipSuccess) return fail(MGCMT_ERR_NOMEM, std::string("hipMalloc of a level vector failed: ") + hipGetErrorString(e));
  MG_HIP(hipMemset(L.base[slot], 0, bytes));
  return MGCMT_OK;
}

}  // namespace mgcmt
namespace {

int check_level(const mgcmt_plan* p, int l) {
  if (!p) return fail(MGCMT_ERR_INVALID, "null plan");
  if (l < 0 || l >= (int)p->levels.size()) return fail(MGCMT_ERR_INVALID, "level out of range");
  return MGCMT_OK;
}

int check_vec(const mgcmt_plan* p, int l, int slot, int vec) {
  MG_TRY(check_level(p, l));
  if (slot < 0 || slot > 3) return fail(MGCMT_ERR_INVALID, "slot out of range");
  if (vec < 0 || vec >= p->nvec) return fail(MGCMT_ERR_INVALID, "vector index out of range");
  return MGCMT_OK;
}

int check_k(const mgcmt_plan* p, int k) {
  if (k < 1 || k > p->nvec) return fail(MGCMT_ERR_INVALID, "k must be in 1..nvec");
  return MGCMT_OK;
}

hipStream_t S(void* s) { return (hipStream_t)s; }

}  // namespace
namespace mgcmt {
int post_launch() {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(MGCMT_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
  return MGCMT_OK;
}

}  // namespace mgcmt
namespace {

// ---- smoothers --------------------------------------------------------------------------------

}  // namespace
namespace mgcmt {
// one fused pass V -> T (then swapped) on a level the fused kernels cover
int fused_pass(mgcmt_plan* p, int l, int kind, int nsweep, double omega, int mode, int k, hipStream_t s, int npre, long out_lo,
               long out_hi, bool swap, long out_lo2, long out_hi2) {
  Level& L = p->levels[l];
  KVec coarse{nullptr, 0};
  long cnc = 0;
  if ((mode & 3) != 0) {
    coarse = p->kvec(l + 1, (mode & 3) == 1 ? MGCMT_SLOT_V : MGCMT_SLOT_F);
    cnc = p->levels[l + 1].gc;
  }
  // rows beyond a strip that hold the neighbours' data (the passes read no further: exchanged_rows)
  const long hx = exchanged_rows(p, l);
  const long row_lo = L.r0 == 0 ? 0 : -hx;
  const long row_hi = L.r0 + L.nr == L.gr ? L.nr : L.nr + hx;
  launch_fused(s, p->kgrid(l), L.dA.k, p->kvec(l, MGCMT_SLOT_V), p->kvec(l, MGCMT_SLOT_F), p->kvec(l, MGCMT_SLOT_T), coarse, cnc,
               p->d_shifts, omega, kind == MGCMT_GS_MC ? 1 : 0, nsweep, mode, npre, row_lo, row_hi, L.gr - 1 - L.r0, k, p->fused_rows,
               out_lo, out_hi, out_lo2, out_hi2);
  if (swap && !(mode & 8)) std::swap(L.base[MGCMT_SLOT_V], L.base[MGCMT_SLOT_T]);  // a no-store pass leaves V as it was
  return MGCMT_OK;
}

bool fused_level(const mgcmt_plan* p, int l, int kind) {
  return p->use_fused && (kind == MGCMT_WJACOBI || kind == MGCMT_GS_MC) && fused_supported(p->kgrid(l), p->levels[l].dA.k);
}

int pass_sweeps(const mgcmt_plan* p, int l, int kind, int left) {
  const int cap = fused_max_sweeps(p->levels[l].dA.k, kind == MGCMT_GS_MC ? 1 : 0);
  return left < cap ? left : cap;
}

int exchanged_rows(const mgcmt_plan* p, int l) {
  const Level& L = p->levels[l];
  const KOp& k = L.dA.k;
  const int want = (k.five_point || k.five_diag) ? 8 : 10;
  return want < L.halo ? want : L.halo;
}
}  // namespace mgcmt
namespace {

// nsweeps generalised lexicographic sweeps on vector slot `slot` (right-hand side: slot `fslot`), each followed by
// slot += gamma * fslot: the wave pipeline where it covers the level (sweeps chained in one launch, the update inside
// the sweep), the one-workgroup kernel otherwise
int lex_sweep(mgcmt_plan* p, int l, int slot, double alpha, double beta, double wU, double wL, int k, hipStream_t s, int nsweeps = 1,
              double gamma = 0.0, int fslot = MGCMT_SLOT_F) {
  auto update = [&]() {
    if (gamma != 0.0)
      for (int q = 0; q < k; ++q) launch_axpy(s, p->interior(l), gamma, p->kvec(l, fslot, q).p, p->kvec(l, slot, q).p);
  };
  const KGrid g = p->kgrid(l);
  const KOp& op = p->levels[l].dA.k;
  if (p->use_lex_wave && p->levels[l].nr == p->levels[l].gr && lex_wave_supported(g, op)) {
    const bool band = p->use_lex_wave == 2;
    const size_t blocks = (size_t)lex_wave_blocks(g);
    const size_t need_scan = (size_t)lex_wave_carry(g, p->nvec, nsweeps), need_band = (size_t)p->nvec * lex_band_count(g) * lex_band_stride(g);
    const size_t need_carry = band ? need_band : need_scan, need_sync = 2 + 4 * (size_t)p->nvec * blocks;  // (2 words used; the rest is the diagnostic build's per-block record)
    if (need_carry > p->lex_carry_doubles || need_sync > p->lex_sync_words) {
      // cached cycle graphs hold the old scratch pointers in their memset / kernel nodes: they go before the buffers do
      // (a graph replayed after this point would write through freed memory)
      p->graphs_invalidate();
      MG_HIP(hipStreamSynchronize(s));
      if (p->lex_carry) (void)hipFree(p->lex_carry);
      if (p->lex_sync) (void)hipFree(p->lex_sync);
      p->lex_carry = nullptr;
      p->lex_sync = nullptr;
      p->lex_carry_doubles = p->lex_sync_words = 0;
      if (hipMalloc((void**)&p->lex_carry, need_carry * sizeof(double)) != hipSuccess ||
          hipMalloc((void**)&p->lex_sync, need_sync * sizeof(unsigned)) != hipSuccess)
        return fail(MGCMT_ERR_NOMEM, "scratch of the lexicographic wave pipeline");
      MG_HIP(hipMemset(p->lex_sync, 0, need_sync * sizeof(unsigned)));
      p->lex_carry_doubles = need_carry;
      p->lex_sync_words = need_sync;
    }
    if (band) {
      for (int it = 0; it < nsweeps; ++it) {
        launch_lex_band(s, g, op, p->kvec(l, slot), p->kvec(l, fslot), p->d_shifts, alpha, beta, wU, wL, k, p->lex_carry, p->lex_sync);
        update();
      }
    } else if (p->lex_chain) {
      launch_lex_wave(s, g, op, p->kvec(l, slot), p->kvec(l, fslot), p->d_shifts, alpha, beta, wU, wL, k, p->lex_carry, p->lex_sync, nsweeps, gamma);
    } else {
      for (int it = 0; it < nsweeps; ++it)
        launch_lex_wave(s, g, op, p->kvec(l, slot), p->kvec(l, fslot), p->d_shifts, alpha, beta, wU, wL, k, p->lex_carry, p->lex_sync, 1, gamma);
    }
    p->lex_wave_used = true;
    return MGCMT_OK;
  }
  for (int it = 0; it < nsweeps; ++it) {
    launch_lex_sweep(s, g, op, p->kvec(l, slot), p->kvec(l, fslot), p->d_shifts, alpha, beta, wU, wL, k);
    update();
  }
  return MGCMT_OK;
}

// a synchronising call looks at the error word of the wave pipeline (a block that gave up waiting)
int lex_wave_check(mgcmt_plan* p) {
  if (!p->lex_wave_used || !p->lex_sync) return MGCMT_OK;
  p->lex_wave_used = false;
  unsigned err = 0;
  MG_HIP(hipMemcpy(&err, p->lex_sync + 1, sizeof(unsigned), hipMemcpyDeviceToHost));
  if (err != 0) {
    MG_HIP(hipMemset(p->lex_sync + 1, 0, sizeof(unsigned)));  // reported: the next sweeps start clean
    return fail(MGCMT_ERR_HIP, "lexicographic wave pipeline: a block timed out waiting for its neighbour");
  }
  return MGCMT_OK;
}

int smooth_impl(mgcmt_plan* p, int l, int kind, int nu, double omega, int k, hipStream_t s) {
  Level& L = p->levels[l];
  MG_TRY(ensure_slot(p, l, MGCMT_SLOT_V));
  MG_TRY(ensure_slot(p, l, MGCMT_SLOT_F));
  const KGrid g = p->kgrid(l);
  const KOp& op = L.dA.k;
  if (nu <= 0) return MGCMT_OK;  // (a V(0,nu2) cycle: nothing to launch — the chained lexicographic sweeps size their scratch by nu)
  if (fused_level(p, l, kind)) {
    MG_TRY(ensure_slot(p, l, MGCMT_SLOT_T));
    for (int left = nu; left > 0;) {
      const int n = pass_sweeps(p, l, kind, left);
      MG_TRY(fused_pass(p, l, kind, n, omega, 0, k, s));
      left -= n;
    }
    return post_launch();
  }
  switch (kind) {
    case MGCMT_WJACOBI: {
      MG_TRY(ensure_slot(p, l, MGCMT_SLOT_T));
      for (int it = 0; it < nu; ++it) {
        launch_wjacobi(s, g, op, p->kvec(l, MGCMT_SLOT_V), p->kvec(l, MGCMT_SLOT_F), p->kvec(l, MGCMT_SLOT_T), p->d_shifts, omega, k);
        std::swap(L.base[MGCMT_SLOT_V], L.base[MGCMT_SLOT_T]);
      }
      break;
    }
    case MGCMT_GS_MC: {
      static const int order[4][2] = {{0, 1}, {1, 0}, {0, 0}, {1, 1}};
      for (int it = 0; it < nu; ++it)
        for (int c = 0; c < 4; ++c)
          launch_mc_colour(s, g, op, p->kvec(l, MGCMT_SLOT_V), p->kvec(l, MGCMT_SLOT_F), p->d_shifts, omega, order[c][0], order[c][1], k);
      break;
    }
    case MGCMT_GS_LEX:
    case MGCMT_SOR_LEX: {
      if (kind == MGCMT_GS_LEX || omega == 1.0) {
        MG_TRY(lex_sweep(p, l, MGCMT_SLOT_V, 0.0, 1.0, 1.0, 1.0, k, s, nu));  // (nu sweeps, chained in one launch where the wave pipeline covers the level)
      } else {
        // reference SOR (MGCMTSolver.py:229-246): v <- (D-wL)^-1((1-w)D + wU) v + w (D-L)^-1 f.
        // T <- (D-L)^-1 f once, then per sweep the homogeneous recurrence followed by v += w T.
        MG_TRY(ensure_slot(p, l, MGCMT_SLOT_T));
        for (int q = 0; q < k; ++q) launch_fill(s, p->kvec(l, MGCMT_SLOT_T, q).p, p->interior(l), 0.0);
        MG_TRY(lex_sweep(p, l, MGCMT_SLOT_T, 0.0, 1.0, 0.0, 1.0, k, s));
        // (beta = 0: the sweeps do not read their right-hand side — T rides in its place and is added as it is stored)
        MG_TRY(lex_sweep(p, l, MGCMT_SLOT_V, 1.0 - omega, 0.0, omega, omega, k, s, nu, omega, MGCMT_SLOT_T));
      }
      break;
    }
    default:
      return fail(MGCMT_ERR_INVALID, "unknown smoother kind");
  }
  return post_launch();
}

int residual_restrict_impl(mgcmt_plan* p, int l, int k, hipStream_t s) {
  if (l + 1 >= (int)p->levels.size()) return fail(MGCMT_ERR_INVALID, "no coarser level");
  MG_TRY(ensure_slot(p, l, MGCMT_SLOT_V));
  MG_TRY(ensure_slot(p, l, MGCMT_SLOT_F));
  MG_TRY(ensure_slot(p, l, MGCMT_SLOT_T));
  MG_TRY(ensure_slot(p, l + 1, MGCMT_SLOT_V));
  MG_TRY(ensure_slot(p, l + 1, MGCMT_SLOT_F));
  launch_residual(s, p->kgrid(l), p->levels[l].dA.k, p->kvec(l, MGCMT_SLOT_V), p->kvec(l, MGCMT_SLOT_F), p->kvec(l, MGCMT_SLOT_T), p->d_shifts, k);
  launch_restrict(s, p->kgrid(l), p->kgrid(l + 1), p->kvec(l, MGCMT_SLOT_T), p->kvec(l + 1, MGCMT_SLOT_F), k);
  for (int q = 0; q < k; ++q) launch_fill(s, p->kvec(l + 1, MGCMT_SLOT_V, q).p, p->interior(l + 1), 0.0);
  return post_launch();
}

int prolong_correct_impl(mgcmt_plan* p, int l, int k, hipStream_t s) {
  if (l + 1 >= (int)p->levels.size()) return fail(MGCMT_ERR_INVALID, "no coarser level");
  MG_TRY(ensure_slot(p, l, MGCMT_SLOT_V));
  MG_TRY(ensure_slot(p, l + 1, MGCMT_SLOT_V));
  launch_prolong(s, p->kgrid(l), p->kgrid(l + 1), p->kvec(l + 1, MGCMT_SLOT_V), p->kvec(l, MGCMT_SLOT_V), 1, k);
  return post_launch();
}

// the factorisation (and, for at most 1024 unknowns, the explicit inverse) of the coarsest-level matrix for the current
// shifts: allocated on first use, redone when the shifts change
int ensure_coarse_ready(mgcmt_plan* p, int l, int k, hipStream_t s) {
  Level& L = p->levels[l];
  if (L.nr != L.gr) return fail(MGCMT_ERR_UNSUPPORTED, "direct solve on a row strip");
  MG_TRY(ensure_slot(p, l, MGCMT_SLOT_V));
  MG_TRY(ensure_slot(p, l, MGCMT_SLOT_F));
  BandState& B = L.band;
  const long n = (long)L.nr * L.gc;
  const int kl = L.nr == 1 ? 1 : (int)L.gc + 1;
  if (kl > 129) return fail(MGCMT_ERR_UNSUPPORTED, "lowest_level too large for the direct solve (2-D: at most 128)");
  if (!B.b.ab) {
    B.b.n = n;
    B.b.kl = kl;
    B.b.width = 3 * kl + 1;
    B.b.ab_stride = n * B.b.width;
    B.b.piv_stride = n;
    MG_HIP(hipMalloc((void**)&B.b.ab, sizeof(double) * B.b.ab_stride * p->nvec));
    MG_HIP(hipMalloc((void**)&B.b.piv, sizeof(int) * B.b.piv_stride * p->nvec));
    if (n <= 1024) MG_HIP(hipMalloc((void**)&B.inv, sizeof(double) * n * n * p->nvec));
  }
  bool same = B.valid && B.k >= k;
  if (same)
    for (int q = 0; q < k; ++q) same = same && B.shifts[q] == p->h_shifts[q];
  if (!same) {
    launch_band_assemble(s, p->kgrid(l), L.dA.k, p->d_shifts, B.b, k);
    launch_band_factor(s, B.b, k);
    if (B.inv) launch_band_invert(s, B.b, B.inv, n * n, k);
    B.valid = true;
    B.k = k;
    B.shifts.assign(p->h_shifts.begin(), p->h_shifts.begin() + k);
  }
  return post_launch();
}

int coarse_solve_impl(mgcmt_plan* p, int l, int k, hipStream_t s) {
  MG_TRY(ensure_coarse_ready(p, l, k, s));
  Level& L = p->levels[l];
  BandState& B = L.band;
  const long n = (long)L.nr * L.gc;
  if (B.inv) launch_dense_solve(s, n, B.inv, n * n, p->kvec(l, MGCMT_SLOT_F), p->kvec(l, MGCMT_SLOT_V), k);
  else launch_band_solve(s, B.b, p->kvec(l, MGCMT_SLOT_F), p->kvec(l, MGCMT_SLOT_V), k);
  return post_launch();
}

// pre-smoothing + residual + restriction (MGCMTSolver.py:313-316); one pass less on fused levels
// zero_in: V[l] is known to be zero (and has not been cleared); true below the level the cycle starts on
// recompute (may be null): out — how many of the pre-smoothing sweeps were NOT stored (the residual was restricted
// from them on the fly) and must be recomputed by up_leg from the untouched V; still_zero: V is still "zero, uncleared"
int down_leg(mgcmt_plan* p, int l, int kind, int nu, double omega, int k, bool zero_in, hipStream_t s, int* recompute = nullptr,
             bool* still_zero = nullptr, int nu_up = 0) {
  if (recompute) *recompute = 0;
  if (still_zero) *still_zero = false;
  MG_TRY(ensure_slot(p, l, MGCMT_SLOT_V));
  MG_TRY(ensure_slot(p, l, MGCMT_SLOT_F));
  MG_TRY(ensure_slot(p, l, MGCMT_SLOT_T));
  MG_TRY(ensure_slot(p, l + 1, MGCMT_SLOT_V));
  MG_TRY(ensure_slot(p, l + 1, MGCMT_SLOT_F));
  // a fused first pass takes "V is zero" as a flag; the one-launch-per-operation path needs V cleared
  if (zero_in && !(nu >= 1 && fused_level(p, l, kind)))
    for (int q = 0; q < k; ++q) launch_fill(s, p->kvec(l, MGCMT_SLOT_V, q).p, p->interior(l), 0.0);
  if (nu >= 1 && fused_level(p, l, kind)) {
    MG_TRY(ensure_slot(p, l, MGCMT_SLOT_V));
    MG_TRY(ensure_slot(p, l, MGCMT_SLOT_F));
    MG_TRY(ensure_slot(p, l, MGCMT_SLOT_T));
    MG_TRY(ensure_slot(p, l + 1, MGCMT_SLOT_V));
    MG_TRY(ensure_slot(p, l + 1, MGCMT_SLOT_F));
    int left = nu, zi = zero_in ? 4 : 0;
    while (left > pass_sweeps(p, l, kind, left)) {
      const int n = pass_sweeps(p, l, kind, left);
      MG_TRY(fused_pass(p, l, kind, n, omega, zi, k, s));
      zi = 0;
      left -= n;
    }
    const int rmax = fused_max_recompute(p->levels[l].dA.k, kind == MGCMT_GS_MC ? 1 : 0, pass_sweeps(p, l, kind, nu_up));
    // worth it where the level is bandwidth-bound; on small levels the longer pipeline of the up-leg pass costs more
    // latency than the saved traffic is worth (measured: 1024^2 cycle 0.148 -> 0.179 ms with it)
    // (the same threshold serves the 9-point Galerkin levels: measured at 16384^2, recompute off on them costs 0.26 ms
    // per cycle, thresholds of 2^20 and 2^18 points are within noise of / slower than 2^22)
    const bool big = p->force_recompute || p->interior(l) >= (1L << 22);
    if (recompute && p->use_recompute && big && left <= rmax) {
      MG_TRY(fused_pass(p, l, kind, left, omega, 2 | 8 | zi, k, s));
      *recompute = left;
      if (still_zero) *still_zero = zi != 0;
    } else {
      MG_TRY(fused_pass(p, l, kind, left, omega, 2 | zi, k, s));
    }
    return post_launch();
  }
  MG_TRY(smooth_impl(p, l, kind, nu, omega, k, s));
  launch_residual(s, p->kgrid(l), p->levels[l].dA.k, p->kvec(l, MGCMT_SLOT_V), p->kvec(l, MGCMT_SLOT_F), p->kvec(l, MGCMT_SLOT_T), p->d_shifts, k);
  launch_restrict(s, p->kgrid(l), p->kgrid(l + 1), p->kvec(l, MGCMT_SLOT_T), p->kvec(l + 1, MGCMT_SLOT_F), k);
  return post_launch();
}

// prolongation + correction + post-smoothing (MGCMTSolver.py:323-326)
int up_leg(mgcmt_plan* p, int l, int kind, int nu, double omega, int k, hipStream_t s, int recompute = 0, bool still_zero = false) {
  if (nu >= 1 && fused_level(p, l, kind)) {
    MG_TRY(ensure_slot(p, l, MGCMT_SLOT_V));
    MG_TRY(ensure_slot(p, l, MGCMT_SLOT_F));
    MG_TRY(ensure_slot(p, l, MGCMT_SLOT_T));
    MG_TRY(ensure_slot(p, l + 1, MGCMT_SLOT_V));
    int left = nu;
    const int first = pass_sweeps(p, l, kind, left);
    MG_TRY(fused_pass(p, l, kind, first, omega, 1 | (still_zero ? 4 : 0), k, s, recompute));
    left -= first;
    while (left > 0) {
      const int n = pass_sweeps(p, l, kind, left);
      MG_TRY(fused_pass(p, l, kind, n, omega, 0, k, s));
      left -= n;
    }
    return post_launch();
  }
  MG_TRY(prolong_correct_impl(p, l, k, s));
  return smooth_impl(p, l, kind, nu, omega, k, s);
}

// ---- Gram-Schmidt -------------------------------------------------------------------------------

int gramschmidt_impl(mgcmt_plan* p, int l, int slot, int k, int modified, hipStream_t s) {
  MG_TRY(ensure_slot(p, l, slot));
  const long n = p->interior(l);
  const long stride = p->levels[l].stride;
  double* a0 = p->kvec(l, slot, 0).p;
  double* sc = p->d_scalars;
  if (modified) {
    // MGCMTProcessor.py:44-50: q_i = a_i/|a_i|; a_j -= (<a_j,q_i>/<q_i,q_i>) q_i for j > i.
    // One launch per column (k_mgs_step): it projects column i out of all later ones, normalises it and leaves the
    // inner products the next column needs; the first set comes from one batched dot launch.
    if (mgs_small_fits(n)) {
      launch_mgs_small(s, n, a0, stride, k);  // short columns: everything in one workgroup
      return post_launch();
    }
    // Long columns: first the blocked form — Gram matrix, its factor, Q = A R^-1: 3 k vector streams instead of k^2 + k —
    // which leaves a gate word up where its rounding errors (cond^2 eps) would show; the column-by-column launches
    // behind it return at once when the gate is down (MGCMT_OPT_MGS_BLOCK = 0: column by column only)
    const double* gate = nullptr;
    // (on every level the one-workgroup kernel does not take — measured with the blocked form only from 2^20 points on: a
    // 1024^2 cycle of 10 columns 1.32 ms against 1.10, a 4096^2 cycle 6.21 against 6.26: the gated launches cost less than
    // the column steps of the middle levels)
    if (p->use_mgs_block && k >= 2 && k <= mgs_block_max() && n >= p->mgs_block_min) {
      launch_mgs_blocked(s, n, a0, stride, k, p->d_partials, p->d_mgs);
      gate = p->d_mgs + mgs_block_gate_word();
    }
    double* pa = p->d_partials;
    double* pb = p->d_partials + (long)(kMaxVec + 1) * 1024;
    launch_dot_partials(s, n, a0, a0, stride, k, pa, gate);  // <a_0, a_t>, t = 0..k-1
    for (int i = 0; i < k; ++i) {
      launch_mgs_step(s, n, pa, a0 + i * stride, stride, k - 1 - i, pb, 0, gate);
      std::swap(pa, pb);
    }
    (void)sc;
  } else {
    // MGCMTProcessor.py:34-42: u_j = a_j - sum_{i<j} (<a_j,u_i>/<u_i,u_i>) u_i with the ORIGINAL a_j in every
    // inner product, then all columns normalised
    for (int j = 1; j < k; ++j) {
      double* aj = a0 + j * stride;
      launch_dots(s, n, aj, a0, stride, j, p->d_partials, sc);                  // <a_j, u_i>, i < j
      for (int i = 0; i < j; ++i) launch_dots(s, n, a0 + i * stride, a0 + i * stride, 0, 1, p->d_partials + kMaxVec * 1024, sc + kMaxVec + i);
      for (int i = 0; i < j; ++i) launch_axpy_dev(s, n, sc + i, sc + kMaxVec + i, -1.0, a0 + i * stride, aj);
    }
    for (int i = 0; i < k; ++i) {
      double* ai = a0 + i * stride;
      launch_dots(s, n, ai, ai, 0, 1, p->d_partials, sc);
      launch_scale_dev(s, n, sc, 1, ai);
    }
  }
  return post_launch();
}

// (re)factor the coarsest-level matrix when the shifts changed; no-op otherwise
int ensure_coarse_factor(mgcmt_plan* p, int l, int k, hipStream_t s) {
  Level& L = p->levels[l];
  BandState& B = L.band;
  bool same = B.b.ab && B.valid && B.k >= k;
  if (same)
    for (int q = 0; q < k; ++q) same = same && B.shifts[q] == p->h_shifts[q];
  if (same) return MGCMT_OK;
  if (!B.b.ab) return MGCMT_OK;  // first use: coarse_solve_impl allocates and factors
  launch_band_assemble(s, p->kgrid(l), L.dA.k, p->d_shifts, B.b, k);
  launch_band_factor(s, B.b, k);
  if (B.inv) launch_band_invert(s, B.b, B.inv, (long)B.b.n * B.b.n, k);
  B.valid = true;
  B.k = k;
  B.shifts.assign(p->h_shifts.begin(), p->h_shifts.begin() + k);
  return post_launch();
}

// First level of the cycle's tail: the levels of at most 32 x 32 points below the level the cycle starts on run as
// ONE launch (kernels_tail.hip).  -1: no tail (1-D, strips, lexicographic smoothers, Gram-Schmidt between the levels,
// a coarsest grid too large for the explicit inverse, or nothing to gain).
int tail_level(const mgcmt_plan* p, int level, int kind, int nu_coarse, int gram_schmidt) {
  const int last = (int)p->levels.size() - 1;
  if (!p->use_tail || !p->use_fused || p->dim != 2 || gram_schmidt || nu_coarse < 1) return -1;
  if (kind != MGCMT_WJACOBI && kind != MGCMT_GS_MC) return -1;
  const Level& C = p->levels[last];
  if (C.nr != C.gr || (long)C.nr * C.gc > 1024) return -1;
  for (int l = level + 1; l < last; ++l) {
    const Level& L = p->levels[l];
    if (L.nr != L.gr || L.gr != L.gc) continue;
    if (tail_fits(L.gr, last - l + 1, L.dA.k.nterms)) return l;
  }
  return -1;
}

TailArgs tail_args(mgcmt_plan* p, int lt, int kind, int nu, double omega) {
  const int last = (int)p->levels.size() - 1;
  TailArgs a{};
  a.g0 = (int)p->levels[lt].gr;
  a.nlev = last - lt + 1;
  a.nterms = p->levels[lt].dA.k.nterms;
  for (int l = lt; l <= last; ++l) {
    const KOp& op = p->levels[l].dA.k;
    for (int m = 0; m < op.nterms; ++m) {
      a.X[l - lt][m] = op.X[m];
      a.Y[l - lt][m] = op.Y[m];
    }
    a.ldx[l - lt] = op.ldx;
    a.ldy[l - lt] = op.ldy;
  }
  a.f_in = p->kvec(lt, MGCMT_SLOT_F).p;
  a.v_out = p->kvec(lt, MGCMT_SLOT_V).p;
  a.vstride = p->kvec(lt, MGCMT_SLOT_V).stride;
  const long n = (long)p->levels[last].nr * p->levels[last].gc;
  a.inv = p->levels[last].band.inv;
  a.inv_stride = n * n;
  a.shifts = p->d_shifts;
  a.omega = omega;
  a.kind = kind;
  a.nu = nu;
  return a;
}

// The tail's matrix per vector for the current shifts (allocated on first use, redone when the shifts or the cycle's
// parameters change — like the coarsest level's factorisation, and like it never inside a graph capture: mgcmt_vcycle
// calls this eagerly before a capture or a replay).
int ensure_tail_matrix(mgcmt_plan* p, int lt, int kind, int nu, double omega, int k, hipStream_t s) {
  mgcmt_plan::TailMatrix& T = p->tailmat;
  const long n = (long)p->levels[lt].gr * p->levels[lt].gc;
  bool same = T.valid && T.lt == lt && T.kind == kind && T.nu == nu && T.omega == omega && T.k >= k && T.n == n;
  if (same)
    for (int q = 0; q < k; ++q) same = same && T.shifts[q] == p->h_shifts[q];
  if (same) return MGCMT_OK;
  const int last = (int)p->levels.size() - 1;
  MG_TRY(ensure_coarse_ready(p, last, k, s));
  if (T.capacity < k || T.n != n) {
    if (T.mt) {
      MG_HIP(hipStreamSynchronize(s));
      (void)hipFree(T.mt);
      T.mt = nullptr;
      p->graphs_invalidate();  // (cached graphs point at the old matrices)
    }
    MG_HIP(hipMalloc((void**)&T.mt, sizeof(double) * n * n * k));
    T.capacity = k;
  }
  const TailArgs a = tail_args(p, lt, kind, nu, omega);
  for (int q = 0; q < k; ++q) launch_tail_matrix(s, a, q, T.mt + (long)q * n * n);
  T.n = n;
  T.lt = lt;
  T.kind = kind;
  T.nu = nu;
  T.omega = omega;
  T.k = k;
  T.shifts.assign(p->h_shifts.begin(), p->h_shifts.begin() + k);
  T.valid = true;
  return post_launch();
}

bool tail_dense(const mgcmt_plan* p, int lt) { return p->use_tail_dense && lt > 0 && tail_dense_fits(p->levels[lt].gr); }

int run_tail(mgcmt_plan* p, int lt, int kind, int nu, double omega, int k, hipStream_t s) {
  const int last = (int)p->levels.size() - 1;
  MG_TRY(ensure_coarse_ready(p, last, k, s));
  MG_TRY(ensure_slot(p, lt, MGCMT_SLOT_V));
  MG_TRY(ensure_slot(p, lt, MGCMT_SLOT_F));
  if (tail_dense(p, lt)) {
    MG_TRY(ensure_tail_matrix(p, lt, kind, nu, omega, k, s));
    const long n = p->tailmat.n;
    launch_tail_dense(s, p->levels[lt].gr, p->tailmat.mt, n * n, p->kvec(lt, MGCMT_SLOT_F).p, p->kvec(lt, MGCMT_SLOT_V).p,
                      p->kvec(lt, MGCMT_SLOT_V).stride, k);
    return post_launch();
  }
  launch_tail(s, tail_args(p, lt, kind, nu, omega), k);
  return post_launch();
}

// what a cycle's tail needs ready outside a graph (the matrix of the dense form), for the cycle's parameters
int ensure_tail_for_cycle(mgcmt_plan* p, int level, int nu_coarse, int kind, double omega, int k, int cycle_flags, hipStream_t s) {
  const int lt = tail_level(p, level, kind, nu_coarse, cycle_flags & MGCMT_CYCLE_GRAM_SCHMIDT);
  if (lt > 0 && tail_dense(p, lt)) return ensure_tail_matrix(p, lt, kind, nu_coarse, omega, k, s);
  return MGCMT_OK;
}

int vcycle_body(mgcmt_plan* p, int level, int nu1, int nu2, int nu_coarse, int kind, double omega, int k, int cycle_flags,
                hipStream_t s) {
  const int gram_schmidt = cycle_flags & MGCMT_CYCLE_GRAM_SCHMIDT;
  // MGCMT_CYCLE_ZERO_START: the caller vouches that the iterate on `level` is zero — the first pass takes that as a
  // flag (V is neither cleared nor read; where no fused pass runs, down_leg clears it)
  const bool zero_start = (cycle_flags & MGCMT_CYCLE_ZERO_START) != 0;
  const int last = (int)p->levels.size() - 1;
  const int lt = tail_level(p, level, kind, nu_coarse, gram_schmidt);
  const int bottom = lt > 0 ? lt : last;  // the levels level .. bottom-1 run as fused passes / single launches
  std::vector<int> recompute(last + 1, 0);
  std::vector<char> still_zero(last + 1, 0);
  for (int l = level; l < bottom; ++l) {
    const int nu_up = l == level ? nu2 : nu_coarse;
    bool sz = false;
    // the up-leg can only recompute the unstored sweeps if it runs a fused pass itself (>= 1 post-smoothing sweep)
    MG_TRY(down_leg(p, l, kind, l == level ? nu1 : nu_coarse, omega, k, l > level || zero_start, s, nu_up >= 1 ? &recompute[l] : nullptr, &sz, nu_up));
    still_zero[l] = sz;
  }
  if (lt > 0) MG_TRY(run_tail(p, lt, kind, nu_coarse, omega, k, s));
  else MG_TRY(coarse_solve_impl(p, last, k, s));
  for (int l = bottom - 1; l >= level; --l) {
    MG_TRY(up_leg(p, l, kind, l == level ? nu2 : nu_coarse, omega, k, s, recompute[l], still_zero[l] != 0));
    if (gram_schmidt) MG_TRY(gramschmidt_impl(p, l, MGCMT_SLOT_V, k, 1, s));
  }
  return MGCMT_OK;
}

}  // namespace

// ================================================================================================
// C-ABI
// ================================================================================================

extern "C" {

const char* mgcmt_last_error(void) { return g_last_error.c_str(); }

int mgcmt_abi_version(void) { return MGCMT_ABI_VERSION; }

int mgcmt_device_count(int* count) {
  if (!count) return fail(MGCMT_ERR_INVALID, "null count");
  MG_HIP(hipGetDeviceCount(count));
  return MGCMT_OK;
}

int mgcmt_device_name(int device, char* buf, int buflen) {
  if (!buf || buflen <= 0) return fail(MGCMT_ERR_INVALID, "bad buffer");
  hipDeviceProp_t prop;
  MG_HIP(hipGetDeviceProperties(&prop, device));
  snprintf(buf, buflen, "%s (%s, %d CUs)", prop.name[0] ? prop.name : "AMD Instinct", prop.gcnArchName, prop.multiProcessorCount);
  return MGCMT_OK;
}

int mgcmt_plan_create(const mgcmt_plan_desc* d, mgcmt_plan** out) {
  if (!d || !out) return fail(MGCMT_ERR_INVALID, "null argument");
  *out = nullptr;
  if (d->dim != 1 && d->dim != 2) return fail(MGCMT_ERR_INVALID, "dim must be 1 or 2");
  if (!is_pow2(d->g) || !is_pow2(d->lowest) || d->lowest > d->g) return fail(MGCMT_ERR_INVALID, "g and lowest must be powers of two with lowest <= g");
  if (d->g < 2) return fail(MGCMT_ERR_INVALID, "Length of start vector is not a power of 2");
  if (d->lowest < 2) return fail(MGCMT_ERR_INVALID, "lowest must be at least 2");
  if (d->nterms < 1 || d->nterms > kMaxTerms || d->m_nterms < 0 || d->m_nterms > kMaxTerms) return fail(MGCMT_ERR_INVALID, "nterms out of range");
  if (!d->yfac || (d->dim == 2 && !d->xfac)) return fail(MGCMT_ERR_INVALID, "missing factor arrays");
  if (d->nvec < 1 || d->nvec > kMaxVec) return fail(MGCMT_ERR_INVALID, "nvec out of range (1..32)");
  int64_t rb = d->row_begin, re = d->row_end;
  if (d->dim == 1 || (rb == 0 && re == 0)) {
    rb = 0;
    re = d->dim == 2 ? d->g : 1;
  }
  if (d->dim == 2 && (rb < 0 || re > d->g || rb >= re)) return fail(MGCMT_ERR_INVALID, "bad row range");
  MG_HIP(hipSetDevice(d->device));

  mgcmt_plan* p = new mgcmt_plan();
  p->dim = d->dim;
  p->nvec = d->nvec;
  p->device = d->device;
  p->g = d->g;
  p->lowest = d->lowest;
  p->has_mass = d->m_nterms > 0;
  p->h_shifts.assign(kMaxVec, 0.0);
  {
    const char* mb = getenv("MGCMT_MGS_BLOCK_MIN");  // points per column from which Gram-Schmidt takes its two-pass form (tests, tuning)
    if (mb && atol(mb) > 1) p->mgs_block_min = atol(mb);
    const char* e = getenv("MGCMT_TAIL_DENSE");  // "0": the tail as the LDS-resident launch by default (the host-only test build:
    p->use_tail_dense = !(e && e[0] == '0');     // emulating the 1024 workgroups that form the matrix takes minutes)
  }

  int nlev = 1;
  for (int64_t s = d->g; s > d->lowest; s >>= 1) ++nlev;
  const bool whole = (rb == 0 && re == (d->dim == 2 ? d->g : 1));
  int strip_levels = whole ? 0 : (d->strip_levels > 0 ? d->strip_levels : nlev);
  if (strip_levels > nlev) strip_levels = nlev;
  if (!whole) {
    const int64_t align = (int64_t)1 << (strip_levels - 1);
    if (rb % align || re % align) {
      delete p;
      return fail(MGCMT_ERR_INVALID, "strip bounds must be multiples of 2^(strip_levels-1)");
    }
  }

  p->levels.resize(nlev);
  auto load = [&](const double* src, int m, int64_t n) {
    Tri t;
    t.n = n;
    t.a.assign(src + (size_t)m * 3 * n, src + (size_t)(m + 1) * 3 * n);
    return t;
  };
  for (int l = 0; l < nlev; ++l) {
    Level& L = p->levels[l];
    L.gc = d->g >> l;
    L.gr = d->dim == 2 ? (d->g >> l) : 1;
    if (l < strip_levels && !whole) {
      L.r0 = rb >> l;
      L.nr = (re - rb) >> l;
    } else {
      L.r0 = 0;
      L.nr = L.gr;
    }
    L.halo = d->dim == 2 ? kHalo : 1;
    L.stride = ((L.nr + 2 * L.halo) * L.gc + 31) / 32 * 32;
    auto build = [&](HostOp& h, const HostOp* finer, int nterms, const double* xf, const double* yf) {
      h.nterms = nterms;
      h.X.resize(nterms);
      h.Y.resize(nterms);
      for (int m = 0; m < nterms; ++m) {
        if (l == 0) {
          h.Y[m] = load(yf, m, d->g);
          h.X[m] = d->dim == 2 ? load(xf, m, d->g) : identity_tri(1);
        } else {
          h.Y[m] = galerkin(finer->Y[m]);
          h.X[m] = d->dim == 2 ? galerkin(finer->X[m]) : identity_tri(1);
        }
      }
    };
    build(L.hA, l ? &p->levels[l - 1].hA : nullptr, d->nterms, d->xfac, d->yfac);
    if (p->has_mass) build(L.hM, l ? &p->levels[l - 1].hM : nullptr, d->m_nterms, d->m_xfac, d->m_yfac);
    int rc = upload_op(L.hA, L, d->dim, &L.dA);
    if (rc == MGCMT_OK && p->has_mass) rc = upload_op(L.hM, L, d->dim, &L.dM);
    if (rc != MGCMT_OK) {
      mgcmt_plan_destroy(p);
      return rc;
    }
  }
  hipError_t e = hipMalloc((void**)&p->d_shifts, sizeof(double) * kMaxVec);
  if (e == hipSuccess) e = hipMalloc((void**)&p->d_zero, sizeof(double) * kMaxVec);
  if (e == hipSuccess) e = hipMalloc((void**)&p->d_partials, sizeof(double) * (kMaxVec + 1) * 1024 * 2);
  if (e == hipSuccess) e = hipMalloc((void**)&p->d_mgs, sizeof(double) * mgs_block_words());
  if (e == hipSuccess) e = hipMalloc((void**)&p->d_scalars, sizeof(double) * 4 * kMaxVec);
  if (e == hipSuccess) e = hipMemset(p->d_shifts, 0, sizeof(double) * kMaxVec);
  if (e == hipSuccess) e = hipMemset(p->d_zero, 0, sizeof(double) * kMaxVec);
  if (e != hipSuccess) {
    mgcmt_plan_destroy(p);
    return fail(MGCMT_ERR_HIP, std::string("plan scratch allocation: ") + hipGetErrorString(e));
  }
  *out = p;
  return MGCMT_OK;
}

int mgcmt_plan_destroy(mgcmt_plan* p) {
  if (!p) return MGCMT_OK;
  comm_release(p);
  for (Level& L : p->levels) {
    for (int s = 0; s < 4; ++s)
      if (L.base[s]) (void)hipFree(L.base[s]);
    for (double* q : L.dA.owned) (void)hipFree(q);
    for (double* q : L.dM.owned) (void)hipFree(q);
    if (L.band.b.ab) (void)hipFree(L.band.b.ab);
    if (L.band.b.piv) (void)hipFree(L.band.b.piv);
    if (L.band.inv) (void)hipFree(L.band.inv);
  }
  for (auto& g : p->graphs)
    if (g.second.exec) (void)hipGraphExecDestroy(g.second.exec);
  if (p->capture_stream) (void)hipStreamDestroy(p->capture_stream);
  if (p->d_rq) (void)hipFree(p->d_rq);
  if (p->d_rqstate) (void)hipFree(p->d_rqstate);
  if (p->d_rqhistory) (void)hipFree(p->d_rqhistory);
  if (p->d_mgs) (void)hipFree(p->d_mgs);
  if (p->tailmat.mt) (void)hipFree(p->tailmat.mt);
  if (p->lex_carry) (void)hipFree(p->lex_carry);
  if (p->lex_sync) (void)hipFree(p->lex_sync);
  if (p->d_shifts) (void)hipFree(p->d_shifts);
  if (p->d_zero) (void)hipFree(p->d_zero);
  if (p->d_partials) (void)hipFree(p->d_partials);
  if (p->d_scalars) (void)hipFree(p->d_scalars);
  delete p;
  return MGCMT_OK;
}

int mgcmt_plan_num_levels(const mgcmt_plan* p, int* levels) {
  if (!p || !levels) return fail(MGCMT_ERR_INVALID, "null argument");
  *levels = (int)p->levels.size();
  return MGCMT_OK;
}

int mgcmt_plan_level_shape(const mgcmt_plan* p, int l, int64_t* rows, int64_t* cols, int64_t* row_begin) {
  MG_TRY(check_level(p, l));
  if (rows) *rows = p->levels[l].nr;
  if (cols) *cols = p->levels[l].gc;
  if (row_begin) *row_begin = p->levels[l].r0;
  return MGCMT_OK;
}

int mgcmt_plan_get_factors(const mgcmt_plan* p, int op, int l, int which, double* out, int64_t capacity) {
  MG_TRY(check_level(p, l));
  const HostOp& h = op == MGCMT_OP_M ? p->levels[l].hM : p->levels[l].hA;
  if (op == MGCMT_OP_M && !p->has_mass) return fail(MGCMT_ERR_INVALID, "plan has no mass operator");
  const std::vector<Tri>& f = which == 0 ? h.X : h.Y;
  int64_t need = 0;
  for (const Tri& t : f) need += (int64_t)t.a.size();
  if (!out || capacity < need) return fail(MGCMT_ERR_INVALID, "factor buffer too small");
  int64_t o = 0;
  for (const Tri& t : f) {
    memcpy(out + o, t.a.data(), t.a.size() * sizeof(double));
    o += (int64_t)t.a.size();
  }
  return MGCMT_OK;
}

int mgcmt_plan_level_halo(const mgcmt_plan* p, int l, int* halo_rows, int* exchanged) {
  MG_TRY(check_level(p, l));
  if (halo_rows) *halo_rows = p->levels[l].halo;
  if (exchanged) *exchanged = exchanged_rows(p, l);
  return MGCMT_OK;
}

int mgcmt_vec_ptr(const mgcmt_plan* p, int l, int slot, int vec, void** device_ptr) {
  MG_TRY(check_vec(p, l, slot, vec));
  if (!device_ptr) return fail(MGCMT_ERR_INVALID, "null pointer");
  MG_TRY(ensure_slot(const_cast<mgcmt_plan*>(p), l, slot));
  *device_ptr = p->kvec(l, slot, vec).p;
  return MGCMT_OK;
}

int mgcmt_set_shifts(mgcmt_plan* p, const double* shifts, int k, void* stream) {
  if (!p || !shifts) return fail(MGCMT_ERR_INVALID, "null argument");
  MG_TRY(check_k(p, k));
  for (int q = 0; q < k; ++q) p->h_shifts[q] = shifts[q];
  // the host array may be reused immediately by the caller: stage through the plan's own copy
  MG_HIP(hipMemcpyAsync(p->d_shifts, p->h_shifts.data(), sizeof(double) * k, hipMemcpyHostToDevice, S(stream)));
  MG_HIP(hipStreamSynchronize(S(stream)));
  return MGCMT_OK;
}

int mgcmt_upload(mgcmt_plan* p, int l, int slot, int vec, const double* host, int64_t count, void* stream) {
  MG_TRY(check_vec(p, l, slot, vec));
  if (!host || count != p->interior(l)) return fail(MGCMT_ERR_INVALID, "upload: count must equal rows*cols of the level");
  MG_TRY(ensure_slot(p, l, slot));
  return transfer(p->device, true, p->kvec(l, slot, vec).p, const_cast<double*>(host), sizeof(double) * count, S(stream));
}

int mgcmt_download(mgcmt_plan* p, int l, int slot, int vec, double* host, int64_t count, void* stream) {
  MG_TRY(check_vec(p, l, slot, vec));
  if (!host || count != p->interior(l)) return fail(MGCMT_ERR_INVALID, "download: count must equal rows*cols of the level");
  MG_TRY(ensure_slot(p, l, slot));
  MG_TRY(transfer(p->device, false, p->kvec(l, slot, vec).p, host, sizeof(double) * count, S(stream)));
  return lex_wave_check(p);
}

int mgcmt_fill(mgcmt_plan* p, int l, int slot, int vec, double value, void* stream) {
  MG_TRY(check_vec(p, l, slot, vec));
  MG_TRY(ensure_slot(p, l, slot));
  launch_fill(S(stream), p->kvec(l, slot, vec).p, p->interior(l), value);
  return post_launch();
}

int mgcmt_zero(mgcmt_plan* p, int l, int slot, int vec, void* stream) {
  MG_TRY(check_vec(p, l, slot, vec));
  MG_TRY(ensure_slot(p, l, slot));
  const Level& L = p->levels[l];
  MG_HIP(hipMemsetAsync(p->kvec(l, slot, vec).p - (long)L.halo * L.gc, 0, sizeof(double) * (size_t)(L.nr + 2 * L.halo) * L.gc, S(stream)));
  return MGCMT_OK;
}

int mgcmt_copy(mgcmt_plan* p, int l, int src_slot, int src_vec, int dst_slot, int dst_vec, void* stream) {
  MG_TRY(check_vec(p, l, src_slot, src_vec));
  MG_TRY(check_vec(p, l, dst_slot, dst_vec));
  MG_TRY(ensure_slot(p, l, src_slot));
  MG_TRY(ensure_slot(p, l, dst_slot));
  MG_HIP(hipMemcpyAsync(p->kvec(l, dst_slot, dst_vec).p, p->kvec(l, src_slot, src_vec).p, sizeof(double) * p->interior(l),
                        hipMemcpyDeviceToDevice, S(stream)));
  return MGCMT_OK;
}

int mgcmt_sync(void* stream) {
  MG_HIP(hipStreamSynchronize(S(stream)));
  return MGCMT_OK;
}

int mgcmt_smooth(mgcmt_plan* p, int l, int kind, int nu, double omega, int k, void* stream) {
  MG_TRY(check_level(p, l));
  MG_TRY(check_k(p, k));
  if (nu < 0) return fail(MGCMT_ERR_INVALID, "nu must be >= 0");
  return smooth_impl(p, l, kind, nu, omega, k, S(stream));
}

int mgcmt_residual_restrict(mgcmt_plan* p, int l, int k, void* stream) {
  MG_TRY(check_level(p, l));
  MG_TRY(check_k(p, k));
  return residual_restrict_impl(p, l, k, S(stream));
}

int mgcmt_prolong_correct(mgcmt_plan* p, int l, int k, void* stream) {
  MG_TRY(check_level(p, l));
  MG_TRY(check_k(p, k));
  return prolong_correct_impl(p, l, k, S(stream));
}

int mgcmt_coarse_solve(mgcmt_plan* p, int l, int k, void* stream) {
  MG_TRY(check_level(p, l));
  MG_TRY(check_k(p, k));
  return coarse_solve_impl(p, l, k, S(stream));
}

int mgcmt_vcycle(mgcmt_plan* p, int level, int nu1, int nu2, int nu_coarse, int kind, double omega, int k, int cycle_flags,
                 void* stream) {
  MG_TRY(check_level(p, level));
  MG_TRY(check_k(p, k));
  if (nu1 < 0 || nu2 < 0 || nu_coarse < 0) return fail(MGCMT_ERR_INVALID, "sweep counts must be >= 0");
  if (cycle_flags & ~(MGCMT_CYCLE_GRAM_SCHMIDT | MGCMT_CYCLE_ZERO_START)) return fail(MGCMT_ERR_INVALID, "unknown cycle flag");
  hipStream_t s = S(stream);
  if (!p->use_graph) return vcycle_body(p, level, nu1, nu2, nu_coarse, kind, omega, k, cycle_flags, s);

  char buf[160];
  snprintf(buf, sizeof(buf), "%d/%d/%d/%d/%d/%.17g/%d/%d", level, nu1, nu2, nu_coarse, kind, omega, k, cycle_flags);
  const std::string params(buf);
  std::string key = params;
  for (const Level& L : p->levels) {
    snprintf(buf, sizeof(buf), "|%p,%p", (void*)L.base[MGCMT_SLOT_V], (void*)L.base[MGCMT_SLOT_T]);
    key += buf;
  }
  auto hit = p->graphs.find(key);
  if (hit != p->graphs.end()) {
    // the coarsest-level factorisation (and the tail's matrix) depend on the shift VALUES; redo them eagerly when they changed
    MG_TRY(ensure_coarse_factor(p, (int)p->levels.size() - 1, k, s));
    MG_TRY(ensure_tail_for_cycle(p, level, nu_coarse, kind, omega, k, cycle_flags, s));
    MG_HIP(hipGraphLaunch(hit->second.exec, s));
    // a replayed cycle runs the wave pipeline too: the next synchronising call must look at its error word
    if (hit->second.lex_wave) p->lex_wave_used = true;
    size_t i = 0;
    for (Level& L : p->levels) {
      L.base[MGCMT_SLOT_V] = hit->second.post_state[i++];
      L.base[MGCMT_SLOT_T] = hit->second.post_state[i++];
    }
    return MGCMT_OK;
  }
  // the first cycle with these parameters runs eagerly: it allocates, factors and queries occupancies
  if (p->cycle_seen[params]++ == 0) return vcycle_body(p, level, nu1, nu2, nu_coarse, kind, omega, k, cycle_flags, s);
  MG_TRY(ensure_coarse_factor(p, (int)p->levels.size() - 1, k, s));
  MG_TRY(ensure_tail_for_cycle(p, level, nu_coarse, kind, omega, k, cycle_flags, s));
  if (!p->capture_stream && hipStreamCreate(&p->capture_stream) != hipSuccess) {
    p->use_graph = false;
    (void)hipGetLastError();
    return vcycle_body(p, level, nu1, nu2, nu_coarse, kind, omega, k, cycle_flags, s);
  }
  if (hipStreamBeginCapture(p->capture_stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
    p->use_graph = false;
    (void)hipGetLastError();
    return vcycle_body(p, level, nu1, nu2, nu_coarse, kind, omega, k, cycle_flags, s);
  }
  const bool lex_before = p->lex_wave_used;
  p->lex_wave_used = false;
  const int rc = vcycle_body(p, level, nu1, nu2, nu_coarse, kind, omega, k, cycle_flags, p->capture_stream);
  const bool lex_captured = p->lex_wave_used;  // the captured body launches a wave-pipeline sweep
  p->lex_wave_used = lex_before || lex_captured;
  hipGraph_t graph = nullptr;
  const hipError_t end = hipStreamEndCapture(p->capture_stream, &graph);
  if (rc != MGCMT_OK || end != hipSuccess || !graph) {
    // nothing was executed during the capture, but the plan's buffer roles were advanced: cannot continue safely
    p->use_graph = false;
    if (graph) (void)hipGraphDestroy(graph);
    return rc != MGCMT_OK ? rc : fail(MGCMT_ERR_HIP, "graph capture of the V-cycle failed");
  }
  mgcmt_plan::CycleGraph cg;
  cg.lex_wave = lex_captured;
  const hipError_t inst = hipGraphInstantiate(&cg.exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (inst != hipSuccess) {
    p->use_graph = false;
    return fail(MGCMT_ERR_HIP, "hipGraphInstantiate failed");
  }
  for (const Level& L : p->levels) {
    cg.post_state.push_back(L.base[MGCMT_SLOT_V]);
    cg.post_state.push_back(L.base[MGCMT_SLOT_T]);
  }
  MG_HIP(hipGraphLaunch(cg.exec, s));
  p->graphs[key] = cg;
  return MGCMT_OK;
}

int mgcmt_twogrid(mgcmt_plan* p, int level, int nu1, int nu2, int kind, double omega, int k, void* stream) {
  MG_TRY(check_level(p, level));
  MG_TRY(check_k(p, k));
  if (level + 1 >= (int)p->levels.size()) return fail(MGCMT_ERR_INVALID, "twogrid needs a coarser level");
  hipStream_t s = S(stream);
  int recompute = 0;
  MG_TRY(down_leg(p, level, kind, nu1, omega, k, false, s, nu2 >= 1 ? &recompute : nullptr, nullptr, nu2));
  MG_TRY(coarse_solve_impl(p, level + 1, k, s));
  MG_TRY(up_leg(p, level, kind, nu2, omega, k, s, recompute));
  return MGCMT_OK;
}

int mgcmt_apply(mgcmt_plan* p, int op, int l, int src_slot, int src_vec, int dst_slot, int dst_vec, int with_shift, void* stream) {
  MG_TRY(check_vec(p, l, src_slot, src_vec));
  MG_TRY(check_vec(p, l, dst_slot, dst_vec));
  if (src_slot == dst_slot && src_vec == dst_vec) return fail(MGCMT_ERR_INVALID, "apply cannot run in place");
  if (op == MGCMT_OP_M && !p->has_mass) return fail(MGCMT_ERR_INVALID, "plan has no mass operator");
  MG_TRY(ensure_slot(p, l, src_slot));
  MG_TRY(ensure_slot(p, l, dst_slot));
  const KOp& k = op == MGCMT_OP_M ? p->levels[l].dM.k : p->levels[l].dA.k;
  const double* sh = with_shift ? p->d_shifts + src_vec : p->d_zero;
  launch_apply(S(stream), p->kgrid(l), k, p->kvec(l, src_slot, src_vec), p->kvec(l, dst_slot, dst_vec), sh, 1);
  return post_launch();
}

int mgcmt_restrict(mgcmt_plan* p, int l, int src_slot, int src_vec, int dst_slot, int dst_vec, void* stream) {
  MG_TRY(check_vec(p, l, src_slot, src_vec));
  MG_TRY(check_vec(p, l + 1, dst_slot, dst_vec));
  MG_TRY(ensure_slot(p, l, src_slot));
  MG_TRY(ensure_slot(p, l + 1, dst_slot));
  launch_restrict(S(stream), p->kgrid(l), p->kgrid(l + 1), p->kvec(l, src_slot, src_vec), p->kvec(l + 1, dst_slot, dst_vec), 1);
  return post_launch();
}

int mgcmt_prolong(mgcmt_plan* p, int l, int src_slot, int src_vec, int dst_slot, int dst_vec, int accumulate, void* stream) {
  MG_TRY(check_vec(p, l + 1, src_slot, src_vec));
  MG_TRY(check_vec(p, l, dst_slot, dst_vec));
  MG_TRY(ensure_slot(p, l + 1, src_slot));
  MG_TRY(ensure_slot(p, l, dst_slot));
  launch_prolong(S(stream), p->kgrid(l), p->kgrid(l + 1), p->kvec(l + 1, src_slot, src_vec), p->kvec(l, dst_slot, dst_vec), accumulate, 1);
  return post_launch();
}

int mgcmt_dot(mgcmt_plan* p, int l, int slot_a, int vec_a, int slot_b, int vec_b, double* host_out, void* stream) {
  MG_TRY(check_vec(p, l, slot_a, vec_a));
  MG_TRY(check_vec(p, l, slot_b, vec_b));
  if (!host_out) return fail(MGCMT_ERR_INVALID, "null output");
  MG_TRY(ensure_slot(p, l, slot_a));
  MG_TRY(ensure_slot(p, l, slot_b));
  launch_dots(S(stream), p->interior(l), p->kvec(l, slot_a, vec_a).p, p->kvec(l, slot_b, vec_b).p, 0, 1, p->d_partials, p->d_scalars);
  MG_TRY(post_launch());
  MG_HIP(hipMemcpyAsync(host_out, p->d_scalars, sizeof(double), hipMemcpyDeviceToHost, S(stream)));
  MG_HIP(hipStreamSynchronize(S(stream)));
  return MGCMT_OK;
}

int mgcmt_gram(mgcmt_plan* p, int l, int nv, const int* slots, const int* vecs, double* host_out, void* stream) {
  MG_TRY(check_level(p, l));
  if (!slots || !vecs || !host_out || nv < 1 || nv > kGramMaxVectors) return fail(MGCMT_ERR_INVALID, "gram: 1..6 vectors, non-null arguments");
  const double* v[kGramMaxVectors];
  for (int a = 0; a < nv; ++a) {
    MG_TRY(check_vec(p, l, slots[a], vecs[a]));
    MG_TRY(ensure_slot(p, l, slots[a]));
    v[a] = p->kvec(l, slots[a], vecs[a]).p;
  }
  constexpr int kPairs = kGramMaxVectors * (kGramMaxVectors + 1) / 2;
  launch_gram(S(stream), p->interior(l), v, nv, p->d_partials, p->d_scalars);
  MG_TRY(post_launch());
  double packed[kPairs];
  MG_HIP(hipMemcpyAsync(packed, p->d_scalars, sizeof(packed), hipMemcpyDeviceToHost, S(stream)));
  MG_HIP(hipStreamSynchronize(S(stream)));
  int t = 0;
  for (int a = 0; a < kGramMaxVectors; ++a)
    for (int b = a; b < kGramMaxVectors; ++b, ++t)
      if (b < nv) host_out[a * nv + b] = host_out[b * nv + a] = packed[t];
  return MGCMT_OK;
}

int mgcmt_ritz_pair(mgcmt_plan* p, int l, int xs, int xv, int ws, int wv, int ss, int sv, double* out5, void* stream) {
  MG_TRY(check_vec(p, l, xs, xv));
  MG_TRY(check_vec(p, l, ws, wv));
  MG_TRY(check_vec(p, l, ss, sv));
  if (!out5) return fail(MGCMT_ERR_INVALID, "null output");
  if ((ss == xs && sv == xv) || (ss == ws && sv == wv)) return fail(MGCMT_ERR_INVALID, "ritz_pair: the scratch vector must differ from x and w");
  MG_TRY(ensure_slot(p, l, xs));
  MG_TRY(ensure_slot(p, l, ws));
  hipStream_t s = S(stream);
  const double* x = p->kvec(l, xs, xv).p;
  const double* w = p->kvec(l, ws, wv).p;
  if (launch_ritz_pair(s, p->kgrid(l), p->levels[l].dA.k, x, w, p->d_partials, p->d_scalars)) {
    MG_TRY(post_launch());
    MG_HIP(hipMemcpyAsync(out5, p->d_scalars, sizeof(double) * 5, hipMemcpyDeviceToHost, s));
    MG_HIP(hipStreamSynchronize(s));
    return MGCMT_OK;
  }
  MG_TRY(ensure_slot(p, l, ss));
  launch_apply(s, p->kgrid(l), p->levels[l].dA.k, p->kvec(l, ws, wv), p->kvec(l, ss, sv), p->d_zero, 1);
  const double* v[kGramMaxVectors] = {x, w, p->kvec(l, ss, sv).p};
  launch_gram(s, p->interior(l), v, 3, p->d_partials, p->d_scalars);
  MG_TRY(post_launch());
  constexpr int kPairs = kGramMaxVectors * (kGramMaxVectors + 1) / 2;
  double packed[kPairs];
  MG_HIP(hipMemcpyAsync(packed, p->d_scalars, sizeof(packed), hipMemcpyDeviceToHost, s));
  MG_HIP(hipStreamSynchronize(s));
  // packed order: (0,0),(0,1),...,(0,5),(1,1),(1,2),...
  out5[0] = packed[0];
  out5[1] = packed[1];
  out5[2] = packed[kGramMaxVectors];
  out5[3] = packed[2];
  out5[4] = packed[kGramMaxVectors + 1];
  return MGCMT_OK;
}

int mgcmt_rayleigh_residual(mgcmt_plan* p, int l, int slot, int k, double* rq_out, double* res_out, void* stream) {
  MG_TRY(check_vec(p, l, slot, 0));
  MG_TRY(check_k(p, k));
  if (slot == MGCMT_SLOT_W) return fail(MGCMT_ERR_INVALID, "rayleigh_residual uses slot W as its scratch");
  if (!rq_out && !res_out) return fail(MGCMT_ERR_INVALID, "null outputs");
  MG_TRY(ensure_slot(p, l, slot));
  MG_TRY(ensure_slot(p, l, MGCMT_SLOT_W));
  constexpr int kPairs = kGramMaxVectors * (kGramMaxVectors + 1) / 2;
  if (!p->d_rq) MG_HIP(hipMalloc((void**)&p->d_rq, sizeof(double) * kPairs * kMaxVec));
  hipStream_t s = S(stream);
  // W_q = (A - mu_q I) v_q for all columns in one launch, then per column <v,v>, <v,r>, <r,r> in one pass each; the
  // host sees all of them after ONE synchronisation
  launch_apply(s, p->kgrid(l), p->levels[l].dA.k, p->kvec(l, slot), p->kvec(l, MGCMT_SLOT_W), p->d_shifts, k);
  for (int q = 0; q < k; ++q) {
    const double* v[kGramMaxVectors] = {p->kvec(l, slot, q).p, p->kvec(l, MGCMT_SLOT_W, q).p};
    launch_gram(s, p->interior(l), v, 2, p->d_partials, p->d_rq + (long)q * kPairs);
  }
  MG_TRY(post_launch());
  std::vector<double> packed((size_t)kPairs * k);
  MG_HIP(hipMemcpyAsync(packed.data(), p->d_rq, sizeof(double) * kPairs * k, hipMemcpyDeviceToHost, s));
  MG_HIP(hipStreamSynchronize(s));
  for (int q = 0; q < k; ++q) {
    const double vv = packed[(size_t)q * kPairs + 0], vr = packed[(size_t)q * kPairs + 1], rr = packed[(size_t)q * kPairs + kGramMaxVectors];
    if (rq_out) rq_out[q] = p->h_shifts[q] + vr / vv;
    if (res_out) res_out[q] = std::sqrt(rr);
  }
  return MGCMT_OK;
}

// rqmin (MGCMTSolver.py:17-57) on `level`, entirely on the device: two passes over the data per step (kernels_rq.hip), the
// 2 x 2 pencil solved by one workgroup, no host round trip; the start vector is vecs[0] of `slot`, which also receives
// the result; vecs[1..5]: five more vectors of the slot as work space (x and p are ping-ponged; g; one for M g).
static int rqmin_check(mgcmt_plan* p, int l, int slot, const int* vecs, int nu) {
  MG_TRY(check_level(p, l));
  if (!vecs || nu < 0) return fail(MGCMT_ERR_INVALID, "rqmin: null vector list or negative step count");
  for (int a = 0; a < 6; ++a) {
    MG_TRY(check_vec(p, l, slot, vecs[a]));
    for (int b = 0; b < a; ++b)
      if (vecs[a] == vecs[b]) return fail(MGCMT_ERR_INVALID, "rqmin: the six vectors must be distinct");
  }
  MG_TRY(ensure_slot(p, l, slot));
  if (!p->d_rqstate) {
    MG_HIP(hipMalloc((void**)&p->d_rqstate, sizeof(double) * rq_state_words()));
    MG_HIP(hipMemset(p->d_rqstate, 0, sizeof(double) * rq_state_words()));
  }
  return MGCMT_OK;
}

// M: the plan's mass operator; none, or one whose factors are identities, is the identity (no application at all)
static bool mass_is_identity(const mgcmt_plan* p, int l) {
  if (!p->has_mass) return true;
  const Level& L = p->levels[l];
  auto is_identity = [](const Tri& t) {
    for (int64_t i = 0; i < t.n; ++i)
      if (t.di(i) != 1.0 || (i > 0 && t.lo(i) != 0.0) || (i + 1 < t.n && t.up(i) != 0.0)) return false;
    return true;
  };
  return L.hM.nterms == 1 && is_identity(L.hM.X[0]) && is_identity(L.hM.Y[0]);
}

static int rqmin_impl(mgcmt_plan* p, int l, int slot, const int* vecs, int nu, int robust, hipStream_t s) {
  const Level& L = p->levels[l];
  const KOp& A = L.dA.k;
  const bool m_identity = mass_is_identity(p, l);
  const KOp& Mo = p->has_mass ? L.dM.k : A;  // (not read when M is the identity)
  const KGrid g = p->kgrid(l);
  double* x = p->kvec(l, slot, vecs[0]).p;
  double* xalt = p->kvec(l, slot, vecs[1]).p;
  double* pv = p->kvec(l, slot, vecs[2]).p;
  double* palt = p->kvec(l, slot, vecs[3]).p;
  double* gv = p->kvec(l, slot, vecs[4]).p;
  double* tmp = p->kvec(l, slot, vecs[5]).p;
  double* st = p->d_rqstate;
  double* part = p->d_partials;
  double* part_dot = p->d_partials + 40000;  // (d_partials holds 67584 doubles: 8 x 4096 for the passes, 1024 for the dot)
  const long n = p->interior(l);
  double* x0 = x;
  // a level of a few thousand points: the whole call in one launch (MGCMT_RQ_SMALL=0: the passes, for A/B measurements)
  const char* small_env = getenv("MGCMT_RQ_SMALL");  // (read per call: the tests compare both forms in one process)
  const bool small_ok = !(small_env && small_env[0] == '0');
  if (small_ok && launch_rq_small(s, g, A, Mo, m_identity ? 1 : 0, x, pv, gv, st, nu, robust)) return post_launch();
  for (int it = -1; it < nu; ++it) {
    const int init = it < 0 ? 1 : (it == 0 ? 2 : 0);
    launch_rq_pass1(s, g, A, Mo, m_identity ? 1 : 0, x, gv, pv, palt, st, init, robust, part);
    if (init != 1) std::swap(pv, palt);
    const int nb = launch_rq_pass2(s, g, A, Mo, m_identity ? 1 : 0, x, pv, xalt, gv, st, init, part);
    if (init != 1) std::swap(x, xalt);  // (the initial pair leaves x where it is)
    int mflag = m_identity ? 1 : 0;
    if (!m_identity) {
      // <g, M g>: one more march over g (nothing stored) where the level takes the march; application + dot product elsewhere
      if (launch_rq_gmg(s, g, Mo, gv, part, nb)) {
        mflag = 2;
      } else {
        launch_apply(s, g, Mo, KVec{gv, 0}, KVec{tmp, 0}, p->d_zero, 1);
        launch_dots(s, n, gv, tmp, 0, 1, part_dot, st + rq_word_gmg());
      }
    }
    launch_rq_scalars2(s, part, nb, st, mflag, init);
  }
  MG_TRY(post_launch());
  if (x != x0) MG_HIP(hipMemcpyAsync(x0, x, sizeof(double) * n, hipMemcpyDeviceToDevice, s));
  return MGCMT_OK;
}

static int rq_result(mgcmt_plan* p, double* rho_out, hipStream_t s) {
  if (!rho_out) return MGCMT_OK;
  MG_HIP(hipMemcpyAsync(rho_out, p->d_rqstate + rq_word_rho(), sizeof(double), hipMemcpyDeviceToHost, s));
  MG_HIP(hipStreamSynchronize(s));
  return MGCMT_OK;
}

int mgcmt_rqmin(mgcmt_plan* p, int l, int slot, const int* vecs, int nu, int robust, double* rho_out, void* stream) {
  MG_TRY(rqmin_check(p, l, slot, vecs, nu));
  MG_TRY(rqmin_impl(p, l, slot, vecs, nu, robust, S(stream)));
  return rq_result(p, rho_out, S(stream));
}

// One line minimisation of the Rayleigh quotient along a direction the CALLER supplies (see mgcmt_hip.h): the passes of
// rqmin with p = w read as it is and not stored, then x' = x + delta w and its gradient.
int mgcmt_rq_line_step(mgcmt_plan* p, int l, const int* xv, const int* wv, const int* xoutv, const int* gv, const int* tmpv, int robust, int record,
                       void* stream) {
  MG_TRY(check_level(p, l));
  if (!xv || !gv) return fail(MGCMT_ERR_INVALID, "rq_line_step: x and g are required");
  if (wv && !xoutv) return fail(MGCMT_ERR_INVALID, "rq_line_step: a step needs a vector for x + delta w");
  const int* all[5] = {xv, wv, xoutv, gv, tmpv};
  for (int a = 0; a < 5; ++a) {
    if (!all[a]) continue;
    MG_TRY(check_vec(p, l, all[a][0], all[a][1]));
    MG_TRY(ensure_slot(p, l, all[a][0]));
    for (int b = 0; b < a; ++b)
      if (all[b] && all[a][0] == all[b][0] && all[a][1] == all[b][1]) return fail(MGCMT_ERR_INVALID, "rq_line_step: the vectors must be distinct");
  }
  if (record >= MGCMT_RQ_HISTORY) return fail(MGCMT_ERR_INVALID, "rq_line_step: history index out of range");
  if (!p->d_rqstate) {
    MG_HIP(hipMalloc((void**)&p->d_rqstate, sizeof(double) * rq_state_words()));
    MG_HIP(hipMemset(p->d_rqstate, 0, sizeof(double) * rq_state_words()));
  }
  if (record >= 0 && !p->d_rqhistory) MG_HIP(hipMalloc((void**)&p->d_rqhistory, sizeof(double) * MGCMT_RQ_HISTORY));
  hipStream_t s = S(stream);
  const Level& L = p->levels[l];
  const KOp& A = L.dA.k;
  const bool m_identity = mass_is_identity(p, l);
  if (!m_identity && !tmpv) return fail(MGCMT_ERR_INVALID, "rq_line_step: with a mass operator a work vector (for M g) is required");
  const KOp& Mo = p->has_mass ? L.dM.k : A;
  const KGrid g = p->kgrid(l);
  const double* x = p->kvec(l, xv[0], xv[1]).p;
  const double* w = wv ? p->kvec(l, wv[0], wv[1]).p : nullptr;
  double* xout = xoutv ? p->kvec(l, xoutv[0], xoutv[1]).p : nullptr;
  double* gout = p->kvec(l, gv[0], gv[1]).p;
  double* st = p->d_rqstate;
  double* part = p->d_partials;
  // without a direction: the initial pair of rqmin (rho and g of x); with one: pass 1 reads p = w (init 3), pass 2 is a step's
  const int init1 = w ? 3 : 1, init2 = w ? 0 : 1;
  launch_rq_pass1(s, g, A, Mo, m_identity ? 1 : 0, x, w, nullptr, nullptr, st, init1, robust, part);
  const int nb = launch_rq_pass2(s, g, A, Mo, m_identity ? 1 : 0, x, w, xout, gout, st, init2, part);
  int mflag = m_identity ? 1 : 0;
  if (!m_identity) {
    if (launch_rq_gmg(s, g, Mo, gout, part, nb)) {
      mflag = 2;
    } else {
      double* tmp = p->kvec(l, tmpv[0], tmpv[1]).p;
      launch_apply(s, g, Mo, KVec{gout, 0}, KVec{tmp, 0}, p->d_zero, 1);
      launch_dots(s, p->interior(l), gout, tmp, 0, 1, p->d_partials + 40000, st + rq_word_gmg());
    }
  }
  launch_rq_scalars2(s, part, nb, st, mflag, init2);
  MG_TRY(post_launch());
  if (record >= 0) MG_HIP(hipMemcpyAsync(p->d_rqhistory + record, st + rq_word_rho(), sizeof(double), hipMemcpyDeviceToDevice, s));
  return MGCMT_OK;
}

int mgcmt_rq_history(mgcmt_plan* p, int first, int count, double* out, void* stream) {
  if (!p || !out || first < 0 || count < 0 || first + count > MGCMT_RQ_HISTORY) return fail(MGCMT_ERR_INVALID, "rq_history: bad range");
  if (count == 0) return MGCMT_OK;
  if (!p->d_rqhistory) return fail(MGCMT_ERR_INVALID, "rq_history: nothing recorded");
  MG_HIP(hipMemcpyAsync(out, p->d_rqhistory + first, sizeof(double) * count, hipMemcpyDeviceToHost, S(stream)));
  MG_HIP(hipStreamSynchronize(S(stream)));
  return MGCMT_OK;
}

// vcycle_rqmg (MGCMTSolver.py:99-122): rqmin, the ITERATE restricted (:113), the recursion on the Galerkin pair (R A P,
// R M P) of the next level, the interpolated coarse iterate added (:116-118), rqmin again — down to the plan's coarsest
// level, which only minimises.  One stream-ordered launch sequence without a host round trip, replayed as a HIP graph
// from its second call (the levels below 512^2 are pure launch latency: seven launches per step).
static int rqmg_body(mgcmt_plan* p, int l, int slot, const int* vecs, int nu1, int nu2, int robust, hipStream_t s) {
  const int last = (int)p->levels.size() - 1;
  MG_TRY(rqmin_impl(p, l, slot, vecs, nu1, robust, s));
  if (l == last) return MGCMT_OK;
  launch_restrict(s, p->kgrid(l), p->kgrid(l + 1), p->kvec(l, slot, vecs[0]), p->kvec(l + 1, slot, vecs[0]), 1);
  MG_TRY(rqmg_body(p, l + 1, slot, vecs, nu1, nu2, robust, s));
  launch_prolong(s, p->kgrid(l), p->kgrid(l + 1), p->kvec(l + 1, slot, vecs[0]), p->kvec(l, slot, vecs[0]), 1, 1);
  MG_TRY(post_launch());
  return rqmin_impl(p, l, slot, vecs, nu2, robust, s);
}

int mgcmt_vcycle_rqmg(mgcmt_plan* p, int slot, const int* vecs, int nu1, int nu2, int robust, double* rho_out, void* stream) {
  if (!p) return fail(MGCMT_ERR_INVALID, "null plan");
  if (nu1 < 0 || nu2 < 0) return fail(MGCMT_ERR_INVALID, "step counts must be >= 0");
  for (int l = 0; l < (int)p->levels.size(); ++l) MG_TRY(rqmin_check(p, l, slot, vecs, nu1));
  hipStream_t s = S(stream);
  char buf[200];
  snprintf(buf, sizeof(buf), "rqmg/%d/%d/%d/%d/%d,%d,%d,%d,%d,%d", nu1, nu2, robust, slot, vecs[0], vecs[1], vecs[2], vecs[3], vecs[4], vecs[5]);
  const std::string params(buf);
  std::string key = params;
  for (const Level& L : p->levels) {
    snprintf(buf, sizeof(buf), "|%p", (void*)L.base[slot]);
    key += buf;
  }
  auto eager = [&]() {
    MG_TRY(rqmg_body(p, 0, slot, vecs, nu1, nu2, robust, s));
    return rq_result(p, rho_out, s);
  };
  if (!p->use_graph) return eager();
  auto hit = p->graphs.find(key);
  if (hit != p->graphs.end()) {
    MG_HIP(hipGraphLaunch(hit->second.exec, s));
    return rq_result(p, rho_out, s);
  }
  if (p->cycle_seen[params]++ == 0) return eager();  // the first call allocates and queries occupancies
  if (!p->capture_stream && hipStreamCreate(&p->capture_stream) != hipSuccess) {
    (void)hipGetLastError();
    return eager();
  }
  if (hipStreamBeginCapture(p->capture_stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
    (void)hipGetLastError();
    return eager();
  }
  const int rc = rqmg_body(p, 0, slot, vecs, nu1, nu2, robust, p->capture_stream);
  hipGraph_t graph = nullptr;
  const hipError_t end = hipStreamEndCapture(p->capture_stream, &graph);
  if (rc != MGCMT_OK || end != hipSuccess || !graph) {
    if (graph) (void)hipGraphDestroy(graph);
    (void)hipGetLastError();
    if (rc != MGCMT_OK) return rc;
    return eager();  // (nothing ran during the capture and the body keeps no host-side state: run it for real)
  }
  mgcmt_plan::CycleGraph cg;
  const hipError_t inst = hipGraphInstantiate(&cg.exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (inst != hipSuccess) {
    (void)hipGetLastError();
    return eager();
  }
  MG_HIP(hipGraphLaunch(cg.exec, s));
  p->graphs[key] = cg;
  return rq_result(p, rho_out, s);
}

int mgcmt_lincomb(mgcmt_plan* p, int l, int nterms, const double* coeffs, const int* slots, const int* vecs, int dst_slot, int dst_vec,
                  void* stream) {
  MG_TRY(check_level(p, l));
  if (!coeffs || !slots || !vecs || nterms < 1 || nterms > 4) return fail(MGCMT_ERR_INVALID, "lincomb: 1..4 terms, non-null arguments");
  MG_TRY(check_vec(p, l, dst_slot, dst_vec));
  MG_TRY(ensure_slot(p, l, dst_slot));
  const double* v[4];
  for (int t = 0; t < nterms; ++t) {
    MG_TRY(check_vec(p, l, slots[t], vecs[t]));
    MG_TRY(ensure_slot(p, l, slots[t]));
    v[t] = p->kvec(l, slots[t], vecs[t]).p;
  }
  launch_lincomb(S(stream), p->interior(l), v, coeffs, nterms, p->kvec(l, dst_slot, dst_vec).p);
  return post_launch();
}

int mgcmt_block_gram(mgcmt_plan* p, int l, int na, const int* a_slots, const int* a_vecs, int nb, const int* b_slots, const int* b_vecs,
                     double* host_out, void* stream) {
  MG_TRY(check_level(p, l));
  if (!a_slots || !a_vecs || !b_slots || !b_vecs || !host_out || na < 1 || na > kBlockMaxA || nb < 1 || nb > kBlockMaxB)
    return fail(MGCMT_ERR_INVALID, "block_gram: 1..12 by 1..4 vectors, non-null arguments");
  const double *a[kBlockMaxA], *b[kBlockMaxB];
  for (int t = 0; t < na; ++t) {
    MG_TRY(check_vec(p, l, a_slots[t], a_vecs[t]));
    MG_TRY(ensure_slot(p, l, a_slots[t]));
    a[t] = p->kvec(l, a_slots[t], a_vecs[t]).p;
  }
  for (int t = 0; t < nb; ++t) {
    MG_TRY(check_vec(p, l, b_slots[t], b_vecs[t]));
    MG_TRY(ensure_slot(p, l, b_slots[t]));
    b[t] = p->kvec(l, b_slots[t], b_vecs[t]).p;
  }
  launch_block_gram(S(stream), p->interior(l), a, na, b, nb, p->d_partials, p->d_scalars);
  MG_TRY(post_launch());
  double packed[kBlockMaxA * kBlockMaxB];
  MG_HIP(hipMemcpyAsync(packed, p->d_scalars, sizeof(packed), hipMemcpyDeviceToHost, S(stream)));
  MG_HIP(hipStreamSynchronize(S(stream)));
  for (int i = 0; i < na; ++i)
    for (int j = 0; j < nb; ++j) host_out[i * nb + j] = packed[i * kBlockMaxB + j];
  return MGCMT_OK;
}

int mgcmt_block_combine(mgcmt_plan* p, int l, int nin, const int* in_slots, const int* in_vecs, int nout, const int* out_slots,
                        const int* out_vecs, const double* coeffs, void* stream) {
  MG_TRY(check_level(p, l));
  if (!in_slots || !in_vecs || !out_slots || !out_vecs || !coeffs || nin < 1 || nin > kBlockMaxA || nout < 1 || nout > kBlockMaxB)
    return fail(MGCMT_ERR_INVALID, "block_combine: 1..12 inputs, 1..4 outputs, non-null arguments");
  const double* in[kBlockMaxA];
  double* out[kBlockMaxB];
  for (int t = 0; t < nin; ++t) {
    MG_TRY(check_vec(p, l, in_slots[t], in_vecs[t]));
    MG_TRY(ensure_slot(p, l, in_slots[t]));
    in[t] = p->kvec(l, in_slots[t], in_vecs[t]).p;
  }
  for (int j = 0; j < nout; ++j) {
    MG_TRY(check_vec(p, l, out_slots[j], out_vecs[j]));
    MG_TRY(ensure_slot(p, l, out_slots[j]));
    out[j] = p->kvec(l, out_slots[j], out_vecs[j]).p;
    for (int i = 0; i < j; ++i)
      if (out[i] == out[j]) return fail(MGCMT_ERR_INVALID, "block_combine: an output vector named twice");
  }
  launch_block_combine(S(stream), p->interior(l), in, nin, out, nout, coeffs);
  return post_launch();
}

int mgcmt_axpy(mgcmt_plan* p, int l, double alpha, int x_slot, int x_vec, int y_slot, int y_vec, void* stream) {
  MG_TRY(check_vec(p, l, x_slot, x_vec));
  MG_TRY(check_vec(p, l, y_slot, y_vec));
  MG_TRY(ensure_slot(p, l, x_slot));
  MG_TRY(ensure_slot(p, l, y_slot));
  launch_axpy(S(stream), p->interior(l), alpha, p->kvec(l, x_slot, x_vec).p, p->kvec(l, y_slot, y_vec).p);
  return post_launch();
}

int mgcmt_scale(mgcmt_plan* p, int l, double alpha, int slot, int vec, void* stream) {
  MG_TRY(check_vec(p, l, slot, vec));
  MG_TRY(ensure_slot(p, l, slot));
  launch_scale(S(stream), p->interior(l), alpha, p->kvec(l, slot, vec).p);
  return post_launch();
}

int mgcmt_gramschmidt(mgcmt_plan* p, int l, int slot, int k, int modified, void* stream) {
  MG_TRY(check_vec(p, l, slot, 0));
  MG_TRY(check_k(p, k));
  return gramschmidt_impl(p, l, slot, k, modified, S(stream));
}

int mgcmt_normalize(mgcmt_plan* p, int l, int slot, int k, void* stream) {
  MG_TRY(check_vec(p, l, slot, 0));
  MG_TRY(check_k(p, k));
  MG_TRY(ensure_slot(p, l, slot));
  for (int i = 0; i < k; ++i) {
    double* a = p->kvec(l, slot, i).p;
    launch_dot_partials(S(stream), p->interior(l), a, a, 0, 1, p->d_partials);
    launch_scale_by_norm(S(stream), p->interior(l), p->d_partials, a);
  }
  return post_launch();
}

static int fused_pass_checked(mgcmt_plan* p, int l, int kind, int nsweep, double omega, int mode, int k, hipStream_t s, int reps) {
  MG_TRY(check_level(p, l));
  MG_TRY(check_k(p, k));
  if (!fused_level(p, l, kind)) return fail(MGCMT_ERR_UNSUPPORTED, "level / smoother not covered by the fused kernels");
  const int transfer = mode & 3, npre = (mode >> 4) & 3;
  if (nsweep < 1 || nsweep > pass_sweeps(p, l, kind, nsweep) || mode < 0 || mode > 63 || transfer == 3 || ((mode & 8) && transfer != 2) ||
      (npre && transfer != 1) || npre > fused_max_recompute(p->levels[l].dA.k, kind == MGCMT_GS_MC ? 1 : 0, nsweep) ||
      (transfer == 1 && (mode & 4) && !npre))
    return fail(MGCMT_ERR_INVALID, "bad nsweep or mode");
  if ((mode & 3) != 0 && l + 1 >= (int)p->levels.size()) return fail(MGCMT_ERR_INVALID, "no coarser level");
  MG_TRY(ensure_slot(p, l, MGCMT_SLOT_V));
  MG_TRY(ensure_slot(p, l, MGCMT_SLOT_F));
  MG_TRY(ensure_slot(p, l, MGCMT_SLOT_T));
  if ((mode & 3) != 0) {
    MG_TRY(ensure_slot(p, l + 1, MGCMT_SLOT_V));
    MG_TRY(ensure_slot(p, l + 1, MGCMT_SLOT_F));
  }
  for (int r = 0; r < reps; ++r) MG_TRY(fused_pass(p, l, kind, nsweep, omega, mode & 15, k, s, npre));
  return post_launch();
}

int mgcmt_fused_pass(mgcmt_plan* p, int l, int kind, int nsweep, double omega, int mode, int k, void* stream) {
  return fused_pass_checked(p, l, kind, nsweep, omega, mode, k, S(stream), 1);
}

int mgcmt_time_fused_pass(mgcmt_plan* p, int l, int kind, int nsweep, double omega, int mode, int reps, double* ms_out, void* stream) {
  if (!ms_out || reps < 1) return fail(MGCMT_ERR_INVALID, "bad arguments");
  MG_TRY(fused_pass_checked(p, l, kind, nsweep, omega, mode, 1, S(stream), 1));  // (allocations, occupancy query: untimed)
  hipEvent_t a, b;
  MG_HIP(hipEventCreate(&a));
  MG_HIP(hipEventCreate(&b));
  MG_HIP(hipEventRecord(a, S(stream)));
  const int rc = fused_pass_checked(p, l, kind, nsweep, omega, mode, 1, S(stream), reps);
  MG_HIP(hipEventRecord(b, S(stream)));
  MG_HIP(hipEventSynchronize(b));
  float ms = 0.f;
  MG_HIP(hipEventElapsedTime(&ms, a, b));
  *ms_out = (double)ms / reps;
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  return rc;
}

int mgcmt_fused_max_sweeps(const mgcmt_plan* p, int l, int kind, int* max_sweeps) {
  MG_TRY(check_level(p, l));
  if (!max_sweeps) return fail(MGCMT_ERR_INVALID, "null output");
  *max_sweeps = fused_level(p, l, kind) ? pass_sweeps(p, l, kind, 1 << 20) : 0;
  return MGCMT_OK;
}

int mgcmt_fused_max_recompute(const mgcmt_plan* p, int l, int kind, int nsweep, int* max_recompute) {
  MG_TRY(check_level(p, l));
  if (!max_recompute) return fail(MGCMT_ERR_INVALID, "null output");
  *max_recompute = fused_level(p, l, kind) ? fused_max_recompute(p->levels[l].dA.k, kind == MGCMT_GS_MC ? 1 : 0, nsweep) : 0;
  return MGCMT_OK;
}

int mgcmt_level_operator_kind(const mgcmt_plan* p, int l, int* kind) {
  MG_TRY(check_level(p, l));
  if (!kind) return fail(MGCMT_ERR_INVALID, "null output");
  const KOp& k = p->levels[l].dA.k;
  *kind = k.five_point ? MGCMT_OPK_FIVE_POINT : k.five_diag ? MGCMT_OPK_FIVE_DIAG : k.nine_const ? MGCMT_OPK_NINE_CONST : k.nine_var ? MGCMT_OPK_NINE_VAR : MGCMT_OPK_GENERAL;
  return MGCMT_OK;
}

int mgcmt_plan_set_option(mgcmt_plan* p, int option, int value) {
  if (!p) return fail(MGCMT_ERR_INVALID, "null plan");
  if (option == MGCMT_OPT_FUSED) {
    p->use_fused = value != 0;
    p->graphs_invalidate();
    return MGCMT_OK;
  }
  if (option == MGCMT_OPT_RECOMPUTE) {
    p->use_recompute = value != 0;
    p->force_recompute = value == 2;
    p->graphs_invalidate();
    return MGCMT_OK;
  }
  if (option == MGCMT_OPT_GRAPH) {
    p->use_graph = value != 0;
    return MGCMT_OK;
  }
  if (option == MGCMT_OPT_LEX_WAVE) {
    p->use_lex_wave = value < 0 ? 0 : (value > 2 ? 2 : (int)value);
    p->graphs_invalidate();
    return MGCMT_OK;
  }
  if (option == MGCMT_OPT_LEX_CHAIN) {
    p->lex_chain = value != 0;
    p->graphs_invalidate();
    return MGCMT_OK;
  }
  if (option == MGCMT_OPT_TAIL) {
    p->use_tail = value != 0;
    p->use_tail_dense = value != 2;  // 1 (default): the tail as one dense product; 2: as the LDS-resident launch of ~45 phases
    p->graphs_invalidate();
    return MGCMT_OK;
  }
  if (option == MGCMT_OPT_MGS_BLOCK) {
    p->use_mgs_block = value != 0;
    p->mgs_block_min = value > 1 ? (long)value : 0;  // (a value > 1: the blocked form from that many points on — tuning)
    p->graphs_invalidate();
    return MGCMT_OK;
  }
  if (option == MGCMT_OPT_FUSED_ROWS) {
    p->fused_rows = value;
    p->graphs_invalidate();
    return MGCMT_OK;
  }
  return fail(MGCMT_ERR_INVALID, "unknown option");
}

int mgcmt_lex_wave_stats(mgcmt_plan* p, uint32_t* out, int64_t capacity) {
  if (!p || !out) return fail(MGCMT_ERR_INVALID, "null argument");
  if (!p->lex_sync) return fail(MGCMT_ERR_INVALID, "no lexicographic wave sweep has run on this plan");
  const int64_t n = capacity < (int64_t)p->lex_sync_words ? capacity : (int64_t)p->lex_sync_words;
  MG_HIP(hipDeviceSynchronize());
  MG_HIP(hipMemcpy(out, p->lex_sync, sizeof(uint32_t) * n, hipMemcpyDeviceToHost));
  return MGCMT_OK;
}

int mgcmt_time_smoother(mgcmt_plan* p, int l, int kind, int nu, double omega, int reps, double* ms_out, void* stream) {
  MG_TRY(check_level(p, l));
  if (!ms_out || reps < 1) return fail(MGCMT_ERR_INVALID, "bad arguments");
  hipEvent_t a, b;
  MG_HIP(hipEventCreate(&a));
  MG_HIP(hipEventCreate(&b));
  MG_HIP(hipEventRecord(a, S(stream)));
  for (int r = 0; r < reps; ++r) MG_TRY(smooth_impl(p, l, kind, nu, omega, 1, S(stream)));
  MG_HIP(hipEventRecord(b, S(stream)));
  MG_HIP(hipEventSynchronize(b));
  float ms = 0.f;
  MG_HIP(hipEventElapsedTime(&ms, a, b));
  *ms_out = (double)ms;
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  return MGCMT_OK;
}

int mgcmt_bandwidth_probe(mgcmt_plan* p, int l, int kind, int blocks, int reps, double* ms_out, void* stream) {
  MG_TRY(check_level(p, l));
  if (!ms_out || reps < 1 || blocks < 1 || kind < 0 || (kind > 17 && (kind < 20 || kind > 23))) return fail(MGCMT_ERR_INVALID, "bad arguments");
  MG_TRY(ensure_slot(p, l, MGCMT_SLOT_V));
  MG_TRY(ensure_slot(p, l, MGCMT_SLOT_F));
  MG_TRY(ensure_slot(p, l, MGCMT_SLOT_T));
  const long n = p->interior(l) & ~1L;
  hipEvent_t a, b;
  MG_HIP(hipEventCreate(&a));
  MG_HIP(hipEventCreate(&b));
  auto go = [&]() {
    if (kind >= 20) launch_probe_issue(S(stream), kind - 20, 20000, blocks, p->kvec(l, MGCMT_SLOT_T).p);  // 64 x 20000 instructions per wave
    else if (kind <= 2) launch_probe(S(stream), kind, n, p->kvec(l, MGCMT_SLOT_V).p, p->kvec(l, MGCMT_SLOT_F).p, p->kvec(l, MGCMT_SLOT_T).p, blocks);
    else  // marching pattern: `blocks` = rows per chunk
      // (kind - 3) % 3: one read stream / two read streams / two reads + one write; (kind - 3) / 3: columns a wave
      // writes out of the 128 it reads: 128 (no overlap), 124 (unaligned), 112 (the fused kernels' geometry), 96
      launch_probe_march(S(stream), p->levels[l].nr, p->levels[l].gc, blocks, (kind - 3) % 3 == 0 ? 1 : 2, (kind - 3) % 3 == 2 ? 1 : 0,
                         (kind - 3) / 3 == 0 ? 128 : ((kind - 3) / 3 == 1 ? 124 : ((kind - 3) / 3 == 2 ? 112 : ((kind - 3) / 3 == 3 ? 96 : 120))), p->kvec(l, MGCMT_SLOT_V).p,
                         p->kvec(l, MGCMT_SLOT_F).p, p->kvec(l, MGCMT_SLOT_T).p);
  };
  go();
  MG_HIP(hipEventRecord(a, S(stream)));
  for (int r = 0; r < reps; ++r) go();
  MG_HIP(hipEventRecord(b, S(stream)));
  MG_HIP(hipEventSynchronize(b));
  float ms = 0.f;
  MG_HIP(hipEventElapsedTime(&ms, a, b));
  *ms_out = (double)ms / reps;
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  return post_launch();
}

}  // extern "C"
