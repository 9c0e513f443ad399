// General sparse, complex128 operators on the device — the k.p Hamiltonians of the reference's ThesisProblem.py:38-40,
// 80,101 / PotWellSolver.py:54-233 (a 4n x 4n or 6n x 6n complex block matrix of tridiagonal blocks, cycled as ONE 1-D
// grid of length 4n with smoother=solver.gseidel and lowest_level=2**5).  SURVEY §8 (f)2.
//
// What runs here for such an operator, all on the GPU:
//   * the hierarchy: Galerkin products R*A*P (MGCMTSolver.py:318) with the 1-D full-weighting R and linear P of
//     MGCMTStencilMaker.py:27-78, built level by level from the CSR matrix by a count pass and a fill pass (one thread
//     per coarse row, its few candidate entries merged in registers);
//   * A x / residual, weighted Jacobi (:182-208), lexicographic Gauss-Seidel / SOR (:210-246, incl. the (D-L)^-1 quirk of
//     :241 through the same generalised sweep as kernels_lex.hip), 1-D restriction / interpolation + correction,
//     and the direct solve of the coarsest level (dense LU with row pivoting in LDS, at most 64 unknowns);
//   * the V-cycle (:281-329, V(4,4) below the top level: :320).
// Lexicographic sweeps: rows are processed in chunks of C consecutive rows by one workgroup.  C is the largest power of
// two (<= 1024) for which every row's strictly-lower entries other than k-1 lie BEFORE the row's chunk — true for the
// block matrices above with C = block size — so that inside a chunk the sweep is the first-order recurrence
// x_k = p_k + q_k x_{k-1}, solved by a scan over complex affine maps; matrices without that structure get C = 1, the
// plain sequential sweep.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "plan_internal.h"

using namespace mgcmt;

namespace {

typedef double2 cplx;
__host__ __device__ __forceinline__ cplx cmake(double re, double im) { return make_double2(re, im); }
__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return cmake(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { return cmake(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ cplx cmul(cplx a, cplx b) { return cmake(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ cplx cscale(double s, cplx a) { return cmake(s * a.x, s * a.y); }
__device__ __forceinline__ cplx cdiv(cplx a, cplx b) {
  const double den = b.x * b.x + b.y * b.y;
  return cmake((a.x * b.x + a.y * b.y) / den, (a.y * b.x - a.x * b.y) / den);
}
__device__ __forceinline__ double cabs2(cplx a) { return a.x * a.x + a.y * a.y; }

struct CsrLevel {
  long n = 0, nnz = 0;
  long* indptr = nullptr;  // [n + 1]
  int* indices = nullptr;  // [nnz], sorted inside a row
  cplx* vals = nullptr;    // [nnz]
  cplx* vec[3] = {nullptr, nullptr, nullptr};  // V, F, T
  int chunk = 1;           // rows per chunk of the lexicographic sweep
};

struct KCsr {
  long n;
  const long* indptr;
  const int* indices;
  const cplx* vals;
};

// ---- kernels -------------------------------------------------------------------------------------------------------

// out = mode 0: (A - mu I) x ; mode 1: f - (A - mu I) x ; mode 2: x + omega (f - (A - mu I) x) / (a_kk - mu)
__global__ void k_csr_apply(KCsr A, const cplx* __restrict__ x, const cplx* __restrict__ f, cplx* __restrict__ out, double mu, double omega,
                            int mode) {
  const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= A.n) return;
  cplx acc = cmake(0.0, 0.0), diag = cmake(0.0, 0.0);
  for (long e = A.indptr[k]; e < A.indptr[k + 1]; ++e) {
    const int j = A.indices[e];
    const cplx a = A.vals[e];
    if (j == k) diag = a;
    acc = cadd(acc, cmul(a, x[j]));
  }
  acc = csub(acc, cscale(mu, x[k]));
  if (mode == 0) {
    out[k] = acc;
  } else if (mode == 1) {
    out[k] = csub(f[k], acc);
  } else {
    const cplx d = cmake(diag.x - mu, diag.y);
    out[k] = cadd(x[k], cscale(omega, cdiv(csub(f[k], acc), d)));
  }
}

// coarse_I = 1/4 r_{2I} + 1/2 r_{2I+1} + 1/4 r_{2I+2}  (zero beyond the end; MGCMTStencilMaker.py:57-78)
__global__ void k_csr_restrict(long n, const cplx* __restrict__ r, cplx* __restrict__ rc) {
  const long I = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (I >= n / 2) return;
  const cplx a = r[2 * I], b = r[2 * I + 1];
  const cplx c = 2 * I + 2 < n ? r[2 * I + 2] : cmake(0.0, 0.0);
  rc[I] = cmake(0.25 * a.x + 0.5 * b.x + 0.25 * c.x, 0.25 * a.y + 0.5 * b.y + 0.25 * c.y);
}

// v_k += (P e)_k: odd k takes e_{(k-1)/2}, even k takes (e_{k/2-1} + e_{k/2}) / 2 with e_{-1} = 0  (:27-54)
__global__ void k_csr_prolong_add(long n, const cplx* __restrict__ e, cplx* __restrict__ v) {
  const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  cplx add;
  if (k & 1) {
    add = e[(k - 1) / 2];
  } else {
    const cplx hi = e[k / 2];
    const cplx lo = k >= 2 ? e[k / 2 - 1] : cmake(0.0, 0.0);
    add = cmake(0.5 * (lo.x + hi.x), 0.5 * (lo.y + hi.y));
  }
  v[k] = cadd(v[k], add);
}

__global__ void k_csr_fill(long n, cplx* v, double re, double im) {
  const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) v[k] = cmake(re, im);
}

__global__ void k_csr_axpy(long n, double alpha, const cplx* __restrict__ x, cplx* __restrict__ y) {
  const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) y[k] = cadd(y[k], cscale(alpha, x[k]));
}

// Generalised lexicographic sweep, in place, one workgroup:
//   x_k <- (alpha d_k x_k + beta f_k - wU sum_{j>k} a_kj x_j - wL sum_{j<k} a_kj x_j^new) / d_k,   d_k = a_kk - mu
struct CAffine {
  cplx p, q;  // x -> p + q x
};
__device__ __forceinline__ CAffine ccompose(CAffine first, CAffine second) {  // `first` applied before `second`
  CAffine r;
  r.p = cadd(second.p, cmul(second.q, first.p));
  r.q = cmul(second.q, first.q);
  return r;
}
__device__ __forceinline__ cplx cshfl_up(cplx v, int d) { return cmake(__shfl_up(v.x, d), __shfl_up(v.y, d)); }

constexpr int kCsrLexThreads = 1024;

__global__ void __launch_bounds__(kCsrLexThreads) k_csr_lex(KCsr A, cplx* __restrict__ x, const cplx* __restrict__ f, double mu, double alpha,
                                                            double beta, double wU, double wL, int chunk) {
  __shared__ double s_pr[kCsrLexThreads / 64], s_pi[kCsrLexThreads / 64], s_qr[kCsrLexThreads / 64], s_qi[kCsrLexThreads / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (long c0 = 0; c0 < A.n; c0 += chunk) {
    const long k = c0 + tid;
    const bool active = tid < chunk && k < A.n;
    CAffine mine;
    mine.p = cmake(0.0, 0.0);
    mine.q = cmake(0.0, 0.0);
    if (active) {
      cplx lower = cmake(0.0, 0.0), upper = cmake(0.0, 0.0), d = cmake(-mu, 0.0), sub = cmake(0.0, 0.0);
      for (long e = A.indptr[k]; e < A.indptr[k + 1]; ++e) {
        const int j = A.indices[e];
        const cplx a = A.vals[e];
        if (j == k) d = cmake(a.x - mu, a.y);
        else if (j > k) upper = cadd(upper, cmul(a, x[j]));
        else if (j == k - 1 && j >= c0) sub = a;                 // the in-chunk neighbour: the recurrence
        else lower = cadd(lower, cmul(a, x[j]));                 // earlier chunks: already new
      }
      const cplx num = csub(csub(cadd(cscale(alpha, cmul(d, x[k])), cscale(beta, f[k])), cscale(wU, upper)), cscale(wL, lower));
      mine.p = cdiv(num, d);
      mine.q = cdiv(cscale(-wL, sub), d);
    }
    __syncthreads();  // every x of this chunk has been read
    // inclusive scan of the maps over the chunk: inside a wave, then across the waves in front
    CAffine inc = mine;
#pragma unroll
    for (int dd = 1; dd < 64; dd <<= 1) {
      CAffine prev;
      prev.p = cshfl_up(inc.p, dd);
      prev.q = cshfl_up(inc.q, dd);
      if (lane >= dd) inc = ccompose(prev, inc);
    }
    if (lane == 63) {
      s_pr[wave] = inc.p.x;
      s_pi[wave] = inc.p.y;
      s_qr[wave] = inc.q.x;
      s_qi[wave] = inc.q.y;
    }
    __syncthreads();
    CAffine before;
    before.p = cmake(0.0, 0.0);
    before.q = cmake(1.0, 0.0);
    for (int w = 0; w < wave; ++w) {
      CAffine a;
      a.p = cmake(s_pr[w], s_pi[w]);
      a.q = cmake(s_qr[w], s_qi[w]);
      before = ccompose(before, a);
    }
    const CAffine total = ccompose(before, inc);
    if (active) x[k] = total.p;  // (the value in front of the chunk's first row enters through `lower`: its q is zero)
    __syncthreads();              // the chunk's new values are visible to the next chunk
  }
}

// ---- Galerkin product R A P, one thread per coarse row -------------------------------------------------------------
constexpr int kRapMax = 96;  // entries a coarse row may have

template <bool FILL>
__global__ void k_csr_rap(KCsr A, long nc, long* __restrict__ counts, const long* __restrict__ cptr, int* __restrict__ cidx,
                          cplx* __restrict__ cval, int* __restrict__ overflow) {
  const long I = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (I >= nc) return;
  int cols[kRapMax];
  cplx vals[kRapMax];
  int m = 0;
  auto add = [&](int J, cplx v) {
    if (J < 0 || J >= nc) return;
    for (int t = 0; t < m; ++t)
      if (cols[t] == J) {
        vals[t] = cadd(vals[t], v);
        return;
      }
    if (m < kRapMax) {
      cols[m] = J;
      vals[m] = v;
      ++m;
    } else {
      *overflow = 1;
    }
  };
  const double rw[3] = {0.25, 0.5, 0.25};
  for (int t = 0; t < 3; ++t) {
    const long a = 2 * I + t;
    if (a >= A.n) continue;
    for (long e = A.indptr[a]; e < A.indptr[a + 1]; ++e) {
      const int b = A.indices[e];
      const cplx v = cscale(rw[t], A.vals[e]);
      if (b & 1) {
        add((b - 1) / 2, v);                 // P[2J+1][J] = 1
      } else {
        add(b / 2, cscale(0.5, v));          // P[2J][J] = 1/2
        add(b / 2 - 1, cscale(0.5, v));      // P[2J+2][J] = 1/2
      }
    }
  }
  if (!FILL) {
    counts[I] = m;
    return;
  }
  // sort by column (insertion sort: a handful of entries) and write the row
  for (int s = 1; s < m; ++s) {
    const int c = cols[s];
    const cplx v = vals[s];
    int t = s - 1;
    while (t >= 0 && cols[t] > c) {
      cols[t + 1] = cols[t];
      vals[t + 1] = vals[t];
      --t;
    }
    cols[t + 1] = c;
    vals[t + 1] = v;
  }
  const long base = cptr[I];
  for (int t = 0; t < m; ++t) {
    cidx[base + t] = cols[t];
    cval[base + t] = vals[t];
  }
}

// ---- coarsest level: dense LU with row pivoting in LDS, one workgroup ------------------------------------------------
constexpr int kDenseMax = 64;

__global__ void __launch_bounds__(256) k_csr_dense_solve(KCsr A, double mu, const cplx* __restrict__ f, cplx* __restrict__ x) {
  __shared__ double mr[kDenseMax][kDenseMax + 1], mi[kDenseMax][kDenseMax + 1];
  __shared__ double br[kDenseMax], bi[kDenseMax];
  __shared__ int s_piv;
  const int n = (int)A.n, tid = threadIdx.x;
  for (int t = tid; t < n * n; t += blockDim.x) {
    mr[t / n][t % n] = 0.0;
    mi[t / n][t % n] = 0.0;
  }
  __syncthreads();
  for (int r = tid; r < n; r += blockDim.x) {
    for (long e = A.indptr[r]; e < A.indptr[r + 1]; ++e) {
      mr[r][A.indices[e]] = A.vals[e].x;
      mi[r][A.indices[e]] = A.vals[e].y;
    }
    mr[r][r] -= mu;
    br[r] = f[r].x;
    bi[r] = f[r].y;
  }
  __syncthreads();
  for (int c = 0; c < n; ++c) {
    if (tid == 0) {
      int best = c;
      double bestv = mr[c][c] * mr[c][c] + mi[c][c] * mi[c][c];
      for (int r = c + 1; r < n; ++r) {
        const double a = mr[r][c] * mr[r][c] + mi[r][c] * mi[r][c];
        if (a > bestv) {
          bestv = a;
          best = r;
        }
      }
      s_piv = best;
    }
    __syncthreads();
    const int pr = s_piv;
    if (pr != c) {
      for (int j = tid; j < n; j += blockDim.x) {
        const double tr = mr[c][j], ti = mi[c][j];
        mr[c][j] = mr[pr][j];
        mi[c][j] = mi[pr][j];
        mr[pr][j] = tr;
        mi[pr][j] = ti;
      }
      if (tid == 0) {
        const double tr = br[c], ti = bi[c];
        br[c] = br[pr];
        bi[c] = bi[pr];
        br[pr] = tr;
        bi[pr] = ti;
      }
    }
    __syncthreads();
    const cplx piv = cmake(mr[c][c], mi[c][c]);
    for (int r = c + 1 + tid; r < n; r += blockDim.x) {
      const cplx l = cdiv(cmake(mr[r][c], mi[r][c]), piv);
      for (int j = c + 1; j < n; ++j) {
        const cplx u = cmul(l, cmake(mr[c][j], mi[c][j]));
        mr[r][j] -= u.x;
        mi[r][j] -= u.y;
      }
      const cplx ub = cmul(l, cmake(br[c], bi[c]));
      br[r] -= ub.x;
      bi[r] -= ub.y;
    }
    __syncthreads();
  }
  if (tid == 0) {  // back substitution (a few dozen unknowns)
    for (int r = n - 1; r >= 0; --r) {
      cplx acc = cmake(br[r], bi[r]);
      for (int j = r + 1; j < n; ++j) acc = csub(acc, cmul(cmake(mr[r][j], mi[r][j]), cmake(br[j], bi[j])));
      const cplx v = cdiv(acc, cmake(mr[r][r], mi[r][r]));
      br[r] = v.x;
      bi[r] = v.y;
    }
  }
  __syncthreads();
  for (int r = tid; r < n; r += blockDim.x) x[r] = cmake(br[r], bi[r]);
}

inline unsigned blocks_for(long n, int threads = 256) { return (unsigned)((n + threads - 1) / threads); }

}  // namespace

struct mgcmt_csr_plan {
  int device = 0;
  long lowest = 2;
  std::vector<CsrLevel> levels;
};

namespace {

KCsr kcsr(const CsrLevel& L) { return KCsr{L.n, L.indptr, L.indices, L.vals}; }

void free_level(CsrLevel& L) {
  if (L.indptr) (void)hipFree(L.indptr);
  if (L.indices) (void)hipFree(L.indices);
  if (L.vals) (void)hipFree(L.vals);
  for (cplx*& v : L.vec)
    if (v) (void)hipFree(v);
  L = CsrLevel{};
}

// largest power-of-two chunk for which every strictly-lower entry other than k-1 lies before the row's chunk
int lex_chunk(long n, const std::vector<long>& ptr, const std::vector<int>& idx) {
  for (int C = kCsrLexThreads; C > 1; C >>= 1) {
    bool ok = true;
    for (long k = 0; k < n && ok; ++k) {
      const long c0 = k - k % C;
      for (long e = ptr[k]; e < ptr[k + 1]; ++e)
        if (idx[e] < k - 1 && idx[e] >= c0) {
          ok = false;
          break;
        }
    }
    if (ok) return C;
  }
  return 1;
}

int alloc_vectors(CsrLevel& L) {
  for (cplx*& v : L.vec) {
    MG_HIP(hipMalloc((void**)&v, sizeof(cplx) * (size_t)L.n));
    MG_HIP(hipMemset(v, 0, sizeof(cplx) * (size_t)L.n));
  }
  return MGCMT_OK;
}

int check(const mgcmt_csr_plan* p, int level) {
  if (!p) return fail(MGCMT_ERR_INVALID, "null plan");
  if (level < 0 || level >= (int)p->levels.size()) return fail(MGCMT_ERR_INVALID, "level out of range");
  return MGCMT_OK;
}

int lex_sweep(mgcmt_csr_plan* p, int l, int slot, double mu, double alpha, double beta, double wU, double wL, hipStream_t s) {
  CsrLevel& L = p->levels[l];
  hipLaunchKernelGGL(k_csr_lex, dim3(1), dim3(kCsrLexThreads), 0, s, kcsr(L), L.vec[slot], L.vec[MGCMT_SLOT_F], mu, alpha, beta, wU, wL, L.chunk);
  return post_launch();
}

int smooth(mgcmt_csr_plan* p, int l, int kind, int nu, double omega, double mu, hipStream_t s) {
  CsrLevel& L = p->levels[l];
  if (kind == MGCMT_WJACOBI) {
    for (int it = 0; it < nu; ++it) {
      hipLaunchKernelGGL(k_csr_apply, dim3(blocks_for(L.n)), dim3(256), 0, s, kcsr(L), L.vec[MGCMT_SLOT_V], L.vec[MGCMT_SLOT_F], L.vec[MGCMT_SLOT_T], mu,
                         omega, 2);
      std::swap(L.vec[MGCMT_SLOT_V], L.vec[MGCMT_SLOT_T]);
    }
    return post_launch();
  }
  if (kind == MGCMT_GS_LEX || (kind == MGCMT_SOR_LEX && omega == 1.0)) {
    for (int it = 0; it < nu; ++it) MG_TRY(lex_sweep(p, l, MGCMT_SLOT_V, mu, 0.0, 1.0, 1.0, 1.0, s));
    return MGCMT_OK;
  }
  if (kind == MGCMT_SOR_LEX) {  // (D-wL)^-1((1-w)D + wU) v + w (D-L)^-1 f, MGCMTSolver.py:229-246 (cf. plan.hip)
    hipLaunchKernelGGL(k_csr_fill, dim3(blocks_for(L.n)), dim3(256), 0, s, L.n, L.vec[MGCMT_SLOT_T], 0.0, 0.0);
    MG_TRY(lex_sweep(p, l, MGCMT_SLOT_T, mu, 0.0, 1.0, 0.0, 1.0, s));
    for (int it = 0; it < nu; ++it) {
      MG_TRY(lex_sweep(p, l, MGCMT_SLOT_V, mu, 1.0 - omega, 0.0, omega, omega, s));
      hipLaunchKernelGGL(k_csr_axpy, dim3(blocks_for(L.n)), dim3(256), 0, s, L.n, omega, L.vec[MGCMT_SLOT_T], L.vec[MGCMT_SLOT_V]);
    }
    return post_launch();
  }
  return fail(MGCMT_ERR_UNSUPPORTED, "general sparse operators take wjacobi, gseidel and sor");
}

}  // namespace

extern "C" {

int mgcmt_csr_plan_create(int device, int64_t n, int64_t lowest, const int64_t* indptr, const int32_t* indices, const double* values,
                          mgcmt_csr_plan** out) {
  if (!indptr || !indices || !values || !out) return fail(MGCMT_ERR_INVALID, "null argument");
  *out = nullptr;
  if (lowest != n) {  // (a single level — smoothing and operator application only — may have any size)
    if (n < 2 || (n & (n - 1)) || lowest < 2 || (lowest & (lowest - 1)) || lowest > n)
      return fail(MGCMT_ERR_INVALID, "n and lowest must be powers of two with 2 <= lowest <= n");
    if (lowest > kDenseMax) return fail(MGCMT_ERR_UNSUPPORTED, "general sparse operators: the coarsest level may have at most 64 unknowns");
  } else if (n < 1) {
    return fail(MGCMT_ERR_INVALID, "empty matrix");
  }
  if (n > (1L << 30)) return fail(MGCMT_ERR_UNSUPPORTED, "matrix too large");
  MG_HIP(hipSetDevice(device));
  mgcmt_csr_plan* p = new mgcmt_csr_plan();
  p->device = device;
  p->lowest = lowest;
  int nlev = 1;
  for (int64_t s = n; s > lowest; s >>= 1) ++nlev;
  p->levels.resize(nlev);
  auto bail = [&](int rc) {
    mgcmt_csr_plan_destroy(p);
    return rc;
  };
  // level 0: the caller's matrix (rows sorted by column, as the chunk analysis and the kernels assume).  The row
  // pointer is checked before anything is sized by it: a caller of the C-ABI need not be scipy.
  if (indptr[0] != 0) return bail(fail(MGCMT_ERR_INVALID, "indptr[0] must be 0"));
  for (int64_t k = 0; k < n; ++k)
    if (indptr[k + 1] < indptr[k]) return bail(fail(MGCMT_ERR_INVALID, "indptr must be non-decreasing"));
  auto copied = [&](hipError_t e, const char* what) -> int {  // a failed copy must not leave a silently wrong hierarchy
    if (e == hipSuccess) return MGCMT_OK;
    return fail(MGCMT_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
  };
  std::vector<long> ptr(indptr, indptr + n + 1);
  std::vector<int> idx(indices, indices + ptr[n]);
  std::vector<cplx> val((size_t)ptr[n]);
  for (long e = 0; e < ptr[n]; ++e) val[e] = cmake(values[2 * e], values[2 * e + 1]);
  for (long k = 0; k < n; ++k) {
    std::vector<std::pair<int, long>> order;
    for (long e = ptr[k]; e < ptr[k + 1]; ++e) {
      if (idx[e] < 0 || idx[e] >= n) return bail(fail(MGCMT_ERR_INVALID, "column index out of range"));
      order.push_back({idx[e], e});
    }
    std::sort(order.begin(), order.end());
    std::vector<int> ci;
    std::vector<cplx> cv;
    for (auto& o : order) {
      ci.push_back(o.first);
      cv.push_back(val[o.second]);
    }
    std::copy(ci.begin(), ci.end(), idx.begin() + ptr[k]);
    std::copy(cv.begin(), cv.end(), val.begin() + ptr[k]);
  }
  for (int l = 0; l < nlev; ++l) {
    CsrLevel& L = p->levels[l];
    L.n = n >> l;
    if (l == 0) {
      L.nnz = ptr[n];
      if (hipMalloc((void**)&L.indptr, sizeof(long) * (L.n + 1)) != hipSuccess || hipMalloc((void**)&L.indices, sizeof(int) * std::max<long>(L.nnz, 1)) != hipSuccess ||
          hipMalloc((void**)&L.vals, sizeof(cplx) * std::max<long>(L.nnz, 1)) != hipSuccess)
        return bail(fail(MGCMT_ERR_NOMEM, "csr level"));
      int rc = copied(hipMemcpy(L.indptr, ptr.data(), sizeof(long) * (L.n + 1), hipMemcpyHostToDevice), "csr level 0: indptr");
      if (rc == MGCMT_OK) rc = copied(hipMemcpy(L.indices, idx.data(), sizeof(int) * L.nnz, hipMemcpyHostToDevice), "csr level 0: indices");
      if (rc == MGCMT_OK) rc = copied(hipMemcpy(L.vals, val.data(), sizeof(cplx) * L.nnz, hipMemcpyHostToDevice), "csr level 0: values");
      if (rc != MGCMT_OK) return bail(rc);
    } else {
      // Galerkin product of the finer level, on the device: count, prefix sum, fill
      const CsrLevel& F = p->levels[l - 1];
      long* d_counts = nullptr;
      int* d_over = nullptr;
      if (hipMalloc((void**)&d_counts, sizeof(long) * L.n) != hipSuccess || hipMalloc((void**)&d_over, sizeof(int)) != hipSuccess)
        return bail(fail(MGCMT_ERR_NOMEM, "csr level"));
      int rc = copied(hipMemset(d_over, 0, sizeof(int)), "Galerkin product: overflow flag");
      std::vector<long> counts(L.n);
      int over = 0;
      if (rc == MGCMT_OK) {
        hipLaunchKernelGGL(k_csr_rap<false>, dim3(blocks_for(L.n, 64)), dim3(64), 0, nullptr, kcsr(F), L.n, d_counts, (const long*)nullptr, (int*)nullptr,
                           (cplx*)nullptr, d_over);
        rc = copied(hipGetLastError(), "Galerkin product: count launch");
      }
      if (rc == MGCMT_OK) rc = copied(hipMemcpy(counts.data(), d_counts, sizeof(long) * L.n, hipMemcpyDeviceToHost), "Galerkin product: counts");
      if (rc == MGCMT_OK) rc = copied(hipMemcpy(&over, d_over, sizeof(int), hipMemcpyDeviceToHost), "Galerkin product: overflow flag");
      (void)hipFree(d_counts);
      if (rc != MGCMT_OK) {
        (void)hipFree(d_over);
        return bail(rc);
      }
      if (over) {
        (void)hipFree(d_over);
        return bail(fail(MGCMT_ERR_UNSUPPORTED, "a Galerkin row has more than 96 entries"));
      }
      std::vector<long> cptr(L.n + 1, 0);
      for (long i = 0; i < L.n; ++i) cptr[i + 1] = cptr[i] + counts[i];
      L.nnz = cptr[L.n];
      if (hipMalloc((void**)&L.indptr, sizeof(long) * (L.n + 1)) != hipSuccess || hipMalloc((void**)&L.indices, sizeof(int) * std::max<long>(L.nnz, 1)) != hipSuccess ||
          hipMalloc((void**)&L.vals, sizeof(cplx) * std::max<long>(L.nnz, 1)) != hipSuccess) {
        (void)hipFree(d_over);
        return bail(fail(MGCMT_ERR_NOMEM, "csr level"));
      }
      rc = copied(hipMemcpy(L.indptr, cptr.data(), sizeof(long) * (L.n + 1), hipMemcpyHostToDevice), "Galerkin product: indptr");
      if (rc != MGCMT_OK) {
        (void)hipFree(d_over);
        return bail(rc);
      }
      hipLaunchKernelGGL(k_csr_rap<true>, dim3(blocks_for(L.n, 64)), dim3(64), 0, nullptr, kcsr(F), L.n, (long*)nullptr, (const long*)L.indptr, L.indices,
                         L.vals, d_over);
      (void)hipFree(d_over);
      if (hipDeviceSynchronize() != hipSuccess) return bail(fail(MGCMT_ERR_HIP, "Galerkin product kernel failed"));
      ptr = cptr;
      idx.resize(L.nnz);
      rc = copied(hipMemcpy(idx.data(), L.indices, sizeof(int) * L.nnz, hipMemcpyDeviceToHost), "Galerkin product: indices");
      if (rc != MGCMT_OK) return bail(rc);
    }
    L.chunk = lex_chunk(L.n, ptr, idx);
    const int rc = alloc_vectors(L);
    if (rc != MGCMT_OK) return bail(rc);
  }
  *out = p;
  return MGCMT_OK;
}

int mgcmt_csr_plan_destroy(mgcmt_csr_plan* p) {
  if (!p) return MGCMT_OK;
  for (CsrLevel& L : p->levels) free_level(L);
  delete p;
  return MGCMT_OK;
}

int mgcmt_csr_num_levels(const mgcmt_csr_plan* p, int* levels) {
  if (!p || !levels) return fail(MGCMT_ERR_INVALID, "null argument");
  *levels = (int)p->levels.size();
  return MGCMT_OK;
}

int mgcmt_csr_level_info(const mgcmt_csr_plan* p, int level, int64_t* n, int64_t* nnz, int32_t* lex_chunk_rows) {
  MG_TRY(check(p, level));
  if (n) *n = p->levels[level].n;
  if (nnz) *nnz = p->levels[level].nnz;
  if (lex_chunk_rows) *lex_chunk_rows = p->levels[level].chunk;
  return MGCMT_OK;
}

int mgcmt_csr_get_matrix(const mgcmt_csr_plan* p, int level, int64_t* indptr, int32_t* indices, double* values) {
  MG_TRY(check(p, level));
  if (!indptr || !indices || !values) return fail(MGCMT_ERR_INVALID, "null argument");
  const CsrLevel& L = p->levels[level];
  std::vector<long> ptr(L.n + 1);
  MG_HIP(hipMemcpy(ptr.data(), L.indptr, sizeof(long) * (L.n + 1), hipMemcpyDeviceToHost));
  for (long i = 0; i <= L.n; ++i) indptr[i] = ptr[i];
  MG_HIP(hipMemcpy(indices, L.indices, sizeof(int) * L.nnz, hipMemcpyDeviceToHost));
  MG_HIP(hipMemcpy(values, L.vals, sizeof(cplx) * L.nnz, hipMemcpyDeviceToHost));
  return MGCMT_OK;
}

int mgcmt_csr_upload(mgcmt_csr_plan* p, int level, int slot, const double* re_im, int64_t count, void* stream) {
  MG_TRY(check(p, level));
  if (slot < 0 || slot > 2 || !re_im || count != p->levels[level].n) return fail(MGCMT_ERR_INVALID, "upload: bad slot or count");
  MG_HIP(hipMemcpyAsync(p->levels[level].vec[slot], re_im, sizeof(cplx) * count, hipMemcpyHostToDevice, (hipStream_t)stream));
  MG_HIP(hipStreamSynchronize((hipStream_t)stream));
  return MGCMT_OK;
}

int mgcmt_csr_download(mgcmt_csr_plan* p, int level, int slot, double* re_im, int64_t count, void* stream) {
  MG_TRY(check(p, level));
  if (slot < 0 || slot > 2 || !re_im || count != p->levels[level].n) return fail(MGCMT_ERR_INVALID, "download: bad slot or count");
  MG_HIP(hipMemcpyAsync(re_im, p->levels[level].vec[slot], sizeof(cplx) * count, hipMemcpyDeviceToHost, (hipStream_t)stream));
  MG_HIP(hipStreamSynchronize((hipStream_t)stream));
  return MGCMT_OK;
}

int mgcmt_csr_apply(mgcmt_csr_plan* p, int level, int src_slot, int dst_slot, double shift, void* stream) {
  MG_TRY(check(p, level));
  if (src_slot < 0 || src_slot > 2 || dst_slot < 0 || dst_slot > 2 || src_slot == dst_slot) return fail(MGCMT_ERR_INVALID, "apply: bad slots");
  CsrLevel& L = p->levels[level];
  hipLaunchKernelGGL(k_csr_apply, dim3(blocks_for(L.n)), dim3(256), 0, (hipStream_t)stream, kcsr(L), L.vec[src_slot], (const cplx*)nullptr, L.vec[dst_slot], shift,
                     0.0, 0);
  return post_launch();
}

int mgcmt_csr_smooth(mgcmt_csr_plan* p, int level, int kind, int nu, double omega, double shift, void* stream) {
  MG_TRY(check(p, level));
  if (nu < 0) return fail(MGCMT_ERR_INVALID, "nu must be >= 0");
  return smooth(p, level, kind, nu, omega, shift, (hipStream_t)stream);
}

int mgcmt_csr_vcycle(mgcmt_csr_plan* p, int nu1, int nu2, int nu_coarse, int kind, double omega, double shift, void* stream) {
  if (!p) return fail(MGCMT_ERR_INVALID, "null plan");
  if (nu1 < 0 || nu2 < 0 || nu_coarse < 0) return fail(MGCMT_ERR_INVALID, "sweep counts must be >= 0");
  hipStream_t s = (hipStream_t)stream;
  const int last = (int)p->levels.size() - 1;
  for (int l = 0; l < last; ++l) {
    CsrLevel& L = p->levels[l];
    CsrLevel& C = p->levels[l + 1];
    if (l > 0) hipLaunchKernelGGL(k_csr_fill, dim3(blocks_for(L.n)), dim3(256), 0, s, L.n, L.vec[MGCMT_SLOT_V], 0.0, 0.0);  // e2h = 0 (:316)
    MG_TRY(smooth(p, l, kind, l == 0 ? nu1 : nu_coarse, omega, shift, s));
    hipLaunchKernelGGL(k_csr_apply, dim3(blocks_for(L.n)), dim3(256), 0, s, kcsr(L), L.vec[MGCMT_SLOT_V], L.vec[MGCMT_SLOT_F], L.vec[MGCMT_SLOT_T], shift, 0.0,
                       1);
    hipLaunchKernelGGL(k_csr_restrict, dim3(blocks_for(C.n)), dim3(256), 0, s, L.n, L.vec[MGCMT_SLOT_T], C.vec[MGCMT_SLOT_F]);
  }
  {
    CsrLevel& L = p->levels[last];
    if (L.n > kDenseMax) return fail(MGCMT_ERR_UNSUPPORTED, "general sparse operators: the coarsest level may have at most 64 unknowns");
    hipLaunchKernelGGL(k_csr_dense_solve, dim3(1), dim3(256), 0, s, kcsr(L), shift, L.vec[MGCMT_SLOT_F], L.vec[MGCMT_SLOT_V]);
  }
  for (int l = last - 1; l >= 0; --l) {
    CsrLevel& L = p->levels[l];
    hipLaunchKernelGGL(k_csr_prolong_add, dim3(blocks_for(L.n)), dim3(256), 0, s, L.n, p->levels[l + 1].vec[MGCMT_SLOT_V], L.vec[MGCMT_SLOT_V]);
    MG_TRY(smooth(p, l, kind, l == 0 ? nu2 : nu_coarse, omega, shift, s));
  }
  return post_launch();
}

}  // extern "C"
