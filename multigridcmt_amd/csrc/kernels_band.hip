// Direct solve on the coarsest level: banded LU with partial pivoting, one workgroup per vector.
//
// Replaces scipy's spsolve(shifted_matrix, f) at grid_dimension == lowest_level
// (MGCMTSolver.py:305-308, :406-409) and twogrid's coarse solve (:362).  The matrix
// (A_coarsest - mu I) has n = rows*cols unknowns (2 ... a few thousand) and half bandwidth
// kl = cols + 1 (9-point rows) or 1 (1-D).  It may be indefinite (mu sits next to an eigenvalue
// in the shift-and-invert drivers), hence row pivoting as in SuperLU.
//
// Storage: row r keeps columns r-kl .. r+2*kl (width 3*kl+1; the extra kl columns take the fill of
// the row interchanges) at ab[r*width + (c - r + kl)].
#include "mgcmt_internal.h"

namespace mgcmt {

namespace {

constexpr int kBandThreads = 256;
constexpr int kMaxKl = 130;

__device__ __forceinline__ double& band_at(double* ab, int width, int kl, long r, long c) { return ab[r * width + (c - r + kl)]; }

__global__ void k_band_assemble(KGrid g, KOp op, const double* __restrict__ shifts, KBand b) {
  const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= b.n) return;
  const int q = blockIdx.y;
  double* ab = b.ab + q * b.ab_stride;
  const long nc = g.nc;
  const long i = r / nc, j = r % nc;
  const double mu = shifts[q];
  for (int w = 0; w < b.width; ++w) ab[r * b.width + w] = 0.0;
  for (int di = -1; di <= 1; ++di) {
    const long ii = i + di;
    if (ii < 0 || ii >= g.nr) continue;
    for (int dj = -1; dj <= 1; ++dj) {
      const long jj = j + dj;
      if (jj < 0 || jj >= nc) continue;
      double c = 0.0;
      if (op.five_point) {
        if (di == 0 && dj == 0) c = op.c0;
        else if (di == 0) c = op.cw;
        else if (dj == 0) c = op.cn;
      } else {
        for (int m = 0; m < op.nterms; ++m) {
          const double x = op.X[m][(di + 1) * op.ldx + i];
          const double y = op.Y[m][(dj + 1) * op.ldy + j];
          c += x * y;
        }
      }
      if (di == 0 && dj == 0) c -= mu;
      band_at(ab, b.width, b.kl, r, ii * nc + jj) = c;
    }
  }
}

__global__ void __launch_bounds__(kBandThreads) k_band_factor(KBand b) {
  __shared__ double s_val[kBandThreads];
  __shared__ int s_idx[kBandThreads];
  __shared__ double s_mult[kMaxKl + 2];
  const int q = blockIdx.x;
  double* ab = b.ab + q * b.ab_stride;
  int* piv = b.piv + q * b.piv_stride;
  const int tid = threadIdx.x, nt = blockDim.x;
  const int kl = b.kl, width = b.width;
  const long n = b.n;
  for (long j = 0; j < n; ++j) {
    const int km = (int)((n - 1 - j) < kl ? (n - 1 - j) : kl);
    const long clast = (j + 2L * kl) < (n - 1) ? (j + 2L * kl) : (n - 1);
    // pivot: largest |a_rj| over rows j..j+km (lowest row index wins ties)
    double best = -1.0;
    int besti = 0;
    for (int t = tid; t <= km; t += nt) {
      const double a = fabs(band_at(ab, width, kl, j + t, j));
      if (a > best) {
        best = a;
        besti = t;
      }
    }
    s_val[tid] = best;
    s_idx[tid] = besti;
    __syncthreads();
    for (int s = nt >> 1; s > 0; s >>= 1) {
      if (tid < s) {
        const double o = s_val[tid + s];
        const int oi = s_idx[tid + s];
        if (o > s_val[tid] || (o == s_val[tid] && oi < s_idx[tid])) {
          s_val[tid] = o;
          s_idx[tid] = oi;
        }
      }
      __syncthreads();
    }
    const long p = j + s_idx[0];
    if (tid == 0) piv[j] = (int)p;
    if (p != j) {
      for (long c = j + tid; c <= clast; c += nt) {
        const double a = band_at(ab, width, kl, j, c);
        band_at(ab, width, kl, j, c) = band_at(ab, width, kl, p, c);
        band_at(ab, width, kl, p, c) = a;
      }
    }
    __syncthreads();
    const double pivot = band_at(ab, width, kl, j, j);
    for (int t = tid; t < km; t += nt) {
      const double m = band_at(ab, width, kl, j + 1 + t, j) / pivot;
      s_mult[t] = m;
      band_at(ab, width, kl, j + 1 + t, j) = m;
    }
    __syncthreads();
    const long ncols = clast - j;  // columns j+1..clast
    const long work = (long)km * ncols;
    for (long w = tid; w < work; w += nt) {
      const int t = (int)(w / ncols);
      const long c = j + 1 + (w % ncols);
      band_at(ab, width, kl, j + 1 + t, c) -= s_mult[t] * band_at(ab, width, kl, j, c);
    }
    __syncthreads();
  }
}

__global__ void __launch_bounds__(kBandThreads) k_band_solve(KBand b, KVec rhs, KVec xx) {
  const int q = blockIdx.x;
  const double* ab = b.ab + q * b.ab_stride;
  const int* piv = b.piv + q * b.piv_stride;
  const double* f = rhs.p + q * rhs.stride;
  double* x = xx.p + q * xx.stride;
  const int tid = threadIdx.x, nt = blockDim.x;
  const int kl = b.kl, width = b.width;
  const long n = b.n;
  for (long r = tid; r < n; r += nt) x[r] = f[r];
  __syncthreads();
  // L y = P b, interchanges interleaved with the elimination as they were in the factorisation
  for (long j = 0; j < n; ++j) {
    const int km = (int)((n - 1 - j) < kl ? (n - 1 - j) : kl);
    if (tid == 0) {
      const long p = piv[j];
      if (p != j) {
        const double a = x[j];
        x[j] = x[p];
        x[p] = a;
      }
    }
    __syncthreads();
    const double xj = x[j];
    for (int t = tid; t < km; t += nt) x[j + 1 + t] -= ab[(j + 1 + t) * width + (j - (j + 1 + t) + kl)] * xj;
    __syncthreads();
  }
  // U x = y
  for (long j = n - 1; j >= 0; --j) {
    if (tid == 0) x[j] = x[j] / ab[j * width + kl];
    __syncthreads();
    const double xj = x[j];
    const long first = (j - 2L * kl) > 0 ? (j - 2L * kl) : 0;
    for (long r = first + tid; r < j; r += nt) x[r] -= ab[r * width + (j - r + kl)] * xj;
    __syncthreads();
  }
}

// Explicit inverse of the factored coarsest-level matrix (small: lowest_level^2 unknowns): thread j runs the
// whole forward / backward substitution for the unit vector e_j on its own column, stored transposed
// (inv[i*n + j]) so that the accesses of neighbouring threads coalesce.  One solve per cycle then is a dense
// matrix-vector product instead of 2n barrier-separated substitution steps.
__global__ void k_band_invert(KBand b, double* __restrict__ inv_all, long inv_stride) {
  const long j = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int q = blockIdx.y;
  const long n = b.n;
  if (j >= n) return;
  const double* ab = b.ab + q * b.ab_stride;
  const int* piv = b.piv + q * b.piv_stride;
  double* x = inv_all + q * inv_stride + j;  // element i at x[i*n]
  const int kl = b.kl, width = b.width;
  for (long i = 0; i < n; ++i) x[i * n] = i == j ? 1.0 : 0.0;
  for (long c = 0; c < n; ++c) {
    const long p = piv[c];
    if (p != c) {
      const double t = x[c * n];
      x[c * n] = x[p * n];
      x[p * n] = t;
    }
    const double xc = x[c * n];
    if (xc != 0.0) {
      const long last = c + kl < n - 1 ? c + kl : n - 1;
      for (long r = c + 1; r <= last; ++r) x[r * n] -= ab[r * width + (c - r + kl)] * xc;
    }
  }
  for (long c = n - 1; c >= 0; --c) {
    const double xc = x[c * n] / ab[c * width + kl];
    x[c * n] = xc;
    const long first = c - 2L * kl > 0 ? c - 2L * kl : 0;
    for (long r = first; r < c; ++r) x[r * n] -= ab[r * width + (c - r + kl)] * xc;
  }
}

// x = inv * f with inv stored transposed (inv[i*n + j] = (A^-1)[i][j]... element (row i, column j) at [i*n + j])
__global__ void __launch_bounds__(kBandThreads) k_dense_solve(long n, const double* __restrict__ inv_all, long inv_stride, KVec rhs, KVec xx) {
  __shared__ double s_f[1024];
  const int q = blockIdx.y;
  const double* inv = inv_all + q * inv_stride;
  const double* f = rhs.p + q * rhs.stride;
  double* x = xx.p + q * xx.stride;
  for (long c = threadIdx.x; c < n; c += blockDim.x) s_f[c] = f[c];
  __syncthreads();
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // row i of the inverse is contiguous: inv[i*n + j]
  const double* row = inv + i * n;
  double acc = 0.0;
  for (long j = 0; j < n; ++j) acc += row[j] * s_f[j];
  x[i] = acc;
}

}  // namespace

void launch_band_invert(hipStream_t s, KBand b, double* inv, long inv_stride, int k) {
  hipLaunchKernelGGL(k_band_invert, dim3((unsigned)((b.n + 63) / 64), (unsigned)k), dim3(64), 0, s, b, inv, inv_stride);
}

void launch_dense_solve(hipStream_t s, long n, const double* inv, long inv_stride, KVec rhs, KVec x, int k) {
  hipLaunchKernelGGL(k_dense_solve, dim3((unsigned)((n + kBandThreads - 1) / kBandThreads), (unsigned)k), dim3(kBandThreads), 0, s, n, inv, inv_stride, rhs, x);
}

void launch_band_assemble(hipStream_t s, KGrid g, KOp op, const double* shifts, KBand b, int k) {
  hipLaunchKernelGGL(k_band_assemble, dim3((unsigned)((b.n + 255) / 256), (unsigned)k), dim3(256), 0, s, g, op, shifts, b);
}

void launch_band_factor(hipStream_t s, KBand b, int k) { hipLaunchKernelGGL(k_band_factor, dim3(k), dim3(kBandThreads), 0, s, b); }

void launch_band_solve(hipStream_t s, KBand b, KVec rhs, KVec x, int k) {
  hipLaunchKernelGGL(k_band_solve, dim3(k), dim3(kBandThreads), 0, s, b, rhs, x);
}

}  // namespace mgcmt
