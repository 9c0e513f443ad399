// Lexicographic (index-order) Gauss-Seidel / SOR sweep — the PARITY-MODE smoother.
//
// The reference's gseidel / sor (MGCMTSolver.py:210-246) are forward sweeps in the vector's index
// order k = i*cols + j.  Row i depends on the new row i-1 and, inside the row, v_j depends on the
// new v_{j-1}: with everything else known that is the first-order recurrence
//         v_j = p_j + q_j v_{j-1},    v_{-1} = 0,
// which one workgroup solves for a whole row with a scan over the affine maps x -> p + q x
// (wave64 shuffles inside a wave, LDS across waves).  Rows are processed in order, so the sweep is
// sequential in i and parallel in j; it exists to reproduce the reference's numbers, not to be fast
// (the performance smoothers are the order-independent ones).
//
// The update is the generalised form
//     v_k <- (alpha d_k v_k + beta f_k - wU sum_{j>k} a_kj v_j - wL sum_{j<k} a_kj v_j^new) / d_k
// Gauss-Seidel: alpha 0, beta 1, wU = wL = 1.  The reference's SOR, including its (D-L)^-1 f term
// (MGCMTSolver.py:241), is two instances of it plus an axpy (see plan.hip, smooth_sor_lex).
#include "mgcmt_internal.h"

namespace mgcmt {

namespace {

constexpr int kLexThreads = 1024;
constexpr int kLexChunk = 4;  // columns per thread and tile

struct Affine {
  double p, q;  // x -> p + q x
};

// `first` is applied before `second`
__device__ __forceinline__ Affine compose(Affine first, Affine second) {
  Affine r;
  r.p = second.p + second.q * first.p;
  r.q = second.q * first.q;
  return r;
}

__global__ void __launch_bounds__(kLexThreads) k_lex_sweep(KGrid g, KOp op, KVec vv, KVec ff, const double* __restrict__ shifts,
                                                           double alpha, double beta, double wU, double wL) {
  __shared__ double s_p[kLexThreads / 64];
  __shared__ double s_q[kLexThreads / 64];
  __shared__ double s_carry;

  const int q = blockIdx.x;
  const double mu = shifts[q];
  double* v = vv.p + q * vv.stride;
  const double* f = ff.p + q * ff.stride;
  const long nc = g.nc;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int nwaves = blockDim.x >> 6;
  const long tile_cols = (long)blockDim.x * kLexChunk;

  for (long i = 0; i < g.nr; ++i) {
    if (tid == 0) s_carry = 0.0;  // v[i][-1] = 0
    __syncthreads();
    for (long t0 = 0; t0 < nc; t0 += tile_cols) {
      const long j0 = t0 + (long)tid * kLexChunk;
      double pj[kLexChunk], qj[kLexChunk];
      Affine mine = {0.0, 1.0};
#pragma unroll
      for (int c = 0; c < kLexChunk; ++c) {
        const long j = j0 + c;
        pj[c] = 0.0;
        qj[c] = 0.0;
        if (j < nc) {
          const double* ctr = v + i * nc + j;
          const bool hw = j > 0, he = j + 1 < nc;
          double lower, upper, cw, d;
          if (op.five_point) {
            lower = op.cn != 0.0 ? op.cn * ctr[-nc] : 0.0;
            upper = (he ? op.cw * ctr[1] : 0.0) + (op.cn != 0.0 ? op.cn * ctr[nc] : 0.0);
            cw = hw ? op.cw : 0.0;
            d = op.c0 - mu;
          } else {
            const double n = ctr[-nc], s = ctr[nc];
            const double e = he ? ctr[1] : 0.0;
            const double nw = hw ? ctr[-nc - 1] : 0.0, ne = he ? ctr[-nc + 1] : 0.0;
            const double sw = hw ? ctr[nc - 1] : 0.0, se = he ? ctr[nc + 1] : 0.0;
            lower = 0.0;
            upper = 0.0;
            cw = 0.0;
            d = 0.0;
            for (int m = 0; m < op.nterms; ++m) {
              const double* X = op.X[m] + i;
              const double* Y = op.Y[m] + j;
              const double xl = X[0], xd = X[op.ldx], xu = X[2 * op.ldx];
              const double yl = Y[0], yd = Y[op.ldy], yu = Y[2 * op.ldy];
              lower += xl * (yl * nw + yd * n + yu * ne);
              upper += xd * (yu * e) + xu * (yl * sw + yd * s + yu * se);
              cw += xd * yl;
              d += xd * yd;
            }
            if (!hw) cw = 0.0;
            d -= mu;
          }
          pj[c] = (alpha * d * ctr[0] + beta * f[i * nc + j] - wU * upper - wL * lower) / d;
          qj[c] = -wL * cw / d;
          Affine a = {pj[c], qj[c]};
          mine = compose(mine, a);
        }
      }
      // inclusive scan of the per-thread maps: inside the wave ...
      Affine inc = mine;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        Affine prev;
        prev.p = __shfl_up(inc.p, d);
        prev.q = __shfl_up(inc.q, d);
        if (lane >= d) inc = compose(prev, inc);
      }
      if (lane == 63) {
        s_p[wave] = inc.p;
        s_q[wave] = inc.q;
      }
      __syncthreads();
      // ... then across the waves in front of this one
      Affine before = {0.0, 1.0};
      for (int w = 0; w < wave; ++w) {
        Affine a = {s_p[w], s_q[w]};
        before = compose(before, a);
      }
      Affine excl_in_wave;  // composition of the lanes in front of this one, inside the wave
      excl_in_wave.p = __shfl_up(inc.p, 1);
      excl_in_wave.q = __shfl_up(inc.q, 1);
      if (lane == 0) {
        excl_in_wave.p = 0.0;
        excl_in_wave.q = 1.0;
      }
      const Affine upto = compose(before, excl_in_wave);
      const double carry = s_carry;
      double x = upto.p + upto.q * carry;  // value of the column left of this thread's chunk
#pragma unroll
      for (int c = 0; c < kLexChunk; ++c) {
        const long j = j0 + c;
        if (j < nc) {
          x = pj[c] + qj[c] * x;
          v[i * nc + j] = x;
        }
      }
      __syncthreads();  // every thread has read s_carry and the wave totals
      if (tid == blockDim.x - 1) {
        // last thread of the tile: its running x is the tile's last column (or the last valid one)
        s_carry = x;
      }
      if (nwaves > 0) __syncthreads();  // stores of this tile/row are visible to the next one
    }
  }
}

}  // namespace

void launch_lex_sweep(hipStream_t s, KGrid g, KOp op, KVec v, KVec f, const double* shifts, double alpha, double beta,
                      double wU, double wL, int k) {
  long need = (g.nc + kLexChunk - 1) / kLexChunk;
  int threads = 64;
  while (threads < need && threads < kLexThreads) threads <<= 1;
  hipLaunchKernelGGL(k_lex_sweep, dim3(k), dim3(threads), 0, s, g, op, v, f, shifts, alpha, beta, wU, wL);
}

}  // namespace mgcmt
