// Wide-window variant of the fused row-streaming pass (fused_kernel.h) for the passes that move the most bytes: the
// constant 5-point operator with weighted Jacobi on large levels.  A wave owns 256 columns with the same 8-column
// overlap, so 240 of 256 loaded columns are kept instead of 112 of 128: the redundant reads of the overlap (about 12 %
// of the traffic of the narrow kernels, which run at the streaming ceiling) are halved.  The window is two 128-column
// halves side by side; lane l owns columns 2l, 2l+1 of EACH half, so every load / store instruction is the same fully
// coalesced 1 KiB as in the narrow kernel (four consecutive columns per lane were measured first: 32-byte lane
// strides made the pass 10 % slower than the narrow one).  At the seam lane 63's right neighbour is lane 0's other
// half: one select and a wave ROTATE instead of a wave shift.  Same stages, flags, arithmetic (bit-identical
// results) and launch geometry rules as the narrow kernel.
#pragma once
#include "fused_kernel.h"

namespace mgcmt {

namespace fused {

#ifndef MGCMT_FUSED_WIDE_DEPTH
#define MGCMT_FUSED_WIDE_DEPTH 1
#endif
constexpr int kWideDepth = MGCMT_FUSED_WIDE_DEPTH;  // rows per prefetch batch (1 or 3)

template <int NSWEEP, int FLAGS>
struct WideShape {
  static constexpr bool PROLONG = (FLAGS & kProlong) != 0, RESTRICT = (FLAGS & kRestrict) != 0;
  static constexpr int NPRE = (FLAGS >> kPreShift) & 3;
  static constexpr int S = NPRE + NSWEEP;
  static constexpr int E = RESTRICT ? 1 : 0;
  static constexpr int halo = 8, cols = 256, wout = cols - 2 * halo;  // 240 columns = 15 whole cache lines stored
  static_assert(S + (RESTRICT ? 2 : (PROLONG ? 1 : 0)) <= halo, "too many stages for the window overlap");
  static_assert(S + E + 1 <= 6, "the right-hand-side ring holds six rows");
};

template <int NSWEEP, int FLAGS>
__global__ void __launch_bounds__(64) k_fused_wide(FusedArgs a) {
  using Shape = WideShape<NSWEEP, FLAGS>;
  constexpr int S = Shape::S, E = Shape::E, HALO = Shape::halo, WOUT = Shape::wout, SPRE = Shape::NPRE;
  constexpr bool PROLONG = Shape::PROLONG, RESTRICT = Shape::RESTRICT;
  constexpr bool ZERO_IN = (FLAGS & kZeroIn) != 0, STORE_V = (FLAGS & kNoStore) == 0;
  constexpr int C = 4;  // columns per lane
  constexpr int D = kWideDepth, B = 6;
  static_assert(B % D == 0 && (B / D) % 2 == 0, "the loop body must hold an even number of prefetch batches");

  const int b = blockIdx.x;
  const int xcd = b & 7, seq = b >> 3;
  const int per_xcd = (a.n_col_groups + 7) >> 3;
  const int group = xcd * per_xcd + seq % per_xcd;
  const int chunk = seq / per_xcd;
  if (seq % per_xcd + xcd * per_xcd >= a.n_col_groups || chunk >= a.n_row_chunks) return;
  if (group >= a.n_col_groups) return;
  const int lane = threadIdx.x & 63;
  const int lane_up = (lane > 0 ? lane - 1 : 0) << 2, lane_dn = (lane < 63 ? lane + 1 : 63) << 2;
  (void)lane_up;
  (void)lane_dn;
  const int nc = (int)a.nc, nr = (int)a.nr, cnc = (int)a.cnc;
  const int row_lo = (int)a.row_lo, row_hi = (int)a.row_hi;
  if (group * WOUT >= nc) return;
  const int q = blockIdx.y;

  // x[0], x[1]: columns ja0, ja0+1 of the left half; x[2], x[3]: columns ja1 = ja0 + 128, ja1+1 of the right half
  const int lane_upr = ((lane + 63) & 63) << 2, lane_dnr = ((lane + 1) & 63) << 2;  // neighbours with wrap-around
  (void)lane_upr;
  (void)lane_dnr;
  const bool first_lane = lane == 0, last_lane = lane == 63;
  const int ja0 = group * WOUT - HALO + 2 * lane, ja1 = ja0 + 128;
  const bool in0 = ja0 >= 0 && ja0 < nc, in1 = ja1 >= 0 && ja1 < nc;
  const bool out0 = in0 && 2 * lane >= HALO, out1 = in1 && 2 * lane < 128 - HALO;
  const int jc0 = ja0 >> 1, jc1 = jc0 + 64;  // coarse columns of the two pairs
  const bool cin0 = jc0 >= 0 && jc0 < cnc, cin1 = jc1 >= 0 && jc1 < cnc;
  const double mask[2] = {in0 ? 1.0 : 0.0, in1 ? 1.0 : 0.0};
  const double cmsk[2] = {cin0 ? 1.0 : 0.0, cin1 ? 1.0 : 0.0};
  const double om[2] = {in0 ? a.omega : 0.0, in1 ? a.omega : 0.0};

  const double* __restrict__ vin = a.vin + q * a.vstride;
  const double* __restrict__ fin = a.f + q * a.vstride;
  double* __restrict__ vout = a.vout + q * a.vstride;
  const double* __restrict__ ec = PROLONG ? a.ec + q * a.cstride : nullptr;
  double* __restrict__ rc = RESTRICT ? a.rc + q * a.cstride : nullptr;

  const int r_begin = chunk * a.rows_per_chunk;
  const int r_end = r_begin + a.rows_per_chunk < nr ? r_begin + a.rows_per_chunk : nr;
  const int rstart = r_begin - (S + E);
  const int rstop = r_end + S + 2 * E;
  auto row_ok = [&](int row) { return row >= row_lo && row < row_hi; };

  const int ja0_ld = ja0 < 0 ? 0 : (ja0 > nc - 2 ? nc - 2 : ja0), ja1_ld = ja1 < 0 ? 0 : (ja1 > nc - 2 ? nc - 2 : ja1);
  const int jc0_ld = jc0 < 0 ? 0 : (jc0 > cnc - 1 ? cnc - 1 : jc0), jc1_ld = jc1 < 0 ? 0 : (jc1 > cnc - 1 ? cnc - 1 : jc1);
  const int crow_lo = (row_lo >> 1) - 1, crow_hi = (row_hi - 1) >> 1;

  const double d = a.c0 - a.shifts[q], invd = 1.0 / d, cn = a.cn, cw = a.cw;

  struct Row {
    double2 v0, v1, f0, f1, e;
  };
  Row setA[D], setB[D];
  int frow = rstart;
  auto fetch = [&](Row& r) __attribute__((always_inline)) {
    const int hi = row_hi - 1;
    const int t = frow < hi ? frow : hi;
    const int frl = t > row_lo ? t : row_lo;
    const long fbase = (long)frl * nc;
    if (ZERO_IN) {
      r.v0 = r.v1 = make_double2(0.0, 0.0);
    } else {
      r.v0 = load2(vin + fbase + ja0_ld);
      r.v1 = load2(vin + fbase + ja1_ld);
    }
    r.f0 = load2_stream(fin + fbase + ja0_ld);
    r.f1 = load2_stream(fin + fbase + ja1_ld);
    r.e = make_double2(0.0, 0.0);
    if (PROLONG) {
      const int I0 = frl >> 1;
      const int I1 = I0 < crow_hi ? I0 : crow_hi;
      const int I = I1 > crow_lo ? I1 : crow_lo;
      r.e = make_double2(ec[I * cnc + jc0_ld], ec[I * cnc + jc1_ld]);
    }
    ++frow;
  };
#pragma unroll
  for (int u = 0; u < D; ++u) fetch(setA[u]);

  double w[S + E][3][C], fr[6][C];
#pragma unroll
  for (int s = 0; s < S + E; ++s)
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < C; ++c) w[s][r][c] = 0.0;
#pragma unroll
  for (int s = 0; s < 6; ++s)
#pragma unroll
    for (int c = 0; c < C; ++c) fr[s][c] = 0.0;
  constexpr int NE = SPRE + 2;
  // coarse correction values of fine row (row - k): er = columns jc0, jc1; el = the columns left of them
  double er[NE][2], el[NE][2];
#pragma unroll
  for (int k = 0; k < NE; ++k) er[k][0] = er[k][1] = el[k][0] = el[k][1] = 0.0;
  // left neighbours of the two coarse columns: lane - 1's, and across the seam lane 63's LEFT-half value for lane 0
  auto coarse_left = [&](const double* e, double* l) __attribute__((always_inline)) {
    l[0] = MGCMT_FETCH_LEFT(lane_up, e[0]);
    l[1] = MGCMT_FETCH_LEFT_ROT(lane_upr, last_lane ? e[0] : e[1]);
  };
  if (PROLONG) {
    const int I = (rstart - 1) >> 1;
    if (I >= crow_lo && I <= crow_hi) {
      if (cin0) er[0][0] = ec[I * cnc + jc0];
      if (cin1) er[0][1] = ec[I * cnc + jc1];
    }
    coarse_left(er[0], el[0]);
  }
  double racc[2] = {0.0, 0.0};
  unsigned okbits = 0;

  auto step = [&](auto pos, auto chk, const int row, const Row& in) __attribute__((always_inline)) {
    constexpr int T = decltype(pos)::value;
    constexpr bool CHK = decltype(chk)::value;
    const bool rok = CHK ? row_ok(row) : true;
    if (CHK) okbits = (okbits << 1) | (rok ? 1u : 0u);
    double x[C] = {0.0, 0.0, 0.0, 0.0};
    if (!ZERO_IN) {
      const double m0 = rok ? mask[0] : 0.0, m1 = rok ? mask[1] : 0.0;
      x[0] = in.v0.x * m0;
      x[1] = in.v0.y * m0;
      x[2] = in.v1.x * m1;
      x[3] = in.v1.y * m1;
    }
    if (PROLONG) {
#pragma unroll
      for (int k = NE - 1; k > 0; --k) {
        er[k][0] = er[k - 1][0];
        er[k][1] = er[k - 1][1];
        el[k][0] = el[k - 1][0];
        el[k][1] = el[k - 1][1];
      }
      er[0][0] = in.e.x * (rok ? cmsk[0] : 0.0);
      er[0][1] = in.e.y * (rok ? cmsk[1] : 0.0);
      coarse_left(er[0], el[0]);
    }
    // V += P e on fine row (row - lag): odd fine column takes c[J], even (c[J-1] + c[J]) / 2; an even fine row the
    // mean of coarse rows I-1 and I
    auto correct = [&](int lag, double* va) __attribute__((always_inline)) {
      double c[C] = {0.5 * (el[lag][0] + er[lag][0]), er[lag][0], 0.5 * (el[lag][1] + er[lag][1]), er[lag][1]};
      if (((T - lag - (S + E)) & 1) == 0) {
        const double p[C] = {0.5 * (el[lag + 1][0] + er[lag + 1][0]), er[lag + 1][0], 0.5 * (el[lag + 1][1] + er[lag + 1][1]), er[lag + 1][1]};
#pragma unroll
        for (int k = 0; k < C; ++k) c[k] = 0.5 * (p[k] + c[k]);
      }
      const bool ok = !CHK || ((okbits >> lag) & 1u);
      const double m0 = ok ? mask[0] : 0.0, m1 = ok ? mask[1] : 0.0;
      va[0] = fma(m0, c[0], va[0]);
      va[1] = fma(m0, c[1], va[1]);
      va[2] = fma(m1, c[2], va[2]);
      va[3] = fma(m1, c[3], va[3]);
    };
    if (PROLONG && SPRE == 0) correct(0, x);

    fr[T][0] = in.f0.x;
    fr[T][1] = in.f0.y;
    fr[T][2] = in.f1.x;
    fr[T][3] = in.f1.y;

    double o[C] = {x[0], x[1], x[2], x[3]};
#pragma unroll
    for (int s = 0; s <= S; ++s) {
      if (s == S && !RESTRICT) break;
      const int sn = mod3(T - s), sa = mod3(T - s + 1), sc = mod3(T - s + 2);
#pragma unroll
      for (int c = 0; c < C; ++c) w[s][sn][c] = o[c];
      const int rs = row - (s + 1);
      double ca[C];
#pragma unroll
      for (int c = 0; c < C; ++c) ca[c] = w[s][sc][c];
      const int fs = modn(T - (s + 1), 6);
      // lateral neighbours of the two pairs: the neighbouring lanes', and across the seam (column 127 | 128) lane 63's
      // left-half pair meets lane 0's right-half pair
      auto neighbours = [&](const double* c4, double* west, double* east) __attribute__((always_inline)) {
        west[0] = MGCMT_FETCH_LEFT(lane_up, c4[1]);
        west[1] = c4[0];
        east[0] = c4[1];
        east[1] = MGCMT_FETCH_RIGHT_ROT(lane_dnr, first_lane ? c4[2] : c4[0]);
        west[2] = MGCMT_FETCH_LEFT_ROT(lane_upr, last_lane ? c4[1] : c4[3]);
        west[3] = c4[2];
        east[2] = c4[3];
        east[3] = MGCMT_FETCH_RIGHT(lane_dn, c4[2]);
      };
      if (s < S) {
        if (!CHK || ((okbits >> (s + 1)) & 1u) != 0) {
          double west[C], east[C];
          neighbours(ca, west, east);
#pragma unroll
          for (int c = 0; c < C; ++c) {
            const double off = fma(cn, w[s][sa][c] + w[s][sn][c], cw * (west[c] + east[c]));
            o[c] = fma(om[c >> 1], (fr[fs][c] - fma(d, ca[c], off)) * invd, ca[c]);
          }
        } else {
#pragma unroll
          for (int c = 0; c < C; ++c) o[c] = ca[c];
        }
        if (PROLONG && SPRE > 0 && s == SPRE - 1) correct(s + 1, o);
        if (s == S - 1) {
          const int rout = row - S;
          if (STORE_V && (!CHK || (rout >= r_begin && rout < r_end))) {
            if (out0) store2_stream(vout + (long)rout * nc + ja0, o[0], o[1]);
            if (out1) store2_stream(vout + (long)rout * nc + ja1, o[2], o[3]);
          }
        }
      } else {
        double r[C] = {0.0, 0.0, 0.0, 0.0};
        if (!CHK || ((okbits >> (s + 1)) & 1u) != 0) {
          double west[C], east[C];
          neighbours(ca, west, east);
#pragma unroll
          for (int c = 0; c < C; ++c) {
            const double off = fma(cn, w[s][sa][c] + w[s][sn][c], cw * (west[c] + east[c]));
            r[c] = mask[c >> 1] * (fr[fs][c] - fma(d, ca[c], off));
          }
        }
        // residuals at the columns right of each pair (ja0 + 2: the right lane's left-half first column, across the
        // seam lane 0's right-half one; ja1 + 2: the right lane's right-half first column)
        const double rn0 = MGCMT_FETCH_RIGHT_ROT(lane_dnr, first_lane ? r[2] : r[0]);
        const double rn1 = MGCMT_FETCH_RIGHT(lane_dn, r[2]);
        const double h[2] = {0.25 * r[0] + 0.5 * r[1] + 0.25 * rn0, 0.25 * r[2] + 0.5 * r[3] + 0.25 * rn1};
        if (((T - (s + 1) - (S + E)) & 1) == 0) {
          const int I = (rs >> 1) - 1;  // coarse row closed by fine row rs = 2I + 2
          if (!CHK || (2 * I >= r_begin && 2 * I < r_end)) {
            if (out0 && cin0) rc[I * cnc + jc0] = racc[0] + 0.25 * h[0];
            if (out1 && cin1) rc[I * cnc + jc1] = racc[1] + 0.25 * h[1];
          }
          racc[0] = 0.25 * h[0];
          racc[1] = 0.25 * h[1];
        } else {
          racc[0] += 0.5 * h[0];
          racc[1] += 0.5 * h[1];
        }
      }
    }
  };

  auto body = [&](auto chk, const int base) __attribute__((always_inline)) {
    static_for<0, B / D>([&](auto g) __attribute__((always_inline)) {
      constexpr int G = decltype(g)::value;
      Row* cur = (G % 2 == 0) ? setA : setB;
      Row* nxt = (G % 2 == 0) ? setB : setA;
#pragma unroll
      for (int u = 0; u < D; ++u) fetch(nxt[u]);
      static_for<0, D>([&](auto u) __attribute__((always_inline)) {
        constexpr int U = decltype(u)::value;
        step(StepIndex<G * D + U>{}, chk, base + G * D + U, cur[U]);
      });
    });
  };
  const int fast_lo = r_begin + S + 3;
  const int fast_hi = (r_end + S < row_hi ? r_end + S : row_hi) - (B - 1);
  int base = rstart;
#pragma nounroll
  for (int phase = 0; phase < 2; ++phase) {
    int stop = rstop;
    if (phase == 0) {
      stop = rstart + ((fast_lo - rstart + B - 1) / B) * B;
      if (stop > rstop) stop = rstop;
    }
#pragma nounroll
    for (; base < stop; base += B) body(Checked<true>{}, base);
    if (phase == 0) {
#pragma nounroll
      for (; base < fast_hi; base += B) {
        body(Checked<false>{}, base);
        okbits = ~0u;
      }
    }
  }
}

template <int NSWEEP, int FLAGS>
void launch_wide_one(hipStream_t s, FusedArgs a, int k) {
  using Shape = WideShape<NSWEEP, FLAGS>;
  const long groups = (a.nc + Shape::wout - 1) / Shape::wout;
  const long groups8 = (groups + 7) / 8 * 8;
  a.n_col_groups = (int)groups;
  static int resident_blocks = 0;
  if (resident_blocks == 0) {
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_fused_wide<NSWEEP, FLAGS>, 64, 0) != hipSuccess ||
        hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess || per_cu < 1)
      resident_blocks = 512;
    else
      resident_blocks = per_cu * prop.multiProcessorCount;
  }
  long rows = fused_rows_override();
  if (rows <= 0) {
    long chunks = (long)(0.9 * resident_blocks) / (groups * k);
    if (chunks < 1) chunks = 1;
    rows = (a.nr + chunks - 1) / chunks;
    if (rows < kFusedMinRows) rows = kFusedMinRows;
  }
  if (rows > a.nr) rows = a.nr;
  rows = (rows + 1) & ~1L;
  a.rows_per_chunk = (int)rows;
  a.n_row_chunks = (int)((a.nr + rows - 1) / rows);
  const unsigned blocks = (unsigned)(groups8 * a.n_row_chunks);
  hipLaunchKernelGGL((k_fused_wide<NSWEEP, FLAGS>), dim3(blocks, (unsigned)k), dim3(64), 0, s, a);
}

// every flag combination the cycle uses with weighted Jacobi on a 5-point level; false: not covered (narrow kernel)
template <int NSWEEP>
bool launch_wide_variant(hipStream_t s, const FusedArgs& a, int flags, int k) {
  switch (flags) {
    case 0: launch_wide_one<NSWEEP, 0>(s, a, k); return true;
    case kZeroIn: launch_wide_one<NSWEEP, kZeroIn>(s, a, k); return true;
    case kRestrict: launch_wide_one<NSWEEP, kRestrict>(s, a, k); return true;
    case kRestrict | kZeroIn: launch_wide_one<NSWEEP, kRestrict | kZeroIn>(s, a, k); return true;
    case kRestrict | kNoStore: launch_wide_one<NSWEEP, kRestrict | kNoStore>(s, a, k); return true;
    case kRestrict | kNoStore | kZeroIn: launch_wide_one<NSWEEP, kRestrict | kNoStore | kZeroIn>(s, a, k); return true;
    case kProlong: launch_wide_one<NSWEEP, kProlong>(s, a, k); return true;
    case kProlong | (1 << kPreShift): launch_wide_one<NSWEEP, kProlong | (1 << kPreShift)>(s, a, k); return true;
    case kProlong | kZeroIn | (1 << kPreShift): launch_wide_one<NSWEEP, kProlong | kZeroIn | (1 << kPreShift)>(s, a, k); return true;
    case kProlong | (2 << kPreShift): launch_wide_one<NSWEEP, kProlong | (2 << kPreShift)>(s, a, k); return true;
    case kProlong | kZeroIn | (2 << kPreShift): launch_wide_one<NSWEEP, kProlong | kZeroIn | (2 << kPreShift)>(s, a, k); return true;
    default: return false;
  }
}

}  // namespace fused

}  // namespace mgcmt
