// Lexicographic Gauss-Seidel / SOR as a wavefront over row bands — the reference's default smoother
// (MGCMTSolver.py:210-246) without a scan: the second pipeline for the sweep kernels_lexwave.hip describes.
//
// Point (i, j) takes NEW values from (i-1, j-1..j+1) and (i, j-1).  A band is 63 consecutive rows, one row per lane
// of ONE wave; at step t lane l works on column j = t - s l of its row (s = 1 for the 5-point operator, 2 when the
// stencil has corners).  Then every new value a lane needs was produced EARLIER — by itself one step ago (W) or by the
// lane above it one to three steps ago (NE, N, NW) — so a step is a handful of multiply-adds and two lane shifts, not a
// 64-wide scan: the band advances one anti-diagonal per step.  Old values come in ONE stream per lane, the lane's own
// row at its own column (+ the right-hand side): E and the point's own value are that stream one step ahead / now,
// and the row below is the NEXT lane's stream shifted up by a lane (lane 63 computes nothing: it only streams the
// first row of the band below).  Each lane reads and writes its own row — 64 rows per load instruction, one cache
// line per lane every 16 steps — which the L1/L2 absorb; the arithmetic per point is what the scan form needs for the
// operator alone.
//
// Bands form a one-directional pipeline like the column blocks of kernels_lexwave.hip: band B needs, for lane 0, the
// new values of the last row of band B-1, one column per step.  They travel as {tag, half a double} granules written by
// ONE write-through store per step ("the data is the flag") into a buffer cleared before the launch, are asked for
// kRec steps ahead and, when they come back incomplete, asked for again visibly (bounded by a timeout that raises the
// error word).  Band B can start about 63 s steps after band B-1: the sweep takes (bands * (63 s + hand-off) + columns)
// steps instead of (blocks + rows) scans.
#include "lex_util.h"

namespace mgcmt {

namespace {

constexpr int kBandRows = 63;

struct LexBandArgs {
  double* v;
  const double* f;
  long vstride;
  int nr, nc, nbands;
  double c[3][3];                    // interior coefficients [di + 1][dj + 1]
  double crow[3], ccol[3], ccorner;  // last row: own-row coefficients (W, C, E); last column: centre column (N, C, S)
  const double* shifts;
  double alpha, beta, wU, wL;
  unsigned long long* carry;  // [vector][band][step][2] granules {tag = 1 : 32, half of a double : 32}: the new value of lane 62
  unsigned* sync;             // [0] ticket, [1] error
  long carry_stride;          // granules per vector
  long band_stride;           // granules per band
};

// FIVE: constant 5-point operator (no corner terms, no special last row / column); OWN: the sweep uses the point's own
// old value (alpha != 0: the homogeneous SOR recurrence)
template <bool FIVE, bool OWN>
__global__ void __launch_bounds__(64) k_lex_band(LexBandArgs a) {
  constexpr int SK = FIVE ? 1 : 2;  // columns a lane trails the lane above it
  constexpr int LA = FIVE ? 1 : 3;  // steps ahead of use at which the old-value stream is taken out of its slot
  constexpr int H = kBandRows;
  const int lane = threadIdx.x;
  // band number = order of arrival: whoever this band waits for has started before it
  unsigned ticket = 0;
  if (lane == 0) ticket = atomicAdd(&a.sync[0], 1u);
  ticket = (unsigned)uniform(__shfl((int)ticket, 0));
  const int q = (int)(ticket / (unsigned)a.nbands);
  const int B = (int)(ticket % (unsigned)a.nbands);
  const int nr = a.nr, nc = a.nc;
  double* __restrict__ v = a.v + (long)q * a.vstride;
  const double* __restrict__ f = a.f + (long)q * a.vstride;
  u64* my_rec = a.carry + (long)q * a.carry_stride + (long)B * a.band_stride;
  const u64* rec_src = B > 0 ? my_rec - a.band_stride : my_rec;  // (band 0 has no band above: any valid address, result unused)
  unsigned* const err_word = a.sync + 1;

  const int r0 = B * H;
  const int nreal = nr - r0 < H ? nr - r0 : H;  // rows of this band
  if (nreal <= 0) return;                       // (cannot happen for nbands = ceil(nr / 63))
  const int r = r0 + lane;
  const bool real = lane < nreal;
  const int rr = r < nr ? r : nr;        // the row this lane streams (row nr: the zero halo row)
  const int T = nc + SK * (nreal - 1);   // steps: the last real lane finishes column nc - 1
  const int Tup = nc + SK * (H - 1);     // ... of the (full) band above

  // per-lane operator constants: the row decides the own-row coefficients (a 9-point level's last row has its own), the
  // column only matters on the last column (second set); lanes without a row get zeros, so they produce zeros
  const double mu = a.shifts[q];
  const double alpha = a.alpha, beta = a.beta, wU = a.wU, wL = a.wL;
  const bool lastrow = !FIVE && r == nr - 1;
  double cW = a.c[1][0], cE = a.c[1][2], dI = a.c[1][1] - mu, dC = a.ccol[1] - mu;
  if (lastrow) {
    cW = a.crow[0];
    cE = a.crow[2];
    dI = a.crow[1] - mu;
    dC = a.ccorner - mu;
  }
  const double invI = 1.0 / dI, invC = 1.0 / dC;
  const double live = real ? 1.0 : 0.0;
  const double kF = live * beta * invI, kO = live * alpha, kE = -live * wU * cE * invI, kS = -live * wU * a.c[2][1] * invI,
               kN = -live * wL * a.c[0][1] * invI, kSW = -live * wU * a.c[2][0] * invI, kSE = -live * wU * a.c[2][2] * invI,
               kNW = -live * wL * a.c[0][0] * invI, kNE = -live * wL * a.c[0][2] * invI, q0 = -live * wL * cW * invI;
  // last column (9-point levels): centre-column coefficients and diagonal of their own; everything right of it is zero
  const double kFc = live * beta * invC, kOc = live * alpha * dC * invC, kSc = -live * wU * a.ccol[2] * invC,
               kNc = -live * wL * a.ccol[0] * invC, kSWc = -live * wU * a.c[2][0] * invC, kNWc = -live * wL * a.c[0][0] * invC,
               q0c = -live * wL * cW * invC;
  const double lane0 = lane == 0 ? 1.0 : 0.0;

  // One stream per lane: step tau -> (v_old, f) at column tau - SK lane of row rr, clamped into the row (masked on use).
  constexpr int kSlots = 12;  // stream steps in flight + the one in use
  constexpr int kRec = 4;     // steps ahead at which the record of the band above is asked for
  constexpr int kOps = 5;     // per step: store, record store, record load, two stream loads
  // A step issues [row store, record store, record load of step t + kRec, stream loads of step t + LA + kSlots]; loads
  // complete in order, so the wait of step t — for the record of step t, issued kRec steps ago — lets only what was
  // issued after it stay in flight; the stream slot taken out at step t (step t + LA's) is older than that record.
  constexpr int kWaitN = 2 + (kRec - 1) * kOps;
  static_assert(kWaitN <= 63, "vmcnt is a 6-bit counter");
  static_assert(kSlots >= LA + kRec, "the stream value taken out at step t must be older than the record of step t");
  static_assert(kSlots % (LA + 1) == 0, "the histories rotate back into place once per trip");
  double Wv[kSlots], Fv[kSlots];
  u64 Rv[kSlots];
#pragma unroll
  for (int d = 0; d < kSlots; ++d) {
    Wv[d] = 0.0;
    Fv[d] = 0.0;
    Rv[d] = 0;
  }
  const char* vb = reinterpret_cast<const char*>(v + (long)r0 * nc);
  const char* fb = reinterpret_cast<const char*>(f + (long)r0 * nc);
  const unsigned row_off = (unsigned)((rr - r0) * nc) * 8u;  // this lane's row inside the band (bytes; 63 rows: fits 32 bits)
  const int skl8 = SK * lane * 8, hi8 = (nc - 1) * 8;
  auto issue_old = [&](auto plain, auto slot, int tau) __attribute__((always_inline)) {
    constexpr int SL = decltype(slot)::value;
    const unsigned off = row_off + (unsigned)clamp0(tau * 8 - skl8, hi8);
    double &w_ = Wv[SL], &f_ = Fv[SL];  // (named here: a variable that only an asm statement mentions is not captured)
    const char *vb_ = vb, *fb_ = fb;
    if (decltype(plain)::value) {
      w_ = *reinterpret_cast<const double*>(vb_ + off);
      f_ = *reinterpret_cast<const double*>(fb_ + off);
      return;
    }
    MGCMT_LEX_LOAD_AT(w_, off, vb_, 0);
    MGCMT_LEX_LOAD_AT(f_, off, fb_, 0);
  };
  // the record lane 0 needs at step t: the band above's lane 62 at column t + SK - 1, which is ITS step t + SK - 1 + 62 SK
  const unsigned roff = (unsigned)(lane & 1) * 8u;
  auto rec_index = [&](int t) __attribute__((always_inline)) {
    const int s = t + (SK - 1) + SK * (H - 1);
    return s < Tup - 1 ? s : Tup - 1;
  };
  auto issue_rec = [&](auto plain, auto slot, int t) __attribute__((always_inline)) {
    constexpr int SL = decltype(slot)::value;
    u64& r_ = Rv[SL];
    const u64* rbase = rec_src + (long)rec_index(t) * 2;
    if (decltype(plain)::value) {
      r_ = load_granule(rbase + (lane & 1));
      return;
    }
    const unsigned ro = roff;
    MGCMT_LEX_LOAD_AT_SC1(r_, ro, rbase);
  };
  bool failed = false;
  auto unpack = [&](u64 R, double& c) __attribute__((always_inline)) {  // false: the record is not complete yet
    const u64 g0 = lane_bits(R, 0), g1 = lane_bits(R, 1);
    c = __builtin_bit_cast(double, (g0 & 0xffffffffull) | (g1 << 32));
    return ((g0 & g1) >> 32) == 1ull;
  };
  auto wait_record = [&](int t, double& c) __attribute__((always_inline)) {  // the slow path: ask until the record is complete
    u64 t0 = 0;
    bool timing = false;
    const u64* p = rec_src + (long)rec_index(t) * 2 + (lane & 1);
    while (true) {
      const u64 R = load_granule(p);
      if (unpack(R, c)) return;
      if (!timing) {
        t0 = now_ticks();
        timing = true;
      }
      nap();
      if (now_ticks() - t0 > kTimeoutTicks || load_word(err_word) != 0u) {
        failed = true;
        c = 0.0;
        return;
      }
    }
  };

  // fill the pipeline with loads the compiler sees, wait for them, and make it see that they are done (see
  // kernels_lexwave.hip: otherwise it throttles the loop with waits of its own)
  for_slots<kSlots>([&](auto sl) __attribute__((always_inline)) { issue_old(Checked<true>{}, sl, decltype(sl)::value); });
  for_slots<kRec>([&](auto sl) __attribute__((always_inline)) { issue_rec(Checked<true>{}, sl, decltype(sl)::value); });
  drain_loads();
#pragma unroll
  for (int d = 0; d < kSlots; ++d) {
    settle(Wv[d]);
    settle(Fv[d]);
    settle(Rv[d]);
  }

  // stream histories: wm[k] = masked old value of step t + k, fh[k] = right-hand side of step t + k (k < LA), ws[k] = the
  // lane below's wm[k] (k >= 1); xs[k] = the lane above's new value of step t - 1 - k (lane 0: the band above's record)
  double wm[LA], fh[LA], ws[LA], xs[LA];
#pragma unroll
  for (int k = 0; k < LA; ++k) wm[k] = fh[k] = ws[k] = xs[k] = 0.0;
  auto masked = [&](double w, int tau) __attribute__((always_inline)) {  // the stream's value of step tau, zero outside the row
    const int col = tau - SK * lane;
    return (unsigned)col < (unsigned)nc ? w : 0.0;
  };
  for_slots<LA>([&](auto sl) __attribute__((always_inline)) {
    constexpr int K = decltype(sl)::value;
    wm[K] = masked(take(Wv[K]), K);
    fh[K] = take(Fv[K]);
    ws[K] = from_right(wm[K], lane);
    issue_old(Checked<false>{}, sl, kSlots + K);
  });
  double x1 = 0.0;  // this lane's new value of the previous step (W)
  if (!FIVE && B > 0) {  // corner stencils: lane 0's N of step 0 is the record of column 0 ("step -1": NE one step earlier)
    double c = 0.0;
    wait_record(-1, c);
    xs[0] = c * lane0;
  }

  auto step = [&](auto ph, int t) __attribute__((always_inline)) {
    constexpr int PH = decltype(ph)::value, SA = (PH + LA) % kSlots;
    // BULK: every lane with a row is on an interior column and every stream value in use lies inside its row: no masks
    const bool BULK = t >= SK * H && t + LA <= nc - 1;
    wait_loads3<kWaitN>(Wv[SA], Fv[SA], Rv[PH]);
    double wnew = take(Wv[SA]);  // (both live on as histories)
    const double fnew = take(Fv[SA]);
    const u64 rq_now = Rv[PH];
    const int j = t - SK * lane;
    bool act = real;
    if (!BULK) {
      wnew = masked(wnew, t + LA);
      act = real && (unsigned)j < (unsigned)nc;
    }
    const double wsnew = from_right(wnew, lane);
    // the band above's value for lane 0 (column t + SK - 1 of the row above the band)
    double c = 0.0;
    if (B > 0 && t + SK - 1 <= nc - 1) {
      if (!unpack(rq_now, c)) wait_record(t, c);
    }
    const double xsnew = fma(c, lane0, from_left(x1, lane));
    // the stencil: W = x1; 5-point: N = xsnew, E = wnew, S = wsnew; 9-point: NE = xsnew, N = xs[0], NW = xs[1],
    // E = wm[1], SW = ws[1], S = ws[2], SE = wsnew
    const double own = wm[0], fv = fh[0];
    double e, s, n, sw = 0.0, se = 0.0, nw = 0.0, ne = 0.0;
    if (FIVE) {
      e = wnew;
      s = wsnew;
      n = xsnew;
    } else {
      e = wm[1 % LA];
      sw = ws[1 % LA];
      s = ws[2 % LA];
      se = wsnew;
      ne = xsnew;
      n = xs[0];
      nw = xs[1 % LA];
    }
    double p = kF * fv;
    if (OWN) p = fma(kO, own, p);
    p = fma(kE, e, p);
    p = fma(kS, s, p);
    p = fma(kN, n, p);
    if (!FIVE) {
      p = fma(kSW, sw, p);
      p = fma(kSE, se, p);
      p = fma(kNW, nw, p);
      p = fma(kNE, ne, p);
    }
    double x = fma(q0, x1, p);
    if (!BULK) {
      if (!FIVE) {  // the lane on the last column: everything right of it is zero already (masked / outside lanes)
        double pc = kFc * fv;
        pc = fma(kOc, own, pc);
        pc = fma(kSc, s, pc);
        pc = fma(kNc, n, pc);
        pc = fma(kSWc, sw, pc);
        pc = fma(kNWc, nw, pc);
        const double xc = fma(q0c, x1, pc);
        if (j == nc - 1) x = xc;
      }
      if (!act) x = 0.0;
    }
    // (wave-uniform band base + step + 32-bit lane offset)
    if (act) *reinterpret_cast<double*>(const_cast<char*>(vb) + (long)t * 8 + (unsigned)(row_off - (unsigned)skl8)) = x;
    {  // the step's record: lane 62's value as {tag, half a double} granules from lanes 62 (low half) and 63 (high half)
      const double t1 = from_left(x, lane);
      const u64 bits = __builtin_bit_cast(u64, lane == 63 ? t1 : x);
      const u64 word = (1ull << 32) | ((lane & 1) ? (bits >> 32) : (bits & 0xffffffffull));
      if (lane >= 62) store_granule(reinterpret_cast<u64*>(reinterpret_cast<char*>(my_rec + (long)t * 2 - 62) + (unsigned)(lane * 8)), word);
    }
    // the histories move on by a step
    x1 = x;
#pragma unroll
    for (int k = LA - 1; k > 0; --k) xs[k] = xs[k - 1];
    xs[0] = xsnew;
#pragma unroll
    for (int k = 0; k + 1 < LA; ++k) {
      wm[k] = wm[k + 1];
      fh[k] = fh[k + 1];
      ws[k] = ws[k + 1];
    }
    wm[LA - 1] = wnew;
    fh[LA - 1] = fnew;
    ws[LA - 1] = wsnew;
    // refill, LAST in the step (the slots' old contents are dead: the loads land in the very registers the slots had)
    issue_rec(Checked<false>{}, Int<(PH + kRec) % kSlots>{}, t + kRec);
    issue_old(Checked<false>{}, Int<SA>{}, t + LA + kSlots);
  };
  for (int t = 0; t < T && !failed; t += kSlots)
    for_slots<kSlots>([&](auto sl) __attribute__((always_inline)) {
      if (t + decltype(sl)::value < T) step(sl, t + decltype(sl)::value);
    });
  if (failed && lane == 0) store_word(err_word, 1u);  // tell the host and release everyone behind this band
}

}  // namespace

long lex_band_count(const KGrid& g) { return (g.nr + kBandRows - 1) / kBandRows; }
// granules of one band's record stream (one pair per step, corner stencils take two steps per row of lag)
long lex_band_stride(const KGrid& g) { return 2 * (g.nc + 2 * (kBandRows - 1) + 2); }

void launch_lex_band(hipStream_t s, KGrid g, KOp op, KVec v, KVec f, const double* shifts, double alpha, double beta, double wU,
                     double wL, int k, double* carry, unsigned* sync) {
  LexBandArgs a{};
  a.v = v.p;
  a.f = f.p;
  a.vstride = v.stride;
  a.nr = (int)g.nr;
  a.nc = (int)g.nc;
  a.nbands = (int)lex_band_count(g);
  if (op.five_point) {
    const double c5[3][3] = {{0.0, op.cn, 0.0}, {op.cw, op.c0, op.cw}, {0.0, op.cn, 0.0}};
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) a.c[i][j] = c5[i][j];
      a.crow[i] = c5[1][i];
      a.ccol[i] = c5[i][1];
    }
    a.ccorner = op.c0;
  } else {
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) a.c[i][j] = op.c9[i][j];
      a.crow[i] = op.c9row[i];
      a.ccol[i] = op.c9col[i];
    }
    a.ccorner = op.c9corner;
  }
  a.shifts = shifts;
  a.alpha = alpha;
  a.beta = beta;
  a.wU = wU;
  a.wL = wL;
  a.carry = reinterpret_cast<unsigned long long*>(carry);
  a.sync = sync;
  a.band_stride = lex_band_stride(g);
  a.carry_stride = a.band_stride * a.nbands;
  (void)hipMemsetAsync(sync, 0, sizeof(unsigned), s);  // (the ticket; the error word stays until the host has looked)
  (void)hipMemsetAsync(carry, 0, sizeof(unsigned long long) * (size_t)k * a.carry_stride, s);
  const dim3 grid((unsigned)(a.nbands * k));
  if (op.five_point && alpha == 0.0) hipLaunchKernelGGL((k_lex_band<true, false>), grid, dim3(64), 0, s, a);
  else if (op.five_point) hipLaunchKernelGGL((k_lex_band<true, true>), grid, dim3(64), 0, s, a);
  else hipLaunchKernelGGL((k_lex_band<false, true>), grid, dim3(64), 0, s, a);
}

}  // namespace mgcmt
