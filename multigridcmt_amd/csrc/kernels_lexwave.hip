// Lexicographic Gauss-Seidel / SOR as a pipeline of waves — the reference's default smoother
// (MGCMTSolver.py:210-246; ThesisProblem.py:101 and the UnitTests use it) on the whole chip instead of one workgroup.
//
// The sweep runs in index order k = i*cols + j: point (i, j) takes NEW values from (i-1, j-1..j+1) and (i, j-1) and
// OLD values from (i, j+1) and (i+1, j-1..j+1).  In the skewed column index j' = j + i every new-value neighbour lies at
// j' - 2 .. j' and every old-value neighbour at j' .. j' + 2, so blocks of 64 consecutive j' (one wave each, one column
// per lane, the window sliding one column to the left per row) depend on their LEFT neighbour block only — for the two
// values at its right edge per row — and read old values only from themselves and from blocks to their RIGHT, which
// run behind them and have not touched those values yet.  The blocks therefore form a one-directional pipeline: block J
// marches down its rows as soon as block J-1 has published the edge values of the same rows.
//
// Inside a row the recurrence v_j = p_j + q_j v_{j-1} is solved by a scan over the affine maps x -> p + q x (DPP row
// shifts and broadcasts, no LDS); the carry-in (the left block's edge value) enters only at the end of the scan, so
// waiting for it is off the critical path.
//
// Hand-off (MI355X_MICROARCH.md, inter-workgroup visibility, "the data is the flag"): the two edge values of a row
// travel as four 8-byte granules {tag, half a double}, each written by ONE agent-scope atomic store (write-through,
// sc1) into a buffer cleared before the launch; the consumer reads them with agent-scope atomic loads (sc1: they bypass
// its CU's L1) and takes a row when its four tags are set — no flag, no fence, no drain of the producer's memory
// pipeline; the load is issued two rows ahead of its use, so a consumer that runs a few rows behind its producer never
// waits.  Block numbers are handed out by a
// ticket counter, so a block only ever waits for one that has already started; every spin is bounded by a timeout
// that raises an error word instead of hanging the GPU.
//
// Covers the constant-coefficient operators (the scaled / shifted Laplacian and its Galerkin coarsenings, whose
// factors are Toeplitz except for their last diagonal entry); other operators and small grids keep kernels_lex.hip.
#include "lex_util.h"

namespace mgcmt {

namespace {

// Rows of old values (and edge records) in flight ahead of the row being processed: vector-memory operations complete in
// order and a row's write-through record store takes microseconds to be acknowledged, so the depth that hides it is as
// many rows as the 6-bit vmcnt counter can hold operations for (5 to 7 per row)

struct LexWaveArgs {
  double* v;
  const double* f;
  long vstride;
  int nr, nc, nblocks;
  int five;                          // constant 5-point operator: no corner terms, no special last row / column
  double c[3][3];                    // interior coefficients [di + 1][dj + 1]
  double crow[3], ccol[3], ccorner;  // last row: own-row coefficients (W, C, E); last column: centre column (N, C, S)
  const double* shifts;
  double alpha, beta, wU, wL;
  double gamma;               // what is stored is x + gamma f (x goes on down the recurrence): reference SOR's v += w (D-L)^-1 f in the sweep
  unsigned long long* carry;  // [sweep][vector][block][row][4] granules {tag = 1 : 32, half of a double : 32}: lanes 62 / 63's new values
  unsigned* sync;             // [1] error word (set by a block that gave up; cleared by the host when it reports it), [2..] diagnostics
  unsigned* ticket;           // block numbers (the first word of the scratch buffer: cleared with it, ONE memset per launch)
  long carry_stride;          // granules per vector
  int nsweeps;                // sweeps chained in this launch (1: the plain sweep)
  int nvec;                   // vectors in this launch
  unsigned long long* progress;  // [sweep][vector][block] rows finished with all stores acknowledged (chained sweeps; cleared per launch)
  int rec_rows;               // rows of a block's record array: nr + kTerminalRows
  long sweep_stride;          // granules per sweep
};

constexpr int kTerminalRows = 16;  // (spare rows behind a block's records: the record prefetch runs a few rows past the last one)
#ifndef MGCMT_LEX_REC
#define MGCMT_LEX_REC 4  // rows ahead at which the left block's records are asked for (tuning)
#endif
#ifndef MGCMT_LEX_PROGRESS_TRIPS
#define MGCMT_LEX_PROGRESS_TRIPS 3  // chained sweeps: trips between two publications of a block's progress (tuning)
#endif
#ifndef MGCMT_LEX_CHAIN_REC
#define MGCMT_LEX_CHAIN_REC 4  // chained sweeps: rows ahead at which records are asked for (tuning)
#endif

// FIVE: constant 5-point operator (no corner terms, no special last row / column); OWN: the sweep uses the point's own
// old value (alpha != 0: the homogeneous SOR recurrence); CH: several sweeps chained in one launch.
//
// Chained sweeps (CH): sweep s runs behind sweep s - 1 on the same vector, in the same launch.  Block (s, J) may read the
// old values of a row R — sweep s - 1's results in its own window and the two columns right of it — once blocks J and J + 1
// of sweep s - 1 have finished row R AND those stores are visible.  The hand-off is the one MI355X_MICROARCH.md lists as
// valid without an acquire: every store of v in these kernels is `sc1` (write-through), every load of v an `sc1` load to
// registers; a block of the sweep ahead publishes its progress every few trips of its row loop — `s_waitcnt vmcnt(0)` (all of its
// stores acknowledged), then ONE `sc1` store of the number of rows it has finished — and the block behind polls that
// number with `sc1` loads before the trip that issues the loads of a row, keeping the last value it saw (so it polls
// about as often as the block ahead publishes).  (An earlier form took the sweep ahead's per-row edge records as the
// progress signal, behind the row loop's partial `vmcnt(N)` wait instead of a drain.  It gave the same bits on every
// level of a 16384^2 plan, but that wait is not the drain the guide's form asks for, and it cost a load, a tag check
// and pointer arithmetic per row in every block: 8.3 against 7.4 ms per 4096^2 V(2,2) cycle.)
// Blocks get their numbers in the order of (J + 2 s, s), so a block only ever waits for blocks that have started.
template <bool FIVE, bool OWN, bool CH>
__global__ void __launch_bounds__(64) k_lex_wave(LexWaveArgs a) {
  const int lane = threadIdx.x;
  // block number = order of arrival: whoever this block waits for has started before it
  unsigned ticket = 0;
  if (lane == 0) ticket = atomicAdd(a.ticket, 1u);
  ticket = (unsigned)uniform(__shfl((int)ticket, 0));
  const int nsw = CH ? a.nsweeps : 1;
  const int per_vector = CH ? (a.nblocks + 2 * (nsw - 1)) * nsw : a.nblocks;  // (chained: some numbers stand for no block)
  const int q = (int)(ticket / (unsigned)per_vector);
  const int tv = (int)(ticket % (unsigned)per_vector);
  const int sw = CH ? tv % nsw : 0;                 // the sweep this block belongs to
  const int J = CH ? tv / nsw - 2 * sw : tv;        // its block of columns
  if (CH && (J < 0 || J >= a.nblocks)) return;
  const int nr = a.nr, nc = a.nc;
  double* __restrict__ v = a.v + (long)q * a.vstride;
  const double* __restrict__ f = a.f + (long)q * a.vstride;
  const long rec_stride = (long)a.rec_rows * 4;  // granules per block
  u64* my_rec = a.carry + (long)sw * a.sweep_stride + (long)q * a.carry_stride + (long)J * rec_stride;
  const u64* left_rec = my_rec - rec_stride;

  const int i0 = J * 64 - (nc - 1) > 0 ? J * 64 - (nc - 1) : 0;  // first row with a column of this block inside the grid
  const int i1 = J * 64 + 63 < nr - 1 ? J * 64 + 63 : nr - 1;    // last one
  const int left_last = J > 0 ? (J * 64 - 1 < nr - 1 ? J * 64 - 1 : nr - 1) : -1;  // last row the left block works on
  unsigned* const err_word = a.sync + 1;  // (locals, not `a`, inside the lambdas: the argument block then stays out of memory)
  if (i0 > i1) return;  // (cannot happen for nblocks = ceil((nr + nc - 1) / 64); kept as a guard)

  const double mu = a.shifts[q];
  const double alpha = a.alpha, beta = a.beta, wU = a.wU, wL = a.wL, gamma = a.gamma;
  // coefficient classes: interior, last column, last row, corner
  const double cNW = a.c[0][0], cNE = a.c[0][2], cSW = a.c[2][0], cSE = a.c[2][2];
  const double cN_int = a.c[0][1], cS_int = a.c[2][1], cW_int = a.c[1][0], cE_int = a.c[1][2];
  const double cN_col = a.ccol[0], cS_col = a.ccol[2], cW_row = a.crow[0], cE_row = a.crow[2];
  const double d_int = a.c[1][1] - mu, d_col = a.ccol[1] - mu, d_row = a.crow[1] - mu, d_cor = a.ccorner - mu;
  const double inv_int = 1.0 / d_int, inv_col = 1.0 / d_col, inv_row = 1.0 / d_row, inv_cor = 1.0 / d_cor;
  // rows whose 64 columns are all interior points: p = kF f + kE e + kS s + ... with these constants, ONE q, so only the
  // p part of the maps is scanned; the products of q the scan needs are per-lane constants
  const double kF = beta * inv_int, kO = alpha, kE = -wU * cE_int * inv_int, kS = -wU * cS_int * inv_int, kN = -wL * cN_int * inv_int;
  const double kSW = -wU * cSW * inv_int, kSE = -wU * cSE * inv_int, kNW = -wL * cNW * inv_int, kNE = -wL * cNE * inv_int;
  // ... and for the lane on a 9-point level's last column (its centre-column coefficients and diagonal differ)
  const double kOc = alpha * d_col * inv_col, kSc = -wU * cS_col * inv_col, kNc = -wL * cN_col * inv_col;
  const double q0 = -wL * cW_int * inv_int;
  const double q2 = q0 * q0, q4 = q2 * q2, q8 = q4 * q4;
  double qpow = q0, Q15 = 0.0, Q31 = 0.0;  // q0^(lane + 1); q0^((lane & 15) + 1) on the second 16 of every 32; q0^(lane - 31) on the upper 32
  {
    double acc = 1.0;
    for (int m = 1; m <= 64; ++m) {
      acc *= q0;  // q0^m
      if ((lane & 16) && m == (lane & 15) + 1) Q15 = acc;
      if (lane >= 32 && m == lane - 31) Q31 = acc;
      if (m == lane + 1) qpow = acc;
    }
  }
  const double qcol = -wL * cW_int * inv_col;  // the last column's own q

  // Old values come in three streams per row r, all read at this lane's column jr = J*64 + lane - r of that row:
  //   W[r] = v[r][jr]   S[r] = v[r][jr + 1]   S2[r] = v[r][jr + 2]     and  F[r] = f[r][jr].
  // For row i: own = W[i], E = S[i]; the row below: SW = W[i+1], S = S[i+1], SE = S2[i+1] (its columns sit one lane to the
  // right).  No cross-lane traffic for old values; the overlapping loads are served by the L1.  Rows nr.. are the zero
  // halo rows.  RAW = the window and the two columns right of it lie inside the grid on that row: no clamping, no masks.
  constexpr bool USE_W = OWN || !FIVE, USE_S2 = !FIVE;
  constexpr int kLoads = 3 + (USE_W ? 1 : 0) + (USE_S2 ? 1 : 0);  // per row: [W] S [S2] F + the edge record
  constexpr int kOps = kLoads + 2;                                 // ... + the row's store and its record's store
  constexpr int kSlots = kOps <= 5 ? 12 : (kOps == 6 ? 10 : (kOps == 7 ? 9 : 8));  // pipeline slots = rows in flight + the one in use
  constexpr int kDepth = kSlots - 1;
  // The left block's record of row r is asked for kRec rows ahead only — not kDepth + 1 like the old values: a block can
  // run no closer behind its left neighbour than the distance at which its prefetched records come back complete, and the
  // sum of those lags over the blocks is the sweep's start-up time.  A step issues [row store, record store, record load of
  // row i + kRec, old values of row i + kDepth + 1]; loads complete in order, so the wait of step i — for the record of row
  // i, issued kRec steps ago — lets only what was issued after it stay in flight.
  constexpr int kRec = CH ? MGCMT_LEX_CHAIN_REC : MGCMT_LEX_REC;  // (chained sweeps load v past the caches: the longer latency wants more steps between a load and the wait behind it)
  constexpr int kProgressTrips = MGCMT_LEX_PROGRESS_TRIPS;  // chained sweeps: trips (of kSlots rows) between two publications of a block's progress
  constexpr int kWaitN = (kLoads - 1) + (kRec - 1) * kOps;
  static_assert(kWaitN <= 63, "vmcnt is a 6-bit counter");
  static_assert(kRec < kDepth, "the old values of rows i, i + 1 are older than the record of row i (row i + 1 is asked for kDepth steps ahead, behind that step's record)");
  double Wv[kSlots], Sv[kSlots], S2v[kSlots], Fv[kSlots];
  u64 Rv[kSlots];
#pragma unroll
  for (int d = 0; d < kSlots; ++d) {
    Wv[d] = 0.0;
    Sv[d] = 0.0;
    S2v[d] = 0.0;
    Fv[d] = 0.0;
    Rv[d] = 0;
  }
  const u64* rec_src = J > 0 ? left_rec : my_rec;  // (block 0 has no left neighbour: any valid address, result unused)
  // issue the loads of the next row's old values into slot SL (rows are asked for in order: i0, i0 + 1, ...).  The
  // addresses are NOT clamped into the row: a window that crosses the grid's edge reads the end of the previous row or the
  // start of the next one (the vectors carry MGCMT_HALO_ROWS rows of halo on either side), and whatever it finds there is
  // masked when the row is used.  So the address is a wave-uniform running pointer — row min(r, nr) (row nr: the zero halo
  // row), column J*64 - r — plus the lane's fixed offset: no vector arithmetic at all.
  // PLAIN: compiler-visible loads (the prologue, which the compiler may schedule and wait for as it likes); otherwise the
  // hand-counted asm loads of the row loop
  const unsigned lane8 = (unsigned)lane * 8u;
  const long row_stride = ((long)nc - 1) * 8;  // one row down, one column left
  int ld_r = i0;
  const char* ldv = reinterpret_cast<const char*>(v + (long)i0 * nc + (J * 64 - i0));
  const char* ldf = reinterpret_cast<const char*>(f + (long)i0 * nc + (J * 64 - i0));
  auto issue_old = [&](auto plain, auto slot) __attribute__((always_inline)) {
    constexpr int SL = decltype(slot)::value;
    double &w_ = Wv[SL], &s_ = Sv[SL], &s2_ = S2v[SL], &f_ = Fv[SL];  // (named here: a variable that only an asm
    const char *pv = ldv, *pf = ldf;                                   //  statement mentions is not captured)
    const unsigned off = lane8;
    if (decltype(plain)::value) {
      if (USE_W) w_ = load_value<CH>(pv + off);
      s_ = load_value<CH>(pv + off + 8);
      if (USE_S2) s2_ = load_value<CH>(pv + off + 16);
      f_ = *reinterpret_cast<const double*>(pf + off);
    } else {
      if (CH) {  // (v is being rewritten by the sweep ahead: past the caches)
        if (USE_W) MGCMT_LEX_LOAD_AT_SC1I(w_, off, pv, 0);
        MGCMT_LEX_LOAD_AT_SC1I(s_, off, pv, 8);
        if (USE_S2) MGCMT_LEX_LOAD_AT_SC1I(s2_, off, pv, 16);
      } else {
        if (USE_W) MGCMT_LEX_LOAD_AT(w_, off, pv, 0);
        MGCMT_LEX_LOAD_AT(s_, off, pv, 8);
        if (USE_S2) MGCMT_LEX_LOAD_AT(s2_, off, pv, 16);
      }
      MGCMT_LEX_LOAD_AT(f_, off, pf, 0);
    }
    const long step = ld_r < nr ? row_stride : 0;  // (rows nr + 1 .. are never used: the pointer rests on row nr)
    ldv += step;
    ldf += step;
    ++ld_r;
  };
  // ... and of the next row's edge record (lanes 0..3 matter; every lane loads).  Rows behind the left block's last one
  // read into the next block's records (inside the buffer; never used)
  const unsigned roff = (unsigned)(lane & 3) * 8u;
  const u64* ldr = rec_src + (long)i0 * 4;
  auto issue_rec = [&](auto plain, auto slot) __attribute__((always_inline)) {
    constexpr int SL = decltype(slot)::value;
    u64& r_ = Rv[SL];
    const u64* rbase = ldr;
    if (decltype(plain)::value) {
      r_ = load_granule(rbase + (lane & 3));
    } else {
      const unsigned ro = roff;  // (a use outside the asm statement: see above)
      MGCMT_LEX_LOAD_AT_SC1(r_, ro, rbase);
    }
    ldr += 4;
  };

  // chained sweeps: the progress words (rows finished, all stores acknowledged) of this block and of the two blocks of
  // the sweep ahead whose results this block reads: its own column block J and J + 1 (which only matters from its first
  // row on; the last block has no right neighbour)
  const int Jn = J + 1 < a.nblocks ? J + 1 : J;
  const int i0n = Jn * 64 - (nc - 1) > 0 ? Jn * 64 - (nc - 1) : 0;
  u64* const my_prog = a.progress + ((long)sw * a.nvec + q) * a.nblocks + J;
  const u64* const lead_prog = my_prog - (long)a.nvec * a.nblocks;  // (sweep > 0 only)
  int seen_own = 0, seen_next = 0;  // rows of the sweep ahead known to be finished and visible
  // edge records of the left block: the four granules of a row are read by lanes 0..3 (every lane loads, the address is
  // clamped), kRec rows before they are needed; the slow path asks again, visibly
  auto load_record = [&](int row) __attribute__((always_inline)) {
    const int rc = row < 0 ? 0 : (row > nr - 1 ? nr - 1 : row);
    return load_granule(rec_src + (long)rc * 4 + (lane & 3));
  };
  bool failed = false;
#ifdef MGCMT_LEXWAVE_DEBUG
  unsigned dbg_slow = 0;
  const u64 dbg_t0 = now_ticks();
#endif
  auto unpack = [&](u64 R, double& c1, double& c2) __attribute__((always_inline)) {  // false: the record is not complete yet
    const unsigned lo = (unsigned)R, hi = (unsigned)(R >> 32);  // (the register's two halves: half a double, the tag)
    c1 = __builtin_bit_cast(double, (u64)lane_word(lo, 2) | ((u64)lane_word(lo, 3) << 32));  // the left block's lane 63
    c2 = 0.0;
    if (!FIVE) c2 = __builtin_bit_cast(double, (u64)lane_word(lo, 0) | ((u64)lane_word(lo, 1) << 32));  // ... lane 62
    const u64 need = FIVE ? 0xCull : 0xFull;  // the tags of granules 2, 3 (and 0, 1): one compare, one vote
    return (vote_eq(hi, 1u) & need) == need;
  };
  auto wait_record = [&](int row, double& c1, double& c2) __attribute__((always_inline)) {  // the slow path: ask until the record is complete
#ifdef MGCMT_LEXWAVE_DEBUG
    ++dbg_slow;
#endif
    u64 t0 = 0;
    bool timing = false;
    while (true) {
      const u64 R = load_record(row);
      if (unpack(R, c1, c2)) return;
      if (!timing) {
        t0 = now_ticks();
        timing = true;
      }
      nap();
      if (now_ticks() - t0 > kTimeoutTicks || load_word(err_word) != 0u) {
        failed = true;
        c1 = 0.0;
        c2 = 0.0;
        return;
      }
    }
  };

  // row R of the sweep ahead is finished and visible (blocks J and, from its first row on, J + 1): the loads of row R may
  // be issued.  Polls (bounded) only when the values seen last do not cover R
  auto chase = [&](int R) __attribute__((always_inline)) {
    u64 t0 = 0;
    bool timing = false;
    while (seen_own <= R || (R >= i0n && seen_next <= R)) {
      seen_own = (int)uniform((int)load_granule(lead_prog));
      seen_next = (int)uniform((int)load_granule(lead_prog + (Jn - J)));
      if (seen_own > R && !(R >= i0n && seen_next <= R)) break;
      if (!timing) {
        t0 = now_ticks();
        timing = true;
      }
      nap();
      if (now_ticks() - t0 > kTimeoutTicks || load_word(err_word) != 0u) {
        failed = true;
        return;
      }
    }
  };

  // fill the pipeline: old values of rows i0 .. i0 + kDepth, records of rows i0 .. i0 + kRec - 1, with loads the compiler sees (it waits for them before the row loop's
  // first asm statement reads their registers; from then on nothing but the loop's own asm touches a slot)
  if (CH && sw > 0) chase(i0 + kSlots - 1 < nr ? i0 + kSlots - 1 : nr - 1);  // the sweep ahead has left the rows the prologue reads
  for_slots<kSlots>([&](auto sl) __attribute__((always_inline)) { issue_old(Checked<true>{}, sl); });
  for_slots<kRec>([&](auto sl) __attribute__((always_inline)) { issue_rec(Checked<true>{}, sl); });
  drain_loads();
  // The compiler does not know that the drain above completed its loads.  Left at that, it carries "slot k's load may be
  // pending" around the loop and puts its own s_waitcnt vmcnt(3 (kSlots - 1 - k)) before the first read of slot k in EVERY
  // trip — on the hardware's counter, which also counts the hand-issued loads, that drains the pipeline towards the end
  // of each trip.  One visible use per slot register here makes it wait once, now, for nothing.
#pragma unroll
  for (int d = 0; d < kSlots; ++d) {
    settle(Wv[d]);
    settle(Sv[d]);
    settle(S2v[d]);
    settle(Fv[d]);
    settle(Rv[d]);
  }

  // rows on which every lane and everything its stencil reaches lies inside the grid: jmin >= 1 and jmin + 64 <= nc - 1 (- 2)
  // (rows int_lo .. int_lo + int_span; a 9-point level's last row takes the general form; an empty range matches nothing)
  const int int_lo = J * 64 + 64 - nc + (FIVE ? 1 : 2);
  const int int_hi = J * 64 - 1 < (FIVE ? nr - 1 : nr - 2) ? J * 64 - 1 : (FIVE ? nr - 1 : nr - 2);
  const unsigned int_span = int_hi >= int_lo ? (unsigned)(int_hi - int_lo) : 0u;
  const int int_lo_eff = int_hi >= int_lo ? int_lo : (1 << 30);
  const double lane0 = lane == 0 ? 1.0 : 0.0;
  // running addresses of the row's stores: the window's first column on row i; the row's record (as seen from lane 60)
  char* vst = reinterpret_cast<char*>(v + (long)i0 * nc + (J * 64 - i0));
  char* rst = reinterpret_cast<char*>(my_rec + (long)i0 * 4 - 60);
  double prev = 0.0;            // new values of the previous row (this lane's column + 1 there)
  double c1p = 0.0, c2p = 0.0;  // the left block's edge values on the previous row
  if (J > 0 && i0 - 1 >= 0 && i0 - 1 <= left_last) wait_record(i0 - 1, c1p, c2p);

  // One row.  PH = the pipelines' phase, (i - i0) mod kSlots: the registers of row i are slot PH, those of row i + 1 slot
  // PH + 1, and the row fetched now takes slot PH over — a ROTATING register file: a value that is still in flight is never
  // moved (a register copy would read a register whose load has not landed).
  //
  // Forms of the row, chosen by wave-uniform tests:
  //   interior   every lane and everything its stencil reaches lies inside the grid: no masks, no tests;
  //   edge       the window crosses the grid's left or right edge (or, 9-point, holds its last column);
  //   general    the last row of a 9-point level (own-row coefficients differ): the scan over (p, q) pairs.
  // The EDGE rows are the sweep's critical path: with T(J, i) = max(T(J - 1, i), T(J, i - 1)) + t(J, i) every monotone path
  // through the (block, row) lattice has the same number of steps, and the diagonal on which each block holds column 0
  // (and the one on which it holds the last column) consists of edge rows only.  So the edge form carries only the masks
  // it cannot do without:
  //   - lanes left of the grid must contribute nothing to the lanes right of them: p = 0 there (their x is then 0 by
  //     itself: the scan has nothing to carry and no block works on that row further left);
  //   - the lanes on columns 0 and nc - 1 must not see what lies beyond: SW (9-point) resp. E, SE are masked, and NE —
  //     the lane's own value of the previous row — by zeroing x right of the grid (9-point only; the 5-point operator
  //     never reads a value right of the grid that the masks do not cover: garbage there stays there);
  //   - the last column of a 9-point level (its own diagonal and N / S coefficients, its own q) is patched AFTER the
  //     regular scan: x_lc = pc + qcol x_(lc - 1) with pc from p by the ratio of the two diagonals.
  const int lc_lo = J * 64 - (nc - 1);  // rows lc_lo .. lc_lo + 63 hold the last column in their window
  const double rho = inv_col * d_int;  // (k..c = rho k.. for every term whose coefficient the last column shares)
  const double dS = kSc - rho * kS, dN = kNc - rho * kN, dO = kOc - rho * kO;
  auto row_body = [&](auto ph, auto interior, int i, double& x, double& c1, double& c2) __attribute__((always_inline)) {
    constexpr int PH = decltype(ph)::value, NEXT = (PH + 1) % kSlots;
    constexpr bool IN = decltype(interior)::value;
    // (the slots are only ever read here: a masked value is a new value — a slot that one form of the row changed and the
    // other did not would meet itself in a phi, and the copies that costs read registers whose loads are in flight)
    const double &own = Wv[PH], &e = Sv[PH], &fv = Fv[PH];
    const double &sw = Wv[NEXT], &s = Sv[NEXT], &se = S2v[NEXT];
    const u64 rq_now = Rv[PH];
    // new values of row i-1: NE = this lane, N = lane - 1, NW = lane - 2 (the left block's edge beyond lane 0)
    const double n = fma(c1p, lane0, from_left(prev, lane));
    const double ne = prev;
    double nw = 0.0;
    if (!FIVE) nw = fma(c2p, lane0, from_left(n, lane));
    c1 = 0.0;
    c2 = 0.0;
    if (IN) {
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
      p = fma(q0, row_shr<1>(p, lane), p);
      p = fma(q2, row_shr<2>(p, lane), p);
      p = fma(q4, row_shr<4>(p, lane), p);
      p = fma(q8, row_shr<8>(p, lane), p);
      p = fma(Q15, bcast15(p, lane), p);
      p = fma(Q31, bcast31(p, lane), p);
      // (an interior row has columns left of its window: the left block works on it)
      if (!unpack(rq_now, c1, c2)) wait_record(i, c1, c2);
      x = fma(qpow, c1, p);
      store_value<CH>(vst + lane8, OWN ? fma(gamma, fv, x) : x);
    } else {
      const int j = J * 64 - i + lane;
      const bool valid = (unsigned)j < (unsigned)nc;
      const bool FAST = FIVE || i < nr - 1;  // (every row but a 9-point level's last one)
      double p, qmul;                        // x = p + qmul * (value left of lane 0)
      if (FAST) {
        const bool right = j < nc - 1;
        const double em = right ? e : 0.0;
        p = kF * fv;
        if (OWN) p = fma(kO, own, p);
        p = fma(kE, em, p);
        p = fma(kS, s, p);
        p = fma(kN, n, p);
        if (!FIVE) {
          const double sem = right ? se : 0.0, swm = j < 1 ? 0.0 : sw;
          p = fma(kSW, swm, p);
          p = fma(kSE, sem, p);
          p = fma(kNW, nw, p);
          p = fma(kNE, ne, p);
        }
        double pc = 0.0;
        const bool has_last_col = !FIVE && (unsigned)(i - lc_lo) <= 63u;  // the window holds column nc - 1 (wave-uniform)
        if (has_last_col) {
          pc = fma(dN, n, fma(dS, s, rho * p));
          if (OWN) pc = fma(dO, own, pc);
        }
        if (!valid) p = 0.0;
        p = fma(q0, row_shr<1>(p, lane), p);
        p = fma(q2, row_shr<2>(p, lane), p);
        p = fma(q4, row_shr<4>(p, lane), p);
        p = fma(q8, row_shr<8>(p, lane), p);
        p = fma(Q15, bcast15(p, lane), p);
        p = fma(Q31, bcast31(p, lane), p);
        qmul = qpow;
        if (i <= left_last) {  // (block 0: left_last = -1)
          if (!unpack(rq_now, c1, c2)) wait_record(i, c1, c2);
        }
        x = fma(qmul, c1, p);
        if (has_last_col) {
          const double xl = fma(c1, lane0, from_left(x, lane));
          const double xc = fma(qcol, xl, pc);
          if (j == nc - 1) x = xc;
        }
      } else {
        // the general form: every value outside the grid reads as zero
        const bool right = j + 1 >= 0 && j + 1 < nc, left = j - 1 >= 0 && j - 1 < nc;
        const double ownm = valid ? own : 0.0, fvm = valid ? fv : 0.0, sm = valid ? s : 0.0;
        const double em = right ? e : 0.0, sem = right ? se : 0.0, swm = left ? sw : 0.0;
        const bool last_col = !FIVE && j == nc - 1, last_row = !FIVE && i == nr - 1;
        // the centre-column / own-row coefficients change on the last column / row (Galerkin levels)
        // (assignments under `if`, not `?:` between the captured constants: a select of two captured variables keeps the
        // whole closure in scratch memory)
        double cN = cN_int, cS = cS_int, cW = cW_int, cE = cE_int, d = d_int, invd = inv_int;
        if (last_col) {
          cN = cN_col;
          cS = cS_col;
          d = d_col;
          invd = inv_col;
        }
        if (last_row) {
          cW = cW_row;
          cE = cE_row;
          d = d_row;
          invd = inv_row;
          if (last_col) {
            d = d_cor;
            invd = inv_cor;
          }
        }
        const double lower = fma(cNW, nw, fma(cN, n, cNE * ne));
        const double upper = fma(cE, em, fma(cSW, swm, fma(cS, sm, cSE * sem)));
        p = (alpha * d * ownm + beta * fvm - wU * upper - wL * lower) * invd;
        double qq = -wL * cW * invd;
        if (!valid) {
          p = 0.0;
          qq = 0.0;
        }
        // inclusive scan of the maps x -> p + q x over the lanes (`first` applied before `second`:
        // p = second.p + second.q * first.p, q = second.q * first.q)
#define MGCMT_LEX_STEP(FETCH, COND)             \
  {                                             \
    const double pp = FETCH(p, lane);           \
    const double pq = FETCH(qq, lane);          \
    if (COND) {                                 \
      p = fma(qq, pp, p);                       \
      qq = qq * pq;                             \
    }                                           \
  }
        MGCMT_LEX_STEP(row_shr<1>, (lane & 15) >= 1)
        MGCMT_LEX_STEP(row_shr<2>, (lane & 15) >= 2)
        MGCMT_LEX_STEP(row_shr<4>, (lane & 15) >= 4)
        MGCMT_LEX_STEP(row_shr<8>, (lane & 15) >= 8)
        MGCMT_LEX_STEP(bcast15, (lane & 16) != 0)
        MGCMT_LEX_STEP(bcast31, lane >= 32)
#undef MGCMT_LEX_STEP
        qmul = qq;
        if (i <= left_last) {
          if (!unpack(rq_now, c1, c2)) wait_record(i, c1, c2);
        }
        x = fma(qmul, c1, p);
      }
      if (!FIVE && !valid) x = 0.0;
      if (valid) store_value<CH>(vst + lane8, OWN ? fma(gamma, fv, x) : x);
    }
    {  // the row's edge record: {tag, half a double} granules, ONE store instruction (lanes 60..63; FIVE: 62, 63)
      const double src = quad_pairs(x, lane);  // lanes 60, 61: x62; lanes 62, 63: x63
      const u64 bits = __builtin_bit_cast(u64, src);
      const u64 word = (1ull << 32) | ((lane & 1) ? (bits >> 32) : (bits & 0xffffffffull));
      if (lane >= (FIVE ? 62 : 60)) store_granule(reinterpret_cast<u64*>(rst + lane8), word);
    }
  };
  auto row_step = [&](auto ph, int i) __attribute__((always_inline)) {
    constexpr int PH = decltype(ph)::value, NEXT = (PH + 1) % kSlots;
    const bool INTERIOR = (unsigned)(i - int_lo_eff) <= int_span;
    // the record of row i — and everything issued before it: the old values of rows i and i + 1 — has landed when at most
    // what was issued after it is in flight
    wait_loads<kWaitN>(Wv[PH], Sv[PH], Fv[PH], Wv[NEXT], Sv[NEXT], S2v[NEXT], S2v[PH], Rv[PH]);
    double x, c1, c2;
    if (INTERIOR) row_body(ph, Checked<true>{}, i, x, c1, c2);
    else row_body(ph, Checked<false>{}, i, x, c1, c2);
    vst += row_stride;
    rst += 32;
    prev = x;
    c1p = c1;
    c2p = c2;
    // refill: row i + kDepth + 1 takes this slot over.  LAST in the step: the slot's old values are dead by now, so the
    // loads land in the very registers the slot had (issued earlier, the old and the new value would be alive together, the
    // slots would rotate through registers, and the copies that restore them at the loop's latch would read registers whose
    // loads are still in flight)
    issue_rec(Checked<false>{}, Int<(PH + kRec) % kSlots>{});
    issue_old(Checked<false>{}, ph);
  };
  // ONE loop over the block's rows, kSlots rows per trip with compile-time phases; the last trip skips the rows behind i1
  // (a skipped step issues nothing, and nothing behind it consumes)
  for (int i = i0; i <= i1 && !failed; i += kSlots) {
    if (CH) {
      // Once per trip, OUTSIDE the unrolled row steps (a polling loop inside them made the compiler move the slots: the
      // audit).  Publish: rows < i are finished, and — after the drain — every store of them has been acknowledged.
      if (sw + 1 < nsw && i > i0 && ((i - i0) / kSlots) % kProgressTrips == 0) {
        drain_loads();
        store_granule(my_prog, (u64)i);
      }
      // Chase: this trip asks for rows up to i + 2 kSlots - 1 (rows behind the grid's last read the halo)
      if (sw > 0) chase(i + 2 * kSlots - 1 < nr ? i + 2 * kSlots - 1 : nr - 1);
    }
    for_slots<kSlots>([&](auto sl) __attribute__((always_inline)) {
      if (i + decltype(sl)::value <= i1) row_step(sl, i + decltype(sl)::value);
    });
  }
  if (CH && sw + 1 < nsw && !failed) {  // finished: every row, for good (the sweep behind never asks for more than nr rows)
    drain_loads();
    store_granule(my_prog, (u64)(nr + 1));
  }
#ifdef MGCMT_LEXWAVE_DEBUG
  if (lane == 0) {  // diagnostic build: per block {ticks (100 MHz), rows, slow-path entries} behind the two sync words
    unsigned* d = a.sync + 2 + 4 * (q * a.nblocks + J);
    const u64 dt = now_ticks() - dbg_t0;
    d[0] = (unsigned)dt;
    d[1] = (unsigned)(i1 - i0 + 1);
    d[2] = dbg_slow;
    d[3] = (unsigned)(dbg_t0 & 0xffffffffu);
  }
#endif
  if (failed && lane == 0) store_word(err_word, 1u);  // tell the host and release everyone behind this block
}

}  // namespace

bool lex_wave_supported(const KGrid& g, const KOp& op) {
  return g.coarsen_rows && (op.five_point || op.nine_const) && g.nc >= 16 && g.nr >= 16 && g.nr + g.nc < (1L << 30);
}

long lex_wave_blocks(const KGrid& g) { return (g.nr + g.nc - 1 + 63) / 64; }

// scratch: carry = lex_wave_carry() granules of 8 bytes (cleared here: the ticket, the records — valid when their tags are
// set —, the progress words), sync = the error word and the diagnostic build's words (not cleared here).  nsweeps > 1: that many sweeps chained in ONE launch (see k_lex_wave)
long lex_wave_carry(const KGrid& g, int k, int nsweeps) {
  return 8 + (long)nsweeps * k * lex_wave_blocks(g) * ((g.nr + kTerminalRows) * 4 + 1) + 64;  // (ticket, records, one progress word per block)
}

void launch_lex_wave(hipStream_t s, KGrid g, KOp op, KVec v, KVec f, const double* shifts, double alpha, double beta, double wU,
                     double wL, int k, double* carry, unsigned* sync, int nsweeps, double gamma) {
  LexWaveArgs a{};
  a.v = v.p;
  a.f = f.p;
  a.vstride = v.stride;
  a.nr = (int)g.nr;
  a.nc = (int)g.nc;
  a.nblocks = (int)lex_wave_blocks(g);
  a.five = op.five_point ? 1 : 0;
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
  a.gamma = gamma;
  a.ticket = reinterpret_cast<unsigned*>(carry);
  a.carry = reinterpret_cast<unsigned long long*>(carry) + 8;
  a.sync = sync;
  a.nsweeps = nsweeps < 1 ? 1 : nsweeps;
  a.rec_rows = (int)g.nr + kTerminalRows;
  a.carry_stride = (long)a.nblocks * a.rec_rows * 4;
  a.sweep_stride = a.carry_stride * k;
  a.nvec = k;
  a.progress = a.carry + (long)a.nsweeps * a.sweep_stride;
  // ONE memset: the ticket, the records, the progress words.  (The error word is not touched: a launch must not wipe out
  // what an earlier one reported before the host has looked.)
  (void)hipMemsetAsync(carry, 0, sizeof(unsigned long long) * (8 + (size_t)a.nsweeps * a.sweep_stride + (size_t)a.nsweeps * k * a.nblocks), s);
  const bool five = op.five_point, own = alpha != 0.0 || gamma != 0.0;
  if (a.nsweeps > 1) {
    const dim3 grid((unsigned)((a.nblocks + 2 * (a.nsweeps - 1)) * a.nsweeps * k));
    if (five && !own) hipLaunchKernelGGL((k_lex_wave<true, false, true>), grid, dim3(64), 0, s, a);
    else if (five) hipLaunchKernelGGL((k_lex_wave<true, true, true>), grid, dim3(64), 0, s, a);
    else if (!own) hipLaunchKernelGGL((k_lex_wave<false, false, true>), grid, dim3(64), 0, s, a);
    else hipLaunchKernelGGL((k_lex_wave<false, true, true>), grid, dim3(64), 0, s, a);
  } else {
    const dim3 grid((unsigned)(a.nblocks * k));
    if (five && !own) hipLaunchKernelGGL((k_lex_wave<true, false, false>), grid, dim3(64), 0, s, a);
    else if (five) hipLaunchKernelGGL((k_lex_wave<true, true, false>), grid, dim3(64), 0, s, a);
    else if (!own) hipLaunchKernelGGL((k_lex_wave<false, false, false>), grid, dim3(64), 0, s, a);
    else hipLaunchKernelGGL((k_lex_wave<false, true, false>), grid, dim3(64), 0, s, a);
  }
}

}  // namespace mgcmt
