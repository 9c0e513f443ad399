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
// Hand-off (MI355X_MICROARCH.md, inter-workgroup visibility): edge values are stored as 8-byte agent-scope atomics
// (write-through, sc1), every R rows the storing wave drains its stores (s_waitcnt vmcnt(0)) and one lane stores the
// progress word (agent-scope atomic); the consumer polls that word (relaxed, bounded, with s_sleep) and reads the
// edge values with agent-scope atomic loads (sc1: they bypass its CU's L1).  Block numbers are handed out by a
// ticket counter, so a block only ever waits for one that has already started; every spin is bounded by a timeout
// that raises an error word instead of hanging the GPU.
//
// Covers the constant-coefficient operators (the scaled / shifted Laplacian and its Galerkin coarsenings, whose
// factors are Toeplitz except for their last diagonal entry); other operators and small grids keep kernels_lex.hip.
#include "mgcmt_internal.h"

namespace mgcmt {

namespace {

#ifndef MGCMT_LEXWAVE_PUBLISH
#define MGCMT_LEXWAVE_PUBLISH 8  // rows between two publications of a block's progress
#endif
constexpr int kPublish = MGCMT_LEXWAVE_PUBLISH;
constexpr int kDepth = 3;  // rows of old values in flight ahead of the row being processed

struct LexWaveArgs {
  double* v;
  const double* f;
  long vstride;
  int nr, nc, nblocks;
  double c[3][3];                    // interior coefficients [di + 1][dj + 1]
  double crow[3], ccol[3], ccorner;  // last row: own-row coefficients (W, C, E); last column: centre column (N, C, S)
  const double* shifts;
  double alpha, beta, wU, wL;
  double* carry;       // [vector][block][row][2]: new values of the block's lanes 63 and 62
  unsigned* sync;      // [0] ticket, [1] error, [2 + vector*nblocks + block] rows completed
  long carry_stride;   // doubles per vector
};

#if defined(__HIP_DEVICE_COMPILE__)
template <int CTRL>
__device__ __forceinline__ double dpp(double v) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)u, CTRL, 0xf, 0xf, true);
  const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTRL, 0xf, 0xf, true);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
template <int D>
__device__ __forceinline__ double row_shr(double v, int) { return dpp<0x110 + D>(v); }    // lane - D inside rows of 16
__device__ __forceinline__ double bcast15(double v, int) { return dpp<0x142>(v); }         // lane 15 of the previous row of 16
__device__ __forceinline__ double bcast31(double v, int) { return dpp<0x143>(v); }         // lane 31
__device__ __forceinline__ double from_left(double v, int) { return dpp<0x138>(v); }       // lane - 1 (lane 0: 0)
__device__ __forceinline__ double from_right(double v, int) { return dpp<0x130>(v); }      // lane + 1 (lane 63: 0)
__device__ __forceinline__ double lane_value(double v, int k) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, k);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), k);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ unsigned load_word(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void store_word(unsigned* p, unsigned x) { __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double load_shared(const double* p) {
  const unsigned long long u = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return __builtin_bit_cast(double, u);
}
__device__ __forceinline__ void store_shared(double* p, double x) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), __builtin_bit_cast(unsigned long long, x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ unsigned long long now_ticks() { return wall_clock64(); }  // 100 MHz
__device__ __forceinline__ void nap() { __builtin_amdgcn_s_sleep(2); }
#else
// host-side stand-ins (the emulator runs one workgroup at a time, in block order: a block never has to wait)
template <int D>
__device__ __forceinline__ double row_shr(double v, int lane) { const double r = __shfl_up(v, D); return (lane & 15) >= D ? r : 0.0; }
__device__ __forceinline__ double bcast15(double v, int lane) { return __shfl(v, ((lane & ~15) - 1) & 63); }
__device__ __forceinline__ double bcast31(double v, int) { return __shfl(v, 31); }
__device__ __forceinline__ double from_left(double v, int lane) { const double r = __shfl_up(v, 1); return lane >= 1 ? r : 0.0; }
__device__ __forceinline__ double from_right(double v, int lane) { const double r = __shfl_down(v, 1); return lane <= 62 ? r : 0.0; }
__device__ __forceinline__ double lane_value(double v, int k) { return __shfl(v, k); }
__device__ __forceinline__ unsigned load_word(const unsigned* p) { return *p; }
__device__ __forceinline__ void store_word(unsigned* p, unsigned x) { *p = x; }
__device__ __forceinline__ double load_shared(const double* p) { return *p; }
__device__ __forceinline__ void store_shared(double* p, double x) { *p = x; }
__device__ __forceinline__ void drain_stores() {}
__device__ __forceinline__ unsigned long long now_ticks() { return 0; }
__device__ __forceinline__ void nap() {}
#endif

constexpr unsigned long long kTimeoutTicks = 200000000ull;  // 2 s of the 100 MHz counter: a stuck pipeline gives up

__global__ void __launch_bounds__(64) k_lex_wave(LexWaveArgs a) {
  const int lane = threadIdx.x;
  // block number = order of arrival: whoever this block waits for has started before it
  unsigned ticket = 0;
  if (lane == 0) ticket = atomicAdd(&a.sync[0], 1u);
  ticket = (unsigned)__shfl((int)ticket, 0);
  const int q = (int)(ticket / (unsigned)a.nblocks);
  const int J = (int)(ticket % (unsigned)a.nblocks);
  const int nr = a.nr, nc = a.nc;
  double* __restrict__ v = a.v + (long)q * a.vstride;
  const double* __restrict__ f = a.f + (long)q * a.vstride;
  double* my_carry = a.carry + (long)q * a.carry_stride + (long)J * nr * 2;
  const double* left_carry = my_carry - (long)nr * 2;
  unsigned* my_progress = a.sync + 2 + (long)q * a.nblocks + J;
  const unsigned* left_progress = my_progress - 1;

  const int i0 = J * 64 - (nc - 1) > 0 ? J * 64 - (nc - 1) : 0;  // first row with a column of this block inside the grid
  const int i1 = J * 64 + 63 < nr - 1 ? J * 64 + 63 : nr - 1;    // last one
  const int left_last = J > 0 ? (J * 64 - 1 < nr - 1 ? J * 64 - 1 : nr - 1) : -1;  // last row the left block works on
  const bool publish = J + 1 < a.nblocks;

  const double mu = a.shifts[q];
  const double alpha = a.alpha, beta = a.beta, wU = a.wU, wL = a.wL;
  // coefficient classes: interior, last column, last row, corner
  const double cNW = a.c[0][0], cNE = a.c[0][2], cSW = a.c[2][0], cSE = a.c[2][2];
  const double d_int = a.c[1][1] - mu, d_col = a.ccol[1] - mu, d_row = a.crow[1] - mu, d_cor = a.ccorner - mu;
  const double inv_int = 1.0 / d_int, inv_col = 1.0 / d_col, inv_row = 1.0 / d_row, inv_cor = 1.0 / d_cor;

  // old values: window of row r = v[r][J*64 + lane - r] (this block's columns on that row) and, in lanes 0 and 1, the
  // two columns to its right.  Rows nr.. are the zero halo rows; columns outside the grid read as zero.
  auto load_window = [&](int r, double& w, double& t, double& fr) {
    const int rr = r < nr ? r : nr;  // rows beyond the grid: the (zero) halo row
    const int jw = J * 64 + lane - r;
    const int jt = J * 64 + 64 + lane - r;
    const int jwc = jw < 0 ? 0 : (jw > nc - 1 ? nc - 1 : jw);
    const int jtc = jt < 0 ? 0 : (jt > nc - 1 ? nc - 1 : jt);
    const double wv = v[(long)rr * nc + jwc];
    const double tv = lane < 2 ? v[(long)rr * nc + jtc] : 0.0;
    const double fv = f[(long)rr * nc + jwc];
    w = (jw >= 0 && jw < nc) ? wv : 0.0;
    t = (lane < 2 && jt >= 0 && jt < nc) ? tv : 0.0;
    fr = fv;
  };

  // edge values of the left block, 64 rows at a time: lane r holds row cbase + r
  int cbase = 0, cvalid = 0;
  double cb1 = 0.0, cb2 = 0.0;
  bool failed = false;
  auto fetch_carries = [&](int row) {
    // wait until the left block has published `row`, then take what is there (up to 64 rows)
    unsigned done = load_word(left_progress);
    if (done < (unsigned)row + 1u) {
      const unsigned long long t0 = now_ticks();
      while (true) {
        nap();
        done = load_word(left_progress);
        if (done >= (unsigned)row + 1u) break;
        if (now_ticks() - t0 > kTimeoutTicks || load_word(a.sync + 1) != 0u) {
          failed = true;
          break;
        }
      }
    }
    int upto = (int)done < left_last + 1 ? (int)done : left_last + 1;  // rows [row, upto) are there
    if (failed) upto = row;
    cbase = row;
    cvalid = upto - row < 64 ? upto - row : 64;
    const int r = row + lane;
    const bool have = lane < cvalid;
    const int rc = have ? r : row;
    const double x1 = load_shared(left_carry + (long)rc * 2);
    const double x2 = load_shared(left_carry + (long)rc * 2 + 1);
    cb1 = have ? x1 : 0.0;
    cb2 = have ? x2 : 0.0;
  };
  auto carries_of = [&](int row, double& c1, double& c2) {
    if (J == 0 || row < 0 || row > left_last) {  // no left block there: Dirichlet zero
      c1 = 0.0;
      c2 = 0.0;
      return;
    }
    if (!(row >= cbase && row < cbase + cvalid)) fetch_carries(row);
    const int k = row - cbase;
    c1 = lane_value(cb1, k);
    c2 = lane_value(cb2, k);
  };

  if (i0 > i1) {  // nothing inside the grid (cannot happen for nblocks = ceil((nr + nc - 1) / 64); kept as a guard)
    if (lane == 0) store_word(my_progress, (unsigned)nr);
    return;
  }

  // pipeline registers: rows i .. i + kDepth of old values
  double wn[kDepth + 1], tl[kDepth + 1], fr[kDepth + 1];
#pragma unroll
  for (int d = 0; d <= kDepth; ++d) load_window(i0 + d, wn[d], tl[d], fr[d]);

  double prev = 0.0;  // new values of the previous row (this lane's column + 1 there)
  double c1p, c2p;    // the left block's edge values on the previous row
  carries_of(i0 - 1, c1p, c2p);
  int since_publish = 0;

  for (int i = i0; i <= i1; ++i) {
    const int j = J * 64 + lane - i;
    const bool valid = j >= 0 && j < nc;
    const bool last_col = j == nc - 1, last_row = i == nr - 1;
    // new values of row i-1: NE = this lane, N = lane - 1, NW = lane - 2 (the left block's edge beyond lane 0)
    double n = from_left(prev, lane);
    if (lane == 0) n = c1p;
    double nw = from_left(n, lane);
    if (lane == 0) nw = c2p;
    const double ne = prev;
    // old values: own row (E = lane + 1) and the row below (SW = this lane of its window, S = lane + 1, SE = lane + 2)
    const double own = wn[0];
    double e = from_right(wn[0], lane);
    const double t00 = lane_value(tl[0], 0);
    if (lane == 63) e = t00;
    const double sw = wn[1];
    double s = from_right(wn[1], lane);
    const double t10 = lane_value(tl[1], 0), t11 = lane_value(tl[1], 1);
    if (lane == 63) s = t10;
    double se = from_right(s, lane);
    if (lane == 63) se = t11;
    const double cN = last_col ? a.ccol[0] : a.c[0][1], cS = last_col ? a.ccol[2] : a.c[2][1];
    const double cW = last_row ? a.crow[0] : a.c[1][0], cE = last_row ? a.crow[2] : a.c[1][2];
    const double d = last_row ? (last_col ? d_cor : d_row) : (last_col ? d_col : d_int);
    const double invd = last_row ? (last_col ? inv_cor : inv_row) : (last_col ? inv_col : inv_int);
    const double lower = fma(cNW, nw, fma(cN, n, cNE * ne));
    const double upper = fma(cE, e, fma(cSW, sw, fma(cS, s, cSE * se)));
    double p = (alpha * d * own + beta * fr[0] - wU * upper - wL * lower) * invd;
    double qq = -wL * cW * invd;
    if (!valid) {
      p = 0.0;
      qq = 0.0;
    }
    // refill the pipeline while the scan runs
#pragma unroll
    for (int dd = 0; dd < kDepth; ++dd) {
      wn[dd] = wn[dd + 1];
      tl[dd] = tl[dd + 1];
      fr[dd] = fr[dd + 1];
    }
    load_window(i + kDepth + 1, wn[kDepth], tl[kDepth], fr[kDepth]);
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
    // the value left of lane 0 on this row: the left block's lane 63
    double c1, c2;
    carries_of(i, c1, c2);
    const double x = valid ? fma(qq, c1, p) : 0.0;
    if (valid) v[(long)i * nc + j] = x;
    if (publish) {
      if (lane >= 62) store_shared(my_carry + (long)i * 2 + (63 - lane), x);
      if (++since_publish == kPublish || i == i1) {
        since_publish = 0;
        drain_stores();
        if (lane == 0) store_word(my_progress, i == i1 ? (unsigned)nr : (unsigned)i + 1u);
      }
    }
    prev = x;
    c1p = c1;
    c2p = c2;
    if (failed) break;
  }
  if (failed && lane == 0) {
    store_word(a.sync + 1, 1u);  // tell the host and release everyone behind this block
    store_word(my_progress, (unsigned)nr);
  }
}

}  // namespace

bool lex_wave_supported(const KGrid& g, const KOp& op) {
  return g.coarsen_rows && (op.five_point || op.nine_const) && g.nc >= 128 && g.nr >= 64 && g.nr + g.nc < (1L << 30);
}

long lex_wave_blocks(const KGrid& g) { return (g.nr + g.nc - 1 + 63) / 64; }

// scratch: carry = k * blocks * nr * 2 doubles, sync = (2 + k * blocks) words; the sync words are cleared here
void launch_lex_wave(hipStream_t s, KGrid g, KOp op, KVec v, KVec f, const double* shifts, double alpha, double beta, double wU,
                     double wL, int k, double* carry, unsigned* sync) {
  LexWaveArgs a{};
  a.v = v.p;
  a.f = f.p;
  a.vstride = v.stride;
  a.nr = (int)g.nr;
  a.nc = (int)g.nc;
  a.nblocks = (int)lex_wave_blocks(g);
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
  a.carry = carry;
  a.sync = sync;
  a.carry_stride = (long)a.nblocks * g.nr * 2;
  (void)hipMemsetAsync(sync, 0, sizeof(unsigned) * (2 + (size_t)k * a.nblocks), s);
  hipLaunchKernelGGL(k_lex_wave, dim3((unsigned)(a.nblocks * k)), dim3(64), 0, s, a);
}

}  // namespace mgcmt
