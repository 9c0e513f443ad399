// Host-side types of libmgcmt_hip.so shared by plan.hip (hierarchy, single-GPU cycle, C-ABI) and sharded.hip
// (communicator, sharded cycle).  Not part of the ABI.
#pragma once
#include <map>
#include <string>
#include <vector>

#include "mgcmt_internal.h"

namespace mgcmt {

// tridiagonal factor on the host: lo, di, up concatenated, length 3n
struct Tri {
  int64_t n = 0;
  std::vector<double> a;
  double lo(int64_t i) const { return a[i]; }
  double di(int64_t i) const { return a[n + i]; }
  double up(int64_t i) const { return a[2 * n + i]; }
  double at(int64_t r, int64_t c) const {
    if (c == r - 1) return lo(r);
    if (c == r) return di(r);
    if (c == r + 1) return up(r);
    return 0.0;
  }
};

struct HostOp {
  int nterms = 0;
  std::vector<Tri> X, Y;  // per term
};

struct DevOp {
  KOp k{};
  std::vector<double*> owned;
};

struct BandState {
  KBand b{};
  double* inv = nullptr;  // explicit inverse per vector when the coarsest level has at most 1024 unknowns
  bool valid = false;
  int k = 0;
  std::vector<double> shifts;
};

struct Level {
  int64_t gr = 1, gc = 1;  // global rows / cols
  int64_t r0 = 0, nr = 1;  // local strip
  int64_t stride = 0;      // elements between vectors (halo rows included)
  int halo = kHalo;        // halo rows above and below every vector: kHalo on 2-D levels, 1 on 1-D levels (a "row" is the whole vector)
  double* base[4] = {nullptr, nullptr, nullptr, nullptr};  // allocation start per slot
  HostOp hA, hM;
  DevOp dA, dM;
  BandState band;
  KGrid grid() const { return KGrid{(long)nr, (long)gc, 0}; }
};

struct ShardComm;  // sharded.hip

}  // namespace mgcmt

struct mgcmt_plan {
  int dim = 1;
  int nvec = 1;
  int device = 0;
  int64_t g = 0, lowest = 0;
  std::vector<mgcmt::Level> levels;
  double* d_shifts = nullptr;   // [kMaxVec] current shifts
  double* d_zero = nullptr;     // [kMaxVec] zeros (apply without shift)
  double* d_partials = nullptr; // reduction scratch
  double* d_scalars = nullptr;  // [4*kMaxVec] reduction results
  double* d_rq = nullptr;       // Gram results of mgcmt_rayleigh_residual, one block per column
  double* d_rqstate = nullptr;  // scalars of the device-resident Rayleigh-quotient minimisation (kernels_rq.hip)
  double* d_mgs = nullptr;  // the blocked Gram-Schmidt's R^-1 and gate word (kernels_blas.hip)
  bool use_mgs_block = true;
  long mgs_block_min = 0;  // points per column from which a single plan takes the blocked form (0: wherever the one-workgroup kernel does not apply)
  double* d_rqhistory = nullptr;  // Rayleigh quotients recorded by mgcmt_rq_line_step (MGCMT_RQ_HISTORY numbers)
  std::vector<double> h_shifts;
  bool has_mass = false;
  bool use_fused = true;
  bool use_tail = true;   // levels of at most 32 x 32 points as one launch (kernels_tail.hip)
  bool use_tail_dense = true;  // ... and that launch as ONE dense product with the tail's matrix (formed once per shift set)
  struct TailMatrix {
    double* mt = nullptr;   // [k][n * n]
    int capacity = 0;       // vectors allocated
    long n = 0;
    int lt = -1, kind = -1, nu = -1, k = 0;
    double omega = 0.0;
    std::vector<double> shifts;
    bool valid = false;
  } tailmat;
  bool use_recompute = true;  // down-leg passes skip storing V', up-leg passes recompute it (fused_kernel.h)
  bool force_recompute = false;  // ... on every fused level, not only the bandwidth-bound ones (tests)
  // HIP-graph replay of whole cycles (mgcmt_vcycle): the launch sequence of a cycle is fixed by its
  // parameters and by which of the two buffers of every level currently is "V", so it is captured once per
  // such state and replayed; small grids are launch-latency-bound otherwise.
  bool use_graph = true;
  long fused_rows = 0;            // tuning: rows per wave chunk of the fused passes, 0 = automatic
  mgcmt::ShardComm* comm = nullptr;  // communicator of a sharded plan (sharded.hip), owned
  int use_lex_wave = 1;           // lexicographic sweeps on the whole chip where the level is covered: 1 = skewed column blocks with a scan per row (kernels_lexwave.hip), 2 = row bands swept as a wavefront (kernels_lexband.hip); 0 = one workgroup
  double* lex_carry = nullptr;    // its scratch (grown on demand)
  unsigned* lex_sync = nullptr;
  size_t lex_carry_doubles = 0, lex_sync_words = 0;
  bool lex_chain = true;          // consecutive Gauss-Seidel sweeps of a smoothing step run chained in one launch (kernels_lexwave.hip)
  bool lex_wave_used = false;     // the error word of lex_sync has not been looked at since the last sweep
  hipStream_t capture_stream = nullptr;
  struct CycleGraph {
    hipGraphExec_t exec = nullptr;
    std::vector<double*> post_state;  // base pointers of slots V and T of every level after the cycle
    bool lex_wave = false;            // the captured body runs the lexicographic wave pipeline (its error word must be checked after a replay)
  };
  std::map<std::string, CycleGraph> graphs;
  std::map<std::string, int> cycle_seen;
  void graphs_invalidate() {  // a captured launch sequence is only valid for the options it was captured under
    for (auto& g : graphs)
      if (g.second.exec) (void)hipGraphExecDestroy(g.second.exec);
    graphs.clear();
    cycle_seen.clear();
  }

  mgcmt::KGrid kgrid(int l) const {
    using namespace mgcmt;
    KGrid kg = levels[l].grid();
    kg.coarsen_rows = dim == 2 ? 1 : 0;
    return kg;
  }
  mgcmt::KVec kvec(int l, int slot, int vec = 0) const {
    using namespace mgcmt;
    const Level& L = levels[l];
    return KVec{L.base[slot] + (long)L.halo * L.gc + (long)vec * L.stride, (long)L.stride};
  }
  long interior(int l) const { return (long)levels[l].nr * levels[l].gc; }
};


namespace mgcmt {
// helpers of plan.hip used by sharded.hip
int fail(int code, const std::string& msg);
int ensure_slot(mgcmt_plan* p, int l, int slot);
int post_launch();
bool fused_level(const mgcmt_plan* p, int l, int kind);
int pass_sweeps(const mgcmt_plan* p, int l, int kind, int left);
// halo rows of level l that a sharded cycle exchanges and its passes read (<= the level's halo): 8 behind a 5-point
// operator (at most eight stages per pass), 10 behind a 9-point one (two four-colour sweeps + the restriction: nine)
int exchanged_rows(const mgcmt_plan* p, int l);
// one fused pass V -> T (then swapped); [out_lo, out_hi) = the rows produced (default: the whole strip), swap = false
// leaves the buffer roles alone (the caller issues the other row ranges of the same pass and swaps once)
int fused_pass(mgcmt_plan* p, int l, int kind, int nsweep, double omega, int mode, int k, hipStream_t s, int npre = 0,
               long out_lo = 0, long out_hi = -1, bool swap = true, long out_lo2 = 0, long out_hi2 = 0);
void comm_release(mgcmt_plan* p);  // sharded.hip: frees p->comm
// transfer.hip: a whole vector between caller memory and the device through the pinned ring; completed on return
int transfer(int device, bool upload, void* dev, void* host, size_t bytes, hipStream_t stream);
}  // namespace mgcmt

#define MG_HIP(expr)                                                                                    \
  do {                                                                                                  \
    hipError_t e_ = (expr);                                                                             \
    if (e_ != hipSuccess)                                                                               \
      return mgcmt::fail(MGCMT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));             \
  } while (0)

#define MG_TRY(expr)          \
  do {                        \
    int rc_ = (expr);         \
    if (rc_ != MGCMT_OK) return rc_; \
  } while (0)
