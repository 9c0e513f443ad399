// Sharded V-cycle: communicator and cycle driver of a row-strip plan (north_star: "the finest grid
// levels domain-decompose across the 8 GPUs of one node with RCCL neighbour halo exchange over xGMI, coarser levels
// agglomerating").  Host code only.
//
// Transport 1 — RCCL, used directly from this library (no Python on the data path): one communicator per strip plan,
// ncclSend/ncclRecv pairs in one group per exchange and one ncclAllGather per cycle, enqueued on HIP streams with no
// host synchronisation anywhere in a cycle.  librccl is loaded with dlopen at mgcmt_comm_init, so single-GPU users
// of libmgcmt_hip.so do not need it; the handful of declarations below restate rccl.h (ROCm 7.2:
// /opt/rocm/include/rccl/rccl.h:40-43,187,220,260,339,448,467,611,678,700,722,923,933).
// Transport 2 — callbacks supplied by the host program (gloo / MPI / anything): the library synchronises its stream
// and hands over device pointers.  The CPU rehearsal of the N > 1 path (tests/test_distributed.py, gloo) and ranks
// sharing one GPU go through it, so the cycle below is the ONE implementation of the schedule.
//
// Overlap: the boundary rows of a strip are produced by their own small launches FIRST; their exchange then runs on
// a second stream beside the launch that produces the interior rows, and the next pass waits for both.
#include <dlfcn.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "plan_internal.h"

using namespace mgcmt;

namespace {

// ---- the slice of the RCCL API this file uses ---------------------------------------------------------------
struct RcclUniqueId {
  char internal[MGCMT_UNIQUE_ID_BYTES];
};
typedef void* RcclComm;
constexpr int kRcclSuccess = 0, kRcclSum = 0, kRcclFloat64 = 8;

struct RcclApi {
  void* handle = nullptr;
  int (*GetUniqueId)(RcclUniqueId*) = nullptr;
  int (*CommInitRank)(RcclComm*, int, RcclUniqueId, int) = nullptr;
  int (*CommDestroy)(RcclComm) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Send)(const void*, size_t, int, int, RcclComm, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, RcclComm, hipStream_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, RcclComm, hipStream_t) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, RcclComm, hipStream_t) = nullptr;
  std::string error;
};

RcclApi* rccl() {
  static RcclApi api;
  static bool tried = false;
  if (tried) return api.handle ? &api : nullptr;
  tried = true;
  // a librccl already in the process (e.g. PyTorch's own copy, same SONAME) is reused by the first name
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names) {
    api.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (api.handle) break;
  }
  if (!api.handle) {
    api.error = std::string("librccl not found: ") + (dlerror() ? dlerror() : "?");
    return nullptr;
  }
  bool ok = true;
  auto sym = [&](const char* name) {
    void* f = dlsym(api.handle, name);
    if (!f) {
      ok = false;
      api.error = std::string("librccl lacks ") + name;
    }
    return f;
  };
  api.GetUniqueId = (int (*)(RcclUniqueId*))sym("ncclGetUniqueId");
  api.CommInitRank = (int (*)(RcclComm*, int, RcclUniqueId, int))sym("ncclCommInitRank");
  api.CommDestroy = (int (*)(RcclComm))sym("ncclCommDestroy");
  api.GetErrorString = (const char* (*)(int))sym("ncclGetErrorString");
  api.GroupStart = (int (*)())sym("ncclGroupStart");
  api.GroupEnd = (int (*)())sym("ncclGroupEnd");
  api.Send = (int (*)(const void*, size_t, int, int, RcclComm, hipStream_t))sym("ncclSend");
  api.Recv = (int (*)(void*, size_t, int, int, RcclComm, hipStream_t))sym("ncclRecv");
  api.AllGather = (int (*)(const void*, void*, size_t, int, RcclComm, hipStream_t))sym("ncclAllGather");
  api.AllReduce = (int (*)(const void*, void*, size_t, int, int, RcclComm, hipStream_t))sym("ncclAllReduce");
  if (!ok) {
    dlclose(api.handle);
    api.handle = nullptr;
    return nullptr;
  }
  return &api;
}

#define MG_RCCL(api, expr)                                                                         \
  do {                                                                                             \
    int r_ = (expr);                                                                               \
    if (r_ != kRcclSuccess)                                                                        \
      return fail(MGCMT_ERR_HIP, std::string(#expr) + ": " + ((api)->GetErrorString ? (api)->GetErrorString(r_) : "rccl error")); \
  } while (0)

}  // namespace

namespace {

// ---- which stream runs BESIDE the cycle's stream? -------------------------------------------------------------------
// HIP deals a handful of hardware queues to the streams of a process round-robin; two streams on one queue run in
// order, and then the exchange lane waits behind the interior launch it is meant to run beside (rocprofv3 traces of
// round 3: 2.44 ms per emulated rank share with both lanes on one queue, 2.13 ms on two).  No API tells the queue of a
// stream, so the communicator tries: a one-wave kernel that spins for a while on the cycle's stream, a trivial kernel on
// the candidate — the candidate is on another queue if its kernel finishes while the spinner still runs.
__global__ void k_spin(long ticks_100mhz, int* sink) {
  int i = 0;
#if defined(__HIP_DEVICE_COMPILE__)
  const long t0 = (long)wall_clock64();
  for (; i < 400000; ++i) {  // (bounded: every wave reaches the exit whatever the clock does)
    if ((long)wall_clock64() - t0 > ticks_100mhz) break;
    __builtin_amdgcn_s_sleep(16);
  }
#endif
  if (threadIdx.x == 0) sink[0] = i;
}
__global__ void k_touch(int* sink) {
  if (threadIdx.x == 0) sink[1] = 1;
}

// true when work on `cand` overtakes a running kernel on `main`
bool runs_beside(hipStream_t main, hipStream_t cand, int* d_sink) {
  hipEvent_t em = nullptr, ec = nullptr;
  if (hipEventCreateWithFlags(&em, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&ec, hipEventDisableTiming) != hipSuccess) {
    if (em) (void)hipEventDestroy(em);
    (void)hipGetLastError();
    return false;
  }
  bool beside = false;
  if (hipStreamSynchronize(main) == hipSuccess && hipStreamSynchronize(cand) == hipSuccess) {
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, main, 300000L /* 3 ms of the 100 MHz clock */, d_sink);
    (void)hipEventRecord(em, main);
    hipLaunchKernelGGL(k_touch, dim3(1), dim3(64), 0, cand, d_sink);
    (void)hipEventRecord(ec, cand);
    if (hipEventSynchronize(ec) == hipSuccess) beside = hipEventQuery(em) == hipErrorNotReady;
    (void)hipStreamSynchronize(main);
  }
  (void)hipGetLastError();
  (void)hipEventDestroy(em);
  (void)hipEventDestroy(ec);
  return beside;
}

}  // namespace

namespace mgcmt {

// words of the reduction buffer: the 2 * kMaxVec of the sums the API takes (mgcmt_allreduce_sum), the 78 entries of a Gram matrix
constexpr int kRedWords = 128;

struct ShardComm {
  int rank = 0, nranks = 1;
  RcclComm nccl = nullptr;  // transport 1
  mgcmt_p2p_fn p2p = nullptr;  // transport 2
  mgcmt_allgather_fn allgather = nullptr;
  mgcmt_allreduce_fn allreduce = nullptr;
  void* user = nullptr;
  hipStream_t comm_stream = nullptr;  // exchanges of boundary rows run here, beside the interior launch
  hipEvent_t ev_boundary = nullptr, ev_done = nullptr, ev_edges = nullptr;
  bool pending = false;   // an exchange is in flight on comm_stream (ev_done follows it): whatever reads halo rows next waits for it
  bool edges_pending = false;  // a pass's edge rows are being produced on comm_stream (ev_edges follows them): whatever reads them next waits
  bool overlap = true;    // RCCL transport only
  int split = 1;          // boundary rows first (both transports): 1 on strips of >= 2^22 points, 2 always (tests), 0 never
  bool self_ring = false; // one-rank self-test: the rank is its own neighbour above and below in every exchange
  bool lane_checked = false;  // comm_stream has been tried against the cycle's stream (pick_exchange_stream)
  hipStream_t lane_for = nullptr;
  int emulate_of = 0;     // > 1 on a ONE-rank communicator in self-ring mode: the plan is rank R's strip of an N-rank job (timing rehearsal)
  double* d_red = nullptr;  // [kRedWords] reduction results (inner products of the sharded Gram-Schmidt, mgcmt_allreduce_sum)
  std::vector<double> h_red;
};

void comm_release(mgcmt_plan* p) {
  if (!p || !p->comm) return;
  ShardComm* c = p->comm;
  if (c->nccl && rccl()) (void)rccl()->CommDestroy(c->nccl);
  if (c->comm_stream) (void)hipStreamDestroy(c->comm_stream);
  if (c->ev_boundary) (void)hipEventDestroy(c->ev_boundary);
  if (c->ev_done) (void)hipEventDestroy(c->ev_done);
  if (c->ev_edges) (void)hipEventDestroy(c->ev_edges);
  if (c->d_red) (void)hipFree(c->d_red);
  delete c;
  p->comm = nullptr;
}

}  // namespace mgcmt

namespace {

constexpr int kRing = 0x100;       // mgcmt_halo_exchange: treat a ONE-rank chain as a ring (transport self-test)
// Rows of a strip's edge produced first by a split pass: what the neighbour's next pass reads of V (the level's
// exchanged rows) and, restricted, of the coarse right-hand side (twice the coarse level's exchanged rows).
long boundary_rows(const mgcmt_plan* p, int l, bool sends_f) {
  // (+ 2: the interior launch — everything but these rows — then reads no halo row at all, whatever the pass fuses: it
  // can run before the previous exchange has arrived)
  long b = exchanged_rows(p, l) + 2;
  if (sends_f && 2L * exchanged_rows(p, l + 1) > b) b = 2L * exchanged_rows(p, l + 1);
  return (b + 1) & ~1L;
}

struct Msg {
  double* ptr;
  size_t count;
  int peer;
  bool send;
};

int check_comm(const mgcmt_plan* p) {
  if (!p) return fail(MGCMT_ERR_INVALID, "null plan");
  if (!p->comm) return fail(MGCMT_ERR_INVALID, "plan has no communicator (mgcmt_comm_init / mgcmt_comm_init_external)");
  return MGCMT_OK;
}

// boundary rows out / halo rows in of vectors 0..k-1 of (level, slot) for the chain neighbours: the level's exchanged
// rows (exchanged_rows: 8 behind a 5-point operator, 10 behind a 9-point one), next to the strip
int halo_msgs(mgcmt_plan* p, int l, int slot, bool ring, int k, std::vector<Msg>* out) {
  const ShardComm* c = p->comm;
  MG_TRY(ensure_slot(p, l, slot));
  const Level& L = p->levels[l];
  if (L.nr == L.gr && !ring) return MGCMT_OK;  // not a strip level: nothing to exchange
  const long hx = exchanged_rows(p, l);
  if (L.nr < hx) return fail(MGCMT_ERR_INVALID, "strip has fewer rows than the halo");
  const size_t cnt = (size_t)hx * L.gc;
  for (int q = 0; q < k; ++q) {
    double* v = p->kvec(l, slot, q).p;
    double* top = v;
    double* bottom = v + (long)(L.nr - hx) * L.gc;
    double* halo_up = v - (long)hx * L.gc;
    double* halo_dn = v + (long)L.nr * L.gc;
    if (ring) {  // one rank, itself above and below: messages to one peer match in order
      out->push_back({bottom, cnt, 0, true});
      out->push_back({top, cnt, 0, true});
      out->push_back({halo_up, cnt, 0, false});
      out->push_back({halo_dn, cnt, 0, false});
      continue;
    }
    const int up = c->rank - 1, down = c->rank + 1;
    if (up >= 0) {
      out->push_back({top, cnt, up, true});
      out->push_back({halo_up, cnt, up, false});
    }
    if (down < c->nranks) {
      out->push_back({bottom, cnt, down, true});
      out->push_back({halo_dn, cnt, down, false});
    }
  }
  return MGCMT_OK;
}

int run_msgs(mgcmt_plan* p, const std::vector<Msg>& msgs, hipStream_t s) {
  ShardComm* c = p->comm;
  if (msgs.empty()) return MGCMT_OK;
  if (c->nccl) {
    RcclApi* api = rccl();
    MG_RCCL(api, api->GroupStart());
    // an error inside the group must not leave it open (every later RCCL call of the process would queue behind it)
    int bad = kRcclSuccess;
    for (const Msg& m : msgs) {
      bad = m.send ? api->Send(m.ptr, m.count, kRcclFloat64, m.peer, c->nccl, s) : api->Recv(m.ptr, m.count, kRcclFloat64, m.peer, c->nccl, s);
      if (bad != kRcclSuccess) break;
    }
    const int end = api->GroupEnd();
    if (bad != kRcclSuccess) return fail(MGCMT_ERR_HIP, std::string("ncclSend/ncclRecv: ") + (api->GetErrorString ? api->GetErrorString(bad) : "rccl error"));
    MG_RCCL(api, end);
    return MGCMT_OK;
  }
  if (!c->p2p) return fail(MGCMT_ERR_INVALID, "communicator has no point-to-point transport");
  std::vector<mgcmt_p2p_op> ops(msgs.size());
  for (size_t i = 0; i < msgs.size(); ++i) ops[i] = mgcmt_p2p_op{msgs[i].ptr, (int64_t)msgs[i].count, msgs[i].peer, msgs[i].send ? 1 : 0};
  MG_HIP(hipStreamSynchronize(s));  // the callback moves the bytes itself: they must be there
  if (c->p2p(c->user, (int)ops.size(), ops.data()) != 0) return fail(MGCMT_ERR_HIP, "external point-to-point transport failed");
  return MGCMT_OK;
}

// the main stream waits for what still runs beside it: an exchange, a pass's edge rows
int join_exchange(mgcmt_plan* p, hipStream_t s) {
  ShardComm* c = p->comm;
  if (c->pending) {
    MG_HIP(hipStreamWaitEvent(s, c->ev_done, 0));
    c->pending = false;
  }
  if (c->edges_pending) {
    MG_HIP(hipStreamWaitEvent(s, c->ev_edges, 0));
    c->edges_pending = false;
  }
  return MGCMT_OK;
}

// once per communicator and cycle stream: an exchange stream that runs beside `s` (a few candidates are created and tried;
// RCCL transport with overlap only — the callback transports synchronise anyway)
int pick_exchange_stream(mgcmt_plan* p, hipStream_t s) {
  ShardComm* c = p->comm;
  if (!c->nccl || !c->overlap || (c->lane_checked && c->lane_for == s)) return MGCMT_OK;
  c->lane_checked = true;
  c->lane_for = s;
  const char* e = getenv("MGCMT_COMM_LANE_CHECK");  // "0": keep the stream the communicator was created with
  if (e && e[0] == '0') return MGCMT_OK;
  int* d_sink = nullptr;
  if (hipMalloc((void**)&d_sink, 2 * sizeof(int)) != hipSuccess) {
    (void)hipGetLastError();
    return MGCMT_OK;
  }
  MG_HIP(hipStreamSynchronize(c->comm_stream));
  if (!runs_beside(s, c->comm_stream, d_sink)) {
    std::vector<hipStream_t> tried;
    for (int t = 0; t < 6; ++t) {
      hipStream_t cand = nullptr;
      if (hipStreamCreateWithFlags(&cand, hipStreamNonBlocking) != hipSuccess) {
        (void)hipGetLastError();
        break;
      }
      if (runs_beside(s, cand, d_sink)) {
        (void)hipStreamDestroy(c->comm_stream);
        c->comm_stream = cand;
        break;
      }
      tried.push_back(cand);  // (kept alive until the search ends: a destroyed stream's queue slot would be dealt again)
    }
    for (hipStream_t t : tried) (void)hipStreamDestroy(t);
  }
  (void)hipFree(d_sink);
  return MGCMT_OK;
}

// One fused pass on strip level l whose products the neighbours need next: V' boundary rows (unless the pass stores
// nothing, or the caller exchanges V itself after a Gram-Schmidt: exchange_v = false) and, after a restriction, the
// coarse right-hand side's boundary rows (when level l+1 is a strip level too).  Boundary rows first, their exchange
// beside the interior rows.
int strip_pass(mgcmt_plan* p, int l, int kind, int n, double omega, int mode, int npre, int k, hipStream_t s, bool coarse_is_strip,
               bool exchange_v = true) {
  ShardComm* c = p->comm;
  Level& L = p->levels[l];
  const bool stores_v = !(mode & 8) && exchange_v;
  const bool sends_f = (mode & 3) == 2 && coarse_is_strip;
  const bool ring = c->self_ring;
  const bool up = c->rank > 0 || ring, down = c->rank + 1 < c->nranks || ring;
  const long B = boundary_rows(p, l, sends_f);
  // two extra launches cost about 10 us on the stream; the exchange they free from the critical path is worth more than
  // that only on strips whose interior launch is long enough to hide it (c->split == 2: always, for tests)
  static const int split_min_log2 = [] {
    const char* e = getenv("MGCMT_COMM_SPLIT_MIN_LOG2");  // tuning: strips of at least 2^this points take the split schedule
    const int v = e ? atoi(e) : 22;
    return v < 10 || v > 40 ? 22 : v;
  }();
  const bool big = c->split == 2 || (long)L.nr * L.gc * k >= (1L << split_min_log2);
  const bool split = c->split && big && (stores_v || sends_f) && (up || down) && L.nr >= 4 * B;
  std::vector<Msg> msgs;
  if (!split) {
    MG_TRY(join_exchange(p, s));
    MG_TRY(fused_pass(p, l, kind, n, omega, mode, k, s, npre));
    if (stores_v) MG_TRY(halo_msgs(p, l, MGCMT_SLOT_V, ring, k, &msgs));
    if (sends_f) MG_TRY(halo_msgs(p, l + 1, MGCMT_SLOT_F, ring, k, &msgs));
    return run_msgs(p, msgs, s);
  }
  const long lo = up ? B : 0, hi = down ? L.nr - B : L.nr;
  // both edges in ONE launch (two row ranges): these launches are pure march latency, 11 - 15 us each
  auto edges = [&](hipStream_t es) {
    if (up && down) return fused_pass(p, l, kind, n, omega, mode, k, es, npre, 0, B, false, L.nr - B, L.nr);
    if (up) return fused_pass(p, l, kind, n, omega, mode, k, es, npre, 0, B, false);
    return fused_pass(p, l, kind, n, omega, mode, k, es, npre, L.nr - B, L.nr, false);
  };
  // the messages name the buffer the pass writes: V' lives in slot T until the roles are swapped (by the interior launch)
  auto messages = [&]() {
    if (stores_v) {
      std::swap(L.base[MGCMT_SLOT_V], L.base[MGCMT_SLOT_T]);
      const int rc = halo_msgs(p, l, MGCMT_SLOT_V, ring, k, &msgs);
      std::swap(L.base[MGCMT_SLOT_V], L.base[MGCMT_SLOT_T]);
      MG_TRY(rc);
    }
    if (sends_f) MG_TRY(halo_msgs(p, l + 1, MGCMT_SLOT_F, ring, k, &msgs));
    return (int)MGCMT_OK;
  };
  if (c->nccl && c->overlap) {
    // Two lanes.  Main stream: the interior launches, back to back — an interior launch reads no halo row (its rows
    // lie boundary_rows inside the strip), so it waits for the previous pass's EDGE rows only, never for an exchange.
    // Exchange stream (high priority): edge rows of this pass — they need the previous pass complete (event on the main
    // stream) and the previous exchange (stream order) —, then their exchange.  What an exchange costs beyond its
    // interior launch (it runs starved of bandwidth beside it and ends some 15 us after it, the trace of round 3 shows)
    // then delays the next pass's edge launch, which runs beside the next interior launch, instead of the whole pass.
    MG_HIP(hipEventRecord(c->ev_boundary, s));                       // everything before this pass, on the main stream
    if (c->edges_pending) MG_HIP(hipStreamWaitEvent(s, c->ev_edges, 0));  // interior <- the previous pass's edge rows
    MG_HIP(hipStreamWaitEvent(c->comm_stream, c->ev_boundary, 0));
    MG_TRY(edges(c->comm_stream));
    MG_HIP(hipEventRecord(c->ev_edges, c->comm_stream));
    c->edges_pending = true;
    MG_TRY(messages());
    MG_TRY(run_msgs(p, msgs, c->comm_stream));
    MG_HIP(hipEventRecord(c->ev_done, c->comm_stream));
    c->pending = true;
    MG_TRY(fused_pass(p, l, kind, n, omega, mode, k, s, npre, lo, hi, true));
    return MGCMT_OK;
  }
  MG_TRY(join_exchange(p, s));
  MG_TRY(edges(s));
  MG_TRY(messages());
  MG_TRY(fused_pass(p, l, kind, n, omega, mode, k, s, npre, lo, hi, true));
  return run_msgs(p, msgs, s);
}

// sum over ranks of n <= kRedWords doubles at c->d_red, in stream order (RCCL) or through the host (callbacks)
int allreduce_device(mgcmt_plan* p, int n, hipStream_t s) {
  ShardComm* c = p->comm;
  if (c->nranks == 1 || n < 1) return MGCMT_OK;
  if (c->nccl) {
    RcclApi* api = rccl();
    MG_RCCL(api, api->AllReduce(c->d_red, c->d_red, (size_t)n, kRcclFloat64, kRcclSum, c->nccl, s));
    return MGCMT_OK;
  }
  if (!c->allreduce) return fail(MGCMT_ERR_INVALID, "communicator has no all-reduce transport");
  c->h_red.resize(kRedWords);
  MG_HIP(hipMemcpyAsync(c->h_red.data(), c->d_red, sizeof(double) * n, hipMemcpyDeviceToHost, s));
  MG_HIP(hipStreamSynchronize(s));
  if (c->allreduce(c->user, c->h_red.data(), n) != 0) return fail(MGCMT_ERR_HIP, "external all-reduce transport failed");
  MG_HIP(hipMemcpyAsync(c->d_red, c->h_red.data(), sizeof(double) * n, hipMemcpyHostToDevice, s));
  MG_HIP(hipStreamSynchronize(s));  // (h_red is reused by the next step)
  return MGCMT_OK;
}

// Modified Gram-Schmidt (MGCMTProcessor.py:44-50) of the k columns of slot V on strip level l: the single plan's one
// launch per column (k_mgs_step: project column i out of the later ones, normalise it, accumulate the inner products the
// next column needs), with the inner products summed over the ranks between the steps — ONE all-reduce of the k - i
// coefficients per column, in stream order.  Every rank ends with the same coefficients, hence with
// its rows of the same orthonormal columns.
int sharded_gramschmidt(mgcmt_plan* p, int l, int k, hipStream_t s) {
  ShardComm* c = p->comm;
  const long n = p->interior(l);
  const long stride = p->levels[l].stride;
  double* a0 = p->kvec(l, MGCMT_SLOT_V, 0).p;
  double* pa = p->d_partials;
  if (p->use_mgs_block && k >= 2 && k <= mgs_block_max() && p->d_mgs) {
    // The two-pass form (kernels_blas.hip): this rank's rows of the Gram matrix, ONE all-reduce of its k (k + 1) / 2
    // entries — every rank then factors the same matrix and takes the same decision —, Q = A R^-1 on this rank's
    // rows.  The decision (the gate word) is read back: the column-by-column form below has collectives in it, which
    // only the host can leave out.
    const int npairs = k * (k + 1) / 2;
    const int blocks = launch_mgs_gram(s, n, a0, stride, k, pa);
    launch_final_sums(s, npairs, blocks, pa, c->d_red);
    MG_TRY(post_launch());
    MG_TRY(allreduce_device(p, npairs, s));
    launch_mgs_factor(s, c->d_red, 1, k, p->d_mgs);
    MG_TRY(post_launch());
    double gate = 1.0;
    MG_HIP(hipMemcpyAsync(&gate, p->d_mgs + mgs_block_gate_word(), sizeof(double), hipMemcpyDeviceToHost, s));
    MG_HIP(hipStreamSynchronize(s));
    if (gate == 0.0) {
      launch_mgs_apply(s, n, a0, stride, k, p->d_mgs);
      return post_launch();
    }
  }
  const int nb = reduce_blocks(n);
  launch_dot_partials(s, n, a0, a0, stride, k, pa);  // <a_0, a_t>, t = 0..k-1, this rank's rows
  launch_final_sums(s, k, nb, pa, c->d_red);
  MG_TRY(post_launch());
  MG_TRY(allreduce_device(p, k, s));
  for (int i = 0; i < k; ++i) {
    const int m = k - 1 - i;
    launch_mgs_step(s, n, c->d_red, a0 + i * stride, stride, m, pa, 1);
    if (m > 0) {
      launch_final_sums(s, m, nb, pa, c->d_red);
      MG_TRY(post_launch());
      MG_TRY(allreduce_device(p, m, s));
    }
  }
  return post_launch();
}

std::vector<int> split_sweeps(const mgcmt_plan* p, int l, int kind, int nu) {
  std::vector<int> out;
  for (int left = nu; left > 0;) {
    const int n = pass_sweeps(p, l, kind, left);
    out.push_back(n);
    left -= n;
  }
  return out;
}

}  // namespace

extern "C" {

int mgcmt_comm_unique_id(void* id_out) {
  if (!id_out) return fail(MGCMT_ERR_INVALID, "null output");
  RcclApi* api = rccl();
  if (!api) return fail(MGCMT_ERR_UNSUPPORTED, "RCCL is not available in this process");
  RcclUniqueId id;
  MG_RCCL(api, api->GetUniqueId(&id));
  memcpy(id_out, id.internal, MGCMT_UNIQUE_ID_BYTES);
  return MGCMT_OK;
}

static int comm_common(mgcmt_plan* p, int rank, int nranks, ShardComm** out) {
  if (!p) return fail(MGCMT_ERR_INVALID, "null plan");
  if (nranks < 1 || rank < 0 || rank >= nranks) return fail(MGCMT_ERR_INVALID, "bad rank / nranks");
  if (p->dim != 2) return fail(MGCMT_ERR_UNSUPPORTED, "only 2-D plans are sharded");
  comm_release(p);
  MG_HIP(hipSetDevice(p->device));
  ShardComm* c = new ShardComm();
  c->rank = rank;
  c->nranks = nranks;
  p->comm = c;
  // non-blocking: work on it must not serialise with the legacy default stream the cycle may be enqueued on.
  // An ORDINARY stream.  A high-priority one (its own class of hardware queue) was tried in round 3 and measured: equal to
  // an ordinary stream in a fresh process (2.22 ms per emulated rank share of the 32768^2 cycle either way) — but created
  // AFTER the eight DMA lane streams of csrc/transfer.hip exist (any process that uploaded >= 16 MiB first, bench.py's
  // one-GPU line for one) every small kernel of the process then takes ~55 us and the share 3.9 ms: the hardware queues are
  // oversubscribed and time-sliced (rocprofv3 trace: profiles/r03_rank_share_priority_stream_oversubscribed.txt).
  // MGCMT_COMM_PRIORITY=1 selects the priority stream (A/B measurements).
  const char* pe = getenv("MGCMT_COMM_PRIORITY");
  hipError_t e = hipErrorInvalidValue;  // (anything but success: no priority stream asked for)
  if (pe && pe[0] == '1') {
    int prio_low = 0, prio_high = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_low, &prio_high);
    e = hipStreamCreateWithPriority(&c->comm_stream, hipStreamNonBlocking, prio_high);
    if (e != hipSuccess) (void)hipGetLastError();
  }
  if (e != hipSuccess) e = hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreate(&c->ev_boundary);
  if (e == hipSuccess) e = hipEventCreate(&c->ev_done);
  if (e == hipSuccess) e = hipEventCreate(&c->ev_edges);
  if (e == hipSuccess) e = hipMalloc((void**)&c->d_red, sizeof(double) * kRedWords);
  if (e != hipSuccess) {
    comm_release(p);
    return fail(MGCMT_ERR_HIP, std::string("communicator resources: ") + hipGetErrorString(e));
  }
  *out = c;
  return MGCMT_OK;
}

int mgcmt_comm_init(mgcmt_plan* p, int rank, int nranks, const void* unique_id) {
  if (!unique_id) return fail(MGCMT_ERR_INVALID, "null unique id");
  RcclApi* api = rccl();
  if (!api) return fail(MGCMT_ERR_UNSUPPORTED, "RCCL is not available in this process");
  ShardComm* c = nullptr;
  MG_TRY(comm_common(p, rank, nranks, &c));
  RcclUniqueId id;
  memcpy(id.internal, unique_id, MGCMT_UNIQUE_ID_BYTES);
  const int r = api->CommInitRank(&c->nccl, nranks, id, rank);
  if (r != kRcclSuccess) {
    c->nccl = nullptr;
    comm_release(p);
    return fail(MGCMT_ERR_HIP, std::string("ncclCommInitRank: ") + api->GetErrorString(r));
  }
  return MGCMT_OK;
}

int mgcmt_comm_init_external(mgcmt_plan* p, int rank, int nranks, mgcmt_p2p_fn p2p, mgcmt_allgather_fn allgather,
                             mgcmt_allreduce_fn allreduce, void* user) {
  if (nranks > 1 && (!p2p || !allgather)) return fail(MGCMT_ERR_INVALID, "missing transport callbacks");
  ShardComm* c = nullptr;
  MG_TRY(comm_common(p, rank, nranks, &c));
  c->p2p = p2p;
  c->allgather = allgather;
  c->allreduce = allreduce;
  c->user = user;
  c->overlap = false;
  return MGCMT_OK;
}

int mgcmt_comm_destroy(mgcmt_plan* p) {
  if (!p) return fail(MGCMT_ERR_INVALID, "null plan");
  comm_release(p);
  return MGCMT_OK;
}

int mgcmt_comm_set_option(mgcmt_plan* p, int option, int value) {
  MG_TRY(check_comm(p));
  if (option == MGCMT_COMM_OPT_OVERLAP) p->comm->overlap = value != 0 && p->comm->nccl != nullptr;
  else if (option == MGCMT_COMM_OPT_SPLIT) p->comm->split = value < 0 ? 0 : (value > 2 ? 2 : value);
  else if (option == MGCMT_COMM_OPT_SELF_RING) {
    if (value && p->comm->nranks != 1) return fail(MGCMT_ERR_INVALID, "the self-ring test mode needs a one-rank communicator");
    p->comm->self_ring = value != 0;
  }
  else if (option == MGCMT_COMM_OPT_EMULATE_OF) {
    if (value > 1 && (p->comm->nranks != 1 || !p->comm->self_ring))
      return fail(MGCMT_ERR_INVALID, "rank emulation needs a one-rank communicator in self-ring mode");
    p->comm->emulate_of = value > 1 ? value : 0;
  }
  else return fail(MGCMT_ERR_INVALID, "unknown communicator option");
  return MGCMT_OK;
}

int mgcmt_halo_exchange(mgcmt_plan* p, int l, int slot_mask, void* stream) {
  MG_TRY(check_comm(p));
  if (l < 0 || l >= (int)p->levels.size()) return fail(MGCMT_ERR_INVALID, "level out of range");
  const bool ring = (slot_mask & kRing) != 0;
  if (ring && p->comm->nranks != 1) return fail(MGCMT_ERR_INVALID, "the ring self-test needs a one-rank communicator");
  hipStream_t s = (hipStream_t)stream;
  MG_TRY(join_exchange(p, s));
  std::vector<Msg> msgs;
  const int k = (slot_mask >> 16) & 0xff ? (slot_mask >> 16) & 0xff : 1;  // bits 16-23: vectors 0..k-1 (0 = one)
  if (k > p->nvec) return fail(MGCMT_ERR_INVALID, "halo exchange of more vectors than the plan holds");
  for (int slot = 0; slot < 4; ++slot)
    if (slot_mask & (1 << slot)) MG_TRY(halo_msgs(p, l, slot, ring, k, &msgs));
  return run_msgs(p, msgs, s);
}

int mgcmt_gather_coarse(mgcmt_plan* p, int l, int slot, mgcmt_plan* coarse, int dst_slot, int k, void* stream) {
  MG_TRY(check_comm(p));
  if (!coarse) return fail(MGCMT_ERR_INVALID, "null coarse plan");
  if (l < 0 || l >= (int)p->levels.size()) return fail(MGCMT_ERR_INVALID, "level out of range");
  if (k < 1 || k > p->nvec || k > coarse->nvec) return fail(MGCMT_ERR_INVALID, "k must be in 1..nvec of both plans");
  ShardComm* c = p->comm;
  hipStream_t s = (hipStream_t)stream;
  MG_TRY(join_exchange(p, s));
  MG_TRY(ensure_slot(p, l, slot));
  MG_TRY(ensure_slot(coarse, 0, dst_slot));
  const Level& L = p->levels[l];
  const Level& C = coarse->levels[0];
  const int parts = c->emulate_of > 1 ? c->emulate_of : c->nranks;
  if (C.nr != C.gr || C.gc != L.gc || L.nr * parts != C.gr)
    return fail(MGCMT_ERR_INVALID, "coarse plan's finest level is not the whole grid of this strip level");
  const size_t cnt = (size_t)L.nr * L.gc;
  if (c->nranks == 1) {
    // one rank: a copy.  Emulating rank R of N (timing rehearsal): the own strip lands where rank R's would, and stands
    // in for the other ranks' strips too, so that the redundant coarse sub-cycle works on data of the right kind
    for (int q = 0; q < k; ++q)
      for (int part = 0; part < parts; ++part)
        MG_HIP(hipMemcpyAsync(coarse->kvec(0, dst_slot, q).p + (size_t)part * cnt, p->kvec(l, slot, q).p, cnt * sizeof(double),
                              hipMemcpyDeviceToDevice, s));
    return MGCMT_OK;
  }
  if (c->nccl) {
    RcclApi* api = rccl();
    if (k > 1) MG_RCCL(api, api->GroupStart());
    int bad = kRcclSuccess;
    for (int q = 0; q < k && bad == kRcclSuccess; ++q)
      bad = api->AllGather(p->kvec(l, slot, q).p, coarse->kvec(0, dst_slot, q).p, cnt, kRcclFloat64, c->nccl, s);
    const int end = k > 1 ? api->GroupEnd() : kRcclSuccess;
    if (bad != kRcclSuccess) return fail(MGCMT_ERR_HIP, std::string("ncclAllGather: ") + (api->GetErrorString ? api->GetErrorString(bad) : "rccl error"));
    MG_RCCL(api, end);
    return MGCMT_OK;
  }
  if (!c->allgather) return fail(MGCMT_ERR_INVALID, "communicator has no all-gather transport");
  MG_HIP(hipStreamSynchronize(s));
  for (int q = 0; q < k; ++q)
    if (c->allgather(c->user, p->kvec(l, slot, q).p, coarse->kvec(0, dst_slot, q).p, (int64_t)cnt) != 0)
      return fail(MGCMT_ERR_HIP, "external all-gather transport failed");
  return MGCMT_OK;
}

int mgcmt_allreduce_sum(mgcmt_plan* p, double* host_inout, int n, void* stream) {
  MG_TRY(check_comm(p));
  if (!host_inout || n < 1 || n > 2 * kMaxVec) return fail(MGCMT_ERR_INVALID, "allreduce: 1..64 values");
  ShardComm* c = p->comm;
  hipStream_t s = (hipStream_t)stream;
  if (c->nranks == 1) return MGCMT_OK;
  if (c->nccl) {
    RcclApi* api = rccl();
    MG_HIP(hipMemcpyAsync(c->d_red, host_inout, sizeof(double) * n, hipMemcpyHostToDevice, s));
    MG_RCCL(api, api->AllReduce(c->d_red, c->d_red, (size_t)n, kRcclFloat64, kRcclSum, c->nccl, s));
    MG_HIP(hipMemcpyAsync(host_inout, c->d_red, sizeof(double) * n, hipMemcpyDeviceToHost, s));
    MG_HIP(hipStreamSynchronize(s));
    return MGCMT_OK;
  }
  if (!c->allreduce) return fail(MGCMT_ERR_INVALID, "communicator has no all-reduce transport");
  if (c->allreduce(c->user, host_inout, n) != 0) return fail(MGCMT_ERR_HIP, "external all-reduce transport failed");
  return MGCMT_OK;
}

// One V(nu1,nu2) cycle of the sharded hierarchy on k vectors: `p` holds levels 0 .. ls-1 as row strips (and level ls as
// the strip buffer the last restriction writes and the first prolongation reads), `coarse` holds the grid of level ls
// WHOLE on every rank.  MGCMTSolver.py:281-329 (k = 1) / :375-436 (vcycle_matrix: k columns with their own shifts and,
// with MGCMT_SHARDED_GRAM_SCHMIDT, the modified Gram-Schmidt of :434 on every level on the way up) with the recursion
// unrolled; weighted Jacobi and multicolour Gauss-Seidel are order-independent, so this is the single-GPU cycle's
// arithmetic (the Gram-Schmidt's inner products are summed rank by rank: equal to rounding).
int mgcmt_sharded_vcycle(mgcmt_plan* p, mgcmt_plan* coarse, int nu1, int nu2, int nu_coarse, int kind, double omega, int k, int flags,
                         void* stream) {
  MG_TRY(check_comm(p));
  if (!coarse) return fail(MGCMT_ERR_INVALID, "null coarse plan");
  if (kind != MGCMT_WJACOBI && kind != MGCMT_GS_MC)
    return fail(MGCMT_ERR_UNSUPPORTED, "only weighted Jacobi and multicolour Gauss-Seidel shard; lexicographic sweeps are sequential");
  if (nu1 < 1 || nu2 < 1 || nu_coarse < 1) return fail(MGCMT_ERR_INVALID, "the sharded cycle needs at least one sweep per leg");
  if (k < 1 || k > p->nvec || k > coarse->nvec) return fail(MGCMT_ERR_INVALID, "k must be in 1..nvec of both plans");
  if (flags & ~(MGCMT_SHARDED_V_HALO_VALID | MGCMT_SHARDED_F_HALO_VALID | MGCMT_SHARDED_GRAM_SCHMIDT)) return fail(MGCMT_ERR_INVALID, "unknown flag");
  const bool gram_schmidt = (flags & MGCMT_SHARDED_GRAM_SCHMIDT) != 0;
  const int ls = (int)p->levels.size() - 1;  // strip levels 0 .. ls-1
  if (ls < 1) return fail(MGCMT_ERR_INVALID, "strip plan has no level below the finest one");
  ShardComm* c = p->comm;
  hipStream_t s = (hipStream_t)stream;
  MG_TRY(pick_exchange_stream(p, s));
  for (int l = 0; l <= ls; ++l) {
    MG_TRY(ensure_slot(p, l, MGCMT_SLOT_V));
    MG_TRY(ensure_slot(p, l, MGCMT_SLOT_F));
    if (l < ls) MG_TRY(ensure_slot(p, l, MGCMT_SLOT_T));
    if (l < ls && !fused_level(p, l, kind)) return fail(MGCMT_ERR_UNSUPPORTED, "a strip level is not covered by the fused kernels");
  }
  const int ring_bit = c->self_ring ? kRing : 0;
  // what the first pass reads from the neighbours: V (unless the caller vouches that nothing changed it since the
  // previous sharded cycle, whose last pass exchanged it) and, once per right-hand side, F
  {
    int mask = 0;
    if (!(flags & MGCMT_SHARDED_V_HALO_VALID)) mask |= 1 << MGCMT_SLOT_V;
    if (!(flags & MGCMT_SHARDED_F_HALO_VALID)) mask |= 1 << MGCMT_SLOT_F;
    if (mask) MG_TRY(mgcmt_halo_exchange(p, 0, mask | ring_bit | (k << 16), stream));
  }
  std::vector<int> recompute(ls, 0);
  std::vector<char> still_zero(ls, 0);
  for (int l = 0; l < ls; ++l) {
    const int nu = l == 0 ? nu1 : nu_coarse, nu_up = l == 0 ? nu2 : nu_coarse;
    const std::vector<int> passes = split_sweeps(p, l, kind, nu);
    const int first_up = split_sweeps(p, l, kind, nu_up)[0];
    for (size_t i = 0; i < passes.size(); ++i) {
      const bool last = i + 1 == passes.size();
      const bool zero_in = i == 0 && l > 0;  // the coarse iterate starts at zero: nothing to read or exchange
      int mode = last ? 2 : 0;
      // recompute instead of store (bandwidth-bound levels): the last down-leg pass writes only the restricted
      // residual, the first up-leg pass re-runs its sweeps from the untouched V (whose halo rows stay valid)
      const bool big = p->force_recompute || p->interior(l) >= (1L << 22);
      if (last && p->use_recompute && big &&
          passes[i] <= fused_max_recompute(p->levels[l].dA.k, kind == MGCMT_GS_MC ? 1 : 0, first_up)) {
        mode |= 8;
        recompute[l] = passes[i];
        still_zero[l] = zero_in;
      }
      MG_TRY(strip_pass(p, l, kind, passes[i], omega, mode | (zero_in ? 4 : 0), 0, k, s, l + 1 < ls));
    }
  }
  // the coarse problem: all-gather, the same sub-cycle on every rank, own rows of the correction with halo rows
  MG_TRY(mgcmt_gather_coarse(p, ls, MGCMT_SLOT_F, coarse, MGCMT_SLOT_F, k, stream));
  MG_TRY(ensure_slot(coarse, 0, MGCMT_SLOT_V));
  MG_TRY(mgcmt_vcycle(coarse, 0, nu_coarse, nu_coarse, nu_coarse, kind, omega, k,
                      MGCMT_CYCLE_ZERO_START | (gram_schmidt ? MGCMT_CYCLE_GRAM_SCHMIDT : 0), stream));
  {
    const Level& L = p->levels[ls];
    const long total = L.gr;
    const long lo = std::max<long>(L.r0 - L.halo, 0), hi = std::min<long>(L.r0 + L.nr + L.halo, total);
    for (int q = 0; q < k; ++q)
      MG_HIP(hipMemcpyAsync(p->kvec(ls, MGCMT_SLOT_V, q).p + (lo - L.r0) * L.gc, coarse->kvec(0, MGCMT_SLOT_V, q).p + lo * L.gc,
                            sizeof(double) * (size_t)(hi - lo) * L.gc, hipMemcpyDeviceToDevice, s));
  }
  for (int l = ls - 1; l >= 0; --l) {
    const std::vector<int> passes = split_sweeps(p, l, kind, l == 0 ? nu2 : nu_coarse);
    for (size_t i = 0; i < passes.size(); ++i) {
      const int mode = i == 0 ? (1 | (still_zero[l] ? 4 : 0)) : 0;
      // with a Gram-Schmidt behind the level's last pass every value of V changes once more: V travels after it
      const bool exchange_v = !(gram_schmidt && i + 1 == passes.size());
      MG_TRY(strip_pass(p, l, kind, passes[i], omega, mode, i == 0 ? recompute[l] : 0, k, s, false, exchange_v));
    }
    if (gram_schmidt) {
      MG_TRY(join_exchange(p, s));
      MG_TRY(sharded_gramschmidt(p, l, k, s));
      MG_TRY(mgcmt_halo_exchange(p, l, (1 << MGCMT_SLOT_V) | ring_bit | (k << 16), stream));
    }
  }
  // leave nothing running beside the caller's stream
  MG_TRY(join_exchange(p, s));
  return post_launch();
}

}  // extern "C"
