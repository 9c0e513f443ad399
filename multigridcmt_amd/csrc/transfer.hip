// Host <-> device transfers of whole level vectors for the drop-in classes (MGCMTSolver.vcycle(v0, f, A, ...) takes and
// returns NumPy arrays, as the reference does): pageable caller memory moves through a ring of pinned chunks, several
// host threads copying into / out of the ring while the DMA engine moves the previous chunks, instead of one
// hipMemcpy that stages everything through the runtime's own bounce buffer on one thread (6-21 GB/s measured; a
// download into a fresh array is bound by first-touch page faults of ONE thread).
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

#include "plan_internal.h"

namespace mgcmt {

namespace {

// Pinned host buffers handed out by mgcmt_host_alloc (start -> bytes): a transfer whose host side lies inside one is a
// single DMA, no staging.  (The drop-in classes return their results in such buffers and recycle them, hostmem.py.)
std::mutex g_pinned_mu;
std::map<const char*, size_t> g_pinned;

bool in_pinned_registry(const void* host, size_t bytes) {
  std::lock_guard<std::mutex> lock(g_pinned_mu);
  auto it = g_pinned.upper_bound((const char*)host);
  if (it == g_pinned.begin()) return false;
  --it;
  return (const char*)host + bytes <= it->first + it->second;
}

}  // namespace

#if defined(__HIP__)  // (the host-only emulation build of the tests takes the plain path)
namespace {

// page-locked memory the runtime knows about (ours, or a caller's own: hipHostMalloc / hipHostRegister / a pinned tensor)
bool is_pinned(const void* host, size_t bytes) {
  if (in_pinned_registry(host, bytes)) return true;
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, host) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return attr.type == hipMemoryTypeHost;
}

constexpr size_t kChunk = 4u << 20;      // bytes per pinned chunk
constexpr int kThreads = 8;              // copying threads, two chunks each (one being filled, one in flight)
constexpr size_t kStagedMin = 16u << 20; // smaller transfers take the plain path

struct Lane {
  char* buf[2] = {nullptr, nullptr};
  hipEvent_t ev[2] = {nullptr, nullptr};
  hipStream_t stream = nullptr;
};

struct Ring {
  std::mutex mu;  // one staged transfer at a time
  bool ready = false, failed = false;
  int device = -1;
  Lane lane[kThreads];
};

Ring g_ring;

bool ring_init(int device) {
  Ring& r = g_ring;
  if (r.failed) return false;
  if (r.ready && r.device == device) return true;
  if (r.ready) return false;  // another device's ring: keep it simple, plain path
  for (int t = 0; t < kThreads; ++t) {
    Lane& L = r.lane[t];
    if (hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking) != hipSuccess) {
      r.failed = true;
      return false;
    }
    for (int b = 0; b < 2; ++b)
      if (hipHostMalloc((void**)&L.buf[b], kChunk, hipHostMallocDefault) != hipSuccess || hipEventCreateWithFlags(&L.ev[b], hipEventDisableTiming) != hipSuccess) {
        r.failed = true;
        return false;
      }
  }
  r.device = device;
  r.ready = true;
  return true;
}

// thread t moves chunks t, t + kThreads, ...: host -> pinned -> device (upload) or device -> pinned -> host (download)
void lane_work(int t, int device, bool upload, char* dev, char* host, size_t bytes, std::atomic<int>* error) {
  if (hipSetDevice(device) != hipSuccess) {
    error->store(1);
    return;
  }
  Lane& L = g_ring.lane[t];
  const size_t nchunks = (bytes + kChunk - 1) / kChunk;
  int b = 0;
  if (upload) {
    bool used[2] = {false, false};
    for (size_t c = t; c < nchunks; c += kThreads, b ^= 1) {
      const size_t off = c * kChunk, len = bytes - off < kChunk ? bytes - off : kChunk;
      if (used[b] && hipEventSynchronize(L.ev[b]) != hipSuccess) error->store(1);  // its previous copy has left the chunk
      memcpy(L.buf[b], host + off, len);
      if (hipMemcpyAsync(dev + off, L.buf[b], len, hipMemcpyHostToDevice, L.stream) != hipSuccess) error->store(1);
      if (hipEventRecord(L.ev[b], L.stream) != hipSuccess) error->store(1);
      used[b] = true;
    }
  } else {
    // keep one chunk in flight while the previous one is copied out
    size_t pend_off = 0, pend_len = 0;
    int pend_b = -1;
    for (size_t c = t; c < nchunks; c += kThreads, b ^= 1) {
      const size_t off = c * kChunk, len = bytes - off < kChunk ? bytes - off : kChunk;
      if (hipMemcpyAsync(L.buf[b], dev + off, len, hipMemcpyDeviceToHost, L.stream) != hipSuccess) error->store(1);
      if (hipEventRecord(L.ev[b], L.stream) != hipSuccess) error->store(1);
      if (pend_b >= 0) {
        if (hipEventSynchronize(L.ev[pend_b]) != hipSuccess) error->store(1);
        memcpy(host + pend_off, L.buf[pend_b], pend_len);
      }
      pend_off = off;
      pend_len = len;
      pend_b = b;
    }
    if (pend_b >= 0) {
      if (hipEventSynchronize(L.ev[pend_b]) != hipSuccess) error->store(1);
      memcpy(host + pend_off, L.buf[pend_b], pend_len);
    }
  }
  if (hipStreamSynchronize(L.stream) != hipSuccess) error->store(1);
}

}  // namespace
#endif

// Both return MGCMT_OK / an error and have completed the transfer when they return; `stream` is synchronised first
// (work enqueued on it may still read or write the device vector).
int transfer(int device, bool upload, void* dev, void* host, size_t bytes, hipStream_t stream) {
  MG_HIP(hipStreamSynchronize(stream));
#if defined(__HIP__)
  static const bool enabled = [] {
    const char* e = getenv("MGCMT_STAGED_TRANSFER");  // "0": always the plain hipMemcpy path (A/B measurements)
    return !(e && e[0] == '0');
  }();
  if (enabled && bytes >= kStagedMin && !is_pinned(host, bytes)) {
    std::lock_guard<std::mutex> lock(g_ring.mu);
    if (ring_init(device)) {
      std::atomic<int> error{0};
      std::vector<std::thread> pool;
      for (int t = 1; t < kThreads; ++t) pool.emplace_back(lane_work, t, device, upload, (char*)dev, (char*)host, bytes, &error);
      lane_work(0, device, upload, (char*)dev, (char*)host, bytes, &error);
      for (std::thread& th : pool) th.join();
      if (error.load()) return fail(MGCMT_ERR_HIP, "staged host transfer failed");
      return MGCMT_OK;
    }
    (void)hipGetLastError();
  }
#endif
  if (upload) MG_HIP(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, stream));
  else MG_HIP(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, stream));
  MG_HIP(hipStreamSynchronize(stream));
  return MGCMT_OK;
}

}  // namespace mgcmt

extern "C" {

int mgcmt_host_alloc(int64_t bytes, void** out) {
  using namespace mgcmt;
  if (!out || bytes <= 0) return fail(MGCMT_ERR_INVALID, "mgcmt_host_alloc: bad arguments");
  void* p = nullptr;
#if defined(__HIP__)
  MG_HIP(hipHostMalloc(&p, (size_t)bytes, hipHostMallocDefault));
#else
  p = malloc((size_t)bytes);
  if (!p) return fail(MGCMT_ERR_NOMEM, "mgcmt_host_alloc: out of memory");
#endif
  {
    std::lock_guard<std::mutex> lock(g_pinned_mu);
    g_pinned[(const char*)p] = (size_t)bytes;
  }
  *out = p;
  return MGCMT_OK;
}

int mgcmt_host_free(void* p) {
  using namespace mgcmt;
  if (!p) return MGCMT_OK;
  {
    std::lock_guard<std::mutex> lock(g_pinned_mu);
    auto it = g_pinned.find((const char*)p);
    if (it == g_pinned.end()) return fail(MGCMT_ERR_INVALID, "mgcmt_host_free: not a buffer of mgcmt_host_alloc");
    g_pinned.erase(it);
  }
#if defined(__HIP__)
  MG_HIP(hipHostFree(p));
#else
  free(p);
#endif
  return MGCMT_OK;
}

}  // extern "C"
