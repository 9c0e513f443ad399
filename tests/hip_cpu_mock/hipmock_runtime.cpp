// TEST INFRASTRUCTURE — runtime of the host-only HIP stand-in (see hip/hip_runtime.h).
#include <hip/hip_runtime.h>

#include <sys/mman.h>

namespace hipmock {

uint3_t g_threadIdx, g_blockIdx;
dim3 g_blockDim, g_gridDim;
double g_xchg_d[1024];
long long g_xchg_i[1024];

namespace {

constexpr size_t kStackBytes = 256 * 1024;
constexpr int kMaxThreads = 1024;

extern "C" void hipmock_ctx_switch(void** from_sp, void* to_sp);
asm(R"(
.text
.globl hipmock_ctx_switch
.type hipmock_ctx_switch,@function
hipmock_ctx_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size hipmock_ctx_switch,.-hipmock_ctx_switch
)");

struct Fiber {
  void* sp = nullptr;
  char* stack = nullptr;
  bool done = true;
  uint3_t tid;
};

Fiber g_fibers[kMaxThreads];
void* g_sched_sp = nullptr;
int g_current = -1;
const std::function<void()>* g_body = nullptr;

void fiber_entry() {
  (*g_body)();
  g_fibers[g_current].done = true;
  hipmock_ctx_switch(&g_fibers[g_current].sp, g_sched_sp);
  abort();  // a finished fiber is never resumed
}

void prepare(Fiber& f) {
  if (!f.stack) {
    f.stack = (char*)mmap(nullptr, kStackBytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (f.stack == (char*)MAP_FAILED) {
      fprintf(stderr, "hipmock: cannot allocate a fiber stack\n");
      abort();
    }
  }
  uintptr_t top = ((uintptr_t)(f.stack + kStackBytes)) & ~(uintptr_t)15;
  void** sp = (void**)(top - 64);
  for (int i = 0; i < 6; ++i) sp[i] = nullptr;
  sp[6] = (void*)&fiber_entry;
  sp[7] = nullptr;
  f.sp = sp;
  f.done = false;
}

}  // namespace

int linear_tid() { return (int)(g_threadIdx.x + g_blockDim.x * (g_threadIdx.y + g_blockDim.y * g_threadIdx.z)); }

void yield_barrier() { hipmock_ctx_switch(&g_fibers[g_current].sp, g_sched_sp); }

void run_grid(dim3 grid, dim3 block, const std::function<void()>& body) {
  const int nthreads = (int)(block.x * block.y * block.z);
  if (nthreads > kMaxThreads || nthreads <= 0) {
    fprintf(stderr, "hipmock: bad block size %d\n", nthreads);
    abort();
  }
  g_blockDim = block;
  g_gridDim = grid;
  g_body = &body;
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        g_blockIdx = uint3_t{bx, by, bz};
        int t = 0;
        for (unsigned tz = 0; tz < block.z; ++tz)
          for (unsigned ty = 0; ty < block.y; ++ty)
            for (unsigned tx = 0; tx < block.x; ++tx, ++t) {
              prepare(g_fibers[t]);
              g_fibers[t].tid = uint3_t{tx, ty, tz};
            }
        int live = nthreads;
        while (live > 0) {
          live = 0;
          for (int i = 0; i < nthreads; ++i) {
            Fiber& f = g_fibers[i];
            if (f.done) continue;
            g_current = i;
            g_threadIdx = f.tid;
            hipmock_ctx_switch(&g_sched_sp, f.sp);
            if (!f.done) ++live;
          }
        }
      }
  g_body = nullptr;
}

}  // namespace hipmock

hipError_t hipMalloc(void** p, size_t bytes) {
  *p = malloc(bytes ? bytes : 1);
  return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipFree(void* p) {
  free(p);
  return hipSuccess;
}
hipError_t hipMemcpy(void* dst, const void* src, size_t bytes, hipMemcpyKind) {
  memmove(dst, src, bytes);
  return hipSuccess;
}
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind k, hipStream_t) { return hipMemcpy(dst, src, bytes, k); }
hipError_t hipMemset(void* dst, int value, size_t bytes) {
  memset(dst, value, bytes);
  return hipSuccess;
}
hipError_t hipMemsetAsync(void* dst, int value, size_t bytes, hipStream_t) { return hipMemset(dst, value, bytes); }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetDevice(int* d) {
  *d = 0;
  return hipSuccess;
}
hipError_t hipGetDeviceCount(int* n) {
  *n = 1;
  return hipSuccess;
}
hipError_t hipGetDeviceProperties(hipDeviceProp_t* prop, int) {
  memset(prop, 0, sizeof(*prop));
  snprintf(prop->name, sizeof(prop->name), "hipmock CPU emulator");
  snprintf(prop->gcnArchName, sizeof(prop->gcnArchName), "cpu-fibers");
  prop->multiProcessorCount = 4;
  return hipSuccess;
}
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "success" : "hipmock error"; }
hipError_t hipGetLastError() { return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) {
  *e = new mock_event();
  return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) {
  e->t = std::chrono::steady_clock::now();
  return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
  *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
  return hipSuccess;
}
hipError_t hipEventDestroy(hipEvent_t e) {
  delete e;
  return hipSuccess;
}
