// TEST INFRASTRUCTURE — a host-only stand-in for <hip/hip_runtime.h>.
//
// It lets g++ compile the UNMODIFIED kernel sources of multigridcmt_amd/csrc/*.hip into
// tests/_build/libmgcmt_emu.so so that kernel logic (indexing, barriers, shuffles) and the whole
// host layer can be exercised in a container without a GPU.  It is never used by the product:
// multigridcmt_amd loads libmgcmt_hip.so (hipcc, gfx950) and raises if that is missing; only tests
// point the loader at the emulated library explicitly.
//
// Model: one workgroup at a time; its threads are fibers on one OS thread.  A fiber runs until it
// reaches __syncthreads() / a wave shuffle (or returns); when every live fiber of the workgroup has
// stopped there, all resume.  Waves are 64 consecutive threads, as on gfx950.
#pragma once

#include <chrono>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)
#define HIP_KERNEL_NAME(...) __VA_ARGS__

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2 };
typedef void* hipStream_t;
struct mock_event { std::chrono::steady_clock::time_point t; };
typedef mock_event* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct double2 {
  double x, y;
};
inline double2 make_double2(double x, double y) { return double2{x, y}; }
struct hipDeviceProp_t {
  char name[256];
  char gcnArchName[256];
  int multiProcessorCount;
};

namespace hipmock {
struct uint3_t { unsigned x, y, z; };
extern uint3_t g_threadIdx, g_blockIdx;
extern dim3 g_blockDim, g_gridDim;
extern double g_xchg_d[1024];
extern long long g_xchg_i[1024];
void yield_barrier();
void run_grid(dim3 grid, dim3 block, const std::function<void()>& body);
int linear_tid();
}  // namespace hipmock

#define threadIdx (hipmock::g_threadIdx)
#define blockIdx (hipmock::g_blockIdx)
#define blockDim (hipmock::g_blockDim)
#define gridDim (hipmock::g_gridDim)
static const int warpSize = 64;

inline void __syncthreads() { hipmock::yield_barrier(); }

// wave64 shuffles: every live thread of the workgroup takes part (as in the kernels of this repo)
template <typename T>
inline T hipmock_shfl_from(T v, int src_lane_in_wave, bool valid) {
  const int tid = hipmock::linear_tid();
  double* slot = hipmock::g_xchg_d;
  static_assert(sizeof(T) <= sizeof(double), "shuffle payload too large");
  std::memcpy(&slot[tid], &v, sizeof(T));
  hipmock::yield_barrier();
  T out = v;
  if (valid) std::memcpy(&out, &slot[(tid & ~63) + src_lane_in_wave], sizeof(T));
  hipmock::yield_barrier();
  return out;
}
template <typename T>
inline T __shfl_up(T v, unsigned d, int width = 64) {
  const int lane = hipmock::linear_tid() & 63;
  return hipmock_shfl_from(v, lane - (int)d, (lane % width) >= (int)d);
}
template <typename T>
inline T __shfl_down(T v, unsigned d, int width = 64) {
  const int lane = hipmock::linear_tid() & 63;
  return hipmock_shfl_from(v, lane + (int)d, (lane % width) + (int)d < width);
}
template <typename T>
inline T __shfl_xor(T v, int mask, int width = 64) {
  const int lane = hipmock::linear_tid() & 63;
  return hipmock_shfl_from(v, lane ^ mask, true);
}
template <typename T>
inline T __shfl(T v, int src, int width = 64) {
  const int lane = hipmock::linear_tid() & 63;
  return hipmock_shfl_from(v, (lane / width) * width + (src % width), true);
}

// the raw cross-lane read behind the shuffles: lane i receives `v` of lane (byte_addr / 4) % 64
inline int __builtin_amdgcn_ds_bpermute(int byte_addr, int v) { return hipmock_shfl_from(v, (byte_addr >> 2) & 63, true); }

template <typename... KArgs, typename... Args>
inline void hipLaunchKernelGGL(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t, hipStream_t, Args... args) {
  hipmock::run_grid(grid, block, [=]() { kernel(args...); });
}

inline void __threadfence() {}
inline unsigned atomicAdd(unsigned* p, unsigned x) { const unsigned old = *p; *p = old + x; return old; }  // one fiber runs at a time

hipError_t hipMalloc(void** p, size_t bytes);
hipError_t hipFree(void* p);
hipError_t hipMemcpy(void* dst, const void* src, size_t bytes, hipMemcpyKind kind);
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t s);
hipError_t hipMemset(void* dst, int value, size_t bytes);
hipError_t hipMemsetAsync(void* dst, int value, size_t bytes, hipStream_t s);
hipError_t hipSetDevice(int d);
hipError_t hipGetDevice(int* d);
template <typename F>
inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* blocks, F, int, size_t) {
  *blocks = 2;
  return hipSuccess;
}
hipError_t hipGetDeviceCount(int* n);
hipError_t hipGetDeviceProperties(hipDeviceProp_t* prop, int device);
const char* hipGetErrorString(hipError_t e);
hipError_t hipGetLastError();
hipError_t hipStreamSynchronize(hipStream_t s);
// graphs are not emulated: capture reports failure and the library falls back to eager launches
typedef void* hipGraph_t;
typedef void* hipGraphExec_t;
enum hipStreamCaptureMode { hipStreamCaptureModeGlobal = 0, hipStreamCaptureModeThreadLocal = 1, hipStreamCaptureModeRelaxed = 2 };
inline hipError_t hipStreamCreate(hipStream_t* s) { *s = (hipStream_t)1; return hipSuccess; }
enum { hipStreamDefault = 0, hipStreamNonBlocking = 1 };
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (hipStream_t)1; return hipSuccess; }
inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = (hipStream_t)1; return hipSuccess; }
inline hipError_t hipDeviceGetStreamPriorityRange(int* lo, int* hi) { *lo = 0; *hi = -1; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { return hipErrorInvalidValue; }
inline hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) { *g = nullptr; return hipErrorInvalidValue; }
inline hipError_t hipGraphInstantiate(hipGraphExec_t*, hipGraph_t, void*, void*, size_t) { return hipErrorInvalidValue; }
inline hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return hipErrorInvalidValue; }
inline hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }
inline hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }
hipError_t hipDeviceSynchronize();
hipError_t hipEventCreate(hipEvent_t* e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);
hipError_t hipEventDestroy(hipEvent_t e);
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }  // everything is synchronous here
enum { hipEventDefault = 0, hipEventDisableTiming = 2 };
enum { hipErrorNotReady = 600 };
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
