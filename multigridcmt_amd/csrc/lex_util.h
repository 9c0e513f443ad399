// Device helpers shared by the two lexicographic pipelines (kernels_lexwave.hip: skewed column blocks with a scan per
// row; kernels_lexband.hip: row bands swept as a wavefront): hand-counted asm loads, DPP lane shifts, agent-scope
// granule loads / stores for the inter-workgroup hand-off, compile-time slot loops.  Host-side stand-ins for the
// emulation build of the tests (one workgroup at a time, in block order: a block never has to wait).
#pragma once
#include "mgcmt_internal.h"

namespace mgcmt {

namespace {

typedef unsigned long long u64;

// Loads of the row loop are hand-counted: issued as asm (the compiler does not see them, so it neither waits for them nor
// drains them), consumed behind ONE s_waitcnt vmcnt(N) per row whose N = the memory operations issued since the youngest
// value that row needs (vector-memory operations complete in order, stores included).  The destinations are read-write
// operands, so a slot keeps its registers; the wait names every register it releases, so no use moves above it
// (cdna_hip_programming.md §5.7, form (ii)).
#if defined(__HIP_DEVICE_COMPILE__)
// value known to be the same in every lane -> a scalar register (everything derived from it runs on the scalar unit)
__device__ __forceinline__ int uniform(int x) { return __builtin_amdgcn_readfirstlane(x); }
// x clamped into [0, hi] in one instruction (hi wave-uniform, hi >= 0)
__device__ __forceinline__ int clamp0(int x, int hi) {
  int r;
  asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(x), "s"(hi));
  return r;
}
// load at (wave-uniform base) + (32-bit per-lane byte offset) [+ immediate]: no 64-bit address arithmetic per lane
#define MGCMT_LEX_LOAD_AT(dst, off, base, imm) \
  asm volatile("global_load_dwordx2 %0, %1, %2 offset:" #imm : "+v"(dst) : "v"(off), "s"(base) : "memory")
#define MGCMT_LEX_LOAD_AT_SC1(dst, off, base) \
  asm volatile("global_load_dwordx2 %0, %1, %2 sc1" : "+v"(dst) : "v"(off), "s"(base) : "memory")
#define MGCMT_LEX_LOAD_AT_SC1I(dst, off, base, imm) \
  asm volatile("global_load_dwordx2 %0, %1, %2 offset:" #imm " sc1" : "+v"(dst) : "v"(off), "s"(base) : "memory")
#else
__device__ __forceinline__ int uniform(int x) { return x; }
__device__ __forceinline__ int clamp0(int x, int hi) { return x < 0 ? 0 : (x > hi ? hi : x); }
#define MGCMT_LEX_LOAD_AT(dst, off, base, imm) \
  (dst) = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(base) + (off) + (imm))
#define MGCMT_LEX_LOAD_AT_SC1(dst, off, base) \
  (dst) = *reinterpret_cast<const unsigned long long*>(reinterpret_cast<const char*>(base) + (off))
#define MGCMT_LEX_LOAD_AT_SC1I(dst, off, base, imm) MGCMT_LEX_LOAD_AT(dst, off, base, imm)
#endif

#if defined(__HIP_DEVICE_COMPILE__)
#define MGCMT_LEX_LOAD(dst, ptr) asm volatile("global_load_dwordx2 %0, %1, off" : "+v"(dst) : "v"(ptr) : "memory")
#define MGCMT_LEX_LOAD_SC1(dst, ptr) asm volatile("global_load_dwordx2 %0, %1, off sc1" : "+v"(dst) : "v"(ptr) : "memory")
template <int N>
__device__ __forceinline__ void wait_loads(double& a, double& b, double& c, double& d, double& e, double& f, double& g, unsigned long long& r) {
  asm volatile("s_waitcnt vmcnt(%8)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(r) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_loads3(double& a, double& b, unsigned long long& r) {
  asm volatile("s_waitcnt vmcnt(%3)" : "+v"(a), "+v"(b), "+v"(r) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_loads9(double& a, double& b, double& c, double& d, double& e, double& f, double& g, unsigned long long& r,
                                            unsigned long long& r2) {
  asm volatile("s_waitcnt vmcnt(%9)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(r), "+v"(r2) : "n"(N) : "memory");
}
__device__ __forceinline__ void drain_loads() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// the value of a slot that has landed, in a register of its own: a value that outlives the step (a stream history) must
// not go on living in the slot's register, or the refill — which needs that register — moves the slot somewhere else, the
// slots start rotating through registers, and the copies that put them back at the loop's latch read loads in flight
__device__ __forceinline__ double take(const double& slot) {
  double r;
  asm volatile("v_mov_b64 %0, %1" : "=&v"(r) : "v"(slot));
  return r;
}
// a use the compiler sees: whatever load it believes pending on this register is waited for HERE, once (see the prologue)
__device__ __forceinline__ void settle(double& x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void settle(unsigned long long& x) { asm volatile("" : "+v"(x)); }
#else
#define MGCMT_LEX_LOAD(dst, ptr) (dst) = *(ptr)
#define MGCMT_LEX_LOAD_SC1(dst, ptr) (dst) = *(ptr)
template <int N>
__device__ __forceinline__ void wait_loads(double&, double&, double&, double&, double&, double&, double&, unsigned long long&) {}
template <int N>
__device__ __forceinline__ void wait_loads3(double&, double&, unsigned long long&) {}
template <int N>
__device__ __forceinline__ void wait_loads9(double&, double&, double&, double&, double&, double&, double&, unsigned long long&, unsigned long long&) {}
__device__ __forceinline__ void drain_loads() {}
__device__ __forceinline__ double take(const double& slot) { return slot; }
__device__ __forceinline__ void settle(double&) {}
__device__ __forceinline__ void settle(unsigned long long&) {}
#endif

#if defined(__HIP_DEVICE_COMPILE__)
template <int CTRL>
__device__ __forceinline__ double dpp(double v) {
  const u64 u = __builtin_bit_cast(u64, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)u, CTRL, 0xf, 0xf, true);
  const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTRL, 0xf, 0xf, true);
  return __builtin_bit_cast(double, ((u64)hi << 32) | lo);
}
template <int D>
__device__ __forceinline__ double row_shr(double v, int) { return dpp<0x110 + D>(v); }    // lane - D inside rows of 16 (else 0)
__device__ __forceinline__ double bcast15(double v, int) { return dpp<0x142>(v); }         // lane 15 of the previous row of 16
__device__ __forceinline__ double bcast31(double v, int) { return dpp<0x143>(v); }         // lane 31
__device__ __forceinline__ double quad_pairs(double v, int) { return dpp<0xFA>(v); }      // lanes 4k..4k+3 <- lanes 4k+2, 4k+2, 4k+3, 4k+3
__device__ __forceinline__ double from_left(double v, int) { return dpp<0x138>(v); }       // lane - 1 (lane 0: 0)
__device__ __forceinline__ double from_right(double v, int) { return dpp<0x130>(v); }      // lane + 1 (lane 63: 0)
__device__ __forceinline__ u64 lane_bits(u64 u, int k) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, k);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), k);
  return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ unsigned lane_word(unsigned x, int k) { return (unsigned)__builtin_amdgcn_readlane((int)x, k); }
__device__ __forceinline__ u64 vote(bool x) { return __ballot(x); }
__device__ __forceinline__ u64 vote_eq(unsigned a, unsigned b) { return __builtin_amdgcn_uicmp(a, b, 32 /* ICMP_EQ */); }  // lanes with a == b
__device__ __forceinline__ unsigned load_word(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void store_word(unsigned* p, unsigned x) { __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u64 load_granule(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void store_granule(u64* p, u64 x) { __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// a double another workgroup of the same launch may have written / will read: agent scope (write-through, past the caches)
template <bool SHARED>
__device__ __forceinline__ double load_value(const char* p) {
  if (SHARED) return __builtin_bit_cast(double, __hip_atomic_load(reinterpret_cast<const u64*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  return *reinterpret_cast<const double*>(p);
}
template <bool SHARED>
__device__ __forceinline__ void store_value(char* p, double x) {
  if (SHARED) __hip_atomic_store(reinterpret_cast<u64*>(p), __builtin_bit_cast(u64, x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *reinterpret_cast<double*>(p) = x;
}
__device__ __forceinline__ u64 now_ticks() { return wall_clock64(); }  // 100 MHz
__device__ __forceinline__ void nap() { __builtin_amdgcn_s_sleep(1); }
#else
// host-side stand-ins (the emulator runs one workgroup at a time, in block order: a block never has to wait)
template <int D>
__device__ __forceinline__ double row_shr(double v, int lane) { const double r = __shfl_up(v, D); return (lane & 15) >= D ? r : 0.0; }
__device__ __forceinline__ double bcast15(double v, int lane) { return __shfl(v, ((lane & ~15) - 1) & 63); }
__device__ __forceinline__ double bcast31(double v, int) { return __shfl(v, 31); }
__device__ __forceinline__ double quad_pairs(double v, int lane) { return __shfl(v, (lane & ~3) + 2 + ((lane & 3) >> 1)); }
__device__ __forceinline__ double from_left(double v, int lane) { const double r = __shfl_up(v, 1); return lane >= 1 ? r : 0.0; }
__device__ __forceinline__ double from_right(double v, int lane) { const double r = __shfl_down(v, 1); return lane <= 62 ? r : 0.0; }
__device__ __forceinline__ u64 lane_bits(u64 u, int k) { return (u64)__shfl((long long)u, k); }
__device__ __forceinline__ unsigned lane_word(unsigned x, int k) { return (unsigned)__shfl((int)x, k); }
__device__ __forceinline__ u64 vote(bool x) {
  u64 m = 0;
  for (int k = 0; k < 64; ++k) m |= (u64)(__shfl(x ? 1 : 0, k) & 1) << k;
  return m;
}
__device__ __forceinline__ u64 vote_eq(unsigned a, unsigned b) { return vote(a == b); }
__device__ __forceinline__ unsigned load_word(const unsigned* p) { return *p; }
__device__ __forceinline__ void store_word(unsigned* p, unsigned x) { *p = x; }
__device__ __forceinline__ u64 load_granule(const u64* p) { return *p; }
__device__ __forceinline__ void store_granule(u64* p, u64 x) { *p = x; }
template <bool SHARED>
__device__ __forceinline__ double load_value(const char* p) { return *reinterpret_cast<const double*>(p); }
template <bool SHARED>
__device__ __forceinline__ void store_value(char* p, double x) { *reinterpret_cast<double*>(p) = x; }
__device__ __forceinline__ u64 now_ticks() { return 0; }
__device__ __forceinline__ void nap() {}
#endif

template <bool B>
struct Checked {
  static constexpr bool value = B;
};
template <int N>
struct Int {
  static constexpr int value = N;
};
template <int N, int I = 0, class F>
__device__ __forceinline__ void for_slots(F&& f) {
  if constexpr (I < N) {
    f(Int<I>{});
    for_slots<N, I + 1>(f);
  }
}

constexpr u64 kTimeoutTicks = 200000000ull;  // 2 s of the 100 MHz counter: a stuck pipeline gives up

}  // namespace

}  // namespace mgcmt
