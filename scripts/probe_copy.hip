// Why does the library's copy probe top out at 5.6 TB/s when MI355X_MICROARCH.md quotes 6.29 TB/s for a float4 copy?
// Variants of a 16-byte-per-lane copy of 2 GiB -> 2 GiB: grid-stride with few / many blocks, one element per thread,
// non-temporal loads / stores, four loads in flight per lane, 1024-thread blocks.  Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v2d __attribute__((ext_vector_type(2)));

__global__ void copy_stride(long n2, const v2d* __restrict__ a, v2d* __restrict__ o) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (long)gridDim.x * blockDim.x) o[i] = a[i];
}
__global__ void copy_stride_nts(long n2, const v2d* __restrict__ a, v2d* __restrict__ o) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (long)gridDim.x * blockDim.x) __builtin_nontemporal_store(a[i], &o[i]);
}
__global__ void copy_stride_ntls(long n2, const v2d* __restrict__ a, v2d* __restrict__ o) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (long)gridDim.x * blockDim.x)
    __builtin_nontemporal_store(__builtin_nontemporal_load(&a[i]), &o[i]);
}
__global__ void copy_one(long n2, const v2d* __restrict__ a, v2d* __restrict__ o) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n2) o[i] = a[i];
}
__global__ void copy_one_nts(long n2, const v2d* __restrict__ a, v2d* __restrict__ o) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n2) __builtin_nontemporal_store(a[i], &o[i]);
}
// a block owns a contiguous chunk; four loads in flight per lane
__global__ void copy_chunk4(long n2, const v2d* __restrict__ a, v2d* __restrict__ o, long per_block) {
  const long b0 = (long)blockIdx.x * per_block, b1 = b0 + per_block < n2 ? b0 + per_block : n2;
  for (long i = b0 + threadIdx.x; i < b1; i += 4L * blockDim.x) {
    v2d x0 = a[i], x1, x2, x3;
    const bool h1 = i + blockDim.x < b1, h2 = i + 2L * blockDim.x < b1, h3 = i + 3L * blockDim.x < b1;
    if (h1) x1 = a[i + blockDim.x];
    if (h2) x2 = a[i + 2L * blockDim.x];
    if (h3) x3 = a[i + 3L * blockDim.x];
    __builtin_nontemporal_store(x0, &o[i]);
    if (h1) __builtin_nontemporal_store(x1, &o[i + blockDim.x]);
    if (h2) __builtin_nontemporal_store(x2, &o[i + 2L * blockDim.x]);
    if (h3) __builtin_nontemporal_store(x3, &o[i + 3L * blockDim.x]);
  }
}
__global__ void read_only(long n2, const v2d* __restrict__ a, double* sink) {
  double acc = 0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (long)gridDim.x * blockDim.x) { v2d x = a[i]; acc += x.x + x.y; }
  if (acc == 1.2345) sink[0] = acc;
}
__global__ void write_only(long n2, v2d* __restrict__ o) {
  v2d z = {1.0, 2.0};
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (long)gridDim.x * blockDim.x) __builtin_nontemporal_store(z, &o[i]);
}

int main() {
  const long n = 1L << 28;  // doubles: 2 GiB
  const long n2 = n / 2;
  double *a, *o;
  if (hipMalloc(&a, n * 8) != hipSuccess || hipMalloc(&o, n * 8) != hipSuccess) return 1;
  hipMemset(a, 1, n * 8);
  hipMemset(o, 0, n * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  auto time = [&](const char* name, double bytes, auto launch) {
    for (int w = 0; w < 3; ++w) launch();
    hipEventRecord(e0, 0);
    const int reps = 20;
    for (int r = 0; r < reps; ++r) launch();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %8.3f ms  %6.2f TB/s\n", name, ms / reps, bytes / (ms / reps * 1e-3) / 1e12);
    fflush(stdout);
  };
  const double cb = 2.0 * n * 8;
  for (int blocks : {256, 1024, 2048, 4096, 8192, 16384, 65536})  {
    char nm[64];
    snprintf(nm, 64, "copy grid-stride, %d blocks x 256", blocks);
    time(nm, cb, [&] { hipLaunchKernelGGL(copy_stride, dim3(blocks), dim3(256), 0, 0, n2, (const v2d*)a, (v2d*)o); });
  }
  time("copy grid-stride nt stores, 1024 x 256", cb, [&] { hipLaunchKernelGGL(copy_stride_nts, dim3(1024), dim3(256), 0, 0, n2, (const v2d*)a, (v2d*)o); });
  time("copy grid-stride nt stores, 8192 x 256", cb, [&] { hipLaunchKernelGGL(copy_stride_nts, dim3(8192), dim3(256), 0, 0, n2, (const v2d*)a, (v2d*)o); });
  time("copy grid-stride nt loads+stores, 8192 x 256", cb, [&] { hipLaunchKernelGGL(copy_stride_ntls, dim3(8192), dim3(256), 0, 0, n2, (const v2d*)a, (v2d*)o); });
  time("copy one element per thread, x 256", cb, [&] { hipLaunchKernelGGL(copy_one, dim3((unsigned)(n2 / 256)), dim3(256), 0, 0, n2, (const v2d*)a, (v2d*)o); });
  time("copy one element per thread, x 1024", cb, [&] { hipLaunchKernelGGL(copy_one, dim3((unsigned)(n2 / 1024)), dim3(1024), 0, 0, n2, (const v2d*)a, (v2d*)o); });
  time("copy one element per thread nt stores, x 256", cb, [&] { hipLaunchKernelGGL(copy_one_nts, dim3((unsigned)(n2 / 256)), dim3(256), 0, 0, n2, (const v2d*)a, (v2d*)o); });
  for (int blocks : {1024, 4096, 16384}) {
    char nm[64];
    snprintf(nm, 64, "copy contiguous chunks, 4 in flight, %d", blocks);
    const long per = (n2 + blocks - 1) / blocks;
    time(nm, cb, [&] { hipLaunchKernelGGL(copy_chunk4, dim3(blocks), dim3(256), 0, 0, n2, (const v2d*)a, (v2d*)o, per); });
  }
  time("read only grid-stride 8192 x 256", n * 8.0, [&] { hipLaunchKernelGGL(read_only, dim3(8192), dim3(256), 0, 0, n2, (const v2d*)a, o); });
  time("write only nt grid-stride 8192 x 256", n * 8.0, [&] { hipLaunchKernelGGL(write_only, dim3(8192), dim3(256), 0, 0, n2, (v2d*)o); });
  // float4 of the guide: the same bytes as 4 floats
  return 0;
}
