// calib.hip — machine calibration micro-benchmarks (not part of the product):
// launch overhead, copy bandwidth, dependent-FMA issue rate, integer-multiply rate.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void k_empty() {}
__global__ void k_copy4(const float4* __restrict__ a, float4* __restrict__ b, int64_t n4) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n4) b[i] = a[i];
}
__global__ void k_copy1(const float* __restrict__ a, float* __restrict__ b, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) b[i] = a[i];
}
__global__ void k_gather1(const float* __restrict__ a, const int* __restrict__ idx, float* __restrict__ b, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) b[i] = a[idx[i]];
}
template <int ITERS>
__global__ void k_fma(float* out, float s, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  float x = (float)i * 1e-9f;
#pragma unroll 16
  for (int k = 0; k < ITERS; ++k) x = __builtin_fmaf(x, s, 0.5f);
  if (i < n) out[i] = x;
}
template <int ITERS, int CH>
__global__ void k_fma_ilp(float* out, float s, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  float x[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) x[c] = (float)(i + c) * 1e-9f;
#pragma unroll 8
  for (int k = 0; k < ITERS; ++k) {
#pragma unroll
    for (int c = 0; c < CH; ++c) x[c] = __builtin_fmaf(x[c], s, 0.5f);
  }
  float r = 0;
#pragma unroll
  for (int c = 0; c < CH; ++c) r += x[c];
  if (i < n) out[i] = r;
}
template <int ITERS, int CH>
__global__ void k_tf_ilp(uint32_t* out, uint32_t s, int64_t n) {
  // threefry-like dependent chain: add, rotate, xor
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  uint32_t x0[CH], x1[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) { x0[c] = (uint32_t)i + c; x1[c] = s + c; }
#pragma unroll 8
  for (int k = 0; k < ITERS; ++k) {
#pragma unroll
    for (int c = 0; c < CH; ++c) { x0[c] += x1[c]; x1[c] = (x1[c] << 13) | (x1[c] >> 19); x1[c] ^= x0[c]; }
  }
  uint32_t r = 0;
#pragma unroll
  for (int c = 0; c < CH; ++c) r += x0[c] ^ x1[c];
  if (i < n) out[i] = r;
}
// which integer op limits a Threefry round?  MODE 0: add only, 1: xor only, 2: alignbit only,
// 3: shift+shift+or rotate, 4: full round with alignbit, 5: full round with shift/or rotate
template <int ITERS, int MODE>
__global__ void k_intop(uint32_t* out, uint32_t s, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  uint32_t x0 = (uint32_t)i, x1 = s ^ (uint32_t)i;
#pragma unroll 16
  for (int k = 0; k < ITERS; ++k) {
    if (MODE == 0) { x0 += x1; }
    else if (MODE == 1) { x0 ^= x1; x1 ^= s; }
    else if (MODE == 2) { x1 = __builtin_amdgcn_alignbit(x1, x1, 19); }
    else if (MODE == 3) { x1 = (x1 << 13) | (x1 >> 19); asm volatile("" : "+v"(x1)); }
    else if (MODE == 4) { x0 += x1; x1 = __builtin_amdgcn_alignbit(x1, x1, 19); x1 ^= x0; }
    else { x0 += x1; uint32_t t = x1 >> 19; asm volatile("" : "+v"(t)); x1 = (x1 << 13) | t; x1 ^= x0; }
  }
  if (i < n) out[i] = x0 ^ x1;
}
template <int ITERS>
__global__ void k_imul(uint32_t* out, uint32_t s, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  uint32_t x = (uint32_t)i;
#pragma unroll 16
  for (int k = 0; k < ITERS; ++k) x = x * s + 12345u;
  if (i < n) out[i] = x;
}
template <int ITERS>
__global__ void k_salu(uint32_t* out, uint32_t s, int64_t n) {
  // wave-uniform integer chain: runs on the scalar unit
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  uint32_t x = __builtin_amdgcn_readfirstlane(blockIdx.x) + s;
#pragma unroll 16
  for (int k = 0; k < ITERS; ++k) x = (x ^ (x >> 3)) + s;
  if (i < n) out[i] = x;
}

template <class F> float time_us(F f, int reps) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  hipEventRecord(a, 0);
  for (int r = 0; r < reps; ++r) f();
  hipEventRecord(b, 0); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms * 1e3f / reps;
}

int main() {
  const int64_t n = 1000000;
  float *a, *b; uint32_t* u;
  CK(hipMalloc(&a, n * 4 * 64)); CK(hipMalloc(&b, n * 4 * 64)); CK(hipMalloc(&u, n * 4));
  CK(hipMemset(a, 0, n * 4 * 64));
  int grid = (int)((n + 255) / 256);
  printf("empty kernel (1 block)        : %8.2f us/launch\n", time_us([&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, 0); }, 500));
  printf("empty kernel (3907 blocks)    : %8.2f us/launch\n", time_us([&] { hipLaunchKernelGGL(k_empty, dim3(grid), dim3(256), 0, 0); }, 500));
  printf("copy 4 MB  (1e6 f32)          : %8.2f us\n", time_us([&] { hipLaunchKernelGGL(k_copy4, dim3((n / 4 + 255) / 256), dim3(256), 0, 0, (const float4*)a, (float4*)b, n / 4); }, 500));
  printf("copy 4 MB dword/lane (calib)  : %8.2f us\n", time_us([&] { hipLaunchKernelGGL(k_copy1, dim3(grid), dim3(256), 0, 0, a, b, n); }, 200));
  { int64_t m = n * 64; float t = time_us([&] { hipLaunchKernelGGL(k_copy1, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, 0, a, b, m); }, 20);
    printf("copy 256 MB dword/lane (calib): %8.2f us  = %.2f TB/s\n", t, 2.0 * m * 4 / t * 1e-6); }
  { int64_t m = n * 64 / 4; float t = time_us([&] { hipLaunchKernelGGL(k_copy4, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, 0, (const float4*)a, (float4*)b, m); }, 50);
    printf("copy 256 MB                   : %8.2f us  = %.2f TB/s (read+write)\n", t, 2.0 * n * 64 * 4 / t * 1e-6); }
  { float t = time_us([&] { hipLaunchKernelGGL(k_fma<1024>, dim3(grid), dim3(256), 0, 0, b, 0.999f, n); }, 100);
    printf("1024 dependent FMA x 1e6 thr  : %8.2f us  = %.2f T lane-ops/s\n", t, 1024.0 * n / t * 1e-6); }
  { float t = time_us([&] { hipLaunchKernelGGL((k_fma_ilp<1024, 2>), dim3(grid), dim3(256), 0, 0, b, 0.999f, n); }, 100);
    printf("2 chains x 1024 FMA x 1e6 thr : %8.2f us  = %.2f T lane-ops/s\n", t, 2048.0 * n / t * 1e-6); }
  { float t = time_us([&] { hipLaunchKernelGGL((k_fma_ilp<1024, 4>), dim3(grid), dim3(256), 0, 0, b, 0.999f, n); }, 100);
    printf("4 chains x 1024 FMA x 1e6 thr : %8.2f us  = %.2f T lane-ops/s\n", t, 4096.0 * n / t * 1e-6); }
  { float t = time_us([&] { hipLaunchKernelGGL((k_fma_ilp<1024, 1>), dim3(grid * 4), dim3(256), 0, 0, b, 0.999f, n * 4); }, 100);
    printf("1 chain  x 1024 FMA x 4e6 thr : %8.2f us  = %.2f T lane-ops/s\n", t, 4096.0 * n / t * 1e-6); }
  { float t = time_us([&] { hipLaunchKernelGGL((k_tf_ilp<256, 1>), dim3(grid), dim3(256), 0, 0, u, 7u, n); }, 100);
    printf("1 chain  x 256 tf-rounds(3op) : %8.2f us  = %.2f T lane-ops/s\n", t, 768.0 * n / t * 1e-6); }
  { float t = time_us([&] { hipLaunchKernelGGL((k_tf_ilp<256, 2>), dim3(grid), dim3(256), 0, 0, u, 7u, n); }, 100);
    printf("2 chains x 256 tf-rounds(3op) : %8.2f us  = %.2f T lane-ops/s\n", t, 1536.0 * n / t * 1e-6); }
  { float t = time_us([&] { hipLaunchKernelGGL((k_intop<1024, 0>), dim3(grid), dim3(256), 0, 0, u, 7u, n); }, 100);
    printf("1024 dependent v_add_u32      : %8.2f us  = %.2f T lane-ops/s\n", t, 1024.0 * n / t * 1e-6); }
  { float t = time_us([&] { hipLaunchKernelGGL((k_intop<1024, 1>), dim3(grid), dim3(256), 0, 0, u, 7u, n); }, 100);
    printf("1024 x (2 xor)                : %8.2f us  = %.2f T lane-ops/s\n", t, 2048.0 * n / t * 1e-6); }
  { float t = time_us([&] { hipLaunchKernelGGL((k_intop<1024, 2>), dim3(grid), dim3(256), 0, 0, u, 7u, n); }, 100);
    printf("1024 dependent v_alignbit     : %8.2f us  = %.2f T lane-ops/s\n", t, 1024.0 * n / t * 1e-6); }
  { float t = time_us([&] { hipLaunchKernelGGL((k_intop<1024, 3>), dim3(grid), dim3(256), 0, 0, u, 7u, n); }, 100);
    printf("1024 x shift/shift/or rotate  : %8.2f us  = %.2f T rot/s\n", t, 1024.0 * n / t * 1e-6); }
  { float t = time_us([&] { hipLaunchKernelGGL((k_intop<256, 4>), dim3(grid), dim3(256), 0, 0, u, 7u, n); }, 100);
    printf("256 tf rounds (alignbit)      : %8.2f us  = %.2f T rounds/s\n", t, 256.0 * n / t * 1e-6); }
  { float t = time_us([&] { hipLaunchKernelGGL((k_intop<256, 5>), dim3(grid), dim3(256), 0, 0, u, 7u, n); }, 100);
    printf("256 tf rounds (shift/or)      : %8.2f us  = %.2f T rounds/s\n", t, 256.0 * n / t * 1e-6); }
  { float t = time_us([&] { hipLaunchKernelGGL(k_imul<1024>, dim3(grid), dim3(256), 0, 0, u, 2654435761u, n); }, 100);
    printf("1024 dependent IMAD x 1e6 thr : %8.2f us  = %.2f T lane-ops/s\n", t, 1024.0 * n / t * 1e-6); }
  { float t = time_us([&] { hipLaunchKernelGGL(k_salu<1024>, dim3(grid), dim3(256), 0, 0, u, 7u, n); }, 100);
    printf("2048 scalar ops x 15625 waves : %8.2f us  = %.2f G wave-ops/s\n", t, 2048.0 * (n / 64) / t * 1e-3); }
  return 0;
}
