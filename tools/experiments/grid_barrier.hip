// What does a grid-wide "everybody finished step s" cost inside ONE launch on gfx950 (977 workgroups x 256 threads, all
// resident)?  ROUNDS barriers back to back, nothing else; microseconds per barrier for:
//   tags      every workgroup stores its tag; every THREAD polls its 4 rows of the 977-row tag table (what the first
//             multi-step kernel did)
//   tags1w    the same table, polled by ONE wave per workgroup (16 rows per lane, dwordx4 loads), then a block barrier
//   counter   one agent-scope atomic add per workgroup on ONE word; thread 0 polls it
//   counter8  eight words (workgroup b adds to word b % 8); thread 0 polls the eight
//   two-level 31 groups of 32 workgroups: the first workgroup of a group waits for its group's 32 tags (one line) and
//             publishes the group's tag; everybody polls the 31 group tags (one line)
//   hipcc --offload-arch=gfx950 -O3 -o grid_barrier grid_barrier.hip && ./grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
#define WGS 977
#define ROUNDS 200
template <int MODE, int SLEEP>
__global__ void __launch_bounds__(256) k_bar(uint32_t* tags, uint32_t* ctr) {
  for (int r = 1; r <= ROUNDS; ++r) {
    if (MODE <= 1) {
      uint32_t* t = tags + (r & 1) * 1024;
      if (threadIdx.x == 0) __hip_atomic_store(t + blockIdx.x, (uint32_t)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (MODE == 0) {
        bool ok;
        do {
          ok = true;
          for (int k = 0; k < 4; ++k) {
            const int row = k * 256 + threadIdx.x;
            const int rc = row < WGS ? row : WGS - 1;
            ok &= __hip_atomic_load(t + rc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (uint32_t)r;
          }
          if (!ok) __builtin_amdgcn_s_sleep(SLEEP);
        } while (!ok);
      } else if (threadIdx.x < 64) {
        bool ok;
        do {
          ok = true;
          for (int k = 0; k < 16; ++k) {
            const int row = k * 64 + threadIdx.x;
            const int rc = row < WGS ? row : WGS - 1;
            ok &= __hip_atomic_load(t + rc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (uint32_t)r;
          }
          ok = __all(ok);
          if (!ok) __builtin_amdgcn_s_sleep(SLEEP);
        } while (!ok);
      }
    } else if (MODE == 4) {       // two levels: 31 groups of 32 workgroups; the group's first workgroup gathers its line
      uint32_t* t1 = tags + (r & 1) * 1024;
      uint32_t* t2 = tags + 2048 + (r & 1) * 64;
      const int g = blockIdx.x >> 5, groups = (WGS + 31) >> 5;
      if (threadIdx.x == 0) __hip_atomic_store(t1 + blockIdx.x, (uint32_t)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (threadIdx.x < 64) {
        if ((blockIdx.x & 31) == 0) {
          const int row = g * 32 + (threadIdx.x & 31);
          const int rc = row < WGS ? row : WGS - 1;
          bool ok;
          do {
            ok = __hip_atomic_load(t1 + rc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (uint32_t)r;
            ok = __all(ok);
            if (!ok) __builtin_amdgcn_s_sleep(SLEEP);
          } while (!ok);
          if (threadIdx.x == 0) __hip_atomic_store(t2 + g, (uint32_t)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const int gc = (int)threadIdx.x < groups ? (int)threadIdx.x : groups - 1;
        bool ok;
        do {
          ok = __hip_atomic_load(t2 + gc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (uint32_t)r;
          ok = __all(ok);
          if (!ok) __builtin_amdgcn_s_sleep(SLEEP);
        } while (!ok);
      }
    } else if (MODE == 2) {
      if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (uint32_t)r * WGS) __builtin_amdgcn_s_sleep(SLEEP);
      }
    } else {
      if (threadIdx.x == 0) __hip_atomic_fetch_add(ctr + 32 * (blockIdx.x & 7), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (threadIdx.x < 8) {
        const uint32_t per = (WGS - threadIdx.x + 7) / 8;     // workgroups b with b % 8 == lane
        bool ok;
        do {
          ok = __hip_atomic_load(ctr + 32 * threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (uint32_t)r * per;
          ok = __all(ok);
          if (!ok) __builtin_amdgcn_s_sleep(SLEEP);
        } while (!ok);
      }
    }
    __syncthreads();
  }
}
template <int MODE, int SLEEP> int run(const char* name, uint32_t* tags, uint32_t* ctr) {
  const int sleep = SLEEP;
  CK(hipMemset(tags, 0, 4096 * 4)); CK(hipMemset(ctr, 0, 1024));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((k_bar<MODE, SLEEP>), dim3(WGS), dim3(256), 0, 0, tags, ctr);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("{\"barrier\": \"%s\", \"s_sleep\": %d, \"us_per_barrier\": %.3f}\n", name, sleep, ms * 1e3f / ROUNDS); fflush(stdout);
  return 0;
}
int main() {
  uint32_t *tags, *ctr; CK(hipMalloc(&tags, 4096 * 4)); CK(hipMalloc(&ctr, 1024));
#define ALL(S) \
  if (run<0, S>("tags, every thread polls 4 rows", tags, ctr)) return 1; \
  if (run<1, S>("tags, one wave polls 16 rows per lane", tags, ctr)) return 1; \
  if (run<2, S>("one counter", tags, ctr)) return 1; \
  if (run<3, S>("eight counters", tags, ctr)) return 1; \
  if (run<4, S>("two-level tags (31 groups of 32)", tags, ctr)) return 1;
  ALL(1) ALL(8) ALL(32)
  return 0;
}
