// How long does a word stored by ONE workgroup take to be seen by the polling workgroups of the same launch (gfx950,
// 8 XCDs with an L2 each)?  977 workgroups x 256 threads; in round r workgroup (r * 37) % 977 stores r + 1 into word
// r % 2 of `flag` at wall-clock time t_store[r]; thread 0 of every other workgroup polls until it sees it and records
// when.  Variants: store scope x load scope, optionally an agent-scope acquire fence (buffer_inv sc1) per poll.
//   hipcc --offload-arch=gfx950 -O3 -o signal_latency signal_latency.hip && ./signal_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
#define WGS 977
#define ROUNDS 24
template <int LOAD, int STORE, bool INV, bool ALLTHREADS>
__global__ void __launch_bounds__(256) k_sig(uint32_t* flag, uint64_t* t_store, uint64_t* t_seen, uint32_t* done) {
  for (int r = 0; r < ROUNDS; ++r) {
    const int src = (r * 37) % WGS;
    uint32_t* f = flag + (r & 1) * 64;
    if ((int)blockIdx.x == src) {
      // wait until everybody has arrived in this round (their arrival words), so that the store finds them all polling
      if (threadIdx.x == 0) {
        for (int w = 0; w < WGS; ++w) {
          if (w == src) continue;
          while (__hip_atomic_load(done + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != (uint32_t)r) __builtin_amdgcn_s_sleep(4);
        }
        t_store[r] = wall_clock64();
        if (STORE == 1) __hip_atomic_store(f, (uint32_t)(r + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else __hip_atomic_store(f, (uint32_t)(r + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    } else {
      if (threadIdx.x == 0) __hip_atomic_store(done + blockIdx.x, (uint32_t)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (ALLTHREADS || threadIdx.x == 0) {
        uint32_t v;
        do {
          if (INV) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          if (LOAD == 1) v = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          else v = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          if (v != (uint32_t)(r + 1)) __builtin_amdgcn_s_sleep(1);
        } while (v != (uint32_t)(r + 1));
        if (threadIdx.x == 0) t_seen[(size_t)r * WGS + blockIdx.x] = wall_clock64();
      }
    }
    __syncthreads();
  }
}
template <int LOAD, int STORE, bool INV, bool ALL>
int run(const char* name, uint32_t* flag, uint64_t* t_store, uint64_t* t_seen, uint32_t* done) {
  CK(hipMemset(flag, 0, 512)); CK(hipMemset(done, 0xff, WGS * 4)); CK(hipMemset(t_seen, 0, sizeof(uint64_t) * ROUNDS * WGS));
  hipLaunchKernelGGL((k_sig<LOAD, STORE, INV, ALL>), dim3(WGS), dim3(256), 0, 0, flag, t_store, t_seen, done);
  CK(hipDeviceSynchronize());
  std::vector<uint64_t> ts(ROUNDS), seen((size_t)ROUNDS * WGS);
  CK(hipMemcpy(ts.data(), t_store, ROUNDS * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(seen.data(), t_seen, seen.size() * 8, hipMemcpyDeviceToHost));
  std::vector<double> same, other;
  for (int r = 4; r < ROUNDS; ++r) {
    const int src = (r * 37) % WGS;
    for (int w = 0; w < WGS; ++w) {
      if (w == src) continue;
      const double us = ((double)seen[(size_t)r * WGS + w] - (double)ts[r]) / 100.0;
      ((w % 8) == (src % 8) ? same : other).push_back(us);
    }
  }
  auto q = [](std::vector<double>& v, double p) { std::sort(v.begin(), v.end()); return v[(size_t)(p * (v.size() - 1))]; };
  printf("{\"variant\": \"%s\", \"same_xcd_us\": {\"p50\": %.2f, \"p99\": %.2f, \"max\": %.2f}, \"other_xcd_us\": {\"p50\": %.2f, \"p99\": %.2f, \"max\": %.2f}}\n",
         name, q(same, .5), q(same, .99), q(same, 1.), q(other, .5), q(other, .99), q(other, 1.));
  fflush(stdout);
  return 0;
}
int main() {
  uint32_t *flag, *done; uint64_t *t_store, *t_seen;
  CK(hipMalloc(&flag, 512)); CK(hipMalloc(&done, WGS * 4)); CK(hipMalloc(&t_store, ROUNDS * 8)); CK(hipMalloc(&t_seen, sizeof(uint64_t) * ROUNDS * WGS));
  if (run<1, 1, false, false>("agent load / agent store, one poller per workgroup", flag, t_store, t_seen, done)) return 1;
  if (run<2, 2, false, false>("system load / system store, one poller per workgroup", flag, t_store, t_seen, done)) return 1;
  if (run<1, 1, true, false>("agent load + acquire fence / agent store, one poller", flag, t_store, t_seen, done)) return 1;
  if (run<1, 1, false, true>("agent load / agent store, 256 pollers per workgroup", flag, t_store, t_seen, done)) return 1;
  if (run<2, 2, false, true>("system load / system store, 256 pollers per workgroup", flag, t_store, t_seen, done)) return 1;
  return 0;
}
