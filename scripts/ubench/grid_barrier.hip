// Latency of a barrier among G co-resident workgroups on gfx950 (the level barrier of a multi-workgroup triangular solve):
// every workgroup stores a value, passes the barrier, reads the value of its neighbour and checks it -- so the timing includes
// making ordinary global stores visible across workgroups (and across XCDs: each XCD has its own L2).
//   mode 0: workgroups on consecutive ids (spread over the 8 XCDs)        mode 1: only ids = 0 mod 8 work (all on ONE XCD)
// Usage: grid_barrier      (prints us per barrier for G = 4, 8, 16, 32 in both modes, and the time per dependent kernel launch)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ bool group_barrier(int* counter, int target) {      // false: gave up (would have hung)
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    int spins = 0;
    while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1 << 22)) { ok = false; break; }
    }
  }
  __syncthreads();
  return ok;
}

__global__ __launch_bounds__(512) void k(int* counter, double* buf, int G, int stride, int iters, int* bad) {
  if ((int)blockIdx.x % stride != 0) return;
  const int g = (int)blockIdx.x / stride;
  for (int it = 0; it < iters; ++it) {
    if (threadIdx.x < 64) buf[(size_t)g * 64 + threadIdx.x] = (double)(it * 1000 + g);
    if (!group_barrier(counter, G * (2 * it + 1))) { if (threadIdx.x == 0) atomicAdd(bad, 1000000); return; }
    const int nb = (g + 1) % G;
    if (threadIdx.x < 64) {
      const double v = buf[(size_t)nb * 64 + threadIdx.x];
      if (v != (double)(it * 1000 + nb)) atomicAdd(bad, 1);
    }
    if (!group_barrier(counter, G * (2 * it + 2))) { if (threadIdx.x == 0) atomicAdd(bad, 1000000); return; }      // nobody overwrites before all have read
  }
}
__global__ void tiny(double* buf) { if (threadIdx.x == 0) buf[blockIdx.x] += 1.0; }

int main() {
  int* counter; double* buf; int* bad;
  CHECK(hipMalloc(&counter, 4)); CHECK(hipMalloc(&buf, 8 * 64 * 64)); CHECK(hipMalloc(&bad, 4));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int iters = 2000;
  for (int mode = 0; mode < 2; ++mode) {
    for (int G = 4; G <= 32; G *= 2) {
      const int stride = mode ? 8 : 1;
      float best = 1e9f; int hb = 0;
      for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipMemset(counter, 0, 4)); CHECK(hipMemset(bad, 0, 4));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(G * stride), dim3(512), 0, 0, counter, buf, G, stride, iters, bad);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        CHECK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
      }
      printf("mode %d (%s)  G = %2d: %.2f us per barrier (store -> barrier -> neighbour's load), mismatches %d\n", mode,
             mode ? "one XCD" : "all XCDs", G, best * 1000.f / (2 * iters), hb);
    }
  }
  CHECK(hipEventRecord(e0));
  for (int it = 0; it < 2000; ++it) hipLaunchKernelGGL(tiny, dim3(16), dim3(512), 0, 0, buf);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  printf("dependent launches of a 16-workgroup kernel: %.2f us each\n", ms * 1000.f / 2000);
  return 0;
}
