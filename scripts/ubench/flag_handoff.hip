// Latency of ONE dependency hand-off through memory between workgroups on gfx950 -- the unit cost of a level of the barrier-free triangular
// solves (laplace_kernels.hip: lap_sptrsv_sf_kernel, "the data is the flag"): workgroup g polls the 8-byte slot of workgroup g - 1 until the
// value of the current lap arrives, then publishes its own.  A ring of K workgroups, `laps` laps: time / (K * laps) = one hand-off.
//   placement : consecutive workgroup ids (round-robin over the 8 XCDs: every hop crosses XCDs)  |  ids = 0 mod 8 (all on ONE XCD: every hop
//               can be served by that XCD's L2)
//   scope bits of the polling load and the publishing store: sc0 (work-group scope: bypasses the CU's L1, may hit the XCD's L2),
//               sc1 (agent scope -- what the production kernels use), sc0 sc1 (system scope)
// Question: is a hand-off inside one XCD with sc0 accesses >= 1.5x cheaper than the device-scope hand-off?  If so, the 50
// independent probe columns of the log-determinant's block solve can be partitioned over the XCDs.  Every run checks the values it read.
// Usage: flag_handoff
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int SCOPE> __device__ __forceinline__ unsigned long long ld(const unsigned long long* p) {
  unsigned long long v;
  if constexpr (SCOPE == 1) asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  else if constexpr (SCOPE == 2) asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  else asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
template <int SCOPE> __device__ __forceinline__ void st(unsigned long long* p, unsigned long long v) {
  if constexpr (SCOPE == 1) asm volatile("global_store_dwordx2 %0, %1, off sc0" :: "v"(p), "v"(v) : "memory");
  else if constexpr (SCOPE == 2) asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
}

// slots are 128 bytes apart (one cache line each)
template <int SCOPE>
__global__ __launch_bounds__(64) void ring(unsigned long long* slots, int K, int stride, int laps, int* bad, unsigned* xcc_of) {
  if ((int)blockIdx.x % stride != 0) return;
  const int g = (int)blockIdx.x / stride;
  if (threadIdx.x == 0) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc_of[g] = xcc & 0xf;
  }
  if (threadIdx.x != 0) return;
  const int prev = (g + K - 1) % K;
  unsigned long long* mine = slots + (size_t)g * 16;
  const unsigned long long* src = slots + (size_t)prev * 16;
  for (int lap = 1; lap <= laps; ++lap) {
    // workgroup 0 starts lap `lap` once the last workgroup has finished lap - 1 (value lap - 1; the slots start at 0)
    const unsigned long long want = (g == 0) ? (unsigned long long)(lap - 1) : (unsigned long long)lap;
    long spins = 0;
    for (;;) {
      const unsigned long long v = ld<SCOPE>(src);
      if (v == want) break;
      if (v > want || ++spins > (1L << 18)) { atomicAdd(bad, 1); return; }
    }
    st<SCOPE>(mine, (unsigned long long)lap);
  }
}

template <int SCOPE>
static void run(const char* scope_name, unsigned long long* slots, int* bad, unsigned* xcc_of, hipEvent_t e0, hipEvent_t e1) {
  const int laps = 2000;
  for (int mode = 0; mode < 2; ++mode) {
    for (int K = 2; K <= 16; K *= 2) {
      const int stride = mode ? 8 : 1;
      float best = 1e9f; int hb = 0; unsigned x[16];
      for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipMemset(slots, 0, 16 * 16 * 8)); CHECK(hipMemset(bad, 0, 4));
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(ring<SCOPE>, dim3(K * stride), dim3(64), 0, 0, slots, K, stride, laps, bad, xcc_of);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        CHECK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
      }
      CHECK(hipMemcpy(x, xcc_of, sizeof(unsigned) * K, hipMemcpyDeviceToHost));
      int nx = 0; unsigned seen = 0;
      for (int g = 0; g < K; ++g) if (!(seen & (1u << x[g]))) { seen |= 1u << x[g]; ++nx; }
      printf("scope %-7s  %-22s K = %2d: %7.3f us per hand-off   (XCDs touched: %d, failures %d)\n", scope_name,
             mode ? "ids = 0 mod 8 (one XCD)" : "consecutive ids", K, best * 1000.f / ((float)laps * K), nx, hb);
    }
  }
}

int main() {
  unsigned long long* slots; int* bad; unsigned* xcc_of;
  CHECK(hipMalloc(&slots, 16 * 16 * 8)); CHECK(hipMalloc(&bad, 4)); CHECK(hipMalloc(&xcc_of, 64));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  run<1>("sc0", slots, bad, xcc_of, e0, e1);
  run<2>("sc1", slots, bad, xcc_of, e0, e1);
  run<3>("sc0 sc1", slots, bad, xcc_of, e0, e1);
  return 0;
}
