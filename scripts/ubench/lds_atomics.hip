// LDS atomic throughput on gfx950: ds_add_f64 / ds_add_u64 / ds_add_u32 / ds_add_f32 (no return), per CU, for
//   pattern 0: conflict-free (lane l -> word l), 1: uniform random bins in a 256-entry table per 16-lane group table (the histogram
//   pattern: 16 sub-histograms, lane l uses table (l + s) & 15), 2: all lanes random in ONE 256-entry table,
//   3: random bins, BIN-major layout word = bin * 16 + ((l + s) & 15) -- the 16 lanes of a group never share a bank pair (the layout
//   of hist_build_kernel).
// Usage: lds_atomics            (prints lane-updates per cycle per CU for every type x pattern at 3 workgroups per CU)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <typename T, int PATTERN>
__global__ __launch_bounds__(256) void k(const unsigned* __restrict__ rnd, T* out, int iters) {
  __shared__ T tab[16][257];
  const int tid = threadIdx.x;
  for (int t = tid; t < 16 * 257; t += 256) (&tab[0][0])[t] = T(0);
  __syncthreads();
  unsigned r = rnd[blockIdx.x * 256 + tid];
  const T v = T(1);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      r = r * 1664525u + 1013904223u;
      int f, b;
      if (PATTERN == 0) { f = (tid >> 6) & 3; b = tid & 63; }
      else if (PATTERN == 1) { f = (s + tid) & 15; b = (r >> 16) & 255; }
      else if (PATTERN == 2) { f = 0; b = (r >> 16) & 255; }
      else { f = 0; b = ((r >> 16) & 255) * 16 + ((s + tid) & 15); }
      atomicAdd(&(&tab[0][0])[f * 257 + b], v);
    }
  }
  __syncthreads();
  T acc = T(0);
  for (int t = tid; t < 16 * 257; t += 256) acc += (&tab[0][0])[t];
  if (acc == T(12345)) out[blockIdx.x * 256 + tid] = acc;
}

template <typename T, int PATTERN>
double run(const unsigned* d_rnd, void* d_out, int nwg, int iters) {
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<T, PATTERN>), dim3(nwg), dim3(256), 0, 0, d_rnd, (T*)d_out, iters);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<T, PATTERN>), dim3(nwg), dim3(256), 0, 0, d_rnd, (T*)d_out, iters);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms;
}

int main() {
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  const int ncu = p.multiProcessorCount; const double ghz = p.clockRate * 1e-6;
  const int nwg = ncu * 3, iters = 2000;
  std::vector<unsigned> h(nwg * 256); for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned)rand();
  unsigned* d_rnd; void* d_out;
  CHECK(hipMalloc(&d_rnd, h.size() * 4)); CHECK(hipMalloc(&d_out, h.size() * 8));
  CHECK(hipMemcpy(d_rnd, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  const double updates = (double)nwg * 256 * iters * 16;
  printf("%d CUs, %.2f GHz (nominal), 3 workgroups of 256 per CU, %d x 16 atomics per lane\n", ncu, ghz, iters);
#define RUN(T, P, name) { double ms = run<T, P>(d_rnd, d_out, nwg, iters); \
    printf("%-8s pattern %d: %8.3f ms  %6.2f lane-updates/clk/CU\n", name, P, ms, updates / (ms * 1e-3) / (ghz * 1e9) / ncu); }
  RUN(double, 0, "f64") RUN(double, 1, "f64") RUN(double, 2, "f64") RUN(double, 3, "f64")
  RUN(unsigned long long, 0, "u64") RUN(unsigned long long, 1, "u64") RUN(unsigned long long, 2, "u64") RUN(unsigned long long, 3, "u64")
  RUN(unsigned, 0, "u32") RUN(unsigned, 1, "u32") RUN(unsigned, 2, "u32") RUN(unsigned, 3, "u32")
  RUN(float, 0, "f32") RUN(float, 1, "f32") RUN(float, 2, "f32")
  return 0;
}
