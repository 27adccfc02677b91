// Micro-benchmark: issue cost (cycles per wave-instruction per SIMD) of the fp64 instructions the Vecchia kernel
// leans on: v_fmac_f64, v_fmac_f64_dpp row_newbcast, v_mov_b64_dpp, v_mfma_f64_4x4x4_4b_f64, v_mfma_f64_16x16x4_f64.
// hipcc --offload-arch=gfx950 -O3 fp64_rates.hip -o fp64_rates && ./fp64_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define N_ITER 2000
#define UNROLL 16
typedef double double4v __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ __launch_bounds__(256) void k(double* out, long long* cyc) {
  double a[UNROLL];
  for (int i = 0; i < UNROLL; ++i) a[i] = threadIdx.x * 1e-3 + i;
  double b = 1.0000001 + threadIdx.x * 1e-9, c = 0.999999;
  double4v acc4[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < N_ITER; ++it) {
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) {
      if (KIND == 0) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      if (KIND == 1) asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b), "v"(c));
      if (KIND == 2) asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(a[i]) : "v"(b));
      if (KIND == 3) asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
      if (KIND == 4) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc4[i & 3]) : "v"(b), "v"(c));
      if (KIND == 5) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, -%1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b), "v"(c));
      if (KIND == 6) asm volatile("v_rsq_f64 %0, %1" : "=v"(a[i]) : "v"(b));
      if (KIND == 7) asm volatile("v_ldexp_f64 %0, %1, 3" : "=v"(a[i]) : "v"(b));
      if (KIND == 8) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(((int*)&a[i])[0]) : "v"(((int*)&b)[0]), "v"(((int*)&c)[0]) : );
      // round 6 (the "two points per DPP row" question): a DPP fmac that writes only HALF of every 16-lane row (bank_mask 0x3 = lanes 0..7) -- does a
      // half-row instruction cost half an issue slot?  (If not, two points sharing a row need two instructions per (column, pivot): one per point's broadcast.)
      if (KIND == 9) asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:3 row_mask:0xf bank_mask:0x3" : "+v"(a[i]) : "v"(b), "v"(c));
      if (KIND == 10) asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:3 row_mask:0x3 bank_mask:0xf" : "+v"(a[i]) : "v"(b), "v"(c));
    }
  }
  long long t1 = __builtin_readcyclecounter();
  double s = 0;
  for (int i = 0; i < UNROLL; ++i) s += a[i];
  for (int i = 0; i < 4; ++i) s += acc4[i][0] + acc4[i][1] + acc4[i][2] + acc4[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND>
void run(const char* name, int waves_per_simd) {
  const int blocks = 256 * waves_per_simd;   // 256 CUs x (4 waves per block = 1 per SIMD) x waves_per_simd
  double* out; long long* cyc;
  hipMalloc(&out, sizeof(double) * blocks * 256); hipMalloc(&cyc, sizeof(long long) * blocks);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, cyc);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, cyc);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(blocks); hipMemcpy(h.data(), cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
  double avg = 0; for (auto v : h) avg += v; avg /= blocks;
  const double insts = (double)N_ITER * UNROLL;
  // readcyclecounter = s_memtime ticks (100 MHz constant clock on some parts) -> report wall-derived cycles at 2.4 GHz too
  printf("%-34s waves/SIMD=%d  wall %.3f ms  -> %.2f ns per wave-instr per SIMD (x2.4GHz = %.2f cyc)  [counter ticks/instr %.3f]\n",
         name, waves_per_simd, ms, ms * 1e6 / (insts * waves_per_simd), ms * 1e6 / (insts * waves_per_simd) * 2.4, avg / insts);
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int w : {1, 2, 4}) {
    run<0>("v_fmac_f64", w);
    run<1>("v_fmac_f64_dpp row_newbcast", w);
    run<5>("s_nop 1 + v_fmac_f64_dpp", w);
    run<2>("v_mov_b64_dpp row_newbcast", w);
    run<3>("v_mfma_f64_4x4x4_4b_f64", w);
    run<4>("v_mfma_f64_16x16x4_f64", w);
    run<6>("v_rsq_f64", w);
    run<7>("v_ldexp_f64", w);
    run<8>("v_cndmask_b32", w);
    run<9>("v_fmac_f64_dpp bank_mask:0x3 (half rows)", w);
    run<10>("v_fmac_f64_dpp row_mask:0x3 (two of four rows)", w);
  }
  return 0;
}
