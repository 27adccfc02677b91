// Micro-benchmark (round 2): can the fp64 MFMA pipe and the fp64 vector pipe of a gfx950 SIMD work at the same time?
// The Vecchia point kernel is VALU-issue bound; if v_mfma_f64_4x4x4_4b co-issues with v_fma_f64 (same wave or different
// waves of a SIMD), moving the elimination to MFMA would free VALU cycles -- if both share one pipe it buys nothing.
// Also: issue cost of the integer / select / convert / move instructions of the assembly phase, and ds_read co-issue.
// hipcc --offload-arch=gfx950 -O3 coissue.hip -o coissue && ./coissue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define N_ITER 2000

// KIND 0: 16 v_fmac_f64                     (VALU only)
// KIND 1: 16 v_mfma_f64_4x4x4_4b            (MFMA only)
// KIND 2: 16 v_fmac_f64 + 16 mfma, interleaved in ONE wave, independent accumulators
// KIND 3: even waves of the workgroup run KIND 0, odd waves KIND 1 (waves/SIMD >= 2 => both kinds on every SIMD)
// KIND 4: 16 v_fmac_f64 + 4 v_rsq_f64 interleaved (transcendental co-issue?)
// KIND 5: 16 v_fmac_f64 + 8 ds_read_b64 interleaved
// KIND 6..: single-instruction costs
template <int KIND>
__global__ __launch_bounds__(256) void k(double* out, int wave_split) {
  __shared__ double lds[1024];
  lds[threadIdx.x] = threadIdx.x; lds[threadIdx.x + 256] = 1.0; lds[threadIdx.x + 512] = 2.0; lds[threadIdx.x + 768] = 3.0;
  __syncthreads();
  double a[16], m[16];
  for (int i = 0; i < 16; ++i) { a[i] = threadIdx.x * 1e-3 + i; m[i] = 0.0; }
  double b = 1.0000001 + threadIdx.x * 1e-9, c = 0.999999;
  int ia = threadIdx.x, ib = 7;
  const int wave = threadIdx.x >> 6;
  const bool do_valu = (KIND == 3) ? ((wave & 1) == 0) : true;
  for (int it = 0; it < N_ITER; ++it) {
    if (KIND == 0 || (KIND == 3 && do_valu)) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
    }
    if (KIND == 1 || (KIND == 3 && !do_valu)) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(m[i]) : "v"(b), "v"(c));
    }
    if (KIND == 2) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(m[i]) : "v"(b), "v"(c));
        asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      }
    }
    if (KIND == 7) {   // 1 mfma : 4 fmac  (the ratio at which both pipes would be equally busy if they were separate)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(m[i]) : "v"(b), "v"(c));
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a[4 * i + j]) : "v"(b), "v"(c));
      }
    }
    if (KIND == 4) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        if ((i & 3) == 0) asm volatile("v_rsq_f64 %0, %1" : "=v"(m[i]) : "v"(b));
      }
    }
    if (KIND == 5) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        if ((i & 1) == 0) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(m[i]) : "v"(ia * 8), "n"(0));
      }
      asm volatile("s_waitcnt lgkmcnt(0)");
    }
    if (KIND == 8) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_and_b32 %0, %1, %2" : "=v"(ib) : "v"(ia), "v"(i + it));
    }
    if (KIND == 9) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_mov_b64 %0, %1" : "=v"(m[i]) : "v"(a[i]));
    }
    if (KIND == 10) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_rndne_f64 %0, %1" : "=v"(m[i]) : "v"(a[i]));
    }
    if (KIND == 11) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(ib) : "v"(a[i]));
    }
    if (KIND == 12) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_add_f64 %0, %1, %2" : "=v"(m[i]) : "v"(a[i]), "v"(b));
    }
    if (KIND == 13) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_lshl_add_u32 %0, %1, 3, %2" : "=v"(ib) : "v"(ia), "v"(i));
    }
    if (KIND == 14) {   // packed fp32 FMA: does it run at 2 results per lane-cycle?
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
    }
    if (KIND == 15) {   // v_cndmask_b32 with an SGPR-pair mask
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(ib) : "v"(ia), "v"(i), "s"(0x0f0f0f0f0f0f0f0full));
    }
    if (KIND == 16) {   // v_fma_f64 with a literal-free SGPR operand
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "s"(1.25));
    }
  }
  double s = ib;
  for (int i = 0; i < 16; ++i) s += a[i] + m[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND>
void run(const char* name, int waves_per_simd, double insts_per_iter) {
  const int blocks = 256 * waves_per_simd;
  double* out;
  hipMalloc(&out, sizeof(double) * blocks * 256);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 0);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 0);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double ns_iter = ms * 1e6 / (N_ITER * (double)waves_per_simd);
  printf("%-52s waves/SIMD=%d  wall %.3f ms  %.1f ns per wave-iteration per SIMD = %.1f cyc @2.4GHz  (%.2f cyc per instr)\n", name, waves_per_simd,
         ms, ns_iter, ns_iter * 2.4, ns_iter * 2.4 / insts_per_iter);
  hipFree(out);
}

int main() {
  for (int w : {2, 4}) {
    run<0>("16 v_fmac_f64", w, 16);
    run<1>("16 v_mfma_f64_4x4x4_4b", w, 16);
    run<2>("16 fmac + 16 mfma interleaved, one wave", w, 32);
    run<7>("16 fmac + 4 mfma interleaved, one wave", w, 20);
    run<3>("even waves 16 fmac / odd waves 16 mfma", w, 16);
    run<4>("16 fmac + 4 v_rsq_f64 interleaved", w, 20);
    run<5>("16 fmac + 8 ds_read_b64 interleaved", w, 16);
    run<8>("16 v_and_b32", w, 16);
    run<9>("16 v_mov_b64", w, 16);
    run<10>("16 v_rndne_f64", w, 16);
    run<11>("16 v_cvt_i32_f64", w, 16);
    run<12>("16 v_add_f64", w, 16);
    run<13>("16 v_lshl_add_u32", w, 16);
    run<14>("16 v_pk_fma_f32", w, 16);
    run<15>("16 v_cndmask_b32 (sgpr mask)", w, 16);
    run<16>("16 v_fma_f64 (sgpr operand)", w, 16);
  }
  return 0;
}
