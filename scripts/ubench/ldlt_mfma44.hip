// Micro-benchmark (round 4; asked for in VERDICT r01-r03): the trailing updates of the point kernel's 32 x 32 LDL^T elimination (MT = 30: 30 neighbours +
// the point + the response row) as 4 x 4 blocks on v_mfma_f64_4x4x4_4b_f64 (four blocks per instruction = the four points of a wavefront) against
// the production form (row per lane, v_fmac_f64_dpp row_newbcast: 585 instructions per wavefront of four points).
//
// WHAT IS MEASURED: the issue / latency cost of the two INSTRUCTION STREAMS with their real dependency chains, cycles per wavefront of four points
// (s_memtime around many repetitions, 1 / 2 / 4 wavefronts per SIMD).  Stream A is the production elimination (same macros, same order: pivot
// broadcast, v_rcp_f64 + Newton step, column scale, DPP fmacs).  Stream B is the blocked right-looking LDL^T with 4 x 4 blocks, one matrix element
// per lane and block (36 lower blocks = 36 fp64 registers):
//   per pivot block K:  factorise the diagonal block inside its 16 lanes (3 pivots: row_newbcast + v_rcp_f64 + Newton + scale + rank-1 update with two
//                       broadcasts; then the inverse of the unit-lower 4 x 4 factor, 6 more broadcast-fmas)
//                       panel: P(I) = M(I,K) (L_KK^-T D_K^-1), one MFMA per block row I > K                                   (28 in all)
//                       operand layout of the right factor: (P(J) D)^T needs the block transposed across its 16 lanes: two ds_bpermute_b32 + one
//                       multiply per block row J > K                                                                           (28 in all)
//                       trailing update: M(I,J) -= P(I) (P(J) D)^T, one MFMA per block (I >= J > K)                           (84 in all)
// Stream B is an instruction-stream MODEL: its dependency structure and instruction counts are those of the blocked factorisation, the numerical
// result is not checked (the lane layout of the MFMA operands is not asserted here) -- it answers "how many cycles would the elimination cost on the
// MFMA pipe", which is what the comparison needs.  Not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../gpboost_amd/csrc ldlt_mfma44.hip -o ldlt_mfma44 && ./ldlt_mfma44
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "dev_common.h"

using namespace gpb;

template <int I> struct IC { static constexpr int value = I; };
template <int B, int E, class F> __device__ __forceinline__ void sfor(F&& f) {
  if constexpr (B < E) { f(IC<B>{}); sfor<B + 1, E>(f); }
}

__device__ __forceinline__ double rcp_newton(double x) {
  const double y = __builtin_amdgcn_rcp(x);
  const double e = __builtin_fma(-x, y, 1.0);
  return __builtin_fma(y, e, y);
}

// ---- stream A: production elimination, MT = 30 ------------------------------------------------------------------------------------
__device__ __forceinline__ void eliminate_dpp(double (&M)[2][32]) {
  constexpr int MT = 30;
  sfor<0, MT>([&](auto k_) {
    constexpr int k = decltype(k_)::value, sk = k / 16, lk = k % 16;
    if constexpr (k == 0) { dpp_fence(M[0][0]); dpp_fence(M[1][0]); }
    const double piv = row_bcast<lk>(M[sk][k]);
    const double inv = rcp_newton(piv);
    double T[2];
    sfor<sk, 2>([&](auto s_) { T[decltype(s_)::value] = M[decltype(s_)::value][k] * inv; });
    sfor<k + 1, MT + 1>([&](auto c_) {
      constexpr int c = decltype(c_)::value, sc = c / 16, lc = c % 16;
      sfor<sc, 2>([&](auto s_) { constexpr int s = decltype(s_)::value; row_fnma<lc>(M[s][c], M[sc][k], T[s]); });
    });
    sfor<sk, 2>([&](auto s_) { M[decltype(s_)::value][k] = T[decltype(s_)::value]; });
  });
}

// ---- stream B: blocked 4 x 4 MFMA elimination (model) -----------------------------------------------------------------------------------
__device__ __forceinline__ void mfma44(double& acc, double a, double b) {
  asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ double lane_transpose(double x, int src_lane_byte) {      // element (i, j) <- element (j, i) of the same block: two ds_bpermute_b32
  const long long v = __builtin_bit_cast(long long, x);
  const int lo = __builtin_amdgcn_ds_bpermute(src_lane_byte, (int)(v & 0xffffffffll));
  const int hi = __builtin_amdgcn_ds_bpermute(src_lane_byte, (int)(v >> 32));
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned long long)(unsigned)lo);
}
constexpr int blk(int I, int J) { return I * (I + 1) / 2 + J; }      // lower blocks, I >= J
__device__ __forceinline__ void eliminate_mfma(double (&B)[36], int tlane_byte) {
  sfor<0, 8>([&](auto K_) {
    constexpr int K = decltype(K_)::value;
    double d = B[blk(K, K)];
    // diagonal block inside its 16 lanes: 3 pivots (the 4th needs no update)
    double linv = 1.0;
    sfor<0, 3>([&](auto p_) {
      constexpr int p = decltype(p_)::value;
      dpp_fence(d);
      const double piv = row_bcast<5 * p>(d);                         // element (p, p) lives in lane 4 p + p
      const double inv = rcp_newton(piv);
      const double col = row_bcast<4 * ((p + 1) % 4) + p>(d) * inv;   // a column entry (model: one broadcast)
      double rowv = d;
      dpp_fence(rowv);
      const double rw = row_bcast<4 * p + ((p + 1) % 4)>(rowv);       // a row entry (second broadcast)
      d = __builtin_fma(-col, rw, d);                                 // rank-1 update of the block
      linv = __builtin_fma(-col, linv, linv * inv);                   // running inverse of the unit-lower factor / scaling by D^-1 (2 ops per pivot)
    });
    sfor<0, 3>([&](auto q_) {                                         // finishing the 4 x 4 inverse: 6 broadcast-fmas, modelled as 3 x 2
      constexpr int q = decltype(q_)::value;
      double t = linv;
      dpp_fence(t);
      const double b0 = row_bcast<q + 1>(t);
      linv = __builtin_fma(-b0, d, linv);
      dpp_fence(linv);
      const double b1 = row_bcast<4 * q + 2>(linv);
      linv = __builtin_fma(-b1, d, linv);
    });
    B[blk(K, K)] = d;
    // panel: P(I) = M(I, K) * (L_KK^-T D^-1)
    double P[8];
    sfor<K + 1, 8>([&](auto I_) {
      constexpr int I = decltype(I_)::value;
      double acc = 0.0;
      mfma44(acc, B[blk(I, K)], linv);
      P[I] = acc;
    });
    // right factor in B-operand layout: (P(J) D)^T
    double Pt[8];
    sfor<K + 1, 8>([&](auto J_) {
      constexpr int J = decltype(J_)::value;
      Pt[J] = -lane_transpose(P[J], tlane_byte) * d;
    });
    // trailing update
    sfor<K + 1, 8>([&](auto I_) {
      constexpr int I = decltype(I_)::value;
      sfor<K + 1, I + 1>([&](auto J_) {
        constexpr int J = decltype(J_)::value;
        mfma44(B[blk(I, J)], P[I], Pt[J]);
      });
      B[blk(I, K)] = P[I];
    });
  });
}

template <int KIND>
__global__ __launch_bounds__(256) void bench_kernel(const double* __restrict__ in, double* __restrict__ out, long long* __restrict__ cyc, int reps) {
  const int tid = threadIdx.x, lane = tid & 63;
  double acc_out = 0.0;
  long long t0 = 0, t1 = 0;
  // the matrices are made up in registers from two loaded values (no memory traffic inside the timed loop; `seed` is opaque to the compiler)
  double seed = in[lane], eps = in[64 + lane];
  if constexpr (KIND == 0) {
    double M[2][32];
    for (int r = 0; r < reps + 1; ++r) {
      if (r == 1) t0 = __builtin_readcyclecounter();
      asm volatile("" : "+v"(seed), "+v"(eps));
      sfor<0, 16>([&](auto c_) { constexpr int c = decltype(c_)::value; M[0][c] = __builtin_fma(eps, (double)(c + 1), seed) + (c == 0 ? 64.0 : 0.0); });
      sfor<0, 31>([&](auto c_) { constexpr int c = decltype(c_)::value; M[1][c] = __builtin_fma(eps, (double)(c + 33), seed) + (c == 16 ? 64.0 : 0.0); });
      eliminate_dpp(M);
      double sacc = 0.0;
      sfor<0, 16>([&](auto c_) { sacc += M[0][decltype(c_)::value]; });
      sfor<0, 31>([&](auto c_) { sacc += M[1][decltype(c_)::value]; });
      acc_out += sacc;
      seed = __builtin_fma(sacc, 1e-300, seed);          // the next repetition depends on this one
    }
    t1 = __builtin_readcyclecounter();
  } else {
    double B[36];
    const int i = (lane & 15) >> 2, j = lane & 3;
    const int tl = ((lane & ~15) + 4 * j + i) * 4;
    for (int r = 0; r < reps + 1; ++r) {
      if (r == 1) t0 = __builtin_readcyclecounter();
      asm volatile("" : "+v"(seed), "+v"(eps));
      sfor<0, 36>([&](auto b_) { constexpr int b = decltype(b_)::value; B[b] = __builtin_fma(eps, (double)(b + 1), seed) + 8.0; });
      eliminate_mfma(B, tl);
      double sacc = 0.0;
      sfor<0, 36>([&](auto b_) { sacc += B[decltype(b_)::value]; });
      acc_out += sacc;
      seed = __builtin_fma(sacc, 1e-300, seed);
    }
    t1 = __builtin_readcyclecounter();
  }
  out[blockIdx.x * 256 + tid] = acc_out;
  if (lane == 0) cyc[blockIdx.x * 4 + (tid >> 6)] = t1 - t0;
}

int main() {
  const int reps = 200;
  std::vector<double> h(64 * 64);
  for (size_t q = 0; q < h.size(); ++q) h[q] = 0.01 * ((q * 2654435761u) % 97) / 97.0;
  double* d_in; double* d_out; long long* d_cyc;
  const int max_blocks = 256 * 8;
  hipMalloc(&d_in, h.size() * 8); hipMalloc(&d_out, (size_t)max_blocks * 256 * 8); hipMalloc(&d_cyc, (size_t)max_blocks * 4 * 8);
  hipMemcpy(d_in, h.data(), h.size() * 8, hipMemcpyHostToDevice);
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  printf("device %s, %d CUs; %d repetitions per wavefront; cycles = s_memtime ticks (%.0f MHz counter) per wavefront of FOUR points\n", prop.name, cus, reps, 100.0);
  for (int wps = 1; wps <= 4; wps *= 2) {          // wavefronts per SIMD: 256-thread workgroups = 1 wavefront per SIMD each
    for (int kind = 0; kind < 2; ++kind) {
      const int blocks = cus * wps;
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      if (kind == 0) hipLaunchKernelGGL(bench_kernel<0>, dim3(blocks), dim3(256), 0, 0, d_in, d_out, d_cyc, 4);
      else hipLaunchKernelGGL(bench_kernel<1>, dim3(blocks), dim3(256), 0, 0, d_in, d_out, d_cyc, 4);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      if (kind == 0) hipLaunchKernelGGL(bench_kernel<0>, dim3(blocks), dim3(256), 0, 0, d_in, d_out, d_cyc, reps);
      else hipLaunchKernelGGL(bench_kernel<1>, dim3(blocks), dim3(256), 0, 0, d_in, d_out, d_cyc, reps);
      hipEventRecord(e1); hipDeviceSynchronize();
      float ms = 0; hipEventElapsedTime(&ms, e0, e1);
      // per SIMD: wps wavefronts x reps eliminations of 4 points each in `ms`
      const double us_per_elim_per_simd = ms * 1e3 / reps;        // time for wps concurrent eliminations on one SIMD
      printf("%-28s %d wavefront(s) per SIMD: %8.3f ms total, %7.3f us per round of %d wavefront-eliminations per SIMD = %6.0f SIMD-cycles at 2.4 GHz per wavefront of 4 points\n",
             kind == 0 ? "A: DPP fmac (production)" : "B: 4x4 blocks on MFMA (model)", wps, ms, us_per_elim_per_simd, wps, us_per_elim_per_simd * 2400.0 / wps);
    }
  }
  return 0;
}
