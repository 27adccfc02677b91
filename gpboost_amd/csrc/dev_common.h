// gpboost_amd/csrc/dev_common.h -- device-side helpers shared by the gfx950 kernels.
//
// Everything here is written for CDNA4 (gfx950) only: 64-lane wavefronts, DPP
// row_newbcast on the fp64 pipe, v_rsq_f64 / v_ldexp_f64.  No other target.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include <utility>

namespace gpb {

// ---- compile-time loop: f(integral_constant<int, I>) for I in [B, E) ------
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}
// descending: I = E-1 ... B
template <int B, int E, class F>
__device__ __forceinline__ void static_for_down(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, E - 1>{});
    static_for_down<B, E - 1>(f);
  }
}

// ---- fp64 DPP within a row of 16 lanes ------------------------------------
// gfx90a+ allows DPP on the DP ALU only with row_newbcast:N (lane N of each 16-lane row is broadcast to
// the row).  There is no clang builtin for the 64-bit form, hence inline asm.  Measured on MI355X
// (scripts/ubench/fp64_rates.hip): v_fmac_f64_dpp issues at the plain v_fmac_f64 rate (4 cycles per wave).
//
// Hazard: "VALU writes VGPR -> DPP reads that VGPR" needs 2 wait states, and the compiler neither sees the
// DPP inside an asm statement nor pads for it.  row_bcast() therefore opens with s_nop 1.  row_fnma() -- issued
// ~600 times per point, where the nops cost ~12 % of the kernel -- carries NO padding: the statements are
// `volatile` (program order is kept) and the elimination is written so that a DPP source register is always
// produced several instructions earlier; scripts/check_dpp_hazards.py verifies that property on the generated
// ISA of every translation unit at build time and fails the build otherwise.
// All 64 lanes must be active (a disabled source lane would read as 0): the kernels never early-exit a lane.
template <int LANE>
__device__ __forceinline__ double row_bcast(double x) {
  static_assert(LANE >= 0 && LANE < 16, "row_newbcast lane");
  double r;
  asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf"
               : "=v"(r)
               : "v"(x), "n"(LANE));
  return r;
}
// acc -= bcast<LANE>(b) * own      (one v_fmac_f64 with the broadcast folded in)
template <int LANE>
__device__ __forceinline__ void row_fnma(double& acc, double b, double own) {
  static_assert(LANE >= 0 && LANE < 16, "row_newbcast lane");
#ifdef GPB_DPP_PAD_EVERY_FMAC
  asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
               : "+v"(acc)
               : "v"(b), "v"(own), "n"(LANE));
#else
  asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
               : "+v"(acc)
               : "v"(b), "v"(own), "n"(LANE));
#endif
}
// Explicit hazard fence: makes the listed values opaque (they must exist before this point) and supplies the two
// wait states, so that DPP reads of them further down are safe whatever the compiler scheduled just before.
__device__ __forceinline__ void dpp_fence(double& a) { asm volatile("s_nop 1" : "+v"(a)); }
// Portable-in-HIP variant of the same two primitives (two 32-bit DPP movs that the
// compiler schedules and pads itself).  Used by the self-test kernel to validate the
// asm forms on the device, and selectable with -DGPB_DPP_VIA_BUILTIN for debugging.
template <int LANE>
__device__ __forceinline__ double row_bcast_builtin(double x) {
  const long long v = __builtin_bit_cast(long long, x);
  int lo = (int)(v & 0xffffffffll), hi = (int)(v >> 32);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0x150 + LANE, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, 0x150 + LANE, 0xf, 0xf, false);
  const long long r = ((long long)hi << 32) | (unsigned long long)(unsigned int)lo;
  return __builtin_bit_cast(double, r);
}
#ifdef GPB_DPP_VIA_BUILTIN
#define GPB_ROW_BCAST(L, x) ::gpb::row_bcast_builtin<L>(x)
#define GPB_ROW_FNMA(L, acc, b, own) (acc) = __builtin_fma(-::gpb::row_bcast_builtin<L>(b), (own), (acc))
#else
#define GPB_ROW_BCAST(L, x) ::gpb::row_bcast<L>(x)
#define GPB_ROW_FNMA(L, acc, b, own) ::gpb::row_fnma<L>((acc), (b), (own))
#endif

// ---- fast fp64 elementary functions (relative error ~1e-15, far inside the 1e-8 parity budget) ----

// 1/x: v_rcp_f64 + one Newton step.
__device__ __forceinline__ double fast_rcp(double x) {
  const double y0 = __builtin_amdgcn_rcp(x);
  const double e = __builtin_fma(-x, y0, 1.0);
  return __builtin_fma(y0, e, y0);
}

#define GPB_EXP_TAB_SIZE 64   // 2^(j/64), j = 0..63 (filled on the host, pre-multiplied by the variance in the kernels)

// ---- isotropic Matern kernels on the transformed scale --------------------
// reference: include/GPBoost/cov_fcts.h:2100-2118 (CovarianceMaternShape0_5/1_5/2_5)
enum CovType : int { kMatern05 = 0, kMatern15 = 1, kMatern25 = 2 };

// ---- scaled-distance form used by the hot kernels -------------------------------------------
// Coordinates are pre-multiplied by a * 64/ln2, so that exp(-a d) = 2^(-r'/64) with r' the scaled distance:
// no multiply by the range, no Cody-Waite reduction, the polynomial absorbs ln2/64, the table absorbs the variance.
constexpr double kLn2Over64 = 0.010830424696249145;   // ln2 / 64
constexpr double k64OverLn2 = 92.332482616893657;     // 64 / ln2

// Everything a kernel evaluation needs, from the scaled squared distance d2s = (a d 64/ln2)^2:
//   ev = var * exp(-a d)   (tabv already carries var),   rp = a d 64/ln2
struct KernEval { double ev, rp; };
__device__ __forceinline__ KernEval exp_of_scaled(double d2s, const double* __restrict__ tabv) {
  const double rs = __builtin_amdgcn_rsq(d2s);
  const double g = d2s * rs, h = 0.5 * rs;
  const double e = __builtin_fma(-h, g, 0.5);
  const double rp = __builtin_fma(g, e, g);                 // sqrt(d2s), one Newton step on v_rsq_f64
  const double kf = __builtin_rint(-rp);
  const double rr = -rp - kf;                               // exact; |rr| <= 1/2, in units of ln2/64
  const int k = (int)kf;                                    // saturates for the dummy rows
  // exp(rr ln2/64) = sum_j (ln2/64)^j rr^j / j!, j <= 5  (remainder < 2e-17)
  double p = __builtin_fma(rr, 1.2417843701716925e-12, 5.732851688640402e-10);
  p = __builtin_fma(p, rr, 2.1173137155464776e-07);
  p = __builtin_fma(p, rr, 5.86490495505617e-05);
  p = __builtin_fma(p, rr, kLn2Over64);
  p = __builtin_fma(p, rr, 1.0);
  KernEval o;
  o.ev = __builtin_ldexp(tabv[k & 63] * p, k >> 6);
  o.rp = rp;
  return o;
}
// include/GPBoost/cov_fcts.h:2100-2118 (CovarianceMaternShape0_5/1_5/2_5), transformed scale
template <int COV>
__device__ __forceinline__ double matern_cov_s(double d2s, const double* __restrict__ tabv) {
  const KernEval k = exp_of_scaled(d2s, tabv);
  if constexpr (COV == kMatern05) return k.ev;
  else if constexpr (COV == kMatern15) return k.ev * __builtin_fma(k.rp, kLn2Over64, 1.0);
  else { const double r = k.rp * kLn2Over64; return k.ev * __builtin_fma(r, __builtin_fma(r, 1.0 / 3.0, 1.0), 1.0); }
}
// d/d log(a) of the kernel (transf_scale == true): include/GPBoost/cov_fcts.h:2182-2193 (cm), :2535-2554
template <int COV>
__device__ __forceinline__ double matern_dlog_range_s(double d2s, const double* __restrict__ tabv) {
  const KernEval k = exp_of_scaled(d2s, tabv);
  const double r = k.rp * kLn2Over64;
  if constexpr (COV == kMatern05) return -r * k.ev;                                 // cm d sigma, cm = -a
  else if constexpr (COV == kMatern15) return -(r * r) * k.ev;                      // cm d^2 e^{-ad}, cm = -var a^2
  else return -(1.0 / 3.0) * (r * r) * __builtin_fma(1.0, r, 1.0) * k.ev;           // cm/3 d^2 (1+ad) e^{-ad}
}

}  // namespace gpb
