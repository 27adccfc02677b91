// gpboost_amd/csrc/vecchia_kernels.hip
//
// Fused Vecchia factor kernels for gfx950 (CDNA4).  One launch does, per point i of the
// Vecchia ordering, everything the reference's per-point loop does
// (src/GPBoost/Vecchia_utils.cpp:1461-1683, CalcCovFactorGradientVecchia) *and* the
// reductions that follow it (include/GPBoost/re_model_template.h:9960-9968 u = B y,
// y^T Psi^-1 y = sum u_i^2 / D_i; :2946-2948 log|Psi| = sum log D_i; :1988-2011 the
// covariance-parameter gradient), so B, D^-1, dB, dD never exist in memory unless a
// caller asks for A and D (MODE_FACTOR).
//
// Mapping (this is the MI355X design, not the reference's):
//   * 16 lanes (one DPP row) per point, 4 points per 64-lane wavefront, 16 per workgroup.
//   * The point's augmented system is held row-per-lane in registers:
//       rows 0..MT-1   the (padded) neighbours            C_nn + nugget
//       row  MT        the point itself                   [c^T, sigma1^2 + nugget]
//       row  MT+1      the responses                      [y_nn^T, y_i]
//     row r lives in lane r%16, register slot r/16.  Right-looking elimination of
//     columns 0..MT-1 leaves D_i in entry (MT, MT) and u_i = (B y)_i in entry (MT+1, MT):
//     no triangular solves, no A_i, for the likelihood.
//   * The rank-1 updates use v_fmac_f64 with the DPP row_newbcast modifier: the
//     multiplier L[c][k] is broadcast from the lane that owns row c *inside* the FMA, so
//     the elimination costs one fp64 VALU op per (slot, c, k) and no LDS traffic.
//   * Short rows (i < m) and m < MT are padded with decoupled dummy neighbours placed
//     1e30 apart (their covariances underflow to exactly 0), so there is no divergence.
//   * Neighbour records {x0,x1,x2,y} (32 B) are gathered once per point into LDS; column
//     operands of the kernel evaluations are LDS broadcast reads.
//   * Block partial sums are written per workgroup and reduced by a second, single-block
//     kernel in a fixed order: results are bit-reproducible run to run.
//
// Roofline notes: algorithmic HBM bytes per point are
// 4m + 8d(m+1) + 8(m+1); the kernel is fp64-VALU bound (m(m+1)/2 exp+sqrt and ~m^3/3
// FMAs per point), see DESIGN.md.
#include "dev_common.h"
#include "vecchia_kernels.h"

namespace gpb {

namespace {

// Dummy (padding) neighbours are placed kDummyCoord * (row + 1) away in *scaled* coordinates: the
// scaled distance r' >= 1e30 makes every kernel value underflow to exactly 0 (v_cvt_i32_f64 saturates,
// v_ldexp_f64 flushes), while r'^3 stays finite for the Matern-2.5 derivative.
constexpr double kDummyCoord = 1e30;

// Row r of the augmented system lives in register slot r/16 and, inside its 16-lane DPP row, in lane
// r%16 for even slots and 15 - r%16 for odd slots.  Reversing the odd slots makes the triangular part of
// slot s (valid lanes: the upper ones) and of slot s+1 (valid lanes: the lower ones) complementary, so one
// kernel evaluation per lane fills both (see assemble()).
__host__ __device__ constexpr int lane_of_row(int r) { return ((r / 16) & 1) ? 15 - (r % 16) : (r % 16); }

template <int MT>
struct Layout {
  static_assert(MT >= 1 && MT <= 62, "1 <= MT <= 62");
  static_assert(MT % 16 != 15, "row MT and row MT+1 must share a register slot");
  static constexpr int R = MT + 2;                 // rows 0..MT-1 neighbours, MT the point, MT+1 the responses
  static constexpr int NS = (R + 15) / 16;         // register slots per lane
  static constexpr int PS = MT / 16, PL = lane_of_row(MT);            // slot / lane of the point's own row
  static constexpr int YS = (MT + 1) / 16, YL = lane_of_row(MT + 1);  // slot / lane of the response row
  static constexpr int NCOL = MT + 1;              // columns 0..MT
  static constexpr int PTS_STRIDE = NS * 16 + 1;   // LDS records per point group with 32-byte records (+1: bank spread)
  static constexpr int PTS_STRIDE24 = NS * 16 + 2; // ... with 24-byte records (d <= 2): keeps every point's block 16-byte aligned
  __host__ __device__ static constexpr int cmax(int s) { return (16 * s + 15 < MT) ? 16 * s + 15 : MT; }
};

// scaled, centred coordinates + response (w); 24 bytes when the third coordinate does not exist (d <= 2)
template <bool D3> struct RecT;
template <> struct RecT<true> { double x, y, z, w; __device__ __forceinline__ double zz() const { return z; } };
template <> struct RecT<false> { double x, y, w; __device__ __forceinline__ double zz() const { return 0.0; } };

// scaled squared distance + 1e-300 (keeps rsq finite for duplicates)
template <bool D3>
__device__ __forceinline__ double sq_dist_s(double px, double py, double pz, double qx, double qy, double qz) {
  const double dx = px - qx, dy = py - qy;
  double d2 = __builtin_fma(dx, dx, 1e-300);
  d2 = __builtin_fma(dy, dy, d2);
  if constexpr (D3) {
    const double dz = pz - qz;
    d2 = __builtin_fma(dz, dz, d2);
  }
  return d2;
}

// Visits every strictly-lower entry (row r <= MT, column c < r) of the augmented system exactly once per owning
// lane, as a list of compile-time "steps"; in a step every lane evaluates ONE entry:
//   rect(s, c)                : all 16 lanes of slot s, column c < 16 s
//   pair(sA, cA, sB, cB, J)   : lanes l <= J take entry (slot sA, column cA), lanes l > J take (slot sB, column cB)
//   solo(s, c)                : slot s, column c inside the slot's own triangle (only lanes whose row > c are meaningful)
// Every callback also receives the step's index E (0 .. num_lower_steps<MT>() - 1) as its last argument.
template <int MT>
__host__ __device__ constexpr int num_lower_steps() {
  using L = Layout<MT>;
  int n = 8 * L::NS * (L::NS - 1) + 15 * (L::NS / 2);
  if (L::NS & 1) { const int s = L::NS - 1; const int chi = (16 * s + 14 < MT - 1) ? 16 * s + 14 : MT - 1; n += chi - 16 * s + 1; }
  return n;
}
template <int MT, class FR, class FP, class FS>
__device__ __forceinline__ void for_each_lower_step(FR&& rect, FP&& pair, FS&& solo) {
  using L = Layout<MT>;
  constexpr int RECT_TOTAL = 8 * L::NS * (L::NS - 1);
  static_for<1, L::NS>([&](auto s_) {
    constexpr int s = decltype(s_)::value;
    static_for<0, 16 * s>([&](auto c_) { rect(s_, c_, std::integral_constant<int, 8 * s * (s - 1) + decltype(c_)::value>{}); });
  });
  static_for<0, L::NS>([&](auto s_) {
    constexpr int s = decltype(s_)::value;
    if constexpr ((s & 1) == 1) {
      static_for<0, 15>([&](auto j_) {
        constexpr int j = decltype(j_)::value;
        constexpr int cA = 16 * s + j, cB = 16 * (s - 1) + 14 - j;
        using E = std::integral_constant<int, RECT_TOTAL + 15 * (s / 2) + j>;
        if constexpr (cA <= MT - 1) pair(s_, std::integral_constant<int, cA>{}, std::integral_constant<int, s - 1>{},
                                         std::integral_constant<int, cB>{}, std::integral_constant<int, 14 - j>{}, E{});
        else solo(std::integral_constant<int, s - 1>{}, std::integral_constant<int, cB>{}, E{});
      });
    } else if constexpr (s == L::NS - 1) {   // unpaired last (even) slot
      constexpr int chi = (16 * s + 14 < MT - 1) ? 16 * s + 14 : MT - 1;
      static_for<16 * s, chi + 1>([&](auto c_) { solo(s_, c_, std::integral_constant<int, RECT_TOTAL + 15 * (L::NS / 2) + decltype(c_)::value - 16 * s>{}); });
    }
  });
}

}  // namespace

// MODE_NLL    : partial sums {sum log D, sum u^2/D, #(D<=0)} only
// MODE_FACTOR : additionally A[n][m], D[n], u[n] to HBM
// MODE_GRAD   : partial sums for the nll terms and the two parameter gradients
template <int MT, int COV, bool D3, int MODE>
__global__ __launch_bounds__(256) void vecchia_point_kernel(VecchiaKernelArgs args) {
  using L = Layout<MT>;
  constexpr int NS = L::NS;
  constexpr bool kNeedSolve = (MODE != MODE_NLL);
  constexpr int NP = (MODE == MODE_GRAD) ? GPB_NUM_PARTIALS : 3;

  using Rec = RecT<D3>;
  // MODE_GRAD with MT <= 30: d/dlog(a) of every kernel entry is kept in LDS from the assembly pass ([step][thread]: conflict-free,
  // 31 x 256 x 8 B = 62 KB for MT = 30, two workgroups per CU) so that the contraction pass evaluates no exp; larger MT re-evaluate.
  constexpr bool kStoreDK = (MODE == MODE_GRAD) && (MT <= 30);
  constexpr int NSTEP = num_lower_steps<MT>();
  // (Tried in round 2: the last 13 steps' values in registers + __launch_bounds__(256, 3), so that three workgroups -- 12 wavefronts instead of
  // 8 -- share a CU: the compiler needs 241 VGPRs for that form, capped at 168 it spills 12 doubles to scratch, and the kernel got 13 % SLOWER.)
  // (Also tried: amdgpu_waves_per_eu(5, 5) on the likelihood kernel -- 102 -> 96 VGPRs, five wavefronts per SIMD instead of four, 84 bytes of
  // scratch per lane: 0.89 -> 1.10 ms.  Any spill to scratch costs more than the extra wavefront hides.)
  constexpr bool kLastDkInReg = kStoreDK && D3;      // d = 3: 32-byte records; the last step's value stays in a register so that two workgroups fit a CU's 160 KB
  constexpr int NSTORE = kStoreDK ? (kLastDkInReg ? NSTEP - 1 : NSTEP) : 1;
  constexpr int PSTRIDE = D3 ? L::PTS_STRIDE : L::PTS_STRIDE24;
  __shared__ double s_tab[GPB_EXP_TAB_SIZE];
  __shared__ __attribute__((aligned(16))) Rec s_pts[16][PSTRIDE];
  __shared__ double s_red[GPB_NUM_PARTIALS][16];
  __shared__ double s_dk[NSTORE][kStoreDK ? 256 : 1];
  // (A~_r, b~_r) pairs of the contraction pass: with kStoreDK they live in the point's record block, which is dead by then (the same 16
  // lanes of one wavefront write and read it); the re-evaluating variant still needs the records and gets its own array
  static_assert((sizeof(Rec) * PSTRIDE) % 16 == 0 && sizeof(Rec) * PSTRIDE >= 16 * (NS * 16), "record block: 16-byte aligned, room for the (A, b) pairs");
  __shared__ double2 s_ab[(MODE == MODE_GRAD && !kStoreDK) ? 16 : 1][(MODE == MODE_GRAD && !kStoreDK) ? NS * 16 : 1];
  double dk_last = 0.0;

  const int tid = threadIdx.x;
  const int g = tid >> 4;   // point within the workgroup
  const int l = tid & 15;   // lane within the point's DPP row
  static_assert(GPB_EXP_TAB_SIZE == 256, "one table entry per thread");
  s_tab[tid] = args.exp_tab[tid] * args.var;            // var * 2^(j/256)

  const long long i_raw = (long long)args.i_begin + (long long)blockIdx.x * 16 + g;
  const bool active = i_raw < (long long)args.i_end;
  const int i = active ? (int)i_raw : args.i_end - 1;   // inactive groups redo the last point, contribute 0
  const int m = args.m;
  const double sc = args.a * kCoordScale;               // half-scaled coordinates: squared distances are (rho/2)^2, exp(-a d) = 2^(-rho/256)

  // ---- gather the rows' records: centred on the point (differences of nearby points stay accurate for
  //      coordinates with a large offset), scaled, staged in LDS for the column operands ------------------
  const double4 ctr = args.pts[i];
  Rec own[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int r = 16 * s + ((s & 1) ? 15 - l : l);
    int idx = -1;
    if (r < m) idx = args.nn[(size_t)i * m + r];
    else if (r == MT) idx = i;
    Rec p;
    if (idx >= 0) {
      const double4 q = args.pts[idx];
      p.x = (q.x - ctr.x) * sc; p.y = (q.y - ctr.y) * sc; p.w = q.w;
      if constexpr (D3) p.z = (q.z - ctr.z) * sc;
    } else {
      p.x = kDummyCoord * (double)(r + 1); p.y = 0.0; p.w = 0.0;
      if constexpr (D3) p.z = 0.0;
    }
    own[s] = p;
    s_pts[g][r] = p;
  }
  __syncthreads();
  const double* tabv = s_tab;
  const Rec* gp = s_pts[g];
  int row_off[NS];      // byte offset of this lane's row of slot s inside the point's record block
#pragma unroll
  for (int s = 0; s < NS; ++s) row_off[s] = (16 * s + ((s & 1) ? 15 - l : l)) * (int)sizeof(Rec);

  // ---- assemble the augmented matrix, row-per-lane, in registers ----------------
  // include/GPBoost/cov_fcts.h:634-755 (CalculateCovMat) + Vecchia_utils.cpp:1599-1609
  double M[NS][L::NCOL];
  // one kernel evaluation; MODE_GRAD with kStoreDK also leaves d/dlog(a) of the entry in LDS (cov_fcts.h:2535-2554)
  auto eval_entry = [&](const Rec& o, const Rec& q, auto e_) -> double {
    const double d2 = sq_dist_s<D3>(o.x, o.y, o.zz(), q.x, q.y, q.zz());
    if constexpr (kStoreDK) {
      double dk;
      const double v = matern_cov_dlog_s<COV>(d2, tabv, dk);
      if constexpr (kLastDkInReg && decltype(e_)::value == NSTEP - 1) dk_last = dk;
      else s_dk[decltype(e_)::value][tid] = dk;
      return v;
    } else {
      return matern_cov_s<COV>(d2, tabv);
    }
  };
  for_each_lower_step<MT>(
      [&](auto s_, auto c_, auto e_) {                            // rect
        constexpr int s = decltype(s_)::value, c = decltype(c_)::value;
        M[s][c] = eval_entry(own[s], gp[c], e_);
      },
      [&](auto sA_, auto cA_, auto sB_, auto cB_, auto J_, auto e_) {   // pair
        constexpr int sA = decltype(sA_)::value, cA = decltype(cA_)::value, sB = decltype(sB_)::value,
                      cB = decltype(cB_)::value, J = decltype(J_)::value;
        // lanes l <= J: entry (slot sA, column cA); lanes l > J: entry (slot sB, column cB).  Both records come from LDS through
        // per-lane addresses picked with a constant lane mask (2 selects instead of 1 compare + 5 selects per step).
        constexpr unsigned long long MA = row_lanes_le(J);
        const int ro = sel_lanes<MA>(row_off[sA], row_off[sB]);
        const int co = sel_lanes<MA>(cA, cB) * (int)sizeof(Rec);
        const Rec o = *reinterpret_cast<const Rec*>(reinterpret_cast<const char*>(gp) + ro);
        const Rec q = *reinterpret_cast<const Rec*>(reinterpret_cast<const char*>(gp) + co);
        const double v = eval_entry(o, q, e_);
        M[sA][cA] = v;    // lanes of the other half hold entries above the diagonal there: never read
        M[sB][cB] = v;
      },
      [&](auto s_, auto c_, auto e_) {                            // solo
        constexpr int s = decltype(s_)::value, c = decltype(c_)::value;
        M[s][c] = eval_entry(own[s], gp[c], e_);
      });
  // diagonal: nugget / jitter (Vecchia_utils.cpp:1599-1609) and the first summand of D_i (:1555-1563): one v_mov_b64 under a constant
  // exec mask per diagonal entry (the lane that owns row c)
  static_for<0, NS>([&](auto s_) {
    constexpr int s = decltype(s_)::value;
    static_for<16 * s, L::cmax(s) + 1>([&](auto c_) {
      constexpr int c = decltype(c_)::value;
      const double dg = (c == MT) ? args.diag_i : args.diag_nn;
      if constexpr (c == 16 * s + 15 || c == MT) M[s][c] = dg;     // column never evaluated: plain init
      else set_lanes<row_lane_eq(lane_of_row(c))>(M[s][c], dg);
    });
  });
  // response row: entries are the gathered y's (only lane YL of slot YS; exec-masked, no DPP inside)
  if (l == L::YL) {
    static_for<0, MT + 1>([&](auto c_) { M[L::YS][decltype(c_)::value] = gp[decltype(c_)::value].w; });
  }

  // ---- right-looking LDL^T elimination of columns 0..MT-1 ------------------------
  // stands in for Eigen LLT + solve (Vecchia_utils.cpp:1617-1623); leaves unit-lower L (scaled) in M[.][k<MT],
  // D_i in entry (MT, MT) and u_i = (B y)_i in entry (MT+1, MT)
  static_for<0, MT>([&](auto k_) {
    constexpr int k = decltype(k_)::value;
    constexpr int sk = k / 16, lk = lane_of_row(k);
    // column-k registers become DPP sources in this sweep: fence them (they were last written by the assembly
    // for k == 0, by sweep k-1's first fmacs otherwise -- the verifier checks the generated code either way)
    if constexpr (k == 0) static_for<0, NS>([&](auto s_) { dpp_fence(M[decltype(s_)::value][0]); });
    const double piv = GPB_ROW_BCAST(lk, M[sk][k]);
    const double inv = fast_rcp(piv);
    double T[NS];
    static_for<sk, NS>([&](auto s_) { T[decltype(s_)::value] = M[decltype(s_)::value][k] * inv; });
    static_for<k + 1, MT + 1>([&](auto c_) {
      constexpr int c = decltype(c_)::value;
      constexpr int sc_ = c / 16, lc = lane_of_row(c);
      static_for<sc_, NS>([&](auto s_) {
        constexpr int s = decltype(s_)::value;
        GPB_ROW_FNMA(lc, M[s][c], M[sc_][k], T[s]);   // M[r][c] -= (L[c][k] d_k) * L[r][k]
      });
    });
    if constexpr (kNeedSolve) static_for<sk, NS>([&](auto s_) { M[decltype(s_)::value][k] = T[decltype(s_)::value]; });
  });
  // (a software-pipelined variant -- pivot k+1 started right after column k+1 of sweep k -- was measured at n = 1e6: no gain in any mode,
  //  and its extra live registers pushed MT >= 40 into AGPR spills next to DPP reads; not kept)

  const double Dv = GPB_ROW_BCAST(L::PL, M[L::PS][MT]);   // D_i  (Vecchia_utils.cpp:1623; the reference stores 1/D_i, :1682)
  const double uv = GPB_ROW_BCAST(L::YL, M[L::YS][MT]);   // u_i = (B y)_i
  const double Dinv = 1.0 / Dv;

  double red[GPB_NUM_PARTIALS];
#pragma unroll
  for (int t = 0; t < GPB_NUM_PARTIALS; ++t) red[t] = 0.0;
  red[GPB_P_LOGDET] = Dv;                      // the logarithm is taken once per workgroup (16 values, wave 0) below
  red[GPB_P_QUAD] = uv * uv * Dinv;
  red[GPB_P_BAD] = (Dv > 0.0) ? 0.0 : 1.0;

  if constexpr (kNeedSolve) {
    // ---- back-substitution x = L^-T (row of L), L unit lower: for the point's row (-> A_i, Vecchia_utils.cpp:1618)
    //      and, in the same instructions, the response row (-> b_i = C^-1 y_nn) ------------------------------
    double X[MT];
    static_for<0, MT>([&](auto k_) { X[decltype(k_)::value] = M[L::PS][decltype(k_)::value]; });
    // the scaled L columns were written by plain multiplies (compiler-scheduled): fence before the DPP reads
    static_for<0, MT>([&](auto k_) { static_for<0, NS>([&](auto s_) { if constexpr (decltype(k_)::value <= L::cmax(decltype(s_)::value)) dpp_fence(M[decltype(s_)::value][decltype(k_)::value]); }); });
    static_for_down<0, MT>([&](auto j_) {
      constexpr int j = decltype(j_)::value;
      constexpr int sj = j / 16, lj = lane_of_row(j);
      static_for_down<0, j>([&](auto k_) {      // k = j-1 first: X[j-1], the next sweep's multiplier, is final early
        constexpr int k = decltype(k_)::value;
        GPB_ROW_FNMA(lj, X[k], M[sj][k], X[j]);
      });
    });
    // lane PL now holds A_i, lane YL holds b_i.  No DPP below this line.
    if constexpr (MODE == MODE_FACTOR) {
      if (active && l == L::PL) {
        double* Arow = args.A + (size_t)i * m;
        static_for<0, MT>([&](auto k_) {
          constexpr int k = decltype(k_)::value;
          if (k < m) Arow[k] = X[k];
        });
      }
      if (active && l == 0) { args.D[i] = Dv; args.u[i] = uv; }
    }
    if constexpr (MODE == MODE_GRAD) {
      // extended vectors over rows 0..MT+1 as pairs: (A~_r, b~_r) with A~ = (A, -1, 0), b~ = (b, 0, 0)
      asm volatile("" ::: "memory");      // every read of the records precedes the pairs that overwrite them (kStoreDK)
      double2* gab = kStoreDK ? reinterpret_cast<double2*>(&s_pts[g][0]) : &s_ab[g][0];
      if (l == L::PL) { static_for<0, MT>([&](auto k_) { gab[decltype(k_)::value].x = X[decltype(k_)::value]; }); gab[MT].x = -1.0; gab[MT + 1].x = 0.0; }
      if (l == L::YL) { static_for<0, MT>([&](auto k_) { gab[decltype(k_)::value].y = X[decltype(k_)::value]; }); gab[MT].y = 0.0; gab[MT + 1].y = 0.0; }
      if (l == 0) { for (int r = MT + 2; r < NS * 16; ++r) gab[r] = make_double2(0.0, 0.0); }
      __syncthreads();
      // range parameter: accD = sum_{c<r<=MT} dK_rc A~_r A~_c ; accU = sum dK_rc (b~_r A~_c + b~_c A~_r)
      // (dD_range = 2 accD, (dB_range y)_i = accU; derivation in DESIGN.md, restating
      //  Vecchia_utils.cpp:1640-1652 without forming dA_i)
      double2 abr[NS];
      int ab_off[NS];
      double sAA = 0.0, sbA = 0.0;
      static_for<0, NS>([&](auto s_) {
        constexpr int s = decltype(s_)::value;
        const int r = 16 * s + ((s & 1) ? 15 - l : l);
        abr[s] = gab[r]; ab_off[s] = r * (int)sizeof(double2);
        if (r < MT) { sAA = __builtin_fma(abr[s].x, abr[s].x, sAA); sbA = __builtin_fma(abr[s].y, abr[s].x, sbA); }
      });
      double accD = 0.0, accU = 0.0;
      auto accumulate = [&](double dk, const double2& r_, const double2& c_) {
        accD = __builtin_fma(dk * r_.x, c_.x, accD);
        accU = __builtin_fma(dk, __builtin_fma(r_.y, c_.x, c_.y * r_.x), accU);
      };
      // d/dlog(a) of the entry of step e: from LDS (kStoreDK) or evaluated again
      auto dk_of = [&](auto e_, const Rec& o, const Rec& q) -> double {
        if constexpr (kStoreDK) {
          if constexpr (kLastDkInReg && decltype(e_)::value == NSTEP - 1) return dk_last;
          else return s_dk[decltype(e_)::value][tid];
        } else {
          return matern_dlog_range_s<COV>(sq_dist_s<D3>(o.x, o.y, o.zz(), q.x, q.y, q.zz()), tabv);
        }
      };
      for_each_lower_step<MT>(
          [&](auto s_, auto c_, auto e_) {
            constexpr int s = decltype(s_)::value, c = decltype(c_)::value;
            accumulate(dk_of(e_, own[s], gp[c]), abr[s], gab[c]);
          },
          [&](auto sA_, auto cA_, auto sB_, auto cB_, auto J_, auto e_) {
            constexpr int sA = decltype(sA_)::value, cA = decltype(cA_)::value, sB = decltype(sB_)::value,
                          cB = decltype(cB_)::value, J = decltype(J_)::value;
            constexpr unsigned long long MA = row_lanes_le(J);
            const int ci = sel_lanes<MA>(cA, cB);
            double dk;
            if constexpr (kStoreDK) dk = dk_of(e_, own[0], own[0]);
            else {
              const int ro = sel_lanes<MA>(row_off[sA], row_off[sB]);
              dk = dk_of(e_, *reinterpret_cast<const Rec*>(reinterpret_cast<const char*>(gp) + ro),
                         *reinterpret_cast<const Rec*>(reinterpret_cast<const char*>(gp) + ci * (int)sizeof(Rec)));
            }
            const int ao = sel_lanes<MA>(ab_off[sA], ab_off[sB]);
            accumulate(dk, *reinterpret_cast<const double2*>(reinterpret_cast<const char*>(gab) + ao), gab[ci]);
          },
          [&](auto s_, auto c_, auto e_) {
            constexpr int s = decltype(s_)::value, c = decltype(c_)::value;
            static_assert((s & 1) == 0, "solo steps only occur in even slots");
            double dk = dk_of(e_, own[s], gp[c]);
            set_lanes<row_lanes_le(c - 16 * s)>(dk, 0.0);              // rows <= c: entries on / above the diagonal do not exist
            accumulate(dk, abr[s], gab[c]);
          });
      // reduce the four accumulators over the 16 lanes of the row (xor butterflies stay inside the row)
#pragma unroll
      for (int off = 8; off >= 1; off >>= 1) {
        accD += __shfl_xor(accD, off, 16);
        accU += __shfl_xor(accU, off, 16);
        sAA += __shfl_xor(sAA, off, 16);
        sbA += __shfl_xor(sbA, off, 16);
      }
      const double up = uv * Dinv;                       // u' = D^-1 B y  (re_model_template.h:1999)
      // variance (ipar 0): dD = D - nugget - sum A^2 (Gaussian: nugget = 1), (dB y)_i = -sum b_r A_r
      const double dD_var = Dv - args.nugget - sAA;
      const double uk_var = -sbA;
      const double dD_rng = 2.0 * accD;
      const double uk_rng = accU;
      red[GPB_P_G1_VAR] = uk_var * up - 0.5 * up * up * dD_var;   // (uk.u - 0.5 u^T dD u) pieces (:2004)
      red[GPB_P_G2_VAR] = 0.5 * Dinv * dD_var;                    // 0.5 sum D^-1 dD
      red[GPB_P_G1_RNG] = uk_rng * up - 0.5 * up * up * dD_rng;
      red[GPB_P_G2_RNG] = 0.5 * Dinv * dD_rng;
    }
  }

  // ---- workgroup partial sums, fixed order; layout [term][workgroup] ---------------
  if (l == 0) {
#pragma unroll
    for (int t = 0; t < NP; ++t) s_red[t][g] = active ? red[t] : (t == GPB_P_LOGDET ? 1.0 : 0.0);
  }
  __syncthreads();
  if (tid < 16) s_red[GPB_P_LOGDET][tid] = log(s_red[GPB_P_LOGDET][tid]);   // sum log D_i (re_model_template.h:2946-2948); NaN for D_i <= 0 as before
  __syncthreads();
  if (tid < NP) {
    double acc = 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc += s_red[tid][q];
    args.partials[(size_t)tid * gridDim.x + blockIdx.x] = acc;
  }
}


// ---- launchers --------------------------------------------------------------------
// The heavy template is compiled once per padded neighbour count in its own translation unit
// (-DGPB_INSTANTIATE_MT=<MT>), so the build parallelises; the dispatcher TU has no template code.
#ifdef GPB_INSTANTIATE_MT
#ifndef GPB_INSTANTIATE_MODE
#error "define GPB_INSTANTIATE_MODE (0 nll, 1 factor, 2 grad) together with GPB_INSTANTIATE_MT"
#endif
template <int MT, bool D3>
static hipError_t launch_cov(int cov, const VecchiaKernelArgs& args, int nblocks, hipStream_t st) {
  switch (cov) {
    case kMatern05: hipLaunchKernelGGL((vecchia_point_kernel<MT, kMatern05, D3, GPB_INSTANTIATE_MODE>), dim3(nblocks), dim3(256), 0, st, args); break;
    case kMatern15: hipLaunchKernelGGL((vecchia_point_kernel<MT, kMatern15, D3, GPB_INSTANTIATE_MODE>), dim3(nblocks), dim3(256), 0, st, args); break;
    case kMatern25: hipLaunchKernelGGL((vecchia_point_kernel<MT, kMatern25, D3, GPB_INSTANTIATE_MODE>), dim3(nblocks), dim3(256), 0, st, args); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
#define GPB_CAT4_(a, b, c, d) a##b##c##d
#define GPB_CAT4(a, b, c, d) GPB_CAT4_(a, b, c, d)
hipError_t GPB_CAT4(launch_vecchia_mt, GPB_INSTANTIATE_MT, _mode, GPB_INSTANTIATE_MODE)(bool d3, int cov, const VecchiaKernelArgs& args,
                                                                                      int nblocks, hipStream_t st) {
  return d3 ? launch_cov<GPB_INSTANTIATE_MT, true>(cov, args, nblocks, st)
            : launch_cov<GPB_INSTANTIATE_MT, false>(cov, args, nblocks, st);
}
#else   // dispatcher TU
#define GPB_CASE(MTV)                                                                                   \
  hipError_t launch_vecchia_mt##MTV##_mode0(bool, int, const VecchiaKernelArgs&, int, hipStream_t);       \
  hipError_t launch_vecchia_mt##MTV##_mode1(bool, int, const VecchiaKernelArgs&, int, hipStream_t);       \
  hipError_t launch_vecchia_mt##MTV##_mode2(bool, int, const VecchiaKernelArgs&, int, hipStream_t);
GPB_MT_CASES
#undef GPB_CASE

int vecchia_padded_m(int m) {
  const int sizes[] = {GPB_MT_LIST};
  for (int s : sizes) if (m <= s) return s;
  return -1;
}

hipError_t launch_vecchia_point_kernel(int mode, int cov, bool d3, const VecchiaKernelArgs& args, hipStream_t st) {
  const int npts = args.i_end - args.i_begin;
  if (npts <= 0) return hipErrorInvalidValue;
  const int nblocks = (npts + 15) / 16;
  const int mt = vecchia_padded_m(args.m);
  switch (mt) {
#define GPB_CASE(MTV)                                                                       \
  case MTV:                                                                                   \
    if (mode == MODE_NLL) return launch_vecchia_mt##MTV##_mode0(d3, cov, args, nblocks, st);  \
    if (mode == MODE_FACTOR) return launch_vecchia_mt##MTV##_mode1(d3, cov, args, nblocks, st); \
    if (mode == MODE_GRAD) return launch_vecchia_mt##MTV##_mode2(d3, cov, args, nblocks, st);  \
    return hipErrorInvalidValue;
    GPB_MT_CASES
#undef GPB_CASE
    default: return hipErrorInvalidValue;
  }
}

#endif  // GPB_INSTANTIATE_MT

}  // namespace gpb
