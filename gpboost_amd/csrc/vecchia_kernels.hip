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
#include <algorithm>
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
  // LDS bytes per point group with 32-byte records (d = 3): NS * 16 records + HALF a record.  Consecutive lanes read consecutive records (stride
  // 8 dwords) with ds_read_b128, i.e. they touch the banks 0-3 mod 8 of the 64; a lane group of that instruction mixes lanes of two points
  // (MI355X_MICROARCH.md, LDS: {0-3, 12-15, 20-27}, ...), so the second point's block must start 4 mod 8 dwords later to take the other half of
  // the banks.  Round 4's stride of NS * 16 + 1 whole records (0 mod 8) put both points on the same banks: 2-way conflicts on every per-lane
  // record read, 50 - 54 % of the LDS cycles of the m = 40, d = 3 instances (profiles/r04_pmc.json).
  static constexpr int PTS_BYTES32 = (NS * 16) * 32 + 16;
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

constexpr unsigned long long kGranuleEmpty = ~0ull;    // "no sum here yet" in the workers' slots (a NaN pattern no arithmetic produces)
template <int NP>
__device__ __forceinline__ void vecchia_finish(const VecchiaKernelArgs& args, int G, double (*s_fin)[4]);

// MODE_NLL    : partial sums {sum log D, sum u^2/D, #(D<=0)} only
// MODE_FACTOR : additionally A[n][m], D[n], u[n] to HBM
// MODE_GRAD   : partial sums for the nll terms and the two parameter gradients
// 30 < MT <= 40, solving modes: TWO workgroups per CU, i.e. a 256-VGPR cap with 60 - 89 values in scratch.  Round 4 measured the alternative the
// spills suggest -- one workgroup per CU, 316 registers (60 of them AGPRs), no scratch, no DPP hazards: 6.30 ms against 5.40 ms per gradient launch at
// n = 1e6, d = 3, Matern-2.5, m = 40 (profiles/r04_j_grad_m40_one_workgroup_per_cu_ab.txt): the second wavefront per SIMD hides more than the spills cost.
// WT (round 5): sample weights (args.nug != nullptr) as a template argument -- the instances without weights carry neither the NS diagonal entries per lane nor the
// selects between them and the uniform diagonal (the d = 3, MT = 40 gradient instance had reached 256 VGPRs and spilled one).
template <int MT, int MODE>
constexpr bool kRuntimeWeightedInstance() { return MODE == MODE_GRAD && MT > 30 && MT <= 40; }

template <int MT, int COV, bool D3, int MODE, bool WT>
__global__ __launch_bounds__(256, (MT > 30 && MT <= 40 && MODE != MODE_NLL) ? 2 : 1) void vecchia_point_kernel(VecchiaKernelArgs args) {
  using L = Layout<MT>;
  constexpr int NS = L::NS;
  constexpr bool kNeedSolve = (MODE != MODE_NLL);
#ifdef GPB_NLL_RIGHT_LOOKING
  constexpr bool kLeftLooking = false;
#else
  // MODE_NLL with MT > 30: left-looking Cholesky with lazily evaluated columns (see below).  Measured at n = 1e6 (profiles/r03_b_*):
  // MT = 40 (d = 3, Matern-2.5) 288 -> 166 VGPRs, one -> three wavefronts per SIMD, 3.3 -> 2.19 ms per launch; MT = 30 gains no
  // occupancy from it (118 -> 124 VGPRs, four wavefronts either way) and loses 4 % to the extra rsq / fences: it keeps the right-looking form.
  // Round 5: the SOLVING modes with 30 < MT <= 46 (three register slots: config 5's m = 40) are left-looking too (kLeftSolve).  Right-looking they need
  // the whole triangle, 89 doubles, live from the first sweep on: 256 VGPRs + 60 - 89 values in scratch (8.3 GB of scratch traffic per gradient launch
  // at n = 1e6, profiles/r04_pmc.json).  Left-looking, what is live at column c is the columns < c of the slots that still have rows >= c, PLUS -- for
  // the back-substitution -- the finished pieces above them: slot 0's 16 columns are parked in LDS ([column][thread], 32 KB per workgroup: two
  // workgroups still fit a CU) once column 15 is done, slots 1 and 2 stay in registers (32 + 41 doubles): no scratch.  The factor is S = L sqrt(D)
  // (Cholesky) with 1 / S_kk parked in the diagonal lane of column k's own slot; the back-substitution below solves S^T x = (row MT / MT + 1 of S).
  constexpr bool kLeftLooking = MT > 30 && (!kNeedSolve || NS == 3);
#endif
  constexpr bool kLeftSolve = kLeftLooking && kNeedSolve;
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
  constexpr int PBYTES = D3 ? L::PTS_BYTES32 : L::PTS_STRIDE24 * (int)sizeof(Rec);      // bytes of one point's record block
  __shared__ double s_tab[GPB_EXP_TAB_SIZE];
  __shared__ __attribute__((aligned(16))) char s_pts_raw[16 * PBYTES];
  __shared__ double s_red[GPB_NUM_PARTIALS][16];
  __shared__ double s_f0[kLeftSolve ? 16 : 1][kLeftSolve ? 256 : 1];   // kLeftSolve: slot 0's pieces of the factor's columns 0..15, [column][thread]
  constexpr bool kDgInLds = WT && (MODE == MODE_NLL) && kLeftLooking;      // (MT <= 30 keeps them in registers: four wavefronts per SIMD either way)
  __shared__ double s_dg[kDgInLds ? 16 : 1][kDgInLds ? NS * 16 : 1];   // sample weights, MODE_NLL with MT > 30: diagonal entry (var + nugget_r) of every row of the point's system
  __shared__ double s_dk[NSTORE][kStoreDK ? 256 : 1];
  // (A~_r, b~_r) pairs of the contraction pass: with kStoreDK they live in the point's record block, which is dead by then (the same 16
  // lanes of one wavefront write and read it); the re-evaluating variant still needs the records and gets its own array
  static_assert(PBYTES % 16 == 0 && PBYTES >= 16 * (NS * 16), "record block: 16-byte aligned, room for the (A, b) pairs");
  __shared__ double2 s_ab[(MODE == MODE_GRAD && !kStoreDK) ? 16 : 1][(MODE == MODE_GRAD && !kStoreDK) ? NS * 16 : 1];
  double dk_last = 0.0;

  __shared__ double s_tot[GPB_NUM_PARTIALS];          // this workgroup's sums over all its groups of 16 points
  __shared__ double s_fin[GPB_NUM_PARTIALS][4];

  const int G = (int)gridDim.x - 1;                     // workers; workgroup G is the finisher
  if ((int)blockIdx.x == G) { vecchia_finish<NP>(args, G, s_fin); return; }
  const int tid0 = threadIdx.x;
  static_assert(GPB_EXP_TAB_SIZE == 256, "one table entry per thread");
  s_tab[tid0] = args.exp_tab[tid0] * args.var;          // var * 2^(j/256)
  if (tid0 < NP) s_tot[tid0] = 0.0;                      // (term t is read and written by thread 64 + t only until the loop ends)
  // log|Psi| = sum log D_i: thread q < 16 keeps the running PRODUCT of the D_i of "its" point of every group as (mantissa in [0.5, 1),
  // exponent) and takes ONE logarithm at the end -- a log per group inside the loop costs little time but its dozen constants would be
  // live in VGPRs across the whole loop (with the other loop invariants: 142 instead of 128 registers, three wavefronts per SIMD
  // instead of four).  Rounding: one multiply + exact rescaling per point; D_i <= 0 still ends in NaN / -inf as log(D_i) did.
  __shared__ double s_pm[16];
  __shared__ int s_pe[16];
  if (tid0 < 16) { s_pm[tid0] = 0.5; s_pe[tid0] = 1; }   // 0.5 * 2^1 = 1

  // PERSISTENT workgroups: the grid is one resident round of workgroups (host: occupancy x CUs, trimmed so that every workgroup makes
  // the same number of trips +- 1); workgroup b takes the groups b, b + G, b + 2 G, ... of 16 points.  The table set-up, the kernel-argument
  // loads and the workgroup launch are paid once per ~60 groups instead of once per group, and there are ~1000 partial sums per term at
  // the end instead of 62 500 -- few enough for the LAST workgroup to add them up itself (below): one launch per evaluation.
  // The neighbour INDICES of the next trip are fetched at the top of the current one (3 VGPRs): one of the two dependent round trips of
  // every gather is gone.  (Measured and dropped: a start delay hashed from the workgroup id, to de-correlate the phases of workgroups that
  // all start together and make equally long trips -- no gain.  What does matter is the NUMBER of workgroups: with exactly one resident
  // round every workgroup runs start to end on "its" CU and the slowest CU sets the time; several rounds balance -- see persistent_grid.)
  const int m = args.m;
  auto fetch_idx = [&](int grp_, int tid_, int (&out)[NS]) {
    const int g_ = tid_ >> 4, l_ = tid_ & 15;
    const long long ir = (long long)args.i_begin + (long long)grp_ * 16 + g_;
    const int ii = ir < (long long)args.i_end ? (int)ir : args.i_end - 1;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int r = 16 * s + ((s & 1) ? 15 - l_ : l_);
      int idx = -1;
      if (r < m) idx = args.nn[(size_t)ii * m + r];
      else if (r == MT) idx = ii;
      out[s] = idx;
    }
  };
  int nidx[NS];
  fetch_idx(blockIdx.x, tid0, nidx);
  for (int grp = blockIdx.x; grp < args.ngroups; grp += G) {
  // the thread index is made opaque once per trip: everything derived from it (row indices, LDS offsets, dummy coordinates, lane
  // predicates: ~35 VGPRs' worth) is then recomputed per trip -- as the one-group-per-workgroup kernel did -- instead of being hoisted
  // out of the loop and held live across it (132 -> <= 128 VGPRs: four wavefronts per SIMD)
  int tid = tid0;
  asm volatile("; per-trip thread index" : "+v"(tid));
  const int g = tid >> 4;   // point within the workgroup
  const int l = tid & 15;   // lane within the point's DPP row
  const long long i_raw = (long long)args.i_begin + (long long)grp * 16 + g;
  const bool active = i_raw < (long long)args.i_end;
  const int i = active ? (int)i_raw : args.i_end - 1;   // inactive groups redo the last point, contribute 0
  int cidx[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) cidx[s] = nidx[s];
  if (grp + G < args.ngroups) fetch_idx(grp + G, tid, nidx);   // (uniform) the next trip's neighbour indices
  const double sc = args.a * kCoordScale;               // half-scaled coordinates: squared distances are (rho/2)^2, exp(-a d) = 2^(-rho/256)

  // ---- gather the rows' records: centred on the point (differences of nearby points stay accurate for
  //      coordinates with a large offset), scaled, staged in LDS for the column operands ------------------
  const double4 ctr = args.pts[i];
  Rec own[NS];
  // sample weights (Gaussian likelihood: observation-specific nugget 1 / w on the transformed scale, GetGaussianNuggetDiagFromWeights,
  // re_model_template.h:6393-6417; Vecchia_utils.cpp:1418-1422, 1610-1614): own_dg[s] = diagonal entry of this lane's row of slot s
  // MODE_NLL with MT > 30 (round 5): the entries wait in LDS, s_dg[point][row], not in NS doubles per lane that are live across the whole factorisation -- those had
  // pushed the d = 3, MT = 40 likelihood instance from 166 to 174 VGPRs, i.e. from three wavefronts per SIMD to two: the 2.20 -> 2.42 ms regression of round 3.
  // The solving modes are not register-limited at that point and MODE_GRAD with stored derivatives has no LDS to spare (two workgroups per CU): registers there.
  // (the launcher picks the instance by args.nug != nullptr -- except for the gradient instance of 30 < MT <= 40: there the WT = true instance serves both cases
  //  with the run-time test of rounds 3-4.  Measured at config 5's shape, d = 3 Matern-2.5: 4.93 ms that way; the compile-time unweighted instance got 256 VGPRs + 5
  //  spilled and 5.24 ms, the compile-time weighted one on unit nuggets 247 VGPRs and 5.12 ms -- profiles/r05_l_*, r05_m_*)
  constexpr bool kRuntimeW = WT && kRuntimeWeightedInstance<MT, MODE>();
  const bool weighted = kRuntimeW ? (args.nug != nullptr) : WT;
  constexpr int kOwnDg = (kDgInLds || !WT) ? 1 : NS;
  double own_dg[kOwnDg];
#pragma unroll
  for (int s = 0; s < kOwnDg; ++s) own_dg[s] = 0.0;
  auto own_dg_of = [&](int s) -> double { if constexpr (kOwnDg == NS) return own_dg[s]; else { (void)s; return own_dg[0]; } };
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int r = 16 * s + ((s & 1) ? 15 - l : l);
    const int idx = cidx[s];
    if (WT && weighted) {
      const double dgv = args.var + (idx >= 0 ? args.nug[idx] : 1.0);
      if constexpr (kDgInLds) s_dg[g][r] = dgv; else own_dg[s] = dgv;
    }
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
    reinterpret_cast<Rec*>(s_pts_raw + g * PBYTES)[r] = p;
  }
  __syncthreads();
  const double* tabv = s_tab;
  const Rec* gp = reinterpret_cast<const Rec*>(s_pts_raw + g * PBYTES);
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
  if constexpr (kLeftLooking) {
    // ---- MODE_NLL: LEFT-looking Cholesky, column by column: kernel entries of column c are evaluated when the column's turn comes, the
    // updates of all earlier columns are applied to it (same DPP fmacs, same order per entry as the right-looking sweeps below), then it is
    // scaled by 1 / sqrt(pivot): S = L sqrt(D), so that ONE register per (slot, column) serves as broadcast source and as multiplicand.
    // Why: what is live at column c is, for the slots that still have rows >= c, the columns < c -- at most 2 x 32 doubles for MT = 40
    // (right-looking: the whole triangle, 89 doubles, from the first sweep on): ~290 -> <= 256 VGPRs = two wavefronts per SIMD instead of
    // one for 31 <= MT <= 46, and ~30 VGPRs less for MT = 30.  D_i and u_i = (B y)_i are entries (MT, MT) and (MT + 1, MT) as before.
    auto eval_plain = [&](const Rec& o, const Rec& q) -> double {
      return matern_cov_s<COV>(sq_dist_s<D3>(o.x, o.y, o.zz(), q.x, q.y, q.zz()), tabv);
    };
    static_for<0, MT + 1>([&](auto c_) {
      constexpr int c = decltype(c_)::value;
      constexpr int sc_ = c / 16, lc = lane_of_row(c);
      // (1) kernel entries of column c: whole slots below the column's own slot ...
      if constexpr (c < MT) {
        static_for<sc_ + 1, NS>([&](auto s_) { constexpr int s = decltype(s_)::value; M[s][c] = eval_plain(own[s], gp[c]); });
        // ... and the sub-diagonal piece inside the own slot.  Even slots: the step that evaluates it also evaluates, on the other lanes, the
        // piece of column cA of the (mirrored) odd slot above -- kept until cA's turn; odd slots therefore have theirs already.
        if constexpr ((c % 16) != 15 && (sc_ & 1) == 0) {
          constexpr int j = 14 - (c - 16 * sc_);
          constexpr int sA = sc_ + 1, cA = 16 * sA + j;
          if constexpr (sA < NS && cA <= MT - 1) {
            constexpr unsigned long long MA = row_lanes_le(14 - j);
            const int ro = sel_lanes<MA>(row_off[sA], row_off[sc_], grp);
            const int co = sel_lanes_const<MA, cA, c>(grp) * (int)sizeof(Rec);
            const double v = eval_plain(*reinterpret_cast<const Rec*>(reinterpret_cast<const char*>(gp) + ro),
                                        *reinterpret_cast<const Rec*>(reinterpret_cast<const char*>(gp) + co));
            M[sA][cA] = v;
            M[sc_][c] = v;
          } else {
            M[sc_][c] = eval_plain(own[sc_], gp[c]);
          }
        }
      }
      // (2) diagonal (nugget / jitter, Vecchia_utils.cpp:1599-1609; first summand of D_i, :1555-1563) and the response row's entry
      {
        double dg;      // (weighted: row c's entry by a broadcast read [MODE_NLL] / every lane offers its own row's entry [solving modes]; the mask picks the owner)
        if constexpr (kDgInLds) dg = weighted ? s_dg[g][c] : ((c == MT) ? args.diag_i : args.diag_nn);
        else dg = weighted ? own_dg_of(sc_) : ((c == MT) ? args.diag_i : args.diag_nn);
        if constexpr (c == 16 * sc_ + 15 || c == MT) M[sc_][c] = dg;     // own-slot piece never evaluated: plain init
        else set_lanes<row_lane_eq(lc)>(M[sc_][c], dg);
        set_lanes<row_lane_eq(L::YL)>(M[L::YS][c], gp[c].w);
      }
      // (3) updates from the columns before it: M[r][c] -= S[c][k] S[r][k]
      static_for<0, c>([&](auto k_) {
        constexpr int k = decltype(k_)::value;
        static_for<sc_, NS>([&](auto s_) {
          constexpr int s = decltype(s_)::value;
          GPB_ROW_FNMA(lc, M[s][c], M[sc_][k], M[s][k]);
        });
      });
      // (4) scale by 1 / sqrt(pivot): v_rsq_f64 + one Newton step
      if constexpr (c < MT) {
        const double piv = GPB_ROW_BCAST(lc, M[sc_][c]);             // (row_bcast carries its own two wait states)
        const double y0 = __builtin_amdgcn_rsq(piv);
        const double e = __builtin_fma(-0.5 * piv * y0, y0, 0.5);
        const double rs = __builtin_fma(y0, e, y0);
        static_for<sc_, NS>([&](auto s_) { constexpr int s = decltype(s_)::value; M[s][c] *= rs; });
        // kLeftSolve: the diagonal lane keeps 1 / S_cc (no update ever reads lane lc of column c's own slot: rows r > c only)
        if constexpr (kLeftSolve) set_lanes<row_lane_eq(lc)>(M[sc_][c], rs);
        // the scaled column was written by plain multiplies (compiler-scheduled) and is a DPP source from the next column on: ONE fence
        static_assert(NS <= 4, "dpp_fence overloads cover four slots");
        if constexpr (NS - sc_ == 1) dpp_fence(M[sc_][c]);
        else if constexpr (NS - sc_ == 2) dpp_fence(M[sc_][c], M[sc_ + 1][c]);
        else if constexpr (NS - sc_ == 3) dpp_fence(M[sc_][c], M[sc_ + 1][c], M[sc_ + 2][c]);
        else dpp_fence(M[sc_][c], M[sc_ + 1][c], M[sc_ + 2][c], M[sc_ + 3][c]);
        // kLeftSolve: slot 0's pieces are dead for the elimination once its last column is done -- parked in LDS for the back-substitution
        if constexpr (kLeftSolve && c == 15) static_for<0, 16>([&](auto q_) { s_f0[decltype(q_)::value][tid] = M[0][decltype(q_)::value]; });
      }
    });
  } else {
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
        const int ro = sel_lanes<MA>(row_off[sA], row_off[sB], grp);
        const int co = sel_lanes_const<MA, cA, cB>(grp) * (int)sizeof(Rec);
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
      double dg;       // (weighted: every lane offers its own row's entry / row c's entry by a broadcast read, the mask picks the owner's)
      if constexpr (kDgInLds) dg = weighted ? s_dg[g][c] : ((c == MT) ? args.diag_i : args.diag_nn);
      else dg = weighted ? own_dg_of(s) : ((c == MT) ? args.diag_i : args.diag_nn);
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
  }   // right-looking (MODE_FACTOR / MODE_GRAD: the back-substitution below needs the whole unit-lower factor)

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
    // Two forms.  MT <= 30: the solution vectors live as X[k] in EVERY lane (lane PL carries A, lane YL carries b), one DPP fmac per
    // (j, k): MT (MT - 1) / 2 instructions, 2 MT registers.  MT > 30 (kLaneX): the solution is DISTRIBUTED -- lane r holds A_r and b_r of its
    // own row -- and x_k = l_k - sum_{r > k} L[r][k] x_r is a 16-lane sum per column: ~2 x 17 instructions per column instead of k fmacs,
    // but 4 NS registers instead of 2 MT (MT = 40: 80 VGPRs less at the point where the whole factor is live -- the difference between
    // one wavefront per SIMD with AGPR spills next to padded DPP reads and two).
    constexpr bool kLaneX = MT > 30;
    double X[kLaneX ? 1 : MT];
    double xa[NS], xb[NS];                          // kLaneX: A_r, b_r of this lane's rows
    if constexpr (kLeftSolve) {
      // S^T x = s (s = row MT / MT + 1 of S = S^-1 c / S^-1 y_nn): x_k = (s_k - sum_{r > k} S[r][k] x_r) / S_kk, lane r holds x_r of its own rows; the
      // column's 16-lane sums as in the LDL^T form below, 1 / S_kk from the diagonal lane, slot 0's pieces from LDS
      static_assert(NS == 3 && L::PS == 2 && L::YS == 2, "kLeftSolve: three slots, the point's and the response row in the last one");
#pragma unroll
      for (int s = 0; s < NS; ++s) { xa[s] = 0.0; xb[s] = 0.0; }
      static_for_down<0, MT>([&](auto k_) {
        constexpr int k = decltype(k_)::value;
        constexpr int sk = k / 16, lk = lane_of_row(k);
        constexpr unsigned long long kKeep = (sk & 1) ? row_lanes_le(14 - (k % 16)) : ~row_lanes_le(k % 16);
        double own_col;                       // column k's piece in its own slot
        if constexpr (sk == 0) own_col = s_f0[k][tid]; else own_col = M[sk][k];
        double ta = own_col * xa[sk], tb = own_col * xb[sk];
        set_lanes<~kKeep>(ta, 0.0);
        set_lanes<~kKeep>(tb, 0.0);
        static_for<sk + 1, NS>([&](auto s_) {       // rows of later slots: MT, MT + 1 and the padding rows carry x = 0
          constexpr int s = decltype(s_)::value;
          ta = __builtin_fma(M[s][k], xa[s], ta);
          tb = __builtin_fma(M[s][k], xb[s], tb);
        });
        ta = row_sum16(ta); tb = row_sum16(tb);
        const double rk = GPB_ROW_BCAST(lk, own_col);               // 1 / S_kk  (row_bcast carries its own wait states)
        const double la = GPB_ROW_BCAST(L::PL, M[L::PS][k]);        // S[MT][k]
        const double lb = GPB_ROW_BCAST(L::YL, M[L::YS][k]);        // S[MT + 1][k]
        set_lanes<row_lane_eq(lk)>(xa[sk], (la - ta) * rk);
        set_lanes<row_lane_eq(lk)>(xb[sk], (lb - tb) * rk);
      });
    } else if constexpr (kLaneX) {
#pragma unroll
      for (int s = 0; s < NS; ++s) { xa[s] = 0.0; xb[s] = 0.0; }
      static_for_down<0, MT>([&](auto k_) {
        constexpr int k = decltype(k_)::value;
        constexpr int sk = k / 16, lk = lane_of_row(k);
        // rows > k of the column's own slot: lanes above (even slot) / below (mirrored odd slot) the owner of row k; everything else 0
        constexpr unsigned long long kKeep = (sk & 1) ? row_lanes_le(14 - (k % 16)) : ~row_lanes_le(k % 16);
        double ta = M[sk][k] * xa[sk], tb = M[sk][k] * xb[sk];
        set_lanes<~kKeep>(ta, 0.0);
        set_lanes<~kKeep>(tb, 0.0);
        static_for<sk + 1, NS>([&](auto s_) {       // rows of later slots: MT, MT + 1 and the padding rows carry x = 0
          constexpr int s = decltype(s_)::value;
          ta = __builtin_fma(M[s][k], xa[s], ta);
          tb = __builtin_fma(M[s][k], xb[s], tb);
        });
        ta = row_sum16(ta); tb = row_sum16(tb);
        const double la = GPB_ROW_BCAST(L::PL, M[L::PS][k]);       // L[MT][k]: the point's row
        const double lb = GPB_ROW_BCAST(L::YL, M[L::YS][k]);       // L[MT + 1][k]: the response row
        set_lanes<row_lane_eq(lk)>(xa[sk], la - ta);
        set_lanes<row_lane_eq(lk)>(xb[sk], lb - tb);
      });
    } else {
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
    }
    // lane PL now holds A_i, lane YL holds b_i (kLaneX: every lane its own rows' entries).  No DPP below this line.
    if constexpr (MODE == MODE_FACTOR) {
      if constexpr (kLaneX) {
        if (active) {
          double* Arow = args.A + (size_t)i * m;
#pragma unroll
          for (int s = 0; s < NS; ++s) { const int r = 16 * s + ((s & 1) ? 15 - l : l); if (r < m) Arow[r] = xa[s]; }
        }
      } else {
      if (active && l == L::PL) {
        double* Arow = args.A + (size_t)i * m;
        static_for<0, MT>([&](auto k_) {
          constexpr int k = decltype(k_)::value;
          if (k < m) Arow[k] = X[k];
        });
      }
      }
      if (active && l == 0) { args.D[i] = Dv; args.u[i] = uv; }
    }
    if constexpr (MODE == MODE_GRAD) {
      // extended vectors over rows 0..MT+1 as pairs: (A~_r, b~_r) with A~ = (A, -1, 0), b~ = (b, 0, 0)
      asm volatile("" ::: "memory");      // every read of the records precedes the pairs that overwrite them (kStoreDK)
      double2* gab = kStoreDK ? reinterpret_cast<double2*>(s_pts_raw + g * PBYTES) : &s_ab[g][0];
      if constexpr (kLaneX) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          const int r = 16 * s + ((s & 1) ? 15 - l : l);
          gab[r] = r < MT ? make_double2(xa[s], xb[s]) : (r == MT ? make_double2(-1.0, 0.0) : make_double2(0.0, 0.0));
        }
      } else {
      if (l == L::PL) { static_for<0, MT>([&](auto k_) { gab[decltype(k_)::value].x = X[decltype(k_)::value]; }); gab[MT].x = -1.0; gab[MT + 1].x = 0.0; }
      if (l == L::YL) { static_for<0, MT>([&](auto k_) { gab[decltype(k_)::value].y = X[decltype(k_)::value]; }); gab[MT].y = 0.0; gab[MT + 1].y = 0.0; }
      if (l == 0) { for (int r = MT + 2; r < NS * 16; ++r) gab[r] = make_double2(0.0, 0.0); }
      }
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
        // variance parameter: dD = D - nug_i - sum_r nug_r A_r^2, (dB y)_i = -sum_r nug_r b_r A_r (uniform nugget 1 without weights)
        const double nr = weighted ? own_dg_of(s) - args.var : 1.0;      // (MODE_GRAD: !kDgInLds)
        if (r < MT) { sAA = __builtin_fma(nr * abr[s].x, abr[s].x, sAA); sbA = __builtin_fma(nr * abr[s].y, abr[s].x, sbA); }
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
            // (re-evaluating variant: the lane's own record is read back from LDS here -- as a register it would be live across the whole back-substitution)
            if constexpr (kStoreDK) accumulate(dk_of(e_, own[s], gp[c]), abr[s], gab[c]);
            else accumulate(dk_of(e_, *reinterpret_cast<const Rec*>(reinterpret_cast<const char*>(gp) + row_off[s]), gp[c]), abr[s], gab[c]);
          },
          [&](auto sA_, auto cA_, auto sB_, auto cB_, auto J_, auto e_) {
            constexpr int sA = decltype(sA_)::value, cA = decltype(cA_)::value, sB = decltype(sB_)::value,
                          cB = decltype(cB_)::value, J = decltype(J_)::value;
            constexpr unsigned long long MA = row_lanes_le(J);
            const int ci = sel_lanes_const<MA, cA, cB>(grp);
            double dk;
            if constexpr (kStoreDK) dk = dk_of(e_, own[0], own[0]);
            else {
              const int ro = sel_lanes<MA>(row_off[sA], row_off[sB], grp);
              dk = dk_of(e_, *reinterpret_cast<const Rec*>(reinterpret_cast<const char*>(gp) + ro),
                         *reinterpret_cast<const Rec*>(reinterpret_cast<const char*>(gp) + ci * (int)sizeof(Rec)));
            }
            const int ao = sel_lanes<MA>(ab_off[sA], ab_off[sB], grp);
            accumulate(dk, *reinterpret_cast<const double2*>(reinterpret_cast<const char*>(gab) + ao), gab[ci]);
          },
          [&](auto s_, auto c_, auto e_) {
            constexpr int s = decltype(s_)::value, c = decltype(c_)::value;
            static_assert((s & 1) == 0, "solo steps only occur in even slots");
            double dk;
            if constexpr (kStoreDK) dk = dk_of(e_, own[s], gp[c]);
            else dk = dk_of(e_, *reinterpret_cast<const Rec*>(reinterpret_cast<const char*>(gp) + row_off[s]), gp[c]);
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
      double nug_i = args.nugget;
      if (WT && weighted) nug_i = args.nug[i];
      const double dD_var = Dv - nug_i - sAA;
      const double uk_var = -sbA;
      const double dD_rng = 2.0 * accD;
      const double uk_rng = accU;
      red[GPB_P_G1_VAR] = uk_var * up - 0.5 * up * up * dD_var;   // (uk.u - 0.5 u^T dD u) pieces (:2004)
      red[GPB_P_G2_VAR] = 0.5 * Dinv * dD_var;                    // 0.5 sum D^-1 dD
      red[GPB_P_G1_RNG] = uk_rng * up - 0.5 * up * up * dD_rng;
      red[GPB_P_G2_RNG] = 0.5 * Dinv * dD_rng;
    }
  }

  // ---- sums of this group of 16 points, fixed order, added to the workgroup's running sums ---------------
  if (l == 0) {
#pragma unroll
    for (int t = 0; t < NP; ++t) s_red[t][g] = active ? red[t] : (t == GPB_P_LOGDET ? 1.0 : 0.0);
  }
  __syncthreads();
  if (tid < 16) {                                        // sum log D_i (re_model_template.h:2946-2948) as a running product, see above
    const double dv = s_red[GPB_P_LOGDET][tid];
    const double p = s_pm[tid] * __builtin_amdgcn_frexp_mant(dv);                    // in [0.25, 1) for D_i > 0
    s_pm[tid] = (dv > 0.0) ? __builtin_amdgcn_frexp_mant(p) : __builtin_nan("");   // log(D_i) is NaN for D_i <= 0 (two negative D_i must not cancel)
    s_pe[tid] += __builtin_amdgcn_frexp_exp(dv) + __builtin_amdgcn_frexp_exp(p);
  } else if (tid >= 64 && tid < 64 + NP && tid - 64 != GPB_P_LOGDET) {               // (a lane of another wavefront: nothing waits for wave 0)
    const int t = tid - 64;
    double acc = 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc += s_red[t][q];
    s_tot[t] += acc;
  }
  }   // groups of this workgroup (s_red is next written behind the following trip's first barrier: its readers have passed by then)
  const int tid = tid0;
  __syncthreads();
  if (tid < 16) s_red[GPB_P_LOGDET][tid] = log(s_pm[tid]) + (double)s_pe[tid] * 0.693147180559945309417232121458;
  __syncthreads();
  if (tid == 0) {
    double acc = 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc += s_red[GPB_P_LOGDET][q];
    s_tot[GPB_P_LOGDET] = acc;
  }
  __syncthreads();

  // ---- publish this workgroup's NP sums: ONE write-through (sc1) 8-byte store per term, fire and forget -- no drain, no ticket, no barrier:
  // a worker's tail costs what the plain store of a partial sum always cost.  The granule protocol of cdna_hip_programming.md Guideline 16
  // (R2: "the data IS the flag"): every slot holds the sentinel kGranuleEmpty until its sum arrives; the finisher below polls for it.
  if (tid < NP)
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(args.partials + (size_t)tid * G + blockIdx.x),
                       (unsigned long long)__double_as_longlong(s_tot[tid]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The LAST workgroup of the grid (index G = number of workers; dispatched after every worker, so all of them are running or done when it
// starts: it can wait for them) adds up the workers' sums in a FIXED order -- thread t takes workers t, t + 256, ... with a compensated
// sum, then a fixed tree -- whatever order they arrive in: bit-reproducible.  It polls each granule with L1-bypassing (sc1) loads until
// the sentinel is gone, puts the sentinel back for the next launch (launches on a handle are stream-ordered), and hands the launch's sums
// to d_out, the caller's device buffer and the pinned host buffer.  (A ticket counter with a last-arriver reduction was measured first:
// every worker then waits ~3 us for its ticket, 0.88 ms instead of 0.85 at 16 workgroups per CU.)
template <int NP>
__device__ __forceinline__ void vecchia_finish(const VecchiaKernelArgs& args, int G, double (*s_fin)[4]) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int t = 0; t < NP; ++t) {
    double acc = 0.0, comp = 0.0;     // Kahan on the per-thread chain
    unsigned long long* base = reinterpret_cast<unsigned long long*>(args.partials + (size_t)t * G);
    constexpr int kBatch = 8;         // granules polled together: eight loads in flight per lane, not one dependent round trip per granule
    for (int b0 = tid; b0 < G; b0 += 256 * kBatch) {
      unsigned long long raw[kBatch];
#pragma unroll
      for (int q = 0; q < kBatch; ++q) {
        const int b = b0 + 256 * q;
        raw[q] = b < G ? __hip_atomic_load(base + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
      }
#pragma unroll
      for (int q = 0; q < kBatch; ++q) {
        const int b = b0 + 256 * q;
        if (b >= G) break;
        while (raw[q] == kGranuleEmpty) {
          __builtin_amdgcn_s_sleep(8);
          raw[q] = __hip_atomic_load(base + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __hip_atomic_store(base + b, kGranuleEmpty, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const double v = __longlong_as_double((long long)raw[q]) - comp;     // (fixed order: b ascending)
        const double tmp = acc + v;
        comp = (tmp - acc) - v;
        acc = tmp;
      }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if ((tid & 63) == 0) s_fin[t][tid >> 6] = acc;
  }
  __syncthreads();
  if (tid < NP) {
    const double v = (s_fin[tid][0] + s_fin[tid][1]) + (s_fin[tid][2] + s_fin[tid][3]);
    args.out[tid] = v;
    // caller-facing layout {quad, logdet, bad, g1v, g2v, g1r, g2r}: terms 0 and 1 swapped w.r.t. GPB_P_*
    if (args.out_user) args.out_user[tid == GPB_P_LOGDET ? 1 : (tid == GPB_P_QUAD ? 0 : tid)] = v;
    // pinned, coherent host memory: the host polls for these stores (vecchia_fetch), no copy and no synchronisation on the stream
    if (args.out_host) __hip_atomic_store(reinterpret_cast<unsigned long long*>(args.out_host + tid), (unsigned long long)__double_as_longlong(v),
                                          __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}


// ---- launchers --------------------------------------------------------------------
// The heavy template is compiled once per padded neighbour count in its own translation unit
// (-DGPB_INSTANTIATE_MT=<MT>), so the build parallelises; the dispatcher TU has no template code.
#ifdef GPB_INSTANTIATE_MT
#ifndef GPB_INSTANTIATE_MODE
#error "define GPB_INSTANTIATE_MODE (0 nll, 1 factor, 2 grad) together with GPB_INSTANTIATE_MT"
#endif
// Persistent workers: `rounds` resident rounds of workgroups -- (workgroups the occupancy calculator admits per CU for THIS instantiation)
// x CUs x rounds, trimmed to ceil(ngroups / trips) so that all workers make the same number of trips (+- 1).  With ONE round every
// worker runs start to end on "its" CU and the slowest CU sets the time (measured at n = 1e6, m = 30: 0.93 ms); a few rounds let the
// dispatcher balance (2 rounds 0.893 ms, 4 rounds 0.885 ms, 8 rounds 0.874 ms; profiles/r03_a_*) while the table set-up, the argument
// loads and the launch of a workgroup are still paid once per ~8 groups.  The occupancy and the CU count are asked once per instantiation and device.
template <int MT, int COV, bool D3, bool WT>
static int persistent_grid(const VecchiaKernelArgs& args) {
  auto kern = vecchia_point_kernel<MT, COV, D3, GPB_INSTANTIATE_MODE, WT>;
  static int cached_dev = -1, per_dev = 0;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return args.ngroups;
  if (dev != cached_dev) {
    int occ = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 256, 0) != hipSuccess || occ < 1) occ = 1;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
    per_dev = occ * cus; cached_dev = dev;
    (void)hipGetLastError();
  }
  // default: about four trips per worker, at most eight rounds (n = 1e6: 8 rounds, 0.874 ms; an 8-GPU shard of 125 000 points: 2 rounds, 0.123 ms)
  const int auto_rounds = (int)std::max(1LL, std::min(8LL, ((long long)args.ngroups + 2LL * per_dev) / (4LL * per_dev)));
  const long long slots = (long long)per_dev * auto_rounds;
  if (args.ngroups <= slots) return args.ngroups;
  const int trips = (int)((args.ngroups + slots - 1) / slots);
  return (args.ngroups + trips - 1) / trips;
}
template <int MT, bool D3>
static hipError_t launch_cov(int cov, const VecchiaKernelArgs& args, int nblocks, hipStream_t st) {
  (void)nblocks;
  switch (cov) {
#define GPB_LAUNCH_COV(C) \
  if (args.nug || kRuntimeWeightedInstance<MT, GPB_INSTANTIATE_MODE>()) hipLaunchKernelGGL((vecchia_point_kernel<MT, C, D3, GPB_INSTANTIATE_MODE, true>), dim3(persistent_grid<MT, C, D3, true>(args) + 1), dim3(256), 0, st, args); \
  else hipLaunchKernelGGL((vecchia_point_kernel<MT, C, D3, GPB_INSTANTIATE_MODE, false>), dim3(persistent_grid<MT, C, D3, false>(args) + 1), dim3(256), 0, st, args); \
  break
    case kMatern05: GPB_LAUNCH_COV(kMatern05);
    case kMatern15: GPB_LAUNCH_COV(kMatern15);
    case kMatern25: GPB_LAUNCH_COV(kMatern25);
#undef GPB_LAUNCH_COV
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
  if (args.ngroups != nblocks || !args.out) return hipErrorInvalidValue;   // persistent kernel: groups of 16 points, result buffer
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
