// gpboost_amd/csrc/hist_kernels.hip
//
// LightGBM feature-histogram build for gfx950.  Restates, for dense uint8 bins,
//   DenseBin<uint8_t,false>::ConstructHistogramInner   src/LightGBM/io/dense_bin.hpp:98-141
//   Dataset::ConstructHistogramsInner                  src/LightGBM/io/dataset.cpp:1143-1245
// i.e. for every row of the leaf: hist[f][bin(row,f)].grad += g[row]; and either .hess += h[row]
// or -- constant hessian -- ++count (converted to count * hess afterwards, dataset.cpp:1223-1226).
//
// MI355X mapping (HBM-bound byte work, no MFMA):
//   * bins are re-laid out once, at create time, from the reference's feature-major storage to
//     row-major [n][fpad]: a leaf's (gathered) rows then cost one 16-byte access per 16 features
//     instead of one byte per cache line;
//   * a workgroup owns 16 features x one chunk of rows; one lane = one row: a 16-byte load brings the
//     row's 16 bins, then the lane issues 16 LDS atomics -- at any instant all lanes of a wavefront update
//     the SAME feature's sub-histogram, so they only collide when two rows share a bin;
//   * sub-histograms are privatised in LDS as 64-bit FIXED-POINT words that carry the row count in their top bits (one ds_add_u64 per
//     row and feature; see "fixed-point accumulation" below), drained into registers, written out per chunk and summed over chunks by
//     a second kernel: counts are exact, and the sums are the correctly rounded totals of the once-rounded gradients -- independent
//     of the order of the atomics, i.e. bit-reproducible (the reference's fp64 sums depend on its thread count).
#include "hist_kernels.h"
#include <algorithm>

namespace gpb {

// ---- fixed-point accumulation -----------------------------------------------------------------------------------------------
// The fp64 sums of the reference are order-dependent (per-thread block buffers merged per thread count).  Here every gradient is
// rounded ONCE to a multiple of q = 2^(ex - 41), 2^ex >= max |g| over the rows handed to gpb_hip_hist_set_gradients (k = rint(g / q),
// |k| <= 2^41), the k are summed as INTEGERS, and the integer total is converted once: hist = fl(q * sum k).  The result does not
// depend on the order of the additions, the chunking or the number of ranks' rows per chunk: it is bit-reproducible, and it differs
// from the exact real sum by at most count * q / 2 + one rounding (q / 2 <= 2.3e-13 max |g| per row; the reference's own sequential
// fp64 sum carries count * 1.1e-16 * sum |g|).  Why: an LDS atomic on 64 bits costs one pass of the atomic unit whatever it adds, and
// ds_add_u64 runs at twice the rate of ds_add_f64 on gfx950 (scripts/ubench/lds_atomics.hip: 5.1 against 2.6 lane-updates per cycle
// per CU with random bins) -- and with integers the row COUNT rides in the same word: one atomic per (row, feature) instead of two.
//   word = count << 53 | (sum k  mod 2^53)      -- sum over at most 1792 rows between two flushes: |sum k| <= 1792 * 2^41 < 2^52
// Every 7 row-iterations (1792 rows of the workgroup) each thread drains "its" 16 words into 64-bit sums and 32-bit counts kept in
// registers; those are written per chunk and summed over chunks -- exactly -- by hist_reduce_kernel.
// Hessians (non-constant case) go to a second word without a count: |k_h| <= 2^51.
constexpr int kSumBits = 53;                                   // low bits of the packed word: the two's-complement sum
// drain of one LDS word between the two barriers of a flush: one returning atomic (a plain read + write of the word was measured equal within the
// run-to-run noise, 0.187 - 0.197 against 0.190 - 0.191 ms for the root pass at n = 1e7: profiles/r04_f_hist_flush_ab.txt)
__device__ __forceinline__ unsigned long long hist_drain_word(unsigned long long* w) {
  return __hip_atomic_exchange(w, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// rows between two flushes: at most 1792 <= 2^11 - 1 (count field), 1792 * 2^41 < 2^52 (sum field)
constexpr double kMagic = 6755399441055744.0;                  // 1.5 * 2^52: x + kMagic holds rint(x) in its mantissa for |x| < 2^51
constexpr unsigned long long kMagicBits = 0x4338000000000000ull;

// 2^(41 - ex) (gradients: HESS = false) or 2^(51 - ex) (hessians) for the largest |value| whose bits are *max_bits; 1 if all are zero
template <bool HESS>
__device__ inline double fixed_point_inv_q(const unsigned long long* max_bits) {
  const double mx = __longlong_as_double((long long)*max_bits);
  if (!(mx > 0.0)) return 1.0;
  int ex; (void)frexp(mx, &ex);                                // mx = f * 2^ex, f in [0.5, 1)
  ex = max(ex, -900);
  return ldexp(1.0, (HESS ? 51 : 41) - ex);
}
__device__ inline unsigned long long fixed_point_bits(double v, double inv_q) {      // rint(v * inv_q) as a 64-bit two's-complement integer
  return (unsigned long long)__double_as_longlong(v * inv_q + kMagic) - kMagicBits;
}

// The two children of a split as the tree grower's kernels see them: counts[0] = rows of THIS rank that went left, counts[1] = of all
// ranks; (begin, cnt) = the parent's segment of the row list, gcnt = its rows over all ranks.  BeforeFindBestSplit
// (serial_tree_learner.cpp:283-323): no search when both children hold fewer than 2 min_data_in_leaf rows; the smaller child is the left
// one iff left_cnt < right_cnt (global counts).
struct ChildSegment { int nl, gnl, smaller_is_left, smaller_begin, smaller_cnt, skip; };
__device__ inline ChildSegment child_segment(const int* counts, int begin, int cnt, int gcnt, int min_data_in_leaf) {
  ChildSegment c;
  c.nl = counts[0]; c.gnl = counts[1];
  const int gl = c.gnl, gr = gcnt - c.gnl;
  c.skip = (gr < 2 * min_data_in_leaf && gl < 2 * min_data_in_leaf) ? 1 : 0;
  c.smaller_is_left = gl < gr ? 1 : 0;
  c.smaller_begin = begin + (c.smaller_is_left ? 0 : c.nl);
  c.smaller_cnt = c.smaller_is_left ? c.nl : cnt - c.nl;
  return c;
}

// LDS layout: BIN-major, word(bin, f) = bin * 16 + f.  A 64-bit word covers one pair of the 64 banks, pair(bin, f) = (16 bin + f) mod 32
// = f + 16 (bin & 1): sixteen lanes that update sixteen DIFFERENT features can never meet in a bank pair, whatever their bins are --
// and the LDS works through a wavefront's 64-bit accesses 16 lanes at a time.  With lane l on feature (s + l) mod 16 at step s every
// such group of 16 consecutive lanes holds all 16 features: the atomics are conflict-free by construction (SQ_LDS_BANK_CONFLICT: 61 %
// of the LDS cycles with the feature-major [16][257] layout, profiles/r02_g_hist_*; none of a group's lanes share an address either).
template <bool HAS_HESS, bool HAS_IDX, int THREADS>
__global__ __launch_bounds__(THREADS) void hist_build_kernel(HistKernelArgs a) {
  constexpr int kOwn = GPB_HIST_MAX_BIN * GPB_HIST_FG / THREADS;               // words drained by one thread
  constexpr int kFlushIters = 1792 / THREADS;                                // rows between two flushes <= 1792 (see above)
  constexpr int kWords = GPB_HIST_MAX_BIN * GPB_HIST_FG;                     // 4096 words = 32 KB
  __shared__ unsigned long long s_acc[kWords];                               // count << 53 | sum
  __shared__ unsigned long long s_hacc[HAS_HESS ? kWords : 1];
  const int tid = threadIdx.x;
  // XCD-aware mapping: workgroups are dealt to the 8 XCDs round-robin by their linear id and every XCD has its own L2, so the
  // feature groups of ONE chunk (which read the same 64-byte row segments) must sit 8 ids apart to meet in the same L2:
  // id = 8 * (groups * (chunk / 8) + fg) + chunk % 8.  (With fg fastest -- the first version -- the four groups of a chunk ran on
  // four XCDs and every row segment was fetched from HBM four times: FETCH_SIZE 1.44 GB per launch against 0.72 GB of rows + gradients.)
  const int id = blockIdx.x, groups = a.fpad / GPB_HIST_FG;
  int fg, chunk;
  if ((a.nchunks & 7) == 0) { chunk = (id / (8 * groups)) * 8 + (id & 7); fg = (id >> 3) % groups; }
  else { fg = id % groups; chunk = id / groups; }
  for (int t = tid; t < kWords; t += THREADS) {
    s_acc[t] = 0ull;
    if constexpr (HAS_HESS) s_hacc[t] = 0ull;
  }
  const double inv_q = fixed_point_inv_q<false>(a.grad_max_bits);
  double inv_qh = 1.0;
  if constexpr (HAS_HESS) inv_qh = fixed_point_inv_q<true>(a.hess_max_bits);
  // thread t drains the words t, t + 256, ... (consecutive lanes, consecutive words): word w = (bin w / 16, feature w % 16)
  long long rk[kOwn], rh[HAS_HESS ? kOwn : 1];
  unsigned rc[kOwn];
#pragma unroll
  for (int i = 0; i < kOwn; ++i) { rk[i] = 0; rc[i] = 0u; if constexpr (HAS_HESS) rh[i] = 0; }
  auto flush = [&]() {          // one exchange per word: the old value comes back, zero goes in
    __syncthreads();
#pragma unroll
    for (int i = 0; i < kOwn; ++i) {
      const unsigned long long v = hist_drain_word(&s_acc[i * THREADS + tid]);
      const long long sum = (long long)(v << (64 - kSumBits)) >> (64 - kSumBits);
      rk[i] += sum;
      rc[i] += (unsigned)((v - (unsigned long long)sum) >> kSumBits);
      if constexpr (HAS_HESS) rh[i] += (long long)hist_drain_word(&s_hacc[i * THREADS + tid]);
    }
    __syncthreads();
  };
  __syncthreads();
  unsigned long long* fcol[GPB_HIST_FG];          // loop-invariant: the word of bin 0 of this lane's feature at step s
#pragma unroll
  for (int s = 0; s < GPB_HIST_FG; ++s) fcol[s] = s_acc + ((s + tid) & 15);
  const int nf = min(GPB_HIST_FG, a.num_features - fg * GPB_HIST_FG);   // real features of this group (the last group may be partial)
  // the rows: [0, num_data) of data_indices as given by the host, or -- tree grower -- the SMALLER child of the split whose left counts
  // the partition kernels have just left in device memory (no host round trip between the partition and this build)
  int num_data = a.num_data, rows_per_chunk = a.rows_per_chunk;
  if constexpr (HAS_IDX) {
    if (a.seg_counts) {
      const ChildSegment cs = child_segment(a.seg_counts, a.seg_begin, a.seg_cnt, a.seg_gcnt, a.seg_min_data_in_leaf);
      a.data_indices += cs.smaller_begin;
      num_data = cs.skip ? 0 : cs.smaller_cnt;
      rows_per_chunk = (num_data + a.nchunks - 1) / a.nchunks;
    }
  }
  const int r0 = chunk * rows_per_chunk;
  const int r1 = min(r0 + rows_per_chunk, num_data);
  const uint8_t* base = a.bins_rm + (size_t)fg * GPB_HIST_FG;
  // one lane = one row: a 16-byte load brings the row's 16 bins of this feature group, the gradient load is coalesced across the
  // wavefront (or gathered through data_indices for a leaf); the lane then issues one 64-bit LDS atomic per feature.
  // software pipeline: the loads of the next THREADS rows are in flight while this iteration's atomics run.  The prefetch is
  // UNCONDITIONAL (row index clamped to the chunk's last row) so that the compiler can wait with vmcnt(2) -- "everything but the two
  // loads just issued" -- instead of vmcnt(0): with a predicated prefetch every iteration waited for its own loads (SQ_WAIT_ANY 71 % of
  // the wave cycles, profiles/r02_h_hist_*).
  struct RowData { uint4 bv; double g, h; };
  auto fetch = [&](int r) -> RowData {
    RowData d;
    const int row = HAS_IDX ? a.data_indices[r] : r;
    d.bv = *reinterpret_cast<const uint4*>(base + (size_t)row * a.fpad);
    d.g = a.grad[row];
    d.h = 0.0;
    if constexpr (HAS_HESS) d.h = a.hess[row];
    return d;
  };
  auto accumulate = [&](const RowData& cur) {
    const uint4 bv = cur.bv;
    const unsigned long long add_g = fixed_point_bits(cur.g, inv_q) + (1ull << kSumBits);
    unsigned long long add_h = 0ull;
    if constexpr (HAS_HESS) add_h = fixed_point_bits(cur.h, inv_qh);
    if (nf < GPB_HIST_FG) {
      // last, partial feature group: only its nf real features are accumulated (the padding features all sit in bin 0 and nobody reads
      // them); lanes of a 16-lane group share features here, so some of these atomics do meet in a bank pair
      const unsigned long long lo = ((unsigned long long)bv.y << 32) | bv.x, hi = ((unsigned long long)bv.w << 32) | bv.z;
      int f = tid % nf;
      for (int s = 0; s < nf; ++s) {
        const int b = (int)(((f & 8) ? hi : lo) >> (8 * (f & 7))) & 0xff;
        atomicAdd(&s_acc[b * GPB_HIST_FG + f], add_g);
        if constexpr (HAS_HESS) atomicAdd(&s_hacc[b * GPB_HIST_FG + f], add_h);
        f = (f + 1 == nf) ? 0 : f + 1;
      }
    } else {
      // lane l handles feature (s + l) % 16 at step s.  The row's 16 bin bytes are rotated by l bytes ONCE (word rotation by l / 4,
      // then v_alignbyte by l % 4), so that step s reads byte s with a constant-offset bit-field extract.
      const unsigned wr = (unsigned)(tid >> 2) & 3u;
      unsigned w0 = bv.x, w1 = bv.y, w2 = bv.z, w3 = bv.w, t0, t1, t2, t3;
      t0 = (wr & 1u) ? w1 : w0; t1 = (wr & 1u) ? w2 : w1; t2 = (wr & 1u) ? w3 : w2; t3 = (wr & 1u) ? w0 : w3;
      w0 = (wr & 2u) ? t2 : t0; w1 = (wr & 2u) ? t3 : t1; w2 = (wr & 2u) ? t0 : t2; w3 = (wr & 2u) ? t1 : t3;
      const unsigned sh = (unsigned)(tid & 3);
      const unsigned rw[4] = {__builtin_amdgcn_alignbyte(w1, w0, sh), __builtin_amdgcn_alignbyte(w2, w1, sh),
                              __builtin_amdgcn_alignbyte(w3, w2, sh), __builtin_amdgcn_alignbyte(w0, w3, sh)};
#pragma unroll
      for (int s = 0; s < GPB_HIST_FG; ++s) {
        const unsigned b = (rw[s >> 2] >> (8 * (s & 3))) & 0xffu;
        unsigned long long* const p = fcol[s] + b * GPB_HIST_FG;
        atomicAdd(p, add_g);
        if constexpr (HAS_HESS) atomicAdd(p + (s_hacc - s_acc), add_h);
      }
    }
  };
  const int nrows = max(r1 - r0, 0), nfull = nrows / THREADS;
  if (nrows > 0) {
    int since_flush = 0;
    RowData cur = fetch(min(r0 + tid, r1 - 1));
    for (int it = 0; it < nfull; ++it) {
      const RowData nxt = fetch(min(r0 + (it + 1) * THREADS + tid, r1 - 1));
      accumulate(cur);
      cur = nxt;
      if (++since_flush == kFlushIters) { flush(); since_flush = 0; }
    }
    if (r0 + nfull * THREADS + tid < r1) accumulate(cur);       // the chunk's last, partial block of rows
  }
  flush();
  // partials: [chunk][feature group][bin][16 features] -- the layout of the LDS words, written as they were drained (coalesced)
  const size_t pbase = ((size_t)chunk * groups + fg) * kWords;
#pragma unroll
  for (int i = 0; i < kOwn; ++i) {
    a.part_grad[pbase + i * THREADS + tid] = rk[i];
    a.part_cnt[pbase + i * THREADS + tid] = rc[i];
    if constexpr (HAS_HESS) a.part_hess[pbase + i * THREADS + tid] = rh[i];
  }
}

// ---- whole rows per lane (constant hessian, four feature groups = 64 features per workgroup) -----------------------------------
// hist_build_kernel keeps three units near 35 % each at n = 1e7 (VALU issue, LDS atomics, L1 tag lookups: profiles/r02_i_hist_*), because
// every workgroup repeats the per-row work -- row index, gradient load and conversion, one 64-byte-line request per 16 features -- for
// its 16 features only.  Here a lane takes a row's whole 64 bytes (four feature groups) and issues its 64 atomics into four 32 KB
// sub-histogram blocks (128 KB of LDS: one workgroup of 512 lanes per CU): the per-row work is paid once per 64 features, what remains
// is the LDS atomic rate.  Same words, same drains (every 3 iterations = 1536 rows <= 1792), same partial layout as hist_build_kernel.
// Constant hessian only: per-row hessians stay on hist_build_kernel (the two-word whole-row form was slower).
// 16 bytes of a row from a 4-byte-aligned address (compact rows: stride F rounded up to 4): ONE global_load_dwordx4 -- the hardware needs dword alignment only
struct __attribute__((packed, aligned(4))) RowQuad { unsigned x, y, z, w; };
template <bool HAS_IDX, int NBK>   // NBK = feature groups of this launch's blocks that exist: NB, or fewer for the data set's last, partial block
// Round 6, measured at n = 1e7, F = 50 and NOT kept (profiles/r06_a_hist_*, r06_b_hist_*, r06_c_hist_*): two blocks of rows in flight ahead of the one being accumulated
// (0.193 against 0.183 ms); two feature groups per workgroup = two workgroups = 4 wavefronts per SIMD per CU (0.191 against 0.193 ms on the same box); the sixteen
// word addresses of a block computed first, into sixteen registers, and the sixteen ds_add_u64 issued back to back instead of the compiler's address / atomic
// pairs on one register (0.208 against 0.193 ms: a burst fills the LDS instruction queue and stalls the wavefront, interleaved VALU hides it).  What pays is the
// compact row copy (0.187 -> 0.183 ms, counter traffic 1.24x -> ~1.05x of the algorithmic bytes).  The pass stays bound by the rate at which ds_add_u64 instructions
// enter the LDS (PMC r05: SQ_ACTIVE_INST_LDS / SQ_INSTS_LDS = 13 cycles of a SIMD's path per wavefront-atomic; the LDS array itself is 53 % busy).
__global__ __launch_bounds__(512) void hist_build_rows_kernel(HistKernelArgs a) {
  // (1024 lanes: 128 VGPRs per lane are not enough for the 32 + 16 drain registers -> spills, 2x slower)
  constexpr int NB = 4;
  constexpr int THREADS = 512, kWords = GPB_HIST_MAX_BIN * GPB_HIST_FG, kOwn = NB * kWords / THREADS, kFlushIters = 1792 / THREADS;
  extern __shared__ unsigned long long s_rows[];                  // [NB][256 bins][16 features] (+ the same again for the hessian sums)
  const int tid = threadIdx.x;
  const int chunk = blockIdx.x, quad = blockIdx.y + a.quad0, groups = a.fpad / GPB_HIST_FG;
  for (int t = tid; t < NB * kWords; t += THREADS) s_rows[t] = 0ull;
  const double inv_q = fixed_point_inv_q<false>(a.grad_max_bits);
  long long rk[kOwn];
  unsigned rc[kOwn];
#pragma unroll
  for (int i = 0; i < kOwn; ++i) { rk[i] = 0; rc[i] = 0u; }
  auto flush = [&]() {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < kOwn; ++i) {
      const unsigned long long v = hist_drain_word(&s_rows[i * THREADS + tid]);
      const long long sum = (long long)(v << (64 - kSumBits)) >> (64 - kSumBits);
      rk[i] += sum;
      rc[i] += (unsigned)((v - (unsigned long long)sum) >> kSumBits);
    }
    __syncthreads();
  };
  __syncthreads();
  const int r0 = chunk * a.rows_per_chunk;
  const int r1 = min(r0 + a.rows_per_chunk, a.num_data);
  // streaming passes (no index list) read the COMPACT copy when there is one: rows of rstride bytes, 4-byte aligned; a block's 16-byte load may run up to 12
  // bytes into the next row (the last row has 16 bytes of slack behind it) -- those bytes belong to features >= num_features, which are never accumulated
  const bool compact = !HAS_IDX && a.bins_cm != nullptr;
  const uint8_t* base = (compact ? a.bins_cm : a.bins_rm) + (size_t)quad * NB * GPB_HIST_FG;
  const size_t rbytes = compact ? (size_t)a.rstride : (size_t)a.fpad;
  constexpr int nblk = NBK;                           // (compile-time: a run-time bound in the unrolled block loop cost 25 %)
  struct RowData { uint4 bv[NB]; double g; };
  auto fetch = [&](int r) -> RowData {
    RowData d;
    const int row = HAS_IDX ? a.data_indices[r] : r;
    const RowQuad* p = reinterpret_cast<const RowQuad*>(base + (size_t)row * rbytes);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      if (b < nblk) { const RowQuad q = p[b]; d.bv[b] = make_uint4(q.x, q.y, q.z, q.w); }
      else d.bv[b] = make_uint4(0u, 0u, 0u, 0u);      // never read past the row's fpad bytes
    }
    d.g = a.grad[row];
    return d;
  };
  const unsigned wr = (unsigned)(tid >> 2) & 3u, sh = (unsigned)(tid & 3), l15 = (unsigned)tid & 15u;
  // real features of the LAST block of this launch's workgroups (16 unless it is the data set's last, partial feature group)
  const int nf_last = min(GPB_HIST_FG, a.num_features - (quad * NB + nblk - 1) * GPB_HIST_FG);
  auto accumulate = [&](const RowData& cur) {
    const unsigned long long add_g = fixed_point_bits(cur.g, inv_q) + (1ull << kSumBits);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      if (b >= nblk) break;
      if (b == nblk - 1 && nf_last < GPB_HIST_FG) {
        // the data set's last, PARTIAL feature group (round 4): only its nf_last real features are accumulated -- the padding features all sit
        // in bin 0 and nobody reads them (hist_reduce_kernel stops at num_features); at F = 50 that is 2 atomics per row instead of 16 in this
        // block, 50 instead of 64 per row in all.  Lanes of a 16-lane group share features here, so some of these atomics meet in a bank pair.
        const unsigned wsel[4] = {cur.bv[b].x, cur.bv[b].y, cur.bv[b].z, cur.bv[b].w};
        int f = tid % nf_last;
        for (int s = 0; s < nf_last; ++s) {
          const unsigned wv = (f & 8) ? ((f & 4) ? wsel[3] : wsel[2]) : ((f & 4) ? wsel[1] : wsel[0]);
          const unsigned bin = (wv >> (8 * (f & 3))) & 0xffu;
          atomicAdd(&s_rows[b * kWords + bin * GPB_HIST_FG + f], add_g);
          f = (f + 1 == nf_last) ? 0 : f + 1;
        }
        continue;
      }
      unsigned w0 = cur.bv[b].x, w1 = cur.bv[b].y, w2 = cur.bv[b].z, w3 = cur.bv[b].w, t0, t1, t2, t3;
      t0 = (wr & 1u) ? w1 : w0; t1 = (wr & 1u) ? w2 : w1; t2 = (wr & 1u) ? w3 : w2; t3 = (wr & 1u) ? w0 : w3;
      w0 = (wr & 2u) ? t2 : t0; w1 = (wr & 2u) ? t3 : t1; w2 = (wr & 2u) ? t0 : t2; w3 = (wr & 2u) ? t1 : t3;
      const unsigned rw[4] = {__builtin_amdgcn_alignbyte(w1, w0, sh), __builtin_amdgcn_alignbyte(w2, w1, sh),
                              __builtin_amdgcn_alignbyte(w3, w2, sh), __builtin_amdgcn_alignbyte(w0, w3, sh)};
#pragma unroll
      for (int s = 0; s < GPB_HIST_FG; ++s) {
        const unsigned bin = (rw[s >> 2] >> (8 * (s & 3))) & 0xffu;
        atomicAdd(&s_rows[b * kWords + bin * GPB_HIST_FG + ((s + l15) & 15u)], add_g);
      }
    }
  };
  const int nrows = max(r1 - r0, 0), nfull = nrows / THREADS;
  if (nrows > 0) {
    int since_flush = 0;
    RowData cur = fetch(min(r0 + tid, r1 - 1));
    for (int it = 0; it < nfull; ++it) {
      const RowData nxt = fetch(min(r0 + (it + 1) * THREADS + tid, r1 - 1));
      accumulate(cur);
      cur = nxt;
      if (++since_flush == kFlushIters) { flush(); since_flush = 0; }
    }
    if (r0 + nfull * THREADS + tid < r1) accumulate(cur);
  }
  flush();
  // word w of block b -> partial (chunk, group 4 quad + b, word w): the layout of hist_build_kernel's partials
#pragma unroll
  for (int i = 0; i < kOwn; ++i) {
    const int w = i * THREADS + tid, b = w / kWords, ww = w - b * kWords;
    if (quad * NB + b < groups) {
      const size_t o = ((size_t)chunk * groups + quad * NB + b) * kWords + ww;
      a.part_grad[o] = rk[i];
      a.part_cnt[o] = rc[i];
    }
  }
}

// Sum of the chunk partials -- integers, so the total is exact and independent of the chunking.  A workgroup owns 64 consecutive words
// (4 bins x 16 features) of one feature group; its 16 slices of 64 lanes take the chunks ch = slice, slice + 16, ... (coalesced 512-byte
// segments, 4 loads in flight per lane).  A 64-bit partial is at most 2^41 * rows-per-chunk; the total over all chunks may pass 2^63,
// so it is carried in two limbs (sum of the high 32 bits, sum of the low 32 bits) and converted once: fl(hi * 2^32 + lo) is the
// correctly rounded integer total, and the scale q is a power of two.
struct Limbs { long long hi = 0; unsigned long long lo = 0; __device__ void add(long long p) { hi += p >> 32; lo += (unsigned long long)(unsigned)p; } };
// one histogram entry from its integer totals: q = 1 / inv_q exactly (powers of two); fl(hi * 2^32 + lo) is the correctly rounded integer
// total.  A non-finite gradient / hessian anywhere makes the sums NaN (the reference's sums would be non-finite in the bins of those rows;
// no tree can be grown from either)
template <bool HAS_HESS>
__device__ inline void hist_convert_entry(const Limbs& g, const Limbs& h, unsigned long long c, const unsigned long long* grad_max_bits,
                                          const unsigned long long* hess_max_bits, double const_hess, double* out2, unsigned long long* cnt_out) {
  const unsigned long long kInfBits = 0x7ff0000000000000ull;
  const double qg = 1.0 / fixed_point_inv_q<false>(grad_max_bits);
  const double tg = ((double)g.hi * 4294967296.0 + (double)g.lo) * qg;
  out2[0] = *grad_max_bits >= kInfBits ? __longlong_as_double(0x7ff8000000000000ll) : tg;
  if constexpr (HAS_HESS) {
    const double qh = 1.0 / fixed_point_inv_q<true>(hess_max_bits);
    const double th = ((double)h.hi * 4294967296.0 + (double)h.lo) * qh;
    out2[1] = *hess_max_bits >= kInfBits ? __longlong_as_double(0x7ff8000000000000ll) : th;
  } else out2[1] = (double)c * const_hess;
  if (cnt_out) *cnt_out = c;
}
template <bool HAS_HESS>
__global__ __launch_bounds__(1024) void hist_reduce_kernel(HistReduceArgs a) {
  __shared__ long long s_ghi[16][64], s_hhi[HAS_HESS ? 16 : 1][64];
  __shared__ unsigned long long s_glo[16][64], s_hlo[HAS_HESS ? 16 : 1][64], s_c[16][64];
  constexpr int kWords = GPB_HIST_MAX_BIN * GPB_HIST_FG;
  const int fg = blockIdx.x, l = threadIdx.x & 63, sl = threadIdx.x >> 6, groups = a.fpad / GPB_HIST_FG;
  const int w = blockIdx.y * 64 + l, b = w >> 4, f = fg * GPB_HIST_FG + (w & 15);
  Limbs g, h;
  unsigned long long c = 0;
  const size_t stride = (size_t)groups * kWords;
  const size_t p0 = (size_t)fg * kWords + w;
  int ch = sl;
  for (; ch + 48 < a.nchunks; ch += 64) {
    const size_t p = p0 + (size_t)ch * stride;
    const long long g0 = a.part_grad[p], g1 = a.part_grad[p + 16 * stride], g2 = a.part_grad[p + 32 * stride], g3 = a.part_grad[p + 48 * stride];
    const uint32_t c0 = a.part_cnt[p], c1 = a.part_cnt[p + 16 * stride], c2 = a.part_cnt[p + 32 * stride], c3 = a.part_cnt[p + 48 * stride];
    g.add(g0); g.add(g1); g.add(g2); g.add(g3);
    c += c0; c += c1; c += c2; c += c3;
    if constexpr (HAS_HESS) {
      const long long h0 = a.part_hess[p], h1 = a.part_hess[p + 16 * stride], h2 = a.part_hess[p + 32 * stride], h3 = a.part_hess[p + 48 * stride];
      h.add(h0); h.add(h1); h.add(h2); h.add(h3);
    }
  }
  for (; ch < a.nchunks; ch += 16) {
    const size_t p = p0 + (size_t)ch * stride;
    g.add(a.part_grad[p]);
    c += a.part_cnt[p];
    if constexpr (HAS_HESS) h.add(a.part_hess[p]);
  }
  s_ghi[sl][l] = g.hi; s_glo[sl][l] = g.lo; s_c[sl][l] = c;
  if constexpr (HAS_HESS) { s_hhi[sl][l] = h.hi; s_hlo[sl][l] = h.lo; }
  __syncthreads();
  if (sl != 0 || f >= a.num_features) return;
  if (b >= a.bin_offsets[f + 1] - a.bin_offsets[f]) return;
  g = Limbs(); h = Limbs(); c = 0;
  for (int k = 0; k < 16; ++k) {
    g.hi += s_ghi[k][l]; g.lo += s_glo[k][l]; c += s_c[k][l];
    if constexpr (HAS_HESS) { h.hi += s_hhi[k][l]; h.lo += s_hlo[k][l]; }
  }
  const size_t o = (size_t)a.bin_offsets[f] + b;
  if (a.limbs_out) {        // sharded handle: the integer totals leave as they are -- summed over the ranks as INTEGERS, converted once afterwards
    // word-major [5][limb_stride]: {grad hi, grad lo, count, hess hi, hess lo} -- the constant-hessian case (GPBoost's Gaussian likelihood) puts
    // only the first THREE words of every bin on the wire (hist_finish_sharded)
    long long* L = a.limbs_out + o;
    const size_t S = (size_t)a.limb_stride;
    L[0] = g.hi; L[S] = (long long)g.lo; L[2 * S] = (long long)c;
    if constexpr (HAS_HESS) { L[3 * S] = h.hi; L[4 * S] = (long long)h.lo; }
    return;
  }
  hist_convert_entry<HAS_HESS>(g, h, c, a.grad_max_bits, a.hess_max_bits, a.const_hess, a.hist_out + 2 * o, a.cnt_out ? a.cnt_out + o : nullptr);
}

// Sharded handles: limbs[5][total_bins] = {grad hi, grad lo, count, hess hi, hess lo} (word-major) summed over the ranks -> the histogram entries.  The
// integer total does not depend on how the rows were dealt to ranks, chunks or lanes, and it is converted by the same expression as on
// one GPU: the histogram of a sharded job is bit-identical to the one-GPU histogram of the same rows (given the same scale: the
// all-reduced max |g|, gpb_hip_hist_set_gradients).
__global__ __launch_bounds__(256) void hist_convert_kernel(const long long* __restrict__ limbs, int total_bins, const unsigned long long* grad_max_bits,
                                                          const unsigned long long* hess_max_bits, double const_hess, int has_hess,
                                                          double* __restrict__ hist_out, unsigned long long* __restrict__ cnt_out) {
  const int o = blockIdx.x * 256 + threadIdx.x;
  if (o >= total_bins) return;
  const long long* L = limbs + o;
  const size_t S = (size_t)total_bins;
  Limbs g, h;
  g.hi = L[0]; g.lo = (unsigned long long)L[S];
  if (has_hess) { h.hi = L[3 * S]; h.lo = (unsigned long long)L[4 * S]; }
  const unsigned long long c = (unsigned long long)L[2 * S];
  if (has_hess) hist_convert_entry<true>(g, h, c, grad_max_bits, hess_max_bits, const_hess, hist_out + 2 * (size_t)o, cnt_out ? cnt_out + o : nullptr);
  else hist_convert_entry<false>(g, h, c, grad_max_bits, hess_max_bits, const_hess, hist_out + 2 * (size_t)o, cnt_out ? cnt_out + o : nullptr);
}
// Feature-block exchange of the data-parallel tree grower (round 5; DataParallelTreeLearner's reduce-scatter by feature block + best-split sync,
// data_parallel_tree_learner.cpp:131, :155-173, :244, with the integer totals on the wire): hist_limbs_pack_kernel lays this rank's totals out block after
// block -- send[r][w][i] = limbs[w][blk_bin0[r] + i] for i < blk_bins[r], zero up to the common block length -- so that ONE reduce-scatter leaves on rank r
// the job-wide totals of ITS features' bins; hist_convert_block_kernel converts them (the same expression, the same bits as the all-reduce form) into the
// entries blk_bin0[r] .. of the flat histogram.
struct HistBlocks { int bin0[16]; int bins[16]; };
__global__ __launch_bounds__(256) void hist_limbs_pack_kernel(const long long* __restrict__ limbs, int total_bins, int nwords, HistBlocks b, int world, int maxblk,
                                                             long long* __restrict__ send) {
  const size_t per_rank = (size_t)nwords * maxblk, total = per_rank * world;
  for (size_t o = (size_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (size_t)gridDim.x * 256) {
    const int r = (int)(o / per_rank), w = (int)((o % per_rank) / maxblk), i = (int)(o % maxblk);
    send[o] = i < b.bins[r] ? limbs[(size_t)w * total_bins + b.bin0[r] + i] : 0LL;
  }
}
__global__ __launch_bounds__(256) void hist_convert_block_kernel(const long long* __restrict__ recv, int maxblk, int nbins, int bin0, const unsigned long long* grad_max_bits,
                                                                const unsigned long long* hess_max_bits, double const_hess, int has_hess, double* __restrict__ hist_out) {
  const int o = blockIdx.x * 256 + threadIdx.x;
  if (o >= nbins) return;
  const long long* L = recv + o;
  const size_t S = (size_t)maxblk;
  Limbs g, h;
  g.hi = L[0]; g.lo = (unsigned long long)L[S];
  if (has_hess) { h.hi = L[3 * S]; h.lo = (unsigned long long)L[4 * S]; }
  const unsigned long long c = (unsigned long long)L[2 * S];
  if (has_hess) hist_convert_entry<true>(g, h, c, grad_max_bits, hess_max_bits, const_hess, hist_out + 2 * ((size_t)bin0 + o), nullptr);
  else hist_convert_entry<false>(g, h, c, grad_max_bits, hess_max_bits, const_hess, hist_out + 2 * ((size_t)bin0 + o), nullptr);
}
hipError_t launch_hist_limbs_pack(const long long* limbs, int total_bins, int nwords, const int* blk_bin0, const int* blk_bins, int world, int maxblk, long long* send,
                                  hipStream_t st) {
  HistBlocks b;
  for (int r = 0; r < 16; ++r) { b.bin0[r] = r < world ? blk_bin0[r] : 0; b.bins[r] = r < world ? blk_bins[r] : 0; }
  const size_t total = (size_t)nwords * maxblk * world;
  hipLaunchKernelGGL(hist_limbs_pack_kernel, dim3((unsigned)std::max<size_t>(1, std::min<size_t>((total + 255) / 256, 1024))), dim3(256), 0, st, limbs, total_bins, nwords, b,
                     world, maxblk, send);
  return hipGetLastError();
}
hipError_t launch_hist_convert_block(const long long* recv, int maxblk, int nbins, int bin0, const unsigned long long* grad_max_bits, const unsigned long long* hess_max_bits,
                                     double const_hess, int has_hess, double* hist_out, hipStream_t st) {
  if (nbins <= 0) return hipSuccess;
  hipLaunchKernelGGL(hist_convert_block_kernel, dim3((nbins + 255) / 256), dim3(256), 0, st, recv, maxblk, nbins, bin0, grad_max_bits, hess_max_bits, const_hess, has_hess, hist_out);
  return hipGetLastError();
}

hipError_t launch_hist_convert(const long long* limbs, int total_bins, const unsigned long long* grad_max_bits, const unsigned long long* hess_max_bits,
                               double const_hess, int has_hess, double* hist_out, unsigned long long* cnt_out, hipStream_t st) {
  hipLaunchKernelGGL(hist_convert_kernel, dim3((total_bins + 255) / 256), dim3(256), 0, st, limbs, total_bins, grad_max_bits, hess_max_bits, const_hess,
                     has_hess, hist_out, cnt_out);
  return hipGetLastError();
}

// Root of a sharded tree: (sum of gradients, sum of hessians, rows) of ALL ranks from the all-reduced integer totals of feature 0's bins
// (every row has exactly one bin there): layout-independent like the histogram itself.  out3 = {sum_gradient, sum_hessian, rows}.
__global__ __launch_bounds__(64) void hist_root_sums_kernel(const long long* __restrict__ limbs, int total_bins, const int* __restrict__ bin_offsets,
                                                           const unsigned long long* grad_max_bits, const unsigned long long* hess_max_bits,
                                                           double const_hess, int has_hess, double* __restrict__ out3) {
  if (threadIdx.x != 0) return;
  Limbs g, h; unsigned long long c = 0;
  const size_t S = (size_t)total_bins;
  for (int o = bin_offsets[0]; o < bin_offsets[1]; ++o) {
    const long long* L = limbs + o;
    g.hi += L[0]; g.lo += (unsigned long long)L[S]; c += (unsigned long long)L[2 * S];
    if (has_hess) { h.hi += L[3 * S]; h.lo += (unsigned long long)L[4 * S]; }
  }
  double e[2];
  if (has_hess) hist_convert_entry<true>(g, h, c, grad_max_bits, hess_max_bits, const_hess, e, nullptr);
  else hist_convert_entry<false>(g, h, c, grad_max_bits, hess_max_bits, const_hess, e, nullptr);
  out3[0] = e[0]; out3[1] = e[1]; out3[2] = (double)c;
}
hipError_t launch_hist_root_sums(const long long* limbs, int total_bins, const int* bin_offsets, const unsigned long long* grad_max_bits,
                                 const unsigned long long* hess_max_bits, double const_hess, int has_hess, double* out3, hipStream_t st) {
  hipLaunchKernelGGL(hist_root_sums_kernel, dim3(1), dim3(64), 0, st, limbs, total_bins, bin_offsets, grad_max_bits, hess_max_bits, const_hess, has_hess, out3);
  return hipGetLastError();
}

// bits of max |v| over v[0..n) (atomicMax on the IEEE bit pattern: monotone for non-negative doubles, NaN compares above infinity);
// one global atomic per workgroup (same-address atomics serialise at ~10 ns each)
__global__ __launch_bounds__(256) void hist_absmax_kernel(const double* __restrict__ v, int n, unsigned long long* __restrict__ out_bits) {
  __shared__ unsigned long long s_m[4];
  unsigned long long m = 0ull;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v[i]) & 0x7fffffffffffffffull;
    m = b > m ? b : m;
  }
  for (int off = 32; off >= 1; off >>= 1) { const unsigned long long o = __shfl_xor(m, off, 64); m = o > m ? o : m; }
  if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < 4; ++k) m = s_m[k] > m ? s_m[k] : m;
    if (m) atomicMax(out_bits, m);
  }
}
hipError_t launch_hist_absmax(const double* v, int n, unsigned long long* out_bits, hipStream_t st) {
  hipError_t e = hipMemsetAsync(out_bits, 0, sizeof(unsigned long long), st);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(hist_absmax_kernel, dim3(std::max(1, std::min((n + 4095) / 4096, 1024))), dim3(256), 0, st, v, n, out_bits);
  return hipGetLastError();
}

// feature-major [F][n] -> row-major [n][fpad] (padding features read as bin 0 and are never reduced)
__global__ void bins_transpose_kernel(const uint8_t* __restrict__ fm, uint8_t* __restrict__ rm, int n, int F, int fpad, int ncols) {   // fpad = row stride in bytes, ncols = bytes of a row that exist
  __shared__ uint8_t tile[16][64 + 4];
  const int f0 = blockIdx.y * 16, i0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;   // 256 threads: 64 x 4
  for (int ff = ty; ff < 16; ff += 4) {
    const int f = f0 + ff, i = i0 + tx;
    tile[ff][tx] = (f < F && i < n) ? fm[(size_t)f * n + i] : 0;
  }
  __syncthreads();
  const int ff = threadIdx.x & 15, ii = threadIdx.x >> 4;    // 16 x 16
  for (int k = ii; k < 64; k += 16) {
    const int i = i0 + k;
    if (i < n && f0 + ff < ncols) rm[(size_t)i * fpad + f0 + ff] = tile[ff][k];
  }
}

// leaf id of every row from the resident row lists: position p belongs to the segment with the largest begin <= p; the segment's rows
// live in one of the two ping-pong buffers of the tree grower (seg_buf)
__global__ void hist_label_rows_kernel(const int* __restrict__ rows0, const int* __restrict__ rows1, int n, const int* __restrict__ seg_begin,
                                       const int* __restrict__ seg_leaf, const int* __restrict__ seg_buf, int nseg, int* __restrict__ out) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  int lo = 0, hi = nseg - 1;
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (seg_begin[mid] <= p) lo = mid; else hi = mid - 1; }
  const int* rows = (seg_buf && seg_buf[lo]) ? rows1 : rows0;
  out[rows[p]] = seg_leaf[lo];
}
hipError_t launch_hist_label_rows(const int* rows0, const int* rows1, int n, const int* seg_begin, const int* seg_leaf, const int* seg_buf, int nseg,
                                  int* out, hipStream_t st) {
  hipLaunchKernelGGL(hist_label_rows_kernel, dim3((n + 255) / 256), dim3(256), 0, st, rows0, rows1, n, seg_begin, seg_leaf, seg_buf, nseg, out);
  return hipGetLastError();
}

template <int THREADS>
static void launch_hist_build_t(const HistKernelArgs& a, hipStream_t st) {
  dim3 grid((a.fpad / GPB_HIST_FG) * a.nchunks), block(THREADS);
  const bool hh = a.hess != nullptr, hi = a.data_indices != nullptr;
  if (hh && hi) hipLaunchKernelGGL((hist_build_kernel<true, true, THREADS>), grid, block, 0, st, a);
  else if (hh) hipLaunchKernelGGL((hist_build_kernel<true, false, THREADS>), grid, block, 0, st, a);
  else if (hi) hipLaunchKernelGGL((hist_build_kernel<false, true, THREADS>), grid, block, 0, st, a);
  else hipLaunchKernelGGL((hist_build_kernel<false, false, THREADS>), grid, block, 0, st, a);
}
hipError_t launch_hist_build(const HistKernelArgs& a, hipStream_t st) {
  // at least four (with per-row hessians: two) feature groups, enough rows to fill the CUs one workgroup each: whole rows per lane
  if (a.use_rows_kernel) {
    constexpr int lds = 4 * GPB_HIST_MAX_BIN * GPB_HIST_FG * 8;
    // full quads (pairs) in one launch, the last partial one in a second launch; the dynamic-LDS attribute is set per launch
    // (it belongs to the current device)
    auto go = [&](auto kern, int nquads, int quad0) -> hipError_t {
      HistKernelArgs b = a;
      b.quad0 = quad0;
      const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      if (e != hipSuccess) return e;
      hipLaunchKernelGGL(kern, dim3(a.nchunks, nquads), dim3(512), lds, st, b);
      return hipGetLastError();
    };
    const int groups = a.fpad / GPB_HIST_FG, full = groups / 4, rest = groups % 4;
    hipError_t e = hipSuccess;
    if (full > 0) e = a.data_indices ? go(hist_build_rows_kernel<true, 4>, full, 0) : go(hist_build_rows_kernel<false, 4>, full, 0);
    if (e == hipSuccess && rest == 1) e = a.data_indices ? go(hist_build_rows_kernel<true, 1>, 1, full) : go(hist_build_rows_kernel<false, 1>, 1, full);
    if (e == hipSuccess && rest == 2) e = a.data_indices ? go(hist_build_rows_kernel<true, 2>, 1, full) : go(hist_build_rows_kernel<false, 2>, 1, full);
    if (e == hipSuccess && rest == 3) e = a.data_indices ? go(hist_build_rows_kernel<true, 3>, 1, full) : go(hist_build_rows_kernel<false, 3>, 1, full);
    return e;
  }
  // 256 threads: 512 and 1024 (twice / four times the wavefronts on the same 32 KB of LDS) time the same within 2 % at n = 1e7
  launch_hist_build_t<256>(a, st);
  return hipGetLastError();
}
hipError_t launch_hist_reduce(const HistReduceArgs& a, hipStream_t st) {
  const dim3 grid(a.fpad / GPB_HIST_FG, GPB_HIST_MAX_BIN * GPB_HIST_FG / 64);
  if (a.has_hess) hipLaunchKernelGGL(hist_reduce_kernel<true>, grid, dim3(1024), 0, st, a);
  else hipLaunchKernelGGL(hist_reduce_kernel<false>, grid, dim3(1024), 0, st, a);
  return hipGetLastError();
}
hipError_t launch_bins_transpose(const uint8_t* bins_fm, uint8_t* bins_rm, int n, int F, int fpad, hipStream_t st, int row_stride) {
  const int stride = row_stride > 0 ? row_stride : fpad;
  hipLaunchKernelGGL(bins_transpose_kernel, dim3((n + 63) / 64, fpad / 16), dim3(256), 0, st, bins_fm, bins_rm, n, F, stride, stride);
  return hipGetLastError();
}

// ---- row a12: Dataset::FixHistogram (src/LightGBM/io/dataset.cpp:1272-1290) and FeatureHistogram::Subtract
// (src/LightGBM/treelearner/feature_histogram.hpp:79-83) on device-resident histograms -------------------------------
// One lane per feature walks its bins in ascending order: the same subtraction order as the reference's loop, so the
// reconstructed most-frequent-bin entry is bit-identical given the same histogram.
__global__ void hist_fix_kernel(double* __restrict__ hist, int num_features, const int* __restrict__ view_offset,
                                const int* __restrict__ num_bin, const int* __restrict__ most_freq_bin, double sum_gradient,
                                double sum_hessian) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= num_features) return;
  const int mfb = most_freq_bin[f];
  if (mfb <= 0) return;
  double* v = hist + (size_t)view_offset[f] * 2;
  double g = sum_gradient, h = sum_hessian;
  const int nb = num_bin[f];
#pragma unroll 8
  for (int i = 0; i < nb; ++i) {                      // branch-free (x - 0.0 == x): eight steps' loads in flight
    const double vg = v[2 * i], vh = v[2 * i + 1];
    g -= (i != mfb) ? vg : 0.0; h -= (i != mfb) ? vh : 0.0;
  }
  v[2 * mfb] = g; v[2 * mfb + 1] = h;
}
__global__ void hist_subtract_kernel(const double* __restrict__ parent, const double* __restrict__ smaller, double* __restrict__ out, int len) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < len) out[i] = parent[i] - smaller[i];
}
hipError_t launch_hist_fix(double* hist, int num_features, const int* view_offset, const int* num_bin, const int* most_freq_bin,
                           double sum_gradient, double sum_hessian, hipStream_t st) {
  hipLaunchKernelGGL(hist_fix_kernel, dim3((num_features + 63) / 64), dim3(64), 0, st, hist, num_features, view_offset, num_bin,
                     most_freq_bin, sum_gradient, sum_hessian);
  return hipGetLastError();
}
hipError_t launch_hist_subtract(const double* parent, const double* smaller, double* out, int len, hipStream_t st) {
  hipLaunchKernelGGL(hist_subtract_kernel, dim3((len + 255) / 256), dim3(256), 0, st, parent, smaller, out, len);
  return hipGetLastError();
}

// ---- split search on a device-resident leaf histogram -----------------------------------------------
// FeatureHistogram::FindBestThreshold for numerical features on the default regularisation path (lambda_l1 = 0, max_delta_step = 0,
// path_smooth = 0, no monotone constraints, no extra_trees): src/LightGBM/treelearner/feature_histogram.hpp:85-95, :97-114, :163-207,
// :857-1084, :797-836, :741-763; the winner among features as SerialTreeLearner::ComputeBestSplitForFeature +
// SplitInfo::operator> pick it (serial_tree_learner.cpp:725-756, split_info.hpp:126-153).
//
// One workgroup per feature.  The reference's scan is sequential only in its running sums; everything else of a step (rounded
// counts, continue / break tests, the gain with its two divisions) depends on those sums alone.  So: (1) the feature's entries and
// their rounded counts (contraction off: `hess * cnt_factor + 0.5f` must round twice) go to LDS; (2) ONE lane per scan direction (two
// wavefronts, concurrently) accumulates the sums in the reference's order, branch-free, and stores them per step; (3) all lanes apply
// the reference's continue / break tests -- the scan ends at the first break in scan order; (4) all lanes evaluate the gains; (5) a
// reduction picks the first maximal gain in scan order (the reference updates on `>` only).  Every field of the result is
// bit-identical to the sequential walk given the same histogram.  (The first version walked the bins with one lane per feature:
// 115 us per call, 2/3 of a tree at config 3.)
namespace {
struct SplitOut { double gain, left_output, right_output, lsg, lsh, rsg, rsh; unsigned threshold; int left_count, right_count, default_left; };
constexpr int kSplitSteps = GPB_HIST_MAX_BIN + 2;      // step index t + 1 (the forward scan may start at t = -1)
}  // namespace

// The regularisation paths of the search (feature_histogram.hpp:137-161 picks the template instance; here run-time flags with the same
// expressions, so that with all three off the default path is reproduced bit for bit): lambda_l1 > 0 (USE_L1), max_delta_step > 0
// (USE_MAX_OUTPUT), path_smooth > kEpsilon (USE_SMOOTHING, parent_output = the leaf's own output).  :737-741 ThresholdL1,
// :743-765 CalculateSplittedLeafOutput, :826-857 GetLeafGain / GetLeafGainGivenOutput.
#pragma clang fp contract(off)
namespace {
struct RegPath { double l1, l2, mds, smooth, parent; bool use_l1, use_mds, use_smooth; };
__device__ __forceinline__ double reg_sign(double x) { return (double)((x > 0.0) - (x < 0.0)); }
__device__ __forceinline__ double reg_threshold_l1(double s, double l1) {
#pragma clang fp contract(off)
  const double reg_s = fmax(0.0, fabs(s) - l1);
  return reg_sign(s) * reg_s;
}
__device__ __forceinline__ double reg_leaf_output(double sg, double sh, const RegPath& r, int num_data) {
#pragma clang fp contract(off)
  double ret = r.use_l1 ? -reg_threshold_l1(sg, r.l1) / (sh + r.l2) : -sg / (sh + r.l2);
  if (r.use_mds && fabs(ret) > r.mds) ret = reg_sign(ret) * r.mds;
  if (r.use_smooth) {
    const double w = num_data / r.smooth;
    ret = ret * w / (w + 1) + r.parent / (w + 1);
  }
  return ret;
}
__device__ __forceinline__ double reg_gain_given_output(double sg, double sh, const RegPath& r, double output) {
#pragma clang fp contract(off)
  const double sg_l1 = r.use_l1 ? reg_threshold_l1(sg, r.l1) : sg;
  return -(2.0 * sg_l1 * output + (sh + r.l2) * output * output);
}
__device__ __forceinline__ double reg_leaf_gain(double sg, double sh, const RegPath& r, int num_data) {
#pragma clang fp contract(off)
  if (!r.use_mds && !r.use_smooth) {
    if (r.use_l1) { const double sg_l1 = reg_threshold_l1(sg, r.l1); return (sg_l1 * sg_l1) / (sh + r.l2); }
    return (sg * sg) / (sh + r.l2);
  }
  return reg_gain_given_output(sg, sh, r, reg_leaf_output(sg, sh, r, num_data));
}
}  // namespace

#pragma clang fp contract(off)
// the search for feature f by the 256 threads of a workgroup (all of them must call it; it ends with a barrier, so it can be called again)
__device__ void best_split_feature(const double* hist, int f, const int* __restrict__ view_offset,
                                   const int* __restrict__ num_bin, const int* __restrict__ meta3 /* offset, default_bin, missing */,
                                   double sum_gradient, double sum_hessian_leaf, int num_data, double lambda_l2, int min_data_in_leaf,
                                   double min_sum_hessian, double min_gain_to_split, SplitReg reg, double* __restrict__ out10,
                                   int* __restrict__ out_default_left, const double* data_in = nullptr /* the feature's view, if not hist + view_offset */) {
#pragma clang fp contract(off)
  __shared__ double s_x[GPB_HIST_MAX_BIN + 1][4];           // per entry: gradient sum, hessian sum, rounded count (as a double), pad
  __shared__ double s_o[2][kSplitSteps][4];                 // per direction and step: the three running sums
  __shared__ signed char s_eval[2][kSplitSteps];
  __shared__ double s_rg[256];
  __shared__ int s_rt[256], s_rs[256];
  __shared__ int s_brk[2];                                 // first break of the reverse (max t) / forward (min t) scan
  const int tid = threadIdx.x;
  const double kEps = (double)1e-15f;                      // include/LightGBM/meta.h:54
  const double* data = data_in ? data_in : hist + (size_t)view_offset[f] * 2;
  const int nb = num_bin[f], offset = meta3[3 * f], default_bin = meta3[3 * f + 1], missing = meta3[3 * f + 2];
  const double sum_hessian = sum_hessian_leaf + 2 * kEps;
  const RegPath rp{ reg.lambda_l1, lambda_l2, reg.max_delta_step, reg.path_smooth, reg.parent_output, reg.lambda_l1 > 0.0, reg.max_delta_step > 0.0,
                    reg.path_smooth > kEps };
  const double min_gain_shift = reg_leaf_gain(sum_gradient, sum_hessian, rp, num_data) + min_gain_to_split;      // BeforeNumercal :103-113
  const bool two_scans = nb > 2 && missing != 0;
  const bool skip_default = two_scans && missing == 1;
  const int na_as_missing = (two_scans && missing != 1) ? 1 : 0;
  const int nent = nb - offset;                            // entries of the feature's view
  for (int i = tid; i < 2 * kSplitSteps; i += 256) (&s_eval[0][0])[i] = 0;
  const double cnt_factor = num_data / sum_hessian;
  for (int i = tid; i < nent; i += 256) {
    // an entry the scans `continue` over before anything is accumulated (the default bin, zero-as-missing) contributes exact zeros
    const bool skipped = skip_default && (i + offset) == default_bin;
    const double g = data[2 * i], hh = data[2 * i + 1];
    s_x[i][0] = skipped ? 0.0 : g;
    s_x[i][1] = skipped ? 0.0 : hh;
    s_x[i][2] = skipped ? 0.0 : (double)(int)(hh * cnt_factor + 0.5f);      // Common::RoundInt, utils/common.h:920-922
  }
  if (tid < 2) s_brk[tid] = tid == 0 ? -2147483647 : 2147483647;
  __syncthreads();
  // (2) the running sums in the reference's order.  Lanes 0 / 1 / 2 of a wavefront carry the gradient sum, the hessian sum and the
  // count (integers below 2^31 are exact in fp64) of ONE scan direction: a step is one LDS read, one v_add_f64 and one LDS write for
  // all three (the first form -- one lane, three scalars, selects for the skipped bin -- spent ~150 cycles per step waiting for its
  // own LDS reads: 16 of the kernel's 23 us at 255 bins).  Reverse scan on wavefront 0, forward scan on wavefront 1, concurrently.
  const int r_hi = nb - 1 - offset - na_as_missing, r_lo = 1 - offset;        // reverse scan: t = r_hi .. r_lo  (:880-960)
  const int f_hi = nb - 2 - offset;                                           // forward scan: t = f_lo .. f_hi  (:962-1050)
  const bool fwd_pre = na_as_missing && offset == 1;
  const int f_lo = fwd_pre ? -1 : 0;
  const int q = tid & 63;
  if (tid < 3) {
    double acc = q == 1 ? kEps : 0.0;
#pragma unroll 8
    for (int t = r_hi; t >= r_lo; --t) { acc += s_x[t][q]; s_o[0][t + 1][q] = acc; }
  } else if (tid >= 64 && tid < 67 && two_scans) {
    double acc = q == 1 ? kEps : 0.0;
    if (fwd_pre) {                              // NaN-as-missing with the first bin outside the view: start from the leaf's totals (:985-1001)
      acc = q == 0 ? sum_gradient : (q == 1 ? sum_hessian - kEps : (double)num_data);
#pragma unroll 8
      for (int i = 0; i < nent; ++i) acc -= s_x[i][q];
      s_o[1][0][q] = acc;                       // step t = -1: nothing accumulated
    }
#pragma unroll 8
    for (int t = 0; t <= f_hi; ++t) { acc += s_x[t][q]; s_o[1][t + 1][q] = acc; }
  }
  __syncthreads();
  // (3) the reference's continue / break conditions of every step, in parallel; the scan stops at the FIRST break in scan order
  for (int dir = 0; dir < (two_scans ? 2 : 1); ++dir) {
    for (int k = tid; k < kSplitSteps; k += 256) {
      const int t = k - 1;
      const bool in_range = dir == 0 ? (t >= r_lo && t <= r_hi) : (t >= f_lo && t <= f_hi);
      if (!in_range) continue;
      if (skip_default && (t + offset) == default_bin) continue;            // `continue` before anything is accumulated
      const double ah = s_o[dir][k][1];
      const int ac = (int)s_o[dir][k][2];
      if (ac < min_data_in_leaf || ah < min_sum_hessian) continue;
      const int other_count = num_data - ac;
      const double other_h = sum_hessian - ah;
      if (other_count < min_data_in_leaf || other_h < min_sum_hessian) { if (dir == 0) atomicMax(&s_brk[0], t); else atomicMin(&s_brk[1], t); continue; }
      s_eval[dir][k] = 1;
    }
  }
  __syncthreads();
  for (int dir = 0; dir < (two_scans ? 2 : 1); ++dir)
    for (int k = tid; k < kSplitSteps; k += 256) {
      const int t = k - 1;
      if (s_eval[dir][k] && (dir == 0 ? t < s_brk[0] : t > s_brk[1])) s_eval[dir][k] = 0;      // after the break in scan order
    }
  __syncthreads();
  SplitOut o;
  o.gain = -INFINITY; o.left_output = 0.0; o.right_output = 0.0; o.lsg = 0.0; o.lsh = 0.0; o.rsg = 0.0; o.rsh = 0.0;
  o.threshold = 0; o.left_count = 0; o.right_count = 0; o.default_left = 1;
  bool splittable = false;
  for (int dir = 0; dir < (two_scans ? 2 : 1); ++dir) {
    // gains of the recorded steps, all lanes; position in scan order: reverse = descending t, forward = ascending t
    double gbest = -INFINITY; int tbest = -1; int any = 0;
    for (int k = tid; k < kSplitSteps; k += 256) {
      if (!s_eval[dir][k]) continue;
      double slg, slh, srg, srh;
      if (dir == 0) { srg = s_o[0][k][0]; srh = s_o[0][k][1]; slh = sum_hessian - srh; slg = sum_gradient - srg; }
      else { slg = s_o[1][k][0]; slh = s_o[1][k][1]; srh = sum_hessian - slh; srg = sum_gradient - slg; }
      const int ac = (int)s_o[dir][k][2];                  // rows on the accumulated side (right in the reverse scan, left in the forward one)
      const int lcnt = dir == 0 ? num_data - ac : ac;
      const double current_gain = reg_leaf_gain(slg, slh, rp, lcnt) + reg_leaf_gain(srg, srh, rp, num_data - lcnt);      // GetSplitGains :797-804
      if (current_gain <= min_gain_shift) continue;
      any = 1;
      const bool earlier = tbest < 0 || (dir == 0 ? k > tbest : k < tbest);
      if (current_gain > gbest || (current_gain == gbest && earlier)) { gbest = current_gain; tbest = k; }
    }
    s_rg[tid] = gbest; s_rt[tid] = tbest; s_rs[tid] = any;
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
      if (tid < w) {
        const double g2 = s_rg[tid + w]; const int t2 = s_rt[tid + w];
        s_rs[tid] |= s_rs[tid + w];
        if (t2 >= 0) {
          const int t1 = s_rt[tid];
          const bool earlier = t1 < 0 || (dir == 0 ? t2 > t1 : t2 < t1);
          if (g2 > s_rg[tid] || (g2 == s_rg[tid] && earlier)) { s_rg[tid] = g2; s_rt[tid] = t2; }
        }
      }
      __syncthreads();
    }
    if (tid == 0) {
      if (s_rs[0]) splittable = true;
      const double best_gain = s_rg[0];
      const int k = s_rt[0];
      if (splittable && k >= 0 && best_gain > o.gain + min_gain_shift) {
        const int t = k - 1;
        double best_slg, best_slh; int best_left_count;
        if (dir == 0) { best_left_count = num_data - (int)s_o[0][k][2]; best_slg = sum_gradient - s_o[0][k][0]; best_slh = sum_hessian - s_o[0][k][1]; o.threshold = (unsigned)(t - 1 + offset); }
        else { best_left_count = (int)s_o[1][k][2]; best_slg = s_o[1][k][0]; best_slh = s_o[1][k][1]; o.threshold = (unsigned)(t + offset); }
        o.left_output = reg_leaf_output(best_slg, best_slh, rp, best_left_count);
        o.left_count = best_left_count;
        o.lsg = best_slg; o.lsh = best_slh - kEps;
        o.right_output = reg_leaf_output(sum_gradient - best_slg, sum_hessian - best_slh, rp, num_data - best_left_count);
        o.right_count = num_data - best_left_count;
        o.rsg = sum_gradient - best_slg; o.rsh = sum_hessian - best_slh - kEps;
        o.gain = best_gain - min_gain_shift;
        o.default_left = dir == 0 ? 1 : 0;
      }
    }
    __syncthreads();
  }
  if (tid == 0) {
    if (!two_scans && missing == 2) o.default_left = 0;
    double* r = out10 + (size_t)f * 10;
    r[0] = o.gain; r[1] = (double)o.threshold; r[2] = o.left_count; r[3] = o.right_count; r[4] = o.left_output; r[5] = o.right_output;
    r[6] = o.lsg; r[7] = o.lsh; r[8] = o.rsg; r[9] = o.rsh;
    out_default_left[f] = o.default_left | (splittable ? 2 : 0);      // bit 1: FeatureHistogram::is_splittable() after the search
  }
  __syncthreads();
}

// ---- categorical features (round 5) -----------------------------------------------------------------------------------------
// FeatureHistogram::FindBestThresholdCategoricalInner (feature_histogram.hpp:278-519; no monotone constraints, no extra_trees) for feature f by the
// 256 threads of a workgroup (all must call it; ends with a barrier).  One-hot (num_bin <= max_cat_to_onehot, :317-370): every bin alone against the
// rest -- independent candidates, all lanes, first maximal gain in ascending bin order.  Otherwise (:371-470): the bins with at least cat_smooth rows
// are STABLE-sorted by sum_grad / (sum_hess + cat_smooth) -- a rank sort, every lane counts the keys before its own (equal keys keep their bin order,
// as std::stable_sort does) --, then one lane per direction walks at most max_cat_threshold of them from its end (the walk carries cnt_cur_group
// across candidates: sequential by construction, <= 32 steps by default).  l2 is raised by cat_l2 for gains and outputs, not for the gain of the
// unsplit leaf (:296-303).  Results: out10 as the numerical search (column 1: number of categories going left), flags (bit 1: splittable; bit 0,
// default_left, is false for categorical features, :284), and the bitset over the feature's BINS of the categories going left (8 words).
#pragma clang fp contract(off)
__device__ void best_split_feature_cat(const double* hist, int f, const int* __restrict__ view_offset, const int* __restrict__ num_bin,
                                       const int* __restrict__ meta3, double sum_gradient, double sum_hessian_leaf, int num_data, double lambda_l2,
                                       int min_data_in_leaf, double min_sum_hessian, double min_gain_to_split, SplitReg reg, CatCfg cat,
                                       double* __restrict__ out10, int* __restrict__ out_flags, unsigned* __restrict__ out_cat,
                                       const double* data_in = nullptr) {
#pragma clang fp contract(off)
  __shared__ double s_cg[GPB_HIST_MAX_BIN], s_ch[GPB_HIST_MAX_BIN], s_key[GPB_HIST_MAX_BIN];
  __shared__ int s_cc[GPB_HIST_MAX_BIN], s_sorted[GPB_HIST_MAX_BIN], s_sel[GPB_HIST_MAX_BIN];
  __shared__ double s_rgain[256];
  __shared__ int s_rt2[256], s_any[256];
  __shared__ double s_dir[2][4];       // per direction: best gain, best left gradient sum, best left hessian sum, (unused)
  __shared__ int s_diri[2][3];         // per direction: best threshold (index into the walk), best left count, splittable
  __shared__ int s_used;
  const int tid = threadIdx.x;
  const double kEps = (double)1e-15f;
  const double* data = data_in ? data_in : hist + (size_t)view_offset[f] * 2;
  const int nb = num_bin[f], offset = meta3[3 * f];
  const double sum_hessian = sum_hessian_leaf + 2 * kEps;
  const RegPath rp0{ reg.lambda_l1, lambda_l2, reg.max_delta_step, reg.path_smooth, reg.parent_output, reg.lambda_l1 > 0.0, reg.max_delta_step > 0.0,
                     reg.path_smooth > kEps };
  const double gain_shift = rp0.use_smooth ? reg_gain_given_output(sum_gradient, sum_hessian, rp0, reg.parent_output)
                                           : reg_leaf_gain(sum_gradient, sum_hessian, rp0, num_data);
  const double min_gain_shift = gain_shift + min_gain_to_split;
  const int bin_start = 1 - offset, bin_end = nb - offset;
  const bool use_onehot = nb <= cat.max_cat_to_onehot;
  const double cnt_factor = num_data / sum_hessian;
  if (tid == 0) s_used = 0;
  for (int t = tid; t < GPB_HIST_MAX_BIN; t += 256) {
    const bool in = t >= bin_start && t < bin_end;
    const double g = in ? data[2 * t] : 0.0, hh = in ? data[2 * t + 1] : 0.0;
    s_cg[t] = g; s_ch[t] = hh;
    s_cc[t] = in ? (int)(hh * cnt_factor + 0.5f) : 0;
    s_sel[t] = 0;
  }
  __syncthreads();
  RegPath rp = rp0;
  double best_gain = -INFINITY, best_slg = 0.0, best_slh = 0.0;
  int best_threshold = -1, best_left_count = 0, best_dir = 1, used_bin = -1;
  bool splittable = false;
  if (use_onehot) {
    double gbest = -INFINITY; int tbest = -1, any = 0;
    for (int t = bin_start + tid; t < bin_end; t += 256) {
      const double grad = s_cg[t], hess = s_ch[t];
      const int cnt = s_cc[t];
      if (cnt < min_data_in_leaf || hess < min_sum_hessian) continue;
      const int other_count = num_data - cnt;
      if (other_count < min_data_in_leaf) continue;
      const double sum_other_hessian = sum_hessian - hess - kEps;
      if (sum_other_hessian < min_sum_hessian) continue;
      const double sum_other_gradient = sum_gradient - grad;
      const double current_gain = reg_leaf_gain(sum_other_gradient, sum_other_hessian, rp, other_count) + reg_leaf_gain(grad, hess + kEps, rp, cnt);
      if (current_gain <= min_gain_shift) continue;
      any = 1;
      if (current_gain > gbest) { gbest = current_gain; tbest = t; }      // (ascending t within a lane: strict > keeps the first)
    }
    s_rgain[tid] = gbest; s_rt2[tid] = tbest; s_any[tid] = any;
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
      if (tid < w) {
        s_any[tid] |= s_any[tid + w];
        const double g2 = s_rgain[tid + w]; const int t2 = s_rt2[tid + w];
        if (t2 >= 0) {
          const int t1 = s_rt2[tid];
          if (t1 < 0 || g2 > s_rgain[tid] || (g2 == s_rgain[tid] && t2 < t1)) { s_rgain[tid] = g2; s_rt2[tid] = t2; }
        }
      }
      __syncthreads();
    }
    splittable = s_any[0] != 0;
    if (splittable) {
      best_threshold = s_rt2[0]; best_gain = s_rgain[0];
      best_slg = s_cg[best_threshold]; best_slh = s_ch[best_threshold] + kEps; best_left_count = s_cc[best_threshold];
    }
  } else {
    // the bins that take part (:372-377) and their keys
    for (int t = bin_start + tid; t < bin_end; t += 256) {
      const bool sel = (double)s_cc[t] >= cat.cat_smooth;
      s_sel[t] = sel ? 1 : 0;
      s_key[t] = s_cg[t] / (s_ch[t] + cat.cat_smooth);
      if (sel) atomicAdd(&s_used, 1);
    }
    __syncthreads();
    used_bin = s_used;
    for (int t = bin_start + tid; t < bin_end; t += 256) {
      if (!s_sel[t]) continue;
      const double k = s_key[t];
      int rank = 0;
      for (int j = bin_start; j < bin_end; ++j) {
        if (!s_sel[j]) continue;
        const double kj = s_key[j];
        rank += (kj < k || (!(k < kj) && j < t)) ? 1 : 0;       // stable order under the comparator `key_i < key_j`
      }
      s_sorted[rank] = t;
    }
    __syncthreads();
    rp.l2 += cat.cat_l2;
    const int max_num_cat = min(cat.max_cat_threshold, (used_bin + 1) / 2);
    if (tid == 0 || tid == 64) {
      const int d = tid == 0 ? 0 : 1, dir = d == 0 ? 1 : -1;
      int start_pos = d == 0 ? 0 : used_bin - 1;
      int cnt_cur_group = 0, left_count = 0;
      double slg = 0.0, slh = kEps;
      double bg = -INFINITY, bslg = 0.0, bslh = 0.0; int bt = -1, blc = 0, spl = 0;
      for (int i = 0; i < used_bin && i < max_num_cat; ++i) {
        const int t = s_sorted[start_pos];
        start_pos += dir;
        const double grad = s_cg[t], hess = s_ch[t];
        const int cnt = s_cc[t];
        slg += grad; slh += hess; left_count += cnt; cnt_cur_group += cnt;
        if (left_count < min_data_in_leaf || slh < min_sum_hessian) continue;
        const int right_count = num_data - left_count;
        if (right_count < min_data_in_leaf || right_count < cat.min_data_per_group) break;
        const double srh = sum_hessian - slh;
        if (srh < min_sum_hessian) break;
        if (cnt_cur_group < cat.min_data_per_group) continue;
        cnt_cur_group = 0;
        const double srg = sum_gradient - slg;
        const double current_gain = reg_leaf_gain(slg, slh, rp, left_count) + reg_leaf_gain(srg, srh, rp, right_count);
        if (current_gain <= min_gain_shift) continue;
        spl = 1;
        if (current_gain > bg) { blc = left_count; bslg = slg; bslh = slh; bt = i; bg = current_gain; }
      }
      s_dir[d][0] = bg; s_dir[d][1] = bslg; s_dir[d][2] = bslh;
      s_diri[d][0] = bt; s_diri[d][1] = blc; s_diri[d][2] = spl;
    }
    __syncthreads();
    splittable = (s_diri[0][2] | s_diri[1][2]) != 0;
    // dir = +1 first; dir = -1 replaces it on a strictly larger gain only (the reference's single running best over both walks)
    if (s_diri[0][0] >= 0) { best_gain = s_dir[0][0]; best_slg = s_dir[0][1]; best_slh = s_dir[0][2]; best_threshold = s_diri[0][0]; best_left_count = s_diri[0][1]; best_dir = 1; }
    if (s_diri[1][0] >= 0 && s_dir[1][0] > best_gain) { best_gain = s_dir[1][0]; best_slg = s_dir[1][1]; best_slh = s_dir[1][2]; best_threshold = s_diri[1][0]; best_left_count = s_diri[1][1]; best_dir = -1; }
  }
  if (tid == 0) {
    double* r = out10 + (size_t)f * 10;
    unsigned bits[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    if (splittable) {
      r[4] = reg_leaf_output(best_slg, best_slh, rp, best_left_count);
      r[2] = best_left_count; r[6] = best_slg; r[7] = best_slh - kEps;
      r[5] = reg_leaf_output(sum_gradient - best_slg, sum_hessian - best_slh, rp, num_data - best_left_count);
      r[3] = num_data - best_left_count; r[8] = sum_gradient - best_slg; r[9] = sum_hessian - best_slh - kEps;
      r[0] = best_gain - min_gain_shift;
      int ncat;
      if (use_onehot) { ncat = 1; const int b = best_threshold + offset; bits[b >> 5] |= 1u << (b & 31); }
      else {
        ncat = best_threshold + 1;
        for (int i = 0; i < ncat; ++i) { const int b = (best_dir == 1 ? s_sorted[i] : s_sorted[used_bin - 1 - i]) + offset; bits[b >> 5] |= 1u << (b & 31); }
      }
      r[1] = (double)ncat;
    } else {
      r[0] = -INFINITY;
      for (int k = 1; k < 10; ++k) r[k] = 0.0;
    }
    out_flags[f] = splittable ? 2 : 0;
    if (out_cat) for (int k = 0; k < 8; ++k) out_cat[(size_t)f * 8 + k] = bits[k];
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void hist_best_split_kernel(const double* __restrict__ hist, int num_features, const int* __restrict__ view_offset,
                                       const int* __restrict__ num_bin, const int* __restrict__ meta3, double sum_gradient, double sum_hessian_leaf,
                                       int num_data, double lambda_l2, int min_data_in_leaf, double min_sum_hessian, double min_gain_to_split,
                                       SplitReg reg, double* __restrict__ out10, int* __restrict__ out_default_left,
                                       const signed char* __restrict__ is_cat, CatCfg cat, unsigned* __restrict__ out_cat) {
  if ((int)blockIdx.x >= num_features) return;
  if (is_cat && is_cat[blockIdx.x]) {
    best_split_feature_cat(hist, blockIdx.x, view_offset, num_bin, meta3, sum_gradient, sum_hessian_leaf, num_data, lambda_l2, min_data_in_leaf,
                           min_sum_hessian, min_gain_to_split, reg, cat, out10, out_default_left, out_cat);
    return;
  }
  best_split_feature(hist, blockIdx.x, view_offset, num_bin, meta3, sum_gradient, sum_hessian_leaf, num_data, lambda_l2, min_data_in_leaf,
                     min_sum_hessian, min_gain_to_split, reg, out10, out_default_left);
}

// Both children of a split in ONE launch (tree grower): workgroup (f, child) handles feature f of the smaller (child = 0) or the larger
// (child = 1) leaf, for the feature's own histogram entries only:
//   child 0: FixHistogram of the freshly built smaller child (dataset.cpp:1272-1290), written to its slot; threshold search -> result set 0
//   child 1: larger = parent - fixed smaller in the parent's slot (serial_tree_learner.cpp:419-421); threshold search -> result set 1.
//            It derives the fixed most-frequent-bin entry of the smaller child itself (same loop, same order) instead of waiting for
//            workgroup (f, 0) to store it.
// Which child is the smaller one -- and whether the pair is searched at all -- is read from the partition's device-resident counts.
__device__ __forceinline__ void children_search_body(const ChildrenSearchArgs& a) {
#pragma clang fp contract(off)
  const int f = blockIdx.x, child = blockIdx.y, tid = threadIdx.x;
  if (f >= a.num_features) return;
  if (f < a.own_f0 || f >= a.own_f1) return;        // feature-block exchange: another rank owns this feature's bins (and searches it)
  const ChildSegment cs = child_segment(a.counts, 0, 0, a.gcnt, a.min_data_in_leaf);
  if (cs.skip) return;
  const double sg_s = cs.smaller_is_left ? a.left_sum_gradient : a.right_sum_gradient, sh_s = cs.smaller_is_left ? a.left_sum_hessian : a.right_sum_hessian;
  const double sg_l = cs.smaller_is_left ? a.right_sum_gradient : a.left_sum_gradient, sh_l = cs.smaller_is_left ? a.right_sum_hessian : a.left_sum_hessian;
  const int n_s = cs.smaller_is_left ? cs.gnl : a.gcnt - cs.gnl, n_l = a.gcnt - n_s;
  // parent_output of a child's search = the child's own output (LeafSplits::weight, serial_tree_learner.cpp:766-768)
  const SplitReg reg_s{ a.lambda_l1, a.max_delta_step, a.path_smooth, cs.smaller_is_left ? a.left_output : a.right_output };
  const SplitReg reg_l{ a.lambda_l1, a.max_delta_step, a.path_smooth, cs.smaller_is_left ? a.right_output : a.left_output };
  const int mfb = a.most_freq_bin[f];
  __shared__ double s_fix[2];
  // the smaller child's entries of THIS feature, local: from the build's chunk partials (summed here, see ChildrenSearchArgs) or from its slot
  __shared__ double s_sm[2 * (GPB_HIST_MAX_BIN + 1)];
  const int bo = a.bin_offsets[f], nbin_f = a.bin_offsets[f + 1] - bo;
  if (a.part_grad) {
    constexpr int kWords = GPB_HIST_MAX_BIN * GPB_HIST_FG;
    const size_t stride = (size_t)(a.fpad / GPB_HIST_FG) * kWords;
    for (int b = tid; b < nbin_f; b += 256) {
      Limbs g, h;
      unsigned long long c = 0;
      size_t p = (size_t)(f / GPB_HIST_FG) * kWords + (size_t)b * GPB_HIST_FG + (f % GPB_HIST_FG);
      for (int ch = 0; ch < a.nchunks; ++ch, p += stride) {
        g.add(a.part_grad[p]); c += a.part_cnt[p];
        if (a.has_hess) h.add(a.part_hess[p]);
      }
      if (a.has_hess) hist_convert_entry<true>(g, h, c, a.grad_max_bits, a.hess_max_bits, a.const_hess, s_sm + 2 * b, nullptr);
      else hist_convert_entry<false>(g, h, c, a.grad_max_bits, a.hess_max_bits, a.const_hess, s_sm + 2 * b, nullptr);
      if (child == 0) { a.smaller[2 * ((size_t)bo + b)] = s_sm[2 * b]; a.smaller[2 * ((size_t)bo + b) + 1] = s_sm[2 * b + 1]; }
    }
  } else {
    for (int e = tid; e < 2 * nbin_f; e += 256) s_sm[e] = a.smaller[2 * (size_t)bo + e];
  }
  __syncthreads();
  // the feature's view (FeatureHistogram::data_) inside its own entries: num_bin entries from view_offset when most_freq_bin > 0, one fewer
  // otherwise (offset = 1).  The tree grower hands over chunk partials only when this holds for every feature; a view that reaches outside is read
  // from the slot as before
  const int vrel = 2 * (a.view_offset[f] - bo);
  const bool local_view = a.part_grad != nullptr || (vrel >= 0 && vrel / 2 + a.num_bin[f] - (mfb == 0 ? 1 : 0) <= nbin_f);
  if (tid == 0 && mfb > 0) {                    // the same subtraction order as the reference's loop
    const double* v = local_view ? s_sm + vrel : a.smaller + (size_t)a.view_offset[f] * 2;
    double g = sg_s, hh = sh_s;
    const int nb = a.num_bin[f];
    // branch-free so that the reads of eight steps are in flight while the two subtraction chains run: x - 0.0 == x, the skipped entry changes nothing
#pragma unroll 8
    for (int i = 0; i < nb; ++i) {
      const double vg = v[2 * i], vh = v[2 * i + 1];
      g -= (i != mfb) ? vg : 0.0; hh -= (i != mfb) ? vh : 0.0;
    }
    s_fix[0] = g; s_fix[1] = hh;
  }
  __syncthreads();
  const size_t fix_at = mfb > 0 ? 2 * ((size_t)a.view_offset[f] + mfb) : (size_t)-1;     // index of the fixed entry in the flat histogram
  if (child == 0) {
    if (tid == 0 && mfb > 0) {
      a.smaller[fix_at] = s_fix[0]; a.smaller[fix_at + 1] = s_fix[1];
      if (local_view) { s_sm[vrel + 2 * mfb] = s_fix[0]; s_sm[vrel + 2 * mfb + 1] = s_fix[1]; }
    }
    __threadfence_block();
    __syncthreads();
    if (a.is_cat && a.is_cat[f])
      best_split_feature_cat(a.smaller, f, a.view_offset, a.num_bin, a.meta3, sg_s, sh_s, n_s, a.lambda_l2, a.min_data_in_leaf, a.min_sum_hessian,
                             a.min_gain_to_split, reg_s, a.cat, a.out10, a.out_flags, a.out_cat, local_view ? s_sm + vrel : nullptr);
    else
    best_split_feature(a.smaller, f, a.view_offset, a.num_bin, a.meta3, sg_s, sh_s, n_s, a.lambda_l2, a.min_data_in_leaf, a.min_sum_hessian,
                       a.min_gain_to_split, reg_s, a.out10, a.out_flags, local_view ? s_sm + vrel : nullptr);
  } else {
    for (size_t i = 2 * (size_t)bo + tid; i < 2 * (size_t)a.bin_offsets[f + 1]; i += 256) {
      const double sm = (i == fix_at || i == fix_at + 1) ? s_fix[i - fix_at] : s_sm[i - 2 * (size_t)bo];
      a.parent[i] = a.parent[i] - sm;
    }
    __threadfence_block();
    __syncthreads();
    if (a.is_cat && a.is_cat[f])
      best_split_feature_cat(a.parent, f, a.view_offset, a.num_bin, a.meta3, sg_l, sh_l, n_l, a.lambda_l2, a.min_data_in_leaf, a.min_sum_hessian,
                             a.min_gain_to_split, reg_l, a.cat, a.out10 + (size_t)a.num_features * 10, a.out_flags + (a.num_features + 1),
                             a.out_cat ? a.out_cat + (size_t)a.num_features * 8 : nullptr);
    else
    best_split_feature(a.parent, f, a.view_offset, a.num_bin, a.meta3, sg_l, sh_l, n_l, a.lambda_l2, a.min_data_in_leaf, a.min_sum_hessian,
                       a.min_gain_to_split, reg_l, a.out10 + (size_t)a.num_features * 10, a.out_flags + (a.num_features + 1));
  }
}
// host_seq != nullptr: the LAST workgroup to finish (a device ticket) tells the host, which polls that word of pinned memory instead of
// synchronising the stream: every workgroup's results are fenced to system scope before its ticket, the flag is written after the last one
__global__ __launch_bounds__(256) void hist_children_search_kernel(ChildrenSearchArgs a) {
  children_search_body(a);
  if (a.host_seq == nullptr) return;
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence_system();
    const unsigned total = gridDim.x * gridDim.y;
    if (atomicAdd(a.ticket, 1u) == total - 1u) {
      *a.ticket = 0u;                                   // (the next launch is behind this one in the stream)
      __threadfence_system();
      __hip_atomic_store(a.host_seq, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}
hipError_t launch_hist_children_search(const ChildrenSearchArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(hist_children_search_kernel, dim3(a.num_features, 2), dim3(256), 0, st, a);
  return hipGetLastError();
}

// the winner: larger gain, equal gains -> smaller feature index; features masked out by is_feature_used never win
__global__ void hist_pick_split_kernel(const double* __restrict__ out10, int num_features, const signed char* __restrict__ is_feature_used,
                                       int* __restrict__ best_feature) {
  // one wavefront: lanes stride over the features, then a shuffle reduction with SplitInfo::operator>'s order
  const int lane = threadIdx.x;
  int best_f = 2147483647;
  double best_gain = -INFINITY;
  for (int f = lane; f < num_features; f += 64) {
    if (is_feature_used && !is_feature_used[f]) continue;
    const double g = out10[(size_t)f * 10];
    if (g != best_gain ? g > best_gain : f < best_f) { best_gain = g; best_f = f; }
  }
  for (int off = 32; off >= 1; off >>= 1) {
    const double g2 = __shfl_xor(best_gain, off, 64);
    const int f2 = __shfl_xor(best_f, off, 64);
    if (g2 != best_gain ? g2 > best_gain : f2 < best_f) { best_gain = g2; best_f = f2; }
  }
  if (lane == 0) *best_feature = best_f == 2147483647 ? -1 : best_f;
}

hipError_t launch_hist_best_split(const double* hist, int num_features, const int* view_offset, const int* num_bin, const int* meta3,
                                  double sum_gradient, double sum_hessian, int num_data, double lambda_l2, int min_data_in_leaf,
                                  double min_sum_hessian, double min_gain_to_split, SplitReg reg, const signed char* is_feature_used, double* out10,
                                  int* out_default_left, int* best_feature, hipStream_t st, const signed char* is_cat, CatCfg cat, unsigned* out_cat) {
  hipLaunchKernelGGL(hist_best_split_kernel, dim3(num_features), dim3(256), 0, st, hist, num_features, view_offset, num_bin, meta3,
                     sum_gradient, sum_hessian, num_data, lambda_l2, min_data_in_leaf, min_sum_hessian, min_gain_to_split, reg, out10,
                     out_default_left, is_cat, cat, out_cat);
  if (best_feature)      // the tree grower picks on the host from the per-feature candidates it needs anyway
    hipLaunchKernelGGL(hist_pick_split_kernel, dim3(1), dim3(64), 0, st, (const double*)out10, num_features, is_feature_used, best_feature);
  return hipGetLastError();
}

// ---- partition of a leaf's rows by a numerical split (second half of SURVEY.md 8f rank 2) ---------------------------------
// DataPartition::Split -> Dataset::Split -> DenseBin::Split / SplitInner (src/LightGBM/io/dense_bin.hpp:176-307, single-feature
// group: min_bin = 1, USE_MIN_BIN = false).  A stable partition: both sides keep the order of data_indices, as the reference's
// per-thread blocks do when they are concatenated.  Three small kernels: per-block count of rows going left, exclusive scan of
// the block counts, classify again + scatter (1024 rows per block, 4 consecutive rows per lane).
namespace {
// (categorical, round 5: DenseBin::SplitCategoricalInner<USE_MIN_BIN = false>, dense_bin.hpp:305-342 -- stored bin 0 = the most frequent bin goes
//  with default_goes_left = its own bit; every other stored bin b stands for the feature's bin b - 1 + cat_offset, looked up in the 256-bit set)
struct SplitRule { int max_bin, t_zero_bin, th, miss_zero, miss_na, mfb_zero, mfb_na, default_goes_left, missing_goes_left;
                   int is_cat, cat_offset; unsigned b0, b1, b2, b3, b4, b5, b6, b7; };
__device__ __forceinline__ bool goes_left(const SplitRule& r, int bin) {
  if (r.is_cat) {
    if (bin == 0) return r.default_goes_left != 0;
    const int b = bin - 1 + r.cat_offset, w = b >> 5;
    const unsigned word = w == 0 ? r.b0 : w == 1 ? r.b1 : w == 2 ? r.b2 : w == 3 ? r.b3 : w == 4 ? r.b4 : w == 5 ? r.b5 : w == 6 ? r.b6 : w == 7 ? r.b7 : 0u;
    return ((word >> (b & 31)) & 1u) != 0u;
  }
  if (1 < r.max_bin) {
    if ((r.miss_zero && !r.mfb_zero && bin == r.t_zero_bin) || (r.miss_na && !r.mfb_na && bin == r.max_bin)) return r.missing_goes_left;
    if (bin == 0) return ((r.miss_na && r.mfb_na) || (r.miss_zero && r.mfb_zero)) ? r.missing_goes_left : r.default_goes_left;
    return !(bin > r.th);
  }
  if (r.miss_zero && !r.mfb_zero && bin == r.t_zero_bin) return r.missing_goes_left;
  if (bin != r.max_bin) return ((r.miss_na && r.mfb_na) || (r.miss_zero && r.mfb_zero)) ? r.missing_goes_left : r.default_goes_left;
  return (r.miss_na && !r.mfb_na) ? r.missing_goes_left : (r.max_bin <= r.th);
}
}  // namespace

template <bool SCATTER>
__global__ __launch_bounds__(256) void hist_partition_kernel(const uint8_t* __restrict__ bins_rm, int fpad, int feature, SplitRule rule,
                                                             const int* __restrict__ data_indices, int cnt, int* __restrict__ blk_cnt,
                                                             const int* __restrict__ blk_off, int* __restrict__ lte, int* __restrict__ gt) {
  __shared__ int s_scan[256];
  const int tid = threadIdx.x;
  const int base = blockIdx.x * 1024 + tid * 4;
  int idx[4]; bool left[4];
  int nl = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int p = base + k;
    left[k] = false; idx[k] = 0;
    if (p < cnt) {
      idx[k] = data_indices ? data_indices[p] : p;
      left[k] = goes_left(rule, (int)bins_rm[(size_t)idx[k] * fpad + feature]);
      nl += left[k] ? 1 : 0;
    }
  }
  s_scan[tid] = nl;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {            // inclusive scan of the per-lane counts
    const int v = tid >= o ? s_scan[tid - o] : 0;
    __syncthreads();
    s_scan[tid] += v;
    __syncthreads();
  }
  if (!SCATTER) {
    if (tid == 255) blk_cnt[blockIdx.x] = s_scan[255];
    return;
  }
  int l = blk_off[blockIdx.x] + s_scan[tid] - nl;                      // rows going left before this lane's first row
  int g = (blockIdx.x * 1024 + tid * 4) - l;                           // rows going right before it = position - lefts
  if (gt == nullptr) gt = lte + blk_off[gridDim.x];                    // one output segment: [left rows | right rows], the total from the scan
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (base + k < cnt) { if (left[k]) lte[l++] = idx[k]; else gt[g++] = idx[k]; }
  }
}

// The three launches above as ONE (segments of at most 256 blocks = 262 144 rows: every leaf of the tree at config 3's n = 1e5): a block
// publishes its count of left rows as an 8-byte granule {epoch, count} with a relaxed agent-scope (sc1) store and then collects the
// granules of ALL blocks -- lane t polls block t until the epoch is this launch's ("the data is the flag", cdna_hip_programming.md
// Guideline 16; the epoch makes a reset between launches unnecessary) -- scans them and scatters its rows.  At most 256 workgroups of 256
// lanes: all resident at once, so the poll cannot starve a block that has not started; it is bounded all the same (error word).
// Block 0 leaves the total in counts[] (device, for the kernels behind it) and, if asked, in pinned host memory followed by the
// sequence number the host polls for.
constexpr int kPartSpinLimit = 1 << 22;
__global__ __launch_bounds__(256) void hist_partition_onepass_kernel(const uint8_t* __restrict__ bins_rm, int fpad, int feature, SplitRule rule,
                                                                     const int* __restrict__ data_indices, int cnt, unsigned long long* tags, unsigned epoch,
                                                                     int* __restrict__ dst, int* __restrict__ counts, int* host_counts, int host_seq, int* err) {
  __shared__ int s_scan[256];
  const int tid = threadIdx.x, b = blockIdx.x, nblk = gridDim.x;
  const int base = b * 1024 + tid * 4;
  int idx[4]; bool left[4];
  int nl = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int p = base + k;
    left[k] = false; idx[k] = 0;
    if (p < cnt) {
      idx[k] = data_indices ? data_indices[p] : p;
      left[k] = goes_left(rule, (int)bins_rm[(size_t)idx[k] * fpad + feature]);
      nl += left[k] ? 1 : 0;
    }
  }
  s_scan[tid] = nl;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    const int v = tid >= o ? s_scan[tid - o] : 0;
    __syncthreads();
    s_scan[tid] += v;
    __syncthreads();
  }
  const int incl = s_scan[tid];
  if (tid == 255) __hip_atomic_store(&tags[b], ((unsigned long long)epoch << 32) | (unsigned)incl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  int v = 0;
  if (tid < nblk) {
    int spins = 0;
    for (;;) {
      const unsigned long long raw = __hip_atomic_load(&tags[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((unsigned)(raw >> 32) == epoch) { v = (int)(unsigned)raw; break; }
      if (++spins > kPartSpinLimit) { *err = 1; break; }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  s_scan[tid] = v;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    const int w = tid >= o ? s_scan[tid - o] : 0;
    __syncthreads();
    s_scan[tid] += w;
    __syncthreads();
  }
  const int total_left = s_scan[255], off_b = b > 0 ? s_scan[b - 1] : 0;
  int l = off_b + incl - nl;
  int g = base - l;
  int* gt = dst + total_left;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (base + k < cnt) { if (left[k]) dst[l++] = idx[k]; else gt[g++] = idx[k]; }
  }
  if (b == 0 && tid == 0) {
    counts[0] = total_left; counts[1] = total_left;
    if (host_counts) {
      host_counts[0] = total_left; host_counts[1] = total_left;
      if (host_seq) { __threadfence_system(); __hip_atomic_store(host_counts + 2, host_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
    }
  }
}

// exclusive scan of the block counts by one workgroup (lane = a contiguous run of blocks), total written to off[nblk] and, for the tree
// grower, to counts[0] (rows of this rank going left) and counts[1] (the same until an all-reduce over the ranks replaces it)
__global__ __launch_bounds__(256) void hist_partition_scan_kernel(const int* __restrict__ blk_cnt, int nblk, int* __restrict__ off, int* __restrict__ counts,
                                                                  int* __restrict__ host_counts) {
  // (counts: device memory, read by the kernels that follow in the stream; host_counts: pinned host memory the host reads after the
  // split's synchronisation -- written from here, no copy launch)
  __shared__ int s_run[256];
  const int tid = threadIdx.x, per = (nblk + 255) / 256, b0 = tid * per, b1 = min(b0 + per, nblk);
  int sum = 0;
  for (int b = b0; b < b1; ++b) sum += blk_cnt[b];
  s_run[tid] = sum;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    const int v = tid >= o ? s_run[tid - o] : 0;
    __syncthreads();
    s_run[tid] += v;
    __syncthreads();
  }
  int run = s_run[tid] - sum;
  for (int b = b0; b < b1; ++b) { off[b] = run; run += blk_cnt[b]; }
  if (tid == 255) {
    off[nblk] = s_run[255];
    if (counts) { counts[0] = s_run[255]; counts[1] = s_run[255]; }
    if (host_counts) { host_counts[0] = s_run[255]; host_counts[1] = s_run[255]; }
  }
}

static SplitRule make_split_rule(int max_bin, int default_bin, int most_freq_bin, int missing_type, int default_left, unsigned threshold,
                                 const unsigned* cat_bits8 = nullptr) {
  SplitRule r;
  r.is_cat = cat_bits8 ? 1 : 0; r.cat_offset = most_freq_bin == 0 ? 1 : 0;
  r.b0 = r.b1 = r.b2 = r.b3 = r.b4 = r.b5 = r.b6 = r.b7 = 0u;
  if (cat_bits8) {
    r.b0 = cat_bits8[0]; r.b1 = cat_bits8[1]; r.b2 = cat_bits8[2]; r.b3 = cat_bits8[3]; r.b4 = cat_bits8[4]; r.b5 = cat_bits8[5]; r.b6 = cat_bits8[6]; r.b7 = cat_bits8[7];
    r.max_bin = max_bin; r.t_zero_bin = 0; r.th = 0; r.miss_zero = r.miss_na = r.mfb_zero = r.mfb_na = 0; r.missing_goes_left = 0;
    r.default_goes_left = most_freq_bin > 0 && most_freq_bin < 256 && ((cat_bits8[most_freq_bin >> 5] >> (most_freq_bin & 31)) & 1u);      // :316-320
    return r;
  }
  r.max_bin = max_bin;
  r.miss_zero = missing_type == 1; r.miss_na = missing_type == 2;
  r.mfb_zero = r.miss_zero && default_bin == most_freq_bin;                          // dense_bin.hpp:268-272
  r.mfb_na = r.miss_na && (max_bin == most_freq_bin + 1 && most_freq_bin > 0);       // :294-298 with min_bin = 1
  int th = (int)((threshold + 1u) & 0xffu), tz = (1 + default_bin) & 0xff;           // VAL_T = uint8_t arithmetic (:182-187)
  if (most_freq_bin == 0) { th = (th - 1) & 0xff; tz = (tz - 1) & 0xff; }
  r.th = th; r.t_zero_bin = tz;
  r.default_goes_left = (unsigned)most_freq_bin <= threshold;                        // :196-199
  r.missing_goes_left = (r.miss_zero || r.miss_na) && default_left;                  // :200-205
  return r;
}

hipError_t launch_hist_partition(const uint8_t* bins_rm, int fpad, int feature, int max_bin, int default_bin, int most_freq_bin,
                                 int missing_type, int default_left, unsigned threshold, const int* data_indices, int cnt, int* blk_cnt,
                                 int* blk_off, int* lte, int* gt, hipStream_t st, const unsigned* cat_bits8) {
  const SplitRule r = make_split_rule(max_bin, default_bin, most_freq_bin, missing_type, default_left, threshold, cat_bits8);
  const int nblk = (cnt + 1023) / 1024;
  if (nblk == 0) return hipSuccess;
  hipLaunchKernelGGL(hist_partition_kernel<false>, dim3(nblk), dim3(256), 0, st, bins_rm, fpad, feature, r, data_indices, cnt, blk_cnt,
                     (const int*)nullptr, (int*)nullptr, (int*)nullptr);
  hipLaunchKernelGGL(hist_partition_scan_kernel, dim3(1), dim3(256), 0, st, (const int*)blk_cnt, nblk, blk_off, (int*)nullptr, (int*)nullptr);
  hipLaunchKernelGGL(hist_partition_kernel<true>, dim3(nblk), dim3(256), 0, st, bins_rm, fpad, feature, r, data_indices, cnt, blk_cnt,
                     (const int*)blk_off, lte, gt);
  return hipGetLastError();
}

// the tree grower's form: the segment's rows (src, or the identity when src == nullptr) are written as [left rows | right rows] to dst
// (the other one of its two row buffers); counts[0] = counts[1] = rows going left stay on the device for the kernels that follow
hipError_t launch_hist_partition_segment(const uint8_t* bins_rm, int fpad, int feature, int max_bin, int default_bin, int most_freq_bin,
                                         int missing_type, int default_left, unsigned threshold, const int* src, int cnt, int* blk_cnt,
                                         int* blk_off, int* dst, int* counts, int* host_counts, hipStream_t st, unsigned long long* tags,
                                         unsigned epoch, int host_seq, int* err, bool* host_seq_written, const unsigned* cat_bits8) {
  const SplitRule r = make_split_rule(max_bin, default_bin, most_freq_bin, missing_type, default_left, threshold, cat_bits8);
  const int nblk = (cnt + 1023) / 1024;
  if (host_seq_written) *host_seq_written = false;
  if (nblk == 0) { if (host_counts) { host_counts[0] = 0; host_counts[1] = 0; } return hipMemsetAsync(counts, 0, 2 * sizeof(int), st); }
  if (tags && nblk <= 256) {
    hipLaunchKernelGGL(hist_partition_onepass_kernel, dim3(nblk), dim3(256), 0, st, bins_rm, fpad, feature, r, src, cnt, tags, epoch, dst, counts,
                       host_counts, host_seq, err);
    if (host_seq_written) *host_seq_written = host_counts != nullptr && host_seq != 0;
    return hipGetLastError();
  }
  hipLaunchKernelGGL(hist_partition_kernel<false>, dim3(nblk), dim3(256), 0, st, bins_rm, fpad, feature, r, src, cnt, blk_cnt,
                     (const int*)nullptr, (int*)nullptr, (int*)nullptr);
  hipLaunchKernelGGL(hist_partition_scan_kernel, dim3(1), dim3(256), 0, st, (const int*)blk_cnt, nblk, blk_off, counts, host_counts);
  hipLaunchKernelGGL(hist_partition_kernel<true>, dim3(nblk), dim3(256), 0, st, bins_rm, fpad, feature, r, src, cnt, blk_cnt,
                     (const int*)blk_off, dst, (int*)nullptr);
  return hipGetLastError();
}

}  // namespace gpb
