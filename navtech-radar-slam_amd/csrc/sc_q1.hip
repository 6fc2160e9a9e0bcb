// sc_q1.hip -- ONE exhaustive ScanContext query in ONE launch (gfx950 / CDNA4): the live detector's regime.
//
// What it replaces.  The reference's loop detection is one query at 1 Hz (laserPosegraphOptimization.cpp:561,577 ->
// detectLoopClosureID, Scancontext.cpp:331-422); in exhaustive mode it scores EVERY eligible entry with
// distanceBtnScanContext (SC.cpp:116-148).  The batched machinery (spectral filter -> select -> window -> re-scoring: six
// dependent launches) amortises its fixed costs over thousands of queries; run with one query it is all
// fixed cost (55 us at 10 000 entries, 0.11 of the HBM roofline north_star names for this regime).  A single query cannot
// re-use anything: every eligible entry has to be streamed out of HBM once, so this kernel is built around that stream.
//
// One launch, a grid of <= 512 four-wave workgroups (two per CU), each walking the 32-entry tiles b, b + G, b + 2G, ...:
//   prep      every workgroup builds the query's images in LDS itself (sector key, column norms, the fp16 image of the
//             direct filter and the fp16 hi/lo key circulant: the bodies of sc_keys / sc_img_query / sc_win_query_keys),
//             while the first tile's database fragments are already in flight;
//   stream    per tile the fp16 image of the 32 entries (tile-major hnT: 75 coalesced 1-KiB fragments, 2400 B per entry)
//             is cut BY K over the four waves (19 K-steps each: every wave has its whole share in flight at once -- one
//             memory latency per tile instead of five ring turns), S[k][e] = sum_i q2[i + 20k] e[i] on the matrix cores
//             exactly as sc_window_kernel computes it; the partial sums of three waves go through LDS to the tile's
//             epilogue wave (rotating), which also runs the sector-key alignment GEMM (24 MFMAs, fp16 hi/lo keys) and the
//             window epilogue of sc_window_dev.h: per entry k* (when unique within the error bound) and a preview pv with
//             |pv - dist| <= WINDOW_MARGIN (a lower bound only when k* is not unique).  2672 B per entry leave HBM.
//   select    the workgroup keeps its k smallest preview upper bounds; an entry whose lower bound pv - margin can still
//             reach the workgroup's k-th smallest upper bound becomes a candidate record (16 B, write-through stores);
//   finish    the workgroup that draws the last arrival ticket (one relaxed agent-scope fetch_add per workgroup, records
//             published with sc1 stores + vmcnt(0) drain, read with sc1 loads: MI355X_MICROARCH "valid forms") merges the
//             per-workgroup bounds into the chip-wide k-th smallest upper bound, keeps the candidates that can still reach it
//             and evaluates those -- a handful -- EXACTLY (align_exact / phase_b32 of sc_exact_dev.h, the same fp64 pair
//             function in the same order as every other path), four per round, in ascending order of their lower bound with
//             the exact k-th best tightening the cut.  It writes the k records; no other launch, no host round trip.
// Results are byte-identical to the exact-all path and to the oracle: every record comes out of the exact pair function, and
// an entry is skipped only if a proven lower bound of its distance exceeds a proven upper bound of the k-th best.
//
// Roofline (HBM): SURVEY 8d prices a pair at 4800 B (one fp32 descriptor); this kernel reads 2672 B per entry (fp16 image
// 2400 + key image 256 + norms 8 + mask 8).  bench.py latency_q1 reports both the algorithmic and the physical fraction.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <cstdint>

#include "rsx_common.h"
#include "sc_kernels.h"
#include "sc_entry_dev.h"
#include "sc_exact_dev.h"
#include "sc_window_dev.h"

namespace rsx {
namespace sc {

namespace {

using win::half8;
using win::floatx16;
using win::u64;
using win::kNonFinite;
using win::W_STEPS;
using win::W_TILE1;
using win::QK_COPY;
using win::QK_LO;
using win::QK_NORM;
using win::W_LDS;

constexpr int Q1_KS = 19;  // K-steps of the image GEMM per wave: 75 = 19 + 19 + 19 + 18
static_assert(4 * Q1_KS >= W_STEPS && 3 * Q1_KS < W_STEPS, "the four waves cover the 75 K-steps");
// LDS of one workgroup
constexpr int Q1_OFF_QIMG = 0;                          // the direct filter's query image (two displaced fp16 copies + mask)
constexpr int Q1_OFF_QK = FILTER_QIMG_BYTES;            // 9984: the key circulant (8 displaced hi copies, 8 lo, norms)
constexpr int Q1_OFF_Q = W_LDS;                         // 14608: WaveLds's query part: fp32 descriptor | norms | sector key
constexpr int Q1_OFF_MISC = Q1_OFF_Q + WaveLds::OFF_ENT;  // 20432
constexpr int Q1_MISC_UB = 0;      // float[32]: the workgroup's k smallest preview upper bounds, ascending
constexpr int Q1_MISC_CNT = 128;   // int: candidate records written so far
constexpr int Q1_MISC_LAST = 132;  // int: this workgroup drew the last ticket
constexpr int Q1_MISC_NSURV = 136; // int: survivors gathered so far
constexpr int Q1_MISC_TOTAL = 224, Q1_MISC_MORE = 228;  // ints of the last workgroup
constexpr int Q1_MISC_TAU = 144;   // double: the cut of the exact rounds
constexpr int Q1_MISC_MASK = 152;  // u64: the query's column mask (between the two prep phases)
constexpr int Q1_MISC_RED = 160;   // float[8]: per-wave minima of the bound merge, its result
constexpr int Q1_MISC_BYTES = 256;
constexpr int Q1_OFF_SCR = Q1_OFF_MISC + Q1_MISC_BYTES;  // 20688
static_assert(Q1_OFF_SCR % 16 == 0, "alignment");
constexpr int Q1_PART_BYTES = 3 * 32 * 64 * 4;           // one buffer of partial sums: 3 waves x 32 accumulators x 64 lanes
constexpr int Q1_SCR_BYTES = 2 * Q1_PART_BYTES;          // 49152
constexpr int Q1_LDS = Q1_OFF_SCR + Q1_SCR_BYTES;        // 69840: two workgroups per CU
// the scratch region during prep ...
constexpr int Q1_SCR_ST = 0;                             // _Float16[1200]: normalised columns
constexpr int Q1_SCR_ST2 = 2432;                         // _Float16[2][128]: doubled hi / lo key
// ... and in the last workgroup
constexpr int Q1_FIN_ENT = 0;                            // 4 x ENT_SIZE: the waves' exact-evaluation regions
constexpr int Q1_FIN_RES = 4 * ENT_SIZE;                 // 13632: 4 x {double dist; int idx; int shift} results of a round
constexpr int Q1_FIN_PREF = Q1_FIN_RES + 64;             // 13696: int[G + 1] prefix sums of the candidate counts beyond the eager ones
constexpr int Q1_MAX_G = 512;
constexpr int Q1_FIN_AB = Q1_FIN_PREF + (Q1_MAX_G + 1) * 4 + 12;  // 15760: bound lists (merge), then the survivors
static_assert(Q1_FIN_AB % 16 == 0, "alignment");
constexpr int Q1_FIN_AB_BYTES = Q1_SCR_BYTES - Q1_FIN_AB;  // 33392
constexpr int Q1_SURV_CAP = 2560;                        // survivors gathered before they are evaluated: lo[], slot[], ks[]
static_assert(3 * 4 * Q1_SURV_CAP <= Q1_FIN_AB_BYTES, "survivor arrays fit");
constexpr int Q1_SORT_MAX = 256;                         // chunks with at most this many survivors are evaluated in ascending lo
// what the workgroups publish (global memory, u64 arrays per query; G = workgroups): hdr[G] = candidate count | smallest upper
// bound << 32; ubs[16][G] = the bound list in pairs; rec[2 * Q1_SOA][G] = the first candidate records {slot << 32 | lo bits,
// k* | shift mask} -- all indexed by workgroup LAST, so that the last workgroup's thread j reads workgroup j and a wavefront's
// load touches 4 cache lines instead of 64 (the first build kept one block per workgroup: 3 400 eight-byte requests out of one
// CU for 313 workgroups, 4.4 us) -- and blk[G][...] the records beyond the eager ones
constexpr int Q1_EAGER = 4;  // candidate records per workgroup the last workgroup requests together with the header
constexpr int Q1_SOA = 16;   // candidate records per workgroup in the workgroup-last arrays (the rest: one block per workgroup)
static_assert(2 * Q1_EAGER * 256 <= Q1_SURV_CAP && Q1_SOA % Q1_EAGER == 0, "a step's records of 512 workgroups fit the survivor list");
// arrival counters: per query 8 group counters (workgroup b arrives at group b % 8: ~13 ns per atomic on ONE word is 4 us
// for 313 workgroups arriving together, MI355X_MICROARCH "fanin") + one for the groups, each on its own 128-byte line
constexpr int Q1_TICKET_STRIDE = 32;  // unsigned per counter
constexpr int Q1_TICKETS_PER_Q = 9;

struct Q1Args {
  DbView db;
  const float *qdesc;      // [nq][1200]
  int64_t n_items;         // local slots that can be eligible at all
  int64_t n_eligible;      // global index limit
  const int64_t *q_elig;   // optional per-query limit
  rsx_sc_hit *out;         // [nq][k]
  int32_t k, kp;           // kp = k rounded up to 4: floats of a workgroup's bound list the merge reads
  int32_t blk;             // u64 per workgroup in ws_blk: 2 x (its entries beyond the eager records)
  unsigned *ticket;        // [nq][Q1_TICKETS_PER_Q][Q1_TICKET_STRIDE]: zero between launches (the last workgroup resets them)
  u64 *ws_hdr, *ws_ubs, *ws_rec, *ws_blk;  // [nq][G], [nq][16][G], [nq][2 * Q1_SOA][G], [nq][G][blk]
  unsigned long long *stats;  // optional (profiling): RESCORE_STAT_WORDS counters
};

__device__ __forceinline__ void store_sc1(u64 *p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u64 load_sc1(const u64 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ u64 wave_min_u64(u64 v) { return dev::wave_minmax_u64<false>(v); }

// RSX_EXPERIMENTS builds with profiling on: the last workgroup adds the length of its phases (10-ns ticks of the constant
// clock) to stats words 4..10, 13, 14: prep | tiles | publish | ticket | header loads | bound merge | total | eager rounds | rest
#ifdef RSX_EXPERIMENTS
#define Q1_MARK(i) do { if (a.stats) tmark[i] = wall_clock64(); } while (0)
#else
#define Q1_MARK(i) do { } while (0)
#endif

template <int SO>
__global__ __launch_bounds__(256, 2) void sc_q1_kernel(Q1Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef RSX_EXPERIMENTS
  unsigned long long tmark[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
  Q1_MARK(0);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // in an SGPR: everything decided per wave is a scalar branch
  const int n = lane & 31, hh = lane >> 5;
  const int G = (int)gridDim.x, b = (int)blockIdx.x, qi = (int)blockIdx.y;
  char *misc = smem + Q1_OFF_MISC;
  char *scr = smem + Q1_OFF_SCR;
  float *s_ub = reinterpret_cast<float *>(misc + Q1_MISC_UB);
  int *s_cnt = reinterpret_cast<int *>(misc + Q1_MISC_CNT);

  int64_t n_elig = a.n_eligible;
  if (a.q_elig) {
    const int64_t e = a.q_elig[qi];
    n_elig = e < n_elig ? e : n_elig;
  }
  int64_t n_rows = 0;  // local slots [0, n_rows) are the eligible ones
  if (n_elig > a.db.idx_base) {
    n_rows = (n_elig - a.db.idx_base + a.db.idx_stride - 1) / a.db.idx_stride;
    n_rows = n_rows < a.n_items ? n_rows : a.n_items;
  }
  const int ntiles = (int)((n_rows + 31) >> 5);

  // ---- the query's column of this lane, requested FIRST (a wave's loads return in order: behind the tile's fragments the
  // 4.8 KB of the query would wait for 27 KB per wave; waves 0 and 1 each keep a copy) ----
  float4 qcol[5] = {};
  if (wave < 2 && lane < NS) {
    const float4 *src = reinterpret_cast<const float4 *>(a.qdesc + (int64_t)qi * DS + lane * NR);
#pragma unroll
    for (int i = 0; i < 5; i++) qcol[i] = src[i];
  }
  // ---- database fragments of one tile: this wave's K-steps of the image, and for the tile's epilogue wave the key
  // fragments (requested first: VMEM returns in order and the alignment runs while the image is still on its way) ----
  const int ks0 = Q1_KS * wave;
  const bool full = wave != 3;  // K-steps of this wave: 19, 19, 19, 18
  half8 frag[Q1_KS];
  half8 bh[4], bl[4];
  float2 en = {0.0f, 0.0f};
  u64 em = 0;
  auto issue = [&](int t, int it) __attribute__((always_inline)) {
    if (wave == (it & 3)) {
      const int64_t slot = (int64_t)t * 32 + n;
      const char *bk = static_cast<const char *>(a.db.vk16) + slot * 256 + 16 * hh;
#pragma unroll
      for (int s = 0; s < 4; s++) {
        bh[s] = *reinterpret_cast<const half8 *>(bk + 32 * s);
        bl[s] = *reinterpret_cast<const half8 *>(bk + 128 + 32 * s);
      }
      en = reinterpret_cast<const float2 *>(a.db.vk_n)[slot];
      em = a.db.cmask[slot];
    }
    const half8 *bt = static_cast<const half8 *>(a.db.hnT) + ((int64_t)t * W_STEPS + ks0) * 64 + lane;
#pragma unroll
    for (int u = 0; u < Q1_KS - 1; u++) frag[u] = bt[u * 64];
    if (full) frag[Q1_KS - 1] = bt[(Q1_KS - 1) * 64];
  };
  if (b < ntiles) issue(b, 0);

  // ---- the query's images, built here (what sc_keys / sc_img_query / sc_win_query_keys do for a batch): wave 0 the column
  // norms + normalised columns (and the fp32 copy the exact evaluation reads), wave 1 the doubled hi / lo key, then every
  // thread a share of the two displaced image copies and the 16 displaced key copies ----
  if (tid < 32) s_ub[tid] = INFINITY;
  if (tid == 0) *s_cnt = 0;
  _Float16 *st = reinterpret_cast<_Float16 *>(scr + Q1_SCR_ST);
  _Float16(*st2)[128] = reinterpret_cast<_Float16(*)[128]>(scr + Q1_SCR_ST2);
  dev::KeySplit ksplit{};
  if (wave == 0) {
    double vk = 0.0, nr = 0.0;
    dev::column_keys<SO>(qcol, vk, nr);
    bool nonzero = false, bad = false;
    if (lane < NS) {
      float4 *qf = reinterpret_cast<float4 *>(smem + Q1_OFF_Q) + lane * (NR / 4);
#pragma unroll
      for (int i = 0; i < 5; i++) qf[i] = qcol[i];
      reinterpret_cast<double *>(smem + Q1_OFF_Q + WaveLds::OFF_QN1)[lane] = nr;
      reinterpret_cast<double *>(smem + Q1_OFF_Q + WaveLds::OFF_QV1)[lane] = vk;
      dev::normalise_column_regs(qcol, nr, &st[lane * NR], nonzero, bad);
    }
    u64 m = __ballot(nonzero && lane < NS);
    if (__ballot(bad && lane < NS)) m |= kNonFinite;
    if (lane == 0) *reinterpret_cast<u64 *>(misc + Q1_MISC_MASK) = m;
  } else if (wave == 1) {
    double vk = 0.0, nr = 0.0;
    dev::column_keys<SO>(qcol, vk, nr);
    ksplit = win::query_keys_stage<true>(lane < NS ? vk : 0.0, st2, lane);
  }
  __syncthreads();
  dev::img_query_image(st, *reinterpret_cast<const u64 *>(misc + Q1_MISC_MASK), reinterpret_cast<uint4 *>(smem + Q1_OFF_QIMG), tid, 256);
  win::query_keys_copies(st2, smem + Q1_OFF_QK, tid, 256);
  if (wave == 1) win::query_keys_norms(ksplit, smem + Q1_OFF_QK, lane);
  __syncthreads();

  Q1_MARK(1);
  const u64 qm = *reinterpret_cast<const u64 *>(smem + Q1_OFF_QIMG + FILTER_QIMG_MASK_OFF);
  const float nq_key = *reinterpret_cast<const float *>(smem + Q1_OFF_QK + QK_NORM);
  const float uq_key = *reinterpret_cast<const float *>(smem + Q1_OFF_QK + QK_NORM + 4);
  // A-fragment addresses of this lane's row (shift n of tile 0; tile 1 = the same address + 40 K-steps, sc_filter.hip)
  const char *ap = smem + Q1_OFF_QIMG + ((n & 1) ? (FILTER_QIMG_ODD + 40 * n - 8) : (40 * n)) + 16 * hh + 32 * ks0;
  const char *kp = smem + Q1_OFF_QK + (n & 7) * QK_COPY + ((n & ~7) + 8 * hh) * 2;  // tile 1: + 64 B
  u64 *my_rec = a.ws_rec + (int64_t)qi * (2 * Q1_SOA) * G + b;   // record e < Q1_SOA: [2 e] and [2 e + 1], stride G
  u64 *my_blk = a.ws_blk + ((int64_t)qi * G + b) * a.blk;

  int it = 0;
  for (int t = b; t < ntiles; t += G, it++) {
    const int ew = it & 3;
    const bool epi = wave == ew;
    u64 win = 0;
    int kstar = -1;
    if (epi) {
      // ---- alignment: 2 tiles x (hi*hi + hi*lo + lo*hi) x 4 K-steps (sc_window.hip) ----
      floatx16 k0 = {0}, k1 = {0};
#pragma unroll
      for (int s = 0; s < 4; s++) {
        const half8 ah0 = *reinterpret_cast<const half8 *>(kp + 32 * s);
        const half8 al0 = *reinterpret_cast<const half8 *>(kp + QK_LO + 32 * s);
        const half8 ah1 = *reinterpret_cast<const half8 *>(kp + 64 + 32 * s);
        const half8 al1 = *reinterpret_cast<const half8 *>(kp + QK_LO + 64 + 32 * s);
        k0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, bh[s], k0, 0, 0, 0);
        k1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1, bh[s], k1, 0, 0, 0);
        k0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, bl[s], k0, 0, 0, 0);
        k1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1, bl[s], k1, 0, 0, 0);
        k0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al0, bh[s], k0, 0, 0, 0);
        k1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al1, bh[s], k1, 0, 0, 0);
      }
      win::alignment_of<true>(k0, k1, nq_key, uq_key, en, hh, win, kstar);
    }
    // ---- this wave's K-steps of the 60 correlation values (the direct filter's GEMM) ----
    floatx16 acc0 = {0}, acc1 = {0};
#pragma unroll
    for (int u = 0; u < Q1_KS; u++) {
      if (u < Q1_KS - 1 || full) {
        const half8 a0 = *reinterpret_cast<const half8 *>(ap + 32 * u);
        const half8 a1 = *reinterpret_cast<const half8 *>(ap + 32 * (u + W_TILE1));
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, frag[u], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, frag[u], acc1, 0, 0, 0);
      }
    }
    const u64 em_t = em;  // (the next tile's epilogue wave is another wave: its loads do not touch this one's mask)
    if (t + G < ntiles) issue(t + G, it + 1);
    // ---- partial sums of the other three waves -> LDS (two buffers: one barrier per tile) ----
    float4 *part = reinterpret_cast<float4 *>(scr + (it & 1) * Q1_PART_BYTES);
    if (!epi) {
      const int ps = (wave - ew - 1) & 3;  // 0, 1, 2
#pragma unroll
      for (int r4 = 0; r4 < 4; r4++) {
        part[(ps * 8 + r4) * 64 + lane] = float4{acc0[4 * r4], acc0[4 * r4 + 1], acc0[4 * r4 + 2], acc0[4 * r4 + 3]};
        part[(ps * 8 + 4 + r4) * 64 + lane] = float4{acc1[4 * r4], acc1[4 * r4 + 1], acc1[4 * r4 + 2], acc1[4 * r4 + 3]};
      }
    }
    __syncthreads();
    if (epi) {
#pragma unroll
      for (int ps = 0; ps < 3; ps++) {
#pragma unroll
        for (int r4 = 0; r4 < 4; r4++) {
          const float4 p0 = part[(ps * 8 + r4) * 64 + lane], p1 = part[(ps * 8 + 4 + r4) * 64 + lane];
          acc0[4 * r4] += p0.x; acc0[4 * r4 + 1] += p0.y; acc0[4 * r4 + 2] += p0.z; acc0[4 * r4 + 3] += p0.w;
          acc1[4 * r4] += p1.x; acc1[4 * r4 + 1] += p1.y; acc1[4 * r4 + 2] += p1.z; acc1[4 * r4 + 3] += p1.w;
        }
      }
      const float pv = win::preview_of<true>(acc0, acc1, qm, em_t, win, kstar, hh);
      // ---- the entry's bounds: lanes 0..31 carry entry t * 32 + n (lanes 32..63 hold the same values) ----
      const int64_t slot = (int64_t)t * 32 + n;
      const bool have = hh == 0 && slot < n_rows;
      const float ub = (have && kstar >= 0 && pv < 3.0e38f) ? pv + WINDOW_MARGIN : INFINITY;  // NaN fails the compare
      float lo = INFINITY;  // +inf: never a hit (no effective column in the window) or not eligible
      if (have) lo = (pv == pv) ? pv - WINDOW_MARGIN : -INFINITY;  // NaN: no preview (non-finite data), must be looked at
      // the workgroup's k smallest upper bounds, ascending, one per lane
      float lub = s_ub[n];
      auto lane_of = [](float x, int l) { return __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(x), l)); };
      float kth = lane_of(lub, a.k - 1);
      if (a.k == 1) {  // (the detector's case: the list is one value)
        kth = fminf(kth, wave_min_f32(ub));
        lub = lane == 0 ? kth : lub;
      }
      u64 pend = a.k == 1 ? 0ull : __ballot(ub < kth);
      while (pend) {
        const int l = __ffsll((long long)pend) - 1;
        pend &= pend - 1;
        const float v = lane_of(ub, l);
        if (!(v < kth)) continue;
        const int pos = __popcll(__ballot(lane < a.k && lub <= v));  // < k: lub[k - 1] = kth > v
        const float up = __shfl_up(lub, 1);
        if (lane < a.k) {
          if (lane > pos) lub = up;
          else if (lane == pos) lub = v;
        }
        kth = lane_of(lub, a.k - 1);
      }
      if (lane < 32) s_ub[lane] = lub;
      // candidates: lower bound not above the workgroup's k-th smallest upper bound (which is never below the chip's)
      const bool cand = lo < INFINITY && !(lo > kth);
      const u64 cb = __ballot(cand);
      if (cb) {
        const int base = *s_cnt;
        if (cand) {
          const int pos = base + __popcll(cb & ((1ull << lane) - 1ull));
          u64 *r0 = pos < Q1_SOA ? my_rec + (int64_t)(2 * pos) * G : my_blk + 2 * (pos - Q1_SOA);
          u64 *r1 = pos < Q1_SOA ? r0 + G : r0 + 1;
          store_sc1(r0, ((u64)(unsigned)slot << 32) | (u64)__float_as_uint(lo));
          store_sc1(r1, (u64)(unsigned)kstar);
        }
        wave_lds_fence();
        if (lane == 0) *s_cnt = base + __popcll(cb);
      }
    }
  }
  __syncthreads();
  Q1_MARK(2);

  // ---- publish: the bound list and the candidate count of this workgroup, then the arrival ticket ----
  if (wave == 0) {
    if (a.k > 1 && lane < (a.kp >> 1))
      store_sc1(a.ws_ubs + ((int64_t)qi * 16 + lane) * G + b, reinterpret_cast<const u64 *>(s_ub)[lane]);
    if (lane == 16) store_sc1(a.ws_hdr + (int64_t)qi * G + b, (u64)(unsigned)*s_cnt | ((u64)__float_as_uint(s_ub[0]) << 32));
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wave: its write-through stores have left
  __syncthreads();
  Q1_MARK(3);
  unsigned *tk = a.ticket + (size_t)qi * Q1_TICKETS_PER_Q * Q1_TICKET_STRIDE;
  if (tid == 0) {
    const int grp = b & 7;
    const unsigned members = (unsigned)((G - grp + 7) >> 3), groups = (unsigned)(G < 8 ? G : 8);
    int last = 0;
    const unsigned old = __hip_atomic_fetch_add(tk + grp * Q1_TICKET_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old + 1u == members) {
      const unsigned old2 = __hip_atomic_fetch_add(tk + 8 * Q1_TICKET_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      last = (old2 + 1u == groups) ? 1 : 0;
    }
    *reinterpret_cast<int *>(misc + Q1_MISC_LAST) = last;
  }
  __syncthreads();
  if (*reinterpret_cast<const int *>(misc + Q1_MISC_LAST) == 0) return;
  Q1_MARK(4);

  // =============================== the last workgroup ===============================
  if (tid < Q1_TICKETS_PER_Q)  // the counters of this query are zero again when the next launch arrives
    __hip_atomic_store(tk + tid * Q1_TICKET_STRIDE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  int *s_pref = reinterpret_cast<int *>(scr + Q1_FIN_PREF);
  // ---- ONE memory round trip: of every workgroup (two per thread) the candidate count + smallest bound, the first
  // Q1_EAGER candidate records and (k > 1) the bound list (straight-line write-through loads: the compiler batches them; in
  // a loop each would wait for the one before) ----
  u64 hdr_r[2], rec[2][2 * Q1_EAGER], lst[2][16];
  const int kp2 = a.k > 1 ? (a.kp >> 1) : 0;
  const u64 *rec_q = a.ws_rec + (int64_t)qi * (2 * Q1_SOA) * G;
  int jj[2];
#pragma unroll
  for (int o = 0; o < 2; o++) jj[o] = tid + 256 * o < G ? tid + 256 * o : 0;
#pragma unroll
  for (int o = 0; o < 2; o++) {
    const bool own = tid + 256 * o < G;
    hdr_r[o] = own ? load_sc1(a.ws_hdr + (int64_t)qi * G + jj[o]) : 0;
#pragma unroll
    for (int c = 0; c < 2 * Q1_EAGER; c++) rec[o][c] = own ? load_sc1(rec_q + (int64_t)c * G + jj[o]) : 0;
#pragma unroll
    for (int c = 0; c < 16; c++) lst[o][c] = (own && c < kp2) ? load_sc1(a.ws_ubs + ((int64_t)qi * 16 + c) * G + jj[o]) : 0;
  }
  int *s_nsurv = reinterpret_cast<int *>(misc + Q1_MISC_NSURV);
  int *s_total = reinterpret_cast<int *>(misc + Q1_MISC_TOTAL);  // candidate records beyond the Q1_SOA per workgroup
  int *s_more = reinterpret_cast<int *>(misc + Q1_MISC_MORE);    // some workgroup has more than Q1_EAGER
  if (tid == 0) {
    *s_total = 0;
    *s_more = 0;
    *s_nsurv = 0;
  }
  __syncthreads();
  float t0 = INFINITY;  // smallest k-th bound of the lists this thread holds
  {
    u64 *L = reinterpret_cast<u64 *>(scr + Q1_FIN_AB);
    int extra_mine = 0, more_mine = 0;
#pragma unroll
    for (int o = 0; o < 2; o++) {
      const int j = tid + 256 * o;
      if (j < G) {
        if (a.k > 1) {
#pragma unroll
          for (int c = 0; c < 16; c++)
            if (c < kp2) L[j * kp2 + c] = lst[o][c];
        } else {
          t0 = fminf(t0, __uint_as_float((unsigned)(hdr_r[o] >> 32)));
        }
        const int cnt = (int)(unsigned)hdr_r[o];
        const int extra = cnt - Q1_SOA;
        s_pref[j + 1] = extra > 0 ? extra : 0;
        extra_mine += extra > 0 ? extra : 0;
        more_mine |= cnt > Q1_EAGER ? 1 : 0;
      }
    }
    if (tid == 0) s_pref[0] = 0;
    if (extra_mine) atomicAdd(s_total, extra_mine);
    if (more_mine) atomicOr(s_more, 1);
  }
  __syncthreads();
  Q1_MARK(5);
  const int total = *s_total;
  const bool more = *s_more != 0;
  if (total > 0 && wave == 0) {  // inclusive scan of s_pref[1 .. G] (G <= 512: 8 per lane); read after later barriers only
    int v[8], sum = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int j = lane * 8 + i;
      v[i] = j < G ? s_pref[j + 1] : 0;
      sum += v[i];
    }
    int inc = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int o = __shfl_up(inc, off);
      if (lane >= off) inc += o;
    }
    int run = inc - sum;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int j = lane * 8 + i;
      run += v[i];
      if (j < G) s_pref[j + 1] = run;
    }
  }
  // ---- the chip-wide k-th smallest upper bound.  k = 1: the smallest of the workgroups' bounds.  k > 1: the sorted lists
  // are merged by ONE wavefront, k rounds of "smallest head" (a lane owns 8 lists and keeps their heads in registers; a
  // round is a lane-local minimum, one DPP reduction and one LDS read by the owner: no barrier) ----
  float tau_ub;
  {
    const float *L = reinterpret_cast<const float *>(scr + Q1_FIN_AB);
    float *s_redf = reinterpret_cast<float *>(misc + Q1_MISC_RED);
    if (a.k == 1) {
      t0 = wave_min_f32(t0);
      if (lane == 0) s_redf[wave] = t0;
      __syncthreads();
      tau_ub = fminf(fminf(s_redf[0], s_redf[1]), fminf(s_redf[2], s_redf[3]));
    } else {
      if (wave == 0) {
        u64 head[8];
        int ptr[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
          const int j = lane + 64 * i;
          ptr[i] = 0;
          head[i] = j < G ? (((u64)dev::enc_f32(L[j * a.kp]) << 32) | (unsigned)j) : ~0ull;
        }
        float kth = INFINITY;
        for (int r = 0; r < a.k; r++) {
          u64 m = head[0];
#pragma unroll
          for (int i = 1; i < 8; i++) m = head[i] < m ? head[i] : m;
          const u64 g = wave_min_u64(m);
          if (g == ~0ull) {  // fewer than k bounds at all
            kth = INFINITY;
            break;
          }
          kth = dev::dec_f32((unsigned)(g >> 32));
          const int owner = (int)(unsigned)g;
          if ((owner & 63) == lane) {  // the owner moves on in that list
            const int oi = owner >> 6;
#pragma unroll
            for (int i = 0; i < 8; i++)
              if (i == oi) {
                ptr[i]++;
                head[i] = ptr[i] < a.k ? (((u64)dev::enc_f32(L[owner * a.kp + ptr[i]]) << 32) | (unsigned)owner) : ~0ull;
              }
          }
        }
        if (lane == 0) s_redf[4] = kth;
      }
      __syncthreads();
      tau_ub = s_redf[4];
    }
  }
  __syncthreads();  // the lists are dead: their space holds the survivors from here on
  Q1_MARK(6);

  // ---- candidates that can still reach the bound; exact evaluation four per round, the next round's entries requested
  // before the current ones are evaluated ----
  float *s_lo = reinterpret_cast<float *>(scr + Q1_FIN_AB);
  int *s_slot = reinterpret_cast<int *>(s_lo + Q1_SURV_CAP);
  int *s_ks = s_slot + Q1_SURV_CAP;
  double *s_tau = reinterpret_cast<double *>(misc + Q1_MISC_TAU);
  struct Res {
    double d;
    int idx, shift;
  };
  Res *s_res = reinterpret_cast<Res *>(scr + Q1_FIN_RES);
  char *wsm = scr + Q1_FIN_ENT + wave * ENT_SIZE;
  const char *qsm = smem + Q1_OFF_Q;
  double ld = INFINITY;  // wave 0: the sorted top-k of exact hits, one record per lane
  int li = 0x7fffffff, ls = 0;
  double tau = (double)tau_ub;
  unsigned n_exact = 0, n_aligned = 0, n_shifts = 0, n_surv_all = 0, n_cand_all = 0;

  auto keep = [&](u64 ra, u64 rb) __attribute__((always_inline)) {  // a candidate record whose lower bound can still reach the cut -> survivor list
    const float lo = __uint_as_float((unsigned)ra);
    if (!((double)lo > tau)) {
      const int i = atomicAdd(s_nsurv, 1);
      s_lo[i] = lo;
      s_slot[i] = (int)(ra >> 32);
      s_ks[i] = (int)(unsigned)rb;
    }
  };
  // the survivors gathered so far: exact evaluation, then the list is empty again (all threads call this, behind a barrier)
  auto rounds = [&]() __attribute__((always_inline)) {
    const int ns = *s_nsurv;
    n_surv_all += (unsigned)ns;
    if (ns == 0) return;  // uniform
    // ascending (lo, slot) when there are few: the exact k-th best then takes over early and the tail is cut
    const bool sorted = ns <= Q1_SORT_MAX;
    if (sorted && ns > 1) {
      float my_lo = 0.0f;
      int my_slot = 0, my_ks = 0, rank = 0;
      if (tid < ns) {
        my_lo = s_lo[tid];
        my_slot = s_slot[tid];
        my_ks = s_ks[tid];
        for (int j = 0; j < ns; j++) {
          const float oj = s_lo[j];
          const int sj = s_slot[j];
          rank += (oj < my_lo || (oj == my_lo && sj < my_slot)) ? 1 : 0;
        }
      }
      __syncthreads();
      if (tid < ns) {
        s_lo[rank] = my_lo;
        s_slot[rank] = my_slot;
        s_ks[rank] = my_ks;
      }
      __syncthreads();
    }
    EntryRegs cur, nxt;
    if (wave < ns && !((double)s_lo[wave] > tau)) load_entry(a.db, s_slot[wave], lane, cur);
    for (int i0 = 0; i0 < ns; i0 += 4) {
      if (sorted && (double)s_lo[i0] > tau) break;  // everything after it is larger still (uniform)
      const int i = i0 + wave;
      // (tau only decreases: an entry that is evaluated below passed this test when its registers were requested)
      if (i + 4 < ns && !((double)s_lo[i + 4] > tau)) load_entry(a.db, s_slot[i + 4], lane, nxt);
      Res r;
      r.d = INFINITY;
      r.idx = 0;
      r.shift = 0;
      if (i < ns && !((double)s_lo[i] > tau)) {  // (wave-uniform)
        const int64_t slot = s_slot[i];
        const int ksm = s_ks[i];
        int ks = ksm;
        unsigned tmask = 0x7fu;
        if (ksm >= 0) {  // k* and the shifts of its window that can be the minimum (sc_window_dev.h)
          ks = ksm & 63;
          const unsigned m7 = ((unsigned)ksm >> 8) & 0x7fu;
          if (m7) tmask = m7;
        }
        if (ks < 0) {
          ks = align_exact<SO>(reinterpret_cast<const double *>(qsm + WaveLds::OFF_QV1), wsm, lane, cur.v);
          n_aligned++;
        }
        double bd;
        int bk;
        phase_b32<SO>(qsm, wsm, lane, cur, ks, tmask, bd, bk);
        n_exact++;
        n_shifts += (unsigned)__builtin_popcount(tmask);
        r.d = bd;
        r.idx = (int)(a.db.idx_base + slot * a.db.idx_stride);
        r.shift = bk;
      }
      if (lane == 0) s_res[wave] = r;
      __syncthreads();
      if (wave == 0) {
#pragma unroll 1
        for (int w = 0; w < 4; w++) {
          const Res o = s_res[w];
          if (o.d < kBig) topk_insert_dpp(ld, li, ls, lane, a.k, o.d, o.idx, o.shift);
        }
        const double kd = __longlong_as_double((long long)dev::readlane_u64((u64)__double_as_longlong(ld), a.k - 1));
        if (lane == 0) *s_tau = kd < (double)tau_ub ? kd : (double)tau_ub;
      }
      __syncthreads();
      tau = *s_tau;
      cur = nxt;
    }
    __syncthreads();
    if (tid == 0) *s_nsurv = 0;
    __syncthreads();
  };
  // (1) the eager records
#pragma unroll
  for (int o = 0; o < 2; o++) {
    const int cnt = (int)(unsigned)hdr_r[o];
    n_cand_all += (unsigned)cnt;
#pragma unroll
    for (int c = 0; c < Q1_EAGER; c++)
      if (c < cnt) keep(rec[o][2 * c], rec[o][2 * c + 1]);
  }
  __syncthreads();
  Q1_MARK(7);
  // (2) the other records in pieces -- first records Q1_EAGER .. Q1_SOA - 1 of every workgroup, four at a time (one batched
  // round trip each), then what lies beyond (a workgroup with more than Q1_SOA candidates: duplicates, k near the tile
  // size) -- and ONE call site of the exact rounds: whenever the next piece might not fit the survivor list, and at the end
  const int nsoa = more ? Q1_SOA / Q1_EAGER - 1 : 0, npieces = nsoa + (total + 4 * 256 - 1) / (4 * 256);
  for (int pc = 0;;) {
    const int need = pc < nsoa ? 2 * Q1_EAGER * 256 : 4 * 256;
    if (pc < npieces && *s_nsurv + need <= Q1_SURV_CAP) {  // (uniform)
      if (pc < nsoa) {
        const int c0 = Q1_EAGER * (pc + 1);
#pragma unroll
        for (int o = 0; o < 2; o++) {
          const int cnt = (int)(unsigned)hdr_r[o];
#pragma unroll
          for (int c = 0; c < 2 * Q1_EAGER; c++) rec[o][c] = cnt > c0 ? load_sc1(rec_q + (int64_t)(2 * c0 + c) * G + jj[o]) : 0;
        }
#pragma unroll
        for (int o = 0; o < 2; o++) {
          const int cnt = (int)(unsigned)hdr_r[o];
#pragma unroll
          for (int c = 0; c < Q1_EAGER; c++)
            if (c0 + c < cnt) keep(rec[o][2 * c], rec[o][2 * c + 1]);
        }
      } else {
        const int base = (pc - nsoa) * 4 * 256;
        const int lim = total - base < 4 * 256 ? total - base : 4 * 256;
        u64 ra[4], rb[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {  // four records per thread in flight
          const int e = tid + 256 * u;
          ra[u] = 0;
          rb[u] = 0;
          if (e < lim) {
            const int ge = base + e;
            int lo_j = 0, hi_j = G;  // the workgroup j with s_pref[j] <= ge < s_pref[j + 1]
            while (hi_j - lo_j > 1) {
              const int mid = (lo_j + hi_j) >> 1;
              if (s_pref[mid] <= ge) lo_j = mid;
              else hi_j = mid;
            }
            const u64 *r = a.ws_blk + ((int64_t)qi * G + lo_j) * a.blk + 2 * (ge - s_pref[lo_j]);
            ra[u] = load_sc1(r);
            rb[u] = load_sc1(r + 1);
          }
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
          if (tid + 256 * u < lim) keep(ra[u], rb[u]);
      }
      pc++;
      __syncthreads();
      continue;
    }
    rounds();
    if (pc >= npieces) break;
  }
  if (wave == 0 && lane < a.k) {
    rsx_sc_hit h;
    if (ld == INFINITY) {
      h.dist = kBig; h.index = 0; h.shift = 0;
    } else {
      h.dist = ld; h.index = li; h.shift = ls;
    }
    a.out[(int64_t)qi * a.k + lane] = h;
  }
  if (a.stats) {  // (profiling) same words as sc_rescore_wave_kernel
    unsigned long long *st = a.stats + (qi % RESCORE_STAT_COPIES) * RESCORE_STAT_WORDS;
    if (lane == 0) {
      atomicAdd(st + 2, (unsigned long long)n_exact);
      atomicAdd(st + 11, (unsigned long long)n_aligned);
      atomicAdd(st + 12, (unsigned long long)n_shifts);
    }
    atomicAdd(st + 3, (unsigned long long)n_cand_all);
    if (tid == 0) {
      atomicAdd(st, (unsigned long long)n_surv_all);
      atomicAdd(st + 1, 1ull);
#ifdef RSX_EXPERIMENTS
      Q1_MARK(8);
      for (int i = 0; i < 6; i++) atomicAdd(st + 4 + i, tmark[i + 1] - tmark[i]);
      atomicAdd(st + 10, tmark[8] - tmark[0]);
      atomicAdd(st + 13, tmark[7] - tmark[6]);
      atomicAdd(st + 14, tmark[8] - tmark[7]);
#endif
    }
  }
}

}  // namespace

// workgroups of one query's launch: as many as there are tiles, at most two per CU (the LDS footprint allows no more); the
// bound lists of all of them have to fit the last workgroup's LDS
// Several queries in one launch (gridDim.y): the device holds 512 workgroups at once, so each query gets 512 / nq of them
// (never fewer than 32) and its workgroups walk more tiles -- round 5 gave EVERY query up to 512, i.e. nq rounds of workgroups,
// each paying the 3 us of prep for one tile: 8 queries at 10 000 entries cost 55 us against 18 for one.
int q1_grid(int64_t n_items, int32_t k, int32_t nq) {
  const int64_t ntiles = (n_items + 31) / 32;
  const int kp = (k + 3) & ~3;
  int gmax = (Q1_FIN_AB_BYTES / (kp * 4) >= Q1_MAX_G) ? Q1_MAX_G : 256;  // whole CUs' worth: 512 up to k = 16, else 256
  if (nq > 1) {
    static const int slots = [] {
      const char *e = rsx::exp_env("RSX_Q1_SLOTS");
      return e ? atoi(e) : Q1_MAX_G;
    }();
    const int share = slots / nq < 32 ? 32 : slots / nq;
    gmax = gmax < share ? gmax : share;
  }
  return (int)(ntiles < 1 ? 1 : (ntiles < gmax ? ntiles : gmax));
}

static int64_t q1_block_u64(int64_t n_items, int g) {
  const int64_t ntiles = (n_items + 31) / 32;
  const int64_t ent = ((ntiles + g - 1) / g) * 32;  // entries of one workgroup = candidate records it may write
  return 2 * (ent > Q1_SOA ? ent - Q1_SOA : 1);
}

size_t q1_workspace_bytes(int64_t n_items, int32_t nq, int32_t k) {
  const int g = q1_grid(n_items, k, nq);
  return (size_t)nq * g * (1 + 16 + 2 * Q1_SOA + q1_block_u64(n_items, g)) * 8 + 256;
}

size_t q1_ticket_bytes() { return (size_t)Q1_MAX_NQ * Q1_TICKETS_PER_Q * Q1_TICKET_STRIDE * sizeof(unsigned); }

const char *q1_kernel_name() { return "sc_q1_kernel"; }

int launch_q1(const DbView &db, const float *d_q, int32_t nq, int64_t n_items, int64_t n_eligible, const int64_t *q_elig,
              int32_t k, rsx_sc_hit *d_out, void *ws, unsigned *d_ticket, unsigned long long *d_stats, hipStream_t s) {
  if (nq <= 0) return RSX_OK;
  if (nq > Q1_MAX_NQ) return fail(RSX_ERR_BAD_ARG, "the single-query path takes at most %d queries", Q1_MAX_NQ);
  if (k < 1 || k > RSX_SC_MAX_TOPK) return fail(RSX_ERR_BAD_ARG, "k=%d out of range [1,%d]", k, RSX_SC_MAX_TOPK);
  if (n_items < 0) n_items = 0;
  if (n_items >= (1ll << 31)) return fail(RSX_ERR_RANGE, "the single-query path addresses local slots with 31 bits");
  {  // 68 KiB of dynamic LDS: opt in once per device (and summation order: each is a kernel of its own)
    static std::atomic<unsigned long long> attr_set[4] = {};
    int dev = 0;
    RSX_HIP(hipGetDevice(&dev));
    const unsigned long long bit = 1ull << (dev & 63);
    const int so = (db.sum_order >= 0 && db.sum_order <= 3) ? db.sum_order : 0;
    if (!(attr_set[so].load(std::memory_order_relaxed) & bit)) {
      hipError_t e = hipSuccess;
      RSX_SO_DISPATCH(so, e = hipFuncSetAttribute(reinterpret_cast<const void *>(&sc_q1_kernel<SO>), hipFuncAttributeMaxDynamicSharedMemorySize, Q1_LDS));
      RSX_HIP(e);
      attr_set[so].fetch_or(bit, std::memory_order_relaxed);
    }
  }
  const int g = q1_grid(n_items, k, nq);
  Q1Args a;
  a.db = db;
  a.qdesc = d_q;
  a.n_items = n_items;
  {  // "every entry" (< 0) or anything past the end: the first global index no local slot reaches -- never INT64_MAX, the
     // kernel's (n_eligible - idx_base + idx_stride - 1) / idx_stride would overflow on a sharded handle (idx_stride > 1)
    const int64_t stride = db.idx_stride > 0 ? db.idx_stride : 1;
    const int64_t past_end = db.idx_base + n_items * stride;
    a.n_eligible = (n_eligible < 0 || n_eligible > past_end) ? past_end : n_eligible;
  }
  a.q_elig = q_elig;
  a.out = d_out;
  a.k = k;
  a.kp = (k + 3) & ~3;
  a.blk = (int32_t)q1_block_u64(n_items, g);
  a.ticket = d_ticket;
  a.ws_hdr = static_cast<u64 *>(ws);
  a.ws_ubs = a.ws_hdr + (size_t)nq * g;
  a.ws_rec = a.ws_ubs + (size_t)nq * 16 * g;
  a.ws_blk = a.ws_rec + (size_t)nq * 2 * Q1_SOA * g;
  a.stats = d_stats;
  RSX_SO_DISPATCH(db.sum_order, hipLaunchKernelGGL(sc_q1_kernel<SO>, dim3((unsigned)g, (unsigned)nq), dim3(256), Q1_LDS, s, a));
  RSX_HIP(hipGetLastError());
  return RSX_OK;
}

}  // namespace sc
}  // namespace rsx
