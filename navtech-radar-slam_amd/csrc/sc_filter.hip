// sc_filter.hip -- MFMA lower-bound filter of the exhaustive ScanContext search (gfx950 / CDNA4).
//
// Why it exists.  distanceBtnScanContext (SC.cpp:116-148) is the minimum of the column-cosine
// distance d_k (SC.cpp:69-90) over a 7-shift window chosen by the sector-key alignment
// (SC.cpp:93-113).  For ANY window
//        dist(query, entry)  >=  L(query, entry) := min over ALL 60 shifts k of d_k ,
// and the 60 values  S_k = sum_j cos(query column (j+k)%60, entry column j)  are one circular
// cross-correlation of the two column-normalised 20x60 images -- a GEMM with K = 1200 whose A
// operand is a circulant of the query:  S[k][n] = sum_i q2[i + 20k] * e_n[i].  That is the one place
// in this path where the matrix cores pay: 144 kflop per (query, entry) pair in fp16 on the MFMA
// pipe instead of ~44k fp64 lane-operations on the VALU.  The filter computes L~ (fp16 inputs, fp32
// accumulation) for every pair, the exact fp64 kernel (sc_pair_kernel) then re-scores only the
// entries whose bound can still reach the top-k:
//        keep entry  <=>  not ( L~ - eps > tau ),   tau = k-th best EXACT distance found so far.
// Because |L~ - L| <= eps (derivation below) and L <= dist, no entry of the exact top-k is ever
// dropped, so the results stay bit-identical to the oracle (tests/test_gpu_sc.py).
//
// Error bound eps (all quantities are means over n_eff effective columns, so bounds on one column
// pair carry over unchanged):
//   * operands are x^ = x/||column|| (fp64), scaled by 2^15 and rounded to fp16: relative error
//     u = 2^-11 per element (the scaling keeps every element that matters in the normal range:
//     fp16 subnormals start at 2^-29 relative to a unit column);
//     |cos~ - cos| <= (2u + u^2) * sum_r |q^_r e^_r| <= 2u + u^2 = 9.77e-4   (Cauchy-Schwarz)
//   * products of two fp16 values are exact in fp32; accumulating <= 1200 of them in fp32:
//     <= 1200 * 2^-23 relative to sum |terms| <= n_eff  (2^-23: no assumption on the MFMA's rounding
//     mode) = 1.43e-4
//   * epilogue (rcp, fma): < 1e-6;  fp64 roundings of the exact side: < 1e-12
//   total < 1.13e-3; kFilterEps = 1.25e-3.
// Non-finite descriptors (NaN/inf elements) are flagged in bit 63 of the column mask and always
// passed on to the exact kernel (L~ = -inf).
//
// Mapping (sc_filter_kernel, one wave per 32 database entries, 4 waves = 128 entries per block):
//   * B operand (entries): the wave's 32 entries x 1200 fp16 stay in 300 VGPRs for the whole kernel
//     (75 K-steps x v_mfma_f32_32x32x16_f16 B fragments; the DB image is stored tile-major so every
//     fragment load is one coalesced 1 KiB read).  One wave per SIMD, 512-register budget.
//   * A operand (query shifts): row k of the circulant is the query image read at element offset
//     20k, so a lane fetches its A fragment with ONE ds_read_b128 from a doubled query image in LDS
//     (two copies, the second displaced by 4 elements, make the read 16-byte aligned for odd k; the
//     copies sit 4960 B (= 96 mod 256) apart, the one displacement that makes the access bank-conflict free).  Shifts 32..63 (tile 1) at
//     K-step s need exactly the fragment of shifts 0..31 (tile 0) at step s+40, so each query costs
//     115 LDS reads for 150 MFMAs.
//   * queries stream through LDS in phases of 4 (double buffered, global_load_lds DMA, one
//     s_barrier per phase).
//   * epilogue per (query, entry): n_eff(k) = popcount(rot60(query mask, k) & entry mask) from two
//     64-bit column masks, d_k = 1 - S_k / n_eff(k), min over the lane's 32 shifts, one cross-half
//     exchange, one float per entry written (128 B per wave per query).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "rsx_common.h"
#include "sc_kernels.h"
#include "sc_entry_dev.h"

namespace rsx {
namespace sc {

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef unsigned frag4 __attribute__((ext_vector_type(4)));  // one MFMA A fragment in flight (8 x fp16)
typedef unsigned long long u64;

#define AS1 __attribute__((address_space(1)))
#define AS3 __attribute__((address_space(3)))

constexpr int F_STEPS = DS / 16;  // 75 K-steps of 16
constexpr int F_TILE1 = 40;       // 32 shifts * 20 elements = 640 = 40 K-steps
constexpr int F_T = F_STEPS + F_TILE1;  // 115 A fragments per query
constexpr int QIMG_ODD = 4960;    // byte offset of the copy read by odd shifts (holds q2[4..]); = 96 mod 256
constexpr int QIMG_EVEN_CHUNKS = 308;  // 2464 elements
constexpr int F_QPP = 4;          // queries per LDS phase
constexpr int F_DEPTH = 5;        // A fragments in flight
constexpr int F_B_VGPR = 44;      // B fragments kept in VGPRs; the rest live in AGPRs
constexpr int F_PHASE_BYTES = F_QPP * FILTER_QIMG_BYTES;  // 39936 = 39 KiB
using dev::kImgScale;  // 2^15 on both operands
using dev::normalise_column;
using dev::wave_lds_fence;
constexpr float kAccScale = 1073741824.0f;            // 2^30 carried by the accumulators
constexpr u64 kNonFinite = 1ull << 63;

static_assert(FILTER_QIMG_BYTES == 9984, "layout");
static_assert(QIMG_ODD == FILTER_QIMG_ODD && QIMG_EVEN_CHUNKS * 16 == FILTER_QIMG_MASK_OFF && kAccScale == FILTER_ACC_SCALE,
              "sc_window.hip reads the same query image");
static_assert(F_PHASE_BYTES % 1024 == 0, "phase must be whole 1 KiB DMA pieces");

// ------------------------------------------------------------------------------------------
// database image: fp16, tile-major [tile of 32 entries][75 K-steps][64 lanes][8 halves]
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sc_img_db_kernel(const float *__restrict__ desc,
                                                        const double *__restrict__ norm, int64_t first,
                                                        int64_t count, uint4 *__restrict__ hnT,
                                                        uint4 *__restrict__ hnR, u64 *__restrict__ cmask) {
  __shared__ __attribute__((aligned(16))) _Float16 st[4][DS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t it = (int64_t)blockIdx.x * 4 + wave;
  if (it >= count) return;
  dev::img_db_entry(desc, norm, first + it, st[wave], hnT, hnR, cmask, lane);
}

// ------------------------------------------------------------------------------------------
// query image: the LDS layout of the filter kernel, 9984 B per query
//   [0, 4928)     q2[0..2464)      (q2[e] = q^[e mod 1200]), read by even shifts
//   [4928, 4936)  the query's column mask (bit j = column j non-zero, bit 63 = non-finite element)
//   [4936, 4960)  zero
//   [4960, 9984)  q2[4..2516)      read by odd shifts
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sc_img_query_kernel(const float *__restrict__ desc,
                                                           const double *__restrict__ norm, int32_t nq,
                                                           char *__restrict__ qimg) {
  __shared__ __attribute__((aligned(16))) _Float16 st[4][DS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = blockIdx.x * 4 + wave;
  if (q >= nq) return;
  bool nonzero = false, bad = false;
  if (lane < NS) normalise_column(desc + (int64_t)q * DS + lane * NR, norm[(int64_t)q * NS + lane], &st[wave][lane * NR], nonzero, bad);
  u64 m = __ballot(nonzero && lane < NS);
  if (__ballot(bad && lane < NS)) m |= kNonFinite;
  wave_lds_fence();
  dev::img_query_image(st[wave], m, reinterpret_cast<uint4 *>(qimg + (int64_t)q * FILTER_QIMG_BYTES), lane);
}

// ------------------------------------------------------------------------------------------
// the filter
// ------------------------------------------------------------------------------------------
struct FilterArgs {
  const uint4 *hnT;
  const u64 *cmask;
  const char *qimg;
  int64_t n_items;
  int32_t nq;
  int64_t per_block;  // (tile-block, query) work items per workgroup (no plan)
  lb_t *lb;
  int64_t ld_lb;
  // optional plan (queries whose eligibility grows with the query index, e.g. every keyframe against the
  // keyframes older than itself): tile-block tb only matters to queries >= tb_qmin[tb];
  // tb_cum[tb] = work items before tb, tb_cum[ntb] = total
  const int32_t *tb_qmin;
  const int64_t *tb_cum;
};

__device__ __forceinline__ void stage_queries(const char *gsrc, char *ldst, int nbytes, int wave, int lane) {
  const int npieces = (nbytes + 1023) >> 10;
  for (int c = wave; c < npieces; c += 4)
    __builtin_amdgcn_global_load_lds(reinterpret_cast<const AS1 void *>(reinterpret_cast<uintptr_t>(gsrc + c * 1024 + lane * 16)),
                                     (AS3 void *)(ldst + c * 1024), 16, 0, 0);
}

// per-lane state of the epilogue of one (query, 32-entry tile): the query's 60-bit column mask,
// doubled and pre-shifted by 4 * (lane >> 5), as four 32-bit words; two running maxima
struct Epi {
  unsigned w[4];
  float m0, m1;
};

__device__ __forceinline__ void epi_begin(Epi &e, u64 qm, int hh) {
  const u64 m1c = qm & ~kNonFinite;
  const u64 lo = m1c | (m1c << 60), hi = m1c >> 4;  // the 60-bit mask twice in a row (120 bits)
  const u64 lo4 = (lo >> 4) | (hi << 60), hi4 = hi >> 4;
  const u64 l = hh ? lo4 : lo, h = hh ? hi4 : hi;
  e.w[0] = (unsigned)l;
  e.w[1] = (unsigned)(l >> 32);
  e.w[2] = (unsigned)h;
  e.w[3] = (unsigned)(h >> 32);
  e.m0 = -INFINITY;
  e.m1 = -INFINITY;
}

// one output of the 32x32 C/D tile: register r of tile tl is shift k = 32*tl + (r&3) + 8*(r>>2) + 4*(lane>>5).
// value = S_k / n_eff(k); n_eff == 0 => S == 0 exactly and rcp = inf: NaN, which fmaxf drops (that
// shift has no effective column: SC.cpp:87-88 gives NaN, never the minimum)
template <int TL, int R>
__device__ __forceinline__ void epi_piece(Epi &e, float S, unsigned m2lo, unsigned m2hi, int hh) {
  constexpr int b = (R & 3) + 8 * (R >> 2);
  const unsigned rlo = __builtin_amdgcn_alignbit(e.w[TL + 1], e.w[TL], b);
  const unsigned rhi = __builtin_amdgcn_alignbit(e.w[TL + 2], e.w[TL + 1], b);
  const int ne = __builtin_popcount(rlo & m2lo) + __builtin_popcount(rhi & m2hi);
  float v = S * __builtin_amdgcn_rcpf((float)ne);
  if (TL == 1 && R >= 12) v = hh ? -INFINITY : v;  // rows 60..63 are padding
  if (R & 1) e.m1 = fmaxf(e.m1, v);
  else e.m0 = fmaxf(e.m0, v);
}

// lower bound of this lane's entry for the finished query: 1 - max_k S_k / n_eff(k) / 2^30
__device__ __forceinline__ float epi_end(const Epi &e) {
  float m = fmaxf(e.m0, e.m1);
  // lanes l and l ^ 32 hold the two halves of the shifts of one entry: v_permlane32_swap of (m, m)
  // yields {m[l & 31]} and {m[(l & 31) + 32]} in every lane
  const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
  m = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
  return fmaf(m, -1.0f / kAccScale, 1.0f);  // m == -inf (no effective column at any shift) -> +inf
}

// One software-pipeline stage: the 150 MFMAs of the current query (DO_MFMA) interleaved with the
// epilogue of the previous one (DO_EPI) whose accumulators were parked in p0/p1.  The
// sched_group_barriers pin the issue order: per K-step one LDS read, the 1-2 MFMAs that consume
// the fragment read F_DEPTH steps earlier, and a few VALU instructions of the epilogue.
// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N-1>{})
template <typename F, int... Is>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {
  static_for_impl(static_cast<F &&>(f), std::make_integer_sequence<int, N>{});
}

// A-fragment reads are issued through inline asm with hand-counted s_waitcnt: LDS returns data in
// issue order, so "the fragment of step t has landed" == "at most min(F_DEPTH-1, F_T-t-1) younger
// reads are outstanding".  (Left to the compiler the waits degrade to lgkmcnt(0) after every few
// reads, which parks the wave for a full LDS round trip ~40 times per query.)  The wait takes the
// fragment as an in/out operand so that the consuming MFMA cannot be scheduled above it.
__device__ __forceinline__ void lds_read_frag(frag4 &dst, unsigned addr, int off) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off));
}
template <int N>
__device__ __forceinline__ void lds_wait_frag(frag4 &frag) {
  asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(frag) : "n"(N));
}
// the same without operands (the "+v" form makes the compiler treat the fragment as freshly written
// by a VALU instruction and pad every MFMA with an s_nop): the caller orders it against the consuming
// MFMA with __builtin_amdgcn_sched_barrier(0) on both sides
template <int N>
__device__ __forceinline__ void lds_wait_count() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N));
}

// Fragment schedule of one stage (all reads are asynchronous; LDS returns them in issue order):
//   fragments 0..F_DEPTH-1      come from `first[]`, read by the PREVIOUS stage at its steps
//                                F_PRE..F_PRE+F_DEPTH-1 (or by first_prologue at the start of a phase)
//   fragment  t >= F_DEPTH       is read into ring[t % F_DEPTH] right after the MFMAs of step t-F_DEPTH
// so nothing is in flight when a stage ends: the ring is consumed and `first` (read >= 15 steps before
// the end) has landed -- the compiler may copy those registers at the loop edge.
constexpr int F_PRE = 95;  // first step that prefetches the next query's first fragments

// reads issued after fragment t's own read and before the wait of step t (t >= F_DEPTH)
constexpr int younger_reads(int t) {
  int n = 0;
  for (int s = t - F_DEPTH; s < t; s++) {
    if (s > t - F_DEPTH && s + F_DEPTH < F_T) n++;          // ring read of step s (younger than ours)
    if (s >= F_PRE && s < F_PRE + F_DEPTH) n++;              // prefetch read of step s (issued after the ring read)
  }
  return n;
}

__device__ __forceinline__ void first_prologue(frag4 (&first)[F_DEPTH], unsigned ap_lds) {
  static_for<F_DEPTH>([&](auto tc) {
    constexpr int t = decltype(tc)::value;
    lds_read_frag(first[t], ap_lds, 32 * t);
  });
  static_for<F_DEPTH>([&](auto tc) {
    constexpr int i = decltype(tc)::value;
    lds_wait_frag<0>(first[i]);
  });
}

// read one accumulator register into a VGPR exactly HERE.  Written as `x = acc[i]` the optimiser hoists
// all 32 reads to the top of the loop body, where they wait for the last MFMA of the previous query.
__device__ __forceinline__ float acc_read(float a) {
  float v;
  asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(a));
  return v;
}

// which accumulator register (0..15) is read out at step t of its 40-step idle window (-1: none)
constexpr int piece_at(int t) {
  for (int i = 0; i < 16; i++)
    if ((i * 5) / 2 == t) return i;
  return -1;
}

// One software-pipeline stage = one query.  Tile 0 (shifts 0..31, acc0) accumulates during steps
// 0..74 and tile 1 (shifts 32..63, acc1) during steps 40..114, so each accumulator is idle for 40
// steps per query.  The matrix pipe hides about five other instructions per MFMA, so the epilogue is
// placed where the MFMAs are dense:
//   steps   0..39   (1 MFMA/step) read the PREVIOUS query's tile 1 out of acc1 (16 v_accvgpr_read)
//   steps  40..74   (2 MFMA/step) all the epilogue arithmetic of the previous query (32 outputs x ~9
//                   VALU), then `fin` (store)
//   steps  75..114  (1 MFMA/step) read THIS query's tile 0 out of acc0 into p0 for the next stage
// No accumulator is copied besides those reads, and the first fragments of the next query (at
// next_lds: the next query of the LDS phase, or any valid image when there is none) are prefetched,
// so MFMAs issue back to back across queries.
template <bool DO_MFMA, bool DO_PREV, typename Fin>
__device__ __forceinline__ void filter_stage(unsigned ap_lds, unsigned next_lds, frag4 (&first)[F_DEPTH],
                                             const half8 (&B)[F_STEPS], floatx16 &acc0, floatx16 &acc1,
                                             floatx16 &p0, u64 prev_mask, unsigned m2lo, unsigned m2hi, int hh,
                                             Fin &&fin) {
  frag4 ring[F_DEPTH];
  floatx16 p1;
  Epi e;
  static_for<F_T>([&](auto tc) {
    constexpr int t = decltype(tc)::value;
    if constexpr (DO_MFMA) {
      half8 af;
      if constexpr (t < F_DEPTH) {
        af = __builtin_bit_cast(half8, first[t]);
      } else {
        lds_wait_count<younger_reads(t)>();
        __builtin_amdgcn_sched_barrier(0);  // the MFMA below stays below the wait
        af = __builtin_bit_cast(half8, ring[t % F_DEPTH]);
      }
      if constexpr (t == 0) {
        floatx16 z;
#pragma unroll
        for (int i = 0; i < 16; i++) z[i] = 0.0f;
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, B[0], z, 0, 0, 0);
      } else if constexpr (t < F_STEPS) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, B[t], acc0, 0, 0, 0);
      }
      if constexpr (t == F_TILE1) {
        floatx16 z;
#pragma unroll
        for (int i = 0; i < 16; i++) z[i] = 0.0f;
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, B[0], z, 0, 0, 0);
      } else if constexpr (t > F_TILE1) {
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, B[t - F_TILE1], acc1, 0, 0, 0);
      }
      if constexpr (t + F_DEPTH < F_T) lds_read_frag(ring[t % F_DEPTH], ap_lds, 32 * (t + F_DEPTH));
      if constexpr (t >= F_PRE && t < F_PRE + F_DEPTH) lds_read_frag(first[t - F_PRE], next_lds, 32 * (t - F_PRE));
    }
    if constexpr (DO_PREV) {
      if constexpr (t < F_TILE1) {  // acc1 still holds the previous query's tile 1
        constexpr int i = piece_at(t);
        if constexpr (i >= 0) p1[i] = acc_read(acc1[i]);
      } else if constexpr (t < F_STEPS) {  // 35 dense steps: 32 outputs, then the store
        if constexpr (t == F_TILE1) epi_begin(e, prev_mask, hh);
        constexpr int i = t - F_TILE1;
        if constexpr (i < 16) epi_piece<0, i & 15>(e, p0[i & 15], m2lo, m2hi, hh);
        else if constexpr (i < 32) epi_piece<1, i & 15>(e, p1[i & 15], m2lo, m2hi, hh);
        if constexpr (t == F_STEPS - 1) fin(e);
      }
    }
    if constexpr (DO_MFMA && t >= F_STEPS) {  // acc0 is complete: park it for the next stage
      constexpr int i = piece_at(t - F_STEPS);
      if constexpr (i >= 0) p0[i] = acc_read(acc0[i]);
    }
    // nothing moves across a step boundary: the compiler would otherwise hoist the (asm) reads of
    // several steps above the MFMAs that should cover their latency, and wait right after issuing
    __builtin_amdgcn_sched_barrier(0);
  });
}

constexpr int QIMG_MASK_OFF = QIMG_EVEN_CHUNKS * 16;  // 4928

// The (tile-block, query) work items -- tile-block = 4 tiles = 128 entries, linear index
// tb * nq + q -- are cut into equal contiguous segments, one per workgroup (one workgroup per CU,
// one wave per SIMD), so every CU gets the same number of MFMAs; a segment touches at most two
// tile-blocks (one extra 300-register B load).
__global__ __launch_bounds__(256, 1) void sc_filter_kernel(FilterArgs a) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 31, hh = lane >> 5;
  const int64_t ntiles = (a.n_items + 31) >> 5;
  const int64_t ntb = (ntiles + 3) >> 2;
  const int64_t total = a.tb_cum ? a.tb_cum[ntb] : ntb * (int64_t)a.nq;
  const int64_t per = a.tb_cum ? (total + gridDim.x - 1) / gridDim.x : a.per_block;
  int64_t L0 = (int64_t)blockIdx.x * per;
  const int64_t L1 = (L0 + per < total) ? (L0 + per) : total;
  // A fragment address of this lane inside a query image (row = shift col of tile 0)
  const int aoff = ((col & 1) ? (QIMG_ODD + 40 * col - 8) : (40 * col)) + 16 * hh;
  const unsigned lds_base = (unsigned)(uintptr_t)((AS3 char *)smem);

  int64_t tb = 0;
  if (a.tb_cum && L0 < L1) {  // last tile-block whose first item is <= L0
    int64_t lo = 0, hi = ntb - 1;
    while (lo < hi) {
      const int64_t mid = (lo + hi + 1) >> 1;
      if (a.tb_cum[mid] <= L0) lo = mid;
      else hi = mid - 1;
    }
    tb = lo;
  }
  while (L0 < L1) {
    int q0, qend;
    if (a.tb_cum) {
      while (a.tb_cum[tb + 1] <= L0) tb++;  // skip tile-blocks without items
      q0 = a.tb_qmin[tb] + (int)(L0 - a.tb_cum[tb]);
      qend = a.nq;
    } else {
      tb = L0 / a.nq;
      q0 = (int)(L0 - tb * a.nq);
      qend = a.nq;
    }
    const int q1 = (L1 - L0 < (int64_t)(qend - q0)) ? (int)(q0 + (L1 - L0)) : qend;
    L0 += q1 - q0;
    const int64_t tile = tb * 4 + wave;
    const bool tile_ok = tile < ntiles;  // wave-uniform
    const int64_t n = tile * 32 + col;
    const bool n_ok = tile_ok && n < a.n_items;
    const int nphase = (q1 - q0 + F_QPP - 1) / F_QPP;

    // phase 0 of the query stream (DMA, overlaps the B loads below)
    {
      const int nqs = (q1 - q0 < F_QPP) ? (q1 - q0) : F_QPP;
      stage_queries(a.qimg + (int64_t)q0 * FILTER_QIMG_BYTES, smem, nqs * FILTER_QIMG_BYTES, wave, lane);
    }

    // B operand: 32 entries x 1200 fp16, register-resident for the whole segment
    half8 B[F_STEPS];
    {
      const uint4 *src = a.hnT + ((tile_ok ? tile : 0) * F_STEPS) * 64 + lane;
#pragma unroll
      for (int s = 0; s < F_STEPS; s++) {
        const uint4 v = src[s * 64];
        B[s] = *reinterpret_cast<const half8 *>(&v);
      }
      // 300 registers of B do not fit the 256 architectural VGPRs: give the tail fragments an AGPR
      // register class up front, otherwise the allocator treats AGPRs as spill space and re-copies
      // ~110 registers per query
#pragma unroll
      for (int s = F_B_VGPR; s < F_STEPS; s++) asm volatile("" : "+a"(B[s]));
    }
    const u64 m2 = n_ok ? a.cmask[n] : 0ull;
    const unsigned m2lo = (unsigned)m2, m2hi = (unsigned)(m2 >> 32) & 0x0fffffffu;

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    floatx16 acc0, acc1, p0;  // p0: tile 0 of the previous query, read out of acc0 at the end of its stage
    frag4 first[F_DEPTH];     // fragments 0..F_DEPTH-1 of the query about to be processed (landed)
    int prev_q = -1;
    u64 prev_mask = 0;
    // finish one (query, tile): non-finite flag, store (128 B per wave); eligibility is applied by
    // the selection kernels, not here
    auto fin = [&](const Epi &ep) {
      float best = epi_end(ep);
      if ((prev_mask | m2) & kNonFinite) best = -INFINITY;  // non-finite input: always re-score exactly
      if (n_ok && hh == 0) a.lb[(int64_t)prev_q * a.ld_lb + n] = lb_pack(best);
    };
    for (int p = 0; p < nphase; p++) {
      const int qp = q0 + p * F_QPP;
      if (p + 1 < nphase) {
        const int qn = qp + F_QPP;
        const int nqs = (q1 - qn < F_QPP) ? (q1 - qn) : F_QPP;
        stage_queries(a.qimg + (int64_t)qn * FILTER_QIMG_BYTES, smem + ((p + 1) & 1) * F_PHASE_BYTES,
                      nqs * FILTER_QIMG_BYTES, wave, lane);
      }
      const int nq_here = (q1 - qp < F_QPP) ? (q1 - qp) : F_QPP;
      if (tile_ok) {
        const unsigned phase_lds = lds_base + (unsigned)((p & 1) * F_PHASE_BYTES + aoff);
        first_prologue(first, phase_lds);
        for (int qq = 0; qq < nq_here; qq++) {
          const unsigned ap_lds = phase_lds + (unsigned)(qq * FILTER_QIMG_BYTES);
          const unsigned next_lds = (qq + 1 < nq_here) ? ap_lds + FILTER_QIMG_BYTES : ap_lds;
          const u64 cur_mask = *reinterpret_cast<const u64 *>(smem + (p & 1) * F_PHASE_BYTES + qq * FILTER_QIMG_BYTES + QIMG_MASK_OFF);
          if (prev_q < 0) filter_stage<true, false>(ap_lds, next_lds, first, B, acc0, acc1, p0, prev_mask, m2lo, m2hi, hh, fin);
          else filter_stage<true, true>(ap_lds, next_lds, first, B, acc0, acc1, p0, prev_mask, m2lo, m2hi, hh, fin);
          prev_q = qp + qq;
          prev_mask = cur_mask;
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    if (prev_q >= 0) filter_stage<false, true>(0u, 0u, first, B, acc0, acc1, p0, prev_mask, m2lo, m2hi, hh, fin);
  }
}

// ------------------------------------------------------------------------------------------
// bound histogram: 2048 bins over [0, 1) (bounds outside land in the end bins).  bin(x) <= b  <=>
// x < (b+1)/2048 exactly (power-of-two scaling), so a "bound < edge" test selects whole bins.
// ------------------------------------------------------------------------------------------
constexpr int H_BINS = 2048;
constexpr int SEL_U = 5;  // 16-byte loads in flight per thread of sc_select_kernel (a 10 016-entry row of fp16 bounds = 1 piece)

__device__ __forceinline__ int lb_bin(float d) {
  if (!(d > 0.0f)) return 0;  // negative, -inf, NaN
  const float x = d * (float)H_BINS;
  return x >= (float)(H_BINS - 1) ? H_BINS - 1 : (int)x;
}

// entries [0, n_elig_items(q)) of a row are eligible for query q: local slot i has global index
// idx_base + i * idx_stride, eligible iff that is < min(n_eligible, q_elig[q])
struct Elig {
  int64_t idx_base, idx_stride, n_eligible;
  const int64_t *q_elig;
};
__device__ __forceinline__ int64_t n_elig_items(const Elig &el, int q, int64_t n_items) {
  int64_t lim = el.n_eligible;
  if (el.q_elig) {
    const int64_t v = el.q_elig[q];
    lim = v < lim ? v : lim;
  }
  if (lim <= el.idx_base) return 0;
  const int64_t c = (lim - el.idx_base + el.idx_stride - 1) / el.idx_stride;
  return c < n_items ? c : n_items;
}

// ------------------------------------------------------------------------------------------
// short list + round edges of one query (feeds sc_rescore_kernel): histogram of the eligible bounds,
// prefix sum, then
//   t_cap  = edge of the last bin b_cap whose cumulative count still fits RESCORE_SHORTLIST_CAP
//   t_r    = edge of the first bin with at least target[r] bounds at or below it, clamped to t_cap
//   slist  = (bound, slot) of every eligible entry in bins <= b_cap  (bound < t_cap)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float bin_edge(int b) {  // upper edge of bin b as a "bound < edge" test
  return b < 0 ? -INFINITY : (b >= H_BINS - 1 ? INFINITY : (float)(b + 1) / (float)H_BINS);
}

__global__ __launch_bounds__(256) void sc_select_kernel(const lb_t *__restrict__ lb, int64_t ld, int64_t n_items_all,
                                                        Elig el, int32_t first_target,
                                                        RescoreEntry *__restrict__ slist,
                                                        int32_t *__restrict__ sl_cnt, float *__restrict__ thr) {
  __shared__ int hist[H_BINS];
  __shared__ int wsum[4];
  __shared__ int s_bcap;
  __shared__ int s_total;
  const int q = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const lb_t *row = lb + (int64_t)q * ld;
  const int64_t n_items = n_elig_items(el, q, n_items_all);
  for (int i = threadIdx.x; i < H_BINS; i += 256) hist[i] = 0;
  if (threadIdx.x == 0) s_total = 0;
  __syncthreads();
  // the row is read in pieces of 256 threads x SEL_U 16-byte loads (8 fp16 bounds each): all SEL_U loads of a thread are in
  // flight together (one 4-byte load per thread and iteration left 8 KB per CU in flight: latency-bound at 1.5 TB/s,
  // 0.22 ms per 8192 rows)
  // A row that fits one piece (<= 256 x SEL_U x 8 = 10 240 entries: the 10 000-keyframe DB) is read ONCE: the registers of
  // the histogram pass serve the compaction pass as well.  (Round 3 read it twice, the second time "from the L2" -- 8192 rows
  // of 20 KB are 0.16 GB, far more than the L2s hold: both readings came over the fabric.)
  typedef _Float16 half8 __attribute__((ext_vector_type(8)));
  const uint4 *row8 = reinterpret_cast<const uint4 *>(row);  // ld is a multiple of 32 elements: 64-byte aligned rows
  const int64_t n8 = n_items >> 3;
  const bool one_piece = n8 <= 256 * SEL_U;
  uint4 keep[SEL_U];
  auto load_piece = [&](int64_t c0, uint4 (&x)[SEL_U]) {
#pragma unroll
    for (int u = 0; u < SEL_U; u++) {
      const int64_t j = c0 + u * 256 + threadIdx.x;
      x[u] = j < n8 ? row8[j] : uint4{0x7c007c00u, 0x7c007c00u, 0x7c007c00u, 0x7c007c00u};  // +inf
    }
  };
  auto walk_piece = [&](int64_t c0, const uint4 (&x)[SEL_U], auto &&f) {
#pragma unroll
    for (int u = 0; u < SEL_U; u++) {
      const int64_t i = (c0 + u * 256 + threadIdx.x) << 3;
      const half8 h = __builtin_bit_cast(half8, x[u]);
#pragma unroll
      for (int e = 0; e < 8; e++) f((float)h[e], i + e);
    }
  };
  // first = true: the pass that loads; a later pass of a one-piece row walks the registers it left behind
  auto for_row = [&](bool first, auto &&f) {
    if (one_piece) {
      if (first) load_piece(0, keep);
      walk_piece(0, keep, f);
    } else {
      for (int64_t c0 = 0; c0 < n8; c0 += 256 * SEL_U) {
        uint4 x[SEL_U];
        load_piece(c0, x);
        walk_piece(c0, x, f);
      }
    }
    const int64_t i = (n8 << 3) + threadIdx.x;  // the last n_items % 8 entries
    if (i < n_items) f((float)row[i], i);
  };
  for_row(true, [&](float d, int64_t) {
    if (d != INFINITY) atomicAdd(&hist[lb_bin(d)], 1);  // +inf: no effective column at any shift, never a hit
  });
  __syncthreads();
  int v[8];
  int run = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    run += hist[threadIdx.x * 8 + i];
    v[i] = run;
  }
  int incl = run;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int o = __shfl_up(incl, off);
    if (lane >= off) incl += o;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  int base = incl - run;
  for (int w = 0; w < wave; w++) base += wsum[w];
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 8; i++) hist[threadIdx.x * 8 + i] = base + v[i];  // inclusive cumulative counts
  __syncthreads();
  // first bin whose cumulative count reaches `target` (H_BINS when none)
  auto first_reaching = [&](int target) {
    int lo = 0, hi = H_BINS;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (hist[mid] >= target) hi = mid;
      else lo = mid + 1;
    }
    return lo;
  };
  if (threadIdx.x == 0) s_bcap = first_reaching(RESCORE_SHORTLIST_CAP + 1) - 1;  // last bin with cum <= CAP
  __syncthreads();
  const int b_cap = s_bcap;
  if (threadIdx.x < RESCORE_NUM_THR) {
    int b = b_cap;
    if (threadIdx.x < RESCORE_NUM_THR - 1) {
      b = first_reaching(first_target << threadIdx.x);
      if (b > H_BINS - 1) b = H_BINS - 1;
      if (b > b_cap) b = b_cap;
    }
    thr[(int64_t)q * RESCORE_THR_STRIDE + threadIdx.x] = bin_edge(b);
    reinterpret_cast<int32_t *>(thr)[(int64_t)q * RESCORE_THR_STRIDE + RESCORE_NUM_THR + threadIdx.x] = b >= 0 ? hist[b] : 0;
  }
  // compaction of bins <= b_cap, ORDERED BY BIN (counting sort: entry of bin b goes to [cum[b-1], cum[b]),
  // any order inside a bin): sc_walk_kernel walks the list in ascending-bound order, sc_rescore_kernel
  // does not care
  RescoreEntry *out = slist + (int64_t)q * RESCORE_SHORTLIST_CAP;
  __shared__ int fill[H_BINS];
  for (int i = threadIdx.x; i < H_BINS; i += 256) fill[i] = i ? hist[i - 1] : 0;  // exclusive prefix = first position
  __syncthreads();
  if (b_cap >= 0) {
    for_row(false, [&](float d, int64_t i) {  // second pass: the kept registers (or, for a long row, a second reading)
      if (d == INFINITY) return;
      const int b = lb_bin(d);
      if (b > b_cap) return;
      RescoreEntry e;
      e.lb = d;
      e.slot = (int32_t)i;
      out[atomicAdd(&fill[b], 1)] = e;
    });
  }
  if (threadIdx.x == 0) s_total = b_cap >= 0 ? hist[b_cap] : 0;
  __syncthreads();
  if (threadIdx.x == 0) sl_cnt[q] = s_total;
}

// ------------------------------------------------------------------------------------------
// filter plan for queries whose eligibility limit does not decrease with the query index: the
// tile-block tb (128 entries, first global index g0) is invisible to every query q with limit(q) <= g0,
// and those queries form a prefix [0, tb_qmin[tb]).  tb_cum = exclusive prefix sums of nq - tb_qmin.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void sc_filter_plan_kernel(Elig el, int32_t nq, int64_t n_items, int32_t qgroup,
                                                              int32_t *__restrict__ tb_qmin, int64_t *__restrict__ tb_cum) {
  __shared__ long long sh[1024];
  const int64_t ntb = (((n_items + 31) >> 5) + 3) >> 2;
  const int t = threadIdx.x;
  long long running = 0;
  for (int64_t base = 0; base < ntb; base += 1024) {
    const int64_t tb = base + t;
    long long items = 0;
    if (tb < ntb) {
      const int64_t g0 = el.idx_base + tb * 128 * el.idx_stride;
      int lo = 0, hi = nq;  // first q whose limit exceeds g0
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        int64_t lim = el.n_eligible;
        if (el.q_elig) {
          const int64_t v = el.q_elig[mid];
          lim = v < lim ? v : lim;
        }
        if (lim > g0) hi = mid;
        else lo = mid + 1;
      }
      tb_qmin[tb] = lo / qgroup;  // in units of qgroup queries (a group that straddles the limit is kept whole)
      items = (nq + qgroup - 1) / qgroup - lo / qgroup;
    }
    sh[t] = items;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
      const long long v = t >= d ? sh[t - d] : 0;
      __syncthreads();
      sh[t] += v;
      __syncthreads();
    }
    if (tb < ntb) tb_cum[tb] = running + sh[t] - items;
    running += sh[1023];
    __syncthreads();
  }
  if (t == 0) tb_cum[ntb] = running;
}

}  // namespace

double filter_eps() {
  // RSX_SC_FILTER_EPS can only LOOSEN the bound (experiments on how the exact stage grows with eps)
  static const double eps = [] {
    const char *e = rsx::exp_env("RSX_SC_FILTER_EPS");
    const double v = e ? atof(e) : 0.0;
    return v > 1.25e-3 ? v : 1.25e-3;
  }();
  return eps;
}

size_t filter_qimg_bytes(int32_t nq) { return (size_t)nq * FILTER_QIMG_BYTES + 1024; }

int launch_db_images(const float *desc, const double *norm, int64_t first, int64_t count, void *hnT, void *hnR,
                     uint64_t *cmask, hipStream_t s) {
  if (count <= 0) return RSX_OK;
  hipLaunchKernelGGL(sc_img_db_kernel, dim3((unsigned)((count + 3) / 4)), dim3(256), 0, s, desc, norm, first, count,
                     static_cast<uint4 *>(hnT), static_cast<uint4 *>(hnR), reinterpret_cast<u64 *>(cmask));
  RSX_HIP(hipGetLastError());
  return RSX_OK;
}

int launch_query_images(const float *desc, const double *norm, int32_t nq, void *qimg, hipStream_t s) {
  if (nq <= 0) return RSX_OK;
  hipLaunchKernelGGL(sc_img_query_kernel, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, s, desc, norm, nq,
                     static_cast<char *>(qimg));
  RSX_HIP(hipGetLastError());
  return RSX_OK;
}

const char *filter_kernel_name() { return "sc_filter_kernel"; }

size_t filter_plan_bytes(int64_t n_items) {
  const int64_t ntb = (((n_items + 31) / 32) + 3) / 4;
  return (size_t)(ntb + 1) * sizeof(int64_t) + (size_t)ntb * sizeof(int32_t) + 64;
}

int launch_filter_plan(const DbView &db, const FilterPlanInput &plan, int32_t nq, int64_t n_items, int32_t qgroup,
                       void *plan_ws, const int32_t **tb_qmin, const int64_t **tb_cum, hipStream_t s) {
  const int64_t ntb = (((n_items + 31) / 32) + 3) / 4;
  int64_t *cum = static_cast<int64_t *>(plan_ws);
  int32_t *qmin = reinterpret_cast<int32_t *>(cum + ntb + 1);
  const Elig el{db.idx_base, db.idx_stride, plan.n_eligible < 0 ? INT64_MAX : plan.n_eligible, plan.q_elig};
  hipLaunchKernelGGL(sc_filter_plan_kernel, dim3(1), dim3(1024), 0, s, el, nq, n_items, qgroup, qmin, cum);
  RSX_HIP(hipGetLastError());
  *tb_qmin = qmin;
  *tb_cum = cum;
  return RSX_OK;
}

int launch_filter(const DbView &db, const void *qimg, int32_t nq, int64_t n_items, lb_t *lb, int64_t ld_lb,
                  const FilterPlanInput *plan, void *plan_ws, hipStream_t s) {
  if (nq <= 0 || n_items <= 0) return RSX_OK;
  static int n_cu = 0;
  const int lds = 2 * F_PHASE_BYTES;
  if (!n_cu) {
    RSX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&sc_filter_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    int dev = 0, cu = 0;
    RSX_HIP(hipGetDevice(&dev));
    RSX_HIP(hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev));
    n_cu = cu > 0 ? cu : 256;
  }
  FilterArgs a;
  a.hnT = static_cast<const uint4 *>(db.hnT);
  a.cmask = reinterpret_cast<const u64 *>(db.cmask);
  a.qimg = static_cast<const char *>(qimg);
  a.n_items = n_items;
  a.nq = nq;
  a.lb = lb;
  a.ld_lb = ld_lb;
  const int64_t ntiles = (n_items + 31) / 32;
  const int64_t total = ((ntiles + 3) / 4) * (int64_t)nq;
  // one workgroup per CU with an equal share; small problems use fewer workgroups so that a
  // 300-register B load is amortised over >= 16 queries
  int64_t per = (total + n_cu - 1) / n_cu;
  if (per < 16) per = 16;
  const int64_t nblk = (total + per - 1) / per;
  a.per_block = per;
  a.tb_qmin = nullptr;
  a.tb_cum = nullptr;
  unsigned grid = (unsigned)nblk;
  if (plan && plan_ws) {
    RSX_TRY(launch_filter_plan(db, *plan, nq, n_items, 1, plan_ws, &a.tb_qmin, &a.tb_cum, s));
    grid = (unsigned)n_cu;  // the total is only known on the device: every workgroup takes total / n_cu
  }
  hipLaunchKernelGGL(sc_filter_kernel, dim3(grid), dim3(256), lds, s, a);
  RSX_HIP(hipGetLastError());
  return RSX_OK;
}

int launch_select(const DbView &db, const lb_t *lb, int64_t ld_lb, int64_t n_items, int32_t nq, int64_t n_eligible,
                  const int64_t *q_elig, int32_t first_target, RescoreEntry *slist, int32_t *sl_cnt, float *thr,
                  hipStream_t s) {
  if (nq <= 0) return RSX_OK;
  if (first_target < 1 || first_target > 128) return fail(RSX_ERR_INTERNAL, "bad first_target");
  const Elig el{db.idx_base, db.idx_stride, n_eligible < 0 ? INT64_MAX : n_eligible, q_elig};
  hipLaunchKernelGGL(sc_select_kernel, dim3(nq), dim3(256), 0, s, lb, ld_lb, n_items, el, first_target, slist, sl_cnt, thr);
  RSX_HIP(hipGetLastError());
  return RSX_OK;
}

}  // namespace sc
}  // namespace rsx
