// sc_window.hip -- window previews of the re-scoring short lists on the matrix cores (gfx950 / CDNA4).
//
// Where it sits.  The lower-bound filter (sc_spec.hip / sc_filter.hip) gives every (query, entry) pair the minimum of
// the column-cosine distance over ALL 60 shifts; sc_select_kernel turns a query's row of bounds into a short list in
// ascending-bound order; sc_rescore_kernel then needs, for every short-list entry it looks at,
//     k*  = the sector-key alignment of the pair (fastAlignUsingVkey, SC.cpp:93-113) and
//     pv ~= dist(query, entry) = min over the 7 shifts k* - 3 .. k* + 3 of d_k (SC.cpp:116-148)
// to decide which few entries deserve the exact fp64 evaluation.  Rounds 1-2 computed both on the VALU, one entry per
// wavefront (phase_a in sc_kernels.hip: ~650 issue slots per entry, 137 entries per query: 73 % of the re-scoring
// kernel).  Both are circular correlations of the query with the entry -- GEMMs whose A operand is a circulant of the
// query -- so for the head of every short list (two passes, see the kernel) they are computed here, 32 entries per wavefront:
//   * alignment: KC[k] = sum_j vkey_q[(j + k) % 60] * vkey_e[j], K = 64, keys scaled by a power of two and split into
//     fp16 hi + lo (hi*hi + hi*lo + lo*hi: 24 v_mfma_f32_32x32x16_f16 for 2 x 32 shifts x 32 entries).  argmin_k of
//     ||vkey_q - shift_k(vkey_e)|| = argmax_k KC[k] (the two squared norms do not depend on k).  The maximum is taken
//     as k* only when it is UNIQUE within the error bound of KC (below); otherwise k* = -1 and the preview is taken
//     over the union of the windows of every shift that could be the reference's choice -- still a valid LOWER bound of
//     the pair distance, so such an entry is usually pruned as well, and only if it survives does the re-scoring kernel
//     run the exact fp64 alignment with the reference's tie rule.
//   * preview: S[k][e] = sum_i q2[i + 20 k] * e[i], K = 1200, exactly the direct filter's GEMM (same fp16 images, same
//     circulant addressing of the query image in LDS, same epilogue arithmetic) -- 150 MFMAs per 32 entries -- but the
//     epilogue takes the minimum of d_k = 1 - S_k / n_eff(k) over the window of k* only.  |pv - dist| <= WINDOW_MARGIN
//     (= the direct filter's error budget, sc_filter.hip: 2u + u^2 from the fp16 operands + 1200 * 2^-23 from the fp32
//     accumulation + epilogue < 1.13e-3; shifts without an effective column are ignored on both sides).
// Cost: 8192 queries x ~146 entries = 1.2 M pairs at 174 MFMAs per 32 = 0.21 Tflop: ~0.1 ms of matrix-core time against
// the ~1.2 ms of VALU time it replaces.  The entries are gathered (2400 + 256 B each, whole rows of the entry-major
// image hnR): 3.1 GB per batch out of a 27 MB database image, i.e. from L2 / MALL -- which is what bounds the kernel
// (0.37 ms).
//
// Error bound of KC (scaled keys x, max |x| in [2^9, 2^10); E = sum x^2):
//   representation  x = hi + lo + r, |r| <= 2^-22 |x| (+ 2^-25 absolute where lo is subnormal)
//   dropped lo*lo   <= 2^-22 |x||y| per term
//   fp32 accumulation of 3 x 64 products in 12 chained MFMAs: <= 192 * 2^-23 relative to sum |terms| <= sqrt(E_q E_e)
//   total < (2.29e-5 + 4 * 2.4e-7) sqrt(E_q E_e);  kWinAlignEps = 3e-5 with sqrt(E) rounded up.
// Two quirks of the reference's search bound where this applies (split_key / `balanced` below): it starts from a best
// distance of 1e7 (keys with norms >= 4e6 are declined), and it works in fp64 on UNSCALED keys (pairs whose key norms
// differ by more than 1e6 are declined).
// A shift other than the true argmax can only reach KC_max - 2 eps sqrt(E_q E_e) when the true values are that close,
// so a unique candidate above that line IS the reference's argmin (whose fp64 arithmetic is off by < 1e-13 relative).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "rsx_common.h"
#include "sc_kernels.h"
#include "sc_entry_dev.h"

namespace rsx {
namespace sc {

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef unsigned long long u64;

#ifndef WIN_OCC
#define WIN_OCC 4  // waves per SIMD the register budget is set for
#endif
#ifndef WIN_RING
#define WIN_RING 15  // B fragments of the image GEMM in flight per wave
#endif
constexpr float kWinAlignEps = 3e-5f;
using dev::KeySplit;
using dev::split_key;
constexpr u64 kNonFinite = 1ull << 63;
constexpr int W_STEPS = DS / 16;            // 75 K-steps of the image GEMM
constexpr int W_TILE1 = 40;                 // tile 1 (shifts 32..63) reads the A fragment 40 K-steps further on
constexpr int QK_COPY = 288;                // one displaced copy of the doubled key: 120 halves + pad; 18 slots = 2 mod 16
constexpr int QK_LO = 8 * QK_COPY;          // 2304: the lo copies
constexpr int QK_NORM = 2 * QK_LO;          // 4608: float sqrt(E_q) (NaN: no matrix-core alignment), then padding
static_assert(QK_NORM + 16 == WINDOW_QK_BYTES, "layout");
constexpr int W_LDS = FILTER_QIMG_BYTES + WINDOW_QK_BYTES;  // 14608

// ------------------------------------------------------------------------------------------
// database side: [slot][hi 0..63 | lo 0..63] fp16 (elements 60..63 zero: the K padding) + the key's scaled norm
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sc_win_db_keys_kernel(const double *__restrict__ vkey, int64_t first, int64_t count,
                                                             _Float16 *__restrict__ vk16, float *__restrict__ vk_n) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t it = (int64_t)blockIdx.x * 4 + wave;
  if (it >= count) return;
  dev::win_db_keys_entry(vkey, first + it, vk16, vk_n, lane);
}

// ------------------------------------------------------------------------------------------
// query side: row k of the circulant reads the doubled key q2[k .. k + 63] (q2[i] = key[i % 60]); 8 copies displaced
// by one element each keep that read 16-byte aligned (row k: copy k % 8 at element k - k % 8), and the copy stride
// of 18 sixteen-byte slots keeps the 16 rows a ds_read_b128 serves together on 16 different slots mod 16
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sc_win_query_keys_kernel(const double *__restrict__ vkey, int32_t nq,
                                                                char *__restrict__ qk) {
  __shared__ _Float16 st[4][2][128];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = blockIdx.x * 4 + wave;
  if (q >= nq) return;
  const KeySplit k = split_key(lane < NS ? vkey[(int64_t)q * NS + lane] : 0.0, lane);
  if (lane < NS) {
    st[wave][0][lane] = k.hi;
    st[wave][1][lane] = k.lo;
    st[wave][0][lane + NS] = k.hi;
    st[wave][1][lane + NS] = k.lo;
  }
  if (lane < 8) {
    st[wave][0][2 * NS + lane] = (_Float16)0.0f;
    st[wave][1][2 * NS + lane] = (_Float16)0.0f;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  char *out = qk + (int64_t)q * WINDOW_QK_BYTES;
  for (int i = lane; i < 2 * 8 * (QK_COPY / 2); i += 64) {
    const int part = i / (8 * (QK_COPY / 2));
    const int r = i % (8 * (QK_COPY / 2));
    const int c = r / (QK_COPY / 2), el = r % (QK_COPY / 2);
    const int src = c + el;  // copy c holds q2[c + el]
    const _Float16 v = src < 2 * NS ? st[wave][part][src] : (_Float16)0.0f;
    *reinterpret_cast<_Float16 *>(out + part * QK_LO + c * QK_COPY + el * 2) = v;
  }
  if (lane < 4) *reinterpret_cast<float *>(out + QK_NORM + lane * 4) = lane == 0 ? k.nrm : (lane == 1 ? k.unrm : 0.0f);
}

// ------------------------------------------------------------------------------------------
// the window kernel: one workgroup per query, wave w takes short-list positions 32 w .. 32 w + 31 (then + 128, ...)
// ------------------------------------------------------------------------------------------
struct WindowArgs {
  const char *hnR;
  const char *vk16;
  const float *vk_n;
  const u64 *cmask;
  const char *qimg;   // [nq][FILTER_QIMG_BYTES]
  const char *qkimg;  // [nq][WINDOW_QK_BYTES]
  const RescoreEntry *slist;
  const int32_t *sl_cnt;
  WindowPreview *out;
  double eps;  // the filter's error budget, as the re-scoring kernel applies it to a bound
  int32_t k;
};


__global__ __launch_bounds__(256, WIN_OCC) void sc_window_kernel(WindowArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ float s_ub[WINDOW_HEAD];
  __shared__ int s_pos[WINDOW_P - WINDOW_HEAD];
  __shared__ int s_n2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int qi = blockIdx.x;
  const int sl_cnt = a.sl_cnt[qi];
  if (sl_cnt <= 0) return;  // uniform
  {
    const uint4 *g0 = reinterpret_cast<const uint4 *>(a.qimg + (int64_t)qi * FILTER_QIMG_BYTES);
    const uint4 *g1 = reinterpret_cast<const uint4 *>(a.qkimg + (int64_t)qi * WINDOW_QK_BYTES);
    uint4 *l = reinterpret_cast<uint4 *>(smem);
    for (int i = threadIdx.x; i < W_LDS / 16; i += 256) l[i] = i < FILTER_QIMG_BYTES / 16 ? g0[i] : g1[i - FILTER_QIMG_BYTES / 16];
  }
  __syncthreads();
  const int n = lane & 31, hh = lane >> 5;
  const RescoreEntry *sl = a.slist + (int64_t)qi * RESCORE_SHORTLIST_CAP;
  const u64 qm = *reinterpret_cast<const u64 *>(smem + FILTER_QIMG_MASK_OFF);
  const float nq_key = *reinterpret_cast<const float *>(smem + FILTER_QIMG_BYTES + QK_NORM);
  const float uq_key = *reinterpret_cast<const float *>(smem + FILTER_QIMG_BYTES + QK_NORM + 4);
  // A-fragment addresses of this lane's row (shift n of tile 0; tile 1 = the same address + 40 K-steps, sc_filter.hip)
  const char *ap = smem + ((n & 1) ? (FILTER_QIMG_ODD + 40 * n - 8) : (40 * n)) + 16 * hh;
  const char *kp = smem + FILTER_QIMG_BYTES + (n & 7) * QK_COPY + ((n & ~7) + 8 * hh) * 2;  // tile 1: + 64 B

  // one group of 32 short-list entries (lane n and n + 32: entry at list position `pos`; have = the lane has one, else it
  // shadows position pos_any): writes the record, returns the upper bound of the pair distance the record implies
  auto do_group = [&](int pos, bool have, int pos_any) -> float {
    const int64_t slot = sl[have ? pos : pos_any].slot;

    // ---- alignment: 2 tiles x (hi*hi + hi*lo + lo*hi) x 4 K-steps ----
    floatx16 k0 = {0}, k1 = {0};
    {
      const char *bk = a.vk16 + slot * 256 + 16 * hh;
      half8 bh[4], bl[4];
#pragma unroll
      for (int s = 0; s < 4; s++) {
        bh[s] = *reinterpret_cast<const half8 *>(bk + 32 * s);
        bl[s] = *reinterpret_cast<const half8 *>(bk + 128 + 32 * s);
      }
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
    }
    // admissible alignments: every shift whose KC is within the error bound of the maximum (bit m of adm = shift m);
    // all 60 when the keys cannot be compared here (non-finite, too large, too lopsided: see split_key / `balanced`)
    u64 win;    // the union of their windows
    int kstar;  // the alignment when exactly one shift is admissible, else -1
    {
      float mx = -INFINITY;
      bool bad = false;
#pragma unroll
      for (int r = 0; r < 16; r++) {
        bad |= !(k0[r] == k0[r]);
        mx = fmaxf(mx, k0[r]);
      }
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const float v = (r >= 12 && hh) ? 0.0f : k1[r];  // rows 60..63 are padding (M60 drops their bits below)
        bad |= !(v == v);
        mx = fmaxf(mx, (r >= 12 && hh) ? -INFINITY : v);
      }
      const float gmx = fmaxf(mx, __shfl_xor(mx, 32));
      const float2 en = reinterpret_cast<const float2 *>(a.vk_n)[slot];
      const float thr = 2.0f * kWinAlignEps * nq_key * en.x;  // NaN when either key is unusable
      // KC is scale-free, the reference's fp64 arithmetic is not: it compares ||vkey_q - shift(vkey_e)||, and when one key
      // is much smaller than the other every shift gives the same double (its search then keeps the first one).  A
      // separation of 2 eps sqrt(E_q E_e) in KC is a RELATIVE separation >= 2.4e-4 * ratio of the squared distances
      // (<= (|q| + |e|)^2 <= 4 max^2): with ratio = min norm / max norm >= 1e-6 that is 2.4e-10, six orders above the
      // 60 * 2^-52 the fp64 sums can be off by; more lopsided pairs count as "cannot be compared"
      const float umin = fminf(uq_key, en.y), umax = fmaxf(uq_key, en.y);
      const bool balanced = umin >= 1e-6f * umax && umax < INFINITY && umin > 0.0f;
      const float line = gmx - thr;
      unsigned m0 = 0, m1 = 0;
#pragma unroll
      for (int r = 0; r < 16; r++) m0 |= (k0[r] >= line) ? (1u << ((r & 3) + 8 * (r >> 2))) : 0u;
#pragma unroll
      for (int r = 0; r < 16; r++) m1 |= (k1[r] >= line) ? (1u << ((r & 3) + 8 * (r >> 2))) : 0u;
      m0 <<= 4 * hh;
      m1 <<= 4 * hh;
      m0 |= (unsigned)__shfl_xor((int)m0, 32);
      m1 |= (unsigned)__shfl_xor((int)m1, 32);
      const bool obad = __shfl_xor((int)bad, 32) != 0;
      constexpr u64 M60 = (1ull << NS) - 1ull;
      u64 adm = (((u64)m1 << 32) | m0) & M60;
      const bool comparable = !bad && !obad && (thr == thr) && thr < 3.0e38f && balanced && adm != 0;
      if (!comparable) adm = M60;
      kstar = (__popcll(adm) == 1) ? (__ffsll((long long)adm) - 1) : -1;
      win = adm;
#pragma unroll
      for (int o = 1; o <= 3; o++) {
        win |= ((adm << o) | (adm >> (NS - o))) & M60;
        win |= ((adm >> o) | (adm << (NS - o))) & M60;
      }
    }

    // ---- the 60 correlation values of the two images (the direct filter's GEMM) ----
    __builtin_amdgcn_sched_barrier(0);  // keep the loads below from being hoisted over the alignment (register pressure)
    floatx16 acc0 = {0}, acc1 = {0};
    {
      const char *brow = a.hnR + slot * (2 * DS) + 16 * hh;
      // B fragments (16 bytes per lane from 32 different rows) through a ring of WIN_RING registers sets: a slot is
      // refilled right after its MFMAs, so WIN_RING - 1 gathers per wave are in flight all the time (two buffers of five,
      // the first version, had 5..10: the kernel is bound by the latency x concurrency of this gather, not by the MFMAs)
      constexpr int R = WIN_RING;
      static_assert(W_STEPS % R == 0, "whole ring turns");
      half8 ring[R];
#pragma unroll
      for (int u = 0; u < R; u++) ring[u] = *reinterpret_cast<const half8 *>(brow + 32 * u);
#pragma unroll 1
      for (int s0 = 0; s0 < W_STEPS - R; s0 += R) {
#pragma unroll
        for (int u = 0; u < R; u++) {
          const half8 a0 = *reinterpret_cast<const half8 *>(ap + 32 * (s0 + u));
          const half8 a1 = *reinterpret_cast<const half8 *>(ap + 32 * (s0 + u + W_TILE1));
          acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, ring[u], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, ring[u], acc1, 0, 0, 0);
          ring[u] = *reinterpret_cast<const half8 *>(brow + 32 * (s0 + R + u));
        }
      }
#pragma unroll
      for (int u = 0; u < R; u++) {  // the last turn: nothing left to request
        const half8 a0 = *reinterpret_cast<const half8 *>(ap + 32 * (W_STEPS - R + u));
        const half8 a1 = *reinterpret_cast<const half8 *>(ap + 32 * (W_STEPS - R + u + W_TILE1));
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, ring[u], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, ring[u], acc1, 0, 0, 0);
      }
    }

    // ---- epilogue: max of S_k / n_eff(k) over the window of k* (n_eff from the two column masks, as the filter) ----
    __builtin_amdgcn_sched_barrier(0);
    const u64 em = a.cmask[slot];
    float pv;
    {
      const u64 m1c = qm & ~kNonFinite;
      const u64 lo = m1c | (m1c << 60), hi = m1c >> 4;  // the 60-bit mask twice in a row (120 bits)
      const u64 lo4 = (lo >> 4) | (hi << 60), hi4 = hi >> 4;
      const u64 l = hh ? lo4 : lo, h = hh ? hi4 : hi;
      const unsigned w[4] = {(unsigned)l, (unsigned)(l >> 32), (unsigned)h, (unsigned)(h >> 32)};
      const unsigned m2lo = (unsigned)em, m2hi = (unsigned)(em >> 32) & 0x0fffffffu;
      const u64 winh = win >> (4 * hh);  // bit (32 tl + b) = shift 32 tl + b + 4 hh
      const unsigned wlo = (unsigned)winh, whi = (unsigned)(winh >> 32);
      float best = -INFINITY;
      auto piece = [&](int tl, int r, float S) -> float {
        const int b = (r & 3) + 8 * (r >> 2);
        const unsigned rlo = __builtin_amdgcn_alignbit(w[tl + 1], w[tl], b);
        const unsigned rhi = __builtin_amdgcn_alignbit(w[tl + 2], w[tl + 1], b);
        const int ne = __builtin_popcount(rlo & m2lo) + __builtin_popcount(rhi & m2hi);
        float v = S * __builtin_amdgcn_rcpf((float)ne);  // n_eff == 0: S == 0 exactly, 0 * inf = NaN, dropped by fmaxf
        const bool inwin = ((tl ? whi : wlo) >> b) & 1u;  // (the padding rows 60..63 are never in the window)
        v = inwin ? v : -INFINITY;
        best = fmaxf(best, v);
        return v;
      };
#pragma unroll
      for (int r = 0; r < 16; r++) acc0[r] = piece(0, r, acc0[r]);  // S_k -> S_k / n_eff(k) inside the window, -inf outside
#pragma unroll
      for (int r = 0; r < 16; r++) acc1[r] = piece(1, r, acc1[r]);
      best = fmaxf(best, __shfl_xor(best, 32));
      pv = fmaf(best, -1.0f / FILTER_ACC_SCALE, 1.0f);  // -inf (no effective column in the window) -> +inf
      // which of the 7 window shifts can be the minimum at all: d_t >= pv_t - margin and d_min <= pv_min + margin, so a
      // shift with pv_t > pv_min + 2 margin is STRICTLY worse than the best one and the exact evaluation may skip it
      // (bit t of the mask = shift k* - 3 + t; only meaningful with a unique alignment)
      if (kstar >= 0) {
        const float line = best - 2.0f * WINDOW_MARGIN * FILTER_ACC_SCALE;
        int k0s = kstar - 3;
        k0s += k0s < 0 ? NS : 0;
        unsigned mask7 = 0;
        auto near = [&](int tl, int r, float v) {
          int t = 32 * tl + (r & 3) + 8 * (r >> 2) + 4 * hh - k0s;
          t += t < 0 ? NS : 0;
          mask7 |= (v >= line && t < 7) ? (1u << t) : 0u;  // v = -inf outside the window, NaN without an effective column
        };
#pragma unroll
        for (int r = 0; r < 16; r++) near(0, r, acc0[r]);
#pragma unroll
        for (int r = 0; r < 16; r++) near(1, r, acc1[r]);
        mask7 |= (unsigned)__shfl_xor((int)mask7, 32);
        kstar |= (int)(mask7 << 8);
      }
    }
    if ((qm | em) & kNonFinite) pv = __builtin_nanf("");
    if (have && hh == 0) {
      WindowPreview o;
      o.pv = pv;
      o.ks = kstar;
      a.out[(int64_t)qi * WINDOW_P + pos] = o;
    }
    return (have && kstar >= 0 && pv < 3.0e38f) ? pv + WINDOW_MARGIN : INFINITY;  // NaN fails the compare
  };

  // ---- pass 1: the head of the list ----
  const int cnt1 = sl_cnt < WINDOW_HEAD ? sl_cnt : WINDOW_HEAD;
  float ub = INFINITY;
  if (wave * 32 < cnt1) ub = do_group(wave * 32 + n, wave * 32 + n < cnt1, wave * 32);
  if (hh == 0) s_ub[wave * 32 + n] = ub;
  __syncthreads();
  if (sl_cnt <= WINDOW_HEAD) return;  // uniform

  // ---- the k-th smallest upper bound of the head: an upper bound of the final k-th best distance.  Only entries whose
  // filter bound does not exceed it can matter to the re-scoring kernel (whose own bound is at least as tight) ----
  float tau_ub;
  {
    const float v0 = s_ub[lane], v1 = s_ub[lane + 64];
    int r0 = 0, r1 = 0;
    const float4 *u4 = reinterpret_cast<const float4 *>(s_ub);
#pragma unroll 4
    for (int j = 0; j < WINDOW_HEAD / 4; j++) {
      const float4 u = u4[j];
      const float x[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const int idx = 4 * j + e;
        r0 += (x[e] < v0 || (x[e] == v0 && idx < lane)) ? 1 : 0;
        r1 += (x[e] < v1 || (x[e] == v1 && idx < lane + 64)) ? 1 : 0;
      }
    }
    const int want = a.k - 1;  // 0 <= want < WINDOW_HEAD (k <= RSX_SC_MAX_TOPK)
    const unsigned long long b0 = __ballot(r0 == want), b1 = __ballot(r1 == want);
    const float c0 = __shfl(v0, b0 ? __ffsll((long long)b0) - 1 : 0), c1 = __shfl(v1, b1 ? __ffsll((long long)b1) - 1 : 0);
    tau_ub = b0 ? c0 : c1;  // ranks are a permutation of 0..127: exactly one of the two ballots has a bit
  }

  // ---- pass 2: list positions WINDOW_HEAD .. WINDOW_P - 1 whose bound can still matter; the others get "no record" ----
  if (wave == 0) {
    int n2 = 0;
    const int lim = sl_cnt < WINDOW_P ? sl_cnt : WINDOW_P;
    for (int p0 = WINDOW_HEAD; p0 < lim; p0 += 64) {
      const int pos = p0 + lane;
      bool pass = false;
      if (pos < lim) {
        const float lb = sl[pos].lb;
        pass = !((double)lb - a.eps > (double)tau_ub);  // NaN / -inf bounds: always
        if (!pass) {
          WindowPreview o;
          o.pv = __builtin_nanf("");
          o.ks = -2;
          a.out[(int64_t)qi * WINDOW_P + pos] = o;
        }
      }
      const unsigned long long bal = __ballot(pass);
      if (pass) s_pos[n2 + __popcll(bal & ((1ull << lane) - 1ull))] = pos;
      n2 += __popcll(bal);
    }
    if (lane == 0) s_n2 = n2;
  }
  __syncthreads();
  const int n2 = s_n2;
  for (int g = wave; g * 32 < n2; g += 4) {
    const bool have = g * 32 + n < n2;
    (void)do_group(s_pos[have ? g * 32 + n : g * 32], have, s_pos[g * 32]);
  }
}

}  // namespace

size_t window_qimg_bytes(int32_t nq) { return (size_t)nq * (FILTER_QIMG_BYTES + WINDOW_QK_BYTES) + 1024; }

int launch_window_db_keys(const double *vkey, int64_t first, int64_t count, void *vk16, float *vk_n, hipStream_t s) {
  if (count <= 0) return RSX_OK;
  hipLaunchKernelGGL(sc_win_db_keys_kernel, dim3((unsigned)((count + 3) / 4)), dim3(256), 0, s, vkey, first, count,
                     static_cast<_Float16 *>(vk16), vk_n);
  RSX_HIP(hipGetLastError());
  return RSX_OK;
}

int launch_window(const DbView &db, const QueryView &q, void *qimg, const RescoreEntry *slist, const int32_t *sl_cnt,
                  int32_t k, double eps, WindowPreview *out, hipStream_t s) {
  if (q.nq <= 0) return RSX_OK;
  char *img = static_cast<char *>(qimg);
  char *kimg = img + (size_t)q.nq * FILTER_QIMG_BYTES;
  RSX_TRY(launch_query_images(q.desc, q.norm, q.nq, img, s));
  hipLaunchKernelGGL(sc_win_query_keys_kernel, dim3((unsigned)((q.nq + 3) / 4)), dim3(256), 0, s, q.vkey, q.nq, kimg);
  RSX_HIP(hipGetLastError());
  WindowArgs a;
  a.hnR = static_cast<const char *>(db.hnR);
  a.vk16 = static_cast<const char *>(db.vk16);
  a.vk_n = db.vk_n;
  a.cmask = reinterpret_cast<const u64 *>(db.cmask);
  a.qimg = img;
  a.qkimg = kimg;
  a.slist = slist;
  a.sl_cnt = sl_cnt;
  a.out = out;
  a.eps = eps;
  a.k = k < 1 ? 1 : (k > WINDOW_HEAD ? WINDOW_HEAD : k);
  hipLaunchKernelGGL(sc_window_kernel, dim3((unsigned)q.nq), dim3(256), W_LDS, s, a);
  RSX_HIP(hipGetLastError());
  return RSX_OK;
}

const char *window_kernel_name() { return "sc_window_kernel"; }

}  // namespace sc
}  // namespace rsx
