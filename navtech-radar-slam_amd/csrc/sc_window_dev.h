// sc_window_dev.h -- the two matrix-core correlations of one (query, 32 entries) group as inline device functions: what to do
// with the accumulators of the sector-key alignment GEMM (which shifts can be the reference's fastAlignUsingVkey choice,
// SC.cpp:93-113) and of the image GEMM (the preview of distanceBtnScanContext over the window(s), SC.cpp:116-148), plus the
// LDS image of the query's sector key.  Shared by sc_window_kernel (sc_window.hip: the heads of the short lists behind the
// filter) and sc_q1_kernel (sc_q1.hip: every entry of the database for a single query).  Error bounds: sc_window.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "sc_kernels.h"
#include "sc_entry_dev.h"

namespace rsx {
namespace sc {
namespace win {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef unsigned long long u64;

constexpr float kWinAlignEps = 3e-5f;
constexpr u64 kNonFinite = 1ull << 63;
constexpr int W_STEPS = DS / 16;            // 75 K-steps of the image GEMM
constexpr int W_TILE1 = 40;                 // tile 1 (shifts 32..63) reads the A fragment 40 K-steps further on
constexpr int QK_COPY = 288;                // one displaced copy of the doubled key: 120 halves + pad; 18 slots = 2 mod 16
constexpr int QK_LO = 8 * QK_COPY;          // 2304: the lo copies
constexpr int QK_NORM = 2 * QK_LO;          // 4608: float sqrt(E_q) (NaN: no matrix-core alignment), then padding
static_assert(QK_NORM + 16 == WINDOW_QK_BYTES, "layout");
constexpr int W_LDS = FILTER_QIMG_BYTES + WINDOW_QK_BYTES;  // 14608

// query side of the alignment: row k of the circulant reads the doubled key q2[k .. k + 63] (q2[i] = key[i % 60]); 8 copies
// displaced by one element each keep that read 16-byte aligned (row k: copy k % 8 at element k - k % 8), and the copy stride
// of 18 sixteen-byte slots keeps the 16 rows a ds_read_b128 serves together on 16 different slots mod 16.
// Stage (one wave; v = the query's sector key, one element per lane): the doubled hi / lo keys into st (2 x 128 halves of LDS)
template <bool DPP = false>
__device__ __forceinline__ dev::KeySplit query_keys_stage(double v, _Float16 (*st)[128], int lane) {
  const dev::KeySplit k = dev::split_key<DPP>(lane < NS ? v : 0.0, lane);
  if (lane < NS) {
    st[0][lane] = k.hi;
    st[1][lane] = k.lo;
    st[0][lane + NS] = k.hi;
    st[1][lane + NS] = k.lo;
  }
  if (lane < 8) {
    st[0][2 * NS + lane] = (_Float16)0.0f;
    st[1][2 * NS + lane] = (_Float16)0.0f;
  }
  return k;
}
// the 2 x 8 displaced copies from st: element i = first, first + step, ... by this thread; out = WINDOW_QK_BYTES (global or LDS)
__device__ __forceinline__ void query_keys_copies(const _Float16 (*st)[128], char *out, int first, int step) {
  for (int i = first; i < 2 * 8 * (QK_COPY / 2); i += step) {
    const int part = i / (8 * (QK_COPY / 2));
    const int r = i % (8 * (QK_COPY / 2));
    const int c = r / (QK_COPY / 2), el = r % (QK_COPY / 2);
    const int src = c + el;  // copy c holds q2[c + el]
    const _Float16 v16 = src < 2 * NS ? st[part][src] : (_Float16)0.0f;
    *reinterpret_cast<_Float16 *>(out + part * QK_LO + c * QK_COPY + el * 2) = v16;
  }
}
__device__ __forceinline__ void query_keys_norms(const dev::KeySplit &k, char *out, int lane) {
  if (lane < 4) *reinterpret_cast<float *>(out + QK_NORM + lane * 4) = lane == 0 ? k.nrm : (lane == 1 ? k.unrm : 0.0f);
}
// all of it by one wave
__device__ __forceinline__ void query_keys_image(double v, _Float16 (*st)[128], char *out, int lane) {
  const dev::KeySplit k = query_keys_stage(v, st, lane);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  query_keys_copies(st, out, lane, 64);
  query_keys_norms(k, out, lane);
}

// admissible alignments of the lane's entry (lane n and n + 32 hold the two halves of entry n's 64 shift rows): every shift
// whose KC is within the error bound of the maximum (bit m of adm = shift m); all 60 when the keys cannot be compared here
// (non-finite, too large, too lopsided: see split_key / `balanced`).  win = the union of their windows; kstar = the alignment
// when exactly one shift is admissible, else -1
// LEAN (sc_q1.hip): the lane halves are joined with v_permlane32_swap instead of four ds_bpermute round trips
template <bool LEAN = false>
__device__ __forceinline__ void alignment_of(const floatx16 &k0, const floatx16 &k1, float nq_key, float uq_key, float2 en,
                                             int hh, u64 &win, int &kstar) {
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
  float gmx;
  if constexpr (LEAN) {
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
    gmx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
  } else {
    gmx = fmaxf(mx, __shfl_xor(mx, 32));
  }
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
  bool obad;
  if constexpr (LEAN) {
    const auto s0 = __builtin_amdgcn_permlane32_swap(m0, m0, false, false);
    const auto s1 = __builtin_amdgcn_permlane32_swap(m1, m1, false, false);
    const auto sb = __builtin_amdgcn_permlane32_swap((unsigned)bad, (unsigned)bad, false, false);
    m0 = s0[0] | s0[1];
    m1 = s1[0] | s1[1];
    obad = (sb[0] | sb[1]) != 0;
  } else {
    m0 |= (unsigned)__shfl_xor((int)m0, 32);
    m1 |= (unsigned)__shfl_xor((int)m1, 32);
    obad = __shfl_xor((int)bad, 32) != 0;
  }
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

// epilogue of the image GEMM: max of S_k / n_eff(k) over the window(s) (n_eff from the two column masks, as the filter);
// returns the preview pv (NaN: non-finite data; +inf: no effective column in the window) and, for a unique alignment, ORs
// into kstar (bits 8..14) which of the 7 window shifts can be the minimum at all.  acc0 / acc1 are overwritten.
// LEAN (sc_q1.hip, whose epilogue sits on the critical path of every tile): the same values with fewer instructions -- a query
// without an empty column meets every entry with n_eff(k) = the entry's column count at every shift (one reciprocal per
// entry instead of two funnel shifts and two popcounts per value), and the shift mask comes from ONE 60-bit mask of the
// values above the line, rotated by the window start, instead of a window position per value
template <bool LEAN = false>
__device__ __forceinline__ float preview_of(floatx16 &acc0, floatx16 &acc1, u64 qm, u64 em, u64 win, int &kstar, int hh) {
  float pv;
  if constexpr (LEAN) {
    constexpr u64 M60 = (1ull << NS) - 1ull;
    const u64 winh = win >> (4 * hh);  // bit (32 tl + b) = shift 32 tl + b + 4 hh
    const unsigned wlo = (unsigned)winh, whi = (unsigned)(winh >> 32);
    float best = -INFINITY;
    if ((qm & M60) == M60) {  // (the query's mask: wave-uniform)
      const float rn = __builtin_amdgcn_rcpf((float)__popcll(em & M60));  // no column at all: S == 0 exactly, 0 * inf = NaN, dropped by fmaxf
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int b = (r & 3) + 8 * (r >> 2);
        const float v0 = (wlo & (1u << b)) ? acc0[r] * rn : -INFINITY, v1 = (whi & (1u << b)) ? acc1[r] * rn : -INFINITY;
        acc0[r] = v0;
        acc1[r] = v1;
        best = fmaxf(best, fmaxf(v0, v1));
      }
    } else {
      const u64 m1c = qm & ~kNonFinite;
      const u64 lo = m1c | (m1c << 60), hi = m1c >> 4;  // the 60-bit mask twice in a row (120 bits)
      const u64 lo4 = (lo >> 4) | (hi << 60), hi4 = hi >> 4;
      const u64 l = hh ? lo4 : lo, h = hh ? hi4 : hi;
      const unsigned w[4] = {(unsigned)l, (unsigned)(l >> 32), (unsigned)h, (unsigned)(h >> 32)};
      const unsigned m2lo = (unsigned)em, m2hi = (unsigned)(em >> 32) & 0x0fffffffu;
      auto piece = [&](int tl, int r, float S) -> float {
        const int b = (r & 3) + 8 * (r >> 2);
        const unsigned rlo = __builtin_amdgcn_alignbit(w[tl + 1], w[tl], b);
        const unsigned rhi = __builtin_amdgcn_alignbit(w[tl + 2], w[tl + 1], b);
        const int ne = __builtin_popcount(rlo & m2lo) + __builtin_popcount(rhi & m2hi);
        float v = S * __builtin_amdgcn_rcpf((float)ne);
        v = (((tl ? whi : wlo) >> b) & 1u) ? v : -INFINITY;
        best = fmaxf(best, v);
        return v;
      };
#pragma unroll
      for (int r = 0; r < 16; r++) acc0[r] = piece(0, r, acc0[r]);
#pragma unroll
      for (int r = 0; r < 16; r++) acc1[r] = piece(1, r, acc1[r]);
    }
    {
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(best), __float_as_uint(best), false, false);
      best = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    }
    pv = fmaf(best, -1.0f / FILTER_ACC_SCALE, 1.0f);  // -inf (no effective column in the window) -> +inf
    if (kstar >= 0) {  // the shift mask (see below): bit t = shift k* - 3 + t is within 2 margins of the best one
      const float line = best - 2.0f * WINDOW_MARGIN * FILTER_ACC_SCALE;
      unsigned g0 = 0, g1 = 0;
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int b = (r & 3) + 8 * (r >> 2);
        g0 |= (acc0[r] >= line) ? (1u << b) : 0u;  // -inf outside the window, NaN without an effective column: no bit
        g1 |= (acc1[r] >= line) ? (1u << b) : 0u;
      }
      u64 ge = (((u64)g1 << 32) | g0) << (4 * hh);
      {
        const auto sl = __builtin_amdgcn_permlane32_swap((unsigned)ge, (unsigned)ge, false, false);
        const auto sh = __builtin_amdgcn_permlane32_swap((unsigned)(ge >> 32), (unsigned)(ge >> 32), false, false);
        ge = ((u64)(sh[0] | sh[1]) << 32) | (sl[0] | sl[1]);
      }
      int k0s = kstar - 3;
      k0s += k0s < 0 ? NS : 0;
      const unsigned mask7 = (unsigned)((ge >> k0s) | (ge << (NS - k0s))) & 0x7fu;
      kstar |= (int)(mask7 << 8);
    }
  } else {
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
  return pv;
}

}  // namespace win
}  // namespace sc
}  // namespace rsx
