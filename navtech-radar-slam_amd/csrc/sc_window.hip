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

#ifndef WIN_OCC
#define WIN_OCC 4  // waves per SIMD the register budget is set for
#endif
#ifndef WIN_RING
#define WIN_RING 15  // B fragments of the image GEMM in flight per wave
#endif
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
  win::query_keys_image(lane < NS ? vkey[(int64_t)q * NS + lane] : 0.0, st[wave], qk + (int64_t)q * WINDOW_QK_BYTES, lane);
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
  int32_t head;   // |head| = list positions pass 1 always processes (a multiple of 32, <= WINDOW_HEAD); negative: the
                  // positions behind them get "no record" (stage 1 of a DB shard: it scores the head only)
};


// TAIL (stage 2 of a DB shard whose stage 1 scored the head only): no pass 1; the positions behind the head whose bound can
// still reach `tau` -- the k-th best distance of the merged stage-1 lists, known only after the exchange -- get their records now
template <bool TAIL>
__device__ __forceinline__ void window_body(const WindowArgs &a, const rsx_sc_hit *__restrict__ global) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ float s_ub[WINDOW_HEAD];
  __shared__ int s_pos[WINDOW_P];
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
    u64 win;    // the union of the windows of the admissible alignments
    int kstar;  // the alignment when exactly one shift is admissible, else -1
    win::alignment_of(k0, k1, nq_key, uq_key, reinterpret_cast<const float2 *>(a.vk_n)[slot], hh, win, kstar);

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
    const float pv = win::preview_of(acc0, acc1, qm, em, win, kstar, hh);
    if (have && hh == 0) {
      WindowPreview o;
      o.pv = pv;
      o.ks = kstar;
      a.out[(int64_t)qi * WINDOW_P + pos] = o;
    }
    return (have && kstar >= 0 && pv < 3.0e38f) ? pv + WINDOW_MARGIN : INFINITY;  // NaN fails the compare
  };

  const int head = a.head < 0 ? -a.head : a.head;
  float tau_ub;
  if constexpr (TAIL) {
    if (sl_cnt <= head) return;  // uniform
    const double t = global[(int64_t)qi * a.k + (a.k - 1)].dist;  // (1e7 while fewer than k entries are known: everything passes)
    tau_ub = __double2float_ru(t);
  } else {
    // ---- pass 1: the head of the list ----
    const int cnt1 = sl_cnt < head ? sl_cnt : head;
    float ub = INFINITY;
    if (wave * 32 < cnt1) ub = do_group(wave * 32 + n, wave * 32 + n < cnt1, wave * 32);
    if (hh == 0) s_ub[wave * 32 + n] = ub;
    __syncthreads();
    if (sl_cnt <= head) return;  // uniform
    if (a.head < 0) {  // (uniform) the rest of the list: no record
      const int lim = sl_cnt < WINDOW_P ? sl_cnt : WINDOW_P;
      for (int pos = head + (int)threadIdx.x; pos < lim; pos += 256) {
        WindowPreview o;
        o.pv = __builtin_nanf("");
        o.ks = -2;
        a.out[(int64_t)qi * WINDOW_P + pos] = o;
      }
      return;
    }

    // ---- the k-th smallest upper bound of the head: an upper bound of the final k-th best distance.  Only entries whose
    // filter bound does not exceed it can matter to the re-scoring kernel (whose own bound is at least as tight) ----
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

  }
  // ---- pass 2: list positions WINDOW_HEAD .. WINDOW_P - 1 whose bound can still matter; the others get "no record" ----
  if (wave == 0) {
    int n2 = 0;
    const int lim = sl_cnt < WINDOW_P ? sl_cnt : WINDOW_P;
    for (int p0 = head; p0 < lim; p0 += 64) {
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

__global__ __launch_bounds__(256, WIN_OCC) void sc_window_kernel(WindowArgs a) { window_body<false>(a, nullptr); }
__global__ __launch_bounds__(256, WIN_OCC) void sc_window_tail_kernel(WindowArgs a, const rsx_sc_hit *__restrict__ global) { window_body<true>(a, global); }

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
                  int32_t k, double eps, WindowPreview *out, hipStream_t s, int32_t head_only) {
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
  a.head = WINDOW_HEAD;
  if (head_only > 0) {  // a DB shard's stage 1 scores only its first `head_only` list positions
    int32_t hd = (head_only + 31) / 32 * 32;
    if (hd > WINDOW_HEAD) hd = WINDOW_HEAD;
    a.head = -hd;
  }
  hipLaunchKernelGGL(sc_window_kernel, dim3((unsigned)q.nq), dim3(256), W_LDS, s, a);
  RSX_HIP(hipGetLastError());
  return RSX_OK;
}

// stage 2 of a DB shard: the records behind the head that stage 1 left out, for the list positions whose bound can still reach
// the k-th best distance of the merged stage-1 lists (d_global [nq][k]).  The query images of stage 1 are still in qimg.
int launch_window_tail(const DbView &db, int32_t nq, void *qimg, const RescoreEntry *slist, const int32_t *sl_cnt, int32_t k, double eps,
                       WindowPreview *out, int32_t head, const rsx_sc_hit *d_global, hipStream_t s) {
  if (nq <= 0) return RSX_OK;
  char *img = static_cast<char *>(qimg);
  WindowArgs a;
  a.hnR = static_cast<const char *>(db.hnR);
  a.vk16 = static_cast<const char *>(db.vk16);
  a.vk_n = db.vk_n;
  a.cmask = reinterpret_cast<const u64 *>(db.cmask);
  a.qimg = img;
  a.qkimg = img + (size_t)nq * FILTER_QIMG_BYTES;
  a.slist = slist;
  a.sl_cnt = sl_cnt;
  a.out = out;
  a.eps = eps;
  a.k = k < 1 ? 1 : (k > WINDOW_HEAD ? WINDOW_HEAD : k);
  int32_t hd = (head + 31) / 32 * 32;
  a.head = hd > WINDOW_HEAD ? WINDOW_HEAD : hd;
  hipLaunchKernelGGL(sc_window_tail_kernel, dim3((unsigned)nq), dim3(256), W_LDS, s, a, d_global);
  RSX_HIP(hipGetLastError());
  return RSX_OK;
}

const char *window_kernel_name() { return "sc_window_kernel"; }

}  // namespace sc
}  // namespace rsx
