// orora.hip -- ORORA scan registration on gfx950: GNC-TLS rotation + A-COTE translation, one
// 256-thread workgroup per scan pair, everything on-chip (points and the interval endpoints live in
// LDS, TIMs and GNC weights in registers); pairs of more than 2048 matches (up to 16384) go through the same
// code with 1024 threads and the arrays in an HBM workspace.
//
// The reference's ORORA sources are an empty submodule (/root/reference/.gitmodules:1-3,
// README.md:19,26-27,44-48), so this implements the published algorithm as restated in
// oracle/orora_ref.{h,c} (PARITY UNPINNED).  It replaces the
// solver calls of the upstream file-based odometry.cpp entry (README.md:27).
//
// Roofline: latency/VALU bound per pair (K x 16 B of input per pair): ~50 GNC iterations of
// O(K/256) fp64 work + 2-3 workgroup reductions each; the matrix pipe has nothing to do here
// (the "SVD" of a 2x2 weighted cross-covariance is atan2 of two sums).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <mutex>
#include <new>

#include "rsx_common.h"
#include "pmc.h"

namespace {

// Two instantiations of the same kernel:
//   on-chip   256 threads, pairs of <= 2048 matches: points, interval endpoints and scan partials in LDS (64 KB: two
//             workgroups per CU);
//   large     1024 threads, pairs of 2049 .. 16384 matches (cen2019 can emit 10 000 keypoints per scan): the same
//             arrays in a per-workgroup HBM workspace (L2-resident), a few persistent workgroups pulling the large
//             pairs from a list.  Round 1 returned identity + status 2 for such a pair -- a silently lost scan pair.
constexpr int MAXK_LDS = 2048;
constexpr int MAXK_BIG = 16384;

struct Params {
  double c2, s_r, s_t, gnc_factor, cost_threshold;
  int max_iterations;
  int complete_graph;  // TIMs on all pairs i < j instead of the ring
  int teaser_cost;     // scalar TLS cost: unweighted residuals + sum of the outliers' bounds
};

// -DRSX_ORORA_PROF=1: cycles per phase of the on-chip kernel (thread 0 of every workgroup, summed), printed by the launcher
#ifndef RSX_ORORA_PROF
#define RSX_ORORA_PROF 0
#endif
#if RSX_ORORA_PROF
__device__ unsigned long long g_orora_prof[8];
#define OPROF_DECL long long oprof_t = clock64()
#define OPROF(slot)                                                            \
  do {                                                                         \
    if (threadIdx.x == 0) {                                                    \
      const long long oprof_n = clock64();                                     \
      atomicAdd(&g_orora_prof[slot], (unsigned long long)(oprof_n - oprof_t)); \
      oprof_t = oprof_n;                                                       \
    }                                                                          \
  } while (0)
#else
#define OPROF_DECL
#define OPROF(slot)
#endif

// ---- workgroup reductions, deterministic order ----
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = fmax(v, __shfl_xor(v, off));
  return v;
}

template <int NW>
struct Red {
  double *buf;  // 3 * NW doubles of LDS
  __device__ double sum(double v) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) buf[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = 0.0;
#pragma unroll
    for (int w = 0; w < NW; w++) r += buf[w];
    return r;
  }
  __device__ void sum2(double &a, double &b) {
    a = wave_sum(a);
    b = wave_sum(b);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
      buf[threadIdx.x >> 6] = a;
      buf[NW + (threadIdx.x >> 6)] = b;
    }
    __syncthreads();
    a = 0.0;
    b = 0.0;
#pragma unroll
    for (int w = 0; w < NW; w++) {
      a += buf[w];
      b += buf[NW + w];
    }
  }
  __device__ void sum3(double &a, double &b, double &c) {
    a = wave_sum(a);
    b = wave_sum(b);
    c = wave_sum(c);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
      buf[threadIdx.x >> 6] = a;
      buf[NW + (threadIdx.x >> 6)] = b;
      buf[2 * NW + (threadIdx.x >> 6)] = c;
    }
    __syncthreads();
    a = 0.0;
    b = 0.0;
    c = 0.0;
#pragma unroll
    for (int w = 0; w < NW; w++) {
      a += buf[w];
      b += buf[NW + w];
      c += buf[2 * NW + w];
    }
  }
  __device__ double max(double v) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) buf[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = buf[0];
#pragma unroll
    for (int w = 1; w < NW; w++) r = fmax(r, buf[w]);
    return r;
  }
};

__device__ __forceinline__ void aniso_bound(double px, double py, double s_r, double s_t, double &bx, double &by) {
  const double rho = sqrt(px * px + py * py);
  double c = 1.0, s = 0.0;
  if (rho > 0.0) {
    c = fabs(px) / rho;
    s = fabs(py) / rho;
  }
  bx += c * s_r + s * rho * s_t;
  by += s * s_r + c * rho * s_t;
}

// (value, signed id) order of std::pair<double,int> (TEASER++ ScalarTLSEstimator)
__device__ __forceinline__ bool ep_less(double av, int ai, double bv, int bi) {
  return (av < bv) || (av == bv && ai < bi);
}

// GNC weight of a TIM with squared residual r2
__device__ __forceinline__ double gnc_weight(double r2, double mu, double c2) {
  const double th1 = (mu + 1.0) / mu * c2, th2 = mu / (mu + 1.0) * c2;
  if (r2 >= th1) return 0.0;
  if (r2 <= th2) return 1.0;
  return sqrt(c2 * mu * (mu + 1.0) / r2) - mu;
}

// Bitonic sort of n2 = 256 * E (value, id) endpoints by a 256-thread workgroup, E per thread in registers (endpoint
// g = tid * E + e: the chunk the sweep below gives to the thread anyway).  Compare-exchanges at a distance < E stay
// inside the thread, at a distance < 64 E they cross lanes (__shfl_xor), and only the two largest distances (the other
// waves) go through LDS: 3 barrier-separated passes instead of the 45 .. 78 of the plain LDS network, which was 60 % of
// the kernel (RSX_ORORA_PROF build: 376 k of 624 k cycles per pair in the two sorts).  Same total order, so the result
// is the same permutation bit for bit.
template <int E>
__device__ __forceinline__ void sort_cx(double &av, int &ai, double &bv, int &bi, bool up) {
  const bool swap = up ? ep_less(bv, bi, av, ai) : ep_less(av, ai, bv, bi);
  const double tv = swap ? bv : av, uv = swap ? av : bv;
  const int ti = swap ? bi : ai, ui = swap ? ai : bi;
  av = tv; bv = uv; ai = ti; bi = ui;
}
// compare-exchanges at the distances ST, ST / 2, .. 1 (compile-time register indices)
template <int E, int ST>
__device__ __forceinline__ void sort_in_thread(double (&v)[E], int (&id)[E], int tid, int size) {
  if constexpr (ST >= 1) {
    if (ST < size) {  // (uniform)
#pragma unroll
      for (int e = 0; e < E; e++) {
        if ((e & ST) == 0) {
          const bool up = ((tid * E + e) & size) == 0;
          sort_cx<E>(v[e], id[e], v[e | ST], id[e | ST], up);
        }
      }
    }
    sort_in_thread<E, ST / 2>(v, id, tid, size);
  }
}
template <int E>
__device__ __forceinline__ void bitonic_sort_regs(int K, const double *sx, const double *sb, double *ev, int *ei) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int n2 = 256 * E;
  double v[E];
  int id[E];
#pragma unroll
  for (int e = 0; e < E; e++) {
    const int g = tid * E + e, i = g >> 1;
    if (i < K) {
      const double xv = sx[i], b = sb[i];
      v[e] = (g & 1) ? xv + b : xv - b;
      id[e] = (g & 1) ? -i - 1 : i + 1;
    } else {
      v[e] = INFINITY;
      id[e] = 0x7fffffff;
    }
  }
#pragma unroll 1
  for (int size = 2; size <= n2; size <<= 1) {
    // distances >= 64 E: the partner is the same (lane, e) of another wave
#pragma unroll 1
    for (int stride = size >> 1; stride >= 64 * E; stride >>= 1) {
      __syncthreads();
      // exchange buffer laid out [e][thread]: consecutive lanes, consecutive words (laid out [thread][e] the 64 lanes of
      // an access were E doubles apart: 16- to 32-way bank conflicts)
#pragma unroll
      for (int e = 0; e < E; e++) {
        ev[e * 256 + tid] = v[e];
        ei[e * 256 + tid] = id[e];
      }
      __syncthreads();
      const bool lower = (wave & (stride / (64 * E))) == 0;
      const int ptid = tid ^ (stride / E);  // the partner endpoint g ^ stride is element e of that thread
#pragma unroll
      for (int e = 0; e < E; e++) {
        const int g = tid * E + e;
        const double ov = ev[e * 256 + ptid];
        const int oi = ei[e * 256 + ptid];
        const bool up = (g & size) == 0;
        const bool other_less = ep_less(ov, oi, v[e], id[e]);
        const bool take = (lower == up) ? other_less : !other_less;  // keep the smaller one at the lower end of an ascending pair
        v[e] = take ? ov : v[e];
        id[e] = take ? oi : id[e];
      }
    }
    // distances E .. 32 E: the partner is the same e of lane ^ (stride / E)
    {
      int stride = size >> 1;
      if (stride > 32 * E) stride = 32 * E;
#pragma unroll 1
      for (; stride >= E; stride >>= 1) {
        const int m = stride / E;
        const bool lower = (lane & m) == 0;
#pragma unroll
        for (int e = 0; e < E; e++) {
          const int g = tid * E + e;
          const double ov = __shfl_xor(v[e], m);
          const int oi = __shfl_xor(id[e], m);
          const bool up = (g & size) == 0;
          const bool other_less = ep_less(ov, oi, v[e], id[e]);
          const bool take = (lower == up) ? other_less : !other_less;
          v[e] = take ? ov : v[e];
          id[e] = take ? oi : id[e];
        }
      }
    }
    // distances < E: inside the thread
    sort_in_thread<E, E / 2>(v, id, tid, size);
  }
  __syncthreads();
#pragma unroll
  for (int e = 0; e < E; e++) ei[e * 256 + tid] = id[e];  // the sweep needs the order only; [e][thread] like the exchange buffer
  __syncthreads();
}

// Scalar TLS estimate over K intervals x[i] +- beta[i] held TPT per thread (point i = tid + t*NT).
// sx[K], sb[K] doubles, ev[2K'] doubles, ei[2K'] ints, part[6*NT] doubles (LDS or the HBM workspace).
template <int NT, int TPT>
__device__ __forceinline__ double scalar_tls_block(const double (&x)[TPT], const double (&beta)[TPT], int K, double *sx, double *sb,
                                   double *ev, int *ei, double *part, int teaser_cost) {
  constexpr int NW = NT / 64;
  const int tid = threadIdx.x;
  int n2 = 2 * NT;
  while (n2 < 2 * K) n2 <<= 1;
  __syncthreads();
  double l_sr = 0.0;
#pragma unroll
  for (int t = 0; t < TPT; t++) {
    const int i = tid + t * NT;
    if (i < K) {
      sx[i] = x[t];
      sb[i] = beta[t];
      l_sr += beta[t];
    }
  }
  __syncthreads();
#if RSX_ORORA_PROF
  long long oprof_t = clock64();
#endif
  if constexpr (NT == 256) {
    switch (n2) {  // (uniform) n2 = 256 E
      case 512: bitonic_sort_regs<2>(K, sx, sb, ev, ei); break;
      case 1024: bitonic_sort_regs<4>(K, sx, sb, ev, ei); break;
      case 2048: bitonic_sort_regs<8>(K, sx, sb, ev, ei); break;
      default: bitonic_sort_regs<16>(K, sx, sb, ev, ei); break;
    }
    // ev (the exchange buffer of the cross-wave passes) aliases sx / sb: put them back for the sweep
#pragma unroll
    for (int t = 0; t < TPT; t++) {
      const int i = tid + t * NT;
      if (i < K) {
        sx[i] = x[t];
        sb[i] = beta[t];
      }
    }
    __syncthreads();
  } else {
    // large pairs (1024 threads, arrays in HBM): the plain network, ascending by (value, id)
    for (int i = tid; i < K; i += NT) {
      ev[2 * i] = sx[i] - sb[i];
      ei[2 * i] = i + 1;
      ev[2 * i + 1] = sx[i] + sb[i];
      ei[2 * i + 1] = -i - 1;
    }
    for (int e = 2 * K + tid; e < n2; e += NT) {
      ev[e] = INFINITY;
      ei[e] = 0x7fffffff;
    }
    __syncthreads();
    for (int size = 2; size <= n2; size <<= 1) {
      for (int stride = size >> 1; stride > 0; stride >>= 1) {
        for (int p = tid; p < (n2 >> 1); p += NT) {
          const int lo = ((p & ~(stride - 1)) << 1) | (p & (stride - 1));
          const int hi = lo | stride;
          const bool up = (lo & size) == 0;
          const double a = ev[lo], b = ev[hi];
          const int ia = ei[lo], ib = ei[hi];
          const bool swap = up ? ep_less(b, ib, a, ia) : ep_less(a, ia, b, ib);
          if (swap) {
            ev[lo] = b; ev[hi] = a;
            ei[lo] = ib; ei[hi] = ia;
          }
        }
        __syncthreads();
      }
    }
  }
  OPROF(4);  // the sort alone (inside slot 2)
  // sweep as a scan: each thread owns a contiguous chunk of E endpoints; six running sums
  // (sum w, sum w x, sum w x^2 for the normalised cost; sum x, sum x^2, sum beta for TEASER++'s form) + cardinality
  const int E = n2 / NT;
  const int e0 = tid * E;
  double l[6] = {0, 0, 0, 0, 0, 0};
  int l_card = 0;
  for (int e = e0; e < e0 + E; e++) {
    const int id = (NT == 256) ? ei[(e - e0) * NT + tid] : ei[e];  // on-chip kernel: [e][thread] (bitonic_sort_regs)
    if (id == 0x7fffffff) continue;
    const int idx = (id > 0 ? id : -id) - 1;
    const double eps = id > 0 ? 1.0 : -1.0;
    const double b = sb[idx], xv = sx[idx];
    const double w = 1.0 / (b * b);
    const double wx = w * xv;
    l[0] += eps * w;
    l[1] += eps * wx;
    l[2] += eps * (wx * xv);
    l[3] += eps * xv;
    l[4] += eps * (xv * xv);
    l[5] += eps * b;
    l_card += id > 0 ? 1 : -1;
  }
#pragma unroll
  for (int q = 0; q < 6; q++) part[q * NT + tid] = l[q];
  part[6 * NT + tid] = (double)l_card;
  part[7 * NT + tid] = l_sr;
  __syncthreads();
  // exclusive scan of the NT chunk totals of the 8 quantities (7: only the total of the bounds is used).  One wavefront
  // per quantity, NT / 64 consecutive totals per lane, a shuffle scan across the lanes -- a single thread per quantity
  // walking its NT totals through LDS was 256 dependent round trips: 10 % of the kernel
  {
    constexpr int PER = NT / 64;
    const int lane = tid & 63, wv = tid >> 6;
    for (int qn = wv; qn < 8; qn += NW) {
      double *q = part + qn * NT;
      double pre[PER];
      double run = 0.0;
#pragma unroll
      for (int j = 0; j < PER; j++) {
        pre[j] = run;
        run += q[lane * PER + j];
      }
      double incl = run;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const double o = __shfl_up(incl, off);
        if (lane >= off) incl += o;
      }
      const double base = incl - run;
#pragma unroll
      for (int j = 0; j < PER; j++) q[lane * PER + j] = base + pre[j];
      if (qn == 7) {
        const double total = __shfl(incl, 63);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) q[0] = total;  // sum of all bounds
      }
    }
  }
  __syncthreads();
  double r[6];
#pragma unroll
  for (int q = 0; q < 6; q++) r[q] = part[q * NT + tid];
  int card = (int)part[6 * NT + tid];
  const double ranges_sum = part[7 * NT];
  double best_cost = INFINITY, best_x = 0.0;
  int best_pos = 0x7fffffff;
  for (int e = e0; e < e0 + E; e++) {
    const int id = (NT == 256) ? ei[(e - e0) * NT + tid] : ei[e];  // on-chip kernel: [e][thread] (bitonic_sort_regs)
    if (id == 0x7fffffff) continue;
    const int idx = (id > 0 ? id : -id) - 1;
    const double eps = id > 0 ? 1.0 : -1.0;
    const double b = sb[idx], xv = sx[idx];
    const double w = 1.0 / (b * b);
    const double wx = w * xv;
    r[0] += eps * w;
    r[1] += eps * wx;
    r[2] += eps * (wx * xv);
    r[3] += eps * xv;
    r[4] += eps * (xv * xv);
    r[5] += eps * b;
    card += id > 0 ? 1 : -1;
    if (card <= 0) continue;
    const double x_hat = r[1] / r[0];
    double cost;
    if (!teaser_cost) cost = (r[2] - 2.0 * r[1] * x_hat + r[0] * x_hat * x_hat) + (double)(K - card);
    else cost = ((double)card * x_hat * x_hat + r[4] - 2.0 * r[3] * x_hat) + (ranges_sum - r[5]);
    if (cost < best_cost) {  // first minimum inside the chunk
      best_cost = cost;
      best_x = x_hat;
      best_pos = e;
    }
  }
  // workgroup argmin under (cost, position): first minimum of the sequential sweep
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const double oc = __shfl_xor(best_cost, off), ox = __shfl_xor(best_x, off);
    const int op = __shfl_xor(best_pos, off);
    if (oc < best_cost || (oc == best_cost && op < best_pos)) {
      best_cost = oc; best_x = ox; best_pos = op;
    }
  }
  __syncthreads();
  if ((tid & 63) == 0) {
    part[tid >> 6] = best_cost;
    part[NW + (tid >> 6)] = best_x;
    part[2 * NW + (tid >> 6)] = (double)best_pos;
  }
  __syncthreads();
  double bc = part[0], bx = part[NW], bp = part[2 * NW];
  for (int w = 1; w < NW; w++)
    if (part[w] < bc || (part[w] == bc && part[2 * NW + w] < bp)) {
      bc = part[w]; bx = part[NW + w]; bp = part[2 * NW + w];
    }
  return bx;
}

// array sizes of one workgroup for pairs of up to MAXK matches
// The on-chip kernel (NT = 256) sorts in registers and needs `ev` only as the exchange buffer of the three cross-wave
// passes, while sx / sb are not read: there ev ALIASES the points / sx, sb (which are re-written after the sort), so a
// workgroup needs 64 KB of LDS and two fit a CU (two waves per SIMD instead of one).
template <int NT, int MAXK>
struct Layout {
  static constexpr int PTS = MAXK * 16;        // src + dst float2 (later sx, sb doubles)
  static constexpr int EV = NT == 256 ? 0 : 2 * MAXK * 8;
  static constexpr int EI = 2 * MAXK * 4;
  static constexpr int PART = 8 * NT * 8;
  static constexpr int TOTAL = PTS + EV + EI + PART;
};
constexpr int LDS_RED = 3 * 16 * 8;  // Red<NW>::sum3: 3 x NW doubles, NW <= 16

// One scan pair by one workgroup.  ws = the pair-sized arrays (LDS or HBM), red_lds = 32 doubles of LDS.
template <int NT, int MAXK>
__device__ void register_pair(const float2 *__restrict__ src, const float2 *__restrict__ dst, int64_t o, int K,
                              const Params &p, char *ws, double *red_lds, rsx_orora_result *out) {
  constexpr int TPT = MAXK / NT, NW = NT / 64;
  using L = Layout<NT, MAXK>;
  float2 *s_src = reinterpret_cast<float2 *>(ws);
  float2 *s_dst = s_src + MAXK;
  double *sx = reinterpret_cast<double *>(ws);  // aliases the points once they are dead
  double *sb = sx + MAXK;
  double *ev = reinterpret_cast<double *>(L::EV ? ws + L::PTS : ws);
  int *ei = reinterpret_cast<int *>(ws + L::PTS + L::EV);
  double *part = reinterpret_cast<double *>(ws + L::PTS + L::EV + L::EI);
  Red<NW> red{red_lds};
  const int tid = threadIdx.x;
  OPROF_DECL;
  __syncthreads();  // (persistent workgroups: the previous pair is done with ws)
  for (int i = tid; i < K; i += NT) {
    s_src[i] = src[o + i];
    s_dst[i] = dst[o + i];
  }
  __syncthreads();

  double mu = 1.0, prev_cost = INFINITY, cs = 1.0, sn = 0.0;
  int it = 0, rot_inliers = 0;
  if (!p.complete_graph) {
    // ---- TIMs on the closed chain, TPT per thread, in registers ----
    double ax[TPT], ay[TPT], bx[TPT], by[TPT], w[TPT], r2[TPT];
#pragma unroll
    for (int t = 0; t < TPT; t++) {
      const int j = tid + t * NT;
      if (j < K) {
        const int n = (j + 1 == K) ? 0 : j + 1;
        ax[t] = (double)s_src[n].x - (double)s_src[j].x;
        ay[t] = (double)s_src[n].y - (double)s_src[j].y;
        bx[t] = (double)s_dst[n].x - (double)s_dst[j].x;
        by[t] = (double)s_dst[n].y - (double)s_dst[j].y;
        w[t] = 1.0;
      } else {
        ax[t] = ay[t] = bx[t] = by[t] = 0.0;
        w[t] = 0.0;
      }
      r2[t] = 0.0;
    }
    // (slots t with t * NT >= K hold no TIM in any thread: zero weight, zero contribution -- skipped, wave-uniformly)
    // The weighted cross-covariance of iteration i + 1 only needs the weights iteration i has just updated, so its
    // partial sums ride on iteration i's cost reduction: one workgroup reduction per iteration instead of two (same
    // per-thread order, same reduction tree: the same values).
    double C = 0.0, S = 0.0;
#pragma unroll
    for (int t = 0; t < TPT; t++) {
      if (t * NT < K) {
        C += w[t] * (ax[t] * bx[t] + ay[t] * by[t]);
        S += w[t] * (ax[t] * by[t] - ay[t] * bx[t]);
      }
    }
    red.sum2(C, S);
    for (it = 0; it < p.max_iterations; it++) {
      const double nrm = sqrt(C * C + S * S);
      if (nrm > 0.0) {
        cs = C / nrm;
        sn = S / nrm;
      } else {
        cs = 1.0;
        sn = 0.0;
      }
      double max_r2 = 0.0;
#pragma unroll
      for (int t = 0; t < TPT; t++) {
        if (t * NT < K) {
          const double ex = bx[t] - (cs * ax[t] - sn * ay[t]);
          const double ey = by[t] - (sn * ax[t] + cs * ay[t]);
          r2[t] = ex * ex + ey * ey;
          max_r2 = fmax(max_r2, (tid + t * NT < K) ? r2[t] : 0.0);
        }
      }
      if (it == 0) {
        max_r2 = red.max(max_r2);
        mu = 1.0 / (2.0 * max_r2 / p.c2 - 1.0);
        if (mu <= 0.0) {
          it = 1;
          break;
        }
      }
      double cost = 0.0;
      C = 0.0;
      S = 0.0;
#pragma unroll
      for (int t = 0; t < TPT; t++) {
        if (t * NT < K) {
          cost += w[t] * r2[t];
          if (tid + t * NT < K) w[t] = gnc_weight(r2[t], mu, p.c2);
          C += w[t] * (ax[t] * bx[t] + ay[t] * by[t]);
          S += w[t] * (ax[t] * by[t] - ay[t] * bx[t]);
        }
      }
      red.sum3(cost, C, S);
      const double cost_diff = fabs(cost - prev_cost);
      mu = mu * p.gnc_factor;
      prev_cost = cost;
      if (cost_diff < p.cost_threshold) {
        it++;
        break;
      }
    }
    double cnt = 0.0;
#pragma unroll
    for (int t = 0; t < TPT; t++) cnt += (tid + t * NT < K && w[t] >= 0.5) ? 1.0 : 0.0;
    rot_inliers = (int)red.sum(cnt);
  } else {
    // ---- TIMs on the complete graph: K (K-1) / 2 of them, too many to store.  The weight of a TIM is a function
    // of its residual under the PREVIOUS rotation and the previous mu, so it is recomputed on the fly: every GNC
    // iteration is two sweeps over all pairs i < j (thread = stripe of j), no per-TIM state at all ----
    double cs_w = 1.0, sn_w = 0.0, mu_w = 0.0;  // the rotation / mu the current weights come from (mu_w == 0: all 1)
    auto tim_weight = [&](double axx, double ayy, double bxx, double byy) {
      if (mu_w == 0.0) return 1.0;
      const double ex = bxx - (cs_w * axx - sn_w * ayy), ey = byy - (sn_w * axx + cs_w * ayy);
      return gnc_weight(ex * ex + ey * ey, mu_w, p.c2);
    };
    for (it = 0; it < p.max_iterations; it++) {
      double C = 0.0, S = 0.0;
      for (int i = 0; i < K - 1; i++) {
        const double six = s_src[i].x, siy = s_src[i].y, dix = s_dst[i].x, diy = s_dst[i].y;
        for (int j = i + 1 + tid; j < K; j += NT) {
          const double axx = (double)s_src[j].x - six, ayy = (double)s_src[j].y - siy;
          const double bxx = (double)s_dst[j].x - dix, byy = (double)s_dst[j].y - diy;
          const double w = tim_weight(axx, ayy, bxx, byy);
          C += w * (axx * bxx + ayy * byy);
          S += w * (axx * byy - ayy * bxx);
        }
      }
      red.sum2(C, S);
      const double nrm = sqrt(C * C + S * S);
      if (nrm > 0.0) {
        cs = C / nrm;
        sn = S / nrm;
      } else {
        cs = 1.0;
        sn = 0.0;
      }
      double max_r2 = 0.0, cost = 0.0;
      for (int i = 0; i < K - 1; i++) {
        const double six = s_src[i].x, siy = s_src[i].y, dix = s_dst[i].x, diy = s_dst[i].y;
        for (int j = i + 1 + tid; j < K; j += NT) {
          const double axx = (double)s_src[j].x - six, ayy = (double)s_src[j].y - siy;
          const double bxx = (double)s_dst[j].x - dix, byy = (double)s_dst[j].y - diy;
          const double ex = bxx - (cs * axx - sn * ayy), ey = byy - (sn * axx + cs * ayy);
          const double r2 = ex * ex + ey * ey;
          max_r2 = fmax(max_r2, r2);
          cost += tim_weight(axx, ayy, bxx, byy) * r2;
        }
      }
      if (it == 0) {
        max_r2 = red.max(max_r2);
        mu = 1.0 / (2.0 * max_r2 / p.c2 - 1.0);
        if (mu <= 0.0) {
          it = 1;
          break;
        }
      }
      cost = red.sum(cost);
      cs_w = cs;  // the weights of the next iteration: residuals under this rotation, this mu
      sn_w = sn;
      mu_w = mu;
      const double cost_diff = fabs(cost - prev_cost);
      mu = mu * p.gnc_factor;
      prev_cost = cost;
      if (cost_diff < p.cost_threshold) {
        it++;
        break;
      }
    }
    double cnt = 0.0;
    for (int i = 0; i < K - 1; i++) {
      const double six = s_src[i].x, siy = s_src[i].y, dix = s_dst[i].x, diy = s_dst[i].y;
      for (int j = i + 1 + tid; j < K; j += NT) {
        const double w = tim_weight((double)s_src[j].x - six, (double)s_src[j].y - siy, (double)s_dst[j].x - dix,
                                    (double)s_dst[j].y - diy);
        cnt += w >= 0.5 ? 1.0 : 0.0;
      }
    }
    rot_inliers = (int)red.sum(cnt);
  }

  OPROF(0);  // load + GNC rotation
#if RSX_ORORA_PROF
  if (tid == 0) atomicAdd(&g_orora_prof[6], (unsigned long long)it);
#endif
  // ---- A-COTE translation: residuals and anisotropic bounds, TPT per thread ----
  double vx[TPT], vy[TPT], betx[TPT], bety[TPT];
#pragma unroll
  for (int t = 0; t < TPT; t++) {
    const int i = tid + t * NT;
    vx[t] = vy[t] = 0.0;
    betx[t] = bety[t] = 1.0;
    if (i < K) {
      const double sxx = s_src[i].x, syy = s_src[i].y, dx = s_dst[i].x, dy = s_dst[i].y;
      const double rx = cs * sxx - sn * syy, ry = sn * sxx + cs * syy;
      vx[t] = dx - rx;
      vy[t] = dy - ry;
      double bxx = 0.0, byy = 0.0;
      aniso_bound(dx, dy, p.s_r, p.s_t, bxx, byy);
      aniso_bound(rx, ry, p.s_r, p.s_t, bxx, byy);
      betx[t] = bxx;
      bety[t] = byy;
    }
  }
  OPROF(1);  // residuals + bounds
  // one copy of the estimator's code (it is inlined: taken by reference the TPT-element arrays would live in scratch)
  double tx = 0.0, ty = 0.0;
#pragma unroll 1
  for (int axis = 0; axis < 2; axis++) {
    double xs[TPT], bs[TPT];
#pragma unroll
    for (int t = 0; t < TPT; t++) {
      xs[t] = axis ? vy[t] : vx[t];
      bs[t] = axis ? bety[t] : betx[t];
    }
    const double est = scalar_tls_block<NT, TPT>(xs, bs, K, sx, sb, ev, ei, part, p.teaser_cost);
    if (axis) ty = est;
    else tx = est;
  }
  OPROF(2);  // two scalar TLS estimates
  double cnt = 0.0;
#pragma unroll
  for (int t = 0; t < TPT; t++)
    cnt += (tid + t * NT < K && fabs(vx[t] - tx) <= betx[t] && fabs(vy[t] - ty) <= bety[t]) ? 1.0 : 0.0;
  const int trans_inliers = (int)red.sum(cnt);
  if (tid == 0) {
    rsx_orora_result r;
    r.x = tx;
    r.y = ty;
    r.yaw = atan2(sn, cs);
    r.iterations = it;
    r.rot_inliers = rot_inliers;
    r.trans_inliers = trans_inliers;
    r.status = 0;
    *out = r;
  }
  OPROF(3);
}

// the max-clique inlier selection (csrc/pmc.hip), when the call asked for it: pair i's selected matches lie at
// [offsets[i], offsets[i] + cnt[i]) of the sel arrays, in their original order; cnt[i] < 0 = the pair passed through
// unpruned (fewer than 2 / more than 2048 matches): the solver reads the caller's arrays
struct Selection {
  const float2 *src, *dst;
  const int32_t *cnt;
  __device__ __forceinline__ void apply(int pair, const float2 *&s, const float2 *&d, int64_t &k) const {
    if (cnt) {
      const int c = cnt[pair];
      if (c >= 0) {
        k = c;
        s = src;
        d = dst;
      }
    }
  }
};

using LdsLayout = Layout<256, MAXK_LDS>;
constexpr int LDS_TOTAL = LdsLayout::TOTAL + LDS_RED;
static_assert(2 * LDS_TOTAL <= 160 * 1024, "LDS budget: two workgroups per CU");

// on-chip kernel: one workgroup per pair; pairs that do not fit are written to the big list
__global__ __launch_bounds__(256, 2) void orora_register_kernel(const float2 *__restrict__ src, const float2 *__restrict__ dst,
                                                             const int64_t *__restrict__ offsets, int n_pairs, Params p,
                                                             rsx_orora_result *__restrict__ out, int *__restrict__ big_list, Selection sel) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int pair = blockIdx.x;
  if (pair >= n_pairs) return;
  const int64_t o = offsets[pair];
  int64_t K64 = offsets[pair + 1] - o;
  const float2 *ps = src, *pd = dst;
  sel.apply(pair, ps, pd, K64);
  if (K64 < 2 || K64 > MAXK_LDS) {
    if (threadIdx.x == 0) {
      if (K64 > MAXK_LDS && K64 <= MAXK_BIG) {
        big_list[1 + atomicAdd(big_list, 1)] = pair;  // scored by orora_register_big_kernel
      } else {
        rsx_orora_result r;
        r.x = r.y = r.yaw = 0.0;
        r.iterations = r.rot_inliers = r.trans_inliers = 0;
        r.status = K64 < 2 ? 1 : 2;
        out[pair] = r;
      }
    }
    return;
  }
  register_pair<256, MAXK_LDS>(ps, pd, o, (int)K64, p, smem, reinterpret_cast<double *>(smem + LdsLayout::TOTAL), out + pair);
}

using BigLayout = Layout<1024, MAXK_BIG>;
constexpr int BIG_BLOCKS = 32;

// large pairs: persistent 1024-thread workgroups, arrays in the HBM workspace (BigLayout::TOTAL bytes per workgroup)
__global__ __launch_bounds__(1024) void orora_register_big_kernel(const float2 *__restrict__ src, const float2 *__restrict__ dst,
                                                                  const int64_t *__restrict__ offsets, Params p,
                                                                  rsx_orora_result *__restrict__ out, const int *__restrict__ big_list,
                                                                  char *__restrict__ workspace, Selection sel) {
  __shared__ double red_lds[48];
  const int n_big = big_list[0];
  char *ws = workspace + (size_t)blockIdx.x * BigLayout::TOTAL;
  for (int b = blockIdx.x; b < n_big; b += gridDim.x) {
    const int pair = big_list[1 + b];
    const int64_t o = offsets[pair];
    int64_t K64 = offsets[pair + 1] - o;
    const float2 *ps = src, *pd = dst;
    sel.apply(pair, ps, pd, K64);
    register_pair<1024, MAXK_BIG>(ps, pd, o, (int)K64, p, ws, red_lds, out + pair);
  }
}

}  // namespace

struct rsx_orora {
  int device = 0;
  std::mutex mu;
  hipStream_t stream = nullptr;
  rsx::DevBuf src, dst, off, res;
  rsx::DevBuf big_list, big_ws;  // pairs of more than 2048 matches: their indices, and the HBM arrays of the workgroups that score them
  // RSX_ORORA_PMC: the selection's workspaces (csrc/pmc.hip) and what it hands to the solver
  rsx::pmc::Workspace pmc_ws;
  rsx::DevBuf sel_src, sel_dst, sel_cnt, pmc_info, member;
  int64_t sel_cap = 0;             // matches sel_src / sel_dst hold (rsx_orora_reserve)
  hipStream_t last_stream = nullptr;  // of the last PMC call (rsx_orora_last_pmc_info waits for it)
  int32_t last_pmc_pairs = 0;
  bool attr_set = false;
};

namespace {
int reserve_selection(rsx_orora *h, int64_t total, hipStream_t s) {
  if (total <= h->sel_cap) return RSX_OK;
  RSX_TRY(h->sel_src.reserve((size_t)total * 8, s, false));
  RSX_TRY(h->sel_dst.reserve((size_t)total * 8, s, false));
  h->sel_cap = (int64_t)(h->sel_src.bytes < h->sel_dst.bytes ? h->sel_src.bytes : h->sel_dst.bytes) / 8;
  return RSX_OK;
}
}  // namespace

using rsx::fail;

extern "C" {

int rsx_orora_default_params(rsx_orora_params *p) try {
  if (!p) return fail(RSX_ERR_BAD_ARG, "null params");
  p->tim_noise_bound = 2.0 * 0.75;
  p->noise_bound_radial = 0.3536;
  p->noise_bound_tangential = 1.8 * M_PI / 180.0;
  p->gnc_factor = 1.4;
  p->cost_threshold = 1e-6;
  p->max_iterations = 100;
  p->flags = 0;
  return RSX_OK;
} RSX_CATCH_ALL

int rsx_orora_max_correspondences(void) { return MAXK_BIG; }

int rsx_orora_create(int device, rsx_orora **out) try {
  if (!out) return fail(RSX_ERR_BAD_ARG, "null out");
  *out = nullptr;
  int ndev = rsx_device_count();
  if (ndev <= 0) return fail(RSX_ERR_NO_DEVICE, "no HIP device visible (librsx has no CPU fallback)");
  if (device < 0 || device >= ndev) return fail(RSX_ERR_NO_DEVICE, "device %d out of range (%d visible)", device, ndev);
  rsx_orora *h = new (std::nothrow) rsx_orora();
  if (!h) return fail(RSX_ERR_OOM, "host alloc");
  h->device = device;
  hipError_t e = hipSetDevice(device);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    delete h;
    return fail(RSX_ERR_HIP, "create: %s", hipGetErrorString(e));
  }
  *out = h;
  return RSX_OK;
} RSX_CATCH_ALL

int rsx_orora_destroy(rsx_orora *h) try {
  if (!h) return RSX_OK;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  h->src.release();
  h->dst.release();
  h->off.release();
  h->res.release();
  h->big_list.release();
  h->big_ws.release();
  h->pmc_ws.release();
  for (rsx::DevBuf *b : {&h->sel_src, &h->sel_dst, &h->sel_cnt, &h->pmc_info, &h->member}) b->release();
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return RSX_OK;
} RSX_CATCH_ALL

int rsx_orora_register_batch_device(rsx_orora *h, const float *d_src_xy, const float *d_dst_xy, const int64_t *d_offsets,
                                    int32_t n_pairs, const rsx_orora_params *params, rsx_orora_result *d_out, void *stream) try {
  if (!h || !d_src_xy || !d_dst_xy || !d_offsets || !d_out || n_pairs < 0) return fail(RSX_ERR_BAD_ARG, "bad arg");
  if (n_pairs == 0) return RSX_OK;
  rsx_orora_params dp;
  rsx_orora_default_params(&dp);
  if (params) dp = *params;
  if (dp.max_iterations < 1 || !(dp.gnc_factor > 1.0)) return fail(RSX_ERR_BAD_ARG, "bad GNC params");
  std::lock_guard<std::mutex> lk(h->mu);
  RSX_HIP(hipSetDevice(h->device));
  if (!h->attr_set) {
    RSX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&orora_register_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL));
    h->attr_set = true;
  }
  Params kp;
  kp.c2 = dp.tim_noise_bound * dp.tim_noise_bound;
  if (kp.c2 < 1e-16) kp.c2 = 1e-2;
  kp.s_r = dp.noise_bound_radial;
  kp.s_t = dp.noise_bound_tangential;
  kp.gnc_factor = dp.gnc_factor;
  kp.cost_threshold = dp.cost_threshold;
  kp.max_iterations = dp.max_iterations;
  kp.complete_graph = (dp.flags & RSX_ORORA_COMPLETE_GRAPH) ? 1 : 0;
  kp.teaser_cost = (dp.flags & RSX_ORORA_TEASER_COST) ? 1 : 0;
  hipStream_t s = stream ? static_cast<hipStream_t>(stream) : h->stream;
  // the list of pairs too large for the on-chip kernel is built on the device (the offsets may live there only); the
  // large-pair kernel always runs its few workgroups, which return at once when the list is empty
  RSX_TRY(h->big_list.reserve((size_t)(n_pairs + 1) * sizeof(int), s, false));
  RSX_TRY(h->big_ws.reserve((size_t)BIG_BLOCKS * BigLayout::TOTAL, s, false));
  RSX_HIP(hipMemsetAsync(h->big_list.p, 0, sizeof(int), s));
  Selection sel{nullptr, nullptr, nullptr};
  if (dp.flags & RSX_ORORA_PMC) {
    // max-clique inlier selection first (csrc/pmc.hip): the solver then reads each pair's selected matches
    if (h->sel_cap <= 0) return fail(RSX_ERR_BAD_ARG, "RSX_ORORA_PMC on the device entry needs rsx_orora_reserve(max_total_matches) first");
    RSX_TRY(h->sel_cnt.reserve((size_t)n_pairs * 4, s, false));
    RSX_TRY(h->pmc_info.reserve((size_t)n_pairs * sizeof(rsx_orora_pmc_info), s, false));
    RSX_TRY(rsx::pmc::launch(h->pmc_ws, h->device, reinterpret_cast<const float2 *>(d_src_xy), reinterpret_cast<const float2 *>(d_dst_xy),
                             d_offsets, n_pairs, dp.tim_noise_bound, nullptr, h->pmc_info.as<rsx_orora_pmc_info>(), h->sel_src.as<float2>(),
                             h->sel_dst.as<float2>(), h->sel_cnt.as<int32_t>(), h->sel_cap, s));
    sel = Selection{h->sel_src.as<float2>(), h->sel_dst.as<float2>(), h->sel_cnt.as<int32_t>()};
    h->last_stream = s;
    h->last_pmc_pairs = n_pairs;
  }
  hipLaunchKernelGGL(orora_register_kernel, dim3(n_pairs), dim3(256), LDS_TOTAL, s,
                     reinterpret_cast<const float2 *>(d_src_xy), reinterpret_cast<const float2 *>(d_dst_xy), d_offsets,
                     n_pairs, kp, d_out, h->big_list.as<int>(), sel);
  hipLaunchKernelGGL(orora_register_big_kernel, dim3(BIG_BLOCKS), dim3(1024), 0, s,
                     reinterpret_cast<const float2 *>(d_src_xy), reinterpret_cast<const float2 *>(d_dst_xy), d_offsets, kp, d_out,
                     h->big_list.as<int>(), h->big_ws.as<char>(), sel);
  RSX_HIP(hipGetLastError());
#if RSX_ORORA_PROF
  {
    unsigned long long v[8] = {0}, z[8] = {0};
    RSX_HIP(hipStreamSynchronize(s));
    RSX_HIP(hipMemcpyFromSymbol(v, HIP_SYMBOL(g_orora_prof), sizeof(v)));
    RSX_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_orora_prof), z, sizeof(z)));
    const double n = n_pairs;
    fprintf(stderr, "[orora prof] per pair (cycles): gnc %.0f (%.1f iterations)  bounds %.0f  tls x+y %.0f (sorts %.0f)  rest %.0f\n",
            v[0] / n, v[6] / n, v[1] / n, v[2] / n, v[4] / n, v[3] / n);
  }
#endif
  return RSX_OK;
} RSX_CATCH_ALL

int rsx_orora_register_batch(rsx_orora *h, const float *src_xy, const float *dst_xy, const int64_t *offsets, int32_t n_pairs,
                             const rsx_orora_params *params, rsx_orora_result *out) try {
  if (!h || !src_xy || !dst_xy || !offsets || !out || n_pairs < 0) return fail(RSX_ERR_BAD_ARG, "bad arg");
  if (n_pairs == 0) return RSX_OK;
  const int64_t m = offsets[n_pairs];
  if (m < 0 || offsets[0] != 0) return fail(RSX_ERR_BAD_ARG, "offsets must start at 0 and be non-decreasing");
  {
    std::lock_guard<std::mutex> lk(h->mu);
    RSX_HIP(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    RSX_TRY(h->src.reserve((size_t)(m ? m : 1) * 8, s, false));
    RSX_TRY(h->dst.reserve((size_t)(m ? m : 1) * 8, s, false));
    RSX_TRY(h->off.reserve((size_t)(n_pairs + 1) * 8, s, false));
    RSX_TRY(h->res.reserve((size_t)n_pairs * sizeof(rsx_orora_result), s, false));
    if (params && (params->flags & RSX_ORORA_PMC)) RSX_TRY(reserve_selection(h, m ? m : 1, s));
    if (m) {
      RSX_HIP(hipMemcpyAsync(h->src.p, src_xy, (size_t)m * 8, hipMemcpyHostToDevice, s));
      RSX_HIP(hipMemcpyAsync(h->dst.p, dst_xy, (size_t)m * 8, hipMemcpyHostToDevice, s));
    }
    RSX_HIP(hipMemcpyAsync(h->off.p, offsets, (size_t)(n_pairs + 1) * 8, hipMemcpyHostToDevice, s));
  }
  RSX_TRY(rsx_orora_register_batch_device(h, h->src.as<float>(), h->dst.as<float>(), h->off.as<int64_t>(), n_pairs, params,
                                          h->res.as<rsx_orora_result>(), h->stream));
  std::lock_guard<std::mutex> lk(h->mu);
  RSX_HIP(hipMemcpyAsync(out, h->res.p, (size_t)n_pairs * sizeof(rsx_orora_result), hipMemcpyDeviceToHost, h->stream));
  RSX_HIP(hipStreamSynchronize(h->stream));
  return RSX_OK;
} RSX_CATCH_ALL

int rsx_orora_max_clique_matches(void) { return rsx::pmc::MAX_K; }

int rsx_orora_reserve(rsx_orora *h, int64_t max_total_matches) try {
  if (!h || max_total_matches < 1) return fail(RSX_ERR_BAD_ARG, "bad arg");
  std::lock_guard<std::mutex> lk(h->mu);
  RSX_HIP(hipSetDevice(h->device));
  return reserve_selection(h, max_total_matches, h->stream);
} RSX_CATCH_ALL

int rsx_orora_max_clique_batch_device(rsx_orora *h, const float *d_src_xy, const float *d_dst_xy, const int64_t *d_offsets, int32_t n_pairs,
                                      const rsx_orora_params *params, uint8_t *d_member, rsx_orora_pmc_info *d_info, void *stream) try {
  if (!h || !d_src_xy || !d_dst_xy || !d_offsets || n_pairs < 0) return fail(RSX_ERR_BAD_ARG, "bad arg");
  if (n_pairs == 0) return RSX_OK;
  rsx_orora_params dp;
  rsx_orora_default_params(&dp);
  if (params) dp = *params;
  std::lock_guard<std::mutex> lk(h->mu);
  RSX_HIP(hipSetDevice(h->device));
  hipStream_t s = stream ? static_cast<hipStream_t>(stream) : h->stream;
  return rsx::pmc::launch(h->pmc_ws, h->device, reinterpret_cast<const float2 *>(d_src_xy), reinterpret_cast<const float2 *>(d_dst_xy), d_offsets,
                          n_pairs, dp.tim_noise_bound, d_member, d_info, nullptr, nullptr, nullptr, 0, s);
} RSX_CATCH_ALL

int rsx_orora_max_clique_batch(rsx_orora *h, const float *src_xy, const float *dst_xy, const int64_t *offsets, int32_t n_pairs,
                               const rsx_orora_params *params, uint8_t *out_member, rsx_orora_pmc_info *out_info) try {
  if (!h || !src_xy || !dst_xy || !offsets || n_pairs < 0) return fail(RSX_ERR_BAD_ARG, "bad arg");
  if (n_pairs == 0) return RSX_OK;
  const int64_t m = offsets[n_pairs];
  if (m < 0 || offsets[0] != 0) return fail(RSX_ERR_BAD_ARG, "offsets must start at 0 and be non-decreasing");
  {
    std::lock_guard<std::mutex> lk(h->mu);
    RSX_HIP(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    RSX_TRY(h->src.reserve((size_t)(m ? m : 1) * 8, s, false));
    RSX_TRY(h->dst.reserve((size_t)(m ? m : 1) * 8, s, false));
    RSX_TRY(h->off.reserve((size_t)(n_pairs + 1) * 8, s, false));
    RSX_TRY(h->member.reserve((size_t)(m ? m : 1), s, false));
    RSX_TRY(h->pmc_info.reserve((size_t)n_pairs * sizeof(rsx_orora_pmc_info), s, false));
    if (m) {
      RSX_HIP(hipMemcpyAsync(h->src.p, src_xy, (size_t)m * 8, hipMemcpyHostToDevice, s));
      RSX_HIP(hipMemcpyAsync(h->dst.p, dst_xy, (size_t)m * 8, hipMemcpyHostToDevice, s));
    }
    RSX_HIP(hipMemcpyAsync(h->off.p, offsets, (size_t)(n_pairs + 1) * 8, hipMemcpyHostToDevice, s));
  }
  RSX_TRY(rsx_orora_max_clique_batch_device(h, h->src.as<float>(), h->dst.as<float>(), h->off.as<int64_t>(), n_pairs, params,
                                            h->member.as<uint8_t>(), h->pmc_info.as<rsx_orora_pmc_info>(), h->stream));
  std::lock_guard<std::mutex> lk(h->mu);
  if (out_member && m) RSX_HIP(hipMemcpyAsync(out_member, h->member.p, (size_t)m, hipMemcpyDeviceToHost, h->stream));
  if (out_info) RSX_HIP(hipMemcpyAsync(out_info, h->pmc_info.p, (size_t)n_pairs * sizeof(rsx_orora_pmc_info), hipMemcpyDeviceToHost, h->stream));
  RSX_HIP(hipStreamSynchronize(h->stream));
  return RSX_OK;
} RSX_CATCH_ALL

int rsx_orora_last_pmc_info(rsx_orora *h, rsx_orora_pmc_info *out_info, int32_t n_pairs) try {
  if (!h || !out_info || n_pairs < 0) return fail(RSX_ERR_BAD_ARG, "bad arg");
  std::lock_guard<std::mutex> lk(h->mu);
  if (n_pairs > h->last_pmc_pairs) return fail(RSX_ERR_BAD_ARG, "the last RSX_ORORA_PMC call had %d pairs", h->last_pmc_pairs);
  if (n_pairs == 0) return RSX_OK;
  RSX_HIP(hipSetDevice(h->device));
  RSX_HIP(hipStreamSynchronize(h->last_stream));
  RSX_HIP(hipMemcpy(out_info, h->pmc_info.p, (size_t)n_pairs * sizeof(rsx_orora_pmc_info), hipMemcpyDeviceToHost));
  return RSX_OK;
} RSX_CATCH_ALL

}  // extern "C"
