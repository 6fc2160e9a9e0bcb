// voxelgrid.hip -- pcl::VoxelGrid<pcl::PointXYZI> downsample on gfx950: the step immediately before
// the ScanContext build in the reference's keyframe path
// (pgo/SC-A-LOAM/src/laserPosegraphOptimization.cpp:98,482-484: downSizeFilterScancontext, leaf 0.4 m
// set at :687-688).  SURVEY.md 8(f) rank 1.  PCL itself is a third-party dependency that is neither
// vendored in the reference checkout nor installed here, so this follows the published algorithm of
// pcl/filters/impl/voxel_grid.hpp as restated in oracle/voxelgrid_ref.c (PARITY UNPINNED; the
// float additions inside a voxel run in ascending input order, which PCL's unstable std::sort leaves
// open).
//
// Two kernels, each ONE launch per cloud:
//   vg_small_kernel  clouds of <= 4096 points (every keyframe cloud of the reference's pipeline): one workgroup -- min / max,
//                    setup, voxel keys, a bitonic sort of (voxel index << 32 | input index) in LDS, heads, scan, centroids;
//   vg_coop_kernel   anything larger (the 51-keyframe submaps of loop verification, the map cloud), and up to TWO clouds per
//                    launch: a persistent kernel of <= 128 workgroups per cloud that meet at grid barriers
//                    (rsx_grid_dev.h).  An optional rigid transform (local2global, PGO.cpp:199-220) is applied to every
//                    point as it is read, so "transform the submap, then VoxelGrid it" is one pass over the store:
//        min / max / count of the finite points (per-workgroup partials, every workgroup reduces them: deterministic)
//        setup (float ops as PCL; every workgroup computes it for itself)
//        LSD radix sort of (voxel index, input index), 8 bits a pass, only as many passes as the grid's volume has bits:
//        per-workgroup digit counts -> barrier -> every workgroup scans the counts it needs -> stable scatter (wavefronts
//        walk their sub-tiles in order; ranks inside a 64-element chunk by eight ballots) -> barrier.  Stable, so a voxel's
//        points stay in input order -- the order their floats are added in.
//        heads -> per-workgroup counts -> barrier -> output slots; the thread of a voxel's first point sums its points
//   Rounds 1-4 ran this as 14 launches around two rocPRIM calls (radix_sort_pairs, exclusive_scan) and a host
//   synchronisation for the count; the count now stays on the device for whoever runs next (icp.hip).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <new>

#include "rsx_common.h"
#include "rsx_grid_dev.h"
#include "rsx_persistent.h"
#include "voxelgrid.h"

namespace {

struct VgParams {
  unsigned mn[3], mx[3];  // order-preserving encodings of the float min / max
  unsigned long long nvalid;
  float inv;
  int min_b[3];
  int mul[3];
  int overflow;
  long long n_out;
};

__device__ __forceinline__ unsigned enc(float f) {
  unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float dec(unsigned u) {
  u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
  return __uint_as_float(u);
}
__device__ __forceinline__ bool finite3(const float *p) { return isfinite(p[0]) && isfinite(p[1]) && isfinite(p[2]); }

// ------------------------------------------------------------------------------------------
// Clouds of up to VG_SMALL_MAX points (a keyframe cloud is ~10^3): the whole chain in ONE workgroup and one launch --
// min / max, setup, keys, a bitonic sort of (voxel index << 32 | input index) in LDS (the input index in the low half keeps a
// voxel's points in input order, as the stable radix sort of the large path does), heads, a block scan for the output slots,
// centroids.  Same float operations in the same order as the kernels above, so the same bits out.
// ------------------------------------------------------------------------------------------
constexpr int VG_SMALL_MAX = 4096;
constexpr int VG_SMALL_NT = 1024;

__global__ __launch_bounds__(VG_SMALL_NT) void vg_small_kernel(const char *__restrict__ pts, int n, int64_t stride, int ioff, float leaf,
                                                               float4 *__restrict__ out, int64_t max_out, VgParams *P) {
  __shared__ unsigned long long sk[VG_SMALL_MAX];
  __shared__ unsigned s_red[6][VG_SMALL_NT / 64];
  __shared__ unsigned s_cnt[VG_SMALL_NT / 64];
  __shared__ unsigned s_pos[VG_SMALL_MAX];
  __shared__ VgParams sp;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  // ---- min / max / count of the finite points ----
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  unsigned cnt = 0;
  for (int i = t; i < n; i += VG_SMALL_NT) {
    const float *p = reinterpret_cast<const float *>(pts + (int64_t)i * stride);
    if (!finite3(p)) continue;
    for (int c = 0; c < 3; c++) {
      mn[c] = fminf(mn[c], p[c]);
      mx[c] = fmaxf(mx[c], p[c]);
    }
    cnt++;
  }
  for (int o = 32; o >= 1; o >>= 1) {
    for (int c = 0; c < 3; c++) {
      mn[c] = fminf(mn[c], __shfl_xor(mn[c], o));
      mx[c] = fmaxf(mx[c], __shfl_xor(mx[c], o));
    }
    cnt += __shfl_xor(cnt, o);
  }
  if (lane == 0) {
    for (int c = 0; c < 3; c++) {
      s_red[c][w] = enc(mn[c]);
      s_red[3 + c][w] = enc(mx[c]);
    }
    s_cnt[w] = cnt;
  }
  __syncthreads();
  if (t == 0) {  // vg_setup, on the reduced values
    unsigned long long nv = 0;
    for (int c = 0; c < 3; c++) {
      unsigned a = enc(INFINITY), b = enc(-INFINITY);
      for (int ww = 0; ww < VG_SMALL_NT / 64; ww++) {
        a = s_red[c][ww] < a ? s_red[c][ww] : a;
        b = s_red[3 + c][ww] > b ? s_red[3 + c][ww] : b;
      }
      sp.mn[c] = a;
      sp.mx[c] = b;
    }
    for (int ww = 0; ww < VG_SMALL_NT / 64; ww++) nv += s_cnt[ww];
    sp.nvalid = nv;
    sp.overflow = 0;
    sp.n_out = 0;
    if (nv) {
      const float inv = __fdiv_rn(1.0f, leaf);
      sp.inv = inv;
      long long d[3];
      float fmn[3], fmx[3];
      for (int c = 0; c < 3; c++) {
        fmn[c] = dec(sp.mn[c]);
        fmx[c] = dec(sp.mx[c]);
        d[c] = (long long)(__fmul_rn(__fsub_rn(fmx[c], fmn[c]), inv)) + 1;
      }
      if (d[0] * d[1] * d[2] > 2147483647ll) {  // "Leaf size is too small for the input dataset"
        sp.overflow = 1;
        sp.n_out = n;
      } else {
        int div_b[3];
        for (int c = 0; c < 3; c++) {
          sp.min_b[c] = (int)floorf(__fmul_rn(fmn[c], inv));
          div_b[c] = (int)floorf(__fmul_rn(fmx[c], inv)) - sp.min_b[c] + 1;
        }
        sp.mul[0] = 1;
        sp.mul[1] = div_b[0];
        sp.mul[2] = div_b[0] * div_b[1];
      }
    }
  }
  __syncthreads();
  if (sp.nvalid == 0 || sp.overflow) {
    if (sp.overflow)  // output = input (packed), non-finite points included, input order
      for (int i = t; i < n && i < max_out; i += VG_SMALL_NT) {
        const char *q = pts + (int64_t)i * stride;
        const float *p = reinterpret_cast<const float *>(q);
        out[i] = make_float4(p[0], p[1], p[2], ioff >= 0 ? *reinterpret_cast<const float *>(q + ioff) : 0.0f);
      }
    if (t == 0) *P = sp;
    return;
  }
  // ---- keys (a power of two of them: the padding sorts to the end) ----
  int m = 1;
  while (m < n) m <<= 1;
  for (int i = t; i < m; i += VG_SMALL_NT) {
    unsigned key = 0xffffffffu;
    if (i < n) {
      const float *p = reinterpret_cast<const float *>(pts + (int64_t)i * stride);
      if (finite3(p)) {
        int idx = 0;
        for (int c = 0; c < 3; c++) idx += (int)(__fsub_rn(floorf(__fmul_rn(p[c], sp.inv)), (float)sp.min_b[c])) * sp.mul[c];
        key = (unsigned)idx;
      }
    }
    sk[i] = ((unsigned long long)key << 32) | (unsigned)i;
  }
  __syncthreads();
  // ---- bitonic sort, ascending ----
  for (int k = 2; k <= m; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = t; i < m; i += VG_SMALL_NT) {
        const int l = i ^ j;
        if (l > i) {
          const unsigned long long a = sk[i], b = sk[l];
          const bool up = (i & k) == 0;
          if ((a > b) == up) {
            sk[i] = b;
            sk[l] = a;
          }
        }
      }
      __syncthreads();
    }
  }
  // ---- heads -> output slots (exclusive scan over the sorted positions, VG_SMALL_MAX / VG_SMALL_NT per thread) ----
  constexpr int PER = VG_SMALL_MAX / VG_SMALL_NT;
  unsigned flag[PER], run = 0;
#pragma unroll
  for (int e = 0; e < PER; e++) {
    const int i = t * PER + e;
    unsigned f = 0;
    if (i < n) {
      const unsigned k = (unsigned)(sk[i] >> 32);
      f = (k != 0xffffffffu && (i == 0 || (unsigned)(sk[i - 1] >> 32) != k)) ? 1u : 0u;
    }
    flag[e] = f;
    run += f;
  }
  unsigned incl = run;
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned v = __shfl_up(incl, o);
    if (lane >= o) incl += v;
  }
  if (lane == 63) s_cnt[w] = incl;
  __syncthreads();
  unsigned base = incl - run;
  for (int ww = 0; ww < w; ww++) base += s_cnt[ww];
  unsigned total = 0;
  for (int ww = 0; ww < VG_SMALL_NT / 64; ww++) total += s_cnt[ww];
#pragma unroll
  for (int e = 0; e < PER; e++) {
    const int i = t * PER + e;
    if (i < VG_SMALL_MAX) s_pos[i] = base;
    base += flag[e];
  }
  __syncthreads();
  // ---- centroids: the thread of a voxel's first point sums its points in input order ----
#pragma unroll
  for (int e = 0; e < PER; e++) {
    const int i = t * PER + e;
    if (!flag[e]) continue;
    const unsigned k = (unsigned)(sk[i] >> 32);
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    int j = i;
    for (; j < n && (unsigned)(sk[j] >> 32) == k; j++) {
      const char *q = pts + (int64_t)(unsigned)(sk[j] & 0xffffffffu) * stride;
      const float *p = reinterpret_cast<const float *>(q);
      s0 = __fadd_rn(s0, p[0]);
      s1 = __fadd_rn(s1, p[1]);
      s2 = __fadd_rn(s2, p[2]);
      if (ioff >= 0) s3 = __fadd_rn(s3, *reinterpret_cast<const float *>(q + ioff));
    }
    const float c = (float)(j - i);
    const unsigned o = s_pos[i];
    if ((int64_t)o < max_out) out[o] = make_float4(__fdiv_rn(s0, c), __fdiv_rn(s1, c), __fdiv_rn(s2, c), __fdiv_rn(s3, c));
  }
  if (t == 0) {
    sp.n_out = total;
    *P = sp;
  }
}


// ------------------------------------------------------------------------------------------
// vg_coop_kernel: one or two clouds, any size, one launch (file header).
// ------------------------------------------------------------------------------------------
constexpr int CO_NT = 256;   // threads per workgroup: four wavefronts, each walks a quarter of the workgroup's tile in order
constexpr int CO_MAX_W = 128;  // workgroups per cloud

struct CoJob {
  const char *pts;  // float x, y, z at byte offsets 0, 4, 8 of each stride, intensity at ioff (< 0: none)
  long long n, stride;
  int ioff, has_T;
  rsx::vg::Mat34 T;  // applied to every point as it is read (has_T)
  float leaf;
  long long max_out;
  float4 *out;
  VgParams *P;
  unsigned *keys0, *keys1, *vals0, *vals1;  // n each: the sort's two buffers
  unsigned *hist;                            // [W][256] digit counts of the workgroups (digit-minor: a wavefront's load is 256 contiguous bytes)
  unsigned *part;                            // [W][8]: min x y z, max x y z (encoded), finite points, heads
  unsigned *bar;                             // grid barrier counters of this cloud (rsx_grid_dev.h)
  int wg_first, W;
};

struct CoShared {
  VgParams sp;
  unsigned invalid_key;  // key of a non-finite point: above every voxel index
  int passes;
};

__device__ __forceinline__ float4 co_point(const CoJob &J, long long i) {
  const char *q = J.pts + i * J.stride;
  const float *p = reinterpret_cast<const float *>(q);
  const float4 v = make_float4(p[0], p[1], p[2], J.ioff >= 0 ? *reinterpret_cast<const float *>(q + J.ioff) : 0.0f);
  if (!J.has_T) return v;
  // local2global (PGO.cpp:210-217): every product and sum in float, left to right, no contraction
  const float *m = J.T.m;
  float4 o;
  o.x = m[0] * v.x + m[1] * v.y + m[2] * v.z + m[3];
  o.y = m[4] * v.x + m[5] * v.y + m[6] * v.z + m[7];
  o.z = m[8] * v.x + m[9] * v.y + m[10] * v.z + m[11];
  o.w = v.w;
  return o;
}
__device__ __forceinline__ bool co_finite(const float4 &p) { return isfinite(p.x) && isfinite(p.y) && isfinite(p.z); }
__device__ __forceinline__ unsigned co_key(const CoShared &S, const float4 &p) {
  if (!co_finite(p)) return S.invalid_key;
  const VgParams &sp = S.sp;
  int idx = (int)(__fsub_rn(floorf(__fmul_rn(p.x, sp.inv)), (float)sp.min_b[0])) * sp.mul[0];
  idx += (int)(__fsub_rn(floorf(__fmul_rn(p.y, sp.inv)), (float)sp.min_b[1])) * sp.mul[1];
  idx += (int)(__fsub_rn(floorf(__fmul_rn(p.z, sp.inv)), (float)sp.min_b[2])) * sp.mul[2];
  return (unsigned)idx;
}

__device__ void co_run(const CoJob &J, unsigned wg) {
  using namespace rsx;
  __shared__ CoShared S;
  __shared__ unsigned s_wh[4][256];  // digit counts of the four wavefronts, then their running output positions
  __shared__ unsigned s_red[8][4];
  __shared__ unsigned s_w4[4];
  __shared__ unsigned s_tmp[CO_MAX_W];
  __shared__ unsigned s_base, s_total;
  grid::Member m{J.bar, wg, (unsigned)J.W, 0u, false};
  // (a barrier that gives up -- rsx_grid_dev.h -- ends the cloud with n_out = -1: the host reports it)
  auto gave_up = [&]() {
    if (wg == 0 && threadIdx.x == 0) {
      S.sp.n_out = -1;
      *J.P = S.sp;
    }
    grid::exit(m);
  };
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
#ifdef RSX_VG_TIMING
  unsigned long long tmk[18] = {0}, tlast = wall_clock64();
  int tmi = 0;
#define VG_MARK()                                 \
  {                                               \
    const unsigned long long n_ = wall_clock64(); \
    if (tmi < 18) tmk[tmi++] = n_ - tlast;        \
    tlast = n_;                                   \
  }
#else
#define VG_MARK()
#endif
  const long long n = J.n;
  const long long per = ((n + J.W - 1) / J.W + CO_NT - 1) / CO_NT * CO_NT;  // the workgroup's tile, a multiple of 256
  const long long lo = (long long)wg * per < n ? (long long)wg * per : n, hi = lo + per < n ? lo + per : n;
  const long long sub = per / 4;                                          // a wavefront's share, a multiple of 64
  const long long wlo = lo + w * sub < hi ? lo + w * sub : hi, whi = wlo + sub < hi ? wlo + sub : hi;
  // ---- min / max / count of the finite points: per-workgroup partials ----
  {
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    unsigned cnt = 0;
    for (long long i = lo + t; i < hi; i += CO_NT) {
      const float4 p = co_point(J, i);
      if (!co_finite(p)) continue;
      mn[0] = fminf(mn[0], p.x);
      mn[1] = fminf(mn[1], p.y);
      mn[2] = fminf(mn[2], p.z);
      mx[0] = fmaxf(mx[0], p.x);
      mx[1] = fmaxf(mx[1], p.y);
      mx[2] = fmaxf(mx[2], p.z);
      cnt++;
    }
    for (int o = 32; o >= 1; o >>= 1) {
      for (int c = 0; c < 3; c++) {
        mn[c] = fminf(mn[c], __shfl_xor(mn[c], o));
        mx[c] = fmaxf(mx[c], __shfl_xor(mx[c], o));
      }
      cnt += __shfl_xor(cnt, o);
    }
    if (lane == 0) {
      for (int c = 0; c < 3; c++) {
        s_red[c][w] = enc(mn[c]);
        s_red[3 + c][w] = enc(mx[c]);
      }
      s_red[6][w] = cnt;
    }
    __syncthreads();
    if (t < 7) {
      unsigned v = s_red[t][0];
      for (int ww = 1; ww < 4; ww++) v = t < 3 ? (s_red[t][ww] < v ? s_red[t][ww] : v) : (t < 6 ? (s_red[t][ww] > v ? s_red[t][ww] : v) : v + s_red[t][ww]);
      grid::st(J.part + (size_t)wg * 8 + t, v);
    }
  }
  VG_MARK()
  if (!grid::sync(m)) {
    gave_up();
    return;
  }
  VG_MARK()
  // ---- setup: every workgroup reduces the partials and computes the grid for itself (vg_setup's float operations) ----
  // (the partials are read by W threads at once, seven loads in flight each)
  {
    unsigned pv[7];
#pragma unroll
    for (int c = 0; c < 7; c++) pv[c] = t < J.W ? grid::ld(J.part + (size_t)t * 8 + c) : (c < 3 ? enc(INFINITY) : (c < 6 ? enc(-INFINITY) : 0u));
    if (t < 128) {  // (W <= 128: two wavefronts)
      for (int o = 32; o >= 1; o >>= 1)
#pragma unroll
        for (int c = 0; c < 7; c++) {
          const unsigned x = __shfl_xor(pv[c], o);
          pv[c] = c < 3 ? (x < pv[c] ? x : pv[c]) : (c < 6 ? (x > pv[c] ? x : pv[c]) : pv[c] + x);
        }
      if (lane == 0)
#pragma unroll
        for (int c = 0; c < 7; c++) s_red[c][w] = pv[c];
    }
  }
  __syncthreads();
  if (t == 0) {
    VgParams sp;
    const unsigned long long nv = (unsigned long long)s_red[6][0] + s_red[6][1];
    for (int c = 0; c < 3; c++) {
      sp.mn[c] = s_red[c][0] < s_red[c][1] ? s_red[c][0] : s_red[c][1];
      sp.mx[c] = s_red[3 + c][0] > s_red[3 + c][1] ? s_red[3 + c][0] : s_red[3 + c][1];
    }
    sp.nvalid = nv;
    sp.overflow = 0;
    sp.n_out = 0;
    sp.inv = 0.0f;
    for (int c = 0; c < 3; c++) sp.min_b[c] = sp.mul[c] = 0;
    unsigned invalid = 1u;
    int passes = 1;
    if (nv) {
      const float inv = __fdiv_rn(1.0f, J.leaf);
      sp.inv = inv;
      long long d[3];
      float fmn[3], fmx[3];
      for (int c = 0; c < 3; c++) {
        fmn[c] = dec(sp.mn[c]);
        fmx[c] = dec(sp.mx[c]);
        d[c] = (long long)(__fmul_rn(__fsub_rn(fmx[c], fmn[c]), inv)) + 1;
      }
      if (d[0] * d[1] * d[2] > 2147483647ll) {  // "Leaf size is too small for the input dataset"
        sp.overflow = 1;
        sp.n_out = n;
      } else {
        long long div_b[3];
        for (int c = 0; c < 3; c++) {
          sp.min_b[c] = (int)floorf(__fmul_rn(fmn[c], inv));
          div_b[c] = (long long)((int)floorf(__fmul_rn(fmx[c], inv)) - sp.min_b[c] + 1);
        }
        sp.mul[0] = 1;
        sp.mul[1] = (int)div_b[0];
        sp.mul[2] = (int)(div_b[0] * div_b[1]);
        long long vol = div_b[0] * div_b[1] * div_b[2];  // voxel indices are < vol
        if (vol < 1) vol = 1;
        int nbits = 0;
        while (nbits < 31 && ((long long)1 << nbits) < vol) nbits++;
        invalid = 1u << nbits;  // >= vol: above every voxel index
        passes = (nbits + 1 + 7) / 8;
      }
    }
    S.sp = sp;
    S.invalid_key = invalid;
    S.passes = passes;
  }
  __syncthreads();
  if (S.sp.nvalid == 0 || S.sp.overflow) {
    if (S.sp.overflow)  // output = input (packed), non-finite points included, input order
      for (long long i = lo + t; i < hi && i < J.max_out; i += CO_NT) J.out[i] = co_point(J, i);
    if (wg == 0 && t == 0) *J.P = S.sp;
    grid::exit(m);
    return;
  }
  VG_MARK()
  // ---- LSD radix sort of (voxel index, input index), 8 bits a pass ----
  // A wavefront's share of up to CO_CH chunks of 64 elements (clouds of up to ~260 000 points) is read ONCE per pass, all
  // loads in flight together, and kept in registers for the count and the scatter; larger shares are read chunk by chunk,
  // twice.  (First build: chunk by chunk for everyone -- a memory round trip per chunk and phase, 30 us a pass.)
  constexpr int CO_CH = 8;
  const bool cached = sub <= (long long)CO_CH * 64;  // (the same for every workgroup of the cloud)
  const int passes = S.passes;
  for (int p = 0; p < passes; p++) {
    const int shift = 8 * p;
    const unsigned *kin = (p & 1) ? J.keys0 : J.keys1, *vin = (p & 1) ? J.vals0 : J.vals1;  // pass p reads what pass p - 1 wrote
    unsigned *kout = (p & 1) ? J.keys1 : J.keys0, *vout = (p & 1) ? J.vals1 : J.vals0;
    auto fetch = [&](long long e, unsigned &key, unsigned &val) {
      key = p == 0 ? co_key(S, co_point(J, e)) : grid::ld(kin + e);
      val = p == 0 ? (unsigned)e : grid::ld(vin + e);
    };
    unsigned rk[CO_CH], rv[CO_CH];
    for (int i = t; i < 4 * 256; i += CO_NT) (&s_wh[0][0])[i] = 0u;
    if (cached) {
#pragma unroll
      for (int c = 0; c < CO_CH; c++) {
        const long long e = wlo + c * 64 + lane;
        rk[c] = rv[c] = 0u;
        if (e < whi) fetch(e, rk[c], rv[c]);
      }
    }
    __syncthreads();
    if (cached) {
#pragma unroll
      for (int c = 0; c < CO_CH; c++)
        if (wlo + c * 64 + lane < whi) atomicAdd(&s_wh[w][(rk[c] >> shift) & 255u], 1u);
    } else {
      for (long long e0 = wlo; e0 < whi; e0 += 64) {
        const long long e = e0 + lane;
        if (e < whi) {
          unsigned key, val;
          fetch(e, key, val);
          atomicAdd(&s_wh[w][(key >> shift) & 255u], 1u);
        }
      }
    }
    __syncthreads();
    grid::st(J.hist + (size_t)wg * 256 + t, s_wh[0][t] + s_wh[1][t] + s_wh[2][t] + s_wh[3][t]);
    VG_MARK()
    if (!grid::sync(m)) {
      gave_up();
      return;
    }
    VG_MARK()
    {  // where digit t of this workgroup starts: all smaller digits of everyone, digit t of the workgroups before
      // (32 loads in flight at a time: a coherent load is ~2 us, one after the other they were 17 us a pass)
      unsigned before = 0, total = 0;
      for (int g0 = 0; g0 < J.W; g0 += 32) {
        unsigned v[32];
#pragma unroll
        for (int u = 0; u < 32; u++) v[u] = g0 + u < J.W ? grid::ld(J.hist + (size_t)(g0 + u) * 256 + t) : 0u;
#pragma unroll
        for (int u = 0; u < 32; u++) {
          total += v[u];
          before += g0 + u < (int)wg ? v[u] : 0u;
        }
      }
      unsigned incl = total;
      for (int o = 1; o < 64; o <<= 1) {
        const unsigned v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
      }
      if (lane == 63) s_w4[w] = incl;
      __syncthreads();
      unsigned base = incl - total + before;
      for (int ww = 0; ww < w; ww++) base += s_w4[ww];
      for (int ww = 0; ww < 4; ww++) {
        const unsigned c = s_wh[ww][t];
        s_wh[ww][t] = base;
        base += c;
      }
    }
    __syncthreads();
    // the wavefront's elements in order, 64 at a time: ranks inside a chunk by eight ballots, the chunk's first element of a
    // digit moves that digit's position on
    auto scatter = [&](bool active, unsigned key, unsigned val) {
      const unsigned digit = (key >> shift) & 255u;
      unsigned long long peers = __ballot(active);
#pragma unroll
      for (int bit = 0; bit < 8; bit++) {
        const bool one = (digit >> bit) & 1u;
        const unsigned long long bal = __ballot(active && one);
        peers &= one ? bal : ~bal;
      }
      unsigned base = 0;
      if (active) base = s_wh[w][digit];
      __builtin_amdgcn_wave_barrier();
      const unsigned long long below = peers & ((1ull << lane) - 1ull);
      if (active && below == 0ull) s_wh[w][digit] = base + (unsigned)__popcll(peers);
      __builtin_amdgcn_wave_barrier();
      if (active) {
        const unsigned dst = base + (unsigned)__popcll(below);
        grid::st(kout + dst, key);
        grid::st(vout + dst, val);
      }
    };
    if (cached) {
#pragma unroll
      for (int c = 0; c < CO_CH; c++)
        if (wlo + c * 64 < whi) scatter(wlo + c * 64 + lane < whi, rk[c], rv[c]);  // (uniform)
    } else {
      for (long long e0 = wlo; e0 < whi; e0 += 64) {
        const long long e = e0 + lane;
        unsigned key = 0, val = 0;
        if (e < whi) fetch(e, key, val);
        scatter(e < whi, key, val);
      }
    }
    VG_MARK()
    if (!grid::sync(m)) {
      gave_up();
      return;
    }
  }
  const unsigned *K = ((passes - 1) & 1) ? J.keys1 : J.keys0, *V = ((passes - 1) & 1) ? J.vals1 : J.vals0;
  // ---- heads: the first point of every voxel; their number per workgroup -> output slots ----
  auto head_of = [&](long long e) -> bool {
    const unsigned k = grid::ld(K + e);
    return k != S.invalid_key && (e == 0 || grid::ld(K + e - 1) != k);
  };
  unsigned long long hb[CO_CH];  // (cached) the chunks' head masks
  {
    unsigned cnt = 0;
    if (cached) {
      unsigned k0[CO_CH], k1[CO_CH];
#pragma unroll
      for (int c = 0; c < CO_CH; c++) {
        const long long e = wlo + c * 64 + lane;
        k0[c] = k1[c] = S.invalid_key;
        if (e < whi) {
          k0[c] = grid::ld(K + e);
          if (e > 0) k1[c] = grid::ld(K + e - 1);
        }
      }
#pragma unroll
      for (int c = 0; c < CO_CH; c++) {
        const long long e = wlo + c * 64 + lane;
        hb[c] = __ballot(e < whi && k0[c] != S.invalid_key && (e == 0 || k1[c] != k0[c]));
        cnt += (unsigned)__popcll(hb[c]);
      }
    } else {
      for (long long e0 = wlo; e0 < whi; e0 += 64) {
        const long long e = e0 + lane;
        cnt += (unsigned)__popcll(__ballot(e < whi && head_of(e)));
      }
    }
    if (lane == 0) s_w4[w] = cnt;
    __syncthreads();
    if (t == 0) grid::st(J.part + (size_t)wg * 8 + 7, s_w4[0] + s_w4[1] + s_w4[2] + s_w4[3]);
  }
  VG_MARK()
  if (!grid::sync(m)) {
    gave_up();
    return;
  }
  VG_MARK()
  if (t < J.W) s_tmp[t] = grid::ld(J.part + (size_t)t * 8 + 7);
  __syncthreads();
  if (t == 0) {
    unsigned b = 0, tot = 0;
    for (int g = 0; g < J.W; g++) {
      if (g == (int)wg) b = tot;
      tot += s_tmp[g];
    }
    s_base = b;
    s_total = tot;
  }
  __syncthreads();
  {
    unsigned pos = s_base;
    for (int ww = 0; ww < w; ww++) pos += s_w4[ww];
    // the rest of a voxel from sorted position j on (the voxel has `have` points so far), four (key, index) pairs per round trip
    auto finish = [&](unsigned k, long long j, float &s0, float &s1, float &s2, float &s3) -> long long {
      bool more = true;
      while (more) {
        unsigned kk[4], vv[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const long long jj = j + u < n ? j + u : n - 1;
          kk[u] = grid::ld(K + jj);
          vv[u] = grid::ld(V + jj);
        }
        float4 q[4];
#pragma unroll
        for (int u = 0; u < 4; u++) q[u] = co_point(J, (long long)vv[u]);
#pragma unroll
        for (int u = 0; u < 4; u++) {
          if (more && j < n && kk[u] == k) {
            s0 = __fadd_rn(s0, q[u].x);
            s1 = __fadd_rn(s1, q[u].y);
            s2 = __fadd_rn(s2, q[u].z);
            if (J.ioff >= 0) s3 = __fadd_rn(s3, q[u].w);
            j++;
          } else {
            more = false;
          }
        }
      }
      return j;
    };
    // one chunk of 64 sorted positions: EVERY lane fetches its own (key, index, point) -- one round trip for the whole chunk --
    // and the lane of a voxel's first point adds the points of the lanes after it, in order, through the wavefront (a voxel
    // with 40 points was ten round trips of one thread: the kernel's tail).  A voxel that runs past the chunk goes on as before.
    auto centroid_chunk = [&](long long e, unsigned long long heads) {
      const bool in = e < whi;
      unsigned k = S.invalid_key, v = 0;
      if (in) {
        k = grid::ld(K + e);
        v = grid::ld(V + e);
      }
      const bool valid = in && k != S.invalid_key;
      float4 q = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      if (valid) q = co_point(J, (long long)v);
      const bool head = (heads >> lane) & 1ull;
      // the points of this lane's voxel inside the chunk: up to the next head or the first non-finite point (the voxel ends
      // there), or up to the end of the wavefront's share / of the chunk (the voxel may go on: `finish` looks)
      const unsigned long long above = ~((2ull << lane) - 1ull);  // (lane 63: empty)
      const unsigned long long ends = (heads | __ballot(in && !valid)) & above, outside = ~__ballot(in) & above;
      const int p_end = ends ? __builtin_ctzll(ends) : 64, p_out = outside ? __builtin_ctzll(outside) : 64;
      const int seglen = (p_end < p_out ? p_end : p_out) - lane;
      const bool open_end = p_out <= p_end;  // nothing inside the chunk closed the voxel
      float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
      int longest = head ? seglen : 0;
      for (int o = 32; o >= 1; o >>= 1) {
        const int x = __shfl_xor(longest, o);
        longest = x > longest ? x : longest;
      }
      for (int d = 0; d < longest; d++) {  // (uniform)
        const int srcl = lane + d < 64 ? lane + d : 63;
        const float x = __shfl(q.x, srcl), y = __shfl(q.y, srcl), z = __shfl(q.z, srcl), wv = __shfl(q.w, srcl);
        if (head && d < seglen) {
          s0 = __fadd_rn(s0, x);
          s1 = __fadd_rn(s1, y);
          s2 = __fadd_rn(s2, z);
          if (J.ioff >= 0) s3 = __fadd_rn(s3, wv);
        }
      }
      if (head) {
        const unsigned o = pos + (unsigned)__popcll(heads & ((1ull << lane) - 1ull));
        long long j = e + seglen;
        if (open_end && j < n) j = finish(k, j, s0, s1, s2, s3);  // the voxel may go on in the next chunk / share / tile
        const float c = (float)(j - e);
        if ((long long)o < J.max_out) J.out[o] = make_float4(__fdiv_rn(s0, c), __fdiv_rn(s1, c), __fdiv_rn(s2, c), __fdiv_rn(s3, c));
      }
      pos += (unsigned)__popcll(heads);
    };
    auto centroid = [&](long long e, bool head, unsigned long long bal) {  // (shares too large for the register path: one thread per voxel)
      if (head) {
        const unsigned o = pos + (unsigned)__popcll(bal & ((1ull << lane) - 1ull));
        const unsigned k = grid::ld(K + e);
        float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
        const long long j = finish(k, e, s0, s1, s2, s3);
        const float c = (float)(j - e);
        if ((long long)o < J.max_out) J.out[o] = make_float4(__fdiv_rn(s0, c), __fdiv_rn(s1, c), __fdiv_rn(s2, c), __fdiv_rn(s3, c));
      }
      pos += (unsigned)__popcll(bal);
    };
    if (cached) {
#pragma unroll
      for (int c = 0; c < CO_CH; c++)
        if (wlo + c * 64 < whi) centroid_chunk(wlo + c * 64 + lane, hb[c]);
    } else {
      for (long long e0 = wlo; e0 < whi; e0 += 64) {
        const long long e = e0 + lane;
        const bool head = e < whi && head_of(e);
        centroid(e, head, __ballot(head));
      }
    }
  }
#ifdef RSX_VG_TIMING
  VG_MARK()
  if (wg == 0 && t == 0 && J.n > 20000)
    printf("vg timing (10 ns): %llu %llu %llu | %llu %llu %llu %llu | %llu %llu %llu %llu | %llu %llu %llu %llu | heads %llu %llu %llu  W %d passes %d\n", tmk[0], tmk[1], tmk[2], tmk[3], tmk[4], tmk[5],
           tmk[6], tmk[7], tmk[8], tmk[9], tmk[10], tmk[11], tmk[12], tmk[13], tmk[14], tmk[15], tmk[16], tmk[17], J.W, passes);
#endif
  if (wg == 0 && t == 0) {
    S.sp.n_out = (long long)s_total;
    *J.P = S.sp;
  }
  grid::exit(m);
}

__global__ __launch_bounds__(CO_NT) void vg_coop_kernel(CoJob a, CoJob b) {
  if ((int)blockIdx.x < a.W)
    co_run(a, blockIdx.x);
  else
    co_run(b, blockIdx.x - (unsigned)a.W);
}

}  // namespace

struct rsx_voxelgrid {
  int device = 0;
  std::mutex mu;
  hipStream_t stream = nullptr;
  rsx::DevBuf pts, keys, keys2, vals, vals2, hist, part, bar, out, params;
};

using rsx::fail;

namespace rsx {
namespace vg {

namespace {

// workgroups per cloud: one per 1024 points, at most CO_MAX_W -- and, two clouds per launch, never more than HALF of what the
// device keeps resident at once (rsx_persistent.h: occupancy query x CU count; a CU-masked or partitioned device shrinks it)
int resident_half(int device) {
  static int cache[64];
  const int d = (device < 0 || device >= 64) ? 0 : device;
  if (!cache[d]) {
    const int lim = rsx::persistent::resident_limit(reinterpret_cast<const void *>(&vg_coop_kernel), CO_NT, 0, device);
    cache[d] = lim < 2 ? 1 : lim / 2;
  }
  return cache[d];
}
int workgroups_for(int64_t n, int device) {
  int64_t w = (n + 1023) / 1024;
  const int cap = resident_half(device) < CO_MAX_W ? resident_half(device) : CO_MAX_W;
  return (int)(w < 1 ? 1 : (w > cap ? cap : w));
}

// buffers of one cloud's job; the barrier counters are zeroed when the buffer is first allocated (the kernel leaves them zero)
int prepare_job(rsx_voxelgrid *h, const void *d_pts, int64_t n, int64_t stride, int32_t ioff, const Mat34 *T, float leaf, int64_t max_out,
                hipStream_t s, CoJob *J) {
  if (n > 0x7fffffff) return fail(RSX_ERR_RANGE, "more than 2^31-1 points");
  const int W = workgroups_for(n, h->device);
  RSX_TRY(h->params.reserve(sizeof(VgParams), s, false));
  RSX_TRY(h->keys.reserve((size_t)n * 4 + 16, s, false));
  RSX_TRY(h->keys2.reserve((size_t)n * 4 + 16, s, false));
  RSX_TRY(h->vals.reserve((size_t)n * 4 + 16, s, false));
  RSX_TRY(h->vals2.reserve((size_t)n * 4 + 16, s, false));
  RSX_TRY(h->hist.reserve((size_t)256 * CO_MAX_W * 4, s, false));
  RSX_TRY(h->part.reserve((size_t)CO_MAX_W * 8 * 4, s, false));
  RSX_TRY(h->out.reserve((size_t)(max_out > 0 ? max_out : 1) * 16, s, false));
  if (!h->bar.p) {
    RSX_TRY(h->bar.reserve(rsx::grid::BYTES, s, false));
    RSX_HIP(hipMemsetAsync(h->bar.p, 0, rsx::grid::BYTES, s));
  }
  J->pts = static_cast<const char *>(d_pts);
  J->n = n;
  J->stride = stride;
  J->ioff = (int)ioff;
  J->has_T = T ? 1 : 0;
  if (T) J->T = *T;
  else std::memset(&J->T, 0, sizeof(J->T));
  J->leaf = leaf;
  J->max_out = max_out;
  J->out = h->out.as<float4>();
  J->P = h->params.as<VgParams>();
  J->keys0 = h->keys.as<unsigned>();
  J->keys1 = h->keys2.as<unsigned>();
  J->vals0 = h->vals.as<unsigned>();
  J->vals1 = h->vals2.as<unsigned>();
  J->hist = h->hist.as<unsigned>();
  J->part = h->part.as<unsigned>();
  J->bar = h->bar.as<unsigned>();
  J->wg_first = 0;
  J->W = W;
  return RSX_OK;
}

}  // namespace

int enqueue(const JobIn *jobs, int njobs, hipStream_t s, DeviceCloud *out) {
  if (njobs < 1 || njobs > 2) return fail(RSX_ERR_BAD_ARG, "one or two clouds per launch");
  CoJob J[2];
  std::memset(J, 0, sizeof(J));
  for (int k = 0; k < njobs; k++) {
    const JobIn &in = jobs[k];
    if (in.n <= 0) return fail(RSX_ERR_BAD_ARG, "empty cloud");
    RSX_TRY(prepare_job(in.h, in.d_pts, in.n, in.stride, in.ioff, in.T, in.leaf, in.max_out, s, &J[k]));
    out[k].d_out = in.h->out.as<float>();
    out[k].d_count = &in.h->params.as<VgParams>()->n_out;
  }
  if (njobs == 2) J[1].wg_first = J[0].W;
  // one grid-barrier kernel at a time per device, ordered on the device (rsx_persistent.h; icp.hip takes the same gate)
  rsx::persistent::Gate gate(jobs[0].h->device, s);
  RSX_TRY(gate.status());
  hipLaunchKernelGGL(vg_coop_kernel, dim3((unsigned)(J[0].W + J[1].W)), dim3(CO_NT), 0, s, J[0], J[1]);
  RSX_HIP(hipGetLastError());
  return RSX_OK;
}

int filter_device(rsx_voxelgrid *h, const void *d_pts, int64_t n, int64_t stride, int32_t ioff, float leaf, int64_t max_out,
                  const float **d_out, int64_t *n_out, hipStream_t s) {
  *d_out = nullptr;
  *n_out = 0;
  if (n <= 0) return RSX_OK;
  if (n > 0x7fffffff) return fail(RSX_ERR_RANGE, "more than 2^31-1 points");
  const char *pts = static_cast<const char *>(d_pts);
  if (n <= VG_SMALL_MAX) {  // a keyframe cloud: one workgroup
    RSX_TRY(h->params.reserve(sizeof(VgParams), s, false));
    RSX_TRY(h->out.reserve((size_t)(max_out > 0 ? max_out : 1) * 16, s, false));
    hipLaunchKernelGGL(vg_small_kernel, dim3(1), dim3(VG_SMALL_NT), 0, s, pts, (int)n, stride, (int)ioff, leaf, h->out.as<float4>(),
                       max_out, h->params.as<VgParams>());
    RSX_HIP(hipGetLastError());
  } else {
    JobIn in{h, d_pts, n, stride, ioff, nullptr, leaf, max_out};
    DeviceCloud dc;
    RSX_TRY(enqueue(&in, 1, s, &dc));
  }
  long long cnt = 0;
  RSX_HIP(hipMemcpyAsync(&cnt, &h->params.as<VgParams>()->n_out, sizeof(cnt), hipMemcpyDeviceToHost, s));
  RSX_HIP(hipStreamSynchronize(s));
  if (cnt < 0) return fail(RSX_ERR_HIP, "a grid barrier of the cooperative VoxelGrid kernel gave up after 5 s: its workgroups were not all resident");
  *d_out = h->out.as<float>();
  *n_out = cnt;
  return RSX_OK;
}

int upload_and_filter(rsx_voxelgrid *h, const void *pts, int64_t n, int64_t stride, int32_t ioff, float leaf, int64_t max_out,
                      const float **d_out, int64_t *n_out) {
  RSX_HIP(hipSetDevice(h->device));
  hipStream_t s = h->stream;
  if (n > 0) {
    RSX_TRY(h->pts.reserve((size_t)n * stride, s, false));
    RSX_HIP(hipMemcpyAsync(h->pts.p, pts, (size_t)n * stride, hipMemcpyHostToDevice, s));
  }
  return filter_device(h, h->pts.p, n, stride, ioff, leaf, max_out, d_out, n_out, s);
}

std::mutex &mutex_of(rsx_voxelgrid *h) { return h->mu; }
hipStream_t stream_of(rsx_voxelgrid *h) { return h->stream; }
int device_of(rsx_voxelgrid *h) { return h->device; }

}  // namespace vg
}  // namespace rsx

extern "C" {

int rsx_voxelgrid_create(int device, rsx_voxelgrid **out) try {
  if (!out) return fail(RSX_ERR_BAD_ARG, "null out");
  *out = nullptr;
  int ndev = rsx_device_count();
  if (ndev <= 0) return fail(RSX_ERR_NO_DEVICE, "no HIP device visible (librsx has no CPU fallback)");
  if (device < 0 || device >= ndev) return fail(RSX_ERR_NO_DEVICE, "device %d out of range (%d visible)", device, ndev);
  rsx_voxelgrid *h = new (std::nothrow) rsx_voxelgrid();
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

int rsx_voxelgrid_destroy(rsx_voxelgrid *h) try {
  if (!h) return RSX_OK;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  for (rsx::DevBuf *b : {&h->pts, &h->keys, &h->keys2, &h->vals, &h->vals2, &h->hist, &h->part, &h->bar, &h->out, &h->params})
    b->release();
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return RSX_OK;
} RSX_CATCH_ALL

int rsx_voxelgrid_filter(rsx_voxelgrid *h, const void *pts, size_t n, size_t stride_bytes, int32_t intensity_offset, float leaf,
                         float *out_xyzi, int64_t max_out, int64_t *out_count) try {
  if (!h || (!pts && n) || !out_count || (!out_xyzi && max_out > 0) || max_out < 0) return fail(RSX_ERR_BAD_ARG, "bad arg");
  if (stride_bytes < 12 || (stride_bytes & 3)) return fail(RSX_ERR_BAD_ARG, "stride_bytes must be >= 12 and a multiple of 4");
  if (intensity_offset >= 0 && ((intensity_offset & 3) || (size_t)intensity_offset + 4 > stride_bytes))
    return fail(RSX_ERR_BAD_ARG, "intensity_offset outside the point");
  if (!(leaf > 0.0f) || !std::isfinite(leaf)) return fail(RSX_ERR_BAD_ARG, "leaf must be positive");
  std::lock_guard<std::mutex> lk(h->mu);
  const float *d_out = nullptr;
  int64_t cnt = 0;
  RSX_TRY(rsx::vg::upload_and_filter(h, pts, (int64_t)n, (int64_t)stride_bytes, intensity_offset, leaf, max_out, &d_out, &cnt));
  const int64_t w = cnt < max_out ? cnt : max_out;
  if (w > 0) {
    RSX_HIP(hipMemcpyAsync(out_xyzi, d_out, (size_t)w * 16, hipMemcpyDeviceToHost, h->stream));
    RSX_HIP(hipStreamSynchronize(h->stream));
  }
  *out_count = cnt;
  return RSX_OK;
} RSX_CATCH_ALL

}  // extern "C"
