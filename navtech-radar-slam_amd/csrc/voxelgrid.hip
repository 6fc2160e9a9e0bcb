// voxelgrid.hip -- pcl::VoxelGrid<pcl::PointXYZI> downsample on gfx950: the step immediately before
// the ScanContext build in the reference's keyframe path
// (pgo/SC-A-LOAM/src/laserPosegraphOptimization.cpp:98,482-484: downSizeFilterScancontext, leaf 0.4 m
// set at :687-688).  SURVEY.md 8(f) rank 1.  PCL itself is a third-party dependency that is neither
// vendored in the reference checkout nor installed here, so this follows the published algorithm of
// pcl/filters/impl/voxel_grid.hpp as restated in oracle/voxelgrid_ref.c (PARITY UNPINNED; the
// float additions inside a voxel run in ascending input order, which PCL's unstable std::sort leaves
// open).
//
// Kernel chain (keyframe clouds are 10^3..10^5 points: latency-bound, everything is L2-resident):
//   vg_minmax     finite points -> min / max per axis (order-preserving integer atomics), count
//   vg_setup      one thread: inverse leaf, overflow test, min_b, divb_mul          (float ops as PCL)
//   vg_keys       voxel index per point (0xffffffff for non-finite points), value = input index
//   rocPRIM radix sort of (voxel index, input index) pairs -- stable, so a voxel's points stay in
//   input order
//   vg_heads      first point of every voxel -> flag; rocPRIM exclusive scan -> output slot
//   vg_centroids  the thread of a voxel's first point sums x, y, z, intensity in float, divides by
//                 the count, writes the packed float4
// Clouds of <= 4096 points (every keyframe cloud of the reference's pipeline) take vg_small_kernel instead: the same steps in
// one workgroup and ONE launch, with a bitonic sort in LDS -- no library call on the keyframe path.
#include <hip/hip_runtime.h>

#include <cstring>  // rocprim's texture_cache_iterator.hpp uses memset without including it

#include <rocprim/rocprim.hpp>

#include <cmath>
#include <cstdint>
#include <mutex>
#include <new>

#include "rsx_common.h"
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

__global__ __launch_bounds__(256) void vg_init(VgParams *P) {
  if (threadIdx.x == 0) {
    for (int c = 0; c < 3; c++) {
      P->mn[c] = enc(INFINITY);
      P->mx[c] = enc(-INFINITY);
    }
    P->nvalid = 0;
    P->overflow = 0;
    P->n_out = 0;
  }
}

__global__ __launch_bounds__(256) void vg_minmax(const char *__restrict__ pts, int64_t n, int64_t stride, VgParams *P) {
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  unsigned long long cnt = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float *p = reinterpret_cast<const float *>(pts + i * stride);
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
  if ((threadIdx.x & 63) == 0 && cnt) {
    for (int c = 0; c < 3; c++) {
      atomicMin(&P->mn[c], enc(mn[c]));
      atomicMax(&P->mx[c], enc(mx[c]));
    }
    atomicAdd(&P->nvalid, cnt);
  }
}

__global__ void vg_setup(VgParams *P, float leaf, int64_t n) {
  if (threadIdx.x || blockIdx.x) return;
  if (P->nvalid == 0) {
    P->n_out = 0;
    return;
  }
  const float inv = __fdiv_rn(1.0f, leaf);
  P->inv = inv;
  long long d[3];
  float mn[3], mx[3];
  for (int c = 0; c < 3; c++) {
    mn[c] = dec(P->mn[c]);
    mx[c] = dec(P->mx[c]);
    d[c] = (long long)(__fmul_rn(__fsub_rn(mx[c], mn[c]), inv)) + 1;
  }
  if (d[0] * d[1] * d[2] > 2147483647ll) {  // "Leaf size is too small for the input dataset"
    P->overflow = 1;
    P->n_out = n;
    return;
  }
  int div_b[3];
  for (int c = 0; c < 3; c++) {
    P->min_b[c] = (int)floorf(__fmul_rn(mn[c], inv));
    div_b[c] = (int)floorf(__fmul_rn(mx[c], inv)) - P->min_b[c] + 1;
  }
  P->mul[0] = 1;
  P->mul[1] = div_b[0];
  P->mul[2] = div_b[0] * div_b[1];
}

__global__ __launch_bounds__(256) void vg_keys(const char *__restrict__ pts, int64_t n, int64_t stride, const VgParams *P,
                                               unsigned *__restrict__ keys, unsigned *__restrict__ vals) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float *p = reinterpret_cast<const float *>(pts + i * stride);
  unsigned key = 0xffffffffu;
  if (finite3(p) && !P->overflow) {
    int idx = 0;
    for (int c = 0; c < 3; c++) idx += (int)(__fsub_rn(floorf(__fmul_rn(p[c], P->inv)), (float)P->min_b[c])) * P->mul[c];
    key = (unsigned)idx;
  }
  keys[i] = key;
  vals[i] = (unsigned)i;
}

__global__ __launch_bounds__(256) void vg_heads(const unsigned *__restrict__ keys, int64_t n, unsigned *__restrict__ flags) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const unsigned k = keys[i];
  flags[i] = (k != 0xffffffffu && (i == 0 || keys[i - 1] != k)) ? 1u : 0u;
}

__global__ __launch_bounds__(256) void vg_centroids(const char *__restrict__ pts, int64_t n, int64_t stride, int ioff,
                                                    const unsigned *__restrict__ keys, const unsigned *__restrict__ vals,
                                                    const unsigned *__restrict__ flags, const unsigned *__restrict__ pos,
                                                    float4 *__restrict__ out, int64_t max_out, VgParams *P) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  if (i == n - 1 && !P->overflow) P->n_out = (long long)pos[i] + flags[i];
  if (!flags[i]) return;
  const unsigned k = keys[i];
  float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
  int64_t j = i;
  for (; j < n && keys[j] == k; j++) {
    const char *q = pts + (int64_t)vals[j] * stride;
    const float *p = reinterpret_cast<const float *>(q);
    s0 = __fadd_rn(s0, p[0]);
    s1 = __fadd_rn(s1, p[1]);
    s2 = __fadd_rn(s2, p[2]);
    if (ioff >= 0) s3 = __fadd_rn(s3, *reinterpret_cast<const float *>(q + ioff));
  }
  const float cnt = (float)(j - i);
  const unsigned o = pos[i];
  if ((int64_t)o < max_out) out[o] = make_float4(__fdiv_rn(s0, cnt), __fdiv_rn(s1, cnt), __fdiv_rn(s2, cnt), __fdiv_rn(s3, cnt));
}

// overflow path: output = input (packed), non-finite points included, input order
__global__ __launch_bounds__(256) void vg_copy(const char *__restrict__ pts, int64_t n, int64_t stride, int ioff,
                                               const VgParams *P, float4 *__restrict__ out, int64_t max_out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n || i >= max_out || !P->overflow) return;
  const char *q = pts + i * stride;
  const float *p = reinterpret_cast<const float *>(q);
  out[i] = make_float4(p[0], p[1], p[2], ioff >= 0 ? *reinterpret_cast<const float *>(q + ioff) : 0.0f);
}

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

}  // namespace

struct rsx_voxelgrid {
  int device = 0;
  std::mutex mu;
  hipStream_t stream = nullptr;
  rsx::DevBuf pts, keys, keys2, vals, vals2, flags, pos, out, params, temp;
};

using rsx::fail;

namespace rsx {
namespace vg {

int filter_device(rsx_voxelgrid *h, const void *d_pts, int64_t n, int64_t stride, int32_t ioff, float leaf, int64_t max_out,
                  const float **d_out, int64_t *n_out, hipStream_t s) {
  *d_out = nullptr;
  *n_out = 0;
  if (n <= 0) return RSX_OK;
  if (n > 0x7fffffff) return fail(RSX_ERR_RANGE, "more than 2^31-1 points");
  RSX_TRY(h->params.reserve(sizeof(VgParams), s, false));
  RSX_TRY(h->keys.reserve((size_t)n * 4, s, false));
  RSX_TRY(h->keys2.reserve((size_t)n * 4, s, false));
  RSX_TRY(h->vals.reserve((size_t)n * 4, s, false));
  RSX_TRY(h->vals2.reserve((size_t)n * 4, s, false));
  RSX_TRY(h->flags.reserve((size_t)n * 4, s, false));
  RSX_TRY(h->pos.reserve((size_t)n * 4, s, false));
  RSX_TRY(h->out.reserve((size_t)(max_out > 0 ? max_out : 1) * 16, s, false));
  VgParams *P = h->params.as<VgParams>();
  const char *pts = static_cast<const char *>(d_pts);
  if (n <= VG_SMALL_MAX) {  // a keyframe cloud: one workgroup, one launch (the large path below is 14)
    hipLaunchKernelGGL(vg_small_kernel, dim3(1), dim3(VG_SMALL_NT), 0, s, pts, (int)n, stride, (int)ioff, leaf, h->out.as<float4>(),
                       max_out, P);
    RSX_HIP(hipGetLastError());
    long long cnt = 0;
    RSX_HIP(hipMemcpyAsync(&cnt, &P->n_out, sizeof(cnt), hipMemcpyDeviceToHost, s));
    RSX_HIP(hipStreamSynchronize(s));
    *d_out = h->out.as<float>();
    *n_out = cnt;
    return RSX_OK;
  }
  const unsigned nb = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(vg_init, dim3(1), dim3(64), 0, s, P);
  hipLaunchKernelGGL(vg_minmax, dim3(nb < 1024 ? nb : 1024), dim3(256), 0, s, pts, n, stride, P);
  hipLaunchKernelGGL(vg_setup, dim3(1), dim3(64), 0, s, P, leaf, n);
  hipLaunchKernelGGL(vg_keys, dim3(nb), dim3(256), 0, s, pts, n, stride, P, h->keys.as<unsigned>(), h->vals.as<unsigned>());
  RSX_HIP(hipGetLastError());
  size_t tb1 = 0, tb2 = 0;
  RSX_HIP(rocprim::radix_sort_pairs(nullptr, tb1, h->keys.as<unsigned>(), h->keys2.as<unsigned>(), h->vals.as<unsigned>(),
                                    h->vals2.as<unsigned>(), (size_t)n, 0, 32, s));
  RSX_HIP(rocprim::exclusive_scan(nullptr, tb2, h->flags.as<unsigned>(), h->pos.as<unsigned>(), 0u, (size_t)n,
                                  rocprim::plus<unsigned>(), s));
  RSX_TRY(h->temp.reserve(tb1 > tb2 ? tb1 : tb2, s, false));
  RSX_HIP(rocprim::radix_sort_pairs(h->temp.p, tb1, h->keys.as<unsigned>(), h->keys2.as<unsigned>(), h->vals.as<unsigned>(),
                                    h->vals2.as<unsigned>(), (size_t)n, 0, 32, s));
  hipLaunchKernelGGL(vg_heads, dim3(nb), dim3(256), 0, s, h->keys2.as<unsigned>(), n, h->flags.as<unsigned>());
  RSX_HIP(rocprim::exclusive_scan(h->temp.p, tb2, h->flags.as<unsigned>(), h->pos.as<unsigned>(), 0u, (size_t)n,
                                  rocprim::plus<unsigned>(), s));
  hipLaunchKernelGGL(vg_centroids, dim3(nb), dim3(256), 0, s, pts, n, stride, (int)ioff, h->keys2.as<unsigned>(),
                     h->vals2.as<unsigned>(), h->flags.as<unsigned>(), h->pos.as<unsigned>(), h->out.as<float4>(), max_out, P);
  hipLaunchKernelGGL(vg_copy, dim3(nb), dim3(256), 0, s, pts, n, stride, (int)ioff, P, h->out.as<float4>(), max_out);
  RSX_HIP(hipGetLastError());
  long long cnt = 0;
  RSX_HIP(hipMemcpyAsync(&cnt, &P->n_out, sizeof(cnt), hipMemcpyDeviceToHost, s));
  RSX_HIP(hipStreamSynchronize(s));
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
  for (rsx::DevBuf *b : {&h->pts, &h->keys, &h->keys2, &h->vals, &h->vals2, &h->flags, &h->pos, &h->out, &h->params, &h->temp})
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
