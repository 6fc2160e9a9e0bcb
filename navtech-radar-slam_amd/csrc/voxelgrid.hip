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
