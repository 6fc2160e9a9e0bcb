// cen2019.hip -- cen2019 radar keypoint extraction on gfx950 (polar 400 x 3360 u8 image -> keypoints).
//
// The reference gets these keypoints from its ORORA submodule, which is an empty directory in the
// reference checkout (.gitmodules:1-3; README.md:29 "extracted via cen2019 method"), so this follows
// the published method as restated in oracle/cen2019_ref.c (PARITY UNPINNED).
// It replaces the feature-extraction call of the upstream file-based odometry.cpp entry (README.md:27).
//
// Kernel chain (all HBM-bound image passes except the sort; 1.34 MB u8 in, <= 8 B per keypoint out):
//   cen_stats       bytes -> sum(bytes), max |fft(r+1) - fft(r-1)|              1 read of the image
//   cen_h           h = (fft - mean)(1 - g/maxg) -> h image, fixed-point sum h   1 read, 1 f32 write
//   cen_candidates  h > mean_h -> 64-bit keys (~order(h) << 32 | pixel index)    1 read of h
//   rocPRIM radix sort of the keys (descending h, ties by azimuth, range) and a stable 9-bit
//   sort of the ranks by azimuth: the greedy region marking of the method is sequential in global
//   intensity order, but its state is per-azimuth -- only the region budget couples azimuths
//   cen_mark        one wavefront per azimuth replays that azimuth's candidates in rank order
//                   (LDS mark row), records which candidate marked each pixel and whether the
//                   candidate opened a new region
//   cen_budget      prefix count of "new region" flags in rank order -> J* = number of candidates
//                   the sequential algorithm would have visited before the budget ran out
//   cen_extract     per azimuth: runs of pixels marked by candidates ranked < J*, adjacency test
//                   against the neighbouring azimuths, argmax of h -> keypoints
//   cen_compact     row-major compaction (+ polar -> Cartesian)
// Arithmetic is order independent by construction (integer byte sum, max, 2^40 fixed-point sum of
// h), so the result is bit-identical to the oracle.
#include <hip/hip_runtime.h>

#include <cstring>  // rocprim's texture_cache_iterator.hpp uses memset without including it

#include <rocprim/rocprim.hpp>

#include <cmath>
#include <cstdint>
#include <mutex>
#include <new>

#include "rsx_common.h"

namespace {

constexpr int ROW_CAP = 1024;  // keypoints kept per azimuth (a 3360-bin row can hold at most 1680 runs)
constexpr double FIX = 1099511627776.0;  // 2^40

struct Scal {
  unsigned long long sum_bytes;
  long long fix_sum;
  unsigned int max_g_bits;
  unsigned int n_cand;
  long long jstar;
  unsigned int n_targets;
  unsigned int pad;
};

__device__ __forceinline__ float px(const uint8_t *img, int a, int r, int stride, int off) {
  return __fdiv_rn((float)img[(int64_t)a * stride + off + r], 255.0f);
}

__device__ __forceinline__ unsigned ord_f32(float f) {
  unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ __launch_bounds__(256) void cen_stats(const uint8_t *__restrict__ img, int rows, int cols, int stride, int off,
                                                 Scal *sc) {
  const int64_t n = (int64_t)rows * cols;
  unsigned long long sb = 0;
  float mg = 0.0f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int a = (int)(i / cols), r = (int)(i - (int64_t)a * cols);
    sb += img[(int64_t)a * stride + off + r];
    if (cols > 1) {
      const int rp = (r + 1 < cols) ? r + 1 : cols - 2, rm = (r >= 1) ? r - 1 : 1;  // reflect 101
      mg = fmaxf(mg, fabsf(__fsub_rn(px(img, a, rp, stride, off), px(img, a, rm, stride, off))));
    }
  }
  for (int o = 32; o >= 1; o >>= 1) {
    sb += __shfl_xor(sb, o);
    mg = fmaxf(mg, __shfl_xor(mg, o));
  }
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&sc->sum_bytes, sb);
    atomicMax(&sc->max_g_bits, __float_as_uint(mg));  // non-negative floats order like their bits
  }
}

__device__ __forceinline__ float mean_fft(const Scal *sc, int64_t n) {
  return (float)((double)sc->sum_bytes / 255.0 / (double)n);
}
__device__ __forceinline__ float mean_h(const Scal *sc, int64_t n) { return (float)((double)sc->fix_sum / FIX / (double)n); }

__global__ __launch_bounds__(256) void cen_h(const uint8_t *__restrict__ img, int rows, int cols, int stride, int off,
                                             Scal *sc, float *__restrict__ h) {
  const int64_t n = (int64_t)rows * cols;
  const float mean = mean_fft(sc, n);
  const float maxg = __uint_as_float(sc->max_g_bits);
  long long fix = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int a = (int)(i / cols), r = (int)(i - (int64_t)a * cols);
    float g = 0.0f;
    if (cols > 1) {
      const int rp = (r + 1 < cols) ? r + 1 : cols - 2, rm = (r >= 1) ? r - 1 : 1;
      g = fabsf(__fsub_rn(px(img, a, rp, stride, off), px(img, a, rm, stride, off)));
    }
    const float gn = (maxg > 0.0f) ? __fdiv_rn(g, maxg) : 0.0f;
    const float s = __fsub_rn(px(img, a, r, stride, off), mean);
    const float hv = __fmul_rn(s, __fsub_rn(1.0f, gn));
    h[i] = hv;
    fix += __double2ll_rn((double)hv * FIX);
  }
  for (int o = 32; o >= 1; o >>= 1) fix += __shfl_xor(fix, o);
  if ((threadIdx.x & 63) == 0) atomicAdd(reinterpret_cast<unsigned long long *>(&sc->fix_sum), (unsigned long long)fix);
}

__global__ __launch_bounds__(256) void cen_candidates(const float *__restrict__ h, int rows, int cols, Scal *sc,
                                                      unsigned long long *__restrict__ keys, unsigned *__restrict__ row_count) {
  const int64_t n = (int64_t)rows * cols;
  const float mh = mean_h(sc, n);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float hv = h[i];
    if (hv > mh) {
      const unsigned pos = atomicAdd(&sc->n_cand, 1u);
      keys[pos] = ((unsigned long long)(~ord_f32(hv)) << 32) | (unsigned long long)(unsigned)i;
      atomicAdd(&row_count[(int)(i / cols)], 1u);
    }
  }
}

__global__ __launch_bounds__(256) void cen_rowkeys(const unsigned long long *__restrict__ keys_sorted, const Scal *sc, int cols,
                                                   unsigned *__restrict__ rowkey, unsigned *__restrict__ rank) {
  const unsigned m = sc->n_cand;
  for (unsigned j = blockIdx.x * 256 + threadIdx.x; j < m; j += gridDim.x * 256) {
    rowkey[j] = (unsigned)(keys_sorted[j] & 0xffffffffull) / (unsigned)cols;
    rank[j] = j;
  }
}

// exclusive scan of up to 1024 row counts (single block)
__global__ __launch_bounds__(1024) void cen_row_offsets(const unsigned *__restrict__ row_count, int rows,
                                                        unsigned *__restrict__ row_off) {
  __shared__ unsigned s[1024];
  const int t = threadIdx.x;
  s[t] = t < rows ? row_count[t] : 0u;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    unsigned v = t >= d ? s[t - d] : 0u;
    __syncthreads();
    s[t] += v;
    __syncthreads();
  }
  if (t < rows) row_off[t + 1] = s[t];
  if (t == 0) row_off[0] = 0;
}

// one wavefront per azimuth: replay the azimuth's candidates in global rank order
__global__ __launch_bounds__(64) void cen_mark(const uint8_t *__restrict__ img, int rows, int cols, int stride, int off,
                                               const Scal *sc, const unsigned long long *__restrict__ keys_sorted,
                                               const unsigned *__restrict__ rank_by_row, const unsigned *__restrict__ row_off,
                                               int *__restrict__ mark, uint8_t *__restrict__ inc) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  int *mk = reinterpret_cast<int *>(lds);                        // [cols] rank of the marking candidate
  uint8_t *neg = reinterpret_cast<uint8_t *>(lds) + (size_t)cols * 4;  // [cols] s < 0
  const int a = blockIdx.x, lane = threadIdx.x;
  const float mean = mean_fft(sc, (int64_t)rows * cols);
  for (int r = lane; r < cols; r += 64) {
    mk[r] = 0x7fffffff;
    neg[r] = __fsub_rn(px(img, a, r, stride, off), mean) < 0.0f;
  }
  __syncthreads();
  const unsigned t0 = row_off[a], t1 = row_off[a + 1];
  for (unsigned t = t0; t < t1; t++) {
    const unsigned j = rank_by_row[t];
    const int r = (int)((unsigned)(keys_sorted[j] & 0xffffffffull) - (unsigned)a * (unsigned)cols);
    if (mk[r] != 0x7fffffff) {  // wave-uniform
      if (lane == 0) inc[j] = 0;
      continue;
    }
    // extend over the adjacent s < 0 pixels, 64 at a time (lane 0 = nearest pixel)
    int rlow = r, rhigh = r;
    for (int base = r - 1; base >= 0; base -= 64) {
      const int p = base - lane;
      const unsigned long long b = __ballot(p >= 0 && neg[p]);
      const int run = (~b) ? (__ffsll((long long)~b) - 1) : 64;
      rlow -= run;
      if (run < 64) break;
    }
    for (int base = r + 1; base < cols; base += 64) {
      const int p = base + lane;
      const unsigned long long b = __ballot(p < cols && neg[p]);
      const int run = (~b) ? (__ffsll((long long)~b) - 1) : 64;
      rhigh += run;
      if (run < 64) break;
    }
    bool already = false;
    for (int base = rlow; base <= rhigh; base += 64) {
      const int p = base + lane;
      const bool in = p <= rhigh;
      const bool was = in && mk[p] != 0x7fffffff;
      if (in && !was) mk[p] = (int)j;
      already |= __ballot(was) != 0ull;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) inc[j] = already ? 0 : 1;
  }
  __syncthreads();
  for (int r = lane; r < cols; r += 64) mark[(int64_t)a * cols + r] = mk[r];
}

// J* = number of candidates visited by `while (l < max_points && j < M)`: the first rank whose
// exclusive prefix count of new regions reaches max_points, or M
__global__ __launch_bounds__(1024) void cen_budget(const uint8_t *__restrict__ inc, Scal *sc, int max_points) {
  __shared__ unsigned s[1024];
  const unsigned m = sc->n_cand;
  const unsigned t = threadIdx.x;
  const unsigned chunk = (m + 1023) / 1024;
  const unsigned lo = t * chunk < m ? t * chunk : m;
  const unsigned hi = lo + chunk < m ? lo + chunk : m;
  unsigned c = 0;
  for (unsigned j = lo; j < hi; j++) c += inc[j];
  s[t] = c;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    unsigned v = t >= (unsigned)d ? s[t - d] : 0u;
    __syncthreads();
    s[t] += v;
    __syncthreads();
  }
  const unsigned total = s[1023];
  if (t == 0 && (max_points <= 0 || total < (unsigned)max_points)) sc->jstar = max_points <= 0 ? 0 : (long long)m;
  if (max_points > 0 && total >= (unsigned)max_points) {
    unsigned excl = s[t] - c;  // new regions before this chunk
    if (excl < (unsigned)max_points && excl + c >= (unsigned)max_points) {
      // the candidate that opens region number max_points is the last one visited
      unsigned l = excl;
      for (unsigned j = lo; j < hi; j++) {
        l += inc[j];
        if (l >= (unsigned)max_points) {
          sc->jstar = (long long)j + 1;
          break;
        }
      }
    }
  }
}

__global__ __launch_bounds__(64) void cen_extract(const float *__restrict__ h, const int *__restrict__ mark, int rows, int cols,
                                                  const Scal *sc, int min_range, int *__restrict__ row_out,
                                                  unsigned *__restrict__ row_n) {
  const int a = blockIdx.x * 64 + threadIdx.x;
  if (a >= rows) return;
  const int js = (int)(sc->jstar > 0x7fffffff ? 0x7fffffff : sc->jstar);
  const int *mr = mark + (int64_t)a * cols;
  const int *below = mark + (int64_t)((a - 1 + rows) % rows) * cols;
  const int *above = mark + (int64_t)((a + 1) % rows) * cols;
  const float *hr = h + (int64_t)a * cols;
  int start = 0, end = 0;
  bool counting = false;
  unsigned cnt = 0;
  for (int r = min_range < 0 ? 0 : min_range; r < cols; r++) {
    if (mr[r] < js) {
      if (!counting) {
        start = r;
        counting = true;
      }
      end = r;
    } else if (counting) {
      bool adj = false;
      for (int i = start; i <= end && !adj; i++) adj = (below[i] < js) || (above[i] < js);
      if (adj) {
        int max_r = start;
        float mx = -INFINITY;
        for (int i = start; i <= end; i++)
          if (hr[i] > mx) {
            mx = hr[i];
            max_r = i;
          }
        if (cnt < ROW_CAP) row_out[(int64_t)a * ROW_CAP + cnt] = max_r;
        cnt++;
      }
      counting = false;
    }
  }
  row_n[a] = cnt < ROW_CAP ? cnt : ROW_CAP;
}

__global__ __launch_bounds__(256) void cen_compact(const int *__restrict__ row_out, const unsigned *__restrict__ row_n,
                                                   const unsigned *__restrict__ row_off, int rows, const float *__restrict__ az,
                                                   float resolution, int max_targets, int *__restrict__ targets,
                                                   float *__restrict__ xy, Scal *sc) {
  const int a = blockIdx.x;
  const unsigned n = row_n[a], o = row_off[a];
  for (unsigned i = threadIdx.x; i < n; i += 256) {
    const unsigned d = o + i;
    if (d >= (unsigned)max_targets) continue;
    const int r = row_out[(int64_t)a * ROW_CAP + i];
    targets[2 * d] = a;
    targets[2 * d + 1] = r;
    if (xy && az) {
      const float range = __fmul_rn(__fadd_rn((float)r, 0.5f), resolution);
      xy[2 * d] = __fmul_rn(range, cosf(az[a]));
      xy[2 * d + 1] = __fmul_rn(range, sinf(az[a]));
    }
  }
  if (a == rows - 1 && threadIdx.x == 0) sc->n_targets = o + n;
}

}  // namespace

struct rsx_cen2019 {
  int device = 0, rows = 0, cols = 0;
  std::mutex mu;
  hipStream_t stream = nullptr;
  rsx::DevBuf img, h, keys, keys2, rowkey, rowkey2, rank, rank2, inc, mark, scal, row_count, row_off, row_out, row_n, row_off2,
      targets, xy, az, temp;
  bool attr_set = false;
};

using rsx::fail;

extern "C" {

int rsx_cen2019_default_params(rsx_cen2019_params *p) {
  if (!p) return fail(RSX_ERR_BAD_ARG, "null params");
  p->max_points = 10000;  // yeti_radar_odometry default for cen2019 (recollection)
  p->min_range = 58;
  return RSX_OK;
}

int rsx_cen2019_create(int device, int32_t rows, int32_t cols, rsx_cen2019 **out) {
  if (!out) return fail(RSX_ERR_BAD_ARG, "null out");
  *out = nullptr;
  if (rows < 1 || rows > 1024 || cols < 2 || cols > 16384) return fail(RSX_ERR_BAD_ARG, "image shape %d x %d unsupported", rows, cols);
  int ndev = rsx_device_count();
  if (ndev <= 0) return fail(RSX_ERR_NO_DEVICE, "no HIP device visible (librsx has no CPU fallback)");
  if (device < 0 || device >= ndev) return fail(RSX_ERR_NO_DEVICE, "device %d out of range (%d visible)", device, ndev);
  rsx_cen2019 *h = new (std::nothrow) rsx_cen2019();
  if (!h) return fail(RSX_ERR_OOM, "host alloc");
  h->device = device;
  h->rows = rows;
  h->cols = cols;
  hipError_t e = hipSetDevice(device);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    delete h;
    return fail(RSX_ERR_HIP, "create: %s", hipGetErrorString(e));
  }
  *out = h;
  return RSX_OK;
}

int rsx_cen2019_destroy(rsx_cen2019 *h) {
  if (!h) return RSX_OK;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  for (rsx::DevBuf *b : {&h->img, &h->h, &h->keys, &h->keys2, &h->rowkey, &h->rowkey2, &h->rank, &h->rank2, &h->inc, &h->mark,
                         &h->scal, &h->row_count, &h->row_off, &h->row_out, &h->row_n, &h->row_off2, &h->targets, &h->xy, &h->az,
                         &h->temp})
    b->release();
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return RSX_OK;
}

// d_img: device image; results stay on the device (d_targets int32 pairs, d_xy optional); *d_count
// is the Scal.n_targets word read back by the host wrapper
static int extract_device(rsx_cen2019 *h, const uint8_t *d_img, int32_t stride, int32_t off, const rsx_cen2019_params &p,
                          const float *d_az, float resolution, int32_t max_targets, hipStream_t s) {
  const int rows = h->rows, cols = h->cols;
  const int64_t n = (int64_t)rows * cols;
  RSX_TRY(h->h.reserve((size_t)n * 4, s, false));
  RSX_TRY(h->keys.reserve((size_t)n * 8, s, false));
  RSX_TRY(h->keys2.reserve((size_t)n * 8, s, false));
  RSX_TRY(h->rowkey.reserve((size_t)n * 4, s, false));
  RSX_TRY(h->rowkey2.reserve((size_t)n * 4, s, false));
  RSX_TRY(h->rank.reserve((size_t)n * 4, s, false));
  RSX_TRY(h->rank2.reserve((size_t)n * 4, s, false));
  RSX_TRY(h->inc.reserve((size_t)n, s, false));
  RSX_TRY(h->mark.reserve((size_t)n * 4, s, false));
  RSX_TRY(h->scal.reserve(sizeof(Scal), s, false));
  RSX_TRY(h->row_count.reserve((size_t)rows * 4, s, false));
  RSX_TRY(h->row_off.reserve((size_t)(rows + 1) * 4, s, false));
  RSX_TRY(h->row_off2.reserve((size_t)(rows + 1) * 4, s, false));
  RSX_TRY(h->row_out.reserve((size_t)rows * ROW_CAP * 4, s, false));
  RSX_TRY(h->row_n.reserve((size_t)rows * 4, s, false));
  RSX_TRY(h->targets.reserve((size_t)(max_targets > 0 ? max_targets : 1) * 8, s, false));
  RSX_TRY(h->xy.reserve((size_t)(max_targets > 0 ? max_targets : 1) * 8, s, false));
  Scal *sc = h->scal.as<Scal>();
  RSX_HIP(hipMemsetAsync(sc, 0, sizeof(Scal), s));
  RSX_HIP(hipMemsetAsync(h->row_count.p, 0, (size_t)rows * 4, s));
  const int grid = 2048;
  hipLaunchKernelGGL(cen_stats, dim3(grid), dim3(256), 0, s, d_img, rows, cols, stride, off, sc);
  hipLaunchKernelGGL(cen_h, dim3(grid), dim3(256), 0, s, d_img, rows, cols, stride, off, sc, h->h.as<float>());
  hipLaunchKernelGGL(cen_candidates, dim3(grid), dim3(256), 0, s, h->h.as<float>(), rows, cols, sc,
                     h->keys.as<unsigned long long>(), h->row_count.as<unsigned>());
  RSX_HIP(hipGetLastError());
  // the candidate count is needed on the host to size the sorts
  unsigned m = 0;
  RSX_HIP(hipMemcpyAsync(&m, &sc->n_cand, sizeof(unsigned), hipMemcpyDeviceToHost, s));
  RSX_HIP(hipStreamSynchronize(s));
  if (m > 0) {
    size_t tb1 = 0, tb2 = 0;
    RSX_HIP(rocprim::radix_sort_keys(nullptr, tb1, h->keys.as<unsigned long long>(), h->keys2.as<unsigned long long>(), m, 0, 64, s));
    RSX_HIP(rocprim::radix_sort_pairs(nullptr, tb2, h->rowkey.as<unsigned>(), h->rowkey2.as<unsigned>(), h->rank.as<unsigned>(),
                                      h->rank2.as<unsigned>(), m, 0, 10, s));
    RSX_TRY(h->temp.reserve(tb1 > tb2 ? tb1 : tb2, s, false));
    RSX_HIP(rocprim::radix_sort_keys(h->temp.p, tb1, h->keys.as<unsigned long long>(), h->keys2.as<unsigned long long>(), m, 0, 64, s));
    hipLaunchKernelGGL(cen_rowkeys, dim3(1024), dim3(256), 0, s, h->keys2.as<unsigned long long>(), sc, cols,
                       h->rowkey.as<unsigned>(), h->rank.as<unsigned>());
    RSX_HIP(rocprim::radix_sort_pairs(h->temp.p, tb2, h->rowkey.as<unsigned>(), h->rowkey2.as<unsigned>(), h->rank.as<unsigned>(),
                                      h->rank2.as<unsigned>(), m, 0, 10, s));
  }
  hipLaunchKernelGGL(cen_row_offsets, dim3(1), dim3(1024), 0, s, h->row_count.as<unsigned>(), rows, h->row_off.as<unsigned>());
  const size_t lds = (size_t)cols * 5;
  if (!h->attr_set) {
    RSX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&cen_mark), hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 5));
    h->attr_set = true;
  }
  hipLaunchKernelGGL(cen_mark, dim3(rows), dim3(64), lds, s, d_img, rows, cols, stride, off, sc, h->keys2.as<unsigned long long>(),
                     h->rank2.as<unsigned>(), h->row_off.as<unsigned>(), h->mark.as<int>(), h->inc.as<uint8_t>());
  hipLaunchKernelGGL(cen_budget, dim3(1), dim3(1024), 0, s, h->inc.as<uint8_t>(), sc, p.max_points);
  hipLaunchKernelGGL(cen_extract, dim3((rows + 63) / 64), dim3(64), 0, s, h->h.as<float>(), h->mark.as<int>(), rows, cols, sc,
                     p.min_range, h->row_out.as<int>(), h->row_n.as<unsigned>());
  hipLaunchKernelGGL(cen_row_offsets, dim3(1), dim3(1024), 0, s, h->row_n.as<unsigned>(), rows, h->row_off2.as<unsigned>());
  hipLaunchKernelGGL(cen_compact, dim3(rows), dim3(256), 0, s, h->row_out.as<int>(), h->row_n.as<unsigned>(),
                     h->row_off2.as<unsigned>(), rows, d_az, resolution, max_targets, h->targets.as<int>(),
                     d_az ? h->xy.as<float>() : nullptr, sc);
  RSX_HIP(hipGetLastError());
  return RSX_OK;
}

int rsx_cen2019_extract(rsx_cen2019 *h, const uint8_t *img, int32_t row_stride, int32_t col_offset,
                        const rsx_cen2019_params *params, const float *azimuths, float resolution, int32_t *out_targets,
                        float *out_xy, int32_t max_targets, int32_t *out_count) {
  if (!h || !img || !out_targets || !out_count || max_targets < 0) return fail(RSX_ERR_BAD_ARG, "bad arg");
  if (col_offset < 0 || row_stride < col_offset + h->cols) return fail(RSX_ERR_BAD_ARG, "row_stride %d too small for offset %d + %d columns", row_stride, col_offset, h->cols);
  if (out_xy && !azimuths) return fail(RSX_ERR_BAD_ARG, "out_xy needs azimuths");
  rsx_cen2019_params p;
  rsx_cen2019_default_params(&p);
  if (params) p = *params;
  std::lock_guard<std::mutex> lk(h->mu);
  RSX_HIP(hipSetDevice(h->device));
  hipStream_t s = h->stream;
  const size_t bytes = (size_t)h->rows * row_stride;
  RSX_TRY(h->img.reserve(bytes, s, false));
  RSX_HIP(hipMemcpyAsync(h->img.p, img, bytes, hipMemcpyHostToDevice, s));
  const float *d_az = nullptr;
  if (azimuths) {
    RSX_TRY(h->az.reserve((size_t)h->rows * 4, s, false));
    RSX_HIP(hipMemcpyAsync(h->az.p, azimuths, (size_t)h->rows * 4, hipMemcpyHostToDevice, s));
    d_az = h->az.as<float>();
  }
  RSX_TRY(extract_device(h, h->img.as<uint8_t>(), row_stride, col_offset, p, d_az, resolution, max_targets, s));
  unsigned n = 0;
  RSX_HIP(hipMemcpyAsync(&n, &h->scal.as<Scal>()->n_targets, sizeof(unsigned), hipMemcpyDeviceToHost, s));
  RSX_HIP(hipStreamSynchronize(s));
  const unsigned w = n < (unsigned)max_targets ? n : (unsigned)max_targets;
  if (w) {
    RSX_HIP(hipMemcpyAsync(out_targets, h->targets.p, (size_t)w * 8, hipMemcpyDeviceToHost, s));
    if (out_xy) RSX_HIP(hipMemcpyAsync(out_xy, h->xy.p, (size_t)w * 8, hipMemcpyDeviceToHost, s));
    RSX_HIP(hipStreamSynchronize(s));
  }
  *out_count = (int32_t)n;
  return RSX_OK;
}

}  // extern "C"
