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
  unsigned int done;     // the region budget ran out (or every candidate was visited): J* is final
  unsigned int regions;  // new regions opened by the candidate windows processed so far
  unsigned int pad;
};

__device__ __forceinline__ float px(const uint8_t *img, int a, int r, int stride, int off) {
  return __fdiv_rn((float)img[(int64_t)a * stride + off + r], 255.0f);
}

__device__ __forceinline__ unsigned ord_f32(float f) {
  unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// one block per azimuth: the row is 3.3 KB and stays in L1, no index arithmetic per pixel
__global__ __launch_bounds__(256) void cen_stats(const uint8_t *__restrict__ img, int rows, int cols, int stride, int off,
                                                 Scal *sc) {
  __shared__ unsigned long long s_sum[4];
  __shared__ float s_max[4];
  const int a = blockIdx.x;
  const uint8_t *row = img + (int64_t)a * stride + off;
  unsigned long long sb = 0;
  float mg = 0.0f;
  for (int r = threadIdx.x; r < cols; r += 256) {
    sb += row[r];
    if (cols > 1) {
      const int rp = (r + 1 < cols) ? r + 1 : cols - 2, rm = (r >= 1) ? r - 1 : 1;  // reflect 101
      mg = fmaxf(mg, fabsf(__fsub_rn(__fdiv_rn((float)row[rp], 255.0f), __fdiv_rn((float)row[rm], 255.0f))));
    }
  }
  for (int o = 32; o >= 1; o >>= 1) {
    sb += __shfl_xor(sb, o);
    mg = fmaxf(mg, __shfl_xor(mg, o));
  }
  if ((threadIdx.x & 63) == 0) {
    s_sum[threadIdx.x >> 6] = sb;
    s_max[threadIdx.x >> 6] = mg;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(&sc->sum_bytes, s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3]);
    atomicMax(&sc->max_g_bits, __float_as_uint(fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3]))));  // non-negative floats order like their bits
  }
}

__device__ __forceinline__ float mean_fft(const Scal *sc, int64_t n) {
  return (float)((double)sc->sum_bytes / 255.0 / (double)n);
}
__device__ __forceinline__ float mean_h(const Scal *sc, int64_t n) { return (float)((double)sc->fix_sum / FIX / (double)n); }

__global__ __launch_bounds__(256) void cen_h(const uint8_t *__restrict__ img, int rows, int cols, int stride, int off,
                                             Scal *sc, float *__restrict__ h) {
  __shared__ long long s_fix[4];
  const int64_t n = (int64_t)rows * cols;
  const float mean = mean_fft(sc, n);
  const float maxg = __uint_as_float(sc->max_g_bits);
  const int a = blockIdx.x;
  const uint8_t *row = img + (int64_t)a * stride + off;
  float *hrow = h + (int64_t)a * cols;
  long long fix = 0;
  for (int r = threadIdx.x; r < cols; r += 256) {
    float g = 0.0f;
    if (cols > 1) {
      const int rp = (r + 1 < cols) ? r + 1 : cols - 2, rm = (r >= 1) ? r - 1 : 1;
      g = fabsf(__fsub_rn(__fdiv_rn((float)row[rp], 255.0f), __fdiv_rn((float)row[rm], 255.0f)));
    }
    const float gn = (maxg > 0.0f) ? __fdiv_rn(g, maxg) : 0.0f;
    const float sv = __fsub_rn(__fdiv_rn((float)row[r], 255.0f), mean);
    const float hv = __fmul_rn(sv, __fsub_rn(1.0f, gn));
    hrow[r] = hv;
    fix += __double2ll_rn((double)hv * FIX);
  }
  for (int o = 32; o >= 1; o >>= 1) fix += __shfl_xor(fix, o);
  if ((threadIdx.x & 63) == 0) s_fix[threadIdx.x >> 6] = fix;
  __syncthreads();
  if (threadIdx.x == 0)
    atomicAdd(reinterpret_cast<unsigned long long *>(&sc->fix_sum), (unsigned long long)(s_fix[0] + s_fix[1] + s_fix[2] + s_fix[3]));
}

// one block per azimuth; two walks over the row (it stays in L1): count, ONE atomic per block to
// reserve the block's slice of the key list, then write.  Key order is irrelevant (sorted next).
__global__ __launch_bounds__(256) void cen_candidates(const float *__restrict__ h, int rows, int cols, Scal *sc,
                                                      unsigned long long *__restrict__ keys, unsigned *__restrict__ row_count) {
  __shared__ unsigned s_cnt[4];
  __shared__ unsigned s_base;
  const int64_t n = (int64_t)rows * cols;
  const float mh = mean_h(sc, n);
  const int a = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float *hrow = h + (int64_t)a * cols;
  unsigned mine = 0;
  for (int base = wave * 64; base < cols; base += 256) {
    const int r = base + lane;
    mine += (unsigned)__popcll(__ballot(r < cols && hrow[r] > mh));
  }
  if (lane == 0) s_cnt[wave] = mine;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned tot = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    row_count[a] = tot;
    s_base = tot ? atomicAdd(&sc->n_cand, tot) : 0u;
  }
  __syncthreads();
  unsigned pos = s_base;
  for (int w = 0; w < wave; w++) pos += s_cnt[w];
  for (int base = wave * 64; base < cols; base += 256) {
    const int r = base + lane;
    const float hv = r < cols ? hrow[r] : 0.0f;
    const bool is = r < cols && hv > mh;
    const unsigned long long bal = __ballot(is);
    if (is)
      keys[pos + (unsigned)__popcll(bal & ((1ull << lane) - 1ull))] =
          ((unsigned long long)(~ord_f32(hv)) << 32) | (unsigned long long)(unsigned)(a * cols + r);
    pos += (unsigned)__popcll(bal);
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

// one wavefront per azimuth: replay the azimuth's candidates in global rank order.
// The sequential method stops when max_points regions have been opened, usually after a few percent
// of the candidates, so the replay runs in rank WINDOWS [win_lo, win_hi): after each window
// cen_budget counts the regions opened so far and sets sc->done once the budget is exhausted; later
// windows return at once.  Between windows the row's marks live in `mark` and its position in the
// (rank-sorted) candidate list in row_cur.
__global__ __launch_bounds__(64) void cen_mark(const uint8_t *__restrict__ img, int rows, int cols, int stride, int off,
                                               const Scal *sc, const unsigned long long *__restrict__ keys_sorted,
                                               const unsigned *__restrict__ rank_by_row, const unsigned *__restrict__ row_off,
                                               unsigned win_hi, int first, unsigned *__restrict__ row_cur,
                                               int *__restrict__ mark, uint8_t *__restrict__ inc) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  int *mk = reinterpret_cast<int *>(lds);                        // [cols] rank of the marking candidate
  uint8_t *neg = reinterpret_cast<uint8_t *>(lds) + (size_t)cols * 4;  // [cols] s < 0
  if (sc->done) return;
  const int a = blockIdx.x, lane = threadIdx.x;
  const float mean = mean_fft(sc, (int64_t)rows * cols);
  for (int r = lane; r < cols; r += 64) {
    mk[r] = first ? 0x7fffffff : mark[(int64_t)a * cols + r];
    neg[r] = __fsub_rn(px(img, a, r, stride, off), mean) < 0.0f;
  }
  __syncthreads();
  const unsigned t1 = row_off[a + 1];
  unsigned cur = first ? row_off[a] : row_cur[a];
  bool more = true;
  while (more && cur < t1) {
   // 64 candidates of this azimuth at a time: two dependent global loads per CHUNK, not per candidate
   const unsigned my_t = cur + lane;
   const unsigned my_j = my_t < t1 ? rank_by_row[my_t] : 0xffffffffu;
   const int my_r = my_t < t1 ? (int)((unsigned)(keys_sorted[my_j] & 0xffffffffull) - (unsigned)a * (unsigned)cols) : 0;
   // the row's list is sorted by rank: the candidates inside the window are a prefix of the chunk
   const unsigned cnt = (unsigned)__popcll(__ballot(my_t < t1 && my_j < win_hi));
   more = cnt == 64u;
   cur += cnt;
   for (unsigned i = 0; i < cnt; i++) {
    const unsigned j = __shfl(my_j, (int)i);
    const int r = __shfl(my_r, (int)i);
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
  }
  __syncthreads();
  for (int r = lane; r < cols; r += 64) mark[(int64_t)a * cols + r] = mk[r];
  if (lane == 0) row_cur[a] = cur;
}

// J* = number of candidates visited by `while (l < max_points && j < M)`: the first rank whose
// exclusive prefix count of new regions reaches max_points, or M.  inc[] holds 0/1 bytes in rank
// order; this looks at the window [win_lo, win_hi) (win_lo a multiple of 16) on top of the regions
// counted in earlier windows; every thread sums a contiguous 16-byte aligned chunk with 16-byte loads.
__global__ __launch_bounds__(1024) void cen_budget(const uint8_t *__restrict__ inc, Scal *sc, int max_points, unsigned win_lo,
                                                   unsigned win_hi) {
  __shared__ unsigned s[1024];
  if (sc->done) return;
  const unsigned m = sc->n_cand;
  const unsigned t = threadIdx.x;
  if (max_points <= 0) {  // while (0 < 0 ...) never runs
    if (t == 0) {
      sc->jstar = 0;
      sc->done = 1;
    }
    return;
  }
  const unsigned before = sc->regions;
  const unsigned w_hi = win_hi < m ? win_hi : m;
  const unsigned w_lo = win_lo < w_hi ? win_lo : w_hi;
  const unsigned len = w_hi - w_lo;
  const unsigned chunk = (((len + 1023) / 1024) + 15u) & ~15u;
  const unsigned lo = w_lo + (t * chunk < len ? t * chunk : len);
  const unsigned hi = lo + chunk < w_hi ? lo + chunk : w_hi;
  unsigned c = 0;
  unsigned j = lo;
  for (; j + 16 <= hi; j += 16) {
    const uint4 v = *reinterpret_cast<const uint4 *>(inc + j);
    c += __popc(v.x & 0x01010101u) + __popc(v.y & 0x01010101u) + __popc(v.z & 0x01010101u) + __popc(v.w & 0x01010101u);
  }
  for (; j < hi; j++) c += inc[j] & 1u;
  s[t] = c;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    unsigned v = t >= (unsigned)d ? s[t - d] : 0u;
    __syncthreads();
    s[t] += v;
    __syncthreads();
  }
  const unsigned total = before + s[1023];
  if (total >= (unsigned)max_points) {
    const unsigned excl = before + s[t] - c;  // new regions before this chunk
    if (excl < (unsigned)max_points && excl + c >= (unsigned)max_points) {
      // the candidate that opens region number max_points is the last one visited
      unsigned l = excl;
      for (unsigned jj = lo; jj < hi; jj++) {
        l += inc[jj] & 1u;
        if (l >= (unsigned)max_points) {
          sc->jstar = (long long)jj + 1;
          sc->done = 1;
          break;
        }
      }
    }
  } else if (t == 0) {
    sc->regions = total;
    if (w_hi >= m) {  // every candidate was visited
      sc->jstar = (long long)m;
      sc->done = 1;
    }
  }
}

// one wavefront per azimuth.  LDS: per pixel a flag byte (bit 0: marked by a visited candidate,
// bit 1: a marked pixel at the same range on the azimuth above or below) and the result slot of a run
// (indexed by the run's first pixel).  Runs are short, so the lane that owns a run's first pixel walks
// it; an ordered ballot compaction then restores ascending range order.
__global__ __launch_bounds__(64) void cen_extract(const float *__restrict__ h, const int *__restrict__ mark, int rows, int cols,
                                                  const Scal *sc, int min_range, int *__restrict__ row_out,
                                                  unsigned *__restrict__ row_n) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  int *res = reinterpret_cast<int *>(lds);                             // [cols]
  uint8_t *flag = reinterpret_cast<uint8_t *>(lds) + (size_t)cols * 4;  // [cols]
  const int a = blockIdx.x, lane = threadIdx.x;
  const int js = (int)(sc->jstar > 0x7fffffff ? 0x7fffffff : sc->jstar);
  const int *mr = mark + (int64_t)a * cols;
  const int *below = mark + (int64_t)((a - 1 + rows) % rows) * cols;
  const int *above = mark + (int64_t)((a + 1) % rows) * cols;
  const float *hr = h + (int64_t)a * cols;
  for (int r = lane; r < cols; r += 64) {
    flag[r] = (uint8_t)((mr[r] < js ? 1 : 0) | ((below[r] < js || above[r] < js) ? 2 : 0));
    res[r] = -1;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const int rmin = min_range < 0 ? 0 : min_range;
  for (int r = rmin + lane; r < cols; r += 64) {
    if (!(flag[r] & 1)) continue;
    if (r > rmin && (flag[r - 1] & 1)) continue;  // not the first pixel of its run (runs are cut at rmin)
    bool adj = false;
    int max_r = r;
    float mx = -INFINITY;
    int i = r;
    for (; i < cols && (flag[i] & 1); i++) {
      adj |= (flag[i] & 2) != 0;
      const float hv = hr[i];
      if (hv > mx) {  // first maximum
        mx = hv;
        max_r = i;
      }
    }
    // a run only counts once an unmarked pixel closes it: a run that reaches the end of the row does not
    if (i < cols && adj) res[r] = max_r;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  unsigned cnt = 0;
  for (int base = 0; base < cols; base += 64) {
    const int r = base + lane;
    const int v = r < cols ? res[r] : -1;
    const unsigned long long bal = __ballot(v >= 0);
    const unsigned pos = cnt + (unsigned)__popcll(bal & ((1ull << lane) - 1ull));
    if (v >= 0 && pos < ROW_CAP) row_out[(int64_t)a * ROW_CAP + pos] = v;
    cnt += (unsigned)__popcll(bal);
  }
  if (lane == 0) row_n[a] = cnt < ROW_CAP ? cnt : ROW_CAP;
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
      targets, xy, az, temp, row_cur;
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
                         &h->temp, &h->row_cur})
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
  hipLaunchKernelGGL(cen_stats, dim3(rows), dim3(256), 0, s, d_img, rows, cols, stride, off, sc);
  hipLaunchKernelGGL(cen_h, dim3(rows), dim3(256), 0, s, d_img, rows, cols, stride, off, sc, h->h.as<float>());
  hipLaunchKernelGGL(cen_candidates, dim3(rows), dim3(256), 0, s, h->h.as<float>(), rows, cols, sc,
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
    RSX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&cen_extract), hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 5));
    h->attr_set = true;
  }
  // rank windows [0, W), [W, 4W), [4W, 16W) ...: the budget usually runs out in the first one
  RSX_TRY(h->row_cur.reserve((size_t)rows * 4, s, false));
  {
    unsigned w = p.max_points > 0 ? 2u * (unsigned)p.max_points : 16384u;
    w = (w + 16383u) & ~16383u;
    unsigned lo = 0;
    int first = 1;
    while (true) {
      const unsigned hi = (m - lo <= w) ? m : lo + w;
      hipLaunchKernelGGL(cen_mark, dim3(rows), dim3(64), lds, s, d_img, rows, cols, stride, off, sc,
                         h->keys2.as<unsigned long long>(), h->rank2.as<unsigned>(), h->row_off.as<unsigned>(), hi, first,
                         h->row_cur.as<unsigned>(), h->mark.as<int>(), h->inc.as<uint8_t>());
      hipLaunchKernelGGL(cen_budget, dim3(1), dim3(1024), 0, s, h->inc.as<uint8_t>(), sc, p.max_points, lo, hi);
      first = 0;
      if (hi >= m) break;
      lo = hi;
      w *= 4;
    }
  }
  hipLaunchKernelGGL(cen_extract, dim3(rows), dim3(64), lds, s, h->h.as<float>(), h->mark.as<int>(), rows, cols, sc,
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
