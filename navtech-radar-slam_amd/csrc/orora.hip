// orora.hip -- ORORA scan registration on gfx950: GNC-TLS rotation + A-COTE translation, one
// 256-thread workgroup per scan pair, everything on-chip (points and the interval endpoints live in
// LDS, TIMs and GNC weights in registers).
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

namespace {

constexpr int MAXK = 2048;   // correspondences per pair handled on-chip
constexpr int TPT = MAXK / 256;  // TIMs / points per thread
constexpr int MAXE = 2 * MAXK;   // interval endpoints per axis

struct Params {
  double c2, s_r, s_t, gnc_factor, cost_threshold;
  int max_iterations;
};

// ---- workgroup reductions (4 waves), deterministic order ----
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

struct Red {
  double *buf;  // 16 doubles of LDS
  __device__ double sum(double v) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) buf[threadIdx.x >> 6] = v;
    __syncthreads();
    return (buf[0] + buf[1]) + (buf[2] + buf[3]);
  }
  __device__ void sum2(double &a, double &b) {
    a = wave_sum(a);
    b = wave_sum(b);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
      buf[threadIdx.x >> 6] = a;
      buf[4 + (threadIdx.x >> 6)] = b;
    }
    __syncthreads();
    a = (buf[0] + buf[1]) + (buf[2] + buf[3]);
    b = (buf[4] + buf[5]) + (buf[6] + buf[7]);
  }
  __device__ double max(double v) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) buf[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmax(fmax(buf[0], buf[1]), fmax(buf[2], buf[3]));
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

// Scalar TLS estimate over K intervals x[i] +- beta[i] held TPT per thread (point i = tid + t*256).
// LDS: sx[MAXK], sb[MAXK] doubles, ev[MAXE] doubles, ei[MAXE] ints, part[6*256] doubles.
__device__ double scalar_tls_block(const double (&x)[TPT], const double (&beta)[TPT], int K, double *sx, double *sb,
                                   double *ev, int *ei, double *part, Red &red) {
  const int tid = threadIdx.x;
  int n2 = 512;
  while (n2 < 2 * K) n2 <<= 1;
  __syncthreads();
#pragma unroll
  for (int t = 0; t < TPT; t++) {
    const int i = tid + t * 256;
    if (i < K) {
      sx[i] = x[t];
      sb[i] = beta[t];
      ev[2 * i] = x[t] - beta[t];
      ei[2 * i] = i + 1;
      ev[2 * i + 1] = x[t] + beta[t];
      ei[2 * i + 1] = -i - 1;
    }
  }
  for (int e = 2 * K + tid; e < n2; e += 256) {
    ev[e] = INFINITY;
    ei[e] = 0x7fffffff;
  }
  __syncthreads();
  // bitonic sort, ascending by (value, id)
  for (int size = 2; size <= n2; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int p = tid; p < (n2 >> 1); p += 256) {
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
  // sweep as a scan: each thread owns a contiguous chunk of E endpoints
  const int E = n2 >> 8;
  const int e0 = tid * E;
  double l_sw = 0, l_swx = 0, l_swxx = 0;
  int l_card = 0;
  for (int e = e0; e < e0 + E; e++) {
    const int id = ei[e];
    if (id == 0x7fffffff) continue;
    const int idx = (id > 0 ? id : -id) - 1;
    const double eps = id > 0 ? 1.0 : -1.0;
    const double b = sb[idx], xv = sx[idx];
    const double w = 1.0 / (b * b);
    const double wx = w * xv;
    l_sw += eps * w;
    l_swx += eps * wx;
    l_swxx += eps * (wx * xv);
    l_card += id > 0 ? 1 : -1;
  }
  part[tid] = l_sw;
  part[256 + tid] = l_swx;
  part[512 + tid] = l_swxx;
  part[768 + tid] = (double)l_card;
  __syncthreads();
  if (tid < 4) {  // exclusive scan of the 256 chunk totals, one quantity per thread
    double run = 0.0;
    double *q = part + tid * 256;
    for (int i = 0; i < 256; i++) {
      const double v = q[i];
      q[i] = run;
      run += v;
    }
  }
  __syncthreads();
  double sw = part[tid], swx = part[256 + tid], swxx = part[512 + tid];
  int card = (int)part[768 + tid];
  double best_cost = INFINITY, best_x = 0.0;
  int best_pos = 0x7fffffff;
  for (int e = e0; e < e0 + E; e++) {
    const int id = ei[e];
    if (id == 0x7fffffff) continue;
    const int idx = (id > 0 ? id : -id) - 1;
    const double eps = id > 0 ? 1.0 : -1.0;
    const double b = sb[idx], xv = sx[idx];
    const double w = 1.0 / (b * b);
    const double wx = w * xv;
    sw += eps * w;
    swx += eps * wx;
    swxx += eps * (wx * xv);
    card += id > 0 ? 1 : -1;
    if (card <= 0) continue;
    const double x_hat = swx / sw;
    const double residual = swxx - 2.0 * swx * x_hat + sw * x_hat * x_hat;
    const double cost = residual + (double)(K - card);
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
    part[4 + (tid >> 6)] = best_x;
    part[8 + (tid >> 6)] = (double)best_pos;
  }
  __syncthreads();
  double bc = part[0], bx = part[4], bp = part[8];
  for (int w = 1; w < 4; w++)
    if (part[w] < bc || (part[w] == bc && part[8 + w] < bp)) {
      bc = part[w]; bx = part[4 + w]; bp = part[8 + w];
    }
  (void)red;
  return bx;
}

constexpr int LDS_PTS = MAXK * 16;                  // src+dst float2 (later sx, sb doubles)
constexpr int LDS_EV = MAXE * 8;
constexpr int LDS_EI = MAXE * 4;
constexpr int LDS_PART = 1024 * 8;
constexpr int LDS_RED = 16 * 8;
constexpr int LDS_TOTAL = LDS_PTS + LDS_EV + LDS_EI + LDS_PART + LDS_RED;  // 90240 B

__global__ __launch_bounds__(256) void orora_register_kernel(const float2 *__restrict__ src, const float2 *__restrict__ dst,
                                                             const int64_t *__restrict__ offsets, int n_pairs, Params p,
                                                             rsx_orora_result *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float2 *s_src = reinterpret_cast<float2 *>(smem);
  float2 *s_dst = s_src + MAXK;
  double *sx = reinterpret_cast<double *>(smem);  // aliases the points once they are dead
  double *sb = sx + MAXK;
  double *ev = reinterpret_cast<double *>(smem + LDS_PTS);
  int *ei = reinterpret_cast<int *>(smem + LDS_PTS + LDS_EV);
  double *part = reinterpret_cast<double *>(smem + LDS_PTS + LDS_EV + LDS_EI);
  Red red{reinterpret_cast<double *>(smem + LDS_PTS + LDS_EV + LDS_EI + LDS_PART)};

  const int pair = blockIdx.x;
  if (pair >= n_pairs) return;
  const int tid = threadIdx.x;
  const int64_t o = offsets[pair];
  const int64_t K64 = offsets[pair + 1] - o;
  if (K64 < 2 || K64 > MAXK) {
    if (tid == 0) {
      rsx_orora_result r;
      r.x = r.y = r.yaw = 0.0;
      r.iterations = r.rot_inliers = r.trans_inliers = 0;
      r.status = K64 < 2 ? 1 : 2;
      out[pair] = r;
    }
    return;
  }
  const int K = (int)K64;
  for (int i = tid; i < K; i += 256) {
    s_src[i] = src[o + i];
    s_dst[i] = dst[o + i];
  }
  __syncthreads();

  // ---- TIMs on the closed chain, TPT per thread ----
  double ax[TPT], ay[TPT], bx[TPT], by[TPT], w[TPT], r2[TPT];
#pragma unroll
  for (int t = 0; t < TPT; t++) {
    const int j = tid + t * 256;
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

  // ---- GNC-TLS rotation ----
  double mu = 1.0, prev_cost = INFINITY, cs = 1.0, sn = 0.0;
  int it = 0;
  for (it = 0; it < p.max_iterations; it++) {
    double C = 0.0, S = 0.0;
#pragma unroll
    for (int t = 0; t < TPT; t++) {
      C += w[t] * (ax[t] * bx[t] + ay[t] * by[t]);
      S += w[t] * (ax[t] * by[t] - ay[t] * bx[t]);
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
    double max_r2 = 0.0;
#pragma unroll
    for (int t = 0; t < TPT; t++) {
      const double ex = bx[t] - (cs * ax[t] - sn * ay[t]);
      const double ey = by[t] - (sn * ax[t] + cs * ay[t]);
      r2[t] = ex * ex + ey * ey;
      max_r2 = fmax(max_r2, (tid + t * 256 < K) ? r2[t] : 0.0);
    }
    if (it == 0) {
      max_r2 = red.max(max_r2);
      mu = 1.0 / (2.0 * max_r2 / p.c2 - 1.0);
      if (mu <= 0.0) {
        it = 1;
        break;
      }
    }
    const double th1 = (mu + 1.0) / mu * p.c2;
    const double th2 = mu / (mu + 1.0) * p.c2;
    double cost = 0.0;
#pragma unroll
    for (int t = 0; t < TPT; t++) {
      cost += w[t] * r2[t];
      if (tid + t * 256 < K) {
        if (r2[t] >= th1) w[t] = 0.0;
        else if (r2[t] <= th2) w[t] = 1.0;
        else w[t] = sqrt(p.c2 * mu * (mu + 1.0) / r2[t]) - mu;
      }
    }
    cost = red.sum(cost);
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
  for (int t = 0; t < TPT; t++) cnt += (tid + t * 256 < K && w[t] >= 0.5) ? 1.0 : 0.0;
  const int rot_inliers = (int)red.sum(cnt);

  // ---- A-COTE translation: residuals and anisotropic bounds, TPT per thread ----
  double vx[TPT], vy[TPT], betx[TPT], bety[TPT];
#pragma unroll
  for (int t = 0; t < TPT; t++) {
    const int i = tid + t * 256;
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
  const double tx = scalar_tls_block(vx, betx, K, sx, sb, ev, ei, part, red);
  const double ty = scalar_tls_block(vy, bety, K, sx, sb, ev, ei, part, red);
  cnt = 0.0;
#pragma unroll
  for (int t = 0; t < TPT; t++)
    cnt += (tid + t * 256 < K && fabs(vx[t] - tx) <= betx[t] && fabs(vy[t] - ty) <= bety[t]) ? 1.0 : 0.0;
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
    out[pair] = r;
  }
}

}  // namespace

struct rsx_orora {
  int device = 0;
  std::mutex mu;
  hipStream_t stream = nullptr;
  rsx::DevBuf src, dst, off, res;
  bool attr_set = false;
};

using rsx::fail;

extern "C" {

int rsx_orora_default_params(rsx_orora_params *p) {
  if (!p) return fail(RSX_ERR_BAD_ARG, "null params");
  p->tim_noise_bound = 2.0 * 0.75;
  p->noise_bound_radial = 0.3536;
  p->noise_bound_tangential = 1.8 * M_PI / 180.0;
  p->gnc_factor = 1.4;
  p->cost_threshold = 1e-6;
  p->max_iterations = 100;
  p->reserved = 0;
  return RSX_OK;
}

int rsx_orora_max_correspondences(void) { return MAXK; }

int rsx_orora_create(int device, rsx_orora **out) {
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
}

int rsx_orora_destroy(rsx_orora *h) {
  if (!h) return RSX_OK;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  h->src.release();
  h->dst.release();
  h->off.release();
  h->res.release();
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return RSX_OK;
}

int rsx_orora_register_batch_device(rsx_orora *h, const float *d_src_xy, const float *d_dst_xy, const int64_t *d_offsets,
                                    int32_t n_pairs, const rsx_orora_params *params, rsx_orora_result *d_out, void *stream) {
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
  hipStream_t s = stream ? static_cast<hipStream_t>(stream) : h->stream;
  hipLaunchKernelGGL(orora_register_kernel, dim3(n_pairs), dim3(256), LDS_TOTAL, s,
                     reinterpret_cast<const float2 *>(d_src_xy), reinterpret_cast<const float2 *>(d_dst_xy), d_offsets,
                     n_pairs, kp, d_out);
  RSX_HIP(hipGetLastError());
  return RSX_OK;
}

int rsx_orora_register_batch(rsx_orora *h, const float *src_xy, const float *dst_xy, const int64_t *offsets, int32_t n_pairs,
                             const rsx_orora_params *params, rsx_orora_result *out) {
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
}

}  // extern "C"
