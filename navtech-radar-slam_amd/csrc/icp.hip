// icp.hip -- point-to-point ICP loop verification on gfx950: the step right
// after a ScanContext loop candidate in the reference's PGO node
// (pgo/SC-A-LOAM/src/laserPosegraphOptimization.cpp:371-392: pcl::IterativeClosestPoint with
// MaxCorrespondenceDistance 150, MaximumIterations 100, TransformationEpsilon 1e-6,
// EuclideanFitnessEpsilon 1e-6, RANSACIterations 0; loop accepted when hasConverged() and
// getFitnessScore() <= 0.3).  PCL is neither vendored in the reference checkout nor installed here:
// this follows the published algorithm as restated in oracle/icp_ref.c (PARITY UNPINNED).
//
// Per iteration (all on the handle's stream, one 4-byte read-back for the convergence flag):
//   icp_nn        nearest target point of every current source point, brute force: a block owns 256
//                 source points x one slice of the target (staged through LDS), slices are combined with
//                 a 64-bit atomicMin on (distance bits << 32 | target index) -- squared distances are
//                 non-negative floats, so their bit patterns order like the values, and ties go to the
//                 lower index like a sequential scan.  Distances are computed with the oracle's exact
//                 float expression, so the correspondences are identical to the oracle's.
//   icp_moments   count, sum of matched source / target points, sum of squared distances   (fp64 sums)
//   icp_cov       covariance sum (dst - mean_dst)(src - mean_src)^T                          (fp64 sums)
//   icp_update    one thread: Umeyama rotation (Jacobi on H^T H), step, final = step * final,
//                 DefaultConvergenceCriteria
//   icp_apply     source <- step * source
// A submap of 51 keyframes against one scan is ~10^5 x 10^3 points: 10^8 distance evaluations per
// iteration, a few tens of microseconds.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <new>

#include "icp.h"
#include "rsx_common.h"

namespace {

constexpr int NN_TILE = 1024;  // target points per LDS tile
constexpr int NN_SLICE_TILES = 1;  // tiles per block: small slices keep all CUs busy for the ~10^3-point sources of this path

struct IcpState {
  double sums[8];     // count, src xyz, tgt xyz, sum d2
  double cov[9];
  float step[16];
  float final_t[16];
  double prev_mse;
  double fit_sum;
  unsigned long long fit_cnt;
  int iterations, converged, state, pad;
};

enum { ST_NOT = 0, ST_ITER = 1, ST_TRANSFORM = 2, ST_ABS_MSE = 3, ST_REL_MSE = 4, ST_NO_CORR = 5 };

__global__ __launch_bounds__(256) void icp_init(const char *__restrict__ src, int64_t ns, int64_t stride, const float *guess,
                                                IcpState *S, float *__restrict__ cur) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i == 0) {
    for (int k = 0; k < 16; k++) S->final_t[k] = guess ? guess[k] : ((k % 5 == 0) ? 1.0f : 0.0f);
    S->prev_mse = 1.7976931348623157e308;
    S->iterations = 0;
    S->converged = 0;
    S->state = ST_NOT;
  }
  if (i >= ns) return;
  const float *p = reinterpret_cast<const float *>(src + i * stride);
  float g[12];
  for (int k = 0; k < 12; k++) g[k] = guess ? guess[k] : ((k % 5 == 0) ? 1.0f : 0.0f);
  for (int r = 0; r < 3; r++)
    cur[3 * i + r] = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(g[4 * r], p[0]), __fmul_rn(g[4 * r + 1], p[1])), __fmul_rn(g[4 * r + 2], p[2])), g[4 * r + 3]);
}

__global__ __launch_bounds__(256) void icp_clear(unsigned long long *__restrict__ best, int64_t ns, IcpState *S) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < ns) best[i] = 0xffffffffffffffffull;
  if (i == 0) {
    for (int k = 0; k < 8; k++) S->sums[k] = 0.0;
    for (int k = 0; k < 9; k++) S->cov[k] = 0.0;
    S->fit_sum = 0.0;
    S->fit_cnt = 0;
  }
}

// grid (ceil(ns/256), slices): block = 256 source points x the target tiles [slice*8, slice*8+8)
__global__ __launch_bounds__(256) void icp_nn(const float *__restrict__ cur, int64_t ns, const char *__restrict__ tgt, int64_t nt,
                                              int64_t tstride, unsigned long long *__restrict__ best) {
  __shared__ float tx[NN_TILE], ty[NN_TILE], tz[NN_TILE];
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  float px = 0, py = 0, pz = 0;
  if (i < ns) {
    px = cur[3 * i];
    py = cur[3 * i + 1];
    pz = cur[3 * i + 2];
  }
  float bd = INFINITY;
  unsigned bi = 0xffffffffu;
  const int64_t t_lo = (int64_t)blockIdx.y * NN_SLICE_TILES * NN_TILE;
  const int64_t t_hi = (t_lo + (int64_t)NN_SLICE_TILES * NN_TILE < nt) ? t_lo + (int64_t)NN_SLICE_TILES * NN_TILE : nt;
  for (int64_t base = t_lo; base < t_hi; base += NN_TILE) {
    __syncthreads();
    for (int j = threadIdx.x; j < NN_TILE; j += 256) {
      const int64_t t = base + j;
      if (t < t_hi) {
        const float *q = reinterpret_cast<const float *>(tgt + t * tstride);
        tx[j] = q[0];
        ty[j] = q[1];
        tz[j] = q[2];
      }
    }
    __syncthreads();
    const int m = (int)((t_hi - base < NN_TILE) ? (t_hi - base) : NN_TILE);
    for (int j = 0; j < m; j++) {
      const float dx = __fsub_rn(px, tx[j]), dy = __fsub_rn(py, ty[j]), dz = __fsub_rn(pz, tz[j]);
      const float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
      if (d < bd) {  // ascending scan, strict <: the lower index wins ties
        bd = d;
        bi = (unsigned)(base + j);
      }
    }
  }
  if (i < ns && bi != 0xffffffffu)
    atomicMin(&best[i], ((unsigned long long)__float_as_uint(bd) << 32) | (unsigned long long)bi);
}

__device__ __forceinline__ double block_sum(double v, double *sh) {
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

// pass 1: count, sums of matched points, sum of squared distances (max_d2 = correspondence gate)
__global__ __launch_bounds__(256) void icp_moments(const float *__restrict__ cur, int64_t ns, const char *__restrict__ tgt,
                                                   int64_t tstride, const unsigned long long *__restrict__ best, float max_d2,
                                                   IcpState *S) {
  __shared__ double sh[4];
  double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < ns; i += (int64_t)gridDim.x * 256) {
    const unsigned long long b = best[i];
    if (b == 0xffffffffffffffffull) continue;
    const float d = __uint_as_float((unsigned)(b >> 32));
    if (!(d <= max_d2)) continue;
    const float *q = reinterpret_cast<const float *>(tgt + (int64_t)(unsigned)(b & 0xffffffffull) * tstride);
    v[0] += 1.0;
    v[1] += cur[3 * i];
    v[2] += cur[3 * i + 1];
    v[3] += cur[3 * i + 2];
    v[4] += q[0];
    v[5] += q[1];
    v[6] += q[2];
    v[7] += (double)d;
  }
  for (int k = 0; k < 8; k++) {
    const double t = block_sum(v[k], sh);
    if (threadIdx.x == 0 && t != 0.0) atomicAdd(&S->sums[k], t);
  }
}

__global__ __launch_bounds__(256) void icp_cov(const float *__restrict__ cur, int64_t ns, const char *__restrict__ tgt, int64_t tstride,
                                               const unsigned long long *__restrict__ best, float max_d2, IcpState *S) {
  __shared__ double sh[4];
  const double cnt = S->sums[0];
  if (cnt < 3.0) return;
  // means rounded to float like the oracle's float means
  float ms[3], md[3];
  for (int r = 0; r < 3; r++) {
    ms[r] = (float)(S->sums[1 + r] / cnt);
    md[r] = (float)(S->sums[4 + r] / cnt);
  }
  double v[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < ns; i += (int64_t)gridDim.x * 256) {
    const unsigned long long b = best[i];
    if (b == 0xffffffffffffffffull) continue;
    const float d = __uint_as_float((unsigned)(b >> 32));
    if (!(d <= max_d2)) continue;
    const float *q = reinterpret_cast<const float *>(tgt + (int64_t)(unsigned)(b & 0xffffffffull) * tstride);
    for (int a = 0; a < 3; a++)
      for (int c = 0; c < 3; c++) v[3 * a + c] += (double)__fmul_rn(__fsub_rn(q[a], md[a]), __fsub_rn(cur[3 * i + c], ms[c]));
  }
  for (int k = 0; k < 9; k++) {
    const double t = block_sum(v[k], sh);
    if (threadIdx.x == 0 && t != 0.0) atomicAdd(&S->cov[k], t);
  }
}

// R (row-major 3x3) from H = sum (dst - md)(src - ms)^T: Umeyama without scaling.  SVD through the
// symmetric eigen-decomposition of H^T H (cyclic Jacobi), third singular vectors by cross products so
// that det R = +1.
__device__ void rotation_from_covariance(const double *H, double *R) {
  double A[9], V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double s = 0;
      for (int k = 0; k < 3; k++) s += H[3 * k + i] * H[3 * k + j];
      A[3 * i + j] = s;
    }
  for (int sweep = 0; sweep < 60; sweep++) {
    const double off = fabs(A[1]) + fabs(A[2]) + fabs(A[5]);
    if (off < 1e-300) break;
    for (int p = 0; p < 2; p++)
      for (int q = p + 1; q < 3; q++) {
        const double apq = A[3 * p + q];
        if (fabs(apq) < 1e-300) continue;
        const double theta = (A[3 * q + q] - A[3 * p + p]) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 3; k++) {
          const double akp = A[3 * k + p], akq = A[3 * k + q];
          A[3 * k + p] = c * akp - s * akq;
          A[3 * k + q] = s * akp + c * akq;
        }
        for (int k = 0; k < 3; k++) {
          const double apk = A[3 * p + k], aqk = A[3 * q + k];
          A[3 * p + k] = c * apk - s * aqk;
          A[3 * q + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < 3; k++) {
          const double vkp = V[3 * k + p], vkq = V[3 * k + q];
          V[3 * k + p] = c * vkp - s * vkq;
          V[3 * k + q] = s * vkp + c * vkq;
        }
      }
  }
  int idx[3] = {0, 1, 2};
  const double ev[3] = {A[0], A[4], A[8]};
  for (int i = 0; i < 2; i++)
    for (int j = i + 1; j < 3; j++)
      if (ev[idx[j]] > ev[idx[i]]) {
        const int t = idx[i];
        idx[i] = idx[j];
        idx[j] = t;
      }
  double Vs[9], U[9], sig[3];
  for (int c = 0; c < 3; c++)
    for (int r = 0; r < 3; r++) Vs[3 * r + c] = V[3 * r + idx[c]];
  for (int c = 0; c < 3; c++) {
    double u[3];
    for (int r = 0; r < 3; r++) u[r] = H[3 * r + 0] * Vs[0 + c] + H[3 * r + 1] * Vs[3 + c] + H[3 * r + 2] * Vs[6 + c];
    sig[c] = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    for (int r = 0; r < 3; r++) U[3 * r + c] = u[r];
  }
  const double tol = 1e-12 * (sig[0] > 0 ? sig[0] : 1.0);
  for (int c = 0; c < 2; c++)
    if (sig[c] > tol)
      for (int r = 0; r < 3; r++) U[3 * r + c] /= sig[c];
  if (!(sig[0] > tol)) {
    for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1.0 : 0.0;
    return;
  }
  if (!(sig[1] > tol)) {
    const double a[3] = {U[0], U[3], U[6]};
    double b[3] = {fabs(a[0]) < 0.9 ? 1.0 : 0.0, fabs(a[0]) < 0.9 ? 0.0 : 1.0, 0.0};
    double d = a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
    for (int r = 0; r < 3; r++) b[r] -= d * a[r];
    d = sqrt(b[0] * b[0] + b[1] * b[1] + b[2] * b[2]);
    U[1] = b[0] / d;
    U[4] = b[1] / d;
    U[7] = b[2] / d;
  }
  const double u2[3] = {U[3] * U[7] - U[6] * U[4], U[6] * U[1] - U[0] * U[7], U[0] * U[4] - U[3] * U[1]};
  const double v2[3] = {Vs[3] * Vs[7] - Vs[6] * Vs[4], Vs[6] * Vs[1] - Vs[0] * Vs[7], Vs[0] * Vs[4] - Vs[3] * Vs[1]};
  for (int r = 0; r < 3; r++) {
    U[3 * r + 2] = u2[r];
    Vs[3 * r + 2] = v2[r];
  }
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) R[3 * i + j] = U[3 * i + 0] * Vs[3 * j + 0] + U[3 * i + 1] * Vs[3 * j + 1] + U[3 * i + 2] * Vs[3 * j + 2];
}

__global__ void icp_update(IcpState *S, int max_iterations, double teps, double feps) {
  if (threadIdx.x || blockIdx.x) return;
  const double cnt = S->sums[0];
  if (cnt < 3.0) {  // "Not enough correspondences found"
    S->converged = 0;
    S->state = ST_NO_CORR;
    S->iterations = -S->iterations - 1;  // negative: tells the host loop to stop
    return;
  }
  float ms[3], md[3];
  for (int r = 0; r < 3; r++) {
    ms[r] = (float)(S->sums[1 + r] / cnt);
    md[r] = (float)(S->sums[4 + r] / cnt);
  }
  double H[9], R[9];
  for (int k = 0; k < 9; k++) H[k] = (double)(float)S->cov[k] / cnt;
  rotation_from_covariance(H, R);
  float step[16];
  for (int k = 0; k < 16; k++) step[k] = (k % 5 == 0) ? 1.0f : 0.0f;
  for (int a = 0; a < 3; a++) {
    for (int b = 0; b < 3; b++) step[4 * a + b] = (float)R[3 * a + b];
    step[4 * a + 3] = (float)((double)md[a] - (R[3 * a] * ms[0] + R[3 * a + 1] * ms[1] + R[3 * a + 2] * ms[2]));
  }
  float fin[16];
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) {
      float s = 0.0f;
      for (int k = 0; k < 4; k++) s = __fadd_rn(s, __fmul_rn(step[4 * i + k], S->final_t[4 * k + j]));
      fin[4 * i + j] = s;
    }
  for (int k = 0; k < 16; k++) {
    S->step[k] = step[k];
    S->final_t[k] = fin[k];
  }
  const int it = ++S->iterations;
  const double mse = S->sums[7] / cnt;
  const double cos_angle = 0.5 * ((double)step[0] + (double)step[5] + (double)step[10] - 1.0);
  const double tsq = (double)step[3] * step[3] + (double)step[7] * step[7] + (double)step[11] * step[11];
  int conv = 0, st = ST_NOT;
  if (it >= max_iterations) {
    conv = 1;
    st = ST_ITER;
  } else if (cos_angle >= 1.0 - teps && tsq <= teps) {
    conv = 1;
    st = ST_TRANSFORM;
  } else if (fabs(mse - S->prev_mse) < 1e-12) {
    conv = 1;
    st = ST_ABS_MSE;
  } else if (fabs(mse - S->prev_mse) / S->prev_mse < feps) {
    conv = 1;
    st = ST_REL_MSE;
  } else {
    S->prev_mse = mse;
  }
  S->converged = conv;
  S->state = st;
}

__global__ __launch_bounds__(256) void icp_apply(float *__restrict__ cur, int64_t ns, const IcpState *S) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= ns) return;
  const float x = cur[3 * i], y = cur[3 * i + 1], z = cur[3 * i + 2];
  for (int r = 0; r < 3; r++)
    cur[3 * i + r] = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(S->step[4 * r], x), __fmul_rn(S->step[4 * r + 1], y)), __fmul_rn(S->step[4 * r + 2], z)), S->step[4 * r + 3]);
}

// cur = final * src (for getFitnessScore)
__global__ __launch_bounds__(256) void icp_final_cloud(const char *__restrict__ src, int64_t ns, int64_t stride, const IcpState *S,
                                                       float *__restrict__ cur) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= ns) return;
  const float *p = reinterpret_cast<const float *>(src + i * stride);
  for (int r = 0; r < 3; r++)
    cur[3 * i + r] = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(S->final_t[4 * r], p[0]), __fmul_rn(S->final_t[4 * r + 1], p[1])), __fmul_rn(S->final_t[4 * r + 2], p[2])), S->final_t[4 * r + 3]);
}

__global__ __launch_bounds__(256) void icp_fitness(const unsigned long long *__restrict__ best, int64_t ns, IcpState *S) {
  __shared__ double sh[4];
  double s = 0.0, c = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < ns; i += (int64_t)gridDim.x * 256) {
    const unsigned long long b = best[i];
    if (b == 0xffffffffffffffffull) continue;
    s += (double)__uint_as_float((unsigned)(b >> 32));
    c += 1.0;
  }
  const double ts = block_sum(s, sh);
  const double tc = block_sum(c, sh);
  if (threadIdx.x == 0 && tc > 0.0) {
    atomicAdd(&S->fit_sum, ts);
    atomicAdd(&S->fit_cnt, (unsigned long long)tc);
  }
}

}  // namespace

struct rsx_icp {
  int device = 0;
  std::mutex mu;
  hipStream_t stream = nullptr;
  rsx::DevBuf src, tgt, cur, best, state, guess;
};

using rsx::fail;

extern "C" {

int rsx_icp_default_params(rsx_icp_params *p) try {
  if (!p) return fail(RSX_ERR_BAD_ARG, "null params");
  p->max_corr_dist = 150.0;             // PGO.cpp:374
  p->transformation_epsilon = 1e-6;     // PGO.cpp:376
  p->euclidean_fitness_epsilon = 1e-6;  // PGO.cpp:377
  p->max_iterations = 100;              // PGO.cpp:375
  p->reserved = 0;
  return RSX_OK;
} RSX_CATCH_ALL

int rsx_icp_create(int device, rsx_icp **out) try {
  if (!out) return fail(RSX_ERR_BAD_ARG, "null out");
  *out = nullptr;
  int ndev = rsx_device_count();
  if (ndev <= 0) return fail(RSX_ERR_NO_DEVICE, "no HIP device visible (librsx has no CPU fallback)");
  if (device < 0 || device >= ndev) return fail(RSX_ERR_NO_DEVICE, "device %d out of range (%d visible)", device, ndev);
  rsx_icp *h = new (std::nothrow) rsx_icp();
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

int rsx_icp_destroy(rsx_icp *h) try {
  if (!h) return RSX_OK;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  for (rsx::DevBuf *b : {&h->src, &h->tgt, &h->cur, &h->best, &h->state, &h->guess}) b->release();
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return RSX_OK;
} RSX_CATCH_ALL

int rsx_icp_align(rsx_icp *h, const void *src, size_t ns, size_t src_stride, const void *tgt, size_t nt, size_t tgt_stride,
                  const rsx_icp_params *params, const float *guess, rsx_icp_result *out) try {
  if (!h || !out || (!src && ns) || (!tgt && nt)) return fail(RSX_ERR_BAD_ARG, "null arg");
  if (src_stride < 12 || (src_stride & 3) || tgt_stride < 12 || (tgt_stride & 3)) return fail(RSX_ERR_BAD_ARG, "strides must be >= 12 and multiples of 4");
  if (nt > 0xfffffffeull || ns > 0x7fffffffull) return fail(RSX_ERR_RANGE, "cloud too large");
  std::lock_guard<std::mutex> lk(h->mu);
  RSX_HIP(hipSetDevice(h->device));
  hipStream_t s = h->stream;
  RSX_TRY(h->src.reserve(ns * src_stride + 16, s, false));
  RSX_TRY(h->tgt.reserve(nt * tgt_stride + 16, s, false));
  if (ns) RSX_HIP(hipMemcpyAsync(h->src.p, src, ns * src_stride, hipMemcpyHostToDevice, s));
  if (nt) RSX_HIP(hipMemcpyAsync(h->tgt.p, tgt, nt * tgt_stride, hipMemcpyHostToDevice, s));
  return rsx::icp::align_device_locked(h, h->src.p, (int64_t)ns, (int64_t)src_stride, h->tgt.p, (int64_t)nt, (int64_t)tgt_stride, params, guess, out);
} RSX_CATCH_ALL

}  // extern "C"

namespace rsx {
namespace icp {

std::mutex &mutex_of(rsx_icp *h) { return h->mu; }
hipStream_t stream_of(rsx_icp *h) { return h->stream; }
int device_of(rsx_icp *h) { return h->device; }

// the alignment proper on clouds resident in device memory (the caller holds h->mu; whatever produced the clouds has
// completed or runs on h's stream)
int align_device_locked(rsx_icp *h, const void *d_src, int64_t n_s, int64_t src_stride, const void *d_tgt, int64_t n_t, int64_t tgt_stride,
                        const rsx_icp_params *params, const float *guess, rsx_icp_result *out) {
  rsx_icp_params p;
  rsx_icp_default_params(&p);
  if (params) p = *params;
  if (p.max_iterations < 1 || !(p.max_corr_dist > 0)) return fail(RSX_ERR_BAD_ARG, "bad ICP params");
  if (n_t > 0xfffffffell || n_s > 0x7fffffffll) return fail(RSX_ERR_RANGE, "cloud too large");
  RSX_HIP(hipSetDevice(h->device));
  hipStream_t s = h->stream;
  const size_t ns = (size_t)n_s;
  RSX_TRY(h->cur.reserve(ns * 12 + 16, s, false));
  RSX_TRY(h->best.reserve(ns * 8 + 16, s, false));
  RSX_TRY(h->state.reserve(sizeof(IcpState), s, false));
  RSX_TRY(h->guess.reserve(64, s, false));
  if (guess) RSX_HIP(hipMemcpyAsync(h->guess.p, guess, 64, hipMemcpyHostToDevice, s));
  IcpState *S = h->state.as<IcpState>();
  float *cur = h->cur.as<float>();
  unsigned long long *best = h->best.as<unsigned long long>();
  const char *dsrc = static_cast<const char *>(d_src), *dtgt = static_cast<const char *>(d_tgt);
  const unsigned nb = (unsigned)((n_s + 255) / 256 > 0 ? (n_s + 255) / 256 : 1);
  const unsigned nslices = (unsigned)((n_t + (int64_t)NN_SLICE_TILES * NN_TILE - 1) / ((int64_t)NN_SLICE_TILES * NN_TILE));
  const float max_d2 = (float)(p.max_corr_dist * p.max_corr_dist);
  hipLaunchKernelGGL(icp_init, dim3(nb), dim3(256), 0, s, dsrc, n_s, (int64_t)src_stride, guess ? h->guess.as<float>() : nullptr, S, cur);
  int iterations = 0, converged = 0, state = ST_NOT;
  struct {
    int iterations, converged, state, pad;
  } hs;
  while (true) {
    hipLaunchKernelGGL(icp_clear, dim3(nb), dim3(256), 0, s, best, n_s, S);
    if (nslices && n_s) hipLaunchKernelGGL(icp_nn, dim3(nb, nslices), dim3(256), 0, s, cur, n_s, dtgt, n_t, (int64_t)tgt_stride, best);
    const unsigned rb = nb < 256 ? nb : 256;
    hipLaunchKernelGGL(icp_moments, dim3(rb), dim3(256), 0, s, cur, n_s, dtgt, (int64_t)tgt_stride, best, max_d2, S);
    hipLaunchKernelGGL(icp_cov, dim3(rb), dim3(256), 0, s, cur, n_s, dtgt, (int64_t)tgt_stride, best, max_d2, S);
    hipLaunchKernelGGL(icp_update, dim3(1), dim3(64), 0, s, S, p.max_iterations, p.transformation_epsilon, p.euclidean_fitness_epsilon);
    hipLaunchKernelGGL(icp_apply, dim3(nb), dim3(256), 0, s, cur, n_s, S);
    RSX_HIP(hipGetLastError());
    RSX_HIP(hipMemcpyAsync(&hs, &S->iterations, sizeof(hs), hipMemcpyDeviceToHost, s));
    RSX_HIP(hipStreamSynchronize(s));
    if (hs.iterations < 0) {  // not enough correspondences: no step was taken
      iterations = -hs.iterations - 1;
      converged = 0;
      state = ST_NO_CORR;
      break;
    }
    iterations = hs.iterations;
    converged = hs.converged;
    state = hs.state;
    if (converged) break;
  }
  // getFitnessScore(): nearest-neighbour distances of final * source
  hipLaunchKernelGGL(icp_final_cloud, dim3(nb), dim3(256), 0, s, dsrc, n_s, (int64_t)src_stride, S, cur);
  hipLaunchKernelGGL(icp_clear, dim3(nb), dim3(256), 0, s, best, n_s, S);
  if (nslices && n_s) hipLaunchKernelGGL(icp_nn, dim3(nb, nslices), dim3(256), 0, s, cur, n_s, dtgt, n_t, (int64_t)tgt_stride, best);
  hipLaunchKernelGGL(icp_fitness, dim3(nb < 256 ? nb : 256), dim3(256), 0, s, best, n_s, S);
  RSX_HIP(hipGetLastError());
  IcpState hstate;
  RSX_HIP(hipMemcpyAsync(&hstate, S, sizeof(hstate), hipMemcpyDeviceToHost, s));
  RSX_HIP(hipStreamSynchronize(s));
  std::memcpy(out->transform, hstate.final_t, sizeof(out->transform));
  out->fitness = hstate.fit_cnt ? hstate.fit_sum / (double)hstate.fit_cnt : 1.7976931348623157e308;
  out->iterations = iterations;
  out->converged = converged;
  out->state = state;
  out->reserved = 0;
  return RSX_OK;
}

}  // namespace icp
}  // namespace rsx
