// icp.hip -- point-to-point ICP loop verification on gfx950: the step right
// after a ScanContext loop candidate in the reference's PGO node
// (pgo/SC-A-LOAM/src/laserPosegraphOptimization.cpp:371-392: pcl::IterativeClosestPoint with
// MaxCorrespondenceDistance 150, MaximumIterations 100, TransformationEpsilon 1e-6,
// EuclideanFitnessEpsilon 1e-6, RANSACIterations 0; loop accepted when hasConverged() and
// getFitnessScore() <= 0.3).  PCL is neither vendored in the reference checkout nor installed here:
// this follows the published algorithm as restated in oracle/icp_ref.c (PARITY UNPINNED).
//
// Round 5: ONE persistent launch per alignment (rounds 1-4: six launches and a host round trip per iteration, ~150 launches
// for the 24 iterations of a loop verification).  Up to 256 workgroups of 1024 threads that meet at ONE grid barrier per
// iteration (rsx_grid_dev.h); the sizes of the two clouds may come from device memory (the VoxelGrid launch before it).
//   nearest neighbours   the workgroups form a (source block) x (target slice) grid; a workgroup keeps its slice of the
//                 target in LDS for the whole alignment, its 1024 threads are 32 groups of 4 source points x 32 sub-slices.  Distances
//                 with the oracle's exact float expression; slices are combined with a 64-bit min of
//                 (distance bits << 32 | target index) -- squared distances are non-negative floats, their bit patterns order
//                 like the values, and ties go to the lower index like a sequential scan: the correspondences are the
//                 oracle's.  The current source points are step * (previous points), recomputed by whoever needs them;
//                 the owner of a block also stores them (double-buffered) for the moments.
//   barrier
//   moments       EVERY workgroup sums count, matched source / target points, squared distances, then the covariance about
//                 the float means, over ALL correspondences, in fp64, in a fixed order: thread t adds the
//                 correspondences t, t + 1024, ... in turn, the 1024 partial sums are added 16 to a lane, the 64 lanes as a balanced tree.  Identical
//                 in every workgroup, identical on every run, and restated by oracle/icp_ref.c (sum order "tree") --
//                 no second and third barrier for means and covariance, no atomics on doubles.
//   update        Umeyama rotation (Jacobi on H^T H), step, final = step * final, DefaultConvergenceCriteria -- every
//                 workgroup for itself.
// After the last iteration: final * source, nearest neighbours once more, barrier, getFitnessScore(); workgroup 0 writes
// the result.  Nothing returns to the host before that.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>

#include "icp.h"
#include "rsx_common.h"
#include "rsx_grid_dev.h"
#include "rsx_persistent.h"

namespace {

constexpr int PI_NT = 1024;   // threads per workgroup
constexpr int PI_SRC = 128;   // source points per block
constexpr int PI_SP = 4;      // ... four per thread: a thread reads a target point ONCE (one 16-byte LDS read) for four distances (eight: spills and twice the LDS atomics, 795 us against 670),
                              // two at a time with packed fp32 arithmetic.  (First build: one source point per thread and three
                              // 4-byte LDS reads per distance -- the LDS pipe, not the VALU, set the pace: 45 us per iteration.)
constexpr int PI_GROUPS = PI_SRC / PI_SP;  // 32 threads cover a block's source points ...
constexpr int PI_SUB = PI_NT / PI_GROUPS;  // ... and 32 such groups split the workgroup's target slice
constexpr int PI_MOM = PI_NT;  // threads that add the moments: thread t the correspondences t, t + 1024, ...
constexpr size_t PI_DYN_LDS = (size_t)9 * PI_NT * 8;  // tree_sum's partial sums
constexpr int PI_TILE = 2048; // target points per LDS tile
constexpr int PI_MAX_G = 256;

struct IcpState {
  float final_t[16];
  double fit_sum;
  unsigned long long fit_cnt;
  int iterations, converged, state, pad;
  long long ns, nt;  // the sizes the alignment ran on
#ifdef RSX_ICP_TIMING
  unsigned long long tm[8];  // workgroup 0: 10 ns ticks in nn / barrier / sums / covariance / update / fitness
#endif
};

enum { ST_NOT = 0, ST_ITER = 1, ST_TRANSFORM = 2, ST_ABS_MSE = 3, ST_REL_MSE = 4, ST_NO_CORR = 5, ST_GAVE_UP = 99 };

struct IcpArgs {
  const char *src, *tgt;  // float x, y, z at byte offsets 0, 4, 8 of each stride
  long long src_stride, tgt_stride;
  const long long *ns_ptr, *nt_ptr;  // sizes in device memory (the launch before wrote them) ...
  long long ns_imm, nt_imm;          // ... or immediate (pointer null)
  long long ns_cap;                  // points the buffers below hold per copy
  const float *guess;                // optional row-major 4 x 4, device
  float4 *cur;                       // [2][ns_cap]: the current source points of the even / odd iterations
  unsigned long long *best;          // [3][ns_cap]: distance bits << 32 | target index
  IcpState *S;
  unsigned *bar;
  float max_d2;
  int max_iterations;
  double teps, feps;
};

// R (row-major 3x3) from H = sum (dst - md)(src - ms)^T: Umeyama without scaling.  SVD through the
// symmetric eigen-decomposition of H^T H (cyclic Jacobi), third singular vectors by cross products so
// that det R = +1.  The same operations in the same order as oracle/icp_ref.c; every index is a compile-time constant
// (the matrices stay in registers: with run-time indices they went to scratch memory, then to LDS -- 100 cycles an element).
__device__ __forceinline__ double sel3(double a, double b, double c, int i) { return i == 0 ? a : (i == 1 ? b : c); }

template <int P, int Q>
__device__ __forceinline__ void jacobi_rotate(double (&A)[9], double (&V)[9]) {
  const double apq = A[3 * P + Q];
  if (fabs(apq) < 1e-300) return;
  const double theta = (A[3 * Q + Q] - A[3 * P + P]) / (2.0 * apq);
  const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
  const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const double akp = A[3 * k + P], akq = A[3 * k + Q];
    A[3 * k + P] = c * akp - s * akq;
    A[3 * k + Q] = s * akp + c * akq;
  }
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const double apk = A[3 * P + k], aqk = A[3 * Q + k];
    A[3 * P + k] = c * apk - s * aqk;
    A[3 * Q + k] = s * apk + c * aqk;
  }
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const double vkp = V[3 * k + P], vkq = V[3 * k + Q];
    V[3 * k + P] = c * vkp - s * vkq;
    V[3 * k + Q] = s * vkp + c * vkq;
  }
}

__device__ __forceinline__ void rotation_from_covariance(const double (&H)[9], double (&R)[9]) {
  double A[9], V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      double s = 0;
#pragma unroll
      for (int k = 0; k < 3; k++) s += H[3 * k + i] * H[3 * k + j];
      A[3 * i + j] = s;
    }
  for (int sweep = 0; sweep < 60; sweep++) {
    const double off = fabs(A[1]) + fabs(A[2]) + fabs(A[5]);
    if (off <= 1e-22 * (fabs(A[0]) + fabs(A[4]) + fabs(A[8]))) break;  // (below 2^-53 of the diagonal a rotation changes nothing)
    jacobi_rotate<0, 1>(A, V);
    jacobi_rotate<0, 2>(A, V);
    jacobi_rotate<1, 2>(A, V);
  }
  // columns by descending eigenvalue (the oracle's three compare-and-swaps on an index array)
  int i0 = 0, i1 = 1, i2 = 2;
  const double e0 = A[0], e1 = A[4], e2 = A[8];
  if (sel3(e0, e1, e2, i1) > sel3(e0, e1, e2, i0)) {
    const int t = i0;
    i0 = i1;
    i1 = t;
  }
  if (sel3(e0, e1, e2, i2) > sel3(e0, e1, e2, i0)) {
    const int t = i0;
    i0 = i2;
    i2 = t;
  }
  if (sel3(e0, e1, e2, i2) > sel3(e0, e1, e2, i1)) {
    const int t = i1;
    i1 = i2;
    i2 = t;
  }
  double Vs[9], U[9], sig[3];
#pragma unroll
  for (int r = 0; r < 3; r++) {
    Vs[3 * r + 0] = sel3(V[3 * r], V[3 * r + 1], V[3 * r + 2], i0);
    Vs[3 * r + 1] = sel3(V[3 * r], V[3 * r + 1], V[3 * r + 2], i1);
    Vs[3 * r + 2] = sel3(V[3 * r], V[3 * r + 1], V[3 * r + 2], i2);
  }
#pragma unroll
  for (int c = 0; c < 3; c++) {
    double u[3];
#pragma unroll
    for (int r = 0; r < 3; r++) u[r] = H[3 * r + 0] * Vs[0 + c] + H[3 * r + 1] * Vs[3 + c] + H[3 * r + 2] * Vs[6 + c];
    sig[c] = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
#pragma unroll
    for (int r = 0; r < 3; r++) U[3 * r + c] = u[r];
  }
  const double tol = 1e-12 * (sig[0] > 0 ? sig[0] : 1.0);
#pragma unroll
  for (int c = 0; c < 2; c++)
    if (sig[c] > tol)
#pragma unroll
      for (int r = 0; r < 3; r++) U[3 * r + c] /= sig[c];
  if (!(sig[0] > tol)) {
#pragma unroll
    for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1.0 : 0.0;
    return;
  }
  if (!(sig[1] > tol)) {
    const double a[3] = {U[0], U[3], U[6]};
    double b[3] = {fabs(a[0]) < 0.9 ? 1.0 : 0.0, fabs(a[0]) < 0.9 ? 0.0 : 1.0, 0.0};
    double d = a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
#pragma unroll
    for (int r = 0; r < 3; r++) b[r] -= d * a[r];
    d = sqrt(b[0] * b[0] + b[1] * b[1] + b[2] * b[2]);
    U[1] = b[0] / d;
    U[4] = b[1] / d;
    U[7] = b[2] / d;
  }
  const double u2[3] = {U[3] * U[7] - U[6] * U[4], U[6] * U[1] - U[0] * U[7], U[0] * U[4] - U[3] * U[1]};
  const double v2[3] = {Vs[3] * Vs[7] - Vs[6] * Vs[4], Vs[6] * Vs[1] - Vs[0] * Vs[7], Vs[0] * Vs[4] - Vs[3] * Vs[1]};
#pragma unroll
  for (int r = 0; r < 3; r++) {
    U[3 * r + 2] = u2[r];
    Vs[3 * r + 2] = v2[r];
  }
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) R[3 * i + j] = U[3 * i + 0] * Vs[3 * j + 0] + U[3 * i + 1] * Vs[3 * j + 1] + U[3 * i + 2] * Vs[3 * j + 2];
}

template <int CTRL>
__device__ __forceinline__ double dpp_f64(double x) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(x);
  const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)u, CTRL, 0xf, 0xf, true);
  const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTRL, 0xf, 0xf, true);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ double readlane_f64(double x, int l) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(x);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, l), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), l);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// the four levels of the tree inside a row of 16 lanes: lanes that already hold the same partial sum add the partner
// group's (quad_perm, then the mirrors, which pair every lane with one of the other group) -- DPP, no LDS traffic (the
// __shfl_xor form was 180 ds_bpermute per wavefront and sum: the sixteen wavefronts of a workgroup queued on the LDS pipe)
__device__ __forceinline__ double row_tree(double x) {
  x += dpp_f64<0xB1>(x);   // quad_perm [1,0,3,2]
  x += dpp_f64<0x4E>(x);   // quad_perm [2,3,0,1]
  x += dpp_f64<0x141>(x);  // row_half_mirror
  x += dpp_f64<0x140>(x);  // row_mirror
  return x;
}
// N sums over the 1024 threads of the workgroup in a fixed order ("tree", restated by oracle/icp_ref.c): thread t holds
// partial sum t; wavefront c adds the sixteen partial sums l, l + 64, l + 128, ... of sum c in its lane l one after the other,
// then its 64 lanes as a balanced tree, neighbours first.  The partial sums go through LDS (part: N x 1024 doubles), so ONE
// wavefront per sum runs ONE tree.  Called by all threads (two workgroup barriers); every thread ends with the N sums.
// (Earlier builds: every wavefront reduced every sum -- with __shfl_xor 180 ds_bpermute per wavefront and sum, sixteen
// wavefronts queued on the LDS pipe; with DPP 4 us of VALU per call on all sixteen.)
template <int N>
__device__ __forceinline__ void tree_sum(double (&v)[N], double *part, double *res) {
  static_assert(N <= PI_NT / 64, "a wavefront per sum");
  const int t = threadIdx.x, w = t >> 6, lane = t & 63;
#pragma unroll
  for (int c = 0; c < N; c++) part[c * PI_NT + t] = v[c];
  __syncthreads();
  if (w < N) {
    double a = part[w * PI_NT + lane];
#pragma unroll
    for (int j = 1; j < PI_NT / 64; j++) a += part[w * PI_NT + lane + 64 * j];
    a = row_tree(a);
    a = (readlane_f64(a, 0) + readlane_f64(a, 16)) + (readlane_f64(a, 32) + readlane_f64(a, 48));
    if (lane == 0) res[w] = a;
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < N; c++) v[c] = res[c];
}

__device__ __forceinline__ void apply16(const float *m, float x, float y, float z, float &ox, float &oy, float &oz) {
  ox = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(m[0], x), __fmul_rn(m[1], y)), __fmul_rn(m[2], z)), m[3]);
  oy = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(m[4], x), __fmul_rn(m[5], y)), __fmul_rn(m[6], z)), m[7]);
  oz = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(m[8], x), __fmul_rn(m[9], y)), __fmul_rn(m[10], z)), m[11]);
}
__device__ __forceinline__ float4 ld_point4(const float4 *p) {
  const unsigned long long *q = reinterpret_cast<const unsigned long long *>(p);
  const unsigned long long a = rsx::grid::ld(q), b = rsx::grid::ld(q + 1);
  return make_float4(__uint_as_float((unsigned)a), __uint_as_float((unsigned)(a >> 32)), __uint_as_float((unsigned)b), 0.0f);
}
__device__ __forceinline__ void st_point4(float4 *p, float x, float y, float z) {
  unsigned long long *q = reinterpret_cast<unsigned long long *>(p);
  rsx::grid::st(q, (unsigned long long)__float_as_uint(x) | ((unsigned long long)__float_as_uint(y) << 32));
  rsx::grid::st(q + 1, (unsigned long long)__float_as_uint(z));
}

__global__ __launch_bounds__(PI_NT) void icp_persistent_kernel(IcpArgs A) {
  using namespace rsx;
  __shared__ float4 tq[PI_TILE];
  __shared__ unsigned long long s_best[PI_SRC];
  extern __shared__ double s_part[];  // 9 x 1024 partial sums (tree_sum)
  __shared__ double s_sum[16];
  __shared__ float s_step[16], s_final[16];
  __shared__ double s_prev_mse;
  __shared__ int s_conv, s_state, s_iter;
  const int t = threadIdx.x;
  const unsigned G = gridDim.x, b = blockIdx.x;
  grid::Member m{A.bar, b, G, 0u, false};
  // (a barrier that gives up -- rsx_grid_dev.h -- ends the alignment with state ST_GAVE_UP: the host reports it)
  auto gave_up = [&]() {
    if (b == 0 && threadIdx.x == 0) {
      A.S->state = ST_GAVE_UP;
      A.S->converged = 0;
      A.S->iterations = 0;
      A.S->fit_cnt = 0;
    }
    grid::exit(m);
  };
  const long long ns = A.ns_ptr ? *A.ns_ptr : A.ns_imm, nt = A.nt_ptr ? *A.nt_ptr : A.nt_imm;
  if (ns < 0 || nt < 0) {  // the launch that was to leave the sizes gave up (uniform: no barrier has been entered)
    gave_up();
    return;
  }
  // the workgroups as (source block) x (target slice)
  const long long nblk = (ns + PI_SRC - 1) / PI_SRC;
  const unsigned Gs = (unsigned)(nblk < 1 ? 1 : (nblk < (long long)G ? nblk : (long long)G)), Gt = G / Gs;
  const bool working = b < Gs * Gt;
  const unsigned gs = b % Gs, gt = b / Gs;
  const long long tper = (nt + Gt - 1) / Gt;
  const long long t_lo = working ? ((long long)gt * tper < nt ? (long long)gt * tper : nt) : 0;
  const long long t_hi = working ? (t_lo + tper < nt ? t_lo + tper : nt) : 0;
  const bool resident = t_hi - t_lo <= PI_TILE;  // the slice stays in LDS for the whole alignment
  const bool owner = working && gt == 0;         // stores the current points of its blocks, clears their records
  const int sg = t & (PI_GROUPS - 1), sub = t / PI_GROUPS;  // source points 4 sg .. 4 sg + 3 of a block, target sub-slice
  unsigned long long *best0 = A.best, *best1 = A.best + A.ns_cap, *best2 = A.best + 2 * A.ns_cap;
  auto best_of = [&](int k) { return k % 3 == 0 ? best0 : (k % 3 == 1 ? best1 : best2); };
  auto load_tile = [&](long long base, int cnt) {
    for (int j = t; j < cnt; j += PI_NT) {
      const float *q = reinterpret_cast<const float *>(A.tgt + (base + j) * A.tgt_stride);
      tq[j] = make_float4(q[0], q[1], q[2], 0.0f);
    }
  };
  if (t == 0) {
    for (int k = 0; k < 16; k++) s_final[k] = A.guess ? A.guess[k] : ((k % 5 == 0) ? 1.0f : 0.0f);
    for (int k = 0; k < 16; k++) s_step[k] = (k % 5 == 0) ? 1.0f : 0.0f;
    s_prev_mse = 1.7976931348623157e308;
    s_conv = 0;
    s_state = ST_NOT;
    s_iter = 0;
  }
  if (resident && working) load_tile(t_lo, (int)(t_hi - t_lo));
  if (owner && t < PI_SRC)
    for (long long blk = gs; blk < nblk; blk += Gs) {
      const long long i = blk * PI_SRC + t;
      if (i < ns) grid::st(best0 + i, ~0ull);
    }
  if (!grid::sync(m)) {
    gave_up();
    return;
  }
  // the nearest neighbours of this workgroup's source blocks in its target slice -> rec (min over the slices).  The points:
  // mode 0 = guess * source, 1 = step * (points of the iteration before), 2 = final * source; the owner stores them / clears `clr`
#ifdef RSX_ICP_TIMING
  unsigned long long tm[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = wall_clock64();
#define ICP_MARK(i)                         \
  {                                         \
    const unsigned long long n_ = wall_clock64(); \
    tm[i] += n_ - tlast;                    \
    tlast = n_;                             \
  }
#else
#define ICP_MARK(i)
#endif
  typedef float f2 __attribute__((ext_vector_type(2)));
  // a workgroup with ONE source block (the usual case: a scan of ~10^3 points is ~10 blocks, the grid 256 workgroups)
  // keeps its four points per thread in registers from iteration to iteration; otherwise they are read back
  const bool one_block = nblk <= (long long)Gs;
  float hx[PI_SP] = {}, hy[PI_SP] = {}, hz[PI_SP] = {};
  auto nn_pass = [&](int mode, const float4 *prev, float4 *store, unsigned long long *rec, unsigned long long *clr) {
    if (!working) return;
    for (long long blk = gs; blk < nblk; blk += Gs) {
      float px[PI_SP], py[PI_SP], pz[PI_SP];
#pragma unroll
      for (int e = 0; e < PI_SP; e++) {
        const long long i = blk * PI_SRC + sg * PI_SP + e;
        px[e] = py[e] = pz[e] = 0.0f;
        if (i < ns) {
          if (mode == 1 && one_block) {
            apply16(s_step, hx[e], hy[e], hz[e], px[e], py[e], pz[e]);
          } else if (mode == 1) {
            const float4 q = ld_point4(prev + i);
            apply16(s_step, q.x, q.y, q.z, px[e], py[e], pz[e]);
          } else {
            const float *q = reinterpret_cast<const float *>(A.src + i * A.src_stride);
            apply16(s_final, q[0], q[1], q[2], px[e], py[e], pz[e]);
          }
          if (owner && sub == 0) {
            if (store) st_point4(store + i, px[e], py[e], pz[e]);
            if (clr) grid::st(clr + i, ~0ull);
          }
        }
        if (mode != 2) {
          hx[e] = px[e];
          hy[e] = py[e];
          hz[e] = pz[e];
        }
      }
      ICP_MARK(6)
      if (t < PI_SRC) s_best[t] = ~0ull;
      float bd[PI_SP];
      unsigned bi[PI_SP];
#pragma unroll
      for (int e = 0; e < PI_SP; e++) {
        bd[e] = INFINITY;
        bi[e] = 0xffffffffu;
      }
      f2 xx[PI_SP / 2], yy[PI_SP / 2], zz[PI_SP / 2];
#pragma unroll
      for (int h2 = 0; h2 < PI_SP / 2; h2++) {
        xx[h2] = f2{px[2 * h2], px[2 * h2 + 1]};
        yy[h2] = f2{py[2 * h2], py[2 * h2 + 1]};
        zz[h2] = f2{pz[2 * h2], pz[2 * h2 + 1]};
      }
      for (long long base = t_lo; base < t_hi; base += PI_TILE) {
        const int cnt = (int)(t_hi - base < PI_TILE ? t_hi - base : PI_TILE);
        if (!resident) {
          __syncthreads();
          load_tile(base, cnt);
        }
        __syncthreads();
        const int len = (cnt + PI_SUB - 1) / PI_SUB, j0 = sub * len, j1 = j0 + len < cnt ? j0 + len : cnt;
        for (int j = j0; j < j1; j++) {
          const float4 T = tq[j];
          const unsigned idx = (unsigned)(base + j);
          // (x - tx)^2 + (y - ty)^2 + (z - tz)^2, every operation rounded on its own (no contraction), two source points per
          // packed instruction -- the oracle's float expression
          const f2 tx = {T.x, T.x}, ty = {T.y, T.y}, tz = {T.z, T.z};
#pragma unroll
          for (int h2 = 0; h2 < PI_SP / 2; h2++) {
            const f2 dx = xx[h2] - tx, dy = yy[h2] - ty, dz = zz[h2] - tz;
            const f2 d = (dx * dx + dy * dy) + dz * dz;
            if (d.x < bd[2 * h2]) {  // ascending scan, strict <: the lower index wins ties
              bd[2 * h2] = d.x;
              bi[2 * h2] = idx;
            }
            if (d.y < bd[2 * h2 + 1]) {
              bd[2 * h2 + 1] = d.y;
              bi[2 * h2 + 1] = idx;
            }
          }
        }
      }
      ICP_MARK(7)
#pragma unroll
      for (int e = 0; e < PI_SP; e++)
        if (blk * PI_SRC + sg * PI_SP + e < ns && bi[e] != 0xffffffffu)
          atomicMin(&s_best[sg * PI_SP + e], ((unsigned long long)__float_as_uint(bd[e]) << 32) | (unsigned long long)bi[e]);
      __syncthreads();
      if (t < PI_SRC && blk * PI_SRC + t < ns && s_best[t] != ~0ull)
        __hip_atomic_fetch_min(rec + blk * PI_SRC + t, s_best[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
    }
  };
  int k = 0;
  while (true) {
    unsigned long long *rec = best_of(k);
    const float4 *curk = A.cur + (size_t)(k & 1) * A.ns_cap;
    nn_pass(k == 0 ? 0 : 1, A.cur + (size_t)((k + 1) & 1) * A.ns_cap, A.cur + (size_t)(k & 1) * A.ns_cap, rec, best_of(k + 1));
    ICP_MARK(0)
    if (!grid::sync(m)) {
      gave_up();
      return;
    }
    ICP_MARK(1)
    // ---- moments over all correspondences, by every workgroup ----
    // (a source cloud of up to KEEP * PI_MOM points -- a keyframe scan -- keeps its correspondences in registers between the two passes)
    constexpr int KEEP = PI_MOM == 1024 ? 2 : 5;
    const bool keep = ns <= (long long)KEEP * PI_MOM;
    const bool mom = t < PI_MOM;
    float kq[KEEP][3], kc[KEEP][3];
    bool kv[KEEP] = {};
    auto corr = [&](long long i, float (&q3)[3], float (&c3)[3], float &d) -> bool {
      const unsigned long long r = grid::ld(rec + i);
      if (r == ~0ull) return false;
      d = __uint_as_float((unsigned)(r >> 32));
      if (!(d <= A.max_d2)) return false;
      const float *q = reinterpret_cast<const float *>(A.tgt + (long long)(unsigned)(r & 0xffffffffull) * A.tgt_stride);
      const float4 c = ld_point4(curk + i);
      q3[0] = q[0];
      q3[1] = q[1];
      q3[2] = q[2];
      c3[0] = c.x;
      c3[1] = c.y;
      c3[2] = c.z;
      return true;
    };
    double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (mom && keep) {
      // the loads in two batches (records and current points; then the matched target points): every load of a batch is
      // issued before the first is waited for -- one load, one wait per correspondence was 13 us an iteration
      unsigned long long rr[KEEP];
      float4 cc[KEEP];
      float dd[KEEP];
#pragma unroll
      for (int e = 0; e < KEEP; e++) {
        const long long i = t + (long long)e * PI_MOM, ii = i < ns ? i : 0;  // (past the end: a valid address, masked below)
        rr[e] = ns > 0 ? grid::ld(rec + ii) : ~0ull;
        cc[e] = ns > 0 ? ld_point4(curk + ii) : make_float4(0, 0, 0, 0);
      }
#pragma unroll
      for (int e = 0; e < KEEP; e++) {
        const long long i = t + (long long)e * PI_MOM;
        dd[e] = __uint_as_float((unsigned)(rr[e] >> 32));
        kv[e] = i < ns && rr[e] != ~0ull && dd[e] <= A.max_d2;
        const float *q = reinterpret_cast<const float *>(A.tgt + (kv[e] ? (long long)(unsigned)(rr[e] & 0xffffffffull) : 0ll) * A.tgt_stride);
        const bool have = nt > 0;
        kq[e][0] = have ? q[0] : 0.0f;
        kq[e][1] = have ? q[1] : 0.0f;
        kq[e][2] = have ? q[2] : 0.0f;
        kc[e][0] = cc[e].x;
        kc[e][1] = cc[e].y;
        kc[e][2] = cc[e].z;
      }
#pragma unroll
      for (int e = 0; e < KEEP; e++)
        if (kv[e]) {
          v[0] += 1.0;
          v[1] += kc[e][0];
          v[2] += kc[e][1];
          v[3] += kc[e][2];
          v[4] += kq[e][0];
          v[5] += kq[e][1];
          v[6] += kq[e][2];
          v[7] += (double)dd[e];
        }
    } else if (mom) {
      for (long long i = t; i < ns; i += PI_MOM) {
        float q3[3], c3[3], d;
        if (!corr(i, q3, c3, d)) continue;
        v[0] += 1.0;
        v[1] += c3[0];
        v[2] += c3[1];
        v[3] += c3[2];
        v[4] += q3[0];
        v[5] += q3[1];
        v[6] += q3[2];
        v[7] += (double)d;
      }
    }
    tree_sum(v, s_part, s_sum);
    ICP_MARK(2)
    double (&sums)[8] = v;
    const double cnt = sums[0];
    if (cnt < 3.0) {  // "Not enough correspondences found": no step is taken (the same decision in every workgroup)
      if (t == 0) {
        s_conv = 0;
        s_state = ST_NO_CORR;
      }
      break;
    }
    float ms[3], md[3];  // means rounded to float like the oracle's float means
    for (int r = 0; r < 3; r++) {
      ms[r] = (float)(sums[1 + r] / cnt);
      md[r] = (float)(sums[4 + r] / cnt);
    }
    double hv[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (!mom) {
    } else if (keep) {
#pragma unroll
      for (int e = 0; e < KEEP; e++)
        if (kv[e])
          for (int a = 0; a < 3; a++)
            for (int c = 0; c < 3; c++) hv[3 * a + c] += (double)__fmul_rn(__fsub_rn(kq[e][a], md[a]), __fsub_rn(kc[e][c], ms[c]));
    } else {
      for (long long i = t; i < ns; i += PI_MOM) {
        float q3[3], c3[3], d;
        if (!corr(i, q3, c3, d)) continue;
        for (int a = 0; a < 3; a++)
          for (int c = 0; c < 3; c++) hv[3 * a + c] += (double)__fmul_rn(__fsub_rn(q3[a], md[a]), __fsub_rn(c3[c], ms[c]));
      }
    }
    tree_sum(hv, s_part, s_sum);
    ICP_MARK(3)
    double (&cov)[9] = hv;
    // ---- update: one thread of every workgroup ----
    if (t == 0) {
      double H[9], R[9];
      for (int c = 0; c < 9; c++) H[c] = (double)(float)cov[c] / cnt;
      rotation_from_covariance(H, R);
      float step[16];
      for (int c = 0; c < 16; c++) step[c] = (c % 5 == 0) ? 1.0f : 0.0f;
      for (int a = 0; a < 3; a++) {
        for (int c = 0; c < 3; c++) step[4 * a + c] = (float)R[3 * a + c];
        step[4 * a + 3] = (float)((double)md[a] - (R[3 * a] * ms[0] + R[3 * a + 1] * ms[1] + R[3 * a + 2] * ms[2]));
      }
      float fin[16];
      for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
          float sacc = 0.0f;
          for (int c = 0; c < 4; c++) sacc = __fadd_rn(sacc, __fmul_rn(step[4 * i + c], s_final[4 * c + j]));
          fin[4 * i + j] = sacc;
        }
      for (int c = 0; c < 16; c++) {
        s_step[c] = step[c];
        s_final[c] = fin[c];
      }
      const int it = ++s_iter;
      const double mse = sums[7] / cnt;
      const double cos_angle = 0.5 * ((double)step[0] + (double)step[5] + (double)step[10] - 1.0);
      const double tsq = (double)step[3] * step[3] + (double)step[7] * step[7] + (double)step[11] * step[11];
      int conv = 0, st = ST_NOT;
      if (it >= A.max_iterations) {
        conv = 1;
        st = ST_ITER;
      } else if (cos_angle >= 1.0 - A.teps && tsq <= A.teps) {
        conv = 1;
        st = ST_TRANSFORM;
      } else if (fabs(mse - s_prev_mse) < 1e-12) {
        conv = 1;
        st = ST_ABS_MSE;
      } else if (fabs(mse - s_prev_mse) / s_prev_mse < A.feps) {
        conv = 1;
        st = ST_REL_MSE;
      } else {
        s_prev_mse = mse;
      }
      s_conv = conv;
      s_state = st;
    }
    __syncthreads();
    ICP_MARK(4)
    if (s_conv) break;
    k++;
  }
  __syncthreads();
  // ---- getFitnessScore(): nearest-neighbour distances of final * source (the record buffer iteration k cleared) ----
  unsigned long long *rec = best_of(k + 1);
  nn_pass(2, nullptr, nullptr, rec, nullptr);
  if (!grid::sync(m)) {
    gave_up();
    return;
  }
  if (b == 0) {
    double fs = 0.0, fc = 0.0;
    for (long long i = t; i < ns && t < PI_MOM; i += PI_MOM) {
      const unsigned long long r = grid::ld(rec + i);
      if (r == ~0ull) continue;
      fs += (double)__uint_as_float((unsigned)(r >> 32));
      fc += 1.0;
    }
    double f2[2] = {fs, fc};
    tree_sum(f2, s_part, s_sum);
    const double ts = f2[0], tc = f2[1];
    if (t == 0) {
      IcpState *S = A.S;
      for (int c = 0; c < 16; c++) S->final_t[c] = s_final[c];
      S->fit_sum = ts;
      S->fit_cnt = (unsigned long long)tc;
      S->iterations = s_iter;
      S->converged = s_conv;
      S->state = s_state;
      S->pad = 0;
      S->ns = ns;
      S->nt = nt;
#ifdef RSX_ICP_TIMING
      ICP_MARK(5)
      for (int c = 0; c < 8; c++) S->tm[c] = tm[c];
#endif
    }
  }
  grid::exit(m);
}

}  // namespace

struct rsx_icp {
  int device = 0;
  std::mutex mu;
  hipStream_t stream = nullptr;
  rsx::DevBuf src, tgt, cur, best, guess, bar;
  void *state_host = nullptr;  // pinned, device-visible: the kernel writes the result there (no read-back to enqueue)
  int n_wg = 0;  // workgroups of the persistent kernel: one per CU, at most PI_MAX_G
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
  for (rsx::DevBuf *b : {&h->src, &h->tgt, &h->cur, &h->best, &h->guess, &h->bar}) b->release();
  if (h->state_host) (void)hipHostFree(h->state_host);
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
// completed or runs on h's stream): ONE launch and one read-back.  d_ns / d_nt (optional): the sizes in device memory, written
// by work enqueued before this on the same stream; n_s / n_t are then upper bounds (what the buffers are sized for).
int align_device_counts_locked(rsx_icp *h, const void *d_src, int64_t n_s, const long long *d_ns, int64_t src_stride, const void *d_tgt,
                               int64_t n_t, const long long *d_nt, int64_t tgt_stride, const rsx_icp_params *params, const float *guess,
                               rsx_icp_result *out, int64_t *ns_out, int64_t *nt_out) {
  rsx_icp_params p;
  rsx_icp_default_params(&p);
  if (params) p = *params;
  if (p.max_iterations < 1 || !(p.max_corr_dist > 0)) return fail(RSX_ERR_BAD_ARG, "bad ICP params");
  if (n_t > 0xfffffffell || n_s > 0x7fffffffll) return fail(RSX_ERR_RANGE, "cloud too large");
  RSX_HIP(hipSetDevice(h->device));
  hipStream_t s = h->stream;
  if (!h->n_wg) {
    int cus = 0;
    RSX_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device));
    RSX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&icp_persistent_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)PI_DYN_LDS));
    // one workgroup per CU, and never more than the runtime says the device keeps resident (a CU-masked / partitioned device)
    const int limit = rsx::persistent::resident_limit(reinterpret_cast<const void *>(&icp_persistent_kernel), PI_NT, PI_DYN_LDS, h->device);
    if (limit < 1) return fail(RSX_ERR_HIP, "occupancy query of the persistent ICP kernel failed");
    h->n_wg = cus < 1 ? 1 : (cus > PI_MAX_G ? PI_MAX_G : cus);
    if (h->n_wg > limit) h->n_wg = limit;
    if (const char *e = rsx::exp_env("RSX_ICP_WGS")) h->n_wg = atoi(e);  // (experiments build: tools/watchdog_check.py oversubscribes the device)
  }
  const size_t cap = (size_t)(n_s > 0 ? n_s : 1);
  RSX_TRY(h->cur.reserve(2 * cap * 16, s, false));
  RSX_TRY(h->best.reserve(3 * cap * 8, s, false));
  if (!h->state_host) RSX_HIP(hipHostMalloc(&h->state_host, sizeof(IcpState), hipHostMallocDefault));
  RSX_TRY(h->guess.reserve(64, s, false));
  if (!h->bar.p) {
    RSX_TRY(h->bar.reserve(rsx::grid::BYTES, s, false));
    RSX_HIP(hipMemsetAsync(h->bar.p, 0, rsx::grid::BYTES, s));  // the kernel leaves the counters zero
  }
  if (guess) RSX_HIP(hipMemcpyAsync(h->guess.p, guess, 64, hipMemcpyHostToDevice, s));
  IcpArgs A;
  A.src = static_cast<const char *>(d_src);
  A.tgt = static_cast<const char *>(d_tgt);
  A.src_stride = src_stride;
  A.tgt_stride = tgt_stride;
  A.ns_ptr = d_ns;
  A.nt_ptr = d_nt;
  A.ns_imm = n_s;
  A.nt_imm = n_t;
  A.ns_cap = (long long)cap;
  A.guess = guess ? h->guess.as<float>() : nullptr;
  A.cur = h->cur.as<float4>();
  A.best = h->best.as<unsigned long long>();
  A.S = static_cast<IcpState *>(h->state_host);
  A.bar = h->bar.as<unsigned>();
  A.max_d2 = (float)(p.max_corr_dist * p.max_corr_dist);
  A.max_iterations = p.max_iterations;
  A.teps = p.transformation_epsilon;
  A.feps = p.euclidean_fitness_epsilon;
  {
    // One grid-barrier kernel at a time per device (rsx_persistent.h): this kernel's workgroups fill every CU (16 wavefronts at
    // 128 VGPRs) and wait for each other at grid barriers, and so do vg_coop_kernel's (voxelgrid.hip) -- two of them dispatched
    // side by side, from other handles, streams or threads, could each hold a part of the chip and wait for the rest until
    // the watchdog.  The gate orders them on the device; kernels without grid barriers beside it only delay it.
    rsx::persistent::Gate gate(h->device, s);
    RSX_TRY(gate.status());
    hipLaunchKernelGGL(icp_persistent_kernel, dim3((unsigned)h->n_wg), dim3(PI_NT), PI_DYN_LDS, s, A);
    RSX_HIP(hipGetLastError());
  }
  RSX_HIP(hipStreamSynchronize(s));
  const IcpState hstate = *static_cast<const IcpState *>(h->state_host);
  if (hstate.state == ST_GAVE_UP)
    return fail(RSX_ERR_HIP, "a grid barrier of the persistent kernel gave up after 5 s: its workgroups were not all resident (CUs masked, or another process holds a part of the device)");
  std::memcpy(out->transform, hstate.final_t, sizeof(out->transform));
  out->fitness = hstate.fit_cnt ? hstate.fit_sum / (double)hstate.fit_cnt : 1.7976931348623157e308;
  out->iterations = hstate.iterations;
  out->converged = hstate.converged;
  out->state = hstate.state;
  out->reserved = 0;
#ifdef RSX_ICP_TIMING
  fprintf(stderr, "icp timing (us): nn-load %.1f nn-scan %.1f nn-merge %.1f barrier %.1f sums %.1f cov %.1f update %.1f fitness %.1f  iterations %d\n", hstate.tm[6] * 0.01, hstate.tm[7] * 0.01, hstate.tm[0] * 0.01, hstate.tm[1] * 0.01,
          hstate.tm[2] * 0.01, hstate.tm[3] * 0.01, hstate.tm[4] * 0.01, hstate.tm[5] * 0.01, hstate.iterations);
#endif
  if (ns_out) *ns_out = hstate.ns;
  if (nt_out) *nt_out = hstate.nt;
  return RSX_OK;
}

int align_device_locked(rsx_icp *h, const void *d_src, int64_t n_s, int64_t src_stride, const void *d_tgt, int64_t n_t, int64_t tgt_stride,
                        const rsx_icp_params *params, const float *guess, rsx_icp_result *out) {
  return align_device_counts_locked(h, d_src, n_s, nullptr, src_stride, d_tgt, n_t, nullptr, tgt_stride, params, guess, out, nullptr, nullptr);
}

}  // namespace icp
}  // namespace rsx
