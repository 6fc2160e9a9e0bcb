// frontend.hip -- the ORORA front end between the cen2019 keypoints and the solver on gfx950:
// polar -> Cartesian remap, ORB-style binary descriptors at the keypoints, brute-force Hamming knnMatch(2) + ratio.
//
// The reference gets these steps from its ORORA submodule (an empty directory in the reference checkout,
// .gitmodules:1-3; README.md:26-29), which calls OpenCV (cv::remap, cv::ORB, cv::BFMatcher) -- absent here.  This
// follows the published steps as restated in oracle/frontend_ref.c (PARITY UNPINNED; the header of that file lists
// every choice the absent sources would pin, among them: the 256 test pairs come from a seeded generator because
// OpenCV's learned pattern table cannot be reproduced offline, so descriptors are not byte-compatible with cv::ORB).
//
// All kernels are small next to the 250 ms scan period (a 964 x 964 image, a few thousand keypoints); they exist so
// that the file-based entry stays on the GPU end to end and the matcher is the natural popcount kernel:
//   fe_remap      one thread per Cartesian pixel: 4 byte taps of the polar image, bilinear in fp32 (HBM-bound:
//                 7.4 MB of map + 3.7 MB out per scan)
//   fe_blur_*     separable 7-tap Gaussian
//   fe_describe   one WAVEFRONT per keypoint: intensity-centroid orientation from order-independent int64 moments
//                 (I quantised to 2^-24, like OpenCV's integer IC_Angle), 30-direction quantisation, 256 comparisons
//                 on the smoothed image through the rotated pair table as 4 ballots
//   fe_match      one thread per query descriptor, train descriptors staged through LDS, 8 x v_bcnt per pair
// fp32 arithmetic in a fixed order without contraction; every transcendental lives in host-side tables computed
// in double exactly like the oracle's, so GPU == oracle bit for bit.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "rsx_common.h"

namespace {

constexpr int HALF_PATCH = 15, NBINS = 30, NPAIRS = 256, BORDER = 19;

// blockIdx.y = image of a batch (images img_stride bytes apart, Cartesian images W * W floats apart)
// the (fractional) azimuth row a pixel at angle th (radians from "forward", in [0, 2 pi)) samples in an image whose
// azimuth grid starts at az[0] with step az[1] - az[0]: (th - az0) / step wrapped into [0, rows), in double like the oracle
// (frontend_ref.c) and like the host-built map of round 3 -- but PER IMAGE: a window of MulRan scans carries one encoder
// grid per scan, and a map built from the first scan's grid rotates every other scan's Cartesian image against its
// own keypoints (round-3 advisor finding)
__device__ __forceinline__ float az_row_of(double th, const float *__restrict__ az, int rows) {
  const double az0 = (double)az[0], st = (double)az[1] - az0, R = (double)rows;
  double a = (th - az0) / st;
  // fmod(a, rows) without the library call where it is a single exact subtraction: th and az0 lie in one turn and the step is
  // a turn / rows, so |a| < 2 rows -- and for rows <= |a| < 2 rows, |a| - rows is exact (Sterbenz), as fmod's result always is.
  // (The generic fmod was most of this function: ~150 fp64 instructions for each of the 93 M pixel evaluations of a window.)
  if (a >= R && a < 2.0 * R) a -= R;
  else if (a <= -R && a > -2.0 * R) a += R;
  else if (!(a > -R && a < R)) a = fmod(a, R);  // (an unusual grid, or NaN)
  if (a < 0) a += rows;
  if (a >= rows) a -= rows;
  return (float)a;
}

__global__ __launch_bounds__(256) void fe_remap(const uint8_t *__restrict__ imgs, int64_t img_stride, int rows, int cols, int row_stride,
                                                int col_offset, int W, const float *__restrict__ map_rb, const double *__restrict__ map_th,
                                                const float *__restrict__ az, int64_t az_stride, float *__restrict__ carts) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)W * W) return;
  const uint8_t *img = imgs + (int64_t)blockIdx.y * img_stride;
  float *cart = carts + (int64_t)blockIdx.y * W * W;
  const float rb = map_rb[i], ab = az_row_of(map_th[i], az + (int64_t)blockIdx.y * az_stride, rows);
  const float r0f = floorf(rb), a0f = floorf(ab);
  const float fr = rb - r0f, fa = ab - a0f;
  const int r0 = (int)r0f;
  int a0 = (int)a0f;
  if (a0 >= rows) a0 -= rows;
  const int a1 = (a0 + 1 == rows) ? 0 : a0 + 1;
  float p[2][2];
#pragma unroll
  for (int da = 0; da < 2; da++)
#pragma unroll
    for (int dr = 0; dr < 2; dr++) {
      const int r = r0 + dr, a = da ? a1 : a0;
      p[da][dr] = (r >= 0 && r < cols) ? __fdiv_rn((float)img[(int64_t)a * row_stride + col_offset + r], 255.0f) : 0.0f;
    }
  const float top = p[0][0] + fr * (p[0][1] - p[0][0]);
  const float bot = p[1][0] + fr * (p[1][1] - p[1][0]);
  cart[i] = top + fa * (bot - top);
}

// one pass of the separable Gaussian along x (horizontal = true) or y; BORDER_REFLECT_101
template <bool HORIZONTAL>
__global__ __launch_bounds__(256) void fe_blur(const float *__restrict__ ins, int W, const float *__restrict__ g, float *__restrict__ outs) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)W * W) return;
  const float *in = ins + (int64_t)blockIdx.y * W * W;
  float *out = outs + (int64_t)blockIdx.y * W * W;
  const int v = (int)(i / W), u = (int)(i - (int64_t)v * W);
  float s = 0.0f;
#pragma unroll
  for (int t = 0; t < 7; t++) {
    int x = (HORIZONTAL ? u : v) + t - 3;
    if (x < 0) x = -x;
    if (x >= W) x = 2 * W - 2 - x;
    s = s + g[t] * (HORIZONTAL ? in[(int64_t)v * W + x] : in[(int64_t)x * W + u]);
  }
  out[i] = s;
}

// fe_remap + both passes of fe_blur in ONE kernel (the batched path): a block produces a 32 x 32 tile of the Cartesian image
// and of its smoothed copy from a 38 x 38 remapped neighbourhood in LDS (halo 3 = the 7-tap kernel; coordinates outside
// the image are reflected, BORDER_REFLECT_101, before the remap -- the same pixels the separate passes read).  Same
// arithmetic per pixel in the same order, so the two images are bit-identical to the three-kernel form; HBM traffic per
// scan drops from 3 writes + 2 reads of 3.7 MB to 2 writes.
constexpr int FT = 32, FH = 3, FTS = FT + 2 * FH;
__global__ __launch_bounds__(256) void fe_cart_fused(const uint8_t *__restrict__ imgs, int64_t img_stride, int rows, int cols, int row_stride,
                                                     int col_offset, int W, const float *__restrict__ map_rb, const double *__restrict__ map_th,
                                                     const float *__restrict__ az, int64_t az_stride, const float *__restrict__ g,
                                                     float *__restrict__ carts, float *__restrict__ blurs) {
  __shared__ float s_c[FTS][FTS + 1];
  __shared__ float s_t[FTS][FT + 1];
  __shared__ float s_tab[256];  // byte -> byte / 255 (correctly rounded once per block instead of four divisions per pixel)
  s_tab[threadIdx.x] = __fdiv_rn((float)threadIdx.x, 255.0f);
  __syncthreads();
  const uint8_t *img = imgs + (int64_t)blockIdx.z * img_stride;
  float *cart = carts + (int64_t)blockIdx.z * W * W, *blur = blurs + (int64_t)blockIdx.z * W * W;
  const int tu0 = blockIdx.x * FT, tv0 = blockIdx.y * FT;
  float gk[7];
#pragma unroll
  for (int t = 0; t < 7; t++) gk[t] = g[t];
  for (int idx = threadIdx.x; idx < FTS * FTS; idx += 256) {
    const int ly = idx / FTS, lx = idx - ly * FTS;
    int v = tv0 + ly - FH, u = tu0 + lx - FH;
    const bool inner = ly >= FH && ly < FH + FT && lx >= FH && lx < FH + FT && v < W && u < W;
    if (v < 0) v = -v;
    if (v >= W) v = 2 * W - 2 - v;
    if (u < 0) u = -u;
    if (u >= W) u = 2 * W - 2 - u;
    if (v < 0) v = 0;  // (tiles hanging far over the edge of a tiny image: values unused)
    if (u < 0) u = 0;
    const int64_t i = (int64_t)v * W + u;
    const float rb = map_rb[i], ab = az_row_of(map_th[i], az + (int64_t)blockIdx.z * az_stride, rows);
    const float r0f = floorf(rb), a0f = floorf(ab);
    const float fr = rb - r0f, fa = ab - a0f;
    const int r0 = (int)r0f;
    int a0 = (int)a0f;
    if (a0 >= rows) a0 -= rows;
    const int a1 = (a0 + 1 == rows) ? 0 : a0 + 1;
    float p[2][2];
#pragma unroll
    for (int da = 0; da < 2; da++)
#pragma unroll
      for (int dr = 0; dr < 2; dr++) {
        const int r = r0 + dr, a = da ? a1 : a0;
        p[da][dr] = (r >= 0 && r < cols) ? s_tab[img[(int64_t)a * row_stride + col_offset + r]] : 0.0f;
      }
    const float top = p[0][0] + fr * (p[0][1] - p[0][0]);
    const float bot = p[1][0] + fr * (p[1][1] - p[1][0]);
    const float c = top + fa * (bot - top);
    s_c[ly][lx] = c;
    if (inner) cart[i] = c;
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < FTS * FT; idx += 256) {  // rows first (horizontal taps, added left to right)
    const int ly = idx / FT, lx = idx - ly * FT;
    float sum = 0.0f;
#pragma unroll
    for (int t = 0; t < 7; t++) sum = sum + gk[t] * s_c[ly][lx + t];
    s_t[ly][lx] = sum;
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < FT * FT; idx += 256) {  // then columns
    const int ly = idx / FT, lx = idx - ly * FT;
    const int v = tv0 + ly, u = tu0 + lx;
    float sum = 0.0f;
#pragma unroll
    for (int t = 0; t < 7; t++) sum = sum + gk[t] * s_t[ly + t][lx];
    if (v < W && u < W) blur[(int64_t)v * W + u] = sum;
  }
}

// Round 6: the same two images from ONE WAVEFRONT PER COLUMN STRIP, no LDS tiles and no barriers.  A wavefront owns 58
// output columns (+ 3 halo columns either side = its 64 lanes) of one image and walks a segment of rows top to bottom:
//   * the remapped pixel c(y, x) of its lane (evaluated at the reflected coordinate outside the image, like the tile form);
//   * the row pass h(y, x) = sum_t g[t] c(y, x + t - 3), added left to right, from the six neighbouring lanes (ds_bpermute);
//   * the column pass from a 7-deep register window of h: blur(y - 3, x) = sum_t g[t] h(y - 6 + t, x), added top to bottom.
// Per pixel the same fp32 operations in the same order as fe_remap / fe_blur<true> / fe_blur<false>, so bit-identical.
// Against the 32 x 32 tile form: 64 / 58 x (seg + 6) / seg = 1.13-1.16 remap evaluations per pixel instead of 1.41, the
// index arithmetic per group of rows instead of per pixel, and RG pixels of a lane in flight at once (their two map loads, then
// their polar taps, issued together: the tile form waited for every tap of every pixel in turn).  The four waves of a block
// take four IMAGES of the same strip, so the strip's map lines (12 bytes per pixel, the largest input) are fetched once per block.
//
// The azimuth row (th - az0) / step is an fp64 division per pixel and image in the oracle.  Here: q = (th - az0) * (1 / step),
// which is within 3 * 2^-53 relative of the correctly rounded quotient d; after the wrap into [0, rows) the two differ by
// less than delta = rows * 2^-48 (the wraps are exact subtractions or one rounded addition each).  If the wrapped value is
// further than delta from 0 and from rows -- the only images of the wrap's thresholds -- and w - delta and w + delta round to
// the SAME float, then so does the oracle's value (rounding is monotone): the float is returned.  Otherwise (probability
// ~2^-24 per pixel, and every pixel exactly on the first azimuth) the exact az_row_of decides.  `delta_scale` widens delta
// for the test that drives a large share of the pixels through the exact branch.
constexpr int SW = 64 - 2 * FH;  // 58 output columns per wavefront
#ifndef FE_LW
#define FE_LW 16
#endif
#ifndef FE_TARGET
#define FE_TARGET 8192
#endif
// The polar taps are where the time goes (measured: the kernel with its taps replaced by coalesced loads runs in 0.64 of the
// time, without the smoothing or without the azimuth arithmetic in the same time): 64 lanes on 64 consecutive Cartesian pixels
// of one row reach across up to 380 range bins and several azimuth rows, a dozen cache lines per load instruction.  So the
// taps are fetched in a SQUARER lane layout -- LW columns x 64 / LW rows per instruction (16 x 4: a quarter of the reach in
// range) -- over a block of 64 columns x RG = 64 / LW rows, the remapped pixels cross to the one-lane-per-column layout of
// the smoothing through LDS (one write, seven reads per pixel, no barrier: a wavefront's LDS operations execute in order),
// and the two range taps of an azimuth row come as ONE 16-bit load.
constexpr int LW = FE_LW, RG = 64 / LW;  // gather layout: LW columns x RG rows per instruction; RG rows per group
constexpr int CROW = 64 + 2 * 4 + (LW == 16 ? 8 : 0);  // floats per staged row (4 pad either side; 80: rows of a 16 x 4 layout on disjoint banks)

struct AzGrid {
  double az0, rst, R, delta;
};

__device__ __forceinline__ float az_row_screened(double th, const AzGrid &G, const float *__restrict__ az, int rows) {
  const double q = (th - G.az0) * G.rst;
  double w = q;
  if (q >= G.R) w = q - G.R;
  if (q < 0.0) w = q + G.R;
  const float lo = (float)(w - G.delta), hi = (float)(w + G.delta);
  if (w > G.delta && w < G.R - G.delta && lo == hi) return lo;  // (|q| >= 2 rows, NaN, a zero or negative step: all fail here)
  return az_row_of(th, az, rows);
}

__global__ __launch_bounds__(256) void fe_cart_strip(const uint8_t *__restrict__ imgs, int64_t img_stride, int rows, int cols, int row_stride,
                                                     int col_offset, int W, const float *__restrict__ map_rb, const double *__restrict__ map_th,
                                                     const float *__restrict__ az, int64_t az_stride, const float *__restrict__ g,
                                                     float *__restrict__ carts, float *__restrict__ blurs, int n_images, int seg,
                                                     double delta_scale) {
  __shared__ float s_tab[257];  // byte -> byte / 255, correctly rounded; [256] = 0
  __shared__ float s_c[4][RG][CROW];
  s_tab[threadIdx.x] = __fdiv_rn((float)threadIdx.x, 255.0f);
  if (threadIdx.x == 0) s_tab[256] = 0.0f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // work item of this wavefront: (image, row segment), images fastest -- the four waves of a block share a strip and, with
  // four or more images, a segment: one fetch of the strip's map lines serves them all
  const int item = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + wave));
  const int segment = item / n_images, image = item - segment * n_images;
  if ((int64_t)segment * seg >= W) return;
  const uint8_t *img = imgs + (int64_t)image * img_stride + col_offset;
  float *cart = carts + (int64_t)image * W * W, *blur = blurs + (int64_t)image * W * W;
  const float *azi = az + (int64_t)image * az_stride;
  AzGrid G;
  G.az0 = (double)azi[0];
  G.rst = 1.0 / ((double)azi[1] - G.az0);
  G.R = (double)rows;
  G.delta = G.R * 0x1p-48 * delta_scale;
  float gk[7];
#pragma unroll
  for (int t = 0; t < 7; t++) gk[t] = g[t];

  auto col_of = [&](int c) {  // reflected image column of the strip's column c (0 .. 63), clamped far beyond the edge (values unused)
    const int ur = (int)blockIdx.y * SW + c - FH;
    int u = ur < 0 ? -ur : ur;
    if (u >= W) u = 2 * W - 2 - u;
    return u < 0 ? 0 : u;
  };
  auto row_of = [&](int y) {  // reflected row of the image, clamped for rows past the segment's halo (unused)
    int v = y < 0 ? -y : y;
    if (v >= W) v = 2 * W - 2 - v;
    return v < 0 ? 0 : (v >= W ? W - 1 : v);
  };
  // smoothing layout: lane = column of the strip
  const int u_raw = (int)blockIdx.y * SW + lane - FH;
  const bool inner = lane >= FH && lane < FH + SW && u_raw < W;
  // gather layout: lane = (row jr of the group, column cc of a block of LW); its RG pixels of a group: columns LW * k + cc of row jr
  const int jr = lane / LW, cc = lane - jr * LW;
  int ug[RG];
#pragma unroll
  for (int k = 0; k < RG; k++) ug[k] = col_of(LW * k + cc);
  float *stage_w = &s_c[wave][jr][4 + cc];            // + LW * k
  const float *stage_r = &s_c[wave][0][4 + lane - FH];  // + CROW * j + t
  const int y0 = segment * seg, y1 = (y0 + seg < W) ? y0 + seg : W;

  float hw[7];
#pragma unroll
  for (int t = 0; t < 7; t++) hw[t] = 0.0f;
  float rb_n[RG];
  double th_n[RG];
  {
    const int v = row_of(y0 - FH + jr);
#pragma unroll
    for (int k = 0; k < RG; k++) {
      const int64_t i = (int64_t)v * W + ug[k];
      rb_n[k] = map_rb[i];
      th_n[k] = map_th[i];
    }
  }
  for (int yb = y0 - FH; yb < y1 + FH; yb += RG) {
    float rb[RG];
    double th[RG];
#pragma unroll
    for (int k = 0; k < RG; k++) rb[k] = rb_n[k], th[k] = th_n[k];
    if (yb + RG < y1 + FH) {
      const int v = row_of(yb + RG + jr);
#pragma unroll
      for (int k = 0; k < RG; k++) {
        const int64_t i = (int64_t)v * W + ug[k];
        rb_n[k] = map_rb[i];
        th_n[k] = map_th[i];
      }
    }
    float fr[RG], fa[RG];
    int r0[RG], bs[RG];
    uint16_t w0[RG], w1[RG];
#pragma unroll
    for (int k = 0; k < RG; k++) {
      const float ab = az_row_screened(th[k], G, azi, rows);
      const float r0f = floorf(rb[k]), a0f = floorf(ab);
      fr[k] = rb[k] - r0f;
      fa[k] = ab - a0f;
      r0[k] = (int)r0f;
      int a0 = (int)a0f;
      if (a0 >= rows) a0 -= rows;
      const int a1 = (a0 + 1 == rows) ? 0 : a0 + 1;
      // the two range taps of an azimuth row as ONE 16-bit load at base = r0 clamped into [0, cols - 2]: base == r0 inside the
      // image; r0 == -1: the pair's low byte is tap 1; r0 == cols - 1: its high byte is tap 0; taps outside are zeroed below
      bs[k] = r0[k] < 0 ? 0 : (r0[k] > cols - 2 ? cols - 2 : r0[k]);
      __builtin_memcpy(&w0[k], img + (uint32_t)(a0 * row_stride + bs[k]), 2);
      __builtin_memcpy(&w1[k], img + (uint32_t)(a1 * row_stride + bs[k]), 2);
    }
#pragma unroll
    for (int k = 0; k < RG; k++) {
      const bool in0 = r0[k] >= 0 && r0[k] < cols, in1 = r0[k] + 1 >= 0 && r0[k] + 1 < cols;
      const uint32_t s0 = r0[k] == bs[k] ? 0u : 8u, s1 = 8u - s0;  // tap 0 = low byte and tap 1 = high byte where base == r0
      // (s_tab[256] = 0: a tap outside the image is a table index, not a branch)
      const uint32_t b00 = in0 ? ((w0[k] >> s0) & 255u) : 256u, b01 = in1 ? ((w0[k] >> s1) & 255u) : 256u;
      const uint32_t b10 = in0 ? ((w1[k] >> s0) & 255u) : 256u, b11 = in1 ? ((w1[k] >> s1) & 255u) : 256u;
      const float p00 = s_tab[b00], p01 = s_tab[b01], p10 = s_tab[b10], p11 = s_tab[b11];
      const float top = p00 + fr[k] * (p01 - p00);
      const float bot = p10 + fr[k] * (p11 - p10);
      stage_w[LW * k] = top + fa[k] * (bot - top);
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < RG; j++) {
      const int y = yb + j;
      if (y >= y1 + FH) break;  // (uniform)
      float ct[7];
#pragma unroll
      for (int t = 0; t < 7; t++) ct[t] = stage_r[CROW * j + t];
      if (inner && y >= y0 && y < y1) cart[(int64_t)y * W + u_raw] = ct[3];
      float h = 0.0f;
#pragma unroll
      for (int t = 0; t < 7; t++) h = h + gk[t] * ct[t];
#pragma unroll
      for (int t = 0; t < 6; t++) hw[t] = hw[t + 1];
      hw[6] = h;
      float sum = 0.0f;
#pragma unroll
      for (int t = 0; t < 7; t++) sum = sum + gk[t] * hw[t];
      const int yo = y - FH;
      if (inner && yo >= y0 && yo < y1) blur[(int64_t)yo * W + u_raw] = sum;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// metric keypoints (x forward, y right) -> nearest Cartesian pixel (u, v), in double like the host form of
// rsx_frontend_describe: blockIdx.y = image, xy [image][stride][2], counts[image] keypoints each
__global__ __launch_bounds__(256) void fe_uv(const float *__restrict__ xy, const int32_t *__restrict__ counts, int stride, double cmr,
                                             double cart_res, int32_t *__restrict__ uv) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  const int n = counts[blockIdx.y] < stride ? counts[blockIdx.y] : stride;
  if (k >= n) return;
  const int64_t o = ((int64_t)blockIdx.y * stride + k) * 2;
  uv[o] = (int32_t)round(((double)xy[o + 1] + cmr) / cart_res);
  uv[o + 1] = (int32_t)round((cmr - (double)xy[o]) / cart_res);
}

// blockIdx.y = image: keypoint k of image b at uv / desc / valid slot b * stride + k; counts == nullptr: n keypoints.
// ONE WAVEFRONT PER KEYPOINT (round 3; rounds 1-2: one thread per keypoint walking the 709-pixel patch and the 256 pairs
// alone).  The intensity-centroid moments are defined ORDER-INDEPENDENTLY, like OpenCV's integer IC_Angle on its 8-bit
// image: I_q = llrint(I * 2^24), m10 = sum dx * I_q, m01 = sum dy * I_q in int64 -- so the 64 lanes take the patch pixels
// in stride and the lanes' sums are added; the direction is the first maximum over the 30 bins of
// (double)m10 * c_b + (double)m01 * s_b (one multiply each, one add: no contraction); the 256 comparisons are 4 ballots.
// A block of 4 waves walks the keypoints k = 4 * blockIdx.x + wave, + 4 * gridDim.x, ...
// Round 6: the 709 pixels of the disc come from a table (12 per lane, the table padded with (0, 0) offsets whose weight is
// zero; before: 16 raster steps with a division by 31 and the circle test), I_q from the fp32 product -- I * 2^24 is exact in
// fp32 as in fp64 and the image holds values in [0, 1], far inside int32 -- and the lanes' sums and the maximum over the bins
// by DPP (before: 36 dependent ds_bpermute round trips a keypoint, most of its time).
__device__ __forceinline__ int fe_wave_sum_i32(int x) {  // every lane: the sum over the 64 lanes
  x += __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]
  x += __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xf, 0xf, true);   // quad_perm [2,3,0,1]
  x += __builtin_amdgcn_update_dpp(0, x, 0x141, 0xf, 0xf, true);  // row_half_mirror
  x += __builtin_amdgcn_update_dpp(0, x, 0x140, 0xf, 0xf, true);  // row_mirror: every lane = its row's sum
  return __builtin_amdgcn_readlane(x, 0) + __builtin_amdgcn_readlane(x, 16) + __builtin_amdgcn_readlane(x, 32) + __builtin_amdgcn_readlane(x, 48);
}
__device__ __forceinline__ long long fe_wave_sum_i64(long long p) {  // |p| < 2^40 per lane: as 16 low bits + the rest, two int32 sums
  const int lo = fe_wave_sum_i32((int)(p & 0xffff)), hi = fe_wave_sum_i32((int)(p >> 16));
  return ((long long)hi << 16) + (long long)lo;
}
template <int CTRL>
__device__ __forceinline__ double fe_dpp_f64(double x) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(x);
  const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)u, (int)(unsigned)u, CTRL, 0xf, 0xf, false);
  const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)(u >> 32), (int)(unsigned)(u >> 32), CTRL, 0xf, 0xf, false);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
#ifndef FE_DESC_BLOCKS
#define FE_DESC_BLOCKS 128
#endif
constexpr int DISC_IT = 12;  // 709 pixels of the radius-15 disc over 64 lanes
__global__ __launch_bounds__(256) void fe_describe(const float *__restrict__ carts, const float *__restrict__ blurs, int W,
                                                   const int32_t *__restrict__ uvs, int n, const int32_t *__restrict__ counts, int stride,
                                                   const float *__restrict__ dir_cs, const int8_t *__restrict__ pairs,
                                                   const int8_t *__restrict__ disc, uint32_t *__restrict__ descs, uint8_t *__restrict__ valids) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (counts) n = counts[blockIdx.y] < stride ? counts[blockIdx.y] : stride;
  if ((int)blockIdx.x * 4 + wave >= n) return;
  const float *cart = carts + (int64_t)blockIdx.y * W * W, *blur = blurs + (int64_t)blockIdx.y * W * W;
  const int32_t *uv = uvs + (int64_t)blockIdx.y * stride * 2;
  uint32_t *desc = descs + (int64_t)blockIdx.y * stride * 8;
  uint8_t *valid = valids + (int64_t)blockIdx.y * stride;
  int dxs[DISC_IT], dys[DISC_IT], offs[DISC_IT];  // this lane's pixels of the disc: the same for every keypoint
#pragma unroll
  for (int it = 0; it < DISC_IT; it++) {
    const char2 d = *reinterpret_cast<const char2 *>(disc + 2 * (it * 64 + lane));
    dxs[it] = d.x;
    dys[it] = d.y;
    offs[it] = d.y * W + d.x;
  }
  double dcs = 0.0, dsn = 0.0;
  if (lane < NBINS) dcs = (double)dir_cs[2 * lane], dsn = (double)dir_cs[2 * lane + 1];
  for (int k = blockIdx.x * 4 + wave; k < n; k += gridDim.x * 4) {
    const int u = uv[2 * k], v = uv[2 * k + 1];
    const bool ok = !(u < BORDER || v < BORDER || u >= W - BORDER || v >= W - BORDER);  // wave-uniform
    if (!ok) {
      if (lane < 8) desc[(int64_t)k * 8 + lane] = 0u;
      if (lane == 0) valid[k] = 0;
      continue;
    }
    const float *cc = cart + ((int64_t)v * W + u);
    float cv[DISC_IT];
#pragma unroll
    for (int it = 0; it < DISC_IT; it++) cv[it] = cc[offs[it]];
    long long m10 = 0, m01 = 0;
#pragma unroll
    for (int it = 0; it < DISC_IT; it++) {
      const int iq = __float2int_rn(cv[it] * 16777216.0f);
      m10 += (long long)dxs[it] * iq;
      m01 += (long long)dys[it] * iq;
    }
    const double dm10 = (double)fe_wave_sum_i64(m10), dm01 = (double)fe_wave_sum_i64(m01);
    double dd = -INFINITY;
    if (lane < NBINS) {
      const double a = dm10 * dcs, b = dm01 * dsn;
      dd = a + b;
    }
    // maximum over the lanes (finite values in the first 30, -inf elsewhere: fmax is a plain maximum): inside the rows of 16 by
    // DPP, the 30 bins live in rows 0 and 1
    double mx = dd;
    mx = fmax(mx, fe_dpp_f64<0xB1>(mx));
    mx = fmax(mx, fe_dpp_f64<0x4E>(mx));
    mx = fmax(mx, fe_dpp_f64<0x141>(mx));
    mx = fmax(mx, fe_dpp_f64<0x140>(mx));
    {
      const unsigned long long u0 = (unsigned long long)__double_as_longlong(mx);
      const unsigned l0 = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u0, 0), h0 = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u0 >> 32), 0);
      const unsigned l1 = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u0, 16), h1 = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u0 >> 32), 16);
      mx = fmax(__longlong_as_double((long long)(((unsigned long long)h0 << 32) | l0)), __longlong_as_double((long long)(((unsigned long long)h1 << 32) | l1)));
    }
    const unsigned long long at = __ballot(lane < NBINS && dd == mx);
    const int bin = at ? __ffsll((long long)at) - 1 : 0;  // first maximum (NaN cannot occur: finite integers)
    const int8_t *pp = pairs + (int64_t)bin * NPAIRS * 4;
    const float *bc = blur + ((int64_t)v * W + u);
    float av[4], bv[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const char4 p4 = *reinterpret_cast<const char4 *>(pp + 4 * (64 * j + lane));
      av[j] = bc[p4.y * W + p4.x];
      bv[j] = bc[p4.w * W + p4.z];
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const unsigned long long bits = __ballot(av[j] < bv[j]);  // bit i of byte i / 8, little-endian words
      if (lane == 0) {
        desc[(int64_t)k * 8 + 2 * j] = (uint32_t)bits;
        desc[(int64_t)k * 8 + 2 * j + 1] = (uint32_t)(bits >> 32);
      }
    }
    if (lane == 0) valid[k] = 1;
  }
}

__global__ __launch_bounds__(256) void fe_match(const uint32_t *__restrict__ q, const uint8_t *__restrict__ qv, int nq,
                                                const uint32_t *__restrict__ t, const uint8_t *__restrict__ tv, int nt, float ratio,
                                                int32_t *__restrict__ out_idx, int32_t *__restrict__ out_d1, int32_t *__restrict__ out_d2) {
  __shared__ uint32_t st[256 * 8];
  __shared__ uint8_t sv[256];
  const int i = blockIdx.x * 256 + threadIdx.x;
  uint32_t me[8];
  const bool live = i < nq && qv[i];
#pragma unroll
  for (int w = 0; w < 8; w++) me[w] = i < nq ? q[(int64_t)i * 8 + w] : 0u;
  int d1 = 1 << 30, d2 = 1 << 30, i1 = -1;
  for (int j0 = 0; j0 < nt; j0 += 256) {
    __syncthreads();
    const int j = j0 + threadIdx.x;
#pragma unroll
    for (int w = 0; w < 8; w++) st[threadIdx.x * 8 + w] = j < nt ? t[(int64_t)j * 8 + w] : 0u;
    sv[threadIdx.x] = j < nt ? tv[j] : 0;
    __syncthreads();
    const int lim = nt - j0 < 256 ? nt - j0 : 256;
    if (live)
      for (int jj = 0; jj < lim; jj++) {  // ascending train index: the first minimum wins, like BFMatcher's scan
        if (!sv[jj]) continue;
        int d = 0;
#pragma unroll
        for (int w = 0; w < 8; w++) d += __popc(me[w] ^ st[jj * 8 + w]);
        if (d < d1) {
          d2 = d1;
          d1 = d;
          i1 = j0 + jj;
        } else if (d < d2) {
          d2 = d;
        }
      }
  }
  if (i < nq) {
    out_d1[i] = i1 >= 0 ? d1 : -1;
    out_d2[i] = d2 < (1 << 30) ? d2 : -1;
    out_idx[i] = (i1 >= 0 && d2 < (1 << 30) && (float)d1 < ratio * (float)d2) ? i1 : -1;
  }
}

// ordered list of the VALID keypoints of slot (first + blockIdx.x): vidx[blockIdx.x][j] = index of the j-th valid one,
// vcount[blockIdx.x] = how many (only ~40 % of the keypoints of a 200 m scan fall inside the 250 m Cartesian image)
__global__ __launch_bounds__(256) void fe_compact_valid(const uint8_t *__restrict__ valids, const int32_t *__restrict__ counts, int stride,
                                                        int first, int32_t *__restrict__ vidx, int32_t *__restrict__ vcount) {
  __shared__ int s_w[4];
  const int slot = first + blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = counts[slot] < stride ? counts[slot] : stride;
  const uint8_t *v = valids + (int64_t)slot * stride;
  int32_t *out = vidx + (int64_t)blockIdx.x * stride;
  int run = 0;
  for (int base = 0; base < n; base += 256) {
    const int i = base + threadIdx.x;
    const bool ok = i < n && v[i];
    const unsigned long long bal = __ballot(ok);
    if (lane == 0) s_w[wave] = __popcll(bal);
    __syncthreads();
    int before = run, total = 0;
    for (int w = 0; w < 4; w++) {
      if (w < wave) before += s_w[w];
      total += s_w[w];
    }
    if (ok) out[before + __popcll(bal & ((1ull << lane) - 1ull))] = i;
    run += total;
    __syncthreads();
  }
  if (threadIdx.x == 0) vcount[blockIdx.x] = run;
}

// knnMatch(2) + ratio between CONSECUTIVE scans of a batch: blockIdx.y = pair j (slots first + j and first + j + 1 of the
// [slot][stride] arrays), blockIdx.z = direction: 0 queries = slot A against train = slot B -> fwd[j][i], 1 the reverse
// -> bwd[j][i].  Queries and train descriptors are taken from the compacted lists of valid keypoints (ascending index, so
// "the first minimum wins" is the scan order of fe_match / BFMatcher); invalid queries get -1.
// FOUR LANES PER QUERY: a block of 256 threads takes 64 queries, lane part p scans entries p, p + 4, ... of every train
// tile and the four partial (d1, i1, d2) are merged under the order of the sequential scan (smaller distance, then the
// smaller train index) -- a lone thread per query scanning 800 descriptors left most of the chip idle.
constexpr int FM_Q = 64;  // queries per block
__global__ __launch_bounds__(256) void fe_match_consecutive(const uint32_t *__restrict__ descs, const uint8_t *__restrict__ valids,
                                                            const int32_t *__restrict__ counts, int stride, int first,
                                                            const int32_t *__restrict__ vidx, const int32_t *__restrict__ vcount, float ratio,
                                                            int32_t *__restrict__ fwd, int32_t *__restrict__ bwd) {
  __shared__ uint32_t st[256 * 8];
  __shared__ int32_t si[256];
  const int j = blockIdx.y, dir = blockIdx.z;
  const int qs = j + dir, ts = j + 1 - dir;  // slots relative to `first`
  const int nq = counts[first + qs] < stride ? counts[first + qs] : stride;
  const uint32_t *q = descs + (int64_t)(first + qs) * stride * 8, *t = descs + (int64_t)(first + ts) * stride * 8;
  const int32_t *qi = vidx + (int64_t)qs * stride, *ti = vidx + (int64_t)ts * stride;
  int32_t *out_idx = (dir ? bwd : fwd) + (int64_t)j * stride;
  // keypoints without a descriptor match nothing: block b clears the invalid ones among keypoints [256 b, 256 b + 256)
  for (int i = blockIdx.x * 256 + threadIdx.x; i < nq; i += gridDim.x * 256)
    if (!valids[(int64_t)(first + qs) * stride + i]) out_idx[i] = -1;
  const int nqv = vcount[qs], ntv = vcount[ts];
  for (int q0 = blockIdx.x * FM_Q; q0 < nqv; q0 += gridDim.x * FM_Q) {  // (uniform per block)
    const int jq = q0 + (threadIdx.x >> 2), part = threadIdx.x & 3;
    const bool live = jq < nqv;
    const int iq = live ? qi[jq] : 0;
    uint32_t me[8];
#pragma unroll
    for (int w = 0; w < 8; w++) me[w] = live ? q[(int64_t)iq * 8 + w] : 0u;
    // the two smallest (distance, train index) pairs as ONE number each, distance << 20 | index (distances <= 256, indices
    // below `stride` <= 2^20): "smaller distance, then the smaller train index" is the numbers' order, the second smallest
    // number carries the second smallest distance (a tie of two entries counts twice, as in the sequential scan), and an
    // update is min / max / min instead of two compares and five selects
    uint32_t k1 = 0xffffffffu, k2 = 0xffffffffu;
    for (int j0 = 0; j0 < ntv; j0 += 256) {
      __syncthreads();
      const int jt = j0 + threadIdx.x;
      const int it = jt < ntv ? ti[jt] : 0;
#pragma unroll
      for (int w = 0; w < 8; w++) st[threadIdx.x * 8 + w] = jt < ntv ? t[(int64_t)it * 8 + w] : 0u;
      si[threadIdx.x] = it;
      __syncthreads();
      const int lim = ntv - j0 < 256 ? ntv - j0 : 256;
      for (int jj = part; jj < lim; jj += 4) {
        uint32_t d = 0;
#pragma unroll
        for (int w = 0; w < 8; w++) d += (uint32_t)__popc(me[w] ^ st[jj * 8 + w]);
        const uint32_t key = (d << 20) | (uint32_t)si[jj];
        k2 = min(k2, max(k1, key));
        k1 = min(k1, key);
      }
    }
#pragma unroll
    for (int o = 1; o <= 2; o <<= 1) {  // merge the four shares
      const uint32_t o1 = (uint32_t)__shfl_xor((int)k1, o), o2 = (uint32_t)__shfl_xor((int)k2, o);
      k2 = min(max(k1, o1), min(k2, o2));
      k1 = min(k1, o1);
    }
    const int d1 = (int)(k1 >> 20), d2 = (int)(k2 >> 20), i1 = (int)(k1 & 0xfffffu);
    if (live && part == 0) out_idx[iq] = (k1 != 0xffffffffu && k2 != 0xffffffffu && (float)d1 < ratio * (float)d2) ? i1 : -1;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// fe_match_consecutive on the MATRIX cores (round 6).  A Hamming distance is a K = 256 contraction: with the bits as +-1,
// a . b = 256 - 2 hamming(a, b).  v_mfma_f32_32x32x64_f8f6f4 takes 64 fp8 values per lane pair and instruction, so FOUR of them
// give the 32 x 32 distances of a tile of 32 train descriptors against a wavefront's 32 queries -- exact small integers in fp32 --
// where the vector form spends 8 x (LDS read + xor + popcount) per pair and lane (27 instructions: 75 M pairs per 64-scan window,
// 113 us).  Rows = train descriptors (the A operand, a tile staged in LDS as fp8 bytes by the block's 256 threads -- one
// descriptor dword = 32 bytes each: 4 bits -> 4 bytes by one multiply, (n x 0x00204081) & 0x01010101, shifted into the sign
// bits of 0x38 = 1.0 -- rows 272 bytes apart: conflict-free ds_read_b128), columns = queries (the B operand, expanded once, in
// registers), so a lane holds ONE query's distances to 16 train descriptors per tile and keeps that query's two smallest
// (distance << 20 | train index) keys itself -- min / max / min per value, the order of the sequential scan as before -- with
// no traffic between the lanes until the two halves of the wavefront (rows r and r + 4) are merged at the end.
// A block = 4 wavefronts = 128 queries over double-buffered train tiles, one barrier per tile.
// ---------------------------------------------------------------------------------------------------------------
typedef int fe_i32x8 __attribute__((ext_vector_type(8)));
typedef float fe_f32x16 __attribute__((ext_vector_type(16)));
constexpr int MM_QB = 128;          // queries per block (32 per wavefront)
constexpr int MM_ROW = 256 + 16;    // bytes between two train rows of a staged tile
#ifndef FE_MM_RT
#define FE_MM_RT 1
#endif
#ifndef FE_MM_GX
#define FE_MM_GX 16  // query blocks per (pair, direction) in the grid (a block strides over the rest)
#endif
constexpr int MM_RT = FE_MM_RT;     // 32-row train tiles per stage (one barrier per stage)
__device__ __forceinline__ void fe_expand32(uint32_t bits, uint32_t (&out)[8]) {  // bit -> fp8 e4m3 byte: 0 -> 1.0 (0x38), 1 -> -1.0 (0xb8)
#pragma unroll
  for (int g = 0; g < 8; g++) out[g] = ((((bits >> (4 * g)) & 0xfu) * 0x00204081u) & 0x01010101u) << 7 | 0x38383838u;
}
__global__ __launch_bounds__(256) void fe_match_consecutive_mfma(const uint32_t *__restrict__ descs, const uint8_t *__restrict__ valids,
                                                                 const int32_t *__restrict__ counts, int stride, int first,
                                                                 const int32_t *__restrict__ vidx, const int32_t *__restrict__ vcount, float ratio,
                                                                 int32_t *__restrict__ fwd, int32_t *__restrict__ bwd) {
  __shared__ __attribute__((aligned(16))) uint8_t s_a[2][MM_RT * 32 * MM_ROW];
  __shared__ __attribute__((aligned(16))) uint32_t s_i[2][MM_RT * 32];
  // blockIdx.x = pair and direction, blockIdx.y = query block: workgroups go to the XCDs round robin by their linear index, and with
  // the query block as the fast index (16 per pair, 6 of them with work) the work sat on 6 of the 8 XCDs
  const int j = (int)blockIdx.x >> 1, dir = (int)blockIdx.x & 1;
  const int qblk = blockIdx.y, nqblk = gridDim.y;
  const int qs = j + dir, ts = j + 1 - dir;  // slots relative to `first`
  const int nq = counts[first + qs] < stride ? counts[first + qs] : stride;
  const uint32_t *q = descs + (int64_t)(first + qs) * stride * 8, *t = descs + (int64_t)(first + ts) * stride * 8;
  const int32_t *qi = vidx + (int64_t)qs * stride, *ti = vidx + (int64_t)ts * stride;
  int32_t *out_idx = (dir ? bwd : fwd) + (int64_t)j * stride;
  for (int i = qblk * 256 + threadIdx.x; i < nq; i += nqblk * 256)
    if (!valids[(int64_t)(first + qs) * stride + i]) out_idx[i] = -1;
  const int nqv = vcount[qs], ntv = vcount[ts];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5;
  constexpr int TR = MM_RT * 32;  // train rows per stage
  const int ntiles = (ntv + TR - 1) / TR;
  // staging: thread = (row r, dword w) of each 32-row tile of a stage.  The train indices of stage tl + 3 and the descriptor dwords
  // of stage tl + 2 are requested while stage tl is multiplied; the dwords of stage tl + 1 -- requested two stages ago -- are
  // expanded into the other buffer at the end of the iteration.  (First build: index and dword loaded, one after the other, in
  // the iteration before their tile -- two memory round trips in front of every barrier, 25 tiles a query block: 105 us in the
  // pipeline against the vector form's 121.)
  const int sr = threadIdx.x >> 3, sw = threadIdx.x & 7;
  struct Idx { int v[MM_RT]; };
  struct Bits { uint32_t v[MM_RT]; };
  auto idx_of = [&](int tile) -> Idx {
    Idx r;
#pragma unroll
    for (int u = 0; u < MM_RT; u++) {
      const int jt = tile * TR + 32 * u + sr;
      r.v[u] = (tile < ntiles && jt < ntv) ? ti[jt] : -1;
    }
    return r;
  };
  auto bits_of = [&](const Idx &it) -> Bits {
    Bits r;
#pragma unroll
    for (int u = 0; u < MM_RT; u++) r.v[u] = it.v[u] >= 0 ? t[(int64_t)it.v[u] * 8 + sw] : 0u;
    return r;
  };
  auto put = [&](const Idx &it, const Bits &bits, int buf) {
#pragma unroll
    for (int u = 0; u < MM_RT; u++) {
      uint32_t e[8];
      fe_expand32(bits.v[u], e);
      uint4 *dst = reinterpret_cast<uint4 *>(&s_a[buf][(32 * u + sr) * MM_ROW + sw * 32]);
      dst[0] = uint4{e[0], e[1], e[2], e[3]};
      dst[1] = uint4{e[4], e[5], e[6], e[7]};
      if (sw == 0) s_i[buf][32 * u + sr] = it.v[u] >= 0 ? (uint32_t)it.v[u] : 0xffffffffu;  // (a row past the list: its key saturates to "nothing")
    }
  };
  for (int q0 = qblk * MM_QB; q0 < nqv; q0 += nqblk * MM_QB) {  // (uniform per block)
    const int jq = q0 + wave * 32 + (lane & 31);
    const bool live = jq < nqv;
    const int iq = live ? qi[jq] : 0;
    Idx it0 = idx_of(0), it1 = idx_of(1), it2 = idx_of(2);
    fe_i32x8 bq[4];
#pragma unroll
    for (int st = 0; st < 4; st++) {
      uint32_t e[8];
      fe_expand32(live ? q[(int64_t)iq * 8 + 2 * st + half] : 0u, e);
      bq[st] = fe_i32x8{(int)e[0], (int)e[1], (int)e[2], (int)e[3], (int)e[4], (int)e[5], (int)e[6], (int)e[7]};
    }
    uint32_t k1 = 0xffffffffu, k2 = 0xffffffffu;
    float thr = -1.0e9f;
    Bits b1 = bits_of(it1);  // stage 1's dwords (stage 0's go straight into the buffer)
    __syncthreads();  // (the tile buffers of the previous query block are free)
    if (ntiles > 0) put(it0, bits_of(it0), 0);
    for (int tl = 0; tl < ntiles; tl++) {
      const Bits b2 = bits_of(it2);     // stage tl + 2
      const Idx it3 = idx_of(tl + 3);
      __syncthreads();  // stage tl is in its buffer; everybody is done with stage tl - 1
      const int buf = tl & 1;
      fe_f32x16 acc[MM_RT];
#pragma unroll
      for (int u = 0; u < MM_RT; u++) {
        acc[u] = fe_f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int st = 0; st < 4; st++) {
          const uint4 *src = reinterpret_cast<const uint4 *>(&s_a[buf][(32 * u + (lane & 31)) * MM_ROW + (2 * st + half) * 32]);
          const uint4 a0 = src[0], a1 = src[1];
          const fe_i32x8 av = {(int)a0.x, (int)a0.y, (int)a0.z, (int)a0.w, (int)a1.x, (int)a1.y, (int)a1.z, (int)a1.w};
          acc[u] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bq[st], acc[u], 0, 0, 0, 0, 0, 0);
        }
      }
      // D: this lane = query column lane & 31, rows (reg & 3) + 8 (reg >> 2) + 4 half.  A value can change the lane's two keys only
      // if its distance does not exceed the second one's, i.e. its dot product reaches thr = 256 - 2 d2: after the first tiles
      // that is rare (random descriptors lie 128 +- 8 apart, a query's second-best match near 100), so the values are
      // compared as floats -- one instruction each -- and the six-instruction key update runs only in the tiles where some lane
      // of the wavefront has such a value (the other lanes' updates change nothing)
#pragma unroll
      for (int u = 0; u < MM_RT; u++) {
        bool any = false;
#pragma unroll
        for (int r = 0; r < 16; r++) any |= acc[u][r] >= thr;
        if (__ballot(any) != 0ull) {  // (uniform)
#pragma unroll
          for (int g = 0; g < 4; g++) {
            const uint4 id = *reinterpret_cast<const uint4 *>(&s_i[buf][32 * u + 8 * g + 4 * half]);
            const uint32_t ids[4] = {id.x, id.y, id.z, id.w};
#pragma unroll
            for (int r = 0; r < 4; r++) {
              const uint32_t d = (uint32_t)__builtin_fmaf(acc[u][4 * g + r], -0.5f, 128.0f);  // hamming = 128 - dot / 2, exact
              const uint32_t key = (d << 20) | ids[r];
              k2 = min(k2, max(k1, key));
              k1 = min(k1, key);
            }
          }
          thr = 256.0f - 2.0f * (float)(k2 >> 20);  // (k2 = "nothing yet": far below every dot product)
        }
      }
      if (tl + 1 < ntiles) put(it1, b1, buf ^ 1);
      it1 = it2;
      it2 = it3;
      b1 = b2;
    }
    {  // the other half of the wavefront saw the other 16 rows of every tile
      const uint32_t o1 = (uint32_t)__shfl_xor((int)k1, 32), o2 = (uint32_t)__shfl_xor((int)k2, 32);
      k2 = min(max(k1, o1), min(k2, o2));
      k1 = min(k1, o1);
    }
    const int d1 = (int)(k1 >> 20), d2 = (int)(k2 >> 20), i1 = (int)(k1 & 0xfffffu);
    if (live && half == 0) out_idx[iq] = (k1 != 0xffffffffu && k2 != 0xffffffffu && (float)d1 < ratio * (float)d2) ? i1 : -1;
  }
}

}  // namespace

struct rsx_frontend {
  int device = 0, rows = 0, cols = 0, W = 0;
  double cart_res = 0.0;
  std::mutex mu;
  hipStream_t stream = nullptr;
  rsx::DevBuf img, map_rb, map_th, az1, cart, tmp, blur, tables, uv, desc, valid, q, qv, t, tv, m_idx, m_d1, m_d2, vidx, vcount;
  // the map depends on the radar's range resolution and azimuth grid: rebuilt only when they change
  double map_radar_res = -1.0;
  bool have_image = false;
  int batch_n = 0;  // Cartesian images held by the last rsx_frontend_cartesian* call
  bool three_pass = false;  // RSX_FRONTEND_THREE_PASS: remap and the two blur passes as separate kernels
  bool tiles = false;       // RSX_FRONTEND_TILES: the 32 x 32 tile kernel of round 3 instead of the strip kernel
  bool exact_az = false;    // RSX_FRONTEND_EXACT_AZIMUTH: every azimuth row through the fp64 division
};

using rsx::fail;

namespace {

// host-side tables, computed in double exactly like oracle/frontend_ref.c (they hold every transcendental of the path)
void build_tables(float *gauss7, float *dir_cs, int8_t *pairs) {
  double g[7], s = 0;
  for (int i = 0; i < 7; i++) {
    g[i] = std::exp(-0.5 * (i - 3) * (i - 3) / 4.0);  // sigma 2
    s += g[i];
  }
  for (int i = 0; i < 7; i++) gauss7[i] = (float)(g[i] / s);
  for (int b = 0; b < NBINS; b++) {
    dir_cs[2 * b] = (float)std::cos(b * 2.0 * M_PI / NBINS);
    dir_cs[2 * b + 1] = (float)std::sin(b * 2.0 * M_PI / NBINS);
  }
  int base[NPAIRS][4];
  uint64_t st = 0x9E3779B97F4A7C15ull;  // seeded integer generator: points uniform in the disc of radius 13
  for (int i = 0; i < NPAIRS; i++)
    for (int e = 0; e < 2; e++)
      for (;;) {
        st = st * 6364136223846793005ull + 1442695040888963407ull;
        const int x = (int)((st >> 33) % 27) - 13;
        st = st * 6364136223846793005ull + 1442695040888963407ull;
        const int y = (int)((st >> 33) % 27) - 13;
        if (x * x + y * y <= 13 * 13) {
          base[i][2 * e] = x;
          base[i][2 * e + 1] = y;
          break;
        }
      }
  for (int b = 0; b < NBINS; b++) {
    const double c = std::cos(b * 2.0 * M_PI / NBINS), sn = std::sin(b * 2.0 * M_PI / NBINS);
    for (int i = 0; i < NPAIRS; i++)
      for (int e = 0; e < 2; e++) {
        const double x = base[i][2 * e], y = base[i][2 * e + 1];
        pairs[((size_t)b * NPAIRS + i) * 4 + 2 * e] = (int8_t)std::lround(x * c - y * sn);
        pairs[((size_t)b * NPAIRS + i) * 4 + 2 * e + 1] = (int8_t)std::lround(x * sn + y * c);
      }
  }
}

constexpr size_t TAB_GAUSS = 0, TAB_DIR = 64, TAB_PAIRS = 64 + NBINS * 2 * 4, TAB_DISC = TAB_PAIRS + (size_t)NBINS * NPAIRS * 4,
                 TAB_BYTES = TAB_DISC + (size_t)DISC_IT * 64 * 2;

// (dx, dy) of the disc's pixels in raster order, padded with (0, 0) -- weight zero in both moments -- to 12 x 64 entries
void build_disc(int8_t *disc) {
  int n = 0;
  for (int dy = -HALF_PATCH; dy <= HALF_PATCH; dy++)
    for (int dx = -HALF_PATCH; dx <= HALF_PATCH; dx++)
      if (dx * dx + dy * dy <= HALF_PATCH * HALF_PATCH) {
        disc[2 * n] = (int8_t)dx;
        disc[2 * n + 1] = (int8_t)dy;
        n++;
      }
  for (; n < DISC_IT * 64; n++) disc[2 * n] = disc[2 * n + 1] = 0;
}

double cart_min_range(int W, double cart_res) { return (W % 2 == 0) ? (W / 2 - 0.5) * cart_res : (W / 2) * cart_res; }

}  // namespace

extern "C" {

int rsx_frontend_default_params(rsx_frontend_params *p) try {
  if (!p) return fail(RSX_ERR_BAD_ARG, "null params");
  p->cart_pixel_width = 964;   // yeti / ORORA defaults for the Navtech CIR204-H (recollection, parameterised)
  p->cart_resolution = 0.2592f;
  p->ratio = 0.8f;
  p->flags = 0;
  return RSX_OK;
} RSX_CATCH_ALL

int rsx_frontend_create(int device, int32_t rows, int32_t cols, const rsx_frontend_params *params, rsx_frontend **out) try {
  if (!out) return fail(RSX_ERR_BAD_ARG, "null out");
  *out = nullptr;
  rsx_frontend_params dp;
  rsx_frontend_default_params(&dp);
  if (params) dp = *params;
  if (rows < 2 || cols < 2 || dp.cart_pixel_width < 2 * BORDER + 1 || dp.cart_pixel_width > 8192 || !(dp.cart_resolution > 0.0f))
    return fail(RSX_ERR_BAD_ARG, "bad image shape / Cartesian parameters");
  const int ndev = rsx_device_count();
  if (ndev <= 0) return fail(RSX_ERR_NO_DEVICE, "no HIP device visible (librsx has no CPU fallback)");
  if (device < 0 || device >= ndev) return fail(RSX_ERR_NO_DEVICE, "device %d out of range (%d visible)", device, ndev);
  rsx_frontend *h = new (std::nothrow) rsx_frontend();
  if (!h) return fail(RSX_ERR_OOM, "host alloc");
  h->device = device;
  h->rows = rows;
  h->cols = cols;
  h->W = dp.cart_pixel_width;
  h->cart_res = (double)dp.cart_resolution;
  h->three_pass = (dp.flags & RSX_FRONTEND_THREE_PASS) != 0;
  h->tiles = (dp.flags & RSX_FRONTEND_TILES) != 0;
  h->exact_az = (dp.flags & RSX_FRONTEND_EXACT_AZIMUTH) != 0;
  hipError_t e = hipSetDevice(device);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    delete h;
    return fail(RSX_ERR_HIP, "create: %s", hipGetErrorString(e));
  }
  const size_t npx = (size_t)h->W * h->W;
  int st = RSX_OK;
  for (rsx::DevBuf *b : {&h->map_rb, &h->map_th, &h->cart, &h->tmp, &h->blur})
    if (st == RSX_OK) st = b->reserve(npx * sizeof(float), h->stream, false);
  if (st == RSX_OK) st = h->tables.reserve(TAB_BYTES, h->stream, false);
  if (st == RSX_OK) {
    std::vector<uint8_t> tab(TAB_BYTES, 0);
    build_tables(reinterpret_cast<float *>(&tab[TAB_GAUSS]), reinterpret_cast<float *>(&tab[TAB_DIR]),
                 reinterpret_cast<int8_t *>(&tab[TAB_PAIRS]));
    build_disc(reinterpret_cast<int8_t *>(&tab[TAB_DISC]));
    e = hipMemcpy(h->tables.p, tab.data(), TAB_BYTES, hipMemcpyHostToDevice);
    if (e != hipSuccess) st = fail(RSX_ERR_HIP, "tables: %s", hipGetErrorString(e));
  }
  if (st != RSX_OK) {
    rsx_frontend_destroy(h);
    return st;
  }
  *out = h;
  return RSX_OK;
} RSX_CATCH_ALL

int rsx_frontend_destroy(rsx_frontend *h) try {
  if (!h) return RSX_OK;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  for (rsx::DevBuf *b : {&h->img, &h->map_rb, &h->map_th, &h->az1, &h->cart, &h->tmp, &h->blur, &h->tables, &h->uv, &h->desc, &h->valid, &h->q,
                         &h->qv, &h->t, &h->tv, &h->m_idx, &h->m_d1, &h->m_d2, &h->vidx, &h->vcount})
    b->release();
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return RSX_OK;
} RSX_CATCH_ALL

// (re)build the pixel -> (range bin, angle) map when the radar's range resolution changed (the azimuth grid is applied
// per image on the device: az_row_of)
static int ensure_map(rsx_frontend *h, float resolution, hipStream_t s) {
  const int W = h->W;
  const size_t npx = (size_t)W * W;
  if (h->map_radar_res == (double)resolution) return RSX_OK;
  // in double on the host: forward = azimuth 0, azimuth grows to the right
  std::vector<float> rb(npx);
  std::vector<double> th(npx);
  const double cmr = cart_min_range(W, h->cart_res);
  for (int v = 0; v < W; v++)
    for (int u = 0; u < W; u++) {
      const double fwd = cmr - v * h->cart_res, right = -cmr + u * h->cart_res;
      const double r = std::sqrt(fwd * fwd + right * right);
      double t = std::atan2(right, fwd);
      if (t < 0) t += 2.0 * M_PI;
      rb[(size_t)v * W + u] = (float)((r - (double)resolution / 2.0) / (double)resolution);
      th[(size_t)v * W + u] = t;
    }
  RSX_TRY(h->map_th.reserve(npx * sizeof(double), s, false));
  RSX_HIP(hipMemcpyAsync(h->map_rb.p, rb.data(), npx * sizeof(float), hipMemcpyHostToDevice, s));
  RSX_HIP(hipMemcpyAsync(h->map_th.p, th.data(), npx * sizeof(double), hipMemcpyHostToDevice, s));
  RSX_HIP(hipStreamSynchronize(s));  // rb / th are locals
  h->map_radar_res = (double)resolution;
  return RSX_OK;
}

// n device images -> n Cartesian images + smoothed copies in the handle's slots 0 .. n-1
// d_az: device azimuths, az_stride floats between the grids of consecutive images (0: one grid for all)
static int cartesian_device(rsx_frontend *h, const uint8_t *d_imgs, int n, int64_t img_stride, int32_t row_stride, int32_t col_offset,
                            const float *d_az, int64_t az_stride, hipStream_t s) {
  const int W = h->W;
  const size_t npx = (size_t)W * W;
  RSX_TRY(h->cart.reserve(npx * sizeof(float) * n, s, false));
  if (h->three_pass) RSX_TRY(h->tmp.reserve(npx * sizeof(float) * n, s, false));
  RSX_TRY(h->blur.reserve(npx * sizeof(float) * n, s, false));
  const float *g = reinterpret_cast<const float *>(static_cast<const char *>(h->tables.p) + TAB_GAUSS);
  if (h->three_pass) {  // the round-2 form (kept for the parity test of the fused kernel): remap, blur rows, blur columns
    const dim3 grid((unsigned)((npx + 255) / 256), (unsigned)n);
    hipLaunchKernelGGL(fe_remap, grid, dim3(256), 0, s, d_imgs, img_stride, h->rows, h->cols, row_stride, col_offset, W, h->map_rb.as<float>(),
                       h->map_th.as<double>(), d_az, az_stride, h->cart.as<float>());
    hipLaunchKernelGGL(fe_blur<true>, grid, dim3(256), 0, s, h->cart.as<float>(), W, g, h->tmp.as<float>());
    hipLaunchKernelGGL(fe_blur<false>, grid, dim3(256), 0, s, h->tmp.as<float>(), W, g, h->blur.as<float>());
  } else if (!h->tiles) {
    // row segments: enough (image, strip, segment) wavefronts to fill the device about once (8 per SIMD), but segments of 16 rows or
    // more (22 evaluated for 16 kept) -- 8 for fewer than four images, where the device is not full either way (measured, 1 / 4 /
    // 16 / 64 images: 11.5 / 24.5 / 69 / 305-338 us; the tile kernel: 15 / 37 / 131 / 557)
    const int strips = (W + SW - 1) / SW, min_seg = n < 4 ? 8 : 16;
    int segs = (int)((FE_TARGET + (int64_t)strips * n / 2) / ((int64_t)strips * n));
    if (segs > W / min_seg) segs = W / min_seg;
    if (segs < 1) segs = 1;
    const int seg = (W + segs - 1) / segs;
    segs = (W + seg - 1) / seg;
    const dim3 grid((unsigned)(((int64_t)n * segs + 3) / 4), (unsigned)strips);
    hipLaunchKernelGGL(fe_cart_strip, grid, dim3(256), 0, s, d_imgs, img_stride, h->rows, h->cols, row_stride, col_offset, W,
                       h->map_rb.as<float>(), h->map_th.as<double>(), d_az, az_stride, g, h->cart.as<float>(), h->blur.as<float>(), n, seg,
                       h->exact_az ? 0x1p60 : 1.0);
  } else {
    const dim3 grid((unsigned)((W + FT - 1) / FT), (unsigned)((W + FT - 1) / FT), (unsigned)n);
    hipLaunchKernelGGL(fe_cart_fused, grid, dim3(256), 0, s, d_imgs, img_stride, h->rows, h->cols, row_stride, col_offset, W,
                       h->map_rb.as<float>(), h->map_th.as<double>(), d_az, az_stride, g, h->cart.as<float>(), h->blur.as<float>());
  }
  RSX_HIP(hipGetLastError());
  h->have_image = true;
  h->batch_n = n;
  return RSX_OK;
}

int rsx_frontend_cartesian(rsx_frontend *h, const uint8_t *img, int32_t row_stride, int32_t col_offset, const float *azimuths,
                           float resolution, float *out_cart) try {
  if (!h || !img || !azimuths) return fail(RSX_ERR_BAD_ARG, "null arg");
  if (row_stride < col_offset + h->cols || col_offset < 0 || !(resolution > 0.0f)) return fail(RSX_ERR_BAD_ARG, "bad image layout");
  std::lock_guard<std::mutex> lk(h->mu);
  RSX_HIP(hipSetDevice(h->device));
  hipStream_t s = h->stream;
  if (!((double)azimuths[1] - (double)azimuths[0] > 0.0)) return fail(RSX_ERR_BAD_ARG, "azimuths must increase");
  RSX_TRY(ensure_map(h, resolution, s));
  const size_t ibytes = (size_t)h->rows * row_stride;
  RSX_TRY(h->img.reserve(ibytes, s, false));
  RSX_TRY(h->az1.reserve(8, s, false));
  RSX_HIP(hipMemcpyAsync(h->img.p, img, ibytes, hipMemcpyHostToDevice, s));
  RSX_HIP(hipMemcpyAsync(h->az1.p, azimuths, 8, hipMemcpyHostToDevice, s));
  RSX_TRY(cartesian_device(h, h->img.as<uint8_t>(), 1, (int64_t)ibytes, row_stride, col_offset, h->az1.as<float>(), 0, s));
  if (out_cart) RSX_HIP(hipMemcpyAsync(out_cart, h->cart.p, (size_t)h->W * h->W * sizeof(float), hipMemcpyDeviceToHost, s));
  RSX_HIP(hipStreamSynchronize(s));
  return RSX_OK;
} RSX_CATCH_ALL

int rsx_frontend_cartesian_batch_device(rsx_frontend *h, const uint8_t *d_imgs, int32_t n_images, int64_t image_stride_bytes, int32_t row_stride,
                                        int32_t col_offset, const float *azimuths, float resolution, void *stream) try {
  if (!h || !d_imgs || !azimuths || n_images < 1) return fail(RSX_ERR_BAD_ARG, "bad arg");
  if (row_stride < col_offset + h->cols || col_offset < 0 || !(resolution > 0.0f)) return fail(RSX_ERR_BAD_ARG, "bad image layout");
  if (n_images > 1 && image_stride_bytes < (int64_t)h->rows * row_stride) return fail(RSX_ERR_BAD_ARG, "image_stride_bytes smaller than an image");
  std::lock_guard<std::mutex> lk(h->mu);
  RSX_HIP(hipSetDevice(h->device));
  hipStream_t s = stream ? static_cast<hipStream_t>(stream) : h->stream;
  if (!((double)azimuths[1] - (double)azimuths[0] > 0.0)) return fail(RSX_ERR_BAD_ARG, "azimuths must increase");
  RSX_TRY(ensure_map(h, resolution, s));
  RSX_TRY(h->az1.reserve(8, s, false));
  RSX_HIP(hipMemcpyAsync(h->az1.p, azimuths, 8, hipMemcpyHostToDevice, s));  // (8 bytes of pageable memory: staged before the call returns)
  return cartesian_device(h, d_imgs, n_images, image_stride_bytes, row_stride, col_offset, h->az1.as<float>(), 0, s);
} RSX_CATCH_ALL

int rsx_frontend_cartesian_batch_device_az(rsx_frontend *h, const uint8_t *d_imgs, int32_t n_images, int64_t image_stride_bytes, int32_t row_stride,
                                           int32_t col_offset, const float *d_azimuths, int64_t azimuth_stride_floats, float resolution,
                                           void *stream) try {
  if (!h || !d_imgs || !d_azimuths || n_images < 1 || azimuth_stride_floats < 0) return fail(RSX_ERR_BAD_ARG, "bad arg");
  if (row_stride < col_offset + h->cols || col_offset < 0 || !(resolution > 0.0f)) return fail(RSX_ERR_BAD_ARG, "bad image layout");
  if (n_images > 1 && image_stride_bytes < (int64_t)h->rows * row_stride) return fail(RSX_ERR_BAD_ARG, "image_stride_bytes smaller than an image");
  std::lock_guard<std::mutex> lk(h->mu);
  RSX_HIP(hipSetDevice(h->device));
  hipStream_t s = stream ? static_cast<hipStream_t>(stream) : h->stream;
  RSX_TRY(ensure_map(h, resolution, s));
  return cartesian_device(h, d_imgs, n_images, image_stride_bytes, row_stride, col_offset, d_azimuths, azimuth_stride_floats, s);
} RSX_CATCH_ALL

int rsx_frontend_read_images(rsx_frontend *h, int32_t image, float *out_cart, float *out_blur) try {
  if (!h || image < 0) return fail(RSX_ERR_BAD_ARG, "bad arg");
  std::lock_guard<std::mutex> lk(h->mu);
  if (!h->have_image || image >= h->batch_n) return fail(RSX_ERR_BAD_ARG, "no Cartesian image %d in the handle", image);
  RSX_HIP(hipSetDevice(h->device));
  RSX_HIP(hipDeviceSynchronize());  // (a diagnostic: whatever stream produced the images)
  const size_t bytes = (size_t)h->W * h->W * sizeof(float);
  if (out_cart) RSX_HIP(hipMemcpy(out_cart, static_cast<const char *>(h->cart.p) + bytes * image, bytes, hipMemcpyDeviceToHost));
  if (out_blur) RSX_HIP(hipMemcpy(out_blur, static_cast<const char *>(h->blur.p) + bytes * image, bytes, hipMemcpyDeviceToHost));
  return RSX_OK;
} RSX_CATCH_ALL

int rsx_frontend_describe(rsx_frontend *h, const float *xy, int32_t n, uint8_t *out_desc, uint8_t *out_valid) try {
  if (!h || (!xy && n) || (!out_desc && n) || (!out_valid && n) || n < 0) return fail(RSX_ERR_BAD_ARG, "bad arg");
  if (n == 0) return RSX_OK;
  std::lock_guard<std::mutex> lk(h->mu);
  if (!h->have_image) return fail(RSX_ERR_BAD_ARG, "describe before rsx_frontend_cartesian");
  RSX_HIP(hipSetDevice(h->device));
  hipStream_t s = h->stream;
  // metric keypoint (x forward, y right) -> nearest pixel, in double on the host
  std::vector<int32_t> uv((size_t)2 * n);
  const double cmr = cart_min_range(h->W, h->cart_res);
  for (int k = 0; k < n; k++) {
    uv[2 * (size_t)k] = (int32_t)std::lround(((double)xy[2 * k + 1] + cmr) / h->cart_res);
    uv[2 * (size_t)k + 1] = (int32_t)std::lround((cmr - (double)xy[2 * k]) / h->cart_res);
  }
  RSX_TRY(h->uv.reserve((size_t)n * 8, s, false));
  RSX_TRY(h->desc.reserve((size_t)n * 32, s, false));
  RSX_TRY(h->valid.reserve((size_t)n, s, false));
  RSX_HIP(hipMemcpyAsync(h->uv.p, uv.data(), (size_t)n * 8, hipMemcpyHostToDevice, s));
  const char *tab = static_cast<const char *>(h->tables.p);
  hipLaunchKernelGGL(fe_describe, dim3((unsigned)((n + 3) / 4 < 2048 ? (n + 3) / 4 : 2048)), dim3(256), 0, s, h->cart.as<float>(), h->blur.as<float>(), h->W,
                     h->uv.as<int32_t>(), n, (const int32_t *)nullptr, n, reinterpret_cast<const float *>(tab + TAB_DIR),
                     reinterpret_cast<const int8_t *>(tab + TAB_PAIRS), reinterpret_cast<const int8_t *>(tab + TAB_DISC), h->desc.as<uint32_t>(),
                     h->valid.as<uint8_t>());
  RSX_HIP(hipGetLastError());
  RSX_HIP(hipMemcpyAsync(out_desc, h->desc.p, (size_t)n * 32, hipMemcpyDeviceToHost, s));
  RSX_HIP(hipMemcpyAsync(out_valid, h->valid.p, (size_t)n, hipMemcpyDeviceToHost, s));
  RSX_HIP(hipStreamSynchronize(s));  // also keeps uv alive until the copy is done
  return RSX_OK;
} RSX_CATCH_ALL

int rsx_frontend_describe_batch_device(rsx_frontend *h, const float *d_xy, const int32_t *d_counts, int32_t n_images, int32_t max_targets,
                                       uint8_t *d_desc, uint8_t *d_valid, void *stream) try {
  if (!h || !d_xy || !d_counts || !d_desc || !d_valid || n_images < 1 || max_targets < 1) return fail(RSX_ERR_BAD_ARG, "bad arg");
  std::lock_guard<std::mutex> lk(h->mu);
  if (!h->have_image || n_images > h->batch_n) return fail(RSX_ERR_BAD_ARG, "describe_batch: %d images, the last Cartesian batch holds %d", n_images, h->batch_n);
  RSX_HIP(hipSetDevice(h->device));
  hipStream_t s = stream ? static_cast<hipStream_t>(stream) : h->stream;
  RSX_TRY(h->uv.reserve((size_t)n_images * max_targets * 8, s, false));
  const double cmr = cart_min_range(h->W, h->cart_res);
  hipLaunchKernelGGL(fe_uv, dim3((unsigned)((max_targets + 255) / 256), (unsigned)n_images), dim3(256), 0, s, d_xy, d_counts, max_targets, cmr,
                     h->cart_res, h->uv.as<int32_t>());
  const char *tab = static_cast<const char *>(h->tables.p);
  // FE_DESC_BLOCKS (128) blocks of 4 waves per image walk the image's keypoints (the counts live on the device); 512 blocks -- most
  // of whose waves find nothing to do -- took 113 us per 64-scan window where 64 or 128 take 91, 256: 99.  (Requesting the next
  // keypoint's patch while the current one is finished: 110 us -- twelve more registers held across the direction and pair phases.)
  hipLaunchKernelGGL(fe_describe, dim3((unsigned)((max_targets + 3) / 4 < FE_DESC_BLOCKS ? (max_targets + 3) / 4 : FE_DESC_BLOCKS), (unsigned)n_images), dim3(256), 0, s, h->cart.as<float>(),
                     h->blur.as<float>(), h->W, h->uv.as<int32_t>(), 0, d_counts, max_targets, reinterpret_cast<const float *>(tab + TAB_DIR),
                     reinterpret_cast<const int8_t *>(tab + TAB_PAIRS), reinterpret_cast<const int8_t *>(tab + TAB_DISC),
                     reinterpret_cast<uint32_t *>(d_desc), d_valid);
  RSX_HIP(hipGetLastError());
  return RSX_OK;
} RSX_CATCH_ALL

int rsx_frontend_match_consecutive_device(rsx_frontend *h, const uint8_t *d_desc, const uint8_t *d_valid, const int32_t *d_counts,
                                          int32_t max_targets, int32_t first_slot, int32_t n_pairs, float ratio, int32_t *d_fwd,
                                          int32_t *d_bwd, void *stream) try {
  if (!h || !d_desc || !d_valid || !d_counts || !d_fwd || !d_bwd || max_targets < 1 || first_slot < 0 || n_pairs < 0)
    return fail(RSX_ERR_BAD_ARG, "bad arg");
  if (max_targets > (1 << 20)) return fail(RSX_ERR_BAD_ARG, "max_targets above 2^20 (the matcher packs a train index into 20 bits)");
  if (n_pairs == 0) return RSX_OK;
  std::lock_guard<std::mutex> lk(h->mu);
  RSX_HIP(hipSetDevice(h->device));
  hipStream_t s = stream ? static_cast<hipStream_t>(stream) : h->stream;
  RSX_TRY(h->vidx.reserve((size_t)(n_pairs + 1) * max_targets * 4, s, false));
  RSX_TRY(h->vcount.reserve((size_t)(n_pairs + 1) * 4, s, false));
  hipLaunchKernelGGL(fe_compact_valid, dim3((unsigned)(n_pairs + 1)), dim3(256), 0, s, d_valid, d_counts, max_targets, first_slot,
                     h->vidx.as<int32_t>(), h->vcount.as<int32_t>());
  static const bool valu_form = [] { const char *e = rsx::exp_env("RSX_FE_MATCH"); return e && e[0] == 'v'; }();  // experiments build: RSX_FE_MATCH=valu
  if (valu_form)
    hipLaunchKernelGGL(fe_match_consecutive, dim3((unsigned)((max_targets + FM_Q - 1) / FM_Q < 32 ? (max_targets + FM_Q - 1) / FM_Q : 32), (unsigned)n_pairs, 2), dim3(256), 0, s,
                       reinterpret_cast<const uint32_t *>(d_desc), d_valid, d_counts, max_targets, first_slot, h->vidx.as<int32_t>(),
                       h->vcount.as<int32_t>(), ratio, d_fwd, d_bwd);
  else
    hipLaunchKernelGGL(fe_match_consecutive_mfma, dim3((unsigned)(2 * n_pairs), (unsigned)((max_targets + MM_QB - 1) / MM_QB < FE_MM_GX ? (max_targets + MM_QB - 1) / MM_QB : FE_MM_GX)), dim3(256), 0, s,
                       reinterpret_cast<const uint32_t *>(d_desc), d_valid, d_counts, max_targets, first_slot, h->vidx.as<int32_t>(),
                       h->vcount.as<int32_t>(), ratio, d_fwd, d_bwd);
  RSX_HIP(hipGetLastError());
  return RSX_OK;
} RSX_CATCH_ALL

int rsx_frontend_match(rsx_frontend *h, const uint8_t *q_desc, const uint8_t *q_valid, int32_t nq, const uint8_t *t_desc,
                       const uint8_t *t_valid, int32_t nt, float ratio, int32_t *out_train_idx, int32_t *out_d1, int32_t *out_d2) try {
  if (!h || nq < 0 || nt < 0 || (nq && (!q_desc || !q_valid || !out_train_idx)) || (nt && (!t_desc || !t_valid)))
    return fail(RSX_ERR_BAD_ARG, "bad arg");
  if (nq == 0) return RSX_OK;
  std::lock_guard<std::mutex> lk(h->mu);
  RSX_HIP(hipSetDevice(h->device));
  hipStream_t s = h->stream;
  RSX_TRY(h->q.reserve((size_t)nq * 32, s, false));
  RSX_TRY(h->qv.reserve((size_t)nq, s, false));
  RSX_TRY(h->t.reserve((size_t)(nt ? nt : 1) * 32, s, false));
  RSX_TRY(h->tv.reserve((size_t)(nt ? nt : 1), s, false));
  RSX_TRY(h->m_idx.reserve((size_t)nq * 4, s, false));
  RSX_TRY(h->m_d1.reserve((size_t)nq * 4, s, false));
  RSX_TRY(h->m_d2.reserve((size_t)nq * 4, s, false));
  RSX_HIP(hipMemcpyAsync(h->q.p, q_desc, (size_t)nq * 32, hipMemcpyHostToDevice, s));
  RSX_HIP(hipMemcpyAsync(h->qv.p, q_valid, (size_t)nq, hipMemcpyHostToDevice, s));
  if (nt) {
    RSX_HIP(hipMemcpyAsync(h->t.p, t_desc, (size_t)nt * 32, hipMemcpyHostToDevice, s));
    RSX_HIP(hipMemcpyAsync(h->tv.p, t_valid, (size_t)nt, hipMemcpyHostToDevice, s));
  }
  hipLaunchKernelGGL(fe_match, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, s, h->q.as<uint32_t>(), h->qv.as<uint8_t>(), nq,
                     h->t.as<uint32_t>(), h->tv.as<uint8_t>(), nt, ratio, h->m_idx.as<int32_t>(), h->m_d1.as<int32_t>(),
                     h->m_d2.as<int32_t>());
  RSX_HIP(hipGetLastError());
  RSX_HIP(hipMemcpyAsync(out_train_idx, h->m_idx.p, (size_t)nq * 4, hipMemcpyDeviceToHost, s));
  if (out_d1) RSX_HIP(hipMemcpyAsync(out_d1, h->m_d1.p, (size_t)nq * 4, hipMemcpyDeviceToHost, s));
  if (out_d2) RSX_HIP(hipMemcpyAsync(out_d2, h->m_d2.p, (size_t)nq * 4, hipMemcpyDeviceToHost, s));
  RSX_HIP(hipStreamSynchronize(s));
  return RSX_OK;
} RSX_CATCH_ALL

}  // extern "C"
