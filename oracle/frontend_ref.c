/*
 * oracle/frontend_ref.c -- CPU ORACLE for the ORORA front end between the cen2019 keypoints and the solver
 * (SURVEY 8f rank 3): polar -> Cartesian remap, ORB-style binary descriptors at the keypoints, brute-force Hamming
 * knnMatch(k = 2) + ratio test.  TEST INFRASTRUCTURE ONLY.
 *
 * PARITY UNPINNED.  The reference's ORORA sources are an empty submodule (/root/reference/.gitmodules:1-3,
 * README.md:19,26-29); upstream derives from yeti_radar_odometry (README.md:100-111) and calls OpenCV
 * (cv::remap, cv::ORB::compute, cv::BFMatcher::knnMatch) -- none of which exist in this image.  This file restates
 * the published steps with every choice that the absent sources would pin written down:
 *   cartesian   yeti's radar_polar_to_cartesian: W x W image centred on the sensor, row 0 = farthest forward,
 *               forward = azimuth 0, azimuth grows clockwise (to the right); pixel -> (range bin, azimuth row) in
 *               double on the host, bilinear taps in fp32, rows wrap around, bins outside the scan read 0.
 *   describe    oriented BRIEF as in ORB: orientation from the intensity centroid of the radius-15 disc (moments in int64
 *               over pixels quantised to 2^-24: order independent, like cv::ORB's integer IC_Angle), quantised to 30
 *               directions of 12 deg (the original ORB table form); 256 intensity comparisons on the image smoothed by a
 *               7 x 7 Gaussian (sigma 2).  OpenCV's learned 256-pair pattern table cannot be reproduced offline: the pairs
 *               come from a seeded integer generator (uniform in the disc of radius 13) -- same structure, different
 *               table, so descriptors are not byte-compatible with cv::ORB.
 *   match       for every query descriptor the two smallest Hamming distances over the valid train descriptors (first
 *               index wins ties, like BFMatcher's linear scan); kept when d1 < ratio * d2 (fp32 comparison).
 * Arithmetic is written so that a GPU can reproduce it bit for bit: fp32 operations in a fixed order, no contraction
 * (built with -ffp-contract=off), every transcendental confined to tables computed here in double.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define FE_HALF_PATCH 15
#define FE_NBINS 30
#define FE_NPAIRS 256
#define FE_BORDER 19 /* 15 (patch) + 3 (blur) + 1: keypoints closer to the image border get no descriptor */

/* pixel -> (range bin, azimuth row) table, W*W float pairs; az0 / az_step in rad, rows azimuths of cols bins */
void feref_cart_map(int W, double cart_res, double radar_res, double az0, double az_step, int rows, float *map_rb, float *map_ab) {
  const double cmr = (W % 2 == 0) ? (W / 2 - 0.5) * cart_res : (W / 2) * cart_res;
  for (int v = 0; v < W; v++)
    for (int u = 0; u < W; u++) {
      const double fwd = cmr - v * cart_res, right = -cmr + u * cart_res;
      const double r = sqrt(fwd * fwd + right * right);
      double th = atan2(right, fwd);
      if (th < 0) th += 2.0 * M_PI;
      double ab = (th - az0) / az_step;
      ab = fmod(ab, (double)rows);
      if (ab < 0) ab += rows;
      if (ab >= rows) ab -= rows;
      map_rb[(size_t)v * W + u] = (float)((r - radar_res / 2.0) / radar_res);
      map_ab[(size_t)v * W + u] = (float)ab;
    }
}

/* cv::remap(INTER_LINEAR, BORDER_CONSTANT 0) with the azimuth axis wrapped; polar bytes / 255 */
void feref_cart_remap(const uint8_t *img, int rows, int cols, int row_stride, int col_offset, int W, const float *map_rb,
                      const float *map_ab, float *cart) {
  for (size_t i = 0; i < (size_t)W * W; i++) {
    const float rb = map_rb[i], ab = map_ab[i];
    const float r0f = floorf(rb), a0f = floorf(ab);
    const float fr = rb - r0f, fa = ab - a0f;
    const int r0 = (int)r0f;
    int a0 = (int)a0f;
    if (a0 >= rows) a0 -= rows;
    const int a1 = (a0 + 1 == rows) ? 0 : a0 + 1;
    float p[2][2];
    for (int da = 0; da < 2; da++)
      for (int dr = 0; dr < 2; dr++) {
        const int r = r0 + dr, a = da ? a1 : a0;
        p[da][dr] = (r >= 0 && r < cols) ? (float)img[(size_t)a * row_stride + col_offset + r] / 255.0f : 0.0f;
      }
    const float top = p[0][0] + fr * (p[0][1] - p[0][0]);
    const float bot = p[1][0] + fr * (p[1][1] - p[1][0]);
    cart[i] = top + fa * (bot - top);
  }
}

/* tables shared by every describe call: 7-tap Gaussian (sigma 2, normalised), 30 unit directions, 30 x 256 rotated pairs */
void feref_tables(float *gauss7, float *dir_cs /* 30 x 2 */, int8_t *pairs /* 30 x 256 x 4 */) {
  double g[7], s = 0;
  for (int i = 0; i < 7; i++) {
    g[i] = exp(-0.5 * (i - 3) * (i - 3) / 4.0);
    s += g[i];
  }
  for (int i = 0; i < 7; i++) gauss7[i] = (float)(g[i] / s);
  for (int b = 0; b < FE_NBINS; b++) {
    dir_cs[2 * b] = (float)cos(b * 2.0 * M_PI / FE_NBINS);
    dir_cs[2 * b + 1] = (float)sin(b * 2.0 * M_PI / FE_NBINS);
  }
  /* base pattern: integer points uniform in the disc of radius 13 (so that every rotation stays inside the patch) */
  int base[FE_NPAIRS][4];
  uint64_t st = 0x9E3779B97F4A7C15ull;
  for (int i = 0; i < FE_NPAIRS; i++)
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
  for (int b = 0; b < FE_NBINS; b++) {
    const double c = cos(b * 2.0 * M_PI / FE_NBINS), sn = sin(b * 2.0 * M_PI / FE_NBINS);
    for (int i = 0; i < FE_NPAIRS; i++)
      for (int e = 0; e < 2; e++) {
        const double x = base[i][2 * e], y = base[i][2 * e + 1];
        pairs[((size_t)b * FE_NPAIRS + i) * 4 + 2 * e] = (int8_t)lround(x * c - y * sn);
        pairs[((size_t)b * FE_NPAIRS + i) * 4 + 2 * e + 1] = (int8_t)lround(x * sn + y * c);
      }
  }
}

/* separable 7 x 7 Gaussian, BORDER_REFLECT_101, rows first then columns, taps added left to right */
void feref_blur(const float *img, int W, const float *g, float *tmp, float *out) {
  for (int v = 0; v < W; v++)
    for (int u = 0; u < W; u++) {
      float s = 0.0f;
      for (int t = 0; t < 7; t++) {
        int x = u + t - 3;
        if (x < 0) x = -x;
        if (x >= W) x = 2 * W - 2 - x;
        s = s + g[t] * img[(size_t)v * W + x];
      }
      tmp[(size_t)v * W + u] = s;
    }
  for (int v = 0; v < W; v++)
    for (int u = 0; u < W; u++) {
      float s = 0.0f;
      for (int t = 0; t < 7; t++) {
        int y = v + t - 3;
        if (y < 0) y = -y;
        if (y >= W) y = 2 * W - 2 - y;
        s = s + g[t] * tmp[(size_t)y * W + u];
      }
      out[(size_t)v * W + u] = s;
    }
}

/* metric keypoint (x forward, y right; what cen2019 emits) -> pixel (u, v); rounded to the nearest pixel */
static void kp_pixel(float x, float y, int W, double cart_res, int *u, int *v) {
  const double cmr = (W % 2 == 0) ? (W / 2 - 0.5) * cart_res : (W / 2) * cart_res;
  *u = (int)lround(((double)y + cmr) / cart_res);
  *v = (int)lround((cmr - (double)x) / cart_res);
}

/* descriptors of n keypoints: desc n x 32 bytes (bit i of byte i/8 = pair i), valid n bytes */
void feref_describe(const float *cart, const float *blur, int W, double cart_res, const float *xy, int n, const float *dir_cs,
                    const int8_t *pairs, uint8_t *desc, uint8_t *valid) {
  for (int k = 0; k < n; k++) {
    int u, v;
    kp_pixel(xy[2 * k], xy[2 * k + 1], W, cart_res, &u, &v);
    memset(desc + (size_t)k * 32, 0, 32);
    valid[k] = 0;
    if (u < FE_BORDER || v < FE_BORDER || u >= W - FE_BORDER || v >= W - FE_BORDER) continue;
    valid[k] = 1;
    /* intensity centroid over the disc of radius 15, ORDER-INDEPENDENT like OpenCV's integer IC_Angle on its 8-bit
     * image: every pixel quantised to I_q = llrint(I * 2^24), the moments summed in int64 (exact in any order) */
    long long m10 = 0, m01 = 0;
    for (int dy = -FE_HALF_PATCH; dy <= FE_HALF_PATCH; dy++)
      for (int dx = -FE_HALF_PATCH; dx <= FE_HALF_PATCH; dx++) {
        if (dx * dx + dy * dy > FE_HALF_PATCH * FE_HALF_PATCH) continue;
        const long long iq = llrint((double)cart[(size_t)(v + dy) * W + (u + dx)] * 16777216.0);
        m10 += (long long)dx * iq;
        m01 += (long long)dy * iq;
      }
    int bin = 0;
    double best = -INFINITY;
    for (int b = 0; b < FE_NBINS; b++) {
      const double pa = (double)m10 * (double)dir_cs[2 * b], pb = (double)m01 * (double)dir_cs[2 * b + 1];
      const double d = pa + pb; /* one multiply each, one add (no contraction): first maximum wins */
      if (d > best) {
        best = d;
        bin = b;
      }
    }
    const int8_t *pp = pairs + (size_t)bin * FE_NPAIRS * 4;
    for (int i = 0; i < FE_NPAIRS; i++) {
      const float a = blur[(size_t)(v + pp[4 * i + 1]) * W + (u + pp[4 * i])];
      const float b = blur[(size_t)(v + pp[4 * i + 3]) * W + (u + pp[4 * i + 2])];
      if (a < b) desc[(size_t)k * 32 + (i >> 3)] |= (uint8_t)(1u << (i & 7));
    }
  }
}

/* knnMatch(k = 2) + ratio: out_idx[i] = train index or -1, out_d1 / out_d2 the two smallest distances (-1 if none) */
void feref_match(const uint8_t *q, const uint8_t *qv, int nq, const uint8_t *t, const uint8_t *tv, int nt, float ratio,
                 int32_t *out_idx, int32_t *out_d1, int32_t *out_d2) {
  for (int i = 0; i < nq; i++) {
    int d1 = 1 << 30, d2 = 1 << 30, i1 = -1;
    if (qv[i])
      for (int j = 0; j < nt; j++) {
        if (!tv[j]) continue;
        int d = 0;
        for (int b = 0; b < 32; b++) d += __builtin_popcount((unsigned)(q[(size_t)i * 32 + b] ^ t[(size_t)j * 32 + b]));
        if (d < d1) {
          d2 = d1;
          d1 = d;
          i1 = j;
        } else if (d < d2) {
          d2 = d;
        }
      }
    out_d1[i] = i1 >= 0 ? d1 : -1;
    out_d2[i] = d2 < (1 << 30) ? d2 : -1;
    out_idx[i] = (i1 >= 0 && d2 < (1 << 30) && (float)d1 < ratio * (float)d2) ? i1 : -1;
  }
}
