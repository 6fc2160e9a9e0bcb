/*
 * oracle/cen2019_ref.c -- CPU ORACLE for cen2019 radar keypoint extraction.  TEST INFRASTRUCTURE ONLY.
 *
 * PARITY UNPINNED.  The reference obtains its "Cartesian 2D feature point cloud (extracted via
 * cen2019 method)" (README.md:29) from the ORORA submodule, which is an empty directory in
 * /root/reference (.gitmodules:1-3).  This restates the published method (Cen & Newman, ICRA
 * 2019, as implemented in yeti_radar_odometry which ORORA builds on, README.md:100-111) from
 * SURVEY.md Appendix B.2.  Where the public implementation's arithmetic is order-dependent
 * (means over 1.3 M pixels) this restatement fixes an ORDER-INDEPENDENT definition so that a
 * parallel implementation can match it bit for bit:
 *   - fft(a,r)   = (float)byte / 255.0f
 *   - mean(fft)  = (float)((double)(sum of bytes) / 255.0 / N)
 *   - mean(h)    = (float)((double)(sum over pixels of llrint(h * 2^40)) / 2^40 / N)
 * and the unspecified order of equal intensities in the descending sort is fixed to
 * (h descending, azimuth ascending, range ascending).
 *
 * Steps (B.2):
 *   1. g = |fft(a,r+1) - fft(a,r-1)| (reflect-101 borders => 0 at both ends), g /= max(g)
 *   2. s = fft - mean(fft);  h = s * (1 - g);  candidates: h > mean(h)
 *   3. visit candidates in descending h while fewer than max_points regions were opened:
 *      an unmarked candidate (a,r) marks [rlow,rhigh] = r extended over the adjacent pixels with
 *      s < 0 on both sides; it counts as a new region unless it touched an already marked pixel
 *   4. per azimuth, for r >= min_range: every maximal marked run that is closed by an unmarked
 *      pixel and has a marked pixel in the same range span on the azimuth above or below
 *      (wrap-around) yields the keypoint (a, argmax_r h) -- first maximum
 *   5. metres: range = (r + 0.5) * resolution, x = range cos(az[a]), y = range sin(az[a])
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  float h;
  int32_t idx; /* a * cols + r */
} cand;

static int cmp_cand(const void *pa, const void *pb) {
  const cand *a = (const cand *)pa, *b = (const cand *)pb;
  if (a->h > b->h) return -1;
  if (a->h < b->h) return 1;
  return (a->idx > b->idx) - (a->idx < b->idx);
}

/* img: rows x row_stride bytes, power samples at [col_offset, col_offset+cols) of every row.
 * out_targets: (a, r) int32 pairs, row-major order; returns the number of keypoints (may exceed
 * max_targets, in which case only the first max_targets are written).
 * debug outputs (optional): h image (rows*cols floats), mean_h, number of candidates, J*. */
int32_t cen2019ref_extract(const uint8_t *img, int32_t rows, int32_t cols, int32_t row_stride, int32_t col_offset,
                           int32_t max_points, int32_t min_range, int32_t *out_targets, int32_t max_targets,
                           float *dbg_h, float *dbg_mean_h, int64_t *dbg_ncand, int64_t *dbg_jstar) {
  const int64_t n = (int64_t)rows * cols;
  float *fft = (float *)malloc(sizeof(float) * (size_t)n);
  float *g = (float *)malloc(sizeof(float) * (size_t)n);
  float *s = (float *)malloc(sizeof(float) * (size_t)n);
  float *h = (float *)malloc(sizeof(float) * (size_t)n);
  int32_t *mark = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
  uint64_t sumb = 0;
  for (int32_t a = 0; a < rows; a++)
    for (int32_t r = 0; r < cols; r++) {
      const uint8_t b = img[(int64_t)a * row_stride + col_offset + r];
      fft[(int64_t)a * cols + r] = (float)b / 255.0f;
      sumb += b;
    }
  float maxg = 0.0f;
  for (int32_t a = 0; a < rows; a++)
    for (int32_t r = 0; r < cols; r++) {
      const float *row = fft + (int64_t)a * cols;
      const int32_t rp = (r + 1 < cols) ? r + 1 : cols - 2; /* reflect 101 */
      const int32_t rm = (r - 1 >= 0) ? r - 1 : 1;
      float d = (cols > 1) ? row[rp] - row[rm] : 0.0f;
      d = fabsf(d);
      g[(int64_t)a * cols + r] = d;
      if (d > maxg) maxg = d;
    }
  const float mean = (float)((double)sumb / 255.0 / (double)n);
  int64_t fix = 0;
  for (int64_t i = 0; i < n; i++) {
    const float gn = (maxg > 0.0f) ? g[i] / maxg : 0.0f;
    s[i] = fft[i] - mean;
    const float om = 1.0f - gn;
    h[i] = s[i] * om;
    fix += llrint((double)h[i] * 1099511627776.0); /* 2^40 */
  }
  const float mean_h = (float)((double)fix / 1099511627776.0 / (double)n);
  if (dbg_h) memcpy(dbg_h, h, sizeof(float) * (size_t)n);
  if (dbg_mean_h) *dbg_mean_h = mean_h;

  int64_t m = 0;
  for (int64_t i = 0; i < n; i++) m += h[i] > mean_h;
  cand *c = (cand *)malloc(sizeof(cand) * (size_t)(m ? m : 1));
  m = 0;
  for (int64_t i = 0; i < n; i++)
    if (h[i] > mean_h) {
      c[m].h = h[i];
      c[m].idx = (int32_t)i;
      m++;
    }
  qsort(c, (size_t)m, sizeof(cand), cmp_cand);
  if (dbg_ncand) *dbg_ncand = m;

  memset(mark, 0, sizeof(int32_t) * (size_t)n);
  int64_t j = 0;
  int32_t l = 0;
  while (l < max_points && j < m) {
    const int32_t a = c[j].idx / cols, r = c[j].idx % cols;
    int32_t *mr = mark + (int64_t)a * cols;
    const float *sr = s + (int64_t)a * cols;
    if (!mr[r]) {
      int32_t rlow = r, rhigh = r;
      for (int32_t i = r - 1; i >= 0; i--) {
        if (sr[i] < 0) rlow = i;
        else break;
      }
      for (int32_t i = r + 1; i < cols; i++) {
        if (sr[i] < 0) rhigh = i;
        else break;
      }
      int already = 0;
      for (int32_t i = rlow; i <= rhigh; i++) {
        if (mr[i]) {
          already = 1;
          continue;
        }
        mr[i] = 1;
      }
      if (!already) l++;
    }
    j++;
  }
  if (dbg_jstar) *dbg_jstar = j;

  int32_t count = 0;
  for (int32_t a = 0; a < rows; a++) {
    const int32_t *mr = mark + (int64_t)a * cols;
    const int32_t *below = mark + (int64_t)((a - 1 + rows) % rows) * cols;
    const int32_t *above = mark + (int64_t)((a + 1) % rows) * cols;
    const float *hr = h + (int64_t)a * cols;
    int32_t start = 0, end = 0, counting = 0;
    for (int32_t r = min_range < 0 ? 0 : min_range; r < cols; r++) {
      if (mr[r]) {
        if (!counting) {
          start = r;
          end = r;
          counting = 1;
        } else {
          end = r;
        }
      } else if (counting) {
        int adj = 0;
        for (int32_t i = start; i <= end && !adj; i++) adj = below[i] || above[i];
        if (adj) {
          int32_t max_r = start;
          float mx = -INFINITY;
          for (int32_t i = start; i <= end; i++)
            if (hr[i] > mx) {
              mx = hr[i];
              max_r = i;
            }
          if (count < max_targets) {
            out_targets[2 * count] = a;
            out_targets[2 * count + 1] = max_r;
          }
          count++;
        }
        counting = 0;
      }
    }
  }
  free(fft);
  free(g);
  free(s);
  free(h);
  free(mark);
  free(c);
  return count;
}

void cen2019ref_to_cartesian(const int32_t *targets, int32_t n, const float *azimuths, float resolution, float *out_xy) {
  for (int32_t i = 0; i < n; i++) {
    const float range = ((float)targets[2 * i + 1] + 0.5f) * resolution;
    const float az = azimuths[targets[2 * i]];
    out_xy[2 * i] = range * cosf(az);
    out_xy[2 * i + 1] = range * sinf(az);
  }
}
