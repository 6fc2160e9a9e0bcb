/*
 * oracle/voxelgrid_ref.c -- CPU ORACLE for the VoxelGrid downsample that feeds the ScanContext build.
 * TEST INFRASTRUCTURE ONLY.
 *
 * PARITY UNPINNED.  The reference calls pcl::VoxelGrid<pcl::PointXYZI> with a 0.4 m leaf right
 * before makeAndSaveScancontextAndKeys (pgo/SC-A-LOAM/src/laserPosegraphOptimization.cpp:98,
 * 482-484, 687-688).  PCL is a third-party dependency that is not vendored under /root/reference
 * and not installed in this image; its version is whatever the ROS distribution ships (1.8-1.10).
 * This restates the published algorithm of pcl/filters/impl/voxel_grid.hpp (applyFilter, default
 * settings: no filter field, downsample_all_data = true, min_points_per_voxel = 0):
 *   1. min / max of the finite points (float), inverse leaf = 1 / leaf (float)
 *   2. if (dx * dy * dz) with d = (int64)((max - min) * inv_leaf) + 1 exceeds INT32_MAX the filter
 *      gives up and returns the input unchanged ("Leaf size is too small")
 *   3. min_b = floor(min * inv_leaf), max_b = floor(max * inv_leaf) (as int), div_b = max_b - min_b + 1,
 *      divb_mul = (1, div_b.x, div_b.x * div_b.y)
 *   4. per finite point: ijk = (int)(floor(p * inv_leaf) - (float)min_b);  idx = ijk . divb_mul
 *   5. sort by idx; every run of equal idx becomes ONE output point = the centroid of its points:
 *      x, y, z, intensity each summed in float and divided by (float)count (CentroidPoint<PointXYZI>)
 *   6. output order = ascending idx
 * PCL sorts with std::sort (not stable), so the order of the float additions inside a voxel is
 * unspecified there; this restatement fixes it to ascending input index (a stable sort), which is
 * what the GPU radix sort produces.  Non-finite points are always skipped (PCL skips them only when
 * the cloud is not flagged dense; a dense cloud with NaNs is a caller error there).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  uint32_t idx;
  uint32_t pt;
} vg_item;

static int vg_cmp(const void *a, const void *b) {
  const vg_item *x = (const vg_item *)a, *y = (const vg_item *)b;
  if (x->idx != y->idx) return x->idx < y->idx ? -1 : 1;
  return (x->pt > y->pt) - (x->pt < y->pt); /* stable: ascending input index inside a voxel */
}

/* pts: n points, stride_bytes apart, float x,y,z at byte offsets 0,4,8 and intensity at
 * intensity_offset (16 for pcl::PointXYZI; < 0: no intensity, output 0).
 * out: packed float4 {x, y, z, intensity}; returns the number of output points (may exceed max_out,
 * in which case only the first max_out are written).  *overflow = 1 when step 2 gave up (the output
 * is then the input, non-finite points included, in input order). */
int64_t vgref_filter(const void *pts, int64_t n, int64_t stride_bytes, int32_t intensity_offset, float leaf, float *out,
                     int64_t max_out, int32_t *overflow) {
  const char *base = (const char *)pts;
  if (overflow) *overflow = 0;
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  int64_t nvalid = 0;
  for (int64_t i = 0; i < n; i++) {
    const float *p = (const float *)(base + i * stride_bytes);
    if (!isfinite(p[0]) || !isfinite(p[1]) || !isfinite(p[2])) continue;
    for (int c = 0; c < 3; c++) {
      if (p[c] < mn[c]) mn[c] = p[c];
      if (p[c] > mx[c]) mx[c] = p[c];
    }
    nvalid++;
  }
  if (nvalid == 0) return 0;
  const float inv = 1.0f / leaf;
  int64_t d[3];
  for (int c = 0; c < 3; c++) d[c] = (int64_t)((mx[c] - mn[c]) * inv) + 1;
  if (d[0] * d[1] * d[2] > (int64_t)INT32_MAX) {
    if (overflow) *overflow = 1;
    for (int64_t i = 0; i < n && i < max_out; i++) {
      const float *p = (const float *)(base + i * stride_bytes);
      out[4 * i + 0] = p[0];
      out[4 * i + 1] = p[1];
      out[4 * i + 2] = p[2];
      out[4 * i + 3] = intensity_offset >= 0 ? *(const float *)(base + i * stride_bytes + intensity_offset) : 0.0f;
    }
    return n;
  }
  int32_t min_b[3], div_b[3];
  for (int c = 0; c < 3; c++) {
    min_b[c] = (int32_t)floorf(mn[c] * inv);
    div_b[c] = (int32_t)floorf(mx[c] * inv) - min_b[c] + 1;
  }
  const int32_t mul[3] = {1, div_b[0], div_b[0] * div_b[1]};
  vg_item *items = (vg_item *)malloc(sizeof(vg_item) * (size_t)nvalid);
  int64_t m = 0;
  for (int64_t i = 0; i < n; i++) {
    const float *p = (const float *)(base + i * stride_bytes);
    if (!isfinite(p[0]) || !isfinite(p[1]) || !isfinite(p[2])) continue;
    int32_t idx = 0;
    for (int c = 0; c < 3; c++) idx += (int32_t)(floorf(p[c] * inv) - (float)min_b[c]) * mul[c];
    items[m].idx = (uint32_t)idx;
    items[m].pt = (uint32_t)i;
    m++;
  }
  qsort(items, (size_t)m, sizeof(vg_item), vg_cmp);
  int64_t nout = 0;
  for (int64_t a = 0; a < m;) {
    int64_t b = a;
    float s[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    while (b < m && items[b].idx == items[a].idx) {
      const char *q = base + (int64_t)items[b].pt * stride_bytes;
      const float *p = (const float *)q;
      s[0] += p[0];
      s[1] += p[1];
      s[2] += p[2];
      if (intensity_offset >= 0) s[3] += *(const float *)(q + intensity_offset);
      b++;
    }
    const float cnt = (float)(b - a);
    if (nout < max_out)
      for (int c = 0; c < 4; c++) out[4 * nout + c] = s[c] / cnt;
    nout++;
    a = b;
  }
  free(items);
  return nout;
}
