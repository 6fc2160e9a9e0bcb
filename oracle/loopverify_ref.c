/*
 * oracle/loopverify_ref.c -- CPU ORACLE for the loop-verification chain behind a ScanContext candidate and for the
 * map assembly.  TEST INFRASTRUCTURE ONLY.
 *
 * Restates, line for line, what pgo/SC-A-LOAM/src/laserPosegraphOptimization.cpp does with the keyframe clouds it
 * keeps (keyframeLaserClouds = the 0.4 m VoxelGrid output of every keyframe, PGO.cpp:482-487):
 *   local2global             PGO.cpp:199-220   every point through pcl::getTransformation(x, y, z, roll, pitch, yaw)
 *   loopFindNearKeyframesCloud PGO.cpp:329-352 keyframes key - size .. key + size, EACH IN ITS OWN LOCAL FRAME, all moved by
 *                                              the ONE pose of the root keyframe ("use same root of loop kf idx",
 *                                              PGO.cpp:340,361), concatenated, VoxelGrid 0.4 m
 *   doICPVirtualRelative     PGO.cpp:355-406   source = current keyframe (submap size 0), target = loop keyframe +- 25,
 *                                              ICP (150 m, 100 iterations, 1e-6, 1e-6), accepted iff converged and
 *                                              fitness <= 0.3, pcl::getTranslationAndEulerAngles, poseFrom.between(poseTo)
 *   pubMap                   PGO.cpp:631-655   every SKIP_FRAMES-th keyframe through ITS OWN pose, concatenated, VoxelGrid
 *
 * PARITY UNPINNED for the PCL pieces (pcl::getTransformation / getTranslationAndEulerAngles: pcl/common/eigen.h of
 * PCL 1.8-1.10, restated from the published source; VoxelGrid and ICP: voxelgrid_ref.c, icp_ref.c) and for
 * gtsam::Pose3::between (GTSAM 4.0: between(p) = inverse() * p).  The control flow -- which clouds, which pose, which
 * gate -- is the reference's own and is what this file pins for the HIP path.
 *
 * pcl::getTransformation(float x, float y, float z, float roll, float pitch, float yaw) (the Affine3f overload the
 * reference calls: its Pose6D doubles are narrowed to float at the call):
 *      A = cos(yaw) B = sin(yaw) C = cos(pitch) D = sin(pitch) E = cos(roll) F = sin(roll) DE = D*E DF = D*F   (float)
 *      | A*C   A*DF - B*E   B*F + A*DE   x |
 *      | B*C   A*E + B*DF   B*DE - A*F   y |
 *      | -D    C*F          C*E          z |
 * pcl::getTranslationAndEulerAngles: x, y, z = t(0..2, 3); roll = atan2(t(2,1), t(2,2)); pitch = asin(-t(2,0));
 * yaw = atan2(t(1,0), t(0,0))  (float).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "icp_ref.h"

int64_t vgref_filter(const void *pts, int64_t n, int64_t stride_bytes, int32_t intensity_offset, float leaf, float *out,
                     int64_t max_out, int32_t *overflow);

typedef struct {
  int32_t accepted;    /* converged && fitness <= threshold (PGO.cpp:385) */
  int32_t converged, iterations, state;
  double fitness;
  float transform[16]; /* icp.getFinalTransformation(), row-major */
  float x, y, z, roll, pitch, yaw; /* pcl::getTranslationAndEulerAngles of it (PGO.cpp:400-403) */
  double relative[16]; /* poseFrom.between(poseTo), poseTo = identity (PGO.cpp:404-407): row-major 4x4 */
  int64_t n_source, n_target; /* points after the VoxelGrid */
} lvref_result;

/* pcl::getTransformation, Affine3f overload; T row-major 4x4 */
void lvref_pose_matrix(const double pose6[6], float T[16]) {
  const float x = (float)pose6[0], y = (float)pose6[1], z = (float)pose6[2];
  const float roll = (float)pose6[3], pitch = (float)pose6[4], yaw = (float)pose6[5];
  const float A = cosf(yaw), B = sinf(yaw), C = cosf(pitch), D = sinf(pitch), E = cosf(roll), F = sinf(roll);
  const float DE = D * E, DF = D * F;
  T[0] = A * C;  T[1] = A * DF - B * E;  T[2] = B * F + A * DE;  T[3] = x;
  T[4] = B * C;  T[5] = A * E + B * DF;  T[6] = B * DE - A * F;  T[7] = y;
  T[8] = -D;     T[9] = C * F;           T[10] = C * E;          T[11] = z;
  T[12] = 0.0f;  T[13] = 0.0f;           T[14] = 0.0f;           T[15] = 1.0f;
}

/* local2global (PGO.cpp:210-217): out = T * p, every product and sum in float, left to right; intensity copied */
static void transform_points(const float *in, int64_t n, const float T[16], float *out) {
  for (int64_t i = 0; i < n; i++) {
    const float px = in[4 * i], py = in[4 * i + 1], pz = in[4 * i + 2];
    out[4 * i + 0] = T[0] * px + T[1] * py + T[2] * pz + T[3];
    out[4 * i + 1] = T[4] * px + T[5] * py + T[6] * pz + T[7];
    out[4 * i + 2] = T[8] * px + T[9] * py + T[10] * pz + T[11];
    out[4 * i + 3] = in[4 * i + 3];
  }
}

/* loopFindNearKeyframesCloud (PGO.cpp:329-352).  clouds: the keyframe clouds back to back as float4 {x, y, z, intensity},
 * keyframe i = points [offsets[i], offsets[i+1]).  out: up to max_out float4; returns the number of points (0: empty) */
int64_t lvref_submap(const float *clouds, const int64_t *offsets, int64_t nkf, int32_t key, int32_t submap_size,
                     const double root_pose[6], float leaf, float *out, int64_t max_out) {
  float T[16];
  lvref_pose_matrix(root_pose, T);
  int64_t lo = (int64_t)key - submap_size, hi = (int64_t)key + submap_size;
  if (lo < 0) lo = 0;
  if (hi > nkf - 1) hi = nkf - 1;
  if (lo > hi) return 0;
  const int64_t n = offsets[hi + 1] - offsets[lo];
  if (n <= 0) return 0; /* nearKeyframes->empty() */
  float *tmp = (float *)malloc((size_t)n * 16);
  transform_points(clouds + 4 * offsets[lo], n, T, tmp);
  int32_t overflow = 0;
  const int64_t m = vgref_filter(tmp, n, 16, 12, leaf, out, max_out, &overflow);
  free(tmp);
  return m;
}

/* pcl::getTranslationAndEulerAngles + gtsam: poseFrom = Pose3(Rot3::RzRyRx(roll, pitch, yaw), Point3(x, y, z));
 * poseFrom.between(identity) = poseFrom.inverse() */
static void euler_and_relative(const float T[16], lvref_result *r) {
  r->x = T[3];
  r->y = T[7];
  r->z = T[11];
  r->roll = atan2f(T[9], T[10]);
  r->pitch = asinf(-T[8]);
  r->yaw = atan2f(T[4], T[0]);
  /* Rot3::RzRyRx(x = roll, y = pitch, z = yaw) in double: R = Rz(yaw) Ry(pitch) Rx(roll) */
  const double cx = cos((double)r->roll), sx = sin((double)r->roll), cy = cos((double)r->pitch), sy = sin((double)r->pitch);
  const double cz = cos((double)r->yaw), sz = sin((double)r->yaw);
  const double R[9] = {cz * cy, cz * sy * sx - sz * cx, cz * sy * cx + sz * sx,
                       sz * cy, sz * sy * sx + cz * cx, sz * sy * cx - cz * sx,
                       -sy,     cy * sx,                cy * cx};
  const double t[3] = {(double)r->x, (double)r->y, (double)r->z};
  /* inverse: [R^T | -R^T t] */
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) r->relative[4 * i + j] = R[3 * j + i];
    r->relative[4 * i + 3] = -(R[0 * 3 + i] * t[0] + R[1 * 3 + i] * t[1] + R[2 * 3 + i] * t[2]);
  }
  r->relative[12] = r->relative[13] = r->relative[14] = 0.0;
  r->relative[15] = 1.0;
}

/* doICPVirtualRelative (PGO.cpp:355-406).  root_pose = keyframePosesUpdated[loop_idx] */
void lvref_verify(const float *clouds, const int64_t *offsets, int64_t nkf, int32_t loop_idx, int32_t curr_idx,
                  const double root_pose[6], int32_t history_num, float leaf, const icpref_params *icp, double fitness_threshold,
                  lvref_result *out) {
  memset(out, 0, sizeof(*out));
  int64_t cap_s = 1, cap_t = 1;
  if (curr_idx >= 0 && curr_idx < nkf) cap_s += offsets[curr_idx + 1] - offsets[curr_idx];
  {
    int64_t lo = (int64_t)loop_idx - history_num, hi = (int64_t)loop_idx + history_num;
    if (lo < 0) lo = 0;
    if (hi > nkf - 1) hi = nkf - 1;
    if (lo <= hi) cap_t += offsets[hi + 1] - offsets[lo];
  }
  float *src = (float *)malloc((size_t)cap_s * 16), *tgt = (float *)malloc((size_t)cap_t * 16);
  const int64_t ns = lvref_submap(clouds, offsets, nkf, curr_idx, 0, root_pose, leaf, src, cap_s);           /* PGO.cpp:361 */
  const int64_t nt = lvref_submap(clouds, offsets, nkf, loop_idx, history_num, root_pose, leaf, tgt, cap_t); /* PGO.cpp:362 */
  out->n_source = ns;
  out->n_target = nt;
  /* icp_ref takes packed xyz */
  float *s3 = (float *)malloc((size_t)(ns + 1) * 12), *t3 = (float *)malloc((size_t)(nt + 1) * 12);
  for (int64_t i = 0; i < ns; i++) memcpy(s3 + 3 * i, src + 4 * i, 12);
  for (int64_t i = 0; i < nt; i++) memcpy(t3 + 3 * i, tgt + 4 * i, 12);
  icpref_result ir;
  icpref_align(s3, ns, t3, nt, icp, NULL, &ir);
  free(src);
  free(tgt);
  free(s3);
  free(t3);
  out->converged = ir.converged;
  out->iterations = ir.iterations;
  out->state = ir.state;
  out->fitness = ir.fitness;
  memcpy(out->transform, ir.transform, sizeof(out->transform));
  out->accepted = !(ir.converged == 0 || ir.fitness > fitness_threshold); /* PGO.cpp:385 */
  euler_and_relative(ir.transform, out);
}

/* pubMap (PGO.cpp:631-655): keyframes 0, skip, 2 skip, ... each through its own pose (poses: nkf x 6 doubles), VoxelGrid */
int64_t lvref_map(const float *clouds, const int64_t *offsets, int64_t nkf, const double *poses, int32_t skip, float leaf,
                  float *out, int64_t max_out) {
  if (skip < 1) skip = 1;
  int64_t n = 0;
  for (int64_t k = 0; k < nkf; k += skip) n += offsets[k + 1] - offsets[k];
  if (n <= 0) return 0;
  float *tmp = (float *)malloc((size_t)n * 16);
  int64_t w = 0;
  for (int64_t k = 0; k < nkf; k += skip) {
    float T[16];
    lvref_pose_matrix(poses + 6 * k, T);
    const int64_t cnt = offsets[k + 1] - offsets[k];
    transform_points(clouds + 4 * offsets[k], cnt, T, tmp + 4 * w);
    w += cnt;
  }
  int32_t overflow = 0;
  const int64_t m = vgref_filter(tmp, n, 16, 12, leaf, out, max_out, &overflow);
  free(tmp);
  return m;
}
