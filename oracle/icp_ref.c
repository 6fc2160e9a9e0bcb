/*
 * oracle/icp_ref.c -- CPU ORACLE for the loop-verification ICP of the reference's PGO node.
 * TEST INFRASTRUCTURE ONLY.
 *
 * PARITY UNPINNED.  The reference runs pcl::IterativeClosestPoint<PointXYZI, PointXYZI> with
 * setMaxCorrespondenceDistance(150), setMaximumIterations(100), setTransformationEpsilon(1e-6),
 * setEuclideanFitnessEpsilon(1e-6), setRANSACIterations(0), then accepts the loop when
 * hasConverged() && getFitnessScore() <= 0.3
 * (pgo/SC-A-LOAM/src/laserPosegraphOptimization.cpp:371-392).  PCL is a third-party dependency,
 * not vendored under /root/reference and not installed here.  This restates the published algorithm
 * of pcl/registration/impl/icp.hpp + default_convergence_criteria.hpp + transformation_estimation_svd.hpp
 * (PCL 1.8-1.10, Scalar = float):
 *   loop:
 *     1. correspondences: for every (transformed) source point its nearest target point; kept when the
 *        squared distance <= max_corr_dist^2 (float); fewer than 3 correspondences -> not converged, stop
 *     2. Umeyama without scaling on the corresponding pairs: means, covariance
 *        (1/n) * sum (dst - mean_dst)(src - mean_src)^T, SVD, R = U diag(1,1,det) V^T, t = mean_dst - R mean_src
 *     3. transform the source cloud by the step, final = step * final, ++iterations
 *     4. DefaultConvergenceCriteria: iterations >= max -> converged (state ITERATIONS);
 *        cos_angle = 0.5 (trace(R_step) - 1) >= 1 - transformation_epsilon and |t_step|^2 <=
 *        transformation_epsilon -> converged (TRANSFORM);  mse = mean squared correspondence distance:
 *        |mse - prev| < 1e-12 -> converged (ABS_MSE);  |mse - prev| / prev < euclidean_fitness_epsilon ->
 *        converged (REL_MSE);  prev = mse
 *   fitness = mean over the source points (transformed by the final transformation) of the squared
 *   distance to the nearest target point.
 * Choices where PCL's arithmetic is not reproducible here: the nearest neighbour is exact (PCL: FLANN
 * kd-tree, also exact) with ties broken by the lower target index; sums run in ascending index order
 * in float; the 3x3 SVD is a cyclic Jacobi eigen-decomposition of H^T H in double (PCL: Eigen
 * JacobiSVD in float).  Poses therefore agree with PCL to float round-off, not bit for bit.
 *
 * sum_order = ICPREF_SUM_TREE restates the ORDER in which the device kernel (csrc/icp.hip, round 5) adds: the moments
 * in double, correspondence i into partial sum i mod 1024 in ascending i, the partial sums l, l + 64, ... one after the other,
 * the 64 results as a balanced tree (neighbours first); means = (float)(sum / n), covariance terms = the float product of the float differences, added in
 * double in the same order, H = (double)(float)sum / n.  Neither order is PCL's (Eigen's vectorised float sums); the +-25
 * submap of the reference is full of nearly tied nearest neighbours, a last-bit difference in a step flips one of them, and
 * two descents that add in different orders settle a centimetre apart.  With the same order the device and this file agree
 * to 1e-4 there too (tests/test_gpu_loopverify.py).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "icp_ref.h"

static void mat4_identity(float *m) {
  memset(m, 0, 16 * sizeof(float));
  m[0] = m[5] = m[10] = m[15] = 1.0f;
}

/* c = a * b (row-major 4x4, float) */
static void mat4_mul(const float *a, const float *b, float *c) {
  float r[16];
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) {
      float s = 0.0f;
      for (int k = 0; k < 4; k++) s += a[4 * i + k] * b[4 * k + j];
      r[4 * i + j] = s;
    }
  memcpy(c, r, sizeof(r));
}

/* R (3x3 row-major, double) from H = sum (dst - md)(src - ms)^T: Umeyama / Kabsch.
 * SVD through the symmetric eigen-decomposition of H^T H (cyclic Jacobi). */
void icpref_rotation_from_covariance(const double H[9], double R[9]) {
  double A[9], V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double s = 0;
      for (int k = 0; k < 3; k++) s += H[3 * k + i] * H[3 * k + j];
      A[3 * i + j] = s; /* A = H^T H */
    }
  for (int sweep = 0; sweep < 60; sweep++) {
    const double off = fabs(A[1]) + fabs(A[2]) + fabs(A[5]);
    if (off <= 1e-22 * (fabs(A[0]) + fabs(A[4]) + fabs(A[8]))) break; /* below 2^-53 of the diagonal a rotation changes nothing */
    for (int p = 0; p < 2; p++)
      for (int q = p + 1; q < 3; q++) {
        const double apq = A[3 * p + q];
        if (fabs(apq) < 1e-300) continue;
        const double theta = (A[3 * q + q] - A[3 * p + p]) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 3; k++) { /* A <- A J */
          const double akp = A[3 * k + p], akq = A[3 * k + q];
          A[3 * k + p] = c * akp - s * akq;
          A[3 * k + q] = s * akp + c * akq;
        }
        for (int k = 0; k < 3; k++) { /* A <- J^T A */
          const double apk = A[3 * p + k], aqk = A[3 * q + k];
          A[3 * p + k] = c * apk - s * aqk;
          A[3 * q + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < 3; k++) { /* V <- V J */
          const double vkp = V[3 * k + p], vkq = V[3 * k + q];
          V[3 * k + p] = c * vkp - s * vkq;
          V[3 * k + q] = s * vkp + c * vkq;
        }
      }
  }
  /* sort eigenvalues descending (columns of V) */
  int idx[3] = {0, 1, 2};
  double ev[3] = {A[0], A[4], A[8]};
  for (int i = 0; i < 2; i++)
    for (int j = i + 1; j < 3; j++)
      if (ev[idx[j]] > ev[idx[i]]) {
        int t = idx[i];
        idx[i] = idx[j];
        idx[j] = t;
      }
  double Vs[9], U[9];
  for (int c = 0; c < 3; c++)
    for (int r = 0; r < 3; r++) Vs[3 * r + c] = V[3 * r + idx[c]];
  /* U columns = H v_i / sigma_i; complete degenerate directions by cross products */
  double sig[3];
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
  if (!(sig[0] > tol)) { /* H == 0: identity */
    for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1.0 : 0.0;
    return;
  }
  if (!(sig[1] > tol)) { /* rank 1: any unit vector orthogonal to u0 */
    double a[3] = {U[0], U[3], U[6]};
    double b[3] = {fabs(a[0]) < 0.9 ? 1.0 : 0.0, fabs(a[0]) < 0.9 ? 0.0 : 1.0, 0.0};
    double d = a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
    for (int r = 0; r < 3; r++) b[r] -= d * a[r];
    d = sqrt(b[0] * b[0] + b[1] * b[1] + b[2] * b[2]);
    U[1] = b[0] / d;
    U[4] = b[1] / d;
    U[7] = b[2] / d;
  }
  /* third columns: right-handed completion; det(U) det(V) decides the reflection like Umeyama's S */
  double u2[3] = {U[3] * U[7] - U[6] * U[4], U[6] * U[1] - U[0] * U[7], U[0] * U[4] - U[3] * U[1]};
  double v2[3] = {Vs[3] * Vs[7] - Vs[6] * Vs[4], Vs[6] * Vs[1] - Vs[0] * Vs[7], Vs[0] * Vs[4] - Vs[3] * Vs[1]};
  /* with u2 = u0 x u1 and v2 = v0 x v1 both bases are right-handed: R = U V^T has det +1, which is
   * exactly U diag(1,1,det(U)det(V)) V^T for the SVD whose third singular vectors are +-u2, +-v2 */
  for (int r = 0; r < 3; r++) {
    U[3 * r + 2] = u2[r];
    Vs[3 * r + 2] = v2[r];
  }
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) R[3 * i + j] = U[3 * i + 0] * Vs[3 * j + 0] + U[3 * i + 1] * Vs[3 * j + 1] + U[3 * i + 2] * Vs[3 * j + 2];
}

static void nearest(const float *p, const float *tgt, int64_t nt, float *d2, int64_t *idx) {
  float best = INFINITY;
  int64_t bi = -1;
  for (int64_t j = 0; j < nt; j++) {
    const float dx = p[0] - tgt[3 * j], dy = p[1] - tgt[3 * j + 1], dz = p[2] - tgt[3 * j + 2];
    const float d = dx * dx + dy * dy + dz * dz;
    if (d < best) {
      best = d;
      bi = j;
    }
  }
  *d2 = best;
  *idx = bi;
}

/* the sum of the 1024 partial sums as the device adds them: lane l the sixteen partial sums l, l + 64, ... one after the
 * other, then the 64 lanes as a balanced tree, neighbours first */
#define TREE_LANES 1024
static double tree_sum(double *p) {
  for (int l = 0; l < 64; l++)
    for (int j = 1; j < TREE_LANES / 64; j++) p[l] += p[l + 64 * j];
  for (int s = 1; s < 64; s <<= 1)
    for (int j = 0; j < 64; j += 2 * s) p[j] += p[j + s];
  return p[0];
}

/* src, tgt: packed xyz float triples.  guess: optional row-major 4x4 (NULL = identity). */
void icpref_align(const float *src, int64_t ns, const float *tgt, int64_t nt, const icpref_params *prm, const float *guess,
                  icpref_result *out) {
  float *cur = (float *)malloc(sizeof(float) * 3 * (size_t)(ns > 0 ? ns : 1));
  float final[16], step[16];
  mat4_identity(final);
  if (guess) memcpy(final, guess, sizeof(final));
  for (int64_t i = 0; i < ns; i++)
    for (int r = 0; r < 3; r++)
      cur[3 * i + r] = final[4 * r + 0] * src[3 * i] + final[4 * r + 1] * src[3 * i + 1] + final[4 * r + 2] * src[3 * i + 2] + final[4 * r + 3];
  const float maxd2 = (float)(prm->max_corr_dist * prm->max_corr_dist);
  double prev_mse = 1.7976931348623157e308; /* std::numeric_limits<double>::max() */
  int iters = 0, converged = 0, state = ICPREF_NOT_CONVERGED;
  int64_t *ci = (int64_t *)malloc(sizeof(int64_t) * (size_t)(ns > 0 ? ns : 1));
  float *cd = (float *)malloc(sizeof(float) * (size_t)(ns > 0 ? ns : 1));
  while (!converged) {
    int64_t cnt = 0;
    for (int64_t i = 0; i < ns; i++) {
      nearest(cur + 3 * i, tgt, nt, &cd[i], &ci[i]);
      if (ci[i] >= 0 && cd[i] <= maxd2) cnt++;
      else ci[i] = -1;
    }
    if (cnt < 3) {
      state = ICPREF_NO_CORRESPONDENCES;
      break;
    }
    float ms[3] = {0, 0, 0}, md[3] = {0, 0, 0};
    double mse = 0.0;
    double H[9], R[9];
    if (prm->sum_order == ICPREF_SUM_TREE) {
      static double part[9][TREE_LANES];
      memset(part, 0, sizeof(part));
      for (int64_t i = 0; i < ns; i++)
        if (ci[i] >= 0) {
          const int l = (int)(i % TREE_LANES);
          part[0][l] += 1.0;
          for (int r = 0; r < 3; r++) {
            part[1 + r][l] += cur[3 * i + r];
            part[4 + r][l] += tgt[3 * ci[i] + r];
          }
          part[7][l] += (double)cd[i];
        }
      double sums[8];
      for (int c = 0; c < 8; c++) sums[c] = tree_sum(part[c]);
      const double n = sums[0];
      for (int r = 0; r < 3; r++) {
        ms[r] = (float)(sums[1 + r] / n);
        md[r] = (float)(sums[4 + r] / n);
      }
      mse = sums[7] / n;
      memset(part, 0, sizeof(part));
      for (int64_t i = 0; i < ns; i++)
        if (ci[i] >= 0) {
          const int l = (int)(i % TREE_LANES);
          for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) {
              const float prod = (tgt[3 * ci[i] + a] - md[a]) * (cur[3 * i + b] - ms[b]);
              part[3 * a + b][l] += (double)prod;
            }
        }
      for (int c = 0; c < 9; c++) H[c] = (double)(float)tree_sum(part[c]) / n;
    } else {
      for (int64_t i = 0; i < ns; i++)
        if (ci[i] >= 0)
          for (int r = 0; r < 3; r++) {
            ms[r] += cur[3 * i + r];
            md[r] += tgt[3 * ci[i] + r];
          }
      for (int r = 0; r < 3; r++) {
        ms[r] /= (float)cnt;
        md[r] /= (float)cnt;
      }
      float Hf[9] = {0};
      for (int64_t i = 0; i < ns; i++)
        if (ci[i] >= 0) {
          for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) Hf[3 * a + b] += (tgt[3 * ci[i] + a] - md[a]) * (cur[3 * i + b] - ms[b]);
          mse += (double)cd[i];
        }
      mse /= (double)cnt;
      for (int i = 0; i < 9; i++) H[i] = (double)Hf[i] / (double)cnt;
    }
    icpref_rotation_from_covariance(H, R);
    mat4_identity(step);
    for (int a = 0; a < 3; a++) {
      for (int b = 0; b < 3; b++) step[4 * a + b] = (float)R[3 * a + b];
      step[4 * a + 3] = (float)((double)md[a] - (R[3 * a] * ms[0] + R[3 * a + 1] * ms[1] + R[3 * a + 2] * ms[2]));
    }
    for (int64_t i = 0; i < ns; i++) {
      const float x = cur[3 * i], y = cur[3 * i + 1], z = cur[3 * i + 2];
      for (int r = 0; r < 3; r++) cur[3 * i + r] = step[4 * r] * x + step[4 * r + 1] * y + step[4 * r + 2] * z + step[4 * r + 3];
    }
    mat4_mul(step, final, final);
    iters++;
    /* DefaultConvergenceCriteria */
    if (iters >= prm->max_iterations) {
      converged = 1;
      state = ICPREF_ITERATIONS;
      break;
    }
    const double cos_angle = 0.5 * ((double)step[0] + (double)step[5] + (double)step[10] - 1.0);
    const double tsq = (double)step[3] * step[3] + (double)step[7] * step[7] + (double)step[11] * step[11];
    if (cos_angle >= 1.0 - prm->transformation_epsilon && tsq <= prm->transformation_epsilon) {
      converged = 1;
      state = ICPREF_TRANSFORM;
      break;
    }
    if (fabs(mse - prev_mse) < 1e-12) {
      converged = 1;
      state = ICPREF_ABS_MSE;
      break;
    }
    if (fabs(mse - prev_mse) / prev_mse < prm->euclidean_fitness_epsilon) {
      converged = 1;
      state = ICPREF_REL_MSE;
      break;
    }
    prev_mse = mse;
  }
  /* getFitnessScore(): mean squared distance of the finally transformed source to its nearest target */
  double fit = 0.0;
  int64_t nr = 0;
  static double fpart[TREE_LANES];
  memset(fpart, 0, sizeof(fpart));
  for (int64_t i = 0; i < ns; i++) {
    float p[3], d2;
    int64_t j;
    for (int r = 0; r < 3; r++)
      p[r] = final[4 * r + 0] * src[3 * i] + final[4 * r + 1] * src[3 * i + 1] + final[4 * r + 2] * src[3 * i + 2] + final[4 * r + 3];
    nearest(p, tgt, nt, &d2, &j);
    if (j >= 0) {
      fit += (double)d2;
      fpart[i % TREE_LANES] += (double)d2;
      nr++;
    }
  }
  if (prm->sum_order == ICPREF_SUM_TREE) fit = tree_sum(fpart);
  memcpy(out->transform, final, sizeof(final));
  out->fitness = nr > 0 ? fit / (double)nr : 1.7976931348623157e308;
  out->iterations = iters;
  out->converged = converged;
  out->state = state;
  free(cur);
  free(ci);
  free(cd);
}
