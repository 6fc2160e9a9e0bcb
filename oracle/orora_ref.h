/*
 * oracle/orora_ref.h -- CPU ORACLE for ORORA scan registration (GNC-TLS rotation + A-COTE
 * translation).  TEST INFRASTRUCTURE ONLY.
 *
 * PARITY UNPINNED.  The reference's ORORA sources are an EMPTY, un-initialised git submodule
 * (outlier-robust-radar-odometry/, url-kaist/outlier-robust-radar-odometry, commit unknown: the
 * mounted tree has no .git) -- /root/reference/.gitmodules:1-3, README.md:19,26-27,44-48.  There
 * is no source, test or golden vector to check against.  This file restates the PUBLISHED
 * algorithm (ORORA, Lim et al. ICRA 2023, arXiv 2303.01876; TEASER++ GNC-TLS rotation and
 * adaptive-voting scalar TLS estimators, Yang et al. T-RO 2020) as recorded in SURVEY.md
 * Appendix B.3/B.4.  Constants are defaults to be parameterised, not parity facts.
 *
 * Planar (SE(2)) problem: given K matched 2-D points src[k] -> dst[k] (after the upstream
 * max-clique pruning), find R(yaw), t with dst ~= R src + t.
 *   1. translation-invariant measurements (TIMs) on the chain graph closed into a ring:
 *        a_j = src[(j+1)%K] - src[j],  b_j = dst[(j+1)%K] - dst[j]        (K TIMs)
 *   2. rotation by graduated non-convexity, truncated least squares:
 *        w = 1; repeat { R = argmin sum w ||b - R a||^2 (2x2 Kabsch = closed form of the SVD);
 *        r2 = ||b - R a||^2; first pass: mu = 1/(2 max r2/c^2 - 1), stop if mu <= 0;
 *        th1 = (mu+1)/mu c^2, th2 = mu/(mu+1) c^2; cost = sum w r2 (old weights);
 *        w = 0 if r2 >= th1, 1 if r2 <= th2, else sqrt(c^2 mu (mu+1)/r2) - mu;
 *        stop if |cost - prev| < cost_threshold; mu *= gnc_factor }   (c = TIM noise bound)
 *   3. translation, component-wise (A-COTE): v_k = dst_k - R src_k; per axis the
 *      scalar TLS estimator over intervals v_k +- beta_k: sort the 2K endpoints by
 *      (value, signed id), sweep the consensus sets C, x^ = weighted mean (weights w = beta^-2),
 *      cost = sum_{C} w (x - x^)^2 + |outliers| (TLS normalised per point), take the minimum.  beta_k is anisotropic: a point at range rho, azimuth
 *      phi has radial bound s_r and tangential bound rho * s_t; projected on the axes and summed
 *      for the two points of the match:
 *        beta_x = sum_{p in {dst_k, R src_k}} |cos phi_p| s_r + |sin phi_p| rho_p s_t   (y: swap)
 */
#ifndef ORORA_REF_H
#define ORORA_REF_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  double tim_noise_bound;        /* c: bound on ||b - R a|| for an inlier TIM (2 x point noise) */
  double noise_bound_radial;     /* s_r  [m]   */
  double noise_bound_tangential; /* s_t  [rad] */
  double gnc_factor;             /* 1.4 in TEASER++ */
  double cost_threshold;         /* stop when |cost - prev_cost| < this */
  int32_t max_iterations;        /* GNC iterations cap */
  int32_t flags;                 /* ORORAREF_FLAG_*: the unpinned modelling choices, switchable */
} ororaref_params;
#define ORORAREF_FLAG_COMPLETE_GRAPH 1 /* TIMs on all K (K-1) / 2 pairs instead of the ring of K */
#define ORORAREF_FLAG_TEASER_COST 2    /* scalar TLS cost in TEASER++'s form (see orora_ref.c) */

typedef struct {
  double x, y, yaw;        /* dst = R(yaw) src + (x,y) */
  int32_t iterations;      /* GNC iterations executed */
  int32_t rot_inliers;     /* TIMs with final weight >= 0.5 */
  int32_t trans_inliers;   /* matches inside both axis intervals at the estimate */
  int32_t status;          /* 0 ok, 1 degenerate (K < 2): identity returned */
} ororaref_result;

void ororaref_default_params(ororaref_params *p);
/* src_xy / dst_xy: K points as float x,y pairs */
void ororaref_register(const float *src_xy, const float *dst_xy, int32_t k, const ororaref_params *p,
                       ororaref_result *out);
/* batch: pair i uses points [offsets[i], offsets[i+1]) of the concatenated arrays */
void ororaref_register_batch(const float *src_xy, const float *dst_xy, const int64_t *offsets,
                             int32_t n_pairs, const ororaref_params *p, ororaref_result *out,
                             int nthreads);
/* scalar TLS estimator on its own (unit-testable): returns the estimate */
double ororaref_scalar_tls(const double *x, const double *beta, int32_t n, int32_t *n_inliers);
double ororaref_scalar_tls_mode(const double *x, const double *beta, int32_t n, int32_t cost_mode, int32_t *n_inliers);

#ifdef __cplusplus
}
#endif
#endif
