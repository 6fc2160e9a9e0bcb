/*
 * oracle/orora_ref.c -- CPU ORACLE for ORORA registration.  TEST INFRASTRUCTURE ONLY.
 * PARITY UNPINNED: restates the published algorithm (see orora_ref.h); the reference's ORORA
 * submodule is absent from /root/reference (.gitmodules:1-3, README.md:44-48).
 */
#include "orora_ref.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

void ororaref_default_params(ororaref_params *p) {
  /* recollection of the upstream defaults (flagged unverified) */
  p->tim_noise_bound = 2.0 * 0.75;
  p->noise_bound_radial = 0.3536;
  p->noise_bound_tangential = 1.8 * M_PI / 180.0; /* 2 azimuth steps of 0.9 deg */
  p->gnc_factor = 1.4;
  p->cost_threshold = 1e-6;
  p->max_iterations = 100;
  p->flags = 0; /* ring of K TIMs, per-point normalised TLS cost */
}

typedef struct {
  double v;
  int32_t id; /* +(i+1): lower endpoint x_i - beta_i, -(i+1): upper endpoint x_i + beta_i */
} endpoint;

static int cmp_endpoint(const void *a, const void *b) {
  const endpoint *p = (const endpoint *)a, *q = (const endpoint *)b;
  if (p->v < q->v) return -1;
  if (p->v > q->v) return 1;
  return (p->id > q->id) - (p->id < q->id); /* std::pair order: value, then signed id */
}

/* TEASER++ ScalarTLSEstimator::estimate (adaptive voting), SURVEY B.4.
 * cost_mode 0 (default): truncated least squares normalised per point,
 *     cost(x^) = sum_{k in C} w_k (x_k - x^)^2 + |outliers|,  w_k = beta_k^-2
 *   (a point on its interval edge costs exactly 1 = an outlier); meaningful for per-point bounds such as A-COTE's.
 * cost_mode 1: the form of TEASER++'s registration.cc as recalled (unverifiable here): UNWEIGHTED residuals of the
 *   consensus set [m^2] plus the sum of the bounds of the outliers [m],
 *     cost(x^) = sum_{k in C} (x_k - x^)^2 + sum_{k not in C} beta_k,
 *   with the same weighted-mean estimate x^ = sum w x / sum w. */
double ororaref_scalar_tls_mode(const double *x, const double *beta, int32_t n, int32_t cost_mode, int32_t *n_inliers) {
  if (n <= 0) {
    if (n_inliers) *n_inliers = 0;
    return 0.0;
  }
  endpoint *h = (endpoint *)malloc(sizeof(endpoint) * 2 * (size_t)n);
  double ranges_sum = 0.0;
  for (int32_t i = 0; i < n; i++) {
    h[2 * i].v = x[i] - beta[i];
    h[2 * i].id = i + 1;
    h[2 * i + 1].v = x[i] + beta[i];
    h[2 * i + 1].id = -i - 1;
    ranges_sum += beta[i];
  }
  qsort(h, 2 * (size_t)n, sizeof(endpoint), cmp_endpoint);
  double sw = 0, swx = 0, swxx = 0, sx = 0, sxx = 0, sr = 0;
  int32_t card = 0;
  double best_cost = INFINITY, best_x = 0.0;
  int have = 0;
  for (int32_t i = 0; i < 2 * n; i++) {
    const int32_t idx = abs(h[i].id) - 1;
    const double eps = h[i].id > 0 ? 1.0 : -1.0;
    const double w = 1.0 / (beta[idx] * beta[idx]);
    const double wx = w * x[idx];
    card += h[i].id > 0 ? 1 : -1;
    sw += eps * w;
    swx += eps * wx;
    swxx += eps * (wx * x[idx]);
    sx += eps * x[idx];
    sxx += eps * (x[idx] * x[idx]);
    sr += eps * beta[idx];
    if (card <= 0) continue; /* empty consensus set: no estimate */
    const double x_hat = swx / sw;
    double cost;
    if (cost_mode == 0) {
      const double residual = swxx - 2.0 * swx * x_hat + sw * x_hat * x_hat;
      cost = residual + (double)(n - card);
    } else {
      const double residual = (double)card * x_hat * x_hat + sxx - 2.0 * sx * x_hat;
      cost = residual + (ranges_sum - sr);
    }
    if (!have || cost < best_cost) {
      best_cost = cost;
      best_x = x_hat;
      have = 1;
    }
  }
  free(h);
  if (n_inliers) {
    int32_t c = 0;
    for (int32_t i = 0; i < n; i++) c += fabs(x[i] - best_x) <= beta[i];
    *n_inliers = c;
  }
  return best_x;
}

double ororaref_scalar_tls(const double *x, const double *beta, int32_t n, int32_t *n_inliers) {
  return ororaref_scalar_tls_mode(x, beta, n, 0, n_inliers);
}

static void aniso_bound(double px, double py, double s_r, double s_t, double *bx, double *by) {
  const double rho = sqrt(px * px + py * py);
  double c = 1.0, s = 0.0;
  if (rho > 0.0) {
    c = fabs(px) / rho;
    s = fabs(py) / rho;
  }
  *bx += c * s_r + s * rho * s_t;
  *by += s * s_r + c * rho * s_t;
}

void ororaref_register(const float *src_xy, const float *dst_xy, int32_t k, const ororaref_params *p,
                       ororaref_result *out) {
  memset(out, 0, sizeof(*out));
  if (k < 2) {
    out->status = 1;
    return;
  }
  /* ---- GNC-TLS rotation on the TIM graph: the ring of K TIMs (default) or, with
   * ORORAREF_FLAG_COMPLETE_GRAPH, all K (K-1) / 2 pairs i < j in row-major order ---- */
  const int complete = (p->flags & ORORAREF_FLAG_COMPLETE_GRAPH) != 0;
  const int64_t m = complete ? (int64_t)k * (k - 1) / 2 : k;
  const int64_t mk = m > k ? m : k; /* the arrays are re-used for the K translation residuals below */
  double *ax = (double *)malloc(sizeof(double) * 6 * (size_t)mk);
  double *ay = ax + mk, *bx = ay + mk, *by = bx + mk, *w = by + mk, *r2 = w + mk;
  if (complete) {
    int64_t t = 0;
    for (int32_t i = 0; i < k; i++)
      for (int32_t j = i + 1; j < k; j++, t++) {
        ax[t] = (double)src_xy[2 * j] - (double)src_xy[2 * i];
        ay[t] = (double)src_xy[2 * j + 1] - (double)src_xy[2 * i + 1];
        bx[t] = (double)dst_xy[2 * j] - (double)dst_xy[2 * i];
        by[t] = (double)dst_xy[2 * j + 1] - (double)dst_xy[2 * i + 1];
        w[t] = 1.0;
      }
  } else {
    for (int32_t j = 0; j < k; j++) {
      const int32_t n = (j + 1) % k;
      ax[j] = (double)src_xy[2 * n] - (double)src_xy[2 * j];
      ay[j] = (double)src_xy[2 * n + 1] - (double)src_xy[2 * j + 1];
      bx[j] = (double)dst_xy[2 * n] - (double)dst_xy[2 * j];
      by[j] = (double)dst_xy[2 * n + 1] - (double)dst_xy[2 * j + 1];
      w[j] = 1.0;
    }
  }
  double c2 = p->tim_noise_bound * p->tim_noise_bound;
  if (c2 < 1e-16) c2 = 1e-2;
  double mu = 1.0, prev_cost = INFINITY, cs = 1.0, sn = 0.0;
  int32_t it = 0;
  for (it = 0; it < p->max_iterations; it++) {
    /* weighted 2x2 Kabsch: yaw = atan2(sum w (a x b), sum w (a . b)) */
    double C = 0.0, S = 0.0;
    for (int64_t j = 0; j < m; j++) {
      C += w[j] * (ax[j] * bx[j] + ay[j] * by[j]);
      S += w[j] * (ax[j] * by[j] - ay[j] * bx[j]);
    }
    const double nrm = sqrt(C * C + S * S);
    if (nrm > 0.0) {
      cs = C / nrm;
      sn = S / nrm;
    } else {
      cs = 1.0;
      sn = 0.0;
    }
    double max_r2 = 0.0;
    for (int64_t j = 0; j < m; j++) {
      const double ex = bx[j] - (cs * ax[j] - sn * ay[j]);
      const double ey = by[j] - (sn * ax[j] + cs * ay[j]);
      r2[j] = ex * ex + ey * ey;
      if (r2[j] > max_r2) max_r2 = r2[j];
    }
    if (it == 0) {
      mu = 1.0 / (2.0 * max_r2 / c2 - 1.0);
      if (mu <= 0.0) { /* every residual already inside the bound */
        it = 1;
        break;
      }
    }
    const double th1 = (mu + 1.0) / mu * c2;
    const double th2 = mu / (mu + 1.0) * c2;
    double cost = 0.0;
    for (int64_t j = 0; j < m; j++) {
      cost += w[j] * r2[j];
      if (r2[j] >= th1) w[j] = 0.0;
      else if (r2[j] <= th2) w[j] = 1.0;
      else w[j] = sqrt(c2 * mu * (mu + 1.0) / r2[j]) - mu;
    }
    const double cost_diff = fabs(cost - prev_cost);
    mu = mu * p->gnc_factor;
    prev_cost = cost;
    if (cost_diff < p->cost_threshold) {
      it++;
      break;
    }
  }
  out->iterations = it;
  for (int64_t j = 0; j < m; j++) out->rot_inliers += w[j] >= 0.5;
  out->yaw = atan2(sn, cs);

  /* ---- A-COTE translation ---- */
  double *vx = ax, *vy = ay, *betx = bx, *bety = by; /* reuse */
  for (int32_t i = 0; i < k; i++) {
    const double sx = src_xy[2 * i], sy = src_xy[2 * i + 1];
    const double dx = dst_xy[2 * i], dy = dst_xy[2 * i + 1];
    const double rx = cs * sx - sn * sy, ry = sn * sx + cs * sy;
    vx[i] = dx - rx;
    vy[i] = dy - ry;
    double bxx = 0.0, byy = 0.0;
    aniso_bound(dx, dy, p->noise_bound_radial, p->noise_bound_tangential, &bxx, &byy);
    aniso_bound(rx, ry, p->noise_bound_radial, p->noise_bound_tangential, &bxx, &byy);
    betx[i] = bxx;
    bety[i] = byy;
  }
  const int32_t cost_mode = (p->flags & ORORAREF_FLAG_TEASER_COST) ? 1 : 0;
  out->x = ororaref_scalar_tls_mode(vx, betx, k, cost_mode, NULL);
  out->y = ororaref_scalar_tls_mode(vy, bety, k, cost_mode, NULL);
  for (int32_t i = 0; i < k; i++)
    out->trans_inliers += (fabs(vx[i] - out->x) <= betx[i]) && (fabs(vy[i] - out->y) <= bety[i]);
  free(ax);
}

void ororaref_register_batch(const float *src_xy, const float *dst_xy, const int64_t *offsets,
                             int32_t n_pairs, const ororaref_params *p, ororaref_result *out,
                             int nthreads) {
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads > 0 ? nthreads : 1)
  for (int32_t i = 0; i < n_pairs; i++) {
    const int64_t o = offsets[i];
    ororaref_register(src_xy + 2 * o, dst_xy + 2 * o, (int32_t)(offsets[i + 1] - o), p, out + i);
  }
}
