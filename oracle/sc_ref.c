/*
 * oracle/sc_ref.c -- CPU ORACLE for the ScanContext hot path.  TEST INFRASTRUCTURE ONLY.
 * See sc_ref.h for the rules and arithmetic conventions.  Every function cites the reference
 * lines it restates ("SC.cpp" = pgo/SC-A-LOAM/include/scancontext/Scancontext.cpp under
 * /root/reference, "SC.h" = Scancontext.h, "NF.hpp" = nanoflann.hpp).
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp -fPIC -shared (oracle/Makefile).
 */
#include "sc_ref.h"
#include "kdtree_ref.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define NR SCREF_NUM_RING
#define NS SCREF_NUM_SECTOR
#define DS SCREF_DESC_SIZE

/* ------------------------------------------------------------------------------------------ */
/* reductions                                                                                 */
/* ------------------------------------------------------------------------------------------ */

/* Every sum the reference takes through Eigen (mean, norm, dot: SC.cpp:78,81,105,208,224) is a linear
 * redux over a freshly allocated dynamic-size object.  Eigen 3.3 (Redux.h, LinearVectorizedTraversal /
 * NoUnrolling) evaluates it with TWO packet accumulators of P doubles over the first floor(n/2P)*2P
 * terms, adds the accumulators, adds one more packet if floor(n/P) is odd, adds the packet's lanes
 * horizontally and finishes with the scalar tail; without vectorisation it adds in ascending order.
 * P is a property of the BUILD of the reference: 2 for x86-64 as the reference's CMakeLists.txt
 * builds it (-O3, SSE2 baseline), 4 (plus fused multiply-adds) if something injects -march=native.
 * The order is selectable so that each can be checked against the reference's own Scancontext.cpp
 * compiled with the matching stand-in (oracle/ref_sc.cpp, tests/test_oracle_pin.py). */
static int g_sum_order = SCREF_ORDER_EIGEN_SSE2;

void scref_set_sum_order(int order) {
  if (order >= SCREF_ORDER_SEQ && order <= SCREF_ORDER_EIGEN34_AVX_FMA) g_sum_order = order;
}
int scref_get_sum_order(void) { return g_sum_order; }

static inline double madd(double a, double b, double acc, int fused) {
  if (fused) return fma(a, b, acc);
  double p = a * b;
  return acc + p;
}

/* sum_i a[i]*b[i] (b != NULL) or sum_i a[i] (b == NULL) in the selected order */
static double redux(int n, const double *a, const double *b) {
  const int P = g_sum_order == SCREF_ORDER_SEQ ? 1 : (g_sum_order == SCREF_ORDER_EIGEN_SSE2 ? 2 : 4);
  const int fused = g_sum_order == SCREF_ORDER_EIGEN_AVX_FMA || g_sum_order == SCREF_ORDER_EIGEN34_AVX_FMA;
  if (n == 0) return 0.0;
  const int aligned2 = (n / (2 * P)) * (2 * P), aligned = (n / P) * P;
  double res;
  if (P > 1 && aligned) {
    double r0[4], r1[4];
    for (int l = 0; l < P; l++) r0[l] = b ? a[l] * b[l] : a[l];
    if (aligned > P) {
      for (int l = 0; l < P; l++) r1[l] = b ? a[P + l] * b[P + l] : a[P + l];
      for (int i = 2 * P; i < aligned2; i += 2 * P)
        for (int l = 0; l < P; l++) {
          r0[l] = b ? madd(a[i + l], b[i + l], r0[l], fused) : r0[l] + a[i + l];
          r1[l] = b ? madd(a[i + P + l], b[i + P + l], r1[l], fused) : r1[l] + a[i + P + l];
        }
      for (int l = 0; l < P; l++) r0[l] = r0[l] + r1[l];
      if (aligned > aligned2)
        for (int l = 0; l < P; l++)
          r0[l] = b ? madd(a[aligned2 + l], b[aligned2 + l], r0[l], fused) : r0[l] + a[aligned2 + l];
    }
    if (P == 2) res = r0[0] + r0[1];
    else if (g_sum_order == SCREF_ORDER_EIGEN34_AVX_FMA) res = (r0[0] + r0[2]) + (r0[1] + r0[3]); /* predux, Eigen 3.4 (AVX) */
    else res = (r0[0] + r0[1]) + (r0[2] + r0[3]);                                            /* predux, Eigen 3.3 */
    for (int i = aligned; i < n; i++) res = b ? madd(a[i], b[i], res, fused) : res + a[i];
  } else {
    res = b ? a[0] * b[0] : a[0];
    for (int i = 1; i < n; i++) res = b ? madd(a[i], b[i], res, fused) : res + a[i];
  }
  return res;
}

/* ------------------------------------------------------------------------------------------ */
/* helpers                                                                                    */
/* ------------------------------------------------------------------------------------------ */

/* SC.cpp:17-20  float deg2rad(float degrees) { return degrees * M_PI / 180.0; } */
float scref_deg2rad_f(float degrees) { return (float)((double)degrees * M_PI / 180.0); }

/* SC.cpp:23-36.  `_y / _x` is a float division; atan is taken in double (C's ::atan -- the
 * float-overload reading differs only by atanf's own rounding error) and the
 * double expression is narrowed to float by the return type.  x==y==0 gives NaN (0/0) exactly
 * like the reference; NaN inputs (no branch taken = UB in the reference) also return NaN. */
float scref_xy2theta(float x, float y) {
  const double k = 180 / M_PI;
  if ((x >= 0) & (y >= 0)) return (float)(k * atan((double)(y / x)));
  if ((x < 0) & (y >= 0)) return (float)(180 - (k * atan((double)(y / (-x)))));
  if ((x < 0) & (y < 0)) return (float)(180 + (k * atan((double)(y / x))));
  if ((x >= 0) & (y < 0)) return (float)(360 - (k * atan((double)((-y) / x))));
  return NAN;
}

/* SC.cpp:39-59: columns rotated right, new_location = (col + k) % cols. */
void scref_circshift(const double *mat, int rows, int cols, int k, double *out) {
  if (k == 0) {
    memcpy(out, mat, sizeof(double) * (size_t)rows * cols);
    return;
  }
  memset(out, 0, sizeof(double) * (size_t)rows * cols);
  for (int col = 0; col < cols; col++) {
    int nl = (col + k) % cols;
    memcpy(out + (size_t)nl * rows, mat + (size_t)col * rows, sizeof(double) * rows);
  }
}

/* ------------------------------------------------------------------------------------------ */
/* descriptor + keys                                                                          */
/* ------------------------------------------------------------------------------------------ */

/* int(ceil(v)) with the x86 result for NaN (cvttsd2si -> INT_MIN), which the reference hits for
 * the point (0,0) (theta = NaN) and then clamps to 1 (SC.cpp:178-179). */
static int ceil_to_int_x86(double v) {
  double c = ceil(v);
  if (!(c == c)) return (-2147483647 - 1);
  if (c >= 2147483648.0 || c < -2147483648.0) return (-2147483647 - 1);
  return (int)c;
}

/* SC.cpp:151-195 */
void scref_make_scancontext(const float *pts, size_t n, size_t stride_floats, double lidar_height,
                            double max_radius, double *desc) {
  const int NO_POINT = -1000; /* SC.cpp:158 */
  for (int i = 0; i < DS; i++) desc[i] = NO_POINT; /* SC.cpp:159 */

  for (size_t i = 0; i < n; i++) {
    const float *p = pts + i * stride_floats;
    float x = p[0], y = p[1];
    float z = (float)((double)p[2] + lidar_height); /* SC.cpp:168: float = float + double */
    /* non-finite x/y/z: UB / never-stored in the reference; we skip them */
    if (!(x == x) || !(y == y) || !(z == z)) continue;

    float xx = x * x, yy = y * y; /* float products, no contraction */
    float ss = xx + yy;
    float azim_range = (float)sqrt((double)ss);            /* SC.cpp:171 (== sqrtf) */
    float azim_angle = scref_xy2theta(x, y);               /* SC.cpp:172 */

    if ((double)azim_range > max_radius) continue;         /* SC.cpp:175 */

    int ring = ceil_to_int_x86(((double)azim_range / max_radius) * NR);   /* SC.cpp:178 */
    if (ring > NR) ring = NR;
    if (ring < 1) ring = 1;
    int sector = ceil_to_int_x86(((double)azim_angle / 360.0) * NS);      /* SC.cpp:179 */
    if (sector > NS) sector = NS;
    if (sector < 1) sector = 1;

    double *cell = &desc[(sector - 1) * NR + (ring - 1)];
    if (*cell < (double)z) *cell = (double)z;              /* SC.cpp:182-183 */
  }
  for (int i = 0; i < DS; i++)
    if (desc[i] == NO_POINT) desc[i] = 0;                  /* SC.cpp:187-190 */
}

/* SC.cpp:198-211: row-wise mean (sum of 60 / 60). */
void scref_ringkey(const double *desc, double *key20) {
  for (int r = 0; r < NR; r++) {
    double row[NS]; /* SC.cpp:207: MatrixXd curr_row = _desc.row(row_idx) */
    for (int c = 0; c < NS; c++) row[c] = desc[c * NR + r];
    key20[r] = redux(NS, row, NULL) / (double)NS; /* SC.cpp:208 mean() = sum / size */
  }
}

/* SC.cpp:214-227: column-wise mean (sum of 20 / 20). */
void scref_sectorkey(const double *desc, double *key60) {
  for (int c = 0; c < NS; c++) key60[c] = redux(NR, desc + c * NR, NULL) / (double)NR; /* SC.cpp:224 */
}

/* SC.cpp:62-66 (eig2stdvec) applied to the ring key: double -> float narrowing. */
void scref_ringkey_f32(const double *desc, float *key20) {
  double k[NR];
  scref_ringkey(desc, k);
  for (int r = 0; r < NR; r++) key20[r] = (float)k[r];
}

static void col_norms(const double *desc, double *norm60) {
  for (int c = 0; c < NS; c++)
    norm60[c] = sqrt(redux(NR, desc + c * NR, desc + c * NR)); /* Eigen norm() = sqrt(squaredNorm()) */
}

/* ------------------------------------------------------------------------------------------ */
/* pair distance                                                                              */
/* ------------------------------------------------------------------------------------------ */

/* SC.cpp:69-90 */
double scref_dist_direct(const double *sc1, const double *sc2) {
  int num_eff_cols = 0;
  double sum_sector_similarity = 0;
  for (int c = 0; c < NS; c++) {
    const double *a = sc1 + c * NR, *b = sc2 + c * NR;
    double na = sqrt(redux(NR, a, a)), nb = sqrt(redux(NR, b, b));
    if ((na == 0) | (nb == 0)) continue;                   /* SC.cpp:78-79 */
    double dot = redux(NR, a, b);
    double sim = dot / (na * nb);                          /* SC.cpp:81 */
    sum_sector_similarity = sum_sector_similarity + sim;   /* SC.cpp:83 */
    num_eff_cols = num_eff_cols + 1;
  }
  double sc_sim = sum_sector_similarity / num_eff_cols;    /* SC.cpp:87 (0/0 -> NaN) */
  return 1.0 - sc_sim;
}

/* SC.cpp:93-113: first strict minimum over shifts 0..59 of ||vkey1 - circshift(vkey2,k)||. */
int scref_fast_align(const double *vkey1, const double *vkey2) {
  int argmin = 0;
  double minv = 10000000;
  for (int k = 0; k < NS; k++) {
    double diff[NS]; /* SC.cpp:103: MatrixXd vkey_diff = _vkey1 - vkey2_shifted */
    for (int c = 0; c < NS; c++) diff[c] = vkey1[c] - vkey2[(c - k + NS) % NS]; /* shifted[(j+k)%60] = vkey2[j] */
    double nrm = sqrt(redux(NS, diff, diff));
    if (nrm < minv) {
      argmin = k;
      minv = nrm;
    }
  }
  return argmin;
}

static int cmp_int(const void *a, const void *b) { return *(const int *)a - *(const int *)b; }

static int search_space(int argmin_vkey_shift, double search_ratio, int *space) {
  const int radius = (int)round(0.5 * search_ratio * NS); /* SC.cpp:123 */
  int n = 0;
  space[n++] = argmin_vkey_shift;
  for (int ii = 1; ii < radius + 1; ii++) {               /* SC.cpp:125-129 */
    space[n++] = (argmin_vkey_shift + ii + NS) % NS;
    space[n++] = (argmin_vkey_shift - ii + NS) % NS;
  }
  qsort(space, n, sizeof(int), cmp_int);                  /* SC.cpp:130 */
  return n;
}

/* SC.cpp:116-148, literally (with the circshift copies). */
void scref_distance_literal(const double *sc1, const double *sc2, double search_ratio,
                            double *dist, int *shift) {
  double vkey1[NS], vkey2[NS], shifted_key[NS];
  scref_sectorkey(sc1, vkey1);
  scref_sectorkey(sc2, vkey2);
  /* fastAlignUsingVkey via literal circshift of the 1x60 key */
  int argmin_vkey_shift = 0;
  double minv = 10000000;
  for (int k = 0; k < NS; k++) {
    scref_circshift(vkey2, 1, NS, k, shifted_key);
    double diff[NS];
    for (int c = 0; c < NS; c++) diff[c] = vkey1[c] - shifted_key[c];
    double nrm = sqrt(redux(NS, diff, diff));
    if (nrm < minv) {
      argmin_vkey_shift = k;
      minv = nrm;
    }
  }
  int space[NS];
  int n = search_space(argmin_vkey_shift, search_ratio, space);

  int argmin_shift = 0;
  double min_sc_dist = 10000000;
  double *sc2_shifted = (double *)malloc(sizeof(double) * DS);
  for (int i = 0; i < n; i++) {
    scref_circshift(sc2, NR, NS, space[i], sc2_shifted);
    double d = scref_dist_direct(sc1, sc2_shifted);
    if (d < min_sc_dist) {
      argmin_shift = space[i];
      min_sc_dist = d;
    }
  }
  free(sc2_shifted);
  *dist = min_sc_dist;
  *shift = argmin_shift;
}

/* same arithmetic as scref_distance_literal with keys/norms supplied and no copies */
static void distance_pre(const double *sc1, const double *vkey1, const double *n1,
                         const double *sc2, const double *vkey2, const double *n2,
                         double search_ratio, double *dist, int *shift) {
  int k0 = scref_fast_align(vkey1, vkey2);
  int space[NS];
  int n = search_space(k0, search_ratio, space);
  int argmin_shift = 0;
  double min_sc_dist = 10000000;
  for (int i = 0; i < n; i++) {
    int k = space[i];
    int neff = 0;
    double sum = 0;
    for (int c = 0; c < NS; c++) {
      int j = (c - k + NS) % NS; /* column of sc2 that lands on column c after the shift */
      if ((n1[c] == 0) | (n2[j] == 0)) continue;
      const double *a = sc1 + c * NR, *b = sc2 + j * NR;
      double dot = redux(NR, a, b);
      sum = sum + dot / (n1[c] * n2[j]);
      neff++;
    }
    double d = 1.0 - sum / neff;
    if (d < min_sc_dist) {
      argmin_shift = k;
      min_sc_dist = d;
    }
  }
  *dist = min_sc_dist;
  *shift = argmin_shift;
}

void scref_distance(const double *sc1, const double *sc2, double search_ratio, double *dist,
                    int *shift) {
  double v1[NS], v2[NS], n1[NS], n2[NS];
  scref_sectorkey(sc1, v1);
  scref_sectorkey(sc2, v2);
  col_norms(sc1, n1);
  col_norms(sc2, n2);
  distance_pre(sc1, v1, n1, sc2, v2, n2, search_ratio, dist, shift);
}

/* NF.hpp:383-408 (L2_Adaptor::evalMetric, worst_dist<=0 path): float, 4 at a time. */
float scref_ringkey_l2(const float *a, const float *b, int dim) {
  float result = 0;
  int d = 0;
  while (d + 3 < dim) {
    const float diff0 = a[d] - b[d];
    const float diff1 = a[d + 1] - b[d + 1];
    const float diff2 = a[d + 2] - b[d + 2];
    const float diff3 = a[d + 3] - b[d + 3];
    result += diff0 * diff0 + diff1 * diff1 + diff2 * diff2 + diff3 * diff3;
    d += 4;
  }
  while (d < dim) {
    const float diff0 = a[d] - b[d];
    result += diff0 * diff0;
    d++;
  }
  return result;
}

/* ------------------------------------------------------------------------------------------ */
/* manager                                                                                    */
/* ------------------------------------------------------------------------------------------ */

struct scref_mgr {
  /* SC.h:83-104 */
  double lidar_height, max_radius, search_ratio, sc_dist_thres;
  int num_exclude_recent, num_candidates, tree_making_period;
  int tree_making_period_counter; /* SC.h:104 */
  int64_t tree_size;              /* frozen prefix = polarcontext_invkeys_to_search_.size() */
  int batch_tree_made;            /* SC.h:119 */
  int64_t batch_tree_size;
  /* candidate stage: 1 (default) = nanoflann's tree and walk restated (kdtree_ref.c: ties in tree visit order, like
   * the reference), 0 = brute force with the lower index first among equal distances (round 1) */
  int knn_mode;
  kdref *tree[2];                 /* [0] detectLoopClosureID's tree, [1] the between-session tree; rebuilt when stale */
  const float *tree_keys[2];
  /* SC.h:110-115 (SoA) */
  int64_t n, cap;
  double *desc;   /* n x 1200 */
  double *vkey;   /* n x 60 */
  double *norm;   /* n x 60 (not in the reference; shift-invariant, same values) */
  float *rkey;    /* n x 20 */
};

scref_mgr *scref_create(void) {
  scref_mgr *m = (scref_mgr *)calloc(1, sizeof(scref_mgr));
  m->lidar_height = 2.0;      /* SC.h:83 */
  m->max_radius = 80.0;       /* SC.h:87 */
  m->num_exclude_recent = 30; /* SC.h:92 */
  m->num_candidates = 3;      /* SC.h:93 */
  m->search_ratio = 0.1;      /* SC.h:96 */
  m->sc_dist_thres = 0.2;     /* SC.h:99 */
  m->tree_making_period = 30; /* SC.h:103 */
  m->knn_mode = 1;
  return m;
}

void scref_destroy(scref_mgr *m) {
  if (!m) return;
  free(m->desc);
  free(m->vkey);
  free(m->norm);
  free(m->rkey);
  kdref_free(m->tree[0]);
  kdref_free(m->tree[1]);
  free(m);
}

void scref_set_knn_mode(scref_mgr *m, int tree_order) { m->knn_mode = tree_order ? 1 : 0; }

void scref_set_dist_thres(scref_mgr *m, double t) { m->sc_dist_thres = t; }

void scref_set_params(scref_mgr *m, double lidar_height, double max_radius, int num_exclude_recent,
                      int num_candidates, int tree_making_period, double search_ratio) {
  m->lidar_height = lidar_height;
  m->max_radius = max_radius;
  m->num_exclude_recent = num_exclude_recent;
  m->num_candidates = num_candidates;
  m->tree_making_period = tree_making_period;
  m->search_ratio = search_ratio;
}

int64_t scref_size(const scref_mgr *m) { return m->n; }
int64_t scref_tree_size(const scref_mgr *m) { return m->tree_size; }

static void grow(scref_mgr *m) {
  if (m->n < m->cap) return;
  int64_t nc = m->cap ? m->cap * 2 : 1024;
  m->desc = (double *)realloc(m->desc, sizeof(double) * DS * nc);
  m->vkey = (double *)realloc(m->vkey, sizeof(double) * NS * nc);
  m->norm = (double *)realloc(m->norm, sizeof(double) * NS * nc);
  m->rkey = (float *)realloc(m->rkey, sizeof(float) * NR * nc);
  m->cap = nc;
}

int64_t scref_add_descriptor(scref_mgr *m, const double *desc) {
  grow(m);
  int64_t i = m->n;
  memcpy(m->desc + i * DS, desc, sizeof(double) * DS);
  scref_sectorkey(desc, m->vkey + i * NS);
  col_norms(desc, m->norm + i * NS);
  scref_ringkey_f32(desc, m->rkey + i * NR);
  m->n = i + 1;
  return i;
}

int64_t scref_add_points(scref_mgr *m, const float *pts, size_t n, size_t stride_floats) {
  double desc[DS];
  scref_make_scancontext(pts, n, stride_floats, m->lidar_height, m->max_radius, desc);
  return scref_add_descriptor(m, desc);
}

const double *scref_get_descriptor(const scref_mgr *m, int64_t idx) { return m->desc + idx * DS; }
const float *scref_get_ringkey_f32(const scref_mgr *m, int64_t idx) { return m->rkey + idx * NR; }
const double *scref_get_sectorkey(const scref_mgr *m, int64_t idx) { return m->vkey + idx * NS; }

int scref_knn(const scref_mgr *m, const float *query_key, int64_t n_search, int k, int64_t *out_idx,
              float *out_dist) {
  int found = 0;
  for (int i = 0; i < k; i++) {
    out_idx[i] = 0; /* SC.cpp:367: zero-initialised candidate_indexes */
    out_dist[i] = INFINITY;
  }
  for (int64_t i = 0; i < n_search; i++) {
    float d = scref_ringkey_l2(query_key, m->rkey + i * NR, NR);
    /* insert keeping (dist, idx) ascending; entries arrive in ascending idx so strict > keeps
     * the lower index first among equals */
    int pos = found < k ? found : k;
    while (pos > 0 && out_dist[pos - 1] > d) pos--;
    if (pos >= k) continue;
    int last = found < k ? found : k - 1;
    for (int j = last; j > pos; j--) {
      out_dist[j] = out_dist[j - 1];
      out_idx[j] = out_idx[j - 1];
    }
    out_dist[pos] = d;
    out_idx[pos] = i;
    if (found < k) found++;
  }
  return found;
}

/* the candidates of SC.cpp:367-374 / 300-307: the tree over entries [0, n_search) as nanoflann builds it (a function of
 * that prefix: rebuilt when the prefix, or the array behind it, changed) and nanoflann's search of it */
static int mgr_knn(scref_mgr *m, int which, const float *query_key, int64_t n_search, int k, int64_t *out_idx, float *out_dist) {
  if (!m->knn_mode || n_search < 1) return scref_knn(m, query_key, n_search, k, out_idx, out_dist);
  if (!m->tree[which] || kdref_size(m->tree[which]) != n_search || m->tree_keys[which] != m->rkey) {
    kdref_free(m->tree[which]);
    m->tree[which] = kdref_build(m->rkey, n_search);
    m->tree_keys[which] = m->rkey;
  }
  return kdref_knn(m->tree[which], query_key, k, out_idx, out_dist);
}

static int score_candidates(const scref_mgr *m, const double *q_desc, const int64_t *cand, int ncand,
                            float *yaw_diff_rad, double *min_dist_out, int *nn_idx_out) {
  double qv[NS], qn[NS];
  scref_sectorkey(q_desc, qv);
  col_norms(q_desc, qn);
  double min_dist = 10000000; /* SC.cpp:362 */
  int nn_align = 0, nn_idx = 0;
  for (int c = 0; c < ncand; c++) { /* SC.cpp:380-395 */
    int64_t idx = cand[c];
    double d;
    int al;
    distance_pre(q_desc, qv, qn, m->desc + idx * DS, m->vkey + idx * NS, m->norm + idx * NS,
                 m->search_ratio, &d, &al);
    if (d < min_dist) {
      min_dist = d;
      nn_align = al;
      nn_idx = (int)idx;
    }
  }
  int loop_id = -1;
  if (min_dist < m->sc_dist_thres) loop_id = nn_idx; /* SC.cpp:401-403 */
  /* SC.cpp:417: deg2rad(nn_align * PC_UNIT_SECTORANGLE): double product narrowed to the float
   * parameter of the float deg2rad (SC.cpp:17) */
  const double unit = 360.0 / (double)NS; /* SC.h:88 */
  if (yaw_diff_rad) *yaw_diff_rad = scref_deg2rad_f((float)(nn_align * unit));
  if (min_dist_out) *min_dist_out = min_dist;
  if (nn_idx_out) *nn_idx_out = nn_idx;
  return loop_id;
}

/* SC.cpp:331-422 */
int scref_detect_loop_closure(scref_mgr *m, float *yaw_diff_rad, double *min_dist, int *nn_idx) {
  if (m->n == 0 || m->n < m->num_exclude_recent + 1) { /* SC.cpp:341-345 */
    if (yaw_diff_rad) *yaw_diff_rad = 0.0f;
    if (min_dist) *min_dist = 10000000;
    if (nn_idx) *nn_idx = 0;
    return -1;
  }
  const float *curr_key = m->rkey + (m->n - 1) * NR;   /* SC.cpp:335 */
  const double *curr_desc = m->desc + (m->n - 1) * DS; /* SC.cpp:336 */

  if (m->tree_making_period_counter % m->tree_making_period == 0) /* SC.cpp:348-359 */
    m->tree_size = m->n - m->num_exclude_recent;
  m->tree_making_period_counter = m->tree_making_period_counter + 1; /* SC.cpp:360 */

  int64_t cand[64];
  float cd[64];
  int k = m->num_candidates > 64 ? 64 : m->num_candidates;
  mgr_knn(m, 0, curr_key, m->tree_size, k, cand, cd); /* SC.cpp:367-374 */
  return score_candidates(m, curr_desc, cand, k, yaw_diff_rad, min_dist, nn_idx);
}

/* SC.cpp:267-328 */
int scref_detect_between_session(scref_mgr *m, const float *curr_key, const double *curr_desc,
                                 float *yaw_diff_rad, double *min_dist, int *nn_idx) {
  if (!m->batch_tree_made) { /* SC.cpp:275-284 */
    m->batch_tree_size = m->n;
    m->batch_tree_made = 1;
  }
  int64_t cand[64];
  float cd[64];
  int k = m->num_candidates > 64 ? 64 : m->num_candidates;
  mgr_knn(m, 1, curr_key, m->batch_tree_size, k, cand, cd);
  return score_candidates(m, curr_desc, cand, k, yaw_diff_rad, min_dist, nn_idx);
}

static int hit_less(const scref_hit *a, const scref_hit *b) {
  /* total order (dist, index); NaN never produced here (distance_pre returns 1e7 instead) */
  if (a->dist < b->dist) return 1;
  if (a->dist > b->dist) return 0;
  return a->index < b->index;
}

static void topk_insert(scref_hit *list, int k, int *count, const scref_hit *h) {
  int pos = *count < k ? *count : k;
  while (pos > 0 && hit_less(h, &list[pos - 1])) pos--;
  if (pos >= k) return;
  int last = *count < k ? *count : k - 1;
  for (int j = last; j > pos; j--) list[j] = list[j - 1];
  list[pos] = *h;
  if (*count < k) (*count)++;
}

void scref_pair_distances(const scref_mgr *m, const double *query_desc, int64_t first,
                          int64_t count, double *dist, int32_t *shift, int nthreads) {
  double qv[NS], qn[NS];
  scref_sectorkey(query_desc, qv);
  col_norms(query_desc, qn);
  (void)nthreads;
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
  for (int64_t i = 0; i < count; i++) {
    int64_t idx = first + i;
    double d;
    int al;
    distance_pre(query_desc, qv, qn, m->desc + idx * DS, m->vkey + idx * NS, m->norm + idx * NS,
                 m->search_ratio, &d, &al);
    dist[i] = d;
    shift[i] = al;
  }
}

void scref_exhaustive(const scref_mgr *m, const double *query_desc, int64_t n_eligible, int k,
                      scref_hit *out, int nthreads) {
  if (n_eligible > m->n) n_eligible = m->n;
  if (n_eligible < 0) n_eligible = 0;
  double *dist = (double *)malloc(sizeof(double) * (size_t)(n_eligible + 1));
  int32_t *shift = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n_eligible + 1));
  scref_pair_distances(m, query_desc, 0, n_eligible, dist, shift, nthreads);
  int count = 0;
  for (int64_t i = 0; i < n_eligible; i++) {
    scref_hit h = {dist[i], (int32_t)i, shift[i]};
    /* SC.cpp:388: a candidate must pass `candidate_dist < min_dist` against the 1e7 init, so an
     * entry whose distance is 1e7 (no effective column at any shift) is never a hit */
    if (h.dist < 10000000) topk_insert(out, k, &count, &h);
  }
  for (int i = count; i < k; i++) { /* SC.cpp:362-364 initial values */
    out[i].dist = 10000000;
    out[i].index = 0;
    out[i].shift = 0;
  }
  free(dist);
  free(shift);
}

void scref_merge_topk(const scref_hit *parts, int nparts, int k, scref_hit *out) {
  int count = 0;
  for (int p = 0; p < nparts; p++)
    for (int i = 0; i < k; i++) {
      const scref_hit *h = &parts[p * k + i];
      /* padding records {1e7,0,0} from short shards are re-padded below */
      if (!(h->dist < 10000000)) continue;
      topk_insert(out, k, &count, h);
    }
  for (int i = count; i < k; i++) {
    out[i].dist = 10000000;
    out[i].index = 0;
    out[i].shift = 0;
  }
}

/* nq queries, OpenMP over queries (each query scans the DB on one thread): the fair multi-core
 * form of the CPU baseline (bench.py cpu_baseline). */
void scref_exhaustive_batch(const scref_mgr *m, const double *query_descs, int nq, int64_t n_eligible,
                            int k, scref_hit *out, int nthreads) {
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads > 0 ? nthreads : 1)
  for (int q = 0; q < nq; q++)
    scref_exhaustive(m, query_descs + (size_t)q * DS, n_eligible, k, out + (size_t)q * k, 1);
}
