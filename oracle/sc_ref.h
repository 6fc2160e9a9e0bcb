/*
 * oracle/sc_ref.h -- CPU ORACLE for the ScanContext hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the reference algorithm in
 *   /root/reference/pgo/SC-A-LOAM/include/scancontext/Scancontext.cpp   ("SC.cpp")
 *   /root/reference/pgo/SC-A-LOAM/include/scancontext/Scancontext.h     ("SC.h")
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it, and only as
 * the checker / reported baseline.  The product (librsx.so) never links or calls this file.
 *
 * Parity pinning: the reference has no tests or golden vectors, and
 * SC.cpp itself cannot be compiled here (needs Eigen/PCL/OpenCV).  The oracle is pinned by
 *   (1) the property tests derivable from the reference text (tests/test_oracle_sc.py), and
 *   (2) the reference's own vendored kd-tree (nanoflann.hpp + KDTreeVectorOfVectorsAdaptor.h),
 *       compiled unmodified from /root/reference into oracle/_ref/ (oracle/ref_kdtree.cpp).
 *
 * Arithmetic conventions (every one is a restatement choice where Eigen's internal order is not
 * observable from the reference text):
 *   - all reductions (sum, dot, squaredNorm) are sequential, ascending index, in double;
 *   - no FMA contraction (compiled with -ffp-contract=off);
 *   - atan/sqrt inside xy2theta/makeScancontext are evaluated in double and narrowed to float
 *     where the reference assigns to a float (SC.cpp:26,171).
 *
 * Layout: a descriptor is 20 rings x 60 sectors, column-major double (Eigen default, SC.cpp:159):
 * element (ring r, sector s) at [s*20 + r].
 */
#ifndef SC_REF_H
#define SC_REF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SCREF_NUM_RING 20   /* SC.h:85 */
#define SCREF_NUM_SECTOR 60 /* SC.h:86 */
#define SCREF_DESC_SIZE (SCREF_NUM_RING * SCREF_NUM_SECTOR)

typedef struct {
  double dist;   /* distanceBtnScanContext().first */
  int32_t index; /* global DB index */
  int32_t shift; /* distanceBtnScanContext().second (argmin column shift) */
} scref_hit;

/* ---- helper functions (SC.cpp:12-66) ---- */
float scref_xy2theta(float x, float y);                             /* SC.cpp:23-36 */
float scref_deg2rad_f(float degrees);                               /* SC.cpp:17-20 */
void scref_circshift(const double *mat, int rows, int cols, int k, double *out); /* SC.cpp:39-59 */

/* ---- descriptor + keys (SC.cpp:151-227) ---- */
/* pts: n points, each `stride_floats` floats apart, x,y,z at offsets 0,1,2 (pcl::PointXYZI = 8 floats). */
void scref_make_scancontext(const float *pts, size_t n, size_t stride_floats, double lidar_height,
                            double max_radius, double *desc);       /* SC.cpp:151-195 */
void scref_ringkey(const double *desc, double *key20);              /* SC.cpp:198-211 */
void scref_sectorkey(const double *desc, double *key60);            /* SC.cpp:214-227 */
void scref_ringkey_f32(const double *desc, float *key20);           /* + eig2stdvec, SC.cpp:62-66 */

/* ---- pair distance (SC.cpp:69-148) ---- */
double scref_dist_direct(const double *sc1, const double *sc2);     /* SC.cpp:69-90 */
int scref_fast_align(const double *vkey1, const double *vkey2);     /* SC.cpp:93-113 */
/* literal restatement: materialises circshift copies exactly like the reference */
void scref_distance_literal(const double *sc1, const double *sc2, double search_ratio,
                            double *dist, int *shift);              /* SC.cpp:116-148 */
/* same arithmetic, no copies (used for the timed CPU baseline and big test cases) */
void scref_distance(const double *sc1, const double *sc2, double search_ratio, double *dist,
                    int *shift);

/* nanoflann L2_Adaptor::evalMetric restated (NF.hpp:383-408): float, groups of 4 */
float scref_ringkey_l2(const float *a, const float *b, int dim);

/* ---- database / manager (SC.h:110-120, SC.cpp:236-422) ---- */
typedef struct scref_mgr scref_mgr;

scref_mgr *scref_create(void);
void scref_destroy(scref_mgr *m);
void scref_set_dist_thres(scref_mgr *m, double t);                  /* SC.cpp:262-265 */
void scref_set_params(scref_mgr *m, double lidar_height, double max_radius, int num_exclude_recent,
                      int num_candidates, int tree_making_period, double search_ratio);
int64_t scref_size(const scref_mgr *m);
/* makeAndSaveScancontextAndKeys, SC.cpp:249-260 */
int64_t scref_add_points(scref_mgr *m, const float *pts, size_t n, size_t stride_floats);
/* saveScancontextAndKeys, SC.cpp:236-246 */
int64_t scref_add_descriptor(scref_mgr *m, const double *desc);
const double *scref_get_descriptor(const scref_mgr *m, int64_t idx);
const float *scref_get_ringkey_f32(const scref_mgr *m, int64_t idx);
const double *scref_get_sectorkey(const scref_mgr *m, int64_t idx);

/* brute-force exact k-NN over ring keys [0, n_search) with nanoflann's float L2; ties -> lower
 * index first (nanoflann ties follow tree visit order; see tests).  Unfilled slots keep index 0
 * (SC.cpp:367 zero-initialised vector). Returns number found. */
int scref_knn(const scref_mgr *m, const float *query_key, int64_t n_search, int k, int64_t *out_idx,
              float *out_dist);

/* detectLoopClosureID, SC.cpp:331-422 (candidate mode, reference semantics incl. the stale
 * "tree" prefix rebuilt every tree_making_period calls).  Returns loop id or -1. */
int scref_detect_loop_closure(scref_mgr *m, float *yaw_diff_rad, double *min_dist, int *nn_idx);
/* detectLoopClosureIDBetweenSession, SC.cpp:267-328 */
int scref_detect_between_session(scref_mgr *m, const float *curr_key, const double *curr_desc,
                                 float *yaw_diff_rad, double *min_dist, int *nn_idx);
/* current frozen searchable prefix length (entries [0,n) are in the "tree") */
int64_t scref_tree_size(const scref_mgr *m);

/* exhaustive mode: score entries {first + i*stride : i < count} (local shard
 * view: global index = index_base + i*index_stride) that satisfy global index < n_eligible with
 * scref_distance; return the k best under the total order (dist, index) among entries with
 * dist < 1e7 (SC.cpp:388 strict `<` against the 1e7 init; NaN never wins); unfilled slots are
 * {1e7, 0, 0} like SC.cpp:362-364.  nthreads>1 uses OpenMP over entries. */
void scref_exhaustive(const scref_mgr *m, const double *query_desc, int64_t n_eligible, int k,
                      scref_hit *out, int nthreads);
/* nq queries at once, OpenMP over queries */
void scref_exhaustive_batch(const scref_mgr *m, const double *query_descs, int nq, int64_t n_eligible,
                            int k, scref_hit *out, int nthreads);
/* dist/shift of the query against every entry in [first, first+count) */
void scref_pair_distances(const scref_mgr *m, const double *query_desc, int64_t first,
                          int64_t count, double *dist, int32_t *shift, int nthreads);
/* merge `nparts` per-shard top-k lists (each k long) into the global top-k, order (dist,index) */
void scref_merge_topk(const scref_hit *parts, int nparts, int k, scref_hit *out);

#ifdef __cplusplus
}
#endif
#endif
