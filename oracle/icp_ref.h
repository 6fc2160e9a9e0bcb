/* oracle/icp_ref.h -- see icp_ref.c.  TEST INFRASTRUCTURE ONLY. */
#ifndef ICP_REF_H
#define ICP_REF_H
#include <stdint.h>

enum { ICPREF_NOT_CONVERGED = 0, ICPREF_ITERATIONS = 1, ICPREF_TRANSFORM = 2, ICPREF_ABS_MSE = 3, ICPREF_REL_MSE = 4,
       ICPREF_NO_CORRESPONDENCES = 5 };

typedef struct {
  double max_corr_dist;             /* setMaxCorrespondenceDistance(150), PGO.cpp:374 */
  double transformation_epsilon;    /* setTransformationEpsilon(1e-6), PGO.cpp:376 */
  double euclidean_fitness_epsilon; /* setEuclideanFitnessEpsilon(1e-6), PGO.cpp:377 */
  int32_t max_iterations;           /* setMaximumIterations(100), PGO.cpp:375 */
  int32_t sum_order;                /* 0: sums in ascending index order in float (the restatement of PCL's float sums);
                                       1: ICPREF_SUM_TREE, the order of the device kernel (icp_ref.c) */
} icpref_params;
enum { ICPREF_SUM_SEQUENTIAL_FLOAT = 0, ICPREF_SUM_TREE = 1 };

typedef struct {
  float transform[16]; /* row-major 4x4: target <- source */
  double fitness;      /* getFitnessScore() */
  int32_t iterations, converged, state, reserved;
} icpref_result;

void icpref_rotation_from_covariance(const double H[9], double R[9]);
void icpref_align(const float *src, int64_t ns, const float *tgt, int64_t nt, const icpref_params *prm, const float *guess,
                  icpref_result *out);
#endif
