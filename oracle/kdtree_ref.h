/* oracle/kdtree_ref.h -- TEST INFRASTRUCTURE: nanoflann's kd-tree restated (see kdtree_ref.c). */
#ifndef RSX_ORACLE_KDTREE_REF_H
#define RSX_ORACLE_KDTREE_REF_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct kdref kdref;
/* tree over keys[0 .. n) (n x 20 floats, row-major; the array must outlive the tree), leaf size 10 */
kdref *kdref_build(const float *keys, int64_t n);
void kdref_free(kdref *t);
int64_t kdref_size(const kdref *t);
/* the permutation of the points as the build leaves it (leaf order = visit order inside a leaf) */
void kdref_vind(const kdref *t, int64_t *out);
/* KNNResultSet semantics: out_idx / out_dist are zeroed first (the caller's zero-initialised vectors,
 * Scancontext.cpp:367-368), neighbours ascending by distance, equal distances in visit order; returns the count */
int kdref_knn(const kdref *t, const float *query, int k, int64_t *out_idx, float *out_dist);
#ifdef __cplusplus
}
#endif
#endif
