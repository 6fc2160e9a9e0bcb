/*
 * oracle/ref_kdtree.cpp -- thin extern "C" wrapper around the REFERENCE'S OWN vendored kd-tree.
 * TEST INFRASTRUCTURE ONLY.
 *
 * Nothing is copied: nanoflann.hpp and KDTreeVectorOfVectorsAdaptor.h are #included from where
 * they lie under /root/reference (oracle/Makefile passes -I$(REF)/pgo/SC-A-LOAM/include) and the
 * output goes to oracle/_ref/ (git-ignored, but it travels to the GPU box).  The instantiation
 * is the one the reference uses (Scancontext.h:41-42): KeyMat = vector<vector<float>>,
 * InvKeyTree = KDTreeVectorOfVectorsAdaptor<KeyMat,float>, leaf size 10, SearchParams(10)
 * (Scancontext.cpp:356,373).  It pins the candidate (ring-key k-NN) stage of the oracle.
 */
#include <cstddef>
#include <cstdint>
#include <memory>
#include <vector>

#include "scancontext/nanoflann.hpp"
#include "scancontext/KDTreeVectorOfVectorsAdaptor.h"

using KeyMat = std::vector<std::vector<float>>;
using InvKeyTree = KDTreeVectorOfVectorsAdaptor<KeyMat, float>;

struct ref_kdtree {
  KeyMat keys;
  std::unique_ptr<InvKeyTree> tree;
};

extern "C" {

ref_kdtree *ref_kdtree_build(const float *keys, int64_t n, int dim) {
  auto *t = new ref_kdtree();
  t->keys.assign((size_t)n, std::vector<float>((size_t)dim));
  for (int64_t i = 0; i < n; i++)
    for (int d = 0; d < dim; d++) t->keys[(size_t)i][(size_t)d] = keys[i * dim + d];
  t->tree = std::make_unique<InvKeyTree>((size_t)dim, t->keys, 10 /* max leaf, SC.cpp:356 */);
  return t;
}

/* Scancontext.cpp:367-373: zero-initialised outputs, KNNResultSet, SearchParams(10). */
int ref_kdtree_knn(const ref_kdtree *t, const float *query, int k, uint64_t *out_idx, float *out_dist) {
  std::vector<size_t> candidate_indexes((size_t)k);
  std::vector<float> out_dists_sqr((size_t)k);
  nanoflann::KNNResultSet<float> knnsearch_result((size_t)k);
  knnsearch_result.init(&candidate_indexes[0], &out_dists_sqr[0]);
  t->tree->index->findNeighbors(knnsearch_result, query, nanoflann::SearchParams(10));
  for (int i = 0; i < k; i++) {
    out_idx[i] = candidate_indexes[(size_t)i];
    out_dist[i] = out_dists_sqr[(size_t)i];
  }
  return (int)knnsearch_result.size();
}

void ref_kdtree_free(ref_kdtree *t) { delete t; }
}
