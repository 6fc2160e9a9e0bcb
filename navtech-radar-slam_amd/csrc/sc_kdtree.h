// sc_kdtree.h -- the ring-key search tree of the candidate stage (InvKeyTree, Scancontext.h:41-42).
//
// The reference finds its NUM_CANDIDATES_FROM_TREE candidates with nanoflann (leaf size 10, exact search,
// Scancontext.cpp:284,356,367-374).  Ring keys of radar scans take few distinct values (every occupied bin is 2.0), so
// exact ties between ring-key distances are common, and nanoflann returns tied points in the order its depth-first
// search VISITS them (KNNResultSet::addPoint keeps the earlier one, nanoflann.hpp:175-202; a leaf only offers points
// strictly closer than the k-th best known when the leaf was entered, :1354-1366).  Which candidates come back -- and in
// which order they are scored -- therefore depends on the tree: its split dimensions, its split values and the order
// in which planeSplit's swaps leave the points inside a leaf.  To return the reference's candidates bit for bit this
// file builds the SAME tree (host, once per TREE_MAKING_PERIOD detections like the reference, :348-359) and walks it
// the SAME way (device, one wavefront per query).  Round 1 searched by brute force with an index tie rule: identical
// distances, but a different third candidate whenever the tie straddled the k-th place.
#pragma once
#include <cstdint>
#include <vector>

#include <hip/hip_runtime.h>

namespace rsx {
namespace sc {

constexpr int KD_DIM = 20;        // PC_NUM_RING
constexpr int KD_LEAF_MAX = 10;   // Scancontext.cpp:284,356
constexpr int KD_STACK = 192;     // depth of the explicit search stack (a deeper tree fails loudly at build time)

struct KdNode {
  int32_t child1, child2;  // node indices; -1 / -1: leaf
  int32_t divfeat;
  float divlow, divhigh;
  int32_t left, right;     // leaf: range of the permutation vind
};

struct KdTreeHost {
  std::vector<KdNode> nodes;  // node 0 = root (preorder)
  std::vector<int32_t> vind;  // permutation of the points, as planeSplit leaves it
  float low[KD_DIM], high[KD_DIM];  // root bounding box
  int depth = 0;
};

// keys: n x 20 floats, row-major.  Returns RSX_OK or an error status (last_error set).
int kdtree_build_host(const float *keys, int64_t n, KdTreeHost *out);

// what the search kernel reads: 16 bytes per node, child1 = the next node (the build numbers nodes in preorder)
struct KdNode16 {
  int32_t a;     // internal: child2;  leaf: first position in vind
  int32_t b;     // internal: divfeat | (first vind position of child2 << 5), >= 0;  leaf: -1 - number of points
  float divlow, divhigh;
};

struct KdSearchArgs {
  const KdNode16 *nodes;
  int32_t n_nodes;
  int64_t n;           // points in the tree
  const int32_t *vind;
  const float *keys;   // the DB's ring keys (device)
  const float *qkey;   // 20 floats (device)
  float low[KD_DIM], high[KD_DIM];
  int32_t k;
  // optional pruning bound: the true k-th smallest distance (bound_dist[k - 1], valid when bound_found[0] >= k), from
  // the brute-force pass over all keys.  Subtrees that cannot hold a point at or below it are skipped; the result is
  // unchanged (sc_kdtree.hip), the walk touches a handful of leaves instead of a third of the tree
  const float *bound_dist;
  const int32_t *bound_found;
  const float *dist_tree; // optional: that pass's distances in tree order, dist_tree[i] = distance of key vind[i] (the same
                          // float expression as evalMetric)
  const int32_t *cand_pos;    // optional (with bound_*): tree positions of the keys within the bound, from the order pass
  const int32_t *cand_count;  // [1]; 1 .. 64 of them -> the reduced walk (sc_kdtree.hip)
  int32_t *out_idx;    // [k], zero where no neighbour was found (Scancontext.cpp:367: zero-initialised vector)
  float *out_dist;     // [k]
  int32_t *out_found;  // [1]
};
int launch_knn_tree(const KdSearchArgs &a, hipStream_t s);
int launch_knn_tree_order(const float *dist_all, const int32_t *vind, int64_t n, float *dist_tree, int32_t k, const float *bound_dist,
                          const int32_t *bound_found, int32_t *cand_pos, int32_t *cand_count, hipStream_t s);

}  // namespace sc
}  // namespace rsx
