// sc_kdtree.cpp -- host build of the ring-key search tree, node for node and leaf order for leaf order the tree that
// nanoflann's KDTreeSingleIndexAdaptor builds for the reference (see sc_kdtree.h for why the order matters).
//
// What is reproduced (nanoflann.hpp, the reference's vendored copy):
//   buildIndex      :1191-1203  vind = 0 .. n-1, root box = per-dimension min / max of all points
//   divideTree      :858-908    <= 10 points: leaf with its exact box; otherwise split, recurse left then right with the
//                               PARENT's box cut at the split value, then divlow = the left child's (by then exact)
//                               upper bound, divhigh = the right child's lower bound, box = union of the children's
//   middleSplit_    :910-957    dimension: among those whose box span exceeds (1 - 1e-5) x the widest span, the first
//                               with the strictly largest spread of the points; value: the box centre clamped to the
//                               points' range; position: lim1 if more than half the points are below the value, lim2
//                               if fewer than half are at or below it, else the middle
//   planeSplit      :968-1004   two two-pointer passes (< value | == value | > value); the swaps decide the order of
//                               the points inside the leaves, hence the visit order of tied neighbours
// All arithmetic is float, as there (ElementType = DistanceType = float, KDTreeVectorOfVectorsAdaptor.h:57-60).
//
// This file restates algorithms of nanoflann (divideTree / middleSplit_ / planeSplit and the kNN walk of
// KDTreeSingleIndexAdaptor, nanoflann.hpp 1.3.2 as vendored by the reference) closely enough -- the tie order of the
// neighbours depends on its exact swap sequence -- that it is a derived work.  nanoflann's licence notice:
//
// Software License Agreement (BSD License)
//
// Copyright 2008-2009  Marius Muja (mariusm@cs.ubc.ca). All rights reserved.
// Copyright 2008-2009  David G. Lowe (lowe@cs.ubc.ca). All rights reserved.
// Copyright 2011-2016  Jose Luis Blanco (joseluisblancoc@gmail.com).
//   All rights reserved.
//
// THE BSD LICENSE
//
// Redistribution and use in source and binary forms, with or without
// modification, are permitted provided that the following conditions
// are met:
//
// 1. Redistributions of source code must retain the above copyright
//    notice, this list of conditions and the following disclaimer.
// 2. Redistributions in binary form must reproduce the above copyright
//    notice, this list of conditions and the following disclaimer in the
//    documentation and/or other materials provided with the distribution.
//
// THIS SOFTWARE IS PROVIDED BY THE AUTHOR ``AS IS'' AND ANY EXPRESS OR
// IMPLIED WARRANTIES, INCLUDING, BUT NOT LIMITED TO, THE IMPLIED WARRANTIES
// OF MERCHANTABILITY AND FITNESS FOR A PARTICULAR PURPOSE ARE DISCLAIMED.
// IN NO EVENT SHALL THE AUTHOR BE LIABLE FOR ANY DIRECT, INDIRECT,
// INCIDENTAL, SPECIAL, EXEMPLARY, OR CONSEQUENTIAL DAMAGES (INCLUDING, BUT
// NOT LIMITED TO, PROCUREMENT OF SUBSTITUTE GOODS OR SERVICES; LOSS OF USE,
// DATA, OR PROFITS; OR BUSINESS INTERRUPTION) HOWEVER CAUSED AND ON ANY
// THEORY OF LIABILITY, WHETHER IN CONTRACT, STRICT LIABILITY, OR TORT
// (INCLUDING NEGLIGENCE OR OTHERWISE) ARISING IN ANY WAY OUT OF THE USE OF
// THIS SOFTWARE, EVEN IF ADVISED OF THE POSSIBILITY OF SUCH DAMAGE.
#include "sc_kdtree.h"

#include <utility>

#include "rsx_common.h"

namespace rsx {
namespace sc {

namespace {

struct Box {
  float low[KD_DIM], high[KD_DIM];
};

struct Builder {
  const float *keys;
  KdTreeHost *t;
  int max_depth = 0;

  float at(int32_t point, int dim) const { return keys[(int64_t)point * KD_DIM + dim]; }

  void min_max(const int32_t *ind, int64_t count, int dim, float &mn, float &mx) const {
    mn = mx = at(ind[0], dim);
    for (int64_t i = 1; i < count; i++) {
      const float v = at(ind[i], dim);
      if (v < mn) mn = v;
      if (v > mx) mx = v;
    }
  }

  // nanoflann.hpp:968-1004; `right` never goes below 0 there either (every decrement is guarded or follows a swap at
  // right >= 1), so signed arithmetic walks the same states as the reference's size_t
  void plane_split(int32_t *ind, int64_t count, int dim, float value, int64_t &lim1, int64_t &lim2) const {
    int64_t left = 0, right = count - 1;
    for (;;) {
      while (left <= right && at(ind[left], dim) < value) ++left;
      while (right && left <= right && at(ind[right], dim) >= value) --right;
      if (left > right || !right) break;
      std::swap(ind[left], ind[right]);
      ++left;
      --right;
    }
    lim1 = left;
    right = count - 1;
    for (;;) {
      while (left <= right && at(ind[left], dim) <= value) ++left;
      while (right && left <= right && at(ind[right], dim) > value) --right;
      if (left > right || !right) break;
      std::swap(ind[left], ind[right]);
      ++left;
      --right;
    }
    lim2 = left;
  }

  void middle_split(int32_t *ind, int64_t count, const Box &box, int64_t &index, int &cutfeat, float &cutval) const {
    const float eps = 0.00001f;
    float max_span = box.high[0] - box.low[0];
    for (int i = 1; i < KD_DIM; i++) {
      const float span = box.high[i] - box.low[i];
      if (span > max_span) max_span = span;
    }
    float max_spread = -1.0f;
    cutfeat = 0;
    for (int i = 0; i < KD_DIM; i++) {
      const float span = box.high[i] - box.low[i];
      if (span > (1 - eps) * max_span) {
        float mn, mx;
        min_max(ind, count, i, mn, mx);
        const float spread = mx - mn;
        if (spread > max_spread) {
          cutfeat = i;
          max_spread = spread;
        }
      }
    }
    const float split_val = (box.low[cutfeat] + box.high[cutfeat]) / 2;
    float mn, mx;
    min_max(ind, count, cutfeat, mn, mx);
    if (split_val < mn) cutval = mn;
    else if (split_val > mx) cutval = mx;
    else cutval = split_val;
    int64_t lim1, lim2;
    plane_split(ind, count, cutfeat, cutval, lim1, lim2);
    if (lim1 > count / 2) index = lim1;
    else if (lim2 < count / 2) index = lim2;
    else index = count / 2;
  }

  // returns the node index; box: in = the parent's cut box, out = this subtree's box
  int32_t divide(int64_t left, int64_t right, Box &box, int depth) {
    if (depth > max_depth) max_depth = depth;
    const int32_t me = (int32_t)t->nodes.size();
    t->nodes.push_back(KdNode{-1, -1, 0, 0.0f, 0.0f, 0, 0});
    if (right - left <= KD_LEAF_MAX) {
      t->nodes[me].left = (int32_t)left;
      t->nodes[me].right = (int32_t)right;
      for (int i = 0; i < KD_DIM; i++) box.low[i] = box.high[i] = at(t->vind[left], i);
      for (int64_t k = left + 1; k < right; k++)
        for (int i = 0; i < KD_DIM; i++) {
          const float v = at(t->vind[k], i);
          if (box.low[i] > v) box.low[i] = v;
          if (box.high[i] < v) box.high[i] = v;
        }
      return me;
    }
    int64_t idx;
    int cutfeat;
    float cutval;
    middle_split(t->vind.data() + left, right - left, box, idx, cutfeat, cutval);
    Box lb = box, rb = box;
    lb.high[cutfeat] = cutval;
    const int32_t c1 = divide(left, left + idx, lb, depth + 1);
    rb.low[cutfeat] = cutval;
    const int32_t c2 = divide(left + idx, right, rb, depth + 1);
    KdNode &nd = t->nodes[me];
    nd.child1 = c1;
    nd.child2 = c2;
    nd.divfeat = cutfeat;
    nd.divlow = lb.high[cutfeat];
    nd.divhigh = rb.low[cutfeat];
    for (int i = 0; i < KD_DIM; i++) {
      box.low[i] = rb.low[i] < lb.low[i] ? rb.low[i] : lb.low[i];      // std::min(left, right)
      box.high[i] = lb.high[i] < rb.high[i] ? rb.high[i] : lb.high[i];  // std::max(left, right)
    }
    return me;
  }
};

}  // namespace

int kdtree_build_host(const float *keys, int64_t n, KdTreeHost *out) {
  if (!keys || !out || n < 1 || n > 0x7fffffff) return fail(RSX_ERR_BAD_ARG, "kd-tree over %lld points", (long long)n);
  out->nodes.clear();
  out->nodes.reserve((size_t)(n / 3 + 16));
  out->vind.resize((size_t)n);
  for (int64_t i = 0; i < n; i++) out->vind[(size_t)i] = (int32_t)i;
  Box box;
  for (int i = 0; i < KD_DIM; i++) box.low[i] = box.high[i] = keys[i];  // nanoflann.hpp:1317-1337
  for (int64_t k = 1; k < n; k++)
    for (int i = 0; i < KD_DIM; i++) {
      const float v = keys[k * KD_DIM + i];
      if (v < box.low[i]) box.low[i] = v;
      if (v > box.high[i]) box.high[i] = v;
    }
  for (int i = 0; i < KD_DIM; i++) {
    out->low[i] = box.low[i];
    out->high[i] = box.high[i];
  }
  Builder b{keys, out};
  (void)b.divide(0, n, box, 1);
  out->depth = b.max_depth;
  // the device search keeps one frame per level on an explicit stack
  if (b.max_depth > KD_STACK) return fail(RSX_ERR_RANGE, "ring-key tree of depth %d exceeds the search stack (%d)", b.max_depth, KD_STACK);
  return RSX_OK;
}

}  // namespace sc
}  // namespace rsx
