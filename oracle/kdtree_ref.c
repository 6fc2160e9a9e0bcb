/* oracle/kdtree_ref.c -- TEST INFRASTRUCTURE (CPU restatement; never linked into the product).
 *
 * The ring-key search tree of the candidate stage, restated from the reference's vendored nanoflann
 * (pgo/SC-A-LOAM/include/scancontext/nanoflann.hpp, used through KDTreeVectorOfVectorsAdaptor.h:49-117 with
 * ElementType = DistanceType = float, leaf size 10, metric L2_Adaptor): the same tree and the same walk, so that
 * neighbours at EQUAL distance come back in the reference's order (tree visit order), not in index order.
 * Pinned by tests/test_oracle_pin.py against the reference's own nanoflann compiled into oracle/_ref/libref_kdtree.so
 * (indices AND order, on binary ring keys full of ties) and against the reference's SCManager (libref_sc_*.so).
 *
 *   build    buildIndex :1191-1203, computeBoundingBox :1317-1337, divideTree :858-908, middleSplit_ :910-957,
 *            planeSplit :968-1004
 *   search   findNeighbors :1222-1243, computeInitialDistances :1006-1023, searchLevel :1347-1410,
 *            KNNResultSet :142-205, L2_Adaptor::evalMetric :383-408
 */
/*
 * This file restates algorithms of nanoflann (divideTree / middleSplit_ / planeSplit and the kNN walk of
 * KDTreeSingleIndexAdaptor, nanoflann.hpp 1.3.2 as vendored by the reference) closely enough -- the tie order of the
 * neighbours depends on its exact swap sequence -- that it is a derived work.  nanoflann's licence notice:
 *
 * Software License Agreement (BSD License)
 *
 * Copyright 2008-2009  Marius Muja (mariusm@cs.ubc.ca). All rights reserved.
 * Copyright 2008-2009  David G. Lowe (lowe@cs.ubc.ca). All rights reserved.
 * Copyright 2011-2016  Jose Luis Blanco (joseluisblancoc@gmail.com).
 *   All rights reserved.
 *
 * THE BSD LICENSE
 *
 * Redistribution and use in source and binary forms, with or without
 * modification, are permitted provided that the following conditions
 * are met:
 *
 * 1. Redistributions of source code must retain the above copyright
 *    notice, this list of conditions and the following disclaimer.
 * 2. Redistributions in binary form must reproduce the above copyright
 *    notice, this list of conditions and the following disclaimer in the
 *    documentation and/or other materials provided with the distribution.
 *
 * THIS SOFTWARE IS PROVIDED BY THE AUTHOR ``AS IS'' AND ANY EXPRESS OR
 * IMPLIED WARRANTIES, INCLUDING, BUT NOT LIMITED TO, THE IMPLIED WARRANTIES
 * OF MERCHANTABILITY AND FITNESS FOR A PARTICULAR PURPOSE ARE DISCLAIMED.
 * IN NO EVENT SHALL THE AUTHOR BE LIABLE FOR ANY DIRECT, INDIRECT,
 * INCIDENTAL, SPECIAL, EXEMPLARY, OR CONSEQUENTIAL DAMAGES (INCLUDING, BUT
 * NOT LIMITED TO, PROCUREMENT OF SUBSTITUTE GOODS OR SERVICES; LOSS OF USE,
 * DATA, OR PROFITS; OR BUSINESS INTERRUPTION) HOWEVER CAUSED AND ON ANY
 * THEORY OF LIABILITY, WHETHER IN CONTRACT, STRICT LIABILITY, OR TORT
 * (INCLUDING NEGLIGENCE OR OTHERWISE) ARISING IN ANY WAY OUT OF THE USE OF
 * THIS SOFTWARE, EVEN IF ADVISED OF THE POSSIBILITY OF SUCH DAMAGE.
 */
#include "kdtree_ref.h"

#include <float.h>
#include <stdlib.h>
#include <string.h>

#define DIM 20
#define LEAF_MAX 10

typedef struct node {
  struct node *child1, *child2;
  int divfeat;
  float divlow, divhigh;
  int64_t left, right;
} node;

struct kdref {
  const float *keys; /* n x 20, owned by the caller, must outlive the tree */
  int64_t n;
  int64_t *vind;
  node *root;
  float low[DIM], high[DIM];
};

static float pt(const kdref *t, int64_t idx, int d) { return t->keys[idx * DIM + d]; }

static void min_max(const kdref *t, const int64_t *ind, int64_t count, int d, float *mn, float *mx) {
  *mn = *mx = pt(t, ind[0], d);
  for (int64_t i = 1; i < count; i++) {
    float v = pt(t, ind[i], d);
    if (v < *mn) *mn = v;
    if (v > *mx) *mx = v;
  }
}

/* :968-1004.  The reference's indices are size_t: "right &&" guards the decrement at 0 and "!right" ends the pass;
 * mirrored literally with an unsigned type. */
static void plane_split(const kdref *t, int64_t *ind, size_t count, int d, float cutval, size_t *lim1, size_t *lim2) {
  size_t left = 0, right = count - 1;
  for (;;) {
    while (left <= right && pt(t, ind[left], d) < cutval) ++left;
    while (right && left <= right && pt(t, ind[right], d) >= cutval) --right;
    if (left > right || !right) break;
    int64_t tmp = ind[left];
    ind[left] = ind[right];
    ind[right] = tmp;
    ++left;
    --right;
  }
  *lim1 = left;
  right = count - 1;
  for (;;) {
    while (left <= right && pt(t, ind[left], d) <= cutval) ++left;
    while (right && left <= right && pt(t, ind[right], d) > cutval) --right;
    if (left > right || !right) break;
    int64_t tmp = ind[left];
    ind[left] = ind[right];
    ind[right] = tmp;
    ++left;
    --right;
  }
  *lim2 = left;
}

static void middle_split(const kdref *t, int64_t *ind, size_t count, size_t *index, int *cutfeat, float *cutval,
                         const float *blow, const float *bhigh) {
  const float EPS = 0.00001f;
  float max_span = bhigh[0] - blow[0];
  for (int i = 1; i < DIM; i++) {
    float span = bhigh[i] - blow[i];
    if (span > max_span) max_span = span;
  }
  float max_spread = -1;
  *cutfeat = 0;
  for (int i = 0; i < DIM; i++) {
    float span = bhigh[i] - blow[i];
    if (span > (1 - EPS) * max_span) {
      float mn, mx;
      min_max(t, ind, (int64_t)count, i, &mn, &mx);
      float spread = mx - mn;
      if (spread > max_spread) {
        *cutfeat = i;
        max_spread = spread;
      }
    }
  }
  float split_val = (blow[*cutfeat] + bhigh[*cutfeat]) / 2;
  float mn, mx;
  min_max(t, ind, (int64_t)count, *cutfeat, &mn, &mx);
  if (split_val < mn) *cutval = mn;
  else if (split_val > mx) *cutval = mx;
  else *cutval = split_val;
  size_t lim1, lim2;
  plane_split(t, ind, count, *cutfeat, *cutval, &lim1, &lim2);
  if (lim1 > count / 2) *index = lim1;
  else if (lim2 < count / 2) *index = lim2;
  else *index = count / 2;
}

static node *divide(kdref *t, int64_t left, int64_t right, float *blow, float *bhigh) {
  node *nd = (node *)calloc(1, sizeof(node));
  if (right - left <= LEAF_MAX) {
    nd->left = left;
    nd->right = right;
    for (int i = 0; i < DIM; i++) blow[i] = bhigh[i] = pt(t, t->vind[left], i);
    for (int64_t k = left + 1; k < right; k++)
      for (int i = 0; i < DIM; i++) {
        if (blow[i] > pt(t, t->vind[k], i)) blow[i] = pt(t, t->vind[k], i);
        if (bhigh[i] < pt(t, t->vind[k], i)) bhigh[i] = pt(t, t->vind[k], i);
      }
    return nd;
  }
  size_t idx;
  int cutfeat;
  float cutval;
  middle_split(t, t->vind + left, (size_t)(right - left), &idx, &cutfeat, &cutval, blow, bhigh);
  nd->divfeat = cutfeat;
  float llow[DIM], lhigh[DIM], rlow[DIM], rhigh[DIM];
  memcpy(llow, blow, sizeof(llow));
  memcpy(lhigh, bhigh, sizeof(lhigh));
  lhigh[cutfeat] = cutval;
  nd->child1 = divide(t, left, left + (int64_t)idx, llow, lhigh);
  memcpy(rlow, blow, sizeof(rlow));
  memcpy(rhigh, bhigh, sizeof(rhigh));
  rlow[cutfeat] = cutval;
  nd->child2 = divide(t, left + (int64_t)idx, right, rlow, rhigh);
  nd->divlow = lhigh[cutfeat];
  nd->divhigh = rlow[cutfeat];
  for (int i = 0; i < DIM; i++) {
    blow[i] = rlow[i] < llow[i] ? rlow[i] : llow[i];     /* std::min(left, right) */
    bhigh[i] = lhigh[i] < rhigh[i] ? rhigh[i] : lhigh[i]; /* std::max(left, right) */
  }
  return nd;
}

kdref *kdref_build(const float *keys, int64_t n) {
  if (!keys || n < 1) return NULL;
  kdref *t = (kdref *)calloc(1, sizeof(kdref));
  t->keys = keys;
  t->n = n;
  t->vind = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
  for (int64_t i = 0; i < n; i++) t->vind[i] = i;
  for (int i = 0; i < DIM; i++) t->low[i] = t->high[i] = pt(t, 0, i);
  for (int64_t k = 1; k < n; k++)
    for (int i = 0; i < DIM; i++) {
      if (pt(t, k, i) < t->low[i]) t->low[i] = pt(t, k, i);
      if (pt(t, k, i) > t->high[i]) t->high[i] = pt(t, k, i);
    }
  float blow[DIM], bhigh[DIM];
  memcpy(blow, t->low, sizeof(blow));
  memcpy(bhigh, t->high, sizeof(bhigh));
  t->root = divide(t, 0, n, blow, bhigh);
  return t;
}

static void free_nodes(node *nd) {
  if (!nd) return;
  free_nodes(nd->child1);
  free_nodes(nd->child2);
  free(nd);
}

void kdref_free(kdref *t) {
  if (!t) return;
  free_nodes(t->root);
  free(t->vind);
  free(t);
}

int64_t kdref_size(const kdref *t) { return t ? t->n : 0; }

void kdref_vind(const kdref *t, int64_t *out) { memcpy(out, t->vind, sizeof(int64_t) * (size_t)t->n); }

/* ---- search ---- */
typedef struct {
  int64_t *indices;
  float *dists;
  int capacity, count;
} result_set;

static void add_point(result_set *r, float dist, int64_t index) { /* :175-202 */
  int i;
  for (i = r->count; i > 0; --i) {
    if (r->dists[i - 1] > dist) {
      if (i < r->capacity) {
        r->dists[i] = r->dists[i - 1];
        r->indices[i] = r->indices[i - 1];
      }
    } else {
      break;
    }
  }
  if (i < r->capacity) {
    r->dists[i] = dist;
    r->indices[i] = index;
  }
  if (r->count < r->capacity) r->count++;
}

static float eval_metric(const float *a, const float *b) { /* :383-408, size = 20 */
  float result = 0;
  for (int d = 0; d < DIM; d += 4) {
    const float diff0 = a[d] - b[d];
    const float diff1 = a[d + 1] - b[d + 1];
    const float diff2 = a[d + 2] - b[d + 2];
    const float diff3 = a[d + 3] - b[d + 3];
    result += diff0 * diff0 + diff1 * diff1 + diff2 * diff2 + diff3 * diff3;
  }
  return result;
}

static void search_level(const kdref *t, result_set *rs, const float *vec, const node *nd, float mindistsq, float *dists,
                         const float eps_error) {
  if (nd->child1 == NULL && nd->child2 == NULL) {
    float worst_dist = rs->dists[rs->capacity - 1];
    for (int64_t i = nd->left; i < nd->right; ++i) {
      const int64_t index = t->vind[i];
      float dist = eval_metric(vec, t->keys + index * DIM);
      if (dist < worst_dist) add_point(rs, dist, t->vind[i]);
    }
    return;
  }
  int idx = nd->divfeat;
  float val = vec[idx];
  float diff1 = val - nd->divlow;
  float diff2 = val - nd->divhigh;
  const node *best, *other;
  float cut_dist;
  if ((diff1 + diff2) < 0) {
    best = nd->child1;
    other = nd->child2;
    cut_dist = (val - nd->divhigh) * (val - nd->divhigh);
  } else {
    best = nd->child2;
    other = nd->child1;
    cut_dist = (val - nd->divlow) * (val - nd->divlow);
  }
  search_level(t, rs, vec, best, mindistsq, dists, eps_error);
  float dst = dists[idx];
  mindistsq = mindistsq + cut_dist - dst;
  dists[idx] = cut_dist;
  if (mindistsq * eps_error <= rs->dists[rs->capacity - 1]) search_level(t, rs, vec, other, mindistsq, dists, eps_error);
  dists[idx] = dst;
}

int kdref_knn(const kdref *t, const float *query, int k, int64_t *out_idx, float *out_dist) {
  /* the caller's vectors: zero-initialised, Scancontext.cpp:367-368 */
  for (int i = 0; i < k; i++) {
    out_idx[i] = 0;
    out_dist[i] = 0;
  }
  if (!t || k < 1) return 0;
  result_set rs = {out_idx, out_dist, k, 0};
  rs.dists[k - 1] = FLT_MAX; /* KNNResultSet::init :158-164 */
  float dists[DIM];
  float distsq = 0;
  for (int i = 0; i < DIM; i++) { /* :1006-1023 */
    dists[i] = 0;
    if (query[i] < t->low[i]) {
      dists[i] = (query[i] - t->low[i]) * (query[i] - t->low[i]);
      distsq += dists[i];
    }
    if (query[i] > t->high[i]) {
      dists[i] = (query[i] - t->high[i]) * (query[i] - t->high[i]);
      distsq += dists[i];
    }
  }
  const float eps_error = 1 + 0.0f; /* SearchParams(10): eps = 0 (:555-559), epsError = 1 + eps (:1233) */
  search_level(t, &rs, query, t->root, distsq, dists, eps_error);
  return rs.count;
}
