// sc_kdtree.hip -- exact k-NN in the ring-key tree, walked the way nanoflann walks it (sc_kdtree.h says why).
//
// One wavefront per query.  The walk is scalar work that every lane performs identically (per-wave state in LDS, no
// synchronisation inside the walk); a leaf's <= 10 distances -- from the parallel brute-force pass, in tree order --
// are read one per lane and offered to the result set in the leaf's order.  Trees of up to ~12 000 keys are staged in
// LDS first (16-byte nodes + distances), larger ones are read through the scalar cache.  What is reproduced (nanoflann.hpp of the reference):
//   findNeighbors            :1222-1243   per-dimension offsets of the query from the root box, their sum
//   searchLevel              :1347-1410   nearer child first ((val - divlow) + (val - divhigh) < 0 -> child1); the other
//                                         child only if mindistsq + cut_dist - dists[idx] <= the k-th best (eps = 0:
//                                         SearchParams(10) sets `checks`, not eps, :555-559); dists[idx] restored after
//   leaf                     :1354-1366   worst_dist is read ONCE per leaf; a point is offered iff dist < that value
//   KNNResultSet::addPoint   :175-202     insertion that shifts strictly larger distances only (ties keep visit order)
//   L2_Adaptor::evalMetric   :383-408     4 differences at a time, left-to-right float sums, no contraction
//
// Pruning with the true k-th distance D (optional; from the parallel brute-force pass over all keys).  nanoflann only
// knows the k-th best SO FAR (>= D at every moment) and walks about a third of a 20-dimensional tree; here a subtree
// whose lower bound exceeds D (1 + 1e-4) is skipped as well.  The result is the same:
//  (i)   Which points at distance <= D end up in the result, and in which order, depends only on the ORDER in which the
//        walk meets the points at distance <= D: such a point is inserted iff fewer than k points at a distance <= its
//        own were met before its leaf was entered (the k-th best is > d iff that count is < k), and equal distances
//        keep their visit order.  Farther points only pass through the result set.
//  (ii)  The depth-first order of the leaves is a property of the tree and the query, not of the pruning, and every
//        leaf that holds a point at distance <= D is visited by nanoflann (its bound is <= D <= the k-th best so far)
//        and by this walk (bound <= D (1 + 1e-4)).
//  (iii) The margin covers float rounding: the bound is accumulated along the path, the distance four dimensions at a
//        time, so a bound can exceed the distance of a point on the cell boundary by an ulp or two (relative 1e-6); a
//        disagreement between the two walks about a subtree in the margin would need a point whose distance is below
//        its own subtree's bound by 1e-4 relative.
// The candidate-guided walk (walk<.., REDUCED = true>, the normal case) goes one step further with the same argument:
// a subtree that holds NO key at distance <= D (1 + 1e-4) cannot change (i) at all, whatever its bound is, so it is
// not entered -- the candidates' tree positions come from the order pass, one per lane, and "does [lo, hi) hold one" is a
// compare and a ballot.  The float bound tests on the way to a candidate's leaf are still performed as nanoflann
// performs them.
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
#include <hip/hip_runtime.h>

#include <mutex>

#include <cfloat>
#include <cmath>
#include <cstdint>

#include "rsx_common.h"
#include "sc_kdtree.h"

namespace rsx {
namespace sc {

namespace {

constexpr int KD_KMAX = 64;
constexpr int KD_CAND_CAP = 64;  // candidates of the reduced walk: one per lane
constexpr int KD_LDS_BUDGET = 150 * 1024;  // nodes + tree-ordered distances are staged in LDS when they fit (~12 k keys)

// One entry of the explicit stack.  node >= 0: the farther child of an inner node, still to be considered once the
// nearer one is done (idx = split dimension, v = cut_dist, mind = the inner node's mindistsq); node < 0: restore
// dists[idx] = v (the line after the second recursive call, nanoflann.hpp:1408).
struct Frame {
  int32_t node;
  int32_t idx;
  float v;
  float mind;
  int32_t lo, hi;  // positions (in vind) the farther child covers
};

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// Every lane runs the same scalar walk on the same values (the per-wave state in LDS is written by all lanes with
// identical data, LDS operations of a wave execute in order): no lane-0 sections, no synchronisation inside the walk --
// with those, a step cost three dependent LDS round trips and the walk of a 10 000-key tree 2 ms.
// q[] and dists[] live in one register each (lane i holds element i; v_readlane / a compare-select with the wave-uniform
// split dimension), the result set and the stack in LDS.
__device__ __forceinline__ float lane_get(float v, int i) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), i));
}
__device__ __forceinline__ float lane_set(float v, int i, float x, int lane) { return lane == i ? x : v; }

// REDUCED: only the subtrees that hold a CANDIDATE -- a key whose distance is at most D (1 + 1e-4), D = the true k-th
// smallest distance; their positions in tree order sit one per lane in `candpos` -- are entered.  The result is
// nanoflann's (file header, (i)-(iii): it depends only on the order in which the walk meets those keys, and that order
// is the depth-first order of their leaves), the bound tests along the way stay nanoflann's float expressions; the
// walk touches (number of candidates) x (depth) nodes instead of a third of the tree.
template <bool RESIDENT, bool REDUCED>
__device__ __forceinline__ void walk(const KdSearchArgs &a, const KdNode16 *nodes_lds, const float *dist_lds, float qreg, float dreg,
                                     Frame *stack, float *rd, int32_t *ri, float bound, float distsq, int candpos, int lane) {
  const int k = a.k;
  int sp = 0, count = 0;
  int cur = 0, clo = 0, chi = (int)a.n;
  float mind = distsq;
  float worst = FLT_MAX;  // mirrors rd[k - 1]
  auto holds_candidate = [&](int lo, int hi) { return !REDUCED || __ballot(candpos >= lo && candpos < hi) != 0ull; };
  for (;;) {
    // ---- descend to a leaf, nearer child first (searchLevel :1371-1391) ----
    for (;;) {
      KdNode16 nd;
      if (RESIDENT) nd = nodes_lds[cur];
      else nd = a.nodes[__builtin_amdgcn_readfirstlane(cur)];
      if (nd.b < 0) {
        // leaf (:1354-1366): worst_dist is read once; the points closer than that are offered in leaf order.  Most
        // leaves offer nothing: one ballot decides (a shuffle per point cost ~120 cycles each)
        const int cnt = -1 - nd.b, left = nd.a;
        float dist = INFINITY;
        if (lane < cnt) dist = RESIDENT ? dist_lds[left + lane] : a.dist_tree[left + lane];
        unsigned long long offer = __ballot(lane < cnt && dist < worst);
        while (offer) {
          const int j = __ffsll((long long)offer) - 1;
          offer &= offer - 1;
          const float dj = lane_get(dist, j);
          const int32_t ij = a.vind[left + j];
          int i;  // KNNResultSet::addPoint :175-202
          for (i = count; i > 0; --i) {
            const float prev = rd[i - 1];
            if (prev > dj) {
              if (i < k) {
                rd[i] = prev;
                ri[i] = ri[i - 1];
              }
            } else {
              break;
            }
          }
          if (i < k) {
            rd[i] = dj;
            ri[i] = ij;
          }
          if (count < k) count++;
        }
        worst = rd[k - 1];
        break;
      }
      const int idx = __builtin_amdgcn_readfirstlane(nd.b & 31), mid = __builtin_amdgcn_readfirstlane(nd.b >> 5);
      const float val = lane_get(qreg, idx);
      const float diff1 = __fsub_rn(val, nd.divlow), diff2 = __fsub_rn(val, nd.divhigh);
      int best, other, blo, bhi, olo, ohi;
      float cut;
      if (__fadd_rn(diff1, diff2) < 0.0f) {
        best = cur + 1; blo = clo; bhi = mid;
        other = nd.a; olo = mid; ohi = chi;
        cut = __fmul_rn(diff2, diff2);  // accum_dist(val, divhigh)
      } else {
        best = nd.a; blo = mid; bhi = chi;
        other = cur + 1; olo = clo; ohi = mid;
        cut = __fmul_rn(diff1, diff1);  // accum_dist(val, divlow)
      }
      stack[sp++] = Frame{other, idx, cut, mind, olo, ohi};
      if (!holds_candidate(blo, bhi)) break;  // nothing in the nearer child can matter: as if it had returned
      cur = best;
      clo = blo;
      chi = bhi;
    }
    // ---- back up (:1397-1409) until a farther child has to be visited ----
    bool again = false;
    while (sp > 0) {
      const Frame f = stack[--sp];
      const int fidx = __builtin_amdgcn_readfirstlane(f.idx);
      if (f.node < 0) {
        dreg = lane_set(dreg, fidx, f.v, lane);
        continue;
      }
      const float dst = lane_get(dreg, fidx);
      const float m2 = __fsub_rn(__fadd_rn(f.mind, f.v), dst);
      dreg = lane_set(dreg, fidx, f.v, lane);
      stack[sp++] = Frame{-1, fidx, dst, 0.0f, 0, 0};
      // mindistsq * epsError <= worstDist(), epsError = 1 + 0
      if (__fmul_rn(m2, 1.0f) <= worst && m2 <= bound && holds_candidate(f.lo, f.hi)) {
        cur = f.node;
        clo = f.lo;
        chi = f.hi;
        mind = m2;
        again = true;
        break;
      }
    }
    if (!again) break;
  }
  if (lane == 0) a.out_found[0] = count;
}

__global__ __launch_bounds__(64) void sc_knn_tree_kernel(KdSearchArgs a, int resident) {
  extern __shared__ __attribute__((aligned(16))) char dyn[];
  __shared__ float q[KD_DIM];
  __shared__ Frame stack[2 * KD_STACK + 2];
  __shared__ float rd[KD_KMAX];
  __shared__ int32_t ri[KD_KMAX];
  const int lane = threadIdx.x;
  const int k = a.k;
  const float bound = (a.bound_dist && a.bound_found && a.bound_found[0] >= k) ? a.bound_dist[k - 1] * 1.0001f + 1e-30f : INFINITY;
  const int ncand = a.cand_count ? a.cand_count[0] : 0;
  const bool reduced = ncand >= 1 && ncand <= KD_CAND_CAP;  // (uniform) else: the full walk
  const int candpos = (reduced && lane < ncand) ? a.cand_pos[lane] : -1;
  KdNode16 *nodes_lds = reinterpret_cast<KdNode16 *>(dyn);
  float *dist_lds = reinterpret_cast<float *>(dyn + (size_t)a.n_nodes * sizeof(KdNode16));
  if (resident && !reduced) {
    const uint4 *src = reinterpret_cast<const uint4 *>(a.nodes);
    uint4 *dst = reinterpret_cast<uint4 *>(nodes_lds);
    for (int i = lane; i < a.n_nodes; i += 64) dst[i] = src[i];
    for (int64_t i = lane; i < a.n; i += 64) dist_lds[i] = a.dist_tree[i];
  }
  if (lane < KD_DIM) q[lane] = a.qkey[lane];
  if (lane < k) {
    rd[lane] = 0.0f;
    ri[lane] = 0;  // Scancontext.cpp:367: the caller's vectors are zero-initialised
  }
  wave_sync();
  rd[k - 1] = FLT_MAX;  // KNNResultSet::init (all lanes, the same value)
  // computeInitialDistances (:1006-1023); `dists` starts at zero (:1235).  Lane i keeps q[i] and dists[i].
  const float qreg = lane < KD_DIM ? q[lane] : 0.0f;
  float dreg = 0.0f, distsq = 0.0f;
  for (int i = 0; i < KD_DIM; i++) {
    const float qi = q[i];
    float di = 0.0f;
    if (qi < a.low[i]) {
      const float d = __fsub_rn(qi, a.low[i]);
      di = __fmul_rn(d, d);
      distsq = __fadd_rn(distsq, di);
    }
    if (qi > a.high[i]) {
      const float d = __fsub_rn(qi, a.high[i]);
      di = __fmul_rn(d, d);
      distsq = __fadd_rn(distsq, di);
    }
    if (lane == i) dreg = di;
  }
  if (reduced) walk<false, true>(a, nodes_lds, dist_lds, qreg, dreg, stack, rd, ri, bound, distsq, candpos, lane);
  else if (resident) walk<true, false>(a, nodes_lds, dist_lds, qreg, dreg, stack, rd, ri, bound, distsq, candpos, lane);
  else walk<false, false>(a, nodes_lds, dist_lds, qreg, dreg, stack, rd, ri, bound, distsq, candpos, lane);
  wave_sync();
  if (lane < k) {
    a.out_idx[lane] = ri[lane];
    a.out_dist[lane] = rd[lane];
  }
}

// distances in tree order: dist_tree[i] = dist_all[vind[i]], so that a leaf reads its <= 10 distances from one line;
// and the candidates of the reduced walk: the tree positions of the keys at distance <= D (1 + 1e-4), in any order
// (every key, when fewer than k keys exist).  cand_count[0] must be 0 on entry; it may exceed KD_CAND_CAP (full walk then)
__global__ __launch_bounds__(256) void sc_knn_tree_order_kernel(const float *__restrict__ dist_all, const int32_t *__restrict__ vind,
                                                                int64_t n, float *__restrict__ dist_tree, int32_t k,
                                                                const float *__restrict__ bound_dist,
                                                                const int32_t *__restrict__ bound_found, int32_t *__restrict__ cand_pos,
                                                                int32_t *__restrict__ cand_count) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float d = dist_all[vind[i]];
  dist_tree[i] = d;
  const float bound = bound_found[0] >= k ? bound_dist[k - 1] * 1.0001f + 1e-30f : INFINITY;
  if (d <= bound) {
    const int slot = atomicAdd(cand_count, 1);
    if (slot < KD_CAND_CAP) cand_pos[slot] = (int32_t)i;
  }
}

}  // namespace

int launch_knn_tree_order(const float *dist_all, const int32_t *vind, int64_t n, float *dist_tree, int32_t k, const float *bound_dist,
                          const int32_t *bound_found, int32_t *cand_pos, int32_t *cand_count, hipStream_t s) {
  if (n <= 0) return RSX_OK;
  RSX_HIP(hipMemsetAsync(cand_count, 0, sizeof(int32_t), s));
  hipLaunchKernelGGL(sc_knn_tree_order_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dist_all, vind, n, dist_tree, k,
                     bound_dist, bound_found, cand_pos, cand_count);
  RSX_HIP(hipGetLastError());
  return RSX_OK;
}

int launch_knn_tree(const KdSearchArgs &a, hipStream_t s) {
  if (a.k < 1 || a.k > KD_KMAX) return fail(RSX_ERR_BAD_ARG, "tree search with k = %d", a.k);
  if (!a.dist_tree) return fail(RSX_ERR_BAD_ARG, "tree search without the distance pass");
  const size_t need = (size_t)a.n_nodes * sizeof(KdNode16) + (size_t)a.n * sizeof(float);
  // the resident form keeps the tree in dynamic LDS: per DEVICE, opt in to what that device can give and fall back to
  // the walk through global memory when the tree (plus the kernel's static LDS) does not fit
  int dev = 0;
  RSX_HIP(hipGetDevice(&dev));
  static std::mutex mu;
  static int lds_limit[64];   // 0 = not asked yet, else usable dynamic LDS bytes of device `dev`
  int limit;
  {
    std::lock_guard<std::mutex> lk(mu);
    if (dev < 0 || dev >= 64) return fail(RSX_ERR_BAD_ARG, "device ordinal %d", dev);
    if (!lds_limit[dev]) {
      int max_block = 0;
      RSX_HIP(hipDeviceGetAttribute(&max_block, hipDeviceAttributeMaxSharedMemoryPerBlock, dev));
      hipFuncAttributes fa;
      RSX_HIP(hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(&sc_knn_tree_kernel)));
      int dyn = max_block - (int)fa.sharedSizeBytes;
      if (dyn > KD_LDS_BUDGET) dyn = KD_LDS_BUDGET;
      if (dyn > 0 && hipFuncSetAttribute(reinterpret_cast<const void *>(&sc_knn_tree_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, dyn) != hipSuccess) {
        (void)hipGetLastError();
        dyn = 0;
      }
      lds_limit[dev] = dyn > 0 ? dyn : -1;
    }
    limit = lds_limit[dev];
  }
  const int resident = (limit > 0 && need <= (size_t)limit) ? 1 : 0;
  hipLaunchKernelGGL(sc_knn_tree_kernel, dim3(1), dim3(64), resident ? need : 0, s, a, resident);
  RSX_HIP(hipGetLastError());
  return RSX_OK;
}

}  // namespace sc
}  // namespace rsx
