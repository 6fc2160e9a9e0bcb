// sc_kdtree.hip -- exact k-NN in the ring-key tree, walked the way nanoflann walks it (sc_kdtree.h says why).
//
// One wavefront per query.  The descent is scalar work (every lane follows the same path; the state lives in LDS and
// is written by lane 0), a leaf's <= 10 points are evaluated one per lane and then offered to the result set in the
// leaf's order.  What is reproduced (nanoflann.hpp of the reference):
//   findNeighbors            :1222-1243   per-dimension offsets of the query from the root box, their sum
//   searchLevel              :1347-1410   nearer child first ((val - divlow) + (val - divhigh) < 0 -> child1); the other
//                                         child only if mindistsq + cut_dist - dists[idx] <= the k-th best (eps = 0:
//                                         SearchParams(10) sets `checks`, not eps, :555-559); dists[idx] restored after
//   leaf                     :1354-1366   worst_dist is read ONCE per leaf; a point is offered iff dist < that value
//   KNNResultSet::addPoint   :175-202     insertion that shifts strictly larger distances only (ties keep visit order)
//   L2_Adaptor::evalMetric   :383-408     4 differences at a time, left-to-right float sums, no contraction
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>
#include <cstdint>

#include "rsx_common.h"
#include "sc_kdtree.h"

namespace rsx {
namespace sc {

namespace {

constexpr int KD_KMAX = 64;

struct Frame {
  int32_t node;
  float mindistsq;
  int32_t state;   // 0: entered, 1: nearer child done, 2: other child done
  int32_t idx;     // split dimension
  int32_t other;   // the farther child
  float cut_dist;
  float saved;     // dists[idx] before the other child was entered
};

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

__global__ __launch_bounds__(64) void sc_knn_tree_kernel(KdSearchArgs a) {
  __shared__ float q[KD_DIM];
  __shared__ float dists[KD_DIM];
  __shared__ Frame stack[KD_STACK + 1];
  __shared__ float rd[KD_KMAX];
  __shared__ int32_t ri[KD_KMAX];
  __shared__ int32_t s_count, s_sp;
  const int lane = threadIdx.x;
  const int k = a.k;
  if (lane < KD_DIM) q[lane] = a.qkey[lane];
  if (lane < k) {
    rd[lane] = 0.0f;
    ri[lane] = 0;  // Scancontext.cpp:367: the caller's vectors are zero-initialised
  }
  wave_sync();
  if (lane == 0) {
    rd[k - 1] = FLT_MAX;  // KNNResultSet::init
    s_count = 0;
    // computeInitialDistances (:1006-1023); `dists` starts at zero (:1235)
    float distsq = 0.0f;
    for (int i = 0; i < KD_DIM; i++) {
      dists[i] = 0.0f;
      if (q[i] < a.low[i]) {
        const float d = __fsub_rn(q[i], a.low[i]);
        dists[i] = __fmul_rn(d, d);
        distsq = __fadd_rn(distsq, dists[i]);
      }
      if (q[i] > a.high[i]) {
        const float d = __fsub_rn(q[i], a.high[i]);
        dists[i] = __fmul_rn(d, d);
        distsq = __fadd_rn(distsq, dists[i]);
      }
    }
    stack[0] = Frame{0, distsq, 0, 0, 0, 0.0f, 0.0f};
    s_sp = 1;
  }
  wave_sync();
  for (;;) {
    const int sp = s_sp;
    if (sp == 0) break;
    const Frame f = stack[sp - 1];
    const KdNode nd = a.nodes[f.node];
    if (nd.child1 < 0) {
      // ---- leaf: distances one point per lane, then the sequential offers ----
      const int cnt = nd.right - nd.left;
      float dist = 0.0f;
      int32_t index = 0;
      if (lane < cnt) {
        index = a.vind[nd.left + lane];
        const float4 *p = reinterpret_cast<const float4 *>(a.keys + (int64_t)index * KD_DIM);
#pragma unroll
        for (int g = 0; g < 5; g++) {
          const float4 v = p[g];
          const float d0 = __fsub_rn(q[4 * g + 0], v.x), d1 = __fsub_rn(q[4 * g + 1], v.y);
          const float d2 = __fsub_rn(q[4 * g + 2], v.z), d3 = __fsub_rn(q[4 * g + 3], v.w);
          const float t = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(d0, d0), __fmul_rn(d1, d1)), __fmul_rn(d2, d2)), __fmul_rn(d3, d3));
          dist = __fadd_rn(dist, t);
        }
      }
      const float worst = rd[k - 1];  // read once per leaf
      for (int j = 0; j < cnt; j++) {
        const float dj = __shfl(dist, j);
        const int32_t ij = __shfl(index, j);
        if (lane == 0 && dj < worst) {
          int count = s_count;
          int i;
          for (i = count; i > 0; --i) {
            if (rd[i - 1] > dj) {
              if (i < k) {
                rd[i] = rd[i - 1];
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
          if (count < k) s_count = count + 1;
        }
      }
      if (lane == 0) s_sp = sp - 1;
      wave_sync();
      continue;
    }
    if (lane == 0) {
      Frame &fr = stack[sp - 1];
      if (f.state == 0) {
        const int idx = nd.divfeat;
        const float val = q[idx];
        const float diff1 = __fsub_rn(val, nd.divlow), diff2 = __fsub_rn(val, nd.divhigh);
        int best, other;
        float cut;
        if (__fadd_rn(diff1, diff2) < 0.0f) {
          best = nd.child1;
          other = nd.child2;
          cut = __fmul_rn(diff2, diff2);  // accum_dist(val, divhigh)
        } else {
          best = nd.child2;
          other = nd.child1;
          cut = __fmul_rn(diff1, diff1);  // accum_dist(val, divlow)
        }
        fr.state = 1;
        fr.idx = idx;
        fr.other = other;
        fr.cut_dist = cut;
        stack[sp] = Frame{best, f.mindistsq, 0, 0, 0, 0.0f, 0.0f};
        s_sp = sp + 1;
      } else if (f.state == 1) {
        const float dst = dists[f.idx];
        const float m2 = __fsub_rn(__fadd_rn(f.mindistsq, f.cut_dist), dst);
        dists[f.idx] = f.cut_dist;
        fr.saved = dst;
        fr.state = 2;
        if (__fmul_rn(m2, 1.0f) <= rd[k - 1]) {  // mindistsq * epsError <= worstDist(), epsError = 1 + 0
          stack[sp] = Frame{f.other, m2, 0, 0, 0, 0.0f, 0.0f};
          s_sp = sp + 1;
        }
      } else {
        dists[f.idx] = f.saved;
        s_sp = sp - 1;
      }
    }
    wave_sync();
  }
  wave_sync();
  if (lane < k) {
    a.out_idx[lane] = ri[lane];
    a.out_dist[lane] = rd[lane];
  }
  if (lane == 0) a.out_found[0] = s_count;
}

}  // namespace

int launch_knn_tree(const KdSearchArgs &a, hipStream_t s) {
  if (a.k < 1 || a.k > KD_KMAX) return fail(RSX_ERR_BAD_ARG, "tree search with k = %d", a.k);
  hipLaunchKernelGGL(sc_knn_tree_kernel, dim3(1), dim3(64), 0, s, a);
  RSX_HIP(hipGetLastError());
  return RSX_OK;
}

}  // namespace sc
}  // namespace rsx
