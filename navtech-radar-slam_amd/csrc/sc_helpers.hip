// sc_helpers.hip -- the reference's public helper methods (Scancontext.h:60-66) as stateless GPU calls on
// arbitrary double descriptors: makeRingkeyFromScancontext / makeSectorkeyFromScancontext (SC.cpp:198-227),
// distDirectSC (SC.cpp:69-90), fastAlignUsingVkey (SC.cpp:93-113), distanceBtnScanContext (SC.cpp:116-148).
//
// These are API-completeness entry points (a caller that used to poke at SCManager's helpers keeps working), not
// part of the batched hot path: one wavefront per call, everything in fp64 on the inputs as given -- no fp32
// storage involved, so any MatrixXd content is accepted.  Arithmetic = the oracle's, bit for bit: Eigen 3.3's
// redux order of the reference build the handle was told (sc_redux_dev.h; default: term i -> accumulator i % 4,
// (a0 + a2) + (a1 + a3), products and sums never fused), the column-similarity sum sequential (SC.cpp:83).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "rsx_common.h"
#include "sc_kernels.h"
#include "sc_redux_dev.h"

namespace rsx {
namespace sc {
namespace {

constexpr double kBig = 10000000.0;

// sum_i a[i*sa] * b[i*sb] (b != nullptr) or sum_i a[i*sa] in the handle's summation order (sc_redux_dev.h)
__device__ __forceinline__ double redux4(int so, const double *a, int sa, const double *b, int sb, int n) {
  return dev::redux_rt(so, n, a, sa, b, sb);
}

// distDirectSC of sc1 against sc2 with its columns rotated right by k (circshift, SC.cpp:39-59); all 64 lanes
// call it, every lane returns the result.  sim / val: 60-element LDS scratch.
__device__ double dist_direct_shifted(int so, const double *sc1, const double *sc2, int k, double *sim, int *val, int lane) {
  __syncthreads();
  if (lane < NS) {
    int j = lane - k;  // column of sc2 that lands on column `lane`
    j += j < 0 ? NS : 0;
    const double *a = sc1 + lane * NR, *b = sc2 + j * NR;
    const double na = sqrt(redux4(so, a, 1, a, 1, NR)), nb = sqrt(redux4(so, b, 1, b, 1, NR));
    const bool skip = (na == 0) | (nb == 0);  // SC.cpp:78 (a NaN norm is not skipped, like the reference)
    val[lane] = skip ? 0 : 1;
    sim[lane] = skip ? 0.0 : redux4(so, a, 1, b, 1, NR) / (na * nb);  // SC.cpp:81
  }
  __syncthreads();
  double s = 0.0;
  int ne = 0;
  for (int c = 0; c < NS; c++) {  // SC.cpp:83-84: sequential, every lane redundantly
    if (val[c]) {
      s = s + sim[c];
      ne = ne + 1;
    }
  }
  return 1.0 - s / (double)ne;  // SC.cpp:87-88 (0/0 -> NaN)
}

// fastAlignUsingVkey (SC.cpp:93-113): first strict minimum over k of ||v1 - circshift(v2, k)||
__device__ int fast_align(int so, const double *v1, const double *v2, double *scratch, int lane) {
  double nrm = INFINITY;
  if (lane < NS) {
    double *d = scratch + lane * NS;  // this lane's 60 differences
    for (int c = 0; c < NS; c++) {
      int j = c - lane;
      j += j < 0 ? NS : 0;
      d[c] = v1[c] - v2[j];
    }
    nrm = sqrt(redux4(so, d, 1, d, 1, NS));
  }
  const bool ok = (lane < NS) && (nrm < kBig);  // SC.cpp:96,104
  double m = ok ? nrm : INFINITY;
  for (int off = 32; off >= 1; off >>= 1) m = fmin(m, __shfl_xor(m, off));
  const unsigned long long bal = __ballot(ok && nrm == m);
  return bal ? (__ffsll((long long)bal) - 1) : 0;
}

struct HelperArgs {
  int so;            // RSX_SC_SUM_*
  int op;            // 0 keys, 1 distDirectSC, 2 fastAlignUsingVkey, 3 distanceBtnScanContext
  const double *a;   // descriptor 1 (1200) or sector key 1 (60)
  const double *b;   // descriptor 2 / sector key 2
  double *out_d;     // op 0: ring key [20] then sector key [60]; op 1/3: distance
  int32_t *out_i;    // op 2/3: shift
};

__global__ __launch_bounds__(64) void sc_helper_kernel(HelperArgs h) {
  __shared__ double s1[DS], s2[DS], sim[NS], v1[NS], v2[NS], dscr[NS * NS];
  __shared__ int val[NS];
  const int lane = threadIdx.x;
  const int na = (h.op == 2) ? NS : DS;
  for (int i = lane; i < na; i += 64) {
    s1[i] = h.a[i];
    if (h.b) s2[i] = h.b[i];
  }
  __syncthreads();
  if (h.op == 0) {
    if (lane < NR) h.out_d[lane] = redux4(h.so, s1 + lane, NR, nullptr, 0, NS) / (double)NS;       // SC.cpp:207-208
    if (lane < NS) h.out_d[NR + lane] = redux4(h.so, s1 + lane * NR, 1, nullptr, 0, NR) / (double)NR;  // SC.cpp:223-224
  } else if (h.op == 1) {
    const double d = dist_direct_shifted(h.so, s1, s2, 0, sim, val, lane);
    if (lane == 0) h.out_d[0] = d;
  } else if (h.op == 2) {
    const int k = fast_align(h.so, s1, s2, dscr, lane);
    if (lane == 0) h.out_i[0] = k;
  } else {
    if (lane < NS) {
      v1[lane] = redux4(h.so, s1 + lane * NR, 1, nullptr, 0, NR) / (double)NR;  // SC.cpp:119-120
      v2[lane] = redux4(h.so, s2 + lane * NR, 1, nullptr, 0, NR) / (double)NR;
    }
    __syncthreads();
    const int ks = fast_align(h.so, v1, v2, dscr, lane);
    // SC.cpp:123-130: {k*, k* +- 1..3} mod 60, evaluated in ascending shift VALUE with strict `<`
    double best = kBig;
    int best_k = 0;
    for (int k = 0; k < NS; k++) {
      int d = k - ks;
      d += d < -NS / 2 ? NS : 0;
      d -= d > NS / 2 ? NS : 0;
      if (d < -3 || d > 3) continue;  // uniform
      const double cur = dist_direct_shifted(h.so, s1, s2, k, sim, val, lane);
      if (cur < best) {
        best = cur;
        best_k = k;
      }
    }
    if (lane == 0) {
      h.out_d[0] = best;
      h.out_i[0] = best_k;
    }
  }
}

}  // namespace

int launch_helper(int op, const double *d_a, const double *d_b, double *d_out_d, int32_t *d_out_i, hipStream_t s, int sum_order) {
  HelperArgs h{sum_order, op, d_a, d_b, d_out_d, d_out_i};
  hipLaunchKernelGGL(sc_helper_kernel, dim3(1), dim3(64), 0, s, h);
  RSX_HIP(hipGetLastError());
  return RSX_OK;
}

}  // namespace sc
}  // namespace rsx
