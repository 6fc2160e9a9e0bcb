// sc_redux_dev.h -- the summation order of the sums the reference takes through Eigen (mean, norm, dot: Scancontext.cpp:78,81,
// 105,208,224) as a compile-time property of the kernels.
//
// Every such sum is a linear redux over a freshly allocated dynamic-size object, which Eigen 3.3 evaluates with two packet
// accumulators of P doubles (Redux.h, LinearVectorizedTraversal / NoUnrolling): term i goes to accumulator i mod 2P, the
// accumulators are added, one more packet follows if floor(n / P) is odd, then the packet's lanes are added horizontally and
// the scalar tail is appended.  P belongs to the BUILD of the reference: 2 for x86-64 as its CMakeLists.txt:5-7 writes it
// (-O3, SSE2 baseline) -- the default here --, 1 without vectorisation, 4 with fused multiply-adds when something injects
// -march=native into the catkin workspace (GTSAM's exported flags can).  For binary radar descriptors the choice moves the
// alignment shift on ~1.6 % of the pairs (tests/test_oracle_pin.py::test_what_the_summation_order_can_change), so it is a
// parameter of the handle (rsx_sc_params.sum_order), oracle/sc_ref.c's `redux` restated for compile-time n and P.
#pragma once
#include <hip/hip_runtime.h>

#include "rsx.h"

namespace rsx {
namespace sc {
namespace dev {

constexpr int SO_SSE2 = RSX_SC_SUM_EIGEN_SSE2, SO_SEQ = RSX_SC_SUM_SEQ, SO_AVX_FMA = RSX_SC_SUM_EIGEN_AVX_FMA, SO_AVX34_FMA = RSX_SC_SUM_EIGEN34_AVX_FMA;

// sum over i < N of a(i) * b(i) (PROD) or of a(i); a, b: callables int -> double
template <int SO, int N, bool PROD, typename A, typename B>
__device__ __forceinline__ double redux_impl(A a, B b) {
  constexpr int P = SO == SO_SEQ ? 1 : (SO == SO_SSE2 ? 2 : 4);
  constexpr bool fused = SO == SO_AVX_FMA || SO == SO_AVX34_FMA;
  auto first = [&](int i) -> double {
    if constexpr (PROD) return a(i) * b(i);
    else return a(i);
  };
  auto acc = [&](double r, int i) -> double {
    if constexpr (PROD) {
      if constexpr (fused) return fma(a(i), b(i), r);
      const double p = a(i) * b(i);  // (-ffp-contract=off: the product is rounded before the addition)
      return r + p;
    } else {
      return r + a(i);
    }
  };
  if constexpr (P == 1) {
    double r = first(0);
#pragma unroll
    for (int i = 1; i < N; i++) r = acc(r, i);
    return r;
  } else {
    constexpr int aligned2 = (N / (2 * P)) * (2 * P), aligned = (N / P) * P;
    static_assert(aligned >= P, "at least one packet");
    double r0[P], r1[P];
#pragma unroll
    for (int l = 0; l < P; l++) r0[l] = first(l);
    if constexpr (aligned > P) {
#pragma unroll
      for (int l = 0; l < P; l++) r1[l] = first(P + l);
#pragma unroll
      for (int i = 2 * P; i < aligned2; i += 2 * P) {
#pragma unroll
        for (int l = 0; l < P; l++) {
          r0[l] = acc(r0[l], i + l);
          r1[l] = acc(r1[l], i + P + l);
        }
      }
#pragma unroll
      for (int l = 0; l < P; l++) r0[l] = r0[l] + r1[l];
      if constexpr (aligned > aligned2) {
#pragma unroll
        for (int l = 0; l < P; l++) r0[l] = acc(r0[l], aligned2 + l);
      }
    }
    double res;
    if constexpr (P == 2) res = r0[0] + r0[1];
    else if constexpr (SO == SO_AVX34_FMA) res = (r0[0] + r0[2]) + (r0[1] + r0[3]);  // predux, Eigen 3.4 (AVX: two 128-bit halves added first)
    else res = (r0[0] + r0[1]) + (r0[2] + r0[3]);  // predux, Eigen 3.3
#pragma unroll
    for (int i = aligned; i < N; i++) res = acc(res, i);
    return res;
  }
}
template <int SO, int N, typename A>
__device__ __forceinline__ double redux_sum(A a) {
  return redux_impl<SO, N, false>(a, a);
}
template <int SO, int N, typename A, typename B>
__device__ __forceinline__ double redux_prod(A a, B b) {
  return redux_impl<SO, N, true>(a, b);
}

// the same for run-time n / order (sc_helpers.hip: one wavefront per call, nothing to tune); a[i * sa], b[i * sb] (b may be null)
__device__ inline double redux_rt(int so, int n, const double *a, int sa, const double *b, int sb) {
  const int P = so == SO_SEQ ? 1 : (so == SO_SSE2 ? 2 : 4);
  const bool fused = so == SO_AVX_FMA || so == SO_AVX34_FMA;
  auto first = [&](int i) { return b ? a[i * sa] * b[i * sb] : a[i * sa]; };
  auto acc = [&](double r, int i) {
    if (!b) return r + a[i * sa];
    if (fused) return fma(a[i * sa], b[i * sb], r);
    const double p = a[i * sa] * b[i * sb];
    return r + p;
  };
  if (n == 0) return 0.0;
  const int aligned2 = (n / (2 * P)) * (2 * P), aligned = (n / P) * P;
  double res;
  if (P > 1 && aligned) {
    double r0[4], r1[4];
    for (int l = 0; l < P; l++) r0[l] = first(l);
    if (aligned > P) {
      for (int l = 0; l < P; l++) r1[l] = first(P + l);
      for (int i = 2 * P; i < aligned2; i += 2 * P)
        for (int l = 0; l < P; l++) {
          r0[l] = acc(r0[l], i + l);
          r1[l] = acc(r1[l], i + P + l);
        }
      for (int l = 0; l < P; l++) r0[l] = r0[l] + r1[l];
      if (aligned > aligned2)
        for (int l = 0; l < P; l++) r0[l] = acc(r0[l], aligned2 + l);
    }
    res = P == 2 ? r0[0] + r0[1] : (so == SO_AVX34_FMA ? (r0[0] + r0[2]) + (r0[1] + r0[3]) : (r0[0] + r0[1]) + (r0[2] + r0[3]));
    for (int i = aligned; i < n; i++) res = acc(res, i);
  } else {
    res = first(0);
    for (int i = 1; i < n; i++) res = acc(res, i);
  }
  return res;
}

}  // namespace dev
}  // namespace sc
}  // namespace rsx

// launch `CALL` (an expression that uses SO as a template argument) for the handle's summation order
#define RSX_SO_DISPATCH(so, CALL)                     \
  do {                                                \
    switch (so) {                                     \
      case RSX_SC_SUM_SEQ: {                          \
        constexpr int SO = RSX_SC_SUM_SEQ;            \
        CALL;                                         \
      } break;                                        \
      case RSX_SC_SUM_EIGEN_AVX_FMA: {                \
        constexpr int SO = RSX_SC_SUM_EIGEN_AVX_FMA;  \
        CALL;                                         \
      } break;                                        \
      case RSX_SC_SUM_EIGEN34_AVX_FMA: {              \
        constexpr int SO = RSX_SC_SUM_EIGEN34_AVX_FMA; \
        CALL;                                         \
      } break;                                        \
      default: {                                      \
        constexpr int SO = RSX_SC_SUM_EIGEN_SSE2;     \
        CALL;                                         \
      } break;                                        \
    }                                                 \
  } while (0)
