// sc_entry_dev.h -- what ONE database entry needs on the device, as inline device functions: the descriptor and its keys
// from a point cloud (SC.cpp:151-227), the fp16 image of the direct filter / the window kernel, and the fp16 hi/lo
// sector-key image of the window kernel.  They are the bodies of sc_build_kernel (sc_kernels.hip), sc_img_db_kernel
// (sc_filter.hip) and sc_win_db_keys_kernel (sc_window.hip); sc_insert_kernel (sc_spec.hip, next to the spectra) runs all
// of them for one new keyframe in ONE launch.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "sc_kernels.h"
#include "sc_redux_dev.h"

namespace rsx {
namespace sc {
namespace dev {

__device__ __forceinline__ void wave_lds_fence() {
  // LDS operations of one wave execute in order; this only stops the compiler from moving LDS
  // accesses across the point where lanes exchange data through LDS.
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// wave-wide reductions on 64-bit values by DPP inside the rows of 16 lanes + four v_readlane (a __shfl_xor butterfly is
// two ds_bpermute per step: ~1.2 k cycles of dependent LDS-crossbar round trips for one 64-bit reduction)
template <int CTRL>
__device__ __forceinline__ unsigned long long dpp_u64(unsigned long long x) {
  const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)x, CTRL, 0xf, 0xf, false);
  const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(x >> 32), CTRL, 0xf, 0xf, false);
  return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long readlane_u64(unsigned long long x, int l) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)x, l), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(x >> 32), l);
  return ((unsigned long long)hi << 32) | lo;
}
template <bool MAX>
__device__ __forceinline__ unsigned long long wave_minmax_u64(unsigned long long v) {
  auto pick = [](unsigned long long a, unsigned long long b) { return MAX ? (a > b ? a : b) : (a < b ? a : b); };
  v = pick(v, dpp_u64<0xB1>(v));   // quad_perm [1,0,3,2]
  v = pick(v, dpp_u64<0x4E>(v));   // quad_perm [2,3,0,1]
  v = pick(v, dpp_u64<0x141>(v));  // row_half_mirror
  v = pick(v, dpp_u64<0x140>(v));  // row_mirror: every lane = its row's result
  return pick(pick(readlane_u64(v, 0), readlane_u64(v, 16)), pick(readlane_u64(v, 32), readlane_u64(v, 48)));
}
__device__ __forceinline__ double wave_sum_f64(double x) {  // (row sums in butterfly order, then row 0 + 1 + 2 + 3)
  auto d = [](double a) { return (unsigned long long)__double_as_longlong(a); };
  auto f = [](unsigned long long a) { return __longlong_as_double((long long)a); };
  x = x + f(dpp_u64<0xB1>(d(x)));
  x = x + f(dpp_u64<0x4E>(d(x)));
  x = x + f(dpp_u64<0x141>(d(x)));
  x = x + f(dpp_u64<0x140>(d(x)));
  return ((f(readlane_u64(d(x), 0)) + f(readlane_u64(d(x), 16))) + f(readlane_u64(d(x), 32))) + f(readlane_u64(d(x), 48));
}

// ------------------------------------------------------------------------------------------
// keys: one wave per descriptor
// ------------------------------------------------------------------------------------------
// sector key + column norm of ONE column held in registers (c[i] = elements 4 i .. 4 i + 3)
// Eigen 3.3 redux order of the reference build (SSE2, 2-double packets; oracle/sc_ref.c "reductions"):
// term i goes to accumulator i % 4 = (packet accumulator i/2 % 2, lane i % 2); result (a0 + a2) + (a1 + a3)
template <int SO = SO_SSE2>
__device__ __forceinline__ void column_keys(const float4 (&c)[5], double &vkey, double &norm) {
  if constexpr (SO != SO_SSE2) {  // the reference built with another packet size (sc_redux_dev.h)
    double x[NR];
#pragma unroll
    for (int i = 0; i < 5; i++) {
      x[4 * i] = c[i].x; x[4 * i + 1] = c[i].y; x[4 * i + 2] = c[i].z; x[4 * i + 3] = c[i].w;
    }
    auto at = [&](int i) { return x[i]; };
    vkey = redux_sum<SO, NR>(at) / (double)NR;
    norm = sqrt(redux_prod<SO, NR>(at, at));
    return;
  }
  double s0, s1, s2, s3, q0, q1, q2, q3;
  {
    const float4 v = c[0];
    double x0 = v.x, x1 = v.y, x2 = v.z, x3 = v.w;
    s0 = x0; s1 = x1; s2 = x2; s3 = x3;
    q0 = x0 * x0; q1 = x1 * x1; q2 = x2 * x2; q3 = x3 * x3;  // exact in fp64
  }
#pragma unroll
  for (int i = 1; i < 5; i++) {
    const float4 v = c[i];
    double x0 = v.x, x1 = v.y, x2 = v.z, x3 = v.w;
    s0 = s0 + x0; q0 = fma(x0, x0, q0);  // x*x exact in fp64 -> fma == mul+add
    s1 = s1 + x1; q1 = fma(x1, x1, q1);
    s2 = s2 + x2; q2 = fma(x2, x2, q2);
    s3 = s3 + x3; q3 = fma(x3, x3, q3);
  }
  vkey = ((s0 + s2) + (s1 + s3)) / (double)NR;  // SC.cpp:224 mean()
  norm = sqrt((q0 + q2) + (q1 + q3));           // Eigen norm()
}

// sector key + column norm of the lane's column (lanes < 60)
template <int SO = SO_SSE2>
__device__ __forceinline__ void wave_keys_columns(const float *__restrict__ d, double *__restrict__ vkey,
                                                  double *__restrict__ norm, int lane) {
  if (lane < NS) {
    const float4 *p = reinterpret_cast<const float4 *>(d + lane * NR);
    const float4 c[5] = {p[0], p[1], p[2], p[3], p[4]};
    double vk, nr;
    column_keys<SO>(c, vk, nr);
    vkey[lane] = vk;
    norm[lane] = nr;
  }
}

template <int SO = SO_SSE2>
__device__ __forceinline__ void wave_keys(const float *__restrict__ d, double *__restrict__ vkey,
                                          double *__restrict__ norm, float *__restrict__ rkey, int lane) {
  wave_keys_columns<SO>(d, vkey, norm, lane);
  if constexpr (SO != SO_SSE2) {
    if (lane < NR) rkey[lane] = (float)(redux_sum<SO, NS>([&](int c) { return (double)d[c * NR + lane]; }) / (double)NS);
    return;
  }
  if (lane < NR) {
    double a[4];
#pragma unroll
    for (int c = 0; c < 4; c++) a[c] = (double)d[c * NR + lane];
    for (int c = 4; c < NS; c += 4) {
#pragma unroll
      for (int l = 0; l < 4; l++) a[l] = a[l] + (double)d[(c + l) * NR + lane];
    }
    rkey[lane] = (float)(((a[0] + a[2]) + (a[1] + a[3])) / (double)NS);  // SC.cpp:208 mean(), SC.cpp:64 float narrowing
  }
}

// ------------------------------------------------------------------------------------------
// build: one 256-thread block per cloud; LDS max-histogram on order-preserving int encodings
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned enc_f32(float f) {
  unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float dec_f32(unsigned u) {
  u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
  return __uint_as_float(u);
}

// SC.cpp:23-36; float division, atan in double, result narrowed to float (oracle/sc_ref.c)
__device__ __forceinline__ float xy2theta_dev(float x, float y) {
  const double k = 180 / M_PI;
  if ((x >= 0) & (y >= 0)) return (float)(k * atan((double)__fdiv_rn(y, x)));
  if ((x < 0) & (y >= 0)) return (float)(180 - (k * atan((double)__fdiv_rn(y, -x))));
  if ((x < 0) & (y < 0)) return (float)(180 + (k * atan((double)__fdiv_rn(y, x))));
  if ((x >= 0) & (y < 0)) return (float)(360 - (k * atan((double)__fdiv_rn(-y, x))));
  return __builtin_nanf("");
}

__device__ __forceinline__ int ceil_clamp(double v, int hi) {
  double c = ceil(v);
  int i;
  if (!(c == c)) i = 1;  // NaN: x86 cvttsd2si gives INT_MIN, then max(.,1) (SC.cpp:178-179)
  else if (c >= (double)hi) i = hi;
  else if (c <= 1.0) i = 1;
  else i = (int)c;
  return i;
}

// makeScancontext + the three key builders for one cloud, by one 256-thread block (SC.cpp:151-227); bins: DS words of LDS,
// which hold the descriptor as floats when the function returns (after a __syncthreads the caller adds if other waves read it)
template <int SO = SO_SSE2>
__device__ __forceinline__ void build_block(const char *__restrict__ pts, int64_t n_pts, int64_t stride, double lidar_height,
                                            double max_radius, unsigned *bins, float *__restrict__ out_desc,
                                            double *__restrict__ out_vkey, double *__restrict__ out_norm,
                                            float *__restrict__ out_rkey) {
  const unsigned no_point = enc_f32(-1000.0f);  // SC.cpp:158-159
  for (int i = threadIdx.x; i < DS; i += 256) bins[i] = no_point;
  __syncthreads();
  for (int64_t i = threadIdx.x; i < n_pts; i += 256) {
    const float *p = reinterpret_cast<const float *>(pts + i * stride);
    float x = p[0], y = p[1];
    float z = (float)((double)p[2] + lidar_height);  // SC.cpp:168
    if (!(x == x) || !(y == y) || !(z == z)) continue;
    float ss = __fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y));
    float azim_range = (float)sqrt((double)ss);  // SC.cpp:171 (== correctly rounded sqrtf)
    float azim_angle = xy2theta_dev(x, y);       // SC.cpp:172
    if ((double)azim_range > max_radius) continue;  // SC.cpp:175
    int ring = ceil_clamp(((double)azim_range / max_radius) * NR, NR);   // SC.cpp:178
    int sector = ceil_clamp(((double)azim_angle / 360.0) * NS, NS);      // SC.cpp:179
    atomicMax(&bins[(sector - 1) * NR + (ring - 1)], enc_f32(z));        // SC.cpp:182-183
  }
  __syncthreads();
  float *sd = reinterpret_cast<float *>(bins);
  for (int i = threadIdx.x; i < DS; i += 256) {
    unsigned u = bins[i];
    float v = (u == no_point) ? 0.0f : dec_f32(u);  // SC.cpp:187-190
    sd[i] = v;
    out_desc[i] = v;
  }
  __syncthreads();
  if (threadIdx.x < 64) wave_keys<SO>(sd, out_vkey, out_norm, out_rkey, threadIdx.x);
}

constexpr double kImgScale = 32768.0;  // 2^15 on both operands of the direct filter's GEMM
constexpr unsigned long long kEntryNonFinite = 1ull << 63;

// one column (registers, c[i] = elements 4 i .. 4 i + 3) -> 20 scaled fp16 values in st[0 .. 20); returns (nonzero, nonfinite)
__device__ __forceinline__ void normalise_column_regs(const float4 (&c)[5], double nrm, _Float16 *st, bool &nonzero, bool &bad) {
  nonzero = !(nrm == 0.0);  // SC.cpp:78: a column takes part unless its norm == 0
  bad = false;
  // one division per column: x * (2^15 / norm) instead of (x / norm) * 2^15 per element -- 2 ulp of fp64 apart, nothing next to
  // the fp16 rounding that follows (a correctly rounded fp64 division is ~35 instructions, and there were 1200 per image)
  const double scale = kImgScale / nrm;
#pragma unroll
  for (int i = 0; i < 5; i++) {
    const float4 v = c[i];
    const float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; e++) {
      double y = nonzero ? (double)x[e] * scale : 0.0;
      bad |= !(fabs(y) <= kImgScale * 1.0000001);  // NaN or inf (|x| <= norm: 2^15 up to the rounding of `scale`)
      st[4 * i + e] = (_Float16)(float)y;
    }
  }
}
// column j of one descriptor -> 20 scaled fp16 values in st[j*20 ..]; returns (nonzero, nonfinite)
__device__ __forceinline__ void normalise_column(const float *__restrict__ d, double nrm, _Float16 *st,
                                                 bool &nonzero, bool &bad) {
  const float4 *p = reinterpret_cast<const float4 *>(d);
  const float4 c[5] = {p[0], p[1], p[2], p[3], p[4]};
  normalise_column_regs(c, nrm, st, nonzero, bad);
}

// query image of the direct filter / the window kernel from the normalised columns in st (DS halves of LDS): two displaced
// copies of the doubled image + the column mask in the gap between them (layout: sc_filter.hip "query image"), FILTER_QIMG_BYTES
// at `out` (global memory or LDS).
// (first, step): chunk c = first, first + step, ... is written by this thread (one wave: lane, 64)
__device__ __forceinline__ void img_query_image(const _Float16 *st, unsigned long long m, uint4 *out, int first, int step = 64) {
  typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
  constexpr int kEven = FILTER_QIMG_MASK_OFF / 16, kGap = (FILTER_QIMG_ODD - FILTER_QIMG_MASK_OFF) / 16;
  for (int c = first; c < FILTER_QIMG_BYTES / 16; c += step) {
    half8_t v;
    if (c < kEven) {
      const int e0 = (8 * c) % DS;  // 1200 is a multiple of 8: no wrap inside a chunk
      v = *reinterpret_cast<const half8_t *>(&st[e0]);
    } else if (c < kEven + kGap) {
      // the gap carries the query's column mask (read by the filter kernel with one ds_read_b64)
      const uint4 g = {c == kEven ? (unsigned)m : 0u, c == kEven ? (unsigned)(m >> 32) : 0u, 0u, 0u};
      v = *reinterpret_cast<const half8_t *>(&g);
    } else {
      const int e0 = 4 + 8 * (c - kEven - kGap);
#pragma unroll
      for (int i = 0; i < 8; i++) v[i] = st[(e0 + i) % DS];
    }
    out[c] = *reinterpret_cast<const uint4 *>(&v);
  }
}

// database image of one entry (one wave; st: DS halves of LDS): fp16, tile-major [tile of 32 entries][75 K-steps][64 lanes]
// [8 halves] (hnT, sc_filter.hip) and once more entry-major (hnR, sc_window.hip gathers single entries) + the column mask
__device__ __forceinline__ void img_db_entry(const float *__restrict__ desc, const double *__restrict__ norm, int64_t slot,
                                             _Float16 *st, uint4 *__restrict__ hnT, uint4 *__restrict__ hnR,
                                             unsigned long long *__restrict__ cmask, int lane) {
  constexpr int kSteps = DS / 16;
  bool nonzero = false, bad = false;
  if (lane < NS) normalise_column(desc + slot * DS + lane * NR, norm[slot * NS + lane], &st[lane * NR], nonzero, bad);
  unsigned long long m = __ballot(nonzero && lane < NS);
  if (__ballot(bad && lane < NS)) m |= kEntryNonFinite;
  wave_lds_fence();
  const int64_t tile = slot >> 5;
  const int col = (int)(slot & 31);
  for (int c = lane; c < 2 * kSteps; c += 64) {
    const uint4 v = *reinterpret_cast<const uint4 *>(&st[c * 8]);
    hnT[(tile * kSteps + (c >> 1)) * 64 + (c & 1) * 32 + col] = v;
    hnR[slot * (2 * kSteps) + c] = v;
  }
  if (lane == 0) cmask[slot] = m;
}

constexpr double kWinMaxKeyNorm = 4.0e6;

// scaled hi/lo split of one 60-element sector key held one element per lane (lanes >= 60: 0)
struct KeySplit {
  _Float16 hi, lo;
  float nrm;   // sqrt(sum x^2), rounded up; NaN when the key has a non-finite element or is too large (below)
  float unrm;  // the same of the unscaled key
};
// DPP: the two wave reductions by DPP instead of shuffle butterflies (sc_q1.hip, where they sit on the critical path of every
// launch): the same maximum (|x| compared as integers: a NaN's bit pattern is above every number's, so it wins as well); the
// sum of squares is taken in another order, which moves the norms -- error-bound slack, rounded up by 1e-6 -- by an ulp
template <bool DPP = false>
__device__ __forceinline__ KeySplit split_key(double v, int lane) {
  const double av = lane < NS ? fabs(v) : 0.0;
  double mx = av;
  if constexpr (DPP) {
    mx = __longlong_as_double((long long)wave_minmax_u64<true>((unsigned long long)__double_as_longlong(av)));
  } else {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const double o = __shfl_xor(mx, off);
      mx = (o > mx || !(o == o)) ? o : mx;  // a NaN wins
    }
  }
  KeySplit r;
  r.hi = (_Float16)0.0f;
  r.lo = (_Float16)0.0f;
  r.nrm = __builtin_nanf("");
  r.unrm = __builtin_nanf("");
  if (!(mx < INFINITY)) return r;  // NaN / inf somewhere (uniform)
  int e = 0;
  if (mx > 0.0) {
    (void)frexp(mx, &e);  // mx = f * 2^e, f in [0.5, 1)
    e = 10 - e;           // scaled maximum in [2^9, 2^10)
  }
  const double x = lane < NS ? ldexp(v, e) : 0.0;
  const _Float16 hi = (_Float16)(float)x;
  const _Float16 lo = (_Float16)(float)(x - (double)(float)hi);
  double s = x * x;
  if constexpr (DPP) {
    s = wave_sum_f64(s);
  } else {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
  }
  r.hi = hi;
  r.lo = lo;
  r.nrm = (float)(sqrt(s) * (1.0 + 1e-6));
  // the reference's search starts from min_veq_norm = 1e7 (SC.cpp:100-106: a shift whose key distance is not below
  // that is never taken, and if none is the alignment stays 0).  ||vkey_q - shift(vkey_e)|| <= ||vkey_q|| + ||vkey_e||:
  // with both norms below 4e6 the test passes for every shift and the argmin is the plain argmin; larger keys are
  // left to the exact alignment of the re-scoring kernel
  const double un = ldexp(sqrt(s), -e);
  if (!(un < kWinMaxKeyNorm)) r.nrm = __builtin_nanf("");
  r.unrm = (float)un;
  return r;
}

// database side of the window kernel's alignment: [slot][hi 0..63 | lo 0..63] fp16 (elements 60..63 zero: the K padding)
// + the key's scaled norm (one wave)
__device__ __forceinline__ void win_db_keys_entry(const double *__restrict__ vkey, int64_t slot, _Float16 *__restrict__ vk16,
                                                  float *__restrict__ vk_n, int lane) {
  const KeySplit k = split_key(lane < NS ? vkey[slot * NS + lane] : 0.0, lane);
  vk16[slot * 128 + lane] = k.hi;
  vk16[slot * 128 + 64 + lane] = k.lo;
  if (lane == 0) {
    vk_n[2 * slot] = k.nrm;
    vk_n[2 * slot + 1] = k.unrm;
  }
}

}  // namespace dev
}  // namespace sc
}  // namespace rsx
