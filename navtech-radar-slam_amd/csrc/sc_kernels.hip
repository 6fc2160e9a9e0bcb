// sc_kernels.hip -- hand-written gfx950 (CDNA4) kernels of the ScanContext hot path.
//
// What each kernel replaces in the reference (pgo/SC-A-LOAM/include/scancontext/Scancontext.cpp):
//   sc_build_kernel   makeScancontext + the three key builders           SC.cpp:151-227
//   sc_keys_kernel    makeRingkey/makeSectorkey (+ column norms)         SC.cpp:198-227, 78
//   pair_group<B>     distanceBtnScanContext for one query x B entries   SC.cpp:69-148   (device function)
//   sc_pair_kernel    pair_group over every entry / a gather list        SC.cpp:380-395  (small problems, the 3
//                     kd-tree candidates, rsx_sc_pair_distances)
//   sc_rescore_kernel pair_group over the entries the MFMA filter (sc_filter.hip) could not exclude,
//                     in rounds of ascending bound with tau tightening (batched exhaustive queries)
//   sc_merge_kernel   the strict-< "first wins" candidate loop           SC.cpp:380-395 (generalised to top-k)
//   sc_knn_kernel     nanoflann 3-NN on ring keys                        SC.cpp:367-374, NF.hpp:383-408
//
// Numerics contract (tests/test_gpu_sc.py asserts it bit-for-bit against oracle/sc_ref.c):
//   the reductions the reference takes through Eigen (mean, norm, dot) follow Eigen 3.3's redux order for
//   the reference build (SSE2: term i -> accumulator i % 4, (a0 + a2) + (a1 + a3)), the one it takes in
//   a scalar loop (SC.cpp:83) is sequential -- exactly like the oracle, which is itself checked bit for bit
//   against the reference's own Scancontext.cpp (tests/test_oracle_pin.py); this file is
//   compiled with -ffp-contract=off and uses fma() only where the product is exact (fp32 x fp32 in
//   fp64), so results are identical to mul+add.  fp64 sqrt and divide are IEEE correctly rounded.
//
// Mapping (one 64-lane wavefront = one query x B database entries per iteration):
//   stage 1  lane = column shift k (60 of 64 lanes): S_k = sum_c (v1[c] - v2[(c-k)%60])^2, four interleaved
//            chains in c (Eigen's order); the entry's sector key sits twice in LDS so lane k reads v2d[60+c-k] with an
//            immediate offset -> conflict-free ds_read_b64, no address VALU in the loop.
//   stage 2  lane = entry column j, held in 40 VGPRs as fp64 (20 cvt per entry, loaded straight from
//            HBM/L2 -- the entry image never goes through LDS); the query lives once per block in LDS
//            as an fp64 image and lane j reads query column (j+k)%60 (10 x ds_read_b128 per shift),
//            7 shifts, 20 fma each.
//   stage 3  the reference sums the 60 column similarities sequentially (SC.cpp:83); the 7 x B
//            series are transposed through LDS so lane (b,t) adds its own series in order.
//   top-k    each wave keeps a sorted k-list one record per lane; per-wave lists are merged by
//            sc_merge_kernel under the total order (dist, global index).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "rsx_common.h"
#include "sc_kernels.h"
#include "sc_entry_dev.h"
#include "sc_exact_dev.h"

namespace rsx {
namespace sc {

namespace {

using dev::wave_keys;


template <int SO>
__global__ __launch_bounds__(256) void sc_keys_kernel(const float *__restrict__ desc, int64_t n,
                                                      double *__restrict__ vkey, double *__restrict__ norm,
                                                      float *__restrict__ rkey) {
  const int lane = threadIdx.x & 63;
  int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  wave_keys<SO>(desc + i * DS, vkey + i * NS, norm + i * NS, rkey + i * NR, lane);
}

template <int SO>
__global__ __launch_bounds__(256) void sc_build_kernel(const char *__restrict__ pts, int64_t n_pts,
                                                       int64_t stride, double lidar_height,
                                                       double max_radius, float *__restrict__ out_desc,
                                                       double *__restrict__ out_vkey,
                                                       double *__restrict__ out_norm,
                                                       float *__restrict__ out_rkey) {
  __shared__ __attribute__((aligned(16))) unsigned bins[DS];
  dev::build_block<SO>(pts, n_pts, stride, lidar_height, max_radius, bins, out_desc, out_vkey, out_norm, out_rkey);
}

// ------------------------------------------------------------------------------------------
// pair kernel
// ------------------------------------------------------------------------------------------
// LDS of one 256-thread block (all 4 waves score the SAME query, blockIdx.y):
//   [0, 10560)          query image in fp64, column stride 176 B (44 dwords: conflict-free
//                       ds_read_b128 for 16 consecutive columns), converted once per block
//   [10560, +512)       query column norms n1[60]
//   [11072, +512)       query sector key v1[60]
//   per wave: B x { sector key images A and B (2 x ~968 B) | similarity terms 7 x 60 doubles (3360 B)
//                   | neff[7] + k* (32 B) }
constexpr int Q_COL_STRIDE = 176;
constexpr int OFF_QIMG = 0;
constexpr int OFF_QN1 = 60 * Q_COL_STRIDE;       // 10560
constexpr int OFF_QV1 = OFF_QN1 + 512;           // 11072
constexpr int OFF_WAVES = OFF_QV1 + 512;         // 11584
// pruning preview (sc_walk_kernel): the query's unit columns in fp32 (column stride 80 B = 5 LDS slots, odd:
// conflict-free ds_read_b128 across the lanes of a group) + a flag "every non-empty column norm is in
// [1e-15, 1e15]"; lives behind the per-wave region of the kernels that use it
constexpr int QP_COL_STRIDE = NR * 4;            // 80
constexpr int QP_FLAG = NS * QP_COL_STRIDE;      // 4800
constexpr int QP_MASK = QP_FLAG + 8;             // 4808: bit c = query column c is non-empty (norm != 0)
constexpr int QP_V1F = QP_FLAG + 16;             // 4816: the query's sector key in fp32 (fast alignment, 240 B)
constexpr int QP_SIZE = QP_V1F + NS * 4;         // 5056

template <int B>
struct PairLds {
  static constexpr int WAVE_SIZE = B * ENT_SIZE;
  static constexpr int SIZE = OFF_WAVES + 4 * WAVE_SIZE;
};

struct PairArgs {
  DbView db;
  QueryView q;
  const int32_t *gather;
  int64_t first, n_items, n_eligible;
  const int64_t *q_elig;
  double *out_dist;
  int32_t *out_shift;
  rsx_sc_hit *partial;
  int32_t k, nslots;
};


// waves per SIMD the LDS footprint allows (4 waves per 256-thread block, 160 KiB LDS per CU)
template <int B>
constexpr int pair_waves_per_simd() {
  constexpr int blocks = (160 * 1024) / PairLds<B>::SIZE;
  return blocks < 1 ? 1 : (blocks > 4 ? 4 : blocks);
}

// ------------------------------------------------------------------------------------------
// distanceBtnScanContext (SC.cpp:116-148) of the block's query (fp64 image, norms and sector key at
// smem + OFF_Q*) against B database entries, by one wavefront.  wsm = the wave's private LDS region
// (B x ENT_SIZE).  Result of entry b: lanes 8*b .. 8*b+7 all hold (bd, bk) = (distance, shift),
// {1e7, 0} when no shift of the window has an effective column (SC.cpp:133-134 initial values).
// ------------------------------------------------------------------------------------------

// FAST alignment (sc_rescore_kernel): fastAlignUsingVkey asks for the FIRST strict minimum over the 60 shifts of a
// 60-term fp64 sum.  The 60 sums are first evaluated in fp32 (15 x ds_read_b128 + 4 packed VALU per lane instead of
// 30 x 2 reads + 180 fp64 operations).  With E = ||v1||^2 + ||v2||^2, every fp32 value D~[k] is within
//   eps_D = 129.3 * 2^-24 * E  (conversion + subtraction rounding 8.02 u E, 60 fused accumulations 121.3 u E)
// of the real sum, the fp64 sums of the reference within 60 * 2^-53 of it, so every shift that can be the fp64
// minimum satisfies D~[k] <= min D~ + 2 eps_D.  When exactly one shift does, it IS the reference's argmin (its
// norm is also < 1e7, checked on D~); otherwise -- exact ties between shifts are common for binary radar
// descriptors, whose sector keys are multiples of 0.1 -- the exact fp64 stage below decides as before.  Non-finite
// or huge values take the exact stage too.
constexpr float kFastAlignEps = 1.6e-5f;  // > 2^-16.5 = 129.3 * 2^-24 * 1.04, on E evaluated in fp32
// four images of the entry's key in fp32, image j displaced by j elements (img_j[t] = vk2[t + j], vk2[e] = v[e % 60])
// so that lane k reads vk2[60 - k + c .. + 3] with one aligned ds_read_b128 from image (-k) mod 4; the bases (in
// 16-byte slots: 0, 37, 73, 109 = 0, 5, 9, 13 mod 16) leave a single 2-way bank conflict over the four lane groups
__host__ __device__ constexpr int fimg_base(int j) { return (j == 0 ? 0 : j == 1 ? 37 : j == 2 ? 73 : 109) * 16; }
typedef float float2p __attribute__((ext_vector_type(2)));

// PREVIEW (sc_walk_kernel): once the alignment k* is known, the 7 window distances are first evaluated
// cheaply in fp32 -- dot products of the query's unit columns (fp32 image in LDS at smem + off_preview)
// with the entry's raw fp32 column, scaled by 1/n2, summed by a wave reduction: no fp64 division, no
// sequential sum -- and the exact stages 2-3 only run when that preview, minus a margin far above its
// error (a few 1e-6: 20-term dot products and a 60-term sum of values <= 1 in fp32), can still reach
// tau_prune = the k-th best exact distance so far.  Column norms outside [1e-15, 1e15] (fp32 products
// could under- or overflow) and non-finite previews disable it.  A pruned group returns {+inf, ...}
// ("not a hit"); with tau_prune = +inf nothing is pruned.
constexpr float kPreviewMargin = 1e-4f;


template <int B, bool PREVIEW = false, bool FAST = false, int SO = dev::SO_SSE2>
__device__ __forceinline__ void pair_group(const DbView &db, const char *smem, char *wsm, int lane,
                                           const int64_t (&eslot)[B], double &bd_out, int &bk_out,
                                           double tau_prune = INFINITY, int off_preview = 0,
                                           const EntryRegs *pre = nullptr, float e1 = 0.0f) {
  static_assert(!FAST || B == 1, "the fast alignment takes its decisions per wave: one entry per group");
  const int cl = lane < NS ? lane : 0;       // entry column owned in stage 2
  const int kk = lane < NS ? lane : NS - 1;  // shift owned in stage 1
  const double *v1 = reinterpret_cast<const double *>(smem + OFF_QV1);
  const double *qn1 = reinterpret_cast<const double *>(smem + OFF_QN1);
  float4 ecol[B][5];
  double en2[B], ev[B];
    wave_lds_fence();
#pragma unroll
    for (int b = 0; b < B; b++) {
      if (pre) {
        ev[b] = pre[b].v;
#pragma unroll
        for (int i = 0; i < 5; i++) ecol[b][i] = pre[b].ecol[i];
        en2[b] = pre[b].n2;
      } else {
        ev[b] = db.vkey[eslot[b] * NS + cl];
        const float4 *src = reinterpret_cast<const float4 *>(db.desc + eslot[b] * DS + cl * NR);
#pragma unroll
        for (int i = 0; i < 5; i++) ecol[b][i] = src[i];
        en2[b] = db.norm[eslot[b] * NS + cl];
      }
    }

    int kstar[B];
    bool need_exact = true;  // wave-uniform
    if constexpr (FAST) {
      const float vf = (float)ev[0];
      if (lane < NS) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
          float *img = reinterpret_cast<float *>(wsm + fimg_base(j));
          if (lane >= j) img[lane - j] = vf;
          img[lane + NS - j] = vf;
          if (lane < j) img[lane + 2 * NS - j] = vf;
        }
      }
      const float e2 = wave_sum_f32(lane < NS ? vf * vf : 0.0f);
      wave_lds_fence();
      const int j = (-kk) & 3;
      const float4 *yp = reinterpret_cast<const float4 *>(wsm + fimg_base(j) + (NS - kk - j) * 4);
      const float4 *xp = reinterpret_cast<const float4 *>(smem + off_preview + QP_V1F);
      float2p a0 = {0.0f, 0.0f}, a1 = {0.0f, 0.0f};
#pragma unroll
      for (int i = 0; i < NS / 4; i++) {
        const float4 x = xp[i], y = yp[i];
        const float2p d0 = float2p{x.x, x.y} - float2p{y.x, y.y}, d1 = float2p{x.z, x.w} - float2p{y.z, y.w};
        a0 = __builtin_elementwise_fma(d0, d0, a0);
        a1 = __builtin_elementwise_fma(d1, d1, a1);
      }
      const float D = (a0[0] + a0[1]) + (a1[0] + a1[1]);
      const float dmin = wave_min_f32(lane < NS ? D : INFINITY);  // fminf drops NaNs: they are caught by the ballot below
      const float thr = dmin + 2.0f * kFastAlignEps * (e1 + e2);
      const unsigned long long nan_bal = __ballot(lane < NS && !(D == D));
      const unsigned long long cand = __ballot(lane < NS && D <= thr);
      // dmin < 1e12: the winning norm is < 1e6 < the reference's 1e7 init (SC.cpp:96); thr finite: E finite
      if (!nan_bal && dmin < 1e12f && thr < 3.0e38f && __popcll(cand) == 1) {
        kstar[0] = __ffsll((long long)cand) - 1;
        need_exact = false;
      }
      wave_lds_fence();  // the fp64 key images / similarity terms overwrite the fp32 images
    }

    if (need_exact) {
#pragma unroll
    for (int b = 0; b < B; b++) {
      const double v = ev[b];
      double *vka = reinterpret_cast<double *>(wsm + b * ENT_SIZE + ENT_VKEY_A);
      double *vkb = reinterpret_cast<double *>(wsm + b * ENT_SIZE + ENT_VKEY_B);
      if (lane < NS) {
        vka[lane] = v;
        vka[lane + NS] = v;
        vkb[lane + 1] = v;
        vkb[lane + NS + 1] = v;
      }
    }
    wave_lds_fence();

    // ---- stage 1: fastAlignUsingVkey (SC.cpp:93-113), lane = shift ----
    // Eigen's redux order (SSE2 build of the reference): term c goes to accumulator c % 4, the norm is
    // sqrt((a0 + a2) + (a1 + a3)); the first term of each accumulator initialises it (0.0 + x*x == x*x)
    double acc[B][4];
#pragma unroll
    for (int b = 0; b < B; b++)
#pragma unroll
      for (int l = 0; l < 4; l++) acc[b][l] = 0.0;
    {
      // lane k needs vk2[60 + c - k] for c = 0..59.  Even k: image A, odd k: image B -- either way the
      // pair (c, c+1), c even, is one aligned 16-byte read
      const double2 *v2[B];
      const int eoff = (kk & 1) ? (ENT_VKEY_B + (NS + 1 - kk) * 8) : (ENT_VKEY_A + (NS - kk) * 8);
#pragma unroll
      for (int b = 0; b < B; b++) v2[b] = reinterpret_cast<const double2 *>(wsm + b * ENT_SIZE + eoff);
      const double2 *v1p = reinterpret_cast<const double2 *>(v1);
      // chunks of 12 columns: bounded live ranges, immediate offsets inside a chunk
#pragma unroll 1
      for (int c0 = 0; c0 < NS / 2; c0 += 6) {
#pragma unroll
        for (int cc = 0; cc < 6; cc++) {
          const double2 x = v1p[c0 + cc];
#pragma unroll
          for (int b = 0; b < B; b++) {
            const double2 y = v2[b][c0 + cc];
            double d0 = x.x - y.x;
            double dd0 = d0 * d0;
            acc[b][2 * (cc & 1)] = acc[b][2 * (cc & 1)] + dd0;          // columns 2(c0+cc): c % 4 in {0, 2}
            double d1 = x.y - y.y;
            double dd1 = d1 * d1;
            acc[b][2 * (cc & 1) + 1] = acc[b][2 * (cc & 1) + 1] + dd1;  // c % 4 in {1, 3}
          }
        }
      }
    }
#pragma unroll
    for (int b = 0; b < B; b++) {
      double nrm = sqrt((acc[b][0] + acc[b][2]) + (acc[b][1] + acc[b][3]));  // SC.cpp:105 norm()
      if constexpr (SO != dev::SO_SSE2) {  // the reference built with another packet size: the same 60 terms in its order
        const int eoff = (kk & 1) ? (ENT_VKEY_B + (NS + 1 - kk) * 8) : (ENT_VKEY_A + (NS - kk) * 8);
        const double *v2d = reinterpret_cast<const double *>(wsm + b * ENT_SIZE + eoff);
        auto d = [&](int c) { return v1[c] - v2d[c]; };
        nrm = sqrt(dev::redux_prod<SO, NS>(d, d));
      }
      bool ok = (lane < NS) && (nrm < kBig);     // SC.cpp:96,104 (NaN never passes `<`)
      double m = ok ? nrm : INFINITY;
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) m = fmin(m, __shfl_xor(m, off));
      unsigned long long bal = __ballot(ok && nrm == m);
      kstar[b] = bal ? (__ffsll((long long)bal) - 1) : 0;  // first strict minimum = lowest shift
    }
    }  // need_exact

    if constexpr (PREVIEW) {
      if (tau_prune < INFINITY && *reinterpret_cast<const int *>(smem + off_preview + QP_FLAG)) {  // wave-uniform
        bool any_alive = false;
#pragma unroll
        for (int b = 0; b < B; b++) {
          const int ks = kstar[b];
          const double n2 = en2[b];
          const bool n2_ok = (n2 == 0.0) || (n2 >= 1e-15 && n2 <= 1e15);
          if (__ballot(!n2_ok && lane < NS)) {  // unusual scale (or NaN): no preview for this entry
            any_alive = true;
            continue;
          }
          const float r2 = (n2 == 0.0) ? 0.0f : (float)(1.0 / n2);
          float best = INFINITY;
#pragma unroll 1
          for (int t = 0; t < 7; t++) {
            int k = ks + t - 3;
            k += (k < 0) ? NS : 0;
            k -= (k >= NS) ? NS : 0;
            int c = cl + k;
            c -= (c >= NS) ? NS : 0;
            const float4 *qp = reinterpret_cast<const float4 *>(smem + off_preview + c * QP_COL_STRIDE);
            float dot = 0.0f;
#pragma unroll
            for (int i = 0; i < 5; i++) {
              const float4 q4 = qp[i];
              dot = fmaf(q4.x, ecol[b][i].x, dot);
              dot = fmaf(q4.y, ecol[b][i].y, dot);
              dot = fmaf(q4.z, ecol[b][i].z, dot);
              dot = fmaf(q4.w, ecol[b][i].w, dot);
            }
            const bool valid = (lane < NS) && !((qn1[c] == 0.0) | (n2 == 0.0));  // SC.cpp:78
            const float sum = wave_sum_f32(valid ? dot * r2 : 0.0f);
            const int ne = __popcll(__ballot(valid));
            const float d = 1.0f - sum / (float)ne;  // ne == 0: NaN, ignored like SC.cpp:87-88,134
            if (!(d == d) && ne != 0) best = -INFINITY;  // non-finite data: never prune
            best = fminf(best, d);                        // fminf drops the 0/0 NaN
          }
          // exact distance >= preview - error: it cannot enter a top-k whose k-th distance is tau_prune
          if (!(best - kPreviewMargin > (float)tau_prune)) any_alive = true;
        }
        if (!any_alive) {
          bd_out = INFINITY;
          bk_out = 0x7fffffff;
          return;
        }
      }
    }

    // ---- stage 2: column cosine terms for the 7 shifts k*-3..k*+3 (SC.cpp:123-144, 69-90) ----
    // lane = entry column j; shifted by k it lands on query column c = (j + k) % 60
    wave_lds_fence();  // the similarity terms overwrite the key images read by stage 1
#pragma unroll
    for (int b = 0; b < B; b++) {
      const int ks = kstar[b];
      double e[NR];
#pragma unroll
      for (int i = 0; i < 5; i++) {
        e[4 * i + 0] = ecol[b][i].x; e[4 * i + 1] = ecol[b][i].y;
        e[4 * i + 2] = ecol[b][i].z; e[4 * i + 3] = ecol[b][i].w;
      }
      const double n2 = en2[b];
      double *simp = reinterpret_cast<double *>(wsm + b * ENT_SIZE + ENT_SIM);
      int *misc = reinterpret_cast<int *>(wsm + b * ENT_SIZE + ENT_MISC);
#pragma unroll
      for (int t = 0; t < 7; t++) {
        int k = ks + t - 3;
        k += (k < 0) ? NS : 0;
        k -= (k >= NS) ? NS : 0;
        int c = cl + k;
        c -= (c >= NS) ? NS : 0;
        const double2 *qp = reinterpret_cast<const double2 *>(smem + OFF_QIMG + c * Q_COL_STRIDE);
        // Eigen's redux order (SSE2 build): term r goes to accumulator r % 4, dot = (a0 + a2) + (a1 + a3)
        double da[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int i = 0; i < 10; i++) {
          double2 q2 = qp[i];
          da[2 * (i & 1)] = fma(q2.x, e[2 * i + 0], da[2 * (i & 1)]);  // fp32 x fp32 exact in fp64: fma == mul + add
          da[2 * (i & 1) + 1] = fma(q2.y, e[2 * i + 1], da[2 * (i & 1) + 1]);
        }
        double dot = (da[0] + da[2]) + (da[1] + da[3]);
        if constexpr (SO != dev::SO_SSE2) {  // (fp32 x fp32 products are exact in fp64: fused or not is the same number)
          const double *qd = reinterpret_cast<const double *>(qp);
          dot = dev::redux_prod<SO, NR>([&](int r) { return qd[r]; }, [&](int r) { return e[r]; });
        }
        const double n1 = qn1[c];
        const bool valid = (lane < NS) && !((n1 == 0.0) | (n2 == 0.0));  // SC.cpp:78
        const double s = dot / (n1 * n2);                                  // SC.cpp:81
        if (lane < NS) simp[t * NS + c] = valid ? s : 0.0;
        const int ne = __popcll(__ballot(valid));
        if (lane == 0) misc[t] = ne;
      }
      if (lane == 0) misc[7] = ks;
    }
    wave_lds_fence();

    // ---- stage 3: sequential column sum (SC.cpp:83,87-88), lane = (entry b, shift t) ----
    const int bb = lane >> 3, tt = lane & 7;
    double bd = INFINITY;
    int bk = 0x7fffffff;
    if (bb < B && tt < 7) {
      const double2 *sp = reinterpret_cast<const double2 *>(wsm + bb * ENT_SIZE + ENT_SIM + tt * (NS * 8));
      const int *misc = reinterpret_cast<const int *>(wsm + bb * ENT_SIZE + ENT_MISC);
      double s = 0.0;
#pragma unroll 1
      for (int c0 = 0; c0 < NS / 2; c0 += 6) {
#pragma unroll
        for (int cc = 0; cc < 6; cc++) {
          const double2 v = sp[c0 + cc];
          s = s + v.x;
          s = s + v.y;
        }
      }
      const int ne = misc[tt];
      const double d = 1.0 - s / (double)ne;  // 0/0 -> NaN when no effective column
      int k = misc[7] + tt - 3;
      k += (k < 0) ? NS : 0;
      k -= (k >= NS) ? NS : 0;
      if (d < kBig) {  // SC.cpp:134,139: strict `<` against the 1e7 init; NaN fails
        bd = d;
        bk = k;
      }
    }
    // window values are evaluated in ascending shift VALUE (SC.cpp:130) with strict `<`:
    // the winner is the minimum under (dist, shift value)
#pragma unroll
    for (int off = 1; off <= 4; off <<= 1) {
      double od = __shfl_xor(bd, off);
      int ok = __shfl_xor(bk, off);
      if (hit_before(od, ok, bd, bk)) {
        bd = od;
        bk = ok;
      }
    }
    if (bd == INFINITY) {  // SC.cpp:133-134 initial values survive
      bd = kBig;
      bk = 0;
    }

    bd_out = bd;
    bk_out = bk;
}


// ------------------------------------------------------------------------------------------
// The same pair function in two phases, one entry per wavefront (sc_rescore_kernel):
//   phase A  alignment k* (fast fp32 form with exact fallback, see above) + the fp32 preview of the 7 window
//            distances: a value pv with |pv - exact distance| <= kPreviewMargin -- cheap (~40 % of a full
//            evaluation), and good for BOTH directions: pv - margin prunes, and the k-th smallest pv + margin over
//            any set of entries is an upper bound of the final k-th best exact distance;
//   phase B  the exact fp64 window evaluation (stages 2 and 3 of pair_group) for a known k*.
// phase_a returns k*; pv = +inf when no shift of the window has an effective column (never a hit), NaN when no
// preview is possible (norms outside [1e-15, 1e15], non-finite data): such an entry must go through phase B.
// ------------------------------------------------------------------------------------------
template <int SO = dev::SO_SSE2>
__device__ __forceinline__ int phase_a(const char *smem, char *wsm, int lane, const EntryRegs &er, int off_preview,
                                       float e1, float &pv) {
  const int cl = lane < NS ? lane : 0;
  const int kk = lane < NS ? lane : NS - 1;
  const double *v1 = reinterpret_cast<const double *>(smem + OFF_QV1);
  wave_lds_fence();
  const double ev = er.v;
  const float4 (&ecol)[5] = er.ecol;
  const double n2 = er.n2;
  int kstar = 0;
  bool need_exact = true;
  {
    const float vf = (float)ev;
    if (lane < NS) {
#pragma unroll
      for (int j = 0; j < 4; j++) {
        float *img = reinterpret_cast<float *>(wsm + fimg_base(j));
        if (lane >= j) img[lane - j] = vf;
        img[lane + NS - j] = vf;
        if (lane < j) img[lane + 2 * NS - j] = vf;
      }
    }
    const float e2 = wave_sum_f32(lane < NS ? vf * vf : 0.0f);
    wave_lds_fence();
    const int j = (-kk) & 3;
    const float4 *yp = reinterpret_cast<const float4 *>(wsm + fimg_base(j) + (NS - kk - j) * 4);
    const float4 *xp = reinterpret_cast<const float4 *>(smem + off_preview + QP_V1F);
    float2p a0 = {0.0f, 0.0f}, a1 = {0.0f, 0.0f};
#pragma unroll 5
    for (int i = 0; i < NS / 4; i++) {
      const float4 x = xp[i], y = yp[i];
      const float2p d0 = float2p{x.x, x.y} - float2p{y.x, y.y}, d1 = float2p{x.z, x.w} - float2p{y.z, y.w};
      a0 = __builtin_elementwise_fma(d0, d0, a0);
      a1 = __builtin_elementwise_fma(d1, d1, a1);
    }
    const float D = (a0[0] + a0[1]) + (a1[0] + a1[1]);
    const float dmin = wave_min_f32(lane < NS ? D : INFINITY);
    const float thr = dmin + 2.0f * kFastAlignEps * (e1 + e2);
    const unsigned long long nan_bal = __ballot(lane < NS && !(D == D));
    const unsigned long long cand = __ballot(lane < NS && D <= thr);
    if (!nan_bal && dmin < 1e12f && thr < 3.0e38f && __popcll(cand) == 1) {
      kstar = __ffsll((long long)cand) - 1;
      need_exact = false;
    }
    wave_lds_fence();
  }
  if (need_exact) kstar = align_exact<SO>(v1, wsm, lane, ev);
  // fp32 preview of the window distances (see PREVIEW above), arranged for few instructions: the query's unit
  // columns are 0 for an empty column and r2 is 0 for an empty entry column, so no per-lane validity select is
  // needed; the effective-column counts come from the two 60-bit column masks on the scalar unit; the seven lane
  // sums are taken by ONE halving butterfly (lane L ends up with the sum of shift L & 7) instead of seven reductions
  pv = __builtin_nanf("");
  if (*reinterpret_cast<const int *>(smem + off_preview + QP_FLAG)) {  // wave-uniform
    const bool n2_ok = (n2 == 0.0) || (n2 >= 1e-15 && n2 <= 1e15);
    if (!__ballot(!n2_ok && lane < NS)) {
      const float r2 = (n2 == 0.0 || lane >= NS) ? 0.0f : (float)(1.0 / n2);
      const unsigned long long me = __ballot(lane < NS && n2 != 0.0);
      const unsigned long long mq = *reinterpret_cast<const unsigned long long *>(smem + off_preview + QP_MASK);
      float p[8];
      int k0 = kstar - 3;
      k0 += (k0 < 0) ? NS : 0;
      int k = k0;
      int c = cl + k;
      c -= (c >= NS) ? NS : 0;
#pragma unroll
      for (int t = 0; t < 7; t++) {
        const float4 *qp = reinterpret_cast<const float4 *>(smem + off_preview + c * QP_COL_STRIDE);
        float2p d0 = {0.0f, 0.0f}, d1 = {0.0f, 0.0f};
#pragma unroll
        for (int i = 0; i < 5; i++) {
          const float4 q4 = qp[i];
          d0 = __builtin_elementwise_fma(float2p{q4.x, q4.y}, float2p{ecol[i].x, ecol[i].y}, d0);
          d1 = __builtin_elementwise_fma(float2p{q4.z, q4.w}, float2p{ecol[i].z, ecol[i].w}, d1);
        }
        p[t] = ((d0[0] + d0[1]) + (d1[0] + d1[1])) * r2;
        c = (c + 1 == NS) ? 0 : c + 1;
      }
      p[7] = 0.0f;
      // halving butterfly: after the three steps lane L holds shift (L & 7) summed over its group of 8 lanes
      const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4;
      float q4v[4];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const float keep = b0 ? p[2 * i + 1] : p[2 * i], send = b0 ? p[2 * i] : p[2 * i + 1];
        q4v[i] = keep + __shfl_xor(send, 1);
      }
      float q2v[2];
#pragma unroll
      for (int i = 0; i < 2; i++) {
        const float keep = b1 ? q4v[2 * i + 1] : q4v[2 * i], send = b1 ? q4v[2 * i] : q4v[2 * i + 1];
        q2v[i] = keep + __shfl_xor(send, 2);
      }
      float sv;
      {
        const float keep = b2 ? q2v[1] : q2v[0], send = b2 ? q2v[0] : q2v[1];
        sv = keep + __shfl_xor(send, 4);
      }
      sv += __shfl_xor(sv, 8);
      sv += __shfl_xor(sv, 16);
      sv += __shfl_xor(sv, 32);
      // lane L: shift index (b0, b1, b2) -> t = (L & 1) + 2 * ((L >> 1) & 1) + 4 * ((L >> 2) & 1) = L & 7
      float best = INFINITY;
      bool broken = false;
      k = k0;
#pragma unroll
      for (int t = 0; t < 7; t++) {
        const float st = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sv), t));
        // entry column j meets query column (j + k) % 60: bit j of the query mask rotated right by k (scalar unit)
        const unsigned long long rq = ((mq >> k) | (mq << (NS - k))) & ((1ull << NS) - 1ull);
        const int ne = __popcll(rq & me);
        k = (k + 1 == NS) ? 0 : k + 1;
        if (ne != 0) {  // uniform; ne == 0: no effective column at this shift, ignored like SC.cpp:87-88,134
          const float d = 1.0f - st * __builtin_amdgcn_rcpf((float)ne);
          if (!(d == d)) broken = true;  // non-finite data
          best = fminf(best, d);
        }
      }
      if (!broken) pv = best;  // +inf: no effective column at any shift of the window
    }
  }
  return kstar;
}

// phase B: distDirectSC over the window of a known k* (stages 2 and 3 of pair_group, B = 1).  Every lane returns
// (bd, bk) = (distance, shift), {1e7, 0} when no shift of the window has an effective column.
template <int SO = dev::SO_SSE2>
__device__ __forceinline__ void phase_b(const char *smem, char *wsm, int lane, const EntryRegs &er, int ks, double &bd_out,
                                        int &bk_out) {
  const int cl = lane < NS ? lane : 0;
  const double *qn1 = reinterpret_cast<const double *>(smem + OFF_QN1);
  wave_lds_fence();
  double e[NR];
#pragma unroll
  for (int i = 0; i < 5; i++) {
    const float4 v = er.ecol[i];
    e[4 * i + 0] = v.x; e[4 * i + 1] = v.y; e[4 * i + 2] = v.z; e[4 * i + 3] = v.w;
  }
  const double n2 = er.n2;
  double *simp = reinterpret_cast<double *>(wsm + ENT_SIM);
  int *misc = reinterpret_cast<int *>(wsm + ENT_MISC);
#pragma unroll
  for (int t = 0; t < 7; t++) {
    int k = ks + t - 3;
    k += (k < 0) ? NS : 0;
    k -= (k >= NS) ? NS : 0;
    int c = cl + k;
    c -= (c >= NS) ? NS : 0;
    const double2 *qp = reinterpret_cast<const double2 *>(smem + OFF_QIMG + c * Q_COL_STRIDE);
    double da[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int i = 0; i < 10; i++) {
      const double2 q2 = qp[i];
      da[2 * (i & 1)] = fma(q2.x, e[2 * i + 0], da[2 * (i & 1)]);
      da[2 * (i & 1) + 1] = fma(q2.y, e[2 * i + 1], da[2 * (i & 1) + 1]);
    }
    double dot = (da[0] + da[2]) + (da[1] + da[3]);
    if constexpr (SO != dev::SO_SSE2) {
      const double *qd = reinterpret_cast<const double *>(qp);
      dot = dev::redux_prod<SO, NR>([&](int r) { return qd[r]; }, [&](int r) { return e[r]; });
    }
    const double n1 = qn1[c];
    const bool valid = (lane < NS) && !((n1 == 0.0) | (n2 == 0.0));
    const double s = dot / (n1 * n2);
    if (lane < NS) simp[t * NS + c] = valid ? s : 0.0;
    const int ne = __popcll(__ballot(valid));
    if (lane == 0) misc[t] = ne;
  }
  wave_lds_fence();
  const int tt = lane & 7;
  double bd = INFINITY;
  int bk = 0x7fffffff;
  if (lane < 8 && tt < 7) {
    const double2 *sp = reinterpret_cast<const double2 *>(wsm + ENT_SIM + tt * (NS * 8));
    double s = 0.0;
#pragma unroll 1
    for (int c0 = 0; c0 < NS / 2; c0 += 6) {
#pragma unroll
      for (int cc = 0; cc < 6; cc++) {
        const double2 v = sp[c0 + cc];
        s = s + v.x;
        s = s + v.y;
      }
    }
    const int ne = misc[tt];
    const double d = 1.0 - s / (double)ne;
    int k = ks + tt - 3;
    k += (k < 0) ? NS : 0;
    k -= (k >= NS) ? NS : 0;
    if (d < kBig) {
      bd = d;
      bk = k;
    }
  }
#pragma unroll
  for (int off = 1; off <= 4; off <<= 1) {
    const double od = __shfl_xor(bd, off);
    const int ok = __shfl_xor(bk, off);
    if (hit_before(od, ok, bd, bk)) {
      bd = od;
      bk = ok;
    }
  }
  bd = __shfl(bd, 0);
  bk = __shfl(bk, 0);
  if (bd == INFINITY) {
    bd = kBig;
    bk = 0;
  }
  bd_out = bd;
  bk_out = bk;
}

// query -> LDS (once per block): fp64 image (column stride Q_COL_STRIDE), norms, sector key
// off_preview >= 0: also the fp32 unit-column image of the pruning preview at smem + off_preview
__device__ __forceinline__ void load_query_to_lds(const QueryView &q, int qi, char *smem, int tid, int nthreads,
                                                  int off_preview = -1) {
  // One pass, four elements (one float4 of a column: 20 % 4 == 0) per thread, every load of a thread issued before the
  // first use: element-wise 4-byte loads in two loops kept ONE load in flight per thread -- 47 k cycles per query in the
  // re-scoring kernel, 10 % of its time (RSX_RESCORE_PROF), for 15 KB of data.
  const float4 *qd4 = reinterpret_cast<const float4 *>(q.desc + (int64_t)qi * DS);
  const double *qn = q.norm + (int64_t)qi * NS, *qv = q.vkey + (int64_t)qi * NS;
  double kn = 0.0, kv = 0.0;
  if (tid < NS) {
    kn = qn[tid];
    kv = qv[tid];
  }
  for (int t = tid; t < DS / 4; t += nthreads) {
    const int c = t / (NR / 4), r0 = (t - c * (NR / 4)) * 4;
    const float4 x = qd4[t];
    const double n1 = off_preview >= 0 ? qn[c] : 1.0;
    const float xe[4] = {x.x, x.y, x.z, x.w};
    double *img = reinterpret_cast<double *>(smem + OFF_QIMG + c * Q_COL_STRIDE + r0 * 8);
#pragma unroll
    for (int e = 0; e < 4; e++) img[e] = (double)xe[e];
    if (off_preview >= 0) {  // unit columns in fp32 (0 for an empty column)
      float4 u;
      u.x = (n1 == 0.0) ? 0.0f : (float)((double)xe[0] / n1);
      u.y = (n1 == 0.0) ? 0.0f : (float)((double)xe[1] / n1);
      u.z = (n1 == 0.0) ? 0.0f : (float)((double)xe[2] / n1);
      u.w = (n1 == 0.0) ? 0.0f : (float)((double)xe[3] / n1);
      *reinterpret_cast<float4 *>(smem + off_preview + (c * NR + r0) * 4) = u;
    }
  }
  if (off_preview >= 0) {
    const bool bad = tid < NS && (kn != 0.0) && !(kn >= 1e-15 && kn <= 1e15);  // also NaN
    // (called by whole waves) any bad column in this wave's share clears the flag; the caller zero-fills it first
    if (__ballot(bad)) *reinterpret_cast<int *>(smem + off_preview + QP_FLAG) = 0;
    if (tid < NS) reinterpret_cast<float *>(smem + off_preview + QP_V1F)[tid] = (float)kv;
    if (tid < 64) {  // (whole first wave) column mask of the query
      const unsigned long long m = __ballot(tid < NS && kn != 0.0);
      if (tid == 0) *reinterpret_cast<unsigned long long *>(smem + off_preview + QP_MASK) = m;
    }
  }
  if (tid < NS) {
    reinterpret_cast<double *>(smem + OFF_QN1)[tid] = kn;
    reinterpret_cast<double *>(smem + OFF_QV1)[tid] = kv;
  }
}


// W = waves per SIMD the register allocator must leave room for (<= what the LDS footprint allows)
template <int B, int W, int SO>
__global__ __launch_bounds__(256, W) void sc_pair_kernel(PairArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  char *wsm = smem + OFF_WAVES + wave * PairLds<B>::WAVE_SIZE;
  const int qi = blockIdx.y;
  const int slot = blockIdx.x * 4 + wave;
  const int nwaves = gridDim.x * 4;

  load_query_to_lds(a.q, qi, smem, threadIdx.x, 256);
  __syncthreads();

  int64_t n_elig = a.n_eligible;
  if (a.q_elig) {
    int64_t e = a.q_elig[qi];
    n_elig = e < n_elig ? e : n_elig;
  }

  // per-wave sorted top-k, one record per lane
  double ld = INFINITY;
  int li = 0x7fffffff, ls = 0;

  const int32_t *gath = a.gather;
  const int64_t n_items = a.n_items;
  const int64_t ngroups = (n_items + B - 1) / B;
  for (int64_t g = slot; g < ngroups; g += nwaves) {
    // ---- stage 0: issue the entry loads (column cl of each entry stays in registers as fp32
    //      until stage 2; the sector key goes to LDS twice for the rotated reads of stage 1) ----
    int64_t eslot[B];
    bool evalid[B];
#pragma unroll
    for (int b = 0; b < B; b++) {
      int64_t item = g * B + b;
      evalid[b] = item < n_items;
      int64_t it = evalid[b] ? item : (n_items - 1);
      eslot[b] = gath ? (int64_t)gath[it] : (a.first + it);
    }
    double bd;
    int bk;
    pair_group<B, false, false, SO>(a.db, smem, wsm, lane, eslot, bd, bk);

    // ---- outputs ----
#pragma unroll
    for (int b = 0; b < B; b++) {
      const double dist = __shfl(bd, b * 8);
      const int shift = __shfl(bk, b * 8);
      if (!evalid[b]) continue;
      const int64_t item = g * B + b;
      if (a.out_dist && lane == 0) {
        a.out_dist[(int64_t)qi * n_items + item] = dist;
        a.out_shift[(int64_t)qi * n_items + item] = shift;
      }
      if (a.partial) {
        const int64_t gidx = a.db.idx_base + eslot[b] * a.db.idx_stride;
        if (gidx < n_elig && dist < kBig) {  // SC.cpp:388: must beat the 1e7 init
          topk_insert(ld, li, ls, lane, a.k, dist, (int)gidx, shift);
        }
      }
    }
  }

  if (a.partial && lane < a.k) {
    rsx_sc_hit h;
    h.dist = ld;
    h.index = li;
    h.shift = ls;
    a.partial[((int64_t)qi * a.nslots + slot) * a.k + lane] = h;
  }
}

// ------------------------------------------------------------------------------------------
// sc_pair2_kernel: the exact path in TOP-K form (no out_dist: the filter is off or the problem is small), one entry per
// wavefront and iteration, in the two phases of the re-scoring kernel: phase A = fp32 fast alignment (exact fp64 fallback
// when more than one shift survives its error band) + fp32 preview of the 7 window distances; phase B = the exact fp64
// window evaluation, only when the preview cannot exclude the entry from THIS wave's top-k (an entry that cannot enter the
// top-k of the wave that scores it cannot enter the merged one).  Every record returned is produced by phase B, i.e. by the
// same arithmetic as pair_group: results are byte-identical to sc_pair_kernel (tests/test_gpu_sc.py, test_gpu_sc_filter.py
// compare the two paths).  Per entry ~250 fp32 instructions instead of ~450 fp64 ones.
// LDS: [query fp64 image, norms, key | 4 x per-wave region | fp32 preview image of the query]
// ------------------------------------------------------------------------------------------
constexpr int PAIR2_OFF_QP32 = OFF_WAVES + 4 * ENT_SIZE;
constexpr int PAIR2_LDS = PAIR2_OFF_QP32 + QP_SIZE;

template <int W, int SO>
__global__ __launch_bounds__(256, W) void sc_pair2_kernel(PairArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  char *wsm = smem + OFF_WAVES + wave * ENT_SIZE;
  const int qi = blockIdx.y;
  const int slot = blockIdx.x * 4 + wave;
  const int nwaves = gridDim.x * 4;

  if (threadIdx.x == 0) *reinterpret_cast<int *>(smem + PAIR2_OFF_QP32 + QP_FLAG) = 1;
  __syncthreads();
  load_query_to_lds(a.q, qi, smem, threadIdx.x, 256, PAIR2_OFF_QP32);
  __syncthreads();
  const float v1f = lane < NS ? reinterpret_cast<const float *>(smem + PAIR2_OFF_QP32 + QP_V1F)[lane] : 0.0f;
  const float e1 = wave_sum_f32(v1f * v1f);

  int64_t n_elig = a.n_eligible;
  if (a.q_elig) {
    const int64_t e = a.q_elig[qi];
    n_elig = e < n_elig ? e : n_elig;
  }
  double ld = INFINITY;  // per-wave sorted top-k, one record per lane
  int li = 0x7fffffff, ls = 0;
  const int32_t *gath = a.gather;
  EntryRegs cur;
  Touch tch;
  for (int64_t g = slot; g < a.n_items; g += nwaves) {
    const int64_t eslot = gath ? (int64_t)gath[g] : (a.first + g);
    const int64_t gidx = a.db.idx_base + eslot * a.db.idx_stride;
    if (gidx >= n_elig) continue;  // (wave-uniform) never a hit
    if (g + nwaves < a.n_items) touch_entry(a.db, gath ? (int64_t)gath[g + nwaves] : (a.first + g + nwaves), lane, tch);
    load_entry(a.db, eslot, lane, cur);
    float pv;
    const int ks = phase_a<SO>(smem, wsm, lane, cur, PAIR2_OFF_QP32, e1, pv);
    touch_keep(tch);  // phase A has waited for cur's registers, which were requested after the touch
    const double kth = __shfl(ld, a.k - 1);  // +inf until the wave holds k hits
    if ((pv == pv) && (double)pv - (double)kPreviewMargin > kth) continue;  // exact >= pv - margin > the wave's k-th best
    double bd;
    int bk;
    phase_b<SO>(smem, wsm, lane, cur, ks, bd, bk);
    if (bd < kBig) topk_insert(ld, li, ls, lane, a.k, bd, (int)gidx, bk);  // SC.cpp:388: must beat the 1e7 init
  }
  if (lane < a.k) {
    rsx_sc_hit h;
    h.dist = ld;
    h.index = li;
    h.shift = ls;
    a.partial[((int64_t)qi * a.nslots + slot) * a.k + lane] = h;
  }
}

// ------------------------------------------------------------------------------------------
// bounds delivered as column blocks (one per filter shard, each [rows][block_ld]) -> one row-major matrix
// lb[q][c] = blocks[c / block_ld][q0 + q][c % block_ld].  block_ld is a multiple of 32: 8 elements stay inside one block
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sc_gather_bounds_kernel(const lb_t *__restrict__ blocks, int64_t block_ld,
                                                               int64_t block_stride, int64_t q0, lb_t *__restrict__ lb,
                                                               int64_t ld) {
  const int64_t c = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
  if (c >= ld) return;
  const int64_t q = blockIdx.y, b = c / block_ld, j = c - b * block_ld;
  const uint4 v = *reinterpret_cast<const uint4 *>(blocks + b * block_stride + (q0 + q) * block_ld + j);
  *reinterpret_cast<uint4 *>(lb + q * ld + c) = v;
}

// ------------------------------------------------------------------------------------------
// merge: per query, k rounds of "smallest record strictly after the previous pick"
// records are unique in (dist,index) except the padding {inf, INT_MAX} / {1e7,0,0}
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void sc_merge_kernel(const rsx_sc_hit *__restrict__ parts, int32_t nparts,
                                                      int64_t part_stride, int64_t q_stride, int32_t per_part,
                                                      int32_t k, rsx_sc_hit *__restrict__ out) {
  // parts: record (p, q, i) at parts[p*part_stride + q*q_stride + i], i < per_part
  const int q = blockIdx.x;
  const int lane = threadIdx.x;
  double pd = -INFINITY;
  int pi = -1;
  const int64_t total = (int64_t)nparts * per_part;
  for (int r = 0; r < k; r++) {
    double bd = INFINITY;
    int bi = 0x7fffffff, bs = 0;
    for (int64_t t = lane; t < total; t += 64) {
      const int p = (int)(t / per_part), i = (int)(t % per_part);
      rsx_sc_hit h = parts[(int64_t)p * part_stride + (int64_t)q * q_stride + i];
      // padding ({inf,..} from the pair kernel, {1e7,0,0} from merged lists) and anything that
      // could not beat the 1e7 init (SC.cpp:388) is never a hit
      const bool pad = !(h.dist < kBig);
      if (pad) continue;
      if (hit_before(pd, pi, h.dist, h.index) && hit_before(h.dist, h.index, bd, bi)) {
        bd = h.dist; bi = h.index; bs = h.shift;
      }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      double od = __shfl_xor(bd, off);
      int oi = __shfl_xor(bi, off), os = __shfl_xor(bs, off);
      if (hit_before(od, oi, bd, bi)) {
        bd = od; bi = oi; bs = os;
      }
    }
    if (lane == 0) {
      rsx_sc_hit h;
      if (bd == INFINITY) {
        h.dist = kBig; h.index = 0; h.shift = 0;  // SC.cpp:362-364
      } else {
        h.dist = bd; h.index = bi; h.shift = bs;
      }
      out[(int64_t)q * k + r] = h;
    }
    pd = bd;
    pi = bi;
  }
}

// ------------------------------------------------------------------------------------------
// knn over ring keys (candidate stage): one 1024-thread block
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void sc_knn_kernel(const float *__restrict__ rkeys, int64_t n,
                                                      const float *__restrict__ qkey, int32_t k,
                                                      float *__restrict__ dist_ws, int32_t *__restrict__ out_idx,
                                                      float *__restrict__ out_dist, int32_t *__restrict__ out_found) {
  __shared__ float sq[NR];
  __shared__ float rd[16];
  __shared__ int ri[16];
  __shared__ float pick_d;
  __shared__ int pick_i;
  if (threadIdx.x < NR) sq[threadIdx.x] = qkey[threadIdx.x];
  __syncthreads();
  for (int64_t i = threadIdx.x; i < n; i += 1024) {
    const float4 *p = reinterpret_cast<const float4 *>(rkeys + i * NR);
    float result = 0.0f;
#pragma unroll
    for (int g = 0; g < 5; g++) {  // NF.hpp:391-401: 4 at a time, left-to-right sum, no contraction
      float4 v = p[g];
      float d0 = __fsub_rn(sq[4 * g + 0], v.x), d1 = __fsub_rn(sq[4 * g + 1], v.y);
      float d2 = __fsub_rn(sq[4 * g + 2], v.z), d3 = __fsub_rn(sq[4 * g + 3], v.w);
      float t = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(d0, d0), __fmul_rn(d1, d1)), __fmul_rn(d2, d2)),
                          __fmul_rn(d3, d3));
      result = __fadd_rn(result, t);
    }
    dist_ws[i] = result;
  }
  __syncthreads();
  float pd = -1.0f;
  int pi = -1;
  int found = 0;
  for (int r = 0; r < k; r++) {
    float bd = INFINITY;
    int bi = 0x7fffffff;
    for (int64_t i = threadIdx.x; i < n; i += 1024) {
      float d = dist_ws[i];
      bool after = (d > pd) || (d == pd && (int)i > pi);
      if (after && ((d < bd) || (d == bd && (int)i < bi))) {
        bd = d;
        bi = (int)i;
      }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      float od = __shfl_xor(bd, off);
      int oi = __shfl_xor(bi, off);
      if ((od < bd) || (od == bd && oi < bi)) {
        bd = od; bi = oi;
      }
    }
    if ((threadIdx.x & 63) == 0) {
      rd[threadIdx.x >> 6] = bd;
      ri[threadIdx.x >> 6] = bi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      float d = rd[0];
      int ix = ri[0];
      for (int w = 1; w < 16; w++)
        if ((rd[w] < d) || (rd[w] == d && ri[w] < ix)) {
          d = rd[w]; ix = ri[w];
        }
      pick_d = d;
      pick_i = ix;
    }
    __syncthreads();
    pd = pick_d;
    pi = pick_i;
    if (pi != 0x7fffffff) found++;
    if (threadIdx.x == 0) {
      out_idx[r] = (pi == 0x7fffffff) ? 0 : pi;  // SC.cpp:367 zero-initialised slots
      out_dist[r] = pd;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) out_found[0] = found;
}


// ------------------------------------------------------------------------------------------
// sc_rescore_kernel: exact re-scoring behind the MFMA filter (sc_filter.hip / sc_spec.hip), one RS_WAVES-wave workgroup
// per query.  The query's candidates arrive as a short list of (bound, slot) records -- the entries
// with the smallest filter bounds -- and are scored in rounds of ascending bound: after every round
// the workgroup merges its per-wave top-k lists into tau (the k-th best exact distance so far) and
// the next round only scores entries with bound - eps <= tau.  It stops as soon as the next bound
// range cannot reach the top-k; entries beyond the short list (bound >= t_cap) are only scanned when
// tau still admits them.  Output: the final top-k, sorted by (dist, global index), padded {1e7,0,0}.
// ------------------------------------------------------------------------------------------
constexpr int RS_CAND_CAP = 1024;  // candidates of one round (or chunk of a round) in LDS: slot | k* << 26, and their previews
static_assert(RESCORE_SHORTLIST_CAP % RS_CAND_CAP == 0, "rounds are processed in chunks of RS_CAND_CAP list entries");
constexpr int RS_SLOT_BITS = 26;   // local slots < 2^26 (64 M entries per shard) when the two-phase scoring is used
#ifndef RSX_RESCORE_PREVIEW
#define RSX_RESCORE_PREVIEW 1
#endif
constexpr bool kRescorePreview = RSX_RESCORE_PREVIEW != 0;

template <int B, int RS_WAVES>
struct RescoreLds {
  static constexpr int OFF_CAND = OFF_WAVES + RS_WAVES * B * ENT_SIZE;
  static constexpr int OFF_PV = OFF_CAND + RS_CAND_CAP * 4;                        // int32 candidate slots (| k* << 26 after phase A)
  static constexpr int OFF_XCH = OFF_PV + RS_CAND_CAP * 4;                         // fp32 previews of the candidates
  static constexpr int OFF_MISC = OFF_XCH + RS_WAVES * RSX_SC_MAX_TOPK * 16;       // per-wave top-k lists
  static constexpr int OFF_QP32 = OFF_MISC + 64;                                    // fp32 preview image of the query
  static constexpr int SIZE = OFF_QP32 + QP_SIZE;
};

struct RescoreArgs {
  DbView db;
  QueryView q;
  const lb_t *lb;  // filter bounds [nq][ld_lb] (only read past the short list)
  int64_t ld_lb, n_items, n_eligible;
  const int64_t *q_elig;
  const RescoreEntry *slist;  // [nq][RS_CAND_CAP]
  const int32_t *sl_cnt;      // [nq]
  const float *thr;           // [nq][RESCORE_THR_STRIDE]: round edges t_0 <= t_1 <= ... (the last one is t_cap), then the
                              // number of short-list entries below each edge as int32
  rsx_sc_hit *out;            // [nq][k]
  const rsx_sc_hit *tau_src;  // optional [nq][k]: a top-k over MORE than this shard (its k-th distance bounds tau)
  const rsx_sc_hit *seed;     // optional [nq][k]: hits this shard already found in an earlier stage
  double eps;
  int32_t k;
  int32_t round_begin, round_end;  // rounds [begin, end) of the short list; end > RESCORE_NUM_THR: also the rest
  unsigned long long *stats;       // optional (bench instrumentation): [0] += candidates scored (phase A or full), [1] += queries
                                   // that scored any, [2] += exact window evaluations (phase B; two-phase scoring only)
  int32_t two_phase;               // B == 1: alignment + fp32 preview of every candidate first, exact evaluation of the few left
  const WindowPreview *win;        // [nq][WINDOW_P] records of sc_window.hip (sc_rescore_wave_kernel only); stats[3] += records used,
                                   // stats[11] += exact alignments
};

// k-th smallest valid record (by (dist, index)) of the nrec records in xch, by RANK COUNTING on one wave:
// every lane takes a record and counts the valid records before it (nrec broadcast LDS reads, no
// dependent cross-lane steps -- the first version selected the minimum k times with a 6-step shuffle
// reduction each, ~20 k cycles per merge, which made every re-scoring round cost as much as 3 pair
// evaluations).  Records are distinct under the order (every entry is scored once).  When out != nullptr
// the k smallest are written to out[0..k) in order, padded with {1e7,0,0} (SC.cpp:362-364).  Returns the
// k-th distance or +inf.
__device__ __forceinline__ double wave_select_kth(const rsx_sc_hit *xch, int nrec, int k, int lane, rsx_sc_hit *out) {
  double kth = INFINITY;
  int nvalid = 0;
  for (int base = 0; base < nrec; base += 64) {  // uniform trip count
    const int t = base + lane;
    rsx_sc_hit me;
    me.dist = kBig; me.index = 0; me.shift = 0;
    if (t < nrec) me = xch[t];
    const bool valid = me.dist < kBig;  // padding is {>= 1e7, ...}
    int rank = 0;
    int j = 0;
    for (; j + 8 <= nrec; j += 8) {  // 8 reads in flight per step (one LDS round trip per 8 records, not per record)
      rsx_sc_hit o[8];
#pragma unroll
      for (int u = 0; u < 8; u++) o[u] = xch[j + u];  // same address in every lane: broadcast reads
#pragma unroll
      for (int u = 0; u < 8; u++) rank += ((o[u].dist < kBig) && hit_before(o[u].dist, o[u].index, me.dist, me.index)) ? 1 : 0;
    }
    for (; j < nrec; j++) {
      const rsx_sc_hit o = xch[j];
      rank += ((o.dist < kBig) && hit_before(o.dist, o.index, me.dist, me.index)) ? 1 : 0;
    }
    if (out && valid && rank < k) out[rank] = me;
    const unsigned long long is_kth = __ballot(valid && rank == k - 1);
    if (is_kth) kth = __shfl(me.dist, __ffsll((long long)is_kth) - 1);
    nvalid += __popcll(__ballot(valid));
  }
  if (out) {
    rsx_sc_hit pad;
    pad.dist = kBig; pad.index = 0; pad.shift = 0;
    for (int r = nvalid + lane; r < k; r += 64) out[r] = pad;
  }
  return kth;
}

// RS_WAVES waves per workgroup, W = waves per SIMD the register allocator leaves room for (several
// workgroups share a CU so that one query's barriers / merges hide behind another's scoring)
template <int B, int RS_WAVES, int W, bool TWO, int SO>
__global__ __launch_bounds__(RS_WAVES * 64, W) void sc_rescore_kernel(RescoreArgs a) {
  static_assert(!TWO || B == 1, "two-phase scoring handles one entry per wavefront");
  if (a.stats) a.stats += (blockIdx.x % RESCORE_STAT_COPIES) * RESCORE_STAT_WORDS;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using L = RescoreLds<B, RS_WAVES>;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int qi = blockIdx.x;
  char *wsm = smem + OFF_WAVES + wave * (B * ENT_SIZE);
  int32_t *cand = reinterpret_cast<int32_t *>(smem + L::OFF_CAND);
  rsx_sc_hit *xch = reinterpret_cast<rsx_sc_hit *>(smem + L::OFF_XCH);
  int *s_ncand = reinterpret_cast<int *>(smem + L::OFF_MISC);
  double *s_tau = reinterpret_cast<double *>(smem + L::OFF_MISC + 8);

  // the query image is only brought into LDS when the first candidate shows up: in the later stages
  // of a sharded search most queries have nothing left to score
  bool query_loaded = false;
  bool scored_any = false;
  if (threadIdx.x == 0) {
    *s_ncand = 0;
    *s_tau = INFINITY;
  }
  __syncthreads();

  int64_t n_elig = a.n_eligible;
  if (a.q_elig) {
    const int64_t e = a.q_elig[qi];
    n_elig = e < n_elig ? e : n_elig;
  }
  // local slots [0, n_rows) are the eligible ones
  int64_t n_rows = 0;
  if (n_elig > a.db.idx_base) {
    n_rows = (n_elig - a.db.idx_base + a.db.idx_stride - 1) / a.db.idx_stride;
    n_rows = n_rows < a.n_items ? n_rows : a.n_items;
  }

  double ld = INFINITY;  // per-wave sorted top-k, one record per lane; lives across rounds
  int li = 0x7fffffff, ls = 0;
  // tau never exceeds the k-th distance of a top-k that covers more than this shard (multi-GPU
  // stage 2): any upper bound of the global k-th best distance prunes correctly
  double tau_init = INFINITY;
  if (a.tau_src) {
    const double d = a.tau_src[(int64_t)qi * a.k + (a.k - 1)].dist;
    if (d < kBig) tau_init = d;
  }
  double tau = tau_init;
  if (a.seed && wave == 0 && lane < a.k) {  // sorted, padded {1e7,0,0} at the end
    const rsx_sc_hit h = a.seed[(int64_t)qi * a.k + lane];
    if (h.dist < kBig) {
      ld = h.dist; li = h.index; ls = h.shift;
    }
  }

  // region timing for profiling runs (wave 0 of every workgroup, s_memtime): [4] query load, [5] phase A, [6] preview
  // merge, [7] phase B, [8] exact merge, [9] gather / append, [10] whole workgroup
  long long t_last = 0;
  const bool timing = a.stats != nullptr && wave == 0;
  auto tick = [&](int slot) {
    if (!timing) return;
    const long long t = clock64();
    if (slot >= 0 && lane == 0) atomicAdd(a.stats + slot, (unsigned long long)(t - t_last));
    t_last = t;
  };
  const long long t_begin = timing ? clock64() : 0;
  tick(-1);
  // score cand[0..ncand) (all waves), then refresh tau
  auto score_and_merge = [&](int ncand) {
    tick(9);
    if (ncand == 0) return;  // (uniform) nothing selected: lists and tau are unchanged
    if (a.stats && threadIdx.x == 0) {
      atomicAdd(a.stats, (unsigned long long)ncand);
      if (!scored_any) atomicAdd(a.stats + 1, 1ull);
    }
    if (!query_loaded) {
      if (threadIdx.x == 0) *reinterpret_cast<int *>(smem + L::OFF_QP32 + QP_FLAG) = 1;
      __syncthreads();
      load_query_to_lds(a.q, qi, smem, threadIdx.x, RS_WAVES * 64, L::OFF_QP32);
      __syncthreads();
      query_loaded = true;
      tick(4);
    }
    scored_any = true;
    // ||v1||^2 in fp32 for the error bound of the fast alignment (every wave for itself)
    const float v1f = lane < NS ? reinterpret_cast<const float *>(smem + L::OFF_QP32 + QP_V1F)[lane] : 0.0f;
    const float e1 = wave_sum_f32(v1f * v1f);
    const int ngroups = (ncand + B - 1) / B;
    if constexpr (TWO) {
      {
        // ---- phase A: k* and the fp32 preview pv of every candidate; the k-th smallest (pv + margin) over this
        // round's candidates and the exact hits known so far is an upper bound of the final k-th best distance ----
        float *pvs = reinterpret_cast<float *>(smem + L::OFF_PV);
        double ud = ld;  // this wave's exact hits so far + its candidates' preview upper bounds
        int ui = li, us = ls;
        // (requesting the next candidate's registers one candidate ahead was tried: 48 more live registers, one
        // workgroup per CU fewer, 3 % slower -- four waves per SIMD already hide the entry loads)
        EntryRegs cur;
        Touch tch;
        for (int g = wave; g < ncand; g += RS_WAVES) {
          const int64_t slot = cand[g];
          const int64_t gidx = a.db.idx_base + slot * a.db.idx_stride;
          float pv = INFINITY;  // ineligible: never scored
          int ks = 0;
          if (gidx < n_elig) {
            if (g + RS_WAVES < ncand) touch_entry(a.db, cand[g + RS_WAVES], lane, tch);
            load_entry(a.db, slot, lane, cur);
            ks = phase_a<SO>(smem, wsm, lane, cur, L::OFF_QP32, e1, pv);
            touch_keep(tch);  // phase A has waited for cur's registers, which were requested after the touch
            const bool usable = (pv == pv) && fabsf(pv) < 3.0e38f;  // NaN / -inf: no preview; +inf: never a hit
            if (usable) topk_insert(ud, ui, us, lane, a.k, (double)pv + (double)kPreviewMargin, (int)gidx, 0);
            if (!(pv == pv)) pv = -INFINITY;  // no preview: phase B must look at it
          }
          if (lane == 0) {
            pvs[g] = pv;
            cand[g] = (int32_t)slot | (ks << RS_SLOT_BITS);
          }
        }
        tick(5);
        if (lane < a.k) {
          rsx_sc_hit h;
          h.dist = ud; h.index = ui; h.shift = us;
          xch[wave * a.k + lane] = h;
        }
        __syncthreads();
        if (wave == 0) {
          const double t = wave_select_kth(xch, RS_WAVES * a.k, a.k, lane, nullptr);
          if (lane == 0) *s_tau = t < tau ? t : tau;  // tau: the exact bound carried in
        }
        __syncthreads();
        const double tau_ub = *s_tau;
        tick(6);
        // (each wave using only its OWN k-th smallest upper bound saves the two barriers but quadruples the exact
        // evaluations: 43 instead of 11 per query, 7.5 instead of 4.2 ms per step)
        // ---- phase B: exact evaluation of the candidates the previews cannot exclude ----
        for (int g = wave; g < ncand; g += RS_WAVES) {
          const float pv = pvs[g];
          // this wave's own k-th exact distance tightens the test as it goes
          const double kth_local = __shfl(ld, a.k - 1);
          const double t_eff = kth_local < tau_ub ? kth_local : tau_ub;
          if ((double)pv - (double)kPreviewMargin > t_eff) continue;  // exact >= pv - margin > k-th best: not in the top-k
          const int32_t packed = cand[g];
          const int64_t slot = packed & ((1 << RS_SLOT_BITS) - 1);
          load_entry(a.db, slot, lane, cur);
          double bd;
          int bk;
          phase_b<SO>(smem, wsm, lane, cur, (packed >> RS_SLOT_BITS) & 63, bd, bk);
          if (a.stats && lane == 0) atomicAdd(a.stats + 2, 1ull);
          const int64_t gidx = a.db.idx_base + slot * a.db.idx_stride;
          if (bd < kBig) topk_insert(ld, li, ls, lane, a.k, bd, (int)gidx, bk);
        }
        __syncthreads();  // phase B of every wave is done with pvs / cand / xch before they are re-used
        tick(7);
      }
    }
    if constexpr (!TWO) {
    for (int g = wave; g < ngroups; g += RS_WAVES) {
      int64_t eslot[B];
      bool evalid[B];
#pragma unroll
      for (int b = 0; b < B; b++) {
        const int item = g * B + b;
        evalid[b] = item < ncand;
        eslot[b] = cand[evalid[b] ? item : (ncand - 1)];
      }
      double bd;
      int bk;
      // tau is finite from the second round on: candidates then leave after the alignment + fp32 preview unless
      // they can still reach the top-k
      pair_group<B, kRescorePreview, B == 1, SO>(a.db, smem, wsm, lane, eslot, bd, bk, tau, L::OFF_QP32, nullptr, e1);
#pragma unroll
      for (int b = 0; b < B; b++) {
        const double dist = __shfl(bd, b * 8);
        const int shift = __shfl(bk, b * 8);
        if (!evalid[b]) continue;
        const int64_t gidx = a.db.idx_base + eslot[b] * a.db.idx_stride;
        if (gidx < n_elig && dist < kBig) topk_insert(ld, li, ls, lane, a.k, dist, (int)gidx, shift);
      }
    }
    }
    if (lane < a.k) {
      rsx_sc_hit h;
      h.dist = ld; h.index = li; h.shift = ls;
      xch[wave * a.k + lane] = h;
    }
    __syncthreads();
    if (wave == 0) {
      const double t = wave_select_kth(xch, RS_WAVES * a.k, a.k, lane, nullptr);
      if (lane == 0) {
        *s_tau = t < tau_init ? t : tau_init;
        *s_ncand = 0;
      }
    }
    __syncthreads();
    tau = *s_tau;
    tick(8);
  };

  // block-wide append of this thread's candidate (wave ballot + one LDS atomic per wave)
  auto append = [&](bool pass, int32_t slot) {
    const unsigned long long bal = __ballot(pass);
    int wbase = 0;
    if (lane == 0 && bal) wbase = atomicAdd(s_ncand, __popcll(bal));
    wbase = __shfl(wbase, 0);
    if (pass) cand[wbase + __popcll(bal & ((1ull << lane) - 1ull))] = slot;
  };

  // ---- rounds over the short list ----
  const int sl_cnt = a.sl_cnt[qi];
  const RescoreEntry *sl = a.slist + (int64_t)qi * RESCORE_SHORTLIST_CAP;
  const float *thr = a.thr + (int64_t)qi * RESCORE_THR_STRIDE;
  const int32_t *rcnt = reinterpret_cast<const int32_t *>(thr) + RESCORE_NUM_THR;
  const float t_cap = thr[RESCORE_NUM_THR - 1];
  float lo = a.round_begin > 0 ? thr[a.round_begin - 1] : -INFINITY;
  bool done = false;
  const int r_end = a.round_end < RESCORE_NUM_THR ? a.round_end : RESCORE_NUM_THR;
  for (int r = a.round_begin; r < r_end; r++) {
    const float hi = thr[r];
    if (!(lo < hi)) continue;  // empty range (uniform)
    if ((double)lo - a.eps > tau) {  // (stage 2: the global tau may already exclude everything left)
      done = true;
      break;
    }
    // the short list is ordered by bin and the round edges are bin edges: round r is one contiguous range
    // (bin 0, first in the list, also holds the NaN / -inf "always re-score" bounds)
    const int i0 = r > 0 ? rcnt[r - 1] : 0, i1 = rcnt[r] < sl_cnt ? rcnt[r] : sl_cnt;
    for (int c0 = i0; c0 < i1; c0 += RS_CAND_CAP) {  // (one chunk, unless a round is longer than the LDS candidate list)
      const int c1 = c0 + RS_CAND_CAP < i1 ? c0 + RS_CAND_CAP : i1;
      for (int i = c0 + threadIdx.x; i < c1; i += RS_WAVES * 64) {
        const RescoreEntry e = sl[i];
        append(!((double)e.lb - a.eps > tau), e.slot);
      }
      __syncthreads();
      const int ncand = *s_ncand;
      score_and_merge(ncand);
    }
    lo = hi;
    if ((double)lo - a.eps > tau) {  // every remaining bound is >= lo: nothing can reach the top-k
      done = true;
      break;
    }
  }

  // ---- entries beyond the short list (bound >= t_cap), only while tau admits them ----
  if (!done && a.round_end > RESCORE_NUM_THR && t_cap < INFINITY && !((double)t_cap - a.eps > tau)) {
    const lb_t *row = a.lb + (int64_t)qi * a.ld_lb;
    const bool take_all = (t_cap == -INFINITY);  // empty short list: NaN bounds are here too
    int64_t pos = 0;
    while (pos < n_rows) {
      // gather up to RS_CAND_CAP candidates, 1024 rows at a time
      int ncand = 0;
      while (pos < n_rows && ncand <= RS_CAND_CAP - RS_WAVES * 64) {
        const int64_t i = pos + threadIdx.x;
        bool pass = false;
        if (i < n_rows) {
          const float d = (float)row[i];
          const bool beyond = take_all ? true : (d >= t_cap);
          pass = beyond && (d != INFINITY) && !((double)d - a.eps > tau);
        }
        append(pass, (int32_t)i);
        pos += RS_WAVES * 64;
        __syncthreads();
        ncand = *s_ncand;
        __syncthreads();
      }
      score_and_merge(ncand);
    }
  }

  // ---- output: top-k of everything this workgroup knows ----
  if (!scored_any && a.seed) {  // nothing scored in this stage: the earlier hits are the answer
    if (threadIdx.x < a.k) a.out[(int64_t)qi * a.k + threadIdx.x] = a.seed[(int64_t)qi * a.k + threadIdx.x];
    return;
  }
  if (lane < a.k) {
    rsx_sc_hit h;
    h.dist = ld; h.index = li; h.shift = ls;
    xch[wave * a.k + lane] = h;
  }
  __syncthreads();
  if (wave == 0) wave_select_kth(xch, RS_WAVES * a.k, a.k, lane, a.out + (int64_t)qi * a.k);
  if (timing && lane == 0) atomicAdd(a.stats + 10, (unsigned long long)(clock64() - t_begin));
}

template <int B, int W, int SO>
int launch_pairs_t(const PairArgs &a, int gx, hipStream_t s) {
  static_assert(W <= pair_waves_per_simd<B>(), "LDS footprint does not allow this occupancy");
  static bool attr_set = false;
  const int lds = PairLds<B>::SIZE;
  if (!attr_set) {
    RSX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&sc_pair_kernel<B, W, SO>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_set = true;
  }
  dim3 grid(gx, a.q.nq);
  hipLaunchKernelGGL((sc_pair_kernel<B, W, SO>), grid, dim3(256), lds, s, a);
  RSX_HIP(hipGetLastError());
  return RSX_OK;
}

// kernel variant = entries per wave iteration (B) x register-occupancy target (W waves/SIMD);
// RSX_SC_PAIR_VARIANT="B,W" overrides the tuned default (measured on MI355X)
struct Variant { int b, w; };
Variant pair_variant() {
  static Variant v = [] {
    Variant d{2, 4};
    const char *e = rsx::exp_env("RSX_SC_PAIR_VARIANT");
    int b = 0, w = 0;
    if (e && sscanf(e, "%d,%d", &b, &w) == 2) {
      if ((b == 1 && w == 4) || (b == 2 && (w == 3 || w == 4)) || (b == 4 && w == 2)) d = Variant{b, w};
    }
    return d;
  }();
  return v;
}
int pair_B() { return pair_variant().b; }

int choose_gx(int64_t n_items, int32_t nq) {
  const int kB = pair_B();
  const int64_t ngroups = (n_items + kB - 1) / kB;
  int64_t max_gx = (ngroups + 3) / 4;
  if (max_gx < 1) max_gx = 1;
  int64_t want = (2048 + 4 * (int64_t)nq - 1) / (4 * (int64_t)nq);  // ~2048 waves in flight
  if (want < 1) want = 1;
  return (int)(want < max_gx ? want : max_gx);
}

}  // namespace

const char *pair_kernel_name() { return "sc_pair_kernel"; }

static thread_local PairProfiler *g_prof = nullptr;
void set_pair_profiler(PairProfiler *p) { g_prof = p; }

int pair_num_slots(int64_t n_items, int32_t nq) { return choose_gx(n_items, nq) * 4; }

size_t pair_partial_bytes(int64_t n_items, int32_t nq, int32_t k) {
  return (size_t)pair_num_slots(n_items, nq) * (size_t)nq * (size_t)k * sizeof(rsx_sc_hit);
}

int launch_keys(const float *desc, int64_t n, double *vkey, double *norm, float *rkey, hipStream_t s, int sum_order) {
  if (n <= 0) return RSX_OK;
  RSX_SO_DISPATCH(sum_order, hipLaunchKernelGGL(sc_keys_kernel<SO>, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, desc, n, vkey, norm, rkey));
  RSX_HIP(hipGetLastError());
  return RSX_OK;
}

int launch_build(const void *d_pts, int64_t n_pts, int64_t stride_bytes, double lidar_height,
                 double max_radius, float *out_desc, double *out_vkey, double *out_norm,
                 float *out_rkey, hipStream_t s, int sum_order) {
  RSX_SO_DISPATCH(sum_order, hipLaunchKernelGGL(sc_build_kernel<SO>, dim3(1), dim3(256), 0, s, static_cast<const char *>(d_pts), n_pts,
                                                stride_bytes, lidar_height, max_radius, out_desc, out_vkey, out_norm, out_rkey));
  RSX_HIP(hipGetLastError());
  return RSX_OK;
}

int launch_merge(const rsx_sc_hit *d_parts, int32_t nparts, int32_t nq, int32_t k, rsx_sc_hit *d_out,
                 hipStream_t s) {
  if (nq <= 0) return RSX_OK;
  hipLaunchKernelGGL(sc_merge_kernel, dim3(nq), dim3(64), 0, s, d_parts, nparts, (int64_t)nq * k, (int64_t)k, k, k, d_out);
  RSX_HIP(hipGetLastError());
  return RSX_OK;
}

int launch_gather_bounds(const lb_t *d_blocks, int64_t block_ld, int64_t block_stride, int64_t q0, int32_t nq, lb_t *d_lb,
                         int64_t ld, hipStream_t s) {
  if (nq <= 0 || ld <= 0) return RSX_OK;
  if (block_ld < 32 || block_ld % 32 || ld % 8 || block_stride % 8) return fail(RSX_ERR_BAD_ARG, "bound blocks must be a multiple of 32 columns wide");
  for (int32_t r0 = 0; r0 < nq; r0 += 65535) {  // gridDim.y
    const int32_t rows = nq - r0 < 65535 ? nq - r0 : 65535;
    hipLaunchKernelGGL(sc_gather_bounds_kernel, dim3((unsigned)((ld / 8 + 255) / 256), (unsigned)rows), dim3(256), 0, s, d_blocks,
                       block_ld, block_stride, q0 + r0, d_lb + (int64_t)r0 * ld, ld);
    RSX_HIP(hipGetLastError());
  }
  return RSX_OK;
}

int launch_pairs(const DbView &db, const QueryView &q, const int32_t *gather, int64_t first,
                 int64_t n_items, int64_t n_eligible, const int64_t *q_elig, double *out_dist,
                 int32_t *out_shift, rsx_sc_hit *d_partial, rsx_sc_hit *d_topk, int32_t k,
                 hipStream_t s) {
  if (q.nq <= 0) return RSX_OK;
  if (k < 0 || k > RSX_SC_MAX_TOPK) return fail(RSX_ERR_BAD_ARG, "k=%d out of range [0,%d]", k, RSX_SC_MAX_TOPK);
  if (n_items <= 0) {
    // nothing to score: result is all padding
    if (d_topk && k > 0) {
      // a merge over zero parts writes the padding
      hipLaunchKernelGGL(sc_merge_kernel, dim3(q.nq), dim3(64), 0, s, (const rsx_sc_hit *)nullptr, 0,
                         (int64_t)0, (int64_t)k, k, k, d_topk);
      RSX_HIP(hipGetLastError());
    }
    return RSX_OK;
  }
  PairArgs a;
  a.db = db;
  a.q = q;
  a.gather = gather;
  a.first = first;
  a.n_items = n_items;
  a.n_eligible = n_eligible < 0 ? INT64_MAX : n_eligible;
  a.q_elig = q_elig;
  a.out_dist = out_dist;
  a.out_shift = out_shift;
  const int gx = choose_gx(n_items, q.nq);
  a.partial = (d_topk && k > 0) ? d_partial : nullptr;
  a.k = k;
  a.nslots = gx * 4;
  PairProfiler *pp = (g_prof && g_prof->on && g_prof->ev && g_prof->used < PairProfiler::kMax) ? g_prof : nullptr;
  if (pp) RSX_HIP(hipEventRecord(pp->ev[2 * pp->used], s));
  const Variant var = pair_variant();
  static const bool one_phase = rsx::exp_env("RSX_SC_PAIR_ONE_PHASE") != nullptr;  // experiments: the round-1/2 kernel for top-k too
  if (a.partial && !out_dist && !one_phase) {
    static_assert(PAIR2_LDS <= 48 * 1024, "within the default dynamic LDS limit: no per-device opt-in needed");
    RSX_SO_DISPATCH(db.sum_order, hipLaunchKernelGGL((sc_pair2_kernel<4, SO>), dim3(gx, q.nq), dim3(256), PAIR2_LDS, s, a));
    RSX_HIP(hipGetLastError());
  } else if (db.sum_order != dev::SO_SSE2) {  // (the other shapes are tuning experiments of the default order)
    int st = RSX_OK;
    if (var.b != 2) return fail(RSX_ERR_BAD_ARG, "RSX_SC_PAIR_VARIANT applies to the default summation order only");
    RSX_SO_DISPATCH(db.sum_order, st = (launch_pairs_t<2, 4, SO>(a, gx, s)));
    RSX_TRY(st);
  } else if (var.b == 1) RSX_TRY((launch_pairs_t<1, 4, dev::SO_SSE2>(a, gx, s)));
  else if (var.b == 2 && var.w == 3) RSX_TRY((launch_pairs_t<2, 3, dev::SO_SSE2>(a, gx, s)));
  else if (var.b == 2) RSX_TRY((launch_pairs_t<2, 4, dev::SO_SSE2>(a, gx, s)));
  else RSX_TRY((launch_pairs_t<4, 2, dev::SO_SSE2>(a, gx, s)));
  if (pp) {
    RSX_HIP(hipEventRecord(pp->ev[2 * pp->used + 1], s));
    pp->used++;
  }
  if (a.partial) {
    // partial layout [q][slot][k]: one "part" per slot with part_stride = k, query stride nslots*k
    hipLaunchKernelGGL(sc_merge_kernel, dim3(q.nq), dim3(64), 0, s, (const rsx_sc_hit *)d_partial, 1,
                       (int64_t)0, (int64_t)a.nslots * k, a.nslots * k, k, d_topk);
    RSX_HIP(hipGetLastError());
  }
  return RSX_OK;
}

// ------------------------------------------------------------------------------------------
// sc_walk_kernel: exact re-scoring behind the filter, ONE WAVE per query (single-GPU path).  The short
// list arrives ordered by bound (by histogram bin, sc_select_kernel), so the wave walks it in ascending
// order: every candidate sees the tau (k-th best exact distance) of everything before it -- no rounds,
// no barriers, no cross-wave merges -- and from the k-th hit on most candidates leave pair_group after
// the alignment + fp32 preview.  It stops at the first bin whose lower edge minus eps exceeds tau.
// Entries beyond the short list (bound >= t_cap) are scanned from the bounds row only while tau still
// admits them.  Output: the final top-k, sorted by (dist, global index), padded {1e7,0,0}.
// LDS per wave: query images (fp64 + fp32 preview) + one pair_group region = 19.9 KB -> 8 waves per CU,
// 2 per SIMD, 256 registers each (the preview does not spill).
// ------------------------------------------------------------------------------------------
struct WalkLds {
  static constexpr int OFF_QP32 = OFF_WAVES + ENT_SIZE;
  static constexpr int SIZE = OFF_QP32 + QP_SIZE;
};

__device__ __forceinline__ float bound_bin_lo(float lb) {  // lower edge of the 2048-bin histogram bin of lb
  if (!(lb > 0.0f)) return -INFINITY;
  const float x = lb * 2048.0f;
  return x >= 2047.0f ? 2047.0f / 2048.0f : floorf(x) / 2048.0f;
}

template <int SO>
__global__ __launch_bounds__(64, 2) void sc_walk_kernel(RescoreArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x;
  const int qi = blockIdx.x;
  char *wsm = smem + OFF_WAVES;

  int64_t n_elig = a.n_eligible;
  if (a.q_elig) {
    const int64_t e = a.q_elig[qi];
    n_elig = e < n_elig ? e : n_elig;
  }
  int64_t n_rows = 0;  // local slots [0, n_rows) are the eligible ones
  if (n_elig > a.db.idx_base) {
    n_rows = (n_elig - a.db.idx_base + a.db.idx_stride - 1) / a.db.idx_stride;
    n_rows = n_rows < a.n_items ? n_rows : a.n_items;
  }

  double ld = INFINITY;  // sorted top-k, one record per lane
  int li = 0x7fffffff, ls = 0;
  double tau = INFINITY;
  bool query_loaded = false;

  auto ensure_query = [&]() {
    if (!query_loaded) {
      if (lane == 0) *reinterpret_cast<int *>(smem + WalkLds::OFF_QP32 + QP_FLAG) = 1;
      load_query_to_lds(a.q, qi, smem, lane, 64, WalkLds::OFF_QP32);
      query_loaded = true;
    }
  };
  // score one entry whose registers are already loaded (or being loaded)
  auto score_regs = [&](int32_t slot, const EntryRegs &er) {
    ensure_query();
    const int64_t eslot[1] = {slot};
    double bd;
    int bk;
    pair_group<1, true, false, SO>(a.db, smem, wsm, lane, eslot, bd, bk, a.round_begin ? INFINITY : tau, WalkLds::OFF_QP32, &er);
    const double dist = __shfl(bd, 0);
    const int shift = __shfl(bk, 0);
    const int64_t gidx = a.db.idx_base + (int64_t)slot * a.db.idx_stride;
    if (gidx < n_elig && dist < kBig) {
      topk_insert(ld, li, ls, lane, a.k, dist, (int)gidx, shift);
      tau = __shfl(ld, a.k - 1);  // +inf until k hits exist
    }
  };
  auto score = [&](int32_t slot) {
    EntryRegs er;
    load_entry(a.db, slot, lane, er);
    score_regs(slot, er);
  };

  // ---- the short list, ascending bin order; the registers of entry i+1 are requested before entry i is
  // scored, so their global-memory latency hides behind its arithmetic ----
  const int sl_cnt = a.sl_cnt[qi];
  const RescoreEntry *sl = a.slist + (int64_t)qi * RESCORE_SHORTLIST_CAP;
  const float t_cap = a.thr[(int64_t)qi * RESCORE_THR_STRIDE + (RESCORE_NUM_THR - 1)];
  bool done = false;
  for (int base = 0; base < sl_cnt && !done; base += 64) {
    const int n_here = (sl_cnt - base < 64) ? (sl_cnt - base) : 64;
    RescoreEntry mine;  // one coalesced read per 64 candidates
    mine.lb = INFINITY;
    mine.slot = 0;
    if (lane < n_here) mine = sl[base + lane];
    EntryRegs cur, nxt;
    load_entry(a.db, __shfl(mine.slot, 0), lane, cur);
    for (int i = 0; i < n_here; i++) {
      const float lb = __shfl(mine.lb, i);
      const int32_t slot = __shfl(mine.slot, i);
      if (i + 1 < n_here) load_entry(a.db, __shfl(mine.slot, i + 1), lane, nxt);
      // every later entry sits in this bin or a higher one
      if ((double)bound_bin_lo(lb) - a.eps > tau) {
        done = true;
        break;
      }
      if (!((double)lb - a.eps > tau)) score_regs(slot, cur);  // NaN / -inf bounds: always scored
      cur = nxt;
    }
  }

  // ---- entries beyond the short list (bound >= t_cap), only while tau admits them ----
  if (!done && t_cap < INFINITY && !((double)t_cap - a.eps > tau)) {
    const lb_t *row = a.lb + (int64_t)qi * a.ld_lb;
    const bool take_all = (t_cap == -INFINITY);  // empty short list: NaN bounds are here too
    for (int64_t pos = 0; pos < n_rows; pos += 64) {
      const int64_t i = pos + lane;
      const float d = (i < n_rows) ? (float)row[i] : INFINITY;
      const bool beyond = take_all ? true : (d >= t_cap);
      unsigned long long bal = __ballot((i < n_rows) && beyond && (d != INFINITY));
      while (bal) {
        const int l = __ffsll((long long)bal) - 1;
        bal &= bal - 1;
        const float dl = __shfl(d, l);
        if ((double)dl - a.eps > tau) continue;
        score((int32_t)(pos + l));
      }
    }
  }

  if (lane < a.k) {
    rsx_sc_hit h;
    if (ld == INFINITY) {
      h.dist = kBig; h.index = 0; h.shift = 0;
    } else {
      h.dist = ld; h.index = li; h.shift = ls;
    }
    a.out[(int64_t)qi * a.k + lane] = h;
  }
}

int launch_walk(const DbView &db, const QueryView &q, const lb_t *lb, int64_t ld_lb, int64_t n_items,
                int64_t n_eligible, const int64_t *q_elig, const RescoreEntry *slist, const int32_t *sl_cnt,
                const float *thr, double eps, rsx_sc_hit *d_out, int32_t k, hipStream_t s) {
  if (q.nq <= 0) return RSX_OK;
  if (k < 1 || k > RSX_SC_MAX_TOPK) return fail(RSX_ERR_BAD_ARG, "k=%d out of range [1,%d]", k, RSX_SC_MAX_TOPK);
  if (db.sum_order != dev::SO_SSE2) return fail(RSX_ERR_BAD_ARG, "the walk kernel (an experiment) exists for the default summation order only");
  static bool attr_set = false;
  if (!attr_set) {
    RSX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&sc_walk_kernel<dev::SO_SSE2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                WalkLds::SIZE));
    attr_set = true;
  }
  RescoreArgs a;
  a.db = db;
  a.q = q;
  a.lb = lb;
  a.ld_lb = ld_lb;
  a.n_items = n_items;
  a.n_eligible = n_eligible < 0 ? INT64_MAX : n_eligible;
  a.q_elig = q_elig;
  a.slist = slist;
  a.sl_cnt = sl_cnt;
  a.thr = thr;
  a.out = d_out;
  a.tau_src = nullptr;
  a.seed = nullptr;
  a.eps = eps;
  a.k = k;
  a.round_begin = rsx::exp_env("RSX_WALK_NOPREVIEW") ? 1 : 0;  // experiment: disable the pruning preview
  a.round_end = RESCORE_ALL_ROUNDS;
  hipLaunchKernelGGL(sc_walk_kernel<dev::SO_SSE2>, dim3(q.nq), dim3(64), WalkLds::SIZE, s, a);
  RSX_HIP(hipGetLastError());
  return RSX_OK;
}

// ------------------------------------------------------------------------------------------
// sc_rescore_wave_kernel: exact re-scoring behind the filter AND the window kernel (sc_window.hip), ONE WAVE per
// query.  With the alignment k* and a preview of the pair distance already there for the head of the short list, what is
// left per query is: pick the k-th smallest preview upper bound (an upper bound of the final k-th best distance), and
// evaluate exactly -- phase B, ~13 entries per query -- the few entries whose preview lower bound does not exceed it, in
// ascending order of that lower bound so that the exact k-th best takes over as early as possible.  No barriers, no
// merges between waves, no imbalance between them (the 4-wave workgroup of sc_rescore_kernel spent 64 % of its wave
// cycles waiting once its phase A was gone), and 9.0 KiB of LDS per query (the query image stays in fp32 and is
// converted on the fly) instead of 40 KiB: 12 queries per CU in flight instead of 4.
// Entries without k* (alignment not unique within the window kernel's error bound, or beyond its WINDOW_P positions)
// get the exact fp64 alignment first; their preview is still a valid LOWER bound (minimum over the union of the candidate
// windows), so most of them are never touched.  Same stage interface as sc_rescore_kernel (rounds, tau_src, seed).
// ------------------------------------------------------------------------------------------
constexpr int RW_CH = (WINDOW_P + 63) / 64;  // window records per lane


#ifndef RW_OCC
#define RW_OCC 3  // waves per SIMD the register budget is set for (168 VGPRs, 13 spilled; 4 = 128 VGPRs with 98 spilled: 0.42 against 0.24 ms)
#endif
template <int SO>
__global__ __launch_bounds__(64, RW_OCC) void sc_rescore_wave_kernel(RescoreArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if (a.stats) a.stats += (blockIdx.x % RESCORE_STAT_COPIES) * RESCORE_STAT_WORDS;
  const int lane = threadIdx.x;
  const int qi = blockIdx.x;
  char *wsm = smem + WaveLds::OFF_ENT;

  int64_t n_elig = a.n_eligible;
  if (a.q_elig) {
    const int64_t e = a.q_elig[qi];
    n_elig = e < n_elig ? e : n_elig;
  }
  int64_t n_rows = 0;  // local slots [0, n_rows) are the eligible ones
  if (n_elig > a.db.idx_base) {
    n_rows = (n_elig - a.db.idx_base + a.db.idx_stride - 1) / a.db.idx_stride;
    n_rows = n_rows < a.n_items ? n_rows : a.n_items;
  }

  double ld = INFINITY;  // sorted top-k of exact hits, one record per lane
  int li = 0x7fffffff, ls = 0;
  double tau_init = INFINITY;  // (multi-GPU stage 2) the k-th distance of a top-k over more than this shard
  if (a.tau_src) {
    const double d = a.tau_src[(int64_t)qi * a.k + (a.k - 1)].dist;
    if (d < kBig) tau_init = d;
  }
  if (a.seed && lane < a.k) {  // this shard's hits of an earlier stage: sorted, padded {1e7,0,0}
    const rsx_sc_hit h = a.seed[(int64_t)qi * a.k + lane];
    if (h.dist < kBig) {
      ld = h.dist; li = h.index; ls = h.shift;
    }
  }
  auto kth_of = [&](double list) {
    const double t = __shfl(list, a.k - 1);
    return t < tau_init ? t : tau_init;
  };
  double tau = kth_of(ld);
  bool query_loaded = false;
  unsigned n_exact = 0, n_looked = 0, n_aligned = 0, n_shifts = 0;  // wave-uniform counters (stats)
  unsigned nl_lane = 0, n_windowed = 0;               // per-lane counters, summed over the wave at the end

  // score one entry exactly (ks < 0: the alignment is not known yet)
  // region cycles for RSX_RESCORE_PROF (experiments builds): [4] query load, [5] records + tau_ub, [6] picking the next
  // survivor, [7] phase B, [8] exact alignments, [9] waiting for the entry's registers, [10] the whole wave
#ifdef RSX_EXPERIMENTS
  const bool timing = a.stats != nullptr;
#else
  constexpr bool timing = false;
#endif
  long long tacc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  long long t_mark = timing ? clock64() : 0;
  const long long t_begin = t_mark;
  auto lap = [&](int slot_) {
    if (!timing) return;
    const long long t = clock64();
    tacc[slot_] += t - t_mark;
    t_mark = t;
  };
  auto eval = [&](int64_t slot, int ksm, const EntryRegs &er) {
    if (timing) {
      lap(2);  // [6] since the last lap: picking / requesting
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      lap(5);  // [9]
    }
    int ks = ksm;
    unsigned tmask = 0x7fu;
    if (ksm >= 0) {  // a window record: k* and the shifts of its window that can be the minimum
      ks = ksm & 63;
      const unsigned m7 = ((unsigned)ksm >> 8) & 0x7fu;
      if (m7) tmask = m7;
    }
    if (!query_loaded) {
      const float4 *src = reinterpret_cast<const float4 *>(a.q.desc + (int64_t)qi * DS);
      float4 *dst = reinterpret_cast<float4 *>(smem + WaveLds::OFF_QF32);
      float4 x[5];
#pragma unroll
      for (int i = 0; i < 5; i++) x[i] = (lane + 64 * i < DS / 4) ? src[lane + 64 * i] : float4{0.f, 0.f, 0.f, 0.f};
      const double kn = lane < NS ? a.q.norm[(int64_t)qi * NS + lane] : 0.0;
      const double kv = lane < NS ? a.q.vkey[(int64_t)qi * NS + lane] : 0.0;
#pragma unroll
      for (int i = 0; i < 5; i++)
        if (lane + 64 * i < DS / 4) dst[lane + 64 * i] = x[i];
      if (lane < NS) {
        reinterpret_cast<double *>(smem + WaveLds::OFF_QN1)[lane] = kn;
        reinterpret_cast<double *>(smem + WaveLds::OFF_QV1)[lane] = kv;
      }
      query_loaded = true;
      wave_lds_fence();
      lap(0);  // [4]
    }
    if (ks < 0) {
      ks = align_exact<SO>(reinterpret_cast<const double *>(smem + WaveLds::OFF_QV1), wsm, lane, er.v);
      n_aligned++;
      lap(4);  // [8]
    }
    double bd;
    int bk;
    phase_b32<SO>(smem, wsm, lane, er, ks, tmask, bd, bk);
    if (timing) {
      asm volatile("" : "+v"(bd));
      lap(3);  // [7]
    }
    n_exact++;
    n_shifts += (unsigned)__builtin_popcount(tmask);
    const int64_t gidx = a.db.idx_base + slot * a.db.idx_stride;
    if (gidx < n_elig && bd < kBig) {
      topk_insert(ld, li, ls, lane, a.k, bd, (int)gidx, bk);
      tau = kth_of(ld);
    }
  };

  // ---- this launch's part of the short list: rounds [round_begin, round_end) = positions [i0, i1) ----
  const int sl_cnt = a.sl_cnt[qi];
  const RescoreEntry *sl = a.slist + (int64_t)qi * RESCORE_SHORTLIST_CAP;
  const float *thr = a.thr + (int64_t)qi * RESCORE_THR_STRIDE;
  const int32_t *rcnt = reinterpret_cast<const int32_t *>(thr) + RESCORE_NUM_THR;
  const float t_cap = thr[RESCORE_NUM_THR - 1];
  const int r_end = a.round_end < RESCORE_NUM_THR ? a.round_end : RESCORE_NUM_THR;
  auto upto = [&](int r) {  // list positions below round edge r - 1
    if (r <= 0) return 0;
    const int c = rcnt[r - 1];
    return c < sl_cnt ? c : sl_cnt;
  };
  const int i0 = upto(a.round_begin), i1 = upto(r_end);
  if (timing) {
    int keep = i0 + i1;
    asm volatile("" : "+s"(keep));
    lap(7);  // [13] header loads
  }
  bool done = false;  // some list entry's bound already exceeds tau: everything after it does too

  // ---- the head of the list: window records (k*, preview) ----
  int pos_next = i0;
  if (a.win && i0 < WINDOW_P && i0 < i1) {
    const WindowPreview *wp = a.win + (int64_t)qi * WINDOW_P;
    const int pw = i1 < WINDOW_P ? i1 : WINDOW_P;
    // per lane RW_CH records (positions i0 + lane + 64 j); only their lower bounds stay in registers -- the slot and k*
    // of a survivor are read again when its turn comes (one uniform load, a candidate ahead of its use)
    float lo[RW_CH], ub[RW_CH], flb[RW_CH];
#pragma unroll
    for (int j = 0; j < RW_CH; j++) {
      const int pos = i0 + lane + 64 * j;
      lo[j] = INFINITY;
      ub[j] = INFINITY;
      flb[j] = INFINITY;
      if (pos < pw) {  // (chunks past pw cost one compare)
        const RescoreEntry e = sl[pos];
        const WindowPreview w = wp[pos];
        const int64_t gidx = a.db.idx_base + (int64_t)e.slot * a.db.idx_stride;
        if (gidx < n_elig && !((double)e.lb - a.eps > tau)) {
          flb[j] = e.lb;
          if (!(w.pv == w.pv)) {
            lo[j] = -INFINITY;  // no preview (non-finite data, or no record: decided below): must be looked at
          } else {
            lo[j] = w.pv - WINDOW_MARGIN;  // +inf stays +inf: no effective column in the window, never a hit
            if (w.ks >= 0 && w.pv < 3.0e38f) ub[j] = w.pv + WINDOW_MARGIN;
          }
        }
      }
    }
    if (timing) {
      asm volatile("" : "+v"(lo[0]), "+v"(ub[0]));
      lap(8);  // [14] record loads
    }
    // the k-th smallest of {exact hits so far} u {preview upper bounds}: an upper bound of the final k-th best
    double ud = ld;
    int ui = li, us = ls;
    for (int it = 0; it < a.k; it++) {
      float m = ub[0];
#pragma unroll
      for (int j = 1; j < RW_CH; j++) m = fminf(m, ub[j]);
      const float wm = wave_min_f32(m);
      if (!((double)wm < __shfl(ud, a.k - 1))) break;
      const unsigned long long bal = __ballot(m == wm);
      const int src = __ffsll((long long)bal) - 1;
      if (lane == src) {
        bool gone = false;
#pragma unroll
        for (int j = 0; j < RW_CH; j++)
          if (!gone && ub[j] == wm) {
            ub[j] = INFINITY;
            gone = true;
          }
      }
      topk_insert(ud, ui, us, lane, a.k, (double)wm, 0x40000000 + it, 0);  // the index only orders ties
    }
    const double tau_ub = kth_of(ud);
    lap(1);  // [5]
    // the filter bound once more, against the bound the previews give: this is what removes the entries the window
    // kernel left without a record (their bound exceeds ITS k-th smallest upper bound, which is never below this one
    // when this launch starts at the head of the list)
#pragma unroll
    for (int j = 0; j < RW_CH; j++)
      if ((double)flb[j] - a.eps > tau_ub) lo[j] = INFINITY;  // flb = +inf: nothing here
#pragma unroll
    for (int j = 0; j < RW_CH; j++) {  // (stats) entries whose filter bound still admits them / those with a window record
      nl_lane += (lo[j] < INFINITY) ? 1u : 0u;
      n_windowed += (lo[j] > -INFINITY && lo[j] < INFINITY) ? 1u : 0u;
    }
    // Survivors.  A memory round trip costs this kernel ~10 k cycles (random 4.8-KB rows of a 48 MB array: RSX_RESCORE_PROF
    // showed 25 k cycles per survivor when each one's records and registers were requested only when its turn came), so the
    // survivors are first compacted into lanes (64 per round), their slots / k* fetched in ONE parallel round trip, the
    // cache lines of ALL of them requested at once, and only then are they evaluated -- in ascending order of their lower
    // bound, the registers of the next one in flight while the current one is evaluated.
    struct Packed {
      float lo;
      int pos;
    };
    Packed *cbuf = reinterpret_cast<Packed *>(wsm);  // the entry region is free until the first evaluation of a round
    for (;;) {
      const double t_now = tau < tau_ub ? tau : tau_ub;
      int nsurv = 0;
#pragma unroll
      for (int j = 0; j < RW_CH; j++) {
        const bool sv = lo[j] < INFINITY && !((double)lo[j] > t_now);
        const unsigned long long bal = __ballot(sv);
        const int rank = nsurv + __popcll(bal & ((1ull << lane) - 1ull));
        if (sv && rank < 64) {
          cbuf[rank] = Packed{lo[j], i0 + lane + 64 * j};
          lo[j] = INFINITY;  // taken
        }
        nsurv += __popcll(bal);
      }
      if (nsurv == 0) break;  // (uniform)
      nsurv = nsurv < 64 ? nsurv : 64;
      wave_lds_fence();
      float mylo = INFINITY;
      int32_t myslot = 0;
      int myks = -1;
      if (lane < nsurv) {
        const Packed c = cbuf[lane];
        mylo = c.lo;
        myslot = sl[c.pos].slot;
        const int k_ = wp[c.pos].ks;
        myks = k_ >= 0 ? k_ : -1;  // (k* | shift mask << 8) or "alignment unknown"
      }
      wave_lds_fence();
      {
        Touch tch;
        for (int i = 0; i < nsurv; i++) touch_entry(a.db, __shfl(myslot, i), lane, tch);
        touch_wait(tch);  // ONE round trip for all of them (they overlap); from here on the entries come out of the L2
      }
      auto pick = [&](int32_t &slot, int &ks) -> float {  // the smallest lower bound left in this round
        const float wm = wave_min_f32(mylo);
        if (wm == INFINITY) return wm;
        const int src = __ffsll((long long)__ballot(mylo == wm)) - 1;
        slot = __shfl(myslot, src);
        ks = __shfl(myks, src);
        if (lane == src) mylo = INFINITY;
        return wm;
      };
      EntryRegs cur, nxt;
      int32_t cur_slot = 0, nxt_slot = 0;
      int cur_ks = -1, nxt_ks = -1;
      float cur_lo = pick(cur_slot, cur_ks);
      if (cur_lo < INFINITY) load_entry(a.db, cur_slot, lane, cur);
      while (cur_lo < INFINITY) {
        const double t_eff = tau < tau_ub ? tau : tau_ub;
        if ((double)cur_lo > t_eff) break;  // exact >= lower bound > an upper bound of the k-th best; the rest is larger still
        const float nxt_lo = pick(nxt_slot, nxt_ks);
        if (nxt_lo < INFINITY && !((double)nxt_lo > t_eff)) load_entry(a.db, nxt_slot, lane, nxt);
        eval(cur_slot, cur_ks, cur);
        cur_lo = nxt_lo;
        cur_slot = nxt_slot;
        cur_ks = nxt_ks;
        cur = nxt;
      }
    }
    pos_next = pw > i0 ? pw : i0;
  }

  // ---- the rest of this launch's list range, ascending bin order: no preview, exact alignment + phase B ----
  for (int base = pos_next; base < i1 && !done; base += 64) {
    const int n_here = (i1 - base < 64) ? (i1 - base) : 64;
    RescoreEntry mine;
    mine.lb = INFINITY;
    mine.slot = 0;
    if (lane < n_here) mine = sl[base + lane];
    for (int i = 0; i < n_here; i++) {
      const float lb = __shfl(mine.lb, i);
      const int32_t slot = __shfl(mine.slot, i);
      if ((double)bound_bin_lo(lb) - a.eps > tau) {  // every later entry sits in this bin or a higher one
        done = true;
        break;
      }
      if ((double)lb - a.eps > tau) continue;  // (NaN / -inf bounds: always scored)
      const int64_t gidx = a.db.idx_base + (int64_t)slot * a.db.idx_stride;
      if (!(gidx < n_elig)) continue;
      EntryRegs er;
      load_entry(a.db, slot, lane, er);
      n_looked++;
      eval(slot, -1, er);
    }
  }

  // ---- entries beyond the short list (bound >= t_cap), only while tau admits them ----
  if (!done && a.round_end > RESCORE_NUM_THR && t_cap < INFINITY && !((double)t_cap - a.eps > tau)) {
    const lb_t *row = a.lb + (int64_t)qi * a.ld_lb;
    const bool take_all = (t_cap == -INFINITY);  // empty short list: NaN bounds are here too
    for (int64_t pos = 0; pos < n_rows; pos += 64) {
      const int64_t i = pos + lane;
      const float d = (i < n_rows) ? (float)row[i] : INFINITY;
      const bool beyond = take_all ? true : (d >= t_cap);
      unsigned long long bal = __ballot((i < n_rows) && beyond && (d != INFINITY));
      while (bal) {
        const int l = __ffsll((long long)bal) - 1;
        bal &= bal - 1;
        const float dl = __shfl(d, l);
        if ((double)dl - a.eps > tau) continue;
        EntryRegs er;
        load_entry(a.db, pos + l, lane, er);
        n_looked++;
        eval(pos + l, -1, er);
      }
    }
  }

  if (lane < a.k) {
    rsx_sc_hit h;
    if (ld == INFINITY) {
      h.dist = kBig; h.index = 0; h.shift = 0;
    } else {
      h.dist = ld; h.index = li; h.shift = ls;
    }
    a.out[(int64_t)qi * a.k + lane] = h;
  }
  if (a.stats) {
    unsigned lk = nl_lane, wd = n_windowed;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      lk += __shfl_xor(lk, off);
      wd += __shfl_xor(wd, off);
    }
    if (lane == 0) {
      atomicAdd(a.stats, (unsigned long long)(lk + n_looked));
      atomicAdd(a.stats + 3, (unsigned long long)wd);
      atomicAdd(a.stats + 2, (unsigned long long)n_exact);
      atomicAdd(a.stats + 11, (unsigned long long)n_aligned);
      atomicAdd(a.stats + 12, (unsigned long long)n_shifts);
      if (timing) {
        for (int i = 0; i < 6; i++) atomicAdd(a.stats + 4 + i, (unsigned long long)tacc[i]);
        atomicAdd(a.stats + 13, (unsigned long long)tacc[7]);
        atomicAdd(a.stats + 14, (unsigned long long)tacc[8]);
        atomicAdd(a.stats + 10, (unsigned long long)(clock64() - t_begin));
      }
      if (n_exact) atomicAdd(a.stats + 1, 1ull);
    }
  }
}

template <int B, int NW, int W, bool TWO = false, int SO = dev::SO_SSE2>
static int launch_rescore_t(const RescoreArgs &a, hipStream_t s) {
  static bool attr_set = false;
  constexpr int lds = RescoreLds<B, NW>::SIZE;
  if (!attr_set) {
    RSX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&sc_rescore_kernel<B, NW, W, TWO, SO>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_set = true;
  }
  hipLaunchKernelGGL((sc_rescore_kernel<B, NW, W, TWO, SO>), dim3(a.q.nq), dim3(NW * 64), lds, s, a);
  return RSX_OK;
}

int launch_rescore(const DbView &db, const QueryView &q, const lb_t *lb, int64_t ld_lb, int64_t n_items,
                   int64_t n_eligible, const int64_t *q_elig, const RescoreEntry *slist, const int32_t *sl_cnt,
                   const float *thr, double eps, int32_t round_begin, int32_t round_end, const rsx_sc_hit *tau_src,
                   const rsx_sc_hit *seed, rsx_sc_hit *d_out, int32_t k, hipStream_t s, unsigned long long *d_stats,
                   const WindowPreview *win) {
  if (q.nq <= 0) return RSX_OK;
  if (k < 1 || k > RSX_SC_MAX_TOPK) return fail(RSX_ERR_BAD_ARG, "k=%d out of range [1,%d]", k, RSX_SC_MAX_TOPK);
  // workgroup shape (entries per wave iteration, waves, waves/SIMD); RSX_SC_RESCORE_VARIANT picks one:
  // 0: (1, 4, 4) four workgroups per CU (default)   1: (1, 16, 4)   2: (2, 12, 3)   3: (2, 6, 3)   4: (1, 8, 4) two per CU
  // 5: (1, 6, 4).  Measured on the bench (10k DB, 8192 queries, ms per step): 4.86 / 6.9 / 6.8 / 7.7 / 5.34 / 5.75 --
  // four 4-wave workgroups per CU keep more queries in flight, so one query's barriers and merge hide behind
  // the others' scoring
  static const int variant = [] {
    const char *e = rsx::exp_env("RSX_SC_RESCORE_VARIANT");
    return (e && *e) ? atoi(e) : 0;
  }();
  RescoreArgs a;
  a.db = db;
  a.q = q;
  a.lb = lb;
  a.ld_lb = ld_lb;
  a.n_items = n_items;
  a.n_eligible = n_eligible < 0 ? INT64_MAX : n_eligible;
  a.q_elig = q_elig;
  a.slist = slist;
  a.sl_cnt = sl_cnt;
  a.thr = thr;
  a.out = d_out;
  a.tau_src = tau_src;
  a.seed = seed;
  a.eps = eps;
  a.k = k;
  a.round_begin = round_begin;
  a.round_end = round_end;
  a.stats = d_stats;
  a.win = win;
  static const bool no_two_phase = [] {
    const char *e = rsx::exp_env("RSX_SC_TWO_PHASE");  // experiments: 0 = the one-pass scoring of round 1
    return e && e[0] == '0';
  }();
  a.two_phase = (!no_two_phase && variant == 0 && n_items < (1ll << RS_SLOT_BITS)) ? 1 : 0;
  // with the window records of sc_window.hip: one wave per query (RSX_SC_RESCORE=rounds, experiments: the 4-wave workgroup)
  static const bool force_rounds = [] {
    const char *e = rsx::exp_env("RSX_SC_RESCORE");
    return e && e[0] == 'r';
  }();
  if (win && !force_rounds) {
    RSX_SO_DISPATCH(db.sum_order, hipLaunchKernelGGL(sc_rescore_wave_kernel<SO>, dim3(q.nq), dim3(64), WaveLds::SIZE, s, a));
    RSX_HIP(hipGetLastError());
    return RSX_OK;
  }
  if (db.sum_order != dev::SO_SSE2) {  // (the other workgroup shapes are tuning experiments of the default order)
    int st = RSX_OK;
    if (a.two_phase) RSX_SO_DISPATCH(db.sum_order, st = (launch_rescore_t<1, 4, 4, true, SO>(a, s)));
    else RSX_SO_DISPATCH(db.sum_order, st = (launch_rescore_t<1, 4, 4, false, SO>(a, s)));
    RSX_TRY(st);
    RSX_HIP(hipGetLastError());
    return RSX_OK;
  }
  switch (variant) {
    case 1: RSX_TRY((launch_rescore_t<1, 16, 4>(a, s))); break;
    case 2: RSX_TRY((launch_rescore_t<2, 12, 3>(a, s))); break;
    case 3: RSX_TRY((launch_rescore_t<2, 6, 3>(a, s))); break;
    case 4: RSX_TRY((launch_rescore_t<1, 8, 4>(a, s))); break;
    case 5: RSX_TRY((launch_rescore_t<1, 6, 4>(a, s))); break;
    default:
      if (a.two_phase) RSX_TRY((launch_rescore_t<1, 4, 4, true>(a, s)));
      else RSX_TRY((launch_rescore_t<1, 4, 4>(a, s)));
      break;
  }
  RSX_HIP(hipGetLastError());
  return RSX_OK;
}

int launch_knn(const float *rkeys, int64_t n_search, const float *qkey, int32_t k, float *d_dist_ws,
               int32_t *out_idx, float *out_dist, int32_t *out_found, hipStream_t s) {
  hipLaunchKernelGGL(sc_knn_kernel, dim3(1), dim3(1024), 0, s, rkeys, n_search, qkey, k, d_dist_ws, out_idx,
                     out_dist, out_found);
  RSX_HIP(hipGetLastError());
  return RSX_OK;
}

}  // namespace sc
}  // namespace rsx
