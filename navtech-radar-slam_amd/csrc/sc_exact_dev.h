// sc_exact_dev.h -- the exact fp64 side of one (query, entry) pair as inline device functions: the entry's registers, the
// exact sector-key alignment (fastAlignUsingVkey, SC.cpp:93-113), the exact window evaluation for a known alignment
// (distDirectSC over the shifts k* - 3 .. k* + 3, SC.cpp:69-90,123-144) and the sorted top-k list of a wavefront.  They are
// the bodies sc_rescore_wave_kernel (sc_kernels.hip) and sc_q1_kernel (sc_q1.hip) share; moved here verbatim from
// sc_kernels.hip, whose header states the numerics contract (Eigen's redux order, -ffp-contract=off).
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "sc_kernels.h"
#include "sc_entry_dev.h"

namespace rsx {
namespace sc {
namespace {

using dev::wave_lds_fence;

constexpr double kBig = 10000000.0;  // SC.cpp:96,134,362 "init with something large"

// per entry: the sector key twice in a row (vk2[e] = v[e % 60], 120 doubles) for the rotated reads of
// stage 1, in TWO images -- A at element offset 0, B shifted by one element -- so that every lane can
// fetch two consecutive elements with ONE 16-byte-aligned ds_read_b128 (the compiler otherwise pairs
// ds_read_b64s into half-rate ds_read2_b64); then the 7 x 60 similarity terms as sim[t][c].
// The similarity terms ALIAS the key images, which are dead once stage 1 has produced k*.
// Image B starts 72 LDS slots (of 16 B) after image A: with the lane groups of ds_read_b128
// ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ...) the even lanes of a group (image A, 16 B apart per 2 lanes) and
// its odd lanes (image B) then land on 16 different slots mod 16 -- at the natural offset 960 B (60 slots) they
// collided pairwise (SQ_LDS_BANK_CONFLICT = 22 % of the LDS cycles of the re-scoring kernel)
constexpr int ENT_VKEY_A = 0, ENT_VKEY_B = 1152, ENT_SIM = 0, ENT_MISC = 3360;
// 3408 B = 852 dwords = 20 (mod 64): an odd multiple of 4 dwords, so the per-entry blocks land on
// disjoint LDS slots in the ds_read_b128 lane groups of stage 3
constexpr int ENT_SIZE = 3408;

__device__ __forceinline__ bool hit_before(double ad, int ai, double bd, int bi) {
  return (ad < bd) || (ad == bd && ai < bi);
}

// sum of v over the 64 lanes (DPP butterfly inside each row of 16, row broadcasts across rows)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f32(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum_f32(float v) {
  v += dpp_f32<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
  v += dpp_f32<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
  v += dpp_f32<0x141, 0xf>(v);  // row_half_mirror
  v += dpp_f32<0x140, 0xf>(v);  // row_mirror: every lane = its row's sum
  v += dpp_f32<0x142, 0xa>(v);  // row_bcast:15 into rows 1, 3
  v += dpp_f32<0x143, 0xc>(v);  // row_bcast:31 into rows 2, 3
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

__device__ __forceinline__ float wave_min_f32(float v) {
  v = fminf(v, dpp_f32<0xB1, 0xf>(v));
  v = fminf(v, dpp_f32<0x4E, 0xf>(v));
  v = fminf(v, dpp_f32<0x141, 0xf>(v));
  v = fminf(v, dpp_f32<0x140, 0xf>(v));
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
  return fminf(fminf(r0, r1), fminf(r2, r3));
}

// the per-lane registers of one entry (lane = column): loaded by pair_group itself, or ahead of time by a
// caller that overlaps the global-memory latency with the previous entry's arithmetic (sc_walk_kernel)
struct EntryRegs {
  float4 ecol[5];
  double v, n2;
};
__device__ __forceinline__ void load_entry(const DbView &db, int64_t slot, int lane, EntryRegs &r) {
  const int cl = lane < NS ? lane : 0;
  r.v = db.vkey[slot * NS + cl];
  const float4 *src = reinterpret_cast<const float4 *>(db.desc + slot * DS + cl * NR);
#pragma unroll
  for (int i = 0; i < 5; i++) r.ecol[i] = src[i];
  r.n2 = db.norm[slot * NS + cl];
}

// cache touch of an entry this wave will score later: one dword per lane and array (60 lanes x 80 B cover every line of
// the 4800-byte descriptor), results discarded -- brings the lines towards this XCD's L2 without holding the entry's 24
// registers.  The three destination registers belong to the CALLER (struct Touch) and must stay reserved until the loads
// have landed: the compiler does not know that an asm load is still in flight, and a destination it considered dead
// would be handed to the next value and overwritten when the load returns.  touch_keep() after the caller's next wait
// for YOUNGER loads (VMEM returns in order), or touch_wait(), ends the reservation.
struct Touch {
  int d0 = 0, d1 = 0, d2 = 0;
};
__device__ __forceinline__ void touch_entry(const DbView &db, int64_t slot, int lane, Touch &t) {
  const int cl = lane < NS ? lane : 0;
  const float *pd = db.desc + slot * DS + cl * NR;
  const double *pk = db.vkey + slot * NS + cl, *pn = db.norm + slot * NS + cl;
  asm volatile("global_load_dword %0, %3, off\n\tglobal_load_dword %1, %4, off\n\tglobal_load_dword %2, %5, off"
               : "+v"(t.d0), "+v"(t.d1), "+v"(t.d2)
               : "v"(pd), "v"(pk), "v"(pn)
               : "memory");
}
__device__ __forceinline__ void touch_keep(Touch &t) { asm volatile("" : "+v"(t.d0), "+v"(t.d1), "+v"(t.d2)); }
__device__ __forceinline__ void touch_wait(Touch &t) {
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(t.d0), "+v"(t.d1), "+v"(t.d2)::"memory");
}

// exact stage 1 of pair_group (fastAlignUsingVkey, SC.cpp:93-113) for one entry: v1 = the query's sector key (60 doubles
// in LDS), ev = the entry's key, one element per lane; uses the key-image part of the wave's LDS region
template <int SO = dev::SO_SSE2>
__device__ __forceinline__ int align_exact(const double *v1, char *wsm, int lane, double ev) {
  const int kk = lane < NS ? lane : NS - 1;
  double *vka = reinterpret_cast<double *>(wsm + ENT_VKEY_A);
  double *vkb = reinterpret_cast<double *>(wsm + ENT_VKEY_B);
  if (lane < NS) {
    vka[lane] = ev;
    vka[lane + NS] = ev;
    vkb[lane + 1] = ev;
    vkb[lane + NS + 1] = ev;
  }
  wave_lds_fence();
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  const int eoff = (kk & 1) ? (ENT_VKEY_B + (NS + 1 - kk) * 8) : (ENT_VKEY_A + (NS - kk) * 8);
  const double2 *v2 = reinterpret_cast<const double2 *>(wsm + eoff);
  const double2 *v1p = reinterpret_cast<const double2 *>(v1);
#pragma unroll 1
  for (int c0 = 0; c0 < NS / 2; c0 += 6) {
#pragma unroll
    for (int cc = 0; cc < 6; cc++) {
      const double2 x = v1p[c0 + cc], y = v2[c0 + cc];
      const double d0 = x.x - y.x;
      const double dd0 = d0 * d0;
      acc[2 * (cc & 1)] = acc[2 * (cc & 1)] + dd0;
      const double d1 = x.y - y.y;
      const double dd1 = d1 * d1;
      acc[2 * (cc & 1) + 1] = acc[2 * (cc & 1) + 1] + dd1;
    }
  }
  double nrm = sqrt((acc[0] + acc[2]) + (acc[1] + acc[3]));
  if constexpr (SO != dev::SO_SSE2) {  // the reference built with another packet size: the same 60 terms, its order
    const double *v1d = v1, *v2d = reinterpret_cast<const double *>(wsm + eoff);
    auto d = [&](int c) { return v1d[c] - v2d[c]; };
    nrm = sqrt(dev::redux_prod<SO, NS>(d, d));
  }
  const bool ok = (lane < NS) && (nrm < kBig);
  double m = ok ? nrm : INFINITY;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) m = fmin(m, __shfl_xor(m, off));
  const unsigned long long bal = __ballot(ok && nrm == m);
  wave_lds_fence();
  return bal ? (__ffsll((long long)bal) - 1) : 0;
}

// insert (dist, idx, shift) into the wave's sorted top-k list (one record per lane, lanes < k)
__device__ __forceinline__ void topk_insert(double &ld, int &li, int &ls, int lane, int k, double dist, int idx,
                                            int shift) {
  const bool before = (lane < k) && hit_before(ld, li, dist, idx);
  const int pos = __popcll(__ballot(before));
  if (pos < k) {
    double ud = __shfl_up(ld, 1);
    int ui = __shfl_up(li, 1), us = __shfl_up(ls, 1);
    if (lane > pos) {
      ld = ud; li = ui; ls = us;
    } else if (lane == pos) {
      ld = dist; li = idx; ls = shift;
    }
  }
}

// the same insertion with the shift-up by one lane as DPP wave_shr:1 (four v_mov_b32_dpp) instead of four ds_bpermute round
// trips: sc_q1.hip, where the insertions of a round sit on the critical path of the last workgroup
__device__ __forceinline__ void topk_insert_dpp(double &ld, int &li, int &ls, int lane, int k, double dist, int idx, int shift) {
  const bool before = (lane < k) && hit_before(ld, li, dist, idx);
  const int pos = __popcll(__ballot(before));
  if (pos < k) {
    auto shr1 = [](int x) { return __builtin_amdgcn_update_dpp(x, x, 0x138 /* wave_shr:1 */, 0xf, 0xf, false); };
    const long long lb = __double_as_longlong(ld);
    const long long ub = ((long long)shr1((int)(lb >> 32)) << 32) | (unsigned)shr1((int)lb);
    const int ui = shr1(li), us = shr1(ls);
    if (lane > pos) {
      ld = __longlong_as_double(ub); li = ui; ls = us;
    } else if (lane == pos) {
      ld = dist; li = idx; ls = shift;
    }
  }
}

struct WaveLds {
  static constexpr int OFF_QF32 = 0;                  // the query descriptor as it is: [60][20] fp32, column stride 80 B
  static constexpr int OFF_QN1 = DS * 4;              // 4800: column norms (fp64)
  static constexpr int OFF_QV1 = OFF_QN1 + 512;       // 5312: sector key (fp64)
  static constexpr int OFF_ENT = OFF_QV1 + 512;       // 5824: the wave's entry region (key images / similarity terms)
  static constexpr int SIZE = OFF_ENT + ENT_SIZE;     // 9232
};

// phase B on the fp32 query image: the same operations in the same order as phase_b (the conversion float -> double is
// exact), half the LDS bytes per column
// tmask: bit t set = window shift ks - 3 + t is evaluated (wave-uniform; the shifts left out are known to be strictly worse
// than the best one, so the winner under (distance, shift value) is the same)
template <int SO = dev::SO_SSE2>
__device__ __forceinline__ void phase_b32(const char *smem, char *wsm, int lane, const EntryRegs &er, int ks, unsigned tmask,
                                          double &bd_out, int &bk_out) {
  const int cl = lane < NS ? lane : 0;
  const double *qn1 = reinterpret_cast<const double *>(smem + WaveLds::OFF_QN1);
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
    if (!((tmask >> t) & 1u)) continue;  // scalar branch
    int k = ks + t - 3;
    k += (k < 0) ? NS : 0;
    k -= (k >= NS) ? NS : 0;
    int c = cl + k;
    c -= (c >= NS) ? NS : 0;
    const float4 *qp = reinterpret_cast<const float4 *>(smem + WaveLds::OFF_QF32 + c * (NR * 4));
    double da[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int i = 0; i < 5; i++) {  // element r feeds accumulator r % 4 (Eigen's redux order), as in phase_b
      const float4 q4 = qp[i];
      da[0] = fma((double)q4.x, e[4 * i + 0], da[0]);
      da[1] = fma((double)q4.y, e[4 * i + 1], da[1]);
      da[2] = fma((double)q4.z, e[4 * i + 2], da[2]);
      da[3] = fma((double)q4.w, e[4 * i + 3], da[3]);
    }
    double dot = (da[0] + da[2]) + (da[1] + da[3]);
    if constexpr (SO != dev::SO_SSE2) {  // (products of two fp32 values are exact in fp64: fused or not is the same number)
      const float *qf = reinterpret_cast<const float *>(qp);
      dot = dev::redux_prod<SO, NR>([&](int r) { return (double)qf[r]; }, [&](int r) { return e[r]; });
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
  if (lane < 8 && tt < 7 && ((tmask >> tt) & 1u)) {
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

}  // namespace
}  // namespace sc
}  // namespace rsx
