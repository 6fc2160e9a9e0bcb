// pmc.hip -- max-clique inlier selection on gfx950: the stage between the matcher and the ORORA solver ("PMC max-clique
// prune").  Upstream takes it from TEASER++ / the PMC library; neither is in the reference
// checkout (the ORORA submodule is an empty directory: /root/reference/.gitmodules:1-3, README.md:19,26-29), so this follows
// the construction restated in oracle/pmc_ref.{h,c} -- PARITY UNPINNED -- and reproduces that oracle's selection exactly.
//
//   consistency graph   vertices = the K matches of a pair; i ~ j iff | ||src_i - src_j|| - ||dst_i - dst_j|| | < tau, evaluated
//                       without square roots in fp64 (pmc_ref.h: s = (A + B) - tau^2; edge <=> s < 0 || s^2 < 4 A B)
//   core numbers        level-synchronous peeling (exact; the result is unique)
//   greedy clique       seeds and candidates in (core descending, index ascending) order, <= MAX_SEEDS seeds
//
// Mapping.  A VERTEX SET IS ONE REGISTER PER LANE OF A WAVEFRONT: 64 lanes x 32 bits = 2048 vertices, vertex u = bit u / 64 of
// lane u % 64 (lane-interleaved, so that lane l of the adjacency build reads points l, 64 + l, ...: consecutive LDS
// addresses).  Set intersection = one v_and against a 256-byte adjacency row (one coalesced load), |P| = v_bcnt + a wave
// reduction, "is u in P" = v_readlane + shift.  The adjacency of a pair lives in a 512 KB slab of HBM (written once, read
// once by the peeling and once per picked vertex).  Three kernels per chunk of <= CHUNK pairs, each with the shape ITS phase
// wants (the first build ran all phases in one 256-thread workgroup per pair with 45 KB of LDS: three workgroups per CU, during
// the walk one of their twelve wavefronts at work, every predicate in fp64 -- 6.6 ms per 3 500 pairs; now 3.8):
//   pmc_build_kernel   a workgroup per block of 256 ROWS of a pair (the cost of a pair goes with K^2: whole pairs left the chip
//                      at 1.9 of 4 waves per SIMD), points in LDS: row i of the graph per wave iteration, 64 lanes x
//                      ceil(K / 64) columns of the predicate -- decided in fp32 wherever fp32 can, edge_f32 -- two columns per
//                      step with the next two on their way -> one word per lane -> one 256-byte row store; degree by popcount
//   pmc_cores_kernel   256 threads: core numbers by level-synchronous peeling -- frontier = alive vertices of degree <=
//                      level (all threads), the frontier rows & alive decrement their neighbours' degrees (LDS atomics, four
//                      rows in flight per wave); empty frontier -> level = the smallest remaining degree -- then a bitonic
//                      sort of (2047 - core) << 11 | index in LDS -> `order`
//   pmc_walk_kernel    ONE WAVEFRONT per pair (up to 20 pairs per CU): the greedy walk.  64 candidates of `order` are
//                      tested against P at once (ds_bpermute + ballot), the rows of the next eight members are loaded
//                      together, each is re-tested against the shrinking P before it joins; then member flags, the selected
//                      matches compacted in their original order for the solver, one info record
// Roofline: the build is VALU-issue bound (SQ_ACTIVE_INST_VALU = 100 % of the SIMD cycles at ~31 instructions per pair of
// matches: 2.4 of the 3.8 ms); peeling and walk are dependent chains per pair (barrier / load row -> and -> popcount) that last
// as long as the longest pair's, hidden by the number of pairs in flight (0.9 + 0.6 ms).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "pmc.h"

namespace {

using rsx::pmc::MAX_K;
using rsx::pmc::MAX_SEEDS;
constexpr int NT = 256;
constexpr int ROWW = 64;  // words per adjacency row
// per-pair record between the kernels: core numbers, the order, the degrees, the largest core number
constexpr int META_CORE = 0, META_ORDER = 2 * MAX_K, META_DEG = 4 * MAX_K, META_HDR = 8 * MAX_K;
constexpr int META_BYTES = rsx::pmc::META_BYTES;
static_assert(META_HDR + 64 <= META_BYTES, "meta layout");

struct Args {
  const float2 *src, *dst;
  const int64_t *offsets;
  int n_pairs, first;  // this launch: pairs [first, first + gridDim.x)
  double tau2;
  uint32_t *slabs;     // [chunk][MAX_K][ROWW]
  char *meta;          // [chunk][META_BYTES]
  uint8_t *member;
  rsx_orora_pmc_info *info;
  float2 *sel_src, *sel_dst;
  int32_t *sel_cnt;
  int64_t sel_cap;  // matches the sel arrays hold
};

__device__ __forceinline__ int wave_sum_i(int x) {  // every lane: the sum over the 64 lanes (DPP + v_readlane: no LDS round trips)
  x += __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]
  x += __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xf, 0xf, true);   // quad_perm [2,3,0,1]
  x += __builtin_amdgcn_update_dpp(0, x, 0x141, 0xf, 0xf, true);  // row_half_mirror
  x += __builtin_amdgcn_update_dpp(0, x, 0x140, 0xf, 0xf, true);  // row_mirror: every lane = its row's sum
  return __builtin_amdgcn_readlane(x, 0) + __builtin_amdgcn_readlane(x, 16) + __builtin_amdgcn_readlane(x, 32) + __builtin_amdgcn_readlane(x, 48);
}
__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = min(v, __shfl_xor(v, off));
  return v;
}

// the edge predicate of oracle/pmc_ref.h, operation for operation (this file is compiled with -ffp-contract=off)
__device__ __forceinline__ bool edge(double six, double siy, double dix, double diy, float2 sj, float2 dj, double tau2) {
  const double dax = (double)sj.x - six, day = (double)sj.y - siy;
  const double dbx = (double)dj.x - dix, dby = (double)dj.y - diy;
  const double A = dax * dax + day * day;
  const double B = dbx * dbx + dby * dby;
  const double sm = (A + B) - tau2;
  return (sm < 0.0) || (sm * sm < 4.0 * (A * B));
}

// the pair of this workgroup: false = not pruned (fewer than 2 / more than MAX_K matches / no room in the sel arrays)
struct Pair {
  int pair, K;
  int64_t o, K64;
  bool no_room;
};
__device__ __forceinline__ bool pair_of(const Args &a, Pair &p) {
  p.pair = a.first + (int)blockIdx.x;
  p.o = a.offsets[p.pair];
  p.K64 = a.offsets[p.pair + 1] - p.o;
  p.no_room = a.sel_src && p.o + p.K64 > a.sel_cap;
  p.K = (int)p.K64;
  return !(p.K64 < 2 || p.K64 > MAX_K || p.no_room);
}

// Everything a whole wavefront decides goes through an SGPR (readfirstlane): hipcc cannot see that `wave`, or a value every
// thread read from the same LDS word, is uniform, and structurises such branches with exec masks -- the first build of the
// peeling loop `if (tid == 0) L.level = m; continue;` parked lane 0's store behind a loop the other 63 lanes of its wave
// could not leave without it (the kernel never returned).  Uniform values in SGPRs make those branches scalar.

// The same predicate decided in fp32 wherever fp32 can decide it (round 6).  The fp64 form alone made the build 5.2 ms of a
// 7.9 ms selection -- not for its fp64 rate but for its instruction count: a wave64 VALU instruction is 4 cycles of its SIMD
// (SQ_ACTIVE_INST_VALU = SQ_INSTS_VALU quad-cycles) and the step was 36 of them.  In fp32, with u = 2^-24: A, B carry a relative
// error <= 4.2 u, v_sqrt_f32 is good to 1 ulp (2 u), so da = sqrt A, db = sqrt B are within 4.1 u of the true distances and
// x = da - db within 4.1 u (da + db) + u |x| of the true difference.  Hence, with e = 8 u (da + db):
//     |x| + e < tau (1 - 4 u)   =>  edge,        |x| - e > tau (1 + 4 u)   =>  no edge
// (tau as a float is within u of the bound, u |x| ~ u tau; the fp64 form of oracle/pmc_ref.h errs by ~1e-16; slack ~4 u (da + db)).
// Whatever is left -- pairs within ~3e-4 m of the bound at 150 m range, NaN coordinates -- takes the fp64 form; a wavefront
// skips it when none of its 64 lanes needs it (tests/test_gpu_pmc.py::test_edges_at_the_bound).
typedef float f2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void edge_f32(f2v si, f2v di, f2v sj, f2v dj, float tau_lo, float tau_hi, bool &yes, bool &no) {
  const f2v da2 = (sj - si) * (sj - si), db2 = (dj - di) * (dj - di);  // (v_pk_add_f32 / v_pk_mul_f32: x and y in one instruction)
  const float A = da2.x + da2.y, B = db2.x + db2.y;                      // relative error <= 4.2 u, as with the fused form
  const float da = __builtin_amdgcn_sqrtf(A), db = __builtin_amdgcn_sqrtf(B);
  const float x = fabsf(da - db), e = (da + db) * 0x1p-21f;
  yes = x + e < tau_lo;
  no = x - e > tau_hi;
}

// pmc_build_kernel: rows [256 blockIdx.y, + 256) of the consistency graph of pair blockIdx.x (rows into the slab, degrees into
// the pair's record).  A workgroup per ROW BLOCK, not per pair: the cost of a pair goes with K^2 (x 25 between 300 and 1500
// matches), and with one workgroup per pair the chip ran at 1.9 of 4 waves per SIMD (SQ_WAVE_CYCLES / GRBM_GUI_ACTIVE) waiting
// for the long ones.  Two columns per step, the next two already on their way from LDS.
constexpr int BUILD_ROWS = 256;
__global__ __launch_bounds__(NT) void pmc_build_kernel(Args a) {
  __shared__ float2 s_src[MAX_K], s_dst[MAX_K];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  Pair p;
  if (!pair_of(a, p)) return;
  const int K = p.K, nc = (K + 63) >> 6, nc2 = (nc + 1) & ~1;
  const int row0 = (int)blockIdx.y * BUILD_ROWS;
  if (row0 >= K) return;
  const int row1 = row0 + BUILD_ROWS < K ? row0 + BUILD_ROWS : K;
  uint32_t *adj = a.slabs + (size_t)blockIdx.x * (rsx::pmc::SLAB_BYTES / 4);
  int *g_deg = reinterpret_cast<int *>(a.meta + (size_t)blockIdx.x * META_BYTES + META_DEG);
  for (int i = tid; i < nc2 * 64 && i < MAX_K; i += NT) {  // (slots past K hold zeros: finite, and masked below)
    s_src[i] = i < K ? a.src[p.o + i] : float2{0.0f, 0.0f};
    s_dst[i] = i < K ? a.dst[p.o + i] : float2{0.0f, 0.0f};
  }
  __syncthreads();
  const float tau = (float)sqrt(a.tau2), tau_lo = tau * (1.0f - 0x1p-22f), tau_hi = tau * (1.0f + 0x1p-22f);
  uint32_t vm = 0;  // this lane's columns that exist: bit c <=> c * 64 + lane < K
  for (int c = 0; c < nc; c++) vm |= (c * 64 + lane < K) ? (1u << c) : 0u;
  const f2v *v_src = reinterpret_cast<const f2v *>(s_src), *v_dst = reinterpret_cast<const f2v *>(s_dst);
  for (int i = row0 + wave; i < row1; i += NT / 64) {
    const f2v si = v_src[i], di = v_dst[i];
    // columns from the last pair down to the first: a row's word is built by shifting the new bit in from the right
    // (w = w + w + bit: a compare into vcc and one add-with-carry), the pair after next already on its way from LDS
    uint32_t w = 0;
    int c = nc2 - 2;
    f2v sj0 = v_src[c * 64 + lane], dj0 = v_dst[c * 64 + lane], sj1 = v_src[c * 64 + 64 + lane], dj1 = v_dst[c * 64 + 64 + lane];
    for (; c >= 0; c -= 2) {
      const f2v a0 = sj0, b0 = dj0, a1 = sj1, b1 = dj1;
      const int jn = ((c - 2) * 64 + lane) & (MAX_K - 1);  // (the read before the first pair of columns is harmless and unused)
      sj0 = v_src[jn];
      dj0 = v_dst[jn];
      sj1 = v_src[(jn + 64) & (MAX_K - 1)];
      dj1 = v_dst[(jn + 64) & (MAX_K - 1)];
      bool y0, n0, y1, n1;
      edge_f32(si, di, a0, b0, tau_lo, tau_hi, y0, n0);
      edge_f32(si, di, a1, b1, tau_lo, tau_hi, y1, n1);
      if (__ballot(!(y0 || n0) || !(y1 || n1)) != 0ull) {  // (wave-uniform) somebody is too close to the bound for fp32
        const double six = si.x, siy = si.y, dix = di.x, diy = di.y;
        const bool e0 = edge(six, siy, dix, diy, float2{a0.x, a0.y}, float2{b0.x, b0.y}, a.tau2),
                   e1 = edge(six, siy, dix, diy, float2{a1.x, a1.y}, float2{b1.x, b1.y}, a.tau2);
        y0 = (y0 || n0) ? y0 : e0;
        y1 = (y1 || n1) ? y1 : e1;
      }
      w = (w << 2) | (y1 ? 2u : 0u) | (y0 ? 1u : 0u);
    }
    w &= vm;                                                   // columns past K
    if (lane == (i & 63)) w &= ~(1u << (i >> 6));              // no self loop
    adj[(size_t)i * ROWW + lane] = w;
    const int d = wave_sum_i(__popc(w));
    if (lane == 0) g_deg[i] = d;
  }
}

struct CoreLds {
  int deg[MAX_K];
  uint32_t keys[MAX_K];    // the sort keys; before the sort: the frontier list (uint16_t[MAX_K])
  uint16_t core[MAX_K];
  uint32_t alive[ROWW];
  int red[4];
  int nfront, level, max_core;
};

__global__ __launch_bounds__(NT) void pmc_cores_kernel(Args a) {
  __shared__ CoreLds L;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  Pair p;
  if (!pair_of(a, p)) return;
  const int K = p.K, nc = (K + 63) >> 6;
  const uint32_t *adj = a.slabs + (size_t)blockIdx.x * (rsx::pmc::SLAB_BYTES / 4);
  char *meta = a.meta + (size_t)blockIdx.x * META_BYTES;
  uint16_t *front = reinterpret_cast<uint16_t *>(L.keys);
  {
    const int *deg = reinterpret_cast<const int *>(meta + META_DEG);
    for (int i = tid; i < K; i += NT) L.deg[i] = deg[i];
  }
  if (tid < ROWW) {
    uint32_t w = 0;  // alive = every vertex < K
    for (int c = 0; c < nc; c++) w |= (c * 64 + tid < K) ? (1u << c) : 0u;
    L.alive[tid] = w;
  }
  if (tid == 0) {
    L.level = 0;
    L.max_core = 0;
  }
  __syncthreads();

  // ---- core numbers by level-synchronous peeling (one exit, three barriers per round) ----
  for (bool peeling = true; peeling;) {
    if (tid == 0) L.nfront = 0;
    __syncthreads();
    const int level = __builtin_amdgcn_readfirstlane(L.level);
    int my_min = 0x7fffffff;
    for (int v = tid; v < K; v += NT) {
      if ((L.alive[v & 63] >> (v >> 6)) & 1u) {
        const int d = L.deg[v];
        if (d <= level) {
          front[atomicAdd(&L.nfront, 1)] = (uint16_t)v;
          L.core[v] = (uint16_t)level;
          atomicAnd(&L.alive[v & 63], ~(1u << (v >> 6)));
        } else {
          my_min = min(my_min, d);
        }
      }
    }
    my_min = wave_min_i(my_min);
    if (lane == 0) L.red[wave] = my_min;
    __syncthreads();
    const int nf = __builtin_amdgcn_readfirstlane(L.nfront);
    if (nf == 0) {
      const int m = __builtin_amdgcn_readfirstlane(min(min(L.red[0], L.red[1]), min(L.red[2], L.red[3])));
      if (m == 0x7fffffff) peeling = false;  // nothing alive: done
      else if (tid == 0) L.level = m;        // jump to the smallest remaining degree
    } else {
      // four frontier rows per wave in flight (one row after the other made every round a chain of memory latencies)
      for (int f0 = 4 * wave; f0 < nf; f0 += 4 * (NT / 64)) {
        uint32_t wr[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
          wr[q] = 0;
          if (f0 + q < nf) {
            const int u = __builtin_amdgcn_readfirstlane((int)front[f0 + q]);
            wr[q] = adj[(size_t)u * ROWW + lane];
          }
        }
        const uint32_t al = L.alive[lane];
#pragma unroll
        for (int q = 0; q < 4; q++) {
          uint32_t w = wr[q] & al;
          while (w) {
            const int c = __ffs((int)w) - 1;
            w &= w - 1;
            atomicSub(&L.deg[c * 64 + lane], 1);
          }
        }
      }
      if (tid == 0) L.max_core = level;  // levels only grow: the last one that removed something is the largest core number
    }
    __syncthreads();  // every wave has read nfront / red and finished its decrements before the next round resets them
  }

  // ---- order = vertices by (core descending, index ascending) ----
  int n2 = 64;
  while (n2 < K) n2 <<= 1;
  for (int i = tid; i < n2; i += NT) L.keys[i] = i < K ? (((uint32_t)(2047 - L.core[i])) << 11) | (uint32_t)i : 0xffffffffu;
  __syncthreads();
  for (int k2 = 2; k2 <= n2; k2 <<= 1)
    for (int j = k2 >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < n2; i += NT) {
        const int q = i ^ j;
        if (q > i) {
          const uint32_t x = L.keys[i], y = L.keys[q];
          if (((i & k2) == 0) == (x > y)) {
            L.keys[i] = y;
            L.keys[q] = x;
          }
        }
      }
      __syncthreads();
    }
  uint16_t *g_core = reinterpret_cast<uint16_t *>(meta + META_CORE), *g_order = reinterpret_cast<uint16_t *>(meta + META_ORDER);
  for (int i = tid; i < K; i += NT) {
    g_core[i] = L.core[i];
    g_order[i] = (uint16_t)(L.keys[i] & 2047u);
  }
  if (tid == 0) *reinterpret_cast<int *>(meta + META_HDR) = L.max_core;
}

__global__ __launch_bounds__(64) void pmc_walk_kernel(Args a) {
  __shared__ uint16_t s_core[MAX_K], s_order[MAX_K];
  const int lane = threadIdx.x;
  Pair p;
  if (!pair_of(a, p)) {  // nothing to prune with / too large for the stage: every match passes
    if (a.member)
      for (int64_t i = lane; i < p.K64; i += 64) a.member[p.o + i] = 1;
    if (lane == 0) {
      if (a.sel_cnt) a.sel_cnt[p.pair] = -1;  // the solver reads the caller's arrays
      if (a.info)
        a.info[p.pair] = rsx_orora_pmc_info{(int32_t)(p.K64 > 0 ? p.K64 : 0), 0, 0,
                                            RSX_ORORA_PMC_PASSTHROUGH | (p.no_room && p.K64 >= 2 && p.K64 <= MAX_K ? RSX_ORORA_PMC_NO_WORKSPACE : 0)};
    }
    return;
  }
  const int K = __builtin_amdgcn_readfirstlane(p.K), nc = (K + 63) >> 6;
  const uint32_t *adj = a.slabs + (size_t)blockIdx.x * (rsx::pmc::SLAB_BYTES / 4);
  const char *meta = a.meta + (size_t)blockIdx.x * META_BYTES;
  {
    const uint16_t *g_core = reinterpret_cast<const uint16_t *>(meta + META_CORE), *g_order = reinterpret_cast<const uint16_t *>(meta + META_ORDER);
    for (int i = lane; i < K; i += 64) {
      s_core[i] = g_core[i];
      s_order[i] = g_order[i];
    }
  }
  const int max_core = __builtin_amdgcn_readfirstlane(*reinterpret_cast<const int *>(meta + META_HDR));
  __syncthreads();

  uint32_t best = 0, cm = 0;
  int best_n = 0, seeds = 0, cm_for = -1;
  for (int t = 0; t < K && seeds < MAX_SEEDS; t++) {
    const int v = __builtin_amdgcn_readfirstlane((int)s_order[t]);
    const int cv = __builtin_amdgcn_readfirstlane((int)s_core[v]);
    if (cv + 1 <= best_n || best_n == max_core + 1) break;
    if ((((uint32_t)__builtin_amdgcn_readlane((int)best, v & 63)) >> (v >> 6)) & 1u) continue;  // a member of the clique in hand
    seeds++;
    if (cm_for != best_n) {  // candidates must have core >= |best|
      cm = 0;
      for (int c = 0; c < nc; c++) {
        const int u = c * 64 + lane;
        if (u < K && (int)s_core[u] >= best_n) cm |= 1u << c;
      }
      cm_for = best_n;
    }
    uint32_t P = adj[(size_t)v * ROWW + lane] & cm;
    uint32_t C = (lane == (v & 63)) ? (1u << (v >> 6)) : 0u;
    int n = 1, np = __builtin_amdgcn_readfirstlane(wave_sum_i(__popc(P)));
    bool abandoned = n + np <= best_n;
    for (int s = 0; s < K && np > 0 && !abandoned; s += 64) {
      const int idx = s + lane;
      const int u = idx < K ? (int)s_order[idx] : 0;
      const uint32_t pw = (uint32_t)__shfl((int)P, u & 63);
      unsigned long long mask = __ballot(idx < K && ((pw >> (u >> 6)) & 1u));
      while (mask && np > 0 && !abandoned) {
        // the next (up to) PF candidates of this chunk that were in P when the mask was taken: rows loaded together
        constexpr int PF = 8;
        int cu[PF];
        uint32_t row[PF];
        int got = 0;
#pragma unroll
        for (int q = 0; q < PF; q++) {
          cu[q] = -1;
          row[q] = 0;
          if (mask) {
            const int i0 = __ffsll((long long)mask) - 1;
            mask &= mask - 1;
            cu[q] = __builtin_amdgcn_readlane(u, i0);
            row[q] = adj[(size_t)cu[q] * ROWW + lane];
            got++;
          }
        }
#pragma unroll
        for (int q = 0; q < PF; q++) {
          if (q < got && np > 0 && !abandoned) {
            const int uq = cu[q];
            const uint32_t pq = (uint32_t)__builtin_amdgcn_readlane((int)P, uq & 63);
            if ((pq >> (uq >> 6)) & 1u) {  // still in P: joins the clique
              if (lane == (uq & 63)) C |= 1u << (uq >> 6);
              n++;
              P &= row[q];
              np = __builtin_amdgcn_readfirstlane(wave_sum_i(__popc(P)));
              if (n + np <= best_n) abandoned = true;
            }
          }
        }
      }
    }
    if (!abandoned && n > best_n) {
      best_n = n;
      best = C;
    }
  }

  // ---- output: flags, the selected matches in their original order, the info record ----
  int run = 0;
  for (int c = 0; c < nc; c++) {  // vertex c * 64 + lane = bit c of this lane's word
    const int v = c * 64 + lane;
    const bool sel = v < K && ((best >> c) & 1u);
    if (v < K && a.member) a.member[p.o + v] = sel ? 1 : 0;
    const unsigned long long bal = __ballot(sel);
    if (sel && a.sel_src) {
      const int pos = run + __popcll(bal & ((1ull << lane) - 1ull));
      a.sel_src[p.o + pos] = a.src[p.o + v];
      a.sel_dst[p.o + pos] = a.dst[p.o + v];
    }
    run += __popcll(bal);
  }
  if (lane == 0) {
    if (a.sel_cnt) a.sel_cnt[p.pair] = best_n;
    if (a.info) a.info[p.pair] = rsx_orora_pmc_info{best_n, max_core, seeds, best_n == max_core + 1 ? RSX_ORORA_PMC_PROVEN : 0};
  }
}

}  // namespace

namespace rsx {
namespace pmc {

// Chunks of <= CHUNK pairs, one after the other on the caller's stream.  (Measured and not kept: two chunks in flight, the
// VALU-bound build of chunk c + 1 on the caller's stream beside the peeling + walk of chunk c on a side stream -- the
// peeling and the walk last as long as their LONGEST pair's chain of barriers and memory latencies whatever the number of
// pairs (0.5 - 0.65 ms each for 875 pairs as for 2 048), so four chunks of 875 pairs cost 5.3 ms against 5.0 ms for two
// chunks in sequence, and one chunk of all 3 500 is the fastest.)
int launch(Workspace &ws, int device, const float2 *d_src, const float2 *d_dst, const int64_t *d_offsets, int n_pairs, double tau,
           uint8_t *d_member, rsx_orora_pmc_info *d_info, float2 *d_sel_src, float2 *d_sel_dst, int32_t *d_sel_cnt, int64_t sel_cap, hipStream_t s) {
  (void)device;
  if (n_pairs <= 0) return RSX_OK;
  if (!(tau > 0.0)) return rsx::fail(RSX_ERR_BAD_ARG, "the consistency bound (tim_noise_bound) must be positive");
  const int chunk = n_pairs < CHUNK ? n_pairs : CHUNK;
  RSX_TRY(ws.slabs.reserve((size_t)chunk * SLAB_BYTES, s, false));
  RSX_TRY(ws.meta.reserve((size_t)chunk * META_BYTES, s, false));
  Args a;
  a.src = d_src;
  a.dst = d_dst;
  a.offsets = d_offsets;
  a.n_pairs = n_pairs;
  a.tau2 = tau * tau;
  a.slabs = ws.slabs.as<uint32_t>();
  a.meta = ws.meta.as<char>();
  a.member = d_member;
  a.info = d_info;
  a.sel_src = d_sel_src;
  a.sel_dst = d_sel_dst;
  a.sel_cnt = d_sel_cnt;
  a.sel_cap = sel_cap;
  for (int first = 0; first < n_pairs; first += chunk) {
    const int n = n_pairs - first < chunk ? n_pairs - first : chunk;
    a.first = first;
    hipLaunchKernelGGL(pmc_build_kernel, dim3((unsigned)n, (unsigned)(MAX_K / BUILD_ROWS)), dim3(NT), 0, s, a);
    hipLaunchKernelGGL(pmc_cores_kernel, dim3((unsigned)n), dim3(NT), 0, s, a);
    hipLaunchKernelGGL(pmc_walk_kernel, dim3((unsigned)n), dim3(64), 0, s, a);
  }
  RSX_HIP(hipGetLastError());
  return RSX_OK;
}

}  // namespace pmc
}  // namespace rsx
