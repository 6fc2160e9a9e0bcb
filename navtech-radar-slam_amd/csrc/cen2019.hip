// cen2019.hip -- cen2019 radar keypoint extraction on gfx950 (polar 400 x 3360 u8 images -> keypoints), batched.
//
// The reference gets these keypoints from its ORORA submodule, which is an empty directory in the
// reference checkout (.gitmodules:1-3; README.md:29 "extracted via cen2019 method"), so this follows
// the published method as restated in oracle/cen2019_ref.c (PARITY UNPINNED).
// It replaces the feature-extraction call of the upstream file-based odometry.cpp entry (README.md:27).
//
// Round 3: NO SORT.  The method's greedy region marking walks the pixels in global intensity order
// under a region budget; rounds 1-2 replayed that walk per azimuth after two rocPRIM radix sorts of
// ~0.5 M candidates (36 launches, a host sync for the candidate count).  The walk has a closed form
// (proof in oracle/cen2019_np.py, checked there against the sequential oracle on the CPU):
//   key(p)  = (h(p) descending, pixel index ascending) as one uint64; neg(p) = s(p) < 0
//   a pixel with s >= 0 is only ever marked by ITSELF; a maximal run N of neg pixels is marked as a
//   whole by the first visited pixel among {left neighbour of N, right neighbour of N, pixels of N}:
//       MK(p) = key(p) if !neg(p) else min key over those "touchers" of p's run          (mark key)
//   a visited candidate p opens a NEW region  <=>  key(p) is the minimum of every run it touches
//   the budget ends the walk at K* = the max_points-th smallest key among the region openers
//   p is marked in the end  <=>  MK(p) <= K*  and  MK(p) is a candidate (h > mean_h)
// so the chain is three passes over the image bytes and ONE selection; the intermediates in HBM are 2 bytes per 8 pixels
// (opener bits + how high the thread's openers reach) and 1 byte per 8 pixels (mark bits) -- h is recomputed from the bytes:
//   cen_stats    bytes -> sum(bytes), max |fft(r+1) - fft(r-1)|                     (global: mean, max g)
//   cen_scalars  (a thread per image) mean and 1 / max g, once
//   cen_hist     per azimuth: which pixels OPEN a region -- per-run maxima of h as plain max-scans of (segment | ord(h))
//                keys, see row_opens -- -> 4096-bin histogram of the openers' h, fixed-point sum of h (mean_h), opener bits
//   cen_pick     (one block per image) the bin B* that holds the max_points-th opener
//   cen_collect  the recorded opener bits of the threads that reach B*: appends the openers of bin B* (a few dozen keys)
//   cen_resolve  (one block per image) selects K* among them
//   cen_runs     per azimuth: marks = OR over run flags (a hit sets the flag of the run(s) it touches); closed runs of marks
//                and their arg-max by two plain max-scans; mark bits
//   cen_adjacent per azimuth: keep the runs that meet a mark of the azimuth above or below, ordered compaction
//   cen_pack     row-major packing of the rows' keypoints (+ polar -> Cartesian)
// One workgroup per (azimuth, image): a launch over a batch of B images is B x rows workgroups, every
// dependency between passes is a kernel boundary, nothing returns to the host.  (A first version
// folded the three small kernels into "last block done" tickets: a device-scope __threadfence per
// workgroup writes back / invalidates the XCD's L2 on this multi-die part and cost 40 us per block.)
// Arithmetic is order independent by construction (integer byte sum, max, 2^40 fixed-point sum of h),
// so the result is bit-identical to the oracle.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <new>
#include <type_traits>

#include "rsx_common.h"

namespace {

constexpr int NBIN = 4096;               // histogram of h over [-1, 1]
constexpr double FIX = 1099511627776.0;  // 2^40
constexpr unsigned long long KINF = ~0ull;
constexpr int MAX_SUB_BATCH = 128;       // images per internal launch group (workspace = 14 MB per image)

struct Scal {  // per image
  unsigned long long sum_bytes;
  long long fix_sum;
  unsigned long long klimit;  // a pixel is marked in the end iff MK(p) < klimit
  unsigned int max_g_bits;
  float mean;      // mean of fft = byte / 255 over the image   } cen_scalars, once per image: every thread of every row block
  float rcp_maxg;  // RN(1 / max g), 0 when max g = 0             } used to redo these divisions (45 fp64 instructions)
  unsigned int pad0;
  int bstar;               // histogram bin of the max_points-th region opener (-1: there are fewer openers)
  unsigned int above;      // openers in the bins above bstar
  unsigned int n_list;     // openers of bin bstar collected so far
  unsigned int n_targets;
  unsigned int pad[2];
};
static_assert(sizeof(Scal) == 64, "Scal layout");

__device__ __forceinline__ int wave_sum_i32(int x) {  // every lane: the sum over the 64 lanes
  x += __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]
  x += __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xf, 0xf, true);   // quad_perm [2,3,0,1]
  x += __builtin_amdgcn_update_dpp(0, x, 0x141, 0xf, 0xf, true);  // row_half_mirror
  x += __builtin_amdgcn_update_dpp(0, x, 0x140, 0xf, 0xf, true);  // row_mirror: every lane = its row's sum
  return __builtin_amdgcn_readlane(x, 0) + __builtin_amdgcn_readlane(x, 16) + __builtin_amdgcn_readlane(x, 32) + __builtin_amdgcn_readlane(x, 48);
}
__device__ __forceinline__ int wave_max_i32(int x) {  // every lane: the maximum over the 64 lanes
  auto mx = [](int a, int b) { return a > b ? a : b; };
  x = mx(x, __builtin_amdgcn_update_dpp(x, x, 0xB1, 0xf, 0xf, false));
  x = mx(x, __builtin_amdgcn_update_dpp(x, x, 0x4E, 0xf, 0xf, false));
  x = mx(x, __builtin_amdgcn_update_dpp(x, x, 0x141, 0xf, 0xf, false));
  x = mx(x, __builtin_amdgcn_update_dpp(x, x, 0x140, 0xf, 0xf, false));
  return mx(mx(__builtin_amdgcn_readlane(x, 0), __builtin_amdgcn_readlane(x, 16)), mx(__builtin_amdgcn_readlane(x, 32), __builtin_amdgcn_readlane(x, 48)));
}
__device__ __forceinline__ unsigned ord_f32(float f) {
  const unsigned u = __float_as_uint(f);
  return u ^ ((unsigned)((int)u >> 31) | 0x80000000u);  // negative: ~u, otherwise u | sign bit
}
__device__ __forceinline__ float canon0(float f) { return f == 0.0f ? 0.0f : f; }  // -0.0 sorts like +0.0 (oracle: h compares)

__device__ __forceinline__ float mean_fft(const Scal *sc, int64_t n) { return (float)((double)sc->sum_bytes / 255.0 / (double)n); }
__device__ __forceinline__ float mean_h_of(long long fix_sum, int64_t n) { return (float)((double)fix_sum / FIX / (double)n); }
// candidates are the pixels with h > mean_h  <=>  key < kmean
__device__ __forceinline__ unsigned long long kmean_of(float mh) { return (unsigned long long)(~ord_f32(canon0(mh))) << 32; }

// one block per (azimuth, image): byte sum and largest range gradient of the row, C bins per thread from aligned dwords
// (round 3, first build: one byte per lane and iteration, 0.6 TB/s).
// Round 4: the gradient maximum in the INTEGER domain.  g(p) = |t[b(p+1)] - t[b(p-1)]| with t[x] = fl(x / 255): t is increasing
// with steps of 1/255 +- 2^-24, the subtraction adds at most another 2^-24, so a pair of bytes d apart gives g within 1.8e-7 of
// d / 255 and a pair with a larger byte difference ALWAYS gives the larger g (1 / 255 = 3.9e-3 apart).  The largest g of the
// image is therefore reached at a pixel with the largest |b(p+1) - b(p-1)|: every thread keeps the integer maximum of its
// pixels, a wave agrees on its maximum D, and only the pixels that reach D (a handful per wave) form their float g -- with the
// same two correctly rounded divisions and the same subtraction the byte -> float table of the other passes holds.  Before:
// a 256-entry division table per block (and its barrier) and two LDS look-ups, a subtraction and a maximum for EVERY pixel.
// Round 6: the bytes stay packed.  Four pixels a dword: the byte sum is one v_sad_u8 per dword; the absolute differences
// |b(p+1) - b(p-1)| are formed on the even and the odd bytes of the dword as two 16-bit lanes each (v_pk_max / v_pk_min /
// v_pk_sub_u16 on the stream shifted one byte left and right, v_alignbyte) and folded into ONE packed running maximum per
// thread over all rows of the block; the rules at the ends of a row (reflect 101: the first and the last pixel have d = 0;
// nothing past the row) are six row-independent masks per thread, so there is no second code path for the edge wavefronts.
// The block agrees on its integer maximum D ONCE (not a wavefront per row), and only a thread that reached D goes back to
// its words -- kept in registers, 4 per row -- to form the float g of the pixels with d = D.  (Round 4 evaluated its
// "handful of lanes" branch in every wavefront and row, since some lane always reaches its own wave's maximum: two
// correctly rounded divisions under an exec mask, 8 pixels deep, for each of the 86 M pixels' wavefronts.)
constexpr int ST_ROWS = 8;  // azimuths per block of cen_stats / cen_collect in a batch (a single scan keeps one per block: 400 blocks fill the chip, 50 would not)
typedef unsigned short cen_u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_absdiff_u16(unsigned a, unsigned b) {
  const cen_u16x2 x = __builtin_bit_cast(cen_u16x2, a), y = __builtin_bit_cast(cen_u16x2, b);
  return __builtin_bit_cast(unsigned, (cen_u16x2)(__builtin_elementwise_max(x, y) - __builtin_elementwise_min(x, y)));
}
__device__ __forceinline__ unsigned pk_max_u16(unsigned a, unsigned b) {
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(cen_u16x2, a), __builtin_bit_cast(cen_u16x2, b)));
}
template <int C, int NT, int NR>  // NR = rows of a block (ST_ROWS in a batch, 1 for a single scan): the block keeps their words in registers
__global__ __launch_bounds__(NT) void cen_stats(const uint8_t *__restrict__ imgs, int64_t img_stride, int rows, int cols, int stride,
                                                int off, Scal *scal, unsigned *__restrict__ hist) {
  static_assert(C % 4 == 0, "a thread's chunk is whole dwords");
  constexpr int ND = C / 4, NWD = ND + 2, rpb = NR;
  __shared__ unsigned s_sum[NT / 64];
  __shared__ float s_max[NT / 64];
  Scal *sc = scal + blockIdx.y;
  // (the image's selection histogram is zeroed here, by the image's first block -- cen_hist runs behind a kernel boundary; it was a
  // memset of its own: one launch of the chain's eleven)
  if (blockIdx.x == 0)
    for (int b = threadIdx.x; b < NBIN / 4; b += NT) reinterpret_cast<uint4 *>(hist + (size_t)blockIdx.y * NBIN)[b] = uint4{0u, 0u, 0u, 0u};
  const int p0 = threadIdx.x * C;
  // which bytes count: in the sum, pixels inside the row; in the maximum, pixels 1 .. cols - 2 (even bytes of dword k in
  // emask[k], odd bytes in omask[k], as 16-bit lanes)
  unsigned smask[ND], emask[ND], omask[ND];
#pragma unroll
  for (int k = 0; k < ND; k++) {
    smask[k] = emask[k] = omask[k] = 0u;
#pragma unroll
    for (int bb = 0; bb < 4; bb++) {
      const int p = p0 + 4 * k + bb;
      if (p < cols) smask[k] |= 0xffu << (8 * bb);
      if (p >= 1 && p <= cols - 2) ((bb & 1) ? omask[k] : emask[k]) |= 0xffffu << (16 * (bb >> 1));
    }
  }
  unsigned sb = 0;  // <= ST_ROWS * C * 255 per thread, a wave's sum stays far below 2^32
  unsigned pmax = 0;  // two 16-bit running maxima
  unsigned pm[NR];    // ... and per row: the thread that reached D goes back to the ROW(S) that did, not to all NR (round 6: the branch --
                      // one lane at work while the block waits at the barrier -- and the two barriers around it are 16 of the kernel's
                      // 39 us: 23.6 us with the branch compiled out; per-row maxima: 36.  A wavefront-level D with two atomics per
                      // wavefront instead of the barriers: 59 us, and 68 for a single scan -- 3 200 atomics on one word)
#pragma unroll
  for (int rr = 0; rr < NR; rr++) pm[rr] = 0u;
  unsigned kw[NR][NWD], kmis[NR];
  // the stream of a row realigned to the thread's first pixel: V[0] = pixels p0 - 4 .. p0 - 1, V[1 + k] = dword k, V[ND + 1] = the pixels behind
  auto realign = [&](const unsigned (&w)[NWD], unsigned mis, unsigned (&V)[ND + 2]) {
#pragma unroll
    for (int k = 0; k < ND + 1; k++) V[k] = __builtin_amdgcn_alignbyte(w[k + 1], w[k], mis);
    V[ND + 1] = __builtin_amdgcn_alignbyte(0u, w[ND + 1], mis);
  };
  // rpb azimuths per block: a row is 3360 bytes, a block per row was mostly block start-up
#pragma unroll
  for (int rr = 0; rr < NR; rr++) {
    const int a = blockIdx.x * rpb + rr;
    unsigned(&w)[NWD] = kw[rr];
#pragma unroll
    for (int j = 0; j < NWD; j++) w[j] = 0u;
    kmis[rr] = 0u;
    if (a >= rows) continue;  // (uniform; the words stay zero: no d, no sum)
    const uint8_t *row = imgs + (int64_t)blockIdx.y * img_stride + (int64_t)a * stride + off;
    const uint8_t *pp0 = row + p0;  // (pointer arithmetic, not integers: the loads stay global_load, not flat_load)

    const unsigned mis = (unsigned)(reinterpret_cast<uintptr_t>(pp0) & 3u);

    const unsigned *wp = reinterpret_cast<const unsigned *>(pp0 - mis);
    kmis[rr] = mis;
    if (p0 < cols) {
      if (p0 > 0) w[0] = wp[-1];
#pragma unroll
      for (int j = 0; j < NWD - 1; j++)
        if (p0 + 4 * j - (int)mis < cols) w[1 + j] = wp[j];
    }
    unsigned V[ND + 2];
    realign(w, mis, V);
#pragma unroll
    for (int k = 0; k < ND; k++) {
      const unsigned X = V[1 + k];
      const unsigned L = __builtin_amdgcn_alignbyte(X, V[k], 3);      // pixels p - 1 of the dword's four
      const unsigned R = __builtin_amdgcn_alignbyte(V[2 + k], X, 1);  // pixels p + 1
      sb = __builtin_amdgcn_sad_u8(X & smask[k], 0u, sb);
      const unsigned de = pk_absdiff_u16(L & 0x00ff00ffu, R & 0x00ff00ffu), dodd = pk_absdiff_u16((L >> 8) & 0x00ff00ffu, (R >> 8) & 0x00ff00ffu);
      pm[rr] = pk_max_u16(pm[rr], de & emask[k]);
      pm[rr] = pk_max_u16(pm[rr], dodd & omask[k]);
    }
    pmax = pk_max_u16(pmax, pm[rr]);
  }
  const int dmax = (int)((pmax & 0xffffu) > (pmax >> 16) ? (pmax & 0xffffu) : (pmax >> 16));
  // D = the WAVEFRONT's largest difference (round 6; before: the block's, through LDS and a barrier -- every wavefront waited for the
  // slowest and then for the one lane that went back to its words).  A wavefront whose D lies below the image's contributes a
  // smaller g: harmless under the maximum.
  const int D = wave_max_i32(dmax);
  float mg = 0.0f;
  if (D > 0 && dmax == D) {  // (a thread or two per block; D = 0: every gradient of the block is exactly 0)
#pragma unroll
    for (int rr = 0; rr < NR; rr++) {
      if ((int)((pm[rr] & 0xffffu) > (pm[rr] >> 16) ? (pm[rr] & 0xffffu) : (pm[rr] >> 16)) != D) continue;
      unsigned V[ND + 2];
      realign(kw[rr], kmis[rr], V);
#pragma unroll
      for (int i = 0; i < C; i++) {
        const int p = p0 + i;
        const unsigned bm = (i == 0) ? (V[0] >> 24) : ((V[1 + ((i - 1) >> 2)] >> (8 * ((i - 1) & 3))) & 0xffu);
        const unsigned bp = (V[1 + ((i + 1) >> 2)] >> (8 * ((i + 1) & 3))) & 0xffu;
        const int d = bp > bm ? (int)(bp - bm) : (int)(bm - bp);
        if (p >= 1 && p <= cols - 2 && d == D)
          mg = fmaxf(mg, fabsf(__fsub_rn(__fdiv_rn((float)bp, 255.0f), __fdiv_rn((float)bm, 255.0f))));
      }
    }
  }
  sb = (unsigned)wave_sum_i32((int)sb);
  mg = __int_as_float(wave_max_i32(__float_as_int(mg)));  // non-negative floats order like their bits
  if ((threadIdx.x & 63) == 0) {
    s_sum[threadIdx.x >> 6] = sb;
    s_max[threadIdx.x >> 6] = mg;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long t = 0;
    float m = 0.0f;
    for (int wv = 0; wv < NT / 64; wv++) {
      t += s_sum[wv];
      m = fmaxf(m, s_max[wv]);
    }
    atomicAdd(&sc->sum_bytes, t);
    atomicMax(&sc->max_g_bits, __float_as_uint(m));  // non-negative floats order like their bits
  }
}

// the two divisions every later pass needs, once per image (a thread per image)
__global__ __launch_bounds__(64) void cen_scalars(Scal *scal, int nb, int64_t n_pixels) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= nb) return;
  Scal *sc = scal + i;
  const float maxg = __uint_as_float(sc->max_g_bits);
  sc->mean = mean_fft(sc, n_pixels);
  sc->rcp_maxg = (maxg > 0.0f) ? __fdiv_rn(1.0f, maxg) : 0.0f;
}

// ---------------------------------------------------------------------------------------------------------------
// Evaluation of one azimuth row by a block of NT threads, C consecutive range bins per thread (NT * C >= cols).
// LDS scratch: the 256-entry byte -> fft table and what the threads hand to each other across wavefronts.
// ---------------------------------------------------------------------------------------------------------------
template <int C, int NT>
struct RowLds {
  static constexpr int NW = NT / 64;
  float tab[256];
  // what a wavefront hands to the others.  Each array has a second half of NW zeros (written once, row_table): wavefront w
  // reads the NW slots before (after) its own, whatever w is -- the slots outside the block are zeros, the identity of max
  // and +.  (Rounds 3-5a: `ww < w ? x[ww] : 0`, two v_cndmask per slot and direction, 48 an evaluation.)
  double fwd[2 * NW];        // [NW + w]: forward scan total (sc_key keys, below); [0, NW): zeros
  double bwd[2 * NW];        // [w]: backward scan total; [NW, 2 NW): zeros
  unsigned nn_cnt[2 * NW];   // [NW + w]: non-neg pixels of the wavefront; [0, NW): zeros
  unsigned edge[NW];         // bit 0: the wavefront's first pixel is neg, bit 1: its last pixel is neg
};

template <int C, int NT>
__device__ __forceinline__ void row_table(RowLds<C, NT> &L) {
  constexpr int NW = NT / 64;
  if (threadIdx.x < 256) L.tab[threadIdx.x] = __fdiv_rn((float)threadIdx.x, 255.0f);
  if (threadIdx.x < NW) {
    L.fwd[threadIdx.x] = 0.0;
    L.bwd[NW + threadIdx.x] = 0.0;
    L.nn_cnt[threadIdx.x] = 0u;
  }
}

// h and the sign of s for the thread's C pixels of one row, straight from the image bytes: the thread's 16 bytes and their
// two neighbours come from ALIGNED dwords around row + p0 (an aligned dword that holds one byte of the image cannot cross
// a page, so the few bytes read beside the row are harmless).  Needs L.tab (visible to the block); no barrier inside.
// the chunks of a wavefront that holds the row's first pixel or reaches its end need the border rules; the others
// (wavefront-uniform: a scalar branch) do not
template <int C>
__device__ __forceinline__ bool edge_wave(int cols) {
  const int wp0 = (int)__builtin_amdgcn_readfirstlane((int)(threadIdx.x & ~63u)) * C;
  return wp0 == 0 || wp0 + 64 * C >= cols;
}

// (p0 = the chunk's first pixel, edge = the chunk's wavefront holds the row's first pixel or reaches its end: the block kernels pass
// threadIdx.x * C and edge_wave, the wave-per-row kernel its own)
template <int C, int NT>
__device__ __forceinline__ void row_load_h_at(const RowLds<C, NT> &L, const uint8_t *__restrict__ row, int cols, float mean, float maxg,
                                              float rcp_maxg, float (&h)[C], unsigned &neg, const int p0, const bool edge) {
  static_assert(C % 4 == 0, "a thread's chunk is whole dwords");
  constexpr int NWD = C / 4 + 2;  // aligned words [-1 .. C / 4] relative to (row + p0) & ~3
  neg = 0;
  unsigned w[NWD];
#pragma unroll
  for (int j = 0; j < NWD; j++) w[j] = 0u;
  const uint8_t *pp0 = row + p0;  // (pointer arithmetic, not integers: the loads stay global_load, not flat_load)

  const unsigned mis = (unsigned)(reinterpret_cast<uintptr_t>(pp0) & 3u);

  const unsigned *wp = reinterpret_cast<const unsigned *>(pp0 - mis);
  if (!edge) {  // (uniform) an inner wavefront: 0 < p0 and p0 + C + 4 <= cols for every lane -- all words exist, no lane-wise guards
#pragma unroll
    for (int j = 0; j < NWD; j++) w[j] = wp[j - 1];
  } else if (p0 < cols) {
    if (p0 > 0) w[0] = wp[-1];
#pragma unroll
    for (int j = 0; j < NWD - 1; j++)
      if (p0 + 4 * j - (int)mis < cols) w[1 + j] = wp[j];
  }
  float ft[C + 2];  // fft of pixels p0-1 .. p0+C
#pragma unroll
  for (int i = -1; i <= C; i++) {
    const int jw = (i + 4) >> 2;  // word that holds pixel i when mis = 0 (index into w: word -1 is w[0])
    const unsigned lo = w[jw], hi = w[jw + 1 < NWD ? jw + 1 : NWD - 1];
    const unsigned v = __builtin_amdgcn_alignbyte(hi, lo, mis);  // bytes mis .. mis+3 of (hi:lo)
    ft[i + 1] = L.tab[(v >> (8 * ((i + 4) & 3))) & 0xffu];
  }
  // reflect 101 at the two ends of the row: the neighbour of bin 0 on the left is bin 1, of the last bin on the right the one
  // before it -- written into the neighbour slots, so that every pixel of every thread runs the SAME code below (rounds 3-4: a
  // second, branchy copy of the pixel code for the first and the last threads, executed by the whole wavefront around them)
  if (edge) {
    if (p0 == 0) ft[0] = ft[2];
#pragma unroll
    for (int i = 0; i < C; i++)
      if (p0 + i == cols - 1) ft[i + 2] = ft[i];
  }
  // g / maxg, correctly rounded, without the ~10-instruction IEEE sequence per pixel: with y = RN(1 / maxg) (one division per
  // image) q = RN(g y), r = fma(-maxg, q, g), q' = fma(r, y, q) IS RN(g / maxg) for every pair of range gradients bytes can
  // produce -- 598 values, all 179 100 pairs with g <= maxg checked (tools/prove_cen_division.py, tests/test_cen2019_arith.py)
#pragma unroll
  for (int i = 0; i < C; i++) {
    const float g = fabsf(__fsub_rn(ft[i + 2], ft[i]));
    const float q0 = __fmul_rn(g, rcp_maxg);
    const float gn = __fmaf_rn(__fmaf_rn(-maxg, q0, g), rcp_maxg, q0);  // (maxg = 0: every g is 0 and rcp_maxg is 0: gn = 0)
    const float sv = __fsub_rn(ft[i + 1], mean);
    h[i] = __fmul_rn(sv, __fsub_rn(1.0f, gn));
    neg |= (__float_as_uint(sv) >> 31) << i;  // sv < 0 (a difference is never -0.0)
  }
  if (edge) {  // pixels past the end of the row: h = 0, not neg
#pragma unroll
    for (int i = 0; i < C; i++)
      if (p0 + i >= cols) {
        h[i] = 0.0f;
        neg &= ~(1u << i);
      }
  }
}
template <int C, int NT>
__device__ __forceinline__ void row_load_h(const RowLds<C, NT> &L, const uint8_t *__restrict__ row, int cols, float mean, float maxg,
                                           float rcp_maxg, float (&h)[C], unsigned &neg) {
  row_load_h_at(L, row, cols, mean, maxg, rcp_maxg, h, neg, (int)threadIdx.x * C, edge_wave<C>(cols));
}

// The sign bits alone (round 6, cen_runs' light path): s < 0 <=> t[b] < mean (the difference of two floats keeps its sign) <=> b < T
// with T = the number of bytes whose t[b] lies below the mean (t is increasing) -- no table look-up, no arithmetic: the chunk's
// bytes against one threshold.  Same words, same alignment, same rule past the row end as row_load_h.
template <int C>
__device__ __forceinline__ unsigned row_load_neg(const uint8_t *__restrict__ row, int cols, unsigned T, const int p0) {
  static_assert(C % 4 == 0, "a thread's chunk is whole dwords");
  constexpr int NWD = C / 4 + 1;  // aligned words [0 .. C / 4] relative to (row + p0) & ~3
  if (p0 >= cols) return 0u;
  unsigned w[NWD];
  const uint8_t *pp0 = row + p0;  // (pointer arithmetic, not integers: the loads stay global_load, not flat_load)

  const unsigned mis = (unsigned)(reinterpret_cast<uintptr_t>(pp0) & 3u);

  const unsigned *wp = reinterpret_cast<const unsigned *>(pp0 - mis);
#pragma unroll
  for (int j = 0; j < NWD; j++) w[j] = (p0 + 4 * j - (int)mis < cols) ? wp[j] : 0u;
  unsigned neg = 0;
#pragma unroll
  for (int i = C - 1; i >= 0; i--) {
    const unsigned v = __builtin_amdgcn_alignbyte(w[(i >> 2) + 1 < NWD ? (i >> 2) + 1 : NWD - 1], w[i >> 2], mis);  // bytes mis .. mis+3 of (hi:lo)
    const unsigned b = (v >> (8 * (i & 3))) & 0xffu;
    asm("v_cmp_lt_u32_e32 vcc, %1, %2\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(neg) : "v"(b), "v"(T) : "vcc");
  }
  if (p0 + C > cols) neg &= (1u << (cols - p0)) - 1u;  // pixels past the end of the row: not neg
  return neg;
}

__device__ __forceinline__ unsigned long long key_of(float hv, unsigned pixel) {
  return ((unsigned long long)(~ord_f32(canon0(hv))) << 32) | (unsigned long long)pixel;
}

// ---------------------------------------------------------------------------------------------------------------
// Round 5: the per-run minima WITHOUT 64-bit keys, markers or selects.
//
// What the walk needs of a row (file header): which pixels OPEN a region -- the pixel with the smallest key of every run it
// touches.  Keys order by (h descending, pixel ascending), so inside ONE row "smallest key" is "largest h, leftmost among
// equals", and with E(p) = the maximal run of neg pixels around p plus the non-neg pixel on either side:
//     F(p) = max h over E ∩ [.., p]   (forward, restarting at every non-neg pixel)
//     G(p) = max h over E ∩ [p, ..]   (backward, likewise)
//     p opens  <=>  (p has nothing to its left in E  or  h(p) >  F(p - 1))     (a tie goes to the pixel on the left)
//              and  (p has nothing to its right in E or  h(p) >= G(p + 1))
// where "nothing to its left" = p is the row's first pixel, or p and p - 1 are both non-neg (then p touches no run there).
// F and G are SEGMENTED max-scans; they become PLAIN max-scans of one 64-bit number per pixel when the segment number is put
// in front of the value: key = (number of non-neg pixels up to p) << 32 | ord(h(p)) -- a later segment beats everything
// before it, inside a segment the larger h wins.  With 0x40000000 added to the high word the 64 bits are a positive normal
// double whose order is the integers' order: one v_max_f64 per step, no compare-and-select pairs, no head flags, and the
// wave-level scan is DPP (row_shr / row_shl inside the rows of 16 lanes, three v_readlane across them) instead of twelve
// ds_bpermute round trips per direction.  (Rounds 3-4: two segmented min-scans over (~ord(h) << 32 | pixel) with head flags:
// 1311 VALU + 1203 scalar instructions per 8 pixels, and a 2-byte marker per pixel written for the later passes -- which
// now ask a different, cheaper question, see cen_runs.)
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double sc_key(unsigned seg, unsigned ordh) {
  return __longlong_as_double((long long)(((unsigned long long)(0x40000000u + seg) << 32) | ordh));
}
// v_max_f64 itself: llvm.maxnum in IEEE mode first canonicalises every operand that might be a signalling NaN (a second
// v_max_f64 x, x per operand: 126 of them in the first build of cen_hist); the keys are never NaNs
__device__ __forceinline__ double kmax(double a, double b) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ unsigned sc_lo(double k) { return (unsigned)(unsigned long long)__double_as_longlong(k); }
// (bound_ctrl: a lane without a source reads 0 -- no register to clear in front of every DPP move)
template <int CTRL, int ROWS = 0xf>
__device__ __forceinline__ double dpp_f64(double x) {  // lanes without a source (and rows outside ROWS): 0.0, below every key
  const unsigned long long u = (unsigned long long)__double_as_longlong(x);
  const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)u, CTRL, ROWS, 0xf, ROWS == 0xf);
  const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTRL, ROWS, 0xf, ROWS == 0xf);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
template <int CTRL, int ROWS = 0xf>
__device__ __forceinline__ unsigned dpp_u32(unsigned x) {
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, ROWS, 0xf, ROWS == 0xf);
}
__device__ __forceinline__ double readlane_f64(double x, int l) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(x);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, l), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), l);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// inclusive max-scan over the 64 lanes (REV: from lane 63 down): Hillis-Steele inside each row of 16 by DPP; across the rows
// forward by the two row broadcasts of the ISA (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3), backward --
// there is no broadcast of a row's FIRST lane -- by three v_readlane
template <bool REV>
__device__ __forceinline__ double wave_incl_max(double x, int lane) {
  if constexpr (!REV) {
    x = kmax(x, dpp_f64<0x111>(x));  // row_shr:1
    x = kmax(x, dpp_f64<0x112>(x));  // row_shr:2
    x = kmax(x, dpp_f64<0x114>(x));  // row_shr:4
    x = kmax(x, dpp_f64<0x118>(x));  // row_shr:8
    x = kmax(x, dpp_f64<0x142, 0xa>(x));  // row_bcast:15
    x = kmax(x, dpp_f64<0x143, 0xc>(x));  // row_bcast:31
    return x;
  } else {
    x = kmax(x, dpp_f64<0x101>(x));  // row_shl:1
    x = kmax(x, dpp_f64<0x102>(x));
    x = kmax(x, dpp_f64<0x104>(x));
    x = kmax(x, dpp_f64<0x108>(x));
    const double t3 = readlane_f64(x, 48), t2 = kmax(t3, readlane_f64(x, 32)), t1 = kmax(t2, readlane_f64(x, 16));
    const double suf = lane >= 48 ? 0.0 : (lane >= 32 ? t3 : (lane >= 16 ? t2 : t1));
    return kmax(x, suf);
  }
}
__device__ __forceinline__ unsigned wave_incl_add(unsigned x, int) {
  x += dpp_u32<0x111>(x);
  x += dpp_u32<0x112>(x);
  x += dpp_u32<0x114>(x);
  x += dpp_u32<0x118>(x);
  x += dpp_u32<0x142, 0xa>(x);
  x += dpp_u32<0x143, 0xc>(x);
  return x;
}
__device__ __forceinline__ unsigned wave_incl_max_u32(unsigned x) {
  auto mx = [](unsigned a, unsigned b) { return a > b ? a : b; };
  x = mx(x, dpp_u32<0x111>(x));
  x = mx(x, dpp_u32<0x112>(x));
  x = mx(x, dpp_u32<0x114>(x));
  x = mx(x, dpp_u32<0x118>(x));
  x = mx(x, dpp_u32<0x142, 0xa>(x));
  x = mx(x, dpp_u32<0x143, 0xc>(x));
  return x;
}

__device__ __forceinline__ unsigned wave_max_u32(unsigned x) {  // (uniform) the maximum over the 64 lanes
  auto mx = [](unsigned a, unsigned b) { return a > b ? a : b; };
  x = mx(x, (unsigned)__builtin_amdgcn_update_dpp((int)x, (int)x, 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
  x = mx(x, (unsigned)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x4E, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
  x = mx(x, (unsigned)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x141, 0xf, 0xf, false));  // row_half_mirror
  x = mx(x, (unsigned)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x140, 0xf, 0xf, false));  // row_mirror: every lane = its row's maximum
  return mx(mx((unsigned)__builtin_amdgcn_readlane((int)x, 0), (unsigned)__builtin_amdgcn_readlane((int)x, 16)),
            mx((unsigned)__builtin_amdgcn_readlane((int)x, 32), (unsigned)__builtin_amdgcn_readlane((int)x, 48)));
}

__device__ __forceinline__ unsigned long long dev_wave_max_u64(unsigned long long v) {  // (rare paths: shuffles)
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const unsigned long long x = ((unsigned long long)(unsigned)__shfl_xor((int)(unsigned)(v >> 32), o) << 32) | (unsigned long long)(unsigned)__shfl_xor((int)(unsigned)v, o);
    v = x > v ? x : v;
  }
  return v;
}

// facts of the thread's C pixels of one row; pixels past the row end are walls: non-neg, below every h
template <int C>
struct RowChunk {
  float h[C];
  unsigned ordh[C];  // ord_f32(canon0(h)): order-preserving, +-0 alike
  unsigned neg;      // bit i: s < 0
  unsigned excl_nn;  // non-neg pixels of the row before the thread's first pixel
};

// h, neg and the non-neg prefix counts of the row.  The caller has filled L.tab; contains two block barriers.
// light (wave-uniform; cen_runs): the wavefront is known to hold no pixel whose h matters -- only the sign bits are formed, from the
// bytes against neg_T (row_load_neg); h and ord(h) of its pixels read 0
template <int C, int NT>
__device__ __forceinline__ void row_chunk(RowLds<C, NT> &L, const uint8_t *__restrict__ row, int cols, float mean, float maxg, float rcp_maxg,
                                          RowChunk<C> &R, bool light = false, unsigned neg_T = 0u) {
  constexpr unsigned FULL = (C == 32) ? 0xffffffffu : ((1u << C) - 1u);
  constexpr int NW = NT / 64;
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  __syncthreads();  // previous users of L are done, L.tab is visible
  if (light) {
    R.neg = row_load_neg<C>(row, cols, neg_T, (int)threadIdx.x * C);
#pragma unroll
    for (int i = 0; i < C; i++) {
      R.h[i] = 0.0f;
      R.ordh[i] = 0u;
    }
  } else {
    row_load_h(L, row, cols, mean, maxg, rcp_maxg, R.h, R.neg);
#pragma unroll
    for (int i = 0; i < C; i++) R.ordh[i] = ord_f32(R.h[i] + 0.0f);  // (-0.0) + 0.0 = +0.0
    if (edge_wave<C>(cols)) {
#pragma unroll
      for (int i = 0; i < C; i++)
        if ((int)threadIdx.x * C + i >= cols) R.ordh[i] = 0u;  // walls
    }
  }
  const unsigned cnt = (unsigned)__popc(~R.neg & FULL);
  const unsigned incl = wave_incl_add(cnt, lane);
  if (lane == 63) L.nn_cnt[NW + w] = incl;
  __syncthreads();
  unsigned before = 0;
#pragma unroll
  for (int k = 0; k < NW; k++) before += L.nn_cnt[NW + w - 1 - k];
  R.excl_nn = before + incl - cnt;
}

// bit i: pixel i of the thread's chunk, when visited, opens a new region.  Contains one block barrier.
template <int C, int NT>
__device__ __forceinline__ unsigned row_opens(RowLds<C, NT> &L, const RowChunk<C> &R, int cols) {
  constexpr unsigned FULL = (C == 32) ? 0xffffffffu : ((1u << C) - 1u);
  constexpr int NW = NT / 64;
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const unsigned nn = ~R.neg & FULL;
  // thread-local runs: fl[i] = forward key maximum over the chunk's pixels before i, bl[i] = backward over those after i
  double fl[C], bl[C];
  double run = 0.0;
#pragma unroll
  for (int i = 0; i < C; i++) {
    fl[i] = run;
    const unsigned seg = R.excl_nn + (unsigned)__popc(nn & ((2u << i) - 1u));  // non-neg pixels in [0, p]
    run = kmax(run, sc_key(seg, R.ordh[i]));
  }
  const double ftot = run;
  run = 0.0;
#pragma unroll
  for (int i = C - 1; i >= 0; i--) {
    bl[i] = run;
    const unsigned seg = (unsigned)(NT * C) - R.excl_nn - (unsigned)__popc(nn & ((1u << i) - 1u));  // non-neg pixels in [p, end] + those of the padding (any constant)
    run = kmax(run, sc_key(seg, R.ordh[i]));
  }
  const double btot = run;
  // across the threads of the wavefront (exclusive: shifted by one lane), then across the wavefronts through LDS
  const double fi = wave_incl_max<false>(ftot, lane), bi = wave_incl_max<true>(btot, lane);
  double fx = dpp_f64<0x138>(fi), bx = dpp_f64<0x130>(bi);  // wave_shr:1 / wave_shl:1 (lane 0 / 63: 0.0)
  unsigned eprev = dpp_u32<0x138>((R.neg >> (C - 1)) & 1u);  // previous thread's last pixel neg
  unsigned enext = dpp_u32<0x130>(R.neg & 1u);               // next thread's first pixel neg
  if (lane == 63) L.fwd[NW + w] = fi;
  if (lane == 0) L.bwd[w] = bi;
  const unsigned e0 = (unsigned)__builtin_amdgcn_readlane((int)(R.neg & 1u), 0), e1 = (unsigned)__builtin_amdgcn_readlane((int)((R.neg >> (C - 1)) & 1u), 63);
  if (lane == 0) L.edge[w] = e0 | (e1 << 1);
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NW; k++) {
    fx = kmax(fx, L.fwd[NW + w - 1 - k]);
    bx = kmax(bx, L.bwd[w + 1 + k]);
  }
  if (lane == 0) eprev = w > 0 ? (L.edge[w > 0 ? w - 1 : 0] >> 1) & 1u : 0u;
  if (lane == 63) enext = w + 1 < NT / 64 ? L.edge[w + 1 < NT / 64 ? w + 1 : 0] & 1u : 0u;
  // the comparisons
  // the comparisons, from the last pixel down: x = x + x + (a > b), a compare into vcc and an add-with-carry per pixel (the
  // compiler's form of `x |= (a > b) << i` is compare, select, or)
  unsigned cl = 0, cr = 0;
#pragma unroll
  for (int i = C - 1; i >= 0; i--) {
    const unsigned f = sc_lo(kmax(fx, fl[i])), b = sc_lo(kmax(bx, bl[i]));
    asm("v_cmp_gt_u32_e32 vcc, %1, %2\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(cl) : "v"(R.ordh[i]), "v"(f) : "vcc");  // h(p) >  F(p - 1)
    asm("v_cmp_ge_u32_e32 vcc, %1, %2\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(cr) : "v"(R.ordh[i]), "v"(b) : "vcc");  // h(p) >= G(p + 1)
  }
  const int p0 = threadIdx.x * C;
  const unsigned nn_left = ((nn << 1) | (eprev ? 0u : 1u)) & FULL;               // bit i: pixel p - 1 is non-neg (the row's first pixel: no neighbour, handled below)
  const unsigned nn_right = (nn >> 1) | ((enext ? 0u : 1u) << (C - 1));          // bit i: pixel p + 1 is non-neg (past the row end: walls)
  const unsigned first = p0 == 0 ? 1u : 0u;
  const unsigned left_ok = first | (nn & nn_left) | cl;
  const unsigned right_ok = (nn & nn_right) | cr;
  const unsigned valid = p0 >= cols ? 0u : (p0 + C <= cols ? FULL : ((1u << (cols - p0)) - 1u));
  return left_ok & right_ok & valid;
}

// llrint((double)hv * 2^40) for |hv| <= 1 without fp64: x = hv * 2^40 is exact in fp32 (a power-of-two scaling);
// |x| < 2^31: round to nearest even in fp32 and convert (exact); otherwise x is an integer multiple of 2^8
// (24-bit significand at magnitude >= 2^31): convert x / 2^8 and shift back
__device__ __forceinline__ long long fix40(float hv) {
  const float x = hv * 1099511627776.0f;
  const float ax = fabsf(x);
  const long long small = (long long)__float2int_rn(x);
  const float q = ax * 0.00390625f;  // |hv| = 1 exactly gives q = 2^32: outside uint32
  const long long big = q >= 4294967296.0f ? (1ll << 40) : (long long)((unsigned long long)__float2uint_rz(q) << 8);
  return ax < 2147483648.0f ? small : (x < 0.0f ? -big : big);
}

// bin of h in the selection histogram: monotone non-decreasing in h, which is all the selection needs.  |h| <= 1, so
// (h + 1) * 2048 lies in [0, 4096]: the conversion truncates (= floor, the value is not negative; -0.0 + 1 = 1 like +0.0)
__device__ __forceinline__ int h_bin(float hv) {
  const unsigned b = __float2uint_rz(__fmul_rn(__fadd_rn(hv, 1.0f), 2048.0f));  // (saturating: a negative value gives 0)
  return (int)(b > (unsigned)(NBIN - 1) ? (unsigned)(NBIN - 1) : b);
}

// what cen_hist leaves per thread for cen_collect: the opener bits of its C pixels | topq << (8 or 16), topq = ceil((1 + the
// highest histogram bin an opener of the thread fell into) / 32) (0: no opener) -- 2 bytes per 8 pixels; cen_runs leaves the C
// mark bits (1 byte per 8 pixels).  Threads past the end of the row write nothing.
template <int C> using OpRec = std::conditional_t<(C <= 8), unsigned short, unsigned>;
template <int C> using MarkT = std::conditional_t<(C <= 8), uint8_t, unsigned short>;
template <int C> constexpr int kTopShift = C <= 8 ? 8 : 16;
constexpr int HIST_ROWS = 4;  // azimuths per block of cen_hist in a batch: the table, the zeroed block histogram and its flush (atomics into the image's 4096 bins: a quarter of the kernel at one row per block) once for all of them (8: the same 210 us)
template <int C, int NT>
__global__ __launch_bounds__(NT) void cen_hist(const uint8_t *__restrict__ imgs, int64_t img_stride, int rows, int cols, int stride, int off,
                                               Scal *scal, unsigned *__restrict__ hist, OpRec<C> *__restrict__ opener,
                                               unsigned *__restrict__ wavemax, int rpb) {
  __shared__ RowLds<C, NT> L;
  __shared__ __attribute__((aligned(16))) unsigned s_hist[NBIN];
  __shared__ long long s_fix[NT / 64];
  Scal *sc = scal + blockIdx.y;
  unsigned *gh = hist + (size_t)blockIdx.y * NBIN;
  const float mean = sc->mean, maxg = __uint_as_float(sc->max_g_bits), rcp_maxg = sc->rcp_maxg;
  row_table(L);
  for (int b = threadIdx.x; b < NBIN / 4; b += NT) reinterpret_cast<uint4 *>(s_hist)[b] = uint4{0u, 0u, 0u, 0u};
  // sum of llrint(h * 2^40) over the rows without 64-bit conversions: x = h * 2^20 (exact), hi = rint(x), x - hi is exact and at
  // most 1/2, lo = rint((x - hi) * 2^20); hi * 2^20 is an even integer, so rint(h * 2^40) = hi * 2^20 + lo (== fix40(h), the
  // emulated 64-bit form this replaces).  A wave's sums of hi and lo stay below 2^29 and 2^28 per row (summed over the wavefront after every second row).
  // rint by the 1.5 * 2^23 trick (|x| <= 2^20): fl(x + M) = M + rint(x) with ties to even, and ITS BITS are bits(M) + rint(x)
  // -- the bits are summed as they are, the bits(M) come off at the end (mod 2^32): six instructions a pixel, not nine
  constexpr float M = 12582912.0f, S = 1048576.0f;
  unsigned ahi = 0, alo = 0;
  long long fix = 0;
  for (int rr = 0; rr < rpb; rr++) {
    const int a = blockIdx.x * rpb + rr;
    if (a >= rows) break;  // (uniform)
    const uint8_t *row = imgs + (int64_t)blockIdx.y * img_stride + (int64_t)a * stride + off;
    RowChunk<C> R;
    row_chunk(L, row, cols, mean, maxg, rcp_maxg, R);
    const unsigned opens = row_opens(L, R, cols);
    {  // for cen_runs: the largest ord(h) among the wavefront's pixels of this row -- a wavefront whose maximum lies below the
       // limit's holds no hit and never looks at its h again (walls read 0)
      unsigned m = R.ordh[0];
#pragma unroll
      for (int i = 1; i < C; i++) m = R.ordh[i] > m ? R.ordh[i] : m;
      m = wave_max_u32(m);
      if ((threadIdx.x & 63) == 0) wavemax[((int64_t)blockIdx.y * rows + a) * (NT / 64) + (threadIdx.x >> 6)] = m;
    }
    int top = 0;  // 1 + the highest histogram bin an opener of this thread fell into (0: the thread has no opener)
#pragma unroll
    for (int i = 0; i < C; i++) {  // (past the row end h = 0 and no opener bit is set)
      const float t = __fmaf_rn(R.h[i], S, M);
      ahi += __float_as_uint(t) - 0x4B400000u;
      const float r = __fmaf_rn(R.h[i], S, -__fsub_rn(t, M));  // x - rint(x), exact
      alo += __float_as_uint(__fmaf_rn(r, S, M)) - 0x4B400000u;
      if ((opens >> i) & 1u) {
        const int b = h_bin(R.h[i]);
        atomicAdd(&s_hist[b], 1u);
        top = b + 1 > top ? b + 1 : top;
      }
    }
    // all the later passes need of this evaluation: the opener bits, and (cen_collect looks only at threads that can hold an
    // opener of the selected bin) how high the thread's openers reach
    if ((int)threadIdx.x * C < cols)
      opener[((int64_t)blockIdx.y * rows + a) * NT + threadIdx.x] = (OpRec<C>)(opens | ((unsigned)((top + 31) >> 5) << kTopShift<C>));
    if ((rr & 1) || rr + 1 == rpb || a + 1 >= rows) {  // (uniform)
      const int fhi = wave_sum_i32((int)ahi);  // (DPP + v_readlane: the xor butterfly was twelve ds_bpermute round trips)
      const int flo = wave_sum_i32((int)alo);
      fix += ((long long)fhi << 20) + (long long)flo;
      ahi = alo = 0;
    }
  }
  if ((threadIdx.x & 63) == 0) s_fix[threadIdx.x >> 6] = fix;
  __syncthreads();
  // (one bin per lane: a wavefront's atomics fall into two cache lines.  Four consecutive bins per lane -- one ds_read_b128 --
  // spread every atomic instruction over eight lines and made the kernel three times slower.)
  for (int b = threadIdx.x; b < NBIN; b += NT)
    if (s_hist[b]) atomicAdd(&gh[b], s_hist[b]);
  if (threadIdx.x == 0) {
    long long f = 0;
    for (int w = 0; w < NT / 64; w++) f += s_fix[w];
    atomicAdd(reinterpret_cast<unsigned long long *>(&sc->fix_sum), (unsigned long long)f);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// cen_hist, ONE WAVEFRONT PER AZIMUTH (round 6; batches, rows of <= 4096 bins; the reasoning of cen_runs_wave below: no
// barrier per row, no idle eighth wavefront, no scan total through LDS).  A wavefront walks its row in chunks of 512 bins:
//   forward   bytes -> h, ord(h), sign bits; non-neg prefix counts; the forward maxima F and the comparisons h(p) > F(p - 1);
//             fixed-point sum of h; per thread the sign bits and the bin of its largest h (for cen_runs_wave); ord(h), sign bits
//             and counts stay in registers
//   backward  (last chunk first) the backward maxima G, h(p) >= G(p + 1), the opener bits, their histogram bins (h back
//             from ord(h): the same bin), the records
// The scan totals travel from chunk to chunk in SGPRs.  Four azimuths (wavefronts) per workgroup share the 4096-bin LDS
// histogram and its flush, as HIST_ROWS = 4 did.  Same records, histogram and sums as cen_hist.
// ---------------------------------------------------------------------------------------------------------------
constexpr int HW_WAVES = 4;  // wavefronts (= azimuths) per workgroup
constexpr int HW_CH = 8;     // chunks of 512 bins: rows of <= 4096 bins
#ifndef HW_OCC
#define HW_OCC 3
#endif
__global__ __launch_bounds__(64 * HW_WAVES) __attribute__((amdgpu_waves_per_eu(HW_OCC))) void cen_hist_wave(
    const uint8_t *__restrict__ imgs, int64_t img_stride, int rows, int cols, int stride, int off, Scal *scal, unsigned *__restrict__ hist,
    OpRec<8> *__restrict__ opener, unsigned short *__restrict__ negmax) {
  constexpr int C = 8, NTR = 64 * HW_CH;
  constexpr unsigned FULL = 0xffu;
  __shared__ RowLds<C, NTR> L;  // (the byte -> float table)
  __shared__ __attribute__((aligned(16))) unsigned s_hist[NBIN];
  __shared__ long long s_fix[HW_WAVES];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int img = blockIdx.y, a = (int)blockIdx.x * HW_WAVES + w;
  Scal *sc = scal + img;
  unsigned *gh = hist + (size_t)img * NBIN;
  const float mean = sc->mean, maxg = __uint_as_float(sc->max_g_bits), rcp_maxg = sc->rcp_maxg;
  row_table(L);
  for (int b = threadIdx.x; b < NBIN / 4; b += 64 * HW_WAVES) reinterpret_cast<uint4 *>(s_hist)[b] = uint4{0u, 0u, 0u, 0u};
  __syncthreads();
  long long fix = 0;
  if (a < rows) {  // (wave-uniform)
    const uint8_t *row = imgs + (int64_t)img * img_stride + (int64_t)a * stride + off;
    const int nch = (cols + 64 * C - 1) / (64 * C);
    constexpr float M = 12582912.0f, S = 1048576.0f;  // (cen_hist: the fixed-point sum)
    unsigned ahi = 0, alo = 0;
    unsigned neg[HW_CH], excl[HW_CH], cl[HW_CH], ordh[HW_CH][C];
    unsigned carry_nn = 0;
    double f_carry = 0.0;
    // ---- forward ----
#pragma unroll
    for (int ch = 0; ch < HW_CH; ch++) {
      neg[ch] = 0u;
      excl[ch] = 0u;
      cl[ch] = 0u;
#pragma unroll
      for (int i = 0; i < C; i++) ordh[ch][i] = 0u;
      if (ch < nch) {  // (uniform)
        const int p0 = (ch * 64 + lane) * C;
        float h[C];
        const bool edge = ch == 0 || (ch + 1) * 64 * C >= cols;
        row_load_h_at(L, row, cols, mean, maxg, rcp_maxg, h, neg[ch], p0, edge);
#pragma unroll
        for (int i = 0; i < C; i++) ordh[ch][i] = ord_f32(h[i] + 0.0f);  // (-0.0) + 0.0 = +0.0
        if (edge) {
#pragma unroll
          for (int i = 0; i < C; i++)
            if (p0 + i >= cols) ordh[ch][i] = 0u;  // walls
        }
#pragma unroll
        for (int i = 0; i < C; i++) {  // (past the row end h = 0)
          const float t = __fmaf_rn(h[i], S, M);
          ahi += __float_as_uint(t) - 0x4B400000u;
          const float r = __fmaf_rn(h[i], S, -__fsub_rn(t, M));  // x - rint(x), exact
          alo += __float_as_uint(__fmaf_rn(r, S, M)) - 0x4B400000u;
        }
        if ((ch & 1) || ch + 1 >= nch) {  // (uniform) 16 pixels a lane at most between two sums, as in cen_hist
          const int fhi = wave_sum_i32((int)ahi), flo = wave_sum_i32((int)alo);
          fix += ((long long)fhi << 20) + (long long)flo;
          ahi = alo = 0;
        }
        const unsigned nn = ~neg[ch] & FULL;
        const unsigned cnt = (unsigned)__popc(nn);
        const unsigned incl = wave_incl_add(cnt, lane);
        excl[ch] = carry_nn + incl - cnt;
        carry_nn += (unsigned)__builtin_amdgcn_readlane((int)incl, 63);
        // forward maxima (row_opens): thread-local, across the lanes, across the chunks
        double fl[C];
        double run = 0.0;
#pragma unroll
        for (int i = 0; i < C; i++) {
          fl[i] = run;
          const unsigned seg = excl[ch] + (unsigned)__popc(nn & ((2u << i) - 1u));  // non-neg pixels in [0, p]
          run = kmax(run, sc_key(seg, ordh[ch][i]));
        }
        const double fi = wave_incl_max<false>(run, lane);
        const double fx = kmax(dpp_f64<0x138>(fi), f_carry);
        f_carry = kmax(f_carry, readlane_f64(fi, 63));
        unsigned c = 0;
#pragma unroll
        for (int i = C - 1; i >= 0; i--) {
          const unsigned f = sc_lo(kmax(fx, fl[i]));
          asm("v_cmp_gt_u32_e32 vcc, %1, %2\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(c) : "v"(ordh[ch][i]), "v"(f) : "vcc");  // h(p) >  F(p - 1)
        }
        cl[ch] = c;
      }
    }
    // ---- backward ----
    double b_carry = 0.0;
#pragma unroll
    for (int ch = HW_CH - 1; ch >= 0; ch--) {
      if (ch < nch) {
        const int p0 = (ch * 64 + lane) * C;
        const unsigned nn = ~neg[ch] & FULL;
        double bl[C];
        double run = 0.0;
#pragma unroll
        for (int i = C - 1; i >= 0; i--) {
          bl[i] = run;
          const unsigned seg = (unsigned)(NTR * C) + 64u - excl[ch] - (unsigned)__popc(nn & ((1u << i) - 1u));  // a constant minus the non-neg pixels before p: larger to the left
          run = kmax(run, sc_key(seg, ordh[ch][i]));
        }
        const double bi = wave_incl_max<true>(run, lane);
        const double bx = kmax(dpp_f64<0x130>(bi), b_carry);
        b_carry = kmax(b_carry, readlane_f64(bi, 0));
        unsigned cr = 0;
#pragma unroll
        for (int i = C - 1; i >= 0; i--) {
          const unsigned b = sc_lo(kmax(bx, bl[i]));
          asm("v_cmp_ge_u32_e32 vcc, %1, %2\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(cr) : "v"(ordh[ch][i]), "v"(b) : "vcc");  // h(p) >= G(p + 1)
        }
        unsigned eprev = dpp_u32<0x138>((neg[ch] >> (C - 1)) & 1u);  // previous thread's last pixel neg
        unsigned enext = dpp_u32<0x130>(neg[ch] & 1u);               // next thread's first pixel neg
        const unsigned pv = ch > 0 ? (unsigned)__builtin_amdgcn_readlane((int)((neg[ch > 0 ? ch - 1 : 0] >> (C - 1)) & 1u), 63) : 0u;
        const unsigned nx = ch + 1 < HW_CH ? (unsigned)__builtin_amdgcn_readlane((int)(neg[ch + 1 < HW_CH ? ch + 1 : ch] & 1u), 0) : 0u;
        if (lane == 0) eprev = pv;
        if (lane == 63) enext = nx;
        const unsigned nn_left = ((nn << 1) | (eprev ? 0u : 1u)) & FULL;       // bit i: pixel p - 1 is non-neg
        const unsigned nn_right = (nn >> 1) | ((enext ? 0u : 1u) << (C - 1));  // bit i: pixel p + 1 is non-neg (past the row end: walls)
        const unsigned first = p0 == 0 ? 1u : 0u;
        const unsigned left_ok = first | (nn & nn_left) | cl[ch];
        const unsigned right_ok = (nn & nn_right) | cr;
        const unsigned valid = p0 >= cols ? 0u : (p0 + C <= cols ? FULL : ((1u << (cols - p0)) - 1u));
        const unsigned opens = left_ok & right_ok & valid;
        int top = 0;
#pragma unroll
        for (int i = 0; i < C; i++) {
          if ((opens >> i) & 1u) {
            const unsigned o = ordh[ch][i];
            const float hv = __uint_as_float((o & 0x80000000u) ? (o ^ 0x80000000u) : ~o);  // ord_f32 backwards (-0.0 came in as +0.0: the same bin)
            const int b = h_bin(hv);
            atomicAdd(&s_hist[b], 1u);
            top = b + 1 > top ? b + 1 : top;
          }
        }
        if (p0 < cols) {  // for cen_runs_wave: the chunk's sign bits and how high its h reaches (histogram bin / 16 of the largest h:
                          // monotone in h), 2 bytes per 8 pixels -- cen_runs_wave looks at the image bytes of the few threads
                          // that can hold a hit only
          unsigned m = ordh[ch][0];
#pragma unroll
          for (int i = 1; i < C; i++) m = ordh[ch][i] > m ? ordh[ch][i] : m;
          const float hm = __uint_as_float((m & 0x80000000u) ? (m ^ 0x80000000u) : ~m);  // ord_f32 backwards
          negmax[((int64_t)img * rows + a) * NTR + ch * 64 + lane] = (unsigned short)(neg[ch] | ((unsigned)(h_bin(hm) >> 4) << 8));
        }
        if (p0 < cols) opener[((int64_t)img * rows + a) * NTR + ch * 64 + lane] = (OpRec<C>)(opens | ((unsigned)((top + 31) >> 5) << kTopShift<C>));
      }
    }
  }
  if (lane == 0) s_fix[w] = fix;
  __syncthreads();
  for (int b = threadIdx.x; b < NBIN; b += 64 * HW_WAVES)
    if (s_hist[b]) atomicAdd(&gh[b], s_hist[b]);
  if (threadIdx.x == 0) {
    long long f = 0;
    for (int ww = 0; ww < HW_WAVES; ww++) f += s_fix[ww];
    atomicAdd(reinterpret_cast<unsigned long long *>(&sc->fix_sum), (unsigned long long)f);
  }
}

// one block of 256 threads per image: the bin of the max_points-th opener
__global__ __launch_bounds__(256) void cen_pick(Scal *scal, const unsigned *__restrict__ hist, int max_points) {
  constexpr int NT = 256;
  __shared__ unsigned s_cnt[NT / 64];
  Scal *sc = scal + blockIdx.x;
  const unsigned *gh = hist + (size_t)blockIdx.x * NBIN;
  // thread t owns the NBIN / NT bins [NBIN - (t+1) per, NBIN - t per): thread 0 = the highest h
  constexpr int PER = NBIN / NT;
  unsigned mine = 0;
  unsigned hv[PER];
#pragma unroll
  for (int j = 0; j < PER; j++) {
    hv[j] = gh[NBIN - 1 - (threadIdx.x * PER + j)];
    mine += hv[j];
  }
  unsigned incl = mine;  // inclusive prefix over threads (descending h)
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned o = __shfl_up(incl, d);
    if ((threadIdx.x & 63) >= d) incl += o;
  }
  if ((threadIdx.x & 63) == 63) s_cnt[threadIdx.x >> 6] = incl;
  __syncthreads();
  unsigned before = 0;
  for (int w = 0; w < (int)(threadIdx.x >> 6); w++) before += s_cnt[w];
  incl += before;
  const unsigned excl = incl - mine;
  unsigned total = 0;
  for (int w = 0; w < NT / 64; w++) total += s_cnt[w];
  if (max_points <= 0 || total < (unsigned)max_points) {
    if (threadIdx.x == 0) {
      sc->bstar = -1;
      sc->above = total;
    }
  } else if (excl < (unsigned)max_points && incl >= (unsigned)max_points) {
    unsigned c = excl;
#pragma unroll
    for (int j = 0; j < PER; j++) {
      if (c + hv[j] >= (unsigned)max_points) {
        sc->bstar = NBIN - 1 - (threadIdx.x * PER + j);
        sc->above = c;
        break;
      }
      c += hv[j];
    }
  }
}

// h of ONE pixel straight from the bytes (the same operations in the same order as row_load_h: the same float)
__device__ __forceinline__ float pixel_h(const uint8_t *__restrict__ row, int cols, int p, float mean, float maxg, float rcp_maxg) {
  const float f0 = __fdiv_rn((float)row[p], 255.0f);
  float g = 0.0f;
  if (cols > 1) {  // reflect 101
    const float fp = __fdiv_rn((float)row[p + 1 < cols ? p + 1 : p - 1], 255.0f), fm = __fdiv_rn((float)row[p >= 1 ? p - 1 : p + 1], 255.0f);
    g = fabsf(__fsub_rn(fp, fm));
  }
  const float q0 = __fmul_rn(g, rcp_maxg);
  const float gn = (maxg > 0.0f) ? __fmaf_rn(__fmaf_rn(-maxg, q0, g), rcp_maxg, q0) : 0.0f;
  return __fmul_rn(__fsub_rn(f0, mean), __fsub_rn(1.0f, gn));
}

// the openers of histogram bin B* (a few dozen keys per image) -> list.  Round 5: cen_hist left, per thread, the highest bin
// an opener of the thread fell into; only the threads that reach B* are looked at (one in twelve on the bench images), their
// opener pixels are queued in LDS and evaluated one pixel per lane, straight from the bytes.  Before: h of EVERY pixel again
// (1807 VALU instructions per wavefront for eight rows, 80 us per 64 images).  A WAVEFRONT per azimuth, no workgroup barrier:
// the wavefront walks the row's records 64 at a time, queues the opener pixels of the threads that qualify (positions by a
// DPP prefix sum) in its own LDS segment and evaluates them; the first build took a workgroup per azimuth and two barriers.
template <int C, int NT>
__global__ __launch_bounds__(NT) void cen_collect(const uint8_t *__restrict__ imgs, int64_t img_stride, int rows, int cols, int stride,
                                                  int off, Scal *scal, const OpRec<C> *__restrict__ opener, unsigned long long *__restrict__ lists,
                                                  int64_t list_stride, int rpb) {
  constexpr int NW = NT / 64, QCAP = 2 * 64 * C;  // (a step of 64 records adds at most 64 C pixels)
  __shared__ unsigned short s_q[NW][QCAP];
  Scal *sc = scal + blockIdx.y;
  unsigned long long *list = lists + (int64_t)blockIdx.y * list_stride;
  const int bstar = sc->bstar;
  if (bstar < 0) return;  // fewer openers than the budget: nothing to select
  const float mean = sc->mean, maxg = __uint_as_float(sc->max_g_bits), rcp_maxg = sc->rcp_maxg;
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  for (int rr = w; rr < rpb; rr += NW) {
    const int a = blockIdx.x * rpb + rr;
    if (a >= rows) break;  // (uniform in the wavefront)
    const uint8_t *row = imgs + (int64_t)blockIdx.y * img_stride + (int64_t)a * stride + off;
    unsigned n = 0;
    auto drain = [&]() {
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      for (unsigned j = lane; j < n; j += 64) {
        const int p = s_q[w][j];
        const float hv = pixel_h(row, cols, p, mean, maxg, rcp_maxg);
        if (h_bin(hv) == bstar) list[atomicAdd(&sc->n_list, 1u)] = key_of(hv, (unsigned)a * (unsigned)cols + (unsigned)p);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      n = 0;
    };
    for (int t0 = 0; t0 * C < cols; t0 += 64) {
      const int t = t0 + lane;
      const unsigned rec = t * C < cols ? (unsigned)opener[((int64_t)blockIdx.y * rows + a) * NT + t] : 0u;
      unsigned op = ((int)((rec >> kTopShift<C>) << 5) > bstar) ? rec & ((1u << kTopShift<C>) - 1u) : 0u;  // 1 + highest opener bin (rounded up to 32) >= B* + 1
      const unsigned cnt = (unsigned)__popc(op), incl = wave_incl_add(cnt, lane);
      const unsigned total = (unsigned)__builtin_amdgcn_readlane((int)incl, 63);
      if (total == 0) continue;              // (uniform)
      if (n + total > (unsigned)QCAP) drain();  // (uniform)
      for (unsigned j = n + incl - cnt; op; op &= op - 1, j++) s_q[w][j] = (unsigned short)(t * C + __builtin_ctz(op));
      n += total;
    }
    drain();
  }
}

// one block of 1024 threads per image: K* = the max_points-th smallest opener key, and the mark limit
__global__ __launch_bounds__(1024) void cen_resolve(Scal *scal, const unsigned long long *__restrict__ lists, int64_t list_stride, int rows,
                                                   int cols, int max_points) {
  constexpr int NT = 1024;
  __shared__ unsigned s_h[256];
  __shared__ unsigned long long s_prefix;
  __shared__ unsigned s_t;
  Scal *sc = scal + blockIdx.x;
  const unsigned long long *list = lists + (int64_t)blockIdx.x * list_stride;
  const int bstar = sc->bstar;
  const unsigned long long kmean = kmean_of(mean_h_of(sc->fix_sum, (int64_t)rows * cols));
  if (max_points <= 0) {
    if (threadIdx.x == 0) sc->klimit = 0ull;  // while (0 < 0 ...) never runs: nothing is marked
    return;
  }
  if (bstar < 0) {  // fewer openers than the budget: every candidate is visited
    if (threadIdx.x == 0) sc->klimit = kmean;
    return;
  }
  const unsigned n = sc->n_list;
  if (n <= 256) {
    // the usual case (a few dozen openers in the bin): every thread takes one key and counts the smaller ones; the key
    // of rank t - 1 is K* (keys are distinct).  One global load per thread, n broadcast LDS reads, no passes.
    __shared__ unsigned long long s_k[256];
    const unsigned long long mine = threadIdx.x < n ? list[threadIdx.x] : KINF;
    if (threadIdx.x < 256) s_k[threadIdx.x] = mine;
    __syncthreads();
    unsigned rank = 0;
    for (unsigned i = 0; i < n; i++) rank += s_k[i] < mine ? 1u : 0u;
    const unsigned t = (unsigned)max_points - sc->above;
    if (threadIdx.x < n && rank == t - 1u) sc->klimit = (mine == KINF || mine + 1ull > kmean) ? kmean : mine + 1ull;
    return;
  }
  // MSD radix select (8 x 8 bits) of the t-th smallest key of the list, t 1-based
  if (threadIdx.x == 0) {
    s_prefix = 0ull;
    s_t = (unsigned)max_points - sc->above;
  }
  // (lists of up to 32 keys per thread -- the odometry drive's images put a few thousand tied openers into the selected bin --
  // are read ONCE and kept in registers for the eight passes: re-reading them pass by pass was 100 us per 64-scan window)
  constexpr int RK = 32;
  const bool in_regs = n <= (unsigned)(RK * NT);
  unsigned long long rk[RK];
  if (in_regs) {
#pragma unroll
    for (int e = 0; e < RK; e++) {
      const unsigned i = threadIdx.x + (unsigned)e * NT;
      rk[e] = i < n ? list[i] : KINF;
    }
  }
  // bytes that ALL keys share need no counting pass (tied openers: the four bytes of h and the top byte of the pixel index --
  // five of the eight passes, 11 us each with 15 000 keys)
  __shared__ unsigned long long s_and[NT / 64], s_or[NT / 64];
  unsigned long long differ;
  {
    unsigned long long va = ~0ull, vo = 0ull;
    if (in_regs) {
#pragma unroll
      for (int e = 0; e < RK; e++)
        if (threadIdx.x + (unsigned)e * NT < n) {
          va &= rk[e];
          vo |= rk[e];
        }
    } else {
      for (unsigned i = threadIdx.x; i < n; i += NT) {
        const unsigned long long k = list[i];
        va &= k;
        vo |= k;
      }
    }
    for (int o = 32; o >= 1; o >>= 1) {
      va &= __shfl_xor(va, o);
      vo |= __shfl_xor(vo, o);
    }
    if ((threadIdx.x & 63) == 0) {
      s_and[threadIdx.x >> 6] = va;
      s_or[threadIdx.x >> 6] = vo;
    }
    __syncthreads();
    va = ~0ull;
    vo = 0ull;
    for (int w = 0; w < NT / 64; w++) {
      va &= s_and[w];
      vo |= s_or[w];
    }
    differ = va ^ vo;
    if (threadIdx.x == 0) s_prefix = va & ~differ;  // (the shared bytes are the prefix's; the passes below fill in the others)
    __syncthreads();
  }
  unsigned long long mask = 0ull;
  for (int pass = 0; pass < 8; pass++) {
    const int shift = 56 - 8 * pass;
    if (((differ >> shift) & 255ull) == 0ull) {  // every key has this byte (uniform)
      mask |= 255ull << shift;
      continue;
    }
    if (threadIdx.x < 256) s_h[threadIdx.x] = 0u;
    __syncthreads();
    const unsigned long long prefix = s_prefix & mask;  // (s_prefix already holds the shared bytes further down)
    // (tied openers share their h: in the four passes over h's bytes every key falls into the SAME bin, and thousands of LDS
    // atomics on one address are served one after the other -- 80 us.  A wavefront adds once per distinct digit.)
    auto count = [&](bool on, unsigned long long key) {
      const unsigned digit = (unsigned)(key >> shift) & 255u;
      unsigned long long peers = __ballot(on);
      if (peers == 0ull) return;  // (uniform)
#pragma unroll
      for (int bit = 0; bit < 8; bit++) {
        const bool one = (digit >> bit) & 1u;
        const unsigned long long bal = __ballot(on && one);
        peers &= one ? bal : ~bal;
      }
      if (on && (peers & ((1ull << (threadIdx.x & 63)) - 1ull)) == 0ull) atomicAdd(&s_h[digit], (unsigned)__popcll(peers));
    };
    if (in_regs) {
#pragma unroll
      for (int e = 0; e < RK; e++) count(threadIdx.x + (unsigned)e * NT < n && (rk[e] & mask) == prefix, rk[e]);
    } else {
      for (unsigned i0 = 0; i0 < n; i0 += NT) {  // (whole wavefronts: the ballots need every lane)
        const unsigned i = i0 + threadIdx.x;
        const unsigned long long k = i < n ? list[i] : KINF;
        count(i < n && (k & mask) == prefix, k);
      }
    }
    __syncthreads();
    if (threadIdx.x < 64) {  // lane l owns digits 4l .. 4l+3
      const unsigned l = threadIdx.x;
      const unsigned c0 = s_h[4 * l], c1 = s_h[4 * l + 1], c2 = s_h[4 * l + 2], c3 = s_h[4 * l + 3];
      const unsigned mine = c0 + c1 + c2 + c3;
      unsigned incl = mine;
      for (int d = 1; d < 64; d <<= 1) {
        const unsigned o = __shfl_up(incl, d);
        if (l >= (unsigned)d) incl += o;
      }
      const unsigned t = s_t, excl = incl - mine;
      if (excl < t && incl >= t) {
        unsigned c = excl, dig;
        if (c + c0 >= t) dig = 0;
        else if ((c += c0) + c1 >= t) dig = 1;
        else if ((c += c1) + c2 >= t) dig = 2;
        else { c += c2; dig = 3; }
        s_t = t - c;
        s_prefix |= (unsigned long long)(4 * l + dig) << shift;
      }
    }
    mask |= 255ull << shift;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const unsigned long long kstar = s_prefix;  // the key of the candidate that opens region number max_points
    sc->klimit = (kstar == KINF || kstar + 1ull > kmean) ? kmean : kstar + 1ull;
  }
}

// runs of marked pixels of ONE azimuth (the part of the extraction that does not look at the neighbours).  Marks (round 5): a
// pixel is marked in the end iff its mark key MK(p) -- its own key, or for a neg pixel the smallest key of its run and the
// two pixels beside it -- is below the limit.  With the limit KNOWN that is no minimum any more but an OR: with
// hit(q) = key(q) < limit, a non-neg pixel is marked iff it is a hit, a neg pixel iff ANY pixel of its run or one of the two
// neighbours is.  Runs are numbered by the non-neg pixels before them (one integer prefix sum per row), a hit sets the flag
// byte of the run(s) it touches, a neg pixel reads its run's flag: no keys of other pixels, no marker read, no gather.
// Then one segmented max-scan over the marked runs at r >= min_range: every run that an unmarked pixel closes is recorded as
// (first bin, last bin, bin of the first maximum of h); the row's mark bits go to HBM for the neighbours' adjacency test.
constexpr int RUNS_ROWS = 2;  // azimuths per block of cen_runs in a batch
#ifndef CEN_RUNS_WAVES
#define CEN_RUNS_WAVES 8  // waves per SIMD the register budget of cen_runs<8, 512> is cut for (8: 62 VGPRs, four row blocks per CU)
#endif
template <int C, int NT>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(NT <= 512 ? CEN_RUNS_WAVES : 4))) void cen_runs(const uint8_t *__restrict__ imgs, int64_t img_stride, int rows, int cols, int stride, int off,
                                               Scal *scal, int min_range, int row_cap, uint2 *__restrict__ row_runs,
                                               unsigned *__restrict__ row_nruns, MarkT<C> *__restrict__ markbits,
                                               const unsigned *__restrict__ wavemax, int rpb) {
  constexpr unsigned FULL = (C == 32) ? 0xffffffffu : ((1u << C) - 1u);
  __shared__ RowLds<C, NT> L;
  __shared__ __attribute__((aligned(16))) uint8_t s_run[C * NT + 16];  // flag of neg run number k: some toucher is a hit
  constexpr int NW = NT / 64;
  __shared__ double s_k[2 * NW];                    // [NW + w]; [0, NW): zeros (RowLds: no selects on the reading side)
  __shared__ unsigned s_cnt[2 * NW], s_cnt2[2 * NW];
  const int img = blockIdx.y;
  Scal *sc = scal + img;
  const float mean = sc->mean, maxg = __uint_as_float(sc->max_g_bits), rcp_maxg = sc->rcp_maxg;
  const unsigned long long klimit = sc->klimit;
  row_table(L);
  // a pixel can be a hit (key < limit) only with ord(h) >= ~(limit >> 32); s < 0 <=> byte < neg_T (row_load_neg)
  const unsigned ord_min = ~(unsigned)(klimit >> 32);
  const unsigned neg_T = (unsigned)__syncthreads_count(threadIdx.x < 256 && L.tab[threadIdx.x < 256 ? threadIdx.x : 0] < mean);  // (a thread reads the entry it wrote)
  const int p0 = threadIdx.x * C;
  if (threadIdx.x < NW) {
    s_k[threadIdx.x] = 0.0;
    s_cnt[threadIdx.x] = 0u;
    s_cnt2[threadIdx.x] = 0u;
  }
  for (int rr = 0; rr < rpb; rr++) {  // (rpb azimuths per block in a batch: the table and the block's start-up once)
    const int a = blockIdx.x * rpb + rr;
    if (a >= rows) break;  // (uniform)
    const uint8_t *row = imgs + (int64_t)img * img_stride + off + (int64_t)a * stride;
    {  // (the readers of the previous azimuth's flags are three barriers behind)
      uint4 *z = reinterpret_cast<uint4 *>(s_run);
      for (int i = threadIdx.x; i < (C * NT + 16) / 16; i += NT) z[i] = uint4{0u, 0u, 0u, 0u};
    }
    RowChunk<C> R;
    // Round 6: hits are sparse (bench images: 0.7 % of the pixels, 37 % of the wavefronts of a row hold one).  cen_hist left the
    // largest ord(h) of every wavefront and row: a wavefront below the limit forms its sign bits from the bytes alone (light:
    // no table, no h) and, with no mark among its pixels either, publishes the identities of the three scans and skips them
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const bool light = (unsigned)__builtin_amdgcn_readfirstlane((int)wavemax[((int64_t)img * rows + a) * NW + w]) < ord_min;
    row_chunk(L, row, cols, mean, maxg, rcp_maxg, R, light, neg_T);  // (its barriers also publish the zeroed flags)
    const unsigned nn = ~R.neg & FULL;
    // hit(p) = key(p) < limit, key = ~ord(h) << 32 | pixel: straight from the chunk's ord(h) (a wall's is 0: the largest key
    // there is, never a hit -- no test against the row's end)
    unsigned hit = 0;
    const unsigned pix0 = (unsigned)a * (unsigned)cols + (unsigned)p0;
    if (!light) {
#pragma unroll
    for (int i = 0; i < C; i++) {
      const unsigned long long key = ((unsigned long long)(~R.ordh[i]) << 32) | (unsigned long long)(pix0 + (unsigned)i);
      if (key < klimit) {
        hit |= 1u << i;
        const unsigned c = R.excl_nn + (unsigned)__popc(nn & ((1u << i) - 1u));  // non-neg pixels before p = the number of the run p is in / that ends at p
        s_run[c] = 1;                                // a neg pixel: its own run; a non-neg pixel: the run on its left ...
        if ((nn >> i) & 1u) s_run[c + 1] = 1;        // ... and the run on its right
      }
    }
    }
    __syncthreads();
    // marks: a non-neg pixel iff it is a hit, a neg pixel iff its run's flag is set (walls are non-neg and no hits)
    unsigned flags = 0;
#pragma unroll
    for (int i = C - 1; i >= 0; i--) {
      const unsigned f = s_run[R.excl_nn + (unsigned)__popc(nn & ((1u << i) - 1u))];
      asm("v_cmp_ne_u32_e32 vcc, 0, %1\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(flags) : "v"(f) : "vcc");
    }
    const unsigned marked = (nn & hit) | (R.neg & flags);
    if (p0 < cols) markbits[((int64_t)img * rows + a) * NT + threadIdx.x] = (MarkT<C>)marked;
    // a wavefront without a marked pixel has nothing to say in the scans below; a LIGHT wavefront with marks (neg pixels of a run
    // that a hit elsewhere flagged) gets the h of its pixels now -- the arg-max of a run may fall on them (a run cut at rmin)
    const bool wave_marks = __ballot(marked != 0u) != 0ull;  // (wave-uniform)
    if (light && wave_marks) {
      unsigned neg2;
      row_load_h(L, row, cols, mean, maxg, rcp_maxg, R.h, neg2);
#pragma unroll
      for (int i = 0; i < C; i++) R.ordh[i] = ord_f32(R.h[i] + 0.0f);
      if (edge_wave<C>(cols)) {
#pragma unroll
        for (int i = 0; i < C; i++)
          if (p0 + i >= cols) R.ordh[i] = 0u;
      }
    }
    // ---- closed runs of marked pixels at r >= rmin: (first bin, last bin, bin of the first maximum of h).  live = marked and
    // r >= rmin; a run starts where the pixel before is not live, and counts once an UNMARKED pixel closes it (a run that
    // reaches the end of the row does not).  Two plain max-scans (sc_key's trick): S(p) = 1 + the bin of the latest start
    // at or before p, then K(p) = S(p) | ord(h) | ~bin in one positive double -- the latest run beats everything before it,
    // inside it the largest h wins and among equals the lowest bin -- so the value at a run's last pixel is the run's result.
    const int rmin = min_range < 0 ? 0 : min_range;
    const unsigned ge = p0 >= rmin ? FULL : (p0 + C <= rmin ? 0u : (FULL & ~((1u << (rmin - p0)) - 1u)));  // bit i: p >= rmin
    const unsigned live = marked & ge;
    unsigned lprev = dpp_u32<0x138>((live >> (C - 1)) & 1u);   // previous thread's last pixel live
    unsigned mnext = dpp_u32<0x130>(marked & 1u);               // next thread's first pixel marked
    {
      const unsigned e0 = (unsigned)__builtin_amdgcn_readlane((int)(marked & 1u), 0), e1 = (unsigned)__builtin_amdgcn_readlane((int)((live >> (C - 1)) & 1u), 63);
      if (lane == 0) L.edge[w] = e0 | (e1 << 1);
    }
    __syncthreads();
    unsigned start = 0, close = 0, sin = 0;
    if (wave_marks) {
      if (lane == 0) lprev = w > 0 ? (L.edge[w > 0 ? w - 1 : 0] >> 1) & 1u : 0u;
      if (lane == 63) mnext = w + 1 < NT / 64 ? L.edge[w + 1 < NT / 64 ? w + 1 : 0] & 1u : 0u;
      start = live & ~(((live << 1) | lprev) & FULL);
      const unsigned in_row = p0 + C < cols ? FULL : (p0 + 1 >= cols ? 0u : ((1u << (cols - 1 - p0)) - 1u));  // bit i: p + 1 < cols
      close = live & ~((marked >> 1) | (mnext << (C - 1))) & in_row;
      // S: 1 + bin of the latest start (0: none yet)
      const unsigned sloc = start ? (unsigned)(p0 + (31 - __builtin_clz(start)) + 1) : 0u;
      const unsigned x = wave_incl_max_u32(sloc);  // inclusive over the wavefront (positions only grow, so max = latest)
      if (lane == 63) s_cnt[NW + w] = x;
      sin = dpp_u32<0x138>(x);  // exclusive
    } else if (lane == 63) {
      s_cnt[NW + w] = 0u;  // (no start in this wavefront)
    }
    __syncthreads();
    // K = 1 << 62 | S << 46 | ord(h) << 14 | ~bin: the high word is one v_alignbit of (S | 1 << 16, ord(h)), the low word one
    // v_lshl_or; a pixel that is not live gets a high word of 0 (a tiny positive double below every key, whatever the low word)
    double kin[C];
    double kx = 0.0;
    unsigned cnt = 0, inc = 0;
    if (wave_marks) {
#pragma unroll
      for (int k = 0; k < NW; k++) {
        const unsigned v = s_cnt[NW + w - 1 - k];
        sin = v > sin ? v : sin;
      }
      double run = 0.0;
      const unsigned sinx = sin | 0x10000u, pbx = (unsigned)p0 + 32u + 0x10000u, c0 = 0x3fffu - ((unsigned)p0 & 0x3fffu);  // (p0 is a multiple of C: no carry into the next 16384)
#pragma unroll
      for (int i = 0; i < C; i++) {
        const unsigned below = start & ((2u << i) - 1u);  // starts of the chunk at or before i
        const unsigned spx = below ? pbx - (unsigned)__builtin_clz(below) : sinx;
        const unsigned hi = __builtin_amdgcn_alignbit(spx, R.ordh[i], 18);
        const unsigned lo = (R.ordh[i] << 14) | (c0 - (unsigned)i);
        const double kd = __hiloint2double((int)(((live >> i) & 1u) ? hi : 0u), (int)lo);
        run = kmax(run, kd);
        kin[i] = run;
      }
      const double ki = wave_incl_max<false>(run, lane);
      kx = dpp_f64<0x138>(ki);
      if (lane == 63) s_k[NW + w] = ki;
      // (the compaction's counts travel with the same barrier)
      cnt = (unsigned)__popc(close);
      inc = wave_incl_add(cnt, lane);
      if (lane == 63) s_cnt2[NW + w] = inc;
    } else {
#pragma unroll
      for (int i = 0; i < C; i++) kin[i] = 0.0;
      if (lane == 63) {
        s_k[NW + w] = 0.0;
        s_cnt2[NW + w] = 0u;
      }
    }
    __syncthreads();
    unsigned before = 0;
#pragma unroll
    for (int k = 0; k < NW; k++) before += s_cnt2[NW + w - 1 - k];
    if (wave_marks) {
#pragma unroll
      for (int k = 0; k < NW; k++) kx = kmax(kx, s_k[NW + w - 1 - k]);
      unsigned pos = before + inc - cnt;
      uint2 *ro = row_runs + ((int64_t)img * rows + a) * row_cap;
#pragma unroll
      for (int i = 0; i < C; i++) {
        if ((close >> i) & 1u) {
          const unsigned long long k = (unsigned long long)__double_as_longlong(kmax(kx, kin[i]));
          const unsigned first = (unsigned)((k >> 46) & 0x7fffu) - 1u, arg = 0x3fffu - (unsigned)(k & 0x3fffu);
          ro[pos++] = uint2{first | ((unsigned)(p0 + i) << 16), arg};  // first | last << 16, arg-max bin
        }
      }
    }
    if (threadIdx.x == NT - 1) row_nruns[(int64_t)img * rows + a] = before + inc;  // the last thread's inclusive count
  }
}

// ---------------------------------------------------------------------------------------------------------------
// cen_runs, ONE WAVEFRONT PER AZIMUTH (round 6; batches, rows of <= 4096 bins).  The workgroup-per-azimuth form above is one chain of
// six barriers per row and every scan crosses its eight wavefronts through LDS; since a wavefront that holds no hit has
// nearly nothing to do (light path), its row time was the time of the busiest wavefront and the others waited (counters:
// 17 % fewer vector instructions, SQ_WAIT_ANY up by the same cycles, the same 174 us).  Here a wavefront walks ITS row in
// chunks of 512 bins, the scans' totals travel from chunk to chunk in SGPRs, the run flags live in the wavefront's own
// 4 KB of LDS -- no barrier after the table, no idle eighth wavefront.  And it does not read the image: cen_hist_wave left,
// per thread (8 bins), the sign bits and the histogram bin / 16 of its largest h; hits are sparse (0.7 % of the pixels, ~14 of
// a row's 420 threads can hold one), so the threads whose record reaches the limit's bin are COMPACTED -- a list in the
// wavefront's LDS -- and one pass of 64 lanes evaluates h for the candidates of the whole row (more than 64: further passes)
// where the chunk-wise form ran the full evaluation for 70 % of the chunks.
//   pass 1a (all chunks)  records -> sign bits, non-neg prefix counts, the candidate list
//   pass 1b               h, ord(h), hits of the listed threads -> run flags; results back to their owners through LDS
//   pass 2a               marks (flag reads), mark bits to HBM
//   pass 2b               closed runs: the three scans chunk by chunk, chunks without a mark skipped
// Same records, same mark bits as cen_runs, which stays for single scans and rows wider than 4096 bins (tests/test_gpu_cen2019.py:
// batches through this kernel, single scans and test_wide_rows through the block form) and for A / B runs (RSX_CEN_FORMS=block,
// experiments build).
// ---------------------------------------------------------------------------------------------------------------
constexpr int RW_WAVES = 4;             // wavefronts (= azimuths) per workgroup
constexpr int RW_CH = 8;                // chunks of 512 bins: rows of <= 4096 bins
constexpr int RW_FLAGS = 4096 + 64;     // run flags of one wavefront (a run's number = the non-neg pixels before it <= 4096 + walls)
struct RwLds {                          // one wavefront's
  uint8_t flags[RW_FLAGS];
  unsigned short list[64 * RW_CH];      // candidate threads (chunk * 64 + lane), in row order
  unsigned cexcl[64 * RW_CH];           // their non-neg prefix counts
  unsigned res[64][10 + 1];             // one pass of 64 candidates: hit bits, ord(h) of the 8 pixels; pass 2b: bits, p0, ord(h) (+ 1: bank spread)
};
#ifndef RW_OCC
#define RW_OCC 3  // waves per SIMD the register budget is cut for (4: spills; 3, 4, 5 measured alike before the candidate list)
#endif
__global__ __launch_bounds__(64 * RW_WAVES) __attribute__((amdgpu_waves_per_eu(RW_OCC))) void cen_runs_wave(const uint8_t *__restrict__ imgs, int64_t img_stride, int rows, int cols, int stride,
                                                                int off, Scal *scal, int min_range, int row_cap, uint2 *__restrict__ row_runs,
                                                                unsigned *__restrict__ row_nruns, MarkT<8> *__restrict__ markbits,
                                                                const unsigned short *__restrict__ negmax) {
  constexpr int C = 8, NTR = 64 * RW_CH;  // NTR: chunks (threads of the block form) a row's records are laid out for
  constexpr unsigned FULL = 0xffu;
  __shared__ RowLds<C, NTR> L;  // (the byte -> float table)
  __shared__ __attribute__((aligned(16))) RwLds s_w[RW_WAVES];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int img = blockIdx.y, a = (int)blockIdx.x * RW_WAVES + w;
  Scal *sc = scal + img;
  const float mean = sc->mean, maxg = __uint_as_float(sc->max_g_bits), rcp_maxg = sc->rcp_maxg;
  const unsigned long long klimit = sc->klimit;
  row_table(L);
  __syncthreads();
  if (a >= rows) return;  // (wave-uniform; no barrier below)
  // a pixel can be a hit (key < limit) only with ord(h) >= ~(limit >> 32), i.e. h >= the limit's h, i.e. (h_bin is monotone) with
  // its bin >= the limit's: the candidate threads are those whose record reaches bin / 16 of the limit (a NaN pattern gives 0: all)
  const unsigned ord_min = ~(unsigned)(klimit >> 32);
  const unsigned t8 = (unsigned)h_bin(__uint_as_float((ord_min & 0x80000000u) ? (ord_min ^ 0x80000000u) : ~ord_min)) >> 4;
  const uint8_t *row = imgs + (int64_t)img * img_stride + off + (int64_t)a * stride;
  const int nch = (cols + 64 * C - 1) / (64 * C);
  RwLds &W = s_w[w];
  uint8_t *s_run = W.flags;
  for (int i = lane; i < RW_FLAGS / 16; i += 64) reinterpret_cast<uint4 *>(s_run)[i] = uint4{0u, 0u, 0u, 0u};

  // ---- pass 1a ----
  unsigned neg[RW_CH], hit[RW_CH], excl[RW_CH], ordh[RW_CH][C];
  int slot[RW_CH];  // this thread's place in the candidate list (-1: not a candidate)
  unsigned carry_nn = 0, n_cand = 0;
  const unsigned short *recs = negmax + ((int64_t)img * rows + a) * NTR;
#pragma unroll
  for (int ch = 0; ch < RW_CH; ch++) {
    neg[ch] = 0u;
    hit[ch] = 0u;
    excl[ch] = 0u;
    slot[ch] = -1;
#pragma unroll
    for (int i = 0; i < C; i++) ordh[ch][i] = 0u;
    if (ch < nch) {  // (uniform)
      const int p0 = (ch * 64 + lane) * C;
      const unsigned rec = p0 < cols ? (unsigned)recs[ch * 64 + lane] : 0u;  // (threads past the row's end wrote nothing: walls, not neg)
      neg[ch] = rec & FULL;
      const unsigned nn = ~neg[ch] & FULL;
      const unsigned cnt = (unsigned)__popc(nn);
      const unsigned incl = wave_incl_add(cnt, lane);
      excl[ch] = carry_nn + incl - cnt;
      carry_nn += (unsigned)__builtin_amdgcn_readlane((int)incl, 63);
      const bool cand = p0 < cols && (rec >> 8) >= t8;
      const unsigned long long cm = __ballot(cand);
      if (cand) {
        slot[ch] = (int)(n_cand + (unsigned)__popcll(cm & ((1ull << lane) - 1ull)));
        W.list[slot[ch]] = (unsigned short)(ch * 64 + lane);
        W.cexcl[slot[ch]] = excl[ch];
      }
      n_cand += (unsigned)__popcll(cm);
    }
  }
  n_cand = (unsigned)__builtin_amdgcn_readfirstlane((int)n_cand);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the wavefront's own LDS stores, in order)

  // ---- pass 1b: the listed threads, 64 per pass ----
  unsigned long long any_flag = 0ull;
  for (unsigned base = 0; base < n_cand; base += 64) {  // (uniform)
    const unsigned idx = base + (unsigned)lane;
    if (idx < n_cand) {
      const int t = (int)W.list[idx];
      const int p0 = t * C;
      const unsigned ex = W.cexcl[idx];
      float h[C];
      unsigned ng;
      row_load_h_at(L, row, cols, mean, maxg, rcp_maxg, h, ng, p0, true);  // (edge rules under their own per-lane conditions)
      const unsigned nn = ~ng & FULL;
      const unsigned pix0 = (unsigned)a * (unsigned)cols + (unsigned)p0;
      unsigned hb = 0;
#pragma unroll
      for (int i = 0; i < C; i++) {
        unsigned o = ord_f32(h[i] + 0.0f);  // (-0.0) + 0.0 = +0.0
        if (p0 + i >= cols) o = 0u;         // walls
        W.res[lane][1 + i] = o;
        const unsigned long long key = ((unsigned long long)(~o) << 32) | (unsigned long long)(pix0 + (unsigned)i);
        if (key < klimit) {
          hb |= 1u << i;
          const unsigned c = ex + (unsigned)__popc(nn & ((1u << i) - 1u));  // non-neg pixels before p = the number of the run p is in / that ends at p
          s_run[c] = 1;                                // a neg pixel: its own run; a non-neg pixel: the run on its left ...
          if ((nn >> i) & 1u) s_run[c + 1] = 1;        // ... and the run on its right
        }
      }
      W.res[lane][0] = hb;
      any_flag |= __ballot(hb != 0u);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // back to the owners: thread (ch, lane) with a slot in this pass
#pragma unroll
    for (int ch = 0; ch < RW_CH; ch++) {
      const int sl = slot[ch] - (int)base;
      if (sl >= 0 && sl < 64) {
        hit[ch] = W.res[sl][0];
#pragma unroll
        for (int i = 0; i < C; i++) ordh[ch][i] = W.res[sl][1 + i];
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the next pass overwrites the slots)
  }
  any_flag = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(any_flag >> 32)) << 32) |
             (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)any_flag);

  // ---- pass 2a: marks ----
  unsigned marked[RW_CH];
#pragma unroll
  for (int ch = 0; ch < RW_CH; ch++) {
    marked[ch] = 0u;
    if (ch < nch) {
      const int p0 = (ch * 64 + lane) * C;
      if (any_flag != 0ull) {  // (uniform) the row has a hit: a neg pixel is marked iff its run's flag is set
        const unsigned nn = ~neg[ch] & FULL;
        unsigned flags = 0;
#pragma unroll
        for (int i = C - 1; i >= 0; i--) {
          const unsigned f = s_run[excl[ch] + (unsigned)__popc(nn & ((1u << i) - 1u))];
          asm("v_cmp_ne_u32_e32 vcc, 0, %1\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(flags) : "v"(f) : "vcc");
        }
        marked[ch] = (~neg[ch] & FULL & hit[ch]) | (neg[ch] & flags);
      } else {
        marked[ch] = 0u;  // (a row without a hit has no mark)
      }
      if (p0 < cols) markbits[((int64_t)img * rows + a) * NTR + ch * 64 + lane] = (MarkT<C>)marked[ch];
    }
  }

  // ---- pass 2b: closed runs.  Marks are as sparse as hits (a dozen threads of a row hold a live pixel), so the three scans do not
  // run chunk by chunk (five of a row's seven chunks hold a mark: 5 x 170 instructions) but ONCE per 64 LIVE THREADS: start and
  // close bits are formed in chunk order (the neighbours' bits by DPP, cheap), the threads with a live pixel are compacted --
  // their bits and ord(h) handed over through the wavefront's LDS, as in pass 1b -- and the scans of cen_runs run over the
  // list: positions only grow along it and every start a live pixel can belong to is in it, so "the latest start" and "the
  // largest key of the latest run" mean what they meant in row order ----
  const int rmin = min_range < 0 ? 0 : min_range;
  uint2 *ro = row_runs + ((int64_t)img * rows + a) * row_cap;
  unsigned lsc[RW_CH];  // live | start << 8 | close << 16 of the chunk's 8 pixels
  int lslot[RW_CH];
  unsigned n_live = 0, lprev_c = 0u;
#pragma unroll
  for (int ch = 0; ch < RW_CH; ch++) {
    lsc[ch] = 0u;
    lslot[ch] = -1;
    if (ch < nch) {
      const bool any = __ballot(marked[ch] != 0u) != 0ull;  // (uniform)
      if (!any) {
        lprev_c = 0u;
      } else {
        const int p0 = (ch * 64 + lane) * C;
        const unsigned ge = p0 >= rmin ? FULL : (p0 + C <= rmin ? 0u : (FULL & ~((1u << (rmin - p0)) - 1u)));  // bit i: p >= rmin
        const unsigned live = marked[ch] & ge;
        unsigned lprev = dpp_u32<0x138>((live >> (C - 1)) & 1u);  // previous thread's last pixel live
        unsigned mnext = dpp_u32<0x130>(marked[ch] & 1u);         // next thread's first pixel marked
        if (lane == 0) lprev = lprev_c;
        const unsigned next0 = ch + 1 < RW_CH ? (unsigned)__builtin_amdgcn_readlane((int)(marked[ch + 1 < RW_CH ? ch + 1 : ch] & 1u), 0) : 0u;
        if (lane == 63) mnext = next0;
        const unsigned start = live & ~(((live << 1) | lprev) & FULL);
        const unsigned in_row = p0 + C < cols ? FULL : (p0 + 1 >= cols ? 0u : ((1u << (cols - 1 - p0)) - 1u));  // bit i: p + 1 < cols
        const unsigned close = live & ~((marked[ch] >> 1) | (mnext << (C - 1))) & in_row;
        lsc[ch] = live | (start << 8) | (close << 16);
        const unsigned long long lm = __ballot(live != 0u);
        if (live != 0u) lslot[ch] = (int)(n_live + (unsigned)__popcll(lm & ((1ull << lane) - 1ull)));
        n_live += (unsigned)__popcll(lm);
        lprev_c = (unsigned)__builtin_amdgcn_readlane((int)((live >> (C - 1)) & 1u), 63);
      }
    }
  }
  n_live = (unsigned)__builtin_amdgcn_readfirstlane((int)n_live);
  // A run's arg-max is exact as long as the run holds a hit (a hit's h is known and beats every other pixel of its run; among
  // equals the hit has the lower bin).  Every run of MARKED pixels holds one -- but the part of it at r >= min_range may not
  // (a dark stretch flagged by a bright pixel below min_range): such a record is recognised by its key (ord(h) below the
  // limit's) and redone from the image bytes after the scans, all 64 lanes on that one run.  At most one per row.
  unsigned fix_pos = 0u, fix_span = 0u;
  bool fix = false;
  unsigned s_carry = 0u, pos_base = 0u;
  double k_carry = 0.0;
  for (unsigned base = 0; base < n_live; base += 64) {  // (uniform)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the slots of the pass before are read)
#pragma unroll
    for (int ch = 0; ch < RW_CH; ch++) {
      const int sl = lslot[ch] - (int)base;
      if (sl >= 0 && sl < 64) {
        W.res[sl][0] = lsc[ch];
        W.res[sl][1] = (unsigned)((ch * 64 + lane) * C);
#pragma unroll
        for (int i = 0; i < C; i++) W.res[sl][2 + i] = ordh[ch][i];
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const bool act = base + (unsigned)lane < n_live;
    unsigned e_lsc = 0u, e_p0 = 0u, e_o[C];
#pragma unroll
    for (int i = 0; i < C; i++) e_o[i] = 0u;
    if (act) {
      e_lsc = W.res[lane][0];
      e_p0 = W.res[lane][1];
#pragma unroll
      for (int i = 0; i < C; i++) e_o[i] = W.res[lane][2 + i];
    }
    const unsigned live = e_lsc & FULL, start = (e_lsc >> 8) & FULL, close = (e_lsc >> 16) & FULL;
    const int p0 = (int)e_p0;
    const unsigned sloc = start ? (unsigned)(p0 + (31 - __builtin_clz(start)) + 1) : 0u;
    const unsigned x = wave_incl_max_u32(sloc);
    unsigned sin = dpp_u32<0x138>(x);
    sin = s_carry > sin ? s_carry : sin;
    {
      const unsigned xt = (unsigned)__builtin_amdgcn_readlane((int)x, 63);
      s_carry = xt > s_carry ? xt : s_carry;
    }
    double kin[C];
    double run = 0.0;
    const unsigned sinx = sin | 0x10000u, pbx = (unsigned)p0 + 32u + 0x10000u, c0 = 0x3fffu - ((unsigned)p0 & 0x3fffu);
#pragma unroll
    for (int i = 0; i < C; i++) {
      const unsigned below = start & ((2u << i) - 1u);
      const unsigned spx = below ? pbx - (unsigned)__builtin_clz(below) : sinx;
      const unsigned hi = __builtin_amdgcn_alignbit(spx, e_o[i], 18);
      const unsigned lo = (e_o[i] << 14) | (c0 - (unsigned)i);
      const double kd = __hiloint2double((int)(((live >> i) & 1u) ? hi : 0u), (int)lo);
      run = kmax(run, kd);
      kin[i] = run;
    }
    const double ki = wave_incl_max<false>(run, lane);
    const double kx = kmax(dpp_f64<0x138>(ki), k_carry);
    k_carry = kmax(k_carry, readlane_f64(ki, 63));
    const unsigned cnt = (unsigned)__popc(close);
    const unsigned inc = wave_incl_add(cnt, lane);
    unsigned pos = pos_base + inc - cnt;
    pos_base += (unsigned)__builtin_amdgcn_readlane((int)inc, 63);
#pragma unroll
    for (int i = 0; i < C; i++) {
      if ((close >> i) & 1u) {
        const unsigned long long k = (unsigned long long)__double_as_longlong(kmax(kx, kin[i]));
        const unsigned first = (unsigned)((k >> 46) & 0x7fffu) - 1u, arg = 0x3fffu - (unsigned)(k & 0x3fffu);
        if ((unsigned)(k >> 14) < ord_min) {  // (no hit in [first, last]: see above)
          fix = true;
          fix_pos = pos;
          fix_span = first | ((unsigned)(p0 + i) << 16);
        }
        ro[pos++] = uint2{first | ((unsigned)(p0 + i) << 16), arg};  // first | last << 16, arg-max bin
      }
    }
  }
  if (lane == 0) row_nruns[(int64_t)img * rows + a] = pos_base;
  for (unsigned long long fm = __ballot(fix); fm != 0ull; fm &= fm - 1ull) {  // (uniform)
    const int src = __ffsll((long long)fm) - 1;
    const unsigned span = (unsigned)__builtin_amdgcn_readlane((int)fix_span, src), at = (unsigned)__builtin_amdgcn_readlane((int)fix_pos, src);
    const int first = (int)(span & 0xffffu), last = (int)(span >> 16);
    unsigned long long best = 0ull;  // ord(h) << 32 | ~bin: the largest h, among equals the lowest bin
    for (int q = first + lane; q <= last; q += 64) {
      const unsigned o = ord_f32(pixel_h(row, cols, q, mean, maxg, rcp_maxg) + 0.0f);
      const unsigned long long kq = ((unsigned long long)o << 32) | (unsigned long long)(~(unsigned)q);
      best = kq > best ? kq : best;
    }
    best = dev_wave_max_u64(best);
    if (lane == 0) ro[at].y = ~(unsigned)best;
  }
}

// the adjacency test of the method: a closed run yields a keypoint when the azimuth below or above (wrap-around) has a
// marked pixel inside the run's range span.  One block per (azimuth, image) over the runs cen_runs recorded and the mark
// bits of the two neighbours (C bits per chunk); ordered compaction of the survivors' arg-max bins.
// (Round 6: a WAVEFRONT per azimuth, four azimuths per workgroup -- a row has a few dozen runs; the workgroup per azimuth staged the
// neighbours' 2 x 420 mark bytes in LDS behind a barrier and compacted through two more: 29 us per 64 scans for 1.3 M runs.)
constexpr int ADJ_WAVES = 4;
template <int C, int NT>
__global__ __launch_bounds__(64 * ADJ_WAVES) void cen_adjacent(int rows, int cols, int row_cap, const uint2 *__restrict__ row_runs,
                                                               const unsigned *__restrict__ row_nruns, const MarkT<C> *__restrict__ markbits,
                                                               int *__restrict__ row_out, unsigned *__restrict__ row_n) {
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int a = (int)blockIdx.x * ADJ_WAVES + w, img = blockIdx.y;
  if (a >= rows) return;  // (wave-uniform; no barrier in this kernel)
  const MarkT<C> *below = markbits + ((int64_t)img * rows + (a - 1 + rows) % rows) * NT;
  const MarkT<C> *above = markbits + ((int64_t)img * rows + (a + 1) % rows) * NT;
  const unsigned n = (unsigned)__builtin_amdgcn_readfirstlane((int)row_nruns[(int64_t)img * rows + a]);
  const uint2 *runs = row_runs + ((int64_t)img * rows + a) * row_cap;
  int *ro = row_out + ((int64_t)img * rows + a) * row_cap;
  unsigned done = 0;
  for (unsigned base = 0; base < n; base += 64) {
    const unsigned j = base + (unsigned)lane;
    bool keep = false;
    int arg = 0;
    if (j < n) {
      const uint2 r = runs[j];
      const int first = (int)(r.x & 0xffffu), last = (int)(r.x >> 16);
      arg = (int)r.y;
      for (int wd = first / C; wd <= last / C && !keep; wd++) {
        const int lo = wd == first / C ? first % C : 0, hi = wd == last / C ? last % C : C - 1;
        const unsigned mask = ((1u << (hi + 1)) - 1u) & ~((1u << lo) - 1u);
        const unsigned nb = wd * C < cols ? (unsigned)below[wd] | (unsigned)above[wd] : 0u;  // (chunks past the row's end were not written)
        keep = (nb & mask) != 0;
      }
    }
    const unsigned long long bal = __ballot(keep);
    if (keep) ro[done + (unsigned)__popcll(bal & ((1ull << lane) - 1ull))] = arg;
    done += (unsigned)__popcll(bal);
  }
  if (lane == 0) row_n[(int64_t)img * rows + a] = done;
}

// one wavefront per (azimuth, image): row-major packing of the rows' keypoints, polar -> Cartesian
constexpr int PACK_WAVES = 4;  // azimuths per workgroup
__global__ __launch_bounds__(64 * PACK_WAVES) void cen_pack(Scal *scal, int rows, int row_cap, const int *__restrict__ row_out,
                                                            const unsigned *__restrict__ row_n, const float *__restrict__ az, int64_t az_stride,
                                                            float resolution, int max_targets, int *__restrict__ targets, float *__restrict__ xy,
                                                            int *__restrict__ counts) {
  const int a = (int)blockIdx.x * PACK_WAVES + (int)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), img = blockIdx.y, lane = threadIdx.x & 63;
  if (a >= rows) return;  // (wave-uniform; no barrier in this kernel)
  const unsigned *rn = row_n + (int64_t)img * rows;
  unsigned before = 0;
  for (int r = lane; r < a; r += 64) before += rn[r];
  for (int o = 32; o >= 1; o >>= 1) before += __shfl_xor(before, o);
  const unsigned n = rn[a];
  int *tg = targets + (int64_t)img * max_targets * 2;
  float *pxy = xy ? xy + (int64_t)img * max_targets * 2 : nullptr;
  const float *azi = az ? az + (int64_t)img * az_stride : nullptr;
  const int *ro = row_out + ((int64_t)img * rows + a) * row_cap;
  for (unsigned i = lane; i < n; i += 64) {
    const unsigned d = before + i;
    if (d >= (unsigned)max_targets) break;
    const int r = ro[i];
    tg[2 * d] = a;
    tg[2 * d + 1] = r;
    if (pxy && azi) {
      const float range = __fmul_rn(__fadd_rn((float)r, 0.5f), resolution);
      pxy[2 * d] = __fmul_rn(range, cosf(azi[a]));
      pxy[2 * d + 1] = __fmul_rn(range, sinf(azi[a]));
    }
  }
  if (a == rows - 1 && lane == 0) {
    scal[img].n_targets = before + n;
    if (counts) counts[img] = (int)(before + n);
  }
}

}  // namespace

struct rsx_cen2019 {
  int device = 0, rows = 0, cols = 0;
  std::mutex mu;
  hipStream_t stream = nullptr;
  rsx::DevBuf img, scal, hist, list, row_out, row_n, targets, xy, az, counts, opener, row_runs, row_nruns, markbits, wavemax, negmax;
  rsx::DevBuf one;          // single-scan entry: [count | targets | xy] in one piece, read back with one copy
  void *one_host = nullptr;  // its pinned mirror
  size_t one_host_bytes = 0;
};

using rsx::fail;

namespace {

template <int C, int NT>
void launch_chain(rsx_cen2019 *h, const uint8_t *d_imgs, int64_t img_stride, int nb, int32_t stride, int32_t off, const rsx_cen2019_params &p,
                  const float *d_az, int64_t az_stride, float resolution, int32_t max_targets, int *d_targets, float *d_xy, int *d_counts,
                  int row_cap, hipStream_t s) {
  const int rows = h->rows, cols = h->cols;
  Scal *sc = h->scal.as<Scal>();
  const dim3 grid((unsigned)rows, (unsigned)nb);
  const int rpb = (int64_t)rows * nb >= 8192 ? ST_ROWS : 1;
  if (rpb > 1)
    hipLaunchKernelGGL((cen_stats<C, NT, ST_ROWS>), dim3((unsigned)((rows + rpb - 1) / rpb), (unsigned)nb), dim3(NT), 0, s, d_imgs, img_stride, rows,
                       cols, stride, off, sc, h->hist.as<unsigned>());
  else
    hipLaunchKernelGGL((cen_stats<C, NT, 1>), dim3((unsigned)rows, (unsigned)nb), dim3(NT), 0, s, d_imgs, img_stride, rows, cols, stride, off, sc, h->hist.as<unsigned>());
  hipLaunchKernelGGL(cen_scalars, dim3((unsigned)((nb + 63) / 64)), dim3(64), 0, s, sc, nb, (int64_t)rows * cols);
  const int hrpb = rpb > 1 ? HIST_ROWS : 1;
  // (experiments build: RSX_CEN_FORMS=block runs the workgroup-per-azimuth forms of cen_hist AND cen_runs in a batch too -- the two go
  // together: cen_runs_wave reads the per-thread records only cen_hist_wave writes, cen_runs the per-wavefront maxima of cen_hist)
  static const bool hist_block_form = [] { const char *e = rsx::exp_env("RSX_CEN_FORMS"); return e && e[0] == 'b'; }();
  bool hist_done = false;
  if constexpr (C == 8 && NT == 64 * HW_CH) {
    if (!hist_block_form && rpb > 1) {  // a batch: a wavefront per azimuth
      hipLaunchKernelGGL(cen_hist_wave, dim3((unsigned)((rows + HW_WAVES - 1) / HW_WAVES), (unsigned)nb), dim3(64 * HW_WAVES), 0, s, d_imgs, img_stride,
                         rows, cols, stride, off, sc, h->hist.as<unsigned>(), h->opener.as<OpRec<C>>(), h->negmax.as<unsigned short>());
      hist_done = true;
    }
  }
  if (!hist_done)
    hipLaunchKernelGGL((cen_hist<C, NT>), dim3((unsigned)((rows + hrpb - 1) / hrpb), (unsigned)nb), dim3(NT), 0, s, d_imgs, img_stride, rows, cols,
                       stride, off, sc, h->hist.as<unsigned>(), h->opener.as<OpRec<C>>(), h->wavemax.as<unsigned>(), hrpb);
  hipLaunchKernelGGL(cen_pick, dim3((unsigned)nb), dim3(256), 0, s, sc, h->hist.as<unsigned>(), p.max_points);
  hipLaunchKernelGGL((cen_collect<C, NT>), dim3((unsigned)((rows + rpb - 1) / rpb), (unsigned)nb), dim3(NT), 0, s, d_imgs, img_stride, rows, cols, stride, off, sc, h->opener.as<OpRec<C>>(),
                     h->list.as<unsigned long long>(), (int64_t)rows * cols, rpb);
  hipLaunchKernelGGL(cen_resolve, dim3((unsigned)nb), dim3(1024), 0, s, sc, h->list.as<unsigned long long>(), (int64_t)rows * cols, rows, cols,
                     p.max_points);
  const int rrpb = rpb > 1 ? RUNS_ROWS : 1;
  const bool block_form = !hist_done;  // (the form cen_hist took)
  if constexpr (C == 8 && NT == 64 * RW_CH) {
    if (!block_form && rpb > 1) {  // a batch: a wavefront per azimuth (cen_runs_wave; a single scan's 400 wavefronts would walk their rows one chunk after the other: 17 us against 7)
      hipLaunchKernelGGL(cen_runs_wave, dim3((unsigned)((rows + RW_WAVES - 1) / RW_WAVES), (unsigned)nb), dim3(64 * RW_WAVES), 0, s, d_imgs, img_stride,
                         rows, cols, stride, off, sc, p.min_range, row_cap, h->row_runs.as<uint2>(), h->row_nruns.as<unsigned>(),
                         h->markbits.as<MarkT<C>>(), h->negmax.as<unsigned short>());
    }
  }
  if (!(C == 8 && NT == 64 * RW_CH) || block_form || rpb <= 1)
    hipLaunchKernelGGL((cen_runs<C, NT>), dim3((unsigned)((rows + rrpb - 1) / rrpb), (unsigned)nb), dim3(NT), 0, s, d_imgs, img_stride, rows, cols,
                       stride, off, sc, p.min_range, row_cap, h->row_runs.as<uint2>(), h->row_nruns.as<unsigned>(),
                       h->markbits.as<MarkT<C>>(), h->wavemax.as<unsigned>(), rrpb);
  hipLaunchKernelGGL((cen_adjacent<C, NT>), dim3((unsigned)((rows + ADJ_WAVES - 1) / ADJ_WAVES), (unsigned)nb), dim3(64 * ADJ_WAVES), 0, s, rows, cols, row_cap, h->row_runs.as<uint2>(), h->row_nruns.as<unsigned>(),
                     h->markbits.as<MarkT<C>>(), h->row_out.as<int>(), h->row_n.as<unsigned>());
  hipLaunchKernelGGL(cen_pack, dim3((unsigned)((rows + PACK_WAVES - 1) / PACK_WAVES), (unsigned)nb), dim3(64 * PACK_WAVES), 0, s, sc, rows, row_cap, h->row_out.as<int>(), h->row_n.as<unsigned>(), d_az, az_stride,
                     resolution, max_targets, d_targets, d_xy, d_counts);
}

// d_imgs: nb device images img_stride bytes apart; results stay on the device: d_targets [nb][max_targets][2] int32,
// d_xy [nb][max_targets][2] float (optional, needs d_az), d_counts [nb] (optional; the counts are also in Scal)
int extract_device(rsx_cen2019 *h, const uint8_t *d_imgs, int64_t img_stride, int nb, int32_t stride, int32_t off, const rsx_cen2019_params &p,
                   const float *d_az, int64_t az_stride, float resolution, int32_t max_targets, int *d_targets, float *d_xy, int *d_counts,
                   hipStream_t s) {
  const int rows = h->rows, cols = h->cols;
  const int row_cap = cols / 2 + 1;  // a row of `cols` bins holds at most ceil(cols / 2) closed runs
  for (int b0 = 0; b0 < nb; b0 += MAX_SUB_BATCH) {
    const int n = nb - b0 < MAX_SUB_BATCH ? nb - b0 : MAX_SUB_BATCH;
    RSX_TRY(h->scal.reserve((size_t)n * sizeof(Scal), s, false));
    RSX_TRY(h->hist.reserve((size_t)n * NBIN * 4, s, false));
    RSX_TRY(h->list.reserve((size_t)n * rows * cols * 8, s, false));
    RSX_TRY(h->row_out.reserve((size_t)n * rows * row_cap * 4, s, false));
    RSX_TRY(h->row_n.reserve((size_t)n * rows * 4, s, false));
    {
      const size_t nt = cols <= 8 * 512 ? 512 : 1024;  // threads per row block: an OpRec and a MarkT per thread
      RSX_TRY(h->opener.reserve((size_t)n * rows * nt * (cols <= 8 * 512 ? 2 : 4), s, false));
      RSX_TRY(h->markbits.reserve((size_t)n * rows * nt * (cols <= 8 * 512 ? 1 : 2), s, false));
      RSX_TRY(h->row_runs.reserve((size_t)n * rows * row_cap * 8, s, false));
      RSX_TRY(h->row_nruns.reserve((size_t)n * rows * 4, s, false));
      RSX_TRY(h->wavemax.reserve((size_t)n * rows * (nt / 64) * 4, s, false));  // largest ord(h) per wavefront and row (cen_hist -> cen_runs)
      RSX_TRY(h->negmax.reserve((size_t)n * rows * nt * 2, s, false));           // sign bits | bin / 16 of the largest h per thread (cen_hist_wave -> cen_runs_wave)
    }
    RSX_HIP(hipMemsetAsync(h->scal.p, 0, (size_t)n * sizeof(Scal), s));
    const uint8_t *im = d_imgs + (int64_t)b0 * img_stride;
    int *tg = d_targets + (int64_t)b0 * max_targets * 2;
    float *pxy = d_xy ? d_xy + (int64_t)b0 * max_targets * 2 : nullptr;
    const float *azp = d_az ? d_az + (int64_t)b0 * az_stride : nullptr;
    int *cn = d_counts ? d_counts + b0 : nullptr;
    // <= 4096 bins: 512 threads x 8 bins (about 100 VGPRs: four waves per SIMD; 256 x 16 needs 176: two); wider rows: 1024 x 16
    // (a Navtech CIR row has 3360 bins = 420 threads.  448 threads -- no eighth wave that executes the row passes with all
    // lanes off -- measured SLOWER: 54.2 k against 57.4 k scans/s batched, seven waves do not spread evenly over four SIMDs;
    // 256 threads x 16 bins: 43 k.  tools/ab_cen.py, experiments build.  The timing exits that took cen_hist apart for DESIGN.md
    // -- `if (dbg & 1) return;` in front of its barriers -- made the 1024-thread instantiation sum garbage even with dbg = 0
    // (tests/test_gpu_cen2019.py::test_wide_rows caught it) and are gone again.)
    static const int cfg = [] { const char *e = rsx::exp_env("RSX_CEN_CFG"); return e ? atoi(e) : 0; }();
    if (cfg == 1 && cols <= 16 * 256)
      launch_chain<16, 256>(h, im, img_stride, n, stride, off, p, azp, az_stride, resolution, max_targets, tg, pxy, cn, row_cap, s);
    else if (cfg == 2 && cols <= 8 * 448)
      launch_chain<8, 448>(h, im, img_stride, n, stride, off, p, azp, az_stride, resolution, max_targets, tg, pxy, cn, row_cap, s);
    else if (cols <= 8 * 512)
      launch_chain<8, 512>(h, im, img_stride, n, stride, off, p, azp, az_stride, resolution, max_targets, tg, pxy, cn, row_cap, s);
    else
      launch_chain<16, 1024>(h, im, img_stride, n, stride, off, p, azp, az_stride, resolution, max_targets, tg, pxy, cn, row_cap, s);
    RSX_HIP(hipGetLastError());
  }
  return RSX_OK;
}

}  // namespace

extern "C" {

int rsx_cen2019_default_params(rsx_cen2019_params *p) try {
  if (!p) return fail(RSX_ERR_BAD_ARG, "null params");
  p->max_points = 10000;  // yeti_radar_odometry default for cen2019 (recollection)
  p->min_range = 58;
  return RSX_OK;
} RSX_CATCH_ALL

int rsx_cen2019_create(int device, int32_t rows, int32_t cols, rsx_cen2019 **out) try {
  if (!out) return fail(RSX_ERR_BAD_ARG, "null out");
  *out = nullptr;
  if (rows < 1 || rows > 1024 || cols < 2 || cols > 16384) return fail(RSX_ERR_BAD_ARG, "image shape %d x %d unsupported", rows, cols);
  int ndev = rsx_device_count();
  if (ndev <= 0) return fail(RSX_ERR_NO_DEVICE, "no HIP device visible (librsx has no CPU fallback)");
  if (device < 0 || device >= ndev) return fail(RSX_ERR_NO_DEVICE, "device %d out of range (%d visible)", device, ndev);
  rsx_cen2019 *h = new (std::nothrow) rsx_cen2019();
  if (!h) return fail(RSX_ERR_OOM, "host alloc");
  h->device = device;
  h->rows = rows;
  h->cols = cols;
  hipError_t e = hipSetDevice(device);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    delete h;
    return fail(RSX_ERR_HIP, "create: %s", hipGetErrorString(e));
  }
  *out = h;
  return RSX_OK;
} RSX_CATCH_ALL

int rsx_cen2019_destroy(rsx_cen2019 *h) try {
  if (!h) return RSX_OK;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  if (h->one_host) (void)hipHostFree(h->one_host);
  h->one.release();
  for (rsx::DevBuf *b : {&h->img, &h->scal, &h->hist, &h->list, &h->row_out, &h->row_n, &h->targets, &h->xy, &h->az, &h->counts, &h->opener, &h->row_runs, &h->row_nruns, &h->markbits, &h->wavemax, &h->negmax}) b->release();
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return RSX_OK;
} RSX_CATCH_ALL

int rsx_cen2019_extract_batch_device(rsx_cen2019 *h, const uint8_t *d_imgs, int32_t n_images, int64_t image_stride_bytes, int32_t row_stride,
                                     int32_t col_offset, const rsx_cen2019_params *params, const float *d_azimuths,
                                     int32_t azimuths_per_image, float resolution, int32_t *d_targets, float *d_xy, int32_t max_targets,
                                     int32_t *d_counts, void *stream) try {
  if (!h || !d_imgs || !d_targets || n_images < 0 || max_targets < 1) return fail(RSX_ERR_BAD_ARG, "bad arg");
  if (col_offset < 0 || row_stride < col_offset + h->cols) return fail(RSX_ERR_BAD_ARG, "row_stride %d too small for offset %d + %d columns", row_stride, col_offset, h->cols);
  if (image_stride_bytes < (int64_t)h->rows * row_stride && n_images > 1) return fail(RSX_ERR_BAD_ARG, "image_stride_bytes smaller than an image");
  if (d_xy && !d_azimuths) return fail(RSX_ERR_BAD_ARG, "d_xy needs d_azimuths");
  if (n_images == 0) return RSX_OK;
  rsx_cen2019_params p;
  rsx_cen2019_default_params(&p);
  if (params) p = *params;
  std::lock_guard<std::mutex> lk(h->mu);
  RSX_HIP(hipSetDevice(h->device));
  hipStream_t s = stream ? static_cast<hipStream_t>(stream) : h->stream;
  return extract_device(h, d_imgs, image_stride_bytes, n_images, row_stride, col_offset, p, d_azimuths, azimuths_per_image ? h->rows : 0,
                        resolution, max_targets, d_targets, d_xy, d_counts, s);
} RSX_CATCH_ALL

int rsx_cen2019_extract_batch(rsx_cen2019 *h, const uint8_t *imgs, int32_t n_images, int64_t image_stride_bytes, int32_t row_stride,
                              int32_t col_offset, const rsx_cen2019_params *params, const float *azimuths, int32_t azimuths_per_image,
                              float resolution, int32_t *out_targets, float *out_xy, int32_t max_targets, int32_t *out_counts) try {
  if (!h || !imgs || !out_targets || !out_counts || n_images < 0 || max_targets < 0) return fail(RSX_ERR_BAD_ARG, "bad arg");
  if (col_offset < 0 || row_stride < col_offset + h->cols) return fail(RSX_ERR_BAD_ARG, "row_stride %d too small for offset %d + %d columns", row_stride, col_offset, h->cols);
  if (n_images > 1 && image_stride_bytes < (int64_t)h->rows * row_stride) return fail(RSX_ERR_BAD_ARG, "image_stride_bytes smaller than an image");
  if (out_xy && !azimuths) return fail(RSX_ERR_BAD_ARG, "out_xy needs azimuths");
  if (n_images == 0) return RSX_OK;
  rsx_cen2019_params p;
  rsx_cen2019_default_params(&p);
  if (params) p = *params;
  std::lock_guard<std::mutex> lk(h->mu);
  RSX_HIP(hipSetDevice(h->device));
  hipStream_t s = h->stream;
  const size_t ibytes = (size_t)h->rows * row_stride;
  const int mt = max_targets > 0 ? max_targets : 1;
  if (n_images == 1) {
    // the live single-scan entry: the image goes up in one copy (asynchronous when the caller's buffer is pinned,
    // rsx_host_alloc_pinned), the count and every keypoint slot come back in ONE copy into pinned memory, one synchronise
    // (before round 4: the count first, a synchronise, then the keypoints, a second synchronise)
    const size_t kp = (size_t)mt * 8, total = 256 + 2 * kp;
    RSX_TRY(h->img.reserve(ibytes, s, false));
    RSX_TRY(h->one.reserve(total, s, false));
    if (total > h->one_host_bytes) {
      if (h->one_host) (void)hipHostFree(h->one_host);
      h->one_host = nullptr;
      h->one_host_bytes = 0;
      RSX_HIP(hipHostMalloc(&h->one_host, total, hipHostMallocDefault));
      h->one_host_bytes = total;
    }
    RSX_HIP(hipMemcpyAsync(h->img.p, imgs, ibytes, hipMemcpyHostToDevice, s));
    const float *d_az = nullptr;
    if (azimuths) {
      RSX_TRY(h->az.reserve((size_t)h->rows * 4, s, false));
      RSX_HIP(hipMemcpyAsync(h->az.p, azimuths, (size_t)h->rows * 4, hipMemcpyHostToDevice, s));
      d_az = h->az.as<float>();
    }
    char *d_one = h->one.as<char>();
    RSX_TRY(extract_device(h, h->img.as<uint8_t>(), (int64_t)ibytes, 1, row_stride, col_offset, p, d_az, 0, resolution, mt,
                           reinterpret_cast<int *>(d_one + 256), d_az ? reinterpret_cast<float *>(d_one + 256 + kp) : nullptr,
                           reinterpret_cast<int *>(d_one), s));
    // the count and the first 16 384 keypoint slots (a scan yields ~3 000) in one copy; a longer list takes a second one
    const size_t first = mt < 16384 ? (size_t)mt : 16384, fb = first * 8;
    RSX_HIP(hipMemcpyAsync(h->one_host, d_one, 256 + fb, hipMemcpyDeviceToHost, s));
    if (out_xy) RSX_HIP(hipMemcpyAsync(static_cast<char *>(h->one_host) + 256 + kp, d_one + 256 + kp, fb, hipMemcpyDeviceToHost, s));
    RSX_HIP(hipStreamSynchronize(s));
    const char *hp = static_cast<const char *>(h->one_host);
    const unsigned cnt = *reinterpret_cast<const unsigned *>(hp);
    out_counts[0] = (int32_t)cnt;
    const unsigned w = cnt < (unsigned)max_targets ? cnt : (unsigned)max_targets;
    if (w > first) {
      RSX_HIP(hipMemcpyAsync(h->one_host, d_one, out_xy ? total : 256 + kp, hipMemcpyDeviceToHost, s));
      RSX_HIP(hipStreamSynchronize(s));
    }
    if (w) {
      std::memcpy(out_targets, hp + 256, (size_t)w * 8);
      if (out_xy) std::memcpy(out_xy, hp + 256 + kp, (size_t)w * 8);
    }
    return RSX_OK;
  }
  // sub-batches bound the staging memory; each one is a single upload, one launch chain, one download
  for (int b0 = 0; b0 < n_images; b0 += MAX_SUB_BATCH) {
    const int n = n_images - b0 < MAX_SUB_BATCH ? n_images - b0 : MAX_SUB_BATCH;
    RSX_TRY(h->img.reserve(ibytes * n, s, false));
    RSX_TRY(h->targets.reserve((size_t)n * mt * 8, s, false));
    RSX_TRY(h->xy.reserve((size_t)n * mt * 8, s, false));
    RSX_TRY(h->counts.reserve((size_t)n * 4, s, false));
    if (n == 1 || image_stride_bytes == (int64_t)ibytes) {
      RSX_HIP(hipMemcpyAsync(h->img.p, imgs + (int64_t)b0 * image_stride_bytes, ibytes * n, hipMemcpyHostToDevice, s));
    } else {
      RSX_HIP(hipMemcpy2DAsync(h->img.p, ibytes, imgs + (int64_t)b0 * image_stride_bytes, (size_t)image_stride_bytes, ibytes, (size_t)n,
                               hipMemcpyHostToDevice, s));
    }
    const float *d_az = nullptr;
    if (azimuths) {
      const size_t na = (size_t)h->rows * (azimuths_per_image ? n : 1);
      RSX_TRY(h->az.reserve(na * 4, s, false));
      RSX_HIP(hipMemcpyAsync(h->az.p, azimuths + (azimuths_per_image ? (size_t)b0 * h->rows : 0), na * 4, hipMemcpyHostToDevice, s));
      d_az = h->az.as<float>();
    }
    RSX_TRY(extract_device(h, h->img.as<uint8_t>(), (int64_t)ibytes, n, row_stride, col_offset, p, d_az, azimuths_per_image ? h->rows : 0, resolution,
                           mt, h->targets.as<int>(), d_az ? h->xy.as<float>() : nullptr, h->counts.as<int>(), s));
    RSX_HIP(hipMemcpyAsync(out_counts + b0, h->counts.p, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    RSX_HIP(hipStreamSynchronize(s));
    for (int i = 0; i < n; i++) {
      const unsigned cnt = (unsigned)out_counts[b0 + i];
      const unsigned w = cnt < (unsigned)max_targets ? cnt : (unsigned)max_targets;
      if (!w) continue;
      RSX_HIP(hipMemcpyAsync(out_targets + (int64_t)(b0 + i) * max_targets * 2, h->targets.as<int>() + (int64_t)i * mt * 2, (size_t)w * 8,
                             hipMemcpyDeviceToHost, s));
      if (out_xy)
        RSX_HIP(hipMemcpyAsync(out_xy + (int64_t)(b0 + i) * max_targets * 2, h->xy.as<float>() + (int64_t)i * mt * 2, (size_t)w * 8,
                               hipMemcpyDeviceToHost, s));
    }
    RSX_HIP(hipStreamSynchronize(s));
  }
  return RSX_OK;
} RSX_CATCH_ALL

int rsx_cen2019_extract(rsx_cen2019 *h, const uint8_t *img, int32_t row_stride, int32_t col_offset, const rsx_cen2019_params *params,
                        const float *azimuths, float resolution, int32_t *out_targets, float *out_xy, int32_t max_targets,
                        int32_t *out_count) try {
  if (!h || !img || !out_targets || !out_count || max_targets < 0) return fail(RSX_ERR_BAD_ARG, "bad arg");
  return rsx_cen2019_extract_batch(h, img, 1, (int64_t)h->rows * row_stride, row_stride, col_offset, params, azimuths, 0, resolution, out_targets,
                                   out_xy, max_targets, out_count);
} RSX_CATCH_ALL

#ifdef RSX_EXPERIMENTS
// experiments builds only (tools/): the per-image scalars of the LAST extraction (sum of bytes, largest gradient, ...)
int rsx_cen2019_debug_scal(rsx_cen2019 *h, void *out64) try {
  if (!h || !out64 || !h->scal.p) return RSX_ERR_BAD_ARG;
  (void)hipSetDevice(h->device);
  (void)hipStreamSynchronize(h->stream);
  return hipMemcpy(out64, h->scal.p, 64, hipMemcpyDeviceToHost) == hipSuccess ? RSX_OK : RSX_ERR_HIP;
} RSX_CATCH_ALL
#endif

}  // extern "C"
