// sc_spec.hip -- spectral form of the ScanContext lower-bound filter (gfx950 / CDNA4).
//
// Same contract as sc_filter.hip: for every (query, entry) pair a float L~ with
//        L~ - filter_eps()  <=  L := min over ALL 60 shifts k of d_k  <=  dist(query, entry)
// (d_k = column-cosine distance at shift k, SC.cpp:69-90; dist = SC.cpp:116-148), so the exact fp64
// kernel only has to re-score the entries whose bound can still reach the top-k and the results stay
// bit-identical to the oracle.  What changes is how the 60 correlation values
//        S_k = sum_j cos(query column (j+k)%60, entry column j)
// are computed: the direct filter evaluates the circular cross-correlation as a K = 1200 GEMM per shift
// (72 000 MAC per pair); here the cyclic group of the 60 sectors is factored by the CRT,
//        Z60 = Z4 x Z15,   sector c  <->  (a, b) = (c mod 4, c mod 15),   shift k <-> (k4, k15),
// and the correlation is diagonalised along the Z15 axis only:
//        X_f[a][r] = sum_b x^[r][c(a,b)] e^(-2 pi i f b / 15)                  (f = 0..7; 8..14 are conjugates)
//   stage 1   C_f[k4]   = sum_{a,r} Q_f[(a+k4) mod 4][r] * conj(E_f[a][r])     (K = 80 complex MAC, direct over Z4)
//   stage 2   S[k4][k15] = 1/15 ( C_0 + 2 sum_{f=1..7} Re( C_f e^(2 pi i f k15 / 15) ) )
// 9 280 + 960 MAC per pair, both stages on the matrix cores (v_mfma_f32_32x32x16_f16):
//   stage 1   one MFMA tile = 4 queries x 32 entries for one f.  Rows = (query, re/im variant, k4):
//             the re rows read the stream [Qr | Qi], the im rows [Qi | -Qr], both against the entry
//             column [Er | Ei]; row k4 reads the stream 40 elements (one `a` block) further on, which is
//             the direct Z4 correlation -- the circulant trick of sc_filter.hip, 4 rows instead of 60.
//             With this row order accumulator register j holds (query j/4, k4 = j%4) with Re C in
//             lanes 0..31 and Im C in lanes 32..63 of the SAME register,
//   stage 2   which is exactly the B-fragment layout of a K = 16 MFMA (lanes 0..31: k = 0..7, lanes
//             32..63: k = 8..15): the 8 accumulators C_0..C_7 of one (query, k4) are packed to fp16 and
//             multiplied by the constant 15 x 16 inverse-DFT matrix (A operand) -- no cross-lane traffic.
//             C_0 is split into an fp16 hi + lo pair (the Im slot of f = 0 is free), and the weights are
//             scaled by 15/16 so that the weight of C_0 (1/16) is exact in fp16.
// n_eff(k) (the number of columns that are non-empty in both images at shift k, SC.cpp:78-88) is a
// circular cross-correlation too, of the two 60-bit column masks: exact small integers, computed by the
// matrix cores as well (v_mfma_f32_32x32x64_f8f6f4 on fp8 0/1 bytes, K = 64 in one instruction: A = circulant
// of the query mask, rows ordered like the shifts of the stage-2 output; B = the entries' mask bytes; 2 MFMAs
// per query, fp32 result).  The division
// S_k / n_eff(k) is replaced by a multiplication with a quadratic upper bound of 1/n on
// [n_lo, n_hi] = [n_q + n_e - 60, min(n_q, n_e)] (chord minus c (n-n_lo)(n_hi-n), c = 1/(n_lo n_hi^2): exact
// at both ends, relative excess < (n_hi-n_lo)^3 / (4 n_lo n_hi^2), ~1e-4 for the usual few empty sectors),
// evaluated two shifts at a time with packed fp32 arithmetic.
//
// Error bound (u = 2^-11, everything in units of S; complex magnitudes throughout; ||Q|| ||E|| = 15 sqrt(n_q n_e)
// by Parseval (sum_f |X_f|^2 = 15 sum_b |x_b|^2) and unit columns):
//   * spectra rounded to fp16 (computed in fp64; re and im each off by <= u relative, so |dQ| <= u |Q|):
//     |dC_f| <= (2u+u^2) T_f,  T_f = sum_{a,r} |Q_f||E_f|;  S_k = 1/15 sum_{f=0..14} C_f w^(fk), |w| = 1, so
//     |dS| <= (2u+u^2) 1/15 sum_f T_f <= (2u+u^2) sqrt(n_q n_e)  (Cauchy-Schwarz over (f,a,r))     = 9.77e-4 sqrt(n_q n_e)
//   * stage 2: Re C_f, Im C_f (f >= 1) and the weights rounded to fp16:
//     <= (2u+u^2) 2/16 (|Re C_f||cos| + |Im C_f||sin|) <= (2u+u^2) 2/16 |C_f| per f; |C_f| <= T_f; times 16/15:
//     <= (2u+u^2) sqrt(n_q n_e)                                                                     = 9.77e-4 sqrt(n_q n_e)
//   * fp32 accumulation (K = 160: 160 * 2^-23 relative to sum |terms|, and K = 16), C_0 hi/lo split (2^-21),
//     fp16 subnormals (2^-25 absolute per element), epilogue arithmetic:                             < 4e-5 sqrt(n_q n_e)
//   The stage-2 term only involves the frequencies f != 0, so its Cauchy-Schwarz bound uses the energy outside
//   f = 0:  a := n - 1/15 sum_{a,r} |X_0[a][r]|^2  (per image, from the exact spectra)  ->  (2u+u^2) sqrt(a_q a_e).
//   kE1 = 1.02e-3 (spectra rounding + the small terms), kE2 = 1.0e-3.  The kernel returns
//        L~ = 1 - max_k S_k u(n_eff(k)) - (kE1 sqrt(n_q n_e) + kE2 sqrt(a_q a_e)) / n_lo + filter_eps()
//   so that the direct filter's contract (L~ - filter_eps() <= L) holds unchanged downstream.
//
// Mapping: one wave per 32 entries (76 B fragments = 304 registers of entry spectra: 64 fragments in the AGPRs,
// 4 in VGPRs, the first 8 parked in LDS), 4 waves = 128 entries per block, one wave per SIMD; queries
// streamed through LDS in tiles of 4 (three tile buffers, global_load_lds two tiles ahead, counted vmcnt +
// raw s_barrier).  Per (4 queries x 32 entries): 76 stage-1 MFMAs (one ds_read_b128 A fragment each, hand-
// issued ring) + 16 stage-2 MFMAs + 8 (double-length fp8) mask MFMAs, against 600 MFMAs for the same pairs
// in the direct filter.  DESIGN.md 4.1b has the cycle budget of a tile and what limits it.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "rsx_common.h"
#include "sc_kernels.h"
#include "sc_entry_dev.h"
#include "sc_plan.h"

namespace rsx {
namespace sc {

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float float2v __attribute__((ext_vector_type(2)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef unsigned long long u64;

#define AS1 __attribute__((address_space(1)))
#define AS3 __attribute__((address_space(3)))

constexpr int SP_DC_STEPS = 6;                      // f = 0: K = 4 blocks x 24 (20 rings + 4 zeros)
constexpr int SP_F_STEPS = 10;                      // f >= 1: K = 4 blocks x (20 re + 20 im)
constexpr int SP_FRAGS = SP_DC_STEPS + 7 * SP_F_STEPS;  // 76 B fragments per 32-entry tile
constexpr int SP_KV = 16 * SP_FRAGS;                // 1216 fp16 per entry
static_assert(SP_KV * 2 == SPEC_DB_BYTES_PER_ENTRY, "layout");

// query image (bytes).  A stream holds the 4 `a` blocks ONCE; row k4 starts k4 blocks in and wraps around at the end of
// the stream (round 1 stored 7 blocks so that no row wrapped: 1.75x the bytes through the LDS-DMA, whose ~10 issue
// slots per wave and tile cost 1.5 k of the tile's 7.5 k cycles).  Every 16-byte chunk lies inside one block pair, so a
// wrapped read is the same ds_read_b128 from (address - stream length): one v_cndmask per stage-1 slot.
// The strides keep every A-fragment read bank-conflict free: a ds_read_b128 is served 16 lanes at a time and those
// 16 lanes (2 queries x 2 variants x 4 k4) must hit 16 different 16-byte slots mod 256: k4 * 80 B = 5 slots (k4 * 48 B
// = 3 slots for f = 0) gives each k4 its own class mod 4 -- a wrap moves a lane by -20 (+4 for f = 0) slots = the same
// class -- and the variant stride (20 slots = 4 mod 16) and the query stride (296 slots = 8 mod 16) fill the class
constexpr int SP_DC_BYTES = 192;                    // 4 blocks x 48 B (20 rings + 4 zeros)
constexpr int SP_VS = 320;                          // one variant stream: 4 blocks x 80 B
constexpr int SP_F_BYTES = 2 * SP_VS;               // 640
constexpr int SP_TAIL = SP_DC_BYTES + 7 * SP_F_BYTES;  // 4672: {int n_q, int flags, float sqrt(n_q), float sqrt(a_q)}
constexpr int SP_QS = SPEC_QIMG_BYTES;              // 4736 (296 slots = 8 mod 16)
static_assert(SP_QS >= SP_TAIL + 16 && SP_QS % 16 == 0, "layout");
static_assert((SP_VS / 16) % 16 == 4 && (SP_QS / 16) % 16 == 8, "bank-conflict-free strides");
// column-mask bytes for the n_eff MFMAs, a second image per query (only tiles with a query that has an empty column
// fetch it): M2[i] = mask bit (i mod 60); row k of the circulant reads 64 consecutive bytes from M2[k]; 16 copies
// displaced by one byte each make that read 16-byte aligned (copy j = M2[j ..], row k uses copy k mod 16 at offset
// k - k mod 16) and, 112 B = 7 slots apart, bank-conflict free for the row order of the MFMA
constexpr int SP_MASK_COPY = 112;
constexpr int SP_MASK_BYTES = 16 * SP_MASK_COPY;    // 1792 per query
constexpr int SP_QPT = 4;                           // queries per MFMA tile
constexpr int SP_TPP = 1;                           // tiles per LDS phase
constexpr int SP_QPP = SP_QPT * SP_TPP;
// one LDS tile buffer: the stream images of 4 queries in whole 1 KiB DMA pieces, then their mask images
constexpr int SP_STREAM_PIECES = (SP_QPP * SP_QS + 1023) / 1024;   // 19
constexpr int SP_MASKREG_OFF = SP_STREAM_PIECES * 1024;            // 19456
constexpr int SP_MASK_PIECES = SP_QPP * SP_MASK_BYTES / 1024;      // 7
static_assert(SP_QPP * SP_MASK_BYTES % 1024 == 0, "layout");
constexpr int SP_PHASE_BYTES = SP_MASKREG_OFF + SP_QPP * SP_MASK_BYTES;  // 26624 = 26 KiB
// global layout of a query batch: [nq4 stream images][nq4 mask images][nq4 flag bytes (1: the query has an empty column)]
__host__ __device__ constexpr int64_t sp_nq4(int32_t nq) { return ((int64_t)nq + 3) / 4 * 4; }
__host__ __device__ constexpr int64_t sp_masks_at(int32_t nq) { return sp_nq4(nq) * SP_QS; }
__host__ __device__ constexpr int64_t sp_flags_at(int32_t nq) { return sp_nq4(nq) * (SP_QS + SP_MASK_BYTES); }
#ifndef SP_OPT_NBUF
#define SP_OPT_NBUF 3
#endif
#ifndef SP_OPT_LOCKSTEP
#define SP_OPT_LOCKSTEP 0   // 1: one unit per workgroup, query ranges outermost (no gain measured: the query stream is not L2-bound)
#endif
constexpr int SP_NBUF = SP_OPT_NBUF;                // LDS tile buffers: the DMA runs SP_NBUF - 1 tiles ahead
constexpr int SP_DMA_PER_Q = 2;  // DMA pieces a wave issues in the tail of one query
static_assert((SP_STREAM_PIECES + SP_MASK_PIECES + 3) / 4 <= 11, "wait_vmcnt_le covers <= 11 DMA instructions per wave and tile");
static_assert((SP_STREAM_PIECES + SP_MASK_PIECES + 3) / 4 <= SP_DMA_PER_Q * SP_QPT, "the tail issues SP_DMA_PER_Q DMA pieces per query");
#ifndef SP_OPT_BV
#define SP_OPT_BV 12   // VGPR-resident B fragments are SP_B_LDS .. SP_OPT_BV-1
#endif
constexpr int SP_B_VGPR = SP_OPT_BV;  // B fragments kept in VGPRs; the rest live in AGPRs
// 76 B fragments = 304 registers: 64 fragments fill the 256 AGPRs, 4 sit in VGPRs, and the first SP_B_LDS
// (the 6 K-steps of f = 0 and the first 2 of f = 1) are parked in LDS (8 KiB per wave) and fetched with the A fragments of each tile
// -- left to the register allocator they were spilled to scratch and reloaded every tile, and every scratch
// reload drains the vmcnt queue (DMA and bound stores in flight)
#ifndef SP_OPT_BLDS
#define SP_OPT_BLDS 8
#endif
constexpr int SP_B_LDS = SP_OPT_BLDS;
constexpr int SP_BPARK_OFF = 0;                              // before the tile buffers: a wrapped read address (row address
                                                             // minus the stream length) then never runs below LDS address 0
constexpr int SP_TILES_OFF = 4 * SP_B_LDS * 1024;
constexpr int SP_LDS_BYTES = SP_TILES_OFF + SP_NBUF * SP_PHASE_BYTES;
static_assert(SP_TILES_OFF >= SP_VS, "wrapped addresses stay non-negative");
static_assert(SP_LDS_BYTES <= 160 * 1024, "LDS budget");

// -DRSX_SPEC_INSTRUMENT=1 compiles in the timing experiments (RSX_SPEC_DBG: run without the DMA / the stores) and
// the per-region s_memtime profile (RSX_SPEC_PROF); they cost registers, so normal builds leave them out
#ifndef RSX_SPEC_INSTRUMENT
#define RSX_SPEC_INSTRUMENT 0
#endif
constexpr bool kInstr = RSX_SPEC_INSTRUMENT != 0;

constexpr float kE1 = 1.02e-3f, kE2 = 1.0e-3f;
constexpr u64 kNonFinite = 1ull << 63;
constexpr u64 kMask60 = (1ull << 60) - 1ull;

__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

template <typename F, int... Is>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {
  static_for_impl(static_cast<F &&>(f), std::make_integer_sequence<int, N>{});
}

// ------------------------------------------------------------------------------------------
// spectra of one descriptor (one wave): xn = column-normalised image (fp64, in LDS), kv = the 1216
// fp16 K-vector  [ f = 0: 4 x (20 values + 4 zeros) | f = 1..7: 4 x (20 re + 20 im) ]
// returns the column mask (bit j = column j non-zero, bit 63 = non-finite element)
// ------------------------------------------------------------------------------------------
// QIMG: kv is the stream image of a query instead (f = 0: the same 96 halves; f >= 1: 320 halves, the re stream
// 4 x [re 20 | im 20] followed by the im stream 4 x [im 20 | -re 20])
template <bool QIMG>
__device__ __forceinline__ u64 spectra_of(const float *__restrict__ d, const double *__restrict__ nrm, double *xn,
                                          _Float16 *kv, int lane, float &sqrt_a) {
  bool nonzero = false, bad = false;
  if (lane < NS) {
    const double n = nrm[lane];
    nonzero = !(n == 0.0);  // SC.cpp:78: a column takes part unless its norm == 0
    const double rn = 1.0 / n;  // one division per column (x * (1 / n): 2 ulp of fp64 from x / n, the image is fp16)
#pragma unroll
    for (int r = 0; r < NR; r++) {
      const double y = nonzero ? (double)d[lane * NR + r] * rn : 0.0;
      bad |= !(fabs(y) <= 1.0000001);  // NaN, inf (or a norm that is not the column's)
      xn[r * NS + lane] = y;
    }
  }
  u64 m = __ballot(nonzero && lane < NS);
  if (__ballot(bad && lane < NS)) m |= kNonFinite;
  wave_lds_fence();
  // twiddles e^(-2 pi i m / 15), in the tail of the (now consumed-in-place) scratch: xn has DS doubles, tw sits behind
  double *tw = xn + DS;  // [0..15) cos, [16..31) sin
  if (lane < 15) {
    tw[lane] = cospi(2.0 * lane / 15.0);
    tw[16 + lane] = sinpi(2.0 * lane / 15.0);
  }
  wave_lds_fence();
  double dc_energy = 0.0;  // this lane's share of sum |X_0|^2
  // One (a, r) pair per lane and pass (80 pairs: lanes 0..63, then 0..15): the pair's 15 column values are read from LDS once
  // and serve all 8 frequencies, the twiddles sit in registers.  (Until round 4 a lane took one (f, a, r) output at a time and
  // read x and two twiddles from LDS for every term: 450 ds_read_b64 per lane instead of 30 + 30.)  Every output is the same
  // left-to-right sum over b as before, and the f = 0 outputs stay on the same lanes in the same order: the image and
  // dc_energy are bit-identical.
  double twc[15], tws[15];
#pragma unroll
  for (int t = 0; t < 15; t++) {
    twc[t] = tw[t];
    tws[t] = tw[16 + t];
  }
#pragma unroll 1
  for (int idx = lane; idx < 80; idx += 64) {
    const int a = idx / NR, r = idx - a * NR;
    int c = 45 * a;  // CRT: c = a mod 4, c = b mod 15  ->  c = (45 a + 16 b) mod 60
    c -= (c >= 120) ? 120 : 0;
    c -= (c >= 60) ? 60 : 0;
    double x[15];
#pragma unroll
    for (int bb = 0; bb < 15; bb++) {
      x[bb] = xn[r * NS + c];
      c += 16;
      c -= (c >= 60) ? 60 : 0;
    }
#pragma unroll
    for (int f = 0; f < 8; f++) {
      double re = 0.0, im = 0.0;
#pragma unroll
      for (int bb = 0; bb < 15; bb++) {
        const int t = (f * bb) % 15;
        re += x[bb] * twc[t];
        im -= x[bb] * tws[t];
      }
      if (f == 0) {
        kv[a * 24 + r] = (_Float16)(float)re;
        dc_energy += re * re;
      } else if constexpr (!QIMG) {
        const int base = 96 + (f - 1) * 160 + a * 40 + r;
        kv[base] = (_Float16)(float)re;
        kv[base + 20] = (_Float16)(float)im;
      } else {
        const int base = (SP_DC_BYTES + (f - 1) * SP_F_BYTES) / 2 + a * 40 + r;
        const _Float16 hr = (_Float16)(float)re, hi = (_Float16)(float)im;
        kv[base] = hr;
        kv[base + 20] = hi;
        kv[base + SP_VS / 2] = hi;
        kv[base + SP_VS / 2 + 20] = -hr;
      }
    }
  }
  if (lane < 16) kv[(lane >> 2) * 24 + 20 + (lane & 3)] = (_Float16)0.0f;  // K padding of f = 0
  wave_lds_fence();
  // energy outside f = 0 (error budget of stage 2): a = n - sum |X_0|^2 / 15, rounded up a little
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) dc_energy += __shfl_xor(dc_energy, off);
  const double a_out = (double)__popcll(m & kMask60) - dc_energy / 15.0;
  sqrt_a = (a_out > 0.0) ? (float)(sqrt(a_out) * (1.0 + 1e-6)) + 1e-6f : 1e-6f;
  return m;
}

// database image: tile-major [tile of 32 entries][76 K-steps][64 lanes][8 halves]
__global__ __launch_bounds__(256) void sc_spec_db_kernel(const float *__restrict__ desc, const double *__restrict__ norm,
                                                         int64_t first, int64_t count, uint4 *__restrict__ spT,
                                                         float *__restrict__ aux) {
  __shared__ double xn[4][DS + 32];
  __shared__ __attribute__((aligned(16))) _Float16 kv[4][SP_KV];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t it = (int64_t)blockIdx.x * 4 + wave;
  if (it >= count) return;
  const int64_t slot = first + it;
  float sqrt_a;
  (void)spectra_of<false>(desc + slot * DS, norm + slot * NS, xn[wave], kv[wave], lane, sqrt_a);
  if (lane == 0) aux[slot] = sqrt_a;
  const int64_t tile = slot >> 5;
  const int col = (int)(slot & 31);
  for (int c = lane; c < 2 * SP_FRAGS; c += 64)
    spT[(tile * SP_FRAGS + (c >> 1)) * 64 + (c & 1) * 32 + col] = *reinterpret_cast<const uint4 *>(&kv[wave][c * 8]);
}

// ------------------------------------------------------------------------------------------
// One new keyframe in ONE launch (the reference's makeAndSaveScancontextAndKeys, SC.cpp:229-247, plus everything the
// query path keeps per entry): the 256-thread block builds the descriptor and its keys from the cloud (sc_entry_dev.h
// build_block == sc_build_kernel), then wave 0 writes the spectral image, wave 1 the fp16 image (tile-major and
// entry-major) and wave 2 the fp16 sector-key image.  Before round 4 these were four dependent launches of one block or one
// wave each (9 + 4.6 + 10 + 4.5 us of kernels, ~60 us per insert with their gaps and the synchronise).
// blockIdx.x = cloud: points [offs[b], offs[b + 1]) of pts (offs = nullptr: one cloud of n_pts points), slot first + b.
// ------------------------------------------------------------------------------------------
struct InsertArgs {
  const char *pts;
  const int64_t *offs;
  int64_t n_pts, stride;
  double lidar_height, max_radius;
  int64_t first;
  float *desc;
  double *vkey, *norm;
  float *rkey;
  uint4 *hnT, *hnR;
  u64 *cmask;
  uint4 *spT;
  float *aux;
  _Float16 *vk16;
  float *vk_n;
};

template <int SO>
__global__ __launch_bounds__(256) void sc_insert_kernel(InsertArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned bins[DS];
  __shared__ double xn[DS + 32];
  __shared__ __attribute__((aligned(16))) _Float16 kv[SP_KV];
  __shared__ __attribute__((aligned(16))) _Float16 st[DS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t slot = a.first + blockIdx.x;
  const int64_t p0 = a.offs ? a.offs[blockIdx.x] : 0, p1 = a.offs ? a.offs[blockIdx.x + 1] : a.n_pts;
  dev::build_block<SO>(a.pts + p0 * a.stride, p1 - p0, a.stride, a.lidar_height, a.max_radius, bins, a.desc + slot * DS,
                   a.vkey + slot * NS, a.norm + slot * NS, a.rkey + slot * NR);
  // the images read the descriptor and its norms back from global memory: written by this block, visible to it after the
  // barrier (the same L1 / write-through path)
  __threadfence_block();
  __syncthreads();
  if (wave == 0) {
    float sqrt_a;
    (void)spectra_of<false>(a.desc + slot * DS, a.norm + slot * NS, xn, kv, lane, sqrt_a);
    if (lane == 0) a.aux[slot] = sqrt_a;
    const int64_t tile = slot >> 5;
    const int col = (int)(slot & 31);
    for (int c = lane; c < 2 * SP_FRAGS; c += 64)
      a.spT[(tile * SP_FRAGS + (c >> 1)) * 64 + (c & 1) * 32 + col] = *reinterpret_cast<const uint4 *>(&kv[c * 8]);
  } else if (wave == 1) {
    dev::img_db_entry(a.desc, a.norm, slot, st, a.hnT, a.hnR, a.cmask, lane);
  } else if (wave == 2) {
    dev::win_db_keys_entry(a.vkey, slot, a.vk16, a.vk_n, lane);
  }
}

// query images (see the layout constants): stream image = the LDS layout of the filter kernel
//   [0, 192)                  f = 0 stream: 4 blocks (a = 0..3) x 24 halves
//   [192 + (f-1)*640 ...)     re stream: 4 blocks x [Qr | Qi];  + 320: im stream: 4 blocks x [Qi | -Qr]
//   [4672, 4688)              n_q, flags, sqrt(n_q), sqrt(a_q)
// mask image = 16 displaced copies of the column-mask byte stream; flag byte = the query has an empty column
__global__ __launch_bounds__(256) void sc_spec_query_kernel(const float *__restrict__ desc, const double *__restrict__ norm,
                                                            int32_t nq, char *__restrict__ qimg) {
  __shared__ double xn[4][DS + 32];
  __shared__ __attribute__((aligned(16))) _Float16 img[4][SP_TAIL / 2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = blockIdx.x * 4 + wave;
  if (q >= nq) return;
  float sqrt_a;
  const u64 m = spectra_of<true>(desc + (int64_t)q * DS, norm + (int64_t)q * NS, xn[wave], img[wave], lane, sqrt_a);
  const int n = __popcll(m & kMask60);
  uint4 *out = reinterpret_cast<uint4 *>(qimg + (int64_t)q * SP_QS);
  for (int c = lane; c < SP_QS / 16; c += 64) {
    uint4 o = uint4{0u, 0u, 0u, 0u};
    if (c < SP_TAIL / 16) o = *reinterpret_cast<const uint4 *>(&img[wave][c * 8]);
    else if (c == SP_TAIL / 16)
      o = uint4{(unsigned)n, (m & kNonFinite) ? 1u : 0u, __float_as_uint(sqrtf((float)n)), __float_as_uint(sqrt_a)};
    out[c] = o;
  }
  uint4 *mout = reinterpret_cast<uint4 *>(qimg + sp_masks_at(nq) + (int64_t)q * SP_MASK_BYTES);
  for (int c = lane; c < SP_MASK_BYTES / 16; c += 64) {
    const int byte0 = c * 16;
    const int j = byte0 / SP_MASK_COPY, i0 = byte0 - j * SP_MASK_COPY;  // 112 = 7 chunks: no chunk straddles copies
    unsigned w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < 16; i++) w[i >> 2] |= (unsigned)(((m >> ((j + i0 + i) % NS)) & 1ull) * 0x38ull) << (8 * (i & 3));
    mout[c] = uint4{w[0], w[1], w[2], w[3]};
  }
  if (lane == 0) reinterpret_cast<unsigned char *>(qimg + sp_flags_at(nq))[q] = (n != NS) ? 1 : 0;
}

// ------------------------------------------------------------------------------------------
// the filter
// ------------------------------------------------------------------------------------------
struct SpecArgs {
  const uint4 *spT;
  const u64 *cmask;
  const float *aux;  // [n] sqrt of the entry's spectral energy outside f = 0
  const char *qimg;
  int64_t n_items;
  int32_t nq;
  int32_t unit_len;   // no plan: workgroup u = (query range u / ntb, tile-block u % ntb), unit_len query tiles per range
  lb_t *lb;
  int64_t ld_lb;
  float eps_direct;
  int32_t dbg;  // timing experiments of RSX_SPEC_INSTRUMENT builds (RSX_SPEC_DBG): 4 = no DMA, 8 = no stores
  unsigned long long *prof;  // RSX_SPEC_PROF: s_memtime sums per region of (workgroup 0, wave 0)
  const int32_t *tb_qmin;  // optional plan, in query-tile units (see sc_filter.hip)
  const int64_t *tb_cum;
  // sc_spec2_filter_kernel without a plan (xcd_len > 0): query tiles per XCD and per sub-range (see the kernel's work split)
  int32_t xcd_nqt, xcd_len;
};

// the DMA of a later tile, issued piecewise from inside the tail of the current one: an LDS-DMA instruction
// costs 100-185 cycles of issue next to ds_reads (stage 1) and 25-60 in VALU-only stretches (the tail)
struct TileDma {
  const char *gsrc;    // stream images of the tile's queries
  const char *gsrc_m;  // their mask images
  char *ldst;          // LDS tile buffer
  int nbytes;          // 0: nothing to load
  int nbytes_m;        // 0: no query of the tile has an empty column
};
// where the bounds of a tile go: a wave-uniform row base (&lb[first query of the tile][0], scalar registers) plus a
// 32-bit per-lane byte offset -- a 64-bit per-lane address was spilled to scratch, and every scratch reload in the
// tile loop is followed by s_waitcnt vmcnt(0): a full drain of the DMA pieces and bound stores in flight
struct TileOut {
  char *rowbase;      // uniform
  unsigned lane_off;  // (this lane's entry + (lane >= 32 ? ld : 0)) * sizeof(lb_t): lanes 32..63 store the odd query of a pair
  unsigned ld_bytes;  // uniform
  int nq_here;        // valid queries in the tile
  bool n_ok;          // this lane's entry exists
};
// The pieces of a tile are numbered 0 .. SP_STREAM_PIECES-1 (streams) and SP_STREAM_PIECES .. +SP_MASK_PIECES-1 (masks);
// piece c belongs to wave c % 4.  Number of DMA instructions the wave issues for the tile (each counts once on vmcnt):
__device__ __forceinline__ int dma_pieces_of_wave(const TileDma &d, int wave) {
  const int np = (d.nbytes + 1023) >> 10, npm = (d.nbytes_m + 1023) >> 10;
  const int w_m = (wave - SP_STREAM_PIECES) & 3;  // first mask piece of this wave
  return (np > wave ? (np - wave + 3) >> 2 : 0) + (npm > w_m ? (npm - w_m + 3) >> 2 : 0);
}
// pieces j0 .. j0+count-1 of this wave (piece j of wave w is chunk w + 4 j)
__device__ __forceinline__ void dma_issue(const TileDma &d, int wave, int lane, int j0, int count) {
  const int wu = __builtin_amdgcn_readfirstlane(wave);  // scalar piece number: scalar base address and LDS address
  for (int j = j0; j < j0 + count; j++) {
    const int c = wu + 4 * j;
    const bool is_mask = c >= SP_STREAM_PIECES;
    const int cc = is_mask ? c - SP_STREAM_PIECES : c;
    // scalar base + 32-bit lane offset, formed HERE: hoisted to the top of the tile the dozen 64-bit addresses
    // do not fit the register file and come back from scratch
    unsigned off = (unsigned)(cc * 1024 + lane * 16);
    asm volatile("" : "+v"(off));
    if ((int)off < (is_mask ? d.nbytes_m : d.nbytes))
      __builtin_amdgcn_global_load_lds(
          reinterpret_cast<const AS1 void *>(reinterpret_cast<uintptr_t>(is_mask ? d.gsrc_m : d.gsrc) + off),
          (AS3 void *)(d.ldst + c * 1024), 16, 0, 0);
  }
}

// one dword through the scalar cache, for a wave-uniform address, waited for on the spot (lgkmcnt(0): only where no
// LDS read is meant to stay in flight).  Left to the compiler the flag words came through global_load_dword, and the
// s_waitcnt vmcnt(0) before their use drained the DMA queue every tile.
__device__ __forceinline__ unsigned scalar_load_u32(const unsigned *p) {
  unsigned v;
  asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
  return v;
}

// s_waitcnt vmcnt(n) for a wave-uniform run-time n (the instruction takes an immediate)
__device__ __forceinline__ void wait_vmcnt_le(int n) {
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
  }
}

__device__ __forceinline__ half8 lds_frag(const char *p) { return *reinterpret_cast<const half8 *>(p); }

__device__ __forceinline__ unsigned pack2(float a, float b) {
  float2v f = {a, b};
  half2v h = __builtin_convertvector(f, half2v);  // round to nearest even (v_cvt_pk_f16_f32)
  return __builtin_bit_cast(unsigned, h);
}

typedef int intx8 __attribute__((ext_vector_type(8)));
typedef unsigned frag4 __attribute__((ext_vector_type(4)));  // one MFMA A fragment in flight (8 x fp16 / 16 x fp8)

// per-lane constants of one segment
struct SpecLane {
  unsigned dc_off, f_off;  // A-fragment offsets inside a tile of 4 query images (f = 0 / f >= 1 streams)
  unsigned c_dc, c_f;      // first 16-byte chunk of the lane's row in its stream: the read of K-step s wraps when c + 2 s >= 12 / 20
  unsigned m_off[2];       // mask-row offsets of the two n_eff M-tiles (k4 = 0,1 / 2,3) inside the tile buffer
  unsigned bpark;          // LDS byte address of this lane's parked B fragments (1 KiB apart)
  int hh;
  half8 W;                 // stage-2 A operand (inverse DFT weights)
  intx8 Bm;                // this lane's entry: column-mask bytes as fp8 0 / 1 (B operand of the n_eff MFMAs)
  int n_e;
  float sqrt_ne, sqrt_ae;
  float r_ne;  // rcp(max(n_e, 1)): 1 / n_lo of a query without empty columns
  bool e_bad;
};

// A-fragment reads are issued through inline asm with hand-counted s_waitcnt (see sc_filter.hip: left to
// the compiler every read is followed by s_waitcnt lgkmcnt(0), a full LDS round trip per MFMA).  LDS
// returns data in issue order, so "fragment t has landed" == "at most (number of reads issued after it)
// are outstanding".
__device__ __forceinline__ void lds_read_frag(frag4 &dst, unsigned addr, int off) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off));
}
// the A fragment of K-step s of a 4-block stream: lanes whose row has run off the end of the stream (chunk c + 2 s
// >= NCH) read from `wrapped` = addr - stream length.  Whether any lane wraps at step s is known at compile time
// (c <= CMAX), so the first steps cost nothing; the others one v_cmp + one v_cndmask in the slot's free VALU issue
template <int S, int NCH, int CMAX>
__device__ __forceinline__ void lds_read_stream(frag4 &dst, unsigned addr, unsigned wrapped, unsigned c, int off) {
  if constexpr (CMAX + 2 * S < NCH) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off));
  } else {
    unsigned a;
    asm volatile("v_cmp_lt_u32_e32 vcc, %4, %3\n\tv_cndmask_b32_e32 %1, %2, %5, vcc\n\tds_read_b128 %0, %1 offset:%6"
                 : "=v"(dst), "=&v"(a)
                 : "v"(addr), "v"(c), "n"(NCH - 1 - 2 * S), "v"(wrapped), "n"(off)
                 : "vcc");
  }
}
template <int N>
__device__ __forceinline__ void lds_wait_count() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N));
}

// region timer for RSX_SPEC_PROF (drains the LDS queue: s_memtime returns on lgkmcnt, out of order with LDS)
__device__ __forceinline__ unsigned long long prof_now() {
  unsigned long long t;
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}

constexpr int SP_S1 = SP_FRAGS;   // 76 stage-1 MFMA slots per tile
#ifndef SP_OPT_DEPTH
#define SP_OPT_DEPTH 8
#endif
constexpr int SP_DEPTH = SP_OPT_DEPTH;  // A fragments in flight
// per-query LDS reads of the tail: the query's {n_q, flags, sqrt n_q, sqrt a_q} words, requested one query ahead, and --
// only for a query with an empty column -- 4 mask reads (2 M-tiles x 32 bytes).  ALL of them go through inline asm: a
// plain C++ LDS load after a global_load_lds makes the compiler insert s_waitcnt vmcnt(0) (the DMA writes LDS, so it
// orders the load behind every outstanding VMEM operation) -- one full drain of the just-issued DMA pieces and bound
// stores per query, ~1 k cycles each (found in the ISA in round 2)
constexpr int SP_MREADS = 1;  // reads of query 0's tail data issued inside stage 1

// Stage-1 slot schedule.  Two frequencies are always in progress and their MFMAs alternate, so that no MFMA
// follows another one on the SAME accumulator with LDS reads / VALU instructions in between (that pattern
// loses the accumulator forwarding of back-to-back dependent MFMAs: ~+40 cycles per MFMA).  When a
// frequency finishes the next one takes its place: f0 (6 steps) and f1 start, f2 replaces f0, f3 replaces
// f1, ...  Frequency f accumulates into acc[f % 4].
struct S1Slot {
  int f, s;  // frequency, K-step
};
struct S1Sched {
  S1Slot slot[SP_S1];
  int done[8];  // slot of the last MFMA of frequency f
};
constexpr int s1_steps(int f) { return f == 0 ? SP_DC_STEPS : SP_F_STEPS; }
constexpr bool kS1Interleave = false;  // measured: alternating two frequencies is not faster than one after the other
constexpr S1Sched make_s1_sched() {
  S1Sched r{};
  if (!kS1Interleave) {
    int t = 0;
    for (int f = 0; f < 8; f++) {
      for (int st = 0; st < s1_steps(f); st++) r.slot[t++] = S1Slot{f, st};
      r.done[f] = t - 1;
    }
    return r;
  }
  int act[2] = {0, 1}, pos[2] = {0, 0}, next_f = 2, turn = 0;
  for (int t = 0; t < SP_S1; t++) {
    if (act[turn] < 0) turn ^= 1;
    r.slot[t] = S1Slot{act[turn], pos[turn]};
    pos[turn]++;
    if (pos[turn] == s1_steps(act[turn])) {
      r.done[act[turn]] = t;
      act[turn] = next_f < 8 ? next_f++ : -1;
      pos[turn] = 0;
    }
    if (act[turn ^ 1] >= 0) turn ^= 1;
  }
  return r;
}
constexpr S1Sched kS1 = make_s1_sched();
// LDS byte offset of the fragment of slot t (relative to the lane's f = 0 or f >= 1 stream base) and its B index
constexpr int s1_off(int t) { return kS1.slot[t].f == 0 ? 32 * kS1.slot[t].s : (kS1.slot[t].f - 1) * SP_F_BYTES + 32 * kS1.slot[t].s; }
constexpr int s1_b(int t) { return kS1.slot[t].f == 0 ? kS1.slot[t].s : SP_DC_STEPS + (kS1.slot[t].f - 1) * SP_F_STEPS + kS1.slot[t].s; }
// fp16 packing of the finished pair G_g = (f_2g, f_2g+1): 4 j per slot from two slots after the pair is
// complete; it must be over before f_2g+4 re-uses acc[2g % 4]
constexpr int pack_begin(int g) { return (kS1.done[2 * g] > kS1.done[2 * g + 1] ? kS1.done[2 * g] : kS1.done[2 * g + 1]) + 2; }
constexpr int first_slot_of(int f) {
  for (int t = 0; t < SP_S1; t++)
    if (kS1.slot[t].f == f) return t;
  return SP_S1;
}

// 1/n <= u(n) = A + n (Bc + n C) on [L, H] = [n_lo, n_hi]
struct Recip {
  float2v A2, B2, C2;
  float rL;
  int n_q;
  unsigned flags;
  float sqrt_nq, sqrt_aq;
};
__device__ __forceinline__ Recip recip_header(const frag4 &tail) {  // the query's tail words only
  Recip r;
  r.n_q = (int)tail[0];
  r.flags = tail[1];
  r.sqrt_nq = __uint_as_float(tail[2]);
  r.sqrt_aq = __uint_as_float(tail[3]);
  return r;
}
// the coefficients of u(n) for this lane's entry (queries with an empty column only: three reciprocals and a dozen
// VALU instructions per query that a query with 60 non-empty columns does not need -- its 1 / n_lo is the lane constant r_ne)
__device__ __forceinline__ void recip_coeffs(Recip &r, int n_e) {
  const int li = (r.n_q + n_e - NS > 1) ? (r.n_q + n_e - NS) : 1;
  const int hmin = r.n_q < n_e ? r.n_q : n_e;
  const float L = (float)li, H = (float)(hmin > li ? hmin : li);
  const float rH = __builtin_amdgcn_rcpf(H), rLH = __builtin_amdgcn_rcpf(L * H);
  const float C = rLH * rH;
  const float A = fmaf(L + H, rLH, rH), Bc = -fmaf(C, L + H, rLH);
  r.A2 = float2v{A, A};
  r.B2 = float2v{Bc, Bc};
  r.C2 = float2v{C, C};
  r.rL = __builtin_amdgcn_rcpf(L);
}

// One (4 queries x 32 entries) tile at LDS byte address `tile_lds`.
//   slots 0..75    stage 1: one MFMA per slot, fragment t+SP_DEPTH requested right after MFMA t; frequency f
//                  accumulates into acc[f % 3], and the fp16 packing of a finished frequency pair
//                  (P[j][g] = {C_2g, C_2g+1} of (query j/4, k4 = j%4)) is spread over the slots of the next frequency
//   tail           per query: n_eff of the 60 shifts (2 fp8 MFMAs, K = 64), 4 stage-2 MFMAs (inverse DFT of one
//                  k4 each), S * u(n_eff) and the running maximum in packed fp32; the tail words of the next
//                  query are in flight meanwhile
// out[q] = the bound of (query q, this lane's entry), identical in both lane halves
#ifndef SP_OPT_PACK
#define SP_OPT_PACK 2   // j packed per slot
#endif
constexpr int kPackPerSlot = SP_OPT_PACK;
constexpr int kPackSlots = 16 / kPackPerSlot;
// three accumulator sets, frequency f -> acc[f % 3]: the pair (f_2g, f_2g+1) is packed while f_2g+2 runs and
// must be done before f_2g+3 re-uses the set of f_2g; the hi/lo split of C_0 (in place) runs during f1
constexpr int kAccSets = 3;
static_assert(pack_begin(0) + kPackSlots <= first_slot_of(3) && pack_begin(1) + kPackSlots <= first_slot_of(5) &&
                  pack_begin(2) + kPackSlots <= first_slot_of(7), "accumulators are re-used before they are packed");
constexpr int split_begin() { return kS1.done[0] + 2; }
static_assert(s1_b(0) == 0 && s1_b(SP_B_LDS - 1) == SP_B_LDS - 1 && SP_B_LDS <= SP_DEPTH, "the parked fragments are the first slots");
static_assert(!kS1Interleave && split_begin() + 8 <= pack_begin(0) && split_begin() + 8 <= first_slot_of(3), "C_0 split window");

__device__ __forceinline__ void spec_tile(unsigned tile_lds, const char *tbase, const half8 (&B)[SP_FRAGS], const SpecLane &ln,
                                          float eps_direct, float (&out)[SP_QPT], int dbg, unsigned long long *tmid,
                                          const TileDma &dma, int wave, int lane, const TileOut &to) {
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  typedef unsigned u8v __attribute__((ext_vector_type(8)));
  u4 P[16];
  floatx16 acc[kAccSets];
  frag4 ring[SP_DEPTH];
  frag4 tw;     // {n_q, flags, sqrt n_q, sqrt a_q} of the current query, requested one query ahead
  frag4 mk[4];  // mask bytes (queries with an empty column only)
  const unsigned a_dc = tile_lds + ln.dc_off, a_f = tile_lds + ln.f_off;
  const unsigned a_dcw = a_dc - SP_DC_BYTES, a_fw = a_f - SP_VS;
  const unsigned a_m0 = tile_lds + ln.m_off[0], a_m1 = tile_lds + ln.m_off[1];
  const unsigned a_tl = tile_lds + SP_TAIL;
  // the A fragment of stage-1 slot t
  auto a_read = [&](frag4 &dst, auto tc) {
    constexpr int t = decltype(tc)::value;
    if constexpr (kS1.slot[t].f == 0) lds_read_stream<kS1.slot[t].s, SP_DC_BYTES / 16, 10>(dst, a_dc, a_dcw, ln.c_dc, s1_off(t));
    else lds_read_stream<kS1.slot[t].s, SP_VS / 16, 16>(dst, a_f, a_fw, ln.c_f, s1_off(t));
  };
  floatx16 z;
#pragma unroll
  for (int i = 0; i < 16; i++) z[i] = 0.0f;

  frag4 bx[SP_B_LDS];  // the parked B fragments
  static_for<SP_DEPTH>([&](auto tc) { a_read(ring[decltype(tc)::value], tc); });
  static_for<SP_B_LDS>([&](auto tc) {
    constexpr int t = decltype(tc)::value;
    lds_read_frag(bx[t], ln.bpark, 1024 * t);
  });
  static_for<SP_S1>([&](auto tc) {
    constexpr int t = decltype(tc)::value;
    constexpr int f = kS1.slot[t].f;
    // reads issued after fragment t's: fragments t+1 .. min(t+DEPTH-1, S1-1), plus the mask reads of query 0
    // (issued at slots S1-4 .. S1-1, each after that slot's ring read slot is gone)
    constexpr int younger_ring = (t + SP_DEPTH - 1 < SP_S1 ? SP_DEPTH - 1 : SP_S1 - 1 - t);
    constexpr int younger_mask = (t > SP_S1 - SP_MREADS) ? (t - (SP_S1 - SP_MREADS)) : 0;
    // slots 0..3 also need parked fragment t, read after the SP_DEPTH prologue fragments: younger than it are
    // the parked fragments t+1..3 and the ring reads of slots 0..t-1
    if constexpr (t < SP_B_LDS) lds_wait_count<SP_B_LDS - 1>();
    else lds_wait_count<younger_ring + younger_mask>();
    __builtin_amdgcn_sched_barrier(0);
    const half8 af = __builtin_bit_cast(half8, ring[t % SP_DEPTH]);
    half8 bf;
    if constexpr (t < SP_B_LDS) bf = __builtin_bit_cast(half8, bx[t]);
    else bf = B[s1_b(t)];
    if constexpr (kS1.slot[t].s == 0) acc[f % kAccSets] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf, z, 0, 0, 0);
    else acc[f % kAccSets] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf, acc[f % kAccSets], 0, 0, 0);
    if constexpr (t + SP_DEPTH < SP_S1) a_read(ring[t % SP_DEPTH], std::integral_constant<int, t + SP_DEPTH>{});
    if constexpr (t >= SP_S1 - SP_MREADS) lds_read_frag(tw, a_tl, 0);  // tail words of query 0
    if constexpr (t >= split_begin() && t < split_begin() + 8) {
      // C_0 as fp16 hi + lo: lanes 0..31 carry hi (k = 0), lanes 32..63 lo (k = 8); both weigh 1/16.  hi = C_0
      // truncated to 11 significant bits (exact in fp16), lo = the rest (rounded to fp16 by the packing)
#pragma unroll
      for (int j = 2 * (t - split_begin()); j < 2 * (t - split_begin()) + 2; j++) {
        const float v = acc[0][j];
        const float hi = __uint_as_float(__float_as_uint(v) & 0xffffe000u);
        float sp = ln.hh ? v - hi : hi;
        asm volatile("" : "+v"(sp));  // computed HERE, in the slot's free VALU issue (the compiler sinks it into the tail otherwise)
        acc[0][j] = sp;
      }
    }
    // packing of finished frequency pairs, spread over the slots of the following frequency
    static_for<3>([&](auto gc) {
      constexpr int g = decltype(gc)::value;
      if constexpr (t >= pack_begin(g) && t < pack_begin(g) + kPackSlots) {
#pragma unroll
        for (int j = kPackPerSlot * (t - pack_begin(g)); j < kPackPerSlot * (t - pack_begin(g) + 1); j++) {
          unsigned pk = pack2(acc[(2 * g) % kAccSets][j], acc[(2 * g + 1) % kAccSets][j]);
          asm volatile("" : "+v"(pk));  // pinned to this slot: left alone the compiler keeps all 8 accumulator sets alive and
          P[j][g] = pk;                 // packs in the tail, where the 64 conversions are not hidden by MFMAs
        }
      }
    });
    __builtin_amdgcn_sched_barrier(0);
  });
  if (kInstr && tmid) {
    asm volatile("" : "+v"(acc[0]), "+v"(acc[1]));
    *tmid = prof_now();
  }
#pragma unroll
  for (int j = 0; j < 16; j++) P[j][3] = pack2(acc[6 % kAccSets][j], acc[7 % kAccSets][j]);

  static_for<SP_QPT>([&](auto qc) {
    constexpr int q = decltype(qc)::value;
    lds_wait_count<0>();
    __builtin_amdgcn_sched_barrier(0);
    frag4 tailw = tw;
    asm volatile("" : "+v"(tailw));  // a value of its own: tw is re-loaded for the next query below
    Recip r = recip_header(tailw);
    if constexpr (q + 1 < SP_QPT) lds_read_frag(tw, a_tl, (q + 1) * SP_QS);
    dma_issue(dma, wave, lane, SP_DMA_PER_Q * q, SP_DMA_PER_Q);
    // A query whose 60 columns are all non-empty (the usual case for a radar scan) meets every entry with
    // n_eff(k) = n_e at EVERY shift (SC.cpp:78: a column pair is skipped only if one of the two is empty), so
    // [n_lo, n_hi] = {n_e}: no mask correlation, no u(n) -- max_k S_k u(n_k) = (max_k S_k) / n_e.  Wave-uniform.
    const bool full_q = __builtin_amdgcn_readfirstlane(r.n_q) == NS;
    float m = 0.0f;  // rows 15..31 of the weight matrix are zero anyway (S >= 0 or clamped: valid)
    floatx16 nacc[2];
    if (!full_q) {
      // the mask image of the tile is in LDS (the tile's flag word said so when its DMA was issued)
      lds_read_frag(mk[0], a_m0, q * SP_MASK_BYTES);
      lds_read_frag(mk[1], a_m0, q * SP_MASK_BYTES + 16);
      lds_read_frag(mk[2], a_m1, q * SP_MASK_BYTES);
      lds_read_frag(mk[3], a_m1, q * SP_MASK_BYTES + 16);
      lds_wait_count<0>();
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mt = 0; mt < 2; mt++) {
        const frag4 lo = mk[2 * mt], hi = mk[2 * mt + 1];
        const u8v am = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        nacc[mt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(__builtin_bit_cast(intx8, am), ln.Bm, z, 0, 0, 0, 0, 0, 0);
      }
    }
    if (full_q) {
      // two result sets: the MFMA of k4 + 1 is in the pipe while the maxima of k4 are taken (one set: every group of
      // v_max3 waited out its own MFMA; four sets: 64 live registers, spills -- see below)
      auto max8 = [&](const floatx16 &dd) {
#pragma unroll
        for (int e = 0; e < 4; e++) m = fmaxf(fmaxf(m, dd[2 * e]), dd[2 * e + 1]);  // one v_max3_f32 per pair
      };
      floatx16 d0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ln.W, __builtin_bit_cast(half8, P[q * 4 + 0]), z, 0, 0, 0);
      floatx16 d1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ln.W, __builtin_bit_cast(half8, P[q * 4 + 1]), z, 0, 0, 0);
      max8(d0);
      d0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ln.W, __builtin_bit_cast(half8, P[q * 4 + 2]), z, 0, 0, 0);
      max8(d1);
      d1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ln.W, __builtin_bit_cast(half8, P[q * 4 + 3]), z, 0, 0, 0);
      max8(d0);
      max8(d1);
      // (issuing the four MFMAs back to back into four result sets and taking the maxima afterwards was tried: the 64
      // live registers push two per-lane address registers to scratch, and their reload at the top of every tile
      // is followed by s_waitcnt vmcnt(0) -- a drain of the DMA in flight)
      r.rL = ln.r_ne;  // n_lo = n_hi = n_e: within 1 ulp of 1 / n_e, covered by the (1 + 4e-6) factor below
      m *= r.rL;
    } else {
    recip_coeffs(r, ln.n_e);
    // u(n) of all 32 n_eff values of the query first (independent of stage 2), then per k4: S * u and the maximum;
    // written stage by stage over 4 independent pairs so that no packed instruction waits for the previous one
    float2v u2[4][4];
#pragma unroll
    for (int k4 = 0; k4 < 4; k4++) {
      float2v t2[4];
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const float2v n2 = {nacc[k4 >> 1][(k4 & 1) * 8 + 2 * e], nacc[k4 >> 1][(k4 & 1) * 8 + 2 * e + 1]};
        t2[e] = __builtin_elementwise_fma(n2, r.C2, r.B2);
      }
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const float2v n2 = {nacc[k4 >> 1][(k4 & 1) * 8 + 2 * e], nacc[k4 >> 1][(k4 & 1) * 8 + 2 * e + 1]};
        u2[k4][e] = __builtin_elementwise_fma(n2, t2[e], r.A2);
      }
    }
#pragma unroll
    for (int k4 = 0; k4 < 4; k4++) {
      const floatx16 dd = __builtin_amdgcn_mfma_f32_32x32x16_f16(ln.W, __builtin_bit_cast(half8, P[q * 4 + k4]), z, 0, 0, 0);
      float2v v2[4];
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const float2v s2 = {dd[2 * e], dd[2 * e + 1]};
        v2[e] = s2 * u2[k4][e];
      }
#pragma unroll
      for (int e = 0; e < 4; e++) m = fmaxf(fmaxf(m, v2[e][0]), v2[e][1]);  // one v_max3_f32 per pair
    }
    }
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
    const float best = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));  // 15/16 max_k S_k u(n_k)
    // the error of S is divided by n_k >= n_lo
    const float err = kE1 * r.sqrt_nq * ln.sqrt_ne + kE2 * r.sqrt_aq * ln.sqrt_ae;
    float v = (1.0f + eps_direct) - fmaf(best, (16.0f / 15.0f) * (1.0f + 4e-6f), err * r.rL);
    if (r.n_q == 0 || ln.n_e == 0) v = INFINITY;    // no effective column at any shift: never a hit
    if (r.flags != 0u || ln.e_bad) v = -INFINITY;   // non-finite input: always re-score exactly
    out[q] = v;
    // both lane halves hold the bounds: lanes 0..31 store the even query of a pair, lanes 32..63 the odd one --
    // one store instruction per two queries, issued here inside the VALU-bound tail rather than after it
    if constexpr (q & 1) {
      const int qq = (q - 1) + ln.hh;
      const float vv = ln.hh ? out[q] : out[q - 1];
      if (to.n_ok && qq < to.nq_here && !(kInstr && (dbg & 8) && vv != 12345.0f))
        *reinterpret_cast<lb_t *>(to.rowbase + (size_t)((unsigned)(q - 1) * to.ld_bytes) + to.lane_off) = lb_pack(vv);
    }
  });
}

__global__ __launch_bounds__(256, 1) void sc_spec_filter_kernel(SpecArgs a) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 31, hh = lane >> 5;
  const int64_t ntiles = (a.n_items + 31) >> 5;
  const int64_t ntb = (ntiles + 3) >> 2;
  const int nqt = (a.nq + SP_QPT - 1) / SP_QPT;
  const int64_t total = a.tb_cum ? a.tb_cum[ntb] : ntb * (int64_t)nqt;
  // Work split.  With a plan: equal contiguous segments of the (tile-block, query tile) list.  Without: one
  // unit per workgroup = one tile-block x one contiguous range of query tiles, ranges outermost, so that the
  // workgroups in flight at any time (consecutive indices) stream the SAME query images at about the same
  // time: each image then comes out of HBM / Infinity Cache once per XCD and out of that XCD's L2 for the
  // other workgroups (the query stream is 10 KB per 4 x 128 pairs -- five times the direct filter's rate).
  int64_t L0, L1;
  if (a.tb_cum || !SP_OPT_LOCKSTEP) {
    const int64_t per = (total + gridDim.x - 1) / gridDim.x;
    L0 = (int64_t)blockIdx.x * per;
    L1 = (L0 + per < total) ? (L0 + per) : total;
  } else {
    const int64_t r = blockIdx.x / ntb, utb = blockIdx.x - r * ntb;
    const int64_t u0 = r * a.unit_len, u1 = (u0 + a.unit_len < nqt) ? (u0 + a.unit_len) : nqt;
    L0 = utb * nqt + u0;
    L1 = u0 < u1 ? utb * nqt + u1 : L0;
  }
  const unsigned lds_base = (unsigned)(uintptr_t)((AS3 char *)smem);
  SpecLane ln;
  ln.hh = hh;
  ln.bpark = lds_base + (unsigned)(SP_BPARK_OFF + wave * (SP_B_LDS * 1024) + lane * 16);
  {
    // A-fragment address of this lane inside a tile of 4 query images: row = col = 8 * query + 4 * variant + k4
    const int rq = col >> 3, rv = (col >> 2) & 1, rk = col & 3;
    ln.c_dc = rk * 3 + hh;
    ln.c_f = rk * 5 + hh;
    ln.dc_off = rq * SP_QS + ln.c_dc * 16;
    ln.f_off = rq * SP_QS + SP_DC_BYTES + rv * SP_VS + ln.c_f * 16;
    // n_eff rows: row = col <-> (k4 = 2 * mt + col / 16, k15 = col % 16), the row order of the stage-2 output
#pragma unroll
    for (int mt = 0; mt < 2; mt++) {
      const int k4 = 2 * mt + (col >> 4), k15 = (col & 15) == 15 ? 0 : (col & 15);
      const int k = (45 * k4 + 16 * k15) % NS;  // CRT
      ln.m_off[mt] = SP_MASKREG_OFF + (k & 15) * SP_MASK_COPY + (k & ~15) + hh * 32;
    }
    // stage-2 A operand: row k15 = col (rows >= 15 are zero), k = part index: lanes 0..31 {C_0 hi, Re C_1..7},
    // lanes 32..63 {C_0 lo, Im C_1..7}; weights scaled by 15/16
#pragma unroll
    for (int i = 0; i < 8; i++) {
      float w = 0.0f;
      if (col < 15) {
        if (i == 0) w = 1.0f / 16.0f;
        else {
          const int t = (i * col) % 15;
          w = hh ? (float)(-sinpi(2.0 * t / 15.0) / 8.0) : (float)(cospi(2.0 * t / 15.0) / 8.0);
        }
      }
      ln.W[i] = (_Float16)w;
    }
  }

  int64_t tb = 0;
  if (a.tb_cum && L0 < L1) {  // last tile-block whose first item is <= L0
    int64_t lo = 0, hi = ntb - 1;
    while (lo < hi) {
      const int64_t mid = (lo + hi + 1) >> 1;
      if (a.tb_cum[mid] <= L0) lo = mid;
      else hi = mid - 1;
    }
    tb = lo;
  }
  while (L0 < L1) {
    int t0;
    if (a.tb_cum) {
      while (a.tb_cum[tb + 1] <= L0) tb++;  // skip tile-blocks without items
      t0 = a.tb_qmin[tb] + (int)(L0 - a.tb_cum[tb]);
    } else {
      tb = L0 / nqt;
      t0 = (int)(L0 - tb * nqt);
    }
    const int t1 = (L1 - L0 < (int64_t)(nqt - t0)) ? (int)(t0 + (L1 - L0)) : nqt;
    L0 += t1 - t0;
    const int q0 = t0 * SP_QPT;
    const int q1 = (t1 * SP_QPT < a.nq) ? t1 * SP_QPT : a.nq;
    const int64_t tile = tb * 4 + wave;
    const bool tile_ok = tile < ntiles;  // wave-uniform
    const int64_t n = tile * 32 + col;
    const bool n_ok = tile_ok && n < a.n_items;
    const int nphase = (q1 - q0 + SP_QPP - 1) / SP_QPP;

    // flag word of query tile t (4 flag bytes): non-zero when one of its queries has an empty column and the tile
    // needs its mask image.  Wave-uniform scalar loads, requested a tile before they are used
    const unsigned *qflags = reinterpret_cast<const unsigned *>(a.qimg + sp_flags_at(a.nq)) + __builtin_amdgcn_readfirstlane(t0);
    auto tile_dma = [&](int p, unsigned flagword) {  // query tile p of this segment -> LDS buffer p % SP_NBUF
      const int qn = q0 + p * SP_QPP;
      const int nqs = (q1 - qn < SP_QPP) ? (q1 - qn) : SP_QPP;
      return TileDma{a.qimg + (int64_t)qn * SP_QS, a.qimg + sp_masks_at(a.nq) + (int64_t)qn * SP_MASK_BYTES,
                     smem + SP_TILES_OFF + (p % SP_NBUF) * SP_PHASE_BYTES, nqs * SP_QS, flagword ? nqs * SP_MASK_BYTES : 0};
    };
    auto stage_tile = [&](int p) {
      if (kInstr && (a.dbg & 4)) return;
      dma_issue(tile_dma(p, scalar_load_u32(qflags + p)), wave, lane, 0, SP_DMA_PER_Q * SP_QPT);
    };
    stage_tile(0);  // overlaps the B loads below
    if (SP_NBUF > 2 && nphase > 1) stage_tile(1);
    unsigned flag_next = (SP_NBUF - 1 < nphase) ? scalar_load_u32(qflags + SP_NBUF - 1) : 0u;  // of tile p + SP_NBUF - 1, p = 0
    half8 B[SP_FRAGS];
    {
      const uint4 *src = a.spT + ((tile_ok ? tile : 0) * SP_FRAGS) * 64 + lane;
#pragma unroll
      for (int s = 0; s < SP_FRAGS; s++) {
        const uint4 v = src[s * 64];
        B[s] = *reinterpret_cast<const half8 *>(&v);
      }
#pragma unroll
      for (int s = SP_B_VGPR; s < SP_FRAGS; s++) asm volatile("" : "+a"(B[s]));
      // park fragments 0..SP_B_LDS-1 in LDS (this wave's private 4 KiB; the previous segment's tiles are done)
#pragma unroll
      for (int s = 0; s < SP_B_LDS; s++)
        *reinterpret_cast<half8 *>(smem + SP_BPARK_OFF + wave * (SP_B_LDS * 1024) + s * 1024 + lane * 16) = B[s];
    }
    const u64 m2 = n_ok ? a.cmask[n] : 0ull;
    ln.n_e = __popcll(m2 & kMask60);
    ln.sqrt_ne = sqrtf((float)ln.n_e);
    ln.r_ne = __builtin_amdgcn_rcpf((float)(ln.n_e > 1 ? ln.n_e : 1));
    ln.sqrt_ae = n_ok ? a.aux[n] : 0.0f;
    ln.e_bad = (m2 & kNonFinite) != 0;
    {
      const unsigned bits = (unsigned)((m2 & kMask60) >> (32 * hh));  // columns 32 hh .. 32 hh + 31 (K index of the lane half)
#pragma unroll
      for (int r = 0; r < 8; r++)  // 4 bits -> 4 bytes of fp8 (e4m3) 0.0 / 1.0 = 0x00 / 0x38
        ln.Bm[r] = (int)(((((bits >> (4 * r)) & 0xfu) * 0x00204081u) & 0x01010101u) * 0x38u);
    }

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // Tile p is computed from buffer p % 3 while the DMA of tile p + 2 is being issued (piecewise, inside the
    // tail of tile p, interleaved with the tile's two bound stores).  At the end of tile p the wave waits for
    // ITS pieces of tile p + 1 only: vmcnt(n_young), n_young = the number of DMA instructions of tile p + 2 it
    // has issued.  VMEM operations retire in issue order and at least n_young + 3 operations are younger than
    // the last piece of tile p + 1 (a store of tile p - 1, the pieces of tile p + 2, the stores of tile p), so
    // that piece has landed; the youngest stores stay in flight (waiting for their acknowledgement every
    // tile cost more than the stage-1 MFMAs).  The raw barrier (no vmcnt(0), unlike __syncthreads with an
    // LDS-DMA pending) makes the other waves' pieces visible and frees buffer (p + 3) % 3 = p % 3.
    const bool prof = kInstr && a.prof && blockIdx.x == 0 && wave == 0;
    const bool clk = a.prof && blockIdx.x == 0 && wave == 0;  // any build: the segment's cycles and wall time
    unsigned long long ps[5] = {0, 0, 0, 0, 0};
    unsigned long long c_begin = 0, r_begin = 0;  // shader clock vs the constant 100 MHz clock: the clock the kernel ran at
    if (clk) {
      c_begin = prof_now();
      asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r_begin)::"memory");
    }
    for (int p = 0; p < nphase; p++) {
      const int qp = q0 + p * SP_QPP;
      unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
      if (prof) t0 = prof_now();
      // the DMA of tile p + SP_NBUF - 1: issued piecewise inside this tile's tail (waves without a tile issue it here)
      TileDma dma{nullptr, nullptr, nullptr, 0, 0};
      if (p + SP_NBUF - 1 < nphase && !(kInstr && (a.dbg & 4))) dma = tile_dma(p + SP_NBUF - 1, flag_next);
      const int n_issued = dma_pieces_of_wave(dma, wave);
      if (!tile_ok) dma_issue(dma, wave, lane, 0, SP_DMA_PER_Q * SP_QPT);
      const int n_young = SP_NBUF > 2 ? n_issued : 0;  // DMA instructions younger than those of tile p + 1
      if (prof) t1 = prof_now();
      const int nq_here = (q1 - qp < SP_QPP) ? (q1 - qp) : SP_QPP;
      if (tile_ok) {
        const char *tbase = smem + SP_TILES_OFF + (p % SP_NBUF) * SP_PHASE_BYTES;
        float out[SP_QPT];
        spec_tile(lds_base + (unsigned)(tbase - smem), tbase, B, ln, a.eps_direct, out, a.dbg, prof ? &t2 : nullptr, dma, wave, lane,
                  TileOut{reinterpret_cast<char *>(a.lb + (int64_t)qp * a.ld_lb), (unsigned)(n * 2) + (unsigned)hh * (unsigned)(a.ld_lb * 2),
                          (unsigned)(a.ld_lb * 2), nq_here, n_ok});
        if (prof) {
          asm volatile("" : "+v"(out[0]), "+v"(out[1]), "+v"(out[2]), "+v"(out[3]));
          t3 = prof_now();
        }
      }
      if (prof) t4 = prof_now();
      if (p + 1 < nphase) {
        // flag word of the tile whose DMA the next iteration issues; its latency falls into the wait for the DMA
        flag_next = (p + SP_NBUF < nphase) ? scalar_load_u32(qflags + p + SP_NBUF) : 0u;
        wait_vmcnt_le(n_young);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
      if (prof) {
        const unsigned long long t5 = prof_now();
        ps[0] += t1 - t0;  // DMA issue
        ps[1] += t2 - t1;  // stage 1
        ps[2] += t3 - t2;  // tail
        ps[3] += t4 - t3;  // stores
        ps[4] += t5 - t4;  // wait + barrier
      }
    }
    if (clk) {
      unsigned long long r_end;
      const unsigned long long c_end = prof_now();
      asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r_end)::"memory");
      if (lane == 0) {
        if (prof) for (int i = 0; i < 5; i++) a.prof[i] = ps[i];
        a.prof[5] = (unsigned long long)nphase;
        a.prof[6] = c_end - c_begin;
        a.prof[7] = r_end - r_begin;
      }
    }
    // segment end: everything retired before the next segment's DMA reuses the buffers
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
}


// ==========================================================================================
// sc_spec2_filter_kernel -- the same filter with TWO waves per SIMD (round 4)
// ==========================================================================================
// sc_spec_filter_kernel keeps the 76 B fragments of a 32-entry tile (304 registers) in ONE wave, so a SIMD holds a
// single wave and everything a tile needs -- 76 stage-1 MFMAs, their LDS reads, the fp16 packing, 16 stage-2 MFMAs,
// the maxima, the bound arithmetic, the DMA of the next tile -- issues one after the other on it: the matrix pipe is
// busy 39 % of the time.  Here the entry tile is split BY FREQUENCY over the two waves w and w + 4 of a
// 512-thread workgroup (they share SIMD w % 4):
//   "late"  wave (0..3): f = 0..3 (B fragments 0..35),  stage 2 + bounds + store of queries 2, 3 of a tile
//   "early" wave (4..7): f = 4..7 (B fragments 36..75), stage 2 + bounds + store of queries 0, 1, and every LDS-DMA piece
// Both run stage 1 on the SAME (4 queries x 32 entries) tile for their own frequencies (the A fragments of a frequency
// are read by one wave only, so the LDS traffic per flop is unchanged), pack their C_f to fp16 and hand the halves of
// the partner's two queries over through LDS (16 registers = 4 KiB per wave and tile).  The two waves are HALF A TILE
// OUT OF PHASE: in interval k the late wave runs [stage 1 of tile k | tail of tile k - 1] and the early wave
// [tail of tile k - 1 | DMA of tile k + 1 | stage 1 of tile k], so one wave's VALU-heavy tail always runs beside the other's
// MFMAs.  The late wave carries its own packed halves of its two queries (16 registers) across its next stage 1 -- it
// has 4 B fragments fewer than the early one, which is exactly that.  ONE s_barrier per interval (it publishes the DMA
// pieces of the next tile and orders the exchange between intervals); within an interval each single-buffered exchange
// direction is guarded by a sequence number: the reader bumps it once the half is in its registers, the writer checks
// it before overwriting (the reads come half an interval before the writes: the check never waits in practice).
// Round-4 builds before this one, both measured at 1.76-1.80 ms per 8192 x 9970 launch against 2.05 for the one-wave
// kernel: (i) symmetric, in phase, two barriers per tile -- both waves sat in their tails together, the pipe idle;
// (ii) producer (f = 4..7 + DMA, one tile ahead) / consumer (f = 0..3 + all four tails) -- during stage 1 the two waves
// alternated on the matrix pipe at exactly 64 cycles per own MFMA (pipe 100 % busy), but the consumer's serial tail
// (2.9 k cycles, ~350 issue states) then ran with the pipe idle.  A static s_setprio on either wave starved the other.
// Everything a lane needs only in the tail (stage-2 weights, the entry's constants, its mask bytes) lives in LDS, not
// in registers: at 256 registers per wave the B fragments leave ~110 for accumulators, the A ring and the packing.
#ifndef S2_OPT_PARK0
#define S2_OPT_PARK0 5   // B fragments of the consumer parked in LDS (read back with the first A fragments of every tile)
#endif
#ifndef S2_OPT_PARK1
#define S2_OPT_PARK1 5   // ... of the producer
#endif
#ifndef S2_OPT_DEPTH
#define S2_OPT_DEPTH 5
#endif
constexpr int S2_DEPTH = S2_OPT_DEPTH;
constexpr int S2_NBUF = 3;
constexpr int S2_XPAIR = 16 * 512;                     // one wave pair: two directions x [8 (query, k4)][2 g][64 lanes] dwords
constexpr int S2_X_OFF = 0;                            // [4 pairs] = 32 KiB
constexpr int S2_W_OFF = S2_X_OFF + 4 * S2_XPAIR;      // stage-2 weights: 64 lanes x 16 B
constexpr int S2_SEQ_OFF = S2_W_OFF + 1024;            // [4 pairs][2 directions] "half of tile t - 1 picked up" sequence numbers
constexpr int S2_EC_OFF = S2_SEQ_OFF + 64;             // [4 tiles of the block][64 lanes] {column mask lo, hi | e_bad << 31, sqrt a_e, sqrt n_e}
constexpr int S2_PARK_OFF = S2_EC_OFF + 4 * 1024;
constexpr int S2_PARK_BYTES = 4 * (S2_OPT_PARK0 + S2_OPT_PARK1) * 1024;
constexpr int S2_TILES_OFF = (S2_PARK_OFF + S2_PARK_BYTES + 255) / 256 * 256;
constexpr int S2_LDS_BYTES = S2_TILES_OFF + S2_NBUF * SP_PHASE_BYTES;
static_assert(S2_LDS_BYTES <= 160 * 1024, "LDS budget");
static_assert(S2_TILES_OFF >= SP_VS, "wrapped addresses stay non-negative");

template <int HALF>
struct S2Half {
  static constexpr int frag0 = HALF == 0 ? 0 : SP_DC_STEPS + 3 * SP_F_STEPS;                    // 0 / 36
  static constexpr int nfrag = HALF == 0 ? SP_DC_STEPS + 3 * SP_F_STEPS : 4 * SP_F_STEPS;       // 36 / 40
  static constexpr int npark = HALF == 0 ? S2_OPT_PARK0 : S2_OPT_PARK1;
  static constexpr int park_off = HALF == 0 ? 0 : 4 * S2_OPT_PARK0 * 1024;                       // inside the park area
  // slot t -> (frequency, K-step); local frequency lf = f - 4 HALF accumulates into acc[lf % 3]
  static constexpr int f_of(int t) { return HALF == 0 ? (t < SP_DC_STEPS ? 0 : 1 + (t - SP_DC_STEPS) / SP_F_STEPS) : 4 + t / SP_F_STEPS; }
  static constexpr int s_of(int t) { return HALF == 0 ? (t < SP_DC_STEPS ? t : (t - SP_DC_STEPS) % SP_F_STEPS) : t % SP_F_STEPS; }
  static constexpr int first_slot(int lf) { return HALF == 0 ? (lf == 0 ? 0 : SP_DC_STEPS + (lf - 1) * SP_F_STEPS) : lf * SP_F_STEPS; }
  static constexpr int last_slot(int lf) { return first_slot(lf + 1) - 1; }
  static constexpr int a_off(int t) { return f_of(t) == 0 ? 32 * s_of(t) : (f_of(t) - 1) * SP_F_BYTES + 32 * s_of(t); }
};

// LDS reads of stage 1 in issue order: A fragments 0 .. DEPTH-1, parked B fragments 0 .. NP-1, then after the MFMA of slot t
// the A fragment of slot t + DEPTH.  LDS returns in issue order, so slot t may start once at most
// (reads issued so far) - 1 - (position of the later of its two reads) are outstanding.
// s2_wait_of(now, t, ...): the count to wait for at the top of slot `now` so that the reads of slot t >= now have returned.
// One s_waitcnt serves TWO slots (the even one waits for its own and the next slot's fragments: the next one was issued
// S2_DEPTH - 1 slots ago, long enough)
constexpr int s2_wait_of(int now, int t, int nf, int np) {
  const int pos_a = t < S2_DEPTH ? t : S2_DEPTH + np + (t - S2_DEPTH);
  const int pos_b = t < np ? S2_DEPTH + t : -1;
  int issued = S2_DEPTH + np;            // prologue
  for (int u = 0; u < now; u++) issued += (u + S2_DEPTH < nf) ? 1 : 0;
  const int last = pos_a > pos_b ? pos_a : pos_b;
  return issued - 1 - last;
}
static_assert(S2_DEPTH >= 3, "a slot's fragment is requested at least two slots before the wait that covers it");

// per-lane A-fragment addressing of stage 1 (the only lane state that lives in registers across a tile)
struct S2Lane {
  unsigned dc_off, f_off, c_dc, c_f;  // as SpecLane
};

// every piece of a tile, over the 4 producer waves: piece c belongs to producer c % 4.  WHOLE pieces, no lane predicate:
// what a piece reads beyond the tile's last query is the next query's image or the allocation's slack (sp_nq4 rounds the
// batch up to whole tiles, spec_qimg_bytes adds 1 KiB) and lands in the unused end of the tile buffer
#ifndef S2_OPT_DMASPLIT
#define S2_OPT_DMASPLIT 0   // 1 (round-6 experiment): the late waves issue stream pieces 4..7 mod 8 of the next tile before their stage 1
#endif
// STREAM pieces of a tile for one issuer out of NI: pieces first, first + NI, ...
template <int NI>
__device__ __forceinline__ void dma_issue_stream(const TileDma &d, int first, int lane) {
  const int wu = __builtin_amdgcn_readfirstlane(first);
  unsigned lo = (unsigned)lane;
  asm volatile("" : "+v"(lo));
  const unsigned voff = lo * 16u;
  const char *gs = d.gsrc + wu * 1024;
  char *ls = d.ldst + wu * 1024;
#pragma unroll
  for (int j = 0; j < (SP_STREAM_PIECES + NI - 1) / NI; j++) {
    if (wu + NI * j < SP_STREAM_PIECES)
      __builtin_amdgcn_global_load_lds(reinterpret_cast<const AS1 void *>(reinterpret_cast<uintptr_t>(gs + j * NI * 1024) + voff),
                                       (AS3 void *)(ls + j * NI * 1024), 16, 0, 0);
  }
}
__device__ __forceinline__ void dma_issue_all(const TileDma &d, int sub, int lane) {
  const int wu = __builtin_amdgcn_readfirstlane(sub);
  unsigned lo = (unsigned)lane;
  asm volatile("" : "+v"(lo));
  const unsigned voff = lo * 16u;
  dma_issue_stream<S2_OPT_DMASPLIT ? 8 : 4>(d, sub, lane);
  if (d.nbytes_m) {  // the tile's flag word: a query with an empty column
    const int m0 = (wu - SP_STREAM_PIECES) & 3;  // first mask piece of this wave
    const char *gm = d.gsrc_m + m0 * 1024;
    char *lm = d.ldst + (SP_STREAM_PIECES + m0) * 1024;
#pragma unroll
    for (int j = 0; j < (SP_MASK_PIECES + 3) / 4; j++) {
      if (m0 + 4 * j < SP_MASK_PIECES)
        __builtin_amdgcn_global_load_lds(reinterpret_cast<const AS1 void *>(reinterpret_cast<uintptr_t>(gm + j * 4096) + voff),
                                         (AS3 void *)(lm + j * 4096), 16, 0, 0);
    }
  }
}

typedef unsigned u2v __attribute__((ext_vector_type(2)));
// two dwords 256 B x {o0, o1} from addr (the st64 forms count their offsets in units of 64 dwords)
__device__ __forceinline__ void lds_read2st64(u2v &dst, unsigned addr, int o0, int o1) {
  asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(dst) : "v"(addr), "n"(o0), "n"(o1));
}
__device__ __forceinline__ void lds_write2st64(unsigned addr, unsigned v0, unsigned v1, int o0, int o1) {
  asm volatile("ds_write2st64_b32 %0, %1, %2 offset0:%3 offset1:%4" ::"v"(addr), "v"(v0), "v"(v1), "n"(o0), "n"(o1) : "memory");
}
__device__ __forceinline__ void lds_write_b128(unsigned addr, frag4 v, int off) {
  asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(off) : "memory");
}
__device__ __forceinline__ void lds_write_b32(unsigned addr, unsigned v) {
  asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
// a dword every lane reads from the same LDS address, waited for on the spot
__device__ __forceinline__ unsigned lds_read_b32_now(unsigned addr) {
  unsigned v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
// a flag word through the scalar cache WITHOUT waiting: the caller reads `dst` only behind a later s_waitcnt lgkmcnt(0)
// (an SMEM load in flight makes the counted lgkmcnt waits of stage 1 stricter, never laxer)
__device__ __forceinline__ void scalar_load_u32_async(unsigned &dst, const unsigned *p) {
  asm volatile("s_load_dword %0, %1, 0x0" : "=s"(dst) : "s"(p) : "memory");
}

// one segment (one tile-block x a range of query tiles) for one wave of a pair.
//   HALF 0 ("late"):  f = 0..3, queries 2, 3 of a tile;  interval k = [stage 1 of tile k | tail of tile k - 1]
//   HALF 1 ("early"): f = 4..7, queries 0, 1;            interval k = [tail of tile k - 1 | DMA of tile k + 1 | stage 1 of tile k]
template <int HALF>
__device__ __forceinline__ void spec2_segment(const SpecArgs &a, char *smem, unsigned lds_base, const S2Lane &ln, int wave, int lane,
                                              int64_t tb, int t0, int t1) {
  using H = S2Half<HALF>;
  constexpr int NF = H::nfrag, NP = H::npark;
  constexpr int QB = HALF == 1 ? 0 : 2;   // this wave's two queries of a tile; (query, k4) = j in [4 QB, 4 QB + 8)
  constexpr int JB = 4 * QB, JO = 8 - JB; // ... and the partner's
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  typedef unsigned u8v __attribute__((ext_vector_type(8)));
  const int sub = wave & 3;
  const int64_t ntiles = (a.n_items + 31) >> 5;
  const int q0 = t0 * SP_QPT;
  const int q1 = (t1 * SP_QPT < a.nq) ? t1 * SP_QPT : a.nq;
  const int64_t tile = tb * 4 + sub;
  const bool tile_ok = tile < ntiles;  // wave-uniform, the same for both waves of a pair
  const int nphase = (q1 - q0 + SP_QPP - 1) / SP_QPP;
  const unsigned lane16 = (unsigned)lane * 16u;

  const unsigned *qflags = reinterpret_cast<const unsigned *>(a.qimg + sp_flags_at(a.nq)) + __builtin_amdgcn_readfirstlane(t0);
  auto tile_dma = [&](int p, unsigned flagword) {
    const int qn = q0 + p * SP_QPP;
    const int nqs = (q1 - qn < SP_QPP) ? (q1 - qn) : SP_QPP;
    return TileDma{a.qimg + (int64_t)qn * SP_QS, a.qimg + sp_masks_at(a.nq) + (int64_t)qn * SP_MASK_BYTES,
                   smem + S2_TILES_OFF + (p % S2_NBUF) * SP_PHASE_BYTES, nqs * SP_QS, flagword ? nqs * SP_MASK_BYTES : 0};
  };
  if (HALF == 1) dma_issue_all(tile_dma(0, scalar_load_u32(qflags)), sub, lane);  // tile 0: overlaps the B loads below
  else if (S2_OPT_DMASPLIT) dma_issue_stream<8>(tile_dma(0, 0u), sub + 4, lane);

  half8 B[NF];
  {
    const uint4 *src = a.spT + ((tile_ok ? tile : 0) * SP_FRAGS + H::frag0) * 64 + lane;
#pragma unroll
    for (int s = 0; s < NF; s++) {
      const uint4 v = src[s * 64];
      B[s] = *reinterpret_cast<const half8 *>(&v);
    }
    // no AGPR constraint here: a function that never names an AGPR gets its whole budget (256 at two waves per SIMD) as
    // VGPRs with -amdgpu-mfma-vgpr-form; naming one makes the compiler split the file 128 / 128
#pragma unroll
    for (int s = NP; s < NF; s++) asm volatile("" : "+v"(B[s]));
#pragma unroll
    for (int s = 0; s < NP; s++)
      lds_write_b128(lds_base + (unsigned)(S2_PARK_OFF + H::park_off) + (unsigned)(sub * (NP * 1024)) + lane16, __builtin_bit_cast(frag4, B[s]), s * 1024);
  }
  if (HALF == 0) {  // the entry constants of the tile and the pair's sequence numbers
    const int col = lane & 31;
    const int64_t n = tile * 32 + col;
    const bool n_ok = tile_ok && n < a.n_items;
    const u64 m2 = n_ok ? a.cmask[n] : 0ull;
    const int n_e = __popcll(m2 & kMask60);
    frag4 ec;
    ec[0] = (unsigned)(m2 & 0xffffffffull);
    ec[1] = (unsigned)((m2 & kMask60) >> 32) | (((m2 & kNonFinite) != 0) ? 0x80000000u : 0u);
    ec[2] = __float_as_uint(n_ok ? a.aux[n] : 0.0f);
    ec[3] = __float_as_uint(sqrtf((float)n_e));
    lds_write_b128(lds_base + (unsigned)S2_EC_OFF + (unsigned)(sub * 1024) + lane16, ec, 0);
    lds_write_b32(lds_base + (unsigned)S2_SEQ_OFF + (unsigned)(sub * 8), 0u);
    lds_write_b32(lds_base + (unsigned)S2_SEQ_OFF + (unsigned)(sub * 8 + 4), 0u);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  const unsigned bpark = lds_base + (unsigned)(S2_PARK_OFF + H::park_off) + (unsigned)(sub * (NP * 1024)) + lane16;
  // exchange area of the pair: [0, 4 KiB) what the early wave leaves for the late one, [4 KiB, 8 KiB) the other direction;
  // sequence word s: the half with direction s of tile t - 1 has been picked up (value t)
  const unsigned x_pair = lds_base + (unsigned)S2_X_OFF + (unsigned)(sub * S2_XPAIR);
  const unsigned seq_mine = lds_base + (unsigned)S2_SEQ_OFF + (unsigned)(sub * 8 + 4 * (1 - HALF));   // bumped by my partner, checked by me
  const unsigned seq_theirs = lds_base + (unsigned)S2_SEQ_OFF + (unsigned)(sub * 8 + 4 * HALF);      // bumped by me
  // RSX_SPEC_INSTRUMENT builds: s_memtime sums per region of an interval, for waves 0 and 4 of workgroup 0
  const bool prof = kInstr && a.prof && blockIdx.x == 0 && sub == 0;
  unsigned long long ps[5] = {0, 0, 0, 0, 0}, pt = 0;
  auto lap = [&](int region) {
    if (kInstr && prof) {
      const unsigned long long now = prof_now();
      ps[region] += now - pt;
      pt = now;
    }
  };
  unsigned long long c_begin = 0, r_begin = 0;
  if (kInstr && prof) {
    c_begin = prof_now();
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r_begin)::"memory");
  }
  unsigned flag_next = (HALF == 1 && nphase > 1) ? scalar_load_u32(qflags + 1) : 0u;  // early wave: flag word of the next tile to stage
  unsigned held[2][8];  // this wave's own packed halves {C_(2g), C_(2g+1)} of ITS two queries of the previous tile
#pragma unroll
  for (int i = 0; i < 8; i++) held[0][i] = held[1][i] = 0u;

  // ---------------- stage 1 of tile p: this wave's frequencies, all 4 queries -> own[gl][j] ----------------
  auto stage1 = [&](int p, unsigned (&own)[2][16]) {
    const unsigned tile_lds = lds_base + (unsigned)(S2_TILES_OFF + (p % S2_NBUF) * SP_PHASE_BYTES);
    const int hh = lane >> 5;
    floatx16 acc[3];
    floatx16 z;
#pragma unroll
    for (int i = 0; i < 16; i++) z[i] = 0.0f;
    frag4 ring[S2_DEPTH];
    frag4 bx[NP > 0 ? NP : 1];
    const unsigned a_dc = tile_lds + ln.dc_off, a_f = tile_lds + ln.f_off;
    const unsigned a_dcw = a_dc - SP_DC_BYTES, a_fw = a_f - SP_VS;
    auto a_read = [&](frag4 &dst, auto tc) {
      constexpr int t = decltype(tc)::value;
      if constexpr (H::f_of(t) == 0) lds_read_stream<H::s_of(t), SP_DC_BYTES / 16, 10>(dst, a_dc, a_dcw, ln.c_dc, H::a_off(t));
      else lds_read_stream<H::s_of(t), SP_VS / 16, 16>(dst, a_f, a_fw, ln.c_f, H::a_off(t));
    };
    static_for<S2_DEPTH>([&](auto tc) { a_read(ring[decltype(tc)::value], tc); });
    static_for<NP>([&](auto tc) {
      constexpr int t = decltype(tc)::value;
      lds_read_frag(bx[t], bpark, 1024 * t);
    });
    static_for<NF>([&](auto tc) {
      constexpr int t = decltype(tc)::value;
      constexpr int f = H::f_of(t), lf = f - 4 * HALF;
      if constexpr (t % 2 == 0) lds_wait_count<s2_wait_of(t, t + 1 < NF ? t + 1 : t, NF, NP)>();
      __builtin_amdgcn_sched_barrier(0);
      const half8 af = __builtin_bit_cast(half8, ring[t % S2_DEPTH]);
      half8 bf;
      if constexpr (t < NP) bf = __builtin_bit_cast(half8, bx[t]);
      else bf = B[t];
      if constexpr (H::s_of(t) == 0) acc[lf % 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf, z, 0, 0, 0);
      else acc[lf % 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf, acc[lf % 3], 0, 0, 0);
      if constexpr (t + S2_DEPTH < NF) a_read(ring[t % S2_DEPTH], std::integral_constant<int, t + S2_DEPTH>{});
      if constexpr (HALF == 0) {
        // C_0 as fp16 hi + lo (see spec_tile): during f = 1, two j per slot
        constexpr int sb = H::last_slot(0) + 2;
        if constexpr (t >= sb && t < sb + 8) {
#pragma unroll
          for (int j = 2 * (t - sb); j < 2 * (t - sb) + 2; j++) {
            const float v = acc[0][j];
            const float hi = __uint_as_float(__float_as_uint(v) & 0xffffe000u);
            float sp = hh ? v - hi : hi;
            asm volatile("" : "+v"(sp));
            acc[0][j] = sp;
          }
        }
      }
      // first pair (lf 0, 1) -> own[0][*], packed during lf 2 (two j per slot), before lf 3 re-uses acc[0]
      {
        constexpr int pb = H::last_slot(1) + 2;
        static_assert(pb + 8 <= H::first_slot(3), "accumulators are re-used before they are packed");
        if constexpr (t >= pb && t < pb + 8) {
#pragma unroll
          for (int j = 2 * (t - pb); j < 2 * (t - pb) + 2; j++) {
            unsigned pk = pack2(acc[0][j], acc[1][j]);
            asm volatile("" : "+v"(pk));
            own[0][j] = pk;
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    // second pair (lf 2, 3) = acc[2], acc[0]
#pragma unroll
    for (int j = 0; j < 16; j++) own[1][j] = pack2(acc[2][j], acc[0][j]);
  };

  // ---------------- hand the partner's two queries of tile p over ----------------
  auto hand_over = [&](int p, const unsigned (&own)[2][16]) {
    unsigned lw = (unsigned)lane;
    asm volatile("" : "+v"(lw));
    // my partner has picked up what I left for tile p - 1 (it reads long before I write: no wait in practice)
    if (!(kInstr && (a.dbg & 12)))   // (timing experiments that idle one wave of the pair: nobody bumps the sequence numbers)
      while ((int)__builtin_amdgcn_readfirstlane(lds_read_b32_now(seq_mine)) < p) __builtin_amdgcn_s_sleep(1);
    const unsigned x_wr = x_pair + (unsigned)(HALF == 1 ? 0 : S2_XPAIR / 2) + lw * 4u;
#pragma unroll
    for (int jl = 0; jl < 8; jl++) lds_write2st64(x_wr, own[0][JO + jl], own[1][JO + jl], 2 * jl, 2 * jl + 1);
  };

  // ---------------- stage 2 + bounds + store of this wave's two queries of tile p ----------------
  auto tail = [&](int p) {
    const unsigned tile_lds = lds_base + (unsigned)(S2_TILES_OFF + (p % S2_NBUF) * SP_PHASE_BYTES);
    // every lane-dependent address of the tail derives from an opaque copy of the lane id: left visible, the compiler
    // hoists a dozen of them out of the tile loop and keeps them in registers across stage 1 (spilling B fragments)
    unsigned lt = (unsigned)lane;
    asm volatile("" : "+v"(lt));
    const unsigned l16 = lt * 16u;
    const int col = (int)(lt & 31u), hh = (int)(lt >> 5);
    const unsigned a_tl = tile_lds + SP_TAIL;
    const unsigned x_rd = x_pair + (unsigned)(HALF == 1 ? S2_XPAIR / 2 : 0) + lt * 4u;
    const int qp = q0 + p * SP_QPP;
    const int nq_here = (q1 - qp < SP_QPP) ? (q1 - qp) : SP_QPP;
    frag4 wv, ec, tw[2];
    u2v rx[8];
    static_for<8>([&](auto jc) {
      constexpr int jl = decltype(jc)::value;
      lds_read2st64(rx[jl], x_rd, 2 * jl, 2 * jl + 1);
    });
    lds_read_frag(wv, lds_base + (unsigned)S2_W_OFF + l16, 0);
    lds_read_frag(ec, lds_base + (unsigned)S2_EC_OFF + (unsigned)(sub * 1024) + l16, 0);
    lds_read_frag(tw[0], a_tl, QB * SP_QS);         // {n_q, flags, sqrt n_q, sqrt a_q} of the two queries
    lds_read_frag(tw[1], a_tl, (QB + 1) * SP_QS);
    lds_wait_count<0>();
    __builtin_amdgcn_sched_barrier(0);
    lds_write_b32(seq_theirs, (unsigned)(p + 1));  // my partner's half of tile p is in registers: it may overwrite the area
    lap(HALF == 1 ? 0 : 2);  // exchange + constant reads
    floatx16 z;
#pragma unroll
    for (int i = 0; i < 16; i++) z[i] = 0.0f;
    const half8 W = __builtin_bit_cast(half8, wv);
    const int n_e = __popc(ec[0]) + __popc(ec[1] & 0x0fffffffu);
    const bool e_bad = (ec[1] & 0x80000000u) != 0;
    const float sqrt_ne = __uint_as_float(ec[3]), sqrt_ae = __uint_as_float(ec[2]);
    const float r_ne = __builtin_amdgcn_rcpf((float)(n_e > 1 ? n_e : 1));
    auto Pq = [&](int ql, int k4) {  // stage-2 B operand of (query QB + ql, k4): K = {g0, g1, g2, g3}
      const int jl = 4 * ql + k4;
      u4 t;
      if constexpr (HALF == 0) t = u4{held[0][jl], held[1][jl], rx[jl][0], rx[jl][1]};
      else t = u4{rx[jl][0], rx[jl][1], held[0][jl], held[1][jl]};
      return __builtin_bit_cast(half8, t);
    };
    auto bound_of = [&](float m_scaled, float rL, const Recip &r) {  // m_scaled = 15/16 max_k S_k u(n_k)
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(m_scaled), __float_as_uint(m_scaled), false, false);
      const float best = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
      const float err = kE1 * r.sqrt_nq * sqrt_ne + kE2 * r.sqrt_aq * sqrt_ae;
      float v = (1.0f + a.eps_direct) - fmaf(best, (16.0f / 15.0f) * (1.0f + 4e-6f), err * rL);
      if (r.n_q == 0 || n_e == 0) v = INFINITY;      // no effective column at any shift: never a hit
      if (r.flags != 0u || e_bad) v = -INFINITY;     // non-finite input: always re-score exactly
      return v;
    };
    float outv[2];
    const Recip ra = recip_header(tw[0]), rb = recip_header(tw[1]);
    const bool both_full = __builtin_amdgcn_readfirstlane(ra.n_q) == NS && __builtin_amdgcn_readfirstlane(rb.n_q) == NS;
    if (both_full) {
      // The usual case (a radar scan has no empty sector): n_eff(k) = n_e at every shift, so a query is 4 stage-2 MFMAs and
      // 16 v_max3.  The 8 MFMAs of the pair go through THREE result sets, each consumed two MFMAs after it was issued
      auto max8 = [&](float m, const floatx16 &dd) {
#pragma unroll
        for (int e = 0; e < 4; e++) m = fmaxf(fmaxf(m, dd[2 * e]), dd[2 * e + 1]);  // one v_max3_f32 per pair
        return m;
      };
      float ma = 0.0f, mb = 0.0f;
      floatx16 d0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(W, Pq(0, 0), z, 0, 0, 0);
      floatx16 d1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(W, Pq(0, 1), z, 0, 0, 0);
      floatx16 d2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(W, Pq(0, 2), z, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      ma = max8(ma, d0);
      d0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(W, Pq(0, 3), z, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      ma = max8(ma, d1);
      d1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(W, Pq(1, 0), z, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      ma = max8(ma, d2);
      d2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(W, Pq(1, 1), z, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      ma = max8(ma, d0);
      d0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(W, Pq(1, 2), z, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      mb = max8(mb, d1);
      d1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(W, Pq(1, 3), z, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      mb = max8(mb, d2);
      __builtin_amdgcn_sched_barrier(0);
      mb = max8(mb, d0);
      mb = max8(mb, d1);
      // ONE bound per lane (round 5).  Lanes 0..31 store query QB, lanes 32..63 query QB + 1, and every lane holds the maxima
      // of its half of the rows for BOTH queries: v_permlane32_swap of (ma, mb) hands the lower half query A's two partial
      // maxima and the upper half query B's, so one swap + one maximum replace two, and the bound arithmetic runs once per
      // lane with its own query's constants instead of twice with half of it thrown away.  (n_q = 60 for both: sqrt n_q is
      // the same number; n_lo = n_hi = n_e: 1 / n_lo within 1 ulp of r_ne, covered by the (1 + 4e-6).)
      {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(ma * r_ne), __float_as_uint(mb * r_ne), false, false);
        const float best = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        const float sqrt_aq = hh ? rb.sqrt_aq : ra.sqrt_aq;
        const unsigned flags = hh ? rb.flags : ra.flags;
        const float err = kE1 * ra.sqrt_nq * sqrt_ne + kE2 * sqrt_aq * sqrt_ae;
        float v = (1.0f + a.eps_direct) - fmaf(best, (16.0f / 15.0f) * (1.0f + 4e-6f), err * r_ne);
        if (n_e == 0) v = INFINITY;                 // no effective column at any shift: never a hit
        if (flags != 0u || e_bad) v = -INFINITY;    // non-finite input: always re-score exactly
        outv[0] = outv[1] = v;
      }
    } else {
      static_for<2>([&](auto qc) {
        constexpr int ql = decltype(qc)::value;
        Recip r = recip_header(tw[ql]);
        const bool full_q = __builtin_amdgcn_readfirstlane(r.n_q) == NS;
        float m = 0.0f;
        if (full_q) {
#pragma unroll
          for (int k4 = 0; k4 < 4; k4++) {
            const floatx16 dd = __builtin_amdgcn_mfma_f32_32x32x16_f16(W, Pq(ql, k4), z, 0, 0, 0);
#pragma unroll
            for (int e = 0; e < 4; e++) m = fmaxf(fmaxf(m, dd[2 * e]), dd[2 * e + 1]);
          }
          r.rL = r_ne;
          m *= r.rL;
        } else {
          recip_coeffs(r, n_e);
          // this lane's entry: column-mask bits 32 hh .. 32 hh + 31 (the K index of the lane half) as fp8 0.0 / 1.0 bytes;
          // the query's mask rows come from the tile buffer; one M-tile (k4 = 2 mt, 2 mt + 1) at a time: n_eff, u(n_eff), S u, maximum
          const unsigned bits = hh ? (ec[1] & 0x0fffffffu) : ec[0];
          u8v bm;
#pragma unroll
          for (int rr = 0; rr < 8; rr++) bm[rr] = ((((bits >> (4 * rr)) & 0xfu) * 0x00204081u) & 0x01010101u) * 0x38u;
#pragma unroll
          for (int mt = 0; mt < 2; mt++) {
            // n_eff rows: row = col <-> (k4 = 2 mt + col / 16, k15 = col % 16), the row order of the stage-2 output
            const int k4 = 2 * mt + (col >> 4), k15 = (col & 15) == 15 ? 0 : (col & 15);
            const int kk0 = (45 * k4 + 16 * k15) % NS;  // CRT
            const unsigned a_m = tile_lds + (unsigned)(SP_MASKREG_OFF + (kk0 & 15) * SP_MASK_COPY + (kk0 & ~15) + hh * 32);
            frag4 mk0, mk1;
            lds_read_frag(mk0, a_m, (QB + ql) * SP_MASK_BYTES);
            lds_read_frag(mk1, a_m, (QB + ql) * SP_MASK_BYTES + 16);
            lds_wait_count<0>();
            __builtin_amdgcn_sched_barrier(0);
            const u8v am = {mk0[0], mk0[1], mk0[2], mk0[3], mk1[0], mk1[1], mk1[2], mk1[3]};
            const floatx16 nacc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(__builtin_bit_cast(intx8, am), __builtin_bit_cast(intx8, bm), z, 0, 0, 0, 0, 0, 0);
#pragma unroll
            for (int kk = 0; kk < 2; kk++) {
              float2v u2[4];
#pragma unroll
              for (int e = 0; e < 4; e++) {
                const float2v n2 = {nacc[kk * 8 + 2 * e], nacc[kk * 8 + 2 * e + 1]};
                const float2v t2 = __builtin_elementwise_fma(n2, r.C2, r.B2);
                u2[e] = __builtin_elementwise_fma(n2, t2, r.A2);
              }
              const floatx16 dd = __builtin_amdgcn_mfma_f32_32x32x16_f16(W, Pq(ql, 2 * mt + kk), z, 0, 0, 0);
#pragma unroll
              for (int e = 0; e < 4; e++) {
                const float2v s2 = {dd[2 * e], dd[2 * e + 1]};
                const float2v v2 = s2 * u2[e];
                m = fmaxf(fmaxf(m, v2[0]), v2[1]);
              }
            }
          }
        }
        outv[ql] = bound_of(m, r.rL, r);
      });
    }
    // lanes 0..31 store query QB, lanes 32..63 query QB + 1: one store per wave and tile
    {
      const int64_t n = tile * 32 + col;
      const int qq = QB + hh;
      const float vv = hh ? outv[1] : outv[0];
      if (n < a.n_items && qq < nq_here) a.lb[(int64_t)(qp + qq) * a.ld_lb + n] = lb_pack(vv);
    }
    lap(HALF == 1 ? 1 : 3);  // stage 2 + bounds + store
  };

  // interval k: stage 1 of query tile k, tails of tile k - 1
  for (int k = 0; k <= nphase; k++) {
    if (kInstr && prof) pt = prof_now();
    // RSX_SPEC_DBG bits 4 / 8 of instrumented builds idle the early / the late wave (barriers and DMA only): what each wave
    // costs when it has the SIMD to itself (results are wrong, only the timing means something)
    const bool idle = kInstr && (a.dbg & (HALF == 1 ? 4 : 8));
    const bool do_s1 = tile_ok && k < nphase && !idle, do_tail = tile_ok && k >= 1 && !idle;
    unsigned own[2][16];  // own[gl][j]: packed {C_(2g), C_(2g+1)} of (query j / 4, k4 = j % 4), g = 2 HALF + gl
    if (HALF == 1) {
      if (do_tail) tail(k - 1);
      // tile k + 1 goes into the buffer tile k - 2 was read from (every wave is past the barrier of interval k - 1; this
      // wave's store of the tail above is older, so the vmcnt(0) at the end of the interval waits for nothing young).
      // Its flag word was requested an interval ago; the one of tile k + 2 is requested now and is DEFINED only behind the
      // lgkmcnt(0) after stage 1 ("+s" there: no use of it can be scheduled above that wait)
      if (k + 1 < nphase) dma_issue_all(tile_dma(k + 1, flag_next), sub, lane);
      if (k + 2 < nphase) scalar_load_u32_async(flag_next, qflags + k + 2);
      lap(2);  // DMA issue
      if (do_s1) {
        stage1(k, own);
        lap(3);  // stage 1 incl. the last packing
        hand_over(k, own);
#pragma unroll
        for (int i = 0; i < 8; i++) held[0][i] = own[0][JB + i], held[1][i] = own[1][JB + i];
      }
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+s"(flag_next)::"memory");
    } else {
      if (S2_OPT_DMASPLIT && k + 1 < nphase) dma_issue_stream<8>(tile_dma(k + 1, 0u), sub + 4, lane);  // stream pieces 4 + sub, 12 + sub
      if (do_s1) {
        stage1(k, own);
        lap(0);  // stage 1 incl. the last packing
        hand_over(k, own);
        lap(1);  // sequence check + exchange writes
      }
      if (do_tail) tail(k - 1);
      if (do_s1) {
#pragma unroll
        for (int i = 0; i < 8; i++) held[0][i] = own[0][JB + i], held[1][i] = own[1][JB + i];
      }
      // this wave issues no DMA; its stores stay in flight (nothing waits for them before the end of the segment)
      if (S2_OPT_DMASPLIT) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // (its pieces of tile k + 1 -- and the tail's store)
      else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    lap(4);  // exchange writes / end-of-interval wait
    __builtin_amdgcn_s_barrier();
  }
  if (kInstr && prof) {
    unsigned long long r_end;
    const unsigned long long c_end = prof_now();
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r_end)::"memory");
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < 5; i++) a.prof[HALF * 8 + i] = ps[i];
      a.prof[HALF * 8 + 5] = (unsigned long long)nphase;
      a.prof[HALF * 8 + 6] = c_end - c_begin;
      a.prof[HALF * 8 + 7] = r_end - r_begin;
    }
  }
  // segment end: the next segment re-writes the parked fragments, the entry constants and the tile buffers
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}

__global__ __launch_bounds__(512, 2) void sc_spec2_filter_kernel(SpecArgs a) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int half = wave >> 2;
  const int col = lane & 31, hh = lane >> 5;
  const int64_t ntiles = (a.n_items + 31) >> 5;
  const int64_t ntb = (ntiles + 3) >> 2;
  const int nqt = (a.nq + SP_QPT - 1) / SP_QPT;
  const int64_t total = a.tb_cum ? a.tb_cum[ntb] : ntb * (int64_t)nqt;
  // Work split.  With a plan (or a batch too small to give every XCD its own queries): equal contiguous segments of the
  // (tile-block, query tile) list, one persistent workgroup per CU.  Otherwise by XCD: workgroup b runs on XCD b % 8 (the
  // dispatcher deals workgroups round-robin over the XCDs) as the j = b / 8-th of that XCD's sequence; the XCD owns query
  // tiles [x * xcd_nqt, (x + 1) * xcd_nqt) in sub-ranges of xcd_len tiles, and workgroup j = (sub-range j / ntb, tile-block
  // j % ntb).  The 32 workgroups resident on an XCD (one per CU) then start on the SAME xcd_len query tiles together and
  // stay within a few tiles of each other: a query image comes out of HBM / Infinity Cache once per launch and out of the
  // XCD's L2 for the other 31.  With contiguous segments the 32 streamed 32 different ranges: 1.58 GB of L2 -> fabric
  // reads per launch for 78 MB of images (profiles/r04_sc_spec_v11).  The price: a 311 KB tile-block load per workgroup
  // (2 x 78 per XCD instead of 78 + 32) and whole rounds of 32 (4.875 -> 5 on the bench).  A persistent variant (32
  // workgroups per XCD walking a (chunk, tile-block, tile) list cut evenly) was measured too: its workgroups sit at 32
  // different tiles of the chunk, the 3.3 MB chunk does not survive in the 4 MB L2 next to the bound stores, and the
  // traffic went UP to 2.05 GB (1.93-1.96 ms against 1.84-1.86 the same run)
  int64_t L0, L1;
  if (a.xcd_len > 0) {
    const int x = blockIdx.x & 7;
    const int64_t j = blockIdx.x >> 3, sub = j / ntb, utb = j - sub * ntb;
    const int64_t qx1 = ((int64_t)(x + 1) * a.xcd_nqt < nqt) ? (int64_t)(x + 1) * a.xcd_nqt : nqt;
    const int64_t u0 = (int64_t)x * a.xcd_nqt + sub * a.xcd_len, u1 = (u0 + a.xcd_len < qx1) ? (u0 + a.xcd_len) : qx1;
    L0 = utb * nqt + u0;
    L1 = u0 < u1 ? utb * nqt + u1 : L0;
  } else {
    const int64_t per = (total + gridDim.x - 1) / gridDim.x;
    L0 = (int64_t)blockIdx.x * per;
    L1 = (L0 + per < total) ? (L0 + per) : total;
  }
  const unsigned lds_base = (unsigned)(uintptr_t)((AS3 char *)smem);
  S2Lane ln;
  {
    // A-fragment address of this lane inside a tile of 4 query images: row = col = 8 * query + 4 * variant + k4
    const int rq = col >> 3, rv = (col >> 2) & 1, rk = col & 3;
    ln.c_dc = rk * 3 + hh;
    ln.c_f = rk * 5 + hh;
    ln.dc_off = rq * SP_QS + ln.c_dc * 16;
    ln.f_off = rq * SP_QS + SP_DC_BYTES + rv * SP_VS + ln.c_f * 16;
  }
  if (wave == 0) {
    // stage-2 A operand (inverse DFT weights, see sc_spec_filter_kernel), once per workgroup, into LDS
    frag4 wv;
    half8 W;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      float w = 0.0f;
      if (col < 15) {
        if (i == 0) w = 1.0f / 16.0f;
        else {
          const int t = (i * col) % 15;
          w = hh ? (float)(-sinpi(2.0 * t / 15.0) / 8.0) : (float)(cospi(2.0 * t / 15.0) / 8.0);
        }
      }
      W[i] = (_Float16)w;
    }
    wv = __builtin_bit_cast(frag4, W);
    lds_write_b128(lds_base + (unsigned)S2_W_OFF + (unsigned)lane * 16u, wv, 0);
  }
  // no static priority: with s_setprio 1 on the consumer (the critical path of an interval) the producer made NO progress
  // while the consumer sat in its tail -- an s_nop of the prioritised wave wins the issue slot too (measured: 200 cycles
  // per producer slot there).  RSX_SPEC_DBG bits 1 / 2 of instrumented builds give the consumer / the producer priority
  if (kInstr && (a.dbg & 1)) {
    if (half == 0) __builtin_amdgcn_s_setprio(1);
  } else if (kInstr && (a.dbg & 2)) {
    if (half == 1) __builtin_amdgcn_s_setprio(1);
  }

  int64_t tb = 0;
  if (a.tb_cum && L0 < L1) {
    int64_t lo = 0, hi = ntb - 1;
    while (lo < hi) {
      const int64_t mid = (lo + hi + 1) >> 1;
      if (a.tb_cum[mid] <= L0) lo = mid;
      else hi = mid - 1;
    }
    tb = lo;
  }
  while (L0 < L1) {
    int t0;
    if (a.tb_cum) {
      while (a.tb_cum[tb + 1] <= L0) tb++;
      t0 = a.tb_qmin[tb] + (int)(L0 - a.tb_cum[tb]);
    } else {
      tb = L0 / nqt;
      t0 = (int)(L0 - tb * nqt);
    }
    const int t1 = (L1 - L0 < (int64_t)(nqt - t0)) ? (int)(t0 + (L1 - L0)) : nqt;
    L0 += t1 - t0;
    if (half == 0) spec2_segment<0>(a, smem, lds_base, ln, wave, lane, tb, t0, t1);
    else spec2_segment<1>(a, smem, lds_base, ln, wave, lane, tb, t0, t1);
  }
}

}  // namespace

size_t spec_qimg_bytes(int32_t nq) { return (size_t)(sp_flags_at(nq) + sp_nq4(nq)) + 1024; }  // + slack: the filter reads flag words up to 3 tiles ahead

int launch_spec_db_images(const float *desc, const double *norm, int64_t first, int64_t count, void *spT, float *aux,
                          hipStream_t s) {
  if (count <= 0) return RSX_OK;
  hipLaunchKernelGGL(sc_spec_db_kernel, dim3((unsigned)((count + 3) / 4)), dim3(256), 0, s, desc, norm, first, count,
                     static_cast<uint4 *>(spT), aux);
  RSX_HIP(hipGetLastError());
  return RSX_OK;
}

int launch_spec_query_images(const float *desc, const double *norm, int32_t nq, void *qimg, hipStream_t s) {
  if (nq <= 0) return RSX_OK;
  hipLaunchKernelGGL(sc_spec_query_kernel, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, s, desc, norm, nq,
                     static_cast<char *>(qimg));
  RSX_HIP(hipGetLastError());
  return RSX_OK;
}

int launch_insert(const void *d_pts, const int64_t *d_offs, int64_t n_pts, int64_t n_clouds, int64_t stride_bytes,
                  double lidar_height, double max_radius, int64_t first_slot, float *desc, double *vkey, double *norm,
                  float *rkey, void *hnT, void *hnR, uint64_t *cmask, void *spT, float *aux, void *vk16, float *vk_n,
                  hipStream_t s, int sum_order) {
  if (n_clouds <= 0) return RSX_OK;
  InsertArgs a;
  a.pts = static_cast<const char *>(d_pts);
  a.offs = d_offs;
  a.n_pts = n_pts;
  a.stride = stride_bytes;
  a.lidar_height = lidar_height;
  a.max_radius = max_radius;
  a.first = first_slot;
  a.desc = desc;
  a.vkey = vkey;
  a.norm = norm;
  a.rkey = rkey;
  a.hnT = static_cast<uint4 *>(hnT);
  a.hnR = static_cast<uint4 *>(hnR);
  a.cmask = reinterpret_cast<u64 *>(cmask);
  a.spT = static_cast<uint4 *>(spT);
  a.aux = aux;
  a.vk16 = static_cast<_Float16 *>(vk16);
  a.vk_n = vk_n;
  RSX_SO_DISPATCH(sum_order, hipLaunchKernelGGL(sc_insert_kernel<SO>, dim3((unsigned)n_clouds), dim3(256), 0, s, a));
  RSX_HIP(hipGetLastError());
  return RSX_OK;
}

const char *spec_filter_kernel_name() { return "sc_spec_filter_kernel"; }

const char *spec2_filter_kernel_name() { return "sc_spec2_filter_kernel"; }

int launch_spec_filter(const DbView &db, const void *qimg, int32_t nq, int64_t n_items, lb_t *lb, int64_t ld_lb,
                       const int32_t *tb_qmin, const int64_t *tb_cum, hipStream_t s, bool two_waves) {
  if (nq <= 0 || n_items <= 0) return RSX_OK;
  static int n_cu = 0;
  const int lds = SP_LDS_BYTES;
  if (!n_cu) {
    RSX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&sc_spec_filter_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    int dev = 0, cu = 0;
    RSX_HIP(hipGetDevice(&dev));
    RSX_HIP(hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev));
    n_cu = cu > 0 ? cu : 256;
  }
  SpecArgs a;
  a.spT = static_cast<const uint4 *>(db.spT);
  a.cmask = reinterpret_cast<const u64 *>(db.cmask);
  a.aux = db.sp_aux;
  a.qimg = static_cast<const char *>(qimg);
  a.n_items = n_items;
  a.nq = nq;
  a.lb = lb;
  a.ld_lb = ld_lb;
  a.eps_direct = (float)filter_eps();
  {
    const char *e = rsx::exp_env("RSX_SPEC_DBG");
    a.dbg = e ? atoi(e) : 0;
  }
  const int64_t ntiles = (n_items + 31) / 32;
  const int64_t nqt = (nq + SP_QPT - 1) / SP_QPT;
  const int64_t ntb = (ntiles + 3) / 4;
  a.tb_qmin = tb_qmin;
  a.tb_cum = tb_cum;
  unsigned grid = (unsigned)n_cu;
  a.unit_len = (int32_t)nqt;
  if (!SP_OPT_LOCKSTEP) {
    int64_t per = (ntb * nqt + n_cu - 1) / n_cu;
    if (per < 4) per = 4;
    grid = (unsigned)((ntb * nqt + per - 1) / per);
  } else if (!tb_cum) {
    // number of query ranges R: fill whole rounds of n_cu workgroups as evenly as possible; a unit keeps
    // >= 16 query tiles so that the 304-register B load stays amortised
    int64_t best_r = 1;
    double best_eff = 0.0;
    for (int64_t r = 1; r <= 64 && (r == 1 || (nqt + r - 1) / r >= 16); r++) {
      const int64_t len = (nqt + r - 1) / r, rr = (nqt + len - 1) / len;  // rr ranges actually used
      const int64_t units = rr * ntb, rounds = (units + n_cu - 1) / n_cu;
      const double eff = (double)(ntb * nqt) / (double)(rounds * n_cu * len);
      if (eff > best_eff + 1e-9) {
        best_eff = eff;
        best_r = rr;
      }
    }
    if (const char *e = rsx::exp_env("RSX_SPEC_R")) {  // experiments: force the number of query ranges
      const int64_t v = atoll(e);
      if (v >= 1 && v <= nqt) best_r = v;
    }
    a.unit_len = (int32_t)((nqt + best_r - 1) / best_r);
    const int64_t rr = (nqt + a.unit_len - 1) / a.unit_len;
    grid = (unsigned)(rr * ntb);
  }
  if (two_waves) {
    // sc_spec2_filter_kernel: 512 threads, the entry tile split by frequency over two waves per SIMD
    static bool attr2 = false;
    if (!attr2) {
      RSX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&sc_spec2_filter_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, S2_LDS_BYTES));
      attr2 = true;
    }
    static unsigned long long *d_prof2 = nullptr;
    static const bool want_prof2 = kInstr && rsx::exp_env("RSX_SPEC_PROF") != nullptr;
    if (want_prof2 && !d_prof2) RSX_HIP(hipMalloc(&d_prof2, 16 * sizeof(unsigned long long)));
    a.prof = want_prof2 ? d_prof2 : nullptr;
    a.xcd_nqt = a.xcd_len = 0;
    {
      // XCD-aware split (no plan, every XCD gets >= 16 query tiles of its own).  Sub-ranges per XCD S: whole rounds of the
      // XCD's workgroup slots (one per CU) filled best, cost = rounds x (tiles per sub-range + 3): a workgroup's start
      // (311 KB tile-block load, LDS constants, pipeline fill) costs about 3 tiles -- measured on the bench launch: S = 16
      // (39 full rounds of 16 tiles) 2.05 ms, S = 2 (4.875 -> 5 rounds of 128) 1.83 ms
      static const bool xcd_on = [] {
        const char *e = rsx::exp_env("RSX_SPEC_XCD");
        return !(e && e[0] == '0');
      }();
      if (xcd_on && !tb_cum) {
        int64_t forced = 0;
        if (const char *e = rsx::exp_env("RSX_SPEC_XCD_S")) forced = atoll(e);
        const plan::XcdSplit sp = plan::xcd_split(nqt, ntb, n_cu, forced);  // sc_plan.h
        if (sp.on) {
          a.xcd_nqt = sp.nqt_x;
          a.xcd_len = sp.len;
          grid = sp.grid;
        }
      }
    }
    hipLaunchKernelGGL(sc_spec2_filter_kernel, dim3(grid), dim3(512), S2_LDS_BYTES, s, a);
    RSX_HIP(hipGetLastError());
    if (want_prof2) {  // debugging aid only: synchronises
      unsigned long long h[16] = {0};
      RSX_HIP(hipStreamSynchronize(s));
      RSX_HIP(hipMemcpy(h, d_prof2, sizeof(h), hipMemcpyDeviceToHost));
      for (int hf = 0; hf < 2; hf++) {
        const double n = h[hf * 8 + 5] ? (double)h[hf * 8 + 5] : 1.0;
        if (hf == 0)
          fprintf(stderr, "[sc_spec2 prof] late wave, %llu tiles of the last segment of workgroup 0: cycles per tile  stage1 %.0f  seq+exchange writes %.0f"
                          "  exchange+const reads %.0f  stage2+bounds+store %.0f  wait %.0f | %.0f cycles per tile at %.0f MHz\n",
                  h[5], h[0] / n, h[1] / n, h[2] / n, h[3] / n, h[4] / n, h[6] / n, h[7] ? 100.0 * h[6] / h[7] : 0.0);
        else
          fprintf(stderr, "[sc_spec2 prof] early wave, %llu tiles: cycles per tile  exchange+const reads %.0f  stage2+bounds+store %.0f  dma issue %.0f"
                          "  stage1 %.0f  seq+exchange writes+wait %.0f | %.0f cycles per tile at %.0f MHz\n",
                  h[13], h[8] / n, h[9] / n, h[10] / n, h[11] / n, h[12] / n, h[14] / n, h[15] ? 100.0 * h[14] / h[15] : 0.0);
      }
    }
    return RSX_OK;
  }
  static unsigned long long *d_prof = nullptr;
  static const bool want_prof = rsx::exp_env("RSX_SPEC_PROF") != nullptr;
  if (want_prof && !d_prof) RSX_HIP(hipMalloc(&d_prof, 8 * sizeof(unsigned long long)));
  a.prof = want_prof ? d_prof : nullptr;
  hipLaunchKernelGGL(sc_spec_filter_kernel, dim3(grid), dim3(256), lds, s, a);
  RSX_HIP(hipGetLastError());
  if (want_prof) {  // debugging aid only: synchronises
    unsigned long long h[8] = {0};
    RSX_HIP(hipStreamSynchronize(s));
    RSX_HIP(hipMemcpy(h, d_prof, sizeof(h), hipMemcpyDeviceToHost));
    const double n = h[5] ? (double)h[5] : 1.0;
    fprintf(stderr, "[sc_spec prof] tiles %llu: cycles per tile  dma %.0f  stage1 %.0f  tail %.0f  stores %.0f  wait+barrier %.0f"
                    "  | tile loop %.0f cycles in %.1f us: shader clock %.0f MHz\n",
            h[5], h[0] / n, h[1] / n, h[2] / n, h[3] / n, h[4] / n, (double)h[6], h[7] / 100.0, h[7] ? 100.0 * h[6] / h[7] : 0.0);
  }
  return RSX_OK;
}

}  // namespace sc
}  // namespace rsx
