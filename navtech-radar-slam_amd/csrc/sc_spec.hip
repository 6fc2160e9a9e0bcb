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
// matrix cores as well (v_mfma_i32_32x32x32_i8: A = circulant of the query mask, rows ordered like the
// shifts of the stage-2 output; B = the entries' mask bytes; 4 MFMAs per query).  The division
// S_k / n_eff(k) is replaced by a multiplication with a quadratic upper bound of 1/n on
// [n_lo, n_hi] = [n_q + n_e - 60, min(n_q, n_e)] (chord minus c (n-n_lo)(n_hi-n), c = 1/(n_lo n_hi^2): exact
// at both ends, relative excess < (n_hi-n_lo)^3 / (4 n_lo n_hi^2), ~1e-4 for the usual few empty sectors),
// evaluated two shifts at a time with packed fp32 arithmetic.
//
// Error bound (u = 2^-11, everything in units of S; ||Q|| ||E|| = 15 sqrt(n_q n_e) by Parseval and
// unit columns):
//   * spectra rounded to fp16 (computed in fp64): each product off by <= 2u+u^2 relative;
//     |dRe C_f| , |dIm C_f| <= (2u+u^2) T_f,  T_f = sum |Q_f||E_f|;  through the inverse DFT
//     (|cos|+|sin| <= sqrt 2, sum_f T_f <= 15 sqrt(n_q n_e)):            <= sqrt2 (2u+u^2) sqrt(n_q n_e) = 1.381e-3 sqrt(n_q n_e)
//   * C_f (f >= 1) and the weights rounded to fp16: <= (2u+u^2) sum_f |C_f| (|cos|,|sin| <= 1 by Cauchy-
//     Schwarz on (Re, Im))                                                  <= 9.77e-4 sqrt(n_q n_e)
//   * fp32 accumulation (K = 160 and K = 16), C_0 hi/lo split (2^-22), fp16 subnormals, epilogue: < 4e-5 sqrt(n_q n_e)
//   total < 2.40e-3 sqrt(n_q n_e); kSpecEps = 2.5e-3.  The kernel returns
//        L~ = 1 - max_k S_k u(n_eff(k)) - kSpecEps sqrt(n_q n_e) / n_lo + filter_eps()
//   so that the direct filter's contract (L~ - filter_eps() <= L) holds unchanged downstream.
//
// Mapping: one wave per 32 entries (304 registers of entry spectra, resident), 4 waves = 128 entries per
// block, queries streamed through LDS in tiles of 4 (one tile per phase, double buffered, global_load_lds).
// Per (4 queries x 32 entries): 76 stage-1 MFMAs (one ds_read_b128 A fragment each) + 16 stage-2 MFMAs +
// 16 mask MFMAs, against 600 MFMAs for the same pairs in the direct filter.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "rsx_common.h"
#include "sc_kernels.h"

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

// query image (bytes); the strides make every A-fragment read bank-conflict free: a ds_read_b128 is served
// 16 lanes at a time and those 16 lanes (2 queries x 2 variants x 4 k4) must hit 16 different 16-byte
// slots mod 256: k4 * 80 B (5 slots), variant stride = 4 slots mod 16, query stride = 8 slots mod 16
constexpr int SP_DC_BYTES = 384;                    // 7 blocks x 48 B = 336, padded
constexpr int SP_VS = 576;                          // one variant stream: 7 blocks x 80 B = 560, padded (36 slots)
constexpr int SP_F_BYTES = 2 * SP_VS;               // 1152
constexpr int SP_TAIL = SP_DC_BYTES + 7 * SP_F_BYTES;  // 8448: {int n_q, int flags, float sqrt(n_q)}
// column-mask bytes for the n_eff MFMAs: M2[i] = mask bit (i mod 60); row k of the circulant reads 64
// consecutive bytes from M2[k]; 16 copies displaced by one byte each make that read 16-byte aligned
// (copy j = M2[j ..], row k uses copy k mod 16 at offset k - k mod 16) and, 112 B = 7 slots apart,
// bank-conflict free for the row order of the MFMA
constexpr int SP_MASK_OFF = SP_TAIL + 128;          // 8576
constexpr int SP_MASK_COPY = 112;
constexpr int SP_QS = SPEC_QIMG_BYTES;              // 10368 (648 slots = 8 mod 16)
static_assert(SP_QS == SP_MASK_OFF + 16 * SP_MASK_COPY, "layout");
static_assert((SP_VS / 16) % 16 == 4 && (SP_QS / 16) % 16 == 8, "bank-conflict-free strides");
constexpr int SP_QPT = 4;                           // queries per MFMA tile
constexpr int SP_TPP = 1;                           // tiles per LDS phase
constexpr int SP_QPP = SP_QPT * SP_TPP;
constexpr int SP_PHASE_BYTES = SP_QPP * SP_QS;      // 41472 = 40.5 KiB
constexpr int SP_B_VGPR = 12;                       // B fragments kept in VGPRs; the rest (64 x 4 registers) live in AGPRs

constexpr float kSpecEps = 2.5e-3f;
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
__device__ __forceinline__ u64 spectra_of(const float *__restrict__ d, const double *__restrict__ nrm, double *xn,
                                          _Float16 *kv, int lane) {
  bool nonzero = false, bad = false;
  if (lane < NS) {
    const double n = nrm[lane];
    nonzero = !(n == 0.0);  // SC.cpp:78: a column takes part unless its norm == 0
#pragma unroll
    for (int r = 0; r < NR; r++) {
      const double y = nonzero ? (double)d[lane * NR + r] / n : 0.0;
      bad |= !(fabs(y) <= 1.0000001);  // NaN, inf (or a norm that is not the column's)
      xn[r * NS + lane] = y;
    }
  }
  u64 m = __ballot(nonzero && lane < NS);
  if (__ballot(bad && lane < NS)) m |= kNonFinite;
  wave_lds_fence();
  // twiddles e^(-2 pi i m / 15)
  double cs[15], sn[15];
#pragma unroll
  for (int i = 0; i < 15; i++) {
    cs[i] = cospi(2.0 * i / 15.0);
    sn[i] = sinpi(2.0 * i / 15.0);
  }
  for (int idx = lane; idx < 8 * 80; idx += 64) {
    const int f = idx / 80, rem = idx - f * 80, a = rem / NR, r = rem - a * NR;
    double re = 0.0, im = 0.0;
#pragma unroll
    for (int b = 0; b < 15; b++) {
      const int c = (45 * a + 16 * b) % 60;  // CRT: c = a mod 4, c = b mod 15
      const double x = xn[r * NS + c];
      const int t = (f * b) % 15;
      double co = 0.0, si = 0.0;
#pragma unroll
      for (int i = 0; i < 15; i++) {  // select without a dynamically indexed (scratch) array
        co = (t == i) ? cs[i] : co;
        si = (t == i) ? sn[i] : si;
      }
      re += x * co;
      im -= x * si;
    }
    if (f == 0) {
      kv[a * 24 + r] = (_Float16)(float)re;
    } else {
      const int base = 96 + (f - 1) * 160 + a * 40 + r;
      kv[base] = (_Float16)(float)re;
      kv[base + 20] = (_Float16)(float)im;
    }
  }
  if (lane < 16) kv[(lane >> 2) * 24 + 20 + (lane & 3)] = (_Float16)0.0f;  // K padding of f = 0
  wave_lds_fence();
  return m;
}

// database image: tile-major [tile of 32 entries][76 K-steps][64 lanes][8 halves]
__global__ __launch_bounds__(256) void sc_spec_db_kernel(const float *__restrict__ desc, const double *__restrict__ norm,
                                                         int64_t first, int64_t count, uint4 *__restrict__ spT) {
  __shared__ double xn[4][DS];
  __shared__ __attribute__((aligned(16))) _Float16 kv[4][SP_KV];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t it = (int64_t)blockIdx.x * 4 + wave;
  if (it >= count) return;
  const int64_t slot = first + it;
  (void)spectra_of(desc + slot * DS, norm + slot * NS, xn[wave], kv[wave], lane);
  const int64_t tile = slot >> 5;
  const int col = (int)(slot & 31);
  for (int c = lane; c < 2 * SP_FRAGS; c += 64)
    spT[(tile * SP_FRAGS + (c >> 1)) * 64 + (c & 1) * 32 + col] = *reinterpret_cast<const uint4 *>(&kv[wave][c * 8]);
}

// query image: the LDS layout of the filter kernel, SP_QS bytes per query
//   [0, 384)                f = 0 stream: 7 blocks (a = 0,1,2,3,0,1,2) x 24
//   [384 + (f-1)*1152 ...)  re stream: 7 blocks x [Qr | Qi];  + 576: im stream: 7 blocks x [Qi | -Qr]
//   [8448, 8576)            n_q, flags, sqrt(n_q)
//   [8576, 10368)           16 displaced copies of the column-mask byte stream
__global__ __launch_bounds__(256) void sc_spec_query_kernel(const float *__restrict__ desc, const double *__restrict__ norm,
                                                            int32_t nq, char *__restrict__ qimg) {
  __shared__ double xn[4][DS];
  __shared__ __attribute__((aligned(16))) _Float16 kv[4][SP_KV];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = blockIdx.x * 4 + wave;
  if (q >= nq) return;
  const u64 m = spectra_of(desc + (int64_t)q * DS, norm + (int64_t)q * NS, xn[wave], kv[wave], lane);
  const _Float16 *k = kv[wave];
  uint4 *out = reinterpret_cast<uint4 *>(qimg + (int64_t)q * SP_QS);
  for (int c = lane; c < SP_QS / 16; c += 64) {
    half8 v;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int h = c * 8 + i;  // half index inside the image
      _Float16 x = (_Float16)0.0f;
      if (h < SP_DC_BYTES / 2) {
        if (h < 7 * 24) x = k[h % 96];
      } else if (h < SP_TAIL / 2) {
        const int h2 = h - SP_DC_BYTES / 2;
        const int f = h2 / (SP_F_BYTES / 2), w = h2 - f * (SP_F_BYTES / 2);  // f = 0..6 <-> frequency f+1
        const int base = 96 + f * 160;
        if (w < SP_VS / 2) {
          if (w < 280) x = k[base + w % 160];
        } else {
          const int p = w - SP_VS / 2;
          if (p < 280) {
            const int blk = (p / 40) & 3, t = p % 40;
            x = (t < 20) ? k[base + blk * 40 + 20 + t] : -k[base + blk * 40 + t - 20];
          }
        }
      }
      v[i] = x;
    }
    uint4 o = *reinterpret_cast<const uint4 *>(&v);
    if (c >= SP_MASK_OFF / 16) {
      const int byte0 = c * 16 - SP_MASK_OFF;
      const int j = byte0 / SP_MASK_COPY, i0 = byte0 - j * SP_MASK_COPY;  // 112 = 7 chunks: no chunk straddles copies
      unsigned w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int i = 0; i < 16; i++) w[i >> 2] |= (unsigned)((m >> ((j + i0 + i) % NS)) & 1ull) << (8 * (i & 3));
      o = uint4{w[0], w[1], w[2], w[3]};
    }
    if (c == SP_TAIL / 16) {
      const int n = __popcll(m & kMask60);
      o.x = (unsigned)n;
      o.y = (m & kNonFinite) ? 1u : 0u;
      o.z = __float_as_uint(sqrtf((float)n));
      o.w = 0u;
    }
    out[c] = o;
  }
}

// ------------------------------------------------------------------------------------------
// the filter
// ------------------------------------------------------------------------------------------
struct SpecArgs {
  const uint4 *spT;
  const u64 *cmask;
  const char *qimg;
  int64_t n_items;
  int32_t nq;
  int64_t per_block;  // (tile-block, query tile) work items per workgroup (no plan)
  float *lb;
  int64_t ld_lb;
  float eps_direct;
  const int32_t *tb_qmin;  // optional plan, in query-tile units (see sc_filter.hip)
  const int64_t *tb_cum;
};

__device__ __forceinline__ void stage_queries(const char *gsrc, char *ldst, int nbytes, int wave, int lane) {
  const int npieces = (nbytes + 1023) >> 10;
  for (int c = wave; c < npieces; c += 4)
    if (c * 1024 + lane * 16 < nbytes)  // nbytes is a multiple of 16; the last piece may be partial
      __builtin_amdgcn_global_load_lds(reinterpret_cast<const AS1 void *>(reinterpret_cast<uintptr_t>(gsrc + c * 1024 + lane * 16)),
                                       (AS3 void *)(ldst + c * 1024), 16, 0, 0);
}

__device__ __forceinline__ half8 lds_frag(const char *p) { return *reinterpret_cast<const half8 *>(p); }

__device__ __forceinline__ unsigned pack2(float a, float b) {
  float2v f = {a, b};
  half2v h = __builtin_convertvector(f, half2v);  // round to nearest even (v_cvt_pk_f16_f32)
  return __builtin_bit_cast(unsigned, h);
}

typedef int intx16 __attribute__((ext_vector_type(16)));
typedef int intx4 __attribute__((ext_vector_type(4)));

// per-lane constants of one segment
struct SpecLane {
  int dc_off, f_off;   // A-fragment offsets inside a tile of 4 query images (f = 0 / f >= 1 streams)
  int m_off[2];        // mask-row offsets of the two n_eff M-tiles (k4 = 0,1 / 2,3)
  int hh;
  half8 W;             // stage-2 A operand (inverse DFT weights)
  intx4 Bm[2];         // this lane's entry: column-mask bytes (B operand of the n_eff MFMAs)
  int n_e;
  float sqrt_ne;
  bool e_bad;
};

// one (4 queries x 32 entries) tile at LDS address `tbase`; best[q] = max_k (15/16) S_k * u(n_eff(k)),
// u = quadratic upper bound of 1/n on [n_lo, n_hi]; rl[q] = 1/n_lo
__device__ __forceinline__ void spec_tile(const char *tbase, const half8 (&B)[SP_FRAGS], const SpecLane &ln,
                                          float (&best)[SP_QPT], float (&rl)[SP_QPT]) {
  const char *dc_ptr = tbase + ln.dc_off, *f_ptr = tbase + ln.f_off;
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  u4 P[16];  // P[j] = the stage-2 B fragment of (query j/4, k4 = j%4): parts {C_0, C_1}, {C_2, C_3}, ...
  floatx16 z;
#pragma unroll
  for (int i = 0; i < 16; i++) z[i] = 0.0f;
  // stage 1, f = 0 and 1
  {
    floatx16 a0 = z, a1 = z;
#pragma unroll
    for (int s = 0; s < SP_DC_STEPS; s++) a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(lds_frag(dc_ptr + 32 * s), B[s], a0, 0, 0, 0);
#pragma unroll
    for (int s = 0; s < SP_F_STEPS; s++)
      a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(lds_frag(f_ptr + 32 * s), B[SP_DC_STEPS + s], a1, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 16; j++) {
      // C_0 as fp16 hi + lo: lanes 0..31 carry hi (k = 0), lanes 32..63 lo (k = 8); both weigh 1/16
      const float v = a0[j];
      const _Float16 h = (_Float16)v;
      const float lo = v - (float)h;
      P[j][0] = pack2(ln.hh ? lo : (float)h, a1[j]);
    }
  }
#pragma unroll
  for (int g = 1; g < 4; g++) {
    floatx16 a0 = z, a1 = z;
#pragma unroll
    for (int s = 0; s < SP_F_STEPS; s++) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(lds_frag(f_ptr + (2 * g - 1) * SP_F_BYTES + 32 * s),
                                                  B[SP_DC_STEPS + (2 * g - 1) * SP_F_STEPS + s], a0, 0, 0, 0);
    }
#pragma unroll
    for (int s = 0; s < SP_F_STEPS; s++) {
      a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(lds_frag(f_ptr + (2 * g) * SP_F_BYTES + 32 * s),
                                                  B[SP_DC_STEPS + (2 * g) * SP_F_STEPS + s], a1, 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 16; j++) P[j][g] = pack2(a0[j], a1[j]);
  }
  // per query: n_eff of the 60 shifts (2 x 2 int8 MFMAs), inverse DFT of every k4 (stage 2), running maximum
#pragma unroll
  for (int q = 0; q < SP_QPT; q++) {
    intx16 nacc[2];
#pragma unroll
    for (int mt = 0; mt < 2; mt++) {
      intx16 acc;
#pragma unroll
      for (int i = 0; i < 16; i++) acc[i] = 0;
#pragma unroll
      for (int s = 0; s < 2; s++)
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(*reinterpret_cast<const intx4 *>(tbase + q * SP_QS + ln.m_off[mt] + 32 * s),
                                                    ln.Bm[s], acc, 0, 0, 0);
      nacc[mt] = acc;
    }
    // 1/n <= u(n) = A + n (Bc + n C) on [L, H]
    const uint4 tail = *reinterpret_cast<const uint4 *>(tbase + q * SP_QS + SP_TAIL);
    const int n_q = (int)tail.x;
    const int li = (n_q + ln.n_e - NS > 1) ? (n_q + ln.n_e - NS) : 1;
    const int hmin = n_q < ln.n_e ? n_q : ln.n_e;
    const float L = (float)li, H = (float)(hmin > li ? hmin : li);
    const float rH = __builtin_amdgcn_rcpf(H), rLH = __builtin_amdgcn_rcpf(L * H);
    const float C = rLH * rH;
    const float A = fmaf(L + H, rLH, rH), Bc = -fmaf(C, L + H, rLH);
    const float2v A2 = {A, A}, B2 = {Bc, Bc}, C2 = {C, C};
    float m = 0.0f;  // rows 15..31 of the weight matrix are zero anyway (S >= 0 or clamped: valid)
#pragma unroll
    for (int k4 = 0; k4 < 4; k4++) {
      const int j = q * 4 + k4;
      const floatx16 d = __builtin_amdgcn_mfma_f32_32x32x16_f16(ln.W, __builtin_bit_cast(half8, P[j]), z, 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 8; i += 2) {
        const float2v n2 = {(float)nacc[k4 >> 1][(k4 & 1) * 8 + i], (float)nacc[k4 >> 1][(k4 & 1) * 8 + i + 1]};
        const float2v s2 = {d[i], d[i + 1]};
        const float2v u2 = __builtin_elementwise_fma(n2, __builtin_elementwise_fma(n2, C2, B2), A2);
        const float2v v2 = s2 * u2;
        m = fmaxf(m, fmaxf(v2[0], v2[1]));
      }
    }
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
    best[q] = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    rl[q] = __builtin_amdgcn_rcpf(L);
  }
}

__global__ __launch_bounds__(256, 1) void sc_spec_filter_kernel(SpecArgs a) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 31, hh = lane >> 5;
  const int64_t ntiles = (a.n_items + 31) >> 5;
  const int64_t ntb = (ntiles + 3) >> 2;
  const int nqt = (a.nq + SP_QPT - 1) / SP_QPT;
  const int64_t total = a.tb_cum ? a.tb_cum[ntb] : ntb * (int64_t)nqt;
  const int64_t per = a.tb_cum ? (total + gridDim.x - 1) / gridDim.x : a.per_block;
  int64_t L0 = (int64_t)blockIdx.x * per;
  const int64_t L1 = (L0 + per < total) ? (L0 + per) : total;
  SpecLane ln;
  ln.hh = hh;
  {
    // A-fragment address of this lane inside a tile of 4 query images: row = col = 8 * query + 4 * variant + k4
    const int rq = col >> 3, rv = (col >> 2) & 1, rk = col & 3;
    ln.dc_off = rq * SP_QS + rk * 48 + hh * 16;
    ln.f_off = rq * SP_QS + SP_DC_BYTES + rv * SP_VS + rk * 80 + hh * 16;
    // n_eff rows: row = col <-> (k4 = 2 * mt + col / 16, k15 = col % 16), the row order of the stage-2 output
#pragma unroll
    for (int mt = 0; mt < 2; mt++) {
      const int k4 = 2 * mt + (col >> 4), k15 = (col & 15) == 15 ? 0 : (col & 15);
      const int k = (45 * k4 + 16 * k15) % NS;  // CRT
      ln.m_off[mt] = SP_MASK_OFF + (k & 15) * SP_MASK_COPY + (k & ~15) + hh * 16;
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

    {  // phase 0 of the query stream (DMA, overlaps the B loads below)
      const int nqs = (q1 - q0 < SP_QPP) ? (q1 - q0) : SP_QPP;
      stage_queries(a.qimg + (int64_t)q0 * SP_QS, smem, nqs * SP_QS, wave, lane);
    }
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
    }
    const u64 m2 = n_ok ? a.cmask[n] : 0ull;
    ln.n_e = __popcll(m2 & kMask60);
    ln.sqrt_ne = sqrtf((float)ln.n_e);
    ln.e_bad = (m2 & kNonFinite) != 0;
#pragma unroll
    for (int s = 0; s < 2; s++) {
      const unsigned bits = (unsigned)((m2 & kMask60) >> (32 * s + 16 * hh)) & 0xffffu;
#pragma unroll
      for (int r = 0; r < 4; r++)  // 4 bits -> 4 bytes of 0/1
        ln.Bm[s][r] = (int)((((bits >> (4 * r)) & 0xfu) * 0x00204081u) & 0x01010101u);
    }

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int p = 0; p < nphase; p++) {
      const int qp = q0 + p * SP_QPP;
      if (p + 1 < nphase) {
        const int qn = qp + SP_QPP;
        const int nqs = (q1 - qn < SP_QPP) ? (q1 - qn) : SP_QPP;
        stage_queries(a.qimg + (int64_t)qn * SP_QS, smem + ((p + 1) & 1) * SP_PHASE_BYTES, nqs * SP_QS, wave, lane);
      }
      const int nq_here = (q1 - qp < SP_QPP) ? (q1 - qp) : SP_QPP;
      if (tile_ok) {
        const char *phase = smem + (p & 1) * SP_PHASE_BYTES;
        for (int t = 0; t * SP_QPT < nq_here; t++) {
          const char *tbase = phase + t * SP_QPT * SP_QS;
          float best[SP_QPT], rl[SP_QPT];
          spec_tile(tbase, B, ln, best, rl);
#pragma unroll
          for (int qq = 0; qq < SP_QPT; qq++) {
            const int q = qp + t * SP_QPT + qq;
            if (q < q1) {
              const uint4 tail = *reinterpret_cast<const uint4 *>(tbase + qq * SP_QS + SP_TAIL);
              const int n_q = (int)tail.x;
              const float err = kSpecEps * __uint_as_float(tail.z) * ln.sqrt_ne;
              // best = 15/16 max_k S_k u(n_k); the error of S is divided by n_k >= n_lo
              float v = (1.0f + a.eps_direct) - fmaf(best[qq], (16.0f / 15.0f) * (1.0f + 4e-6f), err * rl[qq]);
              if (n_q == 0 || ln.n_e == 0) v = INFINITY;    // no effective column at any shift: never a hit
              if (tail.y != 0u || ln.e_bad) v = -INFINITY;  // non-finite input: always re-score exactly
              if (n_ok && hh == 0) a.lb[(int64_t)q * a.ld_lb + n] = v;
            }
          }
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }
}

}  // namespace

size_t spec_qimg_bytes(int32_t nq) { return (size_t)nq * SPEC_QIMG_BYTES + 1024; }

int launch_spec_db_images(const float *desc, const double *norm, int64_t first, int64_t count, void *spT, hipStream_t s) {
  if (count <= 0) return RSX_OK;
  hipLaunchKernelGGL(sc_spec_db_kernel, dim3((unsigned)((count + 3) / 4)), dim3(256), 0, s, desc, norm, first, count,
                     static_cast<uint4 *>(spT));
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

const char *spec_filter_kernel_name() { return "sc_spec_filter_kernel"; }

int launch_spec_filter(const DbView &db, const void *qimg, int32_t nq, int64_t n_items, float *lb, int64_t ld_lb,
                       const int32_t *tb_qmin, const int64_t *tb_cum, hipStream_t s) {
  if (nq <= 0 || n_items <= 0) return RSX_OK;
  static int n_cu = 0;
  const int lds = 2 * SP_PHASE_BYTES;
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
  a.qimg = static_cast<const char *>(qimg);
  a.n_items = n_items;
  a.nq = nq;
  a.lb = lb;
  a.ld_lb = ld_lb;
  a.eps_direct = (float)filter_eps();
  const int64_t ntiles = (n_items + 31) / 32;
  const int64_t nqt = (nq + SP_QPT - 1) / SP_QPT;
  const int64_t total = ((ntiles + 3) / 4) * nqt;
  // one workgroup per CU with an equal share; small problems use fewer workgroups so that a
  // 304-register B load is amortised over >= 4 query tiles
  int64_t per = (total + n_cu - 1) / n_cu;
  if (per < 4) per = 4;
  a.per_block = per;
  a.tb_qmin = tb_qmin;
  a.tb_cum = tb_cum;
  const unsigned grid = tb_cum ? (unsigned)n_cu : (unsigned)((total + per - 1) / per);
  hipLaunchKernelGGL(sc_spec_filter_kernel, dim3(grid), dim3(256), lds, s, a);
  RSX_HIP(hipGetLastError());
  return RSX_OK;
}

}  // namespace sc
}  // namespace rsx
