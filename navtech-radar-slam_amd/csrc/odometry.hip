// odometry.hip -- the per-sequence pipeline of the upstream file-based odometry.cpp entry, device resident and batched.
//
// What the reference runs per scan (its ORORA submodule, an empty directory in the reference checkout: .gitmodules:1-3;
// README.md:26-29,54-60; launch/navtech_radar_slam_mulran.launch:5-8): polar image -> cen2019 keypoints -> Cartesian image
// -> ORB descriptors -> BFMatcher knnMatch(2) + ratio against the previous scan -> ORORA (GNC rotation, A-COTE
// translation) -> pose composition.  The sequence is on disk, so every scan and every consecutive pair is independent
// until the final composition: a WINDOW of n scans goes through each stage in ONE launch chain --
//     rsx_cen2019_extract_batch_device          n images   (csrc/cen2019.hip)
//     rsx_frontend_cartesian_batch_device       n images   (csrc/frontend.hip)
//     rsx_frontend_describe_batch_device        n keypoint sets
//     rsx_frontend_match_consecutive_device     all consecutive pairs, both directions
//     odo_cross / odo_gather                    cross check (the two directions must agree) + correspondence lists
//     rsx_orora_register_batch_device           all pairs of the window in one call (csrc/orora.hip): max-clique inlier
//                                               selection (csrc/pmc.hip, RSX_ORORA_PMC: on by default here) + the solver
// -- with every intermediate (keypoints, descriptors, matches, correspondences) in HBM; one upload of the images and one
// download of 48 bytes per scan (+ the keypoints when the caller wants /orora/cloud_local).  The last scan of a window
// stays on the device as the "previous scan" of the next one.  Pose composition stays on the host (sequential, trivial).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <mutex>
#include <new>

#include "rsx_common.h"

namespace {

#ifndef RSX_ODO_WINDOW
#define RSX_ODO_WINDOW 64
#endif
constexpr int MAX_WINDOW = RSX_ODO_WINDOW;  // scans per internal launch chain (workspace: ~26 MB per scan and extraction lane, two lanes)

// one block per consecutive pair j (slots A = first + j, B = A + 1): keep prev keypoint i when fwd[i] = k >= 0 and
// bwd[k] = i, in ascending i (the order the host loop of round 2 produced); src = the CURRENT scan's point, dst = the
// previous scan's point, so that ORORA returns the motion of the sensor expressed in the previous frame.
__global__ __launch_bounds__(256) void odo_cross(const float *__restrict__ xy, const int32_t *__restrict__ counts, int stride, int first,
                                                 const int32_t *__restrict__ fwd, const int32_t *__restrict__ bwd,
                                                 float2 *__restrict__ stage_src, float2 *__restrict__ stage_dst, int32_t *__restrict__ pair_cnt) {
  __shared__ unsigned s_w[4];
  const int j = blockIdx.x, A = first + j, B = A + 1;
  const int nA = counts[A] < stride ? counts[A] : stride, nB = counts[B] < stride ? counts[B] : stride;
  const int32_t *f = fwd + (int64_t)j * stride, *b = bwd + (int64_t)j * stride;
  const float2 *pa = reinterpret_cast<const float2 *>(xy) + (int64_t)A * stride, *pb = reinterpret_cast<const float2 *>(xy) + (int64_t)B * stride;
  float2 *ss = stage_src + (int64_t)j * stride, *sd = stage_dst + (int64_t)j * stride;
  unsigned run = 0;
  for (int base = 0; base < nA; base += 256) {
    const int i = base + threadIdx.x;
    int k = -1;
    if (i < nA) {
      k = f[i];
      if (k >= nB || (k >= 0 && b[k] != i)) k = -1;
    }
    const unsigned long long bal = __ballot(k >= 0);
    const unsigned lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) s_w[w] = (unsigned)__popcll(bal);
    __syncthreads();
    unsigned before = run, total = 0;
    for (unsigned ww = 0; ww < 4; ww++) {
      if (ww < w) before += s_w[ww];
      total += s_w[ww];
    }
    if (k >= 0) {
      const unsigned pos = before + (unsigned)__popcll(bal & ((1ull << lane) - 1ull));
      ss[pos] = pb[k];
      sd[pos] = pa[i];
    }
    run += total;
    __syncthreads();
  }
  if (threadIdx.x == 0) pair_cnt[j] = (int32_t)run;
}

// one block per pair: contiguous correspondence arrays + the offsets rsx_orora_register_batch_device wants
__global__ __launch_bounds__(256) void odo_gather(const float2 *__restrict__ stage_src, const float2 *__restrict__ stage_dst,
                                                  const int32_t *__restrict__ pair_cnt, int n_pairs, int stride, float2 *__restrict__ src,
                                                  float2 *__restrict__ dst, int64_t *__restrict__ offsets) {
  __shared__ long long s_w[4];
  const int j = blockIdx.x;
  long long before = 0;
  for (int r = threadIdx.x; r < j; r += 256) before += pair_cnt[r];
  for (int o = 32; o >= 1; o >>= 1) before += __shfl_xor(before, o);
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = before;
  __syncthreads();
  before = s_w[0] + s_w[1] + s_w[2] + s_w[3];
  const int n = pair_cnt[j];
  for (int i = threadIdx.x; i < n; i += 256) {
    src[before + i] = stage_src[(int64_t)j * stride + i];
    dst[before + i] = stage_dst[(int64_t)j * stride + i];
  }
  if (threadIdx.x == 0) {
    offsets[j] = before;
    if (j == n_pairs - 1) offsets[n_pairs] = before + n;
  }
}

}  // namespace

// Round 6: several windows in flight.  A window is two stages: EXTRACTION (cen2019, Cartesian images, descriptors: the wide
// kernels, independent of every other window) and MATCHING (consecutive matches, cross check, max-clique selection, solver:
// a chain of small grids -- one or two workgroups per pair -- that leaves most of the device idle, and the only stage that
// needs the previous window).  While window g is matched, windows g + 1 AND g + 2 are extracted, on two extraction LANES (a
// stream, a cen2019 handle and a front-end handle each; window g takes lane g & 1): the extraction chain has its own narrow
// launches (the selection kernels of cen2019: one workgroup per image) that a second chain fills -- two independent handles
// on one device reach 66.7 k scans/s where one with a single lane reaches 59.3 k.  What the stages hand over -- keypoints,
// descriptors, counts, slots 0 .. n of a SET -- exists three times (window g works in set g % 3); the extraction stage ends
// by copying its last scan into slot 0 of the NEXT set ("the previous scan" of window g + 1).
//   lane g & 1:        E(g):   [wait M(g-3)]  cen2019, Cartesian, describe -> set g%3   [wait M(g-2)]  carry -> set (g+1)%3 slot 0
//   matching stream:   M(g):   [wait E(g), E(g-1)]  match, cross, gather, select + solve -> pinned results g%3
// The host enqueues M(g), then E(g + 2), and only then waits for M(g).
constexpr int N_SETS = 3, N_LANES = 2;
struct OdoSet {
  rsx::DevBuf az, targets, xy, counts, desc, valid;  // slot 0 = the previous scan, slots 1 .. MAX_WINDOW = the window
  void *pin = nullptr;  // pinned: counts[MAX_WINDOW + 1], pair_cnt[MAX_WINDOW], results[MAX_WINDOW], then the staged azimuth grids
};

struct rsx_odometry {
  int device = 0, rows = 0, cols = 0;
  rsx_odometry_params prm{};
  std::mutex mu;
  hipStream_t lane_stream[N_LANES] = {}, match_stream = nullptr, copy_stream = nullptr;
  hipEvent_t ev_up = nullptr, ev_e[N_SETS] = {}, ev_m[N_SETS] = {};
  rsx_cen2019 *cen[N_LANES] = {};
  rsx_frontend *fe[N_LANES] = {};
  rsx_orora *reg = nullptr;
  OdoSet set[N_SETS];
  rsx::DevBuf imgs[N_SETS], fwd, bwd, stage_src, stage_dst, pair_cnt, src, dst, offsets, results;
  uint64_t windows = 0;  // windows enqueued since creation: window g works in set[g % N_SETS] on lane g & 1
  bool have_prev = false;
};

using rsx::fail;

namespace {

constexpr size_t PIN_COUNTS = 0, PIN_PAIRS = 4 * (MAX_WINDOW + 1), PIN_RES = PIN_PAIRS + 4 * MAX_WINDOW + 4,
                 PIN_AZ = (PIN_RES + sizeof(rsx_orora_result) * MAX_WINDOW + 255) / 256 * 256;
size_t pin_bytes(const rsx_odometry *h) { return PIN_AZ + (size_t)h->rows * 4 * MAX_WINDOW; }

int reserve_all(rsx_odometry *h, size_t ibytes, hipStream_t s) {
  const size_t K = (size_t)h->prm.max_keypoints, S = MAX_WINDOW + 1;
  for (rsx::DevBuf &b : h->imgs) RSX_TRY(b.reserve(ibytes * MAX_WINDOW, s, false));
  for (OdoSet &q : h->set) {
    RSX_TRY(q.az.reserve((size_t)h->rows * 4 * MAX_WINDOW, s, false));
    RSX_TRY(q.targets.reserve(S * K * 8, s, false));
    RSX_TRY(q.xy.reserve(S * K * 8, s, true));
    RSX_TRY(q.counts.reserve(S * 4, s, true));
    RSX_TRY(q.desc.reserve(S * K * 32, s, true));
    RSX_TRY(q.valid.reserve(S * K, s, true));
  }
  RSX_TRY(h->fwd.reserve((size_t)MAX_WINDOW * K * 4, s, false));
  RSX_TRY(h->bwd.reserve((size_t)MAX_WINDOW * K * 4, s, false));
  RSX_TRY(h->stage_src.reserve((size_t)MAX_WINDOW * K * 8, s, false));
  RSX_TRY(h->stage_dst.reserve((size_t)MAX_WINDOW * K * 8, s, false));
  RSX_TRY(h->pair_cnt.reserve((size_t)MAX_WINDOW * 4, s, false));
  RSX_TRY(h->src.reserve((size_t)MAX_WINDOW * K * 8, s, false));
  RSX_TRY(h->dst.reserve((size_t)MAX_WINDOW * K * 8, s, false));
  RSX_TRY(h->offsets.reserve((size_t)(MAX_WINDOW + 1) * 8, s, false));
  RSX_TRY(h->results.reserve((size_t)MAX_WINDOW * sizeof(rsx_orora_result), s, false));
  return RSX_OK;
}

// E(g): window g (n <= MAX_WINDOW scans whose images are at d_imgs, device) through cen2019, the Cartesian images and the
// descriptors into set g % 3, then its last scan into slot 0 of the next set.  Asynchronous on the stream of lane g & 1.
int enqueue_extract(rsx_odometry *h, uint64_t g, const uint8_t *d_imgs, int n, int64_t img_stride, int32_t row_stride, const float *azimuths,
                    int32_t azimuths_per_image) {
  const int lane = (int)(g & 1);
  hipStream_t s = h->lane_stream[lane];
  OdoSet &q = h->set[g % N_SETS], &nx = h->set[(g + 1) % N_SETS];
  const int K = h->prm.max_keypoints;
  const size_t slot_xy = (size_t)K * 2;
  RSX_HIP(hipStreamWaitEvent(s, h->ev_m[g % N_SETS], 0));  // M(g - 3) read this set (a no-op before the event's first record)
  // the azimuth grids go to the device once: cen2019's polar -> Cartesian and the Cartesian image both read them there.
  // Staged in pinned memory so that the copy does not wait for the stream (the area was last read by E(g - 3): long done)
  const size_t na = (size_t)h->rows * (azimuths_per_image ? n : 1);
  float *paz = reinterpret_cast<float *>(static_cast<char *>(q.pin) + PIN_AZ);
  std::memcpy(paz, azimuths, na * 4);
  RSX_HIP(hipMemcpyAsync(q.az.p, paz, na * 4, hipMemcpyHostToDevice, s));
  int32_t *d_counts = q.counts.as<int32_t>();
  RSX_TRY(rsx_cen2019_extract_batch_device(h->cen[lane], d_imgs, n, img_stride, row_stride, h->prm.col_offset, &h->prm.cen, q.az.as<float>(),
                                           azimuths_per_image, h->prm.radar_resolution, q.targets.as<int32_t>() + slot_xy,
                                           q.xy.as<float>() + slot_xy, K, d_counts + 1, s));
  // the Cartesian image of scan i through scan i's OWN azimuth grid (already in HBM for cen2019): results do not depend on
  // how the sequence is cut into windows, and nothing about the grids is looked at on the host
  RSX_TRY(rsx_frontend_cartesian_batch_device_az(h->fe[lane], d_imgs, n, img_stride, row_stride, h->prm.col_offset, q.az.as<float>(),
                                                 azimuths_per_image ? (int64_t)h->rows : 0, h->prm.radar_resolution, s));
  RSX_TRY(rsx_frontend_describe_batch_device(h->fe[lane], q.xy.as<float>() + slot_xy, d_counts + 1, n, K, q.desc.as<uint8_t>() + (size_t)K * 32,
                                             q.valid.as<uint8_t>() + (size_t)K, s));
  // the last scan of the window becomes the previous scan of the next one (slot 0 of the next set, which M(g - 2) read; the
  // next window's own extraction, on the other lane, writes slots 1 .. n of that set only)
  RSX_HIP(hipStreamWaitEvent(s, h->ev_m[(g + 1) % N_SETS], 0));
  RSX_HIP(hipMemcpyAsync(nx.xy.p, q.xy.as<float>() + (size_t)n * slot_xy, slot_xy * 4, hipMemcpyDeviceToDevice, s));
  RSX_HIP(hipMemcpyAsync(nx.desc.p, q.desc.as<uint8_t>() + (size_t)n * K * 32, (size_t)K * 32, hipMemcpyDeviceToDevice, s));
  RSX_HIP(hipMemcpyAsync(nx.valid.p, q.valid.as<uint8_t>() + (size_t)n * K, (size_t)K, hipMemcpyDeviceToDevice, s));
  RSX_HIP(hipMemcpyAsync(nx.counts.p, d_counts + n, 4, hipMemcpyDeviceToDevice, s));
  RSX_HIP(hipEventRecord(h->ev_e[g % N_SETS], s));
  return RSX_OK;
}

// M(g): the consecutive pairs of window g (with the previous scan in slot 0 when there is one) matched, selected and solved,
// the results on their way to the set's pinned area.  Asynchronous on the matching stream.  *first_out: 0 when slot 0 takes part.
int enqueue_match(rsx_odometry *h, uint64_t g, int n, int *first_out) {
  hipStream_t s = h->match_stream;
  OdoSet &q = h->set[g % N_SETS];
  const int K = h->prm.max_keypoints;
  RSX_HIP(hipStreamWaitEvent(s, h->ev_e[g % N_SETS], 0));
  RSX_HIP(hipStreamWaitEvent(s, h->ev_e[(g + N_SETS - 1) % N_SETS], 0));  // E(g - 1): its carry into slot 0 (the other lane's stream)
  int32_t *d_counts = q.counts.as<int32_t>();
  const int first = h->have_prev ? 0 : 1, n_pairs = n - first;
  *first_out = first;
  if (n_pairs > 0) {
    RSX_TRY(rsx_frontend_match_consecutive_device(h->fe[g & 1], q.desc.as<uint8_t>(), q.valid.as<uint8_t>(), d_counts, K, first, n_pairs,
                                                  h->prm.frontend.ratio, h->fwd.as<int32_t>(), h->bwd.as<int32_t>(), s));
    hipLaunchKernelGGL(odo_cross, dim3((unsigned)n_pairs), dim3(256), 0, s, q.xy.as<float>(), d_counts, K, first, h->fwd.as<int32_t>(),
                       h->bwd.as<int32_t>(), h->stage_src.as<float2>(), h->stage_dst.as<float2>(), h->pair_cnt.as<int32_t>());
    hipLaunchKernelGGL(odo_gather, dim3((unsigned)n_pairs), dim3(256), 0, s, h->stage_src.as<float2>(), h->stage_dst.as<float2>(),
                       h->pair_cnt.as<int32_t>(), n_pairs, K, h->src.as<float2>(), h->dst.as<float2>(), h->offsets.as<int64_t>());
    RSX_HIP(hipGetLastError());
    RSX_TRY(rsx_orora_register_batch_device(h->reg, h->src.as<float>(), h->dst.as<float>(), h->offsets.as<int64_t>(), n_pairs, &h->prm.orora,
                                            h->results.as<rsx_orora_result>(), s));
  }
  char *pin = static_cast<char *>(q.pin);
  RSX_HIP(hipMemcpyAsync(pin + PIN_COUNTS, d_counts, (size_t)(n + 1) * 4, hipMemcpyDeviceToHost, s));
  if (n_pairs > 0) {
    RSX_HIP(hipMemcpyAsync(pin + PIN_PAIRS, h->pair_cnt.p, (size_t)n_pairs * 4, hipMemcpyDeviceToHost, s));
    RSX_HIP(hipMemcpyAsync(pin + PIN_RES, h->results.p, (size_t)n_pairs * sizeof(rsx_orora_result), hipMemcpyDeviceToHost, s));
  }
  RSX_HIP(hipEventRecord(h->ev_m[g % N_SETS], s));
  h->have_prev = true;
  return RSX_OK;
}

// wait for M(g), fill out[0..n) (+ the keypoints)
int finish_window(rsx_odometry *h, uint64_t g, int n, int first, rsx_odometry_scan *out, float *out_xy, int32_t max_xy) {
  OdoSet &q = h->set[g % N_SETS];
  const int K = h->prm.max_keypoints;
  const size_t slot_xy = (size_t)K * 2;
  char *pin = static_cast<char *>(q.pin);
  RSX_HIP(hipEventSynchronize(h->ev_m[g % N_SETS]));
  const int32_t *hc = reinterpret_cast<const int32_t *>(pin + PIN_COUNTS), *hp = reinterpret_cast<const int32_t *>(pin + PIN_PAIRS);
  const rsx_orora_result *hr = reinterpret_cast<const rsx_orora_result *>(pin + PIN_RES);
  for (int i = 0; i < n; i++) {
    rsx_odometry_scan &o = out[i];
    std::memset(&o, 0, sizeof(o));
    o.n_keypoints = hc[1 + i];
    const int pair = first ? i - 1 : i;  // index of the pair (scan i-1, scan i) in this window
    if (pair >= 0) {
      o.reg = hr[pair];
      o.n_matches = hp[pair];
    } else {
      o.reg.status = 3;  // the first scan of a sequence: there is no previous scan
    }
  }
  if (out_xy && max_xy > 0) {  // (set g % 3 is not written again before E(g + 3), which the host enqueues after this returns)
    hipStream_t s = h->match_stream;
    for (int i = 0; i < n; i++) {
      const int c = hc[1 + i] < K ? hc[1 + i] : K, wn = c < max_xy ? c : max_xy;
      if (wn > 0)
        RSX_HIP(hipMemcpyAsync(out_xy + (size_t)i * max_xy * 2, q.xy.as<float>() + (size_t)(1 + i) * slot_xy, (size_t)wn * 8, hipMemcpyDeviceToHost, s));
    }
    RSX_HIP(hipStreamSynchronize(s));
  }
  return RSX_OK;
}

// after an error in the middle of a sequence: nothing in flight, the next scan starts a new sequence
void abandon(rsx_odometry *h) {
  for (hipStream_t ls : h->lane_stream) (void)hipStreamSynchronize(ls);
  (void)hipStreamSynchronize(h->match_stream);
  h->have_prev = false;
}

int check_layout(rsx_odometry *h, int32_t n, int64_t image_stride_bytes, int32_t row_stride) {
  if (row_stride < h->prm.col_offset + h->cols) return fail(RSX_ERR_BAD_ARG, "row_stride %d too small for offset %d + %d columns", row_stride, h->prm.col_offset, h->cols);
  if (n > 1 && image_stride_bytes < (int64_t)h->rows * row_stride) return fail(RSX_ERR_BAD_ARG, "image_stride_bytes smaller than an image");
  return RSX_OK;
}

// the azimuth grids are host arrays here: every grid must ascend (the Cartesian remap divides by az[1] - az[0] on the
// device, frontend.hip az_row_of: a zero or negative step would give NaN row indices and NaN images instead of a status)
int check_azimuths(const rsx_odometry *h, const float *azimuths, int32_t per_image, int32_t n_scans) {
  if (h->rows < 2) return RSX_OK;
  const int grids = per_image ? n_scans : 1;
  for (int g = 0; g < grids; g++) {
    const float *az = azimuths + (size_t)g * h->rows;
    if (!((double)az[1] - (double)az[0] > 0.0)) return fail(RSX_ERR_BAD_ARG, "azimuths must increase (scan %d)", g);
  }
  return RSX_OK;
}

}  // namespace

extern "C" {

int rsx_odometry_default_params(rsx_odometry_params *p) try {
  if (!p) return fail(RSX_ERR_BAD_ARG, "null params");
  std::memset(p, 0, sizeof(*p));
  rsx_cen2019_default_params(&p->cen);
  rsx_frontend_default_params(&p->frontend);
  rsx_orora_default_params(&p->orora);
  p->orora.flags |= RSX_ORORA_PMC;  // the upstream pipeline prunes the matches to the max clique before the solver (csrc/pmc.hip)
  p->radar_resolution = 0.0595f;  // Navtech CIR204-H range bin [m] (MulRan)
  p->col_offset = 11;             // metadata bytes in front of every polar_oxford_form row
  p->max_keypoints = 16384;  // = rsx_orora_max_correspondences(): a pair can never exceed the solver's capacity
  p->device = 0;
  return RSX_OK;
} RSX_CATCH_ALL

int rsx_odometry_create(const rsx_odometry_params *params, int32_t rows, int32_t cols, rsx_odometry **out) try {
  if (!out) return fail(RSX_ERR_BAD_ARG, "null out");
  *out = nullptr;
  rsx_odometry_params p;
  rsx_odometry_default_params(&p);
  if (params) p = *params;
  if (p.max_keypoints < 16 || p.max_keypoints > rsx_orora_max_correspondences())
    return fail(RSX_ERR_BAD_ARG, "max_keypoints %d outside [16, %d]", p.max_keypoints, rsx_orora_max_correspondences());
  if (p.col_offset < 0 || !(p.radar_resolution > 0.0f)) return fail(RSX_ERR_BAD_ARG, "bad col_offset / radar_resolution");
  rsx_odometry *h = new (std::nothrow) rsx_odometry();
  if (!h) return fail(RSX_ERR_OOM, "host alloc");
  h->device = p.device;
  h->rows = rows;
  h->cols = cols;
  h->prm = p;
  int st = RSX_OK;
  for (int l = 0; l < N_LANES && st == RSX_OK; l++) {
    st = rsx_cen2019_create(p.device, rows, cols, &h->cen[l]);
    if (st == RSX_OK) st = rsx_frontend_create(p.device, rows, cols, &p.frontend, &h->fe[l]);
  }
  if (st == RSX_OK) st = rsx_orora_create(p.device, &h->reg);
  if (st == RSX_OK && (p.orora.flags & RSX_ORORA_PMC)) st = rsx_orora_reserve(h->reg, (int64_t)MAX_WINDOW * p.max_keypoints);
  if (st == RSX_OK) {
    hipError_t e = hipSetDevice(p.device);
    for (hipStream_t &ls : h->lane_stream)
      if (e == hipSuccess) e = hipStreamCreateWithFlags(&ls, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->match_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_up, hipEventDisableTiming);
    for (int i = 0; i < N_SETS; i++) {
      if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_e[i], hipEventDisableTiming);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_m[i], hipEventDisableTiming);
      if (e == hipSuccess) e = hipHostMalloc(&h->set[i].pin, pin_bytes(h), hipHostMallocDefault);
    }
    if (e != hipSuccess) st = fail(e == hipErrorOutOfMemory ? RSX_ERR_OOM : RSX_ERR_HIP, "odometry create: %s", hipGetErrorString(e));
  }
  if (st != RSX_OK) {
    rsx_odometry_destroy(h);
    return st;
  }
  *out = h;
  return RSX_OK;
} RSX_CATCH_ALL

int rsx_odometry_destroy(rsx_odometry *h) try {
  if (!h) return RSX_OK;
  (void)hipSetDevice(h->device);
  for (hipStream_t ls : h->lane_stream)
    if (ls) (void)hipStreamSynchronize(ls);
  if (h->match_stream) (void)hipStreamSynchronize(h->match_stream);
  if (h->copy_stream) (void)hipStreamSynchronize(h->copy_stream);
  for (rsx::DevBuf &b : h->imgs) b.release();
  for (rsx::DevBuf *b : {&h->fwd, &h->bwd, &h->stage_src, &h->stage_dst, &h->pair_cnt, &h->src, &h->dst, &h->offsets, &h->results})
    b->release();
  for (OdoSet &q : h->set) {
    for (rsx::DevBuf *b : {&q.az, &q.targets, &q.xy, &q.counts, &q.desc, &q.valid}) b->release();
    if (q.pin) (void)hipHostFree(q.pin);
  }
  for (int l = 0; l < N_LANES; l++) {
    rsx_cen2019_destroy(h->cen[l]);
    rsx_frontend_destroy(h->fe[l]);
  }
  rsx_orora_destroy(h->reg);
  if (h->ev_up) (void)hipEventDestroy(h->ev_up);
  for (int i = 0; i < N_SETS; i++) {
    if (h->ev_e[i]) (void)hipEventDestroy(h->ev_e[i]);
    if (h->ev_m[i]) (void)hipEventDestroy(h->ev_m[i]);
  }
  if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
  if (h->match_stream) (void)hipStreamDestroy(h->match_stream);
  for (hipStream_t ls : h->lane_stream)
    if (ls) (void)hipStreamDestroy(ls);
  delete h;
  return RSX_OK;
} RSX_CATCH_ALL

int rsx_odometry_reset(rsx_odometry *h) try {
  if (!h) return fail(RSX_ERR_BAD_ARG, "null handle");
  std::lock_guard<std::mutex> lk(h->mu);
  h->have_prev = false;
  return RSX_OK;
} RSX_CATCH_ALL

int rsx_odometry_window(void) { return MAX_WINDOW; }

int rsx_odometry_push_device(rsx_odometry *h, const uint8_t *d_imgs, int32_t n_scans, int64_t image_stride_bytes, int32_t row_stride,
                             const float *azimuths, int32_t azimuths_per_image, rsx_odometry_scan *out, float *out_xy, int32_t max_xy) try {
  if (!h || !d_imgs || !azimuths || !out || n_scans < 0 || max_xy < 0) return fail(RSX_ERR_BAD_ARG, "bad arg");
  if (n_scans == 0) return RSX_OK;
  RSX_TRY(check_layout(h, n_scans, image_stride_bytes, row_stride));
  RSX_TRY(check_azimuths(h, azimuths, azimuths_per_image, n_scans));
  std::lock_guard<std::mutex> lk(h->mu);
  RSX_HIP(hipSetDevice(h->device));
  RSX_TRY(reserve_all(h, 0, h->match_stream));
  auto window = [&](int w, uint64_t g) -> int {  // E(g) of the call's window w
    const int b0 = w * MAX_WINDOW, n = n_scans - b0 < MAX_WINDOW ? n_scans - b0 : MAX_WINDOW;
    return enqueue_extract(h, g, d_imgs + (int64_t)b0 * image_stride_bytes, n, image_stride_bytes, row_stride,
                           azimuths + (azimuths_per_image ? (size_t)b0 * h->rows : 0), azimuths_per_image);
  };
  const int nwin = (n_scans + MAX_WINDOW - 1) / MAX_WINDOW;
  int st = window(0, h->windows);
  if (st == RSX_OK && nwin > 1) st = window(1, h->windows + 1);
  for (int w = 0; w < nwin && st == RSX_OK; w++) {
    const uint64_t g = h->windows + (uint64_t)w;
    const int b0 = w * MAX_WINDOW, n = n_scans - b0 < MAX_WINDOW ? n_scans - b0 : MAX_WINDOW;
    int first = 0;
    st = enqueue_match(h, g, n, &first);
    if (st == RSX_OK && w + 2 < nwin) st = window(w + 2, g + 2);  // two extractions run beside this window's matching
    if (st == RSX_OK) st = finish_window(h, g, n, first, out + b0, out_xy ? out_xy + (size_t)b0 * max_xy * 2 : nullptr, max_xy);
  }
  h->windows += (uint64_t)nwin;
  if (st != RSX_OK) abandon(h);
  return st;
} RSX_CATCH_ALL

int rsx_odometry_push(rsx_odometry *h, const uint8_t *imgs, int32_t n_scans, int64_t image_stride_bytes, int32_t row_stride,
                      const float *azimuths, int32_t azimuths_per_image, rsx_odometry_scan *out, float *out_xy, int32_t max_xy) try {
  if (!h || !imgs || !azimuths || !out || n_scans < 0 || max_xy < 0) return fail(RSX_ERR_BAD_ARG, "bad arg");
  if (n_scans == 0) return RSX_OK;
  RSX_TRY(check_layout(h, n_scans, image_stride_bytes, row_stride));
  RSX_TRY(check_azimuths(h, azimuths, azimuths_per_image, n_scans));
  std::lock_guard<std::mutex> lk(h->mu);
  RSX_HIP(hipSetDevice(h->device));
  const size_t ibytes = (size_t)h->rows * row_stride;
  RSX_TRY(reserve_all(h, ibytes, h->match_stream));
  // three image buffers: the upload of window w + 2 (copy stream) runs while the kernels of windows w and w + 1 do
  struct CopyDone {  // the caller's host images must not be in flight when this call returns, error or not
    hipStream_t cs;
    ~CopyDone() { (void)hipStreamSynchronize(cs); }
  } copy_done{h->copy_stream};
  auto upload = [&](int b0, int n, void *dst) -> int {
    const uint8_t *src = imgs + (int64_t)b0 * image_stride_bytes;
    if (n == 1 || image_stride_bytes == (int64_t)ibytes) {
      RSX_HIP(hipMemcpyAsync(dst, src, ibytes * n, hipMemcpyHostToDevice, h->copy_stream));
    } else {
      RSX_HIP(hipMemcpy2DAsync(dst, ibytes, src, (size_t)image_stride_bytes, ibytes, (size_t)n, hipMemcpyHostToDevice, h->copy_stream));
    }
    RSX_HIP(hipEventRecord(h->ev_up, h->copy_stream));
    return RSX_OK;
  };
  auto window = [&](int w, uint64_t g) -> int {  // E(g) of the call's window w, behind its upload
    const int b0 = w * MAX_WINDOW, n = n_scans - b0 < MAX_WINDOW ? n_scans - b0 : MAX_WINDOW;
    RSX_TRY(upload(b0, n, h->imgs[w % N_SETS].p));  // (the buffer was read by the extraction of window w - 3: long done)
    RSX_HIP(hipStreamWaitEvent(h->lane_stream[g & 1], h->ev_up, 0));
    return enqueue_extract(h, g, h->imgs[w % N_SETS].as<uint8_t>(), n, (int64_t)ibytes, row_stride,
                           azimuths + (azimuths_per_image ? (size_t)b0 * h->rows : 0), azimuths_per_image);
  };
  const int nwin = (n_scans + MAX_WINDOW - 1) / MAX_WINDOW;
  int st = window(0, h->windows);
  if (st == RSX_OK && nwin > 1) st = window(1, h->windows + 1);
  for (int w = 0; w < nwin && st == RSX_OK; w++) {
    const uint64_t g = h->windows + (uint64_t)w;
    const int b0 = w * MAX_WINDOW, n = n_scans - b0 < MAX_WINDOW ? n_scans - b0 : MAX_WINDOW;
    int first = 0;
    st = enqueue_match(h, g, n, &first);
    if (st == RSX_OK && w + 2 < nwin) st = window(w + 2, g + 2);
    if (st == RSX_OK) st = finish_window(h, g, n, first, out + b0, out_xy ? out_xy + (size_t)b0 * max_xy * 2 : nullptr, max_xy);
  }
  h->windows += (uint64_t)nwin;
  if (st != RSX_OK) abandon(h);
  return st;
} RSX_CATCH_ALL

int rsx_host_alloc_pinned(size_t bytes, void **out) try {
  if (!out || bytes == 0) return fail(RSX_ERR_BAD_ARG, "bad arg");
  *out = nullptr;
  if (rsx_device_count() <= 0) return fail(RSX_ERR_NO_DEVICE, "no HIP device visible (librsx has no CPU fallback)");
  RSX_HIP(hipHostMalloc(out, bytes, hipHostMallocDefault));
  return RSX_OK;
} RSX_CATCH_ALL

int rsx_host_free_pinned(void *p) try {
  if (p) RSX_HIP(hipHostFree(p));
  return RSX_OK;
} RSX_CATCH_ALL

}  // extern "C"
