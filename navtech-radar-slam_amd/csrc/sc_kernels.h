// sc_kernels.h -- launch wrappers of the ScanContext HIP kernels (sc_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "rsx.h"
#include "rsx_diag.h"

namespace rsx {
namespace sc {

constexpr int NR = RSX_SC_NUM_RING;
constexpr int NS = RSX_SC_NUM_SECTOR;
constexpr int DS = RSX_SC_DESC_SIZE;

// One database shard in HBM (SoA, see DESIGN.md "Data layout"):
struct DbView {
  const float *desc;   // [n_local][60][20] fp32, sector-major == Eigen col-major order
  const double *vkey;  // [n_local][60] sector keys (SC.cpp:214-227)
  const double *norm;  // [n_local][60] column norms (shift-invariant part of SC.cpp:78,81)
  const float *rkey;   // [n_local][20] ring keys as float (SC.cpp:198-211 + 62-66)
  const void *hnT;     // filter image: column-normalised fp16, tile-major (sc_filter.hip); 2400 B per entry
  const uint64_t *cmask;  // [n_local] bit j = column j has a non-zero norm; bit 63 = non-finite element
  const void *spT;     // spectral filter image: fp16 Z15 spectra, tile-major (sc_spec.hip); 2432 B per entry
  const float *sp_aux; // [n_local] sqrt of the spectral energy outside f = 0 (error budget of the spectral filter)
  const void *hnR;     // the fp16 filter image once more, entry-major [n_local][1200] (sc_window.hip gathers single entries)
  const void *vk16;    // [n_local][128] fp16: sector key scaled by a power of two, hi[64] | lo[64] (sc_window.hip)
  const float *vk_n;   // [n_local][2] l2 norm of the scaled key (NaN: no matrix-core alignment for this entry), of the key
  int64_t n_local;
  int64_t idx_base;    // global index of local slot s = idx_base + s * idx_stride
  int64_t idx_stride;
  int32_t sum_order;   // RSX_SC_SUM_*: the order of the sums the reference takes through Eigen (sc_redux_dev.h)
};

// Element of the bound matrix the filter hands to the short-list selection and the re-scoring: fp16, converted with
// round-toward-zero.  A bound v is only ever used as "skip the pair when v - eps > tau": a positive v rounded toward zero
// is still a lower bound, a negative v stays <= 0 <= every distance, +-inf and NaN survive.  Against fp32 this halves the
// matrix (8192 x 10 000: 0.33 -> 0.16 GB written by the filter and read twice by the selection per step) for < 2^-10
// relative looseness (eps is 1.25e-3).
using lb_t = _Float16;
typedef _Float16 lb2_t __attribute__((ext_vector_type(2)));
#ifdef __HIPCC__
__device__ __forceinline__ lb_t lb_pack(float v) {
  return __builtin_bit_cast(lb2_t, __builtin_amdgcn_cvt_pkrtz(v, 0.0f))[0];
}
#endif

struct QueryView {
  const float *desc;   // [nq][1200]
  const double *vkey;  // [nq][60]
  const double *norm;  // [nq][60]
  int32_t nq;
};

// keys for n descriptors (device pointers)
int launch_keys(const float *desc, int64_t n, double *vkey, double *norm, float *rkey, hipStream_t s, int sum_order);

// descriptor from one point cloud resident in device memory, written to out_desc (1200 floats)
// followed by its keys
int launch_build(const void *d_pts, int64_t n_pts, int64_t stride_bytes, double lidar_height,
                 double max_radius, float *out_desc, double *out_vkey, double *out_norm,
                 float *out_rkey, hipStream_t s, int sum_order);

// number of partial top-k slots per query the pair kernel will produce for this problem
int pair_num_slots(int64_t n_items, int32_t nq);
size_t pair_partial_bytes(int64_t n_items, int32_t nq, int32_t k);

// score nq queries against db entries.
//   gather == nullptr: local slots [first, first+n_items); else slots gather[0..n_items)
//   n_eligible: only entries with global index < min(n_eligible, q_elig[q] if given) are ranked
//   out_dist/out_shift (optional, [nq][n_items]): every pair, eligibility ignored
//   d_partial (optional): workspace of pair_partial_bytes(); d_topk [nq][k] receives the merged
//   result sorted by (dist, index), padded with {1e7,0,0}
int launch_pairs(const DbView &db, const QueryView &q, const int32_t *gather, int64_t first,
                 int64_t n_items, int64_t n_eligible, const int64_t *q_elig, double *out_dist,
                 int32_t *out_shift, rsx_sc_hit *d_partial, rsx_sc_hit *d_topk, int32_t k,
                 hipStream_t s);

// one launch per new keyframe: descriptor + keys from the cloud AND every per-entry image of the query path (sc_spec.hip
// sc_insert_kernel).  n_clouds clouds: points [d_offs[b], d_offs[b + 1]) of d_pts go to local slot first_slot + b
// (d_offs = nullptr: one cloud of n_pts points)
int launch_insert(const void *d_pts, const int64_t *d_offs, int64_t n_pts, int64_t n_clouds, int64_t stride_bytes,
                  double lidar_height, double max_radius, int64_t first_slot, float *desc, double *vkey, double *norm,
                  float *rkey, void *hnT, void *hnR, uint64_t *cmask, void *spT, float *aux, void *vk16, float *vk_n,
                  hipStream_t s, int sum_order);

// merge [nparts][nq][k] -> [nq][k]
int launch_merge(const rsx_sc_hit *d_parts, int32_t nparts, int32_t nq, int32_t k, rsx_sc_hit *d_out,
                 hipStream_t s);

// d_lb[q][c] (ld elements per row, nq rows) = d_blocks[c / block_ld][q0 + q][c % block_ld]: the column blocks the filter
// shards of a replicated DB deliver (block b starts block_stride elements after block b - 1) as one bound matrix
int launch_gather_bounds(const lb_t *d_blocks, int64_t block_ld, int64_t block_stride, int64_t q0, int32_t nq, lb_t *d_lb,
                         int64_t ld, hipStream_t s);

// exact k-NN over ring keys [0, n_search) with nanoflann's float L2 (NF.hpp:383-408);
// d_dist_ws: n_search floats of workspace; out_idx[k] (unfilled = 0), out_found[1]
int launch_knn(const float *rkeys, int64_t n_search, const float *qkey, int32_t k, float *d_dist_ws,
               int32_t *out_idx, float *out_dist, int32_t *out_found, hipStream_t s);

// ---- exact re-scoring behind the filter ----
constexpr int RESCORE_SHORTLIST_CAP = 2048;  // short-list records per query
constexpr int RESCORE_NUM_THR = 6;           // round edges t_0..t_4 and t_cap
// per query: RESCORE_NUM_THR float edges, then RESCORE_NUM_THR int32 counts (short-list entries below each edge:
// the list is ordered by bin, so round r is the range [count[r-1], count[r]))
constexpr int RESCORE_THR_STRIDE = 2 * RESCORE_NUM_THR;
// the re-scoring kernels' bookkeeping (profiling only): RESCORE_STAT_COPIES blocks of RESCORE_STAT_WORDS u64 counters; a query
// adds to block (query index % copies) -- one shared block made 8192 waves queue up on a single cache line of atomics (measured:
// +0.15-0.23 ms on a 3.4 ms step); the host sums the blocks
constexpr int RESCORE_STAT_WORDS = 16, RESCORE_STAT_COPIES = 64;
struct WindowPreview;
struct RescoreEntry {
  float lb;      // filter bound
  int32_t slot;  // local DB slot
};
// short list of every query: the eligible entries with bound < t_cap, where t_cap is the largest
// histogram-bin edge with at most RESCORE_SHORTLIST_CAP bounds below it (+inf when all fit, -inf
// when not even the first bin fits), and the round edges (first_target << r bounds, r = 0..4)
// first_target: bounds in round 0; round r covers first_target << r
int launch_select(const DbView &db, const lb_t *lb, int64_t ld_lb, int64_t n_items, int32_t nq, int64_t n_eligible,
                  const int64_t *q_elig, int32_t first_target, RescoreEntry *slist, int32_t *sl_cnt, float *thr,
                  hipStream_t s);
// one workgroup per query: rounds [round_begin, round_end) of ascending bound with tau tightening
// (round_end = RESCORE_ALL_ROUNDS: also the entries beyond the short list); writes the top-k it knows.
// tau_src (optional): a top-k that covers more than this shard -- its k-th distance caps tau;
// seed (optional): this shard's hits from an earlier stage, merged into the output.
constexpr int RESCORE_ALL_ROUNDS = RESCORE_NUM_THR + 1;
int launch_rescore(const DbView &db, const QueryView &q, const lb_t *lb, int64_t ld_lb, int64_t n_items,
                   int64_t n_eligible, const int64_t *q_elig, const RescoreEntry *slist, const int32_t *sl_cnt,
                   const float *thr, double eps, int32_t round_begin, int32_t round_end, const rsx_sc_hit *tau_src,
                   const rsx_sc_hit *seed, rsx_sc_hit *d_out, int32_t k, hipStream_t s,
                   unsigned long long *d_stats = nullptr, const WindowPreview *win = nullptr);

// the same job by one wave per query walking the short list in ascending-bound order (needs the short list
// ordered by histogram bin, which launch_select produces); single-shard, single-stage
int launch_walk(const DbView &db, const QueryView &q, const lb_t *lb, int64_t ld_lb, int64_t n_items,
                int64_t n_eligible, const int64_t *q_elig, const RescoreEntry *slist, const int32_t *sl_cnt,
                const float *thr, double eps, rsx_sc_hit *d_out, int32_t k, hipStream_t s);

// ---- MFMA lower-bound filter (sc_filter.hip) ----
constexpr int FILTER_QIMG_BYTES = 9984;  // LDS image of one query (two displaced fp16 copies)
constexpr int FILTER_QIMG_ODD = 4960;    // byte offset of the copy read by odd shifts (holds q2[4..]); = 96 mod 256
constexpr int FILTER_QIMG_MASK_OFF = 4928;  // the query's column mask (u64) sits in the gap between the copies
constexpr float FILTER_ACC_SCALE = 1073741824.0f;  // 2^30: both fp16 images are scaled by 2^15
constexpr int FILTER_DB_BYTES_PER_ENTRY = 2 * DS;
double filter_eps();
size_t filter_qimg_bytes(int32_t nq);
// fp16 filter images of local slots [first, first+count) (needs their norms)
int launch_db_images(const float *desc, const double *norm, int64_t first, int64_t count, void *hnT, void *hnR,
                     uint64_t *cmask, hipStream_t s);
int launch_query_images(const float *desc, const double *norm, int32_t nq, void *qimg, hipStream_t s);
// lb[q*ld_lb + slot] = lower bound of dist(query q, local slot), slots [0, n_items); +inf when no
// shift has an effective column (never a hit), -inf when the pair must be re-scored regardless
// plan (optional, with plan_ws of filter_plan_bytes(n_items) bytes): the queries' eligibility limits
// min(n_eligible, q_elig[q]) do not decrease with q, so the (tile-block, query) pairs a query cannot see
// are skipped altogether (their lb entries stay unwritten; they lie outside the query's eligible rows)
struct FilterPlanInput {
  int64_t n_eligible;
  const int64_t *q_elig;
};
size_t filter_plan_bytes(int64_t n_items);
// the plan itself (device): tb_qmin in units of `qgroup` queries (1: direct filter, 4: spectral filter's query
// tiles), tb_cum = exclusive prefix sums of ceil(nq / qgroup) - tb_qmin; both inside plan_ws
int launch_filter_plan(const DbView &db, const FilterPlanInput &plan, int32_t nq, int64_t n_items, int32_t qgroup,
                       void *plan_ws, const int32_t **tb_qmin, const int64_t **tb_cum, hipStream_t s);
int launch_filter(const DbView &db, const void *qimg, int32_t nq, int64_t n_items, lb_t *lb, int64_t ld_lb,
                  const FilterPlanInput *plan, void *plan_ws, hipStream_t s);
const char *pair_kernel_name();
const char *filter_kernel_name();

// ---- spectral form of the filter (sc_spec.hip): same bounds contract, ~7x fewer MFMAs ----
constexpr int SPEC_QIMG_BYTES = 4736;       // stream image of a query (sc_spec.hip); the mask image and flag byte follow the images
constexpr int SPEC_DB_BYTES_PER_ENTRY = 2432;
size_t spec_qimg_bytes(int32_t nq);
int launch_spec_db_images(const float *desc, const double *norm, int64_t first, int64_t count, void *spT, float *aux,
                          hipStream_t s);
int launch_spec_query_images(const float *desc, const double *norm, int32_t nq, void *qimg, hipStream_t s);
// two_waves: sc_spec2_filter_kernel (two waves per SIMD, the entry tile split by frequency) instead of sc_spec_filter_kernel;
// same images, same bounds bit for bit
int launch_spec_filter(const DbView &db, const void *qimg, int32_t nq, int64_t n_items, lb_t *lb, int64_t ld_lb,
                       const int32_t *tb_qmin, const int64_t *tb_cum, hipStream_t s, bool two_waves);
const char *spec_filter_kernel_name();
const char *spec2_filter_kernel_name();

// ---- window previews of the short lists on the matrix cores (sc_window.hip) ----
// For the first WINDOW_P short-list entries of every query: the sector-key alignment k* (SC.cpp:93-113) from an fp16
// hi/lo-split correlation (unique within its error bound, else "no preview") and the fp16 direct-form correlation of
// the two images evaluated over the window k* +- 3 only: pv with |pv - dist(query, entry)| <= WINDOW_MARGIN.  The
// re-scoring kernel uses (k*, pv) instead of its own VALU alignment + fp32 preview.
constexpr int WINDOW_P = RSX_SC_WINDOW_P;               // short-list positions per query that can get a record
constexpr int WINDOW_HEAD = 128;                         // the head of the list: always processed (pass 1)
constexpr float WINDOW_MARGIN = RSX_SC_WINDOW_MARGIN;    // same arithmetic as the direct filter: its error budget (sc_filter.hip)
constexpr int WINDOW_QK_BYTES = 4624;                    // key image of a query (sc_window.hip)
static_assert(WINDOW_P % 64 == 0 && WINDOW_HEAD == 128 && WINDOW_P > WINDOW_HEAD, "layout");
// Positions WINDOW_HEAD .. WINDOW_P - 1 are processed (pass 2) only when their filter bound does not exceed the k-th
// smallest upper bound the head yields; the others get the record {NaN, -2} = "none".
struct WindowPreview {
  float pv;    // NaN: no preview (non-finite data, or no record); +inf: no effective column in the window(s)
  int32_t ks;  // >= 0: k* in bits 0..5 and, in bits 8..14, which of the window shifts k* - 3 + t can be the minimum at all
               // (the others are more than 2 margins above the best preview); -1: not unique within the error bound -- pv is
               // then a lower bound only (union of the windows); -2: no record
};
size_t window_qimg_bytes(int32_t nq);  // direct-filter images + key images of a query batch
int launch_window_db_keys(const double *vkey, int64_t first, int64_t count, void *vk16, float *vk_n, hipStream_t s);
// qimg: window_qimg_bytes(nq) of workspace (filled here); out: [nq][WINDOW_P]
// k: the top-k the query batch asks for; eps: the filter's error budget (filter_eps())
// head_only > 0: records for the first head_only list positions (rounded up to 32) only, "no record" behind them -- stage 1
// of a DB shard scores only that many entries per query, and S shards previewing 128 each is S times the work of one GPU
int launch_window(const DbView &db, const QueryView &q, void *qimg, const RescoreEntry *slist, const int32_t *sl_cnt,
                  int32_t k, double eps, WindowPreview *out, hipStream_t s, int32_t head_only = 0);
// stage 2 of a DB shard after a head-only stage 1: records for the list positions behind the head whose bound can still reach
// the k-th best distance of d_global [nq][k] (the merged stage-1 lists); qimg still holds stage 1's query images
int launch_window_tail(const DbView &db, int32_t nq, void *qimg, const RescoreEntry *slist, const int32_t *sl_cnt, int32_t k, double eps,
                       WindowPreview *out, int32_t head, const rsx_sc_hit *d_global, hipStream_t s);
const char *window_kernel_name();

// ---- one query in one launch (sc_q1.hip): the live detector's regime, nq <= Q1_MAX_NQ ----
// Streams the fp16 image + key image of every eligible entry once (2672 B per entry), previews every pair on the matrix cores
// (alignment + window, as sc_window.hip does for the heads of the short lists) and scores the few entries the previews cannot
// exclude exactly, all inside one launch; records identical to launch_pairs / the filtered path.
// ws: q1_workspace_bytes() of device memory; d_ticket: q1_ticket_bytes() of arrival counters, zero before the first launch
// (every launch leaves them zero again) and touched by nothing else
constexpr int Q1_MAX_NQ = 16;
int q1_grid(int64_t n_items, int32_t k, int32_t nq);
size_t q1_workspace_bytes(int64_t n_items, int32_t nq, int32_t k);
size_t q1_ticket_bytes();
int launch_q1(const DbView &db, const float *d_q, int32_t nq, int64_t n_items, int64_t n_eligible, const int64_t *q_elig,
              int32_t k, rsx_sc_hit *d_out, void *ws, unsigned *d_ticket, unsigned long long *d_stats, hipStream_t s);
const char *q1_kernel_name();

// ---- the reference's public helper methods on arbitrary double descriptors (sc_helpers.hip) ----
// op 0: keys of d_a (out_d = ring key [20] + sector key [60]); 1: distDirectSC(d_a, d_b) -> out_d[0];
// 2: fastAlignUsingVkey(d_a, d_b) (60-element keys) -> out_i[0]; 3: distanceBtnScanContext -> out_d[0], out_i[0]
int launch_helper(int op, const double *d_a, const double *d_b, double *d_out_d, int32_t *d_out_i, hipStream_t s, int sum_order);

// optional hipEvent bracket around the dominant (pair) kernel
struct PairProfiler {
  bool on = false;
  static constexpr int kMax = 4096;
  hipEvent_t *ev = nullptr;  // 2*kMax events, created lazily
  int used = 0;
};
void set_pair_profiler(PairProfiler *p);  // thread-local hook consulted by launch_pairs

}  // namespace sc
}  // namespace rsx
