// sc_api.cpp -- C-ABI of the ScanContext path (include/rsx.h): database shard management and the
// reference's detector state machine (Scancontext.cpp:236-422) on top of the HIP kernels.
// Host logic only; every descriptor/key/distance is computed on the GPU.  There is no CPU fallback.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "rsx_common.h"
#include "sc_kernels.h"
#include "sc_plan.h"
#include "sc_kdtree.h"
#include "loopverify.h"
#include "voxelgrid.h"

namespace rsx {
std::string &last_error() {
  static thread_local std::string e;
  return e;
}
}  // namespace rsx

using namespace rsx;
using namespace rsx::sc;

struct rsx_sc {
  rsx_sc_params p;
  std::mutex mu;
  hipStream_t stream = nullptr;
  // database shard (SoA in HBM)
  int64_t n_global = 0, n_local = 0, cap = 0;
  DevBuf desc, vkey, norm, rkey;
  DevBuf hn, cmask;  // fp16 filter image (tile-major) + column masks (sc_filter.hip)
  DevBuf sp, sp_aux; // fp16 spectral filter image + per-entry error-budget scalar (sc_spec.hip)
  DevBuf hnr, vk16, vk_n;  // entry-major fp16 image + fp16 hi/lo sector keys and their norms (sc_window.hip)
  // detector state (SC.h:104,117-120)
  int tree_counter = 0;
  int64_t tree_size = 0;
  bool batch_made = false;
  int64_t batch_size = 0;
  // the ring-key search trees behind them (sc_kdtree.h): over the frozen prefix / over the whole DB of the first
  // between-session call.  Entries are append-only, so a tree is a function of its size and is (re)built lazily
  struct KdTreeDev {
    DevBuf nodes, vind;
    float low[KD_DIM], high[KD_DIM];
    int64_t n = 0;  // 0: not built
    int32_t n_nodes = 0;
    int depth = 0;
  } tree, tree_batch;
  // workspaces
  DevBuf pts_ws, q_desc, topk, knn_ws, small, pair_out, q_elig;
  // what one query batch in flight owns between its keys and its records.  Two sets: the host-buffer entry scores the
  // pieces of a large batch alternately on two streams, each with its own set (rsx_sc_query); everything else uses set 0
  struct QueryWs {
    DevBuf q_vkey, q_norm, q_rkey, partial;
    DevBuf f_qimg, f_lb, f_cand, f_cnt, f_thr, f_plan;  // filter path
    DevBuf f_wimg, f_win;                               // window previews of the short lists (sc_window.hip)
    DevBuf *all[12] = {&q_vkey, &q_norm, &q_rkey, &partial, &f_qimg, &f_lb, &f_cand, &f_cnt, &f_thr, &f_plan, &f_wimg, &f_win};
  } ws[2];
  QueryWs *w = &ws[0];  // the set the calls below work in (guarded by mu like the rest)
  PairProfiler prof;
  const char *prof_kernel = "sc_pair_kernel";  // which kernel the profiler events bracket
  // state between rsx_sc_query_stage1_device and rsx_sc_query_stage2_device
  struct {
    bool valid = false, filtered = false;
    int32_t nq = 0, k = 0;
    int32_t window_head = 0;  // > 0: stage 1 left window records for this many list positions only; stage 2 adds the rest on demand
    int64_t n_items = 0, n_eligible = 0;
    const int64_t *q_elig = nullptr;
    QueryView qv{};
  } st;
  DevBuf st_partial;  // this shard's stage-1 hits
  // the one-launch single-query path (sc_q1.hip): candidate records / bound lists of the workgroups, and the arrival counters
  DevBuf q1_ws, q1_ticket;
  DevBuf helper_ws;   // staging of the stateless helper calls (Scancontext.h:60-66)
  DevBuf stats;       // profiling only: RESCORE_STAT_COPIES blocks of counters (sc_kernels.h), summed by the host when read
  void *pinned = nullptr;  // small pinned host staging (results)
  size_t pinned_bytes = 0;
  // upload pipeline of the host-buffer entry (rsx_sc_query): pieces of the query batch go up on their own stream while
  // the pieces before them are scored
  static constexpr int kMaxPieces = 8;
  // staging ring of rsx_sc_add_points: the cloud is copied into pinned memory, goes up asynchronously and one kernel builds
  // the entry (launch_insert); the host does not wait.  Entries appear in the order of h->stream; a call that brings its own
  // stream is ordered behind the latest insert (use_stream)
  static constexpr int kInsSlots = 8;
  struct InsSlot {
    void *host = nullptr;
    size_t cap = 0;
    hipEvent_t done = nullptr;
    bool used = false;
  } ins[kInsSlots];
  int ins_next = 0;
  hipEvent_t last_insert = nullptr;
  // the handle's workspaces (and the single-query path's arrival tickets, which must be zero between launches) are shared by
  // every call: when a call arrives on another stream than the previous one, the new stream is ordered behind the old
  hipStream_t last_user_stream = nullptr;
  hipEvent_t stream_switch = nullptr;
  hipStream_t up_stream = nullptr, stream_b = nullptr;
  hipEvent_t up_ev[kMaxPieces] = {}, lane_ev = nullptr;
};

namespace {

constexpr size_t kStatBytes = (size_t)RESCORE_STAT_COPIES * RESCORE_STAT_WORDS * sizeof(unsigned long long);

int set_device(rsx_sc *h) {
  RSX_HIP(hipSetDevice(h->p.device));
  return RSX_OK;
}

// the stream a device entry works on: the caller's, ordered behind the inserts still in flight on the handle's own
int use_stream(rsx_sc *h, void *stream, hipStream_t *s) {
  *s = stream ? static_cast<hipStream_t>(stream) : h->stream;
  if (*s != h->stream && h->last_insert) RSX_HIP(hipStreamWaitEvent(*s, h->last_insert, 0));
  if (h->last_user_stream && h->last_user_stream != *s) {
    if (!h->stream_switch) RSX_HIP(hipEventCreateWithFlags(&h->stream_switch, hipEventDisableTiming));
    // (a previous stream the caller has destroyed in the meantime has drained: nothing to wait for)
    if (hipEventRecord(h->stream_switch, h->last_user_stream) == hipSuccess) RSX_HIP(hipStreamWaitEvent(*s, h->stream_switch, 0));
    else (void)hipGetLastError();
  }
  h->last_user_stream = *s;
  return RSX_OK;
}

// one new entry from a cloud in device memory, all of it in one launch
int insert_cloud(rsx_sc *h, const void *d_pts, int64_t n_pts, int64_t stride, int64_t slot, hipStream_t s) {
  return launch_insert(d_pts, nullptr, n_pts, 1, stride, h->p.lidar_height, h->p.max_radius, slot, h->desc.as<float>(),
                       h->vkey.as<double>(), h->norm.as<double>(), h->rkey.as<float>(), h->hn.p, h->hnr.p,
                       h->cmask.as<uint64_t>(), h->sp.p, h->sp_aux.as<float>(), h->vk16.p, h->vk_n.as<float>(), s, h->p.sum_order);
}

int ensure_capacity(rsx_sc *h, int64_t want_local) {
  if (want_local <= h->cap) return RSX_OK;
  int64_t nc = h->cap ? h->cap : 1024;
  while (nc < want_local) nc *= 2;
  RSX_TRY(h->desc.reserve((size_t)nc * DS * sizeof(float), h->stream, true));
  RSX_TRY(h->vkey.reserve((size_t)nc * NS * sizeof(double), h->stream, true));
  RSX_TRY(h->norm.reserve((size_t)nc * NS * sizeof(double), h->stream, true));
  RSX_TRY(h->rkey.reserve((size_t)nc * NR * sizeof(float), h->stream, true));
  RSX_TRY(h->hn.reserve((size_t)nc * FILTER_DB_BYTES_PER_ENTRY, h->stream, true));  // nc is a multiple of 32
  RSX_TRY(h->cmask.reserve((size_t)nc * sizeof(uint64_t), h->stream, true));
  RSX_TRY(h->sp.reserve((size_t)nc * SPEC_DB_BYTES_PER_ENTRY, h->stream, true));
  RSX_TRY(h->sp_aux.reserve((size_t)nc * sizeof(float), h->stream, true));
  RSX_TRY(h->hnr.reserve((size_t)nc * FILTER_DB_BYTES_PER_ENTRY, h->stream, true));
  RSX_TRY(h->vk16.reserve((size_t)nc * 256, h->stream, true));
  RSX_TRY(h->vk_n.reserve((size_t)nc * 2 * sizeof(float), h->stream, true));
  h->cap = nc;
  return RSX_OK;
}

DbView db_view(const rsx_sc *h) {
  DbView v;
  v.desc = h->desc.as<float>();
  v.vkey = h->vkey.as<double>();
  v.norm = h->norm.as<double>();
  v.rkey = h->rkey.as<float>();
  v.hnT = h->hn.p;
  v.cmask = h->cmask.as<uint64_t>();
  v.spT = h->sp.p;
  v.sp_aux = h->sp_aux.as<float>();
  v.hnR = h->hnr.p;
  v.vk16 = h->vk16.p;
  v.vk_n = h->vk_n.as<float>();
  v.n_local = h->n_local;
  v.idx_base = h->p.shard_rank;
  v.idx_stride = h->p.shard_world;
  v.sum_order = h->p.sum_order;
  return v;
}

// both filter images + column masks of local slots [slot, slot+count); synchronises s
int build_db_images(rsx_sc *h, int64_t slot, int64_t count, hipStream_t s) {
  RSX_TRY(launch_db_images(h->desc.as<float>(), h->norm.as<double>(), slot, count, h->hn.p, h->hnr.p, h->cmask.as<uint64_t>(), s));
  RSX_TRY(launch_window_db_keys(h->vkey.as<double>(), slot, count, h->vk16.p, h->vk_n.as<float>(), s));
  RSX_TRY(launch_spec_db_images(h->desc.as<float>(), h->norm.as<double>(), slot, count, h->sp.p, h->sp_aux.as<float>(), s));
  RSX_HIP(hipStreamSynchronize(s));
  return RSX_OK;
}

bool owns(const rsx_sc *h, int64_t g) { return (g % h->p.shard_world) == h->p.shard_rank; }

// number of local slots whose global index is < n_eligible
int64_t local_count_below(const rsx_sc *h, int64_t n_eligible) {
  if (n_eligible < 0 || n_eligible > h->n_global) n_eligible = h->n_global;
  const int64_t w = h->p.shard_world, r = h->p.shard_rank;
  if (n_eligible <= r) return 0;
  int64_t c = (n_eligible - r + w - 1) / w;
  return c < h->n_local ? c : h->n_local;
}

// SC.cpp:17-20,417: deg2rad(float) of nn_align * PC_UNIT_SECTORANGLE
float yaw_from_shift(int shift) {
  const double unit = 360.0 / (double)NS;  // SC.h:88
  const float deg = (float)(shift * unit);
  return (float)((double)deg * M_PI / 180.0);
}

int ensure_pinned(rsx_sc *h, size_t bytes) {
  if (bytes <= h->pinned_bytes) return RSX_OK;
  if (h->pinned) (void)hipHostFree(h->pinned);
  h->pinned = nullptr;
  h->pinned_bytes = 0;
  size_t nb = 4096;
  while (nb < bytes) nb *= 2;
  RSX_HIP(hipHostMalloc(&h->pinned, nb, hipHostMallocDefault));
  h->pinned_bytes = nb;
  return RSX_OK;
}

// keys for nq query descriptors already in h->q_desc (or external device pointer)
int prepare_queries(rsx_sc *h, const float *d_q, int32_t nq, hipStream_t s, QueryView *qv) {
  h->st.valid = false;  // q_vkey / q_norm are about to be overwritten: a pending stage 2 would read the wrong keys
  RSX_TRY(h->w->q_vkey.reserve((size_t)nq * NS * sizeof(double), s, false));
  RSX_TRY(h->w->q_norm.reserve((size_t)nq * NS * sizeof(double), s, false));
  RSX_TRY(h->w->q_rkey.reserve((size_t)nq * NR * sizeof(float), s, false));
  RSX_TRY(launch_keys(d_q, nq, h->w->q_vkey.as<double>(), h->w->q_norm.as<double>(), h->w->q_rkey.as<float>(), s, h->p.sum_order));
  qv->desc = d_q;
  qv->vkey = h->w->q_vkey.as<double>();
  qv->norm = h->w->q_norm.as<double>();
  qv->nq = nq;
  return RSX_OK;
}

// rsx_sc_params.filter_mode (1 off, 2 force) decides; when it says 0 = auto, RSX_SC_FILTER=0/1 may
// force it off / on (profiling scripts).  Same precedence as filter_kind_of: explicit params first.
int filter_mode_of(const rsx_sc *h) {
  static const int env = [] {
    const char *e = rsx::exp_env("RSX_SC_FILTER");
    if (!e || !*e) return -1;
    return atoi(e) ? 2 : 1;
  }();
  if (h->p.filter_mode) return h->p.filter_mode;
  return env >= 0 ? env : 0;
}

bool use_filter(const rsx_sc *h, int32_t nq, int64_t n_items) {
  const int m = filter_mode_of(h);  // (3 = the single-query path where it applies: like auto for everything else)
  if (m == 1 || n_items <= 0) return false;
  if (m == 2) return true;
  // the filter amortises a 300-register DB tile load over the queries of a block and costs five
  // extra launches: worth it for batched queries against a sizeable DB -- and for ANY number of queries once the DB is
  // so large that scoring every entry exactly costs more than the launches (one query, MI355X, us per query
  // exact-all / filtered: 1 000 keyframes 22 / 85, 10 000: 55 / 94, 100 000: 232 / 178)
  // (a handful of queries never gets here in auto mode unless the single-query path declined them: use_q1)
  if (n_items >= 50000) return true;
  return (int64_t)nq * n_items >= (1ll << 17);  // exact-all scores ~1 G pairs/s, the chain's fixed cost is ~85 us
}

struct ProfScope {
  PairProfiler *pp;
  hipStream_t s;
  bool armed = false;
  ProfScope(PairProfiler *p, hipStream_t st) : pp(p), s(st) {
    if (pp && pp->on && pp->ev && pp->used < PairProfiler::kMax) armed = hipEventRecord(pp->ev[2 * pp->used], s) == hipSuccess;
  }
  void stop() {
    if (armed && hipEventRecord(pp->ev[2 * pp->used + 1], s) == hipSuccess) pp->used++;
    armed = false;
  }
};

// 2 = the spectral form with two waves per SIMD (sc_spec2_filter_kernel: same images, same bounds bit for bit; the default
// since round 4: 1.76 against 2.05 ms per 8192 x 9970 launch on the same box)
constexpr int kDefaultFilterKind = 2;
// which form of the lower-bound filter: 0 = direct (sc_filter.hip), 1 = spectral (sc_spec.hip, the default:
// same bounds up to a slightly larger error budget, ~6x fewer MFMAs).  rsx_sc_params.filter_kind or
// RSX_SC_FILTER_KIND=direct|spectral override.
int filter_kind_of(const rsx_sc *h) {
  static const int env = [] {
    const char *e = rsx::exp_env("RSX_SC_FILTER_KIND");
    if (!e || !*e) return -1;
    if (e[0] == '2' || (e[0] == 's' && std::strstr(e, "2"))) return 2;  // spectral2 / spec2
    return (e[0] == 's' || e[0] == '1') ? 1 : 0;
  }();
  if (h->p.filter_kind) return h->p.filter_kind - 1;
  return env >= 0 ? env : kDefaultFilterKind;
}

size_t any_qimg_bytes(int32_t nq) {
  const size_t a = filter_qimg_bytes(nq), b = spec_qimg_bytes(nq);
  return a > b ? a : b;
}

// query images + bounds of one batch with the chosen filter form
// first (a multiple of 32, the images are tile-major): the bounds of local slots [first, first + n_items), lb[q * ld + slot - first]
int run_filter(rsx_sc *h, const QueryView &q, int64_t n_items, lb_t *lb, int64_t ld, const FilterPlanInput *plan,
               hipStream_t s, int64_t first = 0) {
  DbView db = db_view(h);
  if (first) {
    db.hnT = static_cast<const char *>(db.hnT) + first * FILTER_DB_BYTES_PER_ENTRY;
    db.spT = static_cast<const char *>(db.spT) + first * SPEC_DB_BYTES_PER_ENTRY;
    db.cmask += first;
    db.sp_aux += first;
  }
  if (plan) RSX_TRY(h->w->f_plan.reserve(filter_plan_bytes(n_items), s, false));
  if (filter_kind_of(h) >= 1) {
    const bool two_waves = filter_kind_of(h) == 2;
    h->prof_kernel = two_waves ? spec2_filter_kernel_name() : spec_filter_kernel_name();
    RSX_TRY(launch_spec_query_images(q.desc, q.norm, q.nq, h->w->f_qimg.p, s));
    const int32_t *qmin = nullptr;
    const int64_t *cum = nullptr;
    if (plan) RSX_TRY(launch_filter_plan(db, *plan, q.nq, n_items, 4, h->w->f_plan.p, &qmin, &cum, s));
    ProfScope ps(&h->prof, s);
    RSX_TRY(launch_spec_filter(db, h->w->f_qimg.p, q.nq, n_items, lb, ld, qmin, cum, s, two_waves));
    ps.stop();
    return RSX_OK;
  }
  h->prof_kernel = filter_kernel_name();
  RSX_TRY(launch_query_images(q.desc, q.norm, q.nq, h->w->f_qimg.p, s));
  ProfScope ps(&h->prof, s);
  RSX_TRY(launch_filter(db, h->w->f_qimg.p, q.nq, n_items, lb, ld, plan, plan ? h->w->f_plan.p : nullptr, s));
  ps.stop();
  return RSX_OK;
}

// query batch size the filter workspaces are sized for: <= 1 GiB of (fp16) bounds, and the batches of a call equally long
// (8192 queries against 100 000 entries used to run as 3 x 2684 + 140: the last launch chain at a fraction of the rate)
int64_t filter_batch(int64_t n_items, int64_t nq) { return plan::filter_batch(n_items, nq); }  // sc_plan.h

int filter_reserve(rsx_sc *h, int64_t n_items, int64_t qb, hipStream_t s) {
  h->st.valid = false;  // bounds / short lists of a pending stage 2 are about to be overwritten
  const int64_t ld = (n_items + 31) / 32 * 32;
  RSX_TRY(h->w->f_qimg.reserve(any_qimg_bytes((int32_t)qb), s, false));
  RSX_TRY(h->w->f_lb.reserve((size_t)qb * ld * sizeof(lb_t), s, false));
  RSX_TRY(h->w->f_cand.reserve((size_t)qb * RESCORE_SHORTLIST_CAP * sizeof(RescoreEntry), s, false));
  RSX_TRY(h->w->f_cnt.reserve((size_t)qb * sizeof(int32_t), s, false));
  RSX_TRY(h->w->f_thr.reserve((size_t)qb * RESCORE_THR_STRIDE * sizeof(float), s, false));
  RSX_TRY(h->w->f_wimg.reserve(window_qimg_bytes((int32_t)qb), s, false));
  RSX_TRY(h->w->f_win.reserve((size_t)qb * WINDOW_P * sizeof(WindowPreview), s, false));
  return RSX_OK;
}

// RSX_SC_WINDOW=0 (experiments build): re-scoring without the matrix-core window previews
bool use_window() {
  static const bool on = [] {
    const char *e = rsx::exp_env("RSX_SC_WINDOW");
    return !(e && e[0] == '0');
  }();
  return on;
}

// images -> MFMA filter -> short list + round edges of one query batch
// elig_monotone: the per-query limits elig[] do not decrease with the query index (self queries)
int filter_and_select(rsx_sc *h, const QueryView &q, int64_t n_items, int64_t n_eligible, const int64_t *elig,
                      int32_t first_target, int32_t k, hipStream_t s, bool elig_monotone = false, int32_t window_head_only = 0) {
  const DbView db = db_view(h);
  const int64_t ld = (n_items + 31) / 32 * 32;
  lb_t *lb = h->w->f_lb.as<lb_t>();
  {
    FilterPlanInput plan{n_eligible, elig};
    const bool planned = elig_monotone && elig != nullptr;
    RSX_TRY(run_filter(h, q, n_items, lb, ld, planned ? &plan : nullptr, s));
  }
  RSX_TRY(launch_select(db, lb, ld, n_items, q.nq, n_eligible, elig, first_target, h->w->f_cand.as<RescoreEntry>(),
                        h->w->f_cnt.as<int32_t>(), h->w->f_thr.as<float>(), s));
  // alignment + window preview of the head of every short list on the matrix cores (what re-scoring would otherwise
  // do on the VALU, one entry per wavefront)
  if (!use_window()) return RSX_OK;
  return launch_window(db, q, h->w->f_wimg.p, h->w->f_cand.as<RescoreEntry>(), h->w->f_cnt.as<int32_t>(), k, filter_eps(),
                       h->w->f_win.as<WindowPreview>(), s, window_head_only);
}

int rescore(rsx_sc *h, const QueryView &q, int64_t n_items, int64_t n_eligible, const int64_t *elig, int32_t round_begin,
            int32_t round_end, const rsx_sc_hit *tau_src, const rsx_sc_hit *seed, int32_t k, rsx_sc_hit *d_out,
            hipStream_t s) {
  const int64_t ld = (n_items + 31) / 32 * 32;
  return launch_rescore(db_view(h), q, h->w->f_lb.as<lb_t>(), ld, n_items, n_eligible, elig, h->w->f_cand.as<RescoreEntry>(),
                        h->w->f_cnt.as<int32_t>(), h->w->f_thr.as<float>(), filter_eps(), round_begin, round_end, tau_src,
                        seed, d_out, k, s, (h->prof.on && h->stats.p) ? h->stats.as<unsigned long long>() : nullptr,
                        use_window() ? h->w->f_win.as<WindowPreview>() : nullptr);
}

int32_t first_round_target() {
  static const int32_t v = [] {
    const char *e = rsx::exp_env("RSX_SC_FIRST_TARGET");
    const int x = e ? atoi(e) : 0;
    return (x >= 1 && x <= 128) ? x : 128;
  }();
  return v;
}

// exhaustive top-k through the MFMA lower-bound filter (sc_filter.hip): filter -> short list ->
// exact re-scoring in rounds of ascending bound.  Everything stays on the stream; no host sync.
int run_topk_filtered(rsx_sc *h, const QueryView &qv, int64_t n_items, int64_t n_eligible, const int64_t *d_q_elig,
                      int32_t k, rsx_sc_hit *d_out, hipStream_t s, bool elig_monotone) {
  const int64_t qb = filter_batch(n_items, qv.nq);
  RSX_TRY(filter_reserve(h, n_items, qb, s));
  for (int64_t b0 = 0; b0 < qv.nq; b0 += qb) {
    const int32_t bn = (int32_t)((qv.nq - b0 < qb) ? (qv.nq - b0) : qb);
    QueryView q = qv;
    q.desc = qv.desc + b0 * DS;
    q.vkey = qv.vkey + b0 * NS;
    q.norm = qv.norm + b0 * NS;
    q.nq = bn;
    const int64_t *elig = d_q_elig ? d_q_elig + b0 : nullptr;
    // size of the first re-scoring round; later rounds double.  With the two-phase scoring (every candidate of a round
    // gets the cheap alignment + fp32 preview, only the few the previews cannot exclude are evaluated exactly) a round
    // costs more in barriers than in arithmetic, so the first one is large: measured on MI355X (10k trajectory DB, 8192
    // queries, ms per step / exact evaluations per query): 64 -> 4.24 / 11.2, 128 -> 4.18 / 10.4; the one-pass scoring of
    // round 1 (RSX_SC_TWO_PHASE=0: first round scored exactly, 96 evaluations per query) 4.0 with 64
    RSX_TRY(filter_and_select(h, q, n_items, n_eligible, elig, first_round_target(), k, s, elig_monotone));
    // exact re-scoring: the 8-wave workgroup in rounds (sc_rescore_kernel; also what the sharded stages use), or
    // -- RSX_SC_RESCORE=walk, experimental -- one wave per query walking the bound-ordered short list with
    // the fp32 pruning preview (sc_walk_kernel: identical results, 6.3 instead of 5.6 ms per step on the bench:
    // the per-query chain of ~125 dependent candidates is latency-bound at 2 waves per SIMD)
    static const bool use_walk = [] {
      const char *e = rsx::exp_env("RSX_SC_RESCORE");
      return e && e[0] == 'w';
    }();
    if (use_walk) {
      const int64_t ld = (n_items + 31) / 32 * 32;
      RSX_TRY(launch_walk(db_view(h), q, h->w->f_lb.as<lb_t>(), ld, n_items, n_eligible, elig, h->w->f_cand.as<RescoreEntry>(),
                          h->w->f_cnt.as<int32_t>(), h->w->f_thr.as<float>(), filter_eps(), d_out + b0 * k, k, s));
    } else {
      RSX_TRY(rescore(h, q, n_items, n_eligible, elig, 0, RESCORE_ALL_ROUNDS, nullptr, nullptr, k, d_out + b0 * k, s));
    }
  }
  return RSX_OK;
}

// a handful of queries (the live detector asks one at a time, PGO.cpp:561,577): ONE launch that streams the fp16 images of
// every eligible entry once, previews every pair on the matrix cores and scores the few entries the previews cannot exclude
// exactly (sc_q1.hip) -- instead of the exact-all kernel (one entry per wavefront in fp64) or the six launches of the batched
// filter chain.  filter_mode 1 / 2 keep those paths; 3 asks for this one wherever it applies.  Records are identical.
bool use_q1(const rsx_sc *h, int32_t nq, int64_t n_items) {
  static const bool off = [] {
    const char *e = rsx::exp_env("RSX_SC_Q1");
    return e && e[0] == '0';
  }();
  const int m = filter_mode_of(h);
  if (off || nq < 1 || nq > Q1_MAX_NQ || !(m == 0 || m == 3)) return false;
  // every query streams the whole database again, so with several queries against a large database the batched chain (one
  // pass of the spectral filter for all of them) takes over.  Round 6 (the queries of a call share the device's 512 workgroup
  // slots, q1_grid): MI355X, top-1, us per call single-query path / filter chain -- 10 000 keyframes: 1 query 18 / 91, 2: 22 / 94,
  // 4: 27 / 94, 8: 36 / 89; 100 000: 1 query 54 / 173, 2: 71 / 175, 4: 112 / 176, 8: 187 / 191 (round 5: 8 x 10 000 55, 8 x 100 000 305)
  // 9 .. 16 queries (32 workgroups each): 10 000 keyframes 12 queries 47 / 92, 16: 53 / 87; 50 000: 8 queries 103 / 127, 12: 167 / 134, 16: 176 / 147
  return m == 3 || nq == 1 || (int64_t)nq * n_items <= (nq <= 8 ? 800000 : 400000);
}

int run_q1(rsx_sc *h, const float *d_q, int32_t nq, int64_t n_items, int64_t n_eligible, const int64_t *d_q_elig, int32_t k,
           rsx_sc_hit *d_out, hipStream_t s) {
  if (n_items < 0) n_items = 0;
  if (!h->q1_ticket.p) {
    RSX_TRY(h->q1_ticket.reserve(q1_ticket_bytes(), s, false));
    RSX_HIP(hipMemsetAsync(h->q1_ticket.p, 0, q1_ticket_bytes(), s));
  }
  RSX_TRY(h->q1_ws.reserve(q1_workspace_bytes(n_items, nq, k), s, false));
  h->prof_kernel = q1_kernel_name();
  ProfScope ps(&h->prof, s);
  RSX_TRY(launch_q1(db_view(h), d_q, nq, n_items, n_eligible, d_q_elig, k, d_out, h->q1_ws.p, h->q1_ticket.as<unsigned>(),
                    (h->prof.on && h->stats.p) ? h->stats.as<unsigned long long>() : nullptr, s));
  ps.stop();
  return RSX_OK;
}

int run_topk(rsx_sc *h, const QueryView &qv, int64_t n_items, int64_t n_eligible, const int64_t *d_q_elig,
             int32_t k, rsx_sc_hit *d_out, hipStream_t s, bool elig_monotone = false) {
  if (use_q1(h, qv.nq, n_items)) return run_q1(h, qv.desc, qv.nq, n_items, n_eligible, d_q_elig, k, d_out, s);
  if (use_filter(h, qv.nq, n_items)) return run_topk_filtered(h, qv, n_items, n_eligible, d_q_elig, k, d_out, s, elig_monotone);
  h->prof_kernel = pair_kernel_name();
  RSX_TRY(h->w->partial.reserve(pair_partial_bytes(n_items > 0 ? n_items : 1, qv.nq, k), s, false));
  struct Hook {
    explicit Hook(PairProfiler *p) { set_pair_profiler(p); }
    ~Hook() { set_pair_profiler(nullptr); }
  } hook(&h->prof);
  return launch_pairs(db_view(h), qv, nullptr, 0, n_items, n_eligible, d_q_elig, nullptr, nullptr,
                      h->w->partial.as<rsx_sc_hit>(), d_out, k, s);
}

// the search tree over the ring keys of entries [0, n): built on the host exactly like the reference's nanoflann tree
// (SC.cpp:348-359 rebuilds it every TREE_MAKING_PERIOD detections on the CPU as well), searched on the device.
// The build itself (O(n log n) float work: 6 ms at 10 000 keys, 76 ms at 100 000) touches nothing of the handle, so the
// detector runs it with the handle UNLOCKED: the writer thread (PGO.cpp:492 process_pg) is not stalled by the reader's
// (PGO.cpp:561 process_lcd) tree rebuild.  Keys [0, n) are immutable (append-only DB), so the snapshot stays valid.
struct PreparedTree {
  int64_t n = 0;
  std::vector<KdNode16> n16;
  std::vector<int32_t> vind;
  float low[KD_DIM], high[KD_DIM];
  int depth = 0;
};

int prepare_tree_host(const float *keys, int64_t n, PreparedTree *out) {
  KdTreeHost host;
  RSX_TRY(kdtree_build_host(keys, n, &host));
  // the search kernel's 16-byte nodes (child1 = the next node: the build numbers the nodes in preorder)
  std::vector<KdNode16> &n16 = out->n16;
  n16.resize(host.nodes.size());
  for (size_t i = 0; i < host.nodes.size(); i++) {
    const KdNode &nd = host.nodes[i];
    if (nd.child1 < 0) {
      n16[i] = KdNode16{nd.left, -1 - (nd.right - nd.left), 0.0f, 0.0f};
    } else {
      if (nd.child1 != (int32_t)i + 1) return fail(RSX_ERR_RANGE, "kd-tree nodes are not in preorder");
      n16[i] = KdNode16{nd.child2, nd.divfeat, nd.divlow, nd.divhigh};
    }
  }
  // first vind position below child2 (the subtrees cover contiguous ranges of vind; children have larger node numbers)
  {
    std::vector<int32_t> lo(host.nodes.size());
    for (size_t i = host.nodes.size(); i-- > 0;) {
      const KdNode &nd = host.nodes[i];
      lo[i] = nd.child1 < 0 ? nd.left : lo[(size_t)nd.child1];
      if (nd.child1 >= 0) {
        if (lo[(size_t)nd.child2] >= (1 << 26)) return fail(RSX_ERR_RANGE, "ring-key tree over more than 2^26 keys");
        n16[i].b |= lo[(size_t)nd.child2] << 5;
      }
    }
  }
  out->vind.swap(host.vind);
  std::memcpy(out->low, host.low, sizeof(out->low));
  std::memcpy(out->high, host.high, sizeof(out->high));
  out->depth = host.depth;
  out->n = n;
  return RSX_OK;
}

int upload_tree(rsx_sc *h, rsx_sc::KdTreeDev *t, const PreparedTree &p) {
  hipStream_t s = h->stream;
  t->n = 0;
  t->n_nodes = (int32_t)p.n16.size();
  RSX_TRY(t->nodes.reserve(p.n16.size() * sizeof(KdNode16), s, false));
  RSX_TRY(t->vind.reserve(p.vind.size() * sizeof(int32_t), s, false));
  RSX_HIP(hipMemcpyAsync(t->nodes.p, p.n16.data(), p.n16.size() * sizeof(KdNode16), hipMemcpyHostToDevice, s));
  RSX_HIP(hipMemcpyAsync(t->vind.p, p.vind.data(), p.vind.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
  RSX_HIP(hipStreamSynchronize(s));  // the caller's PreparedTree may go out of scope
  std::memcpy(t->low, p.low, sizeof(t->low));
  std::memcpy(t->high, p.high, sizeof(t->high));
  t->depth = p.depth;
  t->n = p.n;
  return RSX_OK;
}

// lk holds h->mu on entry and on return; it is released while the tree is built on the host
int ensure_tree(rsx_sc *h, rsx_sc::KdTreeDev *t, int64_t n, std::unique_lock<std::mutex> &lk) {
  if (t->n == n) return RSX_OK;
  hipStream_t s = h->stream;
  std::vector<float> keys((size_t)n * NR);
  RSX_HIP(hipMemcpyAsync(keys.data(), h->rkey.p, keys.size() * sizeof(float), hipMemcpyDeviceToHost, s));
  RSX_HIP(hipStreamSynchronize(s));
  PreparedTree p;
  lk.unlock();
  const int st = prepare_tree_host(keys.data(), n, &p);
  const std::string err = st == RSX_OK ? std::string() : last_error();
  lk.lock();
  if (st != RSX_OK) {
    last_error() = err;
    return st;
  }
  RSX_TRY(set_device(h));
  if (t->n == n) return RSX_OK;  // another detector thread built the same tree meanwhile
  if (n > h->n_local) return fail(RSX_ERR_RANGE, "the database shrank while the ring-key tree was being built");
  return upload_tree(h, t, p);
}

// candidate scoring shared by detect_loop_closure / between_session (SC.cpp:362-417)
int score_candidates_and_finish(rsx_sc *h, const QueryView &qv, const float *d_qkey, int64_t n_search, rsx_sc::KdTreeDev *tree,
                                int mode, int32_t *loop_id, float *yaw, double *min_dist, int32_t *nn_idx) {
  hipStream_t s = h->stream;
  const int k = h->p.num_candidates;
  double best_d = 10000000;  // SC.cpp:362
  int best_align = 0, best_idx = 0;
  if (mode == RSX_SC_MODE_CANDIDATE) {
    RSX_TRY(h->small.reserve(4096, s, false));
    int32_t *d_idx = h->small.as<int32_t>();              // [k]
    float *d_kd = reinterpret_cast<float *>(d_idx + 64);  // [k]
    int32_t *d_found = d_idx + 128;
    if (n_search > 0) {
      // nanoflann's own walk of nanoflann's own tree: the reference's candidates, ties included (sc_kdtree.h)
      if (tree->n != n_search) return fail(RSX_ERR_INTERNAL, "ring-key tree over %lld keys was not prepared", (long long)n_search);
      // the true k-th smallest distance, by the parallel brute-force pass: lets the walk skip what cannot matter
      int32_t *b_idx = d_idx + 192;
      float *b_kd = reinterpret_cast<float *>(d_idx + 256);
      int32_t *b_found = d_idx + 320;
      RSX_TRY(h->knn_ws.reserve((size_t)n_search * 2 * sizeof(float), s, false));
      float *dist_all = h->knn_ws.as<float>(), *dist_tree = dist_all + n_search;
      RSX_TRY(launch_knn(h->rkey.as<float>(), n_search, d_qkey, k, dist_all, b_idx, b_kd, b_found, s));
      int32_t *cand_pos = d_idx + 384, *cand_count = d_idx + 384 + 64;
      RSX_TRY(launch_knn_tree_order(dist_all, tree->vind.as<int32_t>(), n_search, dist_tree, k, b_kd, b_found, cand_pos, cand_count, s));
      KdSearchArgs ka;
      ka.bound_dist = b_kd;
      ka.bound_found = b_found;
      ka.dist_tree = dist_tree;
      ka.cand_pos = cand_pos;
      ka.cand_count = cand_count;
      ka.nodes = tree->nodes.as<KdNode16>();
      ka.n_nodes = tree->n_nodes;
      ka.n = n_search;
      ka.vind = tree->vind.as<int32_t>();
      ka.keys = h->rkey.as<float>();
      ka.qkey = d_qkey;
      std::memcpy(ka.low, tree->low, sizeof(ka.low));
      std::memcpy(ka.high, tree->high, sizeof(ka.high));
      ka.k = k;
      ka.out_idx = d_idx;
      ka.out_dist = d_kd;
      ka.out_found = d_found;
      RSX_TRY(launch_knn_tree(ka, s));
    } else {  // an empty search set (tree_making_period > 1 with num_exclude_recent changes): no neighbour, slots stay 0
      RSX_TRY(h->knn_ws.reserve(sizeof(float), s, false));
      RSX_TRY(launch_knn(h->rkey.as<float>(), 0, d_qkey, k, h->knn_ws.as<float>(), d_idx, d_kd, d_found, s));
    }
    // the k distances and shifts next to the indices in `small`: ONE read-back for all three
    double *d_dist = reinterpret_cast<double *>(h->small.as<char>() + 1024);
    int32_t *d_shift = reinterpret_cast<int32_t *>(h->small.as<char>() + 2048);
    RSX_TRY(launch_pairs(db_view(h), qv, d_idx, 0, k, -1, nullptr, d_dist, d_shift, nullptr, nullptr, 0, s));
    RSX_TRY(ensure_pinned(h, 4096));
    char *hp = static_cast<char *>(h->pinned);
    RSX_HIP(hipMemcpyAsync(hp, h->small.p, 2048 + 256, hipMemcpyDeviceToHost, s));
    RSX_HIP(hipStreamSynchronize(s));
    const int32_t *ci = reinterpret_cast<const int32_t *>(hp);
    const double *cd = reinterpret_cast<const double *>(hp + 1024);
    const int32_t *cs = reinterpret_cast<const int32_t *>(hp + 2048);
    for (int c = 0; c < k; c++) {  // SC.cpp:380-395: kNN order, strict <
      if (cd[c] < best_d) {
        best_d = cd[c];
        best_align = cs[c];
        best_idx = ci[c];
      }
    }
  } else {
    // the one record goes straight into pinned host memory (device-visible): no copy to enqueue
    RSX_TRY(ensure_pinned(h, 4096));
    RSX_TRY(run_topk(h, qv, n_search, n_search, nullptr, 1, static_cast<rsx_sc_hit *>(h->pinned), s));
    RSX_HIP(hipStreamSynchronize(s));
    const rsx_sc_hit *r = static_cast<const rsx_sc_hit *>(h->pinned);
    if (r->dist < best_d) {
      best_d = r->dist;
      best_align = r->shift;
      best_idx = r->index;
    }
  }
  int lid = -1;
  if (best_d < h->p.dist_thres) lid = best_idx;  // SC.cpp:401-403
  if (loop_id) *loop_id = lid;
  if (yaw) *yaw = yaw_from_shift(best_align);    // SC.cpp:417
  if (min_dist) *min_dist = best_d;
  if (nn_idx) *nn_idx = best_idx;
  return RSX_OK;
}

}  // namespace

extern "C" {

const char *rsx_last_error_string(void) { return rsx::last_error().c_str(); }

const char *rsx_version(void) {
#ifdef RSX_EXPERIMENTS
  return "rsx 0.3 (gfx950, HIP; ScanContext + ORORA hot path) +experiments";
#else
  return "rsx 0.3 (gfx950, HIP; ScanContext + ORORA hot path)";
#endif
}

int rsx_device_count(void) try {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
} RSX_CATCH_ALL

const char *rsx_sc_dominant_kernel_name(void) { return pair_kernel_name(); }

const char *rsx_sc_profiled_kernel_name(rsx_sc *h) {
  if (!h) return "";
  std::lock_guard<std::mutex> lk(h->mu);
  return h->prof_kernel;
}

int rsx_sc_default_params(rsx_sc_params *p) try {
  if (!p) return fail(RSX_ERR_BAD_ARG, "null params");
  p->lidar_height = 2.0;
  p->max_radius = 80.0;
  p->num_exclude_recent = 30;
  p->num_candidates = 3;
  p->search_ratio = 0.1;
  p->dist_thres = 0.2;
  p->tree_making_period = 30;
  p->device = 0;
  p->shard_rank = 0;
  p->shard_world = 1;
  p->capacity_hint = 1024;
  p->filter_mode = 0;
  p->filter_kind = 0;
  p->sum_order = RSX_SC_SUM_EIGEN_SSE2;
  return RSX_OK;
} RSX_CATCH_ALL

int rsx_sc_create(const rsx_sc_params *p, rsx_sc **out) try {
  if (!out) return fail(RSX_ERR_BAD_ARG, "null out");
  *out = nullptr;
  rsx_sc_params d;
  rsx_sc_default_params(&d);
  if (p) d = *p;
  if (d.shard_world < 1 || d.shard_rank < 0 || d.shard_rank >= d.shard_world)
    return fail(RSX_ERR_BAD_ARG, "bad shard %d/%d", d.shard_rank, d.shard_world);
  if (d.num_candidates < 1 || d.num_candidates > RSX_SC_MAX_TOPK)
    return fail(RSX_ERR_BAD_ARG, "num_candidates must be in [1,%d]", RSX_SC_MAX_TOPK);
  if ((int)std::lround(0.5 * d.search_ratio * NS) != 3)
    return fail(RSX_ERR_BAD_ARG, "kernels are specialised for SEARCH_RADIUS 3 (search_ratio 0.1, SC.h:96)");
  if (d.tree_making_period < 1 || d.num_exclude_recent < 0) return fail(RSX_ERR_BAD_ARG, "bad detector params");
  if (d.filter_mode < 0 || d.filter_mode > 3) return fail(RSX_ERR_BAD_ARG, "filter_mode must be 0 (auto), 1 (off), 2 (force) or 3 (single-query path)");
  if (d.sum_order < 0 || d.sum_order > RSX_SC_SUM_EIGEN34_AVX_FMA) return fail(RSX_ERR_BAD_ARG, "sum_order must be RSX_SC_SUM_EIGEN_SSE2, _SEQ, _EIGEN_AVX_FMA or _EIGEN34_AVX_FMA");
  if (d.filter_kind < 0 || d.filter_kind > 3)
    return fail(RSX_ERR_BAD_ARG, "filter_kind must be 0 (auto), 1 (direct), 2 (spectral) or 3 (spectral, two waves per SIMD)");
  int ndev = rsx_device_count();
  if (ndev <= 0) return fail(RSX_ERR_NO_DEVICE, "no HIP device visible (librsx has no CPU fallback)");
  if (d.device < 0 || d.device >= ndev) return fail(RSX_ERR_NO_DEVICE, "device %d out of range (%d visible)", d.device, ndev);
  rsx_sc *h = new (std::nothrow) rsx_sc();
  if (!h) return fail(RSX_ERR_OOM, "host alloc");
  h->p = d;
  int st = set_device(h);
  if (st == RSX_OK) {
    hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (e != hipSuccess) st = fail(RSX_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
  }
  if (st == RSX_OK) st = ensure_capacity(h, d.capacity_hint > 0 ? d.capacity_hint : 1024);
  if (st != RSX_OK) {
    rsx_sc_destroy(h);
    return st;
  }
  *out = h;
  return RSX_OK;
} RSX_CATCH_ALL

int rsx_sc_destroy(rsx_sc *h) try {
  if (!h) return RSX_OK;
  (void)hipSetDevice(h->p.device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  for (DevBuf *b : {&h->hn, &h->cmask, &h->sp, &h->sp_aux, &h->hnr, &h->vk16, &h->vk_n, &h->st_partial, &h->stats, &h->helper_ws, &h->tree.nodes, &h->tree.vind,
                    &h->tree_batch.nodes, &h->tree_batch.vind, &h->q1_ws, &h->q1_ticket}) b->release();
  for (DevBuf *b : {&h->desc, &h->vkey, &h->norm, &h->rkey, &h->pts_ws, &h->q_desc, &h->topk, &h->knn_ws, &h->small, &h->pair_out, &h->q_elig})
    b->release();
  for (auto &w : h->ws)
    for (DevBuf *b : w.all) b->release();
  if (h->pinned) (void)hipHostFree(h->pinned);
  if (h->prof.ev) {
    for (int i = 0; i < 2 * PairProfiler::kMax; i++) (void)hipEventDestroy(h->prof.ev[i]);
    delete[] h->prof.ev;
  }
  for (auto e : h->up_ev)
    if (e) (void)hipEventDestroy(e);
  for (auto &sl : h->ins) {
    if (sl.host) (void)hipHostFree(sl.host);
    if (sl.done) (void)hipEventDestroy(sl.done);
  }
  if (h->lane_ev) (void)hipEventDestroy(h->lane_ev);
  if (h->stream_switch) (void)hipEventDestroy(h->stream_switch);
  if (h->up_stream) (void)hipStreamDestroy(h->up_stream);
  if (h->stream_b) (void)hipStreamDestroy(h->stream_b);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return RSX_OK;
} RSX_CATCH_ALL

int rsx_sc_set_dist_thres(rsx_sc *h, double thres) try {
  if (!h) return fail(RSX_ERR_BAD_ARG, "null handle");
  std::lock_guard<std::mutex> lk(h->mu);
  h->p.dist_thres = thres;
  return RSX_OK;
} RSX_CATCH_ALL

int rsx_sc_size(rsx_sc *h, int64_t *n) try {
  if (!h || !n) return fail(RSX_ERR_BAD_ARG, "null arg");
  std::lock_guard<std::mutex> lk(h->mu);
  *n = h->n_global;
  return RSX_OK;
} RSX_CATCH_ALL

int rsx_sc_local_size(rsx_sc *h, int64_t *n) try {
  if (!h || !n) return fail(RSX_ERR_BAD_ARG, "null arg");
  std::lock_guard<std::mutex> lk(h->mu);
  *n = h->n_local;
  return RSX_OK;
} RSX_CATCH_ALL

int rsx_sc_ringkey_tree_layout(const float *keys20, int64_t n, int32_t *out_vind, int32_t *out_n_nodes, int32_t *out_depth) try {
  if (!keys20 || !out_vind || n < 1) return fail(RSX_ERR_BAD_ARG, "bad arg");
  KdTreeHost host;
  RSX_TRY(kdtree_build_host(keys20, n, &host));
  std::memcpy(out_vind, host.vind.data(), host.vind.size() * sizeof(int32_t));
  if (out_n_nodes) *out_n_nodes = (int32_t)host.nodes.size();
  if (out_depth) *out_depth = host.depth;
  return RSX_OK;
} RSX_CATCH_ALL

int rsx_sc_tree_size(rsx_sc *h, int64_t *n) try {
  if (!h || !n) return fail(RSX_ERR_BAD_ARG, "null arg");
  std::lock_guard<std::mutex> lk(h->mu);
  *n = h->tree_size;
  return RSX_OK;
} RSX_CATCH_ALL

int rsx_sc_add_points(rsx_sc *h, const void *pts, size_t n, size_t stride_bytes, int32_t *out_index) try {
  if (!h || (!pts && n)) return fail(RSX_ERR_BAD_ARG, "null arg");
  if (stride_bytes < 12 || (stride_bytes & 3)) return fail(RSX_ERR_BAD_ARG, "stride_bytes must be >= 12 and a multiple of 4");
  std::lock_guard<std::mutex> lk(h->mu);
  RSX_TRY(set_device(h));
  const int64_t g = h->n_global;
  if (owns(h, g)) {
    const int64_t slot = h->n_local;
    RSX_TRY(ensure_capacity(h, slot + 1));
    const size_t bytes = n * stride_bytes;
    rsx_sc::InsSlot &sl = h->ins[h->ins_next];
    h->ins_next = (h->ins_next + 1) % rsx_sc::kInsSlots;
    if (sl.used) RSX_HIP(hipEventSynchronize(sl.done));  // the insert that used this slot 8 calls ago
    if (bytes > sl.cap) {
      if (sl.host) (void)hipHostFree(sl.host);
      sl.host = nullptr;
      sl.cap = 0;
      size_t cap = 32768;
      while (cap < bytes) cap *= 2;
      RSX_HIP(hipHostMalloc(&sl.host, cap, hipHostMallocDefault));
      sl.cap = cap;
    }
    if (!sl.done) RSX_HIP(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
    if (bytes) std::memcpy(sl.host, pts, bytes);  // the caller's buffer is free again when this call returns
    // the kernel reads the points out of the pinned slot itself (each once, 12 of stride_bytes bytes): no copy to enqueue
    RSX_TRY(insert_cloud(h, sl.host, (int64_t)n, (int64_t)stride_bytes, slot, h->stream));
    RSX_HIP(hipEventRecord(sl.done, h->stream));
    sl.used = true;
    h->last_insert = sl.done;
    h->n_local = slot + 1;
  }
  h->n_global = g + 1;
  if (out_index) *out_index = (int32_t)g;
  return RSX_OK;
} RSX_CATCH_ALL

int rsx_sc_add_points_downsampled(rsx_sc *h, rsx_voxelgrid *vg, const void *pts, size_t n, size_t stride_bytes, float leaf,
                                  int32_t *out_index) try {
  if (!h || !vg || (!pts && n)) return fail(RSX_ERR_BAD_ARG, "null arg");
  if (stride_bytes < 12 || (stride_bytes & 3)) return fail(RSX_ERR_BAD_ARG, "stride_bytes must be >= 12 and a multiple of 4");
  if (!(leaf > 0.0f)) return fail(RSX_ERR_BAD_ARG, "leaf must be positive");
  if (rsx::vg::device_of(vg) != h->p.device) return fail(RSX_ERR_BAD_ARG, "voxel grid and ScanContext handles are on different devices");
  std::lock_guard<std::mutex> lk(h->mu);
  std::lock_guard<std::mutex> lkv(rsx::vg::mutex_of(vg));
  RSX_TRY(set_device(h));
  const int64_t g = h->n_global;
  if (owns(h, g)) {
    // intensity is not used by the descriptor (SC.cpp:151-195 reads x, y, z only)
    const float *d_ds = nullptr;
    int64_t nds = 0;
    RSX_TRY(rsx::vg::upload_and_filter(vg, pts, (int64_t)n, (int64_t)stride_bytes, -1, leaf, (int64_t)(n ? n : 1), &d_ds, &nds));
    const int64_t slot = h->n_local;
    RSX_TRY(ensure_capacity(h, slot + 1));
    // upload_and_filter synchronised vg's stream: d_ds is complete and stays valid under vg's lock
    RSX_TRY(insert_cloud(h, d_ds ? static_cast<const void *>(d_ds) : h->desc.p, nds, 16, slot, h->stream));
    RSX_HIP(hipStreamSynchronize(h->stream));  // d_ds belongs to vg: its next call must not overwrite it under the kernel
    h->n_local = slot + 1;
  }
  h->n_global = g + 1;
  if (out_index) *out_index = (int32_t)g;
  return RSX_OK;
} RSX_CATCH_ALL

int rsx_sc_add_keyframe(rsx_sc *h, rsx_voxelgrid *vg, rsx_kfstore *kf, const void *pts, size_t n, size_t stride_bytes,
                        int32_t intensity_offset, float leaf, int32_t *out_index) try {
  if (!h || !vg || !kf || (!pts && n)) return fail(RSX_ERR_BAD_ARG, "null arg");
  if (stride_bytes < 12 || (stride_bytes & 3)) return fail(RSX_ERR_BAD_ARG, "stride_bytes must be >= 12 and a multiple of 4");
  if (intensity_offset >= 0 && ((intensity_offset & 3) || (size_t)intensity_offset + 4 > stride_bytes))
    return fail(RSX_ERR_BAD_ARG, "intensity_offset outside the point");
  if (!(leaf > 0.0f)) return fail(RSX_ERR_BAD_ARG, "leaf must be positive");
  if (rsx::vg::device_of(vg) != h->p.device || rsx::kf::device_of(kf) != h->p.device)
    return fail(RSX_ERR_BAD_ARG, "voxel grid, keyframe store and ScanContext handles are on different devices");
  std::lock_guard<std::mutex> lk(h->mu);
  std::lock_guard<std::mutex> lkv(rsx::vg::mutex_of(vg));
  std::lock_guard<std::mutex> lkk(rsx::kf::mutex_of(kf));
  RSX_TRY(set_device(h));
  const int64_t g = h->n_global;
  // downSizeFilterScancontext.filter(*thisKeyFrameDS) (PGO.cpp:482-484): every shard keeps the keyframe cloud (the loop
  // verification runs where the pose graph lives), only the owning shard builds the descriptor
  const float *d_ds = nullptr;
  int64_t nds = 0;
  RSX_TRY(rsx::vg::upload_and_filter(vg, pts, (int64_t)n, (int64_t)stride_bytes, intensity_offset, leaf, (int64_t)(n ? n : 1), &d_ds, &nds));
  // the descriptor first, the store last, the counters only when both stand: a failure in either leaves database and store
  // with the same number of keyframes (the slot written here is simply written again by the next call)
  const bool mine = owns(h, g);
  if (mine) {
    const int64_t slot = h->n_local;
    RSX_TRY(ensure_capacity(h, slot + 1));
    RSX_TRY(insert_cloud(h, d_ds ? static_cast<const void *>(d_ds) : h->desc.p, nds, 16, slot, h->stream));  // makeAndSaveScancontextAndKeys (PGO.cpp:492)
    RSX_HIP(hipStreamSynchronize(h->stream));  // d_ds belongs to vg: it must not be overwritten by vg's next call before the build has read it
  }
  int32_t kf_index = -1;
  RSX_TRY(rsx::kf::append_device_locked(kf, d_ds, nds, &kf_index));  // keyframeLaserClouds.push_back (PGO.cpp:487)
  if (mine) h->n_local = h->n_local + 1;
  h->n_global = g + 1;
  if (out_index) *out_index = (int32_t)g;
  return RSX_OK;
} RSX_CATCH_ALL

static int add_f32_locked(rsx_sc *h, const float *src, int64_t n, bool src_is_device, hipStream_t s) {
  // n consecutive global keyframes starting at h->n_global; copy the owned rows
  const int64_t g0 = h->n_global;
  const int64_t w = h->p.shard_world, r = h->p.shard_rank;
  int64_t first = g0 + (((r - g0) % w) + w) % w;  // first owned global index >= g0
  int64_t count = first < g0 + n ? (g0 + n - first + w - 1) / w : 0;
  if (count > 0) {
    const int64_t slot = h->n_local;
    RSX_TRY(ensure_capacity(h, slot + count));
    float *dst = h->desc.as<float>() + slot * DS;
    RSX_HIP(hipMemcpy2DAsync(dst, DS * sizeof(float), src + (first - g0) * DS, (size_t)w * DS * sizeof(float),
                             DS * sizeof(float), (size_t)count,
                             src_is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s));
    RSX_TRY(launch_keys(dst, count, h->vkey.as<double>() + slot * NS, h->norm.as<double>() + slot * NS,
                        h->rkey.as<float>() + slot * NR, s, h->p.sum_order));
    RSX_TRY(build_db_images(h, slot, count, s));
    h->n_local = slot + count;
  }
  h->n_global = g0 + n;
  return RSX_OK;
}

static int add_descriptor_f64(rsx_sc *h, const double *desc, bool allow_rounding, int32_t *out_index, double *max_err) {
  if (!h || !desc) return fail(RSX_ERR_BAD_ARG, "null arg");
  float f[DS];
  double worst = 0.0;
  for (int i = 0; i < DS; i++) {
    f[i] = (float)desc[i];
    if (!((double)f[i] == desc[i])) {  // also NaN
      if (!allow_rounding || !(desc[i] == desc[i]))
        return fail(RSX_ERR_NOT_FP32_EXACT, "descriptor element %d (%.17g) is not exactly representable in fp32", i, desc[i]);
      const double e = std::fabs((double)f[i] - desc[i]);
      if (e > worst) worst = e;
    }
  }
  if (max_err) *max_err = worst;
  std::lock_guard<std::mutex> lk(h->mu);
  RSX_TRY(set_device(h));
  const int64_t g = h->n_global;
  RSX_TRY(add_f32_locked(h, f, 1, false, h->stream));
  if (out_index) *out_index = (int32_t)g;
  return RSX_OK;
}

int rsx_sc_add_descriptor(rsx_sc *h, const double *desc, int32_t *out_index) try {
  return add_descriptor_f64(h, desc, false, out_index, nullptr);
} RSX_CATCH_ALL

int rsx_sc_add_descriptor_rounded(rsx_sc *h, const double *desc, int32_t *out_index, double *max_abs_rounding) try {
  return add_descriptor_f64(h, desc, true, out_index, max_abs_rounding);
} RSX_CATCH_ALL

int rsx_sc_add_descriptors_f32(rsx_sc *h, const float *descs, int64_t n) try {
  if (!h || (!descs && n) || n < 0) return fail(RSX_ERR_BAD_ARG, "bad arg");
  if (n == 0) return RSX_OK;
  std::lock_guard<std::mutex> lk(h->mu);
  RSX_TRY(set_device(h));
  return add_f32_locked(h, descs, n, false, h->stream);
} RSX_CATCH_ALL

int rsx_sc_add_descriptors_f32_device(rsx_sc *h, const float *d_descs, int64_t n, void *stream) try {
  if (!h || (!d_descs && n) || n < 0) return fail(RSX_ERR_BAD_ARG, "bad arg");
  if (n == 0) return RSX_OK;
  std::lock_guard<std::mutex> lk(h->mu);
  RSX_TRY(set_device(h));
  hipStream_t s;
  RSX_TRY(use_stream(h, stream, &s));
  return add_f32_locked(h, d_descs, n, true, s);
} RSX_CATCH_ALL

int rsx_sc_export_descriptors_f32(rsx_sc *h, int64_t first_slot, int64_t count, float *out) try {
  if (!h || (!out && count)) return fail(RSX_ERR_BAD_ARG, "null arg");
  std::lock_guard<std::mutex> lk(h->mu);
  if (first_slot < 0 || count < 0 || first_slot + count > h->n_local) return fail(RSX_ERR_RANGE, "slot range out of bounds");
  if (count == 0) return RSX_OK;
  RSX_TRY(set_device(h));
  RSX_HIP(hipMemcpyAsync(out, h->desc.as<float>() + first_slot * DS, (size_t)count * DS * sizeof(float), hipMemcpyDeviceToHost,
                         h->stream));
  RSX_HIP(hipStreamSynchronize(h->stream));
  return RSX_OK;
} RSX_CATCH_ALL

// ---- on-disk database.  Little-endian; 64-byte header, then n_local fp32 sector-major
// descriptors of 4800 B in slot order.  Keys, norms and filter images are derived data: rebuilt on load.
namespace {
struct DbFileHeader {
  char magic[8];        // "RSXSCDB1"
  uint32_t version;     // 1
  uint32_t num_ring;    // 20
  uint32_t num_sector;  // 60
  uint32_t dtype;       // 0 = fp32
  int64_t n_global;     // keyframes known to the saving handle
  int64_t n_local;      // descriptors in this file
  int32_t shard_rank, shard_world;
  uint8_t reserved[16];
};
static_assert(sizeof(DbFileHeader) == 64, "header layout");
}  // namespace

int rsx_sc_save(rsx_sc *h, const char *path) try {
  if (!h || !path) return fail(RSX_ERR_BAD_ARG, "null arg");
  std::lock_guard<std::mutex> lk(h->mu);
  RSX_TRY(set_device(h));
  std::vector<float> buf((size_t)h->n_local * DS);
  if (h->n_local) {
    RSX_HIP(hipMemcpyAsync(buf.data(), h->desc.p, buf.size() * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    RSX_HIP(hipStreamSynchronize(h->stream));
  }
  DbFileHeader hd{};
  std::memcpy(hd.magic, "RSXSCDB1", 8);
  hd.version = 1;
  hd.num_ring = NR;
  hd.num_sector = NS;
  hd.dtype = 0;
  hd.n_global = h->n_global;
  hd.n_local = h->n_local;
  hd.shard_rank = h->p.shard_rank;
  hd.shard_world = h->p.shard_world;
  FILE *f = std::fopen(path, "wb");
  if (!f) return fail(RSX_ERR_BAD_ARG, "cannot open %s for writing", path);
  bool ok = std::fwrite(&hd, sizeof(hd), 1, f) == 1;
  ok = ok && (buf.empty() || std::fwrite(buf.data(), sizeof(float), buf.size(), f) == buf.size());
  ok = (std::fclose(f) == 0) && ok;
  if (!ok) return fail(RSX_ERR_INTERNAL, "short write to %s", path);
  return RSX_OK;
} RSX_CATCH_ALL

int rsx_sc_load(rsx_sc *h, const char *path, int64_t *n_loaded) try {
  if (!h || !path) return fail(RSX_ERR_BAD_ARG, "null arg");
  FILE *f = std::fopen(path, "rb");
  if (!f) return fail(RSX_ERR_BAD_ARG, "cannot open %s", path);
  DbFileHeader hd{};
  std::vector<float> buf;
  bool ok = std::fread(&hd, sizeof(hd), 1, f) == 1 && std::memcmp(hd.magic, "RSXSCDB1", 8) == 0 && hd.version == 1 &&
            hd.num_ring == (uint32_t)NR && hd.num_sector == (uint32_t)NS && hd.dtype == 0 && hd.n_local >= 0 && hd.n_global >= 0 &&
            hd.n_local <= hd.n_global && hd.shard_world >= 1 && hd.shard_rank >= 0 && hd.shard_rank < hd.shard_world;
  // the header is not trusted: a shard holds exactly the indices i < n_global with i % world == rank, and the file must
  // be as long as it says BEFORE anything is allocated from its numbers
  if (ok) {
    const int64_t expect = hd.n_global > hd.shard_rank ? (hd.n_global - hd.shard_rank + hd.shard_world - 1) / hd.shard_world : 0;
    ok = hd.n_local == expect;
  }
  if (ok) {
    const long here = std::ftell(f);
    ok = here == (long)sizeof(hd) && std::fseek(f, 0, SEEK_END) == 0;
    const long end = ok ? std::ftell(f) : -1;
    ok = ok && end >= 0 && (uint64_t)end == (uint64_t)sizeof(hd) + (uint64_t)hd.n_local * DS * sizeof(float) &&
         std::fseek(f, here, SEEK_SET) == 0;
  }
  if (ok) {
    try {
      buf.resize((size_t)hd.n_local * DS);
    } catch (const std::bad_alloc &) {
      std::fclose(f);
      return fail(RSX_ERR_OOM, "%s: %lld descriptors do not fit in host memory", path, (long long)hd.n_local);
    }
    ok = buf.empty() || std::fread(buf.data(), sizeof(float), buf.size(), f) == buf.size();
  }
  std::fclose(f);
  if (!ok) return fail(RSX_ERR_BAD_ARG, "%s is not a complete, consistent RSXSCDB1 file for 20 x 60 fp32 descriptors", path);
  std::lock_guard<std::mutex> lk(h->mu);
  // a shard file restores the shard it was saved from; an unsharded file can be loaded into any (sharded) handle,
  // which keeps its own residue class
  if (hd.shard_world != 1) {
    if (hd.shard_world != h->p.shard_world || hd.shard_rank != h->p.shard_rank || h->n_global != 0)
      return fail(RSX_ERR_BAD_ARG, "%s holds shard %d/%d: load it into an empty handle of the same shard", path, hd.shard_rank,
                  hd.shard_world);
    RSX_TRY(set_device(h));
    if (hd.n_local) {
      RSX_TRY(ensure_capacity(h, hd.n_local));
      RSX_HIP(hipMemcpyAsync(h->desc.p, buf.data(), buf.size() * sizeof(float), hipMemcpyHostToDevice, h->stream));
      RSX_TRY(launch_keys(h->desc.as<float>(), hd.n_local, h->vkey.as<double>(), h->norm.as<double>(), h->rkey.as<float>(), h->stream, h->p.sum_order));
      RSX_TRY(build_db_images(h, 0, hd.n_local, h->stream));
    }
    h->n_local = hd.n_local;
    h->n_global = hd.n_global;
  } else {
    RSX_TRY(set_device(h));
    if (hd.n_local) RSX_TRY(add_f32_locked(h, buf.data(), hd.n_local, false, h->stream));
  }
  if (n_loaded) *n_loaded = hd.n_local;
  return RSX_OK;
} RSX_CATCH_ALL

static int local_slot_of(rsx_sc *h, int64_t index, int64_t *slot) {
  if (index < 0 || index >= h->n_global) return fail(RSX_ERR_RANGE, "index %lld out of range [0,%lld)", (long long)index, (long long)h->n_global);
  if (!owns(h, index)) return fail(RSX_ERR_RANGE, "index %lld is not stored on shard %d/%d", (long long)index, h->p.shard_rank, h->p.shard_world);
  *slot = index / h->p.shard_world;
  if (*slot >= h->n_local) return fail(RSX_ERR_INTERNAL, "index %lld maps to slot %lld but the shard holds %lld", (long long)index, (long long)*slot, (long long)h->n_local);
  return RSX_OK;
}

int rsx_sc_get_descriptor(rsx_sc *h, int64_t index, double *out) try {
  if (!h || !out) return fail(RSX_ERR_BAD_ARG, "null arg");
  std::lock_guard<std::mutex> lk(h->mu);
  RSX_TRY(set_device(h));
  int64_t slot;
  RSX_TRY(local_slot_of(h, index, &slot));
  float f[DS];
  RSX_HIP(hipMemcpyAsync(f, h->desc.as<float>() + slot * DS, sizeof(f), hipMemcpyDeviceToHost, h->stream));
  RSX_HIP(hipStreamSynchronize(h->stream));
  for (int i = 0; i < DS; i++) out[i] = (double)f[i];
  return RSX_OK;
} RSX_CATCH_ALL

int rsx_sc_get_ringkey(rsx_sc *h, int64_t index, float *out20) try {
  if (!h || !out20) return fail(RSX_ERR_BAD_ARG, "null arg");
  std::lock_guard<std::mutex> lk(h->mu);
  RSX_TRY(set_device(h));
  int64_t slot;
  RSX_TRY(local_slot_of(h, index, &slot));
  RSX_HIP(hipMemcpyAsync(out20, h->rkey.as<float>() + slot * NR, NR * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  RSX_HIP(hipStreamSynchronize(h->stream));
  return RSX_OK;
} RSX_CATCH_ALL

int rsx_sc_get_sectorkey(rsx_sc *h, int64_t index, double *out60) try {
  if (!h || !out60) return fail(RSX_ERR_BAD_ARG, "null arg");
  std::lock_guard<std::mutex> lk(h->mu);
  RSX_TRY(set_device(h));
  int64_t slot;
  RSX_TRY(local_slot_of(h, index, &slot));
  RSX_HIP(hipMemcpyAsync(out60, h->vkey.as<double>() + slot * NS, NS * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  RSX_HIP(hipStreamSynchronize(h->stream));
  return RSX_OK;
} RSX_CATCH_ALL

int rsx_sc_detect_loop_closure_ex(rsx_sc *h, int mode, rsx_sc_detection *out) try {
  if (!h || !out) return fail(RSX_ERR_BAD_ARG, "null arg");
  if (mode != RSX_SC_MODE_CANDIDATE && mode != RSX_SC_MODE_EXHAUSTIVE) return fail(RSX_ERR_BAD_ARG, "bad mode %d", mode);
  std::unique_lock<std::mutex> lk(h->mu);
  if (h->p.shard_world != 1) return fail(RSX_ERR_BAD_ARG, "detect_loop_closure needs an unsharded handle; use rsx_scs_* or rsx_sc_query_device + merge");
  RSX_TRY(set_device(h));
  const int64_t N = h->n_global;  // snapshot under the lock (the reference races here, PGO.cpp:561)
  out->loop_id = -1;
  out->yaw_diff_rad = 0.0f;
  out->min_dist = 10000000;
  out->nn_idx = 0;
  out->query_idx = (int32_t)(N - 1);
  out->searched = 0;
  out->reserved = 0;
  out->dist_thres = h->p.dist_thres;
  if (N == 0 || N < h->p.num_exclude_recent + 1) return RSX_OK;  // SC.cpp:341-345
  int64_t n_search = h->tree_size;
  if (h->tree_counter % h->p.tree_making_period == 0)  // SC.cpp:348-359
    n_search = N - h->p.num_exclude_recent;
  // candidate mode walks the ring-key tree: (re)built here with the handle unlocked; keyframes added meanwhile are not
  // part of this detection (N was taken above), and the device arrays are addressed only after the lock is back
  if (mode == RSX_SC_MODE_CANDIDATE && n_search >= 1) RSX_TRY(ensure_tree(h, &h->tree, n_search, lk));
  // the lock was away during the build: a concurrent rsx_sc_load may have replaced the database
  if (N > h->n_global || n_search > h->n_global)
    return fail(RSX_ERR_RANGE, "the database was replaced while the ring-key tree was being built");
  // the detector's state advances only now that the tree stands (a failed build leaves it where it was)
  h->tree_size = n_search;
  h->tree_counter = h->tree_counter + 1;               // SC.cpp:360
  QueryView qv;
  qv.desc = h->desc.as<float>() + (N - 1) * DS;        // SC.cpp:336
  qv.vkey = h->vkey.as<double>() + (N - 1) * NS;
  qv.norm = h->norm.as<double>() + (N - 1) * NS;
  qv.nq = 1;
  out->searched = 1;
  return score_candidates_and_finish(h, qv, h->rkey.as<float>() + (N - 1) * NR /* SC.cpp:335 */, n_search, &h->tree, mode,
                                     &out->loop_id, &out->yaw_diff_rad, &out->min_dist, &out->nn_idx);
} RSX_CATCH_ALL

int rsx_sc_detect_loop_closure(rsx_sc *h, int mode, int32_t *loop_id, float *yaw, double *min_dist, int32_t *nn_idx) try {
  rsx_sc_detection d;
  RSX_TRY(rsx_sc_detect_loop_closure_ex(h, mode, &d));
  if (loop_id) *loop_id = d.loop_id;
  if (yaw) *yaw = d.yaw_diff_rad;
  if (min_dist) *min_dist = d.min_dist;
  if (nn_idx) *nn_idx = d.nn_idx;
  return RSX_OK;
} RSX_CATCH_ALL

// ---- Scancontext.h:60-66: the public helper methods, stateless, on arbitrary double descriptors ----
static int helper_call(rsx_sc *h, int op, const double *a, size_t na, const double *b, size_t nb, double *out_d, size_t nd,
                       int32_t *out_i) {
  std::lock_guard<std::mutex> lk(h->mu);
  RSX_TRY(set_device(h));
  hipStream_t s = h->stream;
  RSX_TRY(h->helper_ws.reserve((2 * DS + 128) * sizeof(double), s, false));
  double *d_a = h->helper_ws.as<double>(), *d_b = d_a + DS, *d_o = d_b + DS;
  int32_t *d_i = reinterpret_cast<int32_t *>(d_o + 96);
  RSX_HIP(hipMemcpyAsync(d_a, a, na * sizeof(double), hipMemcpyHostToDevice, s));
  if (b) RSX_HIP(hipMemcpyAsync(d_b, b, nb * sizeof(double), hipMemcpyHostToDevice, s));
  RSX_TRY(launch_helper(op, d_a, b ? d_b : nullptr, d_o, d_i, s, h->p.sum_order));
  if (out_d) RSX_HIP(hipMemcpyAsync(out_d, d_o, nd * sizeof(double), hipMemcpyDeviceToHost, s));
  if (out_i) RSX_HIP(hipMemcpyAsync(out_i, d_i, sizeof(int32_t), hipMemcpyDeviceToHost, s));
  RSX_HIP(hipStreamSynchronize(s));
  return RSX_OK;
}

int rsx_sc_make_keys(rsx_sc *h, const double *desc, double *out_ringkey20, double *out_sectorkey60) try {
  if (!h || !desc) return fail(RSX_ERR_BAD_ARG, "null arg");
  double o[NR + NS];
  RSX_TRY(helper_call(h, 0, desc, DS, nullptr, 0, o, NR + NS, nullptr));
  if (out_ringkey20) std::memcpy(out_ringkey20, o, NR * sizeof(double));
  if (out_sectorkey60) std::memcpy(out_sectorkey60, o + NR, NS * sizeof(double));
  return RSX_OK;
} RSX_CATCH_ALL

int rsx_sc_dist_direct(rsx_sc *h, const double *sc1, const double *sc2, double *out_dist) try {
  if (!h || !sc1 || !sc2 || !out_dist) return fail(RSX_ERR_BAD_ARG, "null arg");
  return helper_call(h, 1, sc1, DS, sc2, DS, out_dist, 1, nullptr);
} RSX_CATCH_ALL

int rsx_sc_fast_align(rsx_sc *h, const double *vkey1, const double *vkey2, int32_t *out_shift) try {
  if (!h || !vkey1 || !vkey2 || !out_shift) return fail(RSX_ERR_BAD_ARG, "null arg");
  return helper_call(h, 2, vkey1, NS, vkey2, NS, nullptr, 0, out_shift);
} RSX_CATCH_ALL

int rsx_sc_distance(rsx_sc *h, const double *sc1, const double *sc2, double *out_dist, int32_t *out_shift) try {
  if (!h || !sc1 || !sc2 || !out_dist || !out_shift) return fail(RSX_ERR_BAD_ARG, "null arg");
  return helper_call(h, 3, sc1, DS, sc2, DS, out_dist, 1, out_shift);
} RSX_CATCH_ALL

int rsx_sc_make_scancontext(rsx_sc *h, const void *pts, size_t n, size_t stride_bytes, double *out_desc) try {
  if (!h || (!pts && n) || !out_desc) return fail(RSX_ERR_BAD_ARG, "null arg");
  if (stride_bytes < 12 || (stride_bytes & 3)) return fail(RSX_ERR_BAD_ARG, "stride_bytes must be >= 12 and a multiple of 4");
  std::lock_guard<std::mutex> lk(h->mu);
  RSX_TRY(set_device(h));
  hipStream_t s = h->stream;
  const size_t bytes = n * stride_bytes;
  RSX_TRY(h->pts_ws.reserve(bytes ? bytes : 16, s, false));
  if (bytes) RSX_HIP(hipMemcpyAsync(h->pts_ws.p, pts, bytes, hipMemcpyHostToDevice, s));
  RSX_TRY(h->helper_ws.reserve((2 * DS + 128) * sizeof(double), s, false));
  float *d_desc = h->helper_ws.as<float>();                        // 1200 floats
  double *d_vk = reinterpret_cast<double *>(d_desc + DS);          // + keys (discarded)
  RSX_TRY(launch_build(h->pts_ws.p, (int64_t)n, (int64_t)stride_bytes, h->p.lidar_height, h->p.max_radius, d_desc, d_vk,
                       d_vk + NS, reinterpret_cast<float *>(d_vk + 2 * NS), s, h->p.sum_order));
  float f[DS];
  RSX_HIP(hipMemcpyAsync(f, d_desc, sizeof(f), hipMemcpyDeviceToHost, s));
  RSX_HIP(hipStreamSynchronize(s));
  for (int i = 0; i < DS; i++) out_desc[i] = (double)f[i];
  return RSX_OK;
} RSX_CATCH_ALL

int rsx_sc_detect_between_session(rsx_sc *h, const float *curr_key20, const double *curr_desc, int32_t *loop_id,
                                  float *yaw, double *min_dist, int32_t *nn_idx) try {
  if (!h || !curr_key20 || !curr_desc) return fail(RSX_ERR_BAD_ARG, "null arg");
  float f[DS];
  for (int i = 0; i < DS; i++) {
    f[i] = (float)curr_desc[i];
    if (!((double)f[i] == curr_desc[i])) return fail(RSX_ERR_NOT_FP32_EXACT, "query descriptor element %d is not fp32-exact", i);
  }
  std::unique_lock<std::mutex> lk(h->mu);
  if (h->p.shard_world != 1) return fail(RSX_ERR_BAD_ARG, "unsharded handles only");
  if (h->n_global == 0) return fail(RSX_ERR_RANGE, "empty database");  // reference asserts (KDA.h:61)
  RSX_TRY(set_device(h));
  const int64_t batch = h->batch_made ? h->batch_size : h->n_global;  // SC.cpp:275-284
  RSX_TRY(ensure_tree(h, &h->tree_batch, batch, lk));
  if (batch > h->n_global) return fail(RSX_ERR_RANGE, "the database was replaced while the ring-key tree was being built");
  if (!h->batch_made) {  // committed only once the tree stands
    h->batch_size = batch;
    h->batch_made = true;
  }
  hipStream_t s = h->stream;
  RSX_TRY(h->q_desc.reserve(sizeof(f) + 128, s, false));
  RSX_HIP(hipMemcpyAsync(h->q_desc.p, f, sizeof(f), hipMemcpyHostToDevice, s));
  float *d_key = reinterpret_cast<float *>(static_cast<char *>(h->q_desc.p) + sizeof(f));
  RSX_HIP(hipMemcpyAsync(d_key, curr_key20, NR * sizeof(float), hipMemcpyHostToDevice, s));
  QueryView qv;
  RSX_TRY(prepare_queries(h, h->q_desc.as<float>(), 1, s, &qv));
  return score_candidates_and_finish(h, qv, d_key, h->batch_size, &h->tree_batch, RSX_SC_MODE_CANDIDATE, loop_id, yaw, min_dist, nn_idx);
} RSX_CATCH_ALL

static int query_device_locked(rsx_sc *h, const float *d_q, int32_t nq, int32_t k, int64_t n_eligible, rsx_sc_hit *d_out,
                               hipStream_t s) {
  const int64_t items = local_count_below(h, n_eligible);
  if (use_q1(h, nq, items))  // builds the query's keys and images itself: no keys launch
    return run_q1(h, d_q, nq, items, n_eligible < 0 ? h->n_global : n_eligible, nullptr, k, d_out, s);
  QueryView qv;
  RSX_TRY(prepare_queries(h, d_q, nq, s, &qv));
  return run_topk(h, qv, items, n_eligible < 0 ? h->n_global : n_eligible, nullptr, k, d_out, s);
}

int rsx_sc_query_device(rsx_sc *h, const float *d_q, int32_t nq, int32_t k, int64_t n_eligible, rsx_sc_hit *d_out, void *stream) try {
  if (!h || !d_q || !d_out || nq < 1) return fail(RSX_ERR_BAD_ARG, "bad arg");
  if (k < 1 || k > RSX_SC_MAX_TOPK) return fail(RSX_ERR_BAD_ARG, "k must be in [1,%d]", RSX_SC_MAX_TOPK);
  std::lock_guard<std::mutex> lk(h->mu);
  RSX_TRY(set_device(h));
  hipStream_t s;
  RSX_TRY(use_stream(h, stream, &s));
  return query_device_locked(h, d_q, nq, k, n_eligible, d_out, s);
} RSX_CATCH_ALL

// How the host-buffer entry cuts a batch into pieces: the upload of the first piece is the only one nothing hides, every
// later piece is `growth` times the one before -- a piece is filtered ~2.4x slower than PCIe delivers the next (8192 queries vs
// 10 000 keyframes on MI355X: 0.21 ms of filter, 0.09 ms of upload per 1024 queries at the 54 GB/s measured), so the upload
// of piece c + 1 ends before the filter of piece c does.
// Round 6: only the FILTER runs piece by piece; short lists, window previews and re-scoring run once over the whole batch
// (rsx_sc_query, `one_tail`).  Timeline of one call, 512:2.5 = 512 + 1280 + 3200 + 3200 queries, top-10 (rocprofv3 kernel + copy
// trace, tools/host_timeline.py): first upload 50 us + 20 us to the first kernel; filter launches 139 + 313 + 753 + 745 =
// 1950 us (the whole batch in one launch: 1750 -- a piece of 3200 queries fills 234 of the 256 workgroups of its one round, a piece of
// 512 fills 78: the filter amortises a 304-register tile load over >= 16 query tiles per unit and cannot cut finer); the stages
// behind it 826 us as in the resident step; read-back + synchronise ~50 us.  Measured plans (ms per call pageable / pinned,
// fraction of the resident step's 2.58 ms; tools/ab_host_pieces.py, one box): round 5's three chains on two streams 1024:2.5
// 2.98 / 3.00 = 0.87 / 0.87; one tail 1024:2.5 2.96 / 2.92, **512:2.5 2.86 / 2.91 = 0.90 / 0.89**, 512:2.0 2.88 / 2.93, 256:3.0
// 2.86 / 2.91, 256:2.5 2.93 / 3.01, 128:3.0 2.89 / 2.88, 1024:1.5 2.93 / 2.94, 2048:2.0 2.95 / 3.04.  Batches below 4 pieces'
// worth stay whole.  RSX_SC_HOST_PIECES=first[:growth_x10] (experiments build) overrides; first = 0 keeps every batch whole.
int host_pieces(int32_t nq, int32_t *sizes) {
  static const std::pair<int, int> cfg = [] {
    const char *e = rsx::exp_env("RSX_SC_HOST_PIECES");
    std::pair<int, int> r{512, 25};
    if (e && *e) {
      r.first = atoi(e);
      const char *c = strchr(e, ':');
      if (c) r.second = atoi(c + 1);
    }
    return r;
  }();
  return plan::host_pieces(nq, cfg.first, cfg.second, rsx_sc::kMaxPieces, sizes);  // sc_plan.h
}

int rsx_sc_query(rsx_sc *h, const float *q, int32_t nq, int32_t k, int64_t n_eligible, rsx_sc_hit *out) try {
  if (!h || !q || !out || nq < 1) return fail(RSX_ERR_BAD_ARG, "bad arg");
  if (k < 1 || k > RSX_SC_MAX_TOPK) return fail(RSX_ERR_BAD_ARG, "k must be in [1,%d]", RSX_SC_MAX_TOPK);
  // one lock for staging, query and read-back: two concurrent callers share q_desc / topk
  std::lock_guard<std::mutex> lk(h->mu);
  RSX_TRY(set_device(h));
  RSX_TRY(h->q_desc.reserve((size_t)nq * DS * sizeof(float), h->stream, false));
  RSX_TRY(h->topk.reserve((size_t)nq * k * sizeof(rsx_sc_hit), h->stream, false));
  int32_t sizes[rsx_sc::kMaxPieces];
  const int np = host_pieces(nq, sizes);
  static const bool two_lanes = [] {  // RSX_SC_HOST_LANES=1 (experiments build): every piece on the main stream
    const char *e = rsx::exp_env("RSX_SC_HOST_LANES");
    return !(e && e[0] == '1');
  }();
  const size_t out_bytes = (size_t)nq * k * sizeof(rsx_sc_hit);
  if (np == 1 && out_bytes <= 4096) {
    // the live detector's size (one query, a few records): the last kernel writes the records straight into pinned host
    // memory -- no read-back to enqueue, one synchronise
    // ... and the query goes up from the handle's own pinned buffer (a memcpy of 4.8 KB per query, then ONE asynchronous DMA):
    // from the caller's pageable buffer the runtime stages the copy itself, 5 us more per call
    const size_t q_bytes = (size_t)nq * DS * sizeof(float);
    const bool small_q = q_bytes <= 8 * DS * sizeof(float);
    RSX_TRY(ensure_pinned(h, 4096 + (small_q ? q_bytes : 0)));
    const void *src = q;
    if (small_q) {
      std::memcpy(static_cast<char *>(h->pinned) + 4096, q, q_bytes);
      src = static_cast<char *>(h->pinned) + 4096;
    }
    RSX_HIP(hipMemcpyAsync(h->q_desc.p, src, q_bytes, hipMemcpyHostToDevice, h->stream));
    RSX_TRY(query_device_locked(h, h->q_desc.as<float>(), nq, k, n_eligible, static_cast<rsx_sc_hit *>(h->pinned), h->stream));
    RSX_HIP(hipStreamSynchronize(h->stream));
    std::memcpy(out, h->pinned, out_bytes);
    return RSX_OK;
  }
  if (np == 1) {
    RSX_HIP(hipMemcpyAsync(h->q_desc.p, q, (size_t)nq * DS * sizeof(float), hipMemcpyHostToDevice, h->stream));
    RSX_TRY(query_device_locked(h, h->q_desc.as<float>(), nq, k, n_eligible, h->topk.as<rsx_sc_hit>(), h->stream));
  } else {
    // Piece c + 1 goes up on up_stream while piece c is scored.  The order of the calls matters for pageable memory,
    // where hipMemcpyAsync returns only when the runtime has staged the whole piece: the scoring of piece c is enqueued
    // BEFORE that call, so the device works while this thread feeds the copy engine.  Pinned memory
    // (rsx_host_alloc_pinned) makes the copies asynchronous as well and the whole batch is enqueued at once.
    // Pieces are scored alternately on two streams, each with its own workspace set: a piece ends in short dependent
    // kernels (select, window previews, re-scoring rounds: latency, not throughput) that the next piece's filter
    // launch fills in -- in one stream the pieces of the bench batch cost 3.27 ms of kernels against 2.80 ms whole.
    if (!h->up_stream) {
      RSX_HIP(hipStreamCreateWithFlags(&h->up_stream, hipStreamNonBlocking));
      RSX_HIP(hipStreamCreateWithFlags(&h->stream_b, hipStreamNonBlocking));
      for (auto &e : h->up_ev) RSX_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      RSX_HIP(hipEventCreateWithFlags(&h->lane_ev, hipEventDisableTiming));
    }
    auto upload = [&](int c, int64_t q0) -> int {
      RSX_HIP(hipMemcpyAsync(h->q_desc.as<float>() + q0 * DS, q + q0 * DS, (size_t)sizes[c] * DS * sizeof(float),
                             hipMemcpyHostToDevice, h->up_stream));
      RSX_HIP(hipEventRecord(h->up_ev[c], h->up_stream));
      return RSX_OK;
    };
    // Round 6: where the whole batch goes through ONE filter workspace (the usual case: 8192 queries against 10 000 entries), only
    // the FILTER runs piece by piece -- keys, query images and bound rows of piece c while piece c + 1 goes up -- and the stages
    // behind it (short lists, window previews, re-scoring) run ONCE over the whole batch: their fixed costs (~90 us of dependent
    // small kernels per piece) are what the piecewise chains paid three times.  RSX_SC_HOST_TAIL=pieces (experiments build)
    // keeps the chains.
    static const bool one_tail_allowed = [] {
      const char *e = rsx::exp_env("RSX_SC_HOST_TAIL");
      return !(e && e[0] == 'p');
    }();
    const int64_t items = local_count_below(h, n_eligible), elig_all = n_eligible < 0 ? h->n_global : n_eligible;
    const bool one_tail = one_tail_allowed && !use_q1(h, nq, items) && use_filter(h, nq, items) && filter_batch(items, nq) >= nq;
    auto filter_pieces_one_tail = [&]() -> int {
      hipStream_t s = h->stream;
      h->w = &h->ws[0];
      RSX_TRY(filter_reserve(h, items, nq, s));
      QueryView all;
      {  // (prepare_queries without its launch: the keys are built piece by piece)
        h->st.valid = false;
        RSX_TRY(h->w->q_vkey.reserve((size_t)nq * NS * sizeof(double), s, false));
        RSX_TRY(h->w->q_norm.reserve((size_t)nq * NS * sizeof(double), s, false));
        RSX_TRY(h->w->q_rkey.reserve((size_t)nq * NR * sizeof(float), s, false));
        all.desc = h->q_desc.as<float>();
        all.vkey = h->w->q_vkey.as<double>();
        all.norm = h->w->q_norm.as<double>();
        all.nq = nq;
      }
      const int64_t ld = (items + 31) / 32 * 32;
      lb_t *lb = h->w->f_lb.as<lb_t>();
      RSX_TRY(upload(0, 0));
      int64_t q0 = 0;
      for (int c = 0; c < np; c++) {
        RSX_HIP(hipStreamWaitEvent(s, h->up_ev[c], 0));
        RSX_TRY(launch_keys(all.desc + q0 * DS, sizes[c], h->w->q_vkey.as<double>() + q0 * NS, h->w->q_norm.as<double>() + q0 * NS,
                            h->w->q_rkey.as<float>() + q0 * NR, s, h->p.sum_order));
        QueryView piece = all;
        piece.desc = all.desc + q0 * DS;
        piece.vkey = all.vkey + q0 * NS;
        piece.norm = all.norm + q0 * NS;
        piece.nq = sizes[c];
        RSX_TRY(run_filter(h, piece, items, lb + q0 * ld, ld, nullptr, s));  // (the query images of a piece are consumed by its own launch)
        q0 += sizes[c];
        if (c + 1 < np) RSX_TRY(upload(c + 1, q0));
      }
      const DbView db = db_view(h);
      RSX_TRY(launch_select(db, lb, ld, items, nq, elig_all, nullptr, first_round_target(), h->w->f_cand.as<RescoreEntry>(),
                            h->w->f_cnt.as<int32_t>(), h->w->f_thr.as<float>(), s));
      if (use_window())
        RSX_TRY(launch_window(db, all, h->w->f_wimg.p, h->w->f_cand.as<RescoreEntry>(), h->w->f_cnt.as<int32_t>(), k, filter_eps(),
                              h->w->f_win.as<WindowPreview>(), s, 0));
      return rescore(h, all, items, elig_all, nullptr, 0, RESCORE_ALL_ROUNDS, nullptr, nullptr, k, h->topk.as<rsx_sc_hit>(), s);
    };
    auto all_pieces = [&]() -> int {
      if (one_tail) return filter_pieces_one_tail();
      // lane B starts behind whatever the caller's earlier calls left on the main stream (DB appends ...)
      RSX_HIP(hipEventRecord(h->lane_ev, h->stream));
      RSX_HIP(hipStreamWaitEvent(h->stream_b, h->lane_ev, 0));
      RSX_TRY(upload(0, 0));
      int64_t q0 = 0;
      for (int c = 0; c < np; c++) {
        hipStream_t lane = two_lanes && (c & 1) ? h->stream_b : h->stream;
        h->w = &h->ws[two_lanes ? (c & 1) : 0];
        RSX_HIP(hipStreamWaitEvent(lane, h->up_ev[c], 0));
        RSX_TRY(query_device_locked(h, h->q_desc.as<float>() + q0 * DS, sizes[c], k, n_eligible,
                                    h->topk.as<rsx_sc_hit>() + q0 * k, lane));
        q0 += sizes[c];
        if (c + 1 < np) RSX_TRY(upload(c + 1, q0));
      }
      RSX_HIP(hipEventRecord(h->lane_ev, h->stream_b));
      RSX_HIP(hipStreamWaitEvent(h->stream, h->lane_ev, 0));
      return RSX_OK;
    };
    struct BackToSetZero {  // also when something below throws (the firewall turns that into a status)
      rsx_sc *h;
      ~BackToSetZero() { h->w = &h->ws[0]; }
    } back{h};
    const int st = all_pieces();
    h->w = &h->ws[0];
    if (st != RSX_OK) {  // nothing of this call may still be in flight when the caller gets its buffers back
      (void)hipStreamSynchronize(h->up_stream);
      (void)hipStreamSynchronize(h->stream_b);
      (void)hipStreamSynchronize(h->stream);
      return st;
    }
  }
  RSX_HIP(hipMemcpyAsync(out, h->topk.p, (size_t)nq * k * sizeof(rsx_sc_hit), hipMemcpyDeviceToHost, h->stream));
  RSX_HIP(hipStreamSynchronize(h->stream));
  return RSX_OK;
} RSX_CATCH_ALL

// ---- filter shards over a replicated database (rsx.h: "filter-shard layout") ----
int rsx_sc_filter_range_device(rsx_sc *h, const float *d_q, int32_t nq, int64_t first_slot, int64_t n_slots, rsx_f16 *d_lb,
                               int64_t ld, void *stream) try {
  if (!h || !d_q || !d_lb || nq < 1) return fail(RSX_ERR_BAD_ARG, "bad arg");
  if (first_slot < 0 || first_slot % 32 || n_slots < 0) return fail(RSX_ERR_BAD_ARG, "the range must start at a multiple of 32 slots");
  if (ld < (n_slots + 31) / 32 * 32 || ld % 8) return fail(RSX_ERR_BAD_ARG, "ld must cover the range rounded up to 32 slots");
  if ((reinterpret_cast<uintptr_t>(d_lb) & 15u) != 0) return fail(RSX_ERR_BAD_ARG, "d_lb must be 16-byte aligned");
  if (n_slots == 0) return RSX_OK;  // an empty range (a rank beyond the entries of a small database) names no slot
  std::lock_guard<std::mutex> lk(h->mu);
  RSX_TRY(set_device(h));
  if (first_slot + n_slots > h->n_local) return fail(RSX_ERR_RANGE, "slots [%lld, %lld) of %lld", (long long)first_slot,
                                                     (long long)(first_slot + n_slots), (long long)h->n_local);
  hipStream_t s;
  RSX_TRY(use_stream(h, stream, &s));
  QueryView qv;
  RSX_TRY(prepare_queries(h, d_q, nq, s, &qv));
  RSX_TRY(h->w->f_qimg.reserve(any_qimg_bytes(nq), s, false));
  return run_filter(h, qv, n_slots, reinterpret_cast<lb_t *>(d_lb), ld, nullptr, s, first_slot);
} RSX_CATCH_ALL

int rsx_sc_query_bounds_device(rsx_sc *h, const float *d_q, int32_t nq, int32_t k, int64_t n_eligible, const rsx_f16 *d_lb_blocks,
                               int32_t n_blocks, int64_t block_ld, int64_t block_stride, rsx_sc_hit *d_out, void *stream) try {
  if (!h || !d_q || !d_lb_blocks || !d_out || nq < 1 || n_blocks < 1) return fail(RSX_ERR_BAD_ARG, "bad arg");
  if (k < 1 || k > RSX_SC_MAX_TOPK) return fail(RSX_ERR_BAD_ARG, "k must be in [1,%d]", RSX_SC_MAX_TOPK);
  if ((reinterpret_cast<uintptr_t>(d_lb_blocks) & 15u) != 0)  // the blocks are read with 16-byte loads
    return fail(RSX_ERR_BAD_ARG, "d_lb_blocks must be 16-byte aligned");
  std::lock_guard<std::mutex> lk(h->mu);
  RSX_TRY(set_device(h));
  if (h->p.shard_world != 1) return fail(RSX_ERR_BAD_ARG, "bounds from filter shards need the whole database in this handle (shard_world 1)");
  hipStream_t s;
  RSX_TRY(use_stream(h, stream, &s));
  QueryView qv;
  RSX_TRY(prepare_queries(h, d_q, nq, s, &qv));
  const int64_t n_elig = n_eligible < 0 ? h->n_global : n_eligible;
  const int64_t n_items = local_count_below(h, n_eligible);
  if (n_items <= 0) return launch_pairs(db_view(h), qv, nullptr, 0, 0, n_elig, nullptr, nullptr, nullptr, nullptr, d_out, k, s);
  const int64_t ld = (n_items + 31) / 32 * 32;
  if (block_ld < 32 || block_ld % 32 || (int64_t)n_blocks * block_ld < ld || block_stride < (int64_t)nq * block_ld || block_stride % 8)
    return fail(RSX_ERR_BAD_ARG, "%d blocks of %lld columns do not cover %lld eligible entries", n_blocks, (long long)block_ld,
                (long long)n_items);
  const int64_t qb = filter_batch(n_items, nq);
  RSX_TRY(filter_reserve(h, n_items, qb, s));
  const DbView db = db_view(h);
  for (int64_t b0 = 0; b0 < nq; b0 += qb) {
    QueryView q = qv;
    q.desc = qv.desc + b0 * DS;
    q.vkey = qv.vkey + b0 * NS;
    q.norm = qv.norm + b0 * NS;
    q.nq = (int32_t)((nq - b0 < qb) ? (nq - b0) : qb);
    RSX_TRY(launch_gather_bounds(reinterpret_cast<const lb_t *>(d_lb_blocks), block_ld, block_stride, b0, q.nq, h->w->f_lb.as<lb_t>(), ld, s));
    RSX_TRY(launch_select(db, h->w->f_lb.as<lb_t>(), ld, n_items, q.nq, n_elig, nullptr, first_round_target(), h->w->f_cand.as<RescoreEntry>(),
                          h->w->f_cnt.as<int32_t>(), h->w->f_thr.as<float>(), s));
    if (use_window())
      RSX_TRY(launch_window(db, q, h->w->f_wimg.p, h->w->f_cand.as<RescoreEntry>(), h->w->f_cnt.as<int32_t>(), k, filter_eps(),
                            h->w->f_win.as<WindowPreview>(), s));
    RSX_TRY(rescore(h, q, n_items, n_elig, nullptr, 0, RESCORE_ALL_ROUNDS, nullptr, nullptr, k, d_out + b0 * k, s));
  }
  return RSX_OK;
} RSX_CATCH_ALL

int rsx_sc_query_stage1_device(rsx_sc *h, const float *d_q, int32_t nq, int32_t k, int64_t n_eligible,
                               rsx_sc_hit *d_partial, void *stream) try {
  return rsx_sc_query_stage1_elig_device(h, d_q, nq, k, n_eligible, nullptr, 0, d_partial, stream);
} RSX_CATCH_ALL

int rsx_sc_query_stage1_elig_device(rsx_sc *h, const float *d_q, int32_t nq, int32_t k, int64_t n_eligible,
                                    const int64_t *d_q_elig, int32_t elig_monotone, rsx_sc_hit *d_partial, void *stream) try {
  if (!h || !d_q || !d_partial || nq < 1) return fail(RSX_ERR_BAD_ARG, "bad arg");
  if (k < 1 || k > RSX_SC_MAX_TOPK) return fail(RSX_ERR_BAD_ARG, "k must be in [1,%d]", RSX_SC_MAX_TOPK);
  std::lock_guard<std::mutex> lk(h->mu);
  RSX_TRY(set_device(h));
  hipStream_t s;
  RSX_TRY(use_stream(h, stream, &s));
  h->st.valid = false;
  QueryView qv;
  RSX_TRY(prepare_queries(h, d_q, nq, s, &qv));
  const int64_t items = local_count_below(h, n_eligible);
  const int64_t n_elig = n_eligible < 0 ? h->n_global : n_eligible;
  RSX_TRY(h->st_partial.reserve((size_t)nq * k * sizeof(rsx_sc_hit), s, false));
  const bool filtered = use_filter(h, nq, items) && filter_batch(items, nq) >= nq;
  int32_t window_head = 0;
  if (filtered) {
    // round 0 only: this shard's share of the ~160 lowest bounds per query -- enough for the merged
    // k-th distance to be the final one for almost every query (a single GPU needs ~120 exact scores
    // per query because it can tighten tau after every 64; a shard cannot see the others' hits until
    // the exchange, so stage 1 over-samples instead).  Measured per-rank cost of both stages with 8
    // shards of the 10k DB, 8192 queries: 64 -> 3.4 ms, 128 -> 1.9, 192 -> 1.7, 256 -> 1.8.
    static const int stage1_total = [] {
      const char *e = rsx::exp_env("RSX_SC_STAGE1_TOTAL");  // tuning knob: lowest bounds scored in stage 1, over all shards
      const int v = (e && *e) ? atoi(e) : 0;
      return v > 0 ? v : 160;
    }();
    int32_t first = (stage1_total + h->p.shard_world - 1) / h->p.shard_world;
    if (first < 8) first = 8;
    if (first > 128) first = 128;
    RSX_TRY(filter_reserve(h, items, nq, s));
    // Round 5 experiment, OFF by default (RSX_SC_WINDOW_HEAD=1 in the experiments build): a DB shard's window kernel scores only
    // the head of the list that round 0 re-scores -- S shards previewing 128 list positions each are S times one GPU's window
    // kernel -- and stage 2 makes the records behind the head ON DEMAND, for the positions whose bound can still reach the k-th
    // best distance of the merged lists (sc_window_tail_kernel).  Emulated per-rank compute, 1 x 8, 8192 queries against a
    // 10 000-entry random DB (tools/bench_layouts.py): the whole head in stage 1 1.20 ms; head only and NO tail (stage 2 pays a
    // VALU preview, one entry per wavefront, per candidate) 6.87 ms; head only + tail on demand 1.52 ms -- a second launch of a
    // workgroup per query that stages the query's images again costs more than the previews it saves.
    static const bool head_only = [] {
      const char *e = rsx::exp_env("RSX_SC_WINDOW_HEAD");
      return e && e[0] == '1';
    }();
    window_head = (head_only && h->p.shard_world > 1 && use_window()) ? first : 0;
    RSX_TRY(filter_and_select(h, qv, items, n_elig, d_q_elig, first, k, s, elig_monotone != 0, window_head));
    RSX_TRY(rescore(h, qv, items, n_elig, d_q_elig, 0, 1, nullptr, nullptr, k, h->st_partial.as<rsx_sc_hit>(), s));
  } else {
    RSX_TRY(run_topk(h, qv, items, n_elig, d_q_elig, k, h->st_partial.as<rsx_sc_hit>(), s, elig_monotone != 0));  // complete already
  }
  RSX_HIP(hipMemcpyAsync(d_partial, h->st_partial.p, (size_t)nq * k * sizeof(rsx_sc_hit), hipMemcpyDeviceToDevice, s));
  h->st.valid = true;
  h->st.filtered = filtered;
  h->st.window_head = window_head;
  h->st.nq = nq;
  h->st.k = k;
  h->st.n_items = items;
  h->st.n_eligible = n_elig;
  h->st.q_elig = d_q_elig;
  h->st.qv = qv;
  return RSX_OK;
} RSX_CATCH_ALL

int rsx_sc_query_stage2_device(rsx_sc *h, int32_t nq, int32_t k, const rsx_sc_hit *d_global, rsx_sc_hit *d_out,
                               void *stream) try {
  if (!h || !d_global || !d_out) return fail(RSX_ERR_BAD_ARG, "null arg");
  std::lock_guard<std::mutex> lk(h->mu);
  if (!h->st.valid || h->st.nq != nq || h->st.k != k)
    return fail(RSX_ERR_BAD_ARG, "stage 2 without a matching stage 1 (nq=%d k=%d)", nq, k);
  RSX_TRY(set_device(h));
  hipStream_t s;
  RSX_TRY(use_stream(h, stream, &s));
  h->st.valid = false;
  if (h->st.filtered) {
    if (h->st.window_head > 0)
      RSX_TRY(launch_window_tail(db_view(h), nq, h->w->f_wimg.p, h->w->f_cand.as<RescoreEntry>(), h->w->f_cnt.as<int32_t>(), k, filter_eps(),
                                 h->w->f_win.as<WindowPreview>(), h->st.window_head, d_global, s));
    return rescore(h, h->st.qv, h->st.n_items, h->st.n_eligible, h->st.q_elig, 1, RESCORE_ALL_ROUNDS, d_global,
                   h->st_partial.as<rsx_sc_hit>(), k, d_out, s);
  }
  RSX_HIP(hipMemcpyAsync(d_out, h->st_partial.p, (size_t)nq * k * sizeof(rsx_sc_hit), hipMemcpyDeviceToDevice, s));
  return RSX_OK;
} RSX_CATCH_ALL

int rsx_sc_query_self_device(rsx_sc *h, int64_t q_first, int32_t nq, int32_t k, int64_t n_eligible, int32_t exclude_recent,
                             rsx_sc_hit *d_out, void *stream) try {
  if (!h || !d_out || nq < 1) return fail(RSX_ERR_BAD_ARG, "bad arg");
  if (k < 1 || k > RSX_SC_MAX_TOPK) return fail(RSX_ERR_BAD_ARG, "k must be in [1,%d]", RSX_SC_MAX_TOPK);
  std::lock_guard<std::mutex> lk(h->mu);
  if (h->p.shard_world != 1) return fail(RSX_ERR_BAD_ARG, "query_self needs an unsharded handle (queries must be local)");
  if (q_first < 0 || q_first + nq > h->n_global) return fail(RSX_ERR_RANGE, "query range out of bounds");
  RSX_TRY(set_device(h));
  hipStream_t s;
  RSX_TRY(use_stream(h, stream, &s));
  QueryView qv;
  qv.desc = h->desc.as<float>() + q_first * DS;
  qv.vkey = h->vkey.as<double>() + q_first * NS;
  qv.norm = h->norm.as<double>() + q_first * NS;
  qv.nq = nq;
  if (n_eligible < 0 || n_eligible > h->n_global) n_eligible = h->n_global;
  const int64_t *d_elig = nullptr;
  if (exclude_recent >= 0) {
    std::vector<int64_t> e((size_t)nq);
    for (int i = 0; i < nq; i++) {
      int64_t v = q_first + i - exclude_recent;
      e[(size_t)i] = v < 0 ? 0 : v;
    }
    RSX_TRY(h->q_elig.reserve((size_t)nq * sizeof(int64_t), s, false));
    RSX_HIP(hipMemcpyAsync(h->q_elig.p, e.data(), (size_t)nq * sizeof(int64_t), hipMemcpyHostToDevice, s));
    RSX_HIP(hipStreamSynchronize(s));  // e is a stack/heap temporary
    d_elig = h->q_elig.as<int64_t>();
  }
  // the limits q_first + i - exclude_recent grow with i: the filter skips what a query cannot see
  return run_topk(h, qv, local_count_below(h, n_eligible), n_eligible, d_elig, k, d_out, s, /*elig_monotone=*/true);
} RSX_CATCH_ALL

int rsx_sc_pair_distances(rsx_sc *h, const float *q_desc, int64_t first, int64_t count, double *out_dist, int32_t *out_shift) try {
  if (!h || !q_desc || !out_dist || !out_shift) return fail(RSX_ERR_BAD_ARG, "null arg");
  std::lock_guard<std::mutex> lk(h->mu);
  if (first < 0 || count < 0 || first + count > h->n_local) return fail(RSX_ERR_RANGE, "range out of bounds");
  if (count == 0) return RSX_OK;
  RSX_TRY(set_device(h));
  hipStream_t s = h->stream;
  RSX_TRY(h->q_desc.reserve(DS * sizeof(float), s, false));
  RSX_HIP(hipMemcpyAsync(h->q_desc.p, q_desc, DS * sizeof(float), hipMemcpyHostToDevice, s));
  QueryView qv;
  RSX_TRY(prepare_queries(h, h->q_desc.as<float>(), 1, s, &qv));
  RSX_TRY(h->pair_out.reserve((size_t)count * (sizeof(double) + sizeof(int32_t)), s, false));
  double *d_dist = h->pair_out.as<double>();
  int32_t *d_shift = reinterpret_cast<int32_t *>(d_dist + count);
  RSX_TRY(launch_pairs(db_view(h), qv, nullptr, first, count, -1, nullptr, d_dist, d_shift, nullptr, nullptr, 0, s));
  RSX_HIP(hipMemcpyAsync(out_dist, d_dist, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, s));
  RSX_HIP(hipMemcpyAsync(out_shift, d_shift, (size_t)count * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  RSX_HIP(hipStreamSynchronize(s));
  return RSX_OK;
} RSX_CATCH_ALL

int rsx_sc_filter_bounds(rsx_sc *h, const float *q_descs, int32_t nq, float *out_lb) try {
  if (!h || !q_descs || !out_lb || nq < 1) return fail(RSX_ERR_BAD_ARG, "bad arg");
  std::lock_guard<std::mutex> lk(h->mu);
  RSX_TRY(set_device(h));
  hipStream_t s = h->stream;
  const int64_t n = h->n_local;
  if (n == 0) return RSX_OK;
  const int64_t ld = (n + 31) / 32 * 32;
  RSX_TRY(h->q_desc.reserve((size_t)nq * DS * sizeof(float), s, false));
  RSX_HIP(hipMemcpyAsync(h->q_desc.p, q_descs, (size_t)nq * DS * sizeof(float), hipMemcpyHostToDevice, s));
  QueryView qv;
  RSX_TRY(prepare_queries(h, h->q_desc.as<float>(), nq, s, &qv));
  RSX_TRY(h->w->f_qimg.reserve(any_qimg_bytes(nq), s, false));
  RSX_TRY(h->w->f_lb.reserve((size_t)nq * ld * sizeof(lb_t), s, false));
  RSX_TRY(run_filter(h, qv, n, h->w->f_lb.as<lb_t>(), ld, nullptr, s));
  // the matrix is fp16 on the device (sc_kernels.h lb_t); this diagnostic entry hands out the same values as float
  std::vector<lb_t> host((size_t)nq * (size_t)n);
  RSX_HIP(hipMemcpy2DAsync(host.data(), (size_t)n * sizeof(lb_t), h->w->f_lb.p, (size_t)ld * sizeof(lb_t), (size_t)n * sizeof(lb_t),
                           (size_t)nq, hipMemcpyDeviceToHost, s));
  RSX_HIP(hipStreamSynchronize(s));
  for (size_t i = 0; i < host.size(); i++) out_lb[i] = (float)host[i];
  return RSX_OK;
} RSX_CATCH_ALL

double rsx_sc_filter_eps(void) { return filter_eps(); }

int rsx_sc_window_previews(rsx_sc *h, const float *q_descs, int32_t nq, int32_t k, int32_t *out_slots, float *out_pv,
                           int32_t *out_kstar, int32_t *out_shift_mask, int32_t *out_counts) try {
  if (!h || !q_descs || !out_slots || !out_pv || !out_kstar || !out_shift_mask || !out_counts || nq < 1)
    return fail(RSX_ERR_BAD_ARG, "bad arg");
  if (k < 1 || k > RSX_SC_MAX_TOPK) return fail(RSX_ERR_BAD_ARG, "k=%d out of range [1,%d]", k, RSX_SC_MAX_TOPK);
  std::lock_guard<std::mutex> lk(h->mu);
  RSX_TRY(set_device(h));
  hipStream_t s = h->stream;
  const int64_t n = h->n_local;
  for (int32_t q = 0; q < nq; q++) out_counts[q] = 0;
  if (n == 0) return RSX_OK;
  RSX_TRY(h->q_desc.reserve((size_t)nq * DS * sizeof(float), s, false));
  RSX_HIP(hipMemcpyAsync(h->q_desc.p, q_descs, (size_t)nq * DS * sizeof(float), hipMemcpyHostToDevice, s));
  QueryView qv;
  RSX_TRY(prepare_queries(h, h->q_desc.as<float>(), nq, s, &qv));
  RSX_TRY(filter_reserve(h, n, nq, s));
  RSX_HIP(hipMemsetAsync(h->w->f_win.p, 0xff, (size_t)nq * WINDOW_P * sizeof(WindowPreview), s));
  RSX_TRY(filter_and_select(h, qv, n, h->n_global, nullptr, 128, k, s));
  std::vector<RescoreEntry> sl((size_t)nq * RESCORE_SHORTLIST_CAP);
  std::vector<WindowPreview> wp((size_t)nq * WINDOW_P);
  std::vector<int32_t> cnt((size_t)nq);
  RSX_HIP(hipMemcpyAsync(sl.data(), h->w->f_cand.p, sl.size() * sizeof(RescoreEntry), hipMemcpyDeviceToHost, s));
  RSX_HIP(hipMemcpyAsync(wp.data(), h->w->f_win.p, wp.size() * sizeof(WindowPreview), hipMemcpyDeviceToHost, s));
  RSX_HIP(hipMemcpyAsync(cnt.data(), h->w->f_cnt.p, cnt.size() * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  RSX_HIP(hipStreamSynchronize(s));
  for (int32_t q = 0; q < nq; q++) {
    const int32_t c = cnt[(size_t)q] < WINDOW_P ? cnt[(size_t)q] : WINDOW_P;
    out_counts[q] = c;
    for (int32_t i = 0; i < WINDOW_P; i++) {
      const size_t o = (size_t)q * WINDOW_P + i;
      out_slots[o] = i < c ? sl[(size_t)q * RESCORE_SHORTLIST_CAP + i].slot : -1;
      out_pv[o] = wp[o].pv;
      out_kstar[o] = wp[o].ks >= 0 ? (wp[o].ks & 63) : wp[o].ks;
      out_shift_mask[o] = wp[o].ks >= 0 ? ((wp[o].ks >> 8) & 0x7f) : 0;
    }
  }
  return RSX_OK;
} RSX_CATCH_ALL

int rsx_selftest_firewall(int kind) try {
  if (kind == 0) {
    std::vector<double> v;
    v.reserve(v.max_size() + 1);  // std::length_error
  } else if (kind == 1) {
    throw std::bad_alloc();
  } else if (kind == 2) {
    throw std::runtime_error("KDTreeSingleIndexAdaptor: selftest");
  } else if (kind == 3) {
    throw 42;
  }
  return RSX_OK;
} RSX_CATCH_ALL

int rsx_sc_merge_topk(const rsx_sc_hit *parts, int32_t nparts, int32_t nq, int32_t k, rsx_sc_hit *out) try {
  if (!parts || !out || nparts < 1 || nq < 1 || k < 1) return fail(RSX_ERR_BAD_ARG, "bad arg");
  for (int q = 0; q < nq; q++) {
    rsx_sc_hit *o = out + (size_t)q * k;
    int count = 0;
    for (int p = 0; p < nparts; p++)
      for (int i = 0; i < k; i++) {
        const rsx_sc_hit &r = parts[((size_t)p * nq + q) * k + i];
        if (!(r.dist < 10000000)) continue;  // padding / never-a-hit (SC.cpp:388)
        int pos = count < k ? count : k;
        while (pos > 0 && ((r.dist < o[pos - 1].dist) || (r.dist == o[pos - 1].dist && r.index < o[pos - 1].index))) pos--;
        if (pos >= k) continue;
        int last = count < k ? count : k - 1;
        for (int j = last; j > pos; j--) o[j] = o[j - 1];
        o[pos] = r;
        if (count < k) count++;
      }
    for (int i = count; i < k; i++) {
      o[i].dist = 10000000;
      o[i].index = 0;
      o[i].shift = 0;
    }
  }
  return RSX_OK;
} RSX_CATCH_ALL

int rsx_sc_merge_topk_device(rsx_sc *h, const rsx_sc_hit *d_parts, int32_t nparts, int32_t nq, int32_t k,
                             rsx_sc_hit *d_out, void *stream) try {
  if (!h || !d_parts || !d_out || nparts < 1 || nq < 1 || k < 1 || k > RSX_SC_MAX_TOPK) return fail(RSX_ERR_BAD_ARG, "bad arg");
  std::lock_guard<std::mutex> lk(h->mu);
  RSX_TRY(set_device(h));
  hipStream_t s;
  RSX_TRY(use_stream(h, stream, &s));
  return launch_merge(d_parts, nparts, nq, k, d_out, s);
} RSX_CATCH_ALL

int rsx_sc_profile_enable(rsx_sc *h, int on) try {
  if (!h) return fail(RSX_ERR_BAD_ARG, "null handle");
  std::lock_guard<std::mutex> lk(h->mu);
  RSX_TRY(set_device(h));
  if (on && !h->prof.ev) {
    h->prof.ev = new (std::nothrow) hipEvent_t[2 * PairProfiler::kMax];
    if (!h->prof.ev) return fail(RSX_ERR_OOM, "host alloc");
    for (int i = 0; i < 2 * PairProfiler::kMax; i++) RSX_HIP(hipEventCreate(&h->prof.ev[i]));
  }
  if (on) {
    RSX_TRY(h->stats.reserve(kStatBytes, h->stream, false));
    RSX_HIP(hipMemsetAsync(h->stats.p, 0, kStatBytes, h->stream));
    RSX_HIP(hipStreamSynchronize(h->stream));
  }
  h->prof.on = on != 0;
  h->prof.used = 0;
  return RSX_OK;
} RSX_CATCH_ALL

int rsx_sc_profile_read_rescoring(rsx_sc *h, rsx_sc_rescoring_stats *out) try {
  if (!h || !out || out->struct_size < offsetof(rsx_sc_rescoring_stats, candidates)) return fail(RSX_ERR_BAD_ARG, "null arg / struct_size not set");
  int64_t out6[6];
  int64_t *const out5 = out6;
  int64_t *candidates = out5, *exact_evals = out5 + 1, *queries_rescored = out5 + 2;
  // copies the fields that fit into the caller's struct_size on every return path
  struct Publish {
    rsx_sc_rescoring_stats *o;
    const int64_t *v;
    ~Publish() {
      int64_t *dst = &o->candidates;
      const size_t n = (o->struct_size - offsetof(rsx_sc_rescoring_stats, candidates)) / sizeof(int64_t);
      for (size_t i = 0; i < n && i < 6; i++) dst[i] = v[i];
    }
  } publish{out, out6};
  std::lock_guard<std::mutex> lk(h->mu);
  RSX_TRY(set_device(h));
  for (int i = 0; i < 6; i++) out6[i] = 0;
  if (!h->stats.p) return RSX_OK;
  RSX_HIP(hipDeviceSynchronize());  // the counters are bumped by kernels on the caller's stream
  unsigned long long v[RESCORE_STAT_WORDS] = {0};
  {
    std::vector<unsigned long long> all((size_t)RESCORE_STAT_COPIES * RESCORE_STAT_WORDS);
    RSX_HIP(hipMemcpy(all.data(), h->stats.p, kStatBytes, hipMemcpyDeviceToHost));
    RSX_HIP(hipMemset(h->stats.p, 0, kStatBytes));
    for (int c = 0; c < RESCORE_STAT_COPIES; c++)
      for (int i = 0; i < RESCORE_STAT_WORDS; i++) v[i] += all[(size_t)c * RESCORE_STAT_WORDS + i];
  }
  if (rsx::exp_env("RSX_RESCORE_PROF") && v[1])  // region cycles of wave 0, averaged per scoring workgroup
    fprintf(stderr, "[sc_rescore prof] per query (cycles of wave 0): load %.0f  phaseA %.0f  mergeA %.0f  phaseB %.0f  mergeX %.0f  gather %.0f  total %.0f\n",
            (double)v[4] / v[1], (double)v[5] / v[1], (double)v[6] / v[1], (double)v[7] / v[1], (double)v[8] / v[1], (double)v[9] / v[1],
            (double)v[10] / v[1]);
  if (rsx::exp_env("RSX_RESCORE_PROF") && v[1] && v[13])
    fprintf(stderr, "[sc_rescore prof] wave kernel: header %.0f  records %.0f (then tau_ub = the 'phaseA' figure)\n", (double)v[13] / v[1], (double)v[14] / v[1]);
  *candidates = (int64_t)v[0];
  *exact_evals = v[2] ? (int64_t)v[2] : (int64_t)v[0];  // one-pass scoring: every candidate is an exact evaluation
  *queries_rescored = (int64_t)v[1];
  out5[3] = (int64_t)v[3];   // candidates whose alignment + preview came from the window kernel (sc_window.hip)
  out5[4] = (int64_t)v[11];  // candidates that went through the VALU alignment + fp32 preview / exact alignments (wave kernel)
  out6[5] = v[12] ? (int64_t)v[12] : 7 * *exact_evals;  // window shifts evaluated exactly (7 per evaluation without a shift mask)
  return RSX_OK;
} RSX_CATCH_ALL

int rsx_sc_profile_read(rsx_sc *h, int64_t *launches, double *total_ms) try {
  if (!h || !launches || !total_ms) return fail(RSX_ERR_BAD_ARG, "null arg");
  std::lock_guard<std::mutex> lk(h->mu);
  RSX_TRY(set_device(h));
  *launches = 0;
  *total_ms = 0.0;
  for (int i = 0; i < h->prof.used; i++) {
    RSX_HIP(hipEventSynchronize(h->prof.ev[2 * i + 1]));
    float ms = 0.f;
    RSX_HIP(hipEventElapsedTime(&ms, h->prof.ev[2 * i], h->prof.ev[2 * i + 1]));
    *total_ms += ms;
    (*launches)++;
  }
  h->prof.used = 0;
  return RSX_OK;
} RSX_CATCH_ALL

int rsx_sc_hit_to_loop(rsx_sc *h, const rsx_sc_hit *hit, int32_t *loop_id, float *yaw) try {
  if (!h || !hit) return fail(RSX_ERR_BAD_ARG, "null arg");
  std::lock_guard<std::mutex> lk(h->mu);
  if (loop_id) *loop_id = (hit->dist < h->p.dist_thres) ? hit->index : -1;  // SC.cpp:401-403
  if (yaw) *yaw = yaw_from_shift(hit->shift);                               // SC.cpp:417
  return RSX_OK;
} RSX_CATCH_ALL

}  // extern "C"
