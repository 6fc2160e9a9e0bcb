// sc_sharded.cpp -- rsx_scs_*: ONE process driving a ScanContext database over several GPUs.
// The reference's loop-closure node is a single C++ process (laserPosegraphOptimization.cpp:99,706-710); this gives
// it all GPUs of the node without MPI / torch.
//
// Layout (the same as navtech_radar_slam_amd/sharded.py): G devices = Q query groups x S DB shards.  Device g belongs
// to query group g / S and holds shard g % S (keyframe i on the shards with index i % S, include/rsx.h sharding rule,
// once per query group).  A batch of queries is cut into Q contiguous slices; inside a group the slice runs the
// two-stage query of rsx.h against the group's S shards with one exchange of 16-byte records per stage:
//   exchange = peer copies (default): every shard's list goes to the group's first device with hipMemcpyPeerAsync over
//              xGMI, is merged there (rsx_sc_merge_topk_device) and the merged bound is copied back for stage 2 --
//              2 (S - 1) small copies per stage, no extra library in the process;
//   exchange = RCCL: ncclAllGather inside one ncclGroupStart / ncclGroupEnd over the group's communicators (one per
//              device, ncclCommInitAll), every shard merges for itself -- the collective BASELINE.json's north_star
//              names.  librccl.so is loaded with dlopen on first use, so a host that never asks for it does not link it.
// Same records, same merge, same results as one GPU in every layout and with either exchange.
// Built only on the public C-ABI of the per-shard handle plus HIP memory / stream calls.
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "rsx_common.h"

using namespace rsx;

namespace {

// ---- the six RCCL entry points used, bound at run time (rccl.h: ncclResult_t = int, ncclChar = 0) ----
struct Rccl {
  void *lib = nullptr;
  int (*CommInitAll)(void **comms, int ndev, const int *devlist) = nullptr;
  int (*CommDestroy)(void *comm) = nullptr;
  int (*AllGather)(const void *send, void *recv, size_t count, int dtype, void *comm, hipStream_t stream) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
};

int load_rccl(Rccl **out) {
  static std::mutex mu;
  static Rccl r;
  static bool tried = false;
  std::lock_guard<std::mutex> lk(mu);
  if (!tried) {
    tried = true;
    for (const char *name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
      r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (r.lib) break;
    }
    if (r.lib) {
      r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(dlsym(r.lib, "ncclCommInitAll"));
      r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.lib, "ncclCommDestroy"));
      r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(r.lib, "ncclAllGather"));
      r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(dlsym(r.lib, "ncclGroupStart"));
      r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(dlsym(r.lib, "ncclGroupEnd"));
      r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.lib, "ncclGetErrorString"));
    }
  }
  if (!r.lib || !r.CommInitAll || !r.CommDestroy || !r.AllGather || !r.GroupStart || !r.GroupEnd)
    return fail(RSX_ERR_INTERNAL, "RCCL exchange requested but librccl.so could not be loaded or lacks a symbol");
  *out = &r;
  return RSX_OK;
}

#define RSX_NCCL(rc, expr)                                                                                                  \
  do {                                                                                                                      \
    const int _r = (expr);                                                                                                  \
    if (_r != 0) return fail(RSX_ERR_INTERNAL, "%s failed: %s", #expr, (rc)->GetErrorString ? (rc)->GetErrorString(_r) : "rccl error"); \
  } while (0)

struct Shard {
  rsx_sc *h = nullptr;
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev = nullptr;
  void *comm = nullptr;             // RCCL communicator of this device inside its query group
  DevBuf q, part, bound, out, all;  // all: the gathered lists of the group (RCCL exchange: on every shard)
};

struct Group {                // one query group = S consecutive shards
  int first = 0;              // index of its first shard (the "leader": merges, returns the result)
  DevBuf all, merged;         // on the leader's device
  hipEvent_t ev_merged = nullptr;
};

}  // namespace

struct rsx_scs {
  std::mutex mu;
  rsx_sc_params p;
  std::vector<Shard> sh;
  std::vector<Group> gr;
  int n_shards = 1;   // S: DB shards per query group
  int exchange = RSX_SCS_EXCHANGE_PEER_COPY;
  Rccl *rccl = nullptr;
  int tree_counter = 0;
  int64_t tree_size = 0;
  bool failed = false;  // an add reached some shards but not all: the shards disagree about the keyframe count
};

namespace {

int use(const Shard &s) {
  RSX_HIP(hipSetDevice(s.device));
  return RSX_OK;
}

int usable(const rsx_scs *h) {
  if (h->failed)
    return fail(RSX_ERR_INTERNAL, "this rsx_scs handle is inconsistent (an earlier add failed on one shard after others had taken it): destroy it");
  return RSX_OK;
}

// an add must reach every shard or none: a failure after the first shard leaves ownership (i % S) and the eligible
// prefix different per shard, so the handle is retired instead of answering from a torn database
template <typename F>
int add_to_all(rsx_scs *h, F &&add_one) {
  RSX_TRY(usable(h));
  for (size_t g = 0; g < h->sh.size(); g++) {
    const int st = add_one(h->sh[g]);
    if (st != RSX_OK) {
      if (g > 0) h->failed = true;
      return st;
    }
  }
  return RSX_OK;
}

// one exchange of a query group: every shard's list (`part` or `out`, nq x k records) -> the merged list.
// to_bound: the merged list ends up in every shard's `bound` (stage 2 reads it); else in gr.merged on the leader.
int exchange(rsx_scs *h, Group &gr, bool from_out, int32_t nq, int32_t k, bool to_bound) {
  const size_t bytes = (size_t)nq * k * sizeof(rsx_sc_hit);
  const int S = h->n_shards;
  Shard &s0 = h->sh[(size_t)gr.first];
  if (h->exchange == RSX_SCS_EXCHANGE_RCCL) {
    for (int g = 0; g < S; g++) {
      Shard &s = h->sh[(size_t)(gr.first + g)];
      RSX_TRY(use(s));
      RSX_TRY(s.all.reserve(bytes * S, s.stream, false));
    }
    // a return between GroupStart and GroupEnd would leave the RCCL group open on this thread and wedge every later
    // collective of the handle: the first error is kept, the group is always closed, then the error is returned
    RSX_NCCL(h->rccl, h->rccl->GroupStart());
    auto enqueue = [&]() -> int {
      for (int g = 0; g < S; g++) {
        Shard &s = h->sh[(size_t)(gr.first + g)];
        RSX_TRY(use(s));
        RSX_NCCL(h->rccl, h->rccl->AllGather(from_out ? s.out.p : s.part.p, s.all.p, bytes, /* ncclChar */ 0, s.comm, s.stream));
      }
      return RSX_OK;
    };
    const int st_enq = enqueue();
    const std::string msg_enq = st_enq != RSX_OK ? std::string(rsx_last_error_string()) : std::string();
    RSX_NCCL(h->rccl, h->rccl->GroupEnd());
    if (st_enq != RSX_OK) return fail(st_enq, "%s", msg_enq.c_str());
    for (int g = 0; g < S; g++) {
      if (!to_bound && g != 0) continue;  // the final merge is only needed where the result is read
      Shard &s = h->sh[(size_t)(gr.first + g)];
      RSX_TRY(use(s));
      rsx_sc_hit *dst = s.bound.as<rsx_sc_hit>();
      if (!to_bound) {
        RSX_TRY(gr.merged.reserve(bytes, s.stream, false));
        dst = gr.merged.as<rsx_sc_hit>();
      }
      RSX_TRY(rsx_sc_merge_topk_device(s.h, s.all.as<rsx_sc_hit>(), S, nq, k, dst, s.stream));
    }
    return RSX_OK;
  }
  RSX_TRY(use(s0));
  RSX_TRY(gr.all.reserve(bytes * S, s0.stream, false));
  RSX_TRY(gr.merged.reserve(bytes, s0.stream, false));
  for (int g = 0; g < S; g++) {
    Shard &s = h->sh[(size_t)(gr.first + g)];
    RSX_HIP(hipStreamWaitEvent(s0.stream, s.ev, 0));
    RSX_HIP(hipMemcpyPeerAsync(static_cast<char *>(gr.all.p) + bytes * g, s0.device, from_out ? s.out.p : s.part.p, s.device, bytes, s0.stream));
  }
  RSX_TRY(rsx_sc_merge_topk_device(s0.h, gr.all.as<rsx_sc_hit>(), S, nq, k, gr.merged.as<rsx_sc_hit>(), s0.stream));
  RSX_HIP(hipEventRecord(gr.ev_merged, s0.stream));
  if (to_bound) {
    for (int g = 0; g < S; g++) {
      Shard &s = h->sh[(size_t)(gr.first + g)];
      RSX_TRY(use(s));
      RSX_HIP(hipStreamWaitEvent(s.stream, gr.ev_merged, 0));
      RSX_HIP(hipMemcpyPeerAsync(s.bound.p, s.device, gr.merged.p, s0.device, bytes, s.stream));
    }
  }
  return RSX_OK;
}

// nq queries (host pointer q) on one query group; *d_result = device pointer of the nq x k records on the group's leader
int query_group(rsx_scs *h, Group &gr, const float *q, int32_t nq, int32_t k, int64_t n_eligible, const rsx_sc_hit **d_result) {
  const size_t qbytes = (size_t)nq * RSX_SC_DESC_SIZE * sizeof(float);
  const size_t bytes = (size_t)nq * k * sizeof(rsx_sc_hit);
  const int S = h->n_shards;
  const bool single = S == 1 && h->exchange != RSX_SCS_EXCHANGE_RCCL;  // one shard, nothing to exchange: the one-stage query
  for (int g = 0; g < S; g++) {
    Shard &s = h->sh[(size_t)(gr.first + g)];
    RSX_TRY(use(s));
    RSX_TRY(s.q.reserve(qbytes, s.stream, false));
    RSX_TRY(s.part.reserve(bytes, s.stream, false));
    RSX_TRY(s.bound.reserve(bytes, s.stream, false));
    RSX_TRY(s.out.reserve(bytes, s.stream, false));
    RSX_HIP(hipMemcpyAsync(s.q.p, q, qbytes, hipMemcpyHostToDevice, s.stream));
    if (single) {
      RSX_TRY(rsx_sc_query_device(s.h, s.q.as<float>(), nq, k, n_eligible, s.out.as<rsx_sc_hit>(), s.stream));
      *d_result = s.out.as<rsx_sc_hit>();
      return RSX_OK;
    }
    // stage 1 on every device of the group (asynchronous: the devices work concurrently)
    RSX_TRY(rsx_sc_query_stage1_device(s.h, s.q.as<float>(), nq, k, n_eligible, s.part.as<rsx_sc_hit>(), s.stream));
    RSX_HIP(hipEventRecord(s.ev, s.stream));
  }
  RSX_TRY(exchange(h, gr, false, nq, k, true));  // the k-th distance of the merged lists = the group's bound tau
  for (int g = 0; g < S; g++) {                  // stage 2: what tau still admits
    Shard &s = h->sh[(size_t)(gr.first + g)];
    RSX_TRY(use(s));
    RSX_TRY(rsx_sc_query_stage2_device(s.h, nq, k, s.bound.as<rsx_sc_hit>(), s.out.as<rsx_sc_hit>(), s.stream));
    RSX_HIP(hipEventRecord(s.ev, s.stream));
  }
  // (peer copies: the leader must not overwrite `merged` before every shard has copied it: its stream waits for the
  // stage-2 events inside the exchange)
  RSX_TRY(exchange(h, gr, true, nq, k, false));
  *d_result = gr.merged.as<rsx_sc_hit>();
  return RSX_OK;
}

int query_locked(rsx_scs *h, const float *q, int32_t nq, int32_t k, int64_t n_eligible, rsx_sc_hit *out) {
  const int Q = (int)h->gr.size();
  const int32_t chunk = (nq + Q - 1) / Q;
  struct Pending {
    Shard *lead;
    const rsx_sc_hit *d_res;
    int32_t lo, n;
  };
  std::vector<Pending> pend;
  // nothing of this call may be in flight when the caller gets its buffers back, whatever the status
  auto drain = [&]() {
    for (const Pending &p : pend)
      if (use(*p.lead) == RSX_OK) (void)hipStreamSynchronize(p.lead->stream);
  };
  // after a failure: EVERY shard stream of EVERY group (the group that failed part-way is not in `pend`, and the non-lead
  // shards of the others may still hold uploads from the caller's query buffer or exchanges in flight)
  auto drain_all = [&]() {
    for (Shard &sh : h->sh)
      if (sh.stream && use(sh) == RSX_OK) (void)hipStreamSynchronize(sh.stream);
  };
  // first every group is launched (device work only: the groups run concurrently) ...
  for (int g = 0; g < Q; g++) {
    const int32_t lo = std::min(nq, g * chunk), n = std::min(nq, lo + chunk) - lo;
    if (n <= 0) continue;
    const rsx_sc_hit *d_res = nullptr;
    Group &gr = h->gr[(size_t)g];
    const int st = query_group(h, gr, q + (size_t)lo * RSX_SC_DESC_SIZE, n, k, n_eligible, &d_res);
    if (st != RSX_OK) {
      drain_all();
      return st;
    }
    pend.push_back(Pending{&h->sh[(size_t)gr.first], d_res, lo, n});
  }
  // ... then the read-backs: a copy into the caller's (pageable) memory may hold this thread until it has completed, which
  // between two launches would have run the groups one after the other
  int st = RSX_OK;
  for (const Pending &p : pend) {
    st = use(*p.lead);
    if (st == RSX_OK && hipMemcpyAsync(out + (size_t)p.lo * k, p.d_res, (size_t)p.n * k * sizeof(rsx_sc_hit), hipMemcpyDeviceToHost,
                                       p.lead->stream) != hipSuccess)
      st = fail(RSX_ERR_HIP, "read-back of a query group failed");
    if (st != RSX_OK) break;
  }
  if (st != RSX_OK) drain_all();
  else drain();
  return st;
}

}  // namespace

extern "C" {

int rsx_scs_create_layout(const rsx_sc_params *p, const int32_t *devices, int32_t n_devices, int32_t query_groups, int32_t exchange_kind,
                          rsx_scs **out) try {
  if (!out || !devices || n_devices < 1) return fail(RSX_ERR_BAD_ARG, "bad arg");
  *out = nullptr;
  if (query_groups < 1 || n_devices % query_groups) return fail(RSX_ERR_BAD_ARG, "query_groups %d does not divide %d devices", query_groups, n_devices);
  if (exchange_kind != RSX_SCS_EXCHANGE_PEER_COPY && exchange_kind != RSX_SCS_EXCHANGE_RCCL) return fail(RSX_ERR_BAD_ARG, "bad exchange kind %d", exchange_kind);
  rsx_scs *h = new (std::nothrow) rsx_scs();
  if (!h) return fail(RSX_ERR_OOM, "host alloc");
  rsx_sc_default_params(&h->p);
  if (p) h->p = *p;
  const int S = n_devices / query_groups;
  h->n_shards = S;
  h->exchange = exchange_kind;
  h->sh.resize((size_t)n_devices);
  h->gr.resize((size_t)query_groups);
  int st = RSX_OK;
  if (exchange_kind == RSX_SCS_EXCHANGE_RCCL) st = load_rccl(&h->rccl);
  for (int g = 0; g < n_devices && st == RSX_OK; g++) {
    Shard &s = h->sh[(size_t)g];
    s.device = devices[g];
    rsx_sc_params sp = h->p;
    sp.device = devices[g];
    sp.shard_rank = g % S;
    sp.shard_world = S;
    if (sp.capacity_hint > 0) sp.capacity_hint = sp.capacity_hint / S + 32;
    st = rsx_sc_create(&sp, &s.h);
    if (st != RSX_OK) break;
    hipError_t e = hipSetDevice(s.device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&s.ev, hipEventDisableTiming);
    if (e != hipSuccess) st = fail(RSX_ERR_HIP, "stream/event on device %d: %s", s.device, hipGetErrorString(e));
    for (int o = (g / S) * S; o < g && st == RSX_OK; o++) {  // direct xGMI copies inside a group where the topology allows
      const int od = h->sh[(size_t)o].device;
      if (od == s.device) continue;
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, s.device, od) == hipSuccess && can) {
        (void)hipSetDevice(s.device);
        (void)hipDeviceEnablePeerAccess(od, 0);
        (void)hipSetDevice(od);
        (void)hipDeviceEnablePeerAccess(s.device, 0);
        (void)hipGetLastError();  // "already enabled" is fine
      }
    }
  }
  for (int g = 0; g < query_groups && st == RSX_OK; g++) {
    Group &gr = h->gr[(size_t)g];
    gr.first = g * S;
    hipError_t e = hipSetDevice(h->sh[(size_t)gr.first].device);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&gr.ev_merged, hipEventDisableTiming);
    if (e != hipSuccess) st = fail(RSX_ERR_HIP, "event: %s", hipGetErrorString(e));
    if (st == RSX_OK && exchange_kind == RSX_SCS_EXCHANGE_RCCL) {
      std::vector<int> devs;
      std::vector<void *> comms((size_t)S, nullptr);
      for (int s = 0; s < S; s++) devs.push_back(h->sh[(size_t)(gr.first + s)].device);
      for (int a = 0; a < S && st == RSX_OK; a++)
        for (int b = a + 1; b < S; b++)
          if (devs[(size_t)a] == devs[(size_t)b]) {
            st = fail(RSX_ERR_BAD_ARG, "the RCCL exchange needs distinct devices inside a query group (device %d listed twice)", devs[(size_t)a]);
            break;
          }
      if (st == RSX_OK) {
        const int r = h->rccl->CommInitAll(comms.data(), S, devs.data());
        if (r != 0) {
          st = fail(RSX_ERR_INTERNAL, "ncclCommInitAll failed: %s", h->rccl->GetErrorString ? h->rccl->GetErrorString(r) : "rccl error");
        } else {
          for (int s = 0; s < S; s++) h->sh[(size_t)(gr.first + s)].comm = comms[(size_t)s];
        }
      }
    }
  }
  if (st != RSX_OK) {
    const std::string keep = last_error();
    rsx_scs_destroy(h);
    last_error() = keep;
    return st;
  }
  *out = h;
  return RSX_OK;
} RSX_CATCH_ALL

int rsx_scs_create(const rsx_sc_params *p, const int32_t *devices, int32_t n_devices, rsx_scs **out) try {
  return rsx_scs_create_layout(p, devices, n_devices, 1, RSX_SCS_EXCHANGE_PEER_COPY, out);
} RSX_CATCH_ALL

int rsx_scs_destroy(rsx_scs *h) try {
  if (!h) return RSX_OK;
  for (Shard &s : h->sh) {
    (void)hipSetDevice(s.device);
    if (s.stream) (void)hipStreamSynchronize(s.stream);
    if (s.comm && h->rccl) (void)h->rccl->CommDestroy(s.comm);
    for (DevBuf *b : {&s.q, &s.part, &s.bound, &s.out, &s.all}) b->release();
    if (s.ev) (void)hipEventDestroy(s.ev);
    if (s.stream) (void)hipStreamDestroy(s.stream);
    if (s.h) rsx_sc_destroy(s.h);
  }
  for (Group &g : h->gr) {
    if ((size_t)g.first < h->sh.size()) (void)hipSetDevice(h->sh[(size_t)g.first].device);
    g.all.release();
    g.merged.release();
    if (g.ev_merged) (void)hipEventDestroy(g.ev_merged);
  }
  delete h;
  return RSX_OK;
} RSX_CATCH_ALL

int rsx_scs_num_shards(rsx_scs *h) { return h ? h->n_shards : 0; }
int rsx_scs_num_query_groups(rsx_scs *h) { return h ? (int)h->gr.size() : 0; }

int rsx_scs_set_dist_thres(rsx_scs *h, double thres) try {
  if (!h) return fail(RSX_ERR_BAD_ARG, "null handle");
  std::lock_guard<std::mutex> lk(h->mu);
  h->p.dist_thres = thres;
  for (Shard &s : h->sh) RSX_TRY(rsx_sc_set_dist_thres(s.h, thres));
  return RSX_OK;
} RSX_CATCH_ALL

int rsx_scs_size(rsx_scs *h, int64_t *n_global) try {
  if (!h || !n_global) return fail(RSX_ERR_BAD_ARG, "null arg");
  std::lock_guard<std::mutex> lk(h->mu);
  return rsx_sc_size(h->sh[0].h, n_global);
} RSX_CATCH_ALL

// every shard sees every keyframe (the owner builds and stores it, the others advance their count)
int rsx_scs_add_points(rsx_scs *h, const void *pts, size_t n, size_t stride_bytes, int32_t *out_index) try {
  if (!h) return fail(RSX_ERR_BAD_ARG, "null handle");
  std::lock_guard<std::mutex> lk(h->mu);
  int32_t idx = 0;
  RSX_TRY(add_to_all(h, [&](Shard &s) { return rsx_sc_add_points(s.h, pts, n, stride_bytes, &idx); }));
  if (out_index) *out_index = idx;
  return RSX_OK;
} RSX_CATCH_ALL

int rsx_scs_add_descriptors_f32(rsx_scs *h, const float *descs, int64_t n) try {
  if (!h) return fail(RSX_ERR_BAD_ARG, "null handle");
  std::lock_guard<std::mutex> lk(h->mu);
  return add_to_all(h, [&](Shard &s) { return rsx_sc_add_descriptors_f32(s.h, descs, n); });
} RSX_CATCH_ALL

int rsx_scs_get_descriptor(rsx_scs *h, int64_t index, double *out_colmajor) try {
  if (!h || !out_colmajor) return fail(RSX_ERR_BAD_ARG, "null arg");
  std::lock_guard<std::mutex> lk(h->mu);
  if (index < 0) return fail(RSX_ERR_RANGE, "index %lld out of range", (long long)index);
  return rsx_sc_get_descriptor(h->sh[(size_t)(index % (int64_t)h->n_shards)].h, index, out_colmajor);  // query group 0's copy
} RSX_CATCH_ALL

int rsx_scs_query(rsx_scs *h, const float *q_descs, int32_t nq, int32_t k, int64_t n_eligible, rsx_sc_hit *out) try {
  if (!h || !q_descs || !out || nq < 1) return fail(RSX_ERR_BAD_ARG, "bad arg");
  if (k < 1 || k > RSX_SC_MAX_TOPK) return fail(RSX_ERR_BAD_ARG, "k must be in [1,%d]", RSX_SC_MAX_TOPK);
  std::lock_guard<std::mutex> lk(h->mu);
  RSX_TRY(usable(h));
  return query_locked(h, q_descs, nq, k, n_eligible, out);
} RSX_CATCH_ALL

// detectLoopClosureID (SC.cpp:331-422) over the sharded database, exhaustive mode: the frozen
// searchable prefix and the 30-keyframe exclusion are the reference's; every entry of the prefix is scored.
int rsx_scs_detect_loop_closure(rsx_scs *h, rsx_sc_detection *out) try {
  if (!h || !out) return fail(RSX_ERR_BAD_ARG, "null arg");
  std::lock_guard<std::mutex> lk(h->mu);
  RSX_TRY(usable(h));
  int64_t N = 0;
  RSX_TRY(rsx_sc_size(h->sh[0].h, &N));
  out->loop_id = -1;
  out->yaw_diff_rad = 0.0f;
  out->min_dist = 10000000;
  out->nn_idx = 0;
  out->query_idx = (int32_t)(N - 1);
  out->searched = 0;
  out->reserved = 0;
  out->dist_thres = h->p.dist_thres;
  if (N == 0 || N < h->p.num_exclude_recent + 1) return RSX_OK;                                       // SC.cpp:341-345
  if (h->tree_counter % h->p.tree_making_period == 0) h->tree_size = N - h->p.num_exclude_recent;   // SC.cpp:348-359
  h->tree_counter = h->tree_counter + 1;                                                             // SC.cpp:360
  double d[RSX_SC_DESC_SIZE];
  float f[RSX_SC_DESC_SIZE];
  RSX_TRY(rsx_sc_get_descriptor(h->sh[(size_t)((N - 1) % (int64_t)h->n_shards)].h, N - 1, d));       // SC.cpp:336
  for (int i = 0; i < RSX_SC_DESC_SIZE; i++) f[i] = (float)d[i];  // stored as fp32: exact
  rsx_sc_hit hit;
  RSX_TRY(query_locked(h, f, 1, 1, h->tree_size, &hit));
  out->searched = 1;
  if (hit.dist < 10000000) {  // SC.cpp:388 strict `<` against the 1e7 init
    out->min_dist = hit.dist;
    out->nn_idx = hit.index;
  }
  RSX_TRY(rsx_sc_hit_to_loop(h->sh[0].h, &hit, &out->loop_id, &out->yaw_diff_rad));                  // SC.cpp:401-417
  if (!(hit.dist < 10000000)) out->yaw_diff_rad = 0.0f;
  return RSX_OK;
} RSX_CATCH_ALL

}  // extern "C"
