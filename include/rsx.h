/*
 * rsx.h -- C-ABI of librsx.so: the MI355X-native (gfx950, HIP) ScanContext + ORORA hot path of
 * navtech-radar-slam.  Plain pointers and sizes only; no C++/torch types cross this boundary.
 *
 * Reference interfaces each group replaces (paths under the reference checkout):
 *   SC  = pgo/SC-A-LOAM/include/scancontext/Scancontext.{h,cpp}
 *   PGO = pgo/SC-A-LOAM/src/laserPosegraphOptimization.cpp
 *
 * Conventions
 *   - every function returns an rsx_status (0 = ok, <0 = error); a loop id of -1 is DATA ("no
 *     loop", SC.cpp:333), not an error.  No C++ exception crosses the ABI.
 *     rsx_last_error_string() gives a thread-local diagnostic.
 *   - the caller owns every buffer it passes; the library copies inputs before returning and
 *     writes outputs into caller memory.  Pointers named d_* are DEVICE (HBM) pointers of the
 *     handle's GPU and are used asynchronously on the given hipStream_t (passed as void*).
 *     A NULL stream selects the handle's own private (non-blocking) stream, NOT the legacy default
 *     stream: callers that mix several handles or other GPU work must pass one explicit stream.
 *     Calls on ONE handle that pass different streams are ordered by the library (the handle's workspaces are shared: a call
 *     arriving on another stream than the previous call makes its stream wait for everything enqueued on the old one).
 *   - a handle is internally synchronised: every entry point holds the handle's mutex for its whole
 *     duration, so any number of threads may call concurrently (the reference itself races between its
 *     writer, PGO.cpp:492 process_pg, and its reader, PGO.cpp:561 process_lcd).  The one multi-call
 *     protocol, query stage 1 -> stage 2, is invalidated (stage 2 fails with RSX_ERR_BAD_ARG) by any
 *     other query-type call on the same handle in between, instead of reading clobbered workspaces.
 *   - there is NO CPU fallback: without a usable HIP device rsx_*_create fails with
 *     RSX_ERR_NO_DEVICE.
 *
 * Descriptor layouts
 *   - "colmajor double": 20 x 60 Eigen::MatrixXd memory order (SC.cpp:159), element
 *     (ring r, sector s) at [s*20 + r]; 9600 B.
 *   - "f32 sector-major": the same order in float, 4800 B; every descriptor the reference can
 *     build is exactly representable in fp32 (SC.cpp:168: pt.z is a float).
 */
#ifndef RSX_H
#define RSX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RSX_SC_NUM_RING 20    /* SC.h:85  PC_NUM_RING   */
#define RSX_SC_NUM_SECTOR 60  /* SC.h:86  PC_NUM_SECTOR */
#define RSX_SC_DESC_SIZE 1200
#define RSX_SC_MAX_TOPK 32

typedef enum {
  RSX_OK = 0,
  RSX_ERR_BAD_ARG = -1,
  RSX_ERR_NO_DEVICE = -2,
  RSX_ERR_HIP = -3,
  RSX_ERR_OOM = -4,
  RSX_ERR_NOT_FP32_EXACT = -5, /* a colmajor-double descriptor does not round-trip through fp32 */
  RSX_ERR_RANGE = -6,
  RSX_ERR_INTERNAL = -7
} rsx_status;

const char *rsx_last_error_string(void);
/* "rsx <version> gfx950 ..." */
const char *rsx_version(void);
/* number of visible HIP devices (0 on a box without a GPU; never an error) */
int rsx_device_count(void);

/* ============================== ScanContext (SC.h:62-122) ============================== */

typedef struct rsx_sc rsx_sc; /* replaces a `SCManager` instance (PGO.cpp:99) */

/* one loop candidate: distanceBtnScanContext() result (SC.cpp:116-148) + DB index */
typedef struct {
  double dist;   /* min over the searched column shifts of the mean column cosine distance */
  int32_t index; /* GLOBAL keyframe index */
  int32_t shift; /* argmin column shift; yaw = shift * 6 deg */
} rsx_sc_hit;

typedef struct {
  double lidar_height;        /* SC.h:83  LIDAR_HEIGHT = 2.0 */
  double max_radius;          /* SC.h:87  PC_MAX_RADIUS = 80 */
  int32_t num_exclude_recent; /* SC.h:92  = 30 */
  int32_t num_candidates;     /* SC.h:93  NUM_CANDIDATES_FROM_TREE = 3 (<= 32) */
  double search_ratio;        /* SC.h:96  = 0.1 (kernels are specialised for the resulting +-3 window) */
  double dist_thres;          /* SC.h:99  SC_DIST_THRES = 0.2 (sc_pgo.launch:4 sets 0.45) */
  int32_t tree_making_period; /* SC.h:103 = 30 */
  int32_t device;             /* HIP device ordinal */
  /* DB sharding over GPUs: this handle stores the entries whose global index i has
   * i % shard_world == shard_rank (block-cyclic), local slot i / shard_world. */
  int32_t shard_rank;         /* default 0 */
  int32_t shard_world;        /* default 1 */
  int64_t capacity_hint;      /* initial DB capacity in entries (grows by doubling) */
  /* exhaustive queries: 0 = auto (up to 16 queries per call -- the live detector asks one, PGO.cpp:561,577 -- take the
   * one-launch single-query path, sc_q1.hip; batched queries go through the MFMA lower-bound filter and only
   * the entries that can still reach the top-k are scored by the exact fp64 kernel), 1 = always
   * score every entry exactly, 2 = always the batched filter chain, 3 = the single-query path wherever it applies (<= 16
   * queries per call; auto otherwise).  Results are identical in every mode. */
  int32_t filter_mode;
  /* which form of the filter: 0 = default (3: spectral, two waves per SIMD), 1 = direct (60-shift correlation as one K = 1200
   * MFMA GEMM per shift), 2 = spectral (Z15 DFT + direct Z4 correlation, ~6x fewer MFMAs), 3 = the spectral form
   * with two waves per SIMD (the entry tile split by frequency over a wave pair; same bounds bit for bit).  All give
   * valid lower bounds of the same quantity; results are identical. */
  int32_t filter_kind;
  /* summation order of the sums the reference takes through Eigen (mean, norm, dot): a property of how the REFERENCE was
   * built, which the bit-exact contract follows.  RSX_SC_SUM_EIGEN_SSE2 (0, default) = the reference as its CMakeLists.txt
   * builds it (x86-64, -O3, 2-double packets); RSX_SC_SUM_SEQ = Eigen without vectorisation; RSX_SC_SUM_EIGEN_AVX_FMA = a
   * workspace compiled with -march=native (4-double packets, fused multiply-adds).  csrc/sc_redux_dev.h. */
  int32_t sum_order;
} rsx_sc_params;
#define RSX_SC_SUM_EIGEN_SSE2 0
#define RSX_SC_SUM_SEQ 1
#define RSX_SC_SUM_EIGEN_AVX_FMA 2
#define RSX_SC_SUM_EIGEN34_AVX_FMA 3 /* as 2 against Eigen 3.4: its AVX horizontal add is (l0 + l2) + (l1 + l3), 3.3's (l0 + l1) + (l2 + l3) */

typedef enum {
  RSX_SC_MODE_CANDIDATE = 0, /* reference semantics: ring-key 3-NN then 3 pair distances (SC.cpp:331-422) */
  RSX_SC_MODE_EXHAUSTIVE = 1 /* same pair function against every eligible entry */
} rsx_sc_mode;


int rsx_sc_default_params(rsx_sc_params *p);
int rsx_sc_create(const rsx_sc_params *p, rsx_sc **out);   /* SCManager() */
int rsx_sc_destroy(rsx_sc *h);
int rsx_sc_set_dist_thres(rsx_sc *h, double thres);        /* setSCdistThres, SC.cpp:262-265 */
/* number of keyframes known to this handle (GLOBAL count; a shard stores ~1/world of them) */
int rsx_sc_size(rsx_sc *h, int64_t *n_global);
int rsx_sc_local_size(rsx_sc *h, int64_t *n_local);

/* makeAndSaveScancontextAndKeys (SC.cpp:249-260).  pts: n points, stride_bytes apart
 * (32 for pcl::PointXYZI), float x,y,z at byte offsets 0,4,8.  The descriptor, ring key, sector
 * key and column norms are built on the GPU.  In a sharded handle every rank must call this for
 * every keyframe (same order); ranks that do not own the slot only advance the global count.
 * out_index (optional) = global index of the new keyframe.
 * The call copies the cloud (pts is free again on return) and enqueues ONE kernel; it does not wait for the GPU.  Every
 * later call on the handle sees the entry (calls that bring their own stream are ordered behind it); a device fault of
 * the insert is reported by the next call that synchronises. */
int rsx_sc_add_points(rsx_sc *h, const void *pts, size_t n, size_t stride_bytes, int32_t *out_index);
/* saveScancontextAndKeys (SC.cpp:236-246), colmajor double; RSX_ERR_NOT_FP32_EXACT if lossy */
int rsx_sc_add_descriptor(rsx_sc *h, const double *desc_colmajor, int32_t *out_index);
/* the same for descriptors that did not come out of makeScancontext (e.g. SCDs re-read from decimal text
 * files in multi-session use): every element is rounded to fp32 -- the DB stores fp32 -- and the largest
 * absolute rounding error is reported (optional).  NaN elements are still rejected. */
int rsx_sc_add_descriptor_rounded(rsx_sc *h, const double *desc_colmajor, int32_t *out_index, double *max_abs_rounding);
/* bulk import of n f32 sector-major descriptors (host memory); same sharding rule */
int rsx_sc_add_descriptors_f32(rsx_sc *h, const float *descs, int64_t n);
/* the same from DEVICE memory (no host round trip), n consecutive global keyframes */
int rsx_sc_add_descriptors_f32_device(rsx_sc *h, const float *d_descs, int64_t n, void *stream);

/* bulk export: the f32 sector-major descriptors of LOCAL slots [first_slot, first_slot + count) (host out) */
int rsx_sc_export_descriptors_f32(rsx_sc *h, int64_t first_slot, int64_t count, float *out);
/* on-disk database (the re-ingest side is saveScancontextAndKeys, SC.cpp:236-246): a 64-byte
 * little-endian header {"RSXSCDB1", version, rings, sectors, dtype, n_global, n_local, shard_rank, shard_world}
 * followed by the handle's n_local fp32 sector-major descriptors.  Keys, norms and filter images are rebuilt on the
 * GPU by rsx_sc_load.  An unsharded file appends to any handle (each shard keeps its residue class); a shard file
 * restores that shard into an empty handle with the same shard_rank / shard_world. */
int rsx_sc_save(rsx_sc *h, const char *path);
int rsx_sc_load(rsx_sc *h, const char *path, int64_t *n_loaded);

/* polarcontexts_[i] / getConstRefRecentSCD (SC.h:79,111): global index must be owned by this shard */
int rsx_sc_get_descriptor(rsx_sc *h, int64_t index, double *out_colmajor);
int rsx_sc_get_ringkey(rsx_sc *h, int64_t index, float *out20);     /* polarcontext_invkeys_mat_ */
int rsx_sc_get_sectorkey(rsx_sc *h, int64_t index, double *out60);  /* polarcontext_vkeys_ */

/* detectLoopClosureID (SC.cpp:331-422).  loop_id/-1 and yaw are the reference's return pair;
 * min_dist / nn_idx (optional) are the values of its log line (SC.cpp:406,412).
 * mode CANDIDATE reproduces the reference (frozen searchable prefix rebuilt every
 * tree_making_period calls, NUM_EXCLUDE_RECENT, kNN order, strict-< first-wins);
 * mode EXHAUSTIVE scores the whole frozen prefix.  Unsharded handles only. */
int rsx_sc_detect_loop_closure(rsx_sc *h, int mode, int32_t *loop_id, float *yaw_diff_rad,
                               double *min_dist, int32_t *nn_idx);
/* the same with everything the reference's log line shows (SC.cpp:406,412: "[Loop found] Nearest distance: <min_dist>
 * btn <query_idx> and <nn_idx>."), taken under one lock so that query_idx is the keyframe that was actually the query */
typedef struct {
  int32_t loop_id;      /* nn_idx if min_dist < dist_thres else -1 */
  float yaw_diff_rad;
  double min_dist;
  int32_t nn_idx;
  int32_t query_idx;    /* polarcontexts_.size() - 1 at the time of the call */
  int32_t searched;     /* 0: the early return of SC.cpp:341-345 (fewer than NUM_EXCLUDE_RECENT + 1 keyframes; the
                           reference prints nothing then) */
  int32_t reserved;
  double dist_thres;    /* the threshold that was applied */
} rsx_sc_detection;
int rsx_sc_detect_loop_closure_ex(rsx_sc *h, int mode, rsx_sc_detection *out);
/* detectLoopClosureIDBetweenSession (SC.cpp:267-328): query passed in, tree over the whole DB
 * as of the first call. */
int rsx_sc_detect_between_session(rsx_sc *h, const float *curr_key20, const double *curr_desc_colmajor,
                                  int32_t *loop_id, float *yaw_diff_rad, double *min_dist,
                                  int32_t *nn_idx);
/* current frozen searchable prefix length ("tree" size, SC.cpp:352-353) */
int rsx_sc_tree_size(rsx_sc *h, int64_t *n);

/* The reference's public helper methods (SC.h:60-66), stateless: the handle only lends its device, stream and
 * staging memory.  They run on the GPU in fp64 on the doubles as given (no fp32 storage involved, any MatrixXd
 * content is accepted) and equal the reference bit for bit.  One small launch per call: API completeness, not
 * the batched path.
 *   makeScancontext (SC.cpp:151-195)                          -> 20 x 60 colmajor double */
int rsx_sc_make_scancontext(rsx_sc *h, const void *pts, size_t n, size_t stride_bytes, double *out_desc_colmajor);
/*   makeRingkeyFromScancontext / makeSectorkeyFromScancontext (SC.cpp:198-227), either output optional */
int rsx_sc_make_keys(rsx_sc *h, const double *desc_colmajor, double *out_ringkey20, double *out_sectorkey60);
/*   distDirectSC (SC.cpp:69-90): no alignment, no shift; NaN when no column is effective */
int rsx_sc_dist_direct(rsx_sc *h, const double *sc1_colmajor, const double *sc2_colmajor, double *out_dist);
/*   fastAlignUsingVkey (SC.cpp:93-113) on two 1 x 60 sector keys */
int rsx_sc_fast_align(rsx_sc *h, const double *vkey1_60, const double *vkey2_60, int32_t *out_shift);
/*   distanceBtnScanContext (SC.cpp:116-148) */
int rsx_sc_distance(rsx_sc *h, const double *sc1_colmajor, const double *sc2_colmajor, double *out_dist, int32_t *out_shift);

/* Exhaustive batched query (the north-star path): nq f32 sector-major query descriptors against
 * every LOCAL entry whose global index < n_eligible (n_eligible < 0: all); out = nq x k records
 * sorted by (dist, index), padded with {1e7, 0, 0} (SC.cpp:362-364 initial values).
 * Host-buffer form (synchronous): */
int rsx_sc_query(rsx_sc *h, const float *q_descs, int32_t nq, int32_t k, int64_t n_eligible,
                 rsx_sc_hit *out);
/* Device-buffer form: asynchronous on `stream` (hipStream_t); d_out stays on the GPU so a sharded
 * caller can all-gather it (RCCL) without a host hop. */
int rsx_sc_query_device(rsx_sc *h, const float *d_q_descs, int32_t nq, int32_t k, int64_t n_eligible,
                        rsx_sc_hit *d_out, void *stream);
/* Filter shards over a REPLICATED database (the layout that scales when the batch is large and the DB small
 * next to HBM: 0.83 GB per 100 000 keyframes).  Every rank holds every keyframe (shard_world 1); only the lower-bound
 * filter -- the per-pair cost -- is cut over the ranks by slot range, the per-query stages (short list, window previews,
 * exact re-scoring) run on each rank for ITS slice of the batch against the whole DB:
 *   every rank r   rsx_sc_filter_range_device: all nq queries x slots [first_r, first_r + n_r)  -> d_lb[nq][ld_r]
 *   caller         all-to-all of the row slices (RCCL): rank t receives rows [q_t, q_t + nq_t) of every rank's matrix,
 *                  one column block per sender
 *   every rank t   rsx_sc_query_bounds_device: its nq_t queries with the received column blocks -> d_out[nq_t][k]
 *   caller         all-gather of d_out
 * The records are those of rsx_sc_query_device on one GPU (the bounds of a pair do not depend on who computed them).
 * first_slot must be a multiple of 32 (the filter images are stored in tiles of 32 entries); d_lb[q * ld + j] is the bound
 * of query q against slot first_slot + j, ld >= n_slots rounded up to 32 (columns past n_slots are unspecified).
 * Bounds are IEEE binary16 (rsx_f16), rounded toward zero -- the element type of the library's own bound matrix.
 * d_lb and d_lb_blocks must be 16-byte aligned (RSX_ERR_BAD_ARG otherwise: they are written / read with 16-byte accesses).
 * n_slots == 0 is a no-op whatever first_slot says (a rank whose range lies beyond the entries of a small database). */
typedef uint16_t rsx_f16;
int rsx_sc_filter_range_device(rsx_sc *h, const float *d_q_descs, int32_t nq, int64_t first_slot, int64_t n_slots,
                               rsx_f16 *d_lb, int64_t ld, void *stream);
/* d_lb_blocks: n_blocks column blocks, block b = [nq][block_ld] bounds starting b * block_stride elements (a multiple of 8)
 * into the buffer,
 * its column j = the bound against slot b * block_ld + j (block_ld a multiple of 32; the blocks together must cover every
 * entry below n_eligible).  Scores nq queries exactly like rsx_sc_query_device, with these bounds in place of its own
 * filter launch. */
int rsx_sc_query_bounds_device(rsx_sc *h, const float *d_q_descs, int32_t nq, int32_t k, int64_t n_eligible,
                               const rsx_f16 *d_lb_blocks, int32_t n_blocks, int64_t block_ld, int64_t block_stride,
                               rsx_sc_hit *d_out, void *stream);
/* The same query in two stages, for a DB sharded over several GPUs.  With one stage
 * every shard would have to re-score the entries that look promising against ITS OWN k-th best
 * distance; with two, the shards first agree on a global bound:
 *   stage 1  filter + this shard's share of the lowest-bound entries   -> d_partial[nq][k]
 *   caller   all-gather d_partial over the ranks (RCCL), rsx_sc_merge_topk_device -> d_global[nq][k]
 *   stage 2  only entries whose bound can still beat the k-th distance of d_global are scored;
 *            d_out[nq][k] = this shard's top-k including its stage-1 hits
 *   caller   all-gather d_out, rsx_sc_merge_topk_device -> the global top-k
 * d_q_descs must stay valid until stage 2 has run; stage 2 must follow stage 1 on the same handle
 * with the same nq and k.  The merged result is identical to rsx_sc_query_device + merge. */
int rsx_sc_query_stage1_device(rsx_sc *h, const float *d_q_descs, int32_t nq, int32_t k, int64_t n_eligible,
                               rsx_sc_hit *d_partial, void *stream);
/* the same with a per-query eligibility limit (device array, nq entries): query i only sees entries
 * with global index < min(n_eligible, d_q_elig[i]) -- e.g. "every keyframe against the keyframes at
 * least 30 older than itself" over a sharded DB (BASELINE configs 4 and 5).  elig_monotone != 0
 * promises that d_q_elig does not decrease with i, which lets the filter skip the tile-blocks a
 * query cannot see.  d_q_elig must stay valid until stage 2 has run. */
int rsx_sc_query_stage1_elig_device(rsx_sc *h, const float *d_q_descs, int32_t nq, int32_t k, int64_t n_eligible,
                                    const int64_t *d_q_elig, int32_t elig_monotone, rsx_sc_hit *d_partial,
                                    void *stream);
int rsx_sc_query_stage2_device(rsx_sc *h, int32_t nq, int32_t k, const rsx_sc_hit *d_global, rsx_sc_hit *d_out,
                               void *stream);
/* queries = DB entries [q_first, q_first+nq) of this handle (all-pairs runs, BASELINE config 5);
 * each query i uses n_eligible = min(n_eligible, q_first+i - exclude_recent) when exclude_recent >= 0 */
int rsx_sc_query_self_device(rsx_sc *h, int64_t q_first, int32_t nq, int32_t k, int64_t n_eligible,
                             int32_t exclude_recent, rsx_sc_hit *d_out, void *stream);
/* parity helper: dist/shift of ONE query against local entries [first, first+count) (host out) */
int rsx_sc_pair_distances(rsx_sc *h, const float *q_desc, int64_t first, int64_t count,
                          double *out_dist, int32_t *out_shift);
/* merge nparts per-shard top-k lists (layout [part][nq][k]) into out[nq][k]; pure host logic */
int rsx_sc_merge_topk(const rsx_sc_hit *parts, int32_t nparts, int32_t nq, int32_t k, rsx_sc_hit *out);
/* the same on the GPU (d_parts is what an RCCL all-gather of d_out produces) */
int rsx_sc_merge_topk_device(rsx_sc *h, const rsx_sc_hit *d_parts, int32_t nparts, int32_t nq,
                             int32_t k, rsx_sc_hit *d_out, void *stream);
/* apply the loop threshold + yaw conversion of SC.cpp:401-417 to a top-1 record */
int rsx_sc_hit_to_loop(rsx_sc *h, const rsx_sc_hit *hit, int32_t *loop_id, float *yaw_diff_rad);

/* ---- one process, several GPUs (SURVEY 8e for a single C++ host such as alaserPGO, PGO.cpp:99,706-710) ----
 * rsx_scs = a ScanContext database sharded over the listed devices (keyframe i on shard i % n_devices), driven
 * by the calling process: one stream per device, the two-stage query above with the exchanges done as peer
 * copies over xGMI.  Results are identical to an unsharded handle.  The same device may be listed more than once
 * (several shards on one GPU: used by the tests on one-GPU boxes).  Internally synchronised. */
typedef struct rsx_scs rsx_scs;
int rsx_scs_create(const rsx_sc_params *p, const int32_t *devices, int32_t n_devices, rsx_scs **out);
/* The general form.  Layout: n_devices = query_groups x DB shards; device g belongs to query group g / S and holds DB
 * shard g % S (S = n_devices / query_groups; keyframe i on the shards i % S, once per group).  A batch is cut into
 * query_groups contiguous slices, each answered by its group: per-query costs shrink with the number of groups, per-pair
 * costs with the number of devices (rsx_scs_create = 1 group: every device a shard of ONE database copy).
 * exchange_kind: how the shards of a group exchange their 16-byte records in the two-stage query --
 *   RSX_SCS_EXCHANGE_PEER_COPY  hipMemcpyPeerAsync to the group's first device and back (default; no extra library)
 *   RSX_SCS_EXCHANGE_RCCL       ncclAllGather over the group's communicators, every shard merges for itself; librccl.so is
 *                               loaded on first use (dlopen); devices inside a group must be distinct
 * Results are identical in every layout and with either exchange. */
#define RSX_SCS_EXCHANGE_PEER_COPY 0
#define RSX_SCS_EXCHANGE_RCCL 1
int rsx_scs_create_layout(const rsx_sc_params *p, const int32_t *devices, int32_t n_devices, int32_t query_groups, int32_t exchange_kind,
                          rsx_scs **out);
int rsx_scs_num_query_groups(rsx_scs *h);
int rsx_scs_destroy(rsx_scs *h);
int rsx_scs_num_shards(rsx_scs *h); /* DB shards per query group */
int rsx_scs_set_dist_thres(rsx_scs *h, double thres);
int rsx_scs_size(rsx_scs *h, int64_t *n_global);
int rsx_scs_add_points(rsx_scs *h, const void *pts, size_t n, size_t stride_bytes, int32_t *out_index);
int rsx_scs_add_descriptors_f32(rsx_scs *h, const float *descs, int64_t n);
int rsx_scs_get_descriptor(rsx_scs *h, int64_t index, double *out_colmajor);
/* exhaustive batched query, host buffers in and out (synchronous); same contract as rsx_sc_query */
int rsx_scs_query(rsx_scs *h, const float *q_descs, int32_t nq, int32_t k, int64_t n_eligible, rsx_sc_hit *out);
/* detectLoopClosureID over the sharded database in EXHAUSTIVE mode (frozen prefix, 30-exclusion, threshold and yaw
 * as the reference; every entry of the prefix scored) */
int rsx_scs_detect_loop_closure(rsx_scs *h, rsx_sc_detection *out);


/* ============================== ORORA registration ======================================
 * Replaces the solver stage of the upstream file-based `odometry.cpp` entry (reference
 * README.md:27; package `orora`, launch/navtech_radar_slam_mulran.launch:5-8).  The ORORA sources
 * are an empty submodule in the reference checkout (.gitmodules:1-3), so these entry points follow
 * the published algorithm (oracle/orora_ref.h) -- parity unpinned.
 * Stateless per call; a handle only owns a stream and staging buffers. */

typedef struct rsx_orora rsx_orora;

typedef struct {
  double tim_noise_bound;        /* bound on ||b - R a|| of an inlier TIM (2 x point noise bound) */
  double noise_bound_radial;     /* A-COTE radial bound [m] */
  double noise_bound_tangential; /* A-COTE tangential bound [rad] (x range) */
  double gnc_factor;             /* mu growth per GNC iteration (1.4) */
  double cost_threshold;         /* GNC stops when |cost - prev_cost| < this */
  int32_t max_iterations;        /* GNC iteration cap */
  int32_t flags;                 /* RSX_ORORA_*: the modelling choices the (absent) upstream source would pin */
} rsx_orora_params;
#define RSX_ORORA_COMPLETE_GRAPH 1 /* rotation TIMs on all K (K-1) / 2 pairs instead of the ring of K (O(K^2) per GNC iteration) */
#define RSX_ORORA_TEASER_COST 2    /* scalar TLS cost in TEASER++'s form: unweighted residuals + sum of the outliers' bounds */
#define RSX_ORORA_PMC 4            /* max-clique inlier selection before the solver (upstream: "PMC max-clique prune"): the matches are
                                      pruned to a maximum clique of the consistency graph  i ~ j  <=>
                                      | ||src_i - src_j|| - ||dst_i - dst_j|| | < tim_noise_bound  (csrc/pmc.hip, oracle/pmc_ref.h).
                                      rsx_odometry_default_params turns it on; the device entry needs rsx_orora_reserve first */

typedef struct {
  double x, y, yaw;      /* dst = R(yaw) src + (x, y) */
  int32_t iterations;    /* GNC iterations executed */
  int32_t rot_inliers;   /* TIMs with final weight >= 0.5 */
  int32_t trans_inliers; /* matches inside both axis intervals at the estimate */
  int32_t status;        /* 0 ok; 1 fewer than 2 matches (identity); 2 more than
                            rsx_orora_max_correspondences() = 16384 matches (identity).  Pairs of up to 2048 matches
                            run on-chip (LDS), larger ones through an HBM workspace: same results */
} rsx_orora_result;

/* what the selection did to one pair */
typedef struct {
  int32_t size;     /* matches handed to the solver */
  int32_t max_core; /* largest core number of the consistency graph: no clique is larger than max_core + 1 */
  int32_t seeds;    /* greedy seeds started (<= 2) */
  int32_t flags;    /* RSX_ORORA_PMC_* */
} rsx_orora_pmc_info;
#define RSX_ORORA_PMC_PROVEN 1       /* size == max_core + 1: the clique is a maximum one */
#define RSX_ORORA_PMC_PASSTHROUGH 2  /* fewer than 2 or more than rsx_orora_max_clique_matches() matches: not pruned */
#define RSX_ORORA_PMC_NO_WORKSPACE 4 /* (with PASSTHROUGH) the pair did not fit what rsx_orora_reserve sized: not pruned */

int rsx_orora_default_params(rsx_orora_params *p);
int rsx_orora_max_correspondences(void);
int rsx_orora_max_clique_matches(void); /* 2048: pairs with more matches go to the solver unpruned */
/* Sizes the workspaces of the RSX_ORORA_PMC stage for calls of up to max_total_matches matches (sum over the pairs of a call), so
 * that the asynchronous device entry never allocates.  The host-buffer entries size them by themselves. */
int rsx_orora_reserve(rsx_orora *h, int64_t max_total_matches);
/* The selection on its own (params->tim_noise_bound is the consistency bound; NULL = defaults): out_member[m] = 1 for the
 * matches kept, laid out like the matches; out_info[n_pairs] (either may be NULL).  Host buffers, synchronous. */
int rsx_orora_max_clique_batch(rsx_orora *h, const float *src_xy, const float *dst_xy, const int64_t *offsets, int32_t n_pairs,
                               const rsx_orora_params *params, uint8_t *out_member, rsx_orora_pmc_info *out_info);
/* device buffers, asynchronous on `stream` */
int rsx_orora_max_clique_batch_device(rsx_orora *h, const float *d_src_xy, const float *d_dst_xy, const int64_t *d_offsets,
                                      int32_t n_pairs, const rsx_orora_params *params, uint8_t *d_member, rsx_orora_pmc_info *d_info,
                                      void *stream);
/* the info records of the last rsx_orora_register_batch{,_device} call that ran with RSX_ORORA_PMC (synchronises the handle's
 * last stream; n_pairs as in that call) */
int rsx_orora_last_pmc_info(rsx_orora *h, rsx_orora_pmc_info *out_info, int32_t n_pairs);
int rsx_orora_create(int device, rsx_orora **out);
int rsx_orora_destroy(rsx_orora *h);
/* n_pairs scan pairs; pair i owns matches [offsets[i], offsets[i+1]) of the concatenated
 * src_xy/dst_xy arrays (float x,y per match).  Host buffers, synchronous. */
int rsx_orora_register_batch(rsx_orora *h, const float *src_xy, const float *dst_xy, const int64_t *offsets,
                             int32_t n_pairs, const rsx_orora_params *params, rsx_orora_result *out);
/* device buffers, asynchronous on `stream` */
int rsx_orora_register_batch_device(rsx_orora *h, const float *d_src_xy, const float *d_dst_xy,
                                    const int64_t *d_offsets, int32_t n_pairs, const rsx_orora_params *params,
                                    rsx_orora_result *d_out, void *stream);

/* ============================== cen2019 keypoint extraction ============================
 * Replaces the feature-extraction stage of the upstream file-based `odometry.cpp` entry
 * (reference README.md:27,29).  Source absent from the reference checkout (empty submodule):
 * follows the published method (oracle/cen2019_ref.c) -- parity unpinned. */

typedef struct rsx_cen2019 rsx_cen2019;

typedef struct {
  int32_t max_points; /* budget of marked regions (10000) */
  int32_t min_range;  /* first range bin considered for keypoints (58) */
} rsx_cen2019_params;

int rsx_cen2019_default_params(rsx_cen2019_params *p);
/* one handle per image shape: rows azimuths x cols range bins (MulRan/Navtech: 400 x 3360) */
int rsx_cen2019_create(int device, int32_t rows, int32_t cols, rsx_cen2019 **out);
int rsx_cen2019_destroy(rsx_cen2019 *h);
/* img: rows x row_stride bytes (host), power samples at [col_offset, col_offset+cols) of each row
 * (col_offset = 11 for MulRan polar_oxford_form rows).  out_targets: (azimuth idx, range idx) int32
 * pairs in row-major order.  If azimuths (rows floats, rad) is given, out_xy (optional) receives
 * x = (r+0.5)*resolution*cos(az), y = ...*sin(az) -- the /orora/cloud_local points.
 * *out_count = keypoints found (only the first max_targets are written). */
int rsx_cen2019_extract(rsx_cen2019 *h, const uint8_t *img, int32_t row_stride, int32_t col_offset,
                        const rsx_cen2019_params *params, const float *azimuths, float resolution,
                        int32_t *out_targets, float *out_xy, int32_t max_targets, int32_t *out_count);
/* The same for n_images scans in ONE chain of launches (a file-based sequence is known in advance: README.md:27,54-60).
 * Images image_stride_bytes apart; azimuths: rows floats shared by all images, or n_images x rows when
 * azimuths_per_image != 0.  out_targets [n_images][max_targets][2], out_xy [n_images][max_targets][2] (optional),
 * out_counts [n_images].  Host buffers, synchronous. */
int rsx_cen2019_extract_batch(rsx_cen2019 *h, const uint8_t *imgs, int32_t n_images, int64_t image_stride_bytes, int32_t row_stride,
                              int32_t col_offset, const rsx_cen2019_params *params, const float *azimuths, int32_t azimuths_per_image,
                              float resolution, int32_t *out_targets, float *out_xy, int32_t max_targets, int32_t *out_counts);
/* Device buffers in and out, asynchronous on `stream`, no host synchronisation anywhere in the chain: d_targets / d_xy /
 * d_counts as above in HBM (d_xy, d_counts optional), ready for rsx_frontend_describe_device without a host hop. */
int rsx_cen2019_extract_batch_device(rsx_cen2019 *h, const uint8_t *d_imgs, int32_t n_images, int64_t image_stride_bytes,
                                     int32_t row_stride, int32_t col_offset, const rsx_cen2019_params *params, const float *d_azimuths,
                                     int32_t azimuths_per_image, float resolution, int32_t *d_targets, float *d_xy, int32_t max_targets,
                                     int32_t *d_counts, void *stream);

/* ============================== ORORA front end ========================================
 * The steps between the cen2019 keypoints and the solver in the upstream file-based entry (reference README.md:26-29;
 * SURVEY 8f rank 3): polar -> Cartesian image, ORB-style binary descriptors at the keypoints, brute-force Hamming
 * knnMatch(k = 2) + ratio test.  Upstream uses OpenCV for all three; the sources are absent from the reference checkout,
 * so these follow the published steps as restated in oracle/frontend_ref.c -- parity unpinned (in particular the 256
 * test pairs come from a seeded generator, not OpenCV's learned table: not byte-compatible with cv::ORB). */

typedef struct rsx_frontend rsx_frontend;

typedef struct {
  int32_t cart_pixel_width; /* W: the Cartesian image is W x W, centred on the sensor (964) */
  float cart_resolution;    /* metres per pixel (0.2592) */
  float ratio;              /* default nearest-neighbour distance ratio of rsx_frontend_match callers (0.8) */
  int32_t flags;            /* RSX_FRONTEND_*; 0 = default */
} rsx_frontend_params;
#define RSX_FRONTEND_THREE_PASS 1     /* remap, blur rows, blur columns as three kernels instead of the one strip kernel (same images) */
#define RSX_FRONTEND_EXACT_AZIMUTH 2  /* decide every pixel's azimuth row by the fp64 division (default: a reciprocal multiply with a proven \
                                         margin, the division only where the margin does not decide; same images) */
#define RSX_FRONTEND_TILES 4          /* the 32 x 32 LDS-tile kernel of round 3 instead of the strip kernel (same images) */

int rsx_frontend_default_params(rsx_frontend_params *p);
/* one handle per polar image shape (rows azimuths x cols range bins) */
int rsx_frontend_create(int device, int32_t rows, int32_t cols, const rsx_frontend_params *params, rsx_frontend **out);
int rsx_frontend_destroy(rsx_frontend *h);
/* polar image (host bytes, same layout arguments as rsx_cen2019_extract; azimuths = rows floats in rad, increasing)
 * -> Cartesian fp32 image, kept on the GPU together with its 7 x 7 Gaussian-smoothed copy for rsx_frontend_describe;
 * out_cart (optional, W * W floats, row 0 = farthest forward, forward = azimuth 0, azimuth grows to the right) */
int rsx_frontend_cartesian(rsx_frontend *h, const uint8_t *img, int32_t row_stride, int32_t col_offset, const float *azimuths,
                           float resolution, float *out_cart);
/* 256-bit descriptors of n keypoints given in metres in the sensor frame (x forward, y right: the out_xy of
 * rsx_cen2019_extract) on the last image: out_desc n x 32 bytes, out_valid[i] = 0 when the patch leaves the image */
int rsx_frontend_describe(rsx_frontend *h, const float *xy, int32_t n, uint8_t *out_desc, uint8_t *out_valid);
/* knnMatch(k = 2) + ratio test of nq query descriptors against nt train descriptors: out_train_idx[i] = the nearest
 * train descriptor if d1 < ratio * d2, else -1; out_d1 / out_d2 (optional) the two smallest Hamming distances (-1: none) */
int rsx_frontend_match(rsx_frontend *h, const uint8_t *q_desc, const uint8_t *q_valid, int32_t nq, const uint8_t *t_desc,
                       const uint8_t *t_valid, int32_t nt, float ratio, int32_t *out_train_idx, int32_t *out_d1, int32_t *out_d2);

/* the batched / device-resident forms of the three calls above (what rsx_odometry chains; usable on their own):
 * Cartesian images + smoothed copies of n_images scans resident in HBM, kept in the handle's slots 0 .. n_images-1
 * (azimuths: HOST array, rows floats, ONE grid used for every image of the batch; per-image grids: the _az form below) */
int rsx_frontend_cartesian_batch_device(rsx_frontend *h, const uint8_t *d_imgs, int32_t n_images, int64_t image_stride_bytes, int32_t row_stride,
                                        int32_t col_offset, const float *azimuths, float resolution, void *stream);
/* the same with the azimuth grids in DEVICE memory, one per image: image i samples the grid at d_azimuths + i *
 * azimuth_stride_floats (its first two entries give start and step; 0: one grid for every image).  MulRan scans carry their
 * own encoder grid; a map built from the first scan's grid would rotate every other scan's Cartesian image against its own
 * keypoints.  Nothing is read on the host: no synchronisation, whatever the grids do from window to window. */
int rsx_frontend_cartesian_batch_device_az(rsx_frontend *h, const uint8_t *d_imgs, int32_t n_images, int64_t image_stride_bytes,
                                           int32_t row_stride, int32_t col_offset, const float *d_azimuths, int64_t azimuth_stride_floats,
                                           float resolution, void *stream);
/* descriptors of the keypoints of every image of that batch: d_xy [n_images][max_targets][2] and d_counts [n_images] as
 * rsx_cen2019_extract_batch_device leaves them -> d_desc [n_images][max_targets][32], d_valid [n_images][max_targets] */
int rsx_frontend_describe_batch_device(rsx_frontend *h, const float *d_xy, const int32_t *d_counts, int32_t n_images, int32_t max_targets,
                                       uint8_t *d_desc, uint8_t *d_valid, void *stream);
/* knnMatch(2) + ratio between CONSECUTIVE keypoint sets of [slot][max_targets] arrays: pair j = (slot first_slot + j,
 * slot first_slot + j + 1); d_fwd [n_pairs][max_targets]: for every keypoint of the first slot its match in the second
 * (or -1), d_bwd the reverse direction */
int rsx_frontend_match_consecutive_device(rsx_frontend *h, const uint8_t *d_desc, const uint8_t *d_valid, const int32_t *d_counts,
                                          int32_t max_targets, int32_t first_slot, int32_t n_pairs, float ratio, int32_t *d_fwd,
                                          int32_t *d_bwd, void *stream);

/* ============================== file-based odometry pipeline ============================
 * The main loop of the upstream file-based `odometry.cpp` entry (reference README.md:26-29,54-60; launched through
 * $(find orora)/launch/run_orora.launch, launch/navtech_radar_slam_mulran.launch:5-8): polar scan -> cen2019 keypoints ->
 * Cartesian image + ORB-style descriptors -> knnMatch(2) + ratio + cross check against the previous scan -> ORORA.
 * The sequence is on disk, so a caller hands over WINDOWS of consecutive scans: every stage runs once per window for all
 * scans / all consecutive pairs, everything between the image upload and the 48-byte result per scan stays in HBM, and
 * ORORA registers all pairs of a window in one batch.  The handle remembers the last scan (on the device), so the pair
 * that straddles two calls is registered too.  Pose composition (sequential, trivial) is the caller's.  Sources of the
 * upstream entry are absent from the reference checkout (empty submodule) -- parity unpinned. */

typedef struct rsx_odometry rsx_odometry;

typedef struct {
  rsx_cen2019_params cen;
  rsx_frontend_params frontend;  /* frontend.ratio = the knnMatch ratio */
  rsx_orora_params orora;
  float radar_resolution;        /* metres per range bin (0.0595: Navtech CIR204-H / MulRan) */
  int32_t col_offset;            /* metadata bytes in front of the power samples of every row (11) */
  int32_t max_keypoints;         /* keypoints kept per scan (16384 = rsx_orora_max_correspondences(), its upper limit) */
  int32_t device;
} rsx_odometry_params;

typedef struct {
  rsx_orora_result reg;  /* motion between the previous scan and this one: p_previous = R(yaw) p_this + (x, y);
                            reg.status = 3 for the first scan of a sequence (nothing to register against) */
  int32_t n_keypoints;   /* cen2019 keypoints of this scan (only the first max_keypoints are used) */
  int32_t n_matches;     /* cross-checked ratio matches handed to ORORA */
} rsx_odometry_scan;

int rsx_odometry_default_params(rsx_odometry_params *p);
/* (device memory of a handle for 400 x 3360 scans: about 4 GB -- two extraction lanes of a 64-scan window each, three sets of
 * keypoints / descriptors, allocated at the first push; three windows of a sequence are in flight inside a call) */
int rsx_odometry_create(const rsx_odometry_params *params, int32_t rows, int32_t cols, rsx_odometry **out);
int rsx_odometry_destroy(rsx_odometry *h);
int rsx_odometry_reset(rsx_odometry *h); /* forget the previous scan: the next scan starts a new sequence */
int rsx_odometry_window(void);           /* scans per internal launch chain (longer calls are cut into such windows) */
/* n_scans consecutive scans, host images image_stride_bytes apart (rows x row_stride bytes each); azimuths: rows floats
 * (rad, increasing) shared by all scans or n_scans x rows when azimuths_per_image != 0.  out [n_scans]; out_xy
 * (optional) [n_scans][max_xy][2]: the scan's keypoints in metres in the sensor frame (/orora/cloud_local).  Synchronous. */
int rsx_odometry_push(rsx_odometry *h, const uint8_t *imgs, int32_t n_scans, int64_t image_stride_bytes, int32_t row_stride,
                      const float *azimuths, int32_t azimuths_per_image, rsx_odometry_scan *out, float *out_xy, int32_t max_xy);
/* the same with the images already resident in HBM (d_imgs: device pointer; everything else host) */
int rsx_odometry_push_device(rsx_odometry *h, const uint8_t *d_imgs, int32_t n_scans, int64_t image_stride_bytes, int32_t row_stride,
                             const float *azimuths, int32_t azimuths_per_image, rsx_odometry_scan *out, float *out_xy, int32_t max_xy);
/* page-locked host memory for image windows (decode threads write straight into it: the upload then runs at PCIe speed) */
int rsx_host_alloc_pinned(size_t bytes, void **out);
int rsx_host_free_pinned(void *p);

/* ============================== VoxelGrid downsample ===================================
 * pcl::VoxelGrid<pcl::PointXYZI>::filter with setLeafSize(leaf, leaf, leaf): the step right before
 * makeAndSaveScancontextAndKeys in the reference's keyframe path (PGO.cpp:98,482-484; leaf 0.4 set at
 * PGO.cpp:687-688).  PCL is not part of the reference checkout: follows the published algorithm
 * (oracle/voxelgrid_ref.c) -- parity unpinned.  Output = one centroid {x,y,z,intensity} per occupied
 * voxel, packed float4, in ascending voxel-index order.  If the voxel grid would overflow int32
 * ("leaf size too small") the input is returned unchanged, like PCL does. */

typedef struct rsx_voxelgrid rsx_voxelgrid;

int rsx_voxelgrid_create(int device, rsx_voxelgrid **out);
int rsx_voxelgrid_destroy(rsx_voxelgrid *h);
/* pts: n points stride_bytes apart, float x,y,z at byte offsets 0,4,8, float intensity at
 * intensity_offset (16 for pcl::PointXYZI; < 0: none, output 0).  Non-finite points are dropped.
 * *out_count = number of output points (only the first max_out are written). */
int rsx_voxelgrid_filter(rsx_voxelgrid *h, const void *pts, size_t n, size_t stride_bytes, int32_t intensity_offset,
                         float leaf, float *out_xyzi, int64_t max_out, int64_t *out_count);
/* downSizeFilterScancontext.filter(...) + scManager.makeAndSaveScancontextAndKeys(...) in one call
 * (PGO.cpp:482-492): the downsampled cloud never leaves the GPU.  vg and h must be on the same device. */
int rsx_sc_add_points_downsampled(rsx_sc *h, rsx_voxelgrid *vg, const void *pts, size_t n, size_t stride_bytes,
                                  float leaf, int32_t *out_index);

/* ============================== ICP loop verification ===================================
 * pcl::IterativeClosestPoint<PointXYZI, PointXYZI> as the reference configures it in
 * doICPVirtualRelative (PGO.cpp:371-392): point-to-point, SVD (Umeyama) step, PCL's default
 * convergence criteria, fitness = mean squared nearest-neighbour distance.  PCL is not part of the
 * reference checkout: follows the published algorithm (oracle/icp_ref.c) -- parity unpinned. */

typedef struct rsx_icp rsx_icp;

typedef struct {
  double max_corr_dist;             /* setMaxCorrespondenceDistance(150), PGO.cpp:374 */
  double transformation_epsilon;    /* setTransformationEpsilon(1e-6),    PGO.cpp:376 */
  double euclidean_fitness_epsilon; /* setEuclideanFitnessEpsilon(1e-6),  PGO.cpp:377 */
  int32_t max_iterations;           /* setMaximumIterations(100),         PGO.cpp:375 */
  int32_t reserved;
} rsx_icp_params;

typedef struct {
  float transform[16]; /* getFinalTransformation(): row-major 4x4, target <- source */
  double fitness;      /* getFitnessScore() */
  int32_t iterations;
  int32_t converged;   /* hasConverged() */
  int32_t state;       /* 1 iterations, 2 transform, 3 abs MSE, 4 rel MSE, 5 not enough correspondences */
  int32_t reserved;
} rsx_icp_result;

int rsx_icp_default_params(rsx_icp_params *p);
int rsx_icp_create(int device, rsx_icp **out);
int rsx_icp_destroy(rsx_icp *h);
/* icp.setInputSource(src); icp.setInputTarget(tgt); icp.align(unused, guess).  Points: float x,y,z at
 * byte offsets 0,4,8 of each stride; guess: optional row-major 4x4 (NULL = identity).  The caller
 * applies the acceptance test of PGO.cpp:385-387 (converged && fitness <= 0.3).  One launch: a persistent kernel that
 * occupies the whole device for the duration of the alignment (~0.5 ms).  Every kernel of this library that waits at a grid
 * barrier (this one and the cooperative VoxelGrid kernel behind rsx_voxelgrid_filter / rsx_loop_submap / rsx_loop_verify /
 * rsx_sc_add_keyframe) is ordered against the others ON THE DEVICE, across handles, streams and threads of one process
 * (csrc/rsx_persistent.h), and sizes its grid from the runtime's occupancy answer; two PROCESSES must not run such calls on
 * the same device at the same time (the barrier's 5 s watchdog turns that into RSX_ERR_HIP instead of a hang). */
int rsx_icp_align(rsx_icp *h, const void *src, size_t ns, size_t src_stride, const void *tgt, size_t nt, size_t tgt_stride,
                  const rsx_icp_params *params, const float *guess, rsx_icp_result *out);

/* ============================ keyframe clouds: loop verification and the map ==========================
 * What the pose-graph node does with the keyframe clouds it keeps (keyframeLaserClouds = the 0.4 m VoxelGrid output of
 * every keyframe, PGO.cpp:482-487), resident in HBM:
 *   rsx_loop_verify       = doICPVirtualRelative(loop, curr) (PGO.cpp:355-406): source = loopFindNearKeyframesCloud(curr, 0,
 *                           root = loop), target = loopFindNearKeyframesCloud(loop, 25, root = loop) -- keyframes
 *                           key - size .. key + size, EACH IN ITS OWN LOCAL FRAME, all moved by the one pose of the root
 *                           keyframe, concatenated, VoxelGrid 0.4 m (PGO.cpp:329-352) -- then ICP, the gate
 *                           `converged && fitness <= 0.3` (PGO.cpp:385), pcl::getTranslationAndEulerAngles and
 *                           poseFrom.between(poseTo) (PGO.cpp:400-407)
 *   rsx_kfstore_build_map = the cloud pubMap publishes (PGO.cpp:631-655): every SKIP_FRAMES-th keyframe through its own
 *                           pose, concatenated, VoxelGrid (host/pcd.h writes it to a .pcd file: the "resulting map save
 *                           function" of the reference's TODO list, README.md:139)
 * Poses are the reference's Pose6D: double x, y, z, roll, pitch, yaw (PGO.cpp:199-206).  PCL and GTSAM are not part of
 * the reference checkout; their pieces follow the published sources (oracle/loopverify_ref.c) -- parity unpinned; the
 * control flow is the reference's own.  Thread safety: one handle = one lock (the reference guards the same data with
 * mKF, PGO.cpp:339-341,486-495). */
typedef struct rsx_kfstore rsx_kfstore;

typedef struct {
  int32_t history_keyframe_search_num; /* 25: the target submap is loop - 25 .. loop + 25 (PGO.cpp:358) */
  float leaf;                          /* 0.4: downSizeFilterICP (PGO.cpp:687-689) */
  double fitness_threshold;            /* 0.3: loopFitnessScoreThreshold (PGO.cpp:384) */
  rsx_icp_params icp;                  /* PGO.cpp:374-378 */
} rsx_loop_verify_params;

typedef struct {
  int32_t accepted;    /* !(hasConverged() == false || getFitnessScore() > threshold)  (PGO.cpp:385) */
  int32_t converged, iterations, state; /* as rsx_icp_result */
  double fitness;
  float transform[16]; /* icp.getFinalTransformation(), row-major 4x4 */
  float x, y, z, roll, pitch, yaw; /* pcl::getTranslationAndEulerAngles of it (PGO.cpp:400-403) */
  double relative[16]; /* poseFrom.between(poseTo) with poseTo = identity (PGO.cpp:404-407), row-major 4x4: the loop factor */
  int64_t n_source, n_target; /* points of the two clouds after the VoxelGrid */
} rsx_loop_verify_result;

int rsx_kfstore_create(int device, rsx_kfstore **out);
int rsx_kfstore_destroy(rsx_kfstore *h);
/* keyframeLaserClouds.push_back(cloud) (PGO.cpp:487): n points stride_bytes apart, float x, y, z at byte offsets 0, 4, 8,
 * float intensity at intensity_offset (16 for pcl::PointXYZI; < 0: none, stored as 0).  The cloud is copied; *out_index =
 * its keyframe index.  _device: n packed float4 {x, y, z, intensity} already in this device's memory. */
int rsx_kfstore_add(rsx_kfstore *h, const void *pts, size_t n, size_t stride_bytes, int32_t intensity_offset, int32_t *out_index);
int rsx_kfstore_add_device(rsx_kfstore *h, const void *d_xyzi, size_t n, int32_t *out_index);
int rsx_kfstore_size(rsx_kfstore *h, int64_t *n_keyframes, int64_t *n_points);
/* keyframe `index` as packed float4; *out_count = its size (only the first max_out points are written) */
int rsx_kfstore_get(rsx_kfstore *h, int32_t index, float *out_xyzi, int64_t max_out, int64_t *out_count);
int rsx_loop_verify_default_params(rsx_loop_verify_params *p);
/* loopFindNearKeyframesCloud(out, key, submap_size, root) with root_pose6 = keyframePosesUpdated[root] (PGO.cpp:329-352) */
int rsx_loop_submap(rsx_kfstore *h, int32_t key, int32_t submap_size, const double *root_pose6, float leaf, float *out_xyzi,
                    int64_t max_out, int64_t *out_count);
/* doICPVirtualRelative(loop_idx, curr_idx) with root_pose6 = keyframePosesUpdated[loop_idx] (PGO.cpp:355-406);
 * params NULL = the reference's constants */
int rsx_loop_verify(rsx_kfstore *h, int32_t loop_idx, int32_t curr_idx, const double *root_pose6, const rsx_loop_verify_params *params,
                    rsx_loop_verify_result *out);
/* pubMap's cloud (PGO.cpp:631-655): keyframes 0, skip, 2 skip, ... < min(n_poses, stored), each through poses6[6 k .. 6 k + 5],
 * VoxelGrid `leaf`; *out_count = its size (only the first max_out points are written) */
int rsx_kfstore_build_map(rsx_kfstore *h, const double *poses6, int64_t n_poses, int32_t skip_frames, float leaf, float *out_xyzi,
                          int64_t max_out, int64_t *out_count);
/* downSizeFilterScancontext.filter + keyframeLaserClouds.push_back + scManager.makeAndSaveScancontextAndKeys in one call
 * (PGO.cpp:482-492): the downsampled cloud goes from the VoxelGrid into the keyframe store and the descriptor build
 * without leaving the GPU.  intensity_offset as above.  All three handles on the same device. */
int rsx_sc_add_keyframe(rsx_sc *h, rsx_voxelgrid *vg, rsx_kfstore *kf, const void *pts, size_t n, size_t stride_bytes,
                        int32_t intensity_offset, float leaf, int32_t *out_index);

#ifdef __cplusplus
}
#endif
#endif /* RSX_H */
