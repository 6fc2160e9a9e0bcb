/*
 * rsx_diag.h -- diagnostic, parity-test and profiling entries of librsx.so.  NOT part of the drop-in boundary: nothing a
 * maintainer binds to replace the reference's SCManager / odometry calls is declared here (that is include/rsx.h).  These
 * are what tests/, bench.py and tools/ use to look inside the path: the filter's lower bounds pair by pair, the window
 * stage's previews, the ring-key tree as nanoflann lays it out, hipEvent timing of the dominant kernel, work counters of
 * the re-scoring stage, and the self-test of the exception firewall.  Same conventions as rsx.h (status returns, caller
 * owns every buffer, no C++ exception crosses the ABI).
 */
#ifndef RSX_DIAG_H
#define RSX_DIAG_H

#include "rsx.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Self-test of the exception firewall (every status-returning entry is a function-try-block, rsx_common.h): throws the
 * named exception INSIDE the library and returns what the firewall makes of it -- 0: a host container asked for an
 * impossible size (std::length_error) -> RSX_ERR_OOM; 1: std::bad_alloc -> RSX_ERR_OOM; 2: std::runtime_error (what
 * nanoflann throws through the reference's SCManager, NF.hpp:1228,1324) -> RSX_ERR_INTERNAL; 3: a non-std exception ->
 * RSX_ERR_INTERNAL.  Needs no device. */
int rsx_selftest_firewall(int kind);

/* Parity helper of the ORORA front end: Cartesian image `image` of the handle's last rsx_frontend_cartesian* call and its
 * 7 x 7 Gaussian-smoothed copy (W x W floats each; either pointer may be null) -- the smoothed image is otherwise only
 * observable through the descriptors that sample it (oracle/frontend_ref.c: feref_cart_remap, feref_blur). */
int rsx_frontend_read_images(rsx_frontend *h, int32_t image, float *out_cart, float *out_blur);

/* Introspection of the candidate stage (host only, no device needed): the ring-key search tree the detector would build
 * over `n` keys of 20 floats -- nanoflann's tree (KDTreeVectorOfVectorsAdaptor.h:49-117, leaf size 10, Scancontext.cpp:284,356)
 * rebuilt node for node, because the order in which tied neighbours come back is the order of its leaves.  out_vind[n] =
 * the permutation of the keys as planeSplit leaves it (nanoflann.hpp:968-1004); optional out_n_nodes / out_depth. */
int rsx_sc_ringkey_tree_layout(const float *keys20, int64_t n, int32_t *out_vind, int32_t *out_n_nodes, int32_t *out_depth);

/* parity helper for the MFMA filter: out_lb[q * n_local + slot] = the filter's lower bound of
 * dist(query q, local slot) for every local entry (host out).  Contract checked by the tests:
 * out_lb - rsx_sc_filter_eps() <= the exact distance, for every pair. */
int rsx_sc_filter_bounds(rsx_sc *h, const float *q_descs, int32_t nq, float *out_lb);
double rsx_sc_filter_eps(void);

/* diagnostic entry of the stage between the filter and the exact re-scoring (csrc/sc_window.hip): for each query the first
 * RSX_SC_WINDOW_P entries of its short list (local slots in ascending filter-bound order, -1 past the end; out_counts[q] of
 * them are valid) with the sector-key alignment k* (fastAlignUsingVkey, reference SC.cpp:93-113) and the fp16 matrix-core
 * preview pv of distanceBtnScanContext (SC.cpp:116-148): |pv - distance| <= RSX_SC_WINDOW_MARGIN.  Where the alignment is
 * not unique within the kernel's error bound, k* = -1 and pv - RSX_SC_WINDOW_MARGIN is a lower bound of the distance only;
 * pv = NaN for non-finite data, +inf where no shift of the window has an effective column.  Past the first 128 positions
 * an entry only gets a record when its filter bound can still reach the top-k (k as in the query call), judged by the
 * previews of the first 128; the others carry k* = -2, pv = NaN.  out_shift_mask (k* >= 0 only): bit t set = the window shift
 * k* - 3 + t can be the minimum; the exact evaluation skips the others (their preview is more than two margins above the best
 * one, so they are strictly worse).  All outputs are [nq][RSX_SC_WINDOW_P] host arrays. */
#define RSX_SC_WINDOW_P 320
#define RSX_SC_WINDOW_MARGIN 1.25e-3f
int rsx_sc_window_previews(rsx_sc *h, const float *q_descs, int32_t nq, int32_t k, int32_t *out_slots, float *out_pv,
                           int32_t *out_kstar, int32_t *out_shift_mask, int32_t *out_counts);

/* instrumentation for bench.py: name of the dominant kernel (as rocprofv3 reports it) and, when
 * enabled, hipEvent pairs recorded around every launch of it on the stream it runs on.
 * rsx_sc_profile_read synchronises, returns launches and summed milliseconds since the last read,
 * and resets the counters. */
const char *rsx_sc_dominant_kernel_name(void);
/* name of the kernel the profiler events of this handle bracketed in its last exhaustive query:
 * "sc_filter_kernel" when the query went through the filter, else "sc_pair_kernel" */
const char *rsx_sc_profiled_kernel_name(rsx_sc *h);
int rsx_sc_profile_enable(rsx_sc *h, int on);
int rsx_sc_profile_read(rsx_sc *h, int64_t *launches, double *total_ms);
/* While profiling is enabled the stages behind the filter also count their work (device counters, summed and reset by this
 * call, which synchronises the device).  ONE versioned struct: the caller sets struct_size = sizeof(rsx_sc_rescoring_stats)
 * as it was compiled; the library fills the fields that fit and never writes past struct_size, so fields can be appended
 * without a new entry point (rounds 2-5 had grown three: _rescoring, _rescoring2, _rescoring3). */
typedef struct {
  uint32_t struct_size;        /* in: sizeof(rsx_sc_rescoring_stats) of the caller */
  uint32_t reserved;
  int64_t candidates;          /* short-list entries that went through the cheap phase (alignment + preview) */
  int64_t exact_evals;         /* exact fp64 pair evaluations (distanceBtnScanContext, SC.cpp:116-148) */
  int64_t queries_rescored;    /* (query, launch) pairs that scored at least one candidate */
  int64_t window_previews;     /* candidates whose alignment + preview came from the matrix-core window kernel (sc_window.hip) */
  int64_t valu_previews;       /* candidates that needed the per-wavefront VALU alignment + fp32 preview */
  int64_t exact_window_shifts; /* window shifts evaluated exactly (<= 7 per exact evaluation, SC.cpp:131-139) */
} rsx_sc_rescoring_stats;
int rsx_sc_profile_read_rescoring(rsx_sc *h, rsx_sc_rescoring_stats *out);

#ifdef __cplusplus
}
#endif
#endif /* RSX_DIAG_H */
