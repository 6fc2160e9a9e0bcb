// sc_plan.h -- how a query batch is cut up: the pieces of the host-buffer entry, the batches of the filter workspaces and
// the XCD-aware work split of sc_spec2_filter_kernel.  Pure integer functions without a device or library dependency, so that
// tests/test_plan_logic.py can compile them with the host compiler and check their invariants (every query in exactly one
// piece / batch / workgroup) over a sweep of sizes.
#pragma once
#include <cstdint>

namespace rsx {
namespace sc {
namespace plan {

// Pieces of rsx_sc_query (sc_api.cpp): the first piece `first` queries (its upload is the only one nothing hides), every later
// one growth_x10 / 10 times the one before, multiples of 64; a remainder smaller than half a piece joins the piece before it,
// the last of max_pieces slots takes what is left.  Batches below 4 x first stay whole (first <= 0: always).  -> number of pieces
inline int host_pieces(int32_t nq, int first, int growth_x10, int max_pieces, int32_t *sizes) {
  if (growth_x10 < 10) growth_x10 = 10;
  if (first <= 0 || nq < 4 * (int64_t)first || max_pieces < 2) {
    sizes[0] = nq;
    return 1;
  }
  int n = 0;
  int32_t left = nq;
  double want = first;
  while (left > 0) {
    int32_t take = ((int32_t)want + 63) / 64 * 64;
    if (n == max_pieces - 1 || left - take < take / 2) take = left;
    sizes[n++] = take;
    left -= take;
    want *= growth_x10 / 10.0;
  }
  return n;
}

// Query batch the filter workspaces are sized for: <= 2^29 bound elements (1 GiB of fp16) per batch, at least 64 queries, and
// the batches of one call equally long (8192 queries against 100 000 entries used to run as 3 x 2684 + 140)
inline int64_t filter_batch(int64_t n_items, int64_t nq) {
  const int64_t ld = (n_items + 31) / 32 * 32;
  int64_t qb = ld > 0 ? (1ll << 29) / ld : nq;
  qb = qb < 64 ? 64 : qb / 64 * 64;  // whole multiples of 64 queries
  if (qb >= nq) return nq;
  const int64_t nb = (nq + qb - 1) / qb;
  return ((nq + nb - 1) / nb + 63) / 64 * 64;  // <= qb: qb is a multiple of 64 and >= nq / nb
}

// XCD-aware split of sc_spec2_filter_kernel (sc_spec.hip): nqt query tiles, ntb tile-blocks, n_cu compute units in 8 XCDs.
// on: the split is used; XCD x owns query tiles [x * nqt_x, (x + 1) * nqt_x) in sub-ranges of len tiles, workgroup b of `grid`
// belongs to XCD b % 8 and is (sub-range (b / 8) / ntb, tile-block (b / 8) % ntb).  forced_s > 0: that many sub-ranges,
// whatever the cost.
struct XcdSplit {
  bool on;
  int32_t nqt_x, len;
  unsigned grid;
};
inline XcdSplit xcd_split(int64_t nqt, int64_t ntb, int n_cu, int64_t forced_s = 0) {
  XcdSplit r{false, 0, 0, 0u};
  const int64_t nqt_x = (nqt + 7) / 8;
  if (nqt_x < 16 || n_cu % 8 != 0 || ntb < 1) return r;
  int64_t best_s = 1;
  double best_cost = 1e300;
  const int64_t slots = n_cu / 8;
  for (int64_t S = 1; S <= 64 && (S == 1 || (nqt_x + S - 1) / S >= 16); S++) {
    const int64_t len = (nqt_x + S - 1) / S, used = (nqt_x + len - 1) / len;
    const int64_t cost = ((used * ntb + slots - 1) / slots) * (len + 3);
    if ((double)cost < best_cost * 0.995) {
      best_cost = (double)cost;
      best_s = used;
    }
  }
  if (forced_s >= 1 && forced_s <= nqt_x) best_s = forced_s;
  // against the contiguous split (perfectly balanced, one start per workgroup): only where whole rounds cost <= 6 %
  const double contiguous = (double)(ntb * nqt) / (double)n_cu + 3.0;
  if (!(best_cost <= 1.06 * contiguous) && forced_s <= 0) return r;
  r.on = true;
  r.nqt_x = (int32_t)nqt_x;
  r.len = (int32_t)((nqt_x + best_s - 1) / best_s);
  const int64_t used = (nqt_x + r.len - 1) / r.len;
  r.grid = (unsigned)(8 * used * ntb);
  return r;
}

}  // namespace plan
}  // namespace sc
}  // namespace rsx
