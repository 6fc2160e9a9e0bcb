// pmc.h -- host interface of csrc/pmc.hip: the max-clique inlier selection between the matcher and the ORORA solver
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "rsx_common.h"

namespace rsx {
namespace pmc {

constexpr int MAX_K = 2048;      // matches per pair the stage prunes (a vertex set = 64 lanes x 32 bits); larger pairs pass through
constexpr int MAX_SEEDS = 2;     // greedy seeds per pair (vertices of the clique in hand are not seeds)
constexpr size_t SLAB_BYTES = (size_t)MAX_K * 256;  // adjacency of one pair: MAX_K rows of 64 words
constexpr int META_BYTES = 8 * MAX_K + 256;      // per pair between the kernels: core numbers, order, degrees, header
constexpr int CHUNK = 4096;                      // pairs per launch group (2 GiB of slabs); larger batches run as several

struct Workspace {
  rsx::DevBuf slabs;  // one adjacency slab per pair of a chunk
  rsx::DevBuf meta;   // one record per pair of a chunk
  void release() {
    slabs.release();
    meta.release();
  }
};

// member (optional): 1 / 0 per match, laid out like the matches; info (optional): one per pair; sel_src / sel_dst / sel_cnt
// (optional, all or none; sel_cap = matches they hold): the selected matches of pair i at [offsets[i], offsets[i] + sel_cnt[i])
// in their original order; sel_cnt[i] = -1 for a pair that passed through unpruned (nothing is copied for it)
int launch(Workspace &ws, int device, const float2 *d_src, const float2 *d_dst, const int64_t *d_offsets, int n_pairs, double tau,
           uint8_t *d_member, rsx_orora_pmc_info *d_info, float2 *d_sel_src, float2 *d_sel_dst, int32_t *d_sel_cnt, int64_t sel_cap, hipStream_t s);

}  // namespace pmc
}  // namespace rsx
