// icp.h -- internal interface of the ICP handle (icp.hip) for loopverify.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <mutex>

#include "rsx.h"

namespace rsx {
namespace icp {

std::mutex &mutex_of(rsx_icp *h);
hipStream_t stream_of(rsx_icp *h);
int device_of(rsx_icp *h);
// pcl::IterativeClosestPoint::align on clouds resident in device memory (float x, y, z at byte offsets 0, 4, 8 of each
// stride).  The caller holds the handle's mutex; the clouds are complete or produced on the handle's stream.
int align_device_locked(rsx_icp *h, const void *d_src, int64_t n_s, int64_t src_stride, const void *d_tgt, int64_t n_t, int64_t tgt_stride,
                        const rsx_icp_params *params, const float *guess, rsx_icp_result *out);

// the same with the sizes of the clouds in device memory (d_ns / d_nt, optional; written by work enqueued before this on the
// handle's stream; n_s / n_t are then upper bounds): one launch, one read-back; *ns_out / *nt_out = the sizes it ran on
int align_device_counts_locked(rsx_icp *h, const void *d_src, int64_t n_s, const long long *d_ns, int64_t src_stride, const void *d_tgt,
                               int64_t n_t, const long long *d_nt, int64_t tgt_stride, const rsx_icp_params *params, const float *guess,
                               rsx_icp_result *out, int64_t *ns_out, int64_t *nt_out);

}  // namespace icp
}  // namespace rsx
