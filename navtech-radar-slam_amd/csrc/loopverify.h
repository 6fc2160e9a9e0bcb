// loopverify.h -- internal interface of the keyframe-cloud store (loopverify.hip) for sc_api.cpp.
#pragma once
#include <cstdint>
#include <mutex>

#include "rsx.h"

namespace rsx {
namespace kf {

std::mutex &mutex_of(rsx_kfstore *h);
int device_of(rsx_kfstore *h);
// keyframeLaserClouds.push_back: n packed float4 {x, y, z, intensity} that are complete in this device's memory; the caller
// holds the store's mutex.  Returns once the copy is done (the source may be overwritten).
int append_device_locked(rsx_kfstore *h, const void *d_xyzi, int64_t n, int32_t *out_index);

}  // namespace kf
}  // namespace rsx
