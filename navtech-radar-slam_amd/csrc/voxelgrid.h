// voxelgrid.h -- internal interface of the VoxelGrid handle (voxelgrid.hip) for sc_api.cpp.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <mutex>

#include "rsx.h"

namespace rsx {
namespace vg {

// host points -> device, downsample; *d_out = packed float4 {x,y,z,intensity} owned by the handle
// (valid until its next call), *n_out = number of output points.  Synchronises the handle's stream.
int upload_and_filter(rsx_voxelgrid *h, const void *pts, int64_t n, int64_t stride, int32_t ioff, float leaf, int64_t max_out,
                      const float **d_out, int64_t *n_out);
// the same on points resident in device memory (complete, or produced on stream s); the caller holds the handle's mutex
int filter_device(rsx_voxelgrid *h, const void *d_pts, int64_t n, int64_t stride, int32_t ioff, float leaf, int64_t max_out,
                  const float **d_out, int64_t *n_out, hipStream_t s);
// the same without returning to the host, for one or two clouds in ONE launch on stream s (the callers hold the handles'
// mutexes): T (optional) is applied to every point first -- local2global of PGO.cpp:199-220, bit-exact float -- and the
// number of output points stays in device memory, for whoever is enqueued next on s
struct Mat34 {
  float m[12];  // row-major 3 x 4
};
struct JobIn {
  rsx_voxelgrid *h;  // owns the output and the scratch of this cloud (two clouds: two handles)
  const void *d_pts;
  int64_t n, stride;
  int32_t ioff;
  const Mat34 *T;
  float leaf;
  int64_t max_out;
};
struct DeviceCloud {
  const float *d_out;          // packed float4 {x, y, z, intensity}
  const long long *d_count;    // device memory
};
int enqueue(const JobIn *jobs, int njobs, hipStream_t s, DeviceCloud *out);
std::mutex &mutex_of(rsx_voxelgrid *h);
hipStream_t stream_of(rsx_voxelgrid *h);
int device_of(rsx_voxelgrid *h);

}  // namespace vg
}  // namespace rsx
