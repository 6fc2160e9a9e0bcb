// rsx_persistent.cpp -- see rsx_persistent.h
#include "rsx_persistent.h"

#include "rsx_common.h"

namespace rsx {
namespace persistent {

namespace {
constexpr int kMaxDevices = 64;
struct PerDevice {
  std::mutex mu;
  hipEvent_t ev = nullptr;  // completion of the last grid-barrier kernel launched on this device by this process
  bool recorded = false;
};
PerDevice &slot(int device) {
  static PerDevice table[kMaxDevices];
  return table[(device < 0 || device >= kMaxDevices) ? 0 : device];
}
}  // namespace

int resident_limit(const void *kernel, int block, size_t dyn_lds, int device) {
  int cus = 0, per_cu = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus < 1) return 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block, dyn_lds) != hipSuccess || per_cu < 1) return 0;
  if (per_cu > 8) per_cu = 8;
  if (per_cu > 1) per_cu -= 1;
  return per_cu * cus;
}

Gate::Gate(int device, hipStream_t s) : device_(device), s_(s), st_(RSX_OK) {
  PerDevice &d = slot(device_);
  d.mu.lock();
  if (!d.ev) {
    hipError_t e = hipEventCreateWithFlags(&d.ev, hipEventDisableTiming);
    if (e != hipSuccess) {
      d.ev = nullptr;
      st_ = rsx::fail(RSX_ERR_HIP, "persistent gate: %s", hipGetErrorString(e));
      return;
    }
  }
  if (d.recorded) {
    hipError_t e = hipStreamWaitEvent(s_, d.ev, 0);
    if (e != hipSuccess) st_ = rsx::fail(RSX_ERR_HIP, "persistent gate wait: %s", hipGetErrorString(e));
  }
}

Gate::~Gate() {
  PerDevice &d = slot(device_);
  if (d.ev && st_ == RSX_OK && hipEventRecord(d.ev, s_) == hipSuccess) d.recorded = true;
  d.mu.unlock();
}

}  // namespace persistent
}  // namespace rsx
