// devscope.hpp -- what makes a handle belong to a DEVICE rather than to the process (round 6).
// The design is one process per GPU, but nothing in the C-ABI forbids a host that opens handles on several GPUs of one process:
//   * DevScope: every entry point that touches the device runs with the handle's device current and puts the caller's back;
//   * PerDeviceOnce: hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-device setting -- done once per (kernel, device), under a
//     lock; the bookkeeping is a pure function of the device id so that it can be tested without a second GPU
//     (mcrx_hip_selftest_device_table, tests/test_boundary.py).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <mutex>
#include <stdint.h>

namespace mcrx {

struct DevScope {
    int prev = -1; bool switched = false;
    explicit DevScope(int dev)
    {
        if (dev < 0) return;
        if (hipGetDevice(&prev) == hipSuccess && prev != dev && hipSetDevice(dev) == hipSuccess) switched = true;
    }
    ~DevScope() { if (switched) (void)hipSetDevice(prev); }
    DevScope(const DevScope &) = delete;
    DevScope &operator=(const DevScope &) = delete;
};
inline int current_device() { int d = 0; return hipGetDevice(&d) == hipSuccess ? d : -1; }

// up to 256 devices per process; first(dev) is true exactly once per device, for the caller that then does the work under the lock
class PerDeviceOnce {
    std::atomic<uint64_t> done_[4];
    std::mutex mu_;
public:
    PerDeviceOnce() { for (auto &w : done_) w.store(0); }
    bool is_done(int dev) const { return dev >= 0 && dev < 256 && ((done_[dev >> 6].load(std::memory_order_acquire) >> (dev & 63)) & 1u); }
    template <class F> hipError_t run(int dev, F &&work)
    {
        if (dev < 0 || dev >= 256) return work();              // (outside the table: every time)
        if (is_done(dev)) return hipSuccess;
        std::lock_guard<std::mutex> lk(mu_);
        if (is_done(dev)) return hipSuccess;
        const hipError_t e = work();
        if (e == hipSuccess) done_[dev >> 6].fetch_or(1ull << (dev & 63), std::memory_order_release);
        return e;
    }
};
// raise a kernel's dynamic LDS limit on the current device, once per device
inline hipError_t raise_lds_limit(const void *fn, size_t lds, PerDeviceOnce &once)
{
    return once.run(current_device(), [&]() { return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); });
}

}  // namespace mcrx
