// launch_util.h -- host-side helpers shared by the kernel launchers.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>

namespace nyxhip {

// hipFuncAttributeMaxDynamicSharedMemorySize (the opt-in to more than 64 KiB of dynamic LDS) is a per-DEVICE property of a
// kernel.  A process may hold contexts on several GPUs (nyxhip_init takes any device index; nyxhip_featurize_tiles_sharded
// drives one context per GPU from its own thread), so every launcher opts its kernels in once per device: run() calls `f`
// the first time it sees the calling thread's current device.  Two threads racing on the same device both set the attribute
// (idempotent); the bit is published only after a successful call.
struct DeviceOnce {
    std::atomic<unsigned long long> done{0};
    template <typename F>
    int run(F&& f)
    {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess)
            return (int)e;
        const unsigned long long bit = 1ull << (dev & 63);
        if (done.load(std::memory_order_acquire) & bit)
            return 0;
        const int rc = f();
        if (rc == 0)
            done.fetch_or(bit, std::memory_order_release);
        return rc;
    }
};

} // namespace nyxhip
