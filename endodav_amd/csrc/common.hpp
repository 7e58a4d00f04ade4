// Shared device/host helpers for libendodav_hip (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <string>

namespace edv {

// thread-local error text behind edv_last_error()
void set_error(const std::string &msg);
const char *get_error();

#define EDV_CHECK(cond, msg)                                                     \
    do {                                                                         \
        if (!(cond)) {                                                           \
            ::edv::set_error(std::string(__func__) + ": " + (msg));              \
            return 1;                                                            \
        }                                                                        \
    } while (0)

#define EDV_HIP(expr)                                                            \
    do {                                                                         \
        hipError_t e_ = (expr);                                                  \
        if (e_ != hipSuccess) {                                                  \
            ::edv::set_error(std::string(__func__) + ": " #expr ": " + hipGetErrorString(e_)); \
            return 1;                                                            \
        }                                                                        \
    } while (0)

#define EDV_TRY(expr)                                                            \
    do {                                                                         \
        int r_ = (expr);                                                         \
        if (r_ != 0) return r_;                                                  \
    } while (0)

// after a kernel launch
#define EDV_LAUNCH_OK() EDV_HIP(hipGetLastError())

constexpr int WAVE = 64;

// Kernel timing for edv_profile_enable (round 3).  While the engine has a Bracket open on this thread, every launch carries the bracket's event
// pair INSIDE its dispatch (hipExtLaunchKernelGGL: the start event is stamped when the kernel starts, the stop event when it ends), so a bracket
// measures the kernel(s) alone -- the same interval rocprofv3's kernel trace reports -- instead of "the in-order gap before the launch + the kernel"
// that hipEventRecord before / after the launch measures (12 us against 8 us for a LayerNorm launch in round 2).  A bracket around several launches
// spans from the first kernel's start to the last one's end.
struct LaunchTimer {
    hipEvent_t start = nullptr, stop = nullptr;
    bool started = false;
};
extern thread_local LaunchTimer *g_launch_timer;
#define EDV_LAUNCH(kernel, grid, block, shmem, stream, ...)                                                                                   \
    do {                                                                                                                                      \
        ::edv::LaunchTimer *lt_ = ::edv::g_launch_timer;                                                                                      \
        if (lt_) {                                                                                                                            \
            hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, lt_->started ? nullptr : lt_->start, lt_->stop, 0, __VA_ARGS__);        \
            lt_->started = true;                                                                                                              \
        } else {                                                                                                                              \
            hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);                                                              \
        }                                                                                                                                     \
    } while (0)

// Launch geometry that depends on the DEVICE (resident workgroups of a persistent kernel = CUs x occupancy) is cached per HIP device, not per
// process: nn.DataParallel drives several devices from one process, and a partitioned or mixed node may give them different CU counts.
// `query` runs on the current device; 0 = the query failed (not cached).
constexpr int EDV_MAX_DEVICES = 64;
struct DeviceSlotCache {
    std::atomic<int> v[EDV_MAX_DEVICES];  // static storage: zero-initialised
    template <class Q>
    int get(Q query) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= EDV_MAX_DEVICES) return query();
        int s = v[dev].load(std::memory_order_relaxed);
        if (s == 0) {
            s = query();
            v[dev].store(s, std::memory_order_relaxed);
        }
        return s;
    }
};

// Row remap: logical row m of a [frames * period] matrix lives at physical row
//   (m / period) * stride + offset + inner * (m % period).
// period == 0 means identity.  Used to skip the cls row of each frame's token block, to
// broadcast the position table over frames (stride 0), to broadcast one row per frame over
// that frame's rows (inner 0) and to address sub-ranges in place.
struct RowMap {
    int period, stride, offset;
    int inner = 1;
    __host__ __device__ inline long long operator()(long long m) const {
        if (period == 0) return m;
        long long f = m / period;
        return f * stride + offset + inner * (m - f * period);
    }
};
inline RowMap identity_map() { return RowMap{0, 0, 0}; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// exact-erf GELU, the nn.GELU() / F.gelu default used everywhere in the reference
// (layers/block.py:58, motion_module/attention.py:378)
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

inline int ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }

}  // namespace edv
