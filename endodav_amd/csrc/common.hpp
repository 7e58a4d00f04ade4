// Shared device/host helpers for libendodav_hip (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
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
