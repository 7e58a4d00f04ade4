// Which part of a raw buffer address does the hardware range-check on gfx950?  (decides how the GEMM epilogue masks edge tiles)
//   case 0: voffset in range, soffset pushes the address past num_records     -> dropped iff soffset takes part in the check
//   case 1: voffset past num_records, soffset 0                               -> must be dropped
//   case 2: voffset = 0xffffff00, soffset = 0x200 (sum wraps to 0x100)        -> dropped iff the sum is not taken modulo 2^32
//   case 3: voffset in range, soffset in range, sum in range                  -> must be written
//   case 4: voffset + soffset in range, but voffset alone past num_records    -> (cannot happen with unsigned offsets; skipped)
//   loads: the same five addresses, out-of-range loads must return 0
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(float *p, float *out, unsigned nrec) {
    auto rs = __builtin_amdgcn_make_buffer_rsrc(p, 0, (int)nrec, 0x00020000);
    const int lane = threadIdx.x;
    if (lane == 0) {
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(100.f), rs, 0, 2048, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(101.f), rs, 2052, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(102.f), rs, (int)0xffffff00u, 0x200, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(103.f), rs, 16, 32, 0);
        out[0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, 0, 2048, 0));
        out[1] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, 2052, 0, 0));
        out[2] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)0xffffff00u, 0x200, 0));
        out[3] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, 20, 32, 0));
    }
}
int main() {
    float *p, *o;
    hipMalloc(&p, 1 << 16);
    hipMalloc(&o, 64);
    std::vector<float> h(1 << 14);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)i;
    hipMemcpy(p, h.data(), 1 << 16, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, p, o, 1024u);
    hipDeviceSynchronize();
    std::vector<float> r(1 << 14);
    float out[4];
    hipMemcpy(r.data(), p, 1 << 16, hipMemcpyDeviceToHost);
    hipMemcpy(out, o, 16, hipMemcpyDeviceToHost);
    printf("num_records = 1024 bytes\n");
    printf("store voffset 0 + soffset 2048        : p[512] = %g (%s)\n", r[512], r[512] == 100.f ? "WRITTEN: soffset is NOT range-checked" : "dropped: soffset is range-checked");
    printf("store voffset 2052 + soffset 0        : p[513] = %g (%s)\n", r[513], r[513] == 101.f ? "WRITTEN (?!)" : "dropped");
    printf("store voffset 0xffffff00 + soffset 512: p[64]  = %g (%s)\n", r[64], r[64] == 102.f ? "WRITTEN: the sum wraps" : "dropped: no wrap");
    printf("store voffset 16 + soffset 32         : p[12]  = %g (%s)\n", r[12], r[12] == 103.f ? "written" : "NOT written (?!)");
    printf("loads: %g %g %g %g   (in-range value of the last: 13)\n", out[0], out[1], out[2], out[3]);
    return 0;
}
