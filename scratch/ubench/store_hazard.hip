// A store-data hazard on gfx950 that the compiler's hazard recognizer does not cover (found in round 3 by a wrong fc1 + GELU tile, profiles/r03_notes.txt):
//     buffer_store_dwordx4 v[8:11], v3, s[16:19], s15 offen     ; 128-bit store, scalar offset in an SGPR
//     v_mul_f32 v8, ...                                          ; VALU write of the store's first data register, next instruction
// LLVM's GCNHazardRecognizer::createsVALUHazard requires a wait state between a >64-bit MUBUF store and a VALU write of its data ONLY when soffset is
// not a register.  This reproducer issues exactly that pair (inline asm, fixed registers) from many workgroups per CU and counts how many stored
// words 0 carry the overwriting value instead of the data; variants: 0 = SGPR soffset, no wait state; 1 = SGPR soffset + s_nop 3; 2 = soffset 0 (immediate
// form), no wait state.
//   hipcc --offload-arch=gfx950 -O2 -o scratch/ubench/store_hazard scratch/ubench/store_hazard.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int V>
__global__ __launch_bounds__(256) void k(float *out, int iters, unsigned nbytes) {
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)nbytes, 0x00020000);
    const unsigned lane_off = threadIdx.x * 16u;
    for (int it = 0; it < iters; ++it) {
        const unsigned row = (blockIdx.x * (unsigned)iters + it) * 4096u;  // 256 lanes x 16 B per (block, iteration)
        const float base = (float)(it + 1);
        if (V == 2) {
            const unsigned vo = row + lane_off;
            asm volatile(
                "v_mov_b32 v8, %0\n\tv_mov_b32 v9, %0\n\tv_mov_b32 v10, %0\n\tv_mov_b32 v11, %0\n\t"
                "s_nop 4\n\t"
                "buffer_store_dwordx4 v[8:11], %1, %2, 0 offen\n\t"
                "v_mov_b32 v8, -1.0\n\t"
                :: "v"(base), "v"(vo), "s"(rs) : "v8", "v9", "v10", "v11", "memory");
        } else if (V == 1) {
            asm volatile(
                "v_mov_b32 v8, %0\n\tv_mov_b32 v9, %0\n\tv_mov_b32 v10, %0\n\tv_mov_b32 v11, %0\n\t"
                "s_nop 4\n\t"
                "buffer_store_dwordx4 v[8:11], %1, %2, %3 offen\n\t"
                "s_nop 3\n\t"
                "v_mov_b32 v8, -1.0\n\t"
                :: "v"(base), "v"(lane_off), "s"(rs), "s"(row) : "v8", "v9", "v10", "v11", "memory");
        } else {
            asm volatile(
                "v_mov_b32 v8, %0\n\tv_mov_b32 v9, %0\n\tv_mov_b32 v10, %0\n\tv_mov_b32 v11, %0\n\t"
                "s_nop 4\n\t"
                "buffer_store_dwordx4 v[8:11], %1, %2, %3 offen\n\t"
                "v_mov_b32 v8, -1.0\n\t"
                :: "v"(base), "v"(lane_off), "s"(rs), "s"(row) : "v8", "v9", "v10", "v11", "memory");
        }
    }
}
int main() {
    const int blocks = 256 * 8, iters = 64;
    const size_t n = (size_t)blocks * iters * 1024;  // floats
    float *d;
    hipMalloc(&d, n * 4);
    std::vector<float> h(n);
    const char *names[3] = {"SGPR soffset, VALU write of v8 right after the store", "SGPR soffset, s_nop 3 in between", "soffset = 0 (immediate form), no wait state"};
    for (int v = 0; v < 3; ++v) {
        long long bad_total = 0, bad_lane[4] = {0, 0, 0, 0};
        for (int rep = 0; rep < 5; ++rep) {
            hipMemset(d, 0, n * 4);
            if (v == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, iters, (unsigned)(n * 4));
            if (v == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, iters, (unsigned)(n * 4));
            if (v == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, d, iters, (unsigned)(n * 4));
            hipDeviceSynchronize();
            hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
            for (size_t i = 0; i < n; i += 4)
                if (h[i] == -1.0f) {
                    ++bad_total;
                    ++bad_lane[((i / 4) % 64) / 4 % 4];
                }
        }
        printf("%-60s: %lld of %lld stored word-0 values overwritten (by position of the lane within its group of 16: %lld %lld %lld %lld)\n", names[v], bad_total,
               (long long)(5 * n / 4), bad_lane[0], bad_lane[1], bad_lane[2], bad_lane[3]);
    }
    return 0;
}
