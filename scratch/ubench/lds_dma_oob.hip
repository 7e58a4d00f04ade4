// What does an LDS-DMA (buffer_load ... lds) write into LDS for a lane whose address lies beyond the descriptor's num_records?
// Two product kernels depend on the answer and their comments contradicted each other:
//   conv_dma.hip      padding taps set the lane's offset out of range and rely on the DMA writing ZEROS into that lane's LDS bytes
//   attn_spatial.hip  said out-of-range key rows are NOT written (stale LDS stays) and zero-fills the stages once for that reason
// The stage is pre-filled with a sentinel; lanes >= 32 (dwordx4 case) / odd lanes (dword case) are out of range.
//   hipcc --offload-arch=gfx950 -O2 -o scratch/ubench/lds_dma_oob scratch/ubench/lds_dma_oob.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float *p, float *out, unsigned nrec) {
    __shared__ __attribute__((aligned(16))) float s[64 * 4 + 64];
    const int lane = threadIdx.x;
    for (int i = lane; i < 64 * 4 + 64; i += 64) s[i] = 777.f;
    __syncthreads();
    auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p), 0, (int)nrec, 0x00020000);
    // (a) 16 bytes per lane, lane-linear destination; lanes 32..63 read past num_records = 512 bytes
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)s, 16, (unsigned)(lane * 16), 0, 0, 0);
    // (b) 4 bytes per lane; odd lanes get an offset near 2^32 (how the GEMM epilogue masks columns)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(s + 256), 4, (lane & 1) ? 0xfffff000u : (unsigned)(lane * 4), 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 64 * 4 + 64; i += 64) out[i] = s[i];
}
int main() {
    float *p, *o;
    hipMalloc(&p, 1 << 16);
    hipMalloc(&o, (256 + 64) * 4);
    std::vector<float> h(1 << 14);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 1000.f + (float)i;
    hipMemcpy(p, h.data(), 1 << 16, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, p, o, 512u);
    hipDeviceSynchronize();
    std::vector<float> r(256 + 64);
    hipMemcpy(r.data(), o, r.size() * 4, hipMemcpyDeviceToHost);
    int in_ok = 0, oob_zero = 0, oob_stale = 0, oob_other = 0;
    for (int i = 0; i < 256; ++i) {
        if (i < 128) in_ok += r[i] == 1000.f + i;
        else if (r[i] == 0.f) ++oob_zero;
        else if (r[i] == 777.f) ++oob_stale;
        else ++oob_other;
    }
    printf("dwordx4 DMA, num_records = 512 B: in-range floats correct %d / 128; out-of-range floats: %d zero, %d stale (sentinel), %d other\n", in_ok, oob_zero, oob_stale, oob_other);
    in_ok = oob_zero = oob_stale = oob_other = 0;
    for (int l = 0; l < 64; ++l) {
        const float v = r[256 + l];
        if (!(l & 1)) in_ok += v == 1000.f + l;
        else if (v == 0.f) ++oob_zero;
        else if (v == 777.f) ++oob_stale;
        else ++oob_other;
    }
    printf("dword DMA, odd lanes at offset 0xfffff000: in-range correct %d / 32; out-of-range: %d zero, %d stale, %d other\n", in_ok, oob_zero, oob_stale, oob_other);
    return 0;
}
