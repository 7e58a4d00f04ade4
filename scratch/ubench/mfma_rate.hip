// Bare MFMA issue-rate microbenchmark: waves per SIMD x independent accumulators, fp32 32x32x2 and bf16 32x32x16.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

template <int NACC, bool BF>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 2e-3f;
    bf16x8 ab, bb;
    for (int j = 0; j < 8; ++j) { ab[j] = (__bf16)(a + j); bb[j] = (__bf16)(b - j); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                if (BF) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[i], 0, 0, 0);
                else acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
            }
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, bool BF>
void run(int blocks_per_cu, float *out) {
    const int iters = 2000 / NACC;
    const int grid = 256 * blocks_per_cu;  // 256-thread blocks: 4 waves = 1 per SIMD each
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC, BF><<<grid, 256>>>(out, 10, 1.f, 2.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NACC, BF><<<grid, 256>>>(out, iters, 1.f, 2.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = (double)grid * 4 * iters * 8 * NACC;
    const double flop = mfmas * (BF ? 32768.0 : 4096.0);
    const double cyc_per_mfma_per_simd = (ms * 1e-3 * 2.4e9) / (mfmas / 1024.0);
    printf("%s waves/SIMD=%d acc=%d : %7.1f TF  (%.1f cycles per MFMA per SIMD at 2.4 GHz)\n", BF ? "bf16 32x32x16" : "f32  32x32x2 ", blocks_per_cu, NACC,
           flop / (ms * 1e-3) / 1e12, cyc_per_mfma_per_simd);
}

int main() {
    float *out; hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    for (int w : {1, 2, 4}) { run<1, false>(w, out); run<2, false>(w, out); run<4, false>(w, out); }
    for (int w : {1, 2, 4}) { run<1, true>(w, out); run<2, true>(w, out); run<4, true>(w, out); }
    return 0;
}
