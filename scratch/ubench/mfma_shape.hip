// Does the chip hold a higher clock on one f32 MFMA shape than on the other?  (MI355X_MICROARCH.md "DVFS give-back" item 7 found 1.12-1.15x
// for bf16 16x16x32 over 32x32x16 at equal cycles per FLOP.)  Bare loops on random operands in registers, the same FLOP per wave, every CU busy;
// wall time and the in-kernel clock (s_memtime / s_memrealtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int SHAPE>  // 0: 32x32x2 (one 32x32 accumulator), 1: 16x16x4 (2x2 blocks = the same 32x32 output tile)
__global__ __launch_bounds__(256) void k(const float *__restrict__ in, float *out, unsigned long long *clk, int iters) {
    float a[8], b[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        a[j] = in[(blockIdx.x * 256 + threadIdx.x) * 16 + j];
        b[j] = in[(blockIdx.x * 256 + threadIdx.x) * 16 + 8 + j];
    }
    f32x16 acc32 = {};
    f32x4 acc16[4] = {};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if (SHAPE == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc32 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], acc32, 0, 0, 0);  // 8 x 4096 FLOP
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {  // 16 x 2048 FLOP
                acc16[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2 * j], b[2 * j], acc16[0], 0, 0, 0);
                acc16[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2 * j], b[2 * j + 1], acc16[1], 0, 0, 0);
                acc16[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2 * j + 1], b[2 * j], acc16[2], 0, 0, 0);
                acc16[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2 * j + 1], b[2 * j + 1], acc16[3], 0, 0, 0);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc32[r];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 4; ++r) s += acc16[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int SHAPE>
void run(int per_cu, const float *in, float *out, unsigned long long *clk, int iters) {
    const int grid = 256 * per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) k<SHAPE><<<grid, 256>>>(in, out, clk, iters);  // warm: clocks settle under load
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<SHAPE><<<grid, 256>>>(in, out, clk, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * grid);
    hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> ghz;
    for (int i = 0; i < grid; ++i) ghz.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 0.1);  // memrealtime ticks at 100 MHz
    std::sort(ghz.begin(), ghz.end());
    const double flop = (double)grid * 4 * iters * 8 * 4096.0;
    printf("%s  waves/SIMD %d : %7.1f TF/s   in-kernel clock median %.3f GHz  -> %.1f %% of the 64 FLOP/clk/SIMD rate at that clock\n", SHAPE ? "16x16x4" : "32x32x2", per_cu,
           flop / (ms * 1e-3) / 1e12, ghz[grid / 2], 100.0 * flop / (ms * 1e-3) / (1024.0 * 64.0 * ghz[grid / 2] * 1e9));
}

int main() {
    const int n = 256 * 4 * 256 * 16;
    std::vector<float> h(n);
    srand(1);
    for (auto &v : h) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    float *in, *out; unsigned long long *clk;
    hipMalloc(&in, n * 4); hipMalloc(&out, 256 * 4 * 256 * 4); hipMalloc(&clk, 256 * 4 * 2 * 8);
    hipMemcpy(in, h.data(), n * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep)
        for (int w : {1, 2, 4}) { run<0>(w, in, out, clk, 40000 / w); run<1>(w, in, out, clk, 40000 / w); }
    return 0;
}
