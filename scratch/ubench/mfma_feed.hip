// What keeps the f32 MFMA pipe only ~80 % busy in the attention kernel (profiles/r02_attn_pmc.txt: clock 2.34 GHz, no LDS conflicts, waves
// issue-stalled 75-80 % of their cycles)?  The attention tile loop rebuilt from its parts, one added at a time, 2 waves per SIMD (the product's
// occupancy; 64 KB of LDS per workgroup pins exactly 2 workgroups per CU) and 1 wave per SIMD:
//   0  bare: 128 MFMAs per tile on two accumulators, operands in registers
//   1  + operands re-read from LDS (one ds_read_b128 per operand per 4 MFMAs, like the K fragments)
//   2  + one s_barrier per tile
//   3  + the second half of each tile takes its B operand from the first half's accumulator registers (PV reads the scores)
//   4  + 160 VALU instructions per tile (max / sub / exp2 / mul) as ONE burst after the MFMAs (the round-1 kernel's softmax)
//   5  the same VALU work cut into slices behind the MFMAs of the second half (the pipelined kernel)
//   6  = 5 + b32 LDS reads feeding the second half's A operand (the V reads)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int V, int WPS>
__global__ __launch_bounds__(256, WPS) void k(const float *__restrict__ in, float *out, unsigned long long *clk, int tiles) {
    extern __shared__ __attribute__((aligned(16))) float lds[];  // 64 KB (2 per CU) or 128 KB (1 per CU): only the first 32 KB are used
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
    for (int i = tid; i < 8192; i += 256) lds[i] = in[(blockIdx.x % 64) * 8192 + i];
    __syncthreads();
    f32x4 qf[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) qf[q] = *reinterpret_cast<const f32x4 *>(&in[tid * 32 + 4 * q]);
    f32x16 s0 = {}, s1 = {}, o0 = {}, o1 = {};
    float m_run = -1e30f, l_run = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int t = 0; t < tiles; ++t) {
        if (V >= 2) __builtin_amdgcn_s_barrier();
        // ---- first half: 64 MFMAs, A from LDS (or registers), B = qf
#pragma unroll
        for (int r = 0; r < 16; ++r) s0[r] = s1[r] = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            f32x4 ka, kb;
            if (V >= 1) {
                const int cq = 2 * q + lh;
                ka = *reinterpret_cast<const f32x4 *>(&lds[l31 * 64 + ((cq ^ (l31 & 15)) << 2)]);
                kb = *reinterpret_cast<const f32x4 *>(&lds[(32 + l31) * 64 + ((cq ^ (l31 & 15)) << 2)]);
            } else {
                ka = qf[(q + 1) & 7];
                kb = qf[(q + 3) & 7];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s0 = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[e], qf[q][e], s0, 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f32_32x32x2f32(kb[e], qf[q][e], s1, 0, 0, 0);
            }
        }
        // ---- softmax-like VALU work on s0 / s1 (V >= 4), as one burst (4) or in slices behind the second half's MFMAs (5, 6)
        float mx = s0[0], alpha = 1.f, m_new = 0.f;
        auto slice = [&](int kk) {
            if (kk < 8) mx = fmaxf(fmaxf(mx, fmaxf(s0[2 * kk], s0[2 * kk + 1])), fmaxf(s1[2 * kk], s1[2 * kk + 1]));
            else if (kk == 8) mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            else if (kk == 9) { m_new = fmaxf(m_run, mx); alpha = __builtin_amdgcn_exp2f(m_run - m_new); m_run = m_new; }
            else if (kk < 26) s0[kk - 10] = __builtin_amdgcn_exp2f(s0[kk - 10] * 1e-3f - m_new * 1e-3f);
            else if (kk < 42) s1[kk - 26] = __builtin_amdgcn_exp2f(s1[kk - 26] * 1e-3f - m_new * 1e-3f);
            else if (kk < 58) l_run += s0[kk - 42] + s1[kk - 42];
        };
        f32x16 p0 = s0, p1 = s1;  // what the second half multiplies by: the first half's results (V >= 3), else a register operand
        if (V == 4) {
#pragma unroll
            for (int kk = 0; kk < 58; ++kk) slice(kk);
            p0 = s0; p1 = s1;
        }
        // ---- second half: 64 MFMAs into o0 / o1
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v0, v1, v2, v3;
            if (V >= 6) {
                const int key = (r & 3) + 8 * (r >> 2) + 4 * lh, sw = key & 15;
                const int c0 = (((l31 >> 2) ^ sw) << 2) + (l31 & 3), c1 = ((((l31 >> 2) + 8) ^ sw) << 2) + (l31 & 3);
                v0 = lds[key * 64 + c0]; v1 = lds[key * 64 + c1]; v2 = lds[(32 + key) * 64 + c0]; v3 = lds[(32 + key) * 64 + c1];
            } else {
                v0 = qf[r & 7][0]; v1 = qf[r & 7][1]; v2 = qf[r & 7][2]; v3 = qf[r & 7][3];
            }
            const float b0 = V >= 3 ? p0[r] : qf[(r + 2) & 7][r & 3], b1 = V >= 3 ? p1[r] : qf[(r + 5) & 7][r & 3];
            o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v0, b0, o0, 0, 0, 0);
            if (V >= 5) { slice(4 * r); __builtin_amdgcn_sched_barrier(0); }
            o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v1, b0, o1, 0, 0, 0);
            if (V >= 5) { slice(4 * r + 1); __builtin_amdgcn_sched_barrier(0); }
            o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v2, b1, o0, 0, 0, 0);
            if (V >= 5) { slice(4 * r + 2); __builtin_amdgcn_sched_barrier(0); }
            o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v3, b1, o1, 0, 0, 0);
            if (V >= 5) { slice(4 * r + 3); __builtin_amdgcn_sched_barrier(0); }
        }
        if (V >= 4) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = l_run + m_run;
    for (int r = 0; r < 16; ++r) s += o0[r] + o1[r] + s0[r] + s1[r];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int V, int WPS>
void run(const float *in, float *out, unsigned long long *clk, int tiles) {
    const int grid = 256 * WPS;
    const size_t lds = WPS == 2 ? 65536 : 131072;
    hipFuncSetAttribute((const void *)k<V, WPS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) k<V, WPS><<<grid, 256, lds>>>(in, out, clk, tiles);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<V, WPS><<<grid, 256, lds>>>(in, out, clk, tiles);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * grid);
    hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> ghz, cyc;
    for (int i = 0; i < grid; ++i) { ghz.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 0.1); cyc.push_back((double)h[2 * i] / tiles); }
    std::sort(ghz.begin(), ghz.end()); std::sort(cyc.begin(), cyc.end());
    const double flop = (double)grid * 4 * tiles * 128 * 4096.0;
    printf("variant %d, %d wave(s)/SIMD : %6.1f TF/s  clock %.3f GHz  median %6.0f cycles per tile per workgroup (128 MFMAs = 8192 x %d waves = %d)  pipe %.1f %%\n", V, WPS,
           flop / (ms * 1e-3) / 1e12, ghz[grid / 2], cyc[grid / 2], WPS, 8192 * WPS, 100.0 * 8192 * WPS / cyc[grid / 2]);
}

int main() {
    const int n = 64 * 8192 + 256 * 32 + 1024;
    std::vector<float> h(n);
    srand(1);
    for (auto &v : h) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    float *in, *out; unsigned long long *clk;
    hipMalloc(&in, n * 4); hipMalloc(&out, 512 * 256 * 4); hipMalloc(&clk, 512 * 2 * 8);
    hipMemcpy(in, h.data(), n * 4, hipMemcpyHostToDevice);
    const int tiles = 400;
    run<0, 2>(in, out, clk, tiles); run<1, 2>(in, out, clk, tiles); run<2, 2>(in, out, clk, tiles); run<3, 2>(in, out, clk, tiles);
    run<4, 2>(in, out, clk, tiles); run<5, 2>(in, out, clk, tiles); run<6, 2>(in, out, clk, tiles);
    run<0, 1>(in, out, clk, tiles); run<1, 1>(in, out, clk, tiles); run<2, 1>(in, out, clk, tiles); run<3, 1>(in, out, clk, tiles);
    run<4, 1>(in, out, clk, tiles); run<5, 1>(in, out, clk, tiles); run<6, 1>(in, out, clk, tiles);
    return 0;
}
