// Prototype: fp32-MFMA GEMM C = A * W^T, 64x64x32 tile, operands staged by LDS-DMA (global_load_lds 16 B/lane),
// 3 LDS stages, counted vmcnt + raw s_barrier.  LDS image: unpadded 128-byte rows, 16-byte chunk c of row r stored at
// chunk position c ^ ((r >> 1) & 7) (swizzle applied on the per-lane SOURCE address and on the fragment read).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int BK = 32, BM = 64, BN = 64;
constexpr int STAGE_F = (BM + BN) * BK;  // floats per stage (16 KB)

template <int NST>
__global__ __launch_bounds__(256) void k_glds(const float *__restrict__ A, const float *__restrict__ W, float *__restrict__ C, int M, int N, int K) {
    __shared__ __attribute__((aligned(16))) float smem[NST * STAGE_F];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (N + BN - 1) / BN;
    int bid = blockIdx.x;
    { const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7, x = bid & 7, loc = bid >> 3; bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + loc; }
    const int tm = bid / tiles_n, tn = bid % tiles_n;
    // staging: wave w owns rows [16w, 16w+16) of A and of W; one DMA instruction = 8 rows x 128 B
    const int srow = lane >> 3, spos = lane & 7;  // row within the 8-row group, physical chunk position
    const float *ga[2], *gb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = 16 * wave + 8 * i + srow;                 // row within the tile
        const int c = spos ^ ((r >> 1) & 7);                      // logical chunk stored at this position
        int m = tm * BM + r; m = m < M ? m : M - 1;               // clamped rows produce values that are never stored
        int n = tn * BN + r; n = n < N ? n : N - 1;
        ga[i] = A + (size_t)m * K + c * 4;
        gb[i] = W + (size_t)n * K + c * 4;
    }
    auto issue = [&](int kt, int st) {
        float *sA = smem + st * STAGE_F, *sB = sA + BM * BK;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(ga[i] + kt * BK),
                                             (__attribute__((address_space(3))) void *)(sA + (16 * wave + 8 * i) * BK), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gb[i] + kt * BK),
                                             (__attribute__((address_space(3))) void *)(sB + (16 * wave + 8 * i) * BK), 16, 0, 0);
        }
    };
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int ra = wm * 32 + l31, rb = wn * 32 + l31;
    const int swa = (ra >> 1) & 7, swb = (rb >> 1) & 7;
    const int nkt = K / BK;
    constexpr int PF = NST - 1;  // tiles in flight ahead of the one being multiplied
    for (int t = 0; t < PF; ++t) if (t < nkt) issue(t, t);
    // tile 0 must have landed: allow the DMAs of the (up to PF-1) younger tiles to stay in flight
    {
        const int younger = (nkt - 1 < PF - 1 ? nkt - 1 : PF - 1);
        if (younger >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    for (int kt = 0; kt < nkt; ++kt) {
        const int st = kt % NST;
        if (kt + PF < nkt) issue(kt + PF, (kt + PF) % NST);  // that stage was last read in iteration kt-1 (barrier passed)
        const float *sA = smem + st * STAGE_F, *sB = sA + BM * BK;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int cq = 2 * q + lh;
            const f32x4 fa = *(const f32x4 *)&sA[ra * BK + ((cq ^ swa) << 2)];
            const f32x4 fb = *(const f32x4 *)&sB[rb * BK + ((cq ^ swb) << 2)];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[e], fb[e], acc, 0, 0, 0);
        }
        // before the next iteration reads tile kt+1: everything but the DMAs of the younger tiles must be complete
        {
            int younger = nkt - 1 - (kt + 1);
            younger = younger < 0 ? 0 : (younger > PF - 1 ? PF - 1 : younger);
            if (younger >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    const int n = tn * BN + wn * 32 + l31;
    for (int r = 0; r < 16; ++r) {
        const int m = tm * BM + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m < M && n < N) C[(size_t)m * N + n] = acc[r];
    }
}

template <int NST> void launch(dim3 grid, const float *A, const float *W, float *C, int M, int N, int K) { k_glds<NST><<<grid, 256>>>(A, W, C, M, N, K); }
int main(int argc, char **argv) {
  for (int nst = 2; nst <= 4; ++nst) {
    printf("== %d LDS stages\n", nst);
    auto L = [&](dim3 grid, const float *A, const float *W, float *C, int M, int N, int K) { if (nst == 2) launch<2>(grid, A, W, C, M, N, K); else if (nst == 3) launch<3>(grid, A, W, C, M, N, K); else launch<4>(grid, A, W, C, M, N, K); };
    // ---- correctness on a ragged shape
    {
        const int M = 300, N = 200, K = 96;
        std::vector<float> hA((size_t)M * K), hW((size_t)N * K), hC((size_t)M * N);
        srand(1);
        for (auto &v : hA) v = (rand() % 2001 - 1000) / 1000.f;
        for (auto &v : hW) v = (rand() % 2001 - 1000) / 1000.f;
        float *A, *W, *C; hipMalloc(&A, hA.size() * 4); hipMalloc(&W, hW.size() * 4); hipMalloc(&C, hC.size() * 4);
        hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice); hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice);
        hipMemset(C, 0xff, hC.size() * 4);
        if (nst == 2) k_glds<2><<<((M + 63) / 64) * ((N + 63) / 64), 256>>>(A, W, C, M, N, K);
        else if (nst == 3) k_glds<3><<<((M + 63) / 64) * ((N + 63) / 64), 256>>>(A, W, C, M, N, K);
        else k_glds<4><<<((M + 63) / 64) * ((N + 63) / 64), 256>>>(A, W, C, M, N, K);
        hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost);
        double worst = 0;
        for (int m = 0; m < M; ++m) for (int n = 0; n < N; ++n) {
            double s = 0; for (int k = 0; k < K; ++k) s += (double)hA[(size_t)m * K + k] * hW[(size_t)n * K + k];
            worst = fmax(worst, fabs(s - hC[(size_t)m * N + n]));
        }
        printf("correctness %dx%dx%d: max abs err %.3e %s\n", M, N, K, worst, worst < 1e-4 ? "OK" : "WRONG");
    }
    auto bench = [&](int M, int N, int K, const char *what) {
        float *A, *W, *C; hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&W, (size_t)N * K * 4); hipMalloc(&C, (size_t)M * N * 4);
        std::vector<float> h((size_t)M * K); for (auto &v : h) v = (rand() % 2001 - 1000) / 1000.f; hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        h.resize((size_t)N * K); for (auto &v : h) v = (rand() % 2001 - 1000) / 1000.f; hipMemcpy(W, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        dim3 grid(((M + 63) / 64) * ((N + 63) / 64));
        L(grid, A, W, C, M, N, K); hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0); for (int i = 0; i < 10; ++i) L(grid, A, W, C, M, N, K); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
        printf("%-14s M=%6d N=%5d K=%5d: %8.1f us  %6.1f TF\n", what, M, N, K, ms * 1e3, 2.0 * M * N * K / (ms * 1e-3) / 1e12);
        hipFree(A); hipFree(W); hipFree(C);
    };
    bench(10960, 1152, 384, "qkv T8"); bench(10960, 1536, 384, "fc1 T8"); bench(10960, 384, 384, "proj T8"); bench(10960, 384, 1536, "fc2 T8");
    bench(43840, 1152, 384, "qkv T32"); bench(8192, 8192, 1024, "8k8k1k"); bench(4096, 4096, 4096, "4096^3");
  }
    return 0;
}
