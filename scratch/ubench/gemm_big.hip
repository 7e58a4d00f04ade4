// Prototype: LDS-DMA fp32-MFMA GEMM with larger workgroup tiles (4 waves, each 32*FM x 32*FN), plain grid, 2 LDS stages.
// Question: does a fatter tile (more MFMAs per staged byte and per barrier) lift the ~120 TF/s plateau of the 64x64 tile?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int BK = 32;

template <int FM, int FN, int OCC>
__global__ __launch_bounds__(256, OCC) void k_big(const float *__restrict__ A, const float *__restrict__ W, float *__restrict__ C, int M, int N, int K) {
    constexpr int BM = 64 * FM, BN = 64 * FN;          // 2 x 2 waves
    constexpr int STAGE_F = (BM + BN) * BK;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (N + BN - 1) / BN;
    int bid = blockIdx.x;
    { const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7, x = bid & 7, loc = bid >> 3; bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + loc; }
    const int tm = bid / tiles_n, tn = bid % tiles_n;
    const int srow = lane >> 3, spos = lane & 7;
    constexpr int IA = BM / 32, IB = BN / 32;           // DMA instructions per wave per k-tile (8 rows each)
    const float *ga[IA], *gb[IB];
#pragma unroll
    for (int i = 0; i < IA; ++i) {
        const int r = (BM / 4) * wave + 8 * i + srow;
        const int c = spos ^ ((r >> 1) & 7);
        int m = tm * BM + r; m = m < M ? m : M - 1;
        ga[i] = A + (size_t)m * K + c * 4;
    }
#pragma unroll
    for (int i = 0; i < IB; ++i) {
        const int r = (BN / 4) * wave + 8 * i + srow;
        const int c = spos ^ ((r >> 1) & 7);
        int n = tn * BN + r; n = n < N ? n : N - 1;
        gb[i] = W + (size_t)n * K + c * 4;
    }
    auto issue = [&](int kt, int st) {
        float *sA = smem + st * STAGE_F, *sB = sA + BM * BK;
#pragma unroll
        for (int i = 0; i < IA; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(ga[i] + kt * BK),
                                             (__attribute__((address_space(3))) void *)(sA + ((BM / 4) * wave + 8 * i) * BK), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < IB; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gb[i] + kt * BK),
                                             (__attribute__((address_space(3))) void *)(sB + ((BN / 4) * wave + 8 * i) * BK), 16, 0, 0);
    };
    f32x16 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    int ra[FM], rb[FN], swa[FM], swb[FN];
#pragma unroll
    for (int i = 0; i < FM; ++i) { ra[i] = wm * 32 * FM + i * 32 + l31; swa[i] = (ra[i] >> 1) & 7; }
#pragma unroll
    for (int j = 0; j < FN; ++j) { rb[j] = wn * 32 * FN + j * 32 + l31; swb[j] = (rb[j] >> 1) & 7; }
    const int nkt = K / BK;
    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int kt = 0; kt < nkt; ++kt) {
        const int st = kt & 1;
        if (kt + 1 < nkt) issue(kt + 1, st ^ 1);
        const float *sA = smem + st * STAGE_F, *sB = sA + BM * BK;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int cq = 2 * q + lh;
            f32x4 fa[FM], fb[FN];
#pragma unroll
            for (int i = 0; i < FM; ++i) fa[i] = *(const f32x4 *)&sA[ra[i] * BK + ((cq ^ swa[i]) << 2)];
#pragma unroll
            for (int j = 0; j < FN; ++j) fb[j] = *(const f32x4 *)&sB[rb[j] * BK + ((cq ^ swb[j]) << 2)];
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][e], fb[j][e], acc[i][j], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int n = tn * BN + wn * 32 * FN + j * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = tm * BM + wm * 32 * FM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m < M && n < N) C[(size_t)m * N + n] = acc[i][j][r];
            }
        }
}

template <int FM, int FN, int OCC>
void run(const char *name) {
    constexpr int BM = 64 * FM, BN = 64 * FN;
    const size_t lds = 2 * (size_t)(BM + BN) * BK * 4;
    hipFuncSetAttribute((const void *)k_big<FM, FN, OCC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int occ = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_big<FM, FN, OCC>, 256, lds);
    printf("== tile %dx%d (%s), LDS %zu KB, %d workgroups per CU\n", BM, BN, name, lds / 1024, occ);
    {
        const int M = 300, N = 200, K = 96;
        std::vector<float> hA((size_t)M * K), hW((size_t)N * K), hC((size_t)M * N);
        srand(1);
        for (auto &v : hA) v = (rand() % 2001 - 1000) / 1000.f;
        for (auto &v : hW) v = (rand() % 2001 - 1000) / 1000.f;
        float *A, *W, *C; hipMalloc(&A, hA.size() * 4); hipMalloc(&W, hW.size() * 4); hipMalloc(&C, hC.size() * 4);
        hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice); hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice);
        hipMemset(C, 0xff, hC.size() * 4);
        k_big<FM, FN, OCC><<<((M + BM - 1) / BM) * ((N + BN - 1) / BN), 256, lds>>>(A, W, C, M, N, K);
        hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost);
        double worst = 0;
        for (int m = 0; m < M; ++m) for (int n = 0; n < N; ++n) {
            double s = 0; for (int k = 0; k < K; ++k) s += (double)hA[(size_t)m * K + k] * hW[(size_t)n * K + k];
            worst = fmax(worst, fabs(s - hC[(size_t)m * N + n]));
        }
        printf("correctness %dx%dx%d: max abs err %.3e %s\n", M, N, K, worst, worst < 1e-4 ? "OK" : "WRONG");
        hipFree(A); hipFree(W); hipFree(C);
    }
    auto bench = [&](int M, int N, int K, const char *what) {
        float *A, *W, *C; hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&W, (size_t)N * K * 4); hipMalloc(&C, (size_t)M * N * 4);
        const char *mode = getenv("DATA");  // unset: random; "zero"; "one"
        auto fill = [&](float &v) { v = !mode ? (rand() % 2001 - 1000) / 1000.f : (mode[0] == 'z' ? 0.f : 1.f); };
        std::vector<float> h((size_t)M * K); for (auto &v : h) fill(v); hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        h.resize((size_t)N * K); for (auto &v : h) fill(v); hipMemcpy(W, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        dim3 grid(((M + BM - 1) / BM) * ((N + BN - 1) / BN));
        k_big<FM, FN, OCC><<<grid, 256, lds>>>(A, W, C, M, N, K); hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0); for (int i = 0; i < 10; ++i) k_big<FM, FN, OCC><<<grid, 256, lds>>>(A, W, C, M, N, K); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
        printf("%-10s M=%6d N=%5d K=%5d  grid %6u: %8.1f us  %6.1f TF\n", what, M, N, K, grid.x, ms * 1e3, 2.0 * M * N * K / (ms * 1e-3) / 1e12);
        hipFree(A); hipFree(W); hipFree(C);
    };
    bench(10960, 1152, 384, "qkv T8"); bench(10960, 384, 1536, "fc2 T8"); bench(43840, 1152, 384, "qkv T32");
    bench(8192, 8192, 1024, "8k8k1k"); bench(4096, 4096, 4096, "4096^3"); bench(16384, 4096, 1024, "16k4k1k");
}

int main(int argc, char **argv) {
    if (argc > 1) {  // quick mode: two tile shapes only
        run<1, 1, 4>("1 acc/wave");
        run<4, 4, 1>("16 acc/wave");
        return 0;
    }
    run<1, 1, 4>("1 acc/wave");
    run<2, 1, 3>("2 acc/wave");
    run<2, 2, 2>("4 acc/wave");
    run<4, 2, 1>("8 acc/wave");
    run<4, 4, 1>("16 acc/wave");
    return 0;
}
