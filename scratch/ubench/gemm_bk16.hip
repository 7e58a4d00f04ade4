// Prototype: 64x64 LDS-DMA GEMM with 16-wide k-tiles and 3 LDS stages (24 KB per workgroup -> 5-6 workgroups per CU instead of
// the 4 the 2 x 16 KB version really gets), counted vmcnt.  PAD pads the LDS allocation to steer the occupancy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int BK = 16, BM = 64, BN = 64, NST = 3;
constexpr int STAGE_F = (BM + BN) * BK;  // 2048 floats = 8 KB

__global__ __launch_bounds__(256) void k16(const float *__restrict__ A, const float *__restrict__ W, float *__restrict__ C, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (N + BN - 1) / BN;
    int bid = blockIdx.x;
    { const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7, x = bid & 7, loc = bid >> 3; bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + loc; }
    const int tm = bid / tiles_n, tn = bid % tiles_n;
    // one DMA instruction = 16 rows x 64 B: lane -> (row = lane >> 2, position = lane & 3); wave w stages rows [16w, 16w+16) of A and of W
    const int srow = lane >> 2, spos = lane & 3;
    const int r = 16 * wave + srow;
    const int c = spos ^ ((r >> 2) & 3);  // logical chunk stored at this position
    int m = tm * BM + r; m = m < M ? m : M - 1;
    int n = tn * BN + r; n = n < N ? n : N - 1;
    const float *ga = A + (size_t)m * K + c * 4, *gb = W + (size_t)n * K + c * 4;
    auto issue = [&](int kt, int st) {
        float *sA = smem + st * STAGE_F, *sB = sA + BM * BK;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(ga + kt * BK),
                                         (__attribute__((address_space(3))) void *)(sA + 16 * wave * BK), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gb + kt * BK),
                                         (__attribute__((address_space(3))) void *)(sB + 16 * wave * BK), 16, 0, 0);
    };
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const int ra = wm * 32 + l31, rb = wn * 32 + l31;
    const int swa = (ra >> 2) & 3, swb = (rb >> 2) & 3;
    const int nkt = K / BK;
    issue(0, 0);
    if (nkt > 1) issue(1, 1);
    for (int kt = 0; kt < nkt; ++kt) {
        // tile kt has landed when at most one younger tile (2 DMAs) is still in flight
        if (kt + 1 < nkt) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // everyone's part of tile kt is visible; everyone finished reading tile kt-1
        if (kt + 2 < nkt) issue(kt + 2, (kt + 2) % NST);  // that stage held tile kt-1
        const float *sA = smem + (kt % NST) * STAGE_F, *sB = sA + BM * BK;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int cq = 2 * q + lh;
            const f32x4 fa = *(const f32x4 *)&sA[ra * BK + ((cq ^ swa) << 2)];
            const f32x4 fb = *(const f32x4 *)&sB[rb * BK + ((cq ^ swb) << 2)];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[e], fb[e], acc, 0, 0, 0);
        }
    }
    const int nn = tn * BN + wn * 32 + l31;
    for (int i = 0; i < 16; ++i) {
        const int mm = tm * BM + wm * 32 + (i & 3) + 8 * (i >> 2) + 4 * lh;
        if (mm < M && nn < N) C[(size_t)mm * N + nn] = acc[i];
    }
}

int main(int argc, char **argv) {
    for (int pad_kb : {24, 27, 32, 40}) {
        const size_t lds = (size_t)pad_kb * 1024;
        hipFuncSetAttribute((const void *)k16, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        int occ = 0;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k16, 256, lds);
        printf("== LDS %d KB per workgroup: API says %d workgroups per CU\n", pad_kb, occ);
        {
            const int M = 300, N = 200, K = 96;
            std::vector<float> hA((size_t)M * K), hW((size_t)N * K), hC((size_t)M * N);
            srand(1);
            for (auto &v : hA) v = (rand() % 2001 - 1000) / 1000.f;
            for (auto &v : hW) v = (rand() % 2001 - 1000) / 1000.f;
            float *A, *W, *C; hipMalloc(&A, hA.size() * 4); hipMalloc(&W, hW.size() * 4); hipMalloc(&C, hC.size() * 4);
            hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice); hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice);
            hipMemset(C, 0xff, hC.size() * 4);
            k16<<<((M + 63) / 64) * ((N + 63) / 64), 256, lds>>>(A, W, C, M, N, K);
            hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost);
            double worst = 0;
            for (int m = 0; m < M; ++m) for (int n = 0; n < N; ++n) {
                double s = 0; for (int k = 0; k < K; ++k) s += (double)hA[(size_t)m * K + k] * hW[(size_t)n * K + k];
                worst = fmax(worst, fabs(s - hC[(size_t)m * N + n]));
            }
            printf("correctness: max abs err %.3e %s\n", worst, worst < 1e-4 ? "OK" : "WRONG");
            hipFree(A); hipFree(W); hipFree(C);
        }
        auto bench = [&](int M, int N, int K, const char *what) {
            float *A, *W, *C; hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&W, (size_t)N * K * 4); hipMalloc(&C, (size_t)M * N * 4);
            std::vector<float> h((size_t)M * K); for (auto &v : h) v = (rand() % 2001 - 1000) / 1000.f; hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice);
            h.resize((size_t)N * K); for (auto &v : h) v = (rand() % 2001 - 1000) / 1000.f; hipMemcpy(W, h.data(), h.size() * 4, hipMemcpyHostToDevice);
            dim3 grid(((M + 63) / 64) * ((N + 63) / 64));
            k16<<<grid, 256, lds>>>(A, W, C, M, N, K); hipDeviceSynchronize();
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0); for (int i = 0; i < 20; ++i) k16<<<grid, 256, lds>>>(A, W, C, M, N, K); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
            printf("%-8s M=%6d N=%5d K=%5d: %8.1f us  %6.1f TF\n", what, M, N, K, ms * 1e3, 2.0 * M * N * K / (ms * 1e-3) / 1e12);
            hipFree(A); hipFree(W); hipFree(C);
        };
        bench(10960, 1152, 384, "qkv T8"); bench(10960, 1536, 384, "fc1 T8"); bench(10960, 384, 384, "proj T8"); bench(10960, 384, 1536, "fc2 T8");
        bench(43840, 1152, 384, "qkv T32"); bench(4096, 4096, 4096, "4096^3");
    }
    return 0;
}
