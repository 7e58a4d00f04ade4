// Ablation of the 64x64x32 two-stage fp32-MFMA GEMM main loop (timing only; results are garbage for ABL > 0).
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int BK = 32, LS = 36, BM = 64, BN = 64;

// ABL bit 0: no global loads in the loop; bit 1: no LDS stage stores; bit 2: no barriers; bit 3: no LDS fragment reads
template <int ABL>
__global__ __launch_bounds__(256) void k(const float *A, const float *W, float *C, int M, int N, int K) {
    constexpr int STAGE = (BM + BN) * LS;
    __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = N / BN;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
    const int c = tid & 7, r0 = tid >> 3;
    const float *ap[2], *bp[2];
    for (int i = 0; i < 2; ++i) { ap[i] = A + (size_t)(tm * BM + r0 + 32 * i) * K; bp[i] = W + (size_t)(tn * BN + r0 + 32 * i) * K; }
    f32x4 ra[2], rb[2];
    auto load = [&](int kt) { for (int i = 0; i < 2; ++i) { ra[i] = *(const f32x4 *)(ap[i] + kt * BK + c * 4); rb[i] = *(const f32x4 *)(bp[i] + kt * BK + c * 4); } };
    auto store = [&](int buf) { float *sA = smem + buf * STAGE, *sB = sA + BM * LS; for (int i = 0; i < 2; ++i) { *(f32x4 *)&sA[(r0 + 32 * i) * LS + c * 4] = ra[i]; *(f32x4 *)&sB[(r0 + 32 * i) * LS + c * 4] = rb[i]; } };
    f32x16 acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    f32x4 fa = {1.f, 2.f, 3.f, 4.f}, fb = {0.5f, 0.25f, 0.125f, 1.f};
    auto mq = [&](int buf, int q) {
        const float *sA = smem + buf * STAGE, *sB = sA + BM * LS;
        if (!(ABL & 8)) { fa = *(const f32x4 *)&sA[(wm * 32 + l31) * LS + 8 * q + 4 * lh]; fb = *(const f32x4 *)&sB[(wn * 32 + l31) * LS + 8 * q + 4 * lh]; }
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[e], fb[e], acc, 0, 0, 0);
    };
    const int nkt = K / BK;
    load(0); store(0); load(1); __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        mq(cur, 0); mq(cur, 1);
        if (!(ABL & 2)) store(cur ^ 1);
        if (!(ABL & 1) && kt + 2 < nkt) load(kt + 2);
        mq(cur, 2); mq(cur, 3);
        if (!(ABL & 4)) __syncthreads();
    }
    const int n = tn * BN + wn * 32 + l31;
    for (int r = 0; r < 16; ++r) C[(size_t)(tm * BM + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * N + n] = acc[r] + ra[0].x + rb[0].x;
}

template <int ABL> void run(const float *A, const float *W, float *C, int M, int N, int K, const char *what) {
    dim3 grid((M / BM) * (N / BN));
    k<ABL><<<grid, 256>>>(A, W, C, M, N, K);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) k<ABL><<<grid, 256>>>(A, W, C, M, N, K);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("%-52s %8.1f us  %6.1f TF\n", what, ms * 1e3, 2.0 * M * N * K / (ms * 1e-3) / 1e12);
}
int main() {
    const int M = 8192, N = 8192, K = 1024;
    float *A, *W, *C; hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&W, (size_t)N * K * 4); hipMalloc(&C, (size_t)M * N * 4);
    hipMemset(A, 0x3c, (size_t)M * K * 4); hipMemset(W, 0x3c, (size_t)N * K * 4);
    run<0>(A, W, C, M, N, K, "full");
    run<1>(A, W, C, M, N, K, "no global loads in loop");
    run<3>(A, W, C, M, N, K, "no global loads, no LDS stores");
    run<7>(A, W, C, M, N, K, "no loads, no stores, no barriers");
    run<15>(A, W, C, M, N, K, "MFMA only (no LDS fragment reads either)");
    run<4>(A, W, C, M, N, K, "full but no barriers (racy)");
    run<8>(A, W, C, M, N, K, "full but no LDS fragment reads");
    run<2>(A, W, C, M, N, K, "full but no LDS stores");
    return 0;
}
