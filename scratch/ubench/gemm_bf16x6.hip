// Feasibility microbenchmark (round 3): an fp32-in / fp32-out GEMM C = A W^T whose products run on the bf16 matrix pipe.
//
//   a = a0 + a1 + a2 with a0 = bf16(a), a1 = bf16(a - a0), a2 = bf16(a - a0 - a1)   (|a1| <= 2^-9 |a|, |a2| <= 2^-18 |a|, remainder <= 2^-27 |a|)
//   a b ~= a0 b0 + (a0 b1 + a1 b0) + (a1 b1 + a0 b2 + a2 b0)                         (dropped: a1 b2 + a2 b1 + a2 b2 <= 2^-26 |a b|)
//
// Six v_mfma_f32_32x32x16_bf16 (fp32 accumulate; a bf16 x bf16 product is exact in fp32) replace eight v_mfma_f32_32x32x2_f32 per 16 k: 192 matrix-pipe
// cycles instead of 512, with a per-product error below fp32's own rounding.  W is split once (weights are static); A is split on the way from HBM to LDS
// by the waves that stage it (VALU work, which -- unlike beside the f32 MFMA -- hides in the bf16 MFMA's issue gaps: MI355X_MICROARCH.md, cycle constants).
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++20 scratch/ubench/gemm_bf16x6.hip -o scratch/ubench/gemm_bf16x6 -ldl
// Run  : scratch/ubench/gemm_bf16x6 [path to libendodav_hip.so]     (with the library: the product fp32-MFMA GEMM timed on the same shapes)
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));   \
            exit(1);                                                                            \
        }                                                                                       \
    } while (0)

// ---- split of the static operand: W [N][K] f32 -> planes [3][N][K] bf16 ---------------------------------------------------------------------------------
__global__ void split3_kernel(const float *__restrict__ w, __bf16 *__restrict__ p, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float x = w[i];
        const __bf16 h = (__bf16)x;
        const float r1 = x - (float)h;
        const __bf16 m = (__bf16)r1;
        const float r2 = r1 - (float)m;
        p[i] = h;
        p[n + i] = m;
        p[2 * n + i] = (__bf16)r2;
    }
}

constexpr int BM = 128, BN = 128, BK = 16;            // one stage = 16 k of a 128 x 128 tile
constexpr int PLANE = BM * BK * 2;                    // bytes of one operand plane of a stage (4 KB)
constexpr int STAGE = 6 * PLANE;                      // A0 A1 A2 W0 W1 W2 (24 KB)
constexpr int NST = 3;                                // stages: 72 KB per workgroup, two workgroups per CU

// LDS image of a plane: 32-byte rows (16 bf16), the two 16-byte halves of row r swapped when (r >> 3) & 1: a 16-lane group of a ds_read_b128
// (rows r..r+15 of one half) then covers all 64 banks.
__device__ __forceinline__ int half_pos(int r, int h) { return (h ^ ((r >> 3) & 1)) * 16; }

// workgroup id -> tile: consecutive tiles (same A rows) on the same XCD (workgroups go round-robin over the 8 XCDs)
__device__ __forceinline__ int xcd_tile(int b, int G) {
    const int per = G / 8, rem = G % 8, x = b % 8, i = b / 8;
    return x * per + (x < rem ? x : rem) + i;
}

template <int NMFMA>  // 6 = the scheme above; 3 = a0 b0 + a0 b1 + a1 b0 (error 2^-17: what "bf16x3" would give); 1 = plain bf16
__global__ __launch_bounds__(256, 2) void gemm_bf16x6_kernel(const float *__restrict__ A, const __bf16 *__restrict__ Wp, float *__restrict__ C, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = N / BN;
    const int tile = xcd_tile(blockIdx.x, gridDim.x);
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const long long m0 = (long long)tm * BM;
    const int n0 = tn * BN;
    const int nkt = K / BK;
    const auto rW = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16 *>(Wp), 0, 0xffffffff, 0x00020000);
    const long long plane_bytes = (long long)N * K * 2;

    // ---- staging roles ----
    // W: 12 DMA instructions per stage (3 planes x 4 groups of 32 rows), three per wave.  Lane i fills chunk i of the 1 KB the instruction writes.
    unsigned vw[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int id = wave * 3 + j, p = id >> 2, grp = id & 3;
        const int r = grp * 32 + (lane >> 1), pos = lane & 1;
        const int h = pos ^ ((r >> 3) & 1);
        vw[j] = (unsigned)(p * plane_bytes + ((long long)(n0 + r) * K + h * 8) * 2);
    }
    // A: two 16-byte chunks of f32 per thread per stage (row = c / 4, floats (c % 4) * 4 ..)
    const float *ga[2];
    int dsta[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = tid + 256 * j, r = c >> 2, q = c & 3;
        long long m = m0 + r;
        m = m < M ? m : M - 1;
        ga[j] = A + m * K + q * 4;
        dsta[j] = r * 32 + half_pos(r, q >> 1) + (q & 1) * 8;
    }
    auto issue_w = [&](int kt, int st) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int id = wave_s * 3 + j;
            if (NMFMA == 1 && (id >> 2) > 0) continue;
            if (NMFMA == 3 && (id >> 2) > 1) continue;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (__attribute__((address_space(3))) void *)(smem + st * STAGE + (3 + (id >> 2)) * PLANE + (id & 3) * 1024), 16, vw[j],
                                                     (int)(kt * (BK * 2)), 0, 0);
        }
    };
    auto load_a = [&](int kt, f32x4 (&v)[2]) {
#pragma unroll
        for (int j = 0; j < 2; ++j) v[j] = *reinterpret_cast<const f32x4 *>(ga[j] + kt * BK);
    };
    auto split_store_a = [&](const f32x4 (&v)[2], int st) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            bf16x4 p0, p1, p2;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float x = v[j][e];
                p0[e] = (__bf16)x;
                const float r1 = x - (float)p0[e];
                p1[e] = (__bf16)r1;
                p2[e] = (__bf16)(r1 - (float)p1[e]);
            }
            unsigned char *d = smem + st * STAGE + dsta[j];
            *reinterpret_cast<bf16x4 *>(d) = p0;
            if (NMFMA > 1) *reinterpret_cast<bf16x4 *>(d + PLANE) = p1;
            if (NMFMA > 3) *reinterpret_cast<bf16x4 *>(d + 2 * PLANE) = p2;
        }
    };

    // ---- fragment addresses (stage 0) ----
    int fa[2], fb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ra = wm * 64 + i * 32 + l31, rb = wn * 64 + i * 32 + l31;
        fa[i] = ra * 32 + half_pos(ra, lh);
        fb[i] = 3 * PLANE + rb * 32 + half_pos(rb, lh);
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // prologue: stages 0 and 1 filled, A(2) in registers
    f32x4 areg[2];
    load_a(0, areg);
    issue_w(0, 0);
    split_store_a(areg, 0);
    if (nkt > 1) {
        load_a(1, areg);
        issue_w(1, 1);
        split_store_a(areg, 1);
    }
    if (nkt > 2) load_a(2, areg);
    constexpr int NDMA = NMFMA == 1 ? 1 : (NMFMA == 3 ? 2 : 3);  // (a wave issues at most this many W DMAs per stage; fewer for some waves when planes are skipped)
    if (nkt > 2 && NMFMA == 6)
        asm volatile("s_waitcnt vmcnt(5)" ::: "memory");  // W(0) landed; W(1) x3 and A(2) x2 may stay in flight
    else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    (void)NDMA;
    __syncthreads();

    constexpr int NPL = NMFMA == 1 ? 1 : (NMFMA == 3 ? 2 : 3);
    auto step = [&](int kt, auto st_tag) {
        constexpr int ST = decltype(st_tag)::value, S2 = (ST + 2) % NST;
        if (kt + 2 < nkt) {
            issue_w(kt + 2, S2);        // stage S2 was read during iteration kt - 1
            split_store_a(areg, S2);    // A(kt + 2), loaded one iteration ago
            if (kt + 3 < nkt) load_a(kt + 3, areg);
        }
        bf16x8 a[2][NPL], b[2][NPL];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int p = 0; p < NPL; ++p) {
                a[i][p] = *reinterpret_cast<const bf16x8 *>(smem + ST * STAGE + fa[i] + p * PLANE);
                b[i][p] = *reinterpret_cast<const bf16x8 *>(smem + ST * STAGE + fb[i] + p * PLANE);
            }
        // smallest terms first
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int t = 6 - NMFMA; t < 6; ++t)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][PA[t]], b[j][PB[t]], acc[i][j], 0, 0, 0);
        // W(kt + 1) (issued one iteration ago) has landed: younger than it are A(kt + 2) x2, W(kt + 2) x3, A(kt + 3) x2
        if (kt + 3 < nkt && NMFMA == 6)
            asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    {
        int kt = 0;
        for (; kt + 2 < nkt; kt += 3) {
            step(kt, std::integral_constant<int, 0>{});
            step(kt + 1, std::integral_constant<int, 1>{});
            step(kt + 2, std::integral_constant<int, 2>{});
        }
        if (kt < nkt) step(kt, std::integral_constant<int, 0>{});
        if (kt + 1 < nkt) step(kt + 1, std::integral_constant<int, 1>{});
    }

#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long row = m0 + wm * 64 + i * 32 + (r >> 2) * 8 + lh * 4 + (r & 3);
                const int col = n0 + wn * 64 + j * 32 + l31;
                if (row < M) C[row * N + col] = acc[i][j][r];
            }
}

// fp32 FMA-chain reference of a few output elements is too slow on the host for the big shapes: check a sample of rows in fp64 instead
static void check(const std::vector<float> &A, const std::vector<float> &W, const std::vector<float> &C, int M, int N, int K, const char *tag) {
    double worst = 0, sum2 = 0, ref2 = 0;
    long long cnt = 0;
    for (int s = 0; s < 64; ++s) {
        const long long m = (long long)((s * 2654435761u) % (unsigned)M);
        for (int n = s % 7; n < N; n += 13) {
            double ref = 0, mag = 0;
            for (int k = 0; k < K; ++k) {
                ref += (double)A[m * K + k] * (double)W[(long long)n * K + k];
                mag += std::fabs((double)A[m * K + k] * (double)W[(long long)n * K + k]);
            }
            const double e = std::fabs((double)C[m * N + n] - ref) / mag;  // relative to sum |a_k b_k|: the bound an fp32 dot product is judged by
            worst = e > worst ? e : worst;
            sum2 += e * e;
            ref2 += 1;
            ++cnt;
        }
    }
    printf("    %-22s max |err| / sum|a b| = %.3e   rms = %.3e   (%lld samples; fp32 unit roundoff 5.96e-08)\n", tag, worst, std::sqrt(sum2 / ref2), cnt);
}

typedef int (*gemm_fn)(const float *, const float *, float *, int64_t, int32_t, int32_t, const float *, int32_t, const float *, const float *, float *, size_t, void *);
typedef size_t (*ws_fn)(void);

template <int NMFMA>
static float time_kernel(const float *dA, const __bf16 *dWp, float *dC, int M, int N, int K, int iters) {
    const int grid = ((M + BM - 1) / BM) * (N / BN);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(gemm_bf16x6_kernel<NMFMA>, dim3(grid), dim3(256), NST * STAGE, 0, dA, dWp, dC, M, N, K);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(gemm_bf16x6_kernel<NMFMA>, dim3(grid), dim3(256), NST * STAGE, 0, dA, dWp, dC, M, N, K);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1000.f / iters;
}

int main(int argc, char **argv) {
    gemm_fn edv_gemm = nullptr;
    ws_fn edv_ws = nullptr;
    if (argc > 1) {
        void *h = dlopen(argv[1], RTLD_NOW);
        if (!h) {
            fprintf(stderr, "dlopen: %s\n", dlerror());
            return 1;
        }
        edv_gemm = (gemm_fn)dlsym(h, "edv_gemm");
        edv_ws = (ws_fn)dlsym(h, "edv_gemm_workspace");
    }
    struct Shape { int M, N, K; const char *what; };
    const Shape shapes[] = {
        {10960, 1152, 384, "ViT-S T=8 qkv"},   {10960, 384, 384, "ViT-S T=8 proj"},    {10960, 1536, 384, "ViT-S T=8 fc1"},   {10960, 384, 1536, "ViT-S T=8 fc2"},
        {21920, 2304, 768, "ViT-B T=16 qkv"},  {21920, 768, 3072, "ViT-B T=16 fc2"},   {43840, 3072, 1024, "ViT-L T=32 qkv"}, {43840, 1024, 4096, "ViT-L T=32 fc2"},
        {4096, 4096, 4096, "4096^3"},
    };
    CK(hipFuncSetAttribute((const void *)gemm_bf16x6_kernel<6>, hipFuncAttributeMaxDynamicSharedMemorySize, NST * STAGE));
    CK(hipFuncSetAttribute((const void *)gemm_bf16x6_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, NST * STAGE));
    CK(hipFuncSetAttribute((const void *)gemm_bf16x6_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, NST * STAGE));
    float *dws = nullptr;
    size_t wsb = 0;
    if (edv_ws) {
        wsb = edv_ws();
        CK(hipMalloc(&dws, wsb));
        CK(hipMemset(dws, 0, wsb));
    }
    for (const Shape &s : shapes) {
        const long long na = (long long)s.M * s.K, nw = (long long)s.N * s.K, nc = (long long)s.M * s.N;
        std::vector<float> A(na), W(nw), C(nc);
        uint32_t z = 12345u + s.M;
        auto rnd = [&]() {
            z = z * 1664525u + 1013904223u;
            return ((z >> 8) * (1.0f / 8388608.0f) - 1.0f);  // uniform in [-1, 1), 24 random bits
        };
        for (auto &v : A) v = rnd() * 2.0f;
        for (auto &v : W) v = rnd() * 0.05f;
        float *dA, *dW, *dC;
        __bf16 *dWp;
        CK(hipMalloc(&dA, na * 4));
        CK(hipMalloc(&dW, nw * 4));
        CK(hipMalloc(&dC, nc * 4));
        CK(hipMalloc(&dWp, nw * 2 * 3));
        CK(hipMemcpy(dA, A.data(), na * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dW, W.data(), nw * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(split3_kernel, dim3(1024), dim3(256), 0, 0, dW, dWp, nw);
        CK(hipDeviceSynchronize());
        const double gf = 2.0 * s.M * s.N * s.K * 1e-9;
        const int iters = 50;
        printf("%-16s M=%d N=%d K=%d  (%.2f GFLOP, %d tiles of 128x128)\n", s.what, s.M, s.N, s.K, gf, ((s.M + BM - 1) / BM) * (s.N / BN));
        const float t6 = time_kernel<6>(dA, dWp, dC, s.M, s.N, s.K, iters);
        CK(hipMemcpy(C.data(), dC, nc * 4, hipMemcpyDeviceToHost));
        printf("  bf16 x6 : %8.2f us  %7.1f TFLOP/s (fp32-equivalent)   %7.1f TFLOP/s on the bf16 pipe\n", t6, gf / t6 * 1e3, 6 * gf / t6 * 1e3);
        check(A, W, C, s.M, s.N, s.K, "bf16 x6");
        const float t3 = time_kernel<3>(dA, dWp, dC, s.M, s.N, s.K, iters);
        CK(hipMemcpy(C.data(), dC, nc * 4, hipMemcpyDeviceToHost));
        printf("  bf16 x3 : %8.2f us  %7.1f TFLOP/s\n", t3, gf / t3 * 1e3);
        check(A, W, C, s.M, s.N, s.K, "bf16 x3");
        const float t1 = time_kernel<1>(dA, dWp, dC, s.M, s.N, s.K, iters);
        CK(hipMemcpy(C.data(), dC, nc * 4, hipMemcpyDeviceToHost));
        printf("  bf16 x1 : %8.2f us  %7.1f TFLOP/s\n", t1, gf / t1 * 1e3);
        check(A, W, C, s.M, s.N, s.K, "bf16 x1");
        if (edv_gemm && s.K % 32 == 0) {
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0));
            CK(hipEventCreate(&e1));
            for (int i = 0; i < 5; ++i) edv_gemm(dA, dW, dC, s.M, s.N, s.K, nullptr, 0, nullptr, nullptr, dws, wsb, nullptr);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int i = 0; i < iters; ++i) edv_gemm(dA, dW, dC, s.M, s.N, s.K, nullptr, 0, nullptr, nullptr, dws, wsb, nullptr);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const float tf = ms * 1000.f / iters;
            CK(hipMemcpy(C.data(), dC, nc * 4, hipMemcpyDeviceToHost));
            printf("  f32 MFMA: %8.2f us  %7.1f TFLOP/s   (the product GEMM, stream-K)   x6 speed-up %.2f\n", tf, gf / tf * 1e3, tf / t6);
            check(A, W, C, s.M, s.N, s.K, "f32 MFMA (product)");
        }
        CK(hipFree(dA));
        CK(hipFree(dW));
        CK(hipFree(dC));
        CK(hipFree(dWp));
    }
    return 0;
}
