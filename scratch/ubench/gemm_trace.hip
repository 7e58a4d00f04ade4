// Timeline of the LDS-DMA GEMM: every workgroup stamps wall_clock64() (100 MHz) at entry (0), after issuing its first
// DMA (1), when the first tile has landed (2), after the k loop (3) and after the epilogue (4).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
constexpr int MAXWG = 16384;
__device__ long long g_stamp[MAXWG * 5];
#define EDV_GEMM_STAMP(slot)                                                                          \
    do {                                                                                              \
        if (threadIdx.x == 0 && blockIdx.x < MAXWG) g_stamp[blockIdx.x * 5 + (slot)] = wall_clock64(); \
    } while (0)
#ifdef NOSTORE
#define EDV_EPI_STORE_COND && v == 12345.f
#endif
#include "../../endodav_amd/csrc/gemm_dma.hip"
namespace edv { void set_error(const std::string &m) { fprintf(stderr, "error: %s\n", m.c_str()); } }

int main(int argc, char **argv) {
    const long long M = argc > 1 ? atoll(argv[1]) : 10960;
    const int N = argc > 2 ? atoi(argv[2]) : 1152, K = argc > 3 ? atoi(argv[3]) : 384;
    float *A, *W, *C, *b;
    hipMalloc(&A, M * K * 4); hipMalloc(&W, (size_t)N * K * 4); hipMalloc(&C, M * N * 4); hipMalloc(&b, N * 4);
    hipMemset(A, 0, M * K * 4); hipMemset(W, 0, (size_t)N * K * 4); hipMemset(b, 0, N * 4);
    edv::GemmDesc g;
    g.A = A; g.lda = K; g.W = W; g.ldw = K; g.C = C; g.ldc = N; g.M = M; g.N = N; g.K = K; g.bias = b;
    for (int it = 0; it < 3; ++it) {
        if (edv::gemm_dma(g, nullptr)) return 1;
        hipDeviceSynchronize();
    }
    const int nwg = (int)std::min<long long>(MAXWG, ((M + 63) / 64) * ((N + 63) / 64));
    std::vector<long long> s((size_t)MAXWG * 5);
    hipMemcpyFromSymbol(s.data(), HIP_SYMBOL(g_stamp), s.size() * 8);
    long long t0 = s[0], t1 = 0;
    for (int w = 0; w < nwg; ++w) { t0 = std::min(t0, s[w * 5]); t1 = std::max(t1, s[w * 5 + 4]); }
    printf("M=%lld N=%d K=%d: %d workgroups, first start -> last end %.2f us\n", M, N, K, nwg, (t1 - t0) / 100.0);
    double d[4] = {0, 0, 0, 0};
    for (int w = 0; w < nwg; ++w) for (int k = 0; k < 4; ++k) d[k] += (s[w * 5 + k + 1] - s[w * 5 + k]) / 100.0 / nwg;
    printf("mean per workgroup [us]: entry->first DMA issued %.2f, ->first tile landed %.2f, k loop %.2f, epilogue %.2f\n", d[0], d[1], d[2], d[3]);
    // concurrency profile: workgroups alive per 5 us bucket, and starts per bucket
    const int nb = (int)((t1 - t0) / 500) + 1;
    std::vector<double> alive(nb, 0); std::vector<int> starts(nb, 0), ends(nb, 0);
    for (int w = 0; w < nwg; ++w) {
        const long long a = s[w * 5] - t0, e = s[w * 5 + 4] - t0;
        starts[a / 500]++; ends[e / 500]++;
        for (int k = (int)(a / 500); k <= (int)(e / 500); ++k) {
            const long long lo = std::max<long long>(a, k * 500ll), hi = std::min<long long>(e, (k + 1) * 500ll);
            alive[k] += (hi - lo) / 500.0;
        }
    }
    printf("bucket(5us): alive(avg) starts ends\n");
    for (int k = 0; k < nb; ++k) printf("%3d: %7.1f %5d %5d\n", k * 5, alive[k], starts[k], ends[k]);
    return 0;
}
