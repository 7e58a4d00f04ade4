// Timeline of the bf16 x 6 GEMM (plain grid): every workgroup stamps wall_clock64() (100 MHz) at tile start (0), when its first two stages are filled (1),
// after the k loop (2) and after the epilogue (3).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++20 scratch/ubench/gemm_x6_trace.hip -o scratch/ubench/gemm_x6_trace
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
constexpr int MAXWG = 16384;
__device__ long long g_stamp[MAXWG * 4];
#define EDV_X6_STAMP(slot)                                                                            \
    do {                                                                                              \
        if (threadIdx.x == 0 && blockIdx.x < MAXWG) g_stamp[blockIdx.x * 4 + (slot)] = wall_clock64(); \
    } while (0)
#include "../../endodav_amd/csrc/gemm_x6.hip"
namespace edv {
void set_error(const std::string &m) { fprintf(stderr, "error: %s\n", m.c_str()); }
thread_local LaunchTimer *g_launch_timer = nullptr;
}

int main(int argc, char **argv) {
    const long long M = argc > 1 ? atoll(argv[1]) : 10960;
    const int N = argc > 2 ? atoi(argv[2]) : 1152, K = argc > 3 ? atoi(argv[3]) : 384;
    float *A, *W, *C, *b;
    void *P;
    hipMalloc(&A, M * K * 4); hipMalloc(&W, (size_t)N * K * 4); hipMalloc(&C, M * N * 4); hipMalloc(&b, N * 4); hipMalloc(&P, (size_t)N * K * 6);
    std::vector<float> h((size_t)std::max<long long>(M * K, (long long)N * K));
    unsigned z = 1;
    for (auto &v : h) { z = z * 1664525u + 1013904223u; v = ((z >> 8) * (1.0f / 8388608.0f) - 1.0f); }
    hipMemcpy(A, h.data(), M * K * 4, hipMemcpyHostToDevice); hipMemcpy(W, h.data(), (size_t)N * K * 4, hipMemcpyHostToDevice); hipMemset(b, 0, N * 4);
    if (edv::gemm_x6_split(W, P, N, K, nullptr)) return 1;
    edv::GemmDesc g;
    g.A = A; g.lda = K; g.W = W; g.ldw = K; g.C = C; g.ldc = N; g.M = M; g.N = N; g.K = K; g.bias = b; g.Wx6 = P;
    for (int it = 0; it < 3; ++it) {
        if (edv::gemm_x6(g, nullptr)) return 1;
        hipDeviceSynchronize();
    }
    const int nwg = (int)std::min<long long>(MAXWG, ((M + 127) / 128) * ((N + 127) / 128));
    std::vector<long long> s((size_t)MAXWG * 4);
    hipMemcpyFromSymbol(s.data(), HIP_SYMBOL(g_stamp), s.size() * 8);
    long long t0 = s[0], t1 = 0;
    for (int w = 0; w < nwg; ++w) { t0 = std::min(t0, s[w * 4]); t1 = std::max(t1, s[w * 4 + 3]); }
    printf("M=%lld N=%d K=%d: %d workgroups (%d k-steps each), first start -> last end %.2f us\n", M, N, K, nwg, K / 16, (t1 - t0) / 100.0);
    std::vector<double> d[3];
    for (int w = 0; w < nwg; ++w) for (int k = 0; k < 3; ++k) d[k].push_back((s[w * 4 + k + 1] - s[w * 4 + k]) / 100.0);
    const char *nm[3] = {"tile start -> two stages filled", "k loop", "epilogue"};
    for (int k = 0; k < 3; ++k) {
        std::sort(d[k].begin(), d[k].end());
        double m = 0; for (double v : d[k]) m += v;
        printf("  %-32s mean %6.2f us   p10 %6.2f   median %6.2f   p90 %6.2f\n", nm[k], m / nwg, d[k][nwg / 10], d[k][nwg / 2], d[k][nwg * 9 / 10]);
    }
    const int nb = (int)((t1 - t0) / 500) + 1;
    std::vector<double> alive(nb, 0); std::vector<int> starts(nb, 0);
    for (int w = 0; w < nwg; ++w) {
        const long long a = s[w * 4] - t0, e = s[w * 4 + 3] - t0;
        starts[a / 500]++;
        for (int k = (int)(a / 500); k <= (int)(e / 500); ++k) {
            const long long lo = std::max<long long>(a, k * 500ll), hi = std::min<long long>(e, (k + 1) * 500ll);
            alive[k] += (hi - lo) / 500.0;
        }
    }
    printf("bucket(5us): workgroups alive (avg), starts\n");
    for (int k = 0; k < nb; ++k) printf("%3d: %7.1f %5d\n", k * 5, alive[k], starts[k]);
    return 0;
}
