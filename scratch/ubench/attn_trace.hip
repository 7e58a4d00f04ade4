// Timeline of the attention kernel: every workgroup stamps wall_clock64() (100 MHz) at run start, before each key
// tile, after the last tile and after its stores, into g_stamp[workgroup][run][slot].  Includes the product source.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
constexpr int MAXS = 96, MAXR = 6;
__device__ long long g_stamp[1024 * MAXR * MAXS];
__device__ int g_run[1024];
#define EDV_ATTN_STAMP(slot)                                                                                    \
    do {                                                                                                        \
        if (threadIdx.x == 0 && blockIdx.x < 1024 && (slot) < MAXS && (round + seg) <= MAXR && (round + seg) > 0) \
            g_stamp[(blockIdx.x * MAXR + (round + seg - 1)) * MAXS + (slot)] = wall_clock64();                     \
    } while (0)
#include "../../endodav_amd/csrc/attn_spatial.hip"
namespace edv { void set_error(const std::string &m) { fprintf(stderr, "error: %s\n", m.c_str()); } }

int main(int argc, char **argv) {
    const int F = argc > 1 ? atoi(argv[1]) : 8, N = argc > 2 ? atoi(argv[2]) : 1370, heads = argc > 3 ? atoi(argv[3]) : 6;
    const size_t nq = (size_t)F * N * 3 * heads * 64, no = (size_t)F * N * heads * 64;
    std::vector<float> h(nq);
    for (size_t i = 0; i < nq; ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
    float *qkv, *out, *ws;
    hipMalloc(&qkv, nq * 4); hipMalloc(&out, no * 4);
    const size_t wsf = edv::attn_spatial_workspace(F, N, heads);
    hipMalloc(&ws, std::max<size_t>(wsf, 4) * 4);
    hipMemcpy(qkv, h.data(), nq * 4, hipMemcpyHostToDevice);
    for (int it = 0; it < 3; ++it) {
        std::vector<long long> z((size_t)1024 * MAXR * MAXS, 0);
        hipMemcpyToSymbol(HIP_SYMBOL(g_stamp), z.data(), z.size() * 8);
        if (edv::attn_spatial(qkv, out, F, N, heads, ws, wsf, nullptr)) return 1;
        hipDeviceSynchronize();
    }
    std::vector<long long> s((size_t)1024 * MAXR * MAXS);
    hipMemcpyFromSymbol(s.data(), HIP_SYMBOL(g_stamp), s.size() * 8);
    long long t0 = -1;
    for (int b = 0; b < 1024; ++b) { long long v = s[(size_t)b * MAXR * MAXS]; if (v && (t0 < 0 || v < t0)) t0 = v; }
    // per run index: start skew, mean per-tile time by tile index, end
    for (int r = 0; r < MAXR; ++r) {
        std::vector<double> sum(MAXS, 0); std::vector<int> cnt(MAXS, 0);
        double st_min = 1e18, st_max = 0, en_min = 1e18, en_max = 0; int nwg = 0;
        for (int b = 0; b < 1024; ++b) {
            const long long *p = &s[((size_t)b * MAXR + r) * MAXS];
            if (!p[0]) continue;
            ++nwg;
            int last = 0;
            for (int k = 1; k < MAXS; ++k) if (p[k]) last = k;
            st_min = std::min(st_min, (double)(p[0] - t0)); st_max = std::max(st_max, (double)(p[0] - t0));
            en_min = std::min(en_min, (double)(p[last] - t0)); en_max = std::max(en_max, (double)(p[last] - t0));
            for (int k = 1; k <= last; ++k) { sum[k] += (double)(p[k] - p[k - 1]); cnt[k]++; }
        }
        if (!nwg) continue;
        printf("run %d: %d workgroups, start %.2f..%.2f us, end %.2f..%.2f us\n  mean step [us] (slot k-1 -> k): ", r, nwg, st_min / 100, st_max / 100, en_min / 100, en_max / 100);
        for (int k = 1; k < MAXS; ++k) if (cnt[k]) printf("%.2f ", sum[k] / cnt[k] / 100);
        printf("\n");
    }
    return 0;
}
