// How many 256-thread workgroups with a given static LDS size does one MI355X CU really hold?  (The occupancy API answers 5 for
// 32 KB; the GEMM timeline showed 4.)  Each workgroup stamps its start, idles for a fixed wall-clock interval (bounded: every wave
// leaves after `hold` ticks), and the host counts how many workgroups started within the first few microseconds.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

template <int BYTES>
__global__ __launch_bounds__(256) void hold_kernel(unsigned long long *start, unsigned long long hold, float *sink) {
    __shared__ float lds[BYTES / 4];
    lds[threadIdx.x] = (float)threadIdx.x;
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) start[blockIdx.x] = t0;
    while (wall_clock64() - t0 < hold) __builtin_amdgcn_s_sleep(8);
    if (lds[(threadIdx.x * 7) % (BYTES / 4)] == -1.f) sink[0] = 1.f;
}

template <int BYTES>
void probe(int cus) {
    const int G = cus * 8;
    unsigned long long *d; float *sink;
    hipMalloc(&d, G * sizeof(unsigned long long)); hipMalloc(&sink, 4);
    int api = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&api, hold_kernel<BYTES>, 256, 0);
    const unsigned long long hold = 100 * 100;  // wall_clock64 ticks at 100 MHz: 100 us
    hold_kernel<BYTES><<<G, 256>>>(d, hold, sink);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(G);
    hipMemcpy(h.data(), d, G * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    const unsigned long long first = *std::min_element(h.begin(), h.end());
    int early = 0;
    for (auto v : h) early += (v - first) < hold / 2;  // started in the first wave of residency
    printf("LDS %6d B: occupancy API %d per CU; %d of %d workgroups resident at once = %.2f per CU\n", BYTES, api, early, G, (double)early / cus);
    hipFree(d); hipFree(sink);
}

int main() {
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    printf("%d CUs\n", cus);
    probe<65536>(cus); probe<40960>(cus); probe<33792>(cus); probe<32768>(cus); probe<32256>(cus); probe<31744>(cus); probe<30720>(cus);
    probe<28672>(cus); probe<26624>(cus); probe<24576>(cus); probe<20480>(cus); probe<16384>(cus);
    return 0;
}
