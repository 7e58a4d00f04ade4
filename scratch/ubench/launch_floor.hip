// Per-launch floor of a dependent chain of tiny kernels: plain stream launches vs a captured hipGraph.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void tiny(float *p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.f; }
__global__ void tiny_grid(float *p, int n) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n) p[i] += 1.f; }
int main() {
    float *p; hipMalloc(&p, 1 << 22); hipMemset(p, 0, 1 << 22);
    hipStream_t s; hipStreamCreate(&s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int N = 2000;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0, s);
        for (int i = 0; i < N; ++i) tiny<<<1, 64, 0, s>>>(p);
        hipEventRecord(e1, s); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep) printf("1 workgroup kernels, stream launches: %.2f us per launch\n", ms * 1e3 / N);
        hipEventRecord(e0, s);
        for (int i = 0; i < N; ++i) tiny_grid<<<1024, 256, 0, s>>>(p, 1 << 18);
        hipEventRecord(e1, s); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        if (rep) printf("1024-workgroup elementwise kernels (1 MB), stream launches: %.2f us per launch\n", ms * 1e3 / N);
    }
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < 200; ++i) tiny_grid<<<1024, 256, 0, s>>>(p, 1 << 18);
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, s); hipStreamSynchronize(s);
    hipEventRecord(e0, s);
    for (int i = 0; i < 10; ++i) hipGraphLaunch(ge, s);
    hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("same kernels from a 200-node hipGraph: %.2f us per kernel\n", ms * 1e3 / 2000);
    return 0;
}
