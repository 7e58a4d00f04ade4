// One-off weight packing run by edv_prepare (tiny element-wise kernels).
//   pack_conv3x3  torch Conv2d weight [Co,Ci,3,3] -> [Co][ky][kx][Ci]: K-contiguous rows matching the
//                 channels-last implicit-GEMM loader of gemm.hip
//   pack_convT    torch ConvTranspose2d weight [Ci,Co,s,s] (kernel == stride, dpt.py:71-82) ->
//                 [(dy*s+dx)*Co + co][ci] and the bias tiled s*s times
//   fold_lora     W' = W + scale * (B∘V)(A∘U)   mylora/layers.py:148-157 (lora), :384-393 (dvlora)
//   fold_ssb      W' = a ∘ W ∘ b                mylora/layers.py:423-430
//   fold_dash     W' += U_top diag(idx) Vt_top  mylora/layers.py:581-583
// Folding replaces the reference's rank-r side product x Aᵀ Bᵀ by one GEMM on W'; the two differ by
// fp32 rounding only (checked by tests/test_kernels_gpu.py::test_fold_*).
#include "ops.hpp"

namespace edv {
namespace {

__global__ void pack_conv3x3_kernel(const float *__restrict__ w, float *__restrict__ out, int Co, int Ci) {
    const long long total = (long long)Co * Ci * 9;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int ci = (int)(i % Ci);
    const int tap = (int)((i / Ci) % 9);
    const int co = (int)(i / ((long long)Ci * 9));
    out[i] = w[((long long)co * Ci + ci) * 9 + tap];
}

__global__ void pack_convT_kernel(const float *__restrict__ w, float *__restrict__ wout, const float *__restrict__ b, float *__restrict__ bout, int Ci,
                                  int Co, int s) {
    const long long total = (long long)s * s * Co * Ci;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int ci = (int)(i % Ci);
    const long long n = i / Ci;  // (dy*s+dx)*Co + co
    const int co = (int)(n % Co);
    const int sub = (int)(n / Co);
    wout[i] = w[((long long)ci * Co + co) * s * s + sub];
    if (ci == 0) bout[n] = b[co];
}

__global__ void copy_kernel(const float *__restrict__ src, float *__restrict__ dst, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) dst[i] = src[i];
}

__global__ void fold_lora_kernel(const float *__restrict__ W, const float *__restrict__ A, const float *__restrict__ B, const float *__restrict__ U,
                                 const float *__restrict__ V, float scale, float *__restrict__ out, int nout, int nin, int r) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)nout * nin) return;
    const int k = (int)(i % nin), n = (int)(i / nin);
    float acc = 0.f;
    for (int j = 0; j < r; ++j) {
        float bv = B[(long long)n * r + j], av = A[(long long)j * nin + k];
        if (V) bv *= V[n];
        if (U) av *= U[j];
        acc += bv * av;
    }
    out[i] = W[i] + scale * acc;
}

__global__ void fold_ssb_kernel(const float *__restrict__ W, const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ out, int nout,
                                int nin) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)nout * nin) return;
    const int k = (int)(i % nin), n = (int)(i / nin);
    out[i] = a[k] * W[i] * b[n];
}

__global__ void fold_dash_kernel(const float *__restrict__ Ut, const float *__restrict__ idx, const float *__restrict__ Vt, float *__restrict__ io, int nout,
                                 int nin, int r) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)nout * nin) return;
    const int k = (int)(i % nin), n = (int)(i / nin);
    float acc = 0.f;
    for (int j = 0; j < r; ++j) acc += Ut[(long long)n * r + j] * idx[j] * Vt[(long long)j * nin + k];
    io[i] += acc;
}

inline unsigned blocks_for(long long n) { return (unsigned)((n + 255) / 256); }

// Eval-mode BatchNorm2d folded into the preceding convolution (use_bn=True: util/blocks.py:60-62,80-86):
// bn(conv(x)) = s * (W x + b - mean) + beta with s = gamma / sqrt(var + eps)  ->  W' = s ∘ W (row n of the packed weight), b' = s (b - mean) + beta.
__global__ void fold_bn_kernel(float *__restrict__ w, const float *__restrict__ b, const float *__restrict__ gamma, const float *__restrict__ beta,
                               const float *__restrict__ mean, const float *__restrict__ var, float eps, float *__restrict__ bout, int nout, int K) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)nout * K) return;
    const int n = (int)(i / K);
    const float s = gamma[n] / sqrtf(var[n] + eps);
    w[i] *= s;
    if (i - (long long)n * K == 0) bout[n] = (b[n] - mean[n]) * s + beta[n];
}

// GEGLU weight order (gemm_dma.hip, EP = 6): output row j of the interleaved matrix is value row 32 b + i of the Linear for j = 64 b + i (i < 32) and
// gate row N/2 + 32 b + i for j = 64 b + 32 + i; the bias likewise
__global__ void pack_geglu_kernel(const float *__restrict__ w, const float *__restrict__ b, float *__restrict__ wi, float *__restrict__ bi, int N, int K) {
    const long long total = (long long)N * K;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(idx / K), k = (int)(idx - (long long)j * K);
        const int blk = j >> 6, i = j & 63;
        const int src = i < 32 ? 32 * blk + i : N / 2 + 32 * blk + (i - 32);
        wi[idx] = w[(long long)src * K + k];
        if (k == 0) bi[j] = b[src];
    }
}

}  // namespace

int pack_geglu(const float *w, const float *b, float *wi, float *bi, int N, int K, hipStream_t st) {
    EDV_CHECK(w && b && wi && bi && N > 0 && K > 0 && N % 64 == 0, "pack_geglu: N (both halves) must be a multiple of 64");
    EDV_LAUNCH(pack_geglu_kernel, dim3(blocks_for((long long)N * K)), dim3(256), 0, st, w, b, wi, bi, N, K);
    EDV_LAUNCH_OK();
    return 0;
}

int pack_conv3x3(const float *w, float *out, int Cout, int Cin, hipStream_t st) {
    EDV_CHECK(w && out && Cout > 0 && Cin > 0, "bad operand");
    EDV_LAUNCH(pack_conv3x3_kernel, dim3(blocks_for((long long)Cout * Cin * 9)), dim3(256), 0, st, w, out, Cout, Cin);
    EDV_LAUNCH_OK();
    return 0;
}

int pack_convT(const float *w, float *wout, const float *b, float *bout, int Cin, int Cout, int s, hipStream_t st) {
    EDV_CHECK(w && wout && b && bout && Cin > 0 && Cout > 0 && s > 0, "bad operand");
    EDV_LAUNCH(pack_convT_kernel, dim3(blocks_for((long long)s * s * Cout * Cin)), dim3(256), 0, st, w, wout, b, bout, Cin, Cout, s);
    EDV_LAUNCH_OK();
    return 0;
}

int copy_f32(const float *src, float *dst, long long n, hipStream_t st) {
    EDV_CHECK(src && dst && n > 0, "bad operand");
    const long long b = (n + 255) / 256;
    EDV_LAUNCH(copy_kernel, dim3((unsigned)(b < 8192 ? b : 8192)), dim3(256), 0, st, src, dst, n);
    EDV_LAUNCH_OK();
    return 0;
}

int fold_lora(const float *W, const float *A, const float *B, const float *U, const float *V, float scale, float *out, int nout, int nin, int r,
              hipStream_t st) {
    EDV_CHECK(W && A && B && out && nout > 0 && nin > 0 && r > 0, "bad operand");
    EDV_LAUNCH(fold_lora_kernel, dim3(blocks_for((long long)nout * nin)), dim3(256), 0, st, W, A, B, U, V, scale, out, nout, nin, r);
    EDV_LAUNCH_OK();
    return 0;
}

int fold_ssb(const float *W, const float *a, const float *b, float *out, int nout, int nin, hipStream_t st) {
    EDV_CHECK(W && a && b && out && nout > 0 && nin > 0, "bad operand");
    EDV_LAUNCH(fold_ssb_kernel, dim3(blocks_for((long long)nout * nin)), dim3(256), 0, st, W, a, b, out, nout, nin);
    EDV_LAUNCH_OK();
    return 0;
}

int fold_dash(const float *Utop, const float *idx, const float *Vtop, float *inout, int nout, int nin, int r, hipStream_t st) {
    EDV_CHECK(Utop && idx && Vtop && inout && nout > 0 && nin > 0 && r > 0, "bad operand");
    EDV_LAUNCH(fold_dash_kernel, dim3(blocks_for((long long)nout * nin)), dim3(256), 0, st, Utop, idx, Vtop, inout, nout, nin, r);
    EDV_LAUNCH_OK();
    return 0;
}

int fold_bn(float *w, const float *b, const float *gamma, const float *beta, const float *mean, const float *var, float eps, float *bout, int nout, int K,
            hipStream_t st) {
    EDV_CHECK(w && b && gamma && beta && mean && var && bout && nout > 0 && K > 0, "bad operand");
    EDV_LAUNCH(fold_bn_kernel, dim3(blocks_for((long long)nout * K)), dim3(256), 0, st, w, b, gamma, beta, mean, var, eps, bout, nout, K);
    EDV_LAUNCH_OK();
    return 0;
}

}  // namespace edv
