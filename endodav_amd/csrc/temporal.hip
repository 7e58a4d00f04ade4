// Temporal-attention core and GEGLU of the motion modules (HBM/L2-bound, tiny FLOPs).
//   motion_module.py:230-297: tokens regrouped "(b f) d c -> (b d) f c", so each PIXEL attends over its T
//   frames; attention.py:182-211: softmax(q kᵀ * d^-0.5) v with 8 heads; attention.py:363-384: GEGLU.
// The regrouping is never materialised: the fused q|k|v rows stay in the channels-last token order
// [(b*T + t)*P + p, 3C] and the frame axis is reached by address stride P*3C.
#include "ops.hpp"

namespace edv {
namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

// one thread per (clip b, query frame tq, pixel p, head); heads fastest so that a wave reads
// 8 pixels x 8 heads x d = 8 contiguous C-float rows.
template <int TMAX>
__global__ __launch_bounds__(256) void attn_temporal_kernel(const float *__restrict__ qkv, float *__restrict__ out, int B, int T, int P, int C,
                                                             int heads, float scale) {
    const long long total = (long long)B * T * P * heads;
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= total) return;
    const int head = (int)(gid % heads);
    long long r = gid / heads;
    const int p = (int)(r % P);
    r /= P;
    const int tq = (int)(r % T);
    const int b = (int)(r / T);
    const int d = C / heads, C3 = 3 * C;
    const long long tstride = (long long)P * C3;
    const float *qp = qkv + ((long long)(b * T + tq) * P + p) * C3 + head * d;
    const float *kb = qkv + ((long long)(b * T) * P + p) * C3 + C + head * d;
    const float *vb = kb + C;

    float s[TMAX];
#pragma unroll
    for (int ts = 0; ts < TMAX; ++ts) s[ts] = 0.f;
    for (int c = 0; c < d; c += 4) {
        const f32x4 q4 = *reinterpret_cast<const f32x4 *>(qp + c);
#pragma unroll
        for (int ts = 0; ts < TMAX; ++ts) {
            if (ts < T) {
                const f32x4 k4 = *reinterpret_cast<const f32x4 *>(kb + ts * tstride + c);
                s[ts] += (q4.x * k4.x + q4.y * k4.y) + (q4.z * k4.z + q4.w * k4.w);
            }
        }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int ts = 0; ts < TMAX; ++ts)
        if (ts < T) {
            s[ts] *= scale;
            mx = fmaxf(mx, s[ts]);
        }
    float sum = 0.f;
#pragma unroll
    for (int ts = 0; ts < TMAX; ++ts)
        if (ts < T) {
            s[ts] = expf(s[ts] - mx);
            sum += s[ts];
        }
    const float inv = 1.0f / sum;
    float *op = out + ((long long)(b * T + tq) * P + p) * C + head * d;
    for (int c = 0; c < d; c += 4) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ts = 0; ts < TMAX; ++ts)
            if (ts < T) acc += s[ts] * *reinterpret_cast<const f32x4 *>(vb + ts * tstride + c);
        *reinterpret_cast<f32x4 *>(op + c) = acc * inv;
    }
}

__global__ __launch_bounds__(256) void geglu_kernel(const float *__restrict__ x, float *__restrict__ y, long long total4, int inner4) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long long)gridDim.x * 256) {
        const long long m = i / inner4;
        const int j = (int)(i - m * inner4);
        const float *row = x + m * (long long)inner4 * 8;
        const f32x4 a = *reinterpret_cast<const f32x4 *>(row + 4 * j);
        const f32x4 g = *reinterpret_cast<const f32x4 *>(row + 4 * (inner4 + j));
        f32x4 o = {a.x * gelu_erf(g.x), a.y * gelu_erf(g.y), a.z * gelu_erf(g.z), a.w * gelu_erf(g.w)};
        *reinterpret_cast<f32x4 *>(y + i * 4) = o;
    }
}

}  // namespace

int attn_temporal(const float *qkv, float *out, int B, int T, int P, int C, int heads, hipStream_t st) {
    EDV_CHECK(qkv && out, "null operand");
    EDV_CHECK(B > 0 && T > 0 && P > 0 && C > 0 && heads > 0, "empty problem");
    EDV_CHECK(T <= 32, "T > 32 is not supported by the temporal attention kernel");
    EDV_CHECK(C % heads == 0 && (C / heads) % 4 == 0, "head dim must be a multiple of 4");
    const long long total = (long long)B * T * P * heads;
    const long long blocks = (total + 255) / 256;
    EDV_CHECK(blocks < (1ll << 31), "grid");
    const float scale = 1.0f / sqrtf((float)(C / heads));
    dim3 grid((unsigned)blocks), block(256);
    if (T <= 8)
        hipLaunchKernelGGL(attn_temporal_kernel<8>, grid, block, 0, st, qkv, out, B, T, P, C, heads, scale);
    else if (T <= 16)
        hipLaunchKernelGGL(attn_temporal_kernel<16>, grid, block, 0, st, qkv, out, B, T, P, C, heads, scale);
    else
        hipLaunchKernelGGL(attn_temporal_kernel<32>, grid, block, 0, st, qkv, out, B, T, P, C, heads, scale);
    EDV_LAUNCH_OK();
    return 0;
}

int geglu(const float *x, float *y, long long M, int inner, hipStream_t st) {
    EDV_CHECK(x && y, "null operand");
    EDV_CHECK(M > 0 && inner > 0 && inner % 4 == 0, "shape");
    const long long total4 = M * (inner / 4);
    const int blocks = (int)((total4 + 255) / 256 < 8192 ? (total4 + 255) / 256 : 8192);
    hipLaunchKernelGGL(geglu_kernel, dim3(blocks), dim3(256), 0, st, x, y, total4, inner / 4);
    EDV_LAUNCH_OK();
    return 0;
}

}  // namespace edv
