// Temporal-attention core and GEGLU of the motion modules (HBM/L2-bound, tiny FLOPs).
//   motion_module.py:230-297: tokens regrouped "(b f) d c -> (b d) f c", so each PIXEL attends over its T
//   frames; attention.py:182-211: softmax(q kᵀ * d^-0.5) v with 8 heads; attention.py:363-384: GEGLU.
// The regrouping is never materialised: the fused q|k|v rows stay in the channels-last token order
// [(b*T + t)*P + p, 3C] and the frame axis is reached by address stride P*3C.
#include <cstdlib>

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

// T <= 8: one thread per (clip, pixel, head) computes ALL T queries, so every q / k / v element is read from memory once
// (the per-query kernel above re-reads each K/V row T times through L1/L2: 51 us per call at T=8, 7.7x the HBM time of the
// tensors).  Scores live in T x T registers; two sweeps over the head dimension in float4 steps.
template <int TT>
__global__ __launch_bounds__(256) void attn_temporal_small_kernel(const float *__restrict__ qkv, float *__restrict__ out, int B, int T, int P, int C, int heads,
                                                                   float scale) {
    const long long total = (long long)B * P * heads;
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= total) return;
    const int head = (int)(gid % heads);
    long long r = gid / heads;
    const int p = (int)(r % P);
    const int b = (int)(r / P);
    const int d = C / heads, C3 = 3 * C;
    const long long ts3 = (long long)P * C3, ts1 = (long long)P * C;
    const float *qb = qkv + ((long long)(b * T) * P + p) * C3 + head * d;
    const float *kb = qb + C, *vb = qb + 2 * C;
    float *ob = out + ((long long)(b * T) * P + p) * C + head * d;
    float S[TT][TT];
#pragma unroll
    for (int i = 0; i < TT; ++i)
#pragma unroll
        for (int j = 0; j < TT; ++j) S[i][j] = 0.f;
#pragma unroll 2
    for (int c = 0; c < d; c += 4) {
        f32x4 q4[TT], k4[TT];
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            const int tc = t < T ? t : T - 1;
            q4[t] = *reinterpret_cast<const f32x4 *>(qb + tc * ts3 + c);
            k4[t] = *reinterpret_cast<const f32x4 *>(kb + tc * ts3 + c);
        }
#pragma unroll
        for (int i = 0; i < TT; ++i)
#pragma unroll
            for (int j = 0; j < TT; ++j) S[i][j] += (q4[i].x * k4[j].x + q4[i].y * k4[j].y) + (q4[i].z * k4[j].z + q4[i].w * k4[j].w);
    }
#pragma unroll
    for (int i = 0; i < TT; ++i) {
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < TT; ++j)
            if (j < T) {
                S[i][j] *= scale;
                mx = fmaxf(mx, S[i][j]);
            }
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < TT; ++j) {
            S[i][j] = j < T ? expf(S[i][j] - mx) : 0.f;
            sum += S[i][j];
        }
        const float inv = 1.0f / sum;
#pragma unroll
        for (int j = 0; j < TT; ++j) S[i][j] *= inv;
    }
#pragma unroll 2
    for (int c = 0; c < d; c += 4) {
        f32x4 v4[TT];
#pragma unroll
        for (int t = 0; t < TT; ++t) v4[t] = *reinterpret_cast<const f32x4 *>(vb + (t < T ? t : T - 1) * ts3 + c);
#pragma unroll
        for (int i = 0; i < TT; ++i) {
            if (i < T) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < TT; ++j) acc += S[i][j] * v4[j];
                *reinterpret_cast<f32x4 *>(ob + i * ts1 + c) = acc;
            }
        }
    }
}

// One workgroup per (clip, pixel, group of HG heads): the T rows of that pixel's q|k|v slice are fetched with ONE round of
// coalesced loads into LDS (T * 3 * HG * d floats); thread (query tq, head) then works out of LDS.  The per-thread kernels
// above chain dozens of dependent load batches per thread when d = 48 and have only B * P * 8 threads to hide them behind
// (50 us per call for 18 MB of traffic at T = 8; 795 us per call at ViT-B T = 16 with the per-query kernel).
template <int TMAX>
__global__ __launch_bounds__(256) void attn_temporal_pixel_kernel(const float *__restrict__ qkv, float *__restrict__ out, int T, int P, int C, int heads,
                                                                   int HG, float scale) {
    extern __shared__ __attribute__((aligned(16))) float sm[];  // [T][3][HG * d]
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int ngroups = heads / HG;
    const int hg = blockIdx.x % ngroups;
    const long long bp = blockIdx.x / ngroups;
    const int b = (int)(bp / P), p = (int)(bp - (long long)b * P);
    const int d = C / heads, C3 = 3 * C, W = HG * d, w4 = W >> 2;
    for (int idx = tid; idx < T * 3 * w4; idx += nthr) {
        const int t = idx / (3 * w4), r = idx - t * 3 * w4;
        const int part = r / w4, c4 = r - part * w4;
        *reinterpret_cast<f32x4 *>(&sm[(t * 3 + part) * W + 4 * c4]) =
            *reinterpret_cast<const f32x4 *>(qkv + ((long long)(b * T + t) * P + p) * C3 + part * C + hg * W + 4 * c4);
    }
    __syncthreads();
    const int hl = tid % HG, tq = tid / HG;
    if (tq >= T) return;
    const float *q = sm + (tq * 3 + 0) * W + hl * d;
    float s[TMAX];
#pragma unroll
    for (int ts = 0; ts < TMAX; ++ts) s[ts] = 0.f;
    for (int c = 0; c < d; c += 4) {
        const f32x4 q4 = *reinterpret_cast<const f32x4 *>(q + c);
#pragma unroll
        for (int ts = 0; ts < TMAX; ++ts)
            if (ts < T) {
                const f32x4 k4 = *reinterpret_cast<const f32x4 *>(sm + (ts * 3 + 1) * W + hl * d + c);
                s[ts] += (q4.x * k4.x + q4.y * k4.y) + (q4.z * k4.z + q4.w * k4.w);
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
    for (int ts = 0; ts < TMAX; ++ts) {
        s[ts] = ts < T ? expf(s[ts] - mx) : 0.f;
        sum += s[ts];
    }
    const float inv = 1.0f / sum;
    float *op = out + ((long long)(b * T + tq) * P + p) * C + (hg * HG + hl) * d;
    for (int c = 0; c < d; c += 4) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ts = 0; ts < TMAX; ++ts)
            if (ts < T) acc += s[ts] * *reinterpret_cast<const f32x4 *>(sm + (ts * 3 + 2) * W + hl * d + c);
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


// Rotary position embedding of the temporal attention (pe="rope": motion_module.py:221-225,252-255; attention.py:402-429).
// q and k of row m (frame t = (m / P) % T) are rotated in place, channel pairs (2i, 2i+1) by the angle t * freq_i whose
// cos|sin the host tabulated with the reference's own expression (table [>=T, C/2, 2]).  The rotation spans the full channel
// width C (before the head split).  sign = -1 applies the transpose = the input gradient of the rotation.
__global__ __launch_bounds__(256) void rope_qk_kernel(float *__restrict__ qkv, const float *__restrict__ table, long long n_pairs, int T, int P, int C,
                                                      float sign) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_pairs) return;
    const int half = C >> 1;
    const long long m = i / half;
    const int j = (int)(i - m * half);
    const int t = (int)((m / P) % T);
    const float2 cs = *reinterpret_cast<const float2 *>(table + ((size_t)t * half + j) * 2);
    const float c = cs.x, sn = sign * cs.y;
    float2 *q = reinterpret_cast<float2 *>(qkv + m * 3 * C + 2 * j), *k = reinterpret_cast<float2 *>(qkv + m * 3 * C + C + 2 * j);
    const float2 a = *q, b = *k;
    *q = make_float2(a.x * c - a.y * sn, a.x * sn + a.y * c);
    *k = make_float2(b.x * c - b.y * sn, b.x * sn + b.y * c);
}

}  // namespace

int attn_temporal(const float *qkv, float *out, int B, int T, int P, int C, int heads, hipStream_t st) {
    EDV_CHECK(qkv && out, "null operand");
    EDV_CHECK(B > 0 && T > 0 && P > 0 && C > 0 && heads > 0, "empty problem");
    EDV_CHECK(T <= 64, "T > 64 frames per clip is not built (temporal attention keeps a query's T scores in registers)");
    EDV_CHECK(C % heads == 0 && (C / heads) % 4 == 0, "head dim must be a multiple of 4");
    const long long total = (long long)B * T * P * heads;
    const long long blocks = (total + 255) / 256;
    EDV_CHECK(blocks < (1ll << 31), "grid");
    const float scale = 1.0f / sqrtf((float)(C / heads));
    dim3 grid((unsigned)blocks), block(256);
    static const bool per_query = [] {
        const char *e = getenv("EDV_TATTN_PER_QUERY");  // 1: the one-thread-per-query kernel also for T <= 8 (A/B runs)
        return e && atoi(e) != 0;
    }();
    // pixel-per-workgroup kernel: wide heads at T <= 8, everything at T > 8 (the all-queries-per-thread kernel needs T x T registers)
    const int d = C / heads;
    int HG = heads;  // heads per workgroup: as many as fit 256 threads and 64 KB of LDS
    while (HG > 1 && ((long long)T * HG > 256 || (size_t)T * 3 * HG * d * sizeof(float) > 64 * 1024 || heads % HG != 0)) --HG;
    // one head of a long clip may need more than 64 KB (T = 100 at d = 128: 150 KB): the CU has 160
    const size_t lds_cap = (T > 32 && HG == 1) ? 160 * 1024 : 64 * 1024;
    const bool pixel_fits = (long long)T * HG <= 256 && (size_t)T * 3 * HG * d * sizeof(float) <= lds_cap && (long long)B * P * (heads / HG) < (1ll << 31);
    if (!per_query && pixel_fits && (T > 8 || d >= 24)) {
        const dim3 g3((unsigned)((long long)B * P * (heads / HG)));
        const dim3 b3((unsigned)(((T * HG + 63) / 64) * 64));
        const size_t lds = (size_t)T * 3 * HG * d * sizeof(float);
        if (T <= 8)
            EDV_LAUNCH(attn_temporal_pixel_kernel<8>, g3, b3, lds, st, qkv, out, T, P, C, heads, HG, scale);
        else if (T <= 16)
            EDV_LAUNCH(attn_temporal_pixel_kernel<16>, g3, b3, lds, st, qkv, out, T, P, C, heads, HG, scale);
        else if (T <= 32)
            EDV_LAUNCH(attn_temporal_pixel_kernel<32>, g3, b3, lds, st, qkv, out, T, P, C, heads, HG, scale);
        else {  // 32 < T <= 64: num_frames > 32 (dpt_temporal.py:35-40 takes any; the reference's scripts keep the default 32)
            if (lds > 64 * 1024)
                EDV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(attn_temporal_pixel_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_cap));
            EDV_LAUNCH(attn_temporal_pixel_kernel<64>, g3, b3, lds, st, qkv, out, T, P, C, heads, HG, scale);
        }
    } else if (T > 32) {
        EDV_CHECK(false, "temporal attention over more than 32 frames: one head's q|k|v of a pixel (T x 3 x C/8 floats) must fit the 160 KB of LDS");
    } else if (T <= 8 && !per_query) {
        const long long tot = (long long)B * P * heads;
        const dim3 g2((unsigned)((tot + 255) / 256));
        if (T <= 4)
            EDV_LAUNCH(attn_temporal_small_kernel<4>, g2, block, 0, st, qkv, out, B, T, P, C, heads, scale);
        else
            EDV_LAUNCH(attn_temporal_small_kernel<8>, g2, block, 0, st, qkv, out, B, T, P, C, heads, scale);
    } else if (T <= 8)
        EDV_LAUNCH(attn_temporal_kernel<8>, grid, block, 0, st, qkv, out, B, T, P, C, heads, scale);
    else if (T <= 16)
        EDV_LAUNCH(attn_temporal_kernel<16>, grid, block, 0, st, qkv, out, B, T, P, C, heads, scale);
    else
        EDV_LAUNCH(attn_temporal_kernel<32>, grid, block, 0, st, qkv, out, B, T, P, C, heads, scale);
    EDV_LAUNCH_OK();
    return 0;
}

int geglu(const float *x, float *y, long long M, int inner, hipStream_t st) {
    EDV_CHECK(x && y, "null operand");
    EDV_CHECK(M > 0 && inner > 0 && inner % 4 == 0, "shape");
    const long long total4 = M * (inner / 4);
    const int blocks = (int)((total4 + 255) / 256 < 8192 ? (total4 + 255) / 256 : 8192);
    EDV_LAUNCH(geglu_kernel, dim3(blocks), dim3(256), 0, st, x, y, total4, inner / 4);
    EDV_LAUNCH_OK();
    return 0;
}

int rope_qk(float *qkv, const float *table, int B, int T, int P, int C, bool transpose, hipStream_t st) {
    EDV_CHECK(qkv && table, "null operand");
    EDV_CHECK(B > 0 && T > 0 && P > 0 && C > 0 && C % 2 == 0, "bad shape");
    const long long n = (long long)B * T * P * (C / 2);
    EDV_LAUNCH(rope_qk_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, qkv, table, n, T, P, C, transpose ? -1.f : 1.f);
    EDV_LAUNCH_OK();
    return 0;
}

}  // namespace edv
