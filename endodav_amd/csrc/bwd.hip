// Backward (dX) kernels of the HBM-bound operators, plus the low-rank LoRA weight-gradient products.
// the fine-tune step trains only the LoRA factors (endodav/layers.py:5-34), so every frozen
// operator needs its input gradient and nothing else.  Layouts are the forward's: channels-last activations, fused
// q|k|v rows, rows = (frame, token).  Each kernel cites the forward it differentiates.
#include <cstdlib>

#include "ops.hpp"

// The bilinear adjoint must reproduce the forward's interpolation weights: ATen computes the source coordinate as a
// ROUNDED float product and then subtracts its integer part; contraction would fuse the two (resample.hip has the same
// pragma for the same reason).  Everything in this file is HBM-bound, so nothing is lost.
#pragma clang fp contract(off)

namespace edv {
namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

// d/dx of 0.5 x (1 + erf(x / sqrt 2)):  Phi(x) + x phi(x)
__device__ __forceinline__ float gelu_erf_grad(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}

// ---- LayerNorm backward (norms.hip layernorm_kernel): one wave per row, dim <= 1024 ------------------------------------
constexpr int LNB_MAXV = 4;  // float4 chunks per lane
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float *__restrict__ x, RowMap xmap, const float *__restrict__ w,
                                                            const float *__restrict__ dy, RowMap dymap, float *__restrict__ dx, RowMap dxmap,
                                                            long long rows, int dim, float eps, int accumulate) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float *xr = x + xmap(row) * dim;
    const float *gr = dy + dymap(row) * dim;
    float *dr = dx + dxmap(row) * dim;
    const int nv = dim >> 2;
    f32x4 xv[LNB_MAXV], gv[LNB_MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LNB_MAXV; ++i) {
        const int c = lane + 64 * i;
        xv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        gv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (c < nv) {
            xv[i] = *reinterpret_cast<const f32x4 *>(xr + 4 * c);
            const f32x4 g = *reinterpret_cast<const f32x4 *>(gr + 4 * c);
            const f32x4 ww = *reinterpret_cast<const f32x4 *>(w + 4 * c);
            gv[i] = g * ww;
            s += (xv[i].x + xv[i].y) + (xv[i].z + xv[i].w);
        }
    }
    const float mean = wave_sum(s) / (float)dim;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LNB_MAXV; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
            xv[i] -= mean;
            q += (xv[i].x * xv[i].x + xv[i].y * xv[i].y) + (xv[i].z * xv[i].z + xv[i].w * xv[i].w);
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)dim + eps);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < LNB_MAXV; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
            xv[i] *= rstd;  // x-hat
            s1 += (gv[i].x + gv[i].y) + (gv[i].z + gv[i].w);
            s2 += (gv[i].x * xv[i].x + gv[i].y * xv[i].y) + (gv[i].z * xv[i].z + gv[i].w * xv[i].w);
        }
    }
    s1 = wave_sum(s1) / (float)dim;
    s2 = wave_sum(s2) / (float)dim;
#pragma unroll
    for (int i = 0; i < LNB_MAXV; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
            f32x4 o = (gv[i] - s1 - xv[i] * s2) * rstd;
            if (accumulate) o += *reinterpret_cast<const f32x4 *>(dr + 4 * c);
            *reinterpret_cast<f32x4 *>(dr + 4 * c) = o;
        }
    }
}

// ---- elementwise ---------------------------------------------------------------------------------------------------------
// out = (mode 1: d * gelu'(src) | mode 2: src > 0 ? d : 0 | mode 3: gelu(d) | mode 0: d) + (add ? add : 0)
__global__ __launch_bounds__(256) void ew_bwd_kernel(const float *__restrict__ d, const float *__restrict__ src, const float *__restrict__ add,
                                                     float *__restrict__ out, long long n4, int mode) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        f32x4 v = *reinterpret_cast<const f32x4 *>(d + 4 * i);
        if (mode == 3) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
        } else if (mode) {
            const f32x4 sv = *reinterpret_cast<const f32x4 *>(src + 4 * i);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = mode == 1 ? v[e] * gelu_erf_grad(sv[e]) : (sv[e] > 0.f ? v[e] : 0.f);
        }
        if (add) v += *reinterpret_cast<const f32x4 *>(add + 4 * i);
        *reinterpret_cast<f32x4 *>(out + 4 * i) = v;
    }
}

// out = g * s * (1 - s): through y = sigmoid(z) given y (out_sigmoid of the VDA head, dpt_pyramid.py:97-101); any n
__global__ __launch_bounds__(256) void sigmoid_bwd_kernel(const float *__restrict__ g, const float *__restrict__ s, float *__restrict__ out, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float y = s[i];
        out[i] = g[i] * (y * (1.f - y));
    }
}

// GEGLU backward (temporal.hip geglu_kernel): y = a * gelu(g);  x rows are [a (inner) | g (inner)]
__global__ __launch_bounds__(256) void geglu_bwd_kernel(const float *__restrict__ x, const float *__restrict__ dy, float *__restrict__ dx, long long total4,
                                                        int inner4) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long long)gridDim.x * 256) {
        const long long m = i / inner4;
        const int j = (int)(i - m * inner4);
        const long long ro = m * (long long)inner4 * 8;
        const f32x4 a = *reinterpret_cast<const f32x4 *>(x + ro + 4 * j);
        const f32x4 g = *reinterpret_cast<const f32x4 *>(x + ro + 4 * (inner4 + j));
        const f32x4 d = *reinterpret_cast<const f32x4 *>(dy + i * 4);
        f32x4 da, dg;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            da[e] = d[e] * gelu_erf(g[e]);
            dg[e] = d[e] * a[e] * gelu_erf_grad(g[e]);
        }
        *reinterpret_cast<f32x4 *>(dx + ro + 4 * j) = da;
        *reinterpret_cast<f32x4 *>(dx + ro + 4 * (inner4 + j)) = dg;
    }
}

// Wt[k, n] = W[n, k] * (gamma ? gamma[n] : 1): the weight of the dX GEMM  dX = (dY * gamma) W  in the NT form gemm() takes
__global__ __launch_bounds__(256) void transpose_scale_kernel(const float *__restrict__ W, const float *__restrict__ gamma, float *__restrict__ Wt, int N,
                                                              int K, int ldw) {
    __shared__ float tile[32][33];
    const int n0 = blockIdx.y * 32, k0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int n = n0 + r, k = k0 + tx;
        tile[r][tx] = (n < N && k < K) ? W[(long long)n * ldw + k] * (gamma ? gamma[n] : 1.f) : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int k = k0 + r, n = n0 + tx;
        if (k < K && n < N) Wt[(long long)k * N + n] = tile[tx][r];
    }
}

// ---- low-rank products of the LoRA weight gradients ----------------------------------------------------------------------
// T[m, j] = sum_k X[m, k] * Wr[j, k],  j < R <= 8: one wave per row
template <int R>
__global__ __launch_bounds__(256) void skinny_xwt_kernel(const float *__restrict__ X, long long M, int K, int ldx, const float *__restrict__ Wr,
                                                         float *__restrict__ T) {
    const int lane = threadIdx.x & 63;
    const long long m = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const float *xr = X + m * ldx;
    float acc[R];
#pragma unroll
    for (int j = 0; j < R; ++j) acc[j] = 0.f;
    for (int k = lane * 4; k < K; k += 256) {
        const f32x4 xv = *reinterpret_cast<const f32x4 *>(xr + k);
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const f32x4 wv = *reinterpret_cast<const f32x4 *>(Wr + (long long)j * K + k);
            acc[j] += (xv.x * wv.x + xv.y * wv.y) + (xv.z * wv.z + xv.w * wv.w);
        }
    }
#pragma unroll
    for (int j = 0; j < R; ++j) acc[j] = wave_sum(acc[j]);
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < R; ++j) T[m * R + j] = acc[j];
    }
}

// part[s, n, j] = sum_{m in split s} Y[m, n] * T[m, j]:  thread = column n, 4 waves stride the rows of the split
template <int R>
__global__ __launch_bounds__(256) void tall_tn_partial_kernel(const float *__restrict__ Y, int ldy, const float *__restrict__ T, long long M, int N,
                                                              float *__restrict__ part, int rows_per_split) {
    __shared__ float red[4][64][R];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + lane;
    const long long mb = (long long)blockIdx.y * rows_per_split;
    const long long me = mb + rows_per_split < M ? mb + rows_per_split : M;
    float acc[R];
#pragma unroll
    for (int j = 0; j < R; ++j) acc[j] = 0.f;
    if (n < N)
        for (long long m = mb + wv; m < me; m += 4) {
            const float y = Y[m * ldy + n];
#pragma unroll
            for (int j = 0; j < R; ++j) acc[j] += y * T[m * R + j];
        }
#pragma unroll
    for (int j = 0; j < R; ++j) red[wv][lane][j] = acc[j];
    __syncthreads();
    if (wv == 0 && n < N) {
#pragma unroll
        for (int j = 0; j < R; ++j)
            part[((long long)blockIdx.y * N + n) * R + j] = (red[0][lane][j] + red[1][lane][j]) + (red[2][lane][j] + red[3][lane][j]);
    }
}
// out[n, j] = scale * (rowscale ? rowscale[n] : 1) * sum_s part[s, n, j].  A workgroup = 32 consecutive elements x 8 slices of the
// split range; each thread adds its slice in split order, the slices are added in slice order: a fixed order, hence deterministic
// (one thread per element walking all splits took 16 us per call on a chain of dependent loads).
__global__ __launch_bounds__(256) void tall_tn_reduce_kernel(const float *__restrict__ part, int splits, int N, int R, float scale,
                                                             const float *__restrict__ rowscale, float *__restrict__ out) {
    __shared__ float red[8][32];
    const int e = threadIdx.x & 31, slice = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + e;
    float s = 0.f;
    if (i < N * R) {
        const int per = (splits + 7) / 8, s0 = slice * per, s1 = s0 + per < splits ? s0 + per : splits;
#pragma unroll 8
        for (int sp = s0; sp < s1; ++sp) s += part[(long long)sp * N * R + i];
    }
    red[slice][e] = s;
    __syncthreads();
    if (slice == 0 && i < N * R) {
        float t = red[0][e];
#pragma unroll
        for (int k = 1; k < 8; ++k) t += red[k][e];
        out[i] = t * scale * (rowscale ? rowscale[i / R] : 1.f);
    }
}

// LoRA factor gradients from dBp [out, r] (= s dY^T (X A'^T)) and dApT [in, r] (= s X^T (dY B')), mylora/layers.py:148-157, 384-393.
//   lora:   dB = dBp,  dA[j, k] = dApT[k, j]
//   dvlora: A' = A * U with U [r, 1], B' = B * V with V [out, 1]:
//           dB[n, j] = dBp[n, j] V[n],  dV[n] = sum_j dBp[n, j] B[n, j],  dA[j, k] = dApT[k, j] U[j],  dU[j] = sum_k dApT[k, j] A[j, k]
__global__ __launch_bounds__(256) void lora_grad_finalize_kernel(const float *__restrict__ dBp, const float *__restrict__ dApT, const float *__restrict__ A,
                                                                 const float *__restrict__ Bm, const float *__restrict__ U, const float *__restrict__ V,
                                                                 float *__restrict__ dA, float *__restrict__ dB, float *__restrict__ dV, int nout, int nin,
                                                                 int r) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < nout * r && dB) dB[i] = dBp[i] * (V ? V[i / r] : 1.f);
    if (i < nout && V && dV) {
        float s = 0.f;
        for (int j = 0; j < r; ++j) s += dBp[i * r + j] * Bm[i * r + j];
        dV[i] = s;
    }
    if (i < r * nin && dA) {
        const int j = i / nin, k = i - j * nin;
        dA[i] = dApT[k * r + j] * (U ? U[j] : 1.f);
    }
}
// dU[j] = sum_k dApT[k, j] A[j, k]: one workgroup per j
__global__ __launch_bounds__(256) void lora_grad_u_kernel(const float *__restrict__ dApT, const float *__restrict__ A, float *__restrict__ dU, int nin, int r) {
    __shared__ float red[4];
    const int j = blockIdx.x;
    float s = 0.f;
    for (int k = threadIdx.x; k < nin; k += 256) s += dApT[k * r + j] * A[(long long)j * nin + k];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) dU[j] = (red[0] + red[1]) + (red[2] + red[3]);
}

// ---- bilinear, align_corners=True, backward as a deterministic gather (resample.hip bilinear_*) ---------------------------
// forward: src = (in-1)/(out-1) * o (rounded float product), i0 = floor(src), lam = src - i0, i1 = min(i0+1, in-1)
__device__ __forceinline__ void bl_src(int o, float ratio, int in, int out, int &i0, int &i1, float &lam) {
    if (in == out) {
        i0 = i1 = o;
        lam = 0.f;
        return;
    }
    // the forward's ROUNDED product (resample.hip lin_coord).  Both operations are written here, under this file's
    // contract(off): the header's __fmul_rn / __fsub_rn bodies carry the contract flag and would fuse with each other.
    const float s = ratio * (float)o;
    i0 = (int)s;
    i1 = i0 + (i0 < in - 1 ? 1 : 0);
    lam = fminf(fmaxf(s - (float)i0, 0.f), 1.f);
}
// weight of output coordinate o on input coordinate i
__device__ __forceinline__ float bl_weight(int o, int i, float ratio, int in, int out) {
    int i0, i1;
    float lam;
    bl_src(o, ratio, in, out, i0, i1, lam);
    float w = 0.f;
    if (i0 == i) w += 1.f - lam;
    if (i1 == i) w += lam;
    return w;
}
// one thread per (frame, iy, ix, channel chunk of 4 or 1)
template <int V>
__global__ __launch_bounds__(256) void bilinear_bwd_kernel(const float *__restrict__ dy, float *__restrict__ dx, int F, int ih, int iw, int C, int oh, int ow,
                                                           float ry, float rx, int accumulate) {
    const int cv = C / V;
    const long long total = (long long)F * ih * iw * cv;
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= total) return;
    const int c = (int)(gid % cv) * V;
    long long r = gid / cv;
    const int ix = (int)(r % iw);
    r /= iw;
    const int iy = (int)(r % ih);
    const int f = (int)(r / ih);
    // candidate output range: src in (i-1, i+1)  <=>  o in ((i-1)/ratio, (i+1)/ratio); widened by one and tested exactly
    auto range = [](int i, float ratio, int out, int &lo, int &hi) {
        if (ratio <= 0.f) {  // in == 1 or out == 1: every output reads input 0
            lo = 0;
            hi = out - 1;
            return;
        }
        lo = (int)floorf((float)(i - 1) / ratio) - 1;
        hi = (int)ceilf((float)(i + 1) / ratio) + 1;
        lo = lo < 0 ? 0 : lo;
        hi = hi > out - 1 ? out - 1 : hi;
    };
    int ylo, yhi, xlo, xhi;
    range(iy, ry, oh, ylo, yhi);
    range(ix, rx, ow, xlo, xhi);
    float acc[V];
#pragma unroll
    for (int e = 0; e < V; ++e) acc[e] = 0.f;
    for (int oy = ylo; oy <= yhi; ++oy) {
        const float wy = bl_weight(oy, iy, ry, ih, oh);
        if (wy == 0.f) continue;
        for (int ox = xlo; ox <= xhi; ++ox) {
            const float wgt = wy * bl_weight(ox, ix, rx, iw, ow);
            if (wgt == 0.f) continue;
            const float *p = dy + (((long long)f * oh + oy) * ow + ox) * C + c;
#pragma unroll
            for (int e = 0; e < V; ++e) acc[e] += wgt * p[e];
        }
    }
    float *q = dx + (((long long)f * ih + iy) * iw + ix) * C + c;
#pragma unroll
    for (int e = 0; e < V; ++e) q[e] = accumulate ? q[e] + acc[e] : acc[e];
}

// final 1x1 conv to one channel + output activation (resample.hip dot_channels), input o2 is itself post-ReLU:
//   gz[p] = g[p] * act'(.)      mode 0 (VDA head, ReLU):  disp > 0 ? g : 0
//                               mode 1 / 2 (HeadDepth, sigmoid(+-z), dpt_pyramid.py:103-109):  +-g * disp * (1 - disp)
//   d_o2[p, c] = gz[p] * w[c] * (o2[p, c] > 0);   gz_out (optional) keeps gz for the weight / bias gradient of the 1x1 conv
__global__ __launch_bounds__(256) void dot_channels_bwd_kernel(const float *__restrict__ g, const float *__restrict__ disp, const float *__restrict__ w,
                                                               const float *__restrict__ o2, float *__restrict__ d_o2, float *__restrict__ gz_out,
                                                               long long npix, int C, int mode) {
    const int c4n = C >> 2;
    const long long total = npix * c4n;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long p = i / c4n;
        const int c = (int)(i - p * c4n) * 4;
        const float dv = disp[p];
        float gz;
        if (mode == 0) {
            gz = dv > 0.f ? g[p] : 0.f;
        } else {
            const float t = g[p] * (dv * (1.f - dv));
            gz = mode == 1 ? t : -t;
        }
        if (gz_out && c == 0) gz_out[p] = gz;
        const f32x4 o = *reinterpret_cast<const f32x4 *>(o2 + i * 4);
        const f32x4 ww = *reinterpret_cast<const f32x4 *>(w + c);
        f32x4 d;
#pragma unroll
        for (int e = 0; e < 4; ++e) d[e] = o[e] > 0.f ? gz * ww[e] : 0.f;
        *reinterpret_cast<f32x4 *>(d_o2 + i * 4) = d;
    }
}

// ---- GroupNorm backward (norms.hip groupnorm_*): sums per (frame, group), then apply ------------------------------------
__global__ __launch_bounds__(256) void groupnorm_bwd_sums_kernel(const float *__restrict__ x, const float *__restrict__ stats, const float *__restrict__ w,
                                                                 const float *__restrict__ dy, float *__restrict__ sums, int P, int C, int groups) {
    __shared__ float red[2][4];
    const int f = blockIdx.y, g = blockIdx.x, cg = C / groups;
    const long long base = (long long)f * P * C + g * cg;
    const int n = P * cg;
    const float mean = stats[((long long)f * groups + g) * 2], rstd = stats[((long long)f * groups + g) * 2 + 1];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll 8  // L2-resident slab, one dependent load pair per trip: keep several in flight
    for (int i = threadIdx.x; i < n; i += 256) {
        const int p = i / cg, c = i - p * cg;
        const long long o = base + (long long)p * C + c;
        const float gy = dy[o] * w[g * cg + c];
        s1 += gy;
        s2 += gy * (x[o] - mean) * rstd;
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if (lane == 0) {
        red[0][wv] = s1;
        red[1][wv] = s2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        sums[((long long)f * groups + g) * 2 + 0] = ((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) / (float)n;
        sums[((long long)f * groups + g) * 2 + 1] = ((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) / (float)n;
    }
}
__global__ __launch_bounds__(256) void groupnorm_bwd_apply_kernel(const float *__restrict__ x, const float *__restrict__ stats, const float *__restrict__ w,
                                                                  const float *__restrict__ dy, const float *__restrict__ sums, float *__restrict__ dx,
                                                                  long long total4, int P, int C, int groups, int accumulate) {
    const int cg = C / groups, c4n = C >> 2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long long)gridDim.x * 256) {
        const long long pix = i / c4n;
        const int c = (int)(i - pix * c4n) * 4;
        const long long f = pix / P;
        const f32x4 xv = *reinterpret_cast<const f32x4 *>(x + i * 4);
        const f32x4 gv = *reinterpret_cast<const f32x4 *>(dy + i * 4);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int ch = c + e, g = ch / cg;
            const long long sg = (f * groups + g) * 2;
            const float mean = stats[sg], rstd = stats[sg + 1];
            const float xh = (xv[e] - mean) * rstd;
            o[e] = (gv[e] * w[ch] - sums[sg] - xh * sums[sg + 1]) * rstd;
        }
        if (accumulate) o += *reinterpret_cast<const f32x4 *>(dx + i * 4);
        *reinterpret_cast<f32x4 *>(dx + i * 4) = o;
    }
}

// ---- temporal attention backward (temporal.hip attn_temporal_kernel): one thread per (clip, pixel, head), all T queries --
// float4 over the head dimension (d % 4 == 0); dK / dV rows are accumulated in place (this thread owns them).
template <int TMAX>
__global__ __launch_bounds__(64) void attn_temporal_bwd_kernel(const float *__restrict__ qkv, const float *__restrict__ dout, float *__restrict__ dqkv, int B,
                                                               int T, int P, int C, int heads, float scale) {
    const long long total = (long long)B * P * heads;
    const long long gid = (long long)blockIdx.x * 64 + threadIdx.x;
    if (gid >= total) return;
    const int head = (int)(gid % heads);
    long long r = gid / heads;
    const int p = (int)(r % P);
    const int b = (int)(r / P);
    const int d = C / heads, C3 = 3 * C;
    const long long ts3 = (long long)P * C3, ts1 = (long long)P * C;
    const float *qb = qkv + ((long long)(b * T) * P + p) * C3 + head * d;
    const float *kb = qb + C, *vb = qb + 2 * C;
    const float *gb = dout + ((long long)(b * T) * P + p) * C + head * d;
    float *dqb = dqkv + ((long long)(b * T) * P + p) * C3 + head * d;
    float *dkb = dqb + C, *dvb = dqb + 2 * C;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < T; ++t)
        for (int c = 0; c < d; c += 4) {
            *reinterpret_cast<f32x4 *>(dkb + t * ts3 + c) = zero;
            *reinterpret_cast<f32x4 *>(dvb + t * ts3 + c) = zero;
        }
    for (int tq = 0; tq < T; ++tq) {
        float s[TMAX], dp[TMAX];
#pragma unroll
        for (int ts = 0; ts < TMAX; ++ts) s[ts] = dp[ts] = 0.f;
        for (int c = 0; c < d; c += 4) {
            const f32x4 q = *reinterpret_cast<const f32x4 *>(qb + tq * ts3 + c), g = *reinterpret_cast<const f32x4 *>(gb + tq * ts1 + c);
#pragma unroll
            for (int ts = 0; ts < TMAX; ++ts)
                if (ts < T) {
                    const f32x4 k = *reinterpret_cast<const f32x4 *>(kb + ts * ts3 + c), v = *reinterpret_cast<const f32x4 *>(vb + ts * ts3 + c);
                    s[ts] += (q.x * k.x + q.y * k.y) + (q.z * k.z + q.w * k.w);
                    dp[ts] += (g.x * v.x + g.y * v.y) + (g.z * v.z + g.w * v.w);
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
        float dot = 0.f;
#pragma unroll
        for (int ts = 0; ts < TMAX; ++ts)
            if (ts < T) {
                s[ts] *= inv;  // P
                dot += s[ts] * dp[ts];
            }
#pragma unroll
        for (int ts = 0; ts < TMAX; ++ts)
            if (ts < T) dp[ts] = s[ts] * (dp[ts] - dot) * scale;  // dS * scale
        for (int c = 0; c < d; c += 4) {
            const f32x4 q = *reinterpret_cast<const f32x4 *>(qb + tq * ts3 + c), g = *reinterpret_cast<const f32x4 *>(gb + tq * ts1 + c);
            f32x4 dq = zero;
#pragma unroll
            for (int ts = 0; ts < TMAX; ++ts)
                if (ts < T) {
                    dq += dp[ts] * *reinterpret_cast<const f32x4 *>(kb + ts * ts3 + c);
                    f32x4 *dk = reinterpret_cast<f32x4 *>(dkb + ts * ts3 + c), *dv = reinterpret_cast<f32x4 *>(dvb + ts * ts3 + c);
                    *dk += dp[ts] * q;
                    *dv += s[ts] * g;
                }
            *reinterpret_cast<f32x4 *>(dqb + tq * ts3 + c) = dq;
        }
    }
}

// One workgroup per (clip, pixel, group of HG heads), everything out of LDS (the forward's attn_temporal_pixel_kernel).  The
// T rows of that pixel's q|k|v slice and of dO are fetched with one round of coalesced loads; thread (t, head) first acts as
// query t (scores, P, dS, dQ), publishes its P and dS rows, then acts as key t (dK = sum_q dS[q][t] Q[q], dV = sum_q P[q][t] dO[q]).
template <int TMAX>
__global__ __launch_bounds__(256) void attn_temporal_bwd_pixel_kernel(const float *__restrict__ qkv, const float *__restrict__ dout, float *__restrict__ dqkv,
                                                                       int T, int P, int C, int heads, int HG, float scale) {
    extern __shared__ __attribute__((aligned(16))) float sm[];  // [T][3][W] q|k|v, [T][W] dO, [HG][T][TMAX] P, [HG][T][TMAX] dS;  W = HG * d
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int ngroups = heads / HG;
    const int hg = blockIdx.x % ngroups;
    const long long bp = blockIdx.x / ngroups;
    const int b = (int)(bp / P), p = (int)(bp - (long long)b * P);
    const int d = C / heads, C3 = 3 * C, W = HG * d, w4 = W >> 2;
    float *sG = sm + T * 3 * W, *sP = sG + T * W, *sS = sP + HG * T * TMAX;
    for (int idx = tid; idx < T * 3 * w4; idx += nthr) {
        const int t = idx / (3 * w4), r = idx - t * 3 * w4;
        const int part = r / w4, c4 = r - part * w4;
        *reinterpret_cast<f32x4 *>(&sm[(t * 3 + part) * W + 4 * c4]) =
            *reinterpret_cast<const f32x4 *>(qkv + ((long long)(b * T + t) * P + p) * C3 + part * C + hg * W + 4 * c4);
    }
    for (int idx = tid; idx < T * w4; idx += nthr) {
        const int t = idx / w4, c4 = idx - t * w4;
        *reinterpret_cast<f32x4 *>(&sG[t * W + 4 * c4]) = *reinterpret_cast<const f32x4 *>(dout + ((long long)(b * T + t) * P + p) * C + hg * W + 4 * c4);
    }
    __syncthreads();
    const int hl = tid % HG, t = tid / HG;
    const bool live = t < T;
    const int tc = live ? t : T - 1;
    const float *q = sm + (tc * 3 + 0) * W + hl * d, *g = sG + tc * W + hl * d;
    float s[TMAX], dp[TMAX];
#pragma unroll
    for (int ts = 0; ts < TMAX; ++ts) s[ts] = dp[ts] = 0.f;
    for (int c = 0; c < d; c += 4) {
        const f32x4 q4 = *reinterpret_cast<const f32x4 *>(q + c), g4 = *reinterpret_cast<const f32x4 *>(g + c);
#pragma unroll
        for (int ts = 0; ts < TMAX; ++ts)
            if (ts < T) {
                const f32x4 k4 = *reinterpret_cast<const f32x4 *>(sm + (ts * 3 + 1) * W + hl * d + c);
                const f32x4 v4 = *reinterpret_cast<const f32x4 *>(sm + (ts * 3 + 2) * W + hl * d + c);
                s[ts] += (q4.x * k4.x + q4.y * k4.y) + (q4.z * k4.z + q4.w * k4.w);
                dp[ts] += (g4.x * v4.x + g4.y * v4.y) + (g4.z * v4.z + g4.w * v4.w);
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
    float dot = 0.f;
#pragma unroll
    for (int ts = 0; ts < TMAX; ++ts) {
        s[ts] *= inv;  // P
        dot += s[ts] * dp[ts];
    }
#pragma unroll
    for (int ts = 0; ts < TMAX; ++ts) {
        dp[ts] = s[ts] * (dp[ts] - dot) * scale;  // dS * scale (zero for ts >= T)
        if (live) {  // threads with t >= T only keep the barrier company
            sP[(hl * T + t) * TMAX + ts] = s[ts];
            sS[(hl * T + t) * TMAX + ts] = dp[ts];
        }
    }
    float *orow = dqkv + ((long long)(b * T + tc) * P + p) * C3 + hg * W + hl * d;
    if (live)
        for (int c = 0; c < d; c += 4) {  // dQ[t] = sum_ts dS[t][ts] K[ts]
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ts = 0; ts < TMAX; ++ts)
                if (ts < T) acc += dp[ts] * *reinterpret_cast<const f32x4 *>(sm + (ts * 3 + 1) * W + hl * d + c);
            *reinterpret_cast<f32x4 *>(orow + c) = acc;
        }
    __syncthreads();
    if (!live) return;
    // column t of P and dS (over the queries) replaces the rows in the same registers
#pragma unroll
    for (int tq = 0; tq < TMAX; ++tq) {
        s[tq] = tq < T ? sP[(hl * T + tq) * TMAX + t] : 0.f;
        dp[tq] = tq < T ? sS[(hl * T + tq) * TMAX + t] : 0.f;
    }
    for (int c = 0; c < d; c += 4) {
        f32x4 dk = {0.f, 0.f, 0.f, 0.f}, dv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tq = 0; tq < TMAX; ++tq)
            if (tq < T) {
                dk += dp[tq] * *reinterpret_cast<const f32x4 *>(sm + (tq * 3 + 0) * W + hl * d + c);
                dv += s[tq] * *reinterpret_cast<const f32x4 *>(sG + tq * W + hl * d + c);
            }
        *reinterpret_cast<f32x4 *>(orow + C + c) = dk;
        *reinterpret_cast<f32x4 *>(orow + 2 * C + c) = dv;
    }
}

// zero insertion of a stride-2 convolution's output gradient: z[f, 2*oy, 2*ox, :] = dy[f, oy, ox, :], zero elsewhere; the
// stride-2 input gradient is then the stride-1 input-gradient convolution of z (flipped taps)
__global__ __launch_bounds__(256) void dilate2_kernel(const float *__restrict__ dy, float *__restrict__ z, int F, int H, int W, int C, int OH, int OW) {
    const int c4n = C >> 2;
    const long long total = (long long)F * H * W * c4n;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % c4n) * 4;
        long long r = i / c4n;
        const int x = (int)(r % W);
        r /= W;
        const int y = (int)(r % H);
        const int f = (int)(r / H);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (!(y & 1) && !(x & 1) && (y >> 1) < OH && (x >> 1) < OW)
            v = *reinterpret_cast<const f32x4 *>(dy + (((long long)f * OH + (y >> 1)) * OW + (x >> 1)) * C + c);
        *reinterpret_cast<f32x4 *>(z + i * 4) = v;
    }
}

// pixel-unshuffle of a ConvTranspose(k = s) output gradient: dy [F, h*s, w*s, C] -> A [F*h*w, s*s*C] with column
// (dy*s + dx)*C + co, the row order of the packed forward weight (prep.hip pack_convT)
__global__ __launch_bounds__(256) void pixel_unshuffle_kernel(const float *__restrict__ dy, float *__restrict__ A, int F, int h, int w, int C, int s) {
    const int c4n = C >> 2;
    const long long total = (long long)F * h * w * s * s * c4n;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % c4n) * 4;
        long long r = i / c4n;
        const int sub = (int)(r % (s * s));
        r /= (s * s);
        const int x = (int)(r % w);
        r /= w;
        const int y = (int)(r % h);
        const int f = (int)(r / h);
        const int dyy = sub / s, dxx = sub - dyy * s;
        const float *src = dy + ((((long long)f * h * s + (long long)y * s + dyy) * (w * s)) + (long long)x * s + dxx) * C + c;
        *reinterpret_cast<f32x4 *>(A + i * 4) = *reinterpret_cast<const f32x4 *>(src);
    }
}

// 3x3 stride-2 pad-1 convolution, input gradient, direct form (only head.resize_layers.3: a 19x19 -> 10x10 map).
// w packed [Cout][3][3][Cin] (prep.hip pack_conv3x3); dy [F, OH, OW, Cout]; dx [F, H, W, Cin]
__global__ __launch_bounds__(256) void conv3x3_s2_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ wp, float *__restrict__ dx, int F, int H,
                                                             int W, int Cin, int Cout, int OH, int OW) {
    const long long total = (long long)F * H * W * Cin;
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= total) return;
    const int ci = (int)(gid % Cin);
    long long r = gid / Cin;
    const int x = (int)(r % W);
    r /= W;
    const int y = (int)(r % H);
    const int f = (int)(r / H);
    float acc = 0.f;
    for (int ky = 0; ky < 3; ++ky) {
        const int ty = y + 1 - ky;  // = 2 * oy
        if (ty < 0 || (ty & 1)) continue;
        const int oy = ty >> 1;
        if (oy >= OH) continue;
        for (int kx = 0; kx < 3; ++kx) {
            const int tx = x + 1 - kx;
            if (tx < 0 || (tx & 1)) continue;
            const int ox = tx >> 1;
            if (ox >= OW) continue;
            const float *g = dy + (((long long)f * OH + oy) * OW + ox) * Cout;
            const float *wk = wp + ((long long)(ky * 3 + kx)) * Cin + ci;
            for (int co = 0; co < Cout; ++co) acc += g[co] * wk[(long long)co * 9 * Cin];
        }
    }
    dx[gid] = acc;
}

// w [Cout, Cin, 3, 3] -> the packed weight of the stride-1 input-gradient convolution: [Cin][3][3][Cout], taps flipped
__global__ __launch_bounds__(256) void pack_conv3x3_bwd_kernel(const float *__restrict__ w, float *__restrict__ out, int Cout, int Cin) {
    const int total = Cout * Cin * 9;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int co = i % Cout;
    int r = i / Cout;
    const int tap = r % 9;
    const int ci = r / 9;
    const int ky = tap / 3, kx = tap - ky * 3;
    out[i] = w[(((long long)co * Cin + ci) * 3 + (2 - ky)) * 3 + (2 - kx)];
}

// Aeff [r, nin] = A * U (U [r, 1]);  BgT [r, nout] = (B * V * gamma)^T (V, gamma [nout])   (U, V, gamma optional)
__global__ __launch_bounds__(256) void lora_factors_kernel(const float *__restrict__ A, const float *__restrict__ Bm, const float *__restrict__ U,
                                                           const float *__restrict__ V, const float *__restrict__ gamma, float *__restrict__ Aeff,
                                                           float *__restrict__ BgT, int nout, int nin, int r) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < r * nin) Aeff[i] = A[i] * (U ? U[i / nin] : 1.f);
    if (i < r * nout) {
        const int j = i / nout, n = i - j * nout;
        BgT[i] = Bm[n * r + j] * (V ? V[n] : 1.f) * (gamma ? gamma[n] : 1.f);
    }
}

// ---- Linear_SSB (mylora/layers.py:396-430): W' = b * W * a^T with a = lora_A [in, 1], b = lora_B [out, 1] -------------------
// Wa[n, k] = W[n, k] * a[k]  (weight of z = (x * a) W^T);   gb[n] = gamma[n] * b[n]  (row scale of the transposed weight of u)
__global__ __launch_bounds__(256) void ssb_prep_kernel(const float *__restrict__ W, const float *__restrict__ a, const float *__restrict__ b,
                                                       const float *__restrict__ gamma, float *__restrict__ Wa, float *__restrict__ gb, int nout, int nin) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < (long long)nout * nin) Wa[i] = W[i] * a[i % nin];
    if (i < nout) gb[i] = b[i] * (gamma ? gamma[i] : 1.f);
}
// part[s, n] = sum_{m in split s} P[m, n] * Q[m, n]   (Q = null: P[m, n])
__global__ __launch_bounds__(256) void col_dot_partial_kernel(const float *__restrict__ P, const float *__restrict__ Q, long long M, int N,
                                                              float *__restrict__ part, int rows_per_split) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + lane;
    const long long mb = (long long)blockIdx.y * rows_per_split;
    const long long me = mb + rows_per_split < M ? mb + rows_per_split : M;
    float acc = 0.f;
    if (n < N)
        for (long long m = mb + wv; m < me; m += 4) acc += Q ? P[m * N + n] * Q[m * N + n] : P[m * N + n];
    red[wv][lane] = acc;
    __syncthreads();
    if (wv == 0 && n < N) part[(long long)blockIdx.y * N + n] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

int ew_blocks(long long n) { return (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192); }

}  // namespace

int layernorm_bwd(const float *x, RowMap xmap, const float *w, const float *dy, RowMap dymap, float *dx, RowMap dxmap, long long rows, int dim, float eps,
                  bool accumulate, hipStream_t st) {
    EDV_CHECK(x && w && dy && dx, "null operand");
    EDV_CHECK(rows > 0 && dim % 4 == 0 && dim <= 256 * LNB_MAXV, "dim must be a multiple of 4 and <= 1024");
    const long long blocks = (rows + 3) / 4;
    EDV_CHECK(blocks < (1ll << 31), "grid");
    EDV_LAUNCH(layernorm_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, xmap, w, dy, dymap, dx, dxmap, rows, dim, eps, accumulate ? 1 : 0);
    EDV_LAUNCH_OK();
    return 0;
}

int ew_bwd(const float *d, const float *src, const float *add, float *out, long long n, int mode, hipStream_t st) {
    EDV_CHECK(d && out && n > 0 && n % 4 == 0, "shape");
    EDV_CHECK(mode >= 0 && mode <= 3 && (mode == 0 || mode == 3 || src), "mode");
    EDV_LAUNCH(ew_bwd_kernel, dim3(ew_blocks(n / 4)), dim3(256), 0, st, d, src, add, out, n / 4, mode);
    EDV_LAUNCH_OK();
    return 0;
}

int sigmoid_bwd(const float *g, const float *s, float *out, long long n, hipStream_t st) {
    EDV_CHECK(g && s && out && n > 0, "shape");
    EDV_LAUNCH(sigmoid_bwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, st, g, s, out, n);
    EDV_LAUNCH_OK();
    return 0;
}

int geglu_bwd(const float *x, const float *dy, float *dx, long long M, int inner, hipStream_t st) {
    EDV_CHECK(x && dy && dx && M > 0 && inner > 0 && inner % 4 == 0, "shape");
    const long long total4 = M * (inner / 4);
    EDV_LAUNCH(geglu_bwd_kernel, dim3(ew_blocks(total4)), dim3(256), 0, st, x, dy, dx, total4, inner / 4);
    EDV_LAUNCH_OK();
    return 0;
}

int transpose_scale(const float *W, int ldw, const float *gamma, float *Wt, int N, int K, hipStream_t st) {
    EDV_CHECK(W && Wt && N > 0 && K > 0 && ldw >= K, "shape");
    EDV_LAUNCH(transpose_scale_kernel, dim3((K + 31) / 32, (N + 31) / 32), dim3(256), 0, st, W, gamma, Wt, N, K, ldw);
    EDV_LAUNCH_OK();
    return 0;
}

int skinny_xwt(const float *X, long long M, int K, int ldx, const float *Wr, int r, float *T, hipStream_t st) {
    EDV_CHECK(X && Wr && T && M > 0 && K > 0 && K % 4 == 0 && ldx % 4 == 0, "shape");
    EDV_CHECK(r == 1 || r == 2 || r == 4 || r == 8, "rank must be 1, 2, 4 or 8");
    const long long blocks = (M + 3) / 4;
    EDV_CHECK(blocks < (1ll << 31), "grid");
    dim3 grid((unsigned)blocks), block(256);
    switch (r) {
        case 1: EDV_LAUNCH(skinny_xwt_kernel<1>, grid, block, 0, st, X, M, K, ldx, Wr, T); break;
        case 2: EDV_LAUNCH(skinny_xwt_kernel<2>, grid, block, 0, st, X, M, K, ldx, Wr, T); break;
        case 4: EDV_LAUNCH(skinny_xwt_kernel<4>, grid, block, 0, st, X, M, K, ldx, Wr, T); break;
        default: EDV_LAUNCH(skinny_xwt_kernel<8>, grid, block, 0, st, X, M, K, ldx, Wr, T); break;
    }
    EDV_LAUNCH_OK();
    return 0;
}

size_t tall_tn_workspace(int N, int r) { return (size_t)TALL_SPLITS * N * r; }

int tall_tn(const float *Y, int ldy, const float *T, long long M, int N, int r, float scale, const float *rowscale, float *part, float *out, hipStream_t st) {
    EDV_CHECK(Y && T && part && out && M > 0 && N > 0, "shape");
    EDV_CHECK(r == 1 || r == 2 || r == 4 || r == 8, "rank must be 1, 2, 4 or 8");
    const int rows_per_split = (int)((M + TALL_SPLITS - 1) / TALL_SPLITS);
    const int splits = (int)((M + rows_per_split - 1) / rows_per_split);
    dim3 grid((N + 63) / 64, splits), block(256);
    switch (r) {
        case 1: EDV_LAUNCH(tall_tn_partial_kernel<1>, grid, block, 0, st, Y, ldy, T, M, N, part, rows_per_split); break;
        case 2: EDV_LAUNCH(tall_tn_partial_kernel<2>, grid, block, 0, st, Y, ldy, T, M, N, part, rows_per_split); break;
        case 4: EDV_LAUNCH(tall_tn_partial_kernel<4>, grid, block, 0, st, Y, ldy, T, M, N, part, rows_per_split); break;
        default: EDV_LAUNCH(tall_tn_partial_kernel<8>, grid, block, 0, st, Y, ldy, T, M, N, part, rows_per_split); break;
    }
    EDV_LAUNCH_OK();
    EDV_LAUNCH(tall_tn_reduce_kernel, dim3((N * r + 31) / 32), dim3(256), 0, st, part, splits, N, r, scale, rowscale, out);
    EDV_LAUNCH_OK();
    return 0;
}

int lora_grad_finalize(const float *dBp, const float *dApT, const float *A, const float *Bm, const float *U, const float *V, float *dA, float *dB, float *dU,
                       float *dV, int nout, int nin, int r, hipStream_t st) {
    EDV_CHECK(dBp && dApT && nout > 0 && nin > 0 && r > 0, "shape");
    EDV_CHECK((U == nullptr) == (V == nullptr), "U and V come together");
    EDV_CHECK(!U || (A && Bm), "dvlora needs A and B");
    const int n = (nout > nin ? nout : nin) * r;
    EDV_LAUNCH(lora_grad_finalize_kernel, dim3((n + 255) / 256), dim3(256), 0, st, dBp, dApT, A, Bm, U, V, dA, dB, dV, nout, nin, r);
    EDV_LAUNCH_OK();
    if (U && dU) {
        EDV_LAUNCH(lora_grad_u_kernel, dim3(r), dim3(256), 0, st, dApT, A, dU, nin, r);
        EDV_LAUNCH_OK();
    }
    return 0;
}

size_t lora_grads_workspace(long long M, int nin, int nout, int r) {
    const int nmax = nin > nout ? nin : nout;
    return (size_t)2 * M * r + (size_t)r * (nin + nout) + tall_tn_workspace(nmax, r) + (size_t)r * (nin + nout) + 64;
}

// Gradients of the LoRA factors of y = gamma * (x (W + s B' A')^T + b), A' = A*U, B' = B*V (mylora/layers.py:148-157,
// 384-393), given x [M, nin] and G = dL/dy [M, nout]:   dB' = s gamma (G^T (x A'^T)),   dA' = s (G gamma B')^T x.
int lora_grads(const float *X, int ldx, const float *G, int ldg, long long M, int nin, int nout, int r, const float *A, const float *Bm, const float *U,
               const float *V, float s, const float *gamma, float *ws, size_t ws_floats, float *dA, float *dB, float *dU, float *dV, hipStream_t st) {
    EDV_CHECK(X && G && A && Bm && ws, "null operand");
    EDV_CHECK(ws_floats >= lora_grads_workspace(M, nin, nout, r), "workspace too small (lora_grads_workspace)");
    float *t = ws, *u = t + M * r, *Aeff = u + M * r, *BgT = Aeff + (size_t)r * nin;
    float *part = BgT + (size_t)r * nout;
    float *dBp = part + tall_tn_workspace(nin > nout ? nin : nout, r), *dApT = dBp + (size_t)r * nout;
    const int nf = r * (nin > nout ? nin : nout);
    EDV_LAUNCH(lora_factors_kernel, dim3((nf + 255) / 256), dim3(256), 0, st, A, Bm, U, V, gamma, Aeff, BgT, nout, nin, r);
    EDV_LAUNCH_OK();
    EDV_TRY(skinny_xwt(X, M, nin, ldx, Aeff, r, t, st));
    EDV_TRY(skinny_xwt(G, M, nout, ldg, BgT, r, u, st));
    EDV_TRY(tall_tn(G, ldg, t, M, nout, r, s, gamma, part, dBp, st));
    EDV_TRY(tall_tn(X, ldx, u, M, nin, r, s, nullptr, part, dApT, st));
    return lora_grad_finalize(dBp, dApT, A, Bm, U, V, dA, dB, dU, dV, nout, nin, r, st);
}

int ssb_prep(const float *W, const float *a, const float *b, const float *gamma, float *Wa, float *gb, int nout, int nin, hipStream_t st) {
    EDV_CHECK(W && a && b && Wa && gb && nout > 0 && nin > 0, "shape");
    const long long n = (long long)nout * nin;
    EDV_LAUNCH(ssb_prep_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, W, a, b, gamma, Wa, gb, nout, nin);
    EDV_LAUNCH_OK();
    return 0;
}

// out[n] = scale[n] * sum_m P[m, n] Q[m, n]  (deterministic two-stage reduction; part: TALL_SPLITS * N floats)
int col_dot(const float *P, const float *Q, long long M, int N, const float *scale, float *part, float *out, hipStream_t st) {
    EDV_CHECK(P && part && out && M > 0 && N > 0, "shape");  // Q may be null: plain column sums
    const int rows_per_split = (int)((M + TALL_SPLITS - 1) / TALL_SPLITS);
    const int splits = (int)((M + rows_per_split - 1) / rows_per_split);
    EDV_LAUNCH(col_dot_partial_kernel, dim3((N + 63) / 64, splits), dim3(256), 0, st, P, Q, M, N, part, rows_per_split);
    EDV_LAUNCH_OK();
    EDV_LAUNCH(tall_tn_reduce_kernel, dim3((N + 31) / 32), dim3(256), 0, st, part, splits, N, 1, 1.0f, scale, out);
    EDV_LAUNCH_OK();
    return 0;
}

int bilinear_bwd(const float *dy, float *dx, int F, int ih, int iw, int C, int oh, int ow, bool accumulate, hipStream_t st) {
    EDV_CHECK(dy && dx && F > 0 && ih > 0 && iw > 0 && C > 0 && oh > 0 && ow > 0, "shape");
    const float ry = oh > 1 ? (float)(ih - 1) / (float)(oh - 1) : 0.f, rx = ow > 1 ? (float)(iw - 1) / (float)(ow - 1) : 0.f;
    if (C % 4 == 0) {
        const long long total = (long long)F * ih * iw * (C / 4);
        EDV_CHECK((total + 255) / 256 < (1ll << 31), "grid");
        EDV_LAUNCH(bilinear_bwd_kernel<4>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, dy, dx, F, ih, iw, C, oh, ow, ry, rx, accumulate ? 1 : 0);
    } else {
        const long long total = (long long)F * ih * iw * C;
        EDV_CHECK((total + 255) / 256 < (1ll << 31), "grid");
        EDV_LAUNCH(bilinear_bwd_kernel<1>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, dy, dx, F, ih, iw, C, oh, ow, ry, rx, accumulate ? 1 : 0);
    }
    EDV_LAUNCH_OK();
    return 0;
}

int dot_channels_bwd(const float *g, const float *disp, const float *w, const float *o2, float *d_o2, float *gz_out, long long npix, int C, int mode,
                     hipStream_t st) {
    EDV_CHECK(g && disp && w && o2 && d_o2 && npix > 0 && C % 4 == 0, "shape");
    EDV_CHECK(mode >= 0 && mode <= 2, "mode: 0 ReLU, 1 sigmoid(z), 2 sigmoid(-z)");
    EDV_LAUNCH(dot_channels_bwd_kernel, dim3(ew_blocks(npix * (C / 4))), dim3(256), 0, st, g, disp, w, o2, d_o2, gz_out, npix, C, mode);
    EDV_LAUNCH_OK();
    return 0;
}

int groupnorm_bwd(const float *x, const float *stats, const float *w, const float *dy, float *sums, float *dx, int F, int P, int C, int groups, bool accumulate,
                  hipStream_t st) {
    EDV_CHECK(x && stats && w && dy && sums && dx, "null operand");
    EDV_CHECK(F > 0 && F <= 65535 && P > 0 && C > 0 && groups > 0 && C % groups == 0 && C % 4 == 0, "shape");
    EDV_LAUNCH(groupnorm_bwd_sums_kernel, dim3(groups, F), dim3(256), 0, st, x, stats, w, dy, sums, P, C, groups);
    EDV_LAUNCH_OK();
    const long long total4 = (long long)F * P * C / 4;
    EDV_LAUNCH(groupnorm_bwd_apply_kernel, dim3(ew_blocks(total4)), dim3(256), 0, st, x, stats, w, dy, sums, dx, total4, P, C, groups, accumulate ? 1 : 0);
    EDV_LAUNCH_OK();
    return 0;
}

int attn_temporal_bwd(const float *qkv, const float *dout, float *dqkv, int B, int T, int P, int C, int heads, hipStream_t st) {
    EDV_CHECK(qkv && dout && dqkv, "null operand");
    EDV_CHECK(B > 0 && T > 0 && T <= 32 && P > 0 && C > 0 && heads > 0 && C % heads == 0 && (C / heads) % 4 == 0, "shape (head dim % 4)");
    const long long total = (long long)B * P * heads;
    const long long blocks = (total + 63) / 64;
    EDV_CHECK(blocks < (1ll << 31), "grid");
    const float scale = 1.0f / sqrtf((float)(C / heads));
    dim3 grid((unsigned)blocks), block(64);
    static const bool per_thread = [] {
        const char *e = getenv("EDV_TATTN_BWD_PER_THREAD");  // 1: the one-thread-per-(pixel, head) kernel only (A/B runs)
        return e && atoi(e) != 0;
    }();
    const int d = C / heads, TM = T <= 8 ? 8 : (T <= 16 ? 16 : 32);
    auto lds_of = [&](int hgv) { return ((size_t)T * 4 * hgv * d + (size_t)2 * hgv * T * TM) * sizeof(float); };
    int HG = heads;  // heads per workgroup: as many as fit 256 threads and 64 KB of LDS
    while (HG > 1 && ((long long)T * HG > 256 || lds_of(HG) > 64 * 1024 || heads % HG != 0)) --HG;
    // one head of a wide module at T = 32 needs more than 64 KB (d = 128, ViT-L: 72 KB): the CU has 160.  (Round 2 sent that case to the
    // one-thread-per-(pixel, head) kernel, whose <32> instantiation spilled 232 VGPRs to scratch; it is gone.)
    const size_t lds_cap = HG == 1 ? 160 * 1024 : 64 * 1024;
    const bool pixel_fits = d % 4 == 0 && (long long)T * HG <= 256 && lds_of(HG) <= lds_cap && (long long)B * P * (heads / HG) < (1ll << 31);
    if (pixel_fits && !per_thread) {
        const dim3 g3((unsigned)((long long)B * P * (heads / HG))), b3((unsigned)(((T * HG + 63) / 64) * 64));
        if (lds_of(HG) > 64 * 1024) {
            EDV_CHECK(TM == 32, "temporal attention backward: LDS plan");
            EDV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(attn_temporal_bwd_pixel_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_cap));
        }
        if (TM == 8)
            EDV_LAUNCH(attn_temporal_bwd_pixel_kernel<8>, g3, b3, lds_of(HG), st, qkv, dout, dqkv, T, P, C, heads, HG, scale);
        else if (TM == 16)
            EDV_LAUNCH(attn_temporal_bwd_pixel_kernel<16>, g3, b3, lds_of(HG), st, qkv, dout, dqkv, T, P, C, heads, HG, scale);
        else
            EDV_LAUNCH(attn_temporal_bwd_pixel_kernel<32>, g3, b3, lds_of(HG), st, qkv, dout, dqkv, T, P, C, heads, HG, scale);
    } else if (T <= 8)
        EDV_LAUNCH(attn_temporal_bwd_kernel<8>, grid, block, 0, st, qkv, dout, dqkv, B, T, P, C, heads, scale);
    else if (T <= 16)
        EDV_LAUNCH(attn_temporal_bwd_kernel<16>, grid, block, 0, st, qkv, dout, dqkv, B, T, P, C, heads, scale);
    else
        EDV_CHECK(false, "temporal attention backward at T > 16: one head's rows of a pixel must fit the 160 KB of LDS (head dim <= 272 at T = 32)");
    EDV_LAUNCH_OK();
    return 0;
}

int pixel_unshuffle(const float *dy, float *A, int F, int h, int w, int C, int s, hipStream_t st) {
    EDV_CHECK(dy && A && F > 0 && h > 0 && w > 0 && C % 4 == 0 && s > 0, "shape");
    const long long total = (long long)F * h * w * s * s * (C / 4);
    EDV_LAUNCH(pixel_unshuffle_kernel, dim3(ew_blocks(total)), dim3(256), 0, st, dy, A, F, h, w, C, s);
    EDV_LAUNCH_OK();
    return 0;
}

int conv3x3_s2_bwd(const float *dy, const float *wpacked, float *dx, int F, int H, int W, int Cin, int Cout, hipStream_t st) {
    EDV_CHECK(dy && wpacked && dx && F > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "shape");
    const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
    const long long total = (long long)F * H * W * Cin;
    EDV_CHECK((total + 255) / 256 < (1ll << 31), "grid");
    EDV_LAUNCH(conv3x3_s2_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, dy, wpacked, dx, F, H, W, Cin, Cout, OH, OW);
    EDV_LAUNCH_OK();
    return 0;
}

int dilate2(const float *dy, float *z, int F, int H, int W, int C, hipStream_t st) {
    EDV_CHECK(dy && z && F > 0 && H > 0 && W > 0 && C % 4 == 0, "shape");
    const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
    const long long total = (long long)F * H * W * (C / 4);
    EDV_LAUNCH(dilate2_kernel, dim3(ew_blocks(total)), dim3(256), 0, st, dy, z, F, H, W, C, OH, OW);
    EDV_LAUNCH_OK();
    return 0;
}

int pack_conv3x3_bwd(const float *w, float *out, int Cout, int Cin, hipStream_t st) {
    EDV_CHECK(w && out && Cout > 0 && Cin > 0, "shape");
    const int total = Cout * Cin * 9;
    EDV_LAUNCH(pack_conv3x3_bwd_kernel, dim3((total + 255) / 256), dim3(256), 0, st, w, out, Cout, Cin);
    EDV_LAUNCH_OK();
    return 0;
}

}  // namespace edv
