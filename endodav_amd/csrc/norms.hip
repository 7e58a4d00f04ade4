// LayerNorm and GroupNorm, fp32, HBM-bound row kernels.
//   LayerNorm eps 1e-6: vision_transformer.py:97, block.py:69,85 (norm1/norm2), final norm :317
//   LayerNorm eps 1e-5 + sinusoidal PE add: motion_module.py:155,161,164-172,197
//   GroupNorm(32, eps 1e-6) on the motion-module input: motion_module.py:84,110
// One wave per row; the row lives in registers (float4 per lane), mean and the centred second
// moment are reduced with wavefront shuffles (two-pass in registers: no E[x^2]-E[x]^2 cancellation).
#include <cstdlib>

#include "ops.hpp"

namespace edv {
namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int LN_MAXV = 4;  // float4 per lane -> dim <= 1024

// R rows per wave (the loads of all R rows are issued before the first reduction); NV = float4 per lane per row.
template <int R, int NV>
__global__ __launch_bounds__(256) void layernorm_kernel(const float *__restrict__ x, RowMap in_map, const float *__restrict__ w,
                                                         const float *__restrict__ b, float *y, RowMap out_map, long long rows,
                                                         int dim, float eps, const float *__restrict__ pe, int rows_per_frame, int T, int act,
                                                         int accumulate) {
    const int lane = threadIdx.x & 63;
    const long long row0 = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * R;
    if (row0 >= rows) return;
    const int nv = dim >> 2;
    f32x4 v[R][NV];
    float s[R];
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const long long row = row0 + k < rows ? row0 + k : rows - 1;  // past the end: a valid row again, never stored
        const float *xr = x + in_map(row) * dim;
        s[k] = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
                v[k][i] = *reinterpret_cast<const f32x4 *>(xr + 4 * c);
            } else {
                v[k][i] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    }
    f32x4 g[NV], be[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
            g[i] = *reinterpret_cast<const f32x4 *>(w + 4 * c);
            be[i] = *reinterpret_cast<const f32x4 *>(b + 4 * c);
        }
    }
#pragma unroll
    for (int k = 0; k < R; ++k) {
#pragma unroll
        for (int i = 0; i < NV; ++i) s[k] += (v[k][i].x + v[k][i].y) + (v[k][i].z + v[k][i].w);  // zero beyond the row
    }
#pragma unroll
    for (int k = 0; k < R; ++k) {
        if (row0 + k >= rows) break;
        const long long row = row0 + k;
        const float mean = wave_sum(s[k]) / (float)dim;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
                const f32x4 d = v[k][i] - mean;
                q += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
            }
        }
        const float rstd = rsqrtf(wave_sum(q) / (float)dim + eps);
        float *yr = y + out_map(row) * dim;
        const float *per = pe ? pe + (long long)((row / rows_per_frame) % T) * dim : nullptr;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
                f32x4 o = (v[k][i] - mean) * rstd * g[i] + be[i];
                if (per) o += *reinterpret_cast<const f32x4 *>(per + 4 * c);
                if (act == ACT_GELU) o = f32x4{gelu_erf(o.x), gelu_erf(o.y), gelu_erf(o.z), gelu_erf(o.w)};
                if (accumulate) o += *reinterpret_cast<const f32x4 *>(yr + 4 * c);
                *reinterpret_cast<f32x4 *>(yr + 4 * c) = o;
            }
        }
    }
}

// GroupNorm statistics: one workgroup per (frame, group); x is channels-last [F, P, C].
// Pass 1 mean, pass 2 centred second moment (the slab is re-read from L2).
__global__ __launch_bounds__(256) void groupnorm_stats_kernel(const float *__restrict__ x, float *__restrict__ stats, int P, int C, int groups, float eps) {
    __shared__ float red[4];
    __shared__ float bc;
    const int f = blockIdx.y, g = blockIdx.x, cg = C / groups;
    const float *xf = x + (long long)f * P * C + g * cg;
    const int n = P * cg;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float s = 0.f;
#pragma unroll 8  // the slab is L2-resident and the loop is one dependent load per trip: keep several in flight
    for (int i = threadIdx.x; i < n; i += 256) {
        const int p = i / cg, c = i - p * cg;
        s += xf[(long long)p * C + c];
    }
    s = wave_sum(s);
    if (lane == 0) red[wv] = s;
    __syncthreads();
    if (threadIdx.x == 0) bc = (red[0] + red[1] + red[2] + red[3]) / (float)n;
    __syncthreads();
    const float mean = bc;
    float q = 0.f;
#pragma unroll 8
    for (int i = threadIdx.x; i < n; i += 256) {
        const int p = i / cg, c = i - p * cg;
        const float d = xf[(long long)p * C + c] - mean;
        q += d * d;
    }
    q = wave_sum(q);
    __syncthreads();
    if (lane == 0) red[wv] = q;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float var = (red[0] + red[1] + red[2] + red[3]) / (float)n;
        stats[((long long)f * groups + g) * 2 + 0] = mean;
        stats[((long long)f * groups + g) * 2 + 1] = rsqrtf(var + eps);
    }
}

// Two-stage statistics with coalesced reads (the kernel above reads cg = C / groups floats of every 4C-byte pixel row per workgroup:
// 15 us at [8, 5476, 64]).  Stage 1: one workgroup per (32-pixel chunk, frame) reads whole rows as float4 and leaves, per channel, the
// chunk's mean and centred second moment (two passes over the chunk, the second from L2).  Stage 2: one wave per (frame, group) merges
// the (chunk, channel) pairs with the parallel-variance formula  M2 = sum M2_i + sum n_i (mean_i - mean)^2  -- as accurate as two passes.
constexpr int GN_ROWS = 32;
__global__ __launch_bounds__(256) void groupnorm_partial_kernel(const float *__restrict__ x, float *__restrict__ part, int P, int C, int nch) {
    __shared__ f32x4 red[256];
    __shared__ f32x4 meanq[256];
    const int c4n = C >> 2, rpp = 256 / c4n, tid = threadIdx.x;
    const bool active = tid < rpp * c4n;
    const int cq = tid % c4n, r0 = tid / c4n;
    const int f = blockIdx.y, chunk = blockIdx.x;
    const int p0 = chunk * GN_ROWS, p1 = p0 + GN_ROWS < P ? p0 + GN_ROWS : P;
    const float *xf = x + ((long long)f * P) * C + cq * 4;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (active)
        for (int p = p0 + r0; p < p1; p += rpp) s += *reinterpret_cast<const f32x4 *>(xf + (long long)p * C);
    red[tid] = s;
    __syncthreads();
    if (tid < c4n) {
        f32x4 t = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < rpp; ++k) t += red[k * c4n + tid];
        meanq[tid] = t * (1.0f / (float)(p1 - p0));
    }
    __syncthreads();
    const f32x4 m = meanq[cq];
    f32x4 q = {0.f, 0.f, 0.f, 0.f};
    if (active)
        for (int p = p0 + r0; p < p1; p += rpp) {
            const f32x4 d = *reinterpret_cast<const f32x4 *>(xf + (long long)p * C) - m;
            q += d * d;
        }
    red[tid] = q;
    __syncthreads();
    if (tid < c4n) {
        f32x4 t = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < rpp; ++k) t += red[k * c4n + tid];
        float *o = part + (((long long)f * nch + chunk) * 2) * C + tid * 4;
        *reinterpret_cast<f32x4 *>(o) = meanq[tid];
        *reinterpret_cast<f32x4 *>(o + C) = t;
    }
}

__global__ __launch_bounds__(64) void groupnorm_finish_kernel(const float *__restrict__ part, float *__restrict__ stats, int P, int C, int groups, int nch,
                                                              float eps) {
    const int f = blockIdx.y, g = blockIdx.x, cg = C / groups, lane = threadIdx.x;
    const int items = nch * cg;
    const float *pf = part + ((long long)f * nch * 2) * C + g * cg;
    float sw = 0.f;
    for (int i = lane; i < items; i += 64) {
        const int ch = i / cg, c = i - ch * cg;
        const int n = (ch + 1) * GN_ROWS <= P ? GN_ROWS : P - ch * GN_ROWS;
        sw += (float)n * pf[(long long)ch * 2 * C + c];
    }
    const float mean = wave_sum(sw) / ((float)P * (float)cg);
    float m2 = 0.f;
    for (int i = lane; i < items; i += 64) {
        const int ch = i / cg, c = i - ch * cg;
        const int n = (ch + 1) * GN_ROWS <= P ? GN_ROWS : P - ch * GN_ROWS;
        const float d = pf[(long long)ch * 2 * C + c] - mean;
        m2 += pf[(long long)ch * 2 * C + C + c] + (float)n * d * d;
    }
    m2 = wave_sum(m2);
    if (lane == 0) {
        stats[((long long)f * groups + g) * 2 + 0] = mean;
        stats[((long long)f * groups + g) * 2 + 1] = rsqrtf(m2 / ((float)P * (float)cg) + eps);
    }
}

// y = (x - mean[f, g]) * rstd[f, g] * w[c] + b[c].  Grid (pixel chunks, frames): a thread keeps ONE channel quad for all its pixels, so the four
// (mean, rstd, w, b) sets are loaded once and the loop is load - fma - store (round 3; the first version derived frame, pixel and group from a flat
// index with two 64-bit divisions per float4: 11.7 us on 13 MB).
__global__ __launch_bounds__(256) void groupnorm_apply_kernel(const float *__restrict__ x, const float *__restrict__ stats, const float *__restrict__ w,
                                                               const float *__restrict__ b, float *__restrict__ y, int P, int C, int groups) {
    const int cg = C / groups, c4n = C >> 2, rpp = 256 / c4n, tid = threadIdx.x;
    if (tid >= rpp * c4n) return;
    const int cq = tid % c4n, r0 = tid / c4n, f = blockIdx.y;
    f32x4 mu, sc, sh;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int ch = cq * 4 + e, g = ch / cg;
        mu[e] = stats[((long long)f * groups + g) * 2];
        sc[e] = stats[((long long)f * groups + g) * 2 + 1] * w[ch];
        sh[e] = b[ch];
    }
    const long long base = (long long)f * P * C + cq * 4;
    for (int p = blockIdx.x * rpp + r0; p < P; p += gridDim.x * rpp) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(x + base + (long long)p * C);
        *reinterpret_cast<f32x4 *>(y + base + (long long)p * C) = (v - mu) * sc + sh;  // centre first: no cancellation between v * sc and mean * sc
    }
}

}  // namespace

int layernorm(const float *x, RowMap in_map, const float *w, const float *b, float *y, RowMap out_map, long long rows, int dim, float eps,
              const float *pe, int rows_per_frame, int T, hipStream_t st, int act, bool accumulate) {
    EDV_CHECK(x && w && b && y, "null operand");
    EDV_CHECK(act == ACT_NONE || act == ACT_GELU, "layernorm act must be none or gelu");
    EDV_CHECK(rows > 0, "empty problem");
    EDV_CHECK(dim % 4 == 0 && dim <= 256 * LN_MAXV, "dim must be a multiple of 4 and <= 1024");
    EDV_CHECK(!pe || (rows_per_frame > 0 && T > 0), "pe needs rows_per_frame and T");
    // rows per wave.  Measured on [10960, 384] (ViT-S T=8; profiles/r02_notes.txt): 1 row 12.0 us, 2 rows 12.9, 4 rows 14.3 -- the kernel is
    // bound by how many waves are in flight, not by one wave's load -> reduce -> store chain, so one row per wave stays the default
    // (EDV_LN_ROWS=2 / 4 for A/B runs).
    static const int forced = [] {
        const char *e = getenv("EDV_LN_ROWS");
        return e ? atoi(e) : 0;
    }();
    int R = 1;
    if (forced == 1 || forced == 2 || forced == 4) R = forced;
    if (dim > 512 && R > 2) R = 2;  // R x NV float4 of row data per lane
    const long long blocks = (rows + 4 * R - 1) / (4 * R);
    EDV_CHECK(blocks < (1ll << 31), "grid");
    const int rpf = rows_per_frame > 0 ? rows_per_frame : 1, TT = T > 0 ? T : 1, acc = accumulate ? 1 : 0;
#define EDV_LN_LAUNCH(RR, NVV)                                                                                                                  \
    EDV_LAUNCH((layernorm_kernel<RR, NVV>), dim3((unsigned)blocks), dim3(256), 0, st, x, in_map, w, b, y, out_map, rows, dim, eps, pe, rpf, TT, \
                       act, acc)
    if (dim <= 512) {
        if (R == 4) EDV_LN_LAUNCH(4, 2);
        else if (R == 2) EDV_LN_LAUNCH(2, 2);
        else EDV_LN_LAUNCH(1, 2);
    } else {
        if (R == 2) EDV_LN_LAUNCH(2, 4);
        else EDV_LN_LAUNCH(1, 4);
    }
#undef EDV_LN_LAUNCH
    EDV_LAUNCH_OK();
    return 0;
}

size_t groupnorm_workspace(int F, int P, int C) { return (size_t)F * ((P + GN_ROWS - 1) / GN_ROWS) * 2 * C; }

int groupnorm(const float *x, const float *w, const float *b, float *y, float *stats, int F, int P, int C, int groups, float eps, hipStream_t st, float *part,
              size_t part_floats) {
    EDV_CHECK(x && w && b && y && stats, "null operand");
    EDV_CHECK(F > 0 && P > 0 && C > 0 && groups > 0 && C % groups == 0 && C % 4 == 0, "shape");
    EDV_CHECK(F <= 65535, "grid");
    const int nch = (P + GN_ROWS - 1) / GN_ROWS;
    if (part && C <= 1024 && nch <= 65535) {  // coalesced two-stage statistics
        EDV_CHECK(groupnorm_workspace(F, P, C) <= part_floats && (uintptr_t)part % 16 == 0, "groupnorm workspace too small (groupnorm_workspace)");
        EDV_LAUNCH(groupnorm_partial_kernel, dim3(nch, F), dim3(256), 0, st, x, part, P, C, nch);
        EDV_LAUNCH_OK();
        EDV_LAUNCH(groupnorm_finish_kernel, dim3(groups, F), dim3(64), 0, st, part, stats, P, C, groups, nch, eps);
        EDV_LAUNCH_OK();
    } else {
        EDV_LAUNCH(groupnorm_stats_kernel, dim3(groups, F), dim3(256), 0, st, x, stats, P, C, groups, eps);
        EDV_LAUNCH_OK();
    }
    EDV_CHECK(C <= 1024, "groupnorm: at most 1024 channels");
    const int rpp = 256 / (C >> 2);  // pixel rows a workgroup covers per trip
    int chunks = (P + 4 * rpp - 1) / (4 * rpp);  // about four trips per thread
    chunks = chunks < 1 ? 1 : (chunks > 2048 ? 2048 : chunks);
    EDV_LAUNCH(groupnorm_apply_kernel, dim3((unsigned)chunks, (unsigned)F), dim3(256), 0, st, x, stats, w, b, y, P, C, groups);
    EDV_LAUNCH_OK();
    return 0;
}

}  // namespace edv
