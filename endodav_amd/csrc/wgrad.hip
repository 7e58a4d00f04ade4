// Weight and bias gradients of the trainable convolutions of the DPT output heads (with the reference's
// default options `mark_only_part_as_trainable` leaves conv_depth_* trainable, endodav/layers.py:5-34; `--train_output_conv`
// adds scratch.output_conv*).  Everything else in the head is frozen and contributes its input gradient only (bwd.hip).
//
//   conv3_wgrad   dW[co, ci, ky, kx] = sum over pixels of dY[p, co] * X[p + (ky-1, kx-1), ci]        (3x3, stride 1, padding 1)
//   colsum_rows   out[n] = sum_m rowscale[m] * P[m, n]     bias gradients (rowscale = null) and the 1x1 head's weight gradient
//
// conv3_wgrad is a TN GEMM whose contraction runs over the pixels: M = Cout, N = 9 Cin, K = F*H*W.  One wave owns a 32-row
// block of Cout, three 32-column blocks (column = tap * Cin + ci: with Cin % 32 == 0 a block is one tap x 32 input channels, i.e.
// 128 contiguous bytes of the channels-last input per pixel; any other Cin just spreads a block over neighbouring taps, every
// lane carries its own tap offset) and a few image rows; per MFMA step it contracts two neighbouring pixels: lane
// (l31, h) loads dY[pixel 2s+h, co0 + l31] and, per column block, X[shifted pixel, ci0 + l31] -- coalesced 128-byte segments
// straight from global memory / L1 (the same input row is re-read by the nine taps and by every Cout block), no LDS.  The
// padding is a predicate on the shifted coordinate.  Each wave leaves its raw 32x96 accumulators in the workspace; a second
// kernel sums the pieces in a fixed order (deterministic) and writes the torch layout [Cout, Cin, 3, 3].
#include "common.hpp"
#include "ops.hpp"

namespace edv {
namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int WG_NB = 3;  // column blocks per wave

__global__ __launch_bounds__(256) void conv3_wgrad_kernel(const float *__restrict__ x, const float *__restrict__ dy, float *__restrict__ part, int F, int H,
                                                          int W, int Cin, int Cout, int rows_per_task, int n_rc, int n_cg, long long n_tasks) {
    const int lane = threadIdx.x & 63, l31 = lane & 31, lh = lane >> 5;
    const long long task = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (task >= n_tasks) return;  // whole waves leave together; no barrier below
    // task -> (row chunk, column group, Cout block); the row chunk varies fastest so that neighbouring waves share input rows
    const int rc = (int)(task % n_rc);
    const long long t2 = task / n_rc;
    const int cg = (int)(t2 % n_cg), cb = (int)(t2 / n_cg);
    int dyo[WG_NB], dxo[WG_NB], ci0[WG_NB];
    bool col_ok[WG_NB];
#pragma unroll
    for (int j = 0; j < WG_NB; ++j) {  // this lane's column of block j: tap and input channel
        const int col = (cg * WG_NB + j) * 32 + l31;
        col_ok[j] = col < 9 * Cin;
        const int tap = col_ok[j] ? col / Cin : 0;
        ci0[j] = col_ok[j] ? col - tap * Cin : 0;
        dyo[j] = tap / 3 - 1;
        dxo[j] = tap % 3 - 1;
    }
    f32x16 acc[WG_NB];
#pragma unroll
    for (int j = 0; j < WG_NB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const long long rows = (long long)F * H;
    const long long r0 = (long long)rc * rows_per_task;
    const long long r1 = r0 + rows_per_task < rows ? r0 + rows_per_task : rows;
    const int co = cb * 32 + l31;
    const bool co_ok = co < Cout;
    for (long long row = r0; row < r1; ++row) {
        const int f = (int)(row / H), y = (int)(row - (long long)f * H);
        const float *dyr = dy + row * W * Cout + (co_ok ? co : 0);
        const float *xr[WG_NB];
        bool row_ok[WG_NB];
#pragma unroll
        for (int j = 0; j < WG_NB; ++j) {
            const int yy = y + dyo[j];
            row_ok[j] = col_ok[j] && yy >= 0 && yy < H;
            xr[j] = x + (((long long)f * H + (row_ok[j] ? yy : y)) * W) * Cin + ci0[j];
        }
        // four pixel pairs per trip, every load of the trip issued before its first MFMA (the compiler does not unroll this loop itself)
        for (int xs = 0; xs < W; xs += 8) {
            float a[4], b[4][WG_NB];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int px = xs + 2 * u + lh;  // this lane-half's pixel of pair u
                a[u] = (px < W && co_ok) ? dyr[(long long)px * Cout] : 0.f;
#pragma unroll
                for (int j = 0; j < WG_NB; ++j) {
                    const int xx = px + dxo[j];
                    const bool ok = row_ok[j] && px < W && xx >= 0 && xx < W;
                    b[u][j] = ok ? xr[j][(long long)xx * Cin] : 0.f;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < WG_NB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u][j], acc[j], 0, 0, 0);
        }
    }
    float *dst = part + task * (WG_NB * 1024);
#pragma unroll
    for (int j = 0; j < WG_NB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[j * 1024 + r * 64 + lane] = acc[j][r];
}

// dW[co, ci, ky, kx] (+)= sum over row chunks.  A block = 32 consecutive (co, column) elements x 8 slices of the chunk range; each
// thread adds its slice in chunk order, the slices are added in slice order: a fixed order, hence deterministic.
__global__ __launch_bounds__(256) void conv3_wgrad_reduce_kernel(const float *__restrict__ part, float *__restrict__ dw, int Cin, int Cout, int n_rc, int n_cg,
                                                                 int accumulate) {
    __shared__ float red[8][32];
    const int e = threadIdx.x & 31, slice = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + e;  // (co, column): column = tap * Cin + ci
    const int K9 = 9 * Cin;
    const bool live = i < Cout * K9;
    float s = 0.f;
    int co = 0, col = 0;
    if (live) {
        co = i / K9;
        col = i - co * K9;
        const int cb = co >> 5, m = co & 31;
        const int b = col >> 5, n = col & 31;
        const int cg = b / WG_NB, j = b - cg * WG_NB;
        // accumulator element (m, n) of a 32x32 block: register r of lane (n, h) holds row (r & 3) + 8 (r >> 2) + 4 h
        const int h = (m >> 2) & 1, r = (m & 3) + 4 * (m >> 3);
        const long long t0 = ((long long)cb * n_cg + cg) * n_rc;
        const float *p = part + t0 * (WG_NB * 1024) + j * 1024 + r * 64 + h * 32 + n;
        const int per = (n_rc + 7) / 8, rc0 = slice * per, rc1 = rc0 + per < n_rc ? rc0 + per : n_rc;
#pragma unroll 8
        for (int rc = rc0; rc < rc1; ++rc) s += p[(long long)rc * (WG_NB * 1024)];
    }
    red[slice][e] = s;
    __syncthreads();
    if (slice == 0 && live) {
        float t = red[0][e];
#pragma unroll
        for (int k = 1; k < 8; ++k) t += red[k][e];
        const int tap = col / Cin, ci = col - tap * Cin;
        float *o = dw + ((long long)co * Cin + ci) * 9 + tap;
        *o = accumulate ? *o + t : t;
    }
}

// part[block, n] = sum over the block's share of rows of rowscale[m] * P[m, n]; P is walked as a flat array of float4 with a
// grid stride that is a multiple of N / 4, so every thread stays on one group of four columns.
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float *__restrict__ P, const float *__restrict__ rowscale, long long total4, int n4,
                                                             float *__restrict__ part) {
    __shared__ f32x4 red[256];
    const long long stride = (long long)gridDim.x * 256;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total4; i += stride) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(P + i * 4);
        acc += rowscale ? v * rowscale[i / n4] : v;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    if ((int)threadIdx.x < n4) {  // threads t, t + n4, t + 2 n4, ... share a column group (256 % n4 == 0)
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int t = threadIdx.x; t < 256; t += n4) s += red[t];
        *reinterpret_cast<f32x4 *>(part + ((long long)blockIdx.x * n4 + threadIdx.x) * 4) = s;
    }
}

// out[n] (+)= sum over blocks of part[block, n]: 32 columns x 8 slices of the block range per workgroup, fixed order.  fold != 0 adds the
// N columns into out[0] (a [M,1] input viewed as [M/4,4]; then N = 4 and there is one workgroup).
__global__ __launch_bounds__(256) void colsum_reduce_kernel(const float *__restrict__ part, int blocks, int N, int fold, int accumulate, float *__restrict__ out) {
    __shared__ float red[8][32];
    const int e = threadIdx.x & 31, slice = threadIdx.x >> 5;
    const int n = blockIdx.x * 32 + e;
    float s = 0.f;
    if (n < N) {
        const int per = (blocks + 7) / 8, b0 = slice * per, b1 = b0 + per < blocks ? b0 + per : blocks;
#pragma unroll 8
        for (int b = b0; b < b1; ++b) s += part[(long long)b * N + n];
    }
    red[slice][e] = s;
    __syncthreads();
    if (slice != 0 || n >= N) return;
    float t = red[0][e];
#pragma unroll
    for (int k = 1; k < 8; ++k) t += red[k][e];
    if (!fold) {
        out[n] = accumulate ? out[n] + t : t;
        return;
    }
    red[0][e] = t;  // fold: lanes 0..3 of one wave hold the four sums
    __builtin_amdgcn_wave_barrier();
    if (e == 0) {
        const float f = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        out[0] = accumulate ? out[0] + f : f;
    }
}

constexpr int COLSUM_BLOCKS = 512;

}  // namespace

static void wgrad_plan(int F, int H, int Cin, int Cout, int *rows_per_task, int *n_rc, int *n_cg, int *n_cb) {
    *n_cg = ((9 * Cin + 31) / 32 + WG_NB - 1) / WG_NB;
    *n_cb = (Cout + 31) / 32;
    const long long rows = (long long)F * H;
    // ~4 waves per SIMD of work (1024 SIMDs), at least one image row per wave; fewer, longer tasks = fewer pieces to reduce
    long long want = 4096 / ((long long)*n_cg * *n_cb);
    want = want < 1 ? 1 : want;
    long long rpt = (rows + want - 1) / want;
    rpt = rpt < 1 ? 1 : rpt;
    *rows_per_task = (int)rpt;
    *n_rc = (int)((rows + rpt - 1) / rpt);
}

size_t conv3_wgrad_workspace(int F, int H, int W, int Cin, int Cout) {
    if (F <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return 0;
    int rpt, n_rc, n_cg, n_cb;
    wgrad_plan(F, H, Cin, Cout, &rpt, &n_rc, &n_cg, &n_cb);
    return (size_t)n_rc * n_cg * n_cb * WG_NB * 1024;
}

int conv3_wgrad(const float *x, const float *dy, float *dw, int F, int H, int W, int Cin, int Cout, float *ws, size_t ws_floats, bool accumulate,
                hipStream_t st) {
    EDV_CHECK(x && dy && dw && ws, "null operand");
    EDV_CHECK(F > 0 && H > 0 && W > 0 && Cout > 0 && Cin > 0, "shape");
    int rpt, n_rc, n_cg, n_cb;
    wgrad_plan(F, H, Cin, Cout, &rpt, &n_rc, &n_cg, &n_cb);
    const long long tasks = (long long)n_rc * n_cg * n_cb;
    EDV_CHECK((size_t)tasks * WG_NB * 1024 <= ws_floats, "conv3_wgrad workspace too small (conv3_wgrad_workspace)");
    EDV_CHECK((tasks + 3) / 4 < (1ll << 31), "grid");
    EDV_LAUNCH(conv3_wgrad_kernel, dim3((unsigned)((tasks + 3) / 4)), dim3(256), 0, st, x, dy, ws, F, H, W, Cin, Cout, rpt, n_rc, n_cg, tasks);
    EDV_LAUNCH_OK();
    const int n = Cout * 9 * Cin;
    EDV_LAUNCH(conv3_wgrad_reduce_kernel, dim3((n + 31) / 32), dim3(256), 0, st, ws, dw, Cin, Cout, n_rc, n_cg, accumulate ? 1 : 0);
    EDV_LAUNCH_OK();
    return 0;
}

size_t colsum_workspace(int N) { return (size_t)COLSUM_BLOCKS * (N == 1 ? 4 : N); }

int colsum_rows(const float *P, const float *rowscale, long long M, int N, float *ws, size_t ws_floats, float *out, bool accumulate, hipStream_t st) {
    EDV_CHECK(P && ws && out && M > 0, "null operand");
    int fold = 0;
    if (N == 1) {  // a column vector: view it as [M / 4, 4] and fold the four sums
        EDV_CHECK(M % 4 == 0 && !rowscale, "colsum_rows with N = 1 needs M % 4 == 0 and no row scale");
        M /= 4;
        N = 4;
        fold = 1;
    }
    EDV_CHECK(N >= 4 && N <= 1024 && 1024 % N == 0, "colsum_rows needs N in {1, 4, 8, ..., 1024} (a power of two)");
    EDV_CHECK((size_t)COLSUM_BLOCKS * N <= ws_floats, "colsum_rows workspace too small (colsum_workspace)");
    const int n4 = N / 4;
    const long long total4 = M * n4;
    long long blocks = (total4 + 256 * 8 - 1) / (256 * 8);
    blocks = blocks < 1 ? 1 : (blocks > COLSUM_BLOCKS ? COLSUM_BLOCKS : blocks);
    EDV_LAUNCH(colsum_partial_kernel, dim3((unsigned)blocks), dim3(256), 0, st, P, rowscale, total4, n4, ws);
    EDV_LAUNCH_OK();
    EDV_LAUNCH(colsum_reduce_kernel, dim3((N + 31) / 32), dim3(256), 0, st, ws, (int)blocks, N, fold, accumulate ? 1 : 0, out);
    EDV_LAUNCH_OK();
    return 0;
}

}  // namespace edv
