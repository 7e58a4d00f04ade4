// Pieces shared by the two fused losses (loss.hip: the round-2 photometric subset; loss_trainer.hip: the trainer's whole loss): the camera
// geometry of utils/layers.py:134-189, the SSIM tile constants, per-frame sums and the edge-aware smoothness kernel.  Everything sits in an anonymous
// namespace: each translation unit gets its own copy of the kernels.
#pragma once
#include <cmath>

#include "ops.hpp"

namespace edv {
namespace {

constexpr int TS = 32;            // SSIM output tile (TS x TS pixels per workgroup)
constexpr int TI = TS + 4;        // input tile with halo 2
constexpr int TC = TS + 2;        // coefficient tile with halo 1
constexpr int SUM_PARTS = 64;     // partial sums per frame
constexpr float SSIM_C1 = 0.01f * 0.01f, SSIM_C2 = 0.03f * 0.03f;

struct Cam {  // per frame and neighbour: P = (K T)[:3, :] (row-major 3x4) and inv_K[:3, :3]
    float P[12];
    float iK[9];
};

// P[f][nb] = (K[f] @ T_nb[f])[:3, :], iK = inv_K[f][:3, :3]  (utils/layers.py:177, :168)
__global__ void cam_kernel(const float *__restrict__ K, const float *__restrict__ invK, const float *__restrict__ Tp, const float *__restrict__ Tn, Cam *cams, int N) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * N) return;
    const int f = i >> 1, nb = i & 1;
    const float *k = K + (long long)f * 16, *t = (nb ? Tn : Tp) + (long long)f * 16, *ik = invK + (long long)f * 16;
    Cam c;
    for (int r = 0; r < 3; ++r)
        for (int col = 0; col < 4; ++col) {
            float s = 0.f;
            for (int j = 0; j < 4; ++j) s += k[r * 4 + j] * t[j * 4 + col];
            c.P[r * 4 + col] = s;
        }
    for (int r = 0; r < 3; ++r)
        for (int col = 0; col < 3; ++col) c.iK[r * 3 + col] = ik[r * 4 + col];
    cams[i] = c;
}

// partial[f][b] = sum of x[f][chunk b]; then out[f] = scale * sum_b partial[f][b]
__global__ __launch_bounds__(256) void frame_sum_kernel(const float *__restrict__ x, float *__restrict__ partial, long long P) {
    __shared__ float red[4];
    const int f = blockIdx.y, b = blockIdx.x;
    const long long per = (P + SUM_PARTS - 1) / SUM_PARTS, p0 = b * per, p1 = p0 + per < P ? p0 + per : P;
    float s = 0.f;
    for (long long p = p0 + threadIdx.x; p < p1; p += 256) s += x[(long long)f * P + p];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[f * SUM_PARTS + b] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(64) void frame_sum_finish_kernel(const float *__restrict__ partial, float *__restrict__ out, float scale) {
    const int f = blockIdx.x;
    const float s = wave_sum(partial[f * SUM_PARTS + threadIdx.x]);
    if (threadIdx.x == 0) out[f] = s * scale;
}

// Edge-aware smoothness of the mean-normalised disparity (utils/layers.py:222-236 on norm = D / (mean + 1e-7), trainer :944-946).
// Thread = pixel p.  Loss terms: pairs (p, p + 1x), (p, p + 1y).  Gradient (a gather): d / d norm_p of the four pairs p is part of.
// part[f][b] = {sum t_x, sum t_y, sum g_p D_p} of the block's pixels; g_p = d (sum t_x / Nx + sum t_y / Ny) / d norm_p.
__global__ __launch_bounds__(256) void smooth_kernel(const float *__restrict__ D, const float *__restrict__ img, const float *__restrict__ mean, float *__restrict__ gsm,
                                                     float *__restrict__ part, int H, int W, float inv_nx, float inv_ny) {
    __shared__ float red[3][4];
    const int f = blockIdx.y;
    const long long P = (long long)H * W;
    const float *Df = D + (long long)f * P, *im = img + (long long)f * 3 * P;
    const float den = mean[f] + 1e-7f;
    float tx = 0.f, ty = 0.f, gd = 0.f;
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < P; p += (long long)gridDim.x * 256) {
        const int y = (int)(p / W), x = (int)(p - (long long)y * W);
        const float n0 = Df[p] / den;
        const float c0 = im[p], c1 = im[P + p], c2 = im[2 * P + p];
        float g = 0.f;
        if (x + 1 < W) {
            const float w = expf(-((fabsf(c0 - im[p + 1]) + fabsf(c1 - im[P + p + 1]) + fabsf(c2 - im[2 * P + p + 1])) / 3.0f));
            const float d = n0 - Df[p + 1] / den;
            tx += fabsf(d) * w;
            g += (d > 0.f ? w : (d < 0.f ? -w : 0.f)) * inv_nx;
        }
        if (x > 0) {
            const float w = expf(-((fabsf(im[p - 1] - c0) + fabsf(im[P + p - 1] - c1) + fabsf(im[2 * P + p - 1] - c2)) / 3.0f));
            const float d = Df[p - 1] / den - n0;
            g -= (d > 0.f ? w : (d < 0.f ? -w : 0.f)) * inv_nx;
        }
        if (y + 1 < H) {
            const float w = expf(-((fabsf(c0 - im[p + W]) + fabsf(c1 - im[P + p + W]) + fabsf(c2 - im[2 * P + p + W])) / 3.0f));
            const float d = n0 - Df[p + W] / den;
            ty += fabsf(d) * w;
            g += (d > 0.f ? w : (d < 0.f ? -w : 0.f)) * inv_ny;
        }
        if (y > 0) {
            const float w = expf(-((fabsf(im[p - W] - c0) + fabsf(im[P + p - W] - c1) + fabsf(im[2 * P + p - W] - c2)) / 3.0f));
            const float d = Df[p - W] / den - n0;
            g -= (d > 0.f ? w : (d < 0.f ? -w : 0.f)) * inv_ny;
        }
        gsm[(long long)f * P + p] = g;  // d (t_x / Nx + t_y / Ny) / d norm_p
        gd += g * Df[p];
    }
    const float v[3] = {tx, ty, gd};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float s = wave_sum(v[k]);
        if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = s;
    }
    __syncthreads();
    if (threadIdx.x < 3) part[((long long)f * gridDim.x + blockIdx.x) * 3 + threadIdx.x] = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
}

// Sampling geometry of pixel (x, y) of frame f towards neighbour nb (utils/layers.py:166-189 + F.grid_sample, border padding, align_corners).
struct Sample {
    float ix, iy;      // clipped source coordinates
    float mx, my;      // 1 where the coordinate was not clipped (d ix / d u), else 0
    float X, Y, Z;     // projected point (before the division)
    float rx, ry, rz;  // P3 . ray: d (X, Y, Z) / d depth
};
__device__ __forceinline__ Sample project_pixel(const Cam &c, float depth, int x, int y, int H, int W) {
    const float fx = (float)x, fy = (float)y;
    const float r0 = c.iK[0] * fx + c.iK[1] * fy + c.iK[2], r1 = c.iK[3] * fx + c.iK[4] * fy + c.iK[5], r2 = c.iK[6] * fx + c.iK[7] * fy + c.iK[8];
    const float c0 = depth * r0, c1 = depth * r1, c2 = depth * r2;
    Sample s;
    s.X = c.P[0] * c0 + c.P[1] * c1 + c.P[2] * c2 + c.P[3];
    s.Y = c.P[4] * c0 + c.P[5] * c1 + c.P[6] * c2 + c.P[7];
    s.Z = c.P[8] * c0 + c.P[9] * c1 + c.P[10] * c2 + c.P[11];
    s.rx = c.P[0] * r0 + c.P[1] * r1 + c.P[2] * r2;
    s.ry = c.P[4] * r0 + c.P[5] * r1 + c.P[6] * r2;
    s.rz = c.P[8] * r0 + c.P[9] * r1 + c.P[10] * r2;
    const float zi = s.Z + 1e-7f;
    float gx = (s.X / zi / (float)(W - 1) - 0.5f) * 2.0f, gy = (s.Y / zi / (float)(H - 1) - 0.5f) * 2.0f;
    float ix = (gx + 1.0f) / 2.0f * (float)(W - 1), iy = (gy + 1.0f) / 2.0f * (float)(H - 1);
    // clip_coordinates_set_grad (GridSampler.cuh): the gradient passes only strictly inside (0, size - 1)
    s.mx = (ix > 0.f && ix < (float)(W - 1)) ? 1.f : 0.f;
    s.my = (iy > 0.f && iy < (float)(H - 1)) ? 1.f : 0.f;
    s.ix = fminf((float)(W - 1), fmaxf(ix, 0.f));
    s.iy = fminf((float)(H - 1), fmaxf(iy, 0.f));
    return s;
}

__device__ __forceinline__ int reflect1(int i, int n) {  // nn.ReflectionPad2d(1): -1 -> 1, n -> n - 2
    i = i < 0 ? -i : i;
    return i >= n ? 2 * (n - 1) - i : i;
}

__global__ __launch_bounds__(256) void scale_kernel(float *__restrict__ x, long long n, float a) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) x[i] *= a;
}

}  // namespace
}  // namespace edv
