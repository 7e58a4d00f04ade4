// The fine-tune step's photometric loss and its gradient with respect to the four disparity maps, fused.
//
// What it restates (reference trainer_end_to_end_video.py:808-868 generate_images_pred, :899-911 compute_reprojection_loss, :927-951 the
// per-scale sum; utils/layers.py:11-20 disp_to_depth, :134-189 BackprojectDepth / Project3D, :222-236 get_smooth_loss, :276-306 SSIM), with
// the relative poses and intrinsics as inputs (the pose network is outside the hot path):
//
//   total = 1/4 sum_s [ 1/2 sum_{nb in {prev, next}} mean_{kept frames, pixels} (0.85 mean_c SSIM_c(warp_nb, frame) + 0.15 mean_c |frame - warp_nb|)
//                       + lambda / 2^s (mean |d_x norm| e^{-|d_x img|} + mean |d_y norm| e^{-|d_y img|}) ],   norm = D / (mean_frame D + 1e-7)
//   D = disp_s resized to the frame size (bilinear, align_corners), depth = 1 / (1/max + (1/min - 1/max) D),
//   warp_nb(p) = bilinear sample (border padding, align_corners) of the neighbouring frame at the projection of pixel p's 3-D point.
//
// Why it exists: as ~500 eager PyTorch kernels per step (every one a bandwidth-bound pass over [T, 3, 518, 518]) the loss was 36 % of the
// ViT-S fine-tune step (17.4 of 48.5 ms; profiles/r02_c_train_vits_T8_kernel_stats.csv) -- the condition SURVEY.md §8f rank 4 sets for
// building it.  Here a scale costs seven launches: every tensor is read a few times and the 3x3 SSIM windows are tiled through LDS.
//
// The loss value AND dL/d disp_s come out of one call (a training step always wants both): each kernel is a gather, no float atomics, every
// reduction is two-stage in a fixed order -- results are reproducible run to run.
//
// Kernels per scale (N = B*T frames, P = H*W pixels):
//   bilinear (resample.hip)   D [N, P]                     (skipped when disp_s already has the frame size)
//   frame_sum_kernel x2       m[f] = mean_p D              (64 partials per frame, then one wave per frame)
//   smooth_kernel             g_sm [N, P] = dL_smooth / d norm, partial sums of the loss terms and of g_sm * D per frame
//   warp_kernel               x [2, N, 3, P] = the two warped neighbours
//   ssim_kernel               LDS tile 32x32 + halo 2 per (neighbour, frame, channel): loss partials, g_x = dL / d x
//   finish_kernel             per-frame S_f = sum g_sm D; loss += this scale's terms
//   warp_bwd_kernel           gD [N, P]: g_x through the sampling coordinates, the projection and the depth map, + the smoothness term
//   bilinear_bwd (bwd.hip)    dL/d disp_s                  (skipped when no resize)
#include <cmath>

#include "loss_common.hpp"

namespace edv {
namespace {

// x[nb][f][c][p]: the neighbouring frame sampled where pixel p of frame f lands (trainer :853-857)
__global__ __launch_bounds__(256) void warp_kernel(const float *__restrict__ D, const float *__restrict__ img, const Cam *__restrict__ cams, float *__restrict__ xw, int N,
                                                   int T, int H, int W, float da, float db) {
    const long long P = (long long)H * W;
    const int f = blockIdx.y, nb = blockIdx.z, t = f % T;
    if ((nb == 0 && t == 0) || (nb == 1 && t == T - 1)) return;  // no such neighbour inside the clip
    const float *src = img + (long long)(nb ? f + 1 : f - 1) * 3 * P;
    const Cam c = cams[2 * f + nb];
    float *o = xw + ((long long)nb * N + f) * 3 * P;
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < P; p += (long long)gridDim.x * 256) {
        const int y = (int)(p / W), x = (int)(p - (long long)y * W);
        const float depth = 1.0f / (da + db * D[(long long)f * P + p]);
        const Sample s = project_pixel(c, depth, x, y, H, W);
        const float fx0 = floorf(s.ix), fy0 = floorf(s.iy);
        const int x0 = (int)fx0, y0 = (int)fy0;
        const float wx = s.ix - fx0, wy = s.iy - fy0;
        const bool xin = x0 + 1 < W, yin = y0 + 1 < H;
        const float w00 = (1.f - wx) * (1.f - wy), w01 = wx * (1.f - wy), w10 = (1.f - wx) * wy, w11 = wx * wy;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const float *sc = src + ch * P + (long long)y0 * W + x0;
            float v = sc[0] * w00;
            if (xin) v += sc[1] * w01;
            if (yin) v += sc[W] * w10;
            if (xin && yin) v += sc[W + 1] * w11;
            o[ch * P + p] = v;
        }
    }
}

// One workgroup = (tile, channel, kept frame, neighbour).  Loss partial: sum over the tile of w_ssim * SSIM + w_l1 * |y - x|; g_x = dL/dx.
// SSIM(q) depends on x through the 3x3 window means mu_x, E[x^2], E[xy] (reflection-padded): with A, B, C = w_ssim * dSSIM/d(mu_x, E[x^2], E[xy]) at q,
// dL/dx_p = 1/9 sum_{q : p in window(q)} mult(p, q) (A_q + 2 x_p B_q + y_p C_q); mult counts the padded taps of window q that reflect onto p.
__global__ __launch_bounds__(256) void ssim_kernel(const float *__restrict__ xw, const float *__restrict__ img, float *__restrict__ gx, float *__restrict__ part,
                                                   const int *__restrict__ kept, int N, int H, int W, int tiles_x, float w_ssim, float w_l1) {
    __shared__ float sx[TI * TI], sy[TI * TI], cA[TC * TC], cB[TC * TC], cC[TC * TC];
    __shared__ float red[4];
    const long long P = (long long)H * W;
    const int tile = blockIdx.x, ch = blockIdx.y;
    const int nb = blockIdx.z & 1, f = kept[(blockIdx.z >> 1) * 2 + nb];
    const int ty0 = (tile / tiles_x) * TS, tx0 = (tile % tiles_x) * TS;
    const float *xs = xw + (((long long)nb * N + f) * 3 + ch) * P, *ys = img + ((long long)f * 3 + ch) * P;
    for (int i = threadIdx.x; i < TI * TI; i += 256) {
        const int r = i / TI, c = i - r * TI;
        int y = reflect1(ty0 - 2 + r, H), x = reflect1(tx0 - 2 + c, W);
        y = y < 0 ? 0 : (y >= H ? H - 1 : y);  // positions no window of an image pixel reaches (two rows out): any valid address
        x = x < 0 ? 0 : (x >= W ? W - 1 : x);
        sx[i] = xs[(long long)y * W + x];
        sy[i] = ys[(long long)y * W + x];
    }
    __syncthreads();
    float lsum = 0.f;
    for (int i = threadIdx.x; i < TC * TC; i += 256) {
        const int r = i / TC, c = i - r * TC;
        const int qy = ty0 - 1 + r, qx = tx0 - 1 + c;
        float A = 0.f, B = 0.f, C = 0.f;
        if (qy >= 0 && qy < H && qx >= 0 && qx < W) {
            float s_x = 0.f, s_y = 0.f, s_xx = 0.f, s_yy = 0.f, s_xy = 0.f;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const float a = sx[(r + dy) * TI + c + dx], b = sy[(r + dy) * TI + c + dx];
                    s_x += a; s_y += b; s_xx += a * a; s_yy += b * b; s_xy += a * b;
                }
            const float mu_x = s_x / 9.0f, mu_y = s_y / 9.0f;
            const float sig_x = s_xx / 9.0f - mu_x * mu_x, sig_y = s_yy / 9.0f - mu_y * mu_y, sig_xy = s_xy / 9.0f - mu_x * mu_y;
            const float n1 = 2.0f * mu_x * mu_y + SSIM_C1, n2 = 2.0f * sig_xy + SSIM_C2;
            const float d1 = mu_x * mu_x + mu_y * mu_y + SSIM_C1, d2 = sig_x + sig_y + SSIM_C2;
            const float n = n1 * n2, d = d1 * d2;
            const float raw = (1.0f - n / d) / 2.0f;
            const bool inner = r >= 1 && r <= TS && c >= 1 && c <= TS;  // q belongs to this tile (not to its halo)
            if (inner) lsum += w_ssim * fminf(1.0f, fmaxf(raw, 0.0f));
            if (raw >= 0.0f && raw <= 1.0f) {  // clamp passes the gradient on [0, 1]
                const float k = -0.5f * w_ssim / (d * d);
                const float dn_mu = 2.0f * mu_y * (n2 - n1), dd_mu = 2.0f * mu_x * (d2 - d1);
                A = k * (dn_mu * d - n * dd_mu);
                B = k * (-n * d1);
                C = k * (2.0f * n1 * d);
            }
        }
        cA[i] = A; cB[i] = B; cC[i] = C;
    }
    __syncthreads();
    float *go = gx + (((long long)nb * N + f) * 3 + ch) * P;
    for (int i = threadIdx.x; i < TS * TS; i += 256) {
        const int r = i / TS, c = i - r * TS;
        const int py = ty0 + r, px = tx0 + c;
        if (py >= H || px >= W) continue;
        float SA = 0.f, SB = 0.f, SC = 0.f;
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy) {
            const float my = 1.0f + ((dy == 1 && py == H - 2) || (dy == -1 && py == 1) ? 1.0f : 0.0f);
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const float m = my * (1.0f + ((dx == 1 && px == W - 2) || (dx == -1 && px == 1) ? 1.0f : 0.0f));
                const int j = (r + 1 + dy) * TC + c + 1 + dx;
                SA += m * cA[j]; SB += m * cB[j]; SC += m * cC[j];
            }
        }
        const float xv = sx[(r + 2) * TI + c + 2], yv = sy[(r + 2) * TI + c + 2];
        const float diff = yv - xv;
        lsum += w_l1 * fabsf(diff);
        go[(long long)py * W + px] = (SA + 2.0f * xv * SB + yv * SC) / 9.0f - w_l1 * (diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f));
    }
    lsum = wave_sum(lsum);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = lsum;
    __syncthreads();
    if (threadIdx.x == 0) part[((long long)blockIdx.z * 3 + ch) * gridDim.x + tile] = (red[0] + red[1]) + (red[2] + red[3]);
}

// One workgroup: S[f] = sum of the per-block g_sm * D partials; loss += sum of the SSIM / L1 partials + w_sm (sum t_x / Nx + sum t_y / Ny)
__global__ __launch_bounds__(256) void finish_kernel(const float *__restrict__ ssim_part, long long n_ssim, const float *__restrict__ sm_part, int sm_blocks, int N,
                                                     float *__restrict__ S, float *__restrict__ loss, float w_sm, float inv_nx, float inv_ny) {
    __shared__ float red[3][4];
    float a = 0.f, tx = 0.f, ty = 0.f;
    for (long long i = threadIdx.x; i < n_ssim; i += 256) a += ssim_part[i];
    for (int i = threadIdx.x; i < N * sm_blocks; i += 256) {
        tx += sm_part[(long long)i * 3];
        ty += sm_part[(long long)i * 3 + 1];
    }
    for (int f = threadIdx.x; f < N; f += 256) {
        float s = 0.f;
        for (int b = 0; b < sm_blocks; ++b) s += sm_part[((long long)f * sm_blocks + b) * 3 + 2];
        S[f] = s;
    }
    const float v[3] = {a, tx, ty};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float s = wave_sum(v[k]);
        if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float ssim = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        const float sx_ = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]), sy_ = (red[2][0] + red[2][1]) + (red[2][2] + red[2][3]);
        loss[0] += ssim + w_sm * (sx_ * inv_nx + sy_ * inv_ny);
    }
}

// gD[f][p] = dL/dD: the warped neighbours' gradients through the sampling coordinates (grid_sampler backward), the projection
// (utils/layers.py:176-189) and the depth map (:11-20), plus the smoothness term through norm = D / (mean + 1e-7)
__global__ __launch_bounds__(256) void warp_bwd_kernel(const float *__restrict__ D, const float *__restrict__ img, const Cam *__restrict__ cams, const float *__restrict__ gx,
                                                       const float *__restrict__ gsm, const float *__restrict__ mean, const float *__restrict__ S, float *__restrict__ gD,
                                                       int N, int T, int H, int W, float da, float db, float w_sm) {
    const long long P = (long long)H * W;
    const int f = blockIdx.y, t = f % T;
    const float den = mean[f] + 1e-7f;
    const float sm_all = S[f] / (den * den) / (float)P;
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < P; p += (long long)gridDim.x * 256) {
        const int y = (int)(p / W), x = (int)(p - (long long)y * W);
        const float depth = 1.0f / (da + db * D[(long long)f * P + p]);
        float gdepth = 0.f;
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            if ((nb == 0 && t == 0) || (nb == 1 && t == T - 1)) continue;
            const float *src = img + (long long)(nb ? f + 1 : f - 1) * 3 * P;
            const Sample s = project_pixel(cams[2 * f + nb], depth, x, y, H, W);
            const float fx0 = floorf(s.ix), fy0 = floorf(s.iy);
            const int x0 = (int)fx0, y0 = (int)fy0;
            const float wx = s.ix - fx0, wy = s.iy - fy0;
            const bool xin = x0 + 1 < W, yin = y0 + 1 < H;
            float gix = 0.f, giy = 0.f;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                const float *sc = src + ch * P + (long long)y0 * W + x0;
                const float v00 = sc[0], v01 = xin ? sc[1] : 0.f, v10 = yin ? sc[W] : 0.f, v11 = (xin && yin) ? sc[W + 1] : 0.f;
                const float g = gx[(((long long)nb * N + f) * 3 + ch) * P + p];
                gix += g * ((v01 - v00) * (1.f - wy) + (v11 - v10) * wy);
                giy += g * ((v10 - v00) * (1.f - wx) + (v11 - v01) * wx);
            }
            gix *= s.mx;
            giy *= s.my;
            const float zi = s.Z + 1e-7f;
            // u = X / (Z + eps), d u / d depth = rx / zi - X rz / zi^2   (ix = u up to rounding: grid normalisation and un-normalisation cancel)
            gdepth += gix * (s.rx / zi - s.X * s.rz / (zi * zi)) + giy * (s.ry / zi - s.Y * s.rz / (zi * zi));
        }
        gD[(long long)f * P + p] = gdepth * (-db * depth * depth) + w_sm * (gsm[(long long)f * P + p] / den - sm_all);
    }
}

__global__ void kept_kernel(int *kept, int B, int T) {  // kept[k] = {frame with a previous neighbour, frame with a next neighbour}, k < B (T - 1)
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= B * (T - 1)) return;
    const int b = k / (T - 1), i = k - b * (T - 1);
    kept[2 * k] = b * T + i + 1;
    kept[2 * k + 1] = b * T + i;
}

struct LossWs {  // carve-up of the caller's workspace (floats)
    size_t cams, kept, D, mean, S, sum_part, sm_part, ssim_part, gsm, gD, xw, gx, total;
};
LossWs loss_layout(int B, int T, int H, int W) {
    const size_t N = (size_t)B * T, P = (size_t)H * W;
    const size_t tiles = (size_t)((H + TS - 1) / TS) * ((W + TS - 1) / TS);
    auto up = [](size_t n) { return (n + 3) & ~(size_t)3; };
    LossWs w;
    size_t o = 0;
    w.cams = o; o += up(2 * N * (sizeof(Cam) / sizeof(float)));
    w.kept = o; o += up(2 * (size_t)B * (T - 1));
    w.D = o; o += up(N * P);
    w.mean = o; o += up(N);
    w.S = o; o += up(N);
    w.sum_part = o; o += up(N * SUM_PARTS);
    w.sm_part = o; o += up(N * 64 * 3);
    w.ssim_part = o; o += up(2 * (size_t)B * (T - 1) * 3 * tiles);
    w.gsm = o; o += up(N * P);
    w.gD = o; o += up(N * P);
    w.xw = o; o += up(2 * N * 3 * P);
    w.gx = o; o += up(2 * N * 3 * P);
    w.total = o;
    return w;
}

}  // namespace

size_t photometric_loss_workspace(int B, int T, int H, int W) {
    if (B <= 0 || T < 2 || H < 3 || W < 3) return 0;
    return loss_layout(B, T, H, W).total;
}

int photometric_loss(const float *frames, const float *const disp[4], const int *dh, const int *dw, int B, int T, int H, int W, const float *K, const float *invK,
                     const float *Tprev, const float *Tnext, float min_depth, float max_depth, float smoothness, float *loss, float *const grad[4], float *ws,
                     size_t ws_floats, hipStream_t st) {
    EDV_CHECK(frames && disp && dh && dw && K && invK && Tprev && Tnext && loss && grad && ws, "null argument");
    EDV_CHECK(B > 0 && T >= 2, "the photometric loss needs clips of at least two frames");
    EDV_CHECK(H >= 3 && W >= 3, "frames smaller than the SSIM window");
    EDV_CHECK(min_depth > 0.f && max_depth > min_depth, "depth range");
    const LossWs L = loss_layout(B, T, H, W);
    EDV_CHECK(ws_floats >= L.total && (uintptr_t)ws % 16 == 0, "workspace too small (photometric_loss_workspace)");
    const int N = B * T;
    EDV_CHECK(N <= 65535, "too many frames");
    const long long P = (long long)H * W;
    const int kept_n = B * (T - 1);
    Cam *cams = reinterpret_cast<Cam *>(ws + L.cams);
    int *kept = reinterpret_cast<int *>(ws + L.kept);
    float *Dbuf = ws + L.D, *mean = ws + L.mean, *S = ws + L.S, *sum_part = ws + L.sum_part, *sm_part = ws + L.sm_part, *ssim_part = ws + L.ssim_part;
    float *gsm = ws + L.gsm, *gD = ws + L.gD, *xw = ws + L.xw, *gx = ws + L.gx;
    const float da = (float)(1.0 / (double)max_depth), db = (float)(1.0 / (double)min_depth - 1.0 / (double)max_depth);  // utils/layers.py:16-18
    const int tiles_x = (W + TS - 1) / TS, tiles = tiles_x * ((H + TS - 1) / TS);
    const int pix_blocks = (int)((P + 255) / 256 < 1024 ? (P + 255) / 256 : 1024);
    const int sm_blocks = 64;
    const float cnt = (float)((double)kept_n * (double)P);
    const float inv_nx = (float)(1.0 / ((double)N * H * (W - 1))), inv_ny = (float)(1.0 / ((double)N * (H - 1) * W));

    EDV_HIP(hipMemsetAsync(loss, 0, sizeof(float), st));
    EDV_LAUNCH(cam_kernel, dim3((2 * N + 63) / 64), dim3(64), 0, st, K, invK, Tprev, Tnext, cams, N);
    EDV_LAUNCH_OK();
    EDV_LAUNCH(kept_kernel, dim3((kept_n + 63) / 64), dim3(64), 0, st, kept, B, T);
    EDV_LAUNCH_OK();
    for (int s = 0; s < 4; ++s) {
        EDV_CHECK(disp[s] && grad[s] && dh[s] > 0 && dw[s] > 0, "bad disparity map");
        const bool same = dh[s] == H && dw[s] == W;
        const float *D = disp[s];
        if (!same) {  // F.interpolate(disp, [H, W], mode="bilinear", align_corners=True) (trainer :813-817, :931-933)
            EDV_TRY(bilinear(disp[s], Dbuf, N, dh[s], dw[s], 1, H, W, ACT_NONE, st));
            D = Dbuf;
        }
        // weights of this scale's terms in the total (trainer :948-966: / 2 per neighbour pair, smoothness / 2^s, mean over the 4 scales)
        const float w_rep = 0.25f * 0.5f / cnt, w_sm = 0.25f * smoothness / (float)(1 << s);
        EDV_LAUNCH(frame_sum_kernel, dim3(SUM_PARTS, N), dim3(256), 0, st, D, sum_part, P);
        EDV_LAUNCH_OK();
        EDV_LAUNCH(frame_sum_finish_kernel, dim3(N), dim3(64), 0, st, sum_part, mean, (float)(1.0 / (double)P));
        EDV_LAUNCH_OK();
        EDV_LAUNCH(smooth_kernel, dim3(sm_blocks, N), dim3(256), 0, st, D, frames, mean, gsm, sm_part, H, W, inv_nx, inv_ny);
        EDV_LAUNCH_OK();
        EDV_LAUNCH(warp_kernel, dim3(pix_blocks, N, 2), dim3(256), 0, st, D, frames, cams, xw, N, T, H, W, da, db);
        EDV_LAUNCH_OK();
        EDV_LAUNCH(ssim_kernel, dim3(tiles, 3, 2 * kept_n), dim3(256), 0, st, xw, frames, gx, ssim_part, kept, N, H, W, tiles_x, w_rep * 0.85f / 3.0f,
                           w_rep * 0.15f / 3.0f);
        EDV_LAUNCH_OK();
        EDV_LAUNCH(finish_kernel, dim3(1), dim3(256), 0, st, ssim_part, (long long)2 * kept_n * 3 * tiles, sm_part, sm_blocks, N, S, loss, w_sm, inv_nx, inv_ny);
        EDV_LAUNCH_OK();
        float *gdst = same ? grad[s] : gD;
        EDV_LAUNCH(warp_bwd_kernel, dim3(pix_blocks, N), dim3(256), 0, st, D, frames, cams, gx, gsm, mean, S, gdst, N, T, H, W, da, db, w_sm);
        EDV_LAUNCH_OK();
        if (!same) EDV_TRY(bilinear_bwd(gD, grad[s], N, dh[s], dw[s], 1, H, W, false, st));
    }
    return 0;
}

}  // namespace edv
