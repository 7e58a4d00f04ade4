// The trainer's whole self-supervised loss and every gradient its autograd reaches, fused.
//
// What it restates: generate_images_pred + compute_losses of the reference (trainer_end_to_end_video.py:808-898, :913-971) with the side networks'
// outputs as inputs -- per scale s and neighbour fid in {-1, +1} (N = B*T flattened frames, every frame has both neighbours: the dataset hands the
// trainer T + 2 frames, datasets/scared_video_dataset.py:289-296):
//
//   rep   sum(m (0.85 mean_c SSIM_c + 0.15 mean_c L1)(warp_fid, refined(s, fid))) / sum(m)          m = occu_mask_backward(0, fid), detached   (:940-941)
//   tr    sum(m mean_c |refined(s, fid) - registration(0, fid)|) / sum(m)                                                               (:942-943)
//   cvt   get_smooth_bright(transform_high(s, fid), color(0, 0), registration(s, fid), m)                       utils/layers.py:239-264 (:944-945)
//   drp   mean over {sampled > 1e-3} of |Z of frame i's points in camera i+fid - depth(i+fid) sampled at their projection|  (zeros padding) (:863-877)
//   dfl   mean over {warp > 1e-3} of |depth(i) sampled at p + flow(s, fid)(i, p) - depth(i+fid)(p)|              utils/layers.py:387-426 (:879-890)
//   sm    get_smooth_loss(D_c / (mean + 1e-7), color(0, s)),  D_c = disp_s at the size of color(0, s)                               (:929-933, :949-951)
//   loss_s = rep / 2 + tc tr / 2 + ts cvt / 2 + ds sm / 2^s + tw (dr drp / 2 + df dfl / 2);   total = mean over the four scales         (:953-968)
//
// Gradients: the four disparity maps; on request also refined, transform_high, the poses T(fid) and -- the reference learns its intrinsics by
// default (options.py:94-97) -- K and inv_K.  registration and the mask are detached in the reference and get none.
//
// Structure per scale (P = H*W):
//   bilinear x2              D = disp_s at the frame size (the geometry), D_c at the size of color(0, s) (the smoothness term)
//   frame_sum x2, smooth     as loss.hip, at the size of color(0, s)
//   warp_nb_kernel           x[fid][f][c][p]: the neighbour image sampled at the projection of every pixel (border padding)
//   ssimw_kernel             32x32 tiles + halo in LDS: weighted SSIM + L1 against refined, the loss_transform term, dL/dx and dL/d refined (gather form)
//   cvt_kernel               get_smooth_bright and dL/d transform_high
//   dreproj_fwd / dflow_fwd  sums and counts of the two depth-consistency terms (their means need the counts before any gradient)
//   geom_bwd_kernel          dL/dx through the sampling coordinates, the projection and the depth; the depth-reprojection term's gradient on the source
//                            side; per-frame partial sums of dL/dP (3x4 per neighbour) and dL/d inv_K (3x3)
//   dflow_bwd_kernel         the flow term's gradient
//   depth_grad_kernel        dL/dD from dL/d depth;  bilinear_bwd -> dL/d disp_s (both resize paths accumulate)
//   pose_finish_kernel       dL/dK, dL/d inv_K, dL/dT from the partial sums
// Everything is a gather with fixed-order two-stage sums EXCEPT the gradient that the two depth-consistency terms send into the SAMPLED depth map
// (a bilinear scatter, float atomics, like ATen's grid_sampler backward): with depth_reproj = depth_flow = 0 or tune_temporal off -- the options'
// defaults -- no atomic runs and the call is bit-reproducible.
#include <cmath>

#include "loss_common.hpp"

namespace edv {
namespace {

// ---- colour warp of an explicit neighbour image -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void warp_nb_kernel(const float *__restrict__ D, const float *__restrict__ nb_img, const Cam *__restrict__ cams, int nb,
                                                      float *__restrict__ xw, int H, int W, float da, float db) {
    const long long P = (long long)H * W;
    const int f = blockIdx.y;
    const float *src = nb_img + (long long)f * 3 * P;
    const Cam c = cams[2 * f + nb];
    float *o = xw + (long long)f * 3 * P;
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

// ---- weighted SSIM + L1 against refined, + the loss_transform term ---------------------------------------------------------------------------------
// One workgroup = (tile, channel, frame).  w_q = w_rep m_q / M: the per-pixel weight of the reprojection term; w_tr m_p / M that of |refined - registration_0|.
// SSIM(q) depends on x AND y through the window means: A, B, C = w_q * dSSIM/d(mu_x, E[x^2], E[xy]), A2 = w_q dSSIM/d mu_y (B serves E[y^2] too: the
// expression is symmetric in sigma_x + sigma_y).  part[(f * 3 + ch) * tiles + tile] = {reprojection partial, transform partial}.
__global__ __launch_bounds__(256) void ssimw_kernel(const float *__restrict__ xw, const float *__restrict__ refined, const float *__restrict__ reg0,
                                                    const float *__restrict__ mask, const float *__restrict__ msum, float *__restrict__ gx, float *__restrict__ gy,
                                                    float *__restrict__ part, int H, int W, int tiles_x, float w_rep, float w_tr) {
    __shared__ float sx[TI * TI], sy[TI * TI], cA[TC * TC], cA2[TC * TC], cB[TC * TC], cC[TC * TC];
    __shared__ float red[2][4];
    const long long P = (long long)H * W;
    const int tile = blockIdx.x, ch = blockIdx.y, f = blockIdx.z;
    const int ty0 = (tile / tiles_x) * TS, tx0 = (tile % tiles_x) * TS;
    const float *xs = xw + ((long long)f * 3 + ch) * P, *ys = refined + ((long long)f * 3 + ch) * P, *mf = mask + (long long)f * P;
    const float inv_m = 1.0f / msum[0];
    const float w_ssim = w_rep * 0.85f / 3.0f * inv_m, w_l1 = w_rep * 0.15f / 3.0f * inv_m, w_t = w_tr / 3.0f * inv_m;
    for (int i = threadIdx.x; i < TI * TI; i += 256) {
        const int r = i / TI, c = i - r * TI;
        int y = reflect1(ty0 - 2 + r, H), x = reflect1(tx0 - 2 + c, W);
        y = y < 0 ? 0 : (y >= H ? H - 1 : y);
        x = x < 0 ? 0 : (x >= W ? W - 1 : x);
        sx[i] = xs[(long long)y * W + x];
        sy[i] = ys[(long long)y * W + x];
    }
    __syncthreads();
    float lrep = 0.f, ltr = 0.f;
    for (int i = threadIdx.x; i < TC * TC; i += 256) {
        const int r = i / TC, c = i - r * TC;
        const int qy = ty0 - 1 + r, qx = tx0 - 1 + c;
        float A = 0.f, A2 = 0.f, B = 0.f, C = 0.f;
        if (qy >= 0 && qy < H && qx >= 0 && qx < W) {
            const float wq = w_ssim * mf[(long long)qy * W + qx];
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
            const bool inner = r >= 1 && r <= TS && c >= 1 && c <= TS;
            if (inner) lrep += wq * fminf(1.0f, fmaxf(raw, 0.0f));
            if (raw >= 0.0f && raw <= 1.0f) {
                const float k = -0.5f * wq / (d * d);
                A = k * (2.0f * mu_y * (n2 - n1) * d - n * 2.0f * mu_x * (d2 - d1));
                A2 = k * (2.0f * mu_x * (n2 - n1) * d - n * 2.0f * mu_y * (d2 - d1));
                B = k * (-n * d1);
                C = k * (2.0f * n1 * d);
            }
        }
        cA[i] = A; cA2[i] = A2; cB[i] = B; cC[i] = C;
    }
    __syncthreads();
    float *gox = gx + ((long long)f * 3 + ch) * P, *goy = gy ? gy + ((long long)f * 3 + ch) * P : nullptr;
    const float *rg = reg0 + ((long long)f * 3 + ch) * P;
    for (int i = threadIdx.x; i < TS * TS; i += 256) {
        const int r = i / TS, c = i - r * TS;
        const int py = ty0 + r, px = tx0 + c;
        if (py >= H || px >= W) continue;
        float SA = 0.f, SA2 = 0.f, SB = 0.f, SC = 0.f;
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy) {
            const float my = 1.0f + ((dy == 1 && py == H - 2) || (dy == -1 && py == 1) ? 1.0f : 0.0f);
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const float m = my * (1.0f + ((dx == 1 && px == W - 2) || (dx == -1 && px == 1) ? 1.0f : 0.0f));
                const int j = (r + 1 + dy) * TC + c + 1 + dx;
                SA += m * cA[j]; SA2 += m * cA2[j]; SB += m * cB[j]; SC += m * cC[j];
            }
        }
        const float xv = sx[(r + 2) * TI + c + 2], yv = sy[(r + 2) * TI + c + 2];
        const float mp = mf[(long long)py * W + px];
        const float diff = yv - xv;
        const float sgn = diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f);
        lrep += w_l1 * mp * fabsf(diff);
        gox[(long long)py * W + px] = (SA + 2.0f * xv * SB + yv * SC) / 9.0f - w_l1 * mp * sgn;
        const float dt = yv - rg[(long long)py * W + px];
        ltr += w_t * mp * fabsf(dt);
        if (goy) goy[(long long)py * W + px] = (SA2 + 2.0f * yv * SB + xv * SC) / 9.0f + w_l1 * mp * sgn + w_t * mp * (dt > 0.f ? 1.f : (dt < 0.f ? -1.f : 0.f));
    }
    const float v[2] = {lrep, ltr};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const float s = wave_sum(v[k]);
        if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = s;
    }
    __syncthreads();
    if (threadIdx.x < 2) part[(((long long)f * 3 + ch) * gridDim.x + tile) * 2 + threadIdx.x] = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
}

// ---- get_smooth_bright (utils/layers.py:239-264) and dL/d transform_high ---------------------------------------------------------------------------
// Thread = pixel.  Pairs (p, p + 1x) weigh e^{-mean_c |res_p - res_{p+1x}|} m_p / Mx, pairs (p, p + 1y) likewise with My; res = color_0 - registration_s.
// The loss terms are counted at the pair's first pixel; the gradient of t_c(p) gathers the four pairs p belongs to.  part[f][block] = partial sum.
__global__ __launch_bounds__(256) void cvt_kernel(const float *__restrict__ tr, const float *__restrict__ color0, const float *__restrict__ reg, const float *__restrict__ mask,
                                                  const float *__restrict__ msum, float *__restrict__ gtr, float *__restrict__ part, int H, int W, float w_cvt) {
    __shared__ float red[4];
    const int f = blockIdx.y;
    const long long P = (long long)H * W;
    const float *t = tr + (long long)f * 3 * P, *c0 = color0 + (long long)f * 3 * P, *rg = reg + (long long)f * 3 * P, *m = mask + (long long)f * P;
    const float wx_ = w_cvt / msum[1], wy_ = w_cvt / msum[2];
    float acc = 0.f;
    auto edge = [&](long long a, long long b) {  // e^{-mean_c |res_a - res_b|}
        const float r = fabsf((c0[a] - rg[a]) - (c0[b] - rg[b])) + fabsf((c0[P + a] - rg[P + a]) - (c0[P + b] - rg[P + b])) +
                        fabsf((c0[2 * P + a] - rg[2 * P + a]) - (c0[2 * P + b] - rg[2 * P + b]));
        return expf(-(r / 3.0f));
    };
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < P; p += (long long)gridDim.x * 256) {
        const int y = (int)(p / W), x = (int)(p - (long long)y * W);
        float g[3] = {0.f, 0.f, 0.f};
        if (x + 1 < W) {
            const float e = edge(p, p + 1) * m[p] * wx_;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                const float d = t[ch * P + p] - t[ch * P + p + 1];
                acc += fabsf(d) * e / 3.0f;
                g[ch] += (d > 0.f ? e : (d < 0.f ? -e : 0.f)) / 3.0f;
            }
        }
        if (x > 0) {
            const float e = edge(p - 1, p) * m[p - 1] * wx_;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                const float d = t[ch * P + p - 1] - t[ch * P + p];
                g[ch] -= (d > 0.f ? e : (d < 0.f ? -e : 0.f)) / 3.0f;
            }
        }
        if (y + 1 < H) {
            const float e = edge(p, p + W) * m[p] * wy_;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                const float d = t[ch * P + p] - t[ch * P + p + W];
                acc += fabsf(d) * e / 3.0f;
                g[ch] += (d > 0.f ? e : (d < 0.f ? -e : 0.f)) / 3.0f;
            }
        }
        if (y > 0) {
            const float e = edge(p - W, p) * m[p - W] * wy_;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                const float d = t[ch * P + p - W] - t[ch * P + p];
                g[ch] -= (d > 0.f ? e : (d < 0.f ? -e : 0.f)) / 3.0f;
            }
        }
        if (gtr) {
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) gtr[((long long)f * 3 + ch) * P + p] = g[ch];
        }
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[(long long)f * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// sums of the mask the normalisations need: out[0] = sum m, out[1] = sum m[:, :, :, :-1], out[2] = sum m[:, :, :-1, :]   (MASK_BLOCKS workgroups per frame, then one wave)
constexpr int MASK_BLOCKS = 32;
__global__ __launch_bounds__(256) void mask_sums_kernel(const float *__restrict__ mask, float *__restrict__ part, int H, int W) {
    __shared__ float red[3][4];
    const int f = blockIdx.y;
    const long long P = (long long)H * W;
    const float *m = mask + (long long)f * P;
    float a = 0.f, bx = 0.f, by = 0.f;
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < P; p += (long long)MASK_BLOCKS * 256) {
        const int y = (int)(p / W), x = (int)(p - (long long)y * W);
        const float v = m[p];
        a += v;
        if (x + 1 < W) bx += v;
        if (y + 1 < H) by += v;
    }
    const float v[3] = {a, bx, by};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float s = wave_sum(v[k]);
        if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = s;
    }
    __syncthreads();
    if (threadIdx.x < 3) part[((long long)f * MASK_BLOCKS + blockIdx.x) * 3 + threadIdx.x] = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
}
__global__ __launch_bounds__(64) void mask_sums_finish_kernel(const float *__restrict__ part, float *__restrict__ out, int n) {
    float a = 0.f, b = 0.f, c = 0.f;
    for (int f = threadIdx.x; f < n; f += 64) {
        a += part[f * 3];
        b += part[f * 3 + 1];
        c += part[f * 3 + 2];
    }
    a = wave_sum(a); b = wave_sum(b); c = wave_sum(c);
    if (threadIdx.x == 0) { out[0] = a; out[1] = b; out[2] = c; }
}

// x *= a over a list of tensors in one launch (blockIdx.y = tensor)
constexpr int SCALE_LIST = 32;
struct ScaleList {
    float *p[SCALE_LIST];
    long long n[SCALE_LIST];
};
__global__ __launch_bounds__(256) void scale_list_kernel(ScaleList l, float a) {
    float *p = l.p[blockIdx.y];
    const long long n = l.n[blockIdx.y];
    if ((n & 3) == 0 && ((uintptr_t)p & 15) == 0) {
        float4 *p4 = reinterpret_cast<float4 *>(p);
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n / 4; i += (long long)gridDim.x * 256) {
            float4 v = p4[i];
            v.x *= a; v.y *= a; v.z *= a; v.w *= a;
            p4[i] = v;
        }
    } else {
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) p[i] *= a;
    }
}

// ---- depth-consistency terms ---------------------------------------------------------------------------------------------------------------------------
// bilinear sample of a depth map (given as the disparity-like D: depth = 1 / (da + db D)) with ZEROS padding, un-clipped coordinates
struct ZSample {
    float val, dix, diy;       // value, d val / d ix, d val / d iy
    int x0, y0;
    float w[4];                // corner weights (0 for a corner outside the map)
    bool in[4];
};
__device__ __forceinline__ ZSample sample_depth_zeros(const float *__restrict__ Dj, float ix, float iy, int H, int W, float da, float db) {
    ZSample z;
    z.val = z.dix = z.diy = 0.f;
    z.x0 = z.y0 = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { z.w[k] = 0.f; z.in[k] = false; }
    if (!(ix > -1.0f && ix < (float)W && iy > -1.0f && iy < (float)H)) return z;  // every corner outside (also NaN)
    const float fx0 = floorf(ix), fy0 = floorf(iy);
    const int x0 = (int)fx0, y0 = (int)fy0;
    const float wx = ix - fx0, wy = iy - fy0;
    z.x0 = x0; z.y0 = y0;
    const bool xi0 = x0 >= 0 && x0 < W, xi1 = x0 + 1 >= 0 && x0 + 1 < W, yi0 = y0 >= 0 && y0 < H, yi1 = y0 + 1 >= 0 && y0 + 1 < H;
    z.in[0] = xi0 && yi0; z.in[1] = xi1 && yi0; z.in[2] = xi0 && yi1; z.in[3] = xi1 && yi1;
    const long long o = (long long)y0 * W + x0;
    const float v00 = z.in[0] ? 1.0f / (da + db * Dj[o]) : 0.f, v01 = z.in[1] ? 1.0f / (da + db * Dj[o + 1]) : 0.f;
    const float v10 = z.in[2] ? 1.0f / (da + db * Dj[o + W]) : 0.f, v11 = z.in[3] ? 1.0f / (da + db * Dj[o + W + 1]) : 0.f;
    z.w[0] = (1.f - wx) * (1.f - wy); z.w[1] = wx * (1.f - wy); z.w[2] = (1.f - wx) * wy; z.w[3] = wx * wy;
    z.val = v00 * z.w[0] + v01 * z.w[1] + v10 * z.w[2] + v11 * z.w[3];
    z.dix = (v01 - v00) * (1.f - wy) + (v11 - v10) * wy;
    z.diy = (v10 - v00) * (1.f - wx) + (v11 - v01) * wx;
    return z;
}
// un-clipped source coordinates of pixel (x, y) of frame f in camera f + fid (Project3D + grid_sample's un-normalisation)
__device__ __forceinline__ void project_unclipped(const Cam &c, float depth, int x, int y, int H, int W, float &ix, float &iy, Sample &s) {
    s = project_pixel(c, depth, x, y, H, W);
    const float zi = s.Z + 1e-7f;
    const float gx = (s.X / zi / (float)(W - 1) - 0.5f) * 2.0f, gy = (s.Y / zi / (float)(H - 1) - 0.5f) * 2.0f;
    ix = (gx + 1.0f) / 2.0f * (float)(W - 1);
    iy = (gy + 1.0f) / 2.0f * (float)(H - 1);
}

// part[((nb * N + f) * blocks + b) * 2] = {sum |Z - sampled| over the mask, count}
__global__ __launch_bounds__(256) void dreproj_fwd_kernel(const float *__restrict__ D, const Cam *__restrict__ cams, float *__restrict__ part, int N, int H, int W, float da,
                                                          float db) {
    __shared__ float red[2][4];
    const long long P = (long long)H * W;
    const int f = blockIdx.y, nb = blockIdx.z, j = nb ? f + 1 : f - 1;
    float sum = 0.f, cnt = 0.f;
    if (j >= 0 && j < N) {
        const Cam c = cams[2 * f + nb];
        const float *Dj = D + (long long)j * P;
        for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < P; p += (long long)gridDim.x * 256) {
            const int y = (int)(p / W), x = (int)(p - (long long)y * W);
            const float depth = 1.0f / (da + db * D[(long long)f * P + p]);
            float ix, iy;
            Sample s;
            project_unclipped(c, depth, x, y, H, W, ix, iy, s);
            const ZSample z = sample_depth_zeros(Dj, ix, iy, H, W, da, db);
            if (z.val > 1e-3f) {
                sum += fabsf(s.Z - z.val);
                cnt += 1.f;
            }
        }
    }
    const float v[2] = {sum, cnt};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const float t = wave_sum(v[k]);
        if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = t;
    }
    __syncthreads();
    if (threadIdx.x < 2) part[(((long long)nb * N + f) * gridDim.x + blockIdx.x) * 2 + threadIdx.x] = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
}

// SpatialTransformer (utils/layers.py:409-426): sample at pixel + flow; flow channel 0 is the row displacement
__device__ __forceinline__ void flow_coords(const float *__restrict__ fl, long long P, long long p, int x, int y, int H, int W, float &ix, float &iy) {
    const float ny = 2.0f * (((float)y + fl[p]) / (float)(H - 1) - 0.5f), nx = 2.0f * (((float)x + fl[P + p]) / (float)(W - 1) - 0.5f);
    ix = (nx + 1.0f) / 2.0f * (float)(W - 1);
    iy = (ny + 1.0f) / 2.0f * (float)(H - 1);
}
__global__ __launch_bounds__(256) void dflow_fwd_kernel(const float *__restrict__ D, const float *__restrict__ flow, int nb, float *__restrict__ part, int N, int H, int W,
                                                        float da, float db) {
    __shared__ float red[2][4];
    const long long P = (long long)H * W;
    const int f = blockIdx.y, j = nb ? f + 1 : f - 1;  // origin frame f, forward frame j
    float sum = 0.f, cnt = 0.f;
    if (j >= 0 && j < N) {
        const float *Df = D + (long long)f * P, *Dj = D + (long long)j * P, *fl = flow + (long long)f * 2 * P;
        for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < P; p += (long long)gridDim.x * 256) {
            const int y = (int)(p / W), x = (int)(p - (long long)y * W);
            float ix, iy;
            flow_coords(fl, P, p, x, y, H, W, ix, iy);
            const ZSample z = sample_depth_zeros(Df, ix, iy, H, W, da, db);
            if (z.val > 1e-3f) {
                sum += fabsf(z.val - 1.0f / (da + db * Dj[p]));
                cnt += 1.f;
            }
        }
    }
    const float v[2] = {sum, cnt};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const float t = wave_sum(v[k]);
        if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = t;
    }
    __syncthreads();
    if (threadIdx.x < 2) part[(((long long)nb * N + f) * gridDim.x + blockIdx.x) * 2 + threadIdx.x] = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
}
// tot[nb * 2 + {0, 1}] = {sum, count} over every frame and block (one workgroup, fixed order)
__global__ __launch_bounds__(256) void pair_sums_kernel(const float *__restrict__ part, long long n_per_nb, float *__restrict__ tot) {
    __shared__ float red[4][4];
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    for (int nb = 0; nb < 2; ++nb)
        for (long long i = threadIdx.x; i < n_per_nb; i += 256) {
            v[nb * 2] += part[((long long)nb * n_per_nb + i) * 2];
            v[nb * 2 + 1] += part[((long long)nb * n_per_nb + i) * 2 + 1];
        }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float t = wave_sum(v[k]);
        if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = t;
    }
    __syncthreads();
    if (threadIdx.x < 4) tot[threadIdx.x] = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
}

// the flow term's gradient: -sign / count into the forward frame's depth at p, +sign w_k / count into the origin frame's depth at the four corners
__global__ __launch_bounds__(256) void dflow_bwd_kernel(const float *__restrict__ D, const float *__restrict__ flow, int nb, const float *__restrict__ tot, float *__restrict__ gdep,
                                                        int N, int H, int W, float da, float db, float w_df) {
    const long long P = (long long)H * W;
    const int f = blockIdx.y, j = nb ? f + 1 : f - 1;
    if (j < 0 || j >= N) return;
    const float cnt = tot[nb * 2 + 1];
    if (!(cnt > 0.f)) return;
    const float c = w_df / cnt;
    const float *Df = D + (long long)f * P, *Dj = D + (long long)j * P, *fl = flow + (long long)f * 2 * P;
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < P; p += (long long)gridDim.x * 256) {
        const int y = (int)(p / W), x = (int)(p - (long long)y * W);
        float ix, iy;
        flow_coords(fl, P, p, x, y, H, W, ix, iy);
        const ZSample z = sample_depth_zeros(Df, ix, iy, H, W, da, db);
        if (!(z.val > 1e-3f)) continue;
        const float diff = z.val - 1.0f / (da + db * Dj[p]);
        const float sg = (diff > 0.f ? c : (diff < 0.f ? -c : 0.f));
        if (sg == 0.f) continue;
        atomicAdd(&gdep[(long long)j * P + p], -sg);
        const long long o = (long long)f * P + (long long)z.y0 * W + z.x0;
        if (z.in[0]) atomicAdd(&gdep[o], sg * z.w[0]);
        if (z.in[1]) atomicAdd(&gdep[o + 1], sg * z.w[1]);
        if (z.in[2]) atomicAdd(&gdep[o + W], sg * z.w[2]);
        if (z.in[3]) atomicAdd(&gdep[o + W + 1], sg * z.w[3]);
    }
}

// ---- geometry backward -----------------------------------------------------------------------------------------------------------------------------------
// Per pixel p of frame f and neighbour nb: dL/d(sampling coordinates) from the colour term (gx through the bilinear taps, border padding: the gradient
// passes strictly inside) and, with the depth-reprojection term on, from the sampled depth (zeros padding) plus dL/dZ directly; then through
// u = X / (Z + eps), v = Y / (Z + eps), (X, Y, Z) = P (depth ray, 1), ray = inv_K (x, y, 1):
//   gdepth_geom[f][p]                          (own pixel: plain store)
//   gdep[j][corners] -= sign w_k / count       (the sampled depth of the neighbouring frame: atomics)
//   pose partials: per (frame, block) 33 sums = dL/dP of both neighbours (2 x 12) and dL/d inv_K[:3, :3] (9)
template <bool POSE, bool DREPROJ>
__global__ __launch_bounds__(256) void geom_bwd_kernel(const float *__restrict__ D, const float *__restrict__ nb_prev, const float *__restrict__ nb_next,
                                                       const Cam *__restrict__ cams, const float *__restrict__ gx /* [2][N][3][P] */, const float *__restrict__ tot,
                                                       float *__restrict__ gdepth_geom, float *__restrict__ gdep, float *__restrict__ pose_part, int N, int H, int W,
                                                       float da, float db, float w_dr) {
    __shared__ float red[4][33];
    const long long P = (long long)H * W;
    const int f = blockIdx.y;
    float acc[33];
#pragma unroll
    for (int k = 0; k < 33; ++k) acc[k] = 0.f;
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < P; p += (long long)gridDim.x * 256) {
        const int y = (int)(p / W), x = (int)(p - (long long)y * W);
        const float depth = 1.0f / (da + db * D[(long long)f * P + p]);
        const Cam c0 = cams[2 * f];
        const float fx = (float)x, fy = (float)y;
        const float r0 = c0.iK[0] * fx + c0.iK[1] * fy + c0.iK[2], r1 = c0.iK[3] * fx + c0.iK[4] * fy + c0.iK[5], r2 = c0.iK[6] * fx + c0.iK[7] * fy + c0.iK[8];
        float gdepth = 0.f, gr0 = 0.f, gr1 = 0.f, gr2 = 0.f;  // d L / d depth, d L / d ray
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const Cam c = cams[2 * f + nb];
            const float *src = (nb ? nb_next : nb_prev) + (long long)f * 3 * P;
            const Sample s = project_pixel(c, depth, x, y, H, W);
            const float fx0 = floorf(s.ix), fy0 = floorf(s.iy);
            const int x0 = (int)fx0, y0 = (int)fy0;
            const float wx = s.ix - fx0, wy = s.iy - fy0;
            const bool xin = x0 + 1 < W, yin = y0 + 1 < H;
            float gu = 0.f, gv = 0.f, gz = 0.f;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                const float *sc = src + ch * P + (long long)y0 * W + x0;
                const float v00 = sc[0], v01 = xin ? sc[1] : 0.f, v10 = yin ? sc[W] : 0.f, v11 = (xin && yin) ? sc[W + 1] : 0.f;
                const float g = gx[(((long long)nb * N + f) * 3 + ch) * P + p];
                gu += g * ((v01 - v00) * (1.f - wy) + (v11 - v10) * wy);
                gv += g * ((v10 - v00) * (1.f - wx) + (v11 - v01) * wx);
            }
            gu *= s.mx;
            gv *= s.my;
            if (DREPROJ) {
                const int j = nb ? f + 1 : f - 1;
                const float cnt = tot[nb * 2 + 1];
                if (j >= 0 && j < N && cnt > 0.f) {
                    float ix, iy;
                    Sample s2;
                    project_unclipped(c, depth, x, y, H, W, ix, iy, s2);
                    const ZSample z = sample_depth_zeros(D + (long long)j * P, ix, iy, H, W, da, db);
                    if (z.val > 1e-3f) {
                        const float diff = s.Z - z.val;
                        const float sg = (diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f)) * (w_dr / cnt);
                        gz += sg;
                        gu -= sg * z.dix;
                        gv -= sg * z.diy;
                        const long long o = (long long)j * P + (long long)z.y0 * W + z.x0;
                        if (z.in[0]) atomicAdd(&gdep[o], -sg * z.w[0]);
                        if (z.in[1]) atomicAdd(&gdep[o + 1], -sg * z.w[1]);
                        if (z.in[2]) atomicAdd(&gdep[o + W], -sg * z.w[2]);
                        if (z.in[3]) atomicAdd(&gdep[o + W + 1], -sg * z.w[3]);
                    }
                }
            }
            const float zi = s.Z + 1e-7f;
            const float gX = gu / zi, gY = gv / zi, gZ = -(gu * s.X + gv * s.Y) / (zi * zi) + gz;
            // (X, Y, Z) = P[:, :3] (depth ray) + P[:, 3]
            const float gc0 = c.P[0] * gX + c.P[4] * gY + c.P[8] * gZ, gc1 = c.P[1] * gX + c.P[5] * gY + c.P[9] * gZ, gc2 = c.P[2] * gX + c.P[6] * gY + c.P[10] * gZ;
            gdepth += gc0 * r0 + gc1 * r1 + gc2 * r2;
            if (POSE) {
                gr0 += depth * gc0; gr1 += depth * gc1; gr2 += depth * gc2;
                const float cp[4] = {depth * r0, depth * r1, depth * r2, 1.0f};
                const float gq[3] = {gX, gY, gZ};
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[nb * 12 + r * 4 + k] += gq[r] * cp[k];
            }
        }
        gdepth_geom[(long long)f * P + p] = gdepth;
        if (POSE) {
            const float pix[3] = {fx, fy, 1.0f};
            const float gr[3] = {gr0, gr1, gr2};
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) acc[24 + a * 3 + b] += gr[a] * pix[b];
        }
    }
    if (POSE) {
#pragma unroll
        for (int k = 0; k < 33; ++k) {
            const float t = wave_sum(acc[k]);
            if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = t;
        }
        __syncthreads();
        if (threadIdx.x < 33) pose_part[((long long)f * gridDim.x + blockIdx.x) * 33 + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    }
}

// gD[f][p] = (gdepth_geom + gdep) * d depth / d D      (depth = 1 / (da + db D))
__global__ __launch_bounds__(256) void depth_grad_kernel(const float *__restrict__ D, const float *__restrict__ gdepth_geom, const float *__restrict__ gdep, float *__restrict__ gD,
                                                         long long n, float da, float db) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float depth = 1.0f / (da + db * D[i]);
        gD[i] = (gdepth_geom[i] + (gdep ? gdep[i] : 0.f)) * (-db * depth * depth);
    }
}

// smoothness gradient at the colour scale: gDc = w_sm (g_sm / den - S / den^2 / P_c)
__global__ __launch_bounds__(256) void smooth_grad_kernel(const float *__restrict__ gsm, const float *__restrict__ mean, const float *__restrict__ S, float *__restrict__ gDc,
                                                          long long Pc, float w_sm) {
    const int f = blockIdx.y;
    const float den = mean[f] + 1e-7f, all = S[f] / (den * den) / (float)Pc;
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < Pc; p += (long long)gridDim.x * 256) gDc[(long long)f * Pc + p] = w_sm * (gsm[(long long)f * Pc + p] / den - all);
}
__global__ __launch_bounds__(256) void add_kernel(const float *__restrict__ a, float *__restrict__ out, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) out[i] += a[i];
}

// per frame: dP[nb] (3x4) and d iK (3x3) from the block partials; dK += dP T^T (rows 0..2), dT += K[:3, :]^T dP, d inv_K[:3, :3] += d iK.  One wave per frame:
// lane k < 33 sums partial k over the blocks (in block order), then 12 + 16 + 16 + 9 lanes write one output element each.
__global__ __launch_bounds__(64) void pose_finish_kernel(const float *__restrict__ pose_part, int blocks, const float *__restrict__ K, const float *__restrict__ Tp, const float *__restrict__ Tn,
                                   float *__restrict__ gK, float *__restrict__ gInvK, float *__restrict__ gTp, float *__restrict__ gTn, int N) {
    __shared__ float s[33];
    const int f = blockIdx.x, t = threadIdx.x;
    if (t < 33) {
        float a = 0.f;
        for (int b = 0; b < blocks; ++b) a += pose_part[((long long)f * blocks + b) * 33 + t];
        s[t] = a;
    }
    __syncthreads();
    const float *k4 = K + (long long)f * 16;
    if (gK && t < 12) {  // dK[r][j] = sum over the neighbours (previous first) and c of dP[r][c] T[j][c]
        const int r = t >> 2, j = t & 3;
        float tot = gK[(long long)f * 16 + t];
        for (int nb = 0; nb < 2; ++nb) {
            const float *dP = s + nb * 12, *T = (nb ? Tn : Tp) + (long long)f * 16;
            float a = 0.f;
            for (int c = 0; c < 4; ++c) a += dP[r * 4 + c] * T[j * 4 + c];
            tot += a;
        }
        gK[(long long)f * 16 + t] = tot;
    }
    if (t >= 16 && t < 48) {  // dT[j][c] = sum_r K[r][j] dP[r][c]
        const int nb = (t - 16) >> 4, e = (t - 16) & 15, j = e >> 2, c = e & 3;
        float *gT = nb ? gTn : gTp;
        if (gT) {
            const float *dP = s + nb * 12;
            float a = 0.f;
            for (int r = 0; r < 3; ++r) a += k4[r * 4 + j] * dP[r * 4 + c];
            gT[(long long)f * 16 + e] += a;
        }
    }
    if (gInvK && t >= 48 && t < 57) {
        const int e = t - 48, a = e / 3, b = e - a * 3;
        gInvK[(long long)f * 16 + a * 4 + b] += s[24 + e];
    }
}

// One workgroup: this scale's terms -> losses[s * 7 + ...] and the running total; S[f] for the smoothness gradient
struct FinishArgs {
    const float *ssim_part[2];
    long long n_ssim;  // floats pairs per neighbour
    const float *cvt_part[2];
    int n_cvt;
    const float *sm_part;
    int sm_blocks, N;
    const float *drp_tot, *dfl_tot;  // {sum, cnt} x 2 neighbours, or nullptr
    float w_sm, inv_nx, inv_ny, w_dr, w_df;
};
__global__ __launch_bounds__(256) void finish_scale_kernel(FinishArgs a, float *__restrict__ S, float *__restrict__ losses, int s) {
    __shared__ float red[5][4];
    float rep = 0.f, tr = 0.f, cvt = 0.f, tx = 0.f, ty = 0.f;
    for (int nb = 0; nb < 2; ++nb) {
        for (long long i = threadIdx.x; i < a.n_ssim; i += 256) {
            rep += a.ssim_part[nb][i * 2];
            tr += a.ssim_part[nb][i * 2 + 1];
        }
        for (int i = threadIdx.x; i < a.n_cvt; i += 256) cvt += a.cvt_part[nb][i];
    }
    for (int i = threadIdx.x; i < a.N * a.sm_blocks; i += 256) {
        tx += a.sm_part[(long long)i * 3];
        ty += a.sm_part[(long long)i * 3 + 1];
    }
    for (int f = threadIdx.x; f < a.N; f += 256) {
        float t = 0.f;
        for (int b = 0; b < a.sm_blocks; ++b) t += a.sm_part[((long long)f * a.sm_blocks + b) * 3 + 2];
        S[f] = t;
    }
    const float v[5] = {rep, tr, cvt, tx, ty};
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const float t = wave_sum(v[k]);
        if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = t;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float t[5];
        for (int k = 0; k < 5; ++k) t[k] = (red[k][0] + red[k][1]) + (red[k][2] + red[k][3]);
        // every weight (1/2 per neighbour pair, the options' factors, 1 / 2^s) is already inside the partial sums EXCEPT the mean over the four scales,
        // so that losses[s * 7 + k] are the trainer's per-scale entries (trainer :960-966)
        const float l_rep = t[0], l_tr = t[1], l_cvt = t[2], l_sm = a.w_sm * (t[3] * a.inv_nx + t[4] * a.inv_ny);
        float l_dr = 0.f, l_df = 0.f;
        if (a.drp_tot)
            for (int nb = 0; nb < 2; ++nb) l_dr += a.w_dr * a.drp_tot[nb * 2] / a.drp_tot[nb * 2 + 1];  // (an empty mask gives NaN, as the reference's mean of nothing does)
        if (a.dfl_tot)
            for (int nb = 0; nb < 2; ++nb) l_df += a.w_df * a.dfl_tot[nb * 2] / a.dfl_tot[nb * 2 + 1];
        const float tot = l_rep + l_tr + l_cvt + l_sm + l_dr + l_df;
        float *o = losses + s * 7;
        o[0] = tot; o[1] = l_rep; o[2] = l_tr; o[3] = l_cvt; o[4] = l_sm; o[5] = l_dr; o[6] = l_df;
        losses[28] += 0.25f * tot;
    }
}

struct TLWs {  // carve-up of the caller's workspace (floats)
    size_t cams, msum, mpart, D, Dc, mean, S, sum_part, sm_part, ssim_part[2], cvt_part[2], pair_part, drp_tot, dfl_tot, gsm, gDc, gD, gdg, gdep, xw, gx, pose_part, total;
};
constexpr int SM_BLOCKS = 64, CVT_BLOCKS = 64, PAIR_BLOCKS = 64, POSE_BLOCKS = 128;
TLWs tl_layout(int N, int H, int W) {
    const size_t P = (size_t)H * W;
    const size_t tiles = (size_t)((H + TS - 1) / TS) * ((W + TS - 1) / TS);
    auto up = [](size_t n) { return (n + 3) & ~(size_t)3; };
    TLWs w;
    size_t o = 0;
    w.cams = o; o += up(2 * (size_t)N * (sizeof(Cam) / sizeof(float)));
    w.msum = o; o += up(2 * 4);
    w.mpart = o; o += up((size_t)N * MASK_BLOCKS * 3);
    w.D = o; o += up((size_t)N * P);
    w.Dc = o; o += up((size_t)N * P);
    w.mean = o; o += up(N);
    w.S = o; o += up(N);
    w.sum_part = o; o += up((size_t)N * SUM_PARTS);
    w.sm_part = o; o += up((size_t)N * SM_BLOCKS * 3);
    for (int nb = 0; nb < 2; ++nb) { w.ssim_part[nb] = o; o += up((size_t)N * 3 * tiles * 2); }
    for (int nb = 0; nb < 2; ++nb) { w.cvt_part[nb] = o; o += up((size_t)N * CVT_BLOCKS); }
    w.pair_part = o; o += up(2 * (size_t)N * PAIR_BLOCKS * 2);
    w.drp_tot = o; o += 4;
    w.dfl_tot = o; o += 4;
    w.gsm = o; o += up((size_t)N * P);
    w.gDc = o; o += up((size_t)N * P);
    w.gD = o; o += up((size_t)N * P);
    w.gdg = o; o += up((size_t)N * P);
    w.gdep = o; o += up((size_t)N * P);
    w.xw = o; o += up(2 * (size_t)N * 3 * P);
    w.gx = o; o += up(2 * (size_t)N * 3 * P);
    w.pose_part = o; o += up((size_t)N * POSE_BLOCKS * 33);
    w.total = o;
    return w;
}

}  // namespace

size_t trainer_loss_workspace(int N, int H, int W) {
    if (N <= 0 || H < 16 || W < 16) return 0;
    return tl_layout(N, H, W).total;
}

int trainer_loss(const TrainerLossIn &in, int N, int H, int W, const TrainerLossW &wt, float *losses, const TrainerLossGrads &g, float *ws, size_t ws_floats, hipStream_t st) {
    EDV_CHECK(losses && ws && in.K && in.invK && in.T[0] && in.T[1] && in.color_nb[0] && in.color_nb[1] && in.mask[0] && in.mask[1], "null argument");
    EDV_CHECK(N > 0 && N <= 65535 && H >= 16 && W >= 16, "N frames of at least 16 x 16 pixels (four loss scales)");
    EDV_CHECK(wt.min_depth > 0.f && wt.max_depth > wt.min_depth, "depth range");
    const TLWs L = tl_layout(N, H, W);
    EDV_CHECK(ws_floats >= L.total && (uintptr_t)ws % 16 == 0, "workspace too small (trainer_loss_workspace)");
    const long long P = (long long)H * W;
    Cam *cams = reinterpret_cast<Cam *>(ws + L.cams);
    float *msum = ws + L.msum, *mpart = ws + L.mpart, *Dbuf = ws + L.D, *Dcbuf = ws + L.Dc, *mean = ws + L.mean, *S = ws + L.S, *sum_part = ws + L.sum_part,
          *sm_part = ws + L.sm_part, *pair_part = ws + L.pair_part, *drp_tot = ws + L.drp_tot, *dfl_tot = ws + L.dfl_tot, *gsm = ws + L.gsm, *gDc = ws + L.gDc, *gD = ws + L.gD,
          *gdg = ws + L.gdg, *gdep = ws + L.gdep, *xw = ws + L.xw, *gx = ws + L.gx, *pose_part = ws + L.pose_part;
    const float da = (float)(1.0 / (double)wt.max_depth), db = (float)(1.0 / (double)wt.min_depth - 1.0 / (double)wt.max_depth);
    const int tiles_x = (W + TS - 1) / TS, tiles = tiles_x * ((H + TS - 1) / TS);
    const int pix_blocks = (int)((P + 255) / 256 < 1024 ? (P + 255) / 256 : 1024);
    const float tw = wt.tune_temporal ? 1.0f : 0.0f;
    const bool do_dr = tw * wt.depth_reproj != 0.f, do_df = tw * wt.depth_flow != 0.f;
    const bool pose = g.K || g.invK || g.T[0] || g.T[1];

    EDV_HIP(hipMemsetAsync(losses, 0, 29 * sizeof(float), st));
    if (g.K) EDV_HIP(hipMemsetAsync(g.K, 0, (size_t)N * 16 * sizeof(float), st));
    if (g.invK) EDV_HIP(hipMemsetAsync(g.invK, 0, (size_t)N * 16 * sizeof(float), st));
    for (int nb = 0; nb < 2; ++nb)
        if (g.T[nb]) EDV_HIP(hipMemsetAsync(g.T[nb], 0, (size_t)N * 16 * sizeof(float), st));
    EDV_LAUNCH(cam_kernel, dim3((2 * N + 63) / 64), dim3(64), 0, st, in.K, in.invK, in.T[0], in.T[1], cams, N);
    EDV_LAUNCH_OK();
    for (int nb = 0; nb < 2; ++nb) {
        EDV_LAUNCH(mask_sums_kernel, dim3(MASK_BLOCKS, N), dim3(256), 0, st, in.mask[nb], mpart, H, W);
        EDV_LAUNCH_OK();
        EDV_LAUNCH(mask_sums_finish_kernel, dim3(1), dim3(64), 0, st, mpart, msum + nb * 4, N * MASK_BLOCKS);
        EDV_LAUNCH_OK();
    }
    for (int s = 0; s < 4; ++s) {
        EDV_CHECK(in.disp[s] && g.disp[s] && in.disp_h[s] > 0 && in.disp_w[s] > 0 && in.color[s], "bad disparity map / colour scale");
        const int Hc = H >> s, Wc = W >> s;  // the dataset's colour pyramid: height // 2^s (datasets/scared_video_dataset.py:186-188)
        EDV_CHECK(Hc >= 2 && Wc >= 2, "colour scale too small");
        const long long Pc = (long long)Hc * Wc;
        for (int nb = 0; nb < 2; ++nb)
            EDV_CHECK(in.refined[s][nb] && in.registration[s][nb] && in.registration[0][nb] && in.transform[s][nb] && (!do_df || in.position[s][nb]), "null side-network tensor");
        // ---- D (frame size) and D_c (colour scale) ----
        const bool same = in.disp_h[s] == H && in.disp_w[s] == W, same_c = in.disp_h[s] == Hc && in.disp_w[s] == Wc;
        const float *D = in.disp[s], *Dc = in.disp[s];
        if (!same) {
            EDV_TRY(bilinear(in.disp[s], Dbuf, N, in.disp_h[s], in.disp_w[s], 1, H, W, ACT_NONE, st));
            D = Dbuf;
        }
        if (!same_c) {
            EDV_TRY(bilinear(in.disp[s], Dcbuf, N, in.disp_h[s], in.disp_w[s], 1, Hc, Wc, ACT_NONE, st));
            Dc = Dcbuf;
        }
        // ---- smoothness at the colour scale ----
        const float w_sm = wt.disparity_smoothness / (float)(1 << s);
        const float inv_nx = (float)(1.0 / ((double)N * Hc * (Wc - 1))), inv_ny = (float)(1.0 / ((double)N * (Hc - 1) * Wc));
        EDV_LAUNCH(frame_sum_kernel, dim3(SUM_PARTS, N), dim3(256), 0, st, Dc, sum_part, Pc);
        EDV_LAUNCH_OK();
        EDV_LAUNCH(frame_sum_finish_kernel, dim3(N), dim3(64), 0, st, sum_part, mean, (float)(1.0 / (double)Pc));
        EDV_LAUNCH_OK();
        EDV_LAUNCH(smooth_kernel, dim3(SM_BLOCKS, N), dim3(256), 0, st, Dc, in.color[s], mean, gsm, sm_part, Hc, Wc, inv_nx, inv_ny);
        EDV_LAUNCH_OK();
        // ---- the two neighbours: warp, weighted SSIM / L1 / transform term, appearance-flow smoothness ----
        for (int nb = 0; nb < 2; ++nb) {
            float *xw_nb = xw + (size_t)nb * N * 3 * P, *gx_nb = gx + (size_t)nb * N * 3 * P;
            EDV_LAUNCH(warp_nb_kernel, dim3(pix_blocks, N), dim3(256), 0, st, D, in.color_nb[nb], cams, nb, xw_nb, H, W, da, db);
            EDV_LAUNCH_OK();
            // per-scale weights: rep / 2, tc tr / 2 (trainer :953-954); the 1/4 of the mean over scales multiplies every GRADIENT below (0.25 * ...)
            EDV_LAUNCH(ssimw_kernel, dim3(tiles, 3, N), dim3(256), 0, st, xw_nb, in.refined[s][nb], in.registration[0][nb], in.mask[nb], msum + nb * 4, gx_nb,
                               g.refined[s][nb], ws + L.ssim_part[nb], H, W, tiles_x, 0.5f, 0.5f * wt.transform_constraint);
            EDV_LAUNCH_OK();
            EDV_LAUNCH(cvt_kernel, dim3(CVT_BLOCKS, N), dim3(256), 0, st, in.transform[s][nb], in.color[0], in.registration[s][nb], in.mask[nb], msum + nb * 4,
                               g.transform[s][nb], ws + L.cvt_part[nb], H, W, 0.5f * wt.transform_smoothness);
            EDV_LAUNCH_OK();
        }
        // ---- depth-consistency sums (their means need the counts before the gradients) ----
        const float w_dr = tw * wt.depth_reproj * 0.5f, w_df = tw * wt.depth_flow * 0.5f;
        if (do_dr) {
            EDV_LAUNCH(dreproj_fwd_kernel, dim3(PAIR_BLOCKS, N, 2), dim3(256), 0, st, D, cams, pair_part, N, H, W, da, db);
            EDV_LAUNCH_OK();
            EDV_LAUNCH(pair_sums_kernel, dim3(1), dim3(256), 0, st, pair_part, (long long)N * PAIR_BLOCKS, drp_tot);
            EDV_LAUNCH_OK();
        }
        if (do_df) {
            for (int nb = 0; nb < 2; ++nb) {
                EDV_LAUNCH(dflow_fwd_kernel, dim3(PAIR_BLOCKS, N), dim3(256), 0, st, D, in.position[s][nb], nb, pair_part, N, H, W, da, db);
                EDV_LAUNCH_OK();
            }
            EDV_LAUNCH(pair_sums_kernel, dim3(1), dim3(256), 0, st, pair_part, (long long)N * PAIR_BLOCKS, dfl_tot);
            EDV_LAUNCH_OK();
        }
        FinishArgs fa{{ws + L.ssim_part[0], ws + L.ssim_part[1]}, (long long)N * 3 * tiles, {ws + L.cvt_part[0], ws + L.cvt_part[1]}, N * CVT_BLOCKS, sm_part, SM_BLOCKS, N,
                      do_dr ? drp_tot : nullptr, do_df ? dfl_tot : nullptr, w_sm, inv_nx, inv_ny, w_dr, w_df};
        EDV_LAUNCH(finish_scale_kernel, dim3(1), dim3(256), 0, st, fa, S, losses, s);
        EDV_LAUNCH_OK();
        // ---- gradients.  The kernels above produced d(loss_s)/d(.) for refined, transform_high and the warped images; total = mean over scales -> x 0.25 ----
        if (do_dr || do_df) EDV_HIP(hipMemsetAsync(gdep, 0, (size_t)N * P * sizeof(float), st));
        const int gb = pose ? POSE_BLOCKS : pix_blocks;
        if (pose && do_dr)
            EDV_LAUNCH((geom_bwd_kernel<true, true>), dim3(gb, N), dim3(256), 0, st, D, in.color_nb[0], in.color_nb[1], cams, gx, drp_tot, gdg, gdep, pose_part, N, H, W, da, db, w_dr);
        else if (pose)
            EDV_LAUNCH((geom_bwd_kernel<true, false>), dim3(gb, N), dim3(256), 0, st, D, in.color_nb[0], in.color_nb[1], cams, gx, drp_tot, gdg, gdep, pose_part, N, H, W, da, db, w_dr);
        else if (do_dr)
            EDV_LAUNCH((geom_bwd_kernel<false, true>), dim3(gb, N), dim3(256), 0, st, D, in.color_nb[0], in.color_nb[1], cams, gx, drp_tot, gdg, gdep, pose_part, N, H, W, da, db, w_dr);
        else
            EDV_LAUNCH((geom_bwd_kernel<false, false>), dim3(gb, N), dim3(256), 0, st, D, in.color_nb[0], in.color_nb[1], cams, gx, drp_tot, gdg, gdep, pose_part, N, H, W, da, db, w_dr);
        EDV_LAUNCH_OK();
        if (do_df)
            for (int nb = 0; nb < 2; ++nb) {
                EDV_LAUNCH(dflow_bwd_kernel, dim3(pix_blocks, N), dim3(256), 0, st, D, in.position[s][nb], nb, dfl_tot, gdep, N, H, W, da, db, w_df);
                EDV_LAUNCH_OK();
            }
        float *gdst = same ? g.disp[s] : gD;
        EDV_LAUNCH(depth_grad_kernel, dim3(pix_blocks), dim3(256), 0, st, D, gdg, (do_dr || do_df) ? gdep : nullptr, gdst, (long long)N * P, da, db);
        EDV_LAUNCH_OK();
        if (!same) EDV_TRY(bilinear_bwd(gD, g.disp[s], N, in.disp_h[s], in.disp_w[s], 1, H, W, false, st));
        EDV_LAUNCH(smooth_grad_kernel, dim3(SM_BLOCKS, N), dim3(256), 0, st, gsm, mean, S, gDc, Pc, w_sm);
        EDV_LAUNCH_OK();
        if (same_c) {
            EDV_LAUNCH(add_kernel, dim3(pix_blocks), dim3(256), 0, st, gDc, g.disp[s], (long long)N * Pc);
            EDV_LAUNCH_OK();
        } else {
            EDV_TRY(bilinear_bwd(gDc, g.disp[s], N, in.disp_h[s], in.disp_w[s], 1, Hc, Wc, true, st));
        }
        if (pose) {
            EDV_LAUNCH(pose_finish_kernel, dim3(N), dim3(64), 0, st, pose_part, gb, in.K, in.T[0], in.T[1], g.K, g.invK, g.T[0], g.T[1], N);
            EDV_LAUNCH_OK();
        }
    }
    // the mean over the four scales (trainer :968): every gradient written above is d(loss_s); scale them by 1/4
    ScaleList sl;
    int ns = 0;
    long long nmax = 0;
    auto scale = [&](float *p, long long n) {
        if (!p || n <= 0) return;
        sl.p[ns] = p;
        sl.n[ns++] = n;
        nmax = n > nmax ? n : nmax;
    };
    for (int s = 0; s < 4; ++s) {
        scale(g.disp[s], (long long)N * in.disp_h[s] * in.disp_w[s]);
        for (int nb = 0; nb < 2; ++nb) {
            scale(g.refined[s][nb], (long long)N * 3 * P);
            scale(g.transform[s][nb], (long long)N * 3 * P);
        }
    }
    scale(g.K, (long long)N * 16);
    scale(g.invK, (long long)N * 16);
    scale(g.T[0], (long long)N * 16);
    scale(g.T[1], (long long)N * 16);
    static_assert(4 * 5 + 4 <= SCALE_LIST, "scale list");
    for (int i = ns; i < SCALE_LIST; ++i) { sl.p[i] = nullptr; sl.n[i] = 0; }
    const long long b4 = (nmax / 4 + 255) / 256;
    EDV_LAUNCH(scale_list_kernel, dim3((unsigned)(b4 < 1 ? 1 : b4 > 256 ? 256 : b4), ns), dim3(256), 0, st, sl, 0.25f);
    EDV_LAUNCH_OK();
    return 0;
}

}  // namespace edv
