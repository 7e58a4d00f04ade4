// Resampling and small element-wise kernels (all HBM-bound, channels-last):
//   patchify      endodav.py:153-155 (bilinear resize to image_shape, align_corners=True; ImageNet
//                 normalise) fused with the im2col of the 14x14/14 patch conv (patch_embed.py:75)
//   bilinear      every F.interpolate(mode="bilinear", align_corners=True) of the head
//                 (util/blocks.py:157, dpt_pyramid.py:90,95-97, endodav/layers.py:211)
//   dot_channels  the final 1x1 convs to one channel (dpt.py:121, layers.py:214) + ReLU / sigmoid
//   bicubic_pos   interpolate_pos_encoding (vision_transformer.py:186-217)
// The index arithmetic follows ATen's upsample kernels: ratio = (in-1)/(out-1) in float,
// src = ratio*dst, i0 = int(src), lambda = src - i0, i1 = i0 + (i0 < in-1).
#include "ops.hpp"

// ATen computes the source coordinate as a ROUNDED float product and then subtracts the integer part;
// hipcc's default -ffp-contract=fast fuses the two into one FMA (seen as v_pk_fma_f32 in the ISA), which
// moves the interpolation weight by up to an ulp of the coordinate (3e-5 at 518 px).  These kernels are
// HBM-bound, so contraction buys nothing here: switch it off for the whole file.
#pragma clang fp contract(off)

namespace edv {
namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

__device__ __forceinline__ void lin_coord(int dst, int in, int out, float ratio, int &i0, int &i1, float &l1) {
    if (in == out) {
        i0 = i1 = dst;
        l1 = 0.f;
        return;
    }
    const float src = __fmul_rn(ratio, (float)dst);  // rounded product, as ATen computes it (no FMA into src - i0)
    i0 = (int)src;
    i1 = i0 + (i0 < in - 1 ? 1 : 0);
    l1 = fminf(fmaxf(src - (float)i0, 0.f), 1.f);
}
inline float lin_ratio(int in, int out) { return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f; }

__global__ __launch_bounds__(256) void patchify_kernel(const float *__restrict__ x, float *__restrict__ cols, int F, int H, int W, int ih, int iw,
                                                        float rh, float rw, int ld) {
    const int ph = ih / 14, pw = iw / 14;
    const long long total = (long long)F * ph * pw * ld;
    const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int k = (int)(i % ld);
        long long r = i / ld;
        if (k >= 588) {  // row padding (the engine pads K to a multiple of 32 for the LDS-DMA GEMM)
            cols[i] = 0.f;
            continue;
        }
        const int px = (int)(r % pw);
        r /= pw;
        const int py = (int)(r % ph);
        const int f = (int)(r / ph);
        const int c = k / 196, kk = k - c * 196, ky = kk / 14, kx = kk - ky * 14;
        const int y = py * 14 + ky, xx = px * 14 + kx;
        int y0, y1, x0, x1;
        float ly, lx;
        lin_coord(y, H, ih, rh, y0, y1, ly);
        lin_coord(xx, W, iw, rw, x0, x1, lx);
        const float *pl = x + ((long long)f * 3 + c) * H * W;
        const float v00 = pl[(long long)y0 * W + x0], v01 = pl[(long long)y0 * W + x1];
        const float v10 = pl[(long long)y1 * W + x0], v11 = pl[(long long)y1 * W + x1];
        const float v = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
        cols[i] = (v - mean[c]) / stdv[c];
    }
}

// thread per (output pixel, float4 of channels)
__global__ __launch_bounds__(256) void bilinear_c4_kernel(const float *__restrict__ x, float *__restrict__ y, int F, int H, int W, int C4, int OH,
                                                           int OW, float rh, float rw, const float *__restrict__ add) {
    const long long total = (long long)F * OH * OW * C4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C4);
        long long r = i / C4;
        const int ox = (int)(r % OW);
        r /= OW;
        const int oy = (int)(r % OH);
        const long long f = r / OH;
        int y0, y1, x0, x1;
        float ly, lx;
        lin_coord(oy, H, OH, rh, y0, y1, ly);
        lin_coord(ox, W, OW, rw, x0, x1, lx);
        const f32x4 *pl = reinterpret_cast<const f32x4 *>(x) + f * H * W * C4 + c;
        const f32x4 v00 = pl[((long long)y0 * W + x0) * C4], v01 = pl[((long long)y0 * W + x1) * C4];
        const f32x4 v10 = pl[((long long)y1 * W + x0) * C4], v11 = pl[((long long)y1 * W + x1) * C4];
        f32x4 o = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
        if (add) o += reinterpret_cast<const f32x4 *>(add)[i];  // the skip branch of the next fusion block, shaped like y
        reinterpret_cast<f32x4 *>(y)[i] = o;
    }
}

__global__ __launch_bounds__(256) void bilinear_c1_kernel(const float *__restrict__ x, float *__restrict__ y, int F, int H, int W, int OH, int OW,
                                                           float rh, float rw) {
    const long long total = (long long)F * OH * OW;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int ox = (int)(i % OW);
        long long r = i / OW;
        const int oy = (int)(r % OH);
        const long long f = r / OH;
        int y0, y1, x0, x1;
        float ly, lx;
        lin_coord(oy, H, OH, rh, y0, y1, ly);
        lin_coord(ox, W, OW, rw, x0, x1, lx);
        const float *pl = x + f * H * W;
        const float v00 = pl[(long long)y0 * W + x0], v01 = pl[(long long)y0 * W + x1];
        const float v10 = pl[(long long)y1 * W + x0], v11 = pl[(long long)y1 * W + x1];
        y[i] = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
    }
}

__device__ __forceinline__ float final_act(float v, int act) {
    if (act == ACT_RELU) return fmaxf(v, 0.f);
    if (act == ACT_SIGMOID) return 1.f / (1.f + expf(-v));
    if (act == ACT_SIGMOID_NEG) return 1.f / (1.f + expf(v));
    return v;
}

// LPP = C/4 lanes cooperate on one pixel: coalesced float4 loads + shuffle reduction
template <int LPP>
__global__ __launch_bounds__(256) void dot_channels_kernel(const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ b,
                                                            float *__restrict__ y, long long M, int act) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long pix = gid / LPP;
    const int sub = (int)(gid % LPP);
    const long long pc = pix < M ? pix : M - 1;  // keep every lane in the shuffles
    const f32x4 v = *reinterpret_cast<const f32x4 *>(x + (pc * LPP + sub) * 4);
    const f32x4 ww = *reinterpret_cast<const f32x4 *>(w + sub * 4);
    float s = (v.x * ww.x + v.y * ww.y) + (v.z * ww.z + v.w * ww.w);
#pragma unroll
    for (int o = LPP / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (sub == 0 && pix < M) y[pix] = final_act(s + b[0], act);
}

__global__ void cls_rows_kernel(const float *__restrict__ cls, const float *__restrict__ pos, float *__restrict__ tokens, int F, int ntok, int D) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)F * D) return;
    const int d = (int)(i % D);
    const long long f = i / D;
    tokens[f * ntok * D + d] = cls[d] + pos[d];
}

__global__ void sigmoid_kernel(float *__restrict__ x, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) x[i] = 1.f / (1.f + expf(-x[i]));
}

// ATen upsample_bicubic2d, align_corners=False, A = -0.75, border-clamped taps
__device__ __forceinline__ float cc1(float x) { return ((-0.75f + 2.f) * x - (-0.75f + 3.f)) * x * x + 1.f; }
__device__ __forceinline__ float cc2(float x) { return ((-0.75f * x - 5.f * -0.75f) * x + 8.f * -0.75f) * x - 4.f * -0.75f; }
__global__ void bicubic_pos_kernel(const float *__restrict__ grid, float *__restrict__ out, int S, int D, int oh, int ow, float rh, float rw) {
    const long long total = (long long)oh * ow * D;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int d = (int)(i % D);
    const int ox = (int)((i / D) % ow), oy = (int)(i / ((long long)D * ow));
    const float sy = __fsub_rn(__fmul_rn(rh, (float)oy + 0.5f), 0.5f), sx = __fsub_rn(__fmul_rn(rw, (float)ox + 0.5f), 0.5f);
    const int iy = (int)floorf(sy), ix = (int)floorf(sx);
    const float ty = sy - (float)iy, tx = sx - (float)ix;
    const float wy[4] = {cc2(ty + 1.f), cc1(ty), cc1(1.f - ty), cc2(2.f - ty)};
    const float wx[4] = {cc2(tx + 1.f), cc1(tx), cc1(1.f - tx), cc2(2.f - tx)};
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int yy = min(max(iy - 1 + a, 0), S - 1);
        float rowv = 0.f;
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) {
            const int xx = min(max(ix - 1 + bb, 0), S - 1);
            rowv += wx[bb] * grid[((long long)yy * S + xx) * D + d];
        }
        acc += wy[a] * rowv;
    }
    out[i] = acc;
}

// planar image resize with the same cubic (cv2.INTER_CUBIC / ATen bicubic, align_corners=False)
__global__ void resize_bicubic_kernel(const float *__restrict__ x, float *__restrict__ y, int NP, int H, int W, int OH, int OW, float rh, float rw) {
    const long long total = (long long)NP * OH * OW;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int ox = (int)(i % OW);
        const int oy = (int)((i / OW) % OH);
        const long long pl = i / ((long long)OW * OH);
        const float sy = __fsub_rn(__fmul_rn(rh, (float)oy + 0.5f), 0.5f), sx = __fsub_rn(__fmul_rn(rw, (float)ox + 0.5f), 0.5f);
        const int iy = (int)floorf(sy), ix = (int)floorf(sx);
        const float ty = sy - (float)iy, tx = sx - (float)ix;
        const float wy[4] = {cc2(ty + 1.f), cc1(ty), cc1(1.f - ty), cc2(2.f - ty)};
        const float wx[4] = {cc2(tx + 1.f), cc1(tx), cc1(1.f - tx), cc2(2.f - tx)};
        const float *src = x + pl * H * W;
        float acc = 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int yy = min(max(iy - 1 + a, 0), H - 1);
            float rowv = 0.f;
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) rowv += wx[bb] * src[(long long)yy * W + min(max(ix - 1 + bb, 0), W - 1)];
            acc += wy[a] * rowv;
        }
        y[i] = acc;
    }
}

inline int grid_for(long long total, int cap = 8192) {
    long long b = (total + 255) / 256;
    return (int)(b < 1 ? 1 : (b < cap ? b : cap));
}

}  // namespace

int patchify(const float *x, float *cols, int F, int H, int W, int ih, int iw, hipStream_t st, int ld) {
    EDV_CHECK(x && cols, "null operand");
    EDV_CHECK(F > 0 && H > 0 && W > 0 && ld >= 588, "empty problem");
    EDV_CHECK(ih % 14 == 0 && iw % 14 == 0 && ih > 0 && iw > 0, "image_shape must be a multiple of 14");
    const long long total = (long long)F * (ih / 14) * (iw / 14) * ld;
    EDV_LAUNCH(patchify_kernel, dim3(grid_for(total, 16384)), dim3(256), 0, st, x, cols, F, H, W, ih, iw, lin_ratio(H, ih), lin_ratio(W, iw), ld);
    EDV_LAUNCH_OK();
    return 0;
}

int bilinear(const float *x, float *y, int F, int H, int W, int C, int OH, int OW, int act, hipStream_t st, const float *add) {
    EDV_CHECK(x && y, "null operand");
    EDV_CHECK(F > 0 && H > 0 && W > 0 && OH > 0 && OW > 0, "empty problem");
    EDV_CHECK(act == ACT_NONE, "bilinear: fused activation not implemented");
    EDV_CHECK(C == 1 || C % 4 == 0, "C must be 1 or a multiple of 4");
    EDV_CHECK(!add || C % 4 == 0, "bilinear: the addend needs C % 4 == 0");
    const float rh = lin_ratio(H, OH), rw = lin_ratio(W, OW);
    if (C == 1) {
        EDV_LAUNCH(bilinear_c1_kernel, dim3(grid_for((long long)F * OH * OW)), dim3(256), 0, st, x, y, F, H, W, OH, OW, rh, rw);
    } else {
        EDV_LAUNCH(bilinear_c4_kernel, dim3(grid_for((long long)F * OH * OW * (C / 4), 16384)), dim3(256), 0, st, x, y, F, H, W, C / 4, OH, OW,
                           rh, rw, add);
    }
    EDV_LAUNCH_OK();
    return 0;
}

int dot_channels(const float *x, const float *w, const float *b, float *y, long long M, int C, int act, hipStream_t st) {
    EDV_CHECK(x && w && b && y, "null operand");
    EDV_CHECK(M > 0, "empty problem");
    const int lpp = C / 4;
    EDV_CHECK(C % 4 == 0 && lpp >= 1 && lpp <= 64 && (lpp & (lpp - 1)) == 0, "C/4 must be a power of two <= 64");
    const long long blocks = (M * lpp + 255) / 256;
    EDV_CHECK(blocks < (1ll << 31), "grid");
    dim3 grid((unsigned)blocks), block(256);
    switch (lpp) {
        case 1: EDV_LAUNCH(dot_channels_kernel<1>, grid, block, 0, st, x, w, b, y, M, act); break;
        case 2: EDV_LAUNCH(dot_channels_kernel<2>, grid, block, 0, st, x, w, b, y, M, act); break;
        case 4: EDV_LAUNCH(dot_channels_kernel<4>, grid, block, 0, st, x, w, b, y, M, act); break;
        case 8: EDV_LAUNCH(dot_channels_kernel<8>, grid, block, 0, st, x, w, b, y, M, act); break;
        case 16: EDV_LAUNCH(dot_channels_kernel<16>, grid, block, 0, st, x, w, b, y, M, act); break;
        case 32: EDV_LAUNCH(dot_channels_kernel<32>, grid, block, 0, st, x, w, b, y, M, act); break;
        default: EDV_LAUNCH(dot_channels_kernel<64>, grid, block, 0, st, x, w, b, y, M, act); break;
    }
    EDV_LAUNCH_OK();
    return 0;
}

int cls_rows(const float *cls, const float *pos, float *tokens, int F, int ntok, int D, hipStream_t st) {
    EDV_CHECK(cls && pos && tokens, "null operand");
    EDV_LAUNCH(cls_rows_kernel, dim3(grid_for((long long)F * D, 1 << 30)), dim3(256), 0, st, cls, pos, tokens, F, ntok, D);
    EDV_LAUNCH_OK();
    return 0;
}

int sigmoid_inplace(float *x, long long n, hipStream_t st) {
    EDV_CHECK(x && n > 0, "bad operand");
    EDV_LAUNCH(sigmoid_kernel, dim3(grid_for(n)), dim3(256), 0, st, x, n);
    EDV_LAUNCH_OK();
    return 0;
}

int resize_bicubic(const float *x, float *y, int NP, int H, int W, int OH, int OW, hipStream_t st) {
    EDV_CHECK(x && y && NP > 0 && H > 0 && W > 0 && OH > 0 && OW > 0, "bad operand");
    EDV_LAUNCH(resize_bicubic_kernel, dim3(grid_for((long long)NP * OH * OW, 16384)), dim3(256), 0, st, x, y, NP, H, W, OH, OW,
                       (float)H / (float)OH, (float)W / (float)OW);
    EDV_LAUNCH_OK();
    return 0;
}

int bicubic_pos(const float *grid, float *out, int S, int D, int oh, int ow, float scale_h, float scale_w, hipStream_t st) {
    EDV_CHECK(grid && out && S > 0 && D > 0 && oh > 0 && ow > 0, "bad operand");
    EDV_LAUNCH(bicubic_pos_kernel, dim3(grid_for((long long)oh * ow * D, 1 << 30)), dim3(256), 0, st, grid, out, S, D, oh, ow, scale_h, scale_w);
    EDV_LAUNCH_OK();
    return 0;
}

}  // namespace edv
