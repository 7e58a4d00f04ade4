// extern "C" per-kernel entry points declared in include/endodav_hip.h (unit tests, micro-benchmarks).
#include "../../include/endodav_hip.h"
#include "ops.hpp"

using namespace edv;

extern "C" {

int edv_layernorm(const float *x_dev, const float *w_dev, const float *b_dev, float *y_dev, int64_t rows, int32_t dim, float eps,
                  const float *pe_dev, int32_t rows_per_frame, int32_t T, void *stream) {
    return layernorm(x_dev, identity_map(), w_dev, b_dev, y_dev, identity_map(), rows, dim, eps, pe_dev, rows_per_frame, T, (hipStream_t)stream);
}

size_t edv_gemm_workspace(void) { return gemm_workspace() * sizeof(float); }
int edv_gemm(const float *A_dev, const float *W_dev, float *C_dev, int64_t M, int32_t N, int32_t K, const float *bias_dev, int32_t act,
             const float *gamma_dev, const float *R_dev, float *workspace_dev, size_t workspace_bytes, void *stream) {
    GemmDesc g;
    g.A = A_dev; g.lda = K; g.W = W_dev; g.ldw = K; g.C = C_dev; g.ldc = N; g.M = M; g.N = N; g.K = K;
    g.bias = bias_dev; g.act = act; g.gamma = gamma_dev; g.R1 = R_dev; g.ldr1 = N;
    g.ws = workspace_dev; g.ws_floats = workspace_bytes / sizeof(float);
    EDV_CHECK(act >= ACT_NONE && act <= ACT_RELU, "act must be 0, 1 or 2");
    return gemm(g, (hipStream_t)stream);
}

size_t edv_gemm_x6_planes_bytes(int32_t N, int32_t K) { return N > 0 && K > 0 ? gemm_x6_planes_bytes(N, K) : 0; }

int edv_gemm_x6_split(const float *W_dev, void *planes_dev, int32_t N, int32_t K, void *stream) {
    return gemm_x6_split(W_dev, planes_dev, N, K, (hipStream_t)stream);
}

int edv_gemm_x6(const float *A_dev, const void *planes_dev, float *C_dev, int64_t M, int32_t N, int32_t K, const float *bias_dev, int32_t act,
                const float *gamma_dev, const float *R_dev, float *workspace_dev, size_t workspace_bytes, void *stream) {
    GemmDesc g;
    g.A = A_dev; g.lda = K; g.W = reinterpret_cast<const float *>(planes_dev); g.ldw = K; g.C = C_dev; g.ldc = N; g.M = M; g.N = N; g.K = K;
    g.bias = bias_dev; g.act = act; g.gamma = gamma_dev; g.R1 = R_dev; g.ldr1 = N;
    g.ws = workspace_dev; g.ws_floats = workspace_bytes / sizeof(float);
    g.Wx6 = planes_dev;
    EDV_CHECK(act >= ACT_NONE && act <= ACT_RELU, "act must be 0, 1 or 2");
    EDV_CHECK(M > 0 && N >= 64 && K > 0 && K % 16 == 0, "edv_gemm_x6: K % 16 == 0, N >= 64");
    return gemm_x6(g, (hipStream_t)stream);
}

int edv_pack_geglu(const float *w_dev, const float *b_dev, float *wi_dev, float *bi_dev, int32_t N, int32_t K, void *stream) {
    return pack_geglu(w_dev, b_dev, wi_dev, bi_dev, N, K, (hipStream_t)stream);
}

int edv_gemm_geglu(const float *A_dev, const float *Wi_dev, const float *bi_dev, float *C_dev, int64_t M, int32_t N, int32_t K, void *stream) {
    GemmDesc g;
    g.A = A_dev; g.lda = K; g.W = Wi_dev; g.ldw = K; g.C = C_dev; g.ldc = N / 2; g.M = M; g.N = N; g.K = K;
    g.bias = bi_dev; g.geglu = 1;
    EDV_CHECK(A_dev && Wi_dev && bi_dev && C_dev && M > 0, "null operand");
    EDV_CHECK(gemm_geglu_supported(g), "edv_gemm_geglu: K % 32 == 0, N % 64 == 0, output below 4 GB");
    return gemm(g, (hipStream_t)stream);
}

static int conv3x3_impl(const float *x_dev, const float *wpacked_dev, const float *bias_dev, float *y_dev, int32_t F, int32_t H, int32_t W, int32_t Cin,
                        int32_t Cout, int32_t stride, int32_t pre_relu, int32_t post_relu, const float *R1_dev, const float *R2_dev, float *ws,
                        size_t ws_bytes, void *stream) {
    EDV_CHECK(F > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "empty problem");
    EDV_CHECK(stride == 1 || stride == 2, "stride must be 1 or 2");
    GemmDesc g;
    const int OH = (H - 1) / stride + 1, OW = (W - 1) / stride + 1;
    g.A = x_dev; g.W = wpacked_dev; g.ldw = 9 * Cin; g.C = y_dev; g.ldc = Cout; g.M = (long long)F * OH * OW; g.N = Cout; g.K = 9 * Cin;
    g.bias = bias_dev; g.act = post_relu ? ACT_RELU : ACT_NONE; g.R1 = R1_dev; g.ldr1 = Cout; g.R2 = R2_dev; g.ldr2 = Cout;
    g.loader = LOAD_CONV3; g.cH = H; g.cW = W; g.cC = Cin; g.cOH = OH; g.cOW = OW; g.cS = stride; g.pre_relu = pre_relu ? 1 : 0;
    g.ws = ws; g.ws_floats = ws_bytes / sizeof(float);
    return gemm(g, (hipStream_t)stream);
}

int edv_conv3x3(const float *x_dev, const float *wpacked_dev, const float *bias_dev, float *y_dev, int32_t F, int32_t H, int32_t W, int32_t Cin,
                int32_t Cout, int32_t stride, int32_t pre_relu, int32_t post_relu, const float *R1_dev, const float *R2_dev, void *stream) {
    return conv3x3_impl(x_dev, wpacked_dev, bias_dev, y_dev, F, H, W, Cin, Cout, stride, pre_relu, post_relu, R1_dev, R2_dev, nullptr, 0, stream);
}

int edv_conv3x3_ws(const float *x_dev, const float *wpacked_dev, const float *bias_dev, float *y_dev, int32_t F, int32_t H, int32_t W, int32_t Cin,
                   int32_t Cout, int32_t stride, int32_t pre_relu, int32_t post_relu, const float *R1_dev, const float *R2_dev, float *workspace_dev,
                   size_t workspace_bytes, void *stream) {
    return conv3x3_impl(x_dev, wpacked_dev, bias_dev, y_dev, F, H, W, Cin, Cout, stride, pre_relu, post_relu, R1_dev, R2_dev, workspace_dev,
                        workspace_bytes, stream);
}

int edv_pack_conv3x3(const float *w_dev, float *wpacked_dev, int32_t Cout, int32_t Cin, void *stream) {
    return pack_conv3x3(w_dev, wpacked_dev, Cout, Cin, (hipStream_t)stream);
}

int edv_conv_transpose(const float *x_dev, const float *w_dev, const float *b_dev, float *wpack_dev, float *bpack_dev, float *y_dev, int32_t F, int32_t h,
              int32_t w, int32_t C, int32_t s, void *stream) {
    EDV_CHECK(F > 0 && h > 0 && w > 0 && C > 0 && s > 0, "empty problem");
    EDV_TRY(pack_convT(w_dev, wpack_dev, b_dev, bpack_dev, C, C, s, (hipStream_t)stream));
    GemmDesc g;
    g.A = x_dev; g.lda = C; g.W = wpack_dev; g.ldw = C; g.C = y_dev; g.ldc = C; g.M = (long long)F * h * w; g.N = s * s * C; g.K = C;
    g.bias = bpack_dev; g.store = STORE_SHUFFLE; g.ps_s = s; g.ps_C = C; g.ps_h = h; g.ps_w = w;
    return gemm(g, (hipStream_t)stream);
}

size_t edv_attn_spatial_workspace(int32_t F, int32_t N, int32_t heads) { return attn_spatial_workspace(F, N, heads) * sizeof(float); }
int edv_attn_spatial(const float *qkv_dev, float *out_dev, int32_t F, int32_t N, int32_t heads, float *workspace_dev, size_t workspace_bytes,
                     float *lse_dev, void *stream) {
    return attn_spatial(qkv_dev, out_dev, F, N, heads, workspace_dev, workspace_bytes / sizeof(float), (hipStream_t)stream, lse_dev);
}
size_t edv_attn_spatial_x6_workspace(int32_t F, int32_t N, int32_t heads) { return attn_spatial_workspace(F, N, heads, true) * sizeof(float); }
int edv_attn_spatial_x6(const float *qkv_dev, float *out_dev, int32_t F, int32_t N, int32_t heads, float *workspace_dev, size_t workspace_bytes, void *stream) {
    return attn_spatial(qkv_dev, out_dev, F, N, heads, workspace_dev, workspace_bytes / sizeof(float), (hipStream_t)stream, nullptr, true);
}

size_t edv_attn_spatial_bwd_workspace(int32_t F, int32_t N, int32_t heads) { return attn_spatial_bwd_workspace(F, N, heads) * sizeof(float); }
int edv_attn_spatial_bwd(const float *qkv_dev, const float *out_dev, const float *dout_dev, const float *lse_dev, float *delta_dev, float *dqkv_dev,
                         int32_t F, int32_t N, int32_t heads, float *workspace_dev, size_t workspace_bytes, void *stream) {
    return attn_spatial_bwd(qkv_dev, out_dev, dout_dev, lse_dev, delta_dev, dqkv_dev, F, N, heads, workspace_dev, workspace_bytes / sizeof(float),
                            (hipStream_t)stream);
}
int edv_layernorm_bwd(const float *x_dev, const float *w_dev, const float *dy_dev, float *dx_dev, int64_t rows, int32_t dim, float eps, int32_t accumulate,
                      void *stream) {
    return layernorm_bwd(x_dev, identity_map(), w_dev, dy_dev, identity_map(), dx_dev, identity_map(), rows, dim, eps, accumulate != 0, (hipStream_t)stream);
}
int edv_ew_bwd(const float *d_dev, const float *src_dev, const float *add_dev, float *out_dev, int64_t n, int32_t mode, void *stream) {
    return ew_bwd(d_dev, src_dev, add_dev, out_dev, n, mode, (hipStream_t)stream);
}
int edv_geglu_bwd(const float *x_dev, const float *dy_dev, float *dx_dev, int64_t M, int32_t inner, void *stream) {
    return geglu_bwd(x_dev, dy_dev, dx_dev, M, inner, (hipStream_t)stream);
}
int edv_transpose_scale(const float *W_dev, const float *gamma_dev, float *Wt_dev, int32_t N, int32_t K, void *stream) {
    return transpose_scale(W_dev, K, gamma_dev, Wt_dev, N, K, (hipStream_t)stream);
}
size_t edv_lora_grads_workspace(int64_t M, int32_t nin, int32_t nout, int32_t r) { return lora_grads_workspace(M, nin, nout, r) * sizeof(float); }
int edv_lora_grads(const float *x_dev, const float *g_dev, int64_t M, int32_t nin, int32_t nout, int32_t r, const float *A_dev, const float *B_dev,
                   const float *U_dev, const float *V_dev, float s, const float *gamma_dev, float *workspace_dev, size_t workspace_bytes, float *dA_dev,
                   float *dB_dev, float *dU_dev, float *dV_dev, void *stream) {
    EDV_CHECK(M > 0 && nin > 0 && nout > 0 && nin % 4 == 0 && nout % 4 == 0, "shape");
    return lora_grads(x_dev, nin, g_dev, nout, M, nin, nout, r, A_dev, B_dev, U_dev, V_dev, s, gamma_dev, workspace_dev, workspace_bytes / sizeof(float),
                      dA_dev, dB_dev, dU_dev, dV_dev, (hipStream_t)stream);
}
int edv_bilinear_bwd(const float *dy_dev, float *dx_dev, int32_t F, int32_t ih, int32_t iw, int32_t C, int32_t oh, int32_t ow, int32_t accumulate,
                     void *stream) {
    return bilinear_bwd(dy_dev, dx_dev, F, ih, iw, C, oh, ow, accumulate != 0, (hipStream_t)stream);
}
int edv_dot_channels_bwd(const float *g_dev, const float *disp_dev, const float *w_dev, const float *o2_dev, float *d_o2_dev, float *gz_out_dev,
                         int64_t npix, int32_t C, int32_t mode, void *stream) {
    return dot_channels_bwd(g_dev, disp_dev, w_dev, o2_dev, d_o2_dev, gz_out_dev, npix, C, mode, (hipStream_t)stream);
}
size_t edv_conv3x3_wgrad_workspace(int32_t F, int32_t H, int32_t W, int32_t Cin, int32_t Cout) {
    return conv3_wgrad_workspace(F, H, W, Cin, Cout) * sizeof(float);
}
int edv_conv3x3_wgrad(const float *x_dev, const float *dy_dev, float *dw_dev, int32_t F, int32_t H, int32_t W, int32_t Cin, int32_t Cout,
                      float *workspace_dev, size_t workspace_bytes, int32_t accumulate, void *stream) {
    return conv3_wgrad(x_dev, dy_dev, dw_dev, F, H, W, Cin, Cout, workspace_dev, workspace_bytes / sizeof(float), accumulate != 0, (hipStream_t)stream);
}
size_t edv_colsum_workspace(int32_t N) { return colsum_workspace(N) * sizeof(float); }
int edv_colsum_rows(const float *P_dev, const float *rowscale_dev, int64_t M, int32_t N, float *workspace_dev, size_t workspace_bytes, float *out_dev,
                    int32_t accumulate, void *stream) {
    return colsum_rows(P_dev, rowscale_dev, M, N, workspace_dev, workspace_bytes / sizeof(float), out_dev, accumulate != 0, (hipStream_t)stream);
}
int edv_groupnorm_bwd(const float *x_dev, const float *stats_dev, const float *w_dev, const float *dy_dev, float *sums_dev, float *dx_dev, int32_t F,
                      int32_t P, int32_t C, int32_t groups, int32_t accumulate, void *stream) {
    return groupnorm_bwd(x_dev, stats_dev, w_dev, dy_dev, sums_dev, dx_dev, F, P, C, groups, accumulate != 0, (hipStream_t)stream);
}
int edv_attn_temporal_bwd(const float *qkv_dev, const float *dout_dev, float *dqkv_dev, int32_t B, int32_t T, int32_t P, int32_t C, int32_t heads,
                          void *stream) {
    return attn_temporal_bwd(qkv_dev, dout_dev, dqkv_dev, B, T, P, C, heads, (hipStream_t)stream);
}
int edv_pack_conv3x3_bwd(const float *w_dev, float *wpacked_dev, int32_t Cout, int32_t Cin, void *stream) {
    return pack_conv3x3_bwd(w_dev, wpacked_dev, Cout, Cin, (hipStream_t)stream);
}
int edv_conv3x3_s2_bwd(const float *dy_dev, const float *wpacked_dev, float *dx_dev, int32_t F, int32_t H, int32_t W, int32_t Cin, int32_t Cout,
                       void *stream) {
    return conv3x3_s2_bwd(dy_dev, wpacked_dev, dx_dev, F, H, W, Cin, Cout, (hipStream_t)stream);
}
int edv_dilate2(const float *dy_dev, float *z_dev, int32_t F, int32_t H, int32_t W, int32_t C, void *stream) {
    return dilate2(dy_dev, z_dev, F, H, W, C, (hipStream_t)stream);
}
int edv_pixel_unshuffle(const float *dy_dev, float *A_dev, int32_t F, int32_t h, int32_t w, int32_t C, int32_t s, void *stream) {
    return pixel_unshuffle(dy_dev, A_dev, F, h, w, C, s, (hipStream_t)stream);
}

int edv_attn_temporal(const float *qkv_dev, float *out_dev, int32_t B, int32_t T, int32_t P, int32_t C, int32_t heads, void *stream) {
    return attn_temporal(qkv_dev, out_dev, B, T, P, C, heads, (hipStream_t)stream);
}

int edv_rope_qk(float *qkv_dev, const float *table_dev, int32_t B, int32_t T, int32_t P, int32_t C, int32_t transpose, void *stream) {
    return rope_qk(qkv_dev, table_dev, B, T, P, C, transpose != 0, (hipStream_t)stream);
}

size_t edv_groupnorm_workspace(int32_t F, int32_t P, int32_t C) { return groupnorm_workspace(F, P, C) * sizeof(float); }

int edv_groupnorm(const float *x_dev, const float *w_dev, const float *b_dev, float *y_dev, float *stats_dev, int32_t F, int32_t P, int32_t C,
                  int32_t groups, float eps, float *workspace_dev, size_t workspace_bytes, void *stream) {
    return groupnorm(x_dev, w_dev, b_dev, y_dev, stats_dev, F, P, C, groups, eps, (hipStream_t)stream, workspace_dev, workspace_bytes / sizeof(float));
}

// Test hook: every CU's whole LDS (160 KB) filled with `value` -- LDS keeps its contents between kernels, so a kernel that reads LDS bytes it never
// wrote (or relies on the DMA writing something it does not write) sees the poison.  tests/test_kernels_gpu.py::test_attention_on_poisoned_lds.
namespace edv {
namespace {
__global__ __launch_bounds__(256) void fill_lds_kernel(float value, float *sink) {
    extern __shared__ float lds_all[];
    for (int i = threadIdx.x; i < 160 * 256; i += 256) lds_all[i] = value;
    __syncthreads();
    if (sink && lds_all[(threadIdx.x * 97) % (160 * 256)] != value) *sink = 1.f;  // keeps the stores alive
}
}  // namespace
}  // namespace edv
int edv_debug_fill_lds(float value, void *stream) {
    int dev = 0, cus = 0;
    EDV_HIP(hipGetDevice(&dev));
    EDV_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    EDV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(edv::fill_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    EDV_LAUNCH(edv::fill_lds_kernel, dim3((unsigned)(8 * cus)), dim3(256), 160 * 1024, (hipStream_t)stream, value, (float *)nullptr);
    EDV_LAUNCH_OK();
    return 0;
}

int edv_geglu(const float *x_dev, float *y_dev, int64_t M, int32_t inner, void *stream) { return geglu(x_dev, y_dev, M, inner, (hipStream_t)stream); }

int edv_bilinear(const float *x_dev, float *y_dev, int32_t F, int32_t H, int32_t W, int32_t C, int32_t OH, int32_t OW, void *stream) {
    return bilinear(x_dev, y_dev, F, H, W, C, OH, OW, ACT_NONE, (hipStream_t)stream);
}

int edv_dot_channels(const float *x_dev, const float *w_dev, const float *b_dev, float *y_dev, int64_t M, int32_t C, int32_t act, void *stream) {
    EDV_CHECK(act == ACT_NONE || act == ACT_RELU || act == ACT_SIGMOID || act == ACT_SIGMOID_NEG, "act must be 0, 2, 3 or 4");
    return dot_channels(x_dev, w_dev, b_dev, y_dev, M, C, act, (hipStream_t)stream);
}

int edv_patchify(const float *x_dev, float *cols_dev, int32_t F, int32_t H, int32_t W, int32_t ih, int32_t iw, void *stream) {
    return patchify(x_dev, cols_dev, F, H, W, ih, iw, (hipStream_t)stream);
}

int edv_bicubic_pos(const float *grid_dev, float *out_dev, int32_t S, int32_t D, int32_t oh, int32_t ow, double scale_h, double scale_w, void *stream) {
    EDV_CHECK(scale_h > 0 && scale_w > 0, "scale factors must be positive");
    return bicubic_pos(grid_dev, out_dev, S, D, oh, ow, (float)(1.0 / scale_h), (float)(1.0 / scale_w), (hipStream_t)stream);
}

int edv_resize_bicubic(const float *x_dev, float *y_dev, int32_t planes, int32_t H, int32_t W, int32_t OH, int32_t OW, void *stream) {
    return resize_bicubic(x_dev, y_dev, planes, H, W, OH, OW, (hipStream_t)stream);
}

int edv_fold_lora(const float *W_dev, const float *A_dev, const float *B_dev, const float *U_dev, const float *V_dev, float scale, float *out_dev,
                  int32_t nout, int32_t nin, int32_t r, void *stream) {
    return fold_lora(W_dev, A_dev, B_dev, U_dev, V_dev, scale, out_dev, nout, nin, r, (hipStream_t)stream);
}

size_t edv_trainer_loss_workspace(int32_t N, int32_t H, int32_t W) { return trainer_loss_workspace(N, H, W) * sizeof(float); }
int edv_trainer_loss(const edv_trainer_loss_inputs *in, int32_t N, int32_t H, int32_t W, const edv_trainer_loss_weights *weights, float *losses_dev,
                     const edv_trainer_loss_grads *grads, float *workspace_dev, size_t workspace_bytes, void *stream) {
    EDV_CHECK(in && weights && grads, "null argument");
    TrainerLossIn a{};
    TrainerLossGrads g{};
    for (int s = 0; s < 4; ++s) {
        a.color[s] = in->color[s];
        a.disp[s] = in->disp[s];
        a.disp_h[s] = in->disp_h[s];
        a.disp_w[s] = in->disp_w[s];
        g.disp[s] = grads->disp[s];
        for (int n = 0; n < 2; ++n) {
            a.refined[s][n] = in->refined[s][n];
            a.registration[s][n] = in->registration[s][n];
            a.transform[s][n] = in->transform[s][n];
            a.position[s][n] = in->position[s][n];
            g.refined[s][n] = grads->refined[s][n];
            g.transform[s][n] = grads->transform[s][n];
        }
    }
    for (int n = 0; n < 2; ++n) {
        a.color_nb[n] = in->color_nb[n];
        a.T[n] = in->T[n];
        a.mask[n] = in->mask[n];
        g.T[n] = grads->T[n];
    }
    a.K = in->K;
    a.invK = in->invK;
    g.K = grads->K;
    g.invK = grads->invK;
    const TrainerLossW w{weights->disparity_smoothness, weights->transform_constraint, weights->transform_smoothness, weights->depth_reproj, weights->depth_flow,
                         weights->tune_temporal, weights->min_depth, weights->max_depth};
    return trainer_loss(a, N, H, W, w, losses_dev, g, workspace_dev, workspace_bytes / sizeof(float), (hipStream_t)stream);
}

size_t edv_photometric_loss_workspace(int32_t B, int32_t T, int32_t H, int32_t W) { return photometric_loss_workspace(B, T, H, W) * sizeof(float); }
int edv_photometric_loss(const float *frames_dev, const float *const disp_dev[4], const int32_t disp_h[4], const int32_t disp_w[4], int32_t B, int32_t T, int32_t H,
                         int32_t W, const float *K_dev, const float *invK_dev, const float *Tprev_dev, const float *Tnext_dev, float min_depth, float max_depth,
                         float disparity_smoothness, float *loss_dev, float *const grad_disp_dev[4], float *workspace_dev, size_t workspace_bytes, void *stream) {
    EDV_CHECK(disp_h && disp_w, "null argument");
    const int dh[4] = {disp_h[0], disp_h[1], disp_h[2], disp_h[3]}, dw[4] = {disp_w[0], disp_w[1], disp_w[2], disp_w[3]};
    return photometric_loss(frames_dev, disp_dev, dh, dw, B, T, H, W, K_dev, invK_dev, Tprev_dev, Tnext_dev, min_depth, max_depth, disparity_smoothness, loss_dev,
                            grad_disp_dev, workspace_dev, workspace_bytes / sizeof(float), (hipStream_t)stream);
}

}  // extern "C"
