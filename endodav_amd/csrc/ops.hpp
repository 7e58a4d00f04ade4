// Host-side launch functions of the HIP kernels (one per op family).  All pointers are device
// pointers; every function enqueues on `st` and returns 0 / non-zero (text via edv::get_error()).
#pragma once
#include "common.hpp"

namespace edv {

// ------------------------------------------------------------------------------------------
// GEMM family (gemm.hip):  C = epilogue(Aop · Wᵀ),  W [N,K] row-major (torch Linear layout).
//   epilogue:  v = acc + bias[n] + P1[p1_map(m), n];  v = act(v);  v *= gamma[n];  v += R1[r1_map(m), n];  v += R2[c_row, n]
// A operand:  LOAD_DENSE  A[a_map(m), k]
//             LOAD_CONV3  implicit im2col of a channels-last image x[F,H,W,Cin], k = (ky*3+kx)*Cin+ci,
//                         zero padding 1, stride cs, optional ReLU on the loaded value
// C store:    STORE_ROWS  C[c_map(m), n]
//             STORE_SHUFFLE  ConvTranspose2d with kernel == stride == s as a GEMM: m = (f,y,x) on a
//                         [ps_h, ps_w] grid, n = (dy*s+dx)*ps_C + co -> out[f, y*s+dy, x*s+dx, co]
// ------------------------------------------------------------------------------------------
enum { ACT_NONE = 0, ACT_GELU = 1, ACT_RELU = 2, ACT_SIGMOID = 3, ACT_SIGMOID_NEG = 4 };
enum { LOAD_DENSE = 0, LOAD_CONV3 = 1 };
enum { STORE_ROWS = 0, STORE_SHUFFLE = 1 };

struct GemmDesc {
    const float *A = nullptr;
    int lda = 0;
    RowMap a_map{0, 0, 0};
    const float *W = nullptr;
    int ldw = 0;
    float *C = nullptr;
    int ldc = 0;
    RowMap c_map{0, 0, 0};
    long long M = 0;
    int N = 0, K = 0;
    const float *bias = nullptr;
    int act = ACT_NONE;
    const float *gamma = nullptr;
    const float *R1 = nullptr;
    int ldr1 = 0;
    RowMap r1_map{0, 0, 0};
    const float *R2 = nullptr;
    int ldr2 = 0;
    const float *P1 = nullptr;  // pre-activation addend (per-frame readout bias of use_clstoken, dpt_pyramid.py:54-57)
    int ldp1 = 0;
    RowMap p1_map{0, 0, 0};
    int loader = LOAD_DENSE;
    int cH = 0, cW = 0, cC = 0, cOH = 0, cOW = 0, cS = 1, pre_relu = 0;
    int store = STORE_ROWS;
    // GEGLU epilogue (gemm_dma.hip, EP = 6): W and bias come interleaved in 32-row blocks -- value rows 32b..32b+31 of the Linear, then the gate rows
    // N/2 + 32b.. of the same block (pack_geglu) -- and C [M, N/2] (ldc >= N/2) receives value * gelu(gate): GEGLU.forward, motion_module.py
    int geglu = 0;
    int ps_s = 0, ps_C = 0, ps_h = 0, ps_w = 0;
    // stream-K split workspace (gemm_workspace() floats, 16-byte aligned, one per concurrently running GEMM; its first
    // gemm_counter_bytes() must be zero before the first launch and are left zero by every launch);
    // nullptr = plain grid, one workgroup per output tile
    float *ws = nullptr;
    size_t ws_floats = 0;
    // bf16 planes [3][N][K] of W (gemm_x6_split): with them a dense STORE_ROWS GEMM runs its products on the bf16 matrix pipe (gemm_x6.hip)
    const void *Wx6 = nullptr;
};
int gemm(const GemmDesc &d, hipStream_t st);
// "bf16 x 6" GEMM (gemm_x6.hip): fp32 in / out, six bf16 MFMAs per product term set, error below fp32's unit roundoff per term
size_t gemm_x6_planes_bytes(int N, int K);
int gemm_x6_split(const float *W, void *planes, int N, int K, hipStream_t st);
bool gemm_x6_supported(const GemmDesc &d);
int gemm_x6(const GemmDesc &d, hipStream_t st);
size_t gemm_workspace();  // floats; enough for any shape on the current device
size_t gemm_counter_bytes();  // the zero-initialised arrival counters at the head of the workspace
// LDS-DMA staged variant (gemm_dma.hip): dense A, K % 32 == 0; picked by gemm() for small/medium grids.
bool gemm_dma_supported(const GemmDesc &d);
bool gemm_geglu_supported(const GemmDesc &d);  // the GEGLU epilogue: dense A, K % 32 == 0, N % 64 == 0, 32-bit output offsets
int pack_geglu(const float *w, const float *b, float *wi, float *bi, int N, int K, hipStream_t st);  // N = both halves (8C)
int gemm_dma(const GemmDesc &d, hipStream_t st);
// LDS-DMA implicit-GEMM 3x3 convolution (conv_dma.hip): Cin % 32 == 0, any stride the GemmDesc allows
bool conv_dma_supported(const GemmDesc &d);
int conv_dma(const GemmDesc &d, hipStream_t st);
// flops of the last-launched gemm tile choice, for the bench's roofline bookkeeping
const char *gemm_kernel_name(const GemmDesc &d);

// ------------------------------------------------------------------------------------------
// attention (attn_spatial.hip, temporal.hip)
// ------------------------------------------------------------------------------------------
size_t attn_spatial_workspace(int F, int N, int heads, bool x6 = false);  // floats (x6: the bf16 x 6 kernel's task split)
// lse (optional, [F, heads, N]): per-row log-sum-exp of the scores in base 2, for attn_spatial_bwd
// x6: both products as six bf16 MFMAs on three-term bf16 splits (attn_x6_kernel; fp32 in / out / accumulate, sequences longer than 128)
int attn_spatial(const float *qkv, float *out, int F, int N, int heads, float *ws, size_t ws_floats, hipStream_t st, float *lse = nullptr, bool x6 = false);
size_t attn_spatial_bwd_workspace(int F, int N, int heads);  // floats
int attn_spatial_bwd(const float *qkv, const float *out, const float *dout, const float *lse, float *delta, float *dqkv, int F, int N, int heads, float *ws,
                     size_t ws_floats, hipStream_t st);
int attn_temporal(const float *qkv, float *out, int B, int T, int P, int C, int heads, hipStream_t st);
int geglu(const float *x, float *y, long long M, int inner, hipStream_t st);
// pe="rope": q|k of qkv [B*T*P, 3C] rotated in place by the tabulated cos|sin [>=T, C/2, 2]; transpose = the rotation's adjoint
int rope_qk(float *qkv, const float *table, int B, int T, int P, int C, bool transpose, hipStream_t st);

// ------------------------------------------------------------------------------------------
// norms (norms.hip)
// ------------------------------------------------------------------------------------------
// y[out_map(m)] (+)= act(LN(x[in_map(m)]) * w + b (+ pe[(m / rows_per_frame) % T]));  act: ACT_NONE / ACT_GELU
int layernorm(const float *x, RowMap in_map, const float *w, const float *b, float *y, RowMap out_map, long long rows, int dim,
              float eps, const float *pe, int rows_per_frame, int T, hipStream_t st, int act = ACT_NONE, bool accumulate = false);
size_t groupnorm_workspace(int F, int P, int C);  // floats, for the coalesced two-stage statistics
int groupnorm(const float *x, const float *w, const float *b, float *y, float *stats, int F, int P, int C, int groups, float eps, hipStream_t st,
              float *part = nullptr, size_t part_floats = 0);  // part = null: one workgroup per (frame, group) reads its strided slab

// ------------------------------------------------------------------------------------------
// resampling / elementwise (resample.hip)
// ------------------------------------------------------------------------------------------
int patchify(const float *x, float *cols, int F, int H, int W, int ih, int iw, hipStream_t st, int ld = 588)  /* ld: row stride of cols, >= 588; the tail of a row is zero-filled */;
int bilinear(const float *x, float *y, int F, int H, int W, int C, int OH, int OW, int act, hipStream_t st, const float *add = nullptr);  // y = up(x) (+ add)
int dot_channels(const float *x, const float *w, const float *b, float *y, long long M, int C, int act, hipStream_t st);
int cls_rows(const float *cls, const float *pos, float *tokens, int F, int ntok, int D, hipStream_t st);
int sigmoid_inplace(float *x, long long n, hipStream_t st);
// planar bicubic image resize (cv2.INTER_CUBIC semantics) for infer_video_depth's pre-resize
int resize_bicubic(const float *x, float *y, int NP, int H, int W, int OH, int OW, hipStream_t st);
// pos-embed bicubic resample (vision_transformer.py:186-217): grid [S,S,D] -> [oh,ow,D]
int bicubic_pos(const float *grid, float *out, int S, int D, int oh, int ow, float scale_h, float scale_w, hipStream_t st);

// ------------------------------------------------------------------------------------------
// weight packing (prep.hip)
// ------------------------------------------------------------------------------------------
int pack_conv3x3(const float *w, float *out, int Cout, int Cin, hipStream_t st);            // [Co,Ci,3,3] -> [Co][3][3][Ci]
int pack_convT(const float *w, float *wout, const float *b, float *bout, int Cin, int Cout, int s, hipStream_t st);  // [Ci,Co,s,s] -> [(dy,dx,co)][ci]
int copy_f32(const float *src, float *dst, long long n, hipStream_t st);
// W_eff = W + scale * (B∘V)(A∘U)   (U,V may be null);   ssb: W_eff = a ∘ W ∘ b
int fold_lora(const float *W, const float *A, const float *B, const float *U, const float *V, float scale, float *out, int nout, int nin,
              int r, hipStream_t st);
int fold_ssb(const float *W, const float *a, const float *b, float *out, int nout, int nin, hipStream_t st);
// eval-mode BatchNorm after a convolution: rows of the packed weight [nout, K] scaled in place, bout = folded bias
int fold_bn(float *w, const float *b, const float *gamma, const float *beta, const float *mean, const float *var, float eps, float *bout, int nout, int K,
            hipStream_t st);
// W_eff += Utop diag(idx) Vtop  (DashLinear after warm-up)
int fold_dash(const float *Utop, const float *idx, const float *Vtop, float *inout, int nout, int nin, int r, hipStream_t st);

// ------------------------------------------------------------------------------------------
// backward (bwd.hip, attn_spatial_bwd.hip): input gradients of the frozen operators + LoRA factor gradients
int layernorm_bwd(const float *x, RowMap xmap, const float *w, const float *dy, RowMap dymap, float *dx, RowMap dxmap, long long rows, int dim, float eps,
                  bool accumulate, hipStream_t st);
// out = f(d) + (add ? add : 0);  mode 0: f = d, 1: d * gelu'(src), 2: src > 0 ? d : 0
int ew_bwd(const float *d, const float *src, const float *add, float *out, long long n, int mode, hipStream_t st);
int sigmoid_bwd(const float *g, const float *s, float *out, long long n, hipStream_t st);  // out = g * s * (1 - s), s = the sigmoid's output
int geglu_bwd(const float *x, const float *dy, float *dx, long long M, int inner, hipStream_t st);
int transpose_scale(const float *W, int ldw, const float *gamma, float *Wt, int N, int K, hipStream_t st);  // Wt[k,n] = W[n,k] * gamma[n]
int skinny_xwt(const float *X, long long M, int K, int ldx, const float *Wr, int r, float *T, hipStream_t st);  // T[M,r] = X Wr^T, Wr [r,K]
constexpr int TALL_SPLITS = 64;
size_t tall_tn_workspace(int N, int r);  // floats
// out[N,r] = scale * rowscale[n] * sum_m Y[m,n] T[m,j]   (deterministic two-stage reduction)
int tall_tn(const float *Y, int ldy, const float *T, long long M, int N, int r, float scale, const float *rowscale, float *part, float *out, hipStream_t st);
int lora_grad_finalize(const float *dBp, const float *dApT, const float *A, const float *Bm, const float *U, const float *V, float *dA, float *dB, float *dU,
                       float *dV, int nout, int nin, int r, hipStream_t st);
size_t lora_grads_workspace(long long M, int nin, int nout, int r);  // floats
int lora_grads(const float *X, int ldx, const float *G, int ldg, long long M, int nin, int nout, int r, const float *A, const float *Bm, const float *U,
               const float *V, float s, const float *gamma, float *ws, size_t ws_floats, float *dA, float *dB, float *dU, float *dV, hipStream_t st);
// Linear_SSB: Wa = W * a^T, gb = gamma * b;  col_dot: out[n] = scale[n] * sum_m P[m,n] Q[m,n] (part: TALL_SPLITS * N floats)
int ssb_prep(const float *W, const float *a, const float *b, const float *gamma, float *Wa, float *gb, int nout, int nin, hipStream_t st);
int col_dot(const float *P, const float *Q, long long M, int N, const float *scale, float *part, float *out, hipStream_t st);
int bilinear_bwd(const float *dy, float *dx, int F, int ih, int iw, int C, int oh, int ow, bool accumulate, hipStream_t st);
// 1x1 conv to one channel + output activation: mode 0 ReLU (VDA head), 1 sigmoid(z), 2 sigmoid(-z) (HeadDepth); gz_out (optional) = dL/dz
int dot_channels_bwd(const float *g, const float *disp, const float *w, const float *o2, float *d_o2, float *gz_out, long long npix, int C, int mode,
                     hipStream_t st);
// weight gradient of a 3x3 / stride 1 / padding 1 convolution (wgrad.hip): x [F,H,W,Cin], dy [F,H,W,Cout] -> dw [Cout,Cin,3,3] (torch layout)
size_t conv3_wgrad_workspace(int F, int H, int W, int Cin, int Cout);  // floats
int conv3_wgrad(const float *x, const float *dy, float *dw, int F, int H, int W, int Cin, int Cout, float *ws, size_t ws_floats, bool accumulate,
                hipStream_t st);
// out[n] (+)= sum_m rowscale[m] * P[m, n]  (rowscale may be null; N a power of two in 4..1024, or 1): bias gradients, 1x1-conv weight gradient
size_t colsum_workspace(int N);  // floats
int colsum_rows(const float *P, const float *rowscale, long long M, int N, float *ws, size_t ws_floats, float *out, bool accumulate, hipStream_t st);
int groupnorm_bwd(const float *x, const float *stats, const float *w, const float *dy, float *sums, float *dx, int F, int P, int C, int groups, bool accumulate,
                  hipStream_t st);
int attn_temporal_bwd(const float *qkv, const float *dout, float *dqkv, int B, int T, int P, int C, int heads, hipStream_t st);
int pixel_unshuffle(const float *dy, float *A, int F, int h, int w, int C, int s, hipStream_t st);
int conv3x3_s2_bwd(const float *dy, const float *wpacked, float *dx, int F, int H, int W, int Cin, int Cout, hipStream_t st);
int dilate2(const float *dy, float *z, int F, int H, int W, int C, hipStream_t st);  // z[f,2oy,2ox] = dy[f,oy,ox], zeros elsewhere
int pack_conv3x3_bwd(const float *w, float *out, int Cout, int Cin, hipStream_t st);  // [Co,Ci,3,3] -> [Ci][3][3][Co], taps flipped

// ------------------------------------------------------------------------------------------
// the trainer's whole loss and every gradient its autograd reaches (loss_trainer.hip; include/endodav_hip.h documents the tensors)
struct TrainerLossIn {
    const float *color[4];
    const float *color_nb[2];
    const float *K, *invK;
    const float *T[2];
    const float *refined[4][2], *registration[4][2], *transform[4][2];
    const float *mask[2];
    const float *position[4][2];
    const float *disp[4];
    int disp_h[4], disp_w[4];
};
struct TrainerLossW {
    float disparity_smoothness, transform_constraint, transform_smoothness, depth_reproj, depth_flow;
    int tune_temporal;
    float min_depth, max_depth;
};
struct TrainerLossGrads {
    float *disp[4];
    float *refined[4][2], *transform[4][2];
    float *K, *invK, *T[2];
};
size_t trainer_loss_workspace(int N, int H, int W);  // floats
int trainer_loss(const TrainerLossIn &in, int N, int H, int W, const TrainerLossW &w, float *losses, const TrainerLossGrads &g, float *ws, size_t ws_floats, hipStream_t st);
// the fine-tune step's photometric loss + dL/d disp (loss.hip)
size_t photometric_loss_workspace(int B, int T, int H, int W);  // floats
int photometric_loss(const float *frames, const float *const disp[4], const int *dh, const int *dw, int B, int T, int H, int W, const float *K, const float *invK,
                     const float *Tprev, const float *Tnext, float min_depth, float max_depth, float smoothness, float *loss, float *const grad[4], float *ws,
                     size_t ws_floats, hipStream_t st);

}  // namespace edv
