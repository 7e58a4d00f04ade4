/*
 * endodav_hip.h — C ABI of libendodav_hip.so: the MI355X (gfx950) implementation of
 * EndoDAV's per-clip forward.
 *
 * The reference has no FFI / plugin registry for this path: its boundary is the Python
 * class surface models/endodav/__init__.py:1-2.  This header is the
 * boundary the build's own host module (endodav_amd/endodav.py, a mirror of
 * models/endodav/endodav.py:53-160) binds through ctypes; INTEGRATION.md shows the stub a
 * maintainer of the reference would add.  Rules of the ABI:
 *
 *   - plain C types only: device pointers, sizes, an opaque context; no torch types;
 *   - every pointer named *_dev is a device pointer on the CURRENT HIP device;
 *   - every call is stream-ordered on `stream` (a hipStream_t passed as void*; NULL = the
 *     legacy default stream) and performs no host/device synchronisation, except
 *     edv_create/edv_destroy and the first edv_forward at a new clip geometry, which
 *     allocate the activation workspace (hipMalloc);
 *   - return value 0 = ok, non-zero = error, text in edv_last_error(); nothing throws;
 *   - one context per device and per host thread (nn.DataParallel replicas each own one);
 *     the library keeps no global mutable state besides the thread-local error string.
 *
 * Data layout: activations are fp32, tokens-major [frames*tokens, D] in the
 * encoder and channels-last [frames, h, w, C] in the DPT head; the clip comes in as the
 * reference's [B, T, 3, H, W] and the disparities go out as [B*T, 1, h_s, w_s].
 */
#ifndef ENDODAV_HIP_H
#define ENDODAV_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EDV_ABI_VERSION 10 /* 9: edv_trainer_loss; 8: edv_debug_fill_lds (test hook); 7: the split-bf16 experiment entry points left the library */

enum edv_lora_type { EDV_LORA_NONE = 0, EDV_LORA_LORA = 1, EDV_LORA_DVLORA = 2, EDV_LORA_SSB = 3, EDV_LORA_DASH = 4 };

/* Mirrors the constructor of the reference model, models/endodav/endodav.py:53-73, after the
 * encoder name has been resolved to dimensions (vision_transformer.py:352-398). */
typedef struct edv_config {
    int32_t abi_version;        /* EDV_ABI_VERSION */
    int32_t embed_dim;          /* 384 / 768 / 1024 */
    int32_t depth;              /* 12 / 12 / 24 */
    int32_t num_heads;          /* 6 / 12 / 16 (head dim must be 64) */
    int32_t taps[4];            /* intermediate_layer_idx, endodav.py:76-79 */
    int32_t features;           /* DPT `features` */
    int32_t out_channels[4];    /* DPT `out_channels` */
    int32_t image_h, image_w;   /* `image_shape`; multiples of 14 (patch_embed.py:72-73) */
    int32_t num_frames;         /* temporal_max_len; T <= num_frames (motion_module.py:197) */
    int32_t pos_tokens;         /* rows of pretrained.pos_embed (1370, or 257 for vitl) */
    int32_t lora_type;          /* enum edv_lora_type */
    int32_t lora_rank;          /* r */
    int32_t include_cls_token;  /* vision_transformer.py:229-230 */
    int32_t conv_head;          /* 1 = four HeadDepth heads (default), 0 = disable_conv_head */
    int32_t inv_sigmoid;        /* dpt_pyramid.py:104 */
    int32_t out_sigmoid;        /* dpt_pyramid.py:97-101 */
    int32_t temporal_lora;      /* endodav.py:119-137 */
    int32_t dash_active;        /* DashLinear past warm-up: add U_top diag(idx) Vt_top */
    int32_t use_clstoken;       /* DPT readout projections, dpt_pyramid.py:54-57 */
    uint32_t residual_mask;     /* bit i set: encoder block i carries a ResBottleneckBlock (block.py:146-150);
                                 * the reference hard-wires its grid to 16x20 patches = image_shape (224, 280) */
    int32_t use_bn;             /* eval-mode BatchNorm2d in the ResidualConvUnits (util/blocks.py:60-62,80-86), folded into the convolutions */
    int32_t pe_rope;            /* pe="rope" (motion_module.py:221-225,252-255): rotary q/k instead of the additive sinusoid;
                                 * needs a bound "<attention block>.freqs_cis" table [num_frames, C/2, 2] (cos, sin) per attention block */
} edv_config;

typedef struct edv_ctx edv_ctx;

/* ---- life cycle ------------------------------------------------------------------------ */
int edv_abi_version(void);
const char *edv_last_error(void);
int edv_create(const edv_config *cfg, edv_ctx **out);
int edv_destroy(edv_ctx *ctx);

/* Bind one entry of the model's state_dict (same key names as the reference)
 * to device memory owned by the caller; zero-copy, the pointer must stay valid. */
int edv_bind_param(edv_ctx *ctx, const char *name, const float *data_dev, const int64_t *shape, int32_t ndim);

/* Re-derive the packed weights from the bound parameters: LoRA folded into fc1/fc2
 * (mylora/layers.py:148-157,384-393,423-430), conv kernels repacked to [Cout][kh][kw][Cin],
 * q/k/v concatenated, the position table resampled (vision_transformer.py:186-217).
 * Call after every change of the bound tensors' contents. */
int edv_prepare(edv_ctx *ctx, void *stream);

/* The cheap form of edv_prepare for the fine-tune loop (trainer_end_to_end_video.py:427-431: only the optimizer writes, and only
 * into trainable tensors): re-folds the linears that carry LoRA factors (mlp.fc1/fc2 of every block and, with temporal_lora, ff.net.2
 * of the motion modules), re-packs the trainable convolutions (conv_depth_* or scratch.output_conv*, residual_*) and, once a backward
 * has run, their transposed / flipped copies.  Everything frozen keeps its packing.  Valid only when no bound pointer changed and
 * only those tensors' contents did; otherwise call edv_prepare. */
int edv_refresh_lora(edv_ctx *ctx, void *stream);

/* endodav.forward (endodav.py:150-160).  x_dev: [B,T,3,H,W] fp32 in [0,1].  disp_dev[s] receives
 * ("disp", s): VDA head [B*T,1,ih,iw], [..ih/2..], ...; conv head [B*T,1,2*ph*8*.. see
 * edv_output_shape.  Requires T <= num_frames (motion_module.py:197) and T <= 32: the temporal-attention kernels are built for the
 * reference's window length (INFER_LEN = num_frames default = 32, endodav.py:47,62); a longer clip is refused before any work is
 * enqueued.  On an error the internal streams are drained before the call returns. */
int edv_forward(edv_ctx *ctx, const float *x_dev, int32_t B, int32_t T, int32_t H, int32_t W,
                float *const disp_dev[4], void *stream);
/* h/w of ("disp", s) for this configuration. */
int edv_output_shape(const edv_ctx *ctx, int32_t scale, int32_t *h, int32_t *w);
/* Debug taps: copy of an internal stage of the last edv_forward, for the per-stage parity
 * tests.  name in {"tokens","block0","tap0".."tap3","mm0","mm1","path4".."path1"}; head
 * stages are channels-last [frames,h,w,C].  Returns element count through *n (dst may be NULL). */
int edv_stage_copy(edv_ctx *ctx, const char *name, float *dst_dev, size_t *n, void *stream);
/* Debug: when on, edv_forward also snapshots the in-place residual stream ("tokens", "block0"). */
int edv_set_capture(edv_ctx *ctx, int on);
/* Live per-kernel timing for bench.py's roofline: bracket every launch of the selected kernel classes with a
 * HIP event pair on the launch stream.  Classes: 0 dense GEMM (Linear / 1x1 conv), 1 3x3 conv, 2 spatial
 * attention, 3 temporal attention, 4 LayerNorm, 5 other, 6 (read only; recorded with 0) the dense GEMMs of the encoder blocks,
 * 7 GroupNorm (statistics + apply), 8 bilinear resample, 9 GEGLU, 10 the final 1x1 convolution to one channel, 11 resize +
 * normalise + im2col of the input frames.  edv_profile_read waits for the recorded events, returns the number of launches and
 * their summed duration, and clears that class. */
int edv_profile_enable(edv_ctx *ctx, uint32_t class_mask);
/* Frames are independent in the encoder.  n = 0 (default): automatic -- two frame groups on internal streams while a block's
 * GEMMs are short (tokens x width <= 17 M: ViT-S up to T = 32, ViT-B up to T = 16), one otherwise; n = 1..4: that many.  With more than one group the attention of one
 * group runs beside the GEMMs of another, so per-kernel event brackets then measure time-shared launches: set 1 for the
 * steps whose kernels are being timed (bench.py does).  Environment EDV_ENC_STREAMS sets the initial value; n = -1 restores it. */
int edv_set_encoder_streams(edv_ctx *ctx, int32_t n);
/* Arithmetic of the encoder's linears (qkv, proj, fc1, fc2) in inference.  Inputs, outputs and accumulation are fp32 either way.
 *   EDV_PRODUCTS_F32     v_mfma_f32_32x32x2_f32: fp32 products on the fp32 matrix pipe (157 TFLOP/s peak)
 *   EDV_PRODUCTS_BF16X6  each operand split into three bf16 terms, six bf16 MFMAs per 16 k (gemm_x6.hip): per-term error below fp32's unit
 *                        roundoff (the three dropped cross terms are <= 2^-26 of a product), 192 instead of 512 matrix-pipe cycles.
 * The initial value comes from the environment variable EDV_PRODUCTS ("f32" | "bf16x6"), else F32.  Training forwards always use F32.
 * After edv_prepare the call builds the weights' bf16 planes on `stream` (+ 1.5 x the encoder linears' bytes). */
enum { EDV_PRODUCTS_F32 = 0, EDV_PRODUCTS_BF16X6 = 1 };
int edv_set_products(edv_ctx *ctx, int32_t products, void *stream);
int edv_get_products(const edv_ctx *ctx);
int edv_profile_set_mask(edv_ctx *ctx, uint32_t class_mask); /* change the bracketed classes, keep what was recorded */
int edv_profile_read(edv_ctx *ctx, int32_t kernel_class, int32_t *launches, double *total_ms);
/* Algorithmic work of the launches bracketed since edv_profile_enable / the last call: classes 0, 1, 6 book 2 M N K FLOP and every
 * operand once in bytes (4 (M K + N K + M N (+ M N per residual))); the bandwidth-bound classes 4, 7 .. 11 book bytes only (input
 * tensor(s) + output tensor, once each); 0 for classes that do not account. */
int edv_profile_work(edv_ctx *ctx, int32_t kernel_class, double *flops, double *bytes);
/* Bytes of device memory the context currently holds (packed weights + workspace). */
size_t edv_device_bytes(const edv_ctx *ctx);
/* Seconds spent in the dominant kernels are measured by the caller with HIP events; this
 * returns the number of kernel launches one edv_forward issued last time. */
int edv_last_launch_count(const edv_ctx *ctx);

/* ---- per-kernel entry points (unit tests, micro-benchmarks) --------------------------------
 * All tensors fp32, row-major, device pointers. */

/* y[m,:] = LN(x[m,:]) * w + b (+ pe[(m / rows_per_frame) % T, :] when pe_dev != NULL).
 * vision_transformer.py:97 (eps 1e-6), motion_module.py:155,161 (eps 1e-5) + :197. */
int edv_layernorm(const float *x_dev, const float *w_dev, const float *b_dev, float *y_dev, int64_t rows, int32_t dim,
                  float eps, const float *pe_dev, int32_t rows_per_frame, int32_t T, void *stream);

/* C[M,N] = epilogue(A[M,K] · W[N,K]ᵀ):  v = acc + bias[n]; v = act(v); v *= gamma[n]; v += R[m,n].
 * act: 0 none, 1 exact-erf GELU, 2 ReLU.  bias/gamma/R may be NULL.  K % 4 == 0.
 * F.linear / 1x1 conv with the epilogues of block.py:144-145, mlp.py:34-37. */
int edv_gemm(const float *A_dev, const float *W_dev, float *C_dev, int64_t M, int32_t N, int32_t K, const float *bias_dev,
             int32_t act, const float *gamma_dev, const float *R_dev, float *workspace_dev, size_t workspace_bytes, void *stream);
/* Stream-K workspace for edv_gemm, in BYTES (one size fits every shape on this device; one workspace per concurrently running
 * GEMM).  With a workspace the GEMM runs as persistent workgroups that split the last partial round of output tiles along K;
 * the last piece of a tile to arrive merges the pieces in a fixed order and applies the epilogue (no second launch; same
 * results up to fp32 summation order, reproducible run to run).  The workspace must be ZERO-FILLED once before its first use
 * (it starts with per-tile arrival counters, which every launch leaves at zero).  workspace_dev = NULL: one workgroup per tile. */
size_t edv_gemm_workspace(void);
/* The same GEMM with its products on the bf16 matrix pipe (EDV_PRODUCTS_BF16X6): split the weight once into planes_dev
 * (edv_gemm_x6_planes_bytes(N, K) bytes), then call edv_gemm_x6 with the planes in place of W.  K % 16 == 0, N >= 64. */
size_t edv_gemm_x6_planes_bytes(int32_t N, int32_t K);
int edv_gemm_x6_split(const float *W_dev, void *planes_dev, int32_t N, int32_t K, void *stream);
int edv_gemm_x6(const float *A_dev, const void *planes_dev, float *C_dev, int64_t M, int32_t N, int32_t K, const float *bias_dev, int32_t act,
                const float *gamma_dev, const float *R_dev, float *workspace_dev, size_t workspace_bytes, void *stream);

/* GEGLU feed-forward input projection of the motion modules (motion_module.py: GEGLU.forward = x, gate = proj(h).chunk(2, -1); x * gelu(gate)),
 * fused: C[M, N/2] = (A W_v^T + b_v) * gelu(A W_g^T + b_g) in ONE launch -- the [M, N] projection is never written.  edv_pack_geglu interleaves the
 * Linear's [N, K] weight and [N] bias in 32-row blocks (value rows 32b.., then their gate rows N/2 + 32b..) so that a 64-column tile holds values
 * and gates of the same 32 outputs; edv_gemm_geglu takes the packed pair.  N = both halves (8 C), N % 64 == 0, K % 32 == 0. */
int edv_pack_geglu(const float *w_dev, const float *b_dev, float *wi_dev, float *bi_dev, int32_t N, int32_t K, void *stream);
int edv_gemm_geglu(const float *A_dev, const float *Wi_dev, const float *bi_dev, float *C_dev, int64_t M, int32_t N, int32_t K, void *stream);

/* 3x3 convolution, padding 1, stride 1 or 2, channels-last: x [F,H,W,Cin], w packed
 * [Cout][3][3][Cin], y [F,OH,OW,Cout]; optional ReLU on the input (util/blocks.py:79-85),
 * bias, ReLU on the output and up to two residual tensors shaped like y. */
int edv_conv3x3(const float *x_dev, const float *wpacked_dev, const float *bias_dev, float *y_dev, int32_t F, int32_t H, int32_t W,
                int32_t Cin, int32_t Cout, int32_t stride, int32_t pre_relu, int32_t post_relu, const float *R1_dev,
                const float *R2_dev, void *stream);
/* The same with a stream-K workspace (edv_gemm_workspace() bytes, zero-filled once, as for edv_gemm): grids that do not fill the part
 * (e.g. 384 -> 64 channels at 19x19: 46 tiles of 108 k-tiles) are split along K over the resident workgroups and merged in-kernel. */
int edv_conv3x3_ws(const float *x_dev, const float *wpacked_dev, const float *bias_dev, float *y_dev, int32_t F, int32_t H, int32_t W,
                   int32_t Cin, int32_t Cout, int32_t stride, int32_t pre_relu, int32_t post_relu, const float *R1_dev,
                   const float *R2_dev, float *workspace_dev, size_t workspace_bytes, void *stream);
/* Repack a torch Conv2d weight [Cout,Cin,3,3] to [Cout][3][3][Cin]. */
int edv_pack_conv3x3(const float *w_dev, float *wpacked_dev, int32_t Cout, int32_t Cin, void *stream);

/* Encoder self-attention, layers/attention.py:56-69: qkv [F*N, 3*heads*64] as produced by the
 * qkv linear (columns ordered [3][heads][64]) -> out [F*N, heads*64].  The kernel runs as persistent workgroups
 * that split the last, partial round of (frame, head, query-block) tasks along the key axis; the pieces meet in a
 * device workspace of edv_attn_spatial_workspace(F, N, heads) BYTES (0 when nothing is split; 16-byte aligned,
 * owned by the caller, one per concurrently running call). */
size_t edv_attn_spatial_workspace(int32_t F, int32_t N, int32_t heads);
int edv_attn_spatial(const float *qkv_dev, float *out_dev, int32_t F, int32_t N, int32_t heads, float *workspace_dev,
                     size_t workspace_bytes, float *lse_dev, void *stream);
/* lse_dev (optional, [F, heads, N]): per-row log-sum-exp of the scaled scores in base 2, the only extra state the
 * backward needs.  edv_attn_spatial_bwd: dqkv [F*N, 3*heads*64] (like qkv) from qkv, out, dout [F*N, heads*64] and lse;
 * delta_dev is [F, heads, N] floats of scratch. */
/* The same attention with both products on the bf16 matrix pipe (EDV_PRODUCTS_BF16X6: three-term bf16 splits of q, k, v and of the probabilities,
 * six bf16 MFMAs per 16 k, fp32 accumulate and fp32 softmax; sequences longer than 128, shorter ones run the fp32 kernel).  Its own workspace size. */
size_t edv_attn_spatial_x6_workspace(int32_t F, int32_t N, int32_t heads);
int edv_attn_spatial_x6(const float *qkv_dev, float *out_dev, int32_t F, int32_t N, int32_t heads, float *workspace_dev, size_t workspace_bytes,
                        void *stream);
size_t edv_attn_spatial_bwd_workspace(int32_t F, int32_t N, int32_t heads); /* bytes; the backward splits its last partial round too */
int edv_attn_spatial_bwd(const float *qkv_dev, const float *out_dev, const float *dout_dev, const float *lse_dev, float *delta_dev,
                         float *dqkv_dev, int32_t F, int32_t N, int32_t heads, float *workspace_dev, size_t workspace_bytes, void *stream);

/* Temporal attention core, motion_module.py:230-297 + attention.py:182-211: qkv [B*T*P, 3C]
 * (q|k|v per row, 8 heads), softmax over the T frames of each pixel -> out [B*T*P, C]. */
int edv_attn_temporal(const float *qkv_dev, float *out_dev, int32_t B, int32_t T, int32_t P, int32_t C, int32_t heads, void *stream);

/* Rotary embedding of the temporal attention, attention.py:419-429: q|k of qkv [B*T*P, 3C] rotated in place, channel pairs
 * (2i, 2i+1) of a row of frame t by the angle whose (cos, sin) is table[t, i, :] (table [>=T, C/2, 2], attention.py:402-408).
 * transpose != 0 applies the adjoint (the gradient of the rotation with respect to its input). */
int edv_rope_qk(float *qkv_dev, const float *table_dev, int32_t B, int32_t T, int32_t P, int32_t C, int32_t transpose, void *stream);

/* GroupNorm(32 groups) on channels-last x [F,P,C] (motion_module.py:84,110).  With a workspace (edv_groupnorm_workspace bytes) the
 * statistics are taken in two coalesced stages (per 32-pixel chunk and channel, then merged per group with the parallel-variance
 * formula); with workspace_dev = NULL one workgroup per (frame, group) reads its strided slab twice.  Same values to fp32 rounding. */
size_t edv_groupnorm_workspace(int32_t F, int32_t P, int32_t C);
int edv_groupnorm(const float *x_dev, const float *w_dev, const float *b_dev, float *y_dev, float *stats_dev /* [F*32*2] scratch */,
                  int32_t F, int32_t P, int32_t C, int32_t groups, float eps, float *workspace_dev, size_t workspace_bytes, void *stream);

/* GEGLU, attention.py:363-384: y[m, j] = x[m, j] * gelu(x[m, inner + j]), x [M, 2*inner]. */
int edv_geglu(const float *x_dev, float *y_dev, int64_t M, int32_t inner, void *stream);

/* Bilinear resize, align_corners=True, channels-last [F,H,W,C] -> [F,OH,OW,C]
 * (every F.interpolate of the model). */
/* Test hook: fills the whole LDS (160 KB) of every CU with `value`; LDS keeps its contents between kernels, so a kernel that depends on LDS bytes it
 * never wrote shows.  No reference counterpart. */
int edv_debug_fill_lds(float value, void *stream);

int edv_bilinear(const float *x_dev, float *y_dev, int32_t F, int32_t H, int32_t W, int32_t C, int32_t OH, int32_t OW, void *stream);

/* y[m] = act(x[m,:]·w + b) for the final 1x1 convs; act: 0 none, 2 ReLU, 3 sigmoid, 4 sigmoid(-v). */
int edv_dot_channels(const float *x_dev, const float *w_dev, const float *b_dev, float *y_dev, int64_t M, int32_t C, int32_t act, void *stream);

/* endodav.py:153-155 + patch_embed.py:75-77 staging: bilinear resize (align_corners) of
 * x [F,3,H,W] to [ih,iw], ImageNet normalisation, and im2col into rows
 * [F*(ih/14)*(iw/14), 588] ordered (c, ky, kx) like the Conv2d weight. */
int edv_patchify(const float *x_dev, float *cols_dev, int32_t F, int32_t H, int32_t W, int32_t ih, int32_t iw, void *stream);

/* ConvTranspose2d with kernel == stride == s (dpt.py:71-82), channels-last x [F,h,w,C] -> y [F,h*s,w*s,C];
 * w [C,C,s,s] and b [C] in torch layout; wpack [s*s*C*C] and bpack [s*s*C] are caller scratch. */
int edv_conv_transpose(const float *x_dev, const float *w_dev, const float *b_dev, float *wpack_dev, float *bpack_dev, float *y_dev, int32_t F, int32_t h,
              int32_t w, int32_t C, int32_t s, void *stream);

/* interpolate_pos_encoding's bicubic resample (vision_transformer.py:186-217): grid [S,S,D] -> out [oh,ow,D]
 * with scale factors (oh+0.1)/S, (ow+0.1)/S exactly as F.interpolate(scale_factor=...) receives them. */
int edv_bicubic_pos(const float *grid_dev, float *out_dev, int32_t S, int32_t D, int32_t oh, int32_t ow, double scale_h, double scale_w, void *stream);

/* Bicubic resize of `planes` images [H,W] -> [OH,OW] (Keys cubic a=-0.75, half-pixel centres, clamped
 * borders = cv2.INTER_CUBIC): the frame pre-resize of infer_video_depth (endodav.py:170-181,196). */
int edv_resize_bicubic(const float *x_dev, float *y_dev, int32_t planes, int32_t H, int32_t W, int32_t OH, int32_t OW, void *stream);

/* out = W + scale * (B∘V)(A∘U)  (U, V may be NULL): the LoRA / DV-LoRA fold of mylora/layers.py:148-157,384-393. */
int edv_fold_lora(const float *W_dev, const float *A_dev, const float *B_dev, const float *U_dev, const float *V_dev, float scale, float *out_dev,
                  int32_t nout, int32_t nin, int32_t r, void *stream);

/* ---- fine-tune step (trainer_end_to_end_video.py:731 forward, :427-431 backward/step) ----
 * edv_set_train(ctx, 1): edv_forward keeps the activations the backward needs (per encoder block: both residual-stream
 * values, the normed MLP input, q|k|v, the attention output and its log-sum-exp, the fc1 pre-activation; per motion module
 * and fusion block: the inputs of their norms and ReLUs; per output head its intermediates).  Every constructor option trains
 * (both output heads, lora_type none / lora / dvlora / ssb / dash in both of its phases, temporal_lora, use_clstoken, residual
 * blocks, inv_sigmoid, out_sigmoid, pe rope) except use_bn, whose train-mode BatchNorm (batch statistics) is not built: refused.
 * ONE set of activations per context: each training forward overwrites the previous one's and advances the context's generation
 * counter (edv_generation, read it right after the forward).
 * edv_backward(ctx, generation, disp0, grads): generation = the forward being differentiated (0 = do not check); a mismatch with
 * the kept activations is an error, never a silently wrong gradient.  disp0 = the ("disp", 0) map that forward wrote,
 * grads[k] = dL/d("disp", k), k = 0..3 (all four required, contiguous fp32).  Produces the gradients edv_set_grad_scope selects,
 * named like the state_dict: the LoRA factors of mlp.fc1 / mlp.fc2 ("pretrained.blocks.<i>.mlp.fc<j>.lora_{A,B,U,V,index}"), with
 * temporal_lora those of ff.net.2 in the motion modules, the output-head convolutions, the residual blocks -- the trainable sets
 * of endodav/layers.py:5-34.  Where they land: a name listed in the caller's flat buffer (edv_grad_bind_flat) is written into its
 * slice there; any other into context-owned memory, read with edv_grad / edv_grad_copy.
 * After the optimizer step call edv_refresh_lora (or edv_prepare) before the next forward. */
int edv_set_train(edv_ctx *ctx, int32_t on);
/* Which gradients the next edv_backward has to produce.  The trainer alternates spatial and temporal tuning phases
 * (trainer_end_to_end_video.py:327-339); with encoder_factors = 0 the backward stops at the head (nothing below it is
 * trainable), with temporal_factors = 0 the ff.net.2 products are skipped.  head_convs != 0 adds the weight and bias gradients of
 * the output-head convolutions, named like the state_dict: "head.conv_depth_<k>.head.{0,2,4}.{weight,bias}" with the conv head
 * (trainable by default, endodav/layers.py:5-34), "head.scratch.output_conv1.*", "head.scratch.output_conv2.{0,2}.*" with the VDA
 * head (--train_output_conv).  residual_blocks != 0 adds every parameter of the ResBottleneckBlocks
 * ("pretrained.blocks.<i>.residual_.{conv1,conv2,conv3}.weight", ".norm{1,2,3}.{weight,bias}"; trainable by default in the reference,
 * block.py:146-150).  Default: both factor sets, nothing else. */
int edv_set_grad_scope(edv_ctx *ctx, int32_t encoder_factors, int32_t temporal_factors, int32_t head_convs, int32_t residual_blocks);
int edv_generation(const edv_ctx *ctx, uint64_t *generation);
int edv_backward(edv_ctx *ctx, uint64_t generation, const float *disp0_dev, const float *const grad_disp_dev[4], void *stream);
/* One contiguous gradient buffer owned by the caller, for the data-parallel step (the reference's nn.DataParallel reduce,
 * trainer_end_to_end_video.py:269-271; here ONE all-reduce over this buffer, in place).  names[i] / numels[i], i < n: the trainable
 * tensors in the caller's order.  offsets_out[0..n] receives each slice's offset in floats (slices start on 16-byte boundaries) and,
 * at [n], the total length.  flat_dev = NULL only computes the layout; otherwise flat_dev[flat_floats] must hold that total, and
 * from then on edv_backward writes the gradient of names[i] at flat_dev + offsets_out[i] -- the host's .grad tensors are views of
 * it, nothing is copied per tensor.  edv_backward fails if it produced no gradient for a listed name.  n = 0 unbinds. */
int edv_grad_bind_flat(edv_ctx *ctx, int32_t n, const char *const *names, const int64_t *numels, float *flat_dev, int64_t flat_floats,
                       int64_t *offsets_out);
int edv_grad(edv_ctx *ctx, const char *name, float **grad_dev, int64_t *numel);
int edv_grad_copy(edv_ctx *ctx, const char *name, float *dst_dev, int64_t numel, void *stream); /* stream-ordered copy into caller memory */

/* ---- the fine-tune step's loss ----
 * The trainer's photometric loss on the four disparity maps of B clips of T frames, and its gradient with respect to them, in one
 * call: per scale the map is resized to the frame size (bilinear, align_corners; trainer_end_to_end_video.py:813-817), turned into
 * depth (utils/layers.py:11-20), back-projected and projected into the previous / next frame of the clip (utils/layers.py:134-189),
 * the neighbour is sampled there (F.grid_sample, border padding, align_corners; trainer :853-857) and compared with the frame by
 * 0.85 SSIM + 0.15 L1 (trainer :899-911, utils/layers.py:276-306), plus disparity_smoothness / 2^s times the edge-aware smoothness
 * of the mean-normalised map (utils/layers.py:222-236, trainer :944-946); mean over the four scales (trainer :968).  The pose
 * network's outputs are inputs here: K, invK, Tprev, Tnext are [B*T, 4, 4] row-major (Tprev / Tnext: pose from frame i to frames
 * i-1 / i+1; the first / last frame of a clip has no such neighbour and is left out of that term).  frames [B*T, 3, H, W] in [0, 1];
 * disp_dev[s] [B*T, 1, disp_h[s], disp_w[s]]; loss_dev receives ONE float; grad_disp_dev[s] receives dL/d disp_dev[s] (same shape).
 * Deterministic (gathers and fixed-order two-stage sums, no float atomics).  T >= 2, H, W >= 3. */
size_t edv_photometric_loss_workspace(int32_t B, int32_t T, int32_t H, int32_t W); /* bytes */
int edv_photometric_loss(const float *frames_dev, const float *const disp_dev[4], const int32_t disp_h[4], const int32_t disp_w[4], int32_t B, int32_t T, int32_t H,
                         int32_t W, const float *K_dev, const float *invK_dev, const float *Tprev_dev, const float *Tnext_dev, float min_depth, float max_depth,
                         float disparity_smoothness, float *loss_dev, float *const grad_disp_dev[4], float *workspace_dev, size_t workspace_bytes, void *stream);

/* ---- the trainer's whole loss (round 3): generate_images_pred + compute_losses of trainer_end_to_end_video.py:808-971 with the side networks'
 * outputs as inputs, and every gradient the reference's autograd reaches.  N = B*T flattened frames (trainer :406-409), frame size H x W.
 *   color[s]            inputs[("color", 0, s)]            [N, 3, H >> s, W >> s]   (s = 0: the frame)
 *   color_nb[0 / 1]     inputs[("color", -1 / +1, 0)]      [N, 3, H, W]
 *   K, invK             [N, 4, 4] row-major (inputs[("K", 0)] or, with learn_intrinsics, outputs[("K", 0)]);  T[0 / 1] = outputs[("cam_T_cam", 0, -1 / +1)]
 *   refined[s][n], registration[s][n], transform[s][n] = outputs[("refined" | "registration", s, fid)], outputs[("transform", "high", s, fid)]   [N, 3, H, W]
 *   mask[n]             outputs[("occu_mask_backward", 0, fid)]   [N, 1, H, W];   position[s][n] = outputs[("position", "high", s, fid)] [N, 2, H, W]
 *                       (position may be NULL when depth_flow is 0 or tune_temporal is off)
 *   disp[s]             outputs[("disp", s)]   [N, 1, disp_h[s], disp_w[s]]
 * losses_dev receives 29 floats: for s = 0..3 {loss/s, loss_reprojection, loss_transform, loss_cvt, loss_smooth, loss_depth_reproj, loss_depth_flow}
 * (the trainer's per-scale entries, :960-966), then losses["loss"] (:968).  Gradients of losses["loss"]: grads->disp[s] are required; refined /
 * transform / K / invK / T are optional (NULL = not wanted); registration and the mask are detached in the reference.  Deterministic except for the
 * gradient the two depth-consistency terms scatter into the sampled depth map (float atomics, as ATen's grid_sampler backward); with the options'
 * defaults (depth_reproj = depth_flow = 0) no atomic runs.  H, W >= 16. */
typedef struct {
    const float *color[4];
    const float *color_nb[2];
    const float *K, *invK;
    const float *T[2];
    const float *refined[4][2], *registration[4][2], *transform[4][2];
    const float *mask[2];
    const float *position[4][2];
    const float *disp[4];
    int32_t disp_h[4], disp_w[4];
} edv_trainer_loss_inputs;
typedef struct {
    float disparity_smoothness, transform_constraint, transform_smoothness, depth_reproj, depth_flow; /* options.py:136-159 */
    int32_t tune_temporal;                                                                              /* trainer :951 temporal_weight */
    float min_depth, max_depth;
} edv_trainer_loss_weights;
typedef struct {
    float *disp[4];
    float *refined[4][2], *transform[4][2];
    float *K, *invK, *T[2];
} edv_trainer_loss_grads;
size_t edv_trainer_loss_workspace(int32_t N, int32_t H, int32_t W); /* bytes */
int edv_trainer_loss(const edv_trainer_loss_inputs *in, int32_t N, int32_t H, int32_t W, const edv_trainer_loss_weights *weights, float *losses_dev,
                     const edv_trainer_loss_grads *grads, float *workspace_dev, size_t workspace_bytes, void *stream);

/* ---- backward kernels (input gradients of the frozen operators, gradients of the LoRA factors): input gradients of the frozen operators and the gradients of the
 * LoRA factors, the only trainable tensors (endodav/layers.py:5-34).  Same layouts as the forward kernels. ---- */
/* LayerNorm (no affine gradient): dx (+)= dLN(x; w, eps)(dy);  rows x dim, dim % 4 == 0, dim <= 1024. */
int edv_layernorm_bwd(const float *x_dev, const float *w_dev, const float *dy_dev, float *dx_dev, int64_t rows, int32_t dim, float eps,
                      int32_t accumulate, void *stream);
/* out = f(d) + (add ? add : 0); mode 0: f = d; 1: d * gelu'(src) (exact erf form); 2: src > 0 ? d : 0.  n % 4 == 0. */
int edv_ew_bwd(const float *d_dev, const float *src_dev, const float *add_dev, float *out_dev, int64_t n, int32_t mode, void *stream);
/* GEGLU (attention.py:363-384): x [M, 2*inner] = [a | g], y = a * gelu(g); dx from dy [M, inner]. */
int edv_geglu_bwd(const float *x_dev, const float *dy_dev, float *dx_dev, int64_t M, int32_t inner, void *stream);
/* Wt[k, n] = W[n, k] * gamma[n] (gamma may be NULL): the NT-form weight of dX = (dY * gamma) W. */
int edv_transpose_scale(const float *W_dev, const float *gamma_dev, float *Wt_dev, int32_t N, int32_t K, void *stream);
/* LoRA / DV-LoRA factor gradients of y = gamma * (x (W + s (B*V)(A*U))^T + b) from x [M, nin] and G = dL/dy [M, nout]
 * (mylora/layers.py:148-157, 384-393): A [r, nin], B [nout, r], U [r, 1], V [nout, 1].  U/V NULL = plain LoRA; gamma may be
 * NULL; any output may be NULL.  r in {1,2,4,8}. */
size_t edv_lora_grads_workspace(int64_t M, int32_t nin, int32_t nout, int32_t r); /* bytes */
int edv_lora_grads(const float *x_dev, const float *g_dev, int64_t M, int32_t nin, int32_t nout, int32_t r, const float *A_dev, const float *B_dev,
                   const float *U_dev, const float *V_dev, float s, const float *gamma_dev, float *workspace_dev, size_t workspace_bytes, float *dA_dev,
                   float *dB_dev, float *dU_dev, float *dV_dev, void *stream);
/* bilinear align_corners=True, input gradient: dy [F,oh,ow,C] -> dx [F,ih,iw,C] (deterministic gather). */
int edv_bilinear_bwd(const float *dy_dev, float *dx_dev, int32_t F, int32_t ih, int32_t iw, int32_t C, int32_t oh, int32_t ow, int32_t accumulate,
                     void *stream);
/* final 1x1 conv to one channel + output activation on a post-ReLU input (dpt.py:121-123; endodav/layers.py:206-221 with the sigmoid
 * of dpt_pyramid.py:103-109):  gz[p] = dL/dz;  mode 0 (ReLU): disp > 0 ? g : 0;  mode 1 / 2 (sigmoid(z) / sigmoid(-z)): +-g disp (1 - disp);
 * d_o2[p,c] = gz[p] * w[c] * (o2[p,c] > 0).  gz_out_dev (optional, [npix]) keeps gz for the 1x1 conv's own weight / bias gradient. */
int edv_dot_channels_bwd(const float *g_dev, const float *disp_dev, const float *w_dev, const float *o2_dev, float *d_o2_dev, float *gz_out_dev,
                         int64_t npix, int32_t C, int32_t mode, void *stream);
/* Weight gradient of a 3x3 / stride 1 / padding 1 convolution (the trainable convolutions of the output heads, endodav/layers.py:5-34):
 * x [F,H,W,Cin] channels-last, dy [F,H,W,Cout] -> dw [Cout,Cin,3,3] (torch layout).   Deterministic (two stages). */
size_t edv_conv3x3_wgrad_workspace(int32_t F, int32_t H, int32_t W, int32_t Cin, int32_t Cout); /* bytes */
int edv_conv3x3_wgrad(const float *x_dev, const float *dy_dev, float *dw_dev, int32_t F, int32_t H, int32_t W, int32_t Cin, int32_t Cout,
                      float *workspace_dev, size_t workspace_bytes, int32_t accumulate, void *stream);
/* out[n] (+)= sum_m rowscale[m] * P[m,n] (rowscale may be NULL): bias gradients, and the 1x1 head's weight gradient with rowscale = gz.
 * N a power of two in 4..1024, or 1 (then M % 4 == 0 and no rowscale). */
size_t edv_colsum_workspace(int32_t N); /* bytes */
int edv_colsum_rows(const float *P_dev, const float *rowscale_dev, int64_t M, int32_t N, float *workspace_dev, size_t workspace_bytes, float *out_dev,
                    int32_t accumulate, void *stream);
/* GroupNorm input gradient; stats = the forward's [F, groups, 2] (mean, rstd); sums = [F, groups, 2] scratch. */
int edv_groupnorm_bwd(const float *x_dev, const float *stats_dev, const float *w_dev, const float *dy_dev, float *sums_dev, float *dx_dev, int32_t F,
                      int32_t P, int32_t C, int32_t groups, int32_t accumulate, void *stream);
/* temporal attention core, backward: dqkv [B*T*P, 3C] from qkv and dout [B*T*P, C]. */
int edv_attn_temporal_bwd(const float *qkv_dev, const float *dout_dev, float *dqkv_dev, int32_t B, int32_t T, int32_t P, int32_t C, int32_t heads,
                          void *stream);
/* 3x3 convolution input gradient.  Stride 1: run edv_conv3x3 on dy with the weight repacked by edv_pack_conv3x3_bwd
 * ([Cout,Cin,3,3] -> [Cin][3][3][Cout], taps flipped).  Stride 2: direct kernel on the forward's packed weight. */
int edv_pack_conv3x3_bwd(const float *w_dev, float *wpacked_dev, int32_t Cout, int32_t Cin, void *stream);
int edv_conv3x3_s2_bwd(const float *dy_dev, const float *wpacked_dev, float *dx_dev, int32_t F, int32_t H, int32_t W, int32_t Cin, int32_t Cout,
                       void *stream);
/* The engine's form of the stride-2 input gradient: z [F,H,W,C] = dy [F,OH,OW,C] with zeros inserted (z[2oy,2ox] = dy[oy,ox]),
 * then the stride-1 recipe above on z. */
int edv_dilate2(const float *dy_dev, float *z_dev, int32_t F, int32_t H, int32_t W, int32_t C, void *stream);
/* ConvTranspose(k = s) input gradient = edv_gemm of the pixel-unshuffled dy [F*h*w, s*s*C] with the transposed packed weight. */
int edv_pixel_unshuffle(const float *dy_dev, float *A_dev, int32_t F, int32_t h, int32_t w, int32_t C, int32_t s, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* ENDODAV_HIP_H */
