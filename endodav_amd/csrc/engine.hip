// The forward engine: edv_create / edv_bind_param / edv_prepare / edv_forward.
//
// Sequences EndoDAV's per-clip forward (reference models/endodav/endodav.py:150-160) as ~25 kernel
// launches per encoder block + ~120 for the DPT head on one HIP stream.  No host sync inside a
// forward once the workspace for a clip geometry exists.  Layouts: encoder activations are
// tokens-major [frames*tokens, D]; head activations channels-last [frames, h, w, C], so that
//   - the 1x1 "projects" convs, proj_in/out and every Linear are plain GEMMs on the same buffers,
//   - the five NCHW<->NLC permutes per motion module (motion_module.py:105,112,121,124,232,295)
//     and the tap permute (dpt_pyramid.py:61) vanish,
//   - temporal attention reaches the frame axis by a constant address stride.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/endodav_hip.h"
#include "ops.hpp"

namespace edv {
static thread_local std::string g_err;
thread_local LaunchTimer *g_launch_timer = nullptr;
void set_error(const std::string &m) { g_err = m; }
const char *get_error() { return g_err.c_str(); }
}  // namespace edv

using namespace edv;

struct Param {
    const float *p;
    std::vector<int64_t> shape;
    long long numel() const {
        long long n = 1;
        for (auto s : shape) n *= s;
        return n;
    }
};
struct Buf {
    float *p = nullptr;
    size_t cap = 0;  // floats
};

// kernel classes for the optional HIP-event bracketing (edv_profile_enable / edv_profile_read)
// KC_LINEAR_ENC: the F.linear launches of the encoder blocks (qkv, proj, fc1, fc2: 96 % of the dense-GEMM work), a sub-class bracketed
// with the same mask bit as KC_LINEAR and reported separately (the head's small GEMMs are HBM- and launch-bound, not MFMA-bound)
constexpr int PE_K = 608;  // patch-embed im2col width 3 * 14 * 14 = 588, padded to a multiple of 32

// KC_GROUPNORM .. KC_PATCHIFY: the HBM-bound kernels of the forward, each with its algorithmic bytes (tensor in + tensor out, once) for
// bench.py's roofline_hbm object
enum { KC_LINEAR = 0, KC_CONV3 = 1, KC_ATTN_SPATIAL = 2, KC_ATTN_TEMPORAL = 3, KC_NORM = 4, KC_OTHER = 5, KC_LINEAR_ENC = 6, KC_GROUPNORM = 7,
       KC_BILINEAR = 8, KC_GEGLU = 9, KC_DOT = 10, KC_PATCHIFY = 11, KC_ATTN_SPATIAL_BWD = 12, KC_COUNT = 13 };
struct EvPool {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
    size_t used = 0;
};

struct edv_ctx {
    edv_config cfg{};
    unsigned prof_mask = 0;
    EvPool prof[KC_COUNT];
    double prof_flops[KC_COUNT] = {};  // algorithmic work of the bracketed launches (edv_profile_work)
    double prof_bytes[KC_COUNT] = {};
    int enc_streams = 0;                      // 0: automatic (2 for small clips); n >= 1: that many frame groups on internal streams
    int enc_streams_initial = 0;              // what EDV_ENC_STREAMS asked for at edv_create (edv_set_encoder_streams(-1) restores it)
    hipStream_t sub[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_x[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // cross-stream edges of the head (r3, r1r2, u3, u2, u1)
    std::unordered_map<std::string, Param> params;
    std::unordered_map<std::string, Buf> packed;  // derived weights, owned
    std::unordered_map<std::string, Buf> ws;      // activations, owned
    bool prepared = false;
    int products = EDV_PRODUCTS_F32;  // arithmetic of the encoder's linears in inference (edv_set_products)
    std::unordered_map<const float *, const void *> x6;  // effective weight of an encoder linear -> its bf16 planes (gemm_x6.hip), owned by `packed`
    const float *skws_zeroed = nullptr;  // stream-K workspace whose arrival counters have been zeroed (gemm_dma.hip)
    bool capture = false;
    bool train = false;           // forward keeps the activations the backward needs (edv_set_train)
    bool train_prepared = false;  // transposed / flipped weights of the input-gradient GEMMs are current
    bool have_saved = false;      // a training forward has run since the last backward
    bool grad_encoder = true;     // which factor gradients the caller wants (edv_set_grad_scope): the trainer alternates
    bool grad_temporal = true;    // spatial and temporal tuning phases (trainer_end_to_end_video.py:327-339)
    bool grad_res = false;        // parameters of the residual bottleneck blocks (residual_*, trainable by default in the reference)
    bool grad_head = false;       // weight / bias gradients of the output-head convolutions (conv_depth_*, or scratch.output_conv* with --train_output_conv)
    std::unordered_map<std::string, Buf> grads;  // gradients of the trainable parameters, owned
    // Caller-owned flat gradient buffer (edv_grad_bind_flat): a gradient whose name is listed here is written straight into its slice
    // of that buffer instead of into `grads` -- the host's .grad tensors are views of it and the data-parallel all-reduce runs on
    // it in place (trainer_end_to_end_video.py:269-271's reduce), with no per-tensor copy on either side.
    struct FlatSlot {
        float *p;
        size_t numel;
        bool written;
    };
    std::unordered_map<std::string, FlatSlot> flat;
    int device = 0;                 // HIP device the context was created on (edv_destroy frees there)
    uint64_t generation = 0;        // counts training forwards; the kept activations belong to forward number `saved_generation`
    uint64_t saved_generation = 0;
    int launches = 0;
    size_t bytes = 0;
    // geometry of the last forward (for edv_stage_copy)
    int F = 0, T = 0, ph = 0, pw = 0, ntok = 0;
    std::unordered_map<std::string, std::pair<const float *, size_t>> stages;
};

namespace {

int alloc_buf(edv_ctx *c, std::unordered_map<std::string, Buf> &pool, const std::string &name, size_t n, hipStream_t st, float **out) {
    Buf &b = pool[name];
    if (b.cap < n) {
        if (b.p) {
            EDV_HIP(hipStreamSynchronize(st));  // kernels in flight may still read the old block
            EDV_HIP(hipFree(b.p));
            c->bytes -= b.cap * sizeof(float);
            b.p = nullptr;
            b.cap = 0;
        }
        void *p = nullptr;
        EDV_HIP(hipMalloc(&p, n * sizeof(float)));
        b.p = (float *)p;
        b.cap = n;
        c->bytes += n * sizeof(float);
    }
    *out = b.p;
    return 0;
}

// Times the launch(es) made while it is alive when their class is being profiled: the event pair travels inside the dispatches (EDV_LAUNCH,
// common.hpp), so the pair measures the kernels alone, as rocprofv3's kernel trace does.  Brackets nest like scopes (an inner one times its own launches).
struct Bracket {
    LaunchTimer timer;
    LaunchTimer *prev = nullptr;
    bool on = false;
    hipStream_t st;
    Bracket(edv_ctx *c, int cls, hipStream_t s) : st(s) {
        if (!(c->prof_mask & (1u << (cls == KC_LINEAR_ENC ? KC_LINEAR : cls)))) return;  // the sub-class shares its parent's mask bit
        EvPool &p = c->prof[cls];
        if (p.used == p.ev.size()) {
            hipEvent_t a, b;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
            p.ev.emplace_back(a, b);
        }
        auto &pr = p.ev[p.used++];
        timer.start = pr.first;
        timer.stop = pr.second;
        prev = g_launch_timer;
        g_launch_timer = &timer;
        on = true;
    }
    ~Bracket() {
        if (!on) return;
        g_launch_timer = prev;
        if (!timer.started) {  // nothing was launched inside: keep the pair well-formed (zero-length interval on the stream)
            (void)hipEventRecord(timer.start, st);
            (void)hipEventRecord(timer.stop, st);
        }
    }
};

// Bracket for a bandwidth-bound launch: also books its algorithmic bytes (every tensor it reads or writes, once)
struct HbmScope {
    Bracket b;
    HbmScope(edv_ctx *c, int cls, hipStream_t st, double bytes) : b(c, cls, st) {
        if (c->prof_mask & (1u << cls)) c->prof_bytes[cls] += bytes;
    }
};

// EDV_X6_ATTN=0: in the BF16X6 mode only the linears change, the attention stays on the fp32 kernel (A/B runs)
inline bool attn_x6_on() {
    static const bool on = [] {
        const char *e = getenv("EDV_X6_ATTN");
        return !(e && atoi(e) == 0);
    }();
    return on;
}

struct Run {
    edv_ctx *c;
    hipStream_t st;
    const edv_config &cfg;
    int D, depth, heads, Fe;
    int F = 0, B = 0, T = 0, ph = 0, pw = 0, P0 = 0, ntok = 0, c0 = 0;
    std::string rb_suffix;

    Run(edv_ctx *ctx, hipStream_t s) : c(ctx), st(s), cfg(ctx->cfg) {
        D = cfg.embed_dim;
        depth = cfg.depth;
        heads = cfg.num_heads;
        Fe = cfg.features;
    }

    // ---- lookup helpers -------------------------------------------------------------------
    int param(const std::string &name, const float **out, int ndim_expect = -1) {
        auto it = c->params.find(name);
        EDV_CHECK(it != c->params.end(), "parameter not bound: " + name);
        if (ndim_expect >= 0) EDV_CHECK((int)it->second.shape.size() == ndim_expect, "unexpected rank for " + name);
        *out = it->second.p;
        return 0;
    }
    bool has(const std::string &name) const { return c->params.count(name) != 0; }
    int packedw(const std::string &name, const float **out) {
        auto it = c->packed.find(name);
        EDV_CHECK(it != c->packed.end() && it->second.p, "packed weight missing (edv_prepare not run?): " + name);
        *out = it->second.p;
        return 0;
    }
    int wsbuf(const std::string &name, size_t n, float **out) { return alloc_buf(c, c->ws, name, n, st, out); }
    int pk(const std::string &name, size_t n, float **out) { return alloc_buf(c, c->packed, name, n, st, out); }
    // bias of a ResidualConvUnit convolution: with use_bn the one edv_prepare folded the BatchNorm into
    int rcu_bias(const std::string &conv, const float **out) {
        if (cfg.use_bn) return packedw(conv + ".bias", out);
        return param(conv + ".bias", out);
    }

    // ---- op wrappers ----------------------------------------------------------------------
    int enc_F = 0, enc_f0 = 0;    // all frames of the clip / first frame of the group encoder_range is working on (training buffers hold all frames)
    bool in_encoder = false;      // linear() is being called from the encoder block loop (profiling sub-class KC_LINEAR_ENC)
    bool stagger_record = false;  // encoder_range records ev_x[5] after block 0's qkv GEMM (start signal for the next frame group)
    float *skws = nullptr;  // stream-K split workspace of the stream this Run is enqueueing on
    size_t skws_floats = 0;
    int gemm_ws(GemmDesc &g) {
        g.ws = skws;
        g.ws_floats = skws_floats;
        return gemm(g, st);
    }
    int linear(const float *A, long long M, int K, const float *W, int N, const float *bias, float *C, int act = ACT_NONE,
               const float *gamma = nullptr, const float *R1 = nullptr) {
        GemmDesc g;
        g.A = A; g.lda = K; g.W = W; g.ldw = K; g.C = C; g.ldc = N; g.M = M; g.N = N; g.K = K;
        g.bias = bias; g.act = act; g.gamma = gamma; g.R1 = R1; g.ldr1 = N;
        if (in_encoder && !c->train && c->products == EDV_PRODUCTS_BF16X6) {
            auto it = c->x6.find(W);
            if (it != c->x6.end()) g.Wx6 = it->second;
        }
        c->launches++;
        const int cls = in_encoder ? KC_LINEAR_ENC : KC_LINEAR;
        if (c->prof_mask & (1u << KC_LINEAR)) {  // 2 M N K; A, W read once, C written once (+ the residual read)
            c->prof_flops[cls] += 2.0 * (double)M * N * K;
            c->prof_bytes[cls] += 4.0 * ((double)M * K + (double)N * K + (double)M * N * (R1 ? 2 : 1));
        }
        Bracket b_(c, cls, st);
        return gemm_ws(g);
    }
    int conv3(const float *x, int H, int W, int Cin, const float *wp, const float *bias, int Cout, int stride, float *y, bool pre_relu,
              int act = ACT_NONE, const float *R1 = nullptr, const float *R2 = nullptr) {
        GemmDesc g;
        const int OH = (H - 1) / stride + 1, OW = (W - 1) / stride + 1;
        g.A = x; g.W = wp; g.ldw = 9 * Cin; g.C = y; g.ldc = Cout; g.M = (long long)F * OH * OW; g.N = Cout; g.K = 9 * Cin;
        g.bias = bias; g.act = act; g.R1 = R1; g.ldr1 = Cout; g.R2 = R2; g.ldr2 = Cout;
        g.loader = LOAD_CONV3; g.cH = H; g.cW = W; g.cC = Cin; g.cOH = OH; g.cOW = OW; g.cS = stride; g.pre_relu = pre_relu ? 1 : 0;
        c->launches++;
        if (c->prof_mask & (1u << KC_CONV3)) {  // 2 M N K; input and output tensors once, the packed weight once
            c->prof_flops[KC_CONV3] += 2.0 * (double)g.M * g.N * g.K;
            c->prof_bytes[KC_CONV3] += 4.0 * ((double)F * H * W * Cin + (double)g.N * g.K + (double)g.M * g.N * (1 + (R1 ? 1 : 0) + (R2 ? 1 : 0)));
        }
        Bracket b_(c, KC_CONV3, st);
        return gemm_ws(g);
    }
    int ln(const float *x, RowMap im, const std::string &prefix, float *y, long long rows, int dim, float eps, const float *pe = nullptr,
           int rpf = 0, int TT = 0) {
        const float *w, *b;
        EDV_TRY(param(prefix + ".weight", &w));
        EDV_TRY(param(prefix + ".bias", &b));
        c->launches++;
        HbmScope b_(c, KC_NORM, st, 8.0 * (double)rows * dim);
        return layernorm(x, im, w, b, y, identity_map(), rows, dim, eps, pe, rpf, TT, st);
    }

    // ---- weight packing (edv_prepare) -----------------------------------------------------
    int fold_linear(const std::string &p, bool lora_here) {
        // result registered under packed[p + ".weight"]; a plain pointer alias when no LoRA applies
        const float *W;
        EDV_TRY(param(p + ".weight", &W, 2));
        const Param &pw_ = c->params[p + ".weight"];
        const int nout = (int)pw_.shape[0], nin = (int)pw_.shape[1];
        if (!lora_here || cfg.lora_type == EDV_LORA_NONE || !has(p + ".lora_A")) return 0;
        float *out;
        EDV_TRY(pk(p + ".weight", (size_t)nout * nin, &out));
        const float *A, *Bm;
        EDV_TRY(param(p + ".lora_A", &A));
        EDV_TRY(param(p + ".lora_B", &Bm));
        const int r = cfg.lora_rank;
        switch (cfg.lora_type) {
            case EDV_LORA_LORA:  // lora_alpha = 2r  (endodav.py:111-112)
                return fold_lora(W, A, Bm, nullptr, nullptr, 2.0f, out, nout, nin, r, st);
            case EDV_LORA_DVLORA: {  // lora_alpha = r  (endodav.py:108-109)
                const float *U, *V;
                EDV_TRY(param(p + ".lora_U", &U));
                EDV_TRY(param(p + ".lora_V", &V));
                return fold_lora(W, A, Bm, U, V, 1.0f, out, nout, nin, r, st);
            }
            case EDV_LORA_SSB:
                return fold_ssb(W, A, Bm, out, nout, nin, st);
            case EDV_LORA_DASH: {
                EDV_TRY(fold_lora(W, A, Bm, nullptr, nullptr, 2.0f, out, nout, nin, r, st));
                if (cfg.dash_active) {
                    const float *Ut, *idx, *Vt;
                    EDV_TRY(param(p + ".weight_u_top", &Ut));
                    EDV_TRY(param(p + ".lora_index", &idx));
                    EDV_TRY(param(p + ".weight_vt_top", &Vt));
                    const int ri = (int)c->params[p + ".lora_index"].shape[0];
                    return fold_dash(Ut, idx, Vt, out, nout, nin, ri, st);
                }
                return 0;
            }
            default:
                EDV_CHECK(false, "unknown lora_type");
        }
        return 0;
    }
    // effective weight of a (possibly folded) linear
    int lin_w(const std::string &p, const float **out) {
        auto it = c->packed.find(p + ".weight");
        if (it != c->packed.end() && it->second.p) {
            *out = it->second.p;
            return 0;
        }
        return param(p + ".weight", out);
    }
    // bf16 planes of an encoder linear's effective weight (after the LoRA fold) for gemm_x6.hip
    int make_x6(const std::string &p) {
        const float *W;
        EDV_TRY(lin_w(p, &W));
        const Param &q = c->params[p + ".weight"];
        const int nout = (int)q.shape[0], nin = (int)q.shape[1];
        if (nin % 16 != 0 || nout < 64) return 0;
        float *pl;
        EDV_TRY(pk(p + ".x6", (gemm_x6_planes_bytes(nout, nin) + 3) / 4, &pl));
        EDV_TRY(gemm_x6_split(W, pl, nout, nin, st));
        c->x6[W] = pl;
        return 0;
    }
    int build_x6(bool mlp_only) {
        if (c->products != EDV_PRODUCTS_BF16X6) return 0;
        for (int i = 0; i < depth; ++i) {
            const std::string bp = "pretrained.blocks." + std::to_string(i);
            if (!mlp_only) {
                EDV_TRY(make_x6(bp + ".attn.qkv"));
                EDV_TRY(make_x6(bp + ".attn.proj"));
            }
            EDV_TRY(make_x6(bp + ".mlp.fc1"));
            EDV_TRY(make_x6(bp + ".mlp.fc2"));
        }
        return 0;
    }
    int pack_c3(const std::string &p) {
        const float *w;
        EDV_TRY(param(p + ".weight", &w, 4));
        const Param &q = c->params[p + ".weight"];
        EDV_CHECK(q.shape[2] == 3 && q.shape[3] == 3, "expected a 3x3 kernel: " + p);
        float *out;
        EDV_TRY(pk(p + ".weight", (size_t)q.numel(), &out));
        return pack_conv3x3(w, out, (int)q.shape[0], (int)q.shape[1], st);
    }
    // eval-mode BatchNorm after convolution `conv` (util/blocks.py:80-86) folded into its packed weight and a packed bias
    int fold_bn_into(const std::string &conv, const std::string &bn) {
        const float *b, *g, *beta, *mean, *var;
        EDV_TRY(param(conv + ".bias", &b));
        EDV_TRY(param(bn + ".weight", &g));
        EDV_TRY(param(bn + ".bias", &beta));
        EDV_TRY(param(bn + ".running_mean", &mean));
        EDV_TRY(param(bn + ".running_var", &var));
        const Param &q = c->params[conv + ".weight"];
        const int nout = (int)q.shape[0], K = (int)(q.numel() / q.shape[0]);
        float *w, *bo;
        EDV_TRY(pk(conv + ".weight", (size_t)q.numel(), &w));
        EDV_TRY(pk(conv + ".bias", (size_t)nout, &bo));
        return fold_bn(w, b, g, beta, mean, var, 1e-5f, bo, nout, K, st);
    }

    int prepare() {
        c->launches = 0;
        {   // patch-embed weight [D, 3*14*14 = 588] with its rows zero-padded to PE_K = 608 = 19 x 32: the im2col GEMM then runs on the
            // LDS-DMA kernel (K % 32 == 0) instead of the register-staged one (89 -> 57 us at T=8)
            const float *w;
            float *wp;
            EDV_TRY(param("pretrained.patch_embed.proj.weight", &w, 4));
            EDV_TRY(pk("pretrained.patch_embed.proj.weight", (size_t)D * PE_K, &wp));
            EDV_HIP(hipMemsetAsync(wp, 0, (size_t)D * PE_K * sizeof(float), st));
            EDV_HIP(hipMemcpy2DAsync(wp, PE_K * sizeof(float), w, 588 * sizeof(float), 588 * sizeof(float), (size_t)D, hipMemcpyDeviceToDevice, st));
        }
        for (int i = 0; i < depth; ++i) {
            const std::string b = "pretrained.blocks." + std::to_string(i) + ".mlp.";
            EDV_TRY(fold_linear(b + "fc1", true));
            EDV_TRY(fold_linear(b + "fc2", true));
        }
        for (int i = 0; i < depth; ++i)
            if (cfg.residual_mask & (1u << i)) EDV_TRY(pack_c3("pretrained.blocks." + std::to_string(i) + ".residual_.conv2"));
        const int *oc = cfg.out_channels;
        {  // ConvTranspose k=s -> GEMM weights
            const int ss[2] = {4, 2};
            for (int j = 0; j < 2; ++j) {
                const std::string p = "head.resize_layers." + std::to_string(j);
                const float *w, *b;
                EDV_TRY(param(p + ".weight", &w, 4));
                EDV_TRY(param(p + ".bias", &b));
                float *wo, *bo;
                EDV_TRY(pk(p + ".weight", (size_t)ss[j] * ss[j] * oc[j] * oc[j], &wo));
                EDV_TRY(pk(p + ".bias", (size_t)ss[j] * ss[j] * oc[j], &bo));
                EDV_TRY(pack_convT(w, wo, b, bo, oc[j], oc[j], ss[j], st));
            }
        }
        EDV_TRY(pack_c3("head.resize_layers.3"));
        for (int j = 1; j <= 4; ++j) EDV_TRY(pack_c3("head.scratch.layer" + std::to_string(j) + "_rn"));
        for (int j = 1; j <= 4; ++j)
            for (int u = 1; u <= 2; ++u) {
                if (j == 4 && u == 1) continue;  // refinenet4.resConfUnit1 is never reached (dpt_pyramid.py:81)
                const std::string p = "head.scratch.refinenet" + std::to_string(j) + ".resConfUnit" + std::to_string(u);
                EDV_TRY(pack_c3(p + ".conv1"));
                EDV_TRY(pack_c3(p + ".conv2"));
                if (cfg.use_bn) {
                    EDV_TRY(fold_bn_into(p + ".conv1", p + ".bn1"));
                    EDV_TRY(fold_bn_into(p + ".conv2", p + ".bn2"));
                }
            }
        if (cfg.conv_head) {
            for (int k = 1; k <= 4; ++k) {
                EDV_TRY(pack_c3("head.conv_depth_" + std::to_string(k) + ".head.0"));
                EDV_TRY(pack_c3("head.conv_depth_" + std::to_string(k) + ".head.2"));
            }
        } else {
            EDV_TRY(pack_c3("head.scratch.output_conv1"));
            EDV_TRY(pack_c3("head.scratch.output_conv2.0"));
        }
        const int mmC[4] = {oc[2], oc[3], Fe, Fe};
        for (int m = 0; m < 4; ++m) {
            const std::string tb = "head.motion_modules." + std::to_string(m) + ".temporal_transformer.transformer_blocks.0";
            const int C = mmC[m];
            for (int a = 0; a < 2; ++a) {
                const std::string ab = tb + ".attention_blocks." + std::to_string(a);
                float *qkvw;
                EDV_TRY(pk(ab + ".qkv", (size_t)3 * C * C, &qkvw));
                const char *names[3] = {".to_q.weight", ".to_k.weight", ".to_v.weight"};
                for (int j = 0; j < 3; ++j) {
                    const float *w;
                    EDV_TRY(param(ab + names[j], &w, 2));
                    EDV_TRY(copy_f32(w, qkvw + (size_t)j * C * C, (long long)C * C, st));
                }
            }
            EDV_TRY(fold_linear(tb + ".ff.net.2", cfg.temporal_lora != 0));
            if ((8 * C) % 64 == 0 && C % 32 == 0) {  // interleaved copy of ff.net.0.proj for the fused GEGLU launch of the inference forward
                const float *w0, *b0;
                float *wi, *bi;
                EDV_TRY(param(tb + ".ff.net.0.proj.weight", &w0, 2));
                EDV_TRY(param(tb + ".ff.net.0.proj.bias", &b0));
                EDV_TRY(pk(tb + ".ff.net.0.geglu.w", (size_t)8 * C * C, &wi));
                EDV_TRY(pk(tb + ".ff.net.0.geglu.b", (size_t)8 * C, &bi));
                EDV_TRY(pack_geglu(w0, b0, wi, bi, 8 * C, C, st));
            }
        }
        c->x6.clear();
        EDV_TRY(build_x6(false));
        c->prepared = true;
        c->train_prepared = false;  // the folded LoRA weights changed: their transposes are stale
        return 0;
    }

    // After an optimizer step only trainable tensors changed: re-fold the linears that carry LoRA factors and re-pack the trainable
    // convolutions (HeadDepth heads or scratch.output_conv*, residual blocks) -- and, once the backward has run, their transposed /
    // flipped copies -- instead of re-packing every frozen weight as edv_prepare does.
    int refresh_lora() {
        EDV_CHECK(c->prepared, "edv_prepare has not run");
        for (int i = 0; i < depth; ++i) {
            const std::string bp = "pretrained.blocks." + std::to_string(i);
            EDV_TRY(fold_linear(bp + ".mlp.fc1", true));
            EDV_TRY(fold_linear(bp + ".mlp.fc2", true));
            if (c->train_prepared) {
                const float *g2;
                EDV_TRY(param(bp + ".ls2.gamma", &g2));
                EDV_TRY(make_t_lin(bp + ".mlp.fc1"));
                EDV_TRY(make_t_lin(bp + ".mlp.fc2", g2));
            }
            if (cfg.residual_mask & (1u << i)) {
                EDV_TRY(pack_c3(bp + ".residual_.conv2"));
                if (c->train_prepared) {
                    EDV_TRY(make_t_lin(bp + ".residual_.conv1"));
                    EDV_TRY(make_t_lin(bp + ".residual_.conv3"));
                    EDV_TRY(make_b_c3(bp + ".residual_.conv2"));
                }
            }
        }
        if (cfg.temporal_lora)
            for (int m = 0; m < 4; ++m) {
                const std::string p = "head.motion_modules." + std::to_string(m) + ".temporal_transformer.transformer_blocks.0.ff.net.2";
                EDV_TRY(fold_linear(p, true));
                if (c->train_prepared) EDV_TRY(make_t_lin(p));
            }
        std::vector<std::string> convs;
        if (cfg.conv_head) {
            for (int k = 1; k <= 4; ++k) {
                convs.push_back("head.conv_depth_" + std::to_string(k) + ".head.0");
                convs.push_back("head.conv_depth_" + std::to_string(k) + ".head.2");
            }
        } else {
            convs = {"head.scratch.output_conv1", "head.scratch.output_conv2.0"};
        }
        for (const std::string &cv : convs) {
            EDV_TRY(pack_c3(cv));
            if (c->train_prepared) EDV_TRY(make_b_c3(cv));
        }
        EDV_TRY(build_x6(true));  // fc1 / fc2 carry the factors: their planes follow the fold
        return 0;
    }

    // position table for the current patch grid (vision_transformer.py:186-217)
    int pos_table(const float **out) {
        const float *pos;
        EDV_TRY(param("pretrained.pos_embed", &pos, 3));
        const int Npos = cfg.pos_tokens - 1;
        const int npatch = ntok - 1;
        if (npatch == Npos && cfg.image_h == cfg.image_w) {
            *out = pos;
            return 0;
        }
        const int S = (int)std::lround(std::sqrt((double)Npos));
        EDV_CHECK(S * S == Npos, "pos_embed grid is not square");
        float *tab;
        EDV_TRY(wsbuf("pos_eff", (size_t)ntok * D, &tab));
        // ATen receives scale_factor as double and uses float(1/scale) (UpSample.h compute_scales_value)
        const double sh = ((double)ph + 0.1) / std::sqrt((double)Npos), sw = ((double)pw + 0.1) / std::sqrt((double)Npos);
        EDV_CHECK((int)std::floor(S * sh) == ph && (int)std::floor(S * sw) == pw, "pos-embed resample size mismatch");
        if (c0) EDV_TRY(copy_f32(pos, tab, D, st));
        EDV_TRY(bicubic_pos(pos + D, tab + (size_t)c0 * D, S, D, ph, pw, (float)(1.0 / sh), (float)(1.0 / sw), st));
        c->launches += 2;
        *out = tab;
        return 0;
    }

    int snapshot(const std::string &name, const float *src, size_t n) {
        if (c->capture) {
            float *dst;
            EDV_TRY(wsbuf("stage." + name, n, &dst));
            EDV_TRY(copy_f32(src, dst, (long long)n, st));
            c->stages[name] = {dst, n};
        }
        return 0;
    }

    // ---- motion module, in place on x [F, P, C] channels-last (motion_module.py:102-126,164-177) ----
    // extra (optional, shaped like x): added to the output as well -- the skip branch of the following fusion block
    int motion_module(int m, float *x, int P, int C, const float *extra = nullptr) {
        const std::string p = "head.motion_modules." + std::to_string(m) + ".temporal_transformer";
        const std::string tb = p + ".transformer_blocks.0";
        const long long M = (long long)F * P;
        float *gn, *h, *hn, *qkv3, *att, *ff1, *ff2, *stats;
        // training keeps: the module input, the GroupNorm statistics, h before each of its three residual updates,
        // both q|k|v and the GEGLU input (names tagged with the module index); inference reuses one set of buffers
        // scratch that nobody reads later comes in two sets: "mmb." for module 1, which may run beside module 0 on another stream
        const std::string sc_ = m == 1 ? "mmb." : "mm.";
        const std::string tg = c->train ? "mm" + std::to_string(m) + "." : sc_;
        float *hs[4];  // h after proj_in, after attention 0, after attention 1, after the feed-forward
        const float *xin = x;
        EDV_TRY(wsbuf(sc_ + "gn", (size_t)M * C, &gn));
        EDV_TRY(wsbuf(tg + "h", (size_t)M * C, &h));
        hs[0] = hs[1] = hs[2] = hs[3] = h;
        if (c->train) {
            float *xc;
            EDV_TRY(wsbuf(tg + "xin", (size_t)M * C, &xc));
            EDV_TRY(copy_f32(x, xc, M * C, st));
            xin = xc;
            for (int k = 1; k < 4; ++k) EDV_TRY(wsbuf(tg + "h" + std::to_string(k), (size_t)M * C, &hs[k]));
        }
        EDV_TRY(wsbuf(sc_ + "hn", (size_t)M * C, &hn));
        EDV_TRY(wsbuf(sc_ + "att", (size_t)M * C, &att));
        EDV_TRY(wsbuf(tg + "ff1", (size_t)M * 8 * C, &ff1));
        EDV_TRY(wsbuf(c->train ? tg + "ff2" : sc_ + "ff2", (size_t)M * 4 * C, &ff2));  // input of ff.net.2: its LoRA gradient needs it
        EDV_TRY(wsbuf(tg + "stats", (size_t)F * 32 * 2, &stats));
        const float *w, *b;
        EDV_TRY(param(p + ".norm.weight", &w));
        EDV_TRY(param(p + ".norm.bias", &b));
        // coalesced two-stage statistics for the large maps only: [8,1369,192] 23.3 -> 16.3 us, [8,5476,64] 34.5 -> 18.3 us, but the small ones
        // ([8,361,384] 11.1 -> 14.7 us) lose to the third launch
        float *gnpart = nullptr;
        size_t gnpart_n = 0;
        if ((long long)F * P * C >= 1500000ll) {
            gnpart_n = groupnorm_workspace(F, P, C);
            EDV_TRY(wsbuf(sc_ + "gnpart", gnpart_n, &gnpart));
        }
        {
            HbmScope b_(c, KC_GROUPNORM, st, 8.0 * (double)M * C);  // statistics + apply, 2-3 launches
            EDV_TRY(groupnorm(xin, w, b, gn, stats, F, P, C, 32, 1e-6f, st, gnpart, gnpart_n));
        }
        c->launches += 2;
        EDV_TRY(param(p + ".proj_in.weight", &w));
        EDV_TRY(param(p + ".proj_in.bias", &b));
        EDV_TRY(linear(gn, M, C, w, C, b, hs[0]));
        for (int a = 0; a < 2; ++a) {
            const std::string ab = tb + ".attention_blocks." + std::to_string(a);
            const float *pe = nullptr, *rope = nullptr;  // "ape": sinusoid added by the LayerNorm kernel; "rope": q|k rotated after the projection
            if (cfg.pe_rope) EDV_TRY(param(ab + ".freqs_cis", &rope, 3));
            else EDV_TRY(param(ab + ".pos_encoder.pe", &pe));
            EDV_TRY(wsbuf(c->train ? tg + "qkv" + std::to_string(a) : sc_ + "qkv", (size_t)M * 3 * C, &qkv3));
            EDV_TRY(ln(hs[a], identity_map(), tb + ".norms." + std::to_string(a), hn, M, C, 1e-5f, pe, P, T));
            const float *wqkv;
            EDV_TRY(packedw(ab + ".qkv", &wqkv));
            EDV_TRY(linear(hn, M, C, wqkv, 3 * C, nullptr, qkv3));
            if (rope) {
                EDV_TRY(rope_qk(qkv3, rope, B, T, P, C, false, st));
                c->launches++;
            }
            {
                Bracket b_(c, KC_ATTN_TEMPORAL, st);
                EDV_TRY(attn_temporal(qkv3, att, B, T, P, C, 8, st));
            }
            c->launches++;
            EDV_TRY(param(ab + ".to_out.0.weight", &w));
            EDV_TRY(param(ab + ".to_out.0.bias", &b));
            EDV_TRY(linear(att, M, C, w, C, b, hs[a + 1], ACT_NONE, nullptr, hs[a]));
        }
        EDV_TRY(ln(hs[2], identity_map(), tb + ".ff_norm", hn, M, C, 1e-5f));
        EDV_TRY(param(tb + ".ff.net.0.proj.weight", &w));
        EDV_TRY(param(tb + ".ff.net.0.proj.bias", &b));
        // Inference: the projection and the GEGLU are ONE launch (EP = 6 of gemm_dma.hip on the interleaved weight made by edv_prepare): the [M, 8C]
        // projection is never written.  Training keeps it (the GEGLU backward reads it), so it runs the two launches.  EDV_GEGLU_FUSED=0: A/B.
        static const bool geglu_fused = [] {
            const char *e = getenv("EDV_GEGLU_FUSED");
            return !(e && atoi(e) == 0);
        }();
        bool fused = false;
        if (!c->train && geglu_fused && c->packed.count(tb + ".ff.net.0.geglu.w")) {
            GemmDesc g;
            const float *wi, *bi;
            EDV_TRY(packedw(tb + ".ff.net.0.geglu.w", &wi));
            EDV_TRY(packedw(tb + ".ff.net.0.geglu.b", &bi));
            g.A = hn; g.lda = C; g.W = wi; g.ldw = C; g.C = ff2; g.ldc = 4 * C; g.M = M; g.N = 8 * C; g.K = C; g.bias = bi; g.geglu = 1;
            if (gemm_geglu_supported(g)) {
                c->launches++;
                if (c->prof_mask & (1u << KC_LINEAR)) {  // 2 M N K; A, W read once, the half-width output written once
                    c->prof_flops[KC_LINEAR] += 2.0 * (double)M * (8 * C) * C;
                    c->prof_bytes[KC_LINEAR] += 4.0 * ((double)M * C + (double)8 * C * C + (double)M * 4 * C);
                }
                Bracket b_(c, KC_LINEAR, st);
                EDV_TRY(gemm_ws(g));
                fused = true;
            }
        }
        if (!fused) {
            EDV_TRY(linear(hn, M, C, w, 8 * C, b, ff1));
            {
                HbmScope b_(c, KC_GEGLU, st, 4.0 * (double)M * 12 * C);
                EDV_TRY(geglu(ff1, ff2, M, 4 * C, st));
            }
            c->launches++;
        }
        EDV_TRY(lin_w(tb + ".ff.net.2", &w));
        EDV_TRY(param(tb + ".ff.net.2.bias", &b));
        EDV_TRY(linear(ff2, M, 4 * C, w, C, b, hs[3], ACT_NONE, nullptr, hs[2]));
        EDV_TRY(param(p + ".proj_out.weight", &w));
        EDV_TRY(param(p + ".proj_out.bias", &b));
        {
            GemmDesc g;
            g.A = hs[3]; g.lda = C; g.W = w; g.ldw = C; g.C = x; g.ldc = C; g.M = M; g.N = C; g.K = C;
            g.bias = b; g.R1 = x; g.ldr1 = C; g.R2 = extra; g.ldr2 = C;
            c->launches++;
            Bracket b_(c, KC_LINEAR, st);
            EDV_TRY(gemm_ws(g));
        }
        return 0;
    }

    // ---- FeatureFusionBlock (util/blocks.py:135-162); x, skip: [F,h,w,Fe]; out: [F,oh,ow,Fe] ----
    // The 1x1 out_conv is applied BEFORE the bilinear upsample: both are linear, the interpolation
    // weights sum to one, so conv1x1(up(x)) == up(conv1x1(x)) exactly in real arithmetic, at 1/4 of
    // the GEMM work.
    // ResidualConvUnit (util/blocks.py:68-91): x + conv2(relu(conv1(relu(x)))).  The inner ReLU is applied by conv1's epilogue (its output has no
    // other reader) instead of on conv2's A fragments: 32 v_max_f32 per k-tile less in conv2's loop, same values.  The backward's mask
    // (t1 > 0) reads the same from relu(t1).
    int fusion(int j, const float *x, const float *skip, int h, int w, int oh, int ow, float *out) {
        const std::string p = "head.scratch.refinenet" + std::to_string(j);
        const size_t n = (size_t)F * h * w * Fe;
        float *t1, *t2, *s, *t1a, *t1b;
        // training keeps both conv1 outputs and the sum s (ReLU masks of the backward), tagged with the block index
        const std::string tg = c->train ? "fu" + std::to_string(j) + "." : "fu.";
        EDV_TRY(wsbuf("fu.t1", n, &t1));
        EDV_TRY(wsbuf("fu.t2", n, &t2));
        t1a = t1b = t1;
        if (c->train) {
            EDV_TRY(wsbuf(tg + "t1a", n, &t1a));
            EDV_TRY(wsbuf(tg + "t1b", n, &t1b));
        }
        const float *w1, *b1, *w2, *b2;
        const float *cur = x;
        if (skip) {
            EDV_TRY(wsbuf(tg + "s", n, &s));
            EDV_TRY(packedw(p + ".resConfUnit1.conv1.weight", &w1));
            EDV_TRY(rcu_bias(p + ".resConfUnit1.conv1", &b1));
            EDV_TRY(packedw(p + ".resConfUnit1.conv2.weight", &w2));
            EDV_TRY(rcu_bias(p + ".resConfUnit1.conv2", &b2));
            EDV_TRY(conv3(skip, h, w, Fe, w1, b1, Fe, 1, t1a, true, ACT_RELU));
            // s = x + rcu1(skip) = x + skip + conv2(relu(t1)): both adds ride the conv2 epilogue
            // (skip_add at util/blocks.py:90 and :146)
            EDV_TRY(conv3(t1a, h, w, Fe, w2, b2, Fe, 1, s, false, ACT_NONE, skip, x));
            cur = s;
        }
        EDV_TRY(packedw(p + ".resConfUnit2.conv1.weight", &w1));
        EDV_TRY(rcu_bias(p + ".resConfUnit2.conv1", &b1));
        EDV_TRY(packedw(p + ".resConfUnit2.conv2.weight", &w2));
        EDV_TRY(rcu_bias(p + ".resConfUnit2.conv2", &b2));
        EDV_TRY(conv3(cur, h, w, Fe, w1, b1, Fe, 1, t1b, true, ACT_RELU));
        EDV_TRY(conv3(t1b, h, w, Fe, w2, b2, Fe, 1, t2, false, ACT_NONE, cur, nullptr));
        const float *wo, *bo;
        EDV_TRY(param(p + ".out_conv.weight", &wo));
        EDV_TRY(param(p + ".out_conv.bias", &bo));
        EDV_TRY(linear(t2, (long long)F * h * w, Fe, wo, Fe, bo, t1));
        {
            HbmScope b_(c, KC_BILINEAR, st, 4.0 * (double)F * Fe * ((double)h * w + (double)oh * ow));
            EDV_TRY(bilinear(t1, out, F, h, w, Fe, oh, ow, ACT_NONE, st));
        }
        c->launches++;
        return 0;
    }

    // The skip branch of fusion block j without its x:  u = skip + conv2(relu(conv1(relu(skip))))  (resConfUnit1, blocks.py:146).
    // It depends on layerN_rn only, so it can run beside the fusion chain on another stream; whoever produces x adds u.
    int fusion_skip_branch(int j, const float *skip, int h, int w, float *u) {
        const std::string p = "head.scratch.refinenet" + std::to_string(j);
        float *t;
        EDV_TRY(wsbuf("fus.t" + std::to_string(j), (size_t)F * h * w * Fe, &t));
        const float *w1, *b1, *w2, *b2;
        EDV_TRY(packedw(p + ".resConfUnit1.conv1.weight", &w1));
        EDV_TRY(rcu_bias(p + ".resConfUnit1.conv1", &b1));
        EDV_TRY(packedw(p + ".resConfUnit1.conv2.weight", &w2));
        EDV_TRY(rcu_bias(p + ".resConfUnit1.conv2", &b2));
        EDV_TRY(conv3(skip, h, w, Fe, w1, b1, Fe, 1, t, true, ACT_RELU));
        return conv3(t, h, w, Fe, w2, b2, Fe, 1, u, false, ACT_NONE, skip, nullptr);
    }
    // The rest of fusion block j from s = x + u:  out = up(out_conv(s + conv2(relu(conv1(relu(s)))))) (+ add)
    int fusion_tail(int j, const float *sx, int h, int w, int oh, int ow, float *out, const float *add) {
        const std::string p = "head.scratch.refinenet" + std::to_string(j);
        const size_t n = (size_t)F * h * w * Fe;
        float *t1, *t2;
        EDV_TRY(wsbuf("fu.t1", n, &t1));
        EDV_TRY(wsbuf("fu.t2", n, &t2));
        const float *w1, *b1, *w2, *b2, *wo, *bo;
        EDV_TRY(packedw(p + ".resConfUnit2.conv1.weight", &w1));
        EDV_TRY(rcu_bias(p + ".resConfUnit2.conv1", &b1));
        EDV_TRY(packedw(p + ".resConfUnit2.conv2.weight", &w2));
        EDV_TRY(rcu_bias(p + ".resConfUnit2.conv2", &b2));
        EDV_TRY(conv3(sx, h, w, Fe, w1, b1, Fe, 1, t1, true, ACT_RELU));
        EDV_TRY(conv3(t1, h, w, Fe, w2, b2, Fe, 1, t2, false, ACT_NONE, sx, nullptr));
        EDV_TRY(param(p + ".out_conv.weight", &wo));
        EDV_TRY(param(p + ".out_conv.bias", &bo));
        EDV_TRY(linear(t2, (long long)F * h * w, Fe, wo, Fe, bo, t1));
        {
            HbmScope b_(c, KC_BILINEAR, st, 4.0 * (double)F * Fe * ((double)h * w + (double)oh * ow * (add ? 2 : 1)));
            EDV_TRY(bilinear(t1, out, F, h, w, Fe, oh, ow, ACT_NONE, st, add));
        }
        c->launches++;
        return 0;
    }

    // ---- ResBottleneckBlock on the patch tokens of encoder block i (block.py:146-150, layers/utils.py:90-153):
    // 1x1 -> LN -> GELU -> 3x3 -> LN -> GELU -> 1x1 -> LN, added to the patch rows of the residual stream.
    int res_bottleneck(int i, float *xt) {
        // the reference reshapes to the Block's construction-time grid, input_size=(224,280) -> 16x20 (block.py:70-73)
        EDV_CHECK(ph == 16 && pw == 20, "shape '[B, 16, 20, C]' is invalid for the patch tokens: residual blocks need image_shape (224, 280)");
        const std::string p = "pretrained.blocks." + std::to_string(i) + ".residual_";
        const int Cb = D / 8;
        const long long MP = (long long)F * P0;
        float *t1, *t2, *t3;
        EDV_TRY(wsbuf("rb.t1" + rb_suffix, (size_t)MP * Cb, &t1));
        EDV_TRY(wsbuf("rb.t2" + rb_suffix, (size_t)MP * Cb, &t2));
        EDV_TRY(wsbuf("rb.t3" + rb_suffix, (size_t)MP * D, &t3));
        const float *w, *nw, *nb;
        EDV_TRY(param(p + ".conv1.weight", &w, 4));
        {
            GemmDesc g;
            g.A = xt; g.lda = D; g.a_map = RowMap{P0, ntok, c0}; g.W = w; g.ldw = D; g.C = t1; g.ldc = Cb; g.M = MP; g.N = Cb; g.K = D;
            c->launches++;
            Bracket b_(c, KC_LINEAR, st);
            EDV_TRY(gemm_ws(g));
        }
        EDV_TRY(param(p + ".norm1.weight", &nw));
        EDV_TRY(param(p + ".norm1.bias", &nb));
        EDV_TRY(layernorm(t1, identity_map(), nw, nb, t2, identity_map(), MP, Cb, 1e-6f, nullptr, 0, 0, st, ACT_GELU));
        EDV_TRY(packedw(p + ".conv2.weight", &w));
        EDV_TRY(conv3(t2, ph, pw, Cb, w, nullptr, Cb, 1, t1, false));
        EDV_TRY(param(p + ".norm2.weight", &nw));
        EDV_TRY(param(p + ".norm2.bias", &nb));
        EDV_TRY(layernorm(t1, identity_map(), nw, nb, t2, identity_map(), MP, Cb, 1e-6f, nullptr, 0, 0, st, ACT_GELU));
        EDV_TRY(param(p + ".conv3.weight", &w, 4));
        EDV_TRY(linear(t2, MP, Cb, w, D, nullptr, t3));
        EDV_TRY(param(p + ".norm3.weight", &nw));
        EDV_TRY(param(p + ".norm3.bias", &nb));
        EDV_TRY(layernorm(t3, identity_map(), nw, nb, xt, RowMap{P0, ntok, c0}, MP, D, 1e-6f, nullptr, 0, 0, st, ACT_NONE, true));
        c->launches += 3;
        return 0;
    }

    // Training form of res_bottleneck: the same arithmetic, with LayerNorm and GELU as separate launches so that every
    // intermediate the backward needs is kept (rows = F * 320 patch tokens only: the blocks exist at image_shape (224, 280)).
    int res_bottleneck_train(int i, float *xt) {
        EDV_CHECK(ph == 16 && pw == 20, "shape '[B, 16, 20, C]' is invalid for the patch tokens: residual blocks need image_shape (224, 280)");
        const std::string p = "pretrained.blocks." + std::to_string(i) + ".residual_", tg = "rbt" + std::to_string(i) + ".";
        const int Cb = D / 8;
        const long long MP = (long long)F * P0;
        float *xp, *t1a, *ln1, *a1, *t1b, *ln2, *a2, *t3;
        EDV_TRY(trainbuf(tg + "xp", (size_t)P0 * D, &xp));
        EDV_TRY(trainbuf(tg + "t1a", (size_t)P0 * Cb, &t1a));
        EDV_TRY(trainbuf(tg + "ln1", (size_t)P0 * Cb, &ln1));
        EDV_TRY(trainbuf(tg + "a1", (size_t)P0 * Cb, &a1));
        EDV_TRY(trainbuf(tg + "t1b", (size_t)P0 * Cb, &t1b));
        EDV_TRY(trainbuf(tg + "ln2", (size_t)P0 * Cb, &ln2));
        EDV_TRY(trainbuf(tg + "a2", (size_t)P0 * Cb, &a2));
        EDV_TRY(trainbuf(tg + "t3", (size_t)P0 * D, &t3));
        for (int f = 0; f < F; ++f)  // the patch rows of the residual stream, compact (block.py:146: .clone())
            EDV_TRY(copy_f32(xt + ((size_t)f * ntok + c0) * D, xp + (size_t)f * P0 * D, (long long)P0 * D, st));
        const float *w, *nw, *nb;
        EDV_TRY(param(p + ".conv1.weight", &w, 4));
        EDV_TRY(linear(xp, MP, D, w, Cb, nullptr, t1a));
        EDV_TRY(param(p + ".norm1.weight", &nw));
        EDV_TRY(param(p + ".norm1.bias", &nb));
        EDV_TRY(layernorm(t1a, identity_map(), nw, nb, ln1, identity_map(), MP, Cb, 1e-6f, nullptr, 0, 0, st));
        EDV_TRY(ew_bwd(ln1, nullptr, nullptr, a1, MP * Cb, 3, st));
        EDV_TRY(packedw(p + ".conv2.weight", &w));
        EDV_TRY(conv3(a1, ph, pw, Cb, w, nullptr, Cb, 1, t1b, false));
        EDV_TRY(param(p + ".norm2.weight", &nw));
        EDV_TRY(param(p + ".norm2.bias", &nb));
        EDV_TRY(layernorm(t1b, identity_map(), nw, nb, ln2, identity_map(), MP, Cb, 1e-6f, nullptr, 0, 0, st));
        EDV_TRY(ew_bwd(ln2, nullptr, nullptr, a2, MP * Cb, 3, st));
        EDV_TRY(param(p + ".conv3.weight", &w, 4));
        EDV_TRY(linear(a2, MP, Cb, w, D, nullptr, t3));
        EDV_TRY(param(p + ".norm3.weight", &nw));
        EDV_TRY(param(p + ".norm3.bias", &nb));
        EDV_TRY(layernorm(t3, identity_map(), nw, nb, xt, RowMap{P0, ntok, c0}, MP, D, 1e-6f, nullptr, 0, 0, st, ACT_NONE, true));
        c->launches += 5 + F;
        return 0;
    }
    // y = LN(x) * w + b over `dim` channels: input gradient into dx, and (grad_res) dL/dw = colsum(dy * xhat), dL/db = colsum(dy)
    int ln_affine_bwd(const std::string &norm, const float *x, const float *dy, float *dx, long long rows, int dim) {
        const float *w;
        EDV_TRY(param(norm + ".weight", &w));
        EDV_TRY(layernorm_bwd(x, identity_map(), w, dy, identity_map(), dx, identity_map(), rows, dim, 1e-6f, false, st));
        if (!c->grad_res) return 0;
        float *ones, *zeros, *xhat, *part, *dw, *db;
        EDV_TRY(wsbuf("g.rb.ones", (size_t)D, &ones));
        EDV_TRY(wsbuf("g.rb.zeros", (size_t)D, &zeros));
        EDV_HIP(hipMemsetD32Async((hipDeviceptr_t)ones, 0x3f800000, (size_t)D, st));
        EDV_HIP(hipMemsetAsync(zeros, 0, (size_t)D * sizeof(float), st));
        EDV_TRY(wsbuf("g.rb.xhat", (size_t)rows * D, &xhat));
        EDV_TRY(wsbuf("g.rb.part", (size_t)TALL_SPLITS * D, &part));
        EDV_TRY(gradbuf(norm + ".weight", (size_t)dim, &dw));
        EDV_TRY(gradbuf(norm + ".bias", (size_t)dim, &db));
        EDV_TRY(layernorm(x, identity_map(), ones, zeros, xhat, identity_map(), rows, dim, 1e-6f, nullptr, 0, 0, st));
        EDV_TRY(col_dot(dy, xhat, rows, dim, nullptr, part, dw, st));
        EDV_TRY(col_dot(dy, nullptr, rows, dim, nullptr, part, db, st));
        return 0;
    }
    // dW[N, K] = dY^T X for a 1x1 convolution / linear without bias (dY [M, N], X [M, K]): both operands transposed, then the NT GEMM
    int linear_wgrad(const std::string &name, const float *dY, int N, const float *X, int K, long long M) {
        if (!c->grad_res) return 0;
        float *dyt, *xt_, *dw;
        EDV_TRY(wsbuf("g.rb.dyt", (size_t)M * D, &dyt));
        EDV_TRY(wsbuf("g.rb.xt", (size_t)M * D, &xt_));
        EDV_TRY(gradbuf(name, (size_t)N * K, &dw));
        EDV_TRY(transpose_scale(dY, N, nullptr, dyt, (int)M, N, st));  // [M, N] -> [N, M]
        EDV_TRY(transpose_scale(X, K, nullptr, xt_, (int)M, K, st));   // [M, K] -> [K, M]
        return linear(dyt, N, (int)M, xt_, K, nullptr, dw);
    }
    // backward of the residual block of encoder block i: dxt (gradient of the block output, [F*ntok, D]) gains, on its patch rows,
    // the gradient that flows through conv1 .. norm3 (the identity path is already in dxt)
    int res_bottleneck_bwd(int i, float *dxt) {
        const std::string p = "pretrained.blocks." + std::to_string(i) + ".residual_", tg = "rbt" + std::to_string(i) + ".";
        const int Cb = D / 8;
        const long long MP = (long long)F * P0;
        const float *xp, *t1a, *ln1, *a1, *t1b, *ln2, *a2, *t3;
        EDV_TRY(saved(tg + "xp", &xp));
        EDV_TRY(saved(tg + "t1a", &t1a));
        EDV_TRY(saved(tg + "ln1", &ln1));
        EDV_TRY(saved(tg + "a1", &a1));
        EDV_TRY(saved(tg + "t1b", &t1b));
        EDV_TRY(saved(tg + "ln2", &ln2));
        EDV_TRY(saved(tg + "a2", &a2));
        EDV_TRY(saved(tg + "t3", &t3));
        float *dout, *dD, *dC1, *dC2;
        EDV_TRY(wsbuf("g.rb.dout", (size_t)MP * D, &dout));
        EDV_TRY(wsbuf("g.rb.dD", (size_t)MP * D, &dD));
        EDV_TRY(wsbuf("g.rb.dC1", (size_t)MP * Cb, &dC1));
        EDV_TRY(wsbuf("g.rb.dC2", (size_t)MP * Cb, &dC2));
        for (int f = 0; f < F; ++f) EDV_TRY(copy_f32(dxt + ((size_t)f * ntok + c0) * D, dout + (size_t)f * P0 * D, (long long)P0 * D, st));
        EDV_TRY(ln_affine_bwd(p + ".norm3", t3, dout, dD, MP, D));                 // out = LN3(t3)
        EDV_TRY(linear_wgrad(p + ".conv3.weight", dD, D, a2, Cb, MP));            // t3 = a2 W3^T
        EDV_TRY(dgemm(dD, MP, D, p + ".conv3", Cb, dC1));
        EDV_TRY(ew_bwd(dC1, ln2, nullptr, dC1, MP * Cb, 1, st));                   // a2 = gelu(ln2)
        EDV_TRY(ln_affine_bwd(p + ".norm2", t1b, dC1, dC2, MP, Cb));              // ln2 = LN2(t1b)
        if (c->grad_res) {
            float *dw, *ws;
            EDV_TRY(gradbuf(p + ".conv2.weight", (size_t)Cb * Cb * 9, &dw));
            const size_t need = conv3_wgrad_workspace(F, ph, pw, Cb, Cb);
            EDV_TRY(wsbuf("g.wgrad", need, &ws));
            EDV_TRY(conv3_wgrad(a1, dC2, dw, F, ph, pw, Cb, Cb, ws, need, false, st));  // t1b = conv2(a1)
        }
        EDV_TRY(dconv3(dC2, ph, pw, Cb, p + ".conv2", Cb, dC1));
        EDV_TRY(ew_bwd(dC1, ln1, nullptr, dC1, MP * Cb, 1, st));                   // a1 = gelu(ln1)
        EDV_TRY(ln_affine_bwd(p + ".norm1", t1a, dC1, dC2, MP, Cb));              // ln1 = LN1(t1a)
        EDV_TRY(linear_wgrad(p + ".conv1.weight", dC2, Cb, xp, D, MP));           // t1a = xp W1^T
        EDV_TRY(dgemm(dC2, MP, Cb, p + ".conv1", D, dD));
        for (int f = 0; f < F; ++f) {  // patch rows of dxt += the branch's input gradient
            float *dst = dxt + ((size_t)f * ntok + c0) * D;
            EDV_TRY(ew_bwd(dD + (size_t)f * P0 * D, nullptr, dst, dst, (long long)P0 * D, 0, st));
        }
        c->launches += 20 + 2 * F;
        return 0;
    }

    struct EncBufs {
        float *cols, *xt, *xn, *qkv, *att, *hid;
        float *tap[4], *tapcls[4];
        const float *pos;
        float *attws;      // attention split workspace, one region of attws_each floats per encoder stream
        size_t attws_each;
        float *skws;       // GEMM stream-K workspace, one region of skws_each floats per encoder stream
        size_t skws_each;
    };
    int ensure_streams() {
        if (c->sub[0]) return 0;
        for (int h = 0; h < 4; ++h) {
            EDV_HIP(hipStreamCreateWithFlags(&c->sub[h], hipStreamNonBlocking));
            EDV_HIP(hipEventCreateWithFlags(&c->ev_join[h], hipEventDisableTiming));
        }
        EDV_HIP(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
        for (int k = 0; k < 6; ++k) EDV_HIP(hipEventCreateWithFlags(&c->ev_x[k], hipEventDisableTiming));
        return 0;
    }
    // a kept activation with `per_frame` floats per frame: sized for every frame of the clip, returned at this frame group's offset
    int trainbuf(const std::string &name, size_t per_frame, float **out) {
        float *base;
        EDV_TRY(wsbuf(name, (size_t)enc_F * per_frame, &base));
        *out = base + (size_t)enc_f0 * per_frame;
        return 0;
    }
    // encoder on frames [f0, f0 + nf) enqueued on stream s (vision_transformer.py:279-289 + :317-321)
    int encoder_range(const EncBufs &eb, const float *x, int f0, int nf, int H, int W, hipStream_t s, int lane = 0) {
        st = s;
        skws = eb.skws ? eb.skws + (size_t)lane * eb.skws_each : nullptr;
        skws_floats = eb.skws_each;
        F = nf;
        enc_f0 = f0;
        const long long MT = (long long)nf * ntok;
        float *cols = eb.cols + (size_t)f0 * P0 * PE_K, *xt = eb.xt + (size_t)f0 * ntok * D, *xn = eb.xn + (size_t)f0 * ntok * D;
        float *qkv = eb.qkv + (size_t)f0 * ntok * 3 * D, *att = eb.att + (size_t)f0 * ntok * D, *hid = eb.hid + (size_t)f0 * ntok * 4 * D;
        float *tap[4], *tapcls[4];
        for (int j = 0; j < 4; ++j) {
            tap[j] = eb.tap[j] + (size_t)f0 * P0 * D;
            tapcls[j] = eb.tapcls[j] ? eb.tapcls[j] + (size_t)f0 * D : nullptr;
        }
        const float *pos = eb.pos;
        rb_suffix = "." + std::to_string(f0);
        if (c->train) EDV_TRY(trainbuf("t.x.0", (size_t)ntok * D, &xt));  // block i reads t.x.i and writes t.xmid.i, t.x.(i+1)
        {
            HbmScope b_(c, KC_PATCHIFY, st, 4.0 * (double)F * (3.0 * H * W + (double)P0 * PE_K));
            EDV_TRY(patchify(x + (size_t)f0 * 3 * H * W, cols, F, H, W, cfg.image_h, cfg.image_w, st, PE_K));
        }
        c->launches++;
        {
            const float *w, *b;
            EDV_TRY(packedw("pretrained.patch_embed.proj.weight", &w));  // rows padded from 588 to PE_K (edv_prepare)
            EDV_TRY(param("pretrained.patch_embed.proj.bias", &b));
            GemmDesc g;
            g.A = cols; g.lda = PE_K; g.W = w; g.ldw = PE_K; g.C = xt; g.ldc = D; g.M = (long long)F * P0; g.N = D; g.K = PE_K;
            g.bias = b;
            g.c_map = RowMap{P0, ntok, c0};
            g.R1 = pos; g.ldr1 = D; g.r1_map = RowMap{P0, 0, c0};
            EDV_TRY(gemm_ws(g));
            c->launches++;
            if (c0) {
                const float *cls;
                EDV_TRY(param("pretrained.cls_token", &cls));
                EDV_TRY(cls_rows(cls, pos, xt, F, ntok, D, st));
                c->launches++;
            }
        }
        EDV_TRY(snapshot("tokens", xt, (size_t)MT * D));

        int tapj = 0;
        in_encoder = true;
        for (int i = 0; i < depth; ++i) {
            const std::string bp = "pretrained.blocks." + std::to_string(i);
            const float *w, *b, *gam;
            // inference: one residual stream updated in place; training: every block keeps its input, its mid-point,
            // its normed MLP input, q|k|v, the attention output + log-sum-exp and the fc1 pre-activation
            float *x_in = xt, *x_mid = xt, *x_out = xt, *xn2 = xn, *lse = nullptr, *pre = nullptr;
            if (c->train) {
                const std::string is = "." + std::to_string(i);
                x_in = xt;
                EDV_TRY(trainbuf("t.xmid" + is, (size_t)ntok * D, &x_mid));
                EDV_TRY(trainbuf("t.x." + std::to_string(i + 1), (size_t)ntok * D, &x_out));
                EDV_TRY(trainbuf("t.xn2" + is, (size_t)ntok * D, &xn2));
                EDV_TRY(trainbuf("t.qkv" + is, (size_t)ntok * 3 * D, &qkv));
                EDV_TRY(trainbuf("t.att" + is, (size_t)ntok * D, &att));
                EDV_TRY(trainbuf("t.lse" + is, (size_t)heads * ntok, &lse));
                EDV_TRY(trainbuf("t.pre" + is, (size_t)ntok * 4 * D, &pre));
                EDV_TRY(trainbuf("t.hid" + is, (size_t)ntok * 4 * D, &hid));
            }
            EDV_TRY(ln(x_in, identity_map(), bp + ".norm1", xn, MT, D, 1e-6f));
            EDV_TRY(param(bp + ".attn.qkv.weight", &w));
            EDV_TRY(param(bp + ".attn.qkv.bias", &b));
            EDV_TRY(linear(xn, MT, D, w, 3 * D, b, qkv));
            if (i == 0 && stagger_record) EDV_HIP(hipEventRecord(c->ev_x[5], st));  // the next frame group may start
            {
                Bracket b_(c, KC_ATTN_SPATIAL, st);
                EDV_TRY(attn_spatial(qkv, att, F, ntok, heads, eb.attws + (size_t)lane * eb.attws_each, eb.attws_each, st, lse,
                                     !c->train && c->products == EDV_PRODUCTS_BF16X6 && attn_x6_on()));
            }
            c->launches++;
            EDV_TRY(param(bp + ".attn.proj.weight", &w));
            EDV_TRY(param(bp + ".attn.proj.bias", &b));
            EDV_TRY(param(bp + ".ls1.gamma", &gam));
            EDV_TRY(linear(att, MT, D, w, D, b, x_mid, ACT_NONE, gam, x_in));
            EDV_TRY(ln(x_mid, identity_map(), bp + ".norm2", xn2, MT, D, 1e-6f));
            EDV_TRY(lin_w(bp + ".mlp.fc1", &w));
            EDV_TRY(param(bp + ".mlp.fc1.bias", &b));
            if (pre) {  // same values as the fused epilogue: GELU of the stored fp32 pre-activation
                EDV_TRY(linear(xn2, MT, D, w, 4 * D, b, pre, ACT_NONE));
                EDV_TRY(ew_bwd(pre, nullptr, nullptr, hid, MT * 4 * D, 3, st));
                c->launches++;
            } else {
                EDV_TRY(linear(xn2, MT, D, w, 4 * D, b, hid, ACT_GELU));
            }
            EDV_TRY(lin_w(bp + ".mlp.fc2", &w));
            EDV_TRY(param(bp + ".mlp.fc2.bias", &b));
            EDV_TRY(param(bp + ".ls2.gamma", &gam));
            EDV_TRY(linear(hid, MT, 4 * D, w, D, b, x_out, ACT_NONE, gam, x_mid));
            xt = x_out;
            if (cfg.residual_mask & (1u << i)) EDV_TRY(c->train ? res_bottleneck_train(i, xt) : res_bottleneck(i, xt));
            if (i == 0) EDV_TRY(snapshot("block0", xt, (size_t)MT * D));
            if (tapj < 4 && i == cfg.taps[tapj]) {
                // final norm on the tap, cls row dropped (vision_transformer.py:317-321)
                EDV_TRY(ln(xt, RowMap{P0, ntok, c0}, "pretrained.norm", tap[tapj], (long long)F * P0, D, 1e-6f));
                                // token 0 of every frame, normed: the cls token, or with include_cls_token=False the first patch
                // ("not real cls tokens", vision_transformer.py:322-324)
                if (cfg.use_clstoken) EDV_TRY(ln(xt, RowMap{1, ntok, 0}, "pretrained.norm", tapcls[tapj], F, D, 1e-6f));
                ++tapj;
            }
        }
        in_encoder = false;
        EDV_CHECK(tapj == 4, "taps must be increasing block indices < depth");
        return 0;
    }

    int forward(const float *x, int B_, int T_, int H, int W, float *const disp[4]) {
        B = B_; T = T_; F = B * T;
        ph = cfg.image_h / 14; pw = cfg.image_w / 14; P0 = ph * pw;
        c0 = cfg.include_cls_token ? 1 : 0;
        ntok = P0 + c0;
        c->launches = 0;
        c->stages.clear();
        c->F = F; c->T = T; c->ph = ph; c->pw = pw; c->ntok = ntok;
        const long long MT = (long long)F * ntok;
        const int *oc = cfg.out_channels;

        // ---------------- encoder ----------------
        float *cols, *xt, *xn, *qkv, *att, *hid;
        EDV_TRY(wsbuf("cols", (size_t)F * P0 * PE_K, &cols));
        EDV_TRY(wsbuf("xt", (size_t)MT * D, &xt));
        EDV_TRY(wsbuf("xn", (size_t)MT * D, &xn));
        EDV_TRY(wsbuf("qkv", (size_t)MT * 3 * D, &qkv));
        EDV_TRY(wsbuf("att", (size_t)MT * D, &att));
        EDV_TRY(wsbuf("hid", (size_t)MT * 4 * D, &hid));
        float *tap[4], *tapcls[4] = {nullptr, nullptr, nullptr, nullptr};
        for (int j = 0; j < 4; ++j) EDV_TRY(wsbuf("tap" + std::to_string(j), (size_t)F * P0 * D, &tap[j]));
        if (cfg.use_clstoken)
            for (int j = 0; j < 4; ++j) EDV_TRY(wsbuf("tapcls" + std::to_string(j), (size_t)F * D, &tapcls[j]));

        const float *pos;
        EDV_TRY(pos_table(&pos));
        // Frames are independent in the encoder: with two internal streams the two halves of the batch run as
        // concurrent kernels, so workgroups of different kernels (one half's attention, the other's GEMM) co-reside
        // on the CUs and fill each other's stalls and grid tails.  The head needs all T frames again (temporal attention).
        // Automatic = ONE stream since round 2.  Round 1 ran two frame groups on two streams while a block's GEMMs were short, so that one
        // group's attention filled the launch ramps and drains of the other's GEMMs (+4.8 % at T=8).  With the VALU-free GEMM loop and
        // the VALU-lean attention kernel of round 2 the two-stream form measures equal or slower (ViT-S T=8: 772.7 vs 792.3 frames/s, ViT-B
        // T=16: 280.9 vs 284.9; profiles/r02_notes.txt): co-resident kernels share a SIMD's matrix / vector ALUs, so one kernel's VALU
        // work comes out of the other's matrix time, and the attention kernel's 64 KB of LDS per workgroup leaves room for one GEMM
        // workgroup beside two of its own.  EDV_ENC_STREAMS=2..4 / edv_set_encoder_streams still select the forked form.
        int want = c->enc_streams;
        if (want <= 0) want = 1;
        int nstreams = (want > 1 && !c->capture && !c->train) ? (want > 4 ? 4 : want) : 1;
        enc_F = F;
        if (nstreams > F) nstreams = F;
        size_t attws_each = 0;  // the largest split workspace any stream's share of the frames needs
        for (int h = 0, f0 = 0; h < nstreams; ++h) {
            const int nf = (F - f0) / (nstreams - h);
            const size_t need = (std::max(attn_spatial_workspace(nf, ntok, heads), attn_spatial_workspace(nf, ntok, heads, true)) + 3) & ~(size_t)3;
            attws_each = need > attws_each ? need : attws_each;
            f0 += nf;
        }
        float *attws = nullptr;
        if (attws_each) EDV_TRY(wsbuf("attws", attws_each * nstreams, &attws));
        // Stream-K for the dense GEMMs (gemm_dma.hip): the last partial round of output tiles is split along K over the resident
        // workgroups and merged in-kernel by the last piece to arrive.  One workspace region per stream that launches GEMMs
        // concurrently: the encoder's frame-group streams, and the head's caller / internal stream pair (regions 0 and 1).
        // It applies to deep tiles only (K >= 768, grids under five rounds: gemm_dma.hip).  On by default: fc2 at T=8 138 -> 121 us,
        // ViT-B fc2 at T=8 472 -> 430 us; end to end +0.1 .. +0.7 % on ViT-S T=4/8/16, ViT-B T=8/16 and the fine-tune step
        // (profiles/r01_gemm_tile_sweep.txt).  EDV_GEMM_STREAMK=0 restores one workgroup per tile.
        static const bool gemm_streamk = [] {
            const char *e = getenv("EDV_GEMM_STREAMK");
            return !(e && atoi(e) == 0);
        }();
        const size_t skws_each = gemm_streamk ? gemm_workspace() : 0;
        const int skws_regions = nstreams > 2 ? nstreams : 2;
        float *skws_all = nullptr;
        if (skws_each) {
            EDV_TRY(wsbuf("skws", skws_each * skws_regions, &skws_all));
            if (c->skws_zeroed != skws_all) {  // fresh allocation: the arrival counters at the head of each region start at zero
                for (int h = 0; h < skws_regions; ++h) EDV_HIP(hipMemsetAsync(skws_all + (size_t)h * skws_each, 0, gemm_counter_bytes(), st));
                c->skws_zeroed = skws_all;
            }
        }
        EncBufs eb{cols, xt, xn, qkv, att, hid, {tap[0], tap[1], tap[2], tap[3]}, {tapcls[0], tapcls[1], tapcls[2], tapcls[3]}, pos, attws, attws_each,
                   skws_all, skws_each};
        skws = skws_all;  // the head runs on the caller's stream with region 0 (the encoder streams have joined by then)
        skws_floats = skws_each;
        if (nstreams == 1) {
            EDV_TRY(encoder_range(eb, x, 0, F, H, W, st));
        } else {
            EDV_TRY(ensure_streams());
            const int Fall = F;
            hipStream_t user = st;
            EDV_HIP(hipEventRecord(c->ev_fork, user));
            int f0 = 0;
            static const bool stagger = [] {
                const char *e = getenv("EDV_ENC_STAGGER");  // 0: the groups start together (A/B runs)
                return !(e && atoi(e) == 0);
            }();
            for (int h = 0; h < nstreams; ++h) {
                const int nf = (Fall - f0) / (nstreams - h);  // even split of the remaining frames
                EDV_HIP(hipStreamWaitEvent(c->sub[h], c->ev_fork, 0));
                // Group h starts when group h-1 has launched its first attention: the groups then run half a block apart, so
                // one group's attention (2 workgroups per CU, MFMA-bound) runs beside the other's GEMM ramps and drains
                // instead of beside its own kind.
                if (stagger && h > 0) EDV_HIP(hipStreamWaitEvent(c->sub[h], c->ev_x[5], 0));
                stagger_record = stagger && h + 1 < nstreams;
                const int rc = encoder_range(eb, x, f0, nf, H, W, c->sub[h], h);
                st = user; F = Fall; skws = skws_all;
                if (rc) return rc;  // edv_forward waits for the internal streams before it reports the error
                EDV_HIP(hipEventRecord(c->ev_join[h], c->sub[h]));
                EDV_HIP(hipStreamWaitEvent(user, c->ev_join[h], 0));
                f0 += nf;
            }
        }
        for (int j = 0; j < 4; ++j) c->stages["tap" + std::to_string(j)] = {tap[j], (size_t)F * P0 * D};

        // ---------------- DPT head: reassemble ----------------
        const long long MP = (long long)F * P0;
        const int h1 = 4 * ph, w1 = 4 * pw, h2 = 2 * ph, w2 = 2 * pw, h3 = ph, w3 = pw, h4 = (ph - 1) / 2 + 1, w4 = (pw - 1) / 2 + 1;
        float *pj, *l1, *l2, *l3, *l4;
        EDV_TRY(wsbuf("l1", (size_t)F * h1 * w1 * oc[0], &l1));
        EDV_TRY(wsbuf("l2", (size_t)F * h2 * w2 * oc[1], &l2));
        EDV_TRY(wsbuf("l3", (size_t)F * h3 * w3 * oc[2], &l3));
        EDV_TRY(wsbuf("l4", (size_t)F * h4 * w4 * oc[3], &l4));
        float *pjs[4];  // one projection buffer per level: the four level chains below may run on two streams
        for (int j = 0; j < 4; ++j) EDV_TRY(wsbuf("pj" + std::to_string(j), (size_t)MP * oc[j], &pjs[j]));
        float *r1, *r2, *r3, *r4;
        EDV_TRY(wsbuf("r1", (size_t)F * h1 * w1 * Fe, &r1));
        EDV_TRY(wsbuf("r2", (size_t)F * h2 * w2 * Fe, &r2));
        EDV_TRY(wsbuf("r3", (size_t)F * h3 * w3 * Fe, &r3));
        EDV_TRY(wsbuf("r4", (size_t)F * h4 * w4 * Fe, &r4));
        float *readout = nullptr, *fbias = nullptr;
        if (cfg.use_clstoken) {
            EDV_TRY(wsbuf("readout", (size_t)MP * D, &readout));
            EDV_TRY(wsbuf("readout.fb", (size_t)F * D, &fbias));
        }
        // level j: tap -> 1x1 project -> resize -> (motion module on levels 3, 4) -> 3x3 layerN_rn  (dpt_pyramid.py:52-78)
        auto level = [&](int j) -> int {
            pj = pjs[j];
            const std::string pp = "head.projects." + std::to_string(j);
            const float *w, *b;
            const float *src = tap[j];
            if (cfg.use_clstoken) {
                // readout_projects[j] = GELU(Linear(2D -> D)) on cat(x, cls): W = [W1 | W2], so
                // y = GELU(W1 x + (W2 cls + b)); the bracket is one [F, D] vector per frame (dpt_pyramid.py:54-57)
                const std::string rp = "head.readout_projects." + std::to_string(j) + ".0";
                const float *rw, *rbias;
                EDV_TRY(param(rp + ".weight", &rw, 2));
                EDV_TRY(param(rp + ".bias", &rbias));
                GemmDesc g1;
                g1.A = tapcls[j]; g1.lda = D; g1.W = rw + D; g1.ldw = 2 * D; g1.C = fbias; g1.ldc = D; g1.M = F; g1.N = D; g1.K = D; g1.bias = rbias;
                EDV_TRY(gemm_ws(g1));
                GemmDesc g2;
                g2.A = tap[j]; g2.lda = D; g2.W = rw; g2.ldw = 2 * D; g2.C = readout; g2.ldc = D; g2.M = MP; g2.N = D; g2.K = D;
                g2.P1 = fbias; g2.ldp1 = D; g2.act = ACT_GELU;
                g2.p1_map = RowMap{P0, 1, 0, 0};  // inner 0: one bias row per frame
                if (c->train) {  // keep the pre-activation of every level; GELU from the stored fp32 value (same values as the fused epilogue)
                    float *pre;
                    EDV_TRY(wsbuf("ro" + std::to_string(j) + ".pre", (size_t)MP * D, &pre));
                    g2.C = pre;
                    g2.act = ACT_NONE;
                    EDV_TRY(gemm_ws(g2));
                    EDV_TRY(ew_bwd(pre, nullptr, nullptr, readout, MP * D, 3, st));
                    c->launches++;
                } else {
                    EDV_TRY(gemm_ws(g2));
                }
                c->launches += 2;
                src = readout;
                c->stages["tapcls" + std::to_string(j)] = {tapcls[j], (size_t)F * D};
                c->stages["fbias"] = {fbias, (size_t)F * D};       // last level only (buffers are reused)
                c->stages["readout"] = {readout, (size_t)MP * D};
            }
            EDV_TRY(param(pp + ".weight", &w, 4));
            EDV_TRY(param(pp + ".bias", &b));
            float *dst = (j == 2) ? l3 : pj;  // level 3 is not resized: project straight into l3
            EDV_TRY(linear(src, MP, D, w, oc[j], b, dst));
            if (j < 2) {
                const int s = j == 0 ? 4 : 2;
                const std::string rp = "head.resize_layers." + std::to_string(j);
                const float *wt, *bt;
                EDV_TRY(packedw(rp + ".weight", &wt));
                EDV_TRY(packedw(rp + ".bias", &bt));
                GemmDesc g;
                g.A = pj; g.lda = oc[j]; g.W = wt; g.ldw = oc[j]; g.C = j == 0 ? l1 : l2; g.M = MP; g.N = s * s * oc[j]; g.K = oc[j];
                g.bias = bt; g.store = STORE_SHUFFLE; g.ps_s = s; g.ps_C = oc[j]; g.ps_h = ph; g.ps_w = pw; g.ldc = oc[j];
                EDV_TRY(gemm_ws(g));
                c->launches++;
            } else if (j == 3) {
                const float *wc, *bc;
                EDV_TRY(packedw("head.resize_layers.3.weight", &wc));
                EDV_TRY(param("head.resize_layers.3.bias", &bc));
                EDV_TRY(conv3(pj, ph, pw, oc[3], wc, bc, oc[3], 2, l4, false));
            }
            if (j == 2) EDV_TRY(motion_module(0, l3, h3 * w3, oc[2]));
            if (j == 3) EDV_TRY(motion_module(1, l4, h4 * w4, oc[3]));
            float *const ls[4] = {l1, l2, l3, l4}, *const rs[4] = {r1, r2, r3, r4};
            const int hs[4] = {h1, h2, h3, h4}, wsz[4] = {w1, w2, w3, w4};
            const float *wr;
            EDV_TRY(packedw("head.scratch.layer" + std::to_string(j + 1) + "_rn.weight", &wr));
            EDV_TRY(conv3(ls[j], hs[j], wsz[j], oc[j], wr, nullptr, Fe, 1, rs[j], false));
            return 0;
        };
        // The four level chains are independent until the fusion blocks and made of small kernels (7-80 us, a few hundred
        // workgroups each): level 4 -- the longest, with its stride-2 conv and the C = out_channels[3] motion module -- goes
        // to an internal stream, levels 3, 1, 2 stay on the caller's.  Not while training (saved activations are ordered by
        // the backward), with use_clstoken (shared readout scratch) or during a stage capture.
        static const int head_streams = [] {
            const char *e = getenv("EDV_HEAD_STREAMS");  // 1 = everything on the caller's stream, 2 (default) = one internal stream beside it
            const int v = e ? atoi(e) : 2;
            return v < 1 ? 1 : (v > 2 ? 2 : v);
        }();
        const int h0 = 8 * ph, w0 = 8 * pw;
        float *p4, *p3, *p2, *p1;
        EDV_TRY(wsbuf("p4", (size_t)F * h3 * w3 * Fe, &p4));
        EDV_TRY(wsbuf("p3", (size_t)F * h2 * w2 * Fe, &p3));
        EDV_TRY(wsbuf("p2", (size_t)F * h1 * w1 * Fe, &p2));
        EDV_TRY(wsbuf("p1", (size_t)F * h0 * w0 * Fe, &p1));
        // Measured (profiles/r01_gemm_tile_sweep.txt): +3 % at T = 8 and 16, -0.8 % at T = 32, where the head's kernels fill the
        // GPU on their own -- so only up to 16 frames per clip.
        if (head_streams > 1 && T <= 16 && !c->train && !cfg.use_clstoken && !c->capture) {
            // internal stream: level 4, then the skip branches u3, u2, u1 of the fusion blocks (they need layerN_rn only);
            // caller's stream: levels 3, 1, 2, then the fusion chain, where whoever produces a block's x adds its u:
            // motion modules 2 and 3 in their proj_out epilogue, fusion block 2 in its upsample.
            EDV_TRY(ensure_streams());
            float *u1, *u2, *u3;
            EDV_TRY(wsbuf("fu.u1", (size_t)F * h1 * w1 * Fe, &u1));
            EDV_TRY(wsbuf("fu.u2", (size_t)F * h2 * w2 * Fe, &u2));
            EDV_TRY(wsbuf("fu.u3", (size_t)F * h3 * w3 * Fe, &u3));
            hipStream_t user = st, side = c->sub[0];
            float *const ws_user = skws, *const ws_side = skws ? skws + skws_floats : nullptr;  // stream-K regions 0 and 1
            EDV_HIP(hipEventRecord(c->ev_fork, user));
            EDV_HIP(hipStreamWaitEvent(side, c->ev_fork, 0));
            st = side; skws = ws_side;
            int rc = level(3);
            st = user; skws = ws_user;
            if (rc) return rc;  // edv_forward waits for the internal streams before it reports the error
            EDV_HIP(hipEventRecord(c->ev_join[0], side));     // r4 ready
            EDV_TRY(level(2));
            if (cfg.conv_head) {  // the four HeadDepth heads read path_4..path_1 themselves: no folding of u into them
                EDV_TRY(level(0));
                EDV_TRY(level(1));
                EDV_HIP(hipStreamWaitEvent(user, c->ev_join[0], 0));
                EDV_TRY(fusion(4, r4, nullptr, h4, w4, h3, w3, p4));
                EDV_TRY(motion_module(2, p4, h3 * w3, Fe));
                EDV_TRY(fusion(3, p4, r3, h3, w3, h2, w2, p3));
                EDV_TRY(motion_module(3, p3, h2 * w2, Fe));
                EDV_TRY(fusion(2, p3, r2, h2, w2, h1, w1, p2));
                EDV_TRY(fusion(1, p2, r1, h1, w1, h0, w0, p1));
            } else {
                EDV_HIP(hipEventRecord(c->ev_x[0], user));        // r3 ready
                EDV_HIP(hipStreamWaitEvent(side, c->ev_x[0], 0));
                st = side; skws = ws_side;
                rc = fusion_skip_branch(3, r3, h3, w3, u3);
                st = user; skws = ws_user;
                if (rc) return rc;  // edv_forward waits for the internal streams before it reports the error
                EDV_HIP(hipEventRecord(c->ev_x[2], side));        // u3 ready
                EDV_TRY(level(0));
                EDV_TRY(level(1));
                EDV_HIP(hipEventRecord(c->ev_x[1], user));        // r1, r2 ready
                EDV_HIP(hipStreamWaitEvent(side, c->ev_x[1], 0));
                st = side; skws = ws_side;
                rc = fusion_skip_branch(2, r2, h2, w2, u2);
                if (!rc) EDV_HIP(hipEventRecord(c->ev_x[3], side));  // u2 ready
                if (!rc) rc = fusion_skip_branch(1, r1, h1, w1, u1);
                st = user; skws = ws_user;
                if (rc) return rc;  // edv_forward waits for the internal streams before it reports the error
                EDV_HIP(hipEventRecord(c->ev_x[4], side));        // u1 ready
                EDV_HIP(hipStreamWaitEvent(user, c->ev_join[0], 0));
                EDV_TRY(fusion(4, r4, nullptr, h4, w4, h3, w3, p4));
                EDV_HIP(hipStreamWaitEvent(user, c->ev_x[2], 0));
                EDV_TRY(motion_module(2, p4, h3 * w3, Fe, u3));   // p4 <- motion(p4) + u3 = the s of fusion block 3
                EDV_TRY(fusion_tail(3, p4, h3, w3, h2, w2, p3, nullptr));
                EDV_HIP(hipStreamWaitEvent(user, c->ev_x[3], 0));
                EDV_TRY(motion_module(3, p3, h2 * w2, Fe, u2));   // p3 <- motion(p3) + u2
                EDV_HIP(hipStreamWaitEvent(user, c->ev_x[4], 0));
                EDV_TRY(fusion_tail(2, p3, h2, w2, h1, w1, p2, u1));  // p2 <- up(...) + u1
                EDV_TRY(fusion_tail(1, p2, h1, w1, h0, w0, p1, nullptr));
            }
        } else {
            for (int j = 0; j < 4; ++j) EDV_TRY(level(j));
            EDV_TRY(fusion(4, r4, nullptr, h4, w4, h3, w3, p4));
            EDV_TRY(motion_module(2, p4, h3 * w3, Fe));
            EDV_TRY(fusion(3, p4, r3, h3, w3, h2, w2, p3));
            EDV_TRY(motion_module(3, p3, h2 * w2, Fe));
            EDV_TRY(fusion(2, p3, r2, h2, w2, h1, w1, p2));
            EDV_TRY(fusion(1, p2, r1, h1, w1, h0, w0, p1));
        }
        c->stages["mm0"] = {l3, (size_t)F * h3 * w3 * oc[2]};
        c->stages["mm1"] = {l4, (size_t)F * h4 * w4 * oc[3]};
        c->stages["path4"] = {p4, (size_t)F * h3 * w3 * Fe};
        c->stages["path3"] = {p3, (size_t)F * h2 * w2 * Fe};
        c->stages["path2"] = {p2, (size_t)F * h1 * w1 * Fe};
        c->stages["path1"] = {p1, (size_t)F * h0 * w0 * Fe};

        // ---------------- output heads ----------------
        if (!cfg.conv_head) {  // VDA head: dpt.py:117-124 + dpt_pyramid.py:88-102
            const int ih = cfg.image_h, iw = cfg.image_w, Fh = Fe / 2;
            float *o1, *up, *o2;
            EDV_TRY(wsbuf("hd.o1", (size_t)F * h0 * w0 * Fh, &o1));
            EDV_TRY(wsbuf("hd.up", (size_t)F * ih * iw * Fh, &up));
            EDV_TRY(wsbuf("hd.o2", (size_t)F * ih * iw * 32, &o2));
            const float *w, *b;
            EDV_TRY(packedw("head.scratch.output_conv1.weight", &w));
            EDV_TRY(param("head.scratch.output_conv1.bias", &b));
            EDV_TRY(conv3(p1, h0, w0, Fe, w, b, Fh, 1, o1, false));
            {
                HbmScope b_(c, KC_BILINEAR, st, 4.0 * (double)F * Fh * ((double)h0 * w0 + (double)ih * iw));
                EDV_TRY(bilinear(o1, up, F, h0, w0, Fh, ih, iw, ACT_NONE, st));
            }
            EDV_TRY(packedw("head.scratch.output_conv2.0.weight", &w));
            EDV_TRY(param("head.scratch.output_conv2.0.bias", &b));
            EDV_TRY(conv3(up, ih, iw, Fh, w, b, 32, 1, o2, false, ACT_RELU));
            EDV_TRY(param("head.scratch.output_conv2.2.weight", &w));
            EDV_TRY(param("head.scratch.output_conv2.2.bias", &b));
            {
                HbmScope b_(c, KC_DOT, st, 4.0 * (double)F * ih * iw * 33);
                EDV_TRY(dot_channels(o2, w, b, disp[0], (long long)F * ih * iw, 32, ACT_RELU, st));
            }
            int sh = ih, sw = iw;
            for (int k = 1; k < 4; ++k) {  // F.interpolate(scale_factor=0.5): floor(in/2)
                const int nh = sh / 2, nw = sw / 2;
                HbmScope b_(c, KC_BILINEAR, st, 4.0 * (double)F * ((double)sh * sw + (double)nh * nw));
                EDV_TRY(bilinear(disp[k - 1], disp[k], F, sh, sw, 1, nh, nw, ACT_NONE, st));
                sh = nh; sw = nw;
            }
            c->launches += 5;
            if (cfg.out_sigmoid) {
                if (c->train) {  // the backward needs the ReLU mask of the raw map and every sigmoid output
                    float *raw0;
                    EDV_TRY(wsbuf("hd.raw0", (size_t)F * ih * iw, &raw0));
                    EDV_TRY(copy_f32(disp[0], raw0, (long long)F * ih * iw, st));
                }
                sh = ih; sw = iw;
                for (int k = 0; k < 4; ++k) {
                    EDV_TRY(sigmoid_inplace(disp[k], (long long)F * sh * sw, st));
                    if (c->train) {
                        float *sg;
                        EDV_TRY(wsbuf("hd.sg" + std::to_string(k), (size_t)F * sh * sw, &sg));
                        EDV_TRY(copy_f32(disp[k], sg, (long long)F * sh * sw, st));
                    }
                    sh /= 2; sw /= 2;
                }
                c->launches += 4;
            }
        } else {  // four HeadDepth heads: endodav/layers.py:206-221 + dpt_pyramid.py:103-109
            const float *paths[4] = {p1, p2, p3, p4};
            const int hs[4] = {h0, h1, h2, h3}, wsz[4] = {w0, w1, w2, w3};
            const int Fh = Fe / 2;
            for (int k = 3; k >= 0; --k) {
                const std::string hp = "head.conv_depth_" + std::to_string(k + 1) + ".head.";
                // training keeps every head's intermediates (and its sigmoid output) for the backward; inference shares one scratch set
                const std::string tg = c->train ? "hd" + std::to_string(k) + "." : "hd.";
                const size_t px = (size_t)F * hs[k] * wsz[k], px0 = c->train ? px : (size_t)F * h0 * w0;
                float *o1, *up, *o2;
                EDV_TRY(wsbuf(tg + "o1", px0 * Fh, &o1));
                EDV_TRY(wsbuf(tg + "up", px0 * 4 * Fh, &up));
                EDV_TRY(wsbuf(tg + "o2", px0 * 4 * 32, &o2));
                const float *w, *b;
                EDV_TRY(packedw(hp + "0.weight", &w));
                EDV_TRY(param(hp + "0.bias", &b));
                EDV_TRY(conv3(paths[k], hs[k], wsz[k], Fe, w, b, Fh, 1, o1, false));
                {
                    HbmScope b_(c, KC_BILINEAR, st, 4.0 * (double)F * Fh * 5.0 * hs[k] * wsz[k]);
                    EDV_TRY(bilinear(o1, up, F, hs[k], wsz[k], Fh, 2 * hs[k], 2 * wsz[k], ACT_NONE, st));
                }
                EDV_TRY(packedw(hp + "2.weight", &w));
                EDV_TRY(param(hp + "2.bias", &b));
                EDV_TRY(conv3(up, 2 * hs[k], 2 * wsz[k], Fh, w, b, 32, 1, o2, false, ACT_RELU));
                EDV_TRY(param(hp + "4.weight", &w));
                EDV_TRY(param(hp + "4.bias", &b));
                {
                    HbmScope b_(c, KC_DOT, st, 4.0 * (double)px * 4 * 33);
                    EDV_TRY(dot_channels(o2, w, b, disp[k], (long long)px * 4, 32, cfg.inv_sigmoid ? ACT_SIGMOID_NEG : ACT_SIGMOID, st));
                }
                c->launches += 2;
                if (c->train) {
                    float *dk;
                    EDV_TRY(wsbuf(tg + "disp", px * 4, &dk));
                    EDV_TRY(copy_f32(disp[k], dk, (long long)px * 4, st));
                }
            }
        }
        return 0;
    }

    // =========================================================================================
    // Backward.  Trainable: the LoRA / DV-LoRA factors of mlp.fc1 / mlp.fc2 in every encoder
    // block (endodav/layers.py:5-34 names lora_A, lora_B, lora_U, lora_V); everything else is frozen, so each operator
    // contributes its input gradient only.  Mirrors forward() in reverse on the activations a training forward kept.
    float *lora_ws = nullptr;  // workspace of lora_grads for the whole backward
    size_t lora_ws_n = 0;
    int gradbuf(const std::string &name, size_t n, float **out) {
        auto it = c->flat.find(name);
        if (it != c->flat.end()) {  // the caller's flat buffer holds this gradient (edv_grad_bind_flat)
            EDV_CHECK(it->second.numel == n, "flat gradient slice of " + name + " has " + std::to_string(it->second.numel) + " floats, the gradient " +
                                                 std::to_string(n));
            it->second.written = true;
            *out = it->second.p;
            return 0;
        }
        return alloc_buf(c, c->grads, name, n, st, out);
    }
    int saved(const std::string &name, const float **out) {
        auto it = c->ws.find(name);
        EDV_CHECK(it != c->ws.end() && it->second.p, "activation not saved (run a forward with edv_set_train first): " + name);
        *out = it->second.p;
        return 0;
    }
    // transposed (NT-form) weight of dX = (dY * gamma) W, cached under "T." + key
    int make_t(const std::string &key, const float *W, int ldw, int N, int K, const float *gamma) {
        float *wt;
        EDV_TRY(pk("T." + key, (size_t)N * K, &wt));
        return transpose_scale(W, ldw, gamma, wt, N, K, st);
    }
    int make_t_lin(const std::string &p, const float *gamma = nullptr) {
        const float *W;
        EDV_TRY(lin_w(p, &W));
        const Param &q = c->params[p + ".weight"];
        EDV_CHECK(q.shape.size() >= 2, "rank of " + p);
        long long in = 1;
        for (size_t k = 1; k < q.shape.size(); ++k) in *= q.shape[k];
        return make_t(p, W, (int)in, (int)q.shape[0], (int)in, gamma);
    }
    int make_b_c3(const std::string &p) {  // flipped, in/out-swapped packed weight of the stride-1 input-gradient convolution
        const float *w;
        EDV_TRY(param(p + ".weight", &w, 4));
        const Param &q = c->params[p + ".weight"];
        float *out;
        EDV_TRY(pk("B." + p, (size_t)q.numel(), &out));
        return pack_conv3x3_bwd(w, out, (int)q.shape[0], (int)q.shape[1], st);
    }
    int prepare_train() {
        EDV_CHECK(!cfg.use_bn, "the fine-tune step with use_bn=True is not built (train-mode BatchNorm uses batch statistics)");
        EDV_CHECK(c->prepared, "edv_prepare has not run");
        const int *oc = cfg.out_channels;
        for (int i = 0; i < depth; ++i) {
            const std::string bp = "pretrained.blocks." + std::to_string(i);
            const float *g1, *g2;
            EDV_TRY(param(bp + ".ls1.gamma", &g1));
            EDV_TRY(param(bp + ".ls2.gamma", &g2));
            EDV_TRY(make_t_lin(bp + ".attn.qkv"));
            EDV_TRY(make_t_lin(bp + ".attn.proj", g1));
            EDV_TRY(make_t_lin(bp + ".mlp.fc1"));
            EDV_TRY(make_t_lin(bp + ".mlp.fc2", g2));
        }
        for (int i = 0; i < depth; ++i)
            if (cfg.residual_mask & (1u << i)) {
                const std::string rp = "pretrained.blocks." + std::to_string(i) + ".residual_";
                EDV_TRY(make_t_lin(rp + ".conv1"));
                EDV_TRY(make_t_lin(rp + ".conv3"));
                EDV_TRY(make_b_c3(rp + ".conv2"));
            }
        for (int j = 0; j < 4; ++j) EDV_TRY(make_t_lin("head.projects." + std::to_string(j)));
        if (cfg.use_clstoken)
            for (int j = 0; j < 4; ++j) {  // readout_projects[j].0.weight = [W1 | W2] (dpt.py:92-98): both halves, transposed
                const std::string rp = "head.readout_projects." + std::to_string(j) + ".0";
                const float *rw;
                EDV_TRY(param(rp + ".weight", &rw, 2));
                EDV_TRY(make_t(rp + ".w1", rw, 2 * D, D, D, nullptr));
                EDV_TRY(make_t(rp + ".w2", rw + D, 2 * D, D, D, nullptr));
            }
        for (int j = 0; j < 2; ++j) {
            const std::string rp = "head.resize_layers." + std::to_string(j);
            const int s2 = (j == 0 ? 16 : 4);
            const float *wp;
            EDV_TRY(packedw(rp + ".weight", &wp));
            EDV_TRY(make_t(rp, wp, oc[j], s2 * oc[j], oc[j], nullptr));
        }
        for (int j = 1; j <= 4; ++j) EDV_TRY(make_b_c3("head.scratch.layer" + std::to_string(j) + "_rn"));
        EDV_TRY(make_b_c3("head.resize_layers.3"));
        for (int j = 1; j <= 4; ++j) {
            const std::string p = "head.scratch.refinenet" + std::to_string(j);
            for (int u = 1; u <= 2; ++u) {
                if (j == 4 && u == 1) continue;
                EDV_TRY(make_b_c3(p + ".resConfUnit" + std::to_string(u) + ".conv1"));
                EDV_TRY(make_b_c3(p + ".resConfUnit" + std::to_string(u) + ".conv2"));
            }
            EDV_TRY(make_t_lin(p + ".out_conv"));
        }
        if (cfg.conv_head) {
            for (int k = 1; k <= 4; ++k) {
                EDV_TRY(make_b_c3("head.conv_depth_" + std::to_string(k) + ".head.0"));
                EDV_TRY(make_b_c3("head.conv_depth_" + std::to_string(k) + ".head.2"));
            }
        } else {
            EDV_TRY(make_b_c3("head.scratch.output_conv1"));
            EDV_TRY(make_b_c3("head.scratch.output_conv2.0"));
        }
        const int mmC[4] = {oc[2], oc[3], Fe, Fe};
        for (int m = 0; m < 4; ++m) {
            const std::string p = "head.motion_modules." + std::to_string(m) + ".temporal_transformer";
            const std::string tb = p + ".transformer_blocks.0";
            const int C = mmC[m];
            EDV_TRY(make_t_lin(p + ".proj_in"));
            EDV_TRY(make_t_lin(p + ".proj_out"));
            for (int a = 0; a < 2; ++a) {
                const std::string ab = tb + ".attention_blocks." + std::to_string(a);
                const float *wq;
                EDV_TRY(packedw(ab + ".qkv", &wq));
                EDV_TRY(make_t(ab + ".qkv", wq, C, 3 * C, C, nullptr));
                EDV_TRY(make_t_lin(ab + ".to_out.0"));
            }
            EDV_TRY(make_t_lin(tb + ".ff.net.0.proj"));
            EDV_TRY(make_t_lin(tb + ".ff.net.2"));
        }
        c->train_prepared = true;
        return 0;
    }
    // dX[M, K] = dY[M, N] W  through the NT GEMM with the cached transposed weight ("T." + key is [K, N])
    int dgemm(const float *dY, long long M, int N, const std::string &key, int K, float *dX, const float *R1 = nullptr) {
        const float *wt;
        EDV_TRY(packedw("T." + key, &wt));
        return linear(dY, M, N, wt, K, nullptr, dX, ACT_NONE, nullptr, R1);
    }
    int dconv3(const float *dY, int H, int W, int Cout_fwd, const std::string &p, int Cin_fwd, float *dX, const float *add = nullptr) {
        const float *wb;
        EDV_TRY(packedw("B." + p, &wb));
        return conv3(dY, H, W, Cout_fwd, wb, nullptr, Cin_fwd, 1, dX, false, ACT_NONE, add);
    }
    // weight + bias gradient of a trainable 3x3 convolution p (x: its input, dY: the gradient of its output), when the caller asked for them
    int conv_param_grads(const std::string &p, const float *x, const float *dY, int H, int W, int Cin, int Cout) {
        if (!c->grad_head) return 0;
        float *dw, *db, *ws;
        EDV_TRY(gradbuf(p + ".weight", (size_t)Cout * Cin * 9, &dw));
        EDV_TRY(gradbuf(p + ".bias", (size_t)Cout, &db));
        size_t need = conv3_wgrad_workspace(F, H, W, Cin, Cout);
        const size_t cs = colsum_workspace(Cout);
        need = need > cs ? need : cs;
        EDV_TRY(wsbuf("g.wgrad", need, &ws));
        EDV_TRY(conv3_wgrad(x, dY, dw, F, H, W, Cin, Cout, ws, need, false, st));
        EDV_TRY(colsum_rows(dY, nullptr, (long long)F * H * W, Cout, ws, need, db, false, st));
        c->launches += 4;
        return 0;
    }
    // weight + bias gradient of a 1x1 convolution to one channel: dW[c] = sum_p gz[p] o2[p, c], db = sum_p gz[p]
    int dot_param_grads(const std::string &p, const float *o2, const float *gz, long long npix, int C) {
        if (!c->grad_head) return 0;
        float *dw, *db, *ws;
        EDV_TRY(gradbuf(p + ".weight", (size_t)C, &dw));
        EDV_TRY(gradbuf(p + ".bias", 1, &db));
        const size_t need = colsum_workspace(C);
        EDV_TRY(wsbuf("g.wgrad1", need, &ws));
        EDV_TRY(colsum_rows(o2, gz, npix, C, ws, need, dw, false, st));
        EDV_TRY(colsum_rows(gz, nullptr, npix, 1, ws, need, db, false, st));
        c->launches += 4;
        return 0;
    }

    // motion module backward, in place on d [F, P, C] (dL/d output -> dL/d input)
    int motion_module_bwd(int m, float *d, int P, int C) {
        const std::string p = "head.motion_modules." + std::to_string(m) + ".temporal_transformer";
        const std::string tb = p + ".transformer_blocks.0";
        const std::string tg = "mm" + std::to_string(m) + ".";
        const long long M = (long long)F * P;
        float *dh, *t1, *t3, *t4, *t8, *sums;
        EDV_TRY(wsbuf("g.mm.dh", (size_t)M * C, &dh));
        EDV_TRY(wsbuf("g.mm.t1", (size_t)M * C, &t1));
        EDV_TRY(wsbuf("g.mm.t3", (size_t)M * 3 * C, &t3));
        EDV_TRY(wsbuf("g.mm.t4", (size_t)M * 4 * C, &t4));
        EDV_TRY(wsbuf("g.mm.t8", (size_t)M * 8 * C, &t8));
        EDV_TRY(wsbuf("g.mm.sums", (size_t)F * 32 * 2, &sums));
        const float *xin, *stats, *hsv[3], *qkvs[2], *ff1, *w;
        EDV_TRY(saved(tg + "xin", &xin));
        EDV_TRY(saved(tg + "stats", &stats));
        EDV_TRY(saved(tg + "h", &hsv[0]));
        EDV_TRY(saved(tg + "h1", &hsv[1]));
        EDV_TRY(saved(tg + "h2", &hsv[2]));
        EDV_TRY(saved(tg + "qkv0", &qkvs[0]));
        EDV_TRY(saved(tg + "qkv1", &qkvs[1]));
        EDV_TRY(saved(tg + "ff1", &ff1));
        EDV_TRY(dgemm(d, M, C, p + ".proj_out", C, dh));                   // x = xin + proj_out(h3)
        if (cfg.temporal_lora && cfg.lora_type != EDV_LORA_NONE && c->grad_temporal) {  // temporal LoRA on ff.net.2 (endodav.py:119-137)
            const float *ff2;
            EDV_TRY(saved(tg + "ff2", &ff2));
            EDV_TRY(lora_step(tb + ".ff.net.2", ff2, 4 * C, dh, C, M, cfg.lora_rank, (cfg.lora_type == EDV_LORA_LORA || cfg.lora_type == EDV_LORA_DASH) ? 2.0f : 1.0f, "", lora_ws, lora_ws_n));
        }
        EDV_TRY(dgemm(dh, M, C, tb + ".ff.net.2", 4 * C, t4));             // h3 = h2 + ff2 W2
        EDV_TRY(geglu_bwd(ff1, t4, t8, M, 4 * C, st));
        EDV_TRY(dgemm(t8, M, 8 * C, tb + ".ff.net.0.proj", C, t1));
        EDV_TRY(param(tb + ".ff_norm.weight", &w));
        EDV_TRY(layernorm_bwd(hsv[2], identity_map(), w, t1, identity_map(), dh, identity_map(), M, C, 1e-5f, true, st));
        for (int a = 1; a >= 0; --a) {
            const std::string ab = tb + ".attention_blocks." + std::to_string(a);
            EDV_TRY(dgemm(dh, M, C, ab + ".to_out.0", C, t1));             // h(a+1) = h(a) + to_out(att)
            EDV_TRY(attn_temporal_bwd(qkvs[a], t1, t3, B, T, P, C, 8, st));  // qkvs[a] holds the rotated q|k under pe="rope"
            if (cfg.pe_rope) {
                const float *rope;
                EDV_TRY(param(ab + ".freqs_cis", &rope, 3));
                EDV_TRY(rope_qk(t3, rope, B, T, P, C, true, st));
            }
            EDV_TRY(dgemm(t3, M, 3 * C, ab + ".qkv", C, t1));
            EDV_TRY(param(tb + ".norms." + std::to_string(a) + ".weight", &w));
            EDV_TRY(layernorm_bwd(hsv[a], identity_map(), w, t1, identity_map(), dh, identity_map(), M, C, 1e-5f, true, st));
        }
        EDV_TRY(dgemm(dh, M, C, p + ".proj_in", C, t1));
        EDV_TRY(param(p + ".norm.weight", &w));
        EDV_TRY(groupnorm_bwd(xin, stats, w, t1, sums, d, F, P, C, 32, true, st));
        c->launches += 8;
        return 0;
    }

    // FeatureFusionBlock backward: d_out [F,oh,ow,Fe] -> d_x (and d_skip when the block has a skip input), both [F,h,w,Fe]
    int fusion_bwd(int j, const float *d_out, const float *cur_or_x, const float *skip, int h, int w, int oh, int ow, float *d_x, float *d_skip) {
        const std::string p = "head.scratch.refinenet" + std::to_string(j);
        const std::string tg = "fu" + std::to_string(j) + ".";
        const size_t n = (size_t)F * h * w * Fe;
        const long long MP_ = (long long)F * h * w;
        float *a, *b2;
        EDV_TRY(wsbuf("g.fu.a", n, &a));
        EDV_TRY(wsbuf("g.fu.b", n, &b2));
        const float *t1a = nullptr, *t1b, *cur = cur_or_x;
        EDV_TRY(saved(tg + "t1b", &t1b));
        if (skip) {
            EDV_TRY(saved(tg + "t1a", &t1a));
            EDV_TRY(saved(tg + "s", &cur));
        }
        EDV_TRY(bilinear_bwd(d_out, a, F, h, w, Fe, oh, ow, false, st));                      // out = up(out_conv(t2))
        EDV_TRY(dgemm(a, MP_, Fe, p + ".out_conv", Fe, d_x));                                  // d_x <- d_t2 for now
        EDV_TRY(dconv3(d_x, h, w, Fe, p + ".resConfUnit2.conv2", Fe, a));                      // t2 = cur + conv2(relu(t1b))
        EDV_TRY(ew_bwd(a, t1b, nullptr, a, (long long)n, 2, st));
        EDV_TRY(dconv3(a, h, w, Fe, p + ".resConfUnit2.conv1", Fe, b2));                       // t1b = conv1(relu(cur))
        EDV_TRY(ew_bwd(b2, cur, d_x, d_x, (long long)n, 2, st));                               // d_cur = d_t2 + mask(cur) * .
        if (skip) {                                                                            // cur = x + skip + conv2a(relu(t1a))
            EDV_TRY(dconv3(d_x, h, w, Fe, p + ".resConfUnit1.conv2", Fe, a));
            EDV_TRY(ew_bwd(a, t1a, nullptr, a, (long long)n, 2, st));
            EDV_TRY(dconv3(a, h, w, Fe, p + ".resConfUnit1.conv1", Fe, b2));                   // t1a = conv1a(relu(skip))
            EDV_TRY(ew_bwd(b2, skip, d_x, d_skip, (long long)n, 2, st));
        }
        c->launches += 6;
        return 0;
    }

    int backward(const float *disp0, const float *const g[4]) {
        EDV_CHECK(c->train && c->have_saved, "edv_backward needs the activations of a forward run under edv_set_train(1): none are kept (no such forward yet, "
                                             "a backward already consumed them, or an inference forward on this context ran in between)");
        if (!c->train_prepared) EDV_TRY(prepare_train());
        B = c->F / c->T; T = c->T; F = c->F; ph = c->ph; pw = c->pw; P0 = ph * pw;
        c0 = cfg.include_cls_token ? 1 : 0;
        ntok = c->ntok;
        const long long MT = (long long)F * ntok, MP = (long long)F * P0;
        const int *oc = cfg.out_channels;
        const int h1 = 4 * ph, w1 = 4 * pw, h2 = 2 * ph, w2 = 2 * pw, h3 = ph, w3 = pw, h4 = (ph - 1) / 2 + 1, w4 = (pw - 1) / 2 + 1;
        const int h0 = 8 * ph, w0 = 8 * pw, ih = cfg.image_h, iw = cfg.image_w, Fh = Fe / 2;
        {   // the input-gradient GEMMs run on the caller's stream alone: stream-K region 0 of the forward's workspace, if there is one
            auto it = c->ws.find("skws");
            const bool have = it != c->ws.end() && it->second.p && c->skws_zeroed == it->second.p;
            skws = have ? it->second.p : nullptr;
            skws_floats = have ? gemm_workspace() : 0;
        }
        {   // one workspace for every LoRA-gradient call: encoder MLPs (M = F*ntok, D <-> 4D) and, with temporal_lora, ff.net.2
            size_t need = 4;
            if (cfg.lora_type != EDV_LORA_NONE) {
                need = lora_grads_workspace(MT, D, 4 * D, cfg.lora_rank);
                if (cfg.temporal_lora) {
                    const long long Ms[4] = {(long long)F * h3 * w3, (long long)F * h4 * w4, (long long)F * h3 * w3, (long long)F * h2 * w2};
                    const int Cs[4] = {oc[2], oc[3], Fe, Fe};
                    for (int m = 0; m < 4; ++m) {
                        const size_t n = lora_grads_workspace(Ms[m], 4 * Cs[m], Cs[m], cfg.lora_rank);
                        need = n > need ? n : need;
                    }
                }
            }
            EDV_TRY(wsbuf("g.lora", need, &lora_ws));
            lora_ws_n = need;
        }

        float *d_p1, *d_p2, *d_p3, *d_p4, *d_r[5];
        EDV_TRY(wsbuf("g.p1", (size_t)F * h0 * w0 * Fe, &d_p1));
        EDV_TRY(wsbuf("g.p2", (size_t)F * h1 * w1 * Fe, &d_p2));
        EDV_TRY(wsbuf("g.p3", (size_t)F * h2 * w2 * Fe, &d_p3));
        EDV_TRY(wsbuf("g.p4", (size_t)F * h3 * w3 * Fe, &d_p4));
        // HeadDepth k on path_(k+1) (endodav/layers.py:206-221, dpt_pyramid.py:103-109): gradient of the path, written to dst or added to it
        auto head_depth_bwd = [&](int k, int h, int w, const std::string &path, float *dst, bool add) -> int {
            const std::string hp = "head.conv_depth_" + std::to_string(k + 1) + ".head.", tg = "hd" + std::to_string(k) + ".";
            const long long px = (long long)F * h * w;
            const float *pk, *o1, *up, *o2, *dk, *w4;
            EDV_TRY(saved(path, &pk));
            EDV_TRY(saved(tg + "o1", &o1));
            EDV_TRY(saved(tg + "up", &up));
            EDV_TRY(saved(tg + "o2", &o2));
            EDV_TRY(saved(tg + "disp", &dk));
            (void)o1;
            float *d_o2, *d_up, *d_o1, *gz;
            EDV_TRY(wsbuf("g.o2", (size_t)F * 4 * h0 * w0 * 32, &d_o2));
            EDV_TRY(wsbuf("g.up", (size_t)F * 4 * h0 * w0 * Fh, &d_up));
            EDV_TRY(wsbuf("g.o1", (size_t)F * h0 * w0 * Fh, &d_o1));
            EDV_TRY(wsbuf("g.gz", (size_t)F * 4 * h0 * w0, &gz));
            EDV_TRY(param(hp + "4.weight", &w4));
            EDV_TRY(dot_channels_bwd(g[k], dk, w4, o2, d_o2, gz, px * 4, 32, cfg.inv_sigmoid ? 2 : 1, st));
            EDV_TRY(dot_param_grads(hp + "4", o2, gz, px * 4, 32));
            EDV_TRY(conv_param_grads(hp + "2", up, d_o2, 2 * h, 2 * w, Fh, 32));
            EDV_TRY(dconv3(d_o2, 2 * h, 2 * w, 32, hp + "2", Fh, d_up));
            EDV_TRY(bilinear_bwd(d_up, d_o1, F, h, w, Fh, 2 * h, 2 * w, false, st));
            EDV_TRY(conv_param_grads(hp + "0", pk, d_o1, h, w, Fe, Fh));
            EDV_TRY(dconv3(d_o1, h, w, Fh, hp + "0", Fe, dst, add ? dst : nullptr));
            c->launches += 2;
            return 0;
        };
        if (cfg.conv_head) {
            EDV_TRY(head_depth_bwd(0, h0, w0, "p1", d_p1, false));
        } else {
            // ---------------- VDA head: disp[k] = down(disp[k-1]); disp[0] = relu(dot(relu(conv2(up(conv1(p1)))))) ----
            int sh[4], sw[4];
            sh[0] = ih; sw[0] = iw;
            for (int k = 1; k < 4; ++k) { sh[k] = sh[k - 1] / 2; sw[k] = sw[k - 1] / 2; }
            float *gd[3];
            const float *g3 = g[3], *mask0 = disp0;
            if (cfg.out_sigmoid) {  // disp[k] = sigmoid(raw[k]) (dpt_pyramid.py:97-101): dL/d raw[k] = g[k] s (1 - s); the ReLU mask is the raw map's
                float *g3s;
                const float *sg;
                EDV_TRY(wsbuf("g.d3", (size_t)F * sh[3] * sw[3], &g3s));
                EDV_TRY(saved("hd.sg3", &sg));
                EDV_TRY(sigmoid_bwd(g[3], sg, g3s, (long long)F * sh[3] * sw[3], st));
                g3 = g3s;
                EDV_TRY(saved("hd.raw0", &mask0));
            }
            for (int k = 2; k >= 0; --k) {
                EDV_TRY(wsbuf("g.d" + std::to_string(k), (size_t)F * sh[k] * sw[k], &gd[k]));
                if (cfg.out_sigmoid) {
                    const float *sg;
                    EDV_TRY(saved("hd.sg" + std::to_string(k), &sg));
                    EDV_TRY(sigmoid_bwd(g[k], sg, gd[k], (long long)F * sh[k] * sw[k], st));
                } else {
                    EDV_TRY(copy_f32(g[k], gd[k], (long long)F * sh[k] * sw[k], st));
                }
                EDV_TRY(bilinear_bwd(k == 2 ? g3 : gd[k + 1], gd[k], F, sh[k], sw[k], 1, sh[k + 1], sw[k + 1], true, st));
            }
            float *d_o2, *d_up, *d_o1, *gz = nullptr;
            const float *o2, *w, *p1, *up;
            EDV_TRY(saved("hd.o2", &o2));
            EDV_TRY(wsbuf("g.o2", (size_t)F * ih * iw * 32, &d_o2));
            EDV_TRY(wsbuf("g.up", (size_t)F * ih * iw * Fh, &d_up));
            EDV_TRY(wsbuf("g.o1", (size_t)F * h0 * w0 * Fh, &d_o1));
            if (c->grad_head) EDV_TRY(wsbuf("g.gz", (size_t)F * ih * iw, &gz));  // --train_output_conv (endodav/layers.py:5-34)
            EDV_TRY(param("head.scratch.output_conv2.2.weight", &w));
            EDV_TRY(dot_channels_bwd(gd[0], mask0, w, o2, d_o2, gz, (long long)F * ih * iw, 32, 0, st));
            if (c->grad_head) {
                EDV_TRY(saved("hd.up", &up));
                EDV_TRY(saved("p1", &p1));
                EDV_TRY(dot_param_grads("head.scratch.output_conv2.2", o2, gz, (long long)F * ih * iw, 32));
                EDV_TRY(conv_param_grads("head.scratch.output_conv2.0", up, d_o2, ih, iw, Fh, 32));
            }
            EDV_TRY(dconv3(d_o2, ih, iw, 32, "head.scratch.output_conv2.0", Fh, d_up));
            EDV_TRY(bilinear_bwd(d_up, d_o1, F, h0, w0, Fh, ih, iw, false, st));
            if (c->grad_head) EDV_TRY(conv_param_grads("head.scratch.output_conv1", p1, d_o1, h0, w0, Fe, Fh));
            EDV_TRY(dconv3(d_o1, h0, w0, Fh, "head.scratch.output_conv1", Fe, d_p1));
        }

        // ---------------- fusion blocks and the two motion modules between them ----------------
        EDV_TRY(wsbuf("g.r1", (size_t)F * h1 * w1 * Fe, &d_r[1]));
        EDV_TRY(wsbuf("g.r2", (size_t)F * h2 * w2 * Fe, &d_r[2]));
        EDV_TRY(wsbuf("g.r3", (size_t)F * h3 * w3 * Fe, &d_r[3]));
        EDV_TRY(wsbuf("g.r4", (size_t)F * h4 * w4 * Fe, &d_r[4]));
        const float *r[5];
        for (int j = 1; j <= 4; ++j) EDV_TRY(saved("r" + std::to_string(j), &r[j]));
        EDV_TRY(fusion_bwd(1, d_p1, nullptr, r[1], h1, w1, h0, w0, d_p2, d_r[1]));
        if (cfg.conv_head) EDV_TRY(head_depth_bwd(1, h1, w1, "p2", d_p2, true));   // path_2 also feeds conv_depth_2
        EDV_TRY(fusion_bwd(2, d_p2, nullptr, r[2], h2, w2, h1, w1, d_p3, d_r[2]));
        if (cfg.conv_head) EDV_TRY(head_depth_bwd(2, h2, w2, "p3", d_p3, true));   // path_3 (after motion module 3) feeds conv_depth_3
        EDV_TRY(motion_module_bwd(3, d_p3, h2 * w2, Fe));
        EDV_TRY(fusion_bwd(3, d_p3, nullptr, r[3], h3, w3, h2, w2, d_p4, d_r[3]));
        if (cfg.conv_head) EDV_TRY(head_depth_bwd(3, h3, w3, "p4", d_p4, true));
        EDV_TRY(motion_module_bwd(2, d_p4, h3 * w3, Fe));
        EDV_TRY(fusion_bwd(4, d_p4, r[4], nullptr, h4, w4, h3, w3, d_r[4], nullptr));

        // ---------------- layerN_rn, motion modules 0/1, reassemble, projects -> gradient of the four taps ----------
        float *d_l[5], *d_pj, *d_tap[4], *d_tapcls[4] = {nullptr, nullptr, nullptr, nullptr};
        const int hs_[5] = {0, h1, h2, h3, h4}, ws_[5] = {0, w1, w2, w3, w4};
        for (int j = 1; j <= 4; ++j) {
            EDV_TRY(wsbuf("g.l" + std::to_string(j), (size_t)F * hs_[j] * ws_[j] * oc[j - 1], &d_l[j]));
            EDV_TRY(dconv3(d_r[j], hs_[j], ws_[j], Fe, "head.scratch.layer" + std::to_string(j) + "_rn", oc[j - 1], d_l[j]));
        }
        EDV_TRY(motion_module_bwd(0, d_l[3], h3 * w3, oc[2]));
        EDV_TRY(motion_module_bwd(1, d_l[4], h4 * w4, oc[3]));
        const bool res_grads = c->grad_res && cfg.residual_mask != 0;
        if ((!c->grad_encoder || cfg.lora_type == EDV_LORA_NONE) && !res_grads) {  // temporal-only phase: nothing trainable below the head
            c->have_saved = false;
            return 0;
        }
        {
            int mx = oc[0];
            for (int j = 1; j < 4; ++j) mx = oc[j] > mx ? oc[j] : mx;
            EDV_TRY(wsbuf("g.pj", (size_t)MP * mx, &d_pj));
        }
        for (int j = 0; j < 4; ++j) {
            EDV_TRY(wsbuf("g.tap" + std::to_string(j), (size_t)MP * D, &d_tap[j]));
            const float *src = d_pj;
            if (j < 2) {
                const int s = j == 0 ? 4 : 2;
                float *A;
                EDV_TRY(wsbuf("g.unsh", (size_t)MP * s * s * oc[j], &A));
                EDV_TRY(pixel_unshuffle(d_l[j + 1], A, F, ph, pw, oc[j], s, st));
                EDV_TRY(dgemm(A, MP, s * s * oc[j], "head.resize_layers." + std::to_string(j), oc[j], d_pj));
            } else if (j == 2) {
                src = d_l[3];
            } else {
                // stride-2 input gradient = stride-1 input-gradient convolution of the zero-inserted dY (MFMA path; the
                // direct kernel conv3x3_s2_bwd took 2.2 ms here and stays as the unit-test reference of this identity)
                float *z;
                EDV_TRY(wsbuf("g.dil", (size_t)MP * oc[3], &z));
                EDV_TRY(dilate2(d_l[4], z, F, ph, pw, oc[3], st));
                EDV_TRY(dconv3(z, ph, pw, oc[3], "head.resize_layers.3", oc[3], d_pj));
            }
            EDV_TRY(dgemm(src, MP, oc[j], "head.projects." + std::to_string(j), D, d_tap[j]));
            if (cfg.use_clstoken) {
                // projects[j] read GELU(W1 tap + (W2 cls + b)) (dpt_pyramid.py:54-57): through the GELU, W1 back to the patch rows,
                // the per-frame sums of the pre-activation gradient through W2 back to the frame's cls row of the tap
                const std::string rp = "head.readout_projects." + std::to_string(j) + ".0";
                const float *pre;
                float *dpre, *dfb, *part;
                EDV_TRY(saved("ro" + std::to_string(j) + ".pre", &pre));
                EDV_TRY(wsbuf("g.ro.dpre", (size_t)MP * D, &dpre));
                EDV_TRY(wsbuf("g.ro.dfb", (size_t)F * D, &dfb));
                EDV_TRY(wsbuf("g.ro.part", (size_t)TALL_SPLITS * D, &part));
                EDV_TRY(wsbuf("g.tapcls" + std::to_string(j), (size_t)F * D, &d_tapcls[j]));
                EDV_TRY(ew_bwd(d_tap[j], pre, nullptr, dpre, MP * D, 1, st));
                for (int f = 0; f < F; ++f) EDV_TRY(col_dot(dpre + (size_t)f * P0 * D, nullptr, P0, D, nullptr, part, dfb + (size_t)f * D, st));
                EDV_TRY(dgemm(dpre, MP, D, rp + ".w1", D, d_tap[j]));
                EDV_TRY(dgemm(dfb, F, D, rp + ".w2", D, d_tapcls[j]));
                c->launches += 3 + 2 * F;
            }
        }
        c->launches += 12;

        // ---------------- encoder ----------------
        float *dxt, *t1, *t3, *t4, *delta, *lws;
        EDV_TRY(wsbuf("g.xt", (size_t)MT * D, &dxt));
        EDV_TRY(wsbuf("g.e1", (size_t)MT * D, &t1));
        EDV_TRY(wsbuf("g.e3", (size_t)MT * 3 * D, &t3));
        EDV_TRY(wsbuf("g.e4", (size_t)MT * 4 * D, &t4));
        EDV_TRY(wsbuf("g.delta", (size_t)F * heads * ntok, &delta));
        float *abws = nullptr;
        const size_t abws_n = attn_spatial_bwd_workspace(F, ntok, heads);
        if (abws_n) EDV_TRY(wsbuf("g.attbws", abws_n, &abws));
        const int rank = cfg.lora_rank;
        const bool lora = cfg.lora_type != EDV_LORA_NONE && c->grad_encoder;
        const size_t lws_n = lora_ws_n;
        lws = lora_ws;
        EDV_HIP(hipMemsetAsync(dxt, 0, (size_t)MT * D * sizeof(float), st));
        // lora_alpha / r (endodav.py:108-117).  dash: the gradient of lora_A / lora_B is LoRA's in both phases -- past the warm-up the
        // extra term U_top diag(lora_index) Vt_top is part of the folded (frozen) weight the input gradients already use
        const float lscale = (cfg.lora_type == EDV_LORA_LORA || cfg.lora_type == EDV_LORA_DASH) ? 2.0f : 1.0f;
        const float *nw;
        EDV_TRY(param("pretrained.norm.weight", &nw));
        int tapj = 3;
        for (int i = depth - 1; i >= 0; --i) {
            const std::string bp = "pretrained.blocks." + std::to_string(i), is = "." + std::to_string(i);
            const float *x_in, *x_mid, *x_out, *xn2, *qkv, *att, *lse, *pre, *hid, *w2;
            EDV_TRY(saved("t.x." + std::to_string(i), &x_in));
            EDV_TRY(saved("t.x." + std::to_string(i + 1), &x_out));
            EDV_TRY(saved("t.xmid" + is, &x_mid));
            EDV_TRY(saved("t.xn2" + is, &xn2));
            EDV_TRY(saved("t.qkv" + is, &qkv));
            EDV_TRY(saved("t.att" + is, &att));
            EDV_TRY(saved("t.lse" + is, &lse));
            EDV_TRY(saved("t.pre" + is, &pre));
            EDV_TRY(saved("t.hid" + is, &hid));
            if (tapj >= 0 && cfg.taps[tapj] == i) {  // tap = norm(x_out) on the patch rows (vision_transformer.py:317-321)
                EDV_TRY(layernorm_bwd(x_out, RowMap{P0, ntok, c0}, nw, d_tap[tapj], identity_map(), dxt, RowMap{P0, ntok, c0}, MP, D, 1e-6f, true, st));
                if (cfg.use_clstoken)  // the readout's class-token input: the final norm of token 0 of every frame (vision_transformer.py:322-324)
                    EDV_TRY(layernorm_bwd(x_out, RowMap{1, ntok, 0}, nw, d_tapcls[tapj], identity_map(), dxt, RowMap{1, ntok, 0}, F, D, 1e-6f, true, st));
                --tapj;
            }
            if (cfg.residual_mask & (1u << i)) EDV_TRY(res_bottleneck_bwd(i, dxt));  // x_out = x' + residual_(x' patch rows)
            // x' = x_mid + ls2 * fc2(gelu(fc1(norm2(x_mid))))
            if (lora) EDV_TRY(lora_step(bp + ".mlp.fc2", hid, 4 * D, dxt, D, MT, rank, lscale, bp + ".ls2.gamma", lws, lws_n));
            EDV_TRY(dgemm(dxt, MT, D, bp + ".mlp.fc2", 4 * D, t4));
            EDV_TRY(ew_bwd(t4, pre, nullptr, t4, MT * 4 * D, 1, st));
            if (lora) EDV_TRY(lora_step(bp + ".mlp.fc1", xn2, D, t4, 4 * D, MT, rank, lscale, "", lws, lws_n));
            EDV_TRY(dgemm(t4, MT, 4 * D, bp + ".mlp.fc1", D, t1));
            EDV_TRY(param(bp + ".norm2.weight", &w2));
            EDV_TRY(layernorm_bwd(x_mid, identity_map(), w2, t1, identity_map(), dxt, identity_map(), MT, D, 1e-6f, true, st));
            if (i == 0) break;  // nothing trainable below block 0's MLP
            // x_mid = x_in + ls1 * proj(attn(qkv(norm1(x_in))))
            EDV_TRY(dgemm(dxt, MT, D, bp + ".attn.proj", D, t1));
            {
                Bracket b_(c, KC_ATTN_SPATIAL_BWD, st);  // both passes (dQ; dK, dV) + their combine launches: seven N x N x 64 products per head
                EDV_TRY(attn_spatial_bwd(qkv, att, t1, lse, delta, t3, F, ntok, heads, abws, abws_n, st));
            }
            EDV_TRY(dgemm(t3, MT, 3 * D, bp + ".attn.qkv", D, t1));
            EDV_TRY(param(bp + ".norm1.weight", &w2));
            EDV_TRY(layernorm_bwd(x_in, identity_map(), w2, t1, identity_map(), dxt, identity_map(), MT, D, 1e-6f, true, st));
            c->launches += 6;
        }
        c->have_saved = false;
        return 0;
    }
    // Linear_SSB (mylora/layers.py:396-430), y = gamma * (((x * a) W^T) * b + bias):  with z = (x * a) W^T and
    // u = (G * gamma * b) W:   db[n] = gamma[n] sum_m G[m,n] z[m,n],   da[k] = sum_m x[m,k] u[m,k].
    // Two extra GEMMs per linear (a and b may pass through zero, so neither is recovered by dividing y or dX).
    int ssb_step(const std::string &p, const float *X, int nin, const float *G, int nout, long long M, const std::string &gamma_name) {
        const float *W, *a, *b, *gam = nullptr;
        EDV_TRY(param(p + ".weight", &W, 2));
        EDV_TRY(param(p + ".lora_A", &a));
        EDV_TRY(param(p + ".lora_B", &b));
        if (!gamma_name.empty()) EDV_TRY(param(gamma_name, &gam));
        float *Wa, *gb, *Tu, *z, *u, *part, *da, *db;
        EDV_TRY(wsbuf("g.ssb.wa", (size_t)nout * nin, &Wa));
        EDV_TRY(wsbuf("g.ssb.tu", (size_t)nout * nin, &Tu));
        EDV_TRY(wsbuf("g.ssb.gb", (size_t)nout, &gb));
        EDV_TRY(wsbuf("g.ssb.z", (size_t)M * nout, &z));
        EDV_TRY(wsbuf("g.ssb.u", (size_t)M * nin, &u));
        EDV_TRY(wsbuf("g.ssb.part", (size_t)TALL_SPLITS * (nin > nout ? nin : nout), &part));
        EDV_TRY(gradbuf(p + ".lora_A", (size_t)nin, &da));
        EDV_TRY(gradbuf(p + ".lora_B", (size_t)nout, &db));
        EDV_TRY(ssb_prep(W, a, b, gam, Wa, gb, nout, nin, st));
        EDV_TRY(transpose_scale(W, nin, gb, Tu, nout, nin, st));          // Tu [nin, nout] = (gamma b W)^T
        EDV_TRY(linear(X, M, nin, Wa, nout, nullptr, z));                 // z = (x * a) W^T
        EDV_TRY(linear(G, M, nout, Tu, nin, nullptr, u));                 // u = (G gamma b) W
        EDV_TRY(col_dot(G, z, M, nout, gam, part, db, st));
        EDV_TRY(col_dot(X, u, M, nin, nullptr, part, da, st));
        c->launches += 6;
        return 0;
    }
    // gradients of the LoRA factors of one linear into c->grads["<p>.lora_A"] ... (mylora/layers.py:148-157, 384-393)
    int lora_step(const std::string &p, const float *X, int nin, const float *G, int nout, long long M, int r, float s, const std::string &gamma_name,
                  float *lws, size_t lws_n) {
        if (!has(p + ".lora_A")) return 0;
        if (cfg.lora_type == EDV_LORA_SSB) return ssb_step(p, X, nin, G, nout, M, gamma_name);
        const float *A, *Bm, *U = nullptr, *V = nullptr, *gam = nullptr;
        EDV_TRY(param(p + ".lora_A", &A));
        EDV_TRY(param(p + ".lora_B", &Bm));
        if (cfg.lora_type == EDV_LORA_DVLORA) {
            EDV_TRY(param(p + ".lora_U", &U));
            EDV_TRY(param(p + ".lora_V", &V));
        }
        if (!gamma_name.empty()) EDV_TRY(param(gamma_name, &gam));
        float *dA, *dB, *dU = nullptr, *dV = nullptr;
        EDV_TRY(gradbuf(p + ".lora_A", (size_t)r * nin, &dA));
        EDV_TRY(gradbuf(p + ".lora_B", (size_t)r * nout, &dB));
        if (U) {
            EDV_TRY(gradbuf(p + ".lora_U", (size_t)r, &dU));
            EDV_TRY(gradbuf(p + ".lora_V", (size_t)nout, &dV));
        }
        c->launches += 8;
        EDV_TRY(lora_grads(X, nin, G, nout, M, nin, nout, r, A, Bm, U, V, s, gam, lws, lws_n, dA, dB, dU, dV, st));
        if (cfg.lora_type == EDV_LORA_DASH && cfg.dash_active) {
            // DashLinear past its warm-up adds x (U_top diag(idx) Vt_top)^T (mylora/layers.py:580-582) and frees lora_index:
            // d idx[j] = sum_m ((G * gamma) U_top)[m, j] (x Vt_top^T)[m, j] -- two skinny products and a column dot
            const float *Ut, *Vt;
            EDV_TRY(param(p + ".weight_u_top", &Ut));
            EDV_TRY(param(p + ".weight_vt_top", &Vt));
            const int ri = (int)c->params[p + ".lora_index"].shape[0];
            float *utg, *t1, *t2, *part, *didx;
            EDV_TRY(wsbuf("g.dash.utg", (size_t)ri * nout, &utg));
            EDV_TRY(wsbuf("g.dash.t1", (size_t)M * ri, &t1));
            EDV_TRY(wsbuf("g.dash.t2", (size_t)M * ri, &t2));
            EDV_TRY(wsbuf("g.dash.part", (size_t)TALL_SPLITS * ri, &part));
            EDV_TRY(gradbuf(p + ".lora_index", (size_t)ri, &didx));
            EDV_TRY(transpose_scale(Ut, ri, gam, utg, nout, ri, st));  // [nout, ri] -> [ri, nout], rows scaled by gamma
            EDV_TRY(skinny_xwt(G, M, nout, nout, utg, ri, t1, st));
            EDV_TRY(skinny_xwt(X, M, nin, nin, Vt, ri, t2, st));
            EDV_TRY(col_dot(t1, t2, M, ri, nullptr, part, didx, st));
            c->launches += 5;
        }
        return 0;
    }
};

}  // namespace

// =============================================================================================
extern "C" {

int edv_abi_version(void) { return EDV_ABI_VERSION; }
const char *edv_last_error(void) { return edv::get_error(); }

int edv_create(const edv_config *cfg, edv_ctx **out) {
    EDV_CHECK(cfg && out, "null argument");
    EDV_CHECK(cfg->abi_version == EDV_ABI_VERSION, "ABI version mismatch");
    EDV_CHECK(cfg->embed_dim > 0 && cfg->num_heads > 0 && cfg->embed_dim == cfg->num_heads * 64, "head dim must be 64");
    EDV_CHECK(cfg->embed_dim <= 1024, "embed_dim > 1024 unsupported");
    EDV_CHECK(cfg->depth > 0, "depth");
    EDV_CHECK(cfg->image_h > 0 && cfg->image_w > 0 && cfg->image_h % 14 == 0 && cfg->image_w % 14 == 0,
              "image_shape must be a multiple of the 14-pixel patch");
    EDV_CHECK(cfg->num_frames > 0, "num_frames must be positive");  // dpt_temporal.py:34
    EDV_CHECK(cfg->features > 0 && cfg->features % 32 == 0, "features must be a multiple of 32 (GroupNorm(32) + 8 heads)");
    for (int j = 0; j < 4; ++j) EDV_CHECK(cfg->out_channels[j] > 0 && cfg->out_channels[j] % 4 == 0, "out_channels must be multiples of 4");
    EDV_CHECK(cfg->out_channels[2] % 32 == 0 && cfg->out_channels[3] % 32 == 0, "out_channels[2:] must be multiples of 32");
    EDV_CHECK(cfg->features <= 1024 && cfg->out_channels[2] <= 1024 && cfg->out_channels[3] <= 1024, "temporal width > 1024 unsupported");
    EDV_CHECK(cfg->lora_type >= 0 && cfg->lora_type <= EDV_LORA_DASH, "lora_type");
    EDV_CHECK(cfg->depth <= 32 && (cfg->depth == 32 || (cfg->residual_mask >> cfg->depth) == 0), "residual_mask names a block >= depth");
    EDV_CHECK(cfg->residual_mask == 0 || (cfg->embed_dim / 8) % 4 == 0, "residual blocks need embed_dim / 8 to be a multiple of 4");
    for (int j = 0; j < 4; ++j) EDV_CHECK(cfg->taps[j] >= 0 && cfg->taps[j] < cfg->depth && (j == 0 || cfg->taps[j] > cfg->taps[j - 1]), "taps");
    *out = new edv_ctx();
    (*out)->cfg = *cfg;
    if (hipGetDevice(&(*out)->device) != hipSuccess) (*out)->device = -1;  // no device visible (host-only checks of the configuration)
    if (const char *e = getenv("EDV_ENC_STREAMS")) (*out)->enc_streams = (*out)->enc_streams_initial = atoi(e);
    if (const char *e = getenv("EDV_PRODUCTS")) {  // "f32" | "bf16x6": initial arithmetic of the encoder's linears (edv_set_products changes it)
        const std::string v(e);
        EDV_CHECK(v == "f32" || v == "bf16x6", "EDV_PRODUCTS must be f32 or bf16x6");
        (*out)->products = v == "bf16x6" ? EDV_PRODUCTS_BF16X6 : EDV_PRODUCTS_F32;
    }
    return 0;
}

int edv_destroy(edv_ctx *ctx) {
    if (!ctx) return 0;
    // free on the device the context lives on, whatever device the calling thread has current (nn.DataParallel destroys replicas'
    // contexts from the main thread), and give the caller its device back
    int cur = -1;
    (void)hipGetDevice(&cur);
    if (ctx->device >= 0 && cur != ctx->device) (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    for (auto &kv : ctx->packed)
        if (kv.second.p) (void)hipFree(kv.second.p);
    for (auto &kv : ctx->ws)
        if (kv.second.p) (void)hipFree(kv.second.p);
    for (auto &kv : ctx->grads)
        if (kv.second.p) (void)hipFree(kv.second.p);
    for (int h = 0; h < 4; ++h) {
        if (ctx->sub[h]) (void)hipStreamDestroy(ctx->sub[h]);
        if (ctx->ev_join[h]) (void)hipEventDestroy(ctx->ev_join[h]);
    }
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    for (int k = 0; k < 6; ++k)
        if (ctx->ev_x[k]) (void)hipEventDestroy(ctx->ev_x[k]);
    for (auto &p : ctx->prof)
        for (auto &e : p.ev) {
            (void)hipEventDestroy(e.first);
            (void)hipEventDestroy(e.second);
        }
    const int dev = ctx->device;
    delete ctx;
    if (dev >= 0 && cur >= 0 && cur != dev) (void)hipSetDevice(cur);
    return 0;
}

int edv_bind_param(edv_ctx *ctx, const char *name, const float *data_dev, const int64_t *shape, int32_t ndim) {
    EDV_CHECK(ctx && name && data_dev && (shape || ndim == 0) && ndim >= 0 && ndim <= 8, "bad argument");
    EDV_CHECK(((uintptr_t)data_dev % 16) == 0, std::string("parameter not 16-byte aligned: ") + name);
    Param p;
    p.p = data_dev;
    p.shape.assign(shape, shape + ndim);
    ctx->params[name] = p;
    ctx->prepared = false;
    return 0;
}

int edv_prepare(edv_ctx *ctx, void *stream) {
    EDV_CHECK(ctx, "null context");
    Run r(ctx, (hipStream_t)stream);
    return r.prepare();
}

int edv_refresh_lora(edv_ctx *ctx, void *stream) {
    EDV_CHECK(ctx, "null context");
    Run r(ctx, (hipStream_t)stream);
    return r.refresh_lora();
}

int edv_set_products(edv_ctx *ctx, int32_t products, void *stream) {
    EDV_CHECK(ctx, "null context");
    EDV_CHECK(products == EDV_PRODUCTS_F32 || products == EDV_PRODUCTS_BF16X6, "products: EDV_PRODUCTS_F32 or EDV_PRODUCTS_BF16X6");
    const bool turned_on = products == EDV_PRODUCTS_BF16X6 && ctx->products != EDV_PRODUCTS_BF16X6;
    ctx->products = products;
    if (turned_on && ctx->prepared) {  // planes are kept current only while the mode is on (edv_prepare / edv_refresh_lora): rebuild, never reuse
        ctx->x6.clear();
        Run r(ctx, (hipStream_t)stream);
        return r.build_x6(false);
    }
    return 0;
}

int edv_get_products(const edv_ctx *ctx) { return ctx ? ctx->products : -1; }

int edv_set_capture(edv_ctx *ctx, int on) {
    EDV_CHECK(ctx, "null context");
    ctx->capture = on != 0;
    return 0;
}

int edv_forward(edv_ctx *ctx, const float *x_dev, int32_t B, int32_t T, int32_t H, int32_t W, float *const disp_dev[4], void *stream) {
    EDV_CHECK(ctx && x_dev && disp_dev, "null argument");
    EDV_CHECK(ctx->prepared, "edv_prepare must run after binding parameters");
    EDV_CHECK(B > 0 && T > 0 && H > 1 && W > 1, "empty clip");
    // motion_module.py:197: the position table is sliced to T -> size mismatch beyond num_frames
    EDV_CHECK(T <= ctx->cfg.num_frames, "T exceeds num_frames (temporal_max_len)");
    // The reference takes any T <= num_frames (dpt_temporal.py:35-40, motion_module.py:180-198).  The temporal-attention kernels hold one pixel's
    // T x T scores on chip: the forward is built up to T = 64 (round 3), the backward up to 32 = the reference's own window length and
    // num_frames default (endodav.py:47, :62; its training scripts use T = 16)
    EDV_CHECK(T <= 64, "T > 64 frames per clip is not built");
    EDV_CHECK(!ctx->train || T <= 32, "a training forward with T > 32 frames per clip is not built (the backward of the temporal attention stops at 32)");

    EDV_CHECK((long long)B * T <= 65535, "too many frames in one call");
    for (int k = 0; k < 4; ++k) EDV_CHECK(disp_dev[k], "null output");
    if (ctx->train) {
        EDV_CHECK(!(ctx->cfg.use_bn), "the fine-tune step with use_bn=True is not built (train-mode BatchNorm uses batch statistics)");
        EDV_CHECK(!ctx->capture, "stage capture and training are exclusive");
    }
    Run r(ctx, (hipStream_t)stream);
    if (ctx->train) ++ctx->generation;  // the kept activations are about to be overwritten
    const int rc = r.forward(x_dev, B, T, H, W, disp_dev);
    if (rc && ctx->sub[0]) {
        // The error may have struck between a fork and its join: kernels already enqueued on the internal streams still write the
        // shared workspaces, and nothing makes the caller's stream wait for them.  Drain them before reporting, so that whatever the
        // caller enqueues next (another edv_forward on this context included) cannot overlap them.
        const std::string msg = edv::get_error();
        for (int h = 0; h < 4; ++h)
            if (ctx->sub[h]) (void)hipStreamSynchronize(ctx->sub[h]);
        edv::set_error(msg);
    }
    ctx->have_saved = rc == 0 && ctx->train;
    if (ctx->have_saved) ctx->saved_generation = ctx->generation;
    return rc;
}

int edv_generation(const edv_ctx *ctx, uint64_t *generation) {
    EDV_CHECK(ctx && generation, "null argument");
    *generation = ctx->generation;
    return 0;
}

int edv_set_train(edv_ctx *ctx, int32_t on) {
    EDV_CHECK(ctx, "null context");
    ctx->train = on != 0;
    if (!ctx->train) ctx->have_saved = false;
    return 0;
}

int edv_set_encoder_streams(edv_ctx *ctx, int32_t n) {
    EDV_CHECK(ctx, "null context");
    EDV_CHECK(n >= -1 && n <= 4, "encoder streams: -1 (initial setting), 0 (automatic), 1 .. 4");
    ctx->enc_streams = n < 0 ? ctx->enc_streams_initial : n;
    return 0;
}

int edv_set_grad_scope(edv_ctx *ctx, int32_t encoder_factors, int32_t temporal_factors, int32_t head_convs, int32_t residual_blocks) {
    EDV_CHECK(ctx, "null context");
    ctx->grad_res = residual_blocks != 0;
    ctx->grad_encoder = encoder_factors != 0;
    ctx->grad_temporal = temporal_factors != 0;
    ctx->grad_head = head_convs != 0;
    return 0;
}

int edv_backward(edv_ctx *ctx, uint64_t generation, const float *disp0_dev, const float *const grad_disp_dev[4], void *stream) {
    EDV_CHECK(ctx && disp0_dev && grad_disp_dev, "null argument");
    for (int k = 0; k < 4; ++k) EDV_CHECK(grad_disp_dev[k], "null gradient");
    // One set of kept activations per context: a second training forward overwrites them, and a backward of the first forward's graph
    // would then silently differentiate the second clip.  The caller names the forward it is differentiating.
    EDV_CHECK(generation == 0 || !ctx->have_saved || generation == ctx->saved_generation,
              "edv_backward for training forward #" + std::to_string(generation) + ", but the kept activations are those of forward #" +
                  std::to_string(ctx->saved_generation) + ": a later grad-enabled forward on this context overwrote them (one backward per forward)");
    for (auto &kv : ctx->flat) kv.second.written = false;
    Run r(ctx, (hipStream_t)stream);
    const int rc = r.backward(disp0_dev, grad_disp_dev);
    if (rc) return rc;
    for (auto &kv : ctx->flat)
        EDV_CHECK(kv.second.written, "the flat gradient buffer lists " + kv.first + ", but this backward produced no gradient for it (edv_set_grad_scope)");
    return 0;
}

int edv_grad_bind_flat(edv_ctx *ctx, int32_t n, const char *const *names, const int64_t *numels, float *flat_dev, int64_t flat_floats, int64_t *offsets_out) {
    EDV_CHECK(ctx && n >= 0 && (n == 0 || (names && numels && offsets_out)), "bad argument");
    int64_t off = 0;
    for (int i = 0; i < n; ++i) {
        EDV_CHECK(names[i] && numels[i] > 0, "bad slice");
        offsets_out[i] = off;
        off += (numels[i] + 3) & ~(int64_t)3;  // every slice starts on a 16-byte boundary (float4 stores of the reduction kernels)
    }
    if (n) offsets_out[n] = off;
    if (!flat_dev) return 0;  // layout query
    EDV_CHECK(flat_floats >= off && (uintptr_t)flat_dev % 16 == 0, "flat gradient buffer too small or not 16-byte aligned");
    ctx->flat.clear();
    for (int i = 0; i < n; ++i) {
        ctx->flat[names[i]] = edv_ctx::FlatSlot{flat_dev + offsets_out[i], (size_t)numels[i], false};
        // an owned buffer of an earlier unbound backward would otherwise keep answering edv_grad with a stale gradient once the name is unbound again
        auto old = ctx->grads.find(names[i]);
        if (old != ctx->grads.end()) {
            if (old->second.p) {
                ctx->bytes -= old->second.cap * sizeof(float);
                (void)hipFree(old->second.p);
            }
            ctx->grads.erase(old);
        }
    }
    return 0;
}

// Where the latest backward left the gradient of `name`: its slice of the caller's flat buffer when the name is bound (edv_grad_bind_flat), the
// context-owned buffer otherwise.  A bound name never falls through to an owned buffer of an earlier, unbound backward (binding erases those).
static int find_grad(edv_ctx *ctx, const char *name, float **p, size_t *n) {
    auto f = ctx->flat.find(name);
    if (f != ctx->flat.end()) {
        EDV_CHECK(f->second.written, std::string("no gradient for ") + name + " yet: it is bound to the flat buffer and no backward has written it");
        *p = f->second.p;
        *n = f->second.numel;
        return 0;
    }
    auto it = ctx->grads.find(name);
    EDV_CHECK(it != ctx->grads.end() && it->second.p, std::string("no gradient for ") + name);
    *p = it->second.p;
    *n = it->second.cap;
    return 0;
}

int edv_grad_copy(edv_ctx *ctx, const char *name, float *dst_dev, int64_t numel, void *stream) {
    EDV_CHECK(ctx && name && dst_dev, "null argument");
    float *p;
    size_t n;
    EDV_TRY(find_grad(ctx, name, &p, &n));
    EDV_CHECK(numel > 0 && (size_t)numel <= n, std::string("gradient size mismatch for ") + name);
    return copy_f32(p, dst_dev, numel, (hipStream_t)stream);
}

int edv_grad(edv_ctx *ctx, const char *name, float **grad_dev, int64_t *numel) {
    EDV_CHECK(ctx && name && grad_dev && numel, "null argument");
    size_t n;
    EDV_TRY(find_grad(ctx, name, grad_dev, &n));
    *numel = (int64_t)n;
    return 0;
}

int edv_output_shape(const edv_ctx *ctx, int32_t scale, int32_t *h, int32_t *w) {
    EDV_CHECK(ctx && h && w && scale >= 0 && scale < 4, "bad argument");
    const edv_config &c = ctx->cfg;
    if (!c.conv_head) {
        int sh = c.image_h, sw = c.image_w;
        for (int k = 0; k < scale; ++k) { sh /= 2; sw /= 2; }
        *h = sh; *w = sw;
    } else {
        const int ph = c.image_h / 14, pw = c.image_w / 14;
        const int mul[4] = {16, 8, 4, 2};
        *h = ph * mul[scale]; *w = pw * mul[scale];
    }
    return 0;
}

int edv_stage_copy(edv_ctx *ctx, const char *name, float *dst_dev, size_t *n, void *stream) {
    EDV_CHECK(ctx && name && n, "bad argument");
    if (std::string(name).rfind("ws:", 0) == 0) {  // any workspace buffer by name, whole capacity (gradient-stage parity tests)
        auto w = ctx->ws.find(std::string(name).substr(3));
        EDV_CHECK(w != ctx->ws.end() && w->second.p, std::string("no such workspace buffer: ") + name);
        *n = w->second.cap;
        if (dst_dev) return copy_f32(w->second.p, dst_dev, (long long)w->second.cap, (hipStream_t)stream);
        return 0;
    }
    auto it = ctx->stages.find(name);
    EDV_CHECK(it != ctx->stages.end(), std::string("stage not available (capture off or unknown): ") + name);
    *n = it->second.second;
    if (dst_dev) return copy_f32(it->second.first, dst_dev, (long long)it->second.second, (hipStream_t)stream);
    return 0;
}

int edv_profile_enable(edv_ctx *ctx, uint32_t class_mask) {
    EDV_CHECK(ctx, "null context");
    EDV_CHECK(class_mask < (1u << KC_COUNT), "unknown kernel class in mask");
    ctx->prof_mask = class_mask;
    for (auto &p : ctx->prof) p.used = 0;
    for (int k = 0; k < KC_COUNT; ++k) ctx->prof_flops[k] = ctx->prof_bytes[k] = 0.0;
    return 0;
}

int edv_profile_set_mask(edv_ctx *ctx, uint32_t class_mask) {
    EDV_CHECK(ctx, "null context");
    EDV_CHECK(class_mask < (1u << KC_COUNT), "unknown kernel class in mask");
    ctx->prof_mask = class_mask;  // nothing recorded so far is dropped
    return 0;
}

int edv_profile_read(edv_ctx *ctx, int32_t kernel_class, int32_t *launches, double *total_ms) {
    EDV_CHECK(ctx && launches && total_ms, "null argument");
    EDV_CHECK(kernel_class >= 0 && kernel_class < KC_COUNT, "unknown kernel class");
    EvPool &p = ctx->prof[kernel_class];
    double sum = 0.0;
    for (size_t i = 0; i < p.used; ++i) {
        EDV_HIP(hipEventSynchronize(p.ev[i].second));
        float ms = 0.f;
        EDV_HIP(hipEventElapsedTime(&ms, p.ev[i].first, p.ev[i].second));
        sum += ms;
    }
    *launches = (int32_t)p.used;
    *total_ms = sum;
    p.used = 0;
    return 0;
}

int edv_profile_work(edv_ctx *ctx, int32_t kernel_class, double *flops, double *bytes) {
    EDV_CHECK(ctx && flops && bytes, "null argument");
    EDV_CHECK(kernel_class >= 0 && kernel_class < KC_COUNT, "unknown kernel class");
    *flops = ctx->prof_flops[kernel_class];
    *bytes = ctx->prof_bytes[kernel_class];
    ctx->prof_flops[kernel_class] = ctx->prof_bytes[kernel_class] = 0.0;
    return 0;
}

size_t edv_device_bytes(const edv_ctx *ctx) { return ctx ? ctx->bytes : 0; }
int edv_last_launch_count(const edv_ctx *ctx) { return ctx ? ctx->launches : 0; }

}  // extern "C"
