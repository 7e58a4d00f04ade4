// Encoder self-attention, backward (flash-style: no N x N matrix in HBM).  Differentiates attn_spatial.hip /
// models/backbones/layers/attention.py:60-66:  O = softmax(S) V,  S = (Q d^-1/2) K^T.
//
// With the forward's per-row log-sum-exp L (base 2) and delta_q = sum_d dO[q,d] O[q,d]:
//     P  = exp2(S log2e - L)          dV = P^T dO          dP = dO V^T
//     dS = P * (dP - delta)           dQ = d^-1/2 dS K     dK = d^-1/2 dS^T Q
// Two launches of ONE kernel template, both in the forward's transposed MFMA orientation (resident rows on the lanes,
// streamed 32-row tiles through LDS, v_mfma_f32_32x32x2_f32, permuted-k fragments):
//   MODE_DQ : resident = 128 queries (Q, dO fragments in registers), streamed = key tiles (K, V)  -> dQ
//   MODE_DKV: resident = 128 keys    (K, V  fragments in registers), streamed = query tiles (Q, dO) -> dK, dV
// In both, X1^T = T1 R1^T (scores) and X2^T = T2 R2^T (dP) land as accumulators whose registers are exactly the B
// operand of the third product (rows of the streamed tile contract), as in the forward's P V step.
#include <cmath>
#include <cstdlib>

#include "ops.hpp"

namespace edv {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int HD = 64;
constexpr int TS = HD + 4;  // padded row stride of the streamed tiles: conflict-free b128 row reads and b32 column reads
constexpr int TR = 32;      // rows per streamed tile
enum { MODE_DQ = 0, MODE_DKV = 1 };

// delta[f, h, q] = sum_d dO[(f,q), h*64+d] * O[(f,q), h*64+d]: 16 lanes per (row, head)
__global__ __launch_bounds__(256) void attn_delta_kernel(const float *__restrict__ dO, const float *__restrict__ O, float *__restrict__ delta, int F, int N,
                                                         int heads) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long pair = gid >> 4;  // (row, head)
    const int sub = (int)(gid & 15);
    const long long total = (long long)F * N * heads;
    const long long pc = pair < total ? pair : total - 1;
    const long long row = pc / heads;
    const int head = (int)(pc - row * heads);
    const long long off = row * (long long)heads * HD + head * HD + sub * 4;
    const f32x4 a = *reinterpret_cast<const f32x4 *>(dO + off), b = *reinterpret_cast<const f32x4 *>(O + off);
    float s = (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w);
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (sub == 0 && pair < total) {
        const long long f = row / N;
        const int q = (int)(row - f * N);
        delta[(f * heads + head) * N + q] = s;
    }
}

// NW waves (32 resident rows each) per workgroup share the streamed tiles
// VALU diet (round 2; on this part the f32 MFMA and the VALU do not overlap: attn_spatial.hip, gemm_dma.hip): the score accumulators start
// at -L and the dP accumulators at -delta, so the MFMA itself delivers X1 - L and dP - delta (32 subtractions per tile gone); the
// sequence-end mask is a branch taken on the last tile only (it was 32 compare / select pairs on every tile); and all registers
// live in the VGPR file (__launch_bounds__(.., 2): the compiler had parked the output accumulators in AGPRs and moved them on
// every use).
template <int MODE, int NW>
__global__ __launch_bounds__(NW * 64, 2) void attn_spatial_bwd_kernel(const float *__restrict__ qkv, const float *__restrict__ dO, const float *__restrict__ lse,
                                                               const float *__restrict__ delta, float *__restrict__ dqkv, float *__restrict__ ws, int N,
                                                               int heads, int whole_rounds, long long units, int chunk) {
    __shared__ __attribute__((aligned(16))) float sT1[TR * TS];
    __shared__ __attribute__((aligned(16))) float sT2[TR * TS];
    __shared__ __attribute__((aligned(16))) float sL[TR];  // MODE_DKV: lse / delta of the streamed queries
    __shared__ __attribute__((aligned(16))) float sD[TR];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    constexpr int RB = NW * 32, NT = NW * 64;
    constexpr int OC = MODE == MODE_DQ ? HD : 2 * HD;  // output columns per resident row: dQ | dK, dV
    const int nb = (N + RB - 1) / RB;
    const int D = heads * HD, D3 = 3 * D;
    const int ntiles = (N + TR - 1) / TR;
    // Persistent workgroups, whole tasks first, the last partial round split along the STREAMED axis -- the scheduling of
    // attn_spatial.hip.  The gradients are plain sums over the streamed rows, so a piece just leaves its raw accumulators in
    // workspace slot (id * 2 + run) and attn_bwd_combine_kernel adds the pieces of a task.
    const int G = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = G >> 3, r = G & 7, x = bid & 7, loc = bid >> 3;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + loc;
    }
    const int task_l0 = whole_rounds * G;
    long long u = (long long)bid * chunk;
    const long long u_end = u + chunk < units ? u + chunk : units;
    int round = 0, seg = 0;
    for (;;) {
    int task, kt0, kt1;
    float *part = nullptr;
    if (round < whole_rounds) {
        task = round * G + bid;
        kt0 = 0;
        kt1 = ntiles;
        ++round;
    } else if (u < u_end) {
        const int t = (int)(u / ntiles);
        kt0 = (int)(u - (long long)t * ntiles);
        const long long left = u_end - u;
        kt1 = kt0 + left < ntiles ? kt0 + (int)left : ntiles;
        task = task_l0 + t;
        u += kt1 - kt0;
        if (!(kt0 == 0 && kt1 == ntiles)) part = ws + ((long long)bid * 2 + seg) * (RB * OC);
        ++seg;
    } else {
        break;
    }
    const int blk = task % nb, fh = task / nb;
    const int head = fh % heads, frame = fh / heads;
    const float *base = qkv + (long long)frame * N * D3 + head * HD;    // q columns; k at +D, v at +2D
    const float *dobase = dO + (long long)frame * N * D + head * HD;
    const float *lrow = lse + ((long long)frame * heads + head) * N;
    const float *drow = delta + ((long long)frame * heads + head) * N;

    const int ri = blk * RB + wave * 32 + l31;  // resident row of this lane (query in MODE_DQ, key in MODE_DKV)
    const int rrow = ri < N ? ri : N - 1;
    const float c1 = 0.125f * 1.44269504088896340736f;  // d^-1/2 * log2(e)

    // resident fragments in permuted-k order: element e of f[qq] is d = 8*qq + 4*lh + e
    f32x4 r1f[8], r2f[8];
    {
        const float *p1 = base + (long long)rrow * D3 + (MODE == MODE_DQ ? 0 : D);
        const float *p2 = MODE == MODE_DQ ? dobase + (long long)rrow * D : base + (long long)rrow * D3 + 2 * D;
#pragma unroll
        for (int qq = 0; qq < 8; ++qq) {
            r1f[qq] = *reinterpret_cast<const f32x4 *>(p1 + 8 * qq + 4 * lh) * c1;
            r2f[qq] = *reinterpret_cast<const f32x4 *>(p2 + 8 * qq + 4 * lh);
        }
    }
    float Lq = 0.f, Dq = 0.f;
    if (MODE == MODE_DQ) {
        Lq = lrow[rrow];
        Dq = drow[rrow];
    }

    f32x16 o0, o1, o2, o3;  // MODE_DQ: dQ^T (o0, o1);  MODE_DKV: dK^T (o0, o1), dV^T (o2, o3)
#pragma unroll
    for (int r = 0; r < 16; ++r) o0[r] = o1[r] = o2[r] = o3[r] = 0.f;

    // staging: 32 rows x 16 float4 per array = 512 float4 per array; NT threads -> SP passes of SR rows
    constexpr int SLOTS = TR * 16, SP = (SLOTS + NT - 1) / NT;  // float4 slots per array, passes (slot = tid + NT * pass)
    f32x4 pa[SP], pb[SP];
    float pl = 0.f, pd = 0.f;
    auto load_tile = [&](int t0) {
#pragma unroll
        for (int i = 0; i < SP; ++i) {
            const int slot = tid + NT * i;
            if (SLOTS % NT != 0 && slot >= SLOTS) break;
            const int sc = slot & 15;
            int tr = t0 + (slot >> 4);
            tr = tr < N ? tr : N - 1;  // clamped rows are masked below
            if (MODE == MODE_DQ) {
                const float *p = base + (long long)tr * D3 + sc * 4;
                pa[i] = *reinterpret_cast<const f32x4 *>(p + D);      // K
                pb[i] = *reinterpret_cast<const f32x4 *>(p + 2 * D);  // V
            } else {
                pa[i] = *reinterpret_cast<const f32x4 *>(base + (long long)tr * D3 + sc * 4);  // Q
                pb[i] = *reinterpret_cast<const f32x4 *>(dobase + (long long)tr * D + sc * 4);  // dO
            }
        }
        if (MODE == MODE_DKV && tid < TR) {
            int tr = t0 + tid;
            tr = tr < N ? tr : N - 1;
            pl = lrow[tr];
            pd = drow[tr];
        }
    };

    load_tile(kt0 * TR);
    for (int t = kt0; t < kt1; ++t) {
        const int t0 = t * TR;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < SP; ++i) {
            const int slot = tid + NT * i;
            if (SLOTS % NT != 0 && slot >= SLOTS) break;
            *reinterpret_cast<f32x4 *>(&sT1[(slot >> 4) * TS + (slot & 15) * 4]) = pa[i];
            *reinterpret_cast<f32x4 *>(&sT2[(slot >> 4) * TS + (slot & 15) * 4]) = pb[i];
        }
        if (MODE == MODE_DKV && tid < TR) {
            sL[tid] = pl;
            sD[tid] = pd;
        }
        __syncthreads();
        if (t + 1 < kt1) load_tile(t0 + TR);

        // ---- X1^T - L = T1 R1^T - L (scores, base-2 logits), X2^T - delta = T2 R2^T - delta (dP): L / delta enter as the accumulators' start
        // values.  Register r of lane-half h is streamed row (r&3) + 8*(r>>2) + 4*h; the resident row sits on the lane.
        f32x16 x1, x2;
        if (MODE == MODE_DQ) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                x1[r] = -Lq;
                x2[r] = -Dq;
            }
        } else {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 L4 = *reinterpret_cast<const f32x4 *>(&sL[8 * g + 4 * lh]);
                const f32x4 D4 = *reinterpret_cast<const f32x4 *>(&sD[8 * g + 4 * lh]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    x1[4 * g + e] = -L4[e];
                    x2[4 * g + e] = -D4[e];
                }
            }
        }
#pragma unroll
        for (int qq = 0; qq < 8; ++qq) {
            const f32x4 a1 = *reinterpret_cast<const f32x4 *>(&sT1[l31 * TS + 8 * qq + 4 * lh]);
            const f32x4 a2 = *reinterpret_cast<const f32x4 *>(&sT2[l31 * TS + 8 * qq + 4 * lh]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                x1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], r1f[qq][e], x1, 0, 0, 0);
                x2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[e], r2f[qq][e], x2, 0, 0, 0);
            }
        }
        if (t0 + TR > N) {  // streamed rows past the sequence end (last tile only; the empty asm keeps this a real branch): P = exp2(-inf) = 0
            asm volatile("" ::: "memory");
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (t0 + (r & 3) + 8 * (r >> 2) + 4 * lh >= N) x1[r] = -INFINITY;
        }
        // ---- P = exp2(X1 - L), dS = P (dP - delta)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            x1[r] = __builtin_amdgcn_exp2f(x1[r]);  // P
            x2[r] = x1[r] * x2[r];                  // dS
        }
        // ---- third products: streamed rows contract; step r uses rows {row(r,0), row(r,1)} in its two k-slots
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int trl = (r & 3) + 8 * (r >> 2) + 4 * lh;
            const float a0 = sT1[trl * TS + l31], a1 = sT1[trl * TS + 32 + l31];
            o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, x2[r], o0, 0, 0, 0);  // dQ^T += K^T dS^T   |  dK^T += Q^T dS
            o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, x2[r], o1, 0, 0, 0);
            if (MODE == MODE_DKV) {
                const float b0 = sT2[trl * TS + l31], b1 = sT2[trl * TS + 32 + l31];
                o2 = __builtin_amdgcn_mfma_f32_32x32x2f32(b0, x1[r], o2, 0, 0, 0);  // dV^T += dO^T P
                o3 = __builtin_amdgcn_mfma_f32_32x32x2f32(b1, x1[r], o3, 0, 0, 0);
            }
        }
    }

    if (part) {  // raw accumulators of a piece: row-major [RB][OC], scaled by the combine kernel
        float *prow = part + (wave * 32 + l31) * OC;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            *reinterpret_cast<f32x4 *>(prow + 8 * g + 4 * lh) = f32x4{o0[4 * g], o0[4 * g + 1], o0[4 * g + 2], o0[4 * g + 3]};
            *reinterpret_cast<f32x4 *>(prow + 32 + 8 * g + 4 * lh) = f32x4{o1[4 * g], o1[4 * g + 1], o1[4 * g + 2], o1[4 * g + 3]};
            if (MODE == MODE_DKV) {
                *reinterpret_cast<f32x4 *>(prow + HD + 8 * g + 4 * lh) = f32x4{o2[4 * g], o2[4 * g + 1], o2[4 * g + 2], o2[4 * g + 3]};
                *reinterpret_cast<f32x4 *>(prow + HD + 32 + 8 * g + 4 * lh) = f32x4{o3[4 * g], o3[4 * g + 1], o3[4 * g + 2], o3[4 * g + 3]};
            }
        }
    } else if (ri < N) {
        float *orow = dqkv + ((long long)frame * N + ri) * D3 + head * HD + (MODE == MODE_DQ ? 0 : D);
#pragma unroll
        for (int g = 0; g < 4; ++g) {  // registers 4g..4g+3 are d = 8g + 4*lh + {0..3}
            const f32x4 a = {o0[4 * g], o0[4 * g + 1], o0[4 * g + 2], o0[4 * g + 3]};
            const f32x4 b = {o1[4 * g], o1[4 * g + 1], o1[4 * g + 2], o1[4 * g + 3]};
            *reinterpret_cast<f32x4 *>(orow + 8 * g + 4 * lh) = a * 0.125f;
            *reinterpret_cast<f32x4 *>(orow + 32 + 8 * g + 4 * lh) = b * 0.125f;
            if (MODE == MODE_DKV) {
                const f32x4 c = {o2[4 * g], o2[4 * g + 1], o2[4 * g + 2], o2[4 * g + 3]};
                const f32x4 d = {o3[4 * g], o3[4 * g + 1], o3[4 * g + 2], o3[4 * g + 3]};
                *reinterpret_cast<f32x4 *>(orow + D + 8 * g + 4 * lh) = c;
                *reinterpret_cast<f32x4 *>(orow + D + 32 + 8 * g + 4 * lh) = d;
            }
        }
    }
    }  // work list
}

// Adds the pieces of every split task and writes dQ (x d^-1/2) or dK (x d^-1/2) | dV.  Thread = (row, float4 of columns).
template <int MODE>
__global__ __launch_bounds__(256) void attn_bwd_combine_kernel(const float *__restrict__ ws, float *__restrict__ dqkv, int N, int heads, int RB, int ntiles,
                                                               int task_l0, int chunk) {
    constexpr int OC = MODE == MODE_DQ ? HD : 2 * HD;
    const int t = blockIdx.x;
    const long long ub = (long long)t * ntiles, ue = ub + ntiles;
    const int g0 = (int)(ub / chunk), g1 = (int)((ue - 1) / chunk);
    if (g0 == g1 && (long long)g0 * chunk <= ub && (long long)(g0 + 1) * chunk >= ue) return;  // ran whole
    const int idx = blockIdx.y * 256 + threadIdx.x;  // float4 slot within [RB][OC]
    if (idx >= RB * OC / 4) return;
    const int row = idx / (OC / 4), c4 = idx - row * (OC / 4);
    const int nb = (N + RB - 1) / RB;
    const int task = task_l0 + t;
    const int blk = task % nb, fh = task / nb;
    const int head = fh % heads, frame = fh / heads;
    const int ri = blk * RB + row;
    if (ri >= N) return;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int g = g0; g <= g1; ++g) {
        const float *part = ws + ((long long)g * 2 + ((long long)g * chunk >= ub ? 0 : 1)) * (RB * OC);
        acc += *reinterpret_cast<const f32x4 *>(part + row * OC + 4 * c4);
    }
    const int D = heads * HD, D3 = 3 * D;
    const int col = 4 * c4;  // MODE_DQ: dq column; MODE_DKV: [0, 64) dk, [64, 128) dv
    float *orow = dqkv + ((long long)frame * N + ri) * D3 + head * HD;
    if (MODE == MODE_DQ)
        *reinterpret_cast<f32x4 *>(orow + col) = acc * 0.125f;
    else if (col < HD)
        *reinterpret_cast<f32x4 *>(orow + D + col) = acc * 0.125f;
    else
        *reinterpret_cast<f32x4 *>(orow + 2 * D + (col - HD)) = acc;
}

struct BwdPlan {
    int grid, whole_rounds, chunk, leftover, ntiles;
    long long units;
    size_t ws_floats;
};

template <int MODE, int NW>
int bwd_slots() {
    static DeviceSlotCache cache;
    return cache.get([] {
        int dev = 0, cus = 0, per_cu = 0;
        if (hipGetDevice(&dev) != hipSuccess) return 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, attn_spatial_bwd_kernel<MODE, NW>, NW * 64, 0) != hipSuccess) return 0;
        return cus * per_cu;
    });
}

BwdPlan make_bwd_plan(long long ntasks, int N, int slots, int rb, int oc, bool plain) {
    BwdPlan p;
    p.ntiles = (N + TR - 1) / TR;
    if (plain || slots <= 0) {
        p.grid = (int)ntasks; p.whole_rounds = 1; p.leftover = 0; p.units = 0; p.chunk = 1; p.ws_floats = 0;
        return p;
    }
    p.whole_rounds = (int)(ntasks / slots);
    p.leftover = (int)(ntasks - (long long)p.whole_rounds * slots);
    p.units = (long long)p.leftover * p.ntiles;
    p.chunk = p.units ? (int)((p.units + slots - 1) / slots) : 1;
    p.grid = p.whole_rounds ? slots : (int)((p.units + p.chunk - 1) / p.chunk);
    p.ws_floats = (size_t)((p.units + p.chunk - 1) / p.chunk) * 2 * (size_t)rb * oc;
    return p;
}

}  // namespace

namespace {
bool bwd_plain() {
    static const bool plain = [] {
        const char *e = getenv("EDV_ATTN_BWD_PLAIN");  // 1: one workgroup per task (A/B runs)
        return e && atoi(e) != 0;
    }();
    return plain;
}
}  // namespace

size_t attn_spatial_bwd_workspace(int F, int N, int heads) {
    if (F <= 0 || N <= 0 || heads <= 0) return 0;
    const long long ntasks = (long long)F * heads * ((N + 127) / 128);
    const BwdPlan a = make_bwd_plan(ntasks, N, bwd_slots<MODE_DQ, 4>(), 128, HD, bwd_plain());
    const BwdPlan b = make_bwd_plan(ntasks, N, bwd_slots<MODE_DKV, 4>(), 128, 2 * HD, bwd_plain());
    return a.ws_floats > b.ws_floats ? a.ws_floats : b.ws_floats;
}

// qkv [F*N, 3*heads*64], out/dout [F*N, heads*64], lse/delta [F, heads, N] (delta is scratch written here), dqkv like qkv;
// ws: attn_spatial_bwd_workspace(F, N, heads) floats (the two launches use it one after the other)
int attn_spatial_bwd(const float *qkv, const float *out, const float *dout, const float *lse, float *delta, float *dqkv, int F, int N, int heads, float *ws,
                     size_t ws_floats, hipStream_t st) {
    EDV_CHECK(qkv && out && dout && lse && delta && dqkv, "null operand");
    EDV_CHECK(F > 0 && N > 0 && heads > 0, "empty problem");
    EDV_CHECK(((uintptr_t)qkv % 16 == 0) && ((uintptr_t)out % 16 == 0) && ((uintptr_t)dout % 16 == 0) && ((uintptr_t)dqkv % 16 == 0), "16-byte alignment");
    const long long pairs = (long long)F * N * heads;
    EDV_CHECK((pairs * 16 + 255) / 256 < (1ll << 31), "grid");
    EDV_LAUNCH(attn_delta_kernel, dim3((unsigned)((pairs * 16 + 255) / 256)), dim3(256), 0, st, dout, out, delta, F, N, heads);
    EDV_LAUNCH_OK();
    const long long ntasks = (long long)F * heads * ((N + 127) / 128);
    EDV_CHECK(ntasks < (1ll << 31), "grid");
    // 4 waves per workgroup share each streamed tile (2- and 3-wave workgroups measured slower: 1046 / 962 vs 947 us at T=8,
    // 3012 / 3622 vs 2731 us at T=32).
    const BwdPlan pq = make_bwd_plan(ntasks, N, bwd_slots<MODE_DQ, 4>(), 128, HD, bwd_plain());
    const BwdPlan pk = make_bwd_plan(ntasks, N, bwd_slots<MODE_DKV, 4>(), 128, 2 * HD, bwd_plain());
    const size_t need = pq.ws_floats > pk.ws_floats ? pq.ws_floats : pk.ws_floats;
    EDV_CHECK(need == 0 || (ws && ws_floats >= need && (uintptr_t)ws % 16 == 0), "attention backward workspace too small (attn_spatial_bwd_workspace)");
    EDV_LAUNCH((attn_spatial_bwd_kernel<MODE_DQ, 4>), dim3((unsigned)pq.grid), dim3(256), 0, st, qkv, dout, lse, delta, dqkv, ws, N, heads,
                       pq.whole_rounds, pq.units, pq.chunk);
    EDV_LAUNCH_OK();
    if (pq.leftover) {
        EDV_LAUNCH(attn_bwd_combine_kernel<MODE_DQ>, dim3((unsigned)pq.leftover, (128 * HD / 4 + 255) / 256), dim3(256), 0, st, ws, dqkv, N, heads, 128,
                           pq.ntiles, pq.whole_rounds * pq.grid, pq.chunk);
        EDV_LAUNCH_OK();
    }
    EDV_LAUNCH((attn_spatial_bwd_kernel<MODE_DKV, 4>), dim3((unsigned)pk.grid), dim3(256), 0, st, qkv, dout, lse, delta, dqkv, ws, N, heads,
                       pk.whole_rounds, pk.units, pk.chunk);
    EDV_LAUNCH_OK();
    if (pk.leftover) {
        EDV_LAUNCH(attn_bwd_combine_kernel<MODE_DKV>, dim3((unsigned)pk.leftover, (128 * 2 * HD / 4 + 255) / 256), dim3(256), 0, st, ws, dqkv, N, heads,
                           128, pk.ntiles, pk.whole_rounds * pk.grid, pk.chunk);
        EDV_LAUNCH_OK();
    }
    return 0;
}

}  // namespace edv
