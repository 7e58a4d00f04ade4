// Encoder self-attention core, fp32, flash-style (no N x N score matrix in HBM).
// Restates models/backbones/layers/attention.py:60-66:  softmax((q * d^-0.5) kᵀ) v, d = 64.
//
// Layout: qkv [F*N, 3*heads*64] exactly as the qkv linear writes it (columns [3][heads][64]);
// out [F*N, heads*64] (= attn.transpose(1,2).reshape(B,N,C) of the reference).
//
// Workgroup = 4 waves = 128 queries of one (frame, head); each wave owns 32 queries and sweeps
// the keys in tiles of 64 staged through LDS (shared by the 4 waves).  All products run on
// v_mfma_f32_32x32x2_f32 in the TRANSPOSED orientation, so the query index sits on the lane:
//     Sᵀ[key, q]  = K · Qᵀ      A = K tile (LDS, b128 reads with the permuted-k trick of gemm.hip)
//                               B = Qᵀ (registers for the whole kernel, pre-scaled by d^-0.5 * log2 e)
//     Oᵀ[d, q]   += Vᵀ · Pᵀ     A = Vᵀ (LDS, one b32 per MFMA),  B = Pᵀ = the Sᵀ accumulator registers
// With queries on lanes the softmax row reductions are 31 in-lane max/adds plus ONE cross-half
// shuffle, the running max / sum / rescale factor are per-lane scalars, and the probabilities
// feed the second product straight from the accumulator registers: the accumulator's row map
// key = (r&3) + 8*(r>>2) + 4*(lane>>5) assigns register r of lane-half h exactly the k-slot h
// of MFMA step r, so no data moves between the two products.
#include <cmath>
#include <cstdlib>

#include "ops.hpp"

namespace edv {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int HD = 64;       // head dim
constexpr int KS = HD + 4;   // padded K row stride: 68r mod 64 = 4r -> conflict-free b128
// NW waves per workgroup (NW*32 queries share the staged K/V tiles).  Fewer waves per workgroup means more,
// smaller workgroups: better balance over the 256 CUs when frames*heads*ceil(N/128) is only ~2 per CU.
// KT keys per LDS tile (32 or 64).  32-key tiles halve the prefetch and score registers (159 instead of 198 per lane:
// 3 instead of 2 waves per SIMD) but measured no faster, so 64 stays the default; EDV_ATTN_KT=32 selects the other.
template <int NW, int KT>
__global__ __launch_bounds__(NW * 64) void attn_spatial_kernel(const float *__restrict__ qkv, float *__restrict__ out, int N, int heads) {
    constexpr int QB = NW * 32;
    constexpr int NT = NW * 64;
    __shared__ __attribute__((aligned(16))) float smem[KT * KS + KT * HD];
    float *sK = smem;
    float *sV = smem + KT * KS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    // XCD-aware block -> (frame, head, query tile) map.  Workgroups b and b+8 share an XCD and its L2; the query
    // tiles of one (frame, head) all sweep the same K/V, so hand each XCD a contiguous run of logical tiles
    // (bijective for any grid size).  Measured: L2->fabric reads per launch 290 MB -> see profiles/.
    const int nq = (N + QB - 1) / QB;
    int bid = blockIdx.x;
    {
        const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7, x = bid & 7, loc = bid >> 3;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + loc;
    }
    const int qt = bid % nq, fh = bid / nq;
    const int head = fh % heads, frame = fh / heads;
    const int D = heads * HD, D3 = 3 * D;
    const float *base = qkv + (long long)frame * N * D3 + head * HD;

    const int qi = qt * QB + wave * 32 + l31;
    const int qrow = qi < N ? qi : N - 1;

    // Q fragment in permuted-k order: element e of qf[qq] is d = 8*qq + 4*lh + e
    const float qscale = 0.125f * 1.44269504088896340736f;  // d^-0.5 (exact) * log2(e)
    f32x4 qf[8];
#pragma unroll
    for (int qq = 0; qq < 8; ++qq) {
        f32x4 v = *reinterpret_cast<const f32x4 *>(base + (long long)qrow * D3 + 8 * qq + 4 * lh);
        qf[qq] = v * qscale;
    }

    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) o0[r] = o1[r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    // staging slots: 16 float4 per 64-float row; thread -> (row sr + RS*i, chunk sc)
    constexpr int RS = NT / 16, RI = KT / RS;  // rows per pass, passes
    const int sc = tid & 15, sr = tid >> 4;
    f32x4 rk[RI], rv[RI];
    auto load_tile = [&](int k0) {
#pragma unroll
        for (int i = 0; i < RI; ++i) {
            int kr = k0 + sr + RS * i;
            kr = kr < N ? kr : N - 1;  // clamped rows are masked below
            const float *p = base + (long long)kr * D3 + sc * 4;
            rk[i] = *reinterpret_cast<const f32x4 *>(p + D);
            rv[i] = *reinterpret_cast<const f32x4 *>(p + 2 * D);
        }
    };

    const int ntiles = (N + KT - 1) / KT;
    load_tile(0);
    for (int t = 0; t < ntiles; ++t) {
        const int k0 = t * KT;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < RI; ++i) {
            *reinterpret_cast<f32x4 *>(&sK[(sr + RS * i) * KS + sc * 4]) = rk[i];
            *reinterpret_cast<f32x4 *>(&sV[(sr + RS * i) * HD + sc * 4]) = rv[i];
        }
        __syncthreads();
        if (t + 1 < ntiles) load_tile(k0 + KT);

        // ---- Sᵀ = K Qᵀ : KT/32 sub-tiles of 32 keys
        constexpr int NS = KT / 32;
        f32x16 sc[NS];
#pragma unroll
        for (int u = 0; u < NS; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[u][r] = 0.f;
#pragma unroll
        for (int qq = 0; qq < 8; ++qq) {
            f32x4 kf[NS];
#pragma unroll
            for (int u = 0; u < NS; ++u) kf[u] = *reinterpret_cast<const f32x4 *>(&sK[(32 * u + l31) * KS + 8 * qq + 4 * lh]);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int u = 0; u < NS; ++u) sc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[u][e], qf[qq][e], sc[u], 0, 0, 0);
        }
        if (k0 + KT > N) {  // last tile: keys past the sequence end
#pragma unroll
            for (int u = 0; u < NS; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (k0 + 32 * u + (r & 3) + 8 * (r >> 2) + 4 * lh >= N) sc[u][r] = -INFINITY;
        }

        // ---- online softmax in base 2; this lane holds half of its query's KT scores, lane^32 the rest
        float mx = sc[0][0];
#pragma unroll
        for (int u = 0; u < NS; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sc[u][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);  // first tile: exp2(-inf) = 0
        m_run = m_new;
        float psum = 0.f;
#pragma unroll
        for (int u = 0; u < NS; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                sc[u][r] = __builtin_amdgcn_exp2f(sc[u][r] - m_new);
                psum += sc[u][r];
            }
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            o0[r] *= alpha;
            o1[r] *= alpha;
        }

        // ---- Oᵀ += Vᵀ Pᵀ : step r of sub-tile u contracts keys {key(r,0), key(r,1)}
#pragma unroll
        for (int u = 0; u < NS; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = 32 * u + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const float v0 = sV[key * HD + l31], v1 = sV[key * HD + 32 + l31];
                o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v0, sc[u][r], o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v1, sc[u][r], o1, 0, 0, 0);
            }
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (qi < N) {
        float *orow = out + ((long long)frame * N + qi) * D + head * HD;
#pragma unroll
        for (int g = 0; g < 4; ++g) {  // registers 4g..4g+3 are d = 8g + 4*lh + {0..3}
            f32x4 a = {o0[4 * g] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv};
            f32x4 b = {o1[4 * g] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv};
            *reinterpret_cast<f32x4 *>(orow + 8 * g + 4 * lh) = a;
            *reinterpret_cast<f32x4 *>(orow + 32 + 8 * g + 4 * lh) = b;
        }
    }
}

}  // namespace

int attn_spatial(const float *qkv, float *out, int F, int N, int heads, hipStream_t st) {
    EDV_CHECK(qkv && out, "null operand");
    EDV_CHECK(F > 0 && N > 0 && heads > 0, "empty problem");
    EDV_CHECK((long long)F * heads * ((N + 31) / 32) < (1ll << 31), "grid limits");
    EDV_CHECK(((uintptr_t)qkv % 16 == 0) && ((uintptr_t)out % 16 == 0), "16-byte alignment");
    // 4 waves (128 queries) per workgroup share each staged K/V tile.  The 2- and 1-wave variants give a finer
    // grid but cost registers (195 / 256 VGPRs) and measured slower on every shape tried (T=8: 79.9 vs 78.6 vs
    // 68.5 TF/s); they are kept for sequences shorter than one 128-query block and for experiments.
    static const int forced = [] {
        const char *e = getenv("EDV_ATTN_WAVES");
        return e ? atoi(e) : 0;
    }();
    int nw = N > 64 ? 4 : (N > 32 ? 2 : 1);
    if (forced == 1 || forced == 2 || forced == 4) nw = forced;
    dim3 grid((unsigned)((long long)((N + nw * 32 - 1) / (nw * 32)) * heads * F));
    static const int kt_forced = [] {
        const char *e = getenv("EDV_ATTN_KT");
        return e ? atoi(e) : 0;
    }();
    const int kt = kt_forced == 32 ? 32 : 64;  // measured equal at N = 1370 (283.7 vs 284.0 us), 64 slightly ahead at N = 4096
    if (nw == 4 && kt == 32)
        hipLaunchKernelGGL((attn_spatial_kernel<4, 32>), grid, dim3(256), 0, st, qkv, out, N, heads);
    else if (nw == 4)
        hipLaunchKernelGGL((attn_spatial_kernel<4, 64>), grid, dim3(256), 0, st, qkv, out, N, heads);
    else if (nw == 2)
        hipLaunchKernelGGL((attn_spatial_kernel<2, 32>), grid, dim3(128), 0, st, qkv, out, N, heads);
    else
        hipLaunchKernelGGL((attn_spatial_kernel<1, 32>), grid, dim3(64), 0, st, qkv, out, N, heads);
    EDV_LAUNCH_OK();
    return 0;
}

}  // namespace edv
