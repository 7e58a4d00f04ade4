// fp32 GEMM family on v_mfma_f32_32x32x2_f32 (exact f32, 64 FLOP/clk/SIMD; MI355X peak 157.3 TF).
//
// Covers every dense contraction of EndoDAV's forward outside the two attention cores:
//   F.linear            layers/attention.py:58,67  layers/mlp.py:34-37  motion_module.py:113,120
//   1x1 / 3x3 Conv2d    dpt.py:60-68,86-90,117-124  util/blocks.py:20-32,52-58,117
//   ConvTranspose2d k=s dpt.py:71-82 (as GEMM + pixel-shuffle store)
//   patch-embed conv    patch_embed.py:65,75 (after im2col by the patchify kernel)
//
// Tiling: workgroup = 256 threads = 4 waves, tile BM x BN x 32, two LDS stages.  A and W tiles are staged
// global -> VGPR -> LDS: the loads of tile k+2 are issued and tile k+1 is written to the other stage in the
// middle of tile k's MFMA stream (the T14 split of the CDNA guide), so there is ONE barrier per k-tile.  LDS rows are padded to 36 floats, which
// makes the ds_read_b128 fragment reads conflict-free (36r mod 64 is a bijection on r mod 16).
//
// K-order trick: the f32 MFMA takes ONE k per lane-half per instruction (lane l supplies
// A[l&31][l>>5]).  Instead of two ds_read_b32 per instruction, lane-half h reads the float4
// k = 8q+4h .. 8q+4h+3 of its row once and feeds element e to MFMA (q,e); A and W use the same
// permutation of k, so the sum over k is unchanged, and LDS traffic drops to one b128 per 4 MFMAs.
#include <cstdlib>

#include "gemm_common.hpp"

namespace edv {

namespace {

constexpr int BK = 32;
constexpr int LS = BK + 4;  // padded LDS row stride (floats)

template <int BM, int BN, int WGM, int WGN, int LOADER, int STORE, int EP>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmDesc g) {
    constexpr int WTM = BM / WGM, WTN = BN / WGN;
    constexpr int FM = WTM / 32, FN = WTN / 32;
    constexpr int RA = BM / 32, RB = BN / 32;  // float4 loads per thread per tile
    static_assert(WGM * WGN == 4 && FM >= 1 && FN >= 1, "4 waves");

    // two LDS stages: tile k+1 is written while tile k is being multiplied -> one barrier per k-tile
    constexpr int STAGE = (BM + BN) * LS;
    __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave / WGN, wn = wave % WGN;

    // XCD-aware, bijective block remap: blocks b and b+8 share an XCD (and its L2); give each XCD a
    // contiguous run of tiles, n fastest, so the column tiles of one A row-panel hit the same L2.
    const int tiles_n = (g.N + BN - 1) / BN;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
    const long long m0 = (long long)tm * BM;
    const int n0 = tn * BN;

    // ---- per-thread load slots: chunk c (4 consecutive k), rows r0 + 32*i ----
    const int c = tid & 7, r0 = tid >> 3;
    const float *a_ptr[RA];  // dense: row base; conv: frame base
    int a_iy[RA], a_ix[RA];
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        const long long m = m0 + r0 + 32 * i;
        if (m < g.M) {
            if (LOADER == LOAD_DENSE) {
                a_ptr[i] = g.A + g.a_map(m) * g.lda;
                a_iy[i] = a_ix[i] = 0;
            } else {
                const int opix = g.cOH * g.cOW;
                const long long f = m / opix;
                const int p = (int)(m - f * opix);
                const int oy = p / g.cOW, ox = p - oy * g.cOW;
                a_ptr[i] = g.A + f * (long long)g.cH * g.cW * g.cC;
                a_iy[i] = oy * g.cS - 1;
                a_ix[i] = ox * g.cS - 1;
            }
        } else {
            a_ptr[i] = nullptr;
            a_iy[i] = a_ix[i] = 0;
        }
    }
    const float *b_ptr[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        const int n = n0 + r0 + 32 * i;
        b_ptr[i] = (n < g.N) ? g.W + (long long)n * g.ldw : nullptr;
    }

    // two register sets: tiles k+1 and k+2 are in flight while tile k is multiplied
    f32x4 ra0[RA], rb0[RB], ra1[RA], rb1[RB];
    auto load_tile = [&](int kt, f32x4(&ra)[RA], f32x4(&rb)[RB]) {
        const int k = kt * BK + c * 4;
        const bool kin = k < g.K;  // K % 4 == 0, so a chunk is wholly in or out
        if (LOADER == LOAD_DENSE) {
#pragma unroll
            for (int i = 0; i < RA; ++i) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (kin && a_ptr[i]) v = *reinterpret_cast<const f32x4 *>(a_ptr[i] + k);
                ra[i] = v;
            }
        } else {
            const int tap = k / g.cC, ci = k - tap * g.cC;  // Cin % 4 == 0: chunk never straddles a tap
            const int dy = tap / 3, dx = tap - dy * 3;
#pragma unroll
            for (int i = 0; i < RA; ++i) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                const int iy = a_iy[i] + dy, ix = a_ix[i] + dx;
                if (kin && a_ptr[i] && iy >= 0 && iy < g.cH && ix >= 0 && ix < g.cW) {
                    v = *reinterpret_cast<const f32x4 *>(a_ptr[i] + ((long long)iy * g.cW + ix) * g.cC + ci);
                    if (g.pre_relu) {
                        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                    }
                }
                ra[i] = v;
            }
        }
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (kin && b_ptr[i]) v = *reinterpret_cast<const f32x4 *>(b_ptr[i] + k);
            rb[i] = v;
        }
    };

    EpiCols<FN> cols;
    if (EP != 0) cols = gemm_epilogue_prefetch<FN>(g, n0, wn * WTN, l31);
    f32x16 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nkt = (g.K + BK - 1) / BK;
    auto stage_store = [&](int buf, const f32x4(&ra)[RA], const f32x4(&rb)[RB]) {
        float *sA = smem + buf * STAGE;
        float *sB = sA + BM * LS;
#pragma unroll
        for (int i = 0; i < RA; ++i) *reinterpret_cast<f32x4 *>(&sA[(r0 + 32 * i) * LS + c * 4]) = ra[i];
#pragma unroll
        for (int i = 0; i < RB; ++i) *reinterpret_cast<f32x4 *>(&sB[(r0 + 32 * i) * LS + c * 4]) = rb[i];
    };
    auto mfma_q = [&](int buf, int q) {
        const float *sA = smem + buf * STAGE;
        const float *sB = sA + BM * LS;
        f32x4 fa[FM], fb[FN];
#pragma unroll
        for (int i = 0; i < FM; ++i) fa[i] = *reinterpret_cast<const f32x4 *>(&sA[(wm * WTM + i * 32 + l31) * LS + 8 * q + 4 * lh]);
#pragma unroll
        for (int j = 0; j < FN; ++j) fb[j] = *reinterpret_cast<const f32x4 *>(&sB[(wn * WTN + j * 32 + l31) * LS + 8 * q + 4 * lh]);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][e], fb[j][e], acc[i][j], 0, 0, 0);
    };

    // prologue: tile 0 -> LDS stage 0; tiles 1 and 2 -> register sets 0 and 1
    load_tile(0, ra0, rb0);
    stage_store(0, ra0, rb0);
    if (nkt > 1) load_tile(1, ra0, rb0);
    if (nkt > 2) load_tile(2, ra1, rb1);
    __syncthreads();
    // Steady state, unrolled by two so that the register sets keep static names.  In iteration kt the other LDS
    // stage was last read in iteration kt-1 (barrier passed), so tile kt+1 lands there mid-stream; its global loads
    // were issued TWO iterations ago, which is what hides the L2/HBM latency behind only ~1k MFMA cycles per
    // iteration on the 64x64 tile.  One barrier per k-tile.
    for (int kt = 0; kt < nkt; kt += 2) {
        mfma_q(0, 0);
        mfma_q(0, 1);
        if (kt + 1 < nkt) stage_store(1, ra0, rb0);
        if (kt + 3 < nkt) load_tile(kt + 3, ra0, rb0);
        mfma_q(0, 2);
        mfma_q(0, 3);
        __syncthreads();
        if (kt + 1 >= nkt) break;
        mfma_q(1, 0);
        mfma_q(1, 1);
        if (kt + 2 < nkt) stage_store(0, ra1, rb1);
        if (kt + 4 < nkt) load_tile(kt + 4, ra1, rb1);
        mfma_q(1, 2);
        mfma_q(1, 3);
        __syncthreads();
    }

    gemm_epilogue_ep<FM, FN, STORE, EP>(g, acc, cols, m0, n0, wm * WTM, wn * WTN, l31, lh);
}

// FASTEP: also instantiate the compact epilogues (gemm_common.hpp); the experiment-only wide tiles keep the general one.
template <int BM, int BN, int WGM, int WGN, int LOADER, bool FASTEP>
void launch_rows(const GemmDesc &d, dim3 grid, hipStream_t st) {
    const dim3 block(256);
    const int ep = FASTEP ? epilogue_kind(d) : 0;
    if constexpr (FASTEP) {
        if (ep == 1) {
            EDV_LAUNCH((gemm_kernel<BM, BN, WGM, WGN, LOADER, STORE_ROWS, 1>), grid, block, 0, st, d);
            return;
        }
        if (ep == 2) {
            EDV_LAUNCH((gemm_kernel<BM, BN, WGM, WGN, LOADER, STORE_ROWS, 2>), grid, block, 0, st, d);
            return;
        }
        if (ep == 3) {
            EDV_LAUNCH((gemm_kernel<BM, BN, WGM, WGN, LOADER, STORE_ROWS, 3>), grid, block, 0, st, d);
            return;
        }
    }
    EDV_LAUNCH((gemm_kernel<BM, BN, WGM, WGN, LOADER, STORE_ROWS, 0>), grid, block, 0, st, d);
}

template <int BM, int BN, int WGM, int WGN, bool FASTEP>
int launch_tile(const GemmDesc &d, hipStream_t st) {
    const long long tiles = ((d.M + BM - 1) / BM) * (long long)((d.N + BN - 1) / BN);
    EDV_CHECK(tiles > 0 && tiles < (1ll << 31), "bad grid");
    dim3 grid((unsigned)tiles), block(256);
    if (d.loader == LOAD_DENSE && d.store == STORE_ROWS)
        launch_rows<BM, BN, WGM, WGN, LOAD_DENSE, FASTEP>(d, grid, st);
    else if (d.loader == LOAD_CONV3 && d.store == STORE_ROWS)
        launch_rows<BM, BN, WGM, WGN, LOAD_CONV3, FASTEP>(d, grid, st);
    else if (d.loader == LOAD_DENSE && d.store == STORE_SHUFFLE)
        EDV_LAUNCH((gemm_kernel<BM, BN, WGM, WGN, LOAD_DENSE, STORE_SHUFFLE, 0>), grid, block, 0, st, d);
    else
        EDV_CHECK(false, "unsupported loader/store combination");
    EDV_LAUNCH_OK();
    return 0;
}

// 0: 128x128, 1: 128x64, 2: 128x32, 3: 64x64
int pick_tile(const GemmDesc &d) {
    static const int forced = [] {
        const char *e = getenv("EDV_GEMM_TILE");  // debug override for tile sweeps: 0..3
        return e ? atoi(e) : -1;
    }();
    if (forced >= 0 && forced <= 3) return forced;
    if (d.N <= 32) return 2;
    // Measured on MI355X (profiles/r01_gemm_tile_sweep.txt): with two LDS stages the 64x64 tile (4 workgroups of
    // 4 waves per CU, one 32x32 MFMA tile per wave) matches or beats the 128-wide tiles on every shape tried --
    // 80 vs 58 TF/s on qkv at T=8 (774 big tiles over 512 resident slots quantise badly), 105 vs 71 at T=32,
    // 114 vs 103 at 8192x8192x1024 -- so it is the default; the wide tiles stay reachable for experiments.
    return 3;
}

}  // namespace

int gemm(const GemmDesc &d, hipStream_t st) {
    EDV_CHECK(d.A && d.W && d.C, "null operand");
    EDV_CHECK(d.M > 0 && d.N > 0 && d.K > 0, "empty problem");
    EDV_CHECK(d.K % 4 == 0, "K must be a multiple of 4");
    EDV_CHECK(d.ldw % 4 == 0 && d.ldw >= d.K, "ldw");
    if (d.loader == LOAD_DENSE) {
        EDV_CHECK(d.lda % 4 == 0 && d.lda >= d.K, "lda");
    } else {
        EDV_CHECK(d.cC % 4 == 0 && d.K == 9 * d.cC, "conv3x3: Cin % 4, K = 9*Cin");
        EDV_CHECK(d.cS == 1 || d.cS == 2, "conv stride");
        EDV_CHECK(d.cOH == (d.cH + 2 - 3) / d.cS + 1 && d.cOW == (d.cW + 2 - 3) / d.cS + 1, "conv output size");
    }
    if (d.store == STORE_SHUFFLE) {
        EDV_CHECK(d.ps_s > 0 && d.N == d.ps_s * d.ps_s * d.ps_C, "pixel-shuffle N");
        EDV_CHECK(d.R1 == nullptr && d.R2 == nullptr && d.P1 == nullptr, "pixel-shuffle store takes no residual");
    }
    EDV_CHECK(((uintptr_t)d.A % 16 == 0) && ((uintptr_t)d.W % 16 == 0), "A/W must be 16-byte aligned");
    // LDS-DMA staging (gemm_dma.hip) is ahead on every shape the model produces (ViT-S T=8 +3.2 %, T=32 +2.7 %, ViT-B
    // +2.7 %, ViT-L +2.2 % end to end: profiles/r01_gemm_tile_sweep.txt); the register-staged kernel below keeps the
    // implicit-GEMM convolutions, K tails and narrow outputs.  EDV_GEMM_DMA=0 switches the DMA path off for A/B runs.
    static const bool dma_on = [] {
        const char *e = getenv("EDV_GEMM_DMA");
        return !(e && atoi(e) == 0);
    }();
    if (d.Wx6 && gemm_x6_supported(d)) return gemm_x6(d, st);  // the caller prepared bf16 planes of W: products on the bf16 matrix pipe (gemm_x6.hip)
    if (d.geglu) return gemm_dma(d, st);  // the GEGLU epilogue exists in the LDS-DMA kernel only (it checks gemm_geglu_supported)
    if (dma_on && gemm_dma_supported(d) && d.N > 32) return gemm_dma(d, st);
    static const bool conv_dma_on = [] {
        const char *e = getenv("EDV_CONV_DMA");  // 0 switches the LDS-DMA convolution off (A/B runs)
        return !(e && atoi(e) == 0);
    }();
    if (conv_dma_on && conv_dma_supported(d)) return conv_dma(d, st);
    switch (pick_tile(d)) {
        case 0: return launch_tile<128, 128, 2, 2, false>(d, st);
        case 1: return launch_tile<128, 64, 2, 2, false>(d, st);
        case 2: return launch_tile<128, 32, 4, 1, true>(d, st);
        default: return launch_tile<64, 64, 2, 2, true>(d, st);
    }
}

const char *gemm_kernel_name(const GemmDesc &d) {
    static const char *names[] = {"gemm_128x128", "gemm_128x64", "gemm_128x32", "gemm_64x64"};
    return names[pick_tile(d)];
}

}  // namespace edv
