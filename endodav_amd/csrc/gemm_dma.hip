// fp32-MFMA GEMM with LDS-DMA staging (global_load_lds, 16 bytes per lane): dense A only, K % 32 == 0.
//
// Same contract and epilogues as gemm.hip.  What changes is how the 64x64x32 tiles reach LDS: each wave issues
// four DMA instructions per k-tile (8 rows x 128 B each) that write LDS directly — no VGPR round trip, no ds_write,
// 48 instead of 100 registers per lane and 32 KB of LDS for two stages, so 5 workgroups fit a CU instead of 4.
// The ablation in scratch/ubench/gemm_ablate.hip priced exactly these staging instructions at ~20 % of the loop.
//
// LDS image: unpadded 128-byte rows (the DMA destination is lane-linear), 16-byte chunk c of row r stored at chunk
// position c ^ ((r >> 1) & 7).  The swizzle is applied to the per-lane SOURCE address and to the fragment read
// (rule 21 of the CDNA guide); a 16-lane ds_read_b128 group then covers 16 distinct (r & 1, chunk) pairs = all 64 banks.
// Pipeline: tile k+1 is in flight while tile k is multiplied; each wave waits for its own DMAs with a counted
// s_waitcnt vmcnt before the raw s_barrier that publishes the stage (a plain __syncthreads() would drain vmcnt to 0
// anyway here, but the raw form keeps the wait where the data is needed).
//
// Measured (profiles/r01_gemm_tile_sweep.txt), with the full epilogue: T=8 qkv 120 -> 111 us, fc2 162 -> 152 us,
// T=32 qkv 363 -> 340 us (114 TF/s); end to end +2..3 % on ViT-S/B/L.  On bare 8192x8192x1024 the DMA fill rate per CU
// caps it near 100 TF/s (register staging: 117), a regime the model does not reach with its K <= 4096, N <= 4096.
#include "gemm_common.hpp"

// Timeline hook for scratch/ubench/gemm_trace.hip; expands to nothing in the product build.
#ifndef EDV_GEMM_STAMP
#define EDV_GEMM_STAMP(slot)
#endif

namespace edv {
namespace {

constexpr int DBK = 32, DBM = 64, DBN = 64;
constexpr int DSTAGE = (DBM + DBN) * DBK;  // floats per stage (16 KB)

template <int STORE, int EP>
__global__ __launch_bounds__(256) void gemm_dma_kernel(const GemmDesc g) {
    __shared__ __attribute__((aligned(16))) float smem[2 * DSTAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    EDV_GEMM_STAMP(0);
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (g.N + DBN - 1) / DBN;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
    const long long m0 = (long long)tm * DBM;
    const int n0 = tn * DBN;

    // staging: wave w owns rows [16w, 16w+16) of the A tile and of the W tile; one DMA instruction = 8 rows x 128 B
    const int srow = lane >> 3, spos = lane & 7;
    const float *ga[2], *gb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = 16 * wave + 8 * i + srow;   // row within the tile
        const int c = spos ^ ((r >> 1) & 7);      // logical chunk that lives at this position
        long long m = m0 + r;
        m = m < g.M ? m : g.M - 1;                // rows past the edge read a valid row; their results are never stored
        int n = n0 + r;
        n = n < g.N ? n : g.N - 1;
        ga[i] = g.A + g.a_map(m) * g.lda + c * 4;
        gb[i] = g.W + (long long)n * g.ldw + c * 4;
    }
    auto issue = [&](int kt, int st) {
        float *sA = smem + st * DSTAGE, *sB = sA + DBM * DBK;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(ga[i] + kt * DBK),
                                             (__attribute__((address_space(3))) void *)(sA + (16 * wave + 8 * i) * DBK), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gb[i] + kt * DBK),
                                             (__attribute__((address_space(3))) void *)(sB + (16 * wave + 8 * i) * DBK), 16, 0, 0);
        }
    };

    EpiCols<1> cols;
    if (EP != 0) cols = gemm_epilogue_prefetch<1>(g, n0, wn * 32, l31);
    f32x16 acc[1][1];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;
    const int ra = wm * 32 + l31, rb = wn * 32 + l31;
    const int swa = (ra >> 1) & 7, swb = (rb >> 1) & 7;
    const int nkt = g.K / DBK;

    issue(0, 0);
    EDV_GEMM_STAMP(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    EDV_GEMM_STAMP(2);
    for (int kt = 0; kt < nkt; ++kt) {
        const int st = kt & 1;
        if (kt + 1 < nkt) issue(kt + 1, st ^ 1);  // the other stage was last read in iteration kt-1 (barrier passed)
        const float *sA = smem + st * DSTAGE, *sB = sA + DBM * DBK;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int cq = 2 * q + lh;  // logical 16-byte chunk: k = 8q + 4h .. 8q + 4h + 3 (the permuted-k trick of gemm.hip)
            const f32x4 fa = *reinterpret_cast<const f32x4 *>(&sA[ra * DBK + ((cq ^ swa) << 2)]);
            const f32x4 fb = *reinterpret_cast<const f32x4 *>(&sB[rb * DBK + ((cq ^ swb) << 2)]);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[e], fb[e], acc[0][0], 0, 0, 0);
        }
        // this wave's DMAs of tile kt+1 have landed, its fragment reads of tile kt are done -> publish / release
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    EDV_GEMM_STAMP(3);
    gemm_epilogue_ep<1, 1, STORE, EP>(g, acc, cols, m0, n0, wm * 32, wn * 32, l31, lh);
    EDV_GEMM_STAMP(4);
}

}  // namespace

bool gemm_dma_supported(const GemmDesc &d) {
    return d.loader == LOAD_DENSE && d.K % DBK == 0 && d.lda % 4 == 0 && d.ldw % 4 == 0 && d.M > 0 && d.N > 0;
}

int gemm_dma(const GemmDesc &d, hipStream_t st) {
    EDV_CHECK(d.A && d.W && d.C, "null operand");
    EDV_CHECK(gemm_dma_supported(d), "LDS-DMA GEMM needs a dense A and K % 32 == 0");
    EDV_CHECK(d.lda >= d.K && d.ldw >= d.K, "leading dimensions");
    EDV_CHECK(((uintptr_t)d.A % 16 == 0) && ((uintptr_t)d.W % 16 == 0), "A/W must be 16-byte aligned");
    if (d.store == STORE_SHUFFLE) {
        EDV_CHECK(d.ps_s > 0 && d.N == d.ps_s * d.ps_s * d.ps_C, "pixel-shuffle N");
        EDV_CHECK(d.R1 == nullptr && d.R2 == nullptr && d.P1 == nullptr, "pixel-shuffle store takes no residual");
    }
    const long long tiles = ((d.M + DBM - 1) / DBM) * (long long)((d.N + DBN - 1) / DBN);
    EDV_CHECK(tiles > 0 && tiles < (1ll << 31), "bad grid");
    dim3 grid((unsigned)tiles), block(256);
    if (d.store == STORE_SHUFFLE) {
        hipLaunchKernelGGL((gemm_dma_kernel<STORE_SHUFFLE, 0>), grid, block, 0, st, d);
    } else {
        switch (epilogue_kind(d)) {
            case 1: hipLaunchKernelGGL((gemm_dma_kernel<STORE_ROWS, 1>), grid, block, 0, st, d); break;
            case 2: hipLaunchKernelGGL((gemm_dma_kernel<STORE_ROWS, 2>), grid, block, 0, st, d); break;
            case 3: hipLaunchKernelGGL((gemm_dma_kernel<STORE_ROWS, 3>), grid, block, 0, st, d); break;
            default: hipLaunchKernelGGL((gemm_dma_kernel<STORE_ROWS, 0>), grid, block, 0, st, d); break;
        }
    }
    EDV_LAUNCH_OK();
    return 0;
}

}  // namespace edv
