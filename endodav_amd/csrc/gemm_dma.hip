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
#include <cstdio>

#include "gemm_common.hpp"

// Timeline hook for scratch/ubench/gemm_trace.hip; expands to nothing in the product build.
#ifndef EDV_GEMM_STAMP
#define EDV_GEMM_STAMP(slot)
#endif

namespace edv {
namespace {

constexpr int DBK = 32, DBM = 64, DBN = 64;
constexpr int DSTAGE = (DBM + DBN) * DBK;  // floats per stage (16 KB)

// Work split of one launch (host-made, passed by value).  G = gridDim.x persistent workgroups, all co-resident.
// Every workgroup first computes `whole_rounds` whole output tiles (tile = round * G + id), then its share
// [id * chunk, (id + 1) * chunk) of the `units` = leftover_tiles * k_tiles key-tile units of the remaining
// tiles % G tiles.  A run that does not cover a tile's whole k range leaves its raw accumulators in workspace slot
// (id * 2 + run) and gemm_fixup_kernel sums the pieces and applies the epilogue.  Without a workspace the launch is the
// plain grid: G = tiles, one whole tile each.
struct GemmSplit {
    int whole_rounds, chunk;
    long long units;
    float *ws;
};
constexpr int SLOT = DBM * DBN;  // floats per workspace slot; element (wave, r, lane) at (wave * 16 + r) * 64 + lane

template <int STORE, int EP>
__global__ __launch_bounds__(256) void gemm_dma_kernel(const GemmDesc g, const GemmSplit sp) {
    __shared__ __attribute__((aligned(16))) float smem[2 * DSTAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    EDV_GEMM_STAMP(0);
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (g.N + DBN - 1) / DBN;
    const int G = gridDim.x;
    const int bid = xcd_remap(blockIdx.x, G);
    const int nkt = g.K / DBK;
    const int srow = lane >> 3, spos = lane & 7;
    const int ra = wm * 32 + l31, rb = wn * 32 + l31;
    const int swa = (ra >> 1) & 7, swb = (rb >> 1) & 7;

    const int tile_l0 = sp.whole_rounds * G;
    long long u = (long long)bid * sp.chunk;
    const long long u_end = u + sp.chunk < sp.units ? u + sp.chunk : sp.units;
    int round = 0, seg = 0;
    for (;;) {
        int tile, kt0, kt1;
        float *part = nullptr;
        if (round < sp.whole_rounds) {
            tile = round * G + bid;
            kt0 = 0;
            kt1 = nkt;
            ++round;
        } else if (u < u_end) {
            const int t = (int)(u / nkt);
            kt0 = (int)(u - (long long)t * nkt);
            const long long left = u_end - u;
            kt1 = kt0 + left < nkt ? kt0 + (int)left : nkt;
            tile = tile_l0 + t;
            u += kt1 - kt0;
            if (!(kt0 == 0 && kt1 == nkt)) part = sp.ws + ((long long)bid * 2 + seg) * SLOT;
            ++seg;  // slot 0 = the run holding this workgroup's first unit, slot 1 = the head of the next tile
        } else {
            break;
        }
        const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
        const long long m0 = (long long)tm * DBM;
        const int n0 = tn * DBN;

        // staging: wave w owns rows [16w, 16w+16) of the A tile and of the W tile; one DMA instruction = 8 rows x 128 B
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

        // (the barrier that ended the previous run's last k-tile also released both LDS stages)
        issue(kt0, 0);
        EDV_GEMM_STAMP(1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        EDV_GEMM_STAMP(2);
        for (int kt = kt0; kt < kt1; ++kt) {
            const int st = (kt - kt0) & 1;
            if (kt + 1 < kt1) issue(kt + 1, st ^ 1);  // the other stage was last read in the previous iteration (barrier passed)
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
        if (part) {
#pragma unroll
            for (int r = 0; r < 16; ++r) part[(wave * 16 + r) * 64 + lane] = acc[0][0][r];
        } else {
            gemm_epilogue_ep<1, 1, STORE, EP>(g, acc, cols, m0, n0, wm * 32, wn * 32, l31, lh);
        }
        EDV_GEMM_STAMP(4);
    }
}

// Sums the pieces of every split tile and applies the epilogue.  Pieces of leftover tile t: the workgroups whose unit
// runs intersect [t * nkt, (t + 1) * nkt); a workgroup's piece is in its slot 0 when its first unit lies in this tile.
template <int STORE>
__global__ __launch_bounds__(256) void gemm_fixup_kernel(const GemmDesc g, const GemmSplit sp, int tile_l0) {
    const int nkt = g.K / DBK;
    const int t = blockIdx.x;
    const long long ub = (long long)t * nkt, ue = ub + nkt;
    const int g0 = (int)(ub / sp.chunk), g1 = (int)((ue - 1) / sp.chunk);
    if (g0 == g1 && (long long)g0 * sp.chunk <= ub && (long long)(g0 + 1) * sp.chunk >= ue) return;  // ran whole
    const int e = blockIdx.y * 256 + threadIdx.x;  // element of the 64 x 64 tile in accumulator order
    float v = 0.f;
    for (int gg = g0; gg <= g1; ++gg) v += sp.ws[((long long)gg * 2 + ((long long)gg * sp.chunk >= ub ? 0 : 1)) * SLOT + e];
    const int lane = e & 63, r = (e >> 6) & 15, wave = e >> 10;
    const int tiles_n = (g.N + DBN - 1) / DBN;
    const int tile = tile_l0 + t;
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const long long m = (long long)tm * DBM + (wave >> 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    const int n = tn * DBN + (wave & 1) * 32 + (lane & 31);
    if (m < g.M && n < g.N) gemm_epilogue_elem<STORE>(g, v, m, n);
}

template <int STORE, int EP>
int dma_slots() {
    static const int slots = [] {
        int dev = 0, cus = 0, per_cu = 0;
        if (hipGetDevice(&dev) != hipSuccess) return 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gemm_dma_kernel<STORE, EP>, 256, 0) != hipSuccess) return 0;
        // The occupancy API answers 5 (5 x 32 KB = the whole 160 KB of LDS), the hardware places 4: the timeline of
        // scratch/ubench/gemm_trace.hip shows exactly 4 x 256 workgroups alive.  A persistent grid must match what is really
        // resident, or the surplus workgroups start only when others finish.  EDV_GEMM_SLOTS_PER_CU overrides.
        if (per_cu > 4) per_cu = 4;
        if (const char *e = getenv("EDV_GEMM_SLOTS_PER_CU")) per_cu = atoi(e) > 0 ? atoi(e) : per_cu;
        if (getenv("EDV_DEBUG_SLOTS")) fprintf(stderr, "gemm_dma_kernel<%d,%d>: %d CUs x %d resident workgroups\n", STORE, EP, cus, per_cu);
        return cus * per_cu;
    }();
    return slots;
}

template <int STORE, int EP>
int launch_dma(const GemmDesc &d, long long tiles, hipStream_t st) {
    static const bool plain_forced = [] {
        const char *e = getenv("EDV_GEMM_PLAIN");  // 1: one workgroup per tile even with a workspace (A/B runs)
        return e && atoi(e) != 0;
    }();
    GemmSplit sp{1, 1, 0, nullptr};
    long long grid = tiles;
    const int slots = dma_slots<STORE, EP>();
    EDV_CHECK(slots > 0, "occupancy query failed");
    // worth splitting only when the grid is a few rounds deep: beyond that the tail is a small fraction
    if (d.ws && !plain_forced && tiles > slots / 2 && tiles < 16ll * slots) {
        sp.whole_rounds = (int)(tiles / slots);
        const long long left = tiles - (long long)sp.whole_rounds * slots;
        sp.units = left * (d.K / DBK);
        sp.chunk = sp.units ? (int)((sp.units + slots - 1) / slots) : 1;
        sp.ws = d.ws;
        grid = sp.whole_rounds ? slots : (sp.units + sp.chunk - 1) / sp.chunk;
        const long long split_wgs = (sp.units + sp.chunk - 1) / sp.chunk;
        EDV_CHECK((size_t)split_wgs * 2 * SLOT <= d.ws_floats && (uintptr_t)d.ws % 16 == 0, "stream-K workspace too small (gemm_workspace)");
        hipLaunchKernelGGL((gemm_dma_kernel<STORE, EP>), dim3((unsigned)grid), dim3(256), 0, st, d, sp);
        EDV_LAUNCH_OK();
        if (left) {
            hipLaunchKernelGGL((gemm_fixup_kernel<STORE>), dim3((unsigned)left, SLOT / 256), dim3(256), 0, st, d, sp, sp.whole_rounds * slots);
            EDV_LAUNCH_OK();
        }
        return 0;
    }
    hipLaunchKernelGGL((gemm_dma_kernel<STORE, EP>), dim3((unsigned)grid), dim3(256), 0, st, d, sp);
    EDV_LAUNCH_OK();
    return 0;
}

}  // namespace

size_t gemm_workspace() {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    return (size_t)cus * 8 * 2 * SLOT;  // at most 8 co-resident workgroups per CU (LDS: 32 KB each), two slots each
}

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
    if (d.store == STORE_SHUFFLE) return launch_dma<STORE_SHUFFLE, 0>(d, tiles, st);
    switch (epilogue_kind(d)) {
        case 1: return launch_dma<STORE_ROWS, 1>(d, tiles, st);
        case 2: return launch_dma<STORE_ROWS, 2>(d, tiles, st);
        case 3: return launch_dma<STORE_ROWS, 3>(d, tiles, st);
        default: return launch_dma<STORE_ROWS, 0>(d, tiles, st);
    }
}

}  // namespace edv
