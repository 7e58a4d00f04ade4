// fp32-MFMA GEMM with LDS-DMA staging (global_load_lds, 16 bytes per lane): dense A only, K % 32 == 0.
//
// Same contract and epilogues as gemm.hip.  What changes is how the 64x64x32 tiles reach LDS: each wave issues
// four DMA instructions per k-tile (8 rows x 128 B each) that write LDS directly — no VGPR round trip, no ds_write,
// 48 instead of 100 registers per lane and 32 KB of LDS for two stages (which places 4 workgroups per CU: the LDS allocation
// granule is 1280 B, so five 32 KB blocks do not fit 160 KB -- scratch/ubench/lds_residency.hip).
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
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "gemm_common.hpp"

// Timeline hook for scratch/ubench/gemm_trace.hip; expands to nothing in the product build.
#ifndef EDV_GEMM_STAMP
#define EDV_GEMM_STAMP(slot)
#endif

namespace edv {
namespace {

constexpr int DBK = 32, DBM = 64, DBN = 64;
constexpr int DSTAGE = (DBM + DBN) * DBK;  // floats per stage (16 KB)

// Work split of one launch (host-made, passed by value).  G = gridDim.x persistent workgroups, all co-resident (3 per CU: the split
// instantiation holds ~116 VGPRs, and a grid beyond the real residency loses everything the split gains).  Every workgroup first computes `whole_rounds` whole output tiles
// (tile = round * G + id).  The remaining tiles % G tiles are `units` = leftover_tiles * k_tiles k-tile units; they are cut into
// `nsplit` contiguous runs of `chunk` units, run j going to the workgroup with id j * stride (spread over the CUs).  A run that
// does not cover a tile's whole k range leaves its raw accumulators in workspace slot (j * 2 + segment), bumps the tile's
// arrival counter, and the LAST run to arrive sums all pieces of that tile in run order (so the result does not depend on the
// arrival order) and applies the epilogue: no fix-up launch, no spinning.  The counters live at the head of the workspace, must
// be zero before the first launch, and are left zero by every launch.  Without a workspace the launch is the plain grid:
// G = tiles, one whole tile each.
//
// Why it matters (profiles/r01_gemm_tile_sweep.txt, warm clocks): a CU retires tiles at a fixed MFMA-bound rate, so a grid of
// 1032 tiles on 256 CUs (N = 384 at 8 x 1370 rows) takes as long as 1280 tiles: fc2 138 us vs 112 us at 1020 tiles; with the split
// 121 us.  It pays for deep tiles (K >= 768) and for small grids; 12-k-tile tiles lose more to the static assignment than they gain.
// (GemmSplit, SLOT and MAX_COUNTERS live in gemm_common.hpp: conv_dma.hip uses the same scheme)
constexpr int SLOT = SPLIT_SLOT;
constexpr int MAX_COUNTERS = SPLIT_MAX_COUNTERS;

// SPLIT = false is the plain grid (one whole tile per workgroup): the split bookkeeping and the merge compile away, which
// keeps the hot instantiation at 48 VGPRs and its code in the instruction cache (with the merge inlined the same launches ran
// 2 % slower end to end).
//
// The k loop carries NO vector-ALU instruction (round 2).  On this part the f32 MFMA and the VALU do not overlap: v_mfma_f32_32x32x2_f32 runs at
// exactly the f32 vector rate, and every VALU instruction a wave issues between its MFMAs takes its 4-8 cycles out of the matrix pipe's time, for
// all waves of the SIMD (scratch/ubench/mfma_feed.hip: 160 VALU instructions per 128 MFMAs cost 19 % whether issued as a burst or interleaved;
// LDS reads, barriers and accumulator-to-operand forwarding cost nothing).  The round-1 loop spent ~23 VALU instructions per 16 MFMAs on
// addresses: 64-bit adds for the four DMA sources, a v_readfirstlane per DMA for M0 (the wave index is "divergent" to the compiler), an add
// per fragment read for the stage toggle.  Now: the DMA goes through buffer descriptors (per-lane 32-bit byte offset computed once per tile, the
// k advance in the SCALAR offset), the wave index is made scalar once, and the loop is unrolled by two so that the LDS stage is a
// compile-time immediate of ds_read_b128.  BUF = false keeps the flat-address form for operands beyond 4 GB.
// BUF: 0 = flat 64-bit DMA source addresses, 1 = buffer descriptors, 2 = buffer descriptors and identity A rows (a_map.period == 0: the row map's
// 64-bit division is not even compiled in; most launches).  __launch_bounds__(256, 4): four workgroups per CU is what LDS allows anyway, and with the
// 128-register budget spelled out the compiler keeps the accumulators in architectural VGPRs (no v_accvgpr_write / _read, which are VALU instructions).
template <int STORE, int EP, bool SPLIT, int BUF>
__global__ __launch_bounds__(256, 4) void gemm_dma_kernel(const GemmDesc g, const GemmSplit sp) {
    __shared__ __attribute__((aligned(16))) float smem[2 * DSTAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);  // the same number, provably wave-uniform: M0 and LDS bases become scalar arithmetic
    const auto rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(g.A), 0, 0xffffffff, 0x00020000);
    const auto rW = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(g.W), 0, 0xffffffff, 0x00020000);
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
    // run j = bid / stride of the split (only workgroups with bid % stride == 0 and j < nsplit own one)
    const int run = (SPLIT && sp.units > 0 && bid % sp.stride == 0 && bid / sp.stride < sp.nsplit) ? bid / sp.stride : -1;
    long long u = run >= 0 ? (long long)run * sp.chunk : 0;
    const long long u_end = run >= 0 ? (u + sp.chunk < sp.units ? u + sp.chunk : sp.units) : 0;
    int round = 0;
    for (;;) {
        int tile, kt0, kt1, lt = 0;
        float *part = nullptr;
        if (round < sp.whole_rounds) {
            tile = round * G + bid;
            kt0 = 0;
            kt1 = nkt;
            ++round;
        } else if (SPLIT && u < u_end) {
            const int t = (int)(u / nkt);
            kt0 = (int)(u - (long long)t * nkt);
            const long long left = u_end - u;
            kt1 = kt0 + left < nkt ? kt0 + (int)left : nkt;
            tile = tile_l0 + t;
            lt = t;
            // A run longer than a tile's k range has up to three segments: the tail of a tile, whole tiles, the head of a tile.  Only
            // the first and the last can be pieces.  Slot 0 = the piece that holds the run's first unit, slot 1 = the other one -- the
            // rule the merge applies from the tile's side (run start inside the tile -> slot 0).
            if (!(kt0 == 0 && kt1 == nkt)) part = sp.ws + ((long long)run * 2 + (u == (long long)run * sp.chunk ? 0 : 1)) * SLOT;
            u += kt1 - kt0;
        } else {
            break;
        }
        const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
        const long long m0 = (long long)tm * DBM;
        const int n0 = tn * DBN;

        // staging: wave w owns rows [16w, 16w+16) of the A tile and of the W tile; one DMA instruction = 8 rows x 128 B
        const float *ga[2], *gb[2];
        unsigned va[2], vb[2];  // BUF: byte offsets of this lane's 16 bytes in k-tile 0
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = 16 * wave + 8 * i + srow;   // row within the tile
            const int c = spos ^ ((r >> 1) & 7);      // logical chunk that lives at this position
            long long m = m0 + r;
            m = m < g.M ? m : g.M - 1;                // rows past the edge read a valid row; their results are never stored
            int n = n0 + r;
            n = n < g.N ? n : g.N - 1;
            const long long am = BUF == 2 ? m : g.a_map(m);
            ga[i] = g.A + am * g.lda + c * 4;
            gb[i] = g.W + (long long)n * g.ldw + c * 4;
            va[i] = (unsigned)((am * g.lda + c * 4) * 4);
            vb[i] = (unsigned)(((long long)n * g.ldw + c * 4) * 4);
        }
        auto issue = [&](int kt, int st) {
            float *sA = smem + st * DSTAGE, *sB = sA + DBM * DBK;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if constexpr (BUF) {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (__attribute__((address_space(3))) void *)(sA + (16 * wave_s + 8 * i) * DBK), 16, (unsigned)va[i],
                                                             (int)(kt * (DBK * 4)), 0, 0);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (__attribute__((address_space(3))) void *)(sB + (16 * wave_s + 8 * i) * DBK), 16, (unsigned)vb[i],
                                                             (int)(kt * (DBK * 4)), 0, 0);
                } else {
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(ga[i] + kt * DBK),
                                                     (__attribute__((address_space(3))) void *)(sA + (16 * wave + 8 * i) * DBK), 16, 0, 0);
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gb[i] + kt * DBK),
                                                     (__attribute__((address_space(3))) void *)(sB + (16 * wave + 8 * i) * DBK), 16, 0, 0);
                }
            }
        };
        // fragment read addresses in stage 0 (the stage toggle is an immediate offset): logical chunk 2q + lh of rows ra / rb
        const float *pa[4], *pb[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            pa[q] = smem + ra * DBK + (((2 * q + lh) ^ swa) << 2);
            pb[q] = smem + DBM * DBK + rb * DBK + (((2 * q + lh) ^ swb) << 2);
        }

        EpiCols<1> cols;
        if (EP != 0) cols = gemm_epilogue_prefetch<1>(g, n0, wn * 32, l31);
        f32x16 acc[1][1];
        auto epilogue = [&](long long em0, int en0) {
            if constexpr (EP == 6) {
                // GEGLU: the wave pair (wm, 0) / (wm, 1) holds values / gates of the same 32 output columns.  The gate wave leaves gelu(gate) in LDS
                // (both stages are idle here), the value wave multiplies and stores 32 columns of the half-width output through a buffer descriptor.
                float *ex = smem + (wave_s >> 1) * (16 * 64);
                if (wave_s & 1) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) ex[r * 64 + lane] = gelu_erf(acc[0][0][r] + cols.bias[0]);
                }
                __syncthreads();
                if (!(wave_s & 1)) {
                    const auto rc = __builtin_amdgcn_make_buffer_rsrc(g.C, 0, (int)(unsigned)(g.M * g.ldc * 4), 0x00020000);
                    const unsigned vo = (unsigned)((4 * lh * g.ldc + l31) * 4);
                    const unsigned so = (unsigned)(((em0 + (wave_s >> 1) * 32) * g.ldc + (en0 >> 1)) * 4);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float v = (acc[0][0][r] + cols.bias[0]) * ex[r * 64 + lane];
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rc, (int)vo, (int)(so + (unsigned)(((r & 3) + 8 * (r >> 2)) * g.ldc * 4)), 0);
                    }
                }
                __syncthreads();  // (a persistent instantiation would refill the stages next)
            } else if constexpr (BUF && STORE == STORE_ROWS && EP >= 1 && EP <= 3)
                gemm_epilogue_buf<EP - 1>(g, acc[0][0], cols, em0 + (wave_s >> 1) * 32, en0 + (wave_s & 1) * 32, l31, lh);
            else
                gemm_epilogue_ep<1, 1, STORE, EP>(g, acc, cols, em0, en0, wm * 32, wn * 32, l31, lh);
        };
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;

        // (the barrier that ended the previous run's last k-tile also released both LDS stages)
        issue(kt0, 0);
        EDV_GEMM_STAMP(1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        EDV_GEMM_STAMP(2);
        // one k-tile out of stage ST (compile-time): DMA of the next tile into the other stage, all eight fragment reads, sixteen MFMAs
        auto ktile = [&](int kt, auto st_tag) {
            constexpr int ST = decltype(st_tag)::value;
            if (kt + 1 < kt1) issue(kt + 1, ST ^ 1);  // the other stage was last read in the previous k-tile (barrier passed)
            f32x4 fa[4], fb[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {  // k = 8q + 4h .. 8q + 4h + 3 (the permuted-k trick of gemm.hip)
                fa[q] = *reinterpret_cast<const f32x4 *>(pa[q] + ST * DSTAGE);
                fb[q] = *reinterpret_cast<const f32x4 *>(pb[q] + ST * DSTAGE);
            }
            // all eight reads are in flight before the first MFMA waits for its pair (left alone the scheduler sinks each pair of reads
            // to its four MFMAs and re-uses one register quad: an LDS round trip exposed four times per k-tile)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q][e], fb[q][e], acc[0][0], 0, 0, 0);
            // this wave's DMAs of tile kt+1 have landed, its fragment reads of tile kt are done -> publish / release.  (The fence keeps the
            // MFMAs above the wait: they touch registers only, and the scheduler otherwise sinks fifteen of them below the barrier -- the
            // wave then waits for its DMA one MFMA after issuing it instead of sixteen.)
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        };
        {
            int kt = kt0;
            for (; kt + 1 < kt1; kt += 2) {
                ktile(kt, std::integral_constant<int, 0>{});
                ktile(kt + 1, std::integral_constant<int, 1>{});
            }
            if (kt < kt1) ktile(kt, std::integral_constant<int, 0>{});
        }
        EDV_GEMM_STAMP(3);
        if (SPLIT && part) {
            // Piece hand-off.  Pieces travel between workgroups on different XCDs (separate L2s, non-coherent L1s).  An agent-scope
            // release / acquire fence pair would do it, but on this part the release writes back the WHOLE L2 (buffer_wbl2) -- measured
            // +140 us per launch with every other workgroup's output tiles dirty in it.  Instead:
            //   producer  every piece word is stored with an agent-scope relaxed atomic store (global_store ... sc1: write-through);
            //             every storing wave then executes s_waitcnt vmcnt(0) (its stores are acknowledged), the workgroup meets at a
            //             barrier, and ONE lane adds 1 to the tile's agent-scope counter;
            //   consumer  the workgroup whose add returns "all pieces in" broadcasts that through LDS behind a workgroup barrier and
            //             reads every piece word with agent-scope relaxed atomic loads (global_load ... sc1: served by L2, never by this
            //             CU's L1), in run order whatever the arrival order was.
            // This is the "sc1 stores + drained counter + sc1 loads" hand-off the CDNA4 guide measures as valid (MI355X_MICROARCH.md,
            // Workgroup dispatch ..., Valid forms: one lane signals for all of its workgroup's stores after every wave's vmcnt(0) and
            // the barrier; every load of the handed-off bytes is an sc1 load issued after the add has returned and a barrier) -- measured
            // there at ONE workgroup per CU; this launch runs three, so the merging workgroup also executes the agent-scope acquire the
            // guide prescribes outside its table (split_merge_acquire, gemm_common.hpp).  It is
            // an ISA-level contract of gfx950 / ROCm 7.2, NOT a guarantee of the C++ memory model: the __syncthreads() between the
            // counter add and the loads is what keeps the compiler from hoisting the loads (a workgroup-scope fence), the hardware
            // ordering comes from sc1.  tests/test_streamk_fuzz_gpu.py::test_piece_exchange_contract_under_uneven_load is the gate to
            // re-run after a toolchain change; the fenced form to switch to is spelled out there.
#pragma unroll
            for (int r = 0; r < 16; ++r) __hip_atomic_store(&part[(wave * 16 + r) * 64 + lane], acc[0][0][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // pieces of leftover tile lt: the runs whose unit ranges intersect [lt * nkt, (lt + 1) * nkt)
            const long long ub = (long long)lt * nkt;
            const int g0 = (int)(ub / sp.chunk), g1 = (int)((ub + nkt - 1) / sp.chunk);
            int *s_last = reinterpret_cast<int *>(smem);  // both LDS stages are idle between the k loop and the next tile's first DMA
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                const int arrived = __hip_atomic_fetch_add(&sp.cnt[lt], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int last = arrived == g1 - g0;
                if (last) __hip_atomic_store(&sp.cnt[lt], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // all pieces in: zero for the next launch
                if (last) split_merge_acquire();  // this CU's L1 may hold stale lines of the piece slots (gemm_common.hpp)
                *s_last = last;
            }
            __syncthreads();
            const bool last = *s_last != 0;
            __syncthreads();  // s_last is read before the next tile's DMA may overwrite it
            if (last) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;
                const float *base = sp.ws + (wave * 16) * 64 + lane;
                int gg = g0;
                for (; gg + 1 <= g1; gg += 2) {  // two pieces in flight; summed in run order whatever the arrival order was
                    const float *pa = base + ((long long)gg * 2 + ((long long)gg * sp.chunk >= ub ? 0 : 1)) * SLOT;
                    const float *pb = base + ((long long)(gg + 1) * 2 + ((long long)(gg + 1) * sp.chunk >= ub ? 0 : 1)) * SLOT;
                    float ta[16], tb[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        ta[r] = __hip_atomic_load(pa + r * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        tb[r] = __hip_atomic_load(pb + r * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[0][0][r] = (acc[0][0][r] + ta[r]) + tb[r];
                }
                if (gg <= g1) {
                    const float *pa = base + ((long long)gg * 2 + ((long long)gg * sp.chunk >= ub ? 0 : 1)) * SLOT;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[0][0][r] += __hip_atomic_load(pa + r * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                epilogue(m0, n0);
            }
        } else {
            // (a second, separate epilogue call site on purpose: with one shared call behind a flag the split instantiation ran
            // 142 us instead of 122 us on fc2 at T=8)
            epilogue(m0, n0);
        }
        EDV_GEMM_STAMP(4);
    }
}

// operands the 32-bit buffer offsets reach (byte offset of the last element the loop may touch, rows clamped to the edge)
inline bool fits_buffer(const GemmDesc &d) {
    const long long a_rows = d.a_map(d.M - 1) + 1, lim = (1ll << 32) - (1 << 20);
    if (!(a_rows * d.lda * 4 < lim && (long long)d.N * d.ldw * 4 < lim)) return false;
    // The buffer epilogue (identity row maps only: EP 1..3) addresses C, R1 and R2 with 32-bit byte offsets as well, and it masks ragged edges
    // through the descriptor alone: a tile's rows run up to 63 past M, and their scalar offsets ((row0 + dr) * ld + col0) * 4 are computed in 32
    // bits -- they must not wrap, or a row past the end would come back INTO range and be stored.  Lanes past column N carry 0xfffff000, which
    // must lie at or beyond num_records = M * ld * 4.  Both hold iff (M + 64) rows of the widest operand stay below 2^32 - 4096 bytes.
    const long long ld = std::max(std::max(d.ldc, d.R1 ? d.ldr1 : 0), d.R2 ? d.ldr2 : 0);
    return (long long)(d.M + 64) * ld * 4 < (1ll << 32) - 4096;
}

template <int STORE, int EP>
int dma_slots() {
    static DeviceSlotCache cache;
    return cache.get([] {
        int dev = 0, cus = 0, per_cu = 0;
        if (hipGetDevice(&dev) != hipSuccess) return 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gemm_dma_kernel<STORE, EP, true, 2>, 256, 0) != hipSuccess) return 0;
        // The occupancy API answers 5 (5 x 32 KB = the whole 160 KB of LDS), the hardware places 4: the timeline of
        // scratch/ubench/gemm_trace.hip shows exactly 4 x 256 workgroups alive.  A persistent grid must match what is really
        // resident, or the surplus workgroups start only when others finish.  EDV_GEMM_SLOTS_PER_CU overrides.
        if (per_cu > 4) per_cu = 4;
        // ... and 3 is what the split runs with: at 3 and at 4 persistent workgroups per CU the step takes the same time (interleaved A/B,
        // profiles/r02_notes.txt), and 768 runs leave a quarter fewer pieces to write and re-read than 1024 (fc2 at T = 8: 32 vs 48 MB written).
        if (per_cu > 3) per_cu = 3;
        if (const char *e = getenv("EDV_GEMM_SLOTS_PER_CU")) per_cu = atoi(e) > 0 ? atoi(e) : per_cu;
        if (getenv("EDV_DEBUG_SLOTS")) fprintf(stderr, "gemm_dma_kernel<%d,%d>: %d CUs x %d resident workgroups\n", STORE, EP, cus, per_cu);
        return cus * per_cu;
    });
}

template <int STORE, int EP>
int launch_dma(const GemmDesc &d, long long tiles, hipStream_t st) {
    static const bool plain_forced = [] {
        const char *e = getenv("EDV_GEMM_PLAIN");  // 1: one workgroup per tile even with a workspace (A/B runs)
        return e && atoi(e) != 0;
    }();
    static const bool buf_off = [] {
        const char *e = getenv("EDV_GEMM_BUF");  // 0: flat 64-bit DMA source addresses as in round 1 (A/B runs)
        return e && atoi(e) == 0;
    }();
    const bool buf = !buf_off && fits_buffer(d);
    GemmSplit sp{1, 1, 0, 1, 0, nullptr, nullptr};
    long long grid = tiles;
    const int slots = dma_slots<STORE, EP>();
    EDV_CHECK(slots > 0 && slots <= MAX_COUNTERS, "occupancy query failed");
    const long long left = tiles % slots;
    // worth splitting only when the grid is a few rounds deep (beyond ~5 the last round is a small fraction and the persistent
    // form's 1-2 % deficit outweighs it: ViT-L fc2 at T=32, 10.7 rounds, measures -2 %)
    static const int min_kt = [] {
        const char *e = getenv("EDV_GEMM_SPLIT_MIN_KT");  // k-tiles per tile from which the split is used (A/B runs)
        return e ? atoi(e) : 24;
    }();
    // ... and the tiles are deep: statically assigned persistent workgroups lose 1-10 % against the hardware's dynamic dispatch
    // of the plain grid when a tile is only 12 k-tiles long (K = 384), and win 5-15 % from K = 768 up (warm A/B in
    // profiles/r01_gemm_tile_sweep.txt)
    static const int max_rounds = [] {
        const char *e = getenv("EDV_GEMM_SPLIT_MAX_ROUNDS");  // grids of at least this many rounds run plain (A/B runs)
        return e && atoi(e) > 0 ? atoi(e) : 8;
    }();
    static const int min_tiles = [] {
        const char *e = getenv("EDV_GEMM_SPLIT_MIN_TILES");  // smaller grids run plain (A/B runs)
        return e ? atoi(e) : 16;
    }();
    // small grids with deep tiles gain the most: 46 tiles x 48 k-tiles 33.7 -> 16.1 us, the C = 384 motion module's ff.net.2 (276 tiles) 57.5 -> 40.0
    if (d.ws && !plain_forced && left > 0 && tiles > min_tiles && tiles < (long long)max_rounds * slots &&
        d.K / DBK >= min_kt) {
        const int nkt = d.K / DBK;
        sp.whole_rounds = (int)(tiles / slots);
        long long split_tiles = left;
        // run length: an even share of the units over ALL workgroups.  When the leftover is too small for that (an even share would
        // be under 1/4 of a tile's k range, i.e. a handful of workgroups would carry the whole tail and the merge would chain up to
        // ~5 L2-bypassing loads at the very end of the launch), one whole round joins the split set instead: every workgroup then
        // runs whole_rounds - 1 whole tiles plus 1 + left/slots tiles' worth of k-tiles, and every split tile has 2-3 pieces.
        static const bool widen = [] {
            const char *e = getenv("EDV_GEMM_SPLIT_WIDEN");  // 0: never move a whole round into the split set (A/B runs)
            return !(e && atoi(e) == 0);
        }();
        const long long chunk_min = (nkt + 3) / 4;
        if (widen && sp.whole_rounds > 0 && (left * nkt + slots - 1) / slots < chunk_min && left + slots <= MAX_COUNTERS) {
            --sp.whole_rounds;
            split_tiles += slots;
        }
        sp.units = split_tiles * nkt;
        long long chunk = (sp.units + slots - 1) / slots;
        chunk = chunk > chunk_min ? chunk : chunk_min;
        sp.chunk = (int)chunk;
        sp.nsplit = (int)((sp.units + chunk - 1) / chunk);
        grid = sp.whole_rounds ? slots : (sp.nsplit > 0 ? sp.nsplit : 1);
        sp.stride = (int)(grid / sp.nsplit) > 0 ? (int)(grid / sp.nsplit) : 1;
        sp.cnt = reinterpret_cast<int *>(d.ws);
        sp.ws = d.ws + MAX_COUNTERS;
        EDV_CHECK((size_t)MAX_COUNTERS + (size_t)sp.nsplit * 2 * SLOT <= d.ws_floats && (uintptr_t)d.ws % 16 == 0,
                  "stream-K workspace too small (gemm_workspace)");
        if (buf && d.a_map.period == 0) EDV_LAUNCH((gemm_dma_kernel<STORE, EP, true, 2>), dim3((unsigned)grid), dim3(256), 0, st, d, sp);
        else if (buf) EDV_LAUNCH((gemm_dma_kernel<STORE, EP, true, 1>), dim3((unsigned)grid), dim3(256), 0, st, d, sp);
        else EDV_LAUNCH((gemm_dma_kernel<STORE, EP, true, 0>), dim3((unsigned)grid), dim3(256), 0, st, d, sp);
        EDV_LAUNCH_OK();
        return 0;
    }
    if (buf && d.a_map.period == 0) EDV_LAUNCH((gemm_dma_kernel<STORE, EP, false, 2>), dim3((unsigned)grid), dim3(256), 0, st, d, sp);
    else if (buf) EDV_LAUNCH((gemm_dma_kernel<STORE, EP, false, 1>), dim3((unsigned)grid), dim3(256), 0, st, d, sp);
    else EDV_LAUNCH((gemm_dma_kernel<STORE, EP, false, 0>), dim3((unsigned)grid), dim3(256), 0, st, d, sp);
    EDV_LAUNCH_OK();
    return 0;
}

// GEGLU epilogue (EP = 6): shallow tiles (K = C of a motion module), always the plain grid, buffer descriptors only (gemm_geglu_supported)
int launch_geglu(const GemmDesc &d, long long tiles, hipStream_t st) {
    const GemmSplit sp{1, 1, 0, 1, 0, nullptr, nullptr};
    if (d.a_map.period == 0) EDV_LAUNCH((gemm_dma_kernel<STORE_ROWS, 6, false, 2>), dim3((unsigned)tiles), dim3(256), 0, st, d, sp);
    else EDV_LAUNCH((gemm_dma_kernel<STORE_ROWS, 6, false, 1>), dim3((unsigned)tiles), dim3(256), 0, st, d, sp);
    EDV_LAUNCH_OK();
    return 0;
}

}  // namespace

size_t gemm_workspace() {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    return (size_t)MAX_COUNTERS + (size_t)cus * 8 * 2 * SLOT;  // arrival counters + two slots for each of at most 8 workgroups per CU
}

size_t gemm_counter_bytes() { return (size_t)MAX_COUNTERS * sizeof(int); }

bool gemm_dma_supported(const GemmDesc &d) {
    return d.loader == LOAD_DENSE && d.K % DBK == 0 && d.lda % 4 == 0 && d.ldw % 4 == 0 && d.M > 0 && d.N > 0;
}

bool gemm_geglu_supported(const GemmDesc &d) {
    return gemm_dma_supported(d) && d.store == STORE_ROWS && d.N % 64 == 0 && d.ldc >= d.N / 2 && !d.R1 && !d.R2 && !d.P1 && !d.gamma && d.bias &&
           d.c_map.period == 0 && fits_buffer(d);
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
        case 5: return launch_dma<STORE_ROWS, 5>(d, tiles, st);
        case 6:
            EDV_CHECK(gemm_geglu_supported(d), "GEGLU epilogue: dense A, K % 32 == 0, N % 64 == 0, bias, no residual, outputs below 4 GB");
            return launch_geglu(d, tiles, st);
        default: return launch_dma<STORE_ROWS, 0>(d, tiles, st);
    }
}

}  // namespace edv
