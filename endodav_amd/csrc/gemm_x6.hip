// fp32-in / fp32-out GEMM whose products run on the bf16 matrix pipe ("bf16 x 6", round 3).  Same contract and epilogues as gemm.hip / gemm_dma.hip.
//
//   a = a0 + a1 + a2,  a0 = bf16(a), a1 = bf16(a - a0), a2 = bf16(a - a0 - a1)       |a1| <= 2^-9 |a|, |a2| <= 2^-18 |a|, remainder <= 2^-27 |a|
//   a w ~= a2 w0 + a0 w2 + a1 w1 + a1 w0 + a0 w1 + a0 w0                              dropped: a1 w2 + a2 w1 + a2 w2 <= 2^-26 |a w|
//
// A bf16 x bf16 product is exact in fp32 and the MFMA accumulates in fp32, so each term of the dot product carries an error below fp32's own unit
// roundoff (2^-24) -- measured on the encoder's shapes against fp64 (scratch/ubench/gemm_bf16x6.hip, profiles/r03_gemm_bf16x6_ubench.txt): rms error
// 1.9-2.1e-8 of sum |a w| against 2.1-2.4e-8 for the v_mfma_f32_32x32x2_f32 kernels, the same maximum.  Six v_mfma_f32_32x32x16_bf16 take 192
// matrix-pipe cycles per 16 k where eight v_mfma_f32_32x32x2_f32 take 512; the fp32 pipe's 157 TFLOP/s stops being the ceiling.
//
// W (static) is split once into three bf16 planes [3][N][K] (edv_prepare / edv_refresh_lora) and streams to LDS by LDS-DMA.  A (activations, fp32 in
// HBM, written by fp32 producers) is split on the way: the staging threads load 16-byte pieces to registers, split them (v_cvt_pk_bf16_f32 + two
// subtractions per plane) and write the three planes to LDS.  That is vector-ALU work in the k loop, which gemm_dma.hip avoids at all cost -- but beside the
// bf16 MFMA it hides: the instruction holds the SIMD's issue port for 8 of its 32 cycles (MI355X_MICROARCH.md, cycle constants), and with two
// workgroups per CU the partner wave's MFMAs run under this wave's conversions.
//
// Tile 128 x 128, 16 k per stage, 3 stages of 24 KB (two workgroups per CU), 4 waves each holding 64 x 64 (2 x 2 accumulators).  W DMA and the A loads run
// two stages ahead.  LDS plane image: 32-byte rows, the two 16-byte halves of row r swapped when (r >> 3) & 1, so a 16-lane ds_read_b128 group covers all
// 64 banks.  Work split: the stream-K scheme of gemm_dma.hip (whole rounds of tiles, the leftover round cut along k into runs, the last arriver of a tile
// merges the pieces in run order and applies the epilogue) with 128 x 128 pieces.
//
// Used for the encoder's linears in inference (engine.hip: Run::linear picks it when the weight has planes and the context computes in
// EDV_PRODUCTS_BF16X6); training keeps the fp32-MFMA kernels, whose saved activations the backward expects.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "gemm_common.hpp"

// Timeline hook for scratch/ubench/gemm_x6_trace.hip; expands to nothing in the product build.
#ifndef EDV_X6_STAMP
#define EDV_X6_STAMP(slot)
#endif

namespace edv {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int XBM = 128, XBN = 128, XBK = 16;
constexpr int XPLANE = XBM * XBK * 2;  // bytes of one operand plane of a stage
constexpr int XSTAGE = 6 * XPLANE;     // A0 A1 A2 W0 W1 W2
constexpr int XNST = 3;
constexpr int XSLOT = XBM * XBN;       // floats per piece, ((wave * 4 + i * 2 + j) * 16 + r) * 64 + lane
constexpr int XMAX_COUNTERS = SPLIT_MAX_COUNTERS;

__global__ __launch_bounds__(256) void split3_kernel(const float *__restrict__ w, __bf16 *__restrict__ p, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float x = w[i];
        const __bf16 h = (__bf16)x;
        const float r1 = x - (float)h;
        const __bf16 m = (__bf16)r1;
        p[i] = h;
        p[n + i] = m;
        p[2 * n + i] = (__bf16)(r1 - (float)m);
    }
}

__device__ __forceinline__ int half_pos(int r, int h) { return (h ^ ((r >> 3) & 1)) * 16; }

template <int ACT, bool SPLIT>  // ACT: compile-time activation of the epilogue (ACT_NONE / ACT_GELU / ACT_RELU)
__global__ __launch_bounds__(256, 2) void gemm_x6_kernel(const GemmDesc g, const GemmSplit sp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (g.N + XBN - 1) / XBN, tiles_m = (int)((g.M + XBM - 1) / XBM);
    const int G = gridDim.x;
    const int bid = xcd_remap(blockIdx.x, G);
    const int nkt = g.K / XBK;
    const __bf16 *Wp = reinterpret_cast<const __bf16 *>(g.Wx6);
    const auto rW = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16 *>(Wp), 0, 0xffffffff, 0x00020000);
    const long long plane_bytes = (long long)g.N * g.ldw * 2;

    // fragment addresses in stage 0
    int fa[2], fb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ra = wm * 64 + i * 32 + l31, rb = wn * 64 + i * 32 + l31;
        fa[i] = ra * 32 + half_pos(ra, lh);
        fb[i] = 3 * XPLANE + rb * 32 + half_pos(rb, lh);
    }

    const int tile_l0 = sp.whole_rounds * G;
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
            if (!(kt0 == 0 && kt1 == nkt)) part = sp.ws + ((long long)run * 2 + (u == (long long)run * sp.chunk ? 0 : 1)) * XSLOT;
            u += kt1 - kt0;
        } else {
            break;
        }
        // Tile order: groups of `gm` row blocks swept column by column (row block fastest), so that the workgroups an XCD runs side by side share
        // few W column tiles and a handful of A row blocks instead of one A row block and every W tile (xcd_remap gives an XCD a contiguous run of
        // tile indices).  gm = 1 is the plain row-major order.
        int tm, tn;
        {
            const int per_group = sp.group_m * tiles_n;
            const int gidx = tile / per_group, first = gidx * sp.group_m;
            const int rows = tiles_m - first < sp.group_m ? tiles_m - first : sp.group_m;
            const int r = tile - gidx * per_group;
            tn = r / rows;
            tm = first + (r - tn * rows);
        }
        const long long m0 = (long long)tm * XBM;
        const int n0 = tn * XBN;

        EDV_X6_STAMP(0);
        // ---- staging roles ----
        // W: 12 DMA instructions per stage (3 planes x 4 groups of 32 rows), three per wave; lane i fills 16-byte chunk i of the 1 KB an instruction writes
        unsigned vw[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int id = wave * 3 + j, p = id >> 2, grp = id & 3;
            const int r = grp * 32 + (lane >> 1), pos = lane & 1;
            const int h = pos ^ ((r >> 3) & 1);
            int n = n0 + r;
            n = n < g.N ? n : g.N - 1;  // rows past the edge read a valid row; their columns are never stored
            vw[j] = (unsigned)(p * plane_bytes + ((long long)n * g.ldw + h * 8) * 2);
        }
        // A: two 16-byte pieces of fp32 per thread per stage (row c / 4, floats (c % 4) * 4 ..)
        const float *ga[2];
        int dsta[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c = tid + 256 * j, r = c >> 2, q = c & 3;
            long long m = m0 + r;
            m = m < g.M ? m : g.M - 1;
            ga[j] = g.A + g.a_map(m) * g.lda + q * 4;
            dsta[j] = r * 32 + half_pos(r, q >> 1) + (q & 1) * 8;
        }
        auto issue_w = [&](int kt, int st) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int id = wave_s * 3 + j;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (__attribute__((address_space(3))) void *)(smem + st * XSTAGE + (3 + (id >> 2)) * XPLANE + (id & 3) * 1024), 16,
                                                         vw[j], (int)(kt * (XBK * 2)), 0, 0);
            }
        };
        auto load_a = [&](int kt, f32x4 (&v)[2]) {
#pragma unroll
            for (int j = 0; j < 2; ++j) v[j] = *reinterpret_cast<const f32x4 *>(ga[j] + kt * XBK);
        };
        auto split_store_a = [&](const f32x4 (&v)[2], int st) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                bf16x4 p0, p1, p2;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x = v[j][e];
                    p0[e] = (__bf16)x;
                    const float r1 = x - (float)p0[e];
                    p1[e] = (__bf16)r1;
                    p2[e] = (__bf16)(r1 - (float)p1[e]);
                }
                unsigned char *d = smem + st * XSTAGE + dsta[j];
                *reinterpret_cast<bf16x4 *>(d) = p0;
                *reinterpret_cast<bf16x4 *>(d + XPLANE) = p1;
                *reinterpret_cast<bf16x4 *>(d + 2 * XPLANE) = p2;
            }
        };

        const EpiCols<2> cols = gemm_epilogue_prefetch<2>(g, n0, wn * 64, l31);
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        // prologue: stages 0 and 1 of this run filled, A of the third step in registers.  (The barrier that ended the previous run's last step also
        // released all three stages.)
        // prologue: stages 0 and 1 of this run filled; A of steps 2, 3, 4 in flight in the three register sets (set i is consumed by the steps that run
        // out of stage i and refilled three steps ahead: the conversion never waits for a load younger than three steps)
        const int nk = kt1 - kt0;
        f32x4 aset[3][2];
        load_a(kt0, aset[0]);  // both A pieces of the first two steps are in flight before the first is needed: one load latency, not two
        if (nk > 1) load_a(kt0 + 1, aset[1]);
        issue_w(kt0, 0);
        if (nk > 1) issue_w(kt0 + 1, 1);
        split_store_a(aset[0], 0);
        if (nk > 1) split_store_a(aset[1], 1);
        if (nk > 2) load_a(kt0 + 2, aset[0]);
        if (nk > 3) load_a(kt0 + 3, aset[1]);
        if (nk > 4) load_a(kt0 + 4, aset[2]);
        if (nk > 4)
            asm volatile("s_waitcnt vmcnt(9)" ::: "memory");  // W of the first step landed; W of the second (x3) and A of three steps (x6) may stay in flight
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        EDV_X6_STAMP(1);

        // one step out of stage ST (compile-time): W DMA and A conversion of step + 2 into the stage read last step, A loads of step + 3, 12 fragment reads, 24 MFMAs
        // FULL: a step in the steady state (kt + 5 < kt1): every staging action happens, in ONE basic block, and the schedule is spelled out -- the
        // conversion of A(kt + 2) (about 70 VALU instructions) and its six ds_write_b64 go BETWEEN the 24 MFMAs, three VALU per MFMA: the bf16 MFMA
        // holds the issue port for 8 of its 32 cycles, so the fillers are free there, while as a burst in front of the MFMAs they were a third of the
        // step (timeline of scratch/ubench/gemm_x6_trace.hip: 0.89 us per step for a workgroup alone on its CU against 0.35 us of MFMA time).
        auto step = [&](int kt, auto st_tag, auto full_tag) {
            constexpr int ST = decltype(st_tag)::value, S2 = (ST + 2) % XNST;
            constexpr bool FULL = decltype(full_tag)::value;
            // EDV_X6_NOSTAGE / _NOMFMA / _NOBARRIER / _NOWAIT: ablation builds of scratch/ubench/gemm_x6_trace.hip (wrong results, timing only).  A workgroup
            // alone on its CU, K = 4096: 0.715 us per step; without the staging 0.503; with 4 MFMAs instead of 24 0.540; without the barrier 0.675; fragment
            // reads + 24 MFMAs alone 0.474 = 24 x 32 cycles at 1.62 GHz, the clock the part holds under these MFMAs -- the matrix pipe's real rate here.
#ifndef EDV_X6_NOSTAGE
            if (FULL || kt + 2 < kt1) issue_w(kt + 2, S2);
#endif
            bf16x8 a[2][3], b[2][3];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    a[i][p] = *reinterpret_cast<const bf16x8 *>(smem + ST * XSTAGE + fa[i] + p * XPLANE);
                    b[i][p] = *reinterpret_cast<const bf16x8 *>(smem + ST * XSTAGE + fb[i] + p * XPLANE);
                }
#ifndef EDV_X6_NOSTAGE
            if (FULL || kt + 2 < kt1) split_store_a(aset[ST], S2);  // A(kt + 2), loaded three steps ago
#endif
            constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};  // smallest terms first
#ifdef EDV_X6_NOMFMA
            constexpr int NT = 1;  // one MFMA per accumulator keeps the fragment reads alive
#else
            constexpr int NT = 6;
#endif
#pragma unroll
            for (int t = 6 - NT; t < 6; ++t)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][PA[t]], b[j][PB[t]], acc[i][j], 0, 0, 0);
#ifndef EDV_X6_NOSTAGE
            if (FULL || kt + 5 < kt1) load_a(kt + 5, aset[ST]);
#endif
            if (FULL) {
                __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);  // the fragment reads first
#pragma unroll
                for (int i = 0; i < 24; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA
                    __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);  // three VALU
                    if ((i & 3) == 3) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);  // a plane write after every fourth
                }
            }
            // W of step + 1 (issued one step ago) has landed: younger than it are A(step + 4) x2, W(step + 2) x3, A(step + 5) x2.  The wave's own LDS writes
            // and fragment reads are done (lgkmcnt) before the barrier that publishes the stage written and releases the stage read.
            // (sched_barrier on both sides: with the schedule spelled out above the machine scheduler otherwise moved the NEXT step's fragment reads above
            // these waits and the barrier -- reads of a stage other waves' DMAs have not yet been waited for.  The inline asm's memory clobber does not
            // stop it; tests/test_isa_barriers_cpu.py::test_x6_stage_reads_stay_behind_their_barrier checks the disassembly.)
            __builtin_amdgcn_sched_barrier(0);
#ifdef EDV_X6_NOWAIT  // (ablation, wrong results) is the W DMA's latency what a step waits for?  No: 18.3 us per 24-step k loop either way
            asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
#else
            if (FULL || kt + 5 < kt1)
                asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
            else
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifndef EDV_X6_NOBARRIER
            __builtin_amdgcn_s_barrier();
#endif
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        };
        {
            int kt = kt0;
            for (; kt + 7 < kt1; kt += 3) {  // kt + 2 + 5 < kt1: all three steps are full
                step(kt, std::integral_constant<int, 0>{}, std::true_type{});
                step(kt + 1, std::integral_constant<int, 1>{}, std::true_type{});
                step(kt + 2, std::integral_constant<int, 2>{}, std::true_type{});
            }
            for (; kt + 2 < kt1; kt += 3) {
                step(kt, std::integral_constant<int, 0>{}, std::false_type{});
                step(kt + 1, std::integral_constant<int, 1>{}, std::false_type{});
                step(kt + 2, std::integral_constant<int, 2>{}, std::false_type{});
            }
            if (kt < kt1) step(kt, std::integral_constant<int, 0>{}, std::false_type{});
            if (kt + 1 < kt1) step(kt + 1, std::integral_constant<int, 1>{}, std::false_type{});
        }

        EDV_X6_STAMP(2);
        bool store_tile = true;
        if (SPLIT && part) {
            // piece hand-off: the protocol of gemm_dma.hip (sc1 stores, every wave's vmcnt(0), barrier, one agent-scope counter add; the last arriver
            // acquires, then reads every piece with sc1 loads in run order)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        __hip_atomic_store(&part[((wave * 4 + i * 2 + j) * 16 + r) * 64 + lane], acc[i][j][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const long long ub = (long long)lt * nkt;
            const int g0 = (int)(ub / sp.chunk), g1 = (int)((ub + nkt - 1) / sp.chunk);
            int *s_last = reinterpret_cast<int *>(smem);  // all stages are idle between the k loop and the next run's first DMA
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                const int arrived = __hip_atomic_fetch_add(&sp.cnt[lt], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int last = arrived == g1 - g0;
                if (last) __hip_atomic_store(&sp.cnt[lt], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (last) split_merge_acquire();
                *s_last = last;
            }
            __syncthreads();
            store_tile = *s_last != 0;
            __syncthreads();
            if (store_tile) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
                for (int gg = g0; gg <= g1; ++gg) {  // run order, whatever the arrival order was
                    const float *pp = sp.ws + ((long long)gg * 2 + ((long long)gg * sp.chunk >= ub ? 0 : 1)) * XSLOT + (wave * 4 * 16) * 64 + lane;
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            float t[16];
#pragma unroll
                            for (int r = 0; r < 16; ++r) t[r] = __hip_atomic_load(pp + ((i * 2 + j) * 16 + r) * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                            for (int r = 0; r < 16; ++r) acc[i][j][r] += t[r];
                        }
                }
            }
        }
        // one call site for whole tiles and merged ones (the epilogue is most of the kernel's code: bias, activation, gamma, residual through buffer
        // descriptors, which also mask the ragged edges -- gemm_epilogue_buf, gemm_common.hpp)
        if (store_tile) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    EpiCols<1> c1;
                    c1.bias[0] = cols.bias[j];
                    c1.gam[0] = cols.gam[j];
                    gemm_epilogue_buf<ACT>(g, acc[i][j], c1, m0 + (wave_s >> 1) * 64 + i * 32, n0 + (wave_s & 1) * 64 + j * 32, l31, lh);
                }
        }
        EDV_X6_STAMP(3);
    }
}

template <int ACT>
int x6_slots() {
    static DeviceSlotCache cache;
    return cache.get([] {
        int dev = 0, cus = 0, per_cu = 0;
        if (hipGetDevice(&dev) != hipSuccess) return 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
        if (hipFuncSetAttribute((const void *)gemm_x6_kernel<ACT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, XNST * XSTAGE) != hipSuccess) return 0;
        if (hipFuncSetAttribute((const void *)gemm_x6_kernel<ACT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, XNST * XSTAGE) != hipSuccess) return 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gemm_x6_kernel<ACT, true>, 256, XNST * XSTAGE) != hipSuccess) return 0;
        if (per_cu > 2) per_cu = 2;  // 2 x 72 KB of LDS
        if (const char *e = getenv("EDV_X6_SLOTS_PER_CU")) per_cu = atoi(e) > 0 ? atoi(e) : per_cu;
        if (getenv("EDV_DEBUG_SLOTS")) fprintf(stderr, "gemm_x6_kernel<%d>: %d CUs x %d resident workgroups\n", ACT, cus, per_cu);
        return cus * per_cu;
    });
}

template <int ACT>
int launch_x6(const GemmDesc &d, long long tiles, hipStream_t st) {
    static const bool plain_forced = [] {
        const char *e = getenv("EDV_GEMM_PLAIN");
        return e && atoi(e) != 0;
    }();
    GemmSplit sp{1, 1, 0, 1, 0, nullptr, nullptr};
    static const int group_m = [] {
        const char *e = getenv("EDV_X6_GROUP_M");  // row blocks per tile group (A/B runs); 1 = row-major tile order
        return e && atoi(e) > 0 ? atoi(e) : 8;
    }();
    sp.group_m = group_m;
    long long grid = tiles;
    const int slots = x6_slots<ACT>();
    EDV_CHECK(slots > 0 && slots <= XMAX_COUNTERS, "occupancy query failed");
    const long long left = tiles % slots;
    const int nkt = d.K / XBK;
    // The split pays from K = 768 up (fc2 at T = 8: 110 -> 93 us); at K = 384 a tile is 24 steps, the pieces and their merge cost more than the uneven
    // last round, and the plain grid's dynamic dispatch wins (qkv 87 -> 78 us, proj 40 -> 35 us; profiles/r03_gemm_x6_shapes.txt).  Launching every
    // workgroup pair of a CU in lockstep is not what costs: alternating the order of whole tiles and runs between workgroups changed nothing.
    static const int min_kt = [] {
        const char *e = getenv("EDV_X6_SPLIT_MIN_KT");  // k-steps per tile from which the split is used (A/B runs)
        return e ? atoi(e) : 48;
    }();
    if (d.ws && !plain_forced && left > 0 && tiles > 16 && tiles < 8ll * slots && nkt >= min_kt) {
        sp.whole_rounds = (int)(tiles / slots);
        long long split_tiles = left;
        const long long chunk_min = (nkt + 3) / 4;
        if (sp.whole_rounds > 0 && (left * nkt + slots - 1) / slots < chunk_min && left + slots <= XMAX_COUNTERS) {
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
        sp.ws = d.ws + XMAX_COUNTERS;
        EDV_CHECK((size_t)XMAX_COUNTERS + (size_t)sp.nsplit * 2 * XSLOT <= d.ws_floats && (uintptr_t)d.ws % 16 == 0, "stream-K workspace too small (gemm_workspace)");
        EDV_LAUNCH((gemm_x6_kernel<ACT, true>), dim3((unsigned)grid), dim3(256), XNST * XSTAGE, st, d, sp);
        EDV_LAUNCH_OK();
        return 0;
    }
    EDV_LAUNCH((gemm_x6_kernel<ACT, false>), dim3((unsigned)grid), dim3(256), XNST * XSTAGE, st, d, sp);
    EDV_LAUNCH_OK();
    return 0;
}

}  // namespace

size_t gemm_x6_planes_bytes(int N, int K) { return (size_t)3 * N * K * 2; }

int gemm_x6_split(const float *W, void *planes, int N, int K, hipStream_t st) {
    EDV_CHECK(W && planes && N > 0 && K > 0, "null argument");
    const long long n = (long long)N * K;
    EDV_LAUNCH(split3_kernel, dim3((unsigned)std::min<long long>((n + 255) / 256, 2048)), dim3(256), 0, st, W, reinterpret_cast<__bf16 *>(planes), n);
    EDV_LAUNCH_OK();
    return 0;
}

bool gemm_x6_supported(const GemmDesc &d) {
    const int ek = epilogue_kind(d);  // 1 + ACT: identity row maps, no pre-activation addend, a compile-time activation
    if (!(d.Wx6 && d.loader == LOAD_DENSE && ek >= 1 && ek <= 3 && d.K % XBK == 0 && d.lda % 4 == 0 && d.ldw == d.K && d.M > 0 && d.N >= 64)) return false;
    if (!((long long)3 * d.N * d.K * 2 < (1ll << 32) - (1 << 20))) return false;  // the planes behind one buffer descriptor
    // the buffer epilogue's 32-bit offsets and descriptor masking (see fits_buffer in gemm_dma.hip): a tile's rows run up to 127 past M
    const long long ld = std::max(std::max(d.ldc, d.R1 ? d.ldr1 : 0), d.R2 ? d.ldr2 : 0);
    return (long long)(d.M + XBM) * ld * 4 < (1ll << 32) - 4096;
}

int gemm_x6(const GemmDesc &d, hipStream_t st) {
    EDV_CHECK(d.A && d.Wx6 && d.C, "null operand");
    EDV_CHECK(gemm_x6_supported(d), "bf16x6 GEMM: dense A, planes of a [N, K] weight, K % 16 == 0, identity row maps, outputs below 4 GB");
    EDV_CHECK(d.lda >= d.K, "leading dimension");
    EDV_CHECK(((uintptr_t)d.A % 16 == 0) && ((uintptr_t)d.Wx6 % 16 == 0), "A / planes must be 16-byte aligned");
    const long long tiles = ((d.M + XBM - 1) / XBM) * (long long)((d.N + XBN - 1) / XBN);
    EDV_CHECK(tiles > 0 && tiles < (1ll << 31), "bad grid");
    switch (epilogue_kind(d)) {
        case 1: return launch_x6<ACT_NONE>(d, tiles, st);
        case 2: return launch_x6<ACT_GELU>(d, tiles, st);
        default: return launch_x6<ACT_RELU>(d, tiles, st);
    }
}

}  // namespace edv
