// 3x3 convolution (padding 1, stride 1 or 2, channels-last) as an implicit GEMM with LDS-DMA staging.
//
// Same contract as the LOAD_CONV3 path of gemm.hip (util/blocks.py:20-32,68-91; dpt.py:86-90,117-124), for Cin % 32 == 0:
// then every 32-wide k-tile lies inside ONE tap (dy, dx) and is 128 contiguous bytes of one input pixel, which is exactly
// what a global_load_lds lane group moves.  The im2col gather happens in the per-lane SOURCE address; taps that fall into
// the zero padding read a 256-byte page of zeros instead.  No VGPR round trip, no address arithmetic per element, no
// ds_write: per k-tile a lane issues its DMAs with one 64-bit select each.  The pre-activation ReLU of the
// ResidualConvUnits is applied to the A fragments after the LDS read (4 v_max per 4 MFMAs).
//
// LDS image, swizzle and pipeline are those of gemm_dma.hip.  Tiles: 64x64 (2x2 waves) or, for Cout <= 32, 128x32 (4x1).
//
// Round 2 (BUF = true, operands within 4 GB): the k loop of gemm_dma.hip -- DMA through buffer descriptors with the tap / channel advance
// in the SCALAR offset, scalar wave index, loop unrolled by two so that the LDS stage is an immediate -- because the f32 MFMA and the
// VALU do not overlap on this part.  A padding tap is a lane whose byte offset is switched to one past the descriptor's range: an
// out-of-range lane of a buffer LDS-DMA writes zeros into LDS (checked on the part, scratch/ubench/README), so no zero page is read.
// What is left per k-tile: one compare + select per A-side DMA (is this lane's tap inside the image?) and, with pre_relu, the v_max of
// the A fragments.
#include <cstdlib>
#include <type_traits>

#include "gemm_common.hpp"

namespace edv {
namespace {

constexpr int CBK = 32;
__device__ __attribute__((aligned(256))) float g_zero_page[64];

// SPLIT: the stream-K scheme of gemm_dma.hip for grids that do not fill the part (tiles <= resident workgroups, e.g. layer4_rn: 46
// tiles of 108 k-tiles on 256 CUs): the tiles' k-tile units are cut into equal contiguous runs, one per workgroup; a run that does
// not cover a tile's whole k range leaves its accumulators in a workspace slot and the last piece of a tile to arrive merges them in
// run order and applies the epilogue.  SPLIT = false is the plain grid and compiles to the code it was before.
template <int WGM, int EP, bool SPLIT, bool BUF>
__global__ __launch_bounds__(256, 4) void conv3_dma_kernel(const GemmDesc g, const GemmSplit sp) {
    constexpr int WGN = 4 / WGM;
    constexpr int BM = 32 * WGM, BN = 32 * WGN;
    constexpr int IA = BM / 32, IB = BN / 32;  // DMA instructions per wave per k-tile (8 rows x 128 B each)
    constexpr int STAGE = (BM + BN) * CBK;
    __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    const bool prelu = g.pre_relu != 0;
    const int wm = wave / WGN, wn = wave % WGN;
    const int tiles_n = (g.N + BN - 1) / BN;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    // BUF: descriptors over the input tensor (exactly its bytes: a lane sent past the end reads zeros) and the packed weight
    const unsigned a_bytes = BUF ? (unsigned)((g.M > 0 ? ((g.M - 1) / (g.cOH * g.cOW) + 1) : 0) * (long long)g.cH * g.cW * g.cC * 4) : 0u;
    const auto rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(g.A), 0, a_bytes, 0x00020000);
    const auto rW = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(g.W), 0, 0xffffffff, 0x00020000);
    const int srow = lane >> 3, spos = lane & 7;
    const float *zero = g_zero_page + spos * 4;
    const int opix = g.cOH * g.cOW;
    const int nkt = g.K / CBK;
    const int ra = wm * 32 + l31, rb = wn * 32 + l31;
    const int swa = (ra >> 1) & 7, swb = (rb >> 1) & 7;

    // SPLIT: run `bid` of the split (stride 1: every workgroup of the grid owns one); plain: one whole tile
    long long u = SPLIT ? (long long)bid * sp.chunk : 0;
    const long long u_end = SPLIT ? (u + sp.chunk < sp.units ? u + sp.chunk : sp.units) : 0;
    bool once = true;
    for (;;) {
        int tile, kt0, kt1, lt = 0;
        float *part = nullptr;
        if (SPLIT) {
            if (u >= u_end) break;
            lt = (int)(u / nkt);
            kt0 = (int)(u - (long long)lt * nkt);
            const long long left = u_end - u;
            kt1 = kt0 + left < nkt ? kt0 + (int)left : nkt;
            tile = lt;
            // slot 0 = the piece that holds the run's first unit, slot 1 = the other one (the rule of gemm_dma.hip)
            if (!(kt0 == 0 && kt1 == nkt)) part = sp.ws + ((long long)bid * 2 + (u == (long long)bid * sp.chunk ? 0 : 1)) * SPLIT_SLOT;
            u += kt1 - kt0;
        } else {
            if (!once) break;
            once = false;
            tile = bid;
            kt0 = 0;
            kt1 = nkt;
        }
        const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
        const long long m0 = (long long)tm * BM;
        const int n0 = tn * BN;

        // per-lane A rows: output pixel -> pointer to the (dy, dx) = (0, 0) tap of its window, validity bits per dy and dx
        const float *pa[IA];
        unsigned va[IA], vb[IB];  // BUF: byte offsets (A: of the window's centre tap)
        int okmask[IA];  // bits 0..2: row iy0 + dy inside the image; bits 3..5: column ix0 + dx inside
#pragma unroll
        for (int i = 0; i < IA; ++i) {
            const int r = (BM / 4) * wave + 8 * i + srow;
            const int c = spos ^ ((r >> 1) & 7);  // logical 16-byte chunk that lives at this LDS position
            long long m = m0 + r;
            m = m < g.M ? m : g.M - 1;  // rows past the edge read a valid pixel; their results are never stored
            // (BUF: M < 2^30, so the pixel decode is 32-bit arithmetic -- a 64-bit division is ~95 VALU instructions per row)
            const long long f = BUF ? (long long)((unsigned)m / (unsigned)opix) : m / opix;
            const int p = BUF ? (int)((unsigned)m - (unsigned)f * (unsigned)opix) : (int)(m - f * opix);
            const int oy = p / g.cOW, ox = p - oy * g.cOW;
            const int iy0 = oy * g.cS - 1, ix0 = ox * g.cS - 1;
            pa[i] = g.A + ((f * g.cH + iy0) * (long long)g.cW + ix0) * g.cC + c * 4;
            va[i] = (unsigned)((((f * g.cH + iy0 + 1) * (long long)g.cW + ix0 + 1) * g.cC + c * 4) * 4);  // the window's CENTRE tap: always inside
            int mk = 0;
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                if (iy0 + d >= 0 && iy0 + d < g.cH) mk |= 1 << d;
                if (ix0 + d >= 0 && ix0 + d < g.cW) mk |= 8 << d;
            }
            okmask[i] = mk;
        }
        const float *pb[IB];
#pragma unroll
        for (int i = 0; i < IB; ++i) {
            const int r = (BN / 4) * wave + 8 * i + srow;
            const int c = spos ^ ((r >> 1) & 7);
            int n = n0 + r;
            n = n < g.N ? n : g.N - 1;
            pb[i] = g.W + (long long)n * g.ldw + c * 4;
            vb[i] = (unsigned)(((long long)n * g.ldw + c * 4) * 4);
        }
        // tap state of the NEXT k-tile to stage (uniform; advanced by 32 channels per call instead of dividing k by Cin every time)
        int n_c0, n_dy, n_dx;
        {
            const int k = kt0 * CBK, tap = k / g.cC;
            n_c0 = k - tap * g.cC;
            n_dy = tap / 3;
            n_dx = tap - n_dy * 3;
        }
        auto issue = [&](int kt, int st) {
            float *sA = smem + st * STAGE, *sB = sA + BM * CBK;
            const int k = kt * CBK;
            const int dy = n_dy, dx = n_dx, c0 = n_c0;
            n_c0 += CBK;
            if (n_c0 == g.cC) {
                n_c0 = 0;
                if (++n_dx == 3) {
                    n_dx = 0;
                    ++n_dy;
                }
            }
            const long long off = ((long long)dy * g.cW + dx) * g.cC + c0;
            const int need = (1 << dy) | (8 << dx);
#pragma unroll
            for (int i = 0; i < IA; ++i) {
                if constexpr (BUF) {
                    // valid tap: the centre tap's byte offset plus the tap's (signed, uniform) distance from it -- the sum is formed here, in 32
                    // bits, because the descriptor path adds voffset + soffset without wrapping; the channel advance rides in the scalar
                    // operand.  Padding tap: an offset beyond the descriptor -> the lane writes zeros.
                    const int delta = (((dy - 1) * g.cW + (dx - 1)) * g.cC) * 4;
                    const unsigned vo = (okmask[i] & need) == need ? va[i] + (unsigned)delta : 0x80000000u + a_bytes;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (__attribute__((address_space(3))) void *)(sA + ((BM / 4) * wave_s + 8 * i) * CBK), 16, (unsigned)vo,
                                                             (int)(c0 * 4), 0, 0);
                } else {
                    const float *src = (okmask[i] & need) == need ? pa[i] + off : zero;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                     (__attribute__((address_space(3))) void *)(sA + ((BM / 4) * wave + 8 * i) * CBK), 16, 0, 0);
                }
            }
#pragma unroll
            for (int i = 0; i < IB; ++i) {
                if constexpr (BUF)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (__attribute__((address_space(3))) void *)(sB + ((BN / 4) * wave_s + 8 * i) * CBK), 16, (unsigned)vb[i], (int)(k * 4), 0,
                                                             0);
                else
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(pb[i] + k),
                                                     (__attribute__((address_space(3))) void *)(sB + ((BN / 4) * wave + 8 * i) * CBK), 16, 0, 0);
            }
        };
        const float *fpa[4], *fpb[4];  // fragment read addresses in stage 0
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            fpa[q] = smem + ra * CBK + (((2 * q + lh) ^ swa) << 2);
            fpb[q] = smem + BM * CBK + rb * CBK + (((2 * q + lh) ^ swb) << 2);
        }

        EpiCols<1> cols;
        if (EP != 0) cols = gemm_epilogue_prefetch<1>(g, n0, wn * 32, l31);
        f32x16 acc[1][1];
        auto epilogue = [&]() {
            if constexpr (BUF && EP >= 1 && EP <= 3)  // addresses in scalar registers, edge rows / columns masked by the descriptor (gemm_common.hpp)
                gemm_epilogue_buf<EP - 1>(g, acc[0][0], cols, m0 + (wave_s / WGN) * 32, n0 + (wave_s % WGN) * 32, l31, lh);
            else
                gemm_epilogue_ep<1, 1, STORE_ROWS, EP>(g, acc, cols, m0, n0, wm * 32, wn * 32, l31, lh);
        };
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;

        issue(kt0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        auto ktile = [&](int kt, auto st_tag) {
            constexpr int ST = decltype(st_tag)::value;
            if (kt + 1 < kt1) issue(kt + 1, ST ^ 1);
            f32x4 fa[4], fb[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                fa[q] = *reinterpret_cast<const f32x4 *>(fpa[q] + ST * STAGE);
                fb[q] = *reinterpret_cast<const f32x4 *>(fpb[q] + ST * STAGE);
            }
            __builtin_amdgcn_sched_barrier(0);  // all eight reads in flight before the first MFMA waits (gemm_dma.hip)
            if (prelu) {  // one uniform branch per k-tile; one v_med3 per element: med3(x, 0, +inf) = max(x, 0) (fmaxf costs two instructions)
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) fa[q][e] = __builtin_amdgcn_fmed3f(fa[q][e], 0.f, __builtin_inff());
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q][e], fb[q][e], acc[0][0], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);  // the MFMAs stay above the wait for the next tile's DMA
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
        if (SPLIT && part) {
            // the piece exchange of gemm_dma.hip: agent-scope relaxed atomics (sc1 write-through stores / L2-bypassing loads), an
            // arrival counter per tile, the last piece to arrive sums all pieces in run order
#pragma unroll
            for (int r = 0; r < 16; ++r) __hip_atomic_store(&part[(wave * 16 + r) * 64 + lane], acc[0][0][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const long long ub = (long long)lt * nkt;
            const int g0 = (int)(ub / sp.chunk), g1 = (int)((ub + nkt - 1) / sp.chunk);
            int *s_last = reinterpret_cast<int *>(smem);  // both LDS stages are idle here
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                const int arrived = __hip_atomic_fetch_add(&sp.cnt[lt], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int last = arrived == g1 - g0;
                if (last) __hip_atomic_store(&sp.cnt[lt], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (last) split_merge_acquire();  // this CU's L1 may hold stale lines of the piece slots (gemm_common.hpp)
                *s_last = last;
            }
            __syncthreads();
            const bool last = *s_last != 0;
            __syncthreads();
            if (last) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;
                const float *base = sp.ws + (wave * 16) * 64 + lane;
                int gg = g0;
                for (; gg + 1 <= g1; gg += 2) {
                    const float *qa = base + ((long long)gg * 2 + ((long long)gg * sp.chunk >= ub ? 0 : 1)) * SPLIT_SLOT;
                    const float *qb = base + ((long long)(gg + 1) * 2 + ((long long)(gg + 1) * sp.chunk >= ub ? 0 : 1)) * SPLIT_SLOT;
                    float ta[16], tb[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        ta[r] = __hip_atomic_load(qa + r * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        tb[r] = __hip_atomic_load(qb + r * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[0][0][r] = (acc[0][0][r] + ta[r]) + tb[r];
                }
                if (gg <= g1) {
                    const float *qa = base + ((long long)gg * 2 + ((long long)gg * sp.chunk >= ub ? 0 : 1)) * SPLIT_SLOT;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[0][0][r] += __hip_atomic_load(qa + r * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                epilogue();
            }
        } else {
            // (its own call site on purpose, see gemm_dma.hip)
            epilogue();
        }
    }
}

template <int WGM, int EP, bool SPLIT>
void launch_variant(const GemmDesc &d, const GemmSplit &sp, bool buf, unsigned grid, hipStream_t st) {
    if (buf) EDV_LAUNCH((conv3_dma_kernel<WGM, EP, SPLIT, true>), dim3(grid), dim3(256), 0, st, d, sp);
    else EDV_LAUNCH((conv3_dma_kernel<WGM, EP, SPLIT, false>), dim3(grid), dim3(256), 0, st, d, sp);
}

// (Build note, hipcc 7.2: in the HOST pass an amdgcn builtin whose integer arguments need an implicit conversion from a captured lvalue makes
// the kernel specialization silently invalid -- "no matching function" at the launch, or an undefined __device_stub__ at load time -- while
// the device pass compiles it.  Hence the explicit (unsigned) / (int) casts on the buffer-load offsets.)
template <int WGM, int EP>
int conv_slots_query() {
    int dev = 0, cus = 0, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    void (*kern)(const GemmDesc, const GemmSplit) = conv3_dma_kernel<WGM, EP, true, true>;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 256, 0) != hipSuccess) return 0;
    if (per_cu > 4) per_cu = 4;  // 32 KB of LDS: the API says 5, the part places 4 (scratch/ubench/lds_residency.hip)
    return cus * per_cu;
}
template <int WGM, int EP>
int conv_slots() {
    static DeviceSlotCache cache;
    return cache.get([] { return conv_slots_query<WGM, EP>(); });
}

template <int WGM, int EP>
int launch_conv_ep(const GemmDesc &d, long long tiles, hipStream_t st) {
    static const bool split_on = [] {
        const char *e = getenv("EDV_CONV_SPLIT");  // 0: plain grid always (A/B runs)
        return !(e && atoi(e) == 0);
    }();
    static const bool buf_off = [] {
        const char *e = getenv("EDV_CONV_BUF");  // 0: flat 64-bit DMA source addresses + zero page as in round 1 (A/B runs)
        return e && atoi(e) == 0;
    }();
    // 32-bit byte offsets must reach every tap of every pixel, with room for the "beyond the end" offset of padding lanes
    const long long frames = (d.M - 1) / ((long long)d.cOH * d.cOW) + 1;
    // ... and (the buffer epilogue) every element of C, R1 and R2
    const long long ld_out = std::max(std::max(d.ldc, d.R1 ? d.ldr1 : 0), d.R2 ? d.ldr2 : 0);
    const bool buf = !buf_off && frames * d.cH * d.cW * d.cC * 4 < (1ll << 31) - (1 << 20) && (long long)d.N * d.ldw * 4 < (1ll << 32) - (1 << 20) &&
                     (d.M + 64) * ld_out * 4 < (1ll << 32) - 4096;  // rows up to 63 past M must not wrap the 32-bit scalar offsets (gemm_dma.hip, fits_buffer)
    const int nkt = d.K / CBK;
    GemmSplit sp{0, 1, 0, 1, 0, nullptr, nullptr};
    const int slots = conv_slots<WGM, EP>();
    // grids that do not fill the part and whose tiles are deep (Cin >= 192: 54+ k-tiles), or that leave most of it idle (8 tiles per
    // resident slot).  Measured (scratch/kb_conv_split.py, plain / split us): layer4_rn 46 tiles x 108 k-tiles 91 / 31, resize_layers.3
    // 276 x 108: 144 / 89, layer3_rn 172 x 54: 48 / 38, RCU at 19x19 46 x 18: 18.9 / 14.2 -- but RCU at 37x37 172 x 18: 19.3 / 24.9 and at
    // 74x74 685 x 18: 39.8 / 46.4, so shallow tiles on a half-full grid stay plain.
    if (split_on && d.ws && slots > 0 && tiles > 16 && tiles <= slots && tiles <= SPLIT_MAX_COUNTERS && nkt >= 18 &&
        (nkt >= 48 || tiles * 8 <= slots)) {
        sp.units = tiles * nkt;
        long long chunk = (sp.units + slots - 1) / slots;
        const long long chunk_min = (nkt + 3) / 4;
        chunk = chunk > chunk_min ? chunk : chunk_min;
        sp.chunk = (int)chunk;
        sp.nsplit = (int)((sp.units + chunk - 1) / chunk);
        sp.cnt = reinterpret_cast<int *>(d.ws);
        sp.ws = d.ws + SPLIT_MAX_COUNTERS;
        if ((size_t)SPLIT_MAX_COUNTERS + (size_t)sp.nsplit * 2 * SPLIT_SLOT <= d.ws_floats && (uintptr_t)d.ws % 16 == 0) {
            launch_variant<WGM, EP, true>(d, sp, buf, (unsigned)sp.nsplit, st);
            EDV_LAUNCH_OK();
            return 0;
        }
    }
    launch_variant<WGM, EP, false>(d, sp, buf, (unsigned)tiles, st);
    EDV_LAUNCH_OK();
    return 0;
}

template <int WGM>
int launch_conv(const GemmDesc &d, hipStream_t st) {
    constexpr int BM = 32 * WGM, BN = 32 * (4 / WGM);
    const long long tiles = ((d.M + BM - 1) / BM) * (long long)((d.N + BN - 1) / BN);
    EDV_CHECK(tiles > 0 && tiles < (1ll << 31), "bad grid");
    switch (epilogue_kind(d)) {
        case 1: return launch_conv_ep<WGM, 1>(d, tiles, st);
        case 3: return launch_conv_ep<WGM, 3>(d, tiles, st);
        default: return launch_conv_ep<WGM, 0>(d, tiles, st);
    }
}

}  // namespace

bool conv_dma_supported(const GemmDesc &d) {
    return d.loader == LOAD_CONV3 && d.store == STORE_ROWS && d.cC % CBK == 0 && d.K == 9 * d.cC && d.ldw % 4 == 0 && d.M > 0 && d.N > 0;
}

int conv_dma(const GemmDesc &d, hipStream_t st) {
    EDV_CHECK(d.A && d.W && d.C, "null operand");
    EDV_CHECK(conv_dma_supported(d), "LDS-DMA convolution needs Cin % 32 == 0");
    EDV_CHECK(((uintptr_t)d.A % 16 == 0) && ((uintptr_t)d.W % 16 == 0), "A/W must be 16-byte aligned");
    return d.N <= 32 ? launch_conv<4>(d, st) : launch_conv<2>(d, st);
}

}  // namespace edv
