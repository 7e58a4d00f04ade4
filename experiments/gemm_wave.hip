// Wave-private fp32 GEMM: ONE wave per workgroup, its own LDS staging area, NO workgroup barriers.
//
// Why: in gemm.hip four waves (on four SIMDs) share one LDS tile and meet at a barrier every k-tile.  With only ~3
// co-resident workgroups per CU that coupling leaves the MFMA pipe 27-46 % idle although nothing else is saturated
// (profiles/r01_pmc_*.csv).  An fp32 MFMA takes 64 cycles, so a single wave can afford to stage its own operands:
// each wave owns a WM x WN output tile (64x32 -> 32 MFMAs = 2048 pipe cycles per 32-k tile), loads the next k-tile
// into registers while it multiplies the current one, and rewrites its private LDS tile when it is done reading it.
// Waves never wait for each other; the SIMD's other resident waves (2-3) fill the short staging gap.  The price is
// that A/W tiles are not shared between waves (about 2x the L2->LDS traffic of the 64x64 shared tile).
#include "gemm_common.hpp"

namespace edv {
namespace {

constexpr int WBK = 32;
constexpr int WLS = WBK + 4;  // padded LDS row (floats): conflict-free ds_read_b128, as in gemm.hip

template <int WM, int WN, int LOADER, int STORE>
__global__ __launch_bounds__(64) void gemm_wave_kernel(const GemmDesc g) {
    constexpr int FM = WM / 32, FN = WN / 32;
    constexpr int RA = WM / 8, RB = WN / 8;  // float4 loads per lane per k-tile (8 lanes cover one 32-float row)
    __shared__ __attribute__((aligned(16))) float smem[(WM + WN) * WLS];
    float *sA = smem;
    float *sB = smem + WM * WLS;

    const int lane = threadIdx.x;
    const int l31 = lane & 31, lh = lane >> 5;
    const int tiles_n = (g.N + WN - 1) / WN;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
    const long long m0 = (long long)tm * WM;
    const int n0 = tn * WN;

    const int c = lane & 7, r0 = lane >> 3;
    const float *a_ptr[RA];
    int a_iy[RA], a_ix[RA];
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        const long long m = m0 + r0 + 8 * i;
        a_iy[i] = a_ix[i] = 0;
        if (m < g.M) {
            if (LOADER == LOAD_DENSE) {
                a_ptr[i] = g.A + g.a_map(m) * g.lda;
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
        }
    }
    const float *b_ptr[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        const int n = n0 + r0 + 8 * i;
        b_ptr[i] = (n < g.N) ? g.W + (long long)n * g.ldw : nullptr;
    }

    f32x4 ra[RA], rb[RB];
    auto load_tile = [&](int kt) {
        const int k = kt * WBK + c * 4;
        const bool kin = k < g.K;
        if (LOADER == LOAD_DENSE) {
#pragma unroll
            for (int i = 0; i < RA; ++i) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (kin && a_ptr[i]) v = *reinterpret_cast<const f32x4 *>(a_ptr[i] + k);
                ra[i] = v;
            }
        } else {
            const int tap = k / g.cC, ci = k - tap * g.cC;
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

    f32x16 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nkt = (g.K + WBK - 1) / WBK;
    load_tile(0);
    for (int kt = 0; kt < nkt; ++kt) {
        // single-wave workgroup: the barrier is only a compiler/LDS ordering point (no other wave to wait for)
        __syncthreads();
#pragma unroll
        for (int i = 0; i < RA; ++i) *reinterpret_cast<f32x4 *>(&sA[(r0 + 8 * i) * WLS + c * 4]) = ra[i];
#pragma unroll
        for (int i = 0; i < RB; ++i) *reinterpret_cast<f32x4 *>(&sB[(r0 + 8 * i) * WLS + c * 4]) = rb[i];
        __syncthreads();
        if (kt + 1 < nkt) load_tile(kt + 1);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 fa[FM], fb[FN];
#pragma unroll
            for (int i = 0; i < FM; ++i) fa[i] = *reinterpret_cast<const f32x4 *>(&sA[(i * 32 + l31) * WLS + 8 * q + 4 * lh]);
#pragma unroll
            for (int j = 0; j < FN; ++j) fb[j] = *reinterpret_cast<const f32x4 *>(&sB[(j * 32 + l31) * WLS + 8 * q + 4 * lh]);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][e], fb[j][e], acc[i][j], 0, 0, 0);
        }
    }
    gemm_epilogue<FM, FN, STORE>(g, acc, m0, n0, 0, 0, l31, lh);
}

template <int WM, int WN>
int launch_wave(const GemmDesc &d, hipStream_t st) {
    const long long tiles = ((d.M + WM - 1) / WM) * (long long)((d.N + WN - 1) / WN);
    EDV_CHECK(tiles > 0 && tiles < (1ll << 31), "bad grid");
    dim3 grid((unsigned)tiles), block(64);
    if (d.loader == LOAD_DENSE && d.store == STORE_ROWS)
        hipLaunchKernelGGL((gemm_wave_kernel<WM, WN, LOAD_DENSE, STORE_ROWS>), grid, block, 0, st, d);
    else if (d.loader == LOAD_CONV3 && d.store == STORE_ROWS)
        hipLaunchKernelGGL((gemm_wave_kernel<WM, WN, LOAD_CONV3, STORE_ROWS>), grid, block, 0, st, d);
    else if (d.loader == LOAD_DENSE && d.store == STORE_SHUFFLE)
        hipLaunchKernelGGL((gemm_wave_kernel<WM, WN, LOAD_DENSE, STORE_SHUFFLE>), grid, block, 0, st, d);
    else
        EDV_CHECK(false, "unsupported loader/store combination");
    EDV_LAUNCH_OK();
    return 0;
}

}  // namespace

int gemm_wave(const GemmDesc &d, int variant, hipStream_t st) {
    switch (variant) {
        case 1: return launch_wave<64, 64>(d, st);
        case 2: return launch_wave<32, 32>(d, st);
        case 3: return launch_wave<32, 64>(d, st);
        default: return launch_wave<64, 32>(d, st);
    }
}

}  // namespace edv
