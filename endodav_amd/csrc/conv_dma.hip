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
#include "gemm_common.hpp"

namespace edv {
namespace {

constexpr int CBK = 32;
__device__ __attribute__((aligned(256))) float g_zero_page[64];

template <int WGM, int EP>
__global__ __launch_bounds__(256) void conv3_dma_kernel(const GemmDesc g) {
    constexpr int WGN = 4 / WGM;
    constexpr int BM = 32 * WGM, BN = 32 * WGN;
    constexpr int IA = BM / 32, IB = BN / 32;  // DMA instructions per wave per k-tile (8 rows x 128 B each)
    constexpr int STAGE = (BM + BN) * CBK;
    __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const int wm = wave / WGN, wn = wave % WGN;
    const int tiles_n = (g.N + BN - 1) / BN;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
    const long long m0 = (long long)tm * BM;
    const int n0 = tn * BN;
    const int srow = lane >> 3, spos = lane & 7;
    const float *zero = g_zero_page + spos * 4;

    // per-lane A rows: output pixel -> pointer to the (dy, dx) = (0, 0) tap of its window, validity bits per dy and dx
    const float *pa[IA];
    int okmask[IA];  // bits 0..2: row iy0 + dy inside the image; bits 3..5: column ix0 + dx inside
    const int opix = g.cOH * g.cOW;
#pragma unroll
    for (int i = 0; i < IA; ++i) {
        const int r = (BM / 4) * wave + 8 * i + srow;
        const int c = spos ^ ((r >> 1) & 7);  // logical 16-byte chunk that lives at this LDS position
        long long m = m0 + r;
        m = m < g.M ? m : g.M - 1;  // rows past the edge read a valid pixel; their results are never stored
        const long long f = m / opix;
        const int p = (int)(m - f * opix);
        const int oy = p / g.cOW, ox = p - oy * g.cOW;
        const int iy0 = oy * g.cS - 1, ix0 = ox * g.cS - 1;
        pa[i] = g.A + ((f * g.cH + iy0) * (long long)g.cW + ix0) * g.cC + c * 4;
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
    }
    auto issue = [&](int kt, int st) {
        float *sA = smem + st * STAGE, *sB = sA + BM * CBK;
        const int k = kt * CBK;
        const int tap = k / g.cC, c0 = k - tap * g.cC;  // uniform: the whole k-tile lies inside one tap
        const int dy = tap / 3, dx = tap - dy * 3;
        const long long off = ((long long)dy * g.cW + dx) * g.cC + c0;
        const int need = (1 << dy) | (8 << dx);
#pragma unroll
        for (int i = 0; i < IA; ++i) {
            const float *src = (okmask[i] & need) == need ? pa[i] + off : zero;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(sA + ((BM / 4) * wave + 8 * i) * CBK), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < IB; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(pb[i] + k),
                                             (__attribute__((address_space(3))) void *)(sB + ((BN / 4) * wave + 8 * i) * CBK), 16, 0, 0);
    };

    EpiCols<1> cols;
    if (EP != 0) cols = gemm_epilogue_prefetch<1>(g, n0, wn * 32, l31);
    f32x16 acc[1][1];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;
    const int ra = wm * 32 + l31, rb = wn * 32 + l31;
    const int swa = (ra >> 1) & 7, swb = (rb >> 1) & 7;
    const int nkt = g.K / CBK;
    const bool prelu = g.pre_relu != 0;

    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int kt = 0; kt < nkt; ++kt) {
        const int st = kt & 1;
        if (kt + 1 < nkt) issue(kt + 1, st ^ 1);
        const float *sA = smem + st * STAGE, *sB = sA + BM * CBK;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int cq = 2 * q + lh;
            f32x4 fa = *reinterpret_cast<const f32x4 *>(&sA[ra * CBK + ((cq ^ swa) << 2)]);
            const f32x4 fb = *reinterpret_cast<const f32x4 *>(&sB[rb * CBK + ((cq ^ swb) << 2)]);
            if (prelu) {
                fa.x = fmaxf(fa.x, 0.f); fa.y = fmaxf(fa.y, 0.f); fa.z = fmaxf(fa.z, 0.f); fa.w = fmaxf(fa.w, 0.f);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[e], fb[e], acc[0][0], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    gemm_epilogue_ep<1, 1, STORE_ROWS, EP>(g, acc, cols, m0, n0, wm * 32, wn * 32, l31, lh);
}

template <int WGM>
int launch_conv(const GemmDesc &d, hipStream_t st) {
    constexpr int BM = 32 * WGM, BN = 32 * (4 / WGM);
    const long long tiles = ((d.M + BM - 1) / BM) * (long long)((d.N + BN - 1) / BN);
    EDV_CHECK(tiles > 0 && tiles < (1ll << 31), "bad grid");
    const dim3 grid((unsigned)tiles), block(256);
    switch (epilogue_kind(d)) {
        case 1: hipLaunchKernelGGL((conv3_dma_kernel<WGM, 1>), grid, block, 0, st, d); break;
        case 3: hipLaunchKernelGGL((conv3_dma_kernel<WGM, 3>), grid, block, 0, st, d); break;
        default: hipLaunchKernelGGL((conv3_dma_kernel<WGM, 0>), grid, block, 0, st, d); break;
    }
    EDV_LAUNCH_OK();
    return 0;
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
