// Split-bf16 GEMM: fp32-equivalent accuracy on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16).
//
// Every fp32 operand is written as the EXACT sum of three bf16 pieces, x = x1 + x2 + x3 (round-to-nearest at each
// step: x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2); |x2| <= 2^-9 |x|, |x3| <= 2^-17 |x|).  A product
// a*b is then the sum of nine bf16 x bf16 products, each EXACT in the fp32 accumulator; the six with i + j <= 4
//     a1b1, a1b2, a2b1, a1b3, a2b2, a3b1
// carry everything down to 2^-25 |ab| (the dropped a2b3 + a3b2 + a3b3 are below one fp32 ulp of the product), so
// the result has the accuracy of an fp32 FMA chain.  Cost: 6 bf16 MFMAs (32 cycles, 16 k each) per 32x32x16 block
// = 12 cycles per k against 32 cycles per k on v_mfma_f32_32x32x2_f32: a 2.67x higher ceiling (2.5 PF / 6 = 417
// "fp32-equivalent" TFLOP/s) with no range restriction (bf16 has the fp32 exponent).
//
// Same interface and epilogues as gemm.hip (GemmDesc).  W arrives pre-split (three bf16 planes laid out like W,
// made once per edv_prepare by split_planes); A stays fp32 in HBM and is split while it is staged to LDS
// (5.5 VALU ops per element, on the VALU pipe beside the MFMAs).
//
// Tile: 256 threads = 4 waves; block tile BM x BN x 32, one LDS stage, next tile prefetched into registers.
// LDS rows are 32 bf16 + 8 pad = 80 bytes: the ds_read_b128 fragment reads (lane = row, 16 bytes at k = 8h..8h+7 of
// the 16-k step) are conflict-free (20r mod 64 is a bijection on r mod 16).
#include <cstdlib>

#include "gemm_common.hpp"

namespace edv {
namespace {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
typedef unsigned int u32;
struct u32x2 { u32 x, y; };
using u32x4 = __attribute__((ext_vector_type(4))) u32;

constexpr int SBK = 32;    // k per LDS tile (two 16-k MFMA steps)
constexpr int ROWB = 80;   // bytes per LDS row: 64 data + 16 pad

__device__ __forceinline__ u32 pack_bf16(float lo, float hi) {  // v_cvt_pk_bf16_f32, round-to-nearest-even
    bf16x2 v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(u32, v);
}
__device__ __forceinline__ float bf_lo(u32 p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float bf_hi(u32 p) { return __builtin_bit_cast(float, p & 0xffff0000u); }

// four consecutive fp32 -> three planes of four bf16 (8 bytes each)
__device__ __forceinline__ void split4(const f32x4 v, u32x2 &p1, u32x2 &p2, u32x2 &p3) {
    p1.x = pack_bf16(v.x, v.y);
    p1.y = pack_bf16(v.z, v.w);
    const float r0 = v.x - bf_lo(p1.x), r1 = v.y - bf_hi(p1.x), r2 = v.z - bf_lo(p1.y), r3 = v.w - bf_hi(p1.y);
    p2.x = pack_bf16(r0, r1);
    p2.y = pack_bf16(r2, r3);
    p3.x = pack_bf16(r0 - bf_lo(p2.x), r1 - bf_hi(p2.x));
    p3.y = pack_bf16(r2 - bf_lo(p2.y), r3 - bf_hi(p2.y));
}

template <int BM, int BN, int WGM, int WGN, int LOADER, int STORE>
__global__ __launch_bounds__(256) void gemm_sb_kernel(const GemmDesc g) {
    constexpr int WTM = BM / WGM, WTN = BN / WGN;
    constexpr int FM = WTM / 32, FN = WTN / 32;
    constexpr int RA = BM / 32;               // fp32 float4 loads of A per thread per tile (8 chunks of 4 k per row)
    constexpr int WCH = BN * 4;               // 16-byte chunks per W plane per tile (4 chunks of 8 k per row)
    constexpr int RW = (WCH + 255) / 256;     // W chunks per thread per plane
    static_assert(WGM * WGN == 4 && FM >= 1 && FN >= 1, "4 waves");
    constexpr int A_PLANE = BM * ROWB, B_PLANE = BN * ROWB;

    __shared__ __attribute__((aligned(16))) unsigned char smem[3 * (A_PLANE + B_PLANE)];
    unsigned char *sA = smem;                 // plane p at sA + p * A_PLANE
    unsigned char *sB = smem + 3 * A_PLANE;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave / WGN, wn = wave % WGN;

    const int tiles_n = (g.N + BN - 1) / BN;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
    const long long m0 = (long long)tm * BM;
    const int n0 = tn * BN;

    // ---- A load slots (fp32): chunk c of 4 k, rows r0 + 32 i ----
    const int c = tid & 7, r0 = tid >> 3;
    const float *a_ptr[RA];
    int a_iy[RA], a_ix[RA];
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        const long long m = m0 + r0 + 32 * i;
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
    // ---- W load slots (bf16 planes): chunk cw of 8 k, row rw ----
    const unsigned short *w_ptr[RW];
    int w_row[RW], w_c[RW];
#pragma unroll
    for (int i = 0; i < RW; ++i) {
        const int idx = tid + 256 * i;
        w_row[i] = idx >> 2;
        w_c[i] = idx & 3;
        const int n = n0 + w_row[i];
        w_ptr[i] = (idx < WCH && n < g.N) ? g.Wsb + (long long)n * g.ldw : nullptr;
    }

    // one register set: tile k+1 is in flight while tile k is multiplied.  (A second set -- two tiles in flight --
    // was measured SLOWER: 140 -> 204 registers per lane halves the resident waves, profiles/r01_gemm_tile_sweep.txt.)
    f32x4 ra0[RA];
    u32x4 rw0[3][RW];
    auto load_tile = [&](int kt, f32x4(&ra)[RA], u32x4(&rw)[3][RW]) {
        const int k = kt * SBK + c * 4;
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
        for (int i = 0; i < RW; ++i) {
            const int kw = kt * SBK + w_c[i] * 8;
            const bool in = w_ptr[i] && kw < g.K;  // K % 8 == 0: a chunk is wholly in or out
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                u32x4 v = {0u, 0u, 0u, 0u};
                if (in) v = *reinterpret_cast<const u32x4 *>(w_ptr[i] + (long long)p * g.wsb_plane + kw);
                rw[p][i] = v;
            }
        }
    };

    f32x16 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto stage = [&](const f32x4(&ra)[RA], const u32x4(&rw)[3][RW]) {
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            u32x2 p1, p2, p3;
            split4(ra[i], p1, p2, p3);
            const int off = (r0 + 32 * i) * ROWB + c * 8;
            *reinterpret_cast<u32x2 *>(sA + off) = p1;
            *reinterpret_cast<u32x2 *>(sA + A_PLANE + off) = p2;
            *reinterpret_cast<u32x2 *>(sA + 2 * A_PLANE + off) = p3;
        }
#pragma unroll
        for (int i = 0; i < RW; ++i) {
            if (tid + 256 * i < WCH) {
                const int off = w_row[i] * ROWB + w_c[i] * 16;
#pragma unroll
                for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x4 *>(sB + p * B_PLANE + off) = rw[p][i];
            }
        }
    };
    auto multiply = [&]() {
#pragma unroll
        for (int step = 0; step < 2; ++step) {
            bf16x8 fa[3][FM], fb[3][FN];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
#pragma unroll
                for (int i = 0; i < FM; ++i)
                    fa[p][i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4 *>(sA + p * A_PLANE + (wm * WTM + i * 32 + l31) * ROWB + 32 * step + 16 * lh));
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    fb[p][j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4 *>(sB + p * B_PLANE + (wn * WTN + j * 32 + l31) * ROWB + 32 * step + 16 * lh));
            }
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    f32x16 a = acc[i][j];
                    // smallest terms first
                    a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[2][i], fb[0][j], a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][i], fb[1][j], a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][i], fb[2][j], a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][i], fb[0][j], a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][i], fb[1][j], a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][i], fb[0][j], a, 0, 0, 0);
                    acc[i][j] = a;
                }
        }
    };

    const int nkt = (g.K + SBK - 1) / SBK;
    load_tile(0, ra0, rw0);
    for (int kt = 0; kt < nkt; ++kt) {
        __syncthreads();  // the previous tile's fragment reads are done
        stage(ra0, rw0);
        __syncthreads();
        if (kt + 1 < nkt) load_tile(kt + 1, ra0, rw0);  // in flight under the MFMAs below
        multiply();
    }
    gemm_epilogue<FM, FN, STORE>(g, acc, m0, n0, wm * WTM, wn * WTN, l31, lh);
}

template <int BM, int BN, int WGM, int WGN>
int launch_sb(const GemmDesc &d, hipStream_t st) {
    const long long tiles = ((d.M + BM - 1) / BM) * (long long)((d.N + BN - 1) / BN);
    EDV_CHECK(tiles > 0 && tiles < (1ll << 31), "bad grid");
    dim3 grid((unsigned)tiles), block(256);
    if (d.loader == LOAD_DENSE && d.store == STORE_ROWS)
        hipLaunchKernelGGL((gemm_sb_kernel<BM, BN, WGM, WGN, LOAD_DENSE, STORE_ROWS>), grid, block, 0, st, d);
    else if (d.loader == LOAD_CONV3 && d.store == STORE_ROWS)
        hipLaunchKernelGGL((gemm_sb_kernel<BM, BN, WGM, WGN, LOAD_CONV3, STORE_ROWS>), grid, block, 0, st, d);
    else if (d.loader == LOAD_DENSE && d.store == STORE_SHUFFLE)
        hipLaunchKernelGGL((gemm_sb_kernel<BM, BN, WGM, WGN, LOAD_DENSE, STORE_SHUFFLE>), grid, block, 0, st, d);
    else
        EDV_CHECK(false, "unsupported loader/store combination");
    EDV_LAUNCH_OK();
    return 0;
}

// W [rows, ld] fp32 -> three bf16 planes with the same [rows, ld] layout
__global__ void split_planes_kernel(const float *__restrict__ w, unsigned short *__restrict__ out, long long n, long long plane) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float x = w[i];
        const __bf16 b1 = (__bf16)x;
        const float r1 = x - (float)b1;
        const __bf16 b2 = (__bf16)r1;
        const __bf16 b3 = (__bf16)(r1 - (float)b2);
        out[i] = __builtin_bit_cast(unsigned short, b1);
        out[plane + i] = __builtin_bit_cast(unsigned short, b2);
        out[2 * plane + i] = __builtin_bit_cast(unsigned short, b3);
    }
}

}  // namespace

bool gemm_sb_supported(const GemmDesc &d) { return d.K % 8 == 0 && d.ldw % 8 == 0 && d.N >= 1; }

int gemm_sb(const GemmDesc &d, hipStream_t st) {
    EDV_CHECK(d.A && d.Wsb && d.C, "null operand");
    EDV_CHECK(d.M > 0 && d.N > 0 && d.K > 0, "empty problem");
    EDV_CHECK(gemm_sb_supported(d), "split-bf16 GEMM needs K % 8 == 0 and ldw % 8 == 0");
    EDV_CHECK(d.wsb_plane >= (long long)d.N * d.ldw - (d.ldw - d.K), "plane stride too small");
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
    EDV_CHECK(((uintptr_t)d.A % 16 == 0) && ((uintptr_t)d.Wsb % 16 == 0), "A / W planes must be 16-byte aligned");
    static const int forced = [] {
        const char *e = getenv("EDV_SB_TILE");  // experiments: 0 = 128x64, 1 = 256x32, 2 = 128x128
        return e ? atoi(e) : -1;
    }();
    int t = d.N <= 32 ? 1 : 0;
    if (forced >= 0 && forced <= 2) t = forced;
    switch (t) {
        case 1: return launch_sb<256, 32, 4, 1>(d, st);
        case 2: return launch_sb<128, 128, 2, 2>(d, st);
        default: return launch_sb<128, 64, 2, 2>(d, st);
    }
}

int split_planes(const float *w, unsigned short *out, long long n, hipStream_t st) {
    EDV_CHECK(w && out && n > 0, "bad operand");
    const long long b = (n + 255) / 256;
    hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)(b < 8192 ? b : 8192)), dim3(256), 0, st, w, out, n, n);
    EDV_LAUNCH_OK();
    return 0;
}

}  // namespace edv
