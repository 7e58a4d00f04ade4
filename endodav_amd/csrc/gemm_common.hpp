// Pieces shared by the fp32-MFMA GEMMs (gemm.hip, gemm_dma.hip, conv_dma.hip; experiments/gemm_sb.hip used them too): all accumulate 32x32
// tiles whose C/D register layout is  col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)  (dtype-independent on gfx950).
#pragma once
#include "ops.hpp"

namespace edv {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == ACT_GELU) return gelu_erf(v);
    if (act == ACT_RELU) return fmaxf(v, 0.0f);
    return v;
}

// Stream-K work split of one launch of the LDS-DMA kernels (gemm_dma.hip explains the scheme; conv_dma.hip uses it for small grids).
struct GemmSplit {
    int whole_rounds, chunk, nsplit, stride;
    long long units;
    float *ws;   // piece slots (after the counters)
    int *cnt;    // one arrival counter per split tile
    int group_m = 1;  // gemm_x6.hip: row blocks per tile group of its tile order
};
// The merging workgroup's acquire (round 3).  The piece exchange stores and loads every piece word sc1 behind a drained, barrier-ordered counter add:
// the form MI355X_MICROARCH.md ("Valid forms") measures as sufficient WITHOUT an acquire -- but only at one workgroup per CU, and the split launches
// run three.  Outside that table the guide says "keep the acquire", so the ONE lane whose counter add completes a tile invalidates its CU's L1
// (buffer_inv sc1) and waits for it before the workgroup barrier that precedes the piece loads: poll -> acquire -> vmcnt(0) -> barrier -> loads, the
// guide's consumer sequence.  Only the last arriver of a split tile pays it (priced at 1.7-7 us by the guide; measured on the step in
// profiles/r03_notes.txt).  -DEDV_SPLIT_ACQUIRE=0 builds the round-2 form for A/B runs.
#ifndef EDV_SPLIT_ACQUIRE
#define EDV_SPLIT_ACQUIRE 1
#endif
__device__ __forceinline__ void split_merge_acquire() {
#if EDV_SPLIT_ACQUIRE
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}
constexpr int SPLIT_SLOT = 64 * 64;        // floats per piece: a 64x64 or 128x32 tile in accumulator order, (wave * 16 + r) * 64 + lane
constexpr int SPLIT_MAX_COUNTERS = 4096;   // >= the tiles a launch may split


// XCD-aware, bijective block remap: blocks b and b+8 share an XCD (and its L2); give each XCD a contiguous run of
// logical tiles so that tiles sharing an operand panel hit the same L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, x = bid & 7, loc = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + loc;
}

// v = acc + bias[n] + P1[p1_map(m), n];  v = act(v) * gamma[n];  v += R1[r1_map(m), n] + R2[c_row, n];  store
template <int FM, int FN, int STORE>
__device__ __forceinline__ void gemm_epilogue(const GemmDesc &g, f32x16 (&acc)[FM][FN], long long m0, int n0, int wrow, int wcol, int l31, int lh) {
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        const int n = n0 + wcol + j * 32 + l31;
        if (n >= g.N) continue;
        const float bias = g.bias ? g.bias[n] : 0.f;
        const float gam = g.gamma ? g.gamma[n] : 1.f;
        int ps_sub = 0, ps_co = 0, ps_dy = 0, ps_dx = 0;
        if (STORE == STORE_SHUFFLE) {
            ps_sub = n / g.ps_C;
            ps_co = n - ps_sub * g.ps_C;
            ps_dy = ps_sub / g.ps_s;
            ps_dx = ps_sub - ps_dy * g.ps_s;
        }
#pragma unroll
        for (int i = 0; i < FM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long m = m0 + wrow + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m >= g.M) continue;
                float pre = acc[i][j][r] + bias;
                if (g.P1) pre += g.P1[g.p1_map(m) * g.ldp1 + n];
                float v = apply_act(pre, g.act) * gam;
                if (STORE == STORE_ROWS) {
                    const long long crow = g.c_map(m);
                    if (g.R1) v += g.R1[g.r1_map(m) * g.ldr1 + n];
                    if (g.R2) v += g.R2[crow * g.ldr2 + n];
                    g.C[crow * g.ldc + n] = v;
                } else {
                    const int gp = g.ps_h * g.ps_w;
                    const long long f = m / gp;
                    const int p = (int)(m - f * gp);
                    const int y = p / g.ps_w, x = p - y * g.ps_w;
                    const long long orow = (f * g.ps_h * g.ps_s + (long long)y * g.ps_s + ps_dy) * (g.ps_w * g.ps_s) + x * g.ps_s + ps_dx;
                    g.C[orow * g.ps_C + ps_co] = v;
                }
            }
        }
    }
}


// The same epilogue for ONE element (used by the stream-K fix-up kernel, which sums the pieces of a split tile).
template <int STORE>
__device__ __forceinline__ void gemm_epilogue_elem(const GemmDesc &g, float accv, long long m, int n) {
    float pre = accv + (g.bias ? g.bias[n] : 0.f);
    if (g.P1) pre += g.P1[g.p1_map(m) * g.ldp1 + n];
    float v = apply_act(pre, g.act) * (g.gamma ? g.gamma[n] : 1.f);
    if (STORE == STORE_ROWS) {
        const long long crow = g.c_map(m);
        if (g.R1) v += g.R1[g.r1_map(m) * g.ldr1 + n];
        if (g.R2) v += g.R2[crow * g.ldr2 + n];
        g.C[crow * g.ldc + n] = v;
    } else {
        const int ps_sub = n / g.ps_C, ps_co = n - ps_sub * g.ps_C;
        const int ps_dy = ps_sub / g.ps_s, ps_dx = ps_sub - ps_dy * g.ps_s;
        const int gp = g.ps_h * g.ps_w;
        const long long f = m / gp;
        const int p = (int)(m - f * gp);
        const int y = p / g.ps_w, x = p - y * g.ps_w;
        const long long orow = (f * g.ps_h * g.ps_s + (long long)y * g.ps_s + ps_dy) * (g.ps_w * g.ps_s) + x * g.ps_s + ps_dx;
        g.C[orow * g.ps_C + ps_co] = v;
    }
}

// ---- compact epilogue for the common case -----------------------------------------------------------------------------
// The general epilogue above unrolls 16 rows x every optional feature (row maps with 64-bit divisions, P1, a runtime
// activation switch with an inlined erff per row): ~11 000 instructions, more than the 64 KB instruction cache two CUs
// share.  A workgroup timeline (scratch/ubench/gemm_trace.hip) showed 7.8 us of a 27 us workgroup life inside it.  Every
// encoder linear and every convolution needs only: identity row maps, no P1, a compile-time activation.
//   EP = 0: general;  EP = 1 + ACT (ACT_NONE / ACT_GELU / ACT_RELU): fast.
//   EP = 6: GEGLU (gemm_dma.hip): the tile's columns 0..31 are values, 32..63 their gates; C gets value * gelu(gate), half as wide.
//   EP = 5: row-mapped output (and residual) with one period >= 32 rows, no activation -- the patch-embed GEMM, whose rows go to
//           frame f's token slots behind the class token and take the position table as a per-frame-periodic residual.
inline int epilogue_kind(const GemmDesc &d) {
    if (d.geglu) return 6;  // (gemm_dma.hip only; gemm_geglu_supported() is the caller's gate)
    if (d.store == STORE_ROWS && !d.P1 && !d.R2 && d.act == ACT_NONE && d.c_map.period >= 32 && d.c_map.inner == 1 &&
        (!d.R1 || (d.r1_map.period == d.c_map.period && d.r1_map.inner == 1)))
        return 5;
    if (d.store != STORE_ROWS || d.P1 || d.c_map.period != 0 || (d.R1 && d.r1_map.period != 0)) return 0;
    if (d.act != ACT_NONE && d.act != ACT_GELU && d.act != ACT_RELU) return 0;
    return 1 + d.act;
}

// Per-column epilogue operands, fetched BEFORE the k loop: at the end of the loop the load would queue behind the
// other workgroups' tile traffic and its latency would be fully exposed.
#ifndef EDV_EPI_STORE_COND
#define EDV_EPI_STORE_COND  // ablation hook of scratch/ubench/gemm_trace.hip; empty in the product build
#endif
template <int FN>
struct EpiCols {
    float bias[FN], gam[FN];
};
template <int FN>
__device__ __forceinline__ EpiCols<FN> gemm_epilogue_prefetch(const GemmDesc &g, int n0, int wcol, int l31) {
    EpiCols<FN> c;
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        int n = n0 + wcol + j * 32 + l31;
        n = n < g.N ? n : g.N - 1;
        c.bias[j] = g.bias ? g.bias[n] : 0.f;
        c.gam[j] = g.gamma ? g.gamma[n] : 1.f;
    }
    return c;
}

template <int FM, int FN, int ACT>
__device__ __forceinline__ void gemm_epilogue_fast(const GemmDesc &g, f32x16 (&acc)[FM][FN], const EpiCols<FN> &cols, long long m0, int n0, int wrow,
                                                   int wcol, int l31, int lh) {
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        const int n = n0 + wcol + j * 32 + l31;
        if (n >= g.N) continue;
        const float bias = cols.bias[j], gam = cols.gam[j];
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const long long mb = m0 + wrow + i * 32 + 4 * lh;  // row of register r: mb + (r&3) + 8*(r>>2)
            const int left = (int)(g.M - mb < 32 ? g.M - mb : 32);
            float res[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) res[r] = 0.f;
            if (g.R1) {
                const float *rp = g.R1 + mb * g.ldr1 + n;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dr = (r & 3) + 8 * (r >> 2);
                    if (dr < left) res[r] = rp[(long long)dr * g.ldr1];
                }
            }
            if (g.R2) {
                const float *rp = g.R2 + mb * g.ldr2 + n;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dr = (r & 3) + 8 * (r >> 2);
                    if (dr < left) res[r] += rp[(long long)dr * g.ldr2];
                }
            }
            float *cp = g.C + mb * g.ldc + n;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dr = (r & 3) + 8 * (r >> 2);
                float v = acc[i][j][r] + bias;
                if (ACT == ACT_GELU) v = gelu_erf(v);
                if (ACT == ACT_RELU) v = fmaxf(v, 0.0f);
                if (dr < left EDV_EPI_STORE_COND) cp[(long long)dr * g.ldc] = v * gam + res[r];
            }
        }
    }
}

// Epilogue of one wave's 32x32 block through buffer descriptors (LDS-DMA kernels, gemm_dma.hip / conv_dma.hip) (EP = 1..3: identity row maps, compile-time activation).  The general form
// (gemm_epilogue_fast) spends ~20 VALU instructions per output register on 64-bit addresses, row-bound compares and exec masks -- a few
// hundred per tile, and on this part every VALU instruction is matrix-pipe time (a K = 384 GEMM tile is only 192 MFMAs per wave).  Here the
// per-lane offset (4 lh rows + column l31) is one multiply per tile and the row advance lives in the SCALAR offset of buffer_load /
// buffer_store: the address arithmetic is SALU, the VALU work is the arithmetic the epilogue exists for (bias, activation, gamma, residual).
// Edge tiles need no second code path: the descriptors end at row M (num_records = M x ld x 4 bytes; the hardware range-checks voffset + soffset
// as one sum without wrap-around -- scratch/ubench/buf_range.hip -- so stores to rows past M are dropped and loads return 0), and a lane whose
// column is past N gets a per-lane offset near 2^32, which no scalar offset brings back into range.
// cache policy of the output stores (aux bits of buffer_store: 0 = default, 2 = nt / streaming): A/B builds only
#ifndef EDV_EPI_STORE_AUX
#define EDV_EPI_STORE_AUX 0
#endif
template <int ACT>
__device__ __forceinline__ void gemm_epilogue_buf(const GemmDesc &g, f32x16 &acc, const EpiCols<1> &cols, long long row0, int col0, int l31, int lh) {
    // row0 / col0: first row / column of the wave's block -- wave-uniform (callers derive them from readfirstlane(wave)), so they live in SGPRs
    using f32x2 = __attribute__((ext_vector_type(2))) float;
    f32x2 res[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) res[i] = f32x2{0.f, 0.f};
    if (g.R1) {
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(g.R1), 0, (int)(unsigned)(g.M * g.ldr1 * 4), 0x00020000);
        const unsigned vo = (unsigned)((4 * lh * g.ldr1 + l31) * 4);
        const unsigned so = (unsigned)((row0 * g.ldr1 + col0) * 4);
#pragma unroll
        for (int r = 0; r < 16; ++r)
            res[r >> 1][r & 1] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)vo, (int)(so + (unsigned)(((r & 3) + 8 * (r >> 2)) * g.ldr1 * 4)), 0));
    }
    if (g.R2) {
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(g.R2), 0, (int)(unsigned)(g.M * g.ldr2 * 4), 0x00020000);
        const unsigned vo = (unsigned)((4 * lh * g.ldr2 + l31) * 4);
        const unsigned so = (unsigned)((row0 * g.ldr2 + col0) * 4);
#pragma unroll
        for (int r = 0; r < 16; ++r)
            res[r >> 1][r & 1] += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)vo, (int)(so + (unsigned)(((r & 3) + 8 * (r >> 2)) * g.ldr2 * 4)), 0));
    }
    const unsigned lane_out = col0 + l31 >= g.N ? 0xfffff000u : 0u;  // past the last column: out of every descriptor's range (fits_buffer: < 2^32 - 2^20)
    const auto rc = __builtin_amdgcn_make_buffer_rsrc(g.C, 0, (int)(unsigned)(g.M * g.ldc * 4), 0x00020000);
    const unsigned vo = (unsigned)((4 * lh * g.ldc + l31) * 4) | lane_out;
    const unsigned so = (unsigned)((row0 * g.ldc + col0) * 4);
    const f32x2 bias = {cols.bias[0], cols.bias[0]}, gam = {cols.gam[0], cols.gam[0]};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        f32x2 v = f32x2{acc[2 * i], acc[2 * i + 1]} + bias;
        if (ACT == ACT_GELU) v = f32x2{gelu_erf(v[0]), gelu_erf(v[1])};
        if (ACT == ACT_RELU) v = f32x2{fmaxf(v[0], 0.0f), fmaxf(v[1], 0.0f)};
        v = v * gam + res[i];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int r = 2 * i + e;
            const float ve = e ? v[1] : v[0];  // (a bit_cast applied directly to the vector element v[e] compiles to element 0 for both e)
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(ve), rc, (int)vo, (int)(so + (unsigned)(((r & 3) + 8 * (r >> 2)) * g.ldc * 4)), EDV_EPI_STORE_AUX);
        }
    }
}

// EP = 5 (see epilogue_kind): rows m = mb + dr of a 32-row block cross at most one period boundary, so one division per block suffices
template <int FM, int FN>
__device__ __forceinline__ void gemm_epilogue_mapped(const GemmDesc &g, f32x16 (&acc)[FM][FN], const EpiCols<FN> &cols, long long m0, int n0, int wrow,
                                                     int wcol, int l31, int lh) {
    const int P = g.c_map.period;
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        const int n = n0 + wcol + j * 32 + l31;
        if (n >= g.N) continue;
        const float bias = cols.bias[j], gam = cols.gam[j];
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const long long mb = m0 + wrow + i * 32 + 4 * lh;
            const int left = (int)(g.M - mb < 32 ? g.M - mb : 32);
            const long long f0 = mb / P;
            const int in0 = (int)(mb - f0 * P);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dr = (r & 3) + 8 * (r >> 2);
                if (dr >= left) continue;
                const int t = in0 + dr;
                const int wrap = t >= P ? 1 : 0;
                const long long f = f0 + wrap;
                const int pin = t - wrap * P;
                float v = (acc[i][j][r] + bias) * gam;
                if (g.R1) v += g.R1[(f * g.r1_map.stride + g.r1_map.offset + pin) * g.ldr1 + n];
                g.C[(f * g.c_map.stride + g.c_map.offset + pin) * g.ldc + n] = v;
            }
        }
    }
}

template <int FM, int FN, int STORE, int EP>
__device__ __forceinline__ void gemm_epilogue_ep(const GemmDesc &g, f32x16 (&acc)[FM][FN], const EpiCols<FN> &cols, long long m0, int n0, int wrow,
                                                 int wcol, int l31, int lh) {
    if constexpr (EP == 0)
        gemm_epilogue<FM, FN, STORE>(g, acc, m0, n0, wrow, wcol, l31, lh);
    else if constexpr (EP == 5)
        gemm_epilogue_mapped<FM, FN>(g, acc, cols, m0, n0, wrow, wcol, l31, lh);
    else
        gemm_epilogue_fast<FM, FN, EP - 1>(g, acc, cols, m0, n0, wrow, wcol, l31, lh);
}

}  // namespace edv
