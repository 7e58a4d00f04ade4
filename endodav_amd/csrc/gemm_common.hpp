// Pieces shared by the fp32-MFMA GEMM (gemm.hip) and the split-bf16 GEMM (gemm_sb.hip): both accumulate 32x32
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

}  // namespace edv
