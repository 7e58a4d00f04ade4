// Encoder self-attention core, fp32, flash-style (no N x N score matrix in HBM).
// Restates models/backbones/layers/attention.py:60-66:  softmax((q * d^-0.5) kᵀ) v, d = 64.
//
// Layout: qkv [F*N, 3*heads*64] exactly as the qkv linear writes it (columns [3][heads][64]);
// out [F*N, heads*64] (= attn.transpose(1,2).reshape(B,N,C) of the reference).
//
// Workgroup = 4 waves = 128 queries of one (frame, head); each wave owns 32 queries and sweeps
// the keys in tiles of 64 staged through LDS (shared by the 4 waves).  All products run on
// v_mfma_f32_32x32x2_f32 in the TRANSPOSED orientation, so the query index sits on the lane:
//     Sᵀ[key, q]  = K · Qᵀ      A = K tile (LDS, b128 reads with the permuted-k trick of gemm.hip)
//                               B = Qᵀ (registers for the whole kernel, pre-scaled by d^-0.5 * log2 e)
//     Oᵀ[d, q]   += Vᵀ · Pᵀ     A = Vᵀ (LDS, one b32 per MFMA),  B = Pᵀ = the Sᵀ accumulator registers
// With queries on lanes the softmax row reductions are 31 in-lane max/adds plus ONE cross-half
// shuffle, the running max / sum / rescale factor are per-lane scalars, and the probabilities
// feed the second product straight from the accumulator registers: the accumulator's row map
// key = (r&3) + 8*(r>>2) + 4*(lane>>5) assigns register r of lane-half h exactly the k-slot h
// of MFMA step r, so no data moves between the two products.
//
// Scheduling: persistent workgroups, data-parallel + stream-K hybrid.  A task = (frame, head, 128-query block) swept
// over all key tiles; at T=8 ViT-S there are 528 of them for 512 co-resident workgroup slots (2 per CU), which as a
// plain grid costs three task times where 2.06 are needed (81 TF/s measured vs 129 TF/s on a large grid).  So the
// grid is exactly the number of co-resident slots G; every workgroup first runs floor(tasks / G) whole tasks, then the
// tasks % G leftover tasks are cut along the KEY axis into G equal runs of key tiles.  A run that does not cover a whole
// task leaves (unnormalised Oᵀ, running max, running sum) in a workspace slot — at most two per workgroup — and
// attn_combine_kernel merges the pieces of each leftover task (the usual log-sum-exp merge, in base 2).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "ops.hpp"

// Timeline hook for scratch/ubench/attn_trace.hip (stamps per key tile); expands to nothing in the product build.
#ifndef EDV_ATTN_STAMP
#define EDV_ATTN_STAMP(slot)
#endif

namespace edv {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;

constexpr int HD = 64;       // head dim
constexpr int KS = HD + 4;   // padded K row stride: 68r mod 64 = 4r -> conflict-free b128
// NW waves per workgroup (NW*32 queries share the staged K/V tiles).  Fewer waves per workgroup means more,
// smaller workgroups: better balance over the 256 CUs when frames*heads*ceil(N/128) is only ~2 per CU.
// KT keys per LDS tile (32 or 64).  32-key tiles halve the prefetch and score registers (159 instead of 198 per lane:
// 3 instead of 2 waves per SIMD) but measured no faster, so 64 stays the default; EDV_ATTN_KT=32 selects the other.
template <int NW, int KT>
__global__ __launch_bounds__(NW * 64) void attn_spatial_kernel(const float *__restrict__ qkv, float *__restrict__ out, float *__restrict__ ws,
                                                               float *__restrict__ lse, int N, int heads, int whole_rounds, long long units, int chunk) {
    constexpr int QB = NW * 32;
    constexpr int NT = NW * 64;
    constexpr int SLOTF = QB * (HD + 2);  // workspace slot: O [QB][64], m [QB], l [QB]
    __shared__ __attribute__((aligned(16))) float smem[KT * KS + KT * HD];
    float *sK = smem;
    float *sV = smem + KT * KS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    // XCD-aware block -> logical id map.  Workgroups b and b+8 share an XCD and its L2; the query blocks of one
    // (frame, head) all sweep the same K/V, so hand each XCD a contiguous run of logical ids = of tasks
    // (bijective for any grid size).  Measured: L2->fabric reads per launch 307 MB -> 69 MB (profiles/).
    const int nq = (N + QB - 1) / QB;
    const int G = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = G >> 3, r = G & 7, x = bid & 7, loc = bid >> 3;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + loc;
    }
    const int D = heads * HD, D3 = 3 * D;
    const int ntiles = (N + KT - 1) / KT;
    const float qscale = 0.125f * 1.44269504088896340736f;  // d^-0.5 (exact) * log2(e)
    constexpr int RS = NT / 16, RI = KT / RS;  // staging: rows per pass, passes (16 float4 per 64-float row)
    const int sc = tid & 15, sr = tid >> 4;

    // work list of this workgroup: whole_rounds whole tasks, then key-tile units [u, u_end) of the leftover tasks.
    // (Fetching the next run's first K/V tile under the last tile of the current run was tried: no gain, and the extra
    // control flow in the tile loop cost 3-4 % on every shape.)
    const int task_l0 = whole_rounds * G;
    long long u = (long long)bid * chunk;
    const long long u_end = u + chunk < units ? u + chunk : units;
    int round = 0, seg = 0;
    for (;;) {
        int task, kt0, kt1;
        float *part = nullptr;
        if (round < whole_rounds) {
            task = round * G + bid;
            kt0 = 0;
            kt1 = ntiles;
            ++round;
        } else if (u < u_end) {
            const int t = (int)(u / ntiles);
            kt0 = (int)(u - (long long)t * ntiles);
            const long long left = u_end - u;
            kt1 = kt0 + left < ntiles ? kt0 + (int)left : ntiles;
            task = task_l0 + t;
            u += kt1 - kt0;
            if (!(kt0 == 0 && kt1 == ntiles)) part = ws + ((long long)bid * 2 + seg) * SLOTF;
            ++seg;  // slot 0 = the run holding this workgroup's first unit, slot 1 = the next task's head
        } else {
            break;
        }
        const int qt = task % nq, fh = task / nq;
        const int head = fh % heads, frame = fh / heads;
        const float *base = qkv + (long long)frame * N * D3 + head * HD;
        const int qi = qt * QB + wave * 32 + l31;
        const int qrow = qi < N ? qi : N - 1;

        // Q fragment in permuted-k order: element e of qf[qq] is d = 8*qq + 4*lh + e
        f32x4 qf[8];
#pragma unroll
        for (int qq = 0; qq < 8; ++qq) {
            f32x4 v = *reinterpret_cast<const f32x4 *>(base + (long long)qrow * D3 + 8 * qq + 4 * lh);
            qf[qq] = v * qscale;
        }

        f32x16 o0, o1;
#pragma unroll
        for (int r = 0; r < 16; ++r) o0[r] = o1[r] = 0.f;
        float m_run = -INFINITY, l_run = 0.f;

        f32x4 rk[RI], rv[RI];
        auto load_tile = [&](int k0) {
#pragma unroll
            for (int i = 0; i < RI; ++i) {
                int kr = k0 + sr + RS * i;
                kr = kr < N ? kr : N - 1;  // clamped rows are masked below
                const float *p = base + (long long)kr * D3 + sc * 4;
                rk[i] = *reinterpret_cast<const f32x4 *>(p + D);
                rv[i] = *reinterpret_cast<const f32x4 *>(p + 2 * D);
            }
        };

        load_tile(kt0 * KT);
        EDV_ATTN_STAMP(0);
        for (int t = kt0; t < kt1; ++t) {
            const int k0 = t * KT;
            EDV_ATTN_STAMP(1 + t - kt0);
            __syncthreads();  // also fences the previous run's last tile
#pragma unroll
            for (int i = 0; i < RI; ++i) {
                *reinterpret_cast<f32x4 *>(&sK[(sr + RS * i) * KS + sc * 4]) = rk[i];
                *reinterpret_cast<f32x4 *>(&sV[(sr + RS * i) * HD + sc * 4]) = rv[i];
            }
            __syncthreads();
            if (t + 1 < kt1) load_tile(k0 + KT);

            // ---- Sᵀ = K Qᵀ : KT/32 sub-tiles of 32 keys
            constexpr int NS = KT / 32;
            f32x16 sc[NS];
#pragma unroll
            for (int u = 0; u < NS; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r) sc[u][r] = 0.f;
#pragma unroll
            for (int qq = 0; qq < 8; ++qq) {
                f32x4 kf[NS];
#pragma unroll
                for (int u = 0; u < NS; ++u) kf[u] = *reinterpret_cast<const f32x4 *>(&sK[(32 * u + l31) * KS + 8 * qq + 4 * lh]);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int u = 0; u < NS; ++u) sc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[u][e], qf[qq][e], sc[u], 0, 0, 0);
            }
            if (k0 + KT > N) {  // last tile: keys past the sequence end
#pragma unroll
                for (int u = 0; u < NS; ++u)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (k0 + 32 * u + (r & 3) + 8 * (r >> 2) + 4 * lh >= N) sc[u][r] = -INFINITY;
            }

            // ---- online softmax in base 2; this lane holds half of its query's KT scores, lane^32 the rest
            float mx = sc[0][0];
#pragma unroll
            for (int u = 0; u < NS; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sc[u][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);  // first tile: exp2(-inf) = 0
            m_run = m_new;
            float psum = 0.f;
#pragma unroll
            for (int u = 0; u < NS; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    sc[u][r] = __builtin_amdgcn_exp2f(sc[u][r] - m_new);
                    psum += sc[u][r];
                }
            l_run = l_run * alpha + psum;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                o0[r] *= alpha;
                o1[r] *= alpha;
            }

            // ---- Oᵀ += Vᵀ Pᵀ : step r of sub-tile u contracts keys {key(r,0), key(r,1)}
#pragma unroll
            for (int u = 0; u < NS; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = 32 * u + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    const float v0 = sV[key * HD + l31], v1 = sV[key * HD + 32 + l31];
                    o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v0, sc[u][r], o0, 0, 0, 0);
                    o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v1, sc[u][r], o1, 0, 0, 0);
                }
        }

        EDV_ATTN_STAMP(1 + kt1 - kt0);
        const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
        float inv = 1.0f / l_tot;
        float *orow = nullptr;
        if (part) {  // a piece of a split task: unnormalised, merged by attn_combine_kernel
            const int ql = wave * 32 + l31;
            orow = part + ql * HD;
            inv = 1.0f;
            if (lh == 0) {
                part[QB * HD + ql] = m_run;
                part[QB * HD + QB + ql] = l_tot;
            }
        } else if (qi < N) {
            orow = out + ((long long)frame * N + qi) * D + head * HD;
            // training: per-row log-sum-exp in base 2, read by attn_spatial_bwd.hip
            if (lse && lh == 0) lse[((long long)frame * heads + head) * N + qi] = m_run + log2f(l_tot);
        }
        if (orow) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {  // registers 4g..4g+3 are d = 8g + 4*lh + {0..3}
                f32x4 a = {o0[4 * g] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv};
                f32x4 b = {o1[4 * g] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv};
                *reinterpret_cast<f32x4 *>(orow + 8 * g + 4 * lh) = a;
                *reinterpret_cast<f32x4 *>(orow + 32 + 8 * g + 4 * lh) = b;
            }
        }
        EDV_ATTN_STAMP(2 + kt1 - kt0);
    }
}

// ---- the VALU-lean form (round 2) ------------------------------------------------------------------------------------------------
// Same task list, piece format and result (to fp32 rounding) as attn_spatial_kernel<4, 64>.  What the round-2 measurements said
// (profiles/r02_notes.txt, scratch/ubench/mfma_feed.hip): the chip holds 2.34 GHz under this kernel, LDS has no conflicts, yet the matrix
// pipe is busy 79 % of the cycles -- because on this part the f32 MFMA and the VALU do not overlap (the f32 MFMA runs at exactly the f32
// vector rate): every VALU instruction between MFMAs costs its 4-8 issue cycles of matrix time, for both waves of the SIMD, whether it
// is interleaved with the MFMAs or not (a version that hid the softmax of tile t behind the PV MFMAs of tile t-1 measured 3 % SLOWER).
// The round-1 loop issued ~365 VALU instructions per 128 MFMAs: 127 accumulator-file moves (the compiler parked O in AGPRs), 62 compare /
// select pairs of the sequence-end mask on EVERY tile (if-converted), 33 subtractions of the running max, 32 address adds, ~32 rescale
// multiplies.  This kernel issues ~115:
//   * all registers in the unified VGPR file (__launch_bounds__(256, 2)): no v_accvgpr moves;
//   * the score accumulators start at -m (the running reference, one value per lane = per query) instead of 0, so the MFMA delivers s - m
//     and exp2 applies to the accumulator as it stands;
//   * the reference moves only when a tile's largest score exceeds it by more than 2^10 (or on a run's first tile): then, and only
//     then, the 32 scores, O and l are rebased.  exp2(s - m) stays <= 2^10: same result to rounding, the usual flash-attention freedom;
//   * the sequence-end mask is a real branch (last tile only);
//   * K/V tiles arrive by LDS-DMA through buffer descriptors: per-lane byte offsets are kernel constants, the tile / head / K-or-V
//     advance sits in the SCALAR offset, rows past the sequence end fall outside the descriptor and read as zeros; two LDS stages per
//     operand and a loop unrolled by two make every fragment address a register + immediate: no address arithmetic in the loop, one
//     barrier per tile.  K rows are XOR-swizzled 256-byte lines (b128 reads down a column), V rows are linear (b32 reads along a row).
constexpr int LEAN_NW = 4;
__global__ __launch_bounds__(LEAN_NW * 64, 2) void attn_lean_kernel(const float *__restrict__ qkv, float *__restrict__ out, float *__restrict__ ws,
                                                               float *__restrict__ lse, int N, int heads, int whole_rounds, long long units, int chunk) {
    constexpr int NW = LEAN_NW, KT = 64, QB = NW * 32;
    constexpr int SLOTF = QB * (HD + 2);
    constexpr int TILE = KT * HD;  // floats per staged tile (16 KB)
    constexpr int RPW = KT / NW;   // tile rows each wave stages (16), 4 rows per DMA instruction
    constexpr float TAU = 10.0f;   // rebase the softmax reference when a score exceeds it by more than 2^TAU
    __shared__ __attribute__((aligned(16))) float smem[4 * TILE];  // K stage 0, K stage 1, V stage 0, V stage 1

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    const int l31 = lane & 31, lh = lane >> 5;
    const int nq = (N + QB - 1) / QB;
    const int G = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = G >> 3, r = G & 7, x = bid & 7, loc = bid >> 3;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + loc;
    }
    const int D = heads * HD, D3 = 3 * D;
    const int ntiles = (N + KT - 1) / KT;
    const float qscale = 0.125f * 1.44269504088896340736f;
    // DMA: instruction i of this wave fills tile rows RPW*wave + 4i .. + 3; this lane's 16 bytes: row drow of those, chunk position dpos
    const int drow = lane >> 4, dpos = lane & 15;
    unsigned vk[RPW / 4], vv[RPW / 4];  // byte offsets inside a (frame, tile 0, head 0, column block 0) window
#pragma unroll
    for (int i = 0; i < RPW / 4; ++i) {
        const int r = RPW * wave + 4 * i + drow;
        vk[i] = (unsigned)(r * D3 + ((dpos ^ (r & 15)) << 2)) * 4u;
        vv[i] = (unsigned)(r * D3 + (dpos << 2)) * 4u;
    }
    // fragment addresses in stage 0: K chunk 2qq + lh of row l31 (and of row 32 + l31: + 8 KB); V row 4 lh, float l31
    const float *kaddr[8];
#pragma unroll
    for (int qq = 0; qq < 8; ++qq) kaddr[qq] = smem + l31 * HD + (((2 * qq + lh) ^ (l31 & 15)) << 2);
    const float *vaddr = smem + 2 * TILE + 4 * lh * HD + l31;
    // Key / value rows past the sequence end lie beyond the descriptor's num_records: for such a lane the DMA writes ZEROS into its LDS bytes
    // (measured: scratch/ubench/lds_dma_oob.hip, profiles/r03_notes.txt -- the same property conv_dma.hip's padding taps rely on).  Their scores are
    // masked and their probabilities are exactly 0, and 0 x 0 keeps the PV product clean; no stage needs clearing beforehand.  (Round 2 zero-filled
    // both stages here on the opposite assumption -- that out-of-range rows keep stale LDS contents, where 0 x NaN would poison PV.)

    const int task_l0 = whole_rounds * G;
    long long u = (long long)bid * chunk;
    const long long u_end = u + chunk < units ? u + chunk : units;
    int round = 0, seg = 0;
    for (;;) {
        int task, kt0, kt1;
        float *part = nullptr;
        if (round < whole_rounds) {
            task = round * G + bid;
            kt0 = 0;
            kt1 = ntiles;
            ++round;
        } else if (u < u_end) {
            const int t = (int)(u / ntiles);
            kt0 = (int)(u - (long long)t * ntiles);
            const long long left = u_end - u;
            kt1 = kt0 + left < ntiles ? kt0 + (int)left : ntiles;
            task = task_l0 + t;
            u += kt1 - kt0;
            if (!(kt0 == 0 && kt1 == ntiles)) part = ws + ((long long)bid * 2 + seg) * SLOTF;
            ++seg;
        } else {
            break;
        }
        const int qt = task % nq, fh = task / nq;
        const int head = fh % heads, frame = fh / heads;
        const float *fbase = qkv + (long long)frame * N * D3;  // this frame's q|k|v rows
        const float *base = fbase + head * HD;
        const int qi = qt * QB + wave * 32 + l31;
        const int qrow = qi < N ? qi : N - 1;
        // descriptor over the frame's N rows: a key row >= N lies beyond num_records and reads as zeros (masked below anyway)
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(fbase), 0, (unsigned)N * (unsigned)D3 * 4u, 0x00020000);

        f32x4 qf[8];
#pragma unroll
        for (int qq = 0; qq < 8; ++qq) {
            f32x4 v = *reinterpret_cast<const f32x4 *>(base + (long long)qrow * D3 + 8 * qq + 4 * lh);
            qf[qq] = v * qscale;
        }
        // which = 1: K (column block D), 2: V (column block 2 D) of key tile t into stage st
        auto issue = [&](int t, int st, int which) {
            float *dst = smem + (which == 1 ? 0 : 2 * TILE) + st * TILE + (RPW * wave_s) * HD;
            const int soff = ((t * KT) * D3 + which * D + head * HD) * 4;
#pragma unroll
            for (int i = 0; i < RPW / 4; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)(dst + 4 * i * HD), 16, (unsigned)(which == 1 ? vk[i] : vv[i]), (int)soff, 0, 0);
        };
        f32x16 o0, o1;
#pragma unroll
        for (int r = 0; r < 16; ++r) o0[r] = o1[r] = 0.f;
        float m_run = 0.f, l_run = 0.f;  // m_run: the reference the scores are taken against (set by the run's first tile)

        // one key tile out of stage ST; FIRST = the run's first tile (no reference yet)
        auto tile = [&](int t, auto st_tag, auto first_tag) {
            constexpr int ST = decltype(st_tag)::value;
            constexpr bool FIRST = decltype(first_tag)::value;
            const int k0 = t * KT;
            // This wave's share of K(t) and V(t) has landed (vmcnt) AND its LDS reads of tile t-1 have returned (lgkmcnt): only then may the other
            // waves refill that stage.  Without the lgkmcnt wait the compiler leaves the previous tile's last V read in flight across the barrier
            // (its two MFMAs sink below it); alone on the GPU the read always wins the race against the next DMA, but beside another stream's kernels
            // one wave in a few hundred launches multiplied the NEXT tile's V rows -- 32 query rows of one head off by 1e-2, found by running the
            // two-frame-group encoder (scratch/lanes_micro.py reproduces it in 3 s; tests/test_kernels_gpu.py::test_attention_beside_other_kernels).
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();  // ... everybody's has; and every wave is done with the other stage
            if (t + 1 < kt1) {
                issue(t + 1, ST ^ 1, 1);
                issue(t + 1, ST ^ 1, 2);
            }
            // ---- S^T - m = K Q^T - m: the accumulators start at minus the reference
            f32x16 s0, s1;
            const float init = FIRST ? 0.f : -m_run;
#pragma unroll
            for (int r = 0; r < 16; ++r) s0[r] = s1[r] = init;
#pragma unroll
            for (int qq = 0; qq < 8; ++qq) {
                const f32x4 ka = *reinterpret_cast<const f32x4 *>(kaddr[qq] + ST * TILE);
                const f32x4 kb = *reinterpret_cast<const f32x4 *>(kaddr[qq] + ST * TILE + 32 * HD);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    s0 = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[e], qf[qq][e], s0, 0, 0, 0);
                    s1 = __builtin_amdgcn_mfma_f32_32x32x2f32(kb[e], qf[qq][e], s1, 0, 0, 0);
                }
            }
            if (k0 + KT > N) {  // the sequence's last, partial tile only (the empty asm keeps this a real branch)
                asm volatile("" ::: "memory");
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (k0 + (r & 3) + 8 * (r >> 2) + 4 * lh >= N) s0[r] = -INFINITY;
                    if (k0 + 32 + (r & 3) + 8 * (r >> 2) + 4 * lh >= N) s1[r] = -INFINITY;
                }
            }
            // two chains of three-way maxima (v_max3_f32): 17 instructions for the 32 scores.  This file is compiled with -fno-honor-nans
            // (Makefile): otherwise every MFMA result is first "canonicalised" with a v_max_f32 x, x of its own, 57 instructions instead.
            float mx = fmaxf(s0[0], s1[0]), my = fmaxf(s0[8], s1[8]);
#pragma unroll
            for (int r = 1; r < 8; ++r) {
                mx = fmaxf(fmaxf(mx, s0[r]), s1[r]);
                my = fmaxf(fmaxf(my, s0[r + 8]), s1[r + 8]);
            }
            mx = fmaxf(mx, my);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            // rebase?  (wave-uniform decision; each lane then moves by its own amount)
            if (FIRST || __builtin_amdgcn_ballot_w64(mx > TAU) != 0) {
                const float d = FIRST ? mx : fmaxf(mx, 0.f);
                const float alpha = FIRST ? 0.f : __builtin_amdgcn_exp2f(-d);
                m_run += d;
                l_run *= alpha;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    s0[r] -= d;
                    s1[r] -= d;
                    o0[r] *= alpha;
                    o1[r] *= alpha;
                }
            }
            // row sums two at a time (v_pk_add_f32: 16 instructions for 32 addends)
            f32x2 psum = {0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                s0[r] = __builtin_amdgcn_exp2f(s0[r]);
                s0[r + 1] = __builtin_amdgcn_exp2f(s0[r + 1]);
                psum += f32x2{s0[r], s0[r + 1]};
            }
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                s1[r] = __builtin_amdgcn_exp2f(s1[r]);
                s1[r + 1] = __builtin_amdgcn_exp2f(s1[r + 1]);
                psum += f32x2{s1[r], s1[r + 1]};
            }
            l_run += psum[0] + psum[1];
            // ---- O^T += V^T P^T: register r of lane-half h of sub-tile u is key 32 u + (r&3) + 8 (r>>2) + 4 h
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float *va = vaddr + ST * TILE + ((r & 3) + 8 * (r >> 2)) * HD;
                o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(va[0], s0[r], o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(va[32], s0[r], o1, 0, 0, 0);
                o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(va[32 * HD], s1[r], o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(va[32 * HD + 32], s1[r], o1, 0, 0, 0);
            }
        };
        issue(kt0, 0, 1);
        issue(kt0, 0, 2);
        tile(kt0, std::integral_constant<int, 0>{}, std::true_type{});
        {
            int t = kt0 + 1;
            for (; t + 1 < kt1; t += 2) {
                tile(t, std::integral_constant<int, 1>{}, std::false_type{});
                tile(t + 1, std::integral_constant<int, 0>{}, std::false_type{});
            }
            if (t < kt1) tile(t, std::integral_constant<int, 1>{}, std::false_type{});
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();  // every wave has READ (not just issued the reads of) its last V tile before the next run's first DMA refills stage 0

        const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
        float inv = 1.0f / l_tot;
        float *orow = nullptr;
        if (part) {
            const int ql = wave * 32 + l31;
            orow = part + ql * HD;
            inv = 1.0f;
            if (lh == 0) {
                part[QB * HD + ql] = m_run;
                part[QB * HD + QB + ql] = l_tot;
            }
        } else if (qi < N) {
            orow = out + ((long long)frame * N + qi) * D + head * HD;
            if (lse && lh == 0) lse[((long long)frame * heads + head) * N + qi] = m_run + log2f(l_tot);
        }
        if (orow) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 a = {o0[4 * g] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv};
                f32x4 b = {o1[4 * g] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv};
                *reinterpret_cast<f32x4 *>(orow + 8 * g + 4 * lh) = a;
                *reinterpret_cast<f32x4 *>(orow + 32 + 8 * g + 4 * lh) = b;
            }
        }
    }
}

// ---- the bf16 x 6 form (round 3; EDV_PRODUCTS_BF16X6) --------------------------------------------------------------------------------
// Same task list, piece format and result as attn_lean_kernel -- to fp32 rounding: both products run as six v_mfma_f32_32x32x16_bf16 on three-term
// bf16 splits of BOTH operands with fp32 accumulation, the scheme of gemm_x6.hip (a = a0 + a1 + a2; a w ~= a2 w0 + a0 w2 + a1 w1 + a1 w0 + a0 w1 + a0 w0,
// the three dropped terms <= 2^-26 |a w|).  The softmax is the lean kernel's, on the fp32 accumulators.  What changes with the instruction:
//   * 16 k per MFMA and 8 consecutive k per lane: K^T's A operand reads 16 bytes of one key row (dims 8h .. 8h + 7 of a 16-dim step); the PV product
//     contracts over KEYS, so its A operand needs 8 keys of one dim d contiguous: V is staged TRANSPOSED ([d][key] planes), by the staging threads
//     that split it -- each takes a 4 key x 4 dim block, the transpose is register naming.  Which 8 keys: the S^T accumulator registers
//     8 (j & 1) .. + 7 of sub-tile j >> 1 hold, for lane-half h, keys 16 j + 4 h + {0..3} and 16 j + 8 + 4 h + {0..3}; they ARE the B operand of PV
//     step j (after the split), and V^T's plane rows store each 16-key group in the order [0-3, 8-11 | 4-7, 12-15] so that half h reads one b128;
//   * K / V arrive through registers (global loads of tile t + 1 during tile t, split and written to the other LDS stage at the tile's end): the
//     split needs the VALU anyway, and beside the bf16 MFMA VALU work hides (gemm_x6.hip);
//   * 8 waves = 256 queries per workgroup share a staged tile (2 stages x (3 K planes + 3 V^T planes) x 8 KB = 96 KB: one workgroup per CU, two
//     waves per SIMD); waves 0-3 stage K, waves 4-7 stage V^T.
// LDS plane image: 128-byte rows (64 bf16), 16-byte chunk c of row r at chunk position c ^ ((r >> 1) & 7): a 16-lane b128 group (16 rows, one c) covers
// all 64 banks.
typedef __bf16 xbf8 __attribute__((ext_vector_type(8)));
typedef __bf16 xbf4 __attribute__((ext_vector_type(4)));
constexpr int X6_NW = 8, X6_KT = 64, X6_QB = X6_NW * 32;
constexpr int X6_PLANE = X6_KT * HD * 2;     // bytes: 64 rows x 128 B
constexpr int X6_STAGE = 6 * X6_PLANE;       // K0 K1 K2 V0 V1 V2 (48 KB)

struct Split3 {
    __bf16 p0, p1, p2;
};
__device__ __forceinline__ Split3 split3(float x) {
    Split3 s;
    s.p0 = (__bf16)x;
    const float r1 = x - (float)s.p0;
    s.p1 = (__bf16)r1;
    s.p2 = (__bf16)(r1 - (float)s.p1);
    return s;
}

__global__ __launch_bounds__(X6_NW * 64) void attn_x6_kernel(const float *__restrict__ qkv, float *__restrict__ out, float *__restrict__ ws, float *__restrict__ lse,
                                                             int N, int heads, int whole_rounds, long long units, int chunk) {
    constexpr int KT = X6_KT, QB = X6_QB;
    constexpr int SLOTF = QB * (HD + 2);
    constexpr float TAU = 10.0f;
    extern __shared__ __attribute__((aligned(16))) unsigned char xsm[];  // 2 stages

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int nq = (N + QB - 1) / QB;
    const int G = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = G >> 3, r = G & 7, x = bid & 7, loc = bid >> 3;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + loc;
    }
    const int D = heads * HD, D3 = 3 * D;
    const int ntiles = (N + KT - 1) / KT;
    const float qscale = 0.125f * 1.44269504088896340736f;

    // staging role: threads 0..255 take K, 256..511 V; each a block of 4 keys (4 kg ..) x 4 dims (4 dg ..) of the tile
    const int srole = tid >> 8, st_id = tid & 255, kg = st_id >> 4, dg = st_id & 15;
    int wr[4];  // byte offsets of this thread's four 8-byte writes inside a plane
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (srole == 0) {  // K plane row = key 4 kg + i; dims 4 dg .. + 3 = chunk dg >> 1, half (dg & 1)
            const int key = 4 * kg + i;
            wr[i] = key * 128 + (((dg >> 1) ^ ((key >> 1) & 7)) << 4) + (dg & 1) * 8;
        } else {           // V^T plane row = dim 4 dg + i; keys 4 kg .. + 3 = group kg >> 2, chunk 2 (kg >> 2) + (kg & 1), half (kg >> 1) & 1
            const int d = 4 * dg + i, c = 2 * (kg >> 2) + (kg & 1);
            wr[i] = d * 128 + ((c ^ ((d >> 1) & 7)) << 4) + ((kg >> 1) & 1) * 8;
        }
    }
    // fragment addresses in stage 0, plane 0: K rows l31 (+ 32), V^T rows l31 (+ 32); chunk 2 s + lh of step s
    int fk[4], fv[4];
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) {
        fk[s2] = l31 * 128 + (((2 * s2 + lh) ^ ((l31 >> 1) & 7)) << 4);
        fv[s2] = 3 * X6_PLANE + fk[s2];
    }

    const int task_l0 = whole_rounds * G;
    long long u = (long long)bid * chunk;
    const long long u_end = u + chunk < units ? u + chunk : units;
    int round = 0, seg = 0;
    for (;;) {
        int task, kt0, kt1;
        float *part = nullptr;
        if (round < whole_rounds) {
            task = round * G + bid;
            kt0 = 0;
            kt1 = ntiles;
            ++round;
        } else if (u < u_end) {
            const int t = (int)(u / ntiles);
            kt0 = (int)(u - (long long)t * ntiles);
            const long long left = u_end - u;
            kt1 = kt0 + left < ntiles ? kt0 + (int)left : ntiles;
            task = task_l0 + t;
            u += kt1 - kt0;
            if (!(kt0 == 0 && kt1 == ntiles)) part = ws + ((long long)bid * 2 + seg) * SLOTF;
            ++seg;
        } else {
            break;
        }
        const int qt = task % nq, fh = task / nq;
        const int head = fh % heads, frame = fh / heads;
        const float *fbase = qkv + (long long)frame * N * D3;
        const float *base = fbase + head * HD;
        const int qi = qt * QB + wave * 32 + l31;
        const int qrow = qi < N ? qi : N - 1;
        // a wave whose 32 queries all lie past the sequence end (the last 256-query block of N = 1370 holds 90) stages and meets the barriers but
        // computes nothing: its SIMD's matrix pipe goes to the partner wave
        const bool wact = qt * QB + __builtin_amdgcn_readfirstlane(wave) * 32 < N;

        // Q^T planes (B operand of S^T): dims 16 s + 8 lh .. + 7 of this lane's query, pre-scaled, split
        xbf8 qb[3][4];
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
            const f32x4 a = *reinterpret_cast<const f32x4 *>(base + (long long)qrow * D3 + 16 * s2 + 8 * lh);
            const f32x4 b = *reinterpret_cast<const f32x4 *>(base + (long long)qrow * D3 + 16 * s2 + 8 * lh + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const Split3 sa = split3(a[e] * qscale), sb = split3(b[e] * qscale);
                qb[0][s2][e] = sa.p0; qb[1][s2][e] = sa.p1; qb[2][s2][e] = sa.p2;
                qb[0][s2][4 + e] = sb.p0; qb[1][s2][4 + e] = sb.p1; qb[2][s2][4 + e] = sb.p2;
            }
        }
        // staging: this thread's 4 x 4 block of K (column block D) or V (2 D) of key tile t; rows past the sequence end read the last row (their
        // scores are masked to -inf, their probabilities are exactly 0)
        const float *sbase = base + (srole == 0 ? D : 2 * D) + 4 * dg;
        auto load_tile = [&](int t, f32x4 (&v)[4]) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int key = t * KT + 4 * kg + i;
                key = key < N ? key : N - 1;
                v[i] = *reinterpret_cast<const f32x4 *>(sbase + (long long)key * D3);
            }
        };
        auto split_store = [&](const f32x4 (&v)[4], int st) {
            unsigned char *dst = xsm + st * X6_STAGE + (srole == 0 ? 0 : 3 * X6_PLANE);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                xbf4 p0, p1, p2;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    // K: row = key i, the four dims e;  V^T: row = dim i, the four keys e (the transpose)
                    const Split3 sx = split3(srole == 0 ? v[i][e] : v[e][i]);
                    p0[e] = sx.p0; p1[e] = sx.p1; p2[e] = sx.p2;
                }
                *reinterpret_cast<xbf4 *>(dst + wr[i]) = p0;
                *reinterpret_cast<xbf4 *>(dst + X6_PLANE + wr[i]) = p1;
                *reinterpret_cast<xbf4 *>(dst + 2 * X6_PLANE + wr[i]) = p2;
            }
        };

        f32x16 o0, o1;
#pragma unroll
        for (int r = 0; r < 16; ++r) o0[r] = o1[r] = 0.f;
        float m_run = 0.f, l_run = 0.f;
        f32x4 treg[2][4];  // staging registers: set i holds the tile that goes to LDS stage i (loaded two tiles ahead of its use)

        constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};  // (tile-operand plane, register-operand plane), smallest terms first
        // FULL: a tile in the steady state (t + 2 < kt1): the conversion of tile t + 1 (about 100 VALU instructions and twelve plane writes) is spread
        // under the 48 MFMAs of S^T by an explicit schedule (two VALU per MFMA, a write every fourth) instead of running as a burst at the tile's end,
        // where the two waves of a SIMD -- in lockstep behind the same barrier -- run it at the same time.  Worth 1-4 %.  (Removing the staging
        // altogether is worth 22 %, the softmax 6 %, the P split 8 %, all three 27 %: profiles/r03_notes.txt.  Under these MFMAs the part is
        // power-bound at 1.65-1.8 GHz, so what the kernel pays for is the energy of those instructions, wherever they sit.)
        auto tile = [&](int t, auto st_tag, auto first_tag, auto full_tag) {
            constexpr int ST = decltype(st_tag)::value;
            constexpr bool FIRST = decltype(first_tag)::value;
            constexpr bool FULL = decltype(full_tag)::value;
            const int k0 = t * KT;
            // this wave's plane writes of tile t and its fragment reads of tile t - 1 are done; then everybody's are
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            // EDV_AX6_NOSTAGE / _NOSOFTMAX / _NOPSPLIT: ablation builds for scratch/attn_x6_time.py (wrong results, timing only; profiles/r03_notes.txt)
#ifndef EDV_AX6_NOSTAGE
            if (FULL || t + 2 < kt1) load_tile(t + 2, treg[ST]);  // set ST held tile t (in LDS since the previous tile); tile t + 2 goes to stage ST too
#endif
            const unsigned char *stg = xsm + ST * X6_STAGE;
            if (!wact) {
#ifndef EDV_AX6_NOSTAGE
                if (FULL || t + 1 < kt1) split_store(treg[ST ^ 1], ST ^ 1);
#endif
            } else {
            // ---- S^T - m = K Q^T - m
            f32x16 s0, s1;
            const float init = FIRST ? 0.f : -m_run;
#pragma unroll
            for (int r = 0; r < 16; ++r) s0[r] = s1[r] = init;
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) {
                xbf8 ka[3], kb[3];
#pragma unroll
                for (int p2 = 0; p2 < 3; ++p2) {
                    ka[p2] = *reinterpret_cast<const xbf8 *>(stg + p2 * X6_PLANE + fk[s2]);
                    kb[p2] = *reinterpret_cast<const xbf8 *>(stg + p2 * X6_PLANE + fk[s2] + 32 * 128);
                }
#pragma unroll
                for (int tt = 0; tt < 6; ++tt) {
                    s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[PA[tt]], qb[PB[tt]][s2], s0, 0, 0, 0);
                    s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kb[PA[tt]], qb[PB[tt]][s2], s1, 0, 0, 0);
                }
            }
#ifndef EDV_AX6_NOSTAGE
            // tile t + 1 (loaded during tile t - 1): split and written into the other stage, last read during tile t - 1 (everybody passed this tile's
            // barrier since).  In source order after the MFMAs; FULL spells out where the instructions go.
            if (FULL || t + 1 < kt1) split_store(treg[ST ^ 1], ST ^ 1);
            if (FULL) {
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);  // this step's six fragment reads
#pragma unroll
                    for (int i = 0; i < 12; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA
                        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);  // two VALU of the conversion
                        if ((i & 3) == 3) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);  // a plane write
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#endif
            if (k0 + KT > N) {
                asm volatile("" ::: "memory");
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (k0 + (r & 3) + 8 * (r >> 2) + 4 * lh >= N) s0[r] = -INFINITY;
                    if (k0 + 32 + (r & 3) + 8 * (r >> 2) + 4 * lh >= N) s1[r] = -INFINITY;
                }
            }
#ifndef EDV_AX6_NOSOFTMAX
            float mx = fmaxf(s0[0], s1[0]), my = fmaxf(s0[8], s1[8]);
#pragma unroll
            for (int r = 1; r < 8; ++r) {
                mx = fmaxf(fmaxf(mx, s0[r]), s1[r]);
                my = fmaxf(fmaxf(my, s0[r + 8]), s1[r + 8]);
            }
            mx = fmaxf(mx, my);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            if (FIRST || __builtin_amdgcn_ballot_w64(mx > TAU) != 0) {
                const float d = FIRST ? mx : fmaxf(mx, 0.f);
                const float alpha = FIRST ? 0.f : __builtin_amdgcn_exp2f(-d);
                m_run += d;
                l_run *= alpha;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    s0[r] -= d;
                    s1[r] -= d;
                    o0[r] *= alpha;
                    o1[r] *= alpha;
                }
            }
            float psum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s0[r] = __builtin_amdgcn_exp2f(s0[r]);
                s1[r] = __builtin_amdgcn_exp2f(s1[r]);
                psum += s0[r] + s1[r];
            }
            l_run += psum;
#else
            l_run += s0[0] + s1[15];
#endif
            // ---- O^T += V^T P^T, 16 keys per step: registers 8 (j & 1) .. + 7 of sub-tile j >> 1, split into planes, are the B operand
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                xbf8 pb[3];
#ifndef EDV_AX6_NOPSPLIT
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const Split3 sp = split3((j < 2 ? s0 : s1)[8 * (j & 1) + e]);
                    pb[0][e] = sp.p0; pb[1][e] = sp.p1; pb[2][e] = sp.p2;
                }
#else
                {
                    const f32x4 raw = {(j < 2 ? s0 : s1)[8 * (j & 1)], (j < 2 ? s0 : s1)[8 * (j & 1) + 1], (j < 2 ? s0 : s1)[8 * (j & 1) + 2], (j < 2 ? s0 : s1)[8 * (j & 1) + 3]};
                    pb[0] = pb[1] = pb[2] = __builtin_bit_cast(xbf8, raw);
                }
#endif
                xbf8 va[3], vb[3];
#pragma unroll
                for (int p2 = 0; p2 < 3; ++p2) {
                    va[p2] = *reinterpret_cast<const xbf8 *>(stg + p2 * X6_PLANE + fv[j]);
                    vb[p2] = *reinterpret_cast<const xbf8 *>(stg + p2 * X6_PLANE + fv[j] + 32 * 128);
                }
#pragma unroll
                for (int tt = 0; tt < 6; ++tt) {
                    o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va[PA[tt]], pb[PB[tt]], o0, 0, 0, 0);
                    o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vb[PA[tt]], pb[PB[tt]], o1, 0, 0, 0);
                }
            }
            }  // wact
        };
        load_tile(kt0, treg[0]);
        if (kt0 + 1 < kt1) load_tile(kt0 + 1, treg[1]);
        split_store(treg[0], 0);
        tile(kt0, std::integral_constant<int, 0>{}, std::true_type{}, std::false_type{});
        {
            int t = kt0 + 1;
            for (; t + 3 < kt1; t += 2) {  // (t + 1) + 2 < kt1: both tiles are full
                tile(t, std::integral_constant<int, 1>{}, std::false_type{}, std::true_type{});
                tile(t + 1, std::integral_constant<int, 0>{}, std::false_type{}, std::true_type{});
            }
            for (; t + 1 < kt1; t += 2) {
                tile(t, std::integral_constant<int, 1>{}, std::false_type{}, std::false_type{});
                tile(t + 1, std::integral_constant<int, 0>{}, std::false_type{}, std::false_type{});
            }
            if (t < kt1) tile(t, std::integral_constant<int, 1>{}, std::false_type{}, std::false_type{});
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();  // every wave has read its last tile before the next run's prologue rewrites stage 0

        const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
        float inv = 1.0f / l_tot;
        float *orow = nullptr;
        if (!wact) {
            // (rows past the sequence end: the combine kernel never reads them)
        } else if (part) {
            const int ql = wave * 32 + l31;
            orow = part + ql * HD;
            inv = 1.0f;
            if (lh == 0) {
                part[QB * HD + ql] = m_run;
                part[QB * HD + QB + ql] = l_tot;
            }
        } else if (qi < N) {
            orow = out + ((long long)frame * N + qi) * D + head * HD;
            if (lse && lh == 0) lse[((long long)frame * heads + head) * N + qi] = m_run + log2f(l_tot);
        }
        if (orow) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 a = {o0[4 * g] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv};
                f32x4 b = {o1[4 * g] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv};
                *reinterpret_cast<f32x4 *>(orow + 8 * g + 4 * lh) = a;
                *reinterpret_cast<f32x4 *>(orow + 32 + 8 * g + 4 * lh) = b;
            }
        }
    }
}

// Merge of the pieces of every split (leftover) task.  Piece list of leftover task t: the workgroups whose unit runs
// [g*chunk, (g+1)*chunk) intersect [t*ntiles, (t+1)*ntiles); a workgroup's piece sits in its slot 0 when its first unit
// lies in this task, else in slot 1.  Thread = (query, 16-byte output chunk); 16 queries per 256-thread block.
__global__ __launch_bounds__(256) void attn_combine_kernel(const float *__restrict__ ws, float *__restrict__ out, float *__restrict__ lse, int N, int heads,
                                                           int QB, int ntiles, int task_l0, long long units, int chunk) {
    const int t = blockIdx.x;
    const int ql = blockIdx.y * 16 + (threadIdx.x >> 4), ch = threadIdx.x & 15;
    const long long ub = (long long)t * ntiles, ue = ub + ntiles;
    const int g0 = (int)(ub / chunk), g1 = (int)((ue - 1) / chunk);
    if (g0 == g1 && (long long)g0 * chunk <= ub && (long long)(g0 + 1) * chunk >= ue) return;  // ran whole, already stored
    const int nq = (N + QB - 1) / QB;
    const int task = task_l0 + t;
    const int qt = task % nq, fh = task / nq;
    const int head = fh % heads, frame = fh / heads;
    const int qi = qt * QB + ql;
    if (qi >= N) return;
    const long long slotf = (long long)QB * (HD + 2);
    // two sweeps, loads batched four pieces at a time (the piece count is small but the loads are long-latency)
    auto slot_of = [&](int g) { return ws + ((long long)g * 2 + ((long long)g * chunk >= ub ? 0 : 1)) * slotf; };
    float M = -INFINITY;
    for (int g = g0; g <= g1; g += 4) {
        float mi[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) mi[j] = slot_of(g + j <= g1 ? g + j : g1)[QB * HD + ql];
#pragma unroll
        for (int j = 0; j < 4; ++j) M = fmaxf(M, mi[j]);
    }
    float lsum = 0.f;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int g = g0; g <= g1; g += 4) {
        float mi[4], li[4];
        f32x4 o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float *part = slot_of(g + j <= g1 ? g + j : g1);
            mi[j] = part[QB * HD + ql];
            li[j] = part[QB * HD + QB + ql];
            o[j] = *reinterpret_cast<const f32x4 *>(part + ql * HD + ch * 4);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float w = g + j <= g1 ? __builtin_amdgcn_exp2f(mi[j] - M) : 0.f;
            acc += o[j] * w;
            lsum += li[j] * w;
        }
    }
    const float inv = 1.0f / lsum;
    *reinterpret_cast<f32x4 *>(out + ((long long)frame * N + qi) * (heads * HD) + head * HD + ch * 4) = acc * inv;
    if (lse && ch == 0) lse[((long long)frame * heads + head) * N + qi] = M + log2f(lsum);
}

struct AttnPlan {
    int nw, kt, grid, whole_rounds, chunk, ntasks, ntiles, leftover;
    bool pipe, x6;
    long long units;
    size_t ws_floats;
};

int pipe_slots() {
    static DeviceSlotCache cache;
    return cache.get([] {
        int dev = 0, cus = 0, per_cu = 0;
        if (hipGetDevice(&dev) != hipSuccess) return 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, attn_lean_kernel, 256, 0) != hipSuccess) return 0;
        if (getenv("EDV_DEBUG_SLOTS")) fprintf(stderr, "attn_lean_kernel: %d CUs x %d resident workgroups\n", cus, per_cu);
        return cus * per_cu;
    });
}
bool use_pipe() {
    static const bool on = [] {
        const char *e = getenv("EDV_ATTN_LEAN");  // 0: the round-1 kernel (register-staged K/V, ~365 VALU instructions per key tile), for A/B runs
        return !(e && atoi(e) == 0);
    }();
    return on;
}

template <int NW, int KT>
int resident_slots() {
    static DeviceSlotCache cache;
    return cache.get([] {
        int dev = 0, cus = 0, per_cu = 0;
        if (hipGetDevice(&dev) != hipSuccess) return 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, attn_spatial_kernel<NW, KT>, NW * 64, 0) != hipSuccess) return 0;
        return cus * per_cu;
    });
}

int x6_attn_slots() {
    static DeviceSlotCache cache;
    return cache.get([] {
        int dev = 0, cus = 0, per_cu = 0;
        if (hipGetDevice(&dev) != hipSuccess) return 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
        if (hipFuncSetAttribute((const void *)attn_x6_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * X6_STAGE) != hipSuccess) return 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, attn_x6_kernel, X6_NW * 64, 2 * X6_STAGE) != hipSuccess) return 0;
        if (per_cu > 1) per_cu = 1;  // 96 KB of LDS
        if (getenv("EDV_DEBUG_SLOTS")) fprintf(stderr, "attn_x6_kernel: %d CUs x %d resident workgroups\n", cus, per_cu);
        return cus * per_cu;
    });
}

int make_plan(int F, int N, int heads, AttnPlan *p, bool x6 = false) {
    // 4 waves (128 queries) per workgroup share each staged K/V tile.  The 2- and 1-wave variants cost registers
    // (195 / 256 VGPRs) and measured slower on every shape tried (T=8: 79.9 vs 78.6 vs 68.5 TF/s); they are kept for
    // sequences shorter than one 128-query block and for experiments.
    static const int forced = [] {
        const char *e = getenv("EDV_ATTN_WAVES");
        return e ? atoi(e) : 0;
    }();
    static const int kt_forced = [] {
        const char *e = getenv("EDV_ATTN_KT");
        return e ? atoi(e) : 0;
    }();
    static const int plain = [] {
        const char *e = getenv("EDV_ATTN_PLAIN");  // 1: one workgroup per task (the pre-stream-K grid), for A/B runs
        return e ? atoi(e) : 0;
    }();
    int nw = N > 64 ? 4 : (N > 32 ? 2 : 1);
    if (forced == 1 || forced == 2 || forced == 4) nw = forced;
    p->nw = nw;
    p->kt = nw == 4 ? (kt_forced == 32 ? 32 : 64) : 32;  // 32-key tiles measured equal at N = 1370 (283.7 vs 284.0 us)
    p->x6 = x6 && N > 128;  // (shorter sequences do not fill one 256-query block: the fp32 kernels keep them)
    if (p->x6) {
        nw = p->nw = X6_NW;
        p->kt = X6_KT;
    }
    p->pipe = !p->x6 && nw == 4 && p->kt == 64 && use_pipe();
    const int slots = p->x6 ? x6_attn_slots() : p->pipe ? pipe_slots()
                              : nw == 4 ? (p->kt == 64 ? resident_slots<4, 64>() : resident_slots<4, 32>()) : nw == 2 ? resident_slots<2, 32>() : resident_slots<1, 32>();
    EDV_CHECK(slots > 0, "occupancy query failed");
    const long long ntasks = (long long)F * heads * ((N + nw * 32 - 1) / (nw * 32));
    EDV_CHECK(ntasks < (1ll << 31), "grid limits");
    // the lean kernel addresses one frame's q|k|v rows through a buffer descriptor with 32-bit byte offsets (N x 3 x heads x 64 floats per frame)
    EDV_CHECK(!p->pipe || (long long)(N + 64) * heads * 3 * HD * 4 < (1ll << 31), "one frame's q|k|v rows exceed the 2 GB a buffer descriptor offset reaches");
    p->ntasks = (int)ntasks;
    p->ntiles = (N + p->kt - 1) / p->kt;
    if (plain) {
        p->grid = p->ntasks; p->whole_rounds = 1; p->leftover = 0; p->units = 0; p->chunk = 1; p->ws_floats = 0;
        return 0;
    }
    p->whole_rounds = p->ntasks / slots;
    p->leftover = p->ntasks - p->whole_rounds * slots;
    p->units = (long long)p->leftover * p->ntiles;
    p->chunk = p->units ? (int)((p->units + slots - 1) / slots) : 1;
    p->grid = p->whole_rounds ? slots : (int)((p->units + p->chunk - 1) / p->chunk);
    const int split_wgs = (int)((p->units + p->chunk - 1) / p->chunk);
    p->ws_floats = (size_t)split_wgs * 2 * (size_t)(nw * 32) * (HD + 2);
    return 0;
}

}  // namespace

size_t attn_spatial_workspace(int F, int N, int heads, bool x6) {
    AttnPlan p;
    if (F <= 0 || N <= 0 || heads <= 0 || make_plan(F, N, heads, &p, x6)) return 0;
    return p.ws_floats;
}

int attn_spatial(const float *qkv, float *out, int F, int N, int heads, float *ws, size_t ws_floats, hipStream_t st, float *lse, bool x6) {
    EDV_CHECK(qkv && out, "null operand");
    EDV_CHECK(F > 0 && N > 0 && heads > 0, "empty problem");
    EDV_CHECK(((uintptr_t)qkv % 16 == 0) && ((uintptr_t)out % 16 == 0), "16-byte alignment");
    AttnPlan p;
    EDV_TRY(make_plan(F, N, heads, &p, x6));
    EDV_CHECK(p.ws_floats == 0 || (ws && ws_floats >= p.ws_floats && (uintptr_t)ws % 16 == 0), "attention workspace too small (attn_spatial_workspace)");
    dim3 grid((unsigned)p.grid);
    if (p.x6)
        EDV_LAUNCH((attn_x6_kernel), grid, dim3(X6_NW * 64), 2 * X6_STAGE, st, qkv, out, ws, lse, N, heads, p.whole_rounds, p.units, p.chunk);
    else if (p.pipe)
        EDV_LAUNCH((attn_lean_kernel), grid, dim3(256), 0, st, qkv, out, ws, lse, N, heads, p.whole_rounds, p.units, p.chunk);
    else if (p.nw == 4 && p.kt == 32)
        EDV_LAUNCH((attn_spatial_kernel<4, 32>), grid, dim3(256), 0, st, qkv, out, ws, lse, N, heads, p.whole_rounds, p.units, p.chunk);
    else if (p.nw == 4)
        EDV_LAUNCH((attn_spatial_kernel<4, 64>), grid, dim3(256), 0, st, qkv, out, ws, lse, N, heads, p.whole_rounds, p.units, p.chunk);
    else if (p.nw == 2)
        EDV_LAUNCH((attn_spatial_kernel<2, 32>), grid, dim3(128), 0, st, qkv, out, ws, lse, N, heads, p.whole_rounds, p.units, p.chunk);
    else
        EDV_LAUNCH((attn_spatial_kernel<1, 32>), grid, dim3(64), 0, st, qkv, out, ws, lse, N, heads, p.whole_rounds, p.units, p.chunk);
    EDV_LAUNCH_OK();
    if (p.leftover) {
        const int QB = p.nw * 32;
        EDV_LAUNCH(attn_combine_kernel, dim3((unsigned)p.leftover, (unsigned)(QB / 16)), dim3(256), 0, st, ws, out, lse, N, heads, QB, p.ntiles,
                           p.whole_rounds * p.grid, p.units, p.chunk);
        EDV_LAUNCH_OK();
    }
    return 0;
}

}  // namespace edv
