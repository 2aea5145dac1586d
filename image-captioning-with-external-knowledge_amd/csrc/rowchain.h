// Shared device code of the row-chain kernels (rowchain.hip: forward, rowchain_bwd.hip: backward): the 8-row GEMM on
// v_mfma_f32_4x4x1 with a broadcast A operand and packed weights.  See rowchain.hip for the design notes.
#pragma once
#include "common.h"

namespace ick {
namespace rowchain {

constexpr int kRows = 8;          // rows per workgroup
constexpr int kWaves = 16;
constexpr int kThreads = kWaves * 64;
constexpr int kMaxK = 512;        // widest GEMM input of the forward chains (linear2: dim_feedforward)
constexpr int kLdx = kMaxK + 16 + 4;
constexpr int kMaxD = 320;        // LayerNorm width (5 columns per lane)
constexpr int kMaxN2 = 1024;      // widest second GEMM (in_proj: 3 d)
constexpr int kPartFloats = kRows * 64 * kWaves;   // every (slab, K split) pair is one wave: 8 rows x 64 columns each

struct GemmPlan {   // how the 16 (slab, K split) units cover N columns x K
    int nslab, splits;
};
__host__ __device__ __forceinline__ GemmPlan plan_for(int N, int K) {
    GemmPlan g;
    g.nslab = (N + 63) / 64;
    g.splits = kWaves / g.nslab;
    if (g.splits < 1) g.splits = 1;
    const int maxs = (K + 15) / 16;
    if (g.splits > maxs) g.splits = maxs;
    if (g.splits > 4) g.splits = 4;
    return g;
}

#define ICK_MF(U)                                                                            \
    acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a0, b[(U) >> 2][(U) & 3], acc0, 4, U, 0);        \
    acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a1, b[(U) >> 2][(U) & 3], acc1, 4, U, 0);

// acc[row] (lane = column of the wave's slab) = Xs[row][k range of the wave's split] . W[col][k] from the packed copy
// Wp; Xs is zero beyond K.  Returns false for a wave without work (more waves than slabs x splits).
struct Slab { int slab, h; };
__device__ __forceinline__ Slab unit_of(const GemmPlan& g, int unit) {      // unit = slab + nslab * K split (scalar)
    return Slab{unit % g.nslab, unit / g.nslab};
}
__device__ __forceinline__ Slab slab_of(const GemmPlan& g) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar: K offsets stay in SGPRs
    return unit_of(g, wave);
}
// begin(): descriptor + the loads of the first two K chunks (they depend on nothing but the weights, so a kernel issues
// them before it waits for the rows the GEMM multiplies: the first weight round trip hides behind that wait);
// run(): the pipelined K loop.
struct RowGemm {
    __amdgpu_buffer_rsrc_t rsrc;
    int voff, nchunk, kb, slab_bytes;
    f32x4 bq[3][4];

    // Software pipeline, two chunks ahead, without branches around the loads (the compiler's s_waitcnt counting
    // only stays exact in straight-line code): a chunk beyond this wave's K range is fetched from beyond the
    // descriptor's extent (no memory access, zeros).
    __device__ __forceinline__ void load(f32x4 (&b)[4], int c) {
        const int base = c < nchunk ? (kb + 16 * c) * 256 : slab_bytes;      // scalar; 16 k = 4 KiB
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            b[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, base + j * 1024, 0));
        }
    }

    __device__ __forceinline__ void begin(int K, const float* __restrict__ Wp, const GemmPlan g, const Slab w) {
        // K split in whole 16-k chunks, as even as they go (19 chunks over 3 splits: 7 + 6 + 6, not 7 + 7 + 5)
        const int chunks = (K + 15) >> 4;
        const int per = chunks / g.splits, extra = chunks - per * g.splits;
        const int c0 = w.h * per + min(w.h, extra);
        nchunk = w.h < g.splits ? per + (w.h < extra ? 1 : 0) : 0;      // 0: a wave without work
        kb = nchunk > 0 ? 16 * c0 : 0;
        const int K16 = (K + 15) & ~15;
        slab_bytes = K16 * 64 * 4;                 // one slab of the packed copy
        const int slab = w.h < g.splits ? w.slab : 0;
        rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Wp) + (size_t)slab * K16 * 64, (short)0, slab_bytes,
                                                 0x00020000);
        voff = (threadIdx.x & 63) * 16;
        load(bq[0], 0);
        load(bq[1], 1);
        __builtin_amdgcn_sched_barrier(0);
    }

    // a0 / a1: the A operands of chunk c (X[4 rows][16 k] of the two row groups), read from LDS one chunk ahead
    __device__ __forceinline__ void mfma(const f32x4 (&b)[4], float a0, float a1, f32x4& acc0, f32x4& acc1) {
        ICK_MF(0) ICK_MF(1) ICK_MF(2) ICK_MF(3) ICK_MF(4) ICK_MF(5) ICK_MF(6) ICK_MF(7)
        ICK_MF(8) ICK_MF(9) ICK_MF(10) ICK_MF(11) ICK_MF(12) ICK_MF(13) ICK_MF(14) ICK_MF(15)
    }

    __device__ __forceinline__ void run(const float* Xs, int ldx, f32x4& acc0, f32x4& acc1) {
        const int lane = threadIdx.x & 63;
        acc0 = f32x4{0.f, 0.f, 0.f, 0.f};
        acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
        // chunk c of this wave's K range; the pointer stays inside the (zero padded) row for c up to nchunk
        const float* xa = Xs + (lane & 3) * ldx + (lane >> 2) + kb;
        const int last = max(nchunk - 1, 0);
        float a0 = xa[0], a1 = xa[4 * ldx];
        // sched_barrier: the machine scheduler otherwise sinks the prefetches down to their uses (vmcnt(0) per chunk)
#define ICK_STEP(LD, LC, MF, MC)                                                   \
    load(bq[LD], LC);                                                             \
    {                                                                             \
        const int nx = 16 * min((MC) + 1, last);                                  \
        const float n0 = xa[nx], n1 = xa[4 * ldx + nx];                           \
        __builtin_amdgcn_sched_barrier(0);                                        \
        mfma(bq[MF], a0, a1, acc0, acc1);                                         \
        a0 = n0; a1 = n1;                                                         \
    }                                                                             \
    __builtin_amdgcn_sched_barrier(0);
        int c = 0;
        for (; c + 3 <= nchunk; c += 3) {
            ICK_STEP(2, c + 2, 0, c)
            ICK_STEP(0, c + 3, 1, c + 1)
            ICK_STEP(1, c + 4, 2, c + 2)
        }
        if (c < nchunk) {          // one or two chunks left, already in flight
            const int nx = 16 * min(c + 1, last);
            const float n0 = xa[nx], n1 = xa[4 * ldx + nx];
            mfma(bq[0], a0, a1, acc0, acc1);
            __builtin_amdgcn_sched_barrier(0);
            if (c + 1 < nchunk) mfma(bq[1], n0, n1, acc0, acc1);
        }
#undef ICK_STEP
    }
};

// a / b for 0 <= a < 2^22, 0 < b < 2^22 without the ~40-instruction integer division sequence (every VALU instruction of
// a 16-wave workgroup costs 16 cycles of its CU): float quotient, then one correction step
__device__ __forceinline__ int small_div(int a, int b) {
    int q = (int)((float)a * __builtin_amdgcn_rcpf((float)b));
    const int r = a - q * b;
    q += (r >= b) ? 1 : 0;
    q -= (r < 0) ? 1 : 0;
    return q;
}

// Row offsets of the 8 rows of a workgroup under the (grp, gs, rs) addressing, without a division per row.
struct RowOff {
    int g, i, grp;
    int64_t gs, rs;
    __device__ __forceinline__ RowOff(int row0, int grp_, int64_t gs_, int64_t rs_) : grp(grp_), gs(gs_), rs(rs_) {
        if (grp > 0) { g = small_div(row0, grp); i = row0 - g * grp; } else { g = 0; i = row0; }
    }
    __device__ __forceinline__ int64_t next() {    // offset of the current row; advances to the following one
        const int64_t o = (int64_t)g * gs + (int64_t)i * rs;
        ++i;
        if (grp > 0 && i >= grp) { i = 0; ++g; }
        return o;
    }
};


}  // namespace rowchain
}  // namespace ick
