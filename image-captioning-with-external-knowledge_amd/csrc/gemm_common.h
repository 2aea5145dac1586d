// Device helpers shared by the GEMM kernels (gemm.hip: operands split or exact in the stager; gemm_ps.hip: B operand
// pre-split in global memory, everything staged by LDS-DMA): row maps, the head-split column map, the exact three-way
// bf16 split, and the epilogue.
#pragma once
#include "common.h"

namespace ick {
namespace {

struct RowMap {  // offset of logical row r:  goff(r / grp) + (r % grp) * rs
    int grp;
    int64_t gs;
    const int32_t* gmap;
    int64_t rs;
    __device__ __forceinline__ int64_t operator()(int r) const {
        if (grp <= 0) return (int64_t)r * rs;
        const int g = r / grp;
        const int i = r - g * grp;
        const int64_t gg = gmap ? (int64_t)gmap[g] : (int64_t)g;
        return gg * gs + (int64_t)i * rs;
    }
};

// Offsets of N valid rows at once: the group-map lookups of all rows are issued together (one memory round trip
// instead of one per row: a conditional load is followed by its own s_waitcnt).
template <int N>
__device__ __forceinline__ void map_rows(const RowMap& m, const int (&rows)[N], int64_t (&off)[N]) {
    if (m.grp <= 0) {            // uniform
#pragma unroll
        for (int i = 0; i < N; ++i) off[i] = (int64_t)rows[i] * m.rs;
        return;
    }
    int g[N], in[N];
#pragma unroll
    for (int i = 0; i < N; ++i) { g[i] = rows[i] / m.grp; in[i] = rows[i] - g[i] * m.grp; }
    if (m.gmap != nullptr) {     // uniform
        int gg[N];
#pragma unroll
        for (int i = 0; i < N; ++i) gg[i] = m.gmap[g[i]];
#pragma unroll
        for (int i = 0; i < N; ++i) g[i] = gg[i];
    }
#pragma unroll
    for (int i = 0; i < N; ++i) off[i] = (int64_t)g[i] * m.gs + (int64_t)in[i] * m.rs;
}

// Column offset inside an output row: the column itself, or the head-split scatter
// [segment][head][position][dhp] (see include/ick_amd.h).
__device__ __forceinline__ int64_t col_offset(const ick_gemm_args& p, int col) {
    if (p.hs_dh <= 0) return col;
    const int hd = p.hs_H * p.hs_dh;
    const int seg = col / hd, r = col - seg * hd;
    const int h = r / p.hs_dh, j = r - h * p.hs_dh;
    return ((int64_t)seg * p.hs_H + h) * ((int64_t)p.hs_S * p.hs_dhp) + j;
}

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
constexpr uint32_t kOobOffset = 0x80000000u;   // >= any extent the vector path accepts (< 2 GiB)

// ---- split operands (SPL): fp32 products on the bf16 matrix pipe --------------------------------------------------
// gfx950 has no reduced-precision fp32 MFMA, and the exact one runs at 1/16 of the bf16 rate.  Every fp32 value is the
// EXACT sum of three bf16 numbers (x = hi + mid + lo: 3 x 8 significand bits, round-to-nearest at each level, the
// residuals are exact fp32 subtractions), so a product a*b is the sum of nine bf16 x bf16 products, each of which the
// matrix pipe forms exactly and accumulates in fp32.  The three smallest (mid*lo, lo*mid, lo*lo: <= 2^-24 of a*b
// together) are dropped: six v_mfma_f32_16x16x32_bf16 per 16x16x32 block instead of eight v_mfma_f32_16x16x4_f32 at
// a sixteenth of the rate -- 2.7 x the matrix throughput at an error below one fp32 rounding of the product
// (measured against fp64 beside the exact path: tests/test_gemm_split_gpu.py).  The split happens once per staged
// element, between the global load and the LDS store; LDS holds three bf16 planes per operand tile:
//   k-contiguous operand: plane[row][32 k] in 64-byte rows, 16-byte chunk c of row r at chunk c ^ ((-(r >> 2)) & 3):
//     the ds_read_b128 of the MFMA operand (lane (i, q) takes k = 8q..8q+7 of row i) is conflict-free for the four
//     16-lane groups the LDS serves it in, and so are the ds_write_b64 of the stager;
//   k-major operand: plane[k][rows] with (2 rows + 64)-byte k lines, stored as it arrives (ds_write_b64 of 4 rows) and
//     read with ds_read_b64_tr_b16, the transposing read: two of them deliver the same 8-k operand.  The 16-row blocks
//     of k lines 8-15 and 24-31 are swapped pairwise: the two 16-lane groups a transposing read serves together (k
//     lines q and q + 8) then fall on different banks (bank search: DESIGN.md).
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4_ptr;

__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{a, b}, bf16x2_t));
}
// (a, b) -> packed (hi, mid, lo) pairs, a in the low half
__device__ __forceinline__ void split3(float a, float b, uint32_t& hi, uint32_t& mid, uint32_t& lo) {
    hi = pack_bf16(a, b);
    const float ra = a - __builtin_bit_cast(float, hi << 16), rb = b - __builtin_bit_cast(float, hi & 0xffff0000u);
    mid = pack_bf16(ra, rb);
    const float sa = ra - __builtin_bit_cast(float, mid << 16), sb = rb - __builtin_bit_cast(float, mid & 0xffff0000u);
    lo = pack_bf16(sa, sb);
}
__device__ __forceinline__ int kc_swz(int row) { return (-(row >> 2)) & 3; }

// Epilogue of one workgroup tile whose waves hold TM x TN accumulator blocks of 16 x 16 (C/D map of the 16x16 MFMA:
// col = lane & 15, row = (lane >> 4) * 4 + reg): alpha, bias, ReLU, dropout, gate, accumulate / atomic, grouped or
// head-split rows.  wm / wn: the wave's block coordinates inside the tile; zid: K slice (bias only on slice 0).
template <int TM, int TN>
__device__ __forceinline__ void gemm_epilogue(const ick_gemm_args& p, f32x4 (&acc)[TM][TN], int m0, int n0, int wm, int wn,
                                              int fi, int fq, int zid) {
    // Epilogue.  C/D map of the 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg.
    const bool hs = p.hs_dh > 0;
    const RowMap cmap{p.c_grp, p.c_gs, p.c_gmap, hs ? (int64_t)p.hs_dhp : p.c_rs};
    const int64_t row_bias = hs ? (int64_t)p.hs_s0 * p.hs_dhp : 0;
    const bool relu = p.flags & ICK_GEMM_RELU;
    const int mode = (p.flags & ICK_GEMM_ATOMIC) ? 2 : ((p.flags & ICK_GEMM_ACCUM) ? 1 : 0);
    float bv[TN];
    int cols[TN];
    int64_t co[TN];
#pragma unroll
    for (int b = 0; b < TN; ++b) {
        cols[b] = n0 + (wn * TN + b) * 16 + fi;
        co[b] = col_offset(p, cols[b]);
        bv[b] = (p.bias != nullptr && zid == 0 && cols[b] < p.N) ? p.bias[cols[b]] : 0.f;
    }
    const float alpha = p.alpha;
    int64_t coff[TM][4];
    int rowid[TM][4];
    const Dropout drop = make_dropout(p.drop_p, epoch_seed(p.drop_seed, p.drop_epoch), p.drop_site);
    {
        int rws[TM * 4];
        int64_t mo[TM * 4];
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                rowid[a][r] = m0 + (wm * TM + a) * 16 + fq * 4 + r;
                rws[a * 4 + r] = min(rowid[a][r], p.M - 1);
            }
        map_rows<TM * 4>(cmap, rws, mo);
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) coff[a][r] = rowid[a][r] < p.M ? mo[a * 4 + r] + row_bias : -1;
    }
    // Values first, memory second: vmcnt counts loads and stores in one queue, so a load between two stores (the gate,
    // the old value of an accumulating epilogue) makes every row wait for the previous row's stores to be acknowledged
    // (measured on the cross K/V projection: 4.1 of a workgroup's 22 us).  Every load of the epilogue is issued before
    // its first store.
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                float v = acc[a][b][r] * alpha + bv[b];
                if (relu) v = fmaxf(v, 0.f);
                if (drop.on()) v *= drop.mask((uint32_t)rowid[a][r] * (uint32_t)p.N + (uint32_t)cols[b]);
                acc[a][b][r] = v;
            }
    if (p.gate != nullptr) {      // uniform
        float g[TM][4][TN];
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int b = 0; b < TN; ++b)
                    // unconditional loads (element 0 stands in for what lies outside the matrix; never stored)
                    g[a][r][b] = p.gate[(coff[a][r] >= 0 && cols[b] < p.N) ? (int64_t)rowid[a][r] * p.gate_rs + cols[b] : 0];
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int b = 0; b < TN; ++b) acc[a][b][r] = g[a][r][b] > 0.f ? acc[a][b][r] * p.gate_scale : 0.f;
    }
    if (mode == 1) {              // uniform
        float old[TM][4][TN];
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int b = 0; b < TN; ++b)
                    old[a][r][b] = p.C[(coff[a][r] >= 0 && cols[b] < p.N) ? coff[a][r] + co[b] : 0];
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int b = 0; b < TN; ++b) acc[a][b][r] += old[a][r][b];
    }
#pragma unroll
    for (int a = 0; a < TM; ++a) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (coff[a][r] < 0) continue;
            float* crow = p.C + coff[a][r];
            if (mode != 2) {
#pragma unroll
                for (int b = 0; b < TN; ++b)
                    if (cols[b] < p.N) crow[co[b]] = acc[a][b][r];
            } else {
#pragma unroll
                for (int b = 0; b < TN; ++b)
                    if (cols[b] < p.N) atomicAdd(crow + co[b], acc[a][b][r]);
            }
        }
    }
}

}  // namespace
}  // namespace ick
