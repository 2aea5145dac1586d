// One KV-cached decode step of DecoderTransformer.predict() (geo-aware/models.py:389-443,
// knowledge-aware/models.py:545-608) for R independent rows (R = captions, or captions x beams) in
// 3 launches per decoder layer + 3 for the score head, instead of ~37 dependent launches:
//
//   dec_self_kernel   (row, head)   LN-on-load -> q|k|v rows of the head (GEMV) -> cache write -> causal
//                                   self-attention over the cached positions -> this head's slice of out_proj
//   dec_cross_kernel  (row, head)   LN-on-load -> q rows of the head -> attention over the S memory rows
//                                   [image ; entity ; fact] (K/V streamed once, coalesced 1 KiB per wave
//                                   instruction) -> this head's slice of out_proj
//   dec_ffn_kernel    (row, chunk)  LN-on-load -> 64 hidden units of linear1 + ReLU -> their slice of linear2
//   dec_head_kernel   (row)         final LayerNorm -> h (x predicate gate) -> pointer scores over entities / facts
//   dec_vocab_kernel  (16 words x 32 rows)  vocabulary logits on the fp32 MFMA, K split over the 4 waves,
//                                   per-tile top-2 candidates
//   dec_select_kernel (row)         top-2 over candidates + pointer scores, predict()'s bookkeeping (n-gram
//                                   clean-up, <end>), embedding + position code of the next token
//
// "LN-on-load": a block does not apply its closing residual + LayerNorm itself (that would need all heads /
// chunks of the row in one workgroup, i.e. too few workgroups to stream K/V at HBM rate).  It leaves its
// out-projection as per-head (per-chunk) partial rows; the NEXT kernel forms
//       x = LayerNorm(res + bias + sum_p partial_p) * gamma + beta
// while loading its input row (300 floats: a few hundred flops per workgroup), and the workgroup with
// head/chunk 0 stores x as the residual of the following block.  Sums run in a fixed order: results are
// deterministic (no float atomics anywhere on this path).
//
// Bounds: per token the path moves the cross K/V of every layer once (S x 32 floats x 2 per row, head, layer:
// 53 MB at B=32, S=216 -> HBM-bound) plus ~16 MB of weights that stay L2 / Infinity-Cache
// resident.  Every workgroup's weight loads are issued before the data they multiply is ready (they do not
// depend on it), so a kernel is ~3 memory round trips long.
#include "common.h"

#include <cstdio>
#include <cstdlib>

// Diagnostic build (-DICK_DECODE_STAMPS, tools/debug/decode_stamps.py): workgroup (0, 0) of every decode kernel
// records the shader clock at its phase boundaries.  Compiled out of the product library.
#ifdef ICK_DECODE_STAMPS
__device__ unsigned long long ick_stamps[8][16];
#define ICK_STAMP(kern, i)                                                                          \
    do {                                                                                            \
        if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) ick_stamps[kern][i] = __builtin_amdgcn_s_memtime(); \
    } while (0)
extern "C" int ick_debug_read_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ick_stamps), sizeof(ick_stamps));
}
#else
#define ICK_STAMP(kern, i)
#endif

namespace ick {
namespace {

constexpr int kDMax = 320;   // model width limit of this path: 5 float4 per lane of a 16-lane row group
constexpr int kD4Max = kDMax / 4;
constexpr int kSMax = 1024;  // memory rows (cross-attention keys)
constexpr int kMLMax = 128;  // caption positions (self-attention keys)
constexpr int kDhp = 32;     // padded head width of the head-major K/V layouts
constexpr int kPartsMax = 16;
constexpr int kNumCU = 256;

using f32x2 = __attribute__((ext_vector_type(2))) float;

struct RowSrc {
    const float* res;     // (R, d) residual rows; the input itself when nparts == 0
    const float* part;    // (R, nparts, d) partial out-projection rows of the previous block
    const float* bias;    // (d) bias of that out-projection
    const float* gamma;   // (d) LayerNorm affine
    const float* beta;
    float* out;           // (R, d) normalised rows (written by head/chunk 0), may be null
    int nparts;
    float eps;
};

// What bounds these kernels (in-kernel stamps + ISA counts, round 3): NOT the bytes -- a workgroup of four waves has one
// wave per SIMD, and a lone wave issues one instruction per ~4 cycles whatever its type; ~1500 instructions of address
// arithmetic, loads and reductions in front of the first barrier were 5-6 k cycles.  Hence: (1) 512-thread workgroups
// (two waves per SIMD share the issue slots, every wave holds half the weight registers: no accumulator-file spills,
// whose copies wait for the loads they copy); (2) the wave index is read with readfirstlane, so row and weight base
// addresses are scalar and the loads take the scalar-base + 32-bit-offset form; (3) no branch per source row: clamped
// loads weighted by 0 / 1; (4) the early-exit word is waited for after the loads have been issued.
constexpr int kNT = 512;           // threads of a block kernel
constexpr int kNW = kNT / 64;      // waves

// Workgroups are dealt round-robin over the 8 XCDs (linear id % 8), each with its own L2 that starts a kernel cold.
// The (weight slice, row group) units are therefore numbered so that the workgroups of one XCD are CONSECUTIVE units,
// slice-major: an XCD then pulls 2-3 heads' weights (or one FFN chunk) through its L2 instead of all of them -- the
// beyond-L2 traffic of a decode step drops from ~146 MB to ~82 MB.  Placement affects speed only, never results.
__device__ __forceinline__ void xcd_unit(int& slice, int& group, int ngroups) {
    const int nwg = gridDim.x * gridDim.y, lin = blockIdx.y * gridDim.x + blockIdx.x;
    const int xcd = lin & 7, slot = lin >> 3;
    const int u = xcd * (nwg >> 3) + min(xcd, nwg & 7) + slot;          // XCD x owns units [start_x, start_x + count_x)
    slice = u / ngroups;
    group = u - slice * ngroups;
}

__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
// uniform base + 32-bit byte offset: global_load_dwordx4 v, v_off, s[base:base+1]
__device__ __forceinline__ float4 ld4o(const float* base, uint32_t byteoff) {
    return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + byteoff);
}
__device__ __forceinline__ float4 ld4o_stream(const float* base, uint32_t byteoff) {   // read once per step by one workgroup
#ifdef ICK_DECODE_NT_KV
    const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(base) + byteoff));
    return make_float4(v[0], v[1], v[2], v[3]);
#else
    return ld4o(base, byteoff);
#endif
}
__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ void f4add(float4& a, const float4& b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }

// Lane exchanges inside a row of 16 lanes as DPP operands (one VALU instruction each) instead of ds_bpermute round
// trips through the LDS crossbar (~100 cycles each, four in a row per dot product): quad_perm [1,0,3,2] / [2,3,0,1]
// pair lanes inside a quad, row_half_mirror / row_mirror then pair quads and halves (every lane of a quad / half
// already holds the same partial result), row_ror:8 swaps the halves of a row.  Full waves only (common.h).
template <int CTRL>
__device__ __forceinline__ float dppf(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
template <int CTRL>
__device__ __forceinline__ int dppi(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, false); }
constexpr int kDppXor1 = 0xB1, kDppXor2 = 0x4E, kDppHalfMirror = 0x141, kDppMirror = 0x140, kDppRor8 = 0x128;
__device__ __forceinline__ float sum8(float v) {      // over aligned groups of 8 lanes, result in every lane
    v += dppf<kDppXor1>(v); v += dppf<kDppXor2>(v); v += dppf<kDppHalfMirror>(v);
    return v;
}
__device__ __forceinline__ float sum16(float v) { v = sum8(v); v += dppf<kDppMirror>(v); return v; }

__device__ __forceinline__ float dot4(const float4& a, const float4& b) {
    return fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, a.w * b.w)));
}
// two-lane packed form (v_pk_fma_f32): the even and odd elements run as two independent chains
__device__ __forceinline__ f32x2 pkfma(const float4& a, const float4& b, f32x2 acc) {
    acc = __builtin_elementwise_fma(f32x2{a.x, a.y}, f32x2{b.x, b.y}, acc);
    return __builtin_elementwise_fma(f32x2{a.z, a.w}, f32x2{b.z, b.w}, acc);
}

// ---------------------------------------------------------------------------------------------------------
// "LN-on-load" of the G rows of a workgroup: x = LayerNorm(res + bias + sum_p partial_p) (or res itself when the source
// has no partial rows).  The NWAVES / G waves of a row each fetch and sum a slice of its source rows (residual, bias,
// partial 0, 1, ...: lane l holds float4 columns l and 64 + l), leave the slice sums in LDS, and after one barrier the
// row's first wave adds them in slice order and normalises: mean and variance are wave reductions (DPP + v_readlane).
// issue() only starts the loads -- the kernels issue them FIRST and their weight / K / V streams afterwards: vector-memory
// results return in issue order, so the rows are normalised while the streams are still arriving.  Every base address
// is wave-uniform (scalar); sums run in a fixed order: deterministic.
// ---------------------------------------------------------------------------------------------------------
template <int G, int NWAVES>
struct RowGather {
    static constexpr int WPR = NWAVES / G;                                  // waves per row (G = 3, 5: some waves idle)
    static constexpr int NS = (12 + WPR - 1) / WPR < 6 ? (12 + WPR - 1) / WPR : 6;   // source rows per wave and sweep
    float4 a[NS], b[NS];
    float4 gm0, gm1, bt0, bt1;

    __device__ __forceinline__ static const float* src_row(const RowSrc& s, int64_t row, int d, int q) {
        return q == 0 ? s.res + row * d : (q == 1 ? s.bias : s.part + (row * s.nparts + (q - 2)) * d);
    }
    // source rows q0 .. q0 + NS - 1 of `row` (past the last: a copy of it, weighted 0 by the sum)
    __device__ __forceinline__ void sweep(const RowSrc& s, int64_t row, int d, int q0, int nrows) {
        const int lane = threadIdx.x & 63, d4 = d >> 2;
        const uint32_t o0 = 16u * (uint32_t)min(lane, d4 - 1), o1 = 16u * (uint32_t)min(64 + lane, d4 - 1);
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            const float* src = src_row(s, row, d, min(q0 + j, nrows - 1));
            a[j] = ld4o(src, o0);
            if (d4 > 64) b[j] = ld4o(src, o1);
        }
    }
    __device__ __forceinline__ void issue(const RowSrc& s, int64_t r0, int R, int d) {
        const int lane = threadIdx.x & 63, wave = wave_id(), d4 = d >> 2;
        if (wave >= G * WPR) return;
        const int g = wave / WPR, wr = wave - g * WPR;
        const int64_t row = min(r0 + g, (int64_t)R - 1);
        const uint32_t o0 = 16u * (uint32_t)min(lane, d4 - 1), o1 = 16u * (uint32_t)min(64 + lane, d4 - 1);
        const int nrows = s.nparts > 0 ? s.nparts + 2 : 1;
        if (wr * NS < nrows) sweep(s, row, d, wr * NS, nrows);              // this slice holds source rows (uniform)
        gm0 = gm1 = bt0 = bt1 = f4zero();
        if (wr == 0 && s.nparts > 0) {
            gm0 = ld4o(s.gamma, o0); bt0 = ld4o(s.beta, o0);
            if (d4 > 64) { gm1 = ld4o(s.gamma, o1); bt1 = ld4o(s.beta, o1); }
        }
    }
    // psum: LDS float4 [G][WPR][kD4Max]; follow with a barrier
    __device__ __forceinline__ void slice_sums(const RowSrc& s, int64_t r0, int R, int d, float4* psum) {
        const int lane = threadIdx.x & 63, wave = wave_id(), d4 = d >> 2;
        if (wave >= G * WPR) return;
        const int g = wave / WPR, wr = wave - g * WPR;
        const bool wide = d4 > 64;
        const int nrows = s.nparts > 0 ? s.nparts + 2 : 1;
        f32x2 z0l = {0.f, 0.f}, z0h = {0.f, 0.f}, z1l = {0.f, 0.f}, z1h = {0.f, 0.f};
        auto add_sweep = [&](int q0) {
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                const float f = q0 + j < nrows ? 1.f : 0.f;                 // scalar
                const f32x2 ff = {f, f};
                z0l = __builtin_elementwise_fma(f32x2{a[j].x, a[j].y}, ff, z0l);
                z0h = __builtin_elementwise_fma(f32x2{a[j].z, a[j].w}, ff, z0h);
                if (wide) {
                    z1l = __builtin_elementwise_fma(f32x2{b[j].x, b[j].y}, ff, z1l);
                    z1h = __builtin_elementwise_fma(f32x2{b[j].z, b[j].w}, ff, z1h);
                }
            }
        };
        // The sweep issue() requested is summed in straight-line code: its s_waitcnt then counts only the row loads, not
        // the weight / K / V streams requested after them (inside a loop the compiler waits for vmcnt(0), i.e. for the
        // whole stream).  Sources of more than 12 rows take the loop below.
        if (wr * NS < nrows) add_sweep(wr * NS);
        for (int q0 = (wr + WPR) * NS; q0 < nrows; q0 += WPR * NS) {
            sweep(s, min(r0 + g, (int64_t)R - 1), d, q0, nrows);
            add_sweep(q0);
        }
        float4* dst = psum + (g * WPR + wr) * kD4Max;
        dst[lane] = make_float4(z0l.x, z0l.y, z0h.x, z0h.y);
        if (lane < kD4Max - 64) dst[64 + lane] = make_float4(z1l.x, z1l.y, z1h.x, z1h.y);
    }
    // after the barrier: the first wave of every row normalises it into xs[g] (LDS, kDMax floats, zero beyond d) and,
    // when `writer`, into s.out; follow with a barrier
    __device__ __forceinline__ void finish(const RowSrc& s, int64_t r0, int R, int d, const float4* psum, float* xs,
                                           bool writer) {
        const int lane = threadIdx.x & 63, wave = wave_id(), d4 = d >> 2;
        if (wave >= G * WPR) return;
        const int g = wave / WPR, wr = wave - g * WPR;
        if (wr != 0) return;
        const int64_t row = min(r0 + g, (int64_t)R - 1);
        const bool ok0 = lane < d4, ok1 = 64 + lane < d4;
        const int l1 = min(64 + lane, kD4Max - 1);
        float4 z0 = psum[g * WPR * kD4Max + lane], z1 = psum[g * WPR * kD4Max + l1];
#pragma unroll
        for (int w = 1; w < WPR; ++w) {
            f4add(z0, psum[(g * WPR + w) * kD4Max + lane]);
            f4add(z1, psum[(g * WPR + w) * kD4Max + l1]);
        }
        if (!ok0) z0 = f4zero();
        if (!ok1) z1 = f4zero();
        if (s.nparts > 0) {
            const float mean = wave_sum(((z0.x + z0.y) + (z0.z + z0.w)) + ((z1.x + z1.y) + (z1.z + z1.w))) / (float)d;
            float v = 0.f;
            if (ok0) {
                v = fmaf(z0.x - mean, z0.x - mean, v); v = fmaf(z0.y - mean, z0.y - mean, v);
                v = fmaf(z0.z - mean, z0.z - mean, v); v = fmaf(z0.w - mean, z0.w - mean, v);
            }
            if (ok1) {
                v = fmaf(z1.x - mean, z1.x - mean, v); v = fmaf(z1.y - mean, z1.y - mean, v);
                v = fmaf(z1.z - mean, z1.z - mean, v); v = fmaf(z1.w - mean, z1.w - mean, v);
            }
            const float rstd = rsqrtf(wave_sum(v) / (float)d + s.eps);
            z0.x = (z0.x - mean) * rstd * gm0.x + bt0.x; z0.y = (z0.y - mean) * rstd * gm0.y + bt0.y;
            z0.z = (z0.z - mean) * rstd * gm0.z + bt0.z; z0.w = (z0.w - mean) * rstd * gm0.w + bt0.w;
            z1.x = (z1.x - mean) * rstd * gm1.x + bt1.x; z1.y = (z1.y - mean) * rstd * gm1.y + bt1.y;
            z1.z = (z1.z - mean) * rstd * gm1.z + bt1.z; z1.w = (z1.w - mean) * rstd * gm1.w + bt1.w;
            if (!ok0) z0 = f4zero();
            if (!ok1) z1 = f4zero();
        }
        float* x = xs + g * kDMax;
        reinterpret_cast<float4*>(x)[lane] = z0;
        if (lane < kD4Max - 64) reinterpret_cast<float4*>(x)[64 + lane] = z1;
        if (writer && r0 + g < R && s.out != nullptr) {
            if (ok0) reinterpret_cast<float4*>(s.out + row * d)[lane] = z0;
            if (ok1) reinterpret_cast<float4*>(s.out + row * d)[64 + lane] = z1;
        }
    }
};

// Dot products of up to 4 * kNW * NPASS weight rows (k contiguous) with the G LDS vectors xs[g]: 16 lanes per weight
// row, four weight rows per wave and pass; the weights stay in registers for all G rows.  load() only issues the weight
// loads (nothing depends on xs), run() consumes them.  rowoff(r): element offset of weight row r (and bias index
// bidx(r)); both are evaluated per lane on 32-bit integers.
template <int NPASS>
struct RowDot {
    float4 w[NPASS][5];
    float bv[NPASS];
    template <typename RowIdx>
    __device__ __forceinline__ void load(const float* __restrict__ W, const float* __restrict__ bias, int ld,
                                         RowIdx rowidx, int nrows, int d4) {
        const int lane = threadIdx.x & 63, wave = wave_id(), sub = lane >> 4, i = lane & 15;
        uint32_t col[5];
#pragma unroll
        for (int it = 0; it < 5; ++it) col[it] = 16u * (uint32_t)min(i + 16 * it, d4 - 1);     // beyond d: xs is zero there
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            const int r = min(4 * kNW * p + 4 * wave + sub, nrows - 1);
            const uint32_t ri = (uint32_t)rowidx(r);
            const uint32_t ro = ri * (uint32_t)ld * 4u;
#pragma unroll
            for (int it = 0; it < 5; ++it) w[p][it] = ld4o(W, ro + col[it]);
            bv[p] = bias ? bias[ri] : 0.f;     // fetched with the weights: a load inside run() would stall it
        }
    }
    // ys[g * ystride + r] = (dot(W[rowidx(r)], xs[g]) + bias[rowidx(r)]) * scale   (relu: max(., 0))
    template <int G>
    __device__ __forceinline__ void run(const float* xs, int nrows, float* ys, int ystride, float scale, bool relu) const {
        const int lane = threadIdx.x & 63, wave = wave_id(), sub = lane >> 4, i = lane & 15;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float4 xr[5];
#pragma unroll
            for (int it = 0; it < 5; ++it) xr[it] = reinterpret_cast<const float4*>(xs + g * kDMax)[i + 16 * it];
#pragma unroll
            for (int p = 0; p < NPASS; ++p) {
                f32x2 acc = {0.f, 0.f};
#pragma unroll
                for (int it = 0; it < 5; ++it) acc = pkfma(w[p][it], xr[it], acc);
                const float t = sum16(acc.x + acc.y);
                const int r = 4 * kNW * p + 4 * wave + sub;
                if (i == 0 && r < nrows) {
                    const float y = (t + bv[p]) * scale;
                    ys[g * ystride + r] = relu ? fmaxf(y, 0.f) : y;
                }
            }
        }
    }
};

// out[g][n] = sum_{j < nj} o[g][j] * Wt[(j0 + j) * ld + n] for n < d (Wt = transposed weight: one k per row, n
// contiguous).  NG = min(8, kNT / (d/4)) thread groups take every NG-th j (nj <= NG * JMAX); their float4 partial
// sums meet in LDS.  o[g][.] must be zero for nj <= j < NG * JMAX.
template <int JMAX>
struct ColDot {
    float4 w[JMAX];
    __device__ __forceinline__ void load(const float* __restrict__ Wt, int ld, int j0, int nj, int d4) {
        const int NG = min(kNT / d4, 8);
        const int tid = threadIdx.x;
        const int grp = min(tid / d4, NG - 1), c = tid - grp * d4 < d4 ? tid - grp * d4 : 0;
        const uint32_t co = 16u * (uint32_t)c;
#pragma unroll
        for (int jj = 0; jj < JMAX; ++jj) {
            const int j = min(grp + NG * jj, nj - 1);
            w[jj] = ld4o(Wt, (uint32_t)(j0 + j) * (uint32_t)ld * 4u + co);
        }
    }
    // o: LDS [G][ostride]; part: LDS float4[NG * G * d4] (<= G * kNT); out row g at out + g * out_gs floats, rows g >= nvalid skipped
    template <int G>
    __device__ __forceinline__ void run(const float* o, int ostride, int d4, float4* part, float* __restrict__ out,
                                        int64_t out_gs, int nvalid) const {
        const int NG = min(kNT / d4, 8);
        const int tid = threadIdx.x;
        const int grp = tid / d4, c = tid - grp * d4;
        if (grp < NG) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                f32x2 lo = {0.f, 0.f}, hi = {0.f, 0.f};
#pragma unroll
                for (int jj = 0; jj < JMAX; ++jj) {
                    const float oj = o[g * ostride + grp + NG * jj];
                    lo = __builtin_elementwise_fma(f32x2{oj, oj}, f32x2{w[jj].x, w[jj].y}, lo);
                    hi = __builtin_elementwise_fma(f32x2{oj, oj}, f32x2{w[jj].z, w[jj].w}, hi);
                }
                part[(grp * G + g) * d4 + c] = make_float4(lo.x, lo.y, hi.x, hi.y);
            }
        }
        __syncthreads();
        for (int idx = tid; idx < G * d4; idx += kNT) {
            const int g = idx / d4, t = idx - g * d4;
            float4 s = part[g * d4 + t];
            for (int gg = 1; gg < NG; ++gg) f4add(s, part[(gg * G + g) * d4 + t]);
            if (g < nvalid) reinterpret_cast<float4*>(out + g * out_gs)[t] = s;
        }
    }
};

__device__ __forceinline__ float4 mask_cols(float4 v, int c, int dh) {   // zero the pad columns of a 32-float row
    const int j = 4 * c;
    if (j + 0 >= dh) v.x = 0.f;
    if (j + 1 >= dh) v.y = 0.f;
    if (j + 2 >= dh) v.z = 0.f;
    if (j + 3 >= dh) v.w = 0.f;
    return v;
}

// ---------------------------------------------------------------------------------------------------------
// Attention of one WAVE over a range of key / value rows: lane (p8 = lane >> 3, c = lane & 7) holds float4 c of the
// positions p0 + 8 q + p8 (q < NPC) of a chunk in registers.  Scores, softmax and P.V never touch LDS and need no
// barrier; a range longer than one chunk (8 * NPC positions) continues with a running maximum (the usual rescaling of
// the sums; with one chunk it is the plain two-pass softmax).  NQ query rows can share the same keys / values (beam
// search: the hypotheses of a caption).
// ---------------------------------------------------------------------------------------------------------
template <int NPC>
struct KVRegs {
    float4 k[NPC], v[NPC];
    // offk(p) / offv(p): byte offset (from kbase / vbase, both wave-uniform) of the 32-float row of position p < len
    template <bool STREAM, typename OffK, typename OffV>
    __device__ __forceinline__ void load(const float* kbase, const float* vbase, int p0, int len, OffK offk, OffV offv) {
        const int lane = threadIdx.x & 63, p8 = lane >> 3, c = lane & 7;
#pragma unroll
        for (int q = 0; q < NPC; ++q) {
            const uint32_t o = offk(min(p0 + 8 * q + p8, len - 1)) + 16u * (uint32_t)c;    // past the range: a copy of the last row
            k[q] = STREAM ? ld4o_stream(kbase, o) : ld4o(kbase, o);
        }
#pragma unroll
        for (int q = 0; q < NPC; ++q) {
            const uint32_t o = offv(min(p0 + 8 * q + p8, len - 1)) + 16u * (uint32_t)c;
            v[q] = STREAM ? ld4o_stream(vbase, o) : ld4o(vbase, o);
        }
    }
    // the same with this lane's NPC row offsets already known (off[q]: byte offset of the row of position p0 + 8 q + p8,
    // clamped by the caller; keys and values share the addressing)
    __device__ __forceinline__ void load_at(const float* kbase, const float* vbase, const uint32_t (&off)[NPC]) {
        const uint32_t c16 = 16u * (uint32_t)(threadIdx.x & 7);
#pragma unroll
        for (int q = 0; q < NPC; ++q) k[q] = ld4o(kbase, off[q] + c16);
#pragma unroll
        for (int q = 0; q < NPC; ++q) v[q] = ld4o(vbase, off[q] + c16);
    }
};

struct AttnState {      // running softmax state of one query row in one wave
    float m, l;         // wave-uniform maximum; this lane's share of the sum of weights
    float4 acc;         // this lane's share of sum of weight x value (columns 4c .. 4c+3)
    __device__ __forceinline__ void init() { m = -INFINITY; l = 0.f; acc = f4zero(); }
};

// pnew >= 0: position pnew is not in memory yet -- its key / value (kn, vn: this lane's float4 c) come from registers.
// The chunk's registers are masked in place (pad columns, rows past the range).
template <int NPC, int NQ>
__device__ __forceinline__ void attend_chunk(KVRegs<NPC>& kv, const float4 (&q4)[NQ], int p0, int len, int dh,
                                             int pnew, const float4& kn, const float4& vn, AttnState (&st)[NQ]) {
    const int lane = threadIdx.x & 63, p8 = lane >> 3, c = lane & 7;
    float4 (&kk)[NPC] = kv.k;
    float4 (&vv)[NPC] = kv.v;
#pragma unroll
    for (int q = 0; q < NPC; ++q) {
        const int p = p0 + 8 * q + p8;
        // rows past the range were fetched from a clamped address: their bits must not reach the sums
        kk[q] = p == pnew ? kn : (p < len ? mask_cols(kk[q], c, dh) : f4zero());
        vv[q] = p == pnew ? vn : (p < len ? mask_cols(vv[q], c, dh) : f4zero());
    }
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
        float sc[NPC];
        float mc = -INFINITY;
#pragma unroll
        for (int q = 0; q < NPC; ++q) {
            const float s = sum8(dot4(q4[n], kk[q]));
            sc[q] = p0 + 8 * q + p8 < len ? s : -INFINITY;
            mc = fmaxf(mc, sc[q]);
        }
        mc = wave_max(mc);
        const float mnew = fmaxf(st[n].m, mc);
        const float rescale = st[n].m == -INFINITY ? 0.f : __expf(st[n].m - mnew);
        st[n].l *= rescale;
        st[n].acc.x *= rescale; st[n].acc.y *= rescale; st[n].acc.z *= rescale; st[n].acc.w *= rescale;
#pragma unroll
        for (int q = 0; q < NPC; ++q) {
            const float e = sc[q] == -INFINITY ? 0.f : __expf(sc[q] - mnew);
            st[n].l += e;
            st[n].acc.x = fmaf(e, vv[q].x, st[n].acc.x); st[n].acc.y = fmaf(e, vv[q].y, st[n].acc.y);
            st[n].acc.z = fmaf(e, vv[q].z, st[n].acc.z); st[n].acc.w = fmaf(e, vv[q].w, st[n].acc.w);
        }
        st[n].m = mnew;
    }
}

// Sum a wave's shares: afterwards every lane holds the wave's sum of weights and (in acc) the weighted values of its
// columns 4c .. 4c+3.  A position's weight sits in its 8 column lanes, hence the exact factor 1/8.
__device__ __forceinline__ void attend_reduce(AttnState& st) {
    st.l = wave_sum(st.l) * 0.125f;
    st.acc.x += dppf<kDppRor8>(st.acc.x); st.acc.y += dppf<kDppRor8>(st.acc.y);
    st.acc.z += dppf<kDppRor8>(st.acc.z); st.acc.w += dppf<kDppRor8>(st.acc.w);
#pragma unroll
    for (int off = 16; off < 64; off <<= 1) {
        st.acc.x += __shfl_xor(st.acc.x, off, 64); st.acc.y += __shfl_xor(st.acc.y, off, 64);
        st.acc.z += __shfl_xor(st.acc.z, off, 64); st.acc.w += __shfl_xor(st.acc.w, off, 64);
    }
}

struct LayerW {
    const float *in_w, *in_b;      // self: (3d, d) packed in_proj; cross: its first d rows (q)
    const float *out_wt;           // (d, d) TRANSPOSED out_proj weight (row = input feature)
    RowSrc src;                    // how this block's input row is formed
    float* part;                   // (R, H, d) partial out-projection rows written here
};

// ---------------------------------------------------------------------------------------------------------
// Greedy selection folded into the first self-attention block of the NEXT step (ick_decode_ctx.sel_state != NULL):
// the workgroups of a row group pick the row's token of step `step` = pos - 1 from the vocabulary kernel's per-tile
// candidates and the pointer scores, apply predict()'s bookkeeping (geo-aware/models.py:410-441: <end>, repeated n-gram
// clean-up, pointer masks) and embed the token -- one launch per token less (the selection kernel alone was 6.5 us of a
// ~85 us step).  Every head's workgroup repeats the decision from the SAME inputs: candidates and pointer scores of the
// previous kernels, and a 12-int window {output[step-1 .. step-5], runner-ups of steps step-1 .. step-3, ended flag} kept
// in two alternating buffers, so that the one workgroup that records the decision (head 0: output / history / flags /
// counter / caption buffer / next window) never writes what the others read.
// ---------------------------------------------------------------------------------------------------------
struct SelFuse {
    const float4* cand; int ntiles; const float* ptr;
    int64_t* output; int32_t* hist; int32_t* finished; int32_t* n_done; int64_t* next_token; int64_t* next_mask;
    int64_t* cap_buf;
    const int32_t* st_in; int32_t* st_out;           // (R, 12)
    const float *word_emb, *ee, *fe, *pe;
    int rows_per_sample, V, K, F, step, max_len, has_facts, end_token, pad_token;
    float emb_scale;
};

__device__ __forceinline__ uint64_t top_key(float v, int idx) {      // larger value first, then smaller index
    uint32_t u = __float_as_uint(v);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ((uint64_t)u << 32) | (uint32_t)(0x7fffffff - idx);
}
__device__ __forceinline__ uint64_t wave_max_u64(uint64_t k) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t lo = __shfl_xor((uint32_t)k, off, 64), hi = __shfl_xor((uint32_t)(k >> 32), off, 64);
        const uint64_t o = ((uint64_t)hi << 32) | lo;
        k = o > k ? o : k;
    }
    return k;
}

// xs: LDS [G][kDMax] receives the embedded rows; keys: LDS uint64 [G][kNW][2]; follow with a barrier
template <int G>
__device__ __forceinline__ void fused_select(const SelFuse& f, int64_t r0, int R, int d, int pos, float* xs, uint64_t* keys,
                                             float* xa_out, bool recorder) {
    constexpr int WPR = kNW / G;                                   // waves per row
    const int lane = threadIdx.x & 63, wave = wave_id();
    const int g = min(wave / WPR, G - 1), wr = wave - g * WPR;
    const bool mine = wave < G * WPR;
    const int64_t row = min(r0 + g, (int64_t)R - 1);
    const int np = f.K + f.F, i = f.step;
    // this wave's share of the row's candidates: vocabulary tiles, then pointer scores
    float v1 = -INFINITY, v2 = -INFINITY;
    int i1 = 0x7fffffff, i2 = 0x7fffffff;
    for (int t = wr * 64 + lane; mine && t < f.ntiles + np; t += 64 * WPR) {
        float a1, a2; int j1, j2;
        if (t < f.ntiles) {
            const float4 c = f.cand[row * f.ntiles + t];
            a1 = c.x; j1 = __float_as_int(c.y); a2 = c.z; j2 = __float_as_int(c.w);
        } else {
            a1 = f.ptr[row * np + (t - f.ntiles)]; j1 = f.V + (t - f.ntiles); a2 = -INFINITY; j2 = 0x7fffffff;
        }
        // merge (a1, j1) >= (a2, j2) into the lane's running pair
        const bool b1 = top_key(a1, j1) > top_key(v1, i1);
        const float n2v = b1 ? v1 : a1; const int n2i = b1 ? i1 : j1;       // the loser of the two firsts
        v1 = b1 ? a1 : v1; i1 = b1 ? j1 : i1;
        float s2v = b1 ? a2 : v2; int s2i = b1 ? j2 : i2;                    // the winner's own second
        const bool b2 = top_key(n2v, n2i) > top_key(s2v, s2i);
        v2 = b2 ? n2v : s2v; i2 = b2 ? n2i : s2i;
    }
    const uint64_t kb = wave_max_u64(top_key(v1, i1));
    const int best_i = 0x7fffffff - (int)(uint32_t)kb;
    const uint64_t ks = wave_max_u64(i1 == best_i ? top_key(v2, i2) : top_key(v1, i1));
    if (mine && lane == 0) { keys[(g * kNW + wr) * 2] = kb; keys[(g * kNW + wr) * 2 + 1] = ks; }
    __syncthreads();
    if (!mine || wr != 0) return;
    // the row's first wave: merge the waves' pairs, decide, embed
    uint64_t B = keys[(g * kNW) * 2], S2 = keys[(g * kNW) * 2 + 1];
#pragma unroll
    for (int w = 1; w < WPR; ++w) {
        const uint64_t b = keys[(g * kNW + w) * 2], s2 = keys[(g * kNW + w) * 2 + 1];
        if (b > B) { S2 = B > s2 ? B : s2; B = b; } else { S2 = b > S2 ? b : S2; }
    }
    const int best = 0x7fffffff - (int)(uint32_t)B;
    int second = 0x7fffffff - (int)(uint32_t)S2;
    if (second == 0x7fffffff) second = best;                                    // no runner-up at all
    const int32_t* st = f.st_in + row * 12;
    int po[5], ph[3];
#pragma unroll
    for (int q = 0; q < 5; ++q) po[q] = st[q];
#pragma unroll
    for (int q = 0; q < 3; ++q) ph[q] = st[5 + q];
    const int fin = st[8];
    int tok = 0, msk = 0, cur = 0, nfin = fin;
    int rw1 = po[0], rw2 = po[1], rw3 = po[2];      // output[i-1 .. i-3] after the clean-up
    int nrw = 0;
    if (!fin) {
        if (best == f.end_token) {
            cur = best; nfin = 1;
        } else {
            cur = best;
            // repeated n-gram clean-up (geo-aware/models.py:421-435)
            if (i > 0 && cur == po[0]) {
                cur = second;
            } else if (i > 2 && cur == po[1] && po[0] == po[2]) {
                cur = second; rw1 = ph[0]; nrw = 1;
            } else if (i > 4 && cur == po[2] && po[0] == po[3] && po[1] == po[4]) {
                cur = second; rw1 = ph[0]; rw2 = ph[1]; rw3 = ph[2]; nrw = 3;
            }
            tok = cur;
            msk = (f.has_facts && cur >= f.V + f.K) ? 2 : (cur >= f.V ? 1 : 0);
        }
    }
    if (recorder && r0 + g < R && lane == 0) {
        int64_t* o = f.output + row * f.max_len;
        if (!fin) {
            o[i] = cur;
            if (nfin) { f.finished[row] = 1; atomicAdd(f.n_done, 1); }
            else {
                f.hist[row * f.max_len + i] = second;
                if (nrw >= 1) o[i - 1] = rw1;
                if (nrw >= 3) { o[i - 2] = rw2; o[i - 3] = rw3; }
            }
        }
        f.next_token[row] = tok;
        f.next_mask[row] = msk;
        if (f.cap_buf != nullptr && i + 1 < f.max_len) f.cap_buf[row * f.max_len + i + 1] = tok;
        int32_t* so = f.st_out + row * 12;
        so[0] = fin ? po[0] : cur; so[1] = rw1; so[2] = rw2; so[3] = rw3; so[4] = po[3];
        so[5] = (fin || nfin) ? ph[0] : second; so[6] = ph[0]; so[7] = ph[1];
        so[8] = nfin;
    }
    // CaptionEmbedder + sqrt(d) scale + PositionEncoder of the token (geo-aware/models.py:155-181,355-357)
    const int64_t b = row / f.rows_per_sample;
    const float* src;
    if (msk == 1) {
        int e = tok - f.V;
        if (e < 0 || e >= f.K) e = f.K - 1;
        src = f.ee + (b * f.K + e) * d;
    } else if (msk == 2 && f.fe != nullptr) {
        int e = tok - f.V - f.K;
        if (e < 0 || e >= f.F) e = f.F - 1;
        src = f.fe + (b * f.F + e) * d;
    } else {
        src = f.word_emb + (int64_t)(tok >= 0 && tok < f.V ? tok : f.pad_token) * d;
    }
    const float* pe = f.pe + (int64_t)pos * d;
    const int d4 = d >> 2;
    const bool ok0 = lane < d4, ok1 = 64 + lane < d4;
    float4 z0 = f4zero(), z1 = f4zero();
    if (ok0) {
        const float4 e4 = ld4(src + 4 * lane), p4 = ld4(pe + 4 * lane);
        z0 = make_float4(fmaf(e4.x, f.emb_scale, p4.x), fmaf(e4.y, f.emb_scale, p4.y), fmaf(e4.z, f.emb_scale, p4.z),
                         fmaf(e4.w, f.emb_scale, p4.w));
    }
    if (ok1) {
        const float4 e4 = ld4(src + 4 * (64 + lane)), p4 = ld4(pe + 4 * (64 + lane));
        z1 = make_float4(fmaf(e4.x, f.emb_scale, p4.x), fmaf(e4.y, f.emb_scale, p4.y), fmaf(e4.z, f.emb_scale, p4.z),
                         fmaf(e4.w, f.emb_scale, p4.w));
    }
    float* x = xs + g * kDMax;
    reinterpret_cast<float4*>(x)[lane] = z0;
    if (lane < kD4Max - 64) reinterpret_cast<float4*>(x)[64 + lane] = z1;
    if (recorder && r0 + g < R && xa_out != nullptr) {
        if (ok0) reinterpret_cast<float4*>(xa_out + row * d)[lane] = z0;
        if (ok1) reinterpret_cast<float4*>(xa_out + row * d)[64 + lane] = z1;
    }
}

struct SelfArgs {
    LayerW w;
    SelFuse sel;                   // used by the FSEL instantiation (first layer, pos >= 1) only
    float* kc; float* vc;          // (R, H, ML, 32) key / value cache of this layer
    const int32_t* anc;            // optional (R, ML): cache row that holds position p of row r (beam search)
    int R, d, H, dh, ML, pos;
    float scale;
    const int32_t* n_done; int n_total;
};

// ---------------------------------------------------------------------------------------------------------
// self-attention block, one workgroup per (head, group of G <= 8 rows): the head's q|k|v and out_proj weights are
// streamed once for the G rows.  Wave w normalises row w and, later, attends for it.
// ---------------------------------------------------------------------------------------------------------
constexpr int kOPad = 48;          // attention output row in LDS: head width + the zero tail the out-projection reads
template <int G, bool FSEL>
__global__ __launch_bounds__(kNT) void dec_self_kernel(SelfArgs a) {
    static_assert(G <= kNW, "one wave per row");
    __shared__ __attribute__((aligned(16))) float xs[G][kDMax];
    __shared__ uint64_t selkeys[FSEL ? G * kNW * 2 : 2];
    __shared__ __attribute__((aligned(16))) float qkv[G][128];     // q | k | v of this head (3 x dh <= 96), zero tail
    __shared__ __attribute__((aligned(16))) float o[G][kOPad];
    __shared__ __attribute__((aligned(16))) float4 part[G * kNT];
    int h, grp;
    xcd_unit(h, grp, gridDim.y);
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id();
    const int64_t r0 = (int64_t)grp * G;
    const int d = a.d, d4 = d >> 2, dh = a.dh, pos = a.pos, R = a.R;
    const int done = a.n_done != nullptr ? *a.n_done : 0;           // waited for only after the loads are out
    const int64_t myrow = min(r0 + wave, (int64_t)R - 1);
    const bool have_row = wave < G && r0 + wave < R;                // wave-uniform
    ICK_STAMP(0, 0);
    RowGather<G, kNW> in;
    if constexpr (!FSEL) in.issue(a.w.src, r0, R, d);
    RowDot<3> qd;
    qd.load(a.w.in_w, a.w.in_b, d,
            [&](int rr) { const int seg = (rr >= dh) + (rr >= 2 * dh); return seg * (d - dh) + h * dh + rr; }, 3 * dh, d4);
    if (done >= a.n_total) return;                                   // every caption has ended (uniform)
    ICK_STAMP(0, 1);
    for (int idx = tid; idx < G * 32; idx += kNT) qkv[idx >> 5][96 + (idx & 31)] = 0.f;
    for (int idx = tid; idx < G * kOPad; idx += kNT) (&o[0][0])[idx] = 0.f;
    if constexpr (FSEL) {
        // the token of the previous step is chosen and embedded here: the row source is that embedding
        fused_select<G>(a.sel, r0, R, d, pos, &xs[0][0], selkeys, a.w.src.out, h == 0);
    } else {
        in.slice_sums(a.w.src, r0, R, d, part);
        __syncthreads();
        in.finish(a.w.src, r0, R, d, part, &xs[0][0], h == 0);
    }
    __syncthreads();
    ICK_STAMP(0, 2);
    // the cached keys / values of this wave's row and the out-projection slice are requested before the products run
    constexpr int NPC = 4;
    KVRegs<NPC> kv;
    const int p8 = lane >> 3, c = lane & 7;
    const int plast = max(pos - 1, 0);                               // position `pos` itself is not in the cache yet
    // Row offsets of a chunk's positions.  With beam search the cache row of position p is the ancestor's
    // (anc[row][p]): those indices are fetched for the whole chunk in ONE basic block -- as a per-position lambda inside
    // the load loop every index load sat in a branch of its own and was waited for on the spot (load, vmcnt(0), key load,
    // and the same again for the value: eight memory round trips per chunk, ISA of round 5's build).
    uint32_t coff[NPC];
    auto cache_offs = [&](int p0) {
        int pp[NPC];
        int64_t cr[NPC];
#pragma unroll
        for (int q = 0; q < NPC; ++q) pp[q] = min(min(p0 + 8 * q + p8, pos), plast);
        if (a.anc != nullptr) {      // uniform
#pragma unroll
            for (int q = 0; q < NPC; ++q) cr[q] = (int64_t)a.anc[myrow * a.ML + pp[q]];
        } else {
#pragma unroll
            for (int q = 0; q < NPC; ++q) cr[q] = myrow;
        }
#pragma unroll
        for (int q = 0; q < NPC; ++q) coff[q] = (uint32_t)(((cr[q] * a.H + h) * a.ML + pp[q]) * kDhp) * 4u;
    };
    if (have_row) { cache_offs(0); kv.load_at(a.kc, a.vc, coff); }
    ColDot<6> od;
    od.load(a.w.out_wt, d, h * dh, dh, d4);
    qd.template run<G>(&xs[0][0], 3 * dh, &qkv[0][0], 128, 1.f, false);
    __syncthreads();
    ICK_STAMP(0, 3);
    if (have_row) {                                                   // wave-uniform
        const int g = wave;
        // the new key / value row joins the cache (pad columns stay unwritten and are masked by every reader)
        if (lane < dh) a.kc[((myrow * a.H + h) * a.ML + pos) * kDhp + lane] = qkv[g][dh + lane];
        else if (lane >= 32 && lane < 32 + dh) a.vc[((myrow * a.H + h) * a.ML + pos) * kDhp + lane - 32] = qkv[g][2 * dh + lane - 32];
        float4 q4[1], kn, vn;
        const float sc = a.scale;
        q4[0].x = 4 * c + 0 < dh ? qkv[g][4 * c + 0] * sc : 0.f; q4[0].y = 4 * c + 1 < dh ? qkv[g][4 * c + 1] * sc : 0.f;
        q4[0].z = 4 * c + 2 < dh ? qkv[g][4 * c + 2] * sc : 0.f; q4[0].w = 4 * c + 3 < dh ? qkv[g][4 * c + 3] * sc : 0.f;
        kn.x = 4 * c + 0 < dh ? qkv[g][dh + 4 * c + 0] : 0.f; kn.y = 4 * c + 1 < dh ? qkv[g][dh + 4 * c + 1] : 0.f;
        kn.z = 4 * c + 2 < dh ? qkv[g][dh + 4 * c + 2] : 0.f; kn.w = 4 * c + 3 < dh ? qkv[g][dh + 4 * c + 3] : 0.f;
        vn.x = 4 * c + 0 < dh ? qkv[g][2 * dh + 4 * c + 0] : 0.f; vn.y = 4 * c + 1 < dh ? qkv[g][2 * dh + 4 * c + 1] : 0.f;
        vn.z = 4 * c + 2 < dh ? qkv[g][2 * dh + 4 * c + 2] : 0.f; vn.w = 4 * c + 3 < dh ? qkv[g][2 * dh + 4 * c + 3] : 0.f;
        AttnState st[1];
        st[0].init();
        // positions 0 .. pos; loads cover the cached ones (< pos), the new row comes from the registers above
        for (int p0 = 0; p0 <= pos; p0 += 8 * NPC) {
            if (p0 > 0) { cache_offs(p0); kv.load_at(a.kc, a.vc, coff); }
            attend_chunk<NPC, 1>(kv, q4, p0, pos + 1, dh, pos, kn, vn, st);
        }
        attend_reduce(st[0]);
        if (p8 == 0) {
            const float inv = 1.f / st[0].l;
            reinterpret_cast<float4*>(&o[g][0])[c] = make_float4(st[0].acc.x * inv, st[0].acc.y * inv, st[0].acc.z * inv,
                                                                  st[0].acc.w * inv);
        }
    }
    __syncthreads();
    ICK_STAMP(0, 4);
    od.template run<G>(&o[0][0], kOPad, d4, part, a.w.part + (r0 * a.H + h) * d, (int64_t)a.H * d, (int)min((int64_t)G, R - r0));
    ICK_STAMP(0, 5);
}

struct CrossArgs {
    LayerW w;
    const float* Kmem; const float* Vmem;     // (B, ., H, S, 32) segments of the cross K/V buffer of this layer
    int64_t kv_bs;                            // sample stride of that buffer
    int R, rows_per_sample, d, H, dh, S;
    float scale;
    const int32_t* n_done; int n_total;
};

// ---------------------------------------------------------------------------------------------------------
// cross-attention block, one workgroup per (head, group of G rows).  SHARED: the G rows are hypotheses of one caption
// (beam search) -- its K and V are read once for all of them, each of the eight waves taking an eighth of the S memory
// rows; otherwise the waves are dealt out as (row, part of S).  Every wave keeps its part's keys and values in
// registers; the parts of a row meet in LDS as (maximum, sum, weighted values) records.
// ---------------------------------------------------------------------------------------------------------
template <int G, bool SHARED>
__global__ __launch_bounds__(kNT) void dec_cross_kernel(CrossArgs a) {
    constexpr int PARTS = SHARED ? kNW : kNW / G;       // waves per row
    constexpr int NQ = SHARED ? G : 1;                  // query rows per wave
    constexpr int NPC = 7;                              // 56 memory rows per chunk and wave
    static_assert(SHARED || G == 1 || G == 2 || G == 4 || G == 8, "rows of different captions: 1, 2, 4 or 8 per workgroup");
    static_assert(G <= kNW, "one wave per row");
    __shared__ __attribute__((aligned(16))) float xs[G][kDMax];
    __shared__ __attribute__((aligned(16))) float qs[G][32];
    __shared__ __attribute__((aligned(16))) float o[G][kOPad];
    __shared__ __attribute__((aligned(16))) float4 part[G * kNT];
    __shared__ __attribute__((aligned(16))) float4 pacc[G][kNW][8];
    __shared__ float pm[G][kNW], pl[G][kNW];
    int h, grp;
    xcd_unit(h, grp, gridDim.y);
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id();
    const int64_t r0 = (int64_t)grp * G;
    const int d = a.d, d4 = d >> 2, dh = a.dh, S = a.S, R = a.R;
    const int done = a.n_done != nullptr ? *a.n_done : 0;
    ICK_STAMP(1, 0);
    RowGather<G, kNW> in;
    in.issue(a.w.src, r0, R, d);
    ICK_STAMP(1, 1);
    RowDot<1> qd;
    qd.load(a.w.in_w, a.w.in_b, d, [&](int rr) { return h * dh + rr; }, dh, d4);
    ICK_STAMP(1, 2);
    // this wave's part of the memory
    const int g_mine = SHARED ? 0 : wave / PARTS, prt = SHARED ? wave : wave % PARTS;
    const int Sp = ((S + PARTS - 1) / PARTS + 7) & ~7;
    const int lo = prt * Sp, len = max(0, min(S - lo, Sp));
    const int64_t b = min(r0 + g_mine, (int64_t)R - 1) / a.rows_per_sample;
    const float* Kb = a.Kmem + b * a.kv_bs + ((int64_t)h * S + min(lo, S - 1)) * kDhp;
    const float* Vb = a.Vmem + b * a.kv_bs + ((int64_t)h * S + min(lo, S - 1)) * kDhp;
    auto kv_off = [&](int p) { return (uint32_t)p * (kDhp * 4u); };
    KVRegs<NPC> kv;
    kv.template load<true>(Kb, Vb, 0, max(len, 1), kv_off, kv_off);
    ICK_STAMP(1, 3);
    ColDot<6> od;
    od.load(a.w.out_wt, d, h * dh, dh, d4);
    ICK_STAMP(1, 4);
    if (done >= a.n_total) return;
    ICK_STAMP(1, 5);
    in.slice_sums(a.w.src, r0, R, d, part);
    ICK_STAMP(1, 6);
    for (int idx = tid; idx < G * 32; idx += kNT) (&qs[0][0])[idx] = 0.f;
    for (int idx = tid; idx < G * kOPad; idx += kNT) (&o[0][0])[idx] = 0.f;
    __syncthreads();
    ICK_STAMP(1, 7);
    in.finish(a.w.src, r0, R, d, part, &xs[0][0], h == 0);
    ICK_STAMP(1, 8);
    __syncthreads();
    ICK_STAMP(1, 9);
    qd.template run<G>(&xs[0][0], dh, &qs[0][0], 32, a.scale, false);     // q * 1/sqrt(dh), as nn.MultiheadAttention scales it
    ICK_STAMP(1, 10);
    __syncthreads();
    ICK_STAMP(1, 11);
    const int p8 = lane >> 3, c = lane & 7;
    {
        float4 q4[NQ];
        AttnState st[NQ];
#pragma unroll
        for (int n = 0; n < NQ; ++n) {
            q4[n] = reinterpret_cast<const float4*>(&qs[SHARED ? n : g_mine][0])[c];   // pad entries are zero
            st[n].init();
        }
        const float4 none = f4zero();
        for (int p0 = 0; p0 < len; p0 += 8 * NPC) {
            if (p0 > 0) kv.template load<true>(Kb, Vb, p0, len, kv_off, kv_off);
            attend_chunk<NPC, NQ>(kv, q4, p0, len, dh, -1, none, none, st);
        }
#pragma unroll
        for (int n = 0; n < NQ; ++n) {
            attend_reduce(st[n]);
            const int g = SHARED ? n : g_mine;
            if (p8 == 0) pacc[g][prt][c] = st[n].acc;
            if (lane == 0) { pm[g][prt] = st[n].m; pl[g][prt] = st[n].l; }
        }
    }
    ICK_STAMP(1, 12);
    __syncthreads();
    ICK_STAMP(1, 13);
    if (tid < G * 8) {
        const int g = tid >> 3, cc = tid & 7;
        float M = pm[g][0];
#pragma unroll
        for (int w = 1; w < PARTS; ++w) M = fmaxf(M, pm[g][w]);
        float L = 0.f;
        float4 t = f4zero();
#pragma unroll
        for (int w = 0; w < PARTS; ++w) {
            const float f = pm[g][w] == -INFINITY ? 0.f : __expf(pm[g][w] - M);
            L = fmaf(pl[g][w], f, L);
            const float4 u = pacc[g][w][cc];
            t.x = fmaf(u.x, f, t.x); t.y = fmaf(u.y, f, t.y); t.z = fmaf(u.z, f, t.z); t.w = fmaf(u.w, f, t.w);
        }
        const float inv = 1.f / L;
        reinterpret_cast<float4*>(&o[g][0])[cc] = make_float4(t.x * inv, t.y * inv, t.z * inv, t.w * inv);
    }
    __syncthreads();
    ICK_STAMP(1, 14);
    od.template run<G>(&o[0][0], kOPad, d4, part, a.w.part + (r0 * a.H + h) * d, (int64_t)a.H * d, (int)min((int64_t)G, R - r0));
    ICK_STAMP(1, 15);
}

struct FfnArgs {
    const float *w1, *b1;          // (FF, d), (FF)
    const float *w2t;              // (FF, d) TRANSPOSED linear2 weight
    RowSrc src;
    float* part;                   // (R, FF/64, d)
    int R, d, FF;
    const int32_t* n_done; int n_total;
};

// ---------------------------------------------------------------------------------------------------------
// feed-forward block, one workgroup per (64 hidden units, group of G rows)
// ---------------------------------------------------------------------------------------------------------
template <int G>
__global__ __launch_bounds__(kNT) void dec_ffn_kernel(FfnArgs a) {
    static_assert(G <= kNW, "one wave per row");
    __shared__ __attribute__((aligned(16))) float xs[G][kDMax];
    __shared__ __attribute__((aligned(16))) float f[G][96];
    __shared__ __attribute__((aligned(16))) float4 part[G * kNT];
    int ch, grp;
    xcd_unit(ch, grp, gridDim.y);
    const int tid = threadIdx.x, nch = gridDim.x;
    const int64_t r0 = (int64_t)grp * G;
    const int d = a.d, d4 = d >> 2, R = a.R;
    const int j0 = ch * 64, nj = min(64, a.FF - j0);
    const int done = a.n_done != nullptr ? *a.n_done : 0;
    ICK_STAMP(2, 0);
    RowGather<G, kNW> in;
    in.issue(a.src, r0, R, d);
    RowDot<2> fd;
    fd.load(a.w1, a.b1, d, [&](int rr) { return j0 + rr; }, nj, d4);
    ColDot<11> od;
    od.load(a.w2t, d, j0, nj, d4);
    if (done >= a.n_total) return;
    ICK_STAMP(2, 1);
    in.slice_sums(a.src, r0, R, d, part);
    for (int idx = tid; idx < G * 96; idx += kNT) (&f[0][0])[idx] = 0.f;
    __syncthreads();
    in.finish(a.src, r0, R, d, part, &xs[0][0], ch == 0);
    __syncthreads();
    ICK_STAMP(2, 2);
    fd.template run<G>(&xs[0][0], nj, &f[0][0], 96, 1.f, true);
    __syncthreads();
    ICK_STAMP(2, 3);
    od.template run<G>(&f[0][0], 96, d4, part, a.part + (r0 * nch + ch) * d, (int64_t)nch * d, (int)min((int64_t)G, R - r0));
    ICK_STAMP(2, 4);
}

struct HeadArgs {
    RowSrc src;                    // final LayerNorm of the last layer; out = h (R, d)
    const float* gate;             // optional (R, d): fc_predicate(predicate indicator)
    float* hv;                     // (R, d) h * gate (== h when gate is null)
    const float *ee, *we, *be;     // entities_encoded (B, K, d), fc_entity
    const float *fe, *wf, *bf;     // facts_encoded (B, F, d), fc_fact (null for geo)
    const float* eib;              // (R, F) indicator
    float* ptr;                    // (R, K + F) pointer scores
    int R, rows_per_sample, d, K, F;
    const int32_t* n_done; int n_total;
};

// One row of the score head: final LayerNorm -> h (x gate) -> pointer scores.  Written for four waves; in a wider workgroup
// (the merged head + vocabulary launch) the waves beyond the fourth only keep the barriers company.
__device__ __forceinline__ void head_row(const HeadArgs& a, const int64_t r, float* xs, float4* psum) {
    const int tid = threadIdx.x;
    const int d = a.d, d4 = d >> 2;
    const int64_t b = r / a.rows_per_sample;
    const int lane = tid & 63, wave = wave_id(), sub = lane >> 4, i = lane & 15;
    const int done = a.n_done != nullptr ? *a.n_done : 0;
    RowGather<1, 4> in;
    in.issue(a.src, r, a.R, d);
    const float* wq[2] = {a.we, a.fe != nullptr && a.F > 0 ? a.wf : a.we};
    float4 wr2[2][5];
#pragma unroll
    for (int pp = 0; pp < 2; ++pp)
#pragma unroll
        for (int it = 0; it < 5; ++it) wr2[pp][it] = ld4(wq[pp] + 4 * min(i + 16 * it, d4 - 1));
    const float bias2[2] = {a.be[0], a.fe != nullptr && a.F > 0 ? a.bf[0] : 0.f};
    if (done >= a.n_total) return;
    in.slice_sums(a.src, r, a.R, d, psum);
    // the first context rows are requested while wave 0 normalises
    const float* ctx0 = a.ee;
    const int n0 = a.K;
    float4 cv0[5];
    {
        const float* cr = ctx0 + (b * n0 + min(4 * wave + sub, n0 - 1)) * d;
#pragma unroll
        for (int it = 0; it < 5; ++it) cv0[it] = ld4(cr + 4 * min(i + 16 * it, d4 - 1));
    }
    __syncthreads();
    in.finish(a.src, r, a.R, d, psum, xs, true);
    __syncthreads();
    if (tid < 256)
        for (int c = tid; c < d; c += 256) a.hv[r * d + c] = a.gate ? xs[c] * a.gate[r * d + c] : xs[c];
    if (wave >= 4) return;      // no barrier below
    float4 xr[5];
#pragma unroll
    for (int it = 0; it < 5; ++it) xr[it] = reinterpret_cast<const float4*>(xs)[i + 16 * it];      // zero beyond d
    for (int part = 0; part < 2; ++part) {
        const float* ctx = part == 0 ? a.ee : a.fe;
        const int n = part == 0 ? a.K : a.F;
        if (ctx == nullptr || n <= 0) continue;
        float4 wr[5];
#pragma unroll
        for (int it = 0; it < 5; ++it) wr[it] = wr2[part][it];
        const float bias = bias2[part];
        for (int k0 = 0; k0 < n; k0 += 16) {
            const int k = k0 + 4 * wave + sub;
            float4 cv[5];
            if (part == 0 && k0 == 0) {
#pragma unroll
                for (int it = 0; it < 5; ++it) cv[it] = cv0[it];
            } else {
                const float* cr = ctx + (b * n + min(k, n - 1)) * d;
#pragma unroll
                for (int it = 0; it < 5; ++it) cv[it] = ld4(cr + 4 * min(i + 16 * it, d4 - 1));
            }
            float acc = 0.f;
#pragma unroll
            for (int it = 0; it < 5; ++it) {
                // (h * ctx) * w summed over d, the order of the reference's broadcast product
                acc = fmaf(__fmul_rn(xr[it].x, cv[it].x), wr[it].x, acc);
                acc = fmaf(__fmul_rn(xr[it].y, cv[it].y), wr[it].y, acc);
                acc = fmaf(__fmul_rn(xr[it].z, cv[it].z), wr[it].z, acc);
                acc = fmaf(__fmul_rn(xr[it].w, cv[it].w), wr[it].w, acc);
            }
            acc = sum16(acc);
            if (i == 0 && k < n) {
                const float ind = (part == 1 && a.eib) ? a.eib[r * a.F + k] : 1.f;
                a.ptr[r * (a.K + a.F) + (part == 0 ? 0 : a.K) + k] = acc * ind + bias;
            }
        }
    }
}

__global__ __launch_bounds__(256) void dec_head_kernel(HeadArgs a) {
    __shared__ __attribute__((aligned(16))) float xs[kDMax];
    __shared__ __attribute__((aligned(16))) float4 psum[4 * kD4Max];
    head_row(a, blockIdx.x, xs, psum);
}

struct VocabArgs {
    const float* hv;               // (R, d)
    const float *wv, *bv;          // (V, d), (V)
    float* scores; int64_t ld;     // optional (R, ld) logits
    float4* cand;                  // (R, ntiles) {best value, best index, second value, second index}
    int R, d, V, ntiles;
    const int32_t* n_done; int n_total;
};

struct Top2 {
    float v1, v2;
    int i1, i2;
};
// branch-free: the merges run in every lane of a wave (nested ifs compile to exec-mask branches, ~20 per merge)
__device__ __forceinline__ void top2_push(Top2& s, float v, int i) {
    const bool b1 = v > s.v1 || (v == s.v1 && i < s.i1);
    const bool b2 = v > s.v2 || (v == s.v2 && i < s.i2);
    const float nv2 = b1 ? s.v1 : (b2 ? v : s.v2);
    const int ni2 = b1 ? s.i1 : (b2 ? i : s.i2);
    s.v1 = b1 ? v : s.v1;
    s.i1 = b1 ? i : s.i1;
    s.v2 = nv2;
    s.i2 = ni2;
}
constexpr int kNone = 0x7fffffff;
template <int CTRL>
__device__ __forceinline__ Top2 top2_merge_dpp(Top2 t) {
    const float v1 = dppf<CTRL>(t.v1), v2 = dppf<CTRL>(t.v2);
    const int i1 = dppi<CTRL>(t.i1), i2 = dppi<CTRL>(t.i2);
    top2_push(t, v1, i1);
    top2_push(t, v2, i2);
    return t;
}

// ---------------------------------------------------------------------------------------------------------
// vocabulary logits: workgroup = kVocabTile words x all rows (blocks of 32), the eight waves split K; fp32 MFMA
// 16x16x4.  One workgroup per CU at V = 10 000 (209 of them): its 58 KB weight slice is fetched once and stays in
// registers for every block of 32 rows (beam search decodes captions x beams rows).
// ---------------------------------------------------------------------------------------------------------
constexpr int kVocabTile = 48;
constexpr int kVocabCT = kVocabTile / 16;     // 16-word column tiles per workgroup
constexpr int kVocabKC = 3;                   // 16-wide k chunks per wave: 8 x 3 x 16 = 384 >= kDMax
__global__ __launch_bounds__(kNT) void dec_vocab_kernel(VocabArgs a) {
    __shared__ __attribute__((aligned(16))) float red[kNW][2][kVocabCT][16][17];
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id();
    const int fi = lane & 15, fq = lane >> 4;
    const int n0 = blockIdx.x * kVocabTile;
    const int d = a.d;
    const int done = a.n_done != nullptr ? *a.n_done : 0;
    ICK_STAMP(3, 0);
    // this wave's K slice of the 48 weight rows
    float4 bw[kVocabCT][kVocabKC];
    uint32_t koff[kVocabKC];
    bool kok[kVocabKC];
#pragma unroll
    for (int t = 0; t < kVocabKC; ++t) {
        const int k = 16 * (kVocabKC * wave + t) + 4 * fq;
        kok[t] = k < d;                              // d % 4 == 0: a float4 is entirely inside or outside
        koff[t] = kok[t] ? 4u * (uint32_t)k : 0u;
    }
#pragma unroll
    for (int ct = 0; ct < kVocabCT; ++ct) {
        const uint32_t rb = (uint32_t)min(n0 + 16 * ct + fi, a.V - 1) * (uint32_t)d * 4u;
#pragma unroll
        for (int t = 0; t < kVocabKC; ++t) {
            bw[ct][t] = ld4o(a.wv, rb + koff[t]);
            if (!kok[t]) bw[ct][t] = f4zero();
        }
    }
    const int row = tid >> 4, cp = tid & 15;         // epilogue role: one row, three columns
    constexpr int kCols = kVocabTile / 16;
    float biasv[kCols];
#pragma unroll
    for (int e = 0; e < kCols; ++e) biasv[e] = a.bv[min(n0 + kCols * cp + e, a.V - 1)];
    if (done >= a.n_total) return;
    for (int m0 = 0; m0 < a.R; m0 += 32) {
        const uint32_t ra0 = (uint32_t)min(m0 + fi, a.R - 1) * (uint32_t)d * 4u,
                       ra1 = (uint32_t)min(m0 + 16 + fi, a.R - 1) * (uint32_t)d * 4u;
        float4 av0[kVocabKC], av1[kVocabKC];
#pragma unroll
        for (int t = 0; t < kVocabKC; ++t) {
            av0[t] = ld4o(a.hv, ra0 + koff[t]);
            av1[t] = ld4o(a.hv, ra1 + koff[t]);
            if (!kok[t]) av0[t] = av1[t] = f4zero();
        }
        ICK_STAMP(3, 1);
#pragma unroll
        for (int ct = 0; ct < kVocabCT; ++ct) {
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < kVocabKC; ++t) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av0[t].x, bw[ct][t].x, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av1[t].x, bw[ct][t].x, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av0[t].y, bw[ct][t].y, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av1[t].y, bw[ct][t].y, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av0[t].z, bw[ct][t].z, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av1[t].z, bw[ct][t].z, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av0[t].w, bw[ct][t].w, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av1[t].w, bw[ct][t].w, acc1, 0, 0, 0);
            }
            // C/D map: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                red[wave][0][ct][fq * 4 + rg][fi] = acc0[rg];
                red[wave][1][ct][fq * 4 + rg][fi] = acc1[rg];
            }
        }
        ICK_STAMP(3, 2);
        __syncthreads();
        ICK_STAMP(3, 3);
        // thread (row = tid >> 4, three columns): sum the eight K slices in a fixed order, add the bias
        const int gr = m0 + row;
        Top2 t2{-INFINITY, -INFINITY, kNone, kNone};
#pragma unroll
        for (int e = 0; e < kCols; ++e) {
            const int col = kCols * cp + e, n = n0 + col, ct = col >> 4, cc = col & 15;
            float v = red[0][row >> 4][ct][row & 15][cc];
#pragma unroll
            for (int w = 1; w < kNW; ++w) v += red[w][row >> 4][ct][row & 15][cc];
            if (n < a.V) {
                v += biasv[e];
                if (a.scores != nullptr && gr < a.R) a.scores[(int64_t)gr * a.ld + n] = v;
                top2_push(t2, v, n);
            }
        }
        t2 = top2_merge_dpp<kDppXor1>(t2);      // empty slots carry (-inf, kNone): they never displace anything
        t2 = top2_merge_dpp<kDppXor2>(t2);
        t2 = top2_merge_dpp<kDppHalfMirror>(t2);
        t2 = top2_merge_dpp<kDppMirror>(t2);
        if (cp == 0 && gr < a.R)
            a.cand[(int64_t)gr * a.ntiles + blockIdx.x] =
                make_float4(t2.v1, __int_as_float(t2.i1), t2.v2, __int_as_float(t2.i2));
        ICK_STAMP(3, 4);
        if (m0 + 32 < a.R) __syncthreads();       // the exchange buffer is reused by the next block of rows
    }
}

// ---------------------------------------------------------------------------------------------------------
// Score head + vocabulary logits in ONE launch for R <= 32 rows (greedy decoding): the vocabulary
// workgroups no longer wait for a launch that writes h -- each normalises the R rows itself (LN-on-load into LDS, 16
// lanes per row, four source rows' loads in flight at once: three L2 round trips, hidden behind the fetch of the
// workgroup's 58 KB weight slice from HBM), and R more workgroups of the same launch do what dec_head_kernel does
// (h, h x gate, pointer scores).  No workgroup depends on another.  One dependent launch less per token.
// The normalised rows are summed source by source in the order res, bias, partial 0, 1, ...: deterministic, the same in
// every vocabulary workgroup.
// ---------------------------------------------------------------------------------------------------------
constexpr int kHvSrc = 12;                    // residual + bias + up to 10 partial rows (heads / FFN chunks)
constexpr int kHvChunk = 4;                   // source rows in flight per lane (x 5 float4)
constexpr int kHvLd = kDMax + 4;              // LDS row stride of the normalised rows
__global__ __launch_bounds__(kNT) void dec_headvocab_kernel(HeadArgs ha, VocabArgs a) {
    __shared__ __attribute__((aligned(16))) float red[kNW][2][kVocabCT][16][17];
    __shared__ __attribute__((aligned(16))) float xa[32 * kHvLd];
    __shared__ __attribute__((aligned(16))) float4 psum[4 * kD4Max];
    if ((int)blockIdx.x >= a.ntiles) {          // head role (uniform per workgroup)
        head_row(ha, (int64_t)blockIdx.x - a.ntiles, xa, psum);
        return;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id();
    const int fi = lane & 15, fq = lane >> 4;
    const int n0 = blockIdx.x * kVocabTile;
    const int d = a.d, d4 = d >> 2;
    const int done = a.n_done != nullptr ? *a.n_done : 0;
    // this wave's K slice of the 48 weight rows (HBM: the longest wait of the kernel, requested first)
    float4 bw[kVocabCT][kVocabKC];
    uint32_t koff[kVocabKC];
    bool kok[kVocabKC];
#pragma unroll
    for (int t = 0; t < kVocabKC; ++t) {
        const int k = 16 * (kVocabKC * wave + t) + 4 * fq;
        kok[t] = k < d;
        koff[t] = kok[t] ? 4u * (uint32_t)k : 0u;
    }
#pragma unroll
    for (int ct = 0; ct < kVocabCT; ++ct) {
        const uint32_t rb = (uint32_t)min(n0 + 16 * ct + fi, a.V - 1) * (uint32_t)d * 4u;
#pragma unroll
        for (int t = 0; t < kVocabKC; ++t) {
            bw[ct][t] = ld4o(a.wv, rb + koff[t]);
            if (!kok[t]) bw[ct][t] = f4zero();
        }
    }
    const int row = tid >> 4, cp = tid & 15;         // LN role and epilogue role: one row per 16 lanes
    constexpr int kCols = kVocabTile / 16;
    float biasv[kCols];
#pragma unroll
    for (int e = 0; e < kCols; ++e) biasv[e] = a.bv[min(n0 + kCols * cp + e, a.V - 1)];
    if (done >= a.n_total) return;
    // ---- x[row] = LayerNorm(res + bias + sum of the partial rows) (x gate): lane cp holds float4 columns cp + 16 it
    {
        const RowSrc& s = ha.src;
        const int64_t r = min(row, a.R - 1);
        const int nrows = s.nparts > 0 ? s.nparts + 2 : 1;
        uint32_t col[5];
#pragma unroll
        for (int it = 0; it < 5; ++it) col[it] = 16u * (uint32_t)min(cp + 16 * it, d4 - 1);
        float4 z[5];
#pragma unroll
        for (int it = 0; it < 5; ++it) z[it] = f4zero();
#pragma unroll 1
        for (int half = 0; half < kHvSrc / kHvChunk; ++half) {
            float4 v[kHvChunk][5];
#pragma unroll
            for (int j = 0; j < kHvChunk; ++j) {
                const int q = min(half * kHvChunk + j, nrows - 1);
                const float* src = q == 0 ? s.res + r * d : (q == 1 ? s.bias : s.part + (r * s.nparts + (q - 2)) * d);
#pragma unroll
                for (int it = 0; it < 5; ++it) v[j][it] = ld4o(src, col[it]);
            }
#pragma unroll
            for (int j = 0; j < kHvChunk; ++j) {
                const float f = half * kHvChunk + j < nrows ? 1.f : 0.f;
#pragma unroll
                for (int it = 0; it < 5; ++it) {
                    z[it].x = fmaf(v[j][it].x, f, z[it].x); z[it].y = fmaf(v[j][it].y, f, z[it].y);
                    z[it].z = fmaf(v[j][it].z, f, z[it].z); z[it].w = fmaf(v[j][it].w, f, z[it].w);
                }
            }
            __builtin_amdgcn_sched_barrier(0);     // the next chunk's loads stay behind this chunk's sums (registers)
        }
        bool ok[5];
#pragma unroll
        for (int it = 0; it < 5; ++it) {
            ok[it] = cp + 16 * it < d4;
            if (!ok[it]) z[it] = f4zero();
        }
        if (s.nparts > 0) {
            float4 gm[5], bt[5];
#pragma unroll
            for (int it = 0; it < 5; ++it) { gm[it] = ld4o(s.gamma, col[it]); bt[it] = ld4o(s.beta, col[it]); }
            float sum = 0.f;
#pragma unroll
            for (int it = 0; it < 5; ++it) sum += (z[it].x + z[it].y) + (z[it].z + z[it].w);
            const float mean = sum16(sum) / (float)d;
            float var = 0.f;
#pragma unroll
            for (int it = 0; it < 5; ++it)
                if (ok[it]) {
                    var = fmaf(z[it].x - mean, z[it].x - mean, var); var = fmaf(z[it].y - mean, z[it].y - mean, var);
                    var = fmaf(z[it].z - mean, z[it].z - mean, var); var = fmaf(z[it].w - mean, z[it].w - mean, var);
                }
            const float rstd = rsqrtf(sum16(var) / (float)d + s.eps);
#pragma unroll
            for (int it = 0; it < 5; ++it) {
                z[it].x = (z[it].x - mean) * rstd * gm[it].x + bt[it].x; z[it].y = (z[it].y - mean) * rstd * gm[it].y + bt[it].y;
                z[it].z = (z[it].z - mean) * rstd * gm[it].z + bt[it].z; z[it].w = (z[it].w - mean) * rstd * gm[it].w + bt[it].w;
            }
        }
        if (ha.gate != nullptr) {
#pragma unroll
            for (int it = 0; it < 5; ++it) {
                const float4 g = ld4o(ha.gate + r * d, col[it]);
                z[it].x *= g.x; z[it].y *= g.y; z[it].z *= g.z; z[it].w *= g.w;
            }
        }
#pragma unroll
        for (int it = 0; it < 5; ++it)
            reinterpret_cast<float4*>(xa + row * kHvLd)[cp + 16 * it] = ok[it] ? z[it] : f4zero();
    }
    __syncthreads();
    float4 av0[kVocabKC], av1[kVocabKC];
#pragma unroll
    for (int t = 0; t < kVocabKC; ++t) {
        av0[t] = kok[t] ? *reinterpret_cast<const float4*>(xa + fi * kHvLd + (koff[t] >> 2)) : f4zero();
        av1[t] = kok[t] ? *reinterpret_cast<const float4*>(xa + (16 + fi) * kHvLd + (koff[t] >> 2)) : f4zero();
    }
#pragma unroll
    for (int ct = 0; ct < kVocabCT; ++ct) {
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < kVocabKC; ++t) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av0[t].x, bw[ct][t].x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av1[t].x, bw[ct][t].x, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av0[t].y, bw[ct][t].y, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av1[t].y, bw[ct][t].y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av0[t].z, bw[ct][t].z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av1[t].z, bw[ct][t].z, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av0[t].w, bw[ct][t].w, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av1[t].w, bw[ct][t].w, acc1, 0, 0, 0);
        }
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            red[wave][0][ct][fq * 4 + rg][fi] = acc0[rg];
            red[wave][1][ct][fq * 4 + rg][fi] = acc1[rg];
        }
    }
    __syncthreads();
    // thread (row = tid >> 4, three columns): sum the eight K slices in a fixed order, add the bias
    Top2 t2{-INFINITY, -INFINITY, kNone, kNone};
#pragma unroll
    for (int e = 0; e < kCols; ++e) {
        const int colv = kCols * cp + e, n = n0 + colv, ct = colv >> 4, cc = colv & 15;
        float v = red[0][row >> 4][ct][row & 15][cc];
#pragma unroll
        for (int w = 1; w < kNW; ++w) v += red[w][row >> 4][ct][row & 15][cc];
        if (n < a.V) {
            v += biasv[e];
            if (a.scores != nullptr && row < a.R) a.scores[(int64_t)row * a.ld + n] = v;
            top2_push(t2, v, n);
        }
    }
    t2 = top2_merge_dpp<kDppXor1>(t2);
    t2 = top2_merge_dpp<kDppXor2>(t2);
    t2 = top2_merge_dpp<kDppHalfMirror>(t2);
    t2 = top2_merge_dpp<kDppMirror>(t2);
    if (cp == 0 && row < a.R)
        a.cand[(int64_t)row * a.ntiles + blockIdx.x] =
            make_float4(t2.v1, __int_as_float(t2.i1), t2.v2, __int_as_float(t2.i2));
}

struct SelectArgs {
    const float4* cand; int ntiles;
    const float* ptr;              // (R, K + F)
    int64_t* output;               // (R, max_len)
    int32_t* hist;                 // (R, max_len) runner-up of every step
    int32_t* finished;             // (R)
    int32_t* n_done;               // number of finished rows (early exit of the following steps)
    int64_t* next_token; int64_t* next_mask;   // (R)
    int64_t* cap_buf;              // optional (R, max_len): the caption buffer get_context_indicators reads
    // embedding of the next token
    const float *word_emb, *ee, *fe, *pe;
    float* x0;                     // (R, d)
    int R, rows_per_sample, d, V, K, F, step, max_len, has_facts, end_token, pad_token;
    float emb_scale;
    int n_total;
};

__device__ __forceinline__ Top2 top2_merge_wave(Top2 t) {
    t = top2_merge_dpp<kDppXor1>(t);
    t = top2_merge_dpp<kDppXor2>(t);
    t = top2_merge_dpp<kDppHalfMirror>(t);
    t = top2_merge_dpp<kDppMirror>(t);
#pragma unroll
    for (int off = 16; off < 64; off <<= 1) {
        Top2 o;
        o.v1 = __shfl_xor(t.v1, off, 64); o.i1 = __shfl_xor(t.i1, off, 64);
        o.v2 = __shfl_xor(t.v2, off, 64); o.i2 = __shfl_xor(t.i2, off, 64);
        top2_push(t, o.v1, o.i1);
        top2_push(t, o.v2, o.i2);
    }
    return t;
}

// Selection + predict()'s per-step bookkeeping for one caption (geo-aware/models.py:410-441) + embedding of the
// next input token.  Thread 0 fetches the caption's recent history while the workgroup scans the candidates, so the
// n-gram clean-up runs on registers: the kernel is two memory round trips long (candidates, embedding row).
__global__ __launch_bounds__(256) void dec_select_kernel(SelectArgs a) {
    if (*a.n_done >= a.n_total) return;
    __shared__ Top2 sh[4];
    __shared__ int64_t tok_sh[2];
    const int tid = threadIdx.x, i = a.step;
    const int64_t r = blockIdx.x;
    int64_t* o = a.output + r * a.max_len;
    int32_t* hs = a.hist + r * a.max_len;
    int64_t po[5] = {-1, -1, -1, -1, -1};     // output[i-1 .. i-5]
    int ph[3] = {0, 0, 0};                    // runner-ups of steps i-1 .. i-3
    int fin = 0;
    if (tid == 0) {
        fin = a.finished[r];
#pragma unroll
        for (int q = 0; q < 5; ++q) po[q] = o[max(i - 1 - q, 0)];
#pragma unroll
        for (int q = 0; q < 3; ++q) ph[q] = hs[max(i - 1 - q, 0)];
    }
    Top2 s{-INFINITY, -INFINITY, kNone, kNone};
    constexpr int NC = 4;                      // 1024 candidate tiles (16 384 words) per sweep
    for (int t0 = 0; t0 < a.ntiles; t0 += 256 * NC) {
        float4 cd[NC];
#pragma unroll
        for (int q = 0; q < NC; ++q) cd[q] = a.cand[r * a.ntiles + min(t0 + tid + 256 * q, a.ntiles - 1)];
#pragma unroll
        for (int q = 0; q < NC; ++q) {
            if (t0 + tid + 256 * q >= a.ntiles) continue;
            const int i1 = __float_as_int(cd[q].y), i2 = __float_as_int(cd[q].w);
            top2_push(s, cd[q].x, i1);
            top2_push(s, cd[q].z, i2);
        }
    }
    const int np = a.K + a.F;
    for (int k = tid; k < np; k += 256) top2_push(s, a.ptr[r * np + k], a.V + k);
    s = top2_merge_wave(s);
    if ((tid & 63) == 0) sh[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) {
        Top2 t = sh[0];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            top2_push(t, sh[w].v1, sh[w].i1);
            top2_push(t, sh[w].v2, sh[w].i2);
        }
        const int best = t.i1, second = t.i2 == kNone ? t.i1 : t.i2;
        int64_t tok = 0, msk = 0;
        if (!fin) {
            if (best == a.end_token) {
                o[i] = best;
                a.finished[r] = 1;
                atomicAdd(a.n_done, 1);
            } else {
                hs[i] = second;
                int64_t cur = best;
                // repeated n-gram clean-up (geo-aware/models.py:421-435) on the registers
                if (i > 0 && cur == po[0]) {
                    cur = second;
                } else if (i > 2 && cur == po[1] && po[0] == po[2]) {
                    cur = second;
                    o[i - 1] = ph[0];
                } else if (i > 4 && cur == po[2] && po[0] == po[3] && po[1] == po[4]) {
                    cur = second;
                    o[i - 1] = ph[0]; o[i - 2] = ph[1]; o[i - 3] = ph[2];
                }
                o[i] = cur;
                tok = cur;
                msk = (a.has_facts && cur >= a.V + a.K) ? 2 : (cur >= a.V ? 1 : 0);
            }
        }
        a.next_token[r] = tok;
        a.next_mask[r] = msk;
        tok_sh[0] = tok;
        tok_sh[1] = msk;
        if (a.cap_buf != nullptr && i + 1 < a.max_len) a.cap_buf[r * a.max_len + i + 1] = tok;
    }
    __syncthreads();
    if (i + 1 >= a.max_len) return;
    // CaptionEmbedder + sqrt(d) scale + PositionEncoder of the next input token (geo-aware/models.py:155-181,355-357)
    const int64_t tok = tok_sh[0], msk = tok_sh[1];
    const int64_t b = r / a.rows_per_sample;
    const float* src;
    if (msk == 1) {
        int64_t e = tok - a.V;
        if (e < 0 || e >= a.K) e = a.K - 1;
        src = a.ee + (b * a.K + e) * a.d;
    } else if (msk == 2 && a.fe != nullptr) {
        int64_t e = tok - a.V - a.K;
        if (e < 0 || e >= a.F) e = a.F - 1;
        src = a.fe + (b * a.F + e) * a.d;
    } else {
        src = a.word_emb + (tok >= 0 && tok < a.V ? tok : (int64_t)a.pad_token) * a.d;
    }
    const float* pe = a.pe + (int64_t)(i + 1) * a.d;
    for (int c = tid; c < a.d; c += 256) a.x0[r * a.d + c] = fmaf(src[c], a.emb_scale, pe[c]);
}

// ---------------------------------------------------------------------------------------------------------
// beam selection: one workgroup per caption; its k hypotheses are rows b*k .. b*k+k-1
// ---------------------------------------------------------------------------------------------------------
constexpr int kBeamMax = 8;
struct BeamArgs {
    const float* rec;                   // (R, nchunk, kBeamRec) chunk records of dec_beam_partial_kernel
    float* cum;                         // (R) cumulative log-probability of every hypothesis
    int32_t* fin;                       // (R) hypothesis has produced <end>
    const int64_t* seq_in; int64_t* seq_out;     // (R, max_len) tokens so far
    const int32_t* anc_in; int32_t* anc_out;     // (R, max_len) cache row of every position
    const int64_t* cap_in; int64_t* cap_out;     // optional (R, max_len) caption buffers (<start> + tokens)
    int32_t* n_done;
    int64_t* next_token; int64_t* next_mask;
    const float *word_emb, *ee, *fe, *pe;
    float* x0;
    int R, k, d, V, K, F, step, max_len, has_facts, end_token, pad_token, start_token;
    float emb_scale;
    int n_total;
};

// Block-wide arg-best of (value, code) pairs: larger value wins, ties go to the smaller code.  Result in every thread.
__device__ __forceinline__ void block_best(float& v, int& c, float* shv, int* shc) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float ov = __shfl_xor(v, off, 64);
        const int oc = __shfl_xor(c, off, 64);
        const bool take = ov > v || (ov == v && oc < c);
        v = take ? ov : v;
        c = take ? oc : c;
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { shv[threadIdx.x >> 6] = v; shc[threadIdx.x >> 6] = c; }
    __syncthreads();
    v = shv[0]; c = shc[0];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
        const bool take = shv[w] > v || (shv[w] == v && shc[w] < c);
        v = take ? shv[w] : v;
        c = take ? shc[w] : c;
    }
}

constexpr int kBeamChunk = 1024;      // scores per stage-1 workgroup
constexpr int kBeamCandPerThread = 16; // stage 2 holds beam^2 x chunks candidates in the registers of 256 threads
constexpr int kBeamRec = 2 + 2 * kBeamMax;   // floats per (row, chunk) record: max, sum of exp, kBeamMax x (value, index)

// Stage 1 of the beam selection, one workgroup per (chunk of 1024 scores, row): the chunk's maximum, its sum of
// exp(score - maximum), and its k best (value, index) pairs -- 5 x 10 k scores per caption are reduced by 50
// workgroups instead of one.
struct BeamPartArgs {
    const float* scores; int64_t ld; const float* ptr;
    const float* cum; const int32_t* fin;
    float* rec;                         // (R, nchunk, kBeamRec)
    int R, k, V, np, nchunk;
    const int32_t* n_done; int n_total;
};
__global__ __launch_bounds__(256) void dec_beam_partial_kernel(BeamPartArgs a) {
    if (*a.n_done >= a.n_total) return;
    __shared__ float shv[4];
    __shared__ int shc[4];
    __shared__ float red[8];
    const int tid = threadIdx.x, ch = blockIdx.x;
    const int64_t r = blockIdx.y;
    if (a.fin[r] || a.cum[r] == -INFINITY) return;         // ended / unused hypothesis: nothing to expand (uniform)
    const int Vx = a.V + a.np;
    const float* row = a.scores + r * a.ld;
    const float* pr = a.ptr + r * a.np;
    float x[4];
    int idx[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        idx[q] = ch * kBeamChunk + tid + 256 * q;
        const int v = min(idx[q], Vx - 1);
        x[q] = v < a.V ? row[v] : pr[v - a.V];
        if (idx[q] >= Vx) { x[q] = -INFINITY; idx[q] = kNone; }
    }
    float m = fmaxf(fmaxf(x[0], x[1]), fmaxf(x[2], x[3]));
    m = block_max<4>(m, red);
    float e = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) e += x[q] == -INFINITY ? 0.f : __expf(x[q] - m);
    e = block_sum<4>(e, red);
    float* rec = a.rec + (r * a.nchunk + ch) * kBeamRec;
    if (tid == 0) { rec[0] = m; rec[1] = e; }
    for (int round = 0; round < a.k; ++round) {
        float bv = -INFINITY; int bc = kNone;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool take = x[q] > bv || (x[q] == bv && idx[q] < bc);
            bv = take ? x[q] : bv;
            bc = take ? idx[q] : bc;
        }
        block_best(bv, bc, shv, shc);
        if (tid == 0) { rec[2 + 2 * round] = bv; rec[3 + 2 * round] = __int_as_float(bc); }
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (idx[q] == bc) { x[q] = -INFINITY; idx[q] = kNone; }      // the winner leaves the pool
    }
}

__global__ __launch_bounds__(256) void dec_select_beam_kernel(BeamArgs a) {
    // No exit on *n_done here: other workgroups of this very launch add to it, and the hypothesis tables are
    // ping-pong buffers -- a step that skipped its carry-copy would leave the previous step's rows (in another
    // order) in the buffer the caller reads.  The decision below only looks at this caption's own state, which no
    // other workgroup writes.
    __shared__ float shv[4];
    __shared__ int shc[4];
    __shared__ float cum_s[kBeamMax], lse_s[kBeamMax];
    __shared__ int fin_s[kBeamMax], parent_s[kBeamMax], tok_s[kBeamMax], nfin_s[kBeamMax];
    const int tid = threadIdx.x, k = a.k;
    const int64_t r0 = (int64_t)blockIdx.x * k;
    const int np = a.K + a.F, Vx = a.V + np;
    const int nchunk = (Vx + kBeamChunk - 1) / kBeamChunk;
    if (tid < k) { cum_s[tid] = a.cum[r0 + tid]; fin_s[tid] = a.fin[r0 + tid]; }
    __syncthreads();
    bool live_any = false;
    for (int j = 0; j < k; ++j) live_any = live_any || !(fin_s[j] || cum_s[j] == -INFINITY);
    if (!live_any) {
        // every hypothesis of this caption has ended: the tables move to the other buffer as they are
        for (int idx = tid; idx < k * a.max_len; idx += 256) {
            const int64_t e = r0 * a.max_len + idx;
            a.seq_out[e] = a.seq_in[e];
            a.anc_out[e] = a.anc_in[e];
            if (a.cap_out != nullptr) a.cap_out[e] = a.cap_in[e];
        }
        if (tid < k) { a.next_token[r0 + tid] = 0; a.next_mask[r0 + tid] = 0; }
        return;
    }
    // log-sum-exp of every live row from its chunk records (wave j handles row j, j + 4)
    for (int j = tid >> 6; j < k; j += 4) {
        if (fin_s[j] || cum_s[j] == -INFINITY) continue;
        const int lane = tid & 63;
        float m = -INFINITY;
        for (int c = lane; c < nchunk; c += 64) m = fmaxf(m, a.rec[((r0 + j) * nchunk + c) * kBeamRec]);
        m = wave_max(m);
        float e = 0.f;
        for (int c = lane; c < nchunk; c += 64) {
            const float* rc = a.rec + ((r0 + j) * nchunk + c) * kBeamRec;
            e += rc[1] * __expf(rc[0] - m);
        }
        e = wave_sum(e);
        if (lane == 0) lse_s[j] = m + __logf(e);
    }
    __syncthreads();
    // candidates: k per (live row, chunk), one per ended row; every thread keeps up to NC of them
    constexpr int NC = kBeamCandPerThread;
    float cv[NC]; int cc[NC];
    const int per_row = nchunk * k, total = k * per_row;
#pragma unroll
    for (int q = 0; q < NC; ++q) {
        cv[q] = -INFINITY; cc[q] = kNone;
        const int id = tid + 256 * q;
        if (id < total) {
            const int j = id / per_row, rem = id - j * per_row, c = rem / k, slot = rem - c * k;
            if (cum_s[j] != -INFINITY) {
                if (fin_s[j]) {
                    if (rem == 0) { cv[q] = cum_s[j]; cc[q] = j * Vx; }        // an ended hypothesis competes as it is
                } else {
                    const float* rc = a.rec + ((r0 + j) * nchunk + c) * kBeamRec;
                    const int idx = __float_as_int(rc[3 + 2 * slot]);
                    if (idx != kNone) { cv[q] = cum_s[j] - lse_s[j] + rc[2 + 2 * slot]; cc[q] = j * Vx + idx; }
                }
            }
        }
    }
    for (int round = 0; round < k; ++round) {
        float bv = -INFINITY; int bc = kNone;
#pragma unroll
        for (int q = 0; q < NC; ++q) {
            const bool take = cv[q] > bv || (cv[q] == bv && cc[q] < bc);
            bv = take ? cv[q] : bv;
            bc = take ? cc[q] : bc;
        }
        block_best(bv, bc, shv, shc);
#pragma unroll
        for (int q = 0; q < NC; ++q)
            if (cc[q] == bc) { cv[q] = -INFINITY; cc[q] = kNone; }
        if (tid == 0) {
            float v = bv; const int c = bc;
            int parent = 0, tok = a.pad_token, nf = 1;
            if (c != kNone) {
                parent = c / Vx;
                if (fin_s[parent]) { nf = 1; tok = a.pad_token; }
                else { tok = c - parent * Vx; nf = tok == a.end_token; }
            } else {
                v = -INFINITY;                                      // fewer candidates than beams: a dead slot
            }
            parent_s[round] = parent; tok_s[round] = tok; nfin_s[round] = nf;
            a.cum[r0 + round] = v;
            a.fin[r0 + round] = nf;
            const bool live = c != kNone && !nf;
            a.next_token[r0 + round] = live ? tok : 0;
            a.next_mask[r0 + round] = !live ? 0 : ((a.has_facts && tok >= a.V + a.K) ? 2 : (tok >= a.V ? 1 : 0));
        }
    }
    __syncthreads();
    if (tid == 0) {
        int before = 0, after = 0;
        for (int j = 0; j < k; ++j) { before += fin_s[j] || cum_s[j] == -INFINITY; after += nfin_s[j]; }
        if (after != before) atomicAdd(a.n_done, after - before);
    }
    // histories of the chosen parents move to their new rows
    for (int idx = tid; idx < k * a.max_len; idx += 256) {
        const int j = idx / a.max_len, p = idx - j * a.max_len;
        const int64_t src = (r0 + parent_s[j]) * a.max_len + p, dst = (r0 + j) * a.max_len + p;
        const bool carried = fin_s[parent_s[j]] != 0;
        a.seq_out[dst] = p < a.step ? a.seq_in[src] : (p == a.step && !carried ? (int64_t)tok_s[j]
                                                       : (carried ? a.seq_in[src] : (int64_t)a.pad_token));
        a.anc_out[dst] = p < a.step ? a.anc_in[src] : (p == a.step ? (int32_t)(r0 + parent_s[j]) : 0);
        if (a.cap_out != nullptr)
            a.cap_out[dst] = p <= a.step ? a.cap_in[src]
                                         : (p == a.step + 1 && !nfin_s[j] ? (int64_t)tok_s[j] : (int64_t)a.start_token);
    }
    if (a.step + 1 >= a.max_len) return;
    const float* pe = a.pe + (int64_t)(a.step + 1) * a.d;
    for (int idx = tid; idx < k * a.d; idx += 256) {
        const int j = idx / a.d, c = idx - j * a.d;
        const int tok = nfin_s[j] ? 0 : tok_s[j];
        const int64_t b = blockIdx.x;
        const float* src;
        if (tok >= a.V + a.K && a.has_facts && a.fe != nullptr) {
            int e = tok - a.V - a.K;
            if (e >= a.F) e = a.F - 1;
            src = a.fe + (b * a.F + e) * a.d;
        } else if (tok >= a.V) {
            int e = tok - a.V;
            if (e >= a.K) e = a.K - 1;
            src = a.ee + (b * a.K + e) * a.d;
        } else {
            src = a.word_emb + (int64_t)tok * a.d;
        }
        a.x0[(r0 + j) * a.d + c] = fmaf(src[c], a.emb_scale, pe[c]);
    }
}

}  // namespace
}  // namespace ick

using namespace ick;

static RowSrc make_src(const ick_decode_ctx* c, const float* res, const float* part, int nparts, const float* bias,
                       const float* gamma, const float* beta, float* out) {
    RowSrc s;
    s.res = res; s.part = part; s.nparts = nparts; s.bias = bias; s.gamma = gamma; s.beta = beta; s.out = out;
    s.eps = c->ln_eps;
    return s;
}


// ---------------------------------------------------------------------------------------------------------
// How many rows a workgroup takes.  The decode step is bound by what every CU pulls through its L1 (a head's q|k|v +
// out_proj slices are 144 KB, a 64-unit FFN chunk 154 KB, a (row, head) of cross K/V 55 KB at S = 216): rows that share
// a workgroup share that stream, and the grid should not exceed one workgroup per CU.  ICK_DEC_G="self,cross,ffn"
// overrides the choice (tuning).
// ---------------------------------------------------------------------------------------------------------
struct GroupPlan { int g_self, g_cross, g_ffn; bool cross_shared; };
static int pick_group(int units_per_row_group, int R, const int* cand, int ncand) {
    int g = cand[ncand - 1];
    for (int i = 0; i < ncand; ++i)
        if (units_per_row_group * ceil_div(R, cand[i]) <= kNumCU) { g = cand[i]; break; }
    return g;
}
static GroupPlan plan_groups(const ick_decode_ctx* c) {
    static const int pow2[] = {1, 2, 4, 8};
    GroupPlan p;
    const int R = c->R, rps = c->rows_per_sample;
    p.g_self = pick_group(c->H, R, pow2, 4);
    p.g_ffn = pick_group(ceil_div(c->FF, 64), R, pow2, 4);
    p.cross_shared = false;
    p.g_cross = pick_group(c->H, R, pow2, 4);
    if (rps > 1) {
        // hypotheses of one caption read the same K / V: the largest group that divides the beam
        static const int shared[] = {8, 5, 4, 3, 2};
        p.g_cross = 1;
        for (int g : shared)
            if (rps % g == 0) { p.g_cross = g; p.cross_shared = true; break; }
    }
    return p;
}
static void launch_self(int g, bool fsel, dim3 grid, hipStream_t s, const SelfArgs& a) {
    if (fsel) {
        switch (g) {
        case 1: hipLaunchKernelGGL((dec_self_kernel<1, true>), grid, dim3(kNT), 0, s, a); break;
        case 2: hipLaunchKernelGGL((dec_self_kernel<2, true>), grid, dim3(kNT), 0, s, a); break;
        case 4: hipLaunchKernelGGL((dec_self_kernel<4, true>), grid, dim3(kNT), 0, s, a); break;
        default: hipLaunchKernelGGL((dec_self_kernel<8, true>), grid, dim3(kNT), 0, s, a); break;
        }
        return;
    }
    switch (g) {
    case 1: hipLaunchKernelGGL((dec_self_kernel<1, false>), grid, dim3(kNT), 0, s, a); break;
    case 2: hipLaunchKernelGGL((dec_self_kernel<2, false>), grid, dim3(kNT), 0, s, a); break;
    case 4: hipLaunchKernelGGL((dec_self_kernel<4, false>), grid, dim3(kNT), 0, s, a); break;
    default: hipLaunchKernelGGL((dec_self_kernel<8, false>), grid, dim3(kNT), 0, s, a); break;
    }
}
static void launch_ffn(int g, dim3 grid, hipStream_t s, const FfnArgs& a) {
    switch (g) {
    case 1: hipLaunchKernelGGL(dec_ffn_kernel<1>, grid, dim3(kNT), 0, s, a); break;
    case 2: hipLaunchKernelGGL(dec_ffn_kernel<2>, grid, dim3(kNT), 0, s, a); break;
    case 4: hipLaunchKernelGGL(dec_ffn_kernel<4>, grid, dim3(kNT), 0, s, a); break;
    default: hipLaunchKernelGGL(dec_ffn_kernel<8>, grid, dim3(kNT), 0, s, a); break;
    }
}
static void launch_cross(int g, bool shared, dim3 grid, hipStream_t s, const CrossArgs& a) {
    if (!shared) {
        switch (g) {
        case 1: hipLaunchKernelGGL((dec_cross_kernel<1, false>), grid, dim3(kNT), 0, s, a); break;
        case 2: hipLaunchKernelGGL((dec_cross_kernel<2, false>), grid, dim3(kNT), 0, s, a); break;
        case 4: hipLaunchKernelGGL((dec_cross_kernel<4, false>), grid, dim3(kNT), 0, s, a); break;
        default: hipLaunchKernelGGL((dec_cross_kernel<8, false>), grid, dim3(kNT), 0, s, a); break;
        }
        return;
    }
    switch (g) {
    case 2: hipLaunchKernelGGL((dec_cross_kernel<2, true>), grid, dim3(kNT), 0, s, a); break;
    case 3: hipLaunchKernelGGL((dec_cross_kernel<3, true>), grid, dim3(kNT), 0, s, a); break;
    case 4: hipLaunchKernelGGL((dec_cross_kernel<4, true>), grid, dim3(kNT), 0, s, a); break;
    case 5: hipLaunchKernelGGL((dec_cross_kernel<5, true>), grid, dim3(kNT), 0, s, a); break;
    default: hipLaunchKernelGGL((dec_cross_kernel<8, true>), grid, dim3(kNT), 0, s, a); break;
    }
}

extern "C" int ick_decode_supported(int32_t d, int32_t H, int32_t FF, int32_t S, int32_t max_len) {
    return d > 0 && d % 4 == 0 && d <= kDMax && d >= 64 && H > 0 && H <= kPartsMax && d % H == 0 && d / H <= 32 && FF > 0 &&
           FF % 4 == 0 && FF <= 64 * kPartsMax &&
           S > 0 && S <= kSMax && max_len > 0 && max_len <= kMLMax;
}

extern "C" int ick_decode_beam_supported(int32_t Vx, int32_t beam) {
    if (Vx <= 0 || beam < 1 || beam > kBeamMax) return 0;
    return (int64_t)beam * beam * ceil_div(Vx, kBeamChunk) <= 256 * kBeamCandPerThread;   // candidates the selection holds
}

// which: bit 0 self, 1 cross, 2 ffn, 3 head, 4 vocabulary (all set in the product path; the diagnostic build times subsets)
static int decode_layers_impl(const ick_decode_ctx* c, int32_t pos, void* stream, unsigned which, int part = 0) {
    ICK_CHECK_ARG(c && c->R > 0 && c->layers > 0 && c->layers <= ICK_MAX_LAYERS && pos >= 0 && pos < c->max_len);
    ICK_CHECK_ARG(ick_decode_supported(c->d, c->H, c->FF, c->S, c->max_len));
    ICK_CHECK_ARG(c->rows_per_sample > 0 && c->R % c->rows_per_sample == 0 && c->R <= 65535);
    ICK_CHECK_ARG(c->x0 && c->xa && c->xb && c->xc && c->p1 && c->p2 && c->p3 && c->hfin && c->hv && c->ptr && c->cand);
    ICK_CHECK_ARG(c->H <= kPartsMax && ceil_div(c->FF, 64) <= kPartsMax);
    hipStream_t s = (hipStream_t)stream;
    const int d = c->d, H = c->H, dh = d / H, R = c->R;
    const int nch = ceil_div(c->FF, 64);
    const float scale = 1.f / sqrtf((float)dh);
    const GroupPlan plan = plan_groups(c);
    RowSrc src = make_src(c, c->x0, nullptr, 0, nullptr, nullptr, nullptr, c->xa);
    for (int l = 0; l < c->layers; ++l) {
        const ick_decode_layer& w = c->layer[l];
        ICK_CHECK_ARG(w.sa_in_w && w.sa_in_b && w.sa_out_wt && w.sa_out_b && w.ca_in_w && w.ca_in_b && w.ca_out_wt &&
                      w.ca_out_b && w.w1 && w.b1 && w.w2t && w.b2 && w.n1_g && w.n1_b && w.n2_g && w.n2_b && w.n3_g &&
                      w.n3_b && w.self_k && w.self_v && w.cross_k && w.cross_v);
        SelfArgs sa;
        sa.w.in_w = w.sa_in_w; sa.w.in_b = w.sa_in_b; sa.w.out_wt = w.sa_out_wt; sa.w.src = src; sa.w.part = c->p1;
        sa.kc = w.self_k; sa.vc = w.self_v; sa.anc = c->anc;
        sa.R = R; sa.d = d; sa.H = H; sa.dh = dh; sa.ML = c->max_len; sa.pos = pos; sa.scale = scale;
        sa.n_done = c->n_done; sa.n_total = R;
        const bool fsel = l == 0 && pos >= 1 && c->sel_state != nullptr;
        if (fsel) {
            // the token of step pos - 1 is chosen inside this launch (fused_select)
            ICK_CHECK_ARG(c->rows_per_sample == 1 && c->output && c->hist && c->finished && c->n_done && c->next_token &&
                          c->next_mask && c->word_emb && c->pe && c->cand && c->ptr && c->ee);
            SelFuse& f = sa.sel;
            f.cand = reinterpret_cast<const float4*>(c->cand); f.ntiles = ceil_div(c->V, kVocabTile); f.ptr = c->ptr;
            f.output = c->output; f.hist = c->hist; f.finished = c->finished; f.n_done = c->n_done;
            f.next_token = c->next_token; f.next_mask = c->next_mask; f.cap_buf = c->cap_buf;
            const int i = pos - 1;
            f.st_in = c->sel_state + (size_t)(i & 1) * R * 12; f.st_out = c->sel_state + (size_t)((i + 1) & 1) * R * 12;
            f.word_emb = c->word_emb; f.ee = c->ee; f.fe = c->F > 0 ? c->fe : nullptr; f.pe = c->pe;
            f.rows_per_sample = c->rows_per_sample; f.V = c->V; f.K = c->K; f.F = c->F; f.step = i; f.max_len = c->max_len;
            f.has_facts = c->F > 0; f.end_token = c->end_token; f.pad_token = c->pad_token; f.emb_scale = c->emb_scale;
        }
        const bool first_self = l == 0;
        if ((which & 1u) && !(part == 2 && first_self))
            launch_self(plan.g_self, fsel, dim3(H, ceil_div(R, plan.g_self)), s, sa);
        if (part == 1) { ICK_LAUNCH_RET(); }
        CrossArgs ca;
        ca.w.in_w = w.ca_in_w; ca.w.in_b = w.ca_in_b; ca.w.out_wt = w.ca_out_wt;
        ca.w.src = make_src(c, c->xa, c->p1, H, w.sa_out_b, w.n1_g, w.n1_b, c->xb);
        ca.w.part = c->p2;
        ca.Kmem = w.cross_k; ca.Vmem = w.cross_v; ca.kv_bs = c->kv_bs;
        ca.R = R; ca.rows_per_sample = c->rows_per_sample; ca.d = d; ca.H = H; ca.dh = dh; ca.S = c->S; ca.scale = scale;
        ca.n_done = c->n_done; ca.n_total = R;
        if (which & 2u) launch_cross(plan.g_cross, plan.cross_shared, dim3(H, ceil_div(R, plan.g_cross)), s, ca);
        FfnArgs fa;
        fa.w1 = w.w1; fa.b1 = w.b1; fa.w2t = w.w2t;
        fa.src = make_src(c, c->xb, c->p2, H, w.ca_out_b, w.n2_g, w.n2_b, c->xc);
        fa.part = c->p3; fa.R = R; fa.d = d; fa.FF = c->FF; fa.n_done = c->n_done; fa.n_total = R;
        if (which & 4u) launch_ffn(plan.g_ffn, dim3(nch, ceil_div(R, plan.g_ffn)), s, fa);
        // the next consumer normalises: LayerNorm3(xc + b2 + sum of the chunk partials)
        src = make_src(c, c->xc, c->p3, nch, w.b2, w.n3_g, w.n3_b, l + 1 < c->layers ? c->xa : c->hfin);
    }
    ICK_CHECK_ARG(c->wv && c->bv && c->ee && c->we && c->be && c->V > 0 && c->K > 0);
    ICK_CHECK_ARG(c->F == 0 || (c->fe && c->wf && c->bf));
    HeadArgs ha;
    ha.src = src; ha.gate = c->gate; ha.hv = c->hv; ha.ee = c->ee; ha.we = c->we; ha.be = c->be;
    ha.fe = c->F > 0 ? c->fe : nullptr; ha.wf = c->wf; ha.bf = c->bf; ha.eib = c->eib; ha.ptr = c->ptr;
    ha.R = R; ha.rows_per_sample = c->rows_per_sample; ha.d = d; ha.K = c->K; ha.F = c->F;
    ha.n_done = c->n_done; ha.n_total = R;
    VocabArgs va;
    va.hv = c->hv; va.wv = c->wv; va.bv = c->bv; va.scores = c->scores; va.ld = c->scores_ld;
    va.cand = reinterpret_cast<float4*>(c->cand); va.R = R; va.d = d; va.V = c->V; va.ntiles = ceil_div(c->V, kVocabTile);
    va.n_done = c->n_done; va.n_total = R;
    // one launch for both when the rows fit one 32-row block and the final LayerNorm's sources fit the merged kernel's
    // registers (greedy decoding at cfg5)
    if ((which & 24u) == 24u && R <= 32 && src.nparts + 2 <= kHvSrc) {
        hipLaunchKernelGGL(dec_headvocab_kernel, dim3(va.ntiles + R), dim3(kNT), 0, s, ha, va);
        ICK_LAUNCH_RET();
    }
    if (which & 8u) hipLaunchKernelGGL(dec_head_kernel, dim3(R), dim3(256), 0, s, ha);
    if (which & 16u) hipLaunchKernelGGL(dec_vocab_kernel, dim3(va.ntiles), dim3(kNT), 0, s, va);
    ICK_LAUNCH_RET();
}

extern "C" int ick_decode_layers(const ick_decode_ctx* c, int32_t pos, void* stream) {
    return decode_layers_impl(c, pos, stream, 31u);
}

extern "C" int ick_decode_layers_part(const ick_decode_ctx* c, int32_t pos, int32_t part, void* stream) {
    ICK_CHECK_ARG(part >= 0 && part <= 2);
    return decode_layers_impl(c, pos, stream, 31u, part);
}

// every per-call buffer of a greedy decode in one launch (they were eight fills, an embedding and a copy)
namespace ick { namespace {
__global__ __launch_bounds__(256) void dec_init_kernel(ick_decode_ctx c, int start_token, int n_done_init) {
    const int64_t r = blockIdx.x;
    const int tid = threadIdx.x;
    for (int p = tid; p < c.max_len; p += 256) {
        c.output[r * c.max_len + p] = c.pad_token;
        c.hist[r * c.max_len + p] = 0;
        if (c.cap_buf) c.cap_buf[r * c.max_len + p] = start_token;
    }
    if (tid == 0) {
        c.finished[r] = 0; c.next_token[r] = 0; c.next_mask[r] = 0;
        if (r == 0) *c.n_done = n_done_init;
    }
    if (c.sel_state && tid < 24) c.sel_state[(tid / 12) * (int64_t)c.R * 12 + r * 12 + tid % 12] = 0;
    // embedding of <start> at position 0 (CaptionEmbedder + sqrt(d) + PositionEncoder)
    const float* src = c.word_emb + (int64_t)start_token * c.d;
    for (int col = tid; col < c.d; col += 256) c.x0[r * c.d + col] = fmaf(src[col], c.emb_scale, c.pe[col]);
}
} }

extern "C" int ick_decode_init(const ick_decode_ctx* c, int32_t start_token, int32_t n_done_init, void* stream) {
    ICK_CHECK_ARG(c && c->R > 0 && c->max_len > 0 && c->d > 0 && c->output && c->hist && c->finished && c->n_done &&
                  c->next_token && c->next_mask && c->word_emb && c->pe && c->x0);
    ICK_CHECK_ARG(start_token >= 0 && start_token < c->V);
    hipLaunchKernelGGL(dec_init_kernel, dim3(c->R), dim3(256), 0, (hipStream_t)stream, *c, start_token, n_done_init);
    ICK_LAUNCH_RET();
}

#ifdef ICK_DECODE_STAMPS
// diagnostic build only: `reps` passes over a subset of the step's kernels (tools/debug/decode_repeat.py)
extern "C" int ick_debug_decode_subset(const ick_decode_ctx* c, int32_t pos, unsigned which, int32_t reps, void* stream) {
    for (int i = 0; i < reps; ++i) {
        const int rc = decode_layers_impl(c, pos, stream, which);
        if (rc != 0) return rc;
    }
    return 0;
}
#endif

extern "C" int ick_decode_select_greedy(const ick_decode_ctx* c, int32_t pos, void* stream) {
    ICK_CHECK_ARG(c && c->R > 0 && pos >= 0 && pos < c->max_len);
    ICK_CHECK_ARG(c->output && c->hist && c->finished && c->n_done && c->next_token && c->next_mask && c->word_emb &&
                  c->pe && c->x0 && c->cand && c->ptr && c->ee);
    SelectArgs a;
    a.cand = reinterpret_cast<const float4*>(c->cand); a.ntiles = ceil_div(c->V, kVocabTile); a.ptr = c->ptr;
    a.output = c->output; a.hist = c->hist; a.finished = c->finished; a.n_done = c->n_done;
    a.next_token = c->next_token; a.next_mask = c->next_mask; a.cap_buf = c->cap_buf;
    a.word_emb = c->word_emb; a.ee = c->ee; a.fe = c->F > 0 ? c->fe : nullptr; a.pe = c->pe; a.x0 = c->x0;
    a.R = c->R; a.rows_per_sample = c->rows_per_sample; a.d = c->d; a.V = c->V; a.K = c->K; a.F = c->F; a.step = pos;
    a.max_len = c->max_len; a.has_facts = c->F > 0; a.end_token = c->end_token; a.pad_token = c->pad_token;
    a.emb_scale = c->emb_scale; a.n_total = c->R;
    hipLaunchKernelGGL(dec_select_kernel, dim3(c->R), dim3(256), 0, (hipStream_t)stream, a);
    ICK_LAUNCH_RET();
}

extern "C" int ick_decode_select_beam(const ick_decode_ctx* c, const ick_beam_state* bs, int32_t pos, void* stream) {
    ICK_CHECK_ARG(c && bs && c->R > 0 && pos >= 0 && pos < c->max_len);
    ICK_CHECK_ARG(c->rows_per_sample >= 1 && c->rows_per_sample <= kBeamMax && c->R % c->rows_per_sample == 0);
    ICK_CHECK_ARG(c->scores && c->scores_ld >= c->V && c->ptr && c->n_done && c->next_token && c->next_mask &&
                  c->word_emb && c->pe && c->x0 && c->ee);
    ICK_CHECK_ARG(bs->cum && bs->fin && bs->seq_in && bs->seq_out && bs->anc_in && bs->anc_out);
    ICK_CHECK_ARG((bs->cap_in == nullptr) == (bs->cap_out == nullptr));
    ICK_CHECK_ARG(bs->rec != nullptr);
    const int Vx = c->V + c->K + c->F, nchunk = ceil_div(Vx, kBeamChunk);
    ICK_CHECK_ARG(ick_decode_beam_supported(Vx, c->rows_per_sample));
    BeamPartArgs pa;
    pa.scores = c->scores; pa.ld = c->scores_ld; pa.ptr = c->ptr; pa.cum = bs->cum; pa.fin = bs->fin; pa.rec = bs->rec;
    pa.R = c->R; pa.k = c->rows_per_sample; pa.V = c->V; pa.np = c->K + c->F; pa.nchunk = nchunk;
    pa.n_done = c->n_done; pa.n_total = c->R;
    hipLaunchKernelGGL(dec_beam_partial_kernel, dim3(nchunk, c->R), dim3(256), 0, (hipStream_t)stream, pa);
    BeamArgs a;
    a.rec = bs->rec; a.cum = bs->cum; a.fin = bs->fin;
    a.seq_in = bs->seq_in; a.seq_out = bs->seq_out; a.anc_in = bs->anc_in; a.anc_out = bs->anc_out;
    a.cap_in = bs->cap_in; a.cap_out = bs->cap_out; a.n_done = c->n_done;
    a.next_token = c->next_token; a.next_mask = c->next_mask;
    a.word_emb = c->word_emb; a.ee = c->ee; a.fe = c->F > 0 ? c->fe : nullptr; a.pe = c->pe; a.x0 = c->x0;
    a.R = c->R; a.k = c->rows_per_sample; a.d = c->d; a.V = c->V; a.K = c->K; a.F = c->F; a.step = pos;
    a.max_len = c->max_len; a.has_facts = c->F > 0; a.end_token = c->end_token; a.pad_token = c->pad_token;
    a.start_token = bs->start_token; a.emb_scale = c->emb_scale; a.n_total = c->R;
    hipLaunchKernelGGL(dec_select_beam_kernel, dim3(c->R / c->rows_per_sample), dim3(256), 0, (hipStream_t)stream, a);
    ICK_LAUNCH_RET();
}
