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
constexpr int kSMax = 1024;  // memory rows whose scores fit the LDS score buffer
constexpr int kMLMax = 128;  // caption positions (self-attention keys)
constexpr int kDhp = 32;     // padded head width of the head-major K/V layouts

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

// x = LayerNorm(res + bias + sum partials) (or res itself) -> xs[0..kDMax) in LDS, zero beyond d.
// issue() only starts the loads; the kernels issue them FIRST and their large weight / K / V streams afterwards:
// vector-memory results return in issue order, so the row is normalised (finish()) and projected while the streams
// are still arriving.  Every partial row is loaded before the first add (a `for (p < nparts)` load-add loop would wait
// for each load in turn: ten dependent L2 round trips); the partial rows are then summed in index order.
constexpr int kPartsMax = 16;
// The np + 2 rows to add (residual, bias, partials) are dealt out to G = min(4, 256 / (d/4)) thread groups as float4
// columns: at most 6 + 2 sixteen-byte loads per thread (one dword per element would be 40 load instructions per thread
// -- with the weight stream behind them more than the 63 a wave can keep in flight, which stalls the issue itself).
struct RowIn {
    float4 acc, gm, bt;
    float4 pv[6];
    __device__ __forceinline__ void issue(const RowSrc& s, int64_t row, int d) {
        const int tid = threadIdx.x, np = s.nparts, d4 = d >> 2;
        const int G = min(256 / d4, 4);
        const int g = min(tid / d4, G - 1), c4 = min(tid - g * d4, d4 - 1);
        const int nrows = np > 0 ? np + 2 : 1;
        gm = bt = make_float4(0.f, 0.f, 0.f, 0.f);
        if (np > 0) {
            gm = reinterpret_cast<const float4*>(s.gamma)[c4];
            bt = reinterpret_cast<const float4*>(s.beta)[c4];
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int q = min(g + G * j, nrows - 1);          // row 0: residual, 1: bias, 2 + p: partial p
            const float* src = q == 0 ? s.res + row * d : (q == 1 ? s.bias : s.part + (row * np + (q - 2)) * d);
            pv[j] = reinterpret_cast<const float4*>(src)[c4];
        }
    }
    // scratch: LDS float4[4 * d/4] (the out-projection's exchange buffer, free at this point)
    __device__ __forceinline__ void finish(const RowSrc& s, int64_t row, int d, float* xs, float* red, float4* scratch,
                                           bool writer) {
        const int tid = threadIdx.x, np = s.nparts, d4 = d >> 2;
        const int G = min(256 / d4, 4);
        const int g = tid / d4, c4 = tid - g * d4;
        const int nrows = np > 0 ? np + 2 : 1;
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (g < G) {
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                if (g + G * j < nrows) { t.x += pv[j].x; t.y += pv[j].y; t.z += pv[j].z; t.w += pv[j].w; }
            }
            scratch[g * d4 + c4] = t;
        }
        __syncthreads();
        float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tid < d4) {
            z = scratch[tid];
            for (int gg = 1; gg < G; ++gg) {
                const float4 u = scratch[gg * d4 + tid];
                z.x += u.x; z.y += u.y; z.z += u.z; z.w += u.w;
            }
        }
        if (np > 0) {
            const float mean = block_sum<4>((z.x + z.y) + (z.z + z.w), red) / (float)d;
            float v = 0.f;
            if (tid < d4) {
                v = fmaf(z.x - mean, z.x - mean, v); v = fmaf(z.y - mean, z.y - mean, v);
                v = fmaf(z.z - mean, z.z - mean, v); v = fmaf(z.w - mean, z.w - mean, v);
            }
            const float rstd = rsqrtf(block_sum<4>(v, red) / (float)d + s.eps);
            z.x = (z.x - mean) * rstd * gm.x + bt.x; z.y = (z.y - mean) * rstd * gm.y + bt.y;
            z.z = (z.z - mean) * rstd * gm.z + bt.z; z.w = (z.w - mean) * rstd * gm.w + bt.w;
        }
        if (tid < kD4Max) reinterpret_cast<float4*>(xs)[tid] = tid < d4 ? z : make_float4(0.f, 0.f, 0.f, 0.f);
        if (writer && s.out != nullptr && tid < d4) reinterpret_cast<float4*>(s.out + row * d)[tid] = z;
        __syncthreads();
    }
};

// Lane exchanges inside a row of 16 lanes as DPP operands (one VALU instruction each) instead of ds_bpermute round
// trips through the LDS crossbar (~100 cycles each, four in a row per dot product): quad_perm [1,0,3,2] / [2,3,0,1]
// pair lanes inside a quad, row_half_mirror / row_mirror then pair quads and halves (every lane of a quad / half
// already holds the same partial result), row_ror:8 swaps the halves of a row.
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

// Dot products of up to 16 * NPASS weight rows (k contiguous) with the LDS vector xs: 16 lanes per row, four
// rows per wave and pass.  load() only issues the weight loads (nothing depends on xs), run() consumes them.
template <int NPASS>
struct RowDot {
    float4 w[NPASS][5];
    float bv[NPASS];
    template <typename RowIdx>
    __device__ __forceinline__ void load(const float* __restrict__ W, const float* __restrict__ bias, int64_t ld,
                                         RowIdx rowidx, int nrows, int d4) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = lane >> 4, i = lane & 15;
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            const int r = min(16 * p + 4 * wave + sub, nrows - 1);
            const float4* wr = reinterpret_cast<const float4*>(W + (int64_t)rowidx(r) * ld);
#pragma unroll
            for (int it = 0; it < 5; ++it) w[p][it] = wr[min(i + 16 * it, d4 - 1)];   // beyond d: xs is zero there
            bv[p] = bias ? bias[rowidx(r)] : 0.f;     // fetched with the weights: a load inside run() would stall it
        }
    }
    // ys[r] = (dot(W[rowidx(r)], xs) + bias[rowidx(r)]) * scale
    __device__ __forceinline__ void run(const float* xs, int nrows, float* ys, float scale = 1.f) const {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = lane >> 4, i = lane & 15;
        float4 xr[5];
#pragma unroll
        for (int it = 0; it < 5; ++it) xr[it] = reinterpret_cast<const float4*>(xs)[i + 16 * it];
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            float acc = 0.f;
#pragma unroll
            for (int it = 0; it < 5; ++it) acc += dot4(w[p][it], xr[it]);
            acc = sum16(acc);
            const int r = 16 * p + 4 * wave + sub;
            if (i == 0 && r < nrows) ys[r] = (acc + bv[p]) * scale;
        }
    }
};

// out[n] = sum_{j < nj} o[j] * Wt[(j0 + j) * ld + n] for n < d (Wt = transposed weight: one k per row, n contiguous).
// G = min(4, 256 / (d/4)) thread groups take every G-th j (nj <= 3 * JMAX); their float4 partial sums meet in LDS.
// o[] must be zero for nj <= j < G * JMAX.
template <int JMAX>
struct ColDot {
    float4 w[JMAX];
    __device__ __forceinline__ void load(const float* __restrict__ Wt, int64_t ld, int j0, int nj, int d4) {
        const int G = min(256 / d4, 4);
        const int g = min((int)threadIdx.x / d4, G - 1), c = threadIdx.x - g * d4 < d4 ? threadIdx.x - g * d4 : 0;
#pragma unroll
        for (int jj = 0; jj < JMAX; ++jj) {
            const int j = min(g + G * jj, nj - 1);
            w[jj] = reinterpret_cast<const float4*>(Wt + (int64_t)(j0 + j) * ld)[c];
        }
    }
    // part: LDS float4[G * d4]; out: global row (d floats)
    __device__ __forceinline__ void run(const float* o, int d4, float4* part, float* __restrict__ out) const {
        const int G = min(256 / d4, 4);
        const int tid = threadIdx.x;
        const int g = tid / d4, c = tid - g * d4;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (g < G) {
#pragma unroll
            for (int jj = 0; jj < JMAX; ++jj) {
                const float oj = o[g + G * jj];
                acc.x = fmaf(oj, w[jj].x, acc.x); acc.y = fmaf(oj, w[jj].y, acc.y);
                acc.z = fmaf(oj, w[jj].z, acc.z); acc.w = fmaf(oj, w[jj].w, acc.w);
            }
            part[g * d4 + c] = acc;
        }
        __syncthreads();
        if (tid < d4) {
            float4 s = part[tid];
            for (int gg = 1; gg < G; ++gg) {
                const float4 t = part[gg * d4 + tid];
                s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
            }
            reinterpret_cast<float4*>(out)[tid] = s;
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

// softmax(q . K^T) V for one (row, head) with the keys / values already in registers: thread (p8 = tid >> 3,
// c = tid & 7) holds float4 c of positions p0 + 32 q + p8 (q < NP).  Scores never touch LDS; two exchanges between
// the four waves (running maximum, then sum + weighted values) instead of one barrier per phase.  kreg / vreg must be
// masked (pad columns zero).  Result: o[0 .. 31] in LDS (zero beyond dh), valid after the function returns.
template <int NP>
__device__ __forceinline__ void attend_regs(const float4 (&kreg)[NP], const float4 (&vreg)[NP], const float4& q4, int S,
                                            int dh, float scale, float* red, float4* pvred, float* o) {
    const int tid = threadIdx.x, p8 = tid >> 3, c = tid & 7, wave = tid >> 6;
    float sc[NP];
    float m = -INFINITY;
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        const float s = sum8(dot4(q4, kreg[q]));
        sc[q] = 32 * q + p8 < S ? s * scale : -INFINITY;
        m = fmaxf(m, sc[q]);
    }
    m = fmaxf(m, dppf<kDppRor8>(m));
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    if ((tid & 63) == 0) red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float lsum = 0.f;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        const float e = sc[q] == -INFINITY ? 0.f : __expf(sc[q] - m);
        lsum += e;
        acc.x = fmaf(e, vreg[q].x, acc.x); acc.y = fmaf(e, vreg[q].y, acc.y);
        acc.z = fmaf(e, vreg[q].z, acc.z); acc.w = fmaf(e, vreg[q].w, acc.w);
    }
    lsum += dppf<kDppRor8>(lsum);
    acc.x += dppf<kDppRor8>(acc.x); acc.y += dppf<kDppRor8>(acc.y);
    acc.z += dppf<kDppRor8>(acc.z); acc.w += dppf<kDppRor8>(acc.w);
#pragma unroll
    for (int off = 16; off < 64; off <<= 1) {
        lsum += __shfl_xor(lsum, off, 64);
        acc.x += __shfl_xor(acc.x, off, 64); acc.y += __shfl_xor(acc.y, off, 64);
        acc.z += __shfl_xor(acc.z, off, 64); acc.w += __shfl_xor(acc.w, off, 64);
    }
    if ((tid & 63) < 8) pvred[wave * 8 + c] = acc;
    if ((tid & 63) == 0) red[4 + wave] = lsum;
    __syncthreads();
    if (tid < 8) {
        float4 t = pvred[tid];
#pragma unroll
        for (int w = 1; w < 4; ++w) { const float4 u = pvred[w * 8 + tid]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
        const float inv = 1.f / ((red[4] + red[5]) + (red[6] + red[7]));
        o[4 * tid + 0] = 4 * tid + 0 < dh ? t.x * inv : 0.f; o[4 * tid + 1] = 4 * tid + 1 < dh ? t.y * inv : 0.f;
        o[4 * tid + 2] = 4 * tid + 2 < dh ? t.z * inv : 0.f; o[4 * tid + 3] = 4 * tid + 3 < dh ? t.w * inv : 0.f;
    }
    __syncthreads();
}

struct LayerW {
    const float *in_w, *in_b;      // self: (3d, d) packed in_proj; cross: its first d rows (q)
    const float *out_wt;           // (d, d) TRANSPOSED out_proj weight (row = input feature)
    RowSrc src;                    // how this block's input row is formed
    float* part;                   // (R, H, d) partial out-projection rows written here
};

struct SelfArgs {
    LayerW w;
    float* kc; float* vc;          // (R, H, ML, 32) key / value cache of this layer
    const int32_t* anc;            // optional (R, ML): cache row that holds position p of row r (beam search)
    int R, d, H, dh, ML, pos;
    float scale;
    const int32_t* n_done; int n_total;
};

// ---------------------------------------------------------------------------------------------------------
// self-attention block, one workgroup per (head, row)
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dec_self_kernel(SelfArgs a) {
    if (a.n_done != nullptr && *a.n_done >= a.n_total) return;     // every caption has ended (uniform)
    __shared__ __attribute__((aligned(16))) float xs[kDMax];
    __shared__ __attribute__((aligned(16))) float qkv[96 + 32];    // q | k | v of this head (3 x dh <= 96)
    __shared__ __attribute__((aligned(16))) float o[64];
    __shared__ __attribute__((aligned(16))) float4 part[4 * kD4Max];
    __shared__ float4 pvred[32];
    __shared__ float red[8];
    const int h = blockIdx.x, tid = threadIdx.x;
    const int64_t r = blockIdx.y;
    const int d = a.d, d4 = d >> 2, dh = a.dh, pos = a.pos, S = pos + 1;
    auto rowidx = [&](int rr) { const int seg = rr / dh; return seg * d + h * dh + (rr - seg * dh); };
    ICK_STAMP(0, 0);
    RowIn in;
    in.issue(a.w.src, r, d);
    RowDot<6> qd;
    qd.load(a.w.in_w, a.w.in_b, d, rowidx, 3 * dh, d4);
    // cached keys / values of positions < pos: 8 lanes per position (float4 each), 32 positions per pass
    const int p8 = tid >> 3, c = tid & 7;
    constexpr int NP = kMLMax / 32;
    float4 kreg[NP], vreg[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        const int p = min(32 * q + p8, max(pos - 1, 0));
        const int64_t cr = a.anc ? (int64_t)a.anc[r * a.ML + p] : r;
        const int64_t off = ((cr * a.H + h) * a.ML + p) * kDhp + 4 * c;
        if (32 * q < pos) {           // uniform per pass
            kreg[q] = *reinterpret_cast<const float4*>(a.kc + off);
            vreg[q] = *reinterpret_cast<const float4*>(a.vc + off);
        } else {
            kreg[q] = vreg[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    ColDot<11> od;
    od.load(a.w.out_wt, d, h * dh, dh, d4);
    ICK_STAMP(0, 1);
    in.finish(a.w.src, r, d, xs, red, part, h == 0);
    ICK_STAMP(0, 2);
    if (tid < 32) { qkv[96 + tid] = 0.f; o[32 + tid] = 0.f; }
    qd.run(xs, 3 * dh, qkv);
    __syncthreads();
    ICK_STAMP(0, 3);
    // the new key / value row joins the cache (pad columns stay unwritten and are masked by every reader)
    if (tid < dh) a.kc[((r * a.H + h) * a.ML + pos) * kDhp + tid] = qkv[dh + tid];
    else if (tid >= 32 && tid < 32 + dh) a.vc[((r * a.H + h) * a.ML + pos) * kDhp + tid - 32] = qkv[2 * dh + tid - 32];
    // attention over positions 0 .. pos: cached rows from the registers, the new row from LDS
    float4 q4, kn, vn;
    q4.x = 4 * c + 0 < dh ? qkv[4 * c + 0] : 0.f; q4.y = 4 * c + 1 < dh ? qkv[4 * c + 1] : 0.f;
    q4.z = 4 * c + 2 < dh ? qkv[4 * c + 2] : 0.f; q4.w = 4 * c + 3 < dh ? qkv[4 * c + 3] : 0.f;
    kn.x = 4 * c + 0 < dh ? qkv[dh + 4 * c + 0] : 0.f; kn.y = 4 * c + 1 < dh ? qkv[dh + 4 * c + 1] : 0.f;
    kn.z = 4 * c + 2 < dh ? qkv[dh + 4 * c + 2] : 0.f; kn.w = 4 * c + 3 < dh ? qkv[dh + 4 * c + 3] : 0.f;
    vn.x = 4 * c + 0 < dh ? qkv[2 * dh + 4 * c + 0] : 0.f; vn.y = 4 * c + 1 < dh ? qkv[2 * dh + 4 * c + 1] : 0.f;
    vn.z = 4 * c + 2 < dh ? qkv[2 * dh + 4 * c + 2] : 0.f; vn.w = 4 * c + 3 < dh ? qkv[2 * dh + 4 * c + 3] : 0.f;
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        const int p = 32 * q + p8;
        kreg[q] = p == pos ? kn : (p < pos ? mask_cols(kreg[q], c, dh) : make_float4(0.f, 0.f, 0.f, 0.f));
        vreg[q] = p == pos ? vn : (p < pos ? mask_cols(vreg[q], c, dh) : make_float4(0.f, 0.f, 0.f, 0.f));
    }
    attend_regs<NP>(kreg, vreg, q4, S, dh, a.scale, red, pvred, o);
    ICK_STAMP(0, 4);
    od.run(o, d4, part, a.w.part + (r * a.H + h) * d);
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
// cross-attention block, one workgroup per (head, row): K and V of (sample, head) are read exactly once
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dec_cross_kernel(CrossArgs a) {
    if (a.n_done != nullptr && *a.n_done >= a.n_total) return;
    __shared__ __attribute__((aligned(16))) float xs[kDMax];
    __shared__ __attribute__((aligned(16))) float qs[32];
    __shared__ __attribute__((aligned(16))) float sc[kSMax];
    __shared__ __attribute__((aligned(16))) float o[64];
    __shared__ __attribute__((aligned(16))) float4 part[4 * kD4Max];
    __shared__ float4 pvred[32];
    __shared__ float red[8];
    const int h = blockIdx.x, tid = threadIdx.x;
    const int64_t r = blockIdx.y;
    const int d = a.d, d4 = d >> 2, dh = a.dh, S = a.S;
    auto rowidx = [&](int rr) { return h * dh + rr; };
    ICK_STAMP(1, 0);
    RowIn in;
    in.issue(a.w.src, r, d);
    RowDot<2> qd;
    qd.load(a.w.in_w, a.w.in_b, d, rowidx, dh, d4);
    const int p8 = tid >> 3, c = tid & 7;
    const int64_t b = r / a.rows_per_sample;
    const float* Kb = a.Kmem + b * a.kv_bs + (int64_t)h * S * kDhp;
    const float* Vb = a.Vmem + b * a.kv_bs + (int64_t)h * S * kDhp;
    constexpr int NP = 8;            // 256 memory rows per sweep
    float4 kreg[NP], vreg[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        const int p = min(32 * q + p8, S - 1);
        kreg[q] = *reinterpret_cast<const float4*>(Kb + (int64_t)p * kDhp + 4 * c);
    }
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        const int p = min(32 * q + p8, S - 1);
        vreg[q] = *reinterpret_cast<const float4*>(Vb + (int64_t)p * kDhp + 4 * c);
    }
    ColDot<11> od;
    od.load(a.w.out_wt, d, h * dh, dh, d4);
    ICK_STAMP(1, 1);
    in.finish(a.w.src, r, d, xs, red, part, h == 0);
    ICK_STAMP(1, 2);
    if (tid < 32) { qs[tid] = 0.f; o[32 + tid] = 0.f; }
    __syncthreads();
    qd.run(xs, dh, qs, a.scale);     // q * 1/sqrt(dh), as nn.MultiheadAttention scales it
    __syncthreads();
    ICK_STAMP(1, 3);
    const float4 q4 = reinterpret_cast<const float4*>(qs)[c];   // pad entries are zero; already scaled by 1/sqrt(dh)
    if (S <= 32 * NP) {
        // the whole memory is in the registers: scores, softmax and P.V without an LDS score buffer
#pragma unroll
        for (int q = 0; q < NP; ++q) { kreg[q] = mask_cols(kreg[q], c, dh); vreg[q] = mask_cols(vreg[q], c, dh); }
        attend_regs<NP>(kreg, vreg, q4, S, dh, 1.f, red, pvred, o);
    } else {
        for (int s0 = 0; s0 < S; s0 += 32 * NP) {
            if (s0 > 0) {
    #pragma unroll
                for (int q = 0; q < NP; ++q) {
                    const int p = min(s0 + 32 * q + p8, S - 1);
                    kreg[q] = *reinterpret_cast<const float4*>(Kb + (int64_t)p * kDhp + 4 * c);
                }
            }
    #pragma unroll
            for (int q = 0; q < NP; ++q) {
                const int p = s0 + 32 * q + p8;
                float s = dot4(q4, mask_cols(kreg[q], c, dh));
                s += __shfl_xor(s, 1, 64);
                s += __shfl_xor(s, 2, 64);
                s += __shfl_xor(s, 4, 64);
                if (c == 0 && p < S) sc[p] = s;
            }
        }
        __syncthreads();
        ICK_STAMP(1, 4);
        float m = -INFINITY;
        for (int p = tid; p < S; p += 256) m = fmaxf(m, sc[p]);
        m = block_max<4>(m, red);
        float e = 0.f;
        for (int p = tid; p < S; p += 256) { const float t = __expf(sc[p] - m); sc[p] = t; e += t; }
        const float denom = block_sum<4>(e, red);
        __syncthreads();
        ICK_STAMP(1, 5);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int s0 = 0; s0 < S; s0 += 32 * NP) {
            if (s0 > 0) {
    #pragma unroll
                for (int q = 0; q < NP; ++q) {
                    const int p = min(s0 + 32 * q + p8, S - 1);
                    vreg[q] = *reinterpret_cast<const float4*>(Vb + (int64_t)p * kDhp + 4 * c);
                }
            }
    #pragma unroll
            for (int q = 0; q < NP; ++q) {
                const int p = s0 + 32 * q + p8;
                const float pr = p < S ? sc[p] : 0.f;
                const float4 v4 = mask_cols(vreg[q], c, dh);
                acc.x = fmaf(pr, v4.x, acc.x); acc.y = fmaf(pr, v4.y, acc.y);
                acc.z = fmaf(pr, v4.z, acc.z); acc.w = fmaf(pr, v4.w, acc.w);
            }
        }
    #pragma unroll
        for (int off = 8; off < 64; off <<= 1) {
            acc.x += __shfl_xor(acc.x, off, 64); acc.y += __shfl_xor(acc.y, off, 64);
            acc.z += __shfl_xor(acc.z, off, 64); acc.w += __shfl_xor(acc.w, off, 64);
        }
        if ((tid & 63) < 8) pvred[(tid >> 6) * 8 + c] = acc;
        __syncthreads();
        if (tid < 8) {
            float4 t = pvred[tid];
    #pragma unroll
            for (int w = 1; w < 4; ++w) { const float4 u = pvred[w * 8 + tid]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
            const float inv = 1.f / denom;
            o[4 * tid + 0] = 4 * tid + 0 < dh ? t.x * inv : 0.f; o[4 * tid + 1] = 4 * tid + 1 < dh ? t.y * inv : 0.f;
            o[4 * tid + 2] = 4 * tid + 2 < dh ? t.z * inv : 0.f; o[4 * tid + 3] = 4 * tid + 3 < dh ? t.w * inv : 0.f;
        }
        __syncthreads();
    }
    ICK_STAMP(1, 6);
    od.run(o, d4, part, a.w.part + (r * a.H + h) * d);
    ICK_STAMP(1, 7);
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
// feed-forward block, one workgroup per (64 hidden units, row)
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dec_ffn_kernel(FfnArgs a) {
    if (a.n_done != nullptr && *a.n_done >= a.n_total) return;
    __shared__ __attribute__((aligned(16))) float xs[kDMax];
    __shared__ __attribute__((aligned(16))) float f[96];
    __shared__ __attribute__((aligned(16))) float4 part[4 * kD4Max];
    __shared__ float red[8];
    const int ch = blockIdx.x, tid = threadIdx.x, nch = gridDim.x;
    const int64_t r = blockIdx.y;
    const int d = a.d, d4 = d >> 2;
    const int j0 = ch * 64, nj = min(64, a.FF - j0);
    auto rowidx = [&](int rr) { return j0 + rr; };
    ICK_STAMP(2, 0);
    RowIn in;
    in.issue(a.src, r, d);
    RowDot<4> fd;
    fd.load(a.w1, a.b1, d, rowidx, nj, d4);
    ColDot<22> od;
    od.load(a.w2t, d, j0, nj, d4);
    ICK_STAMP(2, 1);
    in.finish(a.src, r, d, xs, red, part, ch == 0);
    ICK_STAMP(2, 2);
    if (tid < 96) f[tid] = 0.f;
    __syncthreads();
    fd.run(xs, nj, f);
    __syncthreads();
    ICK_STAMP(2, 3);
    if (tid < nj) f[tid] = fmaxf(f[tid], 0.f);
    __syncthreads();
    od.run(f, d4, part, a.part + (r * nch + ch) * d);
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

__global__ __launch_bounds__(256) void dec_head_kernel(HeadArgs a) {
    if (a.n_done != nullptr && *a.n_done >= a.n_total) return;
    __shared__ __attribute__((aligned(16))) float xs[kDMax];
    __shared__ __attribute__((aligned(16))) float4 part[4 * kD4Max];
    __shared__ float red[8];
    const int tid = threadIdx.x;
    const int64_t r = blockIdx.x;
    const int d = a.d, d4 = d >> 2;
    const int64_t b = r / a.rows_per_sample;
    RowIn in;
    in.issue(a.src, r, d);
    in.finish(a.src, r, d, xs, red, part, true);
    for (int c = tid; c < d; c += 256) a.hv[r * d + c] = a.gate ? xs[c] * a.gate[r * d + c] : xs[c];
    const int lane = tid & 63, wave = tid >> 6, sub = lane >> 4, i = lane & 15;
    for (int part = 0; part < 2; ++part) {
        const float* ctx = part == 0 ? a.ee : a.fe;
        const int n = part == 0 ? a.K : a.F;
        if (ctx == nullptr || n <= 0) continue;
        const float* w = part == 0 ? a.we : a.wf;
        float4 xr[5], wr[5];
#pragma unroll
        for (int it = 0; it < 5; ++it) {
            xr[it] = reinterpret_cast<const float4*>(xs)[i + 16 * it];      // zero beyond d
            wr[it] = reinterpret_cast<const float4*>(w)[min(i + 16 * it, d4 - 1)];
        }
        const float bias = part == 0 ? a.be[0] : a.bf[0];
        for (int k0 = 0; k0 < n; k0 += 16) {
            const int k = k0 + 4 * wave + sub;
            const float4* cr = reinterpret_cast<const float4*>(ctx + (b * n + min(k, n - 1)) * d);
            float4 cv[5];
#pragma unroll
            for (int it = 0; it < 5; ++it) cv[it] = cr[min(i + 16 * it, d4 - 1)];
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
// vocabulary logits: workgroup = 16 words x 32 rows, the four waves split K; fp32 MFMA 16x16x4
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dec_vocab_kernel(VocabArgs a) {
    if (a.n_done != nullptr && *a.n_done >= a.n_total) return;
    __shared__ __attribute__((aligned(16))) float red[4][2][16][17];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fi = lane & 15, fq = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int d = a.d;
    const int nchunk = (d + 15) >> 4;                  // 16-wide k chunks; <= 20
    const int per = (nchunk + 3) >> 2;                 // chunks per wave; <= 5
    const int c0 = wave * per, c1 = min(nchunk, c0 + per);
    const int64_t rb = (int64_t)min(n0 + fi, a.V - 1) * d;
    ICK_STAMP(3, 0);
    const int row = tid >> 3, cp = tid & 7;          // epilogue role: one row, two columns
    float bias2[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) bias2[e] = a.bv[min(n0 + 2 * cp + e, a.V - 1)];
    // this wave's K slice of the 16 weight rows: fetched once, reused for every block of 32 rows (beam search
    // decodes captions x beams rows; the vocabulary matrix is the large operand)
    float4 bw[5];
    int koff[5];
    bool kok[5];
#pragma unroll
    for (int t = 0; t < 5; ++t) {
        const int k = 16 * (c0 + t) + 4 * fq;
        kok[t] = c0 + t < c1 && k < d;               // d % 4 == 0: a float4 is entirely inside or outside
        koff[t] = kok[t] ? k : 0;
        bw[t] = *reinterpret_cast<const float4*>(a.wv + rb + koff[t]);
        if (!kok[t]) bw[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int m0 = 0; m0 < a.R; m0 += 32) {
        const int64_t ra0 = (int64_t)min(m0 + fi, a.R - 1) * d, ra1 = (int64_t)min(m0 + 16 + fi, a.R - 1) * d;
        float4 av0[5], av1[5];
#pragma unroll
        for (int t = 0; t < 5; ++t) {
            av0[t] = *reinterpret_cast<const float4*>(a.hv + ra0 + koff[t]);
            av1[t] = *reinterpret_cast<const float4*>(a.hv + ra1 + koff[t]);
            if (!kok[t]) av0[t] = av1[t] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        ICK_STAMP(3, 1);
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 5; ++t) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av0[t].x, bw[t].x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av1[t].x, bw[t].x, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av0[t].y, bw[t].y, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av1[t].y, bw[t].y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av0[t].z, bw[t].z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av1[t].z, bw[t].z, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av0[t].w, bw[t].w, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av1[t].w, bw[t].w, acc1, 0, 0, 0);
        }
        // C/D map: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            red[wave][0][fq * 4 + rg][fi] = acc0[rg];
            red[wave][1][fq * 4 + rg][fi] = acc1[rg];
        }
        ICK_STAMP(3, 2);
        __syncthreads();
        ICK_STAMP(3, 3);
        // thread (row = tid >> 3, two columns): sum the four K slices in a fixed order, add the bias
        const int gr = m0 + row;
        Top2 t2{-INFINITY, -INFINITY, kNone, kNone};
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int col = 2 * cp + e, n = n0 + col;
            float v = ((red[0][row >> 4][row & 15][col] + red[1][row >> 4][row & 15][col]) +
                       red[2][row >> 4][row & 15][col]) + red[3][row >> 4][row & 15][col];
            if (n < a.V) {
                v += bias2[e];
                if (a.scores != nullptr && gr < a.R) a.scores[(int64_t)gr * a.ld + n] = v;
                top2_push(t2, v, n);
            }
        }
        t2 = top2_merge_dpp<kDppXor1>(t2);      // empty slots carry (-inf, kNone): they never displace anything
        t2 = top2_merge_dpp<kDppXor2>(t2);
        t2 = top2_merge_dpp<kDppHalfMirror>(t2);
        if (cp == 0 && gr < a.R)
            a.cand[(int64_t)gr * a.ntiles + blockIdx.x] =
                make_float4(t2.v1, __int_as_float(t2.i1), t2.v2, __int_as_float(t2.i2));
        ICK_STAMP(3, 4);
        if (m0 + 32 < a.R) __syncthreads();       // the exchange buffer is reused by the next block of rows
    }
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

extern "C" int ick_decode_supported(int32_t d, int32_t H, int32_t FF, int32_t S, int32_t max_len) {
    return d > 0 && d % 4 == 0 && d <= kDMax && d >= 64 && H > 0 && H <= kPartsMax && d % H == 0 && d / H <= 32 && FF > 0 &&
           FF % 4 == 0 && FF <= 64 * kPartsMax &&
           S > 0 && S <= kSMax && max_len > 0 && max_len <= kMLMax;
}

extern "C" int ick_decode_beam_supported(int32_t Vx, int32_t beam) {
    if (Vx <= 0 || beam < 1 || beam > kBeamMax) return 0;
    return (int64_t)beam * beam * ceil_div(Vx, kBeamChunk) <= 256 * kBeamCandPerThread;   // candidates the selection holds
}

extern "C" int ick_decode_layers(const ick_decode_ctx* c, int32_t pos, void* stream) {
    ICK_CHECK_ARG(c && c->R > 0 && c->layers > 0 && c->layers <= ICK_MAX_LAYERS && pos >= 0 && pos < c->max_len);
    ICK_CHECK_ARG(ick_decode_supported(c->d, c->H, c->FF, c->S, c->max_len));
    ICK_CHECK_ARG(c->rows_per_sample > 0 && c->R % c->rows_per_sample == 0 && c->R <= 65535);
    ICK_CHECK_ARG(c->x0 && c->xa && c->xb && c->xc && c->p1 && c->p2 && c->p3 && c->hfin && c->hv && c->ptr && c->cand);
    ICK_CHECK_ARG(c->H <= kPartsMax && ceil_div(c->FF, 64) <= kPartsMax);
    hipStream_t s = (hipStream_t)stream;
    const int d = c->d, H = c->H, dh = d / H, R = c->R;
    const int nch = ceil_div(c->FF, 64);
    const float scale = 1.f / sqrtf((float)dh);
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
        hipLaunchKernelGGL(dec_self_kernel, dim3(H, R), dim3(256), 0, s, sa);
        CrossArgs ca;
        ca.w.in_w = w.ca_in_w; ca.w.in_b = w.ca_in_b; ca.w.out_wt = w.ca_out_wt;
        ca.w.src = make_src(c, c->xa, c->p1, H, w.sa_out_b, w.n1_g, w.n1_b, c->xb);
        ca.w.part = c->p2;
        ca.Kmem = w.cross_k; ca.Vmem = w.cross_v; ca.kv_bs = c->kv_bs;
        ca.R = R; ca.rows_per_sample = c->rows_per_sample; ca.d = d; ca.H = H; ca.dh = dh; ca.S = c->S; ca.scale = scale;
        ca.n_done = c->n_done; ca.n_total = R;
        hipLaunchKernelGGL(dec_cross_kernel, dim3(H, R), dim3(256), 0, s, ca);
        FfnArgs fa;
        fa.w1 = w.w1; fa.b1 = w.b1; fa.w2t = w.w2t;
        fa.src = make_src(c, c->xb, c->p2, H, w.ca_out_b, w.n2_g, w.n2_b, c->xc);
        fa.part = c->p3; fa.R = R; fa.d = d; fa.FF = c->FF; fa.n_done = c->n_done; fa.n_total = R;
        hipLaunchKernelGGL(dec_ffn_kernel, dim3(nch, R), dim3(256), 0, s, fa);
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
    hipLaunchKernelGGL(dec_head_kernel, dim3(R), dim3(256), 0, s, ha);
    VocabArgs va;
    va.hv = c->hv; va.wv = c->wv; va.bv = c->bv; va.scores = c->scores; va.ld = c->scores_ld;
    va.cand = reinterpret_cast<float4*>(c->cand); va.R = R; va.d = d; va.V = c->V; va.ntiles = ceil_div(c->V, 16);
    va.n_done = c->n_done; va.n_total = R;
    hipLaunchKernelGGL(dec_vocab_kernel, dim3(va.ntiles), dim3(256), 0, s, va);
    ICK_LAUNCH_RET();
}

extern "C" int ick_decode_select_greedy(const ick_decode_ctx* c, int32_t pos, void* stream) {
    ICK_CHECK_ARG(c && c->R > 0 && pos >= 0 && pos < c->max_len);
    ICK_CHECK_ARG(c->output && c->hist && c->finished && c->n_done && c->next_token && c->next_mask && c->word_emb &&
                  c->pe && c->x0 && c->cand && c->ptr && c->ee);
    SelectArgs a;
    a.cand = reinterpret_cast<const float4*>(c->cand); a.ntiles = ceil_div(c->V, 16); a.ptr = c->ptr;
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
