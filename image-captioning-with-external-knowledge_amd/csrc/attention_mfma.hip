// Matrix-core attention for the teacher-forced shapes of the caption decoder (T <= 64 queries,
// S <= 512 keys, dh <= 32, head-major padded operands): forward and backward of
//   O = softmax(Q K^T * scale [+ causal mask]) V            (see include/ick_amd.h: ick_attention)
// on v_mfma_f32_16x16x4_f32 (exact fp32).  One workgroup of four waves per (sample, head); the key
// tiles (16 keys) are dealt round-robin to the waves and nothing is transposed through memory:
// the MFMA output layout (lane <-> column, 4 registers <-> 4 consecutive rows) of one product is
// exactly the k-major operand layout of the next one.
//
//   operand fragments (lane l: i = l & 15, q = l >> 4; chunk t covers columns 16t .. 16t+15):
//     a row-major matrix X[row][col] serves as A (row = i) or as B (column n = i) of the MFMA with
//     one 16-byte read X[row0 + i][16t + 4q .. +3]; step u of the chunk consumes element u, both
//     operands use the same column order so the sum is the same set of products.
//   result C: lane holds column n = l & 15, rows 4q + r (r = 0..3)  ==  the B operand of a product
//     whose reduction index is that row index (element u <-> row 4q + u).
//
// forward   S^T = K Q^T      (A = K rows from global, B = Q rows from LDS)  -> lane <-> query, regs <-> keys
//           softmax over keys: registers, two lane shuffles, one LDS exchange between the waves
//           O^T = V^T P^T    (A = V^T from LDS [col][key], B = the P registers) -> lane <-> query, regs <-> 4 cols
// backward  S = Q K^T, dP = dO V^T   (A = Q / dO rows from LDS, B = K / V rows from global) -> lane <-> key
//           dV^T = dO^T P, dK^T = Q^T dS (A = dO^T / Q^T from LDS, B = the P / dS registers)
//           dS goes to LDS once ([query][key]); dQ = dS K with A = dS rows from LDS, B = K columns from global (L2).
#include <algorithm>

#include "attention_mfma.h"

// Diagnostic build (-DICK_ATTN_STAMPS, tools/debug/attn_stamps.py): workgroup (0, 0) of the last launch of each kernel
// variant records the shader clock at its phase boundaries.  Compiled out of the product library.
#ifdef ICK_ATTN_STAMPS
__device__ unsigned long long ick_attn_stamps[4][16];
#define ICK_ASTAMP(kern, i)                                                                   \
    do {                                                                                      \
        if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) ick_attn_stamps[kern][i] = __builtin_readcyclecounter(); \
    } while (0)
extern "C" int ick_debug_read_attn_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ick_attn_stamps), sizeof(ick_attn_stamps));
}
#else
#define ICK_ASTAMP(kern, i)
#endif

namespace ick {
namespace {

constexpr int DHP = 32;          // padded head dimension of the head-major layout
constexpr int QLD = DHP + 4;     // LDS row stride of the row-major Q / dO tiles

__device__ __forceinline__ float4 mask_cols(float4 x, int c0, int dh) {
    x.x = c0 + 0 < dh ? x.x : 0.f; x.y = c0 + 1 < dh ? x.y : 0.f;
    x.z = c0 + 2 < dh ? x.z : 0.f; x.w = c0 + 3 < dh ? x.w : 0.f;
    return x;
}

__device__ __forceinline__ int wave_id_of(unsigned tid) { return __builtin_amdgcn_readfirstlane((int)(tid >> 6)); }

__device__ __forceinline__ f32x4 mfma4(const float4& a, const float4& b, f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, c, 0, 0, 0);
    return c;
}

// Stage rows [0, rows) of a head-major matrix (row stride DHP) as X^T in LDS: dst[col][row], row stride SP,
// columns >= dh and rows in [rows, rows_pad) zeroed.  8 lanes read one 128-byte row (coalesced).  All global
// loads of the thread are issued before the first LDS write (IT = 256-thread passes, compile time), so the
// passes share one memory round trip instead of paying one each.
template <int IT>
__device__ __forceinline__ void stage_transposed_load(const float* __restrict__ src, int rows, float4 (&x)[IT]) {
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int f = threadIdx.x + 256 * it;
        const int s = f >> 3, c = f & 7;
        // unconditional (rows beyond the matrix re-read row 0 and are zeroed at store time): a predicated load
        // would be followed by its own wait
        x[it] = *reinterpret_cast<const float4*>(src + (int64_t)(s < rows ? s : 0) * DHP + 4 * c);
    }
}
template <int IT>
__device__ __forceinline__ void stage_transposed_store(const float4 (&x)[IT], float* __restrict__ dst, int rows,
                                                       int rows_pad, int SP, int dh) {
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int f = threadIdx.x + 256 * it;
        const int s = f >> 3, c = f & 7;
        if (s < rows_pad) {
            const float4 v = mask_cols(x[it], 4 * c, s < rows ? dh : 0);
            dst[(4 * c + 0) * SP + s] = v.x; dst[(4 * c + 1) * SP + s] = v.y;
            dst[(4 * c + 2) * SP + s] = v.z; dst[(4 * c + 3) * SP + s] = v.w;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------
template <int NQT, int MAXT>
__global__ __launch_bounds__(256) void attn_fwd_mfma_kernel(ick_attn_args p, int SP) {
    chain_priority();
    constexpr int NQ = NQT * 16;
    [[maybe_unused]] constexpr int SK = MAXT > 1 ? 1 : 0;      // stamp slot: cross (MAXT > 1) / self
    ICK_ASTAMP(SK, 0);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Vt = smem;                    // max(DHP * SP, 4 waves * NQT * 2 tiles * 256)   V^T: [col][key]
    float* Ored = smem;                  // the partial output tiles reuse V^T's space once every wave is done with it
    float* Qs = Vt + max(DHP * SP, 4 * NQT * 2 * 256);   // NQ * QLD      Q rows (zero padded)
    float* red = Qs + NQ * QLD;          // 4 * NQ       per-wave row statistics
    float* stat = red + 4 * NQ;          // 2 * NQ       final max / 1/sum
    float* rsum = stat + 2 * NQ;         // 4 * NQ       per-wave row sums (red still holds the maxima other waves read)

    const int h = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lq = lane >> 4;
    const int T = p.T, S = p.S, dh = p.dh;
    int slen = S;
    if (p.kv_len) slen = min(S, p.kv_len[b]);
    const int nkt = (S + 15) >> 4;       // key tiles
    const Dropout drop = make_dropout(p.drop_p, epoch_seed(p.drop_seed, p.drop_epoch), p.drop_site);
    const float* qb = p.Q + (int64_t)b * p.q_bs + (int64_t)h * p.q_hs;
    const float* kb = p.K + (int64_t)b * p.k_bs + (int64_t)h * p.k_hs;
    const float* vb = p.V + (int64_t)b * p.v_bs + (int64_t)h * p.v_hs;

    // Every global load of the kernel is requested up front, in the order it is needed: Q (staged first, behind its
    // own barrier), this wave's K fragments (S^T), V (transposed into LDS only after S^T, whose matrix work hides its
    // latency).  One memory round trip on the critical path instead of two.
    constexpr int QIT = (NQ * 8 + 255) / 256;
    float4 qx[QIT];
#pragma unroll
    for (int it = 0; it < QIT; ++it) {
        const int f = tid + 256 * it;
        const int t = f >> 3, c = f & 7;
        qx[it] = *reinterpret_cast<const float4*>(qb + (int64_t)(t < T ? t : 0) * DHP + 4 * c);
    }
    float4 kf[MAXT][2];
#pragma unroll
    for (int i = 0; i < MAXT; ++i) {
        const int key = 16 * (wave + 4 * i) + li;
        const float* kr = kb + (int64_t)(key < slen ? key : 0) * DHP + 4 * lq;
#pragma unroll
        for (int t = 0; t < 2; ++t) kf[i][t] = *reinterpret_cast<const float4*>(kr + 16 * t);
    }
    float4 vx[2 * MAXT];
    stage_transposed_load<2 * MAXT>(vb, slen, vx);
#pragma unroll
    for (int it = 0; it < QIT; ++it) {
        const int f = tid + 256 * it;
        const int t = f >> 3, c = f & 7;
        if (f < NQ * 8) *reinterpret_cast<float4*>(Qs + t * QLD + 4 * c) = mask_cols(qx[it], 4 * c, t < T ? dh : 0);
    }
#pragma unroll
    for (int i = 0; i < MAXT; ++i) {
        const bool ok = 16 * (wave + 4 * i) + li < slen;
#pragma unroll
        for (int t = 0; t < 2; ++t) kf[i][t] = mask_cols(kf[i][t], 16 * t + 4 * lq, ok ? dh : 0);
    }
    ICK_ASTAMP(SK, 1);
    __syncthreads();
    ICK_ASTAMP(SK, 2);
    ICK_ASTAMP(SK, 3);

    // S^T tiles: lane <-> query (16 qt + li), registers <-> keys 16 kt + 4 lq + r
    f32x4 sc[MAXT][NQT];
    float mx[NQT];
#pragma unroll
    for (int qt = 0; qt < NQT; ++qt) {
        mx[qt] = -INFINITY;
        const float4 q0 = *reinterpret_cast<const float4*>(Qs + (16 * qt + li) * QLD + 4 * lq);
        const float4 q1 = *reinterpret_cast<const float4*>(Qs + (16 * qt + li) * QLD + 16 + 4 * lq);
        const int query = 16 * qt + li;
#pragma unroll
        for (int i = 0; i < MAXT; ++i) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            if (wave + 4 * i < nkt) {   // wave-uniform
                acc = mfma4(kf[i][0], q0, acc);
                acc = mfma4(kf[i][1], q1, acc);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = 16 * (wave + 4 * i) + 4 * lq + r;
                float v = acc[r] * p.scale;
                if (key >= slen || (p.causal && key > p.q_pos0 + query)) v = -INFINITY;
                acc[r] = v;
                mx[qt] = fmaxf(mx[qt], v);
            }
            sc[i][qt] = acc;
        }
        mx[qt] = fmaxf(mx[qt], __shfl_xor(mx[qt], 16, 64));
        mx[qt] = fmaxf(mx[qt], __shfl_xor(mx[qt], 32, 64));
        if (lq == 0) red[wave * NQ + query] = mx[qt];
    }
    ICK_ASTAMP(SK, 4);
    stage_transposed_store<2 * MAXT>(vx, Vt, slen, nkt * 16, SP, dh);
    __syncthreads();                     // the waves' row maxima and V^T are in LDS
    if (tid < NQ) stat[tid] = fmaxf(fmaxf(red[tid], red[NQ + tid]), fmaxf(red[2 * NQ + tid], red[3 * NQ + tid]));
    ICK_ASTAMP(SK, 5);

    // P = exp(s - max) (the normaliser keeps every key; attention dropout only thins the numerator)
    float sum[NQT];
#pragma unroll
    for (int qt = 0; qt < NQT; ++qt) {
        const int query = 16 * qt + li;
        const float m = fmaxf(fmaxf(red[query], red[NQ + query]), fmaxf(red[2 * NQ + query], red[3 * NQ + query]));
        const uint32_t rowbase = (uint32_t)((b * p.H + h) * T + query) * (uint32_t)S;
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < MAXT; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = 16 * (wave + 4 * i) + 4 * lq + r;
                const float e = (m == -INFINITY) ? 0.f : __expf(sc[i][qt][r] - m);
                s += e;
                sc[i][qt][r] = drop.on() ? e * drop.mask(rowbase + (uint32_t)key) : e;
            }
        }
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        sum[qt] = s;
    }
    // O^T partial sums over this wave's keys: lane <-> query, registers <-> columns 16 jt + 4 lq + r
    f32x4 oacc[NQT][2];
#pragma unroll
    for (int qt = 0; qt < NQT; ++qt) {
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < MAXT; ++i) {
                if (wave + 4 * i < nkt) {
                    const float4 vt = *reinterpret_cast<const float4*>(Vt + (16 * jt + li) * SP + 16 * (wave + 4 * i) + 4 * lq);
                    const float4 pf = make_float4(sc[i][qt][0], sc[i][qt][1], sc[i][qt][2], sc[i][qt][3]);
                    acc = mfma4(vt, pf, acc);
                }
            }
            oacc[qt][jt] = acc;
        }
        if (lq == 0) rsum[wave * NQ + 16 * qt + li] = sum[qt];
    }
    ICK_ASTAMP(SK, 6);
    __syncthreads();     // every wave is done reading V^T: its space now takes the partial tiles
    ICK_ASTAMP(SK, 7);
#pragma unroll
    for (int qt = 0; qt < NQT; ++qt)
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
            *reinterpret_cast<f32x4*>(Ored + ((wave * NQT + qt) * 2 + jt) * 256 + lane * 4) = oacc[qt][jt];
    if (tid < NQ) {
        const float s = (rsum[tid] + rsum[NQ + tid]) + (rsum[2 * NQ + tid] + rsum[3 * NQ + tid]);
        stat[NQ + tid] = s > 0.f ? 1.f / s : 0.f;
        if (p.lse && tid < T) p.lse[((int64_t)b * p.H + h) * T + tid] = stat[tid] + __logf(s);
    }
    __syncthreads();
    // combine the four partial tiles and store: tile tt = (qt, jt), lane <-> query, 4 consecutive columns
    for (int tt = wave; tt < NQT * 2; tt += 4) {
        const int qt = tt >> 1, jt = tt & 1;
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < 4; ++w) o += *reinterpret_cast<const f32x4*>(Ored + ((w * NQT + qt) * 2 + jt) * 256 + lane * 4);
        const int query = 16 * qt + li;
        if (query < T) {
            const float inv = stat[NQ + query];
            float* orow = p.O + (int64_t)b * p.o_bs + (int64_t)query * p.o_ts + h * dh;
            const int j0 = 16 * jt + 4 * lq;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (j0 + r < dh) orow[j0 + r] = o[r] * inv;
        }
    }
    ICK_ASTAMP(SK, 8);
}

// ---------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------
template <int NQT, int MAXT>
__global__ __launch_bounds__(256) void attn_bwd_mfma_kernel(ick_attn_bwd_args p, int SP, bool st2) {
    chain_priority_bwd();
    [[maybe_unused]] constexpr int SK = MAXT > 1 ? 3 : 2;
    ICK_ASTAMP(SK, 0);
    constexpr int NQ = NQT * 16;
    constexpr int TLD = NQ + 4;          // row stride of the transposed Q / dO tiles ([col][query])
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // dS: [query][key], rows for the T real queries only (16-row padding would cost 12 x SP floats = the third or
    // fourth workgroup per CU); the dQ phase reads "rows" T..NQ-1 out of whatever follows -- they only feed output
    // rows that are never stored
    const int TR = (p.T + 3) & ~3;
    float* dSs = smem;                   // TR * SP
    float* Qs = dSs + TR * SP;           // NQ * QLD
    float* Gs = Qs + NQ * QLD;           // NQ * QLD     dO rows
    float* Qt = Gs + NQ * QLD;           // DHP * TLD
    float* Gt = Qt + DHP * TLD;          // DHP * TLD
    float* Ls = Gt + DHP * TLD;          // NQ  lse
    float* Dl = Ls + NQ;                 // NQ  rowsum(dO * O)
    float* Kw = Dl + NQ + (wave_id_of(threadIdx.x)) * (16 * QLD);   // 4 x 16 x QLD: this wave's current key tile of K
    float* dQp = Qs;                     // after the key-tile loop: the waves' partial dQ tiles (4 x NQT x 2 x 256)

    // grid (B, H): workgroups are dealt to the XCDs round robin in linear order, so with the sample index fastest every
    // head of a sample lands on the same XCD (B a multiple of 8) and the heads' 120-byte pieces of the (rows, 6 d) K / V
    // gradient rows meet in ONE L2 before they are written back -- with the head index fastest they were spread over all
    // eight, and every partial line went to memory by itself (train step 1.676 -> 1.665 ms, profiles/r05_x_ab_attn_bwd_grid.txt)
    const int b = blockIdx.x, h = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lq = lane >> 4;
    const int T = p.T, S = p.S, dh = p.dh;
    const int nkt = (S + 15) >> 4;
    const Dropout drop = make_dropout(p.drop_p, epoch_seed(p.drop_seed, p.drop_epoch), p.drop_site);
    const float* qb = p.Q + (int64_t)b * p.q_bs + (int64_t)h * p.q_hs;
    const float* kb = p.K + (int64_t)b * p.k_bs + (int64_t)h * p.k_hs;
    const float* vb = p.V + (int64_t)b * p.v_bs + (int64_t)h * p.v_hs;
    const float* ob = p.O + (int64_t)b * p.o_bs + h * dh;
    const float* gb = p.dO + (int64_t)b * p.o_bs + h * dh;

    // Every global load is requested up front in the order of its use: what the staging needs first (Q, lse, dO, O:
    // small), then this wave's K / V fragments tile by tile -- the staging and the first key tiles run while the
    // later fragments are still on their way (vmcnt counts in order).
    constexpr int QIT = (NQ * 8 + 255) / 256;
    float4 qx[QIT];
#pragma unroll
    for (int it = 0; it < QIT; ++it) {
        const int f = tid + 256 * it;
        const int t = f >> 3, c = f & 7;
        qx[it] = *reinterpret_cast<const float4*>(qb + (int64_t)(t < T ? t : 0) * DHP + 4 * c);
    }
    const float lse_v = p.lse[((int64_t)b * p.H + h) * T + (tid < T ? tid : 0)];
    float gv[NQT * 2], ov[NQT * 2];
#pragma unroll
    for (int it = 0; it < NQT * 2; ++it) {
        const int f = tid + 256 * it;
        const int t = f >> 5, j = f & 31;
        const bool ok = t < T && j < dh;
        const int64_t at = ok ? (int64_t)t * p.o_ts + j : 0;      // unconditional loads (masked below)
        gv[it] = gb[at];
        ov[it] = ob[at];
    }
    // key tile 16 (wave + 4 i), lane <-> key li, columns 16 t + 4 lq .. +3
    float4 kf[MAXT][2], vf[MAXT][2];
#pragma unroll
    for (int i = 0; i < MAXT; ++i) {
        const int key = 16 * (wave + 4 * i) + li;
        const int64_t ro = (int64_t)(key < S ? key : 0) * DHP + 4 * lq;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            kf[i][t] = *reinterpret_cast<const float4*>(kb + ro + 16 * t);
            vf[i][t] = *reinterpret_cast<const float4*>(vb + ro + 16 * t);
        }
    }
    ICK_ASTAMP(SK, 1);
    // ---- staging: Q (row-major and transposed), dO (both), lse, D
#pragma unroll
    for (int it = 0; it < QIT; ++it) {
        const int f = tid + 256 * it;
        const int t = f >> 3, c = f & 7;
        if (f < NQ * 8) {
            const float4 x = mask_cols(qx[it], 4 * c, t < T ? dh : 0);
            *reinterpret_cast<float4*>(Qs + t * QLD + 4 * c) = x;
            Qt[(4 * c + 0) * TLD + t] = x.x; Qt[(4 * c + 1) * TLD + t] = x.y;
            Qt[(4 * c + 2) * TLD + t] = x.z; Qt[(4 * c + 3) * TLD + t] = x.w;
        }
    }
#pragma unroll
    for (int it = 0; it < NQT * 2; ++it) {
        const int f = tid + 256 * it;
        const int t = f >> 5, j = f & 31;
        const bool ok = t < T && j < dh;
        const float g = ok ? gv[it] : 0.f, o = ok ? ov[it] : 0.f;
        Gs[t * QLD + j] = g;
        Gt[j * TLD + t] = g;
        // D[t] = sum_j dO[t][j] O[t][j]: the 32 lanes of a row are one half wave
        float d = g * o;
        d += __shfl_xor(d, 1, 64); d += __shfl_xor(d, 2, 64); d += __shfl_xor(d, 4, 64);
        d += __shfl_xor(d, 8, 64); d += __shfl_xor(d, 16, 64);
        if (j == 0) Dl[t] = d;
    }
    if (tid < NQ) Ls[tid] = tid < T ? lse_v : 0.f;
    ICK_ASTAMP(SK, 2);
    __syncthreads();
    ICK_ASTAMP(SK, 3);

    // ---- per key tile: S, dP (lane <-> key 16 kt + li, registers <-> queries 16 qt + 4 lq + r), dV^T, dK^T, and this
    // tile's share of dQ = dS K (A = the dS rows the wave has just written to LDS, B = its K rows through the wave's
    // scratch; lane <-> column 16 jt + li, registers <-> queries): the four waves' partial tiles are summed at the end
    f32x4 dq[NQT][2];
#pragma unroll
    for (int qt = 0; qt < NQT; ++qt)
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) dq[qt][jt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < MAXT; ++i) {
        const int kt = wave + 4 * i;
        if (kt >= nkt) continue;         // wave-uniform
        const int key = 16 * kt + li;
        const bool kok = key < S;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            kf[i][t] = mask_cols(kf[i][t], 16 * t + 4 * lq, kok ? dh : 0);
            vf[i][t] = mask_cols(vf[i][t], 16 * t + 4 * lq, kok ? dh : 0);
        }
        // K rows of this tile for the dQ product below (B operand: k = key, n = column): through the wave's scratch
#pragma unroll
        for (int t = 0; t < 2; ++t) *reinterpret_cast<float4*>(Kw + li * QLD + 16 * t + 4 * lq) = kf[i][t];
        f32x4 dvt[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        f32x4 dkt[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int qt = 0; qt < NQT; ++qt) {
            f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const float4 qa = *reinterpret_cast<const float4*>(Qs + (16 * qt + li) * QLD + 16 * t + 4 * lq);
                const float4 ga = *reinterpret_cast<const float4*>(Gs + (16 * qt + li) * QLD + 16 * t + 4 * lq);
                s = mfma4(qa, kf[i][t], s);
                dp = mfma4(ga, vf[i][t], dp);
            }
            const float4 l4 = *reinterpret_cast<const float4*>(Ls + 16 * qt + 4 * lq);
            const float4 d4 = *reinterpret_cast<const float4*>(Dl + 16 * qt + 4 * lq);
            const float lr[4] = {l4.x, l4.y, l4.z, l4.w}, dr[4] = {d4.x, d4.y, d4.z, d4.w};
            float pd[4], ds[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int query = 16 * qt + 4 * lq + r;
                float pr = __expf(s[r] * p.scale - lr[r]);
                if (!kok || query >= T || (p.causal && key > p.q_pos0 + query)) pr = 0.f;
                float mk = 1.f;
                if (drop.on()) mk = drop.mask((uint32_t)((b * p.H + h) * T + query) * (uint32_t)S + (uint32_t)key);
                ds[r] = pr * (dp[r] * mk - dr[r]) * p.scale;
                pd[r] = pr * mk;
                if (query < T) dSs[query * SP + key] = ds[r];
            }
            const float4 pf = make_float4(pd[0], pd[1], pd[2], pd[3]);
            const float4 df = make_float4(ds[0], ds[1], ds[2], ds[3]);
#pragma unroll
            for (int jt = 0; jt < 2; ++jt) {
                const float4 gt = *reinterpret_cast<const float4*>(Gt + (16 * jt + li) * TLD + 16 * qt + 4 * lq);
                const float4 qt4 = *reinterpret_cast<const float4*>(Qt + (16 * jt + li) * TLD + 16 * qt + 4 * lq);
                dvt[jt] = mfma4(gt, pf, dvt[jt]);
                dkt[jt] = mfma4(qt4, df, dkt[jt]);
            }
        }
        // dQ += dS(tile) K(tile).  The dS rows and the K scratch were written by other lanes of this wave: LDS
        // operations of one wave execute in order, the fence keeps the compiler from moving the reads up
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        {
            float4 kq[2];
#pragma unroll
            for (int jt = 0; jt < 2; ++jt) {
                const float* kc = Kw + (4 * lq) * QLD + 16 * jt + li;
                kq[jt] = make_float4(kc[0], kc[QLD], kc[2 * QLD], kc[3 * QLD]);
            }
#pragma unroll
            for (int qt = 0; qt < NQT; ++qt) {
                const float4 da = *reinterpret_cast<const float4*>(dSs + (16 * qt + li) * SP + 16 * kt + 4 * lq);
#pragma unroll
                for (int jt = 0; jt < 2; ++jt) dq[qt][jt] = mfma4(da, kq[jt], dq[qt][jt]);
            }
        }
        __builtin_amdgcn_wave_barrier();     // the next tile overwrites the scratch
        // dV^T / dK^T tiles: lane <-> key, registers <-> columns 16 jt + 4 lq + r (4 consecutive floats of a row):
        // two 8-byte stores per tile where the layout allows (row-major gradients at even offsets), else scalars
        if (kok) {
            float* dvr = p.dV + (int64_t)b * p.dv_bs + (int64_t)key * p.dv_ss + h * dh;
            float* dkr = p.dK + (int64_t)b * p.dk_bs + (int64_t)key * p.dk_ss + h * dh;
#pragma unroll
            for (int jt = 0; jt < 2; ++jt) {
                const int j0 = 16 * jt + 4 * lq;
                if (st2) {
                    if (j0 + 1 < dh) {
                        *reinterpret_cast<float2*>(dvr + j0) = make_float2(dvt[jt][0], dvt[jt][1]);
                        *reinterpret_cast<float2*>(dkr + j0) = make_float2(dkt[jt][0], dkt[jt][1]);
                    }
                    if (j0 + 3 < dh) {
                        *reinterpret_cast<float2*>(dvr + j0 + 2) = make_float2(dvt[jt][2], dvt[jt][3]);
                        *reinterpret_cast<float2*>(dkr + j0 + 2) = make_float2(dkt[jt][2], dkt[jt][3]);
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (j0 + r < dh) { dvr[j0 + r] = dvt[jt][r]; dkr[j0 + r] = dkt[jt][r]; }
                }
            }
        }
    }
    ICK_ASTAMP(SK, 4);
    __syncthreads();                     // every wave is done with Q / dO in LDS: their space takes the partial dQ tiles
    ICK_ASTAMP(SK, 5);
#pragma unroll
    for (int qt = 0; qt < NQT; ++qt)
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
            *reinterpret_cast<f32x4*>(dQp + ((wave * NQT + qt) * 2 + jt) * 256 + lane * 4) = dq[qt][jt];
    __syncthreads();
    // combine the four partial tiles and store: tile (qt, jt), lane <-> column 16 jt + li, registers <-> queries 4 lq + r
    for (int tt = wave; tt < NQT * 2; tt += 4) {
        const int qt = tt >> 1, jt = tt & 1;
        const int j = 16 * jt + li;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < 4; ++w) acc += *reinterpret_cast<const f32x4*>(dQp + ((w * NQT + qt) * 2 + jt) * 256 + lane * 4);
        if (j < dh) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int query = 16 * qt + 4 * lq + r;
                if (query < T) p.dQ[(int64_t)b * p.dq_bs + (int64_t)query * p.dq_ts + h * dh + j] = acc[r];
            }
        }
    }
    ICK_ASTAMP(SK, 6);
}

inline bool aligned16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

template <int NQT, int MAXT>
int launch_fwd(const ick_attn_args& a, int SP, hipStream_t s) {
    const size_t fl = std::max<size_t>((size_t)DHP * SP, 4 * NQT * 2 * 256) + (size_t)NQT * 16 * QLD + 10 * NQT * 16;
    if (fl * sizeof(float) > 150 * 1024) return kAttnMfmaUnsupported;
    auto kern = attn_fwd_mfma_kernel<NQT, MAXT>;
    static LdsAttrOnce attr;
    if (int e = attr.ensure((const void*)kern, 160 * 1024)) return e;
    hipLaunchKernelGGL(kern, dim3(a.H, a.B), dim3(256), fl * sizeof(float), s, a, SP);
    ICK_LAUNCH_RET();
}

template <int NQT, int MAXT>
int launch_bwd(const ick_attn_bwd_args& a, int SP, hipStream_t s) {
    constexpr int NQ = NQT * 16;
    const size_t tr = (size_t)((a.T + 3) & ~3);
    // the dQ phase reads NQ "rows" of dS: the allocation covers that extent even though only tr rows are written
    // (+ the four waves' K scratch; the partial dQ tiles reuse the Q / dO space: 2048 NQT <= 2176 NQT + 256 floats)
    const size_t fl = std::max(tr * SP + 2 * (size_t)NQ * QLD + 2 * (size_t)DHP * (NQ + 4) + 2 * NQ + 4 * 16 * QLD,
                               (size_t)NQ * SP + 64);
    if (fl * sizeof(float) > 150 * 1024) return kAttnMfmaUnsupported;
    auto kern = attn_bwd_mfma_kernel<NQT, MAXT>;
    static LdsAttrOnce attr;
    if (int e = attr.ensure((const void*)kern, 160 * 1024)) return e;
    // 8-byte stores of dK / dV: every (sample, key, head) row segment starts at an even float offset
    const bool st2 = a.dh % 2 == 0 && a.dk_bs % 2 == 0 && a.dk_ss % 2 == 0 && a.dv_bs % 2 == 0 && a.dv_ss % 2 == 0 &&
                     (reinterpret_cast<uintptr_t>(a.dK) & 7) == 0 && (reinterpret_cast<uintptr_t>(a.dV) & 7) == 0;
    hipLaunchKernelGGL(kern, dim3(a.B, a.H), dim3(256), fl * sizeof(float), s, a, SP, st2);      // sample index fastest: see the kernel
    ICK_LAUNCH_RET();
}

}  // namespace

bool attn_mfma_shape_ok(int T, int S, int dh) {
    if (!(T >= 2 && T <= 64 && S >= 1 && S <= 512 && dh <= DHP)) return false;
    // the backward keeps dS ([query][key]) in LDS
    const size_t nq = (size_t)((T + 15) / 16) * 16, sp = (size_t)((S + 15) / 16) * 16 + 4, tr = (size_t)((T + 3) & ~3);
    return std::max(tr * sp + 2 * nq * QLD + 2 * DHP * (nq + 4) + 2 * nq + 4 * 16 * QLD, nq * sp + 64) * sizeof(float) <= 150 * 1024;
}

int launch_attn_mfma(const ick_attn_args& a, hipStream_t s) {
    if (!attn_mfma_shape_ok(a.T, a.S, a.dh)) return kAttnMfmaUnsupported;
    if (!(a.q_ts == DHP && a.k_ss == DHP && a.v_ss == DHP && aligned16(a.Q) && aligned16(a.K) && aligned16(a.V) &&
          a.q_bs % 4 == 0 && a.q_hs % 4 == 0 && a.k_bs % 4 == 0 && a.k_hs % 4 == 0 && a.v_bs % 4 == 0 && a.v_hs % 4 == 0))
        return kAttnMfmaUnsupported;
    const int SP = ((a.S + 15) / 16) * 16 + 4;
    const int nqt = (a.T + 15) / 16;
    // key tiles per wave (compile time: register arrays): 1 for S <= 64 (self-attention), 4 up to 256 (geo: 216),
    // 5 up to 320 (knowledge: 267), 8 up to 512
    const int mt = a.S <= 64 ? 1 : (a.S <= 256 ? 4 : (a.S <= 320 ? 5 : 8));
#define ICK_FWD(N)                                                                                   \
    case N: return mt == 1 ? launch_fwd<N, 1>(a, SP, s) : (mt == 4 ? launch_fwd<N, 4>(a, SP, s) :   \
                   (mt == 5 ? launch_fwd<N, 5>(a, SP, s) : launch_fwd<N, 8>(a, SP, s)))
    switch (nqt) {
        ICK_FWD(1); ICK_FWD(2); ICK_FWD(3); ICK_FWD(4);
    }
#undef ICK_FWD
    return kAttnMfmaUnsupported;
}

int launch_attn_bwd_mfma(const ick_attn_bwd_args& a, hipStream_t s) {
    if (!attn_mfma_shape_ok(a.T, a.S, a.dh)) return kAttnMfmaUnsupported;
    const int SP = ((a.S + 15) / 16) * 16 + 4;
    const int mt = a.S <= 64 ? 1 : (a.S <= 256 ? 4 : (a.S <= 320 ? 5 : 8));
#define ICK_BWD(N)                                                                                   \
    case N: return mt == 1 ? launch_bwd<N, 1>(a, SP, s) : (mt == 4 ? launch_bwd<N, 4>(a, SP, s) :   \
                   (mt == 5 ? launch_bwd<N, 5>(a, SP, s) : launch_bwd<N, 8>(a, SP, s)))
    switch ((a.T + 15) / 16) {
        ICK_BWD(1); ICK_BWD(2); ICK_BWD(3); ICK_BWD(4);
    }
#undef ICK_BWD
    return kAttnMfmaUnsupported;
}

}  // namespace ick
