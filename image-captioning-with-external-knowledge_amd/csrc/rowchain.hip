// Row-resident chains of the post-LN Transformer blocks: everything between two attention kernels acts on
// one row at a time, so a workgroup that keeps 8 rows in LDS can run
//     o = A W1^T + b1 ;  x = LayerNorm(res + dropout(o)) gamma + beta ;  y2 = act(x W2^T + b2)
// (out-projection -> add & norm -> the next Linear) as one launch instead of three.  Replaces the call
// sequence torch's Transformer{De,En}coderLayer.forward makes for the layers the reference builds at
// geo-aware/models.py:241-244 (self_attn.out_proj / dropout1 / norm1 / multihead_attn q-projection, ... /
// linear1, linear2 / dropout3 / norm3 / the next layer's in_proj).
//
// gfx950 design.  The chained launches were latency-bound (launch ramp + first dependent HBM round trip of
// every kernel, ~12-16 us each inside the step); here only the weights stream, through the CU's L2 port:
//   * 8 rows per workgroup -> 160 workgroups for the 1280 rows of a cfg2 batch; 16 waves.
//   * v_mfma_f32_4x4x1_16b_f32 with the A operand broadcast (cbsz = 4): one instruction multiplies 4 rows by
//     64 output columns for one k; the A register holds X[4 rows][16 k] (lane = 4 * k + row) and abid walks
//     the 16 k; the B register is W[64 consecutive output columns][k], straight from global memory into the
//     MFMA operand (weights are used once per workgroup: no LDS staging).  Same 64 FLOP/clk/SIMD as the
//     16x16x4 instruction, but 8 instead of 16 rows fill it.
//   * the vector memory pipe moves 4 lanes per clock whatever the access width, so dword loads top out at
//     16 B/clk per CU (measured: 22 cycles per MFMA with one dword load per k); the weights are therefore kept
//     in a PACKED copy, [slab of 64 columns][k / 4][column][k % 4]: lane l of the slab's wave fetches
//     W[64 s + l][4 j .. 4 j + 3] as one dwordx4 of a fully coalesced 1 KiB wave access.
//   * a wave owns a 64-column slab and a K range (slabs x K splits <= 16 waves); K-split partials are summed
//     in a fixed order through LDS (deterministic); the LayerNorm is one wave per row with the arithmetic of
//     add_layernorm_kernel (layernorm.hip), so the normalised rows equal the unfused path's given the same o.
//   * ick_pack_weights (below) refreshes the packed copies once per optimizer step (zero padded to 64 columns
//     and 16 k, so the kernel needs no edge handling).
#include <cstdlib>

#include "common.h"

namespace ick {
namespace {

constexpr int kRows = 8;          // rows per workgroup
constexpr int kWaves = 16;
constexpr int kThreads = kWaves * 64;
constexpr int kMaxK = 512;        // widest GEMM input (linear2: dim_feedforward)
constexpr int kLdx = kMaxK + 16 + 4;
constexpr int kMaxD = 320;        // LayerNorm width (5 columns per lane)
constexpr int kMaxN2 = 1024;      // widest second GEMM (in_proj: 3 d)
constexpr int kPartFloats = kRows * 64 * kWaves;   // every (slab, K split) pair is one wave: 8 rows x 64 columns each

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

struct GemmPlan {   // how the 16 waves cover N columns x K
    int nslab, splits, kper;
};
__host__ __device__ inline GemmPlan plan_for(int N, int K) {
    GemmPlan g;
    g.nslab = (N + 63) / 64;
    g.splits = kWaves / g.nslab;
    if (g.splits < 1) g.splits = 1;
    const int maxs = (K + 15) / 16;
    if (g.splits > maxs) g.splits = maxs;
    if (g.splits > 4) g.splits = 4;
    g.kper = (((K + g.splits - 1) / g.splits) + 15) / 16 * 16;
    return g;
}

#define ICK_MF(U)                                                                            \
    acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a0, b[(U) >> 2][(U) & 3], acc0, 4, U, 0);        \
    acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a1, b[(U) >> 2][(U) & 3], acc1, 4, U, 0);

// acc[row] (lane = column of the wave's slab) = Xs[row][k range of the wave's split] . W[col][k] from the packed copy
// Wp; Xs is zero beyond K.  Returns false for a wave without work (more waves than slabs x splits).
struct Slab { int slab, h; };
__device__ __forceinline__ Slab slab_of(const GemmPlan& g) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar: K offsets stay in SGPRs
    return Slab{wave % g.nslab, wave / g.nslab};
}
template <int DBG>
__device__ __forceinline__ void row_gemm(const float* Xs, int K, const float* __restrict__ Wp, const GemmPlan g,
                                         const Slab w, f32x4& acc0, f32x4& acc1) {
    const int lane = threadIdx.x & 63;
    acc0 = f32x4{0.f, 0.f, 0.f, 0.f};
    acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
    const int kb = w.h * g.kper;
    const int ke = min(K, kb + g.kper);
    if (w.h >= g.splits || kb >= ke) return;
    const int K16 = (K + 15) & ~15;
    const int slab_bytes = K16 * 64 * 4;                 // one slab of the packed copy
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(Wp) + (size_t)w.slab * K16 * 64, (short)0, slab_bytes, 0x00020000);
    const int voff = lane * 16;
    const int nchunk = (ke - kb + 15) >> 4;
    const float* xa = Xs + (lane & 3) * kLdx + (lane >> 2);
    f32x4 bq[3][4];
    // Software pipeline, two chunks ahead, without branches around the loads (the compiler's s_waitcnt counting
    // only stays exact in straight-line code): a chunk beyond this wave's K range is fetched from beyond the
    // descriptor's extent (no memory access, zeros).
    auto load = [&](f32x4 (&b)[4], int c) {
        const int base = c < nchunk ? (kb + 16 * c) * 256 : slab_bytes;      // scalar; 16 k = 4 KiB
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (DBG == 1 || DBG == 3) b[j] = f32x4{(float)base, 1.f, 2.f, (float)j};
            else b[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, base + j * 1024, 0));
        }
    };
    auto mfma = [&](const f32x4 (&b)[4], int c) {
        const int k0 = kb + 16 * c;
        const float a0 = xa[k0], a1 = xa[4 * kLdx + k0];
        if (DBG == 2 || DBG == 3) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { acc0 += b[j] * a0; acc1 += b[j] * a1; }
            return;
        }
        ICK_MF(0) ICK_MF(1) ICK_MF(2) ICK_MF(3) ICK_MF(4) ICK_MF(5) ICK_MF(6) ICK_MF(7)
        ICK_MF(8) ICK_MF(9) ICK_MF(10) ICK_MF(11) ICK_MF(12) ICK_MF(13) ICK_MF(14) ICK_MF(15)
    };
    // sched_barrier: the machine scheduler otherwise sinks the prefetches down to their uses (vmcnt(0) per chunk)
#define ICK_STEP(LD, LC, MF, MC)              \
    load(bq[LD], LC);                        \
    __builtin_amdgcn_sched_barrier(0);       \
    mfma(bq[MF], MC);                        \
    __builtin_amdgcn_sched_barrier(0);
    load(bq[0], 0);
    load(bq[1], 1);
    __builtin_amdgcn_sched_barrier(0);
    int c = 0;
    for (; c + 3 <= nchunk; c += 3) {
        ICK_STEP(2, c + 2, 0, c)
        ICK_STEP(0, c + 3, 1, c + 1)
        ICK_STEP(1, c + 4, 2, c + 2)
    }
    if (c < nchunk) {          // one or two chunks left, already in flight
        mfma(bq[0], c);
        __builtin_amdgcn_sched_barrier(0);
        if (c + 1 < nchunk) mfma(bq[1], c + 1);
    }
#undef ICK_STEP
}

// Row offsets of the 8 rows of a workgroup under the (grp, gs, rs) addressing, without a division per row.
struct RowOff {
    int g, i, grp;
    int64_t gs, rs;
    __device__ __forceinline__ RowOff(int row0, int grp_, int64_t gs_, int64_t rs_) : grp(grp_), gs(gs_), rs(rs_) {
        if (grp > 0) { g = row0 / grp; i = row0 - g * grp; } else { g = 0; i = row0; }
    }
    __device__ __forceinline__ int64_t next() {    // offset of the current row; advances to the following one
        const int64_t o = (int64_t)g * gs + (int64_t)i * rs;
        ++i;
        if (grp > 0 && i >= grp) { i = 0; ++g; }
        return o;
    }
};

template <int DBG>
__global__ __launch_bounds__(kThreads) void rowchain_fwd_kernel(ick_rowchain_args p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xs = smem;                       // [8][kLdx]   GEMM input rows (A, then the normalised rows)
    float* Ps = smem + kRows * kLdx;        // K-split partials
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row0 = blockIdx.x * kRows;
    const int d = p.d, M = p.M;
    // Thread mappings are chosen so that nothing below divides per element: at 16 waves a VALU instruction of
    // every thread costs 16 cycles of the CU, a 32-bit division ~40 of them.

    // ---- the residual row, gamma / beta / bias of the LayerNorm wave: issued first, consumed after GEMM 1
    const bool ln_wave = wave < kRows;
    const int lrow = row0 + wave;
    const bool lrow_ok = ln_wave && lrow < M;
    float rres[5], rg[5], rb[5], rbias[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int c = lane + 64 * j;
        const bool ok = lrow_ok && c < d;
        rres[j] = ok && p.res ? p.res[(int64_t)lrow * p.res_rs + c] : 0.f;
        rg[j] = ok ? p.gamma[c] : 0.f;
        rb[j] = ok ? p.beta[c] : 0.f;
        rbias[j] = ok && p.b1 ? p.b1[c] : 0.f;
    }
    // ---- A rows -> LDS (zero beyond K1 up to the next multiple of 16, zero rows beyond M): wave = (row, half)
    const int K1 = p.K1, K1p = (K1 + 15) & ~15;
    {
        const int r = wave & (kRows - 1), half = wave >> 3;
        const int gr = row0 + r;
        int64_t off = (int64_t)gr * p.a_rs;
        if (p.a_grp > 0) { const int g = gr / p.a_grp; off = (int64_t)g * p.a_gs + (int64_t)(gr - g * p.a_grp) * p.a_rs; }
        const float* arow = p.A + off;
        for (int k = lane + 64 * half; k < K1p; k += 128) Xs[r * kLdx + k] = (gr < M && k < K1) ? arow[k] : 0.f;
    }
    __syncthreads();
    const GemmPlan g1 = plan_for(d, K1);
    {
        const Slab w = slab_of(g1);
        f32x4 acc0, acc1;
        row_gemm<DBG>(Xs, K1, p.w1p, g1, w, acc0, acc1);
        if (w.h < g1.splits) {
            float* q = Ps + (size_t)w.h * kRows * (g1.nslab * 64) + w.slab * 64 + lane;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                q[i * g1.nslab * 64] = acc0[i];
                q[(4 + i) * g1.nslab * 64] = acc1[i];
            }
        }
    }
    __syncthreads();
    // ---- o = sum of the K splits + bias; x = LayerNorm(res + dropout(o)); one wave per row
    const int dp = (d + 15) & ~15;
    if (ln_wave) {
        const Dropout drop = make_dropout(p.drop1_p, p.drop_epoch ? p.drop_seed + *p.drop_epoch : p.drop_seed, p.drop1_site);
        const int npad = g1.nslab * 64;
        float v[5];
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int c = lane + 64 * j;
            float t = 0.f;
            if (c < d) {
                float o = Ps[wave * npad + c];
                for (int h = 1; h < g1.splits; ++h) o += Ps[(h * kRows + wave) * npad + c];
                o += rbias[j];
                if (p.o && lrow_ok) p.o[(int64_t)lrow * p.o_rs + c] = o;
                t = o;
                if (drop.on()) t *= drop.mask((uint32_t)lrow * (uint32_t)d + (uint32_t)c);
                t += rres[j];
            }
            v[j] = t;
            sum += t;
        }
        const float mean = wave_sum(sum) / (float)d;
        float var = 0.f;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int c = lane + 64 * j;
            const float t = c < d ? v[j] - mean : 0.f;
            var = fmaf(t, t, var);
        }
        const float rstd = rsqrtf(wave_sum(var) / (float)d + p.eps);
        float* xr = nullptr;
        if (lrow_ok) {
            int64_t off = (int64_t)lrow * p.x_rs;
            if (p.x_grp > 0) { const int g = lrow / p.x_grp; off = (int64_t)g * p.x_gs + (int64_t)(lrow - g * p.x_grp) * p.x_rs; }
            xr = p.x + off;
        }
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int c = lane + 64 * j;
            if (c < dp) {
                const float y = c < d ? (v[j] - mean) * rstd * rg[j] + rb[j] : 0.f;
                Xs[wave * kLdx + c] = lrow_ok ? y : 0.f;
                if (xr && c < d) xr[c] = y;
            }
        }
        if (p.mean && lrow_ok && lane == 0) {
            p.mean[lrow] = mean;
            p.rstd[lrow] = rstd;
        }
    }
    if (p.w2p == nullptr) return;     // uniform
    __syncthreads();
    // ---- y2 = act(x W2^T + b2), dropout: the K split 0 wave of a slab adds the other splits' partials (LDS) to its
    // accumulators and stores its 64 columns of the 8 rows -- plain rows or the head-split scatter of ick_gemm
    const int N2 = p.N2;
    const GemmPlan g2 = plan_for(N2, d);
    const Slab w = slab_of(g2);
    f32x4 acc0, acc1;
    row_gemm<DBG>(Xs, d, p.w2p, g2, w, acc0, acc1);
    const int npad = g2.nslab * 64;
    if (g2.splits > 1) {
        if (w.h > 0 && w.h < g2.splits) {
            float* q = Ps + (size_t)(w.h - 1) * kRows * npad + w.slab * 64 + lane;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                q[i * npad] = acc0[i];
                q[(4 + i) * npad] = acc1[i];
            }
        }
        __syncthreads();
    }
    if (w.h != 0) return;
    const int col = w.slab * 64 + lane;
    if (col >= N2) return;
    float y[kRows];
#pragma unroll
    for (int i = 0; i < 4; ++i) { y[i] = acc0[i]; y[4 + i] = acc1[i]; }
    for (int h = 1; h < g2.splits; ++h) {
        const float* q = Ps + (size_t)(h - 1) * kRows * npad + col;
#pragma unroll
        for (int i = 0; i < kRows; ++i) y[i] += q[i * npad];
    }
    const float bias = p.b2 ? p.b2[col] : 0.f;
    const Dropout drop = make_dropout(p.drop2_p, p.drop_epoch ? p.drop_seed + *p.drop_epoch : p.drop_seed, p.drop2_site);
    const bool relu = p.flags & ICK_GEMM_RELU;
    const bool hs = p.hs_dh > 0;
    int64_t coff = col;
    if (hs) {
        const int hd = p.hs_H * p.hs_dh;
        const int seg = col / hd, rr = col - seg * hd;
        const int hh = rr / p.hs_dh, jj = rr - hh * p.hs_dh;
        coff = (int64_t)p.hs_s0 * p.hs_dhp + ((int64_t)seg * p.hs_H + hh) * ((int64_t)p.hs_S * p.hs_dhp) + jj;
    }
    RowOff ro(row0, p.y2_grp, p.y2_gs, hs ? (int64_t)p.hs_dhp : p.y2_rs);
#pragma unroll
    for (int i = 0; i < kRows; ++i) {
        const int gr = row0 + i;
        const int64_t roff = ro.next();
        if (gr >= M) break;
        float t = y[i] + bias;
        if (relu) t = fmaxf(t, 0.f);
        if (drop.on()) t *= drop.mask((uint32_t)gr * (uint32_t)N2 + (uint32_t)col);
        p.y2[roff + coff] = t;
    }
}

// Packed weight copies: dst[slab][k4][l][kk] = W[64 slab + l][4 k4 + kk] (zero for rows >= N, k >= K), k4 < K16 / 4.
// A workgroup moves 64 rows x 32 k: 128 contiguous bytes per row in, 1 KiB per (k4, 64 rows) out.
constexpr int kPackMax = 48;
struct PackBatch {
    ick_pack_item it[kPackMax];
    int tile_end[kPackMax];
    int n;
};
__global__ __launch_bounds__(512) void pack_weights_kernel(PackBatch pb) {
    __shared__ float tile[64][33];
    int which = 0;
    while (which + 1 < pb.n && (int)blockIdx.x >= pb.tile_end[which]) ++which;
    const ick_pack_item m = pb.it[which];
    const int local = blockIdx.x - (which > 0 ? pb.tile_end[which - 1] : 0);
    const int K16 = (m.K + 15) & ~15;
    const int ktiles = (K16 + 31) / 32;
    const int slab = local / ktiles, kt = local - slab * ktiles;
    {   // in: thread (row r = tid / 8, 8 threads x float4 along k)
        const int r = threadIdx.x >> 3, q = threadIdx.x & 7;
        const int n = slab * 64 + r;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = kt * 32 + 4 * q + e;
            tile[r][4 * q + e] = (n < m.N && k < m.K) ? m.src[(int64_t)n * m.src_ld + k] : 0.f;
        }
    }
    __syncthreads();
    {   // out: thread (k4 = tid / 64, column l = tid % 64) writes one float4
        const int j = threadIdx.x >> 6, l = threadIdx.x & 63;
        const int k4 = kt * 8 + j;
        if (4 * k4 < K16) {
            float4 v = make_float4(tile[l][4 * j], tile[l][4 * j + 1], tile[l][4 * j + 2], tile[l][4 * j + 3]);
            *reinterpret_cast<float4*>(m.dst + (((int64_t)slab * (K16 / 4) + k4) * 64 + l) * 4) = v;
        }
    }
}

}  // namespace
}  // namespace ick

extern "C" int ick_packed_weight_floats(int32_t N, int32_t K, int64_t* floats) {
    if (N <= 0 || K <= 0 || !floats) return ICK_EINVAL;
    *floats = (int64_t)((N + 63) / 64) * 64 * ((K + 15) & ~15);
    return ICK_OK;
}

extern "C" int ick_pack_weights(const ick_pack_item* items, int32_t count, void* stream) {
    using namespace ick;
    if (!items || count <= 0 || count > kPackMax) return ICK_EINVAL;
    PackBatch pb;
    int total = 0;
    for (int i = 0; i < count; ++i) {
        const ick_pack_item& m = items[i];
        if (!m.src || !m.dst || m.N <= 0 || m.K <= 0 || m.src_ld < m.K) return ICK_EINVAL;
        if (reinterpret_cast<uintptr_t>(m.dst) & 15) return ICK_EALIGN;
        pb.it[i] = m;
        total += ((m.N + 63) / 64) * ((((m.K + 15) & ~15) + 31) / 32);
        pb.tile_end[i] = total;
    }
    for (int i = count; i < kPackMax; ++i) { pb.it[i] = items[0]; pb.tile_end[i] = total; }
    pb.n = count;
    hipLaunchKernelGGL(pack_weights_kernel, dim3(total), dim3(512), 0, (hipStream_t)stream, pb);
    ICK_LAUNCH_RET();
}

extern "C" int ick_rowchain_supported(int32_t K1, int32_t d, int32_t N2) {
    using namespace ick;
    return K1 > 0 && K1 <= kMaxK && d > 0 && d <= kMaxD && N2 >= 0 && N2 <= kMaxN2;
}

extern "C" int ick_rowchain_fwd(const ick_rowchain_args* in, void* stream) {
    using namespace ick;
    if (!in) return ICK_EINVAL;
    const ick_rowchain_args& a = *in;
    ICK_CHECK_ARG(a.A && a.w1p && a.gamma && a.beta && a.x);
    ICK_CHECK_ARG(a.M > 0 && ick_rowchain_supported(a.K1, a.d, a.w2p ? a.N2 : 0));
    ICK_CHECK_ARG((a.mean == nullptr) == (a.rstd == nullptr));
    if (a.w2p) {
        ICK_CHECK_ARG(a.y2 && a.N2 > 0);
        if (a.hs_dh > 0) {
            ICK_CHECK_ARG(a.hs_dhp >= a.hs_dh && a.hs_H > 0 && a.hs_S > 0 && a.hs_s0 >= 0);
            ICK_CHECK_ARG(a.N2 % (a.hs_H * a.hs_dh) == 0);
            ICK_CHECK_ARG(a.hs_s0 + (a.y2_grp > 0 ? a.y2_grp : a.M) <= a.hs_S);
        }
    }
    constexpr size_t smem = (size_t)(kRows * kLdx + kPartFloats) * sizeof(float);
    static_assert(smem <= 64 * 1024, "needs the large-LDS attribute");
    static int dbg = -1;
    if (dbg < 0) { const char* e = getenv("ICK_RC_DBG"); dbg = e ? atoi(e) : 0; }
    if (dbg == 1) hipLaunchKernelGGL(rowchain_fwd_kernel<1>, dim3(ceil_div(a.M, kRows)), dim3(kThreads), smem, (hipStream_t)stream, a);
    else if (dbg == 3) hipLaunchKernelGGL(rowchain_fwd_kernel<3>, dim3(ceil_div(a.M, kRows)), dim3(kThreads), smem, (hipStream_t)stream, a);
    else if (dbg == 2) hipLaunchKernelGGL(rowchain_fwd_kernel<2>, dim3(ceil_div(a.M, kRows)), dim3(kThreads), smem, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(rowchain_fwd_kernel<0>, dim3(ceil_div(a.M, kRows)), dim3(kThreads), smem, (hipStream_t)stream, a);
    ICK_LAUNCH_RET();
}
