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

#include "rowchain.h"

// Diagnostic build (-DICK_CHAIN_STAMPS, tools/debug/chain_stamps.py): thread 0 of workgroup 0 records the shader clock
// at the phase boundaries of the kernel.  Compiled out of the product library.
#ifdef ICK_CHAIN_STAMPS
__device__ unsigned long long ick_chain_stamps[16];
#define ICK_CSTAMP(i)                                                                                     \
    do {                                                                                                  \
        if (threadIdx.x == 0 && blockIdx.x == 0) ick_chain_stamps[i] = __builtin_amdgcn_s_memtime();      \
    } while (0)
extern "C" int ick_debug_read_chain_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ick_chain_stamps), sizeof(ick_chain_stamps));
}
#else
#define ICK_CSTAMP(i)
#endif

namespace ick {
namespace {

using namespace rowchain;

// NW = waves per workgroup.  The 16 (slab, K split) units of a GEMM stage are dealt to the waves round robin: one per
// wave at NW = 16, two at NW = 8.  The 8-wave form does the same matrix work per SIMD, but half the wave slots and
// registers: it still finds room on a CU that already hosts two 8-wave GEMM workgroups (Encoder.conv1) or after a
// single 4-wave one retires, where the 16-wave form waits for the bulk kernel to drain (measured: 137 us for a 25 us
// launch beside the image K/V projection) -- callers pass ICK_CHAIN_SLIM for chains that run beside bulk GEMMs.
template <int NW>
__global__ __launch_bounds__(NW * 64) void rowchain_fwd_kernel(ick_rowchain_args p) {
    chain_priority();
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xs = smem;                       // [8][kLdx]   GEMM input rows (A, then the normalised rows)
    float* Ps = smem + kRows * kLdx;        // K-split partials
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row0 = blockIdx.x * kRows;
    const int d = p.d, M = p.M;
    // Thread mappings are chosen so that nothing below divides per element: at 16 waves a VALU instruction of
    // every thread costs 16 cycles of the CU, a 32-bit division ~40 of them.
    ICK_CSTAMP(0);
    const uint32_t seed = epoch_seed(p.drop_seed, p.drop_epoch);
    // ICK_CHAIN_PROJ: projection only -- y2 = act(A W2^T + b2) with A (M, d) as the rows GEMM 2 multiplies: the first
    // in_proj of a stack as a 160-workgroup priority launch instead of a 1 160-workgroup generic GEMM that queues behind
    // the bulk kernels of the other stream
    const bool proj = (p.flags & ICK_CHAIN_PROJ) != 0;       // uniform
    // ---- the residual row, gamma / beta / bias of the LayerNorm wave: issued first, consumed after GEMM 1
    const bool ln_wave = wave < kRows && !proj;
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
    ICK_CSTAMP(1);
    const int K1 = p.K1, K1p = (K1 + 15) & ~15;
    const GemmPlan g1 = plan_for(d, K1);
    const int units1 = proj ? 0 : g1.nslab * g1.splits;
    RowGemm mm;
    if (!proj) mm.begin(K1, p.w1p, g1, unit_of(g1, wave));       // weights of GEMM 1 start streaming before the rows arrive
    // ---- A rows -> LDS (zero beyond K1 up to the next multiple of 16, zero rows beyond M): wave = (row, part)
    {
        const int r = wave & (kRows - 1), part = wave >> 3;
        const int gr = row0 + r;
        int64_t off = (int64_t)gr * p.a_rs;
        if (p.a_grp > 0) { const int g = small_div(gr, p.a_grp); off = (int64_t)g * p.a_gs + (int64_t)(gr - g * p.a_grp) * p.a_rs; }
        const float* arow = p.A + off;
        // every load of the row is issued before the first LDS store waits for one (as a loop the compiler put a
        // vmcnt(0) in front of every store: one memory round trip per 128 columns, and the first of them also waited
        // for the weight chunks requested above)
        constexpr int NA = (kMaxK + 16 + 8 * NW - 1) / (8 * NW);
        float av[NA];
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int k = lane + 64 * part + j * 8 * NW;
            av[j] = (gr < M && k < K1) ? arow[k] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int k = lane + 64 * part + j * 8 * NW;
            if (k < K1p) Xs[r * kLdx + k] = av[j];
        }
    }
    ICK_CSTAMP(2);
    if (!proj) __syncthreads();      // (projection only: the barrier in front of GEMM 2 below covers the rows)
    ICK_CSTAMP(3);
    for (int u = wave; u < units1; u += NW) {
        const Slab w = unit_of(g1, u);
        if (u != wave) mm.begin(K1, p.w1p, g1, w);
        f32x4 acc0, acc1;
        mm.run(Xs, kLdx, acc0, acc1);
        float* q = Ps + (size_t)w.h * kRows * (g1.nslab * 64) + w.slab * 64 + lane;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            q[i * g1.nslab * 64] = acc0[i];
            q[(4 + i) * g1.nslab * 64] = acc1[i];
        }
    }
    ICK_CSTAMP(4);
    if (!proj) __syncthreads();
    ICK_CSTAMP(5);
    const int N2 = p.N2;
    const GemmPlan g2 = plan_for(max(N2, 1), d);
    const int units2 = p.w2p != nullptr ? g2.nslab * g2.splits : 0;
    // The LayerNorm wave's operands were requested at the head of the kernel and have long arrived, but behind the K loop
    // of GEMM 1 the compiler can only wait for them with vmcnt(0) -- which, placed inside the LayerNorm, would also wait
    // for GEMM 2's first weight chunks requested right below.  Consuming them HERE puts that wait in front of the request:
    // the norm then runs while the chunks are on their way (forward pass 0.715 -> 0.711 ms; the train step is level).
#pragma unroll
    for (int j = 0; j < 5; ++j) asm volatile("" : "+v"(rres[j]), "+v"(rg[j]), "+v"(rb[j]), "+v"(rbias[j]));
    if (wave < units2) mm.begin(d, p.w2p, g2, unit_of(g2, wave));    // ... and those of GEMM 2 behind the LayerNorm
    // ---- o = sum of the K splits + bias; x = LayerNorm(res + dropout(o)); one wave per row
    const int dp = (d + 15) & ~15;
    if (ln_wave) {
        const Dropout drop = make_dropout(p.drop1_p, seed, p.drop1_site);
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
            if (p.x_grp > 0) { const int g = small_div(lrow, p.x_grp); off = (int64_t)g * p.x_gs + (int64_t)(lrow - g * p.x_grp) * p.x_rs; }
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
    ICK_CSTAMP(6);
    if (p.w2p == nullptr) return;     // uniform
    __syncthreads();
    ICK_CSTAMP(7);
    // ---- y2 = act(x W2^T + b2), dropout.  A wave whose only unit is a K split 0 keeps the result in its accumulators
    // and adds the other splits' partials (LDS); otherwise every unit goes through LDS and its split 0 owner reads it back
    const int npad = g2.nslab * 64;
    const bool direct = NW >= 16;     // one unit per wave
    f32x4 acc0, acc1;
    for (int u = wave; u < units2; u += NW) {
        const Slab w = unit_of(g2, u);
        if (u != wave) mm.begin(d, p.w2p, g2, w);
        mm.run(Xs, kLdx, acc0, acc1);
        if (!(direct && w.h == 0)) {
            float* q = Ps + (size_t)w.h * kRows * npad + w.slab * 64 + lane;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                q[i * npad] = acc0[i];
                q[(4 + i) * npad] = acc1[i];
            }
        }
    }
    ICK_CSTAMP(8);
    if (g2.splits > 1 || !direct) __syncthreads();
    ICK_CSTAMP(9);
    const Dropout drop = make_dropout(p.drop2_p, seed, p.drop2_site);
    const bool relu = p.flags & ICK_GEMM_RELU;
    const bool hs = p.hs_dh > 0;
    for (int u = wave; u < units2; u += NW) {
        const Slab w = unit_of(g2, u);
        const int col2 = w.slab * 64 + lane;
        if (w.h != 0 || col2 >= N2) continue;
        float y[kRows];
        if (direct) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { y[i] = acc0[i]; y[4 + i] = acc1[i]; }
        } else {
#pragma unroll
            for (int i = 0; i < kRows; ++i) y[i] = Ps[i * npad + col2];
        }
        for (int h = 1; h < g2.splits; ++h) {
            const float* q = Ps + (size_t)h * kRows * npad + col2;
#pragma unroll
            for (int i = 0; i < kRows; ++i) y[i] += q[i * npad];
        }
        const float bias2 = p.b2 ? p.b2[col2] : 0.f;
        int64_t coff2 = col2;
        if (hs) {
            const int hd = p.hs_H * p.hs_dh;
            const int seg = small_div(col2, hd), rr = col2 - seg * hd;
            const int hh = small_div(rr, p.hs_dh), jj = rr - hh * p.hs_dh;
            coff2 = (int64_t)p.hs_s0 * p.hs_dhp + ((int64_t)seg * p.hs_H + hh) * ((int64_t)p.hs_S * p.hs_dhp) + jj;
        }
        RowOff ro(row0, p.y2_grp, p.y2_gs, hs ? (int64_t)p.hs_dhp : p.y2_rs);
#pragma unroll
        for (int i = 0; i < kRows; ++i) {
            const int gr = row0 + i;
            const int64_t roff = ro.next();
            if (gr >= M) break;
            float t = y[i] + bias2;
            if (relu) t = fmaxf(t, 0.f);
            if (drop.on()) t *= drop.mask((uint32_t)gr * (uint32_t)N2 + (uint32_t)col2);
            p.y2[roff + coff2] = t;
        }
    }
    ICK_CSTAMP(10);
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
    if (m.src_cs == 1) {   // in: thread (row r = tid / 8, 8 threads x 4 floats along the contiguous k)
        const int r = threadIdx.x >> 3, q = threadIdx.x & 7;
        const int n = slab * 64 + r;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = kt * 32 + 4 * q + e;
            tile[r][4 * q + e] = (n < m.N && k < m.K) ? m.src[(int64_t)n * m.src_rs + k] : 0.f;
        }
    } else {               // transposed source (rows contiguous): thread (row r = tid % 64, 8 groups x 4 k)
        const int r = threadIdx.x & 63, q = threadIdx.x >> 6;
        const int n = slab * 64 + r;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = kt * 32 + 4 * q + e;
            tile[r][4 * q + e] = (n < m.N && k < m.K) ? m.src[(int64_t)n * m.src_rs + (int64_t)k * m.src_cs] : 0.f;
        }
    }
    __syncthreads();
    if (m.dst_rs > 0) {   // plain copy: thread (row r = tid / 8, 4 floats along k)
        const int r = threadIdx.x >> 3, q = threadIdx.x & 7;
        const int n = slab * 64 + r;
        if (n < m.N) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = kt * 32 + 4 * q + e;
                if (k < m.K) m.dst[(int64_t)n * m.dst_rs + k] = tile[r][4 * q + e];
            }
        }
    } else {   // out: thread (k4 = tid / 64, column l = tid % 64) writes one float4
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
        if (!m.src || !m.dst || m.N <= 0 || m.K <= 0 || m.src_rs <= 0 || m.src_cs <= 0) return ICK_EINVAL;
        if (m.dst_rs < 0 || (m.dst_rs > 0 && m.dst_rs < m.K)) return ICK_EINVAL;
        if (m.dst_rs == 0 && (reinterpret_cast<uintptr_t>(m.dst) & 15)) return ICK_EALIGN;
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
    if (a.flags & ICK_CHAIN_PROJ) ICK_CHECK_ARG(a.A && a.w2p && a.y2 && a.K1 == a.d);
    else ICK_CHECK_ARG(a.A && a.w1p && a.gamma && a.beta && a.x);
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
    const dim3 grid(ceil_div(a.M, kRows));
    if (a.flags & ICK_CHAIN_SLIM)
        hipLaunchKernelGGL(rowchain_fwd_kernel<8>, grid, dim3(8 * 64), smem, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(rowchain_fwd_kernel<16>, grid, dim3(16 * 64), smem, (hipStream_t)stream, a);
    ICK_LAUNCH_RET();
}
