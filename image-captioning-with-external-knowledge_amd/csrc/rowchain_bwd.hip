// Backward of the row-resident chains (rowchain.hip): between two attention-backward kernels the data-gradient path of
// a post-LN Transformer block acts on one row at a time, so a workgroup that keeps 8 rows in LDS runs
//     dx  = dzin + g0 W0                          data gradient of the Linear that consumed the block's output
//     dz1, do1 = LayerNorm'(dx)                    add & norm backward (gradient of the residual / of the dropped branch)
//     t   = gate(do1 W2) ;  dx2 = dz1 + t W1       linear2 / ReLU / linear1 data gradients       (FFN blocks only)
//     dz2, do2 = LayerNorm'(dx2)                                                                 (FFN blocks only)
//     out = do W3                                  data gradient of the out-projection that fed the norm
// as ONE launch instead of up to six (ick_gemm x4 + ick_layernorm_bwd x2).  That is what autograd runs for
// TransformerDecoderLayer / TransformerEncoderLayer (post-LN) under `loss.backward()` (geo-aware/train.py:284) for the
// layers built at geo-aware/models.py:241-244.  Every intermediate the weight-gradient GEMMs need (do1, t, do2) and
// the per-workgroup gamma / beta partial sums are written exactly where the unfused kernels put them.
//
// Same machinery as the forward chain: 8 rows per workgroup, 16 waves, v_mfma_f32_4x4x1 with broadcast A, packed
// copies of the TRANSPOSED weights streamed straight into the MFMA operand, K-split partials summed in a fixed order
// through LDS, LayerNorm backward one wave per row with the arithmetic of layernorm_bwd_kernel (backward.hip).
#include <cstdlib>

#include "rowchain.h"

namespace ick {
namespace {

using namespace rowchain;

constexpr int kMaxK0 = 1920;                // widest pre-GEMM input (in_proj gradient: 3 d; all-layer cross K/V gradient:
                                            // 2 * layers * d = 1800)
// LDS row stride of the wide input buffer XA, sized by the launch (round 4: a fixed 1924-float stride made every launch
// declare 136 KB, so a workgroup of this latency-bound kernel only found room on a CU that held nothing of the other
// stream's weight-gradient kernels): the widest row it holds -- K0 padded to 16, or the FFN's hidden width padded to
// 64-column slabs -- rounded up to 32 floats, + 4 (the same bank offset between rows as before).
__host__ __device__ __forceinline__ int lda_for(int K0, int N1) {
    const int w = max(max((K0 + 15) & ~15, ((N1 + 63) / 64) * 64), 64);
    return ((w + 31) & ~31) + 4;
}
constexpr int kLdB = kMaxD + 4;             // ... of the d-wide buffers
constexpr int kMaxN1 = 512;                 // dim_feedforward

struct LnIn {      // what one LayerNorm-backward stage reads per row (prefetched into registers at kernel start)
    float o[5], res[5], gamma[5];
    float mean, rstd;
};

__device__ __forceinline__ void ln_prefetch(LnIn& q, const float* o, const float* res, const float* gamma,
                                            const float* mean, const float* rstd, int row, bool ok, int d, int lane) {
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int c = lane + 64 * j;
        const bool v = ok && c < d;
        q.o[j] = v ? o[(int64_t)row * d + c] : 0.f;
        q.res[j] = v && res ? res[(int64_t)row * d + c] : 0.f;
        q.gamma[j] = v ? gamma[c] : 0.f;
    }
    q.mean = ok ? mean[row] : 0.f;
    q.rstd = ok ? rstd[row] : 0.f;
}

// LayerNorm backward of one row held by one wave (lane = column + 64 j): dx in, dz (gradient of the normalised sum) out;
// returns through dgam / dbet the row's contributions to the gamma / beta gradients.  Arithmetic of layernorm_bwd_kernel.
__device__ __forceinline__ void ln_bwd_row(const LnIn& q, const float (&dx)[5], const Dropout& drop, int row, int d, int lane,
                                           float (&dz)[5], float (&dod)[5], float (&dgam)[5], float (&dbet)[5]) {
    float zh[5], g[5];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int c = lane + 64 * j;
        float z = q.o[j];
        if (drop.on()) z *= drop.mask((uint32_t)row * (uint32_t)d + (uint32_t)c);
        z += q.res[j];
        zh[j] = c < d ? (z - q.mean) * q.rstd : 0.f;
        g[j] = dx[j] * q.gamma[j];
        s1 += g[j];
        s2 += g[j] * zh[j];
        dgam[j] = dx[j] * zh[j];
        dbet[j] = dx[j];
    }
    s1 = wave_sum(s1) / (float)d;
    s2 = wave_sum(s2) / (float)d;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int c = lane + 64 * j;
        const float v = c < d ? q.rstd * (g[j] - s1 - zh[j] * s2) : 0.f;
        dz[j] = v;
        dod[j] = drop.on() ? v * drop.mask((uint32_t)row * (uint32_t)d + (uint32_t)c) : v;
    }
}

// ICK_CHAIN_BWD_WAVES (build-time experiment): waves per SIMD the register allocation must allow.  4 (the kernel's own 16
// waves) = 117 VGPRs, i.e. 480 of the 512 registers of every SIMD: the workgroup then only starts on a CU that holds
// nothing else; 5 = 96 VGPRs (36 spilled) leaves a quarter of the file to the other stream's workgroups.
#ifdef ICK_CHAIN_BWD_WAVES
#define ICK_CHAIN_BWD_ATTR __attribute__((amdgpu_waves_per_eu(ICK_CHAIN_BWD_WAVES, ICK_CHAIN_BWD_WAVES)))
#else
#define ICK_CHAIN_BWD_ATTR
#endif
__global__ __launch_bounds__(kThreads) ICK_CHAIN_BWD_ATTR void rowchain_bwd_kernel(ick_rowchain_bwd_args p) {
    chain_priority_bwd();
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int kLdA = lda_for(p.g0 != nullptr ? p.K0 : 0, p.w1p != nullptr ? p.N1 : 0);      // uniform
    float* XA = smem;                          // [8][kLdA]  wide GEMM input: g0 rows, later t
    float* XB = XA + kRows * kLdA;             // [8][kLdB]  d-wide GEMM input: do1 / do2
    float* DZ = XB + kRows * kLdB;             // [8][kLdB]  residual-path gradient between the two norms
    float* RG = DZ + kRows * kLdB;             // [8][2][kLdB] per-row gamma / beta contributions
    float* Ps = RG + kRows * 2 * kLdB;         // K-split partials
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row0 = blockIdx.x * kRows;
    const int d = p.d, M = p.M, dp = (d + 15) & ~15;
    const uint32_t seed = epoch_seed(p.drop_seed, p.drop_epoch);
    const bool ffn = p.w1p != nullptr;         // uniform
    const bool ln_wave = wave < kRows;
    const int lrow = row0 + wave;
    const bool lrow_ok = ln_wave && lrow < M;

    // ---- everything the two norm stages read, up front (one memory round trip behind the first GEMM)
    LnIn q1, q2;
    float dzin[5];
    ln_prefetch(q1, p.o1, p.res1, p.gamma1, p.mean1, p.rstd1, lrow, lrow_ok, d, lane);
    if (ffn) ln_prefetch(q2, p.o2, p.res2, p.gamma2, p.mean2, p.rstd2, lrow, lrow_ok, d, lane);
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int c = lane + 64 * j;
        dzin[j] = (lrow_ok && c < d && p.dzin) ? p.dzin[(int64_t)lrow * p.dzin_rs + c] : 0.f;
    }
    // the ReLU gate of linear2's data gradient: the saved activation of this lane's column, 8 rows
    const int N1 = p.N1;
    const GemmPlan gF1 = plan_for(ffn ? N1 : 64, d);
    const Slab wF1 = slab_of(gF1);
    const int colF1 = wF1.slab * 64 + lane;
    float act[kRows];
#pragma unroll
    for (int i = 0; i < kRows; ++i)
        act[i] = (ffn && wF1.h == 0 && colF1 < N1 && row0 + i < M) ? p.act[(int64_t)(row0 + i) * N1 + colF1] : 0.f;

    // ---- dx = dzin + g0 W0
    const bool pre = p.g0 != nullptr;          // uniform
    const GemmPlan g0p = plan_for(d, pre ? p.K0 : 16);
    const GemmPlan gF2 = plan_for(d, ffn ? N1 : 16);
    const GemmPlan g3 = plan_for(d, d);
    RowGemm mm;                              // one at a time: the next GEMM's first loads go out before the stage
                                               // that produces its rows (norm / epilogue) runs
    if (pre) {
        mm.begin(p.K0, p.w0p, g0p, slab_of(g0p));
        const int K0 = p.K0, K0p = (K0 + 15) & ~15;
        const int r = wave & (kRows - 1), half = wave >> 3;
        const int gr = row0 + r;
        int64_t goff = (int64_t)gr * p.g0_rs;
        if (p.g0_grp > 0) { const int g = small_div(gr, p.g0_grp); goff = (int64_t)g * p.g0_gs + (int64_t)(gr - g * p.g0_grp) * p.g0_rs; }
        const float* grow = p.g0 + goff;
        // all loads of the row first, then the LDS stores (as a loop: one memory round trip per 128 columns, up to 15 for the
        // 1 800-wide all-layer K/V gradient; train step 1.69 -> 1.65 ms together with the forward kernel's rows,
        // profiles/r05_x_ab_chain_row_loads.txt)
        constexpr int NA0 = (kMaxK0 + 16 + 127) / 128;
        float gv[NA0];
#pragma unroll
        for (int j = 0; j < NA0; ++j) {
            const int k = lane + 64 * half + 128 * j;
            gv[j] = (gr < M && k < K0) ? grow[k] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < NA0; ++j) {
            const int k = lane + 64 * half + 128 * j;
            if (k < K0p) XA[r * kLdA + k] = gv[j];
        }
        __syncthreads();
        const Slab w = slab_of(g0p);
        f32x4 acc0, acc1;
        mm.run(XA, kLdA, acc0, acc1);
        if (w.h < g0p.splits) {
            float* q = Ps + (size_t)w.h * kRows * (g0p.nslab * 64) + w.slab * 64 + lane;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                q[i * g0p.nslab * 64] = acc0[i];
                q[(4 + i) * g0p.nslab * 64] = acc1[i];
            }
        }
        __syncthreads();
    }
    // ---- norm stage 1 (one wave per row)
    if (ffn) mm.begin(d, p.w1p, gF1, wF1);
    else mm.begin(d, p.w3p, g3, slab_of(g3));
    if (ln_wave) {
        float dx[5], dz[5], dod[5], dgam[5], dbet[5];
        const int npad = g0p.nslab * 64;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int c = lane + 64 * j;
            float t = dzin[j];
            if (pre && c < d) {
                float s = Ps[wave * npad + c];
                for (int h = 1; h < g0p.splits; ++h) s += Ps[(h * kRows + wave) * npad + c];
                t += s;
            }
            dx[j] = t;
        }
        const Dropout drop = make_dropout(p.drop1_p, seed, p.drop1_site);
        ln_bwd_row(q1, dx, drop, lrow, d, lane, dz, dod, dgam, dbet);
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int c = lane + 64 * j;
            if (c < dp) {
                XB[wave * kLdB + c] = (lrow_ok && c < d) ? dod[j] : 0.f;
                DZ[wave * kLdB + c] = (lrow_ok && c < d) ? dz[j] : 0.f;
                RG[(wave * 2 + 0) * kLdB + c] = (lrow_ok && c < d) ? dgam[j] : 0.f;
                RG[(wave * 2 + 1) * kLdB + c] = (lrow_ok && c < d) ? dbet[j] : 0.f;
                if (lrow_ok && c < d) {
                    p.do1[(int64_t)lrow * d + c] = dod[j];
                    if (!ffn) p.dz_out[(int64_t)lrow * d + c] = dz[j];
                }
            }
        }
    }
    __syncthreads();
    {   // gamma / beta partial sums of this workgroup's rows (fixed order), layout of layernorm_bwd_kernel's partials
        float* pr = p.part1 + (int64_t)blockIdx.x * 2 * d;
        const int which = tid >> 9, c = tid & 511;
        if (c < d) {
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < kRows; ++r) s += RG[(r * 2 + which) * kLdB + c];
            pr[which * d + c] = s;
        }
    }
    if (ffn) {
        // ---- t = gate(do1 W2): the K split 0 wave of a slab owns the result
        {
            f32x4 acc0, acc1;
            mm.run(XB, kLdB, acc0, acc1);
            mm.begin(N1, p.w2p, gF2, slab_of(gF2));
            const int npad = gF1.nslab * 64;
            if (gF1.splits > 1) {
                if (wF1.h > 0 && wF1.h < gF1.splits) {
                    float* q = Ps + (size_t)(wF1.h - 1) * kRows * npad + colF1;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        q[i * npad] = acc0[i];
                        q[(4 + i) * npad] = acc1[i];
                    }
                }
                __syncthreads();
            }
            if (wF1.h == 0) {
                float y[kRows];
#pragma unroll
                for (int i = 0; i < 4; ++i) { y[i] = acc0[i]; y[4 + i] = acc1[i]; }
                for (int h = 1; h < gF1.splits; ++h) {
                    const float* q = Ps + (size_t)(h - 1) * kRows * npad + colF1;
#pragma unroll
                    for (int i = 0; i < kRows; ++i) y[i] += q[i * npad];
                }
#pragma unroll
                for (int i = 0; i < kRows; ++i) {
                    const float t = act[i] > 0.f ? y[i] * p.gate_scale : 0.f;     // 0 beyond N1 / M (act = 0)
                    XA[i * kLdA + colF1] = t;            // colF1 < nslab * 64 <= kLdA: pad columns become zeros
                    if (colF1 < N1 && row0 + i < M) p.t_out[(int64_t)(row0 + i) * N1 + colF1] = t;
                }
            }
            __syncthreads();
        }
        // ---- dx2 = dz1 + t W1
        {
            const Slab w = slab_of(gF2);
            f32x4 acc0, acc1;
            mm.run(XA, kLdA, acc0, acc1);
            mm.begin(d, p.w3p, g3, slab_of(g3));
            if (w.h < gF2.splits) {
                float* q = Ps + (size_t)w.h * kRows * (gF2.nslab * 64) + w.slab * 64 + lane;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    q[i * gF2.nslab * 64] = acc0[i];
                    q[(4 + i) * gF2.nslab * 64] = acc1[i];
                }
            }
            __syncthreads();
        }
        // ---- norm stage 2
        if (ln_wave) {
            float dx[5], dz[5], dod[5], dgam[5], dbet[5];
            const int npad = gF2.nslab * 64;
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int c = lane + 64 * j;
                float t = 0.f;
                if (c < d) {
                    float s = Ps[wave * npad + c];
                    for (int h = 1; h < gF2.splits; ++h) s += Ps[(h * kRows + wave) * npad + c];
                    t = DZ[wave * kLdB + c] + s;
                }
                dx[j] = t;
            }
            const Dropout drop = make_dropout(p.drop2_p, seed, p.drop2_site);
            ln_bwd_row(q2, dx, drop, lrow, d, lane, dz, dod, dgam, dbet);
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int c = lane + 64 * j;
                if (c < dp) {
                    XB[wave * kLdB + c] = (lrow_ok && c < d) ? dod[j] : 0.f;
                    RG[(wave * 2 + 0) * kLdB + c] = (lrow_ok && c < d) ? dgam[j] : 0.f;
                    RG[(wave * 2 + 1) * kLdB + c] = (lrow_ok && c < d) ? dbet[j] : 0.f;
                    if (lrow_ok && c < d) {
                        p.do2[(int64_t)lrow * d + c] = dod[j];
                        p.dz_out[(int64_t)lrow * d + c] = dz[j];
                    }
                }
            }
        }
        __syncthreads();
        {
            float* pr = p.part2 + (int64_t)blockIdx.x * 2 * d;
            const int which = tid >> 9, c = tid & 511;
            if (c < d) {
                float s = 0.f;
#pragma unroll
                for (int r = 0; r < kRows; ++r) s += RG[(r * 2 + which) * kLdB + c];
                pr[which * d + c] = s;
            }
        }
    }
    // ---- out = do W3 (data gradient of the out-projection): K split 0 waves store
    const Slab w3 = slab_of(g3);
    f32x4 acc0, acc1;
    mm.run(XB, kLdB, acc0, acc1);
    const int npad = g3.nslab * 64;
    const int col = w3.slab * 64 + lane;
    if (g3.splits > 1) {
        if (w3.h > 0 && w3.h < g3.splits) {
            float* q = Ps + (size_t)(w3.h - 1) * kRows * npad + col;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                q[i * npad] = acc0[i];
                q[(4 + i) * npad] = acc1[i];
            }
        }
        __syncthreads();
    }
    if (w3.h != 0 || col >= d) return;
    float y[kRows];
#pragma unroll
    for (int i = 0; i < 4; ++i) { y[i] = acc0[i]; y[4 + i] = acc1[i]; }
    for (int h = 1; h < g3.splits; ++h) {
        const float* q = Ps + (size_t)(h - 1) * kRows * npad + col;
#pragma unroll
        for (int i = 0; i < kRows; ++i) y[i] += q[i * npad];
    }
#pragma unroll
    for (int i = 0; i < kRows; ++i)
        if (row0 + i < M) p.out3[(int64_t)(row0 + i) * d + col] = y[i];
}

constexpr size_t kBwdSmemMax = (size_t)(kRows * (kMaxK0 + 4) + 2 * kRows * kLdB + kRows * 2 * kLdB + kPartFloats) * sizeof(float);
inline size_t bwd_smem(int lda) {
    return (size_t)(kRows * lda + 2 * kRows * kLdB + kRows * 2 * kLdB + kPartFloats) * sizeof(float);
}

}  // namespace
}  // namespace ick

extern "C" int ick_rowchain_bwd_supported(int32_t K0, int32_t d, int32_t N1) {
    using namespace ick;
    return K0 >= 0 && K0 <= kMaxK0 && d > 0 && d <= kMaxD && N1 >= 0 && N1 <= kMaxN1;
}

extern "C" int ick_rowchain_bwd(const ick_rowchain_bwd_args* in, void* stream) {
    using namespace ick;
    if (!in) return ICK_EINVAL;
    const ick_rowchain_bwd_args& a = *in;
    ICK_CHECK_ARG(a.M > 0 && ick_rowchain_bwd_supported(a.g0 ? a.K0 : 0, a.d, a.w1p ? a.N1 : 0));
    ICK_CHECK_ARG(a.g0 || a.dzin);
    if (a.g0) ICK_CHECK_ARG(a.w0p && a.K0 > 0 && a.g0_rs >= a.K0);
    ICK_CHECK_ARG(a.o1 && a.mean1 && a.rstd1 && a.gamma1 && a.do1 && a.part1);
    ICK_CHECK_ARG(a.w3p && a.out3 && a.dz_out);
    if (a.w1p) {
        ICK_CHECK_ARG(a.N1 > 0 && a.act && a.t_out && a.w2p);
        ICK_CHECK_ARG(a.o2 && a.mean2 && a.rstd2 && a.gamma2 && a.do2 && a.part2);
    }
    static LdsAttrOnce attr_set;
    if (int e = attr_set.ensure(reinterpret_cast<const void*>(rowchain_bwd_kernel), (int)kBwdSmemMax)) return e;
    // the launch declares the LDS its own stage widths need (85-104 KB; 136 only with the 1 800-wide K/V gradient input)
    const size_t smem = bwd_smem(lda_for(a.g0 != nullptr ? a.K0 : 0, a.w1p != nullptr ? a.N1 : 0));
    hipLaunchKernelGGL(rowchain_bwd_kernel, dim3(ceil_div(a.M, kRows)), dim3(kThreads), smem, (hipStream_t)stream, a);
    ICK_LAUNCH_RET();
}
