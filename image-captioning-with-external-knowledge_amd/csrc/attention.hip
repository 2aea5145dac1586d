// Multi-head attention core for the caption decoder: O = softmax(Q K^T * scale [+mask]) V.
//
// Call sites replaced (all through torch.nn.MultiheadAttention in the reference):
//   decoder causal self-attention, decoder cross-attention over the [196 image rows ; entity
//   rows ; fact rows] memory (geo-aware/models.py:358, knowledge-aware/models.py:508) and the
//   context encoders' self-attention (geo-aware/models.py:348).
//
// Shapes on this path are short and wide: T <= 102 queries, S <= 598 keys, dh = 30, and
// B*H = 640 independent (sample, head) problems, ~0.5 MFLOP each: the op is bound by moving
// K/V (HBM/L2) and by launch count, not by FLOPs (fp32 MFMA runs at the VALU rate anyway).
// One workgroup = one (sample, head, chunk of TQ queries):
//   phase 1  lane <-> key: the lane pulls its key row (dh floats) into registers, Q rows are
//            broadcast from LDS as float4; scores go to LDS as Ps[query][key]
//   phase 2  softmax per query row: one wave per row, wave-shuffle max / sum
//   phase 3  lane <-> (query, 4 output columns): P broadcast + one float4 of V per key
#include <cstdlib>

#include "attention_mfma.h"
#include "common.h"

namespace ick {
namespace {

// VEC: operands are in the head-major padded layout (rows of DHP floats, 16-byte aligned) that
// ick_gemm's head-split epilogue writes: every global access is a coalesced float4; pad columns
// (dh..DHP) may hold anything and are masked to zero in registers.
template <int DHP, bool VEC>
__global__ __launch_bounds__(256) void attn_kernel(ick_attn_args p, int TQ, int SLD) {
    chain_priority();
    constexpr int VLD = DHP + 4;  // float4 rows; 8 lanes x 16 B cover the 32 banks once
    constexpr int G = DHP / 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int S = p.S, dh = p.dh;
    float* Vs = smem;                 // S * VLD
    float* Qs = Vs + S * VLD;         // TQ * DHP
    float* Ps = Qs + TQ * DHP;        // TQ * SLD   (SLD odd: rows of consecutive queries sit on distinct banks)
    float* rowinv = Ps + TQ * SLD;    // TQ

    const int h = blockIdx.x, b = blockIdx.y, t0 = blockIdx.z * TQ;
    const int nt = min(TQ, p.T - t0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const Dropout drop = make_dropout(p.drop_p, epoch_seed(p.drop_seed, p.drop_epoch), p.drop_site);
    int slen = S;
    if (p.kv_len) slen = min(S, p.kv_len[b]);
    const float* qb = p.Q + (int64_t)b * p.q_bs + (int64_t)h * p.q_hs + (int64_t)t0 * p.q_ts;
    const float* kb = p.K + (int64_t)b * p.k_bs + (int64_t)h * p.k_hs;
    const float* vb = p.V + (int64_t)b * p.v_bs + (int64_t)h * p.v_hs;

    // This lane's first key row is requested before anything else so that its latency overlaps the
    // Q / V staging (one memory round trip for the whole workgroup instead of two).
    float kreg[DHP];
    auto load_key = [&](int s) {
        if constexpr (VEC) {
            const float4* kr = reinterpret_cast<const float4*>(kb + (int64_t)min(s, slen - 1) * DHP);
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const float4 x = kr[g];
                kreg[4 * g + 0] = x.x; kreg[4 * g + 1] = x.y; kreg[4 * g + 2] = x.z; kreg[4 * g + 3] = x.w;
            }
        } else {
            const float* kr = kb + (int64_t)min(s, slen - 1) * p.k_ss;
#pragma unroll
            for (int j = 0; j < DHP; ++j) kreg[j] = kr[j < dh ? j : 0];
        }
    };
    if (slen > 0) load_key(tid);

    // stage the Q chunk (zero padded to DHP columns) and V (zero padded to VLD columns)
    if constexpr (VEC) {
        for (int f = tid; f < nt * G; f += 256) {
            const int t = f / G, g = f % G;
            float4 x = *reinterpret_cast<const float4*>(qb + (int64_t)t * DHP + 4 * g);
            x.x = 4 * g + 0 < dh ? x.x : 0.f; x.y = 4 * g + 1 < dh ? x.y : 0.f;
            x.z = 4 * g + 2 < dh ? x.z : 0.f; x.w = 4 * g + 3 < dh ? x.w : 0.f;
            *reinterpret_cast<float4*>(Qs + t * DHP + 4 * g) = x;
        }
        for (int f = tid; f < slen * G; f += 256) {
            const int s = f / G, g = f % G;
            float4 x = *reinterpret_cast<const float4*>(vb + (int64_t)s * DHP + 4 * g);
            x.x = 4 * g + 0 < dh ? x.x : 0.f; x.y = 4 * g + 1 < dh ? x.y : 0.f;
            x.z = 4 * g + 2 < dh ? x.z : 0.f; x.w = 4 * g + 3 < dh ? x.w : 0.f;
            *reinterpret_cast<float4*>(Vs + s * VLD + 4 * g) = x;
        }
    } else {
        for (int idx = tid; idx < nt * DHP; idx += 256) {
            const int t = idx / DHP, j = idx % DHP;
            Qs[idx] = j < dh ? qb[(int64_t)t * p.q_ts + j] : 0.f;
        }
        for (int idx = tid; idx < slen * DHP; idx += 256) {
            const int s = idx / DHP, j = idx % DHP;
            Vs[s * VLD + j] = j < dh ? vb[(int64_t)s * p.v_ss + j] : 0.f;
        }
    }
    __syncthreads();

    // phase 1: lane <-> key.  The key row lives in registers, query rows are LDS broadcasts.
    for (int s = tid; s < slen; s += 256) {
        if (s != tid) load_key(s);
#pragma unroll
        for (int j = 0; j < DHP; ++j) kreg[j] = j < dh ? kreg[j] : 0.f;  // pad columns may hold anything
        for (int t = 0; t < nt; ++t) {
            float a0 = 0.f, a1 = 0.f;
#pragma unroll
            for (int j4 = 0; j4 < G; ++j4) {
                const float4 q = *reinterpret_cast<const float4*>(Qs + t * DHP + 4 * j4);
                a0 = fmaf(q.x, kreg[4 * j4 + 0], a0);
                a1 = fmaf(q.y, kreg[4 * j4 + 1], a1);
                a0 = fmaf(q.z, kreg[4 * j4 + 2], a0);
                a1 = fmaf(q.w, kreg[4 * j4 + 3], a1);
            }
            float a = (a0 + a1) * p.scale;
            if (p.causal && s > p.q_pos0 + t0 + t) a = -INFINITY;
            Ps[t * SLD + s] = a;
        }
    }
    __syncthreads();

    // phase 2: softmax per query row: one wave per row, lanes stride the keys
    for (int t = wave; t < nt; t += 4) {
        float* pr = Ps + t * SLD;
        float m = -INFINITY;
        for (int s = lane; s < slen; s += 64) m = fmaxf(m, pr[s]);
        m = wave_max(m);
        float sum = 0.f;
        const uint32_t rowbase = (uint32_t)((b * p.H + h) * p.T + t0 + t) * (uint32_t)S;
        for (int s = lane; s < slen; s += 64) {
            const float e = (m == -INFINITY) ? 0.f : __expf(pr[s] - m);
            // attention-weight dropout of nn.MultiheadAttention (training): the normaliser keeps every key
            pr[s] = drop.on() ? e * drop.mask(rowbase + (uint32_t)s) : e;
            sum += e;
        }
        sum = wave_sum(sum);
        if (lane == 0) {
            rowinv[t] = sum > 0.f ? 1.f / sum : 0.f;
            if (p.lse) p.lse[((int64_t)b * p.H + h) * p.T + t0 + t] = m + __logf(sum);
        }
    }
    __syncthreads();

    // phase 3: O[t][4g..4g+3] = sum_s P[t][s] V[s][4g..4g+3]; lane <-> (query, 4 output columns)
    for (int idx = tid; idx < nt * G; idx += 256) {
        const int t = idx / G, g = idx % G;
        const float* pr = Ps + t * SLD;
        const float* vc = Vs + 4 * g;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
        for (int s = 0; s < slen; ++s) {
            const float pv = pr[s];
            const float4 v = *reinterpret_cast<const float4*>(vc + s * VLD);
            o.x = fmaf(pv, v.x, o.x);
            o.y = fmaf(pv, v.y, o.y);
            o.z = fmaf(pv, v.z, o.z);
            o.w = fmaf(pv, v.w, o.w);
        }
        const float inv = rowinv[t];
        float* orow = p.O + (int64_t)b * p.o_bs + (int64_t)(t0 + t) * p.o_ts + h * dh;
        const float ov[4] = {o.x * inv, o.y * inv, o.z * inv, o.w * inv};
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (4 * g + q < dh) orow[4 * g + q] = ov[q];
    }
}

template <int DHP>
int launch_attn(const ick_attn_args& a, hipStream_t s) {
    constexpr int VLD = DHP + 4;
    const int SLD = a.S | 1;
    // queries per workgroup: all of them when they fit next to V in LDS, else chunks
    const size_t fixed = (size_t)a.S * VLD;
    int TQ = a.T;
    const size_t budget = 150 * 1024 / sizeof(float);
    while (TQ > 1 && fixed + (size_t)TQ * (DHP + SLD + 1) > budget) TQ = (TQ + 1) / 2;
    const size_t fl = fixed + (size_t)TQ * (DHP + SLD + 1);
    if (fl > budget) return ICK_EINVAL;
    const size_t smem = fl * sizeof(float);
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    const bool vec = a.q_ts == DHP && a.k_ss == DHP && a.v_ss == DHP && al16(a.Q) && al16(a.K) && al16(a.V) &&
                     a.q_bs % 4 == 0 && a.q_hs % 4 == 0 && a.k_bs % 4 == 0 && a.k_hs % 4 == 0 && a.v_bs % 4 == 0 &&
                     a.v_hs % 4 == 0;
    static LdsAttrOnce attr_set[2];
    const void* kern = vec ? (const void*)attn_kernel<DHP, true> : (const void*)attn_kernel<DHP, false>;
    if (int e = attr_set[vec].ensure(kern, 160 * 1024)) return e;
    const dim3 grid(a.H, a.B, ceil_div(a.T, TQ));
    if (vec) hipLaunchKernelGGL((attn_kernel<DHP, true>), grid, dim3(256), smem, s, a, TQ, SLD);
    else hipLaunchKernelGGL((attn_kernel<DHP, false>), grid, dim3(256), smem, s, a, TQ, SLD);
    ICK_LAUNCH_RET();
}

// ---------------------------------------------------------------------------------------------
// Backward of the attention core (training step).  Inputs are the forward's head-major Q/K/V, its
// row-major output O, the upstream gradient dO (row-major) and the saved log-sum-exp; outputs are
// row-major gradients (the layout the weight/data-gradient GEMMs consume):
//   P = exp(QK^T*scale - lse);  dV = P^T dO;  dP = dO V^T;  D = rowsum(dO*O)
//   dS = P*(dP - D);  dQ = scale * dS K;  dK = scale * dS^T Q
// One workgroup per (sample, head, query chunk).  Phase 1: lane <-> key keeps its K and V rows and
// its dK / dV accumulators in registers while the queries stream from LDS; the scaled dS tile is
// parked in LDS.  Phase 2: lane <-> (query, 4 columns) forms dQ from dS and the K tile in LDS.
// ---------------------------------------------------------------------------------------------
template <int DHP>
__global__ __launch_bounds__(256) void attn_bwd_kernel(ick_attn_bwd_args p, int TQ, int SLD) {
    chain_priority();
    constexpr int VLD = DHP + 4;
    constexpr int G = DHP / 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int S = p.S, dh = p.dh;
    float* Ks = smem;                  // S * VLD
    float* Qs = Ks + S * VLD;          // TQ * DHP
    float* Gs = Qs + TQ * DHP;         // TQ * DHP   (dO rows)
    float* dS = Gs + TQ * DHP;         // TQ * SLD
    float* Dl = dS + TQ * SLD;         // TQ  rowsum(dO * O)
    float* Ls = Dl + TQ;               // TQ  lse
    float* Tr = Ls + TQ + ((4 - ((TQ * (2 * DHP + SLD + 2)) & 3)) & 3);   // 256 * DHP transpose staging, 16-byte aligned

    const int h = blockIdx.x, b = blockIdx.y, t0 = blockIdx.z * TQ;
    const int nt = min(TQ, p.T - t0);
    const int tid = threadIdx.x;
    const bool multi = gridDim.z > 1;
    const Dropout drop = make_dropout(p.drop_p, epoch_seed(p.drop_seed, p.drop_epoch), p.drop_site);
    const float* qb = p.Q + (int64_t)b * p.q_bs + (int64_t)h * p.q_hs + (int64_t)t0 * DHP;
    const float* kb = p.K + (int64_t)b * p.k_bs + (int64_t)h * p.k_hs;
    const float* vb = p.V + (int64_t)b * p.v_bs + (int64_t)h * p.v_hs;
    const float* ob = p.O + (int64_t)b * p.o_bs + (int64_t)t0 * p.o_ts + h * dh;
    const float* gb = p.dO + (int64_t)b * p.o_bs + (int64_t)t0 * p.o_ts + h * dh;

    for (int f = tid; f < nt * G; f += 256) {
        const int t = f / G, g = f % G;
        float4 x = *reinterpret_cast<const float4*>(qb + (int64_t)t * DHP + 4 * g);
        x.x = 4 * g + 0 < dh ? x.x : 0.f; x.y = 4 * g + 1 < dh ? x.y : 0.f;
        x.z = 4 * g + 2 < dh ? x.z : 0.f; x.w = 4 * g + 3 < dh ? x.w : 0.f;
        *reinterpret_cast<float4*>(Qs + t * DHP + 4 * g) = x;
    }
    for (int idx = tid; idx < nt * DHP; idx += 256) {
        const int t = idx / DHP, j = idx % DHP;
        Gs[idx] = j < dh ? gb[(int64_t)t * p.o_ts + j] : 0.f;
    }
    for (int f = tid; f < S * G; f += 256) {
        const int s = f / G, g = f % G;
        float4 x = *reinterpret_cast<const float4*>(kb + (int64_t)s * DHP + 4 * g);
        x.x = 4 * g + 0 < dh ? x.x : 0.f; x.y = 4 * g + 1 < dh ? x.y : 0.f;
        x.z = 4 * g + 2 < dh ? x.z : 0.f; x.w = 4 * g + 3 < dh ? x.w : 0.f;
        *reinterpret_cast<float4*>(Ks + s * VLD + 4 * g) = x;
    }
    if (tid < nt) {
        float dsum = 0.f;
        for (int j = 0; j < dh; ++j) dsum = fmaf(gb[(int64_t)tid * p.o_ts + j], ob[(int64_t)tid * p.o_ts + j], dsum);
        Dl[tid] = dsum;
        Ls[tid] = p.lse[((int64_t)b * p.H + h) * p.T + t0 + tid];
    }
    __syncthreads();

    // Phase 1 runs once per 256-key block (S <= 256 on this path's bench shapes: one block); the lane's
    // dK / dV rows stay in registers until they are written out through LDS (coalesced row segments)
    for (int s0 = 0; s0 < S; s0 += 256) {
        const int s = s0 + tid;
        const bool live = s < S;
        const int sc_ = live ? s : S - 1;
        float kreg[DHP], vreg[DHP], dk[DHP], dv[DHP];
        const float4* vr = reinterpret_cast<const float4*>(vb + (int64_t)sc_ * DHP);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const float4 kx = *reinterpret_cast<const float4*>(Ks + sc_ * VLD + 4 * g);
            const float4 vx = vr[g];
            kreg[4 * g + 0] = kx.x; kreg[4 * g + 1] = kx.y; kreg[4 * g + 2] = kx.z; kreg[4 * g + 3] = kx.w;
            vreg[4 * g + 0] = 4 * g + 0 < dh ? vx.x : 0.f; vreg[4 * g + 1] = 4 * g + 1 < dh ? vx.y : 0.f;
            vreg[4 * g + 2] = 4 * g + 2 < dh ? vx.z : 0.f; vreg[4 * g + 3] = 4 * g + 3 < dh ? vx.w : 0.f;
        }
#pragma unroll
        for (int j = 0; j < DHP; ++j) { dk[j] = 0.f; dv[j] = 0.f; }
        if (live) {
            for (int t = 0; t < nt; ++t) {
                float sc = 0.f, dp = 0.f;
                float q[DHP], g[DHP];
#pragma unroll
                for (int j4 = 0; j4 < G; ++j4) {
                    const float4 qx = *reinterpret_cast<const float4*>(Qs + t * DHP + 4 * j4);
                    const float4 gx = *reinterpret_cast<const float4*>(Gs + t * DHP + 4 * j4);
                    q[4 * j4 + 0] = qx.x; q[4 * j4 + 1] = qx.y; q[4 * j4 + 2] = qx.z; q[4 * j4 + 3] = qx.w;
                    g[4 * j4 + 0] = gx.x; g[4 * j4 + 1] = gx.y; g[4 * j4 + 2] = gx.z; g[4 * j4 + 3] = gx.w;
                }
#pragma unroll
                for (int j = 0; j < DHP; ++j) {
                    sc = fmaf(q[j], kreg[j], sc);
                    dp = fmaf(g[j], vreg[j], dp);
                }
                float pr = __expf(sc * p.scale - Ls[t]);
                if (p.causal && s > p.q_pos0 + t0 + t) pr = 0.f;
                float mk = 1.f;
                if (drop.on()) mk = drop.mask((uint32_t)((b * p.H + h) * p.T + t0 + t) * (uint32_t)S + (uint32_t)s);
                const float ds = pr * (dp * mk - Dl[t]) * p.scale;
                const float prd = pr * mk;
#pragma unroll
                for (int j = 0; j < DHP; ++j) {
                    dv[j] = fmaf(prd, g[j], dv[j]);
                    dk[j] = fmaf(ds, q[j], dk[j]);
                }
                dS[t * SLD + s] = ds;
            }
        }
        // hand the rows over through LDS: [key][DHP] per matrix, then lanes walk (key, column) in
        // column-fastest order so that a wave's stores cover whole 120-byte row segments
        const int nkeys = min(256, S - s0);
        for (int which = 0; which < 2; ++which) {
            __syncthreads();
            if (live) {
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const float* src = which == 0 ? dk : dv;
                    *reinterpret_cast<float4*>(Tr + tid * DHP + 4 * g) =
                        make_float4(src[4 * g], src[4 * g + 1], src[4 * g + 2], src[4 * g + 3]);
                }
            }
            __syncthreads();
            float* base = which == 0 ? p.dK + (int64_t)b * p.dk_bs + h * dh : p.dV + (int64_t)b * p.dv_bs + h * dh;
            const int64_t rs = which == 0 ? p.dk_ss : p.dv_ss;
            for (int idx = tid; idx < nkeys * dh; idx += 256) {
                const int kk = idx / dh, j = idx - kk * dh;
                float* dst = base + (int64_t)(s0 + kk) * rs + j;
                const float v = Tr[kk * DHP + j];
                if (multi) atomicAdd(dst, v); else *dst = v;
            }
        }
    }
    __syncthreads();

    for (int idx = tid; idx < nt * G; idx += 256) {
        const int t = idx / G, g = idx % G;
        const float* dr = dS + t * SLD;
        const float* kc = Ks + 4 * g;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
        for (int s = 0; s < S; ++s) {
            const float d = dr[s];
            const float4 kx = *reinterpret_cast<const float4*>(kc + s * VLD);
            o.x = fmaf(d, kx.x, o.x); o.y = fmaf(d, kx.y, o.y); o.z = fmaf(d, kx.z, o.z); o.w = fmaf(d, kx.w, o.w);
        }
        float* dq = p.dQ + (int64_t)b * p.dq_bs + (int64_t)(t0 + t) * p.dq_ts + h * dh;
        const float ov[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (4 * g + q < dh) dq[4 * g + q] = ov[q];
    }
}

// queries per workgroup of the general backward kernel (all of them when they fit in LDS, else chunks whose
// dK / dV contributions are combined with float atomics)
inline int attn_bwd_chunk(int T, int S, int DHP) {
    const int VLD = DHP + 4, SLD = S | 1;
    const size_t fixed = (size_t)S * VLD, budget = 150 * 1024 / sizeof(float), stage = (size_t)256 * DHP + 4;
    int TQ = T;
    while (TQ > 1 && fixed + stage + (size_t)TQ * (2 * DHP + SLD + 2) > budget) TQ = (TQ + 1) / 2;
    return TQ;
}
inline bool attn_bwd_single_chunk(int T, int S, int DHP) { return attn_bwd_chunk(T, S, DHP) >= T; }

template <int DHP>
int launch_attn_bwd(const ick_attn_bwd_args& a, hipStream_t s) {
    constexpr int VLD = DHP + 4;
    const int SLD = a.S | 1;
    const size_t fixed = (size_t)a.S * VLD;
    const size_t budget = 150 * 1024 / sizeof(float);
    const size_t stage = (size_t)256 * DHP + 4;
    const int TQ = attn_bwd_chunk(a.T, a.S, DHP);
    const size_t fl = fixed + stage + (size_t)TQ * (2 * DHP + SLD + 2);
    if (fl > budget) return ICK_EINVAL;
    auto kern = attn_bwd_kernel<DHP>;
    static LdsAttrOnce attr_set;
    if (int e = attr_set.ensure((const void*)kern, 160 * 1024)) return e;
    hipLaunchKernelGGL(kern, dim3(a.H, a.B, ceil_div(a.T, TQ)), dim3(256), fl * sizeof(float), s, a, TQ, SLD);
    ICK_LAUNCH_RET();
}

}  // namespace
}  // namespace ick

extern "C" int ick_attention(const ick_attn_args* in, void* stream) {
    using namespace ick;
    if (!in) return ICK_EINVAL;
    const ick_attn_args& a = *in;
    ICK_CHECK_ARG(a.Q && a.K && a.V && a.O);
    ICK_CHECK_ARG(a.B > 0 && a.H > 0 && a.T > 0 && a.S > 0 && a.dh > 0 && a.dh <= 64);
    ICK_CHECK_ARG(a.B <= 65535);
    hipStream_t s = (hipStream_t)stream;
    static const bool no_mfma = getenv("ICK_ATTN_NO_MFMA") != nullptr;   // experiment hook: force the general kernels
    if (!no_mfma) {
        const int rc = launch_attn_mfma(a, s);
        if (rc != kAttnMfmaUnsupported) return rc;
    }
    if (a.dh <= 32) return launch_attn<32>(a, s);
    return launch_attn<64>(a, s);
}


extern "C" int ick_attention_bwd(const ick_attn_bwd_args* in, void* stream) {
    using namespace ick;
    if (!in) return ICK_EINVAL;
    const ick_attn_bwd_args& a = *in;
    ICK_CHECK_ARG(a.Q && a.K && a.V && a.O && a.dO && a.lse && a.dQ && a.dK && a.dV);
    ICK_CHECK_ARG(a.B > 0 && a.B <= 65535 && a.H > 0 && a.T > 0 && a.S > 0 && a.dh > 0 && a.dh <= 64);
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    const int DHP = a.dh <= 32 ? 32 : 64;
    // head-major padded operands only (what the training forward produces)
    ICK_CHECK_ARG(a.q_ts == DHP && a.k_ss == DHP && a.v_ss == DHP && al16(a.Q) && al16(a.K) && al16(a.V));
    ICK_CHECK_ARG(a.q_bs % 4 == 0 && a.q_hs % 4 == 0 && a.k_bs % 4 == 0 && a.k_hs % 4 == 0 && a.v_bs % 4 == 0 &&
                  a.v_hs % 4 == 0);
    hipStream_t s = (hipStream_t)stream;
    static const bool no_mfma = getenv("ICK_ATTN_NO_MFMA") != nullptr;
    if (DHP == 32 && !no_mfma) {
        const int rc = launch_attn_bwd_mfma(a, s);
        if (rc != kAttnMfmaUnsupported) return rc;
    }
    if (DHP == 32) return launch_attn_bwd<32>(a, s);
    return launch_attn_bwd<64>(a, s);
}

extern "C" int ick_attention_bwd_overwrites(int32_t T, int32_t S, int32_t dh) {
    using namespace ick;
    if (dh <= 32 && attn_mfma_shape_ok(T, S, dh) && getenv("ICK_ATTN_NO_MFMA") == nullptr) return 1;
    return attn_bwd_single_chunk(T, S, dh <= 32 ? 32 : 64) ? 1 : 0;
}
