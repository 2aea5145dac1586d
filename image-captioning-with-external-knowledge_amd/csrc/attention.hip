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
//            broadcast from LDS as float4; scores stay in registers (TQ per lane)
//   phase 2  softmax over keys = across lanes: wave shuffles + one LDS hop between the 4 waves
//   phase 3  probabilities are parked in LDS as Pt[key][query] (float4 along queries) next to
//            V[key][dh]; lane <-> (4 queries, one output column) accumulates P.V
#include "common.h"

namespace ick {
namespace {

template <int DHP, int TQ, int KB>
__global__ __launch_bounds__(256) void attn_kernel(ick_attn_args p) {
    constexpr int VLD = DHP + 1;  // odd stride: lanes reading one column of consecutive keys hit distinct banks
    constexpr int PLD = TQ + 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int S = p.S, dh = p.dh;
    float* Vs = smem;                       // S * VLD
    float* Pt = Vs + ((S * VLD + 3) & ~3);  // S * PLD   (16-byte aligned)
    float* Qs = Pt + S * PLD;               // TQ * DHP
    float* red = Qs + TQ * DHP;             // 4 * TQ
    float* rowinv = red + 4 * TQ;           // TQ

    const int h = blockIdx.x, b = blockIdx.y, t0 = blockIdx.z * TQ;
    const int nt = min(TQ, p.T - t0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int slen = S;
    if (p.kv_len) slen = min(S, p.kv_len[b]);

    // stage Q chunk (zero padded to DHP) and V
    for (int idx = tid; idx < TQ * DHP; idx += 256) {
        const int t = idx / DHP, j = idx % DHP;
        float v = 0.f;
        if (t < nt && j < dh) v = p.Q[(int64_t)b * p.q_bs + (int64_t)(t0 + t) * p.q_ts + h * dh + j];
        Qs[idx] = v;
    }
    {
        const float* vb = p.V + (int64_t)b * p.v_bs + (int64_t)h * p.v_hs;
        for (int idx = tid; idx < slen * dh; idx += 256) {
            const int s = idx / dh, j = idx - s * dh;
            Vs[s * VLD + j] = vb[(int64_t)s * p.v_ss + j];
        }
    }
    __syncthreads();

    // phase 1: scores in registers
    float sc[KB][TQ];
    const float* kb_base = p.K + (int64_t)b * p.k_bs + (int64_t)h * p.k_hs;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        const int s = tid + 256 * kb;
        float kreg[DHP];
        if (s < slen) {
            const float* kr = kb_base + (int64_t)s * p.k_ss;
#pragma unroll
            for (int j = 0; j < DHP; ++j) kreg[j] = j < dh ? kr[j] : 0.f;
        } else {
#pragma unroll
            for (int j = 0; j < DHP; ++j) kreg[j] = 0.f;
        }
#pragma unroll
        for (int t = 0; t < TQ; ++t) {
            float a = 0.f;
#pragma unroll
            for (int j4 = 0; j4 < DHP / 4; ++j4) {
                const float4 q = *reinterpret_cast<const float4*>(Qs + t * DHP + 4 * j4);
                a = fmaf(q.x, kreg[4 * j4 + 0], a);
                a = fmaf(q.y, kreg[4 * j4 + 1], a);
                a = fmaf(q.z, kreg[4 * j4 + 2], a);
                a = fmaf(q.w, kreg[4 * j4 + 3], a);
            }
            a *= p.scale;
            bool ok = s < slen && t < nt;
            if (p.causal && s > p.q_pos0 + t0 + t) ok = false;
            sc[kb][t] = ok ? a : -INFINITY;
        }
    }

    // phase 2: softmax across lanes (keys) per query row
#pragma unroll
    for (int t = 0; t < TQ; ++t) {
        float m = sc[0][t];
#pragma unroll
        for (int kb = 1; kb < KB; ++kb) m = fmaxf(m, sc[kb][t]);
        m = wave_max(m);
        if (lane == 0) red[wave * TQ + t] = m;
    }
    __syncthreads();
    float rmax[TQ];
#pragma unroll
    for (int t = 0; t < TQ; ++t)
        rmax[t] = fmaxf(fmaxf(red[t], red[TQ + t]), fmaxf(red[2 * TQ + t], red[3 * TQ + t]));
    __syncthreads();
#pragma unroll
    for (int t = 0; t < TQ; ++t) {
        float sum = 0.f;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            const float e = (rmax[t] == -INFINITY) ? 0.f : __expf(sc[kb][t] - rmax[t]);
            sc[kb][t] = e;
            sum += e;
        }
        sum = wave_sum(sum);
        if (lane == 0) red[wave * TQ + t] = sum;
    }
    // park unnormalised probabilities: Pt[s][t]
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        const int s = tid + 256 * kb;
        if (s < slen) {
#pragma unroll
            for (int g = 0; g < TQ / 4; ++g)
                *reinterpret_cast<float4*>(Pt + s * PLD + 4 * g) =
                    make_float4(sc[kb][4 * g], sc[kb][4 * g + 1], sc[kb][4 * g + 2], sc[kb][4 * g + 3]);
        }
    }
    __syncthreads();
    if (tid < TQ) {
        const float sum = red[tid] + red[TQ + tid] + red[2 * TQ + tid] + red[3 * TQ + tid];
        rowinv[tid] = sum > 0.f ? 1.f / sum : 0.f;
        if (p.lse && tid < nt)
            p.lse[((int64_t)b * p.H + h) * p.T + t0 + tid] = rmax[tid] + __logf(sum);
    }
    __syncthreads();

    // phase 3: O[t][j] = sum_s P[t][s] V[s][j]
    for (int idx = tid; idx < (TQ / 4) * DHP; idx += 256) {
        const int tg = idx / DHP, j = idx % DHP;
        if (4 * tg >= nt) continue;
        float o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
        if (j < dh) {
#pragma unroll 4
            for (int s = 0; s < slen; ++s) {
                const float v = Vs[s * VLD + j];
                const float4 pr = *reinterpret_cast<const float4*>(Pt + s * PLD + 4 * tg);
                o0 = fmaf(pr.x, v, o0);
                o1 = fmaf(pr.y, v, o1);
                o2 = fmaf(pr.z, v, o2);
                o3 = fmaf(pr.w, v, o3);
            }
            const float o[4] = {o0, o1, o2, o3};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int t = 4 * tg + q;
                if (t < nt)
                    p.O[(int64_t)b * p.o_bs + (int64_t)(t0 + t) * p.o_ts + h * dh + j] = o[q] * rowinv[t];
            }
        }
    }
}

template <int DHP, int TQ, int KB>
int launch_attn(const ick_attn_args& a, hipStream_t s) {
    constexpr int VLD = DHP + 1, PLD = TQ + 4;
    const size_t fl = ((size_t)(a.S * VLD + 3) & ~(size_t)3) + (size_t)a.S * PLD + (size_t)TQ * DHP + 5 * TQ;
    const size_t smem = fl * sizeof(float);
    if (smem > 160 * 1024) return ICK_EINVAL;
    auto kern = attn_kernel<DHP, TQ, KB>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(a.H, a.B, ceil_div(a.T, TQ)), dim3(256), smem, s, a);
    ICK_LAUNCH_RET();
}

template <int DHP>
int dispatch_attn(const ick_attn_args& a, hipStream_t s) {
    if (a.S <= 256) {
        if (a.T <= 4) return launch_attn<DHP, 4, 1>(a, s);
        if (a.T <= 8) return launch_attn<DHP, 8, 1>(a, s);
        if (a.T <= 20) return launch_attn<DHP, 20, 1>(a, s);
        return launch_attn<DHP, 32, 1>(a, s);
    }
    if (a.S <= 512) {
        if (a.T <= 4) return launch_attn<DHP, 4, 2>(a, s);
        return launch_attn<DHP, 16, 2>(a, s);
    }
    if (a.S <= 768) {
        if (a.T <= 4) return launch_attn<DHP, 4, 3>(a, s);
        return launch_attn<DHP, 16, 3>(a, s);
    }
    return ICK_EINVAL;
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
    if (a.dh <= 32) return dispatch_attn<32>(a, s);
    return dispatch_attn<64>(a, s);
}
