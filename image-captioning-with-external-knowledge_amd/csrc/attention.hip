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
#include "common.h"

namespace ick {
namespace {

template <int DHP>
__global__ __launch_bounds__(256) void attn_kernel(ick_attn_args p, int TQ, int SLD) {
    constexpr int VLD = DHP + 4;  // float4 rows; 8 lanes x 16 B cover the 32 banks once
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int S = p.S, dh = p.dh;
    float* Vs = smem;                 // S * VLD
    float* Qs = Vs + S * VLD;         // TQ * DHP
    float* Ps = Qs + TQ * DHP;        // TQ * SLD   (SLD odd: rows of consecutive queries sit on distinct banks)
    float* rowinv = Ps + TQ * SLD;    // TQ

    const int h = blockIdx.x, b = blockIdx.y, t0 = blockIdx.z * TQ;
    const int nt = min(TQ, p.T - t0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int slen = S;
    if (p.kv_len) slen = min(S, p.kv_len[b]);

    // stage the Q chunk (zero padded to DHP columns) and V (zero padded to VLD columns)
    for (int idx = tid; idx < nt * DHP; idx += 256) {
        const int t = idx / DHP, j = idx % DHP;
        Qs[idx] = j < dh ? p.Q[(int64_t)b * p.q_bs + (int64_t)(t0 + t) * p.q_ts + h * dh + j] : 0.f;
    }
    {
        const float* vb = p.V + (int64_t)b * p.v_bs + (int64_t)h * p.v_hs;
        for (int idx = tid; idx < slen * VLD; idx += 256) {
            const int s = idx / VLD, j = idx - s * VLD;
            Vs[idx] = j < dh ? vb[(int64_t)s * p.v_ss + j] : 0.f;
        }
    }
    __syncthreads();

    // phase 1: lane <-> key.  The key row lives in registers, query rows are LDS broadcasts.
    const float* kb_base = p.K + (int64_t)b * p.k_bs + (int64_t)h * p.k_hs;
    for (int s = tid; s < slen; s += 256) {
        float kreg[DHP];
        const float* kr = kb_base + (int64_t)s * p.k_ss;
#pragma unroll
        for (int j = 0; j < DHP; ++j) kreg[j] = j < dh ? kr[j] : 0.f;
        for (int t = 0; t < nt; ++t) {
            float a0 = 0.f, a1 = 0.f;
#pragma unroll
            for (int j4 = 0; j4 < DHP / 4; ++j4) {
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
        for (int s = lane; s < slen; s += 64) {
            const float e = (m == -INFINITY) ? 0.f : __expf(pr[s] - m);
            pr[s] = e;
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
    constexpr int JG = DHP / 4;
    for (int idx = tid; idx < nt * JG; idx += 256) {
        const int t = idx / JG, g = idx % JG;
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
    auto kern = attn_kernel<DHP>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(a.H, a.B, ceil_div(a.T, TQ)), dim3(256), smem, s, a, TQ, SLD);
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
    if (a.dh <= 32) return launch_attn<32>(a, s);
    return launch_attn<64>(a, s);
}
