// Score head pieces around the vocabulary GEMM: pointer scores over the entity / fact context,
// greedy selection and token bookkeeping of predict(), and the packed cross-entropy of train.py.
//   get_scores    geo-aware/models.py:291-313, knowledge-aware/models.py:420-455
//   predict loop  geo-aware/models.py:410-442, knowledge-aware/models.py:573-608
//   loss          geo-aware/train.py:275-281 (pack_padded_sequence + CrossEntropyLoss(ignore_index=<pad>))
// The vocabulary logits themselves are ick_gemm writing straight into the concatenated
// (B, L, V+K+F) rows; the pointer kernel fills columns [V, V+K) and [V+K, V+K+F) of the same
// rows, so the reference's (L,B,K,d) broadcast products and the torch.cat never exist.
#include "common.h"

namespace ick {
namespace {

constexpr int kMaxPerLane = 16;

// one workgroup per (b, t) row of h; each wave walks context rows k = wave, wave+4, ... in batches of KB rows
// whose loads are all issued before the first reduction (one memory round trip per batch, not per row)
template <int NJ>
__global__ __launch_bounds__(256) void pointer_scores_kernel(const float* __restrict__ h, const float* __restrict__ ctx,
                                                             const float* __restrict__ w, const float* __restrict__ bias,
                                                             const float* __restrict__ ind, float* __restrict__ out,
                                                             int T, int Kc, int d, int64_t out_ld, int col0,
                                                             const int32_t* __restrict__ out_gmap) {
    chain_priority();
    constexpr int KB = 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y, t = blockIdx.x;
    const float* hr = h + ((int64_t)b * T + t) * d;
    float hv[NJ], wv[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int c = lane + 64 * j;
        hv[j] = c < d ? hr[c] : 0.f;
        wv[j] = c < d ? w[c] : 0.f;
    }
    const int ob = out_gmap ? out_gmap[b] : b;
    float* orow = out + ((int64_t)ob * T + t) * out_ld + col0;
    const float bs = bias[0];
    for (int k0 = wave; k0 < Kc; k0 += 4 * KB) {
        float cv[KB][NJ];
#pragma unroll
        for (int q = 0; q < KB; ++q) {
            const int k = k0 + 4 * q;
            const float* cr = ctx + ((int64_t)b * Kc + (k < Kc ? k : 0)) * d;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int c = lane + 64 * j;
                cv[q][j] = (k < Kc && c < d) ? cr[c] : 0.f;
            }
        }
#pragma unroll
        for (int q = 0; q < KB; ++q) {
            const int k = k0 + 4 * q;
            if (k >= Kc) break;      // wave-uniform
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc = fmaf(__fmul_rn(hv[j], cv[q][j]), wv[j], acc);
            acc = wave_sum(acc);
            if (lane == 0) {
                const float f = ind ? ind[((int64_t)b * T + t) * Kc + k] : 1.f;
                orow[k] = acc * f + bs;
            }
        }
    }
}

struct Top2 {
    float v1, v2;
    int i1, i2;
};
__device__ __forceinline__ void top2_push(Top2& s, float v, int i) {
    if (v > s.v1 || (v == s.v1 && i < s.i1)) {
        s.v2 = s.v1; s.i2 = s.i1; s.v1 = v; s.i1 = i;
    } else if (v > s.v2 || (v == s.v2 && i < s.i2)) {
        s.v2 = v; s.i2 = i;
    }
}
// Top-2 of one score row by a 256-thread workgroup; the result is valid in thread 0 (sh[0]).  The row is read
// eight elements per thread at a time (loads first, comparisons after: one memory round trip per batch instead of
// one per element).
__device__ __forceinline__ Top2 row_top2(const float* __restrict__ r, int Vx, Top2* sh) {
    const int tid = threadIdx.x;
    Top2 s{-INFINITY, -INFINITY, 0x7fffffff, 0x7fffffff};
    for (int i0 = 0; i0 < Vx; i0 += 256 * 8) {
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = i0 + tid + 256 * j;
            x[j] = i < Vx ? r[i] : -INFINITY;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = i0 + tid + 256 * j;
            if (i < Vx) top2_push(s, x[j], i);
        }
    }
    sh[tid] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) {
            Top2 a = sh[tid];
            const Top2 c = sh[tid + o];
            if (c.i1 != 0x7fffffff) top2_push(a, c.v1, c.i1);
            if (c.i2 != 0x7fffffff) top2_push(a, c.v2, c.i2);
            sh[tid] = a;
        }
        __syncthreads();
    }
    return sh[0];
}

__global__ __launch_bounds__(256) void top2_kernel(const float* __restrict__ scores, int64_t ld, int Vx,
                                                   int32_t* __restrict__ best, int32_t* __restrict__ second) {
    __shared__ Top2 sh[256];
    const int b = blockIdx.x;
    const Top2 t = row_top2(scores + (int64_t)b * ld, Vx, sh);
    if (threadIdx.x == 0) {
        best[b] = t.i1;
        second[b] = t.i2 == 0x7fffffff ? t.i1 : t.i2;
    }
}

// predict()'s per-step bookkeeping for caption b (geo-aware/models.py:410-441), given the step's two best tokens.
__device__ __forceinline__ void greedy_update_one(int b, int best_b, int second_b, int64_t* __restrict__ output,
                                                  int32_t* __restrict__ hist, int32_t* __restrict__ finished,
                                                  int64_t* __restrict__ next_token, int64_t* __restrict__ next_mask,
                                                  int step, int max_len, int V, int K, int has_facts, int end_token) {
    if (finished[b]) {
        next_token[b] = 0;
        next_mask[b] = 0;
        return;
    }
    int64_t* o = output + (int64_t)b * max_len;
    int32_t* hs = hist + (int64_t)b * max_len;
    const int i = step;
    int64_t out = best_b;
    o[i] = out;
    if (out == end_token) {
        finished[b] = 1;
        next_token[b] = 0;
        next_mask[b] = 0;
        return;
    }
    hs[i] = second_b;
    // repeated n-gram clean-up (geo-aware/models.py:421-435)
    for (int dupl = 0; dupl <= 4; dupl += 2) {
        if (i > dupl) {
            const int half = (dupl + 2) / 2;
            bool same = true;
            for (int j = 0; j < half; ++j) same = same && (o[i - j] == o[i - half - j]);
            if (same) {
                const int top = dupl == 0 ? 1 : dupl;
                for (int r = 0; r < top; ++r) o[i - r] = hs[i - r];
                break;
            }
        }
    }
    out = o[i];
    if (i < max_len - 1) {
        next_token[b] = out;
        next_mask[b] = (has_facts && out >= V + K) ? 2 : (out >= V ? 1 : 0);
    }
}

// One lane per caption (separate selection and bookkeeping: ick_top2 + ick_greedy_update).
__global__ void greedy_update_kernel(const int32_t* __restrict__ best, const int32_t* __restrict__ second,
                                     int64_t* __restrict__ output, int32_t* __restrict__ hist,
                                     int32_t* __restrict__ finished, int64_t* __restrict__ next_token,
                                     int64_t* __restrict__ next_mask, int B, int step, int max_len, int V, int K,
                                     int has_facts, int end_token) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    greedy_update_one(b, best[b], second[b], output, hist, finished, next_token, next_mask, step, max_len, V, K,
                      has_facts, end_token);
}

// Selection and bookkeeping of a greedy step in one launch: workgroup b scans caption b's score row, its thread 0
// updates the caption's state (one launch less per token of the decode loop).
__global__ __launch_bounds__(256) void greedy_select_kernel(const float* __restrict__ scores, int64_t ld, int Vx,
                                                            int64_t* __restrict__ output, int32_t* __restrict__ hist,
                                                            int32_t* __restrict__ finished,
                                                            int64_t* __restrict__ next_token,
                                                            int64_t* __restrict__ next_mask, int step, int max_len,
                                                            int V, int K, int has_facts, int end_token) {
    __shared__ Top2 sh[256];
    const int b = blockIdx.x;
    const Top2 t = row_top2(scores + (int64_t)b * ld, Vx, sh);
    if (threadIdx.x == 0)
        greedy_update_one(b, t.i1, t.i2 == 0x7fffffff ? t.i1 : t.i2, output, hist, finished, next_token, next_mask, step,
                          max_len, V, K, has_facts, end_token);
}

// Packed cross entropy: one workgroup per (b, t) score row.
__global__ __launch_bounds__(256) void packed_ce_rows_kernel(const float* __restrict__ scores, int64_t ld,
                                                             const int64_t* __restrict__ caps,
                                                             const int32_t* __restrict__ dl, int L, int Vx, int pad,
                                                             float* __restrict__ row_loss, float* __restrict__ dscores) {
    __shared__ float red[4];
    const int t = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int64_t row = (int64_t)b * L + t;
    const float* r = scores + row * ld;
    float* dr = dscores ? dscores + row * ld : nullptr;
    int64_t target = -1;
    bool use = t < L - 1 && t < dl[b];
    if (use) {
        target = caps[(int64_t)b * L + t + 1];
        use = target != pad && target >= 0 && target < Vx;
    }
    if (!use) {
        if (tid == 0) row_loss[row] = -1.f;  // marker: row does not contribute
        if (dr) {
            // (16-byte stores where the row allows: up to half of a batch's rows lie beyond their caption's length, and at
            // the knowledge vocabulary each is 200 KB of zeros)
            const int z4 = ((ld & 3) == 0 && (reinterpret_cast<uintptr_t>(dscores) & 15) == 0) ? (int)(Vx >> 2) : 0;
            float4* d4 = reinterpret_cast<float4*>(dr);
            for (int i = tid; i < z4; i += 256) d4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int i = 4 * z4 + tid; i < Vx; i += 256) dr[i] = 0.f;
        }
        return;
    }
    // Rows of up to 256 * 4 * kCeVec floats (16-byte aligned) stay in registers between the passes: one read
    // of the logits and one write of the gradient instead of three reads; longer / unaligned rows re-read.
    constexpr int kCeVec = 10;
    const bool in_regs = (Vx & 3) == 0 && (ld & 3) == 0 && Vx <= 256 * 4 * kCeVec &&
                         (reinterpret_cast<uintptr_t>(scores) & 15) == 0 &&
                         (dr == nullptr || (reinterpret_cast<uintptr_t>(dscores) & 15) == 0);
    if (in_regs) {
        const float4* r4 = reinterpret_cast<const float4*>(r);
        const int n4 = Vx >> 2;
        float4 x[kCeVec];
        float m = -INFINITY;
#pragma unroll
        for (int j = 0; j < kCeVec; ++j) {
            const int i = tid + 256 * j;
            x[j] = i < n4 ? r4[i] : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
            m = fmaxf(m, fmaxf(fmaxf(x[j].x, x[j].y), fmaxf(x[j].z, x[j].w)));
        }
        m = block_max<4>(m, red);
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < kCeVec; ++j) {
            x[j].x = __expf(x[j].x - m); x[j].y = __expf(x[j].y - m);
            x[j].z = __expf(x[j].z - m); x[j].w = __expf(x[j].w - m);
            s += (x[j].x + x[j].y) + (x[j].z + x[j].w);
        }
        s = block_sum<4>(s, red);
        if (tid == 0) row_loss[row] = m + __logf(s) - r[target];
        if (dr) {
            const float inv = 1.f / s;
            float4* d4 = reinterpret_cast<float4*>(dr);
            const int tq = (int)(target >> 2), tr = (int)(target & 3);
#pragma unroll
            for (int j = 0; j < kCeVec; ++j) {
                const int i = tid + 256 * j;
                if (i < n4) {
                    float4 g = make_float4(x[j].x * inv, x[j].y * inv, x[j].z * inv, x[j].w * inv);
                    if (i == tq) {
                        if (tr == 0) g.x -= 1.f; else if (tr == 1) g.y -= 1.f; else if (tr == 2) g.z -= 1.f; else g.w -= 1.f;
                    }
                    d4[i] = g;
                }
            }
        }
        return;
    }
    // Long rows (knowledge vocabulary: 50 071 columns): max and sum of exponentials in ONE pass with a running pair
    // (m, s) per thread, rescaled when the maximum moves; the pairs are merged through the block maximum.  Two reads of
    // the row instead of three.  Rows that start 16-byte aligned (the training step pads the row stride to a multiple
    // of 4 floats) are read and written as float4 with a scalar tail: a dword per lane moves 256 B per wave instruction,
    // a dwordx4 1 KiB (round 5: cfg4's loss 161 -> ~115 us for 2 x 256 MB read + 256 MB written).
    const bool vec = (ld & 3) == 0 && (reinterpret_cast<uintptr_t>(scores) & 15) == 0 &&
                     (dr == nullptr || (reinterpret_cast<uintptr_t>(dscores) & 15) == 0);
    const int n4 = vec ? (int)(Vx >> 2) : 0;             // float4 elements of the aligned bulk; [4 n4, Vx) is the scalar tail
    const float4* r4 = reinterpret_cast<const float4*>(r);
    float m = -INFINITY, s = 0.f;
    for (int i0 = 0; i0 < n4; i0 += 256 * 4) {
        float4 x[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = i0 + tid + 256 * j;
            x[j] = i < n4 ? r4[i] : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        }
        float bm = m;
#pragma unroll
        for (int j = 0; j < 4; ++j) bm = fmaxf(bm, fmaxf(fmaxf(x[j].x, x[j].y), fmaxf(x[j].z, x[j].w)));
        if (bm > -INFINITY) {
            s *= __expf(m - bm);            // m = -inf: s is 0 and stays 0
#pragma unroll
            for (int j = 0; j < 4; ++j)     // exp(-inf) = 0 for the tail
                s += (__expf(x[j].x - bm) + __expf(x[j].y - bm)) + (__expf(x[j].z - bm) + __expf(x[j].w - bm));
            m = bm;
        }
    }
    for (int i0 = 4 * n4; i0 < Vx; i0 += 256 * 8) {
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = i0 + tid + 256 * j;
            x[j] = i < Vx ? r[i] : -INFINITY;
        }
        float bm = m;
#pragma unroll
        for (int j = 0; j < 8; ++j) bm = fmaxf(bm, x[j]);
        if (bm > -INFINITY) {
            s *= __expf(m - bm);
#pragma unroll
            for (int j = 0; j < 8; ++j) s += __expf(x[j] - bm);
            m = bm;
        }
    }
    const float mt = m;
    m = block_max<4>(m, red);
    s = block_sum<4>(mt > -INFINITY ? s * __expf(mt - m) : 0.f, red);
    const float lse = m + __logf(s);
    if (tid == 0) row_loss[row] = lse - r[target];
    if (dr) {
        const float inv = 1.f / s;
        float4* d4 = reinterpret_cast<float4*>(dr);
        const int tq = (int)(target >> 2), tr = (int)(target & 3);
        for (int i0 = 0; i0 < n4; i0 += 256 * 4) {
            float4 x[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = i0 + tid + 256 * j;
                x[j] = i < n4 ? r4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = i0 + tid + 256 * j;
                if (i < n4) {
                    float4 gq = make_float4(__expf(x[j].x - m) * inv, __expf(x[j].y - m) * inv, __expf(x[j].z - m) * inv,
                                            __expf(x[j].w - m) * inv);
                    if (i == tq) {
                        if (tr == 0) gq.x -= 1.f; else if (tr == 1) gq.y -= 1.f; else if (tr == 2) gq.z -= 1.f; else gq.w -= 1.f;
                    }
                    d4[i] = gq;
                }
            }
        }
        for (int i0 = 4 * n4; i0 < Vx; i0 += 256 * 8) {
            float x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = i0 + tid + 256 * j;
                x[j] = i < Vx ? r[i] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = i0 + tid + 256 * j;
                if (i < Vx) dr[i] = __expf(x[j] - m) * inv - (i == target ? 1.f : 0.f);
            }
        }
    }
}

// Fixed-order reduction of the per-row losses (deterministic, unlike float atomics).
__global__ __launch_bounds__(256) void packed_ce_reduce_kernel(const float* __restrict__ row_loss, int n,
                                                               float* __restrict__ loss_sum, float* __restrict__ count) {
    __shared__ float red[4];
    float s = 0.f, c = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float v = row_loss[i];
        if (v > -0.5f) { s += v; c += 1.f; }
    }
    s = block_sum<4>(s, red);
    c = block_sum<4>(c, red);
    if (threadIdx.x == 0) { loss_sum[0] = s; count[0] = c; }
}

}  // namespace
}  // namespace ick

extern "C" int ick_pointer_scores(const float* h, const float* ctx, const float* w, const float* bias,
                                  const float* ind, float* out, int32_t B, int32_t T, int32_t Kc, int32_t d,
                                  int64_t out_ld, int32_t col0, const int32_t* out_gmap, void* stream) {
    using namespace ick;
    ICK_CHECK_ARG(h && ctx && w && bias && out && B > 0 && T > 0 && Kc > 0 && d > 0 && d <= 64 * kMaxPerLane);
    ICK_CHECK_ARG(B <= 65535 && col0 >= 0 && out_ld >= col0 + Kc);
    hipStream_t s = (hipStream_t)stream;
    if (d <= 320)
        hipLaunchKernelGGL(pointer_scores_kernel<5>, dim3(T, B), dim3(256), 0, s, h, ctx, w, bias, ind, out, T, Kc, d,
                           out_ld, col0, out_gmap);
    else if (d <= 512)
        hipLaunchKernelGGL(pointer_scores_kernel<8>, dim3(T, B), dim3(256), 0, s, h, ctx, w, bias, ind, out, T, Kc, d,
                           out_ld, col0, out_gmap);
    else
        hipLaunchKernelGGL(pointer_scores_kernel<kMaxPerLane>, dim3(T, B), dim3(256), 0, s, h, ctx, w, bias, ind, out, T,
                           Kc, d, out_ld, col0, out_gmap);
    ICK_LAUNCH_RET();
}

extern "C" int ick_top2(const float* scores, int64_t ld, int32_t B, int32_t Vx, int32_t* best, int32_t* second,
                        void* stream) {
    using namespace ick;
    ICK_CHECK_ARG(scores && best && second && B > 0 && Vx > 0 && ld >= Vx);
    hipLaunchKernelGGL(top2_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, scores, ld, Vx, best, second);
    ICK_LAUNCH_RET();
}

extern "C" int ick_greedy_update(const int32_t* best, const int32_t* second, int64_t* output, int32_t* top2_hist,
                                 int32_t* finished, int64_t* next_token, int64_t* next_mask, int32_t B, int32_t step,
                                 int32_t max_len, int32_t V, int32_t K, int32_t has_facts, int32_t end_token,
                                 void* stream) {
    using namespace ick;
    ICK_CHECK_ARG(best && second && output && top2_hist && finished && next_token && next_mask);
    ICK_CHECK_ARG(B > 0 && step >= 0 && step < max_len);
    hipLaunchKernelGGL(greedy_update_kernel, dim3(ceil_div(B, 64)), dim3(64), 0, (hipStream_t)stream, best, second,
                       output, top2_hist, finished, next_token, next_mask, B, step, max_len, V, K, has_facts,
                       end_token);
    ICK_LAUNCH_RET();
}

extern "C" int ick_greedy_select(const float* scores, int64_t ld, int32_t B, int32_t Vx, int64_t* output,
                                 int32_t* top2_hist, int32_t* finished, int64_t* next_token, int64_t* next_mask,
                                 int32_t step, int32_t max_len, int32_t V, int32_t K, int32_t has_facts,
                                 int32_t end_token, void* stream) {
    using namespace ick;
    ICK_CHECK_ARG(scores && output && top2_hist && finished && next_token && next_mask);
    ICK_CHECK_ARG(B > 0 && Vx > 0 && ld >= Vx && step >= 0 && step < max_len);
    hipLaunchKernelGGL(greedy_select_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, scores, ld, Vx, output,
                       top2_hist, finished, next_token, next_mask, step, max_len, V, K, has_facts, end_token);
    ICK_LAUNCH_RET();
}

extern "C" int ick_packed_ce(const float* scores, int64_t ld, const int64_t* captions_sorted,
                             const int32_t* decode_len, int32_t B, int32_t L, int32_t Vx, int32_t pad_token,
                             float* row_loss, float* loss_sum, float* count, float* dscores, void* stream) {
    using namespace ick;
    ICK_CHECK_ARG(scores && captions_sorted && decode_len && row_loss && loss_sum && count);
    ICK_CHECK_ARG(B > 0 && B <= 65535 && L > 0 && Vx > 0 && ld >= Vx);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(packed_ce_rows_kernel, dim3(L, B), dim3(256), 0, s, scores, ld, captions_sorted, decode_len, L,
                       Vx, pad_token, row_loss, dscores);
    hipLaunchKernelGGL(packed_ce_reduce_kernel, dim3(1), dim3(256), 0, s, row_loss, B * L, loss_sum, count);
    ICK_LAUNCH_RET();
}
