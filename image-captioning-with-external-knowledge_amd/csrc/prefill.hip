// Prefill gathers of the caption decoder: entity / fact encoders, caption embedding (+ scale
// + positional encoding) and the knowledge-variant context indicators.  These replace the
// reference's per-sample Python loops and CPU-only Tensor.apply_ callbacks
// (geo-aware/models.py:82-104,143-181; knowledge-aware/models.py:82-133,170-188,209-259,380-418;
// news-knowledge-aware/models.py:79-134) with one launch each.  All are HBM-bound row gathers:
// one wave per 300-float output row, lanes stride the row so every load/store is coalesced.
#include "common.h"

namespace ick {
namespace {

// ---------------------------------------------------------------------------------------------
// EntityEncoder.forward
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void entity_encode_kernel(int variant, const float* __restrict__ ent, int cols,
                                                            const int64_t* __restrict__ facts,
                                                            const float* __restrict__ type_emb, int ntypes,
                                                            const float* __restrict__ word_emb, int vocab,
                                                            float* __restrict__ out, int B, int K, int F, int d) {
    chain_priority();
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B * K) return;
    const int b = row / K, k = row - b * K;
    const float* e = ent + (int64_t)row * cols;
    float* o = out + (int64_t)row * d;
    const int type_off = variant == ICK_GEO ? 4 : (variant == ICK_KNOWLEDGE ? 6 : 5);

    // fact count of this entity (knowledge-aware/models.py:101-128); <unk_ent> (last row) -> 0
    float count = 0.f;
    if (variant != ICK_GEO) {
        int c = 0;
        if (k != K - 1) {
            const int64_t* fb = facts + (int64_t)b * F * 3;
            for (int j = lane; j < F; j += 64) c += (fb[j * 3 + 1] == (int64_t)k) ? 1 : 0;
        }
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) c += __shfl_xor(c, s, 64);
        count = (float)c;
    }
    float slot[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (variant == ICK_NEWS) {
        slot[0] = e[1]; slot[1] = e[2]; slot[2] = e[3];
        slot[3] = count; slot[4] = count > 0.f ? 1.f : 0.f;
    } else {
        // get_dist_to_north / get_dist_to_east run in Python floats (double) in the reference
        const double az = (double)e[2];
        const double north = fabs(az) / 180.0;
        const double east = (az >= -90.0 ? fabs(90.0 - az) : 90.0 + fabs(az + 180.0)) / 180.0;
        slot[0] = e[1]; slot[1] = (float)north; slot[2] = (float)east; slot[3] = e[3];
        if (variant == ICK_KNOWLEDGE) { slot[4] = count; slot[5] = count > 0.f ? 1.f : 0.f; }
    }
    int ty = (int)e[4];
    ty = ty < 0 ? 0 : (ty >= ntypes ? ntypes - 1 : ty);
    const float* trow = type_emb + (int64_t)ty * (d - type_off);

    int name[5] = {0, 0, 0, 0, 0};
    if (variant == ICK_NEWS) {
#pragma unroll
        for (int w = 0; w < 5; ++w) {
            int n = (int)e[5 + w];
            name[w] = n < 0 ? 0 : (n >= vocab ? vocab - 1 : n);
        }
    }
    for (int c = lane; c < d; c += 64) {
        float v;
        if (c < type_off) {
            v = slot[0];
#pragma unroll
            for (int q = 1; q < 6; ++q) v = (c == q) ? slot[q] : v;
        } else {
            v = trow[c - type_off];
        }
        if (variant == ICK_NEWS) {
            // torch.mean over the 5 name-word rows: sequential sum, then divide
            float s = word_emb[(int64_t)name[0] * d + c];
#pragma unroll
            for (int w = 1; w < 5; ++w) s = __fadd_rn(s, word_emb[(int64_t)name[w] * d + c]);
            v = __fmul_rn(v, __fdiv_rn(s, 5.0f));
        }
        o[c] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// FactEncoder.forward: subject row + predicate embedding
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fact_encode_kernel(const int64_t* __restrict__ facts,
                                                          const float* __restrict__ ee,
                                                          const float* __restrict__ pred_emb, int num_pred,
                                                          float* __restrict__ out, int B, int K, int F, int d) {
    chain_priority();
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B * F) return;
    const int b = row / F;
    int subj = (int)facts[(int64_t)row * 3 + 1];
    int pred = (int)facts[(int64_t)row * 3 + 2];
    subj = subj < 0 ? 0 : (subj >= K ? K - 1 : subj);
    pred = pred < 0 ? 0 : (pred >= num_pred ? num_pred - 1 : pred);
    const float* s = ee + ((int64_t)b * K + subj) * d;
    const float* p = pred_emb + (int64_t)pred * d;
    float* o = out + (int64_t)row * d;
    for (int c = lane; c < d; c += 64) o[c] = __fadd_rn(s[c], p[c]);
}

// ---------------------------------------------------------------------------------------------
// CaptionEmbedder.forward, * sqrt(emb_dim), + positional encoding
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void caption_embed_kernel(const int64_t* __restrict__ captions,
                                                            const int64_t* __restrict__ masks,
                                                            const float* __restrict__ word_emb,
                                                            const float* __restrict__ ee,
                                                            const float* __restrict__ fe, const float* __restrict__ pe,
                                                            float* __restrict__ out, float* __restrict__ emb_out, int B,
                                                            int L, int K, int F, int V, int d, int pad_token,
                                                            float scale, int pos0, DropArg darg) {
    chain_priority();
    const Dropout drop = darg.get();
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B * L) return;
    const int b = row / L, l = row - b * L;
    const int64_t tok = captions[row];
    const int64_t m = masks[row];
    const float* src;
    if (m == 1) {
        int64_t ei = tok - V;
        if (ei < 0 || ei >= K) ei = K - 1;  // not an entity pointer -> <unk_ent>, the last row
        src = ee + ((int64_t)b * K + ei) * d;
    } else if (m == 2 && fe != nullptr) {
        int64_t fi = tok - V - K;
        if (fi < 0 || fi >= F) fi = F - 1;  // -> <unk_fact>
        src = fe + ((int64_t)b * F + fi) * d;
    } else {
        int64_t w = tok >= V ? (int64_t)pad_token : tok;
        if (w < 0) w = pad_token;
        src = word_emb + w * d;
    }
    const float* per = pe + (int64_t)(pos0 + l) * d;
    float* o = out + (int64_t)row * d;
    for (int c = lane; c < d; c += 64) {
        const float v = src[c];
        if (emb_out) emb_out[(int64_t)row * d + c] = v;
        float y = __fadd_rn(__fmul_rn(v, scale), per[c]);
        if (drop.on()) y *= drop.mask((uint32_t)row * (uint32_t)d + (uint32_t)c);   // PositionEncoder dropout (training)
        o[c] = y;
    }
}

// ---------------------------------------------------------------------------------------------
// get_context_indicators, fused with fc_predicate: one workgroup per sample.
//   act[j]  = first caption position at which fact j's subject counts as "already mentioned"
//   rep[j]  = fact j is the first (by act, then index) fact carrying its predicate
//   gate[p] = bias + sum over rep facts with act <= p of fc_predicate.weight[:, pred]   (W^T rows)
// ---------------------------------------------------------------------------------------------
constexpr int kInf = 0x3fffffff;

__global__ __launch_bounds__(256) void context_indicators_kernel(const int64_t* __restrict__ captions,
                                                                 const int64_t* __restrict__ facts,
                                                                 const float* __restrict__ wt,
                                                                 const float* __restrict__ bias, float* __restrict__ eib,
                                                                 float* __restrict__ gate, int L, int T, int K, int F,
                                                                 int V, int num_pred, int d, int mode) {
    extern __shared__ int sm[];
    int* first = sm;        // K
    int* act = first + K;   // F
    int* pred = act + F;    // F
    int* rep = pred + F;    // F
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int k = tid; k < K; k += 256) first[k] = kInf;
    __syncthreads();
    for (int t = tid; t < L; t += 256) {
        const int64_t n = captions[(int64_t)b * L + t] - V;
        if (n >= 0 && n < K) atomicMin(&first[(int)n], t);
    }
    __syncthreads();
    for (int j = tid; j < F; j += 256) {
        const int64_t subj = facts[((int64_t)b * F + j) * 3 + 1];
        const int64_t q = facts[((int64_t)b * F + j) * 3 + 2];
        int a = kInf;
        if (subj >= 0 && subj < K && first[(int)subj] < kInf) a = mode == 0 ? first[(int)subj] + 1 : 0;
        act[j] = a;
        pred[j] = (q >= 0 && q < num_pred) ? (int)q : -1;
    }
    __syncthreads();
    for (int j = tid; j < F; j += 256) {
        int r = act[j] < kInf && pred[j] >= 0;
        if (r) {
            for (int i = 0; i < F; ++i) {
                if (i != j && pred[i] == pred[j] && (act[i] < act[j] || (act[i] == act[j] && i < j))) { r = 0; break; }
            }
        }
        rep[j] = r;
    }
    __syncthreads();
    // entity_idx_before (written by the first workgroup of the sample's column split)
    for (int idx = tid; blockIdx.y == 0 && idx < T * F; idx += 256) {
        const int p = idx / F, j = idx - p * F;
        eib[((int64_t)b * T + p) * F + j] = act[j] <= p ? 1.f : 0.f;
    }
    // gate rows, running over positions: the row only changes at the (at most F) positions where a representative
    // fact becomes active, so the facts are ranked by (act, index) once and the running sum walks that list
    // (same summation order as a position-by-position, fact-by-fact scan, ~T + F steps instead of T * F)
    if (gate) {
        int* ord_act = rep + F;     // F
        int* ord_pred = ord_act + F;  // F
        int* nrep = ord_pred + F;   // 1
        if (tid == 0) *nrep = 0;
        __syncthreads();
        for (int j = tid; j < F; j += 256) {
            if (rep[j] && act[j] < T) {
                int rank = 0;
                for (int i = 0; i < F; ++i)
                    if (rep[i] && act[i] < T && (act[i] < act[j] || (act[i] == act[j] && i < j))) ++rank;
                ord_act[rank] = act[j];
                ord_pred[rank] = pred[j];
                atomicAdd(nrep, 1);
            }
        }
        __syncthreads();
        const int n = *nrep;
        for (int c = blockIdx.y * 256 + tid; c < d; c += 256 * gridDim.y) {
            // wt == nullptr: the dense predicate indicator itself (d == num_pred columns, identity "weight", no bias)
            float acc = wt ? bias[c] : 0.f;
            int r = 0;
            for (int p = 0; p < T; ++p) {
                while (r < n && ord_act[r] <= p) { acc += wt ? wt[(int64_t)ord_pred[r] * d + c] : (ord_pred[r] == c ? 1.f : 0.f); ++r; }
                gate[((int64_t)b * T + p) * d + c] = acc;
            }
        }
    }
}

// mask[r, c] = keep ? 1/(1-p) : 0 of element index r * cols + c (test / inspection helper)
__global__ __launch_bounds__(256) void dropout_mask_kernel(float* __restrict__ out, int64_t rows, int cols, Dropout drop) {
    const int64_t n = rows * cols;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) out[i] = drop.mask((uint32_t)i);
}

__global__ __launch_bounds__(256) void mul_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                  float* __restrict__ y, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) y[i] = a[i] * b[i];
}

}  // namespace
}  // namespace ick

extern "C" int ick_entity_encode(int32_t variant, const float* entities, int32_t ent_cols, const int64_t* facts,
                                 const float* type_emb, int32_t ntypes, const float* word_emb, int32_t vocab,
                                 float* out, int32_t B, int32_t K, int32_t F, int32_t d, void* stream) {
    using namespace ick;
    ICK_CHECK_ARG(entities && type_emb && out && B > 0 && K > 0 && d > 6);
    ICK_CHECK_ARG(variant == ICK_GEO || variant == ICK_KNOWLEDGE || variant == ICK_NEWS);
    ICK_CHECK_ARG(ent_cols >= (variant == ICK_NEWS ? 10 : 5));
    if (variant != ICK_GEO) ICK_CHECK_ARG(facts && F > 0);
    if (variant == ICK_NEWS) ICK_CHECK_ARG(word_emb && vocab > 0);
    hipLaunchKernelGGL(entity_encode_kernel, dim3(ceil_div((int64_t)B * K, 4)), dim3(256), 0, (hipStream_t)stream,
                       variant, entities, ent_cols, facts, type_emb, ntypes, word_emb, vocab, out, B, K, F, d);
    ICK_LAUNCH_RET();
}

extern "C" int ick_fact_encode(const int64_t* facts, const float* entities_encoded, const float* pred_emb,
                               int32_t num_pred, float* out, int32_t B, int32_t K, int32_t F, int32_t d,
                               void* stream) {
    using namespace ick;
    ICK_CHECK_ARG(facts && entities_encoded && pred_emb && out && B > 0 && K > 0 && F > 0 && d > 0 && num_pred > 0);
    hipLaunchKernelGGL(fact_encode_kernel, dim3(ceil_div((int64_t)B * F, 4)), dim3(256), 0, (hipStream_t)stream, facts,
                       entities_encoded, pred_emb, num_pred, out, B, K, F, d);
    ICK_LAUNCH_RET();
}

extern "C" int ick_caption_embed(const int64_t* captions, const int64_t* masks, const float* word_emb,
                                 const float* entities_encoded, const float* facts_encoded, const float* pe,
                                 float* out, float* emb_out, int32_t B, int32_t L, int32_t K, int32_t F, int32_t V,
                                 int32_t d, int32_t pad_token, float scale, int32_t pos0, float drop_p,
                                 uint32_t drop_seed, uint32_t drop_site, const uint32_t* drop_epoch, void* stream) {
    using namespace ick;
    ICK_CHECK_ARG(captions && masks && word_emb && entities_encoded && pe && out);
    ICK_CHECK_ARG(B > 0 && L > 0 && K > 0 && V > 0 && d > 0 && pos0 >= 0);
    if (facts_encoded) ICK_CHECK_ARG(F > 0);
    hipLaunchKernelGGL(caption_embed_kernel, dim3(ceil_div((int64_t)B * L, 4)), dim3(256), 0, (hipStream_t)stream,
                       captions, masks, word_emb, entities_encoded, facts_encoded, pe, out, emb_out, B, L, K, F, V, d,
                       pad_token, scale, pos0, DropArg{drop_p, drop_seed, drop_site, drop_epoch});
    ICK_LAUNCH_RET();
}

extern "C" int ick_context_indicators(const int64_t* captions, const int64_t* facts, const float* fc_pred_wt,
                                      const float* fc_pred_b, float* eib, float* gate, int32_t B, int32_t L,
                                      int32_t T, int32_t K, int32_t F, int32_t V, int32_t num_pred, int32_t d,
                                      int32_t mode, void* stream) {
    using namespace ick;
    ICK_CHECK_ARG(captions && facts && eib && B > 0 && L > 0 && K > 0 && F > 0);
    ICK_CHECK_ARG((mode == 0 && T == L) || (mode == 1 && T == 1));
    if (gate) ICK_CHECK_ARG(num_pred > 0 && d > 0 && ((fc_pred_wt && fc_pred_b) || (!fc_pred_wt && d == num_pred)));
    const size_t smem = (size_t)(K + 5 * F + 1) * sizeof(int);
    ICK_CHECK_ARG(smem <= 64 * 1024);
    const int csplit = gate ? std::max(1, std::min(4, ceil_div(d, 256))) : 1;   // workgroups per sample (gate columns)
    hipLaunchKernelGGL(context_indicators_kernel, dim3(B, csplit), dim3(256), smem, (hipStream_t)stream, captions, facts,
                       fc_pred_wt, fc_pred_b, eib, gate, L, T, K, F, V, num_pred, d, mode);
    ICK_LAUNCH_RET();
}

extern "C" int ick_mul(const float* a, const float* b, float* y, int64_t n, void* stream) {
    using namespace ick;
    ICK_CHECK_ARG(a && b && y && n > 0);
    const int grid = (int)std::min<int64_t>(ceil_div(n, 256), 2048);
    hipLaunchKernelGGL(mul_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a, b, y, n);
    ICK_LAUNCH_RET();
}

extern "C" int ick_dropout_mask(float* out, int64_t rows, int32_t cols, float p, uint32_t seed, uint32_t site,
                                void* stream) {
    using namespace ick;
    ICK_CHECK_ARG(out && rows > 0 && cols > 0 && p >= 0.f && p < 1.f && rows * cols < ((int64_t)1 << 32));
    const int grid = (int)std::min<int64_t>(ceil_div(rows * cols, 256), 2048);
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, out, rows, cols,
                       make_dropout(p, seed, site));
    ICK_LAUNCH_RET();
}
