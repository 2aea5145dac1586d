// Backward kernels of the training step (loss.backward() through the
// decoder of geo-aware/train.py:282-292, then clip_gradient + Adam.step).  The GEMM-shaped
// gradients (data / weight gradients of every Linear) reuse ick_gemm with k-major operands; this
// file holds the rest: LayerNorm, ReLU mask, bias column sums, the gather/scatter backward of the
// embedding and encoder stages, the pointer-score head, the predicate gate, and the fused
// clamp + Adam update.
#include "common.h"

namespace ick {
namespace {

// ---------------------------------------------------------------------------------------------
// LayerNorm backward for y = LN(z) * gamma + beta with z = x + res (recomputed here):
//   zh = (z - mean) * rstd;  g = dy * gamma
//   dz = rstd * (g - mean_d(g) - zh * mean_d(g * zh));   dgamma += sum_rows dy * zh;  dbeta += sum_rows dy
// One wave per row; per-workgroup partial dgamma/dbeta are combined in LDS and then either written to
// row blockIdx.x of `partials` (nblocks x 2d; the caller column-sums it off the critical path -- 160
// workgroups hammering the same 600 addresses with float atomics cost more than the rest of the kernel)
// or, without a workspace, added to dgamma/dbeta with one float atomic per column per workgroup.
// ---------------------------------------------------------------------------------------------
template <int NJ>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ res,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, float* __restrict__ dz,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                            int64_t rows, int d, int rows_per_block,
                                                            float* __restrict__ dx_drop, DropArg darg,
                                                            float* __restrict__ partials) {
    chain_priority();
    extern __shared__ float sm[];  // 4 waves x 2 * d partial sums, added in wave order (no LDS atomics: deterministic)
    const Dropout drop = darg.get();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float ag[NJ], ab[NJ], gm[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int c = lane + 64 * j;
        ag[j] = 0.f; ab[j] = 0.f;
        gm[j] = c < d ? gamma[c] : 0.f;
    }
    const int64_t row0 = (int64_t)blockIdx.x * rows_per_block;
    // two rows per wave and iteration: all loads of both rows are in flight before either is reduced (the
    // kernel is bound by memory latency: 160 workgroups x 4 waves, ~15 small loads per row)
    for (int r = wave; r < rows_per_block; r += 8) {
        int64_t rows2[2] = {row0 + r, row0 + r + 4};
        bool live[2] = {rows2[0] < rows, r + 4 < rows_per_block && rows2[1] < rows};
        float zv[2][NJ], dyv[2][NJ], mu[2], rs[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int64_t row = live[q] ? rows2[q] : row0;   // row0 < rows: a valid address for dead slots
            mu[q] = mean[row]; rs[q] = rstd[row];
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int c = lane + 64 * j;
                float z = 0.f, dv = 0.f, rv = 0.f;
                if (c < d) {
                    z = x[row * d + c];
                    if (res) rv = res[row * d + c];
                    dv = dy[row * d + c];
                }
                if (drop.on()) z *= drop.mask((uint32_t)row * (uint32_t)d + (uint32_t)c);
                zv[q][j] = z + rv;
                dyv[q][j] = dv;
            }
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            if (!live[q]) continue;     // wave-uniform
            const int64_t row = rows2[q];
            float zh[NJ], g[NJ];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int c = lane + 64 * j;
                zh[j] = c < d ? (zv[q][j] - mu[q]) * rs[q] : 0.f;
                g[j] = dyv[q][j] * gm[j];
                s1 += g[j];
                s2 += g[j] * zh[j];
                ag[j] += dyv[q][j] * zh[j];
                ab[j] += dyv[q][j];
            }
            s1 = wave_sum(s1) / (float)d;
            s2 = wave_sum(s2) / (float)d;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int c = lane + 64 * j;
                if (c < d) {
                    const float v = rs[q] * (g[j] - s1 - zh[j] * s2);
                    dz[row * d + c] = v;
                    if (dx_drop) dx_drop[row * d + c] = drop.on() ? v * drop.mask((uint32_t)row * (uint32_t)d + (uint32_t)c) : v;
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int c = lane + 64 * j;
        if (c < d) {
            sm[wave * 2 * d + c] = ag[j];
            sm[wave * 2 * d + d + c] = ab[j];
        }
    }
    __syncthreads();
    if (partials) {
        float* pr = partials + (int64_t)blockIdx.x * 2 * d;
        for (int i = threadIdx.x; i < 2 * d; i += 256) pr[i] = (sm[i] + sm[2 * d + i]) + (sm[4 * d + i] + sm[6 * d + i]);
        return;
    }
    for (int i = threadIdx.x; i < d; i += 256) {
        atomicAdd(dgamma + i, (sm[i] + sm[2 * d + i]) + (sm[4 * d + i] + sm[6 * d + i]));
        atomicAdd(dbeta + i, (sm[d + i] + sm[3 * d + i]) + (sm[5 * d + i] + sm[7 * d + i]));
    }
}

// dpre = dpost where the forward activation was positive (ReLU), in place or out of place
__global__ __launch_bounds__(256) void relu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ act,
                                                       float* __restrict__ dx, int64_t n, float scale) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
        dx[i] = act[i] > 0.f ? dy[i] * scale : 0.f;
}

// out[n] += sum_m a[m, n]   (bias gradients).  grid (column blocks of 64, row slabs); each wave owns
// 64 columns x a run of rows, waves of a workgroup are combined in LDS, slabs by float atomics.
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ a, int64_t M, int N, int64_t ld,
                                                     float* __restrict__ out, int rows_per_block) {
    __shared__ float sm[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    const int64_t r1 = min(M, r0 + rows_per_block);
    float s = 0.f;
    if (c < N)
        for (int64_t r = r0 + wave; r < r1; r += 4) s += a[r * ld + c];
    sm[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && c < N) atomicAdd(out + c, sm[0][lane] + sm[1][lane] + sm[2][lane] + sm[3][lane]);
}

// ---------------------------------------------------------------------------------------------
// CaptionEmbedder backward: dx (B,L,d) * scale is added to the row each token was read from
// (word embedding / encoded entity / encoded fact) -- mirrors ick_caption_embed's selection.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void caption_embed_bwd_kernel(const float* __restrict__ dx,
                                                                const int64_t* __restrict__ captions,
                                                                const int64_t* __restrict__ masks,
                                                                float* __restrict__ dword, float* __restrict__ dee,
                                                                float* __restrict__ dfe, int B, int L, int K, int F,
                                                                int V, int d, int pad_token, float scale,
                                                                DropArg darg) {
    chain_priority();
    const Dropout drop = darg.get();
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B * L) return;
    const int b = row / L;
    const int64_t tok = captions[row];
    const int64_t m = masks[row];
    float* dst;
    if (m == 1) {
        int64_t ei = tok - V;
        if (ei < 0 || ei >= K) ei = K - 1;
        dst = dee + ((int64_t)b * K + ei) * d;
    } else if (m == 2 && dfe != nullptr) {
        int64_t fi = tok - V - K;
        if (fi < 0 || fi >= F) fi = F - 1;
        dst = dfe + ((int64_t)b * F + fi) * d;
    } else {
        if (dword == nullptr) return;
        int64_t w = tok >= V ? (int64_t)pad_token : tok;
        if (w < 0) w = pad_token;
        dst = dword + w * d;
    }
    const float* src = dx + (int64_t)row * d;
    for (int c = lane; c < d; c += 64) {
        float v = src[c] * scale;
        if (drop.on()) v *= drop.mask((uint32_t)row * (uint32_t)d + (uint32_t)c);   // PositionEncoder dropout
        // positions past the caption's end carry exactly zero gradient (nothing after them is in the loss) and all
        // point at the <pad> row: skipping zero addends removes a ~400-way atomic pile-up on that row
        if (v != 0.f) atomicAdd(dst + c, v);
    }
}


// ---------------------------------------------------------------------------------------------
// Deterministic forms of the scatter-adds above (ick_set_deterministic / ICK_DETERMINISTIC=1).  A float atomic lets the
// hardware pick the order in which a destination row receives its terms; here the SOURCE rows are walked in index order
// by every workgroup, and workgroup p applies only the terms whose destination row r has r % kDetParts == p: one
// thread owns one column of a destination row for the whole launch, so its terms arrive in source order.
// ---------------------------------------------------------------------------------------------
constexpr int kDetParts = 32;

__global__ __launch_bounds__(256) void caption_embed_bwd_det_kernel(const float* __restrict__ dx,
                                                                    const int64_t* __restrict__ captions,
                                                                    const int64_t* __restrict__ masks,
                                                                    float* __restrict__ dword, float* __restrict__ dee,
                                                                    float* __restrict__ dfe, int B, int L, int K, int F,
                                                                    int V, int d, int pad_token, float scale,
                                                                    DropArg darg) {
    const Dropout drop = darg.get();
    const int p = blockIdx.x, c = blockIdx.y * 256 + threadIdx.x;
    for (int row = 0; row < B * L; ++row) {
        const int b = row / L;
        const int64_t tok = captions[row];
        const int64_t m = masks[row];
        float* dst;
        int64_t key;
        if (m == 1) {
            int64_t ei = tok - V;
            if (ei < 0 || ei >= K) ei = K - 1;
            key = (int64_t)b * K + ei;
            dst = dee + key * d;
        } else if (m == 2 && dfe != nullptr) {
            int64_t fi = tok - V - K;
            if (fi < 0 || fi >= F) fi = F - 1;
            key = (int64_t)b * F + fi;
            dst = dfe + key * d;
            key += 11;
        } else {
            if (dword == nullptr) continue;
            int64_t w = tok >= V ? (int64_t)pad_token : tok;
            if (w < 0) w = pad_token;
            key = w + 23;
            dst = dword + w * d;
        }
        if ((int)(key % kDetParts) != p || c >= d) continue;       // uniform per workgroup except the column guard
        float v = dx[(int64_t)row * d + c] * scale;
        if (drop.on()) v *= drop.mask((uint32_t)row * (uint32_t)d + (uint32_t)c);
        if (v != 0.f) dst[c] += v;
    }
}

__global__ __launch_bounds__(256) void entity_encode_bwd_det_kernel(int variant, const float* __restrict__ dee,
                                                                    const float* __restrict__ ent, int cols,
                                                                    const float* __restrict__ ee,
                                                                    const float* __restrict__ word_emb, int vocab,
                                                                    float* __restrict__ dtype_emb, int ntypes,
                                                                    float* __restrict__ dword, int B, int K, int d) {
    const int p = blockIdx.x, c = blockIdx.y * 256 + threadIdx.x;
    const int type_off = variant == ICK_GEO ? 4 : (variant == ICK_KNOWLEDGE ? 6 : 5);
    for (int row = 0; row < B * K; ++row) {
        const float* e = ent + (int64_t)row * cols;
        int ty = (int)e[4];
        ty = ty < 0 ? 0 : (ty >= ntypes ? ntypes - 1 : ty);
        int name[5] = {0, 0, 0, 0, 0};
        bool mine = ty % kDetParts == p;
        if (variant == ICK_NEWS) {
#pragma unroll
            for (int w = 0; w < 5; ++w) {
                int n = (int)e[5 + w];
                name[w] = n < 0 ? 0 : (n >= vocab ? vocab - 1 : n);
                mine = mine || (dword != nullptr && name[w] % kDetParts == p);
            }
        }
        if (!mine || c >= d) continue;
        float g = dee[(int64_t)row * d + c];
        if (variant == ICK_NEWS) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < 5; ++w) s += word_emb[(int64_t)name[w] * d + c];
            const float avg = s / 5.0f;
            const float enc_g = g * avg;
            if (dword) {
                const float enc = avg != 0.f ? ee[(int64_t)row * d + c] / avg : 0.f;
                const float gw = g * enc / 5.0f;
#pragma unroll
                for (int w = 0; w < 5; ++w)
                    if (name[w] % kDetParts == p) dword[(int64_t)name[w] * d + c] += gw;
            }
            g = enc_g;
        }
        if (c >= type_off && ty % kDetParts == p) dtype_emb[(int64_t)ty * (d - type_off) + (c - type_off)] += g;
    }
}

__global__ __launch_bounds__(256) void fact_encode_bwd_det_kernel(const float* __restrict__ dfe,
                                                                  const int64_t* __restrict__ facts, float* __restrict__ dee,
                                                                  float* __restrict__ dpred, int num_pred, int B, int K,
                                                                  int F, int d) {
    const int p = blockIdx.x, c = blockIdx.y * 256 + threadIdx.x;
    for (int row = 0; row < B * F; ++row) {
        const int b = row / F;
        int subj = (int)facts[(int64_t)row * 3 + 1];
        int pred = (int)facts[(int64_t)row * 3 + 2];
        subj = subj < 0 ? 0 : (subj >= K ? K - 1 : subj);
        pred = pred < 0 ? 0 : (pred >= num_pred ? num_pred - 1 : pred);
        const bool e_mine = (b * K + subj) % kDetParts == p, p_mine = pred % kDetParts == p;
        if ((!e_mine && !p_mine) || c >= d) continue;
        const float g = dfe[(int64_t)row * d + c];
        if (e_mine) dee[((int64_t)b * K + subj) * d + c] += g;
        if (p_mine) dpred[(int64_t)pred * d + c] += g;
    }
}

// ---------------------------------------------------------------------------------------------
// Pointer-score backward.  s[b,t,k] = ind * sum_d h*ctx*w + bias
//   dh[b,t,:]   += sum_k ds*ind * ctx[b,k,:] * w        (one workgroup per (b,t))
//   dctx[b,k,:] += sum_t ds*ind * h[b,t,:] * w          (one workgroup per (b,k))
//   dw          += sum ds*ind * h * ctx ;  dbias += sum ds
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pointer_bwd_dh_kernel(const float* __restrict__ ds, int64_t ds_ld, int col0,
                                                             const float* __restrict__ ctx, const float* __restrict__ w,
                                                             const float* __restrict__ ind, float* __restrict__ dh,
                                                             int T, int Kc, int d) {
    chain_priority();
    const int b = blockIdx.y, t = blockIdx.x;
    const float* dsr = ds + ((int64_t)b * T + t) * ds_ld + col0;
    for (int c = threadIdx.x; c < d; c += 256) {
        float acc = 0.f;
        const float* cc = ctx + (int64_t)b * Kc * d + c;
        int k = 0;
        for (; k + 8 <= Kc; k += 8) {        // eight context rows in flight
            float v[8], g[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                v[q] = cc[(int64_t)(k + q) * d];
                g[q] = dsr[k + q];
                if (ind) g[q] *= ind[((int64_t)b * T + t) * Kc + k + q];
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) acc = fmaf(g[q], v[q], acc);
        }
        for (; k < Kc; ++k) {
            float g = dsr[k];
            if (ind) g *= ind[((int64_t)b * T + t) * Kc + k];
            acc = fmaf(g, cc[(int64_t)k * d], acc);
        }
        dh[((int64_t)b * T + t) * d + c] += acc * w[c];
    }
}

// One workgroup per (sample, 64 columns): the (T x Kc) score gradients of the sample sit in LDS, lane <-> column,
// the four waves share the Kc context rows.  d w and d bias leave the workgroup as one float atomic per column /
// one per sample (a workgroup per (sample, row) made 1280 workgroups fight over the same 300 addresses: 59 us).
// SEQ (deterministic mode): one workgroup per column block walks the samples in order and adds its d w / d bias terms
// with plain read-modify-writes instead of one float atomic per sample.
template <bool SEQ>
__global__ __launch_bounds__(256) void pointer_bwd_dctx_kernel(const float* __restrict__ ds, int64_t ds_ld, int col0,
                                                               const float* __restrict__ h, const float* __restrict__ ctx,
                                                               const float* __restrict__ w, const float* __restrict__ ind,
                                                               float* __restrict__ dctx, float* __restrict__ dw,
                                                               float* __restrict__ dbias, int B, int T, int Kc, int d) {
    chain_priority();
    extern __shared__ float gs[];   // T * Kc gradients (indicator applied), 4 floats of scratch, 4 x 64 partial d w
    float* red = gs + T * Kc;
    float* dwp = red + 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int b = SEQ ? 0 : blockIdx.x; b < (SEQ ? B : (int)blockIdx.x + 1); ++b) {
    float bsum = 0.f;
    for (int idx = threadIdx.x; idx < T * Kc; idx += 256) {
        const int t = idx / Kc, k = idx - t * Kc;
        float g = ds[((int64_t)b * T + t) * ds_ld + col0 + k];
        bsum += g;
        if (ind) g *= ind[((int64_t)b * T + t) * Kc + k];
        gs[idx] = g;
    }
    bsum = block_sum<4>(bsum, red);   // contains the barriers that publish gs
    if (threadIdx.x == 0 && blockIdx.y == 0) { if (SEQ) dbias[0] += bsum; else atomicAdd(dbias, bsum); }
    const int c = blockIdx.y * 64 + lane;
    const bool ok = c < d;
    const float* hb = h + (int64_t)b * T * d + (ok ? c : 0);
    const float wc = ok ? w[c] : 0.f;
    float dwl = 0.f;
    for (int k = wave; k < Kc; k += 4) {
        float acc = 0.f;
        for (int t = 0; t < T; ++t) acc = fmaf(gs[t * Kc + k], hb[(int64_t)t * d], acc);
        if (ok) {
            const int64_t o = ((int64_t)b * Kc + k) * d + c;
            dctx[o] += acc * wc;
            dwl = fmaf(acc, ctx[o], dwl);
        }
    }
    dwp[wave * 64 + lane] = dwl;
    __syncthreads();
    if (wave == 0 && ok) {
        const float t = (dwp[lane] + dwp[64 + lane]) + (dwp[128 + lane] + dwp[192 + lane]);
        if (SEQ) dw[c] += t; else atomicAdd(dw + c, t);
    }
    if (SEQ) __syncthreads();          // the LDS buffers are rewritten for the next sample
  }
}

// EntityEncoder backward: only the type embedding is trainable (feature slots are inputs).  News:
// e = enc * avg(name words)  =>  d enc = de * avg ; d word_emb[name_w] += de * enc / 5.
__global__ __launch_bounds__(256) void entity_encode_bwd_kernel(int variant, const float* __restrict__ dee,
                                                                const float* __restrict__ ent, int cols,
                                                                const float* __restrict__ ee,
                                                                const float* __restrict__ word_emb, int vocab,
                                                                float* __restrict__ dtype_emb, int ntypes,
                                                                float* __restrict__ dword, int B, int K, int d) {
    chain_priority();
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B * K) return;
    const float* e = ent + (int64_t)row * cols;
    const int type_off = variant == ICK_GEO ? 4 : (variant == ICK_KNOWLEDGE ? 6 : 5);
    int ty = (int)e[4];
    ty = ty < 0 ? 0 : (ty >= ntypes ? ntypes - 1 : ty);
    int name[5] = {0, 0, 0, 0, 0};
    if (variant == ICK_NEWS) {
#pragma unroll
        for (int w = 0; w < 5; ++w) {
            int n = (int)e[5 + w];
            name[w] = n < 0 ? 0 : (n >= vocab ? vocab - 1 : n);
        }
    }
    for (int c = lane; c < d; c += 64) {
        float g = dee[(int64_t)row * d + c];
        if (variant == ICK_NEWS) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < 5; ++w) s += word_emb[(int64_t)name[w] * d + c];
            const float avg = s / 5.0f;
            // enc = ee / avg is not recoverable when avg == 0; recompute enc's trainable part instead
            const float enc_g = g * avg;                       // gradient wrt the un-scaled encoding
            if (dword) {
                // d avg = g * enc ; enc = ee / avg when avg != 0, and the product ee is what we stored
                const float enc = avg != 0.f ? ee[(int64_t)row * d + c] / avg : 0.f;
                const float gw = g * enc / 5.0f;
#pragma unroll
                for (int w = 0; w < 5; ++w) atomicAdd(dword + (int64_t)name[w] * d + c, gw);
            }
            g = enc_g;
        }
        if (c >= type_off) atomicAdd(dtype_emb + (int64_t)ty * (d - type_off) + (c - type_off), g);
    }
}

// FactEncoder backward: f = ee[subject] + pred_emb[predicate]
__global__ __launch_bounds__(256) void fact_encode_bwd_kernel(const float* __restrict__ dfe,
                                                              const int64_t* __restrict__ facts, float* __restrict__ dee,
                                                              float* __restrict__ dpred, int num_pred, int B, int K,
                                                              int F, int d) {
    chain_priority();
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B * F) return;
    const int b = row / F;
    int subj = (int)facts[(int64_t)row * 3 + 1];
    int pred = (int)facts[(int64_t)row * 3 + 2];
    subj = subj < 0 ? 0 : (subj >= K ? K - 1 : subj);
    pred = pred < 0 ? 0 : (pred >= num_pred ? num_pred - 1 : pred);
    for (int c = lane; c < d; c += 64) {
        const float g = dfe[(int64_t)row * d + c];
        atomicAdd(dee + ((int64_t)b * K + subj) * d + c, g);
        atomicAdd(dpred + (int64_t)pred * d + c, g);
    }
}

// Predicate-gate backward (knowledge variants): gate[b,p,:] = bias + sum_{distinct active preds} W[:, pred]
// => dW[:, pred] += sum over positions where pred is active of dgate[b,p,:];  dbias += sum dgate.
// Same activation / representative logic as context_indicators_kernel (prefill.hip).
constexpr int kInf = 0x3fffffff;
template <bool SEQ>
__global__ __launch_bounds__(256) void context_gate_bwd_kernel(const int64_t* __restrict__ captions,
                                                               const int64_t* __restrict__ facts,
                                                               const float* __restrict__ dgate, float* __restrict__ dw,
                                                               float* __restrict__ dbias, int B, int L, int T, int K, int F,
                                                               int V, int num_pred, int d, int mode) {
    extern __shared__ int smi[];
    int* first = smi;
    int* act = first + K;
    int* pred = act + F;
    int* rep = pred + F;
    const int tid = threadIdx.x;
  for (int b = SEQ ? 0 : blockIdx.x; b < (SEQ ? B : (int)blockIdx.x + 1); ++b) {
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
        if (r)
            for (int i = 0; i < F; ++i)
                if (i != j && pred[i] == pred[j] && (act[i] < act[j] || (act[i] == act[j] && i < j))) { r = 0; break; }
        rep[j] = r;
    }
    __syncthreads();
    // suffix sums of dgate over positions, once per column (registers walk p = T-1 .. 0, LDS keeps suf[p]): a predicate
    // active from position a gets suf[a] = sum_{p >= a} dgate[b,p,c]  (was: T loads per (fact, column) pair)
    float* suf = reinterpret_cast<float*>(rep + F);     // T * 256 floats
    const int c = blockIdx.y * 256 + tid;
    if (c < d) {
        float run = 0.f;
        for (int p = T - 1; p >= 0; --p) {
            run += dgate[((int64_t)b * T + p) * d + c];
            suf[p * 256 + tid] = run;
        }
        if (SEQ) dbias[c] += run; else atomicAdd(dbias + c, run);
        for (int j = 0; j < F; ++j) {
            if (!rep[j] || act[j] >= T) continue;
            float* dst = dw + (int64_t)c * num_pred + pred[j];             // fc_predicate.weight is (d, num_pred)
            if (SEQ) *dst += suf[act[j] * 256 + tid]; else atomicAdd(dst, suf[act[j] * 256 + tid]);
        }
    }
    if (SEQ) __syncthreads();          // the LDS tables are rebuilt for the next sample
  }
}

// ---------------------------------------------------------------------------------------------
// clip_gradient (geo-aware/utils.py:75-85) + torch.optim.Adam.step (default betas/eps, no weight
// decay, no amsgrad) over one flat fp32 parameter bucket:
//   g = clamp(g * gscale [/ *gscale_den], -clip, clip);  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2
//   p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adam_clamp_kernel(float* __restrict__ p, float* __restrict__ g,
                                                         float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                         float gscale, float clip, float lr, float b1, float b2,
                                                         float eps, int step0, const uint32_t* step_ptr,
                                                         const float* __restrict__ gscale_den) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    if (gscale_den) {
        // a (global) batch without a single contributing token has no mean loss: leave parameters and moments alone
        // instead of spreading 0 * inf = NaN over every weight
        if (!(gscale_den[0] > 0.f)) return;
        gscale = gscale / gscale_den[0];
    }
    const float t = (float)(step0 + (step_ptr ? (int)*step_ptr : 0));
    const float bc1 = 1.f - powf(b1, t);
    const float bc2_sqrt = sqrtf(1.f - powf(b2, t));
    const float step = lr / bc1;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        float gi = g[i] * gscale;
        if (clip > 0.f) gi = fminf(fmaxf(gi, -clip), clip);
        g[i] = gi;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] -= step * mi / (sqrtf(vi) / bc2_sqrt + eps);
    }
}

// The same update on float4: seven 16-byte streams per lane (the scalar form reaches 0.66 of the HBM rate, 320 MB per
// cfg2 step).  n4 = n / 4 elements of float4; the launcher sends a tail of n % 4 to the scalar kernel.
__global__ __launch_bounds__(256) void adam_clamp_vec4_kernel(float4* __restrict__ p, float4* __restrict__ g,
                                                              float4* __restrict__ m, float4* __restrict__ v, int64_t n4,
                                                              float gscale, float clip, float lr, float b1, float b2,
                                                              float eps, int step0, const uint32_t* step_ptr,
                                                              const float* __restrict__ gscale_den) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    if (gscale_den) {
        if (!(gscale_den[0] > 0.f)) return;
        gscale = gscale / gscale_den[0];
    }
    const float t = (float)(step0 + (step_ptr ? (int)*step_ptr : 0));
    const float bc1 = 1.f - powf(b1, t);
    const float bc2_sqrt = sqrtf(1.f - powf(b2, t));
    const float step = lr / bc1;
    const float c1 = 1.f - b1, c2 = 1.f - b2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        float4 gi = g[i];
        const float4 mo = m[i], vo = v[i];
        float4 pi = p[i];
        float ga[4] = {gi.x, gi.y, gi.z, gi.w};
        const float ma[4] = {mo.x, mo.y, mo.z, mo.w}, va[4] = {vo.x, vo.y, vo.z, vo.w};
        float pa[4] = {pi.x, pi.y, pi.z, pi.w}, mn[4], vn[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {      // the arithmetic of adam_clamp_kernel, operation for operation
            float x = ga[k] * gscale;
            if (clip > 0.f) x = fminf(fmaxf(x, -clip), clip);
            ga[k] = x;
            mn[k] = b1 * ma[k] + c1 * x;
            vn[k] = b2 * va[k] + c2 * x * x;
            pa[k] -= step * mn[k] / (sqrtf(vn[k]) / bc2_sqrt + eps);
        }
        g[i] = make_float4(ga[0], ga[1], ga[2], ga[3]);
        m[i] = make_float4(mn[0], mn[1], mn[2], mn[3]);
        v[i] = make_float4(vn[0], vn[1], vn[2], vn[3]);
        p[i] = make_float4(pa[0], pa[1], pa[2], pa[3]);
    }
}

__global__ void counter_add_kernel(uint32_t* c, uint32_t inc, const float* flag) {
    if (flag == nullptr || flag[0] > 0.f) *c += inc;
}

__global__ __launch_bounds__(256) void scale_kernel(float* __restrict__ x, int64_t n, const float* __restrict__ num,
                                                    const float* __restrict__ den) {
    const float s = num[0] / den[0];
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) x[i] *= s;
}

}  // namespace
}  // namespace ick

using namespace ick;

extern "C" int ick_layernorm_bwd_rows_per_block(void) { return 8; }

extern "C" int ick_layernorm_bwd(const float* dy, const float* x, const float* res, const float* gamma,
                                 const float* mean, const float* rstd, float* dz, float* dgamma, float* dbeta,
                                 int64_t rows, int32_t d, float* dx_drop, float drop_p, uint32_t drop_seed,
                                 uint32_t drop_site, const uint32_t* drop_epoch, float* partials, void* stream) {
    ICK_CHECK_ARG(dy && x && gamma && mean && rstd && dz && rows > 0 && d > 0 && d <= 1024);
    ICK_CHECK_ARG(partials || (dgamma && dbeta));
    ICK_CHECK_ARG(partials || !deterministic());       // without a workspace the workgroups meet in float atomics
    ICK_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || dx_drop != nullptr));
    const int rpb = ick_layernorm_bwd_rows_per_block();   // two rows per wave: 160 workgroups for the 1280 rows of a layer (one row per wave measured slower: 10.0 vs 8.4 us)
    const DropArg dr{drop_p, drop_seed, drop_site, drop_epoch};
    float* dxd = dx_drop;   // written whenever given: dz * mask, or a plain copy of dz without dropout
    const dim3 grid(ceil_div(rows, rpb));
    const size_t sm = 8 * d * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
    if (d <= 320)
        hipLaunchKernelGGL(layernorm_bwd_kernel<5>, grid, dim3(256), sm, s, dy, x, res, gamma, mean, rstd, dz, dgamma,
                           dbeta, rows, d, rpb, dxd, dr, partials);
    else if (d <= 512)
        hipLaunchKernelGGL(layernorm_bwd_kernel<8>, grid, dim3(256), sm, s, dy, x, res, gamma, mean, rstd, dz, dgamma,
                           dbeta, rows, d, rpb, dxd, dr, partials);
    else
        hipLaunchKernelGGL(layernorm_bwd_kernel<16>, grid, dim3(256), sm, s, dy, x, res, gamma, mean, rstd, dz, dgamma,
                           dbeta, rows, d, rpb, dxd, dr, partials);
    ICK_LAUNCH_RET();
}

extern "C" int ick_relu_bwd(const float* dy, const float* act, float* dx, int64_t n, float scale, void* stream) {
    ICK_CHECK_ARG(dy && act && dx && n > 0);
    hipLaunchKernelGGL(relu_bwd_kernel, dim3((int)std::min<int64_t>(ceil_div(n, 256), 2048)), dim3(256), 0,
                       (hipStream_t)stream, dy, act, dx, n, scale);
    ICK_LAUNCH_RET();
}

extern "C" int ick_colsum(const float* a, int64_t M, int32_t N, int64_t ld, float* out, void* stream) {
    ICK_CHECK_ARG(a && out && M > 0 && N > 0 && ld >= N);
    const int rpb = deterministic() ? (int)std::min<int64_t>(M, 1 << 30) : 64;     // one slab: no atomics between slabs
    ICK_CHECK_ARG(ceil_div(M, rpb) <= 65535);
    hipLaunchKernelGGL(colsum_kernel, dim3(ceil_div(N, 64), ceil_div(M, rpb)), dim3(256), 0, (hipStream_t)stream, a, M,
                       N, ld, out, rpb);
    ICK_LAUNCH_RET();
}

extern "C" int ick_caption_embed_bwd(const float* dx, const int64_t* captions, const int64_t* masks, float* dword,
                                     float* dee, float* dfe, int32_t B, int32_t L, int32_t K, int32_t F, int32_t V,
                                     int32_t d, int32_t pad_token, float scale, float drop_p, uint32_t drop_seed,
                                     uint32_t drop_site, const uint32_t* drop_epoch, void* stream) {
    ICK_CHECK_ARG(dx && captions && masks && dee && B > 0 && L > 0 && K > 0 && d > 0);
    if (deterministic())
        hipLaunchKernelGGL(caption_embed_bwd_det_kernel, dim3(kDetParts, ceil_div(d, 256)), dim3(256), 0,
                           (hipStream_t)stream, dx, captions, masks, dword, dee, dfe, B, L, K, F, V, d, pad_token, scale,
                           DropArg{drop_p, drop_seed, drop_site, drop_epoch});
    else
        hipLaunchKernelGGL(caption_embed_bwd_kernel, dim3(ceil_div((int64_t)B * L, 4)), dim3(256), 0, (hipStream_t)stream,
                           dx, captions, masks, dword, dee, dfe, B, L, K, F, V, d, pad_token, scale,
                           DropArg{drop_p, drop_seed, drop_site, drop_epoch});
    ICK_LAUNCH_RET();
}

extern "C" int ick_pointer_scores_bwd(const float* ds, int64_t ds_ld, int32_t col0, const float* h, const float* ctx,
                                      const float* w, const float* ind, float* dh, float* dctx, float* dw,
                                      float* dbias, int32_t B, int32_t T, int32_t Kc, int32_t d, void* stream) {
    ICK_CHECK_ARG(ds && h && ctx && w && dh && dctx && dw && dbias && B > 0 && B <= 65535 && T > 0 && Kc > 0 && d > 0);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(pointer_bwd_dh_kernel, dim3(T, B), dim3(256), 0, s, ds, ds_ld, col0, ctx, w, ind, dh, T, Kc, d);
    const size_t smem = ((size_t)T * Kc + 4 + 256) * sizeof(float);
    ICK_CHECK_ARG(smem <= 64 * 1024);
    if (deterministic())
        hipLaunchKernelGGL(pointer_bwd_dctx_kernel<true>, dim3(1, ceil_div(d, 64)), dim3(256), smem, s, ds, ds_ld, col0, h,
                           ctx, w, ind, dctx, dw, dbias, B, T, Kc, d);
    else
        hipLaunchKernelGGL(pointer_bwd_dctx_kernel<false>, dim3(B, ceil_div(d, 64)), dim3(256), smem, s, ds, ds_ld, col0, h,
                           ctx, w, ind, dctx, dw, dbias, B, T, Kc, d);
    ICK_LAUNCH_RET();
}

extern "C" int ick_entity_encode_bwd(int32_t variant, const float* dee, const float* entities, int32_t ent_cols,
                                     const float* ee, const float* word_emb, int32_t vocab, float* dtype_emb,
                                     int32_t ntypes, float* dword, int32_t B, int32_t K, int32_t d, void* stream) {
    ICK_CHECK_ARG(dee && entities && dtype_emb && B > 0 && K > 0 && d > 6);
    if (variant == ICK_NEWS) ICK_CHECK_ARG(ee && word_emb && vocab > 0);
    if (deterministic())
        hipLaunchKernelGGL(entity_encode_bwd_det_kernel, dim3(kDetParts, ceil_div(d, 256)), dim3(256), 0,
                           (hipStream_t)stream, variant, dee, entities, ent_cols, ee, word_emb, vocab, dtype_emb, ntypes,
                           dword, B, K, d);
    else
        hipLaunchKernelGGL(entity_encode_bwd_kernel, dim3(ceil_div((int64_t)B * K, 4)), dim3(256), 0, (hipStream_t)stream,
                           variant, dee, entities, ent_cols, ee, word_emb, vocab, dtype_emb, ntypes, dword, B, K, d);
    ICK_LAUNCH_RET();
}

extern "C" int ick_fact_encode_bwd(const float* dfe, const int64_t* facts, float* dee, float* dpred, int32_t num_pred,
                                   int32_t B, int32_t K, int32_t F, int32_t d, void* stream) {
    ICK_CHECK_ARG(dfe && facts && dee && dpred && B > 0 && K > 0 && F > 0 && d > 0);
    if (deterministic())
        hipLaunchKernelGGL(fact_encode_bwd_det_kernel, dim3(kDetParts, ceil_div(d, 256)), dim3(256), 0, (hipStream_t)stream,
                           dfe, facts, dee, dpred, num_pred, B, K, F, d);
    else
        hipLaunchKernelGGL(fact_encode_bwd_kernel, dim3(ceil_div((int64_t)B * F, 4)), dim3(256), 0, (hipStream_t)stream,
                           dfe, facts, dee, dpred, num_pred, B, K, F, d);
    ICK_LAUNCH_RET();
}

extern "C" int ick_context_gate_bwd(const int64_t* captions, const int64_t* facts, const float* dgate, float* dw,
                                    float* dbias, int32_t B, int32_t L, int32_t T, int32_t K, int32_t F, int32_t V,
                                    int32_t num_pred, int32_t d, int32_t mode, void* stream) {
    ICK_CHECK_ARG(captions && facts && dgate && dw && dbias && B > 0 && L > 0 && K > 0 && F > 0);
    ICK_CHECK_ARG((mode == 0 && T == L) || (mode == 1 && T == 1));
    const size_t smem = (size_t)(K + 3 * F) * sizeof(int) + (size_t)T * 256 * sizeof(float);
    ICK_CHECK_ARG(smem <= 64 * 1024);
    if (deterministic())
        hipLaunchKernelGGL(context_gate_bwd_kernel<true>, dim3(1, ceil_div(d, 256)), dim3(256), smem, (hipStream_t)stream,
                           captions, facts, dgate, dw, dbias, B, L, T, K, F, V, num_pred, d, mode);
    else
        hipLaunchKernelGGL(context_gate_bwd_kernel<false>, dim3(B, ceil_div(d, 256)), dim3(256), smem, (hipStream_t)stream,
                           captions, facts, dgate, dw, dbias, B, L, T, K, F, V, num_pred, d, mode);
    ICK_LAUNCH_RET();
}

extern "C" int ick_adam_clamp(float* p, float* g, float* m, float* v, int64_t n, float gscale, float clip, float lr,
                              float beta1, float beta2, float eps, int32_t step, const uint32_t* step_ptr,
                              const float* gscale_den, void* stream) {
    ICK_CHECK_ARG(p && g && m && v && n > 0 && (step >= 1 || step_ptr != nullptr));
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    int64_t done = 0;
    if (n >= 4096 && al16(p) && al16(g) && al16(m) && al16(v)) {
        const int64_t n4 = n / 4;
        hipLaunchKernelGGL(adam_clamp_vec4_kernel, dim3((int)std::min<int64_t>(ceil_div(n4, 256), 4096)), dim3(256), 0,
                           (hipStream_t)stream, reinterpret_cast<float4*>(p), reinterpret_cast<float4*>(g),
                           reinterpret_cast<float4*>(m), reinterpret_cast<float4*>(v), n4, gscale, clip, lr, beta1, beta2,
                           eps, step, step_ptr, gscale_den);
        done = 4 * n4;
        if (done == n) ICK_LAUNCH_RET();
    }
    hipLaunchKernelGGL(adam_clamp_kernel, dim3((int)std::min<int64_t>(ceil_div(n - done, 256), 4096)), dim3(256), 0,
                       (hipStream_t)stream, p + done, g + done, m + done, v + done, n - done, gscale, clip, lr, beta1,
                       beta2, eps, step, step_ptr, gscale_den);
    ICK_LAUNCH_RET();
}

extern "C" int ick_scale_by_ratio(float* x, int64_t n, const float* num, const float* den, void* stream) {
    ICK_CHECK_ARG(x && num && den && n > 0);
    hipLaunchKernelGGL(scale_kernel, dim3((int)std::min<int64_t>(ceil_div(n, 256), 4096)), dim3(256), 0,
                       (hipStream_t)stream, x, n, num, den);
    ICK_LAUNCH_RET();
}

extern "C" int ick_counter_add(uint32_t* counter, uint32_t inc, void* stream) {
    ICK_CHECK_ARG(counter != nullptr);
    hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, counter, inc, (const float*)nullptr);
    ICK_LAUNCH_RET();
}

extern "C" int ick_counter_add_if(uint32_t* counter, uint32_t inc, const float* flag, void* stream) {
    ICK_CHECK_ARG(counter != nullptr && flag != nullptr);
    hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, counter, inc, flag);
    ICK_LAUNCH_RET();
}
