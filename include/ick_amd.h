/*
 * ick_amd.h -- C ABI of libick_amd.so: the MI355X (gfx950) caption-decoder hot path of
 * sonniki/image-captioning-with-external-knowledge.
 *
 * The reference has no FFI layer: its boundary for this path is the Python API of
 * <variant>/models.py.  The entry points below are the op boundaries of
 * SURVEY.md §8(a) rows a1-a14 that the package's Python `models` shim binds through ctypes;
 * each comment names the reference call site it replaces (paths relative to the reference
 * root).  Conventions:
 *   - every pointer is a DEVICE pointer to fp32 / int32 / int64 data owned by the caller
 *     (PyTorch's caching allocator in the shipped binding); nothing is allocated or freed here;
 *   - every entry takes the hipStream_t (as void*) to launch on and returns 0 on success,
 *     a negative ICK_E* code for argument errors, or a positive hipError_t;
 *   - all activations are batch-major, row-major fp32: (B, T, d) with d contiguous;
 *   - no entry synchronises the stream or the device.
 */
#ifndef ICK_AMD_H
#define ICK_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ICK_OK 0
#define ICK_EINVAL (-1)   /* bad shape / null pointer / unsupported size */
#define ICK_EALIGN (-2)   /* pointer or stride violates an alignment requirement */
#define ICK_EWORKSPACE (-3) /* workspace too small */

#define ICK_MAX_LAYERS 8

/* Library / device probe (no reference counterpart). */
int ick_version(void);
/* Deterministic mode: gradient reductions that otherwise use float atomics (scatter-adds of the embedding / entity /
 * fact / predicate-gate backward, bias and LayerNorm column sums) run in a fixed order -- slower, bit-reproducible.
 * Default: the environment variable ICK_DETERMINISTIC (1 = on).  Callers must also keep every GEMM unsplit
 * (split_k = 1) and the step on one stream; training.TrainStep(deterministic=True) does. */
int ick_set_deterministic(int on);
int ick_get_deterministic(void);
int ick_device_info(int* num_cu, int* wave_size, char* arch_name, int arch_name_len);

/* ------------------------------------------------------------------------------------------
 * Generic fp32 GEMM on the matrix cores:  C[m, n] = act( sum_k A(m,k) * B(n,k) + bias[n] )
 * Replaces every nn.Linear / 1x1 nn.Conv2d call site on the path:
 *   Encoder.conv1                         geo-aware/models.py:32,45
 *   MultiheadAttention in/out projections geo-aware/models.py:241-244 (torch.nn internals)
 *   linear1 / linear2 of the layers       geo-aware/models.py:241-244
 *   fc_vocab, fc_predicate                geo-aware/models.py:303; knowledge-aware/models.py:436-438
 * A element (m,k) lives at A[a_goff(m) + (m % a_grp) * a_rs + k * a_ks] with
 *   a_goff(m) = (a_gmap ? a_gmap[m / a_grp] : m / a_grp) * a_gs      (a_grp == 0: one group)
 * B element (n,k) at B[n * b_rs + k * b_ks];  C row m starts at
 *   C + (c_gmap ? c_gmap[m / c_grp] : m / c_grp) * c_gs + (m % c_grp) * c_rs.
 * Exactly one of (a_rs, a_ks) and one of (b_rs, b_ks) must be 1.
 * flags: bit0 = ReLU, bit1 = accumulate into C (C += ...), bit2 = atomic accumulate (split-K).
 * Head-split epilogue (hs_dh > 0): the N columns are [segment][head][hs_dh] (packed q|k|v or
 * k|v-per-layer projections) and are scattered into the attention kernel's head-major layout
 *   C[ goff(m) + seg*(hs_H*hs_S*hs_dhp) + head*(hs_S*hs_dhp) + (hs_s0 + m % c_grp)*hs_dhp + j ]
 * i.e. per sample [segment][head][position][hs_dhp] with rows padded to hs_dhp floats (pad columns
 * are not written).  c_rs is ignored in this mode.
 */
typedef struct {
    const float* A; const float* B; float* C; const float* bias;
    int32_t M, N, K;
    int64_t a_rs, a_ks; int32_t a_grp; int64_t a_gs; const int32_t* a_gmap;
    int64_t b_rs, b_ks;
    int64_t c_rs; int32_t c_grp; int64_t c_gs; const int32_t* c_gmap;
    int32_t flags;
    int32_t split_k;       /* >1: K is split over blockIdx.z, partial sums added atomically */
    float   alpha;         /* scales the product before bias (1.0f normally) */
    int32_t hs_dh, hs_dhp, hs_H, hs_S, hs_s0;   /* head-split epilogue, see above (hs_dh == 0: off) */
    float drop_p; uint32_t drop_seed, drop_site; /* training dropout on the output (after bias/ReLU), see ick_dropout_mask */
    const uint32_t* drop_epoch;                  /* optional device counter added to drop_seed (graph replays) */
    int64_t a_extent, b_extent;   /* floats addressable from A / B (bounds of the buffer descriptors); 0 = derive from
                                     the strides -- required when a_gmap is given, otherwise the slow path is taken */
    float* colsum_a;              /* optional, k-major A only: colsum_a[m] += sum_k A(m,k) (float atomics) -- the bias
                                     gradient of a Linear rides on its weight-gradient GEMM (A = dY^T) */
    const float* gate;            /* optional (plain row-major C only): C[m,n] = gate[m*gate_rs + n] > 0 ? v * gate_scale : 0,
                                     applied last -- the ReLU (+ dropout) backward of the FFN rides on the data-gradient
                                     GEMM of linear2 (gate = the saved activation, gate_scale = 1/(1-p)) */
    int64_t gate_rs; float gate_scale;
    const void* b_ps;             /* optional: the pre-split copy of B (ick_presplit_weights of the same (N, K) matrix).  With
                                     it, and the split-bf16 product mode on, large problems run on the kernel that stages both
                                     operands by LDS-DMA and splits only the A fragments (csrc/gemm_ps.hip); B itself must
                                     still be valid (the other kernels read it).  The caller refreshes the copy whenever B
                                     changes (training: once per step, in the packing launches) */
} ick_gemm_args;

#define ICK_GEMM_RELU 1
#define ICK_GEMM_ACCUM 2
#define ICK_GEMM_ATOMIC 4
#define ICK_GEMM_COLSUM_ONLY 16   /* no product: colsum_a[m] += sum_k A(m,k) for a k-major A (B, C, N ignored) -- lets plain
                                    column sums (LayerNorm gamma/beta partials) ride in an ick_gemm_grouped launch */

int ick_gemm(const ick_gemm_args* args, void* stream);
/* The kernel configuration ick_gemm would launch for `args` (nothing is launched): workgroup tile, waves per
 * workgroup, tile grid, K split, operand layouts, 16-byte vector staging.  The parity tests assert through it that
 * the instantiation a benchmark quotes is the one they compared with the oracle. */
typedef struct {
    int32_t tile_m, tile_n, waves, tiles_m, tiles_n, split_k, a_kmajor, b_kmajor, vec;
    int32_t split_bf16;   /* 1: products formed as six bf16 x bf16 partial products of the exact three-way bf16 split of
                           * both fp32 operands, accumulated in fp32 (64x64 / 128x64 tiles); 0: v_mfma_f32_16x16x4_f32 */
    int32_t presplit;     /* 1: the B operand is read from its pre-split copy (b_ps): 64 x 320 or 128 x 128 tiles of
                           * csrc/gemm_ps.hip, both operands staged by LDS-DMA */
} ick_gemm_plan_info;
int ick_gemm_plan(const ick_gemm_args* args, ick_gemm_plan_info* out);
/* Process-wide mode of the split-bf16 product path of the large GEMM tiles.  0: every product on the exact fp32 MFMA
 * (v_mfma_f32_16x16x4_f32).  1 (DEFAULT; environment ICK_GEMM_SPLIT overrides): split products where they are faster (B
 * operand k-contiguous or pre-split: forward GEMMs, Encoder.conv1, the vocabulary data gradient).  2: for every large-tile
 * problem.  The split is exact (x = hi + mid + lo in bf16), six of the
 * nine partial products are formed; error against fp64 no larger than the exact path's (tests/test_gemm_split_gpu.py). */
int ick_set_gemm_split(int mode);
int ick_get_gemm_split(void);
/* Pre-split copies of weight matrices for ick_gemm's b_ps: the exact three-way bf16 split (hi + mid + lo) of the (N, K)
 * matrix whose element (n, k) is src[n * src_rs + k * src_cs] (one of the two strides must be 1: a weight or its
 * transposed view), stored as dst[K slice of 32][plane hi | mid | lo][row n, padded to a multiple of 64][32 k] bf16 with
 * zeros beyond N and K -- the image the matrix cores' B operand is read in.  ick_presplit_bytes gives the size of dst
 * (16-byte aligned).  One launch for up to 16 matrices.  Call sites: the weights of Encoder.conv1, of the all-layer cross
 * K/V projection and of fc_vocab (and fc_vocab's transposed view for its data gradient), geo-aware/models.py:32,241-242,303. */
typedef struct {
    const float* src; void* dst;
    int32_t N, K;
    int64_t src_rs, src_cs;
} ick_presplit_item;
int ick_presplit_bytes(int32_t N, int32_t K, int64_t* bytes);
int ick_presplit_weights(const ick_presplit_item* items, int32_t count, void* stream);
/* `count` (<= 64) independent problems; those that select the same kernel configuration share one launch (up to 8
 * per launch).  Used for the weight-gradient GEMMs of a layer (geo-aware/train.py:284 `loss.backward()`), each of
 * which alone is a latency-bound launch of a few hundred workgroups. */
int ick_gemm_grouped(const ick_gemm_args* problems, int32_t count, void* stream);

/* y[r,:] = LayerNorm(x[r,:] + res[r,:]) * gamma + beta   (res may be NULL), d <= 1024.
 * Replaces norm1/norm2/norm3 + the residual adds of Transformer{En,De}coderLayer
 * (torch/nn/modules/transformer.py post-LN path; built at geo-aware/models.py:241-244).
 * Optional outputs mean/rstd (rows) are kept for the backward pass. */
int ick_add_layernorm(const float* x, const float* res, const float* gamma, const float* beta,
                      float* y, int64_t rows, int32_t d, float eps,
                      int64_t x_ld, int64_t res_ld, int64_t y_ld,
                      float* save_mean, float* save_rstd,
                      float drop_p, uint32_t drop_seed, uint32_t drop_site, /* dropout applied to x (dropout1/2/3) */
                      const uint32_t* drop_epoch, void* stream);

/* Row-resident chain of a post-LN Transformer block, one launch instead of three (GEMM, add & norm, GEMM):
 *     o  = A W1^T + b1                                        (M x d;   A is M x K1)
 *     x  = LayerNorm(res + dropout1(o)) * gamma + beta         (M x d)
 *     y2 = act(x W2^T + b2), dropout2                          (M x N2;  optional: w2p == NULL stops after x)
 * i.e. what torch's TransformerDecoderLayer / TransformerEncoderLayer forward (post-LN) does between two attention
 * cores for the layers built at geo-aware/models.py:241-244, knowledge-aware/models.py:319-330:
 *   self_attn.out_proj -> dropout1 -> norm1 -> multihead_attn q-projection        (K1 = d,   N2 = d, head-split)
 *   multihead_attn.out_proj -> dropout2 -> norm2 -> linear1 + ReLU + dropout      (K1 = d,   N2 = FF)
 *   linear2 -> dropout3 -> norm3 -> the next layer's self_attn in_proj            (K1 = FF,  N2 = 3 d, head-split)
 * Weights are passed as PACKED copies of the nn.Linear weights (w1p of the (d, K1) weight, w2p of the (N2, d) one),
 * laid out for the kernel's operand loads and refreshed by ick_pack_weights after every optimizer step.
 * Row r of A at A + (r / a_grp) * a_gs + (r % a_grp) * a_rs (a_grp <= 0: r * a_rs); x and y2 likewise; with
 * hs_dh > 0 y2 is scattered head-major exactly like ick_gemm's head-split epilogue (y2_grp / y2_gs = c_grp / c_gs).
 * o, mean, rstd are optional outputs for the backward pass.  Dropout masks are those of ick_add_layernorm (drop1,
 * element index r * d + c) and of ick_gemm (drop2, r * N2 + c).  Limits: K1 <= 512, d <= 320, N2 <= 1024. */
typedef struct {
    const float* A; int64_t a_rs; int32_t a_grp; int64_t a_gs;
    int32_t M, K1, d;
    const float* w1p; const float* b1;
    const float* res; int64_t res_rs;
    const float* gamma; const float* beta; float eps;
    float drop1_p; uint32_t drop_seed, drop1_site; const uint32_t* drop_epoch;
    float* o; int64_t o_rs;
    float* x; int64_t x_rs; int32_t x_grp; int64_t x_gs;
    float* mean; float* rstd;
    const float* w2p; const float* b2; int32_t N2; int32_t flags;   /* flags: ICK_GEMM_RELU, ICK_CHAIN_SLIM */
    float drop2_p; uint32_t drop2_site;
    float* y2; int64_t y2_rs; int32_t y2_grp; int64_t y2_gs;
    int32_t hs_dh, hs_dhp, hs_H, hs_S, hs_s0;
} ick_rowchain_args;
#define ICK_CHAIN_SLIM 256   /* 8-wave workgroups: for chains that run beside bulk GEMMs on another stream (they find
                                room on a busy CU where the 16-wave form waits for the bulk kernel to drain) */
#define ICK_CHAIN_PROJ 512   /* ick_rowchain_fwd: projection only, y2 = act(A W2^T + b2) with A (M, d); w1p / norm arguments unused */
int ick_rowchain_supported(int32_t K1, int32_t d, int32_t N2);
int ick_rowchain_fwd(const ick_rowchain_args* args, void* stream);

/* Backward of the row-resident chains, one launch for the data-gradient path between two attention-backward
 * kernels (what `loss.backward()`, geo-aware/train.py:284, runs for a post-LN Transformer block):
 *     dx        = dzin + g0 W0                  (g0: M x K0 gradient of the Linear that consumed the block's output,
 *                                                W0 its (K0, d) weight; either addend may be absent)
 *     dz1, do1  = LayerNorm'(dx)                 (ick_layernorm_bwd semantics: dz1 = gradient of the normalised sum and of
 *                                                the residual operand, do1 = dz1 * dropout mask = gradient of o1)
 *     t         = do1 W2, zeroed / scaled by the ReLU + dropout gate (ick_gemm's gate)     } FFN blocks only
 *     dx2       = dz1 + t W1                                                                } (w1p != NULL):
 *     dz2, do2  = LayerNorm'(dx2)                                                           } linear2, linear1, norm
 *     out3      = do W3                          (do = do2 for FFN blocks, do1 otherwise; W3 = the out-projection)
 *     dz_out    = dz2 (FFN blocks) or dz1
 * Weights are packed copies (ick_pack_weights) of the TRANSPOSED nn.Linear weights: w0p of W0^T (d x K0), w1p of
 * linear2.weight^T (N1 x d), w2p of linear1.weight^T (d x N1), w3p of out_proj.weight^T (d x d).
 * o / res / mean / rstd / gamma are the saved forward tensors of the norms (as for ick_layernorm_bwd); part1 / part2
 * receive the per-workgroup gamma / beta partial sums, ceil(M / 8) x 2d floats each (ick_layernorm_bwd's layout, 8 rows
 * per workgroup).  All row-major with dense rows (d, N1) except g0 / dzin (g0_rs, dzin_rs).
 * Limits: K0 <= 1920, d <= 320, N1 <= 512. */
typedef struct {
    int32_t M, d;
    uint32_t drop_seed; const uint32_t* drop_epoch;
    const float* g0; int64_t g0_rs; int32_t K0; const float* w0p;
    int32_t g0_grp; int64_t g0_gs;     /* g0_grp > 0: row r of g0 at g0 + (r / g0_grp) * g0_gs + (r % g0_grp) * g0_rs */
    const float* dzin; int64_t dzin_rs;
    const float* o1; const float* res1; const float* mean1; const float* rstd1; const float* gamma1;
    float drop1_p; uint32_t drop1_site;
    float* do1; float* part1;
    const float* w1p; int32_t N1; const float* act; float gate_scale; float* t_out;
    const float* w2p;
    const float* o2; const float* res2; const float* mean2; const float* rstd2; const float* gamma2;
    float drop2_p; uint32_t drop2_site;
    float* do2; float* part2;
    const float* w3p; float* out3;
    float* dz_out;
    int32_t flags;                     /* ICK_CHAIN_SLIM: the 8-wave form (same bits), for launches beside another stream's kernels */
} ick_rowchain_bwd_args;
int ick_rowchain_bwd_supported(int32_t K0, int32_t d, int32_t N1);
int ick_rowchain_bwd(const ick_rowchain_bwd_args* args, void* stream);

/* Packed copy of an (N, K) matrix whose element (n, k) is src[n * src_rs + k * src_cs] (a weight: src_cs = 1; its
 * transpose, for the data-gradient chains: src_rs = 1, src_cs = the weight's row stride):
 *   dst[((slab * (K16 / 4) + k4) * 64 + l) * 4 + kk] = M[64 * slab + l][4 * k4 + kk]   (0 for rows >= N, k >= K),
 * K16 = K rounded up to 16, slabs = ceil(N / 64); dst holds *floats of ick_packed_weight_floats(N, K, &floats)
 * floats, 16-byte aligned.  count <= 48 matrices per launch. */
typedef struct {
    const float* src; float* dst; int32_t N, K; int64_t src_rs, src_cs;
    int64_t dst_rs;   /* 0: the packed layout above.  > 0: a plain copy dst[n * dst_rs + k] = src(n, k) riding in the same
                       * launch (the all-layer cross K/V weight / bias the step gathers from the layers' in_proj) */
} ick_pack_item;
int ick_packed_weight_floats(int32_t N, int32_t K, int64_t* floats);
int ick_pack_weights(const ick_pack_item* items, int32_t count, void* stream);

/* Multi-head attention core: O = softmax(Q K^T * scale [+ causal mask]) V per (batch, head).
 * Q element (b,t,h,j) at Q[b*q_bs + h*q_hs + t*q_ts + j]; K/V element (b,s,h,j) at
 * K[b*k_bs + h*k_hs + s*k_ss + j]; O element at O[b*o_bs + t*o_ts + h*dh + j] (row-major rows for the
 * output projection).  The head-major padded layout written by ick_gemm's head-split epilogue
 * (q_ts == k_ss == v_ss == 32, 16-byte aligned) takes the vectorised path.  kv_len (optional, int32[B]) limits
 * the keys of sample b to s < kv_len[b] (KV-cached greedy decode); q_pos0 is the absolute
 * position of query row 0 for the causal mask (key s visible iff s <= q_pos0 + t).
 * Replaces the scaled-dot-product core of nn.MultiheadAttention for
 *   decoder self-attention (causal), decoder cross-attention over [image ; entity ; fact] memory,
 *   context-encoder self-attention            geo-aware/models.py:348,358; knowledge-aware:495-496,508
 * lse (optional, B*H*T) receives log-sum-exp per query row for the backward pass. */
typedef struct {
    const float* Q; const float* K; const float* V; float* O; float* lse;
    int32_t B, H, T, S, dh;
    int64_t q_bs, q_hs, q_ts, k_bs, k_hs, k_ss, v_bs, v_hs, v_ss, o_bs, o_ts;
    float scale; int32_t causal; int32_t q_pos0; const int32_t* kv_len;
    float drop_p; uint32_t drop_seed, drop_site;   /* attention-weight dropout (training), element index ((b*H+h)*T+t)*S+s */
    const uint32_t* drop_epoch;
} ick_attn_args;
int ick_attention(const ick_attn_args* args, void* stream);

/* ------------------------------------------------------------------------------------------
 * Prefill gathers (rows a2-a5, a11).
 * variant: 0 geo, 1 knowledge, 2 news. */
#define ICK_GEO 0
#define ICK_KNOWLEDGE 1
#define ICK_NEWS 2

/* EntityEncoder.forward: geo-aware/models.py:82-104; knowledge-aware/models.py:82-133;
 * news-knowledge-aware/models.py:79-134.  entities (B,K,cols) fp32, facts (B,F,3) int64 (may be
 * NULL for geo), type_emb (ntypes, d-type_off), word_emb (V,d) (news only) -> out (B,K,d). */
int ick_entity_encode(int32_t variant, const float* entities, int32_t ent_cols, const int64_t* facts,
                      const float* type_emb, int32_t ntypes, const float* word_emb, int32_t vocab,
                      float* out, int32_t B, int32_t K, int32_t F, int32_t d, void* stream);

/* FactEncoder.forward: knowledge-aware/models.py:170-188. */
int ick_fact_encode(const int64_t* facts, const float* entities_encoded, const float* pred_emb,
                    int32_t num_pred, float* out, int32_t B, int32_t K, int32_t F, int32_t d, void* stream);

/* CaptionEmbedder.forward + `* sqrt(emb_dim)` + PositionEncoder (eval):
 * geo-aware/models.py:143-181,355-357,199-209; knowledge-aware/models.py:209-259.
 * captions/masks (B,L) int64; pos0 = absolute position of column 0 (greedy decode steps);
 * pe (max_len, d) sinusoid table.  out (B,L,d) = emb*scale + pe[pos0+l].  emb_out (optional)
 * receives the unscaled embedding rows (fixture stage "embeddings"). */
int ick_caption_embed(const int64_t* captions, const int64_t* masks, const float* word_emb,
                      const float* entities_encoded, const float* facts_encoded, const float* pe,
                      float* out, float* emb_out, int32_t B, int32_t L, int32_t K, int32_t F, int32_t V,
                      int32_t d, int32_t pad_token, float scale, int32_t pos0,
                      float drop_p, uint32_t drop_seed, uint32_t drop_site, /* PositionEncoder dropout (training) */
                      const uint32_t* drop_epoch, void* stream);

/* get_context_indicators: knowledge-aware/models.py:380-418.  Produces, per (b, position p):
 *   eib (B,T,F) 0/1: subject of fact j was mentioned before p (T==L) / anywhere so far (T==1)
 *   gate (B,T,d) = fc_predicate(predicate_indicator) = bias + sum over DISTINCT active predicates
 *   of fc_predicate.weight[:, pred]  (the dense (B,L,3000) indicator and its GEMM,
 *   knowledge-aware/models.py:436, are never materialised).  fc_pred_wt is the TRANSPOSED weight,
 *   (num_pred, d) row-major, so one predicate is one contiguous row.  gate may be NULL.
 * With fc_pred_wt == NULL and d == num_pred, `gate` receives the dense 0/1 predicate indicator (B,T,num_pred)
 *   itself (the public get_context_indicators method returns it).
 * mode 0: teacher-forced (strictly-before semantics), mode 1: predict (whole buffer). */
int ick_context_indicators(const int64_t* captions, const int64_t* facts, const float* fc_pred_wt,
                           const float* fc_pred_b, float* eib, float* gate, int32_t B, int32_t L,
                           int32_t T, int32_t K, int32_t F, int32_t V, int32_t num_pred, int32_t d,
                           int32_t mode, void* stream);

/* Pointer scores of get_scores (geo-aware/models.py:305-310; knowledge-aware/models.py:439-452):
 *   out[b,t,col0+k] = sum_d h[b,t,d]*ctx[b,k,d]*w[d] * (ind ? ind[b,t,k] : 1) + bias[0]
 * written in place into the concatenated score rows (row stride out_ld).  out_gmap (optional,
 * int32[B]) redirects sample b's rows to batch slot out_gmap[b]. */
int ick_pointer_scores(const float* h, const float* ctx, const float* w, const float* bias,
                       const float* ind, float* out, int32_t B, int32_t T, int32_t Kc, int32_t d,
                       int64_t out_ld, int32_t col0, const int32_t* out_gmap, void* stream);

/* Training dropout (nn.Dropout / MultiheadAttention(dropout=p) sites of the reference's layers): every
 * kernel that drops takes (p, seed, site); element `idx` of a site is kept iff
 * murmur3_fmix(idx * 0x9E3779B1 ^ (seed + site * 0x85EBCA6B)) >= p * 2^32 and scaled by 1/(1-p), so the
 * backward kernels regenerate the mask.  This entry writes that mask (rows x cols, idx = r*cols + c). */
int ick_dropout_mask(float* out, int64_t rows, int32_t cols, float p, uint32_t seed, uint32_t site, void* stream);

/* elementwise y = a * b  (vocab_input = h * gate, knowledge-aware/models.py:437) */
int ick_mul(const float* a, const float* b, float* y, int64_t n, void* stream);

/* Greedy selection of predict(): softmax -> argmax and runner-up per row
 * (geo-aware/models.py:410-419).  scores (B, Vx) row stride ld -> best/second int32[B]. */
int ick_top2(const float* scores, int64_t ld, int32_t B, int32_t Vx, int32_t* best, int32_t* second,
             void* stream);

/* One step of predict()'s token bookkeeping for B independent captions, entirely on device
 * (geo-aware/models.py:412-442; knowledge-aware/models.py:575-608): writes output[step],
 * applies the repeated-n-gram clean-up, marks finished captions, and emits the next input
 * token + mask.  output (B, max_len) int64 pre-filled with <pad>; top2_hist (B, max_len) int32;
 * finished (B) int32. */
int ick_greedy_update(const int32_t* best, const int32_t* second, int64_t* output, int32_t* top2_hist,
                      int32_t* finished, int64_t* next_token, int64_t* next_mask, int32_t B,
                      int32_t step, int32_t max_len, int32_t V, int32_t K, int32_t has_facts,
                      int32_t end_token, void* stream);
/* ick_top2 + ick_greedy_update in one launch (one workgroup per caption): the per-token selection of predict(). */
int ick_greedy_select(const float* scores, int64_t ld, int32_t B, int32_t Vx, int64_t* output, int32_t* top2_hist,
                      int32_t* finished, int64_t* next_token, int64_t* next_mask, int32_t step, int32_t max_len,
                      int32_t V, int32_t K, int32_t has_facts, int32_t end_token, void* stream);

/* ------------------------------------------------------------------------------------------
 * Fused decode step of predict() (geo-aware/models.py:389-443; knowledge-aware/models.py:545-608): one token
 * for R independent rows (R = captions x rows_per_sample; rows_per_sample > 1 = beams sharing their caption's
 * memory) in 3 launches per decoder layer + 2 (ick_decode_layers) + 1 (ick_decode_select_*), KV-cached.
 * A block's closing residual + LayerNorm is applied by the NEXT kernel while it loads its input row (see
 * csrc/decode.hip); partial out-projection rows travel through p1 / p2 / p3.  out_wt / w2t are the TRANSPOSED
 * out_proj / linear2 weights ((in_features, d) row-major).  All buffers are caller-owned device memory. */
typedef struct {
    const float *sa_in_w, *sa_in_b, *sa_out_wt, *sa_out_b, *n1_g, *n1_b;   /* self_attn, norm1 */
    const float *ca_in_w, *ca_in_b, *ca_out_wt, *ca_out_b, *n2_g, *n2_b;   /* multihead_attn (q rows used), norm2 */
    const float *w1, *b1, *w2t, *b2, *n3_g, *n3_b;                         /* linear1, linear2, norm3 */
    float *self_k, *self_v;           /* (R, H, max_len, 32) key / value cache of the caption positions */
    const float *cross_k, *cross_v;   /* this layer's key / value segment of the (B, 2*layers, H, S, 32) memory projection */
} ick_decode_layer;

typedef struct {
    int32_t R, rows_per_sample, d, H, FF, layers, S, max_len, V, K, F;
    int32_t end_token, pad_token;
    float ln_eps, emb_scale;          /* LayerNorm eps; sqrt(emb_dim) */
    int64_t kv_bs, scores_ld;         /* sample stride of the memory projection; row stride of `scores` */
    ick_decode_layer layer[ICK_MAX_LAYERS];
    const int32_t* anc;               /* optional (R, max_len): cache row holding position p of row r (beam search) */
    const float *wv, *bv, *we, *be, *wf, *bf;   /* fc_vocab, fc_entity, fc_fact (wf / bf NULL without facts) */
    const float *ee, *fe;             /* entities_encoded (B, K, d), facts_encoded (B, F, d) */
    const float *gate, *eib;          /* knowledge variants: (R, d) predicate gate and (R, F) indicator of this step */
    const float *word_emb, *pe;       /* word embedding (V, d), sinusoid table (>= max_len, d) */
    float *x0;                        /* (R, d) embedded input token of this step; select writes the next one */
    float *xa, *xb, *xc;              /* (R, d) residual rows between the blocks */
    float *p1, *p2, *p3;              /* (R, H, d), (R, H, d), (R, ceil(FF/64), d) partial out-projections */
    float *hfin, *hv;                 /* (R, d) decoder output h and h * gate */
    float *ptr;                       /* (R, K+F) pointer scores */
    float *cand;                      /* (R, ceil(V/16), 4) per-tile top-2 candidates {v1, idx1, v2, idx2} */
    float *scores;                    /* optional (R, scores_ld) vocabulary logits */
    int64_t *output;                  /* (R, max_len) generated tokens, pre-filled with <pad> */
    int32_t *hist, *finished, *n_done;/* (R, max_len) runner-ups; (R) ended flags; (1) number of ended rows */
    int64_t *next_token, *next_mask;  /* (R) */
    int64_t *cap_buf;                 /* optional (R, max_len) caption buffer read by ick_context_indicators */
    int32_t *sel_state;               /* optional (2, R, 12) workspace: with it (greedy, rows_per_sample == 1) ick_decode_layers
                                       * of position pos >= 1 chooses the token of step pos - 1 inside its first launch, and
                                       * ick_decode_select_greedy is only called for the last step */
} ick_decode_ctx;

/* 1 when the fused path handles these sizes (d % 4 == 0, 64 <= d <= 320, head width <= 32, S <= 1024,
 * max_len <= 128); callers fall back to the per-op launches otherwise. */
int ick_decode_supported(int32_t d, int32_t H, int32_t FF, int32_t S, int32_t max_len);
/* Decoder stack + score head for position `pos`: reads x0, leaves ptr / cand (/ scores / hfin). */
int ick_decode_layers(const ick_decode_ctx* ctx, int32_t pos, void* stream);
/* The same in two parts (part 1: the first self-attention block -- with sel_state the selection of the previous step --,
 * part 2: everything behind it; part 0 = ick_decode_layers): the knowledge variants run ick_context_indicators on the
 * caption buffer between the two. */
int ick_decode_layers_part(const ick_decode_ctx* ctx, int32_t pos, int32_t part, void* stream);
/* One launch that initialises every per-call buffer of a decode: output = <pad>, history / flags / windows = 0,
 * caption buffer = <start>, *n_done = n_done_init, x0 = embedding of <start> at position 0. */
int ick_decode_init(const ick_decode_ctx* ctx, int32_t start_token, int32_t n_done_init, void* stream);
/* Greedy selection + predict()'s bookkeeping + embedding of the next input token into x0 (models.py:410-442). */
int ick_decode_select_greedy(const ick_decode_ctx* ctx, int32_t pos, void* stream);

/* Beam selection of one step (beam = ctx->rows_per_sample <= 8 hypotheses per caption, rows b*beam..): every live
 * hypothesis offers log_softmax(scores) + its cumulative log-probability for each of the V+K+F tokens, an ended one
 * offers itself unchanged; the best `beam` candidates of a caption (ties: lower hypothesis, lower token) become its
 * new rows.  Token histories, caption buffers and the cache-ancestry table (which cache row holds position p of row
 * r: ick_decode_ctx.anc of the NEXT step) are copied from the chosen parents (double buffered).  Needs ctx->scores.
 * The reference has no beam search (geo-aware/eval.py:61,83 decodes greedily): parity-unpinned; beam 1 is routed
 * to the pinned greedy path by the Python caller. */
typedef struct {
    float* cum; int32_t* fin;                       /* (R) cumulative log-probability (-inf: unused slot), ended flag */
    const int64_t* seq_in; int64_t* seq_out;        /* (R, max_len) */
    const int32_t* anc_in; int32_t* anc_out;        /* (R, max_len) */
    const int64_t* cap_in; int64_t* cap_out;        /* optional (R, max_len) */
    float* rec;                                     /* workspace (R, ceil((V+K+F)/1024), 18): per-chunk max / sum / best-k records */
    int32_t start_token;
} ick_beam_state;
int ick_decode_select_beam(const ick_decode_ctx* ctx, const ick_beam_state* beam, int32_t pos, void* stream);
/* 1 when ick_decode_select_beam handles `beam` hypotheses over Vx = V+K+F scores (beam <= 8 and
 * beam^2 * ceil(Vx / 1024) <= 4096 candidates); callers check this BEFORE capturing a decode graph. */
int ick_decode_beam_supported(int32_t Vx, int32_t beam);

/* fused token-mean cross entropy over the packed rows of train.py
 * (pack_padded_sequence + CrossEntropyLoss(ignore_index=<pad>), geo-aware/train.py:275-281):
 * rows (b,t) with t < decode_len[b] and target != pad contribute.  Writes loss_sum[0] (sum of
 * token losses), count[0] (number of tokens) and, if dscores != NULL, the UNNORMALISED gradient
 * (softmax - onehot) for contributing rows and zeros elsewhere. */
int ick_packed_ce(const float* scores, int64_t ld, const int64_t* captions_sorted, const int32_t* decode_len,
                  int32_t B, int32_t L, int32_t Vx, int32_t pad_token, float* row_loss /* B*L workspace */,
                  float* loss_sum, float* count, float* dscores, void* stream);


/* ------------------------------------------------------------------------------------------
 * Training step (row a14): backward kernels behind loss.backward() of geo-aware/train.py:282-292,
 * and clip_gradient + Adam.step (geo-aware/utils.py:75-85, train.py:287-292).  Weight and data
 * gradients of the Linear layers are ick_gemm calls with k-major operands.
 */
/* Backward of ick_attention.  Q/K/V: the forward's head-major padded buffers (row stride 32 or 64
 * floats); O, dO row-major with (o_bs, o_ts); lse from the forward.  Outputs are row-major:
 * dQ element (b,t,h,j) at dQ[b*dq_bs + t*dq_ts + h*dh + j], dK/dV element (b,s,h,j) at
 * dK[b*dk_bs + s*dk_ss + h*dh + j].  When the queries do not fit one workgroup's LDS the key/value
 * gradients are accumulated with float atomics and dK/dV must be zero on entry. */
typedef struct {
    const float* Q; const float* K; const float* V; const float* O; const float* dO; const float* lse;
    float* dQ; float* dK; float* dV;
    int32_t B, H, T, S, dh;
    int64_t q_bs, q_hs, q_ts, k_bs, k_hs, k_ss, v_bs, v_hs, v_ss, o_bs, o_ts;
    int64_t dq_bs, dq_ts, dk_bs, dk_ss, dv_bs, dv_ss;
    float scale; int32_t causal; int32_t q_pos0;
    float drop_p; uint32_t drop_seed, drop_site;   /* must equal the forward's */
    const uint32_t* drop_epoch;
} ick_attn_bwd_args;
int ick_attention_bwd(const ick_attn_bwd_args* args, void* stream);
/* 1 when ick_attention_bwd writes every element of dQ / dK / dV for this shape (no pre-zeroing needed), 0 when it
 * accumulates query chunks with float atomics into buffers the caller must have zeroed. */
int ick_attention_bwd_overwrites(int32_t T, int32_t S, int32_t dh);

/* dz = dLN/d(x+res).  Parameter gradients: with `partials` == NULL, dgamma += ..., dbeta += ... (float atomics);
 * otherwise workgroup i writes its partial [dgamma | dbeta] sums to partials[i*2d .. i*2d+2d) for
 * i < ceil(rows / ick_layernorm_bwd_rows_per_block()) and the caller reduces them (ick_colsum), off the critical path. */
int ick_layernorm_bwd_rows_per_block(void);
int ick_layernorm_bwd(const float* dy, const float* x, const float* res, const float* gamma, const float* mean,
                      const float* rstd, float* dz, float* dgamma, float* dbeta, int64_t rows, int32_t d,
                      float* dx_drop /* optional: dz * mask = gradient of the dropped operand x (required when drop_p > 0; a copy of dz when drop_p == 0) */,
                      float drop_p, uint32_t drop_seed, uint32_t drop_site, const uint32_t* drop_epoch, float* partials,
                      void* stream);
/* dx = dy where the forward ReLU output `act` was positive, else 0. */
int ick_relu_bwd(const float* dy, const float* act, float* dx, int64_t n, float scale /* 1/(1-p) of the FFN dropout */,
                 void* stream);
/* out[n] += sum_m a[m*ld + n]  (bias gradients). */
int ick_colsum(const float* a, int64_t M, int32_t N, int64_t ld, float* out, void* stream);
/* Backward of ick_caption_embed: dx*scale is scatter-added to word_emb / entity / fact gradient rows. */
int ick_caption_embed_bwd(const float* dx, const int64_t* captions, const int64_t* masks, float* dword,
                          float* dee, float* dfe, int32_t B, int32_t L, int32_t K, int32_t F, int32_t V,
                          int32_t d, int32_t pad_token, float scale, float drop_p, uint32_t drop_seed,
                          uint32_t drop_site, const uint32_t* drop_epoch, void* stream);
/* Backward of ick_pointer_scores: ds = dscores[..., col0:col0+Kc] (row stride ds_ld);
 * dh += ..., dctx += ..., dw += ..., dbias += ... */
int ick_pointer_scores_bwd(const float* ds, int64_t ds_ld, int32_t col0, const float* h, const float* ctx,
                           const float* w, const float* ind, float* dh, float* dctx, float* dw, float* dbias,
                           int32_t B, int32_t T, int32_t Kc, int32_t d, void* stream);
/* Backward of ick_entity_encode (type embedding; news: also the name-word embeddings). */
int ick_entity_encode_bwd(int32_t variant, const float* dee, const float* entities, int32_t ent_cols,
                          const float* ee, const float* word_emb, int32_t vocab, float* dtype_emb,
                          int32_t ntypes, float* dword, int32_t B, int32_t K, int32_t d, void* stream);
/* Backward of ick_fact_encode: dee[b, subject] += dfe, dpred_emb[predicate] += dfe. */
int ick_fact_encode_bwd(const float* dfe, const int64_t* facts, float* dee, float* dpred, int32_t num_pred,
                        int32_t B, int32_t K, int32_t F, int32_t d, void* stream);
/* Backward of the predicate gate of ick_context_indicators: dgate (B,T,d) ->
 * fc_predicate.weight.grad (d, num_pred) and bias.grad (d), accumulated. */
int ick_context_gate_bwd(const int64_t* captions, const int64_t* facts, const float* dgate, float* dw,
                         float* dbias, int32_t B, int32_t L, int32_t T, int32_t K, int32_t F, int32_t V,
                         int32_t num_pred, int32_t d, int32_t mode, void* stream);
/* g = clamp(g*gscale, +-clip) (clip <= 0: no clamp) followed by torch.optim.Adam's update, one flat
 * fp32 bucket; the 1-based step count is step + *step_ptr (step_ptr may be NULL; a device-resident counter
 * lets a captured graph advance the bias correction on every replay).  With gscale_den (device scalar, may be
 * NULL) the gradient scale is gscale / *gscale_den: the token-mean normalisation by the all-reduced token
 * count without a host round trip or a separate pass over the bucket. */
int ick_adam_clamp(float* p, float* g, float* m, float* v, int64_t n, float gscale, float clip, float lr,
                   float beta1, float beta2, float eps, int32_t step, const uint32_t* step_ptr,
                   const float* gscale_den, void* stream);
/* ick_adam_clamp over the whole bucket which ALSO keeps the re-laid-out copies of the weights current that the
 * next step's kernels read (geo-aware/train.py:292 optimizer.step() is the only writer of the parameters, so the
 * packed / transposed / bf16-plane images need no launches of their own: round 4 spent 126 us of kernel time per cfg2
 * step on ick_pack_weights + ick_presplit_weights).  The bucket is cut into workgroup-sized `blocks`:
 *   - a flat run of <= 1024 float4 (item < 0), optionally mirrored into a plain copy (the gathered cross K/V bias);
 *   - a 64 x 64 tile of an `item`: `rows` x K floats, row-major with row stride K, starting `off` floats into the
 *     bucket, which are rows drow0 .. drow0 + rows - 1 of a logical (Nd, K) matrix W.  The tile (tn, tk) covers rows
 *     64 tn .. 64 tn + 63 of W and columns 64 tk .. 64 tk + 63; after the update its new values are written to every
 *     non-NULL image of W: `pack` = ick_pack_weights(W), `pack_t` = ick_pack_weights(W^T), `copy` = W with row stride
 *     copy_ld, `ps` = ick_presplit_weights(W), `ps_t` = ick_presplit_weights(W^T), `tr` = W^T with row stride tr_ld.
 *     The images must have been filled once by those entry points (their zero padding is never rewritten).
 *     Constraints: K % 4 == 0, off % 4 == 0; pack_t needs drow0 % 4 == 0 and rows % 4 == 0; ps_t needs drow0 % 8 == 0
 *     and rows % 8 == 0; copy_ld % 4 == 0.
 * Every float of [0, n) must lie in exactly one block (the caller builds the cover).  items / blocks are DEVICE arrays.
 * Arithmetic: the operations of ick_adam_clamp in the same order (bit-identical parameters, moments, clamped gradients). */
typedef struct ick_adam_item {
    int64_t off;
    int32_t rows, K, drow0, Nd;
    float* pack;
    float* pack_t;
    float* copy;
    void* ps;
    void* ps_t;
    float* tr;
    int64_t copy_ld, tr_ld;
} ick_adam_item;
typedef struct ick_adam_block {
    int32_t item, tn, tk, cnt4;
    int64_t off4;
    float* copy;
} ick_adam_block;
int ick_adam_clamp_derive(float* p, float* g, float* m, float* v, const ick_adam_item* items,
                          const ick_adam_block* blocks, int32_t n_blocks, float gscale, float clip, float lr,
                          float beta1, float beta2, float eps, int32_t step, const uint32_t* step_ptr,
                          const float* gscale_den, void* stream);
/* *counter += inc on the stream (step / dropout-epoch counter of captured training graphs). */
int ick_counter_add(uint32_t* counter, uint32_t inc, void* stream);
/* ... only if *flag > 0 (device scalar): the step counter of a training step whose optimizer update is deferred into the
 * next step's graph advances exactly when that update is applied -- ick_adam_clamp[_derive] with gscale_den = flag is a
 * no-op under the same condition (no pending gradients: first call, or the update was flushed). */
int ick_counter_add_if(uint32_t* counter, uint32_t inc, const float* flag, void* stream);
/* n <= 8 device-to-device copies (src[i] -> dst[i], bytes[i]) in one launch; host arrays of device pointers. */
int ick_copy_batch(const void* const* src, void* const* dst, const long long* bytes, int n, void* stream);
/* Diagnostic: *out = device wall clock (100 MHz ticks) when the stream reaches this point. */
int ick_timestamp(unsigned long long* out, void* stream);
/* x *= num[0] / den[0] with device-resident scalars (token-mean normalisation without a host sync). */
int ick_scale_by_ratio(float* x, int64_t n, const float* num, const float* den, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ICK_AMD_H */
