"""Per-op parity on the MI355X: each HIP kernel through the C ABI vs the CPU oracle / fp64 math."""
import math

import pytest
import torch

import torch.nn.functional as F

import ick_amd.synth as synth
from oracle import restatement as R

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    import ick_amd.ops as ops
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    return ops


def dev(t):
    return t.cuda()


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def close(got, ref, tol, what=""):
    got = got.detach().cpu().double()
    ref = ref.double()
    err = (got - ref).abs().max().item()
    lim = tol * max(1.0, ref.abs().max().item())
    assert err <= lim, "%s max|err| %.3e > %.3e" % (what, err, lim)


# ------------------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("M,N,K,relu", [
    (1280, 900, 300, False),    # QKV projection (64x64 tiles)
    (1280, 300, 512, False),    # FFN down (32x32 tiles)
    (1280, 512, 300, True),     # FFN up + ReLU
    (2560, 1800, 300, False),   # packed cross K/V projection (128x128 tiles)
    (12544, 300, 300, False),   # tall: 256x64 tiles
    (37, 53, 19, True),         # ragged, scalar staging path
    (130, 66, 301, False),      # K not a multiple of 4
    (1, 10, 300, False),
])
def test_gemm_linear(ops, M, N, K, relu):
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.1), rnd(N, seed=3)
    ref = x.double() @ w.double().t() + b.double()
    if relu:
        ref = ref.relu()
    got = ops.linear(dev(x), dev(w), dev(b), relu=relu)
    close(got, ref, 2e-6 * math.sqrt(K), "linear %s" % ((M, N, K),))


def test_gemm_feat_proj_nchw(ops):
    # Encoder.conv1 on an NCHW map: A is k-major with one group per image
    B, Cc, P, d = 3, 2048, 196, 300
    feats = synth.make_feats(B, seed=7)
    w, b = synth.make_conv1(seed=7)
    ref = R.feat_proj(feats.double(), w.double(), b.double()).permute(0, 2, 1)  # (B,P,d) token-major
    f = dev(feats)
    out = torch.empty(B, P, d, device="cuda")
    ops.gemm_raw(f, dev(w.view(d, Cc)), out, B * P, d, Cc, 1, P, Cc, 1, d, bias=dev(b), a_grp=P, a_gs=Cc * P)
    close(out, ref, 1e-4, "feat_proj")


def test_gemm_vocab_remap_and_ld(ops):
    # vocabulary logits written into (B, L, V+K) rows in permuted batch order
    B, Lc, d, V, K = 5, 7, 300, 1000, 20
    h, w, b = rnd(B, Lc, d, seed=4), rnd(V, d, seed=5, scale=0.1), rnd(V, seed=6)
    perm = torch.tensor([3, 0, 4, 1, 2], dtype=torch.int32)
    out = torch.full((B, Lc, V + K), -7.0, device="cuda")
    ops.gemm_raw(dev(h), dev(w), out, B * Lc, V, d, d, 1, d, 1, V + K, bias=dev(b), c_grp=Lc, c_gs=Lc * (V + K),
                 c_gmap=dev(perm))
    ref = torch.full((B, Lc, V + K), -7.0, dtype=torch.double)
    ref[perm.long(), :, :V] = h.double() @ w.double().t() + b.double()
    close(out, ref, 2e-5, "vocab remap")


def test_gemm_agmap_gather(ops):
    # cross K/V projection reads image rows of sample sort_ind[b]
    B, P, d, N = 4, 196, 300, 600
    enc, w = rnd(B, P, d, seed=8), rnd(N, d, seed=9, scale=0.1)
    sort_ind = torch.tensor([2, 0, 3, 1], dtype=torch.int32)
    S = P + 6
    out = torch.zeros(B, S, N, device="cuda")
    ops.gemm_raw(dev(enc), dev(w), out, B * P, N, d, d, 1, d, 1, N, a_grp=P, a_gs=P * d, a_gmap=dev(sort_ind),
                 c_grp=P, c_gs=S * N)
    ref = torch.zeros(B, S, N, dtype=torch.double)
    ref[:, :P] = enc[sort_ind.long()].double() @ w.double().t()
    close(out, ref, 2e-5, "a_gmap")


def test_gemm_kmajor_operands_and_splitk(ops):
    # dgrad form: dX[M,K'] = dY[M,N'] W[N',K']  (B operand k-major);  wgrad form: dW = dY^T X (both k-major)
    M, N, K = 1280, 900, 300
    dy, w, x = rnd(M, N, seed=10), rnd(N, K, seed=11, scale=0.1), rnd(M, K, seed=12)
    dx = torch.empty(M, K, device="cuda")
    ops.gemm_raw(dev(dy), dev(w), dx, M, K, N, N, 1, 1, K, K)
    close(dx, dy.double() @ w.double(), 1e-5 * math.sqrt(N), "dgrad")
    dw = torch.zeros(N, K, device="cuda")
    ops.gemm_raw(dev(dy), dev(x), dw, N, K, M, 1, N, 1, K, K, atomic=True, split_k=4)
    close(dw, dy.double().t() @ x.double(), 1e-5 * math.sqrt(M), "wgrad split-k")
    dw2 = torch.ones(N, K, device="cuda")
    ops.gemm_raw(dev(dy), dev(x), dw2, N, K, M, 1, N, 1, K, K, accumulate=True, alpha=0.5)
    close(dw2, 1.0 + 0.5 * (dy.double().t() @ x.double()), 1e-5 * math.sqrt(M), "wgrad accumulate")


def test_gemm_grouped_and_colsum_fusion(ops):
    """ick_gemm_grouped: weight-gradient problems of different shapes in one launch (plus one that needs another
    kernel configuration), each with the bias gradient (column sums of the k-major A operand) fused in."""
    shapes = [(1280, 300, 300), (1280, 512, 300), (1280, 300, 512), (1280, 900, 300), (260, 7, 300), (1280, 300, 300),
              (13824, 600, 300), (1284, 10, 6), (1280, 304, 300), (1280, 300, 300)]
    problems, keep, refs = [], [], []
    for i, (M, N, K) in enumerate(shapes):
        dy, x = rnd(M, N, seed=100 + i), rnd(M, K, seed=200 + i)
        dw0, db0 = rnd(N, K, seed=300 + i), rnd(N, seed=400 + i)
        ddy, dx_, dw, db = dev(dy), dev(x), dev(dw0), dev(db0)
        keep += [ddy, dx_, dw, db]
        problems.append(ops.gemm_args(ddy, dx_, dw, N, K, M, 1, N, 1, K, K, atomic=True,
                                      split_k=max(1, min(16, M // 256)), colsum_a=db))
        refs.append((dw, db, dw0.double() + dy.double().t() @ x.double(), db0.double() + dy.double().sum(0)))
    # plain column sums (LayerNorm gamma / beta partials) ride in the same launch
    part = rnd(160, 600, seed=77)
    acc0 = rnd(600, seed=78)
    dpart, dacc, dacc2 = dev(part), dev(acc0), dev(acc0)
    problems.insert(3, ops.colsum_problem(dpart, dacc))
    problems.append(ops.colsum_problem(dpart[:, 300:], dacc2[300:]))
    ops.gemm_grouped(problems)
    close(dacc, acc0.double() + part.double().sum(0), 2e-5, "grouped column sum")
    close(dacc2[300:], acc0[300:].double() + part[:, 300:].double().sum(0), 2e-5, "grouped column sum of a column slice")
    assert torch.equal(dacc2[:300].cpu(), acc0[:300])
    for i, (dw, db, rw, rb) in enumerate(refs):
        close(dw, rw, 2e-4, "grouped dw %d" % i)
        close(db, rb, 2e-4, "grouped db %d" % i)
    # colsum_a needs a k-major A operand
    bad = ops.gemm_args(keep[0], keep[1], keep[2], 1280, 300, 300, 300, 1, 300, 1, 300, colsum_a=keep[3])
    with pytest.raises(Exception):
        ops.gemm_grouped([bad])


def test_gemm_gate_epilogue(ops):
    """The FFN's ReLU (+ dropout) backward fused into the data-gradient GEMM: C = gate > 0 ? (A B^T) * s : 0."""
    M, N, K = 1280, 512, 300
    dy, w = rnd(M, K, seed=1), rnd(K, N, seed=2, scale=0.1)     # dx = dy @ w  (w: (out=K, in=N) row-major)
    act = torch.relu(rnd(M, N, seed=3))
    out = torch.full((M, N), float("nan"), device="cuda")
    ops.gemm_raw(dev(dy), dev(w), out, M, N, K, K, 1, 1, N, N, gate=dev(act), gate_scale=2.0)
    ref = (dy.double() @ w.double()) * 2.0 * (act > 0).double()
    close(out, ref, 2e-5, "gate epilogue")
    with pytest.raises(Exception):      # not combinable with accumulation / split-K
        ops.gemm_raw(dev(dy), dev(w), out, M, N, K, K, 1, 1, N, N, gate=dev(act), accumulate=True)


def test_gemm_argument_errors(ops):
    x = torch.zeros(4, 4, device="cuda")
    import ick_amd.lib as L
    with pytest.raises(L.IckError):
        ops.gemm_raw(x, x, x, 4, 4, 4, 2, 2, 4, 1, 4)  # neither A stride is 1
    with pytest.raises(L.IckError):
        ops.gemm_raw(x, x, x, 0, 4, 4, 4, 1, 4, 1, 4)


# ------------------------------------------------------------------------------------ LayerNorm
@pytest.mark.parametrize("rows,d", [(1280, 300), (7, 300), (33, 512), (5, 1000)])
def test_add_layernorm(ops, rows, d):
    x, r, g, b = rnd(rows, d, seed=1), rnd(rows, d, seed=2), 1 + rnd(d, seed=3, scale=0.1), rnd(d, seed=4)
    ref = torch.nn.functional.layer_norm((x + r).double(), (d,), g.double(), b.double(), 1e-5)
    got, mean, rstd = ops.add_layernorm(dev(x), dev(r), dev(g), dev(b), save_stats=True)
    close(got, ref, 3e-6, "add_layernorm")
    close(mean, (x + r).double().mean(-1), 1e-6, "mean")
    got2 = ops.add_layernorm(dev(x), None, dev(g), dev(b))
    close(got2, torch.nn.functional.layer_norm(x.double(), (d,), g.double(), b.double(), 1e-5), 3e-6, "layernorm")


# ------------------------------------------------------------------------------------ attention
def ref_attention(q, k, v, H, causal, kv_len=None, q_pos0=0):
    B, T, d = q.shape
    S = k.shape[1]
    dh = d // H
    qq = q.double().view(B, T, H, dh).transpose(1, 2)
    kk = k.double().view(B, S, H, dh).transpose(1, 2)
    vv = v.double().view(B, S, H, dh).transpose(1, 2)
    att = qq @ kk.transpose(-1, -2) / math.sqrt(dh)
    s_idx = torch.arange(S).view(1, 1, 1, S)
    if causal:
        t_idx = torch.arange(T).view(1, 1, T, 1) + q_pos0
        att = att.masked_fill(s_idx > t_idx, float("-inf"))
    if kv_len is not None:
        att = att.masked_fill(s_idx >= kv_len.view(B, 1, 1, 1), float("-inf"))
    return (att.softmax(-1) @ vv).transpose(1, 2).reshape(B, T, d)


@pytest.mark.parametrize("B,T,S,causal", [
    (3, 20, 216, False),   # cfg2 cross-attention
    (3, 20, 20, True),     # decoder self-attention
    (2, 7, 6, False),      # entity context encoder
    (2, 102, 598, False),  # largest real shapes (news: L=102? knowledge S=598), T chunks + 3 key blocks
    (2, 32, 302, False),
    (2, 33, 33, True),
    (4, 1, 216, False),    # greedy decode step
])
def test_attention(ops, B, T, S, causal):
    H, d = 10, 300
    q, k, v = rnd(B, T, d, seed=1), rnd(B, S, d, seed=2), rnd(B, S, d, seed=3)
    got = ops.attention(dev(q), dev(k), dev(v), H, causal=causal)
    close(got, ref_attention(q, k, v, H, causal), 3e-6, "attention")


@pytest.mark.parametrize("B,T,S,causal,q_pos0,use_len", [
    (3, 20, 216, False, 0, False),     # decoder cross-attention (cfg2)
    (3, 20, 20, True, 0, False),       # decoder self-attention
    (2, 7, 6, False, 0, True),         # one partial tile each way, ragged keys
    (2, 40, 300, False, 0, True),      # three query tiles, five key tiles per wave (MAXT = 8 instantiation)
    (2, 64, 512, True, 448, False),    # the largest shape of the matrix-core path, causal with an offset
    (2, 33, 257, False, 0, True),      # just past the tile boundaries
    (2, 2, 1, False, 0, False),        # smallest
])
def test_attention_matrix_core_path(ops, B, T, S, causal, q_pos0, use_len):
    """Head-major padded operands with T >= 2 take the MFMA kernels (attention_mfma.hip): outputs and the
    log-sum-exp against float64 math, with NaN in the pad columns and in the rows beyond kv_len."""
    H, d = 10, 300
    dh = d // H
    x, mem = rnd(B, T, d, seed=11), rnd(B, S, d, seed=12)
    q = torch.full((B, 1, H, T, ops.DHP), float("nan"), device="cuda")
    kv = torch.full((B, 2, H, S, ops.DHP), float("nan"), device="cuda")
    q[..., :dh] = dev(x).view(B, T, 1, H, dh).permute(0, 2, 3, 1, 4)
    kv[:, 0, :, :, :dh] = dev(mem).view(B, S, H, dh).permute(0, 2, 1, 3)
    kv[:, 1, :, :, :dh] = dev(mem.flip(1)).view(B, S, H, dh).permute(0, 2, 1, 3)
    kv_len = None
    if use_len:
        kv_len = torch.tensor([S, max(1, S // 3)] + [max(1, S - 1)] * (B - 2), dtype=torch.int32)[:B]
        for b in range(B):
            kv[b, :, :, int(kv_len[b]):, :] = float("nan")
    out = torch.full((B, T, d), float("nan"), device="cuda")
    lse = torch.empty(B * H * T, device="cuda")
    ops.attention_heads(q, kv, out, H, dh, T, S, 0, 0, 1, causal=causal, q_pos0=q_pos0,
                        kv_len=None if kv_len is None else dev(kv_len), lse=lse)
    ref = ref_attention(x, mem, mem.flip(1), H, causal, None if kv_len is None else kv_len.long(), q_pos0)
    close(out, ref, 5e-6, "mfma attention")
    qq = x.double().view(B, T, H, dh).transpose(1, 2)
    kk = mem.double().view(B, S, H, dh).transpose(1, 2)
    att = qq @ kk.transpose(-1, -2) / math.sqrt(dh)
    if kv_len is not None:
        att = att.masked_fill(torch.arange(S).view(1, 1, 1, S) >= kv_len.long().view(B, 1, 1, 1), float("-inf"))
    if causal:
        att = att.masked_fill(torch.arange(S).view(1, 1, 1, S) > q_pos0 + torch.arange(T).view(1, 1, T, 1),
                              float("-inf"))
    close(lse.view(B, H, T), att.logsumexp(-1), 5e-6, "mfma lse")


def test_attention_packed_kv_len_and_lse(ops):
    # q/k/v live in one packed (B, T, 3d) projection; per-sample key length; causal offset
    B, T, S, H, d = 3, 1, 12, 10, 300
    dh = d // H
    qkv = rnd(B, 1, 3 * d, seed=4)
    cache = rnd(B, S, 2 * d, seed=5)
    kv_len = torch.tensor([12, 5, 9], dtype=torch.int32)
    out = torch.empty(B, T, d, device="cuda")
    lse = torch.empty(B * H * T, device="cuda")
    dq, dc = dev(qkv), dev(cache)
    ops.attention_raw(dq, dc, dc, out, B, H, T, S, dh, 3 * d, dh, 3 * d, S * 2 * d, dh, 2 * d, S * 2 * d, dh, 2 * d,
                      d, d, causal=False, kv_len=dev(kv_len), lse=lse, v_off=d)
    ref = ref_attention(qkv[:, :, :d], cache[:, :, :d], cache[:, :, d:], H, False, kv_len.long())
    close(out, ref, 3e-6, "attention kv_len")
    qq = qkv[:, :, :d].double().view(B, T, H, dh).transpose(1, 2)
    kk = cache[:, :, :d].double().view(B, S, H, dh).transpose(1, 2)
    att = qq @ kk.transpose(-1, -2) / math.sqrt(dh)
    att = att.masked_fill(torch.arange(S).view(1, 1, 1, S) >= kv_len.long().view(B, 1, 1, 1), float("-inf"))
    close(lse.view(B, H, T), att.logsumexp(-1), 3e-6, "lse")


def test_head_split_projection_and_head_major_attention(ops):
    """ick_gemm's head-split epilogue + the vectorised attention path == row-major math, including the
    KV-cache form (rows appended at a position, queries read at an offset)."""
    B, T, S, H, d = 3, 20, 216, 10, 300
    dh = d // H
    x, mem = rnd(B, T, d, seed=1), rnd(B, S, d, seed=2)
    wq, bq = rnd(d, d, seed=3, scale=0.1), rnd(d, seed=4)
    wkv, bkv = rnd(4 * d, d, seed=5, scale=0.1), rnd(4 * d, seed=6)   # two layers' [K;V]
    q = ops.project_heads(dev(x), dev(wq), dev(bq), 1, H, T)
    kv = torch.full((B, 4, H, S, ops.DHP), float("nan"), device="cuda")     # pads stay NaN: must be masked
    ops.project_heads(dev(mem[:, :200].contiguous()), dev(wkv), dev(bkv), 4, H, S, out=kv, s0=0, grp=200)
    ops.project_heads(dev(mem[:, 200:].contiguous()), dev(wkv), dev(bkv), 4, H, S, out=kv, s0=200, grp=S - 200)
    qr = x.double() @ wq.double().t() + bq.double()
    kvr = mem.double() @ wkv.double().t() + bkv.double()
    got = kv[:, :, :, :, :dh].permute(0, 3, 1, 2, 4).reshape(B, S, 4 * d)
    close(got, kvr, 2e-5, "head-split kv")
    for layer in (0, 1):
        out = torch.empty(B, T, d, device="cuda")
        ops.attention_heads(q, kv, out, H, dh, T, S, q_seg=0, k_seg=2 * layer, v_seg=2 * layer + 1)
        ref = ref_attention(qr.float(), kvr[:, :, 2 * layer * d:(2 * layer + 1) * d].float(),
                            kvr[:, :, (2 * layer + 1) * d:(2 * layer + 2) * d].float(), H, False)
        close(out, ref, 5e-6, "head-major attention layer %d" % layer)
    # KV-cache form: position 7 of a (B, 3, H, 12, 32) cache, attends to keys 0..7
    ML, pos = 12, 7
    w3, b3 = rnd(3 * d, d, seed=7, scale=0.1), rnd(3 * d, seed=8)
    xs = rnd(B, ML, d, seed=9)
    cache = torch.full((B, 3, H, ML, ops.DHP), float("nan"), device="cuda")
    for i in range(pos + 1):
        ops.project_heads(dev(xs[:, i:i + 1].contiguous()), dev(w3), dev(b3), 3, H, ML, out=cache, s0=i, grp=1)
    out = torch.empty(B, 1, d, device="cuda")
    ops.attention_heads(cache, cache, out, H, dh, 1, pos + 1, q_seg=0, k_seg=1, v_seg=2, q_t0=pos)
    pr = xs.double() @ w3.double().t() + b3.double()
    ref = ref_attention(pr[:, pos:pos + 1, :d].float(), pr[:, :pos + 1, d:2 * d].float(),
                        pr[:, :pos + 1, 2 * d:].float(), H, False)
    close(out, ref, 5e-6, "cached step")


# -------------------------------------------------------------------------------------- prefill
@pytest.mark.parametrize("variant", synth.VARIANTS)
def test_prefill_gathers_exact(ops, variant):
    B, Lc, K, V, Fn, seed = 5, 9, 7, 60, (0 if variant == "geo" else 6), 21
    P = synth.make_params(variant, V, seed)
    wm = synth.make_word_map(V)
    cfg = R.config_from_word_map(variant, wm)
    batch = synth.make_batch(variant, B, Lc, K, V, Fn, seed)
    # a few adversarial tokens: pointer out of range, mask says entity but token is a word
    batch["captions"][0, 1] = V + K + Fn + 5
    batch["caption_masks"][0, 1] = 1
    batch["captions"][1, 2] = 3
    batch["caption_masks"][1, 2] = 1
    if variant != "geo":
        batch["captions"][2, 1] = 5
        batch["caption_masks"][2, 1] = 2
    facts = batch.get("facts")
    ee_ref = R.entity_encode(cfg, P, batch["entities"], facts)
    ee = ops.entity_encode(variant, dev(batch["entities"]), dev(P["entity_encoder.type_embedding.weight"]), 300,
                           facts=None if facts is None else dev(facts),
                           word_emb=dev(P["word_embedding.weight"]) if variant == "news" else None)
    if variant == "news":
        close(ee, ee_ref, 1e-7, "entity_encode news")
    else:
        assert torch.equal(ee.cpu(), ee_ref), "entity_encode must be bit-exact"
    fe = fe_ref = None
    if variant != "geo":
        fe_ref = R.fact_encode(P, facts, ee_ref)
        fe = ops.fact_encode(dev(facts), dev(ee_ref), dev(P["predicate_embedding.weight"]))
        assert torch.equal(fe.cpu(), fe_ref)
    emb_ref = R.caption_embed(cfg, P, batch["captions"], batch["caption_masks"], ee_ref, fe_ref)
    pe = R.pe_table(64, 300)
    x, emb = ops.caption_embed(dev(batch["captions"]), dev(batch["caption_masks"]), dev(P["word_embedding.weight"]),
                               dev(ee_ref), None if fe_ref is None else dev(fe_ref), dev(pe), V, cfg.pad,
                               math.sqrt(300), want_emb=True)
    assert torch.equal(emb.cpu(), emb_ref)
    x_ref = emb_ref * math.sqrt(300) + pe[:Lc].unsqueeze(0)
    assert torch.equal(x.cpu(), x_ref)
    # decode-step form: one column at absolute position 4
    x1 = ops.caption_embed(dev(batch["captions"][:, 4:5].contiguous()), dev(batch["caption_masks"][:, 4:5].contiguous()),
                           dev(P["word_embedding.weight"]), dev(ee_ref), None if fe_ref is None else dev(fe_ref),
                           dev(pe), V, cfg.pad, math.sqrt(300), pos0=4)
    assert torch.equal(x1.cpu(), x_ref[:, 4:5])


@pytest.mark.parametrize("variant", ["knowledge", "news"])
def test_context_indicators(ops, variant):
    B, Lc, K, V, Fn, seed = 6, 14, 5, 40, 9, 33
    P = synth.make_params(variant, V, seed)
    cfg = R.config_from_word_map(variant, synth.make_word_map(V))
    batch = synth.make_batch(variant, B, Lc, K, V, Fn, seed)
    caps = batch["captions"]
    caps[:, 2] = V + 1          # make sure entities get mentioned
    caps[0, 5] = V + 3
    facts = batch["facts"]
    facts[:, :4, 1] = torch.tensor([1, 1, 3, 1])
    facts[:, 1, 2] = facts[:, 0, 2]  # duplicate predicate among active facts
    eib_ref, pi_ref = R.context_indicators(cfg, caps, facts, K, Lc)
    gate_ref = torch.nn.functional.linear(pi_ref.double(), P["fc_predicate.weight"].double(),
                                          P["fc_predicate.bias"].double())
    wt = P["fc_predicate.weight"].t().contiguous()
    eib, gate = ops.context_indicators(dev(caps), dev(facts), K, V, dev(wt), dev(P["fc_predicate.bias"]), mode=0)
    assert torch.equal(eib.cpu(), eib_ref)
    close(gate, gate_ref, 1e-6, "gate")
    eib1_ref, pi1_ref = R.context_indicators(cfg, caps, facts, K, 1)
    eib1, gate1 = ops.context_indicators(dev(caps), dev(facts), K, V, dev(wt), dev(P["fc_predicate.bias"]), mode=1)
    assert torch.equal(eib1.cpu(), eib1_ref)
    close(gate1, torch.nn.functional.linear(pi1_ref.double(), P["fc_predicate.weight"].double(),
                                            P["fc_predicate.bias"].double()), 1e-6, "gate predict-mode")


# ----------------------------------------------------------------------------------- score head
def test_pointer_scores(ops):
    B, T, Kc, d, V = 4, 6, 20, 300, 50
    h, ctx, w, b = rnd(B, T, d, seed=1), rnd(B, Kc, d, seed=2), rnd(1, d, seed=3, scale=0.1), rnd(1, seed=4)
    ind = (rnd(B, T, Kc, seed=5) > 0).float()
    out = torch.zeros(B, T, V + Kc, device="cuda")
    gmap = torch.tensor([1, 3, 0, 2], dtype=torch.int32)
    ops.pointer_scores(dev(h), dev(ctx), dev(w), dev(b), out, V, ind=dev(ind), out_gmap=dev(gmap))
    ref = torch.zeros(B, T, V + Kc, dtype=torch.double)
    ref[gmap.long(), :, V:] = ((h.double().unsqueeze(2) * ctx.double().unsqueeze(1) * ind.double().unsqueeze(3))
                               * w.double().view(1, 1, 1, d)).sum(-1) + b.double()
    close(out, ref, 2e-6, "pointer_scores")


def test_top2_and_ties(ops):
    sc = rnd(7, 10020, seed=9)
    sc[3, 17] = sc[3, 4000] = 5.0      # tie for best -> lower index wins, runner-up is the other one
    best, second = ops.top2(dev(sc))
    tk = sc.topk(2, dim=1).indices
    for b in range(7):
        if b == 3:
            assert (best[b].item(), second[b].item()) == (17, 4000)
        else:
            assert (best[b].item(), second[b].item()) == (tk[b, 0].item(), tk[b, 1].item())


def test_greedy_select_equals_top2_plus_update(ops):
    """The fused per-token kernel == ick_top2 followed by ick_greedy_update, over a scripted decode (repeats that
    trigger the n-gram clean-up, an <end>, ties broken towards the lower index)."""
    V, K, max_len, end = 50, 6, 12, 49
    B, Vx = 5, V + K
    g = torch.Generator().manual_seed(3)
    state = [[torch.zeros(B, max_len, dtype=torch.long, device="cuda"), torch.zeros(B, max_len, dtype=torch.int32, device="cuda"),
              torch.zeros(B, dtype=torch.int32, device="cuda"), torch.zeros(B, dtype=torch.long, device="cuda"),
              torch.zeros(B, dtype=torch.long, device="cuda")] for _ in range(2)]
    for i in range(max_len):
        scores = torch.randn(B, Vx, generator=g)
        scores[0, 7] = 9.0                       # caption 0 repeats token 7 (1-gram clean-up)
        scores[1, 3 + (i % 2)] = 9.0             # caption 1 alternates 3, 4 (2-gram clean-up)
        scores[2, end if i == 4 else 11] = 9.0   # caption 2 ends at step 4
        scores[3, 20] = scores[3, 21] = 8.0      # caption 3: exact tie -> lower index wins, other is second
        d = dev(scores)
        best, second = ops.top2(d)
        ops.greedy_update(best, second, *state[0], i, V, K, False, end)
        ops.greedy_select(d, *state[1], i, V, K, False, end)
        for a, b in zip(state[0], state[1]):
            assert torch.equal(a, b), "step %d" % i


def test_greedy_update_matches_reference_cleanup(ops):
    # drive the device bookkeeping with scripted (best, second) streams and compare with the oracle's loop
    V, K, max_len, end = 50, 6, 14, 49
    streams = [
        [5, 5, 5, 7, 8, 7, 8, 9, 1, 2, 3, 1, 2, 3],
        [1, 2, 1, 2, 1, 2, 3, 3, 49, 4, 4, 4, 4, 4],
        [52, 52, 53, 54, 53, 54, 10, 11, 12, 10, 11, 12, 49, 1],
        [9, 8, 7, 6, 5, 4, 3, 2, 1, 9, 8, 7, 6, 5],
    ]
    B = len(streams)
    sec = [[(t * 7 + b * 3 + 20) % 48 + 1 for t in range(max_len)] for b in range(B)]
    output = torch.zeros(B, max_len, dtype=torch.long, device="cuda")
    hist = torch.zeros(B, max_len, dtype=torch.int32, device="cuda")
    fin = torch.zeros(B, dtype=torch.int32, device="cuda")
    nt = torch.zeros(B, dtype=torch.long, device="cuda")
    nm = torch.zeros(B, dtype=torch.long, device="cuda")
    ref_out = [[0] * max_len for _ in range(B)]
    ref_hist = [[] for _ in range(B)]
    ref_fin = [False] * B
    for i in range(max_len):
        best = torch.tensor([s[i] for s in streams], dtype=torch.int32)
        second = torch.tensor([s[i] for s in sec], dtype=torch.int32)
        ops.greedy_update(dev(best), dev(second), output, hist, fin, nt, nm, i, V, K, False, end)
        for b in range(B):
            if ref_fin[b]:
                continue
            ref_out[b][i] = streams[b][i]
            if streams[b][i] == end:
                ref_fin[b] = True
                continue
            ref_hist[b].append(sec[b][i])
            R.loop_cleanup(ref_out[b], ref_hist[b], i)
            if i < max_len - 1:
                assert nt[b].item() == ref_out[b][i]
                assert nm[b].item() == (1 if ref_out[b][i] >= V else 0)
    assert output.cpu().tolist() == ref_out
    assert fin.cpu().tolist() == [int(f) for f in ref_fin]


@pytest.mark.parametrize("Vx", [1020, 1021, 12001])   # register-resident rows / odd width / longer than the registers hold
def test_packed_ce(ops, Vx):
    B, Lc, pad = 6, 9, 0
    sc = rnd(B, Lc, Vx, seed=3, scale=3.0)
    caps = torch.randint(1, Vx, (B, Lc), generator=torch.Generator().manual_seed(1))
    lens = torch.tensor([9, 8, 8, 6, 5, 3])
    for b in range(B):
        caps[b, lens[b]:] = 0
    dl = (lens - 1).to(torch.int32)
    cfg = R.Config("geo", Vx)
    sc_ref = sc.clone().double().requires_grad_(True)
    loss_ref = R.packed_ce_loss(cfg, sc_ref, caps, dl.tolist())
    loss_ref.backward()
    ls, cnt, dsc = ops.packed_ce(dev(sc), dev(caps), dev(dl), pad, want_grad=True)
    assert cnt.item() == float(dl.sum())
    assert abs(ls.item() / cnt.item() - loss_ref.item()) < 1e-5
    close(dsc / cnt, sc_ref.grad, 1e-6, "dscores")


def test_copy_batch(ops):
    """ick_copy_batch: several device-to-device copies of different sizes / dtypes / alignments in one launch."""
    src = [dev(rnd(64, 20, seed=1)), torch.arange(1280, device="cuda"), dev(rnd(3, seed=2)), dev(rnd(64, 196, 300, seed=3)),
           torch.arange(7, device="cuda", dtype=torch.int32), dev(rnd(1025, seed=4))[1:]]
    dst = [torch.empty_like(s) for s in src]
    dst[5] = torch.zeros(1030, device="cuda")[3:1027]      # unaligned destination
    ops.copy_batch(dst, [s.contiguous() for s in src])
    for d, s in zip(dst, src):
        assert torch.equal(d, s)
