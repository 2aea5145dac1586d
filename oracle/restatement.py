"""
TEST INFRASTRUCTURE ONLY -- CPU restatement ("oracle") of the caption-decoder hot path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the product path (image-captioning-with-external-knowledge_amd/) never does.

Explicit-math restatement (plain torch on CPU, fp32) of the reference's models.py for the
three variants ("geo", "knowledge", "news").  Every function cites the reference file:line
(relative to /root/reference) it follows.  The arithmetic of the reference lives in
torch.nn (pinned torch==1.9.0 in */requirements.txt, not vendored); the published algorithm
of nn.TransformerDecoderLayer / nn.TransformerEncoderLayer (post-LN, ReLU, eps 1e-5, packed
in_proj, q scaled by 1/sqrt(dh)) is restated here.

Parity pin: tests/golden/*.npz were produced by tests/golden/make_fixtures.py, which imports
the real reference models.py in the authoring container; tests/test_oracle_golden.py checks
this file against them (forward stages, scores, loss/grads, greedy predict token sequences).

Parameters are passed as a flat dict with the reference's state_dict names
(e.g. "transformer_decoder.layers.0.self_attn.in_proj_weight").
"""
import math
from dataclasses import dataclass

import torch
import torch.nn.functional as F

LN_EPS = 1e-5


@dataclass
class Config:
    variant: str  # "geo" | "knowledge" | "news"
    vocab_size: int
    emb_dim: int = 300
    num_heads: int = 10
    num_layers: int = 3
    pad: int = 0
    start: int = 0
    end: int = 0
    num_predicates: int = 0

    @property
    def has_facts(self):
        return self.variant in ("knowledge", "news")

    @property
    def type_offset(self):
        # first slot of the type embedding inside an encoded entity
        # geo-aware/models.py:102 (4), knowledge-aware/models.py:131 (6), news-knowledge-aware/models.py:126 (5)
        return {"geo": 4, "knowledge": 6, "news": 5}[self.variant]


def config_from_word_map(variant, word_map, emb_dim=300, num_heads=10, num_layers=3):
    return Config(
        variant=variant,
        vocab_size=len(word_map),
        emb_dim=emb_dim,
        num_heads=num_heads,
        num_layers=num_layers,
        pad=word_map["<pad>"],
        start=word_map["<start>"],
        end=word_map["<end>"],
        num_predicates={"geo": 0, "knowledge": 3000, "news": 3500}[variant],
    )


# ----------------------------------------------------------------------------------------
# a1  Encoder.conv1 + view                       geo-aware/models.py:32,45-46
# ----------------------------------------------------------------------------------------
def feat_proj(feats, conv_w, conv_b):
    """feats (B,C,14,14) -> (B, emb_dim, 196): 1x1 conv == per-pixel GEMM + bias."""
    B, C = feats.shape[0], feats.shape[1]
    w = conv_w.reshape(conv_w.shape[0], C)  # (d, C)
    x = feats.reshape(B, C, -1)  # (B, C, P)
    return torch.einsum("dc,bcp->bdp", w, x) + conv_b.view(1, -1, 1)


# ----------------------------------------------------------------------------------------
# a5  PositionEncoder                            geo-aware/models.py:199-209
# ----------------------------------------------------------------------------------------
def pe_table(max_len, d):
    pe = torch.zeros(max_len, d)
    position = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d, 2).float() * (-math.log(10000.0) / d))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe  # (max_len, d); reference buffer is this with a singleton middle dim


# ----------------------------------------------------------------------------------------
# a2  EntityEncoder.forward
#     geo-aware/models.py:82-122, knowledge-aware/models.py:82-151, news-knowledge-aware/models.py:79-134
# ----------------------------------------------------------------------------------------
def azimuth_slots(az):
    """get_dist_to_north / get_dist_to_east (geo-aware/models.py:106-122).  The reference
    evaluates them with Python floats (double) through Tensor.apply_ and stores float32."""
    a = az.double()
    north = a.abs() / 180.0
    east = torch.where(a >= -90.0, (90.0 - a).abs(), 90.0 + (a + 180.0).abs()) / 180.0
    return north.float(), east.float()


def fact_counts(facts, K):
    """Per entity, number of facts whose subject it is; the last entity (<unk_ent>) is
    forced to 0 (knowledge-aware/models.py:101-121)."""
    subj = facts[:, :, 1].long()  # (B,F)
    onehot = subj.unsqueeze(2) == torch.arange(K, device=facts.device).view(1, 1, K)
    counts = onehot.sum(dim=1).float()  # (B,K)
    counts[:, K - 1] = 0.0
    return counts


def entity_encode(cfg, P, entities, facts=None):
    B, K, _ = entities.shape
    d = cfg.emb_dim
    e = torch.zeros(B, K, d)
    ent = entities.float()
    if cfg.variant in ("geo", "knowledge"):
        e[:, :, 0] = ent[:, :, 1]
        north, east = azimuth_slots(ent[:, :, 2])
        e[:, :, 1] = north
        e[:, :, 2] = east
        e[:, :, 3] = ent[:, :, 3]
    else:  # news: count / in-headline / in-first-paragraph  (news-knowledge-aware/models.py:90-95)
        e[:, :, 0] = ent[:, :, 1]
        e[:, :, 1] = ent[:, :, 2]
        e[:, :, 2] = ent[:, :, 3]
    if cfg.has_facts:
        c = fact_counts(facts, K)
        s = 4 if cfg.variant == "knowledge" else 3
        e[:, :, s] = c
        e[:, :, s + 1] = (c > 0).float()
    to = cfg.type_offset
    e[:, :, to:] = P["entity_encoder.type_embedding.weight"][ent[:, :, 4].long()]
    if cfg.variant == "news":
        # mean of the (up to) five name-word embeddings, pad rows included
        # (news-knowledge-aware/models.py:128-133)
        name_idx = ent[:, :, 5:].long()
        avg = P["word_embedding.weight"][name_idx].mean(dim=-2)
        e = e * avg
    return e


# ----------------------------------------------------------------------------------------
# a3  FactEncoder.forward                        knowledge-aware/models.py:170-188
# ----------------------------------------------------------------------------------------
def fact_encode(P, facts, entities_encoded):
    subj = facts[:, :, 1].long()
    d = entities_encoded.shape[2]
    gathered = torch.gather(entities_encoded, 1, subj.unsqueeze(2).expand(-1, -1, d))
    return gathered + P["predicate_embedding.weight"][facts[:, :, 2].long()]


# ----------------------------------------------------------------------------------------
# a4  CaptionEmbedder.forward   geo-aware/models.py:143-181, knowledge-aware/models.py:209-259
# ----------------------------------------------------------------------------------------
def caption_embed(cfg, P, captions, masks, entities_encoded, facts_encoded=None):
    """captions (B,L) int64, masks (B,L) in {0,1,2} -> (B,L,d)."""
    V = cfg.vocab_size
    B, L = captions.shape
    K = entities_encoded.shape[1]
    d = entities_encoded.shape[2]
    words = torch.where(captions >= V, torch.full_like(captions, cfg.pad), captions)
    out = P["word_embedding.weight"][words]
    ei = captions - V
    ei = torch.where((ei < 0) | (ei >= K), torch.full_like(ei, K - 1), ei)
    ent_rows = torch.gather(entities_encoded, 1, ei.unsqueeze(2).expand(-1, -1, d))
    out = torch.where((masks == 1).unsqueeze(2), ent_rows, out)
    if facts_encoded is not None:
        Fn = facts_encoded.shape[1]
        fi = captions - V - K
        fi = torch.where((fi < 0) | (fi >= Fn), torch.full_like(fi, Fn - 1), fi)
        fact_rows = torch.gather(facts_encoded, 1, fi.unsqueeze(2).expand(-1, -1, d))
        out = torch.where((masks == 2).unsqueeze(2), fact_rows, out)
    return out


# ----------------------------------------------------------------------------------------
# torch.nn.MultiheadAttention / Transformer{En,De}coderLayer (post-LN) restated.
# Call sites: geo-aware/models.py:241-244,348,358; knowledge-aware/models.py:319-324,495-496,508
# Batch-major here: x is (B, T, d) (the reference is (T, B, d); samples are independent).
# ----------------------------------------------------------------------------------------
def mha(xq, xkv, in_w, in_b, out_w, out_b, H, causal):
    B, T, d = xq.shape
    S = xkv.shape[1]
    dh = d // H
    q = F.linear(xq, in_w[:d], in_b[:d])
    k = F.linear(xkv, in_w[d : 2 * d], in_b[d : 2 * d])
    v = F.linear(xkv, in_w[2 * d :], in_b[2 * d :])
    q = q.view(B, T, H, dh).transpose(1, 2) * (1.0 / math.sqrt(dh))
    k = k.view(B, S, H, dh).transpose(1, 2)
    v = v.view(B, S, H, dh).transpose(1, 2)
    att = q @ k.transpose(-1, -2)  # (B,H,T,S)
    if causal:
        neg = torch.full((T, S), float("-inf")).triu(1)  # geo-aware/models.py:256-262
        att = att + neg
    att = att.softmax(dim=-1)
    ctx = (att @ v).transpose(1, 2).reshape(B, T, d)
    return F.linear(ctx, out_w, out_b)


def _ln(x, P, name):
    return F.layer_norm(x, (x.shape[-1],), P[name + ".weight"], P[name + ".bias"], LN_EPS)


def encoder_layer(P, pre, x, H):
    a = mha(x, x, P[pre + "self_attn.in_proj_weight"], P[pre + "self_attn.in_proj_bias"],
            P[pre + "self_attn.out_proj.weight"], P[pre + "self_attn.out_proj.bias"], H, False)
    x = _ln(x + a, P, pre + "norm1")
    f = F.linear(F.relu(F.linear(x, P[pre + "linear1.weight"], P[pre + "linear1.bias"])),
                 P[pre + "linear2.weight"], P[pre + "linear2.bias"])
    return _ln(x + f, P, pre + "norm2")


def decoder_layer(P, pre, x, mem, H):
    a = mha(x, x, P[pre + "self_attn.in_proj_weight"], P[pre + "self_attn.in_proj_bias"],
            P[pre + "self_attn.out_proj.weight"], P[pre + "self_attn.out_proj.bias"], H, True)
    x = _ln(x + a, P, pre + "norm1")
    c = mha(x, mem, P[pre + "multihead_attn.in_proj_weight"], P[pre + "multihead_attn.in_proj_bias"],
            P[pre + "multihead_attn.out_proj.weight"], P[pre + "multihead_attn.out_proj.bias"], H, False)
    x = _ln(x + c, P, pre + "norm2")
    f = F.linear(F.relu(F.linear(x, P[pre + "linear1.weight"], P[pre + "linear1.bias"])),
                 P[pre + "linear2.weight"], P[pre + "linear2.bias"])
    return _ln(x + f, P, pre + "norm3")


def context_encoder(cfg, P, stack, x):
    """stack = 'transformer_encoder_entities' | 'transformer_encoder_facts' (a8)."""
    for i in range(cfg.num_layers):
        x = encoder_layer(P, f"{stack}.layers.{i}.", x, cfg.num_heads)
    return x


def build_memory(cfg, P, enc_out, entities_encoded, facts_encoded=None):
    """mem (B, S, d) = [196 image rows ; encoded entity rows ; encoded fact rows].
    geo-aware/models.py:347-349, knowledge-aware/models.py:494-499.  enc_out is (B, d, P)."""
    parts = [enc_out.permute(0, 2, 1), context_encoder(cfg, P, "transformer_encoder_entities", entities_encoded)]
    if facts_encoded is not None:
        parts.append(context_encoder(cfg, P, "transformer_encoder_facts", facts_encoded))
    return torch.cat(parts, dim=1)


def decoder_stack(cfg, P, x, mem):
    for i in range(cfg.num_layers):
        x = decoder_layer(P, f"transformer_decoder.layers.{i}.", x, mem, cfg.num_heads)
    return x


# ----------------------------------------------------------------------------------------
# a11 get_context_indicators                     knowledge-aware/models.py:380-418
# ----------------------------------------------------------------------------------------
def mention_matrix(cfg, captions, K):
    """(B, L, K) bool: caption position t holds a pointer to entity n."""
    ei = captions - cfg.vocab_size
    return ei.unsqueeze(2) == torch.arange(K).view(1, 1, K)


def context_indicators(cfg, captions, facts, K, out_length):
    """Returns entity_idx_before (B, out_length, F) and predicate_indicator
    (B, out_length, num_predicates), both 0/1 float.  out_length == L: position p sees
    entity mentions at positions < p (:406-407).  out_length == 1 (predict): one row that
    sees every mention anywhere in the caption buffer (:408-409)."""
    B, L = captions.shape
    Fn = facts.shape[1]
    m = mention_matrix(cfg, captions, K)  # (B,L,K)
    if out_length != 1:
        csum = m.long().cumsum(dim=1)
        before = torch.zeros_like(csum)
        before[:, 1:] = csum[:, :-1]
        seen = before > 0  # (B,L,K) mention strictly before p
        seen = seen[:, :out_length]
    else:
        seen = m.any(dim=1, keepdim=True)  # (B,1,K)
    subj = facts[:, :, 1].long()
    valid = (subj >= 0) & (subj < K)
    g = torch.gather(seen, 2, subj.clamp(0, K - 1).unsqueeze(1).expand(-1, seen.shape[1], -1))
    eib = (g & valid.unsqueeze(1)).to(torch.get_default_dtype())  # (B,T,F); float32 unless a test runs the oracle in float64
    pred = facts[:, :, 2].long()
    pi = torch.zeros(B, seen.shape[1], cfg.num_predicates)
    pi.scatter_reduce_(2, pred.unsqueeze(1).expand(-1, seen.shape[1], -1), eib, reduce="amax", include_self=True)
    return eib, pi


# ----------------------------------------------------------------------------------------
# a10 get_scores              geo-aware/models.py:291-313, knowledge-aware/models.py:420-455
# ----------------------------------------------------------------------------------------
def get_scores(cfg, P, h, entities_encoded, facts_encoded=None, eib=None, pi=None):
    """h (B,T,d) -> (B,T,V+K[+F])."""
    if cfg.has_facts:
        gate = F.linear(pi, P["fc_predicate.weight"], P["fc_predicate.bias"])  # (B,T,d)
        vocab = F.linear(h * gate, P["fc_vocab.weight"], P["fc_vocab.bias"])
    else:
        vocab = F.linear(h, P["fc_vocab.weight"], P["fc_vocab.bias"])
    we = P["fc_entity.weight"].view(1, 1, 1, -1)
    ent = ((h.unsqueeze(2) * entities_encoded.unsqueeze(1)) * we).sum(-1) + P["fc_entity.bias"]
    parts = [vocab, ent]
    if cfg.has_facts:
        wf = P["fc_fact.weight"].view(1, 1, 1, -1)
        fin = h.unsqueeze(2) * facts_encoded.unsqueeze(1) * eib.unsqueeze(3)
        parts.append((fin * wf).sum(-1) + P["fc_fact.bias"])
    return torch.cat(parts, dim=2)


# ----------------------------------------------------------------------------------------
# a12 DecoderTransformer.forward  geo-aware/models.py:315-361, knowledge-aware/models.py:457-514
# ----------------------------------------------------------------------------------------
def forward(cfg, P, captions, enc_out, caption_masks, caption_lengths, entities, facts=None, stages=None):
    """Returns (scores (B,L,Vx) in length-sorted order, captions_sorted, decode_lengths)."""
    lengths, sort_ind = caption_lengths.squeeze(1).sort(dim=0, descending=True)
    enc_out = enc_out[sort_ind]
    captions = captions[sort_ind]
    caption_masks = caption_masks[sort_ind]
    entities = entities[sort_ind]
    if facts is not None:
        facts = facts[sort_ind]
    decode_lengths = (lengths - 1).tolist()
    ee = entity_encode(cfg, P, entities, facts)
    fe = fact_encode(P, facts, ee) if cfg.has_facts else None
    emb = caption_embed(cfg, P, captions, caption_masks, ee, fe)
    mem = build_memory(cfg, P, enc_out, ee, fe)
    L = captions.shape[1]
    x = emb * math.sqrt(cfg.emb_dim) + pe_table(L, cfg.emb_dim).unsqueeze(0)
    h = decoder_stack(cfg, P, x, mem)
    if cfg.has_facts:
        eib, pi = context_indicators(cfg, captions, facts, entities.shape[1], L)
        scores = get_scores(cfg, P, h, ee, fe, eib, pi)
    else:
        scores = get_scores(cfg, P, h, ee)
    if stages is not None:
        stages.update(entities_encoded=ee, facts_encoded=fe, embeddings=emb, memory=mem, x0=x, h=h,
                      sort_ind=sort_ind)
    return scores, captions, decode_lengths


# ----------------------------------------------------------------------------------------
# a14 loss                                         geo-aware/train.py:275-281,136
# ----------------------------------------------------------------------------------------
def packed_ce_loss(cfg, scores, captions_sorted, decode_lengths):
    """CrossEntropyLoss(ignore_index=<pad>) over pack_padded_sequence(scores/targets): the
    packed rows are exactly the (b, t) with t < decode_lengths[b]; row order does not matter
    to the mean."""
    B, L, Vx = scores.shape
    targets = captions_sorted[:, 1:]
    dl = torch.tensor(decode_lengths).view(B, 1)
    keep = torch.arange(L - 1).view(1, -1) < dl
    rows = scores[:, : L - 1][keep]
    tg = targets[keep]
    return F.cross_entropy(rows, tg, ignore_index=cfg.pad)


# ----------------------------------------------------------------------------------------
# a13 DecoderTransformer.predict  geo-aware/models.py:363-443, knowledge-aware/models.py:516-609
# Greedy, batch 1, full recompute every step (no KV cache), n-gram loop clean-up.
# ----------------------------------------------------------------------------------------
def loop_cleanup(output, prev_top_two, i):
    """geo-aware/models.py:421-435 on Python int lists.  output[0..i] are this caption's
    tokens so far (output[i] just written), prev_top_two[t] the runner-up token of step t."""
    for dupl in (0, 2, 4):
        if i > dupl:
            n = dupl + 2
            s = [output[i - j] for j in range(n)]
            if s[: n // 2] == s[n // 2 :]:
                top = 1 if dupl == 0 else dupl
                for r in range(top):
                    output[i - r] = prev_top_two[-(r + 1)]
                break


def predict(cfg, P, enc_out, max_pred_len, entities, facts=None, return_scores=False):
    """enc_out (1,d,196) -> LongTensor (max_pred_len, 1), <pad> after the caption."""
    assert enc_out.shape[0] == 1
    V, K = cfg.vocab_size, entities.shape[1]
    ee = entity_encode(cfg, P, entities, facts)
    fe = fact_encode(P, facts, ee) if cfg.has_facts else None
    mem = build_memory(cfg, P, enc_out, ee, fe)
    captions = [cfg.start] * max_pred_len
    masks = [0] * max_pred_len
    output = [cfg.pad] * max_pred_len
    prev_top_two = []
    pe = pe_table(max_pred_len, cfg.emb_dim).unsqueeze(0)
    all_scores = []
    for i in range(max_pred_len):
        cap_t = torch.tensor([captions])
        emb = caption_embed(cfg, P, cap_t, torch.tensor([masks]), ee, fe)
        x = emb * math.sqrt(cfg.emb_dim) + pe
        h = decoder_stack(cfg, P, x, mem)[:, i : i + 1]
        if cfg.has_facts:
            eib, pi = context_indicators(cfg, cap_t, facts, K, 1)
            sc = get_scores(cfg, P, h, ee, fe, eib, pi)
        else:
            sc = get_scores(cfg, P, h, ee)
        sc = sc[0, 0]
        all_scores.append(sc)
        prob = sc.softmax(dim=-1)
        out = int(prob.argmax())
        output[i] = out
        if out == cfg.end:
            break
        prev_top_two.append(int(prob.topk(2).indices[1]))
        loop_cleanup(output, prev_top_two, i)
        out = output[i]
        if i < max_pred_len - 1:
            captions[i + 1] = out
            if cfg.has_facts and out >= V + K:
                masks[i + 1] = 2
            elif out >= V:
                masks[i + 1] = 1
    res = torch.tensor(output, dtype=torch.long).view(max_pred_len, 1)
    if return_scores:
        return res, torch.stack(all_scores)
    return res


# ----------------------------------------------------------------------------------------
# Beam search (north_star cfg5).  NOT in the reference (geo-aware/eval.py:61,83 decodes greedily): this CPU
# restatement only pins the build's own device beam search ("parity-unpinned" against the reference).
# Rules: hypotheses are scored by their summed log_softmax; an ended hypothesis competes with its final score;
# candidates are ranked by (score desc, hypothesis index asc, token asc); full recompute per step, no clean-up.
# ----------------------------------------------------------------------------------------
def predict_beam(cfg, P, enc_out, max_pred_len, entities, facts=None, beam_size=5):
    """enc_out (1,d,196) -> (best sequence LongTensor (max_pred_len,), its log-probability, all (seq, score))."""
    assert enc_out.shape[0] == 1
    V, K = cfg.vocab_size, entities.shape[1]
    ee = entity_encode(cfg, P, entities, facts)
    fe = fact_encode(P, facts, ee) if cfg.has_facts else None
    mem = build_memory(cfg, P, enc_out, ee, fe)
    pe = pe_table(max_pred_len, cfg.emb_dim).unsqueeze(0)
    Vx = V + K + (facts.shape[1] if facts is not None else 0)
    hyps = [dict(seq=[], score=0.0, fin=False)] + [None] * (beam_size - 1)
    for i in range(max_pred_len):
        cands = []
        for j, h in enumerate(hyps):
            if h is None:
                continue
            if h["fin"]:
                cands.append((h["score"], j, 0, None))
                continue
            captions = [cfg.start] + h["seq"] + [cfg.start] * (max_pred_len - 1 - len(h["seq"]))
            masks = [0] + [2 if (cfg.has_facts and t >= V + K) else (1 if t >= V else 0) for t in h["seq"]]
            masks = masks + [0] * (max_pred_len - len(masks))
            cap_t = torch.tensor([captions[:max_pred_len]])
            emb = caption_embed(cfg, P, cap_t, torch.tensor([masks[:max_pred_len]]), ee, fe)
            x = emb * math.sqrt(cfg.emb_dim) + pe
            hh = decoder_stack(cfg, P, x, mem)[:, i : i + 1]
            if cfg.has_facts:
                eib, pi = context_indicators(cfg, cap_t, facts, K, 1)
                sc = get_scores(cfg, P, hh, ee, fe, eib, pi)
            else:
                sc = get_scores(cfg, P, hh, ee)
            logp = sc[0, 0].log_softmax(dim=-1)
            top = logp.topk(min(beam_size, Vx))
            for v, idx in zip(top.values.tolist(), top.indices.tolist()):
                cands.append((h["score"] + v, j, idx, logp))
        cands.sort(key=lambda c: (-c[0], c[1], c[2]))
        new = []
        for score, j, tok, _ in cands[:beam_size]:
            h = hyps[j]
            if h["fin"]:
                new.append(dict(seq=list(h["seq"]), score=h["score"], fin=True))
            else:
                new.append(dict(seq=h["seq"] + [tok], score=score, fin=tok == cfg.end))
        hyps = new + [None] * (beam_size - len(new))
        if all(h is None or h["fin"] for h in hyps):
            break
    live = [h for h in hyps if h is not None]
    best = max(range(len(live)), key=lambda q: (live[q]["score"], -q))
    seq = live[best]["seq"] + [cfg.pad] * (max_pred_len - len(live[best]["seq"]))
    return torch.tensor(seq[:max_pred_len], dtype=torch.long), live[best]["score"], [(h["seq"], h["score"]) for h in live]


def sequence_logprob(cfg, P, enc_out, entities, facts, seq, max_pred_len):
    """Summed log-probability the (teacher-forced) model gives the token sequence `seq` (up to and including <end>)."""
    V, K = cfg.vocab_size, entities.shape[1]
    ee = entity_encode(cfg, P, entities, facts)
    fe = fact_encode(P, facts, ee) if cfg.has_facts else None
    mem = build_memory(cfg, P, enc_out, ee, fe)
    pe = pe_table(max_pred_len, cfg.emb_dim).unsqueeze(0)
    toks = []
    for t in seq:
        toks.append(int(t))
        if int(t) == cfg.end:
            break
    total = 0.0
    for i, t in enumerate(toks):
        prev = toks[:i]
        captions = [cfg.start] + prev + [cfg.start] * (max_pred_len - 1 - len(prev))
        masks = [0] + [2 if (cfg.has_facts and q >= V + K) else (1 if q >= V else 0) for q in prev]
        masks = masks + [0] * (max_pred_len - len(masks))
        cap_t = torch.tensor([captions[:max_pred_len]])
        emb = caption_embed(cfg, P, cap_t, torch.tensor([masks[:max_pred_len]]), ee, fe)
        hh = decoder_stack(cfg, P, emb * math.sqrt(cfg.emb_dim) + pe, mem)[:, i : i + 1]
        if cfg.has_facts:
            eib, pi = context_indicators(cfg, cap_t, facts, K, 1)
            sc = get_scores(cfg, P, hh, ee, fe, eib, pi)
        else:
            sc = get_scores(cfg, P, hh, ee)
        total += float(sc[0, 0].log_softmax(dim=-1)[t])
    return total
