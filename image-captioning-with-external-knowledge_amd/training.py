"""Training step of the caption decoder on MI355X.

Reference call surface (geo-aware/train.py:269-292):
    scores, caps_sorted, decode_lengths = decoder(captions, imgs, masks, lengths, entities[, facts])
    loss = CrossEntropyLoss(ignore_index=<pad>)(pack_padded(scores), pack_padded(caps_sorted[:, 1:]))
    decoder_optimizer.zero_grad(); loss.backward(); clip_gradient(optimizer, 5.0); optimizer.step()

Two ways in:
  * drop-in: in train() mode DecoderTransformer.forward routes through DecoderGraphFn, a
    torch.autograd.Function whose backward runs the hand-written HIP backward pass, so the
    reference's train.py works unchanged (its loss / optimizer stay torch objects);
  * fused: TrainStep owns one flat fp32 bucket for parameters, gradients and Adam moments, runs
    forward -> packed cross entropy -> backward -> (RCCL all-reduce of the bucket) -> clamp + Adam,
    all in HIP kernels, one collective per step (data parallel over the GPUs of a node).
Dropout: the reference constructs the decoder with dropout 0.5/0.5/0.1; the masks come from a
counter-based generator (see ick_dropout) and cannot be bit-identical to torch's CPU stream, so
parity of the training math is pinned with dropout disabled (p = 0 / eval-mode fixtures).
"""
import functools
import math
import os

import torch

from . import dp, ops
from .lib import IckError
from .weights import DerivedWeights      # noqa: F401  (the optimizer-maintained weight images; also imported from here)


class Tape:
    """Activations kept from the forward pass for the backward pass."""

    def __init__(self):
        self.enc_layers = {}   # stack name -> list of per-layer dicts
        self.dec_layers = []
        self.misc = {}


class DropSites:
    """Hands out (p, seed, site) triples: one site id per dropout call site of a step, so the backward
    kernels regenerate exactly the masks the forward used (ick_dropout_mask in include/ick_amd.h)."""

    def __init__(self, seed, enabled, epoch=None):
        self.seed, self.enabled, self.next, self.epoch = seed, enabled, 0, epoch

    def site(self, p):
        if not self.enabled or p <= 0.0:
            return None
        self.next += 1
        return (float(p), self.seed, self.next, self.epoch) if self.epoch is not None else \
            (float(p), self.seed, self.next)


def _p(x):
    return x.detach()


# ----------------------------------------------------------------------------------------------
# forward with saved activations
# ----------------------------------------------------------------------------------------------
def _context_encoder_fwd(dec, stack, x, tape_list, ds, pk=None, tag="e", out=None, slim=False):
    """pk: packed weights (dec._chain_pack) -> the row-chain launches (out-projection + norm1 + linear1, linear2 + norm2
    + the next layer's in_proj) replace the separate GEMM / add & norm kernels; `out`: (B, T, d) view that receives
    the stack's output (its rows of the memory buffer)."""
    H, d = dec.num_heads, dec.emb_dim
    B, T, _ = x.shape
    n = len(stack.layers)
    qkv = None
    for li, layer in enumerate(stack.layers):
        p = layer.dropout.p
        last = li == n - 1
        t = {"x": x, "d_att": ds.site(layer.self_attn.dropout), "d1": ds.site(layer.dropout1.p), "d_ff": ds.site(p),
             "d2": ds.site(layer.dropout2.p)}
        if qkv is not None:
            t["qkv"] = qkv
        else:
            t["qkv"] = ops.project_heads(x, _p(layer.self_attn.in_proj_weight), _p(layer.self_attn.in_proj_bias), 3, H, T)
        qkv = None
        t["sa"] = torch.empty_like(x)
        t["lse"] = torch.empty(B * H * T, device=x.device, dtype=torch.float32)
        ops.attention_heads(t["qkv"], t["qkv"], t["sa"], H, d // H, T, T, 0, 1, 2, lse=t["lse"], drop=t["d_att"])
        x2 = out if (last and out is not None) else torch.empty_like(x)
        if pk is not None:
            t["o1"], t["x1"], t["o2"] = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
            t["f"] = torch.empty(B, T, layer.linear1.out_features, device=x.device, dtype=torch.float32)
            t["m1"], t["r1"] = ops.rowchain_fwd(
                t["sa"], pk[(tag, li, "so")], _p(layer.self_attn.out_proj.bias), x, _p(layer.norm1.weight),
                _p(layer.norm1.bias), layer.norm1.eps, t["x1"], drop1=t["d1"], o_out=t["o1"], save_stats=True,
                w2p=pk[(tag, li, "l1")], b2=_p(layer.linear1.bias), y2=t["f"], relu=True, drop2=t["d_ff"], slim=slim)
            nxt = None if last else stack.layers[li + 1]
            if nxt is not None:
                qkv = torch.empty(B, 3, H, T, ops.DHP, device=x.device, dtype=torch.float32)
            t["m2"], t["r2"] = ops.rowchain_fwd(
                t["f"], pk[(tag, li, "l2")], _p(layer.linear2.bias), t["x1"], _p(layer.norm2.weight),
                _p(layer.norm2.bias), layer.norm2.eps, x2, drop1=t["d2"], o_out=t["o2"], save_stats=True,
                w2p=None if nxt is None else pk[(tag, li + 1, "si")],
                b2=None if nxt is None else _p(nxt.self_attn.in_proj_bias), y2=qkv,
                heads=None if nxt is None else (3, H, T, 0, T), slim=slim)
        else:
            t["o1"] = ops.linear(t["sa"], _p(layer.self_attn.out_proj.weight), _p(layer.self_attn.out_proj.bias))
            t["x1"], t["m1"], t["r1"] = ops.add_layernorm(t["o1"], x, _p(layer.norm1.weight), _p(layer.norm1.bias),
                                                          layer.norm1.eps, save_stats=True, drop=t["d1"])
            t["f"] = ops.linear(t["x1"], _p(layer.linear1.weight), _p(layer.linear1.bias), relu=True, drop=t["d_ff"])
            t["o2"] = ops.linear(t["f"], _p(layer.linear2.weight), _p(layer.linear2.bias))
            y, t["m2"], t["r2"] = ops.add_layernorm(t["o2"], t["x1"], _p(layer.norm2.weight), _p(layer.norm2.bias),
                                                    layer.norm2.eps, save_stats=True, drop=t["d2"])
            if last and out is not None:
                out.copy_(y)
            else:
                x2 = y
        x = x2
        tape_list.append(t)
    return x


def _decoder_self_block(dec, li, layer, x, ds, pk=None, qkv=None):
    """Self-attention block of decoder layer li up to the cross-attention query: in_proj (unless qkv came with the
    previous layer's last launch), causal attention, out-projection + norm1 + q-projection.  Nothing here reads the
    memory, so layer 0's block can run beside Encoder.conv1 / the K/V projection.  Returns the layer's tape dict."""
    H, d = dec.num_heads, dec.emb_dim
    dh = d // H
    B, T, _ = x.shape
    t = {"x": x, "d_sa": ds.site(layer.self_attn.dropout), "d1": ds.site(layer.dropout1.p),
         "d_ca": ds.site(layer.multihead_attn.dropout), "d2": ds.site(layer.dropout2.p),
         "d_ff": ds.site(layer.dropout.p), "d3": ds.site(layer.dropout3.p)}
    if qkv is not None:
        t["qkv"] = qkv
    else:
        t["qkv"] = ops.project_heads(x, _p(layer.self_attn.in_proj_weight), _p(layer.self_attn.in_proj_bias), 3, H, T)
    t["sa"] = torch.empty_like(x)
    t["lse_s"] = torch.empty(B * H * T, device=x.device, dtype=torch.float32)
    ops.attention_heads(t["qkv"], t["qkv"], t["sa"], H, dh, T, T, 0, 1, 2, causal=True, lse=t["lse_s"],
                        drop=t["d_sa"])
    ca_w, ca_b = _p(layer.multihead_attn.in_proj_weight), _p(layer.multihead_attn.in_proj_bias)
    if pk is not None:
        for k in ("o1", "x1"):
            t[k] = torch.empty_like(x)
        t["qc"] = torch.empty(B, 1, H, T, ops.DHP, device=x.device, dtype=torch.float32)
        t["m1"], t["r1"] = ops.rowchain_fwd(
            t["sa"], pk[("d", li, "so")], _p(layer.self_attn.out_proj.bias), x, _p(layer.norm1.weight),
            _p(layer.norm1.bias), layer.norm1.eps, t["x1"], drop1=t["d1"], o_out=t["o1"], save_stats=True,
            w2p=pk[("d", li, "cq")], b2=ca_b[:d], y2=t["qc"], heads=(1, H, T, 0, T))
    else:
        t["o1"] = ops.linear(t["sa"], _p(layer.self_attn.out_proj.weight), _p(layer.self_attn.out_proj.bias))
        t["x1"], t["m1"], t["r1"] = ops.add_layernorm(t["o1"], x, _p(layer.norm1.weight), _p(layer.norm1.bias),
                                                      layer.norm1.eps, save_stats=True, drop=t["d1"])
        t["qc"] = ops.project_heads(t["x1"], ca_w[:d], ca_b[:d], 1, H, T)
    return t


def _decoder_layer_fwd(dec, li, layer, x, kv, S, tape_list, ds, side=None, pk=None, qkv=None, t=None):
    """Returns (x, qkv of the next layer or None).  pk: packed weights -> row-chain launches (see above); then the
    in_proj of layer li + 1 rides on this layer's linear2 + norm3 launch.  t: the tape dict of a self-attention block
    that already ran."""
    H, d = dec.num_heads, dec.emb_dim
    dh = d // H
    B, T, _ = x.shape
    if t is None:
        t = _decoder_self_block(dec, li, layer, x, ds, pk, qkv)
    t["ca"] = torch.empty_like(x)
    t["lse_c"] = torch.empty(B * H * T, device=x.device, dtype=torch.float32)
    if pk is not None:
        for k in ("o2", "x2", "o3"):
            t[k] = torch.empty_like(x)
        ops.stamp("fwd: layer %d reaches cross-attention" % li)
        if side is not None:
            side.wait_or_join("ctx")    # the context rows of kv come from the side stream (its tail may still run)
        ops.attention_heads(t["qc"], kv, t["ca"], H, dh, T, S, 0, 2 * li, 2 * li + 1, lse=t["lse_c"], drop=t["d_ca"])
        t["f"] = torch.empty(B, T, layer.linear1.out_features, device=x.device, dtype=torch.float32)
        t["m2"], t["r2"] = ops.rowchain_fwd(
            t["ca"], pk[("d", li, "co")], _p(layer.multihead_attn.out_proj.bias), t["x1"], _p(layer.norm2.weight),
            _p(layer.norm2.bias), layer.norm2.eps, t["x2"], drop1=t["d2"], o_out=t["o2"], save_stats=True,
            w2p=pk[("d", li, "l1")], b2=_p(layer.linear1.bias), y2=t["f"], relu=True, drop2=t["d_ff"])
        layers = dec.transformer_decoder.layers
        nxt = layers[li + 1] if li + 1 < len(layers) else None
        x3 = torch.empty_like(x)
        qkv_n = None if nxt is None else torch.empty(B, 3, H, T, ops.DHP, device=x.device, dtype=torch.float32)
        t["m3"], t["r3"] = ops.rowchain_fwd(
            t["f"], pk[("d", li, "l2")], _p(layer.linear2.bias), t["x2"], _p(layer.norm3.weight), _p(layer.norm3.bias),
            layer.norm3.eps, x3, drop1=t["d3"], o_out=t["o3"], save_stats=True,
            w2p=None if nxt is None else pk[("d", li + 1, "si")],
            b2=None if nxt is None else _p(nxt.self_attn.in_proj_bias), y2=qkv_n,
            heads=None if nxt is None else (3, H, T, 0, T))
        tape_list.append(t)
        return x3, qkv_n
    ops.stamp("fwd: layer %d reaches cross-attention" % li)
    if side is not None:
        side.wait_or_join("ctx")    # the context rows of kv come from the side stream (its tail may still run)
    ops.attention_heads(t["qc"], kv, t["ca"], H, dh, T, S, 0, 2 * li, 2 * li + 1, lse=t["lse_c"], drop=t["d_ca"])
    t["o2"] = ops.linear(t["ca"], _p(layer.multihead_attn.out_proj.weight), _p(layer.multihead_attn.out_proj.bias))
    t["x2"], t["m2"], t["r2"] = ops.add_layernorm(t["o2"], t["x1"], _p(layer.norm2.weight), _p(layer.norm2.bias),
                                                  layer.norm2.eps, save_stats=True, drop=t["d2"])
    t["f"] = ops.linear(t["x2"], _p(layer.linear1.weight), _p(layer.linear1.bias), relu=True, drop=t["d_ff"])
    t["o3"] = ops.linear(t["f"], _p(layer.linear2.weight), _p(layer.linear2.bias))
    x, t["m3"], t["r3"] = ops.add_layernorm(t["o3"], t["x2"], _p(layer.norm3.weight), _p(layer.norm3.bias),
                                            layer.norm3.eps, save_stats=True, drop=t["d3"])
    tape_list.append(t)
    return x, None


def forward_with_tape(dec, captions, caption_masks, entities, facts, enc_tok, gmap, seed=0, epoch=None,
                      fresh_pack=False, overlap=False, feats=None, conv1=None, side_tail=None, derived=None, pre_side=None):
    """Teacher-forced forward on already length-sorted inputs; returns (scores, tape).  Dropout is
    active iff the module is in train() mode (masks derive from `seed` + the device counter `epoch`).
    fresh_pack: rebuild the packed cross-K/V / transposed predicate weights from the live parameters
    (needed when they are updated behind torch's version counters, and inside captured graphs).
    derived: TrainStep's persistent re-laid-out copies of the weights (DerivedWeights: packed row-chain copies, forward
    and transposed, the gathered cross K/V weight / bias and the bf16 planes of the large GEMMs' weights), kept current by
    the optimizer kernel itself -- no packing launch at all in this pass.
    overlap: run the context-encoder chain (small, latency-bound kernels) on a second stream beside the
    image-row K/V projection and the first self-attention block (for captured graphs; eager launches are
    host-bound and gain nothing).
    feats + conv1 = (weight, bias): instead of enc_tok, the (B, 2048, 14, 14) feature map; Encoder.conv1 then runs
    here and writes the image rows straight into the memory buffer (no (B, 196, d) intermediate, no copy).
    side_tail: work of the caller that nothing in the forward pass waits for (TrainStep: zeroing the gradient bucket,
    the decode lengths); with `overlap` it runs on the side stream once the context chain is done, else right away.
    pre_side (needs overlap, feats and derived): work of the caller that must run BEFORE anything here reads a parameter
    -- TrainStep's deferred optimizer update of the previous step.  It opens the side stream, beside Encoder.conv1 on the
    main stream (frozen weights, input features: the one large kernel of the step that reads no trainable parameter);
    the entity / fact encoders follow it there and the main stream waits for that point behind conv1."""
    tape = Tape()
    m = tape.misc
    ds = DropSites(seed, dec.training, epoch)
    d, V, H = dec.emb_dim, dec.vocab_size, dec.num_heads
    B, L = captions.shape
    P = feats.shape[2] * feats.shape[3] if feats is not None else enc_tok.shape[1]
    K = entities.shape[1]
    dev = captions.device
    Fn = facts.shape[1] if dec.has_facts else 0
    nseg = 2 * len(dec.transformer_decoder.layers)
    S = P + K + Fn
    # memory rows [image ; entities ; facts], kept contiguous (B, S, d) for the weight gradient of the K/V
    # projection; its three row groups are projected separately (disjoint rows of the head-major kv buffer)
    mem = torch.empty(B, S, d, device=dev, dtype=torch.float32)
    kv = torch.empty(B, nseg, H, S, ops.DHP, device=dev, dtype=torch.float32)
    tape.enc_layers["entities"] = []
    if dec.has_facts:
        tape.enc_layers["facts"] = []
    side = ops.SideStream(priority=-1) if overlap else None
    if side is None and side_tail is not None:
        side_tail()

    chain = dec.chain_supported()
    staged = chain and fresh_pack and overlap and derived is None
    first = lambda k: k[0] != "d" or (k[1] == 0 and k[2] in ("so", "cq", "si"))     # context encoders + layer 0's self block
    split_on = ops.gemm_split_mode() >= 1 and not ops.is_deterministic()

    def encode_context_inputs():
        ee_ = ops.entity_encode(dec.variant, entities, _p(dec.entity_encoder.type_embedding.weight), d,
                                facts=facts if dec.has_facts else None,
                                word_emb=_p(dec.word_embedding.weight) if dec.variant == "news" else None)
        fe_ = ops.fact_encode(facts, ee_, _p(dec.predicate_embedding.weight)) if dec.has_facts else None
        return ee_, fe_

    lazy = pre_side is not None and side is not None and feats is not None and derived is not None
    if pre_side is not None and not lazy:
        pre_side()
    ee, fe = (None, None) if lazy else encode_context_inputs()
    if derived is not None:
        wkv, bkv, pk = derived.wkv, derived.bkv, derived.pk
        m["pkb"] = derived.pkb
        if split_on:
            m["wkv_ps"] = derived.wkv_ps
            if B * L >= 256:
                m["vocab_ps"], m["vocab_t_ps"] = derived.vocab_ps, derived.vocab_t_ps
    else:
        copies = []
        if fresh_pack:
            layers_ = dec.transformer_decoder.layers
            if chain:
                # the all-layer cross K/V weight and bias gathered by the packing launch below (two torch.cat
                # launches less in front of Encoder.conv1 and the context chain)
                wkv = torch.empty(nseg * d, d, device=dev, dtype=torch.float32)
                bkv = torch.empty(nseg * d, device=dev, dtype=torch.float32)
                for i_, l in enumerate(layers_):
                    copies.append((_p(l.multihead_attn.in_proj_weight)[d:], wkv[2 * d * i_:2 * d * (i_ + 1)]))
                    copies.append((_p(l.multihead_attn.in_proj_bias)[d:].view(1, -1),
                                   bkv[2 * d * i_:2 * d * (i_ + 1)].view(1, -1)))
            else:
                wkv = torch.cat([_p(l.multihead_attn.in_proj_weight)[d:] for l in layers_])
                bkv = torch.cat([_p(l.multihead_attn.in_proj_bias)[d:] for l in layers_])
        else:
            wkv, bkv = dec._packed_cross_kv()
        # packed weight copies of the row-chain launches.  Inside a captured step that does not keep them current itself
        # (derived is None) they are refreshed in three launches placed where they cost least: the context encoders'
        # copies now (the side chain needs them first), the decoder layers' after Encoder.conv1 has been enqueued, the
        # transposed copies of the backward chains on the side stream once the context chain is done.
        pk = dec._chain_pack(fresh=fresh_pack, subset=first if staged else None, copies=copies) if chain else None

    def kv_presplit():
        # pre-split copy (three bf16 planes) of the all-layer cross K/V weight for the image rows' projection (ick_gemm's
        # b_ps, csrc/gemm_ps.hip); with a side stream it is made behind the fork: only the main stream reads it
        if derived is None and split_on:
            if fresh_pack:
                m["wkv_ps"] = ops.presplit_buffer(nseg * d, d, dev)
                ops.presplit_weights([(wkv, m["wkv_ps"])])
            else:
                m["wkv_ps"] = dec._cross_kv_presplit(wkv)

    if side is None:
        kv_presplit()

    def vocab_presplit():
        # fc_vocab's pre-split copies: nothing needs them before the score head, so they are made where the side stream
        # idles.  The transposed one is the B operand of the data gradient dh = dscores @ W on the pre-split kernel (128 x 80
        # tile, 40 tiles x 12 K slices: cfg2 train step 1.770 -> 1.741 ms, profiles/r04_y_ab_vocab_dgrad_ps.txt)
        if derived is None and split_on and B * L >= 256:
            if fresh_pack:
                m["vocab_ps"] = ops.presplit_buffer(V, d, dev)
                m["vocab_t_ps"] = ops.presplit_buffer(d, V, dev)
                ops.presplit_weights([(_p(dec.fc_vocab.weight), m["vocab_ps"]),
                                      (_p(dec.fc_vocab.weight).t(), m["vocab_t_ps"])])
            else:
                m["vocab_ps"] = dec._vocab_presplit()

    head = {}

    def entity_chain():
        nonlocal ee, fe
        if lazy:
            ops.stamp("side: deferred optimizer update starts")
            pre_side()
            ee, fe = head["ee"], head["fe"] = encode_context_inputs()
            side.signal("head")
        ops.stamp("side: context chain starts")
        # the stack's last add & norm writes the entity rows of the memory buffer directly; beside Encoder.conv1 and the
        # image K/V projection the chain runs in its 8-wave form, which finds room on the CUs the bulk GEMMs occupy
        ctx_e = _context_encoder_fwd(dec, dec.transformer_encoder_entities, ee, tape.enc_layers["entities"], ds, pk=pk,
                                     tag="e", out=mem[:, P:P + K], slim=overlap)
        ops.project_heads(ctx_e, wkv, bkv, nseg, H, S, out=kv, s0=P, grp=K)
        if side is not None:
            side.signal("ctx")      # the first cross-attention waits for this point, not for the packing / zeroing below
        ops.stamp("side: context chain done")
        # bulk work nothing waits for before the score head / the backward pass
        if staged and dec.chain_bwd_supported():
            m["pkb"] = dec._chain_pack(fresh=True, bwd=True, extra=[(("kv", "T"), wkv.t())])
        vocab_presplit()
        if side is not None and side_tail is not None:
            side_tail()

    def fact_chain():
        ctx_f = _context_encoder_fwd(dec, dec.transformer_encoder_facts, fe, tape.enc_layers["facts"], ds, pk=pk,
                                     tag="f", out=mem[:, P + K:], slim=overlap)
        ops.project_heads(ctx_f, wkv, bkv, nseg, H, S, out=kv, s0=P + K, grp=Fn)

    pe = dec.pos_encoder.pe.view(-1, d)
    if feats is not None:
        img = mem[:, :P]
    else:
        img = enc_tok.index_select(0, gmap.long()) if gmap is not None else enc_tok
    if side is not None:
        side.submit(entity_chain, ee, fe, mem, kv, wkv, bkv, img, entities, facts)
    else:
        entity_chain()
    if feats is not None:
        cw, cb = conv1[0], conv1[1]
        Cc = feats.shape[1]
        ops.gemm_raw(feats, cw.view(d, Cc), mem, B * P, d, Cc, 1, P, Cc, 1, d, bias=cb, a_grp=P, a_gs=Cc * P,
                     c_grp=P, c_gs=S * d, b_ps=conv1[2] if len(conv1) > 2 else None)
    else:
        mem[:, :P].copy_(img)
    if side is not None:
        side.flush()     # enqueued after the main stream's next kernel (see SideStream)
        if lazy:
            side.wait("head")      # from here on the main stream reads parameters (and the encoded entity / fact rows)
            ee, fe = head["ee"], head["fe"]
        kv_presplit()
    if staged:
        dec._chain_pack(fresh=True, subset=lambda k: not first(k))
    elif derived is None and pk is not None and dec.chain_bwd_supported():
        m["pkb"] = dec._chain_pack(fresh=fresh_pack, bwd=True, extra=[(("kv", "T"), wkv.t())])
    if dec.has_facts:
        # on the main stream, beside the entity chain on the side stream: two chains of small kernels overlap well
        # (a chain beside the large projection below does not)
        fact_chain()
    # One GEMM for the image rows of every layer.  Projecting the later layers' rows on the side stream after the
    # chain was measured (device time stamps): the chain ends 70 us earlier, the first decoder layer 90 us later --
    # a 5000-workgroup GEMM beside a chain of small kernels delays the chain by about its own duration either way.
    ops.project_heads(img, wkv, bkv, nseg, H, S, out=kv, s0=0, grp=P, w_ps=m.get("wkv_ps"))
    ops.stamp("fwd: image K/V projection done")
    m["d_pos"] = ds.site(dec.pos_encoder.dropout.p)
    x = ops.caption_embed(captions, caption_masks, _p(dec.word_embedding.weight), ee, fe, pe, V,
                          dec.word_map["<pad>"], math.sqrt(d), drop=m["d_pos"])
    t0 = _decoder_self_block(dec, 0, dec.transformer_decoder.layers[0], x, ds, pk, None)
    qkv = None
    for li, layer in enumerate(dec.transformer_decoder.layers):
        x, qkv = _decoder_layer_fwd(dec, li, layer, x, kv, S, tape.dec_layers, ds, side=side if li == 0 else None,
                                    pk=pk, qkv=qkv, t=t0 if li == 0 else None)
    if side is not None:
        side.join()
    ops.stamp("fwd: decoder layers done")
    eib = gate = hv = None
    if dec.has_facts:
        if derived is not None:
            pred_wt = derived.pred_wt
        else:
            pred_wt = _p(dec.fc_predicate.weight).t().contiguous() if fresh_pack else dec._pred_wt()
        eib, gate = ops.context_indicators(captions, facts, K, V, pred_wt, _p(dec.fc_predicate.bias), mode=0)
        hv = ops.mul(x, gate)
    Vx = V + K + Fn
    # rows padded to a multiple of 4 floats (knowledge: 50 000 + 20 + 51 = 50 071 columns): the vocabulary's data- and
    # weight-gradient GEMMs read the score gradients with 16-byte loads only from aligned rows (645 -> ~470 us)
    ld = (Vx + 3) // 4 * 4
    scores = torch.empty(B, L, ld, device=x.device, dtype=torch.float32)[:, :, :Vx]
    ops.gemm_raw(hv if dec.has_facts else x, _p(dec.fc_vocab.weight), scores, B * L, V, d, d, 1, d, 1, ld,
                 bias=_p(dec.fc_vocab.bias), b_ps=m.get("vocab_ps"))
    ops.pointer_scores(x, ee, _p(dec.fc_entity.weight), _p(dec.fc_entity.bias), scores, V)
    if dec.has_facts:
        ops.pointer_scores(x, fe, _p(dec.fc_fact.weight), _p(dec.fc_fact.bias), scores, V + K, ind=eib)
    m.update(ee=ee, fe=fe, mem=mem, kv=kv, h=x, hv=hv, eib=eib, gate=gate, captions=captions, masks=caption_masks,
             entities=entities, facts=facts, P=P, K=K, Fn=Fn, S=S, wkv=wkv)
    return scores, tape


# ----------------------------------------------------------------------------------------------
# backward
# ----------------------------------------------------------------------------------------------
def _keep_scale(drop):
    return 1.0 if drop is None else 1.0 / (1.0 - drop[0])


def _g(grads, param):
    """Gradient buffer of a parameter (None when it is frozen)."""
    return grads.get(id(param))


def _lin_bwd(grads, dy2, x2, lin_w, lin_b, w_rows=None, need_dx=True, dx=None, acc=False, group_now=False,
             gate=None, gate_scale=1.0, wt_ps=None, xt_ps=None):
    """Backward of a Linear whose weight is `lin_w` (optionally the row slice w_rows of it)."""
    gw, gb = _g(grads, lin_w), _g(grads, lin_b)
    w = _p(lin_w)
    if w_rows is not None:
        w = w[w_rows]
        gw = gw[w_rows] if gw is not None else None
        gb = gb[w_rows] if gb is not None else None
    return ops.linear_bwd(dy2, x2, w, gw, gb, need_dx=need_dx, dx=dx, accumulate_dx=acc, group_now=group_now,
                          gate=gate, gate_scale=gate_scale, wt_ps=wt_ps, xt_ps=xt_ps)


def _prezeroed(dec, captions, entities, facts):
    """Accumulation targets of the backward pass's first kernels (the vocabulary data gradient's split-K sum, the
    gradient of the encoded entity / fact rows), zeroed where nothing waits for it -- the forward pass's side stream --
    instead of between the loss and the first backward GEMM (two fills + a gap: ~13 us of the step's critical path)."""
    B, L = captions.shape[0], captions.shape[1]
    d, dev = dec.emb_dim, captions.device
    pre = {"dhv": torch.zeros(B * L, d, device=dev, dtype=torch.float32),
           "dee": torch.zeros(B, entities.shape[1], d, device=dev, dtype=torch.float32)}
    if facts is not None:
        pre["dfe"] = torch.zeros(B, facts.shape[1], d, device=dev, dtype=torch.float32)
    return pre


def _norm_args(t, i, res, layer_norm, grads, M, d, dev):
    """The saved forward tensors of add & norm number i of a layer + fresh outputs, as ops.rowchain_bwd wants them."""
    return dict(o=t["o%d" % i], res=res, mean=t["m%d" % i], rstd=t["r%d" % i], gamma=_p(layer_norm.weight),
                drop=t["d%d" % i], do=torch.empty(M, d, device=dev, dtype=torch.float32),
                part=ops.ln_partials(M, d, dev))


def _context_encoder_bwd(dec, stack, tapes, dx, grads, pkb=None, tag="e", g_first=None):
    """pkb: packed transposed weights (dec._chain_pack(bwd=True)) -> one ops.rowchain_bwd launch per layer for
    [in_proj data gradient of the layer above] + norm2' + linear2' + ReLU' + linear1' + norm1' + out_proj' instead
    of six kernels.  g_first = (g0 rows view, packed W0^T): the incoming gradient is g0 @ W0 (no dx tensor)."""
    H, d = dec.num_heads, dec.emb_dim
    layers = list(stack.layers)
    if pkb is not None:
        dev = tapes[0]["x"].device
        dz, g0, w0p = dx, None, None
        if g_first is not None:
            dz, (g0, w0p) = None, g_first
        for li in reversed(range(len(layers))):
            layer, t = layers[li], tapes[li]
            B, T, _ = t["x"].shape
            M = B * T
            n2 = _norm_args(t, 2, t["x1"], layer.norm2, grads, M, d, dev)
            n1 = _norm_args(t, 1, t["x"], layer.norm1, grads, M, d, dev)
            dpre = torch.empty(M, layer.linear1.out_features, device=dev, dtype=torch.float32)
            dsa = torch.empty(M, d, device=dev, dtype=torch.float32)
            dz_out = torch.empty(M, d, device=dev, dtype=torch.float32)
            ops.rowchain_bwd(M, d, n2, pkb[(tag, li, "soT")], dsa, dz_out, g0=g0, w0p=w0p, dzin=dz,
                             ffn=dict(w1p=pkb[(tag, li, "l2T")], w2p=pkb[(tag, li, "l1T")], act=t["f"],
                                      gate_scale=_keep_scale(t["d_ff"]), t_out=dpre), norm2=n1)
            if ops.SIDE is not None:
                ops.SIDE.flush()
            ops.ln_partials_reduce(n2["part"], _g(grads, layer.norm2.weight), _g(grads, layer.norm2.bias))
            ops.ln_partials_reduce(n1["part"], _g(grads, layer.norm1.weight), _g(grads, layer.norm1.bias))
            _lin_bwd(grads, n2["do"], t["f"].view(M, -1), layer.linear2.weight, layer.linear2.bias, need_dx=False)
            _lin_bwd(grads, dpre, t["x1"].view(M, d), layer.linear1.weight, layer.linear1.bias, need_dx=False)
            _lin_bwd(grads, n1["do"], t["sa"].view(M, d), layer.self_attn.out_proj.weight, layer.self_attn.out_proj.bias,
                     need_dx=False)
            if ops.SIDE is not None and li > 0:
                # everything queued so far only needs the chain launch above: it goes out now, beside the attention
                # backward (the in_proj weight gradient follows with the next group) -- the side stream's last group,
                # which nothing on the main stream overlaps any more, shrinks to one problem
                ops.SIDE.flush_group()
            dqkv = ops.attention_bwd_buffer((B, T, 3 * d), T, T, d // H, dev)
            ops.attention_heads_bwd(t["qkv"], t["qkv"], t["sa"], dsa.view(B, T, d), t["lse"], dqkv[:, :, :d],
                                    dqkv[:, :, d:2 * d], dqkv[:, :, 2 * d:], H, d // H, T, T, 0, 1, 2, drop=t["d_att"])
            if ops.SIDE is not None:
                ops.SIDE.flush()
            if li > 0:
                # the in_proj data gradient rides on the next launch (the layer below); only its weight gradient here
                _lin_bwd(grads, dqkv.view(M, 3 * d), t["x"].view(M, d), layer.self_attn.in_proj_weight,
                         layer.self_attn.in_proj_bias, need_dx=False)
                dz, g0, w0p = dz_out, dqkv.view(M, 3 * d), pkb[(tag, li, "siT")]
            else:
                dx = _lin_bwd(grads, dqkv.view(M, 3 * d), t["x"].view(M, d), layer.self_attn.in_proj_weight,
                              layer.self_attn.in_proj_bias, dx=dz_out, acc=True).view(B, T, d)
                if ops.SIDE is not None:
                    # the stack's first layer is the last thing the backward pass computes: the main stream has
                    # nothing left while the side stream still works off the layers above -- its weight gradients
                    # run here instead of queueing behind them
                    ops.SIDE.flush_group_here()
        return dx
    for layer, t in zip(reversed(layers), reversed(tapes)):
        B, T, _ = t["x"].shape
        M = B * T
        dz, do2 = ops.layernorm_bwd(dx, t["o2"], t["x1"], _p(layer.norm2.weight), t["m2"], t["r2"],
                                    _g(grads, layer.norm2.weight), _g(grads, layer.norm2.bias), drop=t["d2"])
        dpre = _lin_bwd(grads, do2.view(M, d), t["f"].view(M, -1), layer.linear2.weight, layer.linear2.bias,
                        gate=t["f"].view(M, -1), gate_scale=_keep_scale(t["d_ff"]))   # ReLU' (+ dropout) fused
        dx1 = _lin_bwd(grads, dpre, t["x1"].view(M, d), layer.linear1.weight, layer.linear1.bias, dx=dz.view(M, d),
                       acc=True)
        dz, do1 = ops.layernorm_bwd(dx1.view(B, T, d), t["o1"], t["x"], _p(layer.norm1.weight), t["m1"], t["r1"],
                                    _g(grads, layer.norm1.weight), _g(grads, layer.norm1.bias), drop=t["d1"])
        dsa = _lin_bwd(grads, do1.view(M, d), t["sa"].view(M, d), layer.self_attn.out_proj.weight,
                       layer.self_attn.out_proj.bias)
        dqkv = ops.attention_bwd_buffer((B, T, 3 * d), T, T, d // H, dx.device)
        ops.attention_heads_bwd(t["qkv"], t["qkv"], t["sa"], dsa.view(B, T, d), t["lse"], dqkv[:, :, :d],
                                dqkv[:, :, d:2 * d], dqkv[:, :, 2 * d:], H, d // H, T, T, 0, 1, 2, drop=t["d_att"])
        dx = _lin_bwd(grads, dqkv.view(M, 3 * d), t["x"].view(M, d), layer.self_attn.in_proj_weight,
                      layer.self_attn.in_proj_bias, dx=dz.view(M, d), acc=True).view(B, T, d)
        if ops.SIDE is not None:
            ops.SIDE.flush_group()      # this layer's weight gradients: one grouped launch
    return dx


def _decoder_layer_bwd_chain(dec, li, layer, t, state, dkv_rows, kv, S, grads, pkb, mem2=None, mem_t_ps=None):
    """Backward of decoder layer li as two ops.rowchain_bwd launches around the cross-attention backward + the
    self-attention backward.  state = (dz, g0, w0p): the residual-path gradient of this layer's output, and -- from the
    layer above -- the in_proj gradient whose data gradient rides on this layer's first launch.  Returns the state
    for the layer below."""
    H, d = dec.num_heads, dec.emb_dim
    dh = d // H
    B, T, _ = t["x"].shape
    M = B * T
    dev = t["x"].device
    dz, g0, w0p = state
    n3 = _norm_args(t, 3, t["x2"], layer.norm3, grads, M, d, dev)
    n2 = _norm_args(t, 2, t["x1"], layer.norm2, grads, M, d, dev)
    dpre = torch.empty(M, layer.linear1.out_features, device=dev, dtype=torch.float32)
    dca = torch.empty(M, d, device=dev, dtype=torch.float32)
    dz_a = torch.empty(M, d, device=dev, dtype=torch.float32)
    ops.rowchain_bwd(M, d, n3, pkb[("d", li, "coT")], dca, dz_a, g0=g0, w0p=w0p, dzin=dz,
                     ffn=dict(w1p=pkb[("d", li, "l2T")], w2p=pkb[("d", li, "l1T")], act=t["f"],
                              gate_scale=_keep_scale(t["d_ff"]), t_out=dpre), norm2=n2)
    if ops.SIDE is not None:
        ops.SIDE.flush()
    ops.ln_partials_reduce(n3["part"], _g(grads, layer.norm3.weight), _g(grads, layer.norm3.bias))
    ops.ln_partials_reduce(n2["part"], _g(grads, layer.norm2.weight), _g(grads, layer.norm2.bias))
    _lin_bwd(grads, n3["do"], t["f"].view(M, -1), layer.linear2.weight, layer.linear2.bias, need_dx=False)
    _lin_bwd(grads, dpre, t["x2"].view(M, d), layer.linear1.weight, layer.linear1.bias, need_dx=False)
    _lin_bwd(grads, n2["do"], t["ca"].view(M, d), layer.multihead_attn.out_proj.weight,
             layer.multihead_attn.out_proj.bias, need_dx=False)
    # The layer's weight gradients go out in two grouped launches: linear2 / linear1 / cross out-projection + the two
    # norms' partials right behind the chain launch above, the rest at the layer's end (the side stream finishes last:
    # starting its work earlier shortens the step, 651 -> 654 k decode-steps/s; a third launch right behind the
    # cross-attention backward -- the K/V-projection weight gradient behind its operand -- costs more in fork points than
    # it gains: round 2 645 k, round 5 1.733 -> 1.739 ms, profiles/r05_x_ab_kv_wgrad_subgroups.txt)
    if ops.SIDE is not None:
        ops.SIDE.flush_group()
    dq = torch.empty(B, T, d, device=dev, dtype=torch.float32)
    c0 = 2 * li * d
    ops.attention_heads_bwd(t["qc"], kv, t["ca"], dca.view(B, T, d), t["lse_c"], dq, dkv_rows[:, :, c0:c0 + d],
                            dkv_rows[:, :, c0 + d:c0 + 2 * d], H, dh, T, S, 0, 2 * li, 2 * li + 1, drop=t["d_ca"])
    if ops.SIDE is not None:
        ops.SIDE.flush()
    if mem2 is not None:
        _kv_proj_param_grads(layer, li, dkv_rows, mem2, grads, d, mem_t_ps)
    n1 = _norm_args(t, 1, t["x"], layer.norm1, grads, M, d, dev)
    dsa = torch.empty(M, d, device=dev, dtype=torch.float32)
    dz_b = torch.empty(M, d, device=dev, dtype=torch.float32)
    ops.rowchain_bwd(M, d, n1, pkb[("d", li, "soT")], dsa, dz_b, g0=dq.view(M, d), w0p=pkb[("d", li, "cqT")], dzin=dz_a)
    if ops.SIDE is not None:
        ops.SIDE.flush()
    ops.ln_partials_reduce(n1["part"], _g(grads, layer.norm1.weight), _g(grads, layer.norm1.bias))
    _lin_bwd(grads, dq.view(M, d), t["x1"].view(M, d), layer.multihead_attn.in_proj_weight,
             layer.multihead_attn.in_proj_bias, w_rows=slice(0, d), need_dx=False)
    _lin_bwd(grads, n1["do"], t["sa"].view(M, d), layer.self_attn.out_proj.weight, layer.self_attn.out_proj.bias,
             need_dx=False)
    dqkv = ops.attention_bwd_buffer((B, T, 3 * d), T, T, dh, dev)
    ops.attention_heads_bwd(t["qkv"], t["qkv"], t["sa"], dsa.view(B, T, d), t["lse_s"], dqkv[:, :, :d],
                            dqkv[:, :, d:2 * d], dqkv[:, :, 2 * d:], H, dh, T, T, 0, 1, 2, causal=True,
                            drop=t["d_sa"])
    if li > 0:
        _lin_bwd(grads, dqkv.view(M, 3 * d), t["x"].view(M, d), layer.self_attn.in_proj_weight,
                 layer.self_attn.in_proj_bias, need_dx=False)
        out = (dz_b, dqkv.view(M, 3 * d), pkb[("d", li, "siT")])
    else:
        dx0 = _lin_bwd(grads, dqkv.view(M, 3 * d), t["x"].view(M, d), layer.self_attn.in_proj_weight,
                       layer.self_attn.in_proj_bias, dx=dz_b, acc=True).view(B, T, d)
        out = (dx0, None, None)
    if ops.SIDE is not None:
        ops.SIDE.flush_group()          # this layer's weight gradients: one grouped launch
    return out


def _decoder_layer_bwd(dec, li, layer, t, dx, dkv_rows, kv, S, grads, mem2=None, mem_t_ps=None):
    H, d = dec.num_heads, dec.emb_dim
    dh = d // H
    B, T, _ = t["x"].shape
    M = B * T
    dz, do3 = ops.layernorm_bwd(dx, t["o3"], t["x2"], _p(layer.norm3.weight), t["m3"], t["r3"],
                                _g(grads, layer.norm3.weight), _g(grads, layer.norm3.bias), drop=t["d3"])
    dpre = _lin_bwd(grads, do3.view(M, d), t["f"].view(M, -1), layer.linear2.weight, layer.linear2.bias,
                    gate=t["f"].view(M, -1), gate_scale=_keep_scale(t["d_ff"]))       # ReLU' (+ dropout) fused
    dx2 = _lin_bwd(grads, dpre, t["x2"].view(M, d), layer.linear1.weight, layer.linear1.bias, dx=dz.view(M, d),
                   acc=True)
    dz, do2 = ops.layernorm_bwd(dx2.view(B, T, d), t["o2"], t["x1"], _p(layer.norm2.weight), t["m2"], t["r2"],
                                _g(grads, layer.norm2.weight), _g(grads, layer.norm2.bias), drop=t["d2"])
    dca = _lin_bwd(grads, do2.view(M, d), t["ca"].view(M, d), layer.multihead_attn.out_proj.weight,
                   layer.multihead_attn.out_proj.bias)
    dq = torch.empty(B, T, d, device=dx.device, dtype=torch.float32)
    c0 = 2 * li * d
    ops.attention_heads_bwd(t["qc"], kv, t["ca"], dca.view(B, T, d), t["lse_c"], dq, dkv_rows[:, :, c0:c0 + d],
                            dkv_rows[:, :, c0 + d:c0 + 2 * d], H, dh, T, S, 0, 2 * li, 2 * li + 1, drop=t["d_ca"])
    if mem2 is not None:
        # this layer's columns of the K/V-projection gradient are complete: its weight (and bias) gradient over
        # all B * S memory rows joins this layer's group instead of waiting for the whole decoder stack
        _kv_proj_param_grads(layer, li, dkv_rows, mem2, grads, d, mem_t_ps)
    dx1 = _lin_bwd(grads, dq.view(M, d), t["x1"].view(M, d), layer.multihead_attn.in_proj_weight,
                   layer.multihead_attn.in_proj_bias, w_rows=slice(0, d), dx=dz.view(M, d), acc=True)
    dz, do1 = ops.layernorm_bwd(dx1.view(B, T, d), t["o1"], t["x"], _p(layer.norm1.weight), t["m1"], t["r1"],
                                _g(grads, layer.norm1.weight), _g(grads, layer.norm1.bias), drop=t["d1"])
    dsa = _lin_bwd(grads, do1.view(M, d), t["sa"].view(M, d), layer.self_attn.out_proj.weight,
                   layer.self_attn.out_proj.bias)
    dqkv = ops.attention_bwd_buffer((B, T, 3 * d), T, T, dh, dx.device)
    ops.attention_heads_bwd(t["qkv"], t["qkv"], t["sa"], dsa.view(B, T, d), t["lse_s"], dqkv[:, :, :d],
                            dqkv[:, :, d:2 * d], dqkv[:, :, 2 * d:], H, dh, T, T, 0, 1, 2, causal=True,
                            drop=t["d_sa"])
    dx0 = _lin_bwd(grads, dqkv.view(M, 3 * d), t["x"].view(M, d), layer.self_attn.in_proj_weight,
                   layer.self_attn.in_proj_bias, dx=dz.view(M, d), acc=True).view(B, T, d)
    if ops.SIDE is not None:
        ops.SIDE.flush_group()          # this layer's weight gradients: one grouped launch
    return dx0


def _kv_proj_param_grads(layer, li, dkv_rows, mem2, grads, d, mem_t_ps=None):
    """dW[d:3d] += dkv[:, layer columns].T @ memory, db[d:3d] += column sums (fused), for one decoder layer.
    mem_t_ps: the pre-split copy of memory^T (d x rows; made once per backward pass on the side stream): the product then
    runs on the LDS-DMA kernel of csrc/gemm_ps.hip (600 x 300 outputs over 13 824 rows: 75-88 -> ~50 us per layer) and the
    bias gradient becomes a plain column sum in the same side-stream group."""
    gw, gb = _g(grads, layer.multihead_attn.in_proj_weight), _g(grads, layer.multihead_attn.in_proj_bias)
    rows = mem2.shape[0]
    sl = dkv_rows.view(rows, -1)[:, 2 * li * d:(2 * li + 2) * d]
    if gw is not None:
        if mem_t_ps is not None:
            # 5 x 4 tiles of 128 x 80: K slices so that ~160 workgroups exist -- 8 slices, one per XCD (the kernel deals the
            # (slice, tile) pairs to the XCDs slice-major), half the float atomics of 16 slices and one workgroup per CU,
            # which leaves the CU's other slot to the main stream's kernels.  Sweeps inside the step: round 4
            # (profiles/r04_y_ab_bwd_splits.txt) 16 slices 1.695-1.705 ms against 12 / 24 / 32 at 1.711-1.720; round 5
            # (profiles/r05_x_ab_kv_wgrad_slices.txt) 8 slices 1.703-1.726 against 16 at 1.727-1.743 on the same boxes,
            # 4 / 6 / 10 / 24 slices 1.85 / 1.77 / 1.76 / 1.74-1.75
            split = max(1, min(32, 160 // (((2 * d + 127) // 128) * ((d + 79) // 80)), rows // 256))
            wg = ops.gemm_args(sl, mem2, gw[d:], 2 * d, d, rows, 1, sl.stride(0), 1, d, d, atomic=True, split_k=split,
                               b_ps=mem_t_ps)
            extra = [] if gb is None else [ops.colsum_problem(sl, gb[d:], split_k=max(1, min(16, rows // 512)))]
        else:
            wg = ops.gemm_args(sl, mem2, gw[d:], 2 * d, d, rows, 1, sl.stride(0), 1, d, d, atomic=True,
                               split_k=1 if ops.is_deterministic() else 16,
                               colsum_a=None if gb is None else gb[d:])
            extra = []
        if ops.SIDE is not None:
            ops.SIDE.add_problem(wg, dkv_rows, mem2, mem_t_ps)
            for e in extra:
                ops.SIDE.add_problem(e, dkv_rows)
        else:
            ops.gemm_grouped([wg] + extra)
    elif gb is not None:
        ops.colsum(sl, gb[d:])


def _memory_t_presplit(m, B, S, d):
    """Pre-split copy of memory^T (csrc/gemm_ps.hip's B operand for the cross K/V weight gradients), made where the side
    stream starts its backward work; None in the exact / deterministic modes."""
    if ops.gemm_split_mode() < 1 or ops.is_deterministic() or B * S < 2048:
        return None
    mem2 = m["mem"].view(B * S, d)
    buf = ops.presplit_buffer(d, B * S, mem2.device)

    def make():
        ops.presplit_weights([(mem2.t(), buf)])

    if ops.SIDE is not None:
        ops.SIDE.submit(make, mem2, buf)
    else:
        make()
    return buf


def _rows_t_presplit(x2):
    """Pre-split copy of x2^T (x2: (rows, d) activations): the B operand of a weight gradient dW = dy^T @ x2 on
    csrc/gemm_ps.hip's kernel -- the vocabulary's (10 000 x 300 outputs over the 1 280 caption rows), made on the side
    stream in front of that gradient; None in the exact / deterministic modes and for few rows."""
    rows, d = x2.shape
    if ops.gemm_split_mode() < 1 or ops.is_deterministic() or rows < 1024:
        return None
    buf = ops.presplit_buffer(d, rows, x2.device)

    def make():
        ops.presplit_weights([(x2.t(), buf)])

    if ops.SIDE is not None:
        ops.SIDE.submit(make, x2, buf)
    else:
        make()
    return buf


def backward_from_tape(dec, tape, dscores, grads, overlap=True, want_image_grad=False):
    """Accumulate parameter gradients of `dec` into `grads` (dict id(param) -> zero-initialised
    tensor shaped like the parameter; frozen parameters are simply absent).  With `overlap` the weight /
    bias gradients of the Linear layers run on a second HIP stream beside the data-gradient chain.
    want_image_grad: also return the gradient of the (length-sorted) image memory rows (B, P, d) -- what
    fine_tune_encoder=True back-propagates into Encoder.conv1; the default step skips that 13.5 GFLOP GEMM because
    the reference never uses the result (geo-aware/train.py:93-100,283-284)."""
    bp = BackwardPass(dec, tape, dscores, grads, overlap, want_image_grad)
    bp.early(join=False)     # one pass: the side stream is only joined at the very end
    bp.late()
    return tape.misc.get("d_img")


def early_parameters(dec):
    """Parameters whose gradients are complete after BackwardPass.early(): the score head and the decoder stack
    (54 % of the geo model's bytes).  The rest -- context encoders, embeddings -- completes in late()."""
    mods = [dec.fc_vocab, dec.fc_entity, dec.transformer_decoder]
    for name in ("fc_fact", "fc_predicate"):
        if getattr(dec, name, None) is not None:
            mods.append(getattr(dec, name))
    seen, out = set(), []
    for mod in mods:
        for prm in mod.parameters():
            if id(prm) not in seen:
                seen.add(id(prm))
                out.append(prm)
    return out


class BackwardPass:
    """The backward pass in two phases, so that a data-parallel step can start reducing the first half of the
    gradient bucket while the second half is still being computed (TrainStep with more than one rank):
    early() = score head + decoder stack, with the side stream joined at its end; late() = context encoders and
    embeddings.  The two phases may be captured into two hipGraphs."""

    def __init__(self, dec, tape, dscores, grads, overlap=True, want_image_grad=False):
        self.side = ops.SideStream() if overlap else None
        self.gen = _backward_phases(dec, tape, dscores, grads, want_image_grad)

    def _run(self, join):
        ops.SIDE = self.side
        try:
            next(self.gen, None)
        finally:
            ops.SIDE = None
            if self.side is not None and join:
                self.side.join()

    def early(self, join=True):
        self._run(join)
        self.joined = join

    def late(self):
        if self.side is not None and getattr(self, "joined", False):
            # the two phases may be captured into two graphs: each graph gets a side stream of its own (one stream
            # object forked into two captures made every kernel of both graphs run ~2.5x slower on ROCm 7.2)
            self.retired = self.side          # keeps the tensors the first phase's side work read alive
            self.side = ops.SideStream()
        self._run(True)


def _backward_phases(dec, tape, dscores, grads, want_image_grad=False):
    m = tape.misc
    d, V, H = dec.emb_dim, dec.vocab_size, dec.num_heads
    h, ee, fe = m["h"], m["ee"], m["fe"]
    # frozen parameters (other than the word embedding, whose scatter is simply skipped) get a scratch
    # buffer so every kernel has somewhere to accumulate; the scratch is dropped afterwards
    grads = dict(grads)
    for prm in unique_parameters(dec):
        if id(prm) not in grads and prm is not dec.word_embedding.weight:
            grads[id(prm)] = torch.zeros_like(prm)
    B, L, _ = h.shape
    P, K, Fn, S = m["P"], m["K"], m["Fn"], m["S"]
    M = B * L
    Vx = dscores.shape[2]
    dsc2 = dscores.view(M, Vx)
    dev = h.device
    pre = m.get("prezero") or {}

    def zeroed(name, like):
        t = pre.get(name)
        return t if t is not None and t.shape == like.shape else torch.zeros_like(like)

    dee = zeroed("dee", ee)
    dfe = zeroed("dfe", fe) if fe is not None else None
    # ---- score head
    hv = m["hv"] if dec.has_facts else h
    # the vocabulary weight gradient is a large problem of its own: it starts beside its data gradient
    dhv0 = pre.get("dhv")
    hv_t_ps = _rows_t_presplit(hv.view(M, d)) if _g(grads, dec.fc_vocab.weight) is not None else None
    if dhv0 is not None and dhv0.shape == (M, d) and not ops.is_deterministic():
        # split-K partial sums add into the buffer the forward pass's side stream zeroed
        dhv = _lin_bwd(grads, dsc2[:, :V], hv.view(M, d), dec.fc_vocab.weight, dec.fc_vocab.bias, dx=dhv0, acc=True,
                       group_now=True, wt_ps=m.get("vocab_t_ps"), xt_ps=hv_t_ps).view(B, L, d)
    else:
        dhv = _lin_bwd(grads, dsc2[:, :V], hv.view(M, d), dec.fc_vocab.weight, dec.fc_vocab.bias,
                       group_now=True, wt_ps=m.get("vocab_t_ps"), xt_ps=hv_t_ps).view(B, L, d)
    if dec.has_facts:
        dh = ops.mul(dhv, m["gate"])
        dgate = ops.mul(dhv, h)
        gw, gb = _g(grads, dec.fc_predicate.weight), _g(grads, dec.fc_predicate.bias)
        if gw is not None:
            ops.context_gate_bwd(m["captions"], m["facts"], dgate, gw, gb, K, V, mode=0)
    else:
        dh = dhv

    def gbuf(param):
        return _g(grads, param)

    ops.pointer_scores_bwd(dscores, V, h, ee, _p(dec.fc_entity.weight), None, dh, dee, gbuf(dec.fc_entity.weight),
                           gbuf(dec.fc_entity.bias))
    if dec.has_facts:
        ops.pointer_scores_bwd(dscores, V + K, h, fe, _p(dec.fc_fact.weight), m["eib"], dh, dfe,
                               gbuf(dec.fc_fact.weight), gbuf(dec.fc_fact.bias))
    # ---- decoder stack
    layers = list(dec.transformer_decoder.layers)
    nseg = 2 * len(layers)
    dkv_rows = ops.attention_bwd_buffer((B, S, nseg * d), L, S, d // H, dev)
    dx = dh
    ops.stamp("bwd: head done")
    pkb = m.get("pkb")
    mem_t_ps = _memory_t_presplit(m, B, S, d)
    if pkb is not None:
        state = (dh.view(M, d), None, None)
        for li in reversed(range(len(layers))):
            state = _decoder_layer_bwd_chain(dec, li, layers[li], tape.dec_layers[li], state, dkv_rows, m["kv"], S,
                                             grads, pkb, mem2=m["mem"].view(B * S, d), mem_t_ps=mem_t_ps)
            ops.stamp("bwd: decoder layer %d done" % li)
        dx = state[0]
    else:
        for li in reversed(range(len(layers))):
            dx = _decoder_layer_bwd(dec, li, layers[li], tape.dec_layers[li], dx, dkv_rows, m["kv"], S, grads,
                                    mem2=m["mem"].view(B * S, d), mem_t_ps=mem_t_ps)
            ops.stamp("bwd: decoder layer %d done" % li)
    yield    # ---- end of the early phase: every gradient of early_parameters() has been enqueued
    # ---- cross K/V projection: the weight gradients went out with the decoder layers; data gradient for the
    # context rows only (the image rows' gradient would be Encoder.conv1's, which the reference never uses)
    nctx = K + Fn
    dkv_ctx = dkv_rows[:, P:]                       # (B, nctx, 2 * layers * d) view of the K/V gradient rows
    fuse_ctx = pkb is not None and ops.rowchain_bwd_supported(nseg * d, d, 0)
    dctx = None
    if not fuse_ctx:
        # K = 2 * layers * d = 1800 over only B * nctx x d outputs: split the reduction (40 -> ~15 us with the fill)
        ksplit = max(1, min(8, (nseg * d) // 450)) if B * nctx * d <= 1280 * 512 else 1
        if ops.is_deterministic():
            ksplit = 1
        dctx = (torch.zeros if ksplit > 1 else torch.empty)(B, nctx, d, device=dev, dtype=torch.float32)
        ops.gemm_raw(dkv_ctx, m["wkv"], dctx, B * nctx, d, nseg * d, nseg * d, 1, 1, d, d, a_grp=nctx,
                     a_gs=S * nseg * d, atomic=ksplit > 1, split_k=ksplit)
    if want_image_grad:
        # image rows: d mem[:, :P] = dK/dV rows @ packed K/V weight (the data gradient of the all-layer projection)
        d_img = torch.empty(B, P, d, device=dev, dtype=torch.float32)
        ops.gemm_raw(dkv_rows, m["wkv"], d_img, B * P, d, nseg * d, nseg * d, 1, 1, d, d, a_grp=P, a_gs=S * nseg * d)
        m["d_img"] = d_img
    # ---- context encoders.  With the row chains the data gradient of the K/V projection for the context rows rides on
    # the first launch of each context encoder's backward (its rows are read straight out of the K/V gradient buffer)
    ops.stamp("bwd: context gradient ready")
    dee_enc = _context_encoder_bwd(dec, dec.transformer_encoder_entities, tape.enc_layers["entities"],
                                   None if fuse_ctx else dctx[:, :K].contiguous(), grads, pkb=pkb, tag="e",
                                   g_first=(dkv_ctx[:, :K], pkb[("kv", "T")]) if fuse_ctx else None)
    ops.stamp("bwd: entity context encoder done")
    dee += dee_enc
    if dec.has_facts:
        dfe_enc = _context_encoder_bwd(dec, dec.transformer_encoder_facts, tape.enc_layers["facts"],
                                       None if fuse_ctx else dctx[:, K:].contiguous(), grads, pkb=pkb, tag="f",
                                       g_first=(dkv_ctx[:, K:], pkb[("kv", "T")]) if fuse_ctx else None)
        dfe += dfe_enc
    # ---- caption embedding, fact encoder, entity encoder
    gword = _g(grads, dec.word_embedding.weight)
    ops.caption_embed_bwd(dx, m["captions"], m["masks"], gword, dee, dfe, V, dec.word_map["<pad>"], math.sqrt(d),
                          drop=m["d_pos"])
    if dec.has_facts:
        ops.fact_encode_bwd(dfe, m["facts"], dee, gbuf(dec.predicate_embedding.weight))
    ops.entity_encode_bwd(dec.variant, dee, m["entities"], ee, gbuf(dec.entity_encoder.type_embedding.weight),
                          word_emb=_p(dec.word_embedding.weight) if dec.variant == "news" else None,
                          dword=gword if dec.variant == "news" else None)


# ----------------------------------------------------------------------------------------------
# autograd bridge (drop-in for the reference's train.py)
# ----------------------------------------------------------------------------------------------
class DecoderGraphFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dec, captions, masks, entities, facts, enc_tok, gmap, *params):
        dec.__dict__["_drop_step"] = dec.__dict__.get("_drop_step", 0) + 1
        seed = (dec.__dict__.get("_drop_seed", 0x1234567) * 2654435761 + dec.__dict__["_drop_step"]) & 0xFFFFFFFF
        scores, tape = forward_with_tape(dec, captions, masks, entities, facts, enc_tok.detach(), gmap, seed=seed)
        ctx.dec, ctx.tape, ctx.params, ctx.gmap = dec, tape, params, gmap
        ctx.enc_shape = enc_tok.shape
        return scores

    @staticmethod
    def backward(ctx, dscores):
        dec, params = ctx.dec, ctx.params
        grads = {id(p): torch.zeros_like(p) for p in params if p.requires_grad}
        want_img = ctx.needs_input_grad[5]      # fine_tune_encoder=True: the loss reaches Encoder.conv1 through enc_tok
        d_img = backward_from_tape(dec, ctx.tape, dscores.contiguous(), grads, overlap=not ops.is_deterministic(),
                                   want_image_grad=want_img)
        d_enc = None
        if want_img:                            # the forward gathered the samples into length order through gmap
            d_enc = torch.empty(ctx.enc_shape, device=d_img.device, dtype=torch.float32)
            d_enc.index_copy_(0, ctx.gmap.long(), d_img) if ctx.gmap is not None else d_enc.copy_(d_img)
        return (None,) * 5 + (d_enc, None) + tuple(grads.get(id(p)) for p in params)


def unique_parameters(dec):
    seen, out = set(), []
    for p in dec.parameters():
        if id(p) not in seen:
            seen.add(id(p))
            out.append(p)
    return out


# ----------------------------------------------------------------------------------------------
# fused data-parallel training step
# ----------------------------------------------------------------------------------------------
class TrainStep:
    """forward -> packed CE -> backward -> all-reduce -> clamp + Adam over one flat fp32 bucket.

    Gradient semantics equal the single-process full-batch step of the reference: every rank
    contributes the SUM of its token losses' gradients plus its token count; after the
    all-reduce(sum) the bucket is divided by the global token count, clamped to +-grad_clip
    (geo-aware/train.py:287-288 clamps the full-batch gradient) and fed to Adam (lr 4e-4).

    With use_graph the device work is two captured hipGraphs around the (eager) collective:
      A  zero the bucket, forward with saved activations, packed cross entropy, backward -- the
         weight / bias gradients run on a second stream beside the data-gradient chain (fork/join
         edges of the graph);
      B  divide by the reduced token count, clamp, Adam, bump the step counter.
    The step counter lives on the device: it seeds the dropout masks and Adam's bias correction, so
    replays advance without re-capturing."""

    def __init__(self, decoder, lr=4e-4, grad_clip=5.0, betas=(0.9, 0.999), eps=1e-8, process_group=None, seed=0,
                 use_graph=True, encoder=None, deterministic=None, lazy_update=False):
        self.dec = decoder
        # lazy_update: the optimizer update of step i runs at the head of step i + 1's graph, on the side stream beside
        # Encoder.conv1 (see forward_with_tape's pre_side) instead of as a graph of its own at the end of step i, where
        # nothing overlaps its ~85 us of HBM streaming.  Between two calls the parameters then lag one update behind the
        # gradients: flush() applies the pending update (state_dict / load_state_dict / set_lr / as_torch_optimizer do it
        # themselves; call it before reading the decoder's parameters, validating or checkpointing).  Same arithmetic,
        # same order of optimizer steps; needs the captured two-stream step with the feature-map input (encoder=) and the
        # optimizer-maintained weight images, else the update stays where it was.
        self.lazy = bool(lazy_update)
        self._pending = False
        self._gb = None
        # deterministic (default: ICK_DETERMINISTIC=1 in the environment): the library's fixed-order reductions, no
        # split GEMMs, everything on one stream -- two runs of the same steps end bit-identical (cost: DESIGN.md)
        # None inherits the library's current mode (ick_get_deterministic: ICK_DETERMINISTIC in the environment unless
        # ops.set_deterministic was called); only an explicit True / False changes the process-wide flag, and a step
        # refuses to run when the flag no longer is what its graphs were captured under.
        if deterministic is None:
            deterministic = ops.is_deterministic()
        else:
            ops.set_deterministic(bool(deterministic))
        self.deterministic = bool(deterministic)
        # with an encoder the step also accepts the (B, 2048, 14, 14) feature map: Encoder.conv1 (frozen, as in the
        # reference's default fine_tune_encoder=False) then runs inside graph A straight into the memory buffer
        self.enc = encoder
        self.seed = seed  # dropout mask stream; give every rank its own seed
        self.lr, self.clip, self.betas, self.eps = lr, grad_clip, betas, eps
        self.pg = process_group
        self.use_graph = use_graph
        self.step_count = 0
        params = [p for p in unique_parameters(decoder) if p.requires_grad]
        # bucket order: the parameters whose gradients are complete first come first, so that with several ranks
        # their part of the bucket can be all-reduced while the rest of the backward pass still runs
        early_ids = {id(p) for p in early_parameters(decoder)}
        params = [p for p in params if id(p) in early_ids] + [p for p in params if id(p) not in early_ids]
        dev = params[0].device
        # every parameter starts at a multiple of 64 floats (256 bytes): the GEMM's 16-byte vector loads need aligned
        # weight rows, and one odd-sized parameter (fc_entity.bias has a single element) would misalign all that follow
        pad = lambda k: (k + 63) // 64 * 64     # noqa: E731
        n = sum(pad(p.numel()) for p in params)
        self.n = n
        self.n_early = sum(pad(p.numel()) for p in params if id(p) in early_ids)
        # +2 trailing floats travel with the gradient bucket: [sum of token losses, token count]
        self.flat_p = torch.zeros(n, device=dev, dtype=torch.float32)
        self.flat_g = torch.zeros(n + 2, device=dev, dtype=torch.float32)
        self.flat_m = torch.zeros(n, device=dev, dtype=torch.float32)
        self.flat_v = torch.zeros(n, device=dev, dtype=torch.float32)
        self.counter = torch.zeros(1, device=dev, dtype=torch.int32)   # steps done (read as uint32 by the kernels)
        self.one = torch.ones(1, device=dev, dtype=torch.float32)
        self.grads = {}
        off = 0
        with torch.no_grad():
            for p in params:
                k = p.numel()
                self.flat_p[off:off + k].copy_(p.reshape(-1))
                p.data = self.flat_p[off:off + k].view(p.shape)      # parameters become views of the bucket
                self.grads[id(p)] = self.flat_g[off:off + k].view(p.shape)
                off += pad(k)
        self.params = params
        self._graphs = {}
        # the re-laid-out copies of the weights (packed row-chain images, cross K/V gather, bf16 planes) that the optimizer
        # kernel keeps current itself: built on the first call (DerivedWeights); ICK_ADAM_DERIVE=0 keeps the per-step
        # packing launches of rounds 2-4 instead (A/B runs, and the fallback for layer widths the images do not cover)
        self.derived = None
        self._derived_tried = False
        # several ranks: every replica starts from rank 0's weights (a freshly built decoder is randomly initialised
        # per process; the reference has a single process, geo-aware/train.py:16-18).  One broadcast of the bucket.
        dp.broadcast_bucket(self.flat_p, self.pg)
        # Several ranks, ICK_SPLIT_ALLREDUCE=1: the step runs as two graphs around two all-reduces -- the early half of
        # the bucket (score head, decoder stack: gradients complete first) travels while the late half (cross K/V
        # projection, context encoders, embeddings) is computed, and only the late half's all-reduce is exposed.  Measured
        # on one GPU the split itself costs ~130 us (a join in the middle of the backward pass, one more graph launch),
        # and nothing has been measured on a multi-GPU node to win that back, so the DEFAULT IS OFF (one all-reduce
        # between graph A and graph B) until a scaling run shows the gain; same bits either way
        # (tests/test_deterministic_gpu.py).  ICK_SPLIT_ALLREDUCE=auto decides from a timed all-reduce of the bucket at
        # construction (split when the collective costs more than twice what the split does; every rank takes the
        # max-reduced time, so every rank decides alike).  allreduce_probe_ms keeps that measurement for bench.py.
        env = os.environ.get("ICK_SPLIT_ALLREDUCE", "0")
        self.allreduce_probe_ms = None
        if dp.world_size(self.pg) > 1 and (env == "auto" or os.environ.get("ICK_ALLREDUCE_PROBE") == "1"):
            self.allreduce_probe_ms = dp.time_all_reduce(self.flat_g, self.pg)
        if env == "auto":
            self.split = self.allreduce_probe_ms is not None and self.allreduce_probe_ms > 0.26
        else:
            self.split = env not in ("", "0")       # an explicit 1 also splits a single rank's step (tests, A/B runs)

    # ---- device-only halves -------------------------------------------------------------------
    def _overlap(self, off_switch):
        """Second-stream work (context chains beside the image projection, weight gradients beside the data-gradient
        chain): in captured steps, unless switched off -- and never in deterministic mode, where every buffer must
        receive its terms in one fixed order."""
        if self.deterministic:
            return False
        return bool((self.use_graph or os.environ.get("ICK_GROUP_SAME_STREAM")) and not os.environ.get(off_switch))

    def _enc_kwargs(self, enc_in):
        if enc_in.dim() == 4:
            c1 = self.enc.conv1
            return dict(enc_tok=None, feats=enc_in, conv1=(c1.weight.detach(), c1.bias.detach(), self.enc.conv1_presplit()))
        return dict(enc_tok=enc_in)

    def _part_a(self, captions, caption_masks, entities, facts, enc_in, gmap, lengths):
        dec = self.dec
        ops.stamp("A: start")
        box = {}

        def tail():      # nothing in the forward pass waits for these: they run on the side stream behind the context chain
            self.flat_g.zero_()
            box["decode_len"] = (lengths.reshape(-1) - 1).to(torch.int32)
            box["prezero"] = _prezeroed(dec, captions, entities, facts)

        scores, tape = forward_with_tape(dec, captions, caption_masks, entities, facts, gmap=gmap,
                                         seed=self.seed * 2654435761 & 0xFFFFFFFF, epoch=self.counter, fresh_pack=True,
                                         overlap=self._overlap("ICK_NO_FWD_OVERLAP"),
                                         side_tail=tail, derived=self.derived,
                                         pre_side=self._deferred_update if self._lazy_active(enc_in) else None,
                                         **self._enc_kwargs(enc_in))
        decode_len = box["decode_len"]
        tape.misc["prezero"] = box.get("prezero")
        ops.stamp("fwd: scores done")
        # the two scalars of the loss go straight into the tail of the gradient bucket (it was zeroed above; nothing else
        # touches those two floats)
        _, _, dscores = ops.packed_ce(scores, captions, decode_len, dec.word_map["<pad>"], want_grad=True,
                                      out_sum=self.flat_g[self.n:self.n + 1], out_count=self.flat_g[self.n + 1:])
        ops.stamp("CE done")
        backward_from_tape(dec, tape, dscores, self.grads, overlap=self._overlap("ICK_NO_BWD_OVERLAP"))
        ops.stamp("A: end (after join)")
        return self.flat_g

    def _lazy_active(self, enc_in):
        return bool(self.lazy and self.use_graph and self.derived is not None and not self.split and not self.deterministic
                    and enc_in is not None and enc_in.dim() == 4 and self._overlap("ICK_NO_FWD_OVERLAP"))

    def _deferred_update(self):
        """The previous step's clamp + Adam and step counter, at the head of this step's graph.  Both are no-ops while the
        token count in the gradient bucket's tail is zero: the very first step, and the step after a flush()."""
        self._adam()
        ops.counter_add_if(self.counter, 1, self.flat_g[self.n + 1:])

    def flush(self):
        """Apply a pending (lazy_update) optimizer update now; afterwards the parameters, moments, step counter and weight
        images are what the eager order would have left."""
        if not self._pending:
            return
        if self._gb is not None:
            self._gb.replay()
        else:
            self._part_b()
        self.flat_g[self.n + 1:].zero_()      # the next step's leading update finds nothing to apply
        self._pending = False
        self.dec.invalidate_caches()
        if self.derived is not None:
            self.derived.mark_current()

    def _adam(self):
        # divide by the global token count (device-resident), clamp, Adam with the device step counter -- and, with
        # self.derived, the re-laid-out copies of the updated weights in the same pass (ick_adam_clamp_derive)
        hyper = (1, self.lr, self.clip, 1.0, self.betas[0], self.betas[1], self.eps)
        if self.derived is not None:
            ops.adam_clamp_derive(self.flat_p, self.flat_g, self.flat_m, self.flat_v, self.derived.items_dev,
                                  self.derived.blocks_dev, self.derived.n_blocks, *hyper, step_tensor=self.counter,
                                  gscale_den=self.flat_g[self.n + 1:], nbytes=self.derived.nbytes)
        else:
            ops.adam_clamp(self.flat_p[:self.n], self.flat_g[:self.n], self.flat_m, self.flat_v, *hyper,
                           step_tensor=self.counter, gscale_den=self.flat_g[self.n + 1:])

    # ---- the same step in two halves (several ranks: the early half's all-reduce overlaps the late half) ----
    def _part_a1(self, captions, caption_masks, entities, facts, enc_in, gmap, lengths):
        dec = self.dec
        box = {}

        def tail():
            self.flat_g.zero_()
            box["decode_len"] = (lengths.reshape(-1) - 1).to(torch.int32)
            box["prezero"] = _prezeroed(dec, captions, entities, facts)

        scores, tape = forward_with_tape(dec, captions, caption_masks, entities, facts, gmap=gmap,
                                         seed=self.seed * 2654435761 & 0xFFFFFFFF, epoch=self.counter, fresh_pack=True,
                                         overlap=self._overlap("ICK_NO_FWD_OVERLAP"),
                                         side_tail=tail, derived=self.derived, **self._enc_kwargs(enc_in))
        decode_len = box["decode_len"]
        tape.misc["prezero"] = box.get("prezero")
        self._loss = ops.packed_ce(scores, captions, decode_len, dec.word_map["<pad>"], want_grad=True,
                                   out_sum=self.flat_g[self.n:self.n + 1], out_count=self.flat_g[self.n + 1:])
        self._bp = BackwardPass(dec, tape, self._loss[2], self.grads,
                                overlap=self._overlap("ICK_NO_BWD_OVERLAP"))
        self._bp.early(join=True)
        return self.flat_g

    def _part_a2(self):
        self._bp.late()
        return self.flat_g

    def _part_b(self):
        # divide by the global token count (device-resident), clamp, Adam with the device step counter
        ops.stamp("B: start")
        self._adam()
        ops.counter_add(self.counter, 1)
        ops.stamp("B: end")
        return self.flat_g

    def _capture(self, fn, inputs):
        static = [None if t is None else t.contiguous().clone() for t in inputs]
        # warm-up off the capture (lazy kernel attributes); the warm-up steps touch the gradient bucket only
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn(*static)
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with ops.capture(g):
            fn(*static)
        return g, static

    def _capture_split(self, inputs):
        """Graphs A1 (forward, CE, early backward) and A2 (late backward) over one memory pool: A2 reads what A1 left
        in the tape."""
        static = [None if t is None else t.contiguous().clone() for t in inputs]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self._part_a1(*static)
            self._part_a2()
        torch.cuda.current_stream().wait_stream(side)
        g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with ops.capture(g1):
            self._part_a1(*static)
        with ops.capture(g2):   # own pool: the tape keeps A1's tensors alive
            self._part_a2()
        return g1, static, g2

    # ---- optimizer state in torch.optim.Adam's layout ------------------------------------------
    def _adam_order(self):
        """Trainable parameters in the order the reference hands them to torch.optim.Adam
        (geo-aware/train.py:85-88: filter(requires_grad, decoder.parameters()))."""
        mine = {id(p) for p in self.params}
        return [p for p in unique_parameters(self.dec) if id(p) in mine]

    def _slot(self, flat, p):
        off = (p.data_ptr() - self.flat_p.data_ptr()) // 4
        return flat[off:off + p.numel()].view(p.shape)

    def state_dict(self):
        """The optimizer state as torch.optim.Adam.state_dict() would report it (exp_avg / exp_avg_sq / step per
        parameter, one param group), so checkpoints interchange with the reference's `decoder_optimizer`
        (geo-aware/utils.py:32-46) and with the fused=False path.  Tensors are copies."""
        self.flush()
        self._check_views()
        order = self._adam_order()
        step = float(int(self.counter.item()) & 0xFFFFFFFF)
        state = {}
        if step > 0:
            for i, p in enumerate(order):
                state[i] = {"step": torch.tensor(step), "exp_avg": self._slot(self.flat_m, p).clone(),
                            "exp_avg_sq": self._slot(self.flat_v, p).clone()}
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": 0, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "decoupled_weight_decay": False, "params": list(range(len(order)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        """Restore Adam moments, step count (which also positions the dropout stream) and learning rate from a
        torch.optim.Adam-layout state dict: ours, the reference's, or the fused=False path's."""
        self.flush()
        self._check_views()
        order = self._adam_order()
        groups = sd["param_groups"]
        ids = [i for g in groups for i in g["params"]]
        if len(ids) != len(order):
            raise IckError("optimizer state holds %d parameters, the decoder has %d trainable ones" %
                           (len(ids), len(order)))
        steps = set()
        with torch.no_grad():
            self.flat_m.zero_()
            self.flat_v.zero_()
            for pos, i in enumerate(ids):
                st = sd["state"].get(i)
                if not st:
                    continue
                p = order[pos]
                if tuple(st["exp_avg"].shape) != tuple(p.shape):
                    raise IckError("optimizer state %d has shape %s, parameter has %s" %
                                   (i, tuple(st["exp_avg"].shape), tuple(p.shape)))
                self._slot(self.flat_m, p).copy_(st["exp_avg"])
                self._slot(self.flat_v, p).copy_(st["exp_avg_sq"])
                steps.add(int(float(st["step"])))
            if len(steps) > 1:
                raise IckError("per-parameter Adam step counts differ (%s): not representable in the fused step" % steps)
            self.counter.fill_(steps.pop() if steps else 0)
        self.lr = float(groups[0]["lr"])
        self.betas = tuple(groups[0].get("betas", self.betas))
        self.eps = float(groups[0].get("eps", self.eps))
        self._graphs.clear()          # lr / betas / eps are baked into the captured optimizer graph
        self.dec.invalidate_caches()
        self.parameters_changed()

    def parameters_changed(self):
        """Tell the step that parameter VALUES were written from outside it (through .data, a raw kernel, a restored
        checkpoint): the re-laid-out copies the optimizer kernel maintains are rebuilt before the next step.  In-place
        torch operations on the parameters are noticed without this call (their version counters move)."""
        self.flush()
        if self.derived is not None:
            self.derived.stale = True

    def as_torch_optimizer(self):
        """A live torch.optim.Adam over the decoder's parameters carrying this step's state: what goes into the
        checkpoint under 'decoder_optimizer', exactly the kind of object the reference pickles there."""
        opt = torch.optim.Adam(self._adam_order(), lr=self.lr, betas=tuple(self.betas), eps=self.eps)
        opt.load_state_dict(self.state_dict())
        return opt

    def input_buffers(self):
        """The static input tensors of the most recently captured graphs, in __call__'s argument order
        (captions, encoder_out-or-features, caption_masks, caption_lengths, entities[, facts]).  A loader that writes
        the next batch straight into them (pinned host-to-device copies) and calls the step with these very tensors
        skips the per-step device-to-device copy of the inputs."""
        st = getattr(self, "_last_static", None)
        if st is None:
            raise IckError("no captured graph yet: run one step first")
        captions, masks, entities, facts, enc_in, _, lengths = st
        return (captions, enc_in, masks, lengths, entities) + ((facts,) if facts is not None else ())

    def set_lr(self, lr):
        self.flush()                  # a pending update was computed under the old rate
        self.lr = float(lr)
        self._graphs.clear()          # baked into the captured optimizer graph

    def _check_views(self):
        """The decoder's parameters must still be views of the flat bucket: decoder.to() / .cuda() /
        load_pretrained_embeddings() re-allocate them and the step would then update memory the module no longer
        reads (silently).  Two pointer compares per call."""
        first, last = self.params[0], self.params[-1]
        lo, hi = self.flat_p.data_ptr(), self.flat_p.data_ptr() + 4 * self.n
        if first.data_ptr() != lo or not (lo <= last.data_ptr() and last.data_ptr() + 4 * last.numel() <= hi):
            raise IckError("the decoder's parameters no longer live in TrainStep's bucket (module moved or a "
                           "Parameter was replaced after TrainStep was built): build a new TrainStep")

    def __call__(self, captions, encoder_out, caption_masks, caption_lengths, entities, facts=None):
        dec = self.dec
        self._check_views()
        encoder_out, entities, facts = dec._prepare_inputs(encoder_out, entities, facts)
        dev = encoder_out.device
        # No length sort and no host round trip here: the loss is a sum over tokens, so the batch order does
        # not matter, and the lengths are only read on the device (packed cross entropy).  The reference sorts
        # because pack_padded_sequence wants it (geo-aware/models.py:330-336); forward() keeps doing so.
        captions = captions.to(dev, non_blocking=True)
        caption_masks = caption_masks.to(dev, non_blocking=True)
        if encoder_out.dim() == 4:
            if self.enc is None:
                raise IckError("TrainStep got a (B, C, H, W) feature map but was built without encoder=")
            enc_in = encoder_out
            self.enc.conv1_presplit()      # refreshed in place, OUTSIDE the captured graphs, if conv1's weight changed
        else:
            enc_in = dec._token_major(encoder_out)
        lengths = caption_lengths.to(dev, non_blocking=True)
        inputs = [captions, caption_masks, entities, facts, enc_in, None, lengths]
        if ops.is_deterministic() != self.deterministic:
            raise IckError("the library's deterministic mode is %s but this TrainStep was built with %s: captured graphs "
                           "keep the mode they were captured in (call ops.set_deterministic before building the step)"
                           % (ops.is_deterministic(), self.deterministic))
        # the large GEMM tiles' product mode is baked into a capture as well
        key = tuple(None if t is None else tuple(t.shape) for t in inputs) + (ops.gemm_split_mode(),)
        if not self._derived_tried:
            self._derived_tried = True
            if os.environ.get("ICK_ADAM_DERIVE", "1") != "0":
                self.derived = DerivedWeights.build(self)
                self._graphs.clear()
        if self.derived is not None:
            self.derived.ensure_current()
        if self.use_graph and key not in self._graphs:
            self.flush()                  # the warm-up runs below must not find gradients waiting to be applied
            if len(self._graphs) >= 4:
                self._graphs.clear()
            # the eager warm-up run of part B is a real optimizer step: snapshot and rewind its state
            state = (self.flat_p, self.flat_m, self.flat_v, self.counter)
            snap = [t.clone() for t in state]
            try:
                if self.split:
                    ga, static, ga2 = self._capture_split(inputs)
                else:
                    ga, static = self._capture(self._part_a, inputs)
                    ga2 = None
                gb, _ = self._capture(self._part_b, [])
                self._gb = gb
                self._graphs[key] = (ga, static, gb, ga2)
                self._last_static = static
            except RuntimeError as e:   # capture refused (driver / collective library state): run eagerly
                import warnings
                warnings.warn("ick_amd TrainStep: hipGraph capture failed (%s); continuing without graphs" % e)
                torch.cuda.synchronize()
                self.use_graph = False
            for t, sv in zip(state, snap):
                t.copy_(sv)
            if self.derived is not None:      # the warm-up's optimizer step wrote the images of weights that were just rewound
                self.derived.stale = True
                self.derived.ensure_current()
            self.flat_g[self.n + 1:].zero_()  # (lazy_update: the warm-up's gradients are not a pending update)
        if self.use_graph:
            ga, static, gb, ga2 = self._graphs[key]
            if self._lazy_active(enc_in) and not self._pending:
                # nothing is waiting to be applied (first step, or the update was flushed / applied eagerly): the leading
                # update of the graph must find a zero token count
                self.flat_g[self.n + 1:].zero_()
            from .decoder import copy_inputs
            copy_inputs(static, inputs)
            ga.replay()
            if ga2 is None:
                dp.allreduce_bucket(self.flat_g, self.pg)
            else:
                # [early gradients] travel while graph A2 computes [late gradients | loss_sum, count]
                w1 = dp.allreduce_bucket(self.flat_g[:self.n_early], self.pg, async_op=True)
                ga2.replay()
                w2 = dp.allreduce_bucket(self.flat_g[self.n_early:], self.pg, async_op=True)
                for w in (w1, w2):
                    if w is not None:
                        w.wait()
            if self._lazy_active(enc_in):
                self._pending = True      # applied at the head of the next step's graph (or by flush())
            else:
                gb.replay()
        elif self.split:
            self._part_a1(*inputs)
            w1 = dp.allreduce_bucket(self.flat_g[:self.n_early], self.pg, async_op=True)
            self._part_a2()
            w2 = dp.allreduce_bucket(self.flat_g[self.n_early:], self.pg, async_op=True)
            for w in (w1, w2):
                if w is not None:
                    w.wait()
            self._part_b()
        else:
            self._part_a(*inputs)
            dp.allreduce_bucket(self.flat_g, self.pg)
            self._part_b()
        self.step_count += 1
        dec.invalidate_caches()   # the update went around torch's version counters
        if self.derived is not None:
            self.derived.mark_current()
        # token-mean loss of the global batch, still on the device
        return self.flat_g[self.n:self.n + 1] / self.flat_g[self.n + 1:]
