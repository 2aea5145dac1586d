"""Encoder / DecoderTransformer engine shared by the three drop-in `models` modules.

Mirrors the reference's operator interface for the hot path:
    models.Encoder(encoded_image_size=14, emb_dim=300, encoder_dim=2048)         geo-aware/models.py:9-60
    models.DecoderTransformer(word_map, emb_dim, decoder_dim, encoder_dim, num_heads, num_layers,
                              dropout_dec=0.5, dropout_enc=0.5, dropout_pos=0.1)  geo-aware/models.py:212-254
    decoder(captions, encoder_out, caption_masks, caption_lengths, entities[, facts])
        -> (scores (B,L,V+K[+F]) in length-sorted order, captions_sorted, decode_lengths)   :315-361
    decoder.predict(encoder_out, max_pred_len, entities[, facts]) -> LongTensor (max_pred_len, B) :363-443
The module tree (sub-module and parameter names) is the reference's, so its whole-object
checkpoints (geo-aware/utils.py:32-49) unpickle into these classes and state_dicts interchange.
torch.nn modules are used ONLY as parameter containers; every forward computation below runs
in libick_amd.so (HIP, gfx950) through ops.py.  There is no CPU / PyTorch fallback.
"""
import math
import os

import torch
from torch import nn

from . import ops
from .lib import IckError

VARIANT_TYPE_OFFSET = {"geo": 4, "knowledge": 6, "news": 5}
VARIANT_NUM_TYPES = {"geo": 1000, "knowledge": 1000, "news": 20}
VARIANT_NUM_PREDICATES = {"geo": 0, "knowledge": 3000, "news": 3500}


def _sinusoid_table(max_len, d):
    """PositionEncoder buffer (geo-aware/models.py:199-205), shape (max_len, 1, d)."""
    pos = torch.arange(max_len, dtype=torch.float32).unsqueeze(1)
    freq = torch.exp(torch.arange(0, d, 2, dtype=torch.float32) * (-math.log(10000.0) / d))
    table = torch.zeros(max_len, d)
    table[:, 0::2] = torch.sin(pos * freq)
    table[:, 1::2] = torch.cos(pos * freq)
    return table.unsqueeze(1)


class _GraphedCall:
    """hipGraph capture of a pure-device function of static-shaped tensors (torch.cuda.CUDAGraph is the
    HIP graph API on ROCm).  The kernels are launched through ctypes on torch's current stream, which is the
    capture stream inside torch.cuda.graph(), so the whole launch sequence -- ~75 kernels for a teacher-forced
    forward, ~900 for a 20-step greedy decode -- replays from one hipGraphLaunch with no host work in between."""

    def __init__(self, fn, example_inputs):
        self.static_in = [None if t is None else t.contiguous().clone() for t in example_inputs]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):            # warm-up off the capture: lazy hipFuncSetAttribute, caches
            for _ in range(2):
                fn(*self.static_in)
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with ops.capture(self.graph):
            self.static_out = fn(*self.static_in)

    def __call__(self, *inputs):
        copy_inputs(self.static_in, inputs)
        self.graph.replay()
        return self.static_out


def copy_inputs(static, inputs):
    """Refresh a graph's input buffers: one foreach call instead of a Python-level copy_ per tensor (the host is
    on the critical path here: forward() has just synchronised for the length sort)."""
    # an input that IS its static buffer (the caller filled it in place) needs no copy
    pairs = [(d, s) for d, s in zip(static, inputs) if d is not None and not (
        s.is_cuda and s.data_ptr() == d.data_ptr() and s.shape == d.shape and s.dtype == d.dtype)]
    if not pairs:
        return
    dst = [d for d, _ in pairs]
    src = [s for _, s in pairs]
    if all(s.is_cuda and s.shape == d.shape and s.dtype == d.dtype and s.is_contiguous() for d, s in zip(dst, src)):
        ops.copy_batch(dst, src)          # one launch for all of them
    elif all(s.is_cuda and s.shape == d.shape for d, s in zip(dst, src)):
        torch._foreach_copy_(dst, src, non_blocking=True)
    else:
        for d, s in zip(dst, src):
            d.copy_(s, non_blocking=True)


class PositionEncoder(nn.Module):
    """Holds the sinusoid buffer `pe` and the dropout rate; applied inside ick_caption_embed."""

    def __init__(self, emb_dim, dropout, max_len=5000):
        super().__init__()
        self.dropout = nn.Dropout(p=dropout)
        self.register_buffer("pe", _sinusoid_table(max_len, emb_dim))


class EntityEncoder(nn.Module):
    """Parameter holder for the entity-type embedding (computation: ick_entity_encode)."""

    def __init__(self, emb_dim, type_embedding):
        super().__init__()
        self.emb_dim = emb_dim
        self.type_embedding = type_embedding


class FactEncoder(nn.Module):
    """Parameter holder for the predicate embedding (computation: ick_fact_encode)."""

    def __init__(self, emb_dim, predicate_embedding):
        super().__init__()
        self.emb_dim = emb_dim
        self.predicate_embedding = predicate_embedding


class CaptionEmbedder(nn.Module):
    def __init__(self, vocab_size):
        super().__init__()
        self.vocab_size = vocab_size


# Device-side caches a module keeps in its __dict__ (captured graphs, packed / pre-split weight copies, pinned staging):
# never part of a pickle (checkpoints pickle whole modules, geo-aware/utils.py:32-46) or of a deep copy -- they are rebuilt
# on first use.
_CACHE_KEYS = ("_graphs", "_kv_pack", "_pred_wt_cache", "_len_pin", "_idx_pin", "_plist", "_pin_ev", "_dec_pack",
               "_chain_cache", "_chain_cache_bwd", "_chain_ok", "_chain_bwd_ok", "_ps_cache", "_last_static", "_enc")


def _state_without_caches(module):
    return {k: v for k, v in module.__dict__.items() if k not in _CACHE_KEYS}


class _Conv1Fn(torch.autograd.Function):
    """Encoder.conv1 (1x1 convolution = GEMM over the NCHW map) with its backward on the same HIP GEMM:
    dW = dY^T . X, db = column sums of dY (riding on that GEMM), dX = dY . W (only when the trunk is being
    fine-tuned).  Used when gradients are wanted (fine_tune_encoder=True, geo-aware/train.py:93-100,282-294)."""

    @staticmethod
    def forward(ctx, feats, weight, bias, w_ps=None):
        B, Cc, Hh, Ww = feats.shape
        P, d = Hh * Ww, weight.shape[0]
        out = torch.empty(B, P, d, device=feats.device, dtype=torch.float32)
        ops.gemm_raw(feats, weight.detach().view(d, Cc), out, B * P, d, Cc, 1, P, Cc, 1, d, bias=bias, a_grp=P, a_gs=Cc * P,
                     b_ps=w_ps)
        ctx.save_for_backward(feats, weight)
        return out

    @staticmethod
    def backward(ctx, dout):
        feats, weight = ctx.saved_tensors
        B, Cc, Hh, Ww = feats.shape
        P, d = Hh * Ww, weight.shape[0]
        dy = dout.contiguous().view(B * P, d)
        dw = db = dx = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            # token-major copy of the map: the reduction index (sample, position) of dW needs one uniform stride
            xt = feats.view(B, Cc, P).permute(0, 2, 1).contiguous().view(B * P, Cc)
            dw = torch.zeros(d, Cc, device=feats.device, dtype=torch.float32)
            db = torch.zeros(d, device=feats.device, dtype=torch.float32)
            ops.gemm_raw(dy, xt, dw, d, Cc, B * P, 1, d, 1, Cc, Cc, atomic=True, split_k=ops.wgrad_split(B * P, d, Cc),
                         colsum_a=db)
            dw = dw.view_as(weight)
        if ctx.needs_input_grad[0]:
            dxt = torch.empty(B * P, Cc, device=feats.device, dtype=torch.float32)
            w2 = weight.view(d, Cc)
            ops.gemm_raw(dy, w2, dxt, B * P, Cc, d, d, 1, 1, Cc, Cc)           # dX = dY . W  (B k-major)
            dx = dxt.view(B, P, Cc).permute(0, 2, 1).reshape(B, Cc, Hh, Ww)
        return dx, dw, db, None


class Encoder(nn.Module):
    """Image encoder (geo-aware/models.py:9-60): [ResNet-101 trunk -> AdaptiveAvgPool2d(14)] -> conv1 (1x1,
    2048 -> emb_dim) -> view (B, emb_dim, 196).

    The hot path starts at the 14x14x2048 feature map (BASELINE configs use precomputed ResNet-101 features):
    forward() takes either that map (B, 2048, H, W) -- conv1 runs as a HIP GEMM straight off the NCHW layout -- or
    raw images (B, 3, H, W), which first go through the trunk (stock torch.nn / MIOpen, resnet.py; built on first
    use or with with_trunk=True; there is no network here, so pretrained weights come from load_state_dict /
    resnet.load_torchvision_state_dict).  Returns (B, emb_dim, 196) like the reference; the storage is token-major
    (B, 196, emb_dim) -- the layout the decoder's cross-attention K/V projection streams -- exposed through a
    permuted view, so the decoder consumes it without a copy."""

    def __init__(self, encoded_image_size=14, emb_dim=300, encoder_dim=2048, with_trunk=None):
        super().__init__()
        self.emb_dim = emb_dim
        self.encoder_dim = encoder_dim
        self.encoded_image_size = encoded_image_size
        self.with_trunk = with_trunk
        if with_trunk:
            self._build_trunk()
        self.conv1 = nn.Conv2d(encoder_dim, emb_dim, 1)
        self.fine_tune()

    def _build_trunk(self):
        from .resnet import resnet101_trunk
        dev = self.conv1.weight.device if "conv1" in self._modules else None
        self.resnet = resnet101_trunk()
        self.adaptive_pool = nn.AdaptiveAvgPool2d((self.encoded_image_size, self.encoded_image_size))
        if dev is not None:
            self.resnet.to(dev)
        self.resnet.train(self.training)
        # a trunk built on first use takes the fine_tune() setting that was asked for before it existed.  Its
        # parameters are new, so an optimizer built earlier does not hold them: train.py builds the trunk up front
        # (with_trunk=True) whenever fine_tune_encoder is set.
        self.fine_tune(self.__dict__.get("_fine_tune", True))

    def features(self, images):
        """images (B, 3, H, W) -> (B, 2048, 14, 14): trunk + adaptive pooling (geo-aware/models.py:42-43)."""
        if "resnet" not in self._modules:
            if self.with_trunk is False:
                raise IckError("this Encoder was built with with_trunk=False: pass the (B, %d, H, W) feature map"
                               % self.encoder_dim)
            self._build_trunk()
        return self.adaptive_pool(self.resnet(images))

    def forward(self, feats):
        if feats.dim() == 4 and feats.shape[1] == 3 and self.encoder_dim != 3:
            feats = self.features(feats)
        if feats.dim() != 4 or feats.shape[1] != self.encoder_dim:
            raise IckError("Encoder.forward expects images (B, 3, H, W) or the (B, %d, H, W) feature map"
                           % self.encoder_dim)
        if feats.dtype != torch.float32:          # a float16 feature file: the kernels read float32
            feats = feats.float()
        feats = feats.contiguous()
        B, Cc, Hh, Ww = feats.shape
        P = Hh * Ww
        d = self.emb_dim
        if torch.is_grad_enabled() and (feats.requires_grad or self.conv1.weight.requires_grad):
            return _Conv1Fn.apply(feats, self.conv1.weight, self.conv1.bias, self.conv1_presplit()).permute(0, 2, 1)
        out = torch.empty(B, P, d, device=feats.device, dtype=torch.float32)
        w = self.conv1.weight.detach().view(d, Cc)
        ops.gemm_raw(feats, w, out, B * P, d, Cc, 1, P, Cc, 1, d, bias=self.conv1.bias.detach(),
                     a_grp=P, a_gs=Cc * P, b_ps=self.conv1_presplit())
        return out.permute(0, 2, 1)

    def __getstate__(self):
        return _state_without_caches(self)

    def conv1_presplit(self):
        """The pre-split copy of conv1's (emb_dim, encoder_dim) weight for ick_gemm's b_ps (refreshed in place when the
        weight changes; None in the exact-fp32 product mode)."""
        w = self.conv1.weight
        return ops.presplit_cached(self, "conv1", w.detach().view(w.shape[0], -1),
                                   (w._version, w.data_ptr(), self.__dict__.get("_param_epoch", 0)))

    def invalidate_caches(self):
        """Call after conv1's weight was written behind torch's version counters (through .data, a fused optimizer): the
        pre-split copy is rebuilt on the next use and graphs that captured this encoder are re-captured."""
        self.__dict__["_param_epoch"] = self.__dict__.get("_param_epoch", 0) + 1

    def fine_tune(self, fine_tune=True):
        """Convolutional blocks 2-4 of the trunk train iff fine_tune (geo-aware/models.py:49-60); conv1 is left
        as the reference leaves it (always requires_grad).  Without a trunk the setting is remembered and applied when
        the trunk is built."""
        self.__dict__["_fine_tune"] = bool(fine_tune)
        if "resnet" not in self._modules:
            return
        for p in self.resnet.parameters():
            p.requires_grad = False
        for c in list(self.resnet.children())[5:]:
            for p in c.parameters():
                p.requires_grad = fine_tune


class DecoderTransformer(nn.Module):
    variant = "geo"
    use_hip_graphs = True   # inference forward / predict replay a captured hipGraph per input shape
    fused_decode = True     # predict(): fused per-block decode kernels (csrc/decode.hip) when the sizes allow
    fuse_select = True      # ... with the greedy selection folded into the next step's first launch

    def __init__(self, word_map, emb_dim, decoder_dim, encoder_dim, num_heads, num_layers, dropout_dec=0.5,
                 dropout_enc=0.5, dropout_pos=0.1):
        super().__init__()
        v = self.variant
        self.word_map = word_map
        self.vocab_size = len(word_map)
        self.emb_dim = emb_dim
        self.softmax = nn.Softmax(dim=-1)
        self.lookahead_mask = None
        self.pos_encoder = PositionEncoder(emb_dim, dropout_pos)
        self.transformer_decoder = nn.TransformerDecoder(
            nn.TransformerDecoderLayer(emb_dim, num_heads, decoder_dim, dropout_dec), num_layers)
        self.transformer_encoder_entities = nn.TransformerEncoder(
            nn.TransformerEncoderLayer(emb_dim, num_heads, encoder_dim, dropout_enc), num_layers,
            enable_nested_tensor=False)
        if v != "geo":
            self.transformer_encoder_facts = nn.TransformerEncoder(
                nn.TransformerEncoderLayer(emb_dim, num_heads, encoder_dim, dropout_enc), num_layers,
                enable_nested_tensor=False)
        self.word_embedding = nn.Embedding(self.vocab_size, emb_dim)
        self.entity_encoder = EntityEncoder(
            emb_dim, nn.Embedding(VARIANT_NUM_TYPES[v], emb_dim - VARIANT_TYPE_OFFSET[v]))
        if v != "geo":
            self.num_predicates = VARIANT_NUM_PREDICATES[v]
            self.predicate_embedding = nn.Embedding(self.num_predicates, emb_dim)
            self.fact_encoder = FactEncoder(emb_dim, self.predicate_embedding)
        self.caption_embedder = CaptionEmbedder(self.vocab_size)
        self.fc_vocab = nn.Linear(emb_dim, self.vocab_size)
        self.fc_entity = nn.Linear(emb_dim, 1)
        if v != "geo":
            self.fc_fact = nn.Linear(emb_dim, 1)
            self.fc_predicate = nn.Linear(self.num_predicates, emb_dim)
        self.init_weights()

    # ------------------------------------------------------------------ reference API surface
    def init_weights(self):
        """U(-0.1, 0.1) weights / zero bias on the score-head linears (geo-aware/models.py:264-272)."""
        heads = [self.fc_vocab, self.fc_entity]
        if self.variant != "geo":
            heads += [self.fc_fact, self.fc_predicate]
        with torch.no_grad():
            for m in heads:
                m.bias.zero_()
                m.weight.uniform_(-0.1, 0.1)

    def load_pretrained_embeddings(self, embeddings):
        self.word_embedding.weight = nn.Parameter(embeddings)
        self.invalidate_caches()      # a new Parameter object: cached parameter list, packed weights, graphs

    def fine_tune_embeddings(self, fine_tune=True):
        for p in self.word_embedding.parameters():
            p.requires_grad = fine_tune

    def _generate_square_subsequent_mask(self, sz):
        """Kept for API parity (geo-aware/models.py:256-262); the HIP attention kernel applies the
        causal mask from indices and never reads this tensor."""
        return torch.full((sz, sz), float("-inf")).triu(1)

    # ------------------------------------------------------------------ helpers
    @property
    def has_facts(self):
        return self.variant != "geo"

    @property
    def num_heads(self):
        return self.transformer_decoder.layers[0].self_attn.num_heads

    def _dropout_rates(self):
        return (self.pos_encoder.dropout.p, self.transformer_decoder.layers[0].dropout.p,
                self.transformer_encoder_entities.layers[0].dropout.p)

    def _wants_grad(self):
        return torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())

    def __getstate__(self):
        return _state_without_caches(self)

    def invalidate_caches(self):
        """Call after parameters were modified outside torch's version tracking (the fused Adam kernel
        writes the flat bucket directly): drops the packed cross-K/V weights, the transposed predicate
        weights and every captured graph."""
        self.__dict__["_param_epoch"] = self.__dict__.get("_param_epoch", 0) + 1
        for k in ("_kv_pack", "_pred_wt_cache", "_graphs", "_plist", "_dec_pack", "_last_static"):   # (_chain_cache keeps its buffer:
            # captured training graphs write it; its key holds _param_epoch, so the contents are refreshed)
            self.__dict__.pop(k, None)

    def _token_major(self, encoder_out):
        """(B, d, P) -> contiguous (B, P, d) storage.  Our Encoder already stores token-major."""
        B, d, P = encoder_out.shape
        if encoder_out.stride() == (P * d, 1, d):
            return encoder_out.permute(0, 2, 1)
        return encoder_out.permute(0, 2, 1).contiguous()

    def attach_encoder(self, encoder):
        """Let forward() / predict() / predict_beam() take the (B, 2048, 14, 14) feature map in place of encoder_out:
        Encoder.conv1 then runs INSIDE the captured graph, on the main stream beside the context-encoder chain on the
        side stream -- instead of a stand-alone launch the graph (and with it the context chain, which never reads the
        image) has to wait for.  Same kernels, same numbers (tests/test_round4_gpu.py); cfg2 forward 0.76 -> ~0.6 ms,
        cfg5 greedy 2.15 -> ~1.9 ms.  encoder_out tensors (B, emb_dim, 196) keep working as before.  The encoder is not
        registered as a submodule (state_dict / parameters() of the decoder stay the reference's)."""
        self.__dict__["_enc"] = encoder
        return self

    def _image_input(self, encoder_out):
        """-> (enc_in, P): the token-major (B, P, d) view of an encoder output, or -- with an attached Encoder and the
        hipGraph paths on -- the float32 feature map itself (4-D) for _encode_context to project inside the graph."""
        if encoder_out.dim() == 4:
            enc = self.__dict__.get("_enc")
            if enc is None:
                raise IckError("a 4-D input is a feature map: attach_encoder(encoder) first, or pass encoder(feats)")
            if self.use_hip_graphs and not self._wants_grad() and not os.environ.get("ICK_NO_FUSED_ENCODER") and \
                    encoder_out.shape[1] == enc.encoder_dim:
                feats = encoder_out if encoder_out.dtype == torch.float32 else encoder_out.float()
                enc.conv1_presplit()       # refreshed in place, outside the captured graph, if conv1's weight changed
                return feats.contiguous(), feats.shape[2] * feats.shape[3]
            encoder_out = enc(encoder_out)
        t = self._token_major(encoder_out)
        return t, t.shape[1]

    def _packed_cross_kv(self):
        """[K_0;V_0;K_1;V_1;...] rows of the decoder layers' cross-attention in_proj, so the memory is
        projected for all layers by one GEMM (cached until a parameter changes)."""
        layers = self.transformer_decoder.layers
        key = tuple(l.multihead_attn.in_proj_weight._version for l in layers) + tuple(
            l.multihead_attn.in_proj_bias._version for l in layers) + tuple(
            l.multihead_attn.in_proj_weight.data_ptr() for l in layers)
        cache = self.__dict__.get("_kv_pack")
        if cache is None or cache[0] != key:
            d = self.emb_dim
            w = torch.cat([l.multihead_attn.in_proj_weight.detach()[d:] for l in layers]).contiguous()
            b = torch.cat([l.multihead_attn.in_proj_bias.detach()[d:] for l in layers]).contiguous()
            cache = (key, w, b)
            self.__dict__["_kv_pack"] = cache
        return cache[1], cache[2]

    def _cross_kv_presplit(self, wkv):
        """Pre-split copy of the all-layer cross K/V weight (the image rows' projection reads it as b_ps)."""
        layers = self.transformer_decoder.layers
        key = tuple(l.multihead_attn.in_proj_weight._version for l in layers) + (
            wkv.data_ptr(), self.__dict__.get("_param_epoch", 0))
        return ops.presplit_cached(self, "wkv", wkv, key)

    def _vocab_presplit(self):
        w = self.fc_vocab.weight
        return ops.presplit_cached(self, "vocab", w.detach(), (w._version, w.data_ptr(), self.__dict__.get("_param_epoch", 0)))

    def _chain_items(self):
        """(key, weight view) of every nn.Linear that a row-chain launch (ops.rowchain_fwd) multiplies with: per layer
        the self-attention out-projection, the cross-attention q-projection and out-projection, linear1, linear2,
        and -- from the second layer of a stack on -- the self-attention in_proj, which rides on the previous layer's
        linear2 + norm launch."""
        d = self.emb_dim
        items = []
        for li, layer in enumerate(self.transformer_decoder.layers):
            items += [(("d", li, "so"), layer.self_attn.out_proj.weight), (("d", li, "cq"), layer.multihead_attn.in_proj_weight[:d]),
                      (("d", li, "co"), layer.multihead_attn.out_proj.weight), (("d", li, "l1"), layer.linear1.weight),
                      (("d", li, "l2"), layer.linear2.weight)]
            if li > 0:
                items.append((("d", li, "si"), layer.self_attn.in_proj_weight))     # layer 0's in_proj is a plain GEMM
        stacks = [("e", self.transformer_encoder_entities)]
        if self.has_facts:
            stacks.append(("f", self.transformer_encoder_facts))
        for tag, stack in stacks:
            for li, layer in enumerate(stack.layers):
                items += [((tag, li, "so"), layer.self_attn.out_proj.weight), ((tag, li, "l1"), layer.linear1.weight),
                          ((tag, li, "l2"), layer.linear2.weight)]
                if li > 0:
                    items.append(((tag, li, "si"), layer.self_attn.in_proj_weight))
        return items

    def chain_supported(self):
        """Do the layer widths fit the row-chain kernel (ick_rowchain_supported)?  ICK_NO_ROWCHAIN=1 turns it off."""
        cached = self.__dict__.get("_chain_ok")
        if cached is None:
            d = self.emb_dim
            ff = max(l.linear1.out_features for st in (self.transformer_decoder, self.transformer_encoder_entities)
                     for l in st.layers)
            cached = (not os.environ.get("ICK_NO_ROWCHAIN")) and ops.rowchain_supported(ff, d, max(3 * d, ff))
            self.__dict__["_chain_ok"] = cached
        return cached

    def _chain_items_bwd(self):
        """(key, transposed weight view) for the backward chains (ops.rowchain_bwd): the data gradient of a Linear
        multiplies with W, i.e. the row-chain GEMM reads a packed copy of W.T."""
        items = []
        for key, w in self._chain_items():
            items.append(((key[0], key[1], key[2] + "T"), w.t()))
        return items

    def _chain_pack(self, fresh=False, bwd=False, subset=None, extra=None, copies=()):
        """Packed copies (ops.pack_weights) of the weights the row-chain launches read, as {key: tensor} views of one
        persistent buffer (bwd: of the transposed weights, for the data-gradient chains).  Refreshed when a parameter's
        version changed, or on every call with fresh=True (inside the captured training step, where the fused Adam
        updates the weights behind torch's version counters); then `subset` (a predicate on the item key) limits the
        launch to the copies the caller needs first -- the rest follows in a later call.  extra: (key, 2-D view) of
        per-call tensors packed in the same launch into persistent buffers of their own (the transposed all-layer cross
        K/V weight of the backward pass); their copies are returned under `key`.  copies: plain (src, dst) 2-D copies
        that ride in the same launch (only with fresh=True)."""
        items = self._chain_items_bwd() if bwd else self._chain_items()
        name = "_chain_cache_bwd" if bwd else "_chain_cache"
        cache = self.__dict__.get(name)
        key = tuple(w._version for _, w in items) + (items[0][1].data_ptr(), self.__dict__.get("_param_epoch", 0))
        if cache is None or cache["ptr"] != items[0][1].data_ptr() or cache["buf"].device != items[0][1].device:
            sizes = [ops.packed_weight_floats(w.shape[0], w.shape[1]) for _, w in items]
            buf = torch.empty(sum(sizes), device=items[0][1].device, dtype=torch.float32)
            views, off = {}, 0
            for (k, _), n in zip(items, sizes):
                views[k] = buf[off:off + n]
                off += n
            cache = {"ptr": items[0][1].data_ptr(), "buf": buf, "views": views, "key": None}
            self.__dict__[name] = cache
        more = []
        for k, w in (extra or []):
            n = ops.packed_weight_floats(w.shape[0], w.shape[1])
            buf = cache.setdefault("extra", {}).get(k)
            if buf is None or buf.numel() != n or buf.device != w.device:
                buf = cache["extra"][k] = torch.empty(n, device=w.device, dtype=torch.float32)
            more.append((w.detach(), buf))
        assert not copies or fresh
        if fresh and subset is not None:
            ops.pack_weights([(w.detach(), cache["views"][k]) for k, w in items if subset(k)] + more, copies)
        elif fresh or cache["key"] != key:
            ops.pack_weights([(w.detach(), cache["views"][k]) for k, w in items] + more, copies)
            cache["key"] = key
        elif more:
            ops.pack_weights(more)
        if extra:
            views = dict(cache["views"])
            views.update({k: cache["extra"][k] for k, _ in extra})
            return views
        return cache["views"]

    def chain_bwd_supported(self):
        cached = self.__dict__.get("_chain_bwd_ok")
        if cached is None:
            d = self.emb_dim
            ff = max(l.linear1.out_features for st in (self.transformer_decoder, self.transformer_encoder_entities)
                     for l in st.layers)
            cached = self.chain_supported() and not os.environ.get("ICK_NO_ROWCHAIN_BWD") and \
                ops.rowchain_bwd_supported(max(3, 2 * len(self.transformer_decoder.layers)) * d, d, ff)
            self.__dict__["_chain_bwd_ok"] = cached
        return cached

    def _pred_wt(self):
        w = self.fc_predicate.weight
        key = (w._version, w.data_ptr())
        cache = self.__dict__.get("_pred_wt_cache")
        if cache is None or cache[0] != key:
            cache = (key, w.detach().t().contiguous())
            self.__dict__["_pred_wt_cache"] = cache
        return cache[1]

    def _context_encoder(self, stack, x, tag="e", out=None, slim=False):
        """Post-LN encoder stack on x (B, T, d); `out` (optional (B, T, d) view, e.g. rows of the memory buffer)
        receives the last layer's output."""
        H = self.num_heads
        d = self.emb_dim
        B, T, _ = x.shape
        chain = self.chain_supported()
        pk = self._chain_pack() if chain else None
        n = len(stack.layers)
        qkv = None
        for li, layer in enumerate(stack.layers):
            if qkv is None:
                qkv = ops.project_heads(x, layer.self_attn.in_proj_weight.detach(), layer.self_attn.in_proj_bias.detach(),
                                        3, H, T)
            sa = torch.empty_like(x)
            ops.attention_heads(qkv, qkv, sa, H, d // H, T, T, q_seg=0, k_seg=1, v_seg=2)
            qkv = None
            last = li == n - 1
            if chain:
                x1 = torch.empty_like(x)
                f = torch.empty(B, T, layer.linear1.out_features, device=x.device, dtype=torch.float32)
                ops.rowchain_fwd(sa, pk[(tag, li, "so")], layer.self_attn.out_proj.bias.detach(), x,
                                 layer.norm1.weight.detach(), layer.norm1.bias.detach(), layer.norm1.eps, x1,
                                 w2p=pk[(tag, li, "l1")], b2=layer.linear1.bias.detach(), y2=f, relu=True, slim=slim)
                x2 = out if (last and out is not None) else torch.empty_like(x)
                nxt = None if last else stack.layers[li + 1]
                if nxt is not None:
                    qkv = torch.empty(B, 3, H, T, ops.DHP, device=x.device, dtype=torch.float32)
                ops.rowchain_fwd(f, pk[(tag, li, "l2")], layer.linear2.bias.detach(), x1, layer.norm2.weight.detach(),
                                 layer.norm2.bias.detach(), layer.norm2.eps, x2,
                                 w2p=None if nxt is None else pk[(tag, li + 1, "si")],
                                 b2=None if nxt is None else nxt.self_attn.in_proj_bias.detach(), y2=qkv,
                                 heads=None if nxt is None else (3, H, T, 0, T), slim=slim)
                x = x2
                continue
            o = ops.linear(sa, layer.self_attn.out_proj.weight.detach(), layer.self_attn.out_proj.bias.detach())
            x = ops.add_layernorm(o, x, layer.norm1.weight.detach(), layer.norm1.bias.detach(), layer.norm1.eps)
            f = ops.linear(x, layer.linear1.weight.detach(), layer.linear1.bias.detach(), relu=True)
            o = ops.linear(f, layer.linear2.weight.detach(), layer.linear2.bias.detach())
            x = ops.add_layernorm(o, x, layer.norm2.weight.detach(), layer.norm2.bias.detach(), layer.norm2.eps)
            if last and out is not None:
                out.copy_(x)
                x = out
        return x

    def _encode_context(self, enc_tok, entities, facts, gmap):
        """Entity / fact encoders, context transformers and the all-layer cross K/V projection.
        Returns (entities_encoded, facts_encoded, kv, contexts); kv is head-major
        (B, 2*layers, H, S, 32): segment 2i = keys of decoder layer i, 2i+1 = its values, over the
        memory rows [196 image positions ; entity rows ; fact rows]."""
        d = self.emb_dim
        H = self.num_heads
        feats = None
        if enc_tok.dim() == 4:
            # the feature map itself (attach_encoder): Encoder.conv1 runs below, beside the context chain
            feats, enc = enc_tok, self.__dict__["_enc"]
            B, Cc = feats.shape[:2]
            P = feats.shape[2] * feats.shape[3]
            enc_tok = torch.empty(B, P, d, device=feats.device, dtype=torch.float32)
        else:
            B, P, _ = enc_tok.shape
        K = entities.shape[1]
        ee = ops.entity_encode(self.variant, entities, self.entity_encoder.type_embedding.weight.detach(), d,
                               facts=facts if self.has_facts else None,
                               word_emb=self.word_embedding.weight.detach() if self.variant == "news" else None)
        fe = None
        Fn = 0
        if self.has_facts:
            Fn = facts.shape[1]
            fe = ops.fact_encode(facts, ee, self.predicate_embedding.weight.detach())
        wkv, bkv = self._packed_cross_kv()
        wkv_ps = self._cross_kv_presplit(wkv)
        if self.chain_supported():
            self._chain_pack()      # refreshed (if stale) on the main stream, before the side stream forks
        nseg = wkv.shape[0] // d
        S = P + K + Fn
        kv = torch.empty(B, nseg, H, S, ops.DHP, device=enc_tok.device, dtype=torch.float32)
        # The context-encoder chain (small, latency-bound kernels) runs on a second stream beside the
        # large image-row projection; both write disjoint key/value rows of `kv`.  The caller joins the
        # side stream before the first cross-attention (`side.join()`).
        side = ops.SideStream(priority=-1)
        ctx = [None, None]
        # (Measured and removed in round 4: only layer 0's image K/V in front of the decoder and the later layers' on the side
        # stream behind the context chain -- forward 0.695 -> 0.826 ms, a second fork / join pair makes the hipGraph
        # executor start the context chain only after conv1)

        def entity_chain():
            ops.stamp("side: context chain starts")
            # beside Encoder.conv1 / the image K/V projection: the 8-wave form finds room on a CU that hosts bulk GEMM
            # workgroups (as in the training step)
            ctx[0] = self._context_encoder(self.transformer_encoder_entities, ee, slim=True)
            ops.project_heads(ctx[0], wkv, bkv, nseg, H, S, out=kv, s0=P, grp=K)
            side.signal("ctx")
            ops.stamp("side: context chain done")

        def fact_chain():
            ctx[1] = self._context_encoder(self.transformer_encoder_facts, fe, tag="f")
            ops.project_heads(ctx[1], wkv, bkv, nseg, H, S, out=kv, s0=P + K, grp=Fn)

        # dependency point now, enqueued after the main stream's next kernel: in a captured graph the main chain
        # must be the first child of the fork node (see SideStream)
        side.submit(entity_chain, ee, fe, kv, wkv, bkv)

        def conv1():
            c1 = enc.conv1
            ops.gemm_raw(feats, c1.weight.detach().view(d, Cc), enc_tok, B * P, d, Cc, 1, P, Cc, 1, d,
                         bias=c1.bias.detach(), a_grp=P, a_gs=Cc * P, b_ps=enc.conv1_presplit())

        if self.has_facts:
            # the fact chain runs on the main stream beside the entity chain (two chains of small kernels overlap well,
            # a chain beside the large projection does not: cfg4 forward 1.84 -> 1.72 ms), the image rows follow
            fact_chain()
            side.flush()
            if feats is not None:
                conv1()
            ops.project_heads(enc_tok, wkv, bkv, nseg, H, S, out=kv, s0=0, grp=P, a_gmap=gmap, a_gs=enc_tok.stride(0),
                              w_ps=wkv_ps)
        else:
            # image rows (gathered through gmap = sort order)
            if feats is not None:
                conv1()
                side.flush()       # the context chain starts beside Encoder.conv1
            ops.project_heads(enc_tok, wkv, bkv, nseg, H, S, out=kv, s0=0, grp=P, a_gmap=gmap, a_gs=enc_tok.stride(0),
                              w_ps=wkv_ps)
            side.flush()
        ops.stamp("main: image K/V projection done")
        ctx_e, ctx_f = ctx
        return ee, fe, kv, (ctx_e, ctx_f), side

    def _decoder_layer(self, li, layer, x, kv, S, qkv_buf=None, pos=None, side=None, qkv=None, want_next=False):
        """One post-LN decoder layer on x (B, T, d).  With qkv_buf (B, 3, H, max_len, 32) the layer runs
        one KV-cached decode step: the new q|k|v row is written at position `pos` and attends to [0, pos].
        qkv: this layer's head-major q|k|v when the previous layer's last launch already projected it (row chains);
        want_next: return (x, next layer's qkv) -- the in_proj of layer li + 1 rides on this layer's linear2 + norm3."""
        H, d = self.num_heads, self.emb_dim
        dh = d // H
        B, T, _ = x.shape
        sa_w, sa_b = layer.self_attn.in_proj_weight.detach(), layer.self_attn.in_proj_bias.detach()
        sa = torch.empty_like(x)
        chain = qkv_buf is None and self.chain_supported()
        if qkv_buf is None:
            if qkv is None:
                qkv = ops.project_heads(x, sa_w, sa_b, 3, H, T)
            ops.attention_heads(qkv, qkv, sa, H, dh, T, T, q_seg=0, k_seg=1, v_seg=2, causal=True)
        else:
            ML = qkv_buf.shape[3]
            ops.project_heads(x, sa_w, sa_b, 3, H, ML, out=qkv_buf, s0=pos, grp=1)
            ops.attention_heads(qkv_buf, qkv_buf, sa, H, dh, 1, pos + 1, q_seg=0, k_seg=1, v_seg=2, q_t0=pos)
        ca_w, ca_b = layer.multihead_attn.in_proj_weight.detach(), layer.multihead_attn.in_proj_bias.detach()
        if chain:
            pk = self._chain_pack()
            x1 = torch.empty_like(x)
            q = torch.empty(B, 1, H, T, ops.DHP, device=x.device, dtype=torch.float32)
            ops.rowchain_fwd(sa, pk[("d", li, "so")], layer.self_attn.out_proj.bias.detach(), x, layer.norm1.weight.detach(),
                             layer.norm1.bias.detach(), layer.norm1.eps, x1, w2p=pk[("d", li, "cq")], b2=ca_b[:d], y2=q,
                             heads=(1, H, T, 0, T))
            ca = torch.empty_like(x)
            if side is not None:
                side.join()       # entity / fact rows of kv come from the side stream
            ops.attention_heads(q, kv, ca, H, dh, T, S, q_seg=0, k_seg=2 * li, v_seg=2 * li + 1)
            x2 = torch.empty_like(x)
            f = torch.empty(B, T, layer.linear1.out_features, device=x.device, dtype=torch.float32)
            ops.rowchain_fwd(ca, pk[("d", li, "co")], layer.multihead_attn.out_proj.bias.detach(), x1,
                             layer.norm2.weight.detach(), layer.norm2.bias.detach(), layer.norm2.eps, x2,
                             w2p=pk[("d", li, "l1")], b2=layer.linear1.bias.detach(), y2=f, relu=True)
            layers = self.transformer_decoder.layers
            nxt = layers[li + 1] if (want_next and li + 1 < len(layers)) else None
            x3 = torch.empty_like(x)
            qkv_n = None if nxt is None else torch.empty(B, 3, H, T, ops.DHP, device=x.device, dtype=torch.float32)
            ops.rowchain_fwd(f, pk[("d", li, "l2")], layer.linear2.bias.detach(), x2, layer.norm3.weight.detach(),
                             layer.norm3.bias.detach(), layer.norm3.eps, x3,
                             w2p=None if nxt is None else pk[("d", li + 1, "si")],
                             b2=None if nxt is None else nxt.self_attn.in_proj_bias.detach(), y2=qkv_n,
                             heads=None if nxt is None else (3, H, T, 0, T))
            return (x3, qkv_n) if want_next else x3
        o = ops.linear(sa, layer.self_attn.out_proj.weight.detach(), layer.self_attn.out_proj.bias.detach())
        x = ops.add_layernorm(o, x, layer.norm1.weight.detach(), layer.norm1.bias.detach(), layer.norm1.eps)
        q = ops.project_heads(x, ca_w[:d], ca_b[:d], 1, H, T)
        ca = torch.empty_like(x)
        if side is not None:
            side.join()    # entity / fact rows of kv come from the side stream
        ops.attention_heads(q, kv, ca, H, dh, T, S, q_seg=0, k_seg=2 * li, v_seg=2 * li + 1)
        o = ops.linear(ca, layer.multihead_attn.out_proj.weight.detach(), layer.multihead_attn.out_proj.bias.detach())
        x = ops.add_layernorm(o, x, layer.norm2.weight.detach(), layer.norm2.bias.detach(), layer.norm2.eps)
        f = ops.linear(x, layer.linear1.weight.detach(), layer.linear1.bias.detach(), relu=True)
        o = ops.linear(f, layer.linear2.weight.detach(), layer.linear2.bias.detach())
        x = ops.add_layernorm(o, x, layer.norm3.weight.detach(), layer.norm3.bias.detach(), layer.norm3.eps)
        return (x, None) if want_next else x

    def _score_head(self, h, ee, fe, eib, gate, out=None):
        """get_scores: vocabulary logits and pointer scores written into one (B, T, V+K[+F]) buffer."""
        B, T, d = h.shape
        V, K = self.vocab_size, ee.shape[1]
        Fn = fe.shape[1] if fe is not None else 0
        Vx = V + K + Fn
        if out is None:
            out = torch.empty(B, T, Vx, device=h.device, dtype=torch.float32)
        hv = ops.mul(h, gate) if self.has_facts else h
        ops.gemm_raw(hv, self.fc_vocab.weight.detach(), out, B * T, V, d, d, 1, d, 1, Vx,
                     bias=self.fc_vocab.bias.detach(), b_ps=self._vocab_presplit() if B * T >= 256 else None)
        ops.pointer_scores(h, ee, self.fc_entity.weight.detach(), self.fc_entity.bias.detach(), out, V)
        if self.has_facts:
            ops.pointer_scores(h, fe, self.fc_fact.weight.detach(), self.fc_fact.bias.detach(), out, V + K, ind=eib)
        return out

    # ------------------------------------------------------------------ public score-head methods
    @torch.no_grad()
    def get_context_indicators(self, captions, facts, entity_context_size, out_length):
        """knowledge-aware/models.py:380-418 (news: same): captions (B, L) token indices, facts (B, F, 3) ->
        (entity_idx_before (B, out_length, F, 1), predicate_indicator (B, out_length, num_predicates, 1)), dense 0/1
        float tensors like the reference's.  out_length == 1 is predict()'s form ("mentioned anywhere in the buffer"),
        otherwise position p sees the mentions strictly before p.  One launch of ick_context_indicators (forward()
        itself never builds the dense predicate indicator: it uses the kernel's fused fc_predicate form)."""
        if not self.has_facts:
            raise AttributeError("the geo variant has no get_context_indicators (geo-aware/models.py)")
        dev = self.fc_vocab.weight.device
        captions = captions.to(device=dev, dtype=torch.int64)
        facts = facts.to(device=dev, dtype=torch.int64).contiguous()
        B, Lc = captions.shape
        mode = 1 if out_length == 1 else 0
        if mode == 0 and out_length > Lc:      # positions past the caption see every mention: pad with <pad> tokens
            captions = torch.cat([captions, captions.new_full((B, out_length - Lc), self.word_map["<pad>"])], dim=1)
        eib, pi = ops.context_indicators(captions.contiguous(), facts, entity_context_size, self.vocab_size, mode=mode,
                                         dense_pred=self.num_predicates)
        return eib[:, :out_length].unsqueeze(3), pi[:, :out_length].unsqueeze(3)

    @torch.no_grad()
    def get_scores(self, h, entities_encoded, facts_encoded=None, entity_idx_before=None, predicate_indicator=None):
        """geo-aware/models.py:291-313 / knowledge-aware/models.py:420-455: h (L, B, d) decoder states,
        entities_encoded (B, K, d) [, facts_encoded (B, F, d), entity_idx_before (B, L, F, 1), predicate_indicator
        (B, L, num_predicates, 1)] -> scores (L, B, V+K[+F]).  Runs the score-head kernels (vocabulary GEMM written in
        place + pointer-score kernels; the predicate gate from the DENSE indicator is one more GEMM).  Inference
        helper: gradients flow through forward() only."""
        dev = self.fc_vocab.weight.device
        hb = h.detach().to(dev, torch.float32).permute(1, 0, 2).contiguous()           # (B, L, d) rows
        ee = entities_encoded.detach().to(dev, torch.float32).contiguous()
        fe = eib = gate = None
        if self.has_facts:
            if facts_encoded is None or entity_idx_before is None or predicate_indicator is None:
                raise IckError("%s variant: get_scores(h, entities_encoded, facts_encoded, entity_idx_before, "
                               "predicate_indicator)" % self.variant)
            B, T, d = hb.shape
            fe = facts_encoded.detach().to(dev, torch.float32).contiguous()
            eib = entity_idx_before.detach().to(dev, torch.float32).reshape(B, T, fe.shape[1]).contiguous()
            pi = predicate_indicator.detach().to(dev, torch.float32).reshape(B * T, self.num_predicates).contiguous()
            gate = ops.linear(pi, self.fc_predicate.weight.detach(), self.fc_predicate.bias.detach()).view(B, T, d)
        return self._score_head(hb, ee, fe, eib, gate).permute(1, 0, 2)

    def _prepare_inputs(self, encoder_out, entities, facts):
        dev = self.fc_vocab.weight.device
        if dev.type != "cuda":
            raise IckError("DecoderTransformer parameters must live on the GPU (decoder.to('cuda'))")
        entities = entities.to(device=dev, dtype=torch.float32)
        if self.has_facts:
            if facts is None:
                raise IckError("%s variant needs the facts tensor" % self.variant)
            facts = facts.to(device=dev, dtype=torch.int64)
        return encoder_out.to(dev), entities, facts

    # ------------------------------------------------------------------ forward (teacher forced)
    def _forward_device(self, captions, caption_masks, entities, facts, enc_tok, gmap, stages=None):
        """Device-only part of forward() on length-sorted inputs (no host synchronisation inside)."""
        d, V = self.emb_dim, self.vocab_size
        K = entities.shape[1]
        ops.stamp("forward: start")
        pe = self.pos_encoder.pe.view(-1, d)

        def embed(ee, fe):
            return ops.caption_embed(captions, caption_masks, self.word_embedding.weight.detach(), ee, fe, pe, V,
                                     self.word_map["<pad>"], math.sqrt(d), want_emb=True)

        # (Measured and removed in round 4: layer 0's caption embedding + in_proj at the head of the side stream -- it delays
        # the context chain, which is what the first cross-attention waits for: forward 0.694 -> 0.715 ms.)
        ee, fe, kv, ctx, side = self._encode_context(enc_tok, entities, facts, gmap)
        x, emb = embed(ee, fe)
        qkv = None
        S = kv.shape[3]
        for li, layer in enumerate(self.transformer_decoder.layers):
            x, qkv = self._decoder_layer(li, layer, x, kv, S, qkv=qkv, want_next=True, side=side if li == 0 else None)
            ops.stamp("main: decoder layer %d done" % li)
        side.join()
        eib = gate = None
        if self.has_facts:
            eib, gate = ops.context_indicators(captions, facts, K, V, self._pred_wt(),
                                               self.fc_predicate.bias.detach(), mode=0)
        scores = self._score_head(x, ee, fe, eib, gate)
        ops.stamp("forward: scores done")
        if stages is not None:
            stages.update(entities_encoded=ee, facts_encoded=fe, embeddings=emb, entity_context=ctx[0],
                          fact_context=ctx[1], h=x, kv=kv, eib=eib, gate=gate)
        return scores

    def _enc_key(self, enc_in):
        """Graph-key part for a feature-map input: the attached encoder's conv1 is read through its device pointers."""
        if enc_in.dim() != 4:
            return ()
        enc = self.__dict__["_enc"]
        c1 = enc.conv1
        return (c1.weight.data_ptr(), c1.bias.data_ptr(), c1.weight._version, c1.bias._version,
                enc.__dict__.get("_param_epoch", 0))

    def input_buffers(self):
        """The static input tensors of the graph the last forward() / predict() / predict_beam() call replayed (None
        before the first graphed call), in that call's order -- forward: [captions, caption_masks, entities, facts,
        image input]; predict*: [image input, entities, facts].  A loader that copies the next batch straight into them
        and passes them back in saves the device-to-device input copy of every call (the feature map is 103 MB at
        B = 64)."""
        return self.__dict__.get("_last_static")

    def _graphed(self, kind, key, fn, inputs):
        """Replay (capturing on first use) the hipGraph of `fn` for this shape key; parameters are read
        through their device pointers, so in-place weight updates are seen, re-allocation is not
        (the key includes the parameters' storage pointers)."""
        cache = self.__dict__.setdefault("_graphs", {})
        plist = self.__dict__.get("_plist")
        if plist is None:       # the module tree is fixed after construction; invalidate_caches() drops this list
            plist = self.__dict__["_plist"] = list(self.parameters())
        pkey = (self.__dict__.get("_param_epoch", 0),) + tuple(p._version for p in plist) + \
            tuple(p.data_ptr() for p in plist)
        # the product mode of the large GEMM tiles is baked into a capture (and decides whether the pre-split weight copies a
        # graph reads are being refreshed at all): a graph is only replayed in the mode it was captured in
        full = (kind, key, pkey, ops.gemm_split_mode())
        g = cache.get(full)
        if g is None:
            if len(cache) >= 8:
                cache.clear()
            g = cache[full] = _GraphedCall(fn, inputs)
        self.__dict__["_last_static"] = g.static_in
        return g(*inputs)

    def forward(self, captions, encoder_out, caption_masks, caption_lengths, entities, facts=None, stages=None):
        encoder_out, entities, facts = self._prepare_inputs(encoder_out, entities, facts)
        dev = encoder_out.device
        captions = captions.to(dev)
        caption_masks = caption_masks.to(dev)
        if encoder_out.dim() == 4 and not (stages is None and not self._wants_grad() and self.use_hip_graphs):
            encoder_out = self.__dict__["_enc"](encoder_out) if "_enc" in self.__dict__ else encoder_out
        enc_tok, _ = self._image_input(encoder_out)

        def sorted_inputs(c, m, e, f, sd):
            """Batch permutation into length order (reference: geo-aware/models.py:330-336), on the device."""
            return (c.index_select(0, sd), m.index_select(0, sd), e.index_select(0, sd),
                    None if f is None else f.index_select(0, sd), sd.to(torch.int32))

        if stages is None and not self._wants_grad() and self.use_hip_graphs:
            # Samples are independent, so the captured graph runs the batch in the caller's order and only the results
            # are permuted into length order.  That takes the host's length sort off the device's critical path: the
            # lengths start their way to the host, the graph is launched, and while it runs the host sorts and enqueues
            # the two gathers behind it.  (Launching the graph after a blocking .cpu() left the GPU idle for ~160 us.)
            lens_dev = caption_lengths.detach().reshape(-1)
            ev = None
            if lens_dev.is_cuda:
                pin = self.__dict__.get("_len_pin")
                if pin is None or pin.shape != lens_dev.shape or pin.dtype != lens_dev.dtype:
                    pin = self.__dict__["_len_pin"] = torch.empty(lens_dev.shape, dtype=lens_dev.dtype, pin_memory=True)
                pin.copy_(lens_dev, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
            key = (tuple(captions.shape), tuple(enc_tok.shape), tuple(entities.shape),
                   None if facts is None else tuple(facts.shape)) + self._enc_key(enc_tok)
            scores_raw = self._graphed("fwd", key, lambda c, m, e, f, t: self._forward_device(c, m, e, f, t, None),
                                       [captions, caption_masks, entities, facts, enc_tok])
            if ev is not None:
                ev.synchronize()
                lens_host = pin.clone()
            else:
                lens_host = lens_dev
            lengths, sort_ind = lens_host.sort(dim=0, descending=True)
            # through pinned memory: a host-to-device copy from pageable memory blocks the host until the stream
            # (the whole graph) has drained
            pidx = self.__dict__.get("_idx_pin")
            if pidx is None or pidx.shape != sort_ind.shape:
                pidx = self.__dict__["_idx_pin"] = torch.empty(sort_ind.shape, dtype=torch.int64, pin_memory=True)
            prev = self.__dict__.get("_pin_ev")
            if prev is not None:
                prev.synchronize()    # the previous call's asynchronous copy out of this pinned buffer has executed
            pidx.copy_(sort_ind)
            sort_dev = pidx.to(dev, non_blocking=True)
            prev = self.__dict__["_pin_ev"] = torch.cuda.Event()
            prev.record()
            return scores_raw.index_select(0, sort_dev), captions.index_select(0, sort_dev), (lengths - 1).tolist()
        # length sort on the host, like the reference's CPU path (the result feeds a Python list anyway)
        lengths, sort_ind = caption_lengths.detach().reshape(-1).cpu().sort(dim=0, descending=True)
        decode_lengths = (lengths - 1).tolist()
        sort_dev = sort_ind.to(dev, non_blocking=True)
        captions, caption_masks, entities, facts, gmap = sorted_inputs(captions, caption_masks, entities, facts, sort_dev)
        if self._wants_grad() and stages is None:
            # autograd bridge: the HIP backward pass runs when the caller's loss.backward() reaches us
            from . import training
            scores = training.DecoderGraphFn.apply(self, captions, caption_masks, entities, facts, enc_tok, gmap,
                                                   *training.unique_parameters(self))
            return scores, captions, decode_lengths
        return self._forward_device(captions, caption_masks, entities, facts, enc_tok, gmap, stages), captions, \
            decode_lengths

    # ------------------------------------------------------------------ greedy decode (KV cached)
    def _decode_pack(self):
        """Transposed out_proj / linear2 weights of the decoder layers for the fused decode kernels (one input
        feature per row, so a workgroup's slice of the out-projection is read coalesced); cached until a parameter
        changes."""
        layers = self.transformer_decoder.layers
        src = [w for l in layers for w in (l.self_attn.out_proj.weight, l.multihead_attn.out_proj.weight, l.linear2.weight)]
        key = tuple(w._version for w in src) + tuple(w.data_ptr() for w in src)
        cache = self.__dict__.get("_dec_pack")
        if cache is None or cache[0] != key:
            cache = (key, [w.detach().t().contiguous() for w in src])
            self.__dict__["_dec_pack"] = cache
        return cache[1]

    def _decode_ctx(self, kv, ee, fe, rows_per_sample, max_len, S, anc=None, want_scores=False, fuse_select=False,
                    n_done_init=0):
        """lib.DecodeCtx over freshly allocated state buffers for R = B * rows_per_sample rows (+ the tensors, kept
        alive by the caller).  See include/ick_amd.h (ick_decode_ctx)."""
        from . import lib as L
        dev = kv.device
        d, H, V = self.emb_dim, self.num_heads, self.vocab_size
        layers = self.transformer_decoder.layers
        nl = len(layers)
        B, K = ee.shape[0], ee.shape[1]
        Fn = fe.shape[1] if fe is not None else 0
        R = B * rows_per_sample
        FF = layers[0].linear1.out_features
        nch = (FF + 63) // 64
        ntiles = (V + 15) // 16
        f32 = dict(device=dev, dtype=torch.float32)
        # state buffers are initialised by ops.decode_init (one launch; they were eight fills, an embedding and a copy)
        t = {"x0": torch.empty(R, d, **f32), "xa": torch.empty(R, d, **f32), "xb": torch.empty(R, d, **f32),
             "xc": torch.empty(R, d, **f32), "p1": torch.empty(R, H, d, **f32), "p2": torch.empty(R, H, d, **f32),
             "p3": torch.empty(R, nch, d, **f32), "hfin": torch.empty(R, d, **f32), "hv": torch.empty(R, d, **f32),
             "ptr": torch.empty(R, K + Fn, **f32), "cand": torch.empty(R, ntiles, 4, **f32),
             "self_kv": torch.empty(nl, 2, R, H, max_len, ops.DHP, **f32),
             "output": torch.empty((R, max_len), dtype=torch.long, device=dev),
             "hist": torch.empty(R, max_len, dtype=torch.int32, device=dev),
             "finished": torch.empty(R, dtype=torch.int32, device=dev),
             "n_done": torch.empty(1, dtype=torch.int32, device=dev),
             "next_token": torch.empty(R, dtype=torch.long, device=dev),
             "next_mask": torch.empty(R, dtype=torch.long, device=dev),
             "pack": self._decode_pack(), "kv": kv, "ee": ee, "fe": fe, "anc": anc}
        if fuse_select:
            t["sel_state"] = torch.empty(2, R, 12, dtype=torch.int32, device=dev)
        if want_scores:
            t["scores"] = torch.empty(R, V, **f32)
        if self.has_facts:
            t["gate"] = torch.empty(R, 1, d, **f32)
            t["eib"] = torch.empty(R, 1, Fn, **f32)
            t["cap_buf"] = torch.empty((R, max_len), dtype=torch.long, device=dev)
        c = L.DecodeCtx()
        c.R, c.rows_per_sample, c.d, c.H, c.FF, c.layers, c.S, c.max_len = R, rows_per_sample, d, H, FF, nl, S, max_len
        c.V, c.K, c.F = V, K, Fn
        c.end_token, c.pad_token = self.word_map["<end>"], self.word_map["<pad>"]
        c.ln_eps, c.emb_scale = layers[0].norm1.eps, math.sqrt(d)
        c.kv_bs = kv.stride(0)
        seg = H * S * ops.DHP * 4      # bytes per (segment) of the memory projection
        pk = t["pack"]
        for li, l in enumerate(layers):
            w = c.layer[li]
            w.sa_in_w, w.sa_in_b = l.self_attn.in_proj_weight.data_ptr(), l.self_attn.in_proj_bias.data_ptr()
            w.sa_out_wt, w.sa_out_b = pk[3 * li].data_ptr(), l.self_attn.out_proj.bias.data_ptr()
            w.n1_g, w.n1_b = l.norm1.weight.data_ptr(), l.norm1.bias.data_ptr()
            w.ca_in_w, w.ca_in_b = l.multihead_attn.in_proj_weight.data_ptr(), l.multihead_attn.in_proj_bias.data_ptr()
            w.ca_out_wt, w.ca_out_b = pk[3 * li + 1].data_ptr(), l.multihead_attn.out_proj.bias.data_ptr()
            w.n2_g, w.n2_b = l.norm2.weight.data_ptr(), l.norm2.bias.data_ptr()
            w.w1, w.b1 = l.linear1.weight.data_ptr(), l.linear1.bias.data_ptr()
            w.w2t, w.b2 = pk[3 * li + 2].data_ptr(), l.linear2.bias.data_ptr()
            w.n3_g, w.n3_b = l.norm3.weight.data_ptr(), l.norm3.bias.data_ptr()
            w.self_k, w.self_v = t["self_kv"][li, 0].data_ptr(), t["self_kv"][li, 1].data_ptr()
            w.cross_k, w.cross_v = kv.data_ptr() + 2 * li * seg, kv.data_ptr() + (2 * li + 1) * seg
        c.anc = None if anc is None else anc.data_ptr()
        c.wv, c.bv = self.fc_vocab.weight.data_ptr(), self.fc_vocab.bias.data_ptr()
        c.we, c.be = self.fc_entity.weight.data_ptr(), self.fc_entity.bias.data_ptr()
        if self.has_facts:
            c.wf, c.bf = self.fc_fact.weight.data_ptr(), self.fc_fact.bias.data_ptr()
            c.fe, c.gate, c.eib, c.cap_buf = fe.data_ptr(), t["gate"].data_ptr(), t["eib"].data_ptr(), t["cap_buf"].data_ptr()
        c.ee = ee.data_ptr()
        c.word_emb, c.pe = self.word_embedding.weight.data_ptr(), self.pos_encoder.pe.data_ptr()
        for name in ("x0", "xa", "xb", "xc", "p1", "p2", "p3", "hfin", "hv", "ptr", "cand", "output", "hist", "finished",
                     "n_done", "next_token", "next_mask"):
            setattr(c, name, t[name].data_ptr())
        if want_scores:
            c.scores, c.scores_ld = t["scores"].data_ptr(), V
        if fuse_select:
            c.sel_state = t["sel_state"].data_ptr()
        ops.decode_init(c, self.word_map["<start>"], n_done_init)
        return c, t

    def _predict_fused(self, enc_tok, entities, facts, max_pred_len):
        """predict() on the fused decode kernels (csrc/decode.hip): 12 launches per token (13 with facts)."""
        B = enc_tok.shape[0]
        d, V, K = self.emb_dim, self.vocab_size, entities.shape[1]
        ee, fe, kv, _, side = self._encode_context(enc_tok, entities, facts, None)
        side.join()
        S = kv.shape[3]
        # the token of step i - 1 is chosen inside the first launch of step i (csrc/decode.hip: fused_select); only the
        # last step needs the selection kernel of its own: 11 launches per token (12 with facts)
        fuse = self.fuse_select
        c, t = self._decode_ctx(kv, ee, fe, 1, max_pred_len, S, fuse_select=fuse)

        def indicators():
            ops.context_indicators(t["cap_buf"], facts, K, V, self._pred_wt(), self.fc_predicate.bias.detach(),
                                   mode=1, eib=t["eib"], gate=t["gate"])

        for i in range(max_pred_len):
            if fuse and i > 0:
                ops.decode_layers_part(c, i, 1)           # selection of step i - 1 + first self-attention block
                if self.has_facts:
                    indicators()
                ops.decode_layers_part(c, i, 2)
            else:
                if self.has_facts:
                    indicators()
                ops.decode_layers(c, i)
            if not fuse or i == max_pred_len - 1:
                ops.decode_select_greedy(c, i)
        return t["output"]

    def _predict_beam_device(self, enc_tok, entities, facts, max_pred_len, beam):
        """Beam search on the fused decode kernels: R = B * beam rows share their caption's cross K/V; the
        self-attention cache is never reordered -- an ancestry table says which cache row holds position p of a
        hypothesis.  Returns (best sequence (B, max_len), its log-probability (B), all sequences, all scores)."""
        from . import lib as L
        B = enc_tok.shape[0]
        d, V, K = self.emb_dim, self.vocab_size, entities.shape[1]
        dev = enc_tok.device
        ee, fe, kv, _, side = self._encode_context(enc_tok, entities, facts, None)
        side.join()
        S = kv.shape[3]
        R = B * beam
        anc = [torch.zeros(R, max_pred_len, dtype=torch.int32, device=dev) for _ in range(2)]
        c, t = self._decode_ctx(kv, ee, fe, beam, max_pred_len, S, anc=anc[0], want_scores=True,
                                n_done_init=B * (beam - 1))             # the unused slots count as ended
        seq = [torch.full((R, max_pred_len), self.word_map["<pad>"], dtype=torch.long, device=dev) for _ in range(2)]
        cum = torch.full((B, beam), float("-inf"), device=dev)
        cum[:, 0] = 0.0                                   # one live hypothesis per caption at the start
        fin = torch.zeros(R, dtype=torch.int32, device=dev)
        facts_r = cap = None
        if self.has_facts:
            facts_r = facts.repeat_interleave(beam, dim=0).contiguous()
            cap = [torch.full((R, max_pred_len), self.word_map["<start>"], dtype=torch.long, device=dev)
                   for _ in range(2)]
        bs = L.BeamState()
        bs.cum, bs.fin, bs.start_token = cum.data_ptr(), fin.data_ptr(), self.word_map["<start>"]
        Vx = V + K + (fe.shape[1] if fe is not None else 0)
        rec = torch.empty(R, (Vx + 1023) // 1024, 18, device=dev, dtype=torch.float32)
        bs.rec = rec.data_ptr()
        for i in range(max_pred_len):
            cur, nxt = i & 1, (i + 1) & 1
            c.anc = anc[cur].data_ptr()
            if self.has_facts:
                ops.context_indicators(cap[cur], facts_r, K, V, self._pred_wt(), self.fc_predicate.bias.detach(), mode=1,
                                       eib=t["eib"], gate=t["gate"])
                bs.cap_in, bs.cap_out = cap[cur].data_ptr(), cap[nxt].data_ptr()
            ops.decode_layers(c, i)
            bs.seq_in, bs.seq_out = seq[cur].data_ptr(), seq[nxt].data_ptr()
            bs.anc_in, bs.anc_out = anc[cur].data_ptr(), anc[nxt].data_ptr()
            ops.decode_select_beam(c, bs, i)
        final = seq[max_pred_len & 1].view(B, beam, max_pred_len)
        best = cum.argmax(dim=1)                          # ties: the lower hypothesis
        out = final[torch.arange(B, device=dev), best]
        return out, cum.gather(1, best.view(B, 1)).view(B), final, cum

    @torch.no_grad()
    def predict_beam(self, encoder_out, max_pred_len, entities, facts=None, beam_size=5, return_all=False):
        """Beam-search decode (north_star cfg5: beam 5, batch 32).  The reference decodes greedily only
        (geo-aware/eval.py:61,83), so beam > 1 has no reference output to pin against ("parity-unpinned"); the tests
        check it against a CPU beam search written to the same rules.  beam_size == 1 IS predict(): the pinned greedy path with
        its n-gram clean-up.  Hypotheses are scored by their summed log-probability (log_softmax over the V+K+F
        scores); an ended hypothesis keeps competing with its final score; the best of the beam is returned as
        LongTensor (max_pred_len, B), <pad> after <end>."""
        if beam_size == 1:
            return DecoderTransformer.predict(self, encoder_out, max_pred_len, entities, facts)
        encoder_out, entities, facts = self._prepare_inputs(encoder_out, entities, facts)
        entities = entities.contiguous()
        enc_tok, P = self._image_input(encoder_out)
        enc_tok = enc_tok.contiguous()
        FF = self.transformer_decoder.layers[0].linear1.out_features
        S_all = P + entities.shape[1] + (facts.shape[1] if facts is not None else 0)
        Vx = self.vocab_size + S_all - P
        if not (1 < beam_size <= 8) or not ops.decode_supported(self.emb_dim, self.num_heads, FF, S_all, max_pred_len) \
                or not ops.decode_beam_supported(Vx, beam_size):
            raise IckError("predict_beam needs 1 <= beam_size <= 8, beam_size^2 * ceil((V+K+F)/1024) <= 4096 and sizes "
                           "the fused decode kernels support")
        if self.use_hip_graphs:
            key = (tuple(enc_tok.shape), tuple(entities.shape), None if facts is None else tuple(facts.shape),
                   max_pred_len, beam_size) + self._enc_key(enc_tok)
            res = self._graphed("beam", key, lambda t, e, f: self._predict_beam_device(t, e, f, max_pred_len, beam_size),
                                [enc_tok, entities, facts])
        else:
            res = self._predict_beam_device(enc_tok, entities, facts, max_pred_len, beam_size)
        out = res[0].t().contiguous()
        return (out, res[1], res[2], res[3]) if return_all else out

    def _predict_device(self, enc_tok, entities, facts, max_pred_len):
        """Whole greedy decode on the device: every step's token choice, clean-up and stop flag are computed
        by kernels (no host round trip), so the loop can be captured as one hipGraph."""
        dev = enc_tok.device
        B = enc_tok.shape[0]
        d, V, K = self.emb_dim, self.vocab_size, entities.shape[1]
        P = enc_tok.shape[1] if enc_tok.dim() == 3 else enc_tok.shape[2] * enc_tok.shape[3]
        S_all = P + K + (facts.shape[1] if facts is not None else 0)
        FF = self.transformer_decoder.layers[0].linear1.out_features
        if self.fused_decode and ops.decode_supported(d, self.num_heads, FF, S_all, max_pred_len):
            return self._predict_fused(enc_tok, entities, facts, max_pred_len)
        ee, fe, kv, _, side = self._encode_context(enc_tok, entities, facts, None)
        side.join()
        S = kv.shape[3]
        pe = self.pos_encoder.pe.view(-1, d)
        nl = len(self.transformer_decoder.layers)
        qkv_cache = [torch.empty(B, 3, self.num_heads, max_pred_len, ops.DHP, device=dev, dtype=torch.float32)
                     for _ in range(nl)]
        output = torch.full((B, max_pred_len), self.word_map["<pad>"], dtype=torch.long, device=dev)
        hist = torch.zeros(B, max_pred_len, dtype=torch.int32, device=dev)
        finished = torch.zeros(B, dtype=torch.int32, device=dev)
        tok = torch.full((B, 1), self.word_map["<start>"], dtype=torch.long, device=dev)
        msk = torch.zeros(B, 1, dtype=torch.long, device=dev)
        cap_buf = torch.full((B, max_pred_len), self.word_map["<start>"], dtype=torch.long, device=dev)
        Fn = fe.shape[1] if fe is not None else 0
        scores = torch.empty(B, 1, V + K + Fn, device=dev, dtype=torch.float32)
        wemb = self.word_embedding.weight.detach()
        for i in range(max_pred_len):
            x = ops.caption_embed(tok, msk, wemb, ee, fe, pe, V, self.word_map["<pad>"], math.sqrt(d), pos0=i)
            for li, layer in enumerate(self.transformer_decoder.layers):
                x = self._decoder_layer(li, layer, x, kv, S, qkv_buf=qkv_cache[li], pos=i)
            eib = gate = None
            if self.has_facts:
                eib, gate = ops.context_indicators(cap_buf, facts, K, V, self._pred_wt(),
                                                   self.fc_predicate.bias.detach(), mode=1)
            self._score_head(x, ee, fe, eib, gate, out=scores)
            ops.greedy_select(scores.view(B, -1), output, hist, finished, tok.view(-1), msk.view(-1), i, V, K,
                              self.has_facts, self.word_map["<end>"])
            if self.has_facts and i + 1 < max_pred_len:
                cap_buf[:, i + 1] = tok.view(-1)
        return output

    @torch.no_grad()
    def predict(self, encoder_out, max_pred_len, entities, facts=None):
        """Greedy decode with the reference's semantics per caption (argmax, <end> stop, repeated
        n-gram clean-up, pointer masks), KV-cached: step i only projects position i.  Works for any
        batch size (B independent captions); returns LongTensor (max_pred_len, B), <pad> after <end>."""
        encoder_out, entities, facts = self._prepare_inputs(encoder_out, entities, facts)
        entities = entities.contiguous()
        enc_tok = self._image_input(encoder_out)[0].contiguous()
        if self.use_hip_graphs:
            key = (tuple(enc_tok.shape), tuple(entities.shape), None if facts is None else tuple(facts.shape),
                   max_pred_len) + self._enc_key(enc_tok)
            output = self._graphed("greedy", key, lambda t, e, f: self._predict_device(t, e, f, max_pred_len),
                                   [enc_tok, entities, facts])
        else:
            output = self._predict_device(enc_tok, entities, facts, max_pred_len)
        return output.t().contiguous()
