"""
Seeded synthetic workloads for the caption-decoder hot path.

Shapes and value ranges follow the reference's preprocessing so that synthetic batches look
like what datasets.CaptionDataset yields (tuple order: geo-aware/datasets.py:54):
  * entity rows: geo-aware/create_input_files.py:160 (dist U(0,1), azimuth U(-179,179),
    size U(0,0.1), type randint[0,500]); knowledge-aware/create_input_files.py:174 (dist U(0,10));
    news-knowledge-aware/create_input_files.py:170-175 (count, in_headline, in_first_paragraph,
    type, five name-word indices).
  * fact rows (idx, subject, predicate), last row = <unk_fact> on <unk_ent>:
    knowledge-aware/create_input_files.py:183-188.
  * word map: <pad>=0, words, <unk>, <start>, <end> (geo-aware/create_input_files.py:311-315).
Parameters use the reference's state_dict names and are drawn from a
per-tensor seeded generator so a fixture only has to store the seed.
"""
import math
import zlib
from collections import OrderedDict

import torch

VARIANTS = ("geo", "knowledge", "news")
NUM_PREDICATES = {"geo": 0, "knowledge": 3000, "news": 3500}
NUM_TYPES = {"geo": 1000, "knowledge": 1000, "news": 20}
TYPE_OFFSET = {"geo": 4, "knowledge": 6, "news": 5}
ENT_COLS = {"geo": 5, "knowledge": 5, "news": 10}


def make_word_map(V):
    wm = {"<pad>": 0}
    for i in range(1, V - 3):
        wm["w%d" % i] = i
    wm["<unk>"] = V - 3
    wm["<start>"] = V - 2
    wm["<end>"] = V - 1
    assert len(wm) == V
    return wm


def param_shapes(variant, V, d=300, decoder_dim=512, encoder_dim=512, num_layers=3):
    s = OrderedDict()

    def attn(pre):
        s[pre + ".in_proj_weight"] = (3 * d, d)
        s[pre + ".in_proj_bias"] = (3 * d,)
        s[pre + ".out_proj.weight"] = (d, d)
        s[pre + ".out_proj.bias"] = (d,)

    def ffn_norms(pre, ff, n_norm):
        s[pre + ".linear1.weight"] = (ff, d)
        s[pre + ".linear1.bias"] = (ff,)
        s[pre + ".linear2.weight"] = (d, ff)
        s[pre + ".linear2.bias"] = (d,)
        for i in range(1, n_norm + 1):
            s[pre + ".norm%d.weight" % i] = (d,)
            s[pre + ".norm%d.bias" % i] = (d,)

    for i in range(num_layers):
        pre = "transformer_decoder.layers.%d" % i
        attn(pre + ".self_attn")
        attn(pre + ".multihead_attn")
        ffn_norms(pre, decoder_dim, 3)
    stacks = ["transformer_encoder_entities"] + (["transformer_encoder_facts"] if variant != "geo" else [])
    for st in stacks:
        for i in range(num_layers):
            pre = "%s.layers.%d" % (st, i)
            attn(pre + ".self_attn")
            ffn_norms(pre, encoder_dim, 2)
    s["word_embedding.weight"] = (V, d)
    s["entity_encoder.type_embedding.weight"] = (NUM_TYPES[variant], d - TYPE_OFFSET[variant])
    if variant != "geo":
        s["predicate_embedding.weight"] = (NUM_PREDICATES[variant], d)
    s["fc_vocab.weight"] = (V, d)
    s["fc_vocab.bias"] = (V,)
    s["fc_entity.weight"] = (1, d)
    s["fc_entity.bias"] = (1,)
    if variant != "geo":
        s["fc_fact.weight"] = (1, d)
        s["fc_fact.bias"] = (1,)
        s["fc_predicate.weight"] = (d, NUM_PREDICATES[variant])
        s["fc_predicate.bias"] = (d,)
    return s


def _gen(name, seed):
    g = torch.Generator()
    g.manual_seed((zlib.crc32(name.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)
    return g


def make_params(variant, V, seed=0, **kw):
    """Deterministic non-trivial parameters (biases and LayerNorm affine terms are non-zero
    on purpose so every term of every kernel is exercised)."""
    P = OrderedDict()
    for name, shape in param_shapes(variant, V, **kw).items():
        g = _gen(name, seed)
        if name.endswith("weight") and len(shape) == 2 and "embedding" not in name:
            a = 1.0 / math.sqrt(shape[1]) if "fc_predicate" not in name else 0.1
            if name.startswith("fc_"):
                a = 0.1  # init_weights(): U(-0.1, 0.1), geo-aware/models.py:264-272
            t = (torch.rand(shape, generator=g) * 2 - 1) * a
        elif "embedding" in name:
            t = (torch.rand(shape, generator=g) * 2 - 1) * 0.1
        elif ".norm" in name and name.endswith("weight"):
            t = 1.0 + (torch.rand(shape, generator=g) * 2 - 1) * 0.1
        else:  # biases
            t = (torch.rand(shape, generator=g) * 2 - 1) * 0.05
        P[name] = t
    if variant != "geo":
        # one shared Embedding object in the reference (knowledge-aware/models.py:330-331)
        P["fact_encoder.predicate_embedding.weight"] = P["predicate_embedding.weight"]
    return P


def make_conv1(seed=0, emb_dim=300, encoder_dim=2048):
    """Encoder.conv1 parameters, nn.Conv2d default-init ranges (geo-aware/models.py:32)."""
    a = 1.0 / math.sqrt(encoder_dim)
    w = (torch.rand((emb_dim, encoder_dim, 1, 1), generator=_gen("conv1.weight", seed)) * 2 - 1) * a
    b = (torch.rand((emb_dim,), generator=_gen("conv1.bias", seed)) * 2 - 1) * a
    return w, b


def make_feats(B, seed=0, encoder_dim=2048, size=14):
    """Post-ReLU ResNet-like features (B, 2048, 14, 14)."""
    g = _gen("feats", seed)
    return torch.randn((B, encoder_dim, size, size), generator=g).relu_()


def make_enc_out(B, seed=0, emb_dim=300, P=196):
    """A stand-in for Encoder output (B, emb_dim, 196) when the feature projection is not under test."""
    g = _gen("enc_out", seed)
    return torch.randn((B, emb_dim, P), generator=g) * 0.5


def make_entities(variant, B, K, V, seed=0):
    g = _gen("entities", seed)
    e = torch.zeros(B, K, ENT_COLS[variant])
    e[:, :, 0] = torch.arange(K).float()
    if variant == "news":
        e[:, :, 1] = torch.randint(0, 6, (B, K), generator=g).float()
        e[:, :, 2] = torch.randint(0, 2, (B, K), generator=g).float()
        e[:, :, 3] = torch.randint(0, 2, (B, K), generator=g).float()
        e[:, :, 4] = torch.randint(0, 20, (B, K), generator=g).float()
        e[:, :, 5:] = torch.randint(0, V, (B, K, 5), generator=g).float()
        # short names are padded with <pad>=0
        e[:, :, 8:] *= (torch.rand((B, K, 2), generator=g) > 0.5).float()
    else:
        hi = 1.0 if variant == "geo" else 10.0
        e[:, :, 1] = torch.rand((B, K), generator=g) * hi
        e[:, :, 2] = torch.rand((B, K), generator=g) * 358.0 - 179.0
        e[:, :, 3] = torch.rand((B, K), generator=g) * 0.1
        e[:, :, 4] = torch.randint(0, 501, (B, K), generator=g).float()
    return e


def make_facts(variant, B, F, K, seed=0):
    g = _gen("facts", seed)
    f = torch.zeros(B, F, 3, dtype=torch.long)
    f[:, :, 0] = torch.arange(F)
    f[:, :, 1] = torch.randint(0, K, (B, F), generator=g)
    f[:, :, 2] = torch.randint(0, NUM_PREDICATES[variant], (B, F), generator=g)
    f[:, F - 1, 1] = K - 1
    f[:, F - 1, 2] = 0
    return f


def make_captions(variant, B, L, K, F, V, seed=0, min_len=5):
    """(captions (B,L) int64, masks (B,L) int64, lengths (B,1) int64)."""
    g = _gen("captions", seed)
    lo = min(min_len, L)
    lengths = torch.randint(lo, L + 1, (B, 1), generator=g)
    tok = torch.randint(1, max(2, V - 4), (B, L), generator=g)
    r = torch.rand((B, L), generator=g)
    ent = V + torch.randint(0, K, (B, L), generator=g)
    masks = torch.zeros(B, L, dtype=torch.long)
    is_ent = r < 0.10
    tok = torch.where(is_ent, ent, tok)
    masks[is_ent] = 1
    if variant != "geo" and F > 0:
        fact = V + K + torch.randint(0, F, (B, L), generator=g)
        is_fact = (r >= 0.10) & (r < 0.15)
        tok = torch.where(is_fact, fact, tok)
        masks[is_fact] = 2
    pos = torch.arange(L).view(1, L)
    tok[:, 0] = V - 2  # <start>
    masks[:, 0] = 0
    end_pos = lengths - 1
    tok = torch.where(pos == end_pos, torch.full_like(tok, V - 1), tok)  # <end>
    tok = torch.where(pos > end_pos, torch.zeros_like(tok), tok)  # <pad>
    masks = torch.where(pos >= end_pos, torch.zeros_like(masks), masks)
    return tok, masks, lengths


def make_batch(variant, B, L, K, V, F=0, seed=0):
    caps, masks, lens = make_captions(variant, B, L, K, F, V, seed)
    out = dict(captions=caps, caption_masks=masks, caption_lengths=lens,
               entities=make_entities(variant, B, K, V, seed))
    if variant != "geo":
        out["facts"] = make_facts(variant, B, F, K, seed)
    return out


# BASELINE.json configs
CONFIGS = {
    "cfg1": dict(variant="geo", B=4, L=3, K=6, V=5000, F=0),
    "cfg2": dict(variant="geo", B=64, L=20, K=20, V=10000, F=0),
    "cfg4": dict(variant="knowledge", B=64, L=20, K=20, V=50000, F=51),
    "cfg5": dict(variant="geo", B=32, L=20, K=20, V=10000, F=0),
}


def write_dataset(data_dir, data_name, variant, n_train=24, n_val=8, n_test=4, L=12, K=6, V=60, F=5, seed=0):
    """Write a small synthetic dataset in the reference's on-disk formats (datasets.CaptionDataset): JSON
    captions / lengths / masks, pickled entity (and fact) features and integer-encoded names, a WORDMAP, and
    precomputed feature maps as <SPLIT>_FEATURES_<name>.npy."""
    import json
    import os
    import pickle

    import numpy as np

    from .utils import str_to_int
    os.makedirs(data_dir, exist_ok=True)
    wm = make_word_map(V)
    with open(os.path.join(data_dir, "WORDMAP_%s.json" % data_name), "w") as f:
        json.dump(wm, f)
    for split, n, sd in (("TRAIN", n_train, seed), ("VAL", n_val, seed + 1), ("TEST", n_test, seed + 2)):
        b = make_batch(variant, n, L, K, V, F, sd)

        def out(kind, ext):
            return os.path.join(data_dir, "%s_%s_%s.%s" % (split, kind, data_name, ext))

        with open(out("CAPTIONS", "json"), "w") as f:
            json.dump(b["captions"].tolist(), f)
        with open(out("CAPLENS", "json"), "w") as f:
            json.dump(b["caption_lengths"].view(-1).tolist(), f)
        with open(out("CAPMASKS", "json"), "w") as f:
            json.dump(b["caption_masks"].tolist(), f)
        with open(out("ENT_FEATURES", "pkl"), "wb") as f:
            pickle.dump(b["entities"].tolist(), f)
        names = [[[k, len("ent%d" % k)] + str_to_int("ent%d" % k) for k in range(K)] for _ in range(n)]
        with open(out("ENT_NAMES", "pkl"), "wb") as f:
            pickle.dump(names, f)
        if variant != "geo":
            with open(out("FACTS", "pkl"), "wb") as f:
                pickle.dump(b["facts"].tolist(), f)
            fn = [[[j, len("obj%d" % j)] + str_to_int("obj%d" % j) for j in range(F)] for _ in range(n)]
            with open(out("FACT_NAMES", "pkl"), "wb") as f:
                pickle.dump(fn, f)
        np.save(out("FEATURES", "npy"), make_feats(n, sd).numpy().astype(np.float16))
    return wm
