"""
Generates tests/golden/*.npz by running the REAL reference (imported from /root/reference,
authoring container only, CPU) on seeded synthetic inputs.  The fixtures hold data only:
seeds/shapes, a few input tensors, and the reference's outputs.  Parameters and inputs are
regenerated at test time from the same seeds by ick_amd.synth (a checksum guards drift).

    python tests/golden/make_fixtures.py

The reference's models.py does `import torchvision` (absent here) although only
Encoder.__init__ uses it; an empty module object is registered under that name so the
decoder classes import.  models.Encoder is never constructed.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F
from torch.nn.utils.rnn import pack_padded_sequence

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import ick_amd.synth as synth  # noqa: E402

REF_DIRS = {"geo": "geo-aware", "knowledge": "knowledge-aware", "news": "news-knowledge-aware"}


def load_reference(variant):
    sys.modules.setdefault("torchvision", types.ModuleType("torchvision"))
    path = os.path.join("/root/reference", REF_DIRS[variant], "models.py")
    spec = importlib.util.spec_from_file_location("ref_models_" + variant, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def build_reference_decoder(variant, V, seed):
    ref = load_reference(variant)
    wm = synth.make_word_map(V)
    dec = ref.DecoderTransformer(word_map=wm, emb_dim=300, decoder_dim=512, encoder_dim=512,
                                 num_heads=10, num_layers=3)
    P = synth.make_params(variant, V, seed)
    missing, unexpected = dec.load_state_dict(P, strict=False)
    assert missing == ["pos_encoder.pe"] and not unexpected, (missing, unexpected)
    dec.eval()
    return dec, P, wm


def checksum(P):
    return np.array([float(sum(v.double().abs().sum() for v in P.values()))])


GRAD_KEYS = [
    "fc_vocab.weight", "fc_vocab.bias", "fc_entity.weight", "fc_entity.bias", "word_embedding.weight",
    "transformer_decoder.layers.0.self_attn.in_proj_bias",
    "transformer_decoder.layers.2.multihead_attn.in_proj_bias",
    "transformer_decoder.layers.1.norm2.weight",
    "transformer_decoder.layers.2.linear1.bias",
    "transformer_encoder_entities.layers.0.linear1.bias",
    "transformer_encoder_entities.layers.2.norm2.bias",
    "fc_fact.weight", "fc_predicate.bias", "transformer_encoder_facts.layers.1.self_attn.out_proj.bias",
]


def forward_case(name, variant, B, L, K, V, Fn, seed, with_stages, with_grads, with_enc_grad=False):
    dec, P, wm = build_reference_decoder(variant, V, seed)
    batch = synth.make_batch(variant, B, L, K, V, Fn, seed)
    enc_out = synth.make_enc_out(B, seed)
    if with_enc_grad:      # fine_tune_encoder=True: the loss gradient reaches Encoder.conv1 through encoder_out
        enc_out.requires_grad_(True)
    stages = {}

    def hook(key):
        def fn(mod, inp, out):
            stages[key] = out.detach().clone()
        return fn

    hs = [dec.entity_encoder.register_forward_hook(hook("entities_encoded")),
          dec.caption_embedder.register_forward_hook(hook("embeddings")),
          dec.transformer_encoder_entities.register_forward_hook(hook("entity_context")),
          dec.transformer_decoder.register_forward_hook(hook("h"))]
    if variant != "geo":
        hs.append(dec.fact_encoder.register_forward_hook(hook("facts_encoded")))
        hs.append(dec.transformer_encoder_facts.register_forward_hook(hook("fact_context")))
    args = [batch["captions"], enc_out, batch["caption_masks"], batch["caption_lengths"], batch["entities"]]
    if variant != "geo":
        args.append(batch["facts"])
    for p in dec.parameters():
        p.requires_grad_(True)
    scores, caps_sorted, dl = dec(*args)
    for h in hs:
        h.remove()
    out = dict(variant=variant, B=B, L=L, K=K, V=V, F=Fn, seed=seed, param_checksum=checksum(P),
               scores=scores.detach().numpy(), captions_sorted=caps_sorted.numpy(),
               decode_lengths=np.array(dl))
    if with_stages:
        out["entities_encoded"] = stages["entities_encoded"].numpy()
        out["embeddings"] = stages["embeddings"].numpy()  # (B,L,d), sorted order
        out["entity_context"] = stages["entity_context"].permute(1, 0, 2).numpy()  # (B,K,d)
        out["h"] = stages["h"].permute(1, 0, 2).numpy()  # (B,L,d)
        if variant != "geo":
            out["facts_encoded"] = stages["facts_encoded"].numpy()
            out["fact_context"] = stages["fact_context"].permute(1, 0, 2).numpy()
    if with_grads:
        # loss exactly as geo-aware/train.py:275-281 (eval mode => dropout off)
        targets = caps_sorted[:, 1:]
        sp = pack_padded_sequence(scores, dl, batch_first=True).data
        tp = pack_padded_sequence(targets, dl, batch_first=True).data
        loss = F.cross_entropy(sp, tp, ignore_index=wm["<pad>"])
        loss.backward()
        out["loss"] = np.array([loss.item()])
        names = dict(dec.named_parameters())
        norms = {}
        for k, p in names.items():
            if p.grad is not None:
                norms[k] = float(p.grad.double().norm())
        out["grad_norm_names"] = np.array(sorted(norms))
        out["grad_norms"] = np.array([norms[k] for k in sorted(norms)])
        for k in GRAD_KEYS:
            if k in names and names[k].grad is not None:
                out["grad::" + k] = names[k].grad.numpy()
        if with_enc_grad:
            out["encoder_out_grad"] = enc_out.grad.numpy()       # (B, d, 196), caller's batch order
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, "scores", scores.shape, "loss" if with_grads else "", out.get("loss"))


def conv1_case(name, B, seed):
    """Row a1: the hot-path part of Encoder is nn.Conv2d(2048,300,1) + view
    (geo-aware/models.py:32,45-46); Encoder itself cannot be built offline."""
    w, b = synth.make_conv1(seed)
    feats = synth.make_feats(B, seed)
    conv = torch.nn.Conv2d(2048, 300, 1)
    with torch.no_grad():
        conv.weight.copy_(w)
        conv.bias.copy_(b)
        out = conv(feats).view(B, 300, -1)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), B=B, seed=seed, out=out.numpy())
    print("wrote", name, out.shape)


def predict_case(name, variant, K, V, Fn, seeds, max_len):
    res = {}
    n_clean = 0
    for seed in seeds:
        dec, P, wm = build_reference_decoder(variant, V, seed)
        ents = synth.make_entities(variant, 1, K, V, seed)
        enc_out = synth.make_enc_out(1, seed)
        args = [enc_out, max_len, ents]
        if variant != "geo":
            args.append(synth.make_facts(variant, 1, Fn, K, seed))
        with torch.no_grad():
            seq = dec.predict(*args)
        res["seq_%d" % seed] = seq.numpy()
        res["cksum_%d" % seed] = checksum(P)
        print(name, seed, seq.view(-1).tolist())
    np.savez_compressed(os.path.join(HERE, name + ".npz"), variant=variant, K=K, V=V, F=Fn,
                        seeds=np.array(seeds), max_len=max_len, **res)


def round2_cases():
    """Added in round 2 (the round-1 fixtures are left untouched): a mid-size news case and the gradient that
    fine_tune_encoder=True sends into the image rows (geo-aware/train.py:93-100,282-294)."""
    forward_case("fwd_mid_news", "news", B=6, L=14, K=21, V=400, Fn=31, seed=13, with_stages=True, with_grads=True)
    forward_case("fwd_encgrad_geo", "geo", B=3, L=9, K=7, V=120, Fn=0, seed=17, with_stages=False, with_grads=True,
                 with_enc_grad=True)
    forward_case("fwd_encgrad_knowledge", "knowledge", B=2, L=8, K=6, V=90, Fn=7, seed=19, with_stages=False,
                 with_grads=True, with_enc_grad=True)


def digest_columns(Vx, V, n=128, seed=0):
    """The score columns a big-vocabulary digest keeps: n seeded vocabulary columns + every pointer column."""
    g = torch.Generator().manual_seed(seed)
    cols = torch.randperm(V, generator=g)[:n].sort().values
    return torch.cat([cols, torch.arange(V, Vx)])


def big_vocab_digest(name, variant, B, L, K, V, Fn, seed, conv1=False):
    """The reference's forward at a vocabulary too wide to store whole (cfg4: 64 x 20 x 50 071 floats = 256 MB): the
    fixture keeps 128 sampled vocabulary columns, every pointer column, and per-row logsumexp / argmax / max over ALL
    columns (any wrong vocabulary column moves the logsumexp or the argmax), plus the loss of train.py's criterion."""
    dec, P, wm = build_reference_decoder(variant, V, seed)
    batch = synth.make_batch(variant, B, L, K, V, Fn, seed)
    extra = {}
    if conv1:
        # the whole bench step's input: 14 x 14 x 2048 features through the stock nn.Conv2d(2048, 300, 1) + view that is
        # the hot-path part of the reference's Encoder (geo-aware/models.py:32,45-46; Encoder itself cannot be built
        # offline), then the reference's DecoderTransformer.forward on its output
        w, b = synth.make_conv1(seed)
        conv = torch.nn.Conv2d(2048, 300, 1)
        with torch.no_grad():
            conv.weight.copy_(w)
            conv.bias.copy_(b)
            enc_out = conv(synth.make_feats(B, seed)).view(B, 300, -1)
        g = torch.Generator().manual_seed(seed)
        pos = torch.randperm(196, generator=g)[:8].sort().values
        extra = dict(conv1=1, enc_pos=pos.numpy(), enc_out_pos=enc_out[:, :, pos].numpy(),
                     enc_out_sum=enc_out.double().sum(dim=(1, 2)).numpy())
    else:
        enc_out = synth.make_enc_out(B, seed)
    args = [batch["captions"], enc_out, batch["caption_masks"], batch["caption_lengths"], batch["entities"]]
    if variant != "geo":
        args.append(batch["facts"])
    with torch.no_grad():
        scores, caps_sorted, dl = dec(*args)
        targets = caps_sorted[:, 1:]
        sp = pack_padded_sequence(scores, dl, batch_first=True).data
        tp = pack_padded_sequence(targets, dl, batch_first=True).data
        loss = F.cross_entropy(sp, tp, ignore_index=wm["<pad>"])
    Vx = scores.shape[2]
    cols = digest_columns(Vx, V)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), variant=variant, B=B, L=L, K=K, V=V, F=Fn, seed=seed,
                        param_checksum=checksum(P), cols=cols.numpy(), scores_cols=scores[:, :, cols].numpy(),
                        lse=scores.double().logsumexp(dim=2).numpy(), argmax=scores.argmax(dim=2).numpy(),
                        rowmax=scores.max(dim=2).values.numpy(), captions_sorted=caps_sorted.numpy(),
                        decode_lengths=np.array(dl), loss=np.array([loss.item()]), **extra)
    print("wrote", name, tuple(scores.shape), "loss", loss.item())


def score_head_case(name, variant, B, L, K, V, Fn, seed):
    """The public score-head methods called directly on the real reference (geo-aware/models.py:291-313,
    knowledge-aware/models.py:380-455): get_context_indicators for the teacher-forced form (out_length = L), a
    shorter out_length and predict()'s out_length = 1; get_scores on seeded h / encoded contexts."""
    dec, P, wm = build_reference_decoder(variant, V, seed)
    ref = sys.modules[type(dec).__module__] if type(dec).__module__ in sys.modules else None
    batch = synth.make_batch(variant, B, L, K, V, Fn, seed)
    g = torch.Generator().manual_seed(seed + 1000)
    h = torch.randn(L, B, 300, generator=g)
    ee = torch.randn(B, K, 300, generator=g) * 0.3
    out = dict(variant=variant, B=B, L=L, K=K, V=V, F=Fn, seed=seed, param_checksum=checksum(P), h=h.numpy(),
               ee=ee.numpy())
    with torch.no_grad():
        if variant == "geo":
            out["scores"] = dec.get_scores(h, ee).numpy()
        else:
            fe = torch.randn(B, Fn, 300, generator=g) * 0.3
            out["fe"] = fe.numpy()
            caps, facts = batch["captions"], batch["facts"].clone()
            facts[:, :, 2] %= 3        # many facts share a predicate: the indicator must count a predicate once
            for tag, ol in (("full", L), ("short", L - 3), ("one", 1)):
                eib, pi = dec.get_context_indicators(caps, facts, K, ol)
                out["eib_" + tag] = eib.numpy().astype(np.uint8)
                out["pi_" + tag] = np.packbits(pi.numpy().astype(np.uint8), axis=2)      # (B, ol, ceil(NP/8), 1)
            eib, pi = dec.get_context_indicators(caps, facts, K, L)
            out["scores"] = dec.get_scores(h, ee, fe, eib, pi).numpy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, out["scores"].shape)


def round3_cases():
    """Added in round 3: the public get_scores / get_context_indicators methods, and digests of the reference's
    forward at the vocabulary sizes the bench quotes (cfg4 at its full batch, the news variant at V = 50 000)."""
    score_head_case("score_head_geo", "geo", B=3, L=6, K=7, V=80, Fn=0, seed=23)
    score_head_case("score_head_knowledge", "knowledge", B=4, L=12, K=6, V=70, Fn=8, seed=32)
    score_head_case("score_head_news", "news", B=3, L=12, K=5, V=60, Fn=7, seed=30)
    c = synth.CONFIGS["cfg4"]
    big_vocab_digest("digest_cfg4_b64", c["variant"], c["B"], c["L"], c["K"], c["V"], c["F"], seed=41)
    big_vocab_digest("digest_news_v50k", "news", B=8, L=20, K=51, V=50000, Fn=51, seed=42)


def round4_cases():
    """Added in round 4: the headline workload itself -- cfg2 at its bench batch (B = 64, L = 20,
    K = 20, V = 10 000), features -> stock conv1 -> the real reference's forward -- in the digest layout of cfg4."""
    c = synth.CONFIGS["cfg2"]
    big_vocab_digest("digest_cfg2_b64", c["variant"], c["B"], c["L"], c["K"], c["V"], c["F"], seed=44, conv1=True)


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(4)
    if "--round4" in sys.argv:
        round4_cases()
        sys.exit(0)
    if "--round3" in sys.argv:
        round3_cases()
        sys.exit(0)
    if "--round2" in sys.argv:
        round2_cases()
        sys.exit(0)
    for v in synth.VARIANTS:
        Fn = 0 if v == "geo" else 5
        forward_case("fwd_tiny_" + v, v, B=4, L=7, K=6, V=50, Fn=Fn, seed=11, with_stages=True, with_grads=True)
    forward_case("fwd_cfg1_geo", "geo", B=4, L=3, K=6, V=5000, Fn=0, seed=1, with_stages=False, with_grads=False)
    forward_case("fwd_mid_geo", "geo", B=8, L=20, K=20, V=1000, Fn=0, seed=2, with_stages=False, with_grads=True)
    forward_case("fwd_mid_knowledge", "knowledge", B=6, L=12, K=9, V=300, Fn=11, seed=3, with_stages=False,
                 with_grads=True)
    conv1_case("conv1_b2", B=2, seed=5)
    predict_case("predict_geo", "geo", K=6, V=50, Fn=0, seeds=[0, 1, 2, 3, 4, 5, 6, 7], max_len=12)
    predict_case("predict_knowledge", "knowledge", K=6, V=50, Fn=5, seeds=[0, 1, 2, 3], max_len=12)
    predict_case("predict_news", "news", K=6, V=50, Fn=5, seeds=[0, 1, 2], max_len=10)
