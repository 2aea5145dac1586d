"""The oracle (oracle/restatement.py) against outputs of the real reference
(tests/golden/*.npz, made by tests/golden/make_fixtures.py)."""
import numpy as np
import pytest
import torch

from helpers import case_from_golden, load_golden, t
import ick_amd.synth as synth
from oracle import restatement as R

FWD = ["fwd_tiny_geo", "fwd_tiny_knowledge", "fwd_tiny_news", "fwd_cfg1_geo", "fwd_mid_geo", "fwd_mid_knowledge",
       "fwd_mid_news", "fwd_encgrad_geo", "fwd_encgrad_knowledge"]
TOL = 2e-5  # fp32, different summation order only


@pytest.mark.parametrize("name", FWD)
def test_forward_scores(name):
    g = load_golden(name)
    cfg, P, wm, batch, enc_out = case_from_golden(g)
    stages = {}
    with torch.no_grad():
        scores, caps, dl = R.forward(cfg, P, batch["captions"], enc_out, batch["caption_masks"],
                                     batch["caption_lengths"], batch["entities"], batch.get("facts"), stages)
    assert dl == g["decode_lengths"].tolist()
    assert torch.equal(caps, t(g["captions_sorted"]))
    ref = t(g["scores"])
    assert scores.shape == ref.shape
    assert (scores - ref).abs().max().item() < TOL
    assert torch.equal(scores.argmax(-1), ref.argmax(-1))
    if "entities_encoded" in g:
        # integer-exact stages: gathers and feature slots
        assert torch.equal(stages["entities_encoded"], t(g["entities_encoded"]))
        assert torch.equal(stages["embeddings"], t(g["embeddings"]))
        K = int(g["K"])
        assert (stages["memory"][:, 196:196 + K] - t(g["entity_context"])).abs().max() < TOL
        assert (stages["h"] - t(g["h"])).abs().max() < TOL
        if "facts_encoded" in g:
            assert torch.equal(stages["facts_encoded"], t(g["facts_encoded"]))
            assert (stages["memory"][:, 196 + K:] - t(g["fact_context"])).abs().max() < TOL


@pytest.mark.parametrize("name", ["fwd_tiny_geo", "fwd_tiny_knowledge", "fwd_tiny_news", "fwd_mid_geo",
                                  "fwd_mid_knowledge", "fwd_mid_news", "fwd_encgrad_geo", "fwd_encgrad_knowledge"])
def test_loss_and_grads(name):
    g = load_golden(name)
    cfg, P, wm, batch, enc_out = case_from_golden(g)
    if "encoder_out_grad" in g:
        enc_out.requires_grad_(True)
    P = {k: v.clone().requires_grad_(True) for k, v in P.items() if not k.startswith("fact_encoder.")}
    if cfg.has_facts:
        P["fact_encoder.predicate_embedding.weight"] = P["predicate_embedding.weight"]
    scores, caps, dl = R.forward(cfg, P, batch["captions"], enc_out, batch["caption_masks"],
                                 batch["caption_lengths"], batch["entities"], batch.get("facts"))
    loss = R.packed_ce_loss(cfg, scores, caps, dl)
    assert abs(loss.item() - float(g["loss"][0])) < 1e-5
    loss.backward()
    norms = dict(zip(g["grad_norm_names"].tolist(), g["grad_norms"].tolist()))
    for k, n in norms.items():
        if k.startswith("fact_encoder."):
            continue
        assert P[k].grad is not None, k
        mine = float(P[k].grad.double().norm())
        assert abs(mine - n) <= 1e-4 * max(n, 1e-3), (k, mine, n)
    for k in g:
        if k.startswith("grad::"):
            ref = t(g[k])
            assert (P[k[6:]].grad - ref).abs().max().item() < 1e-5 * max(1.0, ref.abs().max().item()), k
    if "encoder_out_grad" in g:      # what fine_tune_encoder=True back-propagates into Encoder.conv1
        ref = t(g["encoder_out_grad"])
        assert (enc_out.grad - ref).abs().max().item() < 1e-5 * max(1.0, ref.abs().max().item())


def test_conv1():
    g = load_golden("conv1_b2")
    B, seed = int(g["B"]), int(g["seed"])
    w, b = synth.make_conv1(seed)
    out = R.feat_proj(synth.make_feats(B, seed), w, b)
    assert out.shape == (B, 300, 196)
    assert (out - t(g["out"])).abs().max().item() < 5e-5


@pytest.mark.parametrize("name", ["predict_geo", "predict_knowledge", "predict_news"])
def test_predict_tokens(name):
    g = load_golden(name)
    variant = str(g["variant"])
    K, V, Fn, max_len = int(g["K"]), int(g["V"]), int(g["F"]), int(g["max_len"])
    wm = synth.make_word_map(V)
    cfg = R.config_from_word_map(variant, wm)
    for seed in g["seeds"].tolist():
        P = synth.make_params(variant, V, seed)
        ents = synth.make_entities(variant, 1, K, V, seed)
        facts = synth.make_facts(variant, 1, Fn, K, seed) if variant != "geo" else None
        with torch.no_grad():
            seq = R.predict(cfg, P, synth.make_enc_out(1, seed), max_len, ents, facts)
        assert seq.view(-1).tolist() == g["seq_%d" % seed].reshape(-1).tolist(), (name, seed)


def test_loop_cleanup_cases():
    # hand-checked cases of geo-aware/models.py:421-435
    out = [5, 5]; R.loop_cleanup(out, [90, 91], 1)
    assert out == [5, 91]                       # immediate repeat -> runner-up of the last step
    out = [1, 2, 1, 2]; R.loop_cleanup(out, [90, 91, 92, 93], 3)
    assert out == [1, 2, 92, 93]                # bigram repeat -> last two rewritten
    out = [1, 2, 3, 1, 2, 3]; R.loop_cleanup(out, [90, 91, 92, 93, 94, 95], 5)
    assert out == [1, 2, 92, 93, 94, 95]        # trigram repeat rewrites FOUR positions (as the reference does)
    out = [1, 2, 3, 4]; R.loop_cleanup(out, [90, 91, 92, 93], 3)
    assert out == [1, 2, 3, 4]


@pytest.mark.parametrize("name", ["fwd_tiny_geo", "fwd_tiny_knowledge", "fwd_tiny_news", "fwd_mid_geo"])
def test_stock_module_port_matches_reference(name):
    """oracle/stock.py (the timed CPU baseline of bench.py) against the reference's outputs."""
    from oracle.stock import StockDecoder
    g = load_golden(name)
    cfg, P, wm, batch, enc_out = case_from_golden(g)
    m = StockDecoder(cfg.variant, wm).load_reference_params(P).eval()
    with torch.no_grad():
        scores, caps, dl = m(batch["captions"], enc_out, batch["caption_masks"], batch["caption_lengths"],
                             batch["entities"], batch.get("facts"))
    assert dl == g["decode_lengths"].tolist()
    assert (scores - t(g["scores"])).abs().max().item() < TOL


@pytest.mark.parametrize("name", ["predict_geo", "predict_knowledge"])
def test_stock_module_predict_matches_reference(name):
    """oracle/stock.py's full-recompute greedy decode (bench.py's CPU baseline of the greedy mode)."""
    from oracle.stock import StockDecoder
    g = load_golden(name)
    variant = str(g["variant"])
    K, V, Fn, max_len = int(g["K"]), int(g["V"]), int(g["F"]), int(g["max_len"])
    wm = synth.make_word_map(V)
    for seed in g["seeds"].tolist()[:4]:
        m = StockDecoder(variant, wm).load_reference_params(synth.make_params(variant, V, seed)).eval()
        ents = synth.make_entities(variant, 1, K, V, seed)
        facts = synth.make_facts(variant, 1, Fn, K, seed) if variant != "geo" else None
        seq = m.predict(synth.make_enc_out(1, seed), max_len, ents, facts)
        assert seq.view(-1).tolist() == g["seq_%d" % seed].reshape(-1).tolist(), (name, seed)


# ------------------------------------------------------------------------------------------------ round 3 fixtures
def check_digest(g, scores, tol=TOL):
    """A big-vocabulary digest of the real reference (make_fixtures.big_vocab_digest) against full scores."""
    cols = t(g["cols"])
    assert (scores[:, :, cols].cpu() - t(g["scores_cols"])).abs().max().item() < tol
    assert (scores.double().logsumexp(dim=2).cpu() - t(g["lse"])).abs().max().item() < tol
    assert (scores.max(dim=2).values.cpu() - t(g["rowmax"])).abs().max().item() < tol
    am = scores.argmax(dim=2).cpu()
    ref_am = t(g["argmax"])
    if not torch.equal(am, ref_am):      # a flipped arg-max must be a rounding-level tie
        bad = (am != ref_am).nonzero()
        for b, l in bad.tolist():
            assert abs(scores[b, l, am[b, l]].item() - scores[b, l, ref_am[b, l]].item()) < tol
        assert len(bad) <= 2


@pytest.mark.parametrize("name", ["digest_cfg4_b64", "digest_news_v50k"])
def test_big_vocab_digest(name):
    """The oracle at the sizes the bench quotes for cfg4 (B = 64, V = 50 000, F = 51) and for the news variant at
    V = 50 000, against digests of the real reference's forward."""
    g = load_golden(name)
    cfg, P, wm, batch, enc_out = case_from_golden(g)
    with torch.no_grad():
        scores, caps, dl = R.forward(cfg, P, batch["captions"], enc_out, batch["caption_masks"],
                                     batch["caption_lengths"], batch["entities"], batch.get("facts"))
        loss = R.packed_ce_loss(cfg, scores, caps, dl)
    assert dl == g["decode_lengths"].tolist() and torch.equal(caps, t(g["captions_sorted"]))
    check_digest(g, scores)
    assert abs(loss.item() - float(g["loss"][0])) < 1e-5


def test_cfg2_b64_digest_with_conv1():
    """Round 4: the headline workload (cfg2, B = 64, V = 10 000, K = 20) from the features on -- the oracle's feature
    projection against the stock nn.Conv2d's output rows the fixture keeps, then the oracle's forward against the digest
    of the real reference's forward on that output."""
    g = load_golden("digest_cfg2_b64")
    cfg, P, wm, batch, _ = case_from_golden(g)
    B, seed = int(g["B"]), int(g["seed"])
    assert (B, int(g["L"]), int(g["K"]), int(g["V"])) == (64, 20, 20, 10000) and int(g["conv1"]) == 1
    cw, cb = synth.make_conv1(seed)
    with torch.no_grad():
        enc_out = R.feat_proj(synth.make_feats(B, seed), cw, cb)
        assert (enc_out[:, :, t(g["enc_pos"])] - t(g["enc_out_pos"])).abs().max().item() < 2e-5
        assert (enc_out.double().sum(dim=(1, 2)) - t(g["enc_out_sum"])).abs().max().item() < 1e-2
        scores, caps, dl = R.forward(cfg, P, batch["captions"], enc_out, batch["caption_masks"],
                                     batch["caption_lengths"], batch["entities"])
        loss = R.packed_ce_loss(cfg, scores, caps, dl)
    assert dl == g["decode_lengths"].tolist() and torch.equal(caps, t(g["captions_sorted"]))
    check_digest(g, scores)
    assert abs(loss.item() - float(g["loss"][0])) < 1e-5


def unpack_pi(g, tag, num_pred):
    return t(np.unpackbits(g["pi_" + tag], axis=2)[:, :, :num_pred]).float()


@pytest.mark.parametrize("name", ["score_head_geo", "score_head_knowledge", "score_head_news"])
def test_score_head_methods(name):
    """get_context_indicators / get_scores called directly on the real reference."""
    g = load_golden(name)
    variant = str(g["variant"])
    B, L, K, V, Fn, seed = (int(g[k]) for k in ("B", "L", "K", "V", "F", "seed"))
    P = synth.make_params(variant, V, seed)
    cfg = R.config_from_word_map(variant, synth.make_word_map(V))
    h, ee = t(g["h"]).permute(1, 0, 2), t(g["ee"])
    ref = t(g["scores"]).permute(1, 0, 2)
    if variant == "geo":
        assert (R.get_scores(cfg, P, h, ee) - ref).abs().max().item() < TOL
        return
    batch = synth.make_batch(variant, B, L, K, V, Fn, seed)
    batch["facts"][:, :, 2] %= 3                  # as the fixture: many facts share a predicate
    for tag, ol in (("full", L), ("short", L - 3), ("one", 1)):
        eib, pi = R.context_indicators(cfg, batch["captions"], batch["facts"], K, ol)
        assert torch.equal(eib, t(g["eib_" + tag]).float().squeeze(3)), tag
        assert torch.equal(pi, unpack_pi(g, tag, cfg.num_predicates).squeeze(3)), tag
    eib, pi = R.context_indicators(cfg, batch["captions"], batch["facts"], K, L)
    assert eib.sum() > pi.sum() > 0              # the case exercises the indicators and shared predicates
    assert (R.get_scores(cfg, P, h, ee, t(g["fe"]), eib, pi) - ref).abs().max().item() < TOL
