"""Round-4 parity additions:
  * the headline workload from the features on -- cfg2 at its bench batch (B = 64, L = 20, K = 20, V = 10 000):
    Encoder.conv1 + DecoderTransformer.forward against a digest made by the stock nn.Conv2d + the REAL reference's
    forward (tests/golden/make_fixtures.py --round4), in every product mode of the large GEMM tiles, with the GEMM
    plans of that instantiation asserted;
  * the library's default product mode is what DESIGN.md says it is, and the mode is part of TrainStep's graph key.
"""
import pytest
import torch

import ick_amd
import ick_amd.ops as ops
import ick_amd.synth as synth
from helpers import case_from_golden, load_golden, t
from test_bench_sizes_gpu import make_encoder, plan_log, plans_of  # noqa: F401  (plan_log is a fixture)
from test_forward_gpu import build_decoder, run_forward
from test_oracle_golden import check_digest

pytestmark = pytest.mark.gpu


def test_cfg2_b64_from_features_vs_reference_digest(plan_log, gemm_split):
    g = load_golden("digest_cfg2_b64")
    cfg, P, wm, batch, _ = case_from_golden(g)
    B, seed = int(g["B"]), int(g["seed"])
    assert (B, int(g["L"]), int(g["K"]), int(g["V"])) == (64, 20, 20, 10000)
    enc, cw, cb = make_encoder(seed)
    dec = build_decoder(cfg.variant, cfg.vocab_size, P)
    enc_out = enc(synth.make_feats(B, seed).cuda())
    assert (enc_out[:, :, t(g["enc_pos"]).cuda()].cpu() - t(g["enc_out_pos"])).abs().max().item() < 2e-5
    assert (enc_out.double().sum(dim=(1, 2)).cpu() - t(g["enc_out_sum"])).abs().max().item() < 1e-2
    scores, caps, dl = run_forward(dec, batch, enc_out)
    assert dl == g["decode_lengths"].tolist() and torch.equal(caps.cpu(), t(g["captions_sorted"]))
    check_digest(g, scores.float().cpu(), tol=2e-4)
    # the replayed graph (what bench.py --mode forward times) gives the same scores
    again, _, _ = run_forward(dec, batch, enc_out)
    check_digest(g, again.float().cpu(), tol=2e-4)
    # the instantiations behind the headline: feature projection, image K/V of all layers, vocabulary
    conv = plans_of(plan_log, B * 196, 300, 2048)
    kv = plans_of(plan_log, B * 196, 1800, 300)
    vocab = plans_of(plan_log, B * 20, 10000, 300)
    assert conv and kv and vocab
    assert (conv[0]["tile_m"], conv[0]["tile_n"]) == ops.conv1_tile() and conv[0]["a_kmajor"] == 1
    assert kv[0]["tile_m"] >= 64 and vocab[0]["tile_m"] >= 64
    for pl in (conv[0], kv[0], vocab[0]):
        assert pl["vec"] == 1 and pl["split_bf16"] == (1 if gemm_split else 0)


def test_default_product_mode_and_graph_key():
    """The library's default (no ICK_GEMM_SPLIT in the environment) is the split-bf16 product on the forward layouts;
    switching the mode between two calls of one TrainStep re-captures instead of replaying the other mode's graph."""
    import os
    from ick_amd.training import TrainStep
    from test_training_gpu import zero_dropout
    if "ICK_GEMM_SPLIT" not in os.environ:
        assert ops.gemm_split_mode() == ops.DEFAULT_GEMM_SPLIT
    variant, B, L, K, V, seed = "geo", 4, 7, 6, 120, 3
    P = synth.make_params(variant, V, seed)
    dec = zero_dropout(build_decoder(variant, V, P).train())
    ts = TrainStep(dec, lr=0.0, grad_clip=5.0)
    b = synth.make_batch(variant, B, L, K, V, 0, seed)
    args = [b["captions"].cuda(), synth.make_enc_out(B, seed).cuda(), b["caption_masks"].cuda(),
            b["caption_lengths"].cuda(), b["entities"]]
    before = ops.gemm_split_mode()
    try:
        losses = []
        for mode in (0, 2, 0):
            ops.set_gemm_split(mode)
            losses.append(ts(*args).item())
        assert len(ts._graphs) == 2
        assert abs(losses[0] - losses[1]) < 1e-5 and abs(losses[0] - losses[2]) < 1e-6
    finally:
        ops.set_gemm_split(before)


def test_feature_map_inputs_run_conv1_inside_the_graph(gemm_split):
    """attach_encoder(): forward() / predict() / predict_beam() on the (B, 2048, 14, 14) feature map give exactly what
    they give on Encoder(feats) -- the same conv1 GEMM, only launched inside the captured graph beside the context
    chain -- and passing the graph's own input buffers back in (input_buffers()) changes nothing."""
    variant, B, L, K, V, seed = "geo", 8, 12, 6, 300, 11
    P = synth.make_params(variant, V, seed)
    enc, _, _ = make_encoder(seed)
    dec = build_decoder(variant, V, P).eval()
    b = synth.make_batch(variant, B, L, K, V, 0, seed)
    feats = synth.make_feats(B, seed).cuda()
    caps, masks, lens, ents = b["captions"].cuda(), b["caption_masks"].cuda(), b["caption_lengths"].cuda(), b["entities"].cuda()
    with torch.no_grad():
        e = enc(feats)
        ref_scores, ref_caps, ref_dl = dec(caps, e, masks, lens, ents)
        ref_tok = dec.predict(e, L, ents)
        ref_beam = dec.predict_beam(e, L, ents, beam_size=3)
        from ick_amd.lib import IckError
        with pytest.raises(IckError):
            dec.predict(feats, L, ents)                       # a feature map without an attached encoder
        dec.attach_encoder(enc)
        for rep in range(2):
            scores, caps_s, dl = dec(caps, feats, masks, lens, ents)
            assert dl == ref_dl and torch.equal(caps_s, ref_caps) and torch.equal(scores, ref_scores)
        bufs = dec.input_buffers()
        assert bufs[4].shape == feats.shape and bufs[4].data_ptr() != feats.data_ptr()
        scores, _, _ = dec(bufs[0], bufs[4], bufs[1], lens, bufs[2])          # the graph's own buffers: no input copy
        assert torch.equal(scores, ref_scores)
        assert torch.equal(dec.predict(feats, L, ents), ref_tok)
        img, ent_buf, _ = dec.input_buffers()
        assert img.dim() == 4 and torch.equal(dec.predict(img, L, ent_buf), ref_tok)
        assert torch.equal(dec.predict_beam(feats, L, ents, beam_size=3), ref_beam)
        # encoder outputs keep working on a decoder with an attached encoder
        assert torch.equal(dec.predict(e, L, ents), ref_tok)
    # the attached encoder is not part of the decoder's state
    assert not any(k.startswith("_enc") for k in dec.state_dict()) and "_enc" not in dec.__getstate__()
