"""The fused decode step (csrc/decode.hip: 3 launches per decoder layer + 3 for the score head) against the CPU
oracle's full-recompute predict() (geo-aware/models.py:389-443): per-step scores within 2e-4, tokens identical;
and against the per-op launch sequence it replaces."""
import math

import pytest
import torch

import ick_amd
import ick_amd.synth as synth
from oracle import restatement as R
from test_forward_gpu import build_decoder

pytestmark = pytest.mark.gpu


def fused_steps(dec, enc_out, ents, facts, max_len):
    """Drive ick_decode_layers / ick_decode_select_greedy step by step, keeping every step's score row."""
    import ick_amd.ops as ops
    enc_out, ents, facts = dec._prepare_inputs(enc_out, ents, facts)
    enc_tok = dec._token_major(enc_out).contiguous()
    B, d, V, K = enc_tok.shape[0], dec.emb_dim, dec.vocab_size, ents.shape[1]
    ee, fe, kv, _, side = dec._encode_context(enc_tok, ents.contiguous(), facts, None)
    side.join()
    c, t = dec._decode_ctx(kv, ee, fe, 1, max_len, kv.shape[3], want_scores=True)
    tok = torch.full((B, 1), dec.word_map["<start>"], dtype=torch.long, device="cuda")
    x0 = ops.caption_embed(tok, torch.zeros_like(tok), dec.word_embedding.weight.detach(), ee, fe,
                           dec.pos_encoder.pe.view(-1, d), V, dec.word_map["<pad>"], math.sqrt(d), pos0=0)
    t["x0"].copy_(x0.view(B, d))
    rows = []
    for i in range(max_len):
        if dec.has_facts:
            ops.context_indicators(t["cap_buf"], facts, K, V, dec._pred_wt(), dec.fc_predicate.bias.detach(), mode=1,
                                   eib=t["eib"], gate=t["gate"])
        ops.decode_layers(c, i)
        rows.append(torch.cat([t["scores"], t["ptr"]], dim=1).clone())
        ops.decode_select_greedy(c, i)
    return t["output"].clone(), torch.stack(rows, dim=1), t     # (B, max_len), (B, max_len, Vx)


@pytest.mark.parametrize("variant,K,V,Fn,max_len,seed", [("geo", 6, 50, 0, 12, 3), ("knowledge", 6, 50, 5, 12, 1),
                                                         ("news", 7, 90, 6, 10, 2), ("geo", 20, 1000, 0, 20, 5),
                                                         ("knowledge", 20, 3000, 51, 33, 7),
                                                         ("news", 60, 90, 80, 6, 8)])     # S = 336 > 256: swept memory
def test_fused_decode_scores_and_tokens_vs_oracle(variant, K, V, Fn, max_len, seed):
    B = 3
    P = synth.make_params(variant, V, seed)
    dec = build_decoder(variant, V, P)
    ents = synth.make_entities(variant, B, K, V, seed)
    facts = synth.make_facts(variant, B, Fn, K, seed) if variant != "geo" else None
    enc_out = synth.make_enc_out(B, seed)
    out, scores, t = fused_steps(dec, enc_out.cuda(), ents, None if facts is None else facts.cuda(), max_len)
    cfg = R.config_from_word_map(variant, synth.make_word_map(V))
    for b in range(B):
        with torch.no_grad():
            seq, ref = R.predict(cfg, P, enc_out[b:b + 1], max_len, ents[b:b + 1], None if facts is None else facts[b:b + 1],
                                 return_scores=True)
        n = ref.shape[0]                     # the oracle stops at <end>
        err = (scores[b, :n].cpu() - ref).abs().max().item()
        assert err < 2e-4, (variant, b, err)
        assert out[b].cpu().tolist() == seq.view(-1).tolist(), (variant, b)
    # every kernel of a step after the last caption ended returns at once: the ended-row counter says so
    assert int(t["n_done"].item()) == int(t["finished"].sum().item())


def test_fused_decode_equals_per_op_path_and_graph_replay():
    variant, B, K, V, Fn, max_len, seed = "knowledge", 5, 9, 300, 11, 14, 4
    P = synth.make_params(variant, V, seed)
    dec = build_decoder(variant, V, P)
    ents = synth.make_entities(variant, B, K, V, seed)
    facts = synth.make_facts(variant, B, Fn, K, seed).cuda()
    enc = synth.make_enc_out(B, seed).cuda()
    res = {}
    for fused in (False, True):
        for graphs in (False, True):
            dec.fused_decode, dec.use_hip_graphs = fused, graphs
            res[(fused, graphs)] = dec.predict(enc, max_len, ents, facts).clone()
            again = dec.predict(enc, max_len, ents, facts)                  # replay / second eager run
            assert torch.equal(res[(fused, graphs)], again)
    ref = res[(False, False)]
    for k, v in res.items():
        assert torch.equal(v, ref), k
    dec.fused_decode, dec.use_hip_graphs = True, True


def test_fused_decode_early_exit_keeps_pad_after_end():
    """A vocabulary bias that makes <end> the first choice: every caption ends at step 0, the remaining steps are
    no-ops and the output keeps <pad> after <end> (geo-aware/models.py:386,414-416)."""
    variant, B, K, V, max_len, seed = "geo", 4, 6, 50, 9, 2
    P = synth.make_params(variant, V, seed)
    P["fc_vocab.bias"] = P["fc_vocab.bias"].clone()
    P["fc_vocab.bias"][V - 1] += 50.0
    dec = build_decoder(variant, V, P)
    seq = dec.predict(synth.make_enc_out(B, seed).cuda(), max_len, synth.make_entities(variant, B, K, V, seed))
    assert seq.shape == (max_len, B)
    assert (seq[0] == V - 1).all() and (seq[1:] == 0).all()


# ------------------------------------------------------------------------------------------------ beam search
@pytest.mark.parametrize("variant,K,V,Fn,max_len,beam,seeds", [("geo", 6, 50, 0, 10, 3, (0, 1, 2, 3)),
                                                               ("knowledge", 6, 50, 5, 10, 5, (0, 1, 2)),
                                                               ("geo", 20, 1000, 0, 12, 5, (4, 5))])
def test_beam_search_vs_cpu_beam_reference(variant, K, V, Fn, max_len, beam, seeds):
    """Device beam search against the oracle's CPU beam search (full recompute per step).  PARITY-UNPINNED against
    the reference, which has no beam search (geo-aware/eval.py:61,83).  Sequences must be identical; where fp32
    rounding flips a near-tie the device's sequence must score (under the oracle) within 1e-3 of the oracle's best."""
    cfg = R.config_from_word_map(variant, synth.make_word_map(V))
    exact = 0
    for seed in seeds:
        P = synth.make_params(variant, V, seed)
        dec = build_decoder(variant, V, P)
        ents = synth.make_entities(variant, 1, K, V, seed)
        facts = synth.make_facts(variant, 1, Fn, K, seed) if variant != "geo" else None
        enc = synth.make_enc_out(1, seed)
        args = [enc.cuda(), max_len, ents] + ([facts.cuda()] if facts is not None else [])
        seq, score, allseq, allscore = dec.predict_beam(*args, beam_size=beam, return_all=True)
        with torch.no_grad():
            ref_seq, ref_score, _ = R.predict_beam(cfg, P, enc, max_len, ents, facts, beam)
        mine = seq[:, 0].cpu().tolist()
        assert abs(score.item() - R.sequence_logprob(cfg, P, enc, ents, facts, mine, max_len)) < 1e-3   # its own score is right
        if mine == ref_seq.tolist():
            exact += 1
            assert abs(score.item() - ref_score) < 1e-3
        else:
            assert R.sequence_logprob(cfg, P, enc, ents, facts, mine, max_len) > ref_score - 1e-3, (seed, mine, ref_seq.tolist())
    assert exact >= len(seeds) - 1


def test_beam_one_is_greedy_and_batch_is_independent():
    variant, B, K, V, max_len, seed = "geo", 4, 8, 200, 12, 9
    P = synth.make_params(variant, V, seed)
    dec = build_decoder(variant, V, P)
    ents = synth.make_entities(variant, B, K, V, seed)
    enc = synth.make_enc_out(B, seed).cuda()
    assert torch.equal(dec.predict_beam(enc, max_len, ents, beam_size=1), dec.predict(enc, max_len, ents))
    full = dec.predict_beam(enc, max_len, ents, beam_size=5)
    assert full.shape == (max_len, B)
    for b in range(B):
        one = dec.predict_beam(enc[b:b + 1], max_len, ents[b:b + 1], beam_size=5)
        assert torch.equal(one[:, 0], full[:, b])
    again = dec.predict_beam(enc, max_len, ents, beam_size=5)          # graph replay
    assert torch.equal(again, full)
    with pytest.raises(ick_amd.lib.IckError):
        dec.predict_beam(enc, max_len, ents, beam_size=9)


def test_fused_decode_other_model_sizes():
    """The decode kernels are not specialised to d = 300 / 10 heads / FF = 512: emb_dim 256, 8 heads (head width 32, no
    pad columns), decoder_dim 384 (6 chunks), 2 layers -- forward and greedy decode vs the oracle."""
    variant, B, K, V, max_len, seed = "geo", 3, 7, 120, 9, 11
    d, H, FF, NL = 256, 8, 384, 2
    P = synth.make_params(variant, V, seed, d=d, decoder_dim=FF, encoder_dim=320, num_layers=NL)
    wm = synth.make_word_map(V)
    m = ick_amd.load_models(variant)
    dec = m.DecoderTransformer(word_map=wm, emb_dim=d, decoder_dim=FF, encoder_dim=320, num_heads=H, num_layers=NL)
    missing, unexpected = dec.load_state_dict(P, strict=False)
    assert missing == ["pos_encoder.pe"] and not unexpected
    dec = dec.cuda().eval()
    import ick_amd.ops as ops
    assert ops.decode_supported(d, H, FF, 196 + K, max_len)
    cfg = R.config_from_word_map(variant, wm, emb_dim=d, num_heads=H, num_layers=NL)
    ents = synth.make_entities(variant, B, K, V, seed)
    enc_out = synth.make_enc_out(B, seed, emb_dim=d)
    out, scores, t = fused_steps(dec, enc_out.cuda(), ents, None, max_len)
    for b in range(B):
        with torch.no_grad():
            seq, ref = R.predict(cfg, P, enc_out[b:b + 1], max_len, ents[b:b + 1], None, return_scores=True)
        n = ref.shape[0]
        assert (scores[b, :n].cpu() - ref).abs().max().item() < 2e-4
        assert out[b].cpu().tolist() == seq.view(-1).tolist()
    seqs = dec.predict(enc_out.cuda(), max_len, ents)
    assert torch.equal(seqs.t().contiguous(), out)
