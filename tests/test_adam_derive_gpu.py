"""The optimizer kernel that keeps the re-laid-out weight copies current (ick_adam_clamp_derive, csrc/adam_derive.hip;
VERDICT r4 item 6): the reference's optimizer.step() (geo-aware/train.py:292, clip at :287-288) is the only writer of the
parameters, so the packed / transposed / bf16-plane images the next step's kernels read are written in the same pass.
  * the update itself is ick_adam_clamp's, bit for bit (parameters, moments, clamped gradients), on the real bucket of all
    three variants;
  * every image equals what the stand-alone packing kernels (ick_pack_weights, ick_presplit_weights) make of the updated
    weights, bit for bit, padding included;
  * a captured TrainStep with the images maintained this way computes what the per-step packing launches of rounds 2-4
    compute (ICK_ADAM_DERIVE=0), and notices parameters written from outside.
"""
import pytest
import torch

import ick_amd.ops as ops
import ick_amd.synth as synth
from ick_amd.training import DerivedWeights, TrainStep
from test_forward_gpu import build_decoder
from test_training_gpu import zero_dropout

pytestmark = pytest.mark.gpu


def _images(dw):
    dec = dw.dec
    out = {"pk": dec.__dict__["_chain_cache"]["buf"], "pkb": dec.__dict__["_chain_cache_bwd"]["buf"],
           "kvT": dw.pkb[("kv", "T")], "wkv": dw.wkv, "bkv": dw.bkv, "wkv_ps": dw.wkv_ps, "vocab_ps": dw.vocab_ps,
           "vocab_t_ps": dw.vocab_t_ps}
    if dw.pred_wt is not None:
        out["pred_wt"] = dw.pred_wt
    return out


@pytest.mark.parametrize("variant,V", [("geo", 10000), ("knowledge", 1000), ("news", 200)])
def test_update_and_images_are_bit_identical_to_the_separate_kernels(variant, V):
    P = synth.make_params(variant, V, 3)
    dec = build_decoder(variant, V, P).train()
    ts = TrainStep(dec, lr=4e-4, grad_clip=5.0)
    ts.derived = dw = DerivedWeights.build(ts)
    assert dw is not None and dw.n_items > 20 and dw.n_blocks > 100
    dw.refresh()
    g = torch.Generator(device="cuda").manual_seed(1)
    n = ts.n
    ts.flat_g[:n] = torch.randn(n, device="cuda", generator=g) * 40.0          # some elements beyond the clamp (5 x 7 tokens)
    ts.flat_g[n], ts.flat_g[n + 1] = 3.5, 7.0
    ts.flat_m.copy_(torch.randn(n, device="cuda", generator=g) * 0.01)
    ts.flat_v.copy_(torch.rand(n, device="cuda", generator=g) * 1e-3)
    ts.counter.fill_(4)
    ref = [t.clone() for t in (ts.flat_p, ts.flat_g, ts.flat_m, ts.flat_v)]
    ts._adam()
    ops.adam_clamp(ref[0], ref[1][:n], ref[2], ref[3], 1, 4e-4, 5.0, 1.0, 0.9, 0.999, 1e-8, step_tensor=ts.counter,
                   gscale_den=ts.flat_g[n + 1:])
    torch.cuda.synchronize()
    assert not torch.equal(ref[0], torch.zeros_like(ref[0]))
    for mine, r, what in zip((ts.flat_p, ts.flat_g, ts.flat_m, ts.flat_v), ref, "pgmv"):
        assert torch.equal(mine[:n], r[:n]), what
    assert (ts.flat_g[:n].abs().max().item() - 5.0) == 0.0                      # the clamp was exercised
    mine = {k: v.clone() for k, v in _images(dw).items()}
    dw.refresh()                                                                # the stand-alone kernels on the updated weights
    torch.cuda.synchronize()
    for k, v in _images(dw).items():
        assert torch.equal(mine[k], v), (variant, k)
    # ... and the images are the updated weights' (not the old ones'): spot checks against the parameters themselves
    d = dec.emb_dim
    l0 = dec.transformer_decoder.layers[0]
    assert torch.equal(dw.wkv[:2 * d], l0.multihead_attn.in_proj_weight.detach()[d:])
    assert torch.equal(dw.bkv[:2 * d], l0.multihead_attn.in_proj_bias.detach()[d:])
    if dw.pred_wt is not None:
        assert torch.equal(dw.pred_wt, dec.fc_predicate.weight.detach().t())


def test_zero_token_batch_leaves_parameters_and_images_alone():
    variant, V = "geo", 200
    dec = build_decoder(variant, V, synth.make_params(variant, V, 5)).train()
    ts = TrainStep(dec, lr=4e-4)
    ts.derived = dw = DerivedWeights.build(ts)
    dw.refresh()
    before = [t.clone() for t in (ts.flat_p, ts.flat_m, ts.flat_v)] + [v.clone() for v in _images(dw).values()]
    ts.flat_g.fill_(1.0)
    ts.flat_g[ts.n + 1] = 0.0          # no contributing token in the (global) batch
    ts._adam()
    torch.cuda.synchronize()
    after = [ts.flat_p, ts.flat_m, ts.flat_v] + list(_images(dw).values())
    assert all(torch.equal(a, b) for a, b in zip(before, after))


@pytest.mark.parametrize("variant,Fn", [("geo", 0), ("knowledge", 5)])
def test_captured_step_with_maintained_images_equals_per_step_packing(variant, Fn, monkeypatch):
    B, L, K, V, seed = 6, 9, 6, 160, 7
    P = synth.make_params(variant, V, seed)
    b = synth.make_batch(variant, B, L, K, V, Fn, seed)
    args = [b["captions"].cuda(), synth.make_enc_out(B, seed).cuda(), b["caption_masks"].cuda(),
            b["caption_lengths"].cuda(), b["entities"]] + ([b["facts"].cuda()] if Fn else [])

    def run(derive):
        monkeypatch.setenv("ICK_ADAM_DERIVE", "1" if derive else "0")
        dec = zero_dropout(build_decoder(variant, V, P).train())
        ts = TrainStep(dec, lr=4e-4, grad_clip=5.0)
        losses = [ts(*args).item() for _ in range(4)]
        assert ts.use_graph and (ts.derived is not None) == derive
        return ts, losses

    ts1, l1 = run(True)
    ts0, l0 = run(False)
    assert l1[0] > l1[-1]                                    # it trains
    for a, b_ in zip(l1, l0):
        assert abs(a - b_) < 2e-4 * max(1.0, abs(b_)), (l1, l0)
    # first-step noise of the float atomics passes through Adam's sign-like first step: compare well away from it
    assert (ts1.flat_p - ts0.flat_p).abs().max().item() <= 4 * 4e-4 + 1e-6
    assert ((ts1.flat_p - ts0.flat_p).abs() > 1e-4).float().mean().item() < 0.02


def test_outside_writes_to_the_parameters_are_noticed():
    variant, B, L, K, V, seed = "geo", 4, 7, 6, 120, 9
    P = synth.make_params(variant, V, seed)
    b = synth.make_batch(variant, B, L, K, V, 0, seed)
    args = [b["captions"].cuda(), synth.make_enc_out(B, seed).cuda(), b["caption_masks"].cuda(),
            b["caption_lengths"].cuda(), b["entities"]]
    dec = zero_dropout(build_decoder(variant, V, P).train())
    ts = TrainStep(dec, lr=0.0)
    l_a = ts(*args).item()
    assert ts.derived is not None
    with torch.no_grad():                                     # in-place torch ops: version counters move
        for l in dec.transformer_decoder.layers:
            l.linear1.weight.mul_(1.5)
            l.multihead_attn.in_proj_weight.mul_(0.5)
        dec.fc_vocab.weight.add_(0.01)
    l_b = ts(*args).item()
    P2 = {k: v.detach().cpu().clone() for k, v in dec.state_dict().items() if k != "pos_encoder.pe"}
    dec2 = zero_dropout(build_decoder(variant, V, P2).train())
    l_ref = TrainStep(dec2, lr=0.0)(*args).item()
    assert abs(l_b - l_ref) < 1e-5 and abs(l_a - l_b) > 1e-3
    # a write behind torch's back needs the explicit call
    ts.flat_p[:ts.n].mul_(1.0)          # (a no-op write: only the protocol is exercised)
    ts.parameters_changed()
    assert ts.derived.stale
    assert abs(ts(*args).item() - l_ref) < 1e-5 and not ts.derived.stale


def test_lazy_update_is_the_eager_order_of_optimizer_steps(monkeypatch):
    """TrainStep(lazy_update=True): step i's clamp + Adam runs at the head of step i + 1's graph (side stream, beside
    Encoder.conv1) instead of behind step i.  Same optimizer steps in the same order: the loss sequence over alternating
    batches equals the eager order's, the parameters lag exactly one update until flush(), the step counter advances with
    the applied updates only, and state_dict() sees the flushed state."""
    from test_bench_sizes_gpu import make_encoder
    variant, B, L, K, V, seed = "geo", 6, 9, 6, 160, 13
    P = synth.make_params(variant, V, seed)
    enc, _, _ = make_encoder(seed)
    batches = []
    for s_ in (seed, seed + 1):
        b = synth.make_batch(variant, B, L, K, V, 0, s_)
        batches.append([b["captions"].cuda(), synth.make_feats(B, s_).cuda(), b["caption_masks"].cuda(),
                        b["caption_lengths"].cuda(), b["entities"]])

    def run(lazy):
        dec = zero_dropout(build_decoder(variant, V, P).train())
        ts = TrainStep(dec, lr=4e-4, grad_clip=5.0, encoder=enc, lazy_update=lazy)
        losses, lagged = [], []
        for i in range(5):
            before = ts.flat_p.clone()
            losses.append(ts(*batches[i % 2]).item())
            torch.cuda.synchronize()
            lagged.append(torch.equal(before, ts.flat_p) if i == 0 else None)
            if i == 0:
                assert ts._pending == lazy and int(ts.counter.item()) == (0 if lazy else 1)
        assert ts.use_graph and ts.derived is not None
        if lazy:
            assert lagged[0] is True and int(ts.counter.item()) == 4        # four of the five updates applied so far
            sd = ts.state_dict()                                             # flushes
            assert not ts._pending and float(sd["state"][0]["step"]) == 5.0
        else:
            assert lagged[0] is False
        assert int(ts.counter.item()) == 5
        return ts, losses

    ts1, l1 = run(True)
    ts0, l0 = run(False)
    assert l0[0] > l0[-1]
    for a, b_ in zip(l1, l0):
        assert abs(a - b_) < 2e-4 * max(1.0, abs(b_)), (l1, l0)
    assert (ts1.flat_p - ts0.flat_p).abs().max().item() <= 5 * 4e-4 + 1e-6
    assert ((ts1.flat_p - ts0.flat_p).abs() > 1e-4).float().mean().item() < 0.02
    # a flushed step followed by more steps: nothing is applied twice
    l_next1, l_next0 = ts1(*batches[1]).item(), ts0(*batches[1]).item()
    assert abs(l_next1 - l_next0) < 2e-4 * max(1.0, abs(l_next0))
    ts1.flush()
    torch.cuda.synchronize()
    assert int(ts1.counter.item()) == int(ts0.counter.item()) == 6
