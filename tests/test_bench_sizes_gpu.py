"""Parity at the sizes bench.py quotes (the instantiations behind the headline numbers had no
correctness evidence).  Every test records, through ick_gemm_plan, WHICH kernel configuration ran, so a later
change of the dispatch cannot silently move the bench onto an untested instantiation:
  * Encoder.conv1 at B=64 (M = 12 544 rows, the bench's feature projection) against the oracle;
  * one full cfg2 TrainStep (B=64, L=20, K=20, V=10k; hipGraph path, dropout off) against the reference sequence
    (CE -> backward -> clamp -> Adam) on the CPU oracle: loss, gradients, post-Adam weights;
  * cfg4 at its real vocabulary (V=50 000, F=51): forward and one train step (B=16 keeps the oracle in seconds);
  * the mid-size news golden produced by the real reference;
  * packed cross entropy on rows that are long AND 16-byte aligned (the register-resident path and the re-read path).
"""
import pytest
import torch

import ick_amd
import ick_amd.synth as synth
from helpers import case_from_golden, load_golden, t
from oracle import restatement as R
from test_forward_gpu import build_decoder, check_scores, run_forward
from test_ops_gpu import close, dev, rnd
from test_training_gpu import zero_dropout

pytestmark = pytest.mark.gpu


@pytest.fixture()
def plan_log():
    import ick_amd.ops as ops
    ops.PLAN_LOG = []
    yield ops.PLAN_LOG
    ops.PLAN_LOG = None


def plans_of(log, M, N, K):
    return [p for (m, n, k, p) in log if (m, n, k) == (M, N, K)]


def make_encoder(seed):
    m = ick_amd.load_models("geo")
    enc = m.Encoder(emb_dim=300)
    cw, cb = synth.make_conv1(seed)
    with torch.no_grad():
        enc.conv1.weight.copy_(cw)
        enc.conv1.bias.copy_(cb)
    return enc.cuda().eval(), cw, cb


def test_conv1_at_bench_batch(plan_log, gemm_split):
    """B=64: M = 64*196 = 12 544 rows, the shape whose kernel bench.py's roofline block times."""
    B, seed = 64, 21
    enc, cw, cb = make_encoder(seed)
    feats = synth.make_feats(B, seed)
    out = enc(feats.cuda())
    with torch.no_grad():
        ref = R.feat_proj(feats.double(), cw.double(), cb.double())
    close(out, ref, 2e-5, "conv1 B=64")
    pl = plans_of(plan_log, B * 196, 300, 2048)
    assert len(pl) == 1
    # the long-K narrow-N instantiation (a dispatch change must update this line AND keep the parity above)
    assert pl[0]["vec"] == 1 and pl[0]["a_kmajor"] == 1 and (pl[0]["tile_m"], pl[0]["tile_n"]) == ick_amd.ops.conv1_tile()
    assert pl[0]["split_bf16"] == (1 if gemm_split else 0)


_REF_STEPS = {}


def reference_train_step_cached(key, cfg, P, batch, enc_out):
    """The oracle's step is the same for every product mode of the GEMM tiles: computed once per case."""
    if key not in _REF_STEPS:
        _REF_STEPS.clear()              # one case at a time (cfg4's gradients are 160 MB)
        _REF_STEPS[key] = reference_train_step(cfg, P, batch, enc_out)
    return _REF_STEPS[key]


def reference_train_step(cfg, P, batch, enc_out, lr=4e-4, clip=5.0):
    """geo-aware/train.py:275-292 on the CPU oracle: CE over packed rows, backward, clamp, Adam."""
    Pr = {k: v.clone().requires_grad_(True) for k, v in P.items() if not k.startswith("fact_encoder.")}
    if cfg.has_facts:
        Pr["fact_encoder.predicate_embedding.weight"] = Pr["predicate_embedding.weight"]
    uniq = [v for k, v in Pr.items() if not k.startswith("fact_encoder.")]
    opt = torch.optim.Adam(uniq, lr=lr)
    scores, caps, dl = R.forward(cfg, Pr, batch["captions"], enc_out, batch["caption_masks"],
                                 batch["caption_lengths"], batch["entities"], batch.get("facts"))
    loss = R.packed_ce_loss(cfg, scores, caps, dl)
    loss.backward()
    grads = {k: v.grad.clone() for k, v in Pr.items() if not k.startswith("fact_encoder.")}
    for p in uniq:
        p.grad.clamp_(-clip, clip)
    opt.step()
    return loss.item(), grads, {k: v.detach() for k, v in Pr.items()}


def run_train_step_vs_oracle(variant, B, L, K, V, Fn, seed, plan_log):
    from ick_amd.training import TrainStep
    P = synth.make_params(variant, V, seed)
    wm = synth.make_word_map(V)
    cfg = R.config_from_word_map(variant, wm)
    batch = synth.make_batch(variant, B, L, K, V, Fn, seed)
    enc_out = synth.make_enc_out(B, seed)
    loss_ref, grads_ref, P_after = reference_train_step_cached((variant, B, L, K, V, Fn, seed), cfg, P, batch, enc_out)
    dec = zero_dropout(build_decoder(variant, V, P).train())
    ts = TrainStep(dec, lr=4e-4, grad_clip=5.0)      # use_graph=True: the captured step the bench replays
    args = [batch["captions"].cuda(), enc_out.cuda(), batch["caption_masks"].cuda(), batch["caption_lengths"].cuda(),
            batch["entities"]]
    if variant != "geo":
        args.append(batch["facts"].cuda())
    loss = ts(*args)
    assert ts.use_graph, "hipGraph capture failed: the bench path was not exercised"
    assert abs(loss.item() - loss_ref) < 2e-5, (loss.item(), loss_ref)
    # after the step the bucket holds what Adam consumed: the token-mean gradient, clamped to +-5
    named = dict(dec.named_parameters())
    flipped = {}
    for k, gr in grads_ref.items():
        mine = ts.grads[id(named[k])].detach().cpu()
        gr = gr.clamp(-5.0, 5.0)
        d = (mine - gr).abs()
        scale = max(1e-3, gr.abs().max().item())
        if d.max().item() / scale >= 2e-3 and k.endswith(("linear1.weight", "linear1.bias")):
            # A ReLU input within rounding of zero may land on the other side of it than on the oracle: that one
            # (token, hidden unit) then contributes to the unit's row of dW1 / db1 or not (measured at cfg4, B = 64:
            # one row of decoder layer 0 off by 2e-5, every other row within 3e-9; tools/debug/diag_cfg4_grads.py).
            # At most two such hidden units per Linear are set aside; all other rows must match as usual.
            rows = d.view(d.shape[0], -1).max(dim=1).values
            bad = rows.topk(2).indices[rows.topk(2).values / scale >= 2e-3]
            flipped[k] = bad
            d = d.clone()
            d[bad] = 0
            assert (mine - gr).double().norm().item() <= 5e-3 * gr.double().norm().item(), ("gradient norm", k)
        err = d.max().item() / scale
        assert err < 2e-3, ("gradient", k, err)
    assert len(flipped) <= 2, flipped
    # The optimizer itself, on EVERY element: from the gradient the bucket held, Adam's first
    # step is w - lr * g / (|g| + eps) exactly (bias-corrected moments of a single step are g and g^2), so the post-step
    # weights are checked against that closed form with no conditioning allowance -- the comparison with the oracle's
    # weights below then only has to allow for what the (separately checked) gradient difference does to them.
    for k, p0 in P.items():
        if k.startswith("fact_encoder.") or k not in named:
            continue
        g = ts.grads[id(named[k])].detach().cpu().double()
        expect = p0.double() - 4e-4 * g / (g.abs() + 1e-8)
        err = (named[k].detach().cpu().double() - expect).abs().max().item()
        assert err < 2e-7 + 1e-6 * p0.abs().max().item(), ("Adam closed form", k, err)
    for k, pr in P_after.items():
        if k.startswith("fact_encoder."):
            continue
        diff = (named[k].detach().cpu() - pr).abs()
        if k in flipped:
            diff[flipped[k]] = 0
        # Adam's first step is lr * g / (|g| + eps): where |g| is within ~100 eps = 1e-6 of zero the step amplifies
        # fp32 summation-order noise in g by lr / eps = 4e4, so only well-conditioned elements are held to 5e-5
        # (a wrong gradient there moves the weight by ~lr = 4e-4); the rest may differ by at most one full step
        # ... and, where the (already checked) gradient the bucket held differs from the oracle's by more than 5 % of
        # the element itself -- the fallout of a flipped ReLU upstream on elements a thousand times smaller than the
        # tensor's largest -- the element is not "well conditioned" either; those must stay rare
        gref = grads_ref[k].clamp(-5.0, 5.0)
        big = gref.abs() >= 1e-6
        well = big & ((ts.grads[id(named[k])].detach().cpu() - gref).abs() <= 0.05 * gref.abs())
        assert diff.max().item() <= 8.5e-4, ("post-Adam weight (any element)", k, diff.max().item())
        if big.any():
            assert well.sum().item() >= 0.98 * big.sum().item(), ("too few well-conditioned elements", k)
        if well.any():
            err = diff[well].max().item()
            assert err < 5e-5, ("post-Adam weight", k, err)
    return ts, plan_log


def test_cfg2_train_step_vs_oracle(plan_log, gemm_split):
    """The bench workload itself: B=64, L=20, K=20, V=10 000 (1280 decode positions, score width 10 020)."""
    c = synth.CONFIGS["cfg2"]
    B, L, K, V = c["B"], c["L"], c["K"], c["V"]
    ts, log = run_train_step_vs_oracle("geo", B, L, K, V, 0, 31, plan_log)
    M = B * L
    vocab_fwd = plans_of(log, M, V, 300)
    vocab_dgrad = plans_of(log, M, 300, V)
    assert vocab_fwd and vocab_dgrad, "vocabulary GEMMs were not logged"
    assert vocab_fwd[0]["tile_m"] >= 64 and vocab_fwd[0]["vec"] == 1      # large tiles above 512 workgroups
    assert vocab_dgrad[0]["split_k"] >= 2                                  # long reduction split over workgroups
    assert vocab_fwd[0]["split_bf16"] == (1 if gemm_split else 0)          # the product mode under test really ran
    assert vocab_dgrad[0]["split_bf16"] == (1 if gemm_split else 0)
    if gemm_split:      # forward and data gradient multiply with pre-split weight copies (csrc/gemm_ps.hip: W, W^T)
        assert vocab_fwd[0]["presplit"] == 1 and (vocab_fwd[0]["tile_m"], vocab_fwd[0]["tile_n"]) == (128, 128)
        assert vocab_dgrad[0]["presplit"] == 1 and (vocab_dgrad[0]["tile_m"], vocab_dgrad[0]["tile_n"]) == (128, 80)
    kv = plans_of(log, B * 196, 1800, 300)
    assert kv and kv[0]["tile_m"] >= 64
    assert (V + K) % 4 == 0 and V + K <= 10240      # => packed CE keeps these rows in registers (score_head.hip)


def test_cfg4_real_vocab_forward_and_train_step(plan_log):
    """Knowledge variant at V=50 000, K=20, F=51 (score width 50 071 -> padded row stride 50 072)."""
    c = synth.CONFIGS["cfg4"]
    variant, L, K, V, Fn = c["variant"], c["L"], c["K"], c["V"], c["F"]
    B, seed = 16, 33
    assert V == 50000 and Fn == 51
    P = synth.make_params(variant, V, seed)
    wm = synth.make_word_map(V)
    cfg = R.config_from_word_map(variant, wm)
    batch = synth.make_batch(variant, B, L, K, V, Fn, seed)
    enc_out = synth.make_enc_out(B, seed)
    with torch.no_grad():
        ref, caps_ref, dl_ref = R.forward(cfg, P, batch["captions"], enc_out, batch["caption_masks"],
                                          batch["caption_lengths"], batch["entities"], batch["facts"])
    dec = build_decoder(variant, V, P)
    scores, caps, dl = run_forward(dec, batch, enc_out)
    assert dl == dl_ref and torch.equal(caps.cpu(), caps_ref)
    check_scores(scores, ref, "cfg4 V=50k forward")
    del dec, scores
    run_train_step_vs_oracle(variant, B, L, K, V, Fn, seed, plan_log)


def test_news_mid_golden():
    """News variant beyond the tiny fixture: K=21 entity rows with name words, F=31 facts, V=400 (real reference)."""
    g = load_golden("fwd_mid_news")
    cfg, P, wm, batch, enc_out = case_from_golden(g)
    dec = build_decoder(cfg.variant, cfg.vocab_size, P)
    stages = {}
    scores, caps, dl = run_forward(dec, batch, enc_out, stages)
    assert dl == g["decode_lengths"].tolist()
    assert torch.equal(caps.cpu(), t(g["captions_sorted"]))
    check_scores(scores, t(g["scores"]), "fwd_mid_news")
    assert (stages["entities_encoded"].cpu() - t(g["entities_encoded"])).abs().max() < 1e-7
    assert (stages["facts_encoded"].cpu() - t(g["facts_encoded"])).abs().max() < 1e-7
    assert (stages["fact_context"].cpu() - t(g["fact_context"])).abs().max() < 2e-4
    assert (stages["h"].cpu() - t(g["h"])).abs().max() < 2e-4


def test_news_mid_golden_gradients():
    from test_training_gpu import reference_loss
    g = load_golden("fwd_mid_news")
    cfg, P, wm, batch, enc_out = case_from_golden(g)
    dec = build_decoder(cfg.variant, cfg.vocab_size, P)
    scores, caps, dl = dec(batch["captions"].cuda(), enc_out.cuda(), batch["caption_masks"].cuda(),
                           batch["caption_lengths"].cuda(), batch["entities"], batch["facts"].cuda())
    loss = reference_loss(scores, caps, dl, wm["<pad>"])
    assert abs(loss.item() - float(g["loss"][0])) < 2e-5
    loss.backward()
    named = dict(dec.named_parameters())
    norms = dict(zip(g["grad_norm_names"].tolist(), g["grad_norms"].tolist()))
    for k, n in norms.items():
        if k.startswith("fact_encoder."):
            continue
        mine = float(named[k].grad.double().norm())
        assert abs(mine - n) <= 2e-4 * max(n, 1e-3), (k, mine, n)


@pytest.mark.parametrize("Vx,ld", [(10020, 10020), (12004, 12004), (50071, 50072)])
def test_packed_ce_long_rows(Vx, ld):
    """10 020 = cfg2's score width (aligned, register-resident); 12 004 = aligned but longer than the registers hold
    (the re-read path for a LENGTH reason); 50 071 / 50 072 = cfg4's width in its padded rows."""
    import ick_amd.ops as ops
    B, Lc, pad = 4, 7, 0
    buf = rnd(B, Lc, ld, seed=3, scale=3.0)
    sc = buf[:, :, :Vx]
    caps = torch.randint(1, Vx, (B, Lc), generator=torch.Generator().manual_seed(1))
    lens = torch.tensor([7, 6, 4, 3])
    for b in range(B):
        caps[b, lens[b]:] = 0
    dl = (lens - 1).to(torch.int32)
    cfg = R.Config("geo", Vx)
    sc_ref = sc.clone().double().requires_grad_(True)
    loss_ref = R.packed_ce_loss(cfg, sc_ref, caps, dl.tolist())
    loss_ref.backward()
    ls, cnt, dsc = ops.packed_ce(dev(buf)[:, :, :Vx], dev(caps), dev(dl), pad, want_grad=True)
    assert cnt.item() == float(dl.sum())
    assert abs(ls.item() / cnt.item() - loss_ref.item()) < 1e-5
    close(dsc / cnt, sc_ref.grad, 1e-6, "dscores Vx=%d" % Vx)
