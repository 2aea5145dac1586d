"""Row-resident chain kernel (ick_rowchain_fwd) against the unfused kernels it replaces and against torch.

The chain is out-projection -> dropout -> add & norm -> next Linear of torch's post-LN Transformer layers
(built by the reference at geo-aware/models.py:241-244)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _mk(M, K1, d, N2, seed):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    return dict(a=r(M, K1).cuda(), w1=(r(d, K1) / K1 ** 0.5).cuda(), b1=(0.1 * r(d)).cuda(), res=r(M, d).cuda(),
                gamma=(1 + 0.1 * r(d)).cuda(), beta=(0.1 * r(d)).cuda(),
                w2=(r(N2, d) / d ** 0.5).cuda() if N2 else None, b2=(0.1 * r(N2)).cuda() if N2 else None)


def _torch_ref(t, relu):
    o = t["a"].double() @ t["w1"].double().t() + t["b1"].double()
    x = torch.nn.functional.layer_norm(o + t["res"].double(), (o.shape[1],), t["gamma"].double(), t["beta"].double(), 1e-5)
    y = None
    if t["w2"] is not None:
        y = x @ t["w2"].double().t() + t["b2"].double()
        if relu:
            y = y.relu()
    return o, x, y


@pytest.mark.parametrize("M,K1,d,N2,relu", [(1280, 300, 300, 300, False), (1280, 300, 300, 512, True),
                                             (1280, 512, 300, 900, False), (1280, 512, 300, 0, False),
                                             (37, 300, 300, 512, True), (5, 384, 256, 768, False),
                                             (64, 64, 64, 64, False), (100, 300, 300, 1024, False)])
@pytest.mark.parametrize("slim", [False, True])
def test_chain_matches_torch_and_unfused(M, K1, d, N2, relu, slim):
    from ick_amd import ops
    assert ops.rowchain_supported(K1, d, N2)
    t = _mk(M, K1, d, N2, 7)
    x = torch.empty(M, d, device="cuda")
    o = torch.empty(M, d, device="cuda")
    y2 = torch.empty(M, N2, device="cuda") if N2 else None
    w1p = ops.pack_weight(t["w1"])
    w2p = ops.pack_weight(t["w2"]) if N2 else None
    mean, rstd = ops.rowchain_fwd(t["a"], w1p, t["b1"], t["res"], t["gamma"], t["beta"], 1e-5, x, o_out=o,
                                  save_stats=True, w2p=w2p, b2=t["b2"], y2=y2, relu=relu, slim=slim)
    ro, rx, ry = _torch_ref(t, relu)
    assert (o.double() - ro).abs().max() < 2e-5
    assert (x.double() - rx).abs().max() < 2e-5
    pre = ro + t["res"].double()
    assert (mean.double() - pre.mean(1)).abs().max() < 1e-5
    assert (rstd.double() - (pre.var(1, unbiased=False) + 1e-5).rsqrt()).abs().max() < 1e-4
    if N2:
        assert (y2.double() - ry).abs().max() < 5e-5
    # the unfused kernels: same LayerNorm arithmetic given the same o
    o_u = ops.linear(t["a"], t["w1"], t["b1"])
    x_u, m_u, r_u = ops.add_layernorm(o, t["res"], t["gamma"], t["beta"], 1e-5, save_stats=True)
    assert (o - o_u).abs().max() < 1e-5
    assert torch.equal(x, x_u) and torch.equal(mean, m_u) and torch.equal(rstd, r_u)


def test_packed_weight_copies():
    from ick_amd import ops
    g = torch.Generator().manual_seed(3)
    srcs = [torch.randn(300, 300, generator=g).cuda(), torch.randn(512, 300, generator=g).cuda(),
            torch.randn(900, 300, generator=g).cuda()[:300], torch.randn(33, 70, generator=g).cuda(),
            torch.randn(300, 512, generator=g).cuda()]
    dsts = [torch.full((ops.packed_weight_floats(*s.shape),), float("nan"), device="cuda") for s in srcs]
    ops.pack_weights(list(zip(srcs, dsts)))
    for s, dd in zip(srcs, dsts):
        N, K = s.shape
        ns, K16 = (N + 63) // 64, (K + 15) // 16 * 16
        ref = torch.zeros(ns * 64, K16, device="cuda")
        ref[:N, :K] = s
        ref = ref.view(ns, 64, K16 // 4, 4).permute(0, 2, 1, 3).contiguous().view(-1)
        assert torch.equal(dd, ref)


def test_plain_copies_ride_in_the_packing_launch():
    """ick_pack_item.dst_rs > 0: a plain (N, K) copy in the same launch -- the training step gathers the all-layer
    cross K/V weight (rows d.. of every layer's in_proj_weight) and bias this way instead of two torch.cat launches."""
    from ick_amd import ops
    g = torch.Generator().manual_seed(5)
    w = [torch.randn(900, 300, generator=g).cuda() for _ in range(3)]
    bias = [torch.randn(900, generator=g).cuda() for _ in range(3)]
    packed_src = torch.randn(300, 300, generator=g).cuda()
    packed_dst = torch.full((ops.packed_weight_floats(300, 300),), float("nan"), device="cuda")
    wkv = torch.full((1800, 300), float("nan"), device="cuda")
    bkv = torch.full((1800,), float("nan"), device="cuda")
    odd = torch.randn(70, 33, generator=g).cuda()                      # ragged rows / columns, strided destination
    odd_dst = torch.full((70, 40), float("nan"), device="cuda")
    copies = []
    for i in range(3):
        copies.append((w[i][300:], wkv[600 * i:600 * (i + 1)]))
        copies.append((bias[i][300:].view(1, -1), bkv[600 * i:600 * (i + 1)].view(1, -1)))
    copies.append((odd, odd_dst[:, :33]))
    ops.pack_weights([(packed_src, packed_dst)], copies)
    assert torch.equal(wkv, torch.cat([x[300:] for x in w]))
    assert torch.equal(bkv, torch.cat([x[300:] for x in bias]))
    assert torch.equal(odd_dst[:, :33], odd) and torch.isnan(odd_dst[:, 33:]).all()
    ref = torch.zeros(5 * 64, 304, device="cuda")
    ref[:300, :300] = packed_src
    assert torch.equal(packed_dst, ref.view(5, 64, 76, 4).permute(0, 2, 1, 3).contiguous().view(-1))


def test_chain_dropout_masks_and_head_split_equal_the_unfused_kernels():
    from ick_amd import ops
    B, T, d, H, FF = 8, 20, 300, 10, 512
    M = B * T
    t = _mk(M, d, d, 3 * d, 11)
    d1, d2 = (0.3, 1234, 5), (0.2, 1234, 9)
    # fused: out-projection + dropout1 + norm + in_proj scattered head-major
    x = torch.empty(B, T, d, device="cuda")
    o = torch.empty(M, d, device="cuda")
    qkv = torch.zeros(B, 3, H, T, ops.DHP, device="cuda")
    ops.rowchain_fwd(t["a"], ops.pack_weight(t["w1"]), t["b1"], t["res"], t["gamma"], t["beta"], 1e-5, x, drop1=d1,
                     o_out=o, w2p=ops.pack_weight(t["w2"]), b2=t["b2"], y2=qkv, heads=(3, H, T, 0, T))
    x_u = ops.add_layernorm(o, t["res"], t["gamma"], t["beta"], 1e-5, drop=d1)
    assert torch.equal(x.view(M, d), x_u)
    qkv_u = ops.project_heads(x, t["w2"], t["b2"], 3, H, T)
    assert (qkv[..., :d // H] - qkv_u[..., :d // H]).abs().max() < 2e-5
    # fused: + ReLU + dropout2 (linear1), strided x rows inside a wider buffer
    t = _mk(M, d, d, FF, 12)
    mem = torch.zeros(B, T + 6, d, device="cuda")
    f = torch.empty(M, FF, device="cuda")
    ops.rowchain_fwd(t["a"], ops.pack_weight(t["w1"]), t["b1"], t["res"], t["gamma"], t["beta"], 1e-5, mem[:, 3:3 + T],
                     drop1=d1, o_out=o, w2p=ops.pack_weight(t["w2"]), b2=t["b2"], y2=f, relu=True, drop2=d2)
    x_u = ops.add_layernorm(o, t["res"], t["gamma"], t["beta"], 1e-5, drop=d1)
    assert torch.equal(mem[:, 3:3 + T].reshape(M, d), x_u)
    assert mem[:, :3].abs().max() == 0 and mem[:, 3 + T:].abs().max() == 0
    f_u = ops.linear(x_u, t["w2"], t["b2"], relu=True, drop=d2)
    assert ((f == 0) == (f_u == 0)).all()
    assert (f - f_u).abs().max() < 5e-5


def _ln_bwd_ref(dx, o, res, gamma, mask):
    """float64 reference of the add & norm backward: returns dz (gradient of the normalised sum) and do = dz * mask."""
    z = (o.double() * mask.double() + res.double()).requires_grad_(True)
    y = torch.nn.functional.layer_norm(z, (z.shape[1],), gamma.double(), torch.zeros_like(gamma).double(), 1e-5)
    y.backward(dx.double())
    zh = ((z - z.mean(1, keepdim=True)) * (z.var(1, unbiased=False, keepdim=True) + 1e-5).rsqrt()).detach()
    return z.grad, z.grad * mask.double(), (dx.double() * zh).sum(0), dx.double().sum(0)


@pytest.mark.parametrize("M,ffn,pre,p", [(1280, True, True, 0.0), (1280, False, True, 0.0), (37, True, False, 0.0),
                                          (160, True, True, 0.3), (160, False, True, 0.2)])
def test_backward_chain_matches_reference(M, ffn, pre, p):
    from ick_amd import ops
    d, FF, K0 = 300, 512, 900 if ffn else 300
    g = torch.Generator().manual_seed(5)
    r = lambda *s: torch.randn(*s, generator=g).cuda()
    seed, ep = 77, None
    mk = lambda site: ops.dropout_mask(M, d, p, seed, site) if p > 0 else torch.ones(M, d, device="cuda")
    dr = lambda site: (p, seed, site) if p > 0 else None

    def norm_inputs(site):
        o, res, gamma = r(M, d), r(M, d), 1 + 0.1 * r(d)
        z = o * mk(site) + res
        return dict(o=o, res=res, gamma=gamma, mean=z.mean(1), rstd=(z.var(1, unbiased=False) + 1e-5).rsqrt(),
                    drop=dr(site), do=torch.empty(M, d, device="cuda"), part=ops.ln_partials(M, d, "cuda"))

    n1 = norm_inputs(3)
    w3 = r(d, d) / d ** 0.5
    g0 = r(M, K0) if pre else None
    w0 = r(K0, d) / K0 ** 0.5 if pre else None
    dzin = r(M, d)
    out3, dz_out = torch.empty(M, d, device="cuda"), torch.empty(M, d, device="cuda")
    kw = {}
    if ffn:
        n2 = norm_inputs(4)
        w2l, w1l = r(d, FF) / FF ** 0.5, r(FF, d) / d ** 0.5        # linear2.weight (d, FF), linear1.weight (FF, d)
        act = (r(M, FF).relu() * (torch.rand(M, FF, generator=g).cuda() > 0.3)).contiguous()
        t_out = torch.empty(M, FF, device="cuda")
        kw = dict(ffn=dict(w1p=ops.pack_weight(w2l.t()), w2p=ops.pack_weight(w1l.t()), act=act, gate_scale=1.25,
                           t_out=t_out), norm2=n2)
    ops.rowchain_bwd(M, d, n1, ops.pack_weight(w3.t()), out3, dz_out, g0=g0, w0p=ops.pack_weight(w0.t()) if pre else None,
                     dzin=dzin, **kw)
    # reference
    dx = dzin.double() + (g0.double() @ w0.double() if pre else 0)
    dz1, do1, dg1, db1 = _ln_bwd_ref(dx, n1["o"], n1["res"], n1["gamma"], mk(3))
    tol = 3e-5
    assert (n1["do"].double() - do1).abs().max() < tol
    assert (n1["part"].double().sum(0)[:d] - dg1).abs().max() < 1e-3 and (n1["part"].double().sum(0)[d:] - db1).abs().max() < 1e-3
    last_do, last_dz = do1, dz1
    if ffn:
        t = (do1 @ w2l.double()) * (act.double() > 0) * 1.25
        assert (t_out.double() - t).abs().max() < tol
        dx2 = dz1 + t @ w1l.double()
        dz2, do2, dg2, db2 = _ln_bwd_ref(dx2, n2["o"], n2["res"], n2["gamma"], mk(4))
        assert (n2["do"].double() - do2).abs().max() < 2 * tol
        assert (n2["part"].double().sum(0)[:d] - dg2).abs().max() < 1e-3
        last_do, last_dz = do2, dz2
    assert (dz_out.double() - last_dz).abs().max() < 2 * tol
    assert (out3.double() - last_do @ w3.double()).abs().max() < 2 * tol


@pytest.mark.parametrize("d,H,FF,NL,variant", [(300, 10, 512, 3, "geo"), (256, 8, 384, 2, "geo"), (300, 10, 512, 2, "knowledge")])
def test_model_scores_and_gradients_with_and_without_chains(d, H, FF, NL, variant):
    """The row chains (forward and backward) against the separate GEMM / add & norm / LayerNorm-backward kernels they
    replace, through the whole drop-in module (autograd bridge, dropout off): scores and every parameter gradient."""
    import ick_amd
    import ick_amd.synth as synth
    from torch.nn.utils.rnn import pack_padded_sequence
    B, L, K, V, Fn, seed = 5, 9, 6, 150, 7, 21
    P = synth.make_params(variant, V, seed, d=d, decoder_dim=FF, encoder_dim=320, num_layers=NL)
    wm = synth.make_word_map(V)
    m = ick_amd.load_models(variant)
    batch = synth.make_batch(variant, B, L, K, V, Fn, seed)
    enc_out = synth.make_enc_out(B, seed, emb_dim=d).cuda()
    results = []
    for chains in (True, False):
        dec = m.DecoderTransformer(word_map=wm, emb_dim=d, decoder_dim=FF, encoder_dim=320, num_heads=H, num_layers=NL)
        dec.load_state_dict(P, strict=False)
        dec = dec.cuda().train()
        for mod in dec.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
            if isinstance(mod, torch.nn.MultiheadAttention):
                mod.dropout = 0.0
        assert dec.chain_supported() and dec.chain_bwd_supported()
        if not chains:
            dec.__dict__["_chain_ok"] = False
            dec.__dict__["_chain_bwd_ok"] = False
        args = [batch["captions"].cuda(), enc_out, batch["caption_masks"].cuda(), batch["caption_lengths"].cuda(),
                batch["entities"]]
        if "facts" in batch:
            args.append(batch["facts"].cuda())
        scores, caps, dl = dec(*args)
        sp = pack_padded_sequence(scores, dl, batch_first=True).data
        tp = pack_padded_sequence(caps[:, 1:], dl, batch_first=True).data
        loss = torch.nn.functional.cross_entropy(sp, tp, ignore_index=wm["<pad>"])
        loss.backward()
        results.append((scores.detach(), loss.item(), {k: p.grad.clone() for k, p in dec.named_parameters() if p.grad is not None}))
    (s1, l1, g1), (s0, l0, g0) = results
    assert (s1 - s0).abs().max().item() < 2e-4
    assert abs(l1 - l0) < 2e-5
    assert g1.keys() == g0.keys()
    for k in g1:
        scale = max(1.0, g0[k].abs().max().item())
        assert (g1[k] - g0[k]).abs().max().item() < 3e-5 * scale, k
