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
def test_chain_matches_torch_and_unfused(M, K1, d, N2, relu):
    from ick_amd import ops
    assert ops.rowchain_supported(K1, d, N2)
    t = _mk(M, K1, d, N2, 7)
    x = torch.empty(M, d, device="cuda")
    o = torch.empty(M, d, device="cuda")
    y2 = torch.empty(M, N2, device="cuda") if N2 else None
    w1p = ops.pack_weight(t["w1"])
    w2p = ops.pack_weight(t["w2"]) if N2 else None
    mean, rstd = ops.rowchain_fwd(t["a"], w1p, t["b1"], t["res"], t["gamma"], t["beta"], 1e-5, x, o_out=o,
                                  save_stats=True, w2p=w2p, b2=t["b2"], y2=y2, relu=relu)
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
