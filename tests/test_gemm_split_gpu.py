"""Split-bf16 products of the large GEMM tiles (ick_set_gemm_split; csrc/gemm.hip, opt-in).

Every fp32 operand value is split exactly into three bf16 numbers, six of the nine partial products run on the bf16
matrix pipe with fp32 accumulation.  The claim tested here: against an fp64 product of the same fp32 operands the split
path's error is no larger than the exact fp32 MFMA path's (v_mfma_f32_16x16x4_f32), on every operand layout, tile shape
and epilogue the path uses -- so it is not a reduced-precision mode; and it never changes a result by more than a few
fp32 roundings of the accumulated sum.  The reference's call sites are the same as for ick_gemm (include/ick_amd.h)."""
import math

import pytest
import torch

import ick_amd  # noqa: F401
import ick_amd.lib as L

pytestmark = pytest.mark.gpu


@pytest.fixture()
def ops():
    from ick_amd import ops as o
    yield o
    o.set_gemm_split(0)


def rnd(*shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).cuda()


def errs(out, ref):
    e = out.double() - ref
    return e.abs().max().item(), e.pow(2).mean().sqrt().item()


def both(ops, fn):
    res = []
    for mode in (0, 2):
        ops.set_gemm_split(mode)
        res.append(fn())
    ops.set_gemm_split(0)
    return res


def plan_of(ops, a):
    info = L.GemmPlanInfo()
    L.check(L.load().ick_gemm_plan(a, info), "ick_gemm_plan")
    return info


@pytest.mark.parametrize("M,N,K,akm,bkm,split_k", [
    (1280, 10000, 300, False, False, 1),        # vocabulary projection (K tail of 12)
    (3456, 1800, 300, False, False, 1),         # cross K/V projection
    (8192, 300, 2048, True, False, 1),          # Encoder.conv1: k-major A, 128 x 64 tiles
    (1280, 300, 2000, False, True, 8),          # data gradient: k-major B, split K
    (2000, 300, 1280, True, True, 4),           # weight gradient: both k-major
    (2052, 1028, 516, False, False, 1),         # ragged edges (multiples of 4)
    (4096, 2048, 512, False, False, 1),         # 128 x 128 tiles
])
def test_split_error_not_above_exact_path(ops, M, N, K, akm, bkm, split_k):
    A = rnd(K, M, seed=1) if akm else rnd(M, K, seed=1)
    B = rnd(K, N, seed=2, scale=0.1) if bkm else rnd(N, K, seed=2, scale=0.1)
    ref = (A.double().t() if akm else A.double()) @ (B.double() if bkm else B.double().t())

    def run():
        out = torch.zeros(M, N, device="cuda")
        args = (A, B, out, M, N, K, 1 if akm else K, M if akm else 1, 1 if bkm else K, N if bkm else 1, N)
        kw = dict(atomic=split_k > 1, split_k=split_k)
        info = plan_of(ops, ops.gemm_args(*args, **kw))
        ops.gemm_raw(*args, **kw)
        torch.cuda.synchronize()
        return out, info

    (exact, pe), (split, ps) = both(ops, run)
    assert pe.split_bf16 == 0 and ps.split_bf16 == 1
    if (M, N) == (4096, 2048):
        assert (ps.tile_m, ps.tile_n) == (128, 128)
    if akm and not bkm:
        assert (ps.tile_m, ps.tile_n) == (128, 64)
    emax, erms = errs(exact, ref)
    smax, srms = errs(split, ref)
    assert srms <= 1.05 * erms, (srms, erms)
    assert smax <= 1.5 * emax, (smax, emax)
    # and in absolute terms: a few roundings of an fp32 sum of K products of this size
    unit = 2.0 ** -24 * math.sqrt(K) * 0.1
    assert smax < 200 * unit
    assert (split - exact).abs().max().item() < 250 * unit


def test_split_epilogues_and_grouped_launch(ops):
    """Bias + ReLU, accumulate, the fused column sums (bias gradient) and plain column sums in a grouped launch."""
    M, N, K = 1280, 512, 300
    x, w, b = rnd(M, K, seed=3), rnd(N, K, seed=4, scale=0.1), rnd(N, seed=5)
    ref = (x.double() @ w.double().t() + b.double()).relu()
    exact, split = both(ops, lambda: ops.linear(x, w, b, relu=True))
    assert errs(split, ref)[1] <= 1.05 * errs(exact, ref)[1]

    shapes = [(1280, 300, 300), (1280, 512, 300), (1280, 300, 512), (13824, 600, 300)]

    def grouped():
        problems, outs, keep = [], [], []
        for i, (Mi, Ni, Ki) in enumerate(shapes):
            dy, xx = rnd(Mi, Ni, seed=100 + i), rnd(Mi, Ki, seed=200 + i)
            dw, db = torch.zeros(Ni, Ki, device="cuda"), torch.zeros(Ni, device="cuda")
            keep += [dy, xx]
            problems.append(ops.gemm_args(dy, xx, dw, Ni, Ki, Mi, 1, Ni, 1, Ki, Ki, atomic=True,
                                          split_k=max(1, min(16, Mi // 256)), colsum_a=db))
            outs.append((dw, db, dy.double().t() @ xx.double(), dy.double().sum(0)))
        part = rnd(160, 600, seed=77)
        acc = torch.zeros(600, device="cuda")
        problems.append(ops.colsum_problem(part, acc))
        ops.gemm_grouped(problems)
        torch.cuda.synchronize()
        return outs, (acc, part.double().sum(0))

    (oe, ce), (os_, cs) = both(ops, grouped)
    for (dwe, dbe, rw, rb), (dws, dbs, _, _) in zip(oe, os_):
        assert errs(dws, rw)[1] <= 1.1 * errs(dwe, rw)[1] + 1e-9
        assert errs(dbs, rb)[0] <= 2.0 * errs(dbe, rb)[0] + 1e-4      # fp32 column sums in another order
    assert errs(cs[0], cs[1])[0] < 1e-4 and errs(ce[0], ce[1])[0] < 1e-4


def test_split_mode_one_takes_only_the_forward_layouts(ops):
    M, N, K = 1280, 2048, 300
    A, Bkc, Bkm = rnd(M, K, seed=1), rnd(N, K, seed=2), rnd(K, N, seed=3)
    out = torch.empty(M, N, device="cuda")
    ops.set_gemm_split(1)
    assert ops.gemm_split_mode() == 1
    assert plan_of(ops, ops.gemm_args(A, Bkc, out, M, N, K, K, 1, K, 1, N)).split_bf16 == 1
    assert plan_of(ops, ops.gemm_args(A, Bkm, out, M, N, K, K, 1, 1, N, N)).split_bf16 == 0
    small = torch.empty(64, 64, device="cuda")      # 32 x 32 tiles stay exact
    assert plan_of(ops, ops.gemm_args(A[:64], Bkc[:64], small, 64, 64, K, K, 1, K, 1, 64)).split_bf16 == 0
    ops.set_gemm_split(0)
    assert plan_of(ops, ops.gemm_args(A, Bkc, out, M, N, K, K, 1, K, 1, N)).split_bf16 == 0
    with pytest.raises(L.IckError):
        ops.set_gemm_split(3)


def test_forward_scores_with_split_products_match_the_exact_path():
    """Whole teacher-forced forward (cfg2 shapes, smaller batch) with ICK_GEMM_SPLIT mode 1 against mode 0: the logits
    move by less than the spread the exact path itself has against the reference goldens (1e-5), argmax identical."""
    import ick_amd.synth as synth
    from ick_amd import ops as o
    from test_forward_gpu import build_decoder
    variant, B, Lc, K, V = "geo", 16, 20, 20, 10000
    P = synth.make_params(variant, V, 0)
    batch = synth.make_batch(variant, B, Lc, K, V, 0, 3)
    enc = ick_amd.load_models(variant).Encoder(emb_dim=300).cuda().eval()
    feats = synth.make_feats(B, 3).cuda()
    dec = build_decoder(variant, V, P).eval()
    res = []
    try:
        for mode in (0, 1):
            o.set_gemm_split(mode)
            dec.invalidate_caches()
            with torch.no_grad():
                s, _, _ = dec(batch["captions"].cuda(), enc(feats), batch["caption_masks"].cuda(),
                              batch["caption_lengths"].cuda(), batch["entities"])
            res.append(s.clone())
    finally:
        o.set_gemm_split(0)
    d = (res[0] - res[1]).abs().max().item()
    assert d < 1e-5 * max(1.0, res[0].abs().max().item()), d
    assert torch.equal(res[0].argmax(-1), res[1].argmax(-1))


def test_train_step_with_split_products_matches_the_exact_path():
    """One captured TrainStep at cfg2 shapes (smaller batch), dropout on with the same counter-based masks, split mode 2
    (every large tile) against mode 0: loss within 1e-5, the gradient bucket within 2e-5 of its largest entry, the
    token count identical."""
    import ick_amd.synth as synth
    from ick_amd import ops as o
    from ick_amd.training import TrainStep
    from test_forward_gpu import build_decoder
    variant, B, Lc, K, V = "geo", 32, 20, 20, 10000
    P = synth.make_params(variant, V, 0)
    batch = {k: v.cuda() for k, v in synth.make_batch(variant, B, Lc, K, V, 0, 5).items()}
    enc = ick_amd.load_models(variant).Encoder(emb_dim=300).cuda().eval()
    feats = synth.make_feats(B, 5).cuda()
    res = []
    try:
        for mode in (0, 2):
            o.set_gemm_split(mode)
            dec = build_decoder(variant, V, P).train()
            ts = TrainStep(dec, lr=4e-4, grad_clip=5.0, seed=7, encoder=enc)
            loss = ts(batch["captions"], feats, batch["caption_masks"], batch["caption_lengths"], batch["entities"])
            res.append((loss.item(), ts.flat_g.clone(), ts.n))
    finally:
        o.set_gemm_split(0)
    (l0, g0, n), (l1, g1, _) = res
    assert abs(l0 - l1) < 1e-5 * max(1.0, abs(l0)), (l0, l1)
    assert g0[n + 1].item() == g1[n + 1].item()
    scale = g0[:n].abs().max().item()
    assert (g0[:n] - g1[:n]).abs().max().item() < 2e-5 * max(scale, 1e-3), ((g0[:n] - g1[:n]).abs().max().item(), scale)
