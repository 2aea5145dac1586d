"""Split-bf16 products of the large GEMM tiles (ick_set_gemm_split; csrc/gemm.hip, csrc/gemm_ps.hip; mode 1 is the default).

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
    before = o.gemm_split_mode()
    yield o
    o.set_gemm_split(before)


def rnd(*shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).cuda()


def errs(out, ref):
    e = out.double() - ref
    return e.abs().max().item(), e.pow(2).mean().sqrt().item()


def both(ops, fn):
    res = []
    before = ops.gemm_split_mode()
    for mode in (0, 2):
        ops.set_gemm_split(mode)
        res.append(fn())
    ops.set_gemm_split(before)
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
    before = o.gemm_split_mode()
    try:
        for mode in (0, 1):
            o.set_gemm_split(mode)
            dec.invalidate_caches()
            with torch.no_grad():
                s, _, _ = dec(batch["captions"].cuda(), enc(feats), batch["caption_masks"].cuda(),
                              batch["caption_lengths"].cuda(), batch["entities"])
            res.append(s.clone())
    finally:
        o.set_gemm_split(before)
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
    before = o.gemm_split_mode()
    try:
        for mode in (0, 2):
            o.set_gemm_split(mode)
            dec = build_decoder(variant, V, P).train()
            ts = TrainStep(dec, lr=4e-4, grad_clip=5.0, seed=7, encoder=enc)
            loss = ts(batch["captions"], feats, batch["caption_masks"], batch["caption_lengths"], batch["entities"])
            res.append((loss.item(), ts.flat_g.clone(), ts.n))
    finally:
        o.set_gemm_split(before)
    (l0, g0, n), (l1, g1, _) = res
    assert abs(l0 - l1) < 1e-5 * max(1.0, abs(l0)), (l0, l1)
    assert g0[n + 1].item() == g1[n + 1].item()
    scale = g0[:n].abs().max().item()
    assert (g0[:n] - g1[:n]).abs().max().item() < 2e-5 * max(scale, 1e-3), ((g0[:n] - g1[:n]).abs().max().item(), scale)


# ------------------------------------------------------------------------------------------------ pre-split B operand
def bf16_planes(buf, N, K):
    """(S, 3, Np, 32) float64 view of a pre-split copy."""
    S, Np = (K + 31) // 32, (N + 63) // 64 * 64
    return buf.view(torch.bfloat16).view(S, 3, Np, 32).double()


@pytest.mark.parametrize("N,K,transposed", [(300, 2048, False), (1800, 300, False), (300, 10000, True), (70, 45, False),
                                            (130, 33, True)])
def test_presplit_copy_is_the_exact_three_way_split(ops, N, K, transposed):
    """hi + mid + lo reproduces every fp32 weight EXACTLY (three bf16 numbers carry the 24-bit significand), the planes
    are ordered by magnitude, rows beyond N and k beyond K are zero."""
    w = (rnd(K, N, seed=5).t() if transposed else rnd(N, K, seed=5))
    w = w * torch.logspace(-6, 3, N, device="cuda").view(N, 1)
    buf = ops.presplit_buffer(N, K, "cuda")
    buf.fill_(0x7f)
    ops.presplit_weights([(w, buf)])
    pl = bf16_planes(buf, N, K)
    S, Np = pl.shape[0], pl.shape[2]
    full = pl.permute(1, 2, 0, 3).reshape(3, Np, S * 32)        # (plane, n, k)
    rec = full.sum(0)
    assert torch.equal(rec[:N, :K], w.double())
    assert rec[N:].abs().max().item() == 0 if Np > N else True
    assert rec[:, K:].abs().max().item() == 0 if S * 32 > K else True
    assert (full[1].abs() <= full[0].abs() * 2.0 ** -8 + 1e-45).all() and (full[2].abs() <= full[0].abs() * 2.0 ** -16 + 1e-45).all()


@pytest.mark.parametrize("M,N,K,akm,grp,split_k,tile", [
    (12544, 300, 2048, True, 196, 1, (128, 80)),      # Encoder.conv1: k-major NCHW map, 196 positions per sample
    (1280, 300, 10000, False, 0, 12, (128, 80)),      # vocabulary data gradient: split K, atomics
    (1280, 10000, 300, False, 0, 1, (128, 128)),      # vocabulary projection (K tail of 12)
    (12544, 1800, 300, False, 0, 1, (128, 128)),      # cross K/V projection of the image rows
    (23044, 132, 516, False, 0, 1, (128, 80)),        # ragged: M tail, N < 320 (pieces beyond the padded rows), K tail of 4
    (2052, 1028, 516, False, 0, 1, (128, 128)),       # ragged edges of the square tile
    (2052, 1028, 516, True, 0, 1, (128, 128)),        # k-major A on the square tile
    (23044, 260, 516, True, 0, 1, (128, 80)),         # k-major A, K tail of 4 k lines
    # split K on the pre-split kernel: the (K slice, tile) pairs are dealt to the XCDs slice-major for ANY slice count
    # (a permutation of the workgroups: a wrong one shows as missing or doubled slices)
    (600, 300, 13824, True, 0, 8, (128, 80)),         # one cross K/V weight gradient as the step launches it
    (600, 300, 4096, True, 0, 2, (128, 80)), (600, 300, 4096, True, 0, 3, (128, 80)),
    (600, 300, 4096, True, 0, 5, (128, 80)), (600, 300, 4096, True, 0, 7, (128, 80)),
    (600, 300, 4096, True, 0, 16, (128, 80)), (1280, 300, 4096, False, 0, 9, (128, 80)),
])
def test_presplit_gemm_error_not_above_exact_path(ops, M, N, K, akm, grp, split_k, tile):
    if grp:
        Bn = M // grp
        A = rnd(Bn, K, grp, seed=1)                    # (sample, k, position): element (m, k) = A[m // grp, k, m % grp]
        Amat = A.permute(0, 2, 1).reshape(M, K).double()
    else:
        A = rnd(K, M, seed=1) if akm else rnd(M, K, seed=1)
        Amat = A.double().t() if akm else A.double()
    W = rnd(N, K, seed=2, scale=0.1)
    bias = rnd(N, seed=3)
    ref = Amat @ W.double().t() + (bias.double() if split_k == 1 else 0)
    ps = ops.presplit_buffer(N, K, "cuda")
    ops.presplit_weights([(W, ps)])

    def run(b_ps):
        out = torch.zeros(M, N, device="cuda")
        if grp:
            args = (A, W, out, M, N, K, 1, grp, K, 1, N)
            kw = dict(a_grp=grp, a_gs=K * grp)
        else:
            args = (A, W, out, M, N, K, 1 if akm else K, M if akm else 1, K, 1, N)
            kw = {}
        kw.update(atomic=split_k > 1, split_k=split_k, bias=bias if split_k == 1 else None, b_ps=b_ps)
        info = plan_of(ops, ops.gemm_args(*args, **kw))
        ops.gemm_raw(*args, **kw)
        torch.cuda.synchronize()
        return out, info

    before = ops.gemm_split_mode()
    ops.set_gemm_split(0)
    exact, pe = run(ps)
    ops.set_gemm_split(1)
    split, pp = run(ps)
    ops.set_gemm_split(before)
    assert pe.presplit == 0 and pe.split_bf16 == 0
    assert pp.presplit == 1 and pp.split_bf16 == 1 and (pp.tile_m, pp.tile_n) == tile and pp.split_k == split_k
    emax, erms = errs(exact, ref)
    smax, srms = errs(split, ref)
    assert srms <= 1.05 * erms, (srms, erms)
    assert smax <= 1.5 * emax, (smax, emax)
    # in absolute terms: a few roundings of an fp32 sum of K products of this size (the maximum over up to 12 M outputs)
    unit = 2.0 ** -24 * math.sqrt(K) * 0.1
    assert smax < 300 * unit and (split - exact).abs().max().item() < 500 * unit


def test_narrow_problems_pick_their_tile_by_fill(ops):
    """Encoder.conv1 (N = 300, K = 2048, one column pair): at batch 64 on the pre-split kernel's 128 x 80 tile; at batch 32
    (cfg5's prefill, 6 272 rows = 49 row tiles of 128) on its 64 x 160 four-wave tile (196 workgroups; round 5: greedy
    2.12 -> 2.06 ms against the stager-split 128 x 64 tiles); at batch 8 it stays on the tiles of csrc/gemm.hip.  The batch-32
    instantiation against an fp64 product."""
    before = ops.gemm_split_mode()
    ops.set_gemm_split(1)
    try:
        for Bn, want, tile in ((8, 0, None), (32, 1, (64, 160)), (64, 1, (128, 80))):
            M, N, K = Bn * 196, 300, 2048
            A, W, out = torch.empty(Bn, K, 196, device="cuda"), torch.empty(N, K, device="cuda"), torch.empty(M, N, device="cuda")
            ps = ops.presplit_buffer(N, K, "cuda")
            info = plan_of(ops, ops.gemm_args(A, W, out, M, N, K, 1, 196, K, 1, N, a_grp=196, a_gs=K * 196, b_ps=ps))
            assert info.presplit == want and (tile is None or (info.tile_m, info.tile_n) == tile), (Bn, info.presplit, info.tile_m, info.tile_n)
        Bn, M, N, K = 32, 32 * 196, 300, 2048
        A, W, bias = rnd(Bn, K, 196, seed=5).relu(), rnd(N, K, seed=6, scale=0.05), rnd(N, seed=7)
        out = torch.empty(M, N, device="cuda")
        ps = ops.presplit_buffer(N, K, "cuda")
        ops.presplit_weights([(W, ps)])
        ops.gemm_raw(A, W, out, M, N, K, 1, 196, K, 1, N, bias=bias, a_grp=196, a_gs=K * 196, b_ps=ps)
        ref = A.double().permute(0, 2, 1).reshape(M, K) @ W.double().t() + bias.double()
        assert (out.double() - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())
    finally:
        ops.set_gemm_split(before)


def test_presplit_gemm_epilogues(ops):
    """The K/V projection's epilogue (grouped source rows through a sample map, head-split scatter) and the
    accumulate form, pre-split against the exact path bit-layout for bit-layout."""
    Bn, P, d, H, nseg = 20, 196, 300, 10, 6
    S = P + 20
    enc = rnd(Bn, P, d, seed=1)
    w, b = rnd(nseg * d, d, seed=2, scale=0.1), rnd(nseg * d, seed=3)
    gmap = torch.randperm(Bn, generator=torch.Generator().manual_seed(0)).to(torch.int32).cuda()
    ps = ops.presplit_buffer(nseg * d, d, "cuda")
    ops.presplit_weights([(w, ps)])
    before = ops.gemm_split_mode()
    outs = []
    for mode in (0, 1):
        ops.set_gemm_split(mode)
        kv = torch.full((Bn, nseg, H, S, ops.DHP), 7.0, device="cuda")
        ops.PLAN_LOG = []
        ops.project_heads(enc, w, b, nseg, H, S, out=kv, s0=0, grp=P, a_gmap=gmap, a_gs=enc.stride(0), w_ps=ps)
        plan = ops.PLAN_LOG[-1][3]
        ops.PLAN_LOG = None
        assert plan["presplit"] == mode
        outs.append(kv)
    ops.set_gemm_split(before)
    exact, split = outs
    assert (split - exact).abs().max().item() < 2e-5
    assert torch.equal(split[..., 30:], exact[..., 30:]) and torch.equal(split[:, :, :, P:], exact[:, :, :, P:])   # untouched pads / rows


def test_non_finite_operands_documented_behaviour(ops):
    """the exact three-way split of +-inf (and of a finite value that rounds to an infinite bf16 hi plane) has
    NaN residual planes, so its dot products are NaN in the split modes where the exact fp32 MFMA gives +-inf / a finite
    sum.  Documented (ops.set_gemm_split), not patched: the split is on the instruction stream that bounds
    those tiles.  Finite operands of ordinary magnitude are unaffected -- including 1e30, far beyond anything a network
    holds."""
    M, N, K = 1280, 2048, 300
    A, W = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.1)
    A[3, 7] = float("inf")
    A[5, 9] = 1e30
    A[8, 1] = 3.4e38          # finite, but its bf16 hi plane rounds to infinity
    W[11, 9] = 1e-30
    out = {}
    for mode in (0, 1):
        ops.set_gemm_split(mode)
        o = torch.empty(M, N, device="cuda")
        ops.gemm_raw(A, W, o, M, N, K, K, 1, K, 1, N)
        out[mode] = o.clone()
    assert torch.isinf(out[0][3]).all() and not torch.isfinite(out[1][3]).any()          # +-inf vs NaN
    assert torch.isfinite(out[0][8]).all() and not torch.isfinite(out[1][8]).any()       # finite vs NaN
    ok = torch.ones(M, dtype=torch.bool, device="cuda")
    ok[[3, 8]] = False
    assert torch.isfinite(out[1][ok]).all()
    ref5 = A[5].double() @ W.double().t()
    assert (out[1][5].double() - ref5).abs().max().item() <= 1e-6 * ref5.abs().max().item()
