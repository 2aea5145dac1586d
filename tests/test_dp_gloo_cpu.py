"""world_size-2 gloo test (CPU) of the data-parallel exchange: every rank computes the gradient of the
SUM of its shard's token losses (here with the CPU oracle), one all-reduce of the flat bucket, divide by
the reduced token count == the reference's single-process token-mean gradient over the global batch."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _bucket_for(cfg, P, batch, enc_out, lo, hi, names):
    from oracle import restatement as R
    import torch.nn.functional as F
    Pl = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    sub = {k: v[lo:hi] for k, v in batch.items()}
    scores, caps, dl = R.forward(cfg, Pl, sub["captions"], enc_out[lo:hi], sub["caption_masks"],
                                 sub["caption_lengths"], sub["entities"], sub.get("facts"))
    B, L, Vx = scores.shape
    keep = torch.arange(L - 1).view(1, -1) < torch.tensor(dl).view(B, 1)
    rows, tg = scores[:, :L - 1][keep], caps[:, 1:][keep]
    loss_sum = F.cross_entropy(rows, tg, ignore_index=cfg.pad, reduction="sum")
    count = (tg != cfg.pad).sum().float()
    loss_sum.backward()
    flat = torch.cat([Pl[k].grad.reshape(-1) if Pl[k].grad is not None else torch.zeros(Pl[k].numel()) for k in names]
                     + [loss_sum.detach().view(1), count.view(1)])
    return flat


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    import ick_amd.dp as dp
    import ick_amd.synth as synth
    from oracle import restatement as R
    dp.init_from_env(backend="gloo")
    variant, B, L, K, V, seed = "geo", 6, 8, 5, 60, 3
    P = synth.make_params(variant, V, seed)
    cfg = R.config_from_word_map(variant, synth.make_word_map(V))
    batch = synth.make_batch(variant, B, L, K, V, 0, seed)
    batch["caption_lengths"][:3] = 8   # rank 0 holds more tokens than rank 1: a per-rank mean would be wrong
    batch["caption_lengths"][3:] = 5
    batch["captions"], batch["caption_masks"], _ = synth.make_captions(variant, B, L, K, 0, V, seed)
    for b in range(B):
        n = int(batch["caption_lengths"][b])
        batch["captions"][b, n - 1] = V - 1
        batch["captions"][b, n:] = 0
        batch["captions"][b, 1:n - 1].clamp_(min=1, max=V - 5)
        batch["caption_masks"][b] = 0
    enc_out = synth.make_enc_out(B, seed)
    names = sorted(P)
    n = sum(P[k].numel() for k in names)
    lo, hi = dp.shard(B, rank, world)
    flat = _bucket_for(cfg, P, batch, enc_out, lo, hi, names)
    dp.allreduce_bucket(flat)
    dp.normalise_bucket(flat, n)
    if rank == 0:
        full = _bucket_for(cfg, P, batch, enc_out, 0, B, names)
        dp.normalise_bucket(full, n)
        err = (flat[:n] - full[:n]).abs().max().item()
        naive = None
        torch.save({"err": err, "count": flat[n + 1].item(), "full_count": full[n + 1].item(),
                    "scale": full[:n].abs().max().item()}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_allreduce_equals_full_batch(tmp_path):
    out = str(tmp_path / "res.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r = torch.load(out)
    assert r["count"] == r["full_count"] == 3 * 7 + 3 * 4
    assert r["err"] < 1e-5 * max(1.0, r["scale"]), r


def test_shard_covers_batch():
    sys.path.insert(0, ROOT)
    import ick_amd.dp as dp
    for B in (1, 7, 64, 512):
        for world in (1, 2, 4, 8):
            spans = [dp.shard(B, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))


# ----------------------------------------------------------------------------------------------------------
# train.py's data-parallel plumbing (no HIP kernel involved: buckets, samplers and collectives on the CPU)
# ----------------------------------------------------------------------------------------------------------
def _small_decoder(seed):
    sys.path.insert(0, ROOT)
    import ick_amd
    import ick_amd.synth as synth
    torch.manual_seed(seed)
    m = ick_amd.load_models("knowledge")
    return m.DecoderTransformer(synth.make_word_map(40), 300, 512, 512, 10, 1)


def _plumbing_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    import ick_amd.dp as dp
    from ick_amd.training import TrainStep
    from ick_amd import train as tr
    dp.init_from_env(backend="gloo")
    dec = _small_decoder(seed=100 + rank)          # every process draws its own random initialisation
    before = torch.cat([p.detach().reshape(-1) for p in dec.parameters()]).clone()
    ts = TrainStep(dec, use_graph=False)            # broadcasts rank 0's bucket
    after = torch.cat([p.detach().reshape(-1) for p in dec.parameters()])
    agree = dp.replicas_agree(ts.flat_p)
    ts.flat_p[5] += float(rank)                     # make the replicas differ
    disagree = not dp.replicas_agree(ts.flat_p)
    # samplers: disjoint shards, same number of TRAIN steps on every rank, no VAL sample counted twice
    ds = list(range(23))
    samp = torch.utils.data.distributed.DistributedSampler(ds, num_replicas=world, rank=rank, shuffle=True, seed=7)
    samp.set_epoch(3)
    mine = list(samp)
    val = list(tr.ShardSampler(23, rank, world))
    tot, cnt = dp.reduce_sum_count(10.0 * (rank + 1), 4 + rank)
    torch.save({"changed": not torch.equal(before, after), "params": after, "agree": agree, "disagree": disagree,
                "train_idx": mine, "val_idx": val, "tot": tot, "cnt": cnt}, "%s.%d" % (out, rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_broadcast_samplers_and_validation_reduce(tmp_path):
    out = str(tmp_path / "plumb")
    mp.spawn(_plumbing_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    assert torch.equal(r0["params"], r1["params"])            # identical replicas after TrainStep.__init__
    assert not r0["changed"] and r1["changed"]                # rank 1 took rank 0's weights
    assert r0["agree"] and r1["agree"] and r0["disagree"] and r1["disagree"]
    assert len(r0["train_idx"]) == len(r1["train_idx"]) == 12  # equal step counts (padded by wrap-around)
    assert set(r0["train_idx"]) | set(r1["train_idx"]) == set(range(23))
    assert len(set(r0["train_idx"]) & set(r1["train_idx"])) <= 1
    assert sorted(r0["val_idx"] + r1["val_idx"]) == list(range(23))   # disjoint and complete, no padding
    assert (r0["tot"], r0["cnt"]) == (r1["tot"], r1["cnt"]) == (30.0, 9.0)


def test_fused_step_state_is_torch_adam_layout():
    """TrainStep.state_dict() loads into torch.optim.Adam over the reference's parameter order (and back): the
    checkpoint's `decoder_optimizer` interchanges between the fused step, the fused=False path and the reference."""
    sys.path.insert(0, ROOT)
    from ick_amd.training import TrainStep
    dec = _small_decoder(seed=1)
    dec.fine_tune_embeddings(False)      # a frozen parameter must not shift the indices
    ts = TrainStep(dec, lr=3e-4, use_graph=False)
    g = torch.Generator().manual_seed(0)
    ts.flat_m.copy_(torch.randn(ts.n, generator=g))
    ts.flat_v.copy_(torch.rand(ts.n, generator=g))
    ts.counter.fill_(7)
    sd = ts.state_dict()
    params = [p for p in dec.parameters() if p.requires_grad]      # geo-aware/train.py:85-88
    opt = torch.optim.Adam(params, lr=1.0)
    opt.load_state_dict(sd)
    assert opt.param_groups[0]["lr"] == 3e-4
    for p in params:
        st = opt.state[p]
        assert float(st["step"]) == 7.0 and st["exp_avg"].shape == p.shape
        assert torch.equal(st["exp_avg"], ts._slot(ts.flat_m, p)) and torch.equal(st["exp_avg_sq"], ts._slot(ts.flat_v, p))
    # and back, from an optimizer that really stepped (what a reference-written checkpoint holds)
    for p in params:
        p.grad = torch.randn(p.shape, generator=g)
    opt2 = torch.optim.Adam(params, lr=2e-4)
    snapshot = [p.detach().clone() for p in params]
    opt2.step(); opt2.step()
    with torch.no_grad():
        for p, s0 in zip(params, snapshot):
            p.copy_(s0)
    ts2 = TrainStep(dec, use_graph=False)
    ts2.load_state_dict(opt2.state_dict())
    assert ts2.lr == 2e-4 and int(ts2.counter.item()) == 2
    params2 = [p for p in dec.parameters() if p.requires_grad]
    for p_old, p in zip(params, params2):
        assert torch.equal(ts2._slot(ts2.flat_m, p), opt2.state[p_old]["exp_avg"])
    opt3 = ts2.as_torch_optimizer()
    assert opt3.state_dict()["param_groups"][0]["lr"] == 2e-4
    # moving the module away from the bucket is detected
    import pytest
    from ick_amd.lib import IckError
    dec.word_embedding.weight = torch.nn.Parameter(dec.word_embedding.weight.detach().clone())
    dec.fc_vocab.weight = torch.nn.Parameter(dec.fc_vocab.weight.detach().clone())
    p0 = ts2.params[0]
    p0.data = p0.data.clone()
    with pytest.raises(IckError):
        ts2._check_views()


# ------------------------------------------------------------------------------------------------ round 3
def _state_worker(rank, world, port, out):
    """Each rank builds an Encoder.conv1 stand-in and a module with a frozen parameter + a buffer from its OWN seed; the
    part of the state that lives in the 'bucket' range is skipped, everything else must end as rank 0's copy."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    import ick_amd.dp as dp
    dp.init_from_env(backend="gloo")
    torch.manual_seed(100 + rank)
    enc = torch.nn.Conv2d(8, 4, 1)
    dec = torch.nn.Linear(6, 3)
    dec.weight.requires_grad = False
    dec.register_buffer("pe", torch.randn(5, 3))
    bucket = torch.randn(64)                                   # stands for TrainStep.flat_p
    dec.bias.data = bucket[:3]
    rng = (bucket.data_ptr(), bucket.data_ptr() + 4 * bucket.numel())
    before = dp.module_state_agrees([enc, dec], rng)
    n = dp.broadcast_module_state([enc, dec], rng)
    after = dp.module_state_agrees([enc, dec], rng)
    # raw-byte checksums: an int64 buffer that differs only beyond float32's 24 bits, a bool mask, and a
    # tensor that merely STARTS inside the bucket range but ends outside it is not "covered by the bucket"
    big = dp.replicas_agree(torch.tensor([2 ** 30 + rank, 7], dtype=torch.int64))
    big_same = dp.replicas_agree(torch.tensor([2 ** 30 + 1, 7], dtype=torch.int64))
    flags = dp.replicas_agree(torch.tensor([True, rank == 0, True]))
    outside = dp._outside(torch.empty(0).set_(bucket.untyped_storage(), 60, (4,)), (rng[0], rng[0] + 4 * 62))
    inside = dp._outside(bucket[4:8], rng)
    torch.save({"before": before, "after": after, "n": n, "big": big, "big_same": big_same, "flags": flags,
                "outside": outside, "inside": inside, "conv": enc.weight.detach().clone(), "pe": dec.pe.clone(),
                "w": dec.weight.detach().clone(), "bias": dec.bias.detach().clone()}, out + str(rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_broadcast_of_state_outside_the_bucket(tmp_path):
    out = str(tmp_path / "st")
    mp.spawn(_state_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r0, r1 = torch.load(out + "0"), torch.load(out + "1")
    assert not r0["before"] and not r1["before"]               # per-rank seeds: the replicas start different
    assert r0["after"] and r1["after"] and r0["n"] == r1["n"] == 4   # conv weight, conv bias, frozen weight, buffer
    for k in ("conv", "pe", "w"):
        assert torch.equal(r0[k], r1[k]), k
    assert not torch.equal(r0["bias"], r1["bias"])             # inside the bucket range: left to TrainStep's broadcast
    for r in (r0, r1):
        assert not r["big"] and r["big_same"] and not r["flags"] and r["outside"] and not r["inside"]
