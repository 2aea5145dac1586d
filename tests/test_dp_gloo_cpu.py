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
