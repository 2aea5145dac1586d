"""Training / validation loop with the reference's structure (geo-aware/train.py:57-386; the knowledge and
news variants differ only in the extra `facts` tensors) on top of the MI355X path.

    python -m ick_amd.train  (after editing CONFIG below, like the reference's module-level globals), or
    from ick_amd import train; train.main(train.Config(variant="knowledge", data_dir=..., data_name=...))

Differences from the reference, all deliberate: images are precomputed 14x14x2048 feature maps
(datasets.CaptionDataset); `fused=True` (default) replaces Adam + clip_gradient + loss.backward() by
training.TrainStep (same update, one flat bucket, one all-reduce per step under torchrun); `fused=False` keeps the
reference's exact statement sequence (CrossEntropyLoss on pack_padded_sequence, loss.backward() through the HIP
autograd bridge, utils.clip_gradient, torch Adam).  Checkpoints use the reference's layout (utils.save_checkpoint)."""
import json
import os
import time
from dataclasses import dataclass

import numpy as np
import torch
from torch import nn
from torch.nn.utils.rnn import pack_padded_sequence

from . import dp, load_models, ops
from . import utils as ut
from .datasets import CaptionDataset
from .training import TrainStep


@dataclass
class Config:
    variant: str = "geo"
    data_dir: str = "img_caption_data/input_dataset_files/"
    data_name: str = "geo_aware_georic2"
    pretrained_word_embeddings_file: str = ""          # GloVe text file; empty = keep the random init
    emb_dim: int = 300
    decoder_dim: int = 512
    encoder_dim: int = 512
    num_heads: int = 10
    num_layers: int = 3
    start_epoch: int = 0
    epochs: int = 120
    max_epochs_since_improvement: int = 20
    batch_size: int = 4                                # reference: 4 / 4 / 3 (geo / knowledge / news)
    workers: int = 1
    decoder_lr: float = 4e-4
    encoder_lr: float = 1e-4                           # geo-aware/train.py:47
    fine_tune_encoder: bool = False                    # geo-aware/train.py:52; True trains conv1 (+ trunk blocks 2-4)
    grad_clip: float = 5.0
    print_freq: int = 100
    checkpoint: str = ""
    zero_out_epochs_since_improvement: bool = False
    fused: bool = True
    out_dir: str = "."
    max_batches: int = 0                               # >0: stop an epoch early (smoke runs)
    seed: int = 0                                      # shuffling seed shared by all ranks (torchrun)
    prefetch: bool = True                              # fused path: next batch's host-to-device copy on a copy stream
    loader_threads: int = 4                            # TRAIN batches fetched by threads of this process (0: DataLoader
                                                       # worker processes, `workers` of them, as the reference)
    half_features: bool = True                         # float16 feature files travel as float16, widened on the device
    deterministic: object = None                       # True / False: fixed-order reductions on / off; None: what
                                                       # ICK_DETERMINISTIC says at the time main() runs, else the
                                                       # library's current mode


def _batch_to_device(batch, device, has_facts):
    imgs, caps, caplens, capmasks, ent = batch[0], batch[1], batch[2], batch[3], batch[4]
    facts = batch[6].to(device) if has_facts else None
    # entity features stay on the host exactly as in geo-aware/train.py:263-266; the decoder moves them
    imgs = imgs.to(device)
    if imgs.dtype != torch.float32:                     # a float16 feature file: widened on the device
        imgs = imgs.float()
    return imgs, caps.to(device), caplens.to(device), capmasks.to(device), ent, facts


class ThreadedBatches:
    """TRAIN batches fetched and pinned by a few threads of this process, handed out in sampler order.  Worker
    PROCESSES move every 51 MB batch of float16 feature maps through shared memory (measured on the GPU box: 267 ms
    per batch with 4 workers against 37 ms fetched in-process); the copies out of the memory map and into pinned
    memory release the GIL, so threads overlap them.

    Aliasing contract.  With a FENCED consumer (one that sets `fenced = True` and leaves, for every batch it took, an event
    in `release[batch[0].data_ptr()]` after which the batch's feature tensor is no longer read -- the Prefetcher does, its
    event follows the host-to-device copy) the feature maps are written into a ring of recycled pinned blocks: a batch's
    image tensor is then only valid until that event, and is overwritten `depth + 4` batches later.  Without one
    (prefetch=False, list(loader), a batch kept for logging) every batch owns its memory, as a DataLoader's would."""

    def __init__(self, dataset, batch_sampler, threads, depth=None):
        self.dataset, self.batch_sampler, self.threads = dataset, batch_sampler, threads
        self.depth = depth or threads + 1
        # Pinned feature blocks, recycled (fenced consumers only): a batch's 26-51 MB are copied ONCE, out of the memory map
        # straight into pinned memory (round 3 copied them into a fresh numpy block and then again into a freshly allocated
        # pinned tensor).  A block is free again when the consumer's event for it has passed; the ring is deep enough that
        # this wait never triggers in steady state.  The ring pins (depth + 4) x 26-103 MB for the loader's lifetime.
        self.ring, self.ring_at = [], 0
        self.ring_size = self.depth + 4
        self.fenced = False
        self.release = {}      # data_ptr of a ring block -> event after which it may be overwritten (set by the consumer)

    def __len__(self):
        return len(self.batch_sampler)

    def _block(self, n):
        spec = getattr(self.dataset, "batch_block_spec", lambda n_: None)(n)
        if spec is None or not torch.cuda.is_available() or not self.fenced:
            return None
        shape, dt = spec
        tdt = torch.float16 if dt == np.float16 else torch.float32
        if len(self.ring) < self.ring_size:
            blk = torch.empty(shape, dtype=tdt).pin_memory()
            self.ring.append(blk)
            return blk
        blk = self.ring[self.ring_at % self.ring_size]
        self.ring_at += 1
        if tuple(blk.shape) != tuple(shape):          # the epoch's last, smaller batch: a block of its own
            return None
        ev = self.release.pop(blk.data_ptr(), None)
        if ev is not None:
            ev.synchronize()
        return blk

    def _fetch(self, idx, blk):
        batch = self.dataset.fetch_batch(idx, img_out=blk) if blk is not None else self.dataset.fetch_batch(idx)
        # the feature block is the ring's pinned memory; the small fields (a few KB each) travel pageable: pinning them
        # would be a page-locking allocation per tensor and batch
        if blk is not None:
            return batch
        return tuple(t if t.is_pinned() else t.pin_memory() for t in batch)

    def __iter__(self):
        from collections import deque
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(max_workers=self.threads)
        try:
            pending = deque()
            it = iter(self.batch_sampler)
            for idx in it:
                idx = list(idx)
                pending.append(pool.submit(self._fetch, idx, self._block(len(idx))))
                if len(pending) >= self.depth:
                    yield pending.popleft().result()
            while pending:
                yield pending.popleft().result()
        finally:
            pool.shutdown(wait=False, cancel_futures=True)


STATS = {}     # "last_epoch_steps_per_s": optimizer steps per second of the last pipelined training epoch (tools/train_rate.py)


def _batch_hook(batch):
    """Called with every host batch the Prefetcher takes from the loader (tests observe the shards through it)."""


class Prefetcher:
    """The next batch's host-to-device copies run on a copy stream while the current step computes (the loop of
    geo-aware/train.py:263-270 moves each batch synchronously in front of its step: 102.8 MB of fp32 features per 64
    samples, as long over PCIe as the whole fused step on the GPU).  Batches come out of the DataLoader pinned
    (pin_memory=True); float16 feature maps are widened on the device.  Yields the tuples _batch_to_device yields, plus
    the token count (taken from the host copy: no synchronisation)."""

    def __init__(self, loader, device, has_facts):
        # a ThreadedBatches loader recycles its pinned feature blocks for a consumer that fences their reuse (this one)
        self.release = getattr(loader, "release", None)
        if self.release is not None:
            loader.fenced = True
        self.it = iter(loader)
        self.device, self.has_facts = device, has_facts
        self.stream = torch.cuda.Stream(device)
        self.next = None
        self._load()

    def _load(self):
        try:
            batch = next(self.it)
        except StopIteration:
            self.next = None
            return
        _batch_hook(batch)
        n_tok = int((batch[2] - 1).sum())
        with torch.cuda.stream(self.stream):
            imgs = batch[0].to(self.device, non_blocking=True)
            if imgs.dtype != torch.float32:
                imgs = imgs.float()
            caps = batch[1].to(self.device, non_blocking=True)
            caplens = batch[2].to(self.device, non_blocking=True)
            capmasks = batch[3].to(self.device, non_blocking=True)
            ent = batch[4].to(self.device, non_blocking=True)            # the fused step reads them on the device
            facts = batch[6].to(self.device, non_blocking=True) if self.has_facts else None
            ev = torch.cuda.Event()
            ev.record(self.stream)
        if self.release is not None and batch[0].is_pinned():
            self.release[batch[0].data_ptr()] = ev     # the loader's ring may reuse this block once the copy has run
        self.next = ((imgs, caps, caplens, capmasks, ent, facts), n_tok, ev, batch)   # batch: keeps the pinned source alive

    def __iter__(self):
        return self

    def __next__(self):
        if self.next is None:
            raise StopIteration
        tensors, n_tok, ev, _ = self.next
        cur = torch.cuda.current_stream()
        cur.wait_event(ev)
        for t in tensors:
            if t is not None:
                t.record_stream(cur)
        self._load()
        return tensors, n_tok


def packed_loss(criterion, scores, caps_sorted, decode_lengths):
    targets = caps_sorted[:, 1:]
    s = pack_padded_sequence(scores, decode_lengths, batch_first=True).data
    t = pack_padded_sequence(targets, decode_lengths, batch_first=True).data
    return criterion(s, t)


def train(loader, encoder, decoder, criterion, decoder_optimizer, step, epoch, cfg, device, encoder_optimizer=None):
    decoder.train()
    encoder.train()
    batch_time, losses = ut.AverageMeter(), ut.AverageMeter()
    has_facts = decoder.has_facts
    start = t_epoch = time.time()
    if step is not None and cfg.prefetch:
        return _train_fused_pipelined(loader, encoder, step, epoch, cfg, device, has_facts)
    n_steps = 0
    for i, batch in enumerate(loader):
        n_steps = i + 1
        imgs, caps, caplens, capmasks, ent, facts = _batch_to_device(batch, device, has_facts)
        extra = (facts,) if has_facts else ()
        if encoder_optimizer is not None:
            enc = encoder(imgs)                     # fine_tune_encoder: the loss back-propagates into the encoder
        else:
            with torch.no_grad():
                enc = encoder(imgs)
        n_tok = int((caplens - 1).sum())
        if step is not None:                                             # fused HIP step
            loss = step(caps, enc, capmasks, caplens, ent, *extra).item()
        else:                                                            # the reference's statement sequence
            scores, caps_sorted, dl = decoder(caps, enc, capmasks, caplens, ent, *extra)
            loss_t = packed_loss(criterion, scores, caps_sorted, dl)
            decoder_optimizer.zero_grad()
            if encoder_optimizer is not None:
                encoder_optimizer.zero_grad()
            loss_t.backward()
            if cfg.grad_clip is not None:
                ut.clip_gradient(decoder_optimizer, cfg.grad_clip)
                if encoder_optimizer is not None:
                    ut.clip_gradient(encoder_optimizer, cfg.grad_clip)
            decoder_optimizer.step()
            if encoder_optimizer is not None:
                encoder_optimizer.step()
            loss = loss_t.item()
        losses.update(loss, n_tok)
        batch_time.update(time.time() - start)
        start = time.time()
        if i % cfg.print_freq == 0:
            print("Epoch: [%d][%d/%d]\tBatch Time %.3f (%.3f)\tLoss %.4f (%.4f)" %
                  (epoch, i, len(loader), batch_time.val, batch_time.avg, losses.val, losses.avg))
        if cfg.max_batches and i + 1 >= cfg.max_batches:
            break
    STATS["last_epoch_steps_per_s"] = n_steps / max(time.time() - t_epoch, 1e-9)
    return losses.avg


def _train_fused_pipelined(loader, encoder, step, epoch, cfg, device, has_facts):
    """The fused step fed by the Prefetcher: no host synchronisation inside the loop except when a line is printed --
    the token-weighted loss sum stays on the device until the epoch ends.  When the step owns the (frozen) encoder it
    takes the feature map itself and Encoder.conv1 runs inside the captured step."""
    loss_sum = torch.zeros(1, device=device)
    tok_sum, start, t_epoch = 0, time.time(), time.time()
    n = 0
    for i, (tensors, n_tok) in enumerate(Prefetcher(loader, device, has_facts)):
        imgs, caps, caplens, capmasks, ent, facts = tensors
        extra = (facts,) if has_facts else ()
        if step.enc is not None and imgs.dim() == 4 and imgs.shape[1] == step.enc.encoder_dim:
            loss = step(caps, imgs, capmasks, caplens, ent, *extra)
        else:
            with torch.no_grad():
                enc = encoder(imgs)
            loss = step(caps, enc, capmasks, caplens, ent, *extra)
        loss_sum += loss * float(n_tok)
        tok_sum += n_tok
        n = i + 1
        if i % cfg.print_freq == 0:
            now = time.time()
            print("Epoch: [%d][%d/%d]\tBatch Time %.3f\tLoss %.4f (%.4f)" %
                  (epoch, i, len(loader), now - start, loss.item(), loss_sum.item() / max(tok_sum, 1)))
        start = time.time()
        if cfg.max_batches and i + 1 >= cfg.max_batches:
            break
    step.flush()                                       # (lazy_update) the last step's optimizer update, before validation
    avg = loss_sum.item() / max(tok_sum, 1)            # the epoch's one synchronisation
    STATS["last_epoch_steps_per_s"] = n / max(time.time() - t_epoch, 1e-9)
    return avg


def validate(loader, encoder, decoder, criterion, cfg, device):
    """Token-weighted mean loss over the validation split (geo-aware/train.py:317-386).  Under torchrun every rank
    scores its own shard and the (sum, count) pair is all-reduced, so all ranks return the same number and take the
    same best-checkpoint / lr-decay / early-stop decisions."""
    decoder.eval()
    encoder.eval()
    losses = ut.AverageMeter()
    has_facts = decoder.has_facts
    with torch.no_grad():
        for i, batch in enumerate(loader):
            imgs, caps, caplens, capmasks, ent, facts = _batch_to_device(batch, device, has_facts)
            extra = (facts,) if has_facts else ()
            scores, caps_sorted, dl = decoder(caps, encoder(imgs), capmasks, caplens, ent, *extra)
            losses.update(packed_loss(criterion, scores, caps_sorted, dl).item(), sum(dl))
            if cfg.max_batches and i + 1 >= cfg.max_batches:
                break
    total, count = dp.reduce_sum_count(losses.sum, losses.count, device=device)
    return total / max(count, 1.0)


class ShardSampler(torch.utils.data.Sampler):
    """Rank r of w takes every w-th index of the split, in order, WITHOUT padding (validation: no sample may be
    counted twice; the shards may differ in length by one because no collective runs inside the loop)."""

    def __init__(self, n, rank, world):
        self.idx = list(range(rank, n, world))

    def __iter__(self):
        return iter(self.idx)

    def __len__(self):
        return len(self.idx)


def make_loaders(cfg, rank, world, fused=None):
    """TRAIN: one permutation per epoch shared by all ranks (seed + epoch), dealt out in disjoint equal shards --
    DistributedSampler pads by wrapping around so that every rank runs the same number of steps (each step holds a
    collective).  The global batch is cfg.batch_size * world samples.  VAL: disjoint unpadded shards."""
    fused = cfg.fused if fused is None else fused       # main(): fine_tune_encoder turns the fused step off
    data = {s: CaptionDataset(cfg.data_dir, cfg.data_name, s, keep_half=cfg.half_features and fused and cfg.prefetch
                              and s == "TRAIN") for s in ("TRAIN", "VAL")}
    samplers = {"TRAIN": None, "VAL": None}
    if world > 1:
        samplers["TRAIN"] = torch.utils.data.distributed.DistributedSampler(
            data["TRAIN"], num_replicas=world, rank=rank, shuffle=True, seed=cfg.seed, drop_last=False)
        samplers["VAL"] = ShardSampler(len(data["VAL"]), rank, world)
    gen = torch.Generator()
    gen.manual_seed(cfg.seed)
    def loader(split, shuffle):
        ds, smp = data[split], samplers[split]
        if ds.precomputed and ds.transform is None:
            # whole batches: ds[[indices]] builds the batch in one go (datasets.CaptionDataset.fetch_batch)
            base = smp if smp is not None else (torch.utils.data.RandomSampler(ds, generator=gen) if shuffle
                                                else torch.utils.data.SequentialSampler(ds))
            bs = torch.utils.data.BatchSampler(base, cfg.batch_size, drop_last=False)
            if split == "TRAIN" and cfg.loader_threads > 0:
                return ThreadedBatches(ds, bs, cfg.loader_threads)
            return torch.utils.data.DataLoader(ds, sampler=bs, batch_size=None, num_workers=cfg.workers, pin_memory=True)
        return torch.utils.data.DataLoader(ds, batch_size=cfg.batch_size, shuffle=shuffle and smp is None, sampler=smp,
                                           num_workers=cfg.workers, pin_memory=True, generator=gen if shuffle else None)

    loaders = {"TRAIN": loader("TRAIN", True), "VAL": loader("VAL", False)}
    return loaders, samplers, gen


def _bucket_range(step):
    return (step.flat_p.data_ptr(), step.flat_p.data_ptr() + 4 * step.flat_p.numel())


def main(cfg=None):
    cfg = cfg or Config()
    world = dp.init_from_env()
    rank = int(os.environ.get("RANK", "0"))
    if world > 1 and (not cfg.fused or cfg.fine_tune_encoder):
        raise ValueError("data-parallel training needs fused=True and fine_tune_encoder=False (TrainStep owns the "
                         "gradient all-reduce; the encoder's gradients are not part of its bucket)")
    fused = cfg.fused and not cfg.fine_tune_encoder     # fine-tuning the encoder takes the reference's statement
                                                        # sequence: loss.backward() through the HIP autograd bridge
    device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count()))
    torch.cuda.set_device(device)
    models = load_models(cfg.variant)
    with open(os.path.join(cfg.data_dir, "WORDMAP_" + cfg.data_name + ".json")) as f:
        word_map = json.load(f)
    best_loss, epochs_since_improvement, start_epoch = 1e5, 0, cfg.start_epoch
    if not cfg.checkpoint:
        decoder = models.DecoderTransformer(word_map=word_map, emb_dim=cfg.emb_dim, decoder_dim=cfg.decoder_dim,
                                            encoder_dim=cfg.encoder_dim, num_heads=cfg.num_heads,
                                            num_layers=cfg.num_layers)
        if cfg.pretrained_word_embeddings_file:
            decoder.load_pretrained_embeddings(ut.load_embeddings(cfg.pretrained_word_embeddings_file, word_map))
        decoder.fine_tune_embeddings(True)
        # fine-tuning needs the trunk's parameters to exist before encoder.fine_tune() and the encoder optimizer see them
        # (a trunk built lazily by the first raw-image batch would be missing from the optimizer)
        encoder = models.Encoder(emb_dim=cfg.emb_dim, with_trunk=True if cfg.fine_tune_encoder else None)
        decoder_optimizer = encoder_optimizer = None
    else:
        ck = ut.load_checkpoint(cfg.checkpoint, map_location=device)
        decoder, encoder = ck["decoder"], ck["encoder"]
        decoder_optimizer = None if cfg.zero_out_epochs_since_improvement else ck["decoder_optimizer"]
        encoder_optimizer = None if cfg.zero_out_epochs_since_improvement else ck.get("encoder_optimizer")
        if not cfg.zero_out_epochs_since_improvement:
            start_epoch, epochs_since_improvement, best_loss = ck["epoch"] + 1, ck["epochs_since_improvement"], ck["loss"]
    decoder.to(device)
    encoder.to(device)
    if cfg.fine_tune_encoder and "resnet" not in encoder._modules:      # e.g. a checkpoint saved without the trunk
        encoder._build_trunk()
    encoder.fine_tune(cfg.fine_tune_encoder)
    if cfg.fine_tune_encoder and encoder_optimizer is None:
        encoder_optimizer = torch.optim.Adam([p for p in encoder.parameters() if p.requires_grad], lr=cfg.encoder_lr)
    if not cfg.fine_tune_encoder:
        encoder_optimizer = None
    step = None
    if fused:
        # seed: the dropout stream, one per rank (the ranks hold different samples); the constructor broadcasts rank
        # 0's weights, so a decoder that was randomly initialised per process starts identical everywhere
        # encoder=: with the frozen encoder the step takes the feature map itself (Encoder.conv1 inside the captured step)
        det = cfg.deterministic
        if det is None and os.environ.get("ICK_DETERMINISTIC") is not None:
            det = os.environ["ICK_DETERMINISTIC"] not in ("", "0")      # the script's switch, read when main() runs
        step = TrainStep(decoder, lr=cfg.decoder_lr, grad_clip=cfg.grad_clip, seed=cfg.seed * 1000 + rank,
                         encoder=encoder if cfg.prefetch else None, deterministic=det,
                         # the pipelined loop lets step i's optimizer update run at the head of step i + 1's graph beside
                         # Encoder.conv1; _train_fused_pipelined() flushes the last one of an epoch
                         lazy_update=bool(cfg.prefetch))
        if decoder_optimizer is not None:
            # resume: Adam moments, step count (bias correction + dropout stream position) and the decayed lr come
            # back from the pickled optimizer (ours or one written by the reference, geo-aware/utils.py:32-46)
            step.load_state_dict(decoder_optimizer.state_dict())
            decoder_optimizer = None
        # what the bucket does not cover (frozen Encoder.conv1 and trunk, frozen decoder parameters, buffers) also
        # starts from rank 0's copy: every process initialised its own
        dp.broadcast_module_state([encoder, decoder], _bucket_range(step))
    elif decoder_optimizer is None:
        decoder_optimizer = torch.optim.Adam([p for p in decoder.parameters() if p.requires_grad], lr=cfg.decoder_lr)
    criterion = nn.CrossEntropyLoss(ignore_index=word_map["<pad>"]).to(device)
    loaders, samplers, shuffle_gen = make_loaders(cfg, rank, world, fused)
    history = []
    for epoch in range(start_epoch, cfg.epochs):
        if epochs_since_improvement == cfg.max_epochs_since_improvement:
            break
        if epochs_since_improvement > 0 and epochs_since_improvement % 8 == 0:
            if step is not None:
                step.set_lr(step.lr * 0.8)
            else:
                ut.adjust_learning_rate(decoder_optimizer, 0.8)
            if encoder_optimizer is not None:
                ut.adjust_learning_rate(encoder_optimizer, 0.8)
        # the epoch's permutation depends on (seed, epoch) only, so a resumed run sees the batches the interrupted
        # one would have seen
        shuffle_gen.manual_seed(cfg.seed * 100003 + epoch)
        if samplers["TRAIN"] is not None:
            samplers["TRAIN"].set_epoch(epoch)
        tr = train(loaders["TRAIN"], encoder, decoder, criterion, decoder_optimizer, step, epoch, cfg, device,
                   encoder_optimizer)
        last_loss = validate(loaders["VAL"], encoder, decoder, criterion, cfg, device)   # identical on every rank
        is_best = last_loss < best_loss
        best_loss = min(last_loss, best_loss)
        epochs_since_improvement = 0 if is_best else epochs_since_improvement + 1
        history.append((tr, last_loss))
        if step is not None and world > 1 and not (dp.replicas_agree(step.flat_p) and
                                                   dp.module_state_agrees([encoder, decoder], _bucket_range(step))):
            raise RuntimeError("data-parallel replicas diverged (epoch %d): parameters differ between ranks" % epoch)
        if rank == 0:
            opt = step.as_torch_optimizer() if step is not None else decoder_optimizer
            ut.save_checkpoint(cfg.data_name, epoch, epochs_since_improvement, encoder, decoder, encoder_optimizer,
                               opt, last_loss, is_best, out_dir=cfg.out_dir)
    return history


if __name__ == "__main__":
    main()
