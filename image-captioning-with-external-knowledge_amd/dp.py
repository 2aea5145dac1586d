"""Data-parallel glue: one process per GPU, each rank owns a contiguous shard of
the global minibatch, and the ONLY exchange per step is one all-reduce(sum) of a flat fp32 bucket
[gradient of the SUM of token losses ..., sum of token losses, token count].  Dividing the reduced
gradient by the reduced token count reproduces the reference's single-process token-mean loss over
the global batch (geo-aware/train.py:281,297) -- a per-rank mean followed by an average would not,
because ranks hold different numbers of non-pad tokens.  Backend "nccl" is RCCL over xGMI on ROCm;
the same code runs on "gloo" for the CPU tests."""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from torchrun's environment (no-op for a single process).  Backend: the
    argument, else $ICK_DP_BACKEND, else "nccl" (= RCCL) with a GPU and "gloo" without."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 or dist.is_initialized():
        return world
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend is None:
        backend = os.environ.get("ICK_DP_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    kw = {}
    if backend == "nccl":
        kw["device_id"] = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group(backend=backend, **kw)
    return world


def shard(batch_size, rank, world):
    """Contiguous shard [lo, hi) of the global batch owned by `rank`."""
    per = (batch_size + world - 1) // world
    lo = min(batch_size, rank * per)
    return lo, min(batch_size, lo + per)


def allreduce_bucket(flat, group=None, async_op=False):
    """Sum (a slice of) the flat gradient bucket over all ranks, in place.  async_op: returns the work handle
    (None for a single process) so that the caller can enqueue more device work before waiting."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
        return work if async_op else flat
    return None if async_op else flat


def world_size(group=None):
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def rank(group=None):
    return dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0


def backend_name(group=None):
    """The collective backend in use ("nccl" is RCCL on ROCm), or "none" for a single process."""
    return dist.get_backend(group) if dist.is_available() and dist.is_initialized() else "none"


def broadcast_bucket(flat, group=None, src=0):
    """Every rank takes rank `src`'s copy of a flat bucket (initial parameters), in place."""
    if world_size(group) > 1:
        dist.broadcast(flat, src=src, group=group)
    return flat


def _outside(t, skip):
    """Is the whole storage interval [data_ptr, data_ptr + nbytes) of tensor t outside the byte range (lo, hi) of
    `skip`?  A tensor that straddles an edge of the range counts as outside (it is not covered by the bucket)."""
    if skip is None:
        return True
    lo, hi = skip
    a = t.data_ptr()
    b = a + t.numel() * t.element_size()
    return not (lo <= a and b <= hi)


def module_state(modules, skip_range=None):
    """Every parameter and buffer of `modules` (deduplicated, in module order) whose storage is not inside
    skip_range = (first byte, end byte) -- the flat bucket TrainStep broadcasts and checks itself."""
    seen, out = set(), []
    for m in modules:
        if m is None:
            continue
        for t in list(m.parameters()) + list(m.buffers()):
            if id(t) not in seen and t.numel() > 0 and _outside(t, skip_range):
                seen.add(id(t))
                out.append(t)
    return out


def broadcast_module_state(modules, skip_range=None, group=None, src=0):
    """Every rank takes rank `src`'s copy of the state TrainStep's bucket does not cover: the frozen Encoder.conv1 (each
    process draws its own random initial weights), frozen decoder parameters and buffers.  Without it the ranks would
    project their images with different conv1 weights and the summed gradient would not be the single-process
    full-batch gradient."""
    if world_size(group) <= 1:
        return 0
    n = 0
    for t in module_state(modules, skip_range):
        dist.broadcast(t.data, src=src, group=group)
        n += 1
    return n


def module_state_agrees(modules, skip_range=None, group=None):
    """True when every rank holds bit-identical copies of that state, whatever its dtype (the checksum is taken over
    the raw bytes: int64 buffers such as BatchNorm's num_batches_tracked are not rounded through float32)."""
    if world_size(group) <= 1:
        return True
    ok = True
    for t in module_state(modules, skip_range):
        ok = replicas_agree(t, group) and ok
    return ok


def _raw_words(t):
    """The bytes of t as int64 values 0..255 (any dtype, any size)."""
    return t.detach().contiguous().view(-1).view(torch.uint8).to(torch.int64)


def replicas_agree(flat, group=None):
    """True when every rank holds bit-identical `flat` (compares all-reduced MIN and MAX of two 64-bit checksums of the
    raw bytes; cheap enough to call once per epoch)."""
    if world_size(group) <= 1:
        return True
    t = flat.detach().contiguous().view(-1)
    if t.element_size() % 4 == 0:
        bits = t.view(torch.int32).to(torch.int64)          # 32-bit words: a quarter of the byte form's work
    else:
        bits = _raw_words(t)
    idx = torch.arange(1, bits.numel() + 1, device=bits.device, dtype=torch.int64)
    cs = torch.stack([(bits * (idx % 1000003)).sum(), (bits * (idx % 998244353 + 7)).sum()])
    lo, hi = cs.clone(), cs.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    return bool((lo == hi).all().item())


def time_all_reduce(flat, group=None, repeats=3):
    """Milliseconds one all_reduce(SUM) of `flat` takes (median of `repeats` after a warm-up, MAX over ranks so that
    every rank holds the same number).  The bucket's contents are restored (the sum of equal copies would scale it)."""
    import time
    if world_size(group) <= 1:
        return 0.0
    keep = flat.clone()
    sync = torch.cuda.synchronize if flat.is_cuda else (lambda: None)
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    times = []
    for _ in range(repeats):
        sync()
        dist.barrier(group=group)
        t0 = time.perf_counter()
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        sync()
        times.append(time.perf_counter() - t0)
    flat.copy_(keep)
    t = torch.tensor([sorted(times)[len(times) // 2] * 1e3], dtype=torch.float64, device=flat.device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


def reduce_sum_count(loss_sum, count, device=None, group=None):
    """Global (sum, count) of a weighted mean held as per-rank partial sums (validation loss): every rank gets the
    same two numbers, so every rank takes the same lr-decay / early-stop / best-checkpoint decision."""
    if world_size(group) <= 1:
        return float(loss_sum), float(count)
    t = torch.tensor([float(loss_sum), float(count)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return float(t[0]), float(t[1])


def normalise_bucket(flat, n):
    """Global token-mean: divide the n gradient entries by the reduced token count (CPU/any device)."""
    flat[:n] /= flat[n + 1]
    return flat
