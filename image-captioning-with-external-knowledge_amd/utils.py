"""Host-side helpers around the hot path, mirroring the parts of the reference's utils.py that
train.py / eval.py call (geo-aware/utils.py:8-97 checkpoint + learning helpers, :104-145 embeddings,
:153-192 the entity-name <-> integer codec).  The checkpoint is the reference's layout: one dict of LIVE
objects {epoch, epochs_since_improvement, loss, encoder, decoder, encoder_optimizer, decoder_optimizer}
pickled with torch.save, named checkpoint_<epoch>_<data_name>.pth.tar on even epochs, checkpoint_<data_name>
.pth.tar otherwise, plus a BEST_ copy when the validation loss improved."""
import math
import os

import torch

DUMMY_CHAR = 124          # ord('|'): pads entity-name encodings (geo-aware/utils.py:151)
NAME_LEN = 50


def checkpoint_name(data_name, epoch):
    return "checkpoint_%s%s.pth.tar" % ("%d_" % epoch if epoch % 2 == 0 else "", data_name)


def save_checkpoint(data_name, epoch, epochs_since_improvement, encoder, decoder, encoder_optimizer,
                    decoder_optimizer, loss, is_best, out_dir="."):
    state = dict(epoch=epoch, epochs_since_improvement=epochs_since_improvement, loss=loss, encoder=encoder,
                 decoder=decoder, encoder_optimizer=encoder_optimizer, decoder_optimizer=decoder_optimizer)
    if hasattr(decoder, "__dict__"):
        # device-side caches and pinned staging buffers are not part of a checkpoint
        for k in ("_graphs", "_kv_pack", "_pred_wt_cache", "_len_pin", "_idx_pin", "_plist", "_pin_ev", "_dec_pack"):
            decoder.__dict__.pop(k, None)
    # the packed weight copies of the row-chain kernels stay alive (captured training graphs write into them by
    # address) but are not pickled either
    stash = {}
    if hasattr(decoder, "__dict__"):
        stash = {k: decoder.__dict__.pop(k) for k in ("_chain_cache", "_chain_cache_bwd", "_chain_ok", "_chain_bwd_ok")
                 if k in decoder.__dict__}
    name = checkpoint_name(data_name, epoch)
    try:
        torch.save(state, os.path.join(out_dir, name))
        if is_best:
            torch.save(state, os.path.join(out_dir, "BEST_" + name))
    finally:
        if stash:
            decoder.__dict__.update(stash)
    return os.path.join(out_dir, name)


def load_checkpoint(path, map_location=None):
    """Whole-object pickles need weights_only=False (torch >= 2.6 defaults to True); `models` must already
    resolve to a drop-in module for reference-written files."""
    return torch.load(path, map_location=map_location, weights_only=False)


class AverageMeter:
    """Latest value, running sum / count / mean of a metric (weighted updates)."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


def clip_gradient(optimizer, grad_clip):
    """Elementwise clamp of every gradient the optimizer owns to [-grad_clip, grad_clip]."""
    for group in optimizer.param_groups:
        for p in group["params"]:
            if p.grad is not None:
                p.grad.clamp_(-grad_clip, grad_clip)


def adjust_learning_rate(optimizer, shrink_factor, verbose=True):
    for group in optimizer.param_groups:
        group["lr"] *= shrink_factor
    if verbose:
        print("\nDECAYING learning rate: now %f\n" % optimizer.param_groups[0]["lr"])


def init_embedding(embeddings):
    """U(-sqrt(3/d), sqrt(3/d)) for rows without a pre-trained vector."""
    bound = math.sqrt(3.0 / embeddings.size(1))
    return torch.nn.init.uniform_(embeddings, -bound, bound)


def load_embeddings(emb_file, word_map):
    """GloVe-format text file -> (len(word_map), d) tensor in word_map order."""
    with open(emb_file) as f:
        d = len(f.readline().rstrip().split(" ")) - 1
    emb = init_embedding(torch.empty(len(word_map), d))
    with open(emb_file) as f:
        for line in f:
            parts = line.split()
            idx = word_map.get(parts[0])
            if idx is not None and len(parts) == d + 1:
                emb[idx] = torch.tensor([float(x) for x in parts[1:]])
    return emb


def str_to_int(text):
    codes = [ord(ch) for ch in text[:NAME_LEN]]
    return codes + [DUMMY_CHAR] * (NAME_LEN - len(codes))


def int_to_str(codes, length):
    return "".join(chr(c) for c in list(codes)[:length])
