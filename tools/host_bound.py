#!/usr/bin/env python3
"""Is the train-step loop bound by the host (Python issue time) or by the GPU?  Issues N steps without a
synchronisation and reports the time until the host is done issuing vs the time until the GPU is done."""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import ick_amd  # noqa: E402
import ick_amd.synth as synth  # noqa: E402
from ick_amd.training import TrainStep  # noqa: E402

cfg = dict(synth.CONFIGS["cfg2"])
variant, B, L, K, V, Fn = cfg["variant"], cfg["B"], cfg["L"], cfg["K"], cfg["V"], cfg["F"]
m = ick_amd.load_models(variant)
dec = m.DecoderTransformer(synth.make_word_map(V), 300, 512, 512, 10, 3)
dec.load_state_dict(synth.make_params(variant, V, 0), strict=False)
dec = dec.cuda().train()
enc = m.Encoder(emb_dim=300).cuda().eval()
batch = {k: v.cuda() for k, v in synth.make_batch(variant, B, L, K, V, Fn, 100).items()}
feats = synth.make_feats(B, 100).cuda()
import os  # noqa: E402
from ick_amd import ops  # noqa: E402
if os.environ.get("ICK_TIMESTAMPS"):
    ops.stamps_enable()
IN_GRAPH = os.environ.get("ICK_SEPARATE_ENCODER") is None     # conv1 inside the captured step (bench.py's path)
ts = TrainStep(dec, lr=4e-4, grad_clip=5.0, seed=0, encoder=enc if IN_GRAPH else None,
               lazy_update=IN_GRAPH and os.environ.get("ICK_BENCH_EAGER_UPDATE") != "1")     # bench.py's path


def step():
    if IN_GRAPH:
        return ts(batch["captions"], feats, batch["caption_masks"], batch["caption_lengths"], batch["entities"])
    with torch.no_grad():
        e = enc(feats)
    return ts(batch["captions"], e, batch["caption_masks"], batch["caption_lengths"], batch["entities"])


for _ in range(5):
    step()
torch.cuda.synchronize()
N = 100
t0 = time.perf_counter()
for _ in range(N):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host issue %.3f ms/step, total %.3f ms/step" % ((t1 - t0) / N * 1e3, (t2 - t0) / N * 1e3))
# pieces of the host side
t0 = time.perf_counter()
for _ in range(N):
    with torch.no_grad():
        e = enc(feats)
torch.cuda.synchronize()
print("encoder alone   %.3f ms/step" % ((time.perf_counter() - t0) / N * 1e3))

if ops.STAMPS is not None:
    # the stamps of the captured graphs are rewritten by every replay: these are the last step's
    # (names repeat: warm-up run + capture register each stamp twice; keep the captured = later half)
    rep = ops.stamps_report()
    tmax = max(us for _, us in rep)
    last = [(n, us) for n, us in rep if us > tmax - 3500.0]     # the last replay (a step takes ~2.5 ms)
    t0 = last[0][1]
    for name, us in last:
        print("%9.1f us  %s" % (us - t0, name))
