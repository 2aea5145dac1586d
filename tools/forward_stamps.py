#!/usr/bin/env python3
"""Device-timestamp timeline (ops.stamp, 100 MHz device clock, no profiler) of the captured cfg2 forward pass and of the
cfg5 greedy decode's prefill, with Encoder.conv1 inside the graph (attach_encoder) or in front of it
(ICK_BENCH_SEPARATE_ENCODER=1).   usage: python tools/forward_stamps.py"""
import os
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import ick_amd  # noqa: E402
import ick_amd.synth as synth  # noqa: E402
from ick_amd import ops  # noqa: E402

cfg = dict(synth.CONFIGS["cfg2"])
variant, B, L, K, V, Fn = cfg["variant"], cfg["B"], cfg["L"], cfg["K"], cfg["V"], cfg["F"]
m = ick_amd.load_models(variant)
dec = m.DecoderTransformer(synth.make_word_map(V), 300, 512, 512, 10, 3)
dec.load_state_dict(synth.make_params(variant, V, 0), strict=False)
dec = dec.cuda().eval()
enc = m.Encoder(emb_dim=300).cuda().eval()
batch = {k: v.cuda() for k, v in synth.make_batch(variant, B, L, K, V, Fn, 100).items()}
feats = synth.make_feats(B, 100).cuda()
ops.stamps_enable()
fused = os.environ.get("ICK_BENCH_SEPARATE_ENCODER") != "1"
if fused:
    dec.attach_encoder(enc)
with torch.no_grad():
    for _ in range(6):
        img = feats if fused else enc(feats)
        dec(batch["captions"], img, batch["caption_masks"], batch["caption_lengths"], batch["entities"])
torch.cuda.synchronize()
rep = ops.stamps_report()
tmax = max(us for _, us in rep)
last = [(n, us) for n, us in rep if us > tmax - 1500.0]
t0 = last[0][1]
print("cfg2 forward, Encoder.conv1 %s the captured graph" % ("inside" if fused else "in front of"))
for name, us in last:
    print("%9.1f us  %s" % (us - t0, name))
