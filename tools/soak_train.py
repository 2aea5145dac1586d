#!/usr/bin/env python3
"""Soak: N fused train steps on the cfg2 batch; prints loss trajectory, step time and device memory."""
import sys, time
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import ick_amd, ick_amd.synth as synth
from ick_amd.training import TrainStep
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
cfg = dict(synth.CONFIGS["cfg2"]); variant, B, L, K, V, Fn = cfg["variant"], cfg["B"], cfg["L"], cfg["K"], cfg["V"], cfg["F"]
m = ick_amd.load_models(variant)
dec = m.DecoderTransformer(synth.make_word_map(V), 300, 512, 512, 10, 3)
dec.load_state_dict(synth.make_params(variant, V, 0), strict=False); dec = dec.cuda().train()
enc = m.Encoder(emb_dim=300).cuda().eval()
batches = [{k: v.cuda() for k, v in synth.make_batch(variant, B, L, K, V, Fn, 100 + i).items()} for i in range(4)]
feats = [synth.make_feats(B, 100 + i).cuda() for i in range(4)]
ts = TrainStep(dec, lr=4e-4, grad_clip=5.0, seed=0)
losses = []
t0 = time.perf_counter()
for i in range(N):
    b = batches[i % 4]
    with torch.no_grad():
        e = enc(feats[i % 4])
    loss = ts(b["captions"], e, b["caption_masks"], b["caption_lengths"], b["entities"])
    if i % (N // 10) == 0 or i == N - 1:
        losses.append((i, loss.item(), torch.cuda.memory_allocated() >> 20))
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("steps %d, %.3f ms/step (incl. %d host syncs)" % (N, dt / N * 1e3, len(losses)))
for i, l, mem in losses:
    print("step %5d loss %.4f  allocated %d MiB" % (i, l, mem))
assert all(l == l and l < 1e4 for _, l, _ in losses), "loss diverged"
assert losses[-1][1] < losses[0][1], "loss did not decrease on the 4 repeated batches"
