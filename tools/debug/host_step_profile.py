#!/usr/bin/env python3
"""Where does the host spend a fused train step?  cProfile over 200 steps (no synchronisation in the loop) + the bare
replay times of the two graphs."""
import cProfile, pstats, sys, time
import torch
sys.path.insert(0, __file__.rsplit("/", 3)[0])
import ick_amd, ick_amd.synth as synth
from ick_amd.training import TrainStep
cfg = dict(synth.CONFIGS["cfg2"]); variant, B, L, K, V, Fn = cfg["variant"], cfg["B"], cfg["L"], cfg["K"], cfg["V"], cfg["F"]
m = ick_amd.load_models(variant)
dec = m.DecoderTransformer(synth.make_word_map(V), 300, 512, 512, 10, 3)
dec.load_state_dict(synth.make_params(variant, V, 0), strict=False); dec = dec.cuda().train()
enc = m.Encoder(emb_dim=300).cuda().eval()
b = {k: v.cuda() for k, v in synth.make_batch(variant, B, L, K, V, Fn, 100).items()}
feats = synth.make_feats(B, 100).cuda()
ts = TrainStep(dec, lr=4e-4, grad_clip=5.0, seed=0, encoder=enc)
step = lambda: ts(b["captions"], feats, b["caption_masks"], b["caption_lengths"], b["entities"])
for _ in range(5):
    step()
torch.cuda.synchronize()
N = 200
t0 = time.perf_counter()
for _ in range(N):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host issue %.3f ms/step, total %.3f ms/step" % ((t1 - t0) / N * 1e3, (t2 - t0) / N * 1e3))
ga, static, gb, ga2 = next(iter(ts._graphs.values()))
for name, g in (("graph A", ga), ("graph B", gb)):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        g.replay()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print("%s replay: host %.3f ms" % (name, (t1 - t0) / 50 * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
