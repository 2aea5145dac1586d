#!/usr/bin/env python3
"""Where the host time of the drop-in forward() goes (it synchronises once per call for the length sort)."""
import sys, time
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import ick_amd, ick_amd.synth as synth
cfg = dict(synth.CONFIGS["cfg2"]); variant, B, L, K, V, Fn = cfg["variant"], cfg["B"], cfg["L"], cfg["K"], cfg["V"], cfg["F"]
m = ick_amd.load_models(variant)
dec = m.DecoderTransformer(synth.make_word_map(V), 300, 512, 512, 10, 3)
dec.load_state_dict(synth.make_params(variant, V, 0), strict=False); dec = dec.cuda().eval()
enc = m.Encoder(emb_dim=300).cuda().eval()
batch = {k: v.cuda() for k, v in synth.make_batch(variant, B, L, K, V, Fn, 100).items()}
feats = synth.make_feats(B, 100).cuda()
def step():
    with torch.no_grad():
        e = enc(feats)
        return dec(batch["captions"], e, batch["caption_masks"], batch["caption_lengths"], batch["entities"])
for _ in range(5): step()
torch.cuda.synchronize()
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter()
for _ in range(200): step()
torch.cuda.synchronize()
print("%.3f ms/step" % ((time.perf_counter() - t0) / 200 * 1e3))
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
