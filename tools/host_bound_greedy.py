#!/usr/bin/env python3
"""Is the greedy decode loop (one hipGraph replay per 20-token decode) bound by the host or by the GPU?"""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import ick_amd  # noqa: E402
import ick_amd.synth as synth  # noqa: E402

cfg = dict(synth.CONFIGS["cfg5"])
variant, B, L, K, V = cfg["variant"], cfg["B"], cfg["L"], cfg["K"], cfg["V"]
m = ick_amd.load_models(variant)
dec = m.DecoderTransformer(synth.make_word_map(V), 300, 512, 512, 10, 3)
dec.load_state_dict(synth.make_params(variant, V, 0), strict=False)
dec = dec.cuda().eval()
enc = m.Encoder(emb_dim=300).cuda().eval()
ents = synth.make_entities(variant, B, K, V, 1).cuda()
feats = synth.make_feats(B, 1).cuda()
beam = int(sys.argv[1]) if len(sys.argv) > 1 else 1


def step():
    e = enc(feats)
    return dec.predict_beam(e, L, ents, beam_size=beam) if beam > 1 else dec.predict(e, L, ents)


for _ in range(5):
    step()
torch.cuda.synchronize()
N = 50
t0 = time.perf_counter()
for _ in range(N):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("beam %d: host issue %.3f ms/decode, total %.3f ms/decode" % (beam, (t1 - t0) / N * 1e3, (t2 - t0) / N * 1e3))
