#!/usr/bin/env python3
"""What the optimizer kernel's re-laid-out copies cost: the update alone on the flat kernel, on the tiled cover without any
image, and with every image, on cfg2's bucket (HIP events, median of 20).   python tools/debug/adam_derive_bench.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ick_amd  # noqa: E402
import ick_amd.lib as L  # noqa: E402
import ick_amd.ops as ops  # noqa: E402
import ick_amd.synth as synth  # noqa: E402
from ick_amd.training import DerivedWeights, TrainStep  # noqa: E402

c = synth.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "cfg2"]
m = ick_amd.load_models(c["variant"])
dec = m.DecoderTransformer(synth.make_word_map(c["V"]), 300, 512, 512, 10, 3)
dec.load_state_dict(synth.make_params(c["variant"], c["V"], 0), strict=False)
ts = TrainStep(dec.cuda().train(), lr=4e-4)
ts.derived = dw = DerivedWeights.build(ts)
dw.refresh()
ts.flat_g.normal_()
ts.flat_g[ts.n + 1] = 100.0
ts.counter.fill_(3)


def timed(fn, n=20):
    out = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 1e3)
    return sorted(out)[n // 2]


print("bucket %.1f M floats, %d items, %d blocks, images %.1f MB" % (ts.n / 1e6, dw.n_items, dw.n_blocks, (dw.nbytes - 28 * ts.n) / 1e6))
print("tiled cover + every image : %6.1f us" % timed(ts._adam))
# the same cover with the image pointers cleared
raw = bytearray(dw.items_dev.cpu().numpy().tobytes())
items = (L.AdamItem * dw.n_items).from_buffer(raw)
for it in items:
    it.pack = it.pack_t = it.copy = it.ps = it.ps_t = it.tr = None
keep = dw.items_dev
dw.items_dev = torch.frombuffer(bytearray(bytes(items)), dtype=torch.uint8).cuda()
print("tiled cover, no image     : %6.1f us" % timed(ts._adam))
dw.items_dev = keep
d = ts.derived
ts.derived = None
print("flat kernel               : %6.1f us" % timed(ts._adam))
ts.derived = d
