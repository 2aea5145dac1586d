"""Phase timing of the matrix-core attention kernels (diagnostic): builds a copy of the library with -DICK_ATTN_STAMPS,
runs a few cfg2 train steps eagerly and prints, for workgroup (0,0) of the last launch of each kernel variant (self /
cross, forward / backward), the shader-clock deltas between its phase stamps (2.4 GHz ticks -> us)."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ick_amd.build as b  # noqa: E402
import ick_amd.lib as L  # noqa: E402

dbg = os.path.join(ROOT, "gpurun_out", "libick_amd_astamps.so")
subprocess.check_call([b.HIPCC] + b.FLAGS + ["-DICK_ATTN_STAMPS", "-shared", "-o", dbg] + b.sources())
L.LIB_PATH = dbg
import torch  # noqa: E402
import ick_amd  # noqa: E402
import ick_amd.synth as synth  # noqa: E402
from ick_amd.training import TrainStep  # noqa: E402

cfg = dict(synth.CONFIGS["cfg2"])
variant, B, Lc, K, V, Fn = cfg["variant"], cfg["B"], cfg["L"], cfg["K"], cfg["V"], cfg["F"]
m = ick_amd.load_models(variant)
dec = m.DecoderTransformer(synth.make_word_map(V), 300, 512, 512, 10, 3)
dec.load_state_dict(synth.make_params(variant, V, 0), strict=False)
dec = dec.cuda().train()
enc = m.Encoder(emb_dim=300).cuda().eval()
batch = {k: v.cuda() for k, v in synth.make_batch(variant, B, Lc, K, V, Fn, 100).items()}
feats = synth.make_feats(B, 100).cuda()
ts = TrainStep(dec, lr=4e-4, grad_clip=5.0, seed=0, encoder=enc, use_graph=False)
for _ in range(3):
    ts(batch["captions"], feats, batch["caption_masks"], batch["caption_lengths"], batch["entities"])
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 64)()
lib = ctypes.CDLL(dbg)
assert lib.ick_debug_read_attn_stamps(buf) == 0
FWD = ["loads issued + Q staged", "barrier", "-", "S^T + row max", "V^T staged + barrier", "exp + P V", "barrier",
       "combine + store"]
BWD = ["all loads issued", "Q, dO, lse, D staged", "barrier", "S, dP, dS, dV, dK, dQ share per key tile", "barrier",
       "partial dQ tiles summed + stored"]
names = {0: ("self fwd", FWD), 1: ("cross fwd", FWD), 2: ("self bwd", BWD), 3: ("cross bwd", BWD)}
for k, (nm, ph) in names.items():
    t = [buf[k * 16 + i] for i in range(len(ph) + 1)]
    print("%-10s total %6.2f us: " % (nm, (t[-1] - t[0]) / 2400.0) +
          ", ".join("%s %.2f" % (p, (t[i + 1] - t[i]) / 2400.0) for i, p in enumerate(ph)))
