"""ICK_KV_SPLIT_LAYERS=1 (layer 0's image K/V in front of the decoder, the later layers' on the side stream) gives the same
scores as the single projection (cfg2 forward from the feature map)."""
import os, sys, torch
ROOT = __file__.rsplit("/", 3)[0]
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import ick_amd, ick_amd.synth as synth
from test_forward_gpu import build_decoder
from test_bench_sizes_gpu import make_encoder
variant, B, L, K, V, seed = "geo", 64, 20, 20, 10000, 11
P = synth.make_params(variant, V, seed)
enc, _, _ = make_encoder(seed)
b = synth.make_batch(variant, B, L, K, V, 0, seed)
feats = synth.make_feats(B, seed).cuda()
args = lambda: (b["captions"].cuda(), feats, b["caption_masks"].cuda(), b["caption_lengths"].cuda(), b["entities"].cuda())
outs = []
for env in (None, "1"):
    if env: os.environ["ICK_KV_SPLIT_LAYERS"] = env
    dec = build_decoder(variant, V, P).eval(); dec.attach_encoder(enc)
    with torch.no_grad():
        for _ in range(2): s, c, dl = dec(*args())
    torch.cuda.synchronize(); outs.append(s.clone())
print("split K/V == one GEMM:", torch.equal(outs[0], outs[1]), (outs[0]-outs[1]).abs().max().item())
