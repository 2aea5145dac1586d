"""Is one fused training step reproducible inside one process (same weights, same batch, same dropout seed)?
Runs it three times: fresh, again, and after poisoning the caching allocator's free blocks with NaN."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ick_amd, ick_amd.synth as synth
from ick_amd.training import TrainStep

def poison():
    xs = [torch.full((64 * 1024 * 1024,), float("nan"), device="cuda") for _ in range(8)]
    torch.cuda.synchronize()
    del xs

def run(drop, graph, variant="geo", steps=2):
    B, L, K, V, Fn = 8, 12, 6, 60, (0 if variant == "geo" else 5)
    P = synth.make_params(variant, V, 5)
    m = ick_amd.load_models(variant)
    dec = m.DecoderTransformer(synth.make_word_map(V), 300, 512, 512, 10, 3)
    dec.load_state_dict(P, strict=False)
    dec = dec.cuda().train()
    if not drop:
        for mod in dec.modules():
            if isinstance(mod, torch.nn.Dropout): mod.p = 0.0
            if isinstance(mod, torch.nn.MultiheadAttention): mod.dropout = 0.0
    ts = TrainStep(dec, seed=3, use_graph=graph)
    batch = synth.make_batch(variant, B, L, K, V, Fn, 5)
    enc = synth.make_enc_out(B, 5).cuda()
    args = [batch["captions"].cuda(), enc, batch["caption_masks"].cuda(), batch["caption_lengths"].cuda(), batch["entities"]]
    if variant != "geo": args.append(batch["facts"].cuda())
    out = []
    for _ in range(steps):
        loss = ts(*args).item()
        out.append((loss, ts.flat_g[:ts.n].clone().cpu(), ts.flat_p.clone().cpu()))
    return out, ts

for variant in ("geo", "knowledge"):
    for drop in (False, True):
        for graph in (False, True):
            a, _ = run(drop, graph, variant)
            b, _ = run(drop, graph, variant)
            poison()
            c, ts = run(drop, graph, variant)
            for name, o in (("again", b), ("poisoned", c)):
                msg = []
                for s in range(len(a)):
                    dg = (a[s][1] - o[s][1]).abs()
                    msg.append("step%d loss %.6f/%.6f dg max %.3e (nan %d) dp max %.3e" % (
                        s, a[s][0], o[s][0], dg.max().item(), int(torch.isnan(o[s][1]).sum()), (a[s][2] - o[s][2]).abs().max().item()))
                print(variant, "drop" if drop else "nodrop", "graph" if graph else "eager", name, " | ".join(msg), flush=True)
            # which parameter differs most in the first step's gradient?
            dg = (a[0][1] - c[0][1]).abs()
            if dg.max().item() > 1e-4 or torch.isnan(c[0][1]).any():
                named = dict(ts.dec.named_parameters())
                worst = []
                for k, p in named.items():
                    off = (p.data_ptr() - ts.flat_p.data_ptr()) // 4
                    if 0 <= off < ts.n:
                        seg = dg[off:off + p.numel()]
                        worst.append((float(torch.nan_to_num(seg, nan=1e9).max()), k))
                print("   worst:", sorted(worst, reverse=True)[:6], flush=True)
