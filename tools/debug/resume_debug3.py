import os, sys, tempfile, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ick_amd.synth as synth
from ick_amd import train as tr, utils as ut
from ick_amd.training import TrainStep
tmp = tempfile.mkdtemp()
data_dir = os.path.join(tmp, "data")
synth.write_dataset(data_dir, "toy", "geo", n_train=24, n_val=8, n_test=4, L=12, K=6, V=60, F=0)
base = dict(variant="geo", data_dir=data_dir, data_name="toy", batch_size=8, workers=0, print_freq=1000, fused=True, seed=3,
            max_batches=2)
snaps = {}
orig_call = TrainStep.__call__
def call(self, *a, **k):
    pre = self.flat_p.detach().clone()
    out = orig_call(self, *a, **k)
    torch.cuda.synchronize()
    snaps.setdefault(RUN, []).append((pre.cpu(), self.flat_g[:self.n].detach().clone().cpu(), self.flat_p.detach().clone().cpu(), self))
    return out
TrainStep.__call__ = call
for RUN in range(2):
    os.makedirs(os.path.join(tmp, "r%d" % RUN))
    torch.manual_seed(0)
    tr.main(tr.Config(epochs=1, out_dir=os.path.join(tmp, "r%d" % RUN), **base))
a, b = snaps[0][0], snaps[1][0]
ts = a[3]
print("pre diff", (a[0] - b[0]).abs().max().item(), "g diff", (a[1] - b[1]).abs().max().item(), "post diff", (a[2] - b[2]).abs().max().item())
named = dict(ts.dec.named_parameters())
rows = []
for k, p in named.items():
    off = (p.data_ptr() - ts.flat_p.data_ptr()) // 4
    if 0 <= off < ts.n:
        n = p.numel()
        dpost = (a[2][off:off+n] - b[2][off:off+n]).abs()
        dg = (a[1][off:off+n] - b[1][off:off+n]).abs()
        g = a[1][off:off+n].abs()
        rows.append((dpost.max().item(), k, dg.max().item(), g.max().item(), int((dpost > 1e-4).sum()), n, g[dpost > 1e-4].max().item() if (dpost > 1e-4).any() else 0.0))
for r in sorted(rows, reverse=True)[:25]:
    print("%.3e %-60s dg %.2e gmax %.2e nbig %d/%d gmax@big %.2e" % r)
