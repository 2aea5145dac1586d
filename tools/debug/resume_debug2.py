import os, sys, tempfile, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ick_amd.synth as synth
from ick_amd import train as tr, utils as ut
from ick_amd.training import TrainStep
tmp = tempfile.mkdtemp()
data_dir = os.path.join(tmp, "data")
synth.write_dataset(data_dir, "toy", "geo", n_train=24, n_val=8, n_test=4, L=12, K=6, V=60, F=0)
base = dict(variant="geo", data_dir=data_dir, data_name="toy", batch_size=8, workers=0, print_freq=1000, fused=True, seed=3)
log = []
orig_b2d = tr._batch_to_device
def spy(batch, device, has_facts):
    log.append(("batch", float(batch[1].double().sum()), float(batch[0].double().sum()), float(batch[4].double().sum())))
    return orig_b2d(batch, device, has_facts)
tr._batch_to_device = spy
orig_call = TrainStep.__call__
def call(self, *a, **k):
    pre = float(self.flat_p.double().sum())
    cnt = int(self.counter.item())
    out = orig_call(self, *a, **k)
    log.append(("step", pre, cnt, float(out.item()), float(self.flat_g[:self.n].double().abs().sum())))
    return out
TrainStep.__call__ = call
for run in range(2):
    os.makedirs(os.path.join(tmp, "r%d" % run))
    torch.manual_seed(0)
    log.clear()
    tr.main(tr.Config(epochs=1, out_dir=os.path.join(tmp, "r%d" % run), **base))
    for l in log:
        print(run, l, flush=True)
