import os, sys, tempfile, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ick_amd.synth as synth
from ick_amd import train as tr, utils as ut
tmp = tempfile.mkdtemp()
data_dir = os.path.join(tmp, "data")
synth.write_dataset(data_dir, "toy", "geo", n_train=24, n_val=8, n_test=4, L=12, K=6, V=60, F=0)
base = dict(variant="geo", data_dir=data_dir, data_name="toy", batch_size=8, workers=0, print_freq=1, fused=True, seed=3)
for d in ("full", "part"):
    os.makedirs(os.path.join(tmp, d))
def W(dec):
    return torch.cat([p.detach().reshape(-1).cpu() for p in dec.parameters()])
torch.manual_seed(0)
tr.main(tr.Config(epochs=2, out_dir=os.path.join(tmp, "full"), **base))
torch.manual_seed(0)
tr.main(tr.Config(epochs=1, out_dir=os.path.join(tmp, "part"), **base))
a = ut.load_checkpoint(os.path.join(tmp, "full", "checkpoint_0_toy.pth.tar"), map_location="cuda")
b = ut.load_checkpoint(os.path.join(tmp, "part", "checkpoint_0_toy.pth.tar"), map_location="cuda")
print("epoch-0 checkpoints: weight diff", (W(a["decoder"]) - W(b["decoder"])).abs().max().item(),
      "enc diff", (W(a["encoder"]) - W(b["encoder"])).abs().max().item())
ma = torch.cat([s["exp_avg"].reshape(-1).cpu() for s in a["decoder_optimizer"].state.values()])
mb = torch.cat([s["exp_avg"].reshape(-1).cpu() for s in b["decoder_optimizer"].state.values()])
print("exp_avg diff", (ma - mb).abs().max().item(), "training flag", a["decoder"].training, b["decoder"].training)
print("dropout p", a["decoder"].transformer_decoder.layers[0].dropout.p, b["decoder"].pos_encoder.dropout.p)
tr.main(tr.Config(epochs=2, out_dir=os.path.join(tmp, "part"), checkpoint=os.path.join(tmp, "part", "checkpoint_0_toy.pth.tar"), **base))
