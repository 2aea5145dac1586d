import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
import ick_amd.synth as synth
from oracle import restatement as R
from test_forward_gpu import build_decoder
from test_training_gpu import zero_dropout
from test_bench_sizes_gpu import reference_train_step
from ick_amd.training import TrainStep
c=synth.CONFIGS["cfg4"]; variant,B,L,K,V,Fn=c["variant"],c["B"],c["L"],c["K"],c["V"],c["F"]; seed=41
P=synth.make_params(variant,V,seed); cfg=R.config_from_word_map(variant,synth.make_word_map(V))
batch=synth.make_batch(variant,B,L,K,V,Fn,seed); enc=synth.make_enc_out(B,seed)
loss_ref,grads_ref,_=reference_train_step(cfg,P,batch,enc)
dec=zero_dropout(build_decoder(variant,V,P).train())
for graph in (True, False):
    ts=TrainStep(dec,lr=0.0,grad_clip=5.0,use_graph=graph)
    args=[batch["captions"].cuda(),enc.cuda(),batch["caption_masks"].cuda(),batch["caption_lengths"].cuda(),batch["entities"],batch["facts"].cuda()]
    loss=ts(*args)
    named=dict(dec.named_parameters())
    rows=[]
    for k,gr in grads_ref.items():
        mine=ts.grads[id(named[k])].detach().cpu(); gr=gr.clamp(-5,5)
        d=(mine-gr).abs()
        rows.append((d.max().item()/max(1e-3,gr.abs().max().item()),k,d.max().item(),gr.abs().max().item(),(d.double().norm()/gr.double().norm()).item()))
    print("graph",graph,"loss",loss.item(),loss_ref)
    for r in sorted(rows,reverse=True)[:8]: print("  %.2e %s abs %.2e max|g| %.2e relnorm %.2e"%r)
    k="transformer_decoder.layers.0.linear1.weight"
    mine=ts.grads[id(named[k])].detach().cpu(); d=(mine-grads_ref[k]).abs()
    print("  rows with err>1e-6:", (d.max(dim=1).values>1e-6).nonzero().view(-1).tolist()[:20], "row max", d.max(dim=1).values.topk(3))
    k2="transformer_decoder.layers.0.linear2.weight"
    mine=ts.grads[id(named[k2])].detach().cpu(); d=(mine-grads_ref[k2]).abs()
    print("  linear2 cols with err>1e-6:", (d.max(dim=0).values>1e-6).nonzero().view(-1).tolist()[:20])
