"""Row f4 + fine_tune_encoder: the ResNet-101 trunk in front of Encoder.conv1, conv1's own
backward, and the gradient the decoder sends into the image memory rows (geo-aware/models.py:24-47,
geo-aware/train.py:93-100,282-294)."""
import pytest
import torch

import ick_amd
import ick_amd.synth as synth
from helpers import case_from_golden, load_golden, t
from test_forward_gpu import build_decoder
from test_ops_gpu import close, rnd
from test_training_gpu import reference_loss

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["fwd_encgrad_geo", "fwd_encgrad_knowledge"])
def test_encoder_out_gradient_vs_reference_golden(name):
    """d loss / d encoder_out as the real reference's autograd computes it (what fine_tune_encoder=True feeds conv1)."""
    g = load_golden(name)
    cfg, P, wm, batch, enc_out = case_from_golden(g)
    dec = build_decoder(cfg.variant, cfg.vocab_size, P)
    enc = enc_out.cuda().requires_grad_(True)
    args = [batch["captions"].cuda(), enc, batch["caption_masks"].cuda(), batch["caption_lengths"].cuda(), batch["entities"]]
    if "facts" in batch:
        args.append(batch["facts"].cuda())
    scores, caps, dl = dec(*args)
    loss = reference_loss(scores, caps, dl, wm["<pad>"])
    assert abs(loss.item() - float(g["loss"][0])) < 2e-5
    loss.backward()
    ref = t(g["encoder_out_grad"])
    assert enc.grad is not None and enc.grad.shape == ref.shape
    err = (enc.grad.cpu() - ref).abs().max().item()
    assert err < 2e-5 * max(1.0, ref.abs().max().item()), err
    # the parameter gradients are unchanged by the extra output
    named = dict(dec.named_parameters())
    for k in g:
        if k.startswith("grad::"):
            r = t(g[k])
            assert (named[k[6:]].grad.cpu() - r).abs().max().item() < 2e-5 * max(1.0, r.abs().max().item()), k


def test_conv1_backward_vs_torch_conv2d():
    B, seed = 3, 4
    m = ick_amd.load_models("geo")
    enc = m.Encoder(emb_dim=300)
    w, b = synth.make_conv1(seed)
    with torch.no_grad():
        enc.conv1.weight.copy_(w)
        enc.conv1.bias.copy_(b)
    enc = enc.cuda()
    feats = synth.make_feats(B, seed)
    dy = rnd(B, 300, 196, seed=9)
    fx = feats.cuda().requires_grad_(True)
    out = enc(fx)
    assert out.shape == (B, 300, 196) and out.requires_grad
    (out * dy.cuda()).sum().backward()
    conv = torch.nn.Conv2d(2048, 300, 1).double()
    with torch.no_grad():
        conv.weight.copy_(w.double())
        conv.bias.copy_(b.double())
    fr = feats.double().requires_grad_(True)
    (conv(fr).view(B, 300, -1) * dy.double()).sum().backward()
    close(out, conv(feats.double()).view(B, 300, -1).detach(), 2e-5, "conv1 fwd")
    close(enc.conv1.weight.grad, conv.weight.grad, 2e-5, "conv1 dW")
    close(enc.conv1.bias.grad, conv.bias.grad, 2e-5, "conv1 db")
    close(fx.grad, fr.grad, 2e-5, "conv1 dX")


def test_encoder_with_resnet_trunk_on_images():
    """encoder(image) as eval.py calls it (geo-aware/eval.py:77): 256x256 images -> 8x8x2048 -> adaptive pool 14x14 ->
    conv1 -> (B, 300, 196); the trunk is stock torch (MIOpen), conv1 the HIP GEMM."""
    torch.manual_seed(0)
    m = ick_amd.load_models("geo")
    enc = m.Encoder(emb_dim=300, with_trunk=True).cuda().eval()
    assert sum(p.numel() for p in enc.resnet.parameters()) == 42500160          # torchvision resnet101 minus fc
    imgs = torch.rand(2, 3, 256, 256, device="cuda")
    with torch.no_grad():
        out = enc(imgs)
        feats = enc.adaptive_pool(enc.resnet(imgs))
        assert feats.shape == (2, 2048, 14, 14)
        again = enc(feats)
        ref = torch.nn.functional.conv2d(feats.double().cpu(), enc.conv1.weight.double().cpu(), enc.conv1.bias.double().cpu())
    # (a randomly initialised trunk in eval mode has no normalisation: activations are large; MIOpen may pick another
    # convolution algorithm on the second call, so the two feature maps agree to rounding only)
    assert out.shape == (2, 300, 196)
    close(out, again.double().cpu(), 1e-3, "two passes through the trunk")
    close(again, ref.view(2, 300, -1), 2e-5, "trunk + conv1")
    # fine_tune(): blocks 2-4 only (children 5..7), conv1 stays trainable
    enc.fine_tune(True)
    kids = list(enc.resnet.children())
    assert all(not p.requires_grad for c in kids[:5] for p in c.parameters())
    assert all(p.requires_grad for c in kids[5:] for p in c.parameters()) and enc.conv1.weight.requires_grad
    enc.fine_tune(False)
    assert all(not p.requires_grad for p in enc.resnet.parameters())
    # lazily built trunk; and a decoder consumes the output
    lazy = m.Encoder(emb_dim=300).cuda().eval()
    assert "resnet" not in lazy._modules
    with torch.no_grad():
        assert lazy(imgs).shape == (2, 300, 196) and "resnet" in lazy._modules
    with pytest.raises(ick_amd.lib.IckError):
        m.Encoder(emb_dim=300, with_trunk=False).cuda()(imgs)


def test_train_main_with_fine_tune_encoder(tmp_path):
    """fine_tune_encoder=True: conv1 gets an Adam of its own (lr 1e-4), is updated, and its optimizer is saved."""
    from ick_amd import train as tr, utils as ut
    data_dir = str(tmp_path / "data")
    synth.write_dataset(data_dir, "toy", "geo", n_train=16, n_val=8, n_test=4, L=10, K=6, V=60, F=0)
    torch.manual_seed(0)
    cfg = tr.Config(variant="geo", data_dir=data_dir, data_name="toy", epochs=2, batch_size=8, workers=0, print_freq=1000,
                    fine_tune_encoder=True, out_dir=str(tmp_path))
    hist = tr.main(cfg)
    assert hist[-1][0] < hist[0][0]
    ck0 = ut.load_checkpoint(str(tmp_path / "checkpoint_0_toy.pth.tar"), map_location="cuda")
    ck1 = ut.load_checkpoint(str(tmp_path / "checkpoint_toy.pth.tar"), map_location="cuda")
    assert isinstance(ck1["encoder_optimizer"], torch.optim.Adam)
    assert ck1["encoder_optimizer"].param_groups[0]["lr"] == 1e-4
    w0, w1 = ck0["encoder"].conv1.weight, ck1["encoder"].conv1.weight
    assert (w0 - w1).abs().max().item() > 1e-5                  # conv1 moved between the epochs
