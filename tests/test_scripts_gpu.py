"""The callers either side of the hot path: dataset reader over the reference's file
formats, the train / validate loop (fused step and the reference's exact statement sequence), the reference
checkpoint layout, and eval's greedy decode + detokenisation."""
import os

import pytest
import torch

import ick_amd
import ick_amd.synth as synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("variant,fused", [("geo", True), ("knowledge", True), ("geo", False)])
def test_train_validate_checkpoint_eval_roundtrip(tmp_path, variant, fused):
    from ick_amd import eval as ev, train as tr, utils as ut
    from ick_amd.datasets import CaptionDataset
    data_dir = str(tmp_path / "data")
    wm = synth.write_dataset(data_dir, "toy", variant, n_train=24, n_val=8, n_test=4, L=12, K=6, V=60, F=5)
    ds = CaptionDataset(data_dir, "toy", "TRAIN")
    item = ds[0]
    assert len(item) == (8 if variant != "geo" else 6)
    assert item[0].shape == (2048, 14, 14) and item[1].shape == (12,) and item[2].shape == (1,)
    cfg = tr.Config(variant=variant, data_dir=data_dir, data_name="toy", epochs=3, batch_size=8, workers=0,
                    print_freq=1000, fused=fused, out_dir=str(tmp_path))
    torch.manual_seed(0)
    hist = tr.main(cfg)
    assert len(hist) == 3 and all(map(lambda h: h[0] == h[0] and h[1] == h[1], hist))   # finite
    assert hist[-1][0] < hist[0][0]                                                      # training loss goes down
    # reference checkpoint layout: even epochs keep their number, odd epochs overwrite the rolling file
    for name in ("checkpoint_0_toy.pth.tar", "checkpoint_toy.pth.tar", "checkpoint_2_toy.pth.tar"):
        assert os.path.exists(tmp_path / name), name
    assert any(f.startswith("BEST_") for f in os.listdir(tmp_path))
    ck = ut.load_checkpoint(str(tmp_path / "checkpoint_2_toy.pth.tar"), map_location="cuda")
    assert set(ck) == {"epoch", "epochs_since_improvement", "loss", "encoder", "decoder", "encoder_optimizer",
                       "decoder_optimizer"}
    dec, enc = ck["decoder"].cuda().eval(), ck["encoder"].cuda().eval()
    loader = torch.utils.data.DataLoader(CaptionDataset(data_dir, "toy", "TEST"), batch_size=2, shuffle=False)
    caps, seqs = ev.evaluate(enc, dec, loader, wm, max_caption_len=10, out_csv=str(tmp_path / "generated_captions.csv"))
    assert len(caps) == 4 and os.path.exists(tmp_path / "generated_captions.csv")
    Vx = 60 + 6 + (5 if variant != "geo" else 0)
    assert all(0 <= t < Vx for s in seqs for t in s)


def test_detokenize_names_and_cleanup():
    from ick_amd import eval as ev, utils as ut
    wm = synth.make_word_map(10)
    rev = {v: k for k, v in wm.items()}
    ents = torch.tensor([[0, 5] + ut.str_to_int("Paris"), [1, 9] + ut.str_to_int("<unk_ent>")])
    facts = torch.tensor([[0, 4] + ut.str_to_int("1889")])
    seq = [wm["<start>"], 1, 10, 2, 12, wm["<end>"], 0, 0]
    assert ev.detokenize(seq, wm, rev, ents, facts) == "w1 Paris w2 1889"
    assert ev.detokenize([1, 10 + 5], wm, rev, ents, None) == "w1 <unk_ent>"
    assert ut.int_to_str(ut.str_to_int("Eiffel Tower"), 12) == "Eiffel Tower"


def _weights(dec):
    return torch.cat([p.detach().reshape(-1).cpu() for p in dec.parameters()])


def test_fused_resume_equals_uninterrupted_run(tmp_path):
    """Checkpoint -> resume in fused mode restores Adam's moments, the step counter (bias correction and the position
    of the dropout stream), the learning rate and the batch order (reference: decoder_optimizer is pickled and reused,
    geo-aware/utils.py:32-46, train.py:105-129).  Two GPU runs of the same steps are NOT bitwise equal: float atomics
    leave ~1e-7 noise in the gradients and Adam(eps=1e-8) turns noise-level gradient entries (|g| < 1e-6; thousands of
    them in the first layer's in_proj at initialisation) into +-lr steps.  So the step after a resume is compared with
    the uninterrupted step on the well-conditioned entries, with a run that LOSES the optimizer state as the control."""
    from ick_amd import train as tr, utils as ut
    from ick_amd.training import TrainStep
    data_dir = str(tmp_path / "data")
    synth.write_dataset(data_dir, "toy", "geo", n_train=24, n_val=8, n_test=4, L=12, K=6, V=60, F=0)
    base = dict(variant="geo", data_dir=data_dir, data_name="toy", batch_size=8, workers=0, print_freq=1000, fused=True,
                seed=3)
    torch.manual_seed(0)
    tr.main(tr.Config(epochs=1, out_dir=str(tmp_path), **base))
    path = str(tmp_path / "checkpoint_0_toy.pth.tar")
    ck = ut.load_checkpoint(path, map_location="cuda")
    opt = ck["decoder_optimizer"]
    assert isinstance(opt, torch.optim.Adam) and len(opt.state) > 0
    assert {float(st["step"]) for st in opt.state.values()} == {3.0}            # 24 samples / batch 8
    # ---- the step after the checkpoint, three ways, on one fixed batch
    batch = synth.make_batch("geo", 8, 12, 6, 60, 0, 5)
    enc_out = synth.make_enc_out(8, 5).cuda()

    def next_step(restore):
        c = ut.load_checkpoint(path, map_location="cuda")
        dec = c["decoder"].cuda().train()
        ts = TrainStep(dec, lr=4e-4, seed=3000)
        if restore:
            ts.load_state_dict(c["decoder_optimizer"].state_dict())
        before = ts.flat_p.clone()
        ts(batch["captions"].cuda(), enc_out, batch["caption_masks"].cuda(), batch["caption_lengths"].cuda(),
           batch["entities"])
        return (ts.flat_p - before).cpu(), ts.flat_g[:ts.n].clone().cpu(), int(ts.counter.item()), ts

    d1, g1, n1, ts1 = next_step(True)
    d2, g2, n2, _ = next_step(True)
    d3, g3, n3, _ = next_step(False)
    assert n1 == n2 == 4 and n3 == 1
    well = g1.abs() >= 1e-5
    assert well.float().mean().item() > 0.3
    assert (d1 - d2)[well].abs().max().item() < 1e-5           # restored twice: the same update (lr = 4e-4)
    assert (d1 - d3)[well].abs().max().item() > 1e-4           # state lost: a visibly different update
    # the restored moments are the checkpoint's
    sd = ts1.state_dict()
    ref = ck["decoder_optimizer"].state_dict()
    assert sd["param_groups"][0]["lr"] == ref["param_groups"][0]["lr"]
    # ---- and train.main resumes where it stopped: epoch 1, optimizer steps 4..6
    tr.main(tr.Config(epochs=2, out_dir=str(tmp_path), checkpoint=path, **base))
    part = ut.load_checkpoint(str(tmp_path / "checkpoint_toy.pth.tar"), map_location="cuda")
    assert part["epoch"] == 1
    assert {float(st["step"]) for st in part["decoder_optimizer"].state.values()} == {6.0}
    assert part["loss"] < ck["loss"] + 0.2


def test_two_rank_train_main(tmp_path):
    """train.main under torchrun with two ranks (sharing this box's GPU, collectives over gloo): the ranks start from
    different random weights, must see disjoint samples, and must end with bit-identical parameters."""
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    data_dir = str(tmp_path / "data")
    synth.write_dataset(data_dir, "toy", "knowledge", n_train=24, n_val=8, n_test=4, L=12, K=6, V=60, F=5)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, ICK_DP_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(root, "tests", "dp_train_worker.py"), data_dir,
           str(tmp_path)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-3000:]
    f0, f1 = torch.load(tmp_path / "flat0.pt"), torch.load(tmp_path / "flat1.pt")
    assert f0 is not None and torch.equal(f0, f1)                                  # bit-identical replicas
    assert torch.equal(torch.load(tmp_path / "conv0.pt"), torch.load(tmp_path / "conv1.pt"))   # frozen conv1 too
    r0, r1 = torch.load(tmp_path / "rank0.pt"), torch.load(tmp_path / "rank1.pt")
    assert r0["hist"] == r1["hist"] or all(abs(a[1] - b[1]) < 1e-9 for a, b in zip(r0["hist"], r1["hist"]))  # same val loss
    assert len(r0["seen"]) == len(r1["seen"]) > 0                                  # same number of steps on both ranks
    for b0, b1 in zip(r0["seen"], r1["seen"]):                                     # every global batch: disjoint shards
        s0 = {tuple(row.tolist()) for row in b0}
        s1 = {tuple(row.tolist()) for row in b1}
        assert len(s0 & s1) == 0
    assert os.path.exists(tmp_path / "checkpoint_toy.pth.tar") and not os.path.exists(tmp_path / "r1" / "checkpoint_toy.pth.tar")
