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
