"""Checkpoint layout: the reference pickles whole objects (geo-aware/utils.py:32-49), so a
checkpoint written by the REAL reference must unpickle into our drop-in classes once `models` resolves to our
module, and state_dicts must interchange.  Needs /root/reference (authoring container only; skipped on the GPU box)."""
import importlib.util
import os
import sys
import types

import pytest
import torch

import ick_amd
import ick_amd.synth as synth

REF = "/root/reference"
DIRS = {"geo": "geo-aware", "knowledge": "knowledge-aware", "news": "news-knowledge-aware"}

pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference sources not present")


def _load_reference_models(variant):
    sys.modules.setdefault("torchvision", types.ModuleType("torchvision"))
    spec = importlib.util.spec_from_file_location("models", os.path.join(REF, DIRS[variant], "models.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("variant", ["geo", "knowledge", "news"])
def test_reference_checkpoint_unpickles_into_drop_in(tmp_path, variant):
    V = 40
    wm = synth.make_word_map(V)
    ref_models = _load_reference_models(variant)
    saved = sys.modules.get("models")
    try:
        sys.modules["models"] = ref_models                      # the reference pickles classes as models.<Name>
        ref_dec = ref_models.DecoderTransformer(word_map=wm, emb_dim=300, decoder_dim=512, encoder_dim=512,
                                                num_heads=10, num_layers=3)
        ref_dec.load_state_dict(synth.make_params(variant, V, 1), strict=False)
        opt = torch.optim.Adam(ref_dec.parameters(), lr=4e-4)
        path = str(tmp_path / "checkpoint_geo_aware.pth.tar")
        state = {"epoch": 3, "epochs_since_improvement": 1, "loss": 2.5, "encoder": None, "decoder": ref_dec,
                 "encoder_optimizer": None, "decoder_optimizer": opt}              # utils.save_checkpoint's dict
        torch.save(state, path)
        ours = ick_amd.load_models(variant)
        sys.modules["models"] = ours                            # what INTEGRATION.md tells a maintainer to do
        ck = torch.load(path, weights_only=False)
    finally:
        if saved is None:
            sys.modules.pop("models", None)
        else:
            sys.modules["models"] = saved
    dec = ck["decoder"]
    assert type(dec) is ours.DecoderTransformer and dec.variant == variant
    assert ck["epoch"] == 3 and isinstance(ck["decoder_optimizer"], torch.optim.Adam)
    # everything our forward / predict read exists on the unpickled object
    assert dec.vocab_size == V and dec.emb_dim == 300 and dec.num_heads == 10
    assert dec.pos_encoder.pe.shape == (5000, 1, 300) and dec.pos_encoder.dropout.p == 0.1
    assert dec.has_facts == (variant != "geo")
    ref_sd = ref_dec.state_dict()
    sd = dec.state_dict()
    assert list(sd.keys()) == list(ref_sd.keys())
    assert all(torch.equal(sd[k], ref_sd[k]) for k in sd)
    # and a freshly built drop-in accepts the reference's state_dict verbatim
    fresh = ours.DecoderTransformer(wm, 300, 512, 512, 10, 3)
    assert list(fresh.state_dict().keys()) == list(ref_sd.keys())
    fresh.load_state_dict(ref_sd, strict=True)


def test_resnet_trunk_layout_matches_torchvision_keys():
    """The trunk reproduces the module tree of `nn.Sequential(*list(torchvision.models.resnet101().children())[:-2])`
    (geo-aware/models.py:24-27): same state_dict keys / shapes, so reference encoder weights load unchanged."""
    import torch
    from ick_amd.resnet import load_torchvision_state_dict, resnet101_trunk
    trunk = resnet101_trunk()
    sd = {k: v.clone() for k, v in trunk.state_dict().items()}
    assert sum(p.numel() for p in trunk.parameters()) == 44549160 - 2049000     # resnet101 minus its fc layer
    assert sd["0.weight"].shape == (64, 3, 7, 7) and sd["1.running_var"].shape == (64,)
    assert sd["4.0.downsample.0.weight"].shape == (256, 64, 1, 1) and "4.1.downsample.0.weight" not in sd
    assert sd["5.0.conv2.weight"].shape == (128, 128, 3, 3) and trunk[5][0].conv2.stride == (2, 2)
    assert sd["6.22.bn3.weight"].shape == (1024,) and "6.23.conv1.weight" not in sd
    assert sd["7.2.conv3.weight"].shape == (2048, 512, 1, 1)
    # a torchvision-style state dict (children named conv1 / bn1 / layer1..4 / fc) maps onto it
    names = {"0": "conv1", "1": "bn1", "4": "layer1", "5": "layer2", "6": "layer3", "7": "layer4"}
    tv = {names[k.split(".")[0]] + k[k.index("."):]: v + 1 for k, v in sd.items()}
    tv["fc.weight"], tv["fc.bias"] = torch.zeros(1000, 2048), torch.zeros(1000)
    load_torchvision_state_dict(trunk, tv)
    assert torch.equal(trunk.state_dict()["7.2.conv3.weight"], sd["7.2.conv3.weight"] + 1)


def test_row_chain_weight_items_cover_every_launch():
    """Host logic of the row chains (no GPU): the packed-copy items name every Linear a chain launch multiplies with,
    forward and (transposed) backward, and the staged refresh of the captured step partitions them."""
    import ick_amd
    import ick_amd.synth as synth
    for variant, stacks in (("geo", 1), ("knowledge", 2)):
        m = ick_amd.load_models(variant)
        dec = m.DecoderTransformer(synth.make_word_map(60), 300, 512, 512, 10, 3)
        items = dict(dec._chain_items())
        nl = 3
        assert len(items) == (6 * nl - 1) + stacks * (4 * nl - 1)
        d = dec.emb_dim
        for li in range(nl):
            assert items[("d", li, "so")].shape == (d, d) and items[("d", li, "cq")].shape == (d, d)
            assert items[("d", li, "l1")].shape == (512, d) and items[("d", li, "l2")].shape == (d, 512)
            assert (("d", li, "si") in items) == (li > 0)
            assert items[("e", li, "l1")].shape == (512, d)
        assert items[("d", 1, "si")].shape == (3 * d, d)
        # the q-projection item is the first d rows of the packed in_proj weight (a view, not a copy)
        w = dec.transformer_decoder.layers[0].multihead_attn.in_proj_weight
        assert items[("d", 0, "cq")].data_ptr() == w.data_ptr()
        bwd = dict(dec._chain_items_bwd())
        assert set(bwd) == {(a, b, c + "T") for (a, b, c) in items}
        for (a, b, c), v in items.items():
            assert bwd[(a, b, c + "T")].shape == (v.shape[1], v.shape[0])
        first = lambda k: k[0] != "d" or (k[1] == 0 and k[2] in ("so", "cq"))
        assert {k for k in items if first(k)} | {k for k in items if not first(k)} == set(items)
        assert all(k[0] in ("e", "f") or k[1] == 0 for k in items if first(k))


def test_device_caches_never_reach_a_pickle_or_a_deep_copy():
    """Captured graphs, packed and pre-split weight copies live in the modules' __dict__; whole-object checkpoints
    (geo-aware/utils.py:32-46) and deep copies must not carry them, and the live module must keep them (captured graphs
    read those buffers by address)."""
    import copy
    import io
    import ick_amd
    import ick_amd.synth as synth
    m = ick_amd.load_models("knowledge")
    dec = m.DecoderTransformer(synth.make_word_map(60), 300, 512, 512, 10, 3)
    enc = m.Encoder(emb_dim=300)
    for mod in (dec, enc):
        mod.__dict__["_ps_cache"] = {"w": [("key",), torch.zeros(8, dtype=torch.uint8)]}
        mod.__dict__["_graphs"] = {"shape": object()}
    buf = io.BytesIO()
    torch.save({"decoder": dec, "encoder": enc}, buf)
    buf.seek(0)
    back = torch.load(buf, weights_only=False)
    for live, loaded in ((dec, back["decoder"]), (enc, back["encoder"])):
        assert "_ps_cache" in live.__dict__ and "_graphs" in live.__dict__
        assert "_ps_cache" not in loaded.__dict__ and "_graphs" not in loaded.__dict__
        assert list(live.state_dict()) == list(loaded.state_dict())
        assert all(torch.equal(a, b) for a, b in zip(live.state_dict().values(), loaded.state_dict().values()))
    assert "_ps_cache" not in copy.deepcopy(dec).__dict__
