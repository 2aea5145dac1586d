"""CaptionDataset over the reference's file formats (geo-aware/datasets.py:10-56, knowledge-aware/datasets.py:10-64):
the batched fetch (fetch_batch = ds[[i, j, ...]], maps copied straight out of the memory-mapped feature file) returns exactly what collating the
per-sample tuples returns, in the reference's field order, for float16 and float32 feature files."""
import numpy as np
import pytest
import torch

import ick_amd.synth as synth
from ick_amd.datasets import CaptionDataset


@pytest.mark.parametrize("variant,keep_half", [("geo", False), ("knowledge", True), ("news", False)])
def test_batched_fetch_equals_per_sample_collate(tmp_path, variant, keep_half):
    d = str(tmp_path)
    synth.write_dataset(d, "t", variant, n_train=20, n_val=4, n_test=2, L=10, K=5, V=50, F=4)
    ds = CaptionDataset(d, "t", "TRAIN", keep_half=keep_half)
    assert len(ds[0]) == (6 if variant == "geo" else 8)                 # (img, caption, caplen, capmask, ent, names[, facts, names])
    idx = [7, 2, 19, 0, 11]
    batch = ds[idx]
    ref = torch.utils.data.default_collate([ds[i] for i in idx])
    assert len(batch) == len(ref)
    for a, b in zip(batch, ref):
        assert a.shape == b.shape and a.dtype == b.dtype and torch.equal(a, b)
    assert batch[0].dtype == (torch.float16 if keep_half else torch.float32)
    bs = torch.utils.data.BatchSampler(torch.utils.data.SequentialSampler(ds), 8, drop_last=False)
    loader = torch.utils.data.DataLoader(ds, sampler=bs, batch_size=None)
    shapes = [tuple(b[0].shape) for b in loader]
    assert shapes == [(8, 2048, 14, 14), (8, 2048, 14, 14), (4, 2048, 14, 14)]
    # the default per-sample loader still works (eval.py's loader, the reference's call pattern)
    plain = torch.utils.data.DataLoader(ds, batch_size=8, shuffle=False)
    assert [tuple(b[0].shape) for b in plain] == shapes
    # threads of this process hand the same batches out, pinned when a GPU is there, in sampler order
    from ick_amd.train import ThreadedBatches
    tb = ThreadedBatches(ds, bs, threads=3)
    tb._fetch = lambda idx, blk=None: ds.fetch_batch(idx)          # no pinning on a CPU-only box
    got = list(tb)
    assert len(got) == len(tb) == 3
    for a, b in zip(got, loader):
        assert all(torch.equal(x, y) for x, y in zip(a, b))
