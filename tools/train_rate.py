"""Steps per second of the REAL training loop (python -m ick_amd.train's main) next to bench.py's resident-data number:
a synthetic dataset in the reference's file formats with float16 14x14x2048 feature maps on local disk, batch 64,
L 20, K 20, V 10 000 (cfg2), one epoch of N optimizer steps through DataLoader workers -> pinned batches -> copy-stream
prefetch -> fused step.  usage (GPU box): python tools/train_rate.py [n_train=1024] [loader_threads=4 (0: 4 DataLoader worker processes)] [prefetch=1] [half=1]"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import ick_amd.synth as synth
from ick_amd import train as tr

n_train = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 4
workers = 4 if threads == 0 else 0
prefetch = (sys.argv[3] != "0") if len(sys.argv) > 3 else True
half = (sys.argv[4] != "0") if len(sys.argv) > 4 else True
d = tempfile.mkdtemp(prefix="ick_rate_")
t0 = time.time()
synth.write_dataset(d, "rate", "geo", n_train=n_train, n_val=64, n_test=4, L=20, K=20, V=10000, F=0)
feat = os.path.join(d, "TRAIN_FEATURES_rate.npy")
a = np.load(feat, mmap_mode="r")
if half and a.dtype != np.float16:
    np.save(feat, np.asarray(a, dtype=np.float16))
print("dataset: %d samples, features %s, written in %.1f s" % (n_train, "float16" if half else str(a.dtype), time.time() - t0))
cfg = tr.Config(variant="geo", data_dir=d, data_name="rate", epochs=2, batch_size=64, workers=workers, print_freq=10 ** 9,
                fused=True, out_dir=d, prefetch=prefetch, half_features=half, loader_threads=threads)
t0 = time.time()
tr.main(cfg)
print("prefetch=%s half=%s loader threads=%d (worker processes=%d): %.1f optimizer steps/s in the second epoch (%.2f ms per step of 1280 decode "
      "positions = %.0f decode-steps/s); whole run %.1f s"
      % (prefetch, half, threads, workers, tr.STATS.get("last_epoch_steps_per_s", float("nan")),
         1e3 / tr.STATS.get("last_epoch_steps_per_s", float("nan")), 1280 * tr.STATS.get("last_epoch_steps_per_s", float("nan")),
         time.time() - t0))
