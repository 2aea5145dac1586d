"""Shared helpers for the parity tests (CPU side)."""
import os

import numpy as np
import torch

import ick_amd.synth as synth
from oracle import restatement as R

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


def param_checksum(P):
    return float(sum(v.double().abs().sum() for v in P.values()))


def case_from_golden(g):
    """Rebuild params / inputs of a forward fixture from its seed."""
    variant = str(g["variant"])
    B, L, K, V, Fn, seed = (int(g[k]) for k in ("B", "L", "K", "V", "F", "seed"))
    P = synth.make_params(variant, V, seed)
    assert abs(param_checksum(P) - float(g["param_checksum"][0])) < 1e-6 * float(g["param_checksum"][0]), \
        "synthetic parameter generator drifted from the one the fixture was made with"
    wm = synth.make_word_map(V)
    cfg = R.config_from_word_map(variant, wm)
    batch = synth.make_batch(variant, B, L, K, V, Fn, seed)
    enc_out = synth.make_enc_out(B, seed)
    return cfg, P, wm, batch, enc_out


def t(x):
    return torch.from_numpy(np.asarray(x))
