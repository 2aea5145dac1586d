"""Maps a rocprofv3 kernel name (+ grid size) of the cfg2 train / forward / greedy benches onto the kernel-class names
bench.py's roofline.by_kernel uses, so that tables made from rocprofv3 runs (in-step durations, PMC traffic) can be joined
with the bench line.  The large GEMMs of the split-product default share one kernel instantiation and are told apart by
their grid (cfg2: 12 544 image rows, 1 280 caption rows, V = 10 000)."""
import re

PS = " [split-bf16 products, B pre-split]"
CLASSES = {
    "wgrad": "grouped weight-gradient GEMMs",
    "chain_fwd": "row chain forward (out-projection + add & norm + next Linear)",
    "chain_bwd": "row chain backward (Linear' + norm' [+ FFN' + norm'] + out-projection')",
    "attn_fwd": "attention forward (T=20)",
    "attn_bwd": "attention backward",
    "small": "chain GEMMs (< 2 GFLOP each: projections, FFN, their data gradients)",
    "adam": "clamp + Adam",
    "ce": "packed cross entropy (+ gradient)",
    "pack": "packed weight copies",
    "presplit": "presplit weights",
    "conv1_ps": "GEMM 12544x300x2048 (A k-major, B row-major)" + PS,
    "kv_ps": "GEMM 12544x1800x300 (A row-major, B row-major)" + PS,
    "vocab_ps": "GEMM 1280x10000x300 (A row-major, B row-major)" + PS,
    "conv1": "GEMM 12544x300x2048 (A k-major, B row-major)",
    "vocab_dgrad": "GEMM 1280x300x10000 (A row-major, B k-major, split-K 9)",
    "vocab_dgrad_ps": "GEMM 1280x300x10000 (A row-major, B k-major, split-K 12)" + PS,
    "decode": "fused decode step (3 kernels / layer + head + vocabulary)",
}


def _targs(name, kernel):
    m = re.search(kernel + r"<([^>]*)>", name)
    return [a.strip() for a in m.group(1).split(",")] if m else None


def classify(name, grid_x=None):
    """-> key of CLASSES or None.  grid_x: total threads of the launch (x * y * z)."""
    if "gemm_group_kernel" in name:
        return "wgrad"
    if "rowchain_fwd_kernel" in name:
        return "chain_fwd"
    if "rowchain_bwd_kernel" in name:
        return "chain_bwd"
    if "attn_fwd_mfma_kernel" in name:
        return "attn_fwd"
    if "attn_bwd_mfma_kernel" in name:
        return "attn_bwd"
    if "adam_clamp" in name:
        return "adam"
    if "packed_ce_" in name:
        return "ce"
    if "pack_weights_kernel" in name:
        return "pack"
    if "presplit_kernel" in name:
        return "presplit"
    if re.search(r"dec_(self|cross|ffn|head|vocab|headvocab)_kernel", name):
        return "decode"
    t = _targs(name, "gemm_ps_kernel")
    if t:
        g = int(grid_x) if grid_x else 0              # TOTAL threads of the grid (x * y * z)
        if t[4] == "true":                            # A k-major: Encoder.conv1 (392 tiles of 128 x 80) or a weight gradient
            return "conv1_ps" if g in (200704, 100352) else "wgrad"
        # A row-major: K/V projection (98 x 15 tiles of 128 x 128), vocabulary (10 x 79), its data gradient (40 tiles of
        # 128 x 80 x 12 K slices)
        return {752640: "kv_ps", 404480: "vocab_ps", 245760: "vocab_dgrad_ps"}.get(g)
    t = _targs(name, "gemm_kernel")
    if t:
        if t[4] == "true" and t[5] == "true":
            return "wgrad"                       # stand-alone weight gradients (vocabulary, cross K/V)
        if t[2] == "1":
            return "small"
        if t[0] == "4" and t[4] == "true":
            return "conv1"
        if t[4] == "false" and t[5] == "true":
            return "vocab_dgrad"
    return None
