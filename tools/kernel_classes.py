"""Maps a rocprofv3 kernel name (+ grid size) of a bench.py run onto the kernel-class names of bench.py's
roofline.by_kernel, so that tables made from rocprofv3 runs (in-step durations, PMC traffic) can be joined with the bench
line of the SAME command.  Kernels whose family decides the class are mapped by name; the large GEMMs share two kernel
instantiations and are told apart by their grids, which the bench line itself records per class (`trace_keys`:
"<kernel family>:<threads of the grid>", written by ick_amd/profiling.py from the plan of every launch) -- so the mapping
holds for every workload (cfg2, cfg4, cfg5), not only for the grids of one."""
import json
import re

BY_NAME = [            # (regex on the kernel name, class label)
    (r"gemm_group_kernel", "grouped weight-gradient GEMMs"),
    (r"rowchain_fwd_kernel", "row chain forward (out-projection + add & norm + next Linear)"),
    (r"rowchain_bwd_kernel", "row chain backward (Linear' + norm' [+ FFN' + norm'] + out-projection')"),
    (r"attn_bwd_mfma_kernel|attn_bwd_kernel", "attention backward"),
    (r"adam_derive_kernel", "clamp + Adam + re-laid-out weight copies"),
    (r"adam_clamp", "clamp + Adam"),
    (r"packed_ce_", "packed cross entropy (+ gradient)"),
    (r"pack_weights_kernel", "packed weight copies"),
    (r"presplit_kernel", "presplit weights"),
    (r"dec_(self|cross|ffn|head|vocab|headvocab|beam_partial)_kernel", "fused decode step (3 kernels / layer + head + vocabulary)"),
    (r"dec_select_beam", "decode select beam"),
    (r"dec_init", "decode init"),
    (r"pointer_bwd_", "pointer scores bwd"),
    (r"pointer_scores_kernel", "pointer scores"),
    (r"caption_embed_bwd", "caption embed bwd"),
    (r"caption_embed_kernel", "caption embed"),
    (r"entity_encode_bwd", "entity encode bwd"),
    (r"entity_encode_kernel", "entity encode"),
]


def load_keymap(bench_json):
    """{"family:grid": class label} + {NQT: attention-forward label} from a bench.py line (file with the JSON line last)."""
    line = [ln for ln in open(bench_json).read().splitlines() if ln.startswith("{")][-1]
    rows = json.loads(line)["roofline"]["by_kernel"]
    keys, attn = {}, {}
    for r in rows:
        for k in r.get("trace_keys", []):
            keys.setdefault(k, r["name"])
        m = re.match(r"attention forward \(T=(\d+)\)", r["name"])
        if m:
            attn[(int(m.group(1)) + 15) // 16] = r["name"]
    return {"keys": keys, "attn": attn, "names": {r["name"] for r in rows}}


def _targs(name, kernel):
    m = re.search(kernel + r"<([^>]*)>", name)
    return [a.strip() for a in m.group(1).split(",")] if m else None


def classify(name, grid_threads=None, keymap=None):
    """-> class label or None.  grid_threads: total threads of the launch (x * y * z)."""
    for pat, label in BY_NAME:
        if re.search(pat, name):
            return label
    t = _targs(name, "attn_fwd_mfma_kernel")
    if t:
        if keymap and int(t[0]) in keymap["attn"]:
            return keymap["attn"][int(t[0])]
        return "attention forward (T=20)"
    g = int(grid_threads) if grid_threads else 0
    for fam in ("gemm_ps_kernel", "gemm_kernel"):
        t = _targs(name, fam)
        if t is None:
            continue
        if keymap and "%s:%d" % (fam, g) in keymap["keys"]:
            return keymap["keys"]["%s:%d" % (fam, g)]
        if fam == "gemm_kernel" and t[4] == "true" and t[5] == "true":
            return "grouped weight-gradient GEMMs"          # stand-alone exact weight gradients
        if fam == "gemm_kernel" and t[2] == "1":
            return "chain GEMMs (< 2 GFLOP each: projections, FFN, their data gradients)"
    return None
