"""Phase timing of the fused decode kernels (diagnostic): builds a copy of the library with -DICK_DECODE_STAMPS,
runs a cfg5 greedy decode eagerly and prints, for workgroup (0,0) of the last launch of each kernel, the shader-clock
deltas between its phase stamps (100 MHz s_memtime ticks -> 10 ns each)."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ick_amd.build as b
import ick_amd.lib as L
dbg = os.path.join(ROOT, "gpurun_out", "libick_amd_dbg.so")
cmd = [b.HIPCC] + b.FLAGS + ["-DICK_DECODE_STAMPS", "-shared", "-o", dbg] + b.sources()
subprocess.check_call(cmd)
L.LIB_PATH = dbg
import torch
import ick_amd, ick_amd.synth as synth
variant, B, L_, K, V = "geo", 32, 20, 20, 10000
m = ick_amd.load_models(variant)
dec = m.DecoderTransformer(synth.make_word_map(V), 300, 512, 512, 10, 3)
dec.load_state_dict(synth.make_params(variant, V, 0), strict=False)
dec = dec.cuda().eval()
dec.use_hip_graphs = False
ents = synth.make_entities(variant, B, K, V, 1)
enc = synth.make_enc_out(B, 1).cuda()
for _ in range(3):
    dec.predict(enc, L_, ents)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 128)()
lib = ctypes.CDLL(dbg)
assert lib.ick_debug_read_stamps(buf) == 0
names = {0: ("self", ["loads issued", "LN on load + barrier", "cache/out loads issued + qkv GEMV + barrier", "attention + barrier", "out-proj"]),
         1: ("cross", ["row loads issued", "q weights issued", "K/V issued", "out weights issued", "n_done arrived", "rows arrived + slice sums",
                       "barrier", "LN finish", "barrier", "q GEMV", "barrier", "attention", "barrier", "combine + barrier", "out-proj"]),
         2: ("ffn", ["loads issued", "LN on load + barrier", "linear1 GEMV + barrier", "linear2"]),
         3: ("vocab", ["weights + rows issued", "MFMA + LDS write", "barrier", "reduce + candidates"])}
for k, (nm, ph) in names.items():
    t = [buf[k * 16 + i] for i in range(len(ph) + 1)]
    print(nm, "total %d cycles:" % (t[-1] - t[0]), ", ".join("%s %d" % (p, t[i + 1] - t[i]) for i, p in enumerate(ph)))
