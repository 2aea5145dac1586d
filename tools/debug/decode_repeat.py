"""Is a decode kernel slower in the step's rotation than on its own?  Diagnostic build (-DICK_DECODE_STAMPS): times, with
HIP events, `reps` back-to-back passes over subsets of the step's kernels at cfg5 (R = 32): one class alone (the same
3 launches over and over: warm instruction cache, warm L2) against the whole rotation."""
import ctypes, math, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ick_amd.build as b
import ick_amd.lib as L
dbg = os.path.join(ROOT, "gpurun_out", "libick_amd_dbg.so")
subprocess.check_call([b.HIPCC] + b.FLAGS + ["-DICK_DECODE_STAMPS", "-shared", "-o", dbg] + b.sources())
L.LIB_PATH = dbg
import torch
import ick_amd, ick_amd.synth as synth, ick_amd.ops as ops
variant, B, ML, K, V = "geo", int(os.environ.get("R", 32)), 20, 20, 10000
m = ick_amd.load_models(variant)
dec = m.DecoderTransformer(synth.make_word_map(V), 300, 512, 512, 10, 3)
dec.load_state_dict(synth.make_params(variant, V, 0), strict=False)
dec = dec.cuda().eval()
ents = synth.make_entities(variant, B, K, V, 1)
enc = synth.make_enc_out(B, 1).cuda()
enc_out, ents, facts = dec._prepare_inputs(enc, ents, None)
enc_tok = dec._token_major(enc_out).contiguous()
ee, fe, kv, _, side = dec._encode_context(enc_tok, ents.contiguous(), None, None)
side.join()
c, t = dec._decode_ctx(kv, ee, fe, 1, ML, kv.shape[3])
t["x0"].normal_()
lib = L.load_raw()
fn = lib.ick_debug_decode_subset
fn.argtypes = [ctypes.POINTER(L.DecodeCtx), ctypes.c_int32, ctypes.c_uint, ctypes.c_int32, ctypes.c_void_p]
fn.restype = ctypes.c_int
stream = torch.cuda.current_stream().cuda_stream
def timed(which, reps, pos=10):
    assert fn(ctypes.byref(c), pos, which, 3, stream) == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    assert fn(ctypes.byref(c), pos, which, reps, stream) == 0
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
reps = 200
names = [(1, "self x3"), (2, "cross x3"), (4, "ffn x3"), (8, "head"), (16, "vocab"), (7, "self+cross+ffn x3"), (31, "whole step")]
res = {}
for w, n in names:
    res[w] = timed(w, reps)
    print("%-20s %8.2f us per pass" % (n, res[w]))
print("sum of the classes alone: %.2f us; whole step: %.2f us" % (res[1] + res[2] + res[4] + res[8] + res[16], res[31]))
