"""Per-entry-point timing of the C ABI with HIP events (bench.py's `roofline.by_kernel`, tools/).

While `lib.PROFILE` is a list, every ick_* call that takes a stream is bracketed by two events recorded on torch's
current stream -- the stream the kernels are launched on -- and classified by `classify()` into a kernel class
with its ALGORITHMIC work (FLOP for matrix-core kernels, bytes for streaming kernels).  Meant for eager passes on
one stream (hipGraph replays do not go through the Python wrappers); `summarise()` turns the records into rows
{name, launches_per_step, avg_us, us_per_step, work_per_launch, unit, achieved, peak, frac, bound}."""
import ctypes as C

import torch

from . import lib as L

PEAK_FP32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, 64 FLOP/clk/SIMD
PEAK_HBM_GBS = 8000.0             # HBM3E spec
# Kernels that form their fp32 products as six bf16 partial products (csrc/gemm.hip SPL, csrc/gemm_ps.hip) run on the bf16
# matrix pipe: its dense peak (~2.5 PFLOP/s, MI355X_MICROARCH.md) divided by the six MFMAs per product block is the
# fp32-equivalent ceiling of such a kernel -- a fraction against the 157.3 TF of the exact fp32 MFMA could exceed 1.
PEAK_BF16_MFMA_TFLOPS = 2500.0
PEAK_SPLIT_FP32_EQ_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 6.0
# What bounds a row-chain launch is not the chip's matrix peak: every 8-row workgroup streams the WHOLE weight set of its
# stages from its XCD's L2 through its own CU.  A CU's vector L1 fills at 64 B/clk (134 GB/s at the ~2.1 GHz the chip holds
# in these kernels), and the 4x4x1 MFMA consumes a 4 KiB weight chunk per wave in 256 cycles at 8 rows per workgroup, i.e.
# the same 64 B/clk per CU: bytes per workgroup / that rate is the floor of a launch however many CUs run one.  (Measured:
# a latency-limited probe with 72 KiB in flight takes in 66-73 GB/s per CU -- MI355X_MICROARCH.md "Indexed rows: gather
# into LDS", tools/probes/probe_chain.hip; a 300 x 300 stage of the chain kernel itself ~109 GB/s, profiles/r02_m_chain_stamps.txt.)
PER_CU_L2_STREAM_GBS = 134.0


def _struct(arg):
    return arg._obj if hasattr(arg, "_obj") else arg


def _gemm_label(a):
    lay = ("A k-major" if (a.a_rs == 1 and a.a_ks != 1) else "A row-major") + ", " + \
          ("B k-major" if (a.b_rs == 1 and a.b_ks != 1) else "B row-major")
    return "%dx%dx%d (%s%s)" % (a.M, a.N, a.K, lay, ", split-K %d" % a.split_k if a.split_k > 1 else "")


TRACE_KEYS = {}      # class label -> {"<kernel family>:<threads of the grid>"}: how a profiler trace finds the class again


def _split_plan(a, label=None):
    """Does ick_gemm form this problem's products on the bf16 pipe (plan query; nothing is launched)?  With `label` the
    launch's (kernel family, grid) is noted under that class, so that tables made from rocprofv3 traces of the same command
    (tools/in_step_table.py, tools/traffic_table.py) can tell the shared GEMM instantiations apart by their grids."""
    info = L.GemmPlanInfo()
    if L.load_raw().ick_gemm_plan(C.byref(a), C.byref(info)) != 0:
        return False, False
    if label is not None:
        fam = "gemm_ps_kernel" if info.presplit else "gemm_kernel"
        TRACE_KEYS.setdefault(label, set()).add("%s:%d" % (fam, info.tiles_m * info.tiles_n * info.split_k * info.waves * 64))
    return bool(info.split_bf16), bool(info.presplit)


def classify(name, args):
    """-> (class label, work per launch, 'flop' | 'byte' | 'flop_split' | None).  'flop_split': fp32-equivalent FLOP of a
    kernel that runs on the bf16 matrix pipe (priced against PEAK_SPLIT_FP32_EQ_TFLOPS)."""
    if name == "ick_gemm":
        a = _struct(args[0])
        fl = 2.0 * a.M * a.N * a.K
        spl, ps = _split_plan(a)
        if fl >= 2e9:
            tag = " [split-bf16 products, B pre-split]" if ps else (" [split-bf16 products]" if spl else "")
            label = "GEMM " + _gemm_label(a) + tag
            _split_plan(a, label)
            return label, fl, "flop_split" if spl else "flop"
        _split_plan(a, "chain GEMMs (< 2 GFLOP each: projections, FFN, their data gradients)")
        return "chain GEMMs (< 2 GFLOP each: projections, FFN, their data gradients)", fl, "flop"
    if name == "ick_gemm_grouped":
        arr, n = args[0], args[1]
        fl = fl_split = 0.0
        for i in range(n):
            a = arr[i]
            if not (a.flags & L.GEMM_COLSUM_ONLY):
                f = 2.0 * a.M * a.N * a.K
                fl += f
                spl, ps = _split_plan(a)
                if ps:                   # launched on its own (csrc/gemm_ps.hip): found in a trace by its grid
                    _split_plan(a, "grouped weight-gradient GEMMs")
                if spl:
                    fl_split += f        # this problem's products run on the bf16 pipe (six MFMAs per product block)
        # a mixed launch is priced against the time its parts would take at their own pipes' peaks ("flop_mixed")
        return "grouped weight-gradient GEMMs", fl, "flop_mixed", fl_split
    if name == "ick_attention":
        a = _struct(args[0])
        return "attention forward (T=%d)" % a.T if a.T > 1 else "attention decode step", 4.0 * a.B * a.H * a.T * a.S * a.dh, "flop"
    if name == "ick_attention_bwd":
        a = _struct(args[0])
        return "attention backward", 10.0 * a.B * a.H * a.T * a.S * a.dh, "flop"
    if name == "ick_rowchain_fwd":
        a = _struct(args[0])
        fl = 2.0 * a.M * (a.K1 * a.d + (a.d * a.N2 if a.w2p else 0))
        return ("row chain forward (out-projection + add & norm + next Linear)", fl, "flop", 0.0,
                4.0 * (a.K1 * a.d + (a.d * a.N2 if a.w2p else 0)))
    if name == "ick_rowchain_bwd":
        a = _struct(args[0])
        fl = 2.0 * a.M * ((a.K0 * a.d if a.g0 else 0) + (2 * a.d * a.N1 if a.w1p else 0) + a.d * a.d)
        return ("row chain backward (Linear' + norm' [+ FFN' + norm'] + out-projection')", fl, "flop", 0.0,
                4.0 * ((a.K0 * a.d if a.g0 else 0) + (2 * a.d * a.N1 if a.w1p else 0) + a.d * a.d))
    if name == "ick_pack_weights":
        arr, n = args[0], args[1]
        return "packed weight copies", 8.0 * sum(arr[i].N * arr[i].K for i in range(n)), "byte"
    if name == "ick_presplit_weights":
        arr, n = args[0], args[1]
        # 4 bytes read, three bf16 planes (6 bytes) written per element
        return "presplit weights", 10.0 * sum(arr[i].N * arr[i].K for i in range(n)), "byte"
    if name == "ick_add_layernorm":
        rows, d = args[5], args[6]
        return "residual + LayerNorm", 4.0 * rows * d * 3, "byte"
    if name == "ick_layernorm_bwd":
        rows, d = args[9], args[10]
        return "LayerNorm backward", 4.0 * rows * d * 5, "byte"
    if name == "ick_packed_ce":
        B, Lc, Vx = args[4], args[5], args[6]
        return "packed cross entropy (+ gradient)", 4.0 * B * Lc * Vx * (2 if args[11] else 1), "byte"
    if name == "ick_adam_clamp":
        return "clamp + Adam", 4.0 * args[4] * 7, "byte"
    if name == "ick_adam_clamp_derive":
        # the seven streams of the update + every image of the updated weights (packed, transposed, bf16 planes)
        return "clamp + Adam + re-laid-out weight copies", float(getattr(L, "ADAM_DERIVE_BYTES", 0)), "byte"
    if name == "ick_decode_layers":
        c = _struct(args[0])
        kv = 4.0 * c.R * c.layers * 2 * c.H * c.S * 32
        w = 4.0 * (c.layers * (4 * c.d * c.d + 2 * c.d * c.d + 2 * c.d * c.FF) + c.V * c.d)
        return "fused decode step (3 kernels / layer + head + vocabulary)", kv + w, "byte"
    if name == "ick_decode_layers_part":
        # the same step in two launches (selection of the previous token + layer 0's self-attention block | the rest):
        # one class, the step's algorithmic bytes shared out by what each part streams
        c = _struct(args[0])
        kv = 4.0 * c.R * c.layers * 2 * c.H * c.S * 32
        w = 4.0 * (c.layers * (4 * c.d * c.d + 2 * c.d * c.d + 2 * c.d * c.FF) + c.V * c.d)
        first = 4.0 * 4 * c.d * c.d
        return ("fused decode step (3 kernels / layer + head + vocabulary)", first if args[2] == 1 else kv + w - first, "byte")
    if name == "ick_decode_select_greedy":
        return "greedy selection + next-token embedding", None, None
    return name[4:].replace("_", " "), None, None


class _Profiled:
    def __init__(self, lib):
        self._lib = lib

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        argtypes = L.SIGNATURES.get(name)
        if L.PROFILE is None or not argtypes or argtypes[-1] is not L.vp:      # only entries that take a stream
            return fn

        def timed(*args):
            label, work, unit, *rest = classify(name, args)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = fn(*args)
            e1.record()
            L.PROFILE.append((label, work, unit, e0, e1, rest[0] if rest else 0.0, rest[1] if len(rest) > 1 else 0.0))
            return rc
        return timed


def start():
    L.PROFILE = []
    L._proxy = _Profiled(L.load_raw())


def stop():
    rec, L.PROFILE = L.PROFILE, None
    L._proxy = None
    return rec


def summarise(records, steps):
    torch.cuda.synchronize()
    rows = {}
    for label, work, unit, e0, e1, work_split, cu_bytes in records:
        r = rows.setdefault(label, {"name": label, "launches": 0, "us": 0.0, "work": 0.0, "unit": unit, "work_split": 0.0,
                                    "cu_bytes": 0.0})
        r["launches"] += 1
        r["cu_bytes"] += cu_bytes
        r["us"] += e0.elapsed_time(e1) * 1e3
        if work is not None:
            r["work"] += work
            r["work_split"] += work_split
    out = []
    for r in rows.values():
        n = r["launches"]
        row = {"name": r["name"], "launches_per_step": n / steps, "avg_us": r["us"] / n, "us_per_step": r["us"] / steps}
        if r["unit"] in ("flop", "flop_split", "flop_mixed") and r["us"] > 0 and r["work"] > 0:
            ach = r["work"] / (r["us"] * 1e-6) / 1e12
            if r["unit"] == "flop_mixed":
                # problems of both pipes in one launch: the peak is the rate at which the launch's parts would finish
                # at their own pipes' peaks, one after the other
                ws = r["work_split"]
                peak = r["work"] / (ws / PEAK_SPLIT_FP32_EQ_TFLOPS + (r["work"] - ws) / PEAK_FP32_MFMA_TFLOPS)
                pipe = "%.0f %% of the FLOP on the bf16 MFMA x 6 partial products, the rest on the fp32 MFMA" % (100 * ws / r["work"])
            else:
                peak = PEAK_SPLIT_FP32_EQ_TFLOPS if r["unit"] == "flop_split" else PEAK_FP32_MFMA_TFLOPS
                pipe = "bf16 MFMA x 6 partial products (fp32-equivalent FLOP)" if r["unit"] == "flop_split" else "fp32 MFMA"
            row.update(work_per_launch=r["work"] / n, unit="TFLOP/s", achieved=ach, peak=peak, frac=ach / peak, bound="mfma",
                       pipe=pipe, frac_of_fp32_mfma_peak=ach / PEAK_FP32_MFMA_TFLOPS)
            if r["cu_bytes"] > 0:
                # the launch's real bound: the weight bytes EVERY workgroup pulls through its own CU from L2
                floor_us = r["cu_bytes"] / n / (PER_CU_L2_STREAM_GBS * 1e9) * 1e6
                row.update(bound_detail="l2_stream_per_cu", cu_stream_bytes_per_workgroup=r["cu_bytes"] / n,
                           cu_stream_gbs=PER_CU_L2_STREAM_GBS, cu_stream_floor_us=floor_us,
                           frac_of_cu_stream_floor=floor_us / (r["us"] / n))
        elif r["unit"] == "byte" and r["us"] > 0:
            ach = r["work"] / (r["us"] * 1e-6) / 1e9
            row.update(work_per_launch=r["work"] / n, unit="GB/s", achieved=ach, peak=PEAK_HBM_GBS,
                       frac=ach / PEAK_HBM_GBS, bound="hbm")
        out.append(row)
    for row in out:
        if row["name"] in TRACE_KEYS:
            row["trace_keys"] = sorted(TRACE_KEYS[row["name"]])
    out.sort(key=lambda x: -x["us_per_step"])
    return out
