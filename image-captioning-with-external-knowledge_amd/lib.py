"""ctypes binding of libick_amd.so (C ABI: include/ick_amd.h).

The library is built in-tree by build.py; a missing library is a hard error -- there is no
CPU or PyTorch fallback anywhere in the product path.
"""
import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libick_amd.so")
if os.environ.get("ICK_LIB_PATH"):      # experiment hook: a library built with other compile-time switches
    LIB_PATH = os.environ["ICK_LIB_PATH"]

ICK_GEO, ICK_KNOWLEDGE, ICK_NEWS = 0, 1, 2
VARIANT_ID = {"geo": ICK_GEO, "knowledge": ICK_KNOWLEDGE, "news": ICK_NEWS}
GEMM_RELU, GEMM_ACCUM, GEMM_ATOMIC, GEMM_COLSUM_ONLY = 1, 2, 4, 16

vp = C.c_void_p
i32 = C.c_int32
i64 = C.c_int64
f32 = C.c_float
u32 = C.c_uint32


class GemmArgs(C.Structure):
    _fields_ = [
        ("A", vp), ("B", vp), ("C", vp), ("bias", vp),
        ("M", i32), ("N", i32), ("K", i32),
        ("a_rs", i64), ("a_ks", i64), ("a_grp", i32), ("a_gs", i64), ("a_gmap", vp),
        ("b_rs", i64), ("b_ks", i64),
        ("c_rs", i64), ("c_grp", i32), ("c_gs", i64), ("c_gmap", vp),
        ("flags", i32), ("split_k", i32), ("alpha", f32),
        ("hs_dh", i32), ("hs_dhp", i32), ("hs_H", i32), ("hs_S", i32), ("hs_s0", i32),
        ("drop_p", f32), ("drop_seed", u32), ("drop_site", u32), ("drop_epoch", vp),
        ("a_extent", i64), ("b_extent", i64), ("colsum_a", vp),
        ("gate", vp), ("gate_rs", i64), ("gate_scale", f32),
        ("b_ps", vp),
    ]


class RowChainArgs(C.Structure):
    _fields_ = [
        ("A", vp), ("a_rs", i64), ("a_grp", i32), ("a_gs", i64),
        ("M", i32), ("K1", i32), ("d", i32),
        ("w1p", vp), ("b1", vp), ("res", vp), ("res_rs", i64),
        ("gamma", vp), ("beta", vp), ("eps", f32),
        ("drop1_p", f32), ("drop_seed", u32), ("drop1_site", u32), ("drop_epoch", vp),
        ("o", vp), ("o_rs", i64),
        ("x", vp), ("x_rs", i64), ("x_grp", i32), ("x_gs", i64),
        ("mean", vp), ("rstd", vp),
        ("w2p", vp), ("b2", vp), ("N2", i32), ("flags", i32),
        ("drop2_p", f32), ("drop2_site", u32),
        ("y2", vp), ("y2_rs", i64), ("y2_grp", i32), ("y2_gs", i64),
        ("hs_dh", i32), ("hs_dhp", i32), ("hs_H", i32), ("hs_S", i32), ("hs_s0", i32),
    ]


class RowChainBwdArgs(C.Structure):
    _fields_ = [
        ("M", i32), ("d", i32), ("drop_seed", u32), ("drop_epoch", vp),
        ("g0", vp), ("g0_rs", i64), ("K0", i32), ("w0p", vp), ("g0_grp", i32), ("g0_gs", i64),
        ("dzin", vp), ("dzin_rs", i64),
        ("o1", vp), ("res1", vp), ("mean1", vp), ("rstd1", vp), ("gamma1", vp),
        ("drop1_p", f32), ("drop1_site", u32),
        ("do1", vp), ("part1", vp),
        ("w1p", vp), ("N1", i32), ("act", vp), ("gate_scale", f32), ("t_out", vp),
        ("w2p", vp),
        ("o2", vp), ("res2", vp), ("mean2", vp), ("rstd2", vp), ("gamma2", vp),
        ("drop2_p", f32), ("drop2_site", u32),
        ("do2", vp), ("part2", vp),
        ("w3p", vp), ("out3", vp), ("dz_out", vp), ("flags", i32),
    ]


class PackItem(C.Structure):
    _fields_ = [("src", vp), ("dst", vp), ("N", i32), ("K", i32), ("src_rs", i64), ("src_cs", i64), ("dst_rs", i64)]


class GemmPlanInfo(C.Structure):
    _fields_ = [(n, i32) for n in ("tile_m", "tile_n", "waves", "tiles_m", "tiles_n", "split_k", "a_kmajor", "b_kmajor",
                                   "vec", "split_bf16", "presplit")]


class PresplitItem(C.Structure):
    _fields_ = [("src", vp), ("dst", vp), ("N", i32), ("K", i32), ("src_rs", i64), ("src_cs", i64)]


class AdamItem(C.Structure):      # ick_adam_item: a 2-D block of the bucket + the images of it the optimizer keeps current
    _fields_ = [("off", i64), ("rows", i32), ("K", i32), ("drow0", i32), ("Nd", i32), ("pack", vp), ("pack_t", vp),
                ("copy", vp), ("ps", vp), ("ps_t", vp), ("tr", vp), ("copy_ld", i64), ("tr_ld", i64)]


class AdamBlock(C.Structure):     # ick_adam_block: the work of one workgroup of ick_adam_clamp_derive
    _fields_ = [("item", i32), ("tn", i32), ("tk", i32), ("cnt4", i32), ("off4", i64), ("copy", vp)]


class AttnArgs(C.Structure):
    _fields_ = [
        ("Q", vp), ("K", vp), ("V", vp), ("O", vp), ("lse", vp),
        ("B", i32), ("H", i32), ("T", i32), ("S", i32), ("dh", i32),
        ("q_bs", i64), ("q_hs", i64), ("q_ts", i64), ("k_bs", i64), ("k_hs", i64), ("k_ss", i64),
        ("v_bs", i64), ("v_hs", i64), ("v_ss", i64), ("o_bs", i64), ("o_ts", i64),
        ("scale", f32), ("causal", i32), ("q_pos0", i32), ("kv_len", vp),
        ("drop_p", f32), ("drop_seed", u32), ("drop_site", u32), ("drop_epoch", vp),
    ]


class AttnBwdArgs(C.Structure):
    _fields_ = [
        ("Q", vp), ("K", vp), ("V", vp), ("O", vp), ("dO", vp), ("lse", vp), ("dQ", vp), ("dK", vp), ("dV", vp),
        ("B", i32), ("H", i32), ("T", i32), ("S", i32), ("dh", i32),
        ("q_bs", i64), ("q_hs", i64), ("q_ts", i64), ("k_bs", i64), ("k_hs", i64), ("k_ss", i64),
        ("v_bs", i64), ("v_hs", i64), ("v_ss", i64), ("o_bs", i64), ("o_ts", i64),
        ("dq_bs", i64), ("dq_ts", i64), ("dk_bs", i64), ("dk_ss", i64), ("dv_bs", i64), ("dv_ss", i64),
        ("scale", f32), ("causal", i32), ("q_pos0", i32),
        ("drop_p", f32), ("drop_seed", u32), ("drop_site", u32), ("drop_epoch", vp),
    ]


class DecodeLayer(C.Structure):
    _fields_ = [(n, vp) for n in ("sa_in_w", "sa_in_b", "sa_out_wt", "sa_out_b", "n1_g", "n1_b",
                                  "ca_in_w", "ca_in_b", "ca_out_wt", "ca_out_b", "n2_g", "n2_b",
                                  "w1", "b1", "w2t", "b2", "n3_g", "n3_b", "self_k", "self_v", "cross_k", "cross_v")]


MAX_LAYERS = 8


class DecodeCtx(C.Structure):
    _fields_ = ([(n, i32) for n in ("R", "rows_per_sample", "d", "H", "FF", "layers", "S", "max_len", "V", "K", "F",
                                    "end_token", "pad_token")] +
                [("ln_eps", f32), ("emb_scale", f32), ("kv_bs", i64), ("scores_ld", i64),
                 ("layer", DecodeLayer * MAX_LAYERS)] +
                [(n, vp) for n in ("anc", "wv", "bv", "we", "be", "wf", "bf", "ee", "fe", "gate", "eib", "word_emb", "pe",
                                   "x0", "xa", "xb", "xc", "p1", "p2", "p3", "hfin", "hv", "ptr", "cand", "scores",
                                   "output", "hist", "finished", "n_done", "next_token", "next_mask", "cap_buf", "sel_state")])


class BeamState(C.Structure):
    _fields_ = [(n, vp) for n in ("cum", "fin", "seq_in", "seq_out", "anc_in", "anc_out", "cap_in", "cap_out", "rec")] + \
               [("start_token", i32)]


# name -> argtypes; every entry returns int (0 ok, <0 ICK_E*, >0 hipError_t)
SIGNATURES = {
    "ick_version": [],
    "ick_set_deterministic": [C.c_int],
    "ick_get_deterministic": [],
    "ick_device_info": [C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_char_p, C.c_int],
    "ick_gemm": [C.POINTER(GemmArgs), vp],
    "ick_gemm_plan": [C.POINTER(GemmArgs), C.POINTER(GemmPlanInfo)],
    "ick_set_gemm_split": [i32],
    "ick_get_gemm_split": [],
    "ick_presplit_bytes": [i32, i32, C.POINTER(i64)],
    "ick_presplit_weights": [C.POINTER(PresplitItem), i32, vp],
    "ick_rowchain_supported": [i32, i32, i32],
    "ick_rowchain_fwd": [C.POINTER(RowChainArgs), vp],
    "ick_rowchain_bwd_supported": [i32, i32, i32],
    "ick_rowchain_bwd": [C.POINTER(RowChainBwdArgs), vp],
    "ick_pack_weights": [C.POINTER(PackItem), i32, vp],
    "ick_packed_weight_floats": [i32, i32, C.POINTER(i64)],
    "ick_add_layernorm": [vp, vp, vp, vp, vp, i64, i32, f32, i64, i64, i64, vp, vp, f32, u32, u32, vp, vp],
    "ick_attention": [C.POINTER(AttnArgs), vp],
    "ick_entity_encode": [i32, vp, i32, vp, vp, i32, vp, i32, vp, i32, i32, i32, i32, vp],
    "ick_fact_encode": [vp, vp, vp, i32, vp, i32, i32, i32, i32, vp],
    "ick_caption_embed": [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, f32, i32, f32, u32, u32,
                          vp, vp],
    "ick_context_indicators": [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp],
    "ick_pointer_scores": [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i64, i32, vp, vp],
    "ick_mul": [vp, vp, vp, i64, vp],
    "ick_dropout_mask": [vp, i64, i32, f32, u32, u32, vp],
    "ick_top2": [vp, i64, i32, i32, vp, vp, vp],
    "ick_greedy_update": [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp],
    "ick_greedy_select": [vp, i64, i32, i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp],
    "ick_packed_ce": [vp, i64, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp],
    "ick_decode_supported": [i32, i32, i32, i32, i32],
    "ick_decode_layers": [C.POINTER(DecodeCtx), i32, vp],
    "ick_decode_layers_part": [C.POINTER(DecodeCtx), i32, i32, vp],
    "ick_decode_init": [C.POINTER(DecodeCtx), i32, i32, vp],
    "ick_decode_select_greedy": [C.POINTER(DecodeCtx), i32, vp],
    "ick_decode_select_beam": [C.POINTER(DecodeCtx), C.POINTER(BeamState), i32, vp],
    "ick_decode_beam_supported": [i32, i32],
    "ick_attention_bwd": [C.POINTER(AttnBwdArgs), vp],
    "ick_layernorm_bwd": [vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, vp, f32, u32, u32, vp, vp, vp],
    "ick_layernorm_bwd_rows_per_block": [],
    "ick_gemm_grouped": [vp, i32, vp],
    "ick_attention_bwd_overwrites": [i32, i32, i32],
    "ick_relu_bwd": [vp, vp, vp, i64, f32, vp],
    "ick_colsum": [vp, i64, i32, i64, vp, vp],
    "ick_caption_embed_bwd": [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, f32, f32, u32, u32, vp, vp],
    "ick_pointer_scores_bwd": [vp, i64, i32, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp],
    "ick_entity_encode_bwd": [i32, vp, vp, i32, vp, vp, i32, vp, i32, vp, i32, i32, i32, vp],
    "ick_fact_encode_bwd": [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp],
    "ick_context_gate_bwd": [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp],
    "ick_adam_clamp": [vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, f32, i32, vp, vp, vp],
    "ick_adam_clamp_derive": [vp, vp, vp, vp, vp, vp, i32, f32, f32, f32, f32, f32, f32, i32, vp, vp, vp],
    "ick_counter_add": [vp, u32, vp],
    "ick_counter_add_if": [vp, u32, vp, vp],
    "ick_timestamp": [vp, vp],
    "ick_copy_batch": [vp, vp, vp, i32, vp],
    "ick_scale_by_ratio": [vp, i64, vp, vp, vp],
}

_lib = None
PROFILE = None     # profiling.start() / stop(): list of (class, work, unit, event, event) while enabled
_proxy = None


class IckError(RuntimeError):
    pass


def load():
    """The shared library (loaded on first use); raises if it has not been built.  While profiling.start() is in
    effect a proxy that brackets every launch with HIP events is returned instead."""
    return _proxy if _proxy is not None else load_raw()


def load_raw():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise IckError(
                "libick_amd.so is missing (%s): build it with `python -c 'import __graft_entry__ as g; g.build()'`"
                " -- there is no fallback path" % LIB_PATH)
        # torch first: PyTorch-ROCm ships its own libamdhip64; if ours were loaded before it, the process would hold
        # two HIP runtimes and every launch from this library would fail with hipErrorNoDevice on torch's streams
        import torch  # noqa: F401
        lib = C.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the symbol is not exported
            fn.argtypes = argtypes
            fn.restype = C.c_int
        _lib = lib
    return _lib


_ERRS = {-1: "ICK_EINVAL (bad shape / null pointer / unsupported size)", -2: "ICK_EALIGN", -3: "ICK_EWORKSPACE"}


def check(rc, what):
    if rc != 0:
        raise IckError("%s failed: %s" % (what, _ERRS.get(rc, "hipError_t %d" % rc)))
