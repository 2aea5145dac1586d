"""CPU-side checks of the boundary: the C-ABI library builds for gfx950, loads, and exports every
symbol include/ick_amd.h declares (no compute without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built_lib():
    import ick_amd.build as build
    return build.build()


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "ick_amd.h")).read()
    return sorted(set(re.findall(r"^int\s+(ick_[a-z0-9_]+)\s*\(", src, flags=re.M)))


def test_header_declares_entry_points():
    syms = declared_symbols()
    assert "ick_gemm" in syms and "ick_attention" in syms and len(syms) >= 12


def test_library_exports_every_declared_symbol(built_lib):
    lib = ctypes.CDLL(built_lib)
    for s in declared_symbols():
        assert hasattr(lib, s), "libick_amd.so does not export %s" % s


def test_binding_covers_header(built_lib):
    import ick_amd.lib as L
    assert sorted(L.SIGNATURES) == declared_symbols()
    lib = L.load()
    assert lib.ick_version() >= 100


def test_struct_layout_matches_header(built_lib):
    # the binding's ctypes structs must have the size the C compiler gives the header's structs
    import subprocess
    import tempfile
    import ick_amd.lib as L
    src = '#include <stdio.h>\n#include "ick_amd.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu", sizeof(ick_presplit_item), sizeof(ick_gemm_args), sizeof(ick_attn_args), sizeof(ick_attn_bwd_args), sizeof(ick_gemm_plan_info), sizeof(ick_decode_layer), sizeof(ick_decode_ctx), sizeof(ick_beam_state), sizeof(ick_rowchain_args), sizeof(ick_pack_item), sizeof(ick_rowchain_bwd_args)); printf(" %zu %zu", sizeof(ick_adam_item), sizeof(ick_adam_block));}\n'
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "sz.c")
        open(c, "w").write(src)
        exe = os.path.join(td, "sz")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        ps, g, a, ab, gp, dl, dc, bm, rc, ti, rb, ai, abk = (int(x) for x in subprocess.check_output([exe]).split())
    assert ctypes.sizeof(L.PresplitItem) == ps
    assert ctypes.sizeof(L.GemmArgs) == g
    assert ctypes.sizeof(L.AttnArgs) == a
    assert ctypes.sizeof(L.AttnBwdArgs) == ab
    assert ctypes.sizeof(L.GemmPlanInfo) == gp
    assert ctypes.sizeof(L.DecodeLayer) == dl
    assert ctypes.sizeof(L.DecodeCtx) == dc and L.MAX_LAYERS == 8
    assert ctypes.sizeof(L.BeamState) == bm
    assert ctypes.sizeof(L.RowChainArgs) == rc
    assert ctypes.sizeof(L.PackItem) == ti
    assert ctypes.sizeof(L.RowChainBwdArgs) == rb
    assert ctypes.sizeof(L.AdamItem) == ai and ctypes.sizeof(L.AdamBlock) == abk


def test_code_object_targets_gfx950(built_lib):
    data = open(built_lib, "rb").read()
    assert b"gfx950" in data


def test_product_path_does_not_import_oracle():
    pkg = os.path.join(ROOT, "image-captioning-with-external-knowledge_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in re.sub(r"#.*", "", txt), "%s references oracle/" % f
