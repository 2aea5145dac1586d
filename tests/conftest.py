import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


GEMM_SPLIT_IDS = {0: "exact_fp32_mfma", 1: "split_bf16_fwd_layouts", 2: "split_bf16_all_large_tiles"}


@pytest.fixture(params=[0, 1, 2], ids=lambda m: GEMM_SPLIT_IDS[m])
def gemm_split(request):
    """Runs the test once per product mode of the large GEMM tiles (csrc/gemm.hip; ick_set_gemm_split): 0 = exact fp32
    MFMA, 1 = six bf16 partial products of the exact 3-way bf16 split where the B operand is k-contiguous, 2 = on every
    large tile.  The reference-pinned suites take this fixture: whichever mode is the library's
    default, all three are held to the same goldens and tolerances."""
    import ick_amd.ops as ops
    before = ops.gemm_split_mode()
    ops.set_gemm_split(request.param)
    yield request.param
    ops.set_gemm_split(before)
