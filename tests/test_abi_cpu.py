"""The C-ABI boundary without a GPU: libganslate_hip.so loads, exports every function include/ganslate_hip.h
declares (and the ctypes prototypes cover exactly that set), and the descriptor structs have the header's layout."""
import ctypes
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
HEADER = (ROOT / "include" / "ganslate_hip.h").read_text()


def declared_functions():
    code = re.sub(r"/\*.*?\*/", "", HEADER, flags=re.S)
    return sorted(set(re.findall(r"\b(gs_[a-z0-9_]+)\s*\(", code)))


def test_library_exports_every_declared_symbol():
    from ganslate_amd.hip import lib as L
    if not L.library_path().is_file():
        import __graft_entry__
        __graft_entry__.build()
    lib = L.load()
    names = declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/ganslate_hip.h but not exported"
    assert sorted(L.EXPORTS) == names, "ctypes prototypes and header declarations differ"


def test_descriptor_layouts_match_header():
    from ganslate_amd.hip import lib as L
    # gs_gconv_desc: 26 int32 + float + 3 int32 + 3 * GS_MAX_TAPS int8
    assert ctypes.sizeof(L.GConvDesc) == 26 * 4 + 4 + 3 * 4 + 3 * L.GS_MAX_TAPS
    assert ctypes.sizeof(L.PNormDesc) == 8 + 18 * 4
    assert ctypes.sizeof(L.WGradDesc) == 17 * 4 + 3 * L.GS_MAX_TAPS
    assert int(re.search(r"#define GS_MAX_TAPS (\d+)", HEADER).group(1)) == L.GS_MAX_TAPS
    for name, val in (("GS_BORDER_REFLECT", L.BORDER["reflect"]), ("GS_BORDER_REPLICATE", L.BORDER["replicate"]),
                      ("GS_ACT_RELU", L.ACT["relu"]), ("GS_ACT_LRELU", L.ACT["lrelu"]), ("GS_ACT_TANH", L.ACT["tanh"])):
        assert int(re.search(name + r" = (\d+)", HEADER).group(1)) == val


def test_no_cpu_fallback_without_gpu():
    """the product path must fail loudly when there is no MI355X (no silent CPU path)"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from ganslate_amd.hip.ops import HipOps
    with pytest.raises(RuntimeError, match="no CPU path"):
        HipOps()
