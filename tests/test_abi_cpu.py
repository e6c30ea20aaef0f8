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


def test_header_is_valid_c11():
    """the boundary is a C ABI: the header must compile as C, not only as C++ (hipcc accepted a struct with member
    function declarations in round 1)"""
    import subprocess
    r = subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-fsyntax-only", "-x", "c",
                        str(ROOT / "include" / "ganslate_hip.h")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_c_caller_sees_the_ctypes_layout(tmp_path):
    """a C11 program including the header fills a gs_gconv_desc, dlopens the library and calls its host-side planning
    entry points; sizeof / offsetof of every descriptor must equal the ctypes mirrors the Python host side uses"""
    import subprocess
    from ganslate_amd.hip import lib as L
    if not L.library_path().is_file():
        import __graft_entry__
        __graft_entry__.build()
    exe = tmp_path / "abi_probe"
    r = subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-I", str(ROOT / "include"),
                        str(ROOT / "tests" / "abi" / "abi_probe.c"), "-o", str(exe), "-ldl"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe), str(L.library_path())], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    mirrors = {"gs_gconv_desc": L.GConvDesc, "gs_wgrad_desc": L.WGradDesc, "gs_gconv_fuse": L.GConvFuse,
               "gs_norm_ex_desc": L.NormExDesc, "gs_pnorm_desc": L.PNormDesc}
    calls, seen = {}, set()
    for line in r.stdout.splitlines():
        w = line.split()
        if w[0] == "sizeof":
            assert ctypes.sizeof(mirrors[w[1]]) == int(w[2]), line
            seen.add(w[1])
        elif w[0] == "offsetof":
            assert getattr(mirrors[w[1]], w[2]).offset == int(w[3]), line
        elif w[0] == "call":
            calls[w[1]] = int(w[2])
    assert seen == set(mirrors)
    # the same descriptors through ctypes give the same answers as the C caller got
    lib = L.load()
    d = L.GConvDesc()
    d.N = 8; d.Hi = d.Wi = d.Ho = d.Wo = d.Hc = d.Wc = 64; d.Ci = d.Co = d.in_cs = d.out_cs = 256
    d.Di = d.Do = d.Dc = 1; d.so = d.si = 1; d.T = 9; d.Kp = 9 * 256; d.w_rows = 256; d.border = L.BORDER["reflect"]
    for t in range(9):
        d.dh[t], d.dw[t] = t // 3 - 1, t % 3 - 1
    assert calls["stat_slots"] == lib.gs_gconv_stat_slots(ctypes.byref(d)) > 0
    assert calls["tile_m"] == lib.gs_tile_m(ctypes.byref(d)) > 0
    assert calls["splitk_ws_floats"] == lib.gs_gconv_splitk_ws_floats(ctypes.byref(d)) == 0
    assert calls["tail_splitk_ws_floats"] > 0
