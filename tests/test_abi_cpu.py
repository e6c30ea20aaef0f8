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
    assert ctypes.sizeof(L.WGradDesc) == 18 * 4 + 3 * L.GS_MAX_TAPS
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
               "gs_norm_ex_desc": L.NormExDesc, "gs_pnorm_desc": L.PNormDesc, "gs_patchnce_desc": L.PatchNCEDesc,
               "gs_attn_desc": L.AttnDesc, "gs_attn_params": L.AttnParams}
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
    ad = L.AttnDesc(1, 1000, 256)
    # q | k in fp32, [q | k | v] and A v in bf16, the N x N logits (fp32) and probabilities (bf16), the backward's buffers
    assert calls["attn_work_bytes"] == lib.gs_attn_work_bytes(ctypes.byref(ad)) >= 1000 * 1000 * (4 + 2 + 2)


def test_ring_form_eligibility_rule_matches_the_oracle_restatement():
    """gs_gconv_ring_slots (pure host code, callable without a GPU) and RefOps.fused_ring_plan, which walks the same
    executor branch on CPU, must accept the same data-gradient classes: the zero-border, unpadded descriptor of a
    reflect-padded 3x3 stride-1 conv with Ci % 64 == Co % 128 == 0, sides multiples of 16 and >= 32, >= 192 boxes x
    channel tiles (hconvw.hip hconvw_ring_eligible)."""
    import types
    from ganslate_amd.hip import lib as L
    from ganslate_amd.hip.ops import HipOps
    from ganslate_amd.nn.native.spec import ConvSpec, lower
    from oracle.ops_ref import RefOps
    lib = L.load()
    ref = RefOps()
    shim = types.SimpleNamespace(_desc_cache={})
    seen = set()
    for cin, cout, N, H, W, mode in [(256, 256, 8, 64, 64, "reflect"), (256, 256, 1, 64, 64, "reflect"),
                                     (256, 256, 6, 64, 64, "reflect"), (128, 128, 48, 32, 32, "reflect"),
                                     (128, 64, 48, 32, 32, "reflect"), (256, 256, 64, 16, 16, "reflect"),
                                     (256, 256, 8, 64, 72, "reflect"), (256, 256, 2, 96, 128, "reflect"),
                                     (64, 128, 48, 32, 32, "reflect"), (256, 256, 8, 64, 64, "replicate"),
                                     (256, 256, 8, 64, 64, "zero")]:
        low = lower(ConvSpec("conv", cin, cout, 3, 1, 1, pad_mode=mode), H, W)
        g = low.dgrad_ring
        assert (g is not None) == (mode == "reflect")
        if g is None:
            continue
        assert (g.Hi, g.Wi, g.Ho, g.Wo, g.border) == (H, W, H, W, "zero")
        assert [(a, b) for a, b in zip(g.dh, g.dw)] == [(1 - t // 3, 1 - t % 3) for t in range(9)]
        d = HipOps._gdesc(shim, g, N, g.Ci, 0, g.Co, 0, "none", 0.0, 0, 0)
        slots = lib.gs_gconv_ring_slots(ctypes.byref(d))
        plan = ref.fused_ring_plan(g, N, low.spec.cin_p)
        assert (slots > 0) == (plan is not None), (cin, cout, N, H, W, slots)
        if slots:
            assert slots == (H // 16) * (W // 16)
        seen.add(slots > 0)
    assert seen == {True, False}
