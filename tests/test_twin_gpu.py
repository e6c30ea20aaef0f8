"""Twin batches on the GPU (include/ganslate_hip.h gs_twin, csrc/hconvw.hip): ONE launch over the images of two networks —
the kernel picks the weight set per image, and with more tiles than CUs every workgroup walks several tiles (persistent
form) — must give, bit for bit, what the two separate launches give (and those are pinned to the oracle in
tests/test_ops_gpu.py). Then the recipe level: CycleGAN iterations with twin passes against the two-pass form.
resnet2d.py:80-87 (residual convs), cyclegan.py:126-189 (which passes are independent)."""
import random

import pytest
import torch

from ganslate_amd.nn.native.spec import ConvSpec
from ganslate_amd.nn.native.twin import Twin
from oracle.ops_ref import RefOps

from .helpers import build_product_cyclegan, golden_inputs, load_golden_steps
from .test_ops_gpu import close_bf16, close_f32, make_layer, stats_slots

pytestmark = pytest.mark.gpu

# (channels, images per network, H, W): the headline trunk (512 tiles: two per workgroup), three tiles per workgroup with
# a ragged last round (and a persistent single launch), the smallest batch the kernel takes (6 x 32 = 192 tiles), a ragged
# box grid (60 tiles per image: 4 images per round), a wide map (more tiles per image than a round: one tile per workgroup)
TRUNK_CASES = [(256, 8, 64, 64), (256, 10, 64, 64), (256, 6, 64, 64), (128, 4, 96, 160), (256, 1, 256, 256)]


def _two_layers(C, H, W):
    spec = ConvSpec("conv", C, C, 3, 1, 1, pad_mode="reflect")
    la = make_layer(spec, (H, W), 101)
    lb = make_layer(spec, (H, W), 202)
    return spec, la, lb


@pytest.mark.parametrize("case", TRUNK_CASES, ids=lambda c: "x".join(map(str, c)))
def test_twin_forward_equals_two_launches(hip_ops, case):
    C, N, H, W = case
    dev = hip_ops.device
    spec, (low, _, bias_a, fpack_a, _), (_, _, bias_b, fpack_b, _) = _two_layers(C, H, W)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2 * N, H, W, C, generator=g).to(torch.bfloat16).to(dev)
    packs = torch.stack([fpack_a, fpack_b]).to(dev)           # one allocation: any delta works, this one is small
    biases = torch.stack([bias_a, bias_b]).to(dev)
    slots, offs = stats_slots(hip_ops, low, low.fwd, 2 * N)
    assert slots == stats_slots(hip_ops, low, low.fwd, N)[0], "slot count per image must not depend on the batch"

    def run(xs, pack, bias, n):
        y = torch.zeros(n, H, W, C, dtype=torch.bfloat16, device=dev)
        part = torch.full((n * slots * 2 * C,), float("nan"), dtype=torch.float32, device=dev)
        hip_ops.gconv_classes(low.fwd, xs, pack, bias, y, act="none", stats=part, stats_slots=slots, stats_slot0s=offs)
        return y, part
    y_tw, p_tw = run(x, Twin(packs[0], packs[1]), Twin(biases[0], biases[1]), 2 * N)
    y_a, p_a = run(x[:N], packs[0], biases[0], N)
    y_b, p_b = run(x[N:], packs[1], biases[1], N)
    torch.cuda.synchronize()
    assert torch.equal(y_tw[:N], y_a) and torch.equal(y_tw[N:], y_b), "twin launch differs from the two launches"
    assert torch.equal(p_tw, torch.cat([p_a, p_b])), "statistics slots differ"
    assert not torch.isnan(p_tw).any()
    # and against the oracle (first network's half; the other half is the same kernel with another pointer)
    ref = RefOps()
    y_ref = torch.zeros(N, H, W, C, dtype=torch.bfloat16)
    ref.gconv(low.fwd[0], x[:N].cpu(), fpack_a, bias_a, y_ref)
    close_bf16(y_tw[:N], y_ref, "twin forward vs oracle")


IM2COL_CASES = [      # PatchGAN layers (patchgan2d.py:29-62): the im2col kernel picks the weight set per pixel tile
    (ConvSpec("conv", 64, 128, 4, 2, 1), 8, 128, 128),
    (ConvSpec("conv", 128, 256, 4, 2, 1), 16, 64, 64),
    (ConvSpec("conv", 256, 512, 4, 1, 1), 3, 32, 32),       # 31 x 31 outputs: ragged last tile of every image
    (ConvSpec("conv", 3, 64, 4, 2, 1), 2, 64, 64),
]


@pytest.mark.parametrize("case", IM2COL_CASES, ids=lambda c: f"{c[0].cin}to{c[0].cout}_k{c[0].k}s{c[0].stride}_n{c[1]}")
def test_twin_forward_on_the_im2col_kernel(hip_ops, case):
    """gconv_kernel with a gs_twin: conv outputs bit for bit those of the two launches (the K order does not depend on the
    tile), statistics to summation order (a batch of 2N may pick a larger pixel tile: other slots), and the oracle."""
    spec, N, H, W = case
    dev = hip_ops.device
    low, _, bias_a, fpack_a, _ = make_layer(spec, (H, W), 301)
    _, _, bias_b, fpack_b, _ = make_layer(spec, (H, W), 302)
    g0 = low.fwd[0]
    if not hip_ops.twin_native(g0, 2 * N):      # (halves that would run split-K stay two launches)
        pytest.skip("the halves of this layer run split-K: no twin launch")
    g = torch.Generator().manual_seed(6)
    x = torch.zeros(2 * N, H, W, g0.Ci, dtype=torch.bfloat16)
    x[..., :spec.cin] = torch.randn(2 * N, H, W, spec.cin, generator=g).to(torch.bfloat16)
    x = x.to(dev)
    packs = torch.stack([fpack_a, fpack_b]).to(dev)
    biases = torch.stack([bias_a, bias_b]).to(dev)

    def run(xs, pack, bias, n, twin=False):
        slots = hip_ops.stat_slots(g0, n, twin=twin) if twin else hip_ops.stat_slots(g0, n)
        y = torch.zeros(n, *low.out_dims, g0.Co, dtype=torch.bfloat16, device=dev)
        part = torch.zeros(n * slots * 2 * g0.Co, dtype=torch.float32, device=dev)
        hip_ops.gconv(g0, xs, pack, bias, y, act="lrelu", stats=part, stats_slots=slots)
        return y, part.view(n, slots, 2, g0.Co).double().sum(1)
    y_tw, s_tw = run(x, Twin(packs[0], packs[1]), Twin(biases[0], biases[1]), 2 * N, twin=True)
    y_a, s_a = run(x[:N], packs[0], biases[0], N)
    y_b, s_b = run(x[N:], packs[1], biases[1], N)
    torch.cuda.synchronize()
    assert torch.equal(y_tw[:N], y_a) and torch.equal(y_tw[N:], y_b), "twin launch differs from the two launches"
    want = torch.cat([s_a, s_b])
    assert (s_tw - want).abs().max().item() <= 1e-5 * want.abs().max().item()
    y_ref = torch.zeros(N, *low.out_dims, g0.Co, dtype=torch.bfloat16)
    RefOps().gconv(g0, x[:N].cpu(), fpack_a, bias_a, y_ref, act="lrelu")
    close_bf16(y_tw[:N], y_ref, "twin forward vs oracle")


@pytest.mark.parametrize("case", [
    (ConvSpec("conv", 3, 64, 7, 1, 3, pad_mode="reflect", wfold="in"), 2, 256, 256),
    (ConvSpec("conv", 64, 3, 7, 1, 3, pad_mode="reflect", wfold="out"), 2, 256, 256),
], ids=["stem", "out"])
@pytest.mark.parametrize("regs", [0, 2])
def test_twin_boundary_convs_on_the_strip_kernels(hip_ops, case, regs):
    """hstrip.hip with a gs_twin — one tile per workgroup (weights chosen per tile) and the persistent register-weight form
    (workgroups split between the networks; option value 2 takes every launch): forward and data gradient, with and
    without statistics, bit for bit the two separate launches"""
    spec, N, H, W = case
    dev = hip_ops.device
    low, _, bias_a, fpack_a, dpack_a = make_layer(spec, (H, W), 311)
    _, _, bias_b, fpack_b, dpack_b = make_layer(spec, (H, W), 312)
    default = hip_ops.get_option("hstrip_regs")
    hip_ops.set_option("hstrip_regs", regs)
    try:
        g = torch.Generator().manual_seed(9)
        for cls, pa, pb, ba, bb, with_stats in ((low.fwd[0], fpack_a, fpack_b, bias_a, bias_b, True),
                                                (low.fwd[0], fpack_a, fpack_b, bias_a, bias_b, False),
                                                (low.dgrad[0], dpack_a, dpack_b, None, None, False)):
            if len(low.dgrad) != 1 and cls is low.dgrad[0]:
                continue
            assert hip_ops.twin_native(cls, 2 * N), "the strip kernels take twin batches"
            x = torch.randn(2 * N, cls.Hi, cls.Wi, cls.Ci, generator=g).to(torch.bfloat16).to(dev)
            packs = torch.stack([pa, pb]).to(dev)
            biases = torch.stack([ba, bb]).to(dev) if ba is not None else None

            def run(xs, pack, bias, n, twin=False):
                slots = hip_ops.stat_slots(cls, n, twin=twin) if with_stats else 0
                y = torch.zeros(n, cls.Ho, cls.Wo, cls.Co, dtype=torch.bfloat16, device=dev)
                part = torch.zeros(max(n * slots * 2 * cls.Co, 1), dtype=torch.float32, device=dev)
                hip_ops.gconv(cls, xs, pack, bias, y, stats=part if with_stats else None, stats_slots=slots)
                # (per-image totals: a half of N images may run on another kernel — fewer tiles than the strip kernel's
                # minimum — with other slots)
                return y, (part.view(n, slots, 2, cls.Co).double().sum(1) if with_stats else part)
            y_tw, p_tw = run(x, Twin(packs[0], packs[1]), Twin(biases[0], biases[1]) if biases is not None else None,
                             2 * N, twin=True)
            y_a, p_a = run(x[:N], packs[0], biases[0] if biases is not None else None, N)
            y_b, p_b = run(x[N:], packs[1], biases[1] if biases is not None else None, N)
            torch.cuda.synchronize()
            assert torch.equal(y_tw[:N], y_a) and torch.equal(y_tw[N:], y_b), "twin launch differs from the two launches"
            if with_stats:
                want = torch.cat([p_a, p_b])
                assert (p_tw - want).abs().max().item() <= 1e-5 * want.abs().max().item(), "statistics differ"
    finally:
        hip_ops.set_option("hstrip_regs", default)


@pytest.mark.parametrize("case", [
    (ConvSpec("convT", 256, 128, 3, 2, 1, 1, dims=3), 1, 16, 32, 32),    # 8 classes; N = 1 / 2 pick different pixel tiles
    (ConvSpec("convT", 128, 64, 3, 2, 1, 1, dims=3), 1, 24, 24, 24),
    (ConvSpec("convT", 256, 128, 3, 2, 1, 1), 3, 32, 32),                # 2-D, small grid: two launches
    (ConvSpec("convT", 256, 128, 3, 2, 1, 1), 4, 64, 64),                # 2-D: the halo-resident class kernel, ONE launch
    (ConvSpec("convT", 128, 64, 3, 2, 1, 1), 2, 128, 128),
], ids=lambda c: f"{c[0].dims}d_{c[0].cin}to{c[0].cout}_n{c[1]}_{c[2]}")
def test_twin_batch_of_a_multi_class_layer(hip_ops, case):
    """the output-parity classes of a transposed conv with a twin batch run as two launches of N images each: the statistics
    slots must be planned for THAT batch (stat_slots(..., multi=True)) — a slot count taken from the 2N batch, whose launch
    would pick another pixel tile, scrambles the statistics. Outputs and per-image totals against the two separate launches."""
    spec, N, sizes = case[0], case[1], case[2:]
    dev = hip_ops.device
    low, _, bias_a, fpack_a, _ = make_layer(spec, sizes, 321)
    _, _, bias_b, fpack_b, _ = make_layer(spec, sizes, 322)
    g = torch.Generator().manual_seed(10)
    x = torch.randn(2 * N, *sizes, spec.cin_p, generator=g).to(torch.bfloat16).to(dev)
    packs = torch.stack([fpack_a, fpack_b]).to(dev)
    biases = torch.stack([bias_a, bias_b]).to(dev)

    def run(xs, pack, bias, n, twin):
        slots, offs = 0, []
        for cls in low.fwd:
            offs.append(slots)
            slots += hip_ops.stat_slots(cls, n, twin=twin, multi=low.fwd)
        y = torch.zeros(n, *low.out_dims, spec.cout_p, dtype=torch.bfloat16, device=dev)
        part = torch.zeros(n * slots * 2 * spec.cout_p, dtype=torch.float32, device=dev)
        hip_ops.gconv_classes(low.fwd, xs, pack, bias, y, stats=part, stats_slots=slots, stats_slot0s=offs)
        return y, part.view(n, slots, 2, spec.cout_p).double().sum(1)
    if spec.dims == 2 and sizes[0] >= 64:
        assert hip_ops.multi_twin_native(low.fwd, 2 * N), "the class kernel takes this twin batch as one launch"
    y_tw, s_tw = run(x, Twin(packs[0], packs[1]), Twin(biases[0], biases[1]), 2 * N, True)
    y_a, s_a = run(x[:N], packs[0], biases[0], N, False)
    y_b, s_b = run(x[N:], packs[1], biases[1], N, False)
    torch.cuda.synchronize()
    assert torch.equal(y_tw[:N], y_a) and torch.equal(y_tw[N:], y_b), "twin batch differs from the two launches"
    want = torch.cat([s_a, s_b])
    assert (s_tw - want).abs().max().item() <= 1e-5 * want.abs().max().item(), "statistics differ"


@pytest.mark.parametrize("persist", [1, 0])
def test_many_tiles_per_workgroup_equal_one_tile_each(hip_ops, persist):
    """one network, batch 24 at the trunk shape: 768 tiles = three per workgroup in the persistent form (option
    hconvw_persist), one each otherwise — same bits, and equal to the oracle"""
    C, N, H, W = 256, 24, 64, 64
    dev = hip_ops.device
    spec = ConvSpec("conv", C, C, 3, 1, 1, pad_mode="reflect")
    low, _, bias, fpack, _ = make_layer(spec, (H, W), 7)
    g = torch.Generator().manual_seed(8)
    x = torch.randn(N, H, W, C, generator=g).to(torch.bfloat16)
    default = hip_ops.get_option("hconvw_persist")
    try:
        hip_ops.set_option("hconvw_persist", persist)
        y = torch.zeros(N, H, W, C, dtype=torch.bfloat16, device=dev)
        hip_ops.gconv(low.fwd[0], x.to(dev), fpack.to(dev), bias.to(dev), y, act="relu")
        torch.cuda.synchronize()
    finally:
        hip_ops.set_option("hconvw_persist", default)
    ref = RefOps()
    y_ref = torch.zeros(N, H, W, C, dtype=torch.bfloat16)
    ref.gconv(low.fwd[0], x, fpack, bias, y_ref, act="relu")
    close_bf16(y, y_ref, f"forward, hconvw_persist={persist}")
    if persist == 1:
        test_many_tiles_per_workgroup_equal_one_tile_each.y = y.cpu()
    elif hasattr(test_many_tiles_per_workgroup_equal_one_tile_each, "y"):
        assert torch.equal(y.cpu(), test_many_tiles_per_workgroup_equal_one_tile_each.y)


@pytest.mark.parametrize("case", TRUNK_CASES[:4], ids=lambda c: "x".join(map(str, c)))
@pytest.mark.parametrize("with_g2,act", [(False, "relu"), (True, "none")])
def test_twin_ring_dgrad_equals_two_launches(hip_ops, case, with_g2, act):
    """the fused data gradient on the unpadded domain (hconvw.hip RING) as a twin launch: gradient and the per-box sums of
    the consumer's norm backward, bit for bit, and the first half against the oracle"""
    C, N, H, W = case
    dev = hip_ops.device
    spec, (low, _, _, _, dpack_a), (_, _, _, _, dpack_b) = _two_layers(C, H, W)
    g = torch.Generator().manual_seed(9)
    gy = torch.randn(2 * N, H, W, C, generator=g).to(torch.bfloat16).to(dev)
    y = (torch.randn(2 * N, H, W, C, generator=g) * 1.5 + 0.2).to(torch.bfloat16).to(dev)
    g2 = torch.randn(2 * N, H, W, C, generator=g).to(torch.bfloat16).to(dev) if with_g2 else None
    packs = torch.stack([dpack_a, dpack_b]).to(dev)
    part = torch.stack([y.float().sum((1, 2)), (y.float() ** 2).sum((1, 2))], 1).reshape(-1).contiguous()
    mr = torch.empty(2 * N * 2 * C, dtype=torch.float32, device=dev)
    hip_ops.inorm_finalize(part, 2 * N, 1, C, H * W, mr)

    def run(sl, pack, n):
        ring = hip_ops.fused_ring_plan(low.dgrad_ring, n, C)
        assert ring is not None
        gx = torch.zeros(n, H, W, C, dtype=torch.bfloat16, device=dev)
        fz = {"y": y[sl], "mean_rstd": mr[sl.start * 2 * C:sl.stop * 2 * C], "g2": None if g2 is None else g2[sl],
              "partial": ring[1], "fold": 1, "fold_mode": "reflect", "act": act, "slope": 0.2}
        hip_ops.gconv(low.dgrad_ring, gy[sl], pack, None, gx, fuse=fz)
        return gx, ring[1][:n * ring[0] * 3 * C].clone(), ring
    gx_tw, s_tw, ring_tw = run(slice(0, 2 * N), Twin(packs[0], packs[1]), 2 * N)
    gx_a, s_a, _ = run(slice(0, N), packs[0], N)
    gx_b, s_b, _ = run(slice(N, 2 * N), packs[1], N)
    torch.cuda.synchronize()
    assert torch.equal(gx_tw[:N], gx_a) and torch.equal(gx_tw[N:], gx_b), "twin ring launch differs from the two launches"
    assert torch.equal(s_tw, torch.cat([s_a, s_b])), "norm-backward sums differ"
    # oracle, first half
    ref = RefOps()
    ref.ring_min_blocks = 0
    yc, mrc = y[:N].cpu(), mr[:N * 2 * C].cpu()
    ring = ref.fused_ring_plan(low.dgrad_ring, N, C)
    gx_ref = torch.zeros(N, H, W, C, dtype=torch.bfloat16)
    ref.gconv(low.dgrad_ring, gy[:N].cpu(), dpack_a, None, gx_ref,
              fuse={"y": yc, "mean_rstd": mrc, "g2": None if g2 is None else g2[:N].cpu(), "partial": ring[1], "fold": 1,
                    "fold_mode": "reflect", "act": act, "slope": 0.2})
    close_bf16(gx_tw[:N], gx_ref, "twin ring dgrad vs oracle")
    sums = s_tw.view(2 * N, ring_tw[0], 3, C).sum(1)[:N]
    close_f32(sums, ring[1][:N * 3 * C].view(N, 3, C), "ring partial sums vs oracle", rel=3e-3)


@pytest.mark.parametrize("case", [(256, 8, 64, 64), (256, 3, 64, 64), (128, 2, 96, 160)], ids=lambda c: "x".join(map(str, c)))
@pytest.mark.parametrize("pair", [True, False])
def test_twin_weight_gradient_of_the_residual_convs(hip_ops, case, pair):
    """hwgrad_wide as a twin launch (gs_wgrad_ws_twin): both networks' gradients out of one launch with half the pixel splits
    per network — equal to the two separate launches up to the fp32 summation order, to the oracle, and bit-identical
    between two runs (slabs added in a fixed order)"""
    C, N, H, W = case
    dev = hip_ops.device
    spec = ConvSpec("conv", C, C, 3, 1, 1, pad_mode="reflect")
    low = make_layer(spec, (H, W), 3)[0]
    g = torch.Generator().manual_seed(4)
    mk = lambda: torch.randn(2 * N, H, W, C, generator=g).to(torch.bfloat16).to(dev)
    dy, x, dy2, x2 = mk(), mk(), mk(), mk()
    pr = (dy2, x2) if pair else None
    ph = lambda h: None if pr is None else (dy2[h * N:(h + 1) * N], x2[h * N:(h + 1) * N])
    n = spec.master_numel
    runs = []
    for _ in range(2):
        dw = torch.zeros(2, n, device=dev)
        hip_ops.wgrad(low.wgrad, dy, x, Twin(dw[0], dw[1]), pair=pr)
        runs.append(dw)
    sep = torch.zeros(2, n, device=dev)
    for h in (0, 1):
        hip_ops.wgrad(low.wgrad, dy[h * N:(h + 1) * N], x[h * N:(h + 1) * N], sep[h], pair=ph(h))
    torch.cuda.synchronize()
    assert torch.equal(runs[0], runs[1]), "twin weight gradient is not reproducible"
    close_f32(runs[0], sep.cpu(), "twin vs separate launches", rel=2e-4)
    ref, dref = RefOps(), torch.zeros(n)
    ref.wgrad(low.wgrad, dy[N:].cpu(), x[N:].cpu(), dref, pair=None if pr is None else (dy2[N:].cpu(), x2[N:].cpu()))
    close_f32(runs[0][1], dref, "second network's gradient vs oracle", rel=2e-3)


@pytest.mark.parametrize("case", IM2COL_CASES + [
    (ConvSpec("convT", 256, 128, 3, 2, 1, 1), 4, 32, 32),
    # the W-folded k7 boundary convs: hwgrad_ft's twin form (workgroups split between the networks)
    (ConvSpec("conv", 3, 64, 7, 1, 3, pad_mode="reflect", wfold="in"), 2, 256, 256),
    (ConvSpec("conv", 64, 3, 7, 1, 3, pad_mode="reflect", wfold="out"), 2, 256, 256)],
                         ids=lambda c: f"{c[0].kind}{c[0].cin}to{c[0].cout}_k{c[0].k}s{c[0].stride}_n{c[1]}")
def test_twin_weight_gradient_on_the_im2col_kernel(hip_ops, case):
    """wgrad_kernel as a twin launch: each network's pixel range is split on its own and lands in its own gradient buffer
    — against the two separate launches (summation order), the oracle, and itself (reproducible)."""
    spec, N, H, W = case
    dev = hip_ops.device
    low = make_layer(spec, (H, W), 7)[0]
    w = low.wgrad
    g = torch.Generator().manual_seed(8)
    a = torch.randn(2 * N, w.Ha, w.Wa, w.P, generator=g).to(torch.bfloat16).to(dev)      # dense side, gathered side
    gg = torch.randn(2 * N, w.Hg, w.Wg, w.Q, generator=g).to(torch.bfloat16).to(dev)
    n = w.P * w.T * w.Q
    runs = []
    for _ in range(2):
        dw = torch.zeros(2, n, device=dev)
        hip_ops.wgrad(w, a, gg, Twin(dw[0], dw[1]))
        runs.append(dw)
    sep = torch.zeros(2, n, device=dev)
    for h in (0, 1):
        hip_ops.wgrad(w, a[h * N:(h + 1) * N], gg[h * N:(h + 1) * N], sep[h])
    torch.cuda.synchronize()
    assert torch.equal(runs[0], runs[1]), "twin weight gradient is not reproducible"
    close_f32(runs[0], sep.cpu(), "twin vs separate launches", rel=2e-4)
    dref = torch.zeros(n)
    RefOps().wgrad(w, a[N:].cpu(), gg[N:].cpu(), dref)
    close_f32(runs[0][1], dref, "second network's gradient vs oracle", rel=2e-3)


@pytest.mark.parametrize("name", ["c64_default", "c64_idt_ssim"])
def test_cyclegan_step_with_twin_passes_equals_the_two_pass_step(name, monkeypatch):
    """whole iterations on the GPU: twin passes (generators and discriminators as one batch each) against GS_TWIN=0. The
    convs of a twin batch are the kernels of the separate passes (same tiles, same order), so the first iteration's losses
    agree to fp32 summation noise; weights after two Adam steps stay within sign-flip noise."""
    c = load_golden_steps()[name]["config"]
    runs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("GS_TWIN", mode)
        monkeypatch.setenv("GS_STEP_GRAPH", "0")
        model = build_product_cyclegan(c)
        assert (model.twin_G is not None) == (mode == "1")
        random.seed(c["seed"])
        out = []
        for step in range(2):
            a, b = golden_inputs(c, step)
            model.set_input({"A": a, "B": b})
            model.optimize_parameters()
            out.append({k: float(v.detach()) for k, v in model.losses.items() if v is not None})
        torch.cuda.synchronize()
        runs[mode] = (out, {n: net.master.detach().float().cpu().clone() for n, net in model.networks.items()})
    for s in range(2):
        for k, v in runs["0"][0][s].items():
            assert runs["1"][0][s][k] == pytest.approx(v, rel=2e-3 if s == 0 else 5e-2, abs=1e-5), (s, k)
    for n, w in runs["0"][1].items():
        assert (runs["1"][1][n] - w).abs().mean().item() <= 1e-4, n


def test_twin_vnets_on_the_hip_kernels(hip_ops):
    """the twin form of the V-Net executor (GS_TWIN=all in the recipes; off by default: slower on the brats recipe) on the HIP
    backend: outputs, input gradients and flat gradients of both networks against the two separate passes — the same kernels
    run either way for the per-network launches, the twin-native conv / weight-gradient launches are pinned at op level above"""
    from ganslate_amd.nn.generators import Vnet3D
    from ganslate_amd.nn.native.twin import TwinNet
    make = lambda: Vnet3D(1, 1, "instance", first_layer_channels=16, down_blocks=(1, 2), up_blocks=(2, 1),
                          use_memory_saving=False, use_inverse=False)
    torch.manual_seed(21)
    a, b = make(), make()
    a.init_weights("normal", 0.05); b.init_weights("normal", 0.05)
    a1, b1 = make(), make()
    a1.load_state_dict(a.state_dict()); b1.load_state_dict(b.state_dict())
    assert TwinNet.compatible(a, b)
    dev = hip_ops.device
    g = torch.Generator().manual_seed(22)
    xa, xb = ((torch.rand(1, 1, 16, 32, 32, generator=g) * 2 - 1).to(dev) for _ in range(2))
    ga, gb = (torch.randn(1, 1, 16, 32, 32, generator=g).to(dev) for _ in range(2))
    xs = [t.clone().requires_grad_() for t in (xa, xb, xa, xb)]
    ya, yb = TwinNet(a, b)(xs[0], xs[1])
    ((ya * ga).sum() + (yb * gb).sum()).backward()
    ya1, yb1 = a1(xs[2]), b1(xs[3])
    ((ya1 * ga).sum() + (yb1 * gb).sum()).backward()
    torch.cuda.synchronize()

    def close(x, y, tol):
        scale = max(y.abs().max().item(), 1e-12)
        assert (x - y).abs().max().item() <= tol * scale, ((x - y).abs().max().item(), scale)
    close(ya.detach(), ya1.detach(), 2e-2); close(yb.detach(), yb1.detach(), 2e-2)
    close(xs[0].grad, xs[2].grad, 5e-2); close(xs[1].grad, xs[3].grad, 5e-2)
    for net, ref in ((a, a1), (b, b1)):
        cos = torch.nn.functional.cosine_similarity(net.master.grad, ref.master.grad, dim=0).item()
        assert cos > 0.995, cos
