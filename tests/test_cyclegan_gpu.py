"""Network- and step-level parity of the HIP path (bf16 storage, fp32 accumulate) against the fp32 oracle and the
golden vectors of the real reference. Calls go through the C ABI (ganslate_amd.hip.ops -> libganslate_hip.so).

Stated tolerances (bf16 has 8 mantissa bits; a ResNet-9 pass chains ~24 convs + norms):
  * network outputs: relative L2 error <= 3e-2 and max |err| <= 0.12 * max|ref| (the CPU emulation of the same bf16
    storage, oracle backend with act_dtype=bf16, shows rms 1.1e-2 / max 5.7e-2 on ResNet-9 at 64x64);
    gradients: relative L2 error <= 0.15 and cosine >= 0.99 vs fp32 (see _net_case), <= 0.12 vs the bf16 emulation;
  * step 0 of a training step (same weights, same batch): every loss within 2e-2 relative of the reference's;
  * later steps: inside the reference-vs-reference envelope (see tests/test_cyclegan_cpu.py / DESIGN.md §5).
"""
import pytest
import torch

from oracle import torch_ref

from .helpers import build_product_cyclegan, load_golden_steps, run_product_steps

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    return ((a - b).norm() / (b.norm() + 1e-12)).item()


def cosine(a, b):
    return (a.flatten() @ b.flatten() / (a.norm() * b.norm() + 1e-20)).item()


def _net_case(hip_ops, build_native, shadow, x_shape, seed, grad_tol=0.15, grad_cos=0.99):
    """(a) HIP vs the SAME executor on the CPU oracle backend with bf16 storage: identical rounding points, only
    the accumulation order differs -> tight. (b) HIP vs the fp32 torch restatement -> bf16-level tolerance.
    Gradient tolerance in (b): rounding a pre-activation to bf16 flips the ReLU/LeakyReLU slope of the ~0.5 % of
    elements that sit within one bf16 ulp of the kink, which alone is a 6-8 % relative-L2 change of the gradient
    (measured on the CPU emulation: 0.3 % error entering an InstanceNorm+LeakyReLU backward, 6.1 % leaving it)."""
    from ganslate_amd.nn.native import backend
    from oracle.ops_ref import RefOps
    sd = torch_ref.seeded_state_dict(shadow, seed)
    shadow.load_state_dict(sd)
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(x_shape, generator=g) * 2 - 1
    gy = None
    results = {}
    for name, ops in (("hip", hip_ops), ("cpu_bf16", RefOps(act_dtype=torch.bfloat16))):
        backend.set_ops(ops)
        try:
            net = build_native()
            net.load_state_dict(sd)
            xi = x.clone().to(ops.device).requires_grad_()
            y = net(xi)
            if gy is None:
                gy = torch.randn(y.shape, generator=g)
            y.backward(gy.to(ops.device))
            if ops.device.type == "cuda":
                torch.cuda.synchronize()
            results[name] = (y.detach().cpu(), xi.grad.cpu(), {k: v.cpu() for k, v in net.grads_state_dict().items()},
                             {nd.name for nd in net.nodes if nd.norm})
        finally:
            backend.set_ops(hip_ops)
    xa = x.clone().requires_grad_()
    ya = shadow(xa)
    ya.backward(gy)
    yh, gxh, gh, normed = results["hip"]
    yc, gxc, gc, _ = results["cpu_bf16"]
    # (a) kernels vs emulation
    # (accumulation-order noise moves a few values across a bf16 rounding boundary, and from there across an
    #  activation kink, so even the emulation is only statistically equal; op-level tests are the tight ones)
    assert rel_l2(yh, yc) <= 2e-2, rel_l2(yh, yc)
    assert rel_l2(gxh, gxc) <= grad_tol, rel_l2(gxh, gxc)
    # (b) vs the fp32 reference restatement
    assert rel_l2(yh, ya.detach()) <= 3e-2
    assert (ya.detach() - yh).abs().max().item() <= 0.12 * ya.abs().max().item()
    assert rel_l2(gxh, xa.grad) <= grad_tol and cosine(gxh, xa.grad) >= grad_cos, (rel_l2(gxh, xa.grad), cosine(gxh, xa.grad))
    for n, p in shadow.named_parameters():
        if n.startswith("encoder.") or (n.endswith(".bias") and n[:-5] in normed):
            continue  # aliases / biases in front of an InstanceNorm (exactly-zero true gradient)
        if n.endswith("key_conv.bias"):
            # SelfAttentionBlock: a key bias shifts every logit of a row by the same amount -> exactly-zero true gradient
            assert gh[n].norm().item() <= 5e-2 * gh[n.replace("key_conv", "query_conv")].norm().item() + 1e-6, n
            continue
        if ".query_conv." in n or ".key_conv." in n:
            # SelfAttentionBlock with near-uniform attention (small seeded weights: logits ~ 0): the gradient reaches q and
            # k only in second order — when it is below 1 % of the value conv's it is rounding noise on both sides
            vref = dict(shadow.named_parameters())[n.split(".query_conv.")[0].split(".key_conv.")[0] + ".value_conv.weight"]
            if p.grad.norm().item() <= 1e-2 * vref.grad.norm().item():
                assert gh[n].norm().item() <= 2e-2 * vref.grad.norm().item(), n
                continue
        if n.endswith(".gamma"):
            # SelfAttentionBlock: ONE scalar = sum over voxels x channels of dout * (A v), random-sign terms that cancel;
            # bf16 storage of either factor leaves sqrt(n) * 2^-9 un-averaged (tests/test_ops_gpu.py::
            # test_self_attention_block_forward_backward bounds it). Sign and order of magnitude only; the weight tensors
            # behind the block (its q / k / v convs) are compared like every other layer
            assert gh[n].item() * p.grad.item() > 0 and 0.2 <= gh[n].item() / p.grad.item() <= 5.0, (n, gh[n], p.grad)
            continue
        assert rel_l2(gh[n], gc[n]) <= grad_tol, (n, rel_l2(gh[n], gc[n]))
        assert rel_l2(gh[n], p.grad) <= grad_tol and cosine(gh[n], p.grad) >= grad_cos, (n, rel_l2(gh[n], p.grad))


def test_resnet2d_hip_vs_oracle(hip_ops):
    from ganslate_amd.nn.generators import Resnet2D
    _net_case(hip_ops, lambda: Resnet2D(3, 3, "instance", 9), torch_ref.Resnet2D(3, 3, 9), (2, 3, 64, 64), 41,
              grad_tol=0.30, grad_cos=0.95)   # 21 InstanceNorm+ReLU layers: kink flips compound (0.20 / 0.98 measured)


def test_resnet2d_ragged_hip_vs_oracle(hip_ops):
    from ganslate_amd.nn.generators import Resnet2D
    _net_case(hip_ops, lambda: Resnet2D(3, 3, "instance", 3), torch_ref.Resnet2D(3, 3, 3), (1, 3, 40, 56), 42,
              grad_tol=0.30, grad_cos=0.95)


@pytest.mark.parametrize("in_ch,n_layers,hw", [(3, 3, (64, 64)), (6, 4, (96, 128)), (3, 3, (256, 256))])
def test_patchgan2d_hip_vs_oracle(hip_ops, in_ch, n_layers, hw):
    from ganslate_amd.nn.discriminators import PatchGAN2D
    _net_case(hip_ops, lambda: PatchGAN2D(in_ch, 64, n_layers, (4, 4), "instance"),
              torch_ref.PatchGAN2D(in_ch, 64, n_layers, 4), (1, in_ch, *hw), 43)


def test_resnet2d_matches_reference_golden(hip_ops):
    """same net, weights and input as the golden case generated from the real reference"""
    import json
    from pathlib import Path
    from ganslate_amd.nn.generators import Resnet2D
    gold = json.loads((Path(__file__).parent / "golden" / "nets.json").read_text())["resnet2d_64"]
    net, shadow = Resnet2D(3, 3, "instance", 9), torch_ref.Resnet2D(3, 3, 9)
    net.load_state_dict(torch_ref.seeded_state_dict(shadow, gold["seed"]))
    g = torch.Generator().manual_seed(gold["seed"])
    x = torch.rand(gold["x_shape"], generator=g) * 2 - 1
    y = net(x.to(hip_ops.device)).cpu().flatten()
    ref = torch.tensor(gold["y_samples"])
    assert (y[gold["sample_idx"]] - ref).abs().max().item() <= 0.12 * ref.abs().max().item()
    assert rel_l2(y[gold["sample_idx"]], ref) <= 5e-2   # 32 samples only
    assert abs(y.double().abs().sum().item() - gold["y_abs_sum"]) <= 2e-2 * gold["y_abs_sum"]


from .envelope import step_tolerance  # noqa: E402


def _check_steps(got, gold_steps, n_steps, metrics=True):
    for s in range(n_steps):
        g = gold_steps[s]
        assert got[s]["lrs"] == pytest.approx(g["lrs"], abs=1e-12)
        assert set(got[s]["losses"]) == set(g["losses"])
        for k, v in g["losses"].items():
            assert got[s]["losses"][k] == pytest.approx(v, rel=step_tolerance(k, s)), (s, k, got[s]["losses"][k], v)
        if metrics and s == 0:
            for k, v in g["metrics"].items():
                assert got[s]["metrics"][k] == pytest.approx(v, rel=2e-2, abs=1e-2), (s, k)


@pytest.mark.parametrize("name", ["c64_default", "c64_idt_ssim", "cfg1_256"])
def test_training_step_matches_reference_golden(hip_ops, name):
    gold = load_golden_steps()[name]
    c = gold["config"]
    n_steps = min(c["steps"], 6)
    model = build_product_cyclegan(c)
    got = run_product_steps(model, c, n_steps)
    _check_steps(got, gold["steps"], n_steps)


@pytest.mark.parametrize("name", ["c64_vanilla", "c64_wgangp"])
def test_training_step_with_other_adversarial_objectives(hip_ops, name):
    """`adversarial_loss_type` vanilla / wgangp (adversarial_loss.py:31-34,60-67) through the whole captured step against
    four iterations of the real reference (tests/golden/adv_modes.json). The wgangp terms are differences of means of
    order 0.5 that pass through zero, so the adversarial family gets an absolute floor of 2e-2 (= the relative bf16
    term of the lsgan case on quantities of order one) next to the envelope-derived relative tolerance."""
    import json
    from .helpers import GOLD
    gold = json.loads((GOLD / "adv_modes.json").read_text())["steps"][name]
    c = gold["config"]
    model = build_product_cyclegan(c, (f"train.gan.optimizer.adversarial_loss_type={c['adv']}",))
    got = run_product_steps(model, c, c["steps"])
    assert model._graph is not None, "the step was never captured"
    from .envelope import family
    for s in range(c["steps"]):
        g = gold["steps"][s]
        assert got[s]["lrs"] == pytest.approx(g["lrs"], abs=1e-12)
        for k, v in g["losses"].items():
            floor = 2e-2 * (s + 1) if family(k) == "adv" else 0.0
            assert got[s]["losses"][k] == pytest.approx(v, rel=step_tolerance(k, s), abs=floor), \
                (s, k, got[s]["losses"][k], v)


def test_training_step_at_headline_shape_matches_reference_golden(hip_ops):
    """BASELINE configs[1] (256x256, batch 8: the shape bench.py times, with its own tile selection — gconv<320,128>,
    hconvw, hwgrad_wide pairs) against two iterations of the real reference (tests/golden/cyclegan_grads.json)"""
    from .helpers import load_golden_grads
    gold = load_golden_grads()["cfg2_256_b8"]
    c = gold["config"]
    model = build_product_cyclegan(c)
    got = run_product_steps(model, c, 2)
    _check_steps(got, gold["steps"], 2)


def test_loss_curve_100_iterations_inside_the_reference_envelope(hip_ops):
    """north_star: "loss curves matching reference over 100 steps". The reference does not match ITSELF point-wise over
    100 iterations when only its thread count changes (tests/golden/envelope.json: 1e-3 at iteration 3, tens of percent
    from iteration ~10); what is checked is that the HIP path stays inside that scatter: every iteration within the
    envelope-derived tolerance while that is below 50 %, and decade means afterwards."""
    from .envelope import family, reference_curve, window_tolerance
    c, curve = reference_curve()
    model = build_product_cyclegan(c)
    got = run_product_steps(model, c, c["steps"])
    keys = list(curve[0]["losses"])
    worst = {}
    for s in range(c["steps"]):
        for k in keys:
            tol = step_tolerance(k, s)
            if tol < 0.5:
                ref, mine = curve[s]["losses"][k], got[s]["losses"][k]
                worst[k] = max(worst.get(k, 0.0), abs(mine - ref) / abs(ref) / tol)
                assert mine == pytest.approx(ref, rel=tol), (s, k, mine, ref, tol)
    for k in keys:
        ref = torch.tensor([s["losses"][k] for s in curve])
        mine = torch.tensor([s["losses"][k] for s in got])
        for lo in range(0, c["steps"], 10):
            r, m = ref[lo:lo + 10].mean().item(), mine[lo:lo + 10].mean().item()
            tol = window_tolerance(family(k), 10)      # 3 x the 1-vs-8-thread gap of the decade means: 2 % / 31 %
            assert abs(m - r) <= tol * abs(r), (k, lo, m, r, tol)
    print("\nworst |err| / tolerance per loss:", {k: round(v, 3) for k, v in worst.items()})


def test_loss_curve_stays_in_reference_envelope(hip_ops):
    """30 steps of the horse2zebra configuration at 64x64: window means of every loss against the reference's"""
    gold = load_golden_steps()["c64_default"]
    c = gold["config"]
    model = build_product_cyclegan(c)
    got = run_product_steps(model, c, c["steps"])
    for k in gold["steps"][0]["losses"]:
        ref = torch.tensor([s["losses"][k] for s in gold["steps"]])
        mine = torch.tensor([s["losses"][k] for s in got])
        for lo, hi in ((0, 10), (10, 20), (20, 30)):
            r, m = ref[lo:hi].mean().item(), mine[lo:hi].mean().item()
            from .envelope import family, window_tolerance
            assert abs(m - r) <= window_tolerance(family(k), 10) * abs(r), (k, lo, m, r)
