"""Inference path on the HIP generators: `BaseGAN.infer` and patch-wise inference of a volume larger
than the window — the product's sliding-window inferer driving the HIP Resnet3D against the loop-level MONAI restatement
(oracle/monai_ref.py) driving the fp32 torch restatement of the same network. Tolerance: bf16 forward (rel-L2 3e-2)."""
import pytest
import torch

from ganslate_amd.utils.sliding_window_inferer import SlidingWindowInferer
from oracle import monai_ref, torch_ref

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    return ((a - b).norm() / (b.norm() + 1e-12)).item()


@pytest.mark.parametrize("mode,overlap,sw", [("gaussian", 0.25, 2), ("constant", 0.5, 4)])
def test_sliding_window_volume_inference_hip_vs_oracle(hip_ops, mode, overlap, sw):
    from ganslate_amd.nn.generators import Resnet3D
    shadow = torch_ref.Resnet3D(1, 1, 2)
    sd = torch_ref.seeded_state_dict(shadow, 81)
    shadow.load_state_dict(sd)
    net = Resnet3D(1, 1, "instance", 2)
    net.load_state_dict(sd)
    net.eval()
    g = torch.Generator().manual_seed(82)
    x = torch.rand(1, 1, 32, 48, 40, generator=g) * 2 - 1
    calls = []

    def hip_predict(w):
        calls.append(tuple(w.shape))
        with torch.no_grad():
            return net(w)
    got = SlidingWindowInferer((32, 32, 32), sw, overlap, mode, cval=-1)(x.to(hip_ops.device), hip_predict).cpu()
    with torch.no_grad():
        want = monai_ref.sliding_window_inference(x, [32, 32, 32], sw, lambda w: shadow(w), overlap, mode, -1.0)
    assert got.shape == want.shape == x.shape
    assert rel_l2(got, want) <= 3e-2, rel_l2(got, want)
    assert all(s[1:] == (1, 32, 32, 32) for s in calls) and max(s[0] for s in calls) == sw


def test_infer_is_a_no_grad_forward_of_the_generator(hip_ops):
    from .helpers import build_product_cyclegan, golden_inputs, load_golden_steps
    c = dict(load_golden_steps()["c64_default"]["config"])
    model = build_product_cyclegan(c)
    A, _ = golden_inputs(c, 0)
    out = model.infer(A.to(model.device))
    assert out.shape == A.shape and not out.requires_grad
    assert torch.equal(out, model.networks["G_AB"](A.to(model.device)).detach())
    back = model.infer(A.to(model.device), direction="BA")
    assert not torch.equal(back, out)
