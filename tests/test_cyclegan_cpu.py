"""The product recipe + executor (hand-written backward, folds, residual joins, flat Adam, LR schedule, image
pool) run on the CPU through the op-level oracle backend in fp32 and compared with the golden vectors of the real
reference. This pins all host-side logic of the training step without a GPU; kernels are pinned in -m gpu tests."""
import pytest
import torch

from ganslate_amd.nn.native import backend
from oracle.ops_ref import RefOps

from .helpers import build_product_cyclegan, load_golden_steps, run_product_steps


@pytest.fixture()
def fp32_oracle_backend():
    backend.set_ops(RefOps(act_dtype=torch.float32))
    yield
    backend.set_ops(None)


@pytest.mark.parametrize("name,n_steps", [("c64_default", 3), ("c64_idt_ssim", 2)])
def test_product_step_matches_reference_golden_fp32(fp32_oracle_backend, name, n_steps):
    gold = load_golden_steps()[name]
    c = gold["config"]
    model = build_product_cyclegan(c)
    got = run_product_steps(model, c, n_steps)
    for s in range(n_steps):
        g = gold["steps"][s]
        assert got[s]["lrs"] == pytest.approx(g["lrs"], abs=1e-12)
        assert set(got[s]["losses"]) == set(g["losses"])
        # Step 0 (same weights, same batch) is arithmetic parity: tight. From step 1 on Adam's first updates are
        # +-lr*sign(g), so reduction-order noise on near-zero gradients flips ~0.1 % of the weights by 2*lr and the
        # GAN dynamics amplify it: the reference restatement run with 1 thread instead of 8 drifts from its own
        # golden curve by 1-4 % within 7 steps. Later steps are therefore checked against that envelope.
        tol_adv, tol_cyc = (1e-4, 1e-4) if s == 0 else (0.10, 0.02)
        for k, v in g["losses"].items():
            tol = tol_cyc if k.startswith(("cycle", "idt")) else tol_adv
            assert got[s]["losses"][k] == pytest.approx(v, rel=tol, abs=1e-5), (s, k)
        for k, v in g["metrics"].items():
            tol = tol_adv if s == 0 else (0.02 if k.startswith("ssim") else 0.25)
            assert got[s]["metrics"][k] == pytest.approx(v, rel=tol, abs=2e-2 if s else 1e-5), (s, k)


@pytest.mark.parametrize("name,n_steps", [("v16x24x32_idt", 1), ("vnet_16x32x32", 1)])
def test_product_volume_step_matches_reference_golden_fp32(fp32_oracle_backend, name, n_steps):
    """CycleGAN with Resnet3D + PatchGAN3D (replicate-pad fold, 27/64/343-tap 3-D lowering, 8 parity classes) on the
    fp32 oracle backend against the golden losses of the real reference"""
    from .helpers import build_product_cyclegan3d, load_golden_volumes, run_product_volume_steps
    gold = load_golden_volumes()["steps"][name]
    c = gold["config"]
    model = build_product_cyclegan3d(c)
    got = run_product_volume_steps(model, c, n_steps)
    for s in range(n_steps):
        g = gold["steps"][s]
        assert set(got[s]["losses"]) == set(g["losses"])
        tol_adv, tol_cyc = (1e-4, 1e-4) if s == 0 else (0.10, 0.02)
        for k, v in g["losses"].items():
            tol = tol_cyc if k.startswith(("cycle", "idt")) else tol_adv
            assert got[s]["losses"][k] == pytest.approx(v, rel=tol, abs=1e-5), (s, k)
        for k, v in g["metrics"].items():
            assert got[s]["metrics"][k] == pytest.approx(v, rel=1e-4 if s == 0 else 0.25, abs=1e-5 if s == 0 else 2e-2)


def test_pix2pix_product_step_matches_reference_golden_fp32(fp32_oracle_backend):
    """Pix2PixConditionalGAN + Unet2D executor (skip concat by channel slices, dual-activation norm backward) on the
    fp32 oracle backend against the real reference's golden losses."""
    from .helpers import build_product_pix2pix, load_golden_pix2pix, run_product_pix2pix_steps
    gold = load_golden_pix2pix()["p2p_64x128"]
    c = gold["config"]
    got = run_product_pix2pix_steps(build_product_pix2pix(c), c, 3)
    for s in range(3):
        g = gold["steps"][s]
        assert got[s]["lrs"] == pytest.approx(g["lrs"], abs=1e-12)
        assert set(got[s]["losses"]) == set(g["losses"])
        tol = 1e-4 if s == 0 else 0.10          # see the CycleGAN test above for why later steps are an envelope
        for k, v in g["losses"].items():
            assert got[s]["losses"][k] == pytest.approx(v, rel=0.02 if (s and k == "pix2pix") else tol), (s, k)
        if s == 0:
            for k, v in g["metrics"].items():
                assert got[s]["metrics"][k] == pytest.approx(v, rel=1e-4, abs=1e-5), (s, k)


def test_cut_product_step_matches_reference_golden_fp32(fp32_oracle_backend):
    """CUT recipe: encoder-only partial passes with feature taps gathered from / scattered into the executor's NHWC
    activations, MLP + PatchNCE, D-then-G order — against the real reference's golden losses."""
    from .helpers import build_product_cut, load_golden_cut, run_product_cut_steps
    gold = load_golden_cut()["cut_64"]
    c = gold["config"]
    got = run_product_cut_steps(build_product_cut(c), c, 2)
    for s in range(2):
        g = gold["steps"][s]
        assert got[s]["lrs"] == pytest.approx(g["lrs"], abs=1e-12)
        assert set(got[s]["losses"]) == set(g["losses"])
        for k, v in g["losses"].items():
            tol = 2e-4 if s == 0 else (0.02 if k.startswith("NCE") else 0.10)
            assert got[s]["losses"][k] == pytest.approx(v, rel=tol), (s, k, got[s]["losses"][k], v)
