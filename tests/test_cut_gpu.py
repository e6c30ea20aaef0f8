"""CUT on the HIP path against the reference's golden vectors; tolerances as in
tests/test_cyclegan_gpu.py (bf16 storage). Patch ids are drawn like the reference's (CPU torch.randperm)."""
import pytest
import torch

from .helpers import build_product_cut, load_golden_cut, run_product_cut_steps

pytestmark = pytest.mark.gpu


def test_cut_step_matches_reference_golden(hip_ops):
    gold = load_golden_cut()["cut_64"]
    c = gold["config"]
    got = run_product_cut_steps(build_product_cut(c), c, c["steps"])
    for s in range(c["steps"]):
        g = gold["steps"][s]
        assert got[s]["lrs"] == pytest.approx(g["lrs"], abs=1e-12)
        for k, v in g["losses"].items():
            from .envelope import step_tolerance  # iteration 0: 2e-2; later: the reference's own scatter (envelope.json)
            tol = step_tolerance(k, s, {"adv": 2e-2, "cycle": 2e-2})
            assert got[s]["losses"][k] == pytest.approx(v, rel=tol), (s, k, got[s]["losses"][k], v)


def test_cut_feature_taps_hip_vs_oracle_backend(hip_ops):
    """encoder-only partial pass: sampled features and the gradients they send into the encoder and the input"""
    from ganslate_amd.nn.generators import Resnet2D
    from ganslate_amd.nn.native import backend
    from oracle import torch_ref
    from oracle.ops_ref import RefOps
    sd = torch_ref.seeded_state_dict(torch_ref.Resnet2D(3, 3, 9), 61)
    g = torch.Generator().manual_seed(61)
    x = torch.rand(2, 3, 64, 64, generator=g) * 2 - 1
    layers = [0, 4, 8, 12, 16]
    res = {}
    for name, ops in (("hip", hip_ops), ("cpu", RefOps(act_dtype=torch.bfloat16))):
        backend.set_ops(ops)
        try:
            net = Resnet2D(3, 3, "instance", 9)
            net.load_state_dict(sd)
            gg = torch.Generator().manual_seed(62)
            ids = [torch.randperm(net.tap_extent(e, 64, 64), generator=gg)[:64].to(ops.device) for e in layers]
            xi = x.clone().to(ops.device).requires_grad_()
            feats = net.extract_patch_features(xi, layers, ids)
            w = [torch.randn(f.shape, generator=gg).to(ops.device) for f in feats]
            sum((f * ww).sum() for f, ww in zip(feats, w)).backward()
            res[name] = ([f.detach().cpu() for f in feats], xi.grad.cpu(), net.master.grad.cpu().clone())
        finally:
            backend.set_ops(hip_ops)
    rel = lambda a, b: ((a - b).norm() / (b.norm() + 1e-12)).item()
    for fh, fc in zip(res["hip"][0], res["cpu"][0]):
        assert rel(fh, fc) <= 2e-2
    assert rel(res["hip"][1], res["cpu"][1]) <= 0.30
    assert rel(res["hip"][2], res["cpu"][2]) <= 0.30


def test_cut_step_at_headline_shape_matches_reference_golden(hip_ops):
    """BASELINE configs[3] shape: CUT at 256x256 (batch 2 per rank) against two iterations of the real reference
    (tests/golden/fullsize.json, oracle/gen_golden_r2.py)"""
    import json
    from .envelope import step_tolerance
    from .helpers import GOLD
    gold = json.loads((GOLD / "fullsize.json").read_text())["cut_256_b2"]
    c = gold["config"]
    got = run_product_cut_steps(build_product_cut(c), c, c["steps"])
    for s in range(c["steps"]):
        for k, v in gold["steps"][s]["losses"].items():
            tol = step_tolerance(k, s, {"adv": 2e-2, "cycle": 2e-2})
            assert got[s]["losses"][k] == pytest.approx(v, rel=tol), (s, k, got[s]["losses"][k], v)


def test_flip_equivariance_coin_is_data_of_the_captured_step(hip_ops, monkeypatch):
    """FastCUT's `use_equivariance_flip` (cut.py:146-152, 213-215): the coin is drawn on the host per iteration, the flip of the
    inputs reads it from device memory (gs_flip_w_if) and the mirrored target ids are prepared with the other host state,
    so the step is captured — and must equal the launch-by-launch run, iteration by iteration, for a coin sequence that has
    both outcomes."""
    import numpy as np
    x = torch.randn(2, 3, 5, 7, device=hip_ops.device)
    for f in (0, 1):
        flag = torch.tensor([f], dtype=torch.int32, device=hip_ops.device)
        assert torch.equal(hip_ops.flip_w_if(x, flag), x.flip(-1) if f else x)
    c = load_golden_cut()["cut_64"]["config"]
    runs = {}
    for graph in ("1", "0"):
        monkeypatch.setenv("GS_STEP_GRAPH", graph)
        np.random.seed(7)
        torch.manual_seed(c["seed"])
        model = build_product_cut(c, extra=("train.gan.use_equivariance_flip=true",))
        assert model.use_equivariance_flip and model.graph_capturable
        got, flips = run_product_cut_steps(model, c, 5), []
        runs[graph] = got
        assert (model._graph is not None) == (graph == "1")
    coins = np.random.RandomState(7).random_sample(5) > 0.5
    assert coins.any() and not coins.all(), "pick a seed with both outcomes"
    for s, (a, b) in enumerate(zip(runs["1"], runs["0"])):
        for k in b:
            assert a[k] == pytest.approx(b[k], rel=2e-3, abs=1e-5), (s, k)
