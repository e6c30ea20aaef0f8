"""Pix2Pix / U-Net on the HIP path against the fp32 oracle and the reference's golden
vectors. Tolerances as in tests/test_cyclegan_gpu.py (bf16 storage)."""
import pytest
import torch

from oracle import torch_ref

from .helpers import build_product_pix2pix, load_golden_pix2pix, run_product_pix2pix_steps
from .test_cyclegan_gpu import _net_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("D,ngf,hw", [(5, 16, (32, 64)), (7, 16, (128, 256))])
def test_unet2d_hip_vs_oracle(hip_ops, D, ngf, hw):
    from ganslate_amd.nn.generators import Unet2D
    # the 7-level net chains 14 convs with 12 norm+activation kinks between the loss and the first layer; its
    # 16-element first-layer bias gradient measured 0.32 relative L2 against fp32 (bf16 kink flips)
    _net_case(hip_ops, lambda: Unet2D(3, 3, D, "instance", ngf=ngf), torch_ref.Unet2D(3, 3, D, ngf), (1, 3, *hw), 51,
              grad_tol=0.30 if D == 5 else 0.45, grad_cos=0.95 if D == 5 else 0.90)


@pytest.mark.parametrize("name", ["p2p_64x128", "p2p_cfg3_shape"])
def test_pix2pix_step_matches_reference_golden(hip_ops, name):
    gold = load_golden_pix2pix()[name]
    c = gold["config"]
    n_steps = min(c["steps"], 4)
    got = run_product_pix2pix_steps(build_product_pix2pix(c), c, n_steps)
    for s in range(n_steps):
        g = gold["steps"][s]
        assert got[s]["lrs"] == pytest.approx(g["lrs"], abs=1e-12)
        from .envelope import step_tolerance      # iteration 0: 2e-2; later: the reference's own scatter (envelope.json)
        for k, v in g["losses"].items():
            assert got[s]["losses"][k] == pytest.approx(v, rel=step_tolerance(k, s, {"adv": 2e-2, "cycle": 2e-2})), (s, k)


def test_pix2pix_cfg3_full_width_runs(hip_ops):
    """BASELINE config 3: Unet2D(num_downs=7, ngf=128, dropout) + PatchGAN2D(n_layers=4, 6 ch) at 256x512, batch 1
    (167 M generator parameters): two steps, finite losses, dropout active."""
    c = dict(size=[256, 512], batch=1, steps=2, n_iters=100, n_iters_decay=100, num_downs=7, ngf=128,
             use_dropout=True, n_layers=4, lambda_pix2pix=30.0, seed=33)
    model = build_product_pix2pix(c)
    assert model.networks["G"].numel >= 167_000_000
    got = run_product_pix2pix_steps(model, c, 2)
    for s in got:
        for k, v in s["losses"].items():
            assert v == v and 0 < v < 1e3, (k, v)


def test_dropout_mask_changes_between_graph_replays(hip_ops):
    """nn.Dropout(0.5) of the pix2pix U-Net (unet2d.py:146-147) draws a new mask every forward. The
    captured step replays its launches with identical arguments, so the mask seed is read from device memory and
    refreshed by the recipe before every replay: with the weights frozen and the SAME batch every iteration, the
    generator output must still change from replay to replay (and must not when dropout is off)."""
    import random
    from .helpers import p2p_inputs
    outs = {}
    for use_dropout in (True, False):
        c = dict(size=[64, 128], batch=2, steps=5, n_iters=100, n_iters_decay=100, num_downs=6, ngf=16,
                 use_dropout=use_dropout, n_layers=3, lambda_pix2pix=30.0, seed=34)
        model = build_product_pix2pix(c, ("train.gan.optimizer.lr_G=0.0", "train.gan.optimizer.lr_D=0.0"))
        random.seed(7)
        A, B = p2p_inputs(c, 0)
        fakes = []
        for s in range(5):
            model.set_input({"A": A, "B": B})
            model.optimize_parameters()
            torch.cuda.synchronize()
            fakes.append(model.visuals["fake_B"].detach().float().cpu().clone())
        assert model._graph is not None, "iterations 3-5 must have been graph replays"
        outs[use_dropout] = fakes
    for i in range(5):
        for j in range(i + 1, 5):
            assert not torch.equal(outs[True][i], outs[True][j]), f"iterations {i} and {j} used the same dropout mask"
            assert torch.equal(outs[False][i], outs[False][j]), "without dropout the frozen generator is deterministic"
    # the masks drop half of the dropout layers' activations: outputs differ visibly, not by rounding
    assert (outs[True][3] - outs[True][4]).abs().mean().item() > 1e-3


def test_chunked_generator_update_under_backward(hip_ops, monkeypatch):
    """NativeAdam.arm_early: the generator's update runs layer group by layer group while its one backward pass of the step
    (pix2pix.py:84-88) goes on — between the pass's launches (default) and on the 'opt' stream beside them (GS_EARLY_ADAM=stream),
    launch by launch and as graph replays — against the update after the pass (GS_EARLY_ADAM=0): parameters, moments and packs
    bit for bit, losses equal."""
    from ganslate_amd.nn.optim import NativeAdam
    monkeypatch.setattr(NativeAdam, "EARLY_MIN", 1 << 14)
    c = dict(size=[64, 128], batch=2, steps=5, n_iters=100, n_iters_decay=100, num_downs=6, ngf=16, use_dropout=False,
             n_layers=3, lambda_pix2pix=30.0, seed=36)
    res = {}
    for mode in ("0", "1", "stream"):
        monkeypatch.setenv("GS_EARLY_ADAM", mode)
        model = build_product_pix2pix(c)
        opt, chunks = model.optimizers["G"], []
        inner = opt._update_range
        opt._update_range = lambda p, net, a, b, inner=inner, chunks=chunks: (chunks.append((a, b)), inner(p, net, a, b))[1]
        logs = run_product_pix2pix_steps(model, c, 5)
        torch.cuda.synchronize()
        assert model._graph is not None, "iterations 3-5 must have been graph replays"
        G = model.networks["G"]
        st = opt.state[G.master]
        G.refresh_packs(torch.zeros(2, 3, 64, 128))        # (what the next forward pass would read)
        torch.cuda.synchronize()
        pk = next(iter(G._packs.values()))
        res[mode] = dict(master=G.master.detach().clone(), m=st["exp_avg"].clone(), v=st["exp_avg_sq"].clone(),
                         fpack=pk["fpack"].clone(), dpack=pk["dpack"].clone(), logs=logs, chunks=chunks,
                         tr=set(getattr(G, "_tr_fresh", ())))
    assert not res["0"]["chunks"] and not res["0"]["tr"]
    assert len(res["1"]["tr"]) >= 6, "the fused launches of the inner levels wrote their transposed packs"
    for mode in ("1", "stream"):
        assert len(res[mode]["chunks"]) >= 3, "the early form must have handed chunks over"
        for k in ("master", "m", "v", "fpack", "dpack"):
            assert torch.equal(res[mode][k], res["0"][k]), (mode, k)
        assert res[mode]["logs"] == res["0"]["logs"], mode


def test_pix2pix_data_parallel_graph_step(hip_ops, monkeypatch):
    """the U-Net's backward must not issue its own bucketed all-reduce inside a captured
    data-parallel step (the runner reduces between its two graphs). 1-rank RCCL group, GS_FORCE_DDP."""
    import datetime
    import torch.distributed as dist
    from .test_ddp_graph_gpu import _free_port
    c = dict(size=[64, 128], batch=2, steps=4, n_iters=100, n_iters_decay=100, num_downs=5, ngf=16,
             use_dropout=False, n_layers=3, lambda_pix2pix=30.0, seed=35)
    frozen = ("train.gan.optimizer.lr_G=0.0", "train.gan.optimizer.lr_D=0.0")
    monkeypatch.setenv("LOCAL_RANK", "0")
    single = build_product_pix2pix(c, frozen)
    want = run_product_pix2pix_steps(single, c, 4)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1,
                            timeout=datetime.timedelta(minutes=2))
    try:
        monkeypatch.setenv("GS_FORCE_DDP", "1")
        ddp = build_product_pix2pix(c, frozen)
        got = run_product_pix2pix_steps(ddp, c, 4)
        torch.cuda.synchronize()
        assert ddp._graph is not None and ddp._graph_update is not None
        for s in range(4):
            for k, v in want[s]["losses"].items():
                assert got[s]["losses"][k] == pytest.approx(v, rel=1e-4, abs=1e-6), (s, k)
        for oa, ob in zip(single.optimizers.values(), ddp.optimizers.values()):
            for pa, pb in zip(oa.param_groups[0]["params"], ob.param_groups[0]["params"]):
                a, b = oa.state[pa]["exp_avg"], ob.state[pb]["exp_avg"]
                assert (a - b).norm().item() <= 1e-3 * a.norm().item()
    finally:
        dist.destroy_process_group()
