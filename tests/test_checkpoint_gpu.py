"""Resume on the GPU from a checkpoint in the REFERENCE's layout, and export one the reference's optimizer can read
(ganslate/nn/gans/base.py:226-287, engines/trainer.py:33-41).

The file is written the way the reference's `save_checkpoint` writes it — one `state_dict()` per network plus
`optimizer_G` / `optimizer_D` = `torch.optim.Adam.state_dict()` over per-layer parameters — by the fp32 oracle
(`oracle/torch_ref.CycleGANStep`, which trains with the same `torch.optim.Adam` class the reference uses). The HIP model
reads it through the config path (`train.checkpointing.load_iter`), which has to fill the flat fp32 master buffers, refresh
the bf16 weight packs the kernels read, and scatter the per-parameter moments and step counts into the flat Adam state."""
import random

import pytest
import torch

from oracle.torch_ref import CycleGANStep

from .helpers import CONF, golden_inputs, load_golden_steps

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    return ((a - b).norm() / (b.norm() + 1e-12)).item()


def _case():
    c = dict(load_golden_steps()["c64_default"]["config"])
    c.update(size=64, batch=2, pool_size=0)
    return c


def _oracle(c):
    return CycleGANStep(n_iters=c["n_iters"], n_iters_decay=c["n_iters_decay"], pool_size=0, seed=c["seed"])


def _product(c, out_dir, load_iter=None, seed=999):
    from ganslate_amd.utils.builders import build_conf, build_gan
    extra = [f"train.checkpointing.load_iter={load_iter}"] if load_iter is not None else []
    conf = build_conf([f"config={CONF}", f"train.batch_size={c['batch']}", f"train.n_iters={c['n_iters']}",
                       f"train.n_iters_decay={c['n_iters_decay']}", "train.gan.pool_size=0",
                       f"train.output_dir={out_dir}", *extra])
    torch.manual_seed(seed)          # NOT the oracle's weights: whatever the test finds in the networks came from the file
    return build_gan(conf)


def test_hip_model_resumes_from_a_reference_layout_checkpoint_and_writes_one_back(hip_ops, tmp_path):
    c = _case()
    ref = _oracle(c)
    random.seed(c["seed"])
    for s in range(3):
        ref.step(*golden_inputs(c, s))
        ref.update_learning_rate()
    ck = {name: {k: v.clone() for k, v in net.state_dict().items()} for name, net in ref.nets.items()}
    ck["optimizer_G"], ck["optimizer_D"] = ref.opt_G.state_dict(), ref.opt_D.state_dict()
    (tmp_path / "checkpoints").mkdir()
    torch.save(ck, tmp_path / "checkpoints" / "3.pth")

    model = _product(c, tmp_path, load_iter=3)
    for _ in range(3):                      # the trainer restarts its iteration counter at load_iter (trainer.py:33-41)
        model.update_learning_rate()

    # (1) the kernels see the loaded weights: a forward pass through the bf16 packs against the oracle's network
    x = golden_inputs(c, 7)[0]
    with torch.no_grad():
        y_ref = ref.nets["G_AB"](x)
        y_hip = model.networks["G_AB"](x.to(hip_ops.device)).float().cpu()
    assert rel_l2(y_hip, y_ref) <= 3e-2, rel_l2(y_hip, y_ref)

    # (2) iteration 4 on both: losses, and the UPDATE each tensor received — with the loaded first / second moments and
    # step count 4 in the bias corrections. A fresh optimizer state would step every weight by +-lr (|update| = lr sqrt(n)).
    before = {name: {k: v.clone() for k, v in net.state_dict().items()} for name, net in ref.nets.items()}
    A, B = golden_inputs(c, 3)
    want, _ = ref.step(A, B)
    model.set_input({"A": A, "B": B})
    model.optimize_parameters()
    torch.cuda.synchronize()
    for k, v in want.items():
        assert float(model.losses[k].detach()) == pytest.approx(v, rel=2e-2), (k, float(model.losses[k].detach()), v)
    checked = 0
    for name, net in ref.nets.items():
        mine = {k: v.float().cpu() for k, v in model.networks[name].state_dict().items()}
        noise = {f"{p}.bias" for nd in model.networks[name].nodes if nd.norm for p in (nd.name,) + tuple(nd.aliases)}
        for k, v in net.state_dict().items():
            if k not in mine or k in noise or v.numel() < 1000:
                continue
            du_ref, du_hip = v - before[name][k], mine[k] - before[name][k]
            ratio = (du_hip.norm() / du_ref.norm()).item()
            cos = (du_hip.flatten() @ du_ref.flatten() / (du_hip.norm() * du_ref.norm())).item()
            fresh_norm = 2e-4 * v.numel() ** 0.5          # what Adam's first step from zero moments would be
            assert du_ref.norm().item() < 0.8 * fresh_norm, (name, k)         # the case can tell the two apart
            assert 0.9 <= ratio <= 1.1 and cos >= 0.9, (name, k, ratio, cos)
            checked += 1
    assert checked >= 40

    # (3) export: the product's checkpoint has the reference's keys, and a fresh reference-side optimizer resumes from it
    model.save_checkpoint(4)
    out = torch.load(tmp_path / "checkpoints" / "4.pth", map_location="cpu", weights_only=False)
    assert set(out) == {"G_AB", "G_BA", "D_B", "D_A", "optimizer_G", "optimizer_D"}
    fresh = _oracle(c)
    for name, net in fresh.nets.items():
        net.load_state_dict({k: v for k, v in out[name].items()}, strict=True)
    fresh.opt_G.load_state_dict(out["optimizer_G"])
    fresh.opt_D.load_state_dict(out["optimizer_D"])
    assert float(fresh.opt_G.state_dict()["state"][0]["step"]) == 4.0
    for _ in range(4):
        fresh.update_learning_rate()
    ref.update_learning_rate()
    A, B = golden_inputs(c, 4)
    cont, _ = ref.step(A, B)                 # the oracle continuing its own run
    resumed, _ = fresh.step(A, B)            # the oracle resumed from the product's file
    for k, v in cont.items():
        assert resumed[k] == pytest.approx(v, rel=2e-2), (k, resumed[k], v)


def test_resumed_run_equals_the_uninterrupted_one_on_the_gpu(hip_ops, tmp_path):
    """save at iteration 2, resume in a fresh model, iteration 3 equals the uninterrupted run's (deterministic kernels:
    to 1e-5; the file holds the fp32 masters and the fp32 Adam state, nothing is rounded on the way)"""
    c = _case()
    a = _product(c, tmp_path, seed=5)
    random.seed(1)
    for s in range(2):
        A, B = golden_inputs(c, s)
        a.set_input({"A": A, "B": B})
        a.optimize_parameters()
        a.update_learning_rate()
    a.save_checkpoint(2)
    b = _product(c, tmp_path, load_iter=2, seed=6)
    for _ in range(2):
        b.update_learning_rate()
    A, B = golden_inputs(c, 2)
    for m in (a, b):
        m.set_input({"A": A, "B": B})
        m.optimize_parameters()
    torch.cuda.synchronize()
    for k, v in a.losses.items():
        if v is not None:
            assert float(b.losses[k].detach()) == pytest.approx(float(v.detach()), rel=1e-5), k
    for name in a.networks:
        sa, sb = a.networks[name].state_dict(), b.networks[name].state_dict()
        for k in sa:
            assert torch.equal(sa[k].cpu(), sb[k].cpu()) or (sa[k].float().cpu() - sb[k].float().cpu()).abs().max().item() <= 1e-6, (name, k)
