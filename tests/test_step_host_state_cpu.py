"""Host side of a captured step on the CPU (oracle backend): the pieces BaseGAN's graph runner moves out of
optimize_parameters — optimiser preparation (step counters, device scalar vectors), deferred optimiser launches (the
data-parallel two-graph scheme) and the image pools' pre-drawn coin flips — must leave the same state behind as the plain
step does, and the optimiser's state dict must stay free of the derived device scalars."""
import random

import pytest
import torch

from ganslate_amd.nn.native import backend
from oracle.ops_ref import RefOps

from .helpers import CONF, golden_inputs, load_golden_steps


def build_product_cyclegan(c):
    """the recipe of golden case `c` with 2-block generators and seeded random weights: these tests compare two executions of
    the SAME model with each other (plain vs externally prepared / deferred), not against the reference"""
    from ganslate_amd.utils.builders import build_conf, build_gan
    conf = build_conf([f"config={CONF}", f"train.batch_size={c['batch']}", f"train.n_iters={c['n_iters']}",
                       f"train.n_iters_decay={c['n_iters_decay']}", f"train.gan.pool_size={c['pool_size']}",
                       f"train.gan.optimizer.lambda_identity={c['lambda_identity']}",
                       f"train.gan.optimizer.proportion_ssim={c['proportion_ssim']}",
                       "train.gan.generator.n_residual_blocks=2"])
    torch.manual_seed(c["seed"])
    model = build_gan(conf)
    random.seed(c["seed"])
    return model


@pytest.fixture()
def fp32_oracle_backend():
    backend.set_ops(RefOps(act_dtype=torch.float32))
    yield
    backend.set_ops(None)


def _steps(model, c, n, externally_prepared=False, deferred=False):
    random.seed(c["seed"])
    out = []
    for s in range(n):
        A, B = golden_inputs(c, s)
        model.set_input({"A": A, "B": B})
        if externally_prepared:          # what BaseGAN._prepare_host_state + _capture_step / _replay do around a graph
            model._set_external_host_state(True)
            model._prepare_host_state()
        pending = [] if deferred else None
        for optim in model.optimizers.values():
            optim.deferred_to = pending
        model.optimize_parameters()
        if deferred:
            assert [type(o).__name__ for o in pending] == ["NativeAdam", "NativeAdam"]
            for optim in pending:        # nothing read the updated weights in between: same arithmetic
                optim.launch()
        for optim in model.optimizers.values():
            optim.deferred_to = None
        out.append({k: float(v.detach()) for k, v in model.losses.items() if v is not None})
        model.update_learning_rate()
    return out


@pytest.mark.parametrize("mode", ["prepared", "prepared+deferred"])
def test_externally_prepared_and_deferred_steps_match_the_plain_step(fp32_oracle_backend, mode):
    c = dict(load_golden_steps()["c64_default"]["config"])
    c["pool_size"] = 3                      # swaps start in the second iteration
    plain = build_product_cyclegan(c)
    assert not plain.step_graph_enabled     # no GPU, no graph: the recipe runs as written
    want = _steps(plain, c, 3)
    other = build_product_cyclegan(c)
    got = _steps(other, c, 3, externally_prepared=True, deferred="deferred" in mode)
    for s in range(3):
        for k, v in want[s].items():
            assert got[s][k] == pytest.approx(v, rel=1e-6, abs=1e-7), (s, k)
    for name in plain.networks:
        assert torch.equal(plain.networks[name].master.detach(), other.networks[name].master.detach()), name
    for pa, pb in ((plain.fake_A_pool, other.fake_A_pool), (plain.fake_B_pool, other.fake_B_pool)):
        assert pa.num_imgs == pb.num_imgs and torch.equal(pa.images, pb.images)


def test_optimizer_state_dict_round_trip_keeps_buffers(fp32_oracle_backend):
    c = dict(load_golden_steps()["c64_default"]["config"])
    model = build_product_cyclegan(c)
    _steps(model, c, 2)
    opt = model.optimizers["G"]
    sd = opt.state_dict()
    assert all("hyper" not in st for st in sd["state"].values())
    assert all("hyper" in st for st in opt.state.values())          # the live state still has its device scalars
    before = {id(p): (st["exp_avg"], st["exp_avg_sq"], st["hyper"]) for p, st in opt.state.items()}
    saved = {k: {kk: (vv.clone() if torch.is_tensor(vv) else vv) for kk, vv in st.items()} for k, st in sd["state"].items()}
    _steps(model, c, 1)                                               # moments move on ...
    opt.load_state_dict({"state": saved, "param_groups": sd["param_groups"]})
    for p, st in opt.state.items():                                   # ... and come back INTO the same buffers
        m, v, h = before[id(p)]
        assert st["exp_avg"] is m and st["exp_avg_sq"] is v and st["hyper"] is h
        assert st["step"] == 2
    for k, st in saved.items():
        p = opt.param_groups[0]["params"][k]
        assert torch.equal(opt.state[p]["exp_avg"], st["exp_avg"])
