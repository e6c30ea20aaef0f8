"""The captured training step (BaseGAN._graph_step): a CycleGAN whose iterations run as hipGraph replays must follow the
same trajectory as one that enqueues every launch from the host — same losses, same weights, same image-pool contents —
with the pools' coin flips and Adam's step-dependent scalars living in device memory. Plus the two kernels that made the
step capturable (gs_pool_query, gs_adam_step_dev) against the op-level oracle."""
import random

import pytest
import torch

from oracle.ops_ref import RefOps
from tests.helpers import build_product_cyclegan, golden_inputs, load_golden_steps

pytestmark = pytest.mark.gpu


def _run(model, c, n_steps):
    random.seed(c["seed"])
    out = []
    for s in range(n_steps):
        A, B = golden_inputs(c, s)
        model.set_input({"A": A, "B": B})
        model.optimize_parameters()
        _, losses, visuals, metrics = model.get_loggable_data()
        torch.cuda.synchronize()
        out.append(({k: float(v.detach()) for k, v in losses.items() if v is not None},
                    {k: float(v) for k, v in metrics.items() if v is not None},
                    visuals["fake_B"].detach().float().cpu().clone()))
        model.update_learning_rate()
    return out


def _pair(c, extra, n_steps):
    eager = build_product_cyclegan(c, extra)
    eager.step_graph_enabled = False
    want = _run(eager, c, n_steps)
    graphed = build_product_cyclegan(c, extra)
    assert graphed.step_graph_enabled
    got = _run(graphed, c, n_steps)
    assert graphed._graph is not None, "the step was never captured"
    return eager, graphed, want, got


@pytest.mark.parametrize("pool_size,extra", [(3, ()), (50, ("train.gan.optimizer.lambda_identity=0.5",
                                                           "train.gan.optimizer.proportion_ssim=0.84"))])
def test_graph_replays_reproduce_eager_steps_with_frozen_weights(hip_ops, pool_size, extra):
    """learning rate 0: every iteration's losses depend only on its inputs and the image pools' state, so replayed and
    launch-by-launch iterations must agree tightly (no chaotic amplification of the split-K atomics' rounding order)"""
    c = dict(load_golden_steps()["c64_default"]["config"])
    c["pool_size"] = pool_size          # 3: the pool is full after two iterations, swaps and same-slot draws follow
    n_steps = 8
    frozen = ("train.gan.optimizer.lr_G=0.0", "train.gan.optimizer.lr_D=0.0")
    eager, graphed, want, got = _pair(c, tuple(extra) + frozen, n_steps)
    for s in range(n_steps):
        for k, v in want[s][0].items():
            assert got[s][0][k] == pytest.approx(v, rel=1e-4, abs=1e-6), (s, k)
        for k, v in want[s][1].items():
            assert got[s][1][k] == pytest.approx(v, rel=1e-4, abs=1e-6), (s, k)
        assert torch.equal(got[s][2], want[s][2]), s          # forward kernels are deterministic
    for pa, pb in ((eager.fake_A_pool, graphed.fake_A_pool), (eager.fake_B_pool, graphed.fake_B_pool)):
        assert pa.num_imgs == pb.num_imgs and torch.equal(pa.images, pb.images)


def test_training_is_bitwise_reproducible_and_graph_replays_equal_eager_steps(hip_ops):
    """With the weights moving. Every accumulation of the step is order-fixed (weight / bias gradients: partial slabs +
    fixed-order second stage, gs_wgrad_ws / gs_bias_grad_ws; statistics and norm-backward sums: one slot per tile; loss
    reductions: last-arriver in fixed order), so (i) two launch-by-launch runs are bit-identical — round 1 had fp32
    atomics in the weight gradients and two runs drifted apart from the third iteration on — and (ii) a run whose
    iterations are hipGraph replays follows the launch-by-launch run bit for bit: same kernels, same arguments."""
    c = dict(load_golden_steps()["c64_default"]["config"])
    c["pool_size"] = 3
    n_steps = 8
    eager, graphed, want, got = _pair(c, (), n_steps)
    again = build_product_cyclegan(c, ())
    again.step_graph_enabled = False
    rerun = _run(again, c, n_steps)
    for s in range(n_steps):
        assert torch.equal(rerun[s][2], want[s][2]), f"two launch-by-launch runs differ at iteration {s}"
        assert rerun[s][0] == want[s][0], s
        assert torch.equal(got[s][2], want[s][2]), f"graph replay differs from the launch-by-launch run at iteration {s}"
        assert got[s][0] == want[s][0], s
    for name in eager.networks:
        a, b = eager.networks[name].master.detach(), graphed.networks[name].master.detach()
        assert torch.equal(a, b) and torch.equal(a, again.networks[name].master.detach()), name
    for oa, ob in zip(eager.optimizers.values(), graphed.optimizers.values()):
        assert [st["step"] for st in oa.state.values()] == [st["step"] for st in ob.state.values()] == \
            [n_steps] * len(oa.state)
        for group in ob.param_groups:          # the device scalars the replayed update reads are those of step n
            for p in group["params"]:
                want_h = torch.tensor([group["lr"], 0.5, 0.999, 1e-8, 1 - 0.5 ** n_steps,
                                       (1 - 0.999 ** n_steps) ** 0.5], dtype=torch.float64).float()
                assert torch.equal(ob.state[p]["hyper"].cpu(), want_h)


@pytest.mark.parametrize("extra", [(), ("train.gan.optimizer.lambda_identity=0.5",
                                       "train.gan.optimizer.proportion_ssim=0.84")])
def test_concurrent_streams_accumulate_the_same_gradients(hip_ops, monkeypatch, extra):
    """One stream, launch by launch (the reference's sequential step) against the default (second cycle and discriminator
    update on their own streams, replayed as a graph): with the weights frozen, Adam's moments after a few iterations
    are running sums of every pass' parameter gradients — a lost or torn accumulation between the streams would show
    orders of magnitude above the atomics' rounding noise."""
    c = dict(load_golden_steps()["c64_default"]["config"])
    c["pool_size"] = 3
    frozen = ("train.gan.optimizer.lr_G=0.0", "train.gan.optimizer.lr_D=0.0")
    monkeypatch.setenv("GS_SIDE_STREAM", "0")
    serial = build_product_cyclegan(c, tuple(extra) + frozen)
    serial.step_graph_enabled = False
    want = _run(serial, c, 4)
    monkeypatch.setenv("GS_SIDE_STREAM", "1")
    conc = build_product_cyclegan(c, tuple(extra) + frozen)
    got = _run(conc, c, 4)
    assert conc._graph is not None and conc._side["D"]["stream"] != conc._side["cycle_B"]["stream"]
    for s in range(4):
        for k, v in want[s][0].items():
            assert got[s][0][k] == pytest.approx(v, rel=1e-4, abs=1e-6), (s, k)
    for oa, ob in zip(serial.optimizers.values(), conc.optimizers.values()):
        for pa, pb in zip(oa.param_groups[0]["params"], ob.param_groups[0]["params"]):
            for key in ("exp_avg", "exp_avg_sq"):
                a, b = oa.state[pa][key], ob.state[pb][key]
                assert a.abs().max().item() > 0
                assert (a - b).norm().item() <= 1e-3 * a.norm().item(), key


def test_graph_falls_back_for_another_batch_size(hip_ops):
    """a ragged last batch runs launch by launch and the next full batch replays again"""
    c = dict(load_golden_steps()["c64_default"]["config"])
    model = build_product_cyclegan(c)
    random.seed(1)
    for s, batch in enumerate([2, 2, 2, 1, 2]):
        A, B = golden_inputs(c, s)
        model.set_input({"A": A[:batch], "B": B[:batch]})
        model.optimize_parameters()
        torch.cuda.synchronize()
        assert model.visuals["fake_B"].shape[0] == batch
        assert all(float(v.detach()) == float(v.detach()) for v in model.losses.values() if v is not None)
    assert model._graph is not None
    assert [st["step"] for st in model.optimizers["G"].state.values()] == [5, 5]


def test_pool_query_kernel(hip_ops):
    g = torch.Generator().manual_seed(0)
    B, slots = 5, 4
    pool0 = torch.rand((slots, 3, 8, 8), generator=g)
    imgs = torch.rand((B, 3, 8, 8), generator=g)
    code = torch.tensor([1, -1, 2 | 0x40000000, 2 | 0x40000000, 0 | 0x40000000], dtype=torch.int32)
    outs = []
    for ops, dev in ((RefOps(), "cpu"), (hip_ops, hip_ops.device)):
        pool, x, out = pool0.clone().to(dev), imgs.to(dev), torch.empty_like(imgs).to(dev)
        ops.pool_query(pool, x, out, code.to(dev))
        outs.append((pool.cpu(), out.cpu()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert torch.equal(outs[1][1][3], imgs[2])       # the second draw of slot 2 returns the image the first one stored


def test_adam_dev_matches_host_scalars(hip_ops):
    g = torch.Generator().manual_seed(1)
    n = 50_001
    p0, g0 = torch.randn(n, generator=g), torch.randn(n, generator=g) * 1e-2
    dev = hip_ops.device
    res = []
    for variant in ("host", "dev", "ref"):
        ops = RefOps() if variant == "ref" else hip_ops
        d = "cpu" if variant == "ref" else dev
        p, gr, m, v = p0.clone().to(d), g0.clone().to(d), torch.zeros(n, device=d), torch.zeros(n, device=d)
        for t in (1, 2, 3):
            gr.copy_(g0.to(d) * t)
            if variant == "host":
                ops.adam_step(p, gr, m, v, 2e-4, 0.5, 0.999, 1e-8, t, grad_scale=0.5)
            else:
                hyper = torch.tensor([2e-4, 0.5, 0.999, 1e-8, 1 - 0.5 ** t, (1 - 0.999 ** t) ** 0.5],
                                     dtype=torch.float64).float().to(d)
                ops.adam_step_dev(p, gr, m, v, hyper, grad_scale=0.5)
        res.append((p.cpu(), m.cpu(), v.cpu(), gr.cpu()))
    assert all(torch.equal(a, b) for a, b in zip(res[0], res[1])), "device-scalar Adam must be bit-identical"
    assert (res[1][0] - res[2][0]).abs().max().item() <= 1e-6 * res[2][0].abs().max().item()
    assert res[1][3].abs().max().item() == 0.0


def test_gradient_accumulation_between_replays(hip_ops):
    """`dw_fresh` (gs_wgrad_desc: the first weight gradient of a layer since the optimiser cleared the buffer STORES instead of
    adding) is a contract carried by host bookkeeping (NativeNet.wgrad_fresh) and baked into captured graphs. Between two replays
    a user recipe may accumulate: two eager backward passes WITHOUT an optimiser step in between must leave exactly g + g in
    the flat gradients (a stale `fresh = True` would leave g), and the replays around them must be unaffected."""
    c = dict(load_golden_steps()["c64_default"]["config"])
    c["pool_size"] = 0
    frozen = ("train.gan.optimizer.lr_G=0.0", "train.gan.optimizer.lr_D=0.0")
    model = build_product_cyclegan(c, frozen)
    before = _run(model, c, 4)                      # iterations 2.. are replays
    assert model._graph is not None
    A, B = golden_inputs(c, 0)
    Ds = [model.networks["D_B"], model.networks["D_A"]]

    def eager_backward_G(times):
        model._eager_set_input({"A": A, "B": B})
        model.set_requires_grad(Ds, False)
        model.optimizers["G"].zero_grad(set_to_none=True)
        for _ in range(times):
            model.forward()
            model.backward_G()
        for net in (model.networks["G_AB"], model.networks["G_BA"]):
            net.flush_deferred_wgrads()
        torch.cuda.synchronize()
        return [model.networks[n].master.grad.clone() for n in ("G_AB", "G_BA")]
    g1 = eager_backward_G(1)
    g2 = eager_backward_G(2)
    for a, b in zip(g1, g2):
        assert a.abs().max().item() > 0.0
        # g + g up to the summation order (two passes' operands may share one merged weight-gradient launch); a stale
        # `fresh` would leave g, i.e. an error of |g|
        scale = a.abs().max().item()
        assert (b - 2 * a).abs().max().item() <= 1e-4 * scale, ((b - 2 * a).abs().max().item(), scale)
    model.set_requires_grad(Ds, True)
    after = _run(model, c, 4)                       # replays again (weights frozen: the same losses as before, step for step)
    for s in range(2, 4):
        for k, v in before[s][0].items():
            assert after[s][0][k] == pytest.approx(v, rel=1e-5, abs=1e-7), (s, k)
