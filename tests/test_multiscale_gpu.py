"""MultiScalePatchGAN3D on the GPU: per-scale maps, input gradient and parameter gradients of the HIP
executors against the fp32 oracle twin (which tests/test_multiscale_cpu.py pins to the real reference class) for the same
window draws — bf16 storage tolerances of tests/test_cyclegan_gpu.py — and a Trainer run with multi-scale discriminators."""
import random

import pytest
import torch

from oracle import torch_ref

from .test_cyclegan_gpu import cosine, rel_l2

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cin,ndf,n_layers,scales,shape,seed", [(1, 64, 3, 2, (1, 1, 64, 64, 64), 81),
                                                                (2, 16, 2, 3, (2, 2, 48, 36, 60), 82)])
def test_multiscale_hip_vs_oracle(hip_ops, cin, ndf, n_layers, scales, shape, seed):
    from ganslate_amd.nn.discriminators import MultiScalePatchGAN3D
    twin = torch_ref.MultiScalePatchGAN3D(cin, ndf, n_layers, 4, scales)
    sd = torch_ref.seeded_state_dict(twin, seed)
    twin.load_state_dict(sd)
    net = MultiScalePatchGAN3D(cin, ndf, n_layers, (4, 4, 4), scales, "instance")
    net.load_state_dict(sd)
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(shape, generator=g) * 2 - 1
    xa, xh = x.clone().requires_grad_(), x.clone().to(hip_ops.device).requires_grad_()
    random.seed(seed)
    ma = twin(xa)
    random.seed(seed)
    mh = net(xh)
    assert list(mh) == [str(s) for s in range(1, scales + 1)]
    gys = {s: torch.randn(m.shape, generator=g) for s, m in ma.items()}
    torch.autograd.backward(list(ma.values()), [gys[s] for s in ma])
    torch.autograd.backward(list(mh.values()), [gys[s].to(hip_ops.device) for s in mh])
    torch.cuda.synchronize()
    for s in ma:
        assert mh[s].shape == ma[s].shape
        assert rel_l2(mh[s].detach().cpu(), ma[s].detach()) <= 3e-2, s
    gx = xh.grad.cpu()
    assert rel_l2(gx, xa.grad) <= 0.15 and cosine(gx, xa.grad) >= 0.99, (rel_l2(gx, xa.grad), cosine(gx, xa.grad))
    grads = {k: v.cpu() for k, v in net.grads_state_dict().items()}
    normed = {f"model.{s}.{nd.name}" for s, sub in net.model.items() for nd in sub.nodes if nd.norm}
    for k, p in twin.named_parameters():
        if k.endswith(".bias") and k[:-5] in normed:
            continue
        assert rel_l2(grads[k], p.grad) <= 0.15 and cosine(grads[k], p.grad) >= 0.99, (k, rel_l2(grads[k], p.grad))


def test_trainer_with_multiscale_discriminators(hip_ops, tmp_path):
    """CycleGAN(Resnet3D + MultiScalePatchGAN3D) through init_engine on the GPU: the step runs launch by launch (host-drawn
    windows), losses finite and falling into the reference's keys, both scales of both discriminators updated"""
    from ganslate_amd.engines import init_engine
    args = ["config=tests/configs/cyclegan3d_synthetic.yaml", f"train.output_dir={tmp_path}", "train.n_iters=3",
            "train.n_iters_decay=0", "train.dataset.final_size=[32,32,32]", "train.gan.generator.n_residual_blocks=2",
            "train.gan.discriminator._target_=ganslate.nn.discriminators.MultiScalePatchGAN3D",
            "train.gan.discriminator.n_layers=2", "train.gan.discriminator.scales=2", "train.seed=4"]
    trainer = init_engine("train", args)
    model = trainer.model
    before = {n: {k: v.clone() for k, v in model.networks[n].state_dict().items()} for n in ("D_A", "D_B")}
    trainer.run()
    assert model.step_graph_enabled is False and model._graph is None
    assert sorted(k for k, v in model.losses.items() if v is not None) == ["D_A", "D_B", "G_AB", "G_BA", "cycle_A", "cycle_B"]
    assert all(float(v) == float(v) for v in model.losses.values() if v is not None)
    assert model.metrics["D_A_real"] is None
    for n in ("D_A", "D_B"):
        after = model.networks[n].state_dict()
        for s in ("1", "2"):
            k = f"model.{s}.model.0.weight"
            assert not torch.equal(after[k], before[n][k]), (n, k)
