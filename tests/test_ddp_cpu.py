"""Data-parallel path on CPU: 2 processes, gloo, oracle fp32 backend. Checks the identity the design relies on
(SURVEY.md §8e): InstanceNorm is per-sample and every loss is a batch mean, so the all-reduced-and-averaged
gradient of 2 ranks x batch 1 equals the single-process gradient on the concatenated batch of 2; and that the
bucketed asynchronous all-reduce fires during the last backward pass and leaves all ranks with identical weights
after the optimiser step."""
import os
import sys
from pathlib import Path

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    """a port nobody listens on right now (fixed offsets collided with sockets lingering from an earlier test)"""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]
OVERRIDES = ["train.gan.generator.n_residual_blocks=2", "train.dataset.final_size=[32,32]", "train.gan.pool_size=0",
             "train.metrics.ssim=False", "train.metrics.discriminator_evolution=False"]


def _inputs(n):
    g = torch.Generator().manual_seed(77)
    return torch.rand(n, 3, 32, 32, generator=g) * 2 - 1, torch.rand(n, 3, 32, 32, generator=g) * 2 - 1


def _build(batch, seed=5):
    from ganslate_amd.nn.native import backend
    from ganslate_amd.utils.builders import build_conf, build_gan
    from oracle.ops_ref import RefOps
    backend.set_ops(RefOps(act_dtype=torch.float32))
    conf = build_conf([f"config={ROOT / 'tests/configs/cyclegan_synthetic.yaml'}", f"train.batch_size={batch}",
                       *OVERRIDES])
    torch.manual_seed(seed)
    return build_gan(conf)


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank),
                      LOCAL_RANK=str(rank), GANSLATE_DIST_BACKEND="gloo")
    torch.set_num_threads(2)
    from ganslate_amd.utils import communication
    communication.init_distributed()
    model = _build(1, seed=5 + rank)       # different init per rank: parallelize() must broadcast rank 0's weights
    A, B = _inputs(world)
    model.set_input({"A": A[rank:rank + 1], "B": B[rank:rank + 1]})
    # gradient of the generator objective, reduced over ranks
    model.forward()
    model.set_requires_grad([model.networks["D_B"], model.networks["D_A"]], False)
    model.backward_G()
    grads = {}
    for name in ("G_AB", "G_BA"):
        net = model.networks[name]
        fired_async = len(net._reduce_handles) > 0
        scale = net.finish_grad_reduction()
        grads[name] = (net.master.grad.clone() * scale, fired_async)
    # then a full step: weights must stay identical across ranks
    model.optimizers["G"].zero_grad()
    model.optimize_parameters()
    weights = {n: net.master.detach().clone() for n, net in model.networks.items()}
    torch.save({"grads": grads, "weights": weights}, Path(out_dir) / f"rank{rank}.pt")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_gradient_equals_single_process_batch_two(tmp_path):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = (torch.load(tmp_path / f"rank{r}.pt") for r in range(world))
    for name in r0["weights"]:
        assert torch.equal(r0["weights"][name], r1["weights"][name]), f"{name}: ranks diverged after a step"
    for name in ("G_AB", "G_BA"):
        assert r0["grads"][name][1], "bucketed all-reduce did not start during the last backward pass"
        assert torch.equal(r0["grads"][name][0], r1["grads"][name][0])
    # single-process reference on the concatenated batch
    os.environ.pop("WORLD_SIZE", None)
    from ganslate_amd.nn.native import backend
    single = _build(2, seed=5)             # rank 0's init
    A, B = _inputs(2)
    single.set_input({"A": A, "B": B})
    single.forward()
    single.set_requires_grad([single.networks["D_B"], single.networks["D_A"]], False)
    single.backward_G()
    try:
        for name in ("G_AB", "G_BA"):
            ref = single.networks[name].master.grad
            got = r0["grads"][name][0]
            scale = ref.abs().max().item()
            assert (ref - got).abs().max().item() <= 1e-4 * scale, name
    finally:
        backend.set_ops(None)


def _cut_worker(rank, world, port, out_dir):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank),
                      LOCAL_RANK=str(rank), GANSLATE_DIST_BACKEND="gloo")
    torch.set_num_threads(2)
    from ganslate_amd.nn.native import backend
    from ganslate_amd.utils import communication
    from ganslate_amd.utils.builders import build_conf, build_gan
    from oracle.ops_ref import RefOps
    communication.init_distributed()
    backend.set_ops(RefOps(act_dtype=torch.float32))
    conf = build_conf([f"config={ROOT / 'tests/configs/cut_synthetic.yaml'}", "train.batch_size=1",
                       "train.gan.generator.n_residual_blocks=7", "train.gan.num_patches=16",
                       "train.dataset.final_size=[32,32]"])
    torch.manual_seed(9 + rank)
    model = build_gan(conf)              # CUT under data parallelism: the reference cannot run this (cut.py:205-211)
    A, B = _inputs(world)
    for step in range(2):
        model.set_input({"A": A[rank:rank + 1], "B": B[rank:rank + 1]})
        model.optimize_parameters()
    weights = {n: (net.master.detach().clone() if hasattr(net, "master")
                   else torch.cat([p.detach().flatten() for p in net.parameters()]))
               for n, net in model.networks.items()}
    torch.save(weights, Path(out_dir) / f"cut_rank{rank}.pt")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_cut_two_ranks_stay_in_sync(tmp_path):
    """CUT's encoder-only partial passes leave the upper gradient buckets to the catch-up reduction; generator,
    discriminator and mlp must still end every step identical on all ranks."""
    world = 2
    port = _free_port()
    mp.spawn(_cut_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = (torch.load(tmp_path / f"cut_rank{r}.pt") for r in range(world))
    for name in r0:
        assert torch.equal(r0[name], r1[name]), f"{name}: ranks diverged"


def _vnet_worker(rank, world, port, out_dir):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank),
                      LOCAL_RANK=str(rank), GANSLATE_DIST_BACKEND="gloo")
    torch.set_num_threads(2)
    from ganslate_amd.nn.native import backend
    from ganslate_amd.utils import communication
    from ganslate_amd.utils.builders import build_conf, build_gan
    from oracle.ops_ref import RefOps
    communication.init_distributed()
    backend.set_ops(RefOps(act_dtype=torch.float32))
    # the brats yaml's networks (BASELINE configs[4]: 3-D CycleGAN, DDP) at a small width / patch
    conf = build_conf([f"config={ROOT / 'tests/configs/cyclegan_vnet_synthetic.yaml'}", "train.batch_size=1",
                       "train.gan.generator.first_layer_channels=8", "train.gan.generator.down_blocks=[1,1]",
                       "train.gan.generator.up_blocks=[1,1]", "train.gan.pool_size=0",
                       "train.dataset.final_size=[16,16,16]"])
    torch.manual_seed(11 + rank)
    model = build_gan(conf)
    g = torch.Generator().manual_seed(78)
    A, B = torch.rand(world, 1, 16, 16, 16, generator=g) * 2 - 1, torch.rand(world, 1, 16, 16, 16, generator=g) * 2 - 1
    for step in range(2):
        model.set_input({"A": A[rank:rank + 1], "B": B[rank:rank + 1]})
        model.optimize_parameters()
    weights = {n: net.master.detach().clone() for n, net in model.networks.items()}
    torch.save(weights, Path(out_dir) / f"vnet_rank{rank}.pt")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_vnet3d_cyclegan_two_ranks_stay_in_sync(tmp_path):
    """3-D CycleGAN with Vnet3D + PatchGAN3D under data parallelism (BASELINE configs[4]): conv weights AND the PReLU
    slopes stored behind them in the flat buffer are averaged; all ranks end every step with identical parameters."""
    world = 2
    port = _free_port()
    mp.spawn(_vnet_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = (torch.load(tmp_path / f"vnet_rank{r}.pt") for r in range(world))
    for name in r0:
        assert torch.equal(r0[name], r1[name]), f"{name}: ranks diverged"


def _multiscale_worker(rank, world, port, out_dir):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank),
                      LOCAL_RANK=str(rank), GANSLATE_DIST_BACKEND="gloo")
    torch.set_num_threads(2)
    import random
    from ganslate_amd.nn.native import backend
    from ganslate_amd.utils import communication
    from ganslate_amd.utils.builders import build_conf, build_gan
    from oracle.ops_ref import RefOps
    communication.init_distributed()
    backend.set_ops(RefOps(act_dtype=torch.float32))
    conf = build_conf([f"config={ROOT / 'tests/configs/cyclegan3d_synthetic.yaml'}", "train.batch_size=1",
                       "train.gan.generator.n_residual_blocks=1", "train.gan.pool_size=0",
                       "train.gan.discriminator._target_=ganslate.nn.discriminators.MultiScalePatchGAN3D",
                       "train.gan.discriminator.n_layers=1", "train.gan.discriminator.ndf=8",
                       "train.gan.discriminator.scales=2", "train.dataset.final_size=[16,16,16]"])
    torch.manual_seed(21 + rank)
    model = build_gan(conf)
    random.seed(100 + rank)                   # every rank draws its own crop windows, like it sees its own data
    g = torch.Generator().manual_seed(79)
    A, B = torch.rand(world, 1, 16, 16, 16, generator=g) * 2 - 1, torch.rand(world, 1, 16, 16, 16, generator=g) * 2 - 1
    for step in range(2):
        model.set_input({"A": A[rank:rank + 1], "B": B[rank:rank + 1]})
        model.optimize_parameters()
    weights = {f"{n}.{i}": sub.master.detach().clone() for n, net in model.networks.items()
               for i, sub in enumerate(net.native_children() if hasattr(net, "native_children") else [net])}
    torch.save(weights, Path(out_dir) / f"ms_rank{rank}.pt")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_multiscale_discriminators_two_ranks_stay_in_sync(tmp_path):
    """composite networks under data parallelism: each PatchGAN3D of a MultiScalePatchGAN3D is broadcast from rank 0 and has
    its flat gradient averaged (BaseGAN._native_nets); ranks with different inputs and crop windows end with identical
    parameters in every sub-network"""
    world = 2
    port = _free_port()
    mp.spawn(_multiscale_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = (torch.load(tmp_path / f"ms_rank{r}.pt") for r in range(world))
    assert sum(k.startswith("D_A.") for k in r0) == 2 and sum(k.startswith("D_B.") for k in r0) == 2
    for name in r0:
        assert torch.equal(r0[name], r1[name]), f"{name}: ranks diverged"


def _attn_worker(rank, world, port, out_dir):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank),
                      LOCAL_RANK=str(rank), GANSLATE_DIST_BACKEND="gloo")
    torch.set_num_threads(2)
    from ganslate_amd.nn.native import backend
    from ganslate_amd.utils import communication
    from oracle.ops_ref import RefOps
    communication.init_distributed()
    backend.set_ops(RefOps(act_dtype=torch.float32))
    net, xs, gs = _attn_net_and_data(world)
    net.parallelize(bucket_bytes=1 << 10)       # many small buckets: the tail bucket closes long before the attention blocks run
    y = net(xs[rank])
    y.backward(gs[rank])
    fired = len(net._reduce_handles)
    scale = net.finish_grad_reduction()
    torch.save({"grad": net.master.grad.clone() * scale, "fired": fired}, Path(out_dir) / f"rank{rank}.pt")
    dist.barrier()
    dist.destroy_process_group()


def _attn_net_and_data(world):
    from ganslate_amd.nn.discriminators import SelfAttentionPatchGAN3D
    torch.manual_seed(21)
    net = SelfAttentionPatchGAN3D(1, 8, 2, (4, 4, 4), "instance")
    net.init_weights("normal", 0.05)
    with torch.no_grad():       # gamma = 0 at init would silence the attention path: give it a value
        for ex in net.extras:
            if ex.name.endswith(".gamma"):
                net.master[net.x_off[ex.name]] = 0.7
    g = torch.Generator().manual_seed(22)
    xs = [torch.rand(1, 1, 24, 24, 24, generator=g) * 2 - 1 for _ in range(world)]
    with torch.no_grad():
        shape = net(xs[0]).shape
    gs = [torch.randn(shape, generator=g) for _ in range(world)]
    return net, xs, gs


@pytest.mark.timeout(600)
def test_attention_parameters_are_reduced_after_their_backward(tmp_path):
    """the SelfAttentionBlock parameters sit at the tail of the flat gradient; the bucket that holds them
    must not be all-reduced before the block's backward has written them. Two ranks x one volume against the single-process
    sum of the two gradients, attention parameters included."""
    world = 2
    port = _free_port()
    mp.spawn(_attn_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = (torch.load(tmp_path / f"rank{r}.pt") for r in range(world))
    assert r0["fired"] > 0, "bucketed all-reduce did not start during the backward pass"
    assert torch.equal(r0["grad"], r1["grad"])
    from ganslate_amd.nn.native import backend
    from oracle.ops_ref import RefOps
    backend.set_ops(RefOps(act_dtype=torch.float32))
    try:
        net, xs, gs = _attn_net_and_data(world)
        for x, g in zip(xs, gs):
            net(x).backward(g)
        ref = net.master.grad / world
        x0 = min(net.x_off.values())
        assert ref[x0:].abs().max() > 0, "the case must exercise the attention parameters"
        scale = ref.abs().max().item()
        assert (ref - r0["grad"]).abs().max().item() <= 1e-5 * scale
        tail = ref[x0:].abs().max().item()
        assert (ref[x0:] - r0["grad"][x0:]).abs().max().item() <= 1e-5 * tail, "attention parameter gradients were not summed"
    finally:
        backend.set_ops(None)


# ---- four ranks: odd bucket splits, CUT + mlp, sampler shards ------------------------------------------------
def _worker4(rank, world, port, out_dir):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank),
                      LOCAL_RANK=str(rank), GANSLATE_DIST_BACKEND="gloo")
    torch.set_num_threads(1)
    from ganslate_amd.utils import communication
    from ganslate_amd.data.samplers import InfiniteSampler
    import itertools
    communication.init_distributed()
    # sampler: every rank walks the same seed-shared permutation, strided by rank (samplers.py:20-58)
    sampler = InfiniteSampler(10, shuffle=True)
    shard = [int(i) for i in itertools.islice(iter(sampler), 15)]
    model = _build(1, seed=5 + rank)       # different init per rank: parallelize() must broadcast rank 0's weights
    # re-bucket with a small, odd bucket size: many buckets of uneven length, the last one a remainder
    for net in model.networks.values():
        net.parallelize(bucket_bytes=(3 << 10) + 4)
    nb = {n: len(net._buckets) for n, net in model.networks.items()}
    A, B = _inputs(world)
    model.set_input({"A": A[rank:rank + 1], "B": B[rank:rank + 1]})
    model.forward()
    model.set_requires_grad([model.networks["D_B"], model.networks["D_A"]], False)
    model.backward_G()
    grads = {}
    for name in ("G_AB", "G_BA"):
        net = model.networks[name]
        fired_async = len(net._reduce_handles) > 0
        scale = net.finish_grad_reduction()
        grads[name] = (net.master.grad.clone() * scale, fired_async)
    model.optimizers["G"].zero_grad()
    for _ in range(2):
        model.optimize_parameters()
    weights = {n: net.master.detach().clone() for n, net in model.networks.items()}
    torch.save({"grads": grads, "weights": weights, "shard": shard, "seed": sampler._seed, "buckets": nb},
               Path(out_dir) / f"w4_rank{rank}.pt")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_four_ranks_odd_buckets_gradient_and_sampler_shards(tmp_path):
    """4 ranks x batch 1 = the single-process gradient on the concatenated batch of 4 (1/world folded in by
    finish_grad_reduction), with ~3 KiB buckets (dozens per generator, uneven, a remainder bucket); weights identical on all
    ranks after two steps; the sampler's shards share the seed, are disjoint within an epoch and together cover it"""
    world = 4
    port = _free_port()
    mp.spawn(_worker4, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    rs = [torch.load(tmp_path / f"w4_rank{r}.pt") for r in range(world)]
    assert rs[0]["buckets"]["G_AB"] > 8, rs[0]["buckets"]
    for r in rs[1:]:
        assert r["seed"] == rs[0]["seed"]
        for name in rs[0]["weights"]:
            assert torch.equal(rs[0]["weights"][name], r["weights"][name]), f"{name}: ranks diverged"
        for name in ("G_AB", "G_BA"):
            assert torch.equal(rs[0]["grads"][name][0], r["grads"][name][0])
    assert all(r["grads"]["G_AB"][1] for r in rs), "bucketed all-reduce did not start during the last backward pass"
    # shards: position k of rank r is element r + 4k of the shared stream — the first 40 stream elements are 4 epochs of 10
    stream = [None] * 60
    for r, res in enumerate(rs):
        for k, idx in enumerate(res["shard"]):
            stream[r + 4 * k] = idx
    for e in range(6):
        assert sorted(stream[10 * e:10 * e + 10]) == list(range(10)), (e, stream)
    os.environ.pop("WORLD_SIZE", None)
    from ganslate_amd.nn.native import backend
    single = _build(4, seed=5)
    A, B = _inputs(4)
    single.set_input({"A": A, "B": B})
    single.forward()
    single.set_requires_grad([single.networks["D_B"], single.networks["D_A"]], False)
    single.backward_G()
    try:
        for name in ("G_AB", "G_BA"):
            ref = single.networks[name].master.grad
            got = rs[0]["grads"][name][0]
            scale = ref.abs().max().item()
            assert (ref - got).abs().max().item() <= 1e-4 * scale, name
    finally:
        backend.set_ops(None)


@pytest.mark.timeout(900)
def test_cut_four_ranks_stay_in_sync(tmp_path):
    """CUT (generator, discriminator, patch mlp) on four ranks: identical weights on every rank after two steps"""
    world = 4
    port = _free_port()
    mp.spawn(_cut_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    rs = [torch.load(tmp_path / f"cut_rank{r}.pt") for r in range(world)]
    for r in rs[1:]:
        for name in rs[0]:
            assert torch.equal(rs[0][name], r[name]), f"{name}: ranks diverged"
