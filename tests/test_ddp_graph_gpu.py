"""Data-parallel captured step on one GPU: a 1-rank RCCL group drives the same code a multi-GPU run takes — two graphs
per iteration (everything up to the last backward pass | the held-back Adam launches) with one all-reduce per network
between them — and must reproduce the single-process step. The multi-rank arithmetic (averaged rank gradients = big-batch
gradient) is covered on CPU by tests/test_ddp_cpu.py."""
import socket

import pytest
import torch

from tests.helpers import build_product_cyclegan, load_golden_steps
from tests.test_step_graph_gpu import _run

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("reduction", ["0", "auto"])
def test_data_parallel_graph_step_matches_single_process(hip_ops, monkeypatch, reduction):
    """reduction "0": the all-reduce between the two graphs; "auto" (the default): both forms are built, replayed once with
    frozen weights (image pools restored in between) and compared, and the captured collectives are kept"""
    import datetime
    monkeypatch.setenv("GS_DDP_GRAPH_COLLECTIVES", reduction)
    import torch.distributed as dist
    c = dict(load_golden_steps()["c64_default"]["config"])
    c["pool_size"] = 3
    frozen = ("train.gan.optimizer.lr_G=0.0", "train.gan.optimizer.lr_D=0.0")
    monkeypatch.setenv("LOCAL_RANK", "0")
    single = build_product_cyclegan(c, frozen)
    want = _run(single, c, 5)
    assert single._graph is not None and single._graph_update is None
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1,
                            timeout=datetime.timedelta(minutes=2))
    try:
        monkeypatch.setenv("GS_FORCE_DDP", "1")
        ddp = build_product_cyclegan(c, frozen)
        assert all(net._dist is not None for net in ddp.networks.values())
        got = _run(ddp, c, 5)
        assert ddp._graph is not None and ddp._graph_update is not None
        if reduction == "0":
            assert len(ddp._reduced_nets) == 4 and not ddp._graph_collectives
        else:
            chk = ddp.ddp_self_check
            assert chk["forms_agree"] and chk["ranks_agree"] and chk["max_abs_diff"] == 0.0 and chk["max_abs_grad"] > 0.0
            assert chk["kept"] == "captured" and ddp._graph_collectives and ddp._reduced_nets == []
        for s in range(5):
            for k, v in want[s][0].items():
                assert got[s][0][k] == pytest.approx(v, rel=1e-4, abs=1e-6), (s, k)
        for oa, ob in zip(single.optimizers.values(), ddp.optimizers.values()):
            for pa, pb in zip(oa.param_groups[0]["params"], ob.param_groups[0]["params"]):
                assert oa.state[pa]["step"] == ob.state[pb]["step"] == 5
                for key in ("exp_avg", "exp_avg_sq"):
                    a, b = oa.state[pa][key], ob.state[pb][key]
                    assert (a - b).norm().item() <= 1e-3 * a.norm().item(), key
        # weights moving, a ragged batch in between (launch-by-launch fallback with the bucketed all-reduce)
        live = build_product_cyclegan(c)
        from tests.helpers import golden_inputs
        for s, batch in enumerate([2, 2, 2, 1, 2, 2]):
            A, B = golden_inputs(c, s)
            live.set_input({"A": A[:batch], "B": B[:batch]})
            live.optimize_parameters()
            live.update_learning_rate()
        torch.cuda.synchronize()
        assert live._graph_update is not None
        assert all(float(v.detach()) == float(v.detach()) and abs(float(v.detach())) < 1e3
                   for v in live.losses.values() if v is not None)
        assert [st["step"] for st in live.optimizers["G"].state.values()] == [6, 6]
    finally:
        dist.destroy_process_group()


def test_self_check_falls_back_when_the_collectives_cannot_be_captured(hip_ops, monkeypatch):
    """a runtime that refuses to record the collectives inside a graph (simulated: the capture of that form raises, leaving
    the state a failed capture leaves) must not take the run down: the all-reduce-between-graphs form is kept"""
    import datetime
    import torch.distributed as dist
    from ganslate_amd.nn.gans.base import BaseGAN
    c = dict(load_golden_steps()["c64_default"]["config"])
    c["pool_size"] = 3
    monkeypatch.setenv("LOCAL_RANK", "0")
    monkeypatch.setenv("GS_FORCE_DDP", "1")
    real = BaseGAN._capture_graphs

    def refusing(self, dp_nets, form):
        if form == "captured":
            self._graph_broken = True
            self._set_external_host_state(False)
            raise RuntimeError("simulated: collectives are not capturable")
        return real(self, dp_nets, form)
    monkeypatch.setattr(BaseGAN, "_capture_graphs", refusing)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1,
                            timeout=datetime.timedelta(minutes=2))
    try:
        ddp = build_product_cyclegan(c)
        got = _run(ddp, c, 4)
        assert ddp._graph is not None and ddp._graph_update is not None and not ddp._graph_collectives
        assert ddp.ddp_self_check["kept"] == "between" and "simulated" in ddp.ddp_self_check["error"]
        assert len(ddp._reduced_nets) == 4 and not ddp._graph_broken
        assert all(v == v and abs(v) < 1e3 for s in got for v in s[0].values())
    finally:
        dist.destroy_process_group()


# ---- two ranks over RCCL (runs the moment the box has >= 2 GPUs; the 1-GPU pool skips it) -----------------------------
DDP_PATHS = {
    "launch_by_launch": {"GS_STEP_GRAPH": "0"},                     # bucketed all-reduce overlapped with the last backward pass
    "two_graphs": {"GS_DDP_GRAPH_COLLECTIVES": "0"},               # graph | one all-reduce per network | graph
    "captured_collectives": {"GS_DDP_GRAPH_COLLECTIVES": "1"},     # the bucketed all-reduces captured inside the step graph
    "self_check": {},                                              # default: both built and compared, captured kept
}


def _rccl_worker(rank, world, port, out_dir, path):
    import os
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    sys.path.insert(0, str(root))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank),
                      LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY="0", **DDP_PATHS[path])
    import random
    import torch.distributed as dist
    from ganslate_amd.utils import communication
    from tests.helpers import FROZEN, adam_first_moments, golden_inputs
    if world > 1:
        communication.init_distributed()
    else:          # the data-parallel code path with a 1-rank RCCL group (what a single-GPU box can run)
        import datetime
        os.environ["GS_FORCE_DDP"] = "1"
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                                timeout=datetime.timedelta(minutes=2))
    c = dict(load_golden_steps()["c64_default"]["config"])
    c["pool_size"], c["batch"] = 0, 1
    # (a) frozen weights: three identical iterations (launch by launch, capture, replay) -> the averaged gradient
    random.seed(c["seed"])
    model = build_product_cyclegan(c, FROZEN)
    A, B = golden_inputs(dict(c, batch=world), 0)
    for _ in range(3):
        model.set_input({"A": A[rank:rank + 1], "B": B[rank:rank + 1]})
        model.optimize_parameters()
    torch.cuda.synchronize()
    if path == "launch_by_launch":
        assert model._graph is None
    elif path == "two_graphs":
        assert model._graph is not None and model._graph_update is not None and not model._graph_collectives
    else:
        assert model._graph is not None and model._graph_update is not None and model._graph_collectives
        if path == "self_check":
            chk = model.ddp_self_check
            assert chk["world"] == world and chk["forms_agree"] and chk["ranks_agree"] and chk["kept"] == "captured", chk
    # (b) weights moving: four iterations on this rank's shard -> the replicas must stay identical bit for bit
    random.seed(c["seed"])
    live = build_product_cyclegan(c)
    for s in range(4):
        A2, B2 = golden_inputs(dict(c, batch=world), s)
        live.set_input({"A": A2[rank:rank + 1], "B": B2[rank:rank + 1]})
        live.optimize_parameters()
        live.update_learning_rate()
    torch.cuda.synchronize()
    torch.save({"moments": adam_first_moments(model),
                "losses": {k: float(v.detach()) for k, v in model.losses.items() if v is not None},
                "weights": {n: {k: v.float().cpu() for k, v in net.state_dict().items()} for n, net in live.networks.items()}},
               Path(out_dir) / f"rccl_{path}_rank{rank}.pt")
    dist.barrier()
    dist.destroy_process_group()


def _check_against_single_process(tmp_path, path, world):
    import random
    from tests.helpers import FROZEN, adam_first_moments, golden_inputs
    ranks = [torch.load(tmp_path / f"rccl_{path}_rank{r}.pt") for r in range(world)]
    r0 = ranks[0]
    c = dict(load_golden_steps()["c64_default"]["config"])
    c["pool_size"], c["batch"] = 0, world
    random.seed(c["seed"])
    single = build_product_cyclegan(c, FROZEN)
    A, B = golden_inputs(c, 0)
    for _ in range(3):
        single.set_input({"A": A, "B": B})
        single.optimize_parameters()
    torch.cuda.synchronize()
    want = adam_first_moments(single)
    for net, per in want.items():
        for n, w in per.items():
            for r in ranks[1:]:
                assert torch.equal(r0["moments"][net][n], r["moments"][net][n]), (path, net, n, "ranks differ")
            if w.norm().item() > 1e-9:
                assert (r0["moments"][net][n] - w).norm().item() <= 2e-2 * w.norm().item(), (path, net, n)
    random.seed(c["seed"])
    live = build_product_cyclegan(c)
    for s in range(4):
        A, B = golden_inputs(c, s)
        live.set_input({"A": A, "B": B})
        live.optimize_parameters()
        live.update_learning_rate()
    torch.cuda.synchronize()
    for net, sd in r0["weights"].items():
        mine = {k: v.float().cpu() for k, v in live.networks[net].state_dict().items()}
        for k, v in sd.items():
            for r in ranks[1:]:
                assert torch.equal(v, r["weights"][net][k]), (path, net, k, "replicas diverged")
            # four Adam steps of +-lr per weight: the replicas' path (per-sample gradients averaged over RCCL) and the
            # big-batch path differ by summation order only -> the same update up to the sign of noise-level gradients
            assert (v - mine[k]).abs().max().item() <= 4 * 2.5e-4, (path, net, k)


@pytest.mark.timeout(1800)
@pytest.mark.parametrize("path", list(DDP_PATHS))
def test_data_parallel_paths_with_one_rank_over_rccl(hip_ops, tmp_path, path):
    """the worker of the 2-rank test below with a 1-rank RCCL group in a child process (what this single-GPU pool can run):
    every one of the data-parallel paths goes through its collectives and must reproduce the single-process run"""
    import torch.multiprocessing as mp
    mp.spawn(_rccl_worker, args=(1, _free_port(), str(tmp_path), path), nprocs=1, join=True)
    _check_against_single_process(tmp_path, path, 1)


@pytest.mark.timeout(1800)
@pytest.mark.parametrize("path", list(DDP_PATHS))
def test_two_ranks_over_rccl_average_to_the_big_batch_gradient(hip_ops, tmp_path, path):
    """2 ranks x batch 1 over RCCL (xGMI) against one process on the batch of 2, for each of the three data-parallel paths
    (ganslate/nn/gans/base.py:172-189 wraps every network in DistributedDataParallel): with frozen weights
    Adam's first moment after three identical iterations is 0.875 * g, and the averaged rank gradients must equal the
    big-batch gradient (InstanceNorm is per sample, every loss a batch mean); with moving weights the two replicas must hold
    bit-identical parameters after four iterations and equal the single-process batch-2 run's to bf16-step accuracy."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    import torch.multiprocessing as mp
    world = 2
    mp.spawn(_rccl_worker, args=(world, _free_port(), str(tmp_path), path), nprocs=world, join=True)
    _check_against_single_process(tmp_path, path, world)


@pytest.mark.timeout(600)
def test_captured_collectives_mode_single_rank(hip_ops):
    """GS_DDP_GRAPH_COLLECTIVES=1: the bucketed RCCL all-reduces issued during the last backward pass are captured
    INTO the step graph (reduction overlapped with the remaining backward; the Adam launches are a second graph). Run in a child
    process with a hard timeout (a capture problem in RCCL must not take the test session down); with one rank the
    replayed iterations must equal the single-process run bit for bit."""
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    r = subprocess.run([sys.executable, str(root / "tools" / "ddp_graph_collectives_probe.py")], capture_output=True,
                       text=True, timeout=420, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "collectives captured True" in r.stdout and "bitwise equal to single process: True" in r.stdout, r.stdout[-2000:]
