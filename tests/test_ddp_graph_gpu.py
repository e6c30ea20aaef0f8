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


def test_data_parallel_graph_step_matches_single_process(hip_ops, monkeypatch):
    import datetime
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
        assert ddp._graph is not None and ddp._graph_update is not None and len(ddp._reduced_nets) == 4
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
