"""engines.Trainer (ganslate/engines/trainer.py:11-112) driven through init_engine with a YAML,
on the CPU oracle backend: iteration range, logging cadence, LR schedule, checkpoint cadence and resume."""
import copy
from pathlib import Path

import pytest
import torch

from ganslate_amd.nn.native import backend
from oracle.ops_ref import RefOps

CONF = Path(__file__).parent / "configs" / "cyclegan_synthetic.yaml"


@pytest.fixture()
def fp32_oracle_backend():
    backend.set_ops(RefOps(act_dtype=torch.float32))
    yield
    backend.set_ops(None)


def _args(out_dir, *extra):
    return [f"config={CONF}", "train.cuda=false", f"train.output_dir={out_dir}", "train.batch_size=1",
            "train.n_iters=3", "train.n_iters_decay=2", "train.dataset.final_size=[32,32]", "train.seed=7",
            "train.gan.generator.n_residual_blocks=1", "train.metrics.ssim=false",
            "train.checkpointing.freq=2", "train.logging.freq=1", *extra]


def test_trainer_runs_logs_checkpoints_and_resumes(fp32_oracle_backend, tmp_path):
    from ganslate_amd.engines import init_engine
    tr = init_engine("train", _args(tmp_path))
    assert list(tr.iters) == [1, 2, 3, 4, 5]                       # range(1, 1 + n_iters + n_iters_decay)
    tr.run()
    assert [h[0] for h in tr.history] == [1, 2, 3, 4, 5]
    for _, losses, metrics in tr.history:
        assert set(losses) == {"G_AB", "G_BA", "D_A", "D_B", "cycle_A", "cycle_B"}
        assert all(v == v and abs(v) < 1e4 for v in losses.values())
        assert {"D_A_real", "D_A_fake", "D_B_real", "D_B_fake"} <= set(metrics)
    # linear decay after n_iters (nn/utils.py:83-99): factor 1 - max(0, it + 1 - n_iters) / (n_iters_decay + 1); the
    # scheduler has stepped 5 times -> 1 - 3/3
    assert tr.model.optimizers["G"].param_groups[0]["lr"] == pytest.approx(0.0, abs=1e-12)
    assert [h[0] for h in tr.history][-1] == 5
    ckpts = sorted(p.name for p in (tmp_path / "checkpoints").iterdir())
    assert ckpts == ["2.pth", "4.pth"]                             # freq 2, rank 0
    ck = torch.load(tmp_path / "checkpoints" / "4.pth", map_location="cpu")
    assert {"G_AB", "G_BA", "D_A", "D_B", "optimizer_G", "optimizer_D"} <= set(ck)
    assert "model.1.weight" in ck["G_AB"] and "encoder.1.weight" in ck["G_AB"]   # reference key names

    # resume from iteration 4: weights of the checkpoint, iteration range continues at 5
    tr2 = init_engine("train", _args(tmp_path, "train.checkpointing.load_iter=4"))
    assert list(tr2.iters) == [5]
    sd = tr2.model.networks["G_AB"].state_dict()
    assert torch.equal(sd["model.1.weight"].cpu(), ck["G_AB"]["model.1.weight"])
    tr2.run()
    assert [h[0] for h in tr2.history] == [5]


def test_trainer_is_deterministic_for_a_seed(fp32_oracle_backend, tmp_path):
    from ganslate_amd.engines import init_engine
    runs = []
    for k in range(2):
        tr = init_engine("train", _args(tmp_path / f"r{k}", "train.n_iters=2", "train.n_iters_decay=0"))
        tr.run()
        runs.append(copy.deepcopy(tr.history))
    assert runs[0] == runs[1]


def test_trainer_runs_validation_with_sliding_window(fp32_oracle_backend, tmp_path):
    """Validator cadence inside Trainer.run (trainer.py:103-108: iter % val.freq == 0 and iter >= val.start_after) with
    patch-wise inference through the sliding-window inferer (engines/base.py:28-50), on volumes larger than the window"""
    from ganslate_amd.engines import init_engine
    conf3d = Path(__file__).parent / "configs" / "cyclegan3d_val_synthetic.yaml"
    args = [f"config={conf3d}", "train.cuda=false", f"train.output_dir={tmp_path}", f"val.output_dir={tmp_path}",
            "train.seed=7"]
    tr = init_engine("train", args)
    assert tr.validator is not None and tr.validator.sliding_window_inferer is not None
    calls = []
    infer = tr.model.infer
    tr.model.infer = lambda x, *a, **k: (calls.append(tuple(x.shape)), infer(x, *a, **k))[1]
    tr.run()
    assert [h[0] for h in tr.validator.history] == [2, 4]
    # 16 x 24 x 20 volumes, 16^3 windows at 25 % overlap: 1 x 2 x 2 windows, two per launch
    assert calls and all(s == (2, 1, 16, 16, 16) for s in calls)
    for _, _, m in tr.validator.history:
        assert set(m) == {"mae", "mse", "nmse", "psnr"} and all(v == v for v in m.values())
    # training state is restored after validation
    assert all(getattr(net, "training", True) for net in tr.model.networks.values())


def test_validation_metrics_are_scored_per_sample_and_denormalised(fp32_oracle_backend, tmp_path):
    """ValTestMetrics.get_metrics scores every sample of a batch on its own (val_test_metrics.py:152-153: psnr's data range
    and nmse's norm are per sample) after the dataset's `denormalize` hook, and `compute_over_input` adds the Original_*
    scores of the untranslated input (validator_tester.py:66-86)"""
    import numpy as np
    from ganslate_amd.engines import init_engine
    from ganslate_amd.engines.validator import METRICS
    conf3d = Path(__file__).parent / "configs" / "cyclegan3d_val_synthetic.yaml"
    args = [f"config={conf3d}", "train.cuda=false", f"train.output_dir={tmp_path}", f"val.output_dir={tmp_path}",
            "train.seed=7", "val.batch_size=2", "val.metrics.compute_over_input=true"]
    tr = init_engine("train", args)
    v = tr.validator
    loader = next(iter(v.data_loaders.values()))
    loader.dataset.denormalize = lambda t: (t + 1) * 500.0          # the hook the reference's medical datasets define
    seen = []
    infer = v.infer
    v.infer = lambda x: (lambda y: (seen.append(y.detach().float().cpu()), y)[1])(infer(x))
    v.run(current_idx=0)
    _, _, mean = v.history[-1]
    assert set(mean) == {k for m in ("mae", "mse", "nmse", "psnr") for k in (m, f"Original_{m}")}
    # recompute from the recorded predictions, sample by sample
    rows = []
    for pred, data in zip(seen, loader):
        for i in range(pred.shape[0]):
            p, t = ((pred[i] + 1) * 500.0).numpy(), ((data["B"][i].float() + 1) * 500.0).numpy()
            rows.append({k: METRICS[k](t, p) for k in ("mae", "mse", "nmse", "psnr")})
    assert len(rows) >= 2 and len(seen[0]) == 2
    for k in ("mae", "mse", "nmse", "psnr"):
        assert mean[k] == pytest.approx(float(np.mean([r[k] for r in rows])), rel=1e-6), k
