"""engines.Trainer on the HIP path: a YAML-configured CycleGAN run through init_engine (the reference's entry point,
ganslate/engines/utils.py:14-22) for a few iterations on the GPU, with logging and a reference-layout checkpoint."""
from pathlib import Path

import pytest
import torch

pytestmark = pytest.mark.gpu

CONF = Path(__file__).parent / "configs" / "cyclegan_synthetic.yaml"


def test_trainer_runs_on_gpu(hip_ops, tmp_path):
    from ganslate_amd.engines import init_engine
    tr = init_engine("train", [f"config={CONF}", f"train.output_dir={tmp_path}", "train.batch_size=2",
                               "train.n_iters=4", "train.n_iters_decay=2", "train.checkpointing.freq=3",
                               "train.logging.freq=2", "train.seed=3"])
    assert next(iter(tr.model.networks.values())).ops.name == "hip"
    tr.run()
    torch.cuda.synchronize()
    assert [h[0] for h in tr.history] == [2, 4, 6]
    for _, losses, metrics in tr.history:
        assert all(v == v and abs(v) < 1e4 for v in losses.values()), losses
        assert -0.42 <= metrics["ssim_A"] <= 1.0 and -0.42 <= metrics["ssim_B"] <= 1.0   # 1 - d, d in [0, sqrt 2]
    ck = torch.load(tmp_path / "checkpoints" / "6.pth", map_location="cpu")
    assert ck["G_AB"]["model.1.weight"].shape == (64, 3, 7, 7) and ck["D_A"]["model.0.weight"].shape == (64, 3, 4, 4)
