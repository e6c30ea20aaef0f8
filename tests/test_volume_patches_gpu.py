"""3-D training-patch path on the GPU (csrc/volproc.hip): gs_patch_zscore against vectors recorded from
the reference's own z_score_normalize (tests/golden/volume_patches.json), against the op-level oracle on fp32 and int16
volumes with and without range scaling, the NaN edge, the pipeline against the host dataset path, and a Trainer run fed
through it. Tolerance: the expression is evaluated operation for operation in fp32; mean / std come from double sums
instead of torch's fp32 cascade (last-ulp differences) -> 2e-6 absolute on values in [-1, 1]."""
import json
import random
import time
from pathlib import Path

import numpy as np
import pytest
import torch

from ganslate_amd.data.utils.stochastic_focal_patching import StochasticFocalPatchSampler
from oracle.ops_ref import RefOps
from tests.test_volume_patches_cpu import GOLD, _conf, _volume_folder, seeded_volume

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", sorted(GOLD))
def test_kernel_against_the_reference_vectors(hip_ops, name):
    c = GOLD[name]
    rec = c["draws"][0]
    ps = c["patch_size"] if len(c["patch_size"]) == 3 else [1, *c["patch_size"]]
    for key, shape, seed, start in (("A", c["shape_A"], c["seed"], rec["start_A"]),
                                    ("B", c["shape_B"], c["seed"] + 100, rec["start_B"])):
        vol = seeded_volume(shape, seed).to(hip_ops.device)
        g = rec["z_" + key]
        for scale, want in (((-1.0, 1.0), g["samples"]), (None, rec["z_plain_" + key])):
            out = torch.empty(ps, dtype=torch.float32, device=hip_ops.device)
            hip_ops.patch_zscore(vol, start, ps, out, scale)
            got = out.flatten().cpu()[torch.tensor(g["samples_at"])]
            assert torch.allclose(got, torch.tensor(want), atol=2e-6, rtol=2e-6, equal_nan=True), (name, key, scale)
        if g["min"] == g["min"]:
            z = out.new_empty(ps)
            hip_ops.patch_zscore(vol, start, ps, z, (-1.0, 1.0))
            assert float(z.min()) == pytest.approx(-1.0, abs=1e-6) and float(z.max()) == pytest.approx(1.0, abs=1e-6)
            assert float(z.double().mean()) == pytest.approx(g["mean"], abs=2e-6)


@pytest.mark.parametrize("dtype", [torch.float32, torch.int16])
@pytest.mark.parametrize("shape,start,size", [((155, 240, 240), (13, 50, 61), (32, 128, 128)),
                                              ((40, 50, 60), (0, 0, 0), (40, 50, 60)),
                                              ((9, 33, 17), (8, 30, 3), (1, 3, 13)),
                                              ((140, 200, 210), (7, 9, 11), (128, 128, 128))])
@pytest.mark.parametrize("scale", [(-1.0, 1.0), (0.0, 255.0), None])
def test_kernel_against_the_oracle(hip_ops, dtype, shape, start, size, scale):
    vol = seeded_volume(shape, 3).to(dtype)
    outs = []
    for ops, dev in ((RefOps(), "cpu"), (hip_ops, hip_ops.device)):
        out = torch.full(size, 7.0, dtype=torch.float32, device=dev)
        ops.patch_zscore(vol.to(dev), start, size, out, scale)
        outs.append(out.cpu())
    span = (scale[1] - scale[0]) if scale else 8.0
    assert torch.allclose(outs[1], outs[0], atol=2e-6 * span, rtol=2e-6)


def test_constant_patch_gives_nan_and_bad_windows_are_refused(hip_ops):
    vol = torch.full((8, 9, 10), 5.0, device=hip_ops.device)
    out = torch.zeros((4, 4, 4), device=hip_ops.device)
    hip_ops.patch_zscore(vol, (1, 1, 1), (4, 4, 4), out, (-1.0, 1.0))
    assert torch.isnan(out).all()                            # 0 / 0, like the reference's z_score_normalize
    with pytest.raises(RuntimeError, match="outside the 8 x 9 x 10 volume"):
        hip_ops.patch_zscore(vol, (5, 1, 1), (4, 4, 4), out, (-1.0, 1.0))


@pytest.mark.parametrize("dtype", [np.float32, np.int16])
def test_pipeline_equals_the_host_dataset_path(hip_ops, tmp_path, dtype):
    from ganslate_amd.data.device_volumes import DeviceVolumePipeline
    from ganslate_amd.data.volume_datasets import UnpairedVolumeDataset, collate_raw
    root = _volume_folder(tmp_path, dtype)
    host, raw = UnpairedVolumeDataset(_conf(root, False)), UnpairedVolumeDataset(_conf(root, True))
    pipe = DeviceVolumePipeline(raw, hip_ops.device, ops=hip_ops)
    random.seed(21)
    want = [host[i] for i in range(5)]
    random.seed(21)
    got = pipe(collate_raw([raw[i] for i in range(5)]))
    for key in "AB":
        ref = torch.stack([w[key] for w in want])
        assert got[key].is_cuda and got[key].shape == ref.shape
        assert torch.allclose(got[key].cpu(), ref, atol=2e-6, rtol=0)
    assert all(v.is_cuda for v in pipe.resident.values())


def test_patch_rate_at_the_brats_shape(hip_ops):
    """128^3 patches out of a resident 155 x 240 x 240 volume (BASELINE configs[4] patch size): patches per second through
    the three launches — far above the 13-16 vol/s of the 3-D training steps"""
    vol = seeded_volume((155, 240, 240), 1).to(hip_ops.device)
    out = torch.empty((128, 128, 128), device=hip_ops.device)
    for _ in range(3):
        hip_ops.patch_zscore(vol, (10, 40, 50), (128, 128, 128), out, (-1.0, 1.0))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(50):
        hip_ops.patch_zscore(vol, (k % 20, 40 + k, 50), (128, 128, 128), out, (-1.0, 1.0))
    torch.cuda.synchronize()
    rate = 50 / (time.perf_counter() - t0)
    print(f"\ndevice patch path: {rate:.0f} patches/s of 128^3")
    assert rate > 500


def test_trainer_runs_on_the_device_volume_pipeline(hip_ops, tmp_path):
    """volume folder -> coordinate-only samples -> DeviceVolumePipeline -> CycleGAN(Resnet3D).set_input: three iterations,
    finite losses, and the batches the model saw equal the host path's for the same draws"""
    from ganslate_amd.engines import init_engine
    root = _volume_folder(tmp_path / "data" if (tmp_path / "data").mkdir() is None else tmp_path)
    args = ["config=tests/configs/cyclegan3d_volumefolder.yaml", f"train.output_dir={tmp_path / 'out'}",
            f"train.dataset.root={root}", "train.seed=5", "train.dataset.patch_size=[16,16,16]"]
    runs = {}
    for dev_tf in (True, False):
        trainer = init_engine("train", args + [f"train.dataset.device_transforms={dev_tf}"])
        seen = []
        orig = trainer.model.set_input
        trainer.model.set_input = lambda data, _o=orig, _s=seen: (_s.append({k: v.detach().float().cpu().clone()
                                                                           for k, v in data.items()}), _o(data))[1]
        trainer.run()
        assert (trainer.input_pipeline is not None) == dev_tf
        assert all(float(v) == float(v) for v in trainer.model.losses.values() if v is not None)
        runs[dev_tf] = seen
    assert len(runs[True]) == len(runs[False]) == 3
    for a, b in zip(runs[True], runs[False]):
        for k in ("A", "B"):
            assert a[k].shape == (2, 1, 16, 16, 16) and torch.allclose(a[k], b[k], atol=2e-6, rtol=0), k
