"""3-D training-patch path (second half) on the host: the sampler's draws and the normalisation against
vectors recorded from the reference's own functions (oracle/gen_golden_r2.py volpatch -> tests/golden/volume_patches.json)."""
import json
import random
from pathlib import Path

import numpy as np
import pytest
import torch

from ganslate_amd.data.utils.normalization import (min_max_denormalize, min_max_normalize, z_score_normalize,
                                                   z_score_normalize_with_precomputed_stats)
from ganslate_amd.data.utils.stochastic_focal_patching import StochasticFocalPatchSampler

GOLD = json.loads((Path(__file__).parent / "golden" / "volume_patches.json").read_text())["cases"]


def seeded_volume(shape, seed):
    """same construction as oracle/gen_golden_r2.py::_seeded_volume (numpy Generator streams are version-stable)"""
    rng = np.random.default_rng(seed)
    v = rng.gamma(2.0, 180.0, size=shape).astype(np.float32)
    v[: max(1, shape[0] // 5)] = 0.0
    return torch.from_numpy(np.round(v))


@pytest.mark.parametrize("name", sorted(GOLD))
def test_sampler_draws_the_reference_patches(name):
    c = GOLD[name]
    A, B = seeded_volume(c["shape_A"], c["seed"]), seeded_volume(c["shape_B"], c["seed"] + 100)
    sampler = StochasticFocalPatchSampler(np.array(c["patch_size"]), c["focal_region_proportion"])
    random.seed(c["seed"])
    for rec in c["draws"]:
        state = random.getstate()
        sa, sb = sampler.get_start_pair(A.shape, B.shape)
        assert list(sa) == rec["start_A"] and list(sb) == rec["start_B"]
        after = random.getstate()
        random.setstate(state)
        pa, pb = sampler.get_patch_pair(A, B)            # the reference's entry point consumes the same draws
        assert random.getstate() == after
        assert list(pa.shape) == rec["shape"] and list(pb.shape) == rec["shape"]
        assert torch.equal(pa, sampler.crop(A, sa)) and torch.equal(pb, sampler.crop(B, sb))


@pytest.mark.parametrize("name", sorted(GOLD))
def test_normalisation_matches_the_reference(name):
    c = GOLD[name]
    A, B = seeded_volume(c["shape_A"], c["seed"]), seeded_volume(c["shape_B"], c["seed"] + 100)
    sampler = StochasticFocalPatchSampler(np.array(c["patch_size"]), c["focal_region_proportion"])
    rec = c["draws"][0]
    for key, vol, start in (("A", A, rec["start_A"]), ("B", B, rec["start_B"])):
        patch = sampler.crop(vol, start)
        g = rec["z_" + key]
        assert float(patch.mean()) == pytest.approx(g["patch_mean"], rel=1e-6)
        assert float(patch.std()) == pytest.approx(g["patch_std"], rel=1e-6)
        # (patch_2d / B lies in the zero slab: a constant patch, NaN on both sides — the reference's behaviour, kept)
        z = z_score_normalize(patch.clone(), scale_to_range=(-1, 1)).flatten()
        idx = torch.tensor(g["samples_at"])
        assert torch.allclose(z[idx], torch.tensor(g["samples"]), atol=2e-6, rtol=0, equal_nan=True)
        assert float(z.min()) == pytest.approx(g["min"], abs=1e-6, nan_ok=True)
        assert float(z.max()) == pytest.approx(g["max"], abs=1e-6, nan_ok=True)
        assert float(z.double().mean()) == pytest.approx(g["mean"], abs=1e-6, nan_ok=True)
        assert float((z.double() ** 2).mean()) == pytest.approx(g["sq"], abs=1e-6, nan_ok=True)
        plain = z_score_normalize(patch.clone()).flatten()
        assert torch.allclose(plain[idx], torch.tensor(rec["z_plain_" + key]), atol=2e-6, rtol=1e-6, equal_nan=True)
    pa = sampler.crop(A, rec["start_A"])
    idx = torch.tensor(rec["z_A"]["samples_at"])
    mm = min_max_normalize(pa.clone(), 0.0, 1500.0)
    assert torch.allclose(mm.flatten()[idx], torch.tensor(rec["minmax_A"]), atol=1e-6, rtol=0)
    assert torch.allclose(min_max_denormalize(mm.clone(), 0.0, 1500.0), pa, atol=2e-4, rtol=0)
    pre = z_score_normalize_with_precomputed_stats(pa.clone(), (210.0, 95.0), original_scale=(0.0, 1800.0),
                                                   scale_to_range=(-1, 1))
    assert torch.allclose(pre.flatten()[idx], torch.tensor(rec["precomputed_A"]), atol=1e-6, rtol=0)


def test_sampler_rejects_volumes_smaller_than_the_patch():
    s = StochasticFocalPatchSampler(np.array([8, 16, 16]), 0.0)
    with pytest.raises(RuntimeError, match="smaller than the patch size"):
        s.get_start_pair((7, 16, 16), (8, 16, 16))
    with pytest.raises(RuntimeError, match="smaller than the patch size"):
        s.get_start_pair((8, 16, 16), (8, 15, 16))


def test_constant_patch_normalises_to_nan_like_the_reference():
    z = z_score_normalize(torch.full((4, 4, 4), 3.0), scale_to_range=(-1, 1))
    assert torch.isnan(z).all()


class _D(dict):
    __getattr__ = dict.__getitem__


def _volume_folder(tmp_path, dtype=np.float32):
    shapes = {"A": [(20, 36, 30), (18, 40, 33)], "B": [(24, 33, 31), (19, 34, 38), (22, 30, 30)]}
    for dom, ss in shapes.items():
        (tmp_path / dom).mkdir()
        for k, s in enumerate(ss):
            np.save(tmp_path / dom / f"vol{k}.npy", seeded_volume(s, 40 + k + (10 if dom == "B" else 0)).numpy().astype(dtype))
    return tmp_path


def _conf(root, device_transforms, patch=(8, 16, 16), frp=0.3):
    return _D(mode="train", train=_D(dataset=_D(root=str(root), patch_size=list(patch), focal_region_proportion=frp,
                                              device_transforms=device_transforms)))


@pytest.mark.parametrize("dtype", [np.float32, np.int16])
@pytest.mark.parametrize("patch", [(8, 16, 16), (24, 24)])
def test_device_pipeline_equals_the_host_dataset_path(tmp_path, dtype, patch):
    """device_transforms on the oracle backend: a worker hands over coordinates (same `random` draws as the host path),
    DeviceVolumePipeline crops + normalises the resident volume — the batch equals the host dataset's"""
    from ganslate_amd.data.device_volumes import DeviceVolumePipeline
    from ganslate_amd.data.volume_datasets import RawPatch, UnpairedVolumeDataset, collate_raw
    from oracle.ops_ref import RefOps
    root = _volume_folder(tmp_path, dtype)
    host = UnpairedVolumeDataset(_conf(root, False, patch))
    raw = UnpairedVolumeDataset(_conf(root, True, patch))
    assert host.collate_fn is None and raw.collate_fn is collate_raw
    pipe = DeviceVolumePipeline(raw, "cpu", ops=RefOps())
    random.seed(12)
    want = [host[i] for i in range(4)]
    state_after = random.getstate()
    random.seed(12)
    samples = [raw[i] for i in range(4)]
    assert random.getstate() == state_after                      # both paths consume the same draws
    assert all(isinstance(s["A"], RawPatch) for s in samples)
    got = pipe(collate_raw(samples))
    for key in "AB":
        ref = torch.stack([w[key] for w in want])
        assert got[key].shape == ref.shape and got[key].dtype == torch.float32
        assert torch.allclose(got[key], ref, atol=1e-6, rtol=0, equal_nan=True)     # (a slice of the zero slab is NaN on both sides)
    assert len(pipe.resident) <= 5 and pipe.resident_bytes > 0   # volumes stay resident between iterations
    before = dict(pipe.resident)
    pipe(collate_raw(samples))
    assert all(pipe.resident[k] is v for k, v in before.items())


def test_volume_dataset_through_the_builders(tmp_path):
    """YAML surface: `_target_: ganslate.data.UnpairedVolumeDataset` with BratsDatasetConfig's field names"""
    from ganslate_amd.utils.builders import build_conf, build_loader
    root = _volume_folder(tmp_path)
    conf = build_conf(["config=tests/configs/cyclegan3d_volumefolder.yaml", f"train.dataset.root={root}"])
    conf.mode = "train"
    batch = next(iter(build_loader(conf)))
    assert batch["A"].shape == (2, 1, 8, 16, 16) and batch["B"].shape == (2, 1, 8, 16, 16)
    assert float(batch["A"].min()) == pytest.approx(-1.0, abs=1e-6) and float(batch["A"].max()) == pytest.approx(1.0, abs=1e-6)
