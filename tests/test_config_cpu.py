"""Config engine / plugin mechanism: YAML -> structured config with `<_target_>Config` defaults, interpolation,
MISSING, dotlist overrides; `ganslate.*` targets resolve to this package."""
from pathlib import Path

import pytest

from ganslate_amd.configs.omegalite import MissingMandatoryValue
from ganslate_amd.utils.builders import build_conf
from ganslate_amd.utils.io import import_attr

CONF = Path(__file__).parent / "configs" / "cyclegan_synthetic.yaml"


def test_yaml_defaults_overrides_and_interpolation():
    conf = build_conf([f"config={CONF}", "train.batch_size=8", "train.dataset.final_size=[256,256]"])
    assert conf.mode == "train" and conf.train.batch_size == 8
    assert conf.train.dataset.final_size == [256, 256]
    gan = conf.train.gan
    assert gan.norm_type == "instance" and gan.weight_init_gain == 0.02 and gan.pool_size == 50   # dataclass defaults
    assert gan.optimizer.beta1 == 0.5 and gan.optimizer.adversarial_loss_type == "lsgan"
    assert gan.generator.in_out_channels.BA == [3, 3]        # II("...in_out_channels.AB")
    assert gan.discriminator.in_channels.A == 3 and gan.discriminator.ndf == 64
    assert conf["train"].checkpointing.load_iter is None
    with pytest.raises(AttributeError):
        gan.generator.in_channels                         # the key cut.py:83 reads does not exist (struct mode)


def test_missing_mandatory_value_raises():
    conf = build_conf([f"config={CONF}"])
    conf.train.output_dir = "???"
    with pytest.raises(MissingMandatoryValue):
        conf.train.output_dir


def test_reference_targets_resolve_to_this_package():
    for target in ("ganslate.nn.gans.unpaired.CycleGAN", "ganslate.nn.generators.Resnet2D",
                   "ganslate.nn.discriminators.PatchGAN2D", "ganslate.data.UnpairedImageDataset",
                   "ganslate.data.UnpairedImageDatasetConfig", "ganslate.nn.generators.Resnet2DConfig"):
        obj = import_attr(target)
        assert obj.__module__.startswith("ganslate_amd.")


@pytest.mark.skipif(not Path("/root/reference/projects/horse2zebra/experiments/default.yaml").is_file(),
                    reason="reference checkout not present (GPU box)")
def test_reference_project_yaml_loads_unchanged():
    conf = build_conf(["config=/root/reference/projects/horse2zebra/experiments/default.yaml",
                       "train.batch_size=8", "train.dataset.load_size=[256,256]",
                       "train.dataset.final_size=[256,256]"])
    assert conf.train.gan._target_ == "ganslate.nn.gans.unpaired.CycleGAN"
    assert conf.train.gan.optimizer.lambda_AB == 10.0 and conf.train.gan.optimizer.proportion_ssim == 0
    assert conf.train.dataset.preprocess == ["resize", "random_flip"] and conf.train.batch_size == 8
    assert conf.train.metrics.ssim is True and conf.infer.dataset.num_workers == 16
