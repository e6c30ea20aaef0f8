"""`_target_` plugin mechanism (ganslate/utils/builders.py:16-129): YAML -> config, config -> GAN / networks /
data loader. Two reference defects are resolved as SURVEY.md §2.4 prescribes (DDP batch-size key)."""
import copy

import torch
from torch.utils.data import DataLoader
from torch.utils.data.distributed import DistributedSampler

from ..configs.config import Config
from ..configs.omegalite import DictConfig, OmegaConf
from ..configs.utils import init_config
from ..nn.utils import init_net
from . import communication
from .io import import_attr


def build_conf(omegaconf_args):
    cli = OmegaConf.from_dotlist(omegaconf_args)
    assert "config" in cli, "Please provide path to a YAML config using `config` option."
    yaml_conf = cli.pop("config")
    conf = init_config(yaml_conf, config_class=Config)
    return OmegaConf.merge(conf, cli)


def build_loader(conf):
    from ..data.samplers import InfiniteSampler
    mode_conf = conf[conf.mode]
    if "multi_dataset" in mode_conf and mode_conf.multi_dataset is not None:
        assert mode_conf.dataset is None, "Use either `dataset` or `multi_dataset`."
        loaders = {}
        for name, dataset_conf in mode_conf.multi_dataset.items():
            current = copy.deepcopy(conf)
            current[conf.mode].dataset = dataset_conf
            current[conf.mode].multi_dataset = None
            loaders[name] = build_loader(current)
        return loaders
    dataset = import_attr(mode_conf.dataset._target_)(conf)
    if torch.distributed.is_initialized():
        # reference reads conf[mode].dataset.batch_size, a key that does not exist (builders.py:57)
        ddp_batch = communication.get_world_size() * mode_conf.batch_size
        if ddp_batch > len(dataset):
            raise RuntimeError(f"Dataset has {len(dataset)} examples, while the effective batch size equals to "
                               f"{ddp_batch}. Distributed mode does not work as expected in this situation.")
    if conf.mode == "train":
        sampler = InfiniteSampler(size=len(dataset), shuffle=True)
    else:
        sampler = None
        if torch.distributed.is_initialized():
            sampler = DistributedSampler(dataset, shuffle=False, num_replicas=communication.get_world_size(),
                                         rank=communication.get_rank())
    collate = getattr(dataset, "collate_fn", None)      # device_transforms: raw images of different sizes stay a list
    return DataLoader(dataset, sampler=sampler, batch_size=mode_conf.batch_size,
                      num_workers=mode_conf.dataset.num_workers,
                      pin_memory=mode_conf.dataset.pin_memory and collate is None, collate_fn=collate)


def build_gan(conf):
    return import_attr(conf.train.gan._target_)(conf)


def build_G(conf, direction, device):
    assert direction in ["AB", "BA"]
    return build_network_by_role("generator", conf, direction, device)


def build_D(conf, domain, device):
    assert domain in ["B", "A"]
    return build_network_by_role("discriminator", conf, domain, device)


def build_network_by_role(role, conf, label, device):
    assert role in ["discriminator", "generator"]
    node = conf.train.gan[role]
    network_class = import_attr(node._target_)
    args = {k: node[k] for k in node.keys()}
    args.pop("_target_")
    args["norm_type"] = conf.train.gan.norm_type
    if role == "generator":
        ioc = args.pop("in_out_channels")
        if isinstance(ioc, DictConfig):
            ioc = ioc[label]
        args["in_channels"], args["out_channels"] = ioc
    else:
        if isinstance(args["in_channels"], DictConfig):
            args["in_channels"] = args["in_channels"][label]
    network = network_class(**args)
    return init_net(network, conf, device)
