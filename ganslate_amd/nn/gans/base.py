"""BaseGAN — the plugin boundary the Trainer talks to (ganslate/nn/gans/base.py:16-321): public dicts
`networks / optimizers / losses / visuals / metrics`, abstract `set_input / forward / optimize_parameters /
init_criterions / init_optimizers`, concrete `setup / backward / parallelize_networks / save_checkpoint /
load_networks / set_requires_grad / infer / get_loggable_data / update_learning_rate`.

Differences that are deliberate: networks are HIP executors (NativeNet) instead of nn.Modules;
`train.mixed_precision` selects nothing — activations are always bf16 with fp32 master weights/statistics/
losses (apex AMP O1 fp16, base.py:118-126, has no MI355X counterpart); data parallelism is the executor's own
bucketed all-reduce over RCCL instead of DistributedDataParallel (base.py:172-189)."""
import logging
import os
from abc import ABC, abstractmethod
from pathlib import Path

import torch

from ...utils import communication, io
from ...utils.builders import build_D, build_G
from ...utils.metrics import TrainingMetrics
from ..utils import get_scheduler


class BaseGAN(ABC):

    def __init__(self, conf):
        self.logger = logging.getLogger("ganslate_amd")
        self.conf = conf
        self.is_train = self.conf.mode == "train"
        self.device = self._specify_device()
        self.output_dir = conf[conf.mode].output_dir
        self.visuals, self.metrics, self.losses, self.optimizers, self.networks = {}, {}, {}, {}, {}

    def init_networks(self):
        for name in self.networks.keys():
            if name.startswith("G"):
                direction = "BA" if name.endswith("_BA") else "AB"
                self.networks[name] = build_G(self.conf, direction, self.device)
            elif name.startswith("D"):
                domain = "A" if name.endswith("_A") else "B"
                self.networks[name] = build_D(self.conf, domain, self.device)

    @abstractmethod
    def init_criterions(self):
        """Initialize criterions (losses)"""

    @abstractmethod
    def init_optimizers(self):
        """Initialize optimizers"""

    def init_metrics(self):
        self.training_metrics = TrainingMetrics(self.conf)

    def init_schedulers(self):
        self.schedulers = [get_scheduler(optim, self.conf) for optim in self.optimizers.values()]

    def _specify_device(self):
        from ..native.backend import get_ops
        if torch.distributed.is_initialized() and torch.distributed.get_backend() == "nccl":
            dev = torch.device(f"cuda:{communication.get_local_rank()}")
            torch.cuda.set_device(dev)
        else:
            dev = get_ops().device          # raises loudly when neither the GPU nor the library is there
        if not self.conf[self.conf.mode].cuda and dev.type == "cuda":
            self.logger.warning("`cuda: False` requested, but this build only has the MI355X path; running on %s", dev)
        return get_ops().device

    @abstractmethod
    def set_input(self, input):
        """Unpack input data from the dataloader."""

    @abstractmethod
    def forward(self):
        """Run forward pass."""

    @abstractmethod
    def optimize_parameters(self):
        """Calculate losses, gradients, and update network weights; called in every training iteration"""

    def setup(self):
        assert "G" in self.networks or "G_AB" in self.networks, "The (main) generator has to be named `G` or `G_AB`."
        if self.conf[self.conf.mode].mixed_precision:
            self.logger.info("mixed_precision: bf16 activations / fp32 master weights are always on in this build; "
                             "apex opt_level is ignored")
        self.init_networks()
        if self.is_train:
            self.init_criterions()
            self.init_optimizers()
            self.init_metrics()
            self.init_schedulers()
        else:
            self.eval()
            if len(self.networks.keys()) != 1:
                raise ValueError("When inferring there should be only one network initialized - generator.")
        if self.conf[self.conf.mode].checkpointing.load_iter:
            self.load_networks(self.conf[self.conf.mode].checkpointing.load_iter)
        if int(os.environ.get("WORLD_SIZE", 1)) > 1:
            self.parallelize_networks()

    def backward(self, loss, optimizer, retain_graph=False, loss_id=0):
        loss.backward(retain_graph=retain_graph)

    def parallelize_networks(self):
        if not torch.distributed.is_initialized():
            raise RuntimeError("Multi-GPU runs must be launched in distributed mode (torchrun / "
                               "torch.distributed.run), one process per GPU.")
        for name in self.networks.keys():
            self.networks[name].parallelize()

    def update_learning_rate(self):
        for scheduler in self.schedulers:
            scheduler.step()

    def save_checkpoint(self, iter_idx):
        checkpoint = {}
        path = Path(self.output_dir) / f"checkpoints/{iter_idx}.pth"
        io.mkdirs(path.parent)
        for name, net in self.networks.items():
            checkpoint[name] = {k: v.cpu() for k, v in net.state_dict().items()}
        for name, optim in self.optimizers.items():      # reference saves only G and D (base.py:244-245)
            checkpoint[f"optimizer_{name}"] = optim.state_dict()
        torch.save(checkpoint, path)

    def load_networks(self, iter_idx):
        path = Path(self.output_dir).resolve() / f"checkpoints/{iter_idx}.pth"
        checkpoint = torch.load(path, map_location="cpu")
        self.logger.info(f"Loaded the checkpoint from `{path}`")
        for name in self.networks.keys():
            self.networks[name].load_state_dict(checkpoint[name])
        if self.is_train and self.conf[self.conf.mode].checkpointing.load_optimizers:
            for name, optim in self.optimizers.items():
                key = f"optimizer_{name}"
                if key in checkpoint and _is_native_optimizer_state(checkpoint[key], optim):
                    optim.load_state_dict(checkpoint[key])
                else:
                    self.logger.info(f"{key}: no compatible state in the checkpoint; starting from scratch")

    def set_requires_grad(self, networks, requires_grad=False):
        if not isinstance(networks, list):
            networks = [networks]
        for net in networks:
            if net is not None:
                for param in net.parameters():
                    param.requires_grad = requires_grad

    def eval(self):
        for name in self.networks.keys():
            self.networks[name].eval()

    def infer(self, input):
        generator = "G" if "G" in self.networks.keys() else "G_AB"
        with torch.no_grad():
            return self.networks[generator].forward(input)

    def get_loggable_data(self):
        learning_rates = {f"lr_{name}": optim.param_groups[0]["lr"] for name, optim in self.optimizers.items()}
        return learning_rates, self.losses, self.visuals, self.metrics


def _is_native_optimizer_state(state, optim):
    try:
        n_saved = sum(len(g["params"]) for g in state["param_groups"])
        n_here = sum(len(g["params"]) for g in optim.param_groups)
        return n_saved == n_here
    except (KeyError, TypeError):
        return False
