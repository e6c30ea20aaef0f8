"""Trainer — the caller of the hot path (ganslate/engines/trainer.py:11-112): constructor order (seed -> loader
-> model -> iteration range), per-iteration sequence set_input -> optimize_parameters -> get_loggable_data ->
[log] -> [checkpoint] -> update_learning_rate, `iters = range(1 + load_iter, 1 + n_iters + n_iters_decay)`,
rank-0 checkpoint cadence. Timers are rank-local and synchronise the device only at logging time — the
reference's two per-iteration timer reduces + `.item()` (trackers/base.py:56,61) are not reproduced."""
import logging
import time

import torch

from ..utils import communication, environment
from ..utils.builders import build_gan, build_loader


class Trainer:

    def __init__(self, conf):
        self.conf = conf
        self.conf.mode = "train"
        self.logger = logging.getLogger("ganslate_amd")
        environment.setup_logging()
        if self.conf.train.seed:
            environment.set_seed(self.conf.train.seed)
        self.data_loader = build_loader(self.conf)
        self.model = build_gan(self.conf)
        # device-side input pipeline (train.dataset.device_transforms, data/device_transforms.py): the loader hands over
        # decoded bytes, the transform of data/utils/transforms.py runs on the GPU in front of set_input
        make = getattr(getattr(self.data_loader, "dataset", None), "device_pipeline", None)
        self.input_pipeline = make(self.conf, self.model.device) if make else None
        start_iter = 1
        if self.conf.train.checkpointing.load_iter:
            start_iter += self.conf.train.checkpointing.load_iter
        end_iter = 1 + self.conf.train.n_iters + self.conf.train.n_iters_decay
        assert start_iter < end_iter, "If continuing, define the `n_iters` relative to the loaded iteration."
        self.iters = range(start_iter, end_iter)
        self.iter_idx = 0
        self.history = []           # (iter, losses, metrics) captured at logging time
        self._t_comp, self._n_comp = 0.0, 0
        self.validator = self._init_validator()

    def _init_validator(self):
        """validation engine built from the training conf (trainer.py:95-101); None without a `val` section"""
        if not getattr(self.conf, "val", None):
            return None
        from .validator import Validator
        return Validator(self.conf, self.model)

    def _run_validation(self):
        if self.validator:
            if self.iter_idx % self.conf.val.freq == 0 and self.iter_idx >= self.conf.val.start_after:
                self.validator.run(current_idx=self.iter_idx)

    def run(self):
        self.logger.info("Training started.")
        for i, data in zip(self.iters, self.data_loader):
            self.iter_idx = i
            t0 = time.perf_counter()
            self._run_iteration(data)
            self._t_comp += time.perf_counter() - t0
            self._n_comp += 1
            learning_rates, losses, visuals, metrics = self.model.get_loggable_data()
            self._log_iter(learning_rates, losses, metrics)
            self._save_checkpoint()
            self._perform_scheduler_step()
            self._run_validation()

    def _run_iteration(self, data):
        if self.input_pipeline is not None:
            data = self.input_pipeline(data)
        self.model.set_input(data)
        self.model.optimize_parameters()

    def _perform_scheduler_step(self):
        self.model.update_learning_rate()

    def _log_iter(self, learning_rates, losses, metrics):
        if self.iter_idx % self.conf.train.logging.freq != 0:
            return
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        losses = communication.reduce_dict({k: v.detach() for k, v in losses.items() if v is not None})
        metrics = communication.reduce_dict({k: v.detach() for k, v in metrics.items() if v is not None})
        if communication.get_rank() == 0:
            lo = {k: float(v) for k, v in losses.items()}
            me = {k: float(v) for k, v in metrics.items()}
            t = self._t_comp / max(self._n_comp, 1) / self.conf.train.batch_size
            self.history.append((self.iter_idx, lo, me))
            self.logger.info(f"iter {self.iter_idx} | comp {t:.4f} s/img | {learning_rates} | {lo} | {me}")

    def _save_checkpoint(self):
        if communication.get_rank() == 0:
            freq = self.conf.train.checkpointing.freq
            after = self.conf.train.checkpointing.start_after
            if self.iter_idx % freq == 0 and self.iter_idx >= after:
                self.logger.info(f"Saving the model after {self.iter_idx} iterations.")
                self.model.save_checkpoint(self.iter_idx)
