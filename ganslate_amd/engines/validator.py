"""Validator — the in-training validation pass (ganslate/engines/validator_tester.py:9-60,100-112 +
trainer.py:95-108): shares the Trainer's model, runs `infer` (sliding-window when configured) over the validation
loader(s) and reports the scalar metrics of utils/metrics/val_test_metrics.py that need no third-party package
(mae, mse, nmse, psnr; ssim / nmi / histogram_chi2 rely on scikit-image / scipy in the reference and are skipped with a
log line). Saving generated tensors and the W&B / TensorBoard trackers stay out of scope."""
import numpy as np
import torch

from ..utils.builders import build_loader
from .base import BaseEngineWithInference


def _mae(gt, pred):
    return float(np.mean(np.abs(gt - pred)))


def _mse(gt, pred):
    return float(np.mean((gt - pred) ** 2))


def _nmse(gt, pred):
    return float(np.linalg.norm(gt - pred) ** 2 / np.linalg.norm(gt) ** 2)


def _psnr(gt, pred):
    # skimage.metrics.peak_signal_noise_ratio(gt, pred, data_range=gt.max()) (val_test_metrics.py:56-59)
    err = np.mean((gt.astype(np.float64) - pred.astype(np.float64)) ** 2)
    return float(10 * np.log10((float(gt.max()) ** 2) / err))


METRICS = {"mae": _mae, "mse": _mse, "nmse": _nmse, "psnr": _psnr}


class Validator(BaseEngineWithInference):

    def __init__(self, conf, model):
        super().__init__(conf)
        self.model = model
        self.data_loaders = build_loader(self.conf)
        if not isinstance(self.data_loaders, dict):
            self.data_loaders = {None: self.data_loaders}
        self.history = []          # (iteration, dataset name, {metric: mean over the samples})
        wanted = self.conf.val.metrics
        self.metric_names = [k for k in METRICS if getattr(wanted, k, False)]
        skipped = [k for k in ("ssim", "nmi", "histogram_chi2") if getattr(wanted, k, False)]
        if skipped:
            self.logger.info(f"validation metrics {skipped} need scikit-image / scipy in the reference; skipped here")

    def _set_mode(self):
        self.conf.mode = "val"

    def run(self, current_idx=None):
        self.logger.info("Validation started.")
        was_training = [getattr(net, "training", True) for net in self.model.networks.values()]
        self.model.eval()
        try:
            for name, loader in self.data_loaders.items():
                rows = []
                dataset = loader.dataset
                if getattr(getattr(dataset, "conf", None), "device_transforms", False) or \
                        getattr(dataset, "device_transforms", False):
                    raise NotImplementedError("validation datasets run the host transform path: set "
                                              "`val.dataset.device_transforms: false` (the device-side pipeline batches "
                                              "training samples only)")
                # Denormalize the data if the dataset defines `denormalize` (validator_tester.py:72-77)
                denormalize = getattr(dataset, "denormalize", None)
                over_input = bool(getattr(self.conf.val.metrics, "compute_over_input", False))
                for data in loader:
                    real_A = data["A"].to(self.model.device)
                    with torch.no_grad():
                        fake_B = self.infer(real_A)
                    pred, target, original = fake_B.detach().float().cpu(), data["B"].float(), data["A"].float()
                    if denormalize:
                        pred, target = denormalize(pred.clone()), denormalize(target.clone())
                        if over_input:
                            original = denormalize(original.clone())
                    pred, target, original = pred.numpy(), target.numpy(), original.numpy()
                    # one score per SAMPLE of the batch (ValTestMetrics.get_metrics iterates zip(inputs, targets),
                    # val_test_metrics.py:152-153): psnr's data range and nmse's norm are per sample
                    for i in range(pred.shape[0]):
                        row = {k: METRICS[k](target[i], pred[i]) for k in self.metric_names}
                        if over_input:
                            row.update({f"Original_{k}": METRICS[k](target[i], original[i]) for k in self.metric_names})
                        rows.append(row)
                mean = {k: float(np.mean([r[k] for r in rows])) for k in rows[0]} if rows else {}
                self.history.append((current_idx, name, mean))
                self.logger.info(f"val @ {current_idx} [{name}] {mean}")
        finally:
            for net, flag in zip(self.model.networks.values(), was_training):
                net.train(flag)
