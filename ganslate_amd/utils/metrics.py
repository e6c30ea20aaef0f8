"""Training-time metrics computed inside optimize_parameters (ganslate/utils/metrics/train_metrics.py:5-67):
mean discriminator outputs and SSIM between real and cycle-reconstructed images, both on fused kernels."""
import torch

from ..nn.losses.functional import mean_nograd, scalar_affine, ssim_distance_nograd


class TrainingMetrics:

    def __init__(self, conf):
        self.output_distributions = bool(conf.train.metrics.discriminator_evolution)
        self.ssim = bool(conf.train.metrics.ssim)

    def get_output_metric_D(self, out):
        if not self.output_distributions:
            return None
        if isinstance(out, dict):
            # multi-scale discriminators: the reference builds the per-scale means and then falls off the end of its `if`
            # (train_metrics.py:22-25: no return on this branch), so the metric is None and the tracker drops it. Kept.
            return None
        return mean_nograd(out) if out.dim() > 1 else out.detach()

    def get_SSIM_metric(self, input, target):
        if not self.ssim:
            return None
        return scalar_affine([ssim_distance_nograd(input, target)], [[-1.0]], [1.0])[0]      # 1 - distance, one launch

    def compute_metrics_D(self, discriminator, pred_real, pred_fake):
        return {f"{discriminator}_real": self.get_output_metric_D(pred_real),
                f"{discriminator}_fake": self.get_output_metric_D(pred_fake)}

    def compute_metrics_G(self, visuals):
        m = {}
        if all(k in visuals for k in ("rec_A", "real_A")):
            m["ssim_A"] = self.get_SSIM_metric(visuals["real_A"], visuals["rec_A"])
        if all(k in visuals for k in ("rec_B", "real_B")):
            m["ssim_B"] = self.get_SSIM_metric(visuals["real_B"], visuals["rec_B"])
        return m
