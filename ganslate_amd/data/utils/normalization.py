"""Intensity normalisation of medical volumes, host side — ganslate/data/utils/normalization.py:4-57 (same function
names and argument meaning). The training path of the 3-D datasets is `z_score_normalize(patch, scale_to_range=(-1, 1))`
(projects/brats_mri_sequence_translation/datasets/train_dataset.py:85-86); its device twin is `gs_patch_zscore`
(csrc/volproc.hip) behind data/device_volumes.py."""
import torch


def min_max_normalize(image, min_value, max_value):
    """[min_value, max_value] -> [-1, 1]"""
    unit = (image.float() - min_value) / (max_value - min_value)
    return 2 * unit - 1


def min_max_denormalize(image, min_value, max_value):
    """inverse of min_max_normalize, IN PLACE like the reference (normalization.py:10-15)"""
    image += 1
    image /= 2
    image *= (max_value - min_value)
    image += min_value
    return image


def _rescale(t, t_lo, t_hi, scale_to_range):
    span = scale_to_range[1] - scale_to_range[0]
    return (span * (t - t_lo) / (t_hi - t_lo)) + scale_to_range[0]


def z_score_normalize(tensor, scale_to_range=None):
    """(x - mean) / std with the tensor's own mean and UNBIASED std (torch.std default), then optionally the affine map
    that sends the result's [min, max] onto scale_to_range. A constant tensor gives NaN, as in the reference."""
    t = (tensor - tensor.mean()) / tensor.std()
    if scale_to_range:
        t = _rescale(t, t.min(), t.max(), scale_to_range)
    return t


def z_score_normalize_with_precomputed_stats(tensor, mean_std, original_scale=None, scale_to_range=None):
    """z-score with given (mean, std) — e.g. a slice normalised with its volume's statistics; with scale_to_range the
    volume's (min, max) = original_scale, normalised the same way, are what maps onto the range"""
    mean, std = mean_std[0], mean_std[1]
    t = (tensor - mean) / std
    if scale_to_range:
        lo_hi = (torch.Tensor(original_scale) - mean) / std
        t = _rescale(t, lo_hi[0], lo_hi[1], scale_to_range)
    return t
