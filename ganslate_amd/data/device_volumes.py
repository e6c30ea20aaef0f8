"""Device-side training-patch pipeline of the 3-D volume datasets: volumes resident in HBM, per
iteration only patch coordinates arrive from the loader (data/volume_datasets.py RawPatch), crop + z-score + range
scaling run as gs_patch_zscore (csrc/volproc.hip) and write straight into the fp32 [N, 1, d, h, w] batch `set_input`
takes. Reference path being replaced: projects/brats_mri_sequence_translation/datasets/train_dataset.py:83-92."""
import torch


class DeviceVolumePipeline:
    """callable(raw batch) -> {"A": fp32 [N, 1, *patch] on the device, "B": ...}"""

    def __init__(self, dataset, device, ops=None, max_resident_bytes=200 << 30):
        self.dataset = dataset
        self.device = torch.device(device)
        self._ops = ops
        self.size = [int(v) for v in dataset.patch_sampler.patch_size]       # (d, h, w); d = 1 for 2-D patch sizes
        self.squeeze = dataset.patch_sampler.dims == 2
        self.resident = {}
        self.resident_bytes, self.max_resident_bytes = 0, max_resident_bytes

    @property
    def ops(self):
        if self._ops is None:
            from ..nn.native.backend import get_ops
            self._ops = get_ops()
        return self._ops

    def volume(self, domain, index):
        key = (domain, index)
        v = self.resident.get(key)
        if v is None:
            host = self.dataset.load(domain, index)
            if host.dtype not in (torch.float32, torch.int16):
                host = host.float()
            v = host.to(self.device)
            nbytes = v.numel() * v.element_size()
            if self.resident_bytes + nbytes <= self.max_resident_bytes:      # beyond the budget: upload per use
                self.resident[key] = v
                self.resident_bytes += nbytes
        return v

    def __call__(self, batch):
        from .volume_datasets import RawPatch
        out = {}
        for key, items in batch.items():
            if not items or not isinstance(items[0], RawPatch):
                out[key] = items
                continue
            dst = torch.empty((len(items), 1, *self.size), dtype=torch.float32, device=self.device)
            for n, r in enumerate(items):
                self.ops.patch_zscore(self.volume(r.domain, r.index), r.start, self.size, dst[n, 0], (-1.0, 1.0))
            out[key] = dst.squeeze(2) if self.squeeze else dst
        return out
