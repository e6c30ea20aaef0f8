"""Device input pipeline on the GPU (csrc/imgproc.hip): both kernels against the op-level oracle on the
same tables, the pipeline against the PIL + torch host transform of the datasets bit for bit (integer resampler, and the
fp32 ToTensor / Normalize in torchvision's operation order), and a Trainer run fed through it."""
import random
import time

import numpy as np
import pytest
import torch
from PIL import Image

from ganslate_amd.data.device_transforms import DeviceImagePipeline, RawImage, resample_tables
from oracle import pil_ref
from oracle.ops_ref import RefOps

pytestmark = pytest.mark.gpu


class D(dict):
    __getattr__ = dict.__getitem__


def _conf(pre, load, final, **kw):
    return D(mode="train", train=D(dataset=D(preprocess=list(pre), load_size=list(load), final_size=list(final), **kw)))


def _image(h, w, c, seed):
    return np.random.default_rng(seed).integers(0, 256, (h, w, c), dtype=np.uint8)


@pytest.mark.parametrize("h,w,oh,ow,c", [(256, 256, 286, 286, 3), (300, 400, 286, 286, 3), (37, 53, 64, 64, 1),
                                          (1024, 768, 286, 286, 3), (64, 64, 64, 80, 3)])
def test_kernels_equal_the_oracle_pass_by_pass(hip_ops, h, w, oh, ow, c):
    img = torch.from_numpy(_image(h, w, c, 2))
    bh, kh = (torch.from_numpy(t) for t in resample_tables(w, ow))
    bv, kv = (torch.from_numpy(t) for t in resample_tables(h, oh))
    fh, fw = min(oh, 40), min(ow, 48)
    top, left = oh - fh, (ow - fw) // 2
    outs = []
    for ops, dev in ((RefOps(), "cpu"), (hip_ops, hip_ops.device)):
        tmp = torch.empty((h, ow, c), dtype=torch.uint8, device=dev)
        ops.u8_resample_h(img.to(dev), tmp, bh.to(dev), kh.contiguous().to(dev))
        res = []
        for flip in (False, True):
            out = torch.empty((c, fh, fw), dtype=torch.float32, device=dev)
            ops.u8_resample_v_crop_normalize(tmp, out, oh, bv.to(dev), kv.contiguous().to(dev), top, left, flip)
            res.append(out.cpu())
        outs.append((tmp.cpu(), res))
        mid = torch.empty((oh, ow, c), dtype=torch.uint8, device=dev)
        ops.u8_resample_v(tmp, mid, bv.to(dev), kv.contiguous().to(dev))
        outs[-1] += (mid.cpu(),)
    assert torch.equal(outs[0][0], outs[1][0]), "horizontal pass"
    assert torch.equal(outs[0][2], outs[1][2]), "vertical pass, 8-bit"
    ref = pil_ref.resize_bicubic(img.numpy(), oh, ow)
    assert np.array_equal(outs[1][2].numpy(), ref), "two passes == Pillow's resize"
    for a, b in zip(outs[0][1], outs[1][1]):
        assert torch.equal(a, b), "vertical pass + crop + flip + normalise"


@pytest.mark.parametrize("pre", [("resize", "random_crop", "random_flip"), ("resize",), ("random_crop", "random_flip"),
                                 ("scale_width", "random_crop"), ("resize", "random_zoom", "random_crop", "random_flip"),
                                 ("scale_width", "random_zoom", "random_crop", "random_flip")])
@pytest.mark.parametrize("c", [3, 1])
def test_pipeline_equals_the_host_transform_bit_for_bit(hip_ops, pre, c):
    from ganslate_amd.data.image_datasets import _Transform
    conf = _conf(pre, (72, 80), (64, 64))
    host = _Transform(conf)
    pipe = DeviceImagePipeline(conf, hip_ops.device, ops=hip_ops)
    random.seed(4)
    sizes = [(64, 64)] * 4 if "resize" not in pre and "random_crop" not in pre else [(90, 100), (75, 130), (64, 64), (200, 81)]
    raws, want = [], []
    for k, (h, w) in enumerate(sizes):
        a = _image(h, w, c, 10 + k)
        prm = host.params()
        pil = Image.fromarray(a if c == 3 else a[..., 0], "RGB" if c == 3 else "L")
        want.append(host(pil, prm))
        raws.append(RawImage(torch.from_numpy(a if c == 3 else a[..., 0].copy()), prm["crop"], prm["flip"], prm["zoom"]))
    if len({tuple(t.shape) for t in want}) > 1:
        pytest.skip("without resize or crop the images of a batch keep their own sizes")
    got = pipe({"A": raws})["A"]
    torch.cuda.synchronize()
    assert got.shape == (len(sizes), c) + tuple(want[0].shape[1:])
    for n, t in enumerate(want):
        assert torch.equal(got[n].cpu(), t), n


def test_pipeline_rate_at_the_headline_shape(hip_ops):
    """286x286 bicubic resize, 256x256 crop, flip, normalise of 256x256 RGB images (horse2zebra): images per second through
    the two kernels with the bytes already on the device (the transform itself, not PCIe) — well above the step rate"""
    conf = _conf(("resize", "random_crop", "random_flip"), (286, 286), (256, 256))
    pipe = DeviceImagePipeline(conf, hip_ops.device, ops=hip_ops)
    raws = [RawImage(torch.from_numpy(_image(256, 256, 3, k)).to(hip_ops.device), (0.3, 0.6), k % 2 == 0) for k in range(16)]
    pipe({"A": raws})
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        pipe({"A": raws})
    torch.cuda.synchronize()
    rate = 20 * len(raws) / (time.perf_counter() - t0)
    print(f"\ndevice transform: {rate:.0f} img/s")
    assert rate > 2000


def test_trainer_runs_on_the_device_pipeline(hip_ops, tmp_path):
    """image folder -> decode-only workers -> DeviceImagePipeline -> CycleGAN.set_input: three iterations, finite losses,
    and the batch the model saw equals the host transform of the same files with the same draws"""
    from ganslate_amd.engines import init_engine
    root = tmp_path / "data"
    for dom in "AB":
        (root / dom).mkdir(parents=True)
        for k in range(4):
            Image.fromarray(_image(40 + 3 * k, 52, 3, ord(dom) + k), "RGB").save(root / dom / f"{k}.png")
    args = ["config=tests/configs/cyclegan_imagefolder.yaml", f"train.output_dir={tmp_path / 'out'}",
            f"train.dataset.root={root}", "train.seed=5"]
    runs = {}
    for dev_tf in (True, False):
        trainer = init_engine("train", args + [f"train.dataset.device_transforms={dev_tf}"])
        seen = []
        orig = trainer.model.set_input
        trainer.model.set_input = lambda data, _o=orig, _s=seen: (_s.append({k: v.detach().float().cpu().clone()
                                                                           for k, v in data.items()}), _o(data))[1]
        trainer.run()
        assert trainer.input_pipeline is not None if dev_tf else trainer.input_pipeline is None
        assert all(float(v) == float(v) for v in trainer.model.losses.values() if v is not None)
        runs[dev_tf] = seen
    assert len(runs[True]) == len(runs[False]) == 3
    for a, b in zip(runs[True], runs[False]):
        for k in ("A", "B"):
            assert torch.equal(a[k], b[k]), k
