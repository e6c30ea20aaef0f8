#!/usr/bin/env python
"""Every convolution launch of one CycleGAN training step (bench.py's headline workload), labelled by its lowered
class and timed with HIP events on its stream: launches per step, average duration, TFLOP/s, share of the step's
conv time. Launch-by-launch on one stream (what rocprofv3's kernel trace sees). Shows which layers are far from the
matrix-core roofline — the residual 3x3 convs dominate the flops, the boundary / strided / PatchGAN layers the count.

    python tools/conv_table.py [--batch 8 --size 256 --steps 3] > profiles/r02_conv_table.txt
"""
import argparse
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ["GS_SIDE_STREAM"] = "0"

import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--workload", default="cyclegan", choices=["cyclegan", "pix2pix", "brats"])
    args = ap.parse_args()
    from ganslate_amd.utils.builders import build_gan
    torch.manual_seed(0)
    if args.workload == "pix2pix":          # BASELINE configs[2]: batch 1, 256 x 512
        args.batch = 1 if args.batch == 8 else args.batch
        model = build_gan(bench.make_pix2pix_conf(args.batch, 1000))
        shape = (args.batch, 3, 256, 512)
    elif args.workload == "brats":          # BASELINE configs[4]: V-Net CycleGAN on 128^3 patches, batch 1
        args.batch = 1 if args.batch == 8 else args.batch
        args.size = 128 if args.size == 256 else args.size
        model = build_gan(bench.make_volume_conf(args.batch, args.size, 1000, "vnet"))
        shape = (args.batch, 1, args.size, args.size, args.size)
    else:
        model = build_gan(bench.make_conf(args.batch, args.size, 1000))
        shape = (args.batch, 3, args.size, args.size)
    model.step_graph_enabled = False
    dev = model.device
    g = torch.Generator().manual_seed(1234)
    batch = {"A": (torch.rand(shape, generator=g) * 2 - 1).to(dev), "B": (torch.rand(shape, generator=g) * 2 - 1).to(dev)}

    def step():
        model.set_input(batch)
        model.optimize_parameters()
        model.update_learning_rate()

    for _ in range(3):
        step()
    ops = next(iter(model.networks.values())).ops
    flops = {}

    def select(kind, s, flag):
        if kind == "gconv":
            label = (f"conv{'+normsum' if flag else ''} in {s.Di}x{s.Hi}x{s.Wi}x{s.Ci} -> dom {s.Dc}x{s.Hc}x{s.Wc}x{s.Co} "
                     f"T={s.T} so={s.so} si={s.si} {s.border}")
            flops[label] = 2.0 * s.pixels * s.Co * s.T * s.Ci              # per image
        elif kind == "gconv_multi":
            c = s[0]
            taps = "/".join(str(g.T) for g in s)
            label = (f"conv{'+normsum' if flag else ''} x{len(s)} classes in {c.Di}x{c.Hi}x{c.Wi}x{c.Ci} -> dom {c.Dc}x{c.Hc}x{c.Wc}x{c.Co} "
                     f"T={taps} so={c.so} si={c.si} {c.border}")
            flops[label] = sum(2.0 * g.pixels * g.Co * g.T * g.Ci for g in s)
        else:
            label = (f"wgrad{' pair' if flag else ''} a {s.Da}x{s.Ha}x{s.Wa}x{s.P} g {s.Dg}x{s.Hg}x{s.Wg}x{s.Q} T={s.T} "
                     f"si={s.si} {s.border}")
            # the contraction runs over the DENSE side's pixels (for a stride-2 layer the gathered side has 4x as many, of which a
            # tap touches every fourth)
            flops[label] = 2.0 * s.Da * s.Ha * s.Wa * s.Q * s.T * s.P * (2 if flag else 1)
        return label
    ops.enable_kernel_timing(select)
    for _ in range(args.steps):
        step()
    res = ops.kernel_timing_result()
    imgs = ops.kernel_timing_images()        # average images per launch of a class: a twin launch covers both networks' images,
    ops.disable_kernel_timing()              # the D step's launches real + fake of both discriminators
    rows = sorted(((n / args.steps * ms, lab, n / args.steps, ms) for lab, (n, ms) in res.items()), reverse=True)
    tot = sum(r[0] for r in rows)
    dims = "x".join(str(v) for v in shape[2:])
    print(f"# conv launches of one {args.workload} training step, batch {args.batch} {dims}: {sum(r[2] for r in rows):.0f} "
          f"launches, {tot:.3f} ms (HIP events, one stream; includes the deterministic second-stage reductions of wgrad)")
    print(f"# {'ms/step':>8s} {'n/step':>6s} {'avg us':>8s} {'img/launch':>10s} {'TFLOP/s':>8s}  class")
    for t, lab, n, ms in rows:
        ni = imgs.get(lab) or args.batch
        print(f"  {t:8.3f} {n:6.1f} {ms * 1e3:8.1f} {ni:10.1f} {flops[lab] * ni / (ms * 1e-3) / 1e12:8.1f}  {lab}")


if __name__ == "__main__":
    main()
