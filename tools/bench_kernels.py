#!/usr/bin/env python
"""Micro-benchmark of the hot kernels at BASELINE cfg2 layer shapes (run on the GPU box):
    python tools/bench_kernels.py [--iters 50] [--only rb_fwd]
Prints TFLOP/s per case measured with HIP events on the launch stream."""
import argparse
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from ganslate_amd.hip.ops import HipOps  # noqa: E402
from ganslate_amd.nn.native.spec import ConvSpec, lower  # noqa: E402

CASES = {
    # name: (spec, N, H, W)
    "rb": (ConvSpec("conv", 256, 256, 3, 1, 1, pad_mode="reflect"), 8, 64, 64),
    "rbk64": (ConvSpec("conv", 64, 256, 3, 1, 1, pad_mode="reflect"), 8, 64, 64),      # same tile, 9 / 18 K-steps:
    "rbk128": (ConvSpec("conv", 128, 256, 3, 1, 1, pad_mode="reflect"), 8, 64, 64),    # loop cost vs fixed cost
    "d2": (ConvSpec("conv", 128, 256, 3, 2, 1), 8, 128, 128),
    "d1": (ConvSpec("conv", 64, 128, 3, 2, 1), 8, 256, 256),
    "u1": (ConvSpec("convT", 256, 128, 3, 2, 1, 1), 8, 64, 64),
    "stem": (ConvSpec("conv", 3, 64, 7, 1, 3, pad_mode="reflect"), 8, 256, 256),
    "out": (ConvSpec("conv", 64, 3, 7, 1, 3, pad_mode="reflect"), 8, 256, 256),
    # the same boundary convs as the step runs them: W taps folded into channels (csrc/wfold.hip), 7 H taps left
    "stemw": (ConvSpec("conv", 3, 64, 7, 1, 3, pad_mode="reflect", wfold="in"), 8, 256, 256),
    "outw": (ConvSpec("conv", 64, 3, 7, 1, 3, pad_mode="reflect", wfold="out"), 8, 256, 256),
    "u2": (ConvSpec("convT", 128, 64, 3, 2, 1, 1), 8, 128, 128),
    "dc4": (ConvSpec("conv", 256, 512, 4, 1, 1), 8, 32, 32),
    "dc2": (ConvSpec("conv", 64, 128, 4, 2, 1), 8, 128, 128),       # PatchGAN k4 stride-2 layers
    "dc3": (ConvSpec("conv", 128, 256, 4, 2, 1), 8, 64, 64),
    # 3-D: Resnet3D residual conv at 128^3 / 4, Vnet3D coupling convs (halo-resident kernel)
    "rb3": (ConvSpec("conv", 256, 256, 3, 1, 1, pad_mode="replicate", dims=3), 1, 32, 32, 32),
    "v16": (ConvSpec("conv", 16, 16, 5, 1, 2, dims=3), 1, 128, 128, 128),
    "v16s": (ConvSpec("conv", 16, 16, 5, 1, 2, dims=3), 1, 64, 64, 64),
    "v32": (ConvSpec("conv", 32, 32, 5, 1, 2, dims=3), 1, 64, 64, 64),
    "v32b": (ConvSpec("conv", 32, 32, 5, 1, 2, dims=3), 1, 128, 128, 128),
    "v64": (ConvSpec("conv", 64, 64, 5, 1, 2, dims=3), 1, 32, 32, 32),
    "v32s": (ConvSpec("conv", 32, 32, 5, 1, 2, dims=3), 1, 32, 32, 32),
    "v64s": (ConvSpec("conv", 64, 64, 5, 1, 2, dims=3), 1, 16, 16, 16),
    # PatchGAN3D's last two layers (patchgan3d.py:50-60) on the 4 images of a discriminator update
    "p3l": (ConvSpec("conv", 256, 1, 4, 1, 1, dims=3), 4, 31, 31, 31),
    "p3m": (ConvSpec("conv", 128, 256, 4, 1, 1, dims=3), 4, 32, 32, 32),
    # U-Net(7, 128) innermost levels at 256 x 512 (pix2pix): weight streaming for 8 - 32 pixels
    "un7": (ConvSpec("conv", 1024, 1024, 4, 2, 1), 1, 4, 8),
    "un6": (ConvSpec("conv", 1024, 1024, 4, 2, 1), 1, 8, 16),
    "uu7": (ConvSpec("convT", 1024, 1024, 4, 2, 1, 0), 1, 2, 4),
    "uu6": (ConvSpec("convT", 2048, 1024, 4, 2, 1, 0), 1, 4, 8),
    # V-Net down / up convs (k2 s2)
    "vdn": (ConvSpec("conv", 16, 32, 2, 2, 0, dims=3), 1, 128, 128, 128),
    "vup": (ConvSpec("convT", 64, 16, 2, 2, 0, dims=3), 1, 64, 64, 64),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--only", default="")
    ap.add_argument("--batch", type=int, default=0, help="override the batch of every case (16 = a twin launch's geometry)")
    ap.add_argument("--opt", action="append", default=[], help="library option name=value (gs_set_option), repeatable")
    args = ap.parse_args()
    ops = HipOps()
    for o in args.opt:
        k, v = o.split("=")
        ops.set_option(k, int(v))
    dev = ops.device
    for name, case in CASES.items():
        spec, N, sizes = case[0], args.batch or case[1], case[2:]
        if args.only and not any(args.only in f"{name}_{k}" for k in ("fwd", "dgrad", "wgrad", "dgradf")):
            continue
        low = lower(spec, *sizes)
        in_px = 1
        for v in sizes:
            in_px *= v
        macs = N * low.out_pixels * spec.cout * spec.cin * spec.T if spec.kind == "conv" else \
            N * in_px * spec.cout * spec.cin * spec.T
        flop = 2.0 * macs
        x = torch.randn(N, *sizes, spec.cin_p, device=dev).to(torch.bfloat16)
        gy = torch.randn(N, *low.out_dims, spec.cout_p, device=dev).to(torch.bfloat16)
        fpack = (torch.randn(low.fwd_index.size + 64, device=dev) * 0.05).to(torch.bfloat16)
        dpack = (torch.randn(low.dgrad_index.size + 64, device=dev) * 0.05).to(torch.bfloat16)
        bias = torch.zeros(spec.cout_p, device=dev)
        y = torch.empty(N, *low.out_dims, spec.cout_p, device=dev, dtype=torch.bfloat16)
        gx = torch.empty(N, *low.dgrad_dims, spec.cin_p, device=dev, dtype=torch.bfloat16)
        dw = torch.zeros(spec.master_numel, device=dev)
        slots, offs = 0, []
        for g in low.fwd:
            offs.append(slots)
            slots += ops.stat_slots(g, N)
        part = torch.empty(N * slots * 2 * spec.cout_p, device=dev)
        a, gt = (gy, x) if spec.kind == "conv" else (x, gy)

        def fwd():       # the product's call: all parity classes of a stride-2 layer through gs_gconv_forward_multi
            ops.gconv_classes(low.fwd, x, fpack, bias, y, stats=part, stats_slots=slots, stats_slot0s=offs)

        def dgrad():
            ops.gconv_classes(low.dgrad, gy, dpack, None, gx)

        def wgrad():
            ops.wgrad(low.wgrad, a, gt, dw)

        kinds = [("fwd", fwd), ("dgrad", dgrad), ("wgrad", wgrad)]
        # fused data gradient (consumer's norm-backward sums in the epilogue): ring form where the library offers it
        plan = ops.fused_norm_plan(low.dgrad[0], N, spec.cin_p) if (spec.kind == "conv" and len(low.dgrad) == 1) else None
        if plan is not None:
            ring = ops.fused_ring_plan(low.dgrad_ring, N, spec.cin_p)
            yprev = torch.randn(N, *sizes, spec.cin_p, device=dev).to(torch.bfloat16)
            g2 = torch.randn(N, *sizes, spec.cin_p, device=dev).to(torch.bfloat16)
            mr = torch.ones(N * 2 * spec.cin_p, device=dev)
            gxr = torch.empty(N, *sizes, spec.cin_p, device=dev, dtype=torch.bfloat16)
            fz = {"y": yprev, "mean_rstd": mr, "g2": g2, "partial": (ring or plan)[1], "fold": low.dgrad_fold,
                  "fold_mode": spec.pad_mode if low.dgrad_fold else "reflect", "act": "relu", "slope": 0.2}

            def dgradf():
                if ring is not None:
                    ops.gconv(low.dgrad_ring, gy, dpack, None, gxr, fuse=fz)
                else:
                    ops.gconv(low.dgrad[0], gy, dpack, None, gx, fuse=fz)
            kinds.append(("dgradf", dgradf))

        for kind, fn in kinds:
            tag = f"{name}_{kind}"
            if args.only and args.only not in tag:
                continue
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(args.iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / args.iters
            print(f"{tag:12s} {ms * 1e3:9.1f} us  {flop / ms / 1e9:8.1f} TFLOP/s   ({flop / 1e9:.2f} GFLOP)", flush=True)


if __name__ == "__main__":
    main()
