#!/usr/bin/env python
"""hconv.hip's forms of the 16 -> 16 channel 5x5x5 layer (Vnet3D coupling conv) at 128^3 and 64^3: us per launch.
python tools/probe/hconv_forms.py"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch  # noqa: E402
from ganslate_amd.hip.ops import HipOps  # noqa: E402
from ganslate_amd.nn.native.spec import ConvSpec, lower  # noqa: E402


def main():
    ops = HipOps()
    dev = ops.device
    for cin, cout, size in ((16, 16, 128), (16, 16, 64), (1, 16, 128)):
        spec = ConvSpec("conv", cin, cout, 5, 1, 2, dims=3)
        low = lower(spec, size, size, size)
        x = torch.randn(1, size, size, size, spec.cin_p, device=dev).to(torch.bfloat16)
        fpack = (torch.randn(low.fwd_index.size + 64, device=dev) * 0.05).to(torch.bfloat16)
        bias = torch.zeros(spec.cout_p, device=dev)
        y = torch.empty(1, *low.out_dims, spec.cout_p, device=dev, dtype=torch.bfloat16)
        for label, box8, persist in (("4x8x8 boxes, 4 waves", 0, 0), ("8x8x8 boxes, 8 waves", 1, 0), ("persistent", 0, 1)):
            ops.set_option("hconv_box8", box8)
            ops.set_option("hconv_persist", persist)
            slots = ops.stat_slots(low.fwd[0], 1)
            part = torch.empty(slots * 2 * spec.cout_p, device=dev)
            run = lambda: ops.gconv_classes(low.fwd, x, fpack, bias, y, stats=part, stats_slots=slots, stats_slot0s=[0])
            for _ in range(5):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(50):
                run()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 50
            flop = 2.0 * size ** 3 * spec.cin_p * spec.cout_p * 125
            print(f"{cin:3d} -> {cout} @ {size}^3  {label:24s} {us:8.1f} us  {flop / us / 1e6:7.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
