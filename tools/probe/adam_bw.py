#!/usr/bin/env python
"""HBM rate of gs_adam_step_dev against a plain torch copy / add at the same sizes: python tools/probe/adam_bw.py"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch  # noqa: E402
from ganslate_amd.hip.ops import HipOps  # noqa: E402


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    ops = HipOps()
    dev = ops.device
    hyper = torch.tensor([2e-4, 0.5, 0.999, 1e-8, 0.5, 0.0316], device=dev)
    for n in (2_800_000, 11_400_000, 54_400_000):
        p, g, m, v = (torch.randn(n, device=dev) for _ in range(4))
        v.abs_()
        c = torch.empty_like(p)
        for zg in (True, False):
            us = timed(lambda: ops.adam_step_dev(p, g, m, v, hyper, zero_grad=zg))
            by = (32 if zg else 28) * n
            print(f"n = {n:>10d}  adam zero_grad={zg!s:5s} {us:8.1f} us  {by / us / 1e6:6.2f} TB/s")
        us = timed(lambda: c.copy_(p))
        print(f"n = {n:>10d}  torch copy            {us:8.1f} us  {8 * n / us / 1e6:6.2f} TB/s")
        us = timed(lambda: torch.add(p, g, out=c))
        print(f"n = {n:>10d}  torch add             {us:8.1f} us  {12 * n / us / 1e6:6.2f} TB/s", flush=True)


if __name__ == "__main__":
    main()
