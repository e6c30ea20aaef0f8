#!/usr/bin/env python
"""Micro-benchmark of the InstanceNorm kernels at the residual-block shape (N=8, 64x64, C=256):
    python tools/bench_norm.py [--iters 50]
Prints microseconds and algorithmic GB/s per case (HIP events on the launch stream)."""
import argparse
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from ganslate_amd.hip.ops import HipOps  # noqa: E402


def timeit(fn, iters):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--rotate", type=int, default=8, help="distinct tensor sets cycled through (cold-ish caches)")
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--vol", type=int, default=0, help="3-D cases at N=1, vol^3, C=256 (replicate fold) instead of the 2-D ones")
    args = ap.parse_args()
    ops = HipOps()
    dev = ops.device
    N, H, W, C = 8, args.size, args.size, 256
    R = args.rotate
    mk = lambda *shape: [torch.randn(*shape, device=dev).to(torch.bfloat16) for _ in range(R)]
    y, x, res = mk(N, H, W, C), mk(N, H, W, C), mk(N, H, W, C)
    gpad, g0, dy, gsum, g2 = mk(N, H + 2, W + 2, C), mk(N, H, W, C), mk(N, H, W, C), mk(N, H, W, C), mk(N, H, W, C)
    part = torch.stack([y[0].float().sum((1, 2)), (y[0].float() ** 2).sum((1, 2))], 1).reshape(-1).contiguous()
    mr = torch.empty(N * 2 * C, device=dev)
    ops.inorm_finalize(part, N, 1, C, H * W, mr)
    mb = N * H * W * C * 2 / 1e6
    k = [0]

    def nxt():
        k[0] = (k[0] + 1) % R
        return k[0]

    cases = {
        "fwd act": (lambda i: ops.inorm_act_forward(y[i], mr, None, x[i], act="relu"), 2 * mb),
        "fwd act+res": (lambda i: ops.inorm_act_forward(y[i], mr, res[i], x[i], act="none"), 3 * mb),
        "bwd fold1": (lambda i: ops.inorm_act_backward(gpad[i], None, y[i], mr, dy[i], None, fold=1, act="relu"), 3 * mb * 2 - mb),
        "bwd fold1+g2+gsum": (lambda i: ops.inorm_act_backward(gpad[i], g2[i], y[i], mr, dy[i], gsum[i], fold=1, act="none"), 5 * mb * 2 - 3 * mb),
        "bwd nofold": (lambda i: ops.inorm_act_backward(g0[i], None, y[i], mr, dy[i], None, fold=0, act="relu"), 5 * mb),
    }
    if args.vol:       # the 3-D residual-block shape (Resnet3D at 128^3 / 4): replicate fold
        V = args.vol
        N = 1
        y, dy = mk(N, V, V, V, C), mk(N, V, V, V, C)
        gpad, g0 = mk(N, V + 2, V + 2, V + 2, C), mk(N, V, V, V, C)
        part = torch.stack([y[0].float().sum((1, 2, 3)), (y[0].float() ** 2).sum((1, 2, 3))], 1).reshape(-1).contiguous()
        mr = torch.empty(N * 2 * C, device=dev)
        ops.inorm_finalize(part, N, 1, C, V ** 3, mr)
        mb = N * V ** 3 * C * 2 / 1e6
        cases = {
            "3d bwd replicate fold": (lambda i: ops.inorm_act_backward(gpad[i], None, y[i], mr, dy[i], None, fold=1,
                                                                       fold_mode="replicate", act="relu"), 5 * mb),
            "3d bwd nofold": (lambda i: ops.inorm_act_backward(g0[i], None, y[i], mr, dy[i], None, fold=0, act="relu"), 5 * mb),
        }
    for name, (fn, traffic_mb) in cases.items():
        us = timeit(lambda: fn(nxt()), args.iters)
        print(f"{name:22s} {us:8.1f} us   {traffic_mb / us * 1e3 / 1e3:6.2f} TB/s algorithmic ({traffic_mb:.0f} MB)", flush=True)


if __name__ == "__main__":
    main()
