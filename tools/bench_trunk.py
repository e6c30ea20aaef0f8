#!/usr/bin/env python
"""The three residual-conv launches of the headline step (256 -> 256, 3x3, 64 x 64 maps) alone, back to back, in the forms
the step can run them: one launch per network at batch 8 (256 tiles, one per workgroup) against the twin launch over both
networks' images (512 tiles, two per workgroup). Times are per launch and per 8-image equivalent; HIP events on the launch
stream, random bf16 operands.

    python tools/bench_trunk.py [--iters 200] [--batch 8]
"""
import argparse
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from ganslate_amd.hip.ops import HipOps  # noqa: E402
from ganslate_amd.nn.native.spec import ConvSpec, lower  # noqa: E402
from ganslate_amd.nn.native.twin import Twin  # noqa: E402


def timed(fn, iters, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters        # us per call


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--size", type=int, default=64)
    args = ap.parse_args()
    ops = HipOps()
    import os
    if os.environ.get("GS_RING_DBG"):
        ops.set_option("ring_dbg", int(os.environ["GS_RING_DBG"]))
    dev = ops.device
    C, N, H = 256, args.batch, args.size
    spec = ConvSpec("conv", C, C, 3, 1, 1, pad_mode="reflect")
    low = lower(spec, H, H)
    flop = 2.0 * N * H * H * C * C * 9
    g = torch.Generator(device="cpu").manual_seed(1)
    rnd = lambda *s: torch.randn(*s, generator=g).to(torch.bfloat16).to(dev)
    x, gy, y, g2 = rnd(2 * N, H, H, C), rnd(2 * N, H, H, C), rnd(2 * N, H, H, C), rnd(2 * N, H, H, C)
    packs_f = (torch.randn(2, low.fwd_index.size + 64, generator=g) * 0.05).to(torch.bfloat16).to(dev)
    packs_d = (torch.randn(2, low.dgrad_index.size + 64, generator=g) * 0.05).to(torch.bfloat16).to(dev)
    bias = torch.zeros(2, C, device=dev)
    out = torch.empty(2 * N, H, H, C, dtype=torch.bfloat16, device=dev)
    slots = ops.stat_slots(low.fwd[0], N)
    part = torch.empty(2 * N * slots * 2 * C, device=dev)
    mr = torch.rand(2 * N * 2 * C, device=dev) + 0.5
    rows = []

    def fwd(n, twin):
        w = Twin(packs_f[0], packs_f[1]) if twin else packs_f[0]
        b = Twin(bias[0], bias[1]) if twin else bias[0]
        return lambda: ops.gconv(low.fwd[0], x[:n], w, b, out[:n], stats=part[:n * slots * 2 * C], stats_slots=slots)

    def ring(n, twin):
        w = Twin(packs_d[0], packs_d[1]) if twin else packs_d[0]
        plan = ops.fused_ring_plan(low.dgrad_ring, n, C)
        fz = {"y": y[:n], "mean_rstd": mr[:n * 2 * C], "g2": g2[:n], "partial": plan[1], "fold": 1, "fold_mode": "reflect",
              "act": "relu", "slope": 0.2}
        return lambda: ops.gconv(low.dgrad_ring, gy[:n], w, None, out[:n], fuse=fz)

    dy_out, tot_out = torch.empty_like(out), torch.empty_like(out)

    def ring_then_apply(n, twin):        # the two launches: fused data gradient, then the norm backward's apply pass on its sums
        w = Twin(packs_d[0], packs_d[1]) if twin else packs_d[0]
        plan = ops.fused_ring_plan(low.dgrad_ring, n, C, twin=twin)
        fz = {"y": y[:n], "mean_rstd": mr[:n * 2 * C], "g2": g2[:n], "partial": plan[1], "fold": 1, "fold_mode": "reflect",
              "act": "none", "slope": 0.2}

        def run():
            ops.gconv(low.dgrad_ring, gy[:n], w, None, out[:n], fuse=fz)
            ops.inorm_act_backward(out[:n], g2[:n], y[:n], mr[:n * 2 * C], dy_out[:n], tot_out[:n], fold=0, act="none", pre=plan)
        return run

    def ring_apply(n, twin):             # one launch (gs_gconv_ring_apply)
        w = Twin(packs_d[0], packs_d[1]) if twin else packs_d[0]
        plan = ops.fused_ring_plan(low.dgrad_ring, n, C, twin=twin)
        sync = ops.ring_apply_plan(low.dgrad_ring, n, C, twin=twin)
        fz = {"y": y[:n], "mean_rstd": mr[:n * 2 * C], "g2": g2[:n], "partial": plan[1], "fold": 1, "fold_mode": "reflect",
              "act": "none", "slope": 0.2}
        return lambda: ops.gconv_ring_apply(low.dgrad_ring, gy[:n], w, dy_out[:n], tot_out[:n], fz, sync)

    for label, make in (("forward", fwd), ("dgrad (ring, fused sums)", ring), ("dgrad + norm apply, 2 launches", ring_then_apply),
                        ("dgrad with the apply inside", ring_apply)):
        t1 = timed(make(N, False), args.iters)
        t2 = timed(make(2 * N, True), args.iters)
        if make is ring_apply:           # (needs the persistent grid: every workgroup resident)
            t3 = float("nan")
        else:
            ops.set_option("hconvw_persist", 0)
            t3 = timed(make(2 * N, True), args.iters)
            ops.set_option("hconvw_persist", 1)
        rows.append((label, t1, t2, t3))
        print(f"{label:28s} single batch {N}: {t1:6.1f} us ({flop / t1 * 1e-6:5.0f} TFLOP/s) | twin 2 x {N}: {t2:6.1f} us = "
              f"{t2 / 2:5.1f} per network ({2 * flop / t2 * 1e-6:5.0f} TFLOP/s) | twin, one tile per workgroup: {t3:6.1f} us = "
              f"{t3 / 2:5.1f}", flush=True)


if __name__ == "__main__":
    main()
