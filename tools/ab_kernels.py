#!/usr/bin/env python
"""Interleaved A/B of library options on one kernel case, in ONE process (run-to-run and box-to-box scatter of the
micro-benchmark is +-3 %; cdna_hip_programming.md rule 24):
    python tools/ab_kernels.py --case rb --kind fwd --opt hconvw_persist=0,1 [--opt ...] --rounds 7 --iters 40
Prints median / min microseconds per setting."""
import argparse
import itertools
import statistics
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from ganslate_amd.hip.ops import HipOps  # noqa: E402
from ganslate_amd.nn.native.spec import lower  # noqa: E402
from tools.bench_kernels import CASES  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", default="rb")
    ap.add_argument("--kind", default="fwd", choices=["fwd", "dgrad", "wgrad", "wgrad_pair"])
    ap.add_argument("--opt", action="append", default=[], help="name=v1,v2,...")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--iters", type=int, default=40)
    args = ap.parse_args()
    ops = HipOps()
    dev = ops.device
    case = CASES[args.case]
    spec, N, sizes = case[0], case[1], case[2:]
    low = lower(spec, *sizes)
    x = torch.randn(N, *sizes, spec.cin_p, device=dev).to(torch.bfloat16)
    gy = torch.randn(N, *low.out_dims, spec.cout_p, device=dev).to(torch.bfloat16)
    x2, gy2 = torch.randn_like(x.float()).to(torch.bfloat16), torch.randn_like(gy.float()).to(torch.bfloat16)
    fpack = (torch.randn(low.fwd_index.size + 64, device=dev) * 0.05).to(torch.bfloat16)
    dpack = (torch.randn(low.dgrad_index.size + 64, device=dev) * 0.05).to(torch.bfloat16)
    bias = torch.zeros(spec.cout_p, device=dev)
    y = torch.empty(N, *low.out_dims, spec.cout_p, device=dev, dtype=torch.bfloat16)
    gx = torch.empty(N, *low.dgrad_dims, spec.cin_p, device=dev, dtype=torch.bfloat16)
    dw = torch.zeros(spec.master_numel, device=dev)
    a, gt = (gy, x) if spec.kind == "conv" else (x, gy)
    a2, gt2 = (gy2, x2) if spec.kind == "conv" else (x2, gy2)

    def run():
        if args.kind == "fwd":
            slots, offs = 0, []
            for g in low.fwd:
                offs.append(slots)
                slots += ops.stat_slots(g, N)
            part = torch.empty(N * slots * 2 * spec.cout_p, device=dev)
            for g, o in zip(low.fwd, offs):
                ops.gconv(g, x, fpack, bias, y, stats=part, stats_slots=slots, stats_slot0=o)
        elif args.kind == "dgrad":
            for g in low.dgrad:
                ops.gconv(g, gy, dpack, None, gx)
        elif args.kind == "wgrad":
            ops.wgrad(low.wgrad, a, gt, dw)
        else:
            ops.wgrad(low.wgrad, a, gt, dw, pair=(a2, gt2))

    names = [o.split("=")[0] for o in args.opt]
    values = [[int(v) for v in o.split("=")[1].split(",")] for o in args.opt]
    settings = list(itertools.product(*values)) if values else [()]
    times = {s: [] for s in settings}
    for r in range(args.rounds):
        for s in settings:
            for n, v in zip(names, s):
                ops.set_option(n, v)
            for _ in range(3):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(args.iters):
                run()
            e1.record()
            torch.cuda.synchronize()
            times[s].append(e0.elapsed_time(e1) / args.iters * 1e3)
    for s in settings:
        t = times[s]
        print(f"{args.case}_{args.kind} {dict(zip(names, s))}: median {statistics.median(t):7.1f} us  min {min(t):7.1f} us", flush=True)


if __name__ == "__main__":
    main()
