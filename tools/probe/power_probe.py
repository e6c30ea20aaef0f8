#!/usr/bin/env python
"""Does the residual-conv kernel run against the board's power cap? Loops one kernel form for a few seconds while sampling
rocm-smi (power, sclk): python tools/probe/power_probe.py [--seconds 3]"""
import argparse
import os
import subprocess
import sys
import threading
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
from ganslate_amd.hip.ops import HipOps  # noqa: E402
from ganslate_amd.nn.native.spec import ConvSpec, lower  # noqa: E402


def sample(stop, out):
    while not stop.is_set():
        try:
            r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp"], capture_output=True, text=True, timeout=5)
            keep = [l.strip() for l in r.stdout.splitlines() if any(k in l for k in ("Power", "sclk", "mclk", "junction", "fclk"))]
            out.append((time.time(), keep))
        except Exception as e:  # noqa: BLE001
            out.append((time.time(), [repr(e)]))
        time.sleep(0.2)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=3.0)
    args = ap.parse_args()
    ops = HipOps()
    dev = ops.device
    spec = ConvSpec("conv", 256, 256, 3, 1, 1, pad_mode="reflect")
    N, H, W = 8, 64, 64
    low = lower(spec, H, W)
    x = torch.randn(N, H, W, spec.cin_p, device=dev).to(torch.bfloat16)
    fpack = (torch.randn(low.fwd_index.size + 64, device=dev) * 0.05).to(torch.bfloat16)
    bias = torch.zeros(spec.cout_p, device=dev)
    y = torch.empty(N, *low.out_dims, spec.cout_p, device=dev, dtype=torch.bfloat16)
    slots = ops.stat_slots(low.fwd[0], N)
    part = torch.empty(N * slots * 2 * spec.cout_p, device=dev)
    r = subprocess.run(["rocm-smi", "--showmaxpower", "--showpower"], capture_output=True, text=True)
    print("\n".join(l for l in r.stdout.splitlines() if "ower" in l))
    for label, opt in (("hconvw (16 waves, phase-locked)", 0), ("hconvx (8 waves, self-pipelined)", 1),
                       ("hconvx + 4 loader waves", 100), ("idle", None)):
        if opt is not None:
            ops.set_option("hconvx", opt)
        stop, out = threading.Event(), []
        th = threading.Thread(target=sample, args=(stop, out))
        th.start()
        t0 = time.time()
        n = 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        while time.time() - t0 < args.seconds:
            if opt is None:
                time.sleep(0.05)
                continue
            for _ in range(200):
                ops.gconv_classes(low.fwd, x, fpack, bias, y, stats=part, stats_slots=slots, stats_slot0s=[0])
            n += 200
            torch.cuda.synchronize()
        e1.record()
        torch.cuda.synchronize()
        stop.set()
        th.join()
        ms = e0.elapsed_time(e1)
        print(f"== {label}: {n} launches, {1e3 * ms / max(n, 1):.1f} us per launch (incl. sync gaps)")
        for t, keep in out[-3:]:
            print("   ", " | ".join(keep))


if __name__ == "__main__":
    main()
