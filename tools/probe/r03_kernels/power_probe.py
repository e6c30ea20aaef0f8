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
    ap.add_argument("--ablations", action="store_true",
                    help="energy per launch of hconvx's ablated forms (needs GANSLATE_HIP_LIB=build/libganslate_hip_tl.so, `make timeline`)")
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
    cases = (("hconvw (16 waves, phase-locked)", 0), ("hconvx (8 waves, self-pipelined)", 1),
             ("hconvx + 4 loader waves", 100), ("idle", None))
    if args.ablations:
        cases = (("hconvw", 0), ("hconvw, all-zero input and weights", -1), ("hconvx", 1), ("hconvx, no LDS-DMA in the loop", 2),
                 ("hconvx, no fragment reads", 3), ("hconvx, neither (MFMA + prologue + epilogue)", 4),
                 ("hconvx, neither, no barrier", 8), ("idle", None))
    x0, f0 = x.clone(), fpack.clone()
    for label, opt in cases:
        if opt is not None:
            ops.set_option("hconvx", max(opt, 0))
            x.copy_(x0 if opt >= 0 else torch.zeros_like(x0))
            fpack.copy_(f0 if opt >= 0 else torch.zeros_like(f0))
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
        import re
        mid = out[2:-1] or out
        watts = [float(m.group(1)) for _, keep in mid for l in keep for m in [re.search(r"Power \(W\): ([0-9.]+)", l)] if m]
        clk = [int(m.group(1)) for _, keep in mid for l in keep for m in [re.search(r"sclk clock level: \d+: \((\d+)Mhz", l)] if m]
        if watts and n:
            w = sum(watts) / len(watts)
            print(f"   mean {w:.0f} W, sclk {sum(clk) / max(len(clk), 1):.0f} MHz, {w * ms / n:.1f} mJ per launch "
                  f"({(w - 288) * ms / n:.1f} mJ above idle)")
        for t, keep in out[-3:]:
            print("   ", " | ".join(keep))


if __name__ == "__main__":
    main()
