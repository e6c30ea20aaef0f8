#!/usr/bin/env python
"""Board power and sustained clock under pure MFMA loops (tools/probe/mfma_power.hip) — what the matrix pipes can sustain
under this board's power management, per MFMA shape, with and without LDS operand reads:
    hipcc -O3 --offload-arch=gfx950 -o tools/probe/build/mfma_power tools/probe/mfma_power.hip; python tools/probe/mfma_power.py"""
import re
import subprocess
import sys
import threading
import time
from pathlib import Path

BIN = Path(__file__).resolve().parent / "build" / "mfma_power"


def sample(stop, out):
    while not stop.is_set():
        r = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=10)
        w = re.search(r"Power \(W\): ([0-9.]+)", r.stdout)
        c = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", r.stdout)
        out.append((time.time(), float(w.group(1)) if w else None, int(c.group(1)) if c else None))
        time.sleep(0.15)


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
    cases = [(0, 8, 100, 0), (0, 8, 100, 1), (1, 8, 100, 0), (1, 8, 100, 1), (2, 8, 100, 0), (2, 8, 100, 1), (3, 8, 100, 1),
             (0, 16, 100, 1), (2, 16, 100, 1), (0, 8, 50, 1), (2, 8, 50, 1)]
    for form, waves, duty, rnd in cases:
        stop, out = threading.Event(), []
        th = threading.Thread(target=sample, args=(stop, out))
        th.start()
        r = subprocess.run([str(BIN), str(form), str(secs), str(waves), str(duty), str(rnd)], capture_output=True, text=True)
        stop.set()
        th.join()
        mid = [s for s in out if s[1] is not None][2:-1] or out
        watts = [s[1] for s in mid if s[1]]
        clk = [s[2] for s in mid if s[2]]
        print(r.stdout.strip(), f"| {sum(watts) / max(len(watts), 1):7.1f} W  sclk {sum(clk) / max(len(clk), 1):6.0f} MHz  ({len(mid)} samples)",
              flush=True)
        time.sleep(1.0)


if __name__ == "__main__":
    main()
