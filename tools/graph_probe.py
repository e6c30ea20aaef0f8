#!/usr/bin/env python
"""Host and device cost of one captured CycleGAN step (run on the GPU box): per-replay host time of hipGraphLaunch, and
wall time per step with the device drained, next to the launch-by-launch step."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
from ganslate_amd.utils.builders import build_gan  # noqa: E402


def main():
    batch_size = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    model = build_gan(bench.make_conf(batch_size, 256, 10 ** 6))
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    batch = {k: (torch.rand((batch_size, 3, 256, 256), generator=g) * 2 - 1).to(dev) for k in ("A", "B")}

    def step():
        model.set_input(batch)
        model.optimize_parameters()

    for _ in range(4):
        step()
    torch.cuda.synchronize()
    for mode in ("graph", "eager"):
        model.step_graph_enabled = mode == "graph"
        step(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            step()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        # one step at a time with a drained device: pure latency of a step
        lat = []
        for _ in range(5):
            torch.cuda.synchronize()
            a = time.perf_counter(); step(); b = time.perf_counter(); torch.cuda.synchronize(); c = time.perf_counter()
            lat.append((b - a, c - a))
        print(f"{mode}: host {1e3 * (t1 - t0) / 10:.2f} ms/step, wall {1e3 * (t2 - t0) / 10:.2f} ms/step; drained: "
              f"host {1e3 * min(x[0] for x in lat):.2f} ms, step {1e3 * min(x[1] for x in lat):.2f} ms", flush=True)
    if model._graph is not None:
        gr = model._graph
        torch.cuda.synchronize()
        a = time.perf_counter(); gr.replay(); b = time.perf_counter(); torch.cuda.synchronize(); c = time.perf_counter()
        print(f"bare replay: host {1e3 * (b - a):.2f} ms, done after {1e3 * (c - a):.2f} ms")


if __name__ == "__main__":
    main()
