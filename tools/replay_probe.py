#!/usr/bin/env python
"""What `host_enqueue_ms_per_step` in the bench line is made of: the host-side duration of ONE
`hipGraphLaunch` of the captured training step

  idle      the GPU has nothing queued (synchronised before the call): pure host work of the launch
  pipelined replays back to back without synchronising: the call returns when the runtime has room for it

plus the GPU duration of a replay. If the idle figure is small and the pipelined one approaches the GPU duration, the host
waits inside the launch for the previous replay (back-pressure), it does not work.

    python tools/replay_probe.py [--batch 8 --size 256 --n 20]
"""
import argparse
import statistics
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
from ganslate_amd.utils.builders import build_gan  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--n", type=int, default=20)
    args = ap.parse_args()
    model = build_gan(bench.make_conf(args.batch, args.size, 10 ** 6))
    dev = model.device
    g = torch.Generator().manual_seed(0)
    batch = {k: (torch.rand((args.batch, 3, args.size, args.size), generator=g) * 2 - 1).to(dev) for k in ("A", "B")}

    def step():
        model.set_input(batch)
        model.optimize_parameters()
        model.update_learning_rate()

    for _ in range(4):
        step()
    torch.cuda.synchronize()
    assert model._graph is not None, "the step was not captured"
    idle, gpu = [], []
    for _ in range(args.n):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        t0 = time.perf_counter()
        model._graph.replay()
        t1 = time.perf_counter()
        e1.record()
        torch.cuda.synchronize()
        idle.append((t1 - t0) * 1e3)
        gpu.append(e0.elapsed_time(e1))
    # the host part a real step adds around the replay (RNG draws, lr / step uploads)
    host_only = []
    for _ in range(args.n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model._prepare_host_state()
        t1 = time.perf_counter()
        host_only.append((t1 - t0) * 1e3)
    torch.cuda.synchronize()
    pipe = []
    t_all = time.perf_counter()
    for _ in range(args.n):
        t0 = time.perf_counter()
        model._graph.replay()
        pipe.append((time.perf_counter() - t0) * 1e3)
    t_enq = time.perf_counter() - t_all
    torch.cuda.synchronize()
    t_tot = time.perf_counter() - t_all
    med = statistics.median
    print(f"# one hipGraphLaunch of the captured CycleGAN step (batch {args.batch}, {args.size}^2), {args.n} samples, ms")
    print(f"idle GPU      : host call median {med(idle):.3f} (min {min(idle):.3f}, max {max(idle):.3f}); GPU duration of the replay "
          f"median {med(gpu):.3f}")
    print(f"host state    : _prepare_host_state median {med(host_only):.3f}")
    print(f"back to back  : host call median {med(pipe):.3f} (first {pipe[0]:.3f}, second {pipe[1]:.3f}, min {min(pipe):.3f}, "
          f"max {max(pipe):.3f}); all {args.n} enqueued after {t_enq * 1e3:.2f}, finished after {t_tot * 1e3:.2f}")


if __name__ == "__main__":
    main()
