#!/usr/bin/env python
"""Host-side cost of enqueueing one training step (run on the GPU box): cProfile over a few CycleGAN steps at batch 1,
where the step is host-bound. Prints the top functions by own time."""
import cProfile
import pstats
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
from ganslate_amd.utils.builders import build_gan  # noqa: E402


def main():
    model = build_gan(bench.make_conf(1, 256, 10 ** 6))
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    batch = {k: (torch.rand((1, 3, 256, 256), generator=g) * 2 - 1).to(dev) for k in ("A", "B")}

    def step():
        model.set_input(batch)
        model.optimize_parameters()
        model.update_learning_rate()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(5):
        step()
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(28)


if __name__ == "__main__":
    main()
